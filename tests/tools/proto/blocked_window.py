"""Prototype (numpy float64, round 5) of two BLOCKED forms of the covariance recursion in the co-rotating frame
(lazy_frame.py; kfilter.cpp:189-215 applied a chunk of data at a time instead of one datum at a time).

Inside a window of the co-rotating frame a chunk of m data is a pure sequence of rank-1 downdates with
recursion-independent (h~_j, c~_j):   k~_j = S_{j-1} h~_j + c~_j ,  var_j = s0 + e_j + h~_j.S_{j-1} h~_j ,
S_j = S_{j-1} - k~_j k~_j^T / var_j  (= kfilter.cpp:191-210).

(1) loglik_ldl -- the chunk as ONE symmetric elimination (the round-4 review's formulation): with U = S0 H~ + C~ (p x m),
    M = upper triangle of U^T H~ + diag(e) (M_ik = h~_i.S0 h~_k + R(t_k - t_i): the chunk's predictive covariance), the
    var_j are the pivots of the LDL^T of M, the gains K = U L^-T, the innovations a forward substitution with L, and
    S_m = S0 - K D^-1 K^T, z~_m = z~_0 + K D^-1 innov.

(2) loglik_window -- the same elimination laid out as a wave would hold it: one LANE per datum of the chunk
    (kk = would-be gain, hh = h~, m = would-be variance, nu = would-be innovation) plus p VIRTUAL lanes holding the
    columns of S (hh = e_s, kk = S[:, s], nu = -z~_s): a pivot j sends (kk_j, m_j, nu_j) to every later lane, which does
        G = kk_j . hh ;  t = -G / m_j ;  m += G t ;  kk += kk_j t ;  nu += nu_j t
    -- for a virtual lane that IS the rank-1 downdate of its column of S and the update of z~_s, so S and z~ are never
    formed outside the lanes; the next chunk starts from  kk' = c~' + sum_s kk[lane v_s] hh'_s ,  m' = e' + hh'.kk'
    (s0 = h~.c~ enters through c~), nu' = (y - mu) + sum_s nu[lane v_s] hh'_s.  A chunk ends before a re-base datum.

Both are compared with the CPU oracle; see tests/test_blocked_window_proto.py and profiles/r05/blocked_proto_v1.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from real_modal import real_model, phi  # noqa: E402


def _frames(t, om, pairs, p, h, c, lim_re, lim_im):
    """Per datum: re-base flag, h~, c~ (at a re-base datum: h, c) and the rotation accumulated over the closing window."""
    amax, bmax = np.max(np.abs(om.real)), np.max(np.abs(om.imag))
    n = t.size
    reb = np.zeros(n, bool)
    ht, ct = np.zeros((n, p)), np.zeros((n, p))
    rot = [None] * n
    base = t[0]
    for k in range(n):
        dta = t[k] - base
        if k > 0 and (amax * dta > lim_re or bmax * dta > lim_im):
            reb[k] = True
            rot[k] = phi(om, pairs, dta, p)
            base = t[k]
            dta = 0.0
        ht[k] = phi(om, pairs, dta, p).T @ h
        ct[k] = phi(-om, pairs, dta, p) @ c
    return reb, ht, ct, rot


def _chunks(reb, n, mchunk):
    """Greedy chunks of at most mchunk data; a re-base datum always opens a chunk."""
    out, k = [], 0
    while k < n:
        j = k + 1
        while j < n and j - k < mchunk and not reb[j]:
            j += 1
        out.append((k, j))
        k = j
    return out


def loglik_ldl(t, y, yerr, theta, p, q, mchunk=16, lim_re=300.0, lim_im=262144.0, stats=None):
    om, h, Vz, pairs = real_model(theta, p, q)
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    c = Vz @ h
    s0 = h @ Vz @ h
    reb, ht, ct, rot = _frames(t, om, pairs, p, h, c, lim_re, lim_im)
    S, z, ll = np.zeros((p, p)), np.zeros(p), 0.0
    for (a, b) in _chunks(reb, t.size, mchunk):
        if reb[a]:
            S = rot[a] @ S @ rot[a].T
            z = rot[a] @ z
        m = b - a
        H, C = ht[a:b].T, ct[a:b].T                         # p x m
        U = S @ H + C
        M = np.triu(U.T @ H)                                  # upper triangle: U_i . h~_k, i <= k
        M[np.arange(m), np.arange(m)] = s0 + e[a:b] + np.einsum("ij,ij->j", H, S @ H)
        G = M.copy()                                          # right-looking elimination, rows = pivots
        K = U.copy()
        nu = yc[a:b] - H.T @ z
        var = np.zeros(m)
        for j in range(m):
            var[j] = G[j, j]
            r = 1.0 / var[j]
            for k in range(j + 1, m):
                tk = G[j, k] * r
                G[j + 1:k + 1, k] -= G[j, j + 1:k + 1] * tk   # M_ik -= G_ji G_jk / var_j, i <= k
                K[:, k] -= K[:, j] * tk
                nu[k] -= nu[j] * tk
        ll += np.sum(-0.5 * np.log(var) - 0.5 * nu * nu / var)
        S = S - (K / var) @ K.T
        z = z + K @ (nu / var)
    if stats is not None:
        stats.append(int(reb.sum()))
    return ll


def loglik_window(t, y, yerr, theta, p, q, mchunk=None, lim_re=300.0, lim_im=262144.0, stats=None):
    """Lane form; mchunk defaults to 16 - p (data lanes of a 16-lane row that also carries the p columns of S)."""
    if mchunk is None:
        mchunk = 16 - p
    om, h, Vz, pairs = real_model(theta, p, q)
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    c = Vz @ h
    reb, ht, ct, rot = _frames(t, om, pairs, p, h, c, lim_re, lim_im)
    # virtual lanes: column s of S in kk, -z~_s in nu, hh = e_s
    kkv, nuv, hhv = np.zeros((p, p)), np.zeros(p), np.eye(p)
    ll = 0.0
    for (a, b) in _chunks(reb, t.size, mchunk):
        if reb[a]:                                            # S <- A S A^T, z~ <- A z~ on the virtual lanes
            A = rot[a]
            Smat = A @ kkv.T @ A.T                            # kkv[s] = S[:, s]
            kkv = Smat.T.copy()
            nuv = A @ nuv
        m = b - a
        hh = ht[a:b].copy()                                   # [lane][component]
        # kk'_r = c~_r + sum_s kk_r[lane v_s] hh_s : one broadcast-FMA per (r, s)
        kk = ct[a:b].copy()
        for s in range(p):
            kk += np.outer(hh[:, s], kkv[s])
        mm = e[a:b] + np.einsum("lr,lr->l", hh, kk)           # s0 = h~.c~ comes in through c~
        nu = yc[a:b].copy()
        for s in range(p):
            nu += nuv[s] * hh[:, s]
        for j in range(m):
            var = mm[j]
            r = 1.0 / var
            # data lanes after the pivot
            G = hh[j + 1:] @ kk[j]
            tt = -G * r
            mm[j + 1:] += G * tt
            kk[j + 1:] += np.outer(tt, kk[j])
            nu[j + 1:] += nu[j] * tt
            # virtual lanes: G = kk_j[s]
            Gv = hhv @ kk[j]
            tv = -Gv * r
            kkv += np.outer(tv, kk[j])
            nuv += nu[j] * tv
        ll += np.sum(-0.5 * np.log(mm) - 0.5 * nu * nu / mm)
    if stats is not None:
        stats.append(int(reb.sum()))
    return ll
