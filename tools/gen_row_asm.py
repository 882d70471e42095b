#!/usr/bin/env python3
"""Generates carma_pack_amd/csrc/carma_row_asm.h: the three DPP blocks of filter_loop_row
(carma_core.h) as ONE inline-asm statement each, for p = 2..7.

Why one statement per block: LLVM's hazard recogniser counts inline asm as zero wait states, so
every asm-defined register consumed by a later asm statement costs an s_nop (4-8 cycles in a
single-wave instruction stream); inside one statement the instruction order is ours.  The DPP read
hazard (two wait states after the VALU write of the source) is covered by the two leading
instructions of each block, which the step needs anyway."""
import os

DPP = " row_newbcast:%d row_mask:0xf bank_mask:0xf"


def block_sums(P):
    # operands: 0 var, 1 innov | 2 e, 3 scale, 4 s0, 5 y, 6 mu, 7 w, 8 z, 9.. h_j
    lines = ["v_fma_f64 %0, %2, %3, %4", "v_add_f64 %1, %5, -%6"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%0, %%7, %%%d" % (9 + j) + DPP % j)
        lines.append("v_fmac_f64_dpp %%1, -%%8, %%%d" % (9 + j) + DPP % j)
    outs = '"=&v"(var), "=&v"(innov)'
    ins = '"v"(e), "v"(scale), "v"(s0), "v"(y), "v"(mu), "v"(w), "v"(z), ' + ", ".join('"v"(h[%d])' % j for j in range(P))
    return lines, outs, ins


def block_gain(P):
    # operands: 0 nt, 1 z, 2..2+P-1 D_j | k, s, si
    ik, is_, isi = 2 + P, 3 + P, 4 + P
    lines = ["v_mul_f64 %%0, %%%d, -%%%d" % (ik, is_), "v_fmac_f64 %%1, %%%d, %%%d" % (ik, isi)]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%0" % (2 + j, ik) + DPP % j)
    outs = '"=&v"(nt), "+v"(z), ' + ", ".join('"+v"(D[%d])' % j for j in range(P))
    ins = '"v"(k), "v"(s), "v"(si)'
    return lines, outs, ins


def block_colmix(P):
    # operands: 0..P-1 mm_j | P c, P+1 s, P+2.. D_j
    ic, is_, iD = P, P + 1, P + 2
    lines = ["v_mov_b64 %%%d, 0" % j for j in range(P)]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%%d" % (j, ic, iD + j) + DPP % j)
    for j in range(P & ~1):
        lines.append("v_fmac_f64_dpp %%%d, -%%%d, %%%d" % (j, is_, iD + (j ^ 1)) + DPP % j)
    outs = ", ".join('"=&v"(mm[%d])' % j for j in range(P))
    ins = '"v"(c), "v"(s), ' + ", ".join('"v"(D[%d])' % j for j in range(P))
    return lines, outs, ins


def block_sums_var(P):
    # operands: 0 var, 1 k | 2 e, 3 scale, 4 s0, 5 w, 6 c_own, 7.. h_j
    # the two leading instructions are the wait states for the DPP reads of w; the second one is the gain
    # k = w + c of the step (needed by the gain block further down, where it is then long written)
    lines = ["v_fma_f64 %0, %2, %3, %4", "v_add_f64 %1, %5, %6"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%0, %%5, %%%d" % (7 + j) + DPP % j)
    outs = '"=&v"(var), "=&v"(k)'
    ins = '"v"(e), "v"(scale), "v"(s0), "v"(w), "v"(c_own), ' + ", ".join('"v"(h[%d])' % j for j in range(P))
    return lines, outs, ins


def block_sums_innov(P):
    # operands: 0 innov | 1 y, 2 mu, 3 z, 4.. h_j
    lines = ["v_add_f64 %0, %1, -%2", "s_nop 0"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%0, -%%3, %%%d" % (4 + j) + DPP % j)
    outs = '"=&v"(innov)'
    ins = '"v"(y), "v"(mu), "v"(z), ' + ", ".join('"v"(h[%d])' % j for j in range(P))
    return lines, outs, ins


def block_gain_cov(P):
    # operands: 0 nt, 1..P D_j | k, s        (k is written by the var block, >= p + 4 instructions earlier)
    ik, is_ = 1 + P, 2 + P
    lines = ["v_mul_f64 %%0, %%%d, -%%%d" % (ik, is_)]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%0" % (1 + j, ik) + DPP % j)
    outs = '"=&v"(nt), ' + ", ".join('"+v"(D[%d])' % j for j in range(P))
    ins = '"v"(k), "v"(s)'
    return lines, outs, ins


# ---- split rows (carma_pipe3.h, covariance wave): row r of D lives in TWO lanes of the 16-lane DPP row, lane r
# (half A, bank mask 0x3) and lane 8 + r (half B, bank mask 0xc); root pairs alternate between the halves.
def split_map(P):
    half, slot, cnt = {}, {}, [0, 0]
    for j in range(P):
        hf = (j // 2) & 1
        half[j] = hf
        slot[j] = cnt[hf]
        cnt[hf] += 1
    return half, slot, max(cnt)


BANK = ["0x3", "0xc"]


def no_adjacent_same_dst(lines):
    """A bank-masked v_fmac_f64_dpp immediately followed by another access to the SAME destination register
    with the other bank mask returned stale data in the masked-off lanes (seen on gfx950 for p = 3: the lanes
    of half A lost the term written two instructions earlier).  One instruction in between is enough
    (p = 4, 5, 6, 7 have that by construction and pass); where the order cannot provide it, an s_nop does."""
    out = []
    prev_dst = None
    for l in lines:
        dst = l.split()[1].rstrip(",") if l.startswith("v_fmac_f64_dpp") else None
        if dst is not None and dst == prev_dst:
            out.append("s_nop 0")
        out.append(l)
        prev_dst = dst
    return out


def block_gain_split(P):
    # operands: 0 nt, 1..NS D_slot | k, s
    half, slot, ns = split_map(P)
    ik, is_ = 1 + ns, 2 + ns
    lines = ["v_mul_f64 %%0, %%%d, -%%%d" % (ik, is_)]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%0 row_newbcast:%d row_mask:0xf bank_mask:%s" % (1 + slot[j], ik, j, BANK[half[j]]))
    outs = '"=&v"(nt), ' + ", ".join('"+v"(D[%d])' % i for i in range(ns))
    ins = '"v"(k), "v"(s)'
    return no_adjacent_same_dst(lines), outs, ins


def block_colmix_split(P):
    # operands: 0..NS-1 mm_slot | NS c, NS+1 s, NS+2.. D_slot
    half, slot, ns = split_map(P)
    ic, is_, iD = ns, ns + 1, ns + 2
    lines = ["v_mov_b64 %%%d, 0" % i for i in range(ns)]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d row_mask:0xf bank_mask:%s" % (slot[j], ic, iD + slot[j], j, BANK[half[j]]))
    for j in range(P & ~1):
        lines.append("v_fmac_f64_dpp %%%d, -%%%d, %%%d row_newbcast:%d row_mask:0xf bank_mask:%s" % (slot[j], is_, iD + slot[j ^ 1], j, BANK[half[j]]))
    outs = ", ".join('"=&v"(mm[%d])' % i for i in range(ns))
    ins = '"v"(c), "v"(s), ' + ", ".join('"v"(D[%d])' % i for i in range(ns))
    return no_adjacent_same_dst(lines), outs, ins


# ---- co-rotating frame (carma_pipe3l.h): no rotation of the matrix per step; the per-step vectors ht = A^T h and
# ct = A^-1 c come from the producer waves.
def block_wsum_split(P):
    # operands: 0 wp0 (half A partial), 1 wp1 (half B partial) | 2 ht, 3.. S_slot
    # two accumulators, one per half, so that consecutive instructions never share a destination
    half, slot, ns = split_map(P)
    lines = ["v_mov_b64 %0, 0", "v_mov_b64 %1, 0"]
    order = sorted(range(P), key=lambda j: (slot[j], half[j]))
    for j in order:
        lines.append("v_fmac_f64_dpp %%%d, %%2, %%%d row_newbcast:%d row_mask:0xf bank_mask:%s" % (half[j], 3 + slot[j], j, BANK[half[j]]))
    outs = '"=&v"(wp0), "=&v"(wp1)'
    ins = '"v"(ht), ' + ", ".join('"v"(S[%d])' % i for i in range(ns))
    return no_adjacent_same_dst(lines), outs, ins


def block_sums_t(P):
    # operands: 0 var, 1 k, 2 t | 3 e, 4 scale, 5 s0, 6 w, 7 ct, 8 ht, 9 one
    # t = ht w ; var = |e| scale + s0 + sum_j t@j ; k = w + ct   (the sign of e carries the re-base flag)
    lines = ["v_mul_f64 %2, %8, %6", "v_fma_f64 %0, |%3|, %4, %5", "v_add_f64 %1, %6, %7"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %0, %2, %9" + DPP % j)
    outs = '"=&v"(var), "=&v"(k), "=&v"(t)'
    ins = '"v"(e), "v"(scale), "v"(s0), "v"(w), "v"(ct), "v"(ht), "v"(one)'
    return lines, outs, ins


def block_innov_t(P):
    # operands: 0 innov, 1 t | 2 y, 3 mu, 4 z, 5 ht, 6 one :  t = ht z ; innov = y - mu - sum_j t@j
    lines = ["v_mul_f64 %1, %5, %4", "v_add_f64 %0, %2, -%3", "s_nop 0"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %0, -%1, %6" + DPP % j)
    outs = '"=&v"(innov), "=&v"(t)'
    ins = '"v"(y), "v"(mu), "v"(z), "v"(ht), "v"(one)'
    return lines, outs, ins


def block_lazy_front(P):
    # one evaluation per row, lane r holds row r of S (no split).  operands:
    #   0 w, 1 var, 2 k, 3 t | 4 ht, 5 ct, 6 e (wave-uniform, SGPR), 7 scale, 8 s0, 9 one, 10.. S_j
    # w = sum_j S_j ht@j ; t = ht w ; k = w + ct ; var = |e| scale + s0 + sum_j t@j
    # Single accumulation chains: the wave is issue bound (tools/ubench/ub7.hip: a dependent v_fmac_f64_dpp costs 8
    # cycles, an independent one 6), so the second accumulators of a tree would only add their set-up instructions.
    lines = ["v_mov_b64 %0, 0", "v_fma_f64 %1, |%6|, %7, %8"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%0, %%4, %%%d" % (10 + j) + DPP % j)
    lines.append("v_mul_f64 %3, %4, %0")
    lines.append("v_add_f64 %2, %0, %5")
    lines.append("s_nop 0")
    for j in range(P):
        lines.append("v_fmac_f64_dpp %1, %3, %9" + DPP % j)
    outs = '"=&v"(w), "=&v"(var), "=&v"(k), "=&v"(t)'
    ins = '"v"(ht), "v"(ct), "s"(e), "v"(scale), "v"(s0), "v"(one), ' + ", ".join('"v"(S[%d])' % j for j in range(P))
    return lines, outs, ins


def block_gain_nt(P):
    # operands: 0..P-1 S_j | k, nt :  S_j += k@j nt
    lines = []
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, %%%d, %%%d" % (j, P, P + 1) + DPP % j)
    outs = ", ".join('"+v"(S[%d])' % j for j in range(P))
    ins = '"v"(k), "v"(nt)'
    return lines, outs, ins


def block_innov_t2(P):
    # operands: 0 innov, 1 t, 2 a1 | 3 y, 4 mu, 5 z, 6 ht, 7 one :  t = ht z ; innov = y - mu - sum_j t@j (two chains)
    lines = ["v_mul_f64 %1, %6, %5", "v_add_f64 %0, %3, -%4", "v_mov_b64 %2, 0"]
    for j in range(P):
        lines.append("v_fmac_f64_dpp %%%d, -%%1, %%7" % (0 if (j & 1) == 0 else 2) + DPP % j)
    lines.append("v_add_f64 %0, %0, %2")
    outs = '"=&v"(innov), "=&v"(t), "=&v"(a1)'
    ins = '"s"(y), "v"(mu), "v"(z), "v"(ht), "v"(one)'
    return lines, outs, ins


def emit(lines, outs, ins):
    body = "\n".join('            "%s\\n\\t"' % l for l in lines[:-1]) + '\n            "%s"' % lines[-1]
    return "        asm volatile(\n%s\n            : %s\n            : %s);\n" % (body, outs, ins)


out = ['// GENERATED by tools/gen_row_asm.py -- do not edit.  See that script and filter_loop_row (carma_core.h).',
       '#pragma once', '', 'namespace carma {', '', 'template <int P>', 'struct RowAsm;', '']
for P in range(2, 8):
    out.append('template <>')
    out.append('struct RowAsm<%d> {' % P)
    l, o, i = block_sums(P)
    out.append('    // var = e scale + s0 + sum_j h_j w@j ;  innov = y - mu - sum_j h_j z@j')
    out.append('    static __device__ __forceinline__ void sums(double& var, double& innov, double e, double scale, double s0,')
    out.append('                                                double y, double mu, double w, double z, const double (&h)[%d])' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_gain(P)
    out.append('    // nt = -(k s) ;  z += k si ;  D_j += k@j nt')
    out.append('    static __device__ __forceinline__ void gain(double& nt, double& z, double (&D)[%d], double k, double s, double si)' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_colmix(P)
    out.append('    // mm_j = c@j D_j - s@j D_{j^1}   (last column of an odd p: mm_j = c@j D_j)')
    out.append('    static __device__ __forceinline__ void colmix(double (&mm)[%d], double c, double s, const double (&D)[%d])' % (P, P))
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_sums_var(P)
    out.append('    // covariance wave of the 3-wave pipeline: var = e scale + s0 + sum_j h_j w@j ;  k = w + c_own')
    out.append('    static __device__ __forceinline__ void sums_var(double& var, double& k, double e, double scale, double s0, double w,')
    out.append('                                                    double c_own, const double (&h)[%d])' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_sums_innov(P)
    out.append('    // mean wave: innov = y - mu - sum_j h_j z@j')
    out.append('    static __device__ __forceinline__ void sums_innov(double& innov, double y, double mu, double z,')
    out.append('                                                      const double (&h)[%d])' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_gain_cov(P)
    out.append('    // covariance wave: nt = -(k s) ;  D_j += k@j nt')
    out.append('    static __device__ __forceinline__ void gain_cov(double& nt, double (&D)[%d], double k, double s)' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    half, slot, ns = split_map(P)
    out.append('    // split rows: column j of a row lives in half HALF[j] (0: lanes 0-7, 1: lanes 8-15) at register slot SLOT[j]')
    out.append('    static constexpr int NSLOT = %d;' % ns)
    out.append('    static constexpr int HALF[%d] = {%s};' % (P, ", ".join(str(half[j]) for j in range(P))))
    out.append('    static constexpr int SLOT[%d] = {%s};' % (P, ", ".join(str(slot[j]) for j in range(P))))
    l, o, i = block_gain_split(P)
    out.append('    static __device__ __forceinline__ void gain_split(double& nt, double (&D)[%d], double k, double s)' % ns)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_colmix_split(P)
    out.append('    static __device__ __forceinline__ void colmix_split(double (&mm)[%d], double c, double s, const double (&D)[%d])' % (ns, ns))
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_wsum_split(P)
    out.append('    // co-rotating frame: partial row sums wp0 (half A) / wp1 (half B) of S ht@column')
    out.append('    static __device__ __forceinline__ void wsum_split(double& wp0, double& wp1, double ht, const double (&S)[%d])' % ns)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_sums_t(P)
    out.append('    // t = ht w ;  var = |e| scale + s0 + sum_j t@j ;  k = w + ct')
    out.append('    static __device__ __forceinline__ void sums_t(double& var, double& k, double& t, double e, double scale, double s0,')
    out.append('                                                  double w, double ct, double ht, double one)')
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_innov_t(P)
    out.append('    // t = ht z ;  innov = y - mu - sum_j t@j')
    out.append('    static __device__ __forceinline__ void innov_t(double& innov, double& t, double y, double mu, double z, double ht,')
    out.append('                                                   double one)')
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_lazy_front(P)
    out.append('    // co-rotating frame, whole rows: w = sum_j S_j ht@j ; t = ht w ; k = w + ct ; var = |e| scale + s0 + sum_j t@j')
    out.append('    static __device__ __forceinline__ void lazy_front(double& w, double& var, double& k, double ht, double ct, double e,')
    out.append('                                                      double scale, double s0, double one, const double (&S)[%d])' % P)
    out.append('    {')
    out.append('        double t;')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_gain_nt(P)
    out.append('    // S_j += k@j nt')
    out.append('    static __device__ __forceinline__ void gain_nt(double (&S)[%d], double k, double nt)' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = block_innov_t2(P)
    out.append('    // t = ht z ;  innov = y - mu - sum_j t@j   (two accumulation chains)')
    out.append('    static __device__ __forceinline__ void innov_t2(double& innov, double y, double mu, double z, double ht, double one)')
    out.append('    {')
    out.append('        double t, a1;')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    out.append('};')
    out.append('')
out.append('}  // namespace carma')
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "carma_pack_amd", "csrc", "carma_row_asm.h")
open(path, "w").write("\n".join(out) + "\n")
print("wrote", os.path.normpath(path))
