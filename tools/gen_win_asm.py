#!/usr/bin/env python3
"""Generates carma_pack_amd/csrc/carma_win_asm.h: the two instruction blocks of the WINDOWED recursion wave
(carma_pipew.h) as ONE inline-asm statement each, for p = 2..7.

Layout (one evaluation per 16-lane DPP row): lanes 0 .. ND-1 (ND = 16 - P) hold the data of a chunk -- kk[r] the
would-be gain, hh[r] = h~_r, m the would-be variance, nu the would-be innovation --, lanes ND .. 15 hold the P columns
of S (hh = e_s, kk[r] = S_rs, nu = -z~_s).  "x@j" = the value of x in lane j of the row (DPP row_newbcast on src0).

chunk():  for j = 0 .. ND-1  (EXEC = lanes >= j of every row, so that a finished lane keeps its variance / innovation in
          the register its own pivot READ: m and nu alternate between two registers, pivot j reads A/B for j even/odd)
              G = sum_r kk_r@j hh_r ;  t = -G / m@j (v_rcp_f64 + one Newton step, folded into t) ;
              m' = m + G t ;  nu' = nu + nu@j t ;  kk_r += kk_r@j t
          EXEC is saved on entry and restored at the end.
          ALL pivots are one statement: between two statements the compiler may place copies, and a copy executed under
          the reduced EXEC would not reach the finished lanes.
init():   kn_r += sum_s kk_r@(ND+s) hn_s   (kn_r preloaded with c~_r: k~ = S h~ + c~, kfilter.cpp:191 -- or with 0 at a re-base,
                                            where hn is A^T h~ and the caller forms c~ + A kn afterwards: carma_pipew.h)
          nun  += sum_s nuF@(ND+s) hn_s    (nun preloaded with y - mu: innovation, kfilter.cpp:213)

Hazards covered by the instruction order (the assembler inserts nothing): a DPP source is read no sooner than two
instructions after the VALU that wrote it; the result of v_rcp_f64 (a TRANS op) is not consumed by the next instruction."""
import os
import sys

DPP = " row_newbcast:%d row_mask:0xf bank_mask:0xf"


def chunk_block(P):
    ND = 16 - P
    # operands: 0 G 1 rb 2 bnu 3 r0 4 e 5 t | 6 mA 7 mB 8 nuA 9 nuB | 10.. kk_r | 10+P saved EXEC | 11+P.. hh_r
    ikk, isave, ihh = 10, 10 + P, 11 + P
    # (EXEC is narrowed per pivot; the statement SAVES it on entry and RESTORES it at the end -- not "-1" -- so that a caller under a
    # partial EXEC gets its mask back: round-5 advice)
    L = ["s_mov_b64 %%%d, exec" % isave]
    for j in range(ND):
        cur_m, alt_m, cur_n, alt_n = (6, 7, 8, 9) if j % 2 == 0 else (7, 6, 9, 8)
        mask = (0xFFFF << j) & 0xFFFF
        m32 = mask | (mask << 16)
        L += ["s_mov_b32 exec_lo, 0x%08x" % m32, "s_mov_b32 exec_hi, 0x%08x" % m32,
              "v_mov_b64 %0, 0",
              "v_mov_b64_dpp %%1, %%%d" % cur_m + DPP % j,
              "v_mov_b64_dpp %%2, %%%d" % cur_n + DPP % j]
        g = ["v_fmac_f64_dpp %%0, %%%d, %%%d" % (ikk + r, ihh + r) + DPP % j for r in range(P)]
        # rcp after the first G term, the Newton residual two instructions later
        seq = [g[0], "v_rcp_f64 %3, %1"]
        rest = g[1:]
        if len(rest) >= 2:
            seq += rest[:2] + ["v_fma_f64 %4, -%1, %3, 1.0"] + rest[2:]
        else:
            seq += rest + ["s_nop 0"] * (2 - len(rest)) + ["v_fma_f64 %4, -%1, %3, 1.0"]
        L += seq
        L += ["v_mul_f64 %5, %0, -%3",
              "v_fma_f64 %5, %5, %4, %5",
              "v_fma_f64 %%%d, %%0, %%5, %%%d" % (alt_m, cur_m),
              "v_fma_f64 %%%d, %%2, %%5, %%%d" % (alt_n, cur_n)]
        for r in range(P):
            L.append("v_fmac_f64_dpp %%%d, %%%d, %%5" % (ikk + r, ikk + r) + DPP % j)
        if P < 3:
            L.append("s_nop %d" % (2 - P))          # kk_r written -> DPP read in the next pivot: keep two wait states
    L.append("s_mov_b64 exec, %%%d" % isave)
    outs = ['"=&v"(G)', '"=&v"(rb)', '"=&v"(bnu)', '"=&v"(r0)', '"=&v"(e)', '"=&v"(t)', '"+v"(mA)', '"+v"(mB)', '"+v"(nuA)', '"+v"(nuB)']
    outs += ['"+v"(kk[%d])' % r for r in range(P)]
    outs += ['"=&s"(exec_save)']
    ins = ['"v"(hh[%d])' % r for r in range(P)]
    return L, ", ".join(outs), ", ".join(ins)


def init_block(P):
    ND = 16 - P
    # operands: 0..P-1 kn_r | P nun | ins: kk_r (P), nuF, hn_s (P)
    ikk, inuF, ihn = P + 1, 2 * P + 1, 2 * P + 2
    L = []
    for s in range(P):                                   # s outer: P independent accumulation chains at a time
        for r in range(P):
            L.append("v_fmac_f64_dpp %%%d, %%%d, %%%d" % (r, ikk + r, ihn + s) + DPP % (ND + s))
        L.append("v_fmac_f64_dpp %%%d, %%%d, %%%d" % (P, inuF, ihn + s) + DPP % (ND + s))
    outs = ['"+v"(kn[%d])' % r for r in range(P)] + ['"+v"(nun)']
    ins = ['"v"(kk[%d])' % r for r in range(P)] + ['"v"(nuF)'] + ['"v"(hn[%d])' % s for s in range(P)]
    return L, ", ".join(outs), ", ".join(ins)


def emit(lines, outs, ins, clobber=None):
    body = "\n".join('            "%s\\n\\t"' % l for l in lines[:-1]) + '\n            "%s"' % lines[-1]
    tail = "" if clobber is None else "\n            : %s" % clobber
    return "        asm volatile(\n%s\n            : %s\n            : %s%s);\n" % (body, outs, ins, tail)


out = ['// GENERATED by tools/gen_win_asm.py -- do not edit.  See that script and carma_pipew.h.',
       '#pragma once', '', 'namespace carma {', '', 'template <int P>', 'struct WinAsm;', '']
for P in range(2, 8):
    out.append('template <>')
    out.append('struct WinAsm<%d> {' % P)
    l, o, i = chunk_block(P)
    out.append('    // the %d pivots of a chunk (see tools/gen_win_asm.py)' % (16 - P))
    out.append('    static __device__ __forceinline__ void chunk(double (&kk)[%d], const double (&hh)[%d], double& mA, double& mB, double& nuA,' % (P, P))
    out.append('                                                 double& nuB)')
    out.append('    {')
    out.append('        double G, rb, bnu, r0, e, t;')
    out.append('        unsigned long long exec_save;')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    l, o, i = init_block(P)
    out.append('    // start of the next chunk from the columns of S in lanes %d .. 15' % (16 - P))
    out.append('    static __device__ __forceinline__ void init(double (&kn)[%d], double& nun, const double (&kk)[%d], double nuF,' % (P, P))
    out.append('                                                const double (&hn)[%d])' % P)
    out.append('    {')
    out.append(emit(l, o, i).rstrip("\n"))
    out.append('    }')
    out.append('};')
    out.append('')
out.append('}  // namespace carma')
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "carma_pack_amd", "csrc", "carma_win_asm.h")
text = "\n".join(out) + "\n"
if "--check" in sys.argv:                     # (tests: the committed header IS this script's output; nothing is written)
    ok = os.path.exists(path) and open(path).read() == text
    print("carma_win_asm.h", "is up to date" if ok else "DIFFERS from the generator's output")
    sys.exit(0 if ok else 1)
open(path, "w").write(text)
print("wrote", os.path.normpath(path))
