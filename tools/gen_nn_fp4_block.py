#!/usr/bin/env python3
"""Writes orb_slam2_amd/csrc/nn_fp4_block.inc: the body of ONE inline-asm statement that runs a whole superstep of the FP4 Hamming scan
(k_hamming_nn_fp4b, orbhip_kernels_match.hip) with its instruction order assigned by hand.

Why a generated text: hipcc gathers the sixteen matrix instructions of a tile and puts the threshold tests (v_max3_f32 chains) and the tile's LDS operand
reads in front of them whatever the source says (profiles/r06_exp_config5_ablation.txt): the matrix pipe idles while a wave tests.  Inside one asm statement
the order is the text's.  The statement keeps its accumulators and tile operands in registers it CLOBBERS (named here: v[V0 .. 255]), takes the sixteen query
operand tuples, the four thresholds, the two block scales and the LDS address as inputs and returns one scalar: bit 8 t + u = "tile u of the superstep may hold
a row that matters to query tile t" - the rare kept (tile, query tile) pairs are recomputed and folded in by compiled code behind the statement.

Schedule (TPB tiles, each two HALF tiles of two query tiles = eight matrix instructions on two accumulators, alternating):
  step s:  matrix instructions of half tile s, between them the tests of half tile s - 2 (three accumulator pairs in rotation: a pair is read a whole
           half tile - eight matrix instructions - after its last write, far beyond the 11 wait states a VALU read of a matrix result needs, and written again
           only after its tests have issued); operand kb of the NEXT tile is read into the same four registers behind the last matrix instruction of this
           tile that uses them (second half, instruction 2 kb + 1) and waited for with a counted lgkmcnt in front of its first use - one operand set, not two:
           the sixteen registers decide whether hipcc spills around the statement.
Operands: %0 keep (s, out), %1 scalar scratch, %2 + 4 t + kb the query operand tuple (t, kb), %18 + t the threshold of query tile t, %22 (s) the LDS byte
address of the superstep's first tile.  (Tried and dropped, profiles/r06_exp_config5_superstep.txt: the next superstep's LDS-DMA requests inside the statement,
one per tile in the shadow of a matrix instruction, instead of eight in a burst behind the barrier - the same 3.68 ms.)
"""
import os
import sys

TPB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V0 = 139                      # first clobbered VGPR (3 + 96 accumulator + 16 operand + 2 scratch registers end at v255)
SCA, SCB = V0, V0 + 1         # the two block scales (127 = 2^0 for the rows, 133 = 2^6 for the queries)
ADDR = V0 + 2                 # the lane's LDS byte address, made here from the wave-uniform base (as an input it was spilled - and its reload's vmcnt(0) waited for the prefetch)
ACC = V0 + 3                  # three pairs of 16-register accumulators: 96 registers
AOP = ACC + 96                # ONE set of four 4-register tile operands: operand kb of the next tile is read as soon as this tile's last matrix instruction on it has issued
TMP = AOP + 16                # two running maxima
assert TMP + 2 <= 256


def acc(pair, j):
    b = ACC + 32 * pair + 16 * j
    return b


def vr(b, n):
    return f"v[{b}:{b + n - 1}]"


def mfma(dst, aop, bop, first):
    c = "0" if first else vr(dst, 16)
    return f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(dst, 16)}, {vr(aop, 4)}, {bop}, {c}, v{SCA}, v{SCB} op_sel_hi:[0,0,0] cbsz:4 blgp:4"


def half_mfmas(s, one_operand_set=False):
    """the eight matrix instructions of half tile s: tile u = s // 2, query tiles 2 h and 2 h + 1, alternating accumulators"""
    u, h = divmod(s, 2)
    if one_operand_set:
        u = 0
    out = []
    for kb in range(4):
        for j in range(2):
            t = 2 * h + j
            out.append(mfma(acc(s % 3, j), AOP + 4 * kb, f"%{2 + 4 * t + kb}", kb == 0))
    return out


def half_tests(s):
    """the two threshold tests of half tile s, their chains interleaved: 8 + 8 maxima, then compare + scalar bit each"""
    u, h = divmod(s, 2)
    chains = []
    for j in range(2):
        x = acc(s % 3, j)
        m = f"v{TMP + j}"
        c = [f"v_max3_f32 {m}, v{x}, v{x + 1}, v{x + 2}"]
        for i in range(3, 15, 2):
            c.append(f"v_max3_f32 {m}, {m}, v{x + i}, v{x + i + 1}")
        c.append(f"v_max_f32_e32 {m}, {m}, v{x + 15}")
        chains.append(c)
    out = []
    for a, b in zip(*chains):
        out += [[a], [b]]
    for j in range(2):
        t = 2 * h + j
        out.append([f"v_cmp_gt_f32_e32 vcc, v{TMP + j}, %{18 + t}", "s_cmp_lg_u64 vcc, 0", f"s_cselect_b32 %1, {1 << (8 * t + u):#x}, 0", "s_or_b32 %0, %0, %1"])
    return out                  # 18 groups; a group stays together


def read(u, kb):
    return f"ds_read_b128 {vr(AOP + 4 * kb, 4)}, v{ADDR} offset:{4096 * u + 1024 * kb}"


def reads(u):
    return [read(u, kb) for kb in range(4)]


def body(var):
    """var 0: the production text.  Measurement only (ORBHIP_NN_BLOCK_VAR, wrong answers): 1 = no tests, 3 = matrix instructions alone (the first tile's
    operands reused for every tile)"""
    lines = ["s_mov_b32 %0, 0", f"v_mbcnt_lo_u32_b32 v{ADDR}, -1, 0", f"v_mbcnt_hi_u32_b32 v{ADDR}, -1, v{ADDR}", f"v_mov_b32_e32 v{SCA}, 0x7f", f"v_mov_b32_e32 v{SCB}, 0x85"]
    lines += [f"v_lshl_add_u32 v{ADDR}, v{ADDR}, 4, %22"] + reads(0)
    NH = 2 * TPB
    for s in range(NH + 2):
        mm = half_mfmas(s, var == 3) if s < NH else []
        tt = half_tests(s - 2) if s >= 2 else []
        if var in (1, 3):
            tt = []
        u, h = divmod(s, 2)
        if not mm:
            for g in tt:
                lines += g
            continue
        n = len(tt)
        done = 0
        for i, m in enumerate(mm):
            # operand kb = i // 2 of this tile: read behind the second-half matrix instruction 2 kb + 1 of the tile before (LDS returns in order: three, two,
            # one, no younger reads may still be out when operand 0, 1, 2, 3 is first used)
            if h == 0 and i % 2 == 0 and (var != 3 or s == 0):
                lines.append(f"s_waitcnt lgkmcnt({(3 - i // 2) if s > 0 else 0})")
            lines.append(m)
            if h == 1 and i % 2 == 1 and u + 1 < TPB and var != 3:
                lines.append(read(u + 1, i // 2))
            upto = (n * (i + 1) + len(mm) - 1) // len(mm)
            for g in tt[done:upto]:
                lines += g
            done = upto
    return lines


clob = ", ".join(f'"v{i}"' for i in range(V0, 256))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb_slam2_amd", "csrc", "nn_fp4_block.inc")
with open(out, "w") as f:
    f.write(f"// generated by tools/gen_nn_fp4_block.py {TPB} - the hand-ordered superstep of k_hamming_nn_fp4b; do not edit\n")
    f.write(f"#define NN_FP4B_TPB {TPB}\n#define NN_FP4B_V0 {V0}\n")
    for var in (0, 1, 3):
        lines = [".p2align 6"] + body(var)
        f.write(f'#define NN_FP4B_BODY{var if var else ""} "' + "\\n\\t".join(lines) + '"\n')
    f.write(f"#define NN_FP4B_CLOBBERS {clob}\n")
print(out, len(body(0)), "instructions")
