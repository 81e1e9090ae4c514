#!/usr/bin/env python3
"""Builds ab/liborbhip_qttrace.so: the in-tree kernels with s_memrealtime stamps in k_quadtree's phases (read by tools/qt_trace_experiment.py through
orbhip_debug_qt_trace).  The stamps are inserted into a COPY of orbhip_kernels_extract.hip; the product source carries none.  Run after `make -C orb_slam2_amd/csrc`."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); C = os.path.join(R, "orb_slam2_amd", "csrc")
s = open(os.path.join(C, "orbhip_kernels_extract.hip")).read()
def rep(old, new):
    global s
    assert s.count(old) >= 1, old[:70]
    s = s.replace(old, new, 1)
rep("#define QT_T 256\n", "#define QT_T 256\n__device__ unsigned long long g_qt_trace[64 * 8];\n#define QT_STAMP(i) do { if (tid == 0 && frame == P.frame0) g_qt_trace[level * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n"
    "extern \"C\" void orbhip_debug_qt_trace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_trace), sizeof(unsigned long long) * 64 * 8); }\n")
rep("    const int tid = threadIdx.x;\n    const LevelGeom g = P.geom[level];\n    const int maxn = P.qt_maxn;\n    QtLds L;", "    const int tid = threadIdx.x;\n    QT_STAMP(0);\n    const LevelGeom g = P.geom[level];\n    const int maxn = P.qt_maxn;\n    QtLds L;")
rep("    n = min(n, g.cand_total_cap);\n", "    n = min(n, g.cand_total_cap);\n    QT_STAMP(1);\n")
rep("    __syncthreads();\n    keys.each_loaded(n, tid, qval,", "    __syncthreads();\n    QT_STAMP(2);\n    keys.each_loaded(n, tid, qval,")
rep("    __syncthreads();\n    // ---- B'. the regular passes in one step.", "    __syncthreads();\n    QT_STAMP(3);\n    // ---- B'. the regular passes in one step.")
rep("    // ---- C. passes\n", "    QT_STAMP(4);\n    int npass = 0;\n    // ---- C. passes\n")
rep("    for (int guard = 0; guard < 4096; guard++) {\n        keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {\n            const int p = knode;",
    "    for (int guard = 0; guard < 4096; guard++) {\n        if (npass < 30) QT_STAMP(5 + npass); npass++;\n        keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {\n            const int p = knode;")
rep("    // ---- D. best response per leaf, first wins (:744-760); list order = output order\n",
    "    QT_STAMP(44);\n    if (tid == 0 && frame == P.frame0) { g_qt_trace[level * 64 + 62] = (unsigned long long)npass; g_qt_trace[level * 64 + 63] = (unsigned long long)n; g_qt_trace[level * 64 + 61] = (unsigned long long)K; g_qt_trace[level * 64 + 60] = (unsigned long long)m; }\n"
    "    // ---- D. best response per leaf, first wins (:744-760); list order = output order\n")
rep("    if (tid == 0) P.lvl_n[frame * P.nlevels + level] = mout;\n    (void)wave; (void)lane;\n}", "    if (tid == 0) P.lvl_n[frame * P.nlevels + level] = mout;\n    QT_STAMP(45);\n    (void)wave; (void)lane;\n}")
# finer stamps inside the final-phase pass (slots 20..): after every barrier-separated step
steps = [
 ("        __syncthreads();\n        int Ctot, nsplit;\n", "        __syncthreads();\n        QT_STAMP(20);\n        int Ctot, nsplit;\n"),
 ("            const int E = qt_block_exscan(s_best, m, s_scratch, tid);     // thread t scans the flags thread t wrote\n", "            const int E = qt_block_exscan(s_best, m, s_scratch, tid);     // thread t scans the flags thread t wrote\n            QT_STAMP(21);\n"),
 ("                s_sidx[rank] = s_a[j];\n            }\n            __syncthreads();\n", "                s_sidx[rank] = s_a[j];\n            }\n            QT_STAMP(22);\n            __syncthreads();\n            QT_STAMP(23);\n"),
 ("            const int Call = qt_block_exscan(s_a, E, s_scratch, tid);     // s_a[j] = children created before sorted node j\n", "            QT_STAMP(24);\n            const int Call = qt_block_exscan(s_a, E, s_scratch, tid);     // s_a[j] = children created before sorted node j\n            QT_STAMP(25);\n"),
 ("            const int jstar = s_misc[2];\n", "            QT_STAMP(26);\n            const int jstar = s_misc[2];\n"),
 ("            qt_block_exscan(s_best, m, s_scratch, tid);\n            for (int p = tid; p < m; p += QT_T) {\n                if (s_split[p]) {\n                    int q = s_a[s_split[p] - 1];", "            QT_STAMP(27);\n            qt_block_exscan(s_best, m, s_scratch, tid);\n            QT_STAMP(28);\n            for (int p = tid; p < m; p += QT_T) {\n                if (s_split[p]) {\n                    int q = s_a[s_split[p] - 1];"),
 ("        __syncthreads();\n        const int m2 = Ctot + (m - nsplit);\n", "        QT_STAMP(29);\n        __syncthreads();\n        QT_STAMP(30);\n        const int m2 = Ctot + (m - nsplit);\n"),
 ("        int nexp = 0;\n        for (int p = tid; p < m2; p += QT_T) nexp += cnt2[p] > 1;\n", "        QT_STAMP(31);\n        int nexp = 0;\n        for (int p = tid; p < m2; p += QT_T) nexp += cnt2[p] > 1;\n"),
 ("        const int nToExpand = s_misc[3 + par];\n", "        QT_STAMP(32);\n        const int nToExpand = s_misc[3 + par];\n"),
]
for a, b in steps: rep(a, b)
# ... and inside the regular-pass jump (slots 33..)
jsteps = [
 ("            const int ncD = g.nIni << (2 * D);\n", "            QT_STAMP(33);\n            const int ncD = g.nIni << (2 * D);\n"),
 ("#pragma unroll\n            for (int d = 0; d < 5; d++) {\n                if (d > D) break;\n                const int a1 =", "            QT_STAMP(34);\n#pragma unroll\n            for (int d = 0; d < 5; d++) {\n                if (d > D) break;\n                const int a1 ="),
 ("        __syncthreads();\n        if (s_misc[5] > 0 && s_misc[6] == s_misc[5]) {", "        QT_STAMP(35);\n        __syncthreads();\n        QT_STAMP(36);\n        if (s_misc[5] > 0 && s_misc[6] == s_misc[5]) {"),
 ("        if (K > 0) {\n            const int nc = g.nIni << (2 * K);\n", "        QT_STAMP(37);\n        if (K > 0) {\n            const int nc = g.nIni << (2 * K);\n"),
 ("            cnt = L.cntA; dep = L.depA; cnt2 = L.cntB; dep2 = L.depB;        // (H_(D-1) lives in cntB)\n", "            QT_STAMP(38);\n            cnt = L.cntA; dep = L.depA; cnt2 = L.cntB; dep2 = L.depB;        // (H_(D-1) lives in cntB)\n"),
 ("            __syncthreads();\n            keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {\n                knode = F[qt_jump_xform(", "            __syncthreads();\n            QT_STAMP(39);\n            keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {\n                knode = F[qt_jump_xform("),
 ("            jumpPrev = s_misc[5 + 2 * (K - 1)]; jumpExp = s_misc[6 + 2 * K];\n", "            QT_STAMP(40);\n            jumpPrev = s_misc[5 + 2 * (K - 1)]; jumpExp = s_misc[6 + 2 * K];\n"),
]
for a, b in jsteps: rep(a, b)
open(os.path.join(C, "_qttrace.hip"), "w").write(s)
os.makedirs(os.path.join(R, "ab"), exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", "_qttrace.hip", "-o", "/tmp/qttrace.o"], cwd=C, stderr=subprocess.DEVNULL)
    objs = [o for o in ("orbhip_api.o orbhip_kernels_match.o orbhip_kernels_stereo.o orbhip_kernels_proj.o orbhip_kernels_geom.o orbhip_bow.o orbhip_pool.o").split()]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "/tmp/qttrace.o", "-o", "../../ab/liborbhip_qttrace.so", "-Wl,-rpath,/opt/rocm/lib"], cwd=C)
finally:
    os.remove(os.path.join(C, "_qttrace.hip"))
print("ab/liborbhip_qttrace.so")
