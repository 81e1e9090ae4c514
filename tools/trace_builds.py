#!/usr/bin/env python3
"""Instrumented builds of the kernels for the round's latency experiments: s_memrealtime stamps inserted into a COPY of orbhip_kernels_extract.hip (the product source
carries none), linked with the in-tree objects into ab/liborbhip_<name>.so.  Run after `make -C orb_slam2_amd/csrc`; use with ORBHIP_LIBRARY=$PWD/ab/liborbhip_<name>.so.

  qttrace  phases of k_quadtree                                   -> tools/qt_trace_experiment.py
  pctrace  phases of k_pyramid_cascade (eight workgroups)         -> tools/pc_trace_experiment.py
  fctrace  phases of k_fast_cells (eight workgroups)              -> tools/fc_trace_experiment.py
  span     start / end of every workgroup of a call's kernels     -> tools/span_experiment.py
  pjtrace  rounds / rescans / time of k_proj_select               -> tools/pj_select_experiment.py
usage: tools/trace_builds.py <name> [<name> ...]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); C = os.path.join(R, "orb_slam2_amd", "csrc")
SRC = open(os.path.join(C, "orbhip_kernels_extract.hip")).read()
SRC_PROJ = open(os.path.join(C, "orbhip_kernels_proj.hip")).read()
s = ""
def rep(old, new):
    global s
    assert s.count(old) >= 1, old[:70]
    s = s.replace(old, new, 1)

def qttrace():
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

def pctrace():
    rep("struct PcLevel {", "__device__ unsigned long long g_pc_trace[8 * 32];\n#define PC_STAMP(i) do { if (tid == 0 && frame == P.frame0 && (tile % 11) == 0 && tile / 11 < 8) g_pc_trace[(tile / 11) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)\nstruct PcLevel {")
    rep("    if (tid < 64) {         // ---- round 1 (one wave)", "    PC_STAMP(0);\n    if (tid < 64) {         // ---- round 1 (one wave)")
    rep("    {   // ---- round 2: all requests, then all LDS stores", "    PC_STAMP(1);\n    {   // ---- round 2: all requests, then all LDS stores")
    rep("#pragma unroll\n        for (int k = 0; k < PC_GIT; k++) if (tid + PC_T * k < 3 * gtot) s_grp[tid + PC_T * k] = gv[k];", "        PC_STAMP(2);\n#pragma unroll\n        for (int k = 0; k < PC_GIT; k++) if (tid + PC_T * k < 3 * gtot) s_grp[tid + PC_T * k] = gv[k];")
    rep("    __syncthreads();\n    for (int l = 1; l < L; l++) {\n        const PcLevel R = s_lv[l], S = s_lv[l - 1];", "    __syncthreads();\n    PC_STAMP(3);\n    for (int l = 1; l < L; l++) {\n        const PcLevel R = s_lv[l], S = s_lv[l - 1];")
    rep("        lds_barrier();\n    }\n}\n", "        lds_barrier();\n        PC_STAMP(3 + l);\n    }\n}\nextern \"C\" void orbhip_debug_pc_trace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_trace), sizeof(unsigned long long) * 8 * 32); }\n")

def fctrace():
    rep("template <int CPS, int CSS>\n__global__ __launch_bounds__(256, 8) void k_fast_cells(ExtractParams P)",
        "__device__ unsigned long long g_fc_trace[8 * 16];\n#define FC_STAMP(i) do { if (threadIdx.x == 0 && frame == P.frame0 && (tile % 20) == 0 && tile / 20 < 8) g_fc_trace[(tile / 20) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n"
        "extern \"C\" void orbhip_debug_fc_trace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fc_trace), sizeof(unsigned long long) * 8 * 16); }\n"
        "template <int CPS, int CSS>\n__global__ __launch_bounds__(256, 8) void k_fast_cells(ExtractParams P)")
    rep("    frame += P.frame0;\n    const int cell_last", "    frame += P.frame0;\n    FC_STAMP(0);\n    const int cell_last")
    rep("    if (lane == 0) sh[4 + wave] = cw;\n", "    if (lane == 0) sh[4 + wave] = cw;\n    FC_STAMP(1);\n")
    rep("        lds_dma_wait();\n    }\n    __builtin_amdgcn_wave_barrier();", "        lds_dma_wait();\n    }\n    __builtin_amdgcn_wave_barrier();\n    FC_STAMP(2);")
    rep("        if (lane == 0) sh[wave] = nq;\n        __syncthreads();", "        if (lane == 0) sh[wave] = nq;\n        if (phase == 0) FC_STAMP(3);\n        __syncthreads();\n        if (phase == 0) FC_STAMP(4);")
    rep("        __syncthreads();\n        if (busy)\n        for (int qb = 0; qb < nq; qb += 64) {", "        if (phase == 0) FC_STAMP(5);\n        __syncthreads();\n        if (phase == 0) FC_STAMP(6);\n        if (busy)\n        for (int qb = 0; qb < nq; qb += 64) {")
    rep("        if (count > 0) busy = false;", "        if (phase == 0) FC_STAMP(7);\n        if (count > 0) busy = false;")
    rep("    if (have && lane == 0) P.cell_count[(long long)frame * P.ncells_total + cell_id] = work ? min(count, cd.cand_cap) : 0;\n}", "    if (have && lane == 0) P.cell_count[(long long)frame * P.ncells_total + cell_id] = work ? min(count, cd.cand_cap) : 0;\n    FC_STAMP(8);\n}")

def span():
    rep("struct PcLevel {", "__device__ unsigned long long g_wg[4][2][256];\n#define WG_START(k, i) do { if (threadIdx.x == 0 && (i) < 256) g_wg[k][0][i] = __builtin_amdgcn_s_memrealtime(); } while (0)\n#define WG_END(k, i) do { if (threadIdx.x == 0 && (i) < 256) g_wg[k][1][i] = __builtin_amdgcn_s_memrealtime(); } while (0)\n"
        "extern \"C\" void orbhip_debug_wg(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg), sizeof(unsigned long long) * 4 * 2 * 256); }\nstruct PcLevel {")
    rep("    if (tid < 64) {         // ---- round 1 (one wave)", "    WG_START(0, tile);\n    if (tid < 64) {         // ---- round 1 (one wave)")
    rep("        lds_barrier();\n    }\n}\n", "        lds_barrier();\n    }\n    WG_END(0, tile);\n}\n")
    rep("    frame += P.frame0;\n    const int cell_last = P.fc_cell0 + P.fc_ncells - 1;", "    frame += P.frame0;\n    WG_START(1, tile);\n    const int cell_last = P.fc_cell0 + P.fc_ncells - 1;")
    rep("    if (have && lane == 0) P.cell_count[(long long)frame * P.ncells_total + cell_id] = work ? min(count, cd.cand_cap) : 0;\n}", "    if (have && lane == 0) P.cell_count[(long long)frame * P.ncells_total + cell_id] = work ? min(count, cd.cand_cap) : 0;\n    WG_END(1, tile);\n}")
    rep("    const int nqt = P.nlevels * P.nframes, id = (int)blockIdx.x;\n    if (id < nqt) {", "    const int nqt = P.nlevels * P.nframes, id = (int)blockIdx.x;\n    WG_START(2, id);\n    if (id < nqt) {")
    rep("        blur_mfma_tile(P, b / P.nframes, P.frame0 + b % P.nframes, s_in, s_out, s_band);\n    }\n}", "        blur_mfma_tile(P, b / P.nframes, P.frame0 + b % P.nframes, s_in, s_out, s_band);\n    }\n    WG_END(2, id);\n}")
    rep("    float* s_pat = reinterpret_cast<float*>(&s_win[0][0][0]);\n", "    float* s_pat = reinterpret_cast<float*>(&s_win[0][0][0]);\n    WG_START(3, (int)blockIdx.x);\n")

def pjtrace():
    rep("__device__ __forceinline__ void proj_select_body(const ProjParams& J)\n{", "__device__ unsigned long long g_pj[8];\nextern \"C\" void orbhip_debug_pj(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pj), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pj), z, 64); }\n"
        "__device__ __forceinline__ void proj_select_body(const ProjParams& J)\n{\n    const unsigned long long T0 = __builtin_amdgcn_s_memrealtime(); unsigned long long T1 = 0; int c_steps = 0, c_iter = 0;")
    rep("    fetch(0);\n    for (int qb = 0; qb < J.nq; qb += PJ_T) {", "    T1 = __builtin_amdgcn_s_memrealtime();\n    fetch(0);\n    for (int qb = 0; qb < J.nq; qb += PJ_T) {\n        c_steps++;")
    rep("        for (;;) {\n            int* misc = s_misc + 2 * par;", "        for (;;) {\n            c_iter++;\n            int* misc = s_misc + 2 * par;")
    rep("                const int nc = J.ncand[qs];", "                if (lane == 0) atomicAdd(&g_pj[3], 1ull);\n                const int nc = J.ncand[qs];")
    rep("    if (lane == 0 && wmatches) atomicAdd(&s_misc[PJM_NM], wmatches);", "    if (tid == 0) { const unsigned long long T2 = __builtin_amdgcn_s_memrealtime(); g_pj[0] += 1; g_pj[1] += c_steps; g_pj[2] += c_iter; g_pj[4] += T1 - T0; g_pj[5] += T2 - T1; g_pj[6] += J.nq; }\n    if (lane == 0 && wmatches) atomicAdd(&s_misc[PJM_NM], wmatches);")

flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
allobjs = "orbhip_api.o orbhip_kernels_extract.o orbhip_kernels_match.o orbhip_kernels_stereo.o orbhip_kernels_proj.o orbhip_kernels_geom.o orbhip_bow.o orbhip_pool.o".split()
for name in sys.argv[1:] or ["qttrace"]:
    proj = name == "pjtrace"
    s = SRC_PROJ if proj else SRC
    {"qttrace": qttrace, "pctrace": pctrace, "fctrace": fctrace, "span": span, "pjtrace": pjtrace}[name]()
    tmp = os.path.join(C, "_trace.hip")
    open(tmp, "w").write(s)
    os.makedirs(os.path.join(R, "ab"), exist_ok=True)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", "_trace.hip", "-o", "/tmp/_trace.o"], cwd=C, stderr=subprocess.DEVNULL)
        objs = [o for o in allobjs if o != ("orbhip_kernels_proj.o" if proj else "orbhip_kernels_extract.o")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "/tmp/_trace.o", "-o", "../../ab/liborbhip_%s.so" % name, "-Wl,-rpath,/opt/rocm/lib"], cwd=C)
    finally:
        os.remove(tmp)
    print("ab/liborbhip_%s.so" % name)
