// orbhip_kernels_proj.hip — search core of the projection-guided matchers:
//   mode 0  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)          ORBmatcher.cc:45-129   (TrackLocalMap)
//   mode 1  ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono)  ORBmatcher.cc:1328-1470 (TrackWithMotionModel)
// The caller keeps everything that needs Map / MapPoint / pose types (frustum test, projection, radius, Observations()) and
// passes one flat query per map point (orbhip_proj_query, include/orbhip.h).  On the GPU:
//   k_match_grid (all levels)  Frame::AssignFeaturesToGrid                                  Frame.cc:230-245
//   k_proj_candidates          one wavefront per query: Frame::GetFeaturesInArea in reference order with the level
//                              arguments of the call, the stereo right-coordinate gate and DescriptorDistance
//                              + the best / second-best candidate of the query under the INITIAL map-point state
//   k_proj_select              one workgroup replays the order-dependent loop, 256 queries per step: features claimed by a map point with
//                              observations are skipped by later queries, so a query's precomputed best / second-best is final
//                              unless an earlier query claimed one of the two — those (rare) queries are rescanned one at a time;
//                              same-level ratio rule (mode 0) or best only + rotation histogram (mode 1);
//                              ComputeThreeMaxima                                             ORBmatcher.cc:1601-1642
#include "orbhip_internal.h"
#include <type_traits>

#define IMAX 0x7fffffff
#define PJ_T 256
#define PJ_NONE 0xFFFFFFFFu             // "no candidate" in the per-query records
#define PJ_K 4                          // best candidates recorded per query (under the initial map-point state)
#define PJ_REC (PJ_K + 1)               // + one word: more selectable candidates exist

// wave64 minimum with DPP row shifts / broadcasts (6 dependent VALU steps), broadcast from lane 63
__device__ __forceinline__ int pj_wave_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x114, 0xf, 0xe, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x118, 0xf, 0xc, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// ------------------------------------------------------------------------------------------------ the members' per-point algebra
// What the five projection-guided members of ORBmatcher compute per map point before their window search, as flat float code: every operation is the
// reference's own, in its order, rounded once (the __f*_rn / __d*_rn forms are never contracted).  cv::Mat semantics restated:
//   R*x + t     P.gemm_mode 0: per row ((0 + a0*b0) + a1*b1) + a2*b2 in double (products of floats are exact there), rounded to float, then + t in float
//               (cv::gemm's generic kernel GEMMSingleMul<float,double> followed by a float add; include/cvlite's operator* and operator+);
//               P.gemm_mode 1: OpenCV's small-matrix path of cv::gemm (len 3, flags 0): t0 = a0*b0 + a1*b1 + a2*b2 in float, d = (float)(t0*alpha + c*beta)
//               in double with alpha = beta = 1 - what `Rcw*p3Dw+tcw` compiles to as ONE MatExpr (MatOp_GEMM absorbs the addition);
//               P.gemm_mode 2: the caller's own result is in pts[i].cam_*.
//   a - b       element-wise float subtraction
//   cv::norm(a) sqrt(((0 + a0*a0) + a1*a1) + a2*a2) in double, narrowed by the member's `const float dist = ...`
//   a.dot(b)    ((0 + a0*b0) + a1*b1) + a2*b2 in double
__device__ __forceinline__ float pj_gemm_row(const float* R, int row, float tr, float x, float y, float z, int mode)
{
    const float a0 = R[3 * row], a1 = R[3 * row + 1], a2 = R[3 * row + 2];
    if (mode == 1) {
        const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(a0, x), __fmul_rn(a1, y)), __fmul_rn(a2, z));
        return __double2float_rn(__dadd_rn((double)t0, (double)tr));
    }
    double s = __dadd_rn(0.0, __dmul_rn((double)a0, (double)x));
    s = __dadd_rn(s, __dmul_rn((double)a1, (double)y));
    s = __dadd_rn(s, __dmul_rn((double)a2, (double)z));
    return __fadd_rn(__double2float_rn(s), tr);
}
__device__ __forceinline__ double pj_dot3(float a0, float a1, float a2, float b0, float b1, float b2)
{
    double s = __dadd_rn(0.0, __dmul_rn((double)a0, (double)b0));
    s = __dadd_rn(s, __dmul_rn((double)a1, (double)b1));
    return __dadd_rn(s, __dmul_rn((double)a2, (double)b2));
}
// MapPoint::PredictScale (MapPoint.cc:385-421): ratio = mfMaxDistance/currentDist in float; nScale = ceil(log(ratio)/mfLogScaleFactor) clamped to
// [0, nlevels-1] == the number of host-derived thresholds the ratio reaches (orbhip_predict_scale_table).  A NaN or infinite ratio converts to INT_MIN
// on the reference's x86 (cvttss2si / cvttsd2si), i.e. to level 0.
__device__ __forceinline__ int pj_predict_scale(const orbhip_projection& P, float max_dist, float dist)
{
    const float ratio = __fdiv_rn(max_dist, dist);
    if (!(fabsf(ratio) < __int_as_float(0x7f800000))) return 0;
    int level = 0;
    for (int i = 0; i + 1 < P.nlevels && i < ORBHIP_MAX_PROJ_LEVELS; i++) level += ratio >= P.level_ratio[i] ? 1 : 0;
    return level;
}
struct PjOut { float u, v, radius, ur; int level, min_level, max_level; };
__device__ __forceinline__ bool pj_project(const orbhip_projection& P, const orbhip_map_point& m, PjOut& o)
{
    const int kind = P.kind, gm = P.gemm_mode;
    float X, Y, Z;                                                     // the point in the frame the member projects from
    if (gm == 2) { X = m.cam_x; Y = m.cam_y; Z = m.cam_z; }
    else {
        X = pj_gemm_row(P.R, 0, P.t[0], m.x, m.y, m.z, gm); Y = pj_gemm_row(P.R, 1, P.t[1], m.x, m.y, m.z, gm); Z = pj_gemm_row(P.R, 2, P.t[2], m.x, m.y, m.z, gm);
        if (kind == ORBHIP_PROJ_SIM3) {                                // p3Dc2 = sR21*p3Dc1 + t21 (ORBmatcher.cc:1166-1168 / :1246-1248)
            const float x1 = X, y1 = Y, z1 = Z;
            X = pj_gemm_row(P.R2, 0, P.t2[0], x1, y1, z1, gm); Y = pj_gemm_row(P.R2, 1, P.t2[1], x1, y1, z1, gm); Z = pj_gemm_row(P.R2, 2, P.t2[2], x1, y1, z1, gm);
        }
    }
    float invz, u, v;
    if (kind == ORBHIP_PROJ_LAST_FRAME || kind == ORBHIP_PROJ_FRAME_KF) {
        invz = __double2float_rn(__ddiv_rn(1.0, (double)Z));         // const float invzc = 1.0/x3Dc.at<float>(2)  (:1362, :1498)
        if (kind == ORBHIP_PROJ_LAST_FRAME && invz < 0.0f) return false;                             // :1364-1365 (the relocalisation overload has no such test)
        u = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, X), invz), P.cx);     // CurrentFrame.fx*xc*invzc+CurrentFrame.cx  (:1367-1368, :1500-1501)
        v = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, Y), invz), P.cy);
        if (u < P.min_x || u > P.max_x) return false;                 // :1370-1373, :1503-1506
        if (v < P.min_y || v > P.max_y) return false;
    } else {
        if (Z < 0.0f) return false;                                   // "Depth must be positive" (:323, :857, :1011, :1171, :1251)
        invz = (kind == ORBHIP_PROJ_KF_SIM3 || kind == ORBHIP_PROJ_FUSE) ? __fdiv_rn(1.0f, Z)          // 1/z  (:327, :860)
                                                                          : __double2float_rn(__ddiv_rn(1.0, (double)Z));      // 1.0/z  (:1015, :1174, :1254)
        const float x = __fmul_rn(X, invz), y = __fmul_rn(Y, invz);
        u = __fadd_rn(__fmul_rn(P.fx, x), P.cx);                      // fx*x+cx
        v = __fadd_rn(__fmul_rn(P.fy, y), P.cy);
        if (!(u >= P.min_x && u < P.max_x && v >= P.min_y && v < P.max_y)) return false;                // KeyFrame::IsInImage (KeyFrame.cc:610-613)
    }
    o.u = u; o.v = v; o.ur = 0.0f;
    if (kind == ORBHIP_PROJ_LAST_FRAME) {
        const int oct = m.level;                                      // LastFrame.mvKeys[i].octave (:1375)
        o.level = oct;
        o.radius = __fmul_rn(P.th, P.scale_factors[min(max(oct, 0), ORBHIP_MAX_PROJ_LEVELS - 1)]);     // :1378
        if (P.forward) { o.min_level = oct; o.max_level = -1; }       // :1382-1387
        else if (P.backward) { o.min_level = 0; o.max_level = oct; }
        else { o.min_level = oct - 1; o.max_level = oct + 1; }
        o.ur = __fsub_rn(u, __fmul_rn(P.bf, invz));                   // u - CurrentFrame.mbf*invzc (:1410)
        return true;
    }
    float dist;
    if (kind == ORBHIP_PROJ_SIM3) dist = __double2float_rn(__dsqrt_rn(pj_dot3(X, Y, Z, X, Y, Z)));          // cv::norm(p3Dc2)  (:1184, :1264)
    else {
        const float p0 = __fsub_rn(m.x, P.Ow[0]), p1 = __fsub_rn(m.y, P.Ow[1]), p2 = __fsub_rn(m.z, P.Ow[2]);      // PO = p3Dw-Ow
        dist = __double2float_rn(__dsqrt_rn(pj_dot3(p0, p1, p2, p0, p1, p2)));
        if (dist < m.min_dist || dist > m.max_dist) return false;     // :343-344, :874-875, :1030-1031, :1516-1517
        if (kind != ORBHIP_PROJ_FRAME_KF && pj_dot3(p0, p1, p2, m.nx, m.ny, m.nz) < __dmul_rn(0.5, (double)dist)) return false;    // PO.dot(Pn)<0.5*dist (:349, :881, :1036)
    }
    if (kind == ORBHIP_PROJ_SIM3 && (dist < m.min_dist || dist > m.max_dist)) return false;              // :1187-1188, :1267-1268
    const int level = m.level >= 0 ? m.level : pj_predict_scale(P, m.scale_dist, dist);
    o.level = level;
    o.radius = __fmul_rn(P.th, P.scale_factors[min(max(level, 0), ORBHIP_MAX_PROJ_LEVELS - 1)]);
    o.min_level = level - 1; o.max_level = kind == ORBHIP_PROJ_FRAME_KF ? level + 1 : level;          // :1524 / :364-366
    if (kind == ORBHIP_PROJ_FUSE) o.ur = __fsub_rn(u, __fmul_rn(P.bf, invz));                       // const float ur = u-bf*invz (:868)
    return true;
}

__device__ __forceinline__ void proj_candidates_body(const ProjParams& J, float gwInv, float ghInv)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int iq = blockIdx.x * 4 + wave;
    if (iq >= J.nq) return;
    orbhip_proj_query q;
    bool live = true;
    if (J.pts) {                                                       // the member's projection of map point iq (every lane the same arithmetic)
        const orbhip_map_point m = J.pts[iq];
        PjOut o;
        live = pj_project(*J.proj, m, o);
        q.x = live ? o.u : 0.0f; q.y = live ? o.v : 0.0f; q.radius = live ? o.radius : -1.0f; q.ur = live ? o.ur : 0.0f;
        q.min_level = live ? o.min_level : 0; q.max_level = live ? o.max_level : 0; q.blocks = m.blocks; q.angle = m.angle;
        if (lane == 0) J.q_out[iq] = q;
    } else q = J.q[iq];
    unsigned* cand = J.cand + (long long)iq * J.cand_stride;
    const unsigned long long* d1 = (const unsigned long long*)(J.qdesc + (long long)iq * 32);
    const unsigned long long q0 = d1[0], q1 = d1[1], q2 = d1[2], q3 = d1[3];
    const float x = q.x, y = q.y, r = q.radius;
    const bool check_levels = (q.min_level > 0) || (q.max_level >= 0);                   // Frame.cc:350
    int nc = 0;
    int tk[PJ_K]; unsigned te[PJ_K]; int nsel = 0;                             // this lane's PJ_K smallest (distance, list position) keys + their records
#pragma unroll
    for (int k = 0; k < PJ_K; k++) { tk[k] = IMAX; te[k] = PJ_NONE; }
    const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, J.min_x), r), gwInv)));
    const int maxCX = min(ORBHIP_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, J.min_x), r), gwInv)));
    const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, J.min_y), r), ghInv)));
    const int maxCY = min(ORBHIP_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, J.min_y), r), ghInv)));
    if (live && minCX < ORBHIP_GRID_COLS && maxCX >= 0 && minCY < ORBHIP_GRID_ROWS && maxCY >= 0 && maxCX >= minCX && maxCY >= minCY) {
        const int ny = maxCY - minCY + 1, ncell = (maxCX - minCX + 1) * ny;
        for (int cb = 0; cb < ncell; cb += 64) {
            const int c = cb + lane;
            int a = 0, b = 0;
            if (c < ncell) { const int ix = minCX + c / ny, iy = minCY + c % ny; const int cell = ix * ORBHIP_GRID_ROWS + iy; a = J.grid_start[cell]; b = J.grid_start[cell + 1]; }
            auto passes = [&](int t) -> bool {                                            // the filters of GetFeaturesInArea + the stereo gate
                const float2 k = J.grid_xy[t];
                if (!(fabsf(__fsub_rn(k.x, x)) < r && fabsf(__fsub_rn(k.y, y)) < r)) return false;
                const int i2 = J.grid_items[t];
                if (check_levels) { const int oct = J.kp[i2].octave; if (oct < q.min_level) return false; if (q.max_level >= 0 && oct > q.max_level) return false; }
                if (J.u_right) { const float ur = J.u_right[i2]; if (ur > 0 && fabsf(__fsub_rn(q.ur, ur)) > r) return false; }   // ORBmatcher.cc:91-96, 1408-1414
                return true;
            };
            int cnt = 0;
            for (int t = a; t < b; t++) cnt += passes(t) ? 1 : 0;
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
            int pos = nc + incl - cnt;
            for (int t = a; t < b; t++) {
                if (!passes(t)) continue;
                const int i2 = J.grid_items[t];
                if (pos < J.cand_stride) {
                    const unsigned long long* d2 = (const unsigned long long*)(J.desc + (long long)i2 * 32);
                    const unsigned dist = (unsigned)(__popcll(q0 ^ d2[0]) + __popcll(q1 ^ d2[1]) + __popcll(q2 ^ d2[2]) + __popcll(q3 ^ d2[3]));
                    const unsigned e = (unsigned)i2 | (dist << 19) | ((unsigned)(J.kp[i2].octave & 15) << 28);
                    cand[pos] = e;
                    if (dist < 256u && !(J.blocked_in && J.blocked_in[i2])) {             // selectable under the initial state (ORBmatcher.cc:87-89, 1399-1401)
                        int key = (int)((dist << 19) | (unsigned)pos); unsigned rec = e;
                        nsel++;
#pragma unroll
                        for (int k = 0; k < PJ_K; k++) if (key < tk[k]) { const int tkk = tk[k]; const unsigned tee = te[k]; tk[k] = key; te[k] = rec; key = tkk; rec = tee; }   // sorted insert
                    }
                }
                pos++;
            }
            nc += __shfl(incl, 63);
        }
    }
    if (lane == 0) J.ncand[iq] = min(nc, J.cand_stride);
    // The sequential update "dist < best -> shift, else dist < second" (ORBmatcher.cc:102-114) ends with the two smallest
    // (distance, list position) keys among the candidates that are selectable at that moment.  Keep the PJ_K smallest under the
    // initial state: the select kernel takes the first two that are still unclaimed.
    unsigned out = PJ_NONE; int popped = 0;
#pragma unroll
    for (int r = 0; r < PJ_K; r++) {
        const int m = pj_wave_min(tk[0]);
        if (m == IMAX) break;
        const int owner = __ffsll((long long)__ballot(tk[0] == m)) - 1;
        const unsigned rec = (unsigned)__builtin_amdgcn_readlane((int)te[0], owner);
        if (lane == r) out = rec;
        if (lane == owner) {
            popped++;
#pragma unroll
            for (int k = 0; k + 1 < PJ_K; k++) { tk[k] = tk[k + 1]; te[k] = te[k + 1]; }
            tk[PJ_K - 1] = IMAX; te[PJ_K - 1] = PJ_NONE;
        }
    }
    const bool more = __ballot(nsel > popped) != 0ull;         // selectable candidates beyond the PJ_K recorded ones
    if (lane < PJ_K) J.top[PJ_REC * iq + lane] = out;
    if (lane == PJ_K) J.top[PJ_REC * iq + PJ_K] = more ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_proj_candidates(ProjParams J, float gwInv, float ghInv) { proj_candidates_body(J, gwInv, ghInv); }
// several frames (camera slots) per launch: blockIdx.y = slot, its parameters come from a table in device memory
__global__ __launch_bounds__(256) void k_proj_candidates_batch(const ProjParams* Js, float gwInv, float ghInv) { const ProjParams J = Js[blockIdx.y]; proj_candidates_body(J, gwInv, ghInv); }


// The order-dependent loop, 256 queries per step on all four waves.  A query's decision (its first recorded candidates nobody has claimed) is final once
// no EARLIER undecided query can still claim something it looked at: every undecided query marks what it would claim (atomicMin of its index onto the
// feature's stamp), then checks the stamps of what it looked at; the queries before the first one that is unsure (or whose records are used up) commit,
// the rest go round again.  (One wave did this 64 queries at a time before.  A round is ~450 instructions, 1.0-1.4 us, either way; four times the
// queries per step did NOT mean a quarter of the rounds - neighbouring map points look at the same features, and a search's length is the depth of
// those chains: 33.5 -> 31.1 us per search in the front-end loop, 4-5 % off the matcher calls.)
#define PJM_LOW 0                       // misc words (two sets, alternating by round): lowest undecided query of the round
#define PJM_BAD 1                       //   first query that cannot be decided this round: 2 q + (records used up ? 0 : 1)
#define PJM_NEV 4                       // (one set) rotation-histogram entries written, matches
#define PJM_NM 5
// BIG: the four per-feature tables live in device memory (J.big_ws) instead of LDS - frames of more features than 150 KB of LDS hold (from ~9600 on; the
// reference takes any nFeatures, Tracking.cc:113-125).  The same statements on volatile global words: every access goes to the L2 (the workgroup's waves
// exchange the tables between barriers, and within a wave between wave barriers), atomics are the L2's own; a round then costs global-memory round trips
// instead of LDS ones - the price of a size the LDS form refuses.  The histogram and the round's counters stay in LDS.
template <bool BIG> __device__ __forceinline__ void proj_select_body(const ProjParams& J)
{
    typedef typename std::conditional<BIG, volatile int, int>::type TI;
    typedef typename std::conditional<BIG, volatile float, float>::type TF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = J.n;
    HIP_DYNAMIC_SHARED(int, lds)
    TI* s_blocked = BIG ? J.big_ws : lds;                   // F.mvpMapPoints[i] && Observations() > 0
    TI* s_fq = s_blocked + n;                               // query whose map point sits in F.mvpMapPoints[i]
    TI* s_stamp = s_fq + n;                                 // lowest query index that (speculatively) claims feature i with a blocking map point
    TF* s_fang = reinterpret_cast<TF*>(s_stamp + n);        // mvKeysUn[i].angle
    int* s_hist = BIG ? lds : lds + 4 * n;                  // [HISTO_LENGTH] + 3 maxima + 13 misc
    int* s_misc = s_hist + ORBHIP_HISTO_LENGTH + 3;
    auto amin = [](TI* p, int v) { atomicMin(const_cast<int*>(p), v); };
    auto amax = [](TI* p, int v) { atomicMax(const_cast<int*>(p), v); };
    for (int i = tid; i < n; i += PJ_T) { s_blocked[i] = J.blocked_in ? (int)J.blocked_in[i] : 0; s_fq[i] = -1; s_stamp[i] = IMAX; s_fang[i] = J.kp[i].angle; }
    for (int i = tid; i < ORBHIP_HISTO_LENGTH + 16; i += PJ_T) s_hist[i] = 0;
    __syncthreads();
    if (tid < 4) s_misc[tid] = IMAX;                         // PJM_LOW / PJM_BAD of both sets
    __syncthreads();
    const float factor = 1.0f / ORBHIP_HISTO_LENGTH;
    const bool use_second = J.mode == 0;                     // mode 1 keeps the best candidate only (ORBmatcher.cc:1396-1425)
    auto rot_bin = [&](float qang, int bidx) -> int {                                  // ORBmatcher.cc:1430-1440
        float rot = __fsub_rn(qang, s_fang[bidx]);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, factor));
        if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
        return min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
    };
    // the records of step s + 1 (7 words per query from global memory) are requested before step s is resolved and consumed after it
    unsigned nt[PJ_K], nmore = 0u; int nblocks = 0; float nqang = 0.0f;
    auto fetch = [&](int qb) {
        const int q = qb + tid;
        const bool inb = q < J.nq;
#pragma unroll
        for (int k = 0; k < PJ_K; k++) nt[k] = inb ? J.top[PJ_REC * q + k] : PJ_NONE;
        nmore = inb ? J.top[PJ_REC * q + PJ_K] : 0u;
        nblocks = inb ? J.q[q].blocks : 0;
        nqang = inb ? J.q[q].angle : 0.0f;
    };
    int wmatches = 0, par = 0;                              // this wave's matches (wave-uniform); which misc set the round uses
    fetch(0);
    for (int qb = 0; qb < J.nq; qb += PJ_T) {
        const int q = qb + tid;
        const bool inb = q < J.nq;
        int ei[PJ_K], ed[PJ_K], el[PJ_K]; int nk = 0;
#pragma unroll
        for (int k = 0; k < PJ_K; k++) {
            const unsigned t = nt[k];
            ei[k] = min((int)(t & 0x7FFFFu), n - 1); ed[k] = (int)((t >> 19) & 0x1FFu); el[k] = (int)(t >> 28);      // (an empty record's index is out of range: any valid entry, unused)
            if (t != PJ_NONE) nk = k + 1;
        }
        const bool more = inb && nmore != 0u;
        const int blocks = nblocks;
        const float qang = nqang;
        if (qb + PJ_T < J.nq) fetch(qb + PJ_T);
        bool pend = nk > 0;                                  // this query is still undecided
        int stamped = -1;                                    // feature this query's speculative claim currently marks
        for (;;) {
            int* misc = s_misc + 2 * par; int* other = s_misc + 2 * (par ^ 1);
            // first two recorded candidates nobody has claimed so far (claims only ever add up: a claimed feature stays claimed); the state of all
            // recorded candidates in ONE round of LDS reads (tested one by one behind each other's outcome they were up to twelve dependent latencies)
            int blk[PJ_K];
#pragma unroll
            for (int k = 0; k < PJ_K; k++) blk[k] = s_blocked[ei[k]];
            int a = -1, b = -1;
#pragma unroll
            for (int k = 0; k < PJ_K; k++)
                if (pend && k < nk && b < 0 && !blk[k]) { if (a < 0) a = k; else b = k; }
            if (!use_second) b = -1;
            const int ia = a >= 0 ? ei[a] : 0, da = a >= 0 ? ed[a] : 256, la = a >= 0 ? el[a] : -1;
            const int db = b >= 0 ? ed[b] : 256, lb = b >= 0 ? el[b] : -1;
            // records used up while more selectable candidates exist: the answer is not in the records
            const bool exhausted = pend && more && (a < 0 || (use_second && b < 0));
            // the decision the reference takes if nothing this query looks at is claimed by a query between the committed ones and it
            const bool accept = pend && !exhausted && a >= 0 && da <= J.th_high && !(J.mode == 0 && la == lb && (float)da > __fmul_rn(J.nnratio, (float)db));   // ORBmatcher.cc:116-126
            const int want = (accept && blocks) ? ia : -1;
            if (stamped >= 0 && stamped != want && s_stamp[stamped] == q) s_stamp[stamped] = IMAX;    // withdraw an outdated speculative claim (its own: the stamp is its index)
            { const int lo = pj_wave_min(pend ? q : IMAX); if (lane == 0 && lo != IMAX) atomicMin(&misc[PJM_LOW], lo); }
            __syncthreads();
            if (want >= 0) amin(&s_stamp[want], q);
            stamped = want;
            if (tid == 0) { other[PJM_LOW] = IMAX; other[PJM_BAD] = IMAX; }                           // the next round's set: nobody reads or writes it between these two barriers
            __syncthreads();
            // an earlier, not yet committed query of this step may still claim one of the candidates this decision rests on
            const int lowest = misc[PJM_LOW];
            int stp[PJ_K];
#pragma unroll
            for (int k = 0; k < PJ_K; k++) stp[k] = s_stamp[ei[k]];
            bool unsure = false;
            const int last = use_second ? (b >= 0 ? b : nk - 1) : a;
#pragma unroll
            for (int k = 0; k < PJ_K; k++)
                if (pend && q != lowest && k <= last && !blk[k] && stp[k] < q) unsure = true;          // (nothing is blocked between the reads above and here: commits follow)
            { const int bk = pj_wave_min((unsure || exhausted) ? 2 * q + (exhausted ? 0 : 1) : IMAX); if (lane == 0 && bk != IMAX) atomicMin(&misc[PJM_BAD], bk); }
            __syncthreads();
            const int badkey = misc[PJM_BAD];
            const int first_bad = badkey == IMAX ? IMAX : (badkey >> 1);
            const bool bad_exhausted = badkey != IMAX && !(badkey & 1);
            const bool commit = pend && q < first_bad;
            const bool win = commit && accept;
            const unsigned long long wins = __ballot(win);
            int evbase = 0;
            if (J.mode == 1 && J.check_ori && wins) { if (lane == 0) evbase = atomicAdd(&s_misc[PJM_NEV], (int)__popcll(wins)); evbase = __builtin_amdgcn_readlane(evbase, 0); }
            if (win) {
                amax(&s_fq[ia], q);                                                // later queries overwrite earlier ones (ORBmatcher.cc:123)
                if (blocks) s_blocked[ia] = 1;
                if (J.mode == 1 && J.check_ori) {
                    const int bin = rot_bin(qang, ia);
                    atomicAdd(&s_hist[bin], 1);
                    J.events[evbase + __popcll(wins & ((1ull << lane) - 1ull))] = (bin << 20) | ia;     // rotHist[bin].push_back(bestIdx2): the entries do not interact, any order
                }
            }
            wmatches += (int)__popcll(wins);
            pend = pend && !commit;
            par ^= 1;
            __syncthreads();
            if (first_bad == IMAX) break;
            // the first undecided query: if it merely waited for earlier ones it is re-evaluated now that they are final;
            // only a query whose records are used up is rescanned against the current state - by its own wave, the others wait
            if (first_bad != lowest || !bad_exhausted) continue;
            if (wave == ((first_bad - qb) >> 6)) {
                const int fl = (first_bad - qb) & 63, qs = first_bad;
                if (lane == fl) { pend = false; if (stamped >= 0 && s_stamp[stamped] == q) s_stamp[stamped] = IMAX; stamped = -1; }
                __builtin_amdgcn_wave_barrier();
                const int nc = J.ncand[qs];
                const unsigned* cand = J.cand + (long long)qs * J.cand_stride;
                int best = 256, blevel = -1, second = 256, slevel = -1, bidx = -1;
                for (int cb = 0; cb < nc; cb += 64) {
                    const int t = cb + lane;
                    const unsigned e = t < nc ? cand[t] : 0u;
                    const int i2 = min((int)(e & 0x7FFFFu), n - 1), dist = (int)((e >> 19) & 0x1FFu), lvl = (int)(e >> 28);
                    const bool valid = t < nc && dist < 256 && !s_blocked[i2];
                    // two smallest (distance, lane) keys by DPP min networks; the key carries the level so no extra readlane is needed
                    const int key = valid ? ((dist << 10) | (lane << 4) | lvl) : IMAX;
                    const int k1 = pj_wave_min(key);
                    if (k1 == IMAX) continue;
                    const int f1 = (k1 >> 4) & 63, wmin = k1 >> 10, cl = k1 & 15, ci = __builtin_amdgcn_readlane(i2, f1);      // first candidate with the minimum
                    const int k2 = pj_wave_min(lane == f1 ? IMAX : key);
                    const int wsec = k2 == IMAX ? 256 : (k2 >> 10), l2 = k2 == IMAX ? -1 : (k2 & 15);
                    // the two smallest (distance, list position) keys == the reference's sequential best / second-best update
                    if (wmin < best) {
                        if (best <= wsec) { second = best; slevel = blevel; } else { second = wsec; slevel = l2; }
                        best = wmin; bidx = ci; blevel = cl;
                    } else if (wmin < second) { second = wmin; slevel = cl; }
                }
                if (best <= J.th_high && !(J.mode == 0 && blevel == slevel && (float)best > __fmul_rn(J.nnratio, (float)second))) {
                    const int sblocks = __builtin_amdgcn_readlane(blocks, fl);
                    const float sang = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qang), fl));
                    if (lane == 0) {
                        s_fq[bidx] = qs;
                        if (sblocks) { s_blocked[bidx] = 1; amin(&s_stamp[bidx], qs); }
                        if (J.mode == 1 && J.check_ori) { const int bin = rot_bin(sang, bidx); atomicAdd(&s_hist[bin], 1); J.events[atomicAdd(&s_misc[PJM_NEV], 1)] = (bin << 20) | bidx; }
                    }
                    wmatches++;
                }
            }
            __syncthreads();                                 // the rescanned query's claim precedes the next round's reads
        }
    }
    if (lane == 0 && wmatches) atomicAdd(&s_misc[PJM_NM], wmatches);
    __syncthreads();
    if (wave == 0) {
        int nmatches = s_misc[PJM_NM];
        const int nev = s_misc[PJM_NEV];
        if (J.mode == 1 && J.check_ori) {
            if (lane == 0) {
                int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
                for (int i = 0; i < ORBHIP_HISTO_LENGTH; i++) {
                    const int s = s_hist[i];
                    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                    else if (s > max3) { max3 = s; ind3 = i; }
                }
                if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
                else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
                s_hist[ORBHIP_HISTO_LENGTH] = ind1; s_hist[ORBHIP_HISTO_LENGTH + 1] = ind2; s_hist[ORBHIP_HISTO_LENGTH + 2] = ind3;
            }
            __builtin_amdgcn_wave_barrier();
            const int ind1 = s_hist[ORBHIP_HISTO_LENGTH], ind2 = s_hist[ORBHIP_HISTO_LENGTH + 1], ind3 = s_hist[ORBHIP_HISTO_LENGTH + 2];
            // ORBmatcher.cc:1452-1466: one decrement per histogram entry of a bin outside the three maxima.  The entries do not interact (a feature
            // NULLed through one entry stays NULL whatever its other entries say), so the wave takes 64 of them per step.
            int removed = 0;
            for (int kb = 0; kb < nev; kb += 64) {
                const int k = kb + lane;
                bool out = false; int idx = 0;
                if (k < nev) { const int ev = J.events[k], bin = ev >> 20; idx = ev & 0xFFFFF; out = bin != ind1 && bin != ind2 && bin != ind3; }
                if (out) s_fq[idx] = -2;                                                             // mvpMapPoints[idx] = NULL (:1460): reported as -2, not as "untouched"
                removed += __popcll(__ballot(out));
            }
            __builtin_amdgcn_wave_barrier();
            nmatches -= removed;
        }
        if (lane == 0) *J.nmatches = nmatches;
    }
    __syncthreads();
    for (int i = tid; i < n; i += PJ_T) { J.feature_query[i] = s_fq[i]; if (J.blocked_out) J.blocked_out[i] = (unsigned char)s_blocked[i]; }
}
__global__ __launch_bounds__(PJ_T) void k_proj_select(ProjParams J) { proj_select_body<false>(J); }
__global__ __launch_bounds__(PJ_T) void k_proj_select_big(ProjParams J) { proj_select_body<true>(J); }
// one workgroup per camera slot (big_ws is set for every slot or for none: the launcher decides on the largest slot)
__global__ __launch_bounds__(PJ_T) void k_proj_select_batch(const ProjParams* Js) { const ProjParams J = Js[blockIdx.x]; proj_select_body<false>(J); }
__global__ __launch_bounds__(PJ_T) void k_proj_select_batch_big(const ProjParams* Js) { const ProjParams J = Js[blockIdx.x]; proj_select_body<true>(J); }

#define PJ_LDS_BUDGET (150 * 1024)
size_t orbhip_proj_select_lds(int n) { return sizeof(int) * ((size_t)4 * n + ORBHIP_HISTO_LENGTH + 16); }
bool orbhip_proj_select_big(int n)
{
    const char* env = getenv("ORBHIP_SELECT_BIG");                       // tests: 1 = the device-memory form at any size (read per call: tests switch inside one process)
    const bool force = env && env[0] == '1';
    return force || orbhip_proj_select_lds(n) > PJ_LDS_BUDGET;
}

void orbhip_launch_proj(const ProjParams& J, hipStream_t s)
{
    const float gwInv = (float)ORBHIP_GRID_COLS / (float)(J.max_x - J.min_x), ghInv = (float)ORBHIP_GRID_ROWS / (float)(J.max_y - J.min_y);
    if (J.nq > 0) hipLaunchKernelGGL(k_proj_candidates, dim3((J.nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, J, gwInv, ghInv);
    if (J.big_ws) hipLaunchKernelGGL(k_proj_select_big, dim3(1, 1, 1), dim3(PJ_T, 1, 1), orbhip_proj_select_lds(0), s, J);
    else hipLaunchKernelGGL(k_proj_select, dim3(1, 1, 1), dim3(PJ_T, 1, 1), orbhip_proj_select_lds(J.n), s, J);
}


void orbhip_launch_proj_batch(const ProjParams* d_slots, int nslots, int max_nq, int max_n, float gwInv, float ghInv, hipStream_t s)
{
    if (nslots <= 0) return;
    if (max_nq > 0) hipLaunchKernelGGL(k_proj_candidates_batch, dim3((max_nq + 3) / 4, nslots, 1), dim3(256, 1, 1), 0, s, d_slots, gwInv, ghInv);
    if (orbhip_proj_select_big(max_n)) hipLaunchKernelGGL(k_proj_select_batch_big, dim3(nslots, 1, 1), dim3(PJ_T, 1, 1), orbhip_proj_select_lds(0), s, d_slots);
    else hipLaunchKernelGGL(k_proj_select_batch, dim3(nslots, 1, 1), dim3(PJ_T, 1, 1), orbhip_proj_select_lds(max_n), s, d_slots);
}

// ------------------------------------------------------------------------------------------------ best candidate in a window
// The candidate loop of ORBmatcher::Fuse (both overloads, ORBmatcher.cc:884-948 / 1038-1079) and of the two passes of
// ORBmatcher::SearchBySim3 (:1188-1224 / 1268-1304): KeyFrame::GetFeaturesInArea(x, y, radius) (KeyFrame.cc:569-608), key points
// of level L-1 .. L, Fuse's reprojection chi-square gate (stereo 7.8, mono 5.99; f32 product compared in f64 like the reference's
// literals), FIRST smallest descriptor distance.  Queries do not interact: one wavefront per query scans the ordered bucket table
// (cell ix*ROWS+iy, then index == GetFeaturesInArea's order; a key point within the radius always lies in a visited cell, see
// k_match_candidates), lane-parallel, first-minimum by (distance, table position).
__device__ __forceinline__ void best_in_window_body(const BestParams& B, int iq, int lane)
{
    if (iq >= B.nq) return;
    if (B.skip && ((B.skip[iq] >> B.skip_bit) & 1ull)) { if (lane == 0) { B.best_idx[iq] = -1; B.best_dist[iq] = 256; } return; }     // the point is in this key frame already (ORBmatcher.cc:848-849)
    orbhip_best_query q;
    if (B.pts) {
        PjOut o;
        const bool live = pj_project(*B.proj, B.pts[iq], o);
        q.x = live ? o.u : 0.0f; q.y = live ? o.v : 0.0f; q.radius = live ? o.radius : -1.0f; q.ur = live ? o.ur : 0.0f; q.level = live ? o.level : 0;
        if (lane == 0 && B.q_out) B.q_out[iq] = q;
        if (!live) { if (lane == 0) { B.best_idx[iq] = -1; B.best_dist[iq] = 256; } return; }
    } else q = B.q[iq];
    const uint4* q4 = reinterpret_cast<const uint4*>(B.qdesc + (long long)iq * 32);
    const uint4 qa = q4[0], qb = q4[1];
    // only the run of the table that holds the grid columns the radius can reach (+-1, see k_match_candidates): the table is ordered by column
    int t_lo = 0, nitems = 0;
    {
        const float cl = __fmul_rn(__fsub_rn(__fsub_rn(q.x, B.min_x), q.radius), B.gw_inv), ch = __fmul_rn(__fadd_rn(__fsub_rn(q.x, B.min_x), q.radius), B.gw_inv);
        const bool all = !(B.gw_inv > 0.0f);
        const int col_lo = (!all && cl >= 1.0f) ? (int)fminf(floorf(cl) - 1.0f, (float)ORBHIP_GRID_COLS) : 0;
        const int col_hi = (!all && ch < (float)(ORBHIP_GRID_COLS - 2)) ? (int)fmaxf(ceilf(ch) + 2.0f, 0.0f) : ORBHIP_GRID_COLS;
        t_lo = __builtin_amdgcn_readfirstlane(B.grid_start[min(col_lo, col_hi) * ORBHIP_GRID_ROWS]); nitems = __builtin_amdgcn_readfirstlane(B.grid_start[col_hi * ORBHIP_GRID_ROWS]);
    }
    int best = 256, bidx = -1;
    for (int tb = t_lo; tb < nitems; tb += 64) {
        const int t = tb + lane;
        int idx = 0, dist = 256; bool ok = false;
        if (t < nitems) {
            const float2 k = B.grid_xy[t];
            ok = fabsf(__fsub_rn(k.x, q.x)) < q.radius && fabsf(__fsub_rn(k.y, q.y)) < q.radius;      // KeyFrame.cc:598-602
            if (ok) {
                idx = B.grid_items[t];
                const int lvl = B.kp[idx].octave;
                ok = !(lvl < q.level - 1 || lvl > q.level);                                            // :896-899
                if (ok && B.chi2_gate) {
                    const float ex = __fsub_rn(q.x, k.x), ey = __fsub_rn(q.y, k.y);
                    const float ur = B.u_right ? B.u_right[idx] : -1.0f;
                    if (ur >= 0) {                                                                       // :901-914
                        const float er = __fsub_rn(q.ur, ur);
                        const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                        ok = !((double)__fmul_rn(e2, B.inv_level_sigma2[lvl]) > 7.8);
                    } else {                                                                             // :915-926
                        const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        ok = !((double)__fmul_rn(e2, B.inv_level_sigma2[lvl]) > 5.99);
                    }
                }
                if (ok) {
                    const uint4* d4 = reinterpret_cast<const uint4*>(B.desc + (long long)idx * 32);
                    const uint4 da = d4[0], db = d4[1];
                    dist = __popc(qa.x ^ da.x) + __popc(qa.y ^ da.y) + __popc(qa.z ^ da.z) + __popc(qa.w ^ da.w) +
                           __popc(qb.x ^ db.x) + __popc(qb.y ^ db.y) + __popc(qb.z ^ db.z) + __popc(qb.w ^ db.w);
                }
            }
        }
        const unsigned long long V = __ballot(ok && dist < 256);
        if (V == 0) continue;
        unsigned long long mk = V;
#pragma unroll
        for (int b = 8; b >= 0; b--) { const unsigned long long z = __ballot(((dist >> b) & 1) == 0) & mk; if (z) mk = z; }
        const int first = __ffsll((long long)mk) - 1;
        const int wmin = __builtin_amdgcn_readlane(dist, first), ci = __builtin_amdgcn_readlane(idx, first);
        if (wmin < best) { best = wmin; bidx = ci; }                                                    // strict: earlier table position wins ties
    }
    if (lane == 0) { B.best_idx[iq] = bidx; B.best_dist[iq] = best; }
}
__global__ __launch_bounds__(256) void k_best_in_window(BestParams B)
{
    best_in_window_body(B, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}
// several key frames in one launch (orbhip_search_best_in_window_batch): parameter blocks in device memory, pref[s] = first block of slot s
__global__ __launch_bounds__(256) void k_best_in_window_batch(const BestParams* Bs, const int* pref, int nslots)
{
    int sl = 0; while (sl + 1 < nslots && (int)blockIdx.x >= pref[sl + 1]) sl++;
    best_in_window_body(Bs[sl], ((int)blockIdx.x - pref[sl]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}

void orbhip_launch_best_in_window(const BestParams& B, hipStream_t s)
{
    if (B.nq > 0) hipLaunchKernelGGL(k_best_in_window, dim3((B.nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, B);
}
void orbhip_launch_best_in_window_batch(const BestParams* d_slots, const int* d_pref, int nslots, int nblocks, hipStream_t s)
{
    if (nblocks > 0) hipLaunchKernelGGL(k_best_in_window_batch, dim3(nblocks, 1, 1), dim3(256, 1, 1), 0, s, d_slots, d_pref, nslots);
}
