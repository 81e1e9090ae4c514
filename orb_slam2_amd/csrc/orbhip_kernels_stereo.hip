// orbhip_kernels_stereo.hip — Frame::ComputeStereoMatches (src/Frame.cc:466-640 of the reference) on the device-resident
// keypoints, descriptors and pyramids of a left and a right extractor context (first "next" row).
//
//   k_stereo_rows    right keypoints bucketed by image row band [floor(y-r), ceil(y+r)], r = 2*scale      (Frame.cc:475-493)
//   k_stereo_match   one wavefront per left keypoint: candidates of row int(vL), octave +-1, uR in [uL-maxD, uL], best
//                    Hamming distance < TH_HIGH (lowest index wins ties), then the 11x11 L1 correlation over 11 shifts on
//                    the keypoint's pyramid level of both images, parabola sub-pixel fit, disparity -> depth (:496-620)
//   k_stereo_prune   median of the accepted correlation distances, entries >= 1.5*1.4*median invalidated        (:624-639)
// All sums are integer (exact); the few float operations are individually rounded like the reference's float code.
#include "orbhip_internal.h"

#define IMAX 0x7fffffff
#define ST_TH_HIGH 100                 // ORBmatcher::TH_HIGH, ORBmatcher.cc:37
#define ST_TH_ORB 75                   // (TH_HIGH + TH_LOW) / 2, Frame.cc:471

__device__ __forceinline__ int st_wave_sum(int v)       // DPP row shifts / broadcasts, total in lane 63 -> broadcast
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int st_wave_min(int v)       // same network with min; lanes without a source keep their value
{
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x114, 0xf, 0xe, false));
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x118, 0xf, 0xc, false));
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ const uint8_t* st_level(const StereoSide& S, const LevelGeom* geom, int slot, int level, int& pitch)
{
    if (level == 0) { pitch = S.img0_pitch; return S.img0 + (long long)slot * S.img0_frame_stride; }
    pitch = geom[level].pitch;
    return S.pyr + (long long)slot * S.plane_frame_bytes + geom[level].plane_off;
}

// ------------------------------------------------------------------------------------------------ row table
// 1024 threads per frame: a key point per thread for the usual 1000 - 2000 (its row range computed once and kept in registers between the counting
// and the filling pass; a frame with more than SR_T * SR_K key points recomputes the rest), the prefix sum over the H + 1 rows one row per thread.
// (256 threads re-reading the key points for the second pass: 17 us of a stereo pair's 33 us of matching.)
#define SR_T 1024
#define SR_K 4
__global__ __launch_bounds__(SR_T) void k_stereo_rows(StereoParams T)
{
    HIP_DYNAMIC_SHARED(int, lds)                       // [H + 1] row counters / cursors
    __shared__ int s_scan[SR_T / 64];
    const int slot = blockIdx.x, tid = threadIdx.x, H = T.im_h;
    const int nr = T.R.n[slot];
    const orbhip_keypoint* kp = T.R.kp + (long long)slot * T.cap;
    int* rstart = T.row_start + (long long)slot * (H + 1);
    int* ritems = T.row_items + (long long)slot * T.row_cap;
    int lo[SR_K], hi[SR_K];
#pragma unroll
    for (int k = 0; k < SR_K; k++) {
        const int i = tid + k * SR_T;
        lo[k] = 1; hi[k] = 0;
        if (i < nr) {
            const float r = __fmul_rn(2.0f, T.geom[kp[i].octave].scale);              // 2.0f*mvScaleFactors[octave]
            hi[k] = min((int)ceilf(__fadd_rn(kp[i].y, r)), H - 1); lo[k] = max((int)floorf(__fsub_rn(kp[i].y, r)), 0);
        }
    }
    for (int i = tid; i <= H; i += SR_T) lds[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SR_K; k++) for (int y = lo[k]; y <= hi[k]; y++) atomicAdd(&lds[y], 1);
    for (int i = tid + SR_K * SR_T; i < nr; i += SR_T) {
        const float r = __fmul_rn(2.0f, T.geom[kp[i].octave].scale);
        const int maxr = min((int)ceilf(__fadd_rn(kp[i].y, r)), H - 1), minr = max((int)floorf(__fsub_rn(kp[i].y, r)), 0);
        for (int y = minr; y <= maxr; y++) atomicAdd(&lds[y], 1);
    }
    __syncthreads();
    const int per = (H + SR_T) / SR_T;                  // exclusive scan of H+1 counters, `per` consecutive entries per thread
    int sum = 0;
    for (int k = 0; k < per; k++) { const int i = tid * per + k; if (i <= H) sum += lds[i]; }
    int incl = sum;                                     // shuffle scan per wave, the wave totals through LDS
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (tid >> 6); w++) run += s_scan[w];
    for (int k = 0; k < per; k++) { const int i = tid * per + k; if (i <= H) { const int v = lds[i]; lds[i] = run; rstart[i] = run; run += v; } }
    __syncthreads();
    // a row's items in key point order within a pass is not required: k_stereo_match takes the best distance, the lowest index among equals
#pragma unroll
    for (int k = 0; k < SR_K; k++) for (int y = lo[k]; y <= hi[k]; y++) { const int p = atomicAdd(&lds[y], 1); if (p < T.row_cap) ritems[p] = tid + k * SR_T; }
    for (int i = tid + SR_K * SR_T; i < nr; i += SR_T) {
        const float r = __fmul_rn(2.0f, T.geom[kp[i].octave].scale);
        const int maxr = min((int)ceilf(__fadd_rn(kp[i].y, r)), H - 1), minr = max((int)floorf(__fsub_rn(kp[i].y, r)), 0);
        for (int y = minr; y <= maxr; y++) { const int p = atomicAdd(&lds[y], 1); if (p < T.row_cap) ritems[p] = i; }
    }
}

// ------------------------------------------------------------------------------------------------ match + sub-pixel
__global__ __launch_bounds__(256) void k_stereo_match(StereoParams T)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = blockIdx.y;
    const int iL = blockIdx.x * 4 + wave;
    const int nl = T.L.n[slot];
    if (iL >= nl) return;
    const orbhip_keypoint* kpl = T.L.kp + (long long)slot * T.cap;
    const orbhip_keypoint* kpr = T.R.kp + (long long)slot * T.cap;
    float* uout = T.u_right + (long long)slot * T.cap;
    float* dout = T.depth + (long long)slot * T.cap;
    int* sout = T.sad + (long long)slot * T.cap;
    const orbhip_keypoint kL = kpl[iL];
    float res_u = -1.0f, res_d = -1.0f; int res_sad = -1;
    const int levelL = kL.octave; const float vL = kL.y, uL = kL.x;
    const int row = min(max((int)vL, 0), T.im_h - 1);
    const int* rstart = T.row_start + (long long)slot * (T.im_h + 1);
    const int* ritems = T.row_items + (long long)slot * T.row_cap;
    const int c0 = rstart[row], c1 = min(rstart[row + 1], T.row_cap);
    const float minU = __fsub_rn(uL, T.maxD), maxU = __fsub_rn(uL, 0.0f);
    int best = IMAX;
    if (c1 > c0 && !(maxU < 0)) {
        const unsigned long long* dl = (const unsigned long long*)(T.L.desc + ((long long)slot * T.cap + iL) * 32);
        const unsigned long long q0 = dl[0], q1 = dl[1], q2 = dl[2], q3 = dl[3];
        for (int cb = c0; cb < c1; cb += 64) {
            const int t = cb + lane;
            int key = IMAX;
            if (t < c1) {
                const int iR = ritems[t];
                const orbhip_keypoint kR = kpr[iR];
                if (!(kR.octave < levelL - 1 || kR.octave > levelL + 1) && kR.x >= minU && kR.x <= maxU) {
                    const unsigned long long* dr = (const unsigned long long*)(T.R.desc + ((long long)slot * T.cap + iR) * 32);
                    const int dist = __popcll(q0 ^ dr[0]) + __popcll(q1 ^ dr[1]) + __popcll(q2 ^ dr[2]) + __popcll(q3 ^ dr[3]);
                    if (dist < ST_TH_HIGH) key = (dist << 16) | iR;      // strict '<' scan in ascending iR order == lexicographic minimum of (dist, iR)
                }
            }
            best = min(best, st_wave_min(key));
        }
    }
    if (best != IMAX && (best >> 16) < ST_TH_ORB) {
        const int iR = best & 0xffff;
        const float uR0 = kpr[iR].x;
        const float isf = T.geom[levelL].inv_scale;
        const int suL = (int)roundf(__fmul_rn(uL, isf)), svL = (int)roundf(__fmul_rn(vL, isf)), suR0 = (int)roundf(__fmul_rn(uR0, isf));
        const int w = 5, Lr = 5;
        const int lw = T.geom[levelL].w, lh = T.geom[levelL].h;
        // Frame.cc:571-574 only guards iniu / endu; the patch windows themselves are in range for every keypoint the extractor
        // can produce (>= 19 px from the borders at its own level); the clamps below only protect against foreign inputs.
        const bool in_range = !(suR0 < 0 || suR0 + Lr + w + 1 >= lw) && suL - w >= 0 && suL + w < lw && svL - w >= 0 && svL + w < lh && suR0 - Lr - w >= 0;
        if (in_range) {
            int pl, pr; const uint8_t* imL = st_level(T.L, T.geom, slot, levelL, pl); const uint8_t* imR = st_level(T.R, T.geom, slot, levelL, pr);
            // 121 patch pixels: lane handles p = lane and p = lane + 64
            const int pa = lane, pb = lane + 64;
            const int ya = pa / 11, xa = pa - ya * 11, yb = pb / 11, xb = pb - yb * 11;
            const bool hb = pb < 121;
            const int la = imL[(long long)(svL - w + ya) * pl + suL - w + xa];
            const int lb = hb ? imL[(long long)(svL - w + yb) * pl + suL - w + xb] : 0;
            const int cL = __builtin_amdgcn_readlane(la, 60);                         // IL.at<float>(w, w): p = 5*11+5
            const uint8_t* ra = imR + (long long)(svL - w + ya) * pr + suR0 - w + xa;
            const uint8_t* rb = imR + (long long)(svL - w + yb) * pr + suR0 - w + xb;
            int dist[11];
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int inc = k - Lr;
                const int va = ra[inc], vb = hb ? rb[inc] : 0;
                const int cR = __builtin_amdgcn_readlane(va, 60);
                int d = abs((la - cL) - (va - cR));
                if (hb) d += abs((lb - cL) - (vb - cR));
                dist[k] = st_wave_sum(d);                                              // cv::norm(IL, IR, NORM_L1), exact in integers
            }
            int bestS = IMAX, bestinc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++) if (dist[k] < bestS) { bestS = dist[k]; bestinc = k - Lr; }
            if (!(bestinc == -Lr || bestinc == Lr)) {
                float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int k = 1; k < 10; k++) if (k - Lr == bestinc) { d1 = (float)dist[k - 1]; d2 = (float)dist[k]; d3 = (float)dist[k + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(T.geom[levelL].scale, __fadd_rn(__fadd_rn((float)suR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= 0.0f && disparity < T.maxD) {
                        if (disparity <= 0) { disparity = 0.01f; bestuR = __double2float_rn(__dsub_rn((double)uL, 0.01)); }
                        res_d = __fdiv_rn(T.mbf, disparity); res_u = bestuR; res_sad = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) { uout[iL] = res_u; dout[iL] = res_d; sout[iL] = res_sad; }
}

// ------------------------------------------------------------------------------------------------ median prune
// One workgroup per frame.  Every thread keeps its share of the correlation distances in registers (ST_PR per thread; key points beyond
// 256 * ST_PR are re-read from memory): the 16 steps of the bisection then count registers and meet in LDS, instead of walking the array in
// HBM sixteen times (36 us of a single frame's 60 us stereo step were this kernel's dependent round trips).
#define ST_PR 16
__global__ __launch_bounds__(256) void k_stereo_prune(StereoParams T)
{
    __shared__ int s_part[2][4];
    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = T.L.n[slot];
    float* uout = T.u_right + (long long)slot * T.cap;
    float* dout = T.depth + (long long)slot * T.cap;
    const int* sad = T.sad + (long long)slot * T.cap;
    int v[ST_PR];
#pragma unroll
    for (int k = 0; k < ST_PR; k++) { const int i = tid + 256 * k; v[k] = i < nl ? sad[i] : -1; }
    auto count_le = [&](int mid) -> int {                   // #{0 <= sad <= mid} over the whole frame, in every thread
        int c = 0;
#pragma unroll
        for (int k = 0; k < ST_PR; k++) c += (v[k] >= 0 && v[k] <= mid);
        for (int i = tid + 256 * ST_PR; i < nl; i += 256) { const int s = sad[i]; c += (s >= 0 && s <= mid); }
        return c;
    };
    int par = 0;
    auto block_sum = [&](int c) -> int {                     // two LDS rows used in turn: one barrier per sum
        c = st_wave_sum(c);
        if (lane == 0) s_part[par][wave] = c;
        __syncthreads();
        const int t = s_part[par][0] + s_part[par][1] + s_part[par][2] + s_part[par][3];
        par ^= 1;
        return t;
    };
    const int n = block_sum(count_le(IMAX));
    if (n == 0) return;                                   // reference: vDistIdx[0] of an empty vector (undefined) — nothing to prune
    const int k = n / 2;                                  // sorted vDistIdx[size/2].first: the k-th smallest value (0-based)
    int lo = 0, hi = 121 * 510;                           // smallest v with #{sad <= v} >= k + 1
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (block_sum(count_le(mid)) >= k + 1) hi = mid; else lo = mid + 1;
    }
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)lo);
#pragma unroll
    for (int q = 0; q < ST_PR; q++) { const int i = tid + 256 * q; if (v[q] >= 0 && !((float)v[q] < thDist)) { uout[i] = -1.0f; dout[i] = -1.0f; } }
    for (int i = tid + 256 * ST_PR; i < nl; i += 256) { const int s = sad[i]; if (s >= 0 && !((float)s < thDist)) { uout[i] = -1.0f; dout[i] = -1.0f; } }
}

void orbhip_launch_stereo_rows(const StereoParams& T, int nslots, hipStream_t s)
{
    hipLaunchKernelGGL(k_stereo_rows, dim3(nslots, 1, 1), dim3(SR_T, 1, 1), sizeof(int) * (T.im_h + 1), s, T);
}
void orbhip_launch_stereo(const StereoParams& T, int nslots, int max_left, hipStream_t s, bool rows_ready)
{
    if (!rows_ready) orbhip_launch_stereo_rows(T, nslots, s);
    hipLaunchKernelGGL(k_stereo_match, dim3((max_left + 3) / 4, nslots, 1), dim3(256, 1, 1), 0, s, T);
    hipLaunchKernelGGL(k_stereo_prune, dim3(nslots, 1, 1), dim3(256, 1, 1), 0, s, T);
}
