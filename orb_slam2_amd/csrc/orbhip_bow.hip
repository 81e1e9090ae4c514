// orbhip_bow.hip — DBoW2 vocabulary path of ORB_SLAM2 on gfx950:
//   TemplatedVocabulary::loadFromTextFile      Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425   (host, once)
//   TemplatedVocabulary::transform             :1127-1194, :1218-1262 (tree descent with FORB::distance, FORB.cpp:81-101)
//   BowVector::addWeight / normalize            BowVector.cpp:34-94;  FeatureVector::addFeature  FeatureVector.cpp:29-43
//   scoring objects                             ScoringObject.cpp:24-313                                   (host, tiny)
// as called by Frame::ComputeBoW (Frame.cc:395-402: transform(vCurrentDesc, mBowVec, mFeatVec, 4)).
//
// Device layout: the k-ary tree renumbered breadth first, so that the children of a node are neighbours (child order kept: the first minimum wins), one
// 64-byte record per node: the 32-byte descriptor + {first child, number of children, the file's node id, word id} + the f64 weight.  A child's record carries
// the pointer to ITS children and its weight, so a level of the descent is ONE dependent fetch (the k child records), not three (child range -> child ids ->
// descriptors), and nothing is fetched after the leaf.  ORBvoc.txt (k = 10, L = 6) is 1.1 M nodes = 71 MB of records: resident in HBM, hot upper levels in L2.
//  k_bow_descend   one thread per feature walks the tree (<= L dependent steps, k Hamming distances each, strict '<' so the
//                  first minimum wins) -> word id, weight, node id `levelsup` levels above the leaf.
//  k_bow_assemble  two workgroups per frame turn the per-feature triples into the two std::map's of the reference, flattened in key order: one sorts the keys
//                  (word << IB | feature index), the other (node << IB | index), by a bitonic network on registers; run heads become entries, weights of a
//                  word are added in feature order and the L1 norm is summed in word order as ONE chain of f64 additions (their order is part of the
//                  result), then every value is divided by it.  A frame's results are one contiguous block: one copy takes them home.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <algorithm>
#include <array>
#include "../../include/orbhip.h"
#include "orbhip_internal.h"
#include <map>
#include <mutex>

#define BOWCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return orbhip_set_error(ORBHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct BowNode { uint4 da, db; uint32_t first_child, nchildren, id, word; double weight; double pad; };      // 64 bytes
struct BowOut { uint8_t* base; long long stride; int o_id, o_fn, o_ff, o_fo, o_cnt; };        // a frame's results: [bow_val C f64 | bow_id C | fv_node C | fv_feat C | fv_off C+1 | nbow, -, -, -, nfv] at base + frame * stride

struct BowParams {
    const uint8_t* desc; long long desc_frame_stride;        // features of frame f at desc + f*stride, 32 B each
    const int* nfeat; int nfeat_fixed; int cap;                // per-frame count (device) or one fixed count; output stride
    const BowNode* nodes;                                      // breadth-first numbering, root = 0
    int L, levelsup, accumulate, must_normalize, l2;
    uint32_t* word; double* weight; uint32_t* node;            // [frame][cap]
    BowOut out;
    int lcap;                          // features the LDS arrays of k_bow_assemble are carved for (>= every frame's count, <= cap)
};

__device__ __forceinline__ int bow_count(const BowParams& P, int frame) { return min(P.nfeat ? P.nfeat[frame] : P.nfeat_fixed, P.cap); }

// transform(feature, word_id, weight, nid, levelsup)  (TemplatedVocabulary.h:1218-1262)
// Sixteen lanes per feature (a DPP row), one child record per lane: a level is ONE round trip in which every lane fetches 64 contiguous bytes, the sixteen
// distances are reduced to (distance << 8 | child index) by four row exchanges, and the winner's lane hands on where its children are.  (One thread per
// feature fetched its k records itself: 40 fetch instructions per level from the FOUR compute units a 1000-feature frame occupied - 0.020 - 0.030 ms of
// texture-address time; with a wavefront per four features the frame spreads over 250 units.)
#define BD_G 16
__global__ __launch_bounds__(64) void k_bow_descend(BowParams P)
{
    const int frame = blockIdx.y, l16 = threadIdx.x & (BD_G - 1), i = blockIdx.x * (64 / BD_G) + (threadIdx.x / BD_G);
    if (i >= bow_count(P, frame)) return;
    const uint4* f4 = reinterpret_cast<const uint4*>(P.desc + (long long)frame * P.desc_frame_stride + (long long)i * 32);
    const uint4 fa = f4[0], fb = f4[1];
    const int nid_level = P.L - P.levelsup;
    const uint4* rec = reinterpret_cast<const uint4*>(P.nodes);
    const uint4 root = rec[2];                                                           // {first child, count, id, word}
    uint32_t a = root.x, cnt = root.y, nid = 0;
    int level = 0, win = 0;
    uint4 binfo = root; uint2 bw; bw.x = 0u; bw.y = 0u;                                  // this lane's best child: its record's tail, the weight's bits
    while (cnt > 0) {                                                                    // !isLeaf()   (a vocabulary without words never gets here: the callers return before)
        ++level;
        int bkey = 0x7fffffff;
        for (uint32_t c0 = 0; c0 < cnt; c0 += BD_G) {                                    // k <= 20: at most two rounds
            const uint32_t c = c0 + l16;
            if (c < cnt) {
                const size_t dev = (size_t)(a + c) * 4;
                const uint4 ra = rec[dev], rb = rec[dev + 1], ri = rec[dev + 2]; const uint2 rw = *reinterpret_cast<const uint2*>(rec + dev + 3);
                const int d = __popc(fa.x ^ ra.x) + __popc(fa.y ^ ra.y) + __popc(fa.z ^ ra.z) + __popc(fa.w ^ ra.w) +
                              __popc(fb.x ^ rb.x) + __popc(fb.y ^ rb.y) + __popc(fb.z ^ rb.z) + __popc(fb.w ^ rb.w);    // FORB::distance
                const int key = (d << 8) | (int)c;                                       // the smallest distance, the first child among equals (:1243-1247)
                if (key < bkey) { bkey = key; binfo = ri; bw = rw; }
            }
        }
        int m = bkey;
#pragma unroll
        for (int off = BD_G / 2; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off, BD_G));
        win = m & (BD_G - 1);                                                            // child c sits in lane c % 16
        a = __shfl(binfo.x, win, BD_G); cnt = __shfl(binfo.y, win, BD_G);
        if (level == nid_level) nid = __shfl(binfo.z, win, BD_G);
    }
    if (l16 == win) {
        const long long o = (long long)frame * P.cap + i;
        P.word[o] = binfo.w; reinterpret_cast<uint2*>(P.weight)[o] = bw; P.node[o] = nid;
    }
}

#define BA_T 1024                      // one workgroup of 16 wavefronts per frame: a single frame's latency is this kernel's length
// in-place exclusive scan of a[0..n) by the whole workgroup, returns the total (scratch: BA_T/64 ints)
__device__ __forceinline__ int ba_exscan(int* a, int n, int* scratch, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (n + BA_T - 1) / BA_T, lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += a[i];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += scratch[w];
    int total = 0;
    for (int w = 0; w < BA_T / 64; w++) total += scratch[w];
    int run = base + incl - sum;
    for (int i = lo; i < hi; i++) { const int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

// Ascending sort of NARR sets of E * BA_T keys (32 or 64 bits; unique but for the invalid ones, ~0, which sort last) held in REGISTERS, thread t owning the keys at
// positions e * BA_T + t: a bitonic network in which a key meets its partner (position ^ j) and keeps the smaller or the larger of the two.  Partners less
// than 64 apart are another lane's registers (__shfl_xor: 45 of the 55 stages at 1024 keys), partners BA_T or more apart are the thread's own registers, and
// only the stages in between go through LDS (write, barrier, read the partner, barrier).  x0 / x1: LDS for E * BA_T keys each; the sorted keys are left there.
// (History: a rank sort - every key counted against every other one - was 0.46 ms of ONE workgroup for a 2000-feature frame; the network in LDS with a
// workgroup barrier after each stage, one array after the other, 0.064 ms per 1000 features; both arrays in one pass with wavefront-local stages 0.015 of
// the kernel's 0.035 - every stage two LDS round trips.)
template <typename KT, int NARR, int E> __device__ __forceinline__ void ba_sort(KT (&r0)[E], KT (&r1)[E], KT* x0, KT* x1, int tid)
{
    for (int k = 2; k <= E * BA_T; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= BA_T) {                                                             // the partner is one of this thread's own keys
#pragma unroll
                for (int je = E / 2; je >= 1; je >>= 1) {
                    if (j != je * BA_T) continue;
#pragma unroll
                    for (int e = 0; e < E; e++) {
                        if (e & je) continue;
                        const bool up = (((e * BA_T + tid) & k) == 0);
                        { const KT a = r0[e], b = r0[e | je]; if ((a > b) == up) { r0[e] = b; r0[e | je] = a; } }
                        if (NARR > 1) { const KT a = r1[e], b = r1[e | je]; if ((a > b) == up) { r1[e] = b; r1[e | je] = a; } }
                    }
                }
            } else if (j >= 64) {                                                        // another wavefront's
#pragma unroll
                for (int e = 0; e < E; e++) { x0[e * BA_T + tid] = r0[e]; if (NARR > 1) x1[e * BA_T + tid] = r1[e]; }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int i = e * BA_T + tid;
                    const bool low = ((i & j) == 0) == ((i & k) == 0);                    // this position keeps the smaller key
                    { const KT q = x0[i ^ j]; r0[e] = (q < r0[e]) == low ? q : r0[e]; }
                    if (NARR > 1) { const KT q = x1[i ^ j]; r1[e] = (q < r1[e]) == low ? q : r1[e]; }
                }
                __syncthreads();
            } else {                                                                     // another lane's
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int i = e * BA_T + tid;
                    const bool low = ((i & j) == 0) == ((i & k) == 0);
                    { const KT q = __shfl_xor(r0[e], j); r0[e] = (q < r0[e]) == low ? q : r0[e]; }
                    if (NARR > 1) { const KT q = __shfl_xor(r1[e], j); r1[e] = (q < r1[e]) == low ? q : r1[e]; }
                }
            }
        }
#pragma unroll
    for (int e = 0; e < E; e++) { x0[e * BA_T + tid] = r0[e]; if (NARR > 1) x1[e * BA_T + tid] = r1[e]; }
    __syncthreads();
}
__host__ __device__ inline int ba_keys_per_thread(int lcap) { return lcap <= BA_T ? 1 : lcap <= 2 * BA_T ? 2 : lcap <= 4 * BA_T ? 4 : 8; }
// keys of `ks` bytes: [keys E * BA_T][flag, head: lcap ints each][scratch 128 B][the words' f64 values: their own lcap-block behind 4-byte keys, the keys' block
// behind 8-byte ones][128 B that ba_norm may read past the values]
__host__ __device__ inline size_t ba_lds_bytes(int lcap, int ks) { return (size_t)ba_keys_per_thread(lcap) * BA_T * ks + (size_t)lcap * 8 + 128 + (ks == 4 ? (size_t)lcap * 8 : 0) + 128; }

// sum of |v| (or v * v) over vals[0 .. nb) in index order: the order of the f64 additions is part of the result, so it is ONE chain of dependent additions.
// Every lane of the wavefront reads the same value (an LDS broadcast), eight values ahead of the addition that needs them: the chain never waits for memory.
// (Fed by v_readlane from a 64-value register: two readlanes + a hazard nop per addition, 0.0105 ms per 1000 values.)
template <bool L2> __device__ __forceinline__ double ba_norm(const double* vals, int nb)
{
    // (reads run up to 15 values past nb: still inside the kernel's LDS block - the key arrays are followed by flag / head / scratch - and never added)
    double norm = 0.0, cur[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cur[k] = vals[k];
    int base = 0;
    for (; base + 8 <= nb; base += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = vals[base + 8 + k];
#pragma unroll
        for (int k = 0; k < 8; k++) norm += L2 ? cur[k] * cur[k] : fabs(cur[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) if (base + k < nb) norm += L2 ? cur[k] * cur[k] : fabs(cur[k]);
    return norm;
}

// Two workgroups per frame, side by side: blockIdx.y = 0 builds the BowVector (word keys), 1 the FeatureVector (node keys) - the first one's chain of f64
// additions (0.010 ms per 1000 words, nothing to be done about its order) is the kernel's length, the second finishes under it.
// E = keys per thread: the network sorts E * BA_T >= lcap keys.  KT = uint32_t when (id << IB | feature index) fits 32 bits for every node id of the vocabulary
// (IB = log2(E * BA_T); ORBvoc.txt's 1.1 M nodes with up to 2048 features), else uint64_t (id << 32 | index): half the exchanges and compare instructions of the sort.
template <typename KT, int E> __global__ __launch_bounds__(BA_T) void k_bow_assemble(BowParams P)
{
    HIP_DYNAMIC_SHARED(unsigned long long, lds64)
    constexpr int IB = sizeof(KT) == 4 ? (E == 1 ? 10 : E == 2 ? 11 : E == 4 ? 12 : 13) : 32;
    constexpr KT IM = (KT)(((KT)1 << IB) - 1), INVALID = (KT)~(KT)0;
    const int frame = blockIdx.x, tid = threadIdx.x, n = min(bow_count(P, frame), P.lcap), cap = P.cap;
    const bool words = blockIdx.y == 0;
    const int lcap = P.lcap;                               // flag / head / vals hold lcap entries; `cap` is the row stride of the HBM arrays
    constexpr int p2 = E * BA_T;
    KT* sk = reinterpret_cast<KT*>(lds64);                 // [p2] (word or node << IB | feature), after the sort
    int* flag = reinterpret_cast<int*>(sk + p2);           // [lcap]
    int* head = flag + lcap;                               // [lcap]
    int* scratch = head + lcap;                            // [BA_T / 64] + the norm: 128 bytes
    double* vals = sizeof(KT) == 4 ? reinterpret_cast<double*>(scratch + 32) : reinterpret_cast<double*>(sk);   // [lcap] the f64 values of the words
    const uint32_t* id = (words ? P.word : P.node) + (long long)frame * cap; const double* wt = P.weight + (long long)frame * cap;
    uint8_t* ob = P.out.base + (long long)frame * P.out.stride;
    double* bow_val = reinterpret_cast<double*>(ob); uint32_t* bow_id = reinterpret_cast<uint32_t*>(ob + P.out.o_id);
    uint32_t* fv_node = reinterpret_cast<uint32_t*>(ob + P.out.o_fn); uint32_t* fv_feat = reinterpret_cast<uint32_t*>(ob + P.out.o_ff);
    int* fv_off = reinterpret_cast<int*>(ob + P.out.o_fo); int* counts = reinterpret_cast<int*>(ob + P.out.o_cnt);

    // ---- this map's keys for every feature with w > 0 (:1157-1161), sorted
    KT key[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = e * BA_T + tid;
        const bool on = i < n && wt[i] > 0;
        key[e] = on ? (KT)(((KT)id[i] << IB) | (KT)i) : INVALID;
        if (i < n) flag[i] = on;
    }
    __syncthreads();
    const int m = ba_exscan(flag, n, scratch, tid);        // features that are "not stopped"
    ba_sort<KT, 1, E>(key, key, sk, sk, tid);
    for (int i = tid; i < m; i += BA_T) {
        const int h = (i == 0) || ((sk[i] >> IB) != (sk[i - 1] >> IB)); flag[i] = h; head[i] = h;
        if (!words) fv_feat[i] = (uint32_t)(sk[i] & IM);
    }
    __syncthreads();
    const int nh = ba_exscan(flag, m, scratch, tid);       // entries of the map

    if (!words) {      // ---- FeatureVector: fv.addFeature(nid, i_feature)
        for (int i = tid; i < m; i += BA_T) if (head[i]) { fv_node[flag[i]] = (uint32_t)(sk[i] >> IB); fv_off[flag[i]] = i; }
        if (tid == 0) { fv_off[nh] = m; counts[4] = nh; }
        return;
    }

    // ---- BowVector: v.addWeight(id, w) in feature order
    const int nb = nh;
    double hv[E]; int hp[E];                                                                // the entries this thread owns: value, position
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = tid + r * BA_T;
        hp[r] = -1; hv[r] = 0.0;
        if (i < m && head[i]) {
            const uint32_t w = (uint32_t)(sk[i] >> IB);
            double v = wt[(uint32_t)(sk[i] & IM)];
            if (P.accumulate) for (int j = i + 1; j < m && (uint32_t)(sk[j] >> IB) == w; j++) v += wt[(uint32_t)(sk[j] & IM)];   // `vit->second += v`
            hp[r] = flag[i]; hv[r] = v; bow_id[flag[i]] = w;                                 // addIfNotExist keeps the first
        }
    }
    if (P.accumulate && nb > 0 && !P.must_normalize) {                                      // :1164-1170
        const double nd = (double)nb;
#pragma unroll
        for (int r = 0; r < E; r++) if (hp[r] >= 0) hv[r] = hv[r] / nd;
    }
    if (P.must_normalize) {                                                                 // BowVector::normalize
        __syncthreads();                                                                    // (8-byte keys: the values take the keys' LDS, nobody reads a key any more)
#pragma unroll
        for (int r = 0; r < E; r++) if (hp[r] >= 0) vals[hp[r]] = hv[r];
        __syncthreads();
        if (tid < 64) {
            double norm = P.l2 ? ba_norm<true>(vals, nb) : ba_norm<false>(vals, nb);
            if (P.l2) norm = sqrt(norm);
            if (tid == 0) reinterpret_cast<double*>(scratch)[8] = norm;
        }
        __syncthreads();
        const double norm = reinterpret_cast<double*>(scratch)[8];
        if (norm > 0.0) {
#pragma unroll
            for (int r = 0; r < E; r++) if (hp[r] >= 0) hv[r] = hv[r] / norm;
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++) if (hp[r] >= 0) bow_val[hp[r]] = hv[r];
    if (tid == 0) counts[0] = nb;
}

// ------------------------------------------------------------------------------------------------ SearchByBoW
// ORBmatcher::SearchByBoW (ORBmatcher.cc:159-288 key frame vs frame, :522-655 key frame vs key frame) on flat data.  The
// reference merge-joins the two FeatureVectors and, inside a common vocabulary node, walks side 1's features in order;
// a side-2 feature taken by an earlier one is skipped.  A side-2 feature lives in exactly one node, so nodes are independent:
// one wavefront per node of side 1 (binary search for the same node on side 2), sequential over its side-1 features, lanes
// over the side-2 features of the node (chunks of 64, "taken" flags as per-lane bit masks), ballot arg-min for best /
// second-best exactly like the scan order of the reference (first minimum wins, second = smallest of the rest).
struct BowMatchParams {
    int mode; float nnratio; int check_ori;
    const uint8_t* d1; const float* ang1; const uint8_t* valid1; int n1; const uint32_t* fn1; const int* fo1; const uint32_t* ff1; int nf1;
    const uint8_t* d2; const float* ang2; const uint8_t* valid2; int n2; const uint32_t* fn2; const int* fo2; const uint32_t* ff2; int nf2;
    int* match12; int* bin12; int* hist; int* nmatches; int* overflow;
};
#define BM_CHUNKS 128                   // side-2 features of one node handled per wave: 64 * BM_CHUNKS
#define BM_REG 4                        // ... of which a node with up to 64 * BM_REG keeps them in registers

__device__ __forceinline__ unsigned long long bm_argmin_mask(int d, unsigned long long M)
{   // lanes of M holding the minimum of a 9-bit key: MSB-first elimination with ballots
#pragma unroll
    for (int b = 8; b >= 0; b--) { const unsigned long long z = __ballot(((d >> b) & 1) == 0) & M; if (z) M = z; }
    return M;
}

// slot of a batched launch that block `block` belongs to: pref[s] = first block of slot s (ascending, pref[nslots] = grid size); uniform -> scalar loads
__device__ __forceinline__ int bm_batch_slot(const int* pref, int nslots, int block) { int s = 0; while (s + 1 < nslots && block >= pref[s + 1]) s++; return s; }

// wave64 minimum with DPP row shifts / broadcasts (6 dependent VALU steps against the 9 ballots of bm_argmin_mask); result broadcast from lane 63
__device__ __forceinline__ int bm_wave_min(int v)
{
    const int IMAX = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x114, 0xf, 0xe, false));      // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x118, 0xf, 0xc, false));      // row_shr:8 -> lane 15 of each row = row minimum
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave minimum
    return __builtin_amdgcn_readlane(v, 63);
}

// position of `node` among side 2's ascending node ids, -1 if it is not there (whole wavefront, uniform result)
__device__ __forceinline__ int bm_find_node(const uint32_t* fn2, int nf2, uint32_t node, int lane)
{
    if (nf2 <= 256) {                                          // the usual FeatureVector (levelsup = 4 of ORBvoc: ~100 nodes): the lanes look at all ids at once
        uint32_t id[4];
#pragma unroll
        for (int c = 0; c < 4; c++) id[c] = c * 64 + lane < nf2 ? fn2[c * 64 + lane] : 0xffffffffu;
        int lo = -1;
#pragma unroll
        for (int c = 3; c >= 0; c--) { const unsigned long long B = __ballot(c * 64 + lane < nf2 && id[c] == node); if (B) lo = c * 64 + __ffsll((long long)B) - 1; }
        return lo;
    }
    int lo = 0, hi = nf2;                                      // lower_bound
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fn2[mid] < node) lo = mid + 1; else hi = mid; }
    return (lo < nf2 && fn2[lo] == node) ? lo : -1;
}

__device__ __forceinline__ void bow_match_body(const BowMatchParams& P, int a, int lane)
{
    if (a >= P.nf1) return;
    const uint32_t node = P.fn1[a];
    const int lo = bm_find_node(P.fn2, P.nf2, node, lane);
    if (lo < 0) return;
    const int b1 = P.fo1[a], e1 = P.fo1[a + 1], b2 = P.fo2[lo], len2 = P.fo2[lo + 1] - b2;
    if (len2 > 64 * BM_CHUNKS) { if (lane == 0) atomicAdd(P.overflow, 1); return; }
    if (len2 <= 64 * BM_REG) {
        // The usual node: side 2's features fit the wavefront's registers (lane t holds features t, t + 64, ... of the node: index, flag, descriptor, fetched
        // ONCE), side 1's are fetched 64 at a time, and the walk over them (in order: an earlier one takes a side-2 feature away from the later ones) runs on
        // registers, the query taken from its lane by v_readlane.  (The general walk below fetches index -> flag -> descriptor for every side-1 feature in turn
        // and side 2's for each of them again: some three dependent round trips per side-1 feature, 0.027 ms for a pair of 1000-feature frames with ten
        // features a node and most of a 2000-feature pair's 0.2 ms, whose largest node decides.)
        const int nchr = (len2 + 63) >> 6;
        int idx2[BM_REG]; bool ok2[BM_REG]; uint4 da[BM_REG], db[BM_REG];
#pragma unroll
        for (int c = 0; c < BM_REG; c++) {
            idx2[c] = 0; ok2[c] = false; da[c].x = da[c].y = da[c].z = da[c].w = 0u; db[c] = da[c];
            const int t = c * 64 + lane;
            if (t < len2) {
                idx2[c] = (int)P.ff2[b2 + t]; ok2[c] = !(P.mode == 1 && !P.valid2[idx2[c]]);      // :207-208 / :574-579
                const uint4* d4 = reinterpret_cast<const uint4*>(P.d2 + (long long)idx2[c] * 32); da[c] = d4[0]; db[c] = d4[1];
            }
        }
        unsigned taken = 0;                                    // bit c: this lane's entry of chunk c is matched
        for (int blk = b1; blk < e1; blk += 64) {
            const int cnt = min(64, e1 - blk);
            int idx1 = 0, v1 = 0; uint4 pa; pa.x = pa.y = pa.z = pa.w = 0u; uint4 pb = pa;
            if (lane < cnt) { idx1 = (int)P.ff1[blk + lane]; v1 = P.valid1[idx1]; const uint4* q4 = reinterpret_cast<const uint4*>(P.d1 + (long long)idx1 * 32); pa = q4[0]; pb = q4[1]; }
            for (int i = 0; i < cnt; i++) {
                if (!__builtin_amdgcn_readlane(v1, i)) continue;                               // !pMP || pMP->isBad()   (:195-199)
                const unsigned qx = __builtin_amdgcn_readlane((int)pa.x, i), qy = __builtin_amdgcn_readlane((int)pa.y, i), qz = __builtin_amdgcn_readlane((int)pa.z, i), qw = __builtin_amdgcn_readlane((int)pa.w, i);
                const unsigned rx = __builtin_amdgcn_readlane((int)pb.x, i), ry = __builtin_amdgcn_readlane((int)pb.y, i), rz = __builtin_amdgcn_readlane((int)pb.z, i), rw = __builtin_amdgcn_readlane((int)pb.w, i);
                const int i1 = __builtin_amdgcn_readlane(idx1, i);
                int best = 256, second = 256, bidx = -1, bchunk = 0, blane = 0;
#pragma unroll
                for (int c = 0; c < BM_REG; c++) {
                    if (c >= nchr) continue;
                    const bool ok = ok2[c] && !((taken >> c) & 1u);
                    const int dist = ok ? __popc(qx ^ da[c].x) + __popc(qy ^ da[c].y) + __popc(qz ^ da[c].z) + __popc(qw ^ da[c].w) +
                                          __popc(rx ^ db[c].x) + __popc(ry ^ db[c].y) + __popc(rz ^ db[c].z) + __popc(rw ^ db[c].w) : 256;
                    // smallest (distance, lane) key: the first candidate with the minimum wins; the runner-up is the minimum with that lane masked out.
                    // (dist == 256 can never pass `dist < bestDist`, init 256)
                    const int key = (ok && dist < 256) ? ((dist << 6) | lane) : 0x7fffffff;
                    const int k1 = bm_wave_min(key);
                    if (k1 == 0x7fffffff) continue;
                    const int first = k1 & 63, wmin = k1 >> 6, ci = __builtin_amdgcn_readlane(idx2[c], first);
                    const int k2 = bm_wave_min(lane == first ? 0x7fffffff : key);
                    const int wsec = k2 == 0x7fffffff ? 256 : (k2 >> 6);
                    if (wmin < best) { second = min(best, wsec); best = wmin; bidx = ci; bchunk = c; blane = first; } else second = min(second, wmin);
                }
                const bool close = P.mode == 0 ? best <= ORBHIP_TH_LOW : best < ORBHIP_TH_LOW;                 // :221 / :588
                if (close && (float)best < __fmul_rn(P.nnratio, (float)second)) {                               // :223 / :590
                    if (lane == blane) taken |= 1u << bchunk;
                    if (lane == 0) {
                        P.match12[i1] = bidx;
                        if (P.check_ori) {
                            float rot = __fsub_rn(P.ang1[i1], P.ang2[bidx]);
                            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                            int bin = (int)roundf(__fmul_rn(rot, 1.0f / ORBHIP_HISTO_LENGTH));
                            if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
                            bin = min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
                            P.bin12[i1] = bin; atomicAdd(&P.hist[bin], 1);
                        }
                    }
                }
            }
        }
        return;
    }
    const int nch = (len2 + 63) >> 6;
    unsigned long long taken_lo = 0, taken_hi = 0;             // bit c: this lane's entry of chunk c is matched (c < 64 / c >= 64)
    for (int i1 = b1; i1 < e1; i1++) {
        const unsigned idx1 = P.ff1[i1];
        if (!P.valid1[idx1]) continue;                         // !pMP || pMP->isBad()   (:195-199)
        const uint4* q4 = reinterpret_cast<const uint4*>(P.d1 + (long long)idx1 * 32);
        const uint4 qa = q4[0], qb = q4[1];
        int best = 256, second = 256, bidx = -1, bchunk = 0, blane = 0;
        for (int c = 0; c < nch; c++) {
            const int t = c * 64 + lane;
            int idx2 = 0, dist = 256; bool ok = false;
            if (t < len2) {
                idx2 = (int)P.ff2[b2 + t];
                const bool taken = c < 64 ? (taken_lo >> c) & 1ull : (taken_hi >> (c - 64)) & 1ull;
                ok = !taken && !(P.mode == 1 && !P.valid2[idx2]);           // :207-208 / :574-579
                if (ok) {
                    const uint4* d4 = reinterpret_cast<const uint4*>(P.d2 + (long long)idx2 * 32);
                    const uint4 da = d4[0], db = d4[1];
                    dist = __popc(qa.x ^ da.x) + __popc(qa.y ^ da.y) + __popc(qa.z ^ da.z) + __popc(qa.w ^ da.w) +
                           __popc(qb.x ^ db.x) + __popc(qb.y ^ db.y) + __popc(qb.z ^ db.z) + __popc(qb.w ^ db.w);
                }
            }
            const unsigned long long V = __ballot(ok && dist < 256);        // dist == 256 can never pass `dist < bestDist` (init 256)
            if (V == 0) continue;
            const unsigned long long mk = bm_argmin_mask(dist, V);
            const int first = __ffsll((long long)mk) - 1;
            const int wmin = __builtin_amdgcn_readlane(dist, first), ci = __builtin_amdgcn_readlane(idx2, first);
            const unsigned long long V2 = V & ~(1ull << first);
            int wsec = 256;
            if (V2) { const unsigned long long mk2 = bm_argmin_mask(dist, V2); wsec = __builtin_amdgcn_readlane(dist, __ffsll((long long)mk2) - 1); }
            if (wmin < best) { second = min(best, wsec); best = wmin; bidx = ci; bchunk = c; blane = first; } else second = min(second, wmin);
        }
        const bool close = P.mode == 0 ? best <= ORBHIP_TH_LOW : best < ORBHIP_TH_LOW;                 // :221 / :588
        if (close && (float)best < __fmul_rn(P.nnratio, (float)second)) {                               // :223 / :590
            if (lane == blane) { if (bchunk < 64) taken_lo |= 1ull << bchunk; else taken_hi |= 1ull << (bchunk - 64); }
            if (lane == 0) {
                P.match12[idx1] = bidx;
                if (P.check_ori) {
                    float rot = __fsub_rn(P.ang1[idx1], P.ang2[bidx]);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, 1.0f / ORBHIP_HISTO_LENGTH));
                    if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
                    bin = min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
                    P.bin12[idx1] = bin; atomicAdd(&P.hist[bin], 1);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_bow_match(BowMatchParams P)
{
    bow_match_body(P, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}
// several independent pairs in one launch (orbhip_search_by_bow_batch): the parameter blocks live in device memory
__global__ __launch_bounds__(256) void k_bow_match_batch(const BowMatchParams* Ps, const int* pref, int npairs)
{
    const int sl = bm_batch_slot(pref, npairs, blockIdx.x);
    bow_match_body(Ps[sl], (blockIdx.x - pref[sl]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}

// ComputeThreeMaxima (ORBmatcher.cc:1601-1642), rejection of the other bins (:262-285 / :629-652), nmatches
__device__ __forceinline__ void bow_finish_body(const BowMatchParams& P, int tid)
{
    __shared__ int s_ind[3]; __shared__ int s_cnt;
    if (tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < ORBHIP_HISTO_LENGTH; i++) {
            const int s = P.hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3; s_cnt = 0;
    }
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < P.n1; i += 256) {
        if (P.match12[i] < 0) continue;
        const int b = P.bin12[i];
        if (P.check_ori && b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) P.match12[i] = -1; else cnt++;
    }
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) *P.nmatches = s_cnt;
}
__global__ __launch_bounds__(256) void k_bow_match_finish(BowMatchParams P) { bow_finish_body(P, threadIdx.x); }
__global__ __launch_bounds__(256) void k_bow_match_finish_batch(const BowMatchParams* Ps) { bow_finish_body(Ps[blockIdx.x], threadIdx.x); }

// ------------------------------------------------------------------------------------------------ SearchForTriangulation
// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) + CheckDistEpipolarLine (:140-157) on flat data: features of two
// key frames that have no map point yet, same vocabulary node, descriptor distance <= TH_LOW, not too close to the epipole
// (mono-mono pairs), within 3.84 sigma^2 of the epipolar line.  The reference keeps the candidate if `dist <= bestDist`, so among
// the admissible candidates the LAST one with the smallest distance wins; it never marks side-2 features as taken (vbMatched2
// is read, never written), so the side-1 features of a node are independent.  Same wave-per-node layout as k_bow_match.
struct TriParams {
    BowMatchParams M;                      // descriptors, valid1/valid2 = "already has a map point", FeatureVectors, outputs
    const float* kp1; const float* kp2;    // (x, y, angle, octave) of mvKeysUn
    const uint8_t* st1; const uint8_t* st2;    // mvuRight >= 0
    float F[9]; float ex, ey; const float* scale2; const float* sigma2_2; int only_stereo;
};

__device__ __forceinline__ void bow_triangulate_body(const TriParams& T, int a, int lane)
{
    const BowMatchParams& P = T.M;
    if (a >= P.nf1) return;
    const uint32_t node = P.fn1[a];
    const int lo = bm_find_node(P.fn2, P.nf2, node, lane);
    if (lo < 0) return;
    const int b1 = P.fo1[a], e1 = P.fo1[a + 1], b2 = P.fo2[lo], len2 = P.fo2[lo + 1] - b2;
    const int nch = (len2 + 63) >> 6;
    if (len2 <= 64 * BM_REG) {
        // The usual node (see bow_match_body): side 2's features in registers, side 1's fetched 64 at a time, the walk on registers
        int idx2[BM_REG]; bool ok2[BM_REG], stereo2[BM_REG], near_epipole[BM_REG]; float x2[BM_REG], y2[BM_REG]; double chi2[BM_REG]; uint4 da[BM_REG], db[BM_REG];
#pragma unroll
        for (int c = 0; c < BM_REG; c++) {
            idx2[c] = 0; ok2[c] = false; stereo2[c] = false; near_epipole[c] = false; x2[c] = y2[c] = 0.f; chi2[c] = 0.0; da[c].x = da[c].y = da[c].z = da[c].w = 0u; db[c] = da[c];
            const int t = c * 64 + lane;
            if (t < len2) {
                idx2[c] = (int)P.ff2[b2 + t]; stereo2[c] = T.st2[idx2[c]] != 0;
                ok2[c] = !P.valid2[idx2[c]] && !(T.only_stereo && !stereo2[c]);             // :727-736
                x2[c] = T.kp2[4 * idx2[c]]; y2[c] = T.kp2[4 * idx2[c] + 1]; const int oct2 = (int)T.kp2[4 * idx2[c] + 3];
                const uint4* d4 = reinterpret_cast<const uint4*>(P.d2 + (long long)idx2[c] * 32); da[c] = d4[0]; db[c] = d4[1];
                const float dxe = __fsub_rn(T.ex, x2[c]), dye = __fsub_rn(T.ey, y2[c]);
                near_epipole[c] = __fadd_rn(__fmul_rn(dxe, dxe), __fmul_rn(dye, dye)) < __fmul_rn(100.0f, T.scale2[oct2]);     // :747-753
                chi2[c] = 3.84 * (double)T.sigma2_2[oct2];
            }
        }
        for (int blk = b1; blk < e1; blk += 64) {
            const int cnt = min(64, e1 - blk);
            int idx1 = 0, go1 = 0, st1 = 0; float x1 = 0.f, y1 = 0.f, ang1 = 0.f; uint4 pa; pa.x = pa.y = pa.z = pa.w = 0u; uint4 pb = pa;
            if (lane < cnt) {
                idx1 = (int)P.ff1[blk + lane]; st1 = T.st1[idx1] != 0;
                go1 = !P.valid1[idx1] && !(T.only_stereo && !st1);           // "If there is already a MapPoint skip" (:698-700), :708-711
                x1 = T.kp1[4 * idx1]; y1 = T.kp1[4 * idx1 + 1]; ang1 = T.kp1[4 * idx1 + 2];
                const uint4* q4 = reinterpret_cast<const uint4*>(P.d1 + (long long)idx1 * 32); pa = q4[0]; pb = q4[1];
            }
            for (int i = 0; i < cnt; i++) {
                if (!__builtin_amdgcn_readlane(go1, i)) continue;
                const bool stereo1 = __builtin_amdgcn_readlane(st1, i) != 0;
                const float qx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), i)), qy1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), i));
                const float qang = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ang1), i));
                const int i1 = __builtin_amdgcn_readlane(idx1, i);
                const float la = __fadd_rn(__fadd_rn(__fmul_rn(qx1, T.F[0]), __fmul_rn(qy1, T.F[3])), T.F[6]);      // epipolar line of kp1 in image 2 (:143-145)
                const float lb = __fadd_rn(__fadd_rn(__fmul_rn(qx1, T.F[1]), __fmul_rn(qy1, T.F[4])), T.F[7]);
                const float lc = __fadd_rn(__fadd_rn(__fmul_rn(qx1, T.F[2]), __fmul_rn(qy1, T.F[5])), T.F[8]);
                const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
                const unsigned qx = __builtin_amdgcn_readlane((int)pa.x, i), qy = __builtin_amdgcn_readlane((int)pa.y, i), qz = __builtin_amdgcn_readlane((int)pa.z, i), qw = __builtin_amdgcn_readlane((int)pa.w, i);
                const unsigned rx = __builtin_amdgcn_readlane((int)pb.x, i), ry = __builtin_amdgcn_readlane((int)pb.y, i), rz = __builtin_amdgcn_readlane((int)pb.z, i), rw = __builtin_amdgcn_readlane((int)pb.w, i);
                int best = ORBHIP_TH_LOW, bidx = -1;
#pragma unroll
                for (int c = 0; c < BM_REG; c++) {
                    if (c >= nch) continue;
                    const int dist = __popc(qx ^ da[c].x) + __popc(qy ^ da[c].y) + __popc(qz ^ da[c].z) + __popc(qw ^ da[c].w) +
                                     __popc(rx ^ db[c].x) + __popc(ry ^ db[c].y) + __popc(rz ^ db[c].z) + __popc(rw ^ db[c].w);
                    bool ok = ok2[c] && dist <= ORBHIP_TH_LOW;                                   // :742
                    if (!stereo1 && !stereo2[c] && near_epipole[c]) ok = false;                  // :747-753
                    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, x2[c]), __fmul_rn(lb, y2[c])), lc);
                    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                    if (den == 0.0f || !((double)dsqr < chi2[c])) ok = false;                    // :149-156
                    const int k1 = bm_wave_min(ok ? ((dist << 6) | (63 - lane)) : 0x7fffffff);      // `dist <= bestDist` keeps the LAST of equal candidates
                    if (k1 == 0x7fffffff) continue;
                    const int wmin = k1 >> 6, ci = __builtin_amdgcn_readlane(idx2[c], 63 - (k1 & 63));
                    if (wmin <= best) { best = wmin; bidx = ci; }
                }
                if (bidx >= 0 && lane == 0) {
                    P.match12[i1] = bidx;
                    if (P.check_ori) {
                        float rot = __fsub_rn(qang, T.kp2[4 * bidx + 2]);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, 1.0f / ORBHIP_HISTO_LENGTH));
                        if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
                        bin = min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
                        P.bin12[i1] = bin; atomicAdd(&P.hist[bin], 1);
                    }
                }
            }
        }
        return;
    }
    for (int i1 = b1; i1 < e1; i1++) {
        const unsigned idx1 = P.ff1[i1];
        if (P.valid1[idx1]) continue;                                   // "If there is already a MapPoint skip" (:698-700)
        const bool stereo1 = T.st1[idx1] != 0;
        if (T.only_stereo && !stereo1) continue;
        const float x1 = T.kp1[4 * idx1], y1 = T.kp1[4 * idx1 + 1];
        // epipolar line of kp1 in image 2: l = x1' F12 = [a b c]  (:143-145), left-to-right sums like the reference's expression
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(x1, T.F[0]), __fmul_rn(y1, T.F[3])), T.F[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, T.F[1]), __fmul_rn(y1, T.F[4])), T.F[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, T.F[2]), __fmul_rn(y1, T.F[5])), T.F[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        const uint4* q4 = reinterpret_cast<const uint4*>(P.d1 + (long long)idx1 * 32);
        const uint4 qa = q4[0], qb = q4[1];
        int best = ORBHIP_TH_LOW, bidx = -1;
        for (int c = 0; c < nch; c++) {
            const int t = c * 64 + lane;
            int idx2 = 0, dist = 256; bool ok = false;
            if (t < len2) {
                idx2 = (int)P.ff2[b2 + t];
                const bool stereo2 = T.st2[idx2] != 0;
                ok = !P.valid2[idx2] && !(T.only_stereo && !stereo2);   // :727-736
                if (ok) {
                    const uint4* d4 = reinterpret_cast<const uint4*>(P.d2 + (long long)idx2 * 32);
                    const uint4 da = d4[0], db = d4[1];
                    dist = __popc(qa.x ^ da.x) + __popc(qa.y ^ da.y) + __popc(qa.z ^ da.z) + __popc(qa.w ^ da.w) +
                           __popc(qb.x ^ db.x) + __popc(qb.y ^ db.y) + __popc(qb.z ^ db.z) + __popc(qb.w ^ db.w);
                    ok = dist <= ORBHIP_TH_LOW;                        // :742
                }
                if (ok) {
                    const float x2 = T.kp2[4 * idx2], y2 = T.kp2[4 * idx2 + 1]; const int oct2 = (int)T.kp2[4 * idx2 + 3];
                    if (!stereo1 && !stereo2) {                         // :747-753
                        const float dx = __fsub_rn(T.ex, x2), dy = __fsub_rn(T.ey, y2);
                        if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, T.scale2[oct2])) ok = false;
                    }
                    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, x2), __fmul_rn(lb, y2)), lc);
                    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                    if (den == 0.0f || !((double)dsqr < 3.84 * (double)T.sigma2_2[oct2])) ok = false;      // :149-156
                }
            }
            const unsigned long long V = __ballot(ok);
            if (V == 0) continue;
            const unsigned long long mk = bm_argmin_mask(dist, V);
            const int last = 63 - __clzll((long long)mk);              // `dist <= bestDist` keeps the LAST of equal candidates
            const int wmin = __builtin_amdgcn_readlane(dist, last), ci = __builtin_amdgcn_readlane(idx2, last);
            if (wmin <= best) { best = wmin; bidx = ci; }
        }
        if (bidx >= 0 && lane == 0) {
            P.match12[idx1] = bidx;
            if (P.check_ori) {
                float rot = __fsub_rn(T.kp1[4 * idx1 + 2], T.kp2[4 * bidx + 2]);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / ORBHIP_HISTO_LENGTH));
                if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
                bin = min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
                P.bin12[idx1] = bin; atomicAdd(&P.hist[bin], 1);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bow_triangulate(TriParams T)
{
    bow_triangulate_body(T, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}
// one key frame against several neighbours in one launch (orbhip_search_for_triangulation_batch)
__global__ __launch_bounds__(256) void k_bow_triangulate_batch(const TriParams* Ts, const int* pref, int npairs)
{
    const int sl = bm_batch_slot(pref, npairs, blockIdx.x);
    bow_triangulate_body(Ts[sl], (blockIdx.x - pref[sl]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}
__global__ __launch_bounds__(256) void k_bow_triangulate_finish_batch(const TriParams* Ts) { bow_finish_body(Ts[blockIdx.x].M, threadIdx.x); }

// ------------------------------------------------------------------------------------------------ host side
// Workspace for nframes x cap features.  Grow-only (a transform per frame / key frame with a different feature count must not pay eleven
// hipFree + hipMalloc, each a device-wide synchronisation): `cap` is the row stride of every per-feature array, the feature count of a
// call travels separately.
struct BowWs {
    int frames = 0, cap = 0, last_frames = 0;
    uint8_t* d_desc = nullptr; uint32_t *d_word = nullptr, *d_node = nullptr; double* d_weight = nullptr;
    uint8_t* d_out = nullptr; BowOut out = {};                 // every frame's results as one block (BowOut)
    uint8_t* h_frame = nullptr; size_t h_frame_bytes = 0;      // pinned mirror of ONE frame's block (voc_fetch)
    uint8_t* h_desc = nullptr; size_t h_desc_bytes = 0;        // pinned mirror of the host-descriptor entry points' input
};
// The reference shares ONE vocabulary between the Tracking (Frame.cc:400), LocalMapping and LoopClosing threads (KeyFrame.cc:66) and its
// DBoW2 transform is read-only.  Here the tree is read-only too, the mutable state is not: `host` serves the host-descriptor entry points
// (transform, transform_features) and is held under `m` for the whole call, so concurrent callers take turns; compute_bow / fetch_bow work in
// a workspace that belongs to the extractor context they are called with (a context is used by one thread at a time, include/orbhip.h), found
// under `m`.
struct orbhip_voc {
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0, device = 0;
    BowNode* d_nodes = nullptr;      // breadth-first numbering
    hipStream_t stream = nullptr;
    std::mutex m; BowWs host; std::map<const orbhip_ctx*, BowWs> per_ctx;
};

static void voc_free_ws(BowWs* w)
{
    void* ptrs[] = {w->d_desc, w->d_word, w->d_node, w->d_weight, w->d_out};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (w->h_frame) (void)hipHostFree(w->h_frame);
    if (w->h_desc) (void)hipHostFree(w->h_desc);
    *w = BowWs();
}

#define BOW_MAX_FEATURES 8192          // eight keys per thread of k_bow_assemble: 8 B per key + 8 B per feature of LDS = 128 KB

static orbhip_status voc_ensure_ws(BowWs* w, int nframes, int cap, hipStream_t s)
{
    if (cap > BOW_MAX_FEATURES) return orbhip_set_error(ORBHIP_ERR_UNSUPPORTED, "%d features per frame (at most %d)", cap, BOW_MAX_FEATURES);
    if (nframes <= w->frames && cap <= w->cap) return ORBHIP_OK;
    BOWCHK(hipStreamSynchronize(s));                                               // nothing of this workspace is in flight when it is replaced
    const int nf = std::max(nframes, w->frames), nc = std::min(std::max(cap + cap / 4, w->cap), BOW_MAX_FEATURES);
    voc_free_ws(w);
    const size_t F = (size_t)nf, C = (size_t)nc;
    BOWCHK(orbhip_dmalloc((void**)&w->d_desc, F * C * 32 + 64));
    BOWCHK(orbhip_dmalloc((void**)&w->d_word, F * C * 4)); BOWCHK(orbhip_dmalloc((void**)&w->d_node, F * C * 4)); BOWCHK(orbhip_dmalloc((void**)&w->d_weight, F * C * 8));
    auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };      // 16-byte sections: the copy kernel moves 16 bytes per lane
    BowOut o; memset(&o, 0, sizeof o);
    o.o_id = (int)up(C * 8); o.o_fn = (int)up(o.o_id + C * 4); o.o_ff = (int)up(o.o_fn + C * 4); o.o_fo = (int)up(o.o_ff + C * 4); o.o_cnt = (int)up(o.o_fo + (C + 1) * 4);
    o.stride = (long long)((o.o_cnt + 32 + 255) & ~255);
    BOWCHK(orbhip_dmalloc((void**)&w->d_out, F * (size_t)o.stride));
    o.base = w->d_out; w->out = o;
    w->frames = nf; w->cap = nc;
    return ORBHIP_OK;
}

// nfeat_max: upper bound of the features of one frame in this call (sizes the grid and the LDS of the assembly); w->cap is the row stride
static orbhip_status voc_run(orbhip_voc* v, BowWs* w, const uint8_t* d_desc, long long frame_stride, const int* d_nfeat, int nfeat_fixed, int nframes, int nfeat_max,
                             int levelsup, bool assemble, hipStream_t s)
{
    BowParams P; memset(&P, 0, sizeof P);
    P.desc = d_desc; P.desc_frame_stride = frame_stride; P.nfeat = d_nfeat; P.nfeat_fixed = nfeat_fixed; P.cap = w->cap; P.lcap = nfeat_max;
    P.nodes = v->d_nodes;
    P.L = v->L; P.levelsup = levelsup;
    P.accumulate = (v->weighting == 0 || v->weighting == 1);                     // TF_IDF, TF (:1142) vs IDF, BINARY (:1173)
    P.must_normalize = v->scoring != 5; P.l2 = v->scoring == 1;                    // ScoringObject.h:73-90
    P.word = w->d_word; P.weight = w->d_weight; P.node = w->d_node;
    P.out = w->out;
    if (nfeat_max <= 0 || nframes <= 0) return ORBHIP_OK;
    hipLaunchKernelGGL(k_bow_descend, dim3((nfeat_max + 64 / BD_G - 1) / (64 / BD_G), nframes, 1), dim3(64, 1, 1), 0, s, P);
    if (assemble) {
        const int E = ba_keys_per_thread(nfeat_max);
        const char* force = getenv("ORBHIP_BOW_KEYS");                             // "64": the 8-byte keys also where 4 bytes would do (tests switch it between calls)
        const bool k32 = E <= 4 && ((unsigned long long)v->nnodes * E * BA_T) <= 0xffffffffull && !(force && force[0] == '6' && force[1] == '4');      // (E = 8: the values' own block would not fit the LDS)
        const dim3 g(nframes, 2, 1), b(BA_T, 1, 1); const size_t lds = ba_lds_bytes(nfeat_max, k32 ? 4 : 8);
        switch (E + (k32 ? 100 : 0)) {
        case 101: hipLaunchKernelGGL((k_bow_assemble<uint32_t, 1>), g, b, lds, s, P); break;
        case 102: hipLaunchKernelGGL((k_bow_assemble<uint32_t, 2>), g, b, lds, s, P); break;
        case 104: hipLaunchKernelGGL((k_bow_assemble<uint32_t, 4>), g, b, lds, s, P); break;
        case 1: hipLaunchKernelGGL((k_bow_assemble<unsigned long long, 1>), g, b, lds, s, P); break;
        case 2: hipLaunchKernelGGL((k_bow_assemble<unsigned long long, 2>), g, b, lds, s, P); break;
        case 4: hipLaunchKernelGGL((k_bow_assemble<unsigned long long, 4>), g, b, lds, s, P); break;
        default: hipLaunchKernelGGL((k_bow_assemble<unsigned long long, 8>), g, b, lds, s, P); break;
        }
    }
    BOWCHK(hipGetLastError());
    w->last_frames = nframes;
    return ORBHIP_OK;
}

// Every live vocabulary, so that a context that goes away can take its per-context workspaces with it (orbhip_destroy -> orbhip_bow_forget_ctx):
// the drop-in extractor re-creates its context on every image-size change, and a NEW context allocated at the old one's address must not
// inherit the old one's workspace (stale last_frames: orbhip_fetch_bow before orbhip_compute_bow would have returned the old frames' words).
static std::mutex g_vocs_m; static std::vector<orbhip_voc*> g_vocs;
void orbhip_bow_forget_ctx(const orbhip_ctx* ctx)
{
    std::lock_guard<std::mutex> all(g_vocs_m);
    for (orbhip_voc* v : g_vocs) {
        std::lock_guard<std::mutex> lock(v->m);
        auto it = v->per_ctx.find(ctx);
        if (it == v->per_ctx.end()) continue;
        (void)hipSetDevice(v->device);
        voc_free_ws(&it->second);                                       // the context's streams are drained by its destructor before this is called
        v->per_ctx.erase(it);
    }
}

extern "C" void orbhip_voc_destroy(orbhip_voc* v)
{
    if (!v) return;
    { std::lock_guard<std::mutex> all(g_vocs_m); g_vocs.erase(std::remove(g_vocs.begin(), g_vocs.end(), v), g_vocs.end()); }
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    (void)hipDeviceSynchronize();                                       // per-context workspaces were used on the extractors' streams
    voc_free_ws(&v->host);
    for (auto& kv : v->per_ctx) voc_free_ws(&kv.second);
    void* ptrs[] = {v->d_nodes};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

// TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1425).  Blank lines are ignored (the
// reference reads uninitialised locals on the empty last line of a file that ends with a newline).
extern "C" orbhip_status orbhip_voc_load_text(orbhip_voc** out, const char* path, int device)
{
    if (!out || !path) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    *out = nullptr;
    std::ifstream f(path);
    if (!f.is_open()) return orbhip_set_error(ORBHIP_ERR_INVALID, "cannot open vocabulary file %s", path);
    std::string s;
    std::getline(f, s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    { std::stringstream ss; ss << s; ss >> k; ss >> L; ss >> n1; ss >> n2; }
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)                 // :1358-1362
        return orbhip_set_error(ORBHIP_ERR_INVALID, "Vocabulary loading failure: This is not a correct text file!");
    std::vector<int> parent(1, 0), word(1, -1); std::vector<uint8_t> desc(32, 0); std::vector<double> weight(1, 0.0);
    std::vector<int> nchild(1, 0);
    int nwords = 0;
    while (std::getline(f, s)) {
        if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        const char* p = s.c_str(); char* e = nullptr;
        const int nid = (int)parent.size();
        const long pid = strtol(p, &e, 10); p = e;
        if (pid < 0 || pid >= nid) return orbhip_set_error(ORBHIP_ERR_INVALID, "vocabulary node %d names parent %ld", nid, pid);
        const long leaf = strtol(p, &e, 10); p = e;
        parent.push_back((int)pid); nchild.push_back(0); nchild[pid]++;
        desc.resize((size_t)(nid + 1) * 32);
        for (int i = 0; i < 32; i++) { const long b = strtol(p, &e, 10); p = e; desc[(size_t)nid * 32 + i] = (uint8_t)b; }   // FORB::fromString
        weight.push_back(strtod(p, &e));
        word.push_back(leaf > 0 ? nwords++ : -1);
    }
    const int nn = (int)parent.size();
    orbhip_voc* v = new orbhip_voc();
    v->k = k; v->L = L; v->scoring = n1; v->weighting = n2; v->nnodes = nn; v->nwords = nwords; v->device = device;
    std::vector<int> child_start(nn + 1, 0), child_ids(std::max(nn - 1, 1), 0);
    for (int i = 0; i < nn; i++) child_start[i + 1] = child_start[i] + nchild[i];
    { std::vector<int> fill(child_start.begin(), child_start.end() - 1); for (int i = 1; i < nn; i++) child_ids[fill[parent[i]]++] = i; }   // children in file order
    for (int i = 0; i < nn; i++)                                     // a leaf flag on an inner node / a childless inner node would make the descent undefined
        if ((word[i] >= 0) != (nchild[i] == 0) && i != 0) { delete v; return orbhip_set_error(ORBHIP_ERR_INVALID, "vocabulary node %d: leaf flag and children disagree", i); }
    // breadth-first numbering: the children of a node become neighbours, in the file's child order (parents precede their children in the file, so every
    // node hangs below the root and is reached)
    std::vector<int> order(1, 0); order.reserve(nn);
    std::vector<BowNode> nodes(nn);
    for (size_t q = 0; q < order.size(); q++) {
        const int id = order[q];
        BowNode& r = nodes[q];
        memcpy(&r.da, &desc[(size_t)id * 32], 16); memcpy(&r.db, &desc[(size_t)id * 32 + 16], 16);
        r.first_child = (uint32_t)order.size(); r.nchildren = (uint32_t)nchild[id]; r.id = (uint32_t)id; r.word = (uint32_t)word[id];
        r.weight = weight[id]; r.pad = 0.0;
        for (int c = child_start[id]; c < child_start[id + 1]; c++) order.push_back(child_ids[c]);
    }
    if ((int)order.size() != nn) { delete v; return orbhip_set_error(ORBHIP_ERR_INVALID, "vocabulary: %d of %d nodes hang below the root", (int)order.size(), nn); }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&v->d_nodes, (size_t)nn * sizeof(BowNode));
    if (e == hipSuccess) e = hipMemcpy(v->d_nodes, nodes.data(), (size_t)nn * sizeof(BowNode), hipMemcpyHostToDevice);
    if (e != hipSuccess) { orbhip_voc_destroy(v); return orbhip_set_error(ORBHIP_ERR_HIP, "vocabulary upload failed: %s", hipGetErrorString(e)); }
    { std::lock_guard<std::mutex> all(g_vocs_m); g_vocs.push_back(v); }
    *out = v;
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_voc_info(const orbhip_voc* v, int* k, int* L, int* scoring, int* weighting, int* nnodes, int* nwords)
{
    if (!v) return orbhip_set_error(ORBHIP_ERR_INVALID, "null vocabulary");
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return ORBHIP_OK;
}

static orbhip_status voc_upload(orbhip_voc* v, const uint8_t* desc, int n)
{
    BOWCHK(hipSetDevice(v->device));
    orbhip_status st = voc_ensure_ws(&v->host, 1, std::max(n, 1), v->stream); if (st != ORBHIP_OK) return st;
    if (n > 0) {
        BowWs& w = v->host; const size_t bytes = (size_t)n * 32;
        if (w.h_desc_bytes < bytes) {
            if (w.h_desc) { BOWCHK(hipStreamSynchronize(v->stream)); (void)hipHostFree(w.h_desc); }
            w.h_desc = nullptr; w.h_desc_bytes = 0;
            BOWCHK(hipHostMalloc((void**)&w.h_desc, bytes + bytes / 4, hipHostMallocDefault)); w.h_desc_bytes = bytes + bytes / 4;
        }
        memcpy(w.h_desc, desc, bytes);                                               // (the previous call synchronised the stream: the mirror is free)
        BOWCHK(orbhip_copy_async(w.d_desc, w.h_desc, bytes, hipMemcpyHostToDevice, v->stream));
    }
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_voc_transform_features(orbhip_voc* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node)
{
    OrbApiTimer api_timer;
    if (!v || (n > 0 && !desc) || n < 0) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    if (v->nwords == 0 || n == 0) return ORBHIP_OK;
    std::lock_guard<std::mutex> lock(v->m);                                        // upload, kernels and fetch of ONE caller at a time
    orbhip_status st = voc_upload(v, desc, n); if (st != ORBHIP_OK) return st;
    BowWs* w = &v->host;
    st = voc_run(v, w, w->d_desc, 0, nullptr, n, 1, n, levelsup, false, v->stream);
    if (st == ORBHIP_OK) {
        hipError_t e = hipSuccess;
        if (word) e = hipMemcpyAsync(word, w->d_word, (size_t)n * 4, hipMemcpyDeviceToHost, v->stream);
        if (e == hipSuccess && weight) e = hipMemcpyAsync(weight, w->d_weight, (size_t)n * 8, hipMemcpyDeviceToHost, v->stream);
        if (e == hipSuccess && node) e = hipMemcpyAsync(node, w->d_node, (size_t)n * 4, hipMemcpyDeviceToHost, v->stream);
        const hipError_t e2 = hipStreamSynchronize(v->stream);                     // also on the error path: the workspace is idle when the lock is released
        if (e != hipSuccess || e2 != hipSuccess) st = orbhip_set_error(ORBHIP_ERR_HIP, "orbhip_voc_transform_features: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    } else (void)hipStreamSynchronize(v->stream);
    return st;
}

// One frame's bag of words to the host: its block travels whole (ONE copy kernel on the stream, at the arrays' full row capacity - the counts are not
// known before the copy is issued) into a pinned mirror behind ONE synchronisation and is cut to size on the host.
static orbhip_status voc_fetch(BowWs* w, int frame, hipStream_t s, uint32_t* bow_id, double* bow_val, int* nbow, uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* nfv)
{
    const size_t C = (size_t)w->cap, total = (size_t)w->out.stride;
    if (w->h_frame_bytes < total) {
        if (w->h_frame) (void)hipHostFree(w->h_frame);
        w->h_frame = nullptr; w->h_frame_bytes = 0;
        BOWCHK(hipHostMalloc((void**)&w->h_frame, total + 64, hipHostMallocDefault)); w->h_frame_bytes = total;
    }
    uint8_t* h = w->h_frame;
    BOWCHK(orbhip_copy_async(h, w->d_out + (size_t)frame * total, total, hipMemcpyDeviceToHost, s));
    BOWCHK(hipStreamSynchronize(s));
    const BowOut& o = w->out;
    int cnt[2]; memcpy(&cnt[0], h + o.o_cnt, 4); memcpy(&cnt[1], h + o.o_cnt + 16, 4);
    const int nb = cnt[0], nf = cnt[1];
    if (nb < 0 || nf < 0 || (size_t)nb > C || (size_t)nf > C) return orbhip_set_error(ORBHIP_ERR_HIP, "bag of words: counts %d / %d outside the workspace's capacity %zu", nb, nf, C);
    if (nb > 0 && bow_id) memcpy(bow_id, h + o.o_id, (size_t)nb * 4);
    if (nb > 0 && bow_val) memcpy(bow_val, h, (size_t)nb * 8);
    if (nf > 0 && fv_node) memcpy(fv_node, h + o.o_fn, (size_t)nf * 4);
    if (fv_off) memcpy(fv_off, h + o.o_fo, (size_t)(nf + 1) * 4);
    if (fv_feat) { int m = 0; memcpy(&m, h + o.o_fo + (size_t)nf * 4, 4); if (m > 0 && (size_t)m <= C) memcpy(fv_feat, h + o.o_ff, (size_t)m * 4); }
    if (nbow) *nbow = nb;
    if (nfv) *nfv = nf;
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_voc_transform(orbhip_voc* v, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int* nbow,
                                              uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* nfv)
{
    OrbApiTimer api_timer;
    if (!v || (n > 0 && !desc) || n < 0) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    if (nbow) *nbow = 0;
    if (nfv) *nfv = 0;
    if (fv_off) fv_off[0] = 0;
    if (v->nwords == 0 || n == 0) return ORBHIP_OK;                              // v.clear(); fv.clear(); if(empty()) return;  (:1130-1136)
    std::lock_guard<std::mutex> lock(v->m);                                        // the reference's transform is const and called from three threads
    orbhip_status st = voc_upload(v, desc, n);
    if (st == ORBHIP_OK) st = voc_run(v, &v->host, v->host.d_desc, 0, nullptr, n, 1, n, levelsup, true, v->stream);
    if (st == ORBHIP_OK) st = voc_fetch(&v->host, 0, v->stream, bow_id, bow_val, nbow, fv_node, fv_off, fv_feat, nfv);
    if (st != ORBHIP_OK) (void)hipStreamSynchronize(v->stream);
    return st;
}

// Frame::ComputeBoW (Frame.cc:395-402) for the frames of the extractor's last call: descriptors are read where k_describe left them
extern "C" orbhip_status orbhip_compute_bow(orbhip_ctx* ctx, orbhip_voc* v, int nimg, int levelsup)
{
    OrbApiTimer api_timer;
    if (!ctx || !v) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    const uint8_t* d_desc = nullptr; const int* d_n = nullptr; int cap = 0, last = 0, device = 0; hipStream_t s = nullptr;
    orbhip_internal_outputs(ctx, &d_desc, &d_n, &cap, &last, &device, &s);
    if (nimg < 1 || nimg > last) return orbhip_set_error(ORBHIP_ERR_INVALID, "nimg %d but the last call processed %d frames", nimg, last);
    if (device != v->device) return orbhip_set_error(ORBHIP_ERR_INVALID, "extractor on device %d, vocabulary on device %d", device, v->device);
    BOWCHK(hipSetDevice(v->device));
    BowWs* w = nullptr;
    { std::lock_guard<std::mutex> lock(v->m); w = &v->per_ctx[ctx]; }             // std::map nodes do not move: the pointer outlives the lock
    orbhip_status st = voc_ensure_ws(w, nimg, cap, s); if (st != ORBHIP_OK) return st;
    if (v->nwords == 0) { BOWCHK(hipMemsetAsync(w->d_out, 0, (size_t)nimg * (size_t)w->out.stride, s)); w->last_frames = nimg; return ORBHIP_OK; }
    return voc_run(v, w, d_desc, (long long)cap * 32, d_n, 0, nimg, cap, levelsup, true, s);     // on the extractor's stream: ordered after k_describe
}

extern "C" orbhip_status orbhip_fetch_bow(orbhip_ctx* ctx, orbhip_voc* v, int frame, uint32_t* bow_id, double* bow_val, int* nbow,
                                          uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* nfv)
{
    OrbApiTimer api_timer;
    if (!ctx || !v) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    BowWs* w = nullptr;
    { std::lock_guard<std::mutex> lock(v->m); auto it = v->per_ctx.find(ctx); if (it != v->per_ctx.end()) w = &it->second; }
    if (!w || frame < 0 || frame >= w->last_frames) return orbhip_set_error(ORBHIP_ERR_INVALID, "frame %d outside the %d frames of this extractor's last orbhip_compute_bow", frame, w ? w->last_frames : 0);
    const uint8_t* d_desc = nullptr; const int* d_n = nullptr; int cap = 0, last = 0, device = 0; hipStream_t s = nullptr;
    orbhip_internal_outputs(ctx, &d_desc, &d_n, &cap, &last, &device, &s);
    BOWCHK(hipSetDevice(v->device));
    return voc_fetch(w, frame, s, bow_id, bow_val, nbow, fv_node, fv_off, fv_feat, nfv);
}

// TemplatedVocabulary::score -> the scoring object selected by the file header (ScoringObject.cpp:24-313); two ascending
// (id, value) arrays.  A few thousand flops on the host: not worth a launch.
extern "C" double orbhip_voc_score(const orbhip_voc* v, const uint32_t* id1, const double* val1, int n1, const uint32_t* id2, const double* val2, int n2)
{
    if (!v) return 0.0;
    const int sc = v->scoring;
    static const double LOG_EPS = std::log(2.220446049250313e-16);
    int a = 0, b = 0; double s = 0;
    while (a < n1 && b < n2) {
        const double vi = val1[a], wi = val2[b];
        if (id1[a] == id2[b]) {
            switch (sc) {
            case 0: s += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); break;
            case 1: case 5: s += vi * wi; break;
            case 2: if (vi + wi != 0.0) s += vi * wi / (vi + wi); break;
            case 3: if (vi != 0 && wi != 0) s += vi * std::log(vi / wi); break;
            case 4: s += std::sqrt(vi * wi); break;
            }
            a++; b++;
        } else if (id1[a] < id2[b]) {
            if (sc == 3) { s += vi * (std::log(vi) - LOG_EPS); a++; }
            else a = (int)(std::lower_bound(id1 + a, id1 + n1, id2[b]) - id1);
        } else b = (int)(std::lower_bound(id2 + b, id2 + n2, id1[a]) - id2);
    }
    switch (sc) {
    case 0: return -s / 2.0;
    case 1: return s >= 1 ? 1.0 : 1.0 - std::sqrt(1.0 - s);
    case 2: return 2. * s;
    case 3: for (; a < n1; a++) if (val1[a] != 0) s += val1[a] * (std::log(val1[a]) - LOG_EPS); return s;
    default: return s;
    }
}

void orbhip_bow_thread_release() {}      // (the two matchers below live in the calling thread's arena, orbhip_internal.h: nothing of their own to give back)

// ORBmatcher::SearchByBoW on flat data; see include/orbhip.h.  Every array of the call is one block of the thread's arena: ONE copy up (the -1 / 0
// initial values of the outputs included), two launches, ONE copy down, on the thread's own stream.
extern "C" orbhip_status orbhip_search_by_bow(int device, int mode,
    const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const uint32_t* fv1_node, const int32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
    const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const uint32_t* fv2_node, const int32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
    float nnratio, int check_ori, int32_t* match12, int* nmatches)
{
    OrbApiTimer api_timer;
    if (!match12 || !nmatches || n1 < 0 || n2 < 0 || nfv1 < 0 || nfv2 < 0 || (mode != 0 && mode != 1)) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || nfv1 == 0 || nfv2 == 0) return ORBHIP_OK;
    if (!desc1 || !desc2 || !angle1 || !angle2 || !valid1 || !fv1_node || !fv1_off || !fv1_feat || !fv2_node || !fv2_off || !fv2_feat || (mode == 1 && !valid2))
        return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    BOWCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    const int m1 = fv1_off[nfv1], m2 = fv2_off[nfv2];
    std::vector<uint8_t> ones;
    if (!valid2) { ones.assign(n2, 1); valid2 = ones.data(); }
    BowMatchParams P; memset(&P, 0, sizeof P);
    int zeros[ORBHIP_HISTO_LENGTH + 2] = {0}, tail[ORBHIP_HISTO_LENGTH + 2] = {0};
    uint8_t *d1 = nullptr, *d2 = nullptr, *v1 = nullptr, *v2 = nullptr; float *a1 = nullptr, *a2 = nullptr; uint32_t *fn1 = nullptr, *ff1 = nullptr, *fn2 = nullptr, *ff2 = nullptr;
    int *fo1 = nullptr, *fo2 = nullptr, *m12 = nullptr, *bin12 = nullptr, *hist = nullptr;
    BOWCHK(arena_layout(device, [&](Arena& A) {
        A.io(&d1, (size_t)n1 * 32, desc1, (size_t)n1 * 32); A.io(&a1, n1, angle1, n1); A.io(&v1, n1, valid1, n1);
        A.io(&fn1, nfv1, fv1_node, nfv1); A.io(&fo1, nfv1 + 1, (const int*)fv1_off, nfv1 + 1); A.io(&ff1, std::max(m1, 1), fv1_feat, m1);
        A.io(&d2, (size_t)n2 * 32, desc2, (size_t)n2 * 32); A.io(&a2, n2, angle2, n2); A.io(&v2, n2, valid2, n2);
        A.io(&fn2, nfv2, fv2_node, nfv2); A.io(&fo2, nfv2 + 1, (const int*)fv2_off, nfv2 + 1); A.io(&ff2, std::max(m2, 1), fv2_feat, m2);
        A.io(&bin12, n1, (const int*)match12, n1);                                        // bin12 = -1 (match12 holds n1 of them)
        A.io(&m12, n1, (const int*)match12, n1, (int*)match12, n1);                       // match12 = -1 in, the answer out
        A.io(&hist, ORBHIP_HISTO_LENGTH + 2, (const int*)zeros, ORBHIP_HISTO_LENGTH + 2, tail, ORBHIP_HISTO_LENGTH + 2);      // hist, nmatches, overflow
    }));
    hipError_t e = arena_upload(ts);
    if (e == hipSuccess) {
        P.mode = mode; P.nnratio = nnratio; P.check_ori = check_ori;
        P.d1 = d1; P.ang1 = a1; P.valid1 = v1; P.n1 = n1; P.fn1 = fn1; P.fo1 = fo1; P.ff1 = ff1; P.nf1 = nfv1;
        P.d2 = d2; P.ang2 = a2; P.valid2 = v2; P.n2 = n2; P.fn2 = fn2; P.fo2 = fo2; P.ff2 = ff2; P.nf2 = nfv2;
        P.match12 = m12; P.bin12 = bin12; P.hist = hist; P.nmatches = hist + ORBHIP_HISTO_LENGTH; P.overflow = P.nmatches + 1;
        hipLaunchKernelGGL(k_bow_match, dim3((nfv1 + 3) / 4, 1, 1), dim3(256, 1, 1), 0, ts, P);
        hipLaunchKernelGGL(k_bow_match_finish, dim3(1, 1, 1), dim3(256, 1, 1), 0, ts, P);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = arena_download(ts);
    if (e != hipSuccess) { (void)hipStreamSynchronize(ts); for (int i = 0; i < n1; i++) match12[i] = -1; return orbhip_set_error(ORBHIP_ERR_HIP, "search_by_bow: %s", hipGetErrorString(e)); }
    if (tail[ORBHIP_HISTO_LENGTH + 1]) { for (int i = 0; i < n1; i++) match12[i] = -1; return orbhip_set_error(ORBHIP_ERR_UNSUPPORTED, "a vocabulary node holds more than %d features of side 2", 64 * BM_CHUNKS); }
    *nmatches = tail[ORBHIP_HISTO_LENGTH];
    return ORBHIP_OK;
}

// ORBmatcher::SearchForTriangulation on flat data; see include/orbhip.h
extern "C" orbhip_status orbhip_search_for_triangulation(int device,
    const uint8_t* desc1, const float* kp1, const uint8_t* has_mp1, const uint8_t* stereo1, int n1, const uint32_t* fv1_node, const int32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
    const uint8_t* desc2, const float* kp2, const uint8_t* has_mp2, const uint8_t* stereo2, int n2, const uint32_t* fv2_node, const int32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
    const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels2, int only_stereo, int check_ori,
    int32_t* match12, int* nmatches)
{
    OrbApiTimer api_timer;
    if (!match12 || !nmatches || n1 < 0 || n2 < 0 || nfv1 < 0 || nfv2 < 0 || nlevels2 < 1) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || nfv1 == 0 || nfv2 == 0) return ORBHIP_OK;
    if (!desc1 || !desc2 || !kp1 || !kp2 || !has_mp1 || !has_mp2 || !stereo1 || !stereo2 || !fv1_node || !fv1_off || !fv1_feat || !fv2_node || !fv2_off || !fv2_feat ||
        !F12 || !scale_factors2 || !level_sigma2_2) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    BOWCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    const int m1 = fv1_off[nfv1], m2 = fv2_off[nfv2];
    int zeros[ORBHIP_HISTO_LENGTH + 2] = {0}, tail[ORBHIP_HISTO_LENGTH + 2] = {0};
    uint8_t *d1 = nullptr, *d2 = nullptr, *h1 = nullptr, *h2 = nullptr, *s1 = nullptr, *s2 = nullptr; float *k1 = nullptr, *k2 = nullptr, *sc2 = nullptr, *sg2 = nullptr;
    uint32_t *fn1 = nullptr, *ff1 = nullptr, *fn2 = nullptr, *ff2 = nullptr; int *fo1 = nullptr, *fo2 = nullptr, *m12 = nullptr, *bin12 = nullptr, *hist = nullptr;
    BOWCHK(arena_layout(device, [&](Arena& A) {
        A.io(&d1, (size_t)n1 * 32, desc1, (size_t)n1 * 32); A.io(&k1, (size_t)n1 * 4, kp1, (size_t)n1 * 4); A.io(&h1, n1, has_mp1, n1); A.io(&s1, n1, stereo1, n1);
        A.io(&fn1, nfv1, fv1_node, nfv1); A.io(&fo1, nfv1 + 1, (const int*)fv1_off, nfv1 + 1); A.io(&ff1, std::max(m1, 1), fv1_feat, m1);
        A.io(&d2, (size_t)n2 * 32, desc2, (size_t)n2 * 32); A.io(&k2, (size_t)n2 * 4, kp2, (size_t)n2 * 4); A.io(&h2, n2, has_mp2, n2); A.io(&s2, n2, stereo2, n2);
        A.io(&fn2, nfv2, fv2_node, nfv2); A.io(&fo2, nfv2 + 1, (const int*)fv2_off, nfv2 + 1); A.io(&ff2, std::max(m2, 1), fv2_feat, m2);
        A.io(&sc2, nlevels2, scale_factors2, nlevels2); A.io(&sg2, nlevels2, level_sigma2_2, nlevels2);
        A.io(&bin12, n1, (const int*)match12, n1);
        A.io(&m12, n1, (const int*)match12, n1, (int*)match12, n1);
        A.io(&hist, ORBHIP_HISTO_LENGTH + 2, (const int*)zeros, ORBHIP_HISTO_LENGTH + 2, tail, ORBHIP_HISTO_LENGTH + 2);
    }));
    hipError_t e = arena_upload(ts);
    if (e == hipSuccess) {
        TriParams T; memset(&T, 0, sizeof T);
        BowMatchParams& P = T.M;
        P.mode = 0; P.nnratio = 0.f; P.check_ori = check_ori;
        P.d1 = d1; P.valid1 = h1; P.n1 = n1; P.fn1 = fn1; P.fo1 = fo1; P.ff1 = ff1; P.nf1 = nfv1;
        P.d2 = d2; P.valid2 = h2; P.n2 = n2; P.fn2 = fn2; P.fo2 = fo2; P.ff2 = ff2; P.nf2 = nfv2;
        P.match12 = m12; P.bin12 = bin12; P.hist = hist; P.nmatches = hist + ORBHIP_HISTO_LENGTH; P.overflow = P.nmatches + 1;
        T.kp1 = k1; T.kp2 = k2; T.st1 = s1; T.st2 = s2;
        for (int i = 0; i < 9; i++) T.F[i] = F12[i];
        T.ex = ex; T.ey = ey; T.scale2 = sc2; T.sigma2_2 = sg2; T.only_stereo = only_stereo;
        hipLaunchKernelGGL(k_bow_triangulate, dim3((nfv1 + 3) / 4, 1, 1), dim3(256, 1, 1), 0, ts, T);
        hipLaunchKernelGGL(k_bow_match_finish, dim3(1, 1, 1), dim3(256, 1, 1), 0, ts, P);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = arena_download(ts);
    if (e != hipSuccess) { (void)hipStreamSynchronize(ts); for (int i = 0; i < n1; i++) match12[i] = -1; return orbhip_set_error(ORBHIP_ERR_HIP, "search_for_triangulation: %s", hipGetErrorString(e)); }
    *nmatches = tail[ORBHIP_HISTO_LENGTH];
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------ batched forms (include/orbhip.h)
// The back end calls these matchers in loops over neighbours / candidates (Tracking.cc:1357-1380, LoopClosing.cc:239-375,
// LocalMapping.cc:237-268); a call is launch latency, not work.  Here every distinct side travels once, all pairs run in ONE launch set (the
// parameter blocks in device memory, a prefix table from block to pair) and all answers come back in one copy.
namespace {
struct SideDev { uint8_t* d = nullptr; float* ang = nullptr; uint8_t* valid = nullptr; uint32_t* fn = nullptr; int* fo = nullptr; uint32_t* ff = nullptr; };
}
extern "C" orbhip_status orbhip_search_by_bow_batch(int device, int mode, int npairs, orbhip_bow_pair* pairs, float nnratio, int check_ori)
{
    OrbApiTimer api_timer;
    if (npairs < 0 || (npairs > 0 && !pairs) || (mode != 0 && mode != 1)) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    std::vector<const orbhip_bow_side*> sides;                       // distinct sides, in order of first use
    std::vector<int> live;                                           // pairs with work
    auto side_index = [&](const orbhip_bow_side* sd) { for (size_t i = 0; i < sides.size(); i++) if (sides[i] == sd) return (int)i; sides.push_back(sd); return (int)sides.size() - 1; };
    for (int p = 0; p < npairs; p++) {
        orbhip_bow_pair& Q = pairs[p];
        if (!Q.side1 || !Q.side2 || Q.side1->n < 0 || (Q.side1->n > 0 && !Q.match12) || Q.side2->n < 0 || Q.side1->nfv < 0 || Q.side2->nfv < 0) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument in pair %d", p);
        Q.nmatches = 0;
        for (int i = 0; i < Q.side1->n; i++) Q.match12[i] = -1;
        if (Q.side1->n == 0 || Q.side2->n == 0 || Q.side1->nfv == 0 || Q.side2->nfv == 0) continue;
        for (const orbhip_bow_side* sd : {Q.side1, Q.side2})
            if (!sd->desc || !sd->angle || !sd->fv_node || !sd->fv_off || !sd->fv_feat) return orbhip_set_error(ORBHIP_ERR_INVALID, "null array in pair %d", p);
        if (!Q.side1->valid || (mode == 1 && !Q.side2->valid)) return orbhip_set_error(ORBHIP_ERR_INVALID, "null validity flags in pair %d", p);
        live.push_back(p);
    }
    if (live.empty()) return ORBHIP_OK;
    BOWCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    const int NL = (int)live.size();
    std::vector<int> s1(NL), s2(NL), pref(NL + 1, 0);
    for (int k = 0; k < NL; k++) { s1[k] = side_index(pairs[live[k]].side1); s2[k] = side_index(pairs[live[k]].side2); pref[k + 1] = pref[k] + (pairs[live[k]].side1->nfv + 3) / 4; }
    int nmax2 = 1; for (const orbhip_bow_side* sd : sides) nmax2 = std::max(nmax2, sd->n);
    const std::vector<uint8_t> ones(nmax2, 1);                       // mode 0 ignores side 2's flags
    std::vector<SideDev> D(sides.size());
    std::vector<BowMatchParams> hP(NL);
    std::vector<std::array<int, ORBHIP_HISTO_LENGTH + 2>> zeros(NL), tail(NL);
    for (auto& z : zeros) z.fill(0);
    BowMatchParams* dP = nullptr; int* dpref = nullptr;
    std::vector<int*> dm12(NL), dbin(NL), dhist(NL);
    BOWCHK(arena_layout(device, [&](Arena& A) {
        A.io(&dP, (size_t)NL, (const BowMatchParams*)hP.data(), (size_t)NL);                 // (filled below, read when the arena is uploaded)
        A.io(&dpref, (size_t)NL + 1, (const int*)pref.data(), (size_t)NL + 1);
        for (size_t i = 0; i < sides.size(); i++) {
            const orbhip_bow_side& sd = *sides[i]; const int m = sd.fv_off[sd.nfv];
            A.io(&D[i].d, (size_t)sd.n * 32, sd.desc, (size_t)sd.n * 32); A.io(&D[i].ang, sd.n, sd.angle, sd.n);
            A.io(&D[i].valid, sd.n, sd.valid ? sd.valid : ones.data(), sd.n);
            A.io(&D[i].fn, sd.nfv, sd.fv_node, sd.nfv); A.io(&D[i].fo, sd.nfv + 1, (const int*)sd.fv_off, sd.nfv + 1); A.io(&D[i].ff, std::max(m, 1), sd.fv_feat, m);
        }
        for (int k = 0; k < NL; k++) {
            orbhip_bow_pair& Q = pairs[live[k]]; const int n1 = Q.side1->n;
            A.io(&dbin[k], n1, (const int*)Q.match12, n1);                                   // bin12 = -1
            A.io(&dm12[k], n1, (const int*)Q.match12, n1, (int*)Q.match12, n1);             // match12 = -1 in, the answer out
            A.io(&dhist[k], ORBHIP_HISTO_LENGTH + 2, (const int*)zeros[k].data(), ORBHIP_HISTO_LENGTH + 2, tail[k].data(), ORBHIP_HISTO_LENGTH + 2);
        }
    }));
    for (int k = 0; k < NL; k++) {
        const orbhip_bow_pair& Q = pairs[live[k]]; BowMatchParams& P = hP[k]; memset(&P, 0, sizeof P);
        const SideDev &a = D[s1[k]], &b = D[s2[k]];
        P.mode = mode; P.nnratio = nnratio; P.check_ori = check_ori;
        P.d1 = a.d; P.ang1 = a.ang; P.valid1 = a.valid; P.n1 = Q.side1->n; P.fn1 = a.fn; P.fo1 = a.fo; P.ff1 = a.ff; P.nf1 = Q.side1->nfv;
        P.d2 = b.d; P.ang2 = b.ang; P.valid2 = b.valid; P.n2 = Q.side2->n; P.fn2 = b.fn; P.fo2 = b.fo; P.ff2 = b.ff; P.nf2 = Q.side2->nfv;
        P.match12 = dm12[k]; P.bin12 = dbin[k]; P.hist = dhist[k]; P.nmatches = dhist[k] + ORBHIP_HISTO_LENGTH; P.overflow = P.nmatches + 1;
    }
    hipError_t e = arena_upload(ts);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_bow_match_batch, dim3(pref[NL], 1, 1), dim3(256, 1, 1), 0, ts, (const BowMatchParams*)dP, (const int*)dpref, NL);
        hipLaunchKernelGGL(k_bow_match_finish_batch, dim3(NL, 1, 1), dim3(256, 1, 1), 0, ts, (const BowMatchParams*)dP);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = arena_download(ts);
    bool overflow = false;
    for (int k = 0; k < NL; k++) overflow = overflow || tail[k][ORBHIP_HISTO_LENGTH + 1] != 0;
    if (e != hipSuccess || overflow) {
        if (e != hipSuccess) (void)hipStreamSynchronize(ts);
        for (int k = 0; k < NL; k++) { orbhip_bow_pair& Q = pairs[live[k]]; for (int i = 0; i < Q.side1->n; i++) Q.match12[i] = -1; }
        return e != hipSuccess ? orbhip_set_error(ORBHIP_ERR_HIP, "search_by_bow_batch: %s", hipGetErrorString(e))
                               : orbhip_set_error(ORBHIP_ERR_UNSUPPORTED, "a vocabulary node holds more than %d features of side 2", 64 * BM_CHUNKS);
    }
    for (int k = 0; k < NL; k++) pairs[live[k]].nmatches = tail[k][ORBHIP_HISTO_LENGTH];
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_search_for_triangulation_batch(int device, const orbhip_tri_side* kf1, int npairs, orbhip_tri_pair* pairs, int only_stereo, int check_ori)
{
    OrbApiTimer api_timer;
    if (!kf1 || npairs < 0 || (npairs > 0 && !pairs) || kf1->n < 0 || kf1->nfv < 0) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    std::vector<int> live;
    for (int p = 0; p < npairs; p++) {
        orbhip_tri_pair& Q = pairs[p];
        if (!Q.kf2 || (kf1->n > 0 && !Q.match12) || Q.kf2->n < 0 || Q.kf2->nfv < 0 || Q.kf2->nlevels < 1) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument in pair %d", p);
        Q.nmatches = 0;
        for (int i = 0; i < kf1->n; i++) Q.match12[i] = -1;
        if (kf1->n == 0 || Q.kf2->n == 0 || kf1->nfv == 0 || Q.kf2->nfv == 0) continue;
        const orbhip_tri_side& b = *Q.kf2;
        if (!b.desc || !b.kp || !b.has_mp || !b.stereo || !b.fv_node || !b.fv_off || !b.fv_feat || !b.scale_factors || !b.level_sigma2) return orbhip_set_error(ORBHIP_ERR_INVALID, "null array in pair %d", p);
        live.push_back(p);
    }
    if (live.empty()) return ORBHIP_OK;
    if (!kf1->desc || !kf1->kp || !kf1->has_mp || !kf1->stereo || !kf1->fv_node || !kf1->fv_off || !kf1->fv_feat) return orbhip_set_error(ORBHIP_ERR_INVALID, "null array in key frame 1");
    BOWCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    const int NL = (int)live.size(), n1 = kf1->n, m1 = kf1->fv_off[kf1->nfv], blocks1 = (kf1->nfv + 3) / 4;
    std::vector<int> pref(NL + 1, 0);
    for (int k = 0; k < NL; k++) pref[k + 1] = pref[k] + blocks1;
    std::vector<TriParams> hT(NL);
    std::vector<std::array<int, ORBHIP_HISTO_LENGTH + 2>> zeros(NL), tail(NL);
    for (auto& z : zeros) z.fill(0);
    TriParams* dT = nullptr; int* dpref = nullptr;
    uint8_t *d1 = nullptr, *h1 = nullptr, *st1 = nullptr; float* k1 = nullptr; uint32_t *fn1 = nullptr, *ff1 = nullptr; int* fo1 = nullptr;
    struct Side2 { uint8_t *d, *h, *st; float *k, *sc, *sg; uint32_t *fn, *ff; int* fo; };
    std::vector<Side2> B(NL); std::vector<int*> dm12(NL), dbin(NL), dhist(NL);
    BOWCHK(arena_layout(device, [&](Arena& A) {
        A.io(&dT, (size_t)NL, (const TriParams*)hT.data(), (size_t)NL);
        A.io(&dpref, (size_t)NL + 1, (const int*)pref.data(), (size_t)NL + 1);
        A.io(&d1, (size_t)n1 * 32, kf1->desc, (size_t)n1 * 32); A.io(&k1, (size_t)n1 * 4, kf1->kp, (size_t)n1 * 4); A.io(&h1, n1, kf1->has_mp, n1); A.io(&st1, n1, kf1->stereo, n1);
        A.io(&fn1, kf1->nfv, kf1->fv_node, kf1->nfv); A.io(&fo1, kf1->nfv + 1, (const int*)kf1->fv_off, kf1->nfv + 1); A.io(&ff1, std::max(m1, 1), kf1->fv_feat, m1);
        for (int k = 0; k < NL; k++) {
            orbhip_tri_pair& Q = pairs[live[k]]; const orbhip_tri_side& b = *Q.kf2; const int n2 = b.n, m2 = b.fv_off[b.nfv]; Side2& S = B[k];
            A.io(&S.d, (size_t)n2 * 32, b.desc, (size_t)n2 * 32); A.io(&S.k, (size_t)n2 * 4, b.kp, (size_t)n2 * 4); A.io(&S.h, n2, b.has_mp, n2); A.io(&S.st, n2, b.stereo, n2);
            A.io(&S.fn, b.nfv, b.fv_node, b.nfv); A.io(&S.fo, b.nfv + 1, (const int*)b.fv_off, b.nfv + 1); A.io(&S.ff, std::max(m2, 1), b.fv_feat, m2);
            A.io(&S.sc, b.nlevels, b.scale_factors, b.nlevels); A.io(&S.sg, b.nlevels, b.level_sigma2, b.nlevels);
            A.io(&dbin[k], n1, (const int*)Q.match12, n1);
        }
        // everything that travels back, of every pair, side by side behind the inputs: arena_download copies ONE span [first dst, last dst)
        for (int k = 0; k < NL; k++) {
            orbhip_tri_pair& Q = pairs[live[k]];
            A.io(&dm12[k], n1, (const int*)Q.match12, n1, (int*)Q.match12, n1);
            A.io(&dhist[k], ORBHIP_HISTO_LENGTH + 2, (const int*)zeros[k].data(), ORBHIP_HISTO_LENGTH + 2, tail[k].data(), ORBHIP_HISTO_LENGTH + 2);
        }
    }));
    for (int k = 0; k < NL; k++) {
        const orbhip_tri_pair& Q = pairs[live[k]]; const Side2& S = B[k]; TriParams& T = hT[k]; memset(&T, 0, sizeof T);
        BowMatchParams& P = T.M;
        P.mode = 0; P.nnratio = 0.f; P.check_ori = check_ori;
        P.d1 = d1; P.valid1 = h1; P.n1 = n1; P.fn1 = fn1; P.fo1 = fo1; P.ff1 = ff1; P.nf1 = kf1->nfv;
        P.d2 = S.d; P.valid2 = S.h; P.n2 = Q.kf2->n; P.fn2 = S.fn; P.fo2 = S.fo; P.ff2 = S.ff; P.nf2 = Q.kf2->nfv;
        P.match12 = dm12[k]; P.bin12 = dbin[k]; P.hist = dhist[k]; P.nmatches = dhist[k] + ORBHIP_HISTO_LENGTH; P.overflow = P.nmatches + 1;
        T.kp1 = k1; T.kp2 = S.k; T.st1 = st1; T.st2 = S.st;
        for (int i = 0; i < 9; i++) T.F[i] = Q.F12[i];
        T.ex = Q.ex; T.ey = Q.ey; T.scale2 = S.sc; T.sigma2_2 = S.sg; T.only_stereo = only_stereo;
    }
    hipError_t e = arena_upload(ts);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_bow_triangulate_batch, dim3(pref[NL], 1, 1), dim3(256, 1, 1), 0, ts, (const TriParams*)dT, (const int*)dpref, NL);
        hipLaunchKernelGGL(k_bow_triangulate_finish_batch, dim3(NL, 1, 1), dim3(256, 1, 1), 0, ts, (const TriParams*)dT);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = arena_download(ts);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ts);
        for (int k = 0; k < NL; k++) { orbhip_tri_pair& Q = pairs[live[k]]; for (int i = 0; i < n1; i++) Q.match12[i] = -1; }
        return orbhip_set_error(ORBHIP_ERR_HIP, "search_for_triangulation_batch: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < NL; k++) pairs[live[k]].nmatches = tail[k][ORBHIP_HISTO_LENGTH];
    return ORBHIP_OK;
}
