// orbhip_kernels_match.hip — gfx950 kernels of the matcher side (replaces the Hamming / Frame-to-Frame part of
// src/ORBmatcher.cc and the Frame feature grid it reads).
//
//   k_hamming_nn / k_hamming_merge   brute-force 256-bit Hamming NN, queries in registers, DB rows broadcast from LDS,
//                                    __popcll on 4 x u64 per pair (ORBmatcher::DescriptorDistance, ORBmatcher.cc:1647-1663;
//                                    best/second idiom :102-114, :447-456).  Integer-VALU bound (v_xor + v_bcnt); small databases.
//   k_hamming_nn_mfma                the same scan as +-1 i8 products on the matrix cores (databases from 32 K rows on)
//   k_match_grid                     Frame::AssignFeaturesToGrid (Frame.cc:230-245, 382-392): 64x48 buckets, keypoint order
//   k_match_candidates               Frame::GetFeaturesInArea (Frame.cc:327-380) in canonical order + all Hamming distances,
//                                    one wavefront per previous-frame keypoint; also records each key point's four best candidates
//   k_match_select                   the order-dependent part of SearchForInitialization (ORBmatcher.cc:418-517): one
//                                    wavefront per camera slot decides 64 key points per step from those records (a candidate is
//                                    skipped once matched at a distance <= the query's), rescans the rare key point whose records
//                                    are used up, then rotation histogram, ComputeThreeMaxima (:1601-1642), vbPrevMatched update.
#include "orbhip_internal.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define WAVE 64
#define IMAX 0x7fffffff

// ------------------------------------------------------------------------------------------------ brute-force NN
#define NN_T 256
#define NN_QPT 4                       // queries held in registers per thread
#define NN_ROWS 256                    // DB rows staged in LDS per step (8 KB)
#define NN_CHUNK 8192                  // DB rows per workgroup

struct NNPart { int best, second; long long idx; };

__global__ __launch_bounds__(NN_T) void k_hamming_nn(const unsigned long long* q, int nq, const unsigned long long* db, long long ndb,
                                                     long long base, NNPart* parts, int nchunks)
{
    __shared__ unsigned long long s_db[NN_ROWS * 4];
    const int tid = threadIdx.x, chunk = blockIdx.y;
    const int q0 = (blockIdx.x * NN_T + tid) * NN_QPT;
    unsigned long long qa[NN_QPT][4];
    // per query: the two smallest keys (distance << 13 | row-in-chunk).  Keys are unique and ordered like (distance, index), so
    //   best' = min(best, x);  second' = min(second, max(best, x))
    // is exactly the matcher's `if (d < best) {second = best; best = d;} else if (d < second) second = d` with lowest-index ties.
    unsigned kb[NN_QPT], ks[NN_QPT];
#pragma unroll
    for (int k = 0; k < NN_QPT; k++) {
        const int qi = min(q0 + k, nq - 1);
#pragma unroll
        for (int w = 0; w < 4; w++) qa[k][w] = q[(long long)qi * 4 + w];
        kb[k] = 0xffffffffu; ks[k] = 0xffffffffu;
    }
    const long long row0 = (long long)chunk * NN_CHUNK;
    const long long row1 = min(row0 + (long long)NN_CHUNK, ndb);
    for (long long r = row0; r < row1; r += NN_ROWS) {
        const int nr = (int)min((long long)NN_ROWS, row1 - r);
        __syncthreads();
        for (int i = tid; i < nr * 4; i += NN_T) s_db[i] = db[r * 4 + i];
        __syncthreads();
        const unsigned jbase = (unsigned)(r - row0);
        for (int j = 0; j < nr; j++) {
            const unsigned long long d0 = s_db[4 * j], d1 = s_db[4 * j + 1], d2 = s_db[4 * j + 2], d3 = s_db[4 * j + 3];   // LDS broadcast
#pragma unroll
            for (int k = 0; k < NN_QPT; k++) {
                const unsigned d = (unsigned)(__popcll(qa[k][0] ^ d0) + __popcll(qa[k][1] ^ d1) + __popcll(qa[k][2] ^ d2) + __popcll(qa[k][3] ^ d3));
                const unsigned x = (d << 13) | (jbase + j);
                ks[k] = min(ks[k], max(kb[k], x));
                kb[k] = min(kb[k], x);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NN_QPT; k++)
        if (q0 + k < nq) {
            NNPart p;
            p.best = kb[k] == 0xffffffffu ? IMAX : (int)(kb[k] >> 13);
            p.second = ks[k] == 0xffffffffu ? IMAX : (int)(ks[k] >> 13);
            p.idx = kb[k] == 0xffffffffu ? -1 : row0 + (long long)(kb[k] & 0x1fffu) + base;
            parts[(long long)(q0 + k) * nchunks + chunk] = p;
        }
}

// ---- the same scan on the matrix cores.  With descriptor bits as +-1, <a, b> = 256 - 2 * Hamming(a, b): the distance table of 32 DB rows x
// 32 queries is one 32 x 32 x 256 i8 product (8 x v_mfma_i32_32x32x32_i8), exact, and the popcount formulation's ceiling (1.7e12 pair
// distances/s, VALU-bound on v_xor + v_bcnt) does not apply.  A workgroup keeps 256 queries as B operands in registers (4 waves x 2 tiles of
// 32 queries, lane = query) and streams its chunk of DB rows: 32 rows at a time are expanded from bits to +-1 bytes through a 256-entry LDS
// table (once per workgroup, shared by its eight query tiles) and read back as A operands (lane = row, bytes = 16 bits of one dword).
// The result leaves with lane = query, register = DB row, so the running (best, second) keys of a query are two registers of its lane.
// The accumulators ARE the tile's keys, with no VALU work per pair: the queries are expanded to -+64 instead of -+1 and the C operand of the
// first product of a tile is the constant 256 * 64 + row-in-tile, so that   acc = (256 - dot) * 64 + r = distance << 7 | r   (r < 64).
// A v_min3 / v_med3 tournament picks the tile's two smallest, and only those two are turned into chunk keys distance << 13 | row-in-chunk
// (three operations each) for   second = min(second, max(best, key)), best = min(best, key)
// exactly as in k_hamming_nn, whose partial format and merge kernel are reused.
typedef int nn_v4i __attribute__((vector_size(16)));
typedef int nn_v16i __attribute__((vector_size(64)));
#define NNM_QT 2                        // query tiles (of 32) per wavefront (3: 9.7 ms against 7.9 - a sixth workgroup column of padding; 4: spills)
#define NNM_QG (4 * NNM_QT * 32)        // queries per workgroup
#define ORBHIP_NN_DEFAULT 2             // form of the matrix-core scan orbhip_launch_hamming_nn takes (1 = i8, 2 = FP4 in the shape below: 5.3 ms against 8.0 for 2000 x 20 M, profiles/r05_exp_config5_fp4_shapes.jsonl)
#define ORBHIP_NN_FP4_QT 4
#define ORBHIP_NN_FP4_OCC 2
#define ORBHIP_NN_FP4_LCH 15
#define ORBHIP_NN_FP4_TPB 6
__global__ __launch_bounds__(256, 2) void k_hamming_nn_mfma(const unsigned* q, int nq, const unsigned* db, long long ndb, long long base, NNPart* parts, int nchunks)
{
    __shared__ unsigned long long s_tab[256];                          // byte -> its 8 bits as +-1 bytes
    __shared__ __attribute__((aligned(16))) unsigned s_a[2][16 * 32 * 4];   // expanded DB tile: [K block kb][lane half h][row i] x 16 bytes, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), chunk = blockIdx.y;
    {
        unsigned long long e = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) e |= (unsigned long long)(((tid >> t) & 1) ? 0x01u : 0xffu) << (8 * t);
        s_tab[tid] = e;
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    // ---- the wave's queries as B operands: lane (j, h), K block kb = bits 16h .. 16h+15 of dword kb of query j
    nn_v4i B[NNM_QT][8];
    int qidx[NNM_QT];
#pragma unroll
    for (int t = 0; t < NNM_QT; t++) {
        qidx[t] = blockIdx.x * NNM_QG + (wave * NNM_QT + t) * 32 + j;
        const unsigned* qp = q + (long long)min(qidx[t], nq - 1) * 8;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            const unsigned half = (qp[kb] >> (16 * h)) & 0xffffu;
            // the queries carry the opposite sign and the weight 64: table byte 0x01 (bit set) -> 0xc0 = -64, 0xff (bit clear) -> 0x40 = +64
            const unsigned long long lo = (s_tab[half & 0xff] & 0x8080808080808080ull) ^ 0xc0c0c0c0c0c0c0c0ull, hi = (s_tab[half >> 8] & 0x8080808080808080ull) ^ 0xc0c0c0c0c0c0c0c0ull;
            B[t][kb] = nn_v4i{(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, (int)(unsigned)(hi >> 32)};
        }
    }
    unsigned kbest[NNM_QT], ksec[NNM_QT];
#pragma unroll
    for (int t = 0; t < NNM_QT; t++) { kbest[t] = 0xffffffffu; ksec[t] = 0xffffffffu; }
    const long long row0 = (long long)chunk * NN_CHUNK;
    const int nrows = (int)min((long long)NN_CHUNK, ndb - row0);
    const int ntiles = (nrows + 31) >> 5;
    // staging role of this thread: row r of the tile, dword kb of that row
    const int sr = tid & 31, skb = tid >> 5;
    auto stage = [&](int tile, int buf) {
        const int r = tile * 32 + sr;
        const unsigned w = r < nrows ? db[(row0 + r) * 8 + skb] : 0u;
        unsigned* d0 = s_a[buf] + ((skb * 2 + 0) * 32 + sr) * 4;        // bits 0..15 -> lane half 0
        unsigned* d1 = s_a[buf] + ((skb * 2 + 1) * 32 + sr) * 4;        // bits 16..31 -> lane half 1
        // (the same expansion by VALU arithmetic - n * 0x204081 & 0x01010101, * 0xfe, complement: 40 VALU per thread and tile - measured 8.3 against 7.9 ms)
        const unsigned long long e0 = s_tab[w & 0xff], e1 = s_tab[(w >> 8) & 0xff], e2 = s_tab[(w >> 16) & 0xff], e3 = s_tab[w >> 24];
        *reinterpret_cast<uint4*>(d0) = uint4{(unsigned)e0, (unsigned)(e0 >> 32), (unsigned)e1, (unsigned)(e1 >> 32)};
        *reinterpret_cast<uint4*>(d1) = uint4{(unsigned)e2, (unsigned)(e2 >> 32), (unsigned)e3, (unsigned)(e3 >> 32)};
    };
    // C operand of a tile's first product: 256 * 64 + row-in-tile of D's register reg = (reg & 3) + 8 (reg >> 2) + 4 h (constant registers, shared by the query tiles)
    nn_v16i cinit;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) cinit[reg] = 256 * 64 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    // best two of three keys in two instructions (v_min3 / v_med3), then (best, second) pairs merged in three
    auto top2_of3 = [](unsigned a, unsigned b, unsigned c, unsigned& lo, unsigned& mid) { lo = min(min(a, b), c); mid = max(min(a, b), min(max(a, b), c)); };
    auto merge2 = [](unsigned& b, unsigned& s2, unsigned ob, unsigned os) { s2 = min(min(s2, os), max(b, ob)); b = min(b, ob); };
    // One tile of 32 DB rows against the wave's query tiles: products, then the (best, second) selection.  `ragged` (compile-time) = the chunk's
    // last, partial tile, whose rows past the chunk get keys no real row beats.  (Measured in round 3, 2000 x 20 M: this form 7.8 ms; the same with
    // the products of tile t issued before the selection of tile t-1 on a second accumulator set 10.0 ms - the wave's own matrix / VALU overlap
    // costs more registers and scheduling freedom than the two resident workgroups already provide; four query tiles per wave 10.4 ms, spilling.)
    auto products = [&](int tile, nn_v16i (&acc)[NNM_QT]) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            const nn_v4i A = *reinterpret_cast<const nn_v4i*>(s_a[buf] + ((kb * 2 + h) * 32 + j) * 4);     // lane (i = j, h): row i of the tile
#pragma unroll
            for (int t = 0; t < NNM_QT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[t][kb], kb == 0 ? cinit : acc[t], 0, 0, 0);
        }
    };
    auto select = [&](int tile, const nn_v16i (&acc)[NNM_QT], auto ragged) {
        const unsigned tbase = (unsigned)tile * 32u;
#pragma unroll
        for (int t = 0; t < NNM_QT; t++) {
            unsigned x[16];
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                x[reg] = (unsigned)acc[t][reg];                                                              // distance << 7 | row-in-tile
                if (decltype(ragged)::value && tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h >= nrows) x[reg] = (511u << 7) + (unsigned)((reg & 3) + 8 * (reg >> 2) + 4 * h);   // distance 511: loses to every real row
            }
            unsigned b, s2;
            top2_of3(x[0], x[1], x[2], b, s2);
#pragma unroll
            for (int g = 1; g < 5; g++) { unsigned lo, mid; top2_of3(x[3 * g], x[3 * g + 1], x[3 * g + 2], lo, mid); merge2(b, s2, lo, mid); }
            merge2(b, s2, x[15], 0xffffffffu);
            // tile key d << 7 | r  ->  chunk key d << 13 | (tile * 32 + r):  (k << 6) - 63 r + tile * 32  with r = k & 127 (bit 6 of r is clear)
            const unsigned kb1 = (b << 6) - 63u * (b & 127u) + tbase, ks1 = (s2 << 6) - 63u * (s2 & 127u) + tbase;     // (both are real keys: 16 rows per lane)
            merge2(kbest[t], ksec[t], kb1, ks1);
        }
    };
    stage(0, 0);
    __syncthreads();
    const int nfull = nrows >> 5;                                      // whole tiles; tile nfull (if any) is the ragged one
    nn_v16i acc[NNM_QT];
    for (int tile = 0; tile < ntiles; tile++) {
        products(tile, acc);
        if (tile < nfull) select(tile, acc, std::false_type{}); else select(tile, acc, std::true_type{});
        __syncthreads();
    }
    // ---- a query's rows were split over its two lanes (j, 0) and (j, 1): fold, then one partial per (query, chunk)
#pragma unroll
    for (int t = 0; t < NNM_QT; t++) {
        const unsigned ob = (unsigned)__shfl_xor((int)kbest[t], 32), os = (unsigned)__shfl_xor((int)ksec[t], 32);
        const unsigned b = min(kbest[t], ob), s2 = min(min(ksec[t], os), max(kbest[t], ob));
        if (h == 0 && qidx[t] < nq) {
            NNPart p;
            p.best = (b >> 13) > 256u ? IMAX : (int)(b >> 13);                     // never-set keys and the ragged tile's filler rows carry a distance above 256
            p.second = (s2 >> 13) > 256u ? IMAX : (int)(s2 >> 13);
            p.idx = (b >> 13) > 256u ? -1 : row0 + (long long)(b & 0x1fffu) + base;
            parts[(long long)qidx[t] * nchunks + chunk] = p;
        }
    }
}

// ---- the same scan on the FP4 matrix path (gfx950 only: v_mfma_scale_f32_32x32x64_f8f6f4, twice the i8 rate).  +-1 is exact in E2M1 (+1 = 0x2,
// -1 = 0xA); the query side's block scale 2^6 (E8M0 133) gives the weight 64, so a product is -+64, the f32 accumulator holds integers below 2^16 -
// exact - and with the same C operand 256 * 64 + row-in-tile a tile's accumulators are again its keys  distance << 7 | row-in-tile.  Positive floats
// order like their bit patterns: the v_min3 / v_med3 tournament runs on the raw registers and only the two winners are converted.  A descriptor is 256
// nibbles = 8 x 16 bytes: the expanded DB tile is 4 KB (the i8 form's is 8 KB) and a query tile costs 16 operand registers instead of 32, which is what
// lets a wave keep QT = 3 or 4 query tiles (the i8 form spills at 4) and amortise the tile's expansion and operand reads over more queries.
typedef int nn_v8i __attribute__((vector_size(32)));
typedef float nn_v16f __attribute__((vector_size(64)));
// Seeded scan (round 6): a query's FINAL second-best distance is at most the second-best distance over ANY subset of the rows; seed[q] carries that bound from
// a first pass over the database's head (rows 0 .. 2^15).  A workgroup whose rows all lie BEHIND the head starts its skip threshold at the bound instead of at
// "nothing seen yet": a tile can only matter if it holds a distance strictly below the bound (a row AT the bound loses the tie to the head's rows, which have
// lower indices) - so the skip works from a chunk's first tile on (about one tile in sixteen survives instead of one in three).  chunk0 = index of this
// launch's first chunk in rows / CH; part0, part_stride = where its partials go among all partials of the query (head sub-chunks first, then the rest).
// EXP (round 6): `db` is the database EXPANDED in device memory (k_nn_expand: 128 B per row, 4 KB per tile of 32 rows in exactly the layout of the LDS tile),
// so staging a tile is ONE 16-byte LDS-DMA per thread - no registers, no byte -> E2M1 table, no VALU - requested a whole superstep ahead.
template <int QT, int OCC, int LCH, int TPB, int ABL = 0, bool EXP = false> __global__ __launch_bounds__(256, OCC) void k_hamming_nn_fp4(const unsigned* q, int nq, const unsigned* db, long long ndb, long long base, NNPart* parts, int nchunks,
                                                                                                          const int* seed, int chunk0, int part0)
{
    constexpr int ablate = ABL;                                        // measurement only (ORBHIP_NN_ABLATE): 1 = no threshold tests, 2 = no staging of new tiles; results are wrong
    __shared__ unsigned s_tab[256];                                    // byte -> its 8 bits as FP4 nibbles (bit k -> nibble k): set = +1 (0x2), clear = -1 (0xA)
    __shared__ __attribute__((aligned(16))) unsigned s_a[2][TPB * 8 * 32 * 4];   // TPB expanded DB tiles per workgroup barrier: [tile u][dword d of the row = 2 kb + h][row i] x 16 bytes, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), chunk = blockIdx.y + chunk0;
    {
        unsigned e = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) e |= (((tid >> t) & 1) ? 0x2u : 0xAu) << (4 * t);
        s_tab[tid] = e;
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    constexpr int QG = 4 * QT * 32;
    // ---- the wave's queries as B operands: lane (j, h), K block kb = dword 2 kb + h of query j, the SAME sign as the rows (set bit -> +1): a matching bit
    // contributes +64, a differing one -64, and with C = 0 an accumulator is the similarity  sim = 64 (256 - 2 d) = 16384 - 128 d  of (row, query) - an
    // integer of magnitude <= 2^14, exact in f32.  (Round 5 started the accumulators at 16384 + row so that they WERE the keys d << 7 | row: sixteen
    // registers of constants that the second accumulator set below has no room for.  The keys are now made from the similarities only where a tile is kept.)
    nn_v8i B[QT][4];
    int qidx[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        qidx[t] = blockIdx.x * QG + (wave * QT + t) * 32 + j;
        const unsigned* qp = q + (long long)min(qidx[t], nq - 1) * 8;
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const unsigned w = qp[2 * kb + h];
            B[t][kb] = nn_v8i{(int)s_tab[w & 0xff], (int)s_tab[(w >> 8) & 0xff], (int)s_tab[(w >> 16) & 0xff], (int)s_tab[w >> 24], 0, 0, 0, 0};
        }
    }
    // thr: a tile matters to a query only if it holds a similarity ABOVE it, i.e. a distance strictly below the running (or seeded) second best
    unsigned kbest[QT], ksec[QT]; float thr[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        kbest[t] = 0xffffffffu; ksec[t] = 0xffffffffu; thr[t] = -3.0e38f;
        if (seed) { const int sd2 = seed[min(qidx[t], nq - 1)]; if (sd2 >= 0 && sd2 <= 256) thr[t] = (float)(16384 - 128 * sd2); }
    }
    constexpr int CH = 1 << LCH;                                      // DB rows per workgroup; chunk keys are distance << LCH | row-in-chunk
    const long long row0 = (long long)chunk * CH;
    const int nrows = (int)min((long long)CH, ndb - row0);
    const int ntiles = (nrows + 31) >> 5;
    const int sr = tid & 31, sd = tid >> 5;                            // staging role: row sr of the tile, dword sd of that row
    // the next TPB tiles: loads issued first (they fly while the current tiles are worked on), expanded into LDS just before the barrier
    unsigned wnext[TPB];
    auto fetch = [&](int sup) {
#pragma unroll
        for (int u = 0; u < TPB; u++) { const int r = (sup * TPB + u) * 32 + sr; wnext[u] = r < nrows ? db[(row0 + r) * 8 + sd] : 0u; }
    };
    auto expand = [&](int buf) {
#pragma unroll
        for (int u = 0; u < TPB; u++) {
            const unsigned w = wnext[u];
            *reinterpret_cast<uint4*>(s_a[buf] + u * 1024 + (sd * 32 + sr) * 4) = uint4{s_tab[w & 0xff], s_tab[(w >> 8) & 0xff], s_tab[(w >> 16) & 0xff], s_tab[w >> 24]};
        }
    };
    // EXP: the superstep's tiles by LDS-DMA, thread tid bytes [16 tid, 16 tid + 16) of each 4 KB tile (wave w: the tile's w-th KB)
    auto stage_dma = [&](int sup, int buf) {
#pragma unroll
        for (int u = 0; u < TPB; u++) {
            const int tile = sup * TPB + u;
            if (tile < ntiles)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + ((row0 >> 5) + tile) * 1024 + tid * 4),
                                                 (__attribute__((address_space(3))) void*)(s_a[buf] + u * 1024 + wave * 256), 16, 0, 0);
        }
    };
    const nn_v16f czero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float off_h = (float)(4 * h);
    auto top2_of3 = [](unsigned a, unsigned b, unsigned c, unsigned& lo, unsigned& mid) { lo = min(min(a, b), c); mid = max(min(a, b), min(max(a, b), c)); };
    auto merge2 = [](unsigned& b, unsigned& s2, unsigned ob, unsigned os) { s2 = min(min(s2, os), max(b, ob)); b = min(b, ob); };
    auto products = [&](const unsigned* ta, nn_v16f (&acc)[QT]) {
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(ta + ((2 * kb + h) * 32 + j) * 4);               // lane (i = j, h): row i of the tile
            const nn_v8i A = nn_v8i{(int)a4.x, (int)a4.y, (int)a4.z, (int)a4.w, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < QT; t++)      // (A, B, C, format of A = FP4, format of B = FP4, scale A: byte 0 of 127 = 2^0, scale B: byte 0 of 133 = 2^6)
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B[t][kb], kb == 0 ? czero : acc[t], 4, 4, 0, 127, 0, 133);
        }
    };
    // the threshold test of one query tile: does ANY lane see a similarity above its threshold in this tile?  (eight v_max3_f32, a compare, a ballot)
    auto test = [&](int t, const nn_v16f (&a)[QT]) -> bool {
        const nn_v16f& x = a[t];
        const float m0 = fmaxf(fmaxf(x[0], x[1]), x[2]), m1 = fmaxf(fmaxf(x[3], x[4]), x[5]), m2 = fmaxf(fmaxf(x[6], x[7]), x[8]), m3 = fmaxf(fmaxf(x[9], x[10]), x[11]), m4 = fmaxf(fmaxf(x[12], x[13]), x[14]);
        const float mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), x[15]));
        return __ballot(mx > thr[t]) != 0;
    };
    // a kept (tile, query tile): keys d << 7 | row-in-tile = 16384 + row - sim as floats (positive floats order like their bits: the v_min3 / v_med3 tournament
    // runs on the raw registers, only the two winners are converted), the two smallest merged into the chunk's running pair; `ragged`: rows past the end lose
    auto tournament = [&](int tile, int t, const nn_v16f (&a)[QT], bool ragged) {
        const unsigned tbase = (unsigned)tile * 32u;
        unsigned x[16];
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int rbase = (reg & 3) + 8 * (reg >> 2);
            x[reg] = __float_as_uint(((float)(16384 + rbase) - a[t][reg]) + off_h);
            if (ragged && tile * 32 + rbase + 4 * h >= nrows) x[reg] = __float_as_uint((float)((511 << 7) + rbase) + off_h);
        }
        unsigned b, s2;
        top2_of3(x[0], x[1], x[2], b, s2);
#pragma unroll
        for (int g = 1; g < 5; g++) { unsigned lo, mid; top2_of3(x[3 * g], x[3 * g + 1], x[3 * g + 2], lo, mid); merge2(b, s2, lo, mid); }
        merge2(b, s2, x[15], 0x7f7fffffu);                                                                  // (the largest finite float's bits: loses to every key)
        const unsigned bi = (unsigned)__uint_as_float(b), si = (unsigned)__uint_as_float(s2);               // the two winners back to integers: d << 7 | r
        const unsigned kb1 = ((bi >> 7) << LCH) + (bi & 127u) + tbase, ks1 = ((si >> 7) << LCH) + (si & 127u) + tbase;      // chunk keys d << LCH | (tile * 32 + r)
        merge2(kbest[t], ksec[t], kb1, ks1);
        // a later row at the running second best's own distance has a larger key than it (rows ascend): only a strictly smaller distance matters.
        // (a never-set second best gives a threshold below every similarity; a seeded threshold is never lowered)
        thr[t] = fmaxf(thr[t], 16384.0f - 128.0f * (float)(ksec[t] >> LCH));
    };
    if constexpr (EXP) { stage_dma(0, 0); __builtin_amdgcn_s_waitcnt(0x0f70); }      // (vmcnt(0))
    else { fetch(0); expand(0); }
    __syncthreads();
    const int nfull = nrows >> 5, nsuper = (ntiles + TPB - 1) / TPB;
    // ---- the tile loop.  A wave issues in order: with ONE accumulator set it runs the sixteen matrix instructions of a tile (512 cycles of the matrix pipe),
    // then the threshold tests on their results - each pipe idle while the other works (0.44 of the matrix rate in round 5, two waves per SIMD).
    // PIPELINED form (QT = 4, TPB a multiple of 3; round 6): the matrix instructions of tile i + 1 and the tests of tile i form one straight-line block, so the
    // tests issue in the shadow of the matrix pipe.  A second full accumulator set does not fit beside the query operands (2 x 64 + 64 registers + the
    // rest > 256 at two waves per SIMD: the first build spilled 260 loads per tile).  So the tile is handled as two HALVES of two query tiles and there are
    // THREE register pairs: while the tests read the current tile's first half, the next tile's first half accumulates into the spare pair; the pair the
    // tests just released takes the next tile's second half while the tests read the current second half.  The roles rotate with period three tiles,
    // which is why TPB must be a multiple of 3 (all indices static after unrolling).  Inside a half the matrix instructions alternate between its two
    // accumulators (consecutive instructions never share one: a filler between two instructions on the SAME accumulator costs ~40 cycles, MI355X_MICROARCH.md).
    if constexpr (QT == 4 && TPB % 3 == 0) {
        nn_v16f acc[6];
        uint4 a4[4];
        auto load_a = [&](const unsigned* ta) {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
            if constexpr (EXP) {
                // spelled: while the NEXT superstep's LDS-DMA is in flight the compiler would put `s_waitcnt vmcnt(0)` in front of every ordinary LDS read
                // (it has no address for the DMA's LDS side: orbhip_kernels_extract.hip, lds_read3_issue) - i.e. wait for the prefetch.  Lane (j, h) = lane l:
                // its four operands are 1 KB apart from byte 16 l of the tile on
                typedef int v4i_t __attribute__((vector_size(16)));
                v4i_t q0, q1, q2, q3;
                const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)ta + 16u * (unsigned)lane;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr) : "memory");
                a4[0] = uint4{(unsigned)q0[0], (unsigned)q0[1], (unsigned)q0[2], (unsigned)q0[3]}; a4[1] = uint4{(unsigned)q1[0], (unsigned)q1[1], (unsigned)q1[2], (unsigned)q1[3]};
                a4[2] = uint4{(unsigned)q2[0], (unsigned)q2[1], (unsigned)q2[2], (unsigned)q2[3]}; a4[3] = uint4{(unsigned)q3[0], (unsigned)q3[1], (unsigned)q3[2], (unsigned)q3[3]};
                return;
            }
#endif
#pragma unroll
            for (int kb = 0; kb < 4; kb++) a4[kb] = *reinterpret_cast<const uint4*>(ta + ((2 * kb + h) * 32 + j) * 4);
        };
        auto mm = [&](nn_v16f& d, int t, int kb) {
            const nn_v8i A = nn_v8i{(int)a4[kb].x, (int)a4[kb].y, (int)a4[kb].z, (int)a4[kb].w, 0, 0, 0, 0};
            d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B[t][kb], kb == 0 ? czero : d, 4, 4, 0, 127, 0, 133);
        };
        auto test1 = [&](int t, const nn_v16f& x) -> bool {
            const float m0 = fmaxf(fmaxf(x[0], x[1]), x[2]), m1 = fmaxf(fmaxf(x[3], x[4]), x[5]), m2 = fmaxf(fmaxf(x[6], x[7]), x[8]), m3 = fmaxf(fmaxf(x[9], x[10]), x[11]), m4 = fmaxf(fmaxf(x[12], x[13]), x[14]);
            const float mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), x[15]));
            return __ballot(mx > thr[t]) != 0;
        };
        auto tournament1 = [&](int tile, int t, const nn_v16f& a, bool ragged) {
            const unsigned tbase = (unsigned)tile * 32u;
            unsigned x[16];
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int rbase = (reg & 3) + 8 * (reg >> 2);
                x[reg] = __float_as_uint(((float)(16384 + rbase) - a[reg]) + off_h);
                if (ragged && tile * 32 + rbase + 4 * h >= nrows) x[reg] = __float_as_uint((float)((511 << 7) + rbase) + off_h);
            }
            unsigned b, s2;
            top2_of3(x[0], x[1], x[2], b, s2);
#pragma unroll
            for (int g = 1; g < 5; g++) { unsigned lo, mid; top2_of3(x[3 * g], x[3 * g + 1], x[3 * g + 2], lo, mid); merge2(b, s2, lo, mid); }
            merge2(b, s2, x[15], 0x7f7fffffu);
            const unsigned bi = (unsigned)__uint_as_float(b), si = (unsigned)__uint_as_float(s2);
            const unsigned kb1 = ((bi >> 7) << LCH) + (bi & 127u) + tbase, ks1 = ((si >> 7) << LCH) + (si & 127u) + tbase;
            merge2(kbest[t], ksec[t], kb1, ks1);
            thr[t] = fmaxf(thr[t], 16384.0f - 128.0f * (float)(ksec[t] >> LCH));
        };
        load_a(s_a[0]);                                                // tile 0: first half into pair 0, second half into pair 1
#pragma unroll
        for (int kb = 0; kb < 4; kb++) { mm(acc[0], 0, kb); mm(acc[1], 1, kb); }
#pragma unroll
        for (int kb = 0; kb < 4; kb++) { mm(acc[2], 2, kb); mm(acc[3], 3, kb); }
        for (int sup = 0; sup < nsuper; sup++) {
            const int buf = sup & 1;
            if (sup + 1 < nsuper && !(ablate & 2)) { if constexpr (EXP) stage_dma(sup + 1, buf ^ 1); else fetch(sup + 1); }      // (EXP: everybody left that buffer at the barrier of the superstep before)
#pragma unroll
            for (int u = 0; u < TPB; u++) {
                const int tile = sup * TPB + u;
                if (tile >= ntiles) break;
                const int cA = (3 - u % 3) % 3, cB = (cA + 1) % 3, sp = (cA + 2) % 3;             // static after unrolling: pair holding the current first half / second half / the spare
                nn_v16f &x0 = acc[2 * cA], &x1 = acc[2 * cA + 1], &x2 = acc[2 * cB], &x3 = acc[2 * cB + 1], &n0 = acc[2 * sp], &n1 = acc[2 * sp + 1];
                const bool last_of_super = u == TPB - 1;
                if (last_of_super && sup + 1 < nsuper && !(ablate & 2)) {                            // the next tile lives in the other buffer: it is filled, everybody has left this one
                    if constexpr (EXP) __builtin_amdgcn_s_waitcnt(0x0f70); else expand(buf ^ 1);
                    __syncthreads();
                }
                const bool have_next = tile + 1 < ntiles;
                const unsigned* tnext = last_of_super ? s_a[buf ^ 1] : s_a[buf] + (u + 1) * 1024;
                if (tile < nfull) {
                    unsigned keep = 0;
                    if (have_next && (ablate & 1)) {                // (measurement only, ORBHIP_NN_ABLATE: the matrix instructions alone - results are wrong)
                        load_a(tnext);
                        if (ablate & 4) {                            // each accumulator's four instructions back to back
                            mm(n0, 0, 0); mm(n0, 0, 1); mm(n0, 0, 2); mm(n0, 0, 3); mm(n1, 1, 0); mm(n1, 1, 1); mm(n1, 1, 2); mm(n1, 1, 3);
                            mm(x0, 2, 0); mm(x0, 2, 1); mm(x0, 2, 2); mm(x0, 2, 3); mm(x1, 3, 0); mm(x1, 3, 1); mm(x1, 3, 2); mm(x1, 3, 3);
                        } else {
                            mm(n0, 0, 0); mm(n1, 1, 0); mm(n0, 0, 1); mm(n1, 1, 1); mm(n0, 0, 2); mm(n1, 1, 2); mm(n0, 0, 3); mm(n1, 1, 3);
                            mm(x0, 2, 0); mm(x1, 3, 0); mm(x0, 2, 1); mm(x1, 3, 1); mm(x0, 2, 2); mm(x1, 3, 2); mm(x0, 2, 3); mm(x1, 3, 3);
                        }
                    } else if (have_next) {
                        // Two straight-line blocks of eight matrix instructions and two tests each.  Left to itself the compiler issues the matrix
                        // instructions of a block back to back and the tests behind them, and the two waves of a SIMD then fall into step - both queue
                        // on the matrix pipe, both test while it idles: the tests cost their full 0.65 ms of 3.9 (profiles/r06_exp_config5_ablation.txt).
                        // The group barriers spell the interleave out: one matrix instruction, three VALU instructions, eight times.
                        load_a(tnext);
                        mm(n0, 0, 0); mm(n1, 1, 0); mm(n0, 0, 1); mm(n1, 1, 1);
                        keep |= test1(0, x0) ? 1u : 0u;
                        mm(n0, 0, 2); mm(n1, 1, 2); mm(n0, 0, 3); mm(n1, 1, 3);
                        keep |= test1(1, x1) ? 2u : 0u;
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                        for (int i = 0; i < 8; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
                        if (keep & 1u) tournament1(tile, 0, x0, false);                             // (rare; the released pair is overwritten below)
                        if (keep & 2u) tournament1(tile, 1, x1, false);
                        mm(x0, 2, 0); mm(x1, 3, 0); mm(x0, 2, 1); mm(x1, 3, 1);
                        keep |= test1(2, x2) ? 4u : 0u;
                        mm(x0, 2, 2); mm(x1, 3, 2); mm(x0, 2, 3); mm(x1, 3, 3);
                        keep |= test1(3, x3) ? 8u : 0u;
#pragma unroll
                        for (int i = 0; i < 8; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x002, 3, 1); }
                        if (keep & 4u) tournament1(tile, 2, x2, false);
                        if (keep & 8u) tournament1(tile, 3, x3, false);
                    } else {
                        if (test1(0, x0)) tournament1(tile, 0, x0, false);
                        if (test1(1, x1)) tournament1(tile, 1, x1, false);
                        if (test1(2, x2)) tournament1(tile, 2, x2, false);
                        if (test1(3, x3)) tournament1(tile, 3, x3, false);
                    }
                } else {                                               // the chunk's last, ragged tile (nothing follows it): every query tile, rows past the end lose
                    tournament1(tile, 0, x0, true); tournament1(tile, 1, x1, true); tournament1(tile, 2, x2, true); tournament1(tile, 3, x3, true);
                }
            }
        }
    } else {
        // one accumulator set: matrix instructions, then tests, then the kept tournaments (the experimental shapes of ORBHIP_NN=fp4:...)
        nn_v16f acc[QT];
        for (int sup = 0; sup < nsuper; sup++) {
            const int buf = sup & 1;
            if (sup + 1 < nsuper) fetch(sup + 1);
#pragma unroll
            for (int u = 0; u < TPB; u++) {
                const int tile = sup * TPB + u;
                if (tile >= ntiles) break;
                products(s_a[buf] + u * 1024, acc);
#pragma unroll
                for (int t = 0; t < QT; t++) if (tile >= nfull || test(t, acc)) tournament(tile, t, acc, tile >= nfull);
            }
            if (sup + 1 < nsuper) expand(buf ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < QT; t++) {
        const unsigned ob = (unsigned)__shfl_xor((int)kbest[t], 32), os = (unsigned)__shfl_xor((int)ksec[t], 32);
        const unsigned b = min(kbest[t], ob), s2 = min(min(ksec[t], os), max(kbest[t], ob));
        if (h == 0 && qidx[t] < nq) {
            NNPart p;
            p.best = (b >> LCH) > 256u ? IMAX : (int)(b >> LCH);
            p.second = (s2 >> LCH) > 256u ? IMAX : (int)(s2 >> LCH);
            p.idx = (b >> LCH) > 256u ? -1 : row0 + (long long)(b & (unsigned)(CH - 1)) + base;
            parts[(long long)qidx[t] * nchunks + part0 + (int)blockIdx.y] = p;
        }
    }
}
// ---- the FP4 scan with the superstep's instruction order assigned by hand (k_hamming_nn_fp4b; the production form of the seeded scan).
// k_hamming_nn_fp4 above is everything source order, scheduling barriers and priorities could get out of hipcc: the sixteen matrix instructions of a tile
// come out back to back with the threshold tests and the tile's operand reads (+ their lgkmcnt(0)) in front of them, so a wave's matrix pipe idles while
// it tests and waits (4.5 ms; this kernel: 3.7 on the expanded database, 4.2 on the bit form - profiles/r06_exp_config5_superstep.txt).  Here a whole superstep of NN_FP4B_TPB tiles is ONE asm statement whose text is
// generated (tools/gen_nn_fp4_block.py -> nn_fp4_block.inc): matrix instruction, two or three v_max3_f32 of a tile finished long before, matrix
// instruction, ...; the next tile's operands are read a half tile ahead.  The statement owns its accumulators (registers it clobbers), so nothing of a tile
// outlives it except ONE scalar: bit 8 t + u = tile u may matter to query tile t.  Those rare pairs are recomputed - four matrix instructions - and folded
// in by compiled code behind the statement, while the superstep's tiles are still in LDS.  A threshold is therefore up to one superstep stale: it only ever
// keeps more, never fewer.  Partial supersteps and the ragged tile take the compiled per-tile path.  hipcc must NOT spill across the statement: a reload in
// front of it comes with `s_waitcnt vmcnt(0)`, i.e. waits for the prefetch issued just before - hence one tile-operand set, the lane's LDS address and the
// scales made inside the statement, and addresses rebuilt at their (rare) uses instead of kept (check: no scratch_ access between the loop's barriers).
#include "nn_fp4_block.inc"
#define NN_SHARE_EVERY 16                // supersteps between two reads of the shared bounds (a power of two)
// SHARED BOUNDS.  `seed[q]` (nullptr: none) = the head's second-best distance: the head's rows precede every other row, so a later row AT that distance loses
// the tie on the index and only a strictly smaller distance matters (as in k_hamming_nn_fp4).  `share` (nullptr: none) = two words per query: share[q] the
// smallest, share[nq + q] the second smallest distance among the head's best pair (k_hamming_seed) and EVERY row any workgroup has found below its threshold
// since.  A row at distance d is OFFERED with two non-returning atomics, atomicMin(best, d) and atomicMin(second, max(d, b)), b = the best as last read by
// the offering lane: b is the distance of some OTHER row, so max(d, b) is at least the second smallest of two real rows - `second` never falls below the final
// second-best distance.  (The exact exchange `old = atomicMin(best, d); atomicMin(second, max(old, d))` was built first: its returned value is a round trip of
// microseconds in front of the workgroup's barrier and cost more than the bounds won.)  A tile whose distances all EXCEED some second best S holds neither
// the final best, nor the final second best, nor a row tied with either - whichever rows S came from - so S + 1 is a threshold for everybody.  A workgroup
// re-reads the pair of its queries every NN_SHARE_EVERY-th superstep by LDS-DMA with sc1 (device scope: a plain load is served from the reading XCD's L2, which
// the other XCDs' atomics never reach), requested at a superstep's start and taken behind its barrier: the thresholds follow the best pair found ANYWHERE.
// Under the head's bound alone one (tile, query tile) in twenty-three is kept and recomputed, with the shared bounds one in five hundred
// (ORBHIP_NN_STATS=1; ORBHIP_NN_SHARE=0: without).  The filter only decides which tiles are looked at: the answers do not depend on the order the workgroups
// run in (tests/test_parity_match.py: test_brute_force_nn_ties_across_chunks).
// NW = wavefronts per workgroup: 4 (two workgroups per CU) or 8 (one: the same staged tiles serve 1024 queries instead of 512 - half the LDS-DMA requests and
// bytes per matrix instruction; each half of the workgroup stages every other tile of a superstep).  The superstep's text does not depend on it.
template <int LCH, bool EXP, int VAR = 0, int NW = 4> __global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_hamming_nn_fp4b(const unsigned* q, int nq, const unsigned* db, long long ndb, long long base, NNPart* parts, int nchunks,
                                                                                                        const int* seed, int* share, long long rows0, int chrows, int part0, int* stats, int share_mask)
{
    constexpr int QT = 4, TPB = NN_FP4B_TPB;
    static_assert(TPB <= 8, "the keep mask has eight bits per query tile");
    __shared__ unsigned s_tab[256];
    __shared__ __attribute__((aligned(16))) unsigned s_a[2][TPB * 1024];
    static_assert(NW == 4 || NW == 8, "four or eight wavefronts");
    constexpr int NH = NW / 4;                                         // groups of 256 threads: group g stages tiles g, g + NH, ... of a superstep
    static_assert(TPB % NH == 0, "every group stages the same number of tiles");
    __shared__ int s_bnd[NW][2 * QT][64];                              // the queries' shared pairs as last read: [wave][t] the second best, [wave][QT + t] the best
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y;
    if (tid < 256) {
        unsigned e = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) e |= (((tid >> t) & 1) ? 0x2u : 0xAu) << (4 * t);
        s_tab[tid] = e;
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    constexpr int QG = NW * QT * 32;
    nn_v4i B[QT][4];                                                   // (operand layout, signs and scales: k_hamming_nn_fp4)
    int qidx[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        qidx[t] = bx * QG + (wave * QT + t) * 32 + j;
        const unsigned* qp = q + (long long)min(qidx[t], nq - 1) * 8;
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const unsigned w = qp[2 * kb + h];
            B[t][kb] = nn_v4i{(int)s_tab[w & 0xff], (int)s_tab[(w >> 8) & 0xff], (int)s_tab[(w >> 16) & 0xff], (int)s_tab[w >> 24]};
        }
    }
    unsigned kbest[QT], ksec[QT]; float thr[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        kbest[t] = 0xffffffffu; ksec[t] = 0xffffffffu; thr[t] = -3.0e38f;
        if (seed) { const int sd2 = seed[min(qidx[t], nq - 1)]; if (sd2 >= 0 && sd2 <= 256) thr[t] = (float)(16384 - 128 * sd2); }
    }
    int* const bound = share;
    auto bounds_request = [&]() {      // (aux 16 = sc1: device-scope reads - a plain one is served from this XCD's L2, which the other XCDs' atomics never reach)
#pragma unroll
        for (int t = 0; t < QT; t++) {
            int qc = min(qidx[t], nq - 1);
            asm volatile("" : "+v"(qc));                               // (rebuilt at every request: as eight loop-invariant pointers the addresses were spilled around the superstep)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(share + nq + qc), (__attribute__((address_space(3))) void*)&s_bnd[wave][t][0], 4, 0, 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(share + qc), (__attribute__((address_space(3))) void*)&s_bnd[wave][QT + t][0], 4, 0, 16);
        }
    };
    auto bounds_take = [&]() {
#pragma unroll
        for (int t = 0; t < QT; t++) { const int sd2 = s_bnd[wave][t][lane]; if (sd2 >= 0 && sd2 <= 256) thr[t] = fmaxf(thr[t], (float)(16384 - 128 * (sd2 + 1))); }
    };
    constexpr int CH = 1 << LCH;
    // rows [rows0 + by * chrows, + chrows) are this workgroup's: chrows <= CH (the keys pack a row-in-chunk into LCH bits) and a multiple of 32 (tiles of the
    // expanded database); the launcher sizes it so that the launch fills the chip's workgroup slots a whole number of times (below)
    const long long row0 = rows0 + (long long)by * chrows;
    const int nrows = (int)min((long long)chrows, ndb - row0);
    const int ntiles = (nrows + 31) >> 5;
    const int sr = tid & 31, sd = (tid >> 5) & 7, grp = NH == 1 ? 0 : (wave >> 2), t256 = tid & 255;
    unsigned wnext[TPB / NH];
    auto fetch = [&](int sup) {
#pragma unroll
        for (int u2 = 0; u2 < TPB / NH; u2++) { const int r = (sup * TPB + u2 * NH + grp) * 32 + sr; wnext[u2] = r < nrows ? db[(row0 + r) * 8 + sd] : 0u; }
    };
    auto expand = [&](int buf) {
#pragma unroll
        for (int u2 = 0; u2 < TPB / NH; u2++) {
            const unsigned w = wnext[u2];
            *reinterpret_cast<uint4*>(s_a[buf] + (u2 * NH + grp) * 1024 + (sd * 32 + sr) * 4) = uint4{s_tab[w & 0xff], s_tab[(w >> 8) & 0xff], s_tab[(w >> 16) & 0xff], s_tab[w >> 24]};
        }
    };
    auto stage_dma = [&](int sup, int buf) {
#pragma unroll
        for (int u2 = 0; u2 < TPB / NH; u2++) {
            const int u = u2 * NH + grp, tile = sup * TPB + u;
            if (tile < ntiles)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + ((row0 >> 5) + tile) * 1024 + t256 * 4),
                                                 (__attribute__((address_space(3))) void*)(s_a[buf] + u * 1024 + (wave & 3) * 256), 16, 0, 0);
        }
    };
    const nn_v16f czero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float off_h = (float)(4 * h);
    auto top2_of3 = [](unsigned a, unsigned b, unsigned c, unsigned& lo, unsigned& mid) { lo = min(min(a, b), c); mid = max(min(a, b), min(max(a, b), c)); };
    auto merge2 = [](unsigned& b, unsigned& s2, unsigned ob, unsigned os) { s2 = min(min(s2, os), max(b, ob)); b = min(b, ob); };
    // the similarities of one (tile, query tile): four matrix instructions (compiled: the kept pairs and the per-tile path)
    auto products1 = [&](const unsigned* ta, const nn_v4i (&b)[4]) -> nn_v16f {
        nn_v16f d = czero;
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(ta + ((2 * kb + h) * 32 + j) * 4);
            d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(nn_v8i{(int)a4.x, (int)a4.y, (int)a4.z, (int)a4.w, 0, 0, 0, 0}, nn_v8i{b[kb][0], b[kb][1], b[kb][2], b[kb][3], 0, 0, 0, 0}, d, 4, 4, 0, 127, 0, 133);
        }
        return d;
    };
    auto test1 = [&](int t, const nn_v16f& x) -> bool {
        const float m0 = fmaxf(fmaxf(x[0], x[1]), x[2]), m1 = fmaxf(fmaxf(x[3], x[4]), x[5]), m2 = fmaxf(fmaxf(x[6], x[7]), x[8]), m3 = fmaxf(fmaxf(x[9], x[10]), x[11]), m4 = fmaxf(fmaxf(x[12], x[13]), x[14]);
        const float mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), x[15]));
        return __ballot(mx > thr[t]) != 0;
    };
    auto tournament1 = [&](int tile, int t, const nn_v16f& a, bool ragged) {       // (keys, tournament, threshold: k_hamming_nn_fp4)
        const unsigned tbase = (unsigned)tile * 32u;
        unsigned x[16];
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int rbase = (reg & 3) + 8 * (reg >> 2);
            x[reg] = __float_as_uint(((float)(16384 + rbase) - a[reg]) + off_h);
            if (ragged && tile * 32 + rbase + 4 * h >= nrows) x[reg] = __float_as_uint((float)((511 << 7) + rbase) + off_h);
        }
        unsigned b, s2;
        top2_of3(x[0], x[1], x[2], b, s2);
#pragma unroll
        for (int g = 1; g < 5; g++) { unsigned lo, mid; top2_of3(x[3 * g], x[3 * g + 1], x[3 * g + 2], lo, mid); merge2(b, s2, lo, mid); }
        merge2(b, s2, x[15], 0x7f7fffffu);
        const unsigned bi = (unsigned)__uint_as_float(b), si = (unsigned)__uint_as_float(s2);
        const unsigned kb1 = ((bi >> 7) << LCH) + (bi & 127u) + tbase, ks1 = ((si >> 7) << LCH) + (si & 127u) + tbase;
        if (share && qidx[t] < nq) {
            // this lane's two best rows of the tile, offered if they beat its threshold.  No atomic RETURNS anything (a returned value is a round trip of
            // microseconds in front of the workgroup's barrier: the first build, with `old = atomicMin(best, d); atomicMin(second, max(old, d))`, lost more there
            // than the bounds won): the loser of the exchange with the best is taken to be max(d, b) for b = the best AS LAST READ (or the tile's other row) - the
            // distance of some OTHER row, so max(d, b) is at least the second smallest of two real rows: a valid, at worst slightly loose, offer to the second best
            const int d1 = (int)(bi >> 7), d2 = (int)(si >> 7), bstar = s_bnd[wave][QT + t][lane];
            int qc = qidx[t];
            asm volatile("" : "+v"(qc));
            if (d1 <= 256 && (float)(16384 - 128 * d1) > thr[t]) { atomicMin(share + qc, d1); atomicMin(share + nq + qc, max(d1, bstar)); }
            if (d2 <= 256 && (float)(16384 - 128 * d2) > thr[t]) atomicMin(share + nq + qc, d2);                 // (the tile's other row is no farther)
        }
        merge2(kbest[t], ksec[t], kb1, ks1);
        thr[t] = fmaxf(thr[t], 16384.0f - 128.0f * (float)(ksec[t] >> LCH));
    };
    // one whole superstep: which (tile u, query tile t) hold a similarity above the query tile's threshold - bit 8 t + u
    auto superstep = [&](const unsigned* ta) -> unsigned {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
        unsigned keep, stmp;
        const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)ta);
#define NN_FP4B_ASM(BODY) asm volatile(BODY \
                     : "=&s"(keep), "=&s"(stmp) \
                     : "v"(B[0][0]), "v"(B[0][1]), "v"(B[0][2]), "v"(B[0][3]), "v"(B[1][0]), "v"(B[1][1]), "v"(B[1][2]), "v"(B[1][3]), \
                       "v"(B[2][0]), "v"(B[2][1]), "v"(B[2][2]), "v"(B[2][3]), "v"(B[3][0]), "v"(B[3][1]), "v"(B[3][2]), "v"(B[3][3]), \
                       "v"(thr[0]), "v"(thr[1]), "v"(thr[2]), "v"(thr[3]), "s"(addr) \
                     : "memory", "vcc", "scc", NN_FP4B_CLOBBERS)
        // (VAR != 0: measurement only - ORBHIP_NN_BLOCK_VAR, tools/gen_nn_fp4_block.py: the same superstep with parts of its text left out; wrong answers)
        if constexpr ((VAR & 3) == 1) NN_FP4B_ASM(NN_FP4B_BODY1); else if constexpr ((VAR & 3) == 3) NN_FP4B_ASM(NN_FP4B_BODY3); else NN_FP4B_ASM(NN_FP4B_BODY);
#undef NN_FP4B_ASM
        return keep;
#else
        unsigned keep = 0;                                            // (the CPU emulation of the test suite: the same mask from the compiled pieces)
        for (int u = 0; u < TPB; u++)
            for (int t = 0; t < QT; t++) if (test1(t, products1(ta + u * 1024, B[t]))) keep |= 1u << (8 * t + u);
        return keep;
#endif
    };
    if (bound) bounds_request();                                       // what the others have found so far, with the first tiles
    if constexpr (EXP) stage_dma(0, 0); else { fetch(0); expand(0); }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                // (vmcnt(0))
    __syncthreads();
    if (bound) bounds_take();
    const int nfull = nrows >> 5, nsuper = (ntiles + TPB - 1) / TPB;
    int nkept = 0;                                                     // (ORBHIP_NN_STATS: kept (tile, query tile) pairs of this wave)
    bool pending = false;
    for (int sup = 0; sup < nsuper; sup++) {
        const int buf = sup & 1, tile0 = sup * TPB;
        if (pending) bounds_take();                                                                         // (requested at the start of the superstep before, landed before its barrier)
        pending = false;
        if constexpr (VAR >= 4) {                                                                           // (measurement only: no staging of new tiles, no barrier)
            const unsigned keep = superstep(s_a[0]);
            if (keep == 0x12345678u) tournament1(tile0, 0, products1(s_a[0], B[0]), false);
            continue;
        }
        if (sup + 1 < nsuper) { if constexpr (EXP) stage_dma(sup + 1, buf ^ 1); else fetch(sup + 1); }      // (everybody left that buffer at the barrier of the superstep before)
        // every NN_SHARE_EVERY-th superstep, the workgroups out of step with each other: the reads are device-scope (they go past the L2) and all workgroups of a
        // query group read the same few lines - once per superstep they queued on those lines' memory channel (5.1 ms against 3.3 for the scan without them)
        if (bound && sup + 1 < nsuper && ((sup + by) & share_mask) == 0) { bounds_request(); pending = true; }
        if (tile0 + TPB <= nfull) {                                                                         // a superstep of whole tiles: the hand-ordered statement
            const unsigned keep = superstep(s_a[buf]);
            nkept += __builtin_popcount(keep);
#pragma unroll
            for (int t = 0; t < QT; t++) {
                unsigned m = (keep >> (8 * t)) & 0xffu;
                while (m) {
                    const int u = __builtin_ctz(m); m &= m - 1;
                    tournament1(tile0 + u, t, products1(s_a[buf] + u * 1024, B[t]), false);
                }
            }
        } else {
            for (int u = 0; u < TPB; u++) {
                const int tile = tile0 + u;
                if (tile >= ntiles) break;
#pragma unroll
                for (int t = 0; t < QT; t++) {
                    const nn_v16f d = products1(s_a[buf] + u * 1024, B[t]);
                    if (tile >= nfull || test1(t, d)) tournament1(tile, t, d, tile >= nfull);
                }
            }
        }
        if (sup + 1 < nsuper) { if constexpr (!EXP) expand(buf ^ 1); __builtin_amdgcn_s_waitcnt(0x0f70); }      // (vmcnt(0): the next tiles and the bounds have landed)
        __syncthreads();
    }
    if (stats && lane == 0) { atomicAdd(stats, nkept); atomicAdd(stats + 1, QT * min(ntiles, nfull)); }
#pragma unroll
    for (int t = 0; t < QT; t++) {
        const unsigned ob = (unsigned)__shfl_xor((int)kbest[t], 32), os = (unsigned)__shfl_xor((int)ksec[t], 32);
        const unsigned b = min(kbest[t], ob), s2 = min(min(ksec[t], os), max(kbest[t], ob));
        if (h == 0 && qidx[t] < nq) {
            NNPart p;
            p.best = (b >> LCH) > 256u ? IMAX : (int)(b >> LCH);
            p.second = (s2 >> LCH) > 256u ? IMAX : (int)(s2 >> LCH);
            p.idx = (b >> LCH) > 256u ? -1 : row0 + (long long)(b & (unsigned)(CH - 1)) + base;
            parts[(long long)qidx[t] * nchunks + part0 + by] = p;
        }
    }
}
// The database as the FP4 scan reads it (orbhip_nn_expand_device): tile T = rows 32 T .. 32 T + 31 as 4 KB, [dword d of the row][row i] x 16 bytes = the eight
// E2M1 nibbles of each of the dword's four bytes - byte for byte what k_hamming_nn_fp4's `expand` writes into LDS.  Rows past the end: zeros (never a winner:
// the scan's ragged tile masks them).  One thread per (tile, d, i).
__global__ __launch_bounds__(256) void k_nn_expand(const unsigned* db, long long ndb, uint4* out, long long ntiles)
{
    __shared__ unsigned s_tab[256];
    {
        unsigned e = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) e |= (((threadIdx.x >> t) & 1) ? 0x2u : 0xAu) << (4 * t);
        s_tab[threadIdx.x] = e;
    }
    __syncthreads();
    const long long tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int i = threadIdx.x & 31, d = threadIdx.x >> 5;
    const long long row = tile * 32 + i;
    const unsigned w = row < ndb ? db[row * 8 + d] : 0u;
    out[tile * 256 + d * 32 + i] = row < ndb ? uint4{s_tab[w & 0xff], s_tab[(w >> 8) & 0xff], s_tab[(w >> 16) & 0xff], s_tab[w >> 24]} : uint4{0u, 0u, 0u, 0u};
}
size_t orbhip_nn_expanded_bytes(long long ndb) { return (size_t)((ndb + 31) / 32) * 4096; }
void orbhip_launch_nn_expand(const uint8_t* d_db, long long ndb, uint8_t* d_out, hipStream_t s)
{
    const long long ntiles = (ndb + 31) / 32;
    if (ntiles > 0) hipLaunchKernelGGL(k_nn_expand, dim3((unsigned)ntiles, 1, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_db, ndb, (uint4*)d_out, ntiles);
}

// second-best distance over the head's partials of every query: the seed of the main pass (a head with fewer than two rows in reach gives none)
__global__ __launch_bounds__(256) void k_hamming_seed(const NNPart* parts, int nq, int stride, int nhead, int* seed, int* share)
{   // share (nullptr: none): the shared best / second-best pair of k_hamming_nn_fp4b starts as the head's.
    // One wavefront per query, a lane per head partial (one thread per query walked its 64 partials one load latency after the other: 30 us of a 3.7 ms query)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    int b = IMAX, s2 = IMAX;
    for (int c = lane; c < nhead; c += 64) { const NNPart p = parts[(long long)qi * stride + c]; s2 = min(min(s2, p.second), max(b, p.best)); b = min(b, p.best); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {                 // the two smallest of the union: only the distances matter here, not which row
        const int ob = __shfl_xor(b, off), os = __shfl_xor(s2, off);
        s2 = min(min(s2, os), max(b, ob)); b = min(b, ob);
    }
    if (lane == 0) {
        seed[qi] = s2 == IMAX ? -1 : s2;
        if (share) { share[qi] = b == IMAX ? (1 << 20) : b; share[nq + qi] = s2 == IMAX ? (1 << 20) : s2; }      // (none: above every distance, and atomicMin can still lower it)
    }
}

// fold the per-chunk partials of one query in ascending DB order: stable arg-min + second smallest of the multiset
__device__ __forceinline__ void nn_combine(int& b, long long& i, int& s, int rb, long long ri, int rs)
{
    if (rb < b) { s = min(b, rs); b = rb; i = ri; } else { s = min(s, rb); }
}
__global__ __launch_bounds__(256) void k_hamming_merge(const NNPart* parts, int nq, int nchunks, long long* best_idx, int* best_dist, int* second_dist)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    int b = IMAX, s = IMAX; long long i = -1;
    const int per = (nchunks + 63) / 64;                     // contiguous runs per lane keep the index order
    for (int c = lane * per; c < min((lane + 1) * per, nchunks); c++) { const NNPart p = parts[(long long)qi * nchunks + c]; nn_combine(b, i, s, p.best, p.idx, p.second); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {                 // lanes hold ascending index ranges; combine neighbours left-to-right
        const int rb = __shfl_down(b, off), rs = __shfl_down(s, off); const long long ri = __shfl_down(i, off);
        if (((lane & (2 * off - 1)) == 0) && lane + off < 64) nn_combine(b, i, s, rb, ri, rs);
    }
    if (lane == 0) { best_idx[qi] = i; best_dist[qi] = b; second_dist[qi] = s; }
}

// workspace for the partials lives in a small per-thread cache owned by the API layer

bool orbhip_launch_hamming_nn(const uint8_t* d_q, int nq, const uint8_t* d_db, long long ndb, long long base,
                              long long* d_best_idx, int* d_best_dist, int* d_second, hipStream_t s, const uint8_t* d_dbx)
{   // d_dbx: the same database expanded by orbhip_launch_nn_expand (nullptr: none) - taken by the default FP4 shape's seeded scan
    if (nq <= 0) return true;
    const int nchunks = (int)max(1LL, (ndb + NN_CHUNK - 1) / NN_CHUNK);
    NNPart* parts = (NNPart*)orbhip_nn_workspace(sizeof(NNPart) * (size_t)nq * nchunks, s);
    if (!parts) return false;                                    // the caller reports it: results would be left unwritten
    // the matrix-core scan from a few chunks on (below that a call is latency, not throughput); ORBHIP_NN=valu (measurement only) keeps the popcount kernel
    // ORBHIP_NN = valu | i8 | fp4[:<query tiles 2-4>[:<workgroups per CU 2-3>[:<log2 rows per workgroup 13|15|16>]]] (measurement only): the popcount kernel, the i8
    // matrix-core scan, the FP4 one in the given shape
    int form = ORBHIP_NN_DEFAULT, qt = ORBHIP_NN_FP4_QT, occ = ORBHIP_NN_FP4_OCC, lch = ORBHIP_NN_FP4_LCH, tpb = ORBHIP_NN_FP4_TPB;
    if (const char* e = getenv("ORBHIP_NN")) {      // (read per call: a scan is at least a hundred microseconds; tests switch forms inside one process)
        if (!strcmp(e, "valu")) form = 0; else if (!strcmp(e, "i8")) form = 1;
        else if (!strncmp(e, "fp4", 3)) { form = 2; int a = 0, b2 = 0, c = 0, d = 0; const int k = sscanf(e + 3, ":%d:%d:%d:%d", &a, &b2, &c, &d); if (k >= 1) qt = a; if (k >= 2) occ = b2; if (k >= 3) lch = c; if (k >= 4) tpb = d; }
        // a shape that was not compiled (the switch below lists them; all have >= 2^13 rows per workgroup, so their partials fit the workspace sized above):
        // the default shape scans instead - said once on stderr - rather than a failure the callers could only report as "no device memory"
        static const int known[] = {22131, 32131, 42131, 42151, 42152, 42154, 42158, 23154, 23151, 22154, 42134, 42164, 32154, 42156, 42153};
        bool listed = false;
        for (int v : known) listed = listed || v == qt * 10000 + occ * 1000 + lch * 10 + tpb;
        if (form == 2 && !listed) {
            static bool said = false;
            if (!said) { said = true; fprintf(stderr, "orbhip: ORBHIP_NN=%s names a scan shape that is not built; the default fp4:%d:%d:%d:%d is used\n", e, ORBHIP_NN_FP4_QT, ORBHIP_NN_FP4_OCC, ORBHIP_NN_FP4_LCH, ORBHIP_NN_FP4_TPB); }
            qt = ORBHIP_NN_FP4_QT; occ = ORBHIP_NN_FP4_OCC; lch = ORBHIP_NN_FP4_LCH; tpb = ORBHIP_NN_FP4_TPB;
        }
    }
    if (form >= 2 && ndb >= 4 * NN_CHUNK) {
        const int qg = 4 * qt * 32, nch = (int)((ndb + ((long long)1 << lch) - 1) >> lch);      // (<= nchunks: the partials fit the workspace)
        bool ok = true;
#define NN_FP4(QT, OCC, LCH, TPB, GY, SEED, C0, P0, STRIDE) hipLaunchKernelGGL((k_hamming_nn_fp4<QT, OCC, LCH, TPB>), dim3((nq + qg - 1) / qg, GY, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)d_db, ndb, base, parts, STRIDE, SEED, C0, P0)
        const int shape = qt * 10000 + occ * 1000 + lch * 10 + tpb;
        // The seeded two-pass form for the default shape on databases of more than a few chunks (ORBHIP_NN_SEED=0: one pass, as in round 5): the head = the
        // first 2^15 rows as 64 sub-chunks of 512 rows (256 workgroups, sixteen tiles each - a chunk of 2^15 rows takes ONE workgroup a millisecond), their
        // merged second-best distance per query is the bound, the rest of the database is scanned under it.  Partials: [64 head sub-chunks][chunks 1 .. nch - 1].
        const int ablate = getenv("ORBHIP_NN_ABLATE") ? atoi(getenv("ORBHIP_NN_ABLATE")) : 0;      // measurement only: 1 = no threshold tests, 2 = no staging of new tiles (wrong results)
        const char* seed_env = getenv("ORBHIP_NN_SEED");                              // (read per call, like ORBHIP_NN: tests switch forms inside one process)
        const bool seeded_default = !(seed_env && seed_env[0] == '0');
        if (seeded_default && shape == 42156 && nch >= 2) {
            // ROWS PER WORKGROUP of the main pass.  Every workgroup does the same work, so a launch runs in rounds of (2 x CUs) workgroups: 609 chunks of 2^15
            // rows x 4 query groups = 2436 workgroups on 512 slots took five rounds with the last one three quarters empty.  The rows behind the head are split
            // into the number of chunks that fills a whole number of rounds instead (a multiple of 256 rows, at most 2^15, at least 2048: a short database is
            // spread over the chip instead of scanned by a handful of workgroups).  ORBHIP_NN_BLOCK=0 and the ablations keep chunks of 2^15 rows.
            const char* blk_env0 = getenv("ORBHIP_NN_BLOCK");
            const bool balanced = !(blk_env0 && blk_env0[0] == '0') && !getenv("ORBHIP_NN_ABLATE") && !(getenv("ORBHIP_NN_BALANCE") && getenv("ORBHIP_NN_BALANCE")[0] == '0');
            // ORBHIP_NN_WAVES=8 (measurement): the main pass with eight wavefronts per workgroup, one workgroup per CU, 1024 queries per staged tile
            const char* blk_env1 = getenv("ORBHIP_NN_BLOCK");
            const bool waves8 = getenv("ORBHIP_NN_WAVES") && atoi(getenv("ORBHIP_NN_WAVES")) == 8 && !(blk_env1 && blk_env1[0] == '0') && !getenv("ORBHIP_NN_ABLATE") && !getenv("ORBHIP_NN_BLOCK_VAR");
            const int qgm = waves8 ? 2 * qg : qg;                                  // queries per workgroup of the main pass
            int chrows = 1 << 15, nmain = nch - 1;
            if (balanced) {
                static int ncu_of[64];                                             // (a benign race: every writer stores the same value)
                int dev = 0; (void)hipGetDevice(&dev); dev = std::min(std::max(dev, 0), 63);
                if (!ncu_of[dev]) { hipDeviceProp_t pr; ncu_of[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? std::max(1, pr.multiProcessorCount) : 256; }
                const long long M = ndb - ((long long)1 << 15), nqg = (nq + qgm - 1) / qgm, slots = (waves8 ? 1LL : 2LL) * ncu_of[dev];
                const long long rounds = std::max(1LL, (M * nqg + 32768LL * slots - 1) / (32768LL * slots));
                const long long want = std::max(1LL, rounds * slots / nqg);
                long long cr = ((M + want - 1) / want + 255) / 256 * 256;
                cr = std::min(32768LL, std::max(2048LL, cr));
                chrows = (int)cr; nmain = (int)((M + cr - 1) / cr);
            }
            const int nhead = 64, stride = nhead + nmain;
            NNPart* p2 = (NNPart*)orbhip_nn_workspace(sizeof(NNPart) * (size_t)nq * stride + sizeof(int) * ((size_t)nq * 3 + 2), s);
            if (!p2) return false;
            parts = p2;
            int* seed = reinterpret_cast<int*>(p2 + (size_t)nq * stride);
            int* share = seed + nq;                                                // [nq] best, [nq] second best: k_hamming_nn_fp4b's shared bounds
            int share_mask = NN_SHARE_EVERY - 1;                                   // ORBHIP_NN_SHARE_EVERY (measurement only): supersteps between two reads of the shared bounds, a power of two
            if (const char* ev = getenv("ORBHIP_NN_SHARE_EVERY")) { const int v = atoi(ev); if (v >= 1 && (v & (v - 1)) == 0) share_mask = v - 1; }
            int* stats = nullptr;                                                  // ORBHIP_NN_STATS=1 (measurement only): kept / examined (tile, query tile) pairs of the main pass on stderr
            if (getenv("ORBHIP_NN_STATS")) stats = seed + 3 * (size_t)nq;
            const long long ndb_all = ndb;
            ndb = (long long)1 << 15;                                              // the head pass sees the first chunk only
#define NN_FP4X(LCH, GY, SEED, C0, P0) hipLaunchKernelGGL((k_hamming_nn_fp4<4, 2, LCH, 6, 0, true>), dim3((nq + qg - 1) / qg, GY, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)d_dbx, ndb, base, parts, stride, SEED, C0, P0)
#define NN_FP4B(LCH, EXP, DB, GY, SEED, SHARE, C0, CR, P0) hipLaunchKernelGGL((k_hamming_nn_fp4b<LCH, EXP>), dim3((nq + qg - 1) / qg, GY, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)(DB), ndb, base, parts, stride, SEED, SHARE, C0, CR, P0, stats, share_mask)
            // ORBHIP_NN_BLOCK=0 (measurement only): the compiler-scheduled tile loop of round 5 / 6 instead of the hand-ordered superstep
            const char* blk_env = getenv("ORBHIP_NN_BLOCK");
            const bool block = !(blk_env && blk_env[0] == '0') && ablate == 0;
            if (block && d_dbx) NN_FP4B(9, true, d_dbx, nhead, (const int*)nullptr, (int*)nullptr, 0LL, 512, 0); else if (block) NN_FP4B(9, false, d_db, nhead, (const int*)nullptr, (int*)nullptr, 0LL, 512, 0);
            else if (d_dbx) NN_FP4X(9, nhead, (const int*)nullptr, 0, 0); else NN_FP4(4, 2, 9, 6, nhead, (const int*)nullptr, 0, 0, stride);
            ndb = ndb_all;
            hipLaunchKernelGGL(k_hamming_seed, dim3((nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, (const NNPart*)parts, nq, stride, nhead, seed, share);
            if (stats) (void)hipMemsetAsync(stats, 0, 2 * sizeof(int), s);      // (the head's own pairs are not counted)
            const int bvar = getenv("ORBHIP_NN_BLOCK_VAR") ? atoi(getenv("ORBHIP_NN_BLOCK_VAR")) : 0;              // measurement only
            const char* share_env = getenv("ORBHIP_NN_SHARE");                                                       // ORBHIP_NN_SHARE=0 (measurement only): the head's bound alone
            int* const share_arg = share_env && share_env[0] == '0' ? (int*)nullptr : share;
            if (block && d_dbx && (bvar == 1 || bvar == 3 || bvar == 4 || bvar == 5 || bvar == 7)) {
#define NN_FP4BV(V) hipLaunchKernelGGL((k_hamming_nn_fp4b<15, true, V>), dim3((nq + qg - 1) / qg, nmain, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)d_dbx, ndb, base, parts, stride, (const int*)seed, share_arg, 32768LL, chrows, nhead, stats, share_mask)
                if (bvar == 1) NN_FP4BV(1); else if (bvar == 3) NN_FP4BV(3); else if (bvar == 4) NN_FP4BV(4); else if (bvar == 5) NN_FP4BV(5); else NN_FP4BV(7);
#undef NN_FP4BV
            }
            else if (block && waves8) {
#define NN_FP4B8(EXP, DB) hipLaunchKernelGGL((k_hamming_nn_fp4b<15, EXP, 0, 8>), dim3((nq + qgm - 1) / qgm, nmain, 1), dim3(512, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)(DB), ndb, base, parts, stride, (const int*)seed, share_arg, 32768LL, chrows, nhead, stats, share_mask)
                if (d_dbx) NN_FP4B8(true, d_dbx); else NN_FP4B8(false, d_db);
#undef NN_FP4B8
            }
            else if (block && d_dbx) NN_FP4B(15, true, d_dbx, nmain, (const int*)seed, share_arg, 32768LL, chrows, nhead); else if (block) NN_FP4B(15, false, d_db, nmain, (const int*)seed, share_arg, 32768LL, chrows, nhead);
            else if (ablate == 0 && d_dbx) NN_FP4X(15, nch - 1, (const int*)seed, 1, nhead);
#undef NN_FP4X
#undef NN_FP4B
            else if (ablate == 0) NN_FP4(4, 2, 15, 6, nch - 1, (const int*)seed, 1, nhead, stride);
            else {          // the same launch with parts of the loop compiled out: where the time goes (profiles/r06_exp_config5_ablation.txt)
#define NN_FP4_ABL(A) hipLaunchKernelGGL((k_hamming_nn_fp4<4, 2, 15, 6, A>), dim3((nq + qg - 1) / qg, nch - 1, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq, (const unsigned*)d_db, ndb, base, parts, stride, (const int*)seed, 1, nhead)
                if (ablate == 1) NN_FP4_ABL(1); else if (ablate == 2) NN_FP4_ABL(2); else if (ablate == 7) NN_FP4_ABL(7); else NN_FP4_ABL(3);
#undef NN_FP4_ABL
            }
            hipLaunchKernelGGL(k_hamming_merge, dim3((nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, (const NNPart*)parts, nq, stride, d_best_idx, d_best_dist, d_second);
            if (stats) { int hst[2] = {0, 0}; (void)hipMemcpyAsync(hst, stats, sizeof(hst), hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); fprintf(stderr, "orbhip: FP4 scan kept %d of %d (tile, query tile) pairs\n", hst[0], hst[1]); }
            return true;
        }
#define NN_FP4_PLAIN(QT, OCC, LCH, TPB) NN_FP4(QT, OCC, LCH, TPB, nch, (const int*)nullptr, 0, 0, nch)
        switch (shape) {
        case 22131: NN_FP4_PLAIN(2, 2, 13, 1); break; case 32131: NN_FP4_PLAIN(3, 2, 13, 1); break; case 42131: NN_FP4_PLAIN(4, 2, 13, 1); break;
        case 42151: NN_FP4_PLAIN(4, 2, 15, 1); break; case 42152: NN_FP4_PLAIN(4, 2, 15, 2); break; case 42154: NN_FP4_PLAIN(4, 2, 15, 4); break; case 42158: NN_FP4_PLAIN(4, 2, 15, 8); break;
        case 23154: NN_FP4_PLAIN(2, 3, 15, 4); break; case 23151: NN_FP4_PLAIN(2, 3, 15, 1); break; case 22154: NN_FP4_PLAIN(2, 2, 15, 4); break; case 42134: NN_FP4_PLAIN(4, 2, 13, 4); break;
        case 42164: NN_FP4_PLAIN(4, 2, 16, 4); break; case 32154: NN_FP4_PLAIN(3, 2, 15, 4); break; case 42156: NN_FP4_PLAIN(4, 2, 15, 6); break; case 42153: NN_FP4_PLAIN(4, 2, 15, 3); break;
        default: ok = false;
        }
#undef NN_FP4_PLAIN
#undef NN_FP4
        if (!ok) return false;
        hipLaunchKernelGGL(k_hamming_merge, dim3((nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, (const NNPart*)parts, nq, nch, d_best_idx, d_best_dist, d_second);
        return true;
    } else if (form >= 1 && ndb >= 4 * NN_CHUNK) {
        hipLaunchKernelGGL(k_hamming_nn_mfma, dim3((nq + NNM_QG - 1) / NNM_QG, nchunks, 1), dim3(256, 1, 1), 0, s, (const unsigned*)d_q, nq,
                           (const unsigned*)d_db, ndb, base, parts, nchunks);
    } else {
        const int qblocks = (nq + NN_T * NN_QPT - 1) / (NN_T * NN_QPT);
        hipLaunchKernelGGL(k_hamming_nn, dim3(qblocks, nchunks, 1), dim3(NN_T, 1, 1), 0, s, (const unsigned long long*)d_q, nq,
                           (const unsigned long long*)d_db, ndb, base, parts, nchunks);
    }
    hipLaunchKernelGGL(k_hamming_merge, dim3((nq + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, (const NNPart*)parts, nq, nchunks, d_best_idx, d_best_dist, d_second);
    return true;
}

// ------------------------------------------------------------------------------------------------ frame grid
// Frame::AssignFeaturesToGrid on the current frame F2.  posX = round((x-mnMinX)*inv) — `round`, not floor (Frame.cc:384-385).
// One workgroup per frame, everything between the first read of the key points and the last write of the table in LDS: cell of every key point
// (u16), 3072 counters, the bucket table itself.  A frame's grid is latency, not work - as one launch among three of a single-frame search it
// used to cost 37 us of dependent global-memory round trips (per-cell insertion sort and key point gathers in HBM); now the key points are read
// once, coalesced, and the table is written once.
__device__ __forceinline__ int mg_wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off); if (lane >= off) v += t; }
    return v;
}
__device__ __forceinline__ int mg_cell(const orbhip_keypoint& k, const MatchParams& M, float gwInv, float ghInv)
{
    if (!(M.grid_all_levels || k.octave == 0)) return 0xFFFF;
    const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, M.min_x), gwInv)), py = (int)roundf(__fmul_rn(__fsub_rn(k.y, M.min_y), ghInv));
    return (px < 0 || px >= ORBHIP_GRID_COLS || py < 0 || py >= ORBHIP_GRID_ROWS) ? 0xFFFF : px * ORBHIP_GRID_ROWS + py;
}
__global__ __launch_bounds__(256) void k_match_grid(MatchParams M, float gwInv, float ghInv)
{
    HIP_DYNAMIC_SHARED(int, lds)
    int* s_cnt = lds;                                                    // [CELLS + 1] counters, then cursors
    int* s_items = s_cnt + ORBHIP_GRID_CELLS + 1;                         // [cap] bucket table (key point indices)
    unsigned short* s_cell = reinterpret_cast<unsigned short*>(s_items + M.cap);      // [cap] cell of key point i (0xFFFF: not in the grid)
    __shared__ int s_wsum[4];
    const int slot = blockIdx.x + M.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = min(M.n2[slot], M.cap);
    const orbhip_keypoint* kp = M.kp2 + (long long)slot * M.cap;
    int* gstart = M.grid_start + (long long)slot * (ORBHIP_GRID_CELLS + 1);
    int* gitems = M.grid_items + (long long)slot * M.cap;
    float2* gxy = M.grid_xy + (long long)slot * M.cap;
    for (int c = tid; c <= ORBHIP_GRID_CELLS; c += 256) s_cnt[c] = 0;
    __syncthreads();
    // Only level-0 keypoints can ever be returned by GetFeaturesInArea(.., minLevel 0, maxLevel 0) (Frame.cc:362-370), and
    // filtering a cell keeps the relative order of its entries, so the buckets are built from level-0 keypoints only.
    for (int i = tid; i < n2; i += 256) {
        const int cell = mg_cell(kp[i], M, gwInv, ghInv);
        if (cell != 0xFFFF) atomicAdd(&s_cnt[cell], 1);
        s_cell[i] = (unsigned short)cell;
    }
    __syncthreads();
    // exclusive scan of 3072 counters: 12 consecutive ones per thread, a shuffle scan per wave, the four wave totals through LDS
    const int per = ORBHIP_GRID_CELLS / 256;
    int sum = 0;
    for (int k = 0; k < per; k++) sum += s_cnt[tid * per + k];
    const int incl = mg_wave_incl_scan(sum, lane);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; w++) run += s_wsum[w];
    for (int k = 0; k < per; k++) { const int v = s_cnt[tid * per + k]; s_cnt[tid * per + k] = run; gstart[tid * per + k] = run; run += v; }
    if (tid == 255) { gstart[ORBHIP_GRID_CELLS] = run; s_cnt[ORBHIP_GRID_CELLS] = run; }
    __syncthreads();
    const int total = s_cnt[ORBHIP_GRID_CELLS];
    __syncthreads();
    for (int i = tid; i < n2; i += 256) { const int cell = s_cell[i]; if (cell != 0xFFFF) s_items[atomicAdd(&s_cnt[cell], 1)] = i; }
    __syncthreads();
    // cells are tiny: restore keypoint order inside each cell (mGrid[x][y].push_back(i) for ascending i); after the scatter s_cnt[c] is the END of
    // cell c, which is where cell c + 1 starts
    for (int c = tid; c < ORBHIP_GRID_CELLS; c += 256) {
        const int a = c ? s_cnt[c - 1] : 0, b = s_cnt[c];
        for (int i = a + 1; i < b; i++) { const int v = s_items[i]; int j = i - 1; while (j >= a && s_items[j] > v) { s_items[j + 1] = s_items[j]; j--; } s_items[j + 1] = v; }
    }
    __syncthreads();
    for (int t = tid; t < total; t += 256) { const int i = s_items[t]; gitems[t] = i; const orbhip_keypoint k = kp[i]; float2 xy; xy.x = k.x; xy.y = k.y; gxy[t] = xy; }
}

// The same table for frames whose capacity does not fit the LDS form (12 KB + 6 bytes per key point: from ~23 000 key points per frame): counters and
// cursors in LDS, the bucket table built in place in global memory, the cell of a key point computed twice instead of kept.
__global__ __launch_bounds__(256) void k_match_grid_big(MatchParams M, float gwInv, float ghInv)
{
    __shared__ int s_cnt[ORBHIP_GRID_CELLS + 1];
    __shared__ int s_wsum[4];
    const int slot = blockIdx.x + M.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = min(M.n2[slot], M.cap);
    const orbhip_keypoint* kp = M.kp2 + (long long)slot * M.cap;
    int* gstart = M.grid_start + (long long)slot * (ORBHIP_GRID_CELLS + 1);
    int* gitems = M.grid_items + (long long)slot * M.cap;
    float2* gxy = M.grid_xy + (long long)slot * M.cap;
    for (int c = tid; c <= ORBHIP_GRID_CELLS; c += 256) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += 256) { const int cell = mg_cell(kp[i], M, gwInv, ghInv); if (cell != 0xFFFF) atomicAdd(&s_cnt[cell], 1); }
    __syncthreads();
    const int per = ORBHIP_GRID_CELLS / 256;
    int sum = 0;
    for (int k = 0; k < per; k++) sum += s_cnt[tid * per + k];
    const int incl = mg_wave_incl_scan(sum, lane);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; w++) run += s_wsum[w];
    for (int k = 0; k < per; k++) { const int v = s_cnt[tid * per + k]; s_cnt[tid * per + k] = run; gstart[tid * per + k] = run; run += v; }
    if (tid == 255) { gstart[ORBHIP_GRID_CELLS] = run; s_cnt[ORBHIP_GRID_CELLS] = run; }
    __syncthreads();
    const int total = s_cnt[ORBHIP_GRID_CELLS];
    __syncthreads();
    for (int i = tid; i < n2; i += 256) { const int cell = mg_cell(kp[i], M, gwInv, ghInv); if (cell != 0xFFFF) gitems[atomicAdd(&s_cnt[cell], 1)] = i; }
    __threadfence_block();
    __syncthreads();
    for (int c = tid; c < ORBHIP_GRID_CELLS; c += 256) {                 // key point order inside each cell (see k_match_grid)
        const int a = c ? s_cnt[c - 1] : 0, b = s_cnt[c];
        for (int i = a + 1; i < b; i++) { const int v = gitems[i]; int j = i - 1; while (j >= a && gitems[j] > v) { gitems[j + 1] = gitems[j]; j--; } gitems[j + 1] = v; }
    }
    __threadfence_block();
    __syncthreads();
    for (int t = tid; t < total; t += 256) { const orbhip_keypoint k = kp[gitems[t]]; float2 xy; xy.x = k.x; xy.y = k.y; gxy[t] = xy; }
}

size_t orbhip_match_grid_lds(int cap) { return sizeof(int) * (ORBHIP_GRID_CELLS + 1 + (size_t)cap) + sizeof(unsigned short) * ((size_t)cap + 2); }

void orbhip_launch_match_grid(const MatchParams& M, int nslots, hipStream_t s)
{
    const float gwInv = (float)ORBHIP_GRID_COLS / (float)(M.max_x - M.min_x), ghInv = (float)ORBHIP_GRID_ROWS / (float)(M.max_y - M.min_y);   // Frame.cc:101-102
    size_t lds_max = (size_t)150 * 1024;
#ifdef ORBHIP_TEST_HOOKS      // the CPU emulation build only: small frames through the large-frame form
    if (const char* e = getenv("ORBHIP_TEST_MATCH_GRID_LDS_MAX")) lds_max = (size_t)atol(e);
#endif
    if (orbhip_match_grid_lds(M.cap) <= lds_max) hipLaunchKernelGGL(k_match_grid, dim3(nslots, 1, 1), dim3(256, 1, 1), orbhip_match_grid_lds(M.cap), s, M, gwInv, ghInv);
    else hipLaunchKernelGGL(k_match_grid_big, dim3(nslots, 1, 1), dim3(256, 1, 1), 0, s, M, gwInv, ghInv);
}

// ------------------------------------------------------------------------------------------------ candidates
// For the j-th level-0 keypoint of the previous frame: F2.GetFeaturesInArea(prev.x, prev.y, window, 0, 0) in the
// reference's order (ix outer, iy inner, keypoint order inside a cell) + DescriptorDistance to each candidate.
//
// The bucket table written by k_match_grid lists F2's (level-0) keypoints in exactly that order: by cell ix*ROWS+iy, then by
// index.  GetFeaturesInArea visits the cells [floor((x-r)/w) .. ceil((x+r)/w)] x [..] and keeps keypoints with |dx| < r and
// |dy| < r (Frame.cc:327-380); a keypoint that passes the distance test always lies in a visited cell (its cell is
// round(kx/w), and floor(a) <= round(v) <= ceil(b) for a < v < b; the sub-ulp slack of the float subtraction is far below
// the 0.5 of the rounding), so the result is the sub-sequence of the whole table that passes the distance test.  One
// wavefront scans the table (a few hundred entries: coalesced, all loads in flight at once) instead of walking ~300
// mostly empty cells with dependent loads; ballot ranks keep the order.
// wave64 minimum with DPP row shifts / broadcasts (6 dependent 4-cycle VALU steps); result broadcast from lane 63
__device__ __forceinline__ int wave_min_dpp(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x114, 0xf, 0xe, false));      // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x118, 0xf, 0xc, false));      // row_shr:8 -> lane 15 of each row = row minimum
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(IMAX, v, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave minimum
    return __builtin_amdgcn_readlane(v, 63);
}

#define MS_K 4                          // best candidates recorded per query (under the initial state: nothing matched yet)
#define MS_REC (MS_K + 1)               // + one word: more candidates exist
#define MS_NONE 0xFFFFFFFFu
#define MC_CHUNKS 8
__global__ __launch_bounds__(256) void k_match_candidates(MatchParams M, float gwInv, float ghInv)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = blockIdx.y + M.slot0;
    const int j1 = blockIdx.x * 4 + wave;
    const int n1l = M.n1_lvl0[slot * M.lvl_stride];
    if (j1 >= n1l || j1 >= M.lvl0_cap) return;
    const int i1 = M.list1 ? M.list1[(long long)slot * M.lvl0_cap + j1] : j1;   // level-major extractor output: level 0 = indices [0, n_lvl0)
    const int nitems = M.grid_start[(long long)slot * (ORBHIP_GRID_CELLS + 1) + ORBHIP_GRID_CELLS];
    const int* gitems = M.grid_items + (long long)slot * M.cap;
    const float2* gxy = M.grid_xy + (long long)slot * M.cap;
    unsigned* cand = M.cand + ((long long)slot * M.lvl0_cap + j1) * M.cand_stride;
    const orbhip_keypoint* kp1 = M.kp1 + (long long)slot * M.cap;
    const float x = M.prev_from_kp1 ? kp1[i1].x : M.prev[((long long)slot * M.cap + i1) * 2];
    const float y = M.prev_from_kp1 ? kp1[i1].y : M.prev[((long long)slot * M.cap + i1) * 2 + 1];
    const float r = (float)M.window;
    const unsigned long long* d1 = (const unsigned long long*)(M.desc1 + ((long long)slot * M.cap + i1) * 32);
    const unsigned long long q0 = d1[0], q1 = d1[1], q2 = d1[2], q3 = d1[3];
    const unsigned long long below = (1ull << lane) - 1ull;
    int nc = 0;
    int tk[MS_K]; unsigned te[MS_K]; int nsel = 0;          // this lane's MS_K smallest (distance, list position) keys + their records
#pragma unroll
    for (int k = 0; k < MS_K; k++) { tk[k] = IMAX; te[k] = MS_NONE; }
    // The table is ordered by grid column first, so the entries of the columns the window can reach are ONE contiguous run of it: the columns
    // GetFeaturesInArea visits (Frame.cc:333-343: floor((x - mnMinX - r) inv) .. ceil((x - mnMinX + r) inv)), widened by one on either side (far
    // more than the float slack of that arithmetic).  A key point outside the run fails |dx| < r, so scanning the run alone gives the same
    // sub-sequence as scanning the whole table: at the metric's window (100 px of 1241) that is a fifth of it, 2 passes of 64 entries instead of 7.
    const int* gstart = M.grid_start + (long long)slot * (ORBHIP_GRID_CELLS + 1);
    const float cl = __fmul_rn(__fsub_rn(__fsub_rn(x, M.min_x), r), gwInv), ch = __fmul_rn(__fadd_rn(__fsub_rn(x, M.min_x), r), gwInv);
    // (the comparisons are written so that a NaN position scans the whole table, like the distance test it would fail everywhere)
    const int col_lo = cl >= 1.0f ? (int)fminf(floorf(cl) - 1.0f, (float)ORBHIP_GRID_COLS) : 0;
    const int col_hi = ch < (float)(ORBHIP_GRID_COLS - 2) ? (int)fmaxf(ceilf(ch) + 2.0f, 0.0f) : ORBHIP_GRID_COLS;       // one past the last column scanned
    const int t_lo = __builtin_amdgcn_readfirstlane(gstart[min(col_lo, col_hi) * ORBHIP_GRID_ROWS]), t_hi = __builtin_amdgcn_readfirstlane(gstart[col_hi * ORBHIP_GRID_ROWS]);
    (void)nitems;
    for (int tb = t_lo; tb < t_hi; tb += 64 * MC_CHUNKS) {
        float2 k[MC_CHUNKS]; int it[MC_CHUNKS];
#pragma unroll
        for (int c = 0; c < MC_CHUNKS; c++) {
            const int t = tb + 64 * c + lane;
            k[c].x = 0.0f; k[c].y = 0.0f; it[c] = 0;
            if (t < t_hi) { k[c] = gxy[t]; it[c] = gitems[t]; }
        }
#pragma unroll
        for (int c = 0; c < MC_CHUNKS; c++) {
            if (tb + 64 * c >= t_hi) break;
            const bool ok = tb + 64 * c + lane < t_hi && fabsf(__fsub_rn(k[c].x, x)) < r && fabsf(__fsub_rn(k[c].y, y)) < r;    // Frame.cc:367-371
            const unsigned long long m = __ballot(ok);
            if (m == 0) continue;
            const int pos = nc + __popcll(m & below);
            if (ok && pos < M.cand_stride) {
                const unsigned long long* d2 = (const unsigned long long*)(M.desc2 + ((long long)slot * M.cap + it[c]) * 32);
                const int dist = __popcll(q0 ^ d2[0]) + __popcll(q1 ^ d2[1]) + __popcll(q2 ^ d2[2]) + __popcll(q3 ^ d2[3]);
                unsigned rec = (unsigned)it[c] | ((unsigned)dist << 20);       // DescriptorDistance (ORBmatcher.cc:442)
                cand[pos] = rec;
                int key = (dist << 20) | pos;
                nsel++;
#pragma unroll
                for (int k = 0; k < MS_K; k++) if (key < tk[k]) { const int tkk = tk[k]; const unsigned tee = te[k]; tk[k] = key; te[k] = rec; key = tkk; rec = tee; }   // sorted insert
            }
            nc += __popcll(m);
        }
    }
    nc = min(nc, M.cand_stride);
    if (lane == 0) M.ncand[(long long)slot * M.lvl0_cap + j1] = nc;
    // The best / second-best update of the reference (ORBmatcher.cc:447-456) ends with the two smallest (distance, list position)
    // keys among the candidates it does not skip.  Record the MS_K smallest of the whole list: k_match_select takes the first two
    // that are not skipped at its turn.
    unsigned out = MS_NONE; int popped = 0;
#pragma unroll
    for (int r = 0; r < MS_K; r++) {
        const int m = wave_min_dpp(tk[0]);
        if (m == IMAX) break;
        const int owner = __ffsll((long long)__ballot(tk[0] == m)) - 1;
        const unsigned rec = (unsigned)__builtin_amdgcn_readlane((int)te[0], owner);
        if (lane == r) out = rec;
        if (lane == owner) {
            popped++;
#pragma unroll
            for (int k = 0; k + 1 < MS_K; k++) { tk[k] = tk[k + 1]; te[k] = te[k + 1]; }
            tk[MS_K - 1] = IMAX; te[MS_K - 1] = MS_NONE;
        }
    }
    const bool more = __ballot(nsel > popped) != 0ull;
    unsigned* top = M.top + ((long long)slot * M.lvl0_cap + j1) * MS_REC;
    if (lane < MS_K) top[lane] = out;
    if (lane == MS_K) top[MS_K] = more ? 1u : 0u;
}

void orbhip_launch_match_candidates(const MatchParams& M, int nslots, hipStream_t s)
{
    const float gwInv = (float)ORBHIP_GRID_COLS / (float)(M.max_x - M.min_x), ghInv = (float)ORBHIP_GRID_ROWS / (float)(M.max_y - M.min_y);
    hipLaunchKernelGGL(k_match_candidates, dim3((M.lvl0_cap + 3) / 4, nslots, 1), dim3(256, 1, 1), 0, s, M, gwInv, ghInv);
}

// ------------------------------------------------------------------------------------------------ select

#define MS_T 256

// One workgroup per camera slot.  All 4 waves initialise the per-slot tables in LDS, then wave 0 alone resolves the
// order-dependent loop over F1's level-0 keypoints, 64 of them per step, from the MS_K records k_match_candidates left per key
// point (see the loop).  The candidate lists themselves (i2 | dist<<20, canonical order) stay in HBM/L2 and are only read for the
// rare key point whose records are used up; the kernel's LDS footprint stays small enough not to displace the workgroups of
// the extraction kernels it runs beside (an 80 KB staged copy of the lists used to halve k_fast_cells' occupancy on every CU).
__device__ __forceinline__ int orbhip_match_select_ints_d(int cap, int lvl0_cap) { return 4 * cap + 4 * lvl0_cap + ORBHIP_HISTO_LENGTH + 8; }
// BIG: the per-slot tables in device memory (M.big_ws) instead of LDS, for frames of more key points than the LDS holds (nfeatures from ~8300 on at 1080p; the
// reference takes any nFeatures, Tracking.cc:113-125).  Same statements on volatile global words (one wave resolves the loop: program order is the order), see
// proj_select_body<BIG>.
template <bool BIG> __device__ __forceinline__ void match_select_body(const MatchParams& M)
{
    typedef typename std::conditional<BIG, volatile int, int>::type TI;
    typedef typename std::conditional<BIG, volatile float, float>::type TF;
    const int slot = blockIdx.x + M.slot0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = M.n1[slot], n2 = M.n2[slot];
    const int n1l = min(M.n1_lvl0[slot * M.lvl_stride], M.lvl0_cap);
    const int* list1 = M.list1 ? M.list1 + (long long)slot * M.lvl0_cap : nullptr;
    HIP_DYNAMIC_SHARED(int, lds)
    TI* s_md = BIG ? M.big_ws + (long long)(blockIdx.x + M.slot0) * (long long)orbhip_match_select_ints_d(M.cap, M.lvl0_cap) : lds;     // vMatchedDistance[i2]
    TI* s_m21 = s_md + M.cap;               // vnMatches21[i2]
    TI* s_m12 = s_m21 + M.cap;              // vnMatches12[i1] for level-0 i1
    TI* s_bin = s_m12 + M.lvl0_cap;         // rotation bin of the accepted match of i1, -1 = none
    TI* s_nc = s_bin + M.lvl0_cap;          // candidate count per level-0 i1
    TI* s_hist = s_nc + M.lvl0_cap;         // [HISTO_LENGTH] + misc
    TI* s_stamp = s_hist + ORBHIP_HISTO_LENGTH + 8;                                // lowest undecided query that wants to claim feature i2
    TF* s_ang1 = reinterpret_cast<TF*>(s_stamp + M.cap);                           // angle of F1's level-0 keypoint j1
    TF* s_ang2 = s_ang1 + M.lvl0_cap;                                              // angle of F2's keypoint i2
    auto amin = [](TI* p, int v) { atomicMin(const_cast<int*>(p), v); };
    auto aadd = [](TI* p, int v) { atomicAdd(const_cast<int*>(p), v); };
    const orbhip_keypoint* kp1 = M.kp1 + (long long)slot * M.cap;
    const orbhip_keypoint* kp2 = M.kp2 + (long long)slot * M.cap;
    int* m12 = M.matches12 + (long long)slot * M.cap;
    float* prev = M.prev + (long long)slot * M.cap * 2;
    const unsigned* cand0 = M.cand + (long long)slot * M.lvl0_cap * M.cand_stride;
    for (int i = tid; i < n2; i += MS_T) { s_md[i] = IMAX; s_m21[i] = -1; s_stamp[i] = IMAX; s_ang2[i] = kp2[i].angle; }
    for (int i = tid; i < n1l; i += MS_T) { s_m12[i] = -1; s_bin[i] = -1; s_nc[i] = M.ncand[(long long)slot * M.lvl0_cap + i]; s_ang1[i] = kp1[list1 ? list1[i] : i].angle; }
    for (int i = tid; i < ORBHIP_HISTO_LENGTH + 8; i += MS_T) s_hist[i] = 0;
    for (int i = tid; i < n1; i += MS_T) { m12[i] = -1; if (M.prev_from_kp1) { prev[2 * i] = kp1[i].x; prev[2 * i + 1] = kp1[i].y; } }
    __syncthreads();
    if (wave == 0) {
        const float factor = 1.0f / ORBHIP_HISTO_LENGTH;
        const unsigned* top0 = M.top + (long long)slot * M.lvl0_cap * MS_REC;
        auto rot_bin = [&](int j1, int i2) -> int {                                        // :470-480
            float rot = __fsub_rn(s_ang1[j1], s_ang2[i2]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == ORBHIP_HISTO_LENGTH) bin = 0;
            return min(max(bin, 0), ORBHIP_HISTO_LENGTH - 1);
        };
        // 64 previous-frame key points per step.  A candidate is skipped once it is matched at a distance <= this query's
        // (vMatchedDistance only ever decreases), so a query's best / second-best are the first two recorded candidates that are
        // not skipped when its turn comes.  All lanes decide at once; a lane one of whose relevant records an earlier, still
        // undecided lane wants to claim (atomicMin stamp) waits for the next iteration; only a query whose records are used up
        // while its list holds more is rescanned from the list.
        for (int jb = 0; jb < n1l; jb += 64) {
            const int j1 = jb + lane;
            const bool inb = j1 < n1l;
            int ei[MS_K], ed[MS_K]; int nk = 0;
#pragma unroll
            for (int k = 0; k < MS_K; k++) {
                const unsigned t = (inb && s_nc[j1] != 0) ? top0[(long long)j1 * MS_REC + k] : MS_NONE;
                ei[k] = (int)(t & 0xFFFFFu); ed[k] = (int)((t >> 20) & 0x1FFu);
                if (t != MS_NONE) nk = k + 1;
            }
            const bool more = nk > 0 && top0[(long long)j1 * MS_REC + MS_K] != 0u;
            int stamped = -1;
            unsigned long long todo = __ballot(nk > 0);
            while (todo) {
                const bool mine = (todo >> lane) & 1ull;
                const int lowest = __ffsll((long long)todo) - 1;
                int a = -1, b = -1;
#pragma unroll
                for (int k = 0; k < MS_K; k++)
                    if (mine && k < nk && b < 0 && !(s_md[ei[k]] <= ed[k])) { if (a < 0) a = k; else b = k; }      // :444-445
                const int ia = a >= 0 ? ei[a] : 0, da = a >= 0 ? ed[a] : IMAX, db = b >= 0 ? ed[b] : IMAX;
                const bool exhausted = mine && more && b < 0;
                const bool accept = mine && !exhausted && a >= 0 && da <= ORBHIP_TH_LOW && (float)da < __fmul_rn((float)db, M.nnratio);    // :459-461
                const int want = accept ? ia : -1;
                if (stamped >= 0 && stamped != want && s_stamp[stamped] == j1) s_stamp[stamped] = IMAX;       // withdraw an outdated claim
                __builtin_amdgcn_wave_barrier();
                if (want >= 0) amin(&s_stamp[want], j1);
                stamped = want;
                __builtin_amdgcn_wave_barrier();
                bool unsure = false;
                const int last = b >= 0 ? b : nk - 1;
#pragma unroll
                for (int k = 0; k < MS_K; k++)
                    if (mine && lane != lowest && k <= last && !(s_md[ei[k]] <= ed[k]) && s_stamp[ei[k]] < j1) unsure = true;
                const unsigned long long bad = __ballot(unsure || exhausted);
                const int first_bad = bad ? __ffsll((long long)bad) - 1 : 64;
                const unsigned long long commit = first_bad == 64 ? todo : (todo & ((1ull << first_bad) - 1ull));
                const bool win = ((commit >> lane) & 1ull) && accept;
                if (win) {                                                                  // claimed features are distinct within one commit
                    const int old = s_m21[ia];
                    if (old >= 0) s_m12[old] = -1;                                          // :463-467
                    s_m12[j1] = ia; s_m21[ia] = j1; s_md[ia] = da;
                    if (s_stamp[ia] == j1) s_stamp[ia] = IMAX;                              // a decided claim lives in vMatchedDistance
                    stamped = -1;
                    if (M.check_ori) { const int bin = rot_bin(j1, ia); s_bin[j1] = bin; aadd(&s_hist[bin], 1); }      // rotHist[bin].push_back(i1): never removed when stolen
                }
                __builtin_amdgcn_wave_barrier();
                todo &= ~commit;
                if (first_bad == 64) break;
                if (first_bad != lowest || !((__ballot(exhausted) >> first_bad) & 1ull)) continue;
                // rescan the list of key point jb + first_bad against the current state
                todo &= ~(1ull << first_bad);
                if (lane == first_bad && stamped >= 0 && s_stamp[stamped] == j1) s_stamp[stamped] = IMAX;
                const int js = jb + first_bad, nc = s_nc[js];
                const unsigned* cand = cand0 + (long long)js * M.cand_stride;
                int best = IMAX, second = IMAX, bidx = -1;
                for (int cb = 0; cb < nc; cb += 64) {
                    const int t = cb + lane;
                    const unsigned e = t < nc ? cand[t] : 0u;
                    const int i2 = (int)(e & 0xFFFFFu), dist = (int)(e >> 20);
                    const bool valid = t < nc && !(s_md[i2] <= dist);                    // :444-445
                    // smallest (distance, lane) key by a DPP min network: strict '<' means the first candidate with the minimum
                    // wins (:447-452); the runner-up is the minimum with that lane masked out
                    const int key = valid ? ((dist << 6) | lane) : IMAX;
                    const int k1 = wave_min_dpp(key);
                    if (k1 == IMAX) continue;
                    const int first = k1 & 63, wmin = k1 >> 6, ci = __builtin_amdgcn_readlane(i2, first);
                    const int k2 = wave_min_dpp(lane == first ? IMAX : key);
                    const int wsec = k2 == IMAX ? IMAX : (k2 >> 6);
                    if (wmin < best) { second = min(best, wsec); best = wmin; bidx = ci; } else second = min(second, wmin);
                }
                if (best <= ORBHIP_TH_LOW && (float)best < __fmul_rn((float)second, M.nnratio)) {      // :459-461
                    if (lane == 0) {
                        const int old = s_m21[bidx];
                        if (old >= 0) s_m12[old] = -1;
                        s_m12[js] = bidx; s_m21[bidx] = js; s_md[bidx] = best;
                        if (M.check_ori) { const int bin = rot_bin(js, bidx); s_bin[js] = bin; s_hist[bin] = s_hist[bin] + 1; }
                    }
                }
                __builtin_amdgcn_wave_barrier();                     // lane 0's LDS updates are read by the whole wave next
            }
        }
    }
    if (wave != 0) return;
    __builtin_amdgcn_wave_barrier();
    // nmatches of the reference (++ on accept, -- on steal :463-467 and on rotation reject :504-508) == final count of set entries
    if (M.check_ori) {
        if (lane == 0) {                                          // ComputeThreeMaxima (:1601-1642)
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
        for (int j = lane; j < n1l; j += 64) { const int b = s_bin[j]; if (b >= 0 && b != ind1 && b != ind2 && b != ind3) s_m12[j] = -1; }
        __builtin_amdgcn_wave_barrier();
    }
    int cnt = 0;
    for (int j = lane; j < n1l; j += 64) {
        const int m = s_m12[j];
        if (m >= 0) { const int i1 = list1 ? list1[j] : j; cnt++; m12[i1] = m; prev[2 * i1] = kp2[m].x; prev[2 * i1 + 1] = kp2[m].y; }      // :515-517
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) M.nmatches[slot] = cnt;
}

__global__ __launch_bounds__(MS_T) void k_match_select(MatchParams M) { match_select_body<false>(M); }
__global__ __launch_bounds__(MS_T) void k_match_select_big(MatchParams M) { match_select_body<true>(M); }

#define MS_LDS_BUDGET (158 * 1024)
size_t orbhip_match_select_ints(int cap, int lvl0_cap) { return (size_t)4 * cap + (size_t)4 * lvl0_cap + ORBHIP_HISTO_LENGTH + 8; }
bool orbhip_match_select_big(int cap, int lvl0_cap)
{
    const char* env = getenv("ORBHIP_SELECT_BIG");                       // tests: 1 = the device-memory form at any size (read per call: tests switch inside one process)
    const bool force = env && env[0] == '1';
    return force || sizeof(int) * orbhip_match_select_ints(cap, lvl0_cap) > MS_LDS_BUDGET;
}
void orbhip_launch_match_select(const MatchParams& M, int nslots, hipStream_t s)
{
    if (M.big_ws) hipLaunchKernelGGL(k_match_select_big, dim3(nslots, 1, 1), dim3(MS_T, 1, 1), 0, s, M);
    else hipLaunchKernelGGL(k_match_select, dim3(nslots, 1, 1), dim3(MS_T, 1, 1), sizeof(int) * orbhip_match_select_ints(M.cap, M.lvl0_cap), s, M);
}
