// orbhip_kernels_extract.hip — gfx950 kernels of the ORB extractor (replaces src/ORBextractor.cc of the reference).
//
// Kernels (one launch each covers every camera slot of the batch; grid.y / grid.z = frame):
//   k_pyramid_level  level l from level l-1, fixed-point bilinear        (ORBextractor.cc:1107-1132, cv::resize)
//   k_blur_mfma      7x7 sigma-2 fixed-point Gaussian as two banded i8 products on the matrix cores, LDS staged (default)
//   k_blur           the same on the VALU (ORBHIP_BLUR=valu)             (ORBextractor.cc:1085-1086, cv::GaussianBlur)
//   k_fast_cells     ONE WAVEFRONT PER GRID CELL: FAST-9/16 scores in LDS, iniTh/minTh fallback, 3x3 NMS inside
//                    the cell, row-major emission                          (ORBextractor.cc:789-829, cv::FAST)
//   k_quadtree       one workgroup per (frame, level): DistributeOctTree replayed with exact list order
//                    (ORBextractor.cc:539-763) on per-candidate quad-path codes instead of std::list nodes
//   k_describe       one wavefront per four keypoint slots: IC_Angle, fastAtan2, steered BRIEF via 4 ballots
//                    (ORBextractor.cc:77-147, 1034-1104)
// Everything is integer / bitwise except the orientation and the pattern rotation, which use explicitly
// rounded IEEE ops (__fmul_rn ...; file built with -ffp-contract=off) so results are bit-identical to the CPU
// oracle's two-rounding form.  The one GEMM-shaped piece is the blur (a banded product, exact in i8): it runs on MFMA.
#include "orbhip_internal.h"
#include <type_traits>

#define WAVE 64

// XCD-aware block -> (tile, frame) map.  The dispatcher places consecutive workgroup ids round-robin over the 8 XCDs
// (observed, used for speed only): giving all workgroups of frame f ids congruent to f mod 8 keeps that frame's planes
// (~3 MB) in ONE XCD's 4 MiB L2 instead of being re-fetched from HBM by all eight.
// FEWER THAN EIGHT FRAMES (the drop-in's single-image call): that rule would leave seven of the eight XCDs idle - a single frame's FAST ran its ~300
// workgroups on 32 CUs, two rounds deep, 19 us from the first workgroup's start to the last one's end (tools/span_experiment.py) - and every launch starts
// with a cold L2 anyway.  Then consecutive ids simply walk over (tile, frame): the work spreads over all XCDs.
__device__ __forceinline__ bool xcd_frame_map(int nb, int nframes, int& tile, int& frame)
{
    const int id = blockIdx.x;
    if (nframes == 1) { tile = id; frame = 0; return true; }
    if (nframes < 8) { tile = id / nframes; frame = id - tile * nframes; return true; }
    const int j = id >> 3, fg = j / nb;
    tile = j - fg * nb; frame = fg * 8 + (id & 7);
    return frame < nframes;
}
static inline unsigned xcd_grid(int nb, int nframes) { return nframes < 8 ? (unsigned)nb * (unsigned)nframes : (unsigned)nb * 8u * (unsigned)((nframes + 7) / 8); }

typedef unsigned short pku16 __attribute__((vector_size(4)));      // two u16 lanes in one VGPR -> v_pk_min_u16 / v_pk_max_u16, v_dot2_u32_u16
typedef short pki16 __attribute__((vector_size(4)));
typedef unsigned u32_unaligned __attribute__((aligned(1)));     // a 32-bit global load at any byte address
// a pointer every lane of the wave agrees on, moved to SGPRs so that loads use the scalar-base + 32-bit lane offset form.  The result is
// typed as a GLOBAL pointer: rebuilt from integers it would otherwise be a generic one, and every access through it a flat_load with a
// 64-bit address per lane (k_describe paid 26 v_lshl_add_u64 per key point for that in round 1).
#define ORBHIP_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ ORBHIP_GLOBAL T* uniform_ptr(T* p)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (ORBHIP_GLOBAL T*)(((unsigned long long)hi << 32) | lo);
}
// One element of a read-only table at an index every lane agrees on, through the SCALAR cache (s_load: the result lands in SGPRs and counts on lgkmcnt).
// As an ordinary load the compiler makes it a vector load - the kernel also stores to global memory, so it may not assume the table constant - and waiting
// for a vector load means waiting for every LDS-DMA request issued before it (vmcnt returns in order).  Tables only: written by the host before the launch.
#define ORBHIP_CONSTANT __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ T scalar_load(const T* p)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return *(const ORBHIP_CONSTANT T*)(((unsigned long long)hi << 32) | lo);
}
// LDS-DMA: every lane's dword goes from global memory straight to LDS dword (lane) of the 256-byte block at lds_block, without a
// VGPR round trip (global_load_lds_dword; M0 = block address, wave-uniform).  The source may sit at any byte address
// (tools/lds_dma_probe.hip, measured on MI355X).  The loads count on vmcnt: lds_dma_wait() before the first LDS read.
typedef const __attribute__((address_space(1))) void* orbhip_gptr;
typedef __attribute__((address_space(3))) void* orbhip_lptr;
__device__ __forceinline__ void lds_dma_dword(const ORBHIP_GLOBAL uint8_t* gsrc, uint8_t* lds_block)
{
    __builtin_amdgcn_global_load_lds((orbhip_gptr)gsrc, (orbhip_lptr)lds_block, 4, 0, 0);
}
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
#define ORBHIP_VMCNT(n) (0x0f70 | ((n) & 15) | (((n) >> 4) << 14))
__device__ __forceinline__ void lds_dma_wait() { __builtin_amdgcn_s_waitcnt(ORBHIP_VMCNT(0)); }

__device__ __forceinline__ const uint8_t* level_src(const ExtractParams& P, int frame, int level, int& pitch)
{
    if (level == 0) { pitch = P.img0_pitch; return P.img0 + (long long)frame * P.img0_frame_stride; }
    pitch = P.geom[level].pitch;
    return P.pyr + (long long)frame * P.plane_frame_bytes + P.geom[level].plane_off;
}

// ------------------------------------------------------------------------------------------------ pyramid
// cv::resize INTER_LINEAR CV_8U (OpenCV 3.2 HResizeLinear / VResizeLinear fixed point, 11-bit coefficients).
// Coefficient tables are built on the host exactly as OpenCV builds them; the kernel is pure integer.
// 4 output pixels per thread, one 32-bit store.
#define BM_ROWS 26                      // k_blur_mfma: output rows per wavefront (32 source rows - 6)
#define BM_COLS 32                      // k_blur_mfma: output columns per wavefront
#define BM_BLOCKS 7                     // k_blur_mfma: 32-column blocks per workgroup tile (224 columns: their sources fill one 256-byte row per LDS-DMA)
#define PYR_RPT 4                       // output rows per thread (4 pixels each): coefficient unpacking is amortised over them
// 4 output pixels of PYR_RPT consecutive rows from source rows addressed as rows[(y - row0) * rpitch + x]
__device__ __forceinline__ void pyr_rows(const LevelGeom& g, const int2* xt, const int2* yt, const uint8_t* rows, int rpitch, int row0, int x4, int ytop, uint8_t* dstp)
{
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int2 e = xt[min(x4 + k, g.w - 1)];
        sx[k] = e.x; sx1[k] = min(e.x + 1, g.src_w - 1); a0[k] = e.y & 0xffff; a1[k] = (e.y >> 16) & 0xffff;
    }
#pragma unroll
    for (int rr = 0; rr < PYR_RPT; rr++) {
        const int y = ytop + rr;
        if (y >= g.h) break;
        const int2 ye = yt[y];
        const unsigned b0s = (unsigned)ye.y << 16, b1s = (unsigned)ye.y & 0xffff0000u;    // coefficients << 16: (b * t) >> 16 == mul_hi(b << 16, t)
        const int ya = min(max(ye.x, 0), g.src_h - 1), yb = min(max(ye.x + 1, 0), g.src_h - 1);
        const uint8_t* r0 = rows + (long long)(ya - row0) * rpitch;
        const uint8_t* r1 = rows + (long long)(yb - row0) * rpitch;
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int h0 = r0[sx[k]] * a0[k] + r0[sx1[k]] * a1[k], h1 = r1[sx[k]] * a0[k] + r1[sx1[k]] * a1[k];
            const int v = (int)(__umulhi(b0s, (unsigned)h0 >> 4) + __umulhi(b1s, (unsigned)h1 >> 4) + 2u) >> 2;
            if (x4 + k < g.w) out |= (unsigned)(v & 0xff) << (8 * k);
        }
        *reinterpret_cast<unsigned*>(dstp + (long long)y * g.pitch) = out;
    }
}

// Workgroup = 256 x 16 output pixels: the source footprint (<= 24 rows x ~312 bytes) is staged in LDS with 32-bit loads,
// each thread then produces 4 pixels of 4 rows (one 32-bit store per row).
#define PYR_TW 256
#define PYR_TH (4 * PYR_RPT)
#define PYR_SROWS 24
#define PYR_SDW 84                      // dwords per staged source row (>= (256*1.25+2+3)/4)
// wave w stages source rows w, w+4, ...: row address on the scalar unit, lane = dword (two chunks of 64), all loads in flight
// before the first LDS write; byte path only for the ragged row end / unaligned sources
__device__ __forceinline__ void pyr_stage(const uint8_t* src, int spitch, int src_w, int sxa, int sya, int ndw, int nrows, bool aligned, int tid, unsigned* s_t)
{
    {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        unsigned v[PYR_SROWS / 4][2];
        bool edge[2];
#pragma unroll
        for (int c = 0; c < 2; c++) { const int d = lane + 64 * c; edge[c] = d < ndw && !(aligned && sxa + 4 * d + 3 < src_w); }
#pragma unroll
        for (int j = 0; j < PYR_SROWS / 4; j++) {
            const uint8_t* row = src + (long long)(sya + min(wave + 4 * j, nrows - 1)) * spitch + sxa;     // wave-uniform -> SGPR base
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int d = lane + 64 * c;
                v[j][c] = (d < ndw && !edge[c]) ? *reinterpret_cast<const unsigned*>(row + (unsigned)(4 * d)) : 0u;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
            if (edge[c]) {
                const int d = lane + 64 * c;
#pragma unroll
                for (int j = 0; j < PYR_SROWS / 4; j++) {
                    const uint8_t* row = src + (long long)(sya + min(wave + 4 * j, nrows - 1)) * spitch + sxa + 4 * d;
                    unsigned w = 0;
                    for (int q = 0; q < 4; q++) if (sxa + 4 * d + q < src_w) w |= (unsigned)row[q] << (8 * q);
                    v[j][c] = w;
                }
            }
#pragma unroll
        for (int j = 0; j < PYR_SROWS / 4; j++) {
            const int r = wave + 4 * j;
#pragma unroll
            for (int c = 0; c < 2; c++) { const int d = lane + 64 * c; if (r < nrows && d < ndw) s_t[r * PYR_SDW + d] = v[j][c]; }
        }
    }
}

__global__ __launch_bounds__(256) void k_pyramid_level(ExtractParams P, int level)
{
    const LevelGeom g = P.geom[level];
    const int gx = (g.w + PYR_TW - 1) / PYR_TW, gy = (g.h + PYR_TH - 1) / PYR_TH;
    int tile, frame;
    if (!xcd_frame_map(gx * gy, P.nframes, tile, frame)) return;
    frame += P.frame0;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x0 = (tile % gx) * PYR_TW, y0 = (tile / gx) * PYR_TH;
    int spitch; const uint8_t* src = level_src(P, frame, level - 1, spitch);
    __shared__ unsigned s_t[PYR_SROWS * PYR_SDW];
    const int2* xt = P.xtab + g.xtab_off; const int2* yt = P.ytab + g.ytab_off;
    const int xl = min(x0 + PYR_TW - 1, g.w - 1), yl = min(y0 + PYR_TH - 1, g.h - 1);
    const int sxa = xt[x0].x & ~3, sxb = min(xt[xl].x + 1, g.src_w - 1);                    // staged source columns [sxa, sxb]
    const int sya = min(max(yt[y0].x, 0), g.src_h - 1), syb = min(max(yt[yl].x + 1, 0), g.src_h - 1);
    const int ndw = ((sxb - sxa) >> 2) + 1, nrows = syb - sya + 1;
    const bool aligned = ((((unsigned long long)src) | (unsigned long long)spitch) & 3ull) == 0;
    pyr_stage(src, spitch, g.src_w, sxa, sya, ndw, nrows, aligned, tid, s_t);
    __syncthreads();
    const int x4 = x0 + threadIdx.x * 4;
    if (x4 >= g.w) return;
    uint8_t* dstp = P.pyr + (long long)frame * P.plane_frame_bytes + g.plane_off + x4;
    pyr_rows(g, xt, yt, reinterpret_cast<const uint8_t*>(s_t) - sxa, PYR_SDW * 4, sya, x4, y0 + threadIdx.y * PYR_RPT, dstp);
}

// The same tile, four pixels at a time from dword reads (levels whose 4-pixel groups span <= 8 source bytes, i.e. every scale
// factor below ~1.6; the host decides and fills the PyrGroup table).  Per staged source row a thread reads the three dwords
// around its first source pixel, funnel-shifts them so that byte 0 is that pixel (2 v_alignbyte), picks each output pixel's two
// taps as u16 lanes (v_perm, selector from the table: the right-border clamp of the second tap is already in it) and multiplies
// them with the packed coefficient pair in one v_dot2_u32_u16: 18 VALU per row of four horizontal results against 4 LDS byte
// reads + 2 multiply-adds per result before.  A source row shared with the previous output row is not recomputed (wave-uniform
// test: all lanes of a wave work on the same output rows).
__device__ __forceinline__ void pyr_hrow_from(unsigned w0, unsigned w1, unsigned w2, unsigned sh, const PyrGroup& G, unsigned (&hs)[4])
{
    const unsigned v0 = __builtin_amdgcn_alignbyte(w1, w0, sh), v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
    for (int k = 0; k < 4; k++)
        hs[k] = __builtin_amdgcn_udot2((pku16)__builtin_amdgcn_perm(v1, v0, G.sel[k]), (pku16)G.coef[k], 0u, false) >> 4;
}
__device__ __forceinline__ void pyr_hrow(const unsigned* rowp, unsigned sh, const PyrGroup& G, unsigned (&hs)[4])
{
    pyr_hrow_from(rowp[0], rowp[1], rowp[2], sh, G, hs);
}
// Three consecutive LDS dwords, requested without the compiler's knowledge.  In k_pyramid_level_g these reads happen while the NEXT tile's LDS-DMA is in
// flight; as ordinary loads the compiler guards every one of them with `s_waitcnt vmcnt(0)` (an LDS read "may alias" a pending LDS-DMA write: the DMA's LDS
// side carries no address the compiler could compare - neither distinct __shared__ arrays nor __restrict__ parameters change that) - i.e. it waits for the
// prefetch itself.  The device pass therefore spells the reads and their wait out (lds_read3_issue ... lds_read_wait: the wait names the registers so that
// nothing that uses them moves in front of it); the tile being read was waited for explicitly (s_waitcnt vmcnt(n) + barrier) before.  Other passes: plain loads.
struct Lds3 { unsigned long long w01; unsigned w2; };
__device__ __forceinline__ Lds3 lds_read3_issue(const unsigned* p)
{
    Lds3 r;
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
    const unsigned a = (unsigned)(unsigned long long)(orbhip_lptr)p;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8" : "=&v"(r.w01), "=&v"(r.w2) : "v"(a));      // (early clobber: the address register is read twice)
#else
    r.w01 = (unsigned long long)p[0] | ((unsigned long long)p[1] << 32); r.w2 = p[2];
#endif
    return r;
}
__device__ __forceinline__ void lds_read_wait(Lds3& a, Lds3& b)
{
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.w01), "+v"(a.w2), "+v"(b.w01), "+v"(b.w2));
#else
    (void)a; (void)b;
#endif
}
struct PyrTile { int x0, y0, sxa, sya; };
// one 256 x 16 tile of level `level` from its staged source footprint (lane tx, wave ty of the tile's four)
__device__ __forceinline__ void pyr_tile_compute(const ExtractParams& P, const LevelGeom& g, const PyrTile& T, int frame, int tx, int ty, const unsigned* s_t, const PyrGroup& G)
{   // G = P.xgrp[g.xgrp_off + (x4 >> 2)], loaded by the caller (the same for every tile of a column of tiles)
    const int x4 = T.x0 + tx * 4;
    if (x4 >= g.w) return;
    const int2* yt = P.ytab + g.ytab_off;
    const unsigned sh = (unsigned)G.sx0 & 3u;                      // sxa is a multiple of 4
    const unsigned* colp = s_t + ((G.sx0 - T.sxa) >> 2);
    const unsigned omask = x4 + 4 <= g.w ? 0xffffffffu : (0xffffffffu >> (8 * (x4 + 4 - g.w)));
    uint8_t* dstp = P.pyr + (long long)frame * P.plane_frame_bytes + g.plane_off + x4;
    const int ytop = T.y0 + ty * PYR_RPT;
    unsigned hp[4] = {0, 0, 0, 0}; int prow = -1;                  // horizontal results of staged row prow
    Lds3 wa = {0, 0}, wb = {0, 0};
#pragma unroll
    for (int rr = 0; rr < PYR_RPT; rr++) {
        const int y = ytop + rr;
        if (y >= g.h) break;
        int2 ye; ye.x = scalar_load(&yt[y].x); ye.y = scalar_load(&yt[y].y);            // (y is the same for the whole wave)
        const unsigned b0s = (unsigned)ye.y << 16, b1s = (unsigned)ye.y & 0xffff0000u;    // coefficients << 16: (b * t) >> 16 == mul_hi(b << 16, t)
        const int ra = min(max(ye.x, 0), g.src_h - 1) - T.sya, rb = min(max(ye.x + 1, 0), g.src_h - 1) - T.sya;
        unsigned h0[4], h1[4];
        if (ra != prow) wa = lds_read3_issue(colp + ra * PYR_SDW);
        if (rb != ra) wb = lds_read3_issue(colp + rb * PYR_SDW);
        lds_read_wait(wa, wb);
        if (ra == prow) { h0[0] = hp[0]; h0[1] = hp[1]; h0[2] = hp[2]; h0[3] = hp[3]; }
        else pyr_hrow_from((unsigned)wa.w01, (unsigned)(wa.w01 >> 32), wa.w2, sh, G, h0);
        if (rb == ra) { h1[0] = h0[0]; h1[1] = h0[1]; h1[2] = h0[2]; h1[3] = h0[3]; }
        else pyr_hrow_from((unsigned)wb.w01, (unsigned)(wb.w01 >> 32), wb.w2, sh, G, h1);
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) out |= ((__umulhi(b0s, h0[k]) + __umulhi(b1s, h1[k]) + 2u) >> 2) << (8 * k);
        *reinterpret_cast<unsigned*>(dstp + (long long)y * g.pitch) = out & omask;
        hp[0] = h1[0]; hp[1] = h1[1]; hp[2] = h1[2]; hp[3] = h1[3]; prow = rb;
    }
}
// A workgroup barrier for phases that exchange LDS data only: __syncthreads() is a fence over ALL memory, i.e. `s_waitcnt vmcnt(0)` in front of the barrier -
// it waits for the wave's global stores and, worse, for every LDS-DMA load still in flight (the NEXT tile's prefetch).
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The same for a phase whose LDS accesses are all spelled out (lds_read3_issue, pyr_fix_ragged): the wave's own LDS requests, then the barrier - as one
// statement the compiler neither looks into nor moves memory accesses across.  (With lds_barrier() the tile loop's FIRST barrier still carried vmcnt(0).)
__device__ __forceinline__ void lds_barrier_spelled()
{
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    lds_barrier();
#endif
}
// LDS-DMA staging of a tile's source footprint, so that a workgroup can have the NEXT tile's rows in flight while it computes this one
// (the loads never touch a VGPR): wave w brings rows w, w+4, ... (clamped to the footprint's last row), always 6 rows x 2 instructions
// (dwords 0..63 and 64..83 of the LDS row), so "everything but the newest 12" is an exact s_waitcnt for the tile before.  A lane whose dword
// is not entirely inside the source row reads the last one that is (the DMA takes any byte address); pyr_fix_ragged then shifts the row's ragged last dword into place.
struct PyrSrc { const uint8_t* src; int spitch, ndw, nrows, dfull; };
__device__ __forceinline__ PyrTile pyr_dma_issue(const ExtractParams& P, const LevelGeom& g, int tile_x, int tile_y, int tid, unsigned* s_t, PyrSrc& S)
{   // S.src, S.spitch: the source level's plane of this frame (the same for every tile of the workgroup: set by the kernel)
    PyrTile T; T.x0 = tile_x * PYR_TW; T.y0 = tile_y * PYR_TH;
    const int2* xt = P.xtab + g.xtab_off; const int2* yt = P.ytab + g.ytab_off;
    const int xl = min(T.x0 + PYR_TW - 1, g.w - 1), yl = min(T.y0 + PYR_TH - 1, g.h - 1);
    // the table entries are the same for every lane: scalar loads, so that the footprint and every row address below are scalar arithmetic
    // (as vector loads they made each row's address a 64-bit VALU multiply-add plus two v_readfirstlane: 65 VALU per tile, PMC round 3) and
    // nothing here waits on vmcnt (round 5 read them as vector loads: `s_waitcnt vmcnt(0)` in front of every tile's requests - the tile before's stores)
    const int e_x0 = scalar_load(&xt[T.x0].x), e_xl = scalar_load(&xt[xl].x);
    const int e_y0 = scalar_load(&yt[T.y0].x), e_yl = scalar_load(&yt[yl].x);
    T.sxa = e_x0 & ~3; const int sxb = min(e_xl + 1, g.src_w - 1);
    T.sya = min(max(e_y0, 0), g.src_h - 1); const int syb = min(max(e_yl + 1, 0), g.src_h - 1);
    S.ndw = ((sxb - T.sxa) >> 2) + 1; S.nrows = syb - T.sya + 1; S.dfull = (g.src_w - T.sxa) >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int omax = g.src_w - 4 - T.sxa;                              // byte offset of the last dword that ends inside the row (negative when fewer than 4 columns remain: still in the row)
    const int off0 = min(4 * lane, omax), off1 = min(4 * (lane + 64), omax);
    const ORBHIP_GLOBAL uint8_t* plane = uniform_ptr(S.src);
    const unsigned spitch = (unsigned)__builtin_amdgcn_readfirstlane(S.spitch);
#pragma unroll
    for (int j = 0; j < PYR_SROWS / 4; j++) {
        const int r = wave + 4 * j;
        const unsigned roff = (unsigned)(T.sya + min(r, S.nrows - 1)) * spitch + (unsigned)T.sxa;      // scalar; a level plane is far below 2^31 bytes
        lds_dma_dword(plane + (roff + (unsigned)off0), reinterpret_cast<uint8_t*>(s_t + r * PYR_SDW));
        if (lane < PYR_SDW - 64) lds_dma_dword(plane + (roff + (unsigned)off1), reinterpret_cast<uint8_t*>(s_t + r * PYR_SDW + 64));
    }
    return T;
}
// The ragged last dword of the footprint's rows (source widths that are not multiples of 4).  Its DMA lane read the four bytes that END at the row's end,
// so LDS dword `dfull` holds the row's last `part` bytes in its TOP bytes: shifted down here (the bytes past the row's end become zero, as staging by
// registers makes them).  Every wave fixes the rows it requested itself, right after its own wait: no barrier in between.  (Round 5 brought these
// bytes with three byte loads per tile, carried in registers across the tile loop: registers with a load pending are what the compiler waits for with vmcnt(0).)
__device__ __forceinline__ void pyr_fix_ragged(const LevelGeom& g, const PyrSrc& S, int tid, unsigned* s_t)
{
    const int part = g.src_w & 3;
    if (part == 0 || S.dfull >= S.ndw) return;                         // wave-uniform
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = wave + 4 * lane;
    if (lane < PYR_SROWS / 4) {
        unsigned* p = s_t + r * PYR_SDW + S.dfull;
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
        // (as ordinary LDS accesses these would carry `s_waitcnt vmcnt(0)` for the NEXT tile's LDS-DMA in flight: see lds_read3_issue)
        const unsigned a = (unsigned)(unsigned long long)(orbhip_lptr)p, sh = 8u * (unsigned)(4 - part);
        unsigned w;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_lshrrev_b32 %0, %2, %0\n\tds_write_b32 %1, %0" : "=&v"(w) : "v"(a), "s"(sh) : "memory");
#else
        *p >>= 8 * (4 - part);
#endif
    }
}
// One tile of k_pyramid_level_g: the NEXT tile's requests go out first, then everything but those is waited for (loads return in order; this tile's stores
// only make the count more conservative), then the tile is computed from `cur` while `nxt` fills.  Round 5's form of this loop spent 62 % of its wave
// cycles waiting (2.4 TB/s) because every wait in it was `s_waitcnt vmcnt(0)` - for the prefetch just issued and for the tile's own stores:
//  * __syncthreads() is a fence over all memory: vmcnt(0) in front of the barrier.  lds_barrier() waits for LDS traffic only;
//  * every ordinary LDS access while an LDS-DMA is pending gets a vmcnt(0) from the compiler (lds_read3_issue): the accesses of the tile loop are spelled out;
//  * the row tables were vector loads (scalar_load) and the ragged bytes were carried in registers (pyr_fix_ragged).
// (profiles/r06_exp_pyramid_lds_barriers.txt)
struct PyrStep { PyrTile T; PyrSrc S; };
__device__ __forceinline__ void pyr_step(const ExtractParams& P, const LevelGeom& g, int tx, int tyy, int tyend, int frame, int tid, int lane_x, int wv, const PyrGroup& G,
                                         unsigned* cur, unsigned* nxt, PyrStep& st)
{
    const bool more = tyy + 1 < tyend;
    PyrStep nx = st;
    if (more) nx.T = pyr_dma_issue(P, g, tx, tyy + 1, tid, nxt, nx.S);
    if (more) __builtin_amdgcn_s_waitcnt(ORBHIP_VMCNT(2 * (PYR_SROWS / 4))); else lds_dma_wait();
    __builtin_amdgcn_wave_barrier();
    pyr_fix_ragged(g, st.S, tid, cur);
    lds_barrier_spelled();
    pyr_tile_compute(P, g, st.T, frame, lane_x, wv, cur, G);
    lds_barrier_spelled();
    st = nx;
}
// nt = tiles (one below the other) per workgroup: 2, or 4 from 384 frames on
__global__ __launch_bounds__(256) void k_pyramid_level_g(ExtractParams P, int level, int nt)
{
    const LevelGeom g = P.geom[level];
    const int gx = (g.w + PYR_TW - 1) / PYR_TW, gy = (g.h + PYR_TH - 1) / PYR_TH, gyn = (gy + nt - 1) / nt;
    int tile, frame;
    if (!xcd_frame_map(gx * gyn, P.nframes, tile, frame)) return;
    frame += P.frame0;
    __shared__ unsigned s_t0[PYR_SROWS * PYR_SDW], s_t1[PYR_SROWS * PYR_SDW];
    const int tid = threadIdx.y * 64 + threadIdx.x, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const int tx = tile % gx;
    int ty = (tile / gx) * nt;
    const int tyend = min(ty + nt, gy);
    PyrStep st;
    st.S.src = level_src(P, frame, level - 1, st.S.spitch);
    st.T = pyr_dma_issue(P, g, tx, ty, tid, s_t0, st.S);
    const PyrGroup G = P.xgrp[g.xgrp_off + min(tx * (PYR_TW / 4) + (int)threadIdx.x, (g.w - 1) >> 2)];      // this lane's four columns: the same in every tile below
    for (; ty < tyend; ty += 2) {                                      // (two steps per trip: the buffers swap roles with constant addresses)
        pyr_step(P, g, tx, ty, tyend, frame, tid, threadIdx.x, wv, G, s_t0, s_t1, st);
        if (ty + 1 < tyend) pyr_step(P, g, tx, ty + 1, tyend, frame, tid, threadIdx.x, wv, G, s_t1, s_t0, st);
    }
}

// Same arithmetic straight from global memory: used for scale factors whose source footprint does not fit the LDS tile
// (the host decides per level from the coefficient tables).
__global__ __launch_bounds__(256) void k_pyramid_level_direct(ExtractParams P, int level)
{
    const LevelGeom g = P.geom[level];
    const int gx = (g.w + PYR_TW - 1) / PYR_TW, gy = (g.h + PYR_TH - 1) / PYR_TH;
    int tile, frame;
    if (!xcd_frame_map(gx * gy, P.nframes, tile, frame)) return;
    frame += P.frame0;
    const int x4 = (tile % gx) * PYR_TW + threadIdx.x * 4, ytop = (tile / gx) * PYR_TH + threadIdx.y * PYR_RPT;
    if (x4 >= g.w) return;
    int spitch; const uint8_t* src = level_src(P, frame, level - 1, spitch);
    pyr_rows(g, P.xtab + g.xtab_off, P.ytab + g.ytab_off, src, spitch, 0, x4, ytop, P.pyr + (long long)frame * P.plane_frame_bytes + g.plane_off + x4);
}

// A handful of frames (the drop-in's single-image call): ALL levels in one launch.  Seven dependent launches cost a single frame 34 us for ~8 us of
// arithmetic (2.3 us per dependent launch, then every level starts with a cold L2 and its own chain of load latencies).  Here a workgroup owns one
// small tile of the LAST level and computes, level by level, exactly the rectangle of every level that its tile descends from: the level-0 rectangle
// is staged in LDS, level l's rectangle is computed from level l-1's in LDS and written both to LDS (for level l+1) and to the level's plane.  The
// rectangles of neighbouring workgroups overlap (a halo that grows towards level 0): those pixels are computed more than once and stored more than
// once - the same integer arithmetic on the same sources, so every store of a pixel carries the same value.  The host makes the rectangles cover
// every level completely (also where the next level does not reference a level's last pixels) and aligns their columns to 4.
// Arithmetic: k_pyramid_level_g's (cv::resize INTER_LINEAR, OpenCV 3.2 fixed point), from the same PyrGroup / row tables.
// Memory latency is what a single frame's kernels consist of (every launch starts with a cold L2), so the kernel reads memory in TWO rounds: (1) lane l of
// the first wave fetches level l's constants and the workgroup's rectangle of level l; (2) every thread requests its share of ALL PyrGroup / row-table
// entries and of the level-0 rectangle before it waits for any of them.  After that the seven levels run from LDS, four pixels at a time like
// k_pyramid_level_g (pyr_hrow), behind barriers that wait for LDS only (__syncthreads would also wait for the level's stores to reach memory: 0.5 us a level).
__device__ __forceinline__ int qt_wave_incl_scan(int v, int lane);      // (DPP prefix sum, defined with the quadtree)
#define PC_T 256            // threads.  (1024 - four waves per SIMD to hide the LDS latencies - was slower, 22.6 against 21.3 us: a big workgroup starts and synchronises
                            // slowly; four rows of a group column per step with all their LDS reads requested together measured the same as one row per step)
#define PC_GIT 3            // per thread: 16-byte pieces of PyrGroup entries (<= 256 entries), row-table entries (<= 512), dwords of the level-0 rectangle (<= 4096); the host checks
#define PC_YIT 2
#define PC_RIT 16
struct PcLevel { int w, src_h, pitch, plane_off, xgrp_off, ytab_off, goff, yoff, x0, x1, y0, y1; unsigned rcp; };
__global__ __launch_bounds__(PC_T) void k_pyramid_cascade(ExtractParams P)
{
    HIP_DYNAMIC_SHARED(unsigned char, pc_lds)
    __shared__ PcLevel s_lv[ORBHIP_MAX_LEVELS];
    int tile, frame;
    if (!xcd_frame_map(P.pc_ntx * P.pc_nty, P.nframes, tile, frame)) return;
    frame += P.frame0;
    const int tid = (int)threadIdx.x, L = P.nlevels;
    const int ty = tile / P.pc_ntx, tx = tile - ty * P.pc_ntx;
    if (tid < 64) {         // ---- round 1 (one wave): lane l = level l; the table offsets are prefix sums over the lanes
        const int l = min(tid, L - 1);
        const LevelGeom g = P.geom[l];
        const short2 a = P.pc_xr[l * P.pc_ntx + tx], b = P.pc_yr[l * P.pc_nty + ty];
        const int ng = ((a.y - a.x) >> 2) + 1;
        const int nx = (tid >= 1 && tid < L) ? ng : 0, ny = (tid >= 1 && tid < L) ? b.y - b.x + 1 : 0;
        const int go = qt_wave_incl_scan(nx, tid) - nx, yo = qt_wave_incl_scan(ny, tid) - ny;
        if (tid < L) s_lv[tid] = PcLevel{g.w, g.src_h, g.pitch, g.plane_off, g.xgrp_off, g.ytab_off, go, yo, a.x, a.y, b.x, b.y, 0xffffffffu / (unsigned)ng + 1u};
        if (tid == L - 1) { s_lv[0].goff = go + nx; s_lv[0].yoff = yo + ny; }      // (level 0 has no table: its slots carry the totals)
    }
    __syncthreads();
    // (the two rectangle buffers are addressed as pc_lds + offset, never through an array of pointers: that made them generic pointers and every row read a
    //  flat_load, which waits for the level's outstanding global stores - 5 us for level 1 instead of 1)
    uint4* const s_grp = reinterpret_cast<uint4*>(pc_lds + P.pc_buf0 + P.pc_buf1);       // PyrGroup entries of levels 1 .. L-1 (three 16-byte pieces each)
    int2* const s_yt = reinterpret_cast<int2*>(s_grp + 3 * P.pc_xcap);
    {   // ---- round 2: all requests, then all LDS stores
        const int gtot = s_lv[0].goff, ytot = s_lv[0].yoff;
        int goffs[ORBHIP_MAX_LEVELS], yoffs[ORBHIP_MAX_LEVELS];
#pragma unroll
        for (int l = 0; l < ORBHIP_MAX_LEVELS; l++) { goffs[l] = s_lv[min(l, L - 1)].goff; yoffs[l] = s_lv[min(l, L - 1)].yoff; }
        uint4 gv[PC_GIT]; int2 yv[PC_YIT]; unsigned rv[PC_RIT];
#pragma unroll
        for (int k = 0; k < PC_GIT; k++) {      // (entries past the end re-read the last one: every load is unconditional, nothing waits before the last request is out)
            const int i = min(tid + PC_T * k, 3 * gtot - 1), e = i / 3, piece = i - 3 * e;
            int l = 1;
#pragma unroll
            for (int q = 2; q < ORBHIP_MAX_LEVELS; q++) if (q < L && e >= goffs[q]) l = q;
            const PcLevel& v = s_lv[l];
            gv[k] = reinterpret_cast<const uint4*>(P.xgrp + v.xgrp_off + (v.x0 >> 2) + (e - v.goff))[piece];
        }
#pragma unroll
        for (int k = 0; k < PC_YIT; k++) {
            const int i = min(tid + PC_T * k, ytot - 1);
            int l = 1;
#pragma unroll
            for (int q = 2; q < ORBHIP_MAX_LEVELS; q++) if (q < L && i >= yoffs[q]) l = q;
            const PcLevel& v = s_lv[l];
            yv[k] = P.ytab[v.ytab_off + v.y0 + i - v.yoff];
        }
        const PcLevel R = s_lv[0];
        int spitch; const uint8_t* src = level_src(P, frame, 0, spitch);
        const int ng = ((R.x1 - R.x0) >> 2) + 1, n0 = ng * (R.y1 - R.y0 + 1);
#pragma unroll
        for (int k = 0; k < PC_RIT; k++) {
            const int i = min(tid + PC_T * k, n0 - 1);
            const int r = ng == 1 ? i : (int)__umulhi((unsigned)i, R.rcp), g4 = i - r * ng, x = R.x0 + 4 * g4;       // (ceil(2^32 / 1) does not fit)
            const uint8_t* p = src + (long long)(R.y0 + r) * spitch + x;
            unsigned w;
            if (x + 3 < R.w) w = *reinterpret_cast<const u32_unaligned*>(p);
            else { w = 0; for (int q = 0; q < 4; q++) if (x + q < R.w) w |= (unsigned)p[q] << (8 * q); }
            rv[k] = w;
        }
#pragma unroll
        for (int k = 0; k < PC_GIT; k++) if (tid + PC_T * k < 3 * gtot) s_grp[tid + PC_T * k] = gv[k];
#pragma unroll
        for (int k = 0; k < PC_YIT; k++) if (tid + PC_T * k < ytot) s_yt[tid + PC_T * k] = yv[k];
#pragma unroll
        for (int k = 0; k < PC_RIT; k++) if (tid + PC_T * k < n0) reinterpret_cast<unsigned*>(pc_lds)[tid + PC_T * k] = rv[k];
    }
    __syncthreads();
    for (int l = 1; l < L; l++) {
        const PcLevel R = s_lv[l], S = s_lv[l - 1];
        const int ng = ((R.x1 - R.x0) >> 2) + 1, rows = R.y1 - R.y0 + 1, spd = ((S.x1 - S.x0) >> 2) + 1;
        const unsigned* sb = reinterpret_cast<const unsigned*>(pc_lds + ((l & 1) ? 0 : P.pc_buf0));      // source rectangle: dword (x - S.x0) / 4 of row y - S.y0 at sb[(y - S.y0) * spd + ...]
        unsigned* db = reinterpret_cast<unsigned*>(pc_lds + ((l & 1) ? P.pc_buf0 : 0));
        const PyrGroup* grp = reinterpret_cast<const PyrGroup*>(s_grp) + R.goff; const int2* yt = s_yt + R.yoff;
        uint8_t* gdst = P.pyr + (long long)frame * P.plane_frame_bytes + R.plane_off;
        const bool keep = l + 1 < L;                                       // the last level has no reader in here
        for (int i = tid; i < ng * rows; i += PC_T) {
            const int gy = ng == 1 ? i : (int)__umulhi((unsigned)i, R.rcp), g4 = i - gy * ng, y = R.y0 + gy, x4 = R.x0 + 4 * g4;
            const PyrGroup G = grp[g4];
            const int2 ye = yt[gy];
            const unsigned b0s = (unsigned)ye.y << 16, b1s = (unsigned)ye.y & 0xffff0000u;
            const int ra = min(max(ye.x, 0), R.src_h - 1) - S.y0, rb = min(max(ye.x + 1, 0), R.src_h - 1) - S.y0;
            const unsigned sh = (unsigned)G.sx0 & 3u;
            const unsigned* colp = sb + ((G.sx0 - S.x0) >> 2);             // S.x0 is a multiple of 4
            unsigned h0[4], h1[4];
            pyr_hrow(colp + ra * spd, sh, G, h0);
            pyr_hrow(colp + rb * spd, sh, G, h1);
            unsigned out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) out |= ((__umulhi(b0s, h0[k]) + __umulhi(b1s, h1[k]) + 2u) >> 2) << (8 * k);
            out &= x4 + 4 <= R.w ? 0xffffffffu : (0xffffffffu >> (8 * (x4 + 4 - R.w)));
            if (keep) db[i] = out;
            *reinterpret_cast<unsigned*>(gdst + (long long)y * R.pitch + x4) = out;
        }
        lds_barrier();
    }
}
void orbhip_launch_pyramid_cascade(const ExtractParams& P, int nframes, hipStream_t s)
{
    ExtractParams Q = P; Q.nframes = nframes;
    const size_t lds = (size_t)P.pc_buf0 + P.pc_buf1 + (size_t)P.pc_xcap * sizeof(PyrGroup) + (size_t)P.pc_ycap * sizeof(int2);
    hipLaunchKernelGGL(k_pyramid_cascade, dim3(xcd_grid(P.pc_ntx * P.pc_nty, nframes), 1, 1), dim3(PC_T, 1, 1), lds, s, Q);
}

bool orbhip_pyramid_tile_fits(int src_cols_per_tile, int src_rows_per_tile) { return ((src_cols_per_tile + 3) >> 2) + 1 <= PYR_SDW && src_rows_per_tile <= PYR_SROWS; }
int orbhip_pyramid_tile_dwords() { return PYR_SDW; }
int orbhip_blur_mfma_tile_w() { return BM_COLS * BM_BLOCKS; }
int orbhip_blur_mfma_tile_h() { return BM_ROWS; }
int orbhip_pyramid_tile_w() { return PYR_TW; }
int orbhip_pyramid_tile_h() { return PYR_TH; }

void orbhip_launch_pyramid_level(const ExtractParams& P, int level, int w, int h, int mode, int nframes, hipStream_t s)
{   // mode 2: staged + 4-pixel groups, 1: staged, 0: straight from global memory
    ExtractParams Q = P; Q.nframes = nframes;
    dim3 grid(xcd_grid(((w + PYR_TW - 1) / PYR_TW) * ((h + PYR_TH - 1) / PYR_TH), nframes), 1, 1), block(64, 4, 1);
    static const int nt_env = getenv("ORBHIP_PYR_NT") ? atoi(getenv("ORBHIP_PYR_NT")) : 0;
    const int nt = nt_env > 0 ? nt_env : nframes >= 384 ? 4 : nframes <= 8 ? 1 : 2;          // measured (round 3, same call): B = 512: 181.7 k frames/s with 2, 183.3 k with 4, 181.8 k with 8, 180.3 k with 1; B = 128: 157.6 k with 2, 155.2 k with 4
    const int gyn = (((h + PYR_TH - 1) / PYR_TH) + nt - 1) / nt;
    if (mode == 2) hipLaunchKernelGGL(k_pyramid_level_g, dim3(xcd_grid(((w + PYR_TW - 1) / PYR_TW) * gyn, nframes), 1, 1), block, 0, s, Q, level, nt);
    else if (mode == 1) hipLaunchKernelGGL(k_pyramid_level, grid, block, 0, s, Q, level);
    else hipLaunchKernelGGL(k_pyramid_level_direct, grid, block, 0, s, Q, level);
}

// ------------------------------------------------------------------------------------------------ colour -> gray
// cv::cvtColor(RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) on 8U as Tracking::GrabImage* applies it before the
// extractor sees the frame (Tracking.cc:172-198, 217-229, 248-260): gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14
// (OpenCV 3.2 RGB2Gray<uchar>, 14-bit fixed point; alpha ignored).  One thread converts 4 pixels: 3 or 4 aligned
// 32-bit loads, one 32-bit store; the byte path serves unaligned sources and the ragged end of a row.
struct GrayParams {
    const uint8_t* src; long long src_frame_stride; int src_row_stride;
    uint8_t* dst; long long dst_frame_stride; int dst_pitch;
    int w, h, channels, k0, k2;         // k0 weighs channel 0, k2 channel 2 (4899/1868 for RGB order, swapped for BGR)
};
__device__ __forceinline__ uint32_t gray14(uint32_t c0, uint32_t c1, uint32_t c2, int k0, int k2)
{
    return (c0 * (uint32_t)k0 + c1 * 9617u + c2 * (uint32_t)k2 + 8192u) >> 14;
}
__global__ __launch_bounds__(256) void k_to_gray(GrayParams G)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
    if (x >= G.w) return;
    const uint8_t* srow = G.src + (long long)f * G.src_frame_stride + (long long)y * G.src_row_stride + (long long)x * G.channels;
    uint8_t* drow = G.dst + (long long)f * G.dst_frame_stride + (long long)y * G.dst_pitch + x;
    const bool whole = x + 4 <= G.w, aligned = (((uintptr_t)srow) & 3) == 0;
    if (whole && aligned) {
        const uint32_t* s4 = (const uint32_t*)srow;
        uint32_t g0, g1, g2, g3;
        if (G.channels == 3) {
            const uint32_t a = s4[0], b = s4[1], c = s4[2];          // bytes 0..11 = p0.c0 p0.c1 p0.c2 p1.c0 ...
            g0 = gray14(a & 255, (a >> 8) & 255, (a >> 16) & 255, G.k0, G.k2);
            g1 = gray14(a >> 24, b & 255, (b >> 8) & 255, G.k0, G.k2);
            g2 = gray14((b >> 16) & 255, b >> 24, c & 255, G.k0, G.k2);
            g3 = gray14((c >> 8) & 255, (c >> 16) & 255, c >> 24, G.k0, G.k2);
        } else {
            const uint32_t a = s4[0], b = s4[1], c = s4[2], d = s4[3];
            g0 = gray14(a & 255, (a >> 8) & 255, (a >> 16) & 255, G.k0, G.k2);
            g1 = gray14(b & 255, (b >> 8) & 255, (b >> 16) & 255, G.k0, G.k2);
            g2 = gray14(c & 255, (c >> 8) & 255, (c >> 16) & 255, G.k0, G.k2);
            g3 = gray14(d & 255, (d >> 8) & 255, (d >> 16) & 255, G.k0, G.k2);
        }
        *(uint32_t*)drow = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        const int n = min(4, G.w - x);
        for (int i = 0; i < n; i++) {
            const uint8_t* p = srow + i * G.channels;
            drow[i] = (uint8_t)gray14(p[0], p[1], p[2], G.k0, G.k2);
        }
    }
}
void orbhip_launch_to_gray(const uint8_t* src, long long src_frame_stride, int src_row_stride, uint8_t* dst, long long dst_frame_stride,
                           int dst_pitch, int w, int h, int channels, bool rgb_order, int nframes, hipStream_t s)
{
    GrayParams G{src, src_frame_stride, src_row_stride, dst, dst_frame_stride, dst_pitch, w, h, channels,
                 rgb_order ? 4899 : 1868, rgb_order ? 1868 : 4899};
    dim3 grid((w + 1023) / 1024, h, nframes), block(256, 1, 1);
    hipLaunchKernelGGL(k_to_gray, grid, block, 0, s, G);
}

// ------------------------------------------------------------------------------------------------ blur
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U: int kernel {k3,k2,k1,k0,k1,k2,k3} = {18,34,49,55,49,34,18}
// (computed on the host from getGaussianKernel + cvRound(k*256)), int32 row pass, column pass rounded with 16 bits.
#define BLUR_TW 128
#define BLUR_TH 32
#define BLUR_SROW 144                  // LDS source row stride in bytes (136 used: x0-4 .. x0+131)
__device__ __forceinline__ int reflect101_clamped(int p, int len)
{
    if (p < 0) p = -p; else if (p >= len) p = 2 * (len - 1) - p;
    return min(max(p, 0), len - 1);      // second clamp only touches halo positions no output reads
}
struct BlurK { float k0, k1, k2, k3; unsigned dot_lo, dot_hi; };
#define BLUR_SDW (BLUR_SROW / 4)       // 36 dwords per staged source row, 34 used
#define BLUR_SROWS (BLUR_TH + 6)
// 128x32 outputs per workgroup, 3 phases:
//  1. stage (32+6) x (128+8) source bytes: wave w owns rows w, w+4, ... (row address on the scalar unit, lane = dword),
//     ten 32-bit loads in flight per lane; byte-wise reflect only for lanes on the image border / unaligned sources;
//  2. horizontal pass, 4 pixels per thread-iteration: two v_dot4_u32_u8 per pixel on byte-aligned windows, result
//     (<= 255*257) converted to fp32 and kept in LDS as float4;
//  3. vertical pass, a 4-pixel x 4-row strip per thread in fp32 (add/mul/fma issue at twice the integer rate on this
//     part and every partial sum below 2^24 is exact; a final sum >= 2^24 means >= 256 and saturates either way),
//     rounding = floor(s/65536 + 0.5) (generic OpenCV) or round-half-even (x86 SSE2 build) folded into the
//     saturating round-to-nearest-even of v_cvt_pk_u8_f32; four coalesced 32-bit stores.
__global__ __launch_bounds__(256) void k_blur(ExtractParams P, BlurK K)
{
    int tile, frame;
    if (!xcd_frame_map(P.nblur_tiles, P.nframes, tile, frame)) return;
    frame += P.frame0;
    const TileDesc t = P.blur_tiles[tile];
    const int tid = threadIdx.x;
    const LevelGeom g = P.geom[t.level];
    int spitch; const uint8_t* src = level_src(P, frame, t.level, spitch);
    __shared__ unsigned s_src[BLUR_SROWS * BLUR_SDW];
    __shared__ float4 s_h[BLUR_SROWS * (BLUR_TW / 4)];
    const bool aligned = ((((unsigned long long)src) | (unsigned long long)spitch) & 3ull) == 0;
    {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int gx = t.x0 - 4 + 4 * lane;
        const bool act = lane < 34, edge = act && !(aligned && gx >= 0 && gx + 3 < g.w);
        unsigned v[10];
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const uint8_t* row = src + (long long)reflect101_clamped(t.y0 - 3 + wave + 4 * j, g.h) * spitch + (t.x0 - 4);   // wave-uniform -> SGPR base
            v[j] = (act && !edge) ? *reinterpret_cast<const unsigned*>(row + (unsigned)(4 * lane)) : 0u;
        }
        if (edge) {
            const int x0r = reflect101_clamped(gx, g.w), x1r = reflect101_clamped(gx + 1, g.w), x2r = reflect101_clamped(gx + 2, g.w), x3r = reflect101_clamped(gx + 3, g.w);
#pragma unroll
            for (int j = 0; j < 10; j++) {
                const uint8_t* row = src + (long long)reflect101_clamped(t.y0 - 3 + wave + 4 * j, g.h) * spitch;
                v[j] = (unsigned)row[x0r] | ((unsigned)row[x1r] << 8) | ((unsigned)row[x2r] << 16) | ((unsigned)row[x3r] << 24);
            }
        }
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int r = wave + 4 * j;
            if (act && r < BLUR_SROWS) s_src[r * BLUR_SDW + lane] = v[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < BLUR_SROWS * 32; i += 256) {
        const int r = i >> 5, q = i & 31;
        const unsigned* p = &s_src[r * BLUR_SDW + q];
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];           // tile bytes 4q .. 4q+11; output pixel 4q+k reads bytes 4q+k+1 .. 4q+k+7
        float4 o;
        o.x = (float)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), K.dot_lo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), K.dot_hi, 0u, false), false);
        o.y = (float)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K.dot_lo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), K.dot_hi, 0u, false), false);
        o.z = (float)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), K.dot_lo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), K.dot_hi, 0u, false), false);
        o.w = (float)__builtin_amdgcn_udot4(w1, K.dot_lo, __builtin_amdgcn_udot4(w2, K.dot_hi, 0u, false), false);
        s_h[r * (BLUR_TW / 4) + q] = o;
    }
    __syncthreads();
    const int q = tid & 31, rg = tid >> 5;
    const int gx = t.x0 + 4 * q;
    if (gx >= g.w) return;
    float h[10][4];
#pragma unroll
    for (int r = 0; r < 10; r++) { const float4 v = s_h[(4 * rg + r) * (BLUR_TW / 4) + q]; h[r][0] = v.x; h[r][1] = v.y; h[r][2] = v.z; h[r][3] = v.w; }
    const bool half_even = P.blur_round_mode == 1 && gx < (g.w & ~3);      // x86 SSE2 build of OpenCV: cvtps2dq on whole 4-column groups
    uint8_t* dstbase = P.blur + (long long)frame * P.plane_frame_bytes + g.plane_off + gx;
    auto strip = [&](auto he) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int gy = t.y0 + 4 * rg + r;
            if (gy >= g.h) break;
            unsigned out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float sum = K.k0 * h[r + 3][k];
                sum = __builtin_fmaf(K.k1, h[r + 2][k] + h[r + 4][k], sum);
                sum = __builtin_fmaf(K.k2, h[r + 1][k] + h[r + 5][k], sum);
                sum = __builtin_fmaf(K.k3, h[r][k] + h[r + 6][k], sum);
                // generic FixedPtCastEx: (sum + 2^15) >> 16, the saturating RNE conversion then sees an integer;
                // SSE2 build: the conversion's own round-half-even is the cvtps2dq
                const float v = decltype(he)::value ? sum * (1.0f / 65536.0f) : __builtin_floorf(__builtin_fmaf(sum, 1.0f / 65536.0f, 0.5f));
                out = __builtin_amdgcn_cvt_pk_u8_f32(v, (unsigned)k, out);
            }
            *reinterpret_cast<unsigned*>(dstbase + (long long)gy * g.pitch) = out;   // pitch multiple of 64, gx of 4: pad bytes absorb the tail
        }
    };
    if (half_even) strip(std::true_type{}); else strip(std::false_type{});
}

// ---- the same blur on the matrix cores (the default; k_blur above is the all-VALU form, ORBHIP_BLUR=valu: 0.36 ms against 0.23 ms
// for 256 KITTI frames).
// A departure from "no MFMA" that the instruction-rate table invites: the extraction kernels are bound by VALU issue, the matrix pipe
// is idle and issues beside it, and a 7-tap filter is a banded (Toeplitz) matrix product whose i8 form is EXACT here: pixels - 128 fit
// i8, the taps {18, 34, 49, 55} fit i8, accumulation is i32.  One workgroup = 26 rows x 224 columns; a wavefront owns two of its seven
// 32-column blocks:
//   stage:  the 32 x 256 source bytes behind the tile go to LDS by LDS-DMA, one whole row (256 contiguous bytes) per instruction - the
//           first version loaded the A operand straight from global memory, 16 bytes per lane from 32 different rows, and stored 8
//           bytes per row: the texture addresser was 94 % busy and the kernel slower than k_blur (PMC pass pm1);
//   rows:   lane (i, h) reads 16 staged bytes of row i from column 32c - 4 + 16h as the A operand (XOR 0x80 = -128);
//           D1[i][j] = sum_k A[i][k] * HB[k][j] is the horizontal pass of 32 rows x 32 output columns (two K = 32 blocks: columns
//           32c - 4 .. 32c + 27 and the seven behind them), + 128 * 257 it lies in [0, 65535] - exactly the row filter's int32 result;
//   cols:   D1 leaves the matrix core with lane = column and register = row, which is what the A operand of the second product needs
//           (lane = its row i' = column, bytes = its k = source row) once each value is split into a low and a high byte plane
//           (4 v_perm per 4 values); D2[i' = column][j' = output row] = sum_r VB[r][j'] * plane[r], for both planes, then
//           2^8 D2hi + D2lo (+ constant) is the column filter's int32 sum; rounding, saturation and packing as in k_blur (v_cvt_pk_u8_f32);
//   store:  D2 has lane = output row: the tile is transposed through LDS and leaves as whole rows (224 contiguous bytes per instruction).
// The band matrices (tap k - j - 1, zero elsewhere; VB also zero for the six incomplete output rows) come from a host table.  k is whatever
// (lane half, byte) pair the hardware pairs between A and B: the kernel never needs the nominal k order (tools/mfma_probe.hip checks it).
typedef int v4i __attribute__((vector_size(16)));
typedef int v2i __attribute__((vector_size(8)));
typedef int v16i __attribute__((vector_size(64)));
#define BM_IN_DW 68                     // dwords per staged source row: 64 loaded (columns x0 - 4 .. x0 + 251) + 4 of padding against bank conflicts
#define BM_OUT_DW 57                    // dwords per row of the output tile in LDS (56 used)
// ---- k_blur_mfma in four steps (shared by the one-tile and the pipelined form of the kernel)
struct BlurTile { TileDesc t; LevelGeom g; const uint8_t* src; int spitch; };
__device__ __forceinline__ BlurTile blur_tile(const ExtractParams& P, int frame, int tile)
{
    BlurTile T; T.t = P.blur_tiles[tile]; T.g = P.geom[T.t.level]; T.src = level_src(P, frame, T.t.level, T.spitch);
    return T;
}
// stage: wave w brings source rows 8w .. 8w+7 (BORDER_REFLECT_101 on the row index), lane = dword of the row.  Dwords that are not
// entirely inside the image re-read a dword that is (never used as loaded): the two border fix-ups write them.  8 LDS-DMA instructions per wave.
__device__ __forceinline__ void blur_issue(const BlurTile& T, int wave, int lane, unsigned* s_in)
{
    const LevelGeom& g = T.g; const int x0 = T.t.x0, y0 = T.t.y0, spitch = T.spitch; const uint8_t* src = T.src;
    const int w4 = g.w & ~3;
    const int c0 = x0 - 4 + 4 * lane;
    const unsigned coff = (unsigned)min(max(c0, 0), w4 - 4);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int row = 8 * wave + r;
        const ORBHIP_GLOBAL uint8_t* rp = uniform_ptr(src + (long long)reflect101_clamped(y0 - 3 + row, g.h) * spitch);
        lds_dma_dword(rp + coff, reinterpret_cast<uint8_t*>(s_in + row * BM_IN_DW));
    }
}
// after the tile's rows have landed (and a wave barrier): the dwords that straddle the image's left / right border
__device__ __forceinline__ void blur_fix(const BlurTile& T, int wave, int lane, unsigned* s_in)
{
    const LevelGeom& g = T.g; const int x0 = T.t.x0, y0 = T.t.y0, spitch = T.spitch; const uint8_t* src = T.src;
    const int w4 = g.w & ~3;
    const bool fix_left = x0 == 0, fix_right = x0 + 252 > w4;     // wave-uniform
    if (fix_left || fix_right) {
        // lane = (row of this wave, slot): slot 0 = the dword left of the image, slots 1 / 2 = the two dwords from column w & ~3 on (the last
        // 0..3 real columns, then reflected ones: a tile reads at most 3 columns past the last one it writes)
        const int rr = lane >> 3, slot = lane & 7, row = 8 * wave + rr;
        const int cfix = slot == 0 ? -4 : w4 + 4 * (slot - 1);
        const int dw = (cfix - (x0 - 4)) >> 2;
        const bool act = (slot == 0 ? fix_left : (slot <= 2 && fix_right)) && dw >= 0 && dw < 64;
        if (act) {
            const uint8_t* rowp = src + (long long)reflect101_clamped(y0 - 3 + row, g.h) * spitch;
            unsigned wv = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) wv |= (unsigned)rowp[reflect101_clamped(cfix + b, g.w)] << (8 * b);
            s_in[row * BM_IN_DW + dw] = wv;
        }
    }
}
__device__ __forceinline__ void blur_compute(const ExtractParams& P, const BlurTile& T, int wave, int lane, const unsigned* s_in, unsigned* s_out, const unsigned* s_band)
{
    const LevelGeom& g = T.g; const int x0 = T.t.x0;
    const int w4 = g.w & ~3;
    const int i = lane & 31, h = lane >> 5;
    const v4i HB1 = *reinterpret_cast<const v4i*>(s_band + 4 * lane), HB2 = *reinterpret_cast<const v4i*>(s_band + 256 + 4 * lane), VB = *reinterpret_cast<const v4i*>(s_band + 512 + 4 * lane);
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // C operand = inline constant 0: the -128 of the operand bytes is undone on the VALU
    // x86 SSE2 build of OpenCV: cvtps2dq (round-half-even) on whole 4-column groups below w & ~3, the generic rounding behind them
    const int he_limit = P.blur_round_mode == 1 ? w4 : 0;
    // sum = 2^8 (chi + 128 * 257) + (clo + 128 * 257) with chi / clo the raw products of the (byte - 128) planes: the constant joins the
    // scaling in one fma, exactly (sum < 2^24, K / 2^16 = 129 + 2^-8)
    const float kbias = (float)(257 * 128 * 257) * (1.0f / 65536.0f);
#pragma unroll 1
    for (int c = wave; c < BM_BLOCKS; c += 4) {
        const int xb = x0 + 32 * c;
        if (xb >= g.w) break;
        v4i A1 = *reinterpret_cast<const v4i*>(s_in + i * BM_IN_DW + 8 * c + 4 * h);
        const v2i a2 = *reinterpret_cast<const v2i*>(s_in + i * BM_IN_DW + 8 * c + 8);
        v4i A2 = {a2[0], a2[1], 0, 0};
#pragma unroll
        for (int d = 0; d < 4; d++) { A1[d] ^= (int)0x80808080u; A2[d] ^= (int)0x80808080u; }
        v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, HB1, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2, HB2, acc, 0, 0, 0);
        // row filter result = acc + 128 * 257 in [0, 65535]; its low / high byte planes: bytes 4g .. 4g+3 of a plane = registers 4g .. 4g+3
        v4i Alo, Ahi;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const unsigned r0 = (unsigned)acc[4 * gq] + 128u * 257u, r1 = (unsigned)acc[4 * gq + 1] + 128u * 257u;
            const unsigned r2 = (unsigned)acc[4 * gq + 2] + 128u * 257u, r3 = (unsigned)acc[4 * gq + 3] + 128u * 257u;
            const unsigned t01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);      // r0.b0 r1.b0 r0.b1 r1.b1
            const unsigned t23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
            Alo[gq] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
            Ahi[gq] = (int)(__builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u);
        }
        const v16i clo = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alo, VB, zero, 0, 0, 0);
        const v16i chi = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahi, VB, zero, 0, 0, 0);
        // lane = output row (lane & 31), registers = columns xb + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3)
        if (i < BM_ROWS) {
            const bool all_he = xb + 32 <= he_limit, none_he = xb >= he_limit;  // wave-uniform: one rounding for the whole block (all but the right-most)
            auto finish = [&](auto mode) {                               // mode 0: every group generic, 1: every group half-even, 2: per group
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int gx = xb + 8 * gq + 4 * h;
                    unsigned out = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float raw = (float)((chi[4 * gq + k] << 8) + clo[4 * gq + k]);           // |raw| < 2^24: exact
                        const float q = __builtin_fmaf(raw, 1.0f / 65536.0f, kbias);                   // column filter sum / 2^16, exact
                        // generic FixedPtCastEx: (sum + 2^15) >> 16, the saturating RNE conversion then sees an integer; SSE2 build: the conversion's own round-half-even
                        float v;
                        if (decltype(mode)::value == 1) v = q;
                        else if (decltype(mode)::value == 0) v = __builtin_floorf(q + 0.5f);
                        else v = gx < he_limit ? q : __builtin_floorf(q + 0.5f);
                        out = __builtin_amdgcn_cvt_pk_u8_f32(v, (unsigned)k, out);
                    }
                    s_out[i * BM_OUT_DW + 8 * c + 2 * gq + h] = out;
                }
            };
            if (all_he) finish(std::integral_constant<int, 1>{}); else if (none_he) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 2>{});
        }
    }
}
__device__ __forceinline__ void blur_store(const ExtractParams& P, const BlurTile& T, int frame, int wave, int lane, const unsigned* s_out)
{
    const LevelGeom& g = T.g; const int x0 = T.t.x0, y0 = T.t.y0;
    // ---- store: wave w writes output rows w, w + 4, ..., lane = dword of the row (224 contiguous bytes per instruction)
    uint8_t* dst = P.blur + (long long)frame * P.plane_frame_bytes + g.plane_off;
    const int gx = x0 + 4 * lane;
    if (lane < 8 * BM_BLOCKS && gx < g.w)
        for (int r = wave; r < BM_ROWS && y0 + r < g.h; r += 4)
            *reinterpret_cast<unsigned*>(dst + (long long)(y0 + r) * g.pitch + gx) = s_out[r * BM_OUT_DW + lane];      // pitch multiple of 64, gx of 4: pad bytes absorb the tail
}

// one tile of the matrix-core blur by the calling workgroup (256 threads): stage, patch, two banded products, round, store
__device__ __forceinline__ void blur_mfma_tile(const ExtractParams& P, int tile, int frame, unsigned* s_in, unsigned* s_out, unsigned* s_band)
{
#pragma unroll
    for (int k = 0; k < 3; k++) s_band[256 * k + threadIdx.x] = reinterpret_cast<const unsigned*>(P.blur_band)[256 * k + threadIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const BlurTile T = blur_tile(P, frame, tile);
    blur_issue(T, wave, lane, s_in);
    lds_dma_wait();
    __builtin_amdgcn_wave_barrier();                                  // every lane's dwords have landed before any lane patches one
    blur_fix(T, wave, lane, s_in);
    __syncthreads();
    blur_compute(P, T, wave, lane, s_in, s_out, s_band);
    __syncthreads();
    blur_store(P, T, frame, wave, lane, s_out);
}
__global__ __launch_bounds__(256, 5) void k_blur_mfma(ExtractParams P)
{
    __shared__ __attribute__((aligned(16))) unsigned s_in[32 * BM_IN_DW];
    __shared__ unsigned s_out[BM_ROWS * BM_OUT_DW];
    __shared__ __attribute__((aligned(16))) unsigned s_band[3 * 64 * 4];   // HB1 | HB2 | VB: fetched once per workgroup with coalesced 32-bit loads
    int tile, frame;                                                        // (as three 128-bit loads per lane they cost each wave 3 x 71 cycles of the texture addresser)
    if (!xcd_frame_map(P.nblur_tiles, P.nframes, tile, frame)) return;
    blur_mfma_tile(P, tile, frame + P.frame0, s_in, s_out, s_band);
}

void orbhip_launch_blur(const ExtractParams& P, const int gk[4], int nframes, hipStream_t s, int tile0, int ntiles)
{   // tiles [tile0, tile0 + ntiles) of the level-major tile list (ntiles < 0: all)
    BlurK K; K.k0 = (float)gk[0]; K.k1 = (float)gk[1]; K.k2 = (float)gk[2]; K.k3 = (float)gk[3];
    K.dot_lo = (unsigned)gk[3] | ((unsigned)gk[2] << 8) | ((unsigned)gk[1] << 16) | ((unsigned)gk[0] << 24);    // bytes x-3 .. x
    K.dot_hi = (unsigned)gk[1] | ((unsigned)gk[2] << 8) | ((unsigned)gk[3] << 16);                              // bytes x+1 .. x+3
    ExtractParams Q = P; Q.nframes = nframes;
    if (ntiles >= 0) { Q.blur_tiles = P.blur_tiles + tile0; Q.nblur_tiles = ntiles; }
    if (Q.nblur_tiles <= 0) return;
    if (P.blur_band) hipLaunchKernelGGL(k_blur_mfma, dim3(xcd_grid(Q.nblur_tiles, nframes), 1, 1), dim3(256, 1, 1), 0, s, Q);
    else hipLaunchKernelGGL(k_blur, dim3(xcd_grid(Q.nblur_tiles, nframes), 1, 1), dim3(256, 1, 1), 0, s, Q, K);
}

// ------------------------------------------------------------------------------------------------ FAST per cell
// One wavefront per 30-px grid cell (4 cells per workgroup, no workgroup barrier: every LDS region is wave-private).  The cell's
// sub-image is staged in LDS, then the two cv::FAST calls of the reference (ORBextractor.cc:809-816) are served in turn:
//   1. iniThFAST: a cheap NECESSARY test on every pixel pair (four opposite ring pairs must clear C +- t) -> ordered list of the pairs
//      that may hold a corner; the exact intrinsic score
//        S = max over the 16 nine-pixel arcs of min |centre - ring| margin (dark or bright) - 1     (cornerScore<16>)
//      ("corner at threshold t  <=>  S >= t") only for the listed pairs, one lane each; 3x3 non-max suppression over them (it sees only
//      scores of the same cell: neighbours outside the examined interior count as 0, exactly like FAST on the cell sub-image); survivors
//      emitted in row-major order;
//   2. only if that found nothing: the exact score of every examined pixel at minThFAST, corners compacted in row-major order, NMS, emission.
#define FC_WAVES 4

__device__ __forceinline__ pku16 pmin(pku16 a, pku16 b) { return a < b ? a : b; }
__device__ __forceinline__ pku16 pmax(pku16 a, pku16 b) { return a > b ? a : b; }

// bytes (I, I+1) of the 12-byte row {w0,w1,w2} as two zero-extended u16 lanes (one v_perm_b32)
template <int I> __device__ __forceinline__ pku16 row_pair(unsigned w0, unsigned w1, unsigned w2)
{
    unsigned r;
    if (I <= 6) r = __builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (unsigned)I | ((unsigned)(I + 1) << 16));
    else r = __builtin_amdgcn_perm(w2, w1, 0x0c000c00u | (unsigned)(I - 4) | ((unsigned)(I - 3) << 16));
    return (pku16)r;
}

// Intrinsic FAST score of two horizontally adjacent pixels.  With ring values R_k and centre C:
//   best dark margin   = max_arcs min_k (C - R_k) = C - min_arcs max_k R_k
//   best bright margin = max_arcs min_k (R_k - C) = max_arcs min_k R_k - C
// so only u8 min / max of ring values are needed (no per-element subtraction).  The 16 nine-element circular windows
// come from prefix / suffix minima of the two 8-element blocks: win[i] = op(suffix[i], prefix[(i+8)&15]).
// ONE polarity per pixel is enough: a 9-arc i..i+8 holds both elements of the opposite pair (i, i+8) and one element of every other
// opposite pair, so with m = max_k min(R_k, R_k+8): a pixel whose bright margin is positive has m > C, one whose dark margin is positive has
// every pair's minimum below C, m < C - and a margin that is not positive gives a score below any threshold >= 0, which the caller zeroes
// whatever its value.  Pixels with m <= C are complemented (R -> 255 - R, C -> 255 - C: their dark margin becomes the bright one), then only
// the arc minima are evaluated: 15 + 59 packed min / max and 17 full-rate xors instead of 118 (round 3).
__device__ __forceinline__ pki16 fast_score_pair(const pku16 r[16], pku16 c)
{
    pku16 m = pmin(r[0], r[8]);
#pragma unroll
    for (int k = 1; k < 8; k++) m = pmax(m, pmin(r[k], r[k + 8]));
    const pki16 nb = ~(((pki16)c - (pki16)m) >> 15);                 // all ones where m <= C (not a bright corner)
    const pku16 cm = (pku16)nb & (pku16){0x00ff, 0x00ff};
    pku16 q[16], pn[16], sn[16];
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = r[i] ^ cm;
#pragma unroll
    for (int b = 0; b < 16; b += 8) {
        pn[b] = q[b]; sn[b + 7] = q[b + 7];
#pragma unroll
        for (int i = 1; i < 8; i++) { pn[b + i] = pmin(pn[b + i - 1], q[b + i]); sn[b + 7 - i] = pmin(sn[b + 8 - i], q[b + 7 - i]); }
    }
    pku16 hi = pmin(sn[0], pn[8]);                                   // max over arcs of the arc minimum
#pragma unroll
    for (int i = 1; i < 16; i++) hi = pmax(hi, pmin(sn[i], pn[(i + 8) & 15]));
    const pki16 one = {1, 1};
    return (pki16)hi - (pki16)(c ^ cm) - one;
}

__host__ __device__ __forceinline__ int fc_wave_bytes(int pbytes, int sstride, int srows, int listcap)
{
    return ((pbytes + 15) & ~15) + ((sstride * srows + 4 + 15) & ~15) + ((2 * listcap + 15) & ~15);      // + 4: the dword behind the map's last row is part of it (see the score map's layout in k_fast_cells)
}

// ring of the pixel pair (Q, Q+1) of a 4-pixel group whose 7 x 12-byte window is w[7][3] (circle: FAST 16-point Bresenham);
// ring element k of pixel q sits at row 3+dy_k, byte 3+q+dx_k
// The same ring for a pair that starts at byte 3 + SH of the window rows (SH = 0 or 2, per lane): the byte pair (I, I+1), I = 3 + dx, sits in
// dwords {w0, w1} for I <= 4 and in {w1, w2} for I = 5, 6 whichever SH is, so every ring element is ONE v_perm_b32 whose selector carries the
// lane's shift (selector of byte pair (i, i+1) of a dword pair = 0x0c000c00 | i | (i+1) << 16; + 0x00020002 moves it two bytes on) - no
// funnel shift of the window rows first.  s[i] = that selector for i = 0..4; I = 5, 6 are i = 1, 2 of {w1, w2}.
__device__ __forceinline__ pku16 row_pair_v(unsigned lo, unsigned hi, unsigned sel) { return (pku16)__builtin_amdgcn_perm(hi, lo, sel); }
#define FC_RING_V(s) { \
    row_pair_v(w[6][0], w[6][1], s[3]), row_pair_v(w[6][0], w[6][1], s[4]), row_pair_v(w[5][1], w[5][2], s[1]), row_pair_v(w[4][1], w[4][2], s[2]), \
    row_pair_v(w[3][1], w[3][2], s[2]), row_pair_v(w[2][1], w[2][2], s[2]), row_pair_v(w[1][1], w[1][2], s[1]), row_pair_v(w[0][0], w[0][1], s[4]), \
    row_pair_v(w[0][0], w[0][1], s[3]), row_pair_v(w[0][0], w[0][1], s[2]), row_pair_v(w[1][0], w[1][1], s[1]), row_pair_v(w[2][0], w[2][1], s[0]), \
    row_pair_v(w[3][0], w[3][1], s[0]), row_pair_v(w[4][0], w[4][1], s[0]), row_pair_v(w[5][0], w[5][1], s[1]), row_pair_v(w[6][0], w[6][1], s[2]) }
#define FC_RING(Q) { \
    row_pair<3 + Q + 0>(w[6][0], w[6][1], w[6][2]), row_pair<3 + Q + 1>(w[6][0], w[6][1], w[6][2]), row_pair<3 + Q + 2>(w[5][0], w[5][1], w[5][2]), row_pair<3 + Q + 3>(w[4][0], w[4][1], w[4][2]), \
    row_pair<3 + Q + 3>(w[3][0], w[3][1], w[3][2]), row_pair<3 + Q + 3>(w[2][0], w[2][1], w[2][2]), row_pair<3 + Q + 2>(w[1][0], w[1][1], w[1][2]), row_pair<3 + Q + 1>(w[0][0], w[0][1], w[0][2]), \
    row_pair<3 + Q + 0>(w[0][0], w[0][1], w[0][2]), row_pair<3 + Q - 1>(w[0][0], w[0][1], w[0][2]), row_pair<3 + Q - 2>(w[1][0], w[1][1], w[1][2]), row_pair<3 + Q - 3>(w[2][0], w[2][1], w[2][2]), \
    row_pair<3 + Q - 3>(w[3][0], w[3][1], w[3][2]), row_pair<3 + Q - 3>(w[4][0], w[4][1], w[4][2]), row_pair<3 + Q - 2>(w[5][0], w[5][1], w[5][2]), row_pair<3 + Q - 1>(w[6][0], w[6][1], w[6][2]) }

// Necessary condition for "corner at threshold t" of a pixel pair: a 9-arc contains one pixel of every opposite pair (k, k+8),
// so each max(r_k, r_k+8) exceeds C + t (bright arc) or each min(r_k, r_k+8) is below C - t (dark arc).  Four of the eight
// pairs are tested (compass + diagonals: 14 % of the 4-pixel groups of a textured frame survive at t = 20, against 11 % with
// all eight and 23 % with the compass pairs alone), which needs 5 of the 7 window rows, 9 byte-pair extractions and 19
// packed min/max per pixel pair instead of 17 and 118 for the exact score.  Returns non-zero where either pixel of the pair passes:
// the two comparisons are saturating subtractions (v_pk_sub_u16 clamp), so no compare / select chain is needed.
// (Round 2 measured the alternative the instruction-rate table suggests — the same four pairs for four pixels at a time in byte-SWAR
//  arithmetic built only from full-rate add / sub / and / or / lshr on 7-bit halved values: 65 full-rate + 29 half-rate operations per
//  group instead of 16 + 74, but the halving weakens the test by up to two grey levels, 11.1 % instead of 9.0 % of the pixel pairs reach
//  the exact score (1.37 instead of 1.21 passes of that loop per cell) and the kernel ran 0.624 ms against 0.630 ms: kept as it was.
//  With byte-offset LDS reads instead of v_alignbyte it ran 0.846 ms: a DS read off its natural alignment is replayed.
//  profiles/r02_exp_fast_swar_*.json)
__device__ __forceinline__ pku16 psubsat(pku16 a, pku16 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ unsigned fast_pretest_pair(const pku16 r[16], pku16 c, pku16 tt)
{
    pku16 mn = pmax(r[0], r[8]), mx = pmin(r[0], r[8]);
#pragma unroll
    for (int k = 2; k < 8; k += 2) { mn = pmin(mn, pmax(r[k], r[k + 8])); mx = pmax(mx, pmin(r[k], r[k + 8])); }
    // bright: weakest tested pair's maximum > C + t;  dark: weakest tested pair's minimum < C - t (never true when C - t clamps to 0)
    return (unsigned)(psubsat(mn, c + tt) | psubsat(psubsat(c, tt), mx));
}

// number of set bits of m below this lane, plus base (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int fc_rank(unsigned long long m, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

__device__ __forceinline__ void fc_load_window(const uint8_t* patch, int PS, int y, int grp, unsigned (&w)[7][3])
{
    const unsigned* prow = reinterpret_cast<const unsigned*>(patch + y * PS) + grp;
#pragma unroll
    for (int r = 0; r < 7; r++) { w[r][0] = prow[r * (PS >> 2)]; w[r][1] = prow[r * (PS >> 2) + 1]; w[r][2] = prow[r * (PS >> 2) + 2]; }
}

// CPS / CSS: the patch / score row strides as compile-time constants (0 = read them from the parameters): with the strides of the usual
// 30-px grid (48 and 40 bytes) every window row is an immediate offset of one LDS address instead of an address addition per row.
// (FAST and the blur as ONE launch with workgroups of both kinds alternating on every CU - k_fast_blur, round 3 - was bit-exact and slower: 1.41-1.46 ms
//  against 0.92 + 0.42 standalone; what one kind leaves idle the other cannot use.  profiles/r03_exp_fast_blur_one_launch.jsonl)
template <int CPS, int CSS>
__global__ __launch_bounds__(256, 8) void k_fast_cells(ExtractParams P)       // 8 waves per SIMD = what the LDS block allows at the metric's geometry (20 448 bytes: 8 workgroups per CU): at most 64 VGPRs
{
    HIP_DYNAMIC_SHARED(unsigned, fc_lds)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int tile, frame;
    if (!xcd_frame_map((P.fc_ncells + FC_WAVES - 1) / FC_WAVES, P.nframes, tile, frame)) return;      // this launch: cells [fc_cell0, fc_cell0 + fc_ncells); the whole workgroup leaves
    frame += P.frame0;
    const int cell_last = P.fc_cell0 + P.fc_ncells - 1;
    const bool have = P.fc_cell0 + tile * FC_WAVES + wave <= cell_last;           // a wave without a cell of its own still serves the workgroup's shared stage
    const int cell_id = min(P.fc_cell0 + tile * FC_WAVES + wave, cell_last);
    const CellDesc cd = P.cells[cell_id];
    const int pw = cd.maxX - cd.iniX, ph = cd.maxY - cd.iniY;       // sub-image
    const int cw = pw - 6, ch = ph - 6;                              // examined interior (rows/cols 3 .. n-4)
    const bool work = have && !cd.skipped && cw > 0 && ch > 0;
    const int PS = CPS ? CPS : P.fc_pstride, SS = CSS ? CSS : P.fc_sstride;      // byte strides, multiples of 4
    const int WB = fc_wave_bytes(P.fc_pbytes, SS, P.fc_srows, P.fc_listcap);      // per-wave LDS region: patch | score map | pair list
    const int score_off = (P.fc_pbytes + 15) & ~15, list_off = score_off + ((SS * P.fc_srows + 4 + 15) & ~15);
    uint8_t* const lds0 = reinterpret_cast<uint8_t*>(fc_lds);
    uint8_t* wbase = lds0 + wave * WB;
    uint8_t* patch = wbase;                                          // patch column 0 is 4-byte aligned; row stride PS
    uint8_t* score = wbase + score_off;                              // interior (x, y) at score[(y+1)*SS + 4 + x]; zero ring around it: a row's four bytes in front are the
                                                                     // row above's pad behind (SS = 4 + the interior width rounded up to 4), the last row's pad behind is the spare dword
    unsigned short* list = reinterpret_cast<unsigned short*>(wbase + list_off);
    int* const sh = reinterpret_cast<int*>(lds0 + FC_WAVES * WB);    // workgroup header: [0..3] listed pairs of each wave, [4..7] interior width of each wave's cell
    const int ng = (cw + 3) >> 2;                                    // 4-pixel groups per interior row (<= 16)
    if (lane == 0) sh[4 + wave] = cw;
    if (work) {
        // The cell's sub-image goes to LDS by LDS-DMA at the patch's own byte alignment, so patch column 0 lands on an LDS dword
        // boundary without any shifting: pass k fills LDS dwords [64k, 64k+64) of the patch, lane l the dword (row, col) =
        // divmod(64k + l, PS/4), both from a host table that only depends on PS.  Positions outside the patch (rows >= ph, the
        // row tail) read the patch's last dword instead: any in-bounds address does, nothing reads those LDS bytes.  The last
        // dword of a row may read up to 3 bytes past the patch: still inside the image row (cells end >= 13 px before the border).
        int spitch; const uint8_t* src = level_src(P, frame, cd.level, spitch);
        src += (long long)cd.iniY * spitch + cd.iniX;
        const ORBHIP_GLOBAL uint8_t* gp = uniform_ptr(src);
        const unsigned maxoff = (unsigned)((ph - 1) * spitch + 4 * (((pw + 3) >> 2) - 1));
        const int np = P.fc_np;
        for (int k0 = 0; k0 < np; k0 += 8) {
            const int4* tab = P.fc_dma + (k0 >> 1) * 64 + lane;     // per 4 passes: int4 rows, int4 4*dword columns, [..][64 lanes]
            int4 rr[2], dd[2];
#pragma unroll
            for (int j = 0; j < 2; j++) if (k0 + 4 * j < np) { rr[j] = tab[(2 * j) * 64]; dd[j] = tab[(2 * j + 1) * 64]; }
#pragma unroll
            for (int j = 0; j < 2; j++) if (k0 + 4 * j < np) {
                const int r4[4] = {rr[j].x, rr[j].y, rr[j].z, rr[j].w}, d4[4] = {dd[j].x, dd[j].y, dd[j].z, dd[j].w};
#pragma unroll
                for (int q = 0; q < 4; q++) if (k0 + 4 * j + q < np)
                    lds_dma_dword(gp + min(__umul24((unsigned)r4[q], (unsigned)spitch) + (unsigned)d4[q], maxoff), patch + (k0 + 4 * j + q) * 256);
            }
        }
        // the whole score map starts at zero: the passes below only fill the pixel pairs they examine (16 bytes per lane and store: the map's
        // base and its reserved size are multiples of 16, so rounding the count up stays inside it)
        for (int i = lane; i < ((SS >> 2) * (ch + 2) + 1 + 3) >> 2; i += 64) reinterpret_cast<uint4*>(score)[i] = uint4{0u, 0u, 0u, 0u};
        lds_dma_wait();
    }
    __builtin_amdgcn_wave_barrier();
    // interior rows handled per wave iteration and this lane's (row, group): divisions by the wave-uniform ng through the cell's
    // reciprocal ceil(2^16 / ng) from the host table (exact for lane < 64, ng <= 16)
    const int rp = (64 * cd.inv_ng) >> 16;
    const int sr = (lane * cd.inv_ng) >> 16, grp = lane - sr * ng;
    const bool lane_ok = sr < rp;
    const unsigned long long lok_mask = __builtin_amdgcn_ballot_w64(lane_ok);
    unsigned* out = P.cell_cand + (long long)frame * P.cand_slots_per_frame + cd.cand_idx;

    // The two cv::FAST calls of the reference (ORBextractor.cc:809-816): iniThFAST, and minThFAST only when the first returned
    // nothing.  Only pixels with score >= th matter for a call, and its NMS sees every weaker pixel as 0, so each call is
    //      a. cheap necessary test at th on every pixel pair (a lane tests the two pairs of its 4-pixel group) -> ordered
    //         list of the pairs that may hold such a pixel                                                   [the cell's own wave]
    //      b. exact scores of those pairs only, a lane per listed pair: THE WORKGROUP'S FOUR LISTS AS ONE - a cell lists 40 pairs on
    //         average, so its own wave would run this stage (the most expensive instructions of the kernel) with 40 of 64 lanes busy in
    //         one pass and a handful in a second one (1.21 passes per cell); four cells' pairs fill 3 wave passes instead of 4.8
    //      c. NMS + row-major emission over the listed pairs                                                 [the cell's own wave]
    // Both calls run for the whole workgroup (two barriers each); a wave whose cell is finished lists nothing and only serves stage b.
    int count = 0;
    bool busy = work;
    for (int phase = 0; phase < 2; phase++) {
        const int th = phase ? P.minTh : P.iniTh;
        const pku16 tt = {(unsigned short)th, (unsigned short)th};
        int nq = 0;
        if (busy)
        for (int yb = 0; yb < ch; yb += rp) {
            // branch-free: idle lanes (beyond the last row / the last whole row group) test a clamped row and are masked afterwards
            const int yy = yb + sr, y = min(yy, ch - 1);
            const bool act = lane_ok && yy < ch;
            unsigned pa, pb;
            {
                unsigned w[7][3];
                fc_load_window(patch, PS, y, grp, w);
                const pku16 ra[16] = FC_RING(0);
                const pku16 rb[16] = FC_RING(2);
                pa = fast_pretest_pair(ra, row_pair<3>(w[3][0], w[3][1], w[3][2]), tt);         // score >= t  <=>  best arc margin > t  =>  every opposite pair's margin > t
                pb = fast_pretest_pair(rb, row_pair<5>(w[3][0], w[3][1], w[3][2]), tt);
            }
            // ballots of plain compares are the compare's own lane mask; their conjunctions stay on the scalar unit (a ballot of `a && b` is
            // materialised as select + re-compare, 2 VALU each)
            const unsigned long long am = __builtin_amdgcn_ballot_w64(yy < ch) & lok_mask;
            const unsigned long long ma = __builtin_amdgcn_ballot_w64(pa != 0) & am, mb = __builtin_amdgcn_ballot_w64(pb != 0) & am;
            const bool la = act && pa != 0, lb = act && pb != 0;
            int pos = fc_rank(mb, fc_rank(ma, nq));
            if (la) list[pos++] = (unsigned short)((y << 8) | (2 * grp));               // id = row << 8 | pair index in the row
            if (lb) list[pos] = (unsigned short)((y << 8) | (2 * grp + 1));
            nq += __popcll(ma) + __popcll(mb);
        }
        if (lane == 0) sh[wave] = nq;
        __syncthreads();
        {
            // stage b over the concatenation of the four lists: entry e belongs to wave j's cell (patch, score map, list, width of ITS region)
            const int o1 = __builtin_amdgcn_readfirstlane(sh[0]), o2 = o1 + __builtin_amdgcn_readfirstlane(sh[1]), o3 = o2 + __builtin_amdgcn_readfirstlane(sh[2]);
            const int T = o3 + __builtin_amdgcn_readfirstlane(sh[3]);
            // Chunks of 64 entries go to the waves round-robin from a starting wave that differs between the workgroups resident on a CU
            // (a wave's index fixes its SIMD: with chunk c always on wave c, SIMD 0 would run this stage for every workgroup and SIMD 3 for
            // almost none - measured as 0.899 ms against 0.922 ms without the shared stage, where the instruction count promised 0.84)
            const int rot = (tile ^ (tile >> 2) ^ (tile >> 5) ^ (tile >> 7) ^ frame) & (FC_WAVES - 1);
            for (int e0 = 64 * ((wave - rot) & (FC_WAVES - 1)); e0 < T; e0 += 64 * FC_WAVES) {
                const int e = e0 + lane;
                if (e < T) {
                    // owner j of entry e and its index le in the owner's list: with o1 <= o2 <= o3 the differences e - o_k fall with k, the
                    // negative ones are huge as unsigned numbers, so le is their unsigned minimum and j the number of non-negative ones
                    const int d1 = e - o1, d2 = e - o2, d3 = e - o3;
                    const int le = (int)min(min((unsigned)e, (unsigned)d1), min((unsigned)d2, (unsigned)d3));
                    const int j = 3 + (d1 >> 31) + (d2 >> 31) + (d3 >> 31);
                    const int jo = (int)__umul24((unsigned)j, (unsigned)WB);
                    const int cwj = sh[4 + j];
                    const uint8_t* jb = lds0 + jo;
                    const int id = *reinterpret_cast<const unsigned short*>(lds0 + jo + list_off + 2 * le), y = id >> 8, pr = id & 0xff, x0 = 2 * pr;
                    // the pair's 7 x 8-byte window, shifted so that it starts at byte 0 of the first dword whichever half of the group it is
                    const unsigned* prow = reinterpret_cast<const unsigned*>(jb + y * PS) + (pr >> 1);
                    const unsigned shsel = (pr & 1) ? 0x00020002u : 0u;                      // the pair is the second one of its 4-pixel group: two bytes on
                    const unsigned sel[5] = {0x0c010c00u + shsel, 0x0c020c01u + shsel, 0x0c030c02u + shsel, 0x0c040c03u + shsel, 0x0c050c04u + shsel};
                    unsigned w[7][3];
#pragma unroll
                    for (int r = 0; r < 7; r++) { w[r][0] = prow[r * (PS >> 2)]; w[r][1] = prow[r * (PS >> 2) + 1]; w[r][2] = prow[r * (PS >> 2) + 2]; }
                    const pku16 ra[16] = FC_RING_V(sel);
                    const pki16 sa = fast_score_pair(ra, row_pair_v(w[3][0], w[3][1], sel[3]));
                    int s0 = sa[0], s1 = sa[1];
                    if (s0 < th || x0 >= cwj) s0 = 0;                                        // corner at th  <=>  score >= th
                    if (s1 < th || x0 + 1 >= cwj) s1 = 0;
                    *reinterpret_cast<unsigned short*>(lds0 + jo + score_off + (y + 1) * SS + 4 + x0) = (unsigned short)(s0 | (s1 << 8));
                }
            }
        }
        __syncthreads();
        if (busy)
        for (int qb = 0; qb < nq; qb += 64) {
            unsigned gtm = 0, rbm = 0; int y = 0, x0 = 0;
            if (qb + lane < nq) {
                int li = qb + lane;
                asm volatile("" : "+v"(li));            // the list address is rebuilt here: kept live across stage b it was spilled to scratch (72-VGPR cap)
                const int id = list[li];
                y = id >> 8; x0 = 2 * (id & 0xff);
                // rows above / at / below, bytes x0-1 .. x0+2 (the pad dwords left and right of a row are zero)
                const int qo = (y + 1) * SS + 4 + x0 - 1;                                // byte offset in the score map (its base is 16-byte aligned)
                const unsigned* qa = reinterpret_cast<const unsigned*>(score) + (qo >> 2);      // index arithmetic, not pointer bits: the reads stay ds_read
                const unsigned shb = (unsigned)qo & 3u;
                const int sdw = SS >> 2;
                const unsigned ra4 = __builtin_amdgcn_alignbyte(qa[1 - sdw], qa[-sdw], shb);
                const unsigned rb4 = __builtin_amdgcn_alignbyte(qa[1], qa[0], shb);
                const unsigned rc4 = __builtin_amdgcn_alignbyte(qa[1 + sdw], qa[sdw], shb);
                // both pixels at once as u16 lanes: centre pair, its eight neighbour pairs, one packed maximum, strict comparison by saturating
                // subtraction (a kept pixel beats a neighbour that is >= 0, so it is a corner: no separate score > 0 test)
                auto bp = [](unsigned r, unsigned sel) { return (pku16)__builtin_amdgcn_perm(0u, r, sel); };
                const pku16 cpair = bp(rb4, 0x0c020c01u);
                pku16 nb = pmax(bp(rb4, 0x0c010c00u), bp(rb4, 0x0c030c02u));
                nb = pmax(nb, pmax(bp(ra4, 0x0c010c00u), pmax(bp(ra4, 0x0c020c01u), bp(ra4, 0x0c030c02u))));
                nb = pmax(nb, pmax(bp(rc4, 0x0c010c00u), pmax(bp(rc4, 0x0c020c01u), bp(rc4, 0x0c030c02u))));
                gtm = (unsigned)psubsat(cpair, nb); rbm = rb4;
            }
            const bool keep[2] = {(gtm & 0xffffu) != 0, (gtm >> 16) != 0};                 // idle lanes: gtm = 0
            const int scv[2] = {(int)(rbm >> 8) & 0xff, (int)(rbm >> 16) & 0xff};
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(keep[0]), m1 = __builtin_amdgcn_ballot_w64(keep[1]);
            int rank = fc_rank(m1, fc_rank(m0, count));
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (keep[q]) {
                    // FAST reports cell-local (x, y); the reference adds (j*wCell, i*hCell)  (ORBextractor.cc:822-823)
                    const unsigned px = (unsigned)(x0 + q + 3 + cd.shiftX), py = (unsigned)(y + 3 + cd.shiftY);
                    if (rank < cd.cand_cap) out[rank] = px | (py << 12) | ((unsigned)scv[q] << 24);
                    rank++;
                }
            count += __popcll(m0) + __popcll(m1);
        }
        if (count > 0) busy = false;
        // vKeysCell.empty() -> cv::FAST again with minThFAST (ORBextractor.cc:812-816).  The score map restarts from zero: scores the
        // first call left behind (maxima that suppressed each other) are not this call's.  (Every other wave's stage-b stores into this map
        // are behind the barrier above; the next ones come after the next barrier.)
        if (phase == 0 && busy) for (int i = lane; i < ((SS >> 2) * (ch + 2) + 1 + 3) >> 2; i += 64) reinterpret_cast<uint4*>(score)[i] = uint4{0u, 0u, 0u, 0u};
    }
    if (have && lane == 0) P.cell_count[(long long)frame * P.ncells_total + cell_id] = work ? min(count, cd.cand_cap) : 0;
}
#undef FC_RING
#undef FC_RING_V

void orbhip_launch_fast_cells(const ExtractParams& P, int nframes, hipStream_t s, int cell0, int ncells)
{   // cells [cell0, cell0 + ncells) of the level-major cell table (ncells < 0: all)
    const size_t lds = (size_t)FC_WAVES * fc_wave_bytes(P.fc_pbytes, P.fc_sstride, P.fc_srows, P.fc_listcap) + 32;      // + the workgroup header
    ExtractParams Q = P; Q.nframes = nframes;
    Q.fc_cell0 = ncells < 0 ? 0 : cell0; Q.fc_ncells = ncells < 0 ? P.ncells_total : ncells;
    if (Q.fc_ncells <= 0) return;
    const dim3 grid(xcd_grid((Q.fc_ncells + FC_WAVES - 1) / FC_WAVES, nframes), 1, 1);
    if (Q.fc_pstride == 48 && Q.fc_sstride == 36) hipLaunchKernelGGL((k_fast_cells<48, 36>), grid, dim3(256, 1, 1), lds, s, Q);
    else hipLaunchKernelGGL((k_fast_cells<0, 0>), grid, dim3(256, 1, 1), lds, s, Q);
}

// ------------------------------------------------------------------------------------------------ quadtree
// DistributeOctTree (ORBextractor.cc:539-763) without pointers.  Observations that make it data-parallel:
//  * a key's path through the quadtree depends only on its own coordinates (children halve the parent rectangle
//    with ceil, DivideNode :483-490), so every candidate gets a 2-bit-per-depth path code once;
//  * children are always push_front'ed and parents erased, so the std::list is at any time sorted by DESCENDING
//    creation order -> a node is identified by its position in that order; one pass = one compaction;
//  * the final phase (:673-738) splits nodes largest-first until the list reaches N: sort keys, a prefix sum of
//    (children-1) over that order and the first position where the running size reaches N reproduce the `break`.
// Canonical tie-break H1: equal sizes -> later-created node first == smaller list position first.
// One workgroup per (frame, level); node tables in LDS, per-candidate arrays in an L2-resident HBM workspace.
#define QT_T 256

__device__ __forceinline__ int qt_digit(unsigned code, int depth)
{   // child index of a key at a node of the given depth; beyond the stored depth every key goes to n1 (unsupported sizes only)
    return depth < ORBHIP_QT_DEPTH ? (int)((code >> (2 * (ORBHIP_QT_DEPTH - 1 - depth))) & 3u) : 0;
}

// scan scratch: one int per 64-element chunk of the longest scanned array + the total (sized on the host, see orbhip_quadtree_scr)
__device__ __forceinline__ int qt_wave_incl_scan(int v, int lane)
{   // inclusive wave64 prefix sum from DPP row shifts / broadcasts: seven 4-cycle additions (six ds_bpermute shuffles + selects took 24 cycles apiece)
    (void)lane;
    int t = v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);       // row_shr:1 of the input
    t += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);              // row_shr:2 of the input
    t += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);              // row_shr:3 of the input -> sums of (up to) four
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xe, true);              // row_shr:4, banks 1-3
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xc, true);              // row_shr:8, banks 2-3 -> inclusive sums inside each row of 16
    t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, true);              // row_bcast:15 into rows 1 and 3
    t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, true);              // row_bcast:31 into rows 2 and 3
    return t;
}
template <bool TRAILING_BARRIER = true>
__device__ __forceinline__ int qt_block_exscan(int* a, int n, int* scratch, int tid)
{   // in-place exclusive scan of a[0..n), n <= 64*(scr-1); the total lands in scratch[-1] (the slot in front); all QT_T threads must call.
    // 64-element chunks are scanned inside the waves, chunk totals by wave 0: three barriers in all - two when the caller guarantees a barrier
    // of its own before anything writes `scratch` again.  Thread t reads and writes only elements t, t + QT_T, ...: a caller whose thread t
    // produced exactly those needs no barrier in front of the call, nor behind it before thread t reads them back.
    const int lane = tid & 63, wave = tid >> 6, nch = (n + 63) >> 6;
    for (int c = wave; c < nch; c += QT_T / 64) {
        const int i = c * 64 + lane, v = i < n ? a[i] : 0;
        const int incl = qt_wave_incl_scan(v, lane);
        if (i < n) a[i] = incl - v;
        if (lane == 63) scratch[c] = incl;
    }
    __syncthreads();
    if (wave == 0) {
        int run = 0;
        for (int cb = 0; cb < nch; cb += 64) {
            const int v = (cb + lane < nch) ? scratch[cb + lane] : 0;
            const int incl = qt_wave_incl_scan(v, lane);
            if (cb + lane < nch) scratch[cb + lane] = run + incl - v;
            run += __shfl(incl, 63);
        }
        if (lane == 0) scratch[-1] = run;
    }
    __syncthreads();
    for (int i = tid; i < n; i += QT_T) a[i] += scratch[i >> 6];
    const int total = scratch[-1];
    if (TRAILING_BARRIER) __syncthreads();
    return total;
}

#define QT_KPT 16                       // candidates per thread whose keys stay in registers
#define QT_REGKEYS (QT_KPT * QT_T)      // = 4096 candidates per (frame, level); more -> HBM workspace

int orbhip_quadtree_scr(int maxn, int maxcells) { return (std::max(maxn, maxcells) + 63) / 64 + 2; }
size_t orbhip_quadtree_lds_bytes(int maxn, int maxcells)
{
    // ints: cnt[2][maxn] | cc[4*maxn] | { a | b | sidx | split | best [maxn each] } = { s_pref[maxcells+1] | s_slot[maxcells] } | scratch[scr] | misc[16]
    // then map u16 [4*maxn] | depth u8 [2][maxn]
    // The per-cell prefix sums and slots are only read while the candidates are copied into their dense order (phase A), the five per-node
    // arrays only written from the first dividing pass on (three barriers later): they share one region.  KITTI shape: 23.7 KB (27.2 KB with
    // both) -> six workgroups per CU instead of five.  Round 1 kept the per-candidate keys in LDS too (6 B x 4096 = 24 KB, 51.9 KB in all,
    // three per CU): beside the blur or FAST its three workgroups took 156 of a CU's 160 KB and starved the throughput kernel it was
    // supposed to run under; candidate k is only ever touched by thread k mod 256, so its path code and node now live in that thread's VGPRs.
    const size_t ints = (size_t)maxn * (2 + 4) + (size_t)std::max(2 * maxcells + 1, 5 * maxn) + orbhip_quadtree_scr(maxn, maxcells) + 16;
    return sizeof(int) * ints + (size_t)maxn * 4 * 2 + (((size_t)maxn * 2 + 3) & ~(size_t)3);
}

// per-candidate state of the quadtree replay: path code and current node (list position).  Candidate k belongs to thread k mod QT_T in
// every loop, so up to QT_REGKEYS candidates keep both in registers (the loops are fully unrolled); beyond that they live in HBM.
struct QtKeysReg {
    unsigned code[QT_KPT]; int node[QT_KPT];
    template <class F> __device__ __forceinline__ void each(int n, int tid, F f)
    {
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) { const int k = tid + j * QT_T; if (k < n) f(k, code[j], node[j]); }
    }
    // f(k, src[k], code, node) with every src[k] of the thread requested before the first one is used (one memory latency, not one per key)
    template <class F> __device__ __forceinline__ void each_loaded(int n, int tid, const unsigned* src, F f)
    {
        unsigned v[QT_KPT];
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) { const int k = tid + j * QT_T; v[j] = k < n ? src[k] : 0u; }
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) { const int k = tid + j * QT_T; if (k < n) f(k, v[j], code[j], node[j]); }
    }
};
struct QtKeysHbm {
    unsigned* code; int* node;
    template <class F> __device__ __forceinline__ void each(int n, int tid, F f)
    {
        for (int k = tid; k < n; k += QT_T) f(k, code[k], node[k]);
    }
    template <class F> __device__ __forceinline__ void each_loaded(int n, int tid, const unsigned* src, F f)
    {
        for (int k = tid; k < n; k += QT_T) f(k, src[k], code[k], node[k]);
    }
};

struct QtLds {
    int *pref, *slot, *cntA, *cntB, *cc, *a, *b, *sidx, *split, *best, *scratch, *misc;
    unsigned short* map; unsigned char *depA, *depB;
};

// cell id <-> its position in the list order after K regular passes (see "regular passes" below): an involution
__device__ __forceinline__ int qt_jump_xform(int t, int K, int nIni)
{
    const int bits = 2 * K, low = (t & ((1 << bits) - 1)) ^ (0x33333333 & ((1 << bits) - 1)), root = t >> bits;
    return (((K & 1) ? nIni - 1 - root : root) << bits) | low;
}
template <class Keys>
__device__ __forceinline__ void qt_replay(const ExtractParams& P, const LevelGeom& g, const QtLds& L, Keys& keys, unsigned* qval, int n,
                                          int frame, int level, int tid, int D)
{
    const int N = g.nfeat;
    const int wave = tid >> 6, lane = tid & 63;
    // ---- A. dense canonical candidate order: cells row-major, row-major inside a cell (ORBextractor.cc:789-829).
    //      One thread per CELL copies the cell's records to their dense positions (eight loads in flight at a time), then one thread per
    //      candidate reads its record back (all of a thread's records requested at once).  The first form - a thread per candidate that
    //      bisected the per-cell prefix sums for its cell and then fetched its record - paid nine dependent LDS reads and a memory latency
    //      per key, sixteen keys one after the other: a third of a level-0 workgroup's 67 us in a single-frame call.
    const unsigned* cand = P.cell_cand + (long long)frame * P.cand_slots_per_frame;
    {
        const int* ccount = P.cell_count + (long long)frame * P.ncells_total + g.cell_first;
        for (int c = tid; c < g.ncells; c += QT_T) {
            const int base = L.pref[c], cntc = min(ccount[c], n - base), so = L.slot[c];      // (n caps the total: never more than the level's slots)
            for (int i0 = 0; i0 < cntc; i0 += 8) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = i0 + u < cntc ? cand[so + i0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) if (i0 + u < cntc) qval[base + i0 + u] = v[u];
            }
        }
    }
    __syncthreads();
    keys.each_loaded(n, tid, qval, [&](int idx, unsigned v, unsigned& kcode, int& knode) {
        (void)idx;
        const int x = v & 0xfff, y = (v >> 12) & 0xfff;
        int root = __float2int_rz(__fdiv_rn((float)x, g.hX));                    // vpIniNodes[kp.pt.x/hX]  (:569)
        root = min(max(root, 0), g.nIni - 1);
        int ULx = __float2int_rz(__fmul_rn(g.hX, (float)root)), URx = __float2int_rz(__fmul_rn(g.hX, (float)(root + 1)));   // :555-556
        int ULy = 0, BRy = g.maxBorderY - ORBHIP_EDGE;
        unsigned code = 0;
#pragma unroll
        for (int d = 0; d < ORBHIP_QT_DEPTH; d++) {                               // DivideNode :483-526
            const int mx = ULx + ((URx - ULx + 1) >> 1), my = ULy + ((BRy - ULy + 1) >> 1);
            const int right = !(x < mx), bottom = !(y < my);
            code = (code << 2) | (unsigned)(right + 2 * bottom);                  // n1=0 n2=1 n3=2 n4=3
            if (right) ULx = mx; else URx = mx;
            if (bottom) ULy = my; else BRy = my;
        }
        kcode = code; knode = root; atomicAdd(&L.cntA[root], 1);
        if (D > 0) atomicAdd(&L.cc[(root << (2 * D)) + (int)(code >> (2 * (ORBHIP_QT_DEPTH - D)))], 1);      // keys per depth-D cell (regular passes, below)
    });
    __syncthreads();
    // ---- B'. the regular passes in one step.
    // While every node of the list holds more than one key, a pass divides EVERY node (:606-665): the list after K such passes is exactly the
    // non-empty depth-K cells, and since children are pushed to the front in n1..n4 order while the parents are walked front to back,
    //     list after pass k+1  =  for p in REVERSED(list after pass k): children of p in the order n4, n3, n2, n1,
    // i.e. the lexicographic order of (root, d1 .. dK) with digit j descending when K - j is even and ascending when it is odd (the root like
    // d1).  A node's position is therefore the number of non-empty cells before it in that order - one prefix sum over the per-cell key counts,
    // which the candidates' path codes give in one histogram (one LDS atomic per key onto nIni * 4^D counters, taken above) instead of one
    // contended histogram + five barriers per pass.  K = the number of passes the reference runs in that regime: pass 1 if no root holds a single
    // key; pass k+1 if after pass k the loop goes on (m_k < N, m_k != m_(k-1)), stays in its first phase (m_k + 3 * nToExpand <= N, :673) and
    // again every node holds more than one key.  Measured: 3 of 4 passes of a 1241 x 376 frame's level 0, 20 of its 57 us
    // (profiles/r04_exp_quadtree_phase_times.txt).  Whatever does not fit (K = 0: a root with one key; deeper passes) is replayed as before.
    int* cnt = L.cntB; unsigned char* dep = L.depB; int* cnt2 = L.cntA; unsigned char* dep2 = L.depA;
    unsigned short* s_map = L.map;
    int *s_cc = L.cc, *s_a = L.a, *s_b = L.b, *s_sidx = L.sidx, *s_split = L.split, *s_best = L.best, *s_scratch = L.scratch, *s_misc = L.misc;
    int m = 0, K = 0, jumpPrev = 0, jumpExp = 0;
    if (D > 0) {
        // key counts per depth-d cell at L.cntA + hoff[d] (offsets, not an array of pointers: a pointer picked from an array by a run-time index is a
        // generic one, and reads through it were flat_loads that wait for every outstanding global access as well)
        int hoff[5] = {0, 0, 0, 0, 0};
        hoff[D] = (int)(s_cc - L.cntA);
        if (D >= 1) hoff[D - 1] = (int)(L.cntB - L.cntA);
        if (D >= 2) hoff[D - 2] = (int)(s_best - L.cntA);
        if (D >= 3) hoff[D - 3] = (int)(s_split - L.cntA);
        if (D >= 4) hoff[D - 4] = (int)(s_sidx - L.cntA);
        // [5 + 2d] = non-empty depth-d cells, [6 + 2d] = those with more than one key (zeroed with the counters at kernel start).  Every shallower
        // histogram is summed straight from H_D and counted in the same step, each count reduced inside the wave first: one barrier for all depths
        // (a barrier and 256 same-address LDS atomics per depth cost more than the three passes' worth of work this replaces saves).
        {
            int ne[5] = {0, 0, 0, 0, 0}, ex[5] = {0, 0, 0, 0, 0};
            const int ncD = g.nIni << (2 * D);
            for (int c = tid; c < ncD; c += QT_T) { const int h = s_cc[c]; ne[D] += h > 0; ex[D] += h > 1; }
            int base = 0;
            for (int d = D - 1; d >= 0; d--) {
                const int nc = g.nIni << (2 * d), span = 1 << (2 * (D - d));
                // cells of depth d are handed out behind those of the deeper levels, so that the few long sums of the shallow levels spread over threads
                for (int c = tid - base; c < nc; c += QT_T) {
                    if (c < 0) continue;
                    int h = 0;
                    for (int k = 0; k < span; k += 4) h += (s_cc[c * span + k] + s_cc[c * span + k + 1]) + (s_cc[c * span + k + 2] + s_cc[c * span + k + 3]);      // (span = 4, 16 or 64: four reads in flight)
                    (L.cntA + hoff[d])[c] = h; ne[d] += h > 0; ex[d] += h > 1;
                }
                base = (base + nc) % QT_T;
            }
#pragma unroll
            for (int d = 0; d < 5; d++) {
                if (d > D) break;
                const int a1 = __builtin_amdgcn_readlane(qt_wave_incl_scan(ne[d], lane), 63), a2 = __builtin_amdgcn_readlane(qt_wave_incl_scan(ex[d], lane), 63);
                if (lane == 0) { if (a1) atomicAdd(&s_misc[5 + 2 * d], a1); if (a2) atomicAdd(&s_misc[6 + 2 * d], a2); }
            }
        }
        __syncthreads();
        if (s_misc[5] > 0 && s_misc[6] == s_misc[5]) {
            K = 1;
            while (K < D) {
                const int mk = s_misc[5 + 2 * K], ek = s_misc[6 + 2 * K], mp = s_misc[5 + 2 * (K - 1)];
                if (!(mk < N && mk != mp && ek == mk && 4 * mk <= N)) break;
                K++;
            }
        }
        if (K > 0) {
            const int nc = g.nIni << (2 * K);
            const int* H = L.cntA + hoff[K];
            int* F = s_a;                                                    // [nc <= 4 maxn]: a | b | sidx | split (the shallower histograms there are dead, H_K never lies under F's nc entries)
            if (nc <= QT_T) {                                                // one cell per thread: a wave scan and four partial sums, one barrier
                const int f = tid < nc ? (H[qt_jump_xform(tid, K, g.nIni)] > 0 ? 1 : 0) : 0;
                const int incl = qt_wave_incl_scan(f, lane);
                if (lane == 63) s_scratch[wave] = incl;
                __syncthreads();
                int off = 0;
                for (int w2 = 0; w2 < wave; w2++) off += s_scratch[w2];
                if (tid < nc) F[tid] = off + incl - f;
                m = s_scratch[0] + s_scratch[1] + s_scratch[2] + s_scratch[3];
            } else {
                for (int t = tid; t < nc; t += QT_T) F[t] = H[qt_jump_xform(t, K, g.nIni)] > 0 ? 1 : 0;
                m = qt_block_exscan(F, nc, s_scratch, tid);                  // thread t scans the flags thread t wrote, and reads them back below
            }
            cnt = L.cntA; dep = L.depA; cnt2 = L.cntB; dep2 = L.depB;        // (H_(D-1) lives in cntB)
            for (int t = tid; t < nc; t += QT_T) { const int h = H[qt_jump_xform(t, K, g.nIni)]; if (h > 0) { cnt[F[t]] = h; dep[F[t]] = (unsigned char)K; } }
            __syncthreads();
            keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {
                knode = F[qt_jump_xform((knode << (2 * K)) + (int)(kcode >> (2 * (ORBHIP_QT_DEPTH - K))), K, g.nIni)];
            });
            jumpPrev = s_misc[5 + 2 * (K - 1)]; jumpExp = s_misc[6 + 2 * K];
            __syncthreads();                                                 // F and the histograms are read: their regions become the passes' scratch again
        }
    }
    if (K == 0) {
        // ---- B. initial list: non-empty roots in order (:552-585)
        if (tid == 0) {
            int m0 = 0;
            for (int r = 0; r < g.nIni; r++) { const int c = L.cntA[r]; if (c > 0) { L.map[r] = m0; L.cntB[m0] = c; L.depB[m0] = 0; m0++; } else L.map[r] = 0; }
            L.misc[0] = m0;
        }
        __syncthreads();
        m = L.misc[0];
        keys.each(n, tid, [&](int, unsigned&, int& knode) { knode = L.map[knode]; });
        __syncthreads();
    }

    // ---- C. passes
    // A dividing pass costs five workgroup barriers (it was twelve: the scratch of the NEXT pass is cleared while this one's keys move, the two
    // prefix sums of a pass are one, the expandable-node counter alternates between two words so that nobody waits for its reset).
    bool modeB = false, finished = false;
    int par = 0;                                                            // which of s_misc[3] / s_misc[4] counts this pass's expandable nodes
    if (K > 0) {                                                            // the checks behind pass K (:669-673)
        if (m >= N || m == jumpPrev) finished = true;
        else if ((m + 3 * jumpExp) > N) modeB = true;
    }
    { const int nz = max(4 * m, D > 0 ? (g.nIni << (2 * D)) : 0); for (int i = tid; i < nz; i += QT_T) s_cc[i] = 0; }
    if (tid == 0) { s_misc[1] = 0; s_misc[2] = 0x7fffffff; s_misc[3] = 0; s_misc[4] = 0; }
    __syncthreads();
    if (!finished)
    for (int guard = 0; guard < 4096; guard++) {
        keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {
            const int p = knode;
            if (cnt[p] > 1) atomicAdd(&s_cc[4 * p + qt_digit(kcode, dep[p])], 1);
        });
        __syncthreads();
        int Ctot, nsplit;
        if (!modeB) {
            // every node with more than one key is divided, walking the list front to back (:606-665)
            // one prefix sum for both: children created before node p in the low 17 bits (<= 4 x 16383), undivided nodes before it above them
            for (int p = tid; p < m; p += QT_T) {
                const bool e = cnt[p] > 1;
                s_a[p] = e ? ((s_cc[4 * p] > 0) + (s_cc[4 * p + 1] > 0) + (s_cc[4 * p + 2] > 0) + (s_cc[4 * p + 3] > 0)) : (1 << 17);
                s_split[p] = e ? 1 : 0;
            }
            const int tot = qt_block_exscan<false>(s_a, m, s_scratch, tid);      // thread t scans what thread t wrote and reads it back below
            Ctot = tot & 0x1ffff;
            nsplit = m - (tot >> 17);
            for (int p = tid; p < m; p += QT_T) {
                if (s_split[p]) {
                    int q = s_a[p] & 0x1ffff;
                    for (int d = 0; d < 4; d++) { const int c = s_cc[4 * p + d]; if (c > 0) { const int pos = Ctot - 1 - q; q++; cnt2[pos] = c; dep2[pos] = (unsigned char)(dep[p] + 1); s_map[4 * p + d] = (unsigned short)pos; } }
                } else { const int pos = Ctot + (s_a[p] >> 17); cnt2[pos] = cnt[p]; dep2[pos] = dep[p]; s_map[4 * p] = (unsigned short)pos; }
            }
        } else {
            // final phase: expandable nodes sorted by (size, creation) ascending, processed from the back (:684-732)
            // The expandable nodes are first compacted in list order (one prefix sum), then ranked among themselves: late in the replay most
            // nodes hold a single key, and ranking every node against the whole list (m^2 / 256 steps per thread) was the longest stretch of a
            // level-0 workgroup between two barriers.
            for (int p = tid; p < m; p += QT_T) { s_split[p] = 0; s_best[p] = cnt[p] > 1 ? 1 : 0; }
            const int E = qt_block_exscan(s_best, m, s_scratch, tid);     // thread t scans the flags thread t wrote
            // (size, earlier in the list) as ONE number: size << 14 | 16383 - position (positions < 16384 by the context's limits, sizes < 2^18 checked here),
            // so a node's rank is the number of larger keys - a compare and an add per pair instead of three compares and their logic.  This loop
            // is E steps per thread however it is cut: 11 of a level-0 workgroup's 40 us at E = 255 before, ~2 after.
            bool packed = n < (1 << 18);
#ifdef ORBHIP_TEST_HOOKS      // (the CPU emulation build only: ORBHIP_TEST_QT_UNPACKED=1 runs the three-compare form that levels with >= 2^18 candidates take)
            { const char* e = getenv("ORBHIP_TEST_QT_UNPACKED"); if (e && *e == '1') packed = false; }
#endif
            for (int p = tid; p < m; p += QT_T) { const int c = cnt[p]; if (c > 1) { const int j = s_best[p]; s_b[j] = packed ? (int)(((unsigned)c << 14) | (unsigned)(16383 - j)) : c; s_a[j] = p; } }
            __syncthreads();
            for (int j = tid; j < E; j += QT_T) {
                int rank = 0;
                if (packed) {
                    const unsigned key = (unsigned)s_b[j];
                    int o = 0;
                    for (; o + 8 <= E; o += 8) {
                        unsigned k8[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) k8[u] = (unsigned)s_b[o + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) rank += k8[u] > key;
                    }
                    for (; o < E; o++) rank += (unsigned)s_b[o] > key;
                } else {
                    const int c = s_b[j];
                    for (int o = 0; o < E; o++) { const int co = s_b[o]; rank += (co > c || (co == c && o < j)); }
                }
                s_sidx[rank] = s_a[j];
            }
            __syncthreads();
            for (int j = tid; j < E; j += QT_T) {
                const int p = s_sidx[j];
                s_a[j] = (s_cc[4 * p] > 0) + (s_cc[4 * p + 1] > 0) + (s_cc[4 * p + 2] > 0) + (s_cc[4 * p + 3] > 0);
                s_b[j] = s_a[j];
            }
            __syncthreads();
            const int Call = qt_block_exscan(s_a, E, s_scratch, tid);     // s_a[j] = children created before sorted node j
            for (int j0 = 0; j0 < E; j0 += QT_T) {                         // list size after processing sorted node j
                const int j = j0 + tid;
                const bool hit = j < E && m + s_a[j] + s_b[j] - (j + 1) >= N;
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);   // first position where `break` fires (:730-731): the wave's first hit, one atomic per wave
                if (hm && lane == 0) atomicMin(&s_misc[2], j0 + 64 * wave + __builtin_ctzll(hm));
            }
            __syncthreads();
            const int jstar = s_misc[2];
            nsplit = (jstar == 0x7fffffff) ? E : jstar + 1;
            Ctot = (nsplit == E) ? Call : s_a[nsplit];
            __syncthreads();
            for (int j = tid; j < nsplit; j += QT_T) s_split[s_sidx[j]] = 1 + j;
            __syncthreads();
            for (int p = tid; p < m; p += QT_T) s_best[p] = s_split[p] ? 0 : 1;
            __syncthreads();
            qt_block_exscan(s_best, m, s_scratch, tid);
            for (int p = tid; p < m; p += QT_T) {
                if (s_split[p]) {
                    int q = s_a[s_split[p] - 1];
                    for (int d = 0; d < 4; d++) { const int c = s_cc[4 * p + d]; if (c > 0) { const int pos = Ctot - 1 - q; q++; cnt2[pos] = c; dep2[pos] = (unsigned char)(dep[p] + 1); s_map[4 * p + d] = (unsigned short)pos; } }
                } else { const int pos = Ctot + s_best[p]; cnt2[pos] = cnt[p]; dep2[pos] = dep[p]; s_map[4 * p] = (unsigned short)pos; }
            }
        }
        __syncthreads();
        const int m2 = Ctot + (m - nsplit);
        keys.each(n, tid, [&](int, unsigned& kcode, int& knode) {
            const int p = knode;
            knode = s_split[p] ? s_map[4 * p + qt_digit(kcode, dep[p])] : s_map[4 * p];
        });
        int nexp = 0;
        for (int p = tid; p < m2; p += QT_T) nexp += cnt2[p] > 1;
        nexp = __builtin_amdgcn_readlane(qt_wave_incl_scan(nexp, lane), 63);      // (256 atomics onto one word cost 2 us)
        if (nexp && lane == 0) atomicAdd(&s_misc[3 + par], nexp);
        // the next pass's scratch: nobody reads s_cc, s_misc[1..2] or the other counter word any more (their readers are behind the barrier above)
        for (int i = tid; i < 4 * m2; i += QT_T) s_cc[i] = 0;
        if (tid == 0) { s_misc[1] = 0; s_misc[2] = 0x7fffffff; s_misc[3 + (par ^ 1)] = 0; }
        __syncthreads();
        const int nToExpand = s_misc[3 + par];
        par ^= 1;
        int* t = cnt; cnt = cnt2; cnt2 = t; unsigned char* td = dep; dep = dep2; dep2 = td;
        const int prev = m; m = m2;
        if (m >= N || m == prev) break;                                    // :669-672 / :734-735
        if (!modeB && (m + 3 * nToExpand) > N) modeB = true;               // :673
    }

    // ---- D. best response per leaf, first wins (:744-760); list order = output order
    for (int p = tid; p < m; p += QT_T) s_best[p] = 0;
    __syncthreads();
    keys.each(n, tid, [&](int k, unsigned&, int& knode) {                  // the FAST score is read once, here
        atomicMax((unsigned*)&s_best[knode], (qval[k] & 0xff000000u) | (0xFFFFFFu - (unsigned)k));
    });
    __syncthreads();
    unsigned* outk = P.lvl_kp + (long long)frame * P.lvl_kp_per_frame + g.kp_off;
    const int mout = min(m, g.kp_cap);
    for (int p = tid; p < mout; p += QT_T) {
        const unsigned k = 0xFFFFFFu - ((unsigned)s_best[p] & 0xFFFFFFu);
        const unsigned v = qval[k];
        const unsigned x = (v & 0xfff) + ORBHIP_EDGE, y = ((v >> 12) & 0xfff) + ORBHIP_EDGE;   // :843-844
        outk[p] = x | (y << 12) | (v & 0xff000000u);
    }
    if (tid == 0) P.lvl_n[frame * P.nlevels + level] = mout;
    (void)wave; (void)lane;
}

// DistributeOctTree of one (frame, level) by the calling workgroup (QT_T threads, `lds` = its dynamic LDS block)
__device__ __forceinline__ void quadtree_level(const ExtractParams& P, int frame, int level, int* lds)
{
    const int tid = threadIdx.x;
    const LevelGeom g = P.geom[level];
    const int maxn = P.qt_maxn;
    QtLds L;
    L.cntA = lds; L.cntB = L.cntA + maxn;
    L.cc = L.cntB + maxn; L.a = L.cc + 4 * maxn; L.b = L.a + maxn; L.sidx = L.b + maxn; L.split = L.sidx + maxn;
    L.best = L.split + maxn;
    L.pref = L.a; L.slot = L.pref + (P.qt_maxcells + 1);       // phase A only: the region of the five per-node arrays (first written three barriers after phase A's last read)
    const int shared = max(2 * P.qt_maxcells + 1, 5 * maxn);
    L.scratch = L.a + shared + 1; L.misc = L.scratch - 1 + P.qt_scr;      // scratch[-1] holds the scan total
    L.map = reinterpret_cast<unsigned short*>(L.misc + 16);
    L.depA = reinterpret_cast<unsigned char*>(L.map + 4 * maxn); L.depB = L.depA + maxn;

    const int* ccount = P.cell_count + (long long)frame * P.ncells_total + g.cell_first;
    for (int c = tid; c < g.ncells; c += QT_T) { L.pref[c] = ccount[c]; L.slot[c] = P.cells[g.cell_first + c].cand_idx; }
    for (int r = tid; r < maxn; r += QT_T) L.cntA[r] = 0;
    // depth of the per-cell histogram behind the regular-pass jump of qt_replay: nIni * 4^D counters must fit the child-count array and one prefix sum
    int D = 0;
    { const int cap = min(4 * maxn, 64 * (P.qt_scr - 2)); while (D < 4 && D < ORBHIP_QT_DEPTH && (g.nIni << (2 * (D + 1))) <= cap) D++; }
    for (int i = tid; i < (D > 0 ? (g.nIni << (2 * D)) : 0); i += QT_T) L.cc[i] = 0;
    if (tid < 16) L.misc[tid] = 0;
    __syncthreads();
    int n = qt_block_exscan(L.pref, g.ncells, L.scratch, tid);
    n = min(n, g.cand_total_cap);
    unsigned* qval = P.qt_val + (long long)frame * P.qt_per_frame + g.cand_total_off;
    if (n <= QT_REGKEYS) {
        QtKeysReg keys;
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) { keys.code[j] = 0; keys.node[j] = 0; }
        qt_replay(P, g, L, keys, qval, n, frame, level, tid, D);
    } else {
        QtKeysHbm keys; keys.code = P.qt_code + (long long)frame * P.qt_per_frame + g.cand_total_off;
        keys.node = P.qt_node + (long long)frame * P.qt_per_frame + g.cand_total_off;
        qt_replay(P, g, L, keys, qval, n, frame, level, tid, D);
    }
}

__global__ __launch_bounds__(QT_T) void k_quadtree(ExtractParams P)
{
    __builtin_amdgcn_s_setprio(3);          // latency-bound: win issue arbitration against the VALU-bound blur running beside it
    // level-major ids (still frame == id mod 8 for the XCD affinity): workgroups are dispatched in id order and a level-0
    // workgroup runs ~4x longer than a level-7 one, so the long ones start first and the short ones fill the tail
    int level, frame;
    if (P.nframes < 8) { level = (int)blockIdx.x / P.nframes; frame = (int)blockIdx.x - level * P.nframes; }      // (xcd_grid's rule for a handful of frames)
    else {
        const int nfg = (P.nframes + 7) >> 3, jj = (int)blockIdx.x >> 3;
        level = jj / nfg;
        frame = (jj - level * nfg) * 8 + ((int)blockIdx.x & 7);
    }
    if (frame >= P.nframes) return;
    HIP_DYNAMIC_SHARED(int, lds)
    quadtree_level(P, frame + P.frame0, level, lds);
}

// A handful of frames (the drop-in's single-image call): the blur and the quadtree in ONE launch.  They do not depend on each other (both follow
// FAST / the pyramid, both precede the descriptor kernel); as two launches on one stream a frame's eight quadtree workgroups (59 us at level 0 of a
// 1241 x 376 frame) ran behind the blur's 11 us, on two streams the event hops cost more than the blur.  The quadtree workgroups take the first
// ids (level-major: the longest start first), the blur tiles fill the rest of the chip beside them.
__global__ __launch_bounds__(256) void k_blur_quadtree(ExtractParams P)
{
    __shared__ __attribute__((aligned(16))) unsigned s_in[32 * BM_IN_DW];
    __shared__ unsigned s_out[BM_ROWS * BM_OUT_DW];
    __shared__ __attribute__((aligned(16))) unsigned s_band[3 * 64 * 4];
    HIP_DYNAMIC_SHARED(int, lds)
    const int nqt = P.nlevels * P.nframes, id = (int)blockIdx.x;
    if (id < nqt) {
        __builtin_amdgcn_s_setprio(3);
        quadtree_level(P, P.frame0 + id % P.nframes, id / P.nframes, lds);
    } else {
        const int b = id - nqt;
        blur_mfma_tile(P, b / P.nframes, P.frame0 + b % P.nframes, s_in, s_out, s_band);
    }
}

void orbhip_launch_blur_quadtree(const ExtractParams& P, int nframes, hipStream_t s)
{   // requires the matrix-core blur (P.blur_band); the caller falls back to the two launches otherwise
    ExtractParams Q = P; Q.nframes = nframes;
    const size_t lds = orbhip_quadtree_lds_bytes(P.qt_maxn, P.qt_maxcells);
    hipLaunchKernelGGL(k_blur_quadtree, dim3((unsigned)((P.nlevels + P.nblur_tiles) * nframes), 1, 1), dim3(256, 1, 1), lds, s, Q);
}

void orbhip_launch_quadtree(const ExtractParams& P, int nframes, hipStream_t s)
{
    ExtractParams Q = P; Q.nframes = nframes;
    const size_t lds = orbhip_quadtree_lds_bytes(P.qt_maxn, P.qt_maxcells);
    hipLaunchKernelGGL(k_quadtree, dim3(xcd_grid(P.nlevels, nframes), 1, 1), dim3(QT_T, 1, 1), lds, s, Q);
}

// ------------------------------------------------------------------------------------------------ describe
// One wavefront per keypoint slot.  IC_Angle over the 749-pixel circular patch (two lanes per row), cv::fastAtan2,
// glibc-style sincosf in double, then 4 rounds of 64 rotated BRIEF tests whose ballots ARE the descriptor words.
__device__ __forceinline__ float dev_fast_atan2(float y, float x)       // OpenCV 3.2 mathfuncs_core.cpp, degrees
{
    const float sc = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc, p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16)); c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16)); c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sincosf, glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf.c restated (|y| < 120): every double op individually rounded.
__device__ __forceinline__ void dev_sincosf(float y, float* sinp, float* cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ff;
    double x = (double)y; int n = 0; double sgn = 1.0; bool neg = false;
    if (top < ((0x3f490fdbu >> 20) & 0x7ff)) {                  // abstop12(y) < abstop12(pi/4)
        if (top < ((0x39800000u >> 20) & 0x7ff)) { *sinp = y; *cosp = 1.0f; return; }   // |y| < 2^-12
    } else {
        const double r = __dmul_rn(x, hpi_inv);
        n = (__double2int_rz(r) + 0x800000) >> 24;
        x = __dsub_rn(x, __dmul_rn((double)n, hpi));
        sgn = (n & 1) ? -1.0 : 1.0;  if (n & 2) sgn = -sgn;     // sign[n&3] = {1,-1,-1,1}
        neg = (n & 2) != 0;                                       // second table: cosine coefficients negated
    }
    const double x2 = __dmul_rn(x, x);
    const double xs = __dmul_rn(x, sgn);
    const double c0 = neg ? -C0 : C0, c1k = neg ? -C1 : C1, c2k = neg ? -C2 : C2, c3k = neg ? -C3 : C3, c4k = neg ? -C4 : C4;
    const double x4 = __dmul_rn(x2, x2), x3 = __dmul_rn(x2, xs);
    const double c2 = __dadd_rn(c3k, __dmul_rn(x2, c4k)), s1 = __dadd_rn(S2, __dmul_rn(x2, S3));
    const double c1 = __dadd_rn(c0, __dmul_rn(x2, c1k));
    const double x5 = __dmul_rn(x3, x2), x6 = __dmul_rn(x4, x2);
    const double s = __dadd_rn(xs, __dmul_rn(x3, S1)), c = __dadd_rn(c1, __dmul_rn(x4, c2k));
    const float sv = __double2float_rn(__dadd_rn(s, __dmul_rn(x5, s1)));
    const float cv = __double2float_rn(__dadd_rn(c, __dmul_rn(x6, c2)));
    if (n & 1) { *cosp = sv; *sinp = cv; } else { *sinp = sv; *cosp = cv; }
}

// wave64 sum with DPP row shifts / broadcasts (4-cycle VALU ops) instead of 24-cycle ds_bpermute shuffles; total in lane 63
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3  -> lane 15 of each row = row total
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
    return __builtin_amdgcn_readlane(v, 63);
}
// cvRound for |x| < 2^22: adding 1.5*2^23 rounds to the nearest integer, ties to even, exactly like cvRound / rint
__device__ __forceinline__ int round_half_even_small(float x) { return __float_as_int(__fadd_rn(x, 12582912.0f)) - 0x4B400000; }

#define DS_WAVES 4
#define DS_KPW 4                       // key point slots per wavefront (3 and 5 measured slower: docs/ROUND_LOG.md, round 3 "describe")
#define DS_WROWS 37                    // blurred window rows: pattern reach is +-18 after rotation
#define DS_WSTRIDE 48                  // bytes per staged window row: 37 used, starting 0..3 bytes into the row (staged from the dword boundary below cx - 18).  12 dwords, not the
                                       // 10 that would do: the texture addresser takes an LDS-DMA pass in groups of four lanes, and a group that straddles two image rows is what
                                       // costs - 9.1 cycles for a pass of 5.33 rows x 48 B against 20.1 for 6.4 rows x 40 B (tools/ta_ubench.hip, profiles/r03_ta_cost_*.txt)
#define DS_PPASS 8                     // orientation patch: 32-bit load passes of 4 rows x 16 lanes
#define DS_WPASSES 7                   // LDS-DMA passes per window: 7 x 64 dwords >= 37 rows x 12 dwords (16 slots x 1792 B = 28 KB per workgroup)
#define DS_WDWORDS (DS_WPASSES * 64)
// One wavefront per DS_KPW consecutive key point slots, in three phases, so that what is per-key-point scalar work in the reference is
// done once per lane instead of once per wave:
//   1. IC_Angle (ORBextractor.cc:77-104) of every slot: the 31 x 31 orientation patch as 31 rows x 8 unaligned dwords (4 loads per
//      lane), circular mask from a table, sum I, sum (u+15) I and row sums by v_dot4_u32_u8, two DPP wave sums; m10 / m01 of slot j
//      are kept by lane j;
//   2. cv::fastAtan2 and the glibc sincosf for all slots at once, lane j = slot j;
//   3. per slot: the 37 x 37 blurred window, brought to LDS by LDS-DMA issued at the very start (one buffer per slot), a and b back by
//      v_readlane, 256 rotated BRIEF tests whose 4 ballots are the descriptor words.
// The rBRIEF pattern, the mask and the DMA layout are loaded / computed once per wave.
__global__ __launch_bounds__(256) void k_describe(ExtractParams P)
{
    __shared__ unsigned s_win[DS_WAVES][DS_KPW][DS_WDWORDS];
    // the rBRIEF pattern (4 KB, [component][round][lane]) is fetched once per workgroup with four coalesced 32-bit loads per thread - as 16-byte
    // loads per lane it cost every wave 4 x 71 cycles of the texture addresser, more than the wave's orientation patches (tools/ta_ubench.hip) -
    // and passes through the first window buffers on its way to the lanes' registers (two barriers, before any wave leaves or stages a window):
    // the workgroup's LDS stays at 28 KB, five workgroups per CU with room to spare
    float* s_pat = reinterpret_cast<float*>(&s_win[0][0][0]);
    static_assert(DS_KPW * DS_WDWORDS >= 16 * 64, "the pattern passes through wave 0's window buffers");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // A workgroup's start is a chain of memory latencies (pattern -> LDS -> barriers, slot metadata, orientation patches), and a wave lives for
    // only ~11 us: the pattern, the metadata and the circle mask are requested together, and the patch loads are issued BEFORE the pattern's
    // two barriers, so the pattern's trip through LDS runs under the patches' latency instead of in front of it (round 3: three latencies -> two).
    // Nobody leaves before the second barrier: a wave without slots (live == false) loads in-bounds dummies and skips the patches.
    int tile, frame;
    bool live = xcd_frame_map((P.lvl_kp_per_frame + DS_WAVES * DS_KPW - 1) / (DS_WAVES * DS_KPW), P.nframes, tile, frame);
    frame = live ? frame + P.frame0 : P.frame0;
    const int slot0 = live ? (tile * DS_WAVES + wave) * DS_KPW : 0;
    live = live && slot0 < P.lvl_kp_per_frame;
    float pv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pv[k] = P.patternf[256 * k + threadIdx.x];
#define U(x) __builtin_amdgcn_readfirstlane(x)
    // Everything the wave needs to know about its slots comes from ONE round of vector loads (lane l = level l, lane j = slot j) and is
    // then passed around with v_readlane: a chain of dependent "load, wait, readfirstlane" steps costs a memory latency each.
    const int lq = min(lane, P.nlevels - 1);
    const int g_kpoff = P.geom[lq].kp_off, g_pitch = P.geom[lq].pitch, g_poff = P.geom[lq].plane_off, g_n = P.lvl_n[frame * P.nlevels + lq];
    const float g_scale = P.geom[lq].scale, g_size = P.geom[lq].kp_size;
    const int mySlot = min(slot0 + lane, P.lvl_kp_per_frame - 1);
    const unsigned myV = P.lvl_kp[(long long)frame * P.lvl_kp_per_frame + mySlot];
#define RL(v, l) __builtin_amdgcn_readlane((v), (l))
#define RLF(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))

    // ---- per-lane constants of the wave
    // orientation patch: lane = (row mod 4, dword of the row); 9 of a row's 16 lanes load the aligned dwords around its 31 bytes, which are
    // then funnel-shifted into place (an unaligned 32-bit load of this shape costs the texture addresser 3.5 x an aligned one).
    // (Round 3 packed the rows 9 lanes apiece - 7 rows per pass, 5 passes instead of 8, the next dword by wave_shl:1: 80 VALU fewer per wave and
    //  bit-exact, but 0.674 -> 0.734 ms: groups of four lanes that straddle two rows cost the texture addresser more than the three passes saved.)
    const int prow = lane >> 4, pd = lane & 15;
    unsigned pmask[DS_PPASS];
#pragma unroll
    for (int q = 0; q < DS_PPASS; q++) pmask[q] = P.ic_mask[q * 64 + lane];        // bytes inside the circle (row 31, dwords 8..15: 0)
#pragma unroll
    for (int k = 0; k < 4; k++) s_pat[256 * k + threadIdx.x] = pv[k];              // (the pattern's loads are the oldest outstanding ones: this waits for them alone)
    const unsigned ucoef = 0x03020100u + 0x04040404u * (unsigned)pd;       // u + 15 of the dword's four bytes
    int wrow[DS_WPASSES], wcol[DS_WPASSES];                                // window DMA: pass k, lane l fills dword 64k + l = (row, dword) of the 12-dword rows
#pragma unroll
    for (int k = 0; k < DS_WPASSES; k++) {
        const int pos = 64 * k + lane, row = (pos * 5462) >> 16;          // pos / 12 for pos < 448
        wrow[k] = min(row, DS_WROWS - 1); wcol[k] = 4 * (pos - 12 * row); // the four dwords past the window re-read the start of its last row (in bounds, never used)
    }

    // ---- 0. slot -> (level, index in the level, output index), lane j for slot j; then wave-uniform copies (SGPRs) of the wave's slots.
    //      Valid slots of a wave have consecutive output indices (slots and outputs are both level-major).
    int myLevel = 0, myOi = -1;
    {
        int obase = 0, koff = RL(g_kpoff, 0), nsel = RL(g_n, 0), tot = nsel;
        for (int l = 1; l < P.nlevels; l++) {
            const int kl = RL(g_kpoff, l), nl = RL(g_n, l), nprev = RL(g_n, l - 1);
            tot += nl;
            if (slot0 + lane >= kl) { myLevel = l; obase += nprev; koff = kl; nsel = nl; }
        }
        if (live && slot0 == 0 && lane == 0) P.out_n[frame] = min(tot, P.out_cap);
        const int i = slot0 + lane - koff;
        if (live && lane < DS_KPW && slot0 + lane < P.lvl_kp_per_frame && i < nsel && obase + i < P.out_cap) myOi = obase + i;
    }
    bool ok[DS_KPW]; int lv[DS_KPW], oi[DS_KPW]; unsigned vv[DS_KPW];
    int oi_first = -1;
#pragma unroll
    for (int j = 0; j < DS_KPW; j++) {
        oi[j] = RL(myOi, j); lv[j] = RL(myLevel, j); vv[j] = (unsigned)RL((int)myV, j);
        ok[j] = oi[j] >= 0;
        if (ok[j] && oi_first < 0) oi_first = oi[j];
    }
    // the orientation patches of the wave's slots: requested here, in front of the pattern's barriers
    unsigned pw[DS_KPW][DS_PPASS];
#pragma unroll
    for (int j = 0; j < DS_KPW; j++) if (ok[j]) {
        const int spitch = lv[j] == 0 ? P.img0_pitch : RL(g_pitch, lv[j]);
        const ORBHIP_GLOBAL uint8_t* img = uniform_ptr(lv[j] == 0 ? P.img0 + (long long)frame * P.img0_frame_stride
                                                                  : P.pyr + (long long)frame * P.plane_frame_bytes + RL(g_poff, lv[j]));
        const int cx = vv[j] & 0xfff, cy = (vv[j] >> 12) & 0xfff;
        const unsigned off = (unsigned)((cy - 15 + prow) * spitch + ((cx - 15) & ~3) + 4 * min(pd, 8));
#pragma unroll
        for (int q = 0; q < DS_PPASS; q++)                             // rows prow + 4q - 15 = -15 .. 16; row 16 is loaded (in bounds) and masked
            pw[j][q] = *reinterpret_cast<const ORBHIP_GLOBAL unsigned*>(img + (off + (unsigned)(4 * q * spitch)));
    }
    __syncthreads();
    float4 pt[4];                                                          // rBRIEF pattern of this lane's 4 tests as floats (x0, y0, x1, y1)
#pragma unroll
    for (int r = 0; r < 4; r++) { pt[r].x = s_pat[(0 * 4 + r) * 64 + lane]; pt[r].y = s_pat[(1 * 4 + r) * 64 + lane]; pt[r].z = s_pat[(2 * 4 + r) * 64 + lane]; pt[r].w = s_pat[(3 * 4 + r) * 64 + lane]; }
    __syncthreads();                                                       // (behind it the window buffers are the windows')
    if (oi_first < 0) return;

    // ---- 1. IC_Angle of every slot.  Every memory request of the wave is issued here, before anything is waited for: the patch loads of
    //      all slots, then the LDS-DMA of all blurred windows (one LDS buffer per slot) - a wave pays the memory latency once
    auto window_dma = [&](int level, unsigned v, unsigned* win) {
        const ORBHIP_GLOBAL uint8_t* blv = uniform_ptr(P.blur + (long long)frame * P.plane_frame_bytes + RL(g_poff, level));
        const int bpitch = RL(g_pitch, level);
        const int cx = v & 0xfff, cy = (v >> 12) & 0xfff;
        const unsigned base = (unsigned)((cy - 18) * bpitch + ((cx - 18) & ~3));        // rows start on a dword (planes and pitches are 64-byte aligned): 40 bytes from there still hold the 37 of the window, and an LDS-DMA pass of this shape costs the texture addresser 20 cycles instead of 25 (tools/ta_ubench.hip)
#pragma unroll
        for (int k = 0; k < DS_WPASSES; k++)
            lds_dma_dword(blv + (base + __umul24((unsigned)wrow[k], (unsigned)bpitch) + (unsigned)wcol[k]), reinterpret_cast<uint8_t*>(win) + 256 * k);
    };
    int M10 = 0, M01 = 0;
    {
#pragma unroll
        for (int j = 0; j < DS_KPW; j++) if (ok[j]) window_dma(lv[j], vv[j], s_win[wave][j]);
#pragma unroll
        for (int j = 0; j < DS_KPW; j++) if (ok[j]) {
            unsigned s = 0, su = 0, cs = 0;                                   // cs = sum over passes of the running row sum = sum_q (8 - q) t_q
            const unsigned sh = (unsigned)((vv[j] & 0xfff) - 15) & 3u;        // (cx - 15) & 3: byte 0 of the shifted dword is column cx - 15 + 4 pd
#pragma unroll
            for (int q = 0; q < DS_PPASS; q++) {
                const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pw[j][q], 0x101, 0xf, 0xf, false);     // row_shl:1 = the next dword of the row
                const unsigned x = __builtin_amdgcn_alignbyte(nxt, pw[j][q], sh) & pmask[q];
                s = __builtin_amdgcn_sad_u8(x, 0u, s);                           // running sum of the masked bytes (v_sad_u8 accumulates: no separate addition)
                su = __builtin_amdgcn_udot4(x, ucoef, su, false);
                cs += s;
            }
            // sum_q (prow + 4q - 15) t_q = (prow - 15) S + 4 (8 S - cs): two additions per pass instead of a 32-bit multiply-add
            const int sv = (prow + 17) * (int)s - 4 * (int)cs;
            const int m10 = wave_sum_dpp((int)su - 15 * (int)s), m01 = wave_sum_dpp(sv);     // sum u*I, sum v*I (wave-uniform)
            if (lane == j) { M10 = m10; M01 = m01; }
        }
    }

    // ---- 2. orientation and its sine / cosine, lane j = slot j
    const float angle_l = dev_fast_atan2((float)M01, (float)M10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a_l, b_l; dev_sincosf(__fmul_rn(angle_l, factorPI), &b_l, &a_l);

    // ---- 3. steered BRIEF (ORBextractor.cc:107-147) per slot on the blurred window in LDS; descriptor word r of the wave's k-th output
    //      is collected in lane 4k + r, so nothing is stored (and no store is waited for) inside the loop
    int dlo_l = 0, dhi_l = 0;                                               // the 64-bit word as two registers: v_writelane_b32 drops a wave-uniform value into one lane
    {
        int later = 0;                                                      // valid slots behind slot j: their windows' loads are the youngest outstanding ones
#pragma unroll
        for (int j = 0; j < DS_KPW; j++) later += ok[j] ? 1 : 0;
#pragma unroll
        for (int j = 0; j < DS_KPW; j++) {
            if (!ok[j]) continue;
            later--;
            // vmcnt counts in order: at most 6 * later loads outstanding  <=>  this slot's window (and everything older) has landed
            if (later >= 3) __builtin_amdgcn_s_waitcnt(ORBHIP_VMCNT(3 * DS_WPASSES));
            else if (later == 2) __builtin_amdgcn_s_waitcnt(ORBHIP_VMCNT(2 * DS_WPASSES));
            else if (later == 1) __builtin_amdgcn_s_waitcnt(ORBHIP_VMCNT(DS_WPASSES));
            else lds_dma_wait();
            __builtin_amdgcn_wave_barrier();
            const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), j)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), j));
            const uint8_t* w8 = reinterpret_cast<const uint8_t*>(s_win[wave][j]) + 18 * DS_WSTRIDE + 18 + (((vv[j] & 0xfff) - 18) & 3);      // key point position inside the window (staged from the dword boundary at or below cx - 18)
            const int dlane = 4 * (oi[j] - oi_first);
            // P.fp_contract (wave-uniform): 0 = x*b + y*a as two roundings (a build with -ffp-contract=off, H3), 1 = the fused forms gcc emits
            // for the reference's own flags, fma(x, b, y*a) and fma(x, a, -(y*b))
            // Both points of a test go through the rotation as one packed-f32 pair (v_pk_mul_f32 / v_pk_add_f32: every product and sum still
            // rounded on its own), and cvRound comes out of the float's own bits: x + (2^23 + 64) is an integer-valued float with ulp 1 for
            // |x| < 64 - the same round-half-to-even as rint (64 is even) - whose low 24 bits are 64 + rint(x).  v_mad_u32_u24 takes exactly
            // those of the row coordinate, so   (64 + iy) * 48 + bits(x') = 48 iy + ix + C   with C = 3072 + 64 + 0x4B000000:
            // one multiply-add and one addition of a wave-uniform constant per point instead of two subtractions, a multiplication and two additions.
            typedef float f2v __attribute__((vector_size(8)));
            const f2v A2 = {a, a}, B2 = {b, b}, MG = {8388672.0f, 8388672.0f};
            const unsigned cbias = 3072u + 64u + 0x4B000000u;
            auto brief = [&](auto fused) {
                int t0[4], t1[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const f2v X = {pt[r].x, pt[r].z}, Y = {pt[r].y, pt[r].w};        // (x0, x1), (y0, y1)
                    f2v fy, fx;
                    if (decltype(fused)::value) {
                        fy[0] = __fmaf_rn(X[0], b, __fmul_rn(Y[0], a)); fy[1] = __fmaf_rn(X[1], b, __fmul_rn(Y[1], a));
                        fx[0] = __fmaf_rn(X[0], a, -__fmul_rn(Y[0], b)); fx[1] = __fmaf_rn(X[1], a, -__fmul_rn(Y[1], b));
                    } else {
                        // two roundings each (H3) whatever flags this file is built with: the pragma pins what the Makefile's -ffp-contract=off says
                        // (hipcc's own default, fast-honor-pragmas, would fuse these into v_pk_fma_f32 and change a descriptor bit per few frames)
#pragma clang fp contract(off)
                        fy = X * B2 + Y * A2; fx = X * A2 - Y * B2;
                    }
                    const f2v ry = fy + MG, rx = fx + MG;
                    const int o0 = (int)(__umul24(__float_as_uint(ry[0]), (unsigned)DS_WSTRIDE) + __float_as_uint(rx[0]) - cbias);
                    const int o1 = (int)(__umul24(__float_as_uint(ry[1]), (unsigned)DS_WSTRIDE) + __float_as_uint(rx[1]) - cbias);
                    t0[r] = w8[o0]; t1[r] = w8[o1];
                }
                // all eight byte reads are out before the first comparison: with the ballots and v_writelane (an asm statement: a scheduling boundary) in the
                // same loop, every round waited for its own two reads - eight LDS round trips per slot one after the other (round 5)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned long long bits = __ballot(t0[r] < t1[r]);       // test 64r+lane -> byte (64r+lane)/8, bit lane%8
                    dlo_l = orbhip_writelane((int)(unsigned)bits, dlane + r, dlo_l);          // 2 VALU instead of compare + 2 moves + 2 selects
                    dhi_l = orbhip_writelane((int)(unsigned)(bits >> 32), dlane + r, dhi_l);
                }
            };
            if (P.fp_contract) brief(std::true_type{}); else brief(std::false_type{});
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- outputs: the wave's descriptors are one contiguous run of 8-byte words, its key point records one per lane
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < DS_KPW; j++) nvalid += ok[j] ? 1 : 0;
    if (lane < 4 * nvalid)
        reinterpret_cast<unsigned long long*>(P.out_desc + ((long long)frame * P.out_cap + oi_first) * 32)[lane] = ((unsigned long long)(unsigned)dhi_l << 32) | (unsigned)dlo_l;
    float myScale = RLF(g_scale, 0), mySize = RLF(g_size, 0);
    for (int l = 1; l < P.nlevels; l++) { const float sl = RLF(g_scale, l), zl = RLF(g_size, l); if (myLevel == l) { myScale = sl; mySize = zl; } }
    if (myOi >= 0) {
        const int cx = myV & 0xfff, cy = (myV >> 12) & 0xfff;
        orbhip_keypoint kp;
        kp.x = __fmul_rn((float)cx, myScale); kp.y = __fmul_rn((float)cy, myScale);     // pt *= scale (:1095-1101); scale[0] == 1
        kp.size = mySize; kp.angle = angle_l; kp.response = (float)(myV >> 24); kp.octave = myLevel; kp.class_id = -1;
        P.out_kp[(long long)frame * P.out_cap + myOi] = kp;
    }
#undef RL
#undef RLF
#undef U
}

void orbhip_launch_describe(const ExtractParams& P, int nframes, hipStream_t s)
{
    ExtractParams Q = P; Q.nframes = nframes;
    hipLaunchKernelGGL(k_describe, dim3(xcd_grid((P.lvl_kp_per_frame + DS_WAVES * DS_KPW - 1) / (DS_WAVES * DS_KPW), nframes), 1, 1), dim3(256, 1, 1), 0, s, Q);
}
