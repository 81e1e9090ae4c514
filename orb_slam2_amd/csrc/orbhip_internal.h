// orbhip_internal.h — layouts shared by the host orchestration (orbhip_api.cpp) and the gfx950 kernels.
//
// HBM layout of one context (fixed W x H, B = max_batch camera slots); all arrays are [frame][...]:
//   planes      u8   per frame: levels 0..L-1, level l at plane_off[l], row pitch[l] (multiple of 64 B).
//               Two plane sets: pyramid (its level-0 slot is only used by the host-image API; the device API
//               reads level 0 straight from the caller's buffer) and blurred.  FAST scores never reach HBM:
//               one wavefront per grid cell computes, suppresses and emits them from LDS.
//   cell_count  i32  per frame: one counter per FAST grid cell of every level
//   cell_cand   u32  per frame: per-cell candidate slots (cand_idx + rank), packed x | y<<12 | score<<24 in
//                    cell-space coordinates (ORBextractor.cc:820-825), row-major inside the cell
//   qt_val/qt_code/qt_node  per frame, per level dense candidate arrays in canonical vToDistributeKeys order
//   lvl_kp      u32  per frame: post-quadtree keypoints per level in list order (kp_off[l] + pos), packed like
//                    cell_cand but in level coordinates; lvl_n i32[L] counts
//   out_kp / out_desc / out_n: final level-major output (ORBextractor.cc:1075-1104), double-buffered per call
//                    parity so the previous frame of every camera slot stays resident for the matcher.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/orbhip.h"

// Every device allocation of the library goes through here.  ORBHIP_POISON=<byte 0..255> (test knob) fills each new allocation with that
// byte, so that a kernel which depends on what a previous owner left in the memory shows up on any box, not only on a box whose memory a
// previous process has dirtied (tests/test_00_device.py runs the pipeline under two poison values).
#include <cstdlib>
inline hipError_t orbhip_dmalloc(void** p, size_t bytes)
{
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    static const int poison = [] { const char* v = getenv("ORBHIP_POISON"); return v && *v ? atoi(v) & 255 : -1; }();
    if (poison < 0) return hipSuccess;
    const hipError_t m = hipMemset(*p, poison, bytes);
    return m != hipSuccess ? m : hipDeviceSynchronize();
}

#define ORBHIP_MAX_LEVELS 16
#define ORBHIP_QT_DEPTH 13          // quadtree path digits kept per candidate (2 bits each)
#define ORBHIP_GRID_COLS 64         // Frame.h:37-38
#define ORBHIP_GRID_ROWS 48
#define ORBHIP_GRID_CELLS (ORBHIP_GRID_COLS * ORBHIP_GRID_ROWS)
#define ORBHIP_HISTO_LENGTH 30      // ORBmatcher.cc:39
#define ORBHIP_TH_LOW 50            // ORBmatcher.cc:38
#define ORBHIP_EDGE 16              // minBorder = EDGE_THRESHOLD-3 (ORBextractor.cc:773)

struct LevelGeom {
    int w, h, pitch, plane_off;                 // level image
    int nCols, nRows, wCell, hCell;             // FAST grid (ORBextractor.cc:781-787)
    int maxBorderX, maxBorderY;                 // cols-16, rows-16 (ORBextractor.cc:775-776)
    int cell_first, ncells;                     // index range in the CellDesc table
    int cand_total_off, cand_total_cap;         // dense per-level candidate arrays (quadtree workspace)
    int nfeat, kp_cap, kp_off;                  // mnFeaturesPerLevel, capacity max(nfeat+3, 4*nIni), offset
    int nIni; float hX;                         // quadtree roots (ORBextractor.cc:543-545)
    float scale, kp_size, inv_scale;            // mvScaleFactor[l], (float)(int)(31*scale), mvInvScaleFactor[l]
    int xtab_off, ytab_off;                     // resize coefficient tables (levels >= 1): {src index, a0 | a1<<16}
    int xgrp_off;                               // PyrGroup table (one entry per 4 output columns)
    int src_w, src_h;                           // size of level l-1
};

struct CellDesc {           // one FAST cell (ORBextractor.cc:789-816)
    short level, skipped;
    short iniX, iniY, maxX, maxY;   // sub-image [iniY,maxY) x [iniX,maxX) in level coordinates
    short shiftX, shiftY;           // j*wCell, i*hCell (added to cell-local coordinates, :822-823)
    int cand_idx;                   // first slot of this cell in cell_cand (per frame)
    int cand_cap;
    int inv_ng;                     // ceil(2^16 / ng), ng = 4-pixel groups per interior row: k_fast_cells divides lane ids by ng with it
};

// four output pixels of cv::resize (k_pyramid_level_g): first source column, v_perm selectors that pick each pixel's two taps out of the
// eight source bytes from sx0 on (second tap already clamped at the right border), coefficient pairs a0 | a1 << 16
struct alignas(16) PyrGroup { int sx0; unsigned sel[4]; unsigned coef[4]; int pad[3]; };

struct TileDesc { short level, x0, y0, pad; };

struct ExtractParams {
    const LevelGeom* geom; int nlevels;
    int frame0, nframes;                                                   // first frame (camera slot) and frame count of this launch group
    const uint8_t* img0; long long img0_frame_stride; int img0_pitch;      // level-0 source
    uint8_t* pyr; uint8_t* blur; long long plane_frame_bytes;
    const CellDesc* cells; int ncells_total;
    int* cell_count; unsigned* cell_cand; long long cand_slots_per_frame;
    unsigned* qt_val; unsigned* qt_code; int* qt_node; long long qt_per_frame;
    unsigned* lvl_kp; int lvl_kp_per_frame; int* lvl_n;
    orbhip_keypoint* out_kp; uint8_t* out_desc; int* out_n; int out_cap;
    const TileDesc* blur_tiles; int nblur_tiles;
    const int4* blur_band;                                                 // k_blur_mfma: band matrices HB1 | HB2 | VB as B operands, [3][64 lanes] x 16 bytes; NULL = k_blur (VALU)
    const int2* xtab; const int2* ytab; const PyrGroup* xgrp;
    const unsigned* ic_mask;                                               // IC_Angle: [32 rows][8 dwords] byte masks of the circular patch (u, v in -15..15, |u| <= umax[|v|])
    const float* patternf;                                                 // 256 x (x0,y0,x1,y1) as floats (rBRIEF pattern, ORBextractor.cc:150-408)
    int iniTh, minTh, blur_round_mode, fp_contract;
    int qt_maxn;                                                           // LDS node capacity of the quadtree kernel
    int qt_maxcells;                                                       // max cells of one level
    int qt_scr;                                                            // ints of scan scratch in the quadtree's LDS layout
    int fc_pstride, fc_prows, fc_sstride, fc_srows, fc_listcap;            // per-wave LDS layout of k_fast_cells (largest cell of the context)
    int fc_cell0, fc_ncells;                                               // cell range of this k_fast_cells launch
    int fc_pbytes, fc_np; const int4* fc_dma;                              // patch region = fc_np LDS-DMA passes of 256 bytes; (row, 4*column) of every (pass, lane)
    // k_pyramid_cascade (a handful of frames: every level in ONE launch): column / row ranges {first, last} of each level that the workgroup of tile
    // column tx / tile row ty computes, [level][tx] and [level][ty]; LDS layout: buffer of the even levels | of the odd levels | x tables | y tables
    const short2* pc_xr; const short2* pc_yr; int pc_ntx, pc_nty, pc_buf0, pc_buf1, pc_xcap, pc_ycap;
};

struct MatchParams {        // SearchForInitialization over camera slots (ORBmatcher.cc:405-520)
    const orbhip_keypoint* kp1; const uint8_t* desc1; const int* n1;    // previous frames [slot][cap], counts [slot]
    const int* n1_lvl0; int lvl_stride;                                 // number of level-0 keypoints of F1: n1_lvl0[slot*lvl_stride]
    const int* list1;                                                   // indices of F1's level-0 keypoints [slot][lvl0_cap]; NULL = identity
    int prev_from_kp1;                                                  // 1: vbPrevMatched starts as F1's keypoint positions (Tracking.cc:590-592)
    const orbhip_keypoint* kp2; const uint8_t* desc2; const int* n2;    // current frames
    int cap;                                      // keypoint capacity per frame (stride of the arrays above)
    float min_x, min_y, max_x, max_y;                                   // Frame::mnMinX .. mnMaxY (Frame.cc:436-464)
    int* grid_start; int* grid_items; float2* grid_xy;   // [slot][GRID_CELLS+1], [slot][cap], [slot][cap]  (Frame.cc:230-245 on F2, level 0 only)
    unsigned* cand; int* ncand; int cand_stride;  // [slot][n1_lvl0_cap][cand_stride]: i2 | dist<<16, canonical order
    unsigned* top;                                // [slot][n1_lvl0_cap][5]: records of the 4 best candidates of the whole list + "more" flag
    int lvl0_cap;
    float* prev;                                  // [slot][cap][2] vbPrevMatched (in/out)
    int* matches12; int* nmatches;                // [slot][cap], [slot]
    int window; float nnratio; int check_ori;
    int slot0;                                    // first camera slot of this launch group
    int grid_all_levels;                          // 0: buckets hold level-0 keypoints only (SearchForInitialization); 1: all keypoints
    int* big_ws;                                  // [slot][orbhip_match_select_ints(cap, lvl0_cap)] when k_match_select's tables do not fit LDS (orbhip_match_select_big); nullptr otherwise
};
size_t orbhip_match_select_ints(int cap, int lvl0_cap);      // ints of k_match_select's per-slot tables
bool orbhip_match_select_big(int cap, int lvl0_cap);         // they exceed the LDS budget: the caller provides MatchParams::big_ws

struct StereoSide {         // device-resident results + pyramid of one extractor context (its last call)
    const orbhip_keypoint* kp; const uint8_t* desc; const int* n;
    const uint8_t* img0; long long img0_frame_stride; int img0_pitch;      // level 0
    const uint8_t* pyr; long long plane_frame_bytes;                       // levels >= 1
};
struct StereoParams {       // Frame::ComputeStereoMatches (Frame.cc:466-640)
    const LevelGeom* geom; StereoSide L, R;
    int cap, im_h;
    int* row_start; int* row_items; int row_cap;                           // [slot][im_h+1], [slot][row_cap]
    float* u_right; float* depth; int* sad;                                // [slot][cap]
    float mbf, maxD;                                                       // maxD = mbf / mb
};
// host <-> device copy of a PINNED host buffer on stream s: a kernel for small sizes (no DMA-engine hand-over), hipMemcpyAsync otherwise
hipError_t orbhip_copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
int orbhip_device_numa_node(int device);            // /sys/bus/pci/devices/<bus id>/numa_node of a HIP device, -1 if unknown
bool orbhip_bind_thread_to_node(int node);          // binds the CALLING thread to the node's CPUs (and remembers the node for the copy helpers)
void orbhip_launch_stereo(const StereoParams& T, int nslots, int max_left, hipStream_t s, bool rows_ready = false);
void orbhip_launch_stereo_rows(const StereoParams& T, int nslots, hipStream_t s);     // the right frame's row table alone (needs only T.R, geom, im_h, cap, row_*)

struct ProjParams {         // projection-guided search core (ORBmatcher.cc:45-129 and :1328-1470), one frame
    const orbhip_keypoint* kp; const uint8_t* desc; const float* u_right; int n;           // the Frame being searched
    float min_x, min_y, max_x, max_y;                                   // Frame::mnMinX .. mnMaxY (Frame.cc:436-464)
    const int* grid_start; const int* grid_items; const float2* grid_xy;                   // 64x48 grid over ALL keypoints
    const orbhip_proj_query* q; const uint8_t* qdesc; int nq;
    // projection on the device (orbhip_project_search_*): with pts != nullptr query i is DERIVED from pts[i] under *proj by k_proj_candidates, which also
    // writes it to q_out[i] (== q: the select kernel reads blocks / angle there; radius < 0 = the point failed a gate of its member)
    const orbhip_map_point* pts; const orbhip_projection* proj; orbhip_proj_query* q_out;
    unsigned* cand; int* ncand; int cand_stride;                                            // per query: i2 | dist<<19 | level<<28, reference order
    unsigned* top;                                                                          // per query: records of the best and second-best candidate under the initial state
    const unsigned char* blocked_in; unsigned char* blocked_out; int* feature_query; int* nmatches; int* events;
    int mode; float nnratio; int th_high, check_ori;
    int* big_ws;                // 4 n ints of device memory when the select kernel's per-feature tables do not fit LDS (orbhip_proj_select_big(n)); nullptr otherwise
};
bool orbhip_proj_select_big(int n);       // the per-feature tables of k_proj_select exceed the LDS budget: the caller provides ProjParams::big_ws
void orbhip_launch_proj(const ProjParams& J, hipStream_t s);
void orbhip_launch_proj_batch(const ProjParams* d_slots, int nslots, int max_nq, int max_n, float gwInv, float ghInv, hipStream_t s);
struct BestParams {
    const orbhip_keypoint* kp; const uint8_t* desc; const float* u_right; const float* inv_level_sigma2;
    const int* grid_start; const int* grid_items; const float2* grid_xy;        // ordered bucket table over ALL key points (k_match_grid, grid_all_levels)
    const orbhip_best_query* q; const uint8_t* qdesc; int nq; int chi2_gate;
    const orbhip_map_point* pts; const orbhip_projection* proj; orbhip_best_query* q_out;      // projection on the device, as in ProjParams (q_out may be nullptr)
    int* best_idx; int* best_dist;
    float min_x, gw_inv;        // left image bound and grid columns per pixel of the table's grid (gw_inv = 0: scan the whole table)
    const unsigned long long* skip; int skip_bit;     // (orbhip_project_best_in_window_shared) bit skip_bit of skip[iq] set: the query is not searched; nullptr: all are
};
void orbhip_launch_best_in_window(const BestParams& B, hipStream_t s);
void orbhip_launch_best_in_window_batch(const BestParams* d_slots, const int* d_pref, int nslots, int nblocks, hipStream_t s);
size_t orbhip_proj_select_lds(int n);

// kernel launchers (orbhip_kernels_extract.hip / orbhip_kernels_match.hip)
void orbhip_launch_pyramid_level(const ExtractParams& P, int level, int w, int h, int mode, int nframes, hipStream_t s);
void orbhip_launch_pyramid_cascade(const ExtractParams& P, int nframes, hipStream_t s);     // levels 1 .. L-1 in one launch (needs P.pc_*)
int orbhip_pyramid_tile_dwords();
int orbhip_blur_mfma_tile_w();
int orbhip_blur_mfma_tile_h();
bool orbhip_pyramid_tile_fits(int src_cols_per_tile, int src_rows_per_tile);
void orbhip_launch_to_gray(const uint8_t* src, long long src_frame_stride, int src_row_stride, uint8_t* dst, long long dst_frame_stride,
                           int dst_pitch, int w, int h, int channels, bool rgb_order, int nframes, hipStream_t s);
int orbhip_pyramid_tile_w();
int orbhip_pyramid_tile_h();
void orbhip_launch_blur(const ExtractParams& P, const int gk[4], int nframes, hipStream_t s, int tile0 = 0, int ntiles = -1);
void orbhip_launch_blur_quadtree(const ExtractParams& P, int nframes, hipStream_t s);      // both in one launch (small batches, matrix-core blur only)
void orbhip_launch_fast_cells(const ExtractParams& P, int nframes, hipStream_t s, int cell0 = 0, int ncells = -1);
void orbhip_launch_quadtree(const ExtractParams& P, int nframes, hipStream_t s);
void orbhip_launch_describe(const ExtractParams& P, int nframes, hipStream_t s);
size_t orbhip_quadtree_lds_bytes(int maxn, int maxcells);
int orbhip_quadtree_scr(int maxn, int maxcells);
void* orbhip_nn_workspace(size_t bytes, hipStream_t s);
// orbhip_collect with one destination per frame (NULL = not wanted): the pool scatters camera c's results straight to row c
orbhip_status orbhip_collect_scatter(orbhip_ctx* c, int ticket, orbhip_keypoint* const* kps, uint8_t* const* desc, int cap, int* const* n_out);

// camera geometry kernels (orbhip_kernels_geom.hip)
struct CameraD { double fx, fy, cx, cy, ifx, ify, k1, k2, p1, p2, k3; };       // mK / mDistCoef widened like cvUndistortPoints does
void orbhip_launch_undistort_points(const CameraD& C, const float* d_xy, int n, float* d_out, hipStream_t s);
void orbhip_launch_undistort_keys(const CameraD& C, const orbhip_keypoint* kp, const int* n, orbhip_keypoint* kp_un, int cap, int nslots, hipStream_t s);
void orbhip_launch_stereo_from_rgbd(const orbhip_keypoint* kp, const orbhip_keypoint* kp_un, const int* n, int cap, const uint8_t* depth, long long frame_stride,
                                    int row_stride, int type, int convert, float factor, float mbf, float* u_right, float* out_depth, int nslots, hipStream_t s);
void orbhip_launch_remap(const uint8_t* src, long long src_frame_stride, int src_row_stride, int src_w, int src_h, const int* qx, const int* qy, int q_pitch,
                         uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h, int nframes, hipStream_t s);

void orbhip_launch_repitch(const uint8_t* src, long long src_frame_stride, int src_row_stride, uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h, int nframes, hipStream_t s);

bool orbhip_launch_hamming_nn(const uint8_t* d_q, int nq, const uint8_t* d_db, long long ndb, long long base,
                              long long* d_best_idx, int* d_best_dist, int* d_second, hipStream_t s, const uint8_t* d_dbx = nullptr);
size_t orbhip_nn_expanded_bytes(long long ndb);
void orbhip_launch_nn_expand(const uint8_t* d_db, long long ndb, uint8_t* d_out, hipStream_t s);
void orbhip_launch_match_grid(const MatchParams& M, int nslots, hipStream_t s);
void orbhip_launch_match_candidates(const MatchParams& M, int nslots, hipStream_t s);
void orbhip_launch_match_select(const MatchParams& M, int nslots, hipStream_t s);

// shared with orbhip_bow.hip
void orbhip_bow_thread_release();
void orbhip_bow_forget_ctx(const orbhip_ctx* ctx);      // frees the per-context BoW workspaces every live vocabulary keeps for ctx (called by orbhip_destroy)
// v_writelane_b32: a wave-uniform value dropped into ONE lane of a register (lane index wave-uniform too).  The compiler has no builtin for
// it, so the device pass spells the instruction; every other pass (hipcc's host pass, the test emulation) sees the plain selection.
__device__ __forceinline__ int orbhip_writelane(int v, int dst_lane, int old)
{
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(v), "s"(dst_lane) : "m0");      // one SGPR operand on the constant bus: the lane select goes through M0
    return old;
#else
    return (int)__lane_id() == dst_lane ? v : old;
#endif
}
// ---- the calling thread's arena for the host-pointer ("stateless") matcher entry points: every array of a call is laid out in ONE device
// allocation (grow-only, per thread) whose head mirrors a pinned host block, so that all inputs travel in one copy and all outputs in one copy
#include <vector>
#include <algorithm>
struct OrbXfer { size_t off; const void* src; void* dst; size_t bytes_in, bytes_out; };
extern thread_local void* orbhip_tl_scratch; extern thread_local size_t orbhip_tl_scratch_bytes; extern thread_local int orbhip_tl_scratch_dev;
extern thread_local std::vector<OrbXfer> orbhip_tl_xfers;
struct Arena {
    uint8_t* base = nullptr; size_t off = 0;
    template <typename T> void take(T** p, size_t count) { *p = reinterpret_cast<T*>(base + off); off += (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255; }
    // a buffer with a host side: n_in elements are uploaded from src before the kernels, n_out elements downloaded to dst after them.
    // Take every such buffer before the pure scratch ones so that all of them travel in ONE copy each way (arena_upload / arena_download).
    template <typename T> void io(T** p, size_t count, const T* src, size_t n_in, T* dst = nullptr, size_t n_out = 0)
    {
        const size_t o = off;
        take(p, count);
        if (base && ((src && n_in) || (dst && n_out))) orbhip_tl_xfers.push_back(OrbXfer{o, src && n_in ? src : nullptr, dst && n_out ? dst : nullptr, n_in * sizeof(T), n_out * sizeof(T)});
    }
};
hipError_t orbhip_arena_reserve(int device, size_t bytes);
// floor > 0: the layout starts there and what the thread's previous call left below it stays valid (orbhip_project_best_in_window_held) - such a layout
// never reallocates (hipErrorOutOfMemory instead); every ordinary layout (floor 0) ends the validity of what was held (orbhip_tl_held_valid)
extern thread_local bool orbhip_tl_held_valid;
template <typename Layout> static hipError_t arena_layout(int device, Layout layout, size_t floor = 0)
{
    orbhip_tl_xfers.clear();
    if (floor == 0) orbhip_tl_held_valid = false;
    Arena dry; dry.off = floor; layout(dry);                  // first pass: sizes only (no base: nothing is logged)
    if (floor && (orbhip_tl_scratch_dev != device || orbhip_tl_scratch_bytes < dry.off)) return hipErrorOutOfMemory;
    const hipError_t e = orbhip_arena_reserve(device, dry.off); if (e != hipSuccess) return e;
    Arena real; real.base = static_cast<uint8_t*>(orbhip_tl_scratch); real.off = floor; layout(real);
    return hipSuccess;
}
hipError_t arena_upload(hipStream_t s);       // pageable -> pinned gather on the host, ONE host-to-device copy
hipError_t arena_download(hipStream_t s);     // ONE device-to-host copy, synchronises s, pinned -> pageable scatter
hipStream_t orbhip_thread_stream(int device); // the calling thread's own non-blocking stream for these calls
// time the calling thread has spent inside the library's entry points (orbhip_thread_api_ms): lets a caller's measurement separate the library from its own code
#include <chrono>
extern thread_local double orbhip_tl_api_ms; extern thread_local int orbhip_tl_api_depth;
struct OrbApiTimer {
    std::chrono::steady_clock::time_point t0;
    OrbApiTimer() : t0(std::chrono::steady_clock::now()) { orbhip_tl_api_depth++; }
    ~OrbApiTimer() { if (--orbhip_tl_api_depth == 0) orbhip_tl_api_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
void orbhip_touch_thread_caches();      // makes sure the calling thread's cache holder exists (its destructor releases the caches of worker threads)
orbhip_status orbhip_set_error(orbhip_status st, const char* fmt, ...);
void orbhip_internal_outputs(orbhip_ctx* c, const uint8_t** d_desc, const int** d_n, int* cap, int* last_nimg, int* device, hipStream_t* s);
