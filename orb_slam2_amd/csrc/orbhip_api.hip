// orbhip_api.hip — host orchestration and the C ABI of liborbhip.so (include/orbhip.h).
//
// Host side of the drop-in: what ORBextractor::ORBextractor computes once (scale tables, features per level,
// ORBextractor.cc:410-470) plus everything the reference recomputes per frame although it only depends on the image
// size (level sizes :1111-1112, FAST cell grid :773-806, cv::resize coefficient tables, Gaussian kernel) is computed
// here at context creation and uploaded; per call the host only enqueues kernels on the context's HIP stream.
// Float expressions mirror the reference's types step by step (file built with -ffp-contract=off).
#include "orbhip_internal.h"
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

static thread_local std::string g_err;
static orbhip_status fail(orbhip_status st, const char* fmt, ...)
{
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf; return st;
}
orbhip_status orbhip_set_error(orbhip_status st, const char* fmt, ...)      // for the other translation units (orbhip_bow.hip)
{
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf; return st;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ORBHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static inline int cvRoundF(float v) { return (int)lrintf(v); }         // round-half-even, like cvRound
static inline int cvRoundD(double v) { return (int)lrint(v); }
static inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
static inline short satShort(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

enum { K_PYRAMID = 0, K_FAST, K_BLUR, K_QUADTREE, K_DESCRIBE, K_MGRID, K_MCAND, K_MSELECT, K_UNDISTORT, K_REMAP, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"k_pyramid_level", "k_fast_cells", "k_blur", "k_quadtree", "k_describe",
                                            "k_match_grid", "k_match_candidates", "k_match_select", "k_undistort_keys", "k_remap"};

struct ProfSpan { int k; hipEvent_t a, b; int counts; };

// One batch of the pipelined host-buffer path (orbhip_submit / orbhip_collect, orbhip_extract_batch): its own pinned input mirror,
// device input planes and pinned output mirrors, so that up to ORBHIP_RING batches are in flight; a batch is cut into chunks of
// camera slots and chunk k+1 uploads while chunk k computes and chunk k-1 downloads.
#define ORBHIP_RING 3
#define ORBHIP_MAX_CHUNKS 16
struct HostSet {
    uint8_t* d_packed = nullptr; size_t packed_bytes = 0;     // pinned caller images land here as they are (rows `stride` apart) and are re-pitched on the device
    uint8_t* d_in = nullptr; uint8_t* h_in = nullptr; orbhip_keypoint* h_kp = nullptr; uint8_t* h_desc = nullptr; int* h_n = nullptr; bool owned = false;
    uint8_t* h_block = nullptr;                               // h_n | h_kp | h_desc are parts of this one pinned allocation (same layout as the device block)
    bool busy = false; int ticket = -1, nimg = 0, out_buf = -1, nchunks = 0, chunk_f0[ORBHIP_MAX_CHUNKS + 1] = {0};
    hipEvent_t ev_h2d[ORBHIP_MAX_CHUNKS] = {nullptr}, ev_k[ORBHIP_MAX_CHUNKS] = {nullptr}, ev_d2h[ORBHIP_MAX_CHUNKS] = {nullptr};
    // outputs that went straight into the caller's pinned buffers by DMA (nothing left to copy at collect time)
    orbhip_keypoint* direct_kp = nullptr; uint8_t* direct_desc = nullptr; int direct_cap = 0;
};

struct orbhip_ctx {
    orbhip_config cfg; int L = 0, B = 0, fp_contract = 0;
    std::vector<LevelGeom> geom; std::vector<float> sf, isf, s2, is2; std::vector<int> nfeat;
    std::vector<CellDesc> cells; std::vector<TileDesc> blur_tiles; std::vector<int2> xtab, ytab; std::vector<PyrGroup> xgrp; std::vector<char> pyr_staged;
    int gk[4] = {0, 0, 0, 0};
    hipStream_t stream = nullptr; bool own_stream = false;
    std::vector<hipStream_t> xstreams; std::vector<hipEvent_t> xevents; hipEvent_t ev_fork = nullptr;     // extra streams of a multi-stream context
    int fc_maxpw = 0, fc_maxph = 0;
    long long plane_frame_bytes = 0, cand_slots_per_frame = 0, qt_per_frame = 0; int lvl_kp_per_frame = 0, out_cap = 0, qt_maxn = 0, qt_maxcells = 0, lvl0_cap = 0;
    // device
    LevelGeom* d_geom = nullptr; CellDesc* d_cells = nullptr; TileDesc* d_tiles = nullptr; int2* d_xtab = nullptr; int2* d_ytab = nullptr; float* d_pattern = nullptr; int4* d_fc_dma = nullptr; int fc_np = 0; PyrGroup* d_xgrp = nullptr; unsigned* d_ic_mask = nullptr; int4* d_blur_band = nullptr; bool blur_mfma = true;
    uint8_t* d_pyr = nullptr; uint8_t* d_blur = nullptr; int* d_cell_count = nullptr; unsigned* d_cell_cand = nullptr;
    unsigned* d_qt_val = nullptr; unsigned* d_qt_code = nullptr; int* d_qt_node = nullptr; unsigned* d_lvl_kp = nullptr;
    // outputs are triple-buffered: batch t writes buffer t%3 while the matcher of batch t-1 (own stream) still reads buffers (t-1)%3 and (t-2)%3
    int* d_lvl_n[3] = {nullptr, nullptr, nullptr}; orbhip_keypoint* d_out_kp[3] = {nullptr, nullptr, nullptr}; uint8_t* d_out_desc[3] = {nullptr, nullptr, nullptr}; int* d_out_n[3] = {nullptr, nullptr, nullptr};
    // the three output arrays of a buffer are carved from ONE allocation ([counts | key points | descriptors], 256-byte aligned parts): the whole
    // result of a small batch is one device-to-host copy instead of three (a single-frame call spent 60 us between its second and third copy)
    uint8_t* d_out_block[3] = {nullptr, nullptr, nullptr}; size_t out_off_kp = 0, out_off_desc = 0, out_block_bytes = 0; uint8_t* h_block = nullptr;
    hipStream_t bstream = nullptr, bstream_host = nullptr; hipEvent_t ev_pyr = nullptr, ev_blur = nullptr;      // blur runs beside FAST + quadtree (independent until describe)
    // k_pyramid_cascade (every level in one launch, used for a handful of frames): per level the column ranges of each tile column and the row ranges of
    // each tile row, LDS layout sizes; pc_ok = the context's shape fits
    short2* d_pc_xr = nullptr; short2* d_pc_yr = nullptr; int pc_ntx = 0, pc_nty = 0, pc_buf0 = 0, pc_buf1 = 0, pc_xcap = 0, pc_ycap = 0; bool pc_ok = false;
    hipStream_t mstream = nullptr; hipEvent_t ev_extract = nullptr; hipEvent_t ev_match[3] = {nullptr, nullptr, nullptr}; bool match_pending[3] = {false, false, false};
    int cur = 0; int last_nimg = 0; bool last_matched = false; bool last_from_host = false;
    // Frame epilogues.  ORB_SLAM2 calls one image at a time and follows every extraction with the same steps (Frame.cc:61-117, Tracking.cc:867-928,
    // 1143-1193): the right image's row table for ComputeStereoMatches, the 64x48 feature grid for the projection searches.  Both depend on
    // nothing but the extraction's own results, so once a context has seen such a follow-up it enqueues them BEHIND the result download of every
    // single-image call: they run while the host is still copying key points out, and the follow-up call finds them done instead of launching
    // them on its critical path (18 us each of a stereo frame's ~0.9 ms).  Learned per context (a monocular extractor never pays for a row table).
    bool want_fgrid = false, want_rrows = false, fgrid_valid = false, rrows_valid = false; int fgrid_cur = -1, rrows_cur = -1;
    int* d_fgrid_start = nullptr; int* d_fgrid_items = nullptr; float2* d_fgrid_xy = nullptr; int* d_rrow_start = nullptr; int* d_rrow_items = nullptr; int rrow_cap = 0;
    hipEvent_t ev_epilogue = nullptr;
    bool pair_mode = false; float* h_st = nullptr; hipEvent_t ev_stereo = nullptr;           // the last call was orbhip_extract_stereo: slot 0 = the frame (left image), slot 1 = its right image; pinned mirror of [mvuRight | mvDepth]
    std::vector<int> last_n; bool last_n_valid = false;      // key point counts of the last call as already delivered to the host (the call's results were waited for)
    // host-buffer API staging: one contiguous device input buffer + pinned host mirrors (single bulk copies instead of per-frame pageable copies)
    bool serial = false;      // ORBHIP_SERIAL=1 (profiling aid): every kernel on the main stream, nothing overlaps - per-kernel times are standalone times
    uint8_t* d_in = nullptr; uint8_t* h_in = nullptr; uint8_t* d_col = nullptr; uint8_t* h_col = nullptr; size_t col_bytes = 0; orbhip_keypoint* h_kp = nullptr; uint8_t* h_desc = nullptr; int* h_n = nullptr; int in_pitch = 0;
    // stereo (Frame::ComputeStereoMatches): level-0 source of the last call + lazily allocated workspace on the LEFT context
    const uint8_t* last_img0 = nullptr; long long last_img0_fstride = 0; int last_img0_pitch = 0;
    int* d_st_rowstart = nullptr; int* d_st_rowitems = nullptr; int st_rowcap = 0; float* d_st_u = nullptr; float* d_st_depth = nullptr; int* d_st_sad = nullptr;
    // matcher workspace
    int* d_grid_start = nullptr; int* d_grid_items = nullptr; float2* d_grid_xy = nullptr; unsigned* d_cand = nullptr; unsigned* d_top = nullptr; int* d_ncand = nullptr; float* d_prev = nullptr; int* d_m12 = nullptr; int* d_nm = nullptr;
    // camera geometry: undistorted key points of a distorted camera, rectification maps of a raw stereo camera
    orbhip_bounds bounds = {0, 0, 0, 0}; bool distorted = false; CameraD cam = {}; orbhip_keypoint* d_out_kpun[3] = {nullptr, nullptr, nullptr}; orbhip_keypoint* h_kpun = nullptr;
    int* d_map_x = nullptr; int* d_map_y = nullptr; int src_w = 0, src_h = 0, raw_pitch = 0; uint8_t* d_raw = nullptr; uint8_t* h_raw = nullptr; uint8_t* d_depth = nullptr; size_t depth_bytes = 0; const float* d_last_uright = nullptr; float* d_ucols = nullptr; int* d_match_ws = nullptr; float* h_ucols = nullptr; hipEvent_t ev_ucols = nullptr; bool ucols_pending = false;   // mvuRight [slot][out_cap] of the last stereo / RGB-D step
    // pipelined host-buffer path
    HostSet sets[ORBHIP_RING]; hipStream_t hstream = nullptr, dstream = nullptr; int next_ticket = 0, oldest_ticket = 0, ticket_set[ORBHIP_RING] = {0, 0, 0}; const uint8_t* last_d_in = nullptr; bool plane0_dirty = false;   // plane0_dirty: set 0's level-0 plane was last written by an un-ticketed entry (colour / rectify)
    // profiling
    bool prof = false; std::vector<ProfSpan> pending; std::vector<hipEvent_t> pool; double tot_ms[K_COUNT] = {0}; long long launches[K_COUNT] = {0};
};

static const signed char kPatternHost[256 * 4] = {
#include "brief_pattern_31.inc"
};

// ---------------------------------------------------------------------------------------------- profiling helpers
static hipEvent_t prof_event(orbhip_ctx* c)
{
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
struct ProfScope {       // counts = 0: a further part of a kernel that is launched in pieces (its time adds up, the launch count does not)
    orbhip_ctx* c; int k; hipStream_t s; int counts; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(orbhip_ctx* c_, int k_, hipStream_t s_, int counts_ = 1) : c(c_), k(k_), s(s_), counts(counts_) { if (c->prof) { a = prof_event(c); b = prof_event(c); (void)hipEventRecord(a, s); } }
    ~ProfScope() { if (c->prof) { (void)hipEventRecord(b, s); c->pending.push_back(ProfSpan{k, a, b, counts}); } }
};
static void prof_collect(orbhip_ctx* c)
{
    for (auto& s : c->pending) {
        (void)hipEventSynchronize(s.b);
        float ms = 0; (void)hipEventElapsedTime(&ms, s.a, s.b);
        c->tot_ms[s.k] += ms; c->launches[s.k] += s.counts;
        c->pool.push_back(s.a); c->pool.push_back(s.b);
    }
    c->pending.clear();
}

// ---------------------------------------------------------------------------------------------- creation
template <typename T> static hipError_t dalloc(T** p, size_t count) { return orbhip_dmalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)); }
// Per-thread, grow-only device scratch for the host-pointer matcher entry points: one hipMalloc the first time (or when a call
// needs more), none afterwards — hipMalloc / hipFree cost more than the kernels of a single-frame call.  (Arena: orbhip_internal.h)
thread_local void* orbhip_tl_scratch = nullptr; thread_local size_t orbhip_tl_scratch_bytes = 0; thread_local int orbhip_tl_scratch_dev = -1;
thread_local std::vector<OrbXfer> orbhip_tl_xfers;           // host <-> arena transfers of the call being laid out
// What orbhip_project_best_in_window_shared left in the calling thread's scratch - every slot's key frame, grid table and parameter block - for
// orbhip_project_best_in_window_held: valid until the thread's next ordinary layout (arena_layout with floor 0) or orbhip_thread_release
thread_local bool orbhip_tl_held_valid = false;
static thread_local struct HeldSlots { int device = -1; size_t floor = 0; std::vector<BestParams> B; std::vector<int> live_of_slot; } g_held;
#define g_scratch orbhip_tl_scratch
#define g_scratch_bytes orbhip_tl_scratch_bytes
#define g_scratch_dev orbhip_tl_scratch_dev
#define g_xfers orbhip_tl_xfers
typedef OrbXfer Xfer;
static thread_local uint8_t* g_hstage = nullptr; static thread_local size_t g_hstage_bytes = 0;      // pinned mirror of the arena's host-visible head
// The stream of the calling thread's stateless matcher calls: its own, non-blocking.  (They used to share the NULL stream: Tracking's, LocalMapping's
// and LoopClosing's calls then queue behind each other on the device and every hipStreamSynchronize waits for all three.)
static thread_local hipStream_t g_tstream = nullptr; static thread_local int g_tstream_dev = -1;
hipStream_t orbhip_thread_stream(int device)
{
    orbhip_touch_thread_caches();
    if (g_tstream && g_tstream_dev == device) return g_tstream;
    if (g_tstream) { (void)hipSetDevice(g_tstream_dev); (void)hipStreamSynchronize(g_tstream); (void)hipStreamDestroy(g_tstream); g_tstream = nullptr; (void)hipSetDevice(device); }
    if (hipStreamCreateWithFlags(&g_tstream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); g_tstream = nullptr; g_tstream_dev = -1; return nullptr; }   // the NULL stream still works
    g_tstream_dev = device;
    return g_tstream;
}
static hipError_t hstage_reserve(size_t bytes)
{
    orbhip_touch_thread_caches();
    if (g_hstage_bytes >= bytes) return hipSuccess;
    if (g_hstage) (void)hipHostFree(g_hstage);
    g_hstage = nullptr; g_hstage_bytes = 0;
    const hipError_t e = hipHostMalloc((void**)&g_hstage, bytes + bytes / 4, hipHostMallocDefault);
    if (e == hipSuccess) g_hstage_bytes = bytes + bytes / 4;
    return e;
}
// The per-thread caches above (and the brute-force / BoW matcher workspaces) belong to the thread that made the calls: a worker thread that
// exits gives them back (thread_local holder below), any thread may do so explicitly with orbhip_thread_release().  The main thread's are
// left to process exit: its thread-local destructors run while the interpreter / runtime that loaded this library is already unwinding.
static thread_local void* g_nn_ws = nullptr; static thread_local size_t g_nn_ws_bytes = 0; static thread_local int g_nn_ws_dev = -1; static thread_local hipStream_t g_nn_ws_stream = nullptr;
extern "C" void orbhip_thread_release(void)
{
    int cur = -1; (void)hipGetDevice(&cur);
    if (g_scratch) { (void)hipSetDevice(g_scratch_dev); (void)hipDeviceSynchronize(); (void)hipFree(g_scratch); }
    g_scratch = nullptr; g_scratch_bytes = 0; g_scratch_dev = -1; orbhip_tl_held_valid = false;
    if (g_tstream) { (void)hipSetDevice(g_tstream_dev); (void)hipStreamSynchronize(g_tstream); (void)hipStreamDestroy(g_tstream); }
    g_tstream = nullptr; g_tstream_dev = -1;
    if (g_hstage) (void)hipHostFree(g_hstage);
    g_hstage = nullptr; g_hstage_bytes = 0;
    if (g_nn_ws) { (void)hipSetDevice(g_nn_ws_dev); (void)hipDeviceSynchronize(); (void)hipFree(g_nn_ws); }
    g_nn_ws = nullptr; g_nn_ws_bytes = 0; g_nn_ws_dev = -1; g_nn_ws_stream = nullptr;
    orbhip_bow_thread_release();
    if (cur >= 0) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}
#include <unistd.h>
#include <sys/syscall.h>
namespace { struct ThreadCacheHolder { ~ThreadCacheHolder() { if ((long)syscall(SYS_gettid) != (long)getpid()) orbhip_thread_release(); } }; }
static thread_local ThreadCacheHolder g_cache_holder;
void orbhip_touch_thread_caches() { (void)&g_cache_holder; }

// One pageable->pinned gather on the host and ONE host-to-device copy for all inputs of a call (a hipMemcpy per array costs more
// than the kernels of a single-frame matcher call); likewise one device-to-host copy for all outputs.
hipError_t arena_upload(hipStream_t s)
{
    size_t hi = 0;
    for (const Xfer& x : g_xfers) hi = std::max(hi, x.off + std::max(x.bytes_in, x.bytes_out));
    hipError_t e = hstage_reserve(hi); if (e != hipSuccess) return e;
    size_t in_lo = (size_t)-1, in_hi = 0;
    for (const Xfer& x : g_xfers) if (x.src) { memcpy(g_hstage + x.off, x.src, x.bytes_in); in_lo = std::min(in_lo, x.off); in_hi = std::max(in_hi, x.off + x.bytes_in); }
    if (!in_hi) return hipSuccess;
    in_lo &= ~(size_t)255;                                    // (a layout above a floor leaves what lies below it alone)
    return orbhip_copy_async(static_cast<uint8_t*>(g_scratch) + in_lo, g_hstage + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, s);
}
hipError_t arena_download(hipStream_t s)
{
    size_t lo = (size_t)-1, hi = 0;
    for (const Xfer& x : g_xfers) if (x.dst) { lo = std::min(lo, x.off); hi = std::max(hi, x.off + x.bytes_out); }
    if (hi == 0) return hipStreamSynchronize(s);
    hipError_t e = orbhip_copy_async(g_hstage + (lo & ~(size_t)255), static_cast<uint8_t*>(g_scratch) + (lo & ~(size_t)255), hi - (lo & ~(size_t)255), hipMemcpyDeviceToHost, s); if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s); if (e != hipSuccess) return e;
    for (const Xfer& x : g_xfers) if (x.dst) memcpy(x.dst, g_hstage + x.off, x.bytes_out);
    return hipSuccess;
}
hipError_t orbhip_arena_reserve(int device, size_t bytes)
{
    if (g_scratch_dev != device || g_scratch_bytes < bytes) {
        if (g_scratch) { (void)hipSetDevice(g_scratch_dev); (void)hipFree(g_scratch); (void)hipSetDevice(device); }
        g_scratch = nullptr; g_scratch_bytes = 0;
        const hipError_t e = orbhip_dmalloc(&g_scratch, bytes + bytes / 4);
        if (e != hipSuccess) return e;
        g_scratch_bytes = bytes + bytes / 4; g_scratch_dev = device;
    }
    return hipSuccess;
}
template <typename T> static hipError_t upload(T** p, const std::vector<T>& v)
{
    hipError_t e = dalloc(p, v.size()); if (e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

static void build_resize_table(int sw, int dw, std::vector<int2>& tab)   // cv::resize coefficient tables (OpenCV 3.2 imgwarp.cpp)
{
    const double inv_scale = (double)dw / sw, scale = 1. / inv_scale;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int a0 = satShort(cvRoundF((1.f - fx) * 2048.f)), a1 = satShort(cvRoundF(fx * 2048.f));
        int2 e; e.x = sx; e.y = (a0 & 0xffff) | (a1 << 16);
        tab.push_back(e);
    }
}
static void build_yresize_table(int sh, int dh, std::vector<int2>& tab)  // rows are clipped in the kernel, weights are not reset
{
    const double inv_scale = (double)dh / sh, scale = 1. / inv_scale;
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        const int b0 = satShort(cvRoundF((1.f - fy) * 2048.f)), b1 = satShort(cvRoundF(fy * 2048.f));
        int2 e; e.x = sy; e.y = (b0 & 0xffff) | (b1 << 16);
        tab.push_back(e);
    }
}

// k_pyramid_cascade's rectangles.  A workgroup owns a TW x TH tile of the last level; the rectangle of level l-1 it needs is the source footprint of its
// rectangle of level l (first column aligned down to 4), widened where necessary so that the rectangles of all tiles cover level l-1 completely (a
// level's last columns / rows need not be referenced by the next level).  Columns and rows are independent: one table per tile column, one per tile row.
static hipError_t build_cascade(orbhip_ctx* c)
{
    c->pc_ok = false;
    const int L = c->L;
    if (L < 2 || L > ORBHIP_MAX_LEVELS) return hipSuccess;
    for (int l = 1; l < L; l++) if (c->pyr_staged[l] != 2) return hipSuccess;       // the kernel computes four pixels at a time from PyrGroup entries (scale factors below ~1.6)
    int TW = 32, TH = 16;      // measured on MI355X at 1241 x 376 / 8 levels, ms per call: 32x16 0.139-0.142, 64x8 0.140-0.141, 16x8 0.139-0.142, 32x8 0.141-0.142, 16x16 0.141-0.145 (seven launches: 0.158-0.162; 64x16 and larger exceed the kernel's 256 table entries per workgroup and fall back to them)
    if (const char* e = getenv("ORBHIP_PC_TILE")) { int w = 0, h = 0; if (sscanf(e, "%dx%d", &w, &h) == 2 && w >= 4 && h >= 1 && w % 4 == 0) { TW = w; TH = h; } else if (*e == '0') return hipSuccess; }
    const LevelGeom& gl = c->geom[L - 1];
    const int ntx = (gl.w + TW - 1) / TW, nty = (gl.h + TH - 1) / TH;
    std::vector<short2> xr((size_t)L * ntx), yr((size_t)L * nty);
    for (int t = 0; t < ntx; t++) xr[(size_t)(L - 1) * ntx + t] = short2{(short)(t * TW), (short)(std::min((t + 1) * TW, gl.w) - 1)};
    for (int t = 0; t < nty; t++) yr[(size_t)(L - 1) * nty + t] = short2{(short)(t * TH), (short)(std::min((t + 1) * TH, gl.h) - 1)};
    for (int l = L - 1; l >= 1; l--) {
        const LevelGeom& g = c->geom[l];
        if (g.src_w > 32767 || g.src_h > 32767) return hipSuccess;
        const int2* xt = c->xtab.data() + g.xtab_off; const int2* yt = c->ytab.data() + g.ytab_off;
        short2* sx = &xr[(size_t)(l - 1) * ntx]; short2* sy = &yr[(size_t)(l - 1) * nty];
        // (a rectangle's columns are computed in groups of 4 from its first one: the pixels up to the end of the last group are computed - and stored - too)
        for (int t = 0; t < ntx; t++) { short2& r = xr[(size_t)l * ntx + t]; r.y = (short)std::min(r.x + 4 * (((r.y - r.x) >> 2) + 1) - 1, g.w - 1); }
        for (int t = 0; t < ntx; t++) { const short2 r = xr[(size_t)l * ntx + t]; sx[t] = short2{(short)(xt[r.x].x & ~3), (short)std::min(xt[r.y].x + 1, g.src_w - 1)}; }
        for (int t = 0; t < nty; t++) { const short2 r = yr[(size_t)l * nty + t]; sy[t] = short2{(short)std::min(std::max(yt[r.x].x, 0), g.src_h - 1), (short)std::min(std::max(yt[r.y].x + 1, 0), g.src_h - 1)}; }
        sx[0].x = 0; sx[ntx - 1].y = (short)(g.src_w - 1); sy[0].x = 0; sy[nty - 1].y = (short)(g.src_h - 1);
        for (int t = 0; t + 1 < ntx; t++) sx[t].y = std::max<short>(sx[t].y, (short)(sx[t + 1].x - 1));
        for (int t = 0; t + 1 < nty; t++) sy[t].y = std::max<short>(sy[t].y, (short)(sy[t + 1].x - 1));
    }
    // LDS: the widest x tallest rectangle of the even levels | of the odd levels (the last level is not kept) | the tables
    auto xbytes = [&](int l, int t) { const short2 r = xr[(size_t)l * ntx + t]; return 4 * (((r.y - r.x) >> 2) + 1); };
    auto yrows = [&](int l, int t) { const short2 r = yr[(size_t)l * nty + t]; return r.y - r.x + 1; };
    int buf[2] = {16, 16}, xcap = 0, ycap = 0;
    for (int l = 0; l + 1 < L; l++) {
        int mx = 0, my = 0;
        for (int t = 0; t < ntx; t++) mx = std::max(mx, xbytes(l, t));
        for (int t = 0; t < nty; t++) my = std::max(my, yrows(l, t));
        buf[l & 1] = std::max(buf[l & 1], (mx * my + 16 + 15) & ~15);        // + 16: a group reads the three dwords from its first tap's on, used or not
    }
    for (int t = 0; t < ntx; t++) { int n = 0; for (int l = 1; l < L; l++) n += xbytes(l, t) / 4; xcap = std::max(xcap, n); }        // PyrGroup entries
    for (int t = 0; t < nty; t++) { int n = 0; for (int l = 1; l < L; l++) n += yrows(l, t); ycap = std::max(ycap, n); }
    {   // level 0 too (it has no footprint to take, but its last group is staged whole)
        const LevelGeom& g0 = c->geom[0];
        for (int t = 0; t < ntx; t++) { short2& r = xr[t]; r.y = (short)std::min(r.x + 4 * (((r.y - r.x) >> 2) + 1) - 1, g0.w - 1); }
    }
    int r0max = 0; for (int tx = 0; tx < ntx; tx++) for (int ty = 0; ty < nty; ty++) r0max = std::max(r0max, xbytes(0, tx) / 4 * yrows(0, ty));
    if ((size_t)buf[0] + buf[1] + (size_t)xcap * sizeof(PyrGroup) + (size_t)ycap * sizeof(int2) > 60 * 1024 || xcap > 256 || ycap > 512 || r0max > 4096) return hipSuccess;      // (k_pyramid_cascade's PC_GIT / PC_YIT / PC_RIT)
    // self-check before the tables are used: every level covered without a gap, and every tap of every pixel a workgroup computes (whole 4-pixel groups)
    // inside the rectangle it holds of the level below.  A table that fails is not used (the level kernels run instead).
    for (int l = 0; l < L; l++) {
        const LevelGeom& g = c->geom[l];
        int next = 0;
        for (int t = 0; t < ntx; t++) { const short2 r = xr[(size_t)l * ntx + t]; if (r.x > next || r.y < r.x || (r.x & 3)) return hipSuccess; next = std::max(next, r.y + 1); }
        if (next < g.w) return hipSuccess;
        next = 0;
        for (int t = 0; t < nty; t++) { const short2 r = yr[(size_t)l * nty + t]; if (r.x > next || r.y < r.x) return hipSuccess; next = std::max(next, r.y + 1); }
        if (next < g.h) return hipSuccess;
        if (l == 0) continue;
        const int2* xt = c->xtab.data() + g.xtab_off; const int2* yt = c->ytab.data() + g.ytab_off;
        for (int t = 0; t < ntx; t++) {
            const short2 r = xr[(size_t)l * ntx + t], sr = xr[(size_t)(l - 1) * ntx + t];
            const int sx_last = std::min(sr.x + xbytes(l - 1, t) - 1, g.src_w - 1);            // last source column the workgroup holds
            for (int x = r.x; x <= std::min(r.x + xbytes(l, t) - 1, g.w - 1); x++) { const int a = xt[x].x, b = std::min(a + 1, g.src_w - 1); if (a < sr.x || b > sx_last) return hipSuccess; }
        }
        for (int t = 0; t < nty; t++) {
            const short2 r = yr[(size_t)l * nty + t], sr = yr[(size_t)(l - 1) * nty + t];
            for (int y = r.x; y <= r.y; y++) { const int a = std::min(std::max(yt[y].x, 0), g.src_h - 1), b = std::min(std::max(yt[y].x + 1, 0), g.src_h - 1); if (a < sr.x || b > sr.y) return hipSuccess; }
        }
    }
    hipError_t e = upload(&c->d_pc_xr, xr); if (e != hipSuccess) return e;
    e = upload(&c->d_pc_yr, yr); if (e != hipSuccess) return e;
    c->pc_ntx = ntx; c->pc_nty = nty; c->pc_buf0 = buf[0]; c->pc_buf1 = buf[1]; c->pc_xcap = xcap; c->pc_ycap = ycap; c->pc_ok = true;
    return hipSuccess;
}

extern "C" int orbhip_pyramid_cascade_tiles(const orbhip_ctx* c) { return c && c->pc_ok ? c->pc_ntx * c->pc_nty : 0; }
extern "C" const char* orbhip_version(void) { return "orbhip 0.2 (gfx950)"; }
thread_local double orbhip_tl_api_ms = 0; thread_local int orbhip_tl_api_depth = 0;
extern "C" double orbhip_thread_api_ms(int reset) { const double v = orbhip_tl_api_ms; if (reset) orbhip_tl_api_ms = 0; return v; }
extern "C" int orbhip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }
extern "C" const char* orbhip_last_error(void) { return g_err.c_str(); }

extern "C" void orbhip_destroy(orbhip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    orbhip_bow_forget_ctx(c);                                      // BoW workspaces the vocabularies keep for this context (they ran on c->stream)
    if (c->mstream) { (void)hipStreamSynchronize(c->mstream); (void)hipStreamDestroy(c->mstream); }
    if (c->bstream) { (void)hipStreamSynchronize(c->bstream); (void)hipStreamDestroy(c->bstream); }
    if (c->bstream_host) { (void)hipStreamSynchronize(c->bstream_host); (void)hipStreamDestroy(c->bstream_host); }
    if (c->d_pc_xr) (void)hipFree(c->d_pc_xr);
    if (c->d_pc_yr) (void)hipFree(c->d_pc_yr);
    if (c->ev_pyr) (void)hipEventDestroy(c->ev_pyr);
    if (c->ev_blur) (void)hipEventDestroy(c->ev_blur);
    if (c->ev_extract) (void)hipEventDestroy(c->ev_extract);
    for (auto e : c->ev_match) if (e) (void)hipEventDestroy(e);
    for (auto xs : c->xstreams) { (void)hipStreamSynchronize(xs); (void)hipStreamDestroy(xs); }
    for (auto e : c->xevents) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    prof_collect(c);
    for (auto e : c->pool) (void)hipEventDestroy(e);
    void* ptrs[] = {c->d_blur_band, c->d_ic_mask, c->d_xgrp, c->d_fc_dma, c->d_geom, c->d_cells, c->d_tiles, c->d_xtab, c->d_ytab, c->d_pattern, c->d_pyr, c->d_blur, c->d_cell_count, c->d_cell_cand,
                    c->d_qt_val, c->d_qt_code, c->d_qt_node, c->d_lvl_kp, c->d_lvl_n[0], c->d_lvl_n[1], c->d_lvl_n[2], c->d_out_block[0], c->d_out_block[1], c->d_out_block[2], c->d_grid_start, c->d_grid_items, c->d_grid_xy, c->d_cand, c->d_top, c->d_ncand,
                    c->d_prev, c->d_m12, c->d_nm};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    { void* ep[] = {c->d_fgrid_start, c->d_fgrid_items, c->d_fgrid_xy, c->d_rrow_start, c->d_rrow_items}; for (void* q : ep) if (q) (void)hipFree(q); if (c->ev_epilogue) (void)hipEventDestroy(c->ev_epilogue); if (c->h_st) (void)hipHostFree(c->h_st); if (c->d_ucols) (void)hipFree(c->d_ucols); if (c->d_match_ws) (void)hipFree(c->d_match_ws); if (c->h_ucols) (void)hipHostFree(c->h_ucols); if (c->ev_ucols) (void)hipEventDestroy(c->ev_ucols); if (c->ev_stereo) (void)hipEventDestroy(c->ev_stereo); }
    { void* st[] = {c->d_st_rowstart, c->d_st_rowitems, c->d_st_u /* | d_st_depth */, c->d_st_sad}; for (void* q : st) if (q) (void)hipFree(q); }
    for (auto& hs : c->sets) {
        for (int k = 0; k < ORBHIP_MAX_CHUNKS; k++) { if (hs.ev_h2d[k]) (void)hipEventDestroy(hs.ev_h2d[k]); if (hs.ev_k[k]) (void)hipEventDestroy(hs.ev_k[k]); if (hs.ev_d2h[k]) (void)hipEventDestroy(hs.ev_d2h[k]); }
        if (hs.d_packed) (void)hipFree(hs.d_packed);
        if (hs.owned) { if (hs.d_in) (void)hipFree(hs.d_in); if (hs.h_in) (void)hipHostFree(hs.h_in); if (hs.h_block) (void)hipHostFree(hs.h_block); }
    }
    if (c->hstream) { (void)hipStreamSynchronize(c->hstream); (void)hipStreamDestroy(c->hstream); }
    if (c->dstream) { (void)hipStreamSynchronize(c->dstream); (void)hipStreamDestroy(c->dstream); }
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->d_col) (void)hipFree(c->d_col);
    if (c->h_col) (void)hipHostFree(c->h_col);
    { void* g[] = {c->d_out_kpun[0], c->d_out_kpun[1], c->d_out_kpun[2], c->d_map_x, c->d_map_y, c->d_raw, c->d_depth}; for (void* q : g) if (q) (void)hipFree(q); }
    if (c->h_kpun) (void)hipHostFree(c->h_kpun);
    if (c->h_raw) (void)hipHostFree(c->h_raw);
    if (c->h_block) (void)hipHostFree(c->h_block);             // h_n | h_kp | h_desc
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// LDS row stride of a k_fast_cells patch: 4 * (groups + 2) bytes cover every window read; 48 bytes for the usual 37-px cells puts the
// eight rows a wave reads at once on each LDS bank exactly twice (the minimum for 64 lanes x 4 bytes)
static int fc_pstride(const orbhip_ctx* c)
{
    return (c->fc_maxpw + 8 + 3) & ~3;
}

extern "C" orbhip_status orbhip_create(orbhip_ctx** out, const orbhip_config* cfg)
{
    if (!out || !cfg) return fail(ORBHIP_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->nfeatures < 0 || cfg->nlevels < 1 || cfg->nlevels > ORBHIP_MAX_LEVELS || cfg->scale_factor <= 1.0f || cfg->max_batch < 1 ||
        cfg->width < 1 || cfg->height < 1)
        return fail(ORBHIP_ERR_INVALID, "bad configuration (nfeatures %d, nlevels %d, scale %f, %dx%d, batch %d)", cfg->nfeatures, cfg->nlevels,
                    cfg->scale_factor, cfg->width, cfg->height, cfg->max_batch);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: the ORB front-end has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(ORBHIP_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));

    orbhip_ctx* c = new orbhip_ctx; c->cfg = *cfg; c->L = cfg->nlevels; c->B = cfg->max_batch;
    c->bounds.max_x = (float)cfg->width; c->bounds.max_y = (float)cfg->height;       // undistorted camera until orbhip_set_camera says otherwise
    const int L = c->L;
    // ---- ORBextractor::ORBextractor (ORBextractor.cc:410-446): double scaleFactor member initialised from the float argument
    const double scaleFactor = (double)cfg->scale_factor;
    c->sf.assign(L, 1.0f); c->s2.assign(L, 1.0f); c->isf.assign(L, 1.0f); c->is2.assign(L, 1.0f); c->nfeat.assign(L, 0);
    for (int i = 1; i < L; i++) { c->sf[i] = (float)(c->sf[i - 1] * scaleFactor); c->s2[i] = c->sf[i] * c->sf[i]; }
    for (int i = 0; i < L; i++) { c->isf[i] = 1.0f / c->sf[i]; c->is2[i] = 1.0f / c->s2[i]; }
    {
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = cfg->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
        int sum = 0;
        for (int l = 0; l < L - 1; l++) { c->nfeat[l] = cvRoundF(nDesired); sum += c->nfeat[l]; nDesired *= factor; }
        c->nfeat[L - 1] = std::max(cfg->nfeatures - sum, 0);
    }
    // ---- Gaussian kernel of cv::GaussianBlur(7x7, sigma 2) in 8-bit fixed point (getGaussianKernel + convertTo(CV_32S, 256))
    {
        float cf[7]; double sum = 0; const double scale2X = -0.5 / (2.0 * 2.0);
        for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)std::exp(scale2X * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 4; i++) c->gk[i] = cvRoundF((float)(cf[3 + i] * sum) * 256.f);      // centre, +-1, +-2, +-3
        { const char* e = getenv("ORBHIP_BLUR"); c->blur_mfma = !(e && strcmp(e, "valu") == 0); }       // ORBHIP_BLUR=valu: the all-VALU blur kernel instead of k_blur_mfma
        if (c->gk[0] > 127 || c->gk[0] + 2 * (c->gk[1] + c->gk[2] + c->gk[3]) != 257) c->blur_mfma = false;     // the i8 form assumes these taps
    }
    // ---- per level geometry (ORBextractor.cc:1111-1112, 773-806, 543-545)
    c->geom.resize(L);
    long long plane_off = 0, cand_off = 0; int kp_off = 0;
    for (int l = 0; l < L; l++) {
        LevelGeom& g = c->geom[l]; memset(&g, 0, sizeof g);
        g.w = cvRoundF((float)cfg->width * c->isf[l]); g.h = cvRoundF((float)cfg->height * c->isf[l]);
        if (g.w < 32 + 30 || g.h < 32 + 30 || g.w > 4095 || g.h > 4095) { const int gw = g.w, gh = g.h; delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "level %d is %dx%d: supported level sizes are 62..4095 px per side (the reference divides by zero below 30 px of interior)", l, gw, gh); }
        g.pitch = (g.w + 63) & ~63; g.plane_off = (int)plane_off; plane_off += (long long)g.pitch * g.h;
        g.maxBorderX = g.w - ORBHIP_EDGE; g.maxBorderY = g.h - ORBHIP_EDGE;
        const float width = (float)(g.maxBorderX - ORBHIP_EDGE), height = (float)(g.maxBorderY - ORBHIP_EDGE), W = 30;
        g.nCols = (int)(width / W); g.nRows = (int)(height / W);
        g.wCell = (int)ceil(width / g.nCols); g.hCell = (int)ceil(height / g.nRows);
        if (g.wCell > 59 || g.hCell > 59) { const int cw = g.wCell, ch = g.hCell; delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "cell %dx%d too large", cw, ch); }
        g.cell_first = (int)c->cells.size(); g.ncells = g.nCols * g.nRows;
        g.cand_total_off = (int)cand_off;
        for (int i = 0; i < g.nRows; i++) {
            const float iniY = ORBHIP_EDGE + i * g.hCell; float maxY = iniY + g.hCell + 6;
            const bool skipY = iniY >= g.maxBorderY - 3;
            if (maxY > g.maxBorderY) maxY = (float)g.maxBorderY;
            for (int j = 0; j < g.nCols; j++) {
                const float iniX = ORBHIP_EDGE + j * g.wCell; float maxX = iniX + g.wCell + 6;
                const bool skipX = iniX >= g.maxBorderX - 6;
                if (maxX > g.maxBorderX) maxX = (float)g.maxBorderX;
                CellDesc cd; memset(&cd, 0, sizeof cd);
                cd.level = (short)l; cd.skipped = (skipX || skipY) ? 1 : 0;
                cd.iniX = (short)iniX; cd.iniY = (short)iniY; cd.maxX = (short)maxX; cd.maxY = (short)maxY;
                cd.shiftX = (short)(j * g.wCell); cd.shiftY = (short)(i * g.hCell);
                const int cw = std::max((int)maxX - (int)iniX - 6, 0), ch = std::max((int)maxY - (int)iniY - 6, 0);
                cd.cand_cap = cd.skipped ? 0 : ((cw + 1) / 2) * ((ch + 1) / 2);       // strict 3x3 maxima are never 8-adjacent
                if (!cd.skipped) { c->fc_maxpw = std::max(c->fc_maxpw, (int)maxX - (int)iniX); c->fc_maxph = std::max(c->fc_maxph, (int)maxY - (int)iniY); }
                { const int ng = std::max((cw + 3) / 4, 1); cd.inv_ng = (65536 + ng - 1) / ng; }
                cd.cand_idx = (int)cand_off; cand_off += cd.cand_cap;
                c->cells.push_back(cd);
            }
        }
        g.cand_total_cap = (int)(cand_off - g.cand_total_off);
        g.nfeat = c->nfeat[l];
        g.nIni = (int)roundf((float)(g.maxBorderX - ORBHIP_EDGE) / (g.maxBorderY - ORBHIP_EDGE));
        if (g.nIni < 1) { delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "portrait image: the reference quadtree has zero root nodes (ORBextractor.cc:543-545 divides by zero)"); }
        g.hX = (float)(g.maxBorderX - ORBHIP_EDGE) / g.nIni;
        g.kp_cap = std::max(g.nfeat + 3, 4 * g.nIni); g.kp_off = kp_off; kp_off += g.kp_cap;
        g.scale = c->sf[l]; g.kp_size = (float)(int)(31 * c->sf[l]); g.inv_scale = c->isf[l];   // scaledPatchSize (:837)
        c->qt_maxn = std::max(c->qt_maxn, g.kp_cap); c->qt_maxcells = std::max(c->qt_maxcells, g.ncells);
        if (g.ncells > 65535 || g.kp_cap > 16383) { delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "level %d: too many cells / features for the quadtree scan", l); }
        if (l > 0) {
            g.src_w = c->geom[l - 1].w; g.src_h = c->geom[l - 1].h;
            g.xtab_off = (int)c->xtab.size(); build_resize_table(g.src_w, g.w, c->xtab);
            g.ytab_off = (int)c->ytab.size(); build_yresize_table(g.src_h, g.h, c->ytab);
            // does every output tile's source footprint fit the LDS stage of k_pyramid_level?
            const int tw = orbhip_pyramid_tile_w(), th = orbhip_pyramid_tile_h(); bool fits = true;
            for (int x0 = 0; x0 < g.w; x0 += tw) { const int xl = std::min(x0 + tw - 1, g.w - 1); const int a = c->xtab[g.xtab_off + x0].x & ~3, b = std::min(c->xtab[g.xtab_off + xl].x + 1, g.src_w - 1); fits = fits && orbhip_pyramid_tile_fits(b - a + 1, 1); }
            for (int y0 = 0; y0 < g.h; y0 += th) { const int yl = std::min(y0 + th - 1, g.h - 1); const int a = std::min(std::max(c->ytab[g.ytab_off + y0].x, 0), g.src_h - 1), b = std::min(std::max(c->ytab[g.ytab_off + yl].x + 1, 0), g.src_h - 1); fits = fits && orbhip_pyramid_tile_fits(1, b - a + 1); }
            c->pyr_staged.resize(L, 0); c->pyr_staged[l] = fits ? 1 : 0;
            // 4-pixel groups (k_pyramid_level_g): usable when every group's taps lie within 8 source bytes of its first tap and the three
            // dwords a thread reads stay inside the staged row
            g.xgrp_off = (int)c->xgrp.size(); bool grouped = fits;
            for (int x4 = 0; x4 < g.w; x4 += 4) {
                PyrGroup G; memset(&G, 0, sizeof G);
                G.sx0 = c->xtab[g.xtab_off + x4].x;
                for (int k = 0; k < 4; k++) {
                    const int2 e = c->xtab[g.xtab_off + std::min(x4 + k, g.w - 1)];
                    const int o0 = e.x - G.sx0, o1 = std::min(e.x + 1, g.src_w - 1) - G.sx0;
                    if (o0 < 0 || o1 < 0 || o0 > 7 || o1 > 7) grouped = false;
                    G.sel[k] = (unsigned)(o0 & 7) | (0x0cu << 8) | ((unsigned)(o1 & 7) << 16) | (0x0cu << 24);
                    G.coef[k] = (unsigned)e.y;
                }
                const int sxa = c->xtab[g.xtab_off + (x4 / tw) * tw].x & ~3;
                if (((G.sx0 - sxa) >> 2) + 2 >= orbhip_pyramid_tile_dwords()) grouped = false;
                c->xgrp.push_back(G);
            }
            if (grouped) c->pyr_staged[l] = 2;
        }
        const int btw = c->blur_mfma ? orbhip_blur_mfma_tile_w() : 128, bth = c->blur_mfma ? orbhip_blur_mfma_tile_h() : 32;     // k_blur_mfma / k_blur workgroup tile
        for (int y0 = 0; y0 < g.h; y0 += bth) for (int x0 = 0; x0 < g.w; x0 += btw) { TileDesc t; t.level = (short)l; t.x0 = (short)x0; t.y0 = (short)y0; t.pad = 0; c->blur_tiles.push_back(t); }
    }
    c->plane_frame_bytes = (plane_off + 255) & ~255LL; c->cand_slots_per_frame = cand_off; c->qt_per_frame = cand_off;
    c->lvl_kp_per_frame = kp_off; c->out_cap = kp_off; c->lvl0_cap = c->geom[0].kp_cap;
    if (cand_off >= (1 << 24)) { delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "too many candidate slots"); }
    if (orbhip_quadtree_lds_bytes(c->qt_maxn, c->qt_maxcells) > 150 * 1024) { const int qn = c->qt_maxn; delete c; return fail(ORBHIP_ERR_UNSUPPORTED, "nfeatures too large for the LDS quadtree (%d nodes)", qn); }

    // ---- device
    // Stream priorities (ORBHIP_STREAM_PRIO, default 2): the blur's stream - tens of thousands of independent tiles - gets the LOWEST priority, so that
    // the quadtree's few long workgroups on the main stream, which it runs beside, are always dispatched first (1); additionally the main stream the
    // highest and the matcher's stream the lowest (2).  Without priorities the order in which the launches reach the hardware decided whether the
    // quadtree ran 0.24 or 0.49 ms beside the blur (same binary, box by box: VERDICT r04 weak #6); with them 0.18 ms, and the B = 512 step gains 0.8 %
    // (profiles/r05_exp_stream_priorities.jsonl).  0 = no priorities (the old behaviour).
    int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // (numerically: lo = least urgent, hi = most urgent)
    const int prio_mode = [] { const char* e = getenv("ORBHIP_STREAM_PRIO"); return e ? atoi(e) : 2; }();
    if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
    else {
        hipError_t e = prio_mode >= 2 ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(ORBHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); } c->own_stream = true;
    }
    for (int i = 1; i < std::min(cfg->num_streams, cfg->max_batch); i++) {
        hipStream_t xs = nullptr; hipEvent_t xe = nullptr;
        if (hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&xe, hipEventDisableTiming) != hipSuccess) { orbhip_destroy(c); return fail(ORBHIP_ERR_HIP, "extra stream creation failed"); }
        c->xstreams.push_back(xs); c->xevents.push_back(xe);
    }
    if (!c->xstreams.empty() && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) { orbhip_destroy(c); return fail(ORBHIP_ERR_HIP, "event creation failed"); }
    { const char* e = getenv("ORBHIP_SERIAL"); c->serial = e && e[0] == '1'; }
    {   // the matcher runs on its own stream so that it overlaps the next batch's extraction
        bool ok = (prio_mode >= 2 ? hipStreamCreateWithPriority(&c->mstream, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&c->mstream, hipStreamNonBlocking)) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_extract, hipEventDisableTiming) == hipSuccess &&
                  (prio_mode >= 1 ? hipStreamCreateWithPriority(&c->bstream, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&c->bstream, hipStreamNonBlocking)) == hipSuccess &&
                  hipStreamCreateWithFlags(&c->bstream_host, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_pyr, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_blur, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 3 && ok; k++) ok = hipEventCreateWithFlags(&c->ev_match[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) { orbhip_destroy(c); return fail(ORBHIP_ERR_HIP, "match stream creation failed"); }
    }
    const size_t B = (size_t)c->B;
    std::vector<float> pat(1024);                                      // k_describe reads [component x0 y0 x1 y1][round][lane]: test 64 r + lane
    for (int t = 0; t < 256; t++) for (int k = 0; k < 4; k++) pat[(k * 4 + (t >> 6)) * 64 + (t & 63)] = kPatternHost[4 * t + k];
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    TRY(upload(&c->d_geom, c->geom)); TRY(upload(&c->d_cells, c->cells)); TRY(upload(&c->d_tiles, c->blur_tiles));
    TRY(upload(&c->d_xtab, c->xtab)); TRY(upload(&c->d_ytab, c->ytab)); TRY(upload(&c->d_xgrp, c->xgrp)); TRY(upload(&c->d_pattern, pat));
    TRY(build_cascade(c));
    if (c->blur_mfma) {   // k_blur_mfma: the two band (Toeplitz) matrices of the 7-tap filter laid out as B operands of v_mfma_i32_32x32x32_i8
        const int tap[7] = {c->gk[3], c->gk[2], c->gk[1], c->gk[0], c->gk[1], c->gk[2], c->gk[3]};
        std::vector<int4> band(3 * 64); signed char* bb = reinterpret_cast<signed char*>(band.data());
        for (int l = 0; l < 64; l++) for (int b = 0; b < 16; b++) {
            const int j = l & 31, hh = l >> 5;
            const int t1 = 16 * hh + b - j - 1, t2 = 32 + 16 * hh + b - j - 1;                 // source column x0 - 4 + k feeds output column x0 + j with tap k - j - 1
            const int r = (b & 3) + 8 * (b >> 2) + 4 * hh, tv = r - j;                         // source row y0 - 3 + r (the register / lane-half order of the first product's result)
            bb[(0 * 64 + l) * 16 + b] = (signed char)((t1 >= 0 && t1 <= 6) ? tap[t1] : 0);
            bb[(1 * 64 + l) * 16 + b] = (signed char)((t2 >= 0 && t2 <= 6) ? tap[t2] : 0);
            bb[(2 * 64 + l) * 16 + b] = (signed char)((tv >= 0 && tv <= 6 && j < 26) ? tap[tv] : 0);
        }
        TRY(upload(&c->d_blur_band, band));
    }
    {   // circular patch of IC_Angle as byte masks over 32 rows x 8 dwords (byte b of dword d = column 4d + b - 15); umax as ORBextractor.cc:452-469 computes it
        int umax[16]; const int vmax = (int)floor(15 * sqrt(2.0) / 2 + 1), vmin = (int)ceil(15 * sqrt(2.0) / 2);
        for (int v = 0; v <= vmax; v++) umax[v] = cvRoundF((float)sqrt(225.0 - (double)v * v));
        for (int v = 15, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
        std::vector<unsigned> mask(8 * 64, 0u);                       // k_describe: pass q, lane l = (row 4q + l / 16, dword l % 16 of the row from column -15 on)
        for (int q = 0; q < 8; q++) for (int l = 0; l < 64; l++) for (int b = 0; b < 4; b++) {
            const int r = 4 * q + (l >> 4), d = l & 15, v = r - 15, uu = 4 * d + b - 15;
            if (r <= 30 && d <= 7 && abs(uu) <= 15 && abs(uu) <= umax[abs(v)]) mask[q * 64 + l] |= 0xffu << (8 * b);
        }
        TRY(upload(&c->d_ic_mask, mask));
    }
    {   // k_fast_cells stages a cell's sub-image by LDS-DMA: pass k, lane l fills patch dword 64k + l = (row, column) in the PS-strided LDS layout
        const int psd = fc_pstride(c) / 4, nd = psd * std::max(c->fc_maxph, 1);
        c->fc_np = (nd + 63) / 64;
        const int np8 = (c->fc_np + 7) & ~7;
        std::vector<int4> tab((size_t)(np8 / 4) * 2 * 64);
        for (int k = 0; k < np8; k++) for (int l = 0; l < 64; l++) {
            const int pos = 64 * k + l, row = pos / psd, col4 = 4 * (pos % psd);
            int* r = &tab[(size_t)((k >> 2) * 2) * 64 + l].x; int* d = &tab[(size_t)((k >> 2) * 2 + 1) * 64 + l].x;
            r[k & 3] = row; d[k & 3] = col4;
        }
        TRY(upload(&c->d_fc_dma, tab));
    }
    TRY(dalloc(&c->d_pyr, B * c->plane_frame_bytes + 256)); TRY(dalloc(&c->d_blur, B * c->plane_frame_bytes + 256));
    TRY(dalloc(&c->d_cell_count, B * c->cells.size())); TRY(dalloc(&c->d_cell_cand, B * c->cand_slots_per_frame));
    TRY(dalloc(&c->d_qt_val, B * c->qt_per_frame)); TRY(dalloc(&c->d_qt_code, B * c->qt_per_frame)); TRY(dalloc(&c->d_qt_node, B * c->qt_per_frame));
    TRY(dalloc(&c->d_lvl_kp, B * c->lvl_kp_per_frame));
    for (int k = 0; k < 3; k++) {
        TRY(dalloc(&c->d_lvl_n[k], B * L));
        c->out_off_kp = (B * sizeof(int) + 255) & ~(size_t)255; c->out_off_desc = c->out_off_kp + ((B * c->out_cap * sizeof(orbhip_keypoint) + 255) & ~(size_t)255);
        c->out_block_bytes = c->out_off_desc + B * c->out_cap * 32;
        TRY(dalloc(&c->d_out_block[k], c->out_block_bytes));
        if (e == hipSuccess) { c->d_out_n[k] = reinterpret_cast<int*>(c->d_out_block[k]); c->d_out_kp[k] = reinterpret_cast<orbhip_keypoint*>(c->d_out_block[k] + c->out_off_kp); c->d_out_desc[k] = c->d_out_block[k] + c->out_off_desc; }
        if (e == hipSuccess) e = hipMemset(c->d_lvl_n[k], 0, B * L * sizeof(int));
        if (e == hipSuccess) e = hipMemset(c->d_out_n[k], 0, B * sizeof(int));
    }
    TRY(dalloc(&c->d_grid_start, B * (ORBHIP_GRID_CELLS + 1))); TRY(dalloc(&c->d_grid_items, B * c->out_cap)); TRY(dalloc(&c->d_grid_xy, B * c->out_cap));
    TRY(dalloc(&c->d_cand, B * c->lvl0_cap * (size_t)c->lvl0_cap)); TRY(dalloc(&c->d_ncand, B * c->lvl0_cap)); TRY(dalloc(&c->d_top, B * c->lvl0_cap * (size_t)5));
    TRY(dalloc(&c->d_prev, B * c->out_cap * 2)); TRY(dalloc(&c->d_m12, B * c->out_cap)); TRY(dalloc(&c->d_nm, B));
#undef TRY
    if (e != hipSuccess) { fail(ORBHIP_ERR_HIP, "device allocation failed: %s", hipGetErrorString(e)); orbhip_destroy(c); return ORBHIP_ERR_HIP; }
    *out = c;
    return ORBHIP_OK;
}

extern "C" int orbhip_keypoint_capacity(const orbhip_ctx* c) { return c ? c->out_cap : 0; }

extern "C" orbhip_status orbhip_get_scale_tables(const orbhip_ctx* c, float* sf, float* isf, float* s2, float* is2, int32_t* fpl)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    for (int i = 0; i < c->L; i++) { if (sf) sf[i] = c->sf[i]; if (isf) isf[i] = c->isf[i]; if (s2) s2[i] = c->s2[i]; if (is2) is2[i] = c->is2[i]; if (fpl) fpl[i] = c->nfeat[i]; }
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_level_size(const orbhip_ctx* c, int level, int* w, int* h)
{
    if (!c || level < 0 || level >= c->L) return fail(ORBHIP_ERR_INVALID, "bad level");
    if (w) *w = c->geom[level].w; if (h) *h = c->geom[level].h;
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_set_blur_rounding(orbhip_ctx* c, int mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(ORBHIP_ERR_INVALID, "blur_round_mode is 0 (generic C++) or 1 (x86 SSE2 build)");
    c->cfg.blur_round_mode = mode;          // read when the next call's kernel parameters are laid out
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_set_fp_contract(orbhip_ctx* c, int mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(ORBHIP_ERR_INVALID, "fp_contract is 0 (two roundings) or 1 (gcc's fused forms)");
    c->fp_contract = mode;
    return ORBHIP_OK;
}

// ---------------------------------------------------------------------------------------------- pipeline
static ExtractParams make_params(orbhip_ctx* c, const uint8_t* d_img0, long long frame_stride, int row_stride)
{
    ExtractParams P; memset(&P, 0, sizeof P);
    P.geom = c->d_geom; P.nlevels = c->L;
    P.img0 = d_img0; P.img0_frame_stride = frame_stride; P.img0_pitch = row_stride;
    P.pyr = c->d_pyr; P.blur = c->d_blur; P.plane_frame_bytes = c->plane_frame_bytes;
    P.cells = c->d_cells; P.ncells_total = (int)c->cells.size();
    P.cell_count = c->d_cell_count; P.cell_cand = c->d_cell_cand; P.cand_slots_per_frame = c->cand_slots_per_frame;
    P.qt_val = c->d_qt_val; P.qt_code = c->d_qt_code; P.qt_node = c->d_qt_node; P.qt_per_frame = c->qt_per_frame;
    P.lvl_kp = c->d_lvl_kp; P.lvl_kp_per_frame = c->lvl_kp_per_frame; P.lvl_n = c->d_lvl_n[c->cur];
    P.out_kp = c->d_out_kp[c->cur]; P.out_desc = c->d_out_desc[c->cur]; P.out_n = c->d_out_n[c->cur]; P.out_cap = c->out_cap;
    P.blur_tiles = c->d_tiles; P.nblur_tiles = (int)c->blur_tiles.size(); P.blur_band = c->blur_mfma ? c->d_blur_band : nullptr;
    P.xtab = c->d_xtab; P.ytab = c->d_ytab; P.xgrp = c->d_xgrp; P.patternf = c->d_pattern; P.ic_mask = c->d_ic_mask;
    P.pc_xr = c->d_pc_xr; P.pc_yr = c->d_pc_yr; P.pc_ntx = c->pc_ntx; P.pc_nty = c->pc_nty; P.pc_buf0 = c->pc_buf0; P.pc_buf1 = c->pc_buf1; P.pc_xcap = c->pc_xcap; P.pc_ycap = c->pc_ycap;
    P.iniTh = std::min(std::max(c->cfg.ini_th_fast, 0), 255); P.minTh = std::min(std::max(c->cfg.min_th_fast, 0), 255);      // cv::FAST clamps its threshold to [0, 255] (OpenCV 3.2 fast.cpp, FAST_t)
    P.blur_round_mode = c->cfg.blur_round_mode; P.fp_contract = c->fp_contract;
    P.qt_maxn = c->qt_maxn; P.qt_maxcells = c->qt_maxcells; P.qt_scr = orbhip_quadtree_scr(c->qt_maxn, c->qt_maxcells);
    P.fc_pstride = fc_pstride(c); P.fc_prows = c->fc_maxph; P.fc_np = c->fc_np; P.fc_pbytes = 256 * c->fc_np; P.fc_dma = c->d_fc_dma; P.fc_sstride = 4 + 4 * ((std::max(c->fc_maxpw - 6, 0) + 3) / 4); P.fc_srows = std::max(c->fc_maxph - 6, 0) + 2;
    P.fc_listcap = 2 * ((std::max(c->fc_maxpw - 6, 0) + 3) / 4) * std::max(c->fc_maxph - 6, 0);       // pixel pairs of the largest cell
    return P;
}

// levels 1 .. L-1, each from the one before: seven dependent launches for a batch (fewer were tried and lost there: the small levels in one launch of one
// workgroup per frame, 0.27 -> 0.54 ms at B = 256; docs/ROUND_LOG.md), ONE launch for up to eight frames, where the seven cost 35 us of pure latency
static void launch_pyramid(orbhip_ctx* c, const ExtractParams& P, int nf, hipStream_t s)
{
    static const bool cascade_always = getenv("ORBHIP_PC_ALWAYS") && atoi(getenv("ORBHIP_PC_ALWAYS")) != 0;      // experiment: the cascade at any batch
    if ((nf <= 8 || cascade_always) && c->pc_ok) { orbhip_launch_pyramid_cascade(P, nf, s); return; }      // a handful of frames: one launch for all levels (k_pyramid_cascade)
    for (int l = 1; l < c->L; l++) orbhip_launch_pyramid_level(P, l, c->geom[l].w, c->geom[l].h, (int)c->pyr_staged[l], nf, s);
}

// pyramid -> FAST -> quadtree -> describe for camera slots [f0, f0 + nf) on stream s.  With own_blur_stream (and more than a handful of frames)
// the blur - independent of FAST and the quadtree until the descriptor kernel - runs on the context's second stream beside the quadtree, whose
// workgroups are latency-bound (barriers, one workgroup per (frame, level)); every throughput kernel is alone on the GPU, so its HIP-event time
// in a timed region is its own (bench.py's roofline object relies on that).  Other placements of the second stream landed within 1.3 % of this
// one and are gone from the code (docs/ROUND_LOG.md, round 3 "schedules").
static orbhip_status pipeline_frames(orbhip_ctx* c, ExtractParams& P, int f0, int nf, hipStream_t s, bool own_blur_stream, bool host_path = false)
{
    if (nf <= 0) return ORBHIP_OK;
    P.frame0 = f0;
    { ProfScope ps(c, K_PYRAMID, s); launch_pyramid(c, P, nf, s); }
    { ProfScope ps(c, K_FAST, s); orbhip_launch_fast_cells(P, nf, s); }
    if (own_blur_stream && nf > 8 && !c->serial) {      // (a handful of frames: the two event hops of the second stream cost more than the blur's 11 us - a single-frame call lost 75 us in them)
        // (the host-buffer pipeline keeps a blur stream WITHOUT a priority: with the lowest one its pinned path fell from 110 k to 85 k frames/s - the blur of
        // chunk k starved beside the copies of chunks k - 1 and k + 1, profiles/r05_exp_host_path_stream_priorities.jsonl)
        hipStream_t bs = host_path ? c->bstream_host : c->bstream;
        HIPCHK(hipEventRecord(c->ev_pyr, s)); HIPCHK(hipStreamWaitEvent(bs, c->ev_pyr, 0));
        { ProfScope ps(c, K_QUADTREE, s); orbhip_launch_quadtree(P, nf, s); }
        { ProfScope ps(c, K_BLUR, bs); orbhip_launch_blur(P, c->gk, nf, bs); }
        HIPCHK(hipEventRecord(c->ev_blur, bs));
        HIPCHK(hipStreamWaitEvent(s, c->ev_blur, 0));
    } else if (nf <= 8 && P.blur_band && !c->serial && orbhip_quadtree_lds_bytes(c->qt_maxn, c->qt_maxcells) + 18 * 1024 <= 160 * 1024) {        // (the blur's tile is static LDS beside the quadtree's dynamic block)
        // a handful of frames: one launch for both (the quadtree's few long workgroups beside the blur's tiles); its time is booked on the quadtree
        ProfScope ps(c, K_QUADTREE, s); orbhip_launch_blur_quadtree(P, nf, s);
    } else {
        { ProfScope ps(c, K_BLUR, s); orbhip_launch_blur(P, c->gk, nf, s); }
        { ProfScope ps(c, K_QUADTREE, s); orbhip_launch_quadtree(P, nf, s); }
    }
    { ProfScope ps(c, K_DESCRIBE, s); orbhip_launch_describe(P, nf, s); }
    if (c->distorted) {   // Frame::UndistortKeyPoints (Frame.cc:404-434) behind the descriptor kernel: mvKeysUn stays in HBM beside mvKeys
        ProfScope ps(c, K_UNDISTORT, s);
        orbhip_launch_undistort_keys(c->cam, c->d_out_kp[c->cur] + (size_t)f0 * c->out_cap, c->d_out_n[c->cur] + f0, c->d_out_kpun[c->cur] + (size_t)f0 * c->out_cap, c->out_cap, nf, s);
    }
    return ORBHIP_OK;
}

// a new batch: rotate the output buffers and make the main stream wait for the matcher that still reads the buffer about to be overwritten
static orbhip_status begin_batch(orbhip_ctx* c, const uint8_t* d_img0, long long frame_stride, int row_stride)
{
    HIPCHK(hipSetDevice(c->cfg.device));
    c->cur = (c->cur + 1) % 3;
    const int cur = c->cur;
    c->last_img0 = d_img0; c->last_img0_fstride = frame_stride; c->last_img0_pitch = row_stride; c->d_last_uright = nullptr; c->last_n_valid = false; c->fgrid_valid = false; c->rrows_valid = false;
    // the buffer about to be overwritten was the "previous frame" of the matcher launched two calls ago
    for (int k = 0; k < 3; k++) if (c->match_pending[k] && (k == (cur + 1) % 3)) { HIPCHK(hipStreamWaitEvent(c->stream, c->ev_match[k], 0)); c->match_pending[k] = false; }
    // ... or may still be downloading (a submitted batch that has not been collected while un-ticketed calls rotate the buffers)
    for (auto& hs : c->sets) if (hs.busy && hs.out_buf == cur && hs.nchunks > 0) HIPCHK(hipStreamWaitEvent(c->stream, hs.ev_d2h[hs.nchunks - 1], 0));
    return ORBHIP_OK;
}

static orbhip_status run_pipeline(orbhip_ctx* c, int nimg, const uint8_t* d_img0, long long frame_stride, int row_stride,
                                  int match_prev, int window, float nnratio, int check_ori)
{
    orbhip_status st = begin_batch(c, d_img0, frame_stride, row_stride); if (st != ORBHIP_OK) return st;
    const int cur = c->cur, prev = (cur + 2) % 3;
    // camera slots are independent: optionally split the batch into groups, one HIP stream each
    const int ngroups = std::min((int)c->xstreams.size() + 1, nimg);
    ExtractParams P = make_params(c, d_img0, frame_stride, row_stride);
    if (ngroups > 1) { HIPCHK(hipEventRecord(c->ev_fork, c->stream)); }
    for (int gi = 0; gi < ngroups; gi++) {
        const int f0 = (int)((long long)nimg * gi / ngroups), f1 = (int)((long long)nimg * (gi + 1) / ngroups), nf = f1 - f0;
        hipStream_t s = gi == 0 ? c->stream : c->xstreams[gi - 1];
        if (gi > 0) HIPCHK(hipStreamWaitEvent(s, c->ev_fork, 0));
        if (nf <= 0) continue;
        st = pipeline_frames(c, P, f0, nf, s, ngroups == 1); if (st != ORBHIP_OK) return st;
        if (gi > 0) { HIPCHK(hipEventRecord(c->xevents[gi - 1], s)); HIPCHK(hipStreamWaitEvent(c->stream, c->xevents[gi - 1], 0)); }
    }
    if (match_prev) {
        MatchParams M; memset(&M, 0, sizeof M);
        // the matcher reads mvKeysUn of both frames (ORBmatcher.cc:418, 443 via GetFeaturesInArea)
        M.kp1 = c->distorted ? c->d_out_kpun[prev] : c->d_out_kp[prev]; M.desc1 = c->d_out_desc[prev]; M.n1 = c->d_out_n[prev]; M.n1_lvl0 = c->d_lvl_n[prev];
        M.kp2 = c->distorted ? c->d_out_kpun[cur] : c->d_out_kp[cur]; M.desc2 = c->d_out_desc[cur]; M.n2 = c->d_out_n[cur];
        M.lvl_stride = c->L; M.list1 = nullptr; M.prev_from_kp1 = 1;
        M.cap = c->out_cap; M.min_x = c->bounds.min_x; M.min_y = c->bounds.min_y; M.max_x = c->bounds.max_x; M.max_y = c->bounds.max_y;
        M.grid_start = c->d_grid_start; M.grid_items = c->d_grid_items; M.grid_xy = c->d_grid_xy; M.cand = c->d_cand; M.top = c->d_top; M.ncand = c->d_ncand; M.cand_stride = c->lvl0_cap; M.lvl0_cap = c->lvl0_cap;
        M.prev = c->d_prev; M.matches12 = c->d_m12; M.nmatches = c->d_nm; M.window = window; M.nnratio = nnratio; M.check_ori = check_ori; M.slot0 = 0;
        if (orbhip_match_select_big(c->out_cap, c->lvl0_cap)) {     // nfeatures beyond what LDS holds: the select kernel's tables in device memory, [slot][...], allocated at the first matched call
            if (!c->d_match_ws) HIPCHK(dalloc(&c->d_match_ws, (size_t)c->B * orbhip_match_select_ints(c->out_cap, c->lvl0_cap)));
            M.big_ws = c->d_match_ws;
        }
        // matcher of this batch on its own stream: latency-bound (one wave per slot), overlaps the next call's extraction
        hipStream_t ms = c->serial ? c->stream : c->mstream;
        HIPCHK(hipEventRecord(c->ev_extract, c->stream));
        HIPCHK(hipStreamWaitEvent(ms, c->ev_extract, 0));
        { ProfScope ps(c, K_MGRID, ms); orbhip_launch_match_grid(M, nimg, ms); }
        { ProfScope ps(c, K_MCAND, ms); orbhip_launch_match_candidates(M, nimg, ms); }
        { ProfScope ps(c, K_MSELECT, ms); orbhip_launch_match_select(M, nimg, ms); }
        HIPCHK(hipEventRecord(c->ev_match[cur], ms));
        c->match_pending[cur] = true;
    }
    c->last_matched = match_prev != 0;
    c->last_nimg = nimg;
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_extract_device(orbhip_ctx* c, int nimg, const uint8_t* d_imgs, size_t frame_stride, int row_stride,
                                               int match_prev, int window, float nnratio, int check_ori)
{
    if (!c || !d_imgs) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (nimg < 1 || nimg > c->B) return fail(ORBHIP_ERR_INVALID, "nimg %d outside 1..%d", nimg, c->B);
    if (row_stride < c->cfg.width) return fail(ORBHIP_ERR_INVALID, "row stride %d < width %d", row_stride, c->cfg.width);
    c->last_from_host = false;
    return run_pipeline(c, nimg, d_imgs, (long long)frame_stride, row_stride, match_prev, window, nnratio, check_ori);
}

void orbhip_internal_outputs(orbhip_ctx* c, const uint8_t** d_desc, const int** d_n, int* cap, int* last_nimg, int* device, hipStream_t* s)
{   // where the last extraction left its descriptors (orbhip_bow.hip reads them in place)
    *d_desc = c->d_out_desc[c->cur]; *d_n = c->d_out_n[c->cur]; *cap = c->out_cap; *last_nimg = c->last_nimg; *device = c->cfg.device; *s = c->stream;
}

extern "C" orbhip_status orbhip_sync(orbhip_ctx* c)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto xs : c->xstreams) HIPCHK(hipStreamSynchronize(xs));
    if (c->mstream) HIPCHK(hipStreamSynchronize(c->mstream));
    if (c->bstream) HIPCHK(hipStreamSynchronize(c->bstream));
    if (c->bstream_host) HIPCHK(hipStreamSynchronize(c->bstream_host));
    if (c->hstream) HIPCHK(hipStreamSynchronize(c->hstream));
    if (c->dstream) HIPCHK(hipStreamSynchronize(c->dstream));
    prof_collect(c);
    return ORBHIP_OK;
}

// The upload and the download stream of the pipelined host path.  Which hardware queue - and with it which DMA engine - a HIP stream lands on depends on how many
// streams the process has created before; when the two land on the same engine the PCIe link runs half duplex and the path delivers 88 k frames/s instead of
// 116 k.  Round 5 saw that as "the host path is slower inside a process that has held a resident context" and blamed the stream priorities; round 6's A/B showed the
// good and the bad state swap places with GPU_MAX_HW_QUEUES=2 or ORBHIP_STREAM_PRIO=1 - a property of the process's stream-creation history, not of the priorities
// (profiles/r06_exp_host_path_copy_streams.txt).  The copy pair is one factor of it that the library can choose (in-process 88 -> 101 k; the rest of the gap is open).
// So the pair is CHOSEN: four candidate streams, every pair timed on one concurrent 8 MB upload + 8 MB download (three tries, the best counts), the fastest pair
// kept, the others destroyed - a few milliseconds at the first host-path call of a context, which allocates its pinned ring anyway.  ORBHIP_COPY_STREAM_PROBE=0:
// the first two streams, as before.
static hipError_t create_copy_streams(hipStream_t* up, hipStream_t* down)
{
    const char* env = getenv("ORBHIP_COPY_STREAM_PROBE");
    const int K = (env && env[0] == '0') ? 2 : 4;
    hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int k = 0; k < K && e == hipSuccess; k++) e = hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking);
    int ba = 0, bb = 1;
    if (e == hipSuccess && K > 2) {
        const size_t bytes = (size_t)8 << 20;
        void *h0 = nullptr, *h1 = nullptr, *d0 = nullptr, *d1 = nullptr;
        if (hipHostMalloc(&h0, bytes, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&h1, bytes, hipHostMallocDefault) == hipSuccess &&
            orbhip_dmalloc(&d0, bytes) == hipSuccess && orbhip_dmalloc(&d1, bytes) == hipSuccess) {
            memset(h0, 1, bytes);
            for (int k = 0; k < K; k++) { (void)hipMemcpyAsync(d0, h0, 1 << 16, hipMemcpyHostToDevice, s[k]); (void)hipStreamSynchronize(s[k]); }      // every candidate has carried a copy once
            double best = 1e30;
            for (int a = 0; a < K; a++) for (int b = 0; b < K; b++) {
                if (a == b) continue;
                double t = 1e30;
                for (int rep = 0; rep < 3; rep++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    (void)hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s[a]);
                    (void)hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s[b]);
                    (void)hipStreamSynchronize(s[a]); (void)hipStreamSynchronize(s[b]);
                    t = std::min(t, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                if (t < best * 0.97) { best = t; ba = a; bb = b; }              // (a later pair must be clearly faster to displace an earlier one)
            }
        }
        (void)hipGetLastError();
        if (h0) (void)hipHostFree(h0); if (h1) (void)hipHostFree(h1); if (d0) (void)hipFree(d0); if (d1) (void)hipFree(d1);
    }
    if (e != hipSuccess) { for (int k = 0; k < K; k++) if (s[k]) (void)hipStreamDestroy(s[k]); return e; }
    *up = s[ba]; *down = s[bb];
    for (int k = 0; k < K; k++) if (k != ba && k != bb && s[k]) (void)hipStreamDestroy(s[k]);
    return hipSuccess;
}

static orbhip_status ensure_host_staging(orbhip_ctx* c, bool input)
{
    const size_t B = (size_t)c->B;
    if (!c->h_n) {
        HIPCHK(hipHostMalloc((void**)&c->h_block, c->out_block_bytes, hipHostMallocDefault));
        c->h_n = reinterpret_cast<int*>(c->h_block); c->h_kp = reinterpret_cast<orbhip_keypoint*>(c->h_block + c->out_off_kp); c->h_desc = c->h_block + c->out_off_desc;
    }
    if (input && !c->d_in) {
        c->in_pitch = c->geom[0].pitch;
        const size_t bytes = B * (size_t)c->in_pitch * c->cfg.height + 256;
        HIPCHK(orbhip_dmalloc((void**)&c->d_in, bytes));
        HIPCHK(hipHostMalloc((void**)&c->h_in, bytes, hipHostMallocDefault));
    }
    return ORBHIP_OK;
}

// The un-ticketed host entry points (orbhip_fetch*, the colour / rectify batch calls, the stereo calls) download into the context's OWN pinned
// mirrors - which are also staging set 0 of the ticketed path: while a submitted batch is still in flight they would overwrite (or read) what
// its collect is about to deliver.  They refuse instead, like orbhip_extract_batch does.
static orbhip_status mirrors_free(const orbhip_ctx* c, const char* who)
{
    if (c->oldest_ticket != c->next_ticket) return fail(ORBHIP_ERR_INVALID, "%s with %d submitted batch(es) still in flight: collect them first", who, c->next_ticket - c->oldest_ticket);
    return ORBHIP_OK;
}
static orbhip_status enqueue_fetch(orbhip_ctx* c, int nimg, bool want_kp, bool want_desc)
{   // bulk device-to-host copies into the pinned mirrors, ordered after the extraction on the context's stream
    { const orbhip_status st = mirrors_free(c, "a fetch into the context's mirrors"); if (st != ORBHIP_OK) return st; }
    // (orbhip_copy_async: a kernel for a few frames' worth, the DMA engines beyond 2 MB)
    HIPCHK(orbhip_copy_async(c->h_n, c->d_out_n[c->cur], nimg * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (want_kp) HIPCHK(orbhip_copy_async(c->h_kp, c->d_out_kp[c->cur], (size_t)nimg * c->out_cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, c->stream));
    if (want_desc) HIPCHK(orbhip_copy_async(c->h_desc, c->d_out_desc[c->cur], (size_t)nimg * c->out_cap * 32, hipMemcpyDeviceToHost, c->stream));
    return ORBHIP_OK;
}
static orbhip_status finish_fetch(orbhip_ctx* c, int nimg, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    bool overflow = false;
    c->last_n.assign(c->h_n, c->h_n + nimg); c->last_n_valid = nimg == c->last_nimg;
    for (int f = 0; f < nimg; f++) {
        const int n = c->h_n[f];
        n_out[f] = n;
        const int m = std::min(n, cap);
        if (n > cap) overflow = true;
        if (m > 0 && kps) memcpy(kps + (size_t)f * cap, c->h_kp + (size_t)f * c->out_cap, (size_t)m * sizeof(orbhip_keypoint));
        if (m > 0 && desc) memcpy(desc + (size_t)f * cap * 32, c->h_desc + (size_t)f * c->out_cap * 32, (size_t)m * 32);
    }
    return overflow ? fail(ORBHIP_ERR_CAPACITY, "keypoint buffer too small") : ORBHIP_OK;
}

extern "C" orbhip_status orbhip_fetch(orbhip_ctx* c, int nimg, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!c || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (nimg < 1 || nimg > c->last_nimg) return fail(ORBHIP_ERR_INVALID, "nimg %d but the last call processed %d frames", nimg, c->last_nimg);
    HIPCHK(hipSetDevice(c->cfg.device));
    orbhip_status st = ensure_host_staging(c, false); if (st != ORBHIP_OK) return st;
    st = enqueue_fetch(c, nimg, kps != nullptr, desc != nullptr); if (st != ORBHIP_OK) return st;
    st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    return finish_fetch(c, nimg, kps, desc, cap, n_out);
}

extern "C" orbhip_status orbhip_fetch_matches(orbhip_ctx* c, int nimg, int32_t* matches12, int cap1, int32_t* n1_out, int32_t* nmatches)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    if (!c->last_matched) return fail(ORBHIP_ERR_INVALID, "the last call did not run the matcher");
    if (nimg < 1 || nimg > c->last_nimg) return fail(ORBHIP_ERR_INVALID, "bad nimg");
    orbhip_status st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    std::vector<int> n1(nimg);
    HIPCHK(hipMemcpy(n1.data(), c->d_out_n[(c->cur + 2) % 3], nimg * sizeof(int), hipMemcpyDeviceToHost));
    if (nmatches) HIPCHK(hipMemcpy(nmatches, c->d_nm, nimg * sizeof(int), hipMemcpyDeviceToHost));
    for (int f = 0; f < nimg; f++) {
        if (n1_out) n1_out[f] = n1[f];
        const int m = std::min(n1[f], cap1);
        if (m > 0 && matches12) HIPCHK(hipMemcpy(matches12 + (size_t)f * cap1, c->d_m12 + (size_t)f * c->out_cap, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    }
    return ORBHIP_OK;
}

// ---------------------------------------------------------------------------------------------- pipelined host-buffer path
// A few helper threads for pageable <-> pinned gathers: one core copies ~10 GB/s, a batch of 64 KITTI frames is 30 MB in and
// 8 MB out, so a single-threaded memcpy alone would cap the host path near 20 k frames/s.  Process-wide, created on first use,
// never joined (the threads sleep on a condition variable; a dlclose'd library with live threads is the alternative).
// ---- NUMA placement (used by the pool's workers, orbhip_pool.hip, and by the copy helpers below)
int orbhip_device_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}
static thread_local int tl_numa_node = -1;      // node the calling thread was bound to by orbhip_bind_thread_to_node (-1: not bound)
// the CPUs this process may use, as they were when the library was loaded (taskset / numactl / a cpuset): what an unbound helper thread goes back to
// (it must not inherit the affinity of whichever bound worker happened to create it), and what a node binding never widens
static cpu_set_t g_process_cpus; static const bool g_have_process_cpus = sched_getaffinity(0, sizeof g_process_cpus, &g_process_cpus) == 0;
// "0-63,128-191" -> CPU set; no shared parser state (pool workers and copy helpers come through here at the same moment)
int orbhip_parse_cpulist(const char* text, cpu_set_t* set)
{
    CPU_ZERO(set);
    int ncpu = 0;
    for (const char* p = text; *p;) {
        if (*p < '0' || *p > '9') { p++; continue; }
        char* end = nullptr;
        long a = strtol(p, &end, 10), b = a;
        if (*end == '-' && end[1] >= '0' && end[1] <= '9') b = strtol(end + 1, &end, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (!CPU_ISSET((int)c, set)) { CPU_SET((int)c, set); ncpu++; }
        p = end;
    }
    return ncpu;
}
#ifdef ORBHIP_TEST_HOOKS      // the CPU emulation build only: the parser alone (tests/test_host_pipeline.py calls it from several threads at once)
extern "C" int orbhip_test_parse_cpulist(const char* text, int* cpus, int cap)
{
    cpu_set_t set; const int n = orbhip_parse_cpulist(text, &set);
    for (int c = 0, k = 0; c < CPU_SETSIZE && k < cap; c++) if (CPU_ISSET(c, &set)) cpus[k++] = c;
    return n;
}
#endif
bool orbhip_bind_thread_to_node(int node)
{   // /sys/devices/system/node/node<N>/cpulist
    if (node < 0) return false;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t set;
    int ncpu = orbhip_parse_cpulist(buf, &set);
    if (g_have_process_cpus) {                                   // never beyond what the user allowed this process (taskset, numactl --physcpubind)
        ncpu = 0;
        for (int c = 0; c < CPU_SETSIZE; c++) { if (CPU_ISSET(c, &set) && !CPU_ISSET(c, &g_process_cpus)) CPU_CLR(c, &set); if (CPU_ISSET(c, &set)) ncpu++; }
    }
    const bool ok = ncpu > 0 && sched_setaffinity(0, sizeof set, &set) == 0;      // an empty intersection: the thread stays where the user put it
    if (ok) tl_numa_node = node;
    return ok;
}

namespace {
struct CopyJob { std::atomic<int> next{0}, done{0}; int n = 0; std::function<void(int)> fn; };
class CopyPool {
    std::mutex m; std::condition_variable cv; std::deque<std::shared_ptr<CopyJob>> q; int nthreads = 0;
    static void drain(CopyJob& j) { for (int i; (i = j.next.fetch_add(1)) < j.n;) { j.fn(i); j.done.fetch_add(1); } }
    void worker() {
        for (;;) {
            std::shared_ptr<CopyJob> j;
            { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); j = q.front(); if (j->next.load() >= j->n) { q.pop_front(); continue; } }
            drain(*j);
        }
    }
public:
    explicit CopyPool(int node) {
        const char* e = getenv("ORBHIP_COPY_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        nthreads = e ? atoi(e) : std::min(8, std::max(hw / 4, 1));
        for (int i = 0; i + 1 < nthreads; i++)                                                   // the calling thread is the n-th copier
            std::thread([this, node] {
                if (node >= 0) (void)orbhip_bind_thread_to_node(node);
                else if (g_have_process_cpus) (void)sched_setaffinity(0, sizeof g_process_cpus, &g_process_cpus);
                worker();
            }).detach();
    }
    // one set of helpers per NUMA node: a caller that is bound to a node (a pool worker) gets copiers on that node's CPUs - pageable frames are
    // gathered into that node's pinned ring without crossing the socket; unbound callers share a set that roams
    static CopyPool& get() {
        static std::mutex gm; static CopyPool* pools[18] = {nullptr};
        const int k = (tl_numa_node >= 0 && tl_numa_node < 17) ? tl_numa_node + 1 : 0;
        std::lock_guard<std::mutex> lk(gm);
        if (!pools[k]) pools[k] = new CopyPool(k - 1);
        return *pools[k];
    }
    // fn(0) .. fn(n-1), spread over the helpers and the caller; returns when all are done
    void run(int n, size_t bytes_each, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (nthreads <= 1 || n == 1 || (size_t)n * bytes_each < (size_t)(1 << 19)) { for (int i = 0; i < n; i++) fn(i); return; }      // (a stereo pair's two 0.47 MB images are worth a second thread: ~40 us each from cold memory, a helper wakes in ~10)
        auto j = std::make_shared<CopyJob>(); j->n = n; j->fn = fn;
        { std::lock_guard<std::mutex> lk(m); q.push_back(j); }
        cv.notify_all();
        drain(*j);
        while (j->done.load() < n) std::this_thread::yield();
    }
};
bool host_pointer_is_pinned(const void* p)
{
    hipPointerAttribute_t a; memset(&a, 0, sizeof a);
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }           // an unregistered pointer is an "error": clear it
    return a.type == hipMemoryTypeHost;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- device memory for runtime-less callers
#include <dlfcn.h>
extern "C" orbhip_status orbhip_runtime_info(char* buf, int cap)
{
    if (!buf || cap < 1) return fail(ORBHIP_ERR_INVALID, "null argument");
    int rt = 0, drv = 0, nd = 0;
    HIPCHK(hipRuntimeGetVersion(&rt)); HIPCHK(hipDriverGetVersion(&drv)); HIPCHK(hipGetDeviceCount(&nd));
    if (nd < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available");
    hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, 0));
    Dl_info di; memset(&di, 0, sizeof di);
    const char* where = dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname ? di.dli_fname : "?";
    snprintf(buf, (size_t)cap, "hip runtime %d driver %d from %s; %d device(s); device 0: %s %s, %d CUs", rt, drv, where, nd, pr.name, pr.gcnArchName, pr.multiProcessorCount);
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_device_alloc(int device, size_t bytes, void** out)
{
    if (!out) return fail(ORBHIP_ERR_INVALID, "null argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    HIPCHK(orbhip_dmalloc(out, std::max<size_t>(bytes, 1)));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_device_free(int device, void* p)
{
    if (!p) return ORBHIP_OK;
    HIPCHK(hipSetDevice(device)); HIPCHK(hipFree(p));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_device_upload(int device, void* dst, const void* src_host, size_t bytes)
{
    if (bytes == 0) return ORBHIP_OK;
    if (!dst || !src_host) return fail(ORBHIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(device)); HIPCHK(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_device_download(int device, void* dst_host, const void* src, size_t bytes)
{
    if (bytes == 0) return ORBHIP_OK;
    if (!dst_host || !src) return fail(ORBHIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(device)); HIPCHK(hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_device_synchronize(int device)
{
    HIPCHK(hipSetDevice(device)); HIPCHK(hipDeviceSynchronize());
    return ORBHIP_OK;
}

extern "C" void* orbhip_host_alloc(size_t bytes) { void* p = nullptr; return hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) == hipSuccess ? p : nullptr; }
extern "C" void orbhip_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int orbhip_ring_depth(void) { return ORBHIP_RING; }

static orbhip_status ensure_set(orbhip_ctx* c, int si)
{
    HostSet& hs = c->sets[si];
    orbhip_status st = ensure_host_staging(c, true); if (st != ORBHIP_OK) return st;
    const size_t B = (size_t)c->B;
    if (!hs.d_in) {
        if (si == 0) { hs.d_in = c->d_in; hs.h_in = c->h_in; hs.h_kp = c->h_kp; hs.h_desc = c->h_desc; hs.h_n = c->h_n; hs.h_block = c->h_block; hs.owned = false; }       // set 0 = the context's own mirrors
        else {
            const size_t bytes = B * (size_t)c->in_pitch * c->cfg.height + 256;
            hs.owned = true;
            HIPCHK(orbhip_dmalloc((void**)&hs.d_in, bytes)); HIPCHK(hipHostMalloc((void**)&hs.h_in, bytes, hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void**)&hs.h_block, c->out_block_bytes, hipHostMallocDefault));
            hs.h_n = reinterpret_cast<int*>(hs.h_block); hs.h_kp = reinterpret_cast<orbhip_keypoint*>(hs.h_block + c->out_off_kp); hs.h_desc = hs.h_block + c->out_off_desc;
        }
    }
    if (!c->hstream) { const hipError_t ce = create_copy_streams(&c->hstream, &c->dstream); if (ce != hipSuccess) return fail(ORBHIP_ERR_HIP, "copy stream creation: %s", hipGetErrorString(ce)); }
    for (int k = 0; k < ORBHIP_MAX_CHUNKS; k++)
        if (!hs.ev_d2h[k]) { HIPCHK(hipEventCreateWithFlags(&hs.ev_h2d[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&hs.ev_k[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&hs.ev_d2h[k], hipEventDisableTiming)); }
    return ORBHIP_OK;
}

// frames per chunk of a batch: small batches stay one chunk on the main stream (no cross-stream hops: single-frame latency is what
// the stereo drop-in sees); larger ones are cut so that about eight chunks pipeline through upload / kernels / download
static int host_chunk_frames(int nimg, bool dma_both_ways)
{
    const char* e = getenv("ORBHIP_HOST_CHUNK"); const int forced = e ? atoi(e) : 0;          // tuning / test knob: frames per chunk
    if (forced > 0) return std::max((nimg + ORBHIP_MAX_CHUNKS - 1) / ORBHIP_MAX_CHUNKS, forced);
    if (nimg < 32) return nimg;
    // measured on MI355X at batch 256 (tools/host_io_matrix.py, round 3): pinned frames in + pinned results named at submit (DMA both ways, no
    // host copy) 111.5 k frames/s with 64-frame chunks, 103 k with 128, 70 k with 32 (launch-bound); every path with a host copy in it
    // (pageable frames or results) 93 k with 64-frame chunks and 105 k with 128: fewer, larger chunks amortise the copy threads' hand-offs
    const int ch = dma_both_ways ? std::min(std::max(((nimg + 3) / 4 + 7) & ~7, 16), 64) : std::min(std::max(((nimg + 1) / 2 + 7) & ~7, 16), 128);
    return std::max(ch, (nimg + ORBHIP_MAX_CHUNKS - 1) / ORBHIP_MAX_CHUNKS);
}

static int stereo_row_cap(const orbhip_ctx* c) { return c->out_cap * ((int)ceilf(4.0f * c->sf[c->L - 1]) + 3); }      // rows [floor(y-r), ceil(y+r)], r = 2*scale
static StereoSide stereo_side(orbhip_ctx* c);
// what the follow-up calls of a single-image extraction will ask for, enqueued behind its result download (see orbhip_ctx::want_fgrid)
static orbhip_status frame_epilogues(orbhip_ctx* c, hipStream_t s)
{
    if (c->want_fgrid) {
        if (!c->d_fgrid_start) { HIPCHK(dalloc(&c->d_fgrid_start, (size_t)ORBHIP_GRID_CELLS + 1)); HIPCHK(dalloc(&c->d_fgrid_items, (size_t)c->out_cap)); HIPCHK(dalloc(&c->d_fgrid_xy, (size_t)c->out_cap)); }
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp2 = (c->distorted ? c->d_out_kpun : c->d_out_kp)[c->cur]; M.n2 = c->d_out_n[c->cur]; M.cap = c->out_cap;
        M.min_x = c->bounds.min_x; M.min_y = c->bounds.min_y; M.max_x = c->bounds.max_x; M.max_y = c->bounds.max_y;
        M.grid_start = c->d_fgrid_start; M.grid_items = c->d_fgrid_items; M.grid_xy = c->d_fgrid_xy; M.grid_all_levels = 1;
        orbhip_launch_match_grid(M, 1, s);
        c->fgrid_valid = true; c->fgrid_cur = c->cur;
    }
    if (c->want_rrows) {
        if (!c->d_rrow_start) { c->rrow_cap = stereo_row_cap(c); HIPCHK(dalloc(&c->d_rrow_start, (size_t)c->cfg.height + 1)); HIPCHK(dalloc(&c->d_rrow_items, (size_t)c->rrow_cap)); }
        StereoParams T; memset(&T, 0, sizeof T);
        T.geom = c->d_geom; T.R = stereo_side(c); T.cap = c->out_cap; T.im_h = c->cfg.height; T.row_start = c->d_rrow_start; T.row_items = c->d_rrow_items; T.row_cap = c->rrow_cap;
        orbhip_launch_stereo_rows(T, 1, s);
        c->rrows_valid = true; c->rrows_cur = c->cur;
        if (!c->ev_epilogue) HIPCHK(hipEventCreateWithFlags(&c->ev_epilogue, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->ev_epilogue, s));                  // the LEFT context's stream runs the stereo matcher: it waits for this
    }
    return ORBHIP_OK;
}
static orbhip_status submit_body(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* direct_kp, uint8_t* direct_desc, int direct_cap, int* ticket);
// A submit that fails half-way (a HIP error between the first upload and the last download of the batch) issued no ticket: whatever it
// enqueued is drained here, so that the staging set it used is quiet again and the ring state is exactly what it was before the call.
static orbhip_status submit_impl(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* direct_kp, uint8_t* direct_desc, int direct_cap, int* ticket)
{
    const int before = c->next_ticket;
    const orbhip_status st = submit_body(c, nimg, imgs, stride, direct_kp, direct_desc, direct_cap, ticket);
    if (st != ORBHIP_OK && c->next_ticket == before && st == ORBHIP_ERR_HIP) {
        const std::string msg = orbhip_last_error();
        (void)hipDeviceSynchronize(); (void)hipGetLastError();
        return fail(st, "%s", msg.c_str());
    }
    return st;
}
static orbhip_status submit_body(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* direct_kp, uint8_t* direct_desc, int direct_cap, int* ticket)
{
    if (nimg < 1 || nimg > c->B) return fail(ORBHIP_ERR_INVALID, "nimg %d outside 1..%d", nimg, c->B);
    if (stride < c->cfg.width) return fail(ORBHIP_ERR_INVALID, "stride %d < width %d", stride, c->cfg.width);
    for (int f = 0; f < nimg; f++) if (!imgs[f]) return fail(ORBHIP_ERR_INVALID, "image %d is null", f);
    HIPCHK(hipSetDevice(c->cfg.device));
    int si = 0; while (si < ORBHIP_RING && c->sets[si].busy) si++;         // lowest free set: synchronous use never leaves set 0
    if (si == ORBHIP_RING) return fail(ORBHIP_ERR_INVALID, "ring full: %d batches in flight, collect ticket %d first", ORBHIP_RING, c->oldest_ticket);
    HostSet& hs = c->sets[si];
    orbhip_status st = ensure_set(c, si); if (st != ORBHIP_OK) return st;
    const size_t fbytes = (size_t)c->in_pitch * c->cfg.height;
    const int W = c->cfg.width, H = c->cfg.height;
    if (c->plane0_dirty) { HIPCHK(hipStreamSynchronize(c->stream)); c->plane0_dirty = false; }   // kernels of an un-ticketed colour / rectify call may still read set 0's level-0 plane
    // pinned input: asked of the runtime once per run of images that lie back to back in memory (its first and its last image) - a batch cut
    // from one pinned array is two queries, not one per image (hipPointerGetAttributes costs a microsecond or two each)
    bool pinned_in = true;
    {
        const size_t ibytes_run = (size_t)stride * c->cfg.height;
        for (int f = 0; f < nimg && pinned_in;) {
            int g = f + 1; while (g < nimg && imgs[g] == imgs[g - 1] + ibytes_run) g++;
            pinned_in = host_pointer_is_pinned(imgs[f]) && (g - 1 == f || host_pointer_is_pinned(imgs[g - 1] + (size_t)stride * (c->cfg.height - 1) + c->cfg.width - 1));
            f = g;
        }
    }
    const bool pinned_out = direct_kp && direct_desc && direct_cap > 0 && host_pointer_is_pinned(direct_kp) && host_pointer_is_pinned(direct_desc);
    const int ch = host_chunk_frames(nimg, pinned_in && pinned_out), nch = (nimg + ch - 1) / ch;
    const bool piped = nch > 1;
    hipStream_t hst = piped ? c->hstream : c->stream, dst = piped ? c->dstream : c->stream;
    // A small batch that fills the context (the drop-in's single-frame call): the whole output block comes back in ONE copy into the pinned
    // mirror and is handed over by memcpy - three DMA submissions cost more than copying 120 KB (the second and third of them started 60 us apart)
    const bool whole_block = !piped && nimg == c->B && c->out_block_bytes <= ((size_t)1 << 20);
    const bool direct_out = !whole_block && pinned_out;
    hs.direct_kp = direct_out ? direct_kp : nullptr; hs.direct_desc = direct_out ? direct_desc : nullptr; hs.direct_cap = direct_out ? direct_cap : 0;

    st = begin_batch(c, hs.d_in, (long long)fbytes, c->in_pitch); if (st != ORBHIP_OK) return st;
    c->last_from_host = true; c->last_d_in = hs.d_in;
    ExtractParams P = make_params(c, hs.d_in, (long long)fbytes, c->in_pitch);
    hs.nimg = nimg; hs.nchunks = nch;
    for (int k = 0; k < nch; k++) {
        const int f0 = k * ch, f1 = std::min(nimg, f0 + ch), nf = f1 - f0;
        hs.chunk_f0[k] = f0; hs.chunk_f0[k + 1] = f1;
        // ---- upload
        if (pinned_in && stride == c->in_pitch) {
            for (int f = f0; f < f1;) {                                   // images that lie back to back travel in one copy (one copy per image is launch-bound: 25 of 56 GB/s)
                int g = f + 1; while (g < f1 && imgs[g] == imgs[g - 1] + fbytes) g++;
                HIPCHK(hipMemcpyAsync(hs.d_in + f * fbytes, imgs[f], (size_t)(g - f) * fbytes, hipMemcpyHostToDevice, hst));
                f = g;
            }
        } else if (pinned_in) {
            // dense rows: linear DMA of every image as it lies in the caller's memory (adjacent images in one copy), rows spread to the
            // pipeline's pitch by k_repitch on the main stream
            const size_t ibytes = (size_t)stride * H;
            if (hs.packed_bytes < (size_t)c->B * ibytes) {
                if (hs.d_packed) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(hs.d_packed)); hs.d_packed = nullptr; hs.packed_bytes = 0; }
                HIPCHK(orbhip_dmalloc((void**)&hs.d_packed, (size_t)c->B * ibytes + 256)); hs.packed_bytes = (size_t)c->B * ibytes;
            }
            for (int f = f0; f < f1;) {
                int g = f + 1; while (g < f1 && imgs[g] == imgs[g - 1] + ibytes) g++;
                const size_t bytes = (size_t)(g - f - 1) * ibytes + (size_t)stride * (H - 1) + W;       // never past the last image's last pixel
                HIPCHK(hipMemcpyAsync(hs.d_packed + f * ibytes, imgs[f], bytes, hipMemcpyHostToDevice, hst));
                f = g;
            }
        } else {
            uint8_t* h_in = hs.h_in; const int in_pitch = c->in_pitch;
            CopyPool::get().run(nf, fbytes, [=](int i) {
                const int f = f0 + i; uint8_t* dstp = h_in + f * fbytes;
                if (stride == in_pitch) memcpy(dstp, imgs[f], fbytes);
                else for (int y = 0; y < H; y++) memcpy(dstp + (size_t)y * in_pitch, imgs[f] + (size_t)y * stride, W);
            });
            if (!piped) HIPCHK(orbhip_copy_async(hs.d_in + f0 * fbytes, hs.h_in + f0 * fbytes, nf * fbytes, hipMemcpyHostToDevice, hst));      // a few frames: a copy kernel, no DMA hand-over
            else HIPCHK(hipMemcpyAsync(hs.d_in + f0 * fbytes, hs.h_in + f0 * fbytes, nf * fbytes, hipMemcpyHostToDevice, hst));
        }
        if (piped) { HIPCHK(hipEventRecord(hs.ev_h2d[k], hst)); HIPCHK(hipStreamWaitEvent(c->stream, hs.ev_h2d[k], 0)); }
        if (pinned_in && stride != c->in_pitch)
            orbhip_launch_repitch(hs.d_packed + (size_t)f0 * stride * H, (long long)stride * H, stride, hs.d_in + f0 * fbytes, (long long)fbytes, c->in_pitch, W, H, nf, c->stream);
        // ---- kernels
        st = pipeline_frames(c, P, f0, nf, c->stream, true, true); if (st != ORBHIP_OK) return st;
        if (piped) { HIPCHK(hipEventRecord(hs.ev_k[k], c->stream)); HIPCHK(hipStreamWaitEvent(dst, hs.ev_k[k], 0)); }
        // ---- download
        const int cur = c->cur; const size_t oc = (size_t)c->out_cap;
        if (whole_block) { HIPCHK(orbhip_copy_async(hs.h_block, c->d_out_block[cur], c->out_block_bytes, hipMemcpyDeviceToHost, dst)); }
        else {
        HIPCHK(hipMemcpyAsync(hs.h_n + f0, c->d_out_n[cur] + f0, nf * sizeof(int), hipMemcpyDeviceToHost, dst));
        if (direct_out) {
            const size_t m = (size_t)std::min(direct_cap, c->out_cap);
            if ((size_t)direct_cap == oc) {
                HIPCHK(hipMemcpyAsync(direct_kp + f0 * oc, c->d_out_kp[cur] + f0 * oc, nf * oc * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, dst));
                HIPCHK(hipMemcpyAsync(direct_desc + f0 * oc * 32, c->d_out_desc[cur] + f0 * oc * 32, nf * oc * 32, hipMemcpyDeviceToHost, dst));
            } else {
                HIPCHK(hipMemcpy2DAsync(direct_kp + (size_t)f0 * direct_cap, (size_t)direct_cap * sizeof(orbhip_keypoint), c->d_out_kp[cur] + f0 * oc, oc * sizeof(orbhip_keypoint), m * sizeof(orbhip_keypoint), nf, hipMemcpyDeviceToHost, dst));
                HIPCHK(hipMemcpy2DAsync(direct_desc + (size_t)f0 * direct_cap * 32, (size_t)direct_cap * 32, c->d_out_desc[cur] + f0 * oc * 32, oc * 32, m * 32, nf, hipMemcpyDeviceToHost, dst));
            }
        } else {
            if (!piped) {       // a few frames on the main stream: copy kernels (no DMA hand-over between the last kernel and the download)
                HIPCHK(orbhip_copy_async(hs.h_kp + f0 * oc, c->d_out_kp[cur] + f0 * oc, nf * oc * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, dst));
                HIPCHK(orbhip_copy_async(hs.h_desc + f0 * oc * 32, c->d_out_desc[cur] + f0 * oc * 32, nf * oc * 32, hipMemcpyDeviceToHost, dst));
            } else {
                HIPCHK(hipMemcpyAsync(hs.h_kp + f0 * oc, c->d_out_kp[cur] + f0 * oc, nf * oc * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, dst));
                HIPCHK(hipMemcpyAsync(hs.h_desc + f0 * oc * 32, c->d_out_desc[cur] + f0 * oc * 32, nf * oc * 32, hipMemcpyDeviceToHost, dst));
            }
        }
        }
        HIPCHK(hipEventRecord(hs.ev_d2h[k], dst));
    }
    if (nimg == 1 && !piped && (c->want_fgrid || c->want_rrows)) { st = frame_epilogues(c, c->stream); if (st != ORBHIP_OK) return st; }
    c->last_matched = false; c->last_nimg = nimg; c->pair_mode = false;
    HIPCHK(hipGetLastError());
    hs.busy = true; hs.out_buf = c->cur; hs.ticket = c->next_ticket++; c->ticket_set[hs.ticket % ORBHIP_RING] = si;
    if (ticket) *ticket = hs.ticket;
    return ORBHIP_OK;
}

// per-frame destinations (NULL = not wanted): what the pool uses to scatter camera c's results straight to its rows
orbhip_status orbhip_collect_scatter(orbhip_ctx* c, int ticket, orbhip_keypoint* const* kps, uint8_t* const* desc, int cap, int* const* n_out)
{
    if (!c || !kps || !desc || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (ticket != c->oldest_ticket || ticket >= c->next_ticket) return fail(ORBHIP_ERR_INVALID, "ticket %d is not the oldest batch in flight (%d; %d submitted)", ticket, c->oldest_ticket, c->next_ticket);
    HostSet& hs = c->sets[c->ticket_set[ticket % ORBHIP_RING]];
    if (hs.direct_kp && ((kps[0] && kps[0] != hs.direct_kp) || (desc[0] && desc[0] != hs.direct_desc) || cap != hs.direct_cap))      // the ticket stays collectable
        return fail(ORBHIP_ERR_INVALID, "batch %d was submitted with its result buffers (orbhip_submit_to): collect it with the same buffers and capacity", ticket);
    HIPCHK(hipSetDevice(c->cfg.device));
    bool overflow = false;
    const size_t oc = (size_t)c->out_cap;
    for (int k = 0; k < hs.nchunks; k++) {
        const hipError_t we = hipEventSynchronize(hs.ev_d2h[k]);
        if (we != hipSuccess) {      // the ticket is retired all the same: a ticket that stays "oldest" for ever would block the ring behind it
            hs.busy = false; c->oldest_ticket++;
            return fail(ORBHIP_ERR_HIP, "hipEventSynchronize (batch %d, chunk %d): %s", ticket, k, hipGetErrorString(we));
        }
        const int f0 = hs.chunk_f0[k], nf = hs.chunk_f0[k + 1] - f0;
        const orbhip_keypoint* h_kp = hs.h_kp; const uint8_t* h_desc = hs.h_desc; const int* h_n = hs.h_n; const bool direct = hs.direct_kp != nullptr;
        for (int i = 0; i < nf; i++) { const int n = h_n[f0 + i]; if (n_out[f0 + i]) *n_out[f0 + i] = n; if (n > cap) overflow = true; }
        if (!direct)
            CopyPool::get().run(nf, oc * 60, [=](int i) {
                const int f = f0 + i, m = std::min(h_n[f], cap);
                if (m > 0 && kps[f]) memcpy(kps[f], h_kp + f * oc, (size_t)m * sizeof(orbhip_keypoint));
                if (m > 0 && desc[f]) memcpy(desc[f], h_desc + f * oc * 32, (size_t)m * 32);
            });
    }
    hs.busy = false; c->oldest_ticket++;
    if (c->oldest_ticket == c->next_ticket && hs.out_buf == c->cur && hs.nimg == c->last_nimg) { c->last_n.assign(hs.h_n, hs.h_n + hs.nimg); c->last_n_valid = true; }   // the context's current state
    return overflow ? fail(ORBHIP_ERR_CAPACITY, "keypoint buffer too small") : ORBHIP_OK;
}

static orbhip_status collect_flat(orbhip_ctx* c, int ticket, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (ticket < 0 || ticket >= c->next_ticket) return fail(ORBHIP_ERR_INVALID, "unknown ticket %d", ticket);
    const int nimg = c->sets[c->ticket_set[ticket % ORBHIP_RING]].nimg;
    std::vector<orbhip_keypoint*> pk(nimg); std::vector<uint8_t*> pd(nimg); std::vector<int*> pn(nimg);
    for (int f = 0; f < nimg; f++) { pk[f] = kps ? kps + (size_t)f * cap : nullptr; pd[f] = desc ? desc + (size_t)f * cap * 32 : nullptr; pn[f] = n_out + f; }
    return orbhip_collect_scatter(c, ticket, pk.data(), pd.data(), cap, pn.data());
}

extern "C" orbhip_status orbhip_submit(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, int* ticket)
{
    if (!c || !imgs || !ticket) return fail(ORBHIP_ERR_INVALID, "null argument");
    return submit_impl(c, nimg, imgs, stride, nullptr, nullptr, 0, ticket);
}
extern "C" orbhip_status orbhip_submit_to(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* ticket)
{
    if (!c || !imgs || !ticket) return fail(ORBHIP_ERR_INVALID, "null argument");
    return submit_impl(c, nimg, imgs, stride, kps, desc, cap, ticket);
}
extern "C" orbhip_status orbhip_collect(orbhip_ctx* c, int ticket, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!c || !n_out || cap < 0) return fail(ORBHIP_ERR_INVALID, "null argument");
    return collect_flat(c, ticket, kps, desc, cap, n_out);
}

extern "C" orbhip_status orbhip_extract_batch(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    OrbApiTimer api_timer;
    if (!c || !imgs || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (c->oldest_ticket != c->next_ticket) return fail(ORBHIP_ERR_INVALID, "orbhip_extract_batch with %d submitted batches still in flight: collect them first", c->next_ticket - c->oldest_ticket);
    int ticket = -1;
    // (a hipGraph replay of this whole call was measured and is NOT used: 0.52 ms vs 0.31 ms per single-frame call with plain
    //  launches on ROCm 7.2 — see DESIGN.md §5)
    orbhip_status st = submit_impl(c, nimg, imgs, stride, kps, desc, cap, &ticket); if (st != ORBHIP_OK) return st;
    return collect_flat(c, ticket, kps, desc, cap, n_out);
}

// Colour input (Tracking.cc:172-198, 217-229, 248-260 convert with cv::cvtColor before building the Frame): the
// conversion runs on the device into the context's level-0 plane, so the gray image never exists on the host.
static orbhip_status check_color_args(orbhip_ctx* c, int nimg, int row_stride, int channels)
{
    if (nimg < 1 || nimg > c->B) return fail(ORBHIP_ERR_INVALID, "nimg %d outside 1..%d", nimg, c->B);
    if (channels != 3 && channels != 4) return fail(ORBHIP_ERR_INVALID, "channels %d (3 or 4 expected; 1-channel frames go through orbhip_extract*)", channels);
    if (row_stride < c->cfg.width * channels) return fail(ORBHIP_ERR_INVALID, "row stride %d < width*channels %d", row_stride, c->cfg.width * channels);
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_extract_device_color(orbhip_ctx* c, int nimg, const uint8_t* d_imgs, size_t frame_stride, int row_stride,
                                                     int channels, int rgb_order, int match_prev, int window, float nnratio, int check_ori)
{
    if (!c || !d_imgs) return fail(ORBHIP_ERR_INVALID, "null argument");
    orbhip_status st = check_color_args(c, nimg, row_stride, channels); if (st != ORBHIP_OK) return st;
    HIPCHK(hipSetDevice(c->cfg.device));
    st = ensure_host_staging(c, true); if (st != ORBHIP_OK) return st;
    const size_t fbytes = (size_t)c->in_pitch * c->cfg.height;
    orbhip_launch_to_gray(d_imgs, (long long)frame_stride, row_stride, c->d_in, (long long)fbytes, c->in_pitch, c->cfg.width, c->cfg.height,
                          channels, rgb_order != 0, nimg, c->stream);
    HIPCHK(hipGetLastError());
    c->last_from_host = true; c->last_d_in = c->d_in; c->plane0_dirty = true;   // level 0 lives in the context's own plane
    return run_pipeline(c, nimg, c->d_in, (long long)fbytes, c->in_pitch, match_prev, window, nnratio, check_ori);
}
extern "C" orbhip_status orbhip_extract_batch_color(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, int channels, int rgb_order,
                                                    orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    OrbApiTimer api_timer;
    if (!c || !imgs || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    orbhip_status st = check_color_args(c, nimg, stride, channels); if (st != ORBHIP_OK) return st;
    HIPCHK(hipSetDevice(c->cfg.device));
    st = ensure_host_staging(c, true); if (st != ORBHIP_OK) return st;
    const size_t cpitch = ((size_t)c->cfg.width * 4 + 63) & ~(size_t)63, cfbytes = cpitch * c->cfg.height;   // room for 4 channels
    if (!c->d_col) {
        c->col_bytes = (size_t)c->B * cfbytes + 256;
        HIPCHK(orbhip_dmalloc((void**)&c->d_col, c->col_bytes));
        HIPCHK(hipHostMalloc((void**)&c->h_col, c->col_bytes, hipHostMallocDefault));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t rowbytes = (size_t)c->cfg.width * channels;
    for (int f = 0; f < nimg; f++) {
        if (!imgs[f]) return fail(ORBHIP_ERR_INVALID, "image %d is null", f);
        uint8_t* dst = c->h_col + f * cfbytes;
        for (int y = 0; y < c->cfg.height; y++) memcpy(dst + (size_t)y * cpitch, imgs[f] + (size_t)y * stride, rowbytes);
    }
    HIPCHK(hipMemcpyAsync(c->d_col, c->h_col, nimg * cfbytes, hipMemcpyHostToDevice, c->stream));
    st = orbhip_extract_device_color(c, nimg, c->d_col, cfbytes, (int)cpitch, channels, rgb_order, 0, 0, 0.f, 0);
    if (st != ORBHIP_OK) return st;
    return orbhip_fetch(c, nimg, kps, desc, cap, n_out);
}

extern "C" orbhip_status orbhip_extract(orbhip_ctx* c, const uint8_t* img, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!c || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (!img) { *n_out = 0; return ORBHIP_OK; }                       // if(_image.empty()) return;  (ORBextractor.cc:1046-1047)
    const uint8_t* imgs[1] = {img};
    return orbhip_extract_batch(c, 1, imgs, stride, kps, desc, cap, n_out);
}

static orbhip_status copy_plane(orbhip_ctx* c, const uint8_t* d_base, int pitch, int w, int h, uint8_t* dst, int dst_stride)
{
    orbhip_status st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    HIPCHK(hipMemcpy2DAsync(dst, dst_stride, d_base, pitch, w, h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_pyramid_level(orbhip_ctx* c, int frame, int level, uint8_t* dst, int dst_stride)
{
    if (!c || !dst || level < 0 || level >= c->L || frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "bad argument");
    const LevelGeom& g = c->geom[level];
    if (level == 0 && !c->last_from_host) return fail(ORBHIP_ERR_INVALID, "level 0 of a device-resident call is the caller's own buffer");
    if (level == 0) return copy_plane(c, (c->last_d_in ? c->last_d_in : c->d_in) + (size_t)frame * c->in_pitch * c->cfg.height, c->in_pitch, g.w, g.h, dst, dst_stride);
    return copy_plane(c, c->d_pyr + (size_t)frame * c->plane_frame_bytes + g.plane_off, g.pitch, g.w, g.h, dst, dst_stride);
}
extern "C" orbhip_status orbhip_pyramid_fetch_all(orbhip_ctx* c, int frame, uint8_t* const* dst, const int* dst_stride)
{
    OrbApiTimer api_timer;
    if (!c || !dst || !dst_stride || frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (!c->last_from_host) return fail(ORBHIP_ERR_INVALID, "level 0 of a device-resident call is the caller's own buffer");
    HIPCHK(hipSetDevice(c->cfg.device));
    if (c->bstream) HIPCHK(hipStreamSynchronize(c->bstream));
    if (c->bstream_host) HIPCHK(hipStreamSynchronize(c->bstream_host));               // everything that writes the planes is ordered before the main stream's tail
    for (int l = 0; l < c->L; l++) {
        if (!dst[l]) continue;
        const LevelGeom& g = c->geom[l];
        const uint8_t* src = l == 0 ? (c->last_d_in ? c->last_d_in : c->d_in) + (size_t)frame * c->in_pitch * c->cfg.height : c->d_pyr + (size_t)frame * c->plane_frame_bytes + g.plane_off;
        HIPCHK(hipMemcpy2DAsync(dst[l], dst_stride[l], src, l == 0 ? c->in_pitch : g.pitch, g.w, g.h, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_debug_blurred_level(orbhip_ctx* c, int frame, int level, uint8_t* dst, int dst_stride)
{
    if (!c || !dst || level < 0 || level >= c->L || frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "bad argument");
    const LevelGeom& g = c->geom[level];
    return copy_plane(c, c->d_blur + (size_t)frame * c->plane_frame_bytes + g.plane_off, g.pitch, g.w, g.h, dst, dst_stride);
}
extern "C" orbhip_status orbhip_debug_candidates(orbhip_ctx* c, int frame, int level, int32_t* xys, int cap, int* n_out)
{
    if (!c || !n_out || level < 0 || level >= c->L || frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "bad argument");
    orbhip_status st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    const LevelGeom& g = c->geom[level];
    std::vector<int> counts(g.ncells);
    HIPCHK(hipMemcpy(counts.data(), c->d_cell_count + (size_t)frame * c->cells.size() + g.cell_first, g.ncells * sizeof(int), hipMemcpyDeviceToHost));
    long long n = 0; for (int v : counts) n += v;
    *n_out = (int)n;
    const int m = (int)std::min<long long>(n, cap);
    if (m > 0 && xys) {
        std::vector<unsigned> v(m);
        HIPCHK(hipMemcpy(v.data(), c->d_qt_val + (size_t)frame * c->qt_per_frame + g.cand_total_off, (size_t)m * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) { xys[3 * i] = v[i] & 0xfff; xys[3 * i + 1] = (v[i] >> 12) & 0xfff; xys[3 * i + 2] = v[i] >> 24; }
    }
    return ORBHIP_OK;
}

// ---------------------------------------------------------------------------------------------- projection-guided search
static bool projection_ok(const orbhip_projection* P)
{
    return P && P->kind >= ORBHIP_PROJ_LAST_FRAME && P->kind <= ORBHIP_PROJ_SIM3 && P->gemm_mode >= 0 && P->gemm_mode <= 2 && P->nlevels >= 1 && P->nlevels <= ORBHIP_MAX_PROJ_LEVELS;
}
static void gated_out(orbhip_proj_query* q, int np) { if (q) for (int i = 0; i < np; i++) { memset(&q[i], 0, sizeof q[i]); q[i].radius = -1.0f; } }
static void gated_out(orbhip_best_query* q, int np) { if (q) for (int i = 0; i < np; i++) { memset(&q[i], 0, sizeof q[i]); q[i].radius = -1.0f; } }

// queries given (P == nullptr) or derived on the device from map points under *P (orbhip_project_search_bounds): `queries` is then nullptr and nq = the point count
static orbhip_status search_by_projection_impl(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right,
                                               const uint8_t* blocked, int n, const orbhip_bounds* bounds,
                                               const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                               const orbhip_projection* P, const orbhip_map_point* points, orbhip_proj_query* queries_out,
                                               int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches)
{
    OrbApiTimer api_timer;
    if (n < 0 || nq < 0 || !nmatches || (n > 0 && (!kps || !desc || !feature_query)) || (nq > 0 && ((!queries && !points) || !query_desc)) || !bounds || !(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y) || (mode != 0 && mode != 1) ||
        (points && !projection_ok(P)))
        return fail(ORBHIP_ERR_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n; i++) feature_query[i] = -1;
    if (points) gated_out(queries_out, nq);
    if (n == 0 || nq == 0) return ORBHIP_OK;
    if (n >= (1 << 19)) return fail(ORBHIP_ERR_UNSUPPORTED, "too many features");
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    orbhip_keypoint* dk = nullptr; uint8_t *dd = nullptr, *dqd = nullptr; unsigned char *dbl_in = nullptr; float* dur = nullptr; int *dn = nullptr, *dgs = nullptr, *dgi = nullptr, *dnc = nullptr, *dfq = nullptr, *dev = nullptr, *dbig = nullptr;
    float2* dgxy = nullptr; orbhip_proj_query* dq = nullptr; unsigned* dcand = nullptr; unsigned* dtop = nullptr;
    orbhip_map_point* dpts = nullptr; orbhip_projection* dP = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const int hn[2] = {n, 0}; int hres[2] = {0, 0};
    TRY(arena_layout(device, [&](Arena& A) {
        A.io(&dk, n, kps, n); A.io(&dd, (size_t)n * 32, desc, (size_t)n * 32); A.io(&dqd, (size_t)nq * 32, query_desc, (size_t)nq * 32);
        if (points) { A.io(&dpts, nq, points, nq); A.io(&dP, 1, P, 1); A.io(&dq, nq, (const orbhip_proj_query*)nullptr, 0, queries_out, queries_out ? nq : 0); }
        else A.io(&dq, nq, queries, nq);
        if (u_right) A.io(&dur, n, u_right, n);
        if (blocked) A.io(&dbl_in, n, blocked, n);
        A.io(&dn, 8, hn, 2, hres, 2);                         // [0] = n in, [1] = the return value out
        A.io(&dfq, n, (const int*)nullptr, 0, feature_query, n);
        A.take(&dgs, ORBHIP_GRID_CELLS + 1); A.take(&dgi, n); A.take(&dgxy, n); A.take(&dnc, nq); A.take(&dev, nq);
        A.take(&dcand, (size_t)nq * n); A.take(&dtop, (size_t)nq * 5);
        if (orbhip_proj_select_big(n)) A.take(&dbig, (size_t)4 * n);      // the select kernel's per-feature tables when they do not fit LDS
    }));
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp2 = dk; M.n2 = dn; M.cap = n; M.min_x = bounds->min_x; M.min_y = bounds->min_y; M.max_x = bounds->max_x; M.max_y = bounds->max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.grid_all_levels = 1;
        orbhip_launch_match_grid(M, 1, ts);
        ProjParams J; memset(&J, 0, sizeof J);
        J.kp = dk; J.desc = dd; J.u_right = dur; J.n = n; J.min_x = bounds->min_x; J.min_y = bounds->min_y; J.max_x = bounds->max_x; J.max_y = bounds->max_y; J.grid_start = dgs; J.grid_items = dgi; J.grid_xy = dgxy;
        J.q = dq; J.qdesc = dqd; J.nq = nq; J.cand = dcand; J.ncand = dnc; J.cand_stride = n; J.top = dtop;
        J.pts = dpts; J.proj = dP; J.q_out = dq;
        J.blocked_in = dbl_in; J.blocked_out = nullptr; J.feature_query = dfq; J.nmatches = dn + 1; J.events = dev;
        J.mode = mode; J.nnratio = nnratio; J.th_high = th_high; J.check_ori = check_ori; J.big_ws = dbig;
        orbhip_launch_proj(J, ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);            // never leave a copy in flight on the per-thread mirrors
    if (e == hipSuccess) *nmatches = hres[1];
#undef TRY
    orbhip_status st = ORBHIP_OK;
    if (e != hipSuccess) st = fail(ORBHIP_ERR_HIP, "search_by_projection: %s", hipGetErrorString(e));
    return st;
}
extern "C" orbhip_status orbhip_search_by_projection_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right,
                                                     const uint8_t* blocked, int n, const orbhip_bounds* bounds,
                                                     const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                                     int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches)
{
    if (nq > 0 && !queries) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_by_projection_impl(device, kps, desc, u_right, blocked, n, bounds, queries, query_desc, nq, nullptr, nullptr, nullptr, mode, nnratio, th_high, check_ori, feature_query, nmatches);
}
extern "C" orbhip_status orbhip_project_search_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, const uint8_t* blocked, int n, const orbhip_bounds* bounds,
                                                      const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                      float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches, orbhip_proj_query* queries_out)
{
    if (np > 0 && (!points || !proj)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_by_projection_impl(device, kps, desc, u_right, blocked, n, bounds, nullptr, point_desc, np, np > 0 ? proj : nullptr, np > 0 ? points : nullptr, queries_out, 1, nnratio, th_high, check_ori, feature_query, nmatches);
}

// Several frames in one pass: every per-slot array lives at [slot][cap] of one arena (one copy each way), the order-dependent kernel runs
// one workgroup per slot.
extern "C" orbhip_status orbhip_search_by_projection_batch(int device, int nslots, orbhip_proj_slot* slots, const orbhip_bounds* bounds,
                                                           int mode, float nnratio, int th_high, int check_ori)
{
    OrbApiTimer api_timer;
    if (nslots < 0 || (nslots > 0 && !slots) || !bounds || !(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y) || (mode != 0 && mode != 1))
        return fail(ORBHIP_ERR_INVALID, "bad argument");
    int cap = 1, qcap = 1; bool any_ur = false, any_bl = false, work = false;
    for (int s = 0; s < nslots; s++) {
        orbhip_proj_slot& S = slots[s];
        if (S.n < 0 || S.nq < 0 || (S.n > 0 && (!S.kps || !S.desc || !S.feature_query)) || (S.nq > 0 && (!S.queries || !S.query_desc))) return fail(ORBHIP_ERR_INVALID, "bad argument in slot %d", s);
        S.nmatches = 0;
        for (int i = 0; i < S.n; i++) S.feature_query[i] = -1;
        cap = std::max(cap, S.n); qcap = std::max(qcap, S.nq);
        any_ur = any_ur || S.u_right; any_bl = any_bl || S.blocked; work = work || (S.n > 0 && S.nq > 0);
    }
    if (!work) return ORBHIP_OK;
    if (cap >= (1 << 19)) return fail(ORBHIP_ERR_UNSUPPORTED, "too many features");
    if ((size_t)nslots * qcap * cap * sizeof(unsigned) > ((size_t)2 << 30)) return fail(ORBHIP_ERR_UNSUPPORTED, "candidate lists of %d slots x %d queries x %d features exceed 2 GB: split the batch", nslots, qcap, cap);
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    orbhip_keypoint* dk = nullptr; uint8_t *dd = nullptr, *dqd = nullptr; unsigned char* dbl = nullptr; float* dur = nullptr; int *dn = nullptr, *dnm = nullptr, *dgs = nullptr, *dgi = nullptr, *dnc = nullptr, *dfq = nullptr, *dev = nullptr, *dbig = nullptr;
    float2* dgxy = nullptr; orbhip_proj_query* dq = nullptr; unsigned *dcand = nullptr, *dtop = nullptr; ProjParams* dJ = nullptr;
    std::vector<int> hn(nslots), hnm(nslots, 0); std::vector<ProjParams> hJ(nslots);
    std::vector<float> no_ur(any_ur ? cap : 0, -1.0f); std::vector<uint8_t> no_bl(any_bl ? cap : 0, 0);
    for (int s = 0; s < nslots; s++) hn[s] = slots[s].n;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const size_t C = (size_t)cap, Q = (size_t)qcap;
    {   // the parameter table is read from hJ when the arena is uploaded, i.e. after the device addresses below have been filled in
        TRY(arena_layout(device, [&](Arena& A) {
            A.io(&dJ, (size_t)nslots, (const ProjParams*)hJ.data(), (size_t)nslots);
            A.io(&dn, (size_t)nslots, (const int*)hn.data(), (size_t)nslots);
            A.io(&dnm, (size_t)nslots, (const int*)hnm.data(), (size_t)nslots, hnm.data(), (size_t)nslots);
            for (int s = 0; s < nslots; s++) {                                        // [slot][cap] / [slot][qcap] blocks, each slot's rows from its own host arrays
                const orbhip_proj_slot& S = slots[s];
                orbhip_keypoint* k = nullptr; uint8_t *d = nullptr, *qd = nullptr; orbhip_proj_query* q = nullptr; float* ur = nullptr; unsigned char* bl = nullptr; int* fq = nullptr;
                A.io(&k, C, S.kps, (size_t)S.n); A.io(&d, C * 32, S.desc, (size_t)S.n * 32); A.io(&q, Q, S.queries, (size_t)S.nq); A.io(&qd, Q * 32, S.query_desc, (size_t)S.nq * 32);
                if (any_ur) A.io(&ur, C, S.u_right ? S.u_right : no_ur.data(), (size_t)S.n);
                if (any_bl) A.io(&bl, C, (const unsigned char*)(S.blocked ? S.blocked : no_bl.data()), (size_t)S.n);
                A.io(&fq, C, (const int*)nullptr, 0, S.feature_query, (size_t)S.n);
                if (s == 0) { dk = k; dd = d; dq = q; dqd = qd; dur = ur; dbl = bl; dfq = fq; }
                ProjParams& J = hJ[s]; memset(&J, 0, sizeof J);
                J.kp = k; J.desc = d; J.u_right = S.u_right ? ur : nullptr; J.n = S.n; J.q = q; J.qdesc = qd; J.nq = S.nq; J.blocked_in = S.blocked ? bl : nullptr; J.feature_query = fq;
            }
            A.take(&dgs, (size_t)nslots * (ORBHIP_GRID_CELLS + 1)); A.take(&dgi, nslots * C); A.take(&dgxy, nslots * C); A.take(&dnc, nslots * Q); A.take(&dev, nslots * Q);
            A.take(&dcand, nslots * Q * C); A.take(&dtop, nslots * Q * 5);
            if (orbhip_proj_select_big(cap)) A.take(&dbig, nslots * 4 * C);      // the select kernel's per-feature tables when the largest slot's do not fit LDS
        }));
        for (int s = 0; s < nslots && e == hipSuccess; s++) {
            ProjParams& J = hJ[s];
            J.big_ws = dbig ? dbig + s * 4 * C : nullptr;
            J.min_x = bounds->min_x; J.min_y = bounds->min_y; J.max_x = bounds->max_x; J.max_y = bounds->max_y;
            J.grid_start = dgs + (size_t)s * (ORBHIP_GRID_CELLS + 1); J.grid_items = dgi + s * C; J.grid_xy = dgxy + s * C;
            J.cand = dcand + s * Q * C; J.ncand = dnc + s * Q; J.cand_stride = cap; J.top = dtop + s * Q * 5; J.nmatches = dnm + s; J.events = dev + s * Q;
            J.mode = mode; J.nnratio = nnratio; J.th_high = th_high; J.check_ori = check_ori;
        }
    }
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        // Frame::AssignFeaturesToGrid of every slot: the grid kernel indexes [slot][stride]
        for (int s = 0; s < nslots; s++) {
            MatchParams M; memset(&M, 0, sizeof M);
            M.kp2 = hJ[s].kp; M.n2 = dn + s; M.cap = cap; M.min_x = bounds->min_x; M.min_y = bounds->min_y; M.max_x = bounds->max_x; M.max_y = bounds->max_y;
            M.grid_start = dgs + (size_t)s * (ORBHIP_GRID_CELLS + 1); M.grid_items = dgi + s * C; M.grid_xy = dgxy + s * C; M.grid_all_levels = 1;
            orbhip_launch_match_grid(M, 1, ts);
        }
        const float gwInv = (float)ORBHIP_GRID_COLS / (bounds->max_x - bounds->min_x), ghInv = (float)ORBHIP_GRID_ROWS / (bounds->max_y - bounds->min_y);
        orbhip_launch_proj_batch(dJ, nslots, qcap, cap, gwInv, ghInv, ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);            // never leave a copy in flight on the per-thread mirrors
#undef TRY
    if (e != hipSuccess) return fail(ORBHIP_ERR_HIP, "search_by_projection_batch: %s", hipGetErrorString(e));
    for (int s = 0; s < nslots; s++) slots[s].nmatches = hnm[s];
    return ORBHIP_OK;
}

static orbhip_status search_best_in_window_impl(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, int n, const orbhip_bounds* bounds,
                                                const float* inv_level_sigma2, int nlevels, const orbhip_best_query* queries, const uint8_t* query_desc, int nq,
                                                const orbhip_projection* P, const orbhip_map_point* points, orbhip_best_query* queries_out,
                                                int chi2_gate, int32_t* best_idx, int32_t* best_dist)
{
    OrbApiTimer api_timer;
    if (n < 0 || nq < 0 || (nq > 0 && ((!queries && !points) || !query_desc || !best_idx || !best_dist)) || (n > 0 && (!kps || !desc)) || !bounds || !(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y) ||
        (chi2_gate && (!inv_level_sigma2 || nlevels < 1)) || (points && !projection_ok(P))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (points) gated_out(queries_out, nq);
    if (n == 0 || nq == 0) return ORBHIP_OK;
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    orbhip_keypoint* dk = nullptr; uint8_t *dd = nullptr, *dqd = nullptr; float *dur = nullptr, *dsg = nullptr; int *dn = nullptr, *dgs = nullptr, *dgi = nullptr, *dbi = nullptr, *dbd = nullptr;
    float2* dgxy = nullptr; orbhip_best_query* dq = nullptr; orbhip_map_point* dpts = nullptr; orbhip_projection* dP = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const int hn[2] = {n, 0};
    TRY(arena_layout(device, [&](Arena& A) {
        A.io(&dk, n, kps, n); A.io(&dd, (size_t)n * 32, desc, (size_t)n * 32); A.io(&dqd, (size_t)nq * 32, query_desc, (size_t)nq * 32); A.io(&dn, 8, hn, 2);
        if (points) { A.io(&dpts, nq, points, nq); A.io(&dP, 1, P, 1); if (queries_out) A.io(&dq, nq, (const orbhip_best_query*)nullptr, 0, queries_out, nq); }
        else A.io(&dq, nq, queries, nq);
        if (u_right) A.io(&dur, n, u_right, n);
        if (inv_level_sigma2 && nlevels > 0) A.io(&dsg, nlevels, inv_level_sigma2, nlevels);
        A.io(&dbi, nq, (const int*)nullptr, 0, best_idx, nq); A.io(&dbd, nq, (const int*)nullptr, 0, best_dist, nq);
        A.take(&dgs, ORBHIP_GRID_CELLS + 1); A.take(&dgi, n); A.take(&dgxy, n);
    }));
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp2 = dk; M.n2 = dn; M.cap = n; M.min_x = bounds->min_x; M.min_y = bounds->min_y; M.max_x = bounds->max_x; M.max_y = bounds->max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.grid_all_levels = 1;
        orbhip_launch_match_grid(M, 1, ts);
        BestParams B; memset(&B, 0, sizeof B);
        B.kp = dk; B.desc = dd; B.u_right = dur; B.inv_level_sigma2 = dsg; B.grid_start = dgs; B.grid_items = dgi; B.grid_xy = dgxy;
        B.q = dq; B.qdesc = dqd; B.nq = nq; B.chi2_gate = chi2_gate; B.best_idx = dbi; B.best_dist = dbd;
        B.pts = dpts; B.proj = dP; B.q_out = points ? dq : nullptr;
        B.min_x = bounds->min_x; B.gw_inv = (float)ORBHIP_GRID_COLS / (float)(bounds->max_x - bounds->min_x);      // as orbhip_launch_match_grid lays the grid out
        orbhip_launch_best_in_window(B, ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);            // never leave a copy in flight on the per-thread mirrors
#undef TRY
    orbhip_status st = ORBHIP_OK;
    if (e != hipSuccess) st = fail(ORBHIP_ERR_HIP, "search_best_in_window: %s", hipGetErrorString(e));
    return st;
}
extern "C" orbhip_status orbhip_search_best_in_window_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, int n, const orbhip_bounds* bounds,
                                                      const float* inv_level_sigma2, int nlevels, const orbhip_best_query* queries, const uint8_t* query_desc, int nq,
                                                      int chi2_gate, int32_t* best_idx, int32_t* best_dist)
{
    if (nq > 0 && !queries) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_best_in_window_impl(device, kps, desc, u_right, n, bounds, inv_level_sigma2, nlevels, queries, query_desc, nq, nullptr, nullptr, nullptr, chi2_gate, best_idx, best_dist);
}
extern "C" orbhip_status orbhip_project_best_in_window_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, int n, const orbhip_bounds* bounds,
                                                              const float* inv_level_sigma2, int nlevels, const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                              int chi2_gate, int32_t* best_idx, int32_t* best_dist, orbhip_best_query* queries_out)
{
    if (np > 0 && (!points || !proj)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_best_in_window_impl(device, kps, desc, u_right, n, bounds, inv_level_sigma2, nlevels, nullptr, point_desc, np, np > 0 ? proj : nullptr, np > 0 ? points : nullptr, queries_out, chi2_gate, best_idx, best_dist);
}

// Several key frames in one pass (Fuse over all targets): [slot][cap] key point / descriptor / grid blocks in one arena; the feature grids of all slots are
// built by ONE k_match_grid launch when the slots share their image bounds (key frames of one camera do), the searches by one launch over all queries.
// Queries given per slot, or derived on the device from the slot's map points under its projection (orbhip_project_best_in_window_batch).
struct BestSlotIn {
    const orbhip_keypoint* kps; const uint8_t* desc; const float* u_right; int n; orbhip_bounds bounds; const float* inv_level_sigma2; int nlevels;
    const orbhip_best_query* queries; const uint8_t* query_desc; int nq; const orbhip_projection* proj; const orbhip_map_point* points;
    int32_t* best_idx; int32_t* best_dist;
};
// shared: every slot's queries are slots[0]'s (points / query_desc / nq: uploaded once); skip: see orbhip_project_best_in_window_shared
static orbhip_status search_best_in_window_batch_impl(int device, int nslots, BestSlotIn* slots, int chi2_gate, bool shared = false, const uint64_t* skip = nullptr)
{
    OrbApiTimer api_timer;
    if (nslots < 0 || (nslots > 0 && !slots)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (shared) {
        orbhip_tl_held_valid = false;                                                   // whatever an earlier call left held is not THIS call's (also when nothing is live below)
        if (nslots > 64) return fail(ORBHIP_ERR_INVALID, "at most 64 slots share one set of points");
        for (int s = 1; s < nslots; s++)
            if (slots[s].nq != slots[0].nq || (slots[0].nq > 0 && (slots[s].points != slots[0].points || slots[s].query_desc != slots[0].query_desc || !slots[s].points)))        // (no points: nothing to name)
                return fail(ORBHIP_ERR_INVALID, "slot %d does not name slot 0's points", s);
    }
    std::vector<int> live;
    int cap = 1;
    for (int s = 0; s < nslots; s++) {
        BestSlotIn& S = slots[s];
        if (S.n < 0 || S.nq < 0 || (S.nq > 0 && ((!S.queries && !S.points) || !S.query_desc || !S.best_idx || !S.best_dist)) || (S.n > 0 && (!S.kps || !S.desc)) ||
            !(S.bounds.max_x > S.bounds.min_x) || !(S.bounds.max_y > S.bounds.min_y) || (chi2_gate && (!S.inv_level_sigma2 || S.nlevels < 1)) || (S.nq > 0 && S.points && !projection_ok(S.proj)))
            return fail(ORBHIP_ERR_INVALID, "bad argument in slot %d", s);
        for (int i = 0; i < S.nq; i++) { S.best_idx[i] = -1; S.best_dist[i] = 256; }
        if (S.n == 0 || S.nq == 0) continue;
        live.push_back(s); cap = std::max(cap, S.n);
    }
    if (live.empty()) {
        if (shared) {                                                                   // held: a slot without key points answers -1 / 256; one whose key frame never travelled (no points were offered) cannot answer (-2)
            g_held.device = device; g_held.floor = 0; g_held.B.clear(); g_held.live_of_slot.assign((size_t)nslots, -1);
            for (int s = 0; s < nslots; s++) if (slots[s].n > 0) g_held.live_of_slot[(size_t)s] = -2;
            orbhip_tl_held_valid = true;
        }
        return ORBHIP_OK;
    }
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    cap = (cap + 63) & ~63;                                  // 64 key points = 7 x 256 bytes: the arena's 256-byte blocks then lie exactly cap records apart ([slot][cap])
    const int NL = (int)live.size(); const size_t C = (size_t)cap;
    bool same_bounds = true;
    for (int k = 1; k < NL; k++) same_bounds = same_bounds && !memcmp(&slots[live[k]].bounds, &slots[live[0]].bounds, sizeof(orbhip_bounds));
    std::vector<BestParams> hB(NL); std::vector<int> pref(NL + 1, 0), hn(NL);
    for (int k = 0; k < NL; k++) { pref[k + 1] = pref[k] + (slots[live[k]].nq + 3) / 4; hn[k] = slots[live[k]].n; }
    BestParams* dB = nullptr; int *dpref = nullptr, *dn = nullptr, *dgs = nullptr, *dgi = nullptr; float2* dgxy = nullptr; orbhip_keypoint* dk0 = nullptr;
    uint8_t* dqd0 = nullptr; orbhip_map_point* dpts0 = nullptr; unsigned long long* dskip = nullptr;          // shared: the one copy of the points
    size_t held_floor = 0;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    TRY(arena_layout(device, [&](Arena& A) {
        A.io(&dB, (size_t)NL, (const BestParams*)hB.data(), (size_t)NL);
        A.io(&dpref, (size_t)NL + 1, (const int*)pref.data(), (size_t)NL + 1);
        A.io(&dn, (size_t)NL, (const int*)hn.data(), (size_t)NL);
        for (int k = 0; k < NL; k++) {                                                  // [slot][cap] key points first: k_match_grid indexes them by slot
            orbhip_keypoint* dk = nullptr; A.io(&dk, C, slots[live[k]].kps, (size_t)slots[live[k]].n);
            if (k == 0) dk0 = dk;
            hB[k].kp = dk;
        }
        for (int k = 0; k < NL; k++) {                                                  // every slot's inputs ...
            const BestSlotIn& S = slots[live[k]]; BestParams& B = hB[k];
            uint8_t *dd = nullptr, *dqd = nullptr; float *dur = nullptr, *dsg = nullptr; orbhip_best_query* dq = nullptr; orbhip_map_point* dpts = nullptr; orbhip_projection* dP = nullptr;
            A.io(&dd, (size_t)S.n * 32, S.desc, (size_t)S.n * 32);
            if (shared) {
                if (k == 0) {
                    A.io(&dqd0, (size_t)S.nq * 32, S.query_desc, (size_t)S.nq * 32); A.io(&dpts0, S.nq, S.points, S.nq);
                    if (skip) A.io(&dskip, S.nq, reinterpret_cast<const unsigned long long*>(skip), S.nq);
                }
                dqd = dqd0; dpts = dpts0; A.io(&dP, 1, S.proj, 1);
            } else {
                A.io(&dqd, (size_t)S.nq * 32, S.query_desc, (size_t)S.nq * 32);
                if (S.points) { A.io(&dpts, S.nq, S.points, S.nq); A.io(&dP, 1, S.proj, 1); }
                else A.io(&dq, S.nq, S.queries, S.nq);
            }
            if (S.u_right) A.io(&dur, S.n, S.u_right, S.n);
            if (S.inv_level_sigma2 && S.nlevels > 0) A.io(&dsg, S.nlevels, S.inv_level_sigma2, S.nlevels);
            B.desc = dd; B.u_right = dur; B.inv_level_sigma2 = dsg; B.q = dq; B.qdesc = dqd; B.nq = S.nq; B.chi2_gate = chi2_gate;
            B.pts = dpts; B.proj = dP; B.q_out = nullptr;
            B.skip = shared ? dskip : nullptr; B.skip_bit = live[k];
            B.min_x = S.bounds.min_x; B.gw_inv = (float)ORBHIP_GRID_COLS / (float)(S.bounds.max_x - S.bounds.min_x);
        }
        for (int k = 0; k < NL; k++) {                                                  // ... then every slot's answers, contiguous: the download is one small copy
            const BestSlotIn& S = slots[live[k]]; BestParams& B = hB[k];
            int *dbi = nullptr, *dbd = nullptr;
            A.io(&dbi, S.nq, (const int*)nullptr, 0, S.best_idx, S.nq); A.io(&dbd, S.nq, (const int*)nullptr, 0, S.best_dist, S.nq);
            B.best_idx = dbi; B.best_dist = dbd;
        }
        A.take(&dgs, (size_t)NL * (ORBHIP_GRID_CELLS + 1)); A.take(&dgi, NL * C); A.take(&dgxy, NL * C);
        if (shared) {                                                                   // room for the held entry's queries behind everything: it never reallocates
            held_floor = A.off; uint8_t* pad = nullptr; A.take(&pad, (size_t)slots[live[0]].nq * (sizeof(orbhip_map_point) + 32 + 8) + 4096);
        }
    }));
    for (int k = 0; k < NL && e == hipSuccess; k++) { hB[k].grid_start = dgs + (size_t)k * (ORBHIP_GRID_CELLS + 1); hB[k].grid_items = dgi + k * C; hB[k].grid_xy = dgxy + k * C; }
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        for (int k = 0; k < (same_bounds ? 1 : NL); k++) {
            const orbhip_bounds& b = slots[live[k]].bounds;
            MatchParams M; memset(&M, 0, sizeof M);
            M.kp2 = dk0; M.n2 = dn; M.cap = cap; M.min_x = b.min_x; M.min_y = b.min_y; M.max_x = b.max_x; M.max_y = b.max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.grid_all_levels = 1; M.slot0 = k;
            orbhip_launch_match_grid(M, same_bounds ? NL : 1, ts);
        }
        orbhip_launch_best_in_window_batch(dB, dpref, NL, pref[NL], ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);
#undef TRY
    if (e != hipSuccess) {
        for (int k = 0; k < NL; k++) { BestSlotIn& S = slots[live[k]]; for (int i = 0; i < S.nq; i++) { S.best_idx[i] = -1; S.best_dist[i] = 256; } }
        return fail(ORBHIP_ERR_HIP, "search_best_in_window_batch: %s", hipGetErrorString(e));
    }
    if (shared) {                                                                       // the slots stay where they are for orbhip_project_best_in_window_held
        g_held.device = device; g_held.floor = held_floor; g_held.B = hB; g_held.live_of_slot.assign((size_t)nslots, -1);
        for (int k = 0; k < NL; k++) g_held.live_of_slot[(size_t)live[k]] = k;
        orbhip_tl_held_valid = true;
    }
    return ORBHIP_OK;
}
// One slot of the calling thread's last orbhip_project_best_in_window_shared call searched again with other points: its key frame, descriptors and
// grid table are still in the thread's scratch - only the points travel (ORBmatcher.cc's FuseBatch: the points whose descriptor an earlier target's
// MapPoint::Replace changed, MapPoint.cc:177-215)
extern "C" orbhip_status orbhip_project_best_in_window_held(int device, int slot, const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                            int chi2_gate, int32_t* best_idx, int32_t* best_dist)
{
    OrbApiTimer api_timer;
    if (np < 0 || (np > 0 && (!points || !point_desc || !best_idx || !best_dist || !projection_ok(proj)))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (!orbhip_tl_held_valid || g_held.device != device || slot < 0 || slot >= (int)g_held.live_of_slot.size())
        return fail(ORBHIP_ERR_INVALID, "no held slot %d: the calling thread's last scratch-using call was not orbhip_project_best_in_window_shared on this device", slot);
    for (int i = 0; i < np; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    const int k = g_held.live_of_slot[(size_t)slot];
    if (k == -2) return fail(ORBHIP_ERR_INVALID, "held slot %d: its key frame did not travel (the shared call offered no points)", slot);
    if (k < 0 || np == 0) return ORBHIP_OK;                                             // (a slot without key points or a call without points: nothing to search)
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    uint8_t* dqd = nullptr; orbhip_map_point* dpts = nullptr; orbhip_projection* dP = nullptr; int *dbi = nullptr, *dbd = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    TRY(arena_layout(device, [&](Arena& A) {
        A.io(&dqd, (size_t)np * 32, point_desc, (size_t)np * 32); A.io(&dpts, np, points, np); A.io(&dP, 1, proj, 1);
        A.io(&dbi, np, (const int*)nullptr, 0, best_idx, np); A.io(&dbd, np, (const int*)nullptr, 0, best_dist, np);
    }, g_held.floor));
    if (e == hipErrorOutOfMemory) return fail(ORBHIP_ERR_INVALID, "the held scratch has no room for %d points", np);      // (the caller falls back to the full entry)
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        BestParams B = g_held.B[(size_t)k];
        B.q = nullptr; B.qdesc = dqd; B.nq = np; B.chi2_gate = chi2_gate; B.pts = dpts; B.proj = dP; B.q_out = nullptr; B.best_idx = dbi; B.best_dist = dbd; B.skip = nullptr; B.skip_bit = 0;
        orbhip_launch_best_in_window(B, ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);
#undef TRY
    if (e != hipSuccess) { for (int i = 0; i < np; i++) { best_idx[i] = -1; best_dist[i] = 256; } return fail(ORBHIP_ERR_HIP, "project_best_in_window_held: %s", hipGetErrorString(e)); }
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_search_best_in_window_batch(int device, int nslots, orbhip_best_slot* slots, int chi2_gate)
{
    if (nslots < 0 || (nslots > 0 && !slots)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    std::vector<BestSlotIn> in((size_t)nslots);
    for (int s = 0; s < nslots; s++) {
        const orbhip_best_slot& S = slots[s];
        if (S.nq > 0 && !S.queries) return fail(ORBHIP_ERR_INVALID, "bad argument in slot %d", s);
        in[s] = BestSlotIn{S.kps, S.desc, S.u_right, S.n, S.bounds, S.inv_level_sigma2, S.nlevels, S.queries, S.query_desc, S.nq, nullptr, nullptr, S.best_idx, S.best_dist};
    }
    return search_best_in_window_batch_impl(device, nslots, in.data(), chi2_gate);
}
extern "C" orbhip_status orbhip_project_best_in_window_batch(int device, int nslots, orbhip_project_best_slot* slots, int chi2_gate)
{
    if (nslots < 0 || (nslots > 0 && !slots)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    std::vector<BestSlotIn> in((size_t)nslots);
    for (int s = 0; s < nslots; s++) {
        const orbhip_project_best_slot& S = slots[s];
        if (S.np > 0 && (!S.points || !S.proj)) return fail(ORBHIP_ERR_INVALID, "bad argument in slot %d", s);
        in[s] = BestSlotIn{S.kps, S.desc, S.u_right, S.n, S.bounds, S.inv_level_sigma2, S.nlevels, nullptr, S.point_desc, S.np, S.np > 0 ? S.proj : nullptr, S.np > 0 ? S.points : nullptr, S.best_idx, S.best_dist};
    }
    return search_best_in_window_batch_impl(device, nslots, in.data(), chi2_gate);
}
extern "C" orbhip_status orbhip_project_best_in_window_shared(int device, int nslots, orbhip_project_best_slot* slots, const uint64_t* skip, int chi2_gate)
{
    if (nslots < 0 || (nslots > 0 && !slots)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    std::vector<BestSlotIn> in((size_t)nslots);
    for (int s = 0; s < nslots; s++) {
        const orbhip_project_best_slot& S = slots[s];
        if (S.np > 0 && (!S.points || !S.proj)) return fail(ORBHIP_ERR_INVALID, "bad argument in slot %d", s);
        in[s] = BestSlotIn{S.kps, S.desc, S.u_right, S.n, S.bounds, S.inv_level_sigma2, S.nlevels, nullptr, S.point_desc, S.np, S.np > 0 ? S.proj : nullptr, S.np > 0 ? S.points : nullptr, S.best_idx, S.best_dist};
    }
    return search_best_in_window_batch_impl(device, nslots, in.data(), chi2_gate, true, skip);
}

// The two searches above on a frame that is still on the device: key points (mvKeysUn with a distorted camera attached), descriptors
// and — if asked for — mvuRight of the last stereo / RGB-D step are read where the extraction left them; only the queries travel.
static orbhip_status frame_args(orbhip_ctx* c, int frame, int n, int use_u_right, const orbhip_keypoint** kp, const uint8_t** desc, const float** ur)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    if (frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "frame %d outside the %d frames of the last extraction", frame, c->last_nimg);
    if (n < 0 || n > c->out_cap) return fail(ORBHIP_ERR_INVALID, "n %d outside 0..%d", n, c->out_cap);
    if (use_u_right && !c->d_last_uright) return fail(ORBHIP_ERR_INVALID, "no mvuRight on the device: run orbhip_compute_stereo_matches / orbhip_compute_stereo_from_rgbd on this context first");
    *kp = (c->distorted ? c->d_out_kpun : c->d_out_kp)[c->cur] + (size_t)frame * c->out_cap;
    *desc = c->d_out_desc[c->cur] + (size_t)frame * c->out_cap * 32;
    *ur = use_u_right ? c->d_last_uright + (size_t)frame * c->out_cap : nullptr;
    return ORBHIP_OK;
}
static orbhip_status search_by_projection_frame_impl(orbhip_ctx* c, int frame, int n, int use_u_right, const uint8_t* blocked,
                                                     const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                                     const orbhip_projection* P, const orbhip_map_point* points, orbhip_proj_query* queries_out,
                                                     int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches)
{
    OrbApiTimer api_timer;
    const orbhip_keypoint* dk = nullptr; const uint8_t* dd = nullptr; const float* dur = nullptr;
    orbhip_status st = frame_args(c, frame, n, use_u_right, &dk, &dd, &dur); if (st != ORBHIP_OK) return st;
    if (nq < 0 || !nmatches || (n > 0 && !feature_query) || (nq > 0 && ((!queries && !points) || !query_desc)) || (mode != 0 && mode != 1) || (points && !projection_ok(P))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n; i++) feature_query[i] = -1;
    if (points) gated_out(queries_out, nq);
    if (n == 0 || nq == 0) return ORBHIP_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    uint8_t* dqd = nullptr; unsigned char* dbl_in = nullptr; int *dn = nullptr, *dgs = nullptr, *dgi = nullptr, *dnc = nullptr, *dfq = nullptr, *dev = nullptr, *dbig = nullptr;
    float2* dgxy = nullptr; orbhip_proj_query* dq = nullptr; unsigned* dcand = nullptr; unsigned* dtop = nullptr;
    orbhip_map_point* dpts = nullptr; orbhip_projection* dP = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const int hn[2] = {n, 0}; int hres[2] = {0, 0};
    TRY(arena_layout(c->cfg.device, [&](Arena& A) {
        A.io(&dqd, (size_t)nq * 32, query_desc, (size_t)nq * 32);
        if (points) { A.io(&dpts, nq, points, nq); A.io(&dP, 1, P, 1); A.io(&dq, nq, (const orbhip_proj_query*)nullptr, 0, queries_out, queries_out ? nq : 0); }
        else A.io(&dq, nq, queries, nq);
        if (blocked) A.io(&dbl_in, n, blocked, n);
        A.io(&dn, 8, hn, 2, hres, 2);
        A.io(&dfq, n, (const int*)nullptr, 0, feature_query, n);
        A.take(&dgs, ORBHIP_GRID_CELLS + 1); A.take(&dgi, n); A.take(&dgxy, n); A.take(&dnc, nq); A.take(&dev, nq);
        A.take(&dcand, (size_t)nq * n); A.take(&dtop, (size_t)nq * 5);
        if (orbhip_proj_select_big(n)) A.take(&dbig, (size_t)4 * n);
    }));
    TRY(arena_upload(c->stream));
    if (e == hipSuccess) {
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp2 = dk; M.n2 = dn; M.cap = n; M.min_x = c->bounds.min_x; M.min_y = c->bounds.min_y; M.max_x = c->bounds.max_x; M.max_y = c->bounds.max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.grid_all_levels = 1;
        // the grid of this frame was built behind its extraction (frame epilogue, same stream): take it; from now on it always will be
        if (frame == 0 && c->fgrid_valid && c->fgrid_cur == c->cur && c->last_n_valid && c->last_n[0] == n) { dgs = c->d_fgrid_start; dgi = c->d_fgrid_items; dgxy = c->d_fgrid_xy; }
        else orbhip_launch_match_grid(M, 1, c->stream);
        if (c->last_nimg == 1 || c->pair_mode) c->want_fgrid = true;
        ProjParams J; memset(&J, 0, sizeof J);
        J.kp = dk; J.desc = dd; J.u_right = dur; J.n = n; J.min_x = M.min_x; J.min_y = M.min_y; J.max_x = M.max_x; J.max_y = M.max_y; J.grid_start = dgs; J.grid_items = dgi; J.grid_xy = dgxy;
        J.q = dq; J.qdesc = dqd; J.nq = nq; J.cand = dcand; J.ncand = dnc; J.cand_stride = n; J.top = dtop;
        J.pts = dpts; J.proj = dP; J.q_out = dq;
        J.blocked_in = dbl_in; J.blocked_out = nullptr; J.feature_query = dfq; J.nmatches = dn + 1; J.events = dev;
        J.mode = mode; J.nnratio = nnratio; J.th_high = th_high; J.check_ori = check_ori; J.big_ws = dbig;
        orbhip_launch_proj(J, c->stream);
        e = hipGetLastError();
    }
    TRY(arena_download(c->stream));
    if (e != hipSuccess) (void)hipStreamSynchronize(c->stream);            // never leave a copy in flight on the per-thread mirrors
    if (e == hipSuccess) *nmatches = hres[1];
#undef TRY
    return e == hipSuccess ? ORBHIP_OK : fail(ORBHIP_ERR_HIP, "search_by_projection_frame: %s", hipGetErrorString(e));
}
extern "C" orbhip_status orbhip_search_by_projection_frame(orbhip_ctx* c, int frame, int n, int use_u_right, const uint8_t* blocked,
                                                           const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                                           int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches)
{
    if (nq > 0 && !queries) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_by_projection_frame_impl(c, frame, n, use_u_right, blocked, queries, query_desc, nq, nullptr, nullptr, nullptr, mode, nnratio, th_high, check_ori, feature_query, nmatches);
}
extern "C" orbhip_status orbhip_project_search_frame(orbhip_ctx* c, int frame, int n, int use_u_right, const uint8_t* blocked,
                                                     const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                     float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches, orbhip_proj_query* queries_out)
{
    if (np > 0 && (!points || !proj)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return search_by_projection_frame_impl(c, frame, n, use_u_right, blocked, nullptr, point_desc, np, np > 0 ? proj : nullptr, np > 0 ? points : nullptr, queries_out, 1, nnratio, th_high, check_ori, feature_query, nmatches);
}
extern "C" orbhip_status orbhip_search_best_in_window_frame(orbhip_ctx* c, int frame, int n, int use_u_right, const orbhip_best_query* queries, const uint8_t* query_desc, int nq,
                                                            int chi2_gate, int32_t* best_idx, int32_t* best_dist)
{
    OrbApiTimer api_timer;
    const orbhip_keypoint* dk = nullptr; const uint8_t* dd = nullptr; const float* dur = nullptr;
    orbhip_status st = frame_args(c, frame, n, use_u_right, &dk, &dd, &dur); if (st != ORBHIP_OK) return st;
    if (nq < 0 || (nq > 0 && (!queries || !query_desc || !best_idx || !best_dist))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (n == 0 || nq == 0) return ORBHIP_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    uint8_t* dqd = nullptr; float* dsg = nullptr; int *dn = nullptr, *dgs = nullptr, *dgi = nullptr, *dbi = nullptr, *dbd = nullptr; float2* dgxy = nullptr; orbhip_best_query* dq = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const int hn[2] = {n, 0};
    TRY(arena_layout(c->cfg.device, [&](Arena& A) {
        A.io(&dqd, (size_t)nq * 32, query_desc, (size_t)nq * 32); A.io(&dq, nq, queries, nq); A.io(&dn, 8, hn, 2);
        A.io(&dsg, c->L, (const float*)c->is2.data(), (size_t)c->L);                               // mvInvLevelSigma2 of this extractor
        A.io(&dbi, nq, (const int*)nullptr, 0, best_idx, nq); A.io(&dbd, nq, (const int*)nullptr, 0, best_dist, nq);
        A.take(&dgs, ORBHIP_GRID_CELLS + 1); A.take(&dgi, n); A.take(&dgxy, n);
    }));
    TRY(arena_upload(c->stream));
    if (e == hipSuccess) {
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp2 = dk; M.n2 = dn; M.cap = n; M.min_x = c->bounds.min_x; M.min_y = c->bounds.min_y; M.max_x = c->bounds.max_x; M.max_y = c->bounds.max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.grid_all_levels = 1;
        if (frame == 0 && c->fgrid_valid && c->fgrid_cur == c->cur && c->last_n_valid && c->last_n[0] == n) { dgs = c->d_fgrid_start; dgi = c->d_fgrid_items; dgxy = c->d_fgrid_xy; }   // frame epilogue
        else orbhip_launch_match_grid(M, 1, c->stream);
        if (c->last_nimg == 1 || c->pair_mode) c->want_fgrid = true;
        BestParams B; memset(&B, 0, sizeof B);
        B.kp = dk; B.desc = dd; B.u_right = dur; B.inv_level_sigma2 = dsg; B.grid_start = dgs; B.grid_items = dgi; B.grid_xy = dgxy;
        B.q = dq; B.qdesc = dqd; B.nq = nq; B.chi2_gate = chi2_gate; B.best_idx = dbi; B.best_dist = dbd;
        B.min_x = c->bounds.min_x; B.gw_inv = (float)ORBHIP_GRID_COLS / (float)(c->bounds.max_x - c->bounds.min_x);
        orbhip_launch_best_in_window(B, c->stream);
        e = hipGetLastError();
    }
    TRY(arena_download(c->stream));
    if (e != hipSuccess) (void)hipStreamSynchronize(c->stream);            // never leave a copy in flight on the per-thread mirrors
#undef TRY
    return e == hipSuccess ? ORBHIP_OK : fail(ORBHIP_ERR_HIP, "search_best_in_window_frame: %s", hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------- stereo
static StereoSide stereo_side(orbhip_ctx* c)
{
    StereoSide S; memset(&S, 0, sizeof S);
    S.kp = c->d_out_kp[c->cur]; S.desc = c->d_out_desc[c->cur]; S.n = c->d_out_n[c->cur];
    S.img0 = c->last_img0; S.img0_frame_stride = c->last_img0_fstride; S.img0_pitch = c->last_img0_pitch;
    S.pyr = c->d_pyr; S.plane_frame_bytes = c->plane_frame_bytes;
    return S;
}

extern "C" orbhip_status orbhip_compute_stereo_matches(orbhip_ctx* l, orbhip_ctx* r, int nimg, float mbf, float mb, float* u_right, float* depth, int cap)
{
    OrbApiTimer api_timer;
    if (!l || !r || !u_right || !depth) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (l->cfg.device != r->cfg.device || l->cfg.width != r->cfg.width || l->cfg.height != r->cfg.height || l->L != r->L ||
        l->cfg.scale_factor != r->cfg.scale_factor || l->out_cap != r->out_cap)
        return fail(ORBHIP_ERR_INVALID, "left and right contexts must share device, image size, levels and scale factor");
    if (nimg < 1 || nimg > l->last_nimg || nimg > r->last_nimg) return fail(ORBHIP_ERR_INVALID, "nimg %d but the last calls processed %d / %d frames", nimg, l->last_nimg, r->last_nimg);
    if (!(mb > 0) || !(mbf > 0)) return fail(ORBHIP_ERR_INVALID, "mbf and mb must be positive");
    if (l->out_cap >= 65536) return fail(ORBHIP_ERR_UNSUPPORTED, "too many keypoints per frame for the stereo matcher");
    HIPCHK(hipSetDevice(l->cfg.device));
    // the right frame's results must be complete (left work is stream-ordered): they are if its caller already holds them
    orbhip_status st = ORBHIP_OK;
    if (!r->last_n_valid) { st = orbhip_sync(r); if (st != ORBHIP_OK) return st; }
    const size_t B = (size_t)l->B;
    if (!l->d_st_rowstart) {
        l->st_rowcap = stereo_row_cap(l);
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = dalloc(&l->d_st_rowstart, B * (l->cfg.height + 1));
        if (e == hipSuccess) e = dalloc(&l->d_st_rowitems, B * (size_t)l->st_rowcap);
        if (e == hipSuccess) e = dalloc(&l->d_st_u, 2 * B * l->out_cap);           // [mvuRight | mvDepth]: one allocation, one download when the call fills the context
        if (e == hipSuccess) l->d_st_depth = l->d_st_u + B * l->out_cap;
        if (e == hipSuccess) e = dalloc(&l->d_st_sad, B * l->out_cap);
        if (e != hipSuccess) return fail(ORBHIP_ERR_HIP, "stereo workspace allocation failed: %s", hipGetErrorString(e));
    }
    StereoParams T; memset(&T, 0, sizeof T);
    T.geom = l->d_geom; T.L = stereo_side(l); T.R = stereo_side(r);
    T.cap = l->out_cap; T.im_h = l->cfg.height;
    T.row_start = l->d_st_rowstart; T.row_items = l->d_st_rowitems; T.row_cap = l->st_rowcap;
    // the right context built its frame's row table behind its own extraction (frame epilogue): take it; from now on it always will
    const bool rows_ready = nimg == 1 && r->rrows_valid && r->rrows_cur == r->cur;
    if (rows_ready) { T.row_start = r->d_rrow_start; T.row_items = r->d_rrow_items; T.row_cap = r->rrow_cap; HIPCHK(hipStreamWaitEvent(l->stream, r->ev_epilogue, 0)); }
    if (nimg == 1) r->want_rrows = true;
    T.u_right = l->d_st_u; T.depth = l->d_st_depth; T.sad = l->d_st_sad; l->d_last_uright = l->d_st_u;
    T.mbf = mbf; T.maxD = mbf / mb;                                                 // minZ = mb, maxD = mbf/minZ (Frame.cc:496-498)
    orbhip_launch_stereo(T, nimg, l->out_cap, l->stream, rows_ready);
    HIPCHK(hipGetLastError());
    st = ensure_host_staging(l, false); if (st != ORBHIP_OK) return st;
    { const orbhip_status stf = mirrors_free(l, "orbhip_compute_stereo_matches"); if (stf != ORBHIP_OK) return stf; }
    const bool know_n = l->last_n_valid && (int)l->last_n.size() >= nimg;
    if (!know_n) HIPCHK(hipMemcpyAsync(l->h_n, l->d_out_n[l->cur], nimg * sizeof(int), hipMemcpyDeviceToHost, l->stream));
    float* hu = reinterpret_cast<float*>(l->h_kp); float* hd = hu + (size_t)nimg * l->out_cap;       // pinned mirror reused (28 B/keypoint >= 8 B)
    if ((size_t)nimg == B) HIPCHK(orbhip_copy_async(hu, l->d_st_u, 2 * B * l->out_cap * sizeof(float), hipMemcpyDeviceToHost, l->stream));
    else {
        HIPCHK(hipMemcpyAsync(hu, l->d_st_u, (size_t)nimg * l->out_cap * sizeof(float), hipMemcpyDeviceToHost, l->stream));
        HIPCHK(hipMemcpyAsync(hd, l->d_st_depth, (size_t)nimg * l->out_cap * sizeof(float), hipMemcpyDeviceToHost, l->stream));
    }
    HIPCHK(hipStreamSynchronize(l->stream));
    if (l->prof) prof_collect(l);
    for (int f = 0; f < nimg; f++) {
        const int m = std::min(know_n ? l->last_n[f] : l->h_n[f], cap);
        for (int i = 0; i < cap; i++) { u_right[(size_t)f * cap + i] = -1.0f; depth[(size_t)f * cap + i] = -1.0f; }
        if (m > 0) { memcpy(u_right + (size_t)f * cap, hu + (size_t)f * l->out_cap, m * sizeof(float)); memcpy(depth + (size_t)f * cap, hd + (size_t)f * l->out_cap, m * sizeof(float)); }
    }
    return ORBHIP_OK;
}

// The stereo pair as ONE call (include/orbhip.h): both images through one context with two camera slots - one staging copy + upload, one launch
// chain for both frames, the stereo matcher (slot 0 against slot 1) and the frame's feature grid queued behind it on the same stream; the host copies
// the key points out while the stereo kernels run, then picks up mvuRight / mvDepth.
extern "C" orbhip_status orbhip_extract_stereo(orbhip_ctx* c, const uint8_t* img_left, const uint8_t* img_right, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                                               float mbf, float mb, float* u_right, float* depth)
{
    OrbApiTimer api_timer;
    if (!c || !n_out || !u_right || !depth || cap < 0) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (c->B < 2) return fail(ORBHIP_ERR_INVALID, "orbhip_extract_stereo needs a context with max_batch >= 2 (this one has %d)", c->B);
    if (!(mb > 0) || !(mbf > 0)) return fail(ORBHIP_ERR_INVALID, "mbf and mb must be positive");
    if (c->out_cap >= 65536) return fail(ORBHIP_ERR_UNSUPPORTED, "too many keypoints per frame for the stereo matcher");
    n_out[0] = n_out[1] = 0;
    for (int i = 0; i < cap; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    if (!img_left || !img_right) return ORBHIP_OK;                      // an empty image: the reference's operator() returns silently, the frame has no features
    if (c->oldest_ticket != c->next_ticket) return fail(ORBHIP_ERR_INVALID, "orbhip_extract_stereo with %d submitted batches still in flight: collect them first", c->next_ticket - c->oldest_ticket);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t B = (size_t)c->B, oc = (size_t)c->out_cap;
    if (!c->d_st_rowstart) {
        c->st_rowcap = stereo_row_cap(c);
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = dalloc(&c->d_st_rowstart, B * (c->cfg.height + 1));
        if (e == hipSuccess) e = dalloc(&c->d_st_rowitems, B * (size_t)c->st_rowcap);
        if (e == hipSuccess) e = dalloc(&c->d_st_u, 2 * B * oc);
        if (e == hipSuccess) c->d_st_depth = c->d_st_u + B * oc;
        if (e == hipSuccess) e = dalloc(&c->d_st_sad, B * oc);
        if (e != hipSuccess) return fail(ORBHIP_ERR_HIP, "stereo workspace allocation failed: %s", hipGetErrorString(e));
    }
    if (!c->h_st) HIPCHK(hipHostMalloc((void**)&c->h_st, (B + 1) * oc * sizeof(float), hipHostMallocDefault));
    const uint8_t* imgs[2] = {img_left, img_right};
    int ticket = -1;
    orbhip_status st = submit_impl(c, 2, imgs, stride, nullptr, nullptr, 0, &ticket); if (st != ORBHIP_OK) return st;
    c->pair_mode = true;
    // slot 0 against slot 1 of this context
    StereoParams T; memset(&T, 0, sizeof T);
    T.geom = c->d_geom; T.L = stereo_side(c); T.R = stereo_side(c);
    T.R.kp += oc; T.R.desc += oc * 32; T.R.n += 1; T.R.img0 += T.R.img0_frame_stride; T.R.pyr += T.R.plane_frame_bytes;
    T.cap = c->out_cap; T.im_h = c->cfg.height; T.row_start = c->d_st_rowstart; T.row_items = c->d_st_rowitems; T.row_cap = c->st_rowcap;
    T.u_right = c->d_st_u; T.depth = c->d_st_depth; T.sad = c->d_st_sad; c->d_last_uright = c->d_st_u;
    T.mbf = mbf; T.maxD = mbf / mb;
    orbhip_launch_stereo(T, 1, c->out_cap, c->stream, false);
    hipError_t e = hipGetLastError();
    // [mvuRight of slot 0 .. mvDepth of slot 0]: ONE copy of (B + 1) * out_cap floats (slot 1's unused mvuRight rides along), then an event: the host waits for
    // that, not for the stream - the frame's feature grid (an epilogue of the searches to come, 20 us) is queued behind it and is nobody's business yet
    if (!c->ev_stereo) { if (hipEventCreateWithFlags(&c->ev_stereo, hipEventDisableTiming) != hipSuccess) e = hipErrorOutOfMemory; }
    if (e == hipSuccess) e = orbhip_copy_async(c->h_st, c->d_st_u, (B + 1) * oc * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipEventRecord(c->ev_stereo, c->stream);
    if (e == hipSuccess && c->want_fgrid) { const bool rr = c->want_rrows; c->want_rrows = false; const orbhip_status se = frame_epilogues(c, c->stream); c->want_rrows = rr; if (se != ORBHIP_OK) e = hipErrorInvalidValue; }
    // key points + descriptors of both images (waits for the result block only: the stereo kernels are still running)
    st = collect_flat(c, ticket, kps, desc, cap, n_out);
    const hipError_t es = e == hipSuccess ? hipEventSynchronize(c->ev_stereo) : hipStreamSynchronize(c->stream);
    if (c->prof) prof_collect(c);
    if (e != hipSuccess || es != hipSuccess) return fail(ORBHIP_ERR_HIP, "extract_stereo: %s", hipGetErrorString(e != hipSuccess ? e : es));
    if (st != ORBHIP_OK && st != ORBHIP_ERR_CAPACITY) return st;
    const int m = std::min(n_out[0], cap);
    if (m > 0) { memcpy(u_right, c->h_st, (size_t)m * sizeof(float)); memcpy(depth, c->h_st + B * oc, (size_t)m * sizeof(float)); }
    return st;
}

// ---------------------------------------------------------------------------------------------- measurement
extern "C" orbhip_status orbhip_profile_enable(orbhip_ctx* c, int on) { if (!c) return fail(ORBHIP_ERR_INVALID, "null context"); c->prof = on != 0; return ORBHIP_OK; }
extern "C" int orbhip_profile_num_kernels(const orbhip_ctx*) { return K_COUNT; }
extern "C" orbhip_status orbhip_profile_get(orbhip_ctx* c, int k, const char** name, double* total_ms, int64_t* launches)
{
    if (!c || k < 0 || k >= K_COUNT) return fail(ORBHIP_ERR_INVALID, "bad kernel index");
    prof_collect(c);
    if (name) *name = kKernelNames[k]; if (total_ms) *total_ms = c->tot_ms[k]; if (launches) *launches = c->launches[k];
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_profile_reset(orbhip_ctx* c)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    prof_collect(c);
    for (int k = 0; k < K_COUNT; k++) { c->tot_ms[k] = 0; c->launches[k] = 0; }
    return ORBHIP_OK;
}
extern "C" int64_t orbhip_algorithmic_bytes_per_frame(const orbhip_ctx* c)
{   // B(W,H,N) = P0 + (SP-P0) + (SP-P7) + SP + 2*SP + N*(749+512) + N*(28+32)      
    if (!c) return 0;
    long long SP = 0; for (auto& g : c->geom) SP += (long long)g.w * g.h;
    const long long P0 = (long long)c->geom[0].w * c->geom[0].h, PL = (long long)c->geom[c->L - 1].w * c->geom[c->L - 1].h, N = c->cfg.nfeatures;
    return P0 + (SP - P0) + (SP - PL) + SP + 2 * SP + N * (749 + 512) + N * (28 + 32);
}
extern "C" int64_t orbhip_algorithmic_bytes_per_frame_kernel(const orbhip_ctx* c, int k)
{   // the terms of B(W,H,N) attributed to the kernel that moves them
    if (!c) return 0;
    long long SP = 0; for (auto& g : c->geom) SP += (long long)g.w * g.h;
    const long long P0 = (long long)c->geom[0].w * c->geom[0].h, PL = (long long)c->geom[c->L - 1].w * c->geom[c->L - 1].h, N = c->cfg.nfeatures;
    switch (k) {
        case K_PYRAMID: return (SP - PL) + (SP - P0);
        case K_FAST: return SP;
        case K_BLUR: return 2 * SP;
        case K_DESCRIBE: return N * (749 + 512) + N * (28 + 32);
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------- camera geometry
static bool camera_ok(const orbhip_camera* cam) { return cam && cam->fx != 0.0f && cam->fy != 0.0f; }
static CameraD widen(const orbhip_camera& k)
{   // cvUndistortPoints converts the CV_32F mK / mDistCoef to double and forms ifx = 1./fx on the host
    CameraD C; C.fx = k.fx; C.fy = k.fy; C.cx = k.cx; C.cy = k.cy; C.ifx = 1. / C.fx; C.ify = 1. / C.fy; C.k1 = k.k1; C.k2 = k.k2; C.p1 = k.p1; C.p2 = k.p2; C.k3 = k.k3;
    return C;
}
extern "C" orbhip_status orbhip_undistort_points(int device, const orbhip_camera* cam, const float* xy, int n, float* xy_out)
{
    OrbApiTimer api_timer;
    if (!camera_ok(cam) || n < 0 || (n > 0 && (!xy || !xy_out))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (n == 0) return ORBHIP_OK;
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    float *din = nullptr, *dout = nullptr;
    HIPCHK(arena_layout(device, [&](Arena& A) { A.take(&din, (size_t)n * 2); A.take(&dout, (size_t)n * 2); }));
    HIPCHK(hipMemcpyAsync(din, xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ts));
    orbhip_launch_undistort_points(widen(*cam), din, n, dout, ts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(xy_out, dout, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ts));
    HIPCHK(hipStreamSynchronize(ts));
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_image_bounds(int device, const orbhip_camera* cam, int im_w, int im_h, orbhip_bounds* out)
{
    OrbApiTimer api_timer;
    if (!camera_ok(cam) || !out || im_w < 1 || im_h < 1) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (cam->k1 == 0.0f) { out->min_x = 0.0f; out->max_x = (float)im_w; out->min_y = 0.0f; out->max_y = (float)im_h; return ORBHIP_OK; }     // Frame.cc:455-463
    const float corners[8] = {0.0f, 0.0f, (float)im_w, 0.0f, 0.0f, (float)im_h, (float)im_w, (float)im_h};                                 // Frame.cc:440-444
    float m[8];
    const orbhip_status st = orbhip_undistort_points(device, cam, corners, 4, m); if (st != ORBHIP_OK) return st;
    out->min_x = std::min(m[0], m[4]); out->max_x = std::max(m[2], m[6]); out->min_y = std::min(m[1], m[3]); out->max_y = std::max(m[5], m[7]);   // Frame.cc:451-454
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_set_camera(orbhip_ctx* c, const orbhip_camera* cam)
{
    if (!c || (cam && !camera_ok(cam))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    orbhip_status st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    const bool distorted = cam && cam->k1 != 0.0f;                        // if(mDistCoef.at<float>(0)==0.0) mvKeysUn = mvKeys  (Frame.cc:406-410)
    orbhip_bounds b = {0.0f, 0.0f, (float)c->cfg.width, (float)c->cfg.height};
    if (distorted) {
        st = orbhip_image_bounds(c->cfg.device, cam, c->cfg.width, c->cfg.height, &b); if (st != ORBHIP_OK) return st;
        if (!(b.max_x > b.min_x) || !(b.max_y > b.min_y)) return fail(ORBHIP_ERR_INVALID, "the distortion model folds the image corners (bounds %g..%g x %g..%g)", b.min_x, b.max_x, b.min_y, b.max_y);
        for (int k = 0; k < 3; k++) if (!c->d_out_kpun[k]) HIPCHK(dalloc(&c->d_out_kpun[k], (size_t)c->B * c->out_cap));
        c->cam = widen(*cam);
    }
    c->distorted = distorted; c->bounds = b;
    // frames extracted under the previous camera are no "previous frame" for the matcher any more
    for (int k = 0; k < 3; k++) { HIPCHK(hipMemsetAsync(c->d_out_n[k], 0, (size_t)c->B * sizeof(int), c->stream)); HIPCHK(hipMemsetAsync(c->d_lvl_n[k], 0, (size_t)c->B * c->L * sizeof(int), c->stream)); }
    c->last_nimg = 0; c->last_matched = false;
    return orbhip_sync(c);
}
extern "C" orbhip_status orbhip_get_bounds(const orbhip_ctx* c, orbhip_bounds* out)
{
    if (!c || !out) return fail(ORBHIP_ERR_INVALID, "null argument");
    *out = c->bounds; return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_fetch_undistorted(orbhip_ctx* c, int nimg, orbhip_keypoint* kps_un, int cap)
{
    OrbApiTimer api_timer;
    if (!c || !kps_un || cap < 0) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (nimg < 1 || nimg > c->last_nimg) return fail(ORBHIP_ERR_INVALID, "nimg %d but the last call processed %d frames", nimg, c->last_nimg);
    HIPCHK(hipSetDevice(c->cfg.device));
    orbhip_status st = ensure_host_staging(c, false); if (st != ORBHIP_OK) return st;
    if (!c->h_kpun) HIPCHK(hipHostMalloc((void**)&c->h_kpun, (size_t)c->B * c->out_cap * sizeof(orbhip_keypoint), hipHostMallocDefault));
    { const orbhip_status stf = mirrors_free(c, "orbhip_fetch_undistorted"); if (stf != ORBHIP_OK) return stf; }
    HIPCHK(orbhip_copy_async(c->h_n, c->d_out_n[c->cur], nimg * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(orbhip_copy_async(c->h_kpun, (c->distorted ? c->d_out_kpun : c->d_out_kp)[c->cur], (size_t)nimg * c->out_cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, c->stream));
    st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    bool overflow = false;
    for (int f = 0; f < nimg; f++) {
        const int m = std::min(c->h_n[f], cap);
        if (c->h_n[f] > cap) overflow = true;
        if (m > 0) memcpy(kps_un + (size_t)f * cap, c->h_kpun + (size_t)f * c->out_cap, (size_t)m * sizeof(orbhip_keypoint));
    }
    return overflow ? fail(ORBHIP_ERR_CAPACITY, "keypoint buffer too small") : ORBHIP_OK;
}

// Frame::ComputeStereoFromRGBD (Frame.cc:643-665) on the key points the last extraction left in HBM
extern "C" orbhip_status orbhip_compute_stereo_from_rgbd(orbhip_ctx* c, int nimg, const void* const* depth_maps, int stride_bytes, int depth_type, float depth_factor,
                                                         float mbf, float* u_right, float* depth, int cap)
{
    OrbApiTimer api_timer;
    if (!c || !depth_maps || !u_right || !depth || cap < 0 || (depth_type != 0 && depth_type != 1)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (nimg < 1 || nimg > c->last_nimg) return fail(ORBHIP_ERR_INVALID, "nimg %d but the last call processed %d frames", nimg, c->last_nimg);
    const int esz = depth_type == 0 ? 4 : 2, W = c->cfg.width, H = c->cfg.height;
    if (stride_bytes < W * esz) return fail(ORBHIP_ERR_INVALID, "depth row stride %d < %d", stride_bytes, W * esz);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t pitch = ((size_t)W * 4 + 63) & ~(size_t)63, fbytes = pitch * H, need = (size_t)c->B * fbytes + (size_t)2 * c->B * c->out_cap * sizeof(float);
    if (c->depth_bytes < need) { if (c->d_depth) (void)hipFree(c->d_depth); c->d_depth = nullptr; c->depth_bytes = 0; HIPCHK(orbhip_dmalloc((void**)&c->d_depth, need)); c->depth_bytes = need; }
    float* d_u = (float*)(c->d_depth + (size_t)c->B * fbytes); float* d_z = d_u + (size_t)c->B * c->out_cap; c->d_last_uright = d_u;
    for (int f = 0; f < nimg; f++) {
        if (!depth_maps[f]) return fail(ORBHIP_ERR_INVALID, "depth map %d is null", f);
        HIPCHK(hipMemcpy2DAsync(c->d_depth + f * fbytes, pitch, depth_maps[f], (size_t)stride_bytes, (size_t)W * esz, H, hipMemcpyHostToDevice, c->stream));
    }
    const int convert = (std::fabs(depth_factor - 1.0f) > 1e-5f) || depth_type != 0;          // Tracking.cc:226
    orbhip_launch_stereo_from_rgbd(c->d_out_kp[c->cur], (c->distorted ? c->d_out_kpun : c->d_out_kp)[c->cur], c->d_out_n[c->cur], c->out_cap, c->d_depth, (long long)fbytes,
                                   (int)pitch, depth_type, convert, depth_factor, mbf, d_u, d_z, nimg, c->stream);
    HIPCHK(hipGetLastError());
    orbhip_status st = ensure_host_staging(c, false); if (st != ORBHIP_OK) return st;
    { const orbhip_status stf = mirrors_free(c, "orbhip_compute_stereo_from_rgbd"); if (stf != ORBHIP_OK) return stf; }
    HIPCHK(hipMemcpyAsync(c->h_n, c->d_out_n[c->cur], nimg * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    for (int f = 0; f < nimg; f++) {
        const int m = std::min(c->h_n[f], cap);
        if (m > 0) {
            HIPCHK(hipMemcpy(u_right + (size_t)f * cap, d_u + (size_t)f * c->out_cap, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(depth + (size_t)f * cap, d_z + (size_t)f * c->out_cap, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    return ORBHIP_OK;
}

// mvuRight computed by the caller (Frame::ComputeStereoFromRGBD's own loop, Frame.cc:643-665: N samples of a depth map that lives in host memory) handed to
// the frame that is still on the device, so that the resident searches' right-coordinate test (ORBmatcher.cc:1418-1424, 96-101) reads it in HBM: N floats travel,
// not the depth map.  Asynchronous on the context's stream; the values are copied before the call returns.
extern "C" orbhip_status orbhip_set_stereo_columns(orbhip_ctx* c, int frame, const float* u_right, int n)
{
    OrbApiTimer api_timer;
    if (!c || (!u_right && n > 0)) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (frame < 0 || frame >= c->last_nimg) return fail(ORBHIP_ERR_INVALID, "frame %d outside the %d frames of the last extraction", frame, c->last_nimg);
    if (n < 0 || n > c->out_cap) return fail(ORBHIP_ERR_INVALID, "n %d outside 0..%d", n, c->out_cap);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t total = (size_t)c->B * c->out_cap;
    if (!c->d_ucols) {
        HIPCHK(orbhip_dmalloc((void**)&c->d_ucols, total * sizeof(float)));
        HIPCHK(hipHostMalloc((void**)&c->h_ucols, total * sizeof(float), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->ev_ucols, hipEventDisableTiming));
    }
    if (c->ucols_pending) { HIPCHK(hipEventSynchronize(c->ev_ucols)); c->ucols_pending = false; }      // the pinned block is free again
    if (c->d_last_uright != c->d_ucols) {
        // the block becomes the extraction's mvuRight for EVERY frame: the other frames keep the columns a stereo / RGB-D step left for them, or read
        // "no right coordinate" (-1, Frame.cc:468) - never whatever the allocation held
        if (c->d_last_uright) HIPCHK(orbhip_copy_async(c->d_ucols, c->d_last_uright, total * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        else HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->d_ucols, 0xBF800000, total, c->stream));
    }
    if (n > 0) {
        memcpy(c->h_ucols + (size_t)frame * c->out_cap, u_right, (size_t)n * sizeof(float));
        HIPCHK(orbhip_copy_async(c->d_ucols + (size_t)frame * c->out_cap, c->h_ucols + (size_t)frame * c->out_cap, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->ev_ucols, c->stream)); c->ucols_pending = true;
    }
    c->d_last_uright = c->d_ucols;
    return ORBHIP_OK;
}

// Rectification of raw stereo frames (stereo_euroc.cc:136-137): remapped on the device into the context's level-0 plane
extern "C" orbhip_status orbhip_set_rectification(orbhip_ctx* c, const float* map_x, const float* map_y, int src_w, int src_h)
{
    if (!c) return fail(ORBHIP_ERR_INVALID, "null context");
    orbhip_status st = orbhip_sync(c); if (st != ORBHIP_OK) return st;
    if (!map_x) { c->src_w = c->src_h = 0; return ORBHIP_OK; }
    if (!map_y || src_w < 1 || src_h < 1 || src_w > 32767 || src_h > 32767) return fail(ORBHIP_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(c->cfg.device));
    const int qp = (c->cfg.width + 3) & ~3;
    const size_t n = (size_t)qp * c->cfg.height;
    if (!c->d_map_x) { HIPCHK(dalloc(&c->d_map_x, n)); HIPCHK(dalloc(&c->d_map_y, n)); }
    // cv::remap converts the float maps to 5 fractional bits for every image (RemapInvoker: cvRound(map * INTER_TAB_SIZE)); the maps of
    // a camera never change, so the table is built here once.  cvRound = cvtss2si: round-half-even, INT_MIN when out of range / NaN.
    std::vector<int> q(2 * n, 0);
    for (int y = 0; y < c->cfg.height; y++)
        for (int x = 0; x < c->cfg.width; x++) {
            const float vx = map_x[(size_t)y * c->cfg.width + x] * 32.0f, vy = map_y[(size_t)y * c->cfg.width + x] * 32.0f;
            q[(size_t)y * qp + x] = std::fabs(vx) < 2147483648.0f ? (int)lrintf(vx) : INT32_MIN;
            q[n + (size_t)y * qp + x] = std::fabs(vy) < 2147483648.0f ? (int)lrintf(vy) : INT32_MIN;
        }
    HIPCHK(hipMemcpy(c->d_map_x, q.data(), n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_map_y, q.data() + n, n * sizeof(int), hipMemcpyHostToDevice));
    if (c->d_raw && (src_w != c->src_w || src_h != c->src_h)) { (void)hipFree(c->d_raw); (void)hipHostFree(c->h_raw); c->d_raw = nullptr; c->h_raw = nullptr; }
    c->src_w = src_w; c->src_h = src_h; c->raw_pitch = (src_w + 63) & ~63;
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_extract_device_rectify(orbhip_ctx* c, int nimg, const uint8_t* d_raw, size_t frame_stride, int row_stride,
                                                       int match_prev, int window, float nnratio, int check_ori)
{
    if (!c || !d_raw) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (c->src_w < 1) return fail(ORBHIP_ERR_INVALID, "no rectification maps: call orbhip_set_rectification first");
    if (nimg < 1 || nimg > c->B) return fail(ORBHIP_ERR_INVALID, "nimg %d outside 1..%d", nimg, c->B);
    if (row_stride < c->src_w) return fail(ORBHIP_ERR_INVALID, "row stride %d < raw width %d", row_stride, c->src_w);
    HIPCHK(hipSetDevice(c->cfg.device));
    orbhip_status st = ensure_host_staging(c, true); if (st != ORBHIP_OK) return st;
    const size_t fbytes = (size_t)c->in_pitch * c->cfg.height;
    {
        ProfScope ps(c, K_REMAP, c->stream);
        orbhip_launch_remap(d_raw, (long long)frame_stride, row_stride, c->src_w, c->src_h, c->d_map_x, c->d_map_y, (c->cfg.width + 3) & ~3, c->d_in, (long long)fbytes, c->in_pitch,
                            c->cfg.width, c->cfg.height, nimg, c->stream);
    }
    HIPCHK(hipGetLastError());
    c->last_from_host = true; c->last_d_in = c->d_in; c->plane0_dirty = true;   // level 0 lives in the context's own plane
    return run_pipeline(c, nimg, c->d_in, (long long)fbytes, c->in_pitch, match_prev, window, nnratio, check_ori);
}
extern "C" orbhip_status orbhip_extract_batch_rectify(orbhip_ctx* c, int nimg, const uint8_t* const* imgs, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    OrbApiTimer api_timer;
    if (!c || !imgs || !n_out) return fail(ORBHIP_ERR_INVALID, "null argument");
    if (c->src_w < 1) return fail(ORBHIP_ERR_INVALID, "no rectification maps: call orbhip_set_rectification first");
    if (nimg < 1 || nimg > c->B) return fail(ORBHIP_ERR_INVALID, "nimg %d outside 1..%d", nimg, c->B);
    if (stride < c->src_w) return fail(ORBHIP_ERR_INVALID, "stride %d < raw width %d", stride, c->src_w);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t rfbytes = (size_t)c->raw_pitch * c->src_h;
    if (!c->d_raw) {
        HIPCHK(orbhip_dmalloc((void**)&c->d_raw, (size_t)c->B * rfbytes + 256));
        HIPCHK(hipHostMalloc((void**)&c->h_raw, (size_t)c->B * rfbytes + 256, hipHostMallocDefault));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int f = 0; f < nimg; f++) {
        if (!imgs[f]) return fail(ORBHIP_ERR_INVALID, "image %d is null", f);
        uint8_t* dst = c->h_raw + f * rfbytes;
        for (int y = 0; y < c->src_h; y++) memcpy(dst + (size_t)y * c->raw_pitch, imgs[f] + (size_t)y * stride, c->src_w);
    }
    HIPCHK(hipMemcpyAsync(c->d_raw, c->h_raw, nimg * rfbytes, hipMemcpyHostToDevice, c->stream));
    const orbhip_status st = orbhip_extract_device_rectify(c, nimg, c->d_raw, rfbytes, c->raw_pitch, 0, 0, 0.f, 0);
    if (st != ORBHIP_OK) return st;
    return orbhip_fetch(c, nimg, kps, desc, cap, n_out);
}

// ---------------------------------------------------------------------------------------------- stateless matcher entry points
extern "C" int orbhip_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    unsigned long long x[4], y[4]; memcpy(x, a, 32); memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

// Partials of the brute-force scan: a per-thread, grow-only buffer tied to the device it was allocated on and to the stream that used it
// last (a second stream of the same thread waits for the first before it reuses the buffer).
void* orbhip_nn_workspace(size_t bytes, hipStream_t s)
{
    orbhip_touch_thread_caches();
    int dev = -1; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (g_nn_ws && g_nn_ws_dev == dev && g_nn_ws_stream != s) (void)hipStreamSynchronize(g_nn_ws_stream);
    if (g_nn_ws_dev != dev || bytes > g_nn_ws_bytes) {
        if (g_nn_ws) { const int cur = dev; (void)hipSetDevice(g_nn_ws_dev); (void)hipStreamSynchronize(g_nn_ws_stream); (void)hipFree(g_nn_ws); (void)hipSetDevice(cur); g_nn_ws = nullptr; g_nn_ws_bytes = 0; }
        if (orbhip_dmalloc(&g_nn_ws, bytes) != hipSuccess) { (void)hipGetLastError(); g_nn_ws = nullptr; g_nn_ws_dev = -1; return nullptr; }
        g_nn_ws_bytes = bytes; g_nn_ws_dev = dev;
    }
    g_nn_ws_stream = s;
    return g_nn_ws;
}

extern "C" orbhip_status orbhip_hamming_nn_device(void* stream, const uint8_t* d_q, int nq, const uint8_t* d_db, int64_t ndb, int64_t base,
                                                  int64_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second)
{
    if (nq < 0 || ndb < 0 || (nq > 0 && (!d_q || !d_best_idx || !d_best_dist || !d_second)) || (ndb > 0 && !d_db)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (!orbhip_launch_hamming_nn(d_q, nq, d_db, ndb, base, (long long*)d_best_idx, d_best_dist, d_second, (hipStream_t)stream)) return fail(ORBHIP_ERR_HIP, "hamming_nn: no device memory for the scan partials");
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

// A database that is queried many times (a key frame database: BASELINE.json config 5) expanded ONCE into the form the FP4 scan multiplies - 128 bytes per row
// instead of 32 - so that a query stages tiles by LDS-DMA instead of expanding every row again for every 512 queries (include/orbhip.h)
extern "C" size_t orbhip_nn_expanded_size(int64_t ndb) { return ndb < 0 ? 0 : orbhip_nn_expanded_bytes(ndb); }
extern "C" orbhip_status orbhip_nn_expand_device(void* stream, const uint8_t* d_db, int64_t ndb, uint8_t* d_expanded)
{
    if (ndb < 0 || (ndb > 0 && (!d_db || !d_expanded))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (((uintptr_t)d_expanded & 15) != 0) return fail(ORBHIP_ERR_INVALID, "the expanded database must be 16-byte aligned");
    orbhip_launch_nn_expand(d_db, ndb, d_expanded, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}
extern "C" orbhip_status orbhip_hamming_nn_device_expanded(void* stream, const uint8_t* d_q, int nq, const uint8_t* d_db, const uint8_t* d_expanded, int64_t ndb, int64_t base,
                                                           int64_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second)
{
    if (nq < 0 || ndb < 0 || (nq > 0 && (!d_q || !d_best_idx || !d_best_dist || !d_second)) || (ndb > 0 && (!d_db || !d_expanded))) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (!orbhip_launch_hamming_nn(d_q, nq, d_db, ndb, base, (long long*)d_best_idx, d_best_dist, d_second, (hipStream_t)stream, d_expanded)) return fail(ORBHIP_ERR_HIP, "hamming_nn: no device memory for the scan partials");
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_hamming_nn(int device, const uint8_t* q, int nq, const uint8_t* db, int64_t ndb, int64_t base,
                                           int64_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    if (nq < 0 || ndb < 0 || (nq > 0 && (!q || !best_idx || !best_dist || !second_dist)) || (ndb > 0 && !db)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    if (nq == 0) return ORBHIP_OK;
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    uint8_t *dq = nullptr, *ddb = nullptr; long long* dbi = nullptr; int *dbd = nullptr, *dsd = nullptr;
    orbhip_status st = ORBHIP_OK;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&dq, (size_t)nq * 32);
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&ddb, std::max<size_t>((size_t)ndb * 32, 32));
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&dbi, (size_t)nq * 8);
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&dbd, (size_t)nq * 4);
    if (e == hipSuccess) e = orbhip_dmalloc((void**)&dsd, (size_t)nq * 4);
    if (e == hipSuccess) e = hipMemcpy(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess && ndb > 0) e = hipMemcpy(ddb, db, (size_t)ndb * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) { e = orbhip_launch_hamming_nn(dq, nq, ddb, ndb, base, dbi, dbd, dsd, nullptr) ? hipGetLastError() : hipErrorOutOfMemory; }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(best_idx, dbi, (size_t)nq * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(best_dist, dbd, (size_t)nq * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(second_dist, dsd, (size_t)nq * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) st = fail(ORBHIP_ERR_HIP, "hamming_nn: %s", hipGetErrorString(e));
    (void)hipFree(dq); (void)hipFree(ddb); (void)hipFree(dbi); (void)hipFree(dbd); (void)hipFree(dsd);
    return st;
}

extern "C" orbhip_status orbhip_search_for_initialization_bounds(int device, const orbhip_keypoint* kps1, const uint8_t* desc1, int n1,
                                                          const orbhip_keypoint* kps2, const uint8_t* desc2, int n2, const orbhip_bounds* bounds,
                                                          float* prev_matched, int32_t* matches12, int window, float nnratio, int check_ori, int* nmatches)
{
    OrbApiTimer api_timer;
    if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && (!kps1 || !desc1 || !prev_matched || !matches12)) || (n2 > 0 && (!kps2 || !desc2)) || !bounds || !(bounds->max_x > bounds->min_x) || !(bounds->max_y > bounds->min_y))
        return fail(ORBHIP_ERR_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0) return ORBHIP_OK;
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ORBHIP_ERR_HIP, "no HIP device available: no CPU fallback");
    HIPCHK(hipSetDevice(device));
    hipStream_t ts = orbhip_thread_stream(device);
    // Frame members flattened: level-0 keypoints of F1 in index order (the loop at ORBmatcher.cc:418-423 skips the rest)
    std::vector<int> list1; for (int i = 0; i < n1; i++) if (kps1[i].octave <= 0) list1.push_back(i);
    int n2l0 = 0; for (int i = 0; i < n2; i++) n2l0 += kps2[i].octave == 0;
    const int cap = std::max(std::max(n1, n2), 1), l0cap = std::max((int)list1.size(), 1), cstride = std::max(n2l0, 1);
    orbhip_keypoint *dk1 = nullptr, *dk2 = nullptr; uint8_t *dd1 = nullptr, *dd2 = nullptr; int *dn = nullptr, *dlist = nullptr, *dgs = nullptr, *dgi = nullptr, *dnc = nullptr, *dm12 = nullptr, *dbig = nullptr; float2* dgxy = nullptr;
    unsigned* dcand = nullptr; unsigned* dtop = nullptr; float* dprev = nullptr;
    hipError_t e = hipSuccess;
#define TRY(x) do { if (e == hipSuccess) e = (x); } while (0)
    const int hn[4] = {n1, n2, (int)list1.size(), 0}; int hres[4] = {0, 0, 0, 0};
    TRY(arena_layout(device, [&](Arena& A) {
        A.io(&dk1, cap, kps1, n1); A.io(&dk2, cap, kps2, n2); A.io(&dd1, (size_t)cap * 32, desc1, (size_t)n1 * 32); A.io(&dd2, (size_t)cap * 32, desc2, (size_t)n2 * 32);
        A.io(&dlist, l0cap, (const int*)list1.data(), list1.size());
        A.io(&dn, 8, hn, 4, hres, 4);                          // counts in, [3] = nmatches out
        A.io(&dprev, (size_t)cap * 2, (const float*)prev_matched, (size_t)n1 * 2, prev_matched, (size_t)n1 * 2);
        A.io(&dm12, cap, (const int*)nullptr, 0, matches12, n1);
        A.take(&dgs, ORBHIP_GRID_CELLS + 1); A.take(&dgi, cap); A.take(&dgxy, cap); A.take(&dnc, l0cap);
        A.take(&dcand, (size_t)l0cap * cstride); A.take(&dtop, (size_t)l0cap * 5);
        if (orbhip_match_select_big(cap, l0cap)) A.take(&dbig, orbhip_match_select_ints(cap, l0cap));      // the select kernel's tables when they do not fit LDS
    }));
    TRY(arena_upload(ts));
    if (e == hipSuccess) {
        MatchParams M; memset(&M, 0, sizeof M);
        M.kp1 = dk1; M.desc1 = dd1; M.n1 = dn; M.n1_lvl0 = dn + 2; M.kp2 = dk2; M.desc2 = dd2; M.n2 = dn + 1; M.lvl_stride = 0; M.list1 = dlist; M.prev_from_kp1 = 0;
        M.cap = cap; M.min_x = bounds->min_x; M.min_y = bounds->min_y; M.max_x = bounds->max_x; M.max_y = bounds->max_y; M.grid_start = dgs; M.grid_items = dgi; M.grid_xy = dgxy; M.cand = dcand; M.top = dtop; M.ncand = dnc; M.cand_stride = cstride; M.lvl0_cap = l0cap;
        M.prev = dprev; M.matches12 = dm12; M.nmatches = dn + 3; M.window = window; M.nnratio = nnratio; M.check_ori = check_ori; M.big_ws = dbig;
        orbhip_launch_match_grid(M, 1, ts); orbhip_launch_match_candidates(M, 1, ts); orbhip_launch_match_select(M, 1, ts);
        e = hipGetLastError();
    }
    TRY(arena_download(ts));
    if (e != hipSuccess) (void)hipStreamSynchronize(ts);            // never leave a copy in flight on the per-thread mirrors
    if (e == hipSuccess) *nmatches = hres[3];
#undef TRY
    orbhip_status st = ORBHIP_OK;
    if (e != hipSuccess) st = fail(ORBHIP_ERR_HIP, "search_for_initialization: %s", hipGetErrorString(e));
    return st;
}

// the im_w / im_h forms: an undistorted camera, mnMinX = mnMinY = 0, mnMaxX = cols, mnMaxY = rows (Frame.cc:455-463)
static bool whole_image(int im_w, int im_h, orbhip_bounds* b) { if (im_w < 1 || im_h < 1) return false; b->min_x = 0.0f; b->min_y = 0.0f; b->max_x = (float)im_w; b->max_y = (float)im_h; return true; }
extern "C" orbhip_status orbhip_search_by_projection(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, const uint8_t* blocked, int n,
                                                     int im_w, int im_h, const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                                     int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches)
{
    orbhip_bounds b; if (!whole_image(im_w, im_h, &b)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return orbhip_search_by_projection_bounds(device, kps, desc, u_right, blocked, n, &b, queries, query_desc, nq, mode, nnratio, th_high, check_ori, feature_query, nmatches);
}
extern "C" orbhip_status orbhip_search_best_in_window(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right, int n, int im_w, int im_h,
                                                      const float* inv_level_sigma2, int nlevels, const orbhip_best_query* queries, const uint8_t* query_desc, int nq,
                                                      int chi2_gate, int32_t* best_idx, int32_t* best_dist)
{
    orbhip_bounds b; if (!whole_image(im_w, im_h, &b)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return orbhip_search_best_in_window_bounds(device, kps, desc, u_right, n, &b, inv_level_sigma2, nlevels, queries, query_desc, nq, chi2_gate, best_idx, best_dist);
}
extern "C" orbhip_status orbhip_search_for_initialization(int device, const orbhip_keypoint* kps1, const uint8_t* desc1, int n1,
                                                          const orbhip_keypoint* kps2, const uint8_t* desc2, int n2, int im_w, int im_h,
                                                          float* prev_matched, int32_t* matches12, int window, float nnratio, int check_ori, int* nmatches)
{
    orbhip_bounds b; if (!whole_image(im_w, im_h, &b)) return fail(ORBHIP_ERR_INVALID, "bad argument");
    return orbhip_search_for_initialization_bounds(device, kps1, desc1, n1, kps2, desc2, n2, &b, prev_matched, matches12, window, nnratio, check_ori, nmatches);
}

// ---------------------------------------------------------------------------------------------- relocalisation candidates
// Stands where Tracking::Relocalization asks KeyFrameDatabase::DetectRelocalizationCandidates for key frames that share words with the
// frame (Tracking.cc:1344-1348, KeyFrameDatabase.cc:199-309): here the evidence is the brute-force nearest neighbour of every query
// descriptor over the descriptors of ALL key frames (BASELINE.json config 5), filtered with the matcher's own acceptance idiom
// (distance threshold + ratio to the second best, ORBmatcher.cc:102-114), one vote per accepted descriptor for the owning key frame.
// MapPoint::PredictScale as a table (include/orbhip.h): level_ratio[i] = the smallest positive float ratio the caller's own expression maps to a level > i.
// Host arithmetic only (the caller's libm through level_of); the device compares ratios against the table (pj_predict_scale).
extern "C" orbhip_status orbhip_predict_scale_table(int (*level_of)(float ratio, void* user), void* user, int nlevels, float* level_ratio)
{
    if (!level_of || !level_ratio || nlevels < 1 || nlevels > ORBHIP_MAX_PROJ_LEVELS) return fail(ORBHIP_ERR_INVALID, "bad argument");
    auto as_float = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
    const float inf = as_float(0x7f800000u);
    for (int i = 0; i < ORBHIP_MAX_PROJ_LEVELS; i++) level_ratio[i] = inf;
    for (int i = 0; i + 1 < nlevels; i++) {
        auto above = [&](uint32_t b) { return level_of(as_float(b), user) > i; };
        uint32_t lo = 1u, hi = 0x7f7fffffu;                                  // smallest denormal .. largest finite float: positive floats order like their bits
        if (!above(hi)) continue;                                            // no finite ratio reaches level i + 1
        if (above(lo)) { level_ratio[i] = as_float(lo); continue; }
        while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (above(mid)) hi = mid; else lo = mid; }
        for (uint32_t d = 1; d <= 64u; d++) {                                 // a step function: nothing above the threshold falls back, nothing below reaches over
            if (hi + d <= 0x7f7fffffu && !above(hi + d)) return fail(ORBHIP_ERR_UNSUPPORTED, "PredictScale is not monotone in the distance ratio near %.9g (level %d)", (double)as_float(hi), i + 1);
            if (lo >= d && lo - d >= 1u && above(lo - d)) return fail(ORBHIP_ERR_UNSUPPORTED, "PredictScale is not monotone in the distance ratio near %.9g (level %d)", (double)as_float(hi), i + 1);
        }
        level_ratio[i] = as_float(hi);
    }
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_reloc_candidates(const int64_t* best_idx, const int32_t* best_dist, const int32_t* second_dist, int nq,
                                                 const int32_t* row_keyframe, int64_t ndb, int nkf, int th_dist, float ratio,
                                                 int top_k, int32_t* kf_out, int32_t* votes_out, int* nout)
{
    if (nq < 0 || ndb < 0 || nkf < 0 || top_k < 0 || !nout || (nq > 0 && (!best_idx || !best_dist || !second_dist)) || (ndb > 0 && !row_keyframe) || (top_k > 0 && (!kf_out || !votes_out)))
        return fail(ORBHIP_ERR_INVALID, "bad argument");
    *nout = 0;
    std::vector<int> votes((size_t)std::max(nkf, 1), 0);
    for (int i = 0; i < nq; i++) {
        const int64_t r = best_idx[i];
        if (r < 0 || r >= ndb) continue;
        if (best_dist[i] > th_dist) continue;
        if (!((float)best_dist[i] < ratio * (float)second_dist[i])) continue;
        const int kf = row_keyframe[r];
        if (kf < 0 || kf >= nkf) return fail(ORBHIP_ERR_INVALID, "row %lld belongs to key frame %d outside 0..%d", (long long)r, kf, nkf - 1);
        votes[kf]++;
    }
    std::vector<int> order; order.reserve(nkf);
    for (int k = 0; k < nkf; k++) if (votes[k] > 0) order.push_back(k);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return votes[a] > votes[b]; });      // ties keep ascending key frame id
    const int m = std::min<int>(top_k, (int)order.size());
    for (int i = 0; i < m; i++) { kf_out[i] = order[i]; votes_out[i] = votes[order[i]]; }
    *nout = m;
    return ORBHIP_OK;
}
