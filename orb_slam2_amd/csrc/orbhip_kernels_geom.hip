// orbhip_kernels_geom.hip — camera geometry on either side of the extractor:
//   output side: Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:404-464) = cv::undistortPoints(pts, pts, mK, mDistCoef, Mat(), mK)
//   input side:  the stereo rectification the EuRoC example runs per image, cv::remap(raw, rect, M1, M2, INTER_LINEAR)
//                (Examples/Stereo/stereo_euroc.cc:136-137)
// Both are HBM/latency-bound gathers with a few dozen operations per element; nothing here is shaped for the matrix cores.
#include "orbhip_internal.h"

// ------------------------------------------------------------------------------------------------ cv::undistortPoints
// OpenCV 3.2 imgproc/undistort.cpp cvUndistortPoints with R = identity, P = K, the 5-coefficient model: everything in double,
// x = (u - cx) * (1/fx), five iterations of  r2 = x^2 + y^2,  icdist = 1 / (1 + ((k3 r2 + k2) r2 + k1) r2),
// dX = 2 p1 x y + p2 (r2 + 2 x^2),  dY = p1 (r2 + 2 y^2) + 2 p2 x y,  x = (x0 - dX) icdist,  then u' = (float)(fx x + cx).
// The operation order is the reference build's (left to right, no contraction: the file is compiled with -ffp-contract=off);
// IEEE double add / mul / div on the device round like the host's, so the result is the same float.
__device__ __forceinline__ void undistort_point(const CameraD& C, float u, float v, float& uo, float& vo)
{
    double x = (double)u, y = (double)v;
    x = (x - C.cx) * C.ifx; y = (y - C.cy) * C.ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1 / (1 + ((C.k3 * r2 + C.k2) * r2 + C.k1) * r2);
        const double deltaX = 2 * C.p1 * x * y + C.p2 * (r2 + 2 * x * x);
        const double deltaY = C.p1 * (r2 + 2 * y * y) + 2 * C.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    uo = (float)(C.fx * x + C.cx);
    vo = (float)(C.fy * y + C.cy);
}

__global__ __launch_bounds__(256) void k_undistort_points(CameraD C, const float* xy, int n, float* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float uo, vo; undistort_point(C, xy[2 * i], xy[2 * i + 1], uo, vo);
    out[2 * i] = uo; out[2 * i + 1] = vo;
}
void orbhip_launch_undistort_points(const CameraD& C, const float* d_xy, int n, float* d_out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, C, d_xy, n, d_out);
}

// Frame::UndistortKeyPoints over camera slots: mvKeysUn[i] = mvKeys[i] with the position replaced (Frame.cc:426-433)
__global__ __launch_bounds__(256) void k_undistort_keys(CameraD C, const orbhip_keypoint* kp, const int* n, orbhip_keypoint* kp_un, int cap)
{
    const int slot = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(n[slot], cap)) return;
    orbhip_keypoint k = kp[(long long)slot * cap + i];
    float uo, vo; undistort_point(C, k.x, k.y, uo, vo);
    k.x = uo; k.y = vo;
    kp_un[(long long)slot * cap + i] = k;
}
void orbhip_launch_undistort_keys(const CameraD& C, const orbhip_keypoint* kp, const int* n, orbhip_keypoint* kp_un, int cap, int nslots, hipStream_t s)
{
    if (nslots > 0 && cap > 0) hipLaunchKernelGGL(k_undistort_keys, dim3((cap + 255) / 256, nslots, 1), dim3(256, 1, 1), 0, s, C, kp, n, kp_un, cap);
}

// ------------------------------------------------------------------------------------------------ Frame::ComputeStereoFromRGBD
// Frame.cc:643-665 over camera slots, with the depth-map conversion of Tracking::GrabImageRGBD (Tracking.cc:226-227: OpenCV's
// cvtScale, dst = (float)src * scale + 0) applied to the one pixel each key point reads instead of to the whole map.
struct RgbdParams {
    const orbhip_keypoint* kp; const orbhip_keypoint* kp_un; const int* n; int cap;
    const uint8_t* depth; long long frame_stride; int row_stride, type, convert; float factor, mbf;
    float* u_right; float* out_depth;
};
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(RgbdParams G)
{
    const int slot = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G.cap) return;
    float ur = -1.0f, dz = -1.0f;                                              // mvuRight = vector<float>(N,-1)  (Frame.cc:645-646)
    if (i < G.n[slot]) {
        const orbhip_keypoint k = G.kp[(long long)slot * G.cap + i];
        const int v = (int)k.y, u = (int)k.x;                                  // imDepth.at<float>(v,u) with float arguments: truncation
        const uint8_t* row = G.depth + (long long)slot * G.frame_stride + (long long)v * G.row_stride;
        float d = G.type == 0 ? ((const float*)row)[u] : (float)((const unsigned short*)row)[u];
        if (G.convert) d = __fadd_rn(__fmul_rn(d, G.factor), 0.0f);
        if (d > 0) { dz = d; ur = __fsub_rn(G.kp_un[(long long)slot * G.cap + i].x, __fdiv_rn(G.mbf, d)); }
    }
    G.u_right[(long long)slot * G.cap + i] = ur; G.out_depth[(long long)slot * G.cap + i] = dz;
}
void orbhip_launch_stereo_from_rgbd(const orbhip_keypoint* kp, const orbhip_keypoint* kp_un, const int* n, int cap, const uint8_t* depth, long long frame_stride,
                                    int row_stride, int type, int convert, float factor, float mbf, float* u_right, float* out_depth, int nslots, hipStream_t s)
{
    RgbdParams G{kp, kp_un, n, cap, depth, frame_stride, row_stride, type, convert, factor, mbf, u_right, out_depth};
    if (nslots > 0 && cap > 0) hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((cap + 255) / 256, nslots, 1), dim3(256, 1, 1), 0, s, G);
}

// ------------------------------------------------------------------------------------------------ cv::remap, INTER_LINEAR, 8UC1
// OpenCV 3.2 imgproc/imgwarp.cpp, CV_32FC1 map pair: sx = cvRound(map_x * 32), sy = cvRound(map_y * 32) (round-half-even; the
// product by 32 is exact), integer parts sx >> 5 / sy >> 5 saturated to short, 5-bit fractions a, b; the four taps are weighted
// with the fixed-point table BilinearTab_i = 32768 (1 - b/32)(1 - a/32) ... — integers 32 (32-b)(32-a) etc. — and
// dst = (sum + 16384) >> 15 = ((32-b) ((32-a) p00 + a p01) + b ((32-a) p10 + a p11) + 512) >> 10.
// BORDER_CONSTANT with value 0: taps outside the source read 0 (a window entirely outside gives 0).
//
// The maps are constant for a camera, so the float -> fixed-point step (OpenCV redoes it for every image) is done ONCE when the
// maps are attached (orbhip_set_rectification, host, same cvRound): the kernel reads qx = cvRound(32 map_x), qy likewise.
// Four destination pixels per lane: two 128-bit table loads, then per pixel two unaligned 16-bit gathers (the horizontal tap pair
// of each source row; neighbouring lanes' taps share cache lines because rectification maps are smooth), one v_dot4 per row with
// the packed weights (32-a, a), a two-term vertical blend; one 32-bit store.  ~18 VALU operations per pixel: the kernel is bound
// by VALU issue and the texture-address path, not by HBM (2 B per pixel + the 8 B/pixel table, which stays in L2/MALL across the
// frames of a batch).  A wave in which any window touches the source border takes the per-tap checked path.
struct RemapParams {
    const uint8_t* src; long long src_frame_stride; int src_row_stride, src_w, src_h;
    const int* qx; const int* qy; int q_pitch;          // table rows are q_pitch ints apart (a multiple of 4: aligned 128-bit loads)
    uint8_t* dst; long long dst_frame_stride; int dst_pitch, w, h;
};
typedef unsigned short u16_unaligned __attribute__((aligned(1)));
struct alignas(16) Q4 { int x, y, z, w; };                                     // four table entries: one 128-bit load
__device__ __forceinline__ unsigned remap_pixel_checked(const uint8_t* src, int pitch, int sw, int sh, int sxq, int syq)
{
    const int sx = min(max(sxq >> 5, -32768), 32767), sy = min(max(syq >> 5, -32768), 32767);        // saturate_cast<short>
    const int a = sxq & 31, b = syq & 31;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0u;
    const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw, y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
    const long long o = (long long)sy * pitch + sx;
    const int p00 = (x0 && y0) ? src[o] : 0, p01 = (x1 && y0) ? src[o + 1] : 0, p10 = (x0 && y1) ? src[o + pitch] : 0, p11 = (x1 && y1) ? src[o + pitch + 1] : 0;
    const int top = (32 - a) * p00 + a * p01, bot = (32 - a) * p10 + a * p11;
    return (unsigned)(((32 - b) * top + b * bot + 512) >> 10);
}
__device__ __forceinline__ unsigned remap_pixel_inside(const uint8_t* src, unsigned pitch, int sxq, int syq)
{   // all four taps inside the source: (sx, sy) in [0, sw-2] x [0, sh-2]
    const unsigned a = (unsigned)sxq & 31u, b = (unsigned)syq & 31u;
    const unsigned o = (unsigned)(syq >> 5) * pitch + (unsigned)(sxq >> 5);
    const unsigned r0 = *reinterpret_cast<const u16_unaligned*>(src + o), r1 = *reinterpret_cast<const u16_unaligned*>(src + (o + pitch));
    const unsigned wa = a * 255u + 32u;                                        // (32 - a) | a << 8
    const unsigned top = __builtin_amdgcn_udot4(r0, wa, 0u, false), bot = __builtin_amdgcn_udot4(r1, wa, 0u, false);
    return (top * 32u + b * (bot - top) + 512u) >> 10;                         // (32-b) top + b bot + 512; the difference wraps mod 2^32 harmlessly
}
#define RM_ROWS 4                       // destination rows per workgroup: 4 x 4 pixels per lane, all table loads and gathers of a lane in flight together
__global__ __launch_bounds__(256) void k_remap(RemapParams R)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * RM_ROWS, f = blockIdx.z;
    const bool active = x < R.w;
    const uint8_t* src = R.src + (long long)f * R.src_frame_stride;
    uint8_t* dst = R.dst + (long long)f * R.dst_frame_stride + x;
    const bool whole = active && x + 4 <= R.w;
    const unsigned mw = (unsigned)(R.src_w - 1) << 5, mh = (unsigned)(R.src_h - 1) << 5;      // q < (sw-1)*32  <=>  0 <= q >> 5 <= sw-2
    Q4 vx[RM_ROWS], vy[RM_ROWS];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < RM_ROWS; r++) {
        const int y = min(y0 + r, R.h - 1);                                    // rows past the image repeat the last row and are not stored
        const long long mo = (long long)y * R.q_pitch + (active ? x : 0);
        vx[r] = Q4{0, 0, 0, 0}; vy[r] = Q4{0, 0, 0, 0};
        if (whole) { vx[r] = *(const Q4*)(R.qx + mo); vy[r] = *(const Q4*)(R.qy + mo); }
    }
#pragma unroll
    for (int r = 0; r < RM_ROWS; r++)
        inside = inside && (unsigned)vx[r].x < mw && (unsigned)vx[r].y < mw && (unsigned)vx[r].z < mw && (unsigned)vx[r].w < mw &&
                 (unsigned)vy[r].x < mh && (unsigned)vy[r].y < mh && (unsigned)vy[r].z < mh && (unsigned)vy[r].w < mh;
    if (__all(!active || (whole && inside))) {                                // wave-uniform
        if (!active) return;
        unsigned g[RM_ROWS];
#pragma unroll
        for (int r = 0; r < RM_ROWS; r++) {
            const unsigned g0 = remap_pixel_inside(src, (unsigned)R.src_row_stride, vx[r].x, vy[r].x), g1 = remap_pixel_inside(src, (unsigned)R.src_row_stride, vx[r].y, vy[r].y);
            const unsigned g2 = remap_pixel_inside(src, (unsigned)R.src_row_stride, vx[r].z, vy[r].z), g3 = remap_pixel_inside(src, (unsigned)R.src_row_stride, vx[r].w, vy[r].w);
            g[r] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        }
#pragma unroll
        for (int r = 0; r < RM_ROWS; r++)
            if (y0 + r < R.h) *(uint32_t*)(dst + (long long)(y0 + r) * R.dst_pitch) = g[r];      // dst_pitch is a multiple of 4
        return;
    }
    if (!active) return;
    const int n = min(4, R.w - x);
    for (int r = 0; r < RM_ROWS && y0 + r < R.h; r++) {
        const int* qx = R.qx + (long long)(y0 + r) * R.q_pitch + x;
        const int* qy = R.qy + (long long)(y0 + r) * R.q_pitch + x;
        uint8_t* drow = dst + (long long)(y0 + r) * R.dst_pitch;
        unsigned g[4] = {0u, 0u, 0u, 0u};
        for (int i = 0; i < n; i++) g[i] = remap_pixel_checked(src, R.src_row_stride, R.src_w, R.src_h, qx[i], qy[i]);
        if (n == 4) *(uint32_t*)drow = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
        else for (int i = 0; i < n; i++) drow[i] = (uint8_t)g[i];
    }
}
void orbhip_launch_remap(const uint8_t* src, long long src_frame_stride, int src_row_stride, int src_w, int src_h, const int* qx, const int* qy, int q_pitch,
                         uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h, int nframes, hipStream_t s)
{
    RemapParams R{src, src_frame_stride, src_row_stride, src_w, src_h, qx, qy, q_pitch, dst, dst_frame_stride, dst_pitch, w, h};
    hipLaunchKernelGGL(k_remap, dim3((w + 1023) / 1024, (h + RM_ROWS - 1) / RM_ROWS, nframes), dim3(256, 1, 1), 0, s, R);
}

// ------------------------------------------------------------------------------------------------ row re-pitch
// Host images arrive densely packed (cv::Mat rows `stride` bytes apart, usually stride == width) and are DMA'd as they are — one
// linear copy per image, or per run of adjacent images; a strided host-to-device copy with an odd width falls off the DMA engines'
// fast path by two orders of magnitude.  This kernel spreads the rows to the 64-byte-multiple pitch the pipeline's aligned 32-bit
// loads want: 4 bytes per thread, unaligned load, aligned store (pad bytes past the row end are never read).
typedef unsigned u32_any_align __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void k_repitch(const uint8_t* src, long long src_frame_stride, int src_row_stride, uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* s = src + (long long)f * src_frame_stride + (long long)y * src_row_stride + x;
    unsigned v;
    if (x + 4 <= w || y + 1 < h) v = *reinterpret_cast<const u32_any_align*>(s);         // reading into the next row is inside the frame
    else { v = 0; for (int k = 0; k < w - x; k++) v |= (unsigned)s[k] << (8 * k); }       // the last bytes of the last row
    *reinterpret_cast<unsigned*>(dst + (long long)f * dst_frame_stride + (long long)y * dst_pitch + x) = v;
}
void orbhip_launch_repitch(const uint8_t* src, long long src_frame_stride, int src_row_stride, uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h, int nframes, hipStream_t s)
{
    if (nframes > 0) hipLaunchKernelGGL(k_repitch, dim3((w + 1023) / 1024, h, nframes), dim3(256, 1, 1), 0, s, src, src_frame_stride, src_row_stride, dst, dst_frame_stride, dst_pitch, w, h);
}

// ------------------------------------------------------------------------------------------------ small copies as kernels
// A hipMemcpyAsync between two kernels of a stream runs on a DMA engine: the queue hands over to the engine and back, and each hand-over idles
// the chain for 8-9 us (profiles/r04_single_frame_call_timeline.txt: H2D end -> first kernel 8.2 us, last kernel -> D2H start 8.9 us) - on a
// single-image call that is 17 of 160 us, on a matcher call more than its kernels.  Pinned host memory is addressable from the device, so a copy
// of a few hundred KB is a kernel like any other: 16 bytes per lane over PCIe, no hand-over.  Bulk transfers stay on the DMA engines (they overlap
// with compute and do not occupy CUs).
__global__ __launch_bounds__(256) void k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, long long n16, int tail)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (tail && i == 0) {
        uint8_t* d = reinterpret_cast<uint8_t*>(dst + n16); const uint8_t* sp = reinterpret_cast<const uint8_t*>(src + n16);
        for (int k = 0; k < tail; k++) d[k] = sp[k];
    }
}
hipError_t orbhip_copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    static const bool dma_only = [] { const char* e = getenv("ORBHIP_COPY_KERNELS"); return e && *e == '0'; }();      // ORBHIP_COPY_KERNELS=0: every copy on the DMA engines
    if (bytes == 0) return hipSuccess;
    if (dma_only || bytes > ((size_t)2 << 20) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return hipMemcpyAsync(dst, src, bytes, kind, s);
    const long long n16 = (long long)(bytes >> 4);
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::max<long long>((n16 + 255) / 256, 1), 1, 1), dim3(256, 1, 1), 0, s, (uint4*)dst, (const uint4*)src, n16, (int)(bytes & 15));
    return hipGetLastError();
}
