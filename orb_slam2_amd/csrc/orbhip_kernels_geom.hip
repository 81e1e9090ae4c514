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

// ------------------------------------------------------------------------------------------------ cv::remap, INTER_LINEAR, 8UC1
// OpenCV 3.2 imgproc/imgwarp.cpp, CV_32FC1 map pair: sx = cvRound(map_x * 32), sy = cvRound(map_y * 32) (round-half-even; the
// product by 32 is exact), integer parts sx >> 5 / sy >> 5 saturated to short, 5-bit fractions a, b; the four taps are weighted
// with the fixed-point table BilinearTab_i = 32768 (1 - b/32)(1 - a/32) ... — integers 32 (32-b)(32-a) etc. — and
// dst = (sum + 16384) >> 15 = ((32-b) ((32-a) p00 + a p01) + b ((32-a) p10 + a p11) + 512) >> 10.
// BORDER_CONSTANT with value 0: taps outside the source read 0 (a window entirely outside gives 0).
// Four destination pixels per thread: two 128-bit map loads, 16 byte gathers (the maps are smooth, so the taps of neighbouring
// lanes share cache lines), one 32-bit store.  Per frame it moves 8 B of maps (L2-resident across the frames of a batch)
// + ~1 B read + 1 B written per pixel.
struct RemapParams {
    const uint8_t* src; long long src_frame_stride; int src_row_stride, src_w, src_h;
    const float* map_x; const float* map_y;
    uint8_t* dst; long long dst_frame_stride; int dst_pitch, w, h;
};
__device__ __forceinline__ unsigned remap_pixel(const uint8_t* src, int pitch, int sw, int sh, float mx, float my)
{
    const int sxq = (int)rintf(__fmul_rn(mx, 32.0f)), syq = (int)rintf(__fmul_rn(my, 32.0f));
    const int sx = min(max(sxq >> 5, -32768), 32767), sy = min(max(syq >> 5, -32768), 32767);
    const int a = sxq & 31, b = syq & 31;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0u;
    const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw, y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
    const uint8_t* r0 = src + (long long)sy * pitch + sx;
    const uint8_t* r1 = r0 + pitch;
    const int p00 = (x0 && y0) ? r0[0] : 0, p01 = (x1 && y0) ? r0[1] : 0, p10 = (x0 && y1) ? r1[0] : 0, p11 = (x1 && y1) ? r1[1] : 0;
    const int top = (32 - a) * p00 + a * p01, bot = (32 - a) * p10 + a * p11;
    return (unsigned)(((32 - b) * top + b * bot + 512) >> 10);
}
__global__ __launch_bounds__(256) void k_remap(RemapParams R)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
    if (x >= R.w) return;
    const uint8_t* src = R.src + (long long)f * R.src_frame_stride;
    const float* mx = R.map_x + (long long)y * R.w + x;
    const float* my = R.map_y + (long long)y * R.w + x;
    uint8_t* drow = R.dst + (long long)f * R.dst_frame_stride + (long long)y * R.dst_pitch + x;
    if (x + 4 <= R.w && ((((unsigned long long)mx) | ((unsigned long long)my)) & 15ull) == 0) {
        const float4 vx = *(const float4*)mx, vy = *(const float4*)my;
        const unsigned g0 = remap_pixel(src, R.src_row_stride, R.src_w, R.src_h, vx.x, vy.x), g1 = remap_pixel(src, R.src_row_stride, R.src_w, R.src_h, vx.y, vy.y);
        const unsigned g2 = remap_pixel(src, R.src_row_stride, R.src_w, R.src_h, vx.z, vy.z), g3 = remap_pixel(src, R.src_row_stride, R.src_w, R.src_h, vx.w, vy.w);
        *(uint32_t*)drow = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);          // dst_pitch is a multiple of 4
    } else {
        const int n = min(4, R.w - x);
        for (int i = 0; i < n; i++) drow[i] = (uint8_t)remap_pixel(src, R.src_row_stride, R.src_w, R.src_h, mx[i], my[i]);
    }
}
void orbhip_launch_remap(const uint8_t* src, long long src_frame_stride, int src_row_stride, int src_w, int src_h, const float* map_x, const float* map_y,
                         uint8_t* dst, long long dst_frame_stride, int dst_pitch, int w, int h, int nframes, hipStream_t s)
{
    RemapParams R{src, src_frame_stride, src_row_stride, src_w, src_h, map_x, map_y, dst, dst_frame_stride, dst_pitch, w, h};
    hipLaunchKernelGGL(k_remap, dim3((w + 1023) / 1024, h, nframes), dim3(256, 1, 1), 0, s, R);
}
