// lds_dma_b128_probe.hip — gfx950's 16-byte LDS-DMA (global_load_lds_dwordx4): does it take global addresses at any byte alignment, where do the lanes'
// 16 bytes land (lane l -> LDS bytes [16 l, 16 l + 16) behind the M0 base?), and what does an UNALIGNED global_load_dwordx4 to registers return?
// Also times both against their 4-byte forms on a 2-D gather of the shape k_describe makes (rows of an image, 48 / 32 bytes per row).
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_b128_probe.hip -o tools/lds_dma_b128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
__global__ void k_probe(const unsigned char* src, int misalign, int lane_stride, unsigned* out, uint4* out2)
{
    __shared__ unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned char* g = src + misalign + lane * lane_stride;
    __builtin_amdgcn_global_load_lds((gptr)g, (lptr)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr)(g + 4096), (lptr)(lds + 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
    typedef uint4 u4_unaligned __attribute__((aligned(1)));
    out2[lane] = *reinterpret_cast<const u4_unaligned*>(g);
}
// timing: every wave gathers `rows` rows of `bpr` bytes from an image of pitch `pitch` around a pseudo-random centre, `iters` times
template <int MODE> __global__ void k_time(const unsigned char* img, int pitch, int h, int iters, unsigned* sink)
{
    __shared__ unsigned lds[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    unsigned s = blockIdx.x * 2654435761u + wave * 40503u + 12345u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        const int cx = 32 + (s >> 8) % (pitch - 128), cy = 32 + (s >> 20) % (h - 96);
        const unsigned char* base = img + (size_t)cy * pitch + (cx & ~3);
        if (MODE == 0) {            // window: 7 passes of dword LDS-DMA (37 rows x 12 dwords)
#pragma unroll
            for (int k = 0; k < 7; k++) { const int pos = 64 * k + lane, row = min(pos / 12, 36), col = 4 * (pos - 12 * (pos / 12)); __builtin_amdgcn_global_load_lds((gptr)(base + row * pitch + col), (lptr)(lds[wave] + 64 * k), 4, 0, 0); }
        } else if (MODE == 1) {     // window: 2 passes of 16-byte LDS-DMA (37 rows x 3 x 16 B)
#pragma unroll
            for (int k = 0; k < 2; k++) { const int pos = 64 * k + lane, row = min(pos / 3, 36), col = 16 * (pos - 3 * (pos / 3)); __builtin_amdgcn_global_load_lds((gptr)(base + row * pitch + col), (lptr)(lds[wave] + 256 * k), 16, 0, 0); }
        } else if (MODE == 2) {     // patch: 8 passes of aligned dword loads (4 rows x 16 lanes, 9 used)
#pragma unroll
            for (int q = 0; q < 8; q++) acc += *reinterpret_cast<const unsigned*>(base + ((lane >> 4) + 4 * q) * pitch + 4 * min(lane & 15, 8));
        } else {                    // patch: ONE unaligned 16-byte load per lane, lane = (row, half)
            typedef uint4 u4_unaligned __attribute__((aligned(1)));
            const uint4 v = *reinterpret_cast<const u4_unaligned*>(img + (size_t)(cy + min(lane >> 1, 30)) * pitch + cx + 16 * (lane & 1));
            acc += v.x + v.y + v.z + v.w;
        }
        if (MODE < 2) { __builtin_amdgcn_s_waitcnt(0x0f70); acc += lds[wave][lane]; }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    std::vector<unsigned char> h(16384); for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 37 + (i >> 8) * 11 + 5);
    unsigned char* d; unsigned* o; uint4* o2; hipMalloc(&d, h.size()); hipMalloc(&o, 512 * 4); hipMalloc(&o2, 64 * 16); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int stride : {16, 19, 48, 1}) for (int mis = 0; mis < 4; mis++) {
        hipMemset(o, 0, 2048);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, mis, stride, o, o2);
        unsigned r[512]; unsigned char r2[1024]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 1024, hipMemcpyDeviceToHost);
        int bad = 0, bad2 = 0;
        for (int l = 0; l < 64; l++) for (int p = 0; p < 2; p++) if (memcmp(&r[256 * p + 4 * l], &h[mis + l * stride + 4096 * p], 16)) bad++;
        for (int l = 0; l < 64; l++) if (memcmp(&r2[16 * l], &h[mis + l * stride], 16)) bad2++;
        printf("lane stride %2d misalign %d: LDS-DMA x4 %s (%d wrong of 128), unaligned dwordx4 load %s (%d wrong of 64)\n", stride, mis, bad ? "MISMATCH" : "ok", bad, bad2 ? "MISMATCH" : "ok", bad2);
        bad_total += bad + bad2;
    }
    printf(bad_total ? "b128: NOT usable as assumed\n" : "b128: any byte alignment works; lane l -> LDS bytes [16 l, 16 l + 16)\n");
    const int W = 1280, H = 384, pitch = 1280;
    unsigned char* img; unsigned* sink; hipMalloc(&img, (size_t)pitch * H); hipMemset(img, 7, (size_t)pitch * H); hipMalloc(&sink, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"window, 7 x dword LDS-DMA", "window, 2 x 16-byte LDS-DMA", "patch, 8 x aligned dword loads", "patch, 1 x unaligned 16-byte load"};
    for (int mode = 0; mode < 4; mode++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(4096), dim3(256), 0, 0, img, pitch, H, 200, sink);
        if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(4096), dim3(256), 0, 0, img, pitch, H, 200, sink);
        if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(4096), dim3(256), 0, 0, img, pitch, H, 200, sink);
        if (mode == 3) hipLaunchKernelGGL(k_time<3>, dim3(4096), dim3(256), 0, 0, img, pitch, H, 200, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-36s %.3f ms for %d gathers = %.2f ns per gather per CU-slot (%.1f cycles of a CU at 2.2 GHz)\n", names[mode], ms, 4096 * 4 * 200, ms * 1e6 / (4096.0 * 4 * 200) * 256, ms * 1e6 / (4096.0 * 4 * 200) * 256 * 2.2);
    }
    (void)W;
    return 0;
}
