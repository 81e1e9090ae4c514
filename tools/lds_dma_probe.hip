// lds_dma_probe.hip — does global_load_lds_dword (LDS-DMA) accept global addresses at any byte alignment on gfx950, and in which
// LDS order do the lanes' dwords land?  Build: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o tools/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__global__ void k_probe(const unsigned char* src, int misalign, int lane_stride, unsigned* out)
{
    __shared__ unsigned lds[256];
    const int lane = threadIdx.x;
    lds[lane] = 0xdeadbeefu; lds[lane + 64] = 0xdeadbeefu;
    __syncthreads();
    const unsigned char* g = src + misalign + lane * lane_stride;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 1024), (__attribute__((address_space(3))) void*)(lds + 64), 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    __syncthreads();
    out[lane] = lds[lane]; out[lane + 64] = lds[lane + 64];
}
int main()
{
    std::vector<unsigned char> h(8192); for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 37 + (i >> 8) * 11 + 5);
    unsigned char* d; unsigned* o; hipMalloc(&d, h.size()); hipMalloc(&o, 128 * 4); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int stride : {4, 7, 44}) for (int mis = 0; mis < 4; mis++) {
        hipMemset(o, 0, 512);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, mis, stride, o);
        unsigned r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int p = 0; p < 2; p++) { unsigned e; memcpy(&e, &h[mis + l * stride + 1024 * p], 4); if (r[l + 64 * p] != e) bad++; }
        printf("lane stride %2d misalign %d: %s (%d wrong of 128)\n", stride, mis, bad ? "MISMATCH" : "ok", bad); bad_total += bad;
    }
    printf(bad_total ? "LDS-DMA: unaligned sources NOT usable\n" : "LDS-DMA: any byte alignment works, lane l -> LDS dword l\n");
    return 0;
}
