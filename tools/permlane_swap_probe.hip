// Probe of gfx950's v_permlane16_swap / v_permlane32_swap and of the DPP controls k_describe's moment reduction uses: prints, per lane, where each
// result came from (values are lane ids + 100 * register).  Build: hipcc --offload-arch=gfx950 tools/permlane_swap_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    const unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r32[0]; out[64 + l] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = r16[0]; out[192 + l] = r16[1];
    out[256 + l] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)l, 0x128, 0xf, 0xf, true);   // row_ror:8
    out[320 + l] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)l, 0x141, 0xf, 0xf, true);   // row_half_mirror
    out[384 + l] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)l, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    out[448 + l] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)l, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
}
int main()
{
    unsigned* d; hipMalloc(&d, 512 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[8] = {"permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]", "row_ror:8", "row_half_mirror", "quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]"};
    for (int r = 0; r < 8; r++) { printf("%s:", names[r]); for (int l = 0; l < 64; l++) printf(" %u", h[64 * r + l]); printf("\n"); }
    return 0;
}
