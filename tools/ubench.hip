// Instruction-rate micro-benchmark (gfx950): dependent-free chains of one op type per kernel, 256 CUs x 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short pku16 __attribute__((vector_size(4)));
#define N_IT 4096
template <int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i;
    unsigned b = seed ^ 0x9e3779b9u;
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) { pku16 x = (pku16)a[i], y = (pku16)b; a[i] = (unsigned)(x < y ? x : y) + 1; }                     // v_pk_min_u16 (+ add to defeat idempotence)
            if (OP == 1) a[i] = min(a[i], b) + 1;                                                                              // v_min_u32 + v_add
            if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c010c00u) + 1;                                               // v_perm_b32 + add
            if (OP == 3) a[i] = a[i] * 3 + b;                                                                                  // v_mad_u32_u24-ish / mul+add
            if (OP == 4) a[i] = a[i] + b;                                                                                      // v_add_u32
            if (OP == 5) { pku16 x = (pku16)a[i], y = (pku16)b; a[i] = (unsigned)(x + y); }                                   // v_pk_add_u16
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int ops_per_inner)
{
    unsigned* d; hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(8192), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(8192), dim3(256), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = 8192.0 * 4 * N_IT * 8 * ops_per_inner;          // wave-instructions
    printf("%-28s %8.3f ms  %.2f G wave-inst/s  -> %.2f cycles/inst/SIMD @2.4GHz (1024 SIMDs)\n", name, ms, winst / ms / 1e6, 1024 * 2.4e9 / (winst / (ms * 1e-3)));
    hipFree(d);
}
int main()
{
    run<4>("v_add_u32", 1); run<1>("v_min_u32 + v_add", 2); run<0>("v_pk_min_u16 + v_add", 2); run<5>("v_pk_add_u16", 1);
    run<2>("v_perm_b32 + v_add", 2); run<3>("mul + add", 2);
    return 0;
}
