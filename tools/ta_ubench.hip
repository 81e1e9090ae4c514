// ta_ubench.hip — what a vector-memory instruction costs in the texture addresser (TA) of one CU, by access shape.  Every wave issues
// NI loads of one shape from an L2-resident buffer (different rows per instruction); all CUs busy, 8 waves per SIMD; the figure printed
// is cycles per instruction per CU at 2.4 GHz (= elapsed * clock * CUs / instructions).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32u __attribute__((aligned(1)));
typedef unsigned v4u __attribute__((vector_size(16), aligned(1)));
typedef unsigned v2u __attribute__((vector_size(8), aligned(1)));
#define NI 64
#define PITCH 1280
template <int SHAPE> __global__ __launch_bounds__(256) void k(const uint8_t* buf, unsigned* out, int mis)
{
    __shared__ unsigned lds[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* base = buf + ((size_t)(blockIdx.x * 4 + wave) * 37 % 20000) * PITCH + mis;     // a window of rows somewhere in the buffer
    unsigned acc = 0;
#pragma unroll 8
    for (int n = 0; n < NI; n++) {
        const uint8_t* b = base + (size_t)(n * 3) * PITCH;
        if (SHAPE == 0) acc += *(const u32u*)(b + 4 * lane);                                            // 1 row x 256 B
        if (SHAPE == 1) acc += *(const u32u*)(b + (lane >> 3) * PITCH + 4 * (lane & 7));              // 8 rows x 32 B
        if (SHAPE == 2) { v4u v = *(const v4u*)(b + (lane & 31) * PITCH + 16 * (lane >> 5)); acc += v[0] + v[1] + v[2] + v[3]; }   // 32 rows x 32 B, 16 B per lane
        if (SHAPE == 3) acc += *(const u32u*)(b + (lane >> 1) * PITCH + 4 * (lane & 1));              // 32 rows x 8 B
        if (SHAPE == 4) { v4u v = *(const v4u*)(b + 16 * lane); acc += v[0] + v[1] + v[2] + v[3]; }   // 1 KB contiguous
        if (SHAPE == 5) acc += b[(lane >> 5) * PITCH + (lane & 31)];                                  // 2 rows x 32 bytes (byte loads)
        if (SHAPE == 6) { const int p = lane; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + (p / 10) * PITCH + 4 * (p % 10)), (__attribute__((address_space(3))) void*)(lds[wave] + 64 * (n & 7)), 4, 0, 0); }   // LDS-DMA 6.4 rows x 40 B
        if (SHAPE == 7) { v2u v = *(const v2u*)(b + (lane >> 2) * PITCH + 8 * (lane & 3)); acc += v[0] + v[1]; }      // 16 rows x 32 B, 8 B per lane
        if (SHAPE == 8) { const int p = lane; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + 4 * p), (__attribute__((address_space(3))) void*)(lds[wave] + 64 * (n & 7)), 4, 0, 0); }   // LDS-DMA 1 row x 256 B
        if (SHAPE == 9) acc += *(const u32u*)(b + (lane >> 4) * PITCH + 4 * min(lane & 15, 8));               // 4 rows x 36 B, lanes 9..15 of a row repeat its dword 8 (k_describe's patch load)
        if (SHAPE == 10) { const int r = min(lane / 9, 6); acc += *(const u32u*)(b + r * PITCH + 4 * min(lane - 9 * r, 8)); }   // 7 rows x 36 B, 9 lanes per row
        if (SHAPE == 15) acc += *(const u32u*)(b + (lane >> 4) * PITCH + 4 * (lane & 15));                    // 4 rows x 64 B
        if (SHAPE == 11 || SHAPE == 12 || SHAPE == 13) {
            const int dpr = SHAPE == 11 ? 12 : SHAPE == 12 ? 16 : 8, p = lane;                                  // LDS-DMA, rows of 48 / 64 / 32 bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + (p / dpr) * PITCH + 4 * (p % dpr)), (__attribute__((address_space(3))) void*)(lds[wave] + 64 * (n & 7)), 4, 0, 0);
        }
    }
    if (SHAPE == 6 || SHAPE == 8 || SHAPE == 11 || SHAPE == 12 || SHAPE == 13) { __builtin_amdgcn_s_waitcnt(0x0f70); acc += lds[wave][lane]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int SHAPE> static void run(const char* name, const uint8_t* buf, unsigned* out)
{
    const int blocks = 256 * 8 * 4;       // 8 workgroups per CU -> 8 waves per SIMD
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int mis = 0; mis < 2; mis++) {
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, mis);
        (void)hipEventRecord(a, 0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, mis);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
        const double instr = (double)blocks * 4 * NI;
        printf("%-44s %s  %7.3f ms  %6.1f cycles / instruction / CU\n", name, mis ? "+1 byte " : "aligned ", ms, ms * 1e-3 * 2.4e9 * 256 / instr);
    }
}
int main()
{
    uint8_t* buf; unsigned* out; (void)hipMalloc(&buf, (size_t)PITCH * 20400); (void)hipMemset(buf, 1, (size_t)PITCH * 20400); (void)hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    run<0>("dword, 1 row x 256 B", buf, out);
    run<4>("dwordx4, 1 KB contiguous", buf, out);
    run<1>("dword, 8 rows x 32 B", buf, out);
    run<7>("dwordx2, 16 rows x 32 B", buf, out);
    run<2>("dwordx4, 32 rows x 32 B", buf, out);
    run<3>("dword, 32 rows x 8 B", buf, out);
    run<5>("ubyte, 2 rows x 32 B", buf, out);
    run<8>("LDS-DMA dword, 1 row x 256 B", buf, out);
    run<6>("LDS-DMA dword, 6.4 rows x 40 B", buf, out);
    run<11>("LDS-DMA dword, 5.33 rows x 48 B", buf, out);
    run<12>("LDS-DMA dword, 4 rows x 64 B", buf, out);
    run<13>("LDS-DMA dword, 8 rows x 32 B", buf, out);
    run<9>("dword, 4 rows x 36 B (7 lanes repeat)", buf, out);
    run<15>("dword, 4 rows x 64 B", buf, out);
    run<10>("dword, 7 rows x 36 B (9 lanes per row)", buf, out);
    return 0;
}
