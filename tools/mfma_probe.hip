// mfma_probe.hip — checks the operand / result layout k_blur_mfma assumes for v_mfma_i32_32x32x32_i8 on gfx950:
//   A[i][k]: lane i + 32h, byte b of its 16 bytes  <->  k = 16h + b;   B[k][j]: lane j + 32h, byte b  <->  the same k
//   D[i][j]: lane (j + 32 * ((i >> 2) & 1)), register (i & 3) + 4 * (i >> 3)
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((vector_size(16)));
typedef int v16i __attribute__((vector_size(64)));
__global__ void k_probe(const v4i* a, const v4i* b, v16i* out)
{
    v16i c; for (int i = 0; i < 16; i++) c[i] = 1000 + i;
    out[threadIdx.x] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}
int main()
{
    signed char A[32][32], B[32][32];
    unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 24) & 0xff) - 128; };
    for (int i = 0; i < 32; i++) for (int k = 0; k < 32; k++) { A[i][k] = (signed char)rnd(); B[i][k] = (signed char)rnd(); }
    signed char ha[64][16], hb[64][16];
    for (int l = 0; l < 64; l++) for (int q = 0; q < 16; q++) { ha[l][q] = A[l & 31][16 * (l >> 5) + q]; hb[l][q] = B[16 * (l >> 5) + q][l & 31]; }
    v4i *da, *db; v16i* dout; (void)hipMalloc(&da, 1024); (void)hipMalloc(&db, 1024); (void)hipMalloc(&dout, 64 * 64);
    (void)hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dout);
    int r[64][16]; (void)hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
        int ref = 0; for (int k = 0; k < 32; k++) ref += (int)A[i][k] * (int)B[k][j];
        const int lane = j + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3);
        if (r[lane][reg] != ref + 1000 + reg) bad++;
    }
    printf(bad ? "MFMA i8 32x32x32 layout: MISMATCH (%d of 1024)\n" : "MFMA i8 32x32x32 layout: as assumed (%d wrong)\n", bad);
    return bad != 0;
}
