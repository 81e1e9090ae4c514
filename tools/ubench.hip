// Instruction-rate micro-benchmark (gfx950): 8 independent chains per lane of one instruction (inline asm), 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
#define DEFK(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) { \
    unsigned a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned b = seed ^ 0x3c003c00u, c = seed | 0x11u; \
    for (int it = 0; it < N_IT; it++) { \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }
#define A_ADD(i)      "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MINU(i)     "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_PKMINU16(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PKMINF16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define A_PKMAXI16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_MINF32(i)   "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_FMAF32(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MULF32(i)   "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_MIN3U(i)    "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3F(i)    "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MAD24(i)    "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTF(i)     "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define A_RNDNE(i)    "v_rndne_f32 %" #i ", %" #i "\n"
#define A_CVTI(i)     "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_MULLO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n"
#define A_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define A_PKMULF16(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define A_BCNT(i)     "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_XOR(i)      "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_SADU8(i)    "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_CMPSEL(i)   "v_cmp_lt_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_DPP(i)      "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_PKMADU16(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define A_PKMULU16(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define A_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_DOT4U8(i)   "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_ADDF32(i)   "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_CVTUB1(i)   "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define A_FLOOR(i)    "v_floor_f32 %" #i ", %" #i "\n"
#define A_CVTPKU8(i)  "v_cvt_pk_u8_f32 %" #i ", %" #i ", 1, %8\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define A_ADDSDWA(i)  "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define A_CVTU32F(i)  "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define A_BPERM(i)    "ds_bpermute_b32 %" #i ", %8, %" #i "\n s_waitcnt lgkmcnt(0)\n"
DEFK(k_add, A_ADD) DEFK(k_minu, A_MINU) DEFK(k_pkminu16, A_PKMINU16) DEFK(k_pkminf16, A_PKMINF16) DEFK(k_pkmaxi16, A_PKMAXI16)
DEFK(k_minf32, A_MINF32) DEFK(k_fmaf32, A_FMAF32) DEFK(k_mulf32, A_MULF32) DEFK(k_min3u, A_MIN3U) DEFK(k_max3f, A_MAX3F)
DEFK(k_perm, A_PERM) DEFK(k_mad24, A_MAD24) DEFK(k_lshladd, A_LSHLADD) DEFK(k_andor, A_ANDOR) DEFK(k_cvtf, A_CVTF) DEFK(k_rndne, A_RNDNE)
DEFK(k_cvti, A_CVTI) DEFK(k_mullo, A_MULLO) DEFK(k_bfe, A_BFE) DEFK(k_alignbit, A_ALIGNBIT) DEFK(k_pkaddf16, A_PKADDF16) DEFK(k_pkmulf16, A_PKMULF16)
DEFK(k_pkmadu16, A_PKMADU16) DEFK(k_pkmulu16, A_PKMULU16) DEFK(k_pkaddu16, A_PKADDU16) DEFK(k_dot4u8, A_DOT4U8) DEFK(k_addf32, A_ADDF32)
DEFK(k_cvtub1, A_CVTUB1) DEFK(k_floor, A_FLOOR) DEFK(k_cvtpku8, A_CVTPKU8) DEFK(k_alignbyte, A_ALIGNBYTE) DEFK(k_addsdwa, A_ADDSDWA) DEFK(k_cvtu32f, A_CVTU32F)
DEFK(k_bcnt, A_BCNT) DEFK(k_xor, A_XOR) DEFK(k_sadu8, A_SADU8) DEFK(k_cmpsel, A_CMPSEL) DEFK(k_dpp, A_DPP) DEFK(k_bperm, A_BPERM)
__global__ void k_probe_cvtpk(const float* in, unsigned* out)
{
    if (threadIdx.x < 12) out[threadIdx.x] = __builtin_amdgcn_cvt_pk_u8_f32(in[threadIdx.x], 1u, 0xAA0000BBu);
}
typedef void (*kfn)(unsigned*, unsigned);
static void run(const char* name, kfn f, int inst_per_slot)
{
    unsigned* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(8192), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(f, dim3(8192), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winst = 8192.0 * 4 * N_IT * 8 * inst_per_slot;
    printf("%-16s %8.3f ms  %6.2f cycles/inst/SIMD @2.4GHz\n", name, ms, 1024 * 2.4e9 / (winst / (ms * 1e-3)));
    (void)hipFree(d);
}
int main()
{
#define R(n, c) run(#n, n, c)
    R(k_add, 1); R(k_xor, 1); R(k_minu, 1); R(k_min3u, 1); R(k_pkminu16, 1); R(k_pkmaxi16, 1); R(k_pkminf16, 1); R(k_pkaddf16, 1); R(k_pkmulf16, 1);
    R(k_minf32, 1); R(k_max3f, 1); R(k_mulf32, 1); R(k_fmaf32, 1); R(k_perm, 1); R(k_mad24, 1); R(k_lshladd, 1); R(k_andor, 1); R(k_bfe, 1);
    R(k_alignbit, 1); R(k_cvtf, 1); R(k_rndne, 1); R(k_cvti, 1); R(k_mullo, 1); R(k_bcnt, 1); R(k_sadu8, 1); R(k_cmpsel, 2); R(k_dpp, 1); R(k_bperm, 1);
    R(k_pkmadu16, 1); R(k_pkmulu16, 1); R(k_pkaddu16, 1); R(k_dot4u8, 1); R(k_addf32, 1); R(k_cvtub1, 1); R(k_floor, 1); R(k_cvtpku8, 1);
    R(k_alignbyte, 1); R(k_addsdwa, 1); R(k_cvtu32f, 1);
    // semantics probe: v_cvt_pk_u8_f32 rounding / saturation
    {
        const float in[12] = {0.0f, 0.49f, 0.5f, 0.99f, 1.5f, 2.5f, 254.99f, 255.0f, 255.7f, 256.0f, 300.0f, -1.0f};
        float* di; unsigned* dout; (void)hipMalloc(&di, sizeof(in)); (void)hipMalloc(&dout, 12 * 4);
        (void)hipMemcpy(di, in, sizeof(in), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe_cvtpk, dim3(1), dim3(64), 0, 0, di, dout);
        unsigned out[12]; (void)hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
        for (int i = 0; i < 12; i++) printf("cvt_pk_u8_f32(%g) -> 0x%08x\n", in[i], out[i]);
    }
    return 0;
}
