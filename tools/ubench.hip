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
// the same with 64-bit register pairs (packed f32 operands)
#define DEFK64(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) { \
    unsigned long long a0 = 0x3f8000003f800000ull + seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned long long b = 0x3f8000013f800001ull ^ seed, c = 0x3a0000003a000000ull | seed; \
    for (int it = 0; it < N_IT; it++) { \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); }
#define A_PKMULF32(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_PKADDF32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKFMAF32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
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
// round 2: which simple integer / logic / float ops share the 2-cycle rate of v_add_u32 / v_xor_b32?
#define A_AND(i)      "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_OR(i)       "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_SUB(i)      "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_SUBREV(i)   "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i)     "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_LSHR(i)     "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define A_ASHR(i)     "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define A_NOT(i)      "v_not_b32 %" #i ", %" #i "\n"
#define A_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define A_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_OR3(i)      "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define A_BFI(i)      "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define A_XNOR(i)     "v_xnor_b32 %" #i ", %" #i ", %8\n"
#define A_CNDMASK(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i)      "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define A_SUBF32(i)   "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_FMACF32(i)  "v_fmac_f32 %" #i ", %8, %9\n"
#define A_MAXF32(i)   "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MULU24(i)   "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_ADDU16(i)   "v_add_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUBU16(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUBU16C(i) "v_pk_sub_u16 %" #i ", %" #i ", %8 clamp\n"
#define A_LERPU8(i)   "v_lerp_u8 %" #i ", %" #i ", %8, %9\n"
#define A_MSADU8(i)   "v_msad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_ADDF16(i)   "v_add_f16 %" #i ", %" #i ", %8\n"
#define A_FMAF16(i)   "v_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define A_LDEXP(i)    "v_ldexp_f32 %" #i ", %" #i ", %8\n"
#define A_MED3U(i)    "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_XAD(i)      "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADDLSHL(i)  "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
#define A_CVTUB0(i)   "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define A_MULF32K(i)  "v_mul_f32 %" #i ", 0.5, %" #i "\n"
#define A_ADDK(i)     "v_add_u32 %" #i ", 0x80808080, %" #i "\n"
#define A_ANDK(i)     "v_and_b32 %" #i ", 0x7f7f7f7f, %" #i "\n"
DEFK(k_add, A_ADD) DEFK(k_minu, A_MINU) DEFK(k_pkminu16, A_PKMINU16) DEFK(k_pkminf16, A_PKMINF16) DEFK(k_pkmaxi16, A_PKMAXI16)
DEFK64(k_pkmulf32, A_PKMULF32) DEFK64(k_pkaddf32, A_PKADDF32) DEFK64(k_pkfmaf32, A_PKFMAF32)
DEFK(k_minf32, A_MINF32) DEFK(k_fmaf32, A_FMAF32) DEFK(k_mulf32, A_MULF32) DEFK(k_min3u, A_MIN3U) DEFK(k_max3f, A_MAX3F)
DEFK(k_perm, A_PERM) DEFK(k_mad24, A_MAD24) DEFK(k_lshladd, A_LSHLADD) DEFK(k_andor, A_ANDOR) DEFK(k_cvtf, A_CVTF) DEFK(k_rndne, A_RNDNE)
DEFK(k_cvti, A_CVTI) DEFK(k_mullo, A_MULLO) DEFK(k_bfe, A_BFE) DEFK(k_alignbit, A_ALIGNBIT) DEFK(k_pkaddf16, A_PKADDF16) DEFK(k_pkmulf16, A_PKMULF16)
DEFK(k_pkmadu16, A_PKMADU16) DEFK(k_pkmulu16, A_PKMULU16) DEFK(k_pkaddu16, A_PKADDU16) DEFK(k_dot4u8, A_DOT4U8) DEFK(k_addf32, A_ADDF32)
DEFK(k_cvtub1, A_CVTUB1) DEFK(k_floor, A_FLOOR) DEFK(k_cvtpku8, A_CVTPKU8) DEFK(k_alignbyte, A_ALIGNBYTE) DEFK(k_addsdwa, A_ADDSDWA) DEFK(k_cvtu32f, A_CVTU32F)
DEFK(k_bcnt, A_BCNT) DEFK(k_xor, A_XOR) DEFK(k_sadu8, A_SADU8) DEFK(k_cmpsel, A_CMPSEL) DEFK(k_dpp, A_DPP) DEFK(k_bperm, A_BPERM)
DEFK(k_and, A_AND) DEFK(k_or, A_OR) DEFK(k_sub, A_SUB) DEFK(k_subrev, A_SUBREV) DEFK(k_lshl, A_LSHL) DEFK(k_lshr, A_LSHR) DEFK(k_ashr, A_ASHR) DEFK(k_not, A_NOT) DEFK(k_mov, A_MOV)
DEFK(k_add3, A_ADD3) DEFK(k_or3, A_OR3) DEFK(k_lshlor, A_LSHLOR) DEFK(k_bfi, A_BFI) DEFK(k_xnor, A_XNOR) DEFK(k_cndmask, A_CNDMASK) DEFK(k_cmp, A_CMP) DEFK(k_subf32, A_SUBF32)
DEFK(k_fmacf32, A_FMACF32) DEFK(k_maxf32, A_MAXF32) DEFK(k_mulu24, A_MULU24) DEFK(k_addu16, A_ADDU16) DEFK(k_pksubu16, A_PKSUBU16) DEFK(k_pksubu16c, A_PKSUBU16C) DEFK(k_lerpu8, A_LERPU8)
DEFK(k_msadu8, A_MSADU8) DEFK(k_addf16, A_ADDF16) DEFK(k_fmaf16, A_FMAF16) DEFK(k_ldexp, A_LDEXP) DEFK(k_med3u, A_MED3U) DEFK(k_addco, A_ADDCO) DEFK(k_xad, A_XAD) DEFK(k_addlshl, A_ADDLSHL)
DEFK(k_cvtub0, A_CVTUB0) DEFK(k_mulf32k, A_MULF32K) DEFK(k_addk, A_ADDK) DEFK(k_andk, A_ANDK)

// LDS read rates by width / alignment: every lane reads `LDS_READS` times per iteration from its own address (stride S bytes between lanes, byte offset O)
#define LDS_IT 256
template <int W, int S, int O> __global__ __launch_bounds__(256) void k_lds(unsigned* out, unsigned seed)
{
    __shared__ unsigned buf[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = i * seed;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(void*)buf;      // LDS byte address of buf
    asm volatile("" : "+v"(addr));
    unsigned base = (unsigned)((threadIdx.x & 63) * S + O + (threadIdx.x >> 6) * 2048);
    unsigned acc = 0;
    for (int it = 0; it < LDS_IT; it++) {
        unsigned v0, v1, v2, v3, v4, v5, v6, v7;
        const unsigned a = base + ((it & 15) << 6);
        if (W == 4) asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a));
        else if (W == 1) asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:256\n ds_read_u8 %2, %8 offset:512\n ds_read_u8 %3, %8 offset:768\n ds_read_u8 %4, %8 offset:1024\n ds_read_u8 %5, %8 offset:1280\n ds_read_u8 %6, %8 offset:1536\n ds_read_u8 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a));
        else asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:256\n ds_read_u16 %2, %8 offset:512\n ds_read_u16 %3, %8 offset:768\n ds_read_u16 %4, %8 offset:1024\n ds_read_u16 %5, %8 offset:1280\n ds_read_u16 %6, %8 offset:1536\n ds_read_u16 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a));
        acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + addr;
}
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
    R(k_pkmulf32, 1); R(k_pkaddf32, 1); R(k_pkfmaf32, 1);
    R(k_minf32, 1); R(k_max3f, 1); R(k_mulf32, 1); R(k_fmaf32, 1); R(k_perm, 1); R(k_mad24, 1); R(k_lshladd, 1); R(k_andor, 1); R(k_bfe, 1);
    R(k_alignbit, 1); R(k_cvtf, 1); R(k_rndne, 1); R(k_cvti, 1); R(k_mullo, 1); R(k_bcnt, 1); R(k_sadu8, 1); R(k_cmpsel, 2); R(k_dpp, 1); R(k_bperm, 1);
    R(k_pkmadu16, 1); R(k_pkmulu16, 1); R(k_pkaddu16, 1); R(k_dot4u8, 1); R(k_addf32, 1); R(k_cvtub1, 1); R(k_floor, 1); R(k_cvtpku8, 1);
    R(k_alignbyte, 1); R(k_addsdwa, 1); R(k_cvtu32f, 1);
    R(k_and, 1); R(k_or, 1); R(k_sub, 1); R(k_subrev, 1); R(k_lshl, 1); R(k_lshr, 1); R(k_ashr, 1); R(k_not, 1); R(k_mov, 1); R(k_add3, 1); R(k_or3, 1); R(k_lshlor, 1); R(k_bfi, 1);
    R(k_xnor, 1); R(k_cndmask, 1); R(k_cmp, 1); R(k_subf32, 1); R(k_fmacf32, 1); R(k_maxf32, 1); R(k_mulu24, 1); R(k_addu16, 1); R(k_pksubu16, 1); R(k_pksubu16c, 1); R(k_lerpu8, 1);
    R(k_msadu8, 1); R(k_addf16, 1); R(k_fmaf16, 1); R(k_ldexp, 1); R(k_med3u, 1); R(k_addco, 1); R(k_xad, 1); R(k_addlshl, 1); R(k_cvtub0, 1); R(k_mulf32k, 1); R(k_addk, 1); R(k_andk, 1);
    {   // LDS reads: cycles per wave-instruction per CU (4 waves per workgroup, 8 workgroups per CU resident)
        struct { const char* name; kfn f; } L[] = {
            {"lds b32 aligned stride4", k_lds<4, 4, 0>}, {"lds b32 +1 byte stride4", k_lds<4, 4, 1>}, {"lds b32 +2 byte stride4", k_lds<4, 4, 2>}, {"lds b32 +3 byte stride4", k_lds<4, 4, 3>},
            {"lds b32 aligned stride12", k_lds<4, 12, 0>}, {"lds b32 +3 stride12", k_lds<4, 12, 3>}, {"lds u8 stride1", k_lds<1, 1, 0>}, {"lds u8 stride4", k_lds<1, 4, 1>}, {"lds u16 +1 stride4", k_lds<2, 4, 1>}};
        for (auto& e : L) {
            unsigned* d; (void)hipMalloc(&d, 8192 * 256 * 4);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(e.f, dim3(8192), dim3(256), 0, 0, d, 1u);
            (void)hipEventRecord(e0); hipLaunchKernelGGL(e.f, dim3(8192), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double winst = 8192.0 * 4 * LDS_IT * 8;
            printf("%-28s %8.3f ms  %6.2f cycles/wave-inst/CU @2.4GHz\n", e.name, ms, 256 * 2.4e9 / (winst / (ms * 1e-3)));
            (void)hipFree(d);
        }
    }
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
