// hip_runtime.h — TEST-ONLY functional emulation of the subset of HIP that orb_slam2_amd/csrc uses.
//
// Purpose: this container has no GPU.  To debug kernel LOGIC (index math, barriers, wave collectives,
// list-order semantics) before spending scarce GPU minutes, tests/emu builds the unmodified product sources
// (orb_slam2_amd/csrc/*.hip, *.cpp) with g++ against this header and runs every workgroup as a set of
// cooperatively scheduled fibers (a hand-written x86-64 context switch).  __syncthreads and the wave64 collectives (__ballot, __shfl*,
// __any/__all) are rendezvous points.  It is NOT a product path, NOT a CPU fallback and is never shipped or
// benchmarked: liborbhip.so is built by hipcc for gfx950 only and fails loudly without a GPU.
// Semantics emulated: wave = 64 consecutive threads of a block; collectives are assumed to be called
// convergently by all live lanes of a wave (as on hardware, anything else is a kernel bug).
#pragma once
#include <time.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include <mutex>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::g.dyn_shared);

struct dim3 { unsigned x, y, z; constexpr dim3(unsigned X = 1, unsigned Y = 1, unsigned Z = 1) : x(X), y(Y), z(Z) {} };
struct uint3 { unsigned x, y, z; };
struct short2 { short x, y; };
struct int2 { int x, y; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

namespace hipemu {

struct Fiber {
    void* sp = nullptr; char* stack = nullptr; bool done = false; uint3 tid; int lane, wave;
};
struct Wave {
    int live = 0, arrived = 0; unsigned gen = 0;
    uint64_t in[64]; int src[64]; uint64_t out[64]; bool present[64];
};
struct Globals {
    void* sched_sp = nullptr; Fiber* cur = nullptr; std::vector<Fiber> fibers; std::vector<Wave> waves;
    uint3 bid; dim3 bdim, gdim; int live = 0, bar_count = 0; unsigned bar_gen = 0;
    char* dyn_shared = nullptr; std::function<void()> body; size_t stack_size = 128 * 1024;
};
inline Globals g;
inline std::mutex launch_mutex;          // one kernel at a time: the scheduler state above and the kernels' static __shared__ arrays are process-wide

// Context switch of the cooperative fibers (x86-64 SysV): callee-saved registers + stack pointer, no system call.  glibc's swapcontext saves
// and restores the signal mask with two rt_sigprocmask calls per switch, which was most of the emulation's run time (every wave collective
// is 128 switches).
__attribute__((naked, noinline)) static void switch_to(void** /*save_sp*/, void* /*load_sp*/) {
    __asm__ volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
                     "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
                     "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
#define HIPEMU_NO_TSAN __attribute__((no_sanitize_thread))      // ThreadSanitizer builds (make emu_tsan): frames that do not return are not instrumented
HIPEMU_NO_TSAN inline void yield() { switch_to(&g.cur->sp, g.sched_sp); }
HIPEMU_NO_TSAN inline void fiber_main() {
    g.body();
    Fiber* f = g.cur; f->done = true; g.live--; g.waves[f->wave].live--;
    switch_to(&f->sp, g.sched_sp);
    abort();                                                           // a finished fiber is never resumed
}
inline void syncthreads() {
    unsigned gen = g.bar_gen; g.bar_count++;
    while (g.bar_gen == gen) { if (g.bar_count >= g.live) { g.bar_count = 0; g.bar_gen++; break; } yield(); }
}
enum Op { OP_BALLOT, OP_SHFL };
// rendezvous of all live lanes of the calling wave; the last arriver computes every lane's result
inline uint64_t wave_collective(Op op, uint64_t val, int srclane) {
    Fiber* f = g.cur; Wave& w = g.waves[f->wave];
    unsigned gen = w.gen; w.in[f->lane] = val; w.src[f->lane] = srclane; w.present[f->lane] = true; w.arrived++;
    while (w.gen == gen) {
        if (w.arrived >= w.live) {
            if (op == OP_BALLOT) { uint64_t m = 0; for (int l = 0; l < 64; l++) if (w.present[l] && w.in[l]) m |= 1ull << l; for (int l = 0; l < 64; l++) w.out[l] = m; }
            else for (int l = 0; l < 64; l++) { int s = w.src[l]; w.out[l] = (w.present[l] && s >= 0 && s < 64 && w.present[s]) ? w.in[s] : w.in[l]; }
            for (int l = 0; l < 64; l++) w.present[l] = false;
            w.arrived = 0; w.gen++; break;
        }
        yield();
    }
    return w.out[f->lane];
}

template <typename K, typename... A>
HIPEMU_NO_TSAN void launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
    std::lock_guard<std::mutex> serialise(launch_mutex);   // host threads (the reference extracts left / right images on two std::threads) take turns
    const int T = (int)(block.x * block.y * block.z);
    std::vector<char> dyn(shmem + 64);
    const bool poison_lds = getenv("HIPEMU_POISON_LDS") != nullptr;
    g.dyn_shared = dyn.data(); g.bdim = block; g.gdim = grid;
    if ((int)g.fibers.size() < T) { size_t old = g.fibers.size(); g.fibers.resize(T); for (size_t i = old; i < (size_t)T; i++) g.fibers[i].stack = (char*)malloc(g.stack_size); }
    g.body = [&]() { kernel(args...); };
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        g.bid = uint3{bx, by, bz}; g.live = T; g.bar_count = 0; g.bar_gen = 0;
        if (poison_lds) memset(dyn.data(), 0xa5, dyn.size());          // HIPEMU_POISON_LDS=1: a workgroup starts on garbage, as on the device (reads of never-written LDS show)
        g.waves.assign((T + 63) / 64, Wave());
        for (auto& w : g.waves) for (int l = 0; l < 64; l++) w.present[l] = false;
        for (int t = 0; t < T; t++) {
            Fiber& f = g.fibers[t]; f.done = false;
            f.tid = uint3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
            f.lane = t & 63; f.wave = t >> 6; g.waves[f.wave].live++;
            // first switch into the fiber pops six zeroed registers and returns into fiber_main with the stack the ABI expects at a function entry
            void** sp = (void**)(((uintptr_t)(f.stack + g.stack_size) & ~(uintptr_t)15) - 64);
            for (int k = 0; k < 6; k++) sp[k] = nullptr;
            sp[6] = (void*)(void (*)())fiber_main; sp[7] = nullptr;
            f.sp = sp;
        }
        while (g.live > 0)
            for (int t = 0; t < T; t++) { Fiber& f = g.fibers[t]; if (f.done) continue; g.cur = &f; switch_to(&g.sched_sp, f.sp); }
    }
    g.cur = nullptr;
}
}  // namespace hipemu

#define threadIdx (hipemu::g.cur->tid)
#define blockIdx (hipemu::g.bid)
#define blockDim (hipemu::g.bdim)
#define gridDim (hipemu::g.gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

inline void __syncthreads() { hipemu::syncthreads(); }
inline unsigned long long __ballot(int pred) { return hipemu::wave_collective(hipemu::OP_BALLOT, pred != 0, 0); }
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred ? 1 : 0); }
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return hipemu::wave_collective(hipemu::OP_BALLOT, pred == 0, 0) == 0; }
inline int __lane_id() { return hipemu::g.cur->lane; }
template <typename T> inline T hipemu_shfl(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl size"); uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    raw = hipemu::wave_collective(hipemu::OP_SHFL, raw, src); T r; memcpy(&r, &raw, sizeof(T)); return r;
}
template <typename T> inline T __shfl(T v, int srcLane, int width = 64) { int l = __lane_id(); return hipemu_shfl(v, (l & ~(width - 1)) + (srcLane & (width - 1))); }
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) { int l = __lane_id(); int s = l + (int)d; if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l; return hipemu_shfl(v, s); }
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) { int l = __lane_id(); int s = l - (int)d; if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l; return hipemu_shfl(v, s); }
template <typename T> inline T __shfl_xor(T v, int m, int width = 64) { int l = __lane_id(); int s = l ^ m; if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l; return hipemu_shfl(v, s); }

inline int __builtin_amdgcn_readlane(int v, int srcLane) { return hipemu_shfl(v, srcLane); }     // v_readlane_b32: uniform source lane
// v_mov_b32_dpp semantics for the controls the kernels use: quad_perm (0x00-0xff), row_shl:n (0x100+n), row_shr:n (0x110+n), row_ror:n (0x120+n),
// row_half_mirror (0x141), row_bcast:15 (0x142), row_bcast:31 (0x143) - as tools/permlane_swap_probe.hip prints them on the device
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = __lane_id(), row = l >> 4, bank = (l >> 2) & 3;
    int s = -1;
    if (ctrl >= 0 && ctrl <= 0xff) s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                                         // quad_perm
    else if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl - 0x120; s = (l & ~15) | (((l & 15) - n) & 15); }     // row_ror:n
    else if (ctrl == 0x141) s = (l & ~7) | (7 - (l & 7));                                                                // row_half_mirror
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; s = ((l & 15) >= n) ? l - n : -1; }
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; s = ((l & 15) + n < 16) ? l + n : -1; }      // row_shl:n
    else if (ctrl == 0x130) s = (l + 1 < 64) ? l + 1 : -1;                                                                // wave_shl:1
    else if (ctrl == 0x142) s = (row >= 1) ? row * 16 - 1 : -1;
    else if (ctrl == 0x143) s = (row >= 2) ? 31 : -1;
    const int got = hipemu_shfl(src, s < 0 ? l : s);
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    if (s < 0) return bound_ctrl ? 0 : old;
    return got;
}
// v_permlane32_swap / v_permlane16_swap (gfx950): [0] = a with its upper half / odd rows replaced by b's lower half / even rows, [1] = the rest
typedef unsigned hipemu_v2u __attribute__((vector_size(8)));
inline hipemu_v2u __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const int l = __lane_id(); const unsigned xa = hipemu_shfl(a, l ^ 32), xb = hipemu_shfl(b, l ^ 32);
    hipemu_v2u r; r[0] = l < 32 ? a : xb; r[1] = l < 32 ? xa : b; return r;
}
inline hipemu_v2u __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    const int l = __lane_id(); const unsigned xa = hipemu_shfl(a, l ^ 16), xb = hipemu_shfl(b, l ^ 16);
    hipemu_v2u r; r[0] = (l & 16) ? xb : a; r[1] = (l & 16) ? b : xa; return r;
}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_collective(hipemu::OP_BALLOT, 0, 0); }   // rendezvous of the wave's lanes
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// v_perm_b32: selector bytes 0-3 pick bytes of s1, 4-7 bytes of s0, 0x0c -> 0x00, >= 0x0d -> 0xff
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1; unsigned r = 0;
    for (int i = 0; i < 4; i++) { const unsigned c = (sel >> (8 * i)) & 0xff; unsigned b;
        if (c <= 7) b = (unsigned)(src >> (8 * c)) & 0xff; else if (c == 0x0c) b = 0; else b = 0xff; r |= b << (8 * i); }
    return r;
}
// v_alignbyte_b32: ({hi,lo} >> 8*shift)[31:0]
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned shift) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (shift & 3)));
}
// v_dot4_u32_u8: sum of the four byte products + c
inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c) {                     // v_sad_u8: sum of |a.byte - b.byte| + c
    for (int k = 0; k < 4; k++) { const int x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff; c += (unsigned)(x > y ? x - y : y - x); }
    return c;
}
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool /*clamp*/) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
    return c;
}
// v_mfma_i32_32x32x32_i8: D = A x B + C on one wave.  A[i][k]: lane i + 32h, byte b <-> k = (h, b); B[k][j]: lane j + 32h, same k;
// C / D[i][j]: lane j + 32 * ((i >> 2) & 1), register (i & 3) + 4 * (i >> 3).  Every lane of the wave must call (waves are whole here).
typedef int hipemu_v4i __attribute__((vector_size(16)));
typedef int hipemu_v16i __attribute__((vector_size(64)));
inline hipemu_v16i __builtin_amdgcn_mfma_i32_32x32x32_i8(hipemu_v4i a, hipemu_v4i b, hipemu_v16i c, int, int, int) {
    static signed char A[32][64][16], B[32][64][16];
    const int w = hipemu::g.cur->wave & 31, l = __lane_id();
    memcpy(A[w][l], &a, 16); memcpy(B[w][l], &b, 16);
    __builtin_amdgcn_wave_barrier();
    hipemu_v16i d = c;
    const int j = l & 31, hh = l >> 5;
    for (int q = 0; q < 16; q++) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * hh; int acc = 0;
        for (int h = 0; h < 2; h++) for (int bb = 0; bb < 16; bb++) acc += (int)A[w][i + 32 * h][bb] * (int)B[w][j + 32 * h][bb];
        d[q] += acc;
    }
    __builtin_amdgcn_wave_barrier();
    return d;
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands (cbsz = blgp = 4): D = (2^(sa-127) A) x (2^(sb-127) B) + C on one wave.  As this header assumes it -
// A[i][k]: lane i + 32h, nibble p of its first 16 bytes <-> k = 32h + p; B[k][j]: lane j + 32h, same k; one E8M0 scale per lane = per block of 32;
// C / D as the i8 form.  (Only the hardware can say whether the k order inside a lane is this one; the kernels use the same order on both operands.)
typedef int hipemu_v8i __attribute__((vector_size(32)));
typedef float hipemu_v16f __attribute__((vector_size(64)));
inline hipemu_v16f __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hipemu_v8i a, hipemu_v8i b, hipemu_v16f c, int cbsz, int blgp, int, int scale_a, int, int scale_b) {
    static unsigned char A[32][64][16], B[32][64][16]; static float SA[32][64], SB[32][64];
    static const float e2m1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    if (cbsz != 4 || blgp != 4) abort();                              // only the FP4 form is emulated
    const int w = hipemu::g.cur->wave & 31, l = __lane_id();
    memcpy(A[w][l], &a, 16); memcpy(B[w][l], &b, 16);
    SA[w][l] = ldexpf(1.0f, (scale_a & 0xff) - 127); SB[w][l] = ldexpf(1.0f, (scale_b & 0xff) - 127);
    __builtin_amdgcn_wave_barrier();
    auto val = [&](const unsigned char* p, int k) { const unsigned n = (p[k >> 1] >> (4 * (k & 1))) & 0xf; const float v = e2m1[n & 7]; return (n & 8) ? -v : v; };
    hipemu_v16f d = c;
    const int j = l & 31, hh = l >> 5;
    for (int q = 0; q < 16; q++) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * hh; float acc = 0.f;
        for (int h = 0; h < 2; h++) for (int k = 0; k < 32; k++) acc += SA[w][i + 32 * h] * val(A[w][i + 32 * h], k) * SB[w][j + 32 * h] * val(B[w][j + 32 * h], k);
        d[q] += acc;
    }
    __builtin_amdgcn_wave_barrier();
    return d;
}
// v_dot2_u32_u16: two u16 products + c
template <typename V> inline unsigned __builtin_amdgcn_udot2(V a, V b, unsigned c, bool /*clamp*/) { return (unsigned)a[0] * (unsigned)b[0] + (unsigned)a[1] * (unsigned)b[1] + c; }
// v_cvt_pk_u8_f32: round to nearest even, saturate to [0, 255], insert into byte `pos` of `old` (measured on MI355X:
// 0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 255.7 -> 255, 300 -> 255, -1 -> 0; tools/ubench.hip probe)
inline unsigned __builtin_amdgcn_cvt_pk_u8_f32(float v, unsigned pos, unsigned old) {
    float r = __builtin_nearbyintf(v); if (!(r > 0.0f)) r = 0.0f; if (r > 255.0f) r = 255.0f;
    const unsigned sh = 8 * (pos & 3);
    return (old & ~(0xffu << sh)) | ((unsigned)r << sh);
}
// v_pk_sub_u16 clamp (clang's __builtin_elementwise_sub_sat on a two-lane u16 vector): unsigned saturating subtraction per lane
template <typename V> inline V __builtin_elementwise_sub_sat(V a, V b) { return a > b ? a - b : a - a; }
// v_mbcnt_lo_u32_b32 / v_mbcnt_hi_u32_b32: set bits of the mask half below this lane, plus base
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) { const int l = __lane_id(); return base + (unsigned)__builtin_popcount(mask & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u))); }
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) { const int l = __lane_id(); return base + (unsigned)__builtin_popcount(l > 32 ? mask & ((1u << (l - 32)) - 1u) : 0u); }
// global_load_lds_dword (LDS-DMA): lane l's dword lands at LDS dword l of the block; s_waitcnt is a no-op here
inline void __builtin_amdgcn_global_load_lds(const void* gsrc, void* lds_block, unsigned size, int offset, unsigned) { memcpy((char*)lds_block + offset + (size_t)__lane_id() * size, (const char*)gsrc + offset, size); }
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline int __mul24(int a, int b) { return a * b; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
// v_readfirstlane_b32: the kernels only apply it to wave-uniform values (to move them to the scalar unit)
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// atomics (single OS thread: plain read-modify-write is atomic w.r.t. fibers)
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
inline void __threadfence() {}
template <typename... A> inline void __builtin_amdgcn_fence(int, A...) {}      // (scope [, address space: "local" = LDS only])
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline void __threadfence_block() {}

// IEEE round-to-nearest arithmetic intrinsics (the emu build uses -ffp-contract=off, so plain ops are exact)
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }       // correctly rounded fused multiply-add
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}            // a scheduling directive: nothing to emulate
inline double __ddiv_rn(double a, double b) { return a / b; }
inline double __dsqrt_rn(double a) { return sqrt(a); }
inline int __float2int_rn(float v) { return (int)lrintf(v); }
inline int __float2int_rz(float v) { return (int)v; }
inline int __double2int_rz(double v) { return (int)v; }
inline float __double2float_rn(double v) { return (float)v; }
inline float __int2float_rn(int v) { return (float)v; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
using std::min; using std::max;
inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }

// ---- host API
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipRuntimeGetVersion(int* v) { *v = 0; return hipSuccess; }
inline hipError_t hipDriverGetVersion(int* v) { *v = 0; return hipSuccess; }
// HIPEMU_DEVICE_COUNT=n: pretend n devices (all of them this process's heap) so that the multi-device host logic runs on CPU
inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("HIPEMU_DEVICE_COUNT"); *n = e ? std::max(atoi(e), 1) : 1; return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; };
// HIPEMU_ALL_PINNED=1: every host pointer counts as pinned (exercises the direct-DMA branches of the host path)
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { const char* e = getenv("HIPEMU_ALL_PINNED"); if (e && e[0] == '1') { a->type = hipMemoryTypeHost; a->device = 0; return hipSuccess; } return hipErrorInvalidValue; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu (CPU fibers, test only)"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
typedef void* hipDeviceptr_t;
inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t = nullptr) { for (size_t i = 0; i < n; i++) static_cast<int*>(d)[i] = v; return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetPCIBusId(char* buf, int len, int dev) { snprintf(buf, (size_t)len, "emu0:%02x:00.0", dev); return hipSuccess; }      // no such sysfs entry: NUMA node unknown
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline double hipemu_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = hipemu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
