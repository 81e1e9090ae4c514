// Cost of a device-scope barrier among the workgroups of ONE launch on MI355X (the building block a one-launch pyramid would need between levels):
// NB workgroups of 256 threads, NITER barriers each = agent-scope release (L2 write-back), atomic arrive, spin, agent-scope acquire (L2 invalidate);
// between two barriers every workgroup stores and then loads a word another workgroup wrote (so the fences have something to do).
// build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_ubench.hip -o tools/grid_barrier_ubench     usage: tools/grid_barrier_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_barriers(unsigned* ctr, unsigned* data, int niter, unsigned base)
{
    const unsigned nb = gridDim.x;
    unsigned acc = 0;
    for (int it = 0; it < niter; it++) {
        if (threadIdx.x == 0) data[blockIdx.x] = it + acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(ctr, 1u);
            const unsigned target = base + nb * (unsigned)(it + 1);
            while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        acc += data[(blockIdx.x + 1) % nb];
    }
    if (threadIdx.x == 0) data[nb + blockIdx.x] = acc;
}
__global__ void k_empty(unsigned* p) { if (threadIdx.x == 1000) p[0] = 1; }
int main()
{
    unsigned *ctr, *data; hipMalloc(&ctr, 64); hipMalloc(&data, 1 << 20); hipMemset(ctr, 0, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned base = 0;
    for (int nb : {8, 50, 128, 256, 512}) {
        for (int niter : {1, 7, 64}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a); hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, 0, ctr, data, niter, base); hipEventRecord(b); hipEventSynchronize(b);
                base += (unsigned)nb * niter;
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("workgroups %4d  barriers %3d  kernel %.1f us  -> %.2f us per barrier (incl. the launch for 1)\n", nb, niter, best * 1e3, best * 1e3 / niter);
        }
    }
    // a chain of dependent tiny launches for comparison
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) { hipEventRecord(a); for (int k = 0; k < 64; k++) hipLaunchKernelGGL(k_empty, dim3(50), dim3(256), 0, 0, data); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    printf("64 dependent empty launches of 50 workgroups: %.2f us each\n", best * 1e3 / 64);
    return 0;
}
