// Box probe: the smallest possible use of the system HIP runtime — no product code.  If THIS faults, the box (or its runtime /
// kernel-driver pairing) is broken, not liborbhip.so.   build: make -C tests/cpp hip_touch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } else printf("ok   %s\n", #x); fflush(stdout); } while (0)
__global__ void k_inc(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main()
{
    int nd = 0, rt = 0, drv = 0;
    CK(hipGetDeviceCount(&nd)); CK(hipRuntimeGetVersion(&rt)); CK(hipDriverGetVersion(&drv));
    printf("devices %d runtime %d driver %d\n", nd, rt, drv);
    CK(hipSetDevice(0));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); printf("%s %s CUs %d\n", pr.name, pr.gcnArchName, pr.multiProcessorCount);
    for (size_t n : {(size_t)260, (size_t)8192, (size_t)(1 << 20), (size_t)(16 << 20)}) {
        std::vector<int> h(n, 7), back(n, 0); int* d = nullptr;
        printf("-- %zu ints, pageable\n", n);
        CK(hipMalloc((void**)&d, n * 4));
        CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        k_inc<<<(n + 255) / 256, 256>>>(d, (int)n); CK(hipGetLastError()); CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), d, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) if (back[i] != 8) { printf("FAIL value at %zu\n", i); return 1; }
        int* hp = nullptr; CK(hipHostMalloc((void**)&hp, n * 4, hipHostMallocDefault));
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipMemcpyAsync(hp, d, n * 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        if (hp[n - 1] != 8) { printf("FAIL pinned value\n"); return 1; }
        CK(hipStreamDestroy(s)); CK(hipHostFree(hp)); CK(hipFree(d));
    }
    printf("hip_touch ok\n");
    return 0;
}
