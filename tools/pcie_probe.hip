// PCIe ceiling of the box as the host path sees it: pinned H2D, pinned D2H, both at once (two streams), in the shapes orbhip_submit uses
// (30 MB chunks of frames up, 7.7 MB of results down).   build: hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o tools/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t up = 64ull * 466616, down = 64ull * 121444;      // one 64-frame chunk of 1241x376 frames / of 2024-key-point results
    const int reps = 40;
    uint8_t *h_up, *h_down, *d_up, *d_down;
    CK(hipHostMalloc((void**)&h_up, up * 2, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h_down, down * 2, hipHostMallocDefault));
    CK(hipMalloc((void**)&d_up, up * 2)); CK(hipMalloc((void**)&d_down, down * 2));
    memset(h_up, 1, up * 2); memset(h_down, 0, down * 2);
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; mode++) {
        for (int w = 0; w < 2; w++) { CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2)); }
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; r++) {
            if (mode != 1) CK(hipMemcpyAsync(d_up + (r & 1) * up, h_up + (r & 1) * up, up, hipMemcpyHostToDevice, s1));
            if (mode != 0) CK(hipMemcpyAsync(h_down + (r & 1) * down, d_down + (r & 1) * down, down, hipMemcpyDeviceToHost, s2));
        }
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        const double gb_up = mode != 1 ? reps * (double)up / dt / 1e9 : 0, gb_dn = mode != 0 ? reps * (double)down / dt / 1e9 : 0;
        printf("%s: H2D %.1f GB/s  D2H %.1f GB/s  (frames/s the link alone would allow: %.0f)\n", mode == 0 ? "H2D only" : mode == 1 ? "D2H only" : "both    ", gb_up, gb_dn,
               mode == 2 ? reps * 64 / dt : 0.0);
    }
    // many small copies (one per frame) against one per chunk
    { CK(hipDeviceSynchronize()); const double t0 = now();
      for (int r = 0; r < reps; r++) for (int f = 0; f < 64; f++) CK(hipMemcpyAsync(d_up + (size_t)f * 466616, h_up + (size_t)f * 466616, 466616, hipMemcpyHostToDevice, s1));
      CK(hipDeviceSynchronize()); const double dt = now() - t0; printf("H2D one copy per frame: %.1f GB/s\n", reps * (double)up / dt / 1e9); }
    return 0;
}
