// Do two kernels on streams with disjoint CU masks run side by side?  Each kernel spins a fixed number of VALU iterations in
// every block (one block per CU of its half); the pair is timed against one kernel alone.
// build: hipcc --offload-arch=gfx950 -O3 -o cu_mask_overlap cu_mask_overlap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* out, int iters) {
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    if (x == 123.f) out[0] = x;
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t ma[16] = {0}, mb[16] = {0};
    for (int cu = 0; cu < cus; cu++) (cu < cus / 2 ? ma : mb)[cu / 32] |= 1u << (cu % 32);
    hipStream_t sa, sb, plain;
    CK(hipExtStreamCreateWithCUMask(&sa, (cus + 31) / 32, ma));
    CK(hipExtStreamCreateWithCUMask(&sb, (cus + 31) / 32, mb));
    CK(hipStreamCreate(&plain));
    float* d;
    CK(hipMalloc(&d, 4));
    const int iters = 400000, blocks = cus * 4;
    auto run = [&](int mode) {   // 0: one kernel on the plain stream, 1: one on sa, 2: sa and sb, 3: both kernels on sa
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, plain, d, iters);
        if (mode == 1) hipLaunchKernelGGL(spin, dim3(blocks / 2), dim3(256), 0, sa, d, iters);
        if (mode == 2) {
            hipLaunchKernelGGL(spin, dim3(blocks / 2), dim3(256), 0, sa, d, iters);
            hipLaunchKernelGGL(spin, dim3(blocks / 2), dim3(256), 0, sb, d, iters);
        }
        if (mode == 3) {
            hipLaunchKernelGGL(spin, dim3(blocks / 2), dim3(256), 0, sa, d, iters);
            hipLaunchKernelGGL(spin, dim3(blocks / 2), dim3(256), 0, sa, d, iters);
        }
        CK(hipDeviceSynchronize());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return (int)(ms * 1000);
    };
    for (int rep = 0; rep < 2; rep++)
        printf("CUs %d: full chip, all blocks %d us | half the blocks on half the CUs %d us | two halves side by side %d us | two halves on one stream %d us\n",
               cus, run(0), run(1), run(2), run(3));
    return 0;
}
