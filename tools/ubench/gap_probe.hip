// What opens the ~10 us gaps in front of the large kernels of a scan step (tools/step_sequence.py)?  Chains of empty kernels on one stream, run under
// `rocprofv3 --kernel-trace --output-format csv` and read with tools/step_sequence.py: tiny grid, big grid, big grid + dynamic LDS, big grid + a 640-byte
// kernarg struct, big grid doing 100 us of work.   hipcc --offload-arch=gfx950 -O3 -o gap_probe gap_probe.hip && rocprofv3 ... -- ./gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long long v[80]; };
__global__ void k_tiny(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big_lds(int* p) { extern __shared__ int sm[]; if (p && threadIdx.x == 9999) *p = sm[0]; }
__global__ void k_big_arg(Big b, int* p) { if (p && threadIdx.x == 9999) *p = (int)b.v[3]; }
__global__ void k_big_work(float* p, int n) {     // ~100 us of streaming
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < n; r++) p[i + (size_t)r * gridDim.x * blockDim.x] = (float)r;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float* buf; hipMalloc(&buf, (size_t)8192 * 256 * 4 * 64);
    Big b{};
    hipFuncSetAttribute((const void*)k_big_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big, dim3(8192), dim3(256), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_lds, dim3(8192), dim3(256), 48 * 1024, st, (int*)nullptr);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_arg, dim3(8192), dim3(256), 0, st, b, (int*)nullptr);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_work, dim3(8192), dim3(256), 0, st, buf, 64);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_work, dim3(8192), dim3(256), 0, st, buf, 64);
        hipLaunchKernelGGL(k_big_work, dim3(8192), dim3(256), 0, st, buf, 64);
        hipLaunchKernelGGL(k_big, dim3(8192), dim3(256), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big, dim3(8192), dim3(256), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_lds, dim3(8192), dim3(256), 48 * 1024, st, (int*)nullptr);
        hipLaunchKernelGGL(k_big_lds, dim3(8192), dim3(256), 48 * 1024, st, (int*)nullptr);
    }
    hipStreamSynchronize(st);
    printf("done\n");
    return 0;
}
