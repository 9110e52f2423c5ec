// What a dependent launch costs on this part: N tiny kernels (each reads what the previous one wrote) on one stream, eager and
// replayed from a hipGraph.  The reference-schedule train step is ~430 such launches.
//   hipcc --offload-arch=gfx950 -O2 -o build/launch_floor tools/ubench/launch_floor.hip && build/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
int main() {
    const int N = 2000;
    for (int blocks : {1, 36, 256, 1772}) {
        const int n = blocks * 256;
        float *a, *b;
        CK(hipMalloc(&a, n * 4));
        CK(hipMalloc(&b, n * 4));
        CK(hipMemset(a, 0, n * 4));
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        auto run = [&]() {
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
        };
        run();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        run();
        CK(hipStreamSynchronize(st));
        const double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        run();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("blocks %5d: %.2f us per dependent launch eager, %.2f us replayed from a graph\n", blocks, eager, graph);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipFree(a));
        CK(hipFree(b));
        CK(hipStreamDestroy(st));
    }
    return 0;
}
