// Does VALU work overlap with MFMA work on one SIMD?  (gfx950)  hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, uint32_t seed) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x & 3); b[i] = (_Float16)(threadIdx.x & 1); }
    f32x16 acc[4];
    for (int g = 0; g < 4; g++) for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
    uint32_t v[4] = {seed, seed + 1, seed + 2, seed + 3};
    uint32_t x = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++)
#pragma unroll
                for (int g = 0; g < 4; g++) { v[g] = __builtin_amdgcn_alignbit(v[g], x, 31); x += v[g]; }
        }
        if (MODE == 3) {   // hand interleave: 1 MFMA then ~11 VALU
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 5; r++) { v[(r + g) & 3] = __builtin_amdgcn_alignbit(v[(r + g) & 3], x, 31); x += v[(r + g) & 3]; }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float s = 0;
    for (int g = 0; g < 4; g++) for (int r = 0; r < 16; r++) s += acc[g][r];
    out[blockIdx.x * 512 + threadIdx.x] = v[0] ^ v[1] ^ v[2] ^ v[3] ^ x ^ __float_as_uint(s);
}

template <int MODE>
static void run(const char* name, int threads) {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0;
    printf("%-28s threads/block %4d  %.3f ms  -> %.0f ns per iteration per wave-slot (%.1f waves/SIMD)\n", name, threads, ms,
           ms * 1e6 / iters / waves_per_simd, waves_per_simd);
    hipFree(out);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("12 MFMA 32x32x16", threads);
        run<1>("128 VALU (64 alignbit+64 add)", threads);
        run<2>("both, back to back", threads);
        run<3>("both, hand-interleaved 1:10", threads);
    }
    return 0;
}
