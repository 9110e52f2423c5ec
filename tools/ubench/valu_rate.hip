// Microbenchmark: issue rate of v_pk_add_f16 / v_add_f16 / v_add_f32 / v_pk_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 3 + i;
    unsigned t = 0x3c003c00u + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(t));
            if (OP == 1) asm volatile("v_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(t));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
            if (OP == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(t));
            if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(t));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(t));
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int wavesPerSimd) {
    int blocks = 256 * wavesPerSimd;  // 256 CUs x (4 waves/block = 1 wave per SIMD per block)
    float* d;
    hipMalloc(&d, blocks * 256 * 4);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 16 * wavesPerSimd;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms,
           cyc / instr_per_simd);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_pk_add_f16", w);
        run<1>("v_add_f16", w);
        run<2>("v_add_f32", w);
        run<3>("v_pk_max_i16", w);
        run<4>("v_mov_b32", w);
        run<5>("v_cndmask_b32", w);
    }
    return 0;
}
