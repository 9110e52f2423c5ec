// How should the candidate filter's inner loop be shaped?  One "tile" = 12 MFMA 32x32x16 f16 (4 chains x 3) + the 64
// v_alignbit that pack the signs of its 64 accumulator registers (scan_mfma.hip, cand_read).  Variants:
//   0  as the kernel is now: MFMAs of tile t, then the packing of tile t (compiler's schedule)
//   1  software pipeline: MFMAs of tile t next to the packing of tile t-1 (two accumulator sets), compiler's schedule
//   2  the same with sched_group_barrier: 1 MFMA, then 5 VALU, twelve times; the last 4 VALU after
//   3  variant 2 with 4-long packing chains (tree) instead of 16-long ones
//   4  MFMAs only      5  packing only
// Each is run with 1, 2 and 3 blocks of 256 threads per CU (= waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o cand_pipe cand_pipe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ uint32_t pack16(const f32x16& a) {
    uint32_t v = 0;
#pragma unroll
    for (int r = 15; r >= 0; r--) v = __builtin_amdgcn_alignbit(v, __float_as_uint(a[r]), 31);
    return v;
}
static __device__ __forceinline__ uint32_t pack16_tree(const f32x16& a) {
    uint32_t q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t v = __float_as_uint(a[4 * j + 3]) >> 31;
#pragma unroll
        for (int r = 2; r >= 0; r--) v = __builtin_amdgcn_alignbit(v, __float_as_uint(a[4 * j + r]), 31);
        q[j] = v;
    }
    return (((q[3] << 4 | q[2]) << 4 | q[1]) << 4) | q[0];
}

template <int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(uint32_t* out, int iters, const uint4* bsrc) {
    f16x8 A[4][3], B[3];
    for (int g = 0; g < 4; g++)
        for (int t = 0; t < 3; t++)
            for (int i = 0; i < 8; i++) A[g][t][i] = (_Float16)(((threadIdx.x + g * 7 + t * 3 + i) & 7) - 3.5f);
    f32x16 acc0[4], acc1[4];
    for (int g = 0; g < 4; g++)
        for (int r = 0; r < 16; r++) { acc0[g][r] = __uint_as_float(bsrc[(threadIdx.x + g * 16 + r) & 255].x); acc1[g][r] = (MODE >= 1 && MODE <= 3) ? acc0[g][r] * 2.f : 0.f; }
    uint32_t sink = 0;
    f32x16 C;
    for (int r = 0; r < 16; r++) C[r] = 4.0f;
    for (int it = 0; it < iters; it++) {
        for (int t = 0; t < 3; t++) B[t] = __builtin_bit_cast(f16x8, bsrc[(it * 3 + t + threadIdx.x) & 255]);   // an L1-resident operand stream
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 4; g++) acc0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], C, 0, 0, 0);
#pragma unroll
            for (int t = 1; t < 3; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) acc0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc0[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; g++) sink += pack16(acc0[g]);
        } else if (MODE == 1 || MODE == 2 || MODE == 3) {
            // MFMAs of this tile into acc1 next to the packing of the previous tile (acc0); then the roles swap
#pragma unroll
            for (int half = 0; half < 2; half++) {
                f32x16(&cur)[4] = half ? acc0 : acc1;
                f32x16(&prev)[4] = half ? acc1 : acc0;
#pragma unroll
                for (int g = 0; g < 4; g++) cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], C, 0, 0, 0);
#pragma unroll
                for (int t = 1; t < 3; t++)
#pragma unroll
                    for (int g = 0; g < 4; g++) cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], cur[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; g++) sink += MODE == 3 ? pack16_tree(prev[g]) : pack16(prev[g]);
                if (MODE >= 2) {
#pragma unroll
                    for (int i = 0; i < 12; i++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                }
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int g = 0; g < 4; g++) acc0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], C, 0, 0, 0);
#pragma unroll
            for (int t = 1; t < 3; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) acc0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc0[g], 0, 0, 0);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                sink += pack16(acc0[g]);
                acc0[g][g] += __uint_as_float(sink & 0x3f800000u);
            }
        }
    }
    float s = 0;
    for (int g = 0; g < 4; g++)
        for (int r = 0; r < 16; r++) s += acc0[g][r] + ((MODE >= 1 && MODE <= 3) ? acc1[g][r] : 0.f);
    out[blockIdx.x * 256 + threadIdx.x] = sink ^ __float_as_uint(s);
}

template <int MODE, int WPE>
static void run(const char* name, uint32_t* out, const uint4* bsrc) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WPE>), dim3(256 * WPE), dim3(256), 0, 0, out, iters, bsrc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const int tiles = (MODE >= 1 && MODE <= 3) ? 2 * iters : iters;
    printf("%-44s %d waves/SIMD  %7.3f ms  %6.1f ns per tile per SIMD\n", name, WPE, ms, ms * 1e6 / tiles / WPE);
}

int main() {
    uint32_t* out;
    uint4* bsrc;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMalloc(&bsrc, 256 * 16);
    hipMemset(bsrc, 0x3c, 256 * 16);
#define ALL(M, NAME) run<M, 1>(NAME, out, bsrc); run<M, 2>(NAME, out, bsrc); if (M == 0 || M >= 4) run<M, 3>(NAME, out, bsrc);
    ALL(4, "12 MFMA only")
    ALL(5, "64 alignbit only")
    ALL(0, "MFMA then pack (as now)")
    ALL(1, "pipelined, compiler schedule")
    ALL(2, "pipelined, 1 MFMA : 5 VALU")
    ALL(3, "pipelined, 1 MFMA : 5 VALU, tree packing")
    return 0;
}
