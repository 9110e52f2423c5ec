// Would the sparse matrix instruction pay for the candidate filter?  (round-3 verdict, item 2)
//
// v_smfmac_f32_32x32x32_f16 takes its SPARSE operand as A (rows of D) and the dense one as B (columns of D).  The one-hot window
// is the operand with the 2:4 structure (one 1.0 in every group of four; the base code is the index), so it has to be A: the
// windows land on D's ROWS, i.e. a lane holds one PWM and its 16 accumulator registers are 16 WINDOWS - the transpose of the dense
// form (lane = window, accumulator = PWM), whose sixteen v_alignbit per chain yield the window-major PWM word the cells and
// compact entries are made of.  Getting window-major words out of the sparse form needs a 32 x 32 bit transpose per chain:
// sixteen v_cmp (each leaves the words of two windows in an SGPR pair) and then one v_writelane per word (32 per chain) to put
// them back into lanes for the entry encoding and the stores - or scalar stores of the SGPR words (s_store_dwordx4 assembles for
// gfx950), which only works for the 128-bit cells, not for the compact entries.
//
// One "tile" = 4 chains of 32 PWMs x 32 windows at 12 positions.  Variants, each at 1-4 waves per SIMD:
//   0  dense as the kernel has it: 12 v_mfma_f32_32x32x16_f16 + 64 v_alignbit
//   1  8 v_smfmac_f32_32x32x32_f16 only                      (the matrix time of the sparse form)
//   2  8 smfmac + 64 v_cmp_lt_f32 (signs -> SGPR pairs), the SGPR words folded into a scalar checksum (no way back to lanes)
//   3  8 smfmac + 64 v_cmp + 128 v_writelane (window-major words back in lanes: what the entry encoding needs)
//   4  8 smfmac + 64 v_alignbit (PWM-major window masks: the cheap packing, but the consumer would need the transpose)
//   5  12 dense MFMA only
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o smfmac_pipe smfmac_pipe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ uint32_t pack16(const f32x16& a) {
    uint32_t v = 0;
#pragma unroll
    for (int r = 15; r >= 0; r--) v = __builtin_amdgcn_alignbit(v, __float_as_uint(a[r]), 31);
    return v;
}

template <int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(uint32_t* out, int iters, const uint4* bsrc) {
    const int lane = threadIdx.x & 63;
    f16x8 A[4][3];          // dense: PWM fragments (A operand), 3 k-steps
    f16x16 Bs[4][2];        // sparse: PWM weights as the DENSE B operand, 2 k-steps of 32
    for (int g = 0; g < 4; g++) {
        for (int t = 0; t < 3; t++)
            for (int i = 0; i < 8; i++) A[g][t][i] = (_Float16)(((threadIdx.x + g * 7 + t * 3 + i) & 7) - 3.5f);
        for (int t = 0; t < 2; t++)
            for (int i = 0; i < 16; i++) Bs[g][t][i] = (_Float16)(((threadIdx.x + g * 5 + t * 3 + i) & 7) - 3.5f);
    }
    f16x8 ones;             // sparse A: (1.0, 0) per group of four - a constant
    for (int i = 0; i < 8; i++) ones[i] = (i & 1) ? (_Float16)0.f : (_Float16)1.f;
    f32x16 acc[4];
    for (int g = 0; g < 4; g++)
        for (int r = 0; r < 16; r++) acc[g][r] = __uint_as_float(bsrc[(threadIdx.x + g * 16 + r) & 255].x);
    uint32_t sink = 0;
    unsigned long long ssink = 0;
    f32x16 C;
    for (int r = 0; r < 16; r++) C[r] = -4.0f;
    for (int it = 0; it < iters; it++) {
        const uint4 bw = bsrc[(it * 3 + threadIdx.x) & 255];            // an L1-resident operand stream: window data / index bits
        if (MODE == 0 || MODE == 5) {
            f16x8 B[3];
            for (int t = 0; t < 3; t++) B[t] = __builtin_bit_cast(f16x8, bsrc[(it * 3 + t + threadIdx.x) & 255]);
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], C, 0, 0, 0);
#pragma unroll
            for (int t = 1; t < 3; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc[g], 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int g = 0; g < 4; g++) sink += pack16(acc[g]);
            }
        } else {
            const int idx = (int)bw.x;                                     // the base codes as 2:4 indices, 8 positions per register
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_smfmac_f32_32x32x32_f16(ones, Bs[g][0], C, idx, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_smfmac_f32_32x32x32_f16(ones, Bs[g][1], acc[g], idx, 0, 1);
            if (MODE == 4) {
#pragma unroll
                for (int g = 0; g < 4; g++) sink += pack16(acc[g]);
            }
            if (MODE == 2 || MODE == 3) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    uint32_t word = 0;                                     // MODE 3: lane w ends up with the PWM word of window w
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const unsigned long long m = __ballot(acc[g][r] < 0.f);          // v_cmp_lt_f32 -> SGPR pair: two windows' words
                        if (MODE == 2) {
                            ssink += m;
                        } else {
                            // rows of D: window 8 * (r / 4) + r % 4 + 4 * half
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)m), "n"(8 * (r / 4) + (r % 4)));
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)(m >> 32)), "n"(8 * (r / 4) + (r % 4) + 4));
                        }
                    }
                    if (MODE == 3) sink += word;
                }
            }
        }
        if (MODE != 5 && MODE != 1) {
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g][g] += __uint_as_float((sink + (uint32_t)ssink) & 0x00800000u);   // keep the packing live
        }
    }
    float s = 0;
    for (int g = 0; g < 4; g++)
        for (int r = 0; r < 16; r++) s += acc[g][r];
    out[blockIdx.x * 256 + threadIdx.x] = sink ^ __float_as_uint(s) ^ (uint32_t)ssink ^ (uint32_t)lane;
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
    printf("%-58s %d waves/SIMD  %7.3f ms  %6.1f ns per tile per SIMD\n", name, WPE, ms, ms * 1e6 / iters / WPE);
}

int main() {
    uint32_t* out;
    uint4* bsrc;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMalloc(&bsrc, 256 * 16);
    hipMemset(bsrc, 0x3c, 256 * 16);
#define ALL(M, NAME) run<M, 1>(NAME, out, bsrc); run<M, 2>(NAME, out, bsrc); run<M, 3>(NAME, out, bsrc); run<M, 4>(NAME, out, bsrc);
    ALL(5, "12 dense MFMA 32x32x16 only")
    ALL(1, "8 smfmac 32x32x32 only")
    ALL(0, "dense: 12 MFMA + 64 alignbit (as the kernel)")
    ALL(4, "sparse: 8 smfmac + 64 alignbit (PWM-major masks)")
    ALL(2, "sparse: 8 smfmac + 64 v_cmp (words stay in SGPRs)")
    ALL(3, "sparse: 8 smfmac + 64 v_cmp + 128 v_writelane")
    return 0;
}
