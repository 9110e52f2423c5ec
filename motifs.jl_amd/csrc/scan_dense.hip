// scan_dense.hip — a17's dense score tensor (greedy_search!, src/inference/_h3_1_alignment.jl:18-36, 75-80) in ONE kernel.
//
// The two-kernel form (scan_cand_kernel -> cells -> stage_hits<.., 2>) writes the tensor at the rate of a bare fill, but the
// candidate filter runs BEFORE the writer, on a chip the writer then leaves idle: 0.074 + 0.28 ms per 24 576-read launch, 0.53-0.62
// of the 8 TB/s HBM peak across boxes, under north_star's 0.60 on the slower ones for three rounds.  Every way of running the two
// side by side (streams, CU masks, a block that keeps 16 reads' cells in LDS) lost.  Here the filter runs INSIDE the
// write stream: the 32 columns of an MFMA tile are 32 consecutive READS at one start l (not 8 windows of 4 reads), so the
// accumulator tiles of a wave - all PWMs x 32 reads at one l - cover one CONTIGUOUS span of the (K, N, ld_l) tensor, 32 * K
// halves (12.8 KB at K = 200).  The wave tests the signs, re-scores its candidates (~50 per span at a 0.8 % candidate rate) in the
// reference's binary16 arithmetic from LDS tables, drops the positive scores into a zeroed LDS window of the span's size and streams
// the window out as whole 16-byte stores.  The tensor is written exactly once, in 12.8 KB pieces, while other waves' matrix
// instructions run: the kernel is bound by the HBM write stream alone.
//
// A block = 4 waves = 32 reads: their one-hot images (the B operands; 8 bytes per position, images 8 bytes mod 256 apart so that the
// 32 lanes of a half wave cover the 64 banks once), their code rows (for the exact re-scoring), the binary16 table of the bank, and
// one window per wave; wave w takes the starts l = w, w + 4, ...  One block per CU (140 KB of LDS at BASELINE configs[1]): each
// wave has ~4000 cycles per span before HBM is the limit, and needs ~2500.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "scan_kernels.h"

namespace motifs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DF_READS = 32, DF_WAVES = 4, DF_MAXT = 8;      // reads per block, waves per block, tiles of 32 PWMs a wave may carry

static __device__ __forceinline__ void df_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static __device__ __forceinline__ bool df_half_pos(uint16_t h) { return (int16_t)h > 0 && h <= 0x7c00u; }

// sequential binary16 sum of a PWM's table row over the window whose code words are W (reference order, one rounding per add)
template <int LEN>
static __device__ __forceinline__ uint16_t df_exact_score(const _Float16* row, const uint32_t (&W)[LEN / 4 + 1], int l) {
    uint32_t al[LEN / 4];
#pragma unroll
    for (int j = 0; j < LEN / 4; j++) al[j] = __builtin_amdgcn_alignbyte(W[j + 1], W[j], (uint32_t)(l & 3));
    _Float16 t[LEN];
#pragma unroll
    for (int ind = 0; ind < LEN; ind++) t[ind] = row[ind * 5 + ((al[ind / 4] >> (8 * (ind % 4))) & 0xffu)];
    __builtin_amdgcn_sched_barrier(0);
    _Float16 acc = t[0];
#pragma unroll
    for (int ind = 1; ind < LEN; ind++) acc = acc + t[ind];
    return __builtin_bit_cast(uint16_t, acc);
}

constexpr int DF_QCAP = 512;          // candidate queue entries per wave (a span of 32 reads x 200 PWMs holds ~50 candidates)

// inclusive prefix sum over the 64 lanes with DPP moves
static __device__ __forceinline__ uint32_t df_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// NT = tiles of 32 PWMs the wave carries (>= the bank's: tiles past it hold zero fragments and are masked off)
template <int T, int NT>
__global__ __launch_bounds__(64 * DF_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void scan_dense_fused(const DenseFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int LEN = 4 * T;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // LDS: [images: 32 x opitch x 8 B][code rows: 32 x cpitch B][table: K x stride halves][queues: 4 x DF_QCAP x 2 B][windows: 4 x 32 x K halves]
    uint2* oh = (uint2*)smem;
    uint8_t* crow = (uint8_t*)(oh + (size_t)DF_READS * a.opitch);
    _Float16* tb = (_Float16*)(crow + (size_t)DF_READS * a.cpitch);
    uint16_t* qbase = (uint16_t*)(tb + (((size_t)a.K * a.tabk_stride + 7) & ~(size_t)7));
    uint16_t* queue = qbase + (size_t)wv * DF_QCAP;
    uint16_t* win0 = qbase + (size_t)DF_WAVES * DF_QCAP;
    uint16_t* win = win0 + (size_t)wv * DF_READS * a.K;
    const int64_t n0 = (int64_t)blockIdx.x * DF_READS;
    const int nvalid = (int)(a.N - n0 < DF_READS ? a.N - n0 : DF_READS);

    // ---- stage: images + code rows of the block's reads, the table, zeroed windows
    for (int rr = wv; rr < DF_READS; rr += DF_WAVES) {
        const bool row = rr < nvalid;
        const uint32_t* srow = (const uint32_t*)(a.codes + (n0 + rr) * a.pitch);
        const int pend = a.opitch > a.cpitch ? a.opitch : a.cpitch;    // positions to lay out: the image's and the code row's (padding = code 4)
        for (int p4 = lane; p4 * 4 < pend; p4 += 64) {
            const int keep = a.L - p4 * 4;
            uint32_t wd = 0x04040404u;
            if (row && keep > 0) {
                wd = srow[p4];
                if (keep < 4) {
                    const uint32_t mk = (1u << (8 * keep)) - 1u;
                    wd = (wd & mk) | (0x04040404u & ~mk);
                }
            }
            if (p4 * 4 < a.cpitch) *(uint32_t*)(crow + (size_t)rr * a.cpitch + p4 * 4) = wd;
            const uint32_t sh = wd << 4;             // 16 * code per byte; code 4 -> shift 63: the 1.0 leaves the word
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t su = (sh >> (8 * u)) & 0xffu;
                const uint64_t one = (uint64_t)0x3c00u << (su < 63u ? su : 63u);
                if (p4 * 4 + u < a.opitch) oh[(size_t)rr * a.opitch + p4 * 4 + u] = make_uint2((uint32_t)one, (uint32_t)(one >> 32));
            }
        }
    }
    {
        const int ndw = (a.K * a.tabk_stride + 1) / 2;
        const uint32_t* src = (const uint32_t*)a.tabk;
        uint32_t* dst = (uint32_t*)tb;
        for (int i = tid; i < ndw; i += 64 * DF_WAVES) dst[i] = src[i];
        uint32_t* wz = (uint32_t*)win0;
        for (int i = tid; i < DF_WAVES * DF_READS * a.K / 2; i += 64 * DF_WAVES) wz[i] = 0u;
    }
    // the wave's PWM fragments (A operand), constant for the whole block
    f16x8 A[NT][T];
#pragma unroll
    for (int g = 0; g < NT; g++)
#pragma unroll
        for (int t = 0; t < T; t++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (g < a.ntiles) v = a.afrag[((size_t)g * T + t) * 64 + lane];
            A[g][t] = __builtin_bit_cast(f16x8, v);
        }
    // PWMs 32 g + 16 h + r that exist, as a mask over r; lanes of reads past N take no candidates at all
    uint32_t vm[NT];
#pragma unroll
    for (int g = 0; g < NT; g++) {
        const int k0 = 32 * g + 16 * h, left = a.K - k0;
        vm[g] = (j < nvalid && left > 0) ? (left >= 16 ? 0xffffu : (1u << left) - 1u) : 0u;
    }
    __syncthreads();

    const uint2* ohl = oh + (size_t)j * a.opitch + 2 * h;
    const uint32_t span = (uint32_t)nvalid * (uint32_t)a.K;          // halves of the tensor this block owns per start l
    // The slack constants of the chains, -4 / 2^(g % 4) (pack_mfma scales tile g by 2^(e - g % 4)), as registers that live through the
    // loop: with more than four chains the constants repeat, the compiler then shares one splat between two chains and, short of
    // inline-constant slots, rebuilt every splat inside the loop (16 v_readlane + 8 v_mov_b64 each, ~100 instructions per span).
    f32x16 cs[NT < 4 ? NT : 4];
#pragma unroll
    for (int q = 0; q < (NT < 4 ? NT : 4); q++) {
        const float cv = q == 0 ? -4.0f : q == 1 ? -2.0f : q == 2 ? -1.0f : -0.5f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float x = cv;
            asm volatile("" : "+v"(x));                                // opaque: a value in a VGPR, not a constant to re-materialise
            cs[q][r] = x;
        }
    }
    const int l_lo = blockIdx.y * a.l_per_block, l_hi = l_lo + a.l_per_block < a.Lout ? l_lo + a.l_per_block : a.Lout;
    for (int l = l_lo + wv; l < l_hi; l += DF_WAVES) {
        f16x8 B[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint2 a0 = ohl[l + 4 * t], a1 = ohl[l + 4 * t + 1];
            B[t] = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
        }
        // all chains of the wave interleaved (branch-free): the matrix pipe works on one tile while the VALU packs another's signs
        f32x16 acc[NT];
#pragma unroll
        for (int g = 0; g < NT; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], cs[g & 3], 0, 0, 0);
#pragma unroll
        for (int t = 1; t < T; t++)
#pragma unroll
            for (int g = 0; g < NT; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc[g], 0, 0, 0);
        uint32_t m[NT];
        uint32_t pc = 0;
#pragma unroll
        for (int g = 0; g < NT; g++) {
            uint32_t mm = 0;
#pragma unroll
            for (int r = 15; r >= 0; r--) mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(acc[g][r]), 31);
            m[g] = mm & vm[g];
            pc += (uint32_t)__builtin_popcount(m[g]);
        }
        // candidates -> a queue in LDS (read j << 8 | PWM), so that the exact scores are formed 64 at a time with every lane busy
        const uint32_t inc = df_incl_scan(pc);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);     // wave-uniform
        const bool dirty = tot != 0;
        if (dirty) {
            const bool fits = tot <= (uint32_t)DF_QCAP;
            if (fits) {
                // one loop for all tiles: a lane pops the lowest bit of its 32 * NT-bit mask per turn (turns = the most candidates any
                // lane holds, ~3; a loop per tile ran as many turns per TILE)
                uint16_t* qp = queue + (inc - pc);
                uint32_t x[(NT + 1) / 2];
#pragma unroll
                for (int q = 0; q < (NT + 1) / 2; q++) x[q] = m[2 * q] | ((2 * q + 1 < NT ? m[2 * q + 1 < NT ? 2 * q + 1 : 0] : 0u) << 16);
                uint32_t left = pc;
                while (__ballot(left != 0)) {
                    if (left) {
                        int kk = 0;
                        bool done = false;
#pragma unroll
                        for (int q = 0; q < (NT + 1) / 2; q++) {
                            if (!done && x[q]) {
                                const int b = __builtin_ctz(x[q]);
                                x[q] &= x[q] - 1;
                                kk = 64 * q + 32 * (b >> 4) + 16 * h + (b & 15);
                                done = true;
                            }
                        }
                        *qp++ = (uint16_t)((j << 8) | kk);
                        left--;
                    }
                }
                df_wave_sync();
                for (uint32_t b0 = 0; b0 < tot; b0 += 64) {
                    const uint32_t cw = b0 + lane < tot ? queue[b0 + lane] : 0xffffu;
                    const int jj = (int)(cw >> 8), k = (int)(cw & 255u);
                    if (cw != 0xffffu && (l <= a.lim_min || l <= a.lim[k])) {
                        uint32_t W[LEN / 4 + 1];
                        const uint32_t* sw = (const uint32_t*)(crow + (size_t)jj * a.cpitch + (l & ~3));
#pragma unroll
                        for (int q = 0; q <= LEN / 4; q++) W[q] = sw[q];
                        const uint16_t sc = df_exact_score<LEN>(tb + (size_t)k * a.tabk_stride, W, l);
                        if (df_half_pos(sc)) win[jj * a.K + k] = sc;
                    }
                }
            } else {                                                  // a span with more candidates than the queue holds: lane by lane
                uint32_t W[LEN / 4 + 1];
                const uint32_t* sw = (const uint32_t*)(crow + (size_t)j * a.cpitch + (l & ~3));
#pragma unroll
                for (int q = 0; q <= LEN / 4; q++) W[q] = sw[q];
#pragma unroll
                for (int g = 0; g < NT; g++) {
                    uint32_t xv = m[g];
                    while (__ballot(xv != 0)) {
                        if (xv) {
                            const int k = 32 * g + 16 * h + __builtin_ctz(xv);
                            xv &= xv - 1;
                            if (l <= a.lim_min || l <= a.lim[k]) {
                                const uint16_t sc = df_exact_score<LEN>(tb + (size_t)k * a.tabk_stride, W, l);
                                if (df_half_pos(sc)) win[j * a.K + k] = sc;
                            }
                        }
                    }
                }
            }
            df_wave_sync();
        }
        // the span leaves as whole 16-byte stores (K % 8 == 0), zeros where nothing was dropped; every window read is issued before the
        // first store
        {
            uint16_t* dst = a.out + ((size_t)l * a.N + n0) * a.K;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            constexpr int NV = DF_READS * 32 * NT / 8 / 64;           // 16-byte pieces per lane of the largest span this instance serves (32 reads x 32 NT PWMs)
            uint4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i = (uint32_t)(u * 64 + lane) * 8;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (dirty && i < span) v[u] = *(const uint4*)(win + i);
            }
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i = (uint32_t)(u * 64 + lane) * 8;
                if (i < span) __builtin_nontemporal_store(u32x4{v[u].x, v[u].y, v[u].z, v[u].w}, (u32x4*)(dst + i));
            }
        }
        if (dirty) {                                                  // back to zeros: whatever the candidates may have written
            df_wave_sync();
            if (tot <= (uint32_t)DF_QCAP) {
                for (uint32_t b0 = 0; b0 < tot; b0 += 64)
                    if (b0 + lane < tot) {
                        const uint32_t cw = queue[b0 + lane];
                        win[(cw >> 8) * a.K + (cw & 255u)] = 0;
                    }
            } else {
#pragma unroll
                for (int g = 0; g < NT; g++) {
                    uint32_t xv = m[g];
                    while (xv) {
                        win[j * a.K + 32 * g + 16 * h + __builtin_ctz(xv)] = 0;
                        xv &= xv - 1;
                    }
                }
            }
            df_wave_sync();
        }
    }
}

size_t dense_fused_lds(const DenseFusedArgs& a) {
    return (size_t)DF_READS * a.opitch * 8 + (size_t)DF_READS * a.cpitch + ((((size_t)a.K * a.tabk_stride + 7) & ~(size_t)7) * 2) +
           (size_t)DF_WAVES * DF_QCAP * 2 + (size_t)DF_WAVES * DF_READS * a.K * 2;
}

// the geometry fields the launch derives (ohlen, opitch, cpitch); false if the bank / reads cannot take this kernel
bool dense_fused_plan(DenseFusedArgs& a, int lenp, int uniform_eps) {
    if (!uniform_eps || lenp > 20 || lenp % 4 != 0 || a.K % 8 != 0 || a.ntiles > DF_MAXT || a.K > 32 * DF_MAXT) return false;
    a.ohlen = a.Lout + lenp;                                     // positions a window tile may touch: l + 4 t + 2 h + 1 <= Lout + lenp - 2
    a.opitch = ((a.ohlen - 1 + 31) & ~31) + 1;                   // 1 (mod 32) positions = 8 (mod 256) bytes: 32 lanes x 8 B cover the 64 banks once
    int cp = ((a.L + 3) & ~3) + 4 * ((lenp / 4) + 1);            // the exact re-scoring reads lenp / 4 + 1 dwords from the dword of l
    if ((cp / 4) % 2 == 0) cp += 4;                              // an odd number of dwords: the 32 rows start in 32 different banks
    a.cpitch = cp;
    return dense_fused_lds(a) <= 160 * 1024 - 2048;
}

hipError_t launch_dense_fused(const DenseFusedArgs& a0, int lenp, hipStream_t st) {
    DenseFusedArgs a = a0;
    const size_t lds = dense_fused_lds(a);
    const unsigned gx = (unsigned)((a.N + DF_READS - 1) / DF_READS);
    // one block per CU at a time: split the starts over grid.y until the launch is at least ~4 rounds of blocks (a block's set-up - 32
    // images, the table - is ~5 % of a whole read's worth of spans, so halves and quarters are still cheap), so that the last round
    // is a small share of the launch
    int ly = 1;
    while (ly < 8 && (int64_t)gx * ly < 4 * 256 && a.Lout / (ly * 2) >= 4 * DF_WAVES) ly *= 2;
    a.l_per_block = ((a.Lout + ly - 1) / ly + DF_WAVES - 1) / DF_WAVES * DF_WAVES;
    ly = (a.Lout + a.l_per_block - 1) / a.l_per_block;
    const dim3 grid(gx, (unsigned)ly, 1);
#define DF_LAUNCH(TT, NN)                                                                                                            \
    {                                                                                                                                \
        (void)hipFuncSetAttribute((const void*)scan_dense_fused<TT, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL((scan_dense_fused<TT, NN>), grid, dim3(64 * DF_WAVES), lds, st, a);                                       \
    }
#define DF_TILES(TT)                                        \
    if (a.ntiles <= 2) DF_LAUNCH(TT, 2)                     \
    else if (a.ntiles <= 4) DF_LAUNCH(TT, 4)                \
    else if (a.ntiles <= 7) DF_LAUNCH(TT, 7)                \
    else DF_LAUNCH(TT, 8)
    switch (lenp) {
        case 8: DF_TILES(2) break;
        case 12: DF_TILES(3) break;
        case 16: DF_TILES(4) break;
        case 20: DF_TILES(5) break;
        default: return hipErrorInvalidValue;
    }
#undef DF_TILES
#undef DF_LAUNCH
    return hipGetLastError();
}

}  // namespace motifs
