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

constexpr int DF_READS = 32, DF_MAXT = 8;      // reads per block, tiles of 32 PWMs a wave may carry (waves per block: 8, or 4 when the LDS is short)

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

// NT = tiles of 32 PWMs the wave carries (>= the bank's: tiles past it hold zero fragments and are masked off); NW = waves per block.
// The B operand (the one-hot of two positions: 8 binary16 values) comes from a 25-entry table in LDS indexed by the two base codes, which
// the lane takes from its read's code row: one-hot IMAGES of the 32 reads (57 KB) would leave room for only one 4-wave block per CU, and
// with one wave per SIMD nothing hides the LDS round trips of a span (measured: 37 % of the wave cycles waiting); without them 8 waves
// share the code rows and the bank's table, two per SIMD.
template <int T, int NT, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void scan_dense_fused(const DenseFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int LEN = 4 * T;
    constexpr int DF_WAVES = NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // LDS: [pair table: 25 x 16 B][code rows: 32 x cpitch B][table: K x stride halves][queues: NW x DF_QCAP x 2 B][windows: NW x 32 x K halves]
    uint4* lut = (uint4*)smem;
    uint8_t* crow = (uint8_t*)(lut + 32);
    _Float16* tb = (_Float16*)(crow + (size_t)DF_READS * a.cpitch);
    uint16_t* qbase = (uint16_t*)(tb + (((size_t)a.K * a.tabk_stride + 7) & ~(size_t)7));
    uint16_t* queue = qbase + (size_t)wv * DF_QCAP;
    uint16_t* win0 = qbase + (size_t)DF_WAVES * DF_QCAP;
    uint16_t* win = win0 + (size_t)wv * DF_READS * a.K;
    const int64_t n0 = (int64_t)blockIdx.x * DF_READS;
    const int nvalid = (int)(a.N - n0 < DF_READS ? a.N - n0 : DF_READS);

    // ---- stage: the code rows of the block's reads (padding = code 4), the pair table, the bank's table, zeroed windows
    for (int rr = wv; rr < DF_READS; rr += DF_WAVES) {
        const bool row = rr < nvalid;
        const uint32_t* srow = (const uint32_t*)(a.codes + (n0 + rr) * a.pitch);
        for (int p4 = lane; p4 * 4 < a.cpitch; p4 += 64) {
            const int keep = a.L - p4 * 4;
            uint32_t wd = 0x04040404u;
            if (row && keep > 0) {
                wd = srow[p4];
                if (keep < 4) {
                    const uint32_t mk = (1u << (8 * keep)) - 1u;
                    wd = (wd & mk) | (0x04040404u & ~mk);
                }
            }
            *(uint32_t*)(crow + (size_t)rr * a.cpitch + p4 * 4) = wd;
        }
    }
    if (tid < 25) {                                                    // entry c0 + 5 c1: the one-hot columns of codes c0, c1 (code 4: all zero)
        const int c0 = tid % 5, c1 = tid / 5;
        const uint64_t o0 = c0 < 4 ? (uint64_t)0x3c00u << (16 * c0) : 0ull, o1 = c1 < 4 ? (uint64_t)0x3c00u << (16 * c1) : 0ull;
        lut[tid] = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)o1, (uint32_t)(o1 >> 32));
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

    const uint8_t* cj = crow + (size_t)j * a.cpitch;
    const uint32_t span = (uint32_t)nvalid * (uint32_t)a.K;          // halves of the tensor this block owns per start l
    // The slack constants -4, -2, -1 of the chains as registers that live through the loop (-0.5, used by fewer chains, stays an inline
    // constant): more than four chains share constants, and a shared splat the compiler parks in lanes of a spare register and rebuilds
    // for every span (16 v_readlane + 8 v_mov_b64 each: ~70 instructions per span, seen in the ISA).
    f32x16 cs[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float x = q == 0 ? -4.0f : q == 1 ? -2.0f : -1.0f;
            asm volatile("" : "+v"(x));
            cs[q][r] = x;
        }
    }
    const int l_lo = blockIdx.y * a.l_per_block, l_hi = l_lo + a.l_per_block < a.Lout ? l_lo + a.l_per_block : a.Lout;
    for (int l = l_lo + wv; l < l_hi; l += DF_WAVES) {
        // B operand: lane (read j, half h) needs the one-hot columns of positions l + 4 t + 2 h, + 1 of its read
        f16x8 B[T];
        {
            uint32_t Wj[T + 1];
            const uint32_t* sw = (const uint32_t*)(cj + (l & ~3));
#pragma unroll
            for (int q = 0; q <= T; q++) Wj[q] = sw[q];
#pragma unroll
            for (int t = 0; t < T; t++) {
                const uint32_t al = __builtin_amdgcn_alignbyte(Wj[t + 1], Wj[t], (uint32_t)(l & 3)) >> (16 * h);   // codes of l + 4 t + 2 h, + 1 in the low bytes
                const uint32_t e = (al & 0xffu) + 5u * ((al >> 8) & 0xffu);
                B[t] = __builtin_bit_cast(f16x8, lut[e]);
            }
        }
        // The chains go in groups of four, interleaved inside a group (branch-free): the matrix pipe works on one tile while the VALU packs
        // another's signs.  Four at a time because each chain of a group then has a slack constant of its own, -4 / 2^(g % 4) (pack_mfma
        // scales tile g by 2^(e - g % 4)), which the instruction takes as an INLINE constant; and 64 accumulator registers live, not 32 NT.
        uint32_t m[NT];
        uint32_t pc = 0;
#pragma unroll
        for (int g0 = 0; g0 < NT; g0 += 4) {
            constexpr int dummy = 0;
            (void)dummy;
            f32x16 acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (g0 + u < NT) {
                    if (u < 3) {
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g0 + u][0], B[0], cs[u], 0, 0, 0);
                    } else {
                        f32x16 c;
#pragma unroll
                        for (int r = 0; r < 16; r++) c[r] = -0.5f;
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g0 + u][0], B[0], c, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 1; t < T; t++)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (g0 + u < NT) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g0 + u][t], B[t], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (g0 + u < NT) {
                    uint32_t mm = 0;
#pragma unroll
                    for (int r = 15; r >= 0; r--) mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(acc[u][r]), 31);
                    m[g0 + u] = mm & vm[g0 + u];
                    pc += (uint32_t)__builtin_popcount(m[g0 + u]);
                }
            }
            if (g0 + 4 < NT) __builtin_amdgcn_sched_barrier(0);       // (keeps the second group's constants from being shared with the first's)
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
            // (the window is read whether or not anything was dropped into it - it is all zeros then: no selects in this loop; pieces past the
            // span - a short last block, the unrolling's tail - are read from the window's own first bytes and not stored)
            uint4 v[NV];
            const uint32_t i0 = (uint32_t)lane * 8;
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i = i0 + (uint32_t)u * 512;
                v[u] = *(const uint4*)(win + (i < span ? i : i0));
            }
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i = i0 + (uint32_t)u * 512;
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

static size_t dense_fused_lds_w(const DenseFusedArgs& a, int nw) {
    return 32 * 16 + (size_t)DF_READS * a.cpitch + ((((size_t)a.K * a.tabk_stride + 7) & ~(size_t)7) * 2) + (size_t)nw * DF_QCAP * 2 +
           (size_t)nw * DF_READS * a.K * 2;
}
size_t dense_fused_lds(const DenseFusedArgs& a) { return dense_fused_lds_w(a, a.nwaves); }

// the geometry fields the launch derives (cpitch, waves per block); false if the bank / reads cannot take this kernel
bool dense_fused_plan(DenseFusedArgs& a, int lenp, int uniform_eps) {
    if (!uniform_eps || lenp > 20 || lenp % 4 != 0 || a.K % 8 != 0 || a.ntiles > DF_MAXT || a.K > 32 * DF_MAXT) return false;
    a.ohlen = a.Lout + lenp;
    a.opitch = 0;
    int cp = ((a.L + 3) & ~3) + 4 * ((lenp / 4) + 1);            // the operand build and the exact re-scoring read lenp / 4 + 1 dwords from the dword of l
    if ((cp / 4) % 2 == 0) cp += 4;                              // an odd number of dwords: the 32 rows start in 32 different banks
    a.cpitch = cp;
    for (int nw : {8, 4}) {
        if (nw == 8 && a.ntiles > 4 && lenp >= 16) continue;       // 7-8 tiles of 16-20 positions: the fragments do not fit 256 VGPRs (two waves per SIMD)
        a.nwaves = nw;
        if (dense_fused_lds_w(a, nw) <= 160 * 1024 - 1024) return true;
    }
    return false;
}

hipError_t launch_dense_fused(const DenseFusedArgs& a0, int lenp, hipStream_t st) {
    DenseFusedArgs a = a0;
    const size_t lds = dense_fused_lds(a);
    const unsigned gx = (unsigned)((a.N + DF_READS - 1) / DF_READS);
    // one block per CU at a time: split the starts over grid.y until the launch is at least ~4 rounds of blocks (a block's set-up - the
    // code rows, the table - is a few per cent of a whole read's worth of spans), so that the last round is a small share of the launch
    int ly = 1;
    while (ly < 8 && (int64_t)gx * ly < 4 * 256 && a.Lout / (ly * 2) >= 4 * a.nwaves) ly *= 2;
    a.l_per_block = ((a.Lout + ly - 1) / ly + a.nwaves - 1) / a.nwaves * a.nwaves;
    ly = (a.Lout + a.l_per_block - 1) / a.l_per_block;
    const dim3 grid(gx, (unsigned)ly, 1);
#define DF_LAUNCH(TT, NN, WW)                                                                                                            \
    {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)scan_dense_fused<TT, NN, WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL((scan_dense_fused<TT, NN, WW>), grid, dim3(64 * WW), lds, st, a);                                             \
    }
#define DF_WAVESEL(TT, NN)                  \
    if (a.nwaves == 8) DF_LAUNCH(TT, NN, 8) \
    else DF_LAUNCH(TT, NN, 4)
#define DF_TILES(TT)                                         \
    if (a.ntiles <= 2) DF_WAVESEL(TT, 2)                     \
    else if (a.ntiles <= 4) DF_WAVESEL(TT, 4)                \
    else if (a.ntiles <= 7) DF_WAVESEL(TT, 7)                \
    else DF_WAVESEL(TT, 8)
    switch (lenp) {
        case 8: DF_TILES(2) break;
        case 12: DF_TILES(3) break;
        case 16: DF_TILES(4) break;
        case 20: DF_TILES(5) break;
        default: return hipErrorInvalidValue;
    }
#undef DF_TILES
#undef DF_WAVESEL
#undef DF_LAUNCH
    return hipGetLastError();
}

}  // namespace motifs
