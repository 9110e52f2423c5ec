// scan_mfma.hip — the hit-record path of the PWM scan on the matrix cores.
//
// A PWM score is a 4*len-long dot product of the PWM with the one-hot window: a GEMM
// [PWMs x 4*len] x [4*len x windows].  v_mfma_f32_32x32x16_f16 forms it exactly in f32 (products of
// binary16 weights with 0/1 are exact), which is NOT the reference's arithmetic — greedy_search!
// (src/inference/_h3_1_alignment.jl:26-31) adds in binary16, rounding after every add.  So the matrix
// cores only FILTER: with S the exact sum and s the sequentially rounded one,
//     |s - S| <= eps_k := 2^-10 * len_k * A_k,   A_k = sum_ind max_a |pwm[k, a, ind]|
// (each of the len adds rounds by at most 2^-11 of a partial sum bounded by A_k; the factor 2 covers
// second-order terms and binary16 subnormals), hence s > 0 implies S > -eps_k.  Every (PWM, window)
// with S > -eps_k becomes a candidate bit; candidates (about 1 % of the pairs) are then re-scored in the
// reference's arithmetic and only true hits survive (fill_verify_row_sums), so records and scores stay
// bit-identical to the reference while 99 % of the pairs never touch the slow fp16 chain.
//
// Cells: uint4 per (batch, l, n-in-batch, chunk of 128 PWMs), bit i = PWM 128*chunk + i.  This order is
// the reference's record order (5000-read batches, then findall's column-major walk), so record
// offsets are a plain exclusive scan of popcounts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "scan_kernels.h"

namespace motifs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ unsigned xcd_swz(unsigned b, unsigned nb) {
    const unsigned q = nb / 8, r = nb % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

// One wave = one read at a time x PG PWM tiles of 32.  D[row = PWM][col = window]:
//   A operand (rows): the wave's PWM tiles, constant, in registers (afrag, packed on the host so that the
//                     PWM with local index q = 16h + r sits in the MFMA row that accumulator r of lane half
//                     h reports);
//   B operand (cols): 32 consecutive windows; lane (w, h) needs the one-hot of positions l0+w+4t+2h, +1,
//                     read as two 8-byte LDS words from the read's one-hot image;
//   C = eps_k per PWM row, so that the sign bit of the result is "not a candidate".
template <int T, int PG>
__global__ __launch_bounds__(512) void scan_cand_kernel(const uint4* __restrict__ afrag, const float* __restrict__ cinit,
                                                        const uint8_t* __restrict__ codes, uint32_t* __restrict__ cells,
                                                        const CandDims d) {
    extern __shared__ uint2 oh_all[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = lane & 31, h = lane >> 5;
    const int swap_addr = (lane ^ 32) << 2;          // ds_bpermute byte address of the partner lane (other half)
    uint2* oh = oh_all + (size_t)wave * ((d.ohlen + 3) & ~3);
    // 8 waves = 4 reads x 2 tile groups: the two halves of a 128-byte line of cells (4 reads x 2 chunks) are
    // written by the same block at about the same time
    const int slot = wave & 3;
    const int tg = blockIdx.y * 2 + (wave >> 2);     // tile group: PWM tiles [tg*PG, tg*PG + PG)
    if (tg * PG >= d.used_tiles) return;
    const int tile0 = tg * PG;
    const int chunk = tile0 >> 2, word0 = tile0 & 3;

    f16x8 A[PG][T];
    f32x16 C0[PG];
#pragma unroll
    for (int g = 0; g < PG; g++) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint4 v = afrag[((size_t)(tile0 + g) * T + t) * 64 + lane];
            A[g][t] = __builtin_bit_cast(f16x8, v);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) C0[g][r] = cinit[((size_t)(tile0 + g) * 2 + h) * 16 + r];
    }

    const unsigned lb = xcd_swz(blockIdx.x, gridDim.x);
    const int ntile = (d.Lout + 31) / 32;
    for (int s = 0; s < d.spw; s++) {
        const int64_t n = ((int64_t)lb * d.spw + s) * 4 + slot;        // wave-uniform
        if (n >= d.N) break;
        // stage the read's one-hot image: 4 halves per position (1.0 at the base, all zero for code 4 / padding);
        // one dword (4 bases) per lane and round
        const uint32_t* srow = (const uint32_t*)(codes + n * d.pitch);
        for (int p4 = lane; p4 * 4 < d.ohlen; p4 += 64) {
            const uint32_t wv = p4 * 4 < d.L ? srow[p4] : 0x04040404u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = p4 * 4 + u;
                const uint32_t c = p < d.L ? (wv >> (8 * u)) & 0xffu : 4u;
                uint2 v = make_uint2(0u, 0u);
                if (c < 2) v.x = 0x3c00u << (16 * c);
                else if (c < 4) v.y = 0x3c00u << (16 * (c - 2));
                if (p < d.ohlen) oh[p] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t bq = n / d.batch;
        const size_t cell0 = ((size_t)bq * d.Lout * d.batch + (size_t)(n - bq * d.batch)) * d.nch + chunk;   // l = 0
        const size_t lstride = (size_t)d.batch * d.nch;
        for (int wt = 0; wt < ntile; wt++) {
            const int l0 = wt * 32;
            f16x8 B[T];
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int pos = l0 + w + 4 * t + 2 * h;
                const uint2 a0 = oh[pos], a1 = oh[pos + 1];
                B[t] = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
            }
            uint32_t word[PG];
#pragma unroll
            for (int g = 0; g < PG; g++) {
                word[g] = 0;
                if (tile0 + g < d.used_tiles) {                        // wave-uniform: tiles past K hold no PWM
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], C0[g], 0, 0, 0);
#pragma unroll
                    for (int t = 1; t < T; t++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc, 0, 0, 0);
                    uint32_t m = 0;           // bit r = sign of accumulator r (set = below -eps = no candidate)
#pragma unroll
                    for (int r = 15; r >= 0; r--) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[r]), 31);
                    m = ~m & 0xffffu;
                    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute(swap_addr, (int)m);   // the other 16 PWMs
                    word[g] = h ? (other | (m << 16)) : (m | (other << 16));
                }
            }
            const int l = l0 + w;
            if (h == 0 && l < d.Lout) {
                uint32_t* cp = cells + (cell0 + (size_t)l * lstride) * 4 + word0;
                if (PG == 4) *(uint4*)cp = make_uint4(word[0], word[1 % PG], word[2 % PG], word[3 % PG]);
                else if (PG == 2) *(uint2*)cp = make_uint2(word[0], word[1 % PG]);
                else cp[0] = word[0];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- exact re-scoring of one (PWM, window) in the reference's arithmetic --------------------------------
template <int LEN>
struct WindowBases {
    uint32_t rowofs[LEN];     // (ind*4 + b) * KP, or ~0 for an all-zero column
    // raw words: issued for every cell BEFORE its mask is known, so the two loads overlap
    static __device__ __forceinline__ void fetch(const uint8_t* codes, int64_t n, int pitch, int l, uint32_t (&W)[LEN / 4 + 1]) {
        const uint32_t* sw = (const uint32_t*)(codes + n * pitch + (l & ~3));
#pragma unroll
        for (int q = 0; q <= LEN / 4; q++) W[q] = sw[q];
    }
    __device__ __forceinline__ void decode(const uint32_t (&W)[LEN / 4 + 1], int l, int KP) {
#pragma unroll
        for (int ind = 0; ind < LEN; ind++) {
            const uint32_t al = __builtin_amdgcn_alignbyte(W[ind / 4 + 1], W[ind / 4], (uint32_t)(l & 3));
            const uint32_t b = (al >> (8 * (ind % 4))) & 0xffu;
            rowofs[ind] = b < 4 ? (uint32_t)(ind * 4 + b) * KP : 0xffffffffu;
        }
    }
    __device__ __forceinline__ void load(const uint8_t* codes, int64_t n, int pitch, int l, int KP) {
        uint32_t W[LEN / 4 + 1];
        fetch(codes, n, pitch, l, W);
        decode(W, l, KP);
    }
    // sequential binary16 sum of PWM k over the window (zero-padded table: entries beyond lens[k] add +0)
    __device__ __forceinline__ uint16_t score(const uint32_t* tab, uint32_t k) const {
        const uint32_t kp = k >> 1, sh = (k & 1u) * 16;
        uint32_t t[LEN];
#pragma unroll
        for (int ind = 0; ind < LEN; ind++) t[ind] = rowofs[ind] == 0xffffffffu ? 0u : tab[rowofs[ind] + kp];
        _Float16 acc = __builtin_bit_cast(_Float16, (uint16_t)(t[0] >> sh));
#pragma unroll
        for (int ind = 1; ind < LEN; ind++) acc = acc + __builtin_bit_cast(_Float16, (uint16_t)(t[ind] >> sh));
        return __builtin_bit_cast(uint16_t, acc);
    }
};
static __device__ __forceinline__ bool half_pos(uint16_t h) { return (int16_t)h > 0; }   // > 0 (finite inputs: no NaN)

// V1: candidates -> hits, in place, plus hits per cell row (one row = all (n, chunk) cells of one (batch, l)).
template <int LEN, bool LDS_TAB>
__global__ __launch_bounds__(FILL_THREADS) void fill_verify_row_sums(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* red = smem;                        // [FILL_THREADS/64]
    uint32_t* ltab = smem + 16;
    const int tid = threadIdx.x;
    if (LDS_TAB)
        for (int i = tid; i < LEN * 4 * a.KP; i += FILL_THREADS) ltab[i] = a.tab[i];
    __syncthreads();
    const uint32_t* tb = LDS_TAB ? ltab : a.tab;
    uint4* cells = const_cast<uint4*>(a.masks);
    for (int64_t r = blockIdx.x; r < a.nrows; r += gridDim.x) {
        const int l = (int)(r % a.LoutP);
        const int64_t bq = r / a.LoutP;
        uint4* row = cells + r * a.row_cells;
        const bool all_valid = l <= a.lim_min;   // every PWM fits at this start
        uint32_t s = 0;
        for (uint32_t idx = tid; idx < a.row_cells; idx += FILL_THREADS) {
            const uint32_t nin = a.div_nch.div(idx);
            const int ch = (int)(idx - nin * a.nch);
            const int64_t n = bq * a.batch + nin;
            uint32_t W[LEN / 4 + 1];
            WindowBases<LEN>::fetch(a.codes, n < a.N ? n : 0, a.pitch, l, W);
            uint4 m = row[idx];
            if ((m.x | m.y | m.z | m.w) == 0u) continue;
            if (n >= a.N) {                      // the tail of the last batch was never scanned
                row[idx] = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            WindowBases<LEN> wb;
            wb.decode(W, l, a.KP);
            uint32_t wd[4] = {m.x, m.y, m.z, m.w};
            bool changed = false;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t bits = wd[q];
                while (bits) {
                    const int i = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const uint32_t k = (uint32_t)(ch * 4 + q) * 32 + i;
                    const bool hit = (int)k < a.K && (all_valid || l <= a.lim[k]) && half_pos(wb.score(tb, k));
                    if (!hit) {
                        wd[q] &= ~(1u << i);
                        changed = true;
                    }
                }
                s += __builtin_popcount(wd[q]);
            }
            if (changed) row[idx] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
        for (int dd = 32; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int i = 0; i < FILL_THREADS / 64; i++) t += red[i];
            a.row_sum[r] = t;
        }
        __syncthreads();
    }
}

static __device__ __forceinline__ uint32_t excl_scan_256(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = __shfl_up(inc, dd);
        if (lane >= dd) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < FILL_THREADS / 64; q++) {
        if (q < wv) wbase += wsum[q];
        tot += wsum[q];
    }
    __syncthreads();
    total = tot;
    return wbase + inc - v;
}

// V2: every (verified) bit becomes a record: (m, n, l) 1-based + the fp16 score, reference order.
template <int LEN, bool LDS_TAB>
__global__ __launch_bounds__(FILL_THREADS) void fill_records_plain(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* wsum = smem;                       // [FILL_THREADS/64]
    uint32_t* hist = smem + 16;                  // [hist_bins]
    uint32_t* ltab = hist + a.hist_bins;
    const int tid = threadIdx.x;
    for (int i = tid; i < a.hist_bins; i += FILL_THREADS) hist[i] = 0;
    if (LDS_TAB)
        for (int i = tid; i < LEN * 4 * a.KP; i += FILL_THREADS) ltab[i] = a.tab[i];
    __syncthreads();
    const uint32_t* tb = LDS_TAB ? ltab : a.tab;
    for (int64_t r = blockIdx.x; r < a.nrows; r += gridDim.x) {
        if (a.row_sum[r] == 0) continue;         // block-uniform
        const int l = (int)(r % a.LoutP);
        const int64_t bq = r / a.LoutP;
        const uint4* row = a.masks + r * a.row_cells;
        int64_t run = a.base0 + a.row_base[r];
        for (uint32_t i0 = 0; i0 < a.row_cells; i0 += FILL_THREADS) {
            const uint32_t idx = i0 + tid;
            const uint32_t idc = idx < a.row_cells ? idx : a.row_cells - 1;
            const uint32_t nin = a.div_nch.div(idc);
            const int ch = (int)(idc - nin * a.nch);
            const int64_t n = bq * a.batch + nin;
            uint32_t W[LEN / 4 + 1];
            WindowBases<LEN>::fetch(a.codes, n < a.N ? n : 0, a.pitch, l, W);
            uint4 m = row[idc];
            if (idx >= a.row_cells) m = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t pc = __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
            uint32_t tot;
            const uint32_t ex = excl_scan_256(pc, wsum, tot);
            int64_t at = run + ex;
            run += tot;
            if (pc) {
                WindowBases<LEN> wb;
                wb.decode(W, l, a.KP);
                const uint32_t nn = (uint32_t)(n + a.n0 + 1), ll = (uint32_t)(l + 1);
                const uint32_t wd[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t bits = wd[q];
                    while (bits) {
                        const int i = __builtin_ctz(bits);
                        bits &= bits - 1;
                        const uint32_t k = (uint32_t)(ch * 4 + q) * 32 + i;
                        a.hits[at] = HitRec{k + 1, nn, ll};
                        a.hit_scores[at] = wb.score(tb, k);
                        at++;
                        if (a.hist_bins) atomicAdd(&hist[k], 1u);
                    }
                }
            }
        }
    }
    if (a.hist_bins) {
        __syncthreads();
        for (int i = tid; i < a.hist_bins; i += FILL_THREADS)
            if (hist[i]) atomicAdd((unsigned long long*)&a.pwm_counts[i], (unsigned long long)hist[i]);
    }
}

__global__ __launch_bounds__(FILL_THREADS) void cell_histogram(FillArgs a) {
    const int64_t ncells = a.nrows * a.row_cells;
    for (int64_t cell = (int64_t)blockIdx.x * FILL_THREADS + threadIdx.x; cell < ncells; cell += (int64_t)gridDim.x * FILL_THREADS) {
        const uint4 m = a.masks[cell];
        if ((m.x | m.y | m.z | m.w) == 0u) continue;
        const int ch = (int)(cell % a.nch);
        const uint32_t wd[4] = {m.x, m.y, m.z, m.w};
        for (int q = 0; q < 4; q++) {
            uint32_t bits = wd[q];
            while (bits) {
                const int i = __builtin_ctz(bits);
                bits &= bits - 1;
                atomicAdd((unsigned long long*)&a.pwm_counts[(ch * 4 + q) * 32 + i], 1ull);
            }
        }
    }
}

// ---- launchers -------------------------------------------------------------------------------------------
template <int T, int PG>
static hipError_t launch_cand_tp(const CandArgs& a, hipStream_t st) {
    const int64_t per_block = (int64_t)4 * a.d.spw;
    const int ntg = a.ntiles / PG;
    dim3 grid((unsigned)((a.d.N + per_block - 1) / per_block), (unsigned)((ntg + 1) / 2), 1);
    hipLaunchKernelGGL((scan_cand_kernel<T, PG>), grid, dim3(512), (size_t)8 * ((a.d.ohlen + 3) & ~3) * 8, st, a.afrag, a.cinit, a.codes, a.cells,
                       a.d);
    return hipGetLastError();
}

int cand_tile_group(int lenp) { return lenp <= 20 ? 4 : 2; }

hipError_t launch_cand(const CandArgs& a, hipStream_t st) {
    switch (a.lenp) {
        case 8: return launch_cand_tp<2, 4>(a, st);
        case 12: return launch_cand_tp<3, 4>(a, st);
        case 16: return launch_cand_tp<4, 4>(a, st);
        case 20: return launch_cand_tp<5, 4>(a, st);
        case 24: return launch_cand_tp<6, 2>(a, st);
        case 32: return launch_cand_tp<8, 2>(a, st);
        default: return hipErrorInvalidValue;
    }
}

static unsigned fill_grid2(int64_t nrows) { return (unsigned)std::min<int64_t>(nrows, 256 * 8); }

template <int LEN>
static hipError_t launch_verify_len(const FillArgs& a, hipStream_t st) {
    const size_t tab_bytes = (size_t)LEN * 4 * a.KP * 4;
    const bool lds_tab = 64 + tab_bytes <= 64 * 1024;
    if (lds_tab)
        hipLaunchKernelGGL((fill_verify_row_sums<LEN, true>), dim3(fill_grid2(a.nrows)), dim3(FILL_THREADS), 64 + tab_bytes, st, a);
    else
        hipLaunchKernelGGL((fill_verify_row_sums<LEN, false>), dim3(fill_grid2(a.nrows)), dim3(FILL_THREADS), 64, st, a);
    return hipGetLastError();
}
template <int LEN>
static hipError_t launch_records_len(const FillArgs& a, hipStream_t st) {
    const size_t tab_bytes = (size_t)LEN * 4 * a.KP * 4;
    const size_t base = (16 + (size_t)a.hist_bins) * 4;
    const bool lds_tab = base + tab_bytes <= 64 * 1024;
    if (lds_tab)
        hipLaunchKernelGGL((fill_records_plain<LEN, true>), dim3(fill_grid2(a.nrows)), dim3(FILL_THREADS), base + tab_bytes, st, a);
    else
        hipLaunchKernelGGL((fill_records_plain<LEN, false>), dim3(fill_grid2(a.nrows)), dim3(FILL_THREADS), base, st, a);
    return hipGetLastError();
}

#define MOTIFS_LEN_SWITCH(fn)                           \
    switch (a.lenp) {                                   \
        case 8: return fn<8>(a, st);                    \
        case 12: return fn<12>(a, st);                  \
        case 16: return fn<16>(a, st);                  \
        case 20: return fn<20>(a, st);                  \
        case 24: return fn<24>(a, st);                  \
        case 32: return fn<32>(a, st);                  \
        default: return hipErrorInvalidValue;           \
    }
hipError_t launch_verify_row_sums(const FillArgs& a, hipStream_t st) { MOTIFS_LEN_SWITCH(launch_verify_len) }
hipError_t launch_fill_records_plain(const FillArgs& a, hipStream_t st) { MOTIFS_LEN_SWITCH(launch_records_len) }
hipError_t launch_cell_histogram(const FillArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(cell_histogram, dim3(256 * 8), dim3(FILL_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace motifs
