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

// ---- exact re-scoring in the reference's arithmetic, load-balanced over a wave ---------------------------
// Each lane owns one cell (its mask words and the table row offsets of its window's bases); the set bits of
// the 64 cells are queued in LDS and then taken round-robin by all 64 lanes, so a cell with five candidates
// does not hold up 63 lanes with none.
constexpr uint32_t NOROW = 0xffffffffu;

// table row offsets of the LEN bases of a window from its raw code words (all-zero column: column 4 of the
// 5-column LDS table, or NOROW for the 4-column global table)
template <int LEN, bool LDS_TAB>
static __device__ __forceinline__ void window_offsets(const uint32_t (&W)[LEN / 4 + 1], int l, int KP, uint32_t* dst) {
#pragma unroll
    for (int ind = 0; ind < LEN; ind++) {
        const uint32_t al = __builtin_amdgcn_alignbyte(W[ind / 4 + 1], W[ind / 4], (uint32_t)(l & 3));
        const uint32_t bb = (al >> (8 * (ind % 4))) & 0xffu;
        if (LDS_TAB) dst[ind] = (uint32_t)(ind * 5 + (bb < 4 ? bb : 4)) * KP;
        else dst[ind] = bb < 4 ? (uint32_t)(ind * 4 + bb) * KP : NOROW;
    }
}
template <int LEN>
static __device__ __forceinline__ void fetch_codes(const uint8_t* codes, int64_t n, int pitch, int l, uint32_t (&W)[LEN / 4 + 1]) {
    const uint32_t* sw = (const uint32_t*)(codes + n * pitch + (l & ~3));
#pragma unroll
    for (int q = 0; q <= LEN / 4; q++) W[q] = sw[q];
}
// sequential binary16 sum of PWM k over the owner's window (table entries beyond lens[k] are +0)
template <int LEN, bool LDS_TAB>
static __device__ __forceinline__ uint16_t exact_score(const uint32_t* tb, const uint32_t* rofs, uint32_t k) {
    const uint32_t kp = k >> 1, sh = (k & 1u) * 16;
    uint32_t t[LEN];
#pragma unroll
    for (int ind = 0; ind < LEN; ind++) {
        const uint32_t o = rofs[ind];
        t[ind] = (!LDS_TAB && o == NOROW) ? 0u : tb[o + kp];
    }
    _Float16 acc = __builtin_bit_cast(_Float16, (uint16_t)(t[0] >> sh));
#pragma unroll
    for (int ind = 1; ind < LEN; ind++) acc = acc + __builtin_bit_cast(_Float16, (uint16_t)(t[ind] >> sh));
    return __builtin_bit_cast(uint16_t, acc);
}
static __device__ __forceinline__ bool half_pos(uint16_t h) { return (int16_t)h > 0; }   // > 0 (finite inputs: no NaN)

// the packed bank into LDS; with room, as 5 columns per position (column 4 = zeros for all-zero data columns)
template <int LEN, bool LDS_TAB, int THREADS>
static __device__ __forceinline__ const uint32_t* stage_table(const uint32_t* tab, int KP, uint32_t* ltab) {
    if (!LDS_TAB) return tab;
    for (int i = threadIdx.x; i < LEN * 5 * KP; i += THREADS) {
        const int kp = i % KP, cb = i / KP, b = cb % 5, ind = cb / 5;
        ltab[i] = b < 4 ? tab[(size_t)(ind * 4 + b) * KP + kp] : 0u;
    }
    __syncthreads();
    return ltab;
}

static __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = __shfl_up(v, dd);
        if (lane >= dd) v += t;
    }
    return v;
}
// LDS traffic between lanes of one wave: program order is execution order, the fence stops the compiler
static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Both kernels below give every WAVE its own rows of cells and its own LDS queue: a wave walks a row 64 cells
// at a time, pushes every set bit into the queue, and scores the queue 64 candidates at a time with all lanes
// busy (a cell with five candidates does not stall the 63 lanes beside it; cells without candidates cost a
// load and a popcount).  No block barriers after the table is staged, so the waves of a CU hide each other's
// load latency.
constexpr int VF_THREADS = 512;
constexpr int VF_WAVES = VF_THREADS / 64;
constexpr int QN = 256;           // ring slots per wave (power of two); fewer than 64 stay behind after a push round

// the candidates of one 64-cell slab into the wave's FIFO ring, draining full batches of 64 through
// fn(candidate word, ordinal of the candidate in the row).  head = candidates of this row drained so far.
// candidate word = cell index in the row << 7 | word << 5 | bit
template <typename F>
static __device__ __forceinline__ void push_and_drain(uint32_t* queue, uint32_t& head, uint32_t& qlen, const uint32_t (&wd)[4],
                                                      uint32_t idx, uint32_t ex, uint32_t tot, F&& fn) {
    const int lane = threadIdx.x & 63;
    uint32_t done = 0;
    while (true) {                                                    // wave-uniform
        const uint32_t room = QN - qlen;
        const uint32_t take = tot - done < room ? tot - done : room;
        uint32_t g = ex;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t bits = wd[q];
            while (bits) {
                const int i = __builtin_ctz(bits);
                bits &= bits - 1;
                if (g >= done && g < done + take)
                    queue[(head + qlen + g - done) & (QN - 1)] = (idx << 7) | ((uint32_t)q << 5) | (uint32_t)i;
                g++;
            }
        }
        qlen += take;
        done += take;
        wave_lds_sync();
        while (qlen >= 64) {
            fn(queue[(head + lane) & (QN - 1)], head + lane);
            head += 64;
            qlen -= 64;
        }
        wave_lds_sync();
        if (done >= tot) break;
    }
}

// V1: candidates -> hits, in place, plus hits per cell row (one row = the (n, chunk) cells of one (batch, l, part)).
template <int LEN, bool LDS_TAB>
__global__ __launch_bounds__(VF_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void fill_verify_row_sums(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* queue = smem + wv * QN;                                 // [VF_WAVES][QN]
    uint32_t* ltab = smem + VF_WAVES * QN;
    const uint32_t* tb = stage_table<LEN, LDS_TAB, VF_THREADS>(a.tab, a.KP, ltab);
    uint32_t* cells = (uint32_t*)const_cast<uint4*>(a.masks);
    const uint32_t part_reads = (uint32_t)(a.batch / a.parts);
    const int64_t nwaves = (int64_t)gridDim.x * VF_WAVES;
    for (int64_t r = (int64_t)wv * gridDim.x + blockIdx.x; r < a.nrows; r += nwaves) {
        const int part = (int)(r % a.parts);
        const int64_t rl = r / a.parts;
        const int l = (int)(rl % a.LoutP);
        const int64_t bq = rl / a.LoutP;
        uint32_t* row = cells + (size_t)r * a.row_cells * 4;
        const bool all_valid = l <= a.lim_min;                        // every PWM fits at this start
        const int64_t nrow0 = bq * a.batch + (int64_t)part * part_reads;
        uint32_t qlen = 0, head = 0, nhit = 0;                        // wave-uniform
        auto score = [&](const uint32_t cw) {                         // one candidate per lane
            const uint32_t idx = cw >> 7, q = (cw >> 5) & 3u, i = cw & 31u;
            const uint32_t nin = a.div_nch.div(idx);
            const uint32_t ch = idx - nin * a.nch;
            const int64_t n = nrow0 + nin;
            const uint32_t k = (ch * 4 + q) * 32 + i;
            bool hit = false;
            if (n < a.N && (int)k < a.K && (all_valid || l <= a.lim[k])) {
                uint32_t W[LEN / 4 + 1], rofs[LEN];
                fetch_codes<LEN>(a.codes, n, a.pitch, l, W);
                window_offsets<LEN, LDS_TAB>(W, l, a.KP, rofs);
                hit = half_pos(exact_score<LEN, LDS_TAB>(tb, rofs, k));
            }
            if (!hit) atomicAnd(&row[idx * 4 + q], ~(1u << i));
            return hit;
        };
        auto score_count = [&](const uint32_t cw, uint32_t) { nhit += (uint32_t)__builtin_popcountll(__ballot(score(cw))); };
        uint4 m_next = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)lane < a.row_cells) m_next = ((const uint4*)row)[lane];
        for (uint32_t i0 = 0; i0 < a.row_cells; i0 += 64) {           // wave-uniform trip count
            const uint32_t idx = i0 + lane;
            const uint4 m = m_next;
            m_next = make_uint4(0u, 0u, 0u, 0u);
            if (idx + 64 < a.row_cells) m_next = ((const uint4*)row)[idx + 64];
            const uint32_t wd[4] = {m.x, m.y, m.z, m.w};
            const uint32_t pc = __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
            const uint32_t inc = wave_incl_scan(pc, lane);
            const uint32_t tot = __shfl(inc, 63);
            if (tot) push_and_drain(queue, head, qlen, wd, idx, inc - pc, tot, score_count);
        }
        {                                                             // the remainder (< 64)
            bool hit = false;
            if ((uint32_t)lane < qlen) hit = score(queue[(head + lane) & (QN - 1)]);
            nhit += (uint32_t)__builtin_popcountll(__ballot(hit));
        }
        wave_lds_sync();
        if (lane == 0) a.row_sum[r] = nhit;
    }
}

// V2: every (verified) bit becomes a record: (m, n, l) 1-based + the fp16 score, reference order.
template <int LEN, bool LDS_TAB>
__global__ __launch_bounds__(VF_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void fill_records_plain(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* queue = smem + wv * QN;                                 // [VF_WAVES][QN]
    uint32_t* hist = smem + VF_WAVES * QN;                            // [hist_bins]
    uint32_t* ltab = hist + a.hist_bins;
    for (int i = tid; i < a.hist_bins; i += VF_THREADS) hist[i] = 0;
    const uint32_t* tb = stage_table<LEN, LDS_TAB, VF_THREADS>(a.tab, a.KP, ltab);
    __syncthreads();
    const uint32_t* cells = (const uint32_t*)a.masks;
    const uint32_t part_reads = (uint32_t)(a.batch / a.parts);
    const int64_t nwaves = (int64_t)gridDim.x * VF_WAVES;
    for (int64_t r = (int64_t)wv * gridDim.x + blockIdx.x; r < a.nrows; r += nwaves) {
        if (a.row_sum[r] == 0) continue;                              // wave-uniform
        const int part = (int)(r % a.parts);
        const int64_t rl = r / a.parts;
        const int l = (int)(rl % a.LoutP);
        const int64_t bq = rl / a.LoutP;
        const uint4* row = (const uint4*)(cells + (size_t)r * a.row_cells * 4);
        const int64_t row_at = a.base0 + a.row_base[r];
        const int64_t nrow0 = bq * a.batch + (int64_t)part * part_reads;
        uint32_t qlen = 0, head = 0;                                  // wave-uniform
        auto emit = [&](const uint32_t cw, const uint32_t ord) {      // ord-th bit of the row = ord-th record
            const uint32_t idx = cw >> 7, q = (cw >> 5) & 3u, i = cw & 31u;
            const uint32_t nin = a.div_nch.div(idx);
            const uint32_t ch = idx - nin * a.nch;
            const int64_t n = nrow0 + nin;
            const uint32_t k = (ch * 4 + q) * 32 + i;
            uint32_t W[LEN / 4 + 1], rofs[LEN];
            fetch_codes<LEN>(a.codes, n, a.pitch, l, W);
            window_offsets<LEN, LDS_TAB>(W, l, a.KP, rofs);
            const int64_t at = row_at + ord;
            a.hits[at] = HitRec{k + 1, (uint32_t)(n + a.n0 + 1), (uint32_t)(l + 1)};
            a.hit_scores[at] = exact_score<LEN, LDS_TAB>(tb, rofs, k);
            if (a.hist_bins) atomicAdd(&hist[k], 1u);
        };
        uint4 m_next = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)lane < a.row_cells) m_next = row[lane];
        for (uint32_t i0 = 0; i0 < a.row_cells; i0 += 64) {
            const uint32_t idx = i0 + lane;
            const uint4 m = m_next;
            m_next = make_uint4(0u, 0u, 0u, 0u);
            if (idx + 64 < a.row_cells) m_next = row[idx + 64];
            const uint32_t wd[4] = {m.x, m.y, m.z, m.w};
            const uint32_t pc = __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
            const uint32_t inc = wave_incl_scan(pc, lane);            // hits up to and with this cell inside the slab
            const uint32_t tot = __shfl(inc, 63);
            if (tot) push_and_drain(queue, head, qlen, wd, idx, inc - pc, tot, emit);
        }
        if ((uint32_t)lane < qlen) emit(queue[(head + lane) & (QN - 1)], head + lane);
        wave_lds_sync();
    }
    if (a.hist_bins) {
        __syncthreads();
        for (int i = tid; i < a.hist_bins; i += VF_THREADS)
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

static unsigned fill_grid2(int64_t nrows) { return (unsigned)std::min<int64_t>((nrows + VF_WAVES - 1) / VF_WAVES, 256 * 4); }

template <int LEN>
static hipError_t launch_verify_len(const FillArgs& a, hipStream_t st) {
    const size_t base = (size_t)VF_WAVES * QN * 4;
    const size_t tab_bytes = (size_t)LEN * 5 * a.KP * 4;
    if (base + tab_bytes <= 64 * 1024)
        hipLaunchKernelGGL((fill_verify_row_sums<LEN, true>), dim3(fill_grid2(a.nrows)), dim3(VF_THREADS), base + tab_bytes, st, a);
    else
        hipLaunchKernelGGL((fill_verify_row_sums<LEN, false>), dim3(fill_grid2(a.nrows)), dim3(VF_THREADS), base, st, a);
    return hipGetLastError();
}
template <int LEN>
static hipError_t launch_records_len(const FillArgs& a, hipStream_t st) {
    const size_t base = (size_t)VF_WAVES * QN * 4 + (size_t)a.hist_bins * 4;
    const size_t tab_bytes = (size_t)LEN * 5 * a.KP * 4;
    if (base + tab_bytes <= 64 * 1024)
        hipLaunchKernelGGL((fill_records_plain<LEN, true>), dim3(fill_grid2(a.nrows)), dim3(VF_THREADS), base + tab_bytes, st, a);
    else
        hipLaunchKernelGGL((fill_records_plain<LEN, false>), dim3(fill_grid2(a.nrows)), dim3(VF_THREADS), base, st, a);
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
