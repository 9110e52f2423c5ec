// scan_kernels.hip — PWM log-odds scan for gfx950 (MI355X).
//
// Replaces the reference's only hand-written GPU kernel, `greedy_search!`
// (src/inference/_h3_1_alignment.jl:18-36), and the dense D2H + CPU `findall`
// that follows it (:81-84).
//
// Design:
//   * lanes = PWMs, two per lane packed as half2, so one v_pk_add_f16 advances
//     128 PWMs; a wave owns ONE sequence at a time, so the base at every
//     position is wave-uniform: it is fetched with scalar loads and the 4-way
//     choice of PWM column is a scalar branch, not a per-lane select;
//   * the wave's slice of the PWM bank (LEN x 4 half2) lives in VGPRs for the
//     whole kernel; LEN rotating accumulators slide along the sequence, so each
//     base is decoded once and feeds LEN adds;
//   * adds happen in the reference's order (ind ascending, one binary16 rounding
//     per add), so scores and the `> 0` hit decision are bit-identical to it;
//   * MODE_DENSE writes the reference's dense (K, N, ld_l) fp16 tensor;
//     MODE_COUNT / MODE_FILL fuse threshold + ordered compaction (count per
//     (sequence, chunk, position) -> transposed exclusive scan -> fill), which
//     yields exactly the record order of the reference's column-major `findall`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "scan_kernels.h"

namespace motifs {

template <int... Is, class F>
static __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
static __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ half2_t as_half2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
static __device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// One packed binary16 add (round-to-nearest-even per half).  Written as inline
// asm on purpose: with plain `+` LLVM sinks the four switch arms into one block
// of adds fed by 12 v_mov per base, doubling the VALU work.
static __device__ __forceinline__ void pk_add(half2_t& acc, const half2_t t) {
    asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc) : "v"(t));
}

// max(x, 0) on both halves; on binary16 bit patterns a signed 16-bit max with 0
// maps every negative value (and -0) to +0 and keeps positives untouched.
static __device__ __forceinline__ uint32_t clamp_pos(half2_t v) {
    short2_t s = __builtin_bit_cast(short2_t, v);
    short2_t z = {0, 0};
    s = __builtin_elementwise_max(s, z);
    return __builtin_bit_cast(uint32_t, s);
}

// Block -> logical block so that blocks sharing an XCD (observed round-robin over
// 8 XCDs) cover a contiguous range of sequences: their writes for one start
// position l then fall into neighbouring lines of one L2.  Speed only.
static __device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nb) {
    const unsigned q = nb / 8, r = nb % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

template <int LEN, int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_kernel(
    const uint32_t* __restrict__ tab, const int32_t* __restrict__ lim, const uint8_t* __restrict__ codes,
    uint16_t* __restrict__ scores, uint4* __restrict__ masks, const ScanDims a) {
    // NOTE: the sequence loop below must stay free of divergent branches (every
    // per-lane condition is a select or an out-of-range buffer offset).  One
    // divergent branch makes LLVM structurize the whole loop, after which the
    // accumulators are shuffled through v_mov at every base.
    constexpr int G = 64 / LEN;                 // blocks of LEN windows per 64-lane flush group
    constexpr int GW = G * LEN;                 // windows per group
    constexpr uint32_t OOR = 0x80000000u;       // buffer offset the range check always drops

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = blockIdx.y * a.cpb + (wave % a.cpb);   // PWM chunk: pairs [ch*64, ch*64+64)
    const int sl = wave / a.cpb;                           // which of the block's parallel sequences
    const int spb = SCAN_WAVES / a.cpb;                    // sequences a block advances per step
    if (ch >= a.nch) return;
    const int kp = ch * 64 + lane;              // this lane's PWM pair
    if (MODE == MODE_DENSE && 2 * kp >= a.K) return;       // absent pair: nothing to write

    // ---- this lane's PWM columns: T[ind][b] = {pwm[2kp][b][ind], pwm[2kp+1][b][ind]} ----
    half2_t T[LEN][4];
#pragma unroll
    for (int ind = 0; ind < LEN; ind++)
#pragma unroll
        for (int b = 0; b < 4; b++) T[ind][b] = as_half2(tab[(ind * 4 + b) * a.KP + kp]);

    // last valid start (0-based) for each half; -1 when the PWM does not exist
    const int lim_lo = lim[2 * kp], lim_hi = lim[2 * kp + 1];
    const uint32_t lane_off = (uint32_t)lane << 2;         // byte offset of this pair inside the chunk's row slice
    const uint32_t row_off_hi = (2 * kp + 1 < a.K) ? lane_off + 2 : OOR;

    const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
    const int nblk = (a.Lout + LEN - 1 + LEN - 1) / LEN;   // blocks of LEN positions covering p < Lout+LEN-1
    const int ngrp = (nblk + G - 1) / G;
    const size_t l_stride = (size_t)a.K * a.N * 2;         // DENSE: bytes between consecutive l planes
    const size_t p_stride = (size_t)a.batch * a.nch;       // MASK: cells between consecutive positions

    for (int s = 0; s < a.spw; s++) {
        const int64_t n = ((int64_t)lb * a.spw + s) * spb + sl;   // wave-uniform
        if (n >= a.N) break;
        const uint32_t* __restrict__ srow = (const uint32_t*)(codes + n * a.pitch);
        const bool row_plain = srow[(a.pitch >> 2) - 1] == 0;      // no all-zero column in this row

        half2_t acc[LEN];
#pragma unroll
        for (int i = 0; i < LEN; i++) acc[i] = half2_t{0, 0};

        // MASK: cell of (batch, p = 0, n, ch) in the (batch, p, n-in-batch, chunk) array
        const int64_t bq = n / a.batch;
        uint4* const mrow = masks + ((size_t)bq * a.LoutP * a.batch + (size_t)(n - bq * a.batch)) * a.nch + ch;
        // DENSE: this chunk's slice of the K-row of (n, l = 0)
        char* const row0 = (char*)scores + ((size_t)a.K * n + (size_t)ch * 128) * 2;

        uint32_t w[LEN / 4];
#pragma unroll
        for (int i = 0; i < LEN / 4; i++) w[i] = srow[i];

        // one block = LEN consecutive positions p0..p0+LEN-1; the window that completes at
        // position p starts at l = p - (LEN-1) and is indexed by p in the mask array
        auto block = [&](auto edge_tag, const int p0, const int gbase, uint4& lanebuf, char*& rp) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            static_for<LEN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr uint32_t B0 = 1u << (8 * (j % 4)), B1 = 2u << (8 * (j % 4)), B2 = 4u << (8 * (j % 4));
                const uint32_t word = w[j / 4];
                // window l = p - ind lives in acc[(j - ind) mod LEN]; the window that starts here
                // (ind = 0) is assigned, which also recycles the slot.  Positions p >= L need no
                // test: they only meet table entries beyond lens[k] (zeros) or windows that the
                // lim mask removes.
#define MOTIFS_ARM(B)                                                  \
    {                                                                  \
        acc[j] = T[0][B];                                              \
        static_for<LEN - 1>([&](auto ic) {                             \
            constexpr int ind = decltype(ic)::value + 1;               \
            pk_add(acc[(j - ind + LEN) % LEN], T[ind][B]);             \
        });                                                            \
    }
                if (EDGE && (word & B2)) {
                    acc[j] = half2_t{0, 0};      // all-zero column: every product is +-0 (:29)
                } else if (word & B1) {
                    if (word & B0) MOTIFS_ARM(3) else MOTIFS_ARM(2)
                } else {
                    if (word & B0) MOTIFS_ARM(1) else MOTIFS_ARM(0)
                }
#undef MOTIFS_ARM
                constexpr int slot = (j + 1) % LEN;
                const int l = p0 + j - (LEN - 1);
                bool emit = true;
                if (EDGE) emit = l >= 0 && l < a.Lout;
                if (emit) {
                    uint32_t v = clamp_pos(acc[slot]);      // :33
                    if (EDGE && l > a.lim_min) {            // some PWMs are too long for this start (:25)
                        asm volatile("");                   // keep this a (uniform) branch, not 6 selects per window
                        uint32_t m = (l <= lim_lo ? 0x0000ffffu : 0u) | (l <= lim_hi ? 0xffff0000u : 0u);
                        v &= m;
                    }
                    if (MODE == MODE_DENSE) {
                        if (a.k_even) {
                            *(uint32_t*)(rp + lane_off) = v;            // uniform base + lane offset
                        } else {
                            auto rs = __builtin_amdgcn_make_buffer_rsrc(rp, 0, 256, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, rs, lane_off, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(v >> 16), rs, row_off_hi, 0, 0);
                        }
                    } else if (__builtin_amdgcn_ballot_w64(v != 0u) != 0ull) {   // wave-uniform
                        // MASK: 128-bit hit mask of this window -> lane (gbase + j) of the group buffer
                        const uint64_t mlo = __builtin_amdgcn_ballot_w64((v & 0xffffu) != 0u);
                        const uint64_t mhi = __builtin_amdgcn_ballot_w64(v > 0xffffu);
                        const bool me = lane == gbase + j;
                        lanebuf.x = me ? (uint32_t)mlo : lanebuf.x;
                        lanebuf.y = me ? (uint32_t)(mlo >> 32) : lanebuf.y;
                        lanebuf.z = me ? (uint32_t)mhi : lanebuf.z;
                        lanebuf.w = me ? (uint32_t)(mhi >> 32) : lanebuf.w;
                    }
                }
                if (MODE == MODE_DENSE) rp += l_stride;
            });
        };

        for (int g = 0; g < ngrp; g++) {
            uint4 lanebuf = {0u, 0u, 0u, 0u};   // MASK: lane i = hit mask of the window completing at g*GW + i
            const int b_hi = (g + 1) * G < nblk ? (g + 1) * G : nblk;
            for (int blk = g * G; blk < b_hi; blk++) {
                const int p0 = blk * LEN;
                uint32_t wn[LEN / 4];           // prefetch the next LEN bases (guard bytes make this safe)
#pragma unroll
                for (int i = 0; i < LEN / 4; i++) wn[i] = srow[(p0 + LEN) / 4 + i];
                char* rp = row0 + (int64_t)(p0 - (LEN - 1)) * (int64_t)l_stride;
                const int gbase = (blk - g * G) * LEN;
                // fast blocks: every window complete, in range and valid for every PWM
                if (row_plain && blk >= 1 && p0 <= a.lim_min)
                    block(std::false_type{}, p0, gbase, lanebuf, rp);
                else
                    block(std::true_type{}, p0, gbase, lanebuf, rp);
#pragma unroll
                for (int i = 0; i < LEN / 4; i++) w[i] = wn[i];
            }
            // all 64 lanes store (positions < LoutP by construction); the lanes beyond GW hold
            // zeros and are rewritten by the next group's store
            if (MODE == MODE_MASK) mrow[((size_t)g * GW + lane) * p_stride] = lanebuf;
        }
    }
}

// ---------------------------------------------------------------------------
// Hit records from the masks.  The mask array is laid out (batch, p, n, chunk),
// which IS the reference's record order (5000-sequence batches, then findall's
// column-major walk: l slowest, then n, then k; _h3_1_alignment.jl:71-84), so
// record offsets are a plain exclusive scan of popcounts in memory order.
// ---------------------------------------------------------------------------

static __device__ __forceinline__ uint32_t cell_pop(const uint4 m) {
    return __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
}

static __device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
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

// F1: hits per mask row (one row = all (n, chunk) cells of one (batch, p)).
__global__ __launch_bounds__(FILL_THREADS) void fill_row_sums(FillArgs a) {
    __shared__ uint32_t red[FILL_THREADS / 64];
    for (int64_t r = blockIdx.x; r < a.nrows; r += gridDim.x) {
        const int p = (int)(r % a.LoutP);
        uint32_t s = 0;
        if (p >= a.lshift && p < a.lshift + a.Lout) {      // other rows hold no window
            const uint4* row = a.masks + r * a.row_cells;
            for (uint32_t i = threadIdx.x; i < a.row_cells; i += FILL_THREADS) s += cell_pop(row[i]);
        }
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int i = 0; i < FILL_THREADS / 64; i++) t += red[i];
            a.row_sum[r] = t;
        }
        __syncthreads();
    }
}

// F2: one block: exclusive scan of the row sums (64-bit), grand total.
__global__ __launch_bounds__(1024) void fill_row_scan(FillArgs a) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (a.nrows + 1023) / 1024;
    int64_t lo = tid * per, hi = lo + per;
    if (lo > a.nrows) lo = a.nrows;
    if (hi > a.nrows) hi = a.nrows;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += a.row_sum[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (int64_t i = lo; i < hi; i++) {
        a.row_base[i] = (int64_t)run;
        run += a.row_sum[i];
    }
    if (tid == 1023) *a.total = (int64_t)part[1023];
}

// F3: expand every set mask bit into a record.  The score of a hit is
// recomputed from the packed bank in the reference's order (ind ascending, one
// binary16 rounding per add), which reproduces the scan kernel's value bit for bit.
template <int LEN, bool LDS_TAB>
__global__ __launch_bounds__(FILL_THREADS) void fill_records(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* wsum = smem;                       // [FILL_THREADS/64]
    uint32_t* hist = smem + 16;                  // [hist_bins]
    uint32_t* ltab = hist + a.hist_bins;         // [LEN*4*KP] when LDS_TAB
    const int tid = threadIdx.x;
    for (int i = tid; i < a.hist_bins; i += FILL_THREADS) hist[i] = 0;
    if (LDS_TAB)
        for (int i = tid; i < LEN * 4 * a.KP; i += FILL_THREADS) ltab[i] = a.tab[i];
    __syncthreads();
    const uint32_t* __restrict__ gtab = a.tab;
    const auto dnch = a.div_nch;

    for (int64_t r = blockIdx.x; r < a.nrows; r += gridDim.x) {
        if (a.row_sum[r] == 0) continue;         // block-uniform
        const int p = (int)(r % a.LoutP);
        const int64_t bq = r / a.LoutP;
        const int l = p - a.lshift;
        const uint4* row = a.masks + r * a.row_cells;
        int64_t run = a.base0 + a.row_base[r];
        for (uint32_t i0 = 0; i0 < a.row_cells; i0 += FILL_THREADS) {
            const uint32_t idx = i0 + tid;
            uint4 m = {0u, 0u, 0u, 0u};
            if (idx < a.row_cells) m = row[idx];
            const uint32_t pc = cell_pop(m);
            uint32_t tot;
            const uint32_t ex = block_excl_scan(pc, wsum, tot);
            int64_t at = run + ex;
            run += tot;
            if (pc) {
                const uint32_t nin = dnch.div(idx);
                const int ch = (int)(idx - nin * a.nch);
                const int64_t n = bq * a.batch + nin;
                // the LEN bases of this window, as table row offsets (ind*4 + b) * KP
                const uint32_t* sw = (const uint32_t*)(a.codes + n * a.pitch + (l & ~3));
                uint32_t W[LEN / 4 + 1];
#pragma unroll
                for (int q = 0; q <= LEN / 4; q++) W[q] = sw[q];
                uint32_t rowofs[LEN];
#pragma unroll
                for (int ind = 0; ind < LEN; ind++) {
                    const uint32_t aligned = __builtin_amdgcn_alignbyte(W[ind / 4 + 1], W[ind / 4], (uint32_t)(l & 3));
                    const uint32_t b = (aligned >> (8 * (ind % 4))) & 0xffu;
                    rowofs[ind] = b < 4 ? (uint32_t)(ind * 4 + b) * a.KP : 0xffffffffu;
                }
                uint64_t mlo = ((uint64_t)m.y << 32) | m.x, mhi = ((uint64_t)m.w << 32) | m.z;
                uint64_t any = mlo | mhi;
                const uint32_t nn = (uint32_t)(n + a.n0 + 1), ll = (uint32_t)(l + 1);
                while (any) {
                    const int bit = __builtin_ctzll(any);
                    any &= any - 1;
                    const uint32_t kp = ch * 64 + bit;
                    uint32_t t[LEN];
#pragma unroll
                    for (int ind = 0; ind < LEN; ind++) {
                        const uint32_t o = rowofs[ind];
                        t[ind] = o == 0xffffffffu ? 0u : (LDS_TAB ? ltab[o + kp] : gtab[o + kp]);
                    }
                    half2_t acc = as_half2(t[0]);    // both halves of the pair; only the hit ones are emitted
#pragma unroll
                    for (int ind = 1; ind < LEN; ind++) acc += as_half2(t[ind]);
                    const uint32_t v = clamp_pos(acc);
                    if ((mlo >> bit) & 1) {
                        a.hits[at] = HitRec{2 * kp + 1, nn, ll};
                        a.hit_scores[at] = (uint16_t)v;
                        at++;
                        if (a.hist_bins) atomicAdd(&hist[2 * kp], 1u);
                    }
                    if ((mhi >> bit) & 1) {
                        a.hits[at] = HitRec{2 * kp + 2, nn, ll};
                        a.hit_scores[at] = (uint16_t)(v >> 16);
                        at++;
                        if (a.hist_bins) atomicAdd(&hist[2 * kp + 1], 1u);
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

// Per-PWM hit histogram straight from the masks (count-only calls, or banks too
// large for the LDS histogram of fill_records).
__global__ __launch_bounds__(FILL_THREADS) void mask_histogram(FillArgs a) {
    const int64_t ncells = a.nrows * a.row_cells;
    for (int64_t cell = (int64_t)blockIdx.x * FILL_THREADS + threadIdx.x; cell < ncells;
         cell += (int64_t)gridDim.x * FILL_THREADS) {
        const uint4 m = a.masks[cell];
        if ((m.x | m.y | m.z | m.w) == 0u) continue;
        const int ch = (int)(cell % a.nch);
        uint64_t mlo = ((uint64_t)m.y << 32) | m.x, mhi = ((uint64_t)m.w << 32) | m.z;
        while (mlo) {
            const int bit = __builtin_ctzll(mlo);
            mlo &= mlo - 1;
            atomicAdd((unsigned long long*)&a.pwm_counts[2 * (ch * 64 + bit)], 1ull);
        }
        while (mhi) {
            const int bit = __builtin_ctzll(mhi);
            mhi &= mhi - 1;
            atomicAdd((unsigned long long*)&a.pwm_counts[2 * (ch * 64 + bit) + 1], 1ull);
        }
    }
}

// ---------------------------------------------------------------------------
// Sequence encoding: one-hot (4L, N) f32 / f16 -> 1 byte per base.
// (input layout: loadfasta/helpers.jl:110-139; fp16 cast: _h3_1_alignment.jl:74)
// ---------------------------------------------------------------------------
template <typename T4>
__device__ __forceinline__ void onehot_decode(const T4 v, const float one, uint32_t& code, bool& bad);

__global__ __launch_bounds__(256) void encode_f32(const float4* __restrict__ x, int64_t N, int L, int pitch,
                                                  uint8_t* __restrict__ codes, int32_t* bad_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // base index n*L + p
    if (i >= N * L) return;
    const int64_t n = i / L;
    const int p = (int)(i - n * L);
    const float4 v = x[i];
    const int ones = (v.x == 1.0f) + (v.y == 1.0f) + (v.z == 1.0f) + (v.w == 1.0f);
    const int zeros = (v.x == 0.0f) + (v.y == 0.0f) + (v.z == 0.0f) + (v.w == 0.0f);
    uint32_t code = 4;
    if (ones == 1 && zeros == 3) code = v.x == 1.0f ? 0 : v.y == 1.0f ? 1 : v.z == 1.0f ? 2 : 3;
    else if (zeros != 4 && bad_flag) *bad_flag = 1;
    codes[n * pitch + p] = (uint8_t)code;
    if (code == 4) codes[n * pitch + pitch - 4] = 1;   // row flag: has an all-zero column
}

__global__ __launch_bounds__(256) void encode_f16(const uint2* __restrict__ x, int64_t N, int L, int pitch,
                                                  uint8_t* __restrict__ codes, int32_t* bad_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * L) return;
    const int64_t n = i / L;
    const int p = (int)(i - n * L);
    const uint2 v = x[i];
    const uint16_t h[4] = {(uint16_t)v.x, (uint16_t)(v.x >> 16), (uint16_t)v.y, (uint16_t)(v.y >> 16)};
    int ones = 0, zeros = 0, which = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (h[q] == 0x3c00u) { ones++; which = q; }
        if ((h[q] & 0x7fffu) == 0) zeros++;
    }
    uint32_t code = 4;
    if (ones == 1 && zeros == 3) code = which;
    else if (zeros != 4 && bad_flag) *bad_flag = 1;
    codes[n * pitch + p] = (uint8_t)code;
    if (code == 4) codes[n * pitch + pitch - 4] = 1;
}

__global__ __launch_bounds__(256) void encode_u8(const uint8_t* __restrict__ x, int64_t N, int L, int pitch,
                                                 uint8_t* __restrict__ codes, int32_t* bad_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * L) return;
    const int64_t n = i / L;
    const int p = (int)(i - n * L);
    const uint8_t c = x[i];
    if (c > 4 && bad_flag) *bad_flag = 1;
    codes[n * pitch + p] = c > 4 ? 4 : c;
    if (c >= 4) codes[n * pitch + pitch - 4] = 1;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int LEN>
static hipError_t launch_len(int mode, const ScanArgs& a, dim3 grid, hipStream_t st) {
    switch (mode) {
        case MODE_DENSE:
            hipLaunchKernelGGL((scan_kernel<LEN, MODE_DENSE>), grid, dim3(SCAN_BLOCK), 0, st, a.tab, a.lim, a.codes, a.scores, a.masks, a.d);
            break;
        case MODE_MASK:
            hipLaunchKernelGGL((scan_kernel<LEN, MODE_MASK>), grid, dim3(SCAN_BLOCK), 0, st, a.tab, a.lim, a.codes, a.scores, a.masks, a.d);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int scan_len_padded(int maxlen) {
    static const int sizes[] = {8, 12, 16, 20, 24, 32, 40, 48, 64};   // > 32: matrix-core path only
    for (int s : sizes)
        if (maxlen <= s) return s;
    return (maxlen + 3) & ~3;                                         // > 64: the run-time-length kernels (scan_mfma.hip)
}

int scan_lout_padded(int Lout, int lenp) {
    const int G = 64 / lenp, GW = G * lenp;
    const int nblk = (Lout + lenp - 1 + lenp - 1) / lenp;
    const int ngrp = (nblk + G - 1) / G;
    return (ngrp - 1) * GW + 64;
}

hipError_t launch_scan(int mode, int len_padded, const ScanArgs& a, hipStream_t st) {
    const int64_t seqs_per_block = (int64_t)(SCAN_WAVES / a.d.cpb) * a.d.spw;
    dim3 grid((unsigned)((a.d.N + seqs_per_block - 1) / seqs_per_block), (unsigned)((a.d.nch + a.d.cpb - 1) / a.d.cpb), 1);
    switch (len_padded) {
        case 8: return launch_len<8>(mode, a, grid, st);
        case 12: return launch_len<12>(mode, a, grid, st);
        case 16: return launch_len<16>(mode, a, grid, st);
        case 20: return launch_len<20>(mode, a, grid, st);
        case 24: return launch_len<24>(mode, a, grid, st);
        case 32: return launch_len<32>(mode, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

static unsigned fill_grid(int64_t nrows) { return (unsigned)std::min<int64_t>(nrows, 256 * 8); }

hipError_t launch_fill_sums(const FillArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(fill_row_sums, dim3(fill_grid(a.nrows)), dim3(FILL_THREADS), 0, st, a);
    hipLaunchKernelGGL(fill_row_scan, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_fill_scan(const FillArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(fill_row_scan, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

template <int LEN>
static hipError_t launch_fill_len(const FillArgs& a, hipStream_t st) {
    const size_t tab_bytes = (size_t)LEN * 4 * a.KP * 4;
    const size_t base = (16 + (size_t)a.hist_bins) * 4;
    const bool lds_tab = base + tab_bytes <= 64 * 1024;
    if (lds_tab)
        hipLaunchKernelGGL((fill_records<LEN, true>), dim3(fill_grid(a.nrows)), dim3(FILL_THREADS), base + tab_bytes, st, a);
    else
        hipLaunchKernelGGL((fill_records<LEN, false>), dim3(fill_grid(a.nrows)), dim3(FILL_THREADS), base, st, a);
    return hipGetLastError();
}

hipError_t launch_fill_records(const FillArgs& a, hipStream_t st) {
    switch (a.lenp) {
        case 8: return launch_fill_len<8>(a, st);
        case 12: return launch_fill_len<12>(a, st);
        case 16: return launch_fill_len<16>(a, st);
        case 20: return launch_fill_len<20>(a, st);
        case 24: return launch_fill_len<24>(a, st);
        case 32: return launch_fill_len<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_mask_histogram(const FillArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(mask_histogram, dim3(256 * 8), dim3(FILL_THREADS), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_encode(int kind, const void* x, int64_t N, int L, int pitch, uint8_t* codes, int32_t* bad,
                         hipStream_t st) {
    const int64_t total = N * L;
    // padding, row flags and the guard behind the last row start out as zero
    hipError_t e = hipMemsetAsync(codes, 0, (size_t)N * pitch + SCAN_GUARD_BYTES, st);
    if (e != hipSuccess) return e;
    if (total == 0) return hipSuccess;
    dim3 g((unsigned)((total + 255) / 256));
    switch (kind) {
        case 0: hipLaunchKernelGGL(encode_u8, g, dim3(256), 0, st, (const uint8_t*)x, N, L, pitch, codes, bad); break;
        case 1: hipLaunchKernelGGL(encode_f32, g, dim3(256), 0, st, (const float4*)x, N, L, pitch, codes, bad); break;
        case 2: hipLaunchKernelGGL(encode_f16, g, dim3(256), 0, st, (const uint2*)x, N, L, pitch, codes, bad); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace motifs
