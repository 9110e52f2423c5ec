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

template <int LEN, int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_kernel(
    const uint32_t* __restrict__ tab, const int32_t* __restrict__ lim, const uint8_t* __restrict__ codes,
    uint16_t* __restrict__ scores, uint16_t* __restrict__ cnt, const uint32_t* __restrict__ off,
    const int64_t* __restrict__ batch_base, HitRec* __restrict__ hits, uint16_t* __restrict__ hit_scores,
    int64_t* __restrict__ pwm_counts, const ScanDims a) {
    // NOTE: the sequence loop below must stay free of divergent branches (every
    // per-lane condition is a select or an out-of-range buffer offset).  One
    // divergent branch makes LLVM structurize the whole loop, after which the
    // accumulators are shuffled through v_mov at every base.
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = blockIdx.y;                  // PWM chunk: pairs [ch*64, ch*64+64)
    const int kp = ch * 64 + lane;              // this lane's PWM pair

    // ---- this lane's PWM columns: T[ind][b] = {pwm[2kp][b][ind], pwm[2kp+1][b][ind]} ----
    half2_t T[LEN][4];
#pragma unroll
    for (int ind = 0; ind < LEN; ind++)
#pragma unroll
        for (int b = 0; b < 4; b++) T[ind][b] = as_half2(tab[(ind * 4 + b) * a.KP + kp]);

    // last valid start (0-based) for each half; -1 when the PWM does not exist
    const int lim_lo = lim[2 * kp], lim_hi = lim[2 * kp + 1];
    constexpr uint32_t OOR = 0x80000000u;       // buffer offset the range check always drops
    // DENSE: byte offset of this lane's pair inside a K-row (dropped when the pair is absent)
    const uint32_t row_off_pair = (2 * kp < a.K) ? (uint32_t)kp << 2 : OOR;
    const uint32_t row_off_lo = (2 * kp < a.K) ? (uint32_t)kp << 2 : OOR;
    const uint32_t row_off_hi = (2 * kp + 1 < a.K) ? ((uint32_t)kp << 2) + 2 : OOR;

    uint32_t pwm_cnt_lo = 0, pwm_cnt_hi = 0;    // MODE_COUNT: per-PWM hit histogram

    const int64_t n_first = ((int64_t)blockIdx.x * SCAN_WAVES + wave) * a.spw;
    for (int s = 0; s < a.spw; s++) {
        const int64_t n = n_first + s;          // wave-uniform
        if (n >= a.N) break;
        const uint32_t* __restrict__ srow = (const uint32_t*)(codes + n * a.pitch);

        half2_t acc[LEN];
#pragma unroll
        for (int i = 0; i < LEN; i++) acc[i] = half2_t{0, 0};

        uint32_t lanebuf = 0;                   // COUNT: count of window (l&63); FILL: its offset
        const size_t rowbase = ((size_t)n * a.nch + ch) * a.LoutP;
        if (MODE == MODE_FILL) lanebuf = off[rowbase + lane];
        int64_t bbase = 0;
        if (MODE == MODE_FILL) bbase = batch_base[n / a.batch];

        // DENSE: K-row of (n, l = 0); consecutive l are l_stride bytes apart
        char* const row = (char*)scores + (size_t)a.K * n * 2;
        const size_t l_stride = (size_t)a.K * a.N * 2;

        const int P_end = a.Lout + LEN - 1;     // positions that feed some emitted window
        uint32_t w[LEN / 4];
#pragma unroll
        for (int i = 0; i < LEN / 4; i++) w[i] = srow[i];

        for (int p0 = 0; p0 < P_end; p0 += LEN) {
            uint32_t wn[LEN / 4];               // prefetch next LEN bases (guard bytes make this safe)
#pragma unroll
            for (int i = 0; i < LEN / 4; i++) wn[i] = srow[(p0 + LEN) / 4 + i];

            static_for<LEN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int p = p0 + j;
                uint32_t b = (w[j / 4] >> (8 * (j % 4))) & 0xffu;
                b = (p < a.L) ? b : 4u;
                // window l = p - ind lives in acc[(j - ind) mod LEN]; the window that
                // starts here (ind = 0) is assigned, which also recycles the slot
                switch (b) {
                    case 0:
                        acc[j] = T[0][0];
                        static_for<LEN - 1>([&](auto ic) {
                            constexpr int ind = decltype(ic)::value + 1;
                            pk_add(acc[(j - ind + LEN) % LEN], T[ind][0]);
                        });
                        break;
                    case 1:
                        acc[j] = T[0][1];
                        static_for<LEN - 1>([&](auto ic) {
                            constexpr int ind = decltype(ic)::value + 1;
                            pk_add(acc[(j - ind + LEN) % LEN], T[ind][1]);
                        });
                        break;
                    case 2:
                        acc[j] = T[0][2];
                        static_for<LEN - 1>([&](auto ic) {
                            constexpr int ind = decltype(ic)::value + 1;
                            pk_add(acc[(j - ind + LEN) % LEN], T[ind][2]);
                        });
                        break;
                    case 3:
                        acc[j] = T[0][3];
                        static_for<LEN - 1>([&](auto ic) {
                            constexpr int ind = decltype(ic)::value + 1;
                            pk_add(acc[(j - ind + LEN) % LEN], T[ind][3]);
                        });
                        break;
                    default:
                        acc[j] = half2_t{0, 0};  // all-zero column: every product is +-0 (:29)
                        break;
                }
                // window l = p - LEN + 1 is complete
                const int l = p - (LEN - 1);
                constexpr int slot = (j + 1) % LEN;
                if (l >= 0 && l < a.Lout) {
                    uint32_t v = clamp_pos(acc[slot]);      // :33
                    if (l > a.lim_min) {                    // some PWMs are too long for this start (:25)
                        asm volatile("");                   // keep this a (uniform) branch, not 6 selects per window
                        uint32_t m = (l <= lim_lo ? 0x0000ffffu : 0u) | (l <= lim_hi ? 0xffff0000u : 0u);
                        v &= m;
                    }
                    if (MODE == MODE_DENSE) {
                        // one descriptor per K-row: wave-uniform base, per-lane offset, absent PWMs dropped
                        auto rs = __builtin_amdgcn_make_buffer_rsrc(row + (size_t)l * l_stride, 0, a.K * 2, 0x00020000);
                        if (a.k_even) {
                            __builtin_amdgcn_raw_buffer_store_b32(v, rs, row_off_pair, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, rs, row_off_lo, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(v >> 16), rs, row_off_hi, 0, 0);
                        }
                    } else {
                        const bool any = v != 0u;
                        if (MODE == MODE_COUNT) {
                            uint32_t c = 0;
                            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // rare, wave-uniform
                                const bool hlo = (v & 0xffffu) != 0u, hhi = (v >> 16) != 0u;
                                c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(hlo)) +
                                    __builtin_popcountll(__builtin_amdgcn_ballot_w64(hhi));
                                pwm_cnt_lo += hlo;
                                pwm_cnt_hi += hhi;
                            }
                            lanebuf = (lane == (l & 63)) ? c : lanebuf;
                            if ((l & 63) == 63 || l == a.Lout - 1) {
                                cnt[rowbase + (l & ~63) + lane] = (uint16_t)lanebuf;
                                lanebuf = 0;
                            }
                        } else {  // MODE_FILL
                            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                                const bool hlo = (v & 0xffffu) != 0u, hhi = (v >> 16) != 0u;
                                const uint64_t mlo = __builtin_amdgcn_ballot_w64(hlo);
                                const uint64_t mhi = __builtin_amdgcn_ballot_w64(hhi);
                                // rank in k order: every lane below contributes both of its halves
                                const uint32_t rank_lo = __builtin_amdgcn_mbcnt_hi((uint32_t)(mlo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mlo, 0)) +
                                                         __builtin_amdgcn_mbcnt_hi((uint32_t)(mhi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mhi, 0));
                                const uint32_t rank_hi = rank_lo + (hlo ? 1u : 0u);
                                const uint32_t o = __builtin_amdgcn_readlane(lanebuf, l & 63);
                                const int64_t base = bbase + o;   // wave-uniform
                                const uint32_t nh = __builtin_popcountll(mlo) + __builtin_popcountll(mhi);
                                auto rh = __builtin_amdgcn_make_buffer_rsrc((char*)(hits + base), 0, nh * 12, 0x00020000);
                                auto rsc = __builtin_amdgcn_make_buffer_rsrc((char*)(hit_scores + base), 0, nh * 2, 0x00020000);
                                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                                const uint32_t nn = (uint32_t)(n + a.n0 + 1), ll = (uint32_t)(l + 1);
                                __builtin_amdgcn_raw_buffer_store_b96(u32x3{(uint32_t)(2 * kp + 1), nn, ll}, rh, hlo ? rank_lo * 12u : OOR, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, rsc, hlo ? rank_lo * 2u : OOR, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b96(u32x3{(uint32_t)(2 * kp + 2), nn, ll}, rh, hhi ? rank_hi * 12u : OOR, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(v >> 16), rsc, hhi ? rank_hi * 2u : OOR, 0, 0);
                            }
                            if ((l & 63) == 63 && l + 1 < a.Lout) lanebuf = off[rowbase + l + 1 + lane];
                        }
                    }
                }
            });
#pragma unroll
            for (int i = 0; i < LEN / 4; i++) w[i] = wn[i];
        }
    }

    if (MODE == MODE_COUNT && pwm_counts != nullptr) {
        if (pwm_cnt_lo) atomicAdd((unsigned long long*)&pwm_counts[2 * kp], (unsigned long long)pwm_cnt_lo);
        if (pwm_cnt_hi) atomicAdd((unsigned long long*)&pwm_counts[2 * kp + 1], (unsigned long long)pwm_cnt_hi);
    }
}

// ---------------------------------------------------------------------------
// Offsets for the reference's record order.  cnt[(n, ch, l)] -> off[(n, ch, l)]
// = exclusive prefix over (batch, l, n, ch) relative to the batch start.
// ---------------------------------------------------------------------------

// S1: per (batch, l, tile of OFFS_TILE sequences) sum.  One thread per l.
__global__ __launch_bounds__(64) void offsets_tile_sums(OffsArgs a) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;                   // tile inside the batch
    const int b = blockIdx.z;                   // batch
    if (l >= a.Lout) return;
    const int64_t nb0 = (int64_t)b * a.batch;
    const int64_t n_lo = nb0 + (int64_t)t * OFFS_TILE;
    int64_t n_hi = n_lo + OFFS_TILE;
    const int64_t bend = nb0 + a.batch < a.N ? nb0 + a.batch : a.N;
    if (n_hi > bend) n_hi = bend;
    uint32_t s = 0;
    for (int64_t n = n_lo; n < n_hi; n++)
        for (int ch = 0; ch < a.nch; ch++) s += a.cnt[((size_t)n * a.nch + ch) * a.LoutP + l];
    a.tilesum[((size_t)b * a.Lout + l) * a.tiles + t] = s;
}

// S2: one block; exclusive scan of tilesum in memory order (b, l, t) restarted
// at every batch; batch totals -> batch_base (exclusive, int64) and total.
__global__ __launch_bounds__(1024) void offsets_scan(OffsArgs a) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per_batch = (int64_t)a.Lout * a.tiles;
    unsigned long long run_total = 0;           // hits in earlier batches
    for (int b = 0; b < a.nbatch; b++) {
        uint32_t* ts = a.tilesum + (size_t)b * per_batch;
        const int64_t chunk = (per_batch + 1023) / 1024;
        int64_t lo = tid * chunk, hi = lo + chunk;
        if (lo > per_batch) lo = per_batch;
        if (hi > per_batch) hi = per_batch;
        unsigned long long s = 0;
        for (int64_t i = lo; i < hi; i++) s += ts[i];
        part[tid] = s;
        __syncthreads();
        // Hillis-Steele inclusive scan over 1024 partials
        for (int d = 1; d < 1024; d <<= 1) {
            unsigned long long v = tid >= d ? part[tid - d] : 0ull;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        unsigned long long run = part[tid] - s;  // exclusive prefix of this thread's chunk
        const unsigned long long btotal = part[1023];
        __syncthreads();
        if (btotal > 0xffffffffull && tid == 0) *a.overflow = 1;
        for (int64_t i = lo; i < hi; i++) {
            const uint32_t c = ts[i];
            ts[i] = (uint32_t)run;
            run += c;
        }
        if (tid == 0) a.batch_base[b] = (int64_t)(a.base0 + run_total);
        run_total += btotal;
    }
    if (tid == 0) *a.total = (int64_t)run_total;
}

// S3: expand tile offsets to per-(n, ch, l) offsets.
__global__ __launch_bounds__(64) void offsets_expand(OffsArgs a) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;
    const int b = blockIdx.z;
    if (l >= a.Lout) return;
    const int64_t nb0 = (int64_t)b * a.batch;
    const int64_t n_lo = nb0 + (int64_t)t * OFFS_TILE;
    int64_t n_hi = n_lo + OFFS_TILE;
    const int64_t bend = nb0 + a.batch < a.N ? nb0 + a.batch : a.N;
    if (n_hi > bend) n_hi = bend;
    uint32_t run = a.tilesum[((size_t)b * a.Lout + l) * a.tiles + t];
    for (int64_t n = n_lo; n < n_hi; n++)
        for (int ch = 0; ch < a.nch; ch++) {
            const size_t i = ((size_t)n * a.nch + ch) * a.LoutP + l;
            a.off[i] = run;
            run += a.cnt[i];
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
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int LEN>
static hipError_t launch_len(int mode, const ScanArgs& a, dim3 grid, hipStream_t st) {
    switch (mode) {
#define MOTIFS_SCAN_ARGS a.tab, a.lim, a.codes, a.scores, a.cnt, a.off, a.batch_base, a.hits, a.hit_scores, a.pwm_counts, a.d
        case MODE_DENSE: hipLaunchKernelGGL((scan_kernel<LEN, MODE_DENSE>), grid, dim3(SCAN_BLOCK), 0, st, MOTIFS_SCAN_ARGS); break;
        case MODE_COUNT: hipLaunchKernelGGL((scan_kernel<LEN, MODE_COUNT>), grid, dim3(SCAN_BLOCK), 0, st, MOTIFS_SCAN_ARGS); break;
        case MODE_FILL: hipLaunchKernelGGL((scan_kernel<LEN, MODE_FILL>), grid, dim3(SCAN_BLOCK), 0, st, MOTIFS_SCAN_ARGS); break;
#undef MOTIFS_SCAN_ARGS
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int scan_len_padded(int maxlen) {
    static const int sizes[] = {8, 12, 16, 20, 24, 32};
    for (int s : sizes)
        if (maxlen <= s) return s;
    return -1;
}

hipError_t launch_scan(int mode, int len_padded, const ScanArgs& a, hipStream_t st) {
    const int64_t seqs_per_block = (int64_t)SCAN_WAVES * a.d.spw;
    dim3 grid((unsigned)((a.d.N + seqs_per_block - 1) / seqs_per_block), (unsigned)a.d.nch, 1);
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

hipError_t launch_offsets(const OffsArgs& a, hipStream_t st) {
    dim3 g((unsigned)((a.Lout + 63) / 64), (unsigned)a.tiles, (unsigned)a.nbatch);
    hipLaunchKernelGGL(offsets_tile_sums, g, dim3(64), 0, st, a);
    hipLaunchKernelGGL(offsets_scan, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(offsets_expand, g, dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_encode(int kind, const void* x, int64_t N, int L, int pitch, uint8_t* codes, int32_t* bad,
                         hipStream_t st) {
    const int64_t total = N * L;
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
