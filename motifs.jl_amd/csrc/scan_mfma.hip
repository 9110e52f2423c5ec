// scan_mfma.hip — the hit-record path of the PWM scan on the matrix cores.
//
// A PWM score is a 4*len-long dot product of the PWM with the one-hot window: a GEMM
// [PWMs x 4*len] x [4*len x windows].  v_mfma_f32_32x32x16_f16 forms it exactly in f32 (products of
// binary16 weights with 0/1 are exact), which is NOT the reference's arithmetic — greedy_search!
// (src/inference/_h3_1_alignment.jl:26-31) adds in binary16, rounding after every add.  So the matrix
// cores only FILTER: with S the exact sum and s the sequentially rounded one, a window with s > 0 has S > -eps_k, where
// eps_k = c * sum_{i=2..len_k} B_i / (1 - c * len_k), c = (1 + 2^-11)^len_k * 2^-11, and B_i bounds the i-th exact prefix
// sum of a window whose total lies in [-eps_k, 0]: it is pinned by what the first i positions can reach AND by what the
// remaining ones can still undo (pack_mfma in scan_api.hip derives it; for log-odds banks it is ~5x below the plain
// sum_{j<=i} max_a |pwm[k, a, j]|); small absolute terms cover binary16 subnormals and the f32 GEMM's own rounding,
// and eps = inf (keep everything) where a partial sum could overflow binary16.  Every (PWM, window)
// with S > -eps_k becomes a candidate bit; candidates (under 1 % of the pairs) are then re-scored in the
// reference's arithmetic and only true hits survive (stage_hits), so records and scores stay
// bit-identical to the reference while 99 % of the pairs never touch the slow fp16 chain.
//
// Cells: uint4 per (batch, l, n-in-batch, chunk of 128 PWMs), bit i = PWM 128*chunk + i.  This order is
// the reference's record order (5000-read batches, then findall's column-major walk), so record
// offsets are a plain exclusive scan of popcounts.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
//   A and C arrive negated (pack_mfma): C = -eps_k per PWM row, the result is -(S + eps_k) and its sign bit says
//                     "candidate".
// NG = live tiles of the group (tiles past the bank hold no PWM).  The body is branch-free so that the NG
// accumulator chains interleave: the matrix pipe works on one tile while the VALU packs the signs of another.
// emit(l0, wa, wb) receives the candidate words of window l0 + w: PG = 4: lane half h holds words 2h, 2h + 1 of the
// chunk; PG = 2: wa = word h of the wave's pair (wb = 0); PG = 1: wa = the word (both halves).
// ohl = the lane's column in the one-hot image at window tile 0 (position of its window + 2h); WSTEP = windows a tile
// advances by (32: one read per wave; 8: four reads per wave, scan_cand_kernel_q).
template <int T, int PG, int NG, int NC, int WSTEP, typename E>
static __device__ __forceinline__ void cand_read(const f16x8 (&A)[PG][T], const f32x16 (&C0)[NC], const uint2* ohl, int ntile, E&& emit) {
    for (int wt = 0; wt < ntile; wt++) {
        const int l0 = wt * WSTEP;
        f16x8 B[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint2 a0 = ohl[l0 + 4 * t], a1 = ohl[l0 + 4 * t + 1];
            B[t] = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
        }
        f32x16 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            f32x16 c;
            if (NC == 1) {                // a constant per chain: one shared splat is materialised in 16 VGPRs per tile instead
                const float cv = g == 0 ? -4.0f : g == 1 ? -2.0f : g == 2 ? -1.0f : -0.5f;
#pragma unroll
                for (int r = 0; r < 16; r++) c[r] = cv;
            } else {
                c = C0[g];
            }
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][0], B[0], c, 0, 0, 0);
        }
#pragma unroll
        for (int t = 1; t < T; t++)
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g][t], B[t], acc[g], 0, 0, 0);
        uint32_t m[PG];           // bit r = accumulator r is a candidate (sign set: -(S + eps) < 0)
        // PG == 4: the 16 sign bits of tile 1 (3) follow those of tile 0 (2) into the same word - m[0] = tile 0 << 16 | tile 1, m[2] likewise -
        // so that ONE v_permlane32_swap hands every lane both halves of its two tiles
#pragma unroll
        for (int g = 0; g < PG; g++) {
            const bool cont = PG == 4 && (g & 1);                 // continues the word of tile g - 1
            if (!cont) m[g] = 0;
            uint32_t& mg = m[cont ? g - 1 : g];
            if (g < NG) {
#pragma unroll
                for (int r = 15; r >= 0; r--) mg = __builtin_amdgcn_alignbit(mg, __float_as_uint(acc[g][r]), 31);
            } else if (cont) {
                mg <<= 16;
            }
        }
        // the other 16 PWMs of a tile sit in the other half of the wave: v_permlane32_swap hands lane (w, 0) both
        // halves of one tile and lane (w, 1) both halves of another, so all 64 lanes have words to deliver
        if (PG == 4) {
            const auto sw = __builtin_amdgcn_permlane32_swap(m[0], m[2], false, false);
            // (v_perm_b32: the high halves of the two words = tile 0 / 2, the low halves = tile 1 / 3)
            emit(l0, __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u), __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u));   // h=0: words 0,1; h=1: words 2,3
        } else if (PG == 2) {
            const auto s01 = __builtin_amdgcn_permlane32_swap(m[0], m[1 % PG], false, false);
            emit(l0, s01[0] | (s01[1] << 16), 0u);                                        // h=0: word 0; h=1: word 1
        } else {
            const auto s00 = __builtin_amdgcn_permlane32_swap(m[0], m[0], false, false);
            emit(l0, s00[0] | (s00[1] << 16), 0u);
        }
    }
}

// the compact entry of a half cell (see scan_cand_kernel_q): count << 12 | (63 - last) << 6 | first, or 3 << 12 | popcount
static __device__ __forceinline__ uint32_t half_cell_entry(uint32_t wa, uint32_t wb, uint32_t& cnt) {
    // v_ffbl_b32 / v_ffbh_u32 answer -1 for an empty word, which `| 32` leaves at -1 and the unsigned minimum then ignores
    uint32_t fa, fb, la, lb;
    asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(wa));
    asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(wb));
    asm("v_ffbh_u32 %0, %1" : "=v"(la) : "v"(wb));
    asm("v_ffbh_u32 %0, %1" : "=v"(lb) : "v"(wa));
    cnt = (uint32_t)__builtin_popcount(wa) + (uint32_t)__builtin_popcount(wb);
    const uint32_t first = min(fa, fb | 32u), lastp = min(la, lb | 32u);   // 0..63 whenever cnt >= 1 (garbage for an empty half cell)
    // selects, not branches: this sits between MFMAs.  Count 0: the low bits (the garbage, cut to 12 bits) are ignored by the consumers
    return cnt < 3u ? (((lastp << 6) | first) & 0xfffu) | (cnt << 12) : (cnt | 0x3000u);
}

// UEPS: the bank was scaled by powers of two on the host so that the slack of tile g of a group is the inline constant
// -4 / 2^g (C is then not a register operand and the wave needs 64 VGPRs fewer); otherwise C = -eps_k from cinit.
template <int T, int PG, bool UEPS, int RPB, int TGB, bool COMPACT = false>
static __device__ __forceinline__ void scan_cand_body(const uint4* __restrict__ afrag, const float* __restrict__ cinit,
                                                      const uint8_t* __restrict__ codes, uint32_t* __restrict__ cells,
                                                      const CandDims& d, uint16_t* __restrict__ centries = nullptr) {
    extern __shared__ uint2 oh_all[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = lane & 31, h = lane >> 5;
    uint2* oh = oh_all + (size_t)wave * ((d.ohlen + 3) & ~3);
    // a block = RPB reads x TGB tile groups: with TGB = 2 the block writes both chunks of its reads (half lines of
    // cells; 4 reads x 1 group was measured 1.7x slower on a two-group bank: the halves of a line then come from
    // different blocks at different times); TGB = 1 serves banks with a single tile group.
    const int slot = wave % RPB;
    const int tg = blockIdx.y * TGB + wave / RPB;    // tile group: PWM tiles [tg*PG, tg*PG + PG)
    if (tg * PG >= d.used_tiles) return;
    const int tile0 = tg * PG;
    const int chunk = tile0 >> 2, word0 = tile0 & 3;
    const int ng = d.used_tiles - tile0 < PG ? d.used_tiles - tile0 : PG;   // wave-uniform

    constexpr int NC = UEPS ? 1 : PG;
    f16x8 A[PG][T];
    f32x16 C0[NC];
#pragma unroll
    for (int g = 0; g < PG; g++) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint4 v = afrag[((size_t)(tile0 + g) * T + t) * 64 + lane];
            A[g][t] = __builtin_bit_cast(f16x8, v);
        }
        if (!UEPS) {
#pragma unroll
            for (int r = 0; r < 16; r++) C0[g][r] = cinit[((size_t)(tile0 + g) * 2 + h) * 16 + r];
        }
    }
    if (UEPS) {
#pragma unroll
        for (int r = 0; r < 16; r++) C0[0][r] = 0.f;   // unused: cand_read takes the inline constants
    }

    const unsigned lb = xcd_swz(blockIdx.x, gridDim.x);
    const int ntile = (d.Lout + 31) / 32;
    const size_t lstride4 = (size_t)d.batch * (COMPACT ? d.cgc : d.nch) * 4;      // words between the cells of l and l + 1
    for (int s = 0; s < d.spw; s++) {
        const int64_t n = ((int64_t)lb * d.spw + s) * RPB + slot;      // wave-uniform
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
                const uint64_t one = c < 4 ? (uint64_t)0x3c00u << (16 * c) : 0ull;
                if (p < d.ohlen) oh[p] = make_uint2((uint32_t)one, (uint32_t)(one >> 32));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t bq = n / d.batch;
        // l = 0.  With chunk groups (d.cgc < d.nch) a batch's cells are group-major: (group, l, read, chunk in group)
        const size_t cell0 = COMPACT ? (((size_t)bq * (d.nch / d.cgc) + chunk / d.cgc) * d.Lout * d.batch + (size_t)(n - bq * d.batch)) * d.cgc + chunk % d.cgc
                                     : ((size_t)bq * d.Lout * d.batch + (size_t)(n - bq * d.batch)) * d.nch + chunk;
        // the lane's cell of window tile 0; the window tiles follow 32 cell lines apart (a running pointer: the 64-bit
        // multiply per store cost two v_mad_u64_u32 per tile)
        uint32_t* cp = cells + cell0 * 4 + word0 + (size_t)w * lstride4 + (PG == 4 ? 2 * h : PG == 2 ? h : 0);
        const size_t tile_step = 32 * lstride4;
        uint32_t ei = COMPACT ? (uint32_t)(cell0 * 2 + h + (size_t)w * (lstride4 / 2)) : 0u;     // compact entries: see scan_cand_kernel_q
        const uint32_t ei_step = (uint32_t)(16 * lstride4);
        if (COMPACT) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no load of this wave is pending inside the tile loop
        auto store_cells = [&](int l0, uint32_t wa, uint32_t wb) {
            const int l = l0 + w;
            if (COMPACT && PG == 4) {
                uint32_t cnt;
                const uint32_t e = half_cell_entry(wa, wb, cnt);
                if (l < d.Lout) centries[ei] = (uint16_t)e;
                if (l < d.Lout && cnt >= 3u) *(uint2*)(cells + (size_t)ei * 2) = make_uint2(wa, wb);
                ei += ei_step;
                return;
            }
            if (PG == 4) {
                if (l < d.Lout && (ng > 2 || h == 0)) *(uint2*)cp = make_uint2(wa, wb);
            } else if (PG == 2) {
                if (l < d.Lout) *cp = wa;
            } else {
                if (l < d.Lout && h == 0) *cp = wa;
            }
            cp += tile_step;
        };
        const uint2* ohl = oh + w + 2 * h;
        switch (ng) {
            case 1: cand_read<T, PG, 1, NC, 32>(A, C0, ohl, ntile, store_cells); break;
            case 2: cand_read<T, PG, (PG >= 2 ? 2 : PG), NC, 32>(A, C0, ohl, ntile, store_cells); break;
            case 3: cand_read<T, PG, (PG >= 3 ? 3 : PG), NC, 32>(A, C0, ohl, ntile, store_cells); break;
            default: cand_read<T, PG, PG, NC, 32>(A, C0, ohl, ntile, store_cells); break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int T, int PG>
__global__ __launch_bounds__(512) void scan_cand_kernel(const uint4* __restrict__ afrag, const float* __restrict__ cinit,
                                                        const uint8_t* __restrict__ codes, uint32_t* __restrict__ cells,
                                                        const CandDims d) {
    scan_cand_body<T, PG, false, 4, 2>(afrag, cinit, codes, cells, d);
}
// (round 5: the one-read-per-wave kernel for uniform-slack banks, scan_cand_kernel_u, is gone - since the four-reads kernel takes segments
// of a read's window tiles it serves reads of any length, and nothing reached the other form but an A/B switch)

// Four reads per wave (uniform slack, short reads): the 32 columns of a tile are 8 consecutive windows of 4 consecutive
// reads, lane (w, h) -> read w >> 3, window w & 7.  The cells of one start l and 4 reads x 2 chunks are one 128-byte line,
// so a wave's store touches 8 lines instead of 32 (the 32-line scatter cost 10 % of the kernel: the same bytes stored
// contiguously ran 0.343 ms against 0.381), and the per-read overhead (staging, cell address) is shared by 24 tiles
// instead of 6.  The four images sit `opitch` apart in LDS, opitch = 8 (mod 32) positions = 64 (mod 256) bytes: the 8-byte
// B-operand reads of a half wave then cover the 64 banks exactly once.
// A-fragment registers (T * PG) up to which the four-reads kernel is held to three waves per SIMD (168 VGPRs).  20 = PWMs of 17-20 positions
// (BASELINE configs[3] / [4]): the kernel wants 178 and is given 168 - ten dwords spill, all of them outside the tile loop - for a third wave
// on the matrix pipe: candidates 3.30 -> 3.21 ms at the configs[3] shard, 10.45 -> 9.98 at configs[4] (16: two waves there).
#ifndef CAND_Q_W3
#define CAND_Q_W3 20
#endif
static __host__ __device__ inline int quad_pitch(int ohlen) { return ((((ohlen + 3) & ~3) - 8 + 31) & ~31) + 8; }

// Compact entries (COMPACT, tile groups of 4 only): candidates are under 1 % of the (PWM, window) pairs, so a half cell - the 64
// PWMs a lane holds after the exchange - is empty or holds one or two candidates 98.6 % of the time.  Instead of its 8 bytes of
// bits the lane stores ONE 16-bit entry, count << 12 | (63 - last) << 6 | first, and only a half cell with three or more
// candidates (entry 3 << 12 | popcount) also leaves its bits in the cell array, where the consumer fetches them.  The entries
// keep the cells' order (batch, l, read, chunk, half), so the consumers' rows, offsets and record order do not change; what
// changes is the traffic of the round trip: 151 MB + the rare cells instead of 605 MB written and read back per strand of
// BASELINE configs[1].
template <int T, int PG, int TGB, bool COMPACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((T * PG <= 12) ? 4 : (T * PG <= CAND_Q_W3 && PG > 1) ? 3 : 2, 4))) void scan_cand_kernel_q(
    const uint4* __restrict__ afrag, const uint8_t* __restrict__ codes, uint32_t* __restrict__ cells, uint16_t* __restrict__ centries, const CandDims d,
    const uint4* __restrict__ afrag2, uint32_t* __restrict__ cells2, uint16_t* __restrict__ centries2) {
    // afrag2 != nullptr (COMPACT only): the reverse strand's bank goes over the SAME staged reads right after the forward one's -
    // gpu_scan (_h3_1_alignment.jl:89-99) scans every read with both banks, and what a wave does before its first tile (code loads,
    // four one-hot images, the cell address) and after its last is 9 % of a one-strand pass (measured by running the tile loop
    // twice: 1.128 ms for two passes' worth against 2 x 0.591).  The second bank's fragments replace the first's in the same registers.
    extern __shared__ __attribute__((aligned(16))) uint2 oh_all[];
    constexpr int QPB = 4 / TGB;                     // quads of reads per block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = lane & 31, h = lane >> 5, rq = w >> 3, wq = w & 7;
    const int opitch = quad_pitch(d.ohseg);
    uint2* oh = oh_all + (size_t)wave * 4 * opitch;
    const int slot = wave % QPB;
    const int tg = blockIdx.y * TGB + wave / QPB;
    if (d.zero_n && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < d.zero_n; i += 256) d.zero_ptr[i] = 0ull;
    if (tg * PG >= d.used_tiles) return;
    // the block's segment of the reads: window tiles [wt0, wt0 + ntile), image positions [p0, p0 + ohs)
    const int wt0 = (int)blockIdx.z * d.seg_tiles;
    const int ntile = min(d.seg_tiles, (d.Lout + 7) / 8 - wt0);
    if (ntile <= 0) return;
    const int p0 = wt0 * 8, ohs = d.nseg == 1 ? d.ohlen : ntile * 8 + 4 * T;
    const int tile0 = tg * PG;
    const int chunk = tile0 >> 2, word0 = tile0 & 3;
    const int ng = d.used_tiles - tile0 < PG ? d.used_tiles - tile0 : PG;   // wave-uniform

    f16x8 A[PG][T];
    f32x16 C0[1];
#pragma unroll
    for (int r = 0; r < 16; r++) C0[0][r] = 0.f;     // unused: cand_read takes the inline constants
    const int nstrand = (COMPACT && afrag2) ? 2 : 1;
    auto load_bank = [&](const uint4* af) {
#pragma unroll
        for (int g = 0; g < PG; g++)
#pragma unroll
            for (int t = 0; t < T; t++) A[g][t] = __builtin_bit_cast(f16x8, af[((size_t)(tile0 + g) * T + t) * 64 + lane]);
    };
    load_bank(afrag);

    const unsigned lb = xcd_swz(blockIdx.x, gridDim.x);
    const size_t lstride4 = (size_t)d.batch * (COMPACT ? d.cgc : d.nch) * 4;      // words between the cells of l and l + 1
    const uint2* ohl = oh + rq * opitch + wq + 2 * h;
    // (ordering batch, read in batch) of the wave's first read; later quads advance it without dividing
    int64_t nq = ((int64_t)lb * d.spw * QPB + slot) * 4;
    int64_t bq0 = nq / d.batch;
    int r0 = (int)(nq - bq0 * d.batch);
    for (int s = 0; s < d.spw; s++, nq += 4 * QPB) {
        if (nq >= d.N) break;
        // stage the one-hot images: 4 halves per position (1.0 at the base; all zero for code 4, padding and reads past N);
        // a lane turns one dword of codes into four positions
        // (the four reads' code words are requested together: one trip to memory per round of 256 positions, not four)
        for (int p4 = lane; p4 * 4 < ohs; p4 += 64) {
            const int pa = p4 + p0 / 4;              // the dword of the read this image dword comes from
            const int keep = d.L - pa * 4;           // positions of this dword inside the read
            uint32_t wv4[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const uint32_t* srow = (const uint32_t*)(codes + (nq + rr) * d.pitch);
                wv4[rr] = (nq + rr < d.N && keep > 0) ? srow[pa] : 0x04040404u;
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                uint32_t wv = wv4[rr];
                if (keep > 0 && keep < 4) {
                    const uint32_t mk = (1u << (8 * keep)) - 1u;
                    wv = (wv & mk) | (0x04040404u & ~mk);
                }
                const uint32_t sh = wv << 4;         // 16 * code per byte; code 4 -> shift 63: the 1.0 leaves the word
                uint64_t one[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t su = (sh >> (8 * u)) & 0xffu;
                    one[u] = (uint64_t)0x3c00u << (su < 63u ? su : 63u);
                }
                uint4* dst = (uint4*)(oh + rr * opitch + p4 * 4);
                dst[0] = make_uint4((uint32_t)one[0], (uint32_t)(one[0] >> 32), (uint32_t)one[1], (uint32_t)(one[1] >> 32));
                dst[1] = make_uint4((uint32_t)one[2], (uint32_t)(one[2] >> 32), (uint32_t)one[3], (uint32_t)(one[3] >> 32));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the lane's read and its cell of window tile 0; window tiles follow 8 cell lines apart
        int64_t bql = bq0;
        int rl = r0 + rq;
        while (rl >= d.batch) rl -= d.batch, bql++;
        const bool rowl = nq + rq < d.N;
        // (chunk groups, d.cgc < d.nch: a batch's cells are group-major - (group, l, read, chunk in group))
        const size_t cell0 = COMPACT ? (((size_t)bql * (d.nch / d.cgc) + chunk / d.cgc) * d.Lout * d.batch + (size_t)rl) * d.cgc + chunk % d.cgc
                                     : ((size_t)bql * d.Lout * d.batch + (size_t)rl) * d.nch + chunk;
        const size_t tile_step = 8 * lstride4;
        uint32_t* cp = cells + cell0 * 4 + word0 + (size_t)wq * lstride4 + (PG == 4 ? 2 * h : PG == 2 ? h : 0) + (size_t)wt0 * tile_step;
        // compact entries: a 32-bit running entry index (entries of a super-batch number < 2^32) instead of a second 64-bit pointer
        const uint32_t ei_step = (uint32_t)(4 * lstride4) * 2u;
        const uint32_t ei0 = COMPACT ? (uint32_t)(cell0 * 2 + h + (size_t)wq * (lstride4 / 2)) * 2u + (uint32_t)wt0 * ei_step : 0u;   // BYTE offset of the lane's entry (< 2^32)
        for (int strand = 0; strand < nstrand; strand++) {        // wave-uniform
            uint32_t* const cells_s = strand ? cells2 : cells;
            uint16_t* const centries_s = strand ? centries2 : centries;
            if (nstrand == 2 && (strand || s)) load_bank(strand ? afrag2 : afrag);  // (the first quad's forward bank is already in the registers)
            uint32_t ei = ei0;
            auto store_cells = [&](int l0, uint32_t wa, uint32_t wb) {
                const bool live = rowl && wq < d.Lout - p0 - l0;                                      // (a scalar bound: no vector add per tile)
                if (COMPACT && PG == 4) {
                    uint32_t cnt;
                    const uint32_t e = half_cell_entry(wa, wb, cnt);
                    if (live) *(uint16_t*)((char*)centries_s + ei) = (uint16_t)e;                     // scalar base + 32-bit lane offset
                    if (live && cnt >= 3u) *(uint2*)((char*)cells_s + (size_t)ei * 4) = make_uint2(wa, wb);
                    ei += ei_step;
                    return;
                }
                if (PG == 4) {
                    if (live && (ng > 2 || h == 0)) *(uint2*)cp = make_uint2(wa, wb);
                } else if (PG == 2) {
                    if (live) *cp = wa;
                } else {
                    if (live && h == 0) *cp = wa;
                }
                cp += tile_step;
            };
            // Every load of this wave (PWM fragments, code words) has landed before the tile loop starts: without this the compiler
            // guards the fragments' first uses with s_waitcnt vmcnt(n) INSIDE the loop, and since vmcnt counts in order those waits
            // also drain the tile stores of the previous turn (seen in the ISA of the compact-entry build: vmcnt(8) ... vmcnt(0)
            // between the twelve MFMAs of a tile).
            if (COMPACT) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) only
            switch (ng) {
                case 1: cand_read<T, PG, 1, 1, 8>(A, C0, ohl, ntile, store_cells); break;
                case 2: cand_read<T, PG, (PG >= 2 ? 2 : PG), 1, 8>(A, C0, ohl, ntile, store_cells); break;
                case 3: cand_read<T, PG, (PG >= 3 ? 3 : PG), 1, 8>(A, C0, ohl, ntile, store_cells); break;
                default: cand_read<T, PG, PG, 1, 8>(A, C0, ohl, ntile, store_cells); break;
            }
        }
        __builtin_amdgcn_wave_barrier();
        r0 += 4 * QPB;
        while (r0 >= d.batch) r0 -= d.batch, bq0++;
    }
}

// Any PWM length (the reference has no cap: _h3_1_alignment.jl:25-31; motif length is d13 + h, _2_enumerate.jl:43, and grows
// in the expansions of _h5): the k-steps of the contraction are a run-time loop, so the PWM fragments cannot live in
// registers - they are re-read from L2 per step (16 bytes per lane and MFMA).  One read per wave, one tile of 32 PWMs per wave
// (the PG = 1 cell layout of cand_read); the slack is the inline -4 of a scaled bank or the per-PWM row of cinit.  Not a
// fast path - a bank this long is rare and its scan is bound by these loads - but no bank is refused.
__global__ __launch_bounds__(256) void scan_cand_kernel_g(const uint4* __restrict__ afrag, const float* __restrict__ cinit,
                                                          const uint8_t* __restrict__ codes, uint32_t* __restrict__ cells, const CandDims d,
                                                          const int T, const int uniform_eps) {
    extern __shared__ uint2 oh_all[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = lane & 31, h = lane >> 5;
    const int wpb = blockDim.x >> 6;
    uint2* oh = oh_all + (size_t)wave * ((d.ohlen + 3) & ~3);
    const int tile = blockIdx.y;                                   // one tile of 32 PWMs per wave
    if (tile >= d.used_tiles) return;
    const int chunk = tile >> 2, word0 = tile & 3;
    f32x16 c0;
#pragma unroll
    for (int r = 0; r < 16; r++) c0[r] = uniform_eps ? -4.0f : cinit[((size_t)tile * 2 + h) * 16 + r];
    const uint4* af = afrag + (size_t)tile * T * 64 + lane;
    const int ntile = (d.Lout + 31) / 32;
    const size_t lstride4 = (size_t)d.batch * d.nch * 4;
    for (int s = 0; s < d.spw; s++) {
        const int64_t n = ((int64_t)blockIdx.x * d.spw + s) * wpb + wave;      // wave-uniform
        if (n >= d.N) break;
        const uint32_t* srow = (const uint32_t*)(codes + n * d.pitch);
        for (int p4 = lane; p4 * 4 < d.ohlen; p4 += 64) {
            const uint32_t wv = p4 * 4 < d.L ? srow[p4] : 0x04040404u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = p4 * 4 + u;
                const uint32_t c = p < d.L ? (wv >> (8 * u)) & 0xffu : 4u;
                const uint64_t one = c < 4 ? (uint64_t)0x3c00u << (16 * c) : 0ull;
                if (p < d.ohlen) oh[p] = make_uint2((uint32_t)one, (uint32_t)(one >> 32));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t bq = n / d.batch;
        const size_t cell0 = ((size_t)bq * d.Lout * d.batch + (size_t)(n - bq * d.batch)) * d.nch + chunk;
        uint32_t* cp = cells + cell0 * 4 + word0 + (size_t)w * lstride4;
        const uint2* ohl = oh + w + 2 * h;
        for (int wt = 0; wt < ntile; wt++) {
            const int l0 = wt * 32;
            f32x16 acc = c0;
            for (int t = 0; t < T; t++) {
                const uint2 a0 = ohl[l0 + 4 * t], a1 = ohl[l0 + 4 * t + 1];
                const f16x8 B = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
                const f16x8 A = __builtin_bit_cast(f16x8, af[(size_t)t * 64]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
            }
            uint32_t m = 0;
#pragma unroll
            for (int r = 15; r >= 0; r--) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[r]), 31);
            const auto s00 = __builtin_amdgcn_permlane32_swap(m, m, false, false);
            if (l0 + w < d.Lout && h == 0) *cp = s00[0] | (s00[1] << 16);
            cp += 32 * lstride4;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- candidates -> records ---------------------------------------------------------------------------------
// stage_hits: every WAVE takes rows of cells (a row = the (read, chunk) cells of a few reads at one start l),
// pushes the set bits of a row into an LDS ring and re-scores the ring 64 candidates at a time in the
// reference's arithmetic with all lanes busy; a hit becomes one packed word (cell, bit, score) in the row's
// staging slots and the row's hit count is stored.  No block barriers after the table is staged and no
// traffic between waves.  After an exclusive scan of the row counts, emit_records turns the staged words
// into (m, n, l, score) records at their final offsets; a row with more hits than staging slots is re-scored
// from its cells there (rare: more than two hits per cell on average).
constexpr int VF_THREADS = 512;
constexpr int VF_WAVES = VF_THREADS / 64;
constexpr int QN = 512;           // candidate queue slots per wave; fewer than 64 stay behind after a push
constexpr int ROW_CELLS_MAX = 512;

// The re-scoring table: one row of binary16 weights per PWM, [k][ind][5] with column 4 = +0 for an all-zero data
// column and +0 beyond lens[k]; rows are padded to an odd number of dwords (a.tabk_stride halves) so that lanes
// scoring different PWMs hit different LDS banks.  One weight = v_bfe (the base) + v_lshl_add (the address) +
// ds_read_u16 + v_add_f16.
// sequential binary16 sum of a PWM's row over the window whose raw code words are W (reference order, one rounding per add)
template <int LEN, bool LDS = false>
static __device__ __forceinline__ uint16_t exact_score(const _Float16* row, const uint32_t (&W)[LEN / 4 + 1], int l) {
    _Float16 t[LEN];
    if constexpr (LDS) {
        // table in LDS: the window's bytes are doubled once (codes are 0..4: no carry between bytes; before the byte alignment, so that the
        // shift cannot be folded back into the extraction), and a position's address is the row's LDS byte address plus one byte of a
        // register - ONE v_add_u32 with a byte-select operand (SDWA) where the generic form spends a v_bfe and a v_lshl_add
        typedef const __attribute__((address_space(3))) _Float16* lds_half;
        uint32_t al[LEN / 4];
#pragma unroll
        for (int j = 0; j < LEN / 4; j++) al[j] = __builtin_amdgcn_alignbyte(W[j + 1] << 1, W[j] << 1, (uint32_t)(l & 3));
        const uint32_t rb = (uint32_t)(uintptr_t)row;                 // a generic pointer into LDS: its low word is the LDS address
#pragma unroll
        for (int ind = 0; ind < LEN; ind++)
            t[ind] = *(lds_half)(uintptr_t)(rb + ((al[ind / 4] >> (8 * (ind % 4))) & 0xffu) + (uint32_t)(ind * 10));
    } else {
        uint32_t al[LEN / 4];
#pragma unroll
        for (int j = 0; j < LEN / 4; j++) al[j] = __builtin_amdgcn_alignbyte(W[j + 1], W[j], (uint32_t)(l & 3));
#pragma unroll
        for (int ind = 0; ind < LEN; ind++) t[ind] = row[ind * 5 + ((al[ind / 4] >> (8 * (ind % 4))) & 0xffu)];
    }
    __builtin_amdgcn_sched_barrier(0);        // all LEN reads in flight before the dependent chain of adds starts
    _Float16 acc = t[0];
#pragma unroll
    for (int ind = 1; ind < LEN; ind++) acc = acc + t[ind];
    return __builtin_bit_cast(uint16_t, acc);
}
// the same for a run-time length (banks past the template sizes): one byte load per position of THIS PWM (len = lens[k]: the
// template form runs to the padded length, where the table holds +0 and the reads stay inside the guard bytes; here the padded
// length has no bound, so the loop stops at the window's end), adds in the reference's order
static __device__ __forceinline__ uint16_t exact_score_dyn(const _Float16* row, const uint8_t* win, int len) {
    _Float16 acc = row[win[0]];
    for (int ind = 1; ind < len; ind++) acc = acc + row[ind * 5 + win[ind]];
    return __builtin_bit_cast(uint16_t, acc);
}
// binary16 bits > 0, as `pos_scores > 0f0` decides it (_h3_1_alignment.jl:33): +Inf counts, NaN (an Inf - Inf of
// overflowing partial sums) does not
static __device__ __forceinline__ bool half_pos(uint16_t h) { return (int16_t)h > 0 && h <= 0x7c00u; }

// the table into LDS when it fits (a straight copy)
template <bool LDS_TAB, int THREADS>
static __device__ __forceinline__ const _Float16* stage_table(const FillArgs& a, uint32_t* ltab) {
    if (!LDS_TAB) return (const _Float16*)a.tabk;
    const int ndw = (a.K * a.tabk_stride + 1) / 2;
    const uint32_t* src = (const uint32_t*)a.tabk;
    for (int i = threadIdx.x; i < ndw; i += THREADS) ltab[i] = src[i];
    __syncthreads();
    return (const _Float16*)ltab;
}

// inclusive prefix sum over the 64 lanes with DPP moves (no LDS round trips): within rows of 16 by row_shr, then
// row_bcast:15 / row_bcast:31 carry the row totals forward
static __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);   // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);   // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);   // row_shr:8, banks 2-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}
// LDS traffic between lanes of one wave: program order is execution order, the fences stop the compiler
static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the candidates of one 64-cell slab into the wave's queue, draining full batches of 64 through fn(word)
// candidate word = cell index in the row << 7 | word << 5 | bit.  The queue is a linear buffer (whatever a drain leaves,
// fewer than 64 words, moves to its front), so that a lane writes its candidates through a running pointer: per set bit
// ctz, or, store, clear - no ring arithmetic and no capacity test (a slab that does not fit whole, > 448 candidates in 64
// cells, goes in pieces through the checked loop).
// NW = 4 words per cell x the cells a lane holds (idx = its first cell)
template <int NW, typename F>
static __device__ __forceinline__ void push_and_drain(uint16_t* queue, uint32_t& qlen, const uint32_t (&wd)[NW], uint32_t idx, uint32_t ex,
                                                      uint32_t tot, F&& fn) {
    const int lane = threadIdx.x & 63;
    uint32_t done = 0;
    while (true) {                                                    // wave-uniform
        const uint32_t room = QN - qlen;
        const uint32_t take = tot - done < room ? tot - done : room;
        if (take == tot) {
            uint16_t* p = queue + qlen + ex;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                uint32_t bits = wd[q];
                const uint32_t pre = ((idx + (uint32_t)(q >> 2)) << 7) | ((uint32_t)(q & 3) << 5);
                while (bits) {
                    *p++ = (uint16_t)(pre | (uint32_t)__builtin_ctz(bits));
                    bits &= bits - 1;
                }
            }
        } else {
            uint32_t g = ex;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                uint32_t bits = wd[q];
                while (bits) {
                    const int i = __builtin_ctz(bits);
                    bits &= bits - 1;
                    if (g >= done && g < done + take)
                        queue[qlen + g - done] = (uint16_t)(((idx + (uint32_t)(q >> 2)) << 7) | ((uint32_t)(q & 3) << 5) | (uint32_t)i);
                    g++;
                }
            }
        }
        qlen += take;
        done += take;
        wave_lds_sync();
        uint32_t at = 0;
        for (; at + 64 <= qlen; at += 64) fn((uint32_t)queue[at + lane], true);
        if (at) {                                                     // the remainder to the front
            const uint16_t v = at + lane < (uint32_t)QN ? queue[at + lane] : (uint16_t)0;
            wave_lds_sync();
            qlen -= at;
            if ((uint32_t)lane < qlen) queue[lane] = v;
        }
        wave_lds_sync();
        if (done >= tot) break;
    }
}

// geometry of row r: reads [n_lo, n_lo + nreads) of batch bq at start l
struct RowGeom {
    int l;
    int64_t bq, nrow0;            // nrow0 = index of the row's first read in the super-batch
    uint32_t row_cells;
    uint32_t nreads;              // reads of the row (row_cells = nreads * nch)
    uint32_t nvalid;              // reads of the row that exist (the last batch may be short)
    const uint4* cells;
    size_t cell0;                 // index of the row's first cell in the cell array (compact entries: 2 per cell)
    const uint8_t* codes;         // the row's first read, advanced to the dword that holds start l
};
static __device__ __forceinline__ RowGeom row_geom(const FillArgs& a, int64_t r) {
    RowGeom g;
    const uint32_t part = (uint32_t)r % (uint32_t)a.parts;
    const uint32_t rl = (uint32_t)r / (uint32_t)a.parts;
    g.l = (int)(rl % (uint32_t)a.Lout);
    g.bq = rl / (uint32_t)a.Lout;
    const uint32_t n_lo = part * (uint32_t)a.rpr;
    const uint32_t nreads = (uint32_t)a.batch - n_lo < (uint32_t)a.rpr ? (uint32_t)a.batch - n_lo : (uint32_t)a.rpr;
    g.row_cells = nreads * (uint32_t)a.nch;
    g.nreads = nreads;
    // chunk groups: the row's cells of group 0 (group g's are g * Lout * batch * cgc cells further on)
    g.cell0 = a.cgc ? (((size_t)g.bq * a.ncg * a.Lout + g.l) * a.batch + n_lo) * a.cgc : (((size_t)g.bq * a.Lout + g.l) * a.batch + n_lo) * a.nch;
    g.cells = a.masks + g.cell0;
    g.nrow0 = g.bq * a.batch + n_lo;
    const int64_t left = a.N - g.nrow0;
    g.nvalid = left <= 0 ? 0u : (left < (int64_t)nreads ? (uint32_t)left : nreads);
    g.codes = a.codes + g.nrow0 * a.pitch + (g.l & ~3);
    return g;
}

// walk a row's cells and hand every candidate to fn(candidate word, live) 64 at a time.  CPL = cells per lane and step: with
// two (adjacent ones, so the candidates still come out in cell order) the popcount / wave scan / loop overhead of a step
// is shared by 128 cells.
template <int CPL, typename F>
static __device__ __forceinline__ void for_row_candidates(const RowGeom& g, uint16_t* queue, F&& fn) {
    const int lane = threadIdx.x & 63;
    constexpr uint32_t STEP = 64 * CPL;
    uint32_t qlen = 0;                                                // wave-uniform
    uint4 m_next[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        m_next[c] = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)(lane * CPL + c) < g.row_cells) m_next[c] = g.cells[lane * CPL + c];
    }
    for (uint32_t i0 = 0; i0 < g.row_cells; i0 += STEP) {             // wave-uniform trip count
        const uint32_t idx = i0 + lane * CPL;
        uint32_t wd[4 * CPL];
        uint32_t pc = 0;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint4 m = m_next[c];
            m_next[c] = make_uint4(0u, 0u, 0u, 0u);
            if (idx + c + STEP < g.row_cells) m_next[c] = g.cells[idx + c + STEP];
            wd[4 * c + 0] = m.x, wd[4 * c + 1] = m.y, wd[4 * c + 2] = m.z, wd[4 * c + 3] = m.w;
            pc += __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
        }
        const uint32_t inc = wave_incl_scan(pc);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (tot) push_and_drain<4 * CPL>(queue, qlen, wd, idx, inc - pc, tot, fn);
    }
    if (qlen) fn((uint32_t)queue[lane], (uint32_t)lane < qlen);      // the remainder (< 64)
    wave_lds_sync();
}

// The same walk over COMPACT entries (scan_cand_kernel_q<.., COMPACT>): one dword per cell = {entry of half 0, entry of half 1},
// entry = count << 12 | (63 - last) << 6 | first for up to two candidates among the half cell's 64 PWMs, 3 << 12 | popcount when
// there are more (their bits then sit in the cell array).  A lane takes 8 consecutive cells per step - 512 cells, a whole row, per
// step of the wave - so the candidates still come out in cell order; there is no per-bit loop: an entry yields its candidates with
// two predicated LDS stores.  The rare half cells with three or more candidates (1.4 % at BASELINE configs[1]) are fetched from
// the cell array: the loads of a lane's first two are issued before the other entries are written out, the rest on demand.
//
// The cells walked are `row_cells` consecutive ones from cell g.cell0 + base_off on (a row; with chunk groups a (row, group)
// sub-row, whose cells are contiguous in the group-major layout: read-major, chunk in group fastest).  MAP = 1 (emit_records_cg's
// slow path only): a whole row of a group-major layout in the reference's (read, chunk) order - index i stands for read i / nch,
// chunk i % nch, whose cell sits (chunk / cgc) * Lout * batch * cgc + read * cgc + chunk % cgc cells from the row's first.
template <int MAP, typename F>
static __device__ __forceinline__ void for_row_candidates_c(const FillArgs& a, const RowGeom& g, uint32_t row_cells, size_t base_off, uint16_t* queue, F&& fn) {
    const int lane = threadIdx.x & 63;
    constexpr uint32_t CPLC = 8, STEP = 64 * CPLC;
    uint32_t qlen = 0;                                                // wave-uniform
    const uint32_t* ent = (const uint32_t*)a.centries + g.cell0 + base_off;      // one dword per cell
    const uint4* cellsb = a.masks + g.cell0 + base_off;
    const size_t cg_stride = MAP ? (size_t)a.Lout * a.batch * a.cgc : 0;
    auto full = [&](uint32_t i) -> size_t {
        if (MAP == 0) return i;
        const uint32_t nin = a.div_nch.div(i), ch = i - nin * (uint32_t)a.nch;
        const uint32_t cgi = ch / (uint32_t)a.cgc;
        return (size_t)cgi * cg_stride + (size_t)nin * a.cgc + (ch - cgi * (uint32_t)a.cgc);
    };
    for (uint32_t i0 = 0; i0 < row_cells; i0 += STEP) {               // wave-uniform trip count (1 for rows of <= 512 cells)
        const uint32_t idx = i0 + lane * CPLC;
        uint32_t done = 0, tot = 0;                                   // wave-uniform
        // One turn unless the queue is short of room.  A turn starts from the entries in memory again (they come from L2 then), so
        // that nothing but a few scalars lives across the scoring calls of the drain below.
        do {
            uint32_t e[CPLC];
            if (MAP == 0 && idx + CPLC <= row_cells) {
                const uint4 v0 = *(const uint4*)(ent + idx), v1 = *(const uint4*)(ent + idx + 4);
                e[0] = v0.x, e[1] = v0.y, e[2] = v0.z, e[3] = v0.w, e[4] = v1.x, e[5] = v1.y, e[6] = v1.z, e[7] = v1.w;
            } else {
#pragma unroll
                for (int c = 0; c < (int)CPLC; c++) e[c] = idx + c < row_cells ? ent[full(idx + c)] : 0u;
            }
            // candidates of the lane: the two count fields of a dword are summed in packed form; a count of 3 stands for "popcount
            // in the low 12 bits" and is put right below
            uint32_t acc = 0, ovf = 0;                                // ovf: bit 2c + half set = that entry holds a popcount
#pragma unroll
            for (int c = 0; c < (int)CPLC; c++) {
                const uint32_t t = (e[c] >> 12) & 0x00030003u;
                acc += t;
                const uint32_t both = t & (t >> 1) & 0x00010001u;      // field == 3
                ovf |= ((both & 1u) | (both >> 15)) << (2 * c);
            }
            uint32_t pc = (acc & 0xffffu) + (acc >> 16);
            // Half cells with three or more candidates (1.4 % of them): their bits are fetched from the cell array.  A lane issues
            // the loads of its first two now (two cover 99.9 % of the lanes) and writes their candidates after everybody's ordinary
            // entries, so that the round trip hides behind that work and the bit loops run once per step for all such lanes
            // together instead of once per entry position.
            uint2 m0 = make_uint2(0u, 0u), m1 = make_uint2(0u, 0u);
            uint32_t j0 = 0, j1 = 0, g0 = 0, g1 = 0;
            if (ovf) {                                                // rare lanes
#pragma unroll
                for (int j = 0; j < 2 * (int)CPLC; j++)
                    if ((ovf >> j) & 1u) pc += ((e[j >> 1] >> (16 * (j & 1))) & 0xfffu) - 3u;
                j0 = (uint32_t)__builtin_ctz(ovf);
                m0 = ((const uint2*)(cellsb + full(idx + (j0 >> 1))))[j0 & 1];
                const uint32_t o1 = ovf & (ovf - 1u);
                if (o1) {
                    j1 = (uint32_t)__builtin_ctz(o1);
                    m1 = ((const uint2*)(cellsb + full(idx + (j1 >> 1))))[j1 & 1];
                }
            }
            const uint32_t inc = wave_incl_scan(pc);
            tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (!tot) break;
            const uint32_t room = QN - qlen;
            const uint32_t take = tot - done < room ? tot - done : room;
            uint16_t* qb = queue + qlen;
            uint32_t gp = inc - pc - done;                            // the lane's next candidate's place in this turn's window (may wrap: unsigned)
#pragma unroll
            for (int j = 0; j < 2 * (int)CPLC; j++) {
                const uint32_t f = (e[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                const uint32_t cnt = f >> 12;
                const uint32_t base = ((idx + (uint32_t)(j >> 1)) << 7) | ((uint32_t)(j & 1) << 6);
                if (cnt >= 1u && cnt < 3u) {
                    if (gp < take) qb[gp] = (uint16_t)(base | (f & 63u));
                }
                if (cnt == 2u) {
                    if (gp + 1u < take) qb[gp + 1u] = (uint16_t)(base | (63u - ((f >> 6) & 63u)));
                }
                if (cnt == 3u) {                                      // its candidates go in later: remember where
                    if ((uint32_t)j == j0) g0 = gp;
                    if ((uint32_t)j == j1) g1 = gp;
                    gp += f & 0xfffu;
                } else {
                    gp += cnt;
                }
            }
            if (__ballot(ovf != 0)) {
                auto spill = [&](uint2 mk, uint32_t j, uint32_t gq) {
                    uint64_t x = (uint64_t)mk.x | ((uint64_t)mk.y << 32);
                    const uint32_t base = ((idx + (j >> 1)) << 7) | ((j & 1u) << 6);
                    while (x) {
                        if (gq < take) qb[gq] = (uint16_t)(base | (uint32_t)__builtin_ctzll(x));
                        x &= x - 1;
                        gq++;
                    }
                };
                if (ovf) spill(m0, j0, g0);
                const uint32_t o1 = ovf & (ovf - 1u);
                if (__ballot(o1 != 0)) {
                    if (o1) spill(m1, j1, g1);
                    uint32_t o2 = o1 & (o1 - 1u);
                    if (__ballot(o2 != 0)) {                          // a lane with three or more such half cells: one by one
                        while (o2) {
                            const uint32_t j = (uint32_t)__builtin_ctz(o2);
                            o2 &= o2 - 1u;
                            uint32_t gq = inc - pc - done;            // the entries before j
                            for (uint32_t i = 0; i < j; i++) {
                                const uint32_t f = (ent[full(idx + (i >> 1))] >> (16 * (i & 1))) & 0xffffu;
                                gq += (f >> 12) == 3u ? (f & 0xfffu) : (f >> 12);
                            }
                            spill(((const uint2*)(cellsb + full(idx + (j >> 1))))[j & 1], j, gq);
                        }
                    }
                }
            }
            qlen += take;
            done += take;
            wave_lds_sync();
            uint32_t at = 0;
            for (; at + 64 <= qlen; at += 64) fn((uint32_t)queue[at + lane], true);
            if (at) {                                                 // the remainder to the front
                const uint16_t v = at + lane < (uint32_t)QN ? queue[at + lane] : (uint16_t)0;
                wave_lds_sync();
                qlen -= at;
                if ((uint32_t)lane < qlen) queue[lane] = v;
            }
            wave_lds_sync();
        } while (done < tot);
    }
    if (qlen) fn((uint32_t)queue[lane], (uint32_t)lane < qlen);      // the remainder (< 64)
    wave_lds_sync();
}

// exact score of one candidate word of row g; false if the candidate is not a hit
template <int LEN, bool LDS = false>
static __device__ __forceinline__ bool score_candidate(const FillArgs& a, const RowGeom& g, const _Float16* tb, uint32_t cw, bool live,
                                                       uint32_t& k, uint32_t& nin, uint16_t& sc) {
    const uint32_t idx = cw >> 7, q = (cw >> 5) & 3u, i = cw & 31u;
    nin = a.div_nch.div(idx);
    const uint32_t ch = idx - __umul24(nin, (uint32_t)a.nch);         // 24-bit multiplies are full rate, 32-bit ones a quarter
    k = (ch * 4 + q) * 32 + i;
    sc = 0;
    if (!(live && nin < g.nvalid && (int)k < a.K && (g.l <= a.lim_min || g.l <= a.lim[k]))) return false;
    if constexpr (LEN == 0) {      // run-time length: the window's bytes straight from the code row
        const uint8_t* win = g.codes + (size_t)nin * (size_t)a.pitch + (g.l & 3);
        sc = exact_score_dyn(tb + (size_t)k * (size_t)a.tabk_stride, win, a.L - a.lim[k]);
    } else {
        uint32_t W[LEN / 4 + 1];
        const uint32_t* sw = (const uint32_t*)(g.codes + __umul24(nin, (uint32_t)a.pitch));
#pragma unroll
        for (int j = 0; j <= LEN / 4; j++) W[j] = sw[j];
        sc = exact_score<LEN, LDS>(tb + __umul24(k, (uint32_t)a.tabk_stride), W, g.l);
    }
    return half_pos(sc);
}

// staged hit word: candidate word << 16 | binary16 score
// MODE: 0 = count the hits of every row; 1 = count and stage them; 2 = a17's dense tensor.
// MODE 2: max(score, 0) is +0 for every pair that is not a hit, and a row of cells (reads n_lo.. at one start l) is
// one contiguous run of the (K, N, ld_l) tensor, walked by the hits in ascending address order.  The wave
// therefore streams the run out through a 1 KB LDS window: scores of hits are dropped into the window, full
// windows leave as whole 128-byte lines, windows without hits leave as zeros.  Every byte of the tensor is
// written exactly once, in linear order: the pass is bound by the HBM write, as the dense contract says it
// should be.  Needs K % 8 == 0 (16-byte stores).
constexpr int DWIN = 2048;        // halves per window (4 KB: four 16-byte stores per lane and flush; 2-3 % faster than 1 KB windows)
// NW = waves per block: 8; 16 (one 1024-thread block per CU, four waves per SIMD) for a bank whose table needs more LDS than two
// blocks of 8 can share - 512 PWMs of 20 positions, 104 KB: with 8 waves per CU that form lost to gathers from L2, with 16 it does not
template <int LEN, bool LDS_TAB, int MODE, int NW = VF_WAVES>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 16 ? 4 : (LEN != 0 && LEN <= 32) ? 8 : 4, NW == 16 ? 4 : 8))) void stage_hits(FillArgs a0, FillArgs a1) {
    // blockIdx.y = strand: gpu_scan's two strands in one launch (a1 = a0 and gridDim.y = 1 for one)
    const FillArgs& a = blockIdx.y ? a1 : a0;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the row geometry stays on the scalar unit
    uint16_t* queue = (uint16_t*)smem + wv * QN;                      // [NW][QN]
    uint32_t* hist = smem + NW * QN / 2;                        // [hist_bins]
    uint32_t* winbase = hist + a.hist_bins;                           // MODE 2: [NW][DWIN] halves
    uint16_t* win = (uint16_t*)winbase + wv * DWIN;
    uint32_t* ltab = winbase + (MODE == 2 ? NW * DWIN / 2 : 0);
    for (int i = tid; i < a.hist_bins; i += (NW * 64)) hist[i] = 0;
    if (MODE == 2)
        for (int i = tid; i < NW * DWIN / 2; i += (NW * 64)) winbase[i] = 0u;
    const _Float16* tb = stage_table<LDS_TAB, (NW * 64)>(a, ltab);
    __syncthreads();
    // (rows handed out through one atomic counter instead of this fixed stride: 1.0 ms against 0.28 - ~85k device-scope
    // atomics on one address serialise at ~10 ns each)
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    for (int64_t r = (int64_t)wv * gridDim.x + blockIdx.x; r < a.nrows; r += nwaves) {
        const RowGeom g = row_geom(a, r);
        uint32_t* slots = MODE == 1 ? a.staging + (size_t)r * a.row_slots : nullptr;
        uint32_t nhit = 0;                                            // wave-uniform
        // MODE 2: the row's run of the tensor and the window over it
        uint16_t* seg = MODE == 2 ? a.dense + (size_t)a.K * ((size_t)g.nrow0 + (size_t)a.N * g.l) : nullptr;
        const uint32_t seg_len = MODE == 2 ? (g.row_cells / (uint32_t)a.nch) * (uint32_t)a.K : 0u;   // halves
        uint32_t win_lo = 0;                                          // wave-uniform
        bool dirty = false;                                           // the window holds a score
        auto flush = [&]() {                                          // window -> tensor, window back to zeros
            if (dirty) wave_lds_sync();
#pragma unroll
            for (int j = 0; j < DWIN / 8 / 64; j++) {
                const uint32_t u = j * 64 + lane, at = win_lo + u * 8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (dirty) {
                    v = ((const uint4*)win)[u];
                    ((uint4*)win)[u] = make_uint4(0u, 0u, 0u, 0u);
                }
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                if (at < seg_len) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, (u32x4*)(seg + at));
            }
            if (dirty) wave_lds_sync();
            dirty = false;
            win_lo += DWIN;
        };
        auto on_cand = [&](const uint32_t cw, const bool live) {
            uint32_t k, nin;
            uint16_t sc;
            bool hit = score_candidate<LEN, LDS_TAB>(a, g, tb, cw, live, k, nin, sc);
            const unsigned long long hb = __ballot(hit);
            if (hit) {
                if (MODE == 1) {
                    const uint32_t at = nhit + (uint32_t)__builtin_popcountll(hb & ((1ull << lane) - 1ull));
                    if (at < (uint32_t)a.row_slots) slots[at] = (cw << 16) | sc;
                }
                if (a.hist_bins) atomicAdd(&hist[k], 1u);
                else if (a.pwm_counts) atomicAdd((unsigned long long*)&a.pwm_counts[k], 1ull);
            }
            nhit += (uint32_t)__builtin_popcountll(hb);
            if (MODE == 2) {                                          // hits come in ascending offset order
                const uint32_t off = nin * (uint32_t)a.K + k;
                while (true) {                                        // wave-uniform
                    const bool in = hit && off < win_lo + DWIN;
                    if (in) win[off - win_lo] = sc;
                    if (__ballot(in)) dirty = true;
                    hit = hit && !in;
                    if (!__ballot(hit)) break;
                    flush();
                }
            }
        };
        // (mode 2 keeps the cells: its rows are 64 cells, 8 lanes' worth of entries - measured 0.49 of the HBM peak with entries
        // against 0.53 with cells on the same box)
        // (compact entries: only the cells of reads that exist are walked - the entries of the reads a short last batch lacks are never written)
        if (MODE != 2 && a.centries) for_row_candidates_c<0>(a, g, g.nvalid * (uint32_t)a.nch, (size_t)0, queue, on_cand);
        else for_row_candidates<(MODE == 2 ? 1 : 2)>(g, queue, on_cand);
        if (MODE == 2)
            while (win_lo < seg_len) flush();                         // the rest of the run
        if (MODE != 2 && lane == 0) a.row_sum[r] = nhit;
    }
    if (a.hist_bins) {
        __syncthreads();
        for (int i = tid; i < a.hist_bins; i += (NW * 64))
            if (hist[i]) atomicAdd((unsigned long long*)&a.pwm_counts[i], (unsigned long long)hist[i]);
    }
}

// ---- chunk groups: banks whose re-scoring table does not fit the LDS -------------------------------------------------------
// The table of 512 PWMs of 20 positions is 104 KB, of 2048 PWMs 418 KB: stage_hits<LEN, false, .> gathered it from L2 (2.2x slower per
// candidate than from LDS; 65 % of a BASELINE configs[4] step in round 3).  Here the bank is cut into GROUPS of CGC chunks of 128 PWMs
// whose table slice (26 KB per chunk at 20 positions) does fit, and a block serves ONE group for its whole life: it stages the slice
// once and walks the rows, visiting only its group's cells - CGC entry dwords per read, nch dwords apart.  Rows are 512 / CGC reads x
// all chunks, so a (row, group) sub-row is 512 cells, one step of the walk.  What the groups of a row find is in the reference's order
// only inside a group (read, PWM); the record order interleaves the groups read by read (findall's k runs fastest), and
// emit_records_cg restores it from the staged words themselves.  Counts and staging slots are per (row, group).
// Blocks -> (group, stripe of rows): the ncg blocks that walk the same rows are neighbours on ONE XCD (block b runs on XCD b % 8), so
// the 128-byte lines of entries they share - every group reads CGC * 4 bytes of each read's nch * 4 - come from that XCD's L2.
template <int LEN, int CGC>
static __device__ __forceinline__ bool score_candidate_cg(const FillArgs& a, const RowGeom& g, const _Float16* tbl, uint32_t cg0, uint32_t cw, bool live,
                                                          uint32_t& k, uint32_t& nin, uint16_t& sc) {
    constexpr uint32_t CGL = CGC == 4 ? 2 : CGC == 2 ? 1 : 0;
    const uint32_t idx = cw >> 7;
    nin = idx >> CGL;
    const uint32_t kl = ((idx & (uint32_t)(CGC - 1)) << 7) | (cw & 127u);     // PWM inside the group
    k = (cg0 << 7) + kl;
    sc = 0;
    if (!(live && nin < g.nvalid && (int)k < a.K && (g.l <= a.lim_min || g.l <= a.lim[k]))) return false;
    uint32_t W[LEN / 4 + 1];
    const uint32_t* sw = (const uint32_t*)(g.codes + __umul24(nin, (uint32_t)a.pitch));
#pragma unroll
    for (int j = 0; j <= LEN / 4; j++) W[j] = sw[j];
    sc = exact_score<LEN, true>(tbl + __umul24(kl, (uint32_t)a.tabk_stride), W, g.l);
    return half_pos(sc);
}

// MODE: 0 = count the hits of every (row, group); 1 = count and stage them.  A (row, group)'s staging block is
// [R / 2 dwords: hits per read, 16 bits each][row_slots staged words], R = 512 / CGC reads: the per-read counts are what
// emit_records_cg needs to interleave the groups, and the wave that scores the hits has them for one LDS atomic each.
// (NW = waves per block: 16 - one block per CU - for groups of 4 chunks, whose 104 KB slice leaves room for one block only)
template <int LEN, int CGC, int MODE, int NW = (CGC == 4 ? 16 : VF_WAVES)>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(CGC == 1 ? 8 : 4, CGC == 4 ? 4 : 8))) void stage_hits_cg(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr uint32_t R = 512 / CGC;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t* queue = (uint16_t*)smem + wv * QN;                      // [NW][QN]
    uint32_t* hist = smem + NW * QN / 2;                        // [CGC * 128]
    uint32_t* rcnt = hist + CGC * 128 + wv * (R / 2);                 // [NW][R / 2]: hits per read of the wave's sub-row
    uint32_t* ltab = hist + CGC * 128 + NW * (R / 2);
    const uint32_t ncg = (uint32_t)a.ncg;
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t cg = j % ncg, cg0 = cg * CGC;                      // first chunk of the block's group
    const int64_t stripe = (int64_t)(j / ncg) * 8 + xcd, nstripes = gridDim.x / ncg;   // gridDim.x is a multiple of 8 * ncg
    for (int i = tid; i < CGC * 128; i += (NW * 64)) hist[i] = 0;
    {   // the group's slice of the table (rows of PWMs past K do not exist: the last group may be short)
        const int k0 = (int)cg0 * 128;
        const int nk = a.K - k0 < CGC * 128 ? a.K - k0 : CGC * 128;
        const int ndw = nk > 0 ? nk * (a.tabk_stride / 2) : 0;
        const uint32_t* src = (const uint32_t*)(a.tabk + (size_t)k0 * a.tabk_stride);
        for (int i = tid; i < ndw; i += (NW * 64)) ltab[i] = src[i];
    }
    __syncthreads();
    const _Float16* tbl = (const _Float16*)ltab;
    const size_t sub_stride = (size_t)a.row_slots + R / 2;
    for (int64_t r = (int64_t)wv * nstripes + stripe; r < a.nrows; r += nstripes * NW) {
        const RowGeom g = row_geom(a, r);
        uint32_t* blk = MODE == 1 ? a.staging + ((size_t)r * ncg + cg) * sub_stride : nullptr;
        uint32_t* slots = MODE == 1 ? blk + R / 2 : nullptr;
        uint32_t nhit = 0;                                            // wave-uniform
        if (MODE == 1) {
            for (uint32_t i = lane; i < R / 2; i += 64) rcnt[i] = 0u;
            wave_lds_sync();
        }
        auto on_cand = [&](const uint32_t cw, const bool live) {
            uint32_t k, nin;
            uint16_t sc;
            const bool hit = score_candidate_cg<LEN, CGC>(a, g, tbl, cg0, cw, live, k, nin, sc);
            const unsigned long long hb = __ballot(hit);
            if (hit) {
                if (MODE == 1) {
                    const uint32_t at = nhit + (uint32_t)__builtin_popcountll(hb & ((1ull << lane) - 1ull));
                    if (at < (uint32_t)a.row_slots) slots[at] = (cw << 16) | sc;
                    atomicAdd(&rcnt[nin >> 1], 1u << (16 * (nin & 1u)));
                }
                if (a.pwm_counts) atomicAdd(&hist[k - (cg0 << 7)], 1u);
            }
            nhit += (uint32_t)__builtin_popcountll(hb);
        };
        for_row_candidates_c<0>(a, g, g.nvalid * (uint32_t)CGC, (size_t)cg * a.Lout * a.batch * CGC, queue, on_cand);
        if (MODE == 1 && nhit) {                                      // (an empty group's counts are never read)
            wave_lds_sync();
            for (uint32_t i = lane; i < R / 2; i += 64) blk[i] = rcnt[i];
        }
        if (lane == 0) a.row_sum[(size_t)r * ncg + cg] = nhit;
    }
    if (a.pwm_counts) {
        __syncthreads();
        for (int i = tid; i < CGC * 128; i += (NW * 64))
            if (hist[i]) atomicAdd((unsigned long long*)&a.pwm_counts[cg0 * 128 + i], (unsigned long long)hist[i]);
    }
}

// exclusive scan of the row counts in three small steps: per 1024 rows, over the block totals, (added back in emit_records)
// (ncg > 1, chunk groups: a row's count is the sum of its groups')
struct RowScan2 {                     // the row-scan arguments of one or two strands (blockIdx.y)
    int64_t* ticket_host[2];
    int64_t ticket[2];
    const uint32_t* row_sum[2];
    uint32_t* row_excl[2];
    unsigned long long* blk[2];
    const int64_t* base_in[2];
    int64_t* total_out[2];
    int64_t* total_host[2];
};
__global__ __launch_bounds__(1024) void row_scan_local(const RowScan2 rs, int64_t nrows, const int ncg) {
    const uint32_t* __restrict__ row_sum = rs.row_sum[blockIdx.y];
    uint32_t* __restrict__ row_excl = rs.row_excl[blockIdx.y];
    unsigned long long* __restrict__ blk_total = rs.blk[blockIdx.y];
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the row geometry stays on the scalar unit
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    uint32_t v = 0u;
    if (i < nrows) {
        if (ncg <= 1) v = row_sum[i];
        else
            for (int c = 0; c < ncg; c++) v += row_sum[i * ncg + c];
    }
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if (q < wv) wbase += wsum[q];
        tot += wsum[q];
    }
    if (i < nrows) row_excl[i] = wbase + inc - v;                     // < 2^32: at most 1024 rows x 65536 candidates (x 8 reads per row in chunk-group mode)
    if (tid == 0) blk_total[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void row_scan_blocks(const RowScan2 rs, int64_t nblk) {
    unsigned long long* __restrict__ blk = rs.blk[blockIdx.x];
    const int64_t* __restrict__ base_in = rs.base_in[blockIdx.x];
    int64_t* __restrict__ total_out = rs.total_out[blockIdx.x];
    int64_t* __restrict__ total_host = rs.total_host[blockIdx.x];
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (nblk + 1023) / 1024;
    int64_t lo = tid * per, hi = lo + per;
    if (lo > nblk) lo = nblk;
    if (hi > nblk) hi = nblk;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += blk[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long base = base_in ? (unsigned long long)*base_in : 0ull;
    unsigned long long run = part[tid] - s + base;
    for (int64_t i = lo; i < hi; i++) {
        const unsigned long long v = blk[i];
        blk[i] = run;                                                 // records before this block of rows
        run += v;
    }
    if (tid == 1023) {
        *total_out = (int64_t)(part[1023] + base);
        if (total_host) *total_host = (int64_t)(part[1023] + base);   // pinned host memory: no copy kernel behind this one
        if (rs.ticket_host[blockIdx.x]) {                             // ... and the word the host polls, behind the total
            __threadfence_system();
            __hip_atomic_store(rs.ticket_host[blockIdx.x], rs.ticket[blockIdx.x], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// staged words -> records.  One wave per row; a row that overflowed its staging slots is re-scored from its cells (with the table in
// L2: such rows are rare, and a table staged lazily in LDS needed a block barrier in every turn of this loop).  The waves are independent.
// A row of BASELINE configs[1] holds ~100 hits, i.e. one turn: the row's count, its offsets and its first 128 staged words are
// requested TOGETHER (the words speculatively: beyond the count they are whatever the slots held), one trip to memory per row
// instead of three dependent ones.
template <int LEN>
__global__ __launch_bounds__(VF_THREADS) void emit_records(FillArgs a0, FillArgs a1) {
    const FillArgs& a = blockIdx.y ? a1 : a0;                         // blockIdx.y = strand
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the row geometry stays on the scalar unit
    uint16_t* queue = (uint16_t*)smem + wv * QN;
    const _Float16* tb = (const _Float16*)a.tabk;
    const int64_t nwaves = (int64_t)gridDim.x * VF_WAVES;
    for (int64_t r = (int64_t)blockIdx.x * VF_WAVES + wv; r < a.nrows; r += nwaves) {
        const uint32_t* slots = a.staging + (size_t)r * a.row_slots;
        const uint32_t cnt = a.row_sum[r];
        const uint32_t excl = a.row_excl[r];
        const unsigned long long bbase = a.blk_base[r >> 10];
        uint32_t e[4];
        e[0] = lane < a.row_slots ? slots[lane] : 0u;
        e[1] = 64 + lane < a.row_slots ? slots[64 + lane] : 0u;
        if (cnt == 0) continue;
        const bool big = cnt > (uint32_t)a.row_slots;
        const RowGeom g = row_geom(a, r);
        const int64_t row_at = (int64_t)bbase + excl;
        auto put = [&](const int64_t at, const uint32_t k, const uint32_t nin, const uint16_t sc) {
            if (at < a.cap) {
                a.hits[at] = HitRec{k + 1, (uint32_t)(g.nrow0 + nin + a.n0 + 1), (uint32_t)(g.l + 1)};
                a.hit_scores[at] = sc;
            }
        };
        if (!big) {
            // four 256-byte loads of staged words in flight per wave (one at a time left the pass bound by load latency);
            // the capacity check is folded into the trip count
            const int64_t room = a.cap - row_at;
            const uint32_t lim = room <= 0 ? 0u : (room < (int64_t)cnt ? (uint32_t)room : cnt);
            HitRec* hrow = a.hits + row_at;
            uint16_t* srow = a.hit_scores + row_at;
            const uint32_t rec_n = (uint32_t)(g.nrow0 + a.n0 + 1), rec_l = (uint32_t)(g.l + 1);
            for (uint32_t j0 = 0; j0 < lim; j0 += 256) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    if (j0 != 0 || u >= 2) e[u] = j < lim ? slots[j] : 0u;        // (the first 128 words are already here)
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    const uint32_t idx = e[u] >> 23, q = (e[u] >> 21) & 3u, i = (e[u] >> 16) & 31u;
                    const uint32_t nin = a.div_nch.div(idx);
                    if (j < lim) {
                        hrow[j] = HitRec{((idx - nin * a.nch) * 4 + q) * 32 + i + 1, rec_n + nin, rec_l};
                        srow[j] = (uint16_t)e[u];
                    }
                }
            }
        } else {
            uint32_t nhit = 0;
            auto on_cand = [&](const uint32_t cw, const bool live) {
                uint32_t k, nin;
                uint16_t sc;
                const bool hit = score_candidate<LEN>(a, g, tb, cw, live, k, nin, sc);
                const unsigned long long hb = __ballot(hit);
                if (hit) put(row_at + nhit + (uint32_t)__builtin_popcountll(hb & ((1ull << lane) - 1ull)), k, nin, sc);
                nhit += (uint32_t)__builtin_popcountll(hb);
            };
            if (a.centries) for_row_candidates_c<0>(a, g, g.nvalid * (uint32_t)a.nch, (size_t)0, queue, on_cand);
            else for_row_candidates<2>(g, queue, on_cand);
        }
    }
}

// Chunk groups: staged words -> records.  One wave per row (512 / CGC reads x all chunks).  The groups of a row staged their hits apart,
// each list in (read, PWM) order; the record order is (read, group, PWM).  With c[g][n] = hits of read n in group g (the headers
// stage_hits_cg wrote), word j of group g - its read n known from the word - goes to D[n][g] + (j - S[g][n]), D = exclusive scan of c
// in (n, g) order, S[g] = exclusive scan of c[g] over n.  The wave turns c into E = D - S in LDS (16-bit arithmetic mod 2^16: a row on
// this path holds at most ncg * row_slots <= 16384 hits), so every word finds its place with one LDS read.
// The places of consecutive words of a list are a read's worth of records apart, and a store whose 64 lanes go to 64 different lines
// is 64 requests to the L2: writing the records straight from the lists ran at 1.2 TB/s (9-11 ms of a 33 ms BASELINE configs[4] step,
// 0.6 ms with the stores taken out).  So the words of 32 reads at a time are first dropped into an LDS window at their places (a
// 4-byte word + a 1-byte group), and the window then leaves as whole consecutive records, 64 lanes = 768 contiguous bytes, as
// emit_records writes them.  A row with an overflowed group is re-scored from its entries in the reference's order, as emit_records
// does it, 512 cells at a time with the table in L2 (rare: more than two hits per cell on average over 512 cells).
constexpr uint32_t CG_WIN = 768;      // records per LDS window (32 reads x 2048 PWMs at a 1 % hit rate: ~650)
template <int LEN, int CGC>
__global__ __launch_bounds__(VF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void emit_records_cg(FillArgs a, const int rpr_small) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr uint32_t R = 512 / CGC, CGL = CGC == 4 ? 2 : CGC == 2 ? 1 : 0, NH = R / 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ncg = (uint32_t)a.ncg;
    // per wave: the window (words, then group bytes; the slow path's queue shares it), E, the half-block starts
    uint32_t* W = smem + (size_t)wv * (CG_WIN + CG_WIN / 4);
    uint8_t* G = (uint8_t*)(W + CG_WIN);
    uint16_t* queue = (uint16_t*)W;
    uint32_t* M32 = smem + (size_t)VF_WAVES * (CG_WIN + CG_WIN / 4) + (size_t)wv * (R * ncg / 2);     // [ncg][R] 16-bit, two per dword
    uint16_t* M = (uint16_t*)M32;
    uint32_t* Sb = smem + (size_t)VF_WAVES * (CG_WIN + CG_WIN / 4) + (size_t)VF_WAVES * (R * ncg / 2) + (size_t)wv * ((NH + 1) * 17);   // [NH + 1][16 S + 1 D]
    const _Float16* tb = (const _Float16*)a.tabk;
    const int64_t rstep = (int64_t)gridDim.x * VF_WAVES;
    int64_t r = (int64_t)blockIdx.x * VF_WAVES + wv;
    uint32_t c_next = (r < a.nrows && (uint32_t)lane < ncg) ? a.row_sum[(size_t)r * ncg + lane] : 0u;
    for (; r < a.nrows; r += rstep) {
        const uint32_t c_l = c_next;                                  // the counts of the row after this one are on their way while it is written
        c_next = (r + rstep < a.nrows && (uint32_t)lane < ncg) ? a.row_sum[(size_t)(r + rstep) * ncg + lane] : 0u;
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(c_l), 63);
        if (tot == 0) continue;
        const bool big = __ballot(c_l > (uint32_t)a.row_slots) != 0;
        const RowGeom g = row_geom(a, r);
        const int64_t row_at = (int64_t)a.blk_base[r >> 10] + a.row_excl[r];
        const size_t sub_stride = (size_t)a.row_slots + R / 2;
        const uint32_t* blk0 = a.staging + (size_t)r * ncg * sub_stride;
        if (!big) {
            uint32_t jlo[8], jhi[8], e[8][2];
            const bool one_window = tot <= CG_WIN;                    // wave-uniform: the whole row fits the LDS window (sparse hits)
            if (one_window) {                                         // ... its words are asked for now, beside the headers: one trip less
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    jlo[u] = 0u;
                    jhi[u] = (uint32_t)u < ncg ? (uint32_t)__builtin_amdgcn_readlane((int)c_l, u) : 0u;
                    const uint32_t* slots = blk0 + (size_t)((uint32_t)u < ncg ? u : 0) * sub_stride + R / 2;
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const uint32_t j = 64 * v + lane;
                        e[u][v] = j < jhi[u] ? slots[j] : 0u;
                    }
                }
            }
            for (uint32_t cg = 0; cg < ncg; cg++) {                   // c[g][n]: the headers stage_hits_cg wrote (zeros for a group without hits)
                const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)c_l, (int)cg);
                const uint32_t* hdr = blk0 + (size_t)cg * sub_stride;
                for (uint32_t i = lane; i < R / 2; i += 64) M32[cg * (R / 2) + i] = cnt ? hdr[i] : 0u;
            }
            wave_lds_sync();
            uint32_t scar = 0, dcar = 0;                              // lane g: hits of group g in the reads before this block; hits before it
            for (uint32_t b0 = 0; b0 < R; b0 += 64) {
                const uint32_t n = b0 + lane;
                uint32_t t = 0;
                for (uint32_t cg = 0; cg < ncg; cg++) t += M[cg * R + n];
                const uint32_t inc = wave_incl_scan(t);
                uint32_t D = dcar + inc - t;
                dcar += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if ((lane & 31) == 0) Sb[((b0 >> 5) + (lane >> 5)) * 17 + 16] = D;          // where the half block's records start
                for (uint32_t cg = 0; cg < ncg; cg++) {
                    const uint32_t c = M[cg * R + n];
                    const uint32_t incg = wave_incl_scan(c);
                    const uint32_t S = (uint32_t)__builtin_amdgcn_readlane((int)scar, (int)cg) + incg - c;
                    if ((lane & 31) == 0) Sb[((b0 >> 5) + (lane >> 5)) * 17 + cg] = S;     // ... and its words in every group's list
                    M[cg * R + n] = (uint16_t)(D - S);
                    D += c;
                    const uint32_t totg = (uint32_t)__builtin_amdgcn_readlane((int)incg, 63);
                    if ((uint32_t)lane == cg) scar += totg;
                }
            }
            if (lane < 16) Sb[NH * 17 + lane] = scar;
            if (lane == 16) Sb[NH * 17 + 16] = dcar;
            wave_lds_sync();
            const int64_t room = a.cap - row_at;
            HitRec* hrow = a.hits + row_at;
            uint16_t* srow = a.hit_scores + row_at;
            const uint32_t rec_n = (uint32_t)(g.nrow0 + a.n0 + 1), rec_l = (uint32_t)(g.l + 1);
            const uint32_t dv = (uint32_t)lane <= NH ? Sb[lane * 17 + 16] : 0u;
            auto sb = [&](uint32_t h, uint32_t col) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)Sb[h * 17 + col]); };
            auto drop = [&](const uint32_t w0, const uint32_t cg, const uint32_t j, const uint32_t ew) {
                const uint32_t nin = (ew >> 23) >> CGL;
                const uint32_t dl = ((((uint32_t)M[cg * R + nin] + j) & 0xffffu) - w0) & 0xffffu;
#if defined(EXP_EMIT) && EXP_EMIT == 2
                if (dl < CG_WIN && ew == 0x12345u) {                           // experiment: no LDS scatter
#else
                if (dl < CG_WIN) {
#endif
                    W[dl] = ew;
                    G[dl] = (uint8_t)cg;
                }
            };
            auto copy_out = [&](const uint32_t w0, const uint32_t nw) {       // window -> records, consecutive lanes = consecutive records
#pragma unroll
                for (uint32_t i = 0; i < CG_WIN / 64; i++) {                 // (a fixed trip count: the stores behind a prefetch can be counted)
                    const uint32_t d = i * 64 + lane;
                    if (d < nw) {
                        const uint32_t ew = W[d], cg = G[d];
                        const uint32_t idx = ew >> 23, nin = idx >> CGL;
#if defined(EXP_EMIT) && EXP_EMIT == 1
                        if ((int64_t)(w0 + d) < room && ew == 0x12345u) {      // experiment: no record stores
#else
                        if ((int64_t)(w0 + d) < room) {
#endif
                            hrow[w0 + d] = HitRec{cg * (CGC * 128) + 1 + ((idx & (uint32_t)(CGC - 1)) << 7) + ((ew >> 16) & 127u), rec_n + nin, rec_l};
                            srow[w0 + d] = (uint16_t)ew;
                        }
                    }
                }
            };
            // A window = as many consecutive half blocks (32 reads) as fit CG_WIN records: the whole row where hits are sparse (one trip
            // to memory for all its words), one half block at BASELINE configs[4] density.  The words of the NEXT window are requested
            // before this one is written out, so the trip overlaps the stores.
            uint32_t hb = 0, he = 0, d_lo = 0, d_hi = 0;
            auto plan = [&](const uint32_t h0) {
                // the starts of every half block's records sit in lane h of dv; the window's word ranges come with ONE LDS read (lanes
                // 0-15: where the groups' lists stand at half block hb, lanes 16-31: at he) - read one value at a time through
                // readfirstlane, the ~26 dependent LDS round trips of a window cost as much as its trip to memory
                hb = h0;
                d_lo = (uint32_t)__builtin_amdgcn_readlane((int)dv, (int)h0);
                const unsigned long long fits = __ballot((uint32_t)lane > h0 && (uint32_t)lane <= NH && dv - d_lo <= CG_WIN);
                he = fits ? 63u - (uint32_t)__builtin_clzll(fits) : h0 + 1;
                d_hi = (uint32_t)__builtin_amdgcn_readlane((int)dv, (int)he);
                const uint32_t x = Sb[(lane < 16 ? hb : he) * 17 + (lane & 15)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t cg = (uint32_t)u < ncg ? (uint32_t)u : 0u;
                    jlo[u] = (uint32_t)__builtin_amdgcn_readlane((int)x, u);
                    jhi[u] = (uint32_t)u < ncg ? (uint32_t)__builtin_amdgcn_readlane((int)x, 16 + u) : jlo[u];
                    const uint32_t* slots = blk0 + (size_t)cg * sub_stride + R / 2;
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const uint32_t j = jlo[u] + 64 * v + lane;
                        e[u][v] = j < jhi[u] ? slots[j] : 0u;
                    }
                }
            };
            if (one_window) hb = 0, he = NH, d_lo = 0, d_hi = tot;
            else plan(0);
            while (true) {
                const uint32_t c_hb = hb, c_he = he, c_lo = d_lo, c_hi = d_hi;
                if (c_hi - c_lo <= CG_WIN) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
#pragma unroll
                        for (int v = 0; v < 2; v++) {
                            const uint32_t j = jlo[u] + 64 * v + lane;
                            if (j < jhi[u]) drop(c_lo, (uint32_t)u, j, e[u][v]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {                    // what the 16 requested loads did not cover (wave-uniform, rare)
                        if (jhi[u] - jlo[u] > 128u) {
                            const uint32_t* slots = blk0 + (size_t)u * sub_stride + R / 2;
                            for (uint32_t j = jlo[u] + 128 + lane; j < jhi[u]; j += 64) drop(c_lo, (uint32_t)u, j, slots[j]);
                        }
                    }
                    for (uint32_t cg = 8; cg < ncg; cg++) {          // groups past the eighth (banks of more than 4096 PWMs)
                        const uint32_t lo = sb(c_hb, cg), hi = sb(c_he, cg);
                        const uint32_t* slots = blk0 + (size_t)cg * sub_stride + R / 2;
                        for (uint32_t j = lo + lane; j < hi; j += 64) drop(c_lo, cg, j, slots[j]);
                    }
                    if (c_he < NH) plan(c_he);
                    wave_lds_sync();
                    copy_out(c_lo, c_hi - c_lo);
                    wave_lds_sync();
                } else {                                             // 32 reads with more than CG_WIN hits: window by window, every word re-read per window
                    for (uint32_t w0 = c_lo; w0 < c_hi; w0 += CG_WIN) {
                        for (uint32_t cg = 0; cg < ncg; cg++) {
                            const uint32_t lo = sb(c_hb, cg), hi = sb(c_he, cg);
                            const uint32_t* slots = blk0 + (size_t)cg * sub_stride + R / 2;
                            for (uint32_t j = lo + lane; j < hi; j += 64) drop(w0, cg, j, slots[j]);
                        }
                        wave_lds_sync();
                        copy_out(w0, c_hi - w0 < CG_WIN ? c_hi - w0 : CG_WIN);
                        wave_lds_sync();
                    }
                    if (c_he < NH) plan(c_he);
                }
                if (c_he >= NH) break;
            }
        } else {
            uint32_t nhit = 0;
            for (uint32_t n_off = 0; n_off < g.nreads; n_off += (uint32_t)rpr_small) {       // wave-uniform
                RowGeom g2 = g;
                g2.nreads = g.nreads - n_off < (uint32_t)rpr_small ? g.nreads - n_off : (uint32_t)rpr_small;
                g2.row_cells = g2.nreads * (uint32_t)a.nch;
                g2.nrow0 = g.nrow0 + n_off;
                g2.cell0 = g.cell0 + (size_t)n_off * a.cgc;         // group-major layout: the reads of group 0 are cgc cells apart
                g2.cells = a.masks + g2.cell0;
                g2.nvalid = g.nvalid > n_off ? (g.nvalid - n_off < g2.nreads ? g.nvalid - n_off : g2.nreads) : 0u;
                g2.codes = g.codes + (size_t)n_off * a.pitch;
                auto on_cand = [&](const uint32_t cw, const bool live) {
                    uint32_t k, nin;
                    uint16_t sc;
                    const bool hit = score_candidate<LEN>(a, g2, tb, cw, live, k, nin, sc);
                    const unsigned long long hb = __ballot(hit);
                    if (hit) {
                        const int64_t at = row_at + nhit + (uint32_t)__builtin_popcountll(hb & ((1ull << lane) - 1ull));
                        if (at < a.cap) {
                            a.hits[at] = HitRec{k + 1, (uint32_t)(g2.nrow0 + nin + a.n0 + 1), (uint32_t)(g.l + 1)};
                            a.hit_scores[at] = sc;
                        }
                    }
                    nhit += (uint32_t)__builtin_popcountll(hb);
                };
                for_row_candidates_c<1>(a, g2, g2.nvalid * (uint32_t)a.nch, (size_t)0, queue, on_cand);
            }
        }
    }
}

// the four-reads kernel's one-hot images (16 per block) against the LDS its blocks per CU leave each other
static size_t cand_q_lds_cap(int wpe) { return (size_t)(160 * 1024 / wpe) - 1024; }
// Segments of window tiles per read (grid z of scan_cand_kernel_q).  More than one when (a) the whole images do not fit beside the
// other blocks of the CU (reads past ~600 positions at two blocks per CU: at BASELINE configs[4]'s 1000 positions the one-read
// kernel ran before, 11.9 ms of candidates per 25 000 reads against 10.1 with two segments), or (b) the launch does not fill the CU
// slots once (1 024 reads: 28 -> 15 us; 4 096: 41 -> 38).  Past one round segments cost more than they balance (12 500 reads, 1 563
// blocks on 1 024 slots: 96 us with one segment, 100 with four - a first round takes 61 us where a later one takes 45 because its
// blocks stage and multiply in step, and segments do not change that).
static int cand_q_segments(const CandDims& d, int lenp, int wpe, int64_t blocks) {
    const int nt = (d.Lout + 7) / 8;
    auto lds_of = [&](int ns) {
        const int st = (nt + ns - 1) / ns;
        return (size_t)4 * 4 * quad_pitch(ns == 1 ? d.ohlen : st * 8 + lenp) * 8;
    };
    int ns = 1;
    while (ns < nt && lds_of(ns) > cand_q_lds_cap(wpe)) ns++;
    const int64_t slots = (int64_t)256 * wpe;
    while (blocks * (ns + 1) <= slots && ns < 8 && (nt + ns) / (ns + 1) >= 4) ns++;
    return ns;
}
// a candidate launch: with events attached the extended launch (they take the kernel's own time stamps), otherwise a plain one
#define CAND_LAUNCH(kern, grid, block, lds, ...)                                                                      \
    do {                                                                                                               \
        if (ev0 || ev1) hipExtLaunchKernelGGL(kern, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__);                  \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                             \
    } while (0)
template <int T, int PG>
static hipError_t launch_cand_tp(const CandArgs& a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const int ntg = (a.d.used_tiles + PG - 1) / PG;           // tile groups that hold PWMs
    const int tgb = (a.uniform_eps && ntg == 1) ? 1 : 2;
    // four reads per wave, the LDS images leaving the CU as many blocks as the registers do (4, 3 or 2 per CU)
    constexpr int wpe = (T * PG <= 12) ? 4 : (T * PG <= CAND_Q_W3 && PG > 1) ? 3 : 2;
    if (a.uniform_eps) {
        CandDims d = a.d;
        const bool compact = PG == 4 && a.centries != nullptr;
        // quads per wave: many small blocks balance the CUs best (N = 100k: 1 or 2 per wave 0.337 ms, 3: 0.354, 8: 0.390)
        d.spw = (int)std::max<int64_t>(1, std::min<int64_t>(8, a.d.N / (16 * 8192)));
        const int64_t per_block = (int64_t)(4 / tgb) * 4 * d.spw;
        const unsigned gx = (unsigned)((d.N + per_block - 1) / per_block), gy = (unsigned)((ntg + tgb - 1) / tgb);
        d.nseg = cand_q_segments(a.d, 4 * T, wpe, (int64_t)gx * gy);
        if (d.nseg > 1) {
            const int nt = (d.Lout + 7) / 8;
            d.seg_tiles = (nt + d.nseg - 1) / d.nseg;
            d.nseg = (nt + d.seg_tiles - 1) / d.seg_tiles;
            d.ohseg = d.seg_tiles * 8 + 4 * T;
        }
        const size_t lds_q = (size_t)4 * 4 * quad_pitch(d.ohseg) * 8;
        dim3 grid(gx, gy, (unsigned)d.nseg);
        if (lds_q > 64 * 1024) {                   // (reads of ~500 positions at two blocks per CU: 65 KB of one-hot images per block)
            if constexpr (PG == 4) {
                (void)hipFuncSetAttribute((const void*)scan_cand_kernel_q<T, PG, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
                (void)hipFuncSetAttribute((const void*)scan_cand_kernel_q<T, PG, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            }
            (void)hipFuncSetAttribute((const void*)scan_cand_kernel_q<T, PG, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            (void)hipFuncSetAttribute((const void*)scan_cand_kernel_q<T, PG, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        }
        if (compact) {
            if constexpr (PG == 4) {
                if (tgb == 1) CAND_LAUNCH((scan_cand_kernel_q<T, PG, 1, true>), grid, dim3(256), (uint32_t)lds_q, a.afrag, a.codes, a.cells, a.centries, d, a.afrag2, a.cells2, a.centries2);
                else CAND_LAUNCH((scan_cand_kernel_q<T, PG, 2, true>), grid, dim3(256), (uint32_t)lds_q, a.afrag, a.codes, a.cells, a.centries, d, a.afrag2, a.cells2, a.centries2);
            }
        } else if (tgb == 1) CAND_LAUNCH((scan_cand_kernel_q<T, PG, 1, false>), grid, dim3(256), (uint32_t)lds_q, a.afrag, a.codes, a.cells, (uint16_t*)nullptr, d, (const uint4*)nullptr, (uint32_t*)nullptr, (uint16_t*)nullptr);
        else CAND_LAUNCH((scan_cand_kernel_q<T, PG, 2, false>), grid, dim3(256), (uint32_t)lds_q, a.afrag, a.codes, a.cells, (uint16_t*)nullptr, d, (const uint4*)nullptr, (uint32_t*)nullptr, (uint16_t*)nullptr);
        return hipGetLastError();
    }
    // banks without a uniform slack: one read per wave, cells only
    const int64_t per_block = (int64_t)4 * a.d.spw;
    dim3 grid((unsigned)((a.d.N + per_block - 1) / per_block), (unsigned)((ntg + tgb - 1) / tgb), 1);
    const size_t lds = (size_t)tgb * 4 * ((a.d.ohlen + 3) & ~3) * 8;
    if (a.afrag2 || a.centries) return hipErrorInvalidValue;       // two strands per launch / compact entries: the four-reads kernel only (cand_compact_ok)
    CAND_LAUNCH((scan_cand_kernel<T, PG>), grid, dim3(512), (uint32_t)lds, a.afrag, a.cinit, a.codes, a.cells, a.d);
    return hipGetLastError();
}

// tiles of 32 PWMs a wave carries: its A fragments are PG * lenp / 4 registers x 4
int cand_tile_group(int lenp) { return lenp <= 20 ? 4 : lenp <= 32 ? 2 : 1; }

// compact entries are written by the kernels for banks scaled to one slack, with tile groups of 4 (PWMs of up to 20 positions)
bool cand_compact_ok(const CandArgs& a) { return a.uniform_eps && a.lenp <= 20; }
// both strands of gpu_scan in one launch: the four-reads-per-wave kernel with compact entries (the condition of launch_cand_tp)
bool cand_two_strands_ok(const CandArgs& a) {
    return cand_compact_ok(a);           // (reads of any length: the kernel's blocks take segments of them)
}

// ev0 / ev1 (optional): events that take the kernel's own start and stop time stamps (hipExtLaunchKernelGGL; without events a plain launch -
// the extended launch leaves a ~9 us hole behind the kernel whether or not it stamps anything): timing the
// dominant kernel then puts no extra packets on the stream (an event recorded before and after cost ~5 us each per launch)
// run-time length: one tile per wave, as many waves per block as the one-hot images leave room for
static hipError_t launch_cand_generic(const CandArgs& a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t per_wave = (size_t)((a.d.ohlen + 3) & ~3) * 8;
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
    if (per_wave * wpb > 160 * 1024 || a.d.used_tiles > 65535) return hipErrorInvalidValue;   // a read of > ~20k positions
    CandDims d = a.d;
    d.spw = (int)std::max<int64_t>(1, std::min<int64_t>(8, a.d.N * a.d.used_tiles / 65536));
    const int64_t per_block = (int64_t)wpb * d.spw;
    dim3 grid((unsigned)((d.N + per_block - 1) / per_block), (unsigned)d.used_tiles, 1);
    const size_t lds = per_wave * wpb;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)scan_cand_kernel_g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CAND_LAUNCH(scan_cand_kernel_g, grid, dim3(64 * wpb), (uint32_t)lds, a.afrag, a.cinit, a.codes, a.cells, d, a.lenp / 4,
                          a.uniform_eps);
    return hipGetLastError();
}

hipError_t launch_cand(const CandArgs& a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if (a.lenp > 64) return launch_cand_generic(a, st, ev0, ev1);
    switch (a.lenp) {
        case 8: return launch_cand_tp<2, 4>(a, st, ev0, ev1);
        case 12: return launch_cand_tp<3, 4>(a, st, ev0, ev1);
        case 16: return launch_cand_tp<4, 4>(a, st, ev0, ev1);
        case 20: return launch_cand_tp<5, 4>(a, st, ev0, ev1);
        case 24: return launch_cand_tp<6, 2>(a, st, ev0, ev1);
        case 32: return launch_cand_tp<8, 2>(a, st, ev0, ev1);
        case 40: return launch_cand_tp<10, 1>(a, st, ev0, ev1);
        case 48: return launch_cand_tp<12, 1>(a, st, ev0, ev1);
        case 64: return launch_cand_tp<16, 1>(a, st, ev0, ev1);
        default: return hipErrorInvalidValue;
    }
}

int stage_row_reads(int nch) { return std::max(1, ROW_CELLS_MAX / nch); }
// a bank whose table does not fit beside the queues of an 8-wave block (64 KB) but does fit ONE 16-wave block per CU
bool stage_big_lds_ok(int K, int KP, int tabk_stride, bool hist) {
    const size_t tab_bytes = ((size_t)K * tabk_stride * 2 + 3) & ~(size_t)3;
    const size_t hist_bytes = (hist && 2 * KP <= FILL_HIST_MAX) ? (size_t)2 * KP * 4 : 0;
    return (size_t)VF_WAVES * QN * 2 + hist_bytes + tab_bytes > 64 * 1024 && (size_t)16 * QN * 2 + hist_bytes + tab_bytes <= 160 * 1024 - 1024;
}
// Dense mode: short runs (64 cells: 16 reads x 400 B at K = 200) keep the write streams of the waves that run
// together close to each other in memory; measured 0.36 ms per 1.5 GB against 0.40 with 512-cell rows.
int dense_row_reads(int nch) { return std::max(1, 64 / nch); }

template <int LEN, int MODE>
static hipError_t launch_stage_mode(const FillArgs& a, hipStream_t st, const FillArgs* b) {
    if ((int64_t)a.rpr * a.nch > ROW_CELLS_MAX || a.nrows >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const size_t base = (size_t)VF_WAVES * QN * 2 + (size_t)a.hist_bins * 4 + (MODE == 2 ? (size_t)VF_WAVES * DWIN * 2 : 0);
    const size_t tab_bytes = ((size_t)a.K * a.tabk_stride * 2 + 3) & ~(size_t)3;
    const bool lds_tab = base + tab_bytes <= 64 * 1024;
    if (MODE == 2 && a.K % 8 != 0) return hipErrorInvalidValue;
    // (one round of blocks: each stages the table first; 768 blocks 0.421 ms at configs[1], 1024: 0.386, 2048: 0.394)
    const unsigned grid = (unsigned)std::min<int64_t>((a.nrows + VF_WAVES - 1) / VF_WAVES, 256 * 4);
    // b != nullptr: the other strand's rows in the same launch (same bank shape, same geometry: grid.y = 2, half the blocks each)
    const dim3 g2(b ? std::max(1u, (grid + 1) / 2) : grid, b ? 2 : 1, 1);
    const FillArgs& a1 = b ? *b : a;
    if (lds_tab) {
        hipLaunchKernelGGL((stage_hits<LEN, true, MODE>), g2, dim3(VF_THREADS), base + tab_bytes, st, a, a1);
    } else if (MODE != 2 && LEN != 0 && LEN <= 32 && stage_big_lds_ok(a.K, a.KP, a.tabk_stride, a.hist_bins != 0)) {
        // the whole table in the LDS of ONE 16-wave block per CU
        const size_t lds16 = (size_t)16 * QN * 2 + (size_t)a.hist_bins * 4 + tab_bytes;
        const unsigned grid16 = (unsigned)std::min<int64_t>((a.nrows + 15) / 16, 256);
        const dim3 g16(b ? std::max(1u, (grid16 + 1) / 2) : grid16, b ? 2 : 1, 1);
        auto kern = stage_hits<LEN, true, (MODE == 2 ? 1 : MODE), 16>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(kern, g16, dim3(1024), lds16, st, a, a1);
    } else {
        hipLaunchKernelGGL((stage_hits<LEN, false, MODE>), g2, dim3(VF_THREADS), base, st, a, a1);
    }
    return hipGetLastError();
}
template <int LEN>
static hipError_t launch_stage_len(const FillArgs& a, int mode, hipStream_t st, const FillArgs* b) {
    return mode == 0 ? launch_stage_mode<LEN, 0>(a, st, b) : mode == 1 ? launch_stage_mode<LEN, 1>(a, st, b) : launch_stage_mode<LEN, 2>(a, st, b);
}
template <int LEN>
static hipError_t launch_emit_len(const FillArgs& a, hipStream_t st, const FillArgs* b) {
    const size_t base = (size_t)VF_WAVES * QN * 2;
    // a wave per row, no walk over several rows: a row is a chain of trips to memory (count and offset, staged words, records) and only more waves
    // in flight cover it (2048 blocks: 0.228 ms at configs[1], 4096: 0.214, one per 8 rows - 9.3k blocks - 0.204)
    const unsigned grid = (unsigned)std::min<int64_t>((a.nrows + VF_WAVES - 1) / VF_WAVES, (int64_t)1 << 22);
    const dim3 g2(b ? std::max(1u, (grid + 1) / 2) : grid, b ? 2 : 1, 1);
    const FillArgs& a1 = b ? *b : a;
    hipLaunchKernelGGL((emit_records<LEN>), g2, dim3(VF_THREADS), base, st, a, a1);
    return hipGetLastError();
}

// ---- chunk-group launchers
static int stage_cg_waves(int cgc) { return cgc == 4 ? 16 : VF_WAVES; }
size_t stage_cg_lds_bytes(int cgc, int tabk_stride) {
    const size_t nw = (size_t)stage_cg_waves(cgc);
    return nw * QN * 2 + (size_t)cgc * 128 * 4 + nw * (512 / cgc) * 2 + (((size_t)cgc * 128 * tabk_stride * 2 + 3) & ~(size_t)3);
}
int stage_cg_chunks(int K, int nch, int lenp, int tabk_stride, int want) {
    if (lenp > 20 || lenp % 4 != 0) return 0;                          // compact entries exist for PWMs of up to 20 positions
    auto ok = [&](int cgc) {
        // the slice beside the queues in the LDS a CU has; emit_records_cg's count matrix (512 / cgc reads x groups, 16-bit) at most 4 KB per wave
        return nch % cgc == 0 && stage_cg_lds_bytes(cgc, tabk_stride) <= 160 * 1024 - 1024 && (512 / cgc) * ((nch + cgc - 1) / cgc) <= 2048;
    };
    if (want > 0) return (want == 1 || want == 2 || want == 4) && ok(want) ? want : 0;
    const size_t whole = (size_t)VF_WAVES * QN * 2 + (size_t)2 * nch * 64 * 4 + (((size_t)K * tabk_stride * 2 + 3) & ~(size_t)3);
    if (whole <= 64 * 1024) return 0;                                  // stage_hits<LEN, true, .> holds the whole table
    if (stage_big_lds_ok(K, nch * 64, tabk_stride, true)) return 0;    // ... or one 16-wave block per CU does (no reordering of the records needed)
    for (int cgc : {2, 1, 4})
        if (ok(cgc) && (cgc == 4 || stage_cg_lds_bytes(cgc, tabk_stride) <= 64 * 1024)) return cgc;
    return 0;
}
template <int LEN, int CGC>
static hipError_t launch_stage_cg_t(const FillArgs& a, int mode, hipStream_t st) {
    const size_t lds = stage_cg_lds_bytes(CGC, a.tabk_stride);
    const int nw = stage_cg_waves(CGC);
    const int bpc = (int)std::max<size_t>(1, std::min<size_t>(32 / nw, (160 * 1024) / (lds + 512)));     // blocks a CU holds
    const int64_t unit = 8 * (int64_t)a.ncg;                            // one block per (XCD, group)
    int64_t grid = std::max<int64_t>(1, (256 * (int64_t)bpc) / unit) * unit;
    const int64_t need = ((a.nrows + nw - 1) / nw + 7) / 8 * unit;      // stripes that have a row, in whole units
    grid = std::max<int64_t>(unit, std::min(grid, need));
    if (a.nrows >= (int64_t)1 << 31) return hipErrorInvalidValue;
    auto k0 = stage_hits_cg<LEN, CGC, 0>;
    auto k1 = stage_hits_cg<LEN, CGC, 1>;
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (mode == 0) hipLaunchKernelGGL(k0, dim3((unsigned)grid), dim3(nw * 64), lds, st, a);
    else hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(nw * 64), lds, st, a);
    return hipGetLastError();
}
template <int LEN>
static hipError_t launch_stage_cg_len(const FillArgs& a, int mode, hipStream_t st) {
    return a.cgc == 1 ? launch_stage_cg_t<LEN, 1>(a, mode, st) : a.cgc == 2 ? launch_stage_cg_t<LEN, 2>(a, mode, st) : launch_stage_cg_t<LEN, 4>(a, mode, st);
}
template <int LEN, int CGC>
static hipError_t launch_emit_cg_t(const FillArgs& a, hipStream_t st) {
    const size_t lds = (size_t)VF_WAVES * (CG_WIN + CG_WIN / 4) * 4 + (size_t)VF_WAVES * (512 / CGC) * a.ncg * 2 + (size_t)VF_WAVES * (512 / CGC / 32 + 1) * 17 * 4;
    if (lds > 160 * 1024 - 1024 || a.ncg > 16) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)std::min<int64_t>((a.nrows + VF_WAVES - 1) / VF_WAVES, 256 * 16);    // (1024 / 2048 / 4096 / 16384 blocks: 4.54 / 4.10 / 3.97 / 4.00 ms at configs[4])
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)emit_records_cg<LEN, CGC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((emit_records_cg<LEN, CGC>), dim3(grid), dim3(VF_THREADS), lds, st, a, stage_row_reads(a.nch));
    return hipGetLastError();
}
template <int LEN>
static hipError_t launch_emit_cg_len(const FillArgs& a, hipStream_t st) {
    return a.cgc == 1 ? launch_emit_cg_t<LEN, 1>(a, st) : a.cgc == 2 ? launch_emit_cg_t<LEN, 2>(a, st) : launch_emit_cg_t<LEN, 4>(a, st);
}
static hipError_t launch_stage_cg(const FillArgs& a, int mode, hipStream_t st) {
    if (mode > 1 || !a.centries) return hipErrorInvalidValue;
    switch (a.lenp) {
        case 8: return launch_stage_cg_len<8>(a, mode, st);
        case 12: return launch_stage_cg_len<12>(a, mode, st);
        case 16: return launch_stage_cg_len<16>(a, mode, st);
        case 20: return launch_stage_cg_len<20>(a, mode, st);
        default: return hipErrorInvalidValue;
    }
}
static hipError_t launch_emit_cg(const FillArgs& a, hipStream_t st) {
    switch (a.lenp) {
        case 8: return launch_emit_cg_len<8>(a, st);
        case 12: return launch_emit_cg_len<12>(a, st);
        case 16: return launch_emit_cg_len<16>(a, st);
        case 20: return launch_emit_cg_len<20>(a, st);
        default: return hipErrorInvalidValue;
    }
}

#define MOTIFS_LEN_SWITCH(lenp, CALL)       \
    switch (lenp) {                         \
        case 8: return CALL(8);             \
        case 12: return CALL(12);           \
        case 16: return CALL(16);           \
        case 20: return CALL(20);           \
        case 24: return CALL(24);           \
        case 32: return CALL(32);           \
        case 40: return CALL(40);           \
        case 48: return CALL(48);           \
        case 64: return CALL(64);           \
        default: return (lenp) > 64 ? CALL(0) : hipErrorInvalidValue;   /* LEN = 0: run-time length */ \
    }

hipError_t launch_stage_hits(const FillArgs& a, int mode, hipStream_t st, const FillArgs* b) {
    if (a.cgc) return b ? hipErrorInvalidValue : launch_stage_cg(a, mode, st);
    if (b && mode == 2) return hipErrorInvalidValue;
#define CALL(LEN) launch_stage_len<LEN>(a, mode, st, b)
    MOTIFS_LEN_SWITCH(a.lenp, CALL)
#undef CALL
}
hipError_t launch_row_scan(const FillArgs& a, hipStream_t st, const FillArgs* b) {
    const int64_t nblk = (a.nrows + 1023) / 1024;
    RowScan2 rs{};
    const FillArgs* f[2] = {&a, b ? b : &a};
    for (int i = 0; i < 2; i++) {
        rs.row_sum[i] = f[i]->row_sum, rs.row_excl[i] = f[i]->row_excl, rs.blk[i] = f[i]->blk_base;
        rs.base_in[i] = f[i]->base_in, rs.total_out[i] = f[i]->total, rs.total_host[i] = f[i]->total_host;
        rs.ticket_host[i] = f[i]->ticket_host, rs.ticket[i] = f[i]->ticket;
    }
    hipLaunchKernelGGL(row_scan_local, dim3((unsigned)nblk, b ? 2 : 1, 1), dim3(1024), 0, st, rs, a.nrows, a.cgc ? a.ncg : 1);
    hipLaunchKernelGGL(row_scan_blocks, dim3(b ? 2 : 1), dim3(1024), 0, st, rs, nblk);
    return hipGetLastError();
}
hipError_t launch_emit_records(const FillArgs& a, hipStream_t st, const FillArgs* b) {
    if (a.cgc) return b ? hipErrorInvalidValue : launch_emit_cg(a, st);
#define CALL(LEN) launch_emit_len<LEN>(a, st, b)
    MOTIFS_LEN_SWITCH(a.lenp, CALL)
#undef CALL
}

}  // namespace motifs
