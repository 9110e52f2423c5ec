// post_kernels.hip — the consumers and the producer on either side of the scan:
//   * FASTA text -> base codes                      (src/loadfasta/helpers.jl:83-139)
//   * per-PWM score range, threshold sweep counts and threshold filtering of hit records
//                                                   (src/inference/_s2_filter_pos_w_scores.jl:3-35, :103-125)
//   * hit positions -> count matrices                (src/inference/_h6_positions2countmat.jl:26-55)
// The device parts work on the record arrays motifs_pwm_scan_hits_dev leaves in HBM.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <climits>
#include <functional>
#include <thread>

#include "api_common.h"
#include "scan_kernels.h"

using namespace motifs;

static __device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }

// ---- score range ----------------------------------------------------------------------------------
// order-preserving 16-bit key of a binary16 value (so integer atomicMin/Max follow the float order)
static __device__ __forceinline__ uint32_t hkey(uint16_t h) { return (h & 0x8000u) ? (uint16_t)~h : (uint16_t)(h | 0x8000u); }
static __device__ __forceinline__ uint16_t hunkey(uint32_t k) { return (k & 0x8000u) ? (uint16_t)(k & 0x7fffu) : (uint16_t)~k; }

__global__ void k_minmax_init(uint32_t* kmin, uint32_t* kmax, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        kmin[i] = hkey(0x7c00u);   // +Inf: get_min_score starts there (:25)
        kmax[i] = hkey(0xfc00u);   // -Inf: get_max_score (:12)
    }
}
__global__ void k_minmax(const HitRec* hits, const uint16_t* scores, int64_t n, uint32_t* kmin, uint32_t* kmax) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = hits[i].m - 1, k = hkey(scores[i]);
        if (k < kmin[m]) atomicMin(&kmin[m], k);      // plain pre-check: most records do not move the bound
        if (k > kmax[m]) atomicMax(&kmax[m], k);
    }
}
// the same with a block's bounds in LDS (K <= 4096 PWMs): the pre-check above is two global loads per record behind the record's own
__global__ __launch_bounds__(256) void k_minmax_lds(const HitRec* hits, const uint16_t* scores, int64_t n, int K, uint32_t* kmin, uint32_t* kmax) {
    extern __shared__ uint32_t sb[];               // [K] minima, [K] maxima (order keys)
    for (int i = threadIdx.x; i < K; i += 256) sb[i] = hkey(0x7c00u), sb[K + i] = hkey(0xfc00u);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {       // four records requested together
        uint32_t m4[4];
        uint16_t s4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = min(i0 + u * stride, n - 1);          // (a clamped turn repeats the last record: bounds unchanged)
            m4[u] = hits[i].m - 1;
            s4[u] = scores[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t m = m4[u], k = hkey(s4[u]);
            if (m >= (uint32_t)K) continue;                         // a record of another bank: no counter in this block's LDS
            if (k < sb[m]) atomicMin(&sb[m], k);
            if (k > sb[K + m]) atomicMax(&sb[K + m], k);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) {
        if (sb[i] < kmin[i]) atomicMin(&kmin[i], sb[i]);
        if (sb[K + i] > kmax[i]) atomicMax(&kmax[i], sb[K + i]);
    }
}
__global__ void k_minmax_out(const uint32_t* kmin, const uint32_t* kmax, int K, uint16_t* mn, uint16_t* mx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        mn[i] = hunkey(kmin[i]);
        mx[i] = hunkey(kmax[i]);
    }
}

// ---- threshold sweep: counts[m][j] += #{records of m with score > thr[m][j]} (get_hits, :3-9) ----
__global__ void k_thr_hist(const HitRec* hits, const uint16_t* scores, int64_t n, const uint16_t* thr, int T,
                           unsigned long long* hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = hits[i].m - 1;
        const float s = h2f(scores[i]);
        const uint16_t* t = thr + (size_t)m * T;
        int lo = 0, hi = T;                       // number of thresholds strictly below the score (thr ascending)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (h2f(t[mid]) < s) lo = mid + 1; else hi = mid;
        }
        if (lo > 0) atomicAdd(&hist[(size_t)m * (T + 1) + lo], 1ull);
    }
}
// the same with the counters (and the thresholds) in LDS: one global atomic per record onto K (T + 1) addresses was what the pass cost (3.6 ms for
// 28.5 M records against 8 600 counters at configs[1]); a block's non-zero counters are added to the global ones once, at its end
__global__ __launch_bounds__(1024) void k_thr_hist_lds(const HitRec* hits, const uint16_t* scores, int64_t n, const uint16_t* thr, int K, int T,
                                                      unsigned long long* hist) {
    extern __shared__ uint32_t sh[];               // [K (T + 1)] counters, then [K T] thresholds (halves)
    const int nb = K * (T + 1);
    uint16_t* st = (uint16_t*)(sh + nb);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) sh[i] = 0;
    for (int i = threadIdx.x; i < K * T; i += blockDim.x) st[i] = thr[i];
    __syncthreads();
    // four records per thread and turn, requested together (one at a time the pass waited on ~110 dependent trips to memory per thread)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int top = 1;
    while (top * 2 <= T) top *= 2;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint32_t m4[4];
        uint16_t s4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = i0 + u * stride;
            m4[u] = i < n ? hits[i].m - 1 : 0u;
            s4[u] = i < n ? scores[i] : (uint16_t)0xfc00u;        // -Inf: below every threshold, counted nowhere
            if (m4[u] >= (uint32_t)K) m4[u] = 0u, s4[u] = (uint16_t)0xfc00u;    // a record of another bank: counted nowhere
        }
        // thresholds strictly below the score, by steps of falling powers of two: the same trips for every record, so the four searches
        // interleave (a while (lo < hi) per record ran them one after the other)
        int lo4[4] = {0, 0, 0, 0};
        for (int step = top; step; step >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int np = lo4[u] + step;
                const float tv = h2f(st[(size_t)m4[u] * T + min(np, T) - 1]);
                if (np <= T && tv < h2f(s4[u])) lo4[u] = np;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (lo4[u] > 0 && i0 + u * stride < n) atomicAdd(&sh[m4[u] * (T + 1) + lo4[u]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
__global__ void k_thr_suffix(const unsigned long long* hist, int K, int T, int64_t* counts) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= K) return;
    unsigned long long run = 0;
    for (int j = T - 1; j >= 0; j--) {            // records whose position is > j pass threshold j
        run += hist[(size_t)m * (T + 1) + j + 1];
        counts[(size_t)m * T + j] += (int64_t)run;
    }
}

// ---- threshold filter: stable compaction of records with score > thresh[m] (:116-125) ----
constexpr int FCH = 1024;
__global__ __launch_bounds__(256) void k_filt_count(const HitRec* hits, const uint16_t* scores, int64_t n, const uint16_t* thresh,
                                                    uint32_t* chunk_cnt) {
    __shared__ uint32_t red[4];
    const int64_t c = blockIdx.x;
    uint32_t k = 0;
    for (int j = 0; j < FCH / 256; j++) {
        const int64_t i = c * FCH + j * 256 + threadIdx.x;
        if (i < n) k += h2f(scores[i]) > h2f(thresh[hits[i].m - 1]);
    }
    for (int d = 32; d >= 1; d >>= 1) k += __shfl_xor(k, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[c] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void k_filt_scan(const uint32_t* chunk_cnt, int64_t nchunks, int64_t* chunk_base, int64_t* total) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (nchunks + 1023) / 1024;
    int64_t lo = tid * per, hi = lo + per;
    if (lo > nchunks) lo = nchunks;
    if (hi > nchunks) hi = nchunks;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += chunk_cnt[i];
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
        chunk_base[i] = (int64_t)run;
        run += chunk_cnt[i];
    }
    if (tid == 1023) *total = (int64_t)part[1023];
}
__global__ __launch_bounds__(256) void k_filt_write(const HitRec* hits, const uint16_t* scores, int64_t n, const uint16_t* thresh,
                                                    const int64_t* chunk_base, HitRec* oh, uint16_t* os) {
    __shared__ uint32_t wsum[4];
    const int64_t c = blockIdx.x;
    int64_t run = chunk_base[c];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = 0; j < FCH / 256; j++) {
        const int64_t i = c * FCH + j * 256 + threadIdx.x;
        const bool keep = i < n && h2f(scores[i]) > h2f(thresh[hits[i].m - 1]);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) wsum[wv] = __builtin_popcountll(m);
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (int q = 0; q < 4; q++) {
            if (q < wv) base += wsum[q];
            tot += wsum[q];
        }
        if (keep) {
            const int64_t at = run + base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            oh[at] = hits[i];
            os[at] = scores[i];
        }
        run += tot;
        __syncthreads();
    }
}

// ---- positions -> count matrices (posdicts2countmats, _h6:26-55; reverse strand = submat_comlement, _3:49-52) ----
// counts[m][ind][a] (the bytes of Julia's (4, maxlen, K) array) += one-hot window of every hit
__global__ __launch_bounds__(256) void k_count_mats(const HitRec* hits, int64_t n, const uint8_t* codes, int pitch, int64_t n0,
                                                    const int32_t* lens, int K, int maxlen, int comp, int use_lds,
                                                    unsigned int* counts) {
    extern __shared__ unsigned int lh[];
    const int bins = K * maxlen * 4;
    if (use_lds) {
        for (int i = threadIdx.x; i < bins; i += 256) lh[i] = 0;
        __syncthreads();
    }
    unsigned int* dst = use_lds ? lh : counts;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const HitRec h = hits[i];
        const int m = (int)h.m - 1, len = lens[m];
        const uint8_t* s = codes + ((int64_t)h.n - 1 - n0) * pitch + (h.l - 1);
        for (int ind = 0; ind < len; ind++) {
            const int b = s[ind];
            if (b > 3) continue;
            const int a = comp ? 3 - b : b, pos = comp ? len - 1 - ind : ind;
            atomicAdd(&dst[((size_t)m * maxlen + pos) * 4 + a], 1u);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < bins; i += 256)
            if (lh[i]) atomicAdd(&counts[i], lh[i]);
    }
}

// The same with a hit's window fetched as NW aligned dwords (requested together, funnel-shifted to the window's first base) instead of
// `len` dependent byte loads: windows of up to 4 (NW - 1) positions.  A dword index is clamped to the row (pitch / 4 - 1): what a clamped
// dword would have held lies past the window.  28 M records of a configs[1] strand: 0.84 ms with the byte loads, 0.24 now; by blocks
// (one LDS copy of the matrices each, flushed with one global atomic per non-zero counter): 2 048: 0.27, 1 024: 0.24, 512: 0.34 ms.
template <int NW>
__global__ __launch_bounds__(256) void k_count_mats_w(const HitRec* hits, int64_t n, const uint8_t* codes, int pitch, int64_t n0,
                                                      const int32_t* lens, int K, int maxlen, int comp, unsigned int* counts) {
    extern __shared__ unsigned int lh[];
    const int bins = K * maxlen * 4;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int dmax = pitch / 4 - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const HitRec h = hits[i];
        const int m = (int)h.m - 1, off = (int)h.l - 1;
        if ((unsigned)m >= (unsigned)K) continue;                   // a record of another bank: no matrix in this block's LDS
        const int len = lens[m];
        const uint32_t* rw = (const uint32_t*)(codes + ((int64_t)h.n - 1 - n0) * pitch);
        const int d0 = off >> 2;
        const uint32_t sh = (uint32_t)(off & 3) * 8u;
        uint32_t w[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) w[j] = rw[min(d0 + j, dmax)];
        unsigned int* row = lh + (size_t)m * maxlen * 4;
#pragma unroll
        for (int j = 0; j < NW - 1; j++) {
            const uint32_t v = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);     // positions 4j .. 4j + 3 of the window
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ind = 4 * j + u;
                const int b = (int)((v >> (8 * u)) & 0xffu);
                if (ind < len && b <= 3) {
                    const int a = comp ? 3 - b : b, pos = comp ? len - 1 - ind : ind;
                    atomicAdd(&row[pos * 4 + a], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
        if (lh[i]) atomicAdd(&counts[i], lh[i]);
}

static int need(motifs_ctx* c, const char* fn) {
    if (!c) {
        set_error("%s: null context", fn);
        return MOTIFS_ERR_INVALID;
    }
    return MOTIFS_OK;
}

extern "C" {

int motifs_hits_minmax_dev(motifs_ctx* c, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n, int K,
                           uint16_t* min_dev, uint16_t* max_dev) {
    int r = need(c, "motifs_hits_minmax_dev");
    if (r) return r;
    if (K < 1 || n < 0 || !min_dev || !max_dev || (n > 0 && (!hits_dev || !scores_dev))) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(c->small.reserve((size_t)K * 8 + 64));
    uint32_t* kmin = (uint32_t*)c->small.p;
    uint32_t* kmax = kmin + K;
    hipLaunchKernelGGL(k_minmax_init, dim3((K + 255) / 256), dim3(256), 0, c->stream, kmin, kmax, K);
    if (n >= 2048 && K <= 4096)
        hipLaunchKernelGGL(k_minmax_lds, dim3((unsigned)std::min<int64_t>((n + 256 * 16 - 1) / (256 * 16), 2048)), dim3(256), (size_t)K * 8, c->stream,
                           (const HitRec*)hits_dev, scores_dev, n, K, kmin, kmax);
    else if (n > 0)
        hipLaunchKernelGGL(k_minmax, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, c->stream,
                           (const HitRec*)hits_dev, scores_dev, n, kmin, kmax);
    hipLaunchKernelGGL(k_minmax_out, dim3((K + 255) / 256), dim3(256), 0, c->stream, kmin, kmax, K, min_dev, max_dev);
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

int motifs_hits_threshold_counts_dev(motifs_ctx* c, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n, int K,
                                     const uint16_t* thr_dev, int T, int64_t* counts_dev) {
    int r = need(c, "motifs_hits_threshold_counts_dev");
    if (r) return r;
    if (K < 1 || T < 1 || n < 0 || !thr_dev || !counts_dev || (n > 0 && (!hits_dev || !scores_dev))) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)K * (T + 1) * 8));
    unsigned long long* hist = (unsigned long long*)c->tilesum.p;
    MOTIFS_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)K * (T + 1) * 8, c->stream));
    const size_t lds = (size_t)K * (T + 1) * 4 + (size_t)K * T * 2;
    if (n >= 2048 && lds <= 64 * 1024) {          // (a block counts fewer than 2^32 records)
        // 1 024-thread blocks, as many per CU as their LDS lets in (two at configs[1]: 51 KB each), in ONE round: more waves per CU behind
        // fewer LDS copies to flush.  28.5 M records: 0.34 ms with 1 024 blocks of 256 threads, 0.22 with one round of them (768), 0.126 now.
        const int bt = 1024;
        const int64_t slots = 256 * std::min<int64_t>(2048 / bt, (160 * 1024) / (lds + 512));
        hipLaunchKernelGGL(k_thr_hist_lds, dim3((unsigned)std::min<int64_t>((n + bt * 16 - 1) / (bt * 16), slots)), dim3(bt), lds, c->stream,
                           (const HitRec*)hits_dev, scores_dev, n, thr_dev, K, T, hist);
    }
    else if (n > 0)
        hipLaunchKernelGGL(k_thr_hist, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, c->stream,
                           (const HitRec*)hits_dev, scores_dev, n, thr_dev, T, hist);
    hipLaunchKernelGGL(k_thr_suffix, dim3((K + 63) / 64), dim3(64), 0, c->stream, hist, K, T, counts_dev);
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

int motifs_hits_filter_dev(motifs_ctx* c, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n, int K,
                           const uint16_t* thresh_dev, motifs_hit* out_hits_dev, uint16_t* out_scores_dev, int64_t* n_out) {
    int r = need(c, "motifs_hits_filter_dev");
    if (r) return r;
    if (K < 1 || n < 0 || !thresh_dev || !n_out || (n > 0 && (!hits_dev || !scores_dev || !out_hits_dev || !out_scores_dev)))
        return MOTIFS_ERR_INVALID;
    *n_out = 0;
    if (n == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    const int64_t nch = (n + FCH - 1) / FCH;
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)nch * 4));
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)nch * 8 + 64));
    uint32_t* cc = (uint32_t*)c->tilesum.p;
    int64_t* cb = (int64_t*)c->off.p;
    int64_t* total = cb + nch;
    hipLaunchKernelGGL(k_filt_count, dim3((unsigned)nch), dim3(256), 0, c->stream, (const HitRec*)hits_dev, scores_dev, n, thresh_dev, cc);
    hipLaunchKernelGGL(k_filt_scan, dim3(1), dim3(1024), 0, c->stream, cc, nch, cb, total);
    hipLaunchKernelGGL(k_filt_write, dim3((unsigned)nch), dim3(256), 0, c->stream, (const HitRec*)hits_dev, scores_dev, n, thresh_dev, cb,
                       (HitRec*)out_hits_dev, out_scores_dev);
    int64_t* h_total = (int64_t*)c->pinned;
    MOTIFS_HIP_CHECK(hipMemcpyAsync(h_total, total, 8, hipMemcpyDeviceToHost, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *n_out = *h_total;
    return MOTIFS_OK;
}

int motifs_hits_count_matrices_dev(motifs_ctx* c, const motifs_hit* hits_dev, int64_t n, const uint8_t* codes_dev, int L, int64_t n0,
                                   const int64_t* lens, int K, int maxlen, int comp, uint32_t* counts_dev) {
    int r = need(c, "motifs_hits_count_matrices_dev");
    if (r) return r;
    if (K < 1 || maxlen < 1 || n < 0 || !lens || !counts_dev || (n > 0 && (!hits_dev || !codes_dev))) return MOTIFS_ERR_INVALID;
    if (n == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    std::vector<int32_t> l32(K);
    for (int k = 0; k < K; k++) l32[k] = (int32_t)lens[k];
    if (l32 != c->cm_lens_host) {                  // (posdicts2countmats is called per strand and per refinement round with the same motifs)
        c->cm_lens_host.clear();
        MOTIFS_HIP_CHECK(c->cm_lens.reserve((size_t)K * 4));
        MOTIFS_HIP_CHECK(hipMemcpyAsync(c->cm_lens.p, l32.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->cm_lens_host = l32;
    }
    const size_t bins = (size_t)K * maxlen * 4;
    const int use_lds = bins * 4 <= 48 * 1024;
    if (use_lds && maxlen <= 28 && ((uintptr_t)codes_dev & 3) == 0) {
        const int nblk = 1024;           // (2 048 blocks: 0.27 ms, 1 024: 0.24, 512: 0.34 per 28 M records)
        const int cbt = 256;             // (512- and 1 024-thread blocks, 256-768 of them: 0.21-0.26 ms - the pass is bound by its LDS atomics)
        const dim3 grid((unsigned)std::min<int64_t>((n + cbt - 1) / cbt, std::max(nblk, 1)));
        const int pitch = motifs_codes_pitch(L);
        if (maxlen <= 12)
            hipLaunchKernelGGL(k_count_mats_w<4>, grid, dim3(cbt), bins * 4, c->stream, (const HitRec*)hits_dev, n, codes_dev, pitch, n0,
                               (const int32_t*)c->cm_lens.p, K, maxlen, comp, counts_dev);
        else if (maxlen <= 20)
            hipLaunchKernelGGL(k_count_mats_w<6>, grid, dim3(cbt), bins * 4, c->stream, (const HitRec*)hits_dev, n, codes_dev, pitch, n0,
                               (const int32_t*)c->cm_lens.p, K, maxlen, comp, counts_dev);
        else
            hipLaunchKernelGGL(k_count_mats_w<8>, grid, dim3(cbt), bins * 4, c->stream, (const HitRec*)hits_dev, n, codes_dev, pitch, n0,
                               (const int32_t*)c->cm_lens.p, K, maxlen, comp, counts_dev);
        MOTIFS_HIP_CHECK(hipGetLastError());
        return MOTIFS_OK;
    }
    hipLaunchKernelGGL(k_count_mats, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), use_lds ? bins * 4 : 0, c->stream,
                       (const HitRec*)hits_dev, n, codes_dev, motifs_codes_pitch(L), n0, (const int32_t*)c->cm_lens.p, K, maxlen, comp,
                       use_lds, counts_dev);
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

// `reading` + `read_fasta` (loadfasta/helpers.jl:83-108): split at '>', drop the header line, join the rest, drop
// reads containing N/n, keep the first max_entries, then only reads as long as the first; upper-case; A,C,G,T -> 0..3.
// codes_out: n_reads rows of L bytes.  Query mode: codes_out == NULL returns the sizes.
int motifs_fasta_read(const char* path, int64_t max_entries, uint8_t* codes_out, int64_t cap_bytes, int64_t* n_reads, int32_t* L) {
    if (!path || !n_reads || !L || max_entries < 0) {
        set_error("motifs_fasta_read: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("motifs_fasta_read: cannot open %s", path);
        return MOTIFS_ERR_INVALID;
    }
    std::vector<char> text;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long sz = ftell(f);
        rewind(f);
        if (sz > 0) text.resize((size_t)sz);
        const size_t got = text.empty() ? 0 : fread(text.data(), 1, text.size(), f);
        text.resize(got);
    }
    fclose(f);
    const char* t = text.data();
    const size_t tn = text.size();
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    // fn(begin, end) over [0, n) in contiguous pieces, one host thread each (the file is tens of MB: worth the threads)
    auto parallel = [&](size_t n, size_t grain, const std::function<void(size_t, size_t, unsigned)>& fn) -> unsigned {
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(hw, n / std::max<size_t>(grain, 1)));
        std::vector<std::thread> th;
        for (unsigned i = 1; i < T; i++) th.emplace_back(fn, n * i / T, n * (i + 1) / T, i);
        fn(0, n / T, 0u);
        for (auto& x : th) x.join();
        return T;
    };
    // 1. records lie between '>' characters (`split(reads, '>')`, helpers.jl:85)
    std::vector<std::vector<size_t>> marks(hw);
    parallel(tn, 1 << 20, [&](size_t lo, size_t hi, unsigned ti) {
        for (const char* p = t + lo; p < t + hi;) {
            const char* q = (const char*)memchr(p, '>', (size_t)(t + hi - p));
            if (!q) break;
            marks[ti].push_back((size_t)(q - t));
            p = q + 1;
        }
    });
    struct Rec {
        size_t b, e;        // the sequence text: after the header line, up to the next '>'
        size_t len;         // its characters without the line breaks (`join(splits[2:end])`, :88)
        bool has_n;
    };
    std::vector<Rec> recs;
    {
        size_t pos = 0;
        auto add = [&](size_t lo, size_t hi) {
            if (hi > lo) recs.push_back(Rec{lo, hi, 0, false});      // !isempty(i)
        };
        for (const auto& v : marks)
            for (size_t m : v) {
                add(pos, m);
                pos = m + 1;
            }
        add(pos, tn);
    }
    // 2. per record: drop the header line, measure the sequence, look for N / n (:92)
    parallel(recs.size(), 4096, [&](size_t lo, size_t hi, unsigned) {
        for (size_t i = lo; i < hi; i++) {
            Rec& r = recs[i];
            const char* nl = (const char*)memchr(t + r.b, '\n', r.e - r.b);
            if (!nl) {                                               // a header without a sequence: an empty read
                r.b = r.e;
                continue;
            }
            r.b = (size_t)(nl + 1 - t);
            size_t breaks = 0;                                       // memchr hops: the lines are long, libc's scan is vectorised
            for (const char* p = t + r.b; p < t + r.e;) {
                const char* q = (const char*)memchr(p, '\n', (size_t)(t + r.e - p));
                if (!q) break;
                breaks++;
                p = q + 1;
            }
            r.len = r.e - r.b - breaks;
            r.has_n = memchr(t + r.b, 'N', r.e - r.b) != nullptr || memchr(t + r.b, 'n', r.e - r.b) != nullptr;
        }
    });
    // 3. reads with N dropped, the first max_entries kept (:95), then only reads as long as the first (:98)
    std::vector<const Rec*> keep;
    for (const Rec& r : recs) {
        if (r.has_n) continue;
        if ((int64_t)keep.size() >= max_entries) break;
        keep.push_back(&r);
    }
    if (keep.empty()) {
        set_error("There aren't DNA strings found in the input");          // helpers.jl:131
        return MOTIFS_ERR_INVALID;
    }
    const size_t len0 = keep[0]->len;
    {
        size_t w = 0;
        for (const Rec* r : keep)
            if (r->len == len0) keep[w++] = r;
        keep.resize(w);
    }
    *n_reads = (int64_t)keep.size();
    *L = (int32_t)len0;
    if (!codes_out) return MOTIFS_OK;
    if ((int64_t)(keep.size() * len0) > cap_bytes) {
        set_error("motifs_fasta_read: buffer too small (%zu bytes needed)", keep.size() * len0);
        return MOTIFS_ERR_BUFFER_TOO_SMALL;
    }
    // 4. base codes (dna2dummy's Dict, :110-117): A C G T in either case; anything else is the reference's KeyError.
    // Line by line, branch-free so that the compiler vectorises it: with x = (ch >> 1) & 3, A C G T give 0 1 3 2,
    // and x ^ (x >> 1) is 0 1 2 3.
    std::vector<size_t> bad_read(hw, SIZE_MAX), bad_pos(hw, 0);
    std::vector<char> bad_ch(hw, 0);
    parallel(keep.size(), 2048, [&](size_t lo, size_t hi, unsigned ti) {
        for (size_t i = lo; i < hi; i++) {
            uint8_t* __restrict out = codes_out + i * len0;
            size_t k = 0;
            for (const char* p = t + keep[i]->b; p < t + keep[i]->e;) {
                const char* q = (const char*)memchr(p, '\n', (size_t)(t + keep[i]->e - p));
                const size_t n = (size_t)((q ? q : t + keep[i]->e) - p);
                const unsigned char* __restrict in = (const unsigned char*)p;
                unsigned bad = 0;
                for (size_t j = 0; j < n; j++) {
                    const unsigned ch = in[j], up = ch & 0xdfu, x = (ch >> 1) & 3u;
                    bad |= (unsigned)((up != 'A') & (up != 'C') & (up != 'G') & (up != 'T'));
                    out[k + j] = (uint8_t)(x ^ (x >> 1));
                }
                if (bad) {
                    for (size_t j = 0; j < n; j++) {
                        const unsigned up = in[j] & 0xdfu;
                        if (up != 'A' && up != 'C' && up != 'G' && up != 'T') {
                            bad_read[ti] = i;
                            bad_pos[ti] = k + j;
                            bad_ch[ti] = (char)in[j];
                            return;
                        }
                    }
                }
                k += n;
                p += n + 1;
            }
        }
    });
    for (unsigned ti = 0; ti < hw; ti++)                               // pieces are in read order: the first failing piece holds the first failing read
        if (bad_read[ti] != SIZE_MAX) {
            set_error("motifs_fasta_read: read %zu has '%c' at %zu (the reference's dna2dummy raises a KeyError)", bad_read[ti] + 1, bad_ch[ti],
                      bad_pos[ti] + 1);
            return MOTIFS_ERR_INVALID;
        }
    return MOTIFS_OK;
}

}  // extern "C"
