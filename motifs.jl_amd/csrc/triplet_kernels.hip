// triplet_kernels.hip — consumers of the code records: the magnitude quantile filter and the
// triplet enumeration of src/inference/_2_enumerate.jl, on device-resident record arrays.
//
//   filter_code_components_using_quantile! (:10-13)  -> motifs_codes_mag_histogram_dev (the order statistics
//        of the binary16 magnitudes come from a 65536-bin histogram; the caller interpolates as Statistics.quantile
//        does) + motifs_codes_filter_dev (ordered compaction of `mag > threshold`)
//   enumerate_triplets (:50-65) + insert_H! (:37-46) -> motifs_triplets_offsets_dev / _enumerate_dev: every
//        (i < j < k) of every scanning range, in the reference's insertion order, as a packed key + a value;
//        motifs_triplets_group_dev: the Dictionary those insertions build (unique keys in first-insertion order,
//        their counts, and the values of each key in insertion order)
//
// get_scanning_range_of_filtered_code_components (:25-35) is a sequential scan over the `seq` column with a
// data-dependent counter; it stays on the host (motifs.jl_amd/post.py, 4 bytes per record).
// The sorts, the run-length encoding and the scans of the grouping step are rocPRIM's; the rest is hand-written.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include <algorithm>

#include "../../include/motifs_hip.h"
#include "api_common.h"

namespace motifs {

static_assert(sizeof(motifs_code_rec) == 12, "motifs_code_rec layout");
static_assert(sizeof(motifs_triplet_val) == 8, "motifs_triplet_val layout");

constexpr int CCH = 2048;        // records per compaction chunk
constexpr int TRIP_MAX = 256;    // records per scanning range the enumeration keeps in LDS

__global__ void k_mag_hist(const motifs_code_rec* recs, int64_t n, uint32_t* hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&hist[recs[i].mag], 1u);
}

static __device__ __forceinline__ bool mag_above(uint16_t mag, double thresh) {
    return (double)__half2float(__ushort_as_half(mag)) > thresh;     // Float16 > Float64 promotes exactly
}
__global__ __launch_bounds__(256) void k_cfilt_count(const motifs_code_rec* recs, int64_t n, double thresh, uint32_t* chunk_cnt) {
    const int64_t lo = (int64_t)blockIdx.x * CCH;
    uint32_t c = 0;
    for (int i = threadIdx.x; i < CCH; i += 256)
        if (lo + i < n && mag_above(recs[lo + i].mag, thresh)) c++;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    __shared__ uint32_t red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one block: exclusive scan of up to a few million 32/64-bit counts (64-bit sums), grand total behind the last entry
template <typename T>
__global__ __launch_bounds__(1024) void k_excl_scan(const T* cnt, int64_t n, int64_t* base, int64_t* total) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    int64_t lo = tid * per, hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += (unsigned long long)cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (int64_t i = lo; i < hi; i++) {
        const unsigned long long v = (unsigned long long)cnt[i];
        base[i] = (int64_t)run;
        run += v;
    }
    if (tid == 1023) *total = (int64_t)part[1023];
}
__global__ __launch_bounds__(256) void k_cfilt_write(const motifs_code_rec* recs, int64_t n, double thresh, const int64_t* chunk_base,
                                                     motifs_code_rec* out) {
    const int64_t lo = (int64_t)blockIdx.x * CCH;
    __shared__ uint32_t wsum[4];
    int64_t at = chunk_base[blockIdx.x];
    for (int i0 = 0; i0 < CCH; i0 += 256) {                          // order inside the chunk = record order
        const int64_t i = lo + i0 + threadIdx.x;
        motifs_code_rec r{};
        const bool keep = i < n && (r = recs[i], mag_above(r.mag, thresh));
        const unsigned long long m = __ballot(keep);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wsum[wv] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (int q = 0; q < 4; q++) {
            if (q < wv) before += wsum[q];
            tot += wsum[q];
        }
        if (keep) out[at + before + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = r;
        at += tot;
        __syncthreads();
    }
}

__global__ void k_trip_counts(const uint32_t* rlen, int64_t nranges, int64_t* cnt) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nranges) return;
    const int64_t n = rlen[r];
    cnt[r] = n >= 3 ? n * (n - 1) * (n - 2) / 6 : 0;
}

// One wave per scanning range: the range's records sorted by position (stable, like sort(by = x -> x[1]), :57),
// then every i < j < k in lexicographic order (:59-63) with the key and value of insert_H! (:37-46).
__global__ __launch_bounds__(64) void k_trip_enum(const motifs_code_rec* recs, const uint32_t* rstart, const uint32_t* rlen, int64_t nranges,
                                                  int h, const int64_t* offsets, unsigned long long* keys, motifs_triplet_val* vals,
                                                  int64_t cap, int* too_long) {
    __shared__ uint16_t pos0[TRIP_MAX], fil0[TRIP_MAX], sp[TRIP_MAX], sf[TRIP_MAX];
    const int lane = threadIdx.x;
    for (int64_t r = blockIdx.x; r < nranges; r += gridDim.x) {
        const int n = (int)rlen[r];
        if (n < 3) continue;
        if (n > TRIP_MAX) {
            if (lane == 0) atomicExch(too_long, 1);
            continue;
        }
        const motifs_code_rec* rr = recs + rstart[r];
        for (int i = lane; i < n; i += 64) {
            pos0[i] = rr[i].position;
            fil0[i] = rr[i].fil;
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) {                          // stable rank by position
            const uint16_t p = pos0[i];
            int rank = 0;
            for (int j = 0; j < n; j++) rank += (pos0[j] < p) || (pos0[j] == p && j < i);
            sp[rank] = p;
            sf[rank] = fil0[i];
        }
        __syncthreads();
        int64_t at = offsets[r];
        for (int i = 0; i < n - 2; i++) {                             // wave-uniform
            const int m = n - i - 1;                                  // elements after i
            const int npairs = m * (m - 1) / 2;
            const unsigned long long f1 = sf[i] & 0xffu;
            const uint16_t p1 = sp[i];
            for (int p = lane; p < npairs; p += 64) {
                // unrank the p-th pair (a < b) of 0..m-1 in lexicographic order: p = a(2m - a - 1)/2 + (b - a - 1)
                int a = (int)(((double)(2 * m - 1) - sqrt((double)(2 * m - 1) * (2 * m - 1) - 8.0 * p)) * 0.5);
                while (a > 0 && a * (2 * m - a - 1) / 2 > p) a--;
                while ((a + 1) * (2 * m - a - 2) / 2 <= p) a++;
                const int b = p - a * (2 * m - a - 1) / 2 + a + 1;
                const int j = i + 1 + a, k = i + 1 + b;
                const unsigned long long d12 = (uint16_t)(sp[j] - p1), d13 = (uint16_t)(sp[k] - p1);
                const int64_t o = at + p;
                if (o < cap) {
                    keys[o] = (f1 << 48) | ((unsigned long long)(sf[j] & 0xffu) << 40) | ((unsigned long long)(sf[k] & 0xffu) << 32) |
                              (d12 << 16) | d13;
                    vals[o] = motifs_triplet_val{(uint32_t)(r + 1), p1, 0};
                }
            }
            at += npairs;
        }
        (void)h;   // len = d13 + h is implied by the key
        __syncthreads();
    }
}

__global__ void k_iota(uint32_t* x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = (uint32_t)i;
}
__global__ void k_gather_first(const uint32_t* idx_sorted, const int64_t* off_sorted, int64_t U, uint32_t* first) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x)
        first[u] = idx_sorted[off_sorted[u]];
}
__global__ void k_reorder_groups(const unsigned long long* uk_sorted, const uint32_t* cnt_sorted, const uint32_t* first_sorted_by_first,
                                 const uint32_t* order, int64_t U, unsigned long long* uk, int64_t* first, int64_t* counts) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < U; t += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = order[t];
        uk[t] = uk_sorted[u];
        counts[t] = cnt_sorted[u];
        first[t] = first_sorted_by_first[t];
    }
}
// values of group t (insertion order): the sorted index run of its key
__global__ __launch_bounds__(64) void k_fill_perm(const uint32_t* idx_sorted, const int64_t* off_sorted, const uint32_t* order, const int64_t* counts,
                                                  const int64_t* group_off, int64_t U, int64_t* perm) {
    for (int64_t t = blockIdx.x; t < U; t += gridDim.x) {
        const int64_t src = off_sorted[order[t]], dst = group_off[t], c = counts[t];
        for (int64_t j = threadIdx.x; j < c; j += 64) perm[dst + j] = idx_sorted[src + j];
    }
}

}  // namespace motifs

using namespace motifs;

static int need_ctx(motifs_ctx* c, const char* fn) {
    if (!c) {
        set_error("%s: null context", fn);
        return MOTIFS_ERR_INVALID;
    }
    return MOTIFS_OK;
}
static unsigned grid_for(int64_t n, int per = 256, int cap = 4096) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + per - 1) / per, cap));
}

extern "C" {

int motifs_codes_mag_histogram_dev(motifs_ctx* c, const motifs_code_rec* recs_dev, int64_t n, uint32_t* hist_dev) {
    int r = need_ctx(c, "motifs_codes_mag_histogram_dev");
    if (r) return r;
    if (n < 0 || !hist_dev || (n > 0 && !recs_dev)) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipMemsetAsync(hist_dev, 0, 65536 * 4, c->stream));
    if (n > 0) hipLaunchKernelGGL(k_mag_hist, dim3(grid_for(n)), dim3(256), 0, c->stream, recs_dev, n, hist_dev);
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

int motifs_codes_filter_dev(motifs_ctx* c, const motifs_code_rec* recs_dev, int64_t n, double thresh, motifs_code_rec* out_dev,
                            int64_t* n_out) {
    int r = need_ctx(c, "motifs_codes_filter_dev");
    if (r) return r;
    if (n < 0 || !n_out || (n > 0 && (!recs_dev || !out_dev))) return MOTIFS_ERR_INVALID;
    *n_out = 0;
    if (n == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    const int64_t nch = (n + CCH - 1) / CCH;
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)nch * 4));
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)nch * 8 + 64));
    uint32_t* cc = (uint32_t*)c->tilesum.p;
    int64_t* cb = (int64_t*)c->off.p;
    int64_t* total = cb + nch;
    hipLaunchKernelGGL(k_cfilt_count, dim3((unsigned)nch), dim3(256), 0, c->stream, recs_dev, n, thresh, cc);
    hipLaunchKernelGGL(k_excl_scan<uint32_t>, dim3(1), dim3(1024), 0, c->stream, cc, nch, cb, total);
    hipLaunchKernelGGL(k_cfilt_write, dim3((unsigned)nch), dim3(256), 0, c->stream, recs_dev, n, thresh, cb, out_dev);
    int64_t* h_total = (int64_t*)c->pinned;
    MOTIFS_HIP_CHECK(hipMemcpyAsync(h_total, total, 8, hipMemcpyDeviceToHost, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *n_out = *h_total;
    return MOTIFS_OK;
}

int motifs_triplets_offsets_dev(motifs_ctx* c, const uint32_t* range_len_dev, int64_t nranges, int64_t* offsets_dev, int64_t* total) {
    int r = need_ctx(c, "motifs_triplets_offsets_dev");
    if (r) return r;
    if (nranges < 0 || !total || (nranges > 0 && (!range_len_dev || !offsets_dev))) return MOTIFS_ERR_INVALID;
    *total = 0;
    if (nranges == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)nranges * 8 + 64));
    int64_t* cnt = (int64_t*)c->off.p;
    int64_t* tot = cnt + nranges;
    hipLaunchKernelGGL(k_trip_counts, dim3((unsigned)((nranges + 255) / 256)), dim3(256), 0, c->stream, range_len_dev, nranges, cnt);
    hipLaunchKernelGGL(k_excl_scan<int64_t>, dim3(1), dim3(1024), 0, c->stream, cnt, nranges, offsets_dev, tot);
    int64_t* h_total = (int64_t*)c->pinned;
    MOTIFS_HIP_CHECK(hipMemcpyAsync(h_total, tot, 8, hipMemcpyDeviceToHost, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *total = *h_total;
    return MOTIFS_OK;
}

int motifs_triplets_enumerate_dev(motifs_ctx* c, const motifs_code_rec* recs_dev, const uint32_t* range_start_dev,
                                  const uint32_t* range_len_dev, int64_t nranges, int h, const int64_t* offsets_dev, uint64_t* keys_dev,
                                  motifs_triplet_val* vals_dev, int64_t cap) {
    int r = need_ctx(c, "motifs_triplets_enumerate_dev");
    if (r) return r;
    if (nranges < 0 || cap < 0 || h < 0 ||
        (nranges > 0 && (!recs_dev || !range_start_dev || !range_len_dev || !offsets_dev || (cap > 0 && (!keys_dev || !vals_dev)))))
        return MOTIFS_ERR_INVALID;
    if (nranges == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(c->small.reserve(64));
    int* flag = (int*)c->small.p;
    MOTIFS_HIP_CHECK(hipMemsetAsync(flag, 0, 4, c->stream));
    hipLaunchKernelGGL(k_trip_enum, dim3(grid_for(nranges, 1, 8192)), dim3(64), 0, c->stream, recs_dev, range_start_dev, range_len_dev, nranges, h,
                       offsets_dev, (unsigned long long*)keys_dev, vals_dev, cap, flag);
    int* h_flag = (int*)c->pinned;
    MOTIFS_HIP_CHECK(hipMemcpyAsync(h_flag, flag, 4, hipMemcpyDeviceToHost, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (*h_flag) {
        set_error("motifs_triplets_enumerate_dev: a scanning range holds more than %d records", TRIP_MAX);
        return MOTIFS_ERR_UNSUPPORTED;
    }
    return MOTIFS_OK;
}

int motifs_triplets_group_dev(motifs_ctx* c, const uint64_t* keys_dev, int64_t n, uint64_t* uniq_keys_dev, int64_t* first_dev,
                              int64_t* counts_dev, int64_t* group_off_dev, int64_t* perm_dev, int64_t* n_unique) {
    int r = need_ctx(c, "motifs_triplets_group_dev");
    if (r) return r;
    if (n < 0 || !n_unique || (n > 0 && (!keys_dev || !uniq_keys_dev || !first_dev || !counts_dev || !group_off_dev || !perm_dev)))
        return MOTIFS_ERR_INVALID;
    *n_unique = 0;
    if (n == 0) return MOTIFS_OK;
    if (n >= ((int64_t)1 << 32)) {
        set_error("motifs_triplets_group_dev: %lld triplets exceed the 32-bit index", (long long)n);
        return MOTIFS_ERR_UNSUPPORTED;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    typedef unsigned long long u64;
    // workspace: sorted keys, two index arrays, unique keys / counts / offsets / firsts / order (each at most n entries)
    const size_t nn = (size_t)n;
    const size_t seg = ((nn + 3) & ~(size_t)3);               // entries per array, rounded so that every array starts 16-byte aligned
    MOTIFS_HIP_CHECK(c->staging.reserve(seg * (8 + 4 + 4 + 8 + 4 + 8 + 4 + 4 + 4) + 256));
    char* w = (char*)c->staging.p;
    u64* ks = (u64*)w;                 w += seg * 8;
    uint32_t* idx = (uint32_t*)w;      w += seg * 4;
    uint32_t* idxs = (uint32_t*)w;     w += seg * 4;
    u64* uks = (u64*)w;                w += seg * 8;
    uint32_t* cnts = (uint32_t*)w;     w += seg * 4;
    int64_t* offs = (int64_t*)w;       w += seg * 8;
    uint32_t* firsts = (uint32_t*)w;   w += seg * 4;
    uint32_t* firsts2 = (uint32_t*)w;  w += seg * 4;
    uint32_t* order = (uint32_t*)w;    w += seg * 4;
    uint32_t* runs = (uint32_t*)w;
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(256), 0, st, idx, n);
    size_t tb = 0, tb2 = 0;
    MOTIFS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, (const u64*)keys_dev, ks, idx, idxs, nn, 0, 56, st));
    MOTIFS_HIP_CHECK(rocprim::run_length_encode(nullptr, tb2, ks, (unsigned int)nn, uks, cnts, runs, st));
    tb = std::max(tb, tb2);
    MOTIFS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb2, firsts, firsts2, idx, order, nn, 0, 32, st));
    tb = std::max(tb, tb2);
    MOTIFS_HIP_CHECK(c->data_tmp.reserve(tb + 256));
    void* tmp = c->data_tmp.p;
    size_t t = tb;
    MOTIFS_HIP_CHECK(rocprim::radix_sort_pairs(tmp, t, (const u64*)keys_dev, ks, idx, idxs, nn, 0, 56, st));   // stable: ties keep insertion order
    t = tb;
    MOTIFS_HIP_CHECK(rocprim::run_length_encode(tmp, t, ks, (unsigned int)nn, uks, cnts, runs, st));
    uint32_t* h_runs = (uint32_t*)c->pinned;
    MOTIFS_HIP_CHECK(hipMemcpyAsync(h_runs, runs, 4, hipMemcpyDeviceToHost, st));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(st));
    const int64_t U = *h_runs;
    int64_t* tot = offs + U;   // offs has room for n + ... entries; U <= n, the total lands behind the last offset when U < n
    if (U == n) {
        MOTIFS_HIP_CHECK(c->small.reserve(64));
        tot = (int64_t*)c->small.p;
    }
    hipLaunchKernelGGL(k_excl_scan<uint32_t>, dim3(1), dim3(1024), 0, st, cnts, U, offs, tot);
    hipLaunchKernelGGL(k_gather_first, dim3(grid_for(U)), dim3(256), 0, st, idxs, offs, U, firsts);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(U)), dim3(256), 0, st, idx, U);
    t = tb;
    MOTIFS_HIP_CHECK(rocprim::radix_sort_pairs(tmp, t, firsts, firsts2, idx, order, (size_t)U, 0, 32, st));   // Dictionary order = first insertion
    hipLaunchKernelGGL(k_reorder_groups, dim3(grid_for(U)), dim3(256), 0, st, uks, cnts, firsts2, order, U, (u64*)uniq_keys_dev, first_dev,
                       counts_dev);
    MOTIFS_HIP_CHECK(c->small.reserve(64));
    hipLaunchKernelGGL(k_excl_scan<int64_t>, dim3(1), dim3(1024), 0, st, counts_dev, U, group_off_dev, (int64_t*)c->small.p + 1);
    hipLaunchKernelGGL(k_fill_perm, dim3(grid_for(U, 1, 8192)), dim3(64), 0, st, idxs, offs, order, counts_dev, group_off_dev, U, perm_dev);
    MOTIFS_HIP_CHECK(hipGetLastError());
    MOTIFS_HIP_CHECK(hipStreamSynchronize(st));
    *n_unique = U;
    return MOTIFS_OK;
}

}  // extern "C"
