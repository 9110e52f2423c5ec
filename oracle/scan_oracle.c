/*
 * scan_oracle.c — CPU restatement of MOTIFs.jl's PWM log-odds scan.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * (test/runtests.jl:4-6 is an empty testset) and Julia is not installed in
 * the build container, so this restatement could not be checked against the
 * reference's own outputs.  It is pinned instead by hand-computed known-answer
 * cases and identities (tests/test_oracle_scan.py).
 *
 * What it restates (paths relative to the reference checkout):
 *   src/inference/_h3_1_alignment.jl:18-36   greedy_search!   -> oracle_greedy_search
 *   src/inference/_h3_1_alignment.jl:57-87   get_pos_scores_arr -> oracle_get_pos_scores_arr
 *   src/inference/_0_const.jl:1              float_type_retrieval = Float16
 *
 * Arithmetic: IEEE binary16 with round-to-nearest-even after EVERY multiply
 * and EVERY add, accumulated in the output array exactly as the reference
 * kernel does (`pos_scores[k,n,l] += pwms[k,a,ind]*data[(i-1)*4+a,n]`, :29).
 * Products and sums of two binary16 values are formed in binary32 (both are
 * exact or differ from the exact result by less than a quarter binary16 ulp in
 * binary32, so the single rounding to binary16 that follows is the correctly
 * rounded binary16 result) and rounded back with f32_to_f16 below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* binary32 -> binary16, round to nearest even, IEEE overflow to inf. */
static inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u | ((ax >> 13) & 0x3ffu) : 0));
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) { /* below smallest normal half: subnormal or zero */
        if (ax < 0x33000000u) return sign; /* < 2^-25 rounds to zero (2^-25 itself ties to even = 0) */
        uint32_t e = ax >> 23;             /* biased exponent, 102..112 */
        uint32_t man = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126 - e;          /* 14..24 */
        uint32_t q = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t e = (ax >> 23) - 112; /* 1..30 */
    uint32_t man = ax & 0x7fffffu;
    uint32_t q = (e << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++; /* may carry into exponent: correct */
    return (uint16_t)(sign | q);
}

static inline uint16_t h_mul(uint16_t a, uint16_t b) { return f32_to_f16(f16_to_f32(a) * f16_to_f32(b)); }
static inline uint16_t h_add(uint16_t a, uint16_t b) { return f32_to_f16(f16_to_f32(a) + f16_to_f32(b)); }

uint16_t oracle_f32_to_f16(float f) { return f32_to_f16(f); }
float oracle_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint16_t oracle_h_add(uint16_t a, uint16_t b) { return h_add(a, b); }

/*
 * greedy_search!  (_h3_1_alignment.jl:18-36), one CPU loop iteration per GPU thread.
 *   pwms       (K,4,maxlen) col-major fp16            — `pwms[k,a,ind]`
 *   data       (L4, N) col-major fp16 one-hot         — `data_dat_gpu[(i-1)*4+a, n]`
 *   lens       (K) Int64
 *   pos_scores (K, N, L4) col-major fp16, PRE-ZEROED by the caller (:75)
 * Indices k,n,l,ind,a below are the reference's 1-based ones.
 */
void oracle_greedy_search(const uint16_t* pwms, const uint16_t* data, const int64_t* lens, int K,
                          int maxlen, int64_t N, int L4, uint16_t* pos_scores) {
    (void)maxlen;
    const int L_div_4 = L4 / 4;
#pragma omp parallel for schedule(static)
    for (int64_t n = 1; n <= N; n++) {
        for (int l = 1; l <= L4; l++) { /* grid covers the whole third dim (:76, _0_const.jl:41) */
            for (int k = 1; k <= K; k++) {
                if (!(l <= L_div_4 - lens[k - 1] + 1)) continue; /* :25 */
                uint16_t* out = &pos_scores[(k - 1) + (int64_t)K * ((n - 1) + N * (int64_t)(l - 1))];
                int ind = 0;
                for (int i = l; i <= l + lens[k - 1] - 1; i++) { /* :26 enumerate(l:l+lens[k]-1) */
                    ind++;
                    for (int a = 1; a <= 4; a++) { /* :28 */
                        uint16_t w = pwms[(k - 1) + K * ((a - 1) + 4 * (ind - 1))];
                        uint16_t d = data[((i - 1) * 4 + a - 1) + (int64_t)L4 * (n - 1)];
                        *out = h_add(*out, h_mul(w, d)); /* :29 */
                    }
                }
                /* :33  pos_scores > 0f0 ? pos_scores : 0f0  (NaN -> 0) */
                *out = (f16_to_f32(*out) > 0.0f) ? *out : (uint16_t)0;
            }
        }
    }
}

typedef struct {
    uint32_t m, n, l;
} oracle_hit;

/*
 * get_pos_scores_arr (_h3_1_alignment.jl:57-87) for one strand.
 *   pwms_in  (K,4,maxlen) col-major fp16 = the forward bank built at :66-69 with rc=false
 *            (zero-padded `ms.pwms[i]`); with rc != 0 this routine applies
 *            `reverse(ms.pwms[i])` (both dims) per motif before padding, as :68 does.
 *   data_f32 (L4, N) col-major Float32 one-hot = data.data_matrix[:,1,:]
 * Output records in the reference's order: batches of 5000 sequences (:71), inside a batch
 * `findall(pos_scores_arr .> 0)` (:82) walks the (K, nb, L4) array column-major.
 * Returns the number of hits; writes at most `cap` of them.
 */
int64_t oracle_get_pos_scores_arr(const uint16_t* pwms_in, const int64_t* lens, int K, int maxlen,
                                  const float* data_f32, int64_t N, int L4, int rc, int batch_size,
                                  oracle_hit* found, uint16_t* score_record, int64_t cap) {
    uint16_t* pwms = (uint16_t*)calloc((size_t)K * 4 * maxlen, 2);
    for (int k = 0; k < K; k++) {
        int len = (int)lens[k];
        for (int ind = 0; ind < len; ind++)
            for (int a = 0; a < 4; a++) {
                /* reverse(pwm): element (a, ind) <- (4-1-a, len-1-ind) */
                int sa = rc ? 3 - a : a, si = rc ? len - 1 - ind : ind;
                pwms[k + K * (a + 4 * ind)] = pwms_in[k + K * (sa + 4 * si)];
            }
    }
    int64_t nfound = 0;
    for (int64_t n0 = 1; n0 <= N; n0 += batch_size) { /* :71 */
        int64_t nend = n0 + batch_size - 1 < N ? n0 + batch_size - 1 : N;
        int64_t nb = nend - n0 + 1;
        /* :74 float_type_retrieval.(data_matrix[:,1,n:nend]) */
        uint16_t* data16 = (uint16_t*)malloc((size_t)L4 * nb * 2);
        for (int64_t i = 0; i < (int64_t)L4 * nb; i++)
            data16[i] = f32_to_f16(data_f32[(int64_t)L4 * (n0 - 1) + i]);
        uint16_t* pos_scores = (uint16_t*)calloc((size_t)K * nb * L4, 2); /* :75 */
        oracle_greedy_search(pwms, data16, lens, K, maxlen, nb, L4, pos_scores);
        /* :82-84 findall(.> 0) in column-major order; n -> n + n0 - 1 */
        for (int l = 1; l <= L4; l++)
            for (int64_t n = 1; n <= nb; n++)
                for (int k = 1; k <= K; k++) {
                    uint16_t s = pos_scores[(k - 1) + (int64_t)K * ((n - 1) + nb * (int64_t)(l - 1))];
                    if (f16_to_f32(s) > 0.0f) {
                        if (nfound < cap) {
                            found[nfound].m = (uint32_t)k;
                            found[nfound].n = (uint32_t)(n + n0 - 1);
                            found[nfound].l = (uint32_t)l;
                            score_record[nfound] = s;
                        }
                        nfound++;
                    }
                }
        free(pos_scores);
        free(data16);
    }
    free(pwms);
    return nfound;
}

/*
 * Same function as oracle_greedy_search computed the cheap way (gather of the
 * one selected weight per position, no dense L4 third dimension): used as the
 * "optimised CPU" leg of bench.py's cpu_baseline and to cross-check the literal
 * loop.  codes: (N rows of L bytes) 0..3, 4 = all-zero column.
 * scores: (K, N, Lout) col-major, Lout = L - minlen + 1.
 */
void oracle_scan_gather(const uint16_t* pwms, const int64_t* lens, int K, const uint8_t* codes,
                        int64_t N, int L, int Lout, uint16_t* scores) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        const uint8_t* s = codes + n * L;
        for (int l = 0; l < Lout; l++)
            for (int k = 0; k < K; k++) {
                int len = (int)lens[k];
                uint16_t acc = 0;
                if (l <= L - len) {
                    for (int ind = 0; ind < len; ind++) {
                        int b = s[l + ind];
                        if (b < 4) acc = h_add(acc, pwms[k + K * (b + 4 * ind)]);
                    }
                    if (!(f16_to_f32(acc) > 0.0f)) acc = 0;
                }
                scores[k + (int64_t)K * (n + N * (int64_t)l)] = acc;
            }
    }
}

/*
 * The optimised CPU form of oracle_get_pos_scores_arr: what a careful CPU implementation of the SAME arithmetic
 * looks like (bench.py's cpu_baseline, kind "port"; the literal loop above stays as "port-literal").
 *   - gather form (one add per PWM position: the three w*0 = +-0 terms of :28-29 are exact no-ops);
 *   - 8 PWMs per AVX register, 5-8 registers (independent chains) in flight, binary16 adds as binary32 add + VCVTPS2PH
 *     (round to nearest even) + VCVTPH2PS:
 *     the binary32 sum of two binary16 values rounded again to binary16 is the correctly rounded binary16 sum
 *     (24 >= 2*11 + 2), i.e. the same bits as h_add, in the same order (ind ascending);
 *   - no dense (K, nb, 4L) tensor: threads own ranges of start positions l of a batch and append their hits, the
 *     ranges are concatenated in l order, which is findall's column-major order (k fastest, then n, then l).
 * codes: N rows of L bytes 0..3 (4 = all-zero column).  Returns the number of hits (records past cap are dropped),
 * or -1 when the CPU lacks AVX2/F16C (the caller falls back to the literal form).
 */
#include <immintrin.h>
#define FAST_CHAINS 8

typedef struct {
    oracle_hit* h;
    uint16_t* s;
    int64_t n, cap;
} hitbuf;

static void hb_push(hitbuf* b, uint32_t m, uint32_t n, uint32_t l, uint16_t sc) {
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 4096;
        b->h = (oracle_hit*)realloc(b->h, (size_t)b->cap * sizeof(oracle_hit));
        b->s = (uint16_t*)realloc(b->s, (size_t)b->cap * 2);
    }
    b->h[b->n].m = m;
    b->h[b->n].n = n;
    b->h[b->n].l = l;
    b->s[b->n] = sc;
    b->n++;
}

__attribute__((target("avx2,f16c"))) static void scan_rows_f16c(const float* wt, const float* lenmask, const int64_t* lens,
                                                                 int K, int Kp, int maxlen, const uint8_t* codes, int L, int64_t nb,
                                                                 int64_t n0, int l_lo, int l_hi, int uniform, hitbuf* out) {
    const int ng = Kp / 8;
    for (int l = l_lo; l < l_hi; l++) {
        for (int64_t n = 0; n < nb; n++) {
            const uint8_t* s = codes + (n0 + n) * L;
            for (int g0 = 0, gn = 0; g0 < ng; g0 += gn) {
                /* gn independent chains of 8 PWMs at a time: an add step is add + 2 converts (~15 cycles of latency), so 5-8
                 * chains in flight keep the ports busy where 4 left them idle most of the time; the groups of a bank are dealt out
                 * evenly (25 groups = 7 + 6 + 6 + 6) so that no block runs with a single chain */
                const int nblk = (ng + FAST_CHAINS - 1) / FAST_CHAINS, left = nblk - g0 * nblk / ng;   /* blocks still to go */
                gn = (ng - g0 + left - 1) / (left > 0 ? left : 1);
                if (gn > FAST_CHAINS) gn = FAST_CHAINS;
                if (gn > ng - g0) gn = ng - g0;
                __m256 acc[FAST_CHAINS];
                for (int j = 0; j < FAST_CHAINS; j++) acc[j] = _mm256_setzero_ps();
                const int span = L - l < maxlen ? L - l : maxlen;   /* positions past the read are never added: l <= L - len is checked below */
#define CHAIN_BLOCK(N)                                                                                                   \
    for (int ind = 0; ind < span; ind++) {                                                                               \
        const int b = s[l + ind];                                                                                        \
        if (b >= 4) continue;                                                                                            \
        const float* w = wt + ((size_t)ind * 4 + b) * Kp + (size_t)g0 * 8;                                               \
        _Pragma("GCC unroll 8") for (int j = 0; j < N; j++)                                                              \
            acc[j] = _mm256_cvtph_ps(_mm256_cvtps_ph(_mm256_add_ps(acc[j], _mm256_loadu_ps(w + 8 * j)),                  \
                                                      _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));                  \
    }
                if (uniform && gn == 8) { CHAIN_BLOCK(8) }
                else if (uniform && gn == 7) { CHAIN_BLOCK(7) }
                else if (uniform && gn == 6) { CHAIN_BLOCK(6) }
                else if (uniform && gn == 5) { CHAIN_BLOCK(5) }
                else {
                    for (int ind = 0; ind < span; ind++) {
                        const int b = s[l + ind];
                        if (b >= 4) continue;
                        const float* w = wt + ((size_t)ind * 4 + b) * Kp + (size_t)g0 * 8;
                        for (int j = 0; j < gn; j++) {
                            const __m256 sum = _mm256_add_ps(acc[j], _mm256_loadu_ps(w + 8 * j));
                            const __m256 r = _mm256_cvtph_ps(_mm256_cvtps_ph(sum, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
                            if (uniform) acc[j] = r;
                            else acc[j] = _mm256_blendv_ps(acc[j], r, _mm256_loadu_ps(lenmask + (size_t)ind * Kp + (size_t)(g0 + j) * 8));
                        }
                    }
                }
#undef CHAIN_BLOCK
                for (int j = 0; j < gn; j++) {
                    const int pos = _mm256_movemask_ps(_mm256_cmp_ps(acc[j], _mm256_setzero_ps(), _CMP_GT_OQ));
                    if (!pos) continue;
                    uint16_t hs[8];
                    _mm_storeu_si128((__m128i*)hs, _mm256_cvtps_ph(acc[j], _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
                    for (int u = 0; u < 8; u++) {
                        const int k = (g0 + j) * 8 + u;
                        if (((pos >> u) & 1) && k < K && l <= L - (int)lens[k])
                            hb_push(out, (uint32_t)(k + 1), (uint32_t)(n0 + n + 1), (uint32_t)(l + 1), hs[u]);
                    }
                }
            }
        }
    }
}

int64_t oracle_get_pos_scores_arr_fast(const uint16_t* pwms_in, const int64_t* lens, int K, int maxlen, const uint8_t* codes,
                                       int64_t N, int L, int rc, int batch_size, oracle_hit* found, uint16_t* score_record,
                                       int64_t cap) {
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("f16c")) return -1;
    const int Kp = (K + 7) / 8 * 8;
    int minlen = maxlen, maxtrue = 0;
    for (int k = 0; k < K; k++) {
        if (lens[k] < minlen) minlen = (int)lens[k];
        if (lens[k] > maxtrue) maxtrue = (int)lens[k];
    }
    const int uniform = minlen == maxtrue;
    const int Lout = L - minlen + 1;
    float* wt = (float*)calloc((size_t)maxtrue * 4 * Kp, 4);
    float* lenmask = (float*)calloc((size_t)maxtrue * Kp, 4);
    for (int k = 0; k < K; k++) {
        const int len = (int)lens[k];
        for (int ind = 0; ind < len; ind++) {
            uint32_t ones = 0xffffffffu;
            memcpy(&lenmask[(size_t)ind * Kp + k], &ones, 4);
            for (int a = 0; a < 4; a++) {
                const int sa = rc ? 3 - a : a, si = rc ? len - 1 - ind : ind;   /* reverse(pwm), :68 */
                wt[((size_t)ind * 4 + a) * Kp + k] = f16_to_f32(pwms_in[k + K * (sa + 4 * si)]);
            }
        }
    }
    int64_t nfound = 0;
    int nthr = 1;
#ifdef _OPENMP
    nthr = omp_get_max_threads();
#endif
    hitbuf* bufs = (hitbuf*)calloc((size_t)nthr, sizeof(hitbuf));
    for (int64_t n0 = 0; n0 < N && Lout > 0; n0 += batch_size) {   /* :71 */
        const int64_t nb = N - n0 < batch_size ? N - n0 : batch_size;
#pragma omp parallel num_threads(nthr)
        {
            int t = 0, T = 1;
#ifdef _OPENMP
            t = omp_get_thread_num();
            T = omp_get_num_threads();
#endif
            bufs[t].n = 0;
            const int lo = (int)((int64_t)Lout * t / T), hi = (int)((int64_t)Lout * (t + 1) / T);
            scan_rows_f16c(wt, lenmask, lens, K, Kp, maxtrue, codes, L, nb, n0, lo, hi, uniform, &bufs[t]);
        }
        for (int t = 0; t < nthr; t++) {                            /* l ranges in order = findall's order */
            for (int64_t i = 0; i < bufs[t].n; i++) {
                if (nfound < cap) {
                    found[nfound] = bufs[t].h[i];
                    score_record[nfound] = bufs[t].s[i];
                }
                nfound++;
            }
        }
    }
    for (int t = 0; t < nthr; t++) {
        free(bufs[t].h);
        free(bufs[t].s);
    }
    free(bufs);
    free(wt);
    free(lenmask);
    return nfound;
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
