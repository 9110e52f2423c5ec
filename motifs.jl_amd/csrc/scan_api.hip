// scan_api.hip — C-ABI entry points of the PWM scan (include/motifs_hip.h).
//
// Host side of what `get_pos_scores_arr` does around the kernel
// (src/inference/_h3_1_alignment.jl:57-87): build the (optionally reversed)
// padded PWM bank (:66-69), walk the sequences, collect `(m, n, l)` records and
// fp16 scores (:82-84).  Everything numeric runs on the device; there is no CPU
// fallback (a missing GPU is MOTIFS_ERR_NO_DEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "api_common.h"
#include "scan_kernels.h"

namespace motifs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error_text() { return g_err; }

struct PackedBank {
    std::vector<uint32_t> tab;
    std::vector<int32_t> lim;
    int KP = 0, nch = 0, lenp = 0, minlen = 0, maxlen_true = 0;
};

// Build the lane-major packed bank.  `rc` applies reverse(pwm) in both dims per
// motif (_h3_1_alignment.jl:68) before the zero padding of :66.
static int pack_bank(const uint16_t* pwms, const int64_t* lens, int K, int maxlen, int rc, int L, PackedBank& out) {
    if (!pwms || !lens || K <= 0 || maxlen <= 0) {
        set_error("pwm bank: null pointer or K/maxlen <= 0 (K=%d maxlen=%d)", K, maxlen);
        return MOTIFS_ERR_INVALID;
    }
    int minlen = maxlen, maxtrue = 0;
    for (int k = 0; k < K; k++) {
        if (lens[k] < 1 || lens[k] > maxlen) {
            set_error("pwm bank: lens[%d] = %lld outside 1..maxlen=%d", k, (long long)lens[k], maxlen);
            return MOTIFS_ERR_INVALID;
        }
        minlen = std::min<int>(minlen, (int)lens[k]);
        maxtrue = std::max<int>(maxtrue, (int)lens[k]);
    }
    out.lenp = scan_len_padded(maxtrue);
    out.minlen = minlen;
    out.maxlen_true = maxtrue;
    const int pairs = (K + 1) / 2;
    out.nch = (pairs + 63) / 64;
    out.KP = out.nch * 64;
    out.tab.assign((size_t)out.lenp * 4 * out.KP, 0u);
    out.lim.assign((size_t)2 * out.KP, -1);
    for (int k = 0; k < K; k++) {
        const int len = (int)lens[k];
        const int kp = k >> 1, hi = k & 1;
        out.lim[k] = L - len;  // < 0: no valid start; the kernel compares l <= lim
        for (int ind = 0; ind < len; ind++)
            for (int b = 0; b < 4; b++) {
                const int sb = rc ? 3 - b : b, si = rc ? len - 1 - ind : ind;
                const uint16_t w = pwms[k + (size_t)K * (sb + 4 * si)];
                if ((w & 0x7c00u) == 0x7c00u) {
                    set_error("pwm bank: entry (k=%d, a=%d, ind=%d) is Inf/NaN", k + 1, sb + 1, si + 1);
                    return MOTIFS_ERR_NONFINITE;
                }
                uint32_t& cell = out.tab[(size_t)(ind * 4 + b) * out.KP + kp];
                cell |= hi ? (uint32_t)w << 16 : (uint32_t)w;
            }
    }
    return MOTIFS_OK;
}

static float h2f_host(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (ex == 0) {
        float f = std::ldexp((float)man, -24);
        memcpy(&bits, &f, 4);
        bits |= sign;
    } else if (ex == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (ex + 112) << 23 | man << 13;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// Operands of the matrix-core candidate kernel (scan_mfma.hip) from the packed bank: PWM fragments in MFMA
// A-operand order and the per-PWM slack eps_k = 2^-10 * len_k * sum_ind max_a |w| in accumulator order.
struct MfmaBank {
    std::vector<uint32_t> afrag;   // [tiles][T][64][4]
    std::vector<float> cinit;      // [tiles][2][16]
    int ntiles = 0, T = 0;
    int uniform_eps = 0;           // afrag scaled by powers of two so that the slack is an inline constant for every PWM
};
static uint16_t f2h_exact_scaled(uint16_t h, int e) {   // h * 2^e for a finite binary16 h; the caller guarantees exactness
    if ((h & 0x7fffu) == 0) return h;
    const float f = std::ldexp(h2f_host(h), e);
    // exact conversion: |f| is a binary16 value times a power of two inside the normal range
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ex = (x >> 23) & 0xffu, man = x & 0x7fffffu;
    return (uint16_t)(sign | ((ex - 112) << 10) | (man >> 13));
}

static void pack_mfma(const PackedBank& bank, const int64_t* lens, int K, MfmaBank& out) {
    const int T = bank.lenp / 4, ntiles = bank.nch * 4;
    out.T = T;
    out.ntiles = ntiles;
    out.afrag.assign((size_t)ntiles * T * 64 * 4, 0u);
    out.cinit.assign((size_t)ntiles * 32, 1.0f);   // rows without a PWM: never negative
    auto wbits = [&](int k, int a, int ind) -> uint16_t {
        const uint32_t cell = bank.tab[(size_t)(ind * 4 + a) * bank.KP + (k >> 1)];
        return (uint16_t)((k & 1) ? cell >> 16 : cell);
    };
    // How far can the sequentially rounded sum s lie above the exact one S when it matters?  With S_i the exact and s_i
    // the rounded prefix sums, s_i = (s_{i-1} + w_i)(1 + d_i), |d_i| <= 2^-11, so e_i = s_i - S_i obeys
    // |e_i| <= |e_{i-1}| (1 + 2^-11) + 2^-11 |S_i| and |e_n| <= g 2^-11 sum_{i=2..n} |S_i|, g = (1 + 2^-11)^n (the first
    // add is exact).  The filter only has to keep the windows with s_n > 0: either S_n > 0 (kept by any slack) or
    // -|e_n| <= S_n <= 0.  For those the prefix sums are pinned from both ends: S_i lies in [-N_i, P_i] (N_i, P_i = the
    // most negative / most positive a prefix of i positions can be) and in [S_n - P'_i, S_n + N'_i] (the same for the
    // suffix after i), hence |S_i| <= B_i + |e_n| with B_i = max(min(N_i, P'_i), min(P_i, N'_i)) and
    // |e_n| <= c sum B_i / (1 - c n), c = g 2^-11.  With log-odds weights (large negative, small positive) B_i is a
    // fraction of the plain bound sum_{j<=i} max_a |w_j| (0.085 against 0.42 for the bank of BASELINE configs[1]), which
    // means fewer false candidates for stage_hits.  Subnormal adds round by at most 2^-25 each.
    std::vector<float> eps(K);
    float eps_max = 0.f, wmax = 0.f, wmin_nz = INFINITY;
    std::vector<double> pos, neg;
    for (int k = 0; k < K; k++) {
        const int len = (int)lens[k];
        pos.assign(len, 0.0);
        neg.assign(len, 0.0);
        double A = 0;
        for (int ind = 0; ind < len; ind++) {
            double mx = 0;
            for (int a = 0; a < 4; a++) {
                const double w = h2f_host(wbits(k, a, ind));
                pos[ind] = std::max(pos[ind], w);        // an all-zero data column adds 0: inside [-neg, pos] too
                neg[ind] = std::max(neg[ind], -w);
                mx = std::max(mx, std::fabs(w));
                if (w != 0) wmin_nz = std::min(wmin_nz, (float)std::fabs(w));
            }
            A += mx;
            wmax = std::max(wmax, (float)mx);
        }
        double Psuf = 0, Nsuf = 0;
        for (int ind = 0; ind < len; ind++) Psuf += pos[ind], Nsuf += neg[ind];
        double Ppre = 0, Npre = 0, E = 0;
        for (int ind = 0; ind < len; ind++) {
            Ppre += pos[ind], Npre += neg[ind];
            Psuf -= pos[ind], Nsuf -= neg[ind];
            if (ind > 0) E += std::max(std::min(Npre, std::max(Psuf, 0.0)), std::min(Ppre, std::max(Nsuf, 0.0)));
        }
        const double c = std::pow(1.0 + std::ldexp(1.0, -11), len) * std::ldexp(1.0, -11);
        eps[k] = (float)(1.001 * c * E / (1.0 - c * len) + std::ldexp((double)len, -24) + std::ldexp(A, -20));
        if (A * 1.04 >= 60000.0 || c * len >= 0.5) eps[k] = INFINITY;      // a partial sum may overflow binary16 (or the bound has no fixed point: len > ~900): keep every window
        eps_max = std::max(eps_max, eps[k]);
    }
    // The kernel wants "candidate" as a SET sign bit (its packing then needs no complement), so the operands are negated:
    // A = -w, C = -eps, result -(S + eps) < 0 <=> S > -eps (S = -eps exactly gives +0 and is dropped: the bound is strict).
    // One slack for all: scale the bank by 2^e (exact in binary16) so that eps_max * 2^e <= 4.0 and C can be an inline
    // constant of the MFMA instead of 16 registers.  The compiler only encodes it inline when every accumulator chain of
    // the wave has a constant of its own, so tile g of a wave's group (g = tile mod PG) is scaled by 2^(e-g) and tested
    // against -4 / 2^g: -(S 2^(e-g) + 4 / 2^g) < 0 is S > -4 / 2^e for every g, with 4 / 2^e >= eps_k for every k.
    const int PG = cand_tile_group(bank.lenp);
    int e = 0;
    bool uniform = std::isfinite(eps_max) && eps_max > 0.f && eps_max <= 4.0f;
    if (uniform) {
        e = (int)std::floor(std::log2(4.0 / eps_max));
        e = std::min(e, 8);
        // scaled weights must stay normal binary16 numbers (no overflow, no bits lost at the bottom)
        if (wmax * std::ldexp(1.0f, e) >= 32768.0f ||
            (std::isfinite(wmin_nz) && wmin_nz * std::ldexp(1.0f, std::min(0, e - (PG - 1))) < 6.2e-5f))
            uniform = false;
    }
    out.uniform_eps = uniform ? 1 : 0;
    for (int tile = 0; tile < ntiles; tile++) {
        for (int t = 0; t < T; t++)
            for (int lane = 0; lane < 64; lane++) {
                const int rho = lane & 31, hh = lane >> 5;
                const int q = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);   // PWM that accumulator order expects in row rho
                const int k = tile * 32 + q;
                uint16_t hv[8];
                for (int j = 0; j < 8; j++) {
                    const int kk = 16 * t + 8 * hh + j, ind = kk >> 2, a = kk & 3;
                    hv[j] = (k < K && ind < (int)lens[k]) ? wbits(k, a, ind) : (uint16_t)0;
                    if (uniform) {
                        if (k < K) hv[j] = f2h_exact_scaled(hv[j], e - tile % PG);
                        else if (ind == 0) hv[j] = 0xec00u;   // -4096: a row without a PWM never becomes a candidate
                    }
                    hv[j] ^= 0x8000u;                         // negated operands (above)
                }
                uint32_t* dst = &out.afrag[(((size_t)tile * T + t) * 64 + lane) * 4];
                for (int u = 0; u < 4; u++) dst[u] = (uint32_t)hv[2 * u] | ((uint32_t)hv[2 * u + 1] << 16);
            }
        for (int q = 0; q < 32; q++) {
            const int k = tile * 32 + q;
            if (k < K) out.cinit[((size_t)tile * 2 + (q >> 4)) * 16 + (q & 15)] = -eps[k];
        }
    }
}

// padded PWM length of a bank (-1: longer than the kernels support); banks past 32 positions exist on the
// matrix-core path only
static int bank_lenp(const int64_t* lens, int K) {
    int64_t mx = 0;
    for (int k = 0; lens && k < K; k++) mx = std::max<int64_t>(mx, lens[k]);
    return scan_len_padded((int)std::max<int64_t>(mx, 1));
}

static int pick_cpb(int nch) {
    int cpb = 1;
    while (cpb * 2 <= nch && cpb * 2 <= SCAN_WAVES) cpb *= 2;
    return cpb;
}

static int pick_spw(int64_t N, int nch) {
    const int64_t items = N * nch;
    int64_t spw = items / 16384;  // aim for >= 16k wave-items over the launch
    if (spw < 1) spw = 1;
    if (spw > 16) spw = 16;
    return (int)spw;
}

static int upload_bank(motifs_ctx* c, const PackedBank& bank) {
    MOTIFS_HIP_CHECK(c->tab.reserve(bank.tab.size() * 4));
    MOTIFS_HIP_CHECK(c->lim.reserve(bank.lim.size() * 4));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(c->tab.p, bank.tab.data(), bank.tab.size() * 4, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(c->lim.p, bank.lim.data(), bank.lim.size() * 4, hipMemcpyHostToDevice, c->stream));
    // the vectors die with the caller's frame: make sure the copies are done
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MOTIFS_OK;
}

// hit records -> their places in the (K, N, ld_l) tensor (the fallback form of a17: zeros + hits)
__global__ __launch_bounds__(256) void scatter_hits_dense(const HitRec* __restrict__ hits, const uint16_t* __restrict__ sc, int64_t n,
                                                          int K, int64_t N, int64_t n0, uint16_t* __restrict__ dense) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HitRec h = hits[i];
    dense[(size_t)(h.m - 1) + (size_t)K * ((size_t)(h.n - 1 - n0) + (size_t)N * (h.l - 1))] = sc[i];
}

}  // namespace motifs

using namespace motifs;

// The bank of this call on the device, from the context's cache when the caller passes the same bank again.
static int cached_bank(motifs_ctx* c, const uint16_t* pwms, const int64_t* lens, int K, int maxlen, int rc, int L, BankSlot** out) {
    BankSlot& bs = c->bank_slot[rc ? 1 : 0];
    const size_t nb_p = (size_t)K * 4 * maxlen * 2, nb_l = (size_t)K * 8;
    const int32_t shape[4] = {K, maxlen, rc, L};
    std::vector<uint8_t> key(sizeof(shape) + nb_p + nb_l);
    if (K > 0 && maxlen > 0 && pwms && lens) {
        memcpy(key.data(), shape, sizeof(shape));
        memcpy(key.data() + sizeof(shape), pwms, nb_p);
        memcpy(key.data() + sizeof(shape) + nb_p, lens, nb_l);
        if (key == bs.key) {
            *out = &bs;
            return MOTIFS_OK;
        }
    }
    bs.key.clear();
    PackedBank bank;
    const int rcode = pack_bank(pwms, lens, K, maxlen, rc, L, bank);   // validates the arguments
    if (rcode) return rcode;
    MfmaBank mb;
    pack_mfma(bank, lens, K, mb);
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(bs.tab.reserve(bank.tab.size() * 4));
    MOTIFS_HIP_CHECK(bs.lim.reserve(bank.lim.size() * 4));
    MOTIFS_HIP_CHECK(bs.afrag.reserve(mb.afrag.size() * 4));
    MOTIFS_HIP_CHECK(bs.cinit.reserve(mb.cinit.size() * 4));
    // a scan that still reads the slot's old contents may be in flight on the stream: the copies queue behind it
    MOTIFS_HIP_CHECK(hipMemcpyAsync(bs.tab.p, bank.tab.data(), bank.tab.size() * 4, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(bs.lim.p, bank.lim.data(), bank.lim.size() * 4, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(bs.afrag.p, mb.afrag.data(), mb.afrag.size() * 4, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(bs.cinit.p, mb.cinit.data(), mb.cinit.size() * 4, hipMemcpyHostToDevice, c->stream));
    // re-scoring table for stage_hits: [k][ind][5] halves, rows padded to an odd dword count
    const int rs_dw = ((bank.lenp * 5 + 1) / 2) | 1;
    bs.tabk_stride = rs_dw * 2;
    std::vector<uint16_t> tabk((size_t)K * bs.tabk_stride + 2, 0);
    for (int k = 0; k < K; k++)
        for (int ind = 0; ind < (int)lens[k]; ind++)
            for (int b = 0; b < 4; b++) {
                const uint32_t cell = bank.tab[(size_t)(ind * 4 + b) * bank.KP + (k >> 1)];
                tabk[(size_t)k * bs.tabk_stride + ind * 5 + b] = (uint16_t)((k & 1) ? cell >> 16 : cell);
            }
    MOTIFS_HIP_CHECK(bs.tabk.reserve(tabk.size() * 2));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(bs.tabk.p, tabk.data(), tabk.size() * 2, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));   // the host vectors die with this frame
    bs.KP = bank.KP;
    bs.nch = bank.nch;
    bs.lenp = bank.lenp;
    bs.minlen = bank.minlen;
    bs.maxlen_true = bank.maxlen_true;
    bs.ntiles = mb.ntiles;
    bs.uniform_eps = mb.uniform_eps;
    bs.key.swap(key);
    *out = &bs;
    return MOTIFS_OK;
}

// The arguments the candidate kernel and the row kernels share, for `ns` reads starting at `codes` (cells in c->cnt).
static void scan_args(motifs_ctx* c, const BankSlot& bank, int K, const uint8_t* codes, int64_t ns, int L, int Lout, int batch,
                      int rpr, CandArgs& a, FillArgs& f, void* cells = nullptr) {
    if (!cells) cells = c->cnt.p;
    const int PG = cand_tile_group(bank.lenp);
    const int parts = (batch + rpr - 1) / rpr;
    const int64_t nb = (ns + batch - 1) / batch;
    a.afrag = (const uint4*)bank.afrag.p;
    a.cinit = (const float*)bank.cinit.p;
    a.codes = codes;
    a.cells = (uint32_t*)cells;
    a.centries = nullptr;
    a.afrag2 = nullptr;
    a.cells2 = nullptr;
    a.centries2 = nullptr;
    f.centries = nullptr;
    a.lenp = bank.lenp;
    a.ntiles = bank.ntiles;
    a.uniform_eps = bank.uniform_eps;
    a.d.N = ns;
    a.d.L = L;
    a.d.pitch = motifs_codes_pitch(L);
    a.d.Lout = Lout;
    a.d.nch = bank.nch;
    a.d.cgc = bank.nch;               // plain cell order; scan_hits_mfma narrows it to the chunk group of its plan
    a.d.batch = batch;
    a.d.ohlen = (Lout + 31) / 32 * 32 + bank.lenp;
    a.d.used_tiles = (K + 31) / 32;
    a.d.nseg = 1;
    a.d.seg_tiles = (Lout + 7) / 8;
    a.d.ohseg = a.d.ohlen;
    {
        int64_t spw = ns * (bank.ntiles / PG) / 16384;
        a.d.spw = (int)std::max<int64_t>(1, std::min<int64_t>(spw, 8));   // measured: 4-8 reads per wave best, 16 is 12 % slower
    }
    f.masks = (const uint4*)cells;
    f.parts = parts;
    f.rpr = rpr;
    f.nrows = nb * Lout * parts;
    f.tab = (const uint32_t*)bank.tab.p;
    f.tabk = (const uint16_t*)bank.tabk.p;
    f.tabk_stride = bank.tabk_stride;
    f.codes = codes;
    f.nch = bank.nch;
    f.batch = batch;
    f.Lout = Lout;
    f.LoutP = Lout;
    f.lenp = bank.lenp;
    f.KP = bank.KP;
    f.pitch = a.d.pitch;
    f.lim = (const int32_t*)bank.lim.p;
    f.N = ns;
    f.K = K;
    f.L = L;
    f.lim_min = L - bank.maxlen_true;
    uint32_t d = (uint32_t)bank.nch, sh = 0;
    while ((1u << sh) < d) sh++;
    f.div_nch.d = d;
    f.div_nch.s = sh;
    f.div_nch.m = d == 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}

// Geometry of the hit-record pipeline for one bank: rows, staging slots, bytes per ordering batch, the super-batch size under the
// workspace bound.  Compact entries need a bank scaled to one slack with PWMs of up to 20 positions (cand_compact_ok); chunk groups
// (scan_mfma.hip) need compact entries.  The bound counts everything a launch allocates per ordering batch: cells (16 B each),
// entries (4 B per cell), staged words (8 B per cell) - twice the first two when one candidate launch serves both strands.
struct HitGeom {
    bool compact;
    int cgc, ncg;              // chunk groups (0 = off)
    int rpr, parts, row_slots;
    size_t per_batch, stage_per_batch;
    int64_t nb_max;            // ordering batches per launch
};
// two_strands: 0 = a launch serves one strand; 1 = one candidate launch writes both strands' cells / entries, the later stages run per
// strand and share one set of staged words (scan_hits_mfma's cand_mode); 2 = every stage serves both strands (scan_hits_pair: staged words,
// row counts and offsets twice as well)
static HitGeom hit_geom(const motifs_ctx* c, const BankSlot& bank, int K, int Lout, int batch, bool emit, int64_t N, int two_strands = 0) {
    HitGeom g{};
    g.compact = c->compact_cells && bank.uniform_eps && bank.lenp <= 20;
    g.cgc = (g.compact && c->cg_chunks != 0) ? stage_cg_chunks(K, bank.nch, bank.lenp, bank.tabk_stride, c->cg_chunks > 0 ? c->cg_chunks : 0) : 0;
    g.ncg = g.cgc ? (bank.nch + g.cgc - 1) / g.cgc : 1;
    g.rpr = g.cgc ? 512 / g.cgc : stage_row_reads(bank.nch);
    g.parts = (batch + g.rpr - 1) / g.rpr;
    g.row_slots = g.cgc ? 2 * 512 : 2 * g.rpr * bank.nch;            // staged hits per (row, group) before the slow path
    g.per_batch = (size_t)Lout * batch * bank.nch * 16;
    // (chunk groups: + the per-read hit counts in front of every (row, group)'s slots, 16 bits per read)
    g.stage_per_batch = emit ? (size_t)Lout * g.parts * g.ncg * (g.row_slots + (g.cgc ? 256 / g.cgc : 0)) * 4 : 0;
    const size_t cand_bytes = g.per_batch + (g.compact ? g.per_batch / 4 : 0);
    const size_t rows_bytes = (size_t)Lout * g.parts * g.ncg * 8;            // row counts + exclusive offsets
    const size_t all = (two_strands ? 2 : 1) * cand_bytes + (two_strands == 2 ? 2 : 1) * (g.stage_per_batch + rows_bytes);
    int64_t nb_max = (int64_t)(c->ws_limit / all);
    // compact entries: the kernels keep 32-bit BYTE offsets into the entry array (2 bytes per half cell = per_batch / 4 bytes per
    // batch), so a launch holds fewer than 2^31 entries
    if (g.compact) nb_max = std::min<int64_t>(nb_max, (int64_t)((((size_t)1 << 31) - 1) / (g.per_batch / 8)));
    g.nb_max = std::max<int64_t>(1, std::min<int64_t>(nb_max, (N + batch - 1) / batch));
    if (g.compact && (size_t)g.nb_max * (g.per_batch / 8) >= ((size_t)1 << 31)) {     // a single ordering batch past the offsets' range
        g.compact = false;
        g.cgc = 0;
        g.ncg = 1;
        g.rpr = stage_row_reads(bank.nch);
        g.parts = (batch + g.rpr - 1) / g.rpr;
        g.row_slots = 2 * g.rpr * bank.nch;
        g.stage_per_batch = emit ? (size_t)Lout * g.parts * g.row_slots * 4 : 0;
    }
    return g;
}

// Hit records through the matrix cores: candidates (scan_cand_kernel) -> exact verification, staged hits and
// row counts (stage_hits) -> scan -> records (emit_records); no host round trip in between.  The bank has already been uploaded (tab, lim).
static int scan_hits_mfma(motifs_ctx* c, const BankSlot& bank, int K, const uint8_t* codes_dev, int64_t N, int L, int Lout, int64_t n0,
                          int batch, motifs_hit* hits_dev, uint16_t* hit_scores_dev, int64_t cap, int64_t* n_out,
                          int64_t* per_pwm_counts_dev, int slot = 0, bool finish = true, int cand_mode = 0, const BankSlot* bank2 = nullptr) {
    // cand_mode (gpu_scan's two strands through ONE candidate launch, single super-batch only): 1 = this is the forward strand and the
    // launch also takes bank2, the reverse strand's bank, writing its entries / cells into the second buffer set; 2 = this is the
    // reverse strand: its candidates are already in the second buffer set, no launch
    // slot: which pair of running totals in c->small this strand uses; finish = false: everything is enqueued, the
    // total stays on the device at totals_of(slot) and the caller reads it after its own synchronisation
    const bool emit = hits_dev != nullptr && cap > 0;
    const HitGeom hg = hit_geom(c, bank, K, Lout, batch, emit, N, cand_mode != 0 ? 1 : 0);
    const int rpr = hg.rpr;                                          // reads per row of cells
    const int parts = hg.parts;
    const int row_slots = hg.row_slots;                              // staged hits per row (and chunk group) before the slow path
    const size_t per_batch = hg.per_batch, stage_per_batch = hg.stage_per_batch;
    const int64_t nb_max = hg.nb_max;
    const int64_t sb = nb_max * batch;
    const int64_t rows_max = nb_max * Lout * parts;
    MOTIFS_HIP_CHECK(c->cnt.reserve((size_t)nb_max * per_batch));
    const bool compact = hg.compact;
    if (cand_mode != 0 && (!compact || sb < N)) {
        set_error("internal: a two-strand candidate launch needs compact entries and a single super-batch");
        return MOTIFS_ERR_INVALID;
    }
    if (compact) MOTIFS_HIP_CHECK(c->centries.reserve((size_t)nb_max * per_batch / 4));
    if (cand_mode == 1) {
        MOTIFS_HIP_CHECK(c->cnt2.reserve((size_t)nb_max * per_batch));
        MOTIFS_HIP_CHECK(c->centries2.reserve((size_t)nb_max * per_batch / 4));
    }
    void* const cells_buf = cand_mode == 2 ? c->cnt2.p : c->cnt.p;
    void* const entries_buf = cand_mode == 2 ? c->centries2.p : c->centries.p;
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)rows_max * hg.ncg * 4));
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)((rows_max + 1023) / 1024) * 8));
    MOTIFS_HIP_CHECK(c->rowx.reserve((size_t)rows_max * 4));
    if (emit) MOTIFS_HIP_CHECK(c->staging.reserve((size_t)nb_max * stage_per_batch));
    MOTIFS_HIP_CHECK(c->small.reserve(64));
    // small: [0], [1] record totals (ping-pong between super-batches)
    // (the first super-batch takes no base, and the last one writes its total into pinned host memory itself: no memset
    // and no copy kernel around the launches - 9 us per strand)
    int64_t* totals = (int64_t*)c->small.p + 2 * slot;
    int64_t* h_total = (int64_t*)c->pinned + slot;
    if (N <= 0) *h_total = 0;

    int launch_no = 0;
    for (int64_t s0 = 0; s0 < N; s0 += sb, launch_no++) {
        const int64_t ns = std::min<int64_t>(sb, N - s0);
        const int64_t nb = (ns + batch - 1) / batch;
        CandArgs a{};
        FillArgs f{};
        scan_args(c, bank, K, codes_dev + (size_t)s0 * motifs_codes_pitch(L), ns, L, Lout, batch, rpr, a, f, cells_buf);
        if (compact) {
            a.centries = (uint16_t*)entries_buf;
            f.centries = (const uint16_t*)entries_buf;
        }
        if (hg.cgc) a.d.cgc = hg.cgc;        // entries and cells group-major: a group's cells contiguous for the block that re-scores them
        if (cand_mode == 1) {
            a.afrag2 = (const uint4*)bank2->afrag.p;
            a.cells2 = (uint32_t*)c->cnt2.p;
            a.centries2 = (uint16_t*)c->centries2.p;
        }
        f.row_sum = (uint32_t*)c->tilesum.p;
        f.blk_base = (unsigned long long*)c->off.p;
        f.staging = (uint32_t*)c->staging.p;
        f.row_slots = row_slots;
        f.cgc = hg.cgc;
        f.ncg = hg.ncg;
        f.row_excl = (uint32_t*)c->rowx.p;
        f.base_in = launch_no == 0 ? nullptr : totals + (launch_no & 1);
        f.total = totals + ((launch_no + 1) & 1);
        f.total_host = s0 + sb >= N ? h_total : nullptr;
        f.cap = cap;
        f.hits = (HitRec*)hits_dev;
        f.hit_scores = hit_scores_dev;
        f.pwm_counts = per_pwm_counts_dev;   // zeroed by the caller of this function; only bins k < K are ever touched
        f.n0 = n0 + s0;
        f.hist_bins = (per_pwm_counts_dev && 2 * bank.KP <= FILL_HIST_MAX && !hg.cgc) ? 2 * bank.KP : 0;   // (chunk groups keep their own)
        // cells of reads a short last batch does not have are never written by the scan: the walk over compact entries stops at the
        // reads that exist (row_geom's nvalid); the 128-bit cells are cleared
        if (ns < nb * batch && !compact) MOTIFS_HIP_CHECK(hipMemsetAsync((char*)c->cnt.p + (size_t)(nb - 1) * per_batch, 0, per_batch, c->stream));
        if (cand_mode != 2) {
            KernelTimer t(c, KS_SCAN_COUNT, true);
            const hipError_t le = launch_cand(a, c->stream, t.e0, t.e1);
            t.stamped = le == hipSuccess;            // events of a failed launch are never stamped: do not queue them
            MOTIFS_HIP_CHECK(le);
        }
        {
            KernelTimer t(c, KS_SCAN_OFFSETS);
            MOTIFS_HIP_CHECK(launch_stage_hits(f, emit ? 1 : 0, c->stream));
            MOTIFS_HIP_CHECK(launch_row_scan(f, c->stream));
        }
        if (emit) {
            KernelTimer t(c, KS_SCAN_FILL);
            MOTIFS_HIP_CHECK(launch_emit_records(f, c->stream));
        }
    }
    c->scan_plan[0] = compact, c->scan_plan[1] = hg.cgc, c->scan_plan[2] = hg.ncg, c->scan_plan[3] = launch_no;
    // records are written up to cap in any case; the total says whether they all fitted
    if (!finish) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int64_t emitted = *h_total;
    const bool too_small = emitted > cap;
    *n_out = emitted;
    if (too_small && !(cap == 0 && hits_dev == nullptr)) {
        set_error("hit buffer too small: need %lld records, cap %lld", (long long)emitted, (long long)cap);
        return MOTIFS_ERR_BUFFER_TOO_SMALL;
    }
    return MOTIFS_OK;
}

// gpu_scan's two strands through ONE launch of every stage (single super-batch, compact entries, no chunk groups: what
// motifs_pwm_scan_hits_both_dev checks): candidates of both banks (scan_cand_kernel_q with afrag2), then stage_hits, the row scans and
// emit_records with the strand as blockIdx.y - 5 launches instead of 9 and no fill in front of them.  A step on a rank's 12 500 reads
// of BASELINE configs[2] is 0.26 ms, of which the launches' gaps were a tenth (bench.py strong_proxy).
static int scan_hits_pair(motifs_ctx* c, BankSlot* const bs[2], int K, const uint8_t* codes_dev, int64_t N, int L, int Lout, int64_t n0, int batch,
                          motifs_hit* const hits[2], uint16_t* const scores[2], int64_t cap, int64_t* counts2_dev) {
    const bool emit = hits[0] != nullptr && cap > 0;
    const HitGeom hg = hit_geom(c, *bs[0], K, Lout, batch, emit, N, 2);
    const int64_t nb = (N + batch - 1) / batch;
    const int64_t rows = nb * Lout * hg.parts, nblk = (rows + 1023) / 1024;
    const size_t stage_words = (size_t)nb * hg.stage_per_batch / 4;
    MOTIFS_HIP_CHECK(c->cnt.reserve((size_t)nb * hg.per_batch));
    MOTIFS_HIP_CHECK(c->cnt2.reserve((size_t)nb * hg.per_batch));
    MOTIFS_HIP_CHECK(c->centries.reserve((size_t)nb * hg.per_batch / 4));
    MOTIFS_HIP_CHECK(c->centries2.reserve((size_t)nb * hg.per_batch / 4));
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)rows * 4 * 2));
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)nblk * 8 * 2));
    MOTIFS_HIP_CHECK(c->rowx.reserve((size_t)rows * 4 * 2));
    if (emit) MOTIFS_HIP_CHECK(c->staging.reserve(stage_words * 4 * 2));
    MOTIFS_HIP_CHECK(c->small.reserve(64));
    CandArgs a[2]{};
    FillArgs f[2]{};
    for (int rc = 0; rc < 2; rc++) {
        scan_args(c, *bs[rc], K, codes_dev, N, L, Lout, batch, hg.rpr, a[rc], f[rc], rc ? c->cnt2.p : c->cnt.p);
        f[rc].centries = (const uint16_t*)(rc ? c->centries2.p : c->centries.p);
        f[rc].row_sum = (uint32_t*)c->tilesum.p + (size_t)rc * rows;
        f[rc].blk_base = (unsigned long long*)c->off.p + (size_t)rc * nblk;
        f[rc].staging = (uint32_t*)c->staging.p + (size_t)rc * stage_words;
        f[rc].row_slots = hg.row_slots;
        f[rc].row_excl = (uint32_t*)c->rowx.p + (size_t)rc * rows;
        f[rc].base_in = nullptr;
        f[rc].total = (int64_t*)c->small.p + 2 * rc + 1;
        f[rc].total_host = (int64_t*)c->pinned + rc;
        // records in stream order: the host will poll pinned words 8 + rc for this call's ticket (written behind the totals by the row scan)
        f[rc].ticket_host = c->records_async ? (int64_t*)c->pinned + 8 + rc : nullptr;
        f[rc].ticket = c->ticket_seq + 1;
        f[rc].cap = cap;
        f[rc].hits = (HitRec*)hits[rc];
        f[rc].hit_scores = scores[rc];
        f[rc].pwm_counts = counts2_dev ? counts2_dev + (size_t)rc * K : nullptr;
        f[rc].n0 = n0;
        f[rc].hist_bins = (counts2_dev && 2 * bs[rc]->KP <= FILL_HIST_MAX) ? 2 * bs[rc]->KP : 0;
    }
    a[0].centries = (uint16_t*)c->centries.p;
    a[0].d.zero_ptr = (unsigned long long*)counts2_dev;
    a[0].d.zero_n = counts2_dev ? 2 * K : 0;
    a[0].afrag2 = (const uint4*)bs[1]->afrag.p;
    a[0].cells2 = (uint32_t*)c->cnt2.p;
    a[0].centries2 = (uint16_t*)c->centries2.p;
    {
        KernelTimer t(c, KS_SCAN_COUNT, true);
        const hipError_t le = launch_cand(a[0], c->stream, t.e0, t.e1);
        t.stamped = le == hipSuccess;
        MOTIFS_HIP_CHECK(le);
    }
    {
        KernelTimer t(c, KS_SCAN_OFFSETS);
        MOTIFS_HIP_CHECK(launch_stage_hits(f[0], emit ? 1 : 0, c->stream, &f[1]));
        MOTIFS_HIP_CHECK(launch_row_scan(f[0], c->stream, &f[1]));
    }
    if (emit) {
        KernelTimer t(c, KS_SCAN_FILL);
        MOTIFS_HIP_CHECK(launch_emit_records(f[0], c->stream, &f[1]));
    }
    if (c->records_async) {
        c->ticket_seq++;
        c->ticket_wait = true;
    }
    c->scan_plan[0] = 1, c->scan_plan[1] = 0, c->scan_plan[2] = 1, c->scan_plan[3] = 1;
    return MOTIFS_OK;
}

extern "C" {

int motifs_abi_version(void) { return MOTIFS_ABI_VERSION; }
const char* motifs_last_error(void) { return motifs::g_err; }

int motifs_ctx_create(int device, motifs_ctx** out) {
    if (!out) {
        set_error("motifs_ctx_create: out is NULL");
        return MOTIFS_ERR_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device visible (%s); libmotifs_hip has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return MOTIFS_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (0..%d)", device, ndev - 1);
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MOTIFS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
        return MOTIFS_ERR_NO_DEVICE;
    }
    motifs_ctx* c = new motifs_ctx();
    c->device = device;
    MOTIFS_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    MOTIFS_HIP_CHECK(hipHostMalloc(&c->pinned, 256, hipHostMallocDefault));
    memset(c->pinned, 0, 256);        // (words 8, 9: the tickets the row scan writes in stream-order mode; they count up from 1)
    const char* ev = getenv("MOTIFS_SCAN_VALU");
    c->scan_valu = ev && ev[0] == '1';
    if (const char* cgv = getenv("MOTIFS_CG_CHUNKS")) c->cg_chunks = atoi(cgv);      // chunk groups: 0 = never, 1 / 2 / 4 = that size for every bank that can take it (tests, A/B)
    *out = c;
    return MOTIFS_OK;
}

void motifs_ctx_destroy(motifs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->tab, &c->lim, &c->cnt, &c->off, &c->tilesum, &c->small, &c->codes, &c->hits_tmp,
                      &c->scores_tmp, &c->pwmcnt, &c->data_tmp, &c->afrag, &c->cinit, &c->staging, &c->rowx, &c->dp_scratch, &c->centries, &c->cnt2, &c->centries2, &c->cm_lens})
        b->release();
    for (BankSlot& bs : c->bank_slot)
        for (DevBuf* b : {&bs.tab, &bs.lim, &bs.afrag, &bs.cinit, &bs.tabk}) b->release();

    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pin_stage) (void)hipHostFree(c->pin_stage);
    resolve_timing(c);
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    if (c->ev_totals) (void)hipEventDestroy(c->ev_totals);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int motifs_ctx_set_stream(motifs_ctx* c, void* hip_stream) {
    if (!c) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    // NULL is a stream too: HIP's null stream, which is what torch hands out as its default "current stream".
    // (ABI 1 read NULL as "make a private stream": work queued by the host framework on its default stream was
    // then silently unordered against the library's kernels.)
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return MOTIFS_OK;
}

int motifs_ctx_use_private_stream(motifs_ctx* c) {
    if (!c) return MOTIFS_ERR_INVALID;
    if (c->own_stream) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    hipStream_t st = nullptr;
    MOTIFS_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->stream = st;
    c->own_stream = true;
    return MOTIFS_OK;
}

int motifs_ctx_get_stream(motifs_ctx* c, void** hip_stream_out) {
    if (!c || !hip_stream_out) return MOTIFS_ERR_INVALID;
    *hip_stream_out = (void*)c->stream;
    return MOTIFS_OK;
}

int motifs_ctx_set_records_in_stream_order(motifs_ctx* c, int on) {
    if (!c) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    if (on && !c->ev_totals) MOTIFS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_totals, hipEventDisableTiming));
    c->records_async = on != 0;
    return MOTIFS_OK;
}

int motifs_ctx_set_workspace_limit(motifs_ctx* c, size_t bytes) {
    if (!c) return MOTIFS_ERR_INVALID;
    c->ws_limit = bytes ? bytes : (size_t)8 << 30;
    return MOTIFS_OK;
}

int motifs_ctx_synchronize(motifs_ctx* c) {
    if (!c) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MOTIFS_OK;
}

int motifs_ctx_enable_timing(motifs_ctx* c, int on) {
    if (!c) return MOTIFS_ERR_INVALID;
    c->timing = on == 0 ? 0u : (on & 1) ? 0xffffffffu : ((uint32_t)on >> 1);
    // events for the launches to come are made now: a timed launch that found the pool empty paid two hipEventCreate (~10 us of host
    // time per step of bench.py's timed region, whose launches are all stamped before any is resolved)
    if (c->timing && hipSetDevice(c->device) == hipSuccess)
        while (c->free_events.size() < 128) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            c->free_events.push_back(e);
        }
    return MOTIFS_OK;
}

int motifs_ctx_reset_timing(motifs_ctx* c) {
    if (!c) return MOTIFS_ERR_INVALID;
    resolve_timing(c);
    for (int i = 0; i < KS_COUNT_; i++) {
        c->kernel_ms[i] = 0;
        c->kernel_launches[i] = 0;
    }
    return MOTIFS_OK;
}

int motifs_ctx_scan_plan(motifs_ctx* c, int32_t plan[4]) {
    if (!c || !plan) {
        set_error("motifs_ctx_scan_plan: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    for (int i = 0; i < 4; i++) plan[i] = c->scan_plan[i];
    return MOTIFS_OK;
}

int motifs_ctx_kernel_ms(motifs_ctx* c, int slot, double* ms, int64_t* launches) {
    if (!c || slot < 0 || slot >= KS_COUNT_) {
        set_error("motifs_ctx_kernel_ms: bad slot %d", slot);
        return MOTIFS_ERR_INVALID;
    }
    resolve_timing(c);
    if (ms) *ms = c->kernel_ms[slot];
    if (launches) *launches = c->kernel_launches[slot];
    return MOTIFS_OK;
}

// row = L codes, zero padding to a multiple of 4, then 4 flag bytes (byte 0: row has an all-zero column)
int motifs_codes_pitch(int L) { return ((L + 3) & ~3) + 4; }
size_t motifs_codes_bytes(int64_t N, int L) { return (size_t)N * motifs_codes_pitch(L) + SCAN_GUARD_BYTES; }

int motifs_encode_dev(motifs_ctx* c, const void* data_dev, int kind, int64_t N, int L, uint8_t* codes_dev,
                      int32_t* bad_flag_dev) {
    if (!c || (N > 0 && (!data_dev || !codes_dev)) || N < 0 || L <= 0 || kind < 0 || kind > 2) {
        set_error("motifs_encode_dev: bad argument (N=%lld L=%d kind=%d)", (long long)N, L, kind);
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    KernelTimer t(c, KS_ENCODE);
    MOTIFS_HIP_CHECK(launch_encode(kind, data_dev, N, L, motifs_codes_pitch(L), codes_dev, bad_flag_dev, c->stream));
    return MOTIFS_OK;
}

int motifs_pwm_scan_dense_dev(motifs_ctx* c, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen,
                              const uint8_t* codes_dev, int64_t N, int L, uint16_t* scores_dev, int64_t ld_l) {
    if (!c || N < 0 || L <= 0 || (N > 0 && (!codes_dev || !scores_dev))) {
        set_error("motifs_pwm_scan_dense_dev: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    const int lenp0 = bank_lenp(lens, K);
    if ((!c->scan_valu || lenp0 > 32) && K % 8 == 0) {
        // Matrix-core path: candidates (scan_cand_kernel) -> exact scores of the hits streamed into the tensor
        // together with the zeros around them (stage_hits, mode 2): every byte written once, in linear order.
        BankSlot* bs = nullptr;
        const int rcode = cached_bank(c, pwms_fp16, lens, K, maxlen, 0, L, &bs);
        if (rcode) return rcode;
        const int Lout = L - bs->minlen + 1;
        if (ld_l < std::max(Lout, 0)) {
            set_error("motifs_pwm_scan_dense_dev: ld_l=%lld < L-minlen+1=%d", (long long)ld_l, Lout);
            return MOTIFS_ERR_INVALID;
        }
        if (N == 0) return MOTIFS_OK;
        const size_t cells_bytes = (size_t)std::max(Lout, 0) * N * bs->nch * 16;
        if (cells_bytes <= ((size_t)16 << 30) && N < ((int64_t)1 << 31) && (int64_t)Lout * ((N + dense_row_reads(bs->nch) - 1) / dense_row_reads(bs->nch)) < ((int64_t)1 << 31)) {
            MOTIFS_HIP_CHECK(hipSetDevice(c->device));
            const int lo = std::max(Lout, 0);
            KernelTimer t(c, KS_SCAN_DENSE);
            if (ld_l > lo)  // the l-planes no window reaches (the reference pre-zeroes the tensor, :75)
                MOTIFS_HIP_CHECK(hipMemsetAsync(scores_dev + (size_t)K * N * lo, 0, (size_t)K * N * (ld_l - lo) * 2, c->stream));
            if (Lout <= 0) return MOTIFS_OK;
            if (c->dense_fused) {          // filter, exact re-scoring and the write stream in one kernel, when the bank fits its LDS plan
                DenseFusedArgs d{};
                d.afrag = (const uint4*)bs->afrag.p;
                d.tabk = (const uint16_t*)bs->tabk.p;
                d.lim = (const int32_t*)bs->lim.p;
                d.codes = codes_dev;
                d.out = scores_dev;
                d.N = N;
                d.L = L;
                d.pitch = motifs_codes_pitch(L);
                d.Lout = Lout;
                d.K = K;
                d.lim_min = L - bs->maxlen_true;
                d.ntiles = (K + 31) / 32;
                d.tabk_stride = bs->tabk_stride;
                if (dense_fused_plan(d, bs->lenp, bs->uniform_eps)) {
                    MOTIFS_HIP_CHECK(launch_dense_fused(d, bs->lenp, c->stream));
                    return MOTIFS_OK;
                }
            }
            MOTIFS_HIP_CHECK(c->cnt.reserve(cells_bytes));
            CandArgs a{};
            FillArgs f{};
            scan_args(c, *bs, K, codes_dev, N, L, Lout, (int)N, dense_row_reads(bs->nch), a, f);   // one "batch": cells in (l, n, chunk) order
            f.dense = scores_dev;
            MOTIFS_HIP_CHECK(launch_cand(a, c->stream));
            MOTIFS_HIP_CHECK(launch_stage_hits(f, 2, c->stream));
            return MOTIFS_OK;
        }
    }
    if (lenp0 > 32) {
        // A long bank whose K is not a multiple of 8 (the streamed form needs 16-byte runs): zeros, then the hit
        // records dropped into place.  Correct for any shape; not the fast path.
        BankSlot* bs = nullptr;
        int rcode = cached_bank(c, pwms_fp16, lens, K, maxlen, 0, L, &bs);
        if (rcode) return rcode;
        const int Lout = L - bs->minlen + 1;
        if (ld_l < std::max(Lout, 0)) {
            set_error("motifs_pwm_scan_dense_dev: ld_l=%lld < L-minlen+1=%d", (long long)ld_l, Lout);
            return MOTIFS_ERR_INVALID;
        }
        if (N == 0) return MOTIFS_OK;
        MOTIFS_HIP_CHECK(hipSetDevice(c->device));
        KernelTimer t(c, KS_SCAN_DENSE);
        MOTIFS_HIP_CHECK(hipMemsetAsync(scores_dev, 0, (size_t)K * N * ld_l * 2, c->stream));
        if (Lout <= 0) return MOTIFS_OK;
        int64_t need = 0;
        rcode = scan_hits_mfma(c, *bs, K, codes_dev, N, L, Lout, 0, (int)std::min<int64_t>(N, MOTIFS_SCAN_BATCH), nullptr, nullptr, 0, &need, nullptr);
        if (rcode) return rcode;
        if (need == 0) return MOTIFS_OK;
        MOTIFS_HIP_CHECK(c->hits_tmp.reserve((size_t)need * sizeof(motifs_hit)));
        MOTIFS_HIP_CHECK(c->scores_tmp.reserve((size_t)need * 2));
        rcode = scan_hits_mfma(c, *bs, K, codes_dev, N, L, Lout, 0, (int)std::min<int64_t>(N, MOTIFS_SCAN_BATCH), (motifs_hit*)c->hits_tmp.p,
                               (uint16_t*)c->scores_tmp.p, need, &need, nullptr);
        if (rcode) return rcode;
        hipLaunchKernelGGL(scatter_hits_dense, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, c->stream, (const HitRec*)c->hits_tmp.p,
                           (const uint16_t*)c->scores_tmp.p, need, K, N, (int64_t)0, scores_dev);
        MOTIFS_HIP_CHECK(hipGetLastError());
        return MOTIFS_OK;
    }
    PackedBank bank;
    int rcode = pack_bank(pwms_fp16, lens, K, maxlen, 0, L, bank);
    if (rcode) return rcode;
    const int Lout = L - bank.minlen + 1;
    if (ld_l < std::max(Lout, 0)) {
        set_error("motifs_pwm_scan_dense_dev: ld_l=%lld < L-minlen+1=%d", (long long)ld_l, Lout);
        return MOTIFS_ERR_INVALID;
    }
    if (N == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    rcode = upload_bank(c, bank);
    if (rcode) return rcode;
    const int lo = std::max(Lout, 0);
    if (ld_l > lo)  // the l-planes no window reaches (the reference pre-zeroes the tensor, :75)
        MOTIFS_HIP_CHECK(hipMemsetAsync(scores_dev + (size_t)K * N * lo, 0, (size_t)K * N * (ld_l - lo) * 2, c->stream));
    if (Lout <= 0) return MOTIFS_OK;
    ScanArgs a{};
    a.tab = (const uint32_t*)c->tab.p;
    a.lim = (const int32_t*)c->lim.p;
    a.codes = codes_dev;
    a.d.N = N;
    a.d.L = L;
    a.d.pitch = motifs_codes_pitch(L);
    a.d.K = K;
    a.d.KP = bank.KP;
    a.d.nch = bank.nch;
    a.d.Lout = Lout;
    a.d.LoutP = scan_lout_padded(Lout, bank.lenp);
    a.d.cpb = pick_cpb(bank.nch);
    a.d.lim_min = L - bank.maxlen_true;
    a.d.spw = pick_spw(N, bank.nch);
    a.d.k_even = (K % 2 == 0);
    a.d.batch = MOTIFS_SCAN_BATCH;
    a.scores = scores_dev;
    KernelTimer t(c, KS_SCAN_DENSE);
    MOTIFS_HIP_CHECK(launch_scan(MODE_DENSE, bank.lenp, a, c->stream));
    return MOTIFS_OK;
}

int motifs_pwm_scan_hits_dev(motifs_ctx* c, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen,
                             const uint8_t* codes_dev, int64_t N, int L, int rc, int64_t n0, int batch,
                             motifs_hit* hits_dev, uint16_t* hit_scores_dev, int64_t cap, int64_t* n_out,
                             int64_t* per_pwm_counts_dev) {
    if (!c || !n_out || N < 0 || L <= 0 || batch <= 0 || cap < 0 || (N > 0 && !codes_dev) ||
        (cap > 0 && (!hits_dev || !hit_scores_dev)) || n0 < 0 || n0 + N > 0xffffffffll) {
        set_error("motifs_pwm_scan_hits_dev: bad argument (N=%lld L=%d batch=%d cap=%lld n0=%lld)", (long long)N, L,
                  batch, (long long)cap, (long long)n0);
        return MOTIFS_ERR_INVALID;
    }
    *n_out = 0;
    if (!c->scan_valu || bank_lenp(lens, K) > 32) {
        BankSlot* bs = nullptr;
        const int rcode = cached_bank(c, pwms_fp16, lens, K, maxlen, rc, L, &bs);
        if (rcode) return rcode;
        MOTIFS_HIP_CHECK(hipSetDevice(c->device));
        if (per_pwm_counts_dev) MOTIFS_HIP_CHECK(hipMemsetAsync(per_pwm_counts_dev, 0, (size_t)K * 8, c->stream));
        const int Lout = L - bs->minlen + 1;
        if (N == 0 || Lout <= 0) return MOTIFS_OK;
        return scan_hits_mfma(c, *bs, K, codes_dev, N, L, Lout, n0, batch, hits_dev, hit_scores_dev, cap, n_out, per_pwm_counts_dev);
    }
    PackedBank bank;
    int rcode = pack_bank(pwms_fp16, lens, K, maxlen, rc, L, bank);
    if (rcode) return rcode;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    if (per_pwm_counts_dev) MOTIFS_HIP_CHECK(hipMemsetAsync(per_pwm_counts_dev, 0, (size_t)K * 8, c->stream));
    const int Lout = L - bank.minlen + 1;
    if (N == 0 || Lout <= 0) return MOTIFS_OK;
    rcode = upload_bank(c, bank);
    if (rcode) return rcode;

    const int LoutP = scan_lout_padded(Lout, bank.lenp);
    // super-batch: as many ordering batches as fit an ~8 GiB mask workspace
    const size_t per_batch = (size_t)LoutP * batch * bank.nch * 16;
    int64_t nb_max = (int64_t)(c->ws_limit / per_batch);
    nb_max = std::max<int64_t>(1, std::min<int64_t>(nb_max, (N + batch - 1) / batch));
    const int64_t sb = nb_max * batch;
    const int64_t cells_max = (int64_t)nb_max * LoutP * batch * bank.nch;
    const int64_t chunks_max = (int64_t)nb_max * LoutP;   // mask rows

    MOTIFS_HIP_CHECK(c->cnt.reserve((size_t)cells_max * 16));        // masks
    MOTIFS_HIP_CHECK(c->tilesum.reserve((size_t)chunks_max * 4));    // chunk sums
    MOTIFS_HIP_CHECK(c->off.reserve((size_t)chunks_max * 8));        // chunk bases
    MOTIFS_HIP_CHECK(c->small.reserve(64));
    MOTIFS_HIP_CHECK(c->pwmcnt.reserve((size_t)2 * bank.KP * 8));
    if (per_pwm_counts_dev) MOTIFS_HIP_CHECK(hipMemsetAsync(c->pwmcnt.p, 0, (size_t)2 * bank.KP * 8, c->stream));

    int64_t* total_dev = (int64_t*)c->small.p;
    int64_t* h_total = (int64_t*)c->pinned;

    int64_t emitted = 0;
    bool too_small = false;
    for (int64_t s0 = 0; s0 < N; s0 += sb) {
        const int64_t ns = std::min<int64_t>(sb, N - s0);
        const int64_t nb = (ns + batch - 1) / batch;
        ScanArgs a{};
        a.tab = (const uint32_t*)c->tab.p;
        a.lim = (const int32_t*)c->lim.p;
        a.codes = codes_dev + (size_t)s0 * motifs_codes_pitch(L);
        a.masks = (uint4*)c->cnt.p;
        a.d.N = ns;
        a.d.L = L;
        a.d.pitch = motifs_codes_pitch(L);
        a.d.K = K;
        a.d.KP = bank.KP;
        a.d.nch = bank.nch;
        a.d.Lout = Lout;
        a.d.LoutP = LoutP;
        a.d.cpb = pick_cpb(bank.nch);
        a.d.lim_min = L - bank.maxlen_true;
        a.d.spw = pick_spw(ns, bank.nch);
        a.d.k_even = (K % 2 == 0);
        a.d.batch = batch;
        FillArgs f{};
        f.masks = a.masks;
        f.nrows = nb * LoutP;
        f.row_cells = (uint32_t)(batch * bank.nch);
        f.row_sum = (uint32_t*)c->tilesum.p;
        f.row_base = (int64_t*)c->off.p;
        f.Lout = Lout;
        f.hist_bins = (per_pwm_counts_dev && 2 * bank.KP <= FILL_HIST_MAX) ? 2 * bank.KP : 0;
        {   // magic numbers for idx / nch
            uint32_t d = (uint32_t)bank.nch, sh = 0;
            while ((1u << sh) < d) sh++;
            f.div_nch.d = d;
            f.div_nch.s = sh;
            f.div_nch.m = d == 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
        }
        f.total = total_dev;
        f.tab = a.tab;
        f.codes = a.codes;
        f.hits = (HitRec*)hits_dev;
        f.hit_scores = hit_scores_dev;
        f.pwm_counts = per_pwm_counts_dev ? (int64_t*)c->pwmcnt.p : nullptr;
        f.base0 = emitted;
        f.n0 = n0 + s0;
        f.nch = bank.nch;
        f.batch = batch;
        f.LoutP = LoutP;
        f.lshift = bank.lenp - 1;
        f.lenp = bank.lenp;
        f.KP = bank.KP;
        f.pitch = a.d.pitch;
        if (ns < nb * batch)  // sequences the last (partial) batch does not have: their cells must read as empty
            MOTIFS_HIP_CHECK(hipMemsetAsync(c->cnt.p, 0, (size_t)f.nrows * f.row_cells * 16, c->stream));
        {
            KernelTimer t(c, KS_SCAN_COUNT);
            MOTIFS_HIP_CHECK(launch_scan(MODE_MASK, bank.lenp, a, c->stream));
        }
        {
            KernelTimer t(c, KS_SCAN_OFFSETS);
            MOTIFS_HIP_CHECK(launch_fill_sums(f, c->stream));
        }
        MOTIFS_HIP_CHECK(hipMemcpyAsync(h_total, total_dev, 8, hipMemcpyDeviceToHost, c->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        const int64_t sb_total = *h_total;
        if (emitted + sb_total > cap) too_small = true;
        if (!too_small && sb_total > 0) {
            KernelTimer t(c, KS_SCAN_FILL);
            MOTIFS_HIP_CHECK(launch_fill_records(f, c->stream));
        }
        if (f.pwm_counts && (too_small || 2 * bank.KP > FILL_HIST_MAX))
            MOTIFS_HIP_CHECK(launch_mask_histogram(f, c->stream));
        emitted += sb_total;
    }
    *n_out = emitted;
    if (per_pwm_counts_dev)
        MOTIFS_HIP_CHECK(hipMemcpyAsync(per_pwm_counts_dev, c->pwmcnt.p, (size_t)K * 8, hipMemcpyDeviceToDevice, c->stream));
    if (too_small && !(cap == 0 && hits_dev == nullptr)) {
        set_error("hit buffer too small: need %lld records, cap %lld", (long long)emitted, (long long)cap);
        return MOTIFS_ERR_BUFFER_TOO_SMALL;
    }
    return MOTIFS_OK;
}

// gpu_scan (_h3_1_alignment.jl:89-99): both strands of one shard in one call - the reverse-strand kernels are enqueued
// behind the forward ones and the host waits once.
int motifs_pwm_scan_hits_both_dev(motifs_ctx* c, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen,
                                  const uint8_t* codes_dev, int64_t N, int L, int64_t n0, int batch, motifs_hit* hits_fwd_dev,
                                  uint16_t* scores_fwd_dev, motifs_hit* hits_rc_dev, uint16_t* scores_rc_dev, int64_t cap,
                                  int64_t* n_out2, int64_t* per_pwm_counts2_dev) {
    if (!c || !n_out2) {
        set_error("motifs_pwm_scan_hits_both_dev: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    motifs_hit* hits[2] = {hits_fwd_dev, hits_rc_dev};
    uint16_t* scores[2] = {scores_fwd_dev, scores_rc_dev};
    n_out2[0] = n_out2[1] = 0;
    const bool fast = (!c->scan_valu || bank_lenp(lens, K) > 32) && N > 0 && L > 0 && batch > 0 && cap >= 0 && codes_dev && n0 >= 0 && n0 + N <= 0xffffffffll &&
                      (cap == 0 || (hits[0] && hits[1] && scores[0] && scores[1]));
    BankSlot* bs[2] = {nullptr, nullptr};
    if (fast) {
        for (int rc = 0; rc < 2; rc++) {
            const int rcode = cached_bank(c, pwms_fp16, lens, K, maxlen, rc, L, &bs[rc]);
            if (rcode) return rcode;
        }
    }
    if (!fast || L - bs[0]->minlen + 1 <= 0) {     // anything unusual: the single-strand entry, twice
        for (int rc = 0; rc < 2; rc++) {
            const int rcode = motifs_pwm_scan_hits_dev(c, pwms_fp16, lens, K, maxlen, codes_dev, N, L, rc, n0, batch, hits[rc], scores[rc], cap,
                                                       &n_out2[rc], per_pwm_counts2_dev ? per_pwm_counts2_dev + (size_t)rc * K : nullptr);
            if (rcode) return rcode;
        }
        return MOTIFS_OK;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    c->ev_totals_set = false;
    c->ticket_wait = false;
    // One candidate launch for both strands when the four-reads kernel with compact entries serves both banks and the shard is one
    // super-batch: the reverse bank then goes over the reads the forward bank's waves have already staged (scan_mfma.hip).
    bool fuse = c->compact_cells && c->fuse_strands;
    bool pair_ok = true, pair_fits = false;
    {
        const int Lout0 = L - bs[0]->minlen + 1;
        CandArgs a0{}, a1{};
        FillArgs f0{};
        const int rpr0 = stage_row_reads(bs[0]->nch);
        scan_args(c, *bs[0], K, codes_dev, N, L, Lout0, batch, rpr0, a0, f0);
        scan_args(c, *bs[1], K, codes_dev, N, L, Lout0, batch, rpr0, a1, f0);
        const bool emit = hits[0] != nullptr && cap > 0;
        const HitGeom hg0 = hit_geom(c, *bs[0], K, Lout0, batch, emit, N, 1);
        pair_fits = hit_geom(c, *bs[0], K, Lout0, batch, emit, N, 2).nb_max * batch >= N;   // the pair launches hold both strands' staged words
        fuse = fuse && cand_two_strands_ok(a0) && cand_two_strands_ok(a1) && bs[0]->lenp == bs[1]->lenp && bs[0]->nch == bs[1]->nch &&
               hg0.compact && hg0.nb_max * batch >= N;
        pair_ok = hg0.cgc == 0;           // (chunk groups keep their launches per strand)
    }
    const bool pair_plan = fuse && pair_ok && pair_fits && c->pair_launches && bs[0]->minlen == bs[1]->minlen;
    // the counts start at zero: the pair plan's candidate kernel does it itself (its first block: no fill in front of the step)
    if (per_pwm_counts2_dev && !pair_plan) MOTIFS_HIP_CHECK(hipMemsetAsync(per_pwm_counts2_dev, 0, (size_t)2 * K * 8, c->stream));
    if (pair_plan) {
        const int rcode = scan_hits_pair(c, bs, K, codes_dev, N, L, L - bs[0]->minlen + 1, n0, batch, hits, scores, cap, per_pwm_counts2_dev);
        if (rcode) return rcode;
    } else
    for (int rc = 0; rc < 2; rc++) {
        const int Lout = L - bs[rc]->minlen + 1;
        int64_t dummy = 0;
        const int rcode = scan_hits_mfma(c, *bs[rc], K, codes_dev, N, L, Lout, n0, batch, hits[rc], scores[rc], cap, &dummy,
                                         per_pwm_counts2_dev ? per_pwm_counts2_dev + (size_t)rc * K : nullptr, rc, false,
                                         fuse ? rc + 1 : 0, fuse ? bs[1] : nullptr);
        if (rcode) return rcode;
    }
    if (c->ticket_wait) {          // records in stream order: wait for the row scans only - they write this call's ticket into pinned memory behind
        c->ticket_wait = false;    // the totals; polling it costs a core for the length of the scan and spares the stream an event (and the host its wake-up)
        const volatile int64_t* tk = (const volatile int64_t*)c->pinned + 8;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = false;
        for (uint64_t spin = 0;; spin++) {
            if (__atomic_load_n(&tk[0], __ATOMIC_ACQUIRE) == c->ticket_seq && __atomic_load_n(&tk[1], __ATOMIC_ACQUIRE) == c->ticket_seq) {
                ok = true;
                break;
            }
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;   // (a failed launch: fall through to the stream wait and its error)
        }
        if (!ok) MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    } else if (c->ev_totals_set) {
        c->ev_totals_set = false;
        MOTIFS_HIP_CHECK(hipEventSynchronize(c->ev_totals));
    } else {
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    const int64_t* h_total = (const int64_t*)c->pinned;
    for (int rc = 0; rc < 2; rc++) {
        n_out2[rc] = h_total[rc];
        if (h_total[rc] > cap && !(cap == 0 && hits[rc] == nullptr)) {
            // records in stream order: the kernel that writes the first `cap` records may still be running, and a caller that gets this
            // error frees or re-allocates its buffers next - the error path waits for it
            if (c->records_async) MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
            set_error("hit buffer too small: strand %d needs %lld records, cap %lld", rc, (long long)h_total[rc], (long long)cap);
            return MOTIFS_ERR_BUFFER_TOO_SMALL;
        }
    }
    return MOTIFS_OK;
}

}  // extern "C"

// ---- host-buffer entries: what crosses PCIe -------------------------------------------------------------------------
// The reference's data matrix is Float32 one-hot, 16 bytes per base; the device wants 1 byte per base.  Host threads turn
// the one-hot columns into code rows in a pinned buffer (the same bytes encode_f32 / encode_f16 would produce, the same
// one-hot check), so 1/16 of the matrix crosses the bus; the records come back through pinned chunks that host threads
// copy out while the next chunk is on the wire (a pageable hipMemcpy stages every byte through the runtime's own bounce
// buffer on one thread).
static hipError_t pin_reserve(motifs_ctx* c, size_t bytes) {
    if (bytes <= c->pin_stage_cap) return hipSuccess;
    if (c->pin_stage) (void)hipHostFree(c->pin_stage);
    c->pin_stage = nullptr;
    c->pin_stage_cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipHostMalloc(&c->pin_stage, want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    c->pin_stage_cap = want;
    return hipSuccess;
}
static int host_threads(int64_t work_items, int64_t per_thread) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t want = std::max<int64_t>(1, work_items / std::max<int64_t>(per_thread, 1));
    return (int)std::min<int64_t>(std::min<int64_t>(hw ? hw : 1, 32), want);
}
template <typename F>
static void run_threads(int T, F&& fn) {
    if (T <= 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (int t = 1; t < T; t++) th.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& x : th) x.join();
}

// one-hot rows [n0, n1) -> code rows (pitch bytes each: codes, zero padding, the row flag in byte pitch - 4); returns
// whether a column was neither one-hot nor all-zero
template <typename T4, typename Dec>
static bool encode_rows_host(const T4* x, int64_t n0, int64_t n1, int L, int pitch, uint8_t* rows, Dec&& decode) {
    bool bad = false;
    for (int64_t n = n0; n < n1; n++) {
        uint8_t* row = rows + (size_t)n * pitch;
        const T4* src = x + (size_t)n * L;
        uint8_t flag = 0;
        for (int p = 0; p < L; p++) {
            const int code = decode(src[p], bad);
            row[p] = (uint8_t)code;
            flag |= (uint8_t)(code == 4);
        }
        memset(row + L, 0, (size_t)(pitch - L));
        row[pitch - 4] = flag;
    }
    return bad;
}
struct F32x4 {
    float v[4];
};
struct F16x4 {
    uint16_t v[4];
};

// host matrix of `kind` -> c->codes (the internal code matrix), checked for one-hot columns
int motifs::upload_and_encode(motifs_ctx* c, const void* data, int kind, int64_t N, int L) {
    MOTIFS_HIP_CHECK(c->codes.reserve(motifs_codes_bytes(N, L)));
    MOTIFS_HIP_CHECK(c->small.reserve(4096));
    const int pitch = motifs_codes_pitch(L);
    // Rows are encoded into a ring of pinned chunks (a few tens of MiB each, whatever N is) by host threads while the
    // previous chunk is on the wire; the page-locked footprint no longer grows with the input.
    constexpr size_t CH_BYTES = (size_t)32 << 20;
    constexpr int NB = 3;
    const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(CH_BYTES / (size_t)pitch));
    const size_t slot_bytes = (size_t)std::min<int64_t>(rows_per_chunk, std::max<int64_t>(N, 1)) * pitch;
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));        // an earlier call may still read the staging block
    MOTIFS_HIP_CHECK(pin_reserve(c, slot_bytes * NB));
    hipEvent_t ev[NB] = {nullptr, nullptr, nullptr};
    for (auto& e : ev) MOTIFS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::atomic<int> bad{0};
    hipError_t err = hipSuccess;
    int64_t chunk_no = 0;
    for (int64_t r0 = 0; r0 < N && err == hipSuccess && !bad.load(); r0 += rows_per_chunk, chunk_no++) {
        const int64_t nr = std::min<int64_t>(rows_per_chunk, N - r0);
        const int slot = (int)(chunk_no % NB);
        if (chunk_no >= NB) err = hipEventSynchronize(ev[slot]);          // the slot's previous upload has left it
        if (err != hipSuccess) break;
        uint8_t* rows = (uint8_t*)c->pin_stage + (size_t)slot * slot_bytes;   // row r0 + i of the matrix at rows + i * pitch
        const int T = host_threads(nr * (int64_t)L, 1 << 18);
        run_threads(T, [&](int t) {
            const int64_t n0 = nr * t / T, n1 = nr * (t + 1) / T;
            bool b;
            if (kind == MOTIFS_DATA_CODES_U8)   // 1 byte per base already: the rows are laid out (padding, row flag) on the way into the ring
                b = encode_rows_host((const uint8_t*)data + (size_t)r0 * L, n0, n1, L, pitch, rows, [](const uint8_t& q, bool& bad_) {
                    if (q > 4) bad_ = true;
                    return q > 4 ? 4 : (int)q;
                });
            else if (kind == MOTIFS_DATA_ONEHOT_F32)
                b = encode_rows_host((const F32x4*)data + (size_t)r0 * L, n0, n1, L, pitch, rows, [](const F32x4& q, bool& bad_) {
                    const int ones = (q.v[0] == 1.0f) + (q.v[1] == 1.0f) + (q.v[2] == 1.0f) + (q.v[3] == 1.0f);
                    const int zeros = (q.v[0] == 0.0f) + (q.v[1] == 0.0f) + (q.v[2] == 0.0f) + (q.v[3] == 0.0f);
                    if (ones == 1 && zeros == 3) return q.v[0] == 1.0f ? 0 : q.v[1] == 1.0f ? 1 : q.v[2] == 1.0f ? 2 : 3;
                    if (zeros != 4) bad_ = true;
                    return 4;
                });
            else
                b = encode_rows_host((const F16x4*)data + (size_t)r0 * L, n0, n1, L, pitch, rows, [](const F16x4& q, bool& bad_) {
                    int ones = 0, zeros = 0, which = 0;
                    for (int u = 0; u < 4; u++) {
                        if (q.v[u] == 0x3c00u) ones++, which = u;
                        if ((q.v[u] & 0x7fffu) == 0) zeros++;
                    }
                    if (ones == 1 && zeros == 3) return which;
                    if (zeros != 4) bad_ = true;
                    return 4;
                });
            if (b) bad.store(1);
        });
        err = hipMemcpyAsync((uint8_t*)c->codes.p + (size_t)r0 * pitch, rows, (size_t)nr * pitch, hipMemcpyHostToDevice, c->stream);
        if (err == hipSuccess) err = hipEventRecord(ev[slot], c->stream);
    }
    if (err == hipSuccess)                                       // the guard bytes behind the last row
        err = hipMemsetAsync((uint8_t*)c->codes.p + (size_t)N * pitch, 0, motifs_codes_bytes(N, L) - (size_t)N * pitch, c->stream);
    const hipError_t serr = hipStreamSynchronize(c->stream);     // the staging block is reused by the download
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (bad.load()) {
        set_error(kind == MOTIFS_DATA_CODES_U8 ? "data matrix has a code outside 0..4" : "data matrix has a column that is neither one-hot nor all-zero");
        return MOTIFS_ERR_NOT_ONEHOT;
    }
    MOTIFS_HIP_CHECK(err);
    MOTIFS_HIP_CHECK(serr);
    return MOTIFS_OK;
}

// device bytes -> pageable host memory: chunks cross the bus into a ring of pinned buffers; host threads copy chunk i out
// while chunk i + 1 is in flight.  The stream is idle when this returns.
int motifs::download_chunked(motifs_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    if (bytes == 0) return MOTIFS_OK;
    constexpr size_t CH = (size_t)16 << 20;
    constexpr int NB = 4;
    if (bytes <= (size_t)4 << 20) {
        MOTIFS_HIP_CHECK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return MOTIFS_OK;
    }
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    MOTIFS_HIP_CHECK(pin_reserve(c, CH * NB));
    const int64_t nchunk = (int64_t)((bytes + CH - 1) / CH);
    const int T = std::max(1, std::min(8, host_threads((int64_t)bytes, 8 << 20)));
    std::vector<hipEvent_t> ev(NB, nullptr);
    for (auto& e : ev) MOTIFS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::atomic<int64_t> landed{0};                 // chunks whose DMA has completed
    std::vector<std::atomic<int>> copied(nchunk);   // copier threads done with chunk i
    for (auto& x : copied) x.store(0);
    std::atomic<int> failed{0};
    auto copier = [&](int t) {
        for (int64_t i = 0; i < nchunk; i++) {
            while (landed.load(std::memory_order_acquire) <= i) {
                if (failed.load()) return;
                std::this_thread::yield();
            }
            const size_t off = (size_t)i * CH, len = std::min(CH, bytes - off);
            const size_t a = len * t / T, b = len * (t + 1) / T;
            memcpy((char*)dst + off + a, (const char*)c->pin_stage + (size_t)(i % NB) * CH + a, b - a);
            copied[i].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(copier, t);
    hipError_t err = hipSuccess;
    for (int64_t i = 0; i < nchunk + 1 && err == hipSuccess; i++) {
        if (i < nchunk) {
            if (i >= NB)                              // the ring slot must have been copied out
                while (copied[i - NB].load(std::memory_order_acquire) < T) std::this_thread::yield();
            const size_t off = (size_t)i * CH, len = std::min(CH, bytes - off);
            err = hipMemcpyAsync((char*)c->pin_stage + (size_t)(i % NB) * CH, (const char*)src_dev + off, len, hipMemcpyDeviceToHost, c->stream);
            if (err == hipSuccess) err = hipEventRecord(ev[i % NB], c->stream);
        }
        if (i >= 1 && err == hipSuccess) {            // chunk i - 1 has landed once its event fires
            err = hipEventSynchronize(ev[(i - 1) % NB]);
            if (err == hipSuccess) landed.store(i, std::memory_order_release);
        }
    }
    if (err != hipSuccess) failed.store(1);
    for (auto& x : th) x.join();
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (err != hipSuccess) {
        set_error("record download failed: %s", hipGetErrorString(err));
        return MOTIFS_ERR_HIP;
    }
    return MOTIFS_OK;
}

extern "C" {

int motifs_pwm_scan(motifs_ctx* c, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen, const void* data,
                    int kind, int64_t N, int L, int rc, motifs_hit* hits, uint16_t* hit_scores, int64_t cap,
                    int64_t* n_out, int64_t* per_pwm_counts) {
    if (!c || !n_out || N < 0 || L <= 0 || kind < 0 || kind > 2 || (N > 0 && !data) || cap < 0 ||
        (cap > 0 && (!hits || !hit_scores))) {
        set_error("motifs_pwm_scan: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    int r = upload_and_encode(c, data, kind, N, L);
    if (r) return r;
    if (cap > 0) {
        MOTIFS_HIP_CHECK(c->hits_tmp.reserve((size_t)cap * sizeof(motifs_hit)));
        MOTIFS_HIP_CHECK(c->scores_tmp.reserve((size_t)cap * 2));
    }
    int64_t* counts_dev = nullptr;
    if (per_pwm_counts) {
        MOTIFS_HIP_CHECK(c->data_tmp.reserve((size_t)K * 8));
        counts_dev = (int64_t*)c->data_tmp.p;
    }
    r = motifs_pwm_scan_hits_dev(c, pwms_fp16, lens, K, maxlen, (const uint8_t*)c->codes.p, N, L, rc, 0,
                                 MOTIFS_SCAN_BATCH, cap > 0 ? (motifs_hit*)c->hits_tmp.p : nullptr,
                                 cap > 0 ? (uint16_t*)c->scores_tmp.p : nullptr, cap, n_out, counts_dev);
    if (per_pwm_counts && (r == MOTIFS_OK || r == MOTIFS_ERR_BUFFER_TOO_SMALL))
        MOTIFS_HIP_CHECK(hipMemcpyAsync(per_pwm_counts, counts_dev, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    if (r == MOTIFS_OK && cap > 0 && *n_out > 0) {
        int d = download_chunked(c, hits, c->hits_tmp.p, (size_t)*n_out * sizeof(motifs_hit));
        if (d == MOTIFS_OK) d = download_chunked(c, hit_scores, c->scores_tmp.p, (size_t)*n_out * 2);
        if (d) return d;
    }
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return r;
}

// gpu_scan (_h3_1_alignment.jl:89-99) on host buffers: the matrix is uploaded and encoded once, both strands are scanned
// by one device call, the two record lists come back.
int motifs_pwm_scan_both(motifs_ctx* c, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen, const void* data, int kind,
                         int64_t N, int L, motifs_hit* hits_fwd, uint16_t* scores_fwd, motifs_hit* hits_rc, uint16_t* scores_rc, int64_t cap,
                         int64_t* n_out2, int64_t* per_pwm_counts2) {
    if (!c || !n_out2 || N < 0 || L <= 0 || kind < 0 || kind > 2 || (N > 0 && !data) || cap < 0 ||
        (cap > 0 && (!hits_fwd || !scores_fwd || !hits_rc || !scores_rc))) {
        set_error("motifs_pwm_scan_both: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    int r = upload_and_encode(c, data, kind, N, L);
    if (r) return r;
    motifs_hit* hd[2] = {nullptr, nullptr};
    uint16_t* sd[2] = {nullptr, nullptr};
    if (cap > 0) {
        const int64_t cap8 = (cap + 7) & ~(int64_t)7;      // the reverse strand's scores start 16-byte aligned
        MOTIFS_HIP_CHECK(c->hits_tmp.reserve((size_t)2 * cap * sizeof(motifs_hit)));
        MOTIFS_HIP_CHECK(c->scores_tmp.reserve((size_t)(cap8 + cap) * 2));
        hd[0] = (motifs_hit*)c->hits_tmp.p;
        hd[1] = hd[0] + cap;
        sd[0] = (uint16_t*)c->scores_tmp.p;
        sd[1] = sd[0] + cap8;
    }
    int64_t* counts_dev = nullptr;
    if (per_pwm_counts2) {
        MOTIFS_HIP_CHECK(c->data_tmp.reserve((size_t)2 * K * 8));
        counts_dev = (int64_t*)c->data_tmp.p;
    }
    r = motifs_pwm_scan_hits_both_dev(c, pwms_fp16, lens, K, maxlen, (const uint8_t*)c->codes.p, N, L, 0, MOTIFS_SCAN_BATCH, hd[0], sd[0], hd[1],
                                      sd[1], cap, n_out2, counts_dev);
    if (per_pwm_counts2 && (r == MOTIFS_OK || r == MOTIFS_ERR_BUFFER_TOO_SMALL))
        MOTIFS_HIP_CHECK(hipMemcpyAsync(per_pwm_counts2, counts_dev, (size_t)2 * K * 8, hipMemcpyDeviceToHost, c->stream));
    if (r == MOTIFS_OK && cap > 0) {
        motifs_hit* hh[2] = {hits_fwd, hits_rc};
        uint16_t* sh[2] = {scores_fwd, scores_rc};
        for (int s = 0; s < 2; s++)
            if (n_out2[s] > 0) {
                int d = download_chunked(c, hh[s], hd[s], (size_t)n_out2[s] * sizeof(motifs_hit));
                if (d == MOTIFS_OK) d = download_chunked(c, sh[s], sd[s], (size_t)n_out2[s] * 2);
                if (d) return d;
            }
    }
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return r;
}

// ---- device memory for a host without a GPU array package (include/motifs_hip.h) ---------------------------------------
int motifs_dev_alloc(motifs_ctx* c, size_t bytes, void** out_dev) {
    if (!c || !out_dev) {
        set_error("motifs_dev_alloc: null argument");
        return MOTIFS_ERR_INVALID;
    }
    *out_dev = nullptr;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipMalloc(out_dev, std::max<size_t>(bytes, 16)));
    return MOTIFS_OK;
}

int motifs_dev_free(motifs_ctx* c, void* ptr_dev) {
    if (!c) return MOTIFS_ERR_INVALID;
    if (!ptr_dev) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));     // nothing queued by the library may still touch the buffer
    MOTIFS_HIP_CHECK(hipFree(ptr_dev));
    return MOTIFS_OK;
}

int motifs_dev_upload(motifs_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
    if (!c || (bytes > 0 && (!dst_dev || !src_host))) {
        set_error("motifs_dev_upload: null argument");
        return MOTIFS_ERR_INVALID;
    }
    if (bytes == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MOTIFS_OK;
}

int motifs_dev_download(motifs_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
    if (!c || (bytes > 0 && (!dst_host || !src_dev))) {
        set_error("motifs_dev_download: null argument");
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    return download_chunked(c, dst_host, src_dev, bytes);    // pinned ring for large blocks; the stream is idle on return
}

int motifs_dev_memset(motifs_ctx* c, void* dst_dev, int byte_value, size_t bytes) {
    if (!c || (bytes > 0 && !dst_dev)) {
        set_error("motifs_dev_memset: null argument");
        return MOTIFS_ERR_INVALID;
    }
    if (bytes == 0) return MOTIFS_OK;
    MOTIFS_HIP_CHECK(hipSetDevice(c->device));
    MOTIFS_HIP_CHECK(hipMemsetAsync(dst_dev, byte_value, bytes, c->stream));
    return MOTIFS_OK;
}

// gpu_scan of a host matrix over several devices of one process: one host thread per device (upload + encode, count,
// fill, download into the device's slice of the caller's buffers); records concatenated in device order.
int motifs_pwm_scan_both_sharded(motifs_ctx* const* ctxs, motifs_comm* const* comms, int n_dev, const uint16_t* pwms_fp16,
                                 const int64_t* lens, int K, int maxlen, const void* data, int kind, int64_t N, int L,
                                 int64_t shard_align, motifs_hit* hits_fwd, uint16_t* scores_fwd, motifs_hit* hits_rc, uint16_t* scores_rc,
                                 int64_t cap, int64_t* n_out2, int64_t* per_pwm_counts2, int64_t* shard_counts) {
    if (!ctxs || n_dev < 1 || !n_out2 || N < 0 || L <= 0 || K <= 0 || kind < 0 || kind > 2 || (N > 0 && !data) || cap < 0 || shard_align < 1 ||
        (cap > 0 && (!hits_fwd || !scores_fwd || !hits_rc || !scores_rc))) {
        set_error("motifs_pwm_scan_both_sharded: bad argument (n_dev=%d N=%lld L=%d K=%d cap=%lld align=%lld)", n_dev, (long long)N, L, K,
                  (long long)cap, (long long)shard_align);
        return MOTIFS_ERR_INVALID;
    }
    for (int d = 0; d < n_dev; d++) {
        if (!ctxs[d] || (comms && comm_ctx(comms[d]) != ctxs[d])) {
            set_error("motifs_pwm_scan_both_sharded: slot %d: null context, or a communicator made on another context", d);
            return MOTIFS_ERR_INVALID;
        }
        for (int j = 0; j < d; j++)
            if (ctxs[j] == ctxs[d]) {
                set_error("motifs_pwm_scan_both_sharded: slots %d and %d are the same context", j, d);
                return MOTIFS_ERR_INVALID;
            }
    }
    if (comms && comm_group_depth() > 0) {
        set_error("motifs_pwm_scan_both_sharded opens its own group around the histogram sums: call it outside motifs_comm_group_start/_end");
        return MOTIFS_ERR_INVALID;
    }
    n_out2[0] = n_out2[1] = 0;
    // contiguous blocks of whole units of shard_align reads; sizes differ by at most one unit (parallel.shard_range)
    const int64_t units = (N + shard_align - 1) / shard_align;
    std::vector<int64_t> lo(n_dev + 1, 0);
    for (int d = 0; d < n_dev; d++) {
        const int64_t u = units / n_dev + (d < units % n_dev ? 1 : 0);
        lo[d + 1] = std::min<int64_t>(N, lo[d] + u * shard_align);
    }
    const size_t elt = kind == MOTIFS_DATA_ONEHOT_F32 ? 16 : kind == MOTIFS_DATA_ONEHOT_F16 ? 8 : 1;
    std::vector<int64_t> need(2 * (size_t)n_dev, 0);
    auto on_devices = [&](auto&& fn) -> int { return for_each_device(n_dev, fn); };
    // phase A: reads to the device, count-only scan (the records of a shard need their offset in the caller's buffers)
    int r = on_devices([&](int d) -> int {
        motifs_ctx* c = ctxs[d];
        const int64_t n = lo[d + 1] - lo[d];
        MOTIFS_HIP_CHECK(hipSetDevice(c->device));
        MOTIFS_HIP_CHECK(c->pwmcnt.reserve((size_t)2 * K * 8 + 64));
        MOTIFS_HIP_CHECK(hipMemsetAsync(c->pwmcnt.p, 0, (size_t)2 * K * 8, c->stream));
        if (n == 0) return MOTIFS_OK;
        int rr = upload_and_encode(c, (const char*)data + (size_t)lo[d] * L * elt, kind, n, L);
        if (rr) return rr;
        return motifs_pwm_scan_hits_both_dev(c, pwms_fp16, lens, K, maxlen, (const uint8_t*)c->codes.p, n, L, lo[d], MOTIFS_SCAN_BATCH, nullptr, nullptr,
                                             nullptr, nullptr, 0, &need[2 * (size_t)d], nullptr);
    });
    if (r) return r;
    std::vector<int64_t> off(2 * (size_t)(n_dev + 1), 0);     // off[2 d + s]: records of strand s before device d
    for (int d = 0; d < n_dev; d++)
        for (int sdx = 0; sdx < 2; sdx++) off[2 * (size_t)(d + 1) + sdx] = off[2 * (size_t)d + sdx] + need[2 * (size_t)d + sdx];
    n_out2[0] = off[2 * (size_t)n_dev];
    n_out2[1] = off[2 * (size_t)n_dev + 1];
    if (shard_counts) memcpy(shard_counts, need.data(), need.size() * 8);
    const bool too_small = n_out2[0] > cap || n_out2[1] > cap;
    const bool count_only = cap == 0 && !hits_fwd;
    // phase B: the records (and the histogram: a shard's counts are only written by a filling or counting scan; take the filling one)
    if (!too_small || count_only) {
        r = on_devices([&](int d) -> int {
            motifs_ctx* c = ctxs[d];
            const int64_t n = lo[d + 1] - lo[d];
            if (n == 0) return MOTIFS_OK;
            MOTIFS_HIP_CHECK(hipSetDevice(c->device));
            const int64_t cap_d = count_only ? 0 : std::max<int64_t>(std::max(need[2 * (size_t)d], need[2 * (size_t)d + 1]), 1);
            motifs_hit* hd[2] = {nullptr, nullptr};
            uint16_t* sd[2] = {nullptr, nullptr};
            if (!count_only) {
                const int64_t cap8 = (cap_d + 7) & ~(int64_t)7;
                MOTIFS_HIP_CHECK(c->hits_tmp.reserve((size_t)2 * cap_d * sizeof(motifs_hit)));
                MOTIFS_HIP_CHECK(c->scores_tmp.reserve((size_t)(cap8 + cap_d) * 2));
                hd[0] = (motifs_hit*)c->hits_tmp.p;
                hd[1] = hd[0] + cap_d;
                sd[0] = (uint16_t*)c->scores_tmp.p;
                sd[1] = sd[0] + cap8;
            }
            int64_t got[2] = {0, 0};
            int rr = motifs_pwm_scan_hits_both_dev(c, pwms_fp16, lens, K, maxlen, (const uint8_t*)c->codes.p, n, L, lo[d], MOTIFS_SCAN_BATCH, hd[0],
                                                   sd[0], hd[1], sd[1], cap_d, got, (int64_t*)c->pwmcnt.p);
            if (rr) return rr;
            if (count_only) return MOTIFS_OK;
            motifs_hit* hh[2] = {hits_fwd, hits_rc};
            uint16_t* sh[2] = {scores_fwd, scores_rc};
            for (int sdx = 0; sdx < 2; sdx++)
                if (got[sdx] > 0) {
                    rr = download_chunked(c, hh[sdx] + off[2 * (size_t)d + sdx], hd[sdx], (size_t)got[sdx] * sizeof(motifs_hit));
                    if (rr == MOTIFS_OK) rr = download_chunked(c, sh[sdx] + off[2 * (size_t)d + sdx], sd[sdx], (size_t)got[sdx] * 2);
                    if (rr) return rr;
                }
            return MOTIFS_OK;
        });
        if (r) return r;
    } else if (per_pwm_counts2) {
        // too small a buffer: the caller still gets the histogram (a counting scan writes it as well)
        r = on_devices([&](int d) -> int {
            motifs_ctx* c = ctxs[d];
            const int64_t n = lo[d + 1] - lo[d];
            if (n == 0) return MOTIFS_OK;
            MOTIFS_HIP_CHECK(hipSetDevice(c->device));
            int64_t got[2] = {0, 0};
            return motifs_pwm_scan_hits_both_dev(c, pwms_fp16, lens, K, maxlen, (const uint8_t*)c->codes.p, n, L, lo[d], MOTIFS_SCAN_BATCH, nullptr, nullptr,
                                                 nullptr, nullptr, 0, got, (int64_t*)c->pwmcnt.p);
        });
        if (r) return r;
    }
    // phase C: the one exchange of the scan: the 2 x K hit counts
    if (per_pwm_counts2) {
        if (comms && n_dev > 1) {
            r = motifs_comm_group_start();
            if (r) return r;
            int rr = MOTIFS_OK;
            for (int d = 0; d < n_dev && rr == MOTIFS_OK; d++) rr = motifs_hist_allreduce(comms[d], (int64_t*)ctxs[d]->pwmcnt.p, K, 2);
            std::string why = rr ? last_error_text() : "";
            r = motifs_comm_group_end();
            if (rr) {
                set_error("%s", why.c_str());
                return rr;
            }
            if (r) return r;
            MOTIFS_HIP_CHECK(hipSetDevice(ctxs[0]->device));
            MOTIFS_HIP_CHECK(hipMemcpyAsync(per_pwm_counts2, ctxs[0]->pwmcnt.p, (size_t)2 * K * 8, hipMemcpyDeviceToHost, ctxs[0]->stream));
            for (int d = 0; d < n_dev; d++) {
                MOTIFS_HIP_CHECK(hipSetDevice(ctxs[d]->device));
                MOTIFS_HIP_CHECK(hipStreamSynchronize(ctxs[d]->stream));
            }
        } else {
            std::vector<int64_t> part((size_t)2 * K);
            memset(per_pwm_counts2, 0, (size_t)2 * K * 8);
            for (int d = 0; d < n_dev; d++) {
                MOTIFS_HIP_CHECK(hipSetDevice(ctxs[d]->device));
                MOTIFS_HIP_CHECK(hipMemcpyAsync(part.data(), ctxs[d]->pwmcnt.p, (size_t)2 * K * 8, hipMemcpyDeviceToHost, ctxs[d]->stream));
                MOTIFS_HIP_CHECK(hipStreamSynchronize(ctxs[d]->stream));
                for (int i = 0; i < 2 * K; i++) per_pwm_counts2[i] += part[i];
            }
        }
    }
    if (too_small && !count_only) {
        set_error("hit buffer too small: need %lld / %lld records (forward / reverse), cap %lld", (long long)n_out2[0], (long long)n_out2[1],
                  (long long)cap);
        return MOTIFS_ERR_BUFFER_TOO_SMALL;
    }
    return MOTIFS_OK;
}

}  // extern "C"
