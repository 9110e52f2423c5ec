// cdl_model.hip — the unrolled-ADMM graph of src/model.jl on the engine of
// cdl_engine.h, the training step of src/train.jl:41-52 and the code retrieval of
// src/inference/_1_code_retrieval.jl:33-56, behind the C ABI of motifs_hip.h.
//
// The graph is the reference's, function by function (cited below), written on
// the compact stride-4 tensors: `z_mask_n`, `mapclarge` and `mapdrange`
// (model.jl:39-65) only encode "every 4th row" and "keep f_len lags" and vanish.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "api_common.h"
#include "cdl_engine.h"

using namespace motifs;

struct motifs_model {
    motifs_ctx* ctx = nullptr;
    motifs_hparams hp{};
    int L = 0;                       // bp
    int fl, f_len, M, twoM, h, K, q, B, L4, c, l;
    size_t nD, nF, nV, nP;           // parameter counts: D, F, the 33 vector entries, total
    float warm[3] = {0, 0, 0};       // lambda_sparsity_warmup, lambda_stepsize_warmup, omega_stepsize_warmup (not trained)
    float* params = nullptr;         // device [D | F | vecs]
    float* grads = nullptr;          // device, same layout
    float* ada_m = nullptr;
    float* ada_s = nullptr;
    double b1p = 0.9, b2p = 0.999;   // running powers of beta (Flux AdaBelief state)
    Engine eng;
    size_t arena_bytes = 0;
    // vector offsets inside the 33-entry block, Flux.params order (model.jl:68-82)
    int o_ls, o_ks, o_lst, o_ost, o_kst, o_pen, o_mu;
    Tensor last_X = nullptr;
    int last_groups = 0;
    // The forward/backward graph of a step is a fixed sequence of ~350 short launches on fixed arena addresses (527 in round 2):
    // from the second call with the same (mini-batch count, buffers) on it is captured once into a hipGraph and replayed, which
    // takes the host out of the reference's one-step-per-6-reads schedule (train.jl:40-46).  A dependent launch is 1.6-2.0 us
    // replayed against 3.0 us from the stream (tools/ubench/launch_floor.hip); the kernels' own 2-10 us bound the step, so the
    // replay is worth a percent or two.  Every node is a kernel (see motifs_model_loss_grad_dev).  Steps of more than
    // graph_max_groups mini-batches hide their launches and stay eager.  MOTIFS_NO_GRAPH=1: always eager.
    struct StepGraph {
        int n_groups = 0;
        const void* codes = nullptr;
        void* loss = nullptr;
        void* grad = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<StepGraph> step_graphs;
    hipStream_t cap_stream = nullptr;
    bool use_graphs = true;
    int graph_max_groups = 8;
};

namespace {

struct Graph {
    motifs_model* m;
    Engine& e;
    int G, S;
    ToepGeom gD1, gD2, gF1, gF2;
    Tensor Sone;
    const uint8_t* codes = nullptr;   // the base codes Sone was made from (rows of motifs_codes_pitch(L) bytes): a4 reads them directly
    Tensor pre_of = nullptr;          // codes written by a step kernel that filled pass 0 of their median select, and its workspace
    float* pre_ws = nullptr;
    float* med_ws() { return e.zeros(median_workspace_bytes(G) / 4 + 64); }
    Tensor thr_of = nullptr;          // the codes whose medians were taken last, and where they are
    const float* thr_last = nullptr;

    Graph(motifs_model* mm, int groups) : m(mm), e(mm->eng), G(groups), S(groups * mm->B) {
        const int L4 = m->L4, c = m->c, l = m->l, twoM = m->twoM, K = m->K, h = m->h, fl = m->fl;
        gD1 = ToepGeom{S, c, 4 * fl, twoM, 4, 0, L4, L4, (int64_t)c * twoM, m->B, 0};
        gD2 = ToepGeom{S, m->L, fl * twoM, 4, twoM, -(fl - 1) * twoM, c * twoM, (int64_t)c * twoM, L4, m->B, 0};
        gF1 = ToepGeom{S, l, h * twoM, K, twoM, 0, c * twoM, (int64_t)c * twoM, (int64_t)l * K, m->B, 0};
        gF2 = ToepGeom{S, c, h * K, twoM, K, -(h - 1) * K, l * K, (int64_t)l * K, (int64_t)c * twoM, m->B, 0};
    }
    // geometry with the filter stride of a bank that has `g` copies (1 = shared)
    ToepGeom with(const ToepGeom& gm, int g) const {
        ToepGeom r = gm;
        r.ldb = g == 1 ? 0 : (int64_t)gm.Q * gm.N;
        return r;
    }
    Tensor sq(Tensor x) { return e.mul(x, x); }

    // cat_ZY (model.jl:206-210): magnifying_factor * (ZY >= median of the positive entries of the mini-batch) .* ZY
    Tensor cat_ZY(Tensor ZY) {
        const float* thr = zy_thr(ZY);
        if (e.failed) return ZY;
        return e.thrmul(ZY, thr, G, m->hp.magnifying_factor);
    }
    // the medians behind cat_ZY's mask, one per mini-batch (constants in the backward, @ignore :208): consumers compare
    // against them in their own pass instead of reading a 0/1 mask
    const float* zy_thr(Tensor ZY) {
        if (ZY == thr_of) return thr_last;                                // the final codes are thresholded twice (:251, :364)
        Tensor thr = e.make((size_t)G + 64, false);
        const bool pre = ZY == pre_of && pre_ws;                          // the step kernel counted the top digits on the way
        float* ws = pre ? pre_ws : e.zeros(median_workspace_bytes(G) / 4 + 64);
        if (e.failed) return nullptr;
        median_threshold(e.st, ZY->v, thr->v, G, (int)(ZY->n / G), ws, pre);
        thr_of = ZY;
        thr_last = thr->v;
        return thr->v;
    }
    // project_X (model.jl:181-192): keep the entries >= the q-th largest of each sequence
    Tensor project_X(Tensor Xu, float scale = 1.0f) { return e.x_project(Xu, nullptr, nullptr, S, m->q, scale); }   // project_X(scale * Xu)
    // update_X's step and projection in one (:253-254)
    Tensor step_project_X(Tensor X, Tensor xg, Tensor ost) { return e.x_project(X, xg, ost, S, m->q); }
    // the two filter banks in GEMM layout, analysis and (flipped) synthesis form
    struct Bank {
        Tensor an, syn, raw;   // analysis layout, flipped synthesis layout, the reference layout it came from
        int g;
    };
    SpDims spd(int g) const {
        return SpDims{S, m->B, m->l, m->K, m->c, m->twoM, m->h, g == 1 ? 0 : (int64_t)m->h * m->twoM * m->K};
    }
    Bank bankD(Tensor D, int g) {   // D [g][M][4fl]: both forms and the re-layouts their consumers ask for, in one launch
        auto b = e.bankD(D, g, m->M, m->fl);
        return Bank{b.first, b.second, D, g};
    }
    Bank bankF(Tensor F, int g) {   // F [g][K][2M][h]: both forms in one launch
        auto b = e.bankF(F, g, m->K, m->twoM, m->h);
        return Bank{b.first, b.second, F, g};
    }
    Tensor synD(Tensor ZY, const Bank& b) { return e.toep(ZY, b.syn, with(gD2, b.g)); }     // sum_m conv(Z,D)+conv(Y,D,flipped)
    Tensor synD_plus(Tensor ZY, const Bank& b, float sgn) { return e.toep_plus(ZY, b.syn, with(gD2, b.g), Sone, sgn); }   // ... + sgn * S in the same pass
    Tensor anaD(Tensor sig, const Bank& b) {      // [conv(.,D,flipped) | conv(.,D)] rows 1:4:end
        if (sig == Sone && codes) return e.toep_onehot(sig, codes, motifs_codes_pitch(m->L), b.an, with(gD1, b.g));   // a4: the reads themselves
        return e.toep(sig, b.an, with(gD1, b.g));
    }
    // syntax layer: X keeps ~q entries per read, so synthesis and every adjoint run per non-zero
    Tensor synF(Tensor X, const Bank& b) { return e.sp_syn(X, b.syn, b.raw, spd(b.g)); }    // sum(conv(X,F,pad,groups=K),dims=3)
    Tensor anaF(Tensor img, const Bank& b) { return e.ana_sp(img, b.an, b.syn, spd(b.g), with(gF1, b.g)); }   // conv(img,F,flipped)
};

struct Scalars {
    std::vector<Tensor> ls, ks, lst, ost, kst, pen, mu;   // prepped (squared), model.jl:154-163
};

}  // namespace

static Scalars prep_scalars(motifs_model* m, Graph& gr, bool train) {
    Engine& e = m->eng;
    float* v = m->params + m->nD + m->nF;
    float* g = m->grads + m->nD + m->nF;
    // all 33 entries are squared in one launch; the scalars are one-element views of the result and of its gradient
    Tensor sqv = gr.sq(e.wrap(v, g, m->nV, train));
    float* sg = train ? e.grad(sqv) : nullptr;
    auto block = [&](int off, int n) {
        std::vector<Tensor> out;
        for (int i = 0; i < n; i++) out.push_back(e.wrap(sqv->v + off + i, sg ? sg + off + i : nullptr, 1, train && sg));
        return out;
    };
    Scalars s;
    const int x = m->hp.num_pass_xyz, d = m->hp.num_pass_df;
    s.ls = block(m->o_ls, x);
    s.ks = block(m->o_ks, d);
    s.lst = block(m->o_lst, x);
    s.ost = block(m->o_ost, x);
    s.kst = block(m->o_kst, d);
    s.pen = block(m->o_pen, x);
    s.mu = block(m->o_mu, d);
    return s;
}

// ADMM_XYZ (model.jl:330-357) on G mini-batches.  Returns ZY (unmagnified codes) and X.
static void admm_xyz(motifs_model* m, Graph& gr, const Scalars& sc, const Graph::Bank& bD, const Graph::Bank& bF,
                     Tensor& ZY, Tensor& X, Tensor* FX_out = nullptr) {
    Engine& e = m->eng;
    const float lspw = m->warm[0] * m->warm[0], lsw = m->warm[1] * m->warm[1], osw = m->warm[2] * m->warm[2];
    // warm-up (:224-232, :171-179, :212-216)
    Tensor raw = gr.anaD(gr.Sone, bD);                                   // D'S | DS on the aligned rows
    ZY = e.shrink(raw, lsw, -lspw * lsw);
    X = gr.project_X(gr.anaF(gr.cat_ZY(ZY), bF), osw);
    Tensor FX = gr.synF(X, bF);
    Tensor ab = nullptr;                                                  // scaled duals alpha|beta, zero at start (:338)
    e.note("ZY0", ZY);
    e.note("X0", X);
    for (int t = 0; t < m->hp.num_pass_xyz; t++) {
        // update_ZY (:237-245)
        Tensor diff = gr.synD_plus(ZY, bD, -1.0f);
        Tensor g1 = gr.anaD(diff, bD);
        float* mws = gr.med_ws();                                         // the median select of the new codes starts in the step kernel
        uint32_t* h0 = mws ? median_hist_ptr(mws, gr.G) : nullptr;
        if (t == 0) {
            ZY = e.zy_step(ZY, g1, FX, nullptr, sc.pen[t], sc.lst[t], sc.ls[t], h0, gr.G);   // z_grad/y_grad + the shrinkage (:240-244)
        } else {
            // the dual update that closes the previous pass (:263-266: ab += FX - ZY) rides in the same kernel
            auto r = e.zy_step2(ZY, g1, FX, ab, sc.pen[t], sc.lst[t], sc.ls[t], h0, gr.G);
            ZY = r.first;
            ab = r.second;
        }
        gr.pre_of = h0 ? ZY : nullptr;
        gr.pre_ws = mws;
        // update_X (:247-254); `sum(FX, dims=3)` is a no-op on the already summed FX
        // FX - (cat_ZY(ZY) - [alpha beta]); the magnified, median-masked image is formed inside the same pass
        const float* zt = gr.zy_thr(ZY);
        // (for t >= 1 the VJP of this combination is folded into the backward kernel of the step that made ZY and ab)
        Tensor img = t == 0 ? e.lin3(FX, 1.0f, ZY, -m->hp.magnifying_factor, ab, 1.0f, zt, gr.G)
                            : e.lin3_zy(FX, ZY, -m->hp.magnifying_factor, ab, zt, gr.G);
        Tensor xg = gr.anaF(img, bF);
        X = gr.step_project_X(X, xg, sc.ost[t]);
        FX = gr.synF(X, bF);                                              // the duals advance at the top of the next pass
    }
    e.note("ZY", ZY);
    e.note("X", X);
    if (FX_out) *FX_out = FX;                                             // sum(conv(X, F)) of the final codes
}

// forward_pass_return_loss (model.jl:375-395) for G mini-batches; per-group losses in `loss` ([G]).
static Tensor forward_loss(motifs_model* m, Graph& gr, bool train) {
    Engine& e = m->eng;
    Tensor Draw = e.wrap(m->params, m->grads, m->nD, train);
    Tensor Fraw = e.wrap(m->params + m->nD, m->grads + m->nD, m->nF, train);
    Scalars sc = prep_scalars(m, gr, train);
    // prep_filters (:139-146), prep_syntax_filters (:148-151)
    Tensor Dp = e.norm4sq(Draw, 0.001f);
    Tensor Fp = e.norml2(Fraw, m->h * m->twoM, true);
    e.note("Dp", Dp);
    e.note("Fp", Fp);
    Graph::Bank bD = gr.bankD(Dp, 1), bF = gr.bankF(Fp, 1);
    Tensor ZY, X, FXfin;
    admm_xyz(m, gr, sc, bD, bF, ZY, X, &FXfin);

    // ADMM_DF (:362-373)
    // ZYm = cat_ZY(ZY) (:364) is never materialised: its consumers take ZY, the factor and the medians of the mini-batches
    // and apply the mask in their own pass (the same products, -mag * ZY where ZY >= median)
    const float* zmt = gr.zy_thr(ZY);
    const float zmf = -m->hp.magnifying_factor;
    Tensor Dc = Dp, Fc = Fp, theta = nullptr;
    int gD = 1, gF = 1;
    Graph::Bank bDc = bD, bFc = bF;
    const int G = gr.G;
    Tensor FXcur = FXfin;                                                  // sum(conv(X, F)) with the bank in force (:294, :305, :321)
    // The residual of update_F, R_t = FX(F_t) - ZYm - theta_{t-1} (:294-296), with theta_t = theta_{t-1} + FX(F_{t+1}) - ZYm
    // (:370), telescopes: the synthesis that closes pass t is the one that opens pass t + 1, so
    //     R_0 = FX(F_0) - ZYm,   R_t = -theta_{t-2}  (t >= 1, theta_{-1} = 0: R_1 is identically zero),
    // in any precision up to the rounding of one addition.  The literal sequence forms R_t from three image-sized
    // tensors and keeps every theta and every closing synthesis; here only what a later pass or the loss consumes is
    // formed (theta_t for t <= P - 3, the closing synthesis of those passes and of the last).  MOTIFS_DF_LITERAL=1 runs
    // the literal sequence (tests/test_model_gpu.py compares the two).
    static const bool literal = getenv("MOTIFS_DF_LITERAL") != nullptr;
    const int P = m->hp.num_pass_df;
    const size_t nbank = (size_t)G * m->h * m->twoM * m->K;
    std::vector<Tensor> thetas;                                            // theta_0 ..
    // the proximal step on the gradient wgrad_sp wrote, [G][h][2M][K]: read through swap02's map while the bank is small (a bank
    // per mini-batch of a large step goes through the tiled swap, whose reads are contiguous)
    const int fseg = m->h * m->twoM;
    auto f_step_of = [&](Tensor Fg_hnk, Tensor Fcur, float sg, int t) {      // ... and the normalisation after it (:306-308)
        if (nbank <= ((size_t)1 << 20)) return e.f_step_norm(Fcur, Fg_hnk, sg, sc.kst[t], sc.ks[t], nbank, fseg, m->h, m->twoM, m->K);
        return e.f_step_norm(Fcur, e.swap02(Fg_hnk, G, m->h, m->twoM, m->K), sg, sc.kst[t], sc.ks[t], nbank, fseg);
    };
    for (int t = 0; t < P; t++) {
        // update_D (:275-290): D_grad = Z'(sumZD + sumYRD + S) + reverse(Y'(...)), only the f_len needed lags
        Tensor sig = gr.synD_plus(ZY, bDc, 1.0f);
        Dc = e.d_step(e.wgrad(sig, ZY, gr.gD1), sc.mu[t], Dc, G, m->M, m->fl);     // collapseD of the expanded gradient rides in the step
        gD = G;
        bDc = gr.bankD(Dc, gD);
        // update_F (:292-308)
        if (literal) {
            Tensor R = e.lin3(FXcur, 1.0f, ZY, zmf, theta, -1.0f, zmt, G);
            Tensor Fgrad = e.swap02(e.wgrad_sp(R, X, gr.spd(G)), G, m->h, m->twoM, m->K);
            Fc = e.norml2(e.f_step(Fc, Fgrad, 1.0f, sc.kst[t], sc.ks[t], nbank), m->h * m->twoM);
        } else if (t == 0) {
            Tensor R = e.lin3(FXcur, 1.0f, ZY, zmf, nullptr, 0.0f, zmt, G);
            Fc = f_step_of(e.wgrad_sp(R, X, gr.spd(G)), Fc, 1.0f, t);
        } else if (t == 1) {                                               // R_1 = 0: the gradient of F vanishes identically
            Fc = e.f_step_norm(Fc, nullptr, 1.0f, sc.kst[t], sc.ks[t], nbank, fseg);
        } else {                                                           // R_t = -theta_{t-2}: the sign goes into the step
            Fc = f_step_of(e.wgrad_sp(thetas[t - 2], X, gr.spd(G)), Fc, -1.0f, t);
        }
        gF = G;
        bFc = gr.bankF(Fc, gF);
        // theta (:370) and the synthesis with the new bank, where something consumes them
        const bool need_theta = literal || t + 2 < P;
        if (need_theta || t == P - 1) FXcur = gr.synF(X, bFc);
        if (need_theta) {
            theta = e.lin3(FXcur, 1.0f, ZY, zmf, theta, 1.0f, zmt, G);
            thetas.push_back(theta);
        }
    }
    e.note("Dfinal", Dc);
    e.note("Ffinal", Fc);
    // loss (:310-325)
    const float nf = 1.0f / (float)m->B;
    Tensor r1 = gr.synD_plus(ZY, bDc, -1.0f);
    Tensor Lv = e.resid_sumsq_groups(FXcur, ZY, zmf, zmt, nf, G, e.sumsq_groups(r1, nf, G));     // the two terms meet in one buffer
    (void)gD;
    (void)gF;
    return Lv;
}

static int check_model(motifs_model* m, const char* fn) {
    if (!m || !m->ctx) {
        set_error("%s: null model", fn);
        return MOTIFS_ERR_INVALID;
    }
    return MOTIFS_OK;
}

// one-hot signal of G*B sequences from a code matrix (rows of motifs_codes_pitch(L) bytes)
static Tensor make_onehot(motifs_model* m, const uint8_t* codes_dev, int S) {
    Tensor t = m->eng.make((size_t)S * m->L4, false);
    if (!m->eng.failed) onehot_from_codes(m->eng.st, codes_dev, motifs_codes_pitch(m->L), t->v, S, m->L);
    return t;
}

__global__ void k_fill(float* x, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_abs_sum(const float* x, size_t n, float* out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += fabsf(x[i]);
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (float)acc);
}

// ---- code retrieval kernels ---------------------------------------------------------------------
__global__ void k_count_pos(const float* X, int n, int32_t* cnt) {
    const float* xs = X + (size_t)blockIdx.x * n;
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) c += xs[i] > 0.0f;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (threadIdx.x == 0) cnt[blockIdx.x] = c;
}
// one wave per sequence; records in the order of `findall(X .> 0)` over (l, 1, K, B):
// position fastest, then syntax filter, then sequence (_1_code_retrieval.jl:27-31)
__global__ void k_write_codes(const float* X, int l, int K, const int64_t* off, int64_t seq0, motifs_code_rec* out) {
    const float* xs = X + (size_t)blockIdx.x * l * K;        // [l][K]
    int64_t at = off[blockIdx.x];
    const int lane = threadIdx.x;
    for (int k = 0; k < K; k++)
        for (int p0 = 0; p0 < l; p0 += 64) {
            const int p = p0 + lane;
            const float v = p < l ? xs[(size_t)p * K + k] : 0.0f;
            const bool hit = v > 0.0f;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const int r = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                motifs_code_rec rec;
                rec.position = (uint16_t)(p + 1);
                rec.fil = (uint16_t)(k + 1);
                rec.seq = (uint32_t)(seq0 + blockIdx.x + 1);
                rec.mag = __half_as_ushort(__float2half_rn(v));
                rec.pad_ = 0;
                out[at + r] = rec;
            }
            at += __builtin_popcountll(mask);
        }
}

// every launch of one loss + gradient evaluation, on stream st (the context's stream, or a capturing one)
static int enqueue_loss_grad(motifs_model* m, hipStream_t st, const uint8_t* codes_dev, int n_groups, float* loss_dev, float* grad_flat_dev,
                             int keep_intermediates) {
    Engine& e = m->eng;
    e.st = st;
    struct RestoreStream {          // the engine goes back to the context's stream on every way out of this function
        motifs_model* m;
        ~RestoreStream() { m->eng.st = m->ctx->stream; }
    } restore{m};
    e.reset();
    e.recording = grad_flat_dev != nullptr;
    e.keep_named = keep_intermediates != 0;
    dev_zero(e.st, m->grads, m->nP);
    Graph gr(m, n_groups);
    gr.Sone = make_onehot(m, codes_dev, gr.S);
    gr.codes = codes_dev;
    Tensor Lv = forward_loss(m, gr, e.recording);
    if (!e.failed && e.recording) {
        float* g = e.grad(Lv);
        if (g) hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, e.st, g, (size_t)n_groups, 1.0f);
        e.backward();
    }
    if (e.failed) {
        set_error("engine arena exhausted (%zu bytes): lower n_groups or create the model with a larger arena",
                  m->arena_bytes);
        return MOTIFS_ERR_UNSUPPORTED;
    }
    if (loss_dev) dev_copy(e.st, loss_dev, Lv->v, (size_t)n_groups);
    if (grad_flat_dev) dev_copy(e.st, grad_flat_dev, m->grads, m->nP);
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

extern "C" {

int motifs_model_create(motifs_ctx* ctx, const motifs_hparams* hp, int L, size_t arena_bytes, motifs_model** out) {
    if (!ctx || !hp || !out || L <= 0) {
        set_error("motifs_model_create: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    *out = nullptr;
    const int c = L - hp->filter_len + 1, l = c - hp->h + 1;
    if (hp->filter_len < 1 || hp->M < 1 || hp->h < 1 || hp->K < 1 || hp->batch_size < 1 || hp->num_pass_xyz < 1 ||
        hp->num_pass_df < 1 || c < 1 || l < 1 || hp->q < 1 || hp->q > l * hp->K) {
        set_error("motifs_model_create: hyper-parameters do not fit L=%d (c=%d, l=%d, q=%d)", L, c, l, hp->q);
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(ctx->device));
    motifs_model* m = new motifs_model();
    m->ctx = ctx;
    m->hp = *hp;
    m->L = L;
    m->fl = hp->filter_len;
    m->f_len = 4 * hp->filter_len;
    m->M = hp->M;
    m->twoM = 2 * hp->M;
    m->h = hp->h;
    m->K = hp->K;
    m->q = hp->q;
    m->B = hp->batch_size;
    m->L4 = 4 * L;
    m->c = c;
    m->l = l;
    m->nD = (size_t)m->M * m->f_len;
    m->nF = (size_t)m->K * m->twoM * m->h;
    const int x = hp->num_pass_xyz, d = hp->num_pass_df;
    m->o_ls = 0;
    m->o_ks = m->o_ls + x;
    m->o_lst = m->o_ks + d;
    m->o_ost = m->o_lst + x;
    m->o_kst = m->o_ost + x;
    m->o_pen = m->o_kst + d;
    m->o_mu = m->o_pen + x;
    m->nV = (size_t)(m->o_mu + d);
    m->nP = m->nD + m->nF + m->nV;
    MOTIFS_HIP_CHECK(hipMalloc(&m->params, m->nP * 4));
    MOTIFS_HIP_CHECK(hipMalloc(&m->grads, m->nP * 4));
    MOTIFS_HIP_CHECK(hipMalloc(&m->ada_m, m->nP * 4));
    MOTIFS_HIP_CHECK(hipMalloc(&m->ada_s, m->nP * 4));
    // (hipMemset of device memory returns before the fill has run, and it runs on the NULL stream: on a context with its own
    // non-blocking stream nothing ordered these fills against the first parameter upload - found in round 3 as NaN losses of a
    // model created right after a large one was freed.  The fills go on the context's stream and are waited for below.)
    MOTIFS_HIP_CHECK(hipMemsetAsync(m->params, 0, m->nP * 4, ctx->stream));
    MOTIFS_HIP_CHECK(hipMemsetAsync(m->ada_m, 0, m->nP * 4, ctx->stream));
    MOTIFS_HIP_CHECK(hipMemsetAsync(m->ada_s, 0, m->nP * 4, ctx->stream));
    m->arena_bytes = arena_bytes ? arena_bytes : ((size_t)8 << 30);
    void* base = nullptr;
    MOTIFS_HIP_CHECK(hipMalloc(&base, m->arena_bytes));
    m->eng.arena.base = (char*)base;
    m->eng.arena.cap = m->arena_bytes;
    // MOTIFS_POISON_ARENA=1 (debugging aid): the arena starts as NaN bit patterns (at most its first 2 GiB), so a kernel that reads
    // what nothing wrote shows up as NaN in the loss instead of hiding behind the zero pages of a fresh allocation
    if (getenv("MOTIFS_POISON_ARENA")) MOTIFS_HIP_CHECK(hipMemsetAsync(base, 0xFF, std::min<size_t>(m->arena_bytes, (size_t)2 << 30), ctx->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(ctx->stream));               // the caller may bind another stream before its first call
    m->eng.st = ctx->stream;
    m->use_graphs = getenv("MOTIFS_NO_GRAPH") == nullptr;
    *out = m;
    return MOTIFS_OK;
}

void motifs_model_destroy(motifs_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    m->eng.reset();
    for (auto& g : m->step_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    for (void* p : {(void*)m->params, (void*)m->grads, (void*)m->ada_m, (void*)m->ada_s, (void*)m->eng.arena.base})
        if (p) (void)hipFree(p);
    delete m;
}

int motifs_model_sizes(motifs_model* m, int64_t* nD, int64_t* nF, int64_t* nV, int64_t* c, int64_t* l) {
    int r = check_model(m, "motifs_model_sizes");
    if (r) return r;
    if (nD) *nD = (int64_t)m->nD;
    if (nF) *nF = (int64_t)m->nF;
    if (nV) *nV = (int64_t)m->nV;
    if (c) *c = m->c;
    if (l) *l = m->l;
    return MOTIFS_OK;
}

int motifs_model_set_params(motifs_model* m, const float* D, const float* F, const float* warmup3, const float* vecs) {
    int r = check_model(m, "motifs_model_set_params");
    if (r) return r;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    if (D) MOTIFS_HIP_CHECK(hipMemcpyAsync(m->params, D, m->nD * 4, hipMemcpyHostToDevice, st));
    if (F) MOTIFS_HIP_CHECK(hipMemcpyAsync(m->params + m->nD, F, m->nF * 4, hipMemcpyHostToDevice, st));
    if (vecs) MOTIFS_HIP_CHECK(hipMemcpyAsync(m->params + m->nD + m->nF, vecs, m->nV * 4, hipMemcpyHostToDevice, st));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(st));
    if (warmup3) {
        // the warm-up scalars are host-side constants of the step: a captured step graph carries the old ones as kernel arguments
        memcpy(m->warm, warmup3, 12);
        for (auto& g : m->step_graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        m->step_graphs.clear();
    }
    return MOTIFS_OK;
}

int motifs_model_arena_peak(motifs_model* m, size_t* bytes) {
    int r = check_model(m, "motifs_model_arena_peak");
    if (r) return r;
    if (!bytes) {
        set_error("motifs_model_arena_peak: bytes is NULL");
        return MOTIFS_ERR_INVALID;
    }
    *bytes = m->eng.arena.peak;
    return MOTIFS_OK;
}

int motifs_model_get_params(motifs_model* m, float* D, float* F, float* warmup3, float* vecs) {
    int r = check_model(m, "motifs_model_get_params");
    if (r) return r;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    if (D) MOTIFS_HIP_CHECK(hipMemcpyAsync(D, m->params, m->nD * 4, hipMemcpyDeviceToHost, st));
    if (F) MOTIFS_HIP_CHECK(hipMemcpyAsync(F, m->params + m->nD, m->nF * 4, hipMemcpyDeviceToHost, st));
    if (vecs) MOTIFS_HIP_CHECK(hipMemcpyAsync(vecs, m->params + m->nD + m->nF, m->nV * 4, hipMemcpyDeviceToHost, st));
    if (warmup3) memcpy(warmup3, m->warm, 12);
    MOTIFS_HIP_CHECK(hipStreamSynchronize(st));
    return MOTIFS_OK;
}

// ucdl(hp) (model.jl:84-100) with a seeded splitmix64 stream (the reference draws from Julia's
// unseeded global RNG, so only the distribution can be reproduced).
int motifs_model_init_random(motifs_model* m, uint64_t seed) {
    int r = check_model(m, "motifs_model_init_random");
    if (r) return r;
    uint64_t s = seed;
    auto next = [&]() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto uni = [&]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
    auto nrm = [&]() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    };
    std::vector<float> D(m->nD), F(m->nF), V(m->nV);
    for (int mm = 0; mm < m->M; mm++)            // randomly_initialize_filters (MOTIFs.jl:17-33) then sqrt (:88)
        for (int k = 0; k < m->fl; k++) {
            double u[3] = {uni(), uni(), uni()};
            std::sort(u, u + 3);
            const double sp[4] = {u[0], u[1] - u[0], u[2] - u[1], 1.0 - u[2]};
            for (int a = 0; a < 4; a++) D[(size_t)mm * m->f_len + 4 * k + a] = (float)std::sqrt(sp[a]);
        }
    for (size_t i = 0; i < m->nF; i++) F[i] = (float)std::fabs(0.1 * nrm());   // :90
    float warm[3];
    warm[0] = (float)(0.05 * uni());
    for (size_t i = 0; i < m->nV; i++) V[i] = (float)(0.05 * uni());
    warm[1] = (float)(0.05 * uni());
    warm[2] = (float)(0.05 * uni());
    m->b1p = 0.9;
    m->b2p = 0.999;
    MOTIFS_HIP_CHECK(hipMemsetAsync(m->ada_m, 0, m->nP * 4, m->ctx->stream));    // on the stream the optimiser runs on (see motifs_model_create)
    MOTIFS_HIP_CHECK(hipMemsetAsync(m->ada_s, 0, m->nP * 4, m->ctx->stream));
    return motifs_model_set_params(m, D.data(), F.data(), warm, V.data());
}

// loss and gradient of G mini-batches: train.jl:42-44.  grad_flat_dev receives the SUM over the
// groups of d loss_g / d [D | F | vecs]; loss_dev the G losses.
int motifs_model_loss_grad_dev(motifs_model* m, const uint8_t* codes_dev, int n_groups, float* loss_dev,
                               float* grad_flat_dev, int keep_intermediates) {
    int r = check_model(m, "motifs_model_loss_grad_dev");
    if (r) return r;
    if (!codes_dev || n_groups < 1 || (int64_t)n_groups * m->B > 65535) {
        set_error("motifs_model_loss_grad_dev: bad argument (n_groups=%d)", n_groups);
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    KernelTimer tm(m->ctx, KS_TRAIN_STEP);
    // Every node of the captured step is a kernel: dev_zero / dev_copy instead of hipMemsetAsync / hipMemcpyAsync.  With memset
    // and memcpy nodes in it (rounds 2 and 3, ROCm 7.0 / 7.2) a replay was not ordered with its neighbours: on the context's
    // own stream later replays returned the loss of a half-updated buffer whenever the reads behind the same pointers had
    // changed, and on HIP's legacy null stream gradients of ~1e28 (tests/test_model_gpu.py::
    // test_replayed_steps_follow_the_reads, tests/test_round3_gpu.py::test_step_graph_on_the_null_stream).
    if (!m->use_graphs || keep_intermediates || n_groups > m->graph_max_groups) {
        Engine::Probe probe;
        if ((m->ctx->timing >> KS_TRAIN_ISTA_BWD) & 1u) {
            motifs_ctx* c = m->ctx;
            probe.get = [c]() { return KernelTimer::get(c); };
            m->eng.probe = &probe;
        }
        const int rc = enqueue_loss_grad(m, m->ctx->stream, codes_dev, n_groups, loss_dev, grad_flat_dev, keep_intermediates);
        m->eng.probe = nullptr;
        for (auto& pr : probe.pairs) m->ctx->pending.push_back({KS_TRAIN_ISTA_BWD, pr.first, pr.second});
        return rc;
    }
    motifs_model::StepGraph* sg = nullptr;
    for (auto& g : m->step_graphs)
        if (g.n_groups == n_groups && g.codes == codes_dev && g.loss == loss_dev && g.grad == grad_flat_dev) sg = &g;
    if (!sg) {       // first sight of this step: eager (lazy kernel loading and first-use set-up happen outside any capture)
        if (m->step_graphs.size() >= 8) {
            if (m->step_graphs.front().exec) (void)hipGraphExecDestroy(m->step_graphs.front().exec);
            m->step_graphs.erase(m->step_graphs.begin());
        }
        motifs_model::StepGraph g;
        g.n_groups = n_groups;
        g.codes = codes_dev;
        g.loss = loss_dev;
        g.grad = grad_flat_dev;
        m->step_graphs.push_back(g);
        return enqueue_loss_grad(m, m->ctx->stream, codes_dev, n_groups, loss_dev, grad_flat_dev, 0);
    }
    if (!sg->exec) {   // second call: record the launches instead of running them
        if (!m->cap_stream) MOTIFS_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        MOTIFS_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        const int rc = enqueue_loss_grad(m, m->cap_stream, codes_dev, n_groups, loss_dev, grad_flat_dev, 0);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(m->cap_stream, &graph);
        if (rc != MOTIFS_OK || ce != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (rc != MOTIFS_OK) return rc;   // the step itself failed (arena exhausted, a HIP error): not a capture problem, graphs stay on
            m->use_graphs = false;     // something in the step cannot be captured on this runtime: stay eager from here on
            return enqueue_loss_grad(m, m->ctx->stream, codes_dev, n_groups, loss_dev, grad_flat_dev, 0);
        }
        const hipError_t ie = hipGraphInstantiate(&sg->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) {
            sg->exec = nullptr;
            (void)hipGetLastError();
            m->use_graphs = false;
            return enqueue_loss_grad(m, m->ctx->stream, codes_dev, n_groups, loss_dev, grad_flat_dev, 0);
        }
    }
    MOTIFS_HIP_CHECK(hipGraphLaunch(sg->exec, m->ctx->stream));
    return MOTIFS_OK;
}

// Flux.Optimise.update!(opt, ps, gs) with opt = AdaBelief() (train.jl:35, :46) on the flat parameter block;
// the gradient used is gscale * grad (1/n_groups turns the summed gradient into the mean).
int motifs_model_adabelief_dev(motifs_model* m, const float* grad_flat_dev, float gscale) {
    int r = check_model(m, "motifs_model_adabelief_dev");
    if (r) return r;
    if (!grad_flat_dev) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    adabelief_step(m->ctx->stream, m->params, m->ada_m, m->ada_s, grad_flat_dev, m->nP, gscale, 1e-3f, 0.9f, 0.999f, 1e-8f,
                   m->b1p, m->b2p);
    m->b1p *= 0.9;
    m->b2p *= 0.999;
    MOTIFS_HIP_CHECK(hipGetLastError());
    return MOTIFS_OK;
}

// The exchange of the data-parallel step: one in-place sum of the flat gradient over the ranks, on the
// context's stream, i.e. behind the backward kernels that wrote it and in front of the AdaBelief kernel.
int motifs_model_allreduce_grad(motifs_model* m, motifs_comm* comm, float* grad_flat_dev) {
    int r = check_model(m, "motifs_model_allreduce_grad");
    if (r) return r;
    if (!comm || !grad_flat_dev) {
        set_error("motifs_model_allreduce_grad: comm or gradient is NULL");
        return MOTIFS_ERR_INVALID;
    }
    return motifs_comm_allreduce_sum_f32_dev(comm, grad_flat_dev, (int64_t)m->nP);
}

// train.jl:42-46 for a shard, in three phases: local summed gradient -> sum over ranks -> AdaBelief on the mean over all
// mini-batches.  Every rank applies the same update to the same parameters, so the replicas never diverge.  A rank without
// mini-batches (more ranks than mini-batches, or the tail of an uneven split) contributes zeros and still joins
// the exchange: skipping it would leave the other ranks waiting in the all-reduce.
int motifs_model_dp_grad_dev(motifs_model* m, const uint8_t* codes_dev, int n_groups_local, float* loss_dev, float* grad_flat_dev) {
    int r = check_model(m, "motifs_model_dp_grad_dev");
    if (r) return r;
    if (!grad_flat_dev || n_groups_local < 0 || (n_groups_local > 0 && !codes_dev)) {
        set_error("motifs_model_dp_grad_dev: bad argument (local=%d)", n_groups_local);
        return MOTIFS_ERR_INVALID;
    }
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    if (n_groups_local > 0) return motifs_model_loss_grad_dev(m, codes_dev, n_groups_local, loss_dev, grad_flat_dev, 0);
    MOTIFS_HIP_CHECK(hipMemsetAsync(grad_flat_dev, 0, m->nP * 4, m->ctx->stream));
    return MOTIFS_OK;
}

int motifs_model_dp_update_dev(motifs_model* m, const float* grad_flat_dev, int64_t n_groups_total) {
    int r = check_model(m, "motifs_model_dp_update_dev");
    if (r) return r;
    if (!grad_flat_dev || n_groups_total < 1) {
        set_error("motifs_model_dp_update_dev: bad argument (total=%lld)", (long long)n_groups_total);
        return MOTIFS_ERR_INVALID;
    }
    return motifs_model_adabelief_dev(m, grad_flat_dev, (float)(1.0 / (double)n_groups_total));
}

// One rank of a one-rank-per-process (or per-thread) communicator.  Inside an open ncclGroupStart/End RCCL defers the
// all-reduce to the group's end, and the AdaBelief kernel below would be enqueued - and run - before it: refused.
int motifs_model_dp_train_step_dev(motifs_model* m, motifs_comm* comm, const uint8_t* codes_dev, int n_groups_local,
                                   int64_t n_groups_total, float* loss_dev, float* grad_flat_dev) {
    int r = check_model(m, "motifs_model_dp_train_step_dev");
    if (r) return r;
    if (!grad_flat_dev || n_groups_local < 0 || n_groups_total < 1 || n_groups_total < n_groups_local ||
        (!comm && n_groups_local != n_groups_total)) {
        set_error("motifs_model_dp_train_step_dev: bad argument (local=%d total=%lld)", n_groups_local, (long long)n_groups_total);
        return MOTIFS_ERR_INVALID;
    }
    if (comm && comm_group_depth() > 0) {
        set_error("motifs_model_dp_train_step_dev called between motifs_comm_group_start and _end: the all-reduce would only be launched "
                  "at the group's end, after the optimiser kernel; use motifs_model_dp_train_step_all, or group only motifs_model_allreduce_grad");
        return MOTIFS_ERR_INVALID;
    }
    r = motifs_model_dp_grad_dev(m, codes_dev, n_groups_local, loss_dev, grad_flat_dev);
    if (r) return r;
    if (comm) {
        r = motifs_model_allreduce_grad(m, comm, grad_flat_dev);
        if (r) return r;
    }
    return motifs_model_dp_update_dev(m, grad_flat_dev, n_groups_total);
}

// One host thread, n_dev replicas (motifs_comm_create_all): gradients first, then the grouped all-reduces, then the updates.
int motifs_model_dp_train_step_all(motifs_model* const* models, motifs_comm* const* comms, int n_dev, const uint8_t* const* codes_dev,
                                   const int* n_groups_local, int64_t n_groups_total, float* const* loss_dev, float* const* grad_dev,
                                   float* const* reduced_dev) {
    if (!models || n_dev < 1 || !codes_dev || !n_groups_local || !loss_dev || !grad_dev || n_groups_total < 1 || (!comms && n_dev != 1)) {
        set_error("motifs_model_dp_train_step_all: bad argument (n_dev=%d total=%lld)", n_dev, (long long)n_groups_total);
        return MOTIFS_ERR_INVALID;
    }
    int64_t local_sum = 0;
    for (int d = 0; d < n_dev; d++) {
        int r = check_model(models[d], "motifs_model_dp_train_step_all");
        if (r) return r;
        if (!grad_dev[d] || n_groups_local[d] < 0 || (reduced_dev && !reduced_dev[d]) || models[d]->nP != models[0]->nP ||
            (comms && comm_ctx(comms[d]) != models[d]->ctx)) {
            set_error("motifs_model_dp_train_step_all: slot %d: null buffer, negative count, a replica of another shape, or a communicator "
                      "that was not made on the model's context", d);
            return MOTIFS_ERR_INVALID;
        }
        for (int j = 0; j < d; j++)
            if (models[j]->ctx == models[d]->ctx) {
                set_error("motifs_model_dp_train_step_all: slots %d and %d share a context (one replica per device)", j, d);
                return MOTIFS_ERR_INVALID;
            }
        local_sum += n_groups_local[d];
    }
    if (local_sum > n_groups_total) {
        set_error("motifs_model_dp_train_step_all: %lld local mini-batches > total %lld", (long long)local_sum, (long long)n_groups_total);
        return MOTIFS_ERR_INVALID;
    }
    if (comm_group_depth() > 0) {
        set_error("motifs_model_dp_train_step_all opens its own group around the all-reduces: call it outside motifs_comm_group_start/_end");
        return MOTIFS_ERR_INVALID;
    }
    // phase 1: every device's gradient (the launches of one device are enqueued by one host thread)
    int r = for_each_device(n_dev, [&](int d) {
        return motifs_model_dp_grad_dev(models[d], codes_dev[d], n_groups_local[d], loss_dev[d], grad_dev[d]);
    });
    if (r) return r;
    // phase 2: the sums - the only part inside the group; RCCL launches them at the group's end, each on its device's stream
    // behind that device's backward kernels
    if (comms) {
        r = motifs_comm_group_start();
        if (r) return r;
        int rr = MOTIFS_OK;
        for (int d = 0; d < n_dev && rr == MOTIFS_OK; d++)
            rr = reduced_dev ? motifs_comm_allreduce_sum_f32_to_dev(comms[d], grad_dev[d], reduced_dev[d], (int64_t)models[d]->nP)
                             : motifs_model_allreduce_grad(models[d], comms[d], grad_dev[d]);
        std::string why = rr ? last_error_text() : "";
        r = motifs_comm_group_end();                     // the group is closed whatever happened inside it
        if (rr) {
            set_error("%s", why.c_str());
            return rr;
        }
        if (r) return r;
    } else if (reduced_dev) {
        MOTIFS_HIP_CHECK(hipSetDevice(models[0]->ctx->device));
        MOTIFS_HIP_CHECK(hipMemcpyAsync(reduced_dev[0], grad_dev[0], models[0]->nP * 4, hipMemcpyDeviceToDevice, models[0]->ctx->stream));
    }
    // phase 3: the identical update on every replica, enqueued behind its device's all-reduce
    for (int d = 0; d < n_dev; d++) {
        r = motifs_model_dp_update_dev(models[d], reduced_dev ? reduced_dev[d] : grad_dev[d], n_groups_total);
        if (r) return r;
    }
    return MOTIFS_OK;
}

// The step above for a host that holds no device pointers: mini-batches dealt to the replicas in contiguous blocks.
int motifs_model_dp_train_step_host(motifs_model* const* models, motifs_comm* const* comms, int n_dev, const void* data, int kind,
                                    int n_groups, float* loss_out, float* l1F_out) {
    if (!models || n_dev < 1 || !data || n_groups < 1 || kind < 0 || kind > 2 || (!comms && n_dev != 1)) {
        set_error("motifs_model_dp_train_step_host: bad argument (n_dev=%d n_groups=%d kind=%d)", n_dev, n_groups, kind);
        return MOTIFS_ERR_INVALID;
    }
    for (int d = 0; d < n_dev; d++) {
        int r = check_model(models[d], "motifs_model_dp_train_step_host");
        if (r) return r;
        if (models[d]->B != models[0]->B || models[d]->L != models[0]->L || models[d]->nP != models[0]->nP) {
            set_error("motifs_model_dp_train_step_host: replica %d has another shape than replica 0", d);
            return MOTIFS_ERR_INVALID;
        }
    }
    const int B = models[0]->B, L = models[0]->L;
    const size_t nP = models[0]->nP;
    const size_t elt = kind == MOTIFS_DATA_ONEHOT_F32 ? 16 : kind == MOTIFS_DATA_ONEHOT_F16 ? 8 : 1;
    std::vector<int> g_lo(n_dev + 1, 0), g_n(n_dev, 0);
    for (int d = 0; d < n_dev; d++) {                      // contiguous blocks, sizes differ by at most one mini-batch
        g_n[d] = n_groups / n_dev + (d < n_groups % n_dev ? 1 : 0);
        g_lo[d + 1] = g_lo[d] + g_n[d];
    }
    std::vector<const uint8_t*> codes(n_dev, nullptr);
    std::vector<float*> loss(n_dev, nullptr), grad(n_dev, nullptr);
    int r = for_each_device(n_dev, [&](int d) -> int {
        motifs_ctx* c = models[d]->ctx;
        MOTIFS_HIP_CHECK(hipSetDevice(c->device));
        if (g_n[d] > 0) {
            const int rr = upload_and_encode(c, (const char*)data + (size_t)g_lo[d] * B * L * elt, kind, (int64_t)g_n[d] * B, L);
            if (rr) return rr;
            codes[d] = (const uint8_t*)c->codes.p;
        }
        MOTIFS_HIP_CHECK(c->dp_scratch.reserve(nP * 4 + (size_t)std::max(g_n[d], 1) * 4 + 64));
        grad[d] = (float*)c->dp_scratch.p;
        loss[d] = grad[d] + nP;
        return MOTIFS_OK;
    });
    if (r) return r;
    r = motifs_model_dp_train_step_all(models, comms, n_dev, codes.data(), g_n.data(), n_groups, loss.data(), grad.data(), nullptr);
    if (r) return r;
    for (int d = 0; d < n_dev; d++) {
        motifs_ctx* c = models[d]->ctx;
        MOTIFS_HIP_CHECK(hipSetDevice(c->device));
        if (loss_out && g_n[d] > 0)
            MOTIFS_HIP_CHECK(hipMemcpyAsync(loss_out + g_lo[d], loss[d], (size_t)g_n[d] * 4, hipMemcpyDeviceToHost, c->stream));
    }
    for (int d = 0; d < n_dev; d++) {
        MOTIFS_HIP_CHECK(hipSetDevice(models[d]->ctx->device));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(models[d]->ctx->stream));
    }
    if (l1F_out) return motifs_model_l1_syntax(models[0], l1F_out);
    return MOTIFS_OK;
}

// sum(abs.(prep_syntax_filters(cdl.F))) (train.jl:47): the early-stop statistic
int motifs_model_l1_syntax(motifs_model* m, float* out) {
    int r = check_model(m, "motifs_model_l1_syntax");
    if (r) return r;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    Engine& e = m->eng;
    e.st = m->ctx->stream;
    e.reset();
    e.recording = false;
    Tensor Fraw = e.wrap(m->params + m->nD, nullptr, m->nF, false);
    Tensor Fp = e.norml2(Fraw, m->h * m->twoM, true);
    Tensor acc = e.make(1, false);
    if (e.failed) return MOTIFS_ERR_UNSUPPORTED;
    MOTIFS_HIP_CHECK(hipMemsetAsync(acc->v, 0, 4, e.st));
    hipLaunchKernelGGL(k_abs_sum, dim3(64), dim3(256), 0, e.st, Fp->v, m->nF, acc->v);
    MOTIFS_HIP_CHECK(hipMemcpyAsync(out, acc->v, 4, hipMemcpyDeviceToHost, e.st));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(e.st));
    return MOTIFS_OK;
}

// One optimiser step on host data: the body of the loop at train.jl:40-52 for n_groups mini-batches
// (n_groups = 1 is exactly the reference step).  data: n_groups*batch_size reads of `kind`.
static int train_step_host(motifs_model* m, const void* data, int kind, int n_groups, float* loss_out, float* l1F_out) {
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    motifs_ctx* c = m->ctx;
    const int64_t S = (int64_t)n_groups * m->B;
    const size_t elt = kind == MOTIFS_DATA_ONEHOT_F32 ? 16 : kind == MOTIFS_DATA_ONEHOT_F16 ? 8 : 1;
    const size_t raw = ((size_t)S * m->L * elt + 63) & ~(size_t)63;
    MOTIFS_HIP_CHECK(c->codes.reserve(motifs_codes_bytes(S, m->L)));
    MOTIFS_HIP_CHECK(c->data_tmp.reserve(raw + m->nP * 4 + (size_t)n_groups * 4 + 128));
    MOTIFS_HIP_CHECK(hipMemcpyAsync(c->data_tmp.p, data, (size_t)S * m->L * elt, hipMemcpyHostToDevice, c->stream));
    float* gflat = (float*)((char*)c->data_tmp.p + raw);
    float* lossd = gflat + m->nP;
    int32_t* bad_dev = (int32_t*)(lossd + n_groups);
    MOTIFS_HIP_CHECK(hipMemsetAsync(bad_dev, 0, 4, c->stream));
    int r = motifs_encode_dev(c, c->data_tmp.p, kind, S, m->L, (uint8_t*)c->codes.p, bad_dev);
    if (r) return r;
    if (kind != MOTIFS_DATA_CODES_U8) {       // a batch that is not one-hot must not reach the optimiser
        int32_t* h_bad = (int32_t*)((char*)c->pinned + 64);
        MOTIFS_HIP_CHECK(hipMemcpyAsync(h_bad, bad_dev, 4, hipMemcpyDeviceToHost, c->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (*h_bad) {
            set_error("training batch has a column that is neither one-hot nor all-zero");
            return MOTIFS_ERR_NOT_ONEHOT;
        }
    }
    r = motifs_model_loss_grad_dev(m, (const uint8_t*)c->codes.p, n_groups, lossd, gflat, 0);
    if (r) return r;
    r = motifs_model_adabelief_dev(m, gflat, 1.0f / (float)n_groups);
    if (r) return r;
    if (loss_out) MOTIFS_HIP_CHECK(hipMemcpyAsync(loss_out, lossd, (size_t)n_groups * 4, hipMemcpyDeviceToHost, c->stream));
    MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (l1F_out) return motifs_model_l1_syntax(m, l1F_out);
    return MOTIFS_OK;
}

// codes: n_groups*batch_size rows of L bytes (0..3).
int motifs_model_train_step(motifs_model* m, const uint8_t* codes, int n_groups, float* loss_out, float* l1F_out) {
    int r = check_model(m, "motifs_model_train_step");
    if (r) return r;
    if (!codes || n_groups < 1) return MOTIFS_ERR_INVALID;
    return train_step_host(m, codes, MOTIFS_DATA_CODES_U8, n_groups, loss_out, l1F_out);
}

// S: the reference's own batch, `S = data.data_matrix[:, :, batch]` (train.jl:33,41): (4L, 1, n_groups*batch_size) Float32.
int motifs_model_train_step_onehot(motifs_model* m, const float* S, int n_groups, float* loss_out, float* l1F_out) {
    int r = check_model(m, "motifs_model_train_step_onehot");
    if (r) return r;
    if (!S || n_groups < 1) return MOTIFS_ERR_INVALID;
    return train_step_host(m, S, MOTIFS_DATA_ONEHOT_F32, n_groups, loss_out, l1F_out);
}

// code_retrieval (_1_code_retrieval.jl:33-56): ADMM_XYZ only, batches of batch_size in file order,
// remainder dropped (partial=false, :38); records as `findall(X .> 0)` yields them.
int motifs_model_retrieve_codes(motifs_model* m, const void* data, int kind, int64_t N, motifs_code_rec* out, int64_t cap,
                                int64_t* n_out) {
    int r = check_model(m, "motifs_model_retrieve_codes");
    if (r) return r;
    if (!n_out || N < 0 || (N > 0 && !data) || cap < 0 || (cap > 0 && !out) || kind < 0 || kind > 2) {
        set_error("motifs_model_retrieve_codes: bad argument");
        return MOTIFS_ERR_INVALID;
    }
    *n_out = 0;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    motifs_ctx* c = m->ctx;
    Engine& e = m->eng;
    e.st = c->stream;
    const int64_t nfull = N - N % m->B;
    // groups per launch: bounded by the arena (forward only keeps ~40 code images alive) and the grid limit
    const size_t per_group = (size_t)m->B * ((size_t)m->c * m->twoM * 48 + (size_t)m->L4 * 16 + (size_t)m->l * m->K * 24) * 4;
    int64_t gmax = (int64_t)(m->arena_bytes / 2 / std::max<size_t>(per_group, 1));
    gmax = std::max<int64_t>(1, std::min<int64_t>(gmax, 65535 / m->B));
    const size_t elt = kind == MOTIFS_DATA_ONEHOT_F32 ? 16 : kind == MOTIFS_DATA_ONEHOT_F16 ? 8 : 1;
    int64_t total = 0;
    bool too_small = false;
    std::vector<int32_t> h_cnt;
    std::vector<int64_t> h_off;
    for (int64_t s0 = 0; s0 < nfull; s0 += gmax * m->B) {
        const int64_t S = std::min<int64_t>(gmax * m->B, nfull - s0);
        const int G = (int)(S / m->B);
        MOTIFS_HIP_CHECK(c->codes.reserve(motifs_codes_bytes(S, m->L)));
        MOTIFS_HIP_CHECK(c->data_tmp.reserve((size_t)S * m->L * elt + 64));
        MOTIFS_HIP_CHECK(hipMemcpyAsync(c->data_tmp.p, (const char*)data + (size_t)s0 * m->L * elt, (size_t)S * m->L * elt,
                                        hipMemcpyHostToDevice, c->stream));
        MOTIFS_HIP_CHECK(c->small.reserve(4096));
        int32_t* bad_dev = (int32_t*)c->small.p;
        MOTIFS_HIP_CHECK(hipMemsetAsync(bad_dev, 0, 4, c->stream));
        r = motifs_encode_dev(c, c->data_tmp.p, kind, S, m->L, (uint8_t*)c->codes.p, bad_dev);
        if (r) return r;
        e.reset();
        e.recording = false;
        e.keep_named = false;
        Graph gr(m, G);
        gr.Sone = make_onehot(m, (const uint8_t*)c->codes.p, (int)S);
        gr.codes = (const uint8_t*)c->codes.p;
        Tensor Draw = e.wrap(m->params, nullptr, m->nD, false), Fraw = e.wrap(m->params + m->nD, nullptr, m->nF, false);
        Scalars sc = prep_scalars(m, gr, false);
        Tensor Dp = e.norm4sq(Draw, 0.001f);
        Tensor Fp = e.norml2(Fraw, m->h * m->twoM, true);
        Graph::Bank bD = gr.bankD(Dp, 1), bF = gr.bankF(Fp, 1);
        Tensor ZY, X;
        admm_xyz(m, gr, sc, bD, bF, ZY, X);
        if (e.failed) {
            set_error("engine arena exhausted during code retrieval");
            return MOTIFS_ERR_UNSUPPORTED;
        }
        // counts -> host offsets -> ordered records
        MOTIFS_HIP_CHECK(c->pwmcnt.reserve((size_t)S * 12 + 64));
        int32_t* cnt_dev = (int32_t*)c->pwmcnt.p;
        int64_t* off_dev = (int64_t*)((char*)c->pwmcnt.p + (((size_t)S * 4 + 63) & ~(size_t)63));
        hipLaunchKernelGGL(k_count_pos, dim3((unsigned)S), dim3(64), 0, c->stream, X->v, m->l * m->K, cnt_dev);
        h_cnt.resize(S);
        h_off.resize(S);
        MOTIFS_HIP_CHECK(hipMemcpyAsync(h_cnt.data(), cnt_dev, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
        int32_t h_bad = 0;
        MOTIFS_HIP_CHECK(hipMemcpyAsync(&h_bad, bad_dev, 4, hipMemcpyDeviceToHost, c->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (h_bad) {
            set_error("data matrix has a column that is neither one-hot nor all-zero");
            return MOTIFS_ERR_NOT_ONEHOT;
        }
        int64_t run = 0;
        for (int64_t i = 0; i < S; i++) {
            h_off[i] = run;
            run += h_cnt[i];
        }
        if (total + run > cap) too_small = true;
        if (!too_small && run > 0) {
            MOTIFS_HIP_CHECK(hipMemcpyAsync(off_dev, h_off.data(), (size_t)S * 8, hipMemcpyHostToDevice, c->stream));
            MOTIFS_HIP_CHECK(c->hits_tmp.reserve((size_t)run * sizeof(motifs_code_rec)));
            hipLaunchKernelGGL(k_write_codes, dim3((unsigned)S), dim3(64), 0, c->stream, X->v, m->l, m->K, off_dev, s0,
                               (motifs_code_rec*)c->hits_tmp.p);
            MOTIFS_HIP_CHECK(hipMemcpyAsync(out + total, c->hits_tmp.p, (size_t)run * sizeof(motifs_code_rec),
                                            hipMemcpyDeviceToHost, c->stream));
            MOTIFS_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        total += run;
    }
    *n_out = total;
    if (too_small && !(cap == 0 && out == nullptr)) {
        set_error("code buffer too small: need %lld records, cap %lld", (long long)total, (long long)cap);
        return MOTIFS_ERR_BUFFER_TOO_SMALL;
    }
    return MOTIFS_OK;
}

// Measurement hook: a4 (the forward filter-bank scan of warmup_ZY, model.jl:171-173) on its own.
int motifs_model_time_filter_scan(motifs_model* m, const uint8_t* codes_dev, int n_groups, int reps, float* ms_out) {
    int r = check_model(m, "motifs_model_time_filter_scan");
    if (r) return r;
    if (!codes_dev || !ms_out || n_groups < 1 || reps < 1 || (int64_t)n_groups * m->B > 65535) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    Engine& e = m->eng;
    e.st = m->ctx->stream;
    e.reset();
    e.recording = false;
    e.keep_named = false;
    Graph gr(m, n_groups);
    gr.Sone = make_onehot(m, codes_dev, gr.S);
    gr.codes = codes_dev;
    Tensor Draw = e.wrap(m->params, nullptr, m->nD, false);
    Tensor Dp = e.norm4sq(Draw, 0.001f);
    Graph::Bank bD = gr.bankD(Dp, 1);
    (void)gr.anaD(gr.Sone, bD);                       // warm-up: the bank's fragment re-layout is built here
    const size_t mark = e.arena.off;
    hipEvent_t e0, e1;
    MOTIFS_HIP_CHECK(hipEventCreate(&e0));
    MOTIFS_HIP_CHECK(hipEventCreate(&e1));
    MOTIFS_HIP_CHECK(hipEventRecord(e0, e.st));
    for (int i = 0; i < reps; i++) {
        e.arena.off = mark;                           // the same output buffer every time
        (void)gr.anaD(gr.Sone, bD);
    }
    MOTIFS_HIP_CHECK(hipEventRecord(e1, e.st));
    MOTIFS_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    MOTIFS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e.failed) {
        set_error("engine arena exhausted during motifs_model_time_filter_scan");
        return MOTIFS_ERR_UNSUPPORTED;
    }
    *ms_out = ms / (float)reps;
    return MOTIFS_OK;
}

// Measurement hook: a7's dense contraction (warmup_X / update_X's conv(ZY, F, flipped=true), model.jl:214,251) on its own:
// the syntax-layer analysis GEMM of n_groups mini-batches, rows = reads x l, columns = K, reduction = h * 2M.
int motifs_model_time_syntax_conv(motifs_model* m, const uint8_t* codes_dev, int n_groups, int reps, float* ms_out) {
    int r = check_model(m, "motifs_model_time_syntax_conv");
    if (r) return r;
    if (!codes_dev || !ms_out || n_groups < 1 || reps < 1 || (int64_t)n_groups * m->B > 65535) return MOTIFS_ERR_INVALID;
    MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
    Engine& e = m->eng;
    e.st = m->ctx->stream;
    e.reset();
    e.recording = false;
    e.keep_named = false;
    Graph gr(m, n_groups);
    gr.Sone = make_onehot(m, codes_dev, gr.S);
    gr.codes = codes_dev;
    Tensor Draw = e.wrap(m->params, nullptr, m->nD, false), Fraw = e.wrap(m->params + m->nD, nullptr, m->nF, false);
    Tensor Dp = e.norm4sq(Draw, 0.001f);
    Tensor Fp = e.norml2(Fraw, m->h * m->twoM, true);
    Graph::Bank bD = gr.bankD(Dp, 1), bF = gr.bankF(Fp, 1);
    Tensor ZY = gr.anaD(gr.Sone, bD);                 // an image of the right shape and scale: [S][c][2M]
    (void)gr.anaF(ZY, bF);                            // warm-up: the bank's fragment re-layout is built here
    const size_t mark = e.arena.off;
    hipEvent_t e0, e1;
    MOTIFS_HIP_CHECK(hipEventCreate(&e0));
    MOTIFS_HIP_CHECK(hipEventCreate(&e1));
    MOTIFS_HIP_CHECK(hipEventRecord(e0, e.st));
    for (int i = 0; i < reps; i++) {
        e.arena.off = mark;
        (void)gr.anaF(ZY, bF);
    }
    MOTIFS_HIP_CHECK(hipEventRecord(e1, e.st));
    MOTIFS_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    MOTIFS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e.failed) {
        set_error("engine arena exhausted during motifs_model_time_syntax_conv");
        return MOTIFS_ERR_UNSUPPORTED;
    }
    *ms_out = ms / (float)reps;
    return MOTIFS_OK;
}

// Test hook: copy a named intermediate of the last motifs_model_loss_grad_dev(keep_intermediates=1) call.
int motifs_model_dump(motifs_model* m, const char* name, float* out, int64_t cap, int64_t* n) {
    int r = check_model(m, "motifs_model_dump");
    if (r) return r;
    auto it = m->eng.named.find(name ? name : "");
    if (it == m->eng.named.end()) {
        set_error("motifs_model_dump: no intermediate named '%s'", name ? name : "");
        return MOTIFS_ERR_INVALID;
    }
    if (n) *n = (int64_t)it->second->n;
    if (out) {
        if ((int64_t)it->second->n > cap) return MOTIFS_ERR_BUFFER_TOO_SMALL;
        MOTIFS_HIP_CHECK(hipSetDevice(m->ctx->device));
        MOTIFS_HIP_CHECK(hipMemcpyAsync(out, it->second->v, it->second->n * 4, hipMemcpyDeviceToHost, m->ctx->stream));
        MOTIFS_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    }
    return MOTIFS_OK;
}

}  // extern "C"
