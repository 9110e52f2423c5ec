// cdl_engine.h — a small reverse-mode engine for the unrolled-ADMM sparse-coding
// graph of src/model.jl, batched over G independent mini-batches ("groups" of
// batch_size sequences, the reference's unit of coupling: median mask
// model.jl:194-204, batch sums :285/:300, loss 1/B :311).
//
// Everything is float32 on the device.  Tensors are flat arrays carved from one
// arena; every forward primitive pushes the closure of its vector-Jacobian
// product on a tape, and backward() replays the tape in reverse — the role
// Zygote plays in the reference (train.jl:42-44).  The reference's stride-1
// convolutions + masks are stored in their compact stride-4 form:
//   img   [S][c][2M]   code images ZY, FX, duals (row = aligned position, col = filter/strand)
//   x     [S][l][K]    syntax codes
//   sig   [S][4L]      one-hot sequences, reconstructions, residuals
//   D     [g][M][4fl]  filters in the reference layout (model.jl:67-90: D[(p-1)*4+a, 1, m])
//   F     [g][K][2M][h] syntax filters in the reference layout (F[i, j, 1, k])
// with g = 1 (shared parameters) or g = G (after the per-batch D/F updates).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <map>
#include <tuple>
#include <string>
#include <utility>
#include <vector>

namespace motifs {

struct TNode {
    float* v = nullptr;   // values
    float* g = nullptr;   // gradient (allocated on first use, zero-initialised)
    size_t n = 0;
    bool needs_grad = false;
    const float* gmask = nullptr;   // if set: the gradient is only ever consumed where gmask != 0
    // cached non-zero lists (built on first use): of the values, and of gmask
    int* nz_cnt = nullptr;
    uint2* nz_ent = nullptr;
    int* gm_cnt = nullptr;
    uint2* gm_ent = nullptr;
    // set on the first output of zy_step2 by lin3_zy: the combination FX + b*[zy >= thr]*zy + abn formed right after the
    // step.  Its VJP rides in the step's own backward kernel (the two are neighbours on the tape).
    TNode* fl_img = nullptr;
    TNode* fl_x = nullptr;
    float fl_b = 0.0f;
    const float* fl_thr = nullptr;
    int fl_groups = 1;
};

struct NzView {
    const int* cnt;       // [S]
    const uint2* ent;     // [S][cap]: {flat index p*K + k, value bits}
    int cap;
};
typedef TNode* Tensor;    // nodes are owned by the engine and die at reset()

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    float* alloc(size_t n_floats) {
        size_t bytes = (n_floats * 4 + 255) & ~(size_t)255;
        if (off + bytes > cap) return nullptr;
        float* p = (float*)(base + off);
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    void reset() { off = 0; }
};

// Toeplitz GEMM geometry: C[s][p][n] = sum_q Aw(s,p,q) * B[grp(s)][q][n],
// Aw(s,p,q) = A[s*lda + e], e = a0 + p*sa + q, zero when e is outside [0, amax).
struct ToepGeom {
    int S, P, Q, N;        // sequences, output rows, reduction length, output channels
    int sa, a0, amax;      // row stride, window offset, valid flat range of A per sequence
    int64_t lda, ldc;      // elements per sequence of A and C
    int B;                 // sequences per group
    int64_t ldb;           // elements per group of Bm (0 = shared)
};

// Syntax-layer geometry for the sparse forms: codes T [S][l][K], images [S][c][W], filters of height h.
struct SpDims {
    int S, B, l, K, c, W, h;
    int64_t ldf;   // filter elements per group (h*W*K), 0 = shared
};

void dev_zero(hipStream_t st, float* p, size_t n);                    // n floats of zero, as a kernel
void dev_copy(hipStream_t st, float* dst, const float* src, size_t n);   // n floats device to device, as a kernel

struct Engine {
    hipStream_t st = nullptr;
    Arena arena;
    std::vector<std::function<void()>> tape;
    std::vector<TNode*> nodes;
    bool recording = true;
    bool failed = false;           // arena exhausted
    bool wgrad_reads_attr_set = false;   // the dynamic-LDS attribute of k_sp_wgrad_*_reads
    bool onehot_attr_set = false;  // k_onehot_bank_scan's dynamic-LDS attribute has been set on this engine's device
    // re-laid-out copies of filter banks (fragment order, flipped, transposed), keyed by source and kind: a bank serves
    // many calls of a pass and its contents are final when the first of them runs
    struct RelayoutKey {
        const void* src;
        int kind, d0, d1, d2;      // which re-layout and the dimensions it was made for
        size_t n;                  // its size in floats (the group count is in here)
        bool operator<(const RelayoutKey& o) const {
            return std::tie(src, kind, d0, d1, d2, n) < std::tie(o.src, o.kind, o.d0, o.d1, o.d2, o.n);
        }
    };
    std::map<RelayoutKey, float*> derived;
    std::map<const void*, uint32_t*> absmax_of;   // tensor -> bits of its largest magnitude (device), where a producer kernel keeps it (f16x3 GEMM scale)
    float* relayout(const float* src, int kind, int d0, int d1, int d2, size_t n, bool& fresh);   // fresh: the caller fills it
    float* zpool = nullptr;        // current pre-zeroed chunk (dies with the arena at reset())
    size_t zleft = 0;
    std::map<std::string, Tensor> named;
    bool keep_named = false;
    // measurement hook (bench.py train.roofline): when set, an event pair is recorded around every launch of the step's largest kernel
    // (k_zy_step2_bwd); the owner of the probe hands out the events and collects the pairs
    struct Probe {
        std::function<hipEvent_t()> get;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    };
    Probe* probe = nullptr;

    ~Engine() { reset(); }
    void reset();                  // drop every node and the tape, rewind the arena
    Tensor make(size_t n, bool needs_grad);
    Tensor wrap(float* v, float* g, size_t n, bool needs_grad);   // external storage (parameters)
    float* grad(Tensor t);         // allocate + zero on first use
    float* zeros(size_t n);        // n zeroed floats: small requests are carved from chunks zeroed with one fill each
    float* grad_first(Tensor t, int& acc);   // allocate on first use without the zero fill (acc = 0: overwrite)
    NzView nz_build(const float* data, int S, int n_per);   // non-zero list of [S][n_per] values (memory order)
    NzView nz_of(Tensor t, int S);                           // cached list of t's values
    NzView nz_of_mask(Tensor t, int S);                      // cached list of t->gmask
    void note(const char* name, Tensor t) {
        if (keep_named) named[name] = t;
    }
    void backward();

    // ---- primitives (each records its VJP) ----
    Tensor shrink(Tensor x, float a, float c);                            // relu(a*x + c)
    Tensor lin(Tensor x, float a, Tensor y, float b, float cst);          // a*x + b*y(bcast modulo y.n) + cst
    // a*x + b*(ymask .* y) + c*z in one pass (z, ymask optional; ymask is a constant 0/1 mask)
    Tensor lin3(Tensor x, float a, Tensor y, float b, Tensor z, float c, const float* ythr = nullptr, int groups = 1);   // ythr: y counts where y >= ythr[group]
    // relu(ZY - lst*(g1 + pen*(ZY - FX - ab)) - ls*lst): the ISTA step of update_ZY fused (ab optional; scalars are 1-element tensors)
    // hist0 (optional): [groups][2*2048] zeroed counters; the kernel adds the top-digit histogram of the new codes (median pass 0)
    Tensor zy_step(Tensor ZY, Tensor g1, Tensor FX, Tensor ab, Tensor pen, Tensor lst, Tensor ls, uint32_t* hist0 = nullptr, int groups = 1);
    // FX + b*[zy >= thr[group]]*zy + abn for the outputs (zy, abn) of zy_step2(.., FX, ..): no tape entry of its own
    Tensor lin3_zy(Tensor FX, Tensor zy, float b, Tensor abn, const float* thr, int groups);
    // relu((Fc - sg*Fgrad*kst) - kst*ks) with n outputs; Fc is broadcast over groups if smaller, Fgrad may be null (zero)
    Tensor f_step(Tensor Fc, Tensor Fgrad, float sg, Tensor kst, Tensor ks, size_t n, int sw_h = 0, int sw_n2 = 0, int sw_k = 0);
    // norml2(f_step(...), seg) in one kernel per direction (Fc of the output's size; else the two launches)
    Tensor f_step_norm(Tensor Fc, Tensor Fgrad, float sg, Tensor kst, Tensor ks, size_t n, int seg, int sw_h = 0, int sw_n2 = 0, int sw_k = 0);
    Tensor x_step(Tensor X, Tensor xg, Tensor ost);   // X - ost * xg (update_X before the projection), VJP in one pass
    // the same with the dual update folded in: abn = FX - ZY + abp (abp optional), then the step with abn; returns {out, abn}
    std::pair<Tensor, Tensor> zy_step2(Tensor ZY, Tensor g1, Tensor FX, Tensor abp, Tensor pen, Tensor lst, Tensor ls, uint32_t* hist0 = nullptr,
                                       int groups = 1);
    Tensor mul(Tensor x, Tensor y);                                       // x .* y (y broadcast modulo y.n)
    Tensor relu(Tensor x);
    Tensor maskmul(Tensor x, const float* mask, float c);                 // c * mask .* x, mask constant
    Tensor thrmul(Tensor x, const float* thr, int groups, float c);       // c * [x >= thr[group]] .* x, the selection constant
    Tensor expo(Tensor x);
    Tensor norm4(Tensor x);                                               // x / sum over each 4 consecutive
    Tensor norml2(Tensor x, int seg, bool squared = false);               // per segment x / ||x||_2 (squared: of x .* x)
    Tensor norm4sq(Tensor x, float eps);                                  // norm4(x .* x + eps)
    Tensor sumsq_groups(Tensor x, float coef, int groups);                // [groups]: coef * sum x^2 per group
    // coef * sum_group (x + b*[y >= thr[group]]*y)^2 without writing the residual
    Tensor resid_sumsq_groups(Tensor x, Tensor y, float b, const float* thr, float coef, int groups, Tensor into = nullptr);
    Tensor toep(Tensor A, Tensor Bm, const ToepGeom& gm, const float* y = nullptr, float yb = 0.0f);   // Toeplitz GEMM (+ yb * y: tall forms only, see toep_plus)
    // project_X(X - ost * xg) (xg == null: project_X(X)), model.jl:253 + :181-192, with the entry lists of the result and of its mask
    Tensor x_project(Tensor X, Tensor xg, Tensor ost, int S, int q, float scale = 1.0f);   // scale (xg == null only): project_X(scale * X)
    std::pair<Tensor, Tensor> bankD(Tensor D, int g, int M, int fl);       // (analysis form, flipped synthesis form) of a D bank + their fragment re-layouts
    void prelayout_an(const float* an, int g, int M, int fl);               // a bank given in analysis form: its flip and fragment re-layouts, one launch
    std::pair<Tensor, Tensor> bankF(Tensor F, int g, int K, int N2, int h);   // (analysis form [h][2M][K], flipped synthesis form [h][K][2M]) of an F bank
    // norm4(exp(-mu * Dgrad) .* Dc), model.jl:285-289; M > 0: Dgrad is the expanded gradient [g][4 fl][2M], collapsed on the way in
    Tensor d_step(Tensor Dgrad, Tensor mu, Tensor Dc, int g = 1, int M = 0, int fl = 0);
    Tensor toep_plus(Tensor A, Tensor Bm, const ToepGeom& gm, Tensor y, float b);   // toep(A, Bm) + b * y, y a constant image
    // the same product when A is the one-hot image of `codes` (rows of `pitch` bytes, 0..3, 4 = all-zero column) and the windows
    // advance by whole positions (gm.sa == 4): the forward is fl gathered bank rows per output row, no GEMM; the backward is toep's
    Tensor toep_onehot(Tensor A, const uint8_t* codes, int pitch, Tensor Bm, const ToepGeom& gm);
    Tensor wgrad(Tensor A, Tensor C, const ToepGeom& gm);                 // [G][Q][N] = sum_{s,p} Aw * C
    // syntax layer with sparse codes (X keeps ~q entries per read; gradients into it are masked):
    Tensor sp_syn(Tensor T, Tensor FAf, Tensor Fk, const SpDims& d);      // FX = sum(conv(X,F,pad,groups=K),dims=3)
    Tensor ana_sp(Tensor img, Tensor FA, Tensor FAf, const SpDims& d, const ToepGeom& gm);   // conv(img,F,flipped)
    Tensor wgrad_sp(Tensor img, Tensor T, const SpDims& d);               // [G][h][W][K] = sum_{s,p} img[p+i][j] T[p][k]
    Tensor expandD(Tensor D, int g, int M, int fl);                       // [g][M][4fl] -> [g][fl*4][2M]
    Tensor collapseD(Tensor GA, int g, int M, int fl);                    // adjoint of expandD
    Tensor swap02(Tensor x, int g, int d0, int d1, int d2);               // per group [d0][d1][d2] -> [d2][d1][d0]
    Tensor flipT(Tensor Bm, int g, int H, int W, int N);                  // [H][W][N] -> [H][N][W], H reversed
};

// selections (constants in the backward: @ignore, model.jl:190, :208)
void topq_mask(hipStream_t st, const float* X, float* bitmat, int S, int n_per_seq, int q);
size_t median_workspace_bytes(int G);
void median_threshold(hipStream_t st, const float* ZY, float* thr, int G, int n_per_group, void* workspace, bool have_pass0 = false);
uint32_t* median_hist_ptr(void* workspace, int G);   // the [G][2*2048] counters inside a (zeroed) workspace
void onehot_from_codes(hipStream_t st, const uint8_t* codes, int pitch, float* S, int nseq, int L);
void adabelief_step(hipStream_t st, float* x, float* m, float* s, const float* grad, size_t n, float gscale, float eta,
                    float b1, float b2, float eps, double b1p, double b2p);   // b1p, b2p: running powers of beta, kept in double

}  // namespace motifs
