"""model_oracle.py — CPU restatement of MOTIFs.jl's unrolled-ADMM convolutional
sparse coding (src/model.jl), its training step (src/train.jl) and code retrieval
(src/inference/_1_code_retrieval.jl).

TEST INFRASTRUCTURE ONLY — never imported by the product package.

PARITY UNPINNED: the reference has no tests, fixtures or golden vectors
(test/runtests.jl:4-6) and Julia is not installed here; the conv arithmetic of the
reference lives in un-vendored NNlib 0.9.8 / cuDNN (Manifest.toml:691-707).  This
file restates model.jl function by function on top of torch's CPU conv1d/conv2d,
using NNlib's published semantics: output length W + 2*pad - k + 1; `flipped=true`
is cross-correlation, the default reverses the kernel; `groups=G` pairs output
block g with input block g.  Gradients come from torch autograd, the role Zygote
plays in the reference (train.jl:42-44).  It is pinned by identities and
hand-computable cases in tests/test_oracle_model.py.

Layout convention: every tensor here has its dims REVERSED with respect to the
Julia array it restates, so its C-order memory is byte-for-byte the Julia
column-major array and Julia `reshape` is a plain `.reshape` here:
    Julia (W, C, B)      <->  torch (B, C, W)        (= torch conv1d layout)
    Julia (W, H, C, B)   <->  torch (B, C, H, W)     (torch's "W" is Julia dim 1)
Comments cite src/model.jl lines unless another file is named.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as Fn

float_type = torch.float32  # MOTIFs.jl:14


@dataclass
class Hyperparam:           # model.jl:1-14
    filter_len: int = 8
    M: int = 50
    h: int = 12
    K: int = 24
    q: int = 32
    batch_size: int = 6
    num_pass_xyz: int = 6
    num_pass_df: int = 3
    magnifying_factor: float = 10.0
    gamma: float = 0.1

    @property
    def f_len(self):
        return self.filter_len * 4

    @property
    def twoM(self):
        return 2 * self.M


@dataclass
class LengthInfo:           # model.jl:16-37 (data.L = sequence length in bp)
    L: int
    C: int
    c: int
    l: int
    MB: int
    KB: int
    CS_vlen: int

    @staticmethod
    def make(hp, data_L):
        L = 4 * data_L
        C = L - hp.f_len + 1
        c = data_L - hp.filter_len + 1
        return LengthInfo(L, C, c, c - hp.h + 1, hp.M * hp.batch_size, hp.K * hp.batch_size, C + L - 1)


class Projectors:           # model.jl:39-65
    def __init__(self, hp, ln, dtype=float_type):
        md = torch.zeros((ln.C + ln.L - 1, hp.f_len), dtype=dtype)          # Julia (f_len, C+L-1)
        for j in range(hp.f_len):                                            # [:, C:C+f_len-1] = I   (:47)
            md[ln.C - 1 + j, j] = 1
        self.mapdrange = md
        mc = torch.zeros((ln.c, ln.C), dtype=dtype)                          # Julia (C, c); rows 1:4:end = I (:51)
        for i in range(ln.c):
            mc[i, 4 * i] = 1
        self.mapclarge = mc
        zm = torch.zeros((hp.batch_size, hp.M, ln.C), dtype=dtype)           # Julia (C, M, B) (:54-55)
        zm[:, :, 0::4] = 1
        self.z_mask_n = zm
        self.pseudocount_matrix = torch.full((hp.M, hp.filter_len, 4), 0.001, dtype=dtype)  # (:56)


# ---- NNlib.conv restated on reversed-dim tensors -------------------------------------------------

def conv1(x, w, pad=0, flipped=False, groups=1):
    """NNlib.conv for x Julia (W, Cin, B), w Julia (k, Cin/groups, Cout)."""
    if not flipped:
        w = w.flip(-1)
    return Fn.conv1d(x, w, padding=pad, groups=groups)


def conv2(x, w, pad=(0, 0), flipped=False, groups=1):
    """NNlib.conv for x Julia (W, H, Cin, B), w Julia (k1, k2, Cin/groups, Cout); pad = (pad_dim1, pad_dim2)."""
    if not flipped:
        w = w.flip(-1, -2)
    return Fn.conv2d(x, w, padding=(pad[1], pad[0]), groups=groups)


# ---- learnable state ------------------------------------------------------------------------------

PARAM_VECS = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize",
              "penalty_xyz", "mu"]


def vec_sizes(hp):
    x, d = hp.num_pass_xyz, hp.num_pass_df
    return {"lambda_sparsity": x, "kappa_sparsity": d, "lambda_stepsize": x, "omega_stepsize": x,
            "kappa_stepsize": d, "penalty_xyz": x, "mu": d}


class UCDL:
    """The `ucdl` struct (model.jl:67-137).  Arrays are trained (Flux.params, train.jl:34); the three
    *_warmup scalars are plain Float32 fields and are not."""

    def __init__(self, hp, rng=None, dtype=float_type, eta1=0.05):
        rng = rng or np.random.default_rng(0)
        # randomly_initialize_filters (MOTIFs.jl:17-33): per position, sorted uniforms -> spacings (a simplex point)
        arr = np.zeros((hp.M, hp.filter_len, 5))
        for i in range(hp.filter_len):
            for j in range(hp.M):
                arr[j, i, 1:4] = np.sort(rng.random(3))
        arr[:, :, 4] = 1
        D = np.sqrt(np.diff(arr, axis=2).reshape(hp.M, 1, hp.f_len))        # (:88)  Julia (f_len,1,M)
        self.D = torch.tensor(D, dtype=dtype)
        self.F = torch.tensor(np.abs(0.1 * rng.standard_normal((hp.K, 1, hp.twoM, hp.h))), dtype=dtype)  # (:90)
        self.lambda_sparsity_warmup = float(eta1 * rng.random())
        sz = vec_sizes(hp)
        self.lambda_sparsity = torch.tensor(eta1 * rng.random(sz["lambda_sparsity"]), dtype=dtype)
        self.kappa_sparsity = torch.tensor(eta1 * rng.random(sz["kappa_sparsity"]), dtype=dtype)
        self.lambda_stepsize_warmup = float(eta1 * rng.random())
        self.omega_stepsize_warmup = float(eta1 * rng.random())
        self.lambda_stepsize = torch.tensor(eta1 * rng.random(sz["lambda_stepsize"]), dtype=dtype)
        self.omega_stepsize = torch.tensor(eta1 * rng.random(sz["omega_stepsize"]), dtype=dtype)
        self.kappa_stepsize = torch.tensor(eta1 * rng.random(sz["kappa_stepsize"]), dtype=dtype)
        self.penalty_xyz = torch.tensor(eta1 * rng.random(sz["penalty_xyz"]), dtype=dtype)
        self.mu = torch.tensor(eta1 * rng.random(sz["mu"]), dtype=dtype)

    def params(self):
        """Flux.params(cdl) order = field order of the struct (model.jl:68-82)."""
        return [self.lambda_sparsity, self.kappa_sparsity, self.lambda_stepsize, self.omega_stepsize,
                self.kappa_stepsize, self.D, self.F, self.penalty_xyz, self.mu]

    def to(self, dtype):
        for name in PARAM_VECS + ["D", "F"]:
            setattr(self, name, getattr(self, name).detach().to(dtype))
        return self

    def requires_grad_(self, flag=True):
        for name in PARAM_VECS + ["D", "F"]:
            setattr(self, name, getattr(self, name).detach().requires_grad_(flag))
        return self


def prep_filters(D, hp, projs):                                               # :139-146
    D_init = D ** 2
    D_init_r = D_init.reshape(hp.M, hp.filter_len, 4)
    D_init_r = D_init_r + projs.pseudocount_matrix.to(D.dtype)
    D_init_r = D_init_r / D_init_r.sum(dim=-1, keepdim=True)
    return D_init_r.reshape(hp.M, 1, hp.f_len)


def prep_syntax_filters(F):                                                    # :148-151
    F = F ** 2
    return F / torch.sqrt((F ** 2).sum(dim=(-1, -2), keepdim=True))


def prep_params(u, hp, projs):                                                 # :153-169
    return (u.lambda_sparsity_warmup ** 2, u.lambda_sparsity ** 2, u.kappa_sparsity ** 2,
            u.lambda_stepsize_warmup ** 2, u.omega_stepsize_warmup ** 2, u.lambda_stepsize ** 2,
            u.omega_stepsize ** 2, u.kappa_stepsize ** 2, u.penalty_xyz ** 2, u.mu ** 2,
            prep_filters(u.D, hp, projs), prep_syntax_filters(u.F))


def warmup_ZY(S, D, lambda_stepsize_warmup, lambda_sparsity_warmup, projs):    # :171-179
    DtS = conv1(S, D, pad=0, flipped=True)
    DS = conv1(S, D, pad=0)
    Z_update = lambda_stepsize_warmup * DtS - lambda_sparsity_warmup * lambda_stepsize_warmup
    Y_update = lambda_stepsize_warmup * DS - lambda_sparsity_warmup * lambda_stepsize_warmup
    zm = projs.z_mask_n.to(S.dtype)
    return torch.relu(zm * Z_update), torch.relu(zm * Y_update)


# The reference computes in Float32 (float_type, _0_const.jl:1): its two data-dependent selections - the q-th largest code of a read
# (partialsort, :183-184) and the median of the non-zero ZY entries (Statistics.median = middle(a, b) = a/2 + b/2 for an even count,
# :198-199) - are taken on Float32 values.  A float64 run of this file takes them on float64 values, and that is a different FUNCTION
# wherever the two middle values are less than one Float32 ulp apart: a/2 + b/2 then rounds to a (ties to even) and `ZY .>= med` keeps
# one entry more.  At BASELINE configs[1] (453 600 non-zero codes of a mini-batch inside [0, 1.5e-2]: neighbours ~0.6 ulp apart at the
# median) that happens in about one mini-batch of six.  DECISIONS_F32 = True rounds the operands of the two selections to float32 first
# (everything else stays in the run's dtype): float64 arithmetic on the function the Float32 reference evaluates.
DECISIONS_F32 = False


def _dec(x):
    return x.float() if DECISIONS_F32 else x


def generate_bitmat(X, hp):                                                    # :181-187
    with torch.no_grad():
        B = X.shape[0]
        Xd = _dec(X)
        Xr = Xd.reshape(B, -1)                                                 # (l*K, B) columns
        vals = torch.topk(Xr, hp.q, dim=1).values[:, hp.q - 1]                # partialsort(col, q, rev=true)
        return (Xd >= vals.reshape(B, 1, 1, 1)).to(X.dtype)


def project_X(X, hp):                                                          # :189-192
    return X * generate_bitmat(X, hp)


def julia_median(v):
    """Statistics.median of a vector: middle value, or middle(a, b) = a/2 + b/2 for an even count."""
    s, _ = torch.sort(v)
    n = s.numel()
    if n % 2 == 1:
        return s[n // 2]
    return s[n // 2 - 1] / 2 + s[n // 2] / 2


def create_ZY_mask(ZY):                                                        # :194-204
    with torch.no_grad():
        ZYd = _dec(ZY)
        Z_nz = ZYd[ZYd > 0]
        if Z_nz.numel() == 0:
            return None
        med = julia_median(Z_nz)
        return (ZYd >= med).to(ZY.dtype)


def cat_ZY(Z, Y, hp, ln):                                                      # :206-210
    ZY = torch.cat((Z[..., 0::4], Y[..., 0::4]), dim=-2).reshape(hp.batch_size, 1, hp.twoM, ln.c)
    mask = create_ZY_mask(ZY)
    return hp.magnifying_factor * ZY if mask is None else hp.magnifying_factor * (mask * ZY)


def warmup_X(F, Z, Y, omega_stepsize_warmup, hp, ln):                          # :212-216
    ZY = cat_ZY(Z, Y, hp, ln)
    X_updated = omega_stepsize_warmup * conv2(ZY, F, pad=(0, 0), flipped=True)
    return project_X(X_updated, hp)


# torch's conv2d cannot run the literal padded grouped form of syn_FX at BASELINE configs[3] shape (it asks for a 290 GB
# work buffer) and is slow on F_gradient's 144-group form.  FAST_SYNTAX = True uses the same sums written directly:
# FX = conv_transpose2d(X, F) (sum_k sum_p X[b,k,p] F[k,j,i-p]) and F_grad = einsum over the h windows of the residual.
# Equality with the literal forms: tests/test_oracle_model.py::test_needed_lag_update_D_equals_the_literal_one.
FAST_SYNTAX = False


def syn_FX(X, F, hp):
    """sum(convolution(X, F, pad=(h-1, twoM-1), groups=K), dims=3)  (:229, :263, :294, :316, :370)."""
    if FAST_SYNTAX:
        return Fn.conv_transpose2d(X, F)
    return conv2(X, F, pad=(hp.h - 1, hp.twoM - 1), groups=hp.K).sum(dim=1, keepdim=True)


def return_left_right_FX(FX, hp, ln):                                          # :218-222
    left = FX[:, :, :hp.M, :].reshape(hp.batch_size, hp.M, ln.c)
    right = FX[:, :, hp.M:, :].reshape(hp.batch_size, hp.M, ln.c)
    return left, right


def warmup_XYZ(S, D, F, lsw, lspw, osw, hp, ln, projs):                        # :224-232
    Z, Y = warmup_ZY(S, D, lsw, lspw, projs)
    X = warmup_X(F, Z, Y, osw, hp, ln)
    FX = syn_FX(X, F, hp)
    left_FX, right_FX = return_left_right_FX(FX, hp, ln)
    return Z, Y, X, FX, left_FX, right_FX


def grouped_D(D, hp):
    """D Julia (f_len, 1, M) used as the weight of a groups=M conv: Julia (k, Cin/groups=1, Cout=M).
    On reversed dims that is torch (M, 1, f_len) — the very same memory."""
    return D.reshape(hp.M, 1, hp.f_len)


def syn_ZD(Z, D, hp, flipped=False):
    """convolution(Z, D, pad=f_len-1, groups=M[, flipped=true])  (:238, :276-277, :313-314)."""
    return conv1(Z, grouped_D(D, hp), pad=hp.f_len - 1, groups=hp.M, flipped=flipped)


def update_ZY(S, Z, Y, D, left_FX, right_FX, alpha, beta, lambda_sparsity, lambda_stepsize, penalty_xyz, hp, projs,
              num_pass):                                                       # :237-245
    ZD, YD = syn_ZD(Z, D, hp), syn_ZD(Y, D, hp, flipped=True)
    diff = (ZD + YD).sum(dim=1, keepdim=True) - S
    mc = projs.mapclarge.to(S.dtype)
    z_grad = conv1(diff, D, pad=0, flipped=True) + penalty_xyz[num_pass] * (Z - (left_FX + alpha) @ mc)
    y_grad = conv1(diff, D, pad=0) + penalty_xyz[num_pass] * (Y - (right_FX + beta) @ mc)
    Z_updated = Z - lambda_stepsize[num_pass] * z_grad - lambda_sparsity[num_pass] * lambda_stepsize[num_pass]
    Y_updated = Y - lambda_stepsize[num_pass] * y_grad - lambda_sparsity[num_pass] * lambda_stepsize[num_pass]
    zm = projs.z_mask_n.to(S.dtype)
    return torch.relu(zm * Z_updated), torch.relu(zm * Y_updated)


def update_X(FX, Z, Y, X, F, alpha, beta, omega_stepsize, hp, ln, num_pass):   # :247-254
    alpha_beta = torch.cat((alpha, beta), dim=-2).reshape(hp.batch_size, 1, hp.twoM, ln.c)
    ZY = cat_ZY(Z, Y, hp, ln)
    diff = FX.sum(dim=1, keepdim=True) - (ZY - alpha_beta)
    x_grad = conv2(diff, F, pad=(0, 0), flipped=True)
    X_updated = X - omega_stepsize[num_pass] * x_grad
    return project_X(X_updated, hp)


def one_forward_step_XYZ(S, Z, Y, D, X, F, left_FX, right_FX, FX, alpha, beta, lambda_sparsity, lambda_stepsize,
                         omega_stepsize, penalty_xyz, hp, ln, projs, num_pass):  # :256-268
    Z, Y = update_ZY(S, Z, Y, D, left_FX, right_FX, alpha, beta, lambda_sparsity, lambda_stepsize, penalty_xyz, hp,
                     projs, num_pass)
    X = update_X(FX, Z, Y, X, F, alpha, beta, omega_stepsize, hp, ln, num_pass)
    FX = syn_FX(X, F, hp)
    left_FX, right_FX = return_left_right_FX(FX, hp, ln)
    alpha = alpha + left_FX - Z[..., 0::4]
    beta = beta + right_FX - Y[..., 0::4]
    return Z, Y, X, FX, left_FX, right_FX, alpha, beta


def conv_code_diff(code, diff, hp, ln):                                        # :270-273
    up = diff.repeat_interleave(hp.M, dim=1)                                   # upsample_nearest(diff, (1, M, 1))
    x = up.reshape(1, ln.MB, ln.L)                                             # Julia (L, MB, 1)
    w = code.reshape(ln.MB, 1, ln.C)                                           # Julia (C, 1, MB)
    out = conv1(x, w, pad=ln.C - 1, groups=ln.MB, flipped=True)                # Julia (CS_vlen, MB, 1)
    return out.reshape(hp.batch_size, hp.M, ln.CS_vlen)


# The six conv_code_diff calls of update_D compute C + L - 1 lags each and mapdrange keeps f_len of them (3 % at
# configs[1]); torch's grouped conv1d with groups = M*B makes them ~95 % of the oracle's time.  NEEDED_LAGS = True forms
# only the kept lags (D_grad[m, j] = sum_b sum_p code[b, m, 4p] * sig[b, 4p + j], reversed in j for the reverse strand) -
# the identity tests/test_oracle_model.py::test_adjointness_and_needed_lags proves against the literal form.  It exists so
# that a golden mini-batch at BASELINE configs[3] shape can be generated in minutes; everything else stays literal.
NEEDED_LAGS = False


def _needed_lag_corr(code, sig, hp, ln, reverse):
    win = sig[:, 0, :].unfold(1, hp.f_len, 4)                                  # (B, c, f_len): sig[b, 4p + j]
    out = torch.einsum("bmp,bpj->mj", code[..., 0::4], win)
    return out.flip(-1) if reverse else out


def update_D(S, Z, Y, D, mu, hp, ln, projs, num_pass):                         # :275-290
    sumZD = syn_ZD(Z, D, hp).sum(dim=1, keepdim=True)
    sumYRD = syn_ZD(Y, D, hp, flipped=True).sum(dim=1, keepdim=True)
    if NEEDED_LAGS:
        D_grad = sum(_needed_lag_corr(Z, sg, hp, ln, False) + _needed_lag_corr(Y, sg, hp, ln, True) for sg in (sumZD, sumYRD, S))
        Breg_num = (D * torch.exp(-mu[num_pass] * D_grad.reshape(hp.M, 1, hp.f_len))).reshape(hp.M, 1, hp.filter_len, 4)
        return (Breg_num / Breg_num.sum(dim=-1, keepdim=True)).reshape(hp.M, 1, hp.f_len)
    ZtsumZD = conv_code_diff(Z, sumZD, hp, ln)
    YtsumZD = conv_code_diff(Y, sumZD, hp, ln)
    ZtsumYRD = conv_code_diff(Z, sumYRD, hp, ln)
    YtsumYRD = conv_code_diff(Y, sumYRD, hp, ln)
    ZtS = conv_code_diff(Z, S, hp, ln)
    YtS = conv_code_diff(Y, S, hp, ln)
    tot = (ZtsumZD + ZtsumYRD + ZtS + (YtsumZD + YtsumYRD + YtS).flip(-1)).sum(dim=0, keepdim=True)  # sum over batch
    D_grad = (tot.reshape(hp.M, ln.C + ln.L - 1) @ projs.mapdrange.to(S.dtype)).reshape(hp.M, 1, hp.f_len)
    Breg_num = (D * torch.exp(-mu[num_pass] * D_grad)).reshape(hp.M, 1, hp.filter_len, 4)
    return (Breg_num / Breg_num.sum(dim=-1, keepdim=True)).reshape(hp.M, 1, hp.f_len)


def F_gradient(ZY, X, F, hp, ln, theta):                                       # :292-302
    if FAST_SYNTAX:
        R = (syn_FX(X, F, hp) - (ZY + theta))[:, 0]                             # (B, twoM, c)
        return torch.einsum("bjip,bkp->kji", R.unfold(2, ln.l, 1), X[:, :, 0, :]).reshape(hp.K, 1, hp.twoM, hp.h)
    diff_X_upsampled = (syn_FX(X, F, hp) - (ZY + theta)).repeat_interleave(hp.K, dim=1)
    diff_r = diff_X_upsampled.reshape(1, hp.K * hp.batch_size, hp.twoM, ln.c)
    X_r = X.reshape(hp.K * hp.batch_size, 1, 1, ln.l)
    conv_diff_X = conv2(diff_r, X_r, pad=(0, 0), flipped=True, groups=ln.KB)
    F_conv = conv_diff_X.reshape(hp.batch_size, hp.K, hp.twoM, hp.h)
    return F_conv.sum(dim=0, keepdim=True).reshape(hp.K, 1, hp.twoM, hp.h)


def update_F(ZY, X, F, hp, ln, theta, kappa_stepsize, kappa_sparsity, num_pass):  # :304-308
    F_grad = F_gradient(ZY, X, F, hp, ln, theta)
    F_updated = torch.relu(F - kappa_stepsize[num_pass] * F_grad - kappa_stepsize[num_pass] * kappa_sparsity[num_pass])
    return F_updated / torch.sqrt((F_updated ** 2).sum(dim=(-1, -2), keepdim=True))


def loss(S, Z, Y, X, D, ZY, F, hp):                                            # :310-325
    nf = 1.0 / hp.batch_size
    DZ = syn_ZD(Z, D, hp).sum(dim=1, keepdim=True)
    DY = syn_ZD(Y, D, hp, flipped=True).sum(dim=1, keepdim=True)
    reconstruction_loss = nf * ((DZ + DY - S) ** 2).sum()
    FX = syn_FX(X, F, hp)
    syntax_reconstruction_loss = nf * ((FX - ZY) ** 2).sum()
    return reconstruction_loss + syntax_reconstruction_loss


def ADMM_XYZ(S, D, F, lsw, ls, lspw, lsp, osw, os_, pen, hp, ln, projs, trace=None):   # :330-357
    alpha = torch.zeros((hp.batch_size, hp.M, ln.c), dtype=S.dtype)            # :338 (@ignore)
    beta = torch.zeros_like(alpha)
    Z, Y, X, FX, left_FX, right_FX = warmup_XYZ(S, D, F, lsw, lspw, osw, hp, ln, projs)
    if trace is not None:
        trace.append(dict(Z=Z, Y=Y, X=X, FX=FX))
    for num_pass in range(hp.num_pass_xyz):
        Z, Y, X, FX, left_FX, right_FX, alpha, beta = one_forward_step_XYZ(
            S, Z, Y, D, X, F, left_FX, right_FX, FX, alpha, beta, lsp, ls, os_, pen, hp, ln, projs, num_pass)
        if trace is not None:
            trace.append(dict(Z=Z, Y=Y, X=X, FX=FX, alpha=alpha, beta=beta))
    return Z, Y, X


def ADMM_DF(S, Z, Y, X, D, F, mu, kappa_sparsity, kappa_stepsize, hp, ln, projs, trace=None):  # :362-373
    theta = torch.zeros((hp.batch_size, 1, hp.twoM, ln.c), dtype=S.dtype)      # :365 (@ignore)
    ZY = cat_ZY(Z, Y, hp, ln)
    for num_pass in range(hp.num_pass_df):
        D = update_D(S, Z, Y, D, mu, hp, ln, projs, num_pass)
        F = update_F(ZY, X, F, hp, ln, theta, kappa_stepsize, kappa_sparsity, num_pass)
        theta = theta + syn_FX(X, F, hp) - ZY
        if trace is not None:
            trace.append(dict(D=D, F=F, theta=theta))
    return ZY, D, F


def forward_pass_return_loss(S, cdl, hp, ln, projs, trace=None):               # :375-395
    (lspw, lsp, ksp, lsw, osw, ls, os_, ks, pen, mu, D, F_orig) = prep_params(cdl, hp, projs)
    Z, Y, X = ADMM_XYZ(S, D, F_orig, lsw, ls, lspw, lsp, osw, os_, pen, hp, ln, projs, trace)
    ZY, D2, F2 = ADMM_DF(S, Z, Y, X, D, F_orig, mu, ksp, ks, hp, ln, projs, trace)
    return loss(S, Z, Y, X, D2, ZY, F2, hp)


def retrieve_code(S, cdl, hp, ln, projs):                                      # :398-411
    (lspw, lsp, _, lsw, osw, ls, os_, _, pen, _, D, F_orig) = prep_params(cdl, hp, projs)
    Z, Y, X = ADMM_XYZ(S, D, F_orig, lsw, ls, lspw, lsp, osw, os_, pen, hp, ln, projs)
    return F_orig, Z, Y, X


def onehot_batch(codes, dtype=float_type):
    """(B, L) codes -> S with the bytes of Julia (4L, 1, B): torch (B, 1, 4L)  (loadfasta/helpers.jl:110-139)."""
    B, L = codes.shape
    S = torch.zeros((B, L, 4), dtype=dtype)
    idx = torch.as_tensor(np.asarray(codes), dtype=torch.long)
    S.scatter_(2, idx.unsqueeze(-1), 1.0)
    return S.reshape(B, 1, 4 * L)


CODE_REC = np.dtype([("position", "<u2"), ("fil", "<u2"), ("seq", "<u4"), ("mag", "<f2")])  # _0_const.jl:3-4


def code_retrieval(codes_all, cdl, hp, dtype=float_type):
    """_1_code_retrieval.jl:33-56: batches of `batch_size` in file order, remainder dropped (partial=false);
    `findall(X .> 0)` walks (l, 1, K, B) column-major: position fastest, then filter, then sequence."""
    N, Lbp = codes_all.shape
    ln = LengthInfo.make(hp, Lbp)
    projs = Projectors(hp, ln, dtype)
    out = []
    with torch.no_grad():
        for i0 in range(0, N - N % hp.batch_size, hp.batch_size):
            S = onehot_batch(codes_all[i0:i0 + hp.batch_size], dtype)
            _, _, _, X = retrieve_code(S, cdl, hp, ln, projs)
            Xn = X.numpy()                                                     # (B, K, 1, l)
            b, k, _, p = np.nonzero(Xn > 0)                                    # C-order nonzero == Julia column-major walk
            rec = np.zeros(len(b), dtype=CODE_REC)
            rec["position"], rec["fil"], rec["seq"] = p + 1, k + 1, b + i0 + 1
            rec["mag"] = Xn[b, k, 0, p].astype(np.float16)
            out.append(rec)
    return np.concatenate(out) if out else np.zeros(0, dtype=CODE_REC)


# ---- training step (train.jl:42-52) ---------------------------------------------------------------

def loss_and_grads(codes, cdl, hp, dtype=float_type):
    """One mini-batch: loss and d loss / d (the 9 trained arrays), in Flux.params order."""
    B, Lbp = codes.shape
    assert B == hp.batch_size
    ln = LengthInfo.make(hp, Lbp)
    projs = Projectors(hp, ln, dtype)
    cdl.to(dtype).requires_grad_(True)
    S = onehot_batch(codes, dtype)
    val = forward_pass_return_loss(S, cdl, hp, ln, projs)
    grads = torch.autograd.grad(val, cdl.params(), allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, cdl.params())]
    cdl.requires_grad_(False)
    return val.detach(), [g.detach() for g in grads]


class AdaBelief:
    """Flux 0.14.6 `Flux.AdaBelief()` (legacy Optimise API), defaults eta=1e-3, beta=(0.9, 0.999), eps=1e-8.
    The Flux source is not under the reference checkout; this is the published update rule
    (m, s zero-initialised; bias correction by running powers of beta), SURVEY.md §8 a15."""

    def __init__(self, eta=1e-3, beta=(0.9, 0.999), eps=1e-8):
        self.eta, self.beta, self.eps = eta, beta, eps
        self.state = {}

    def apply(self, idx, x, delta):
        b1, b2 = self.beta
        mt, st, bp = self.state.get(idx, (torch.zeros_like(x), torch.zeros_like(x), [b1, b2]))
        mt = b1 * mt + (1 - b1) * delta
        st = b2 * st + (1 - b2) * (delta - mt) ** 2 + self.eps
        step = self.eta * mt / (1 - bp[0]) / (torch.sqrt(st / (1 - bp[1])) + self.eps)
        self.state[idx] = (mt, st, [bp[0] * b1, bp[1] * b2])
        return x - step

    def update(self, cdl, grads):
        """Flux.Optimise.update!(opt, ps, gs)  (train.jl:46)."""
        names = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
                 "penalty_xyz", "mu"]
        for i, (name, g) in enumerate(zip(names, grads)):
            setattr(cdl, name, self.apply(i, getattr(cdl, name).detach(), g))


def l1_syntax(cdl):
    """train.jl:47  sum(abs.(prep_syntax_filters(cdl.F)))  (early stop when < 95)."""
    return prep_syntax_filters(cdl.F.detach()).abs().sum()
