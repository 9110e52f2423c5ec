"""Host mirror of src/model.jl's public pieces and src/train.jl / _1_code_retrieval.jl drivers.

Same names and argument meaning as the Julia functions; all numeric work happens in
libmotifs_hip (`motifs_model_*`).  Array layouts are the reference's: `D` has the bytes of
Julia's (f_len, 1, M) array, i.e. numpy shape (M, 1, f_len); `F` those of (h, twoM, 1, K),
numpy (K, 1, twoM, h)."""
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class Hyperparam:                       # model.jl:1-14
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
        return 4 * self.filter_len

    @property
    def twoM(self):
        return 2 * self.M

    def to_c(self):
        return _lib.HParams(self.filter_len, self.M, self.h, self.K, self.q, self.batch_size, self.num_pass_xyz,
                            self.num_pass_df, self.magnifying_factor, self.gamma)


VEC_FIELDS = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize",
              "penalty_xyz", "mu"]      # array fields of `ucdl` in Flux.params order, D and F aside (model.jl:68-82)


def vec_sizes(hp):
    x, d = hp.num_pass_xyz, hp.num_pass_df
    return [x, d, x, x, d, x, d]


class ucdl:
    """The learnable state (model.jl:67-137) living on the device."""

    def __init__(self, hp, L, ctx=None, seed=None, arena_bytes=0):
        from .scan import default_context

        self.hp, self.L = hp, int(L)
        self.ctx = ctx or default_context()
        self.model = _lib.Model(self.ctx, hp.to_c(), L, arena_bytes)
        if seed is not None:
            self.model.init_random(seed)

    # -- field access in the reference's layouts --
    def fields(self):
        D, F, w, v = self.model.get_params()
        out = {"D": D.reshape(self.hp.M, 1, self.hp.f_len), "F": F.reshape(self.hp.K, 1, self.hp.twoM, self.hp.h),
               "lambda_sparsity_warmup": float(w[0]), "lambda_stepsize_warmup": float(w[1]),
               "omega_stepsize_warmup": float(w[2])}
        o = 0
        for name, n in zip(VEC_FIELDS, vec_sizes(self.hp)):
            out[name] = v[o:o + n].copy()
            o += n
        return out

    def set_fields(self, **kw):
        D = kw.get("D")
        F = kw.get("F")
        w = None
        if any(k.endswith("_warmup") for k in kw):
            cur = self.fields()
            w = np.array([kw.get(k, cur[k]) for k in ("lambda_sparsity_warmup", "lambda_stepsize_warmup",
                                                      "omega_stepsize_warmup")], dtype=np.float32)
        v = None
        if any(k in kw for k in VEC_FIELDS):
            cur = self.fields()
            v = np.concatenate([np.asarray(kw.get(k, cur[k]), dtype=np.float32) for k in VEC_FIELDS])
        self.model.set_params(D, F, w, v)


def setup_num_epochs(number_training_samples):     # train.jl:1-11
    if number_training_samples < 1000:
        return 25
    if number_training_samples < 10000:
        return 10
    if number_training_samples < 100000:
        return 5
    return 3


def train_ucdl(codes, hp=None, num_epochs=None, l1_loss_thresh=95.0, groups_per_step=1, seed=0, ctx=None,
               shuffle_seed=0, verbose=False, arena_bytes=0):
    """train.jl:13-58.  `codes`: (N, L) uint8 base codes of the training reads.
    groups_per_step = 1 reproduces the reference schedule (one AdaBelief step per mini-batch of
    hp.batch_size reads, shuffled, partial=false); larger values average the gradient of that many
    mini-batches per step (the data-parallel form)."""
    hp = hp or Hyperparam()
    N, L = codes.shape
    cdl = ucdl(hp, L, ctx=ctx, seed=seed, arena_bytes=arena_bytes)
    num_epochs = setup_num_epochs(N) if num_epochs is None else num_epochs
    rng = np.random.default_rng(shuffle_seed)
    per_step = hp.batch_size * groups_per_step
    losses = []
    stop = False
    for epoch in range(num_epochs):
        order = rng.permutation(N)                                    # DataLoader(shuffle=true)
        nfull = (N // hp.batch_size) * hp.batch_size                  # partial=false
        for i0 in range(0, nfull, per_step):
            idx = order[i0:min(i0 + per_step, nfull)]
            g = len(idx) // hp.batch_size
            loss, l1 = cdl.model.train_step(codes[idx[: g * hp.batch_size]], g)
            losses.extend(loss.tolist())
            if verbose:
                print("loss", loss.mean())                            # model.jl:392
            if l1 < l1_loss_thresh:                                   # train.jl:47-52
                stop = True
                break
        if stop:
            break
        if verbose:
            print(f"Epoch: {epoch + 1} completed")                    # train.jl:55
    return cdl, hp, losses


def code_retrieval(codes, cdl):
    """_1_code_retrieval.jl:33-56: stored_code_component_t records (position, fil, seq, mag) for every
    positive entry of X, mini-batches of hp.batch_size in file order, remainder dropped."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return cdl.model.retrieve_codes(codes, _lib.DATA_CODES_U8, codes.shape[0])
