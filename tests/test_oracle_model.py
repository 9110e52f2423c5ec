"""Pins oracle/model_oracle.py (the restatement of src/model.jl).  The reference has no tests
(test/runtests.jl:4-6), so the pins are: NNlib's conv definition written as scalar loops, the algebraic
identities of SURVEY.md §8(c), invariants, finite differences, and a hand-rolled AdaBelief trace."""
import numpy as np
import pytest
import torch

from oracle import model_oracle as mo

DT = torch.float64


def tiny(seed=0, B=3, Lbp=30):
    hp = mo.Hyperparam(filter_len=4, M=5, h=3, K=4, q=6, batch_size=B, num_pass_xyz=2, num_pass_df=2)
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(B, Lbp))
    cdl = mo.UCDL(hp, rng).to(DT)
    ln = mo.LengthInfo.make(hp, Lbp)
    return hp, codes, cdl, ln, mo.Projectors(hp, ln, DT)


def nnlib_conv1_loops(x, w, pad, flipped, groups):
    """NNlib.conv written out: x (W, Cin, B), w (k, Cin/g, Cout), Julia index order, 0-based loops."""
    W, Cin, B = x.shape
    k, cg, Cout = w.shape
    xp = np.zeros((W + 2 * pad, Cin, B))
    xp[pad:pad + W] = x
    Wo = W + 2 * pad - k + 1
    y = np.zeros((Wo, Cout, B))
    opg = Cout // groups
    for b in range(B):
        for co in range(Cout):
            g = co // opg
            for t in range(Wo):
                s = 0.0
                for ci in range(cg):
                    for j in range(k):
                        wj = w[j, ci, co] if flipped else w[k - 1 - j, ci, co]
                        s += xp[t + j, g * cg + ci, b] * wj
                y[t, co, b] = s
    return y


def rev(a):
    return np.ascontiguousarray(np.transpose(a, tuple(reversed(range(a.ndim)))))


@pytest.mark.parametrize("flipped", [False, True])
@pytest.mark.parametrize("groups,pad", [(1, 0), (1, 2), (3, 4)])
def test_conv1_matches_nnlib_definition(flipped, groups, pad):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((11, 3, 2))
    w = rng.standard_normal((5, 3 // groups, 6))
    want = nnlib_conv1_loops(x, w, pad, flipped, groups)
    got = mo.conv1(torch.tensor(rev(x)), torch.tensor(rev(w)), pad=pad, flipped=flipped, groups=groups).numpy()
    assert np.allclose(rev(got), want, atol=1e-12)


def test_conv2_matches_loops():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((7, 5, 2, 2))          # Julia (W, H, Cin, B)
    w = rng.standard_normal((3, 5, 2, 3))          # (k1, k2, Cin, Cout): kernel spans the full height
    for flipped in (False, True):
        want = np.zeros((5, 1, 3, 2))
        for b in range(2):
            for co in range(3):
                for t in range(5):
                    s = 0.0
                    for ci in range(2):
                        for i in range(3):
                            for j in range(5):
                                wij = w[i, j, ci, co] if flipped else w[2 - i, 4 - j, ci, co]
                                s += x[t + i, j, ci, b] * wij
                    want[t, 0, co, b] = s
        got = mo.conv2(torch.tensor(rev(x)), torch.tensor(rev(w)), flipped=flipped).numpy()
        assert np.allclose(rev(got), want, atol=1e-12)


def test_gather_form_of_warmup_scan():
    """stride-1 conv + z_mask == per-position gather of fl weights, both strands."""
    hp, codes, cdl, ln, projs = tiny()
    D = mo.prep_filters(cdl.D, hp, projs)
    S = mo.onehot_batch(codes, DT)
    DtS = mo.conv1(S, D, flipped=True)[..., 0::4].numpy()      # (B, M, c)
    DS = mo.conv1(S, D)[..., 0::4].numpy()
    Dn = D.numpy().reshape(hp.M, hp.filter_len, 4)
    for b in range(codes.shape[0]):
        for p in range(ln.c):
            fwd = sum(Dn[:, k, codes[b, p + k]] for k in range(hp.filter_len))
            rc = sum(Dn[:, hp.filter_len - 1 - k, 3 - codes[b, p + k]] for k in range(hp.filter_len))
            assert np.allclose(DtS[b, :, p], fwd) and np.allclose(DS[b, :, p], rc)
    # prep_filters output is a PFM: sums to one over the four bases at every position
    assert np.allclose(Dn.sum(-1), 1)


def test_delta_filter_known_answer():
    """A one-hot filter planted in a sequence scores filter_len at the planted site and nowhere more."""
    hp = mo.Hyperparam(filter_len=4, M=2, h=3, K=4, q=6, batch_size=1)
    motif = [2, 0, 3, 1]
    D = torch.zeros((hp.M, 1, hp.f_len), dtype=DT)
    for k, a in enumerate(motif):
        D[0, 0, 4 * k + a] = 1
    codes = np.array([[0, 1, 1, 2, 0, 3, 1, 0, 0, 2]])
    DtS = mo.conv1(mo.onehot_batch(codes, DT), D, flipped=True)[0, 0, 0::4]
    assert DtS.argmax().item() == 3 and DtS.max().item() == 4.0


def test_adjointness_and_needed_lags():
    hp, codes, cdl, ln, projs = tiny(1)
    rng = np.random.default_rng(2)
    D = mo.prep_filters(cdl.D, hp, projs)
    B = hp.batch_size
    zm = projs.z_mask_n
    Z = torch.tensor(rng.random((B, hp.M, ln.C))) * zm
    Y = torch.tensor(rng.random((B, hp.M, ln.C))) * zm
    r = torch.tensor(rng.standard_normal((B, 1, ln.L)))
    # <analysis(r), Z> == <r, synthesis(Z)> for both strands (a4 <-> a9)
    assert torch.allclose((mo.conv1(r, D, flipped=True) * Z).sum(), (r * mo.syn_ZD(Z, D, hp).sum(1, keepdim=True)).sum())
    assert torch.allclose((mo.conv1(r, D) * Y).sum(), (r * mo.syn_ZD(Y, D, hp, flipped=True).sum(1, keepdim=True)).sum())
    # syntax layer (a7 <-> a8)
    F = mo.prep_syntax_filters(cdl.F)
    img = torch.tensor(rng.standard_normal((B, 1, hp.twoM, ln.c)))
    X = torch.tensor(rng.standard_normal((B, hp.K, 1, ln.l)))
    assert torch.allclose((mo.conv2(img, F, flipped=True) * X).sum(), (img * mo.syn_FX(X, F, hp)).sum())
    # full-lag conv_code_diff + mapdrange == needed-lag correlation (a11)
    full = (mo.conv_code_diff(Z, r, hp, ln).sum(0).reshape(hp.M, ln.CS_vlen) @ projs.mapdrange).numpy()
    Zc, rn = Z.numpy()[..., 0::4], r.numpy()[:, 0]
    want = np.zeros((hp.M, hp.f_len))
    for b in range(B):
        for p in range(ln.c):
            want += np.outer(Zc[b, :, p], rn[b, 4 * p:4 * p + hp.f_len])
    assert np.allclose(full, want)
    fullY = (mo.conv_code_diff(Y, r, hp, ln).flip(-1).sum(0).reshape(hp.M, ln.CS_vlen) @ projs.mapdrange).numpy()
    Yc = Y.numpy()[..., 0::4]
    wantY = np.zeros((hp.M, hp.f_len))
    for b in range(B):
        for p in range(ln.c):
            wantY += np.outer(Yc[b, :, p], rn[b, 4 * p:4 * p + hp.f_len][::-1])
    assert np.allclose(fullY, wantY)


def test_invariants_along_the_forward_pass():
    hp, codes, cdl, ln, projs = tiny(5)
    S = mo.onehot_batch(codes, DT)
    trace = []
    with torch.no_grad():
        val = mo.forward_pass_return_loss(S, cdl, hp, ln, projs, trace)
    assert torch.isfinite(val)
    for t in trace[: hp.num_pass_xyz + 1]:
        for code in (t["Z"], t["Y"]):
            assert (code >= 0).all()
            off = code.clone()
            off[..., 0::4] = 0
            assert not off.any()                       # only rows 1:4:end are ever non-zero
        nz = (t["X"] != 0).reshape(hp.batch_size, -1).sum(1)
        assert (nz <= max(hp.q, 1) * 4).all()          # top-q projection (ties may keep a few more)
    for t in trace[hp.num_pass_xyz + 1:]:
        assert torch.allclose(t["D"].reshape(hp.M, hp.filter_len, 4).sum(-1), torch.ones(hp.M, hp.filter_len, dtype=DT))
        assert torch.allclose((t["F"] ** 2).sum((-1, -2)), torch.ones(hp.K, 1, dtype=DT))


def test_median_and_topq_semantics():
    v = torch.tensor([3.0, 1.0, 4.0, 1.5])
    assert mo.julia_median(v).item() == 2.25         # even count: mean of the two middle values
    assert mo.julia_median(torch.tensor([3.0, 1.0, 4.0])).item() == 3.0
    hp = mo.Hyperparam(q=2, batch_size=1, K=1)
    X = torch.tensor([[[[5.0, 1.0, 5.0, 7.0, 0.0]]]])
    assert mo.generate_bitmat(X, hp).flatten().tolist() == [1, 0, 1, 1, 0]   # ties with the q-th value are kept (>=)


def test_float32_decisions_mode():
    """DECISIONS_F32: the median mask decided on float32-rounded values, as the Float32 reference decides it (model.jl:198-199,
    Statistics.middle(a, b) = a/2 + b/2).  Two middle values one float32 ulp apart: in float32 their mean rounds to the lower one
    (ties to even), which the mask then keeps; in float64 the mean lies strictly between.  Values well apart: both modes agree."""
    a = np.float32(1.6663759e-4)
    a = a if (a.view(np.uint32) & 1) == 0 else np.nextafter(a, np.float32(1))     # even mantissa: the tie rounds to a
    b = np.nextafter(a, np.float32(1))
    ZY = torch.tensor([[0.0, float(a) / 4, float(a), float(b), float(b) * 3]], dtype=torch.float64)
    try:
        mo.DECISIONS_F32 = False
        assert mo.create_ZY_mask(ZY).tolist() == [[0.0, 0.0, 0.0, 1.0, 1.0]]
        mo.DECISIONS_F32 = True
        assert mo.create_ZY_mask(ZY).tolist() == [[0.0, 0.0, 1.0, 1.0, 1.0]]
        far = torch.tensor([[0.0, 0.1, 0.2, 0.4, 0.8]], dtype=torch.float64)
        m32 = mo.create_ZY_mask(far)
        X = torch.rand((2, 4, 1, 9), dtype=torch.float64)
        hp = mo.Hyperparam(filter_len=4, M=5, h=3, K=4, q=6, batch_size=2)
        b32 = mo.generate_bitmat(X, hp)
        mo.DECISIONS_F32 = False
        assert torch.equal(m32, mo.create_ZY_mask(far)) and torch.equal(b32, mo.generate_bitmat(X, hp))
        assert m32.dtype == torch.float64 and b32.dtype == torch.float64
    finally:
        mo.DECISIONS_F32 = False


def test_gradients_against_finite_differences():
    hp, codes, cdl, ln, projs = tiny(7)
    val, grads = mo.loss_and_grads(codes, cdl, hp, DT)
    S = mo.onehot_batch(codes, DT)

    def f():
        with torch.no_grad():
            return mo.forward_pass_return_loss(S, cdl, hp, ln, projs).item()

    rng = np.random.default_rng(0)
    names = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
             "penalty_xyz", "mu"]
    checked = 0
    for name, g in zip(names, grads):
        p = getattr(cdl, name)
        flat = p.view(-1)
        for idx in rng.choice(flat.numel(), size=min(3, flat.numel()), replace=False):
            old = flat[idx].item()
            eps = 1e-6 * max(1.0, abs(old))
            flat[idx] = old + eps
            up = f()
            flat[idx] = old - eps
            dn = f()
            flat[idx] = old
            fd = (up - dn) / (2 * eps)
            an = g.view(-1)[idx].item()
            # the discrete selections (top-q, median) are constants in the backward (@ignore, :190, :208);
            # a finite difference that flips one is discontinuous, so only require agreement when smooth
            if abs(fd - an) <= 1e-4 * max(1.0, abs(fd), abs(an)):
                checked += 1
    assert checked >= 20


def test_adabelief_three_steps_by_hand():
    x = torch.tensor([1.0, -2.0, 0.5, 3.0], dtype=DT)
    grads = [torch.tensor(g, dtype=DT) for g in ([0.1, -0.2, 0.3, 0.0], [0.2, 0.1, -0.3, 0.5], [-0.1, 0.0, 0.2, 0.4])]
    opt = mo.AdaBelief()
    xs = x.clone()
    for g in grads:
        xs = opt.apply(0, xs, g)
    m = np.zeros(4)
    s = np.zeros(4)
    xh = x.numpy().copy()
    b1p, b2p = 0.9, 0.999
    for g in grads:
        g = g.numpy()
        m = 0.9 * m + 0.1 * g
        s = 0.999 * s + 0.001 * (g - m) ** 2 + 1e-8
        xh = xh - 1e-3 * m / (1 - b1p) / (np.sqrt(s / (1 - b2p)) + 1e-8)
        b1p *= 0.9
        b2p *= 0.999
    assert np.allclose(xs.numpy(), xh, rtol=1e-12)


def test_needed_lag_update_D_equals_the_literal_one():
    """mo.NEEDED_LAGS and mo.FAST_SYNTAX (used only to generate the configs[3]-shape golden mini-batch, which the literal
    torch forms cannot do) change nothing: the loss and every gradient agree with the literal forms."""
    hp, codes, cdl, ln, projs = tiny(3)
    try:
        mo.NEEDED_LAGS = mo.FAST_SYNTAX = False
        v0, g0 = mo.loss_and_grads(codes, cdl, hp, DT)
        mo.NEEDED_LAGS = mo.FAST_SYNTAX = True
        v1, g1 = mo.loss_and_grads(codes, cdl, hp, DT)
    finally:
        mo.NEEDED_LAGS = mo.FAST_SYNTAX = False
    assert abs(v0.item() - v1.item()) <= 1e-12 * abs(v0.item())
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-12 * float(a.abs().max() + 1))
