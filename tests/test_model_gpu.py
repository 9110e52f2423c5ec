"""Parity of the HIP sparse-coding engine (motifs_model_* through the C ABI) against the CPU oracle
(oracle/model_oracle.py, a restatement of src/model.jl + train.jl + _1_code_retrieval.jl).
Tolerance: BASELINE.json north_star asks 1e-5 relative for float32 values.  The tolerances here are set from the ACHIEVED errors
(tools/parity_errors.py, three runs of every golden on one MI355X; profiles/r04_parity_errors.json): losses 2.5e-7,
gradients 5.8e-7 of the largest entry of their array, 1.5e-4 element-wise on the entries above 1e-3 of the largest (float32
cancellation in the D gradient), ZY / X intermediates 1.8e-7 - each bound below is 2-4x its achieved value."""
import os

import numpy as np
import pytest
import torch

from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-5            # north_star's bar (kept for the GPU-against-GPU comparisons further down)
LOSS_RTOL = 1e-6       # achieved 2.5e-7
GRAD_INF = 2e-6        # |got - want|_inf / |want|_inf per array; achieved 5.8e-7
GRAD_ELEM = 3e-4       # element-wise relative, entries above 1e-3 of the largest; achieved 1.5e-4
INTER_INF = 1e-6       # ZY / X intermediates; achieved 1.8e-7
NAMES = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
         "penalty_xyz", "mu"]


def to_model(pkg, ctx, hp_o, L, cdl_o, arena=1 << 30):
    md = pkg.model
    hp = md.Hyperparam(filter_len=hp_o.filter_len, M=hp_o.M, h=hp_o.h, K=hp_o.K, q=hp_o.q, batch_size=hp_o.batch_size,
                       num_pass_xyz=hp_o.num_pass_xyz, num_pass_df=hp_o.num_pass_df,
                       magnifying_factor=hp_o.magnifying_factor, gamma=hp_o.gamma)
    cdl = md.ucdl(hp, L, ctx=ctx, arena_bytes=arena)
    vecs = {n: getattr(cdl_o, n).detach().numpy() for n in mo.PARAM_VECS}
    cdl.set_fields(D=cdl_o.D.detach().numpy(), F=cdl_o.F.detach().numpy(),
                   lambda_sparsity_warmup=cdl_o.lambda_sparsity_warmup,
                   lambda_stepsize_warmup=cdl_o.lambda_stepsize_warmup,
                   omega_stepsize_warmup=cdl_o.omega_stepsize_warmup, **vecs)
    return cdl


def gpu_loss_grad(pkg, ctx, cdl, codes, n_groups, keep=False):
    lib = pkg._lib
    S, L = codes.shape
    raw = torch.from_numpy(np.ascontiguousarray(codes, dtype=np.uint8)).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
    loss = torch.zeros(n_groups, dtype=torch.float32, device="cuda")
    grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dcodes.data_ptr())
    cdl.model.loss_grad_dev(dcodes.data_ptr(), n_groups, loss.data_ptr(), grad.data_ptr(), keep)
    ctx.synchronize()
    return loss.cpu().numpy(), grad.cpu().numpy()


def split_grad(cdl, flat):
    m = cdl.model
    out = {"D": flat[: m.nD], "F": flat[m.nD: m.nD + m.nF]}
    o = m.nD + m.nF
    for name, n in zip(pkg_vec_fields(), pkg_vec_sizes(cdl.hp)):
        out[name] = flat[o:o + n]
        o += n
    return out


def pkg_vec_fields():
    return ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "penalty_xyz", "mu"]


def pkg_vec_sizes(hp):
    x, d = hp.num_pass_xyz, hp.num_pass_df
    return [x, d, x, x, d, x, d]


def rel_inf(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel_elem(a, b, floor=1e-3):
    """Largest element-wise relative error over the entries of b above `floor` of its largest."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    big = np.abs(b) > floor * max(np.abs(b).max(), 1e-300)
    return float((np.abs(a - b)[big] / np.abs(b)[big]).max()) if big.any() else 0.0


def assert_grad(got, want, name):
    assert rel_inf(got, want) <= GRAD_INF, (name, rel_inf(got, want))
    assert rel_elem(got, want) <= GRAD_ELEM, (name, rel_elem(got, want))


def tiny(seed, G=2, B=3, Lbp=30):
    hp = mo.Hyperparam(filter_len=4, M=5, h=3, K=4, q=6, batch_size=B, num_pass_xyz=2, num_pass_df=2)
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(G * B, Lbp)).astype(np.uint8)
    cdl = mo.UCDL(hp, rng).to(torch.float64)
    return hp, codes, cdl


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tiny_loss_grads_and_intermediates(ctx, pkg, seed):
    hp, codes, cdl_o = tiny(seed)
    G, B = 2, hp.batch_size
    cdl = to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, G, keep=True)
    got = split_grad(cdl, flat)
    ln = mo.LengthInfo.make(hp, codes.shape[1])
    projs = mo.Projectors(hp, ln, torch.float64)
    want = {n: 0.0 for n in NAMES}
    for g in range(G):
        val, grads = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl_o, hp, torch.float64)
        assert abs(loss[g] - val.item()) <= LOSS_RTOL * abs(val.item()), (g, loss[g], val.item())
        for n, gr in zip(NAMES, grads):
            want[n] = want[n] + gr.numpy()
        # intermediates of this mini-batch
        S = mo.onehot_batch(codes[g * B:(g + 1) * B], torch.float64)
        with torch.no_grad():
            _, Z, Y, X = mo.retrieve_code(S, cdl_o.to(torch.float64), hp, ln, projs)
        zy = torch.cat((Z[..., 0::4], Y[..., 0::4]), dim=1).permute(0, 2, 1).numpy()      # [B][c][2M]
        gzy = cdl.model.dump("ZY").reshape(G, B, ln.c, hp.twoM)[g]
        assert rel_inf(gzy, zy) <= INTER_INF
        gx = cdl.model.dump("X").reshape(G, B, ln.l, hp.K)[g]
        assert rel_inf(gx, X[:, :, 0, :].permute(0, 2, 1).numpy()) <= INTER_INF
    for n in NAMES:                      # the flat gradient is the SUM over the mini-batches
        assert_grad(got[n], want[n], n)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_mid_shapes_golden(ctx, pkg, i):
    """Three mid-size shapes against the float64 oracle (tests/golden/make_model_mid_golden.py): other template
    instances of the LDS-resident matrix-core kernels than configs[0]/[1] take (window heights 8 and 12, channel
    chunks of 32 / 80, odd K, 40- and 48-wide row GEMMs)."""
    g = np.load(os.path.join(HERE, "golden", "model_mid.npz"))
    fl, M, h, K, q, bp = [int(x) for x in g["shapes"][i]]
    G, B = 2, 3
    hp = mo.Hyperparam(filter_len=fl, M=M, h=h, K=K, q=q, batch_size=B, num_pass_xyz=2, num_pass_df=2)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g[f"s{i}_init_{n}"].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g[f"s{i}_warm"]]
    cdl = to_model(pkg, ctx, hp, bp, cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, g[f"s{i}_codes"], G)
    got = split_grad(cdl, flat)
    for k in range(G):
        assert abs(loss[k] - g[f"s{i}_loss{k}"]) <= LOSS_RTOL * g[f"s{i}_loss{k}"]
    for n in NAMES:
        want = g[f"s{i}_grad_{n}"].astype(np.float64)
        assert_grad(got[n], want, n)
    cdl.model.close()


def test_cfg1_golden(ctx, pkg):
    g = np.load(os.path.join(HERE, "golden", "model_cfg1.npz"))
    hp = mo.Hyperparam(filter_len=8, M=32)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g["init_" + n]))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
    cdl = to_model(pkg, ctx, hp, 100, cdl_o)
    codes = g["codes"]
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, 2)
    got = split_grad(cdl, flat)
    for k in range(2):
        assert abs(loss[k] - g[f"loss{k}"]) <= LOSS_RTOL * g[f"loss{k}"]
    for n in NAMES:
        want = g[f"grad0_{n}"] + g[f"grad1_{n}"]
        assert_grad(got[n], want, n)
    # code retrieval: positions / filters / sequence numbers exact, magnitudes to one fp16 ulp
    rec = pkg.model.code_retrieval(codes, cdl)
    assert np.array_equal(np.stack([rec["position"], rec["fil"], rec["seq"]], 1).astype(np.int64), g["codes_rec"])
    d = np.abs(rec["mag"].view(np.uint16).astype(np.int64) - g["codes_mag"].astype(np.int64))
    assert d.max() <= 1 and (d == 1).sum() <= max(1, len(d) // 100)      # achieved: 0 of 384 magnitudes differ (r04_parity_errors.json)


def test_cfg2_golden(ctx, pkg):
    """One mini-batch at BASELINE configs[1] shape (200 bp, 200 filters of length 12) against the float64 oracle
    (tests/golden/make_model_cfg2_golden.py): the matrix-core, tall and sparse paths at their real sizes."""
    g = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
    hp = mo.Hyperparam(filter_len=12, M=200)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g["init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
    cdl = to_model(pkg, ctx, hp, 200, cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, g["codes"], 1)
    got = split_grad(cdl, flat)
    assert abs(loss[0] - g["loss0"]) <= LOSS_RTOL * g["loss0"]
    for n in NAMES:
        assert_grad(got[n], g[f"grad0_{n}"].astype(np.float64), n)
    cdl.model.close()


def test_cfg3_golden(ctx, pkg):
    """One mini-batch at BASELINE configs[3] shape (500 bp, 512 filters of length 20) against the float64 oracle
    (tests/golden/make_model_cfg3_golden.py; the oracle's needed-lag / direct syntax forms, proven equal to the literal
    ones on small shapes): loss and every gradient.  The gradient of F (294 912 entries) is stored as every 7th entry +
    sums."""
    g = np.load(os.path.join(HERE, "golden", "model_cfg3.npz"))
    hp = mo.Hyperparam(filter_len=20, M=512)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g["init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
    cdl = to_model(pkg, ctx, hp, 500, cdl_o, arena=16 << 30)
    try:
        loss, flat = gpu_loss_grad(pkg, ctx, cdl, g["codes"], 1)
        got = split_grad(cdl, flat)
        assert abs(loss[0] - g["loss0"]) <= LOSS_RTOL * g["loss0"], (loss[0], g["loss0"])
        for n in NAMES:
            if n == "F":
                continue
            assert_grad(got[n], g[f"grad0_{n}"].astype(np.float64), n)
        gf = got["F"].astype(np.float64)
        stride = int(g["grad0_F_sample_stride"])
        assert np.abs(gf[::stride] - g["grad0_F_sample"].astype(np.float64)).max() <= GRAD_INF * g["grad0_F_absmax"]
        assert abs(np.abs(gf).max() - g["grad0_F_absmax"]) <= GRAD_INF * g["grad0_F_absmax"]
        assert abs((gf * gf).sum() - g["grad0_F_sumsq"]) <= 1e-3 * g["grad0_F_sumsq"]
    finally:
        cdl.model.close()


def test_cfg2_groups_are_independent(ctx, pkg):
    """Two mini-batches at configs[1] shape in one launch == one at a time: the per-group filter banks of the ADMM_DF
    passes and the group indexing of the LDS-resident GEMM kernels at their real sizes."""
    g = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
    hp = mo.Hyperparam(filter_len=12, M=200)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g["init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
    cdl = to_model(pkg, ctx, hp, 200, cdl_o)
    B = hp.batch_size
    codes = np.concatenate([g["codes"], np.random.default_rng(7).integers(0, 4, size=g["codes"].shape).astype(np.uint8)])
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, 2)
    acc = np.zeros_like(flat, dtype=np.float64)
    for k in range(2):
        l1, f1 = gpu_loss_grad(pkg, ctx, cdl, codes[k * B:(k + 1) * B], 1)
        assert abs(l1[0] - loss[k]) <= 1e-6 * abs(loss[k])
        acc += f1
    assert abs(loss[0] - g["loss0"]) <= LOSS_RTOL * g["loss0"]
    assert rel_inf(flat, acc) <= 1e-5
    cdl.model.close()


@pytest.mark.parametrize("passes", [(1, 1), (3, 4), (2, 5)])
def test_tiny_other_pass_counts(ctx, pkg, passes):
    """Pass counts other than the default: the dual-update folding of ADMM_XYZ and the telescoped residuals of ADMM_DF
    (R_t = -theta_{t-2}) for one, four and five DF passes, against the oracle's literal sequence."""
    px, pdf = passes
    G, B = 2, 3
    hp = mo.Hyperparam(filter_len=4, M=5, h=3, K=4, q=6, batch_size=B, num_pass_xyz=px, num_pass_df=pdf)
    rng = np.random.default_rng(100 + 10 * px + pdf)
    codes = rng.integers(0, 4, size=(G * B, 30)).astype(np.uint8)
    cdl_o = mo.UCDL(hp, rng).to(torch.float64)
    cdl = to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, G)
    got = split_grad(cdl, flat)
    want = {n: 0.0 for n in NAMES}
    for g in range(G):
        val, grads = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl_o, hp, torch.float64)
        assert abs(loss[g] - val.item()) <= LOSS_RTOL * abs(val.item()), (g, loss[g], val.item())
        for n, gr in zip(NAMES, grads):
            want[n] = want[n] + gr.numpy()
    for n in NAMES:
        assert rel_inf(got[n], want[n]) <= 5 * RTOL, (n, rel_inf(got[n], want[n]))
    cdl.model.close()


def test_df_telescoping_equals_literal_sequence(ctx, pkg, tmp_path):
    """ADMM_DF forms only the residuals / duals / syntheses that something consumes (R_1 = 0, R_t = -theta_{t-2}); the
    literal sequence of model.jl:362-373 (MOTIFS_DF_LITERAL=1, read once per process) gives the same loss and gradient."""
    import subprocess
    import sys
    outs = {}
    for tag, env in (("short", {}), ("literal", {"MOTIFS_DF_LITERAL": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        e = dict(os.environ)
        e.pop("MOTIFS_DF_LITERAL", None)
        e.update(env)
        subprocess.run([sys.executable, os.path.join(HERE, "_df_literal_helper.py"), path], check=True, env=e, timeout=300)
        outs[tag] = np.load(path)
    assert abs(outs["short"]["loss"][0] - outs["literal"]["loss"][0]) <= 1e-6 * abs(outs["literal"]["loss"][0])
    assert rel_inf(outs["short"]["flat"], outs["literal"]["flat"]) <= 1e-5


def test_fused_forms_equal_the_separate_launches(ctx, pkg, tmp_path):
    """Round 3 folded neighbouring kernels of the engine into one another (update_X's step + projection + entry lists, the +-S image
    into the synthesis gather, row GEMM + gather, the scattered rows / windows formed inside the row filter-gradient kernel,
    the a4 scan on base codes).  Every fold keeps its separate-launch form behind a switch that is read once per process: the
    configs[1]-shape mini-batch with all of them off gives the same loss and gradient (the fixture is also held against the
    float64 oracle in test_cfg2_golden)."""
    import subprocess
    import sys
    switches = {"MOTIFS_NO_F_STEP_NORM": "1", "MOTIFS_NO_BANK_FUSION": "1", "MOTIFS_NO_D_STEP": "1", "MOTIFS_NO_X_PROJECT": "1", "MOTIFS_NO_TOEP_PLUS": "1", "MOTIFS_NO_TALL_FUSED": "1", "MOTIFS_NO_ROW_SRC": "1", "MOTIFS_NO_ONEHOT_SCAN": "1"}
    outs = {}
    # third run: the arena starts as NaN bit patterns (MOTIFS_POISON_ARENA), so a kernel that reads what nothing wrote would show
    for tag, env in (("fused", {}), ("separate", switches), ("poisoned", {"MOTIFS_POISON_ARENA": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        e = dict(os.environ)
        for k in list(switches) + ["MOTIFS_POISON_ARENA"]:
            e.pop(k, None)
        e.update(env)
        subprocess.run([sys.executable, os.path.join(HERE, "_df_literal_helper.py"), path], check=True, env=e, timeout=300)
        outs[tag] = np.load(path)
    assert abs(outs["fused"]["loss"][0] - outs["separate"]["loss"][0]) <= 2e-6 * abs(outs["separate"]["loss"][0])
    assert rel_inf(outs["fused"]["flat"], outs["separate"]["flat"]) <= 1e-5
    assert np.isfinite(outs["poisoned"]["flat"]).all()
    assert abs(outs["fused"]["loss"][0] - outs["poisoned"]["loss"][0]) <= 2e-6 * abs(outs["fused"]["loss"][0])
    assert rel_inf(outs["fused"]["flat"], outs["poisoned"]["flat"]) <= 1e-5


def test_f16x3_syntax_gemm_meets_the_oracle(ctx, pkg, tmp_path):
    """Steps of many mini-batches run the syntax-layer analysis GEMM on the binary16 matrix instruction with three products per term
    (k_ana_f16x3: every float32 operand split into two binary16 numbers at a power-of-two scale).  MOTIFS_GEMM_F16_MIN=1 sends the
    one-mini-batch configs[1] fixture through it (a fresh process: the switch is read once): loss and every gradient against the float64
    oracle's at the tolerances of test_cfg2_golden, i.e. the split form is as close to the oracle as the float32 instruction."""
    import subprocess
    import sys
    path = str(tmp_path / "f16x3.npz")
    e = dict(os.environ)
    e.pop("MOTIFS_GEMM_F32", None)
    e["MOTIFS_GEMM_F16_MIN"] = "1"        # all four: the syntax GEMM, the D-layer GEMMs (k_rowgemm16, k_toep_wide16) and their filter gradients (k_rowwgrad16)
    subprocess.run([sys.executable, os.path.join(HERE, "_df_literal_helper.py"), path], check=True, env=e, timeout=300)
    out = np.load(path)
    g = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
    hp = mo.Hyperparam(filter_len=12, M=200)
    assert abs(out["loss"][0] - g["loss0"]) <= LOSS_RTOL * g["loss0"]
    nD, nF = 48 * 200, 12 * 400 * 24
    got = {"D": out["flat"][:nD], "F": out["flat"][nD:nD + nF]}
    o = nD + nF
    for name, n in zip(pkg_vec_fields(), pkg_vec_sizes(hp)):
        got[name] = out["flat"][o:o + n]
        o += n
    for n in NAMES:
        assert_grad(got[n], g[f"grad0_{n}"].astype(np.float64), n)


def test_binary16_gemms_on_other_shapes(ctx, pkg, tmp_path):
    """The same switches on the three mid-shape fixtures and the configs[0]-shape one (other template instances of the binary16 GEMMs: one and
    two column tiles, 40- / 36- / 48-wide filter gradients, banks of 64-128 channels, two mini-batches per launch), against the float64
    oracle's losses and gradients at the usual tolerances."""
    import subprocess
    import sys
    path = str(tmp_path / "f16forms.npz")
    e = dict(os.environ)
    e.pop("MOTIFS_GEMM_F32", None)
    e["MOTIFS_GEMM_F16_MIN"] = "1"
    subprocess.run([sys.executable, os.path.join(HERE, "_f16_forms_helper.py"), path], check=True, env=e, timeout=300)
    out = np.load(path)

    def split(flat, hp):
        nD, nF = hp.M * 4 * hp.filter_len, hp.K * 2 * hp.M * hp.h
        got = {"D": flat[:nD], "F": flat[nD:nD + nF]}
        o = nD + nF
        for name, n in zip(pkg_vec_fields(), pkg_vec_sizes(hp)):
            got[name] = flat[o:o + n]
            o += n
        return got

    gm = np.load(os.path.join(HERE, "golden", "model_mid.npz"))
    for i in range(3):
        fl, M, h, K, q, bp = [int(x) for x in gm["shapes"][i]]
        hp = mo.Hyperparam(filter_len=fl, M=M, h=h, K=K, q=q, batch_size=3, num_pass_xyz=2, num_pass_df=2)
        for k in range(2):
            assert abs(out[f"mid{i}_loss"][k] - gm[f"s{i}_loss{k}"]) <= LOSS_RTOL * gm[f"s{i}_loss{k}"]
        got = split(out[f"mid{i}_flat"], hp)
        for n in NAMES:
            assert_grad(got[n], gm[f"s{i}_grad_{n}"].astype(np.float64), n)
    g1 = np.load(os.path.join(HERE, "golden", "model_cfg1.npz"))
    hp = mo.Hyperparam(filter_len=8, M=32)
    for k in range(2):
        assert abs(out["cfg0_loss"][k] - g1[f"loss{k}"]) <= LOSS_RTOL * g1[f"loss{k}"]
    got = split(out["cfg0_flat"], hp)
    for n in NAMES:
        assert_grad(got[n], g1[f"grad0_{n}"] + g1[f"grad1_{n}"], n)


@pytest.mark.parametrize("G", [16, 32, 40, 64])
def test_step_sizes_take_different_kernels_and_agree(ctx, pkg, G):
    """The engine picks its kernel forms by step size (per-read sparse gradients to 24 mini-batches, 2-row synthesis blocks and
    split row walks below 192 reads, the fused tall form to ~100 reads, fused bank forms for small banks, ...).  The one-mini-batch
    forms are held against the float64 oracle (test_cfg2_golden); here a launch of G mini-batches at the configs[1] shape - 16: the
    middle forms, 40: the large-step forms, 64: the step bench.py times - against G launches of one: the same losses, and the sum of the gradients."""
    sy = pkg.synth
    hp = mo.Hyperparam(filter_len=12, M=200)
    L = 200
    # the state of the configs[1] fixture: its codes survive the shrinkage (loss ~150 of 200; at the library's own init every code dies in
    # the first pass and the code-image GEMMs would multiply zeros)
    gold = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(gold["init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in gold["warm"]]
    cdl = to_model(pkg, ctx, hp, L, cdl_o, arena=int((0.3 * G + 2) * (1 << 30)))
    try:
        codes = sy.gen_codes(G * hp.batch_size, L, 91, n_plant=5, k=12)
        l_all, g_all = gpu_loss_grad(pkg, ctx, cdl, codes, G)
        assert np.all(l_all < 190.0), "the codes died: this test would compare zeros"
        l_one, g_sum = [], np.zeros(cdl.model.nP, dtype=np.float64)
        for g in range(G):
            l, gr = gpu_loss_grad(pkg, ctx, cdl, codes[g * hp.batch_size:(g + 1) * hp.batch_size], 1)
            l_one.append(l[0])
            g_sum += gr
        # achieved (round 5, tools-free run of this comparison at 16 / 32 / 40 / 64 mini-batches): losses 2e-7; the flat gradient 6.5e-7 of its largest
        # entry (2e-5 was asked until round 4); per array: D 8.6e-7, the scalar vectors 6.5e-7, F 1e-6 at its 99.9 % quantile with 5-29 entries up to
        # 1.9e-4 of F's own largest entry - the launch sizes take different kernels, their float32 values differ in the last bits, and a median whose
        # middle values are an ulp apart then falls differently (the allowance of the multi-mini-batch golden test)
        assert np.all(np.isfinite(l_all)) and np.allclose(np.array(l_one), l_all, rtol=1e-6)
        assert rel_inf(g_all, g_sum.astype(np.float32)) <= 2e-6
        m = cdl.model
        assert rel_inf(g_all[: m.nD], g_sum[: m.nD]) <= GRAD_INF
        eF = np.abs(g_all[m.nD: m.nD + m.nF] - g_sum[m.nD: m.nD + m.nF]) / np.abs(g_sum[m.nD: m.nD + m.nF]).max()
        assert int((eF > GRAD_INF).sum()) <= F_TIE_ENTRIES and eF.max() <= F_TIE_INF and np.quantile(eF, 0.999) <= GRAD_INF, (int((eF > GRAD_INF).sum()), eF.max())
    finally:
        cdl.model.close()


def multi_golden_state():
    """The state and reads of tests/golden/model_cfg2_multi.npz (model_cfg2.npz's state: every code alive)."""
    gm = np.load(os.path.join(HERE, "golden", "model_cfg2_multi.npz"))
    gold = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
    hp = mo.Hyperparam(filter_len=12, M=200)
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(gold["init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in gold["warm"]]
    return gm, hp, cdl_o


VEC_INF = 4.5e-6       # the scalar-vector gradients of a 24- / 64-mini-batch launch (see check_multi_launch)
F_TIE_ENTRIES = 64     # entries of the F gradient (of 115 200) a launch may have outside GRAD_INF: medians within rounding noise of a one-ulp gap
F_TIE_INF = 5e-3       # ... and how far outside (|got - want|_inf / |want|_inf); one such median moved 1-9 entries by up to 1.7e-3


@pytest.mark.parametrize("G", [24, 64])
def test_multi_mini_batch_launch_meets_the_float64_oracle(ctx, pkg, G):
    """The launches bench.py's train leg times, against the float64 oracle DIRECTLY (tests/golden/make_model_cfg2_multi_golden.py:
    384 reads = 64 mini-batches at the configs[1] shape on the state of model_cfg2.npz).  Default switches: at 24 mini-batches the
    four-column sparse filter gradients (k_sp_wgrad_syn4 / _ana4 start at 21), at 64 the binary16 forms of the four GEMMs at their own
    thresholds, k_zy_step2_bwd<8> and the large-step k_lin3.  Every mini-batch's loss and the summed gradient (train.jl:42-44, one
    gradient per mini-batch; a launch returns their sum) at the tolerances of test_cfg2_golden - with ONE allowance, for the F
    gradient: the median of create_ZY_mask (model.jl:198-199) is a Float32 `a/2 + b/2` of two of 453 600 values that are less than
    one ulp apart in a sixth of the mini-batches, and whether a float32 run sees them as the same, adjacent or two floats apart is
    rounding noise (the golden takes the selections on float32-rounded float64 values; the generator's header has the mechanism).
    A median decided the other way moves a handful of entries of one filter's F gradient by up to 1.7e-3 of the largest entry and
    nothing else above 1e-6: at most F_TIE_ENTRIES entries of F may sit outside GRAD_INF, none further than F_TIE_INF."""
    gm, hp, cdl_o = multi_golden_state()
    assert G in (int(gm["g_mid"]), len(gm["losses"]))
    cdl = to_model(pkg, ctx, hp, 200, cdl_o, arena=int((0.3 * G + 2) * (1 << 30)))
    try:
        loss, flat = gpu_loss_grad(pkg, ctx, cdl, gm["codes"][: G * hp.batch_size], G)
        check_multi_launch(gm, cdl.hp, G, loss, flat)
    finally:
        cdl.model.close()


def check_multi_launch(gm, hp, G, loss, flat):
    want = gm["losses"][:G]
    assert np.all(want < 190.0), "the codes died: this test would compare zeros"
    assert np.abs(loss.astype(np.float64) - want).max() <= LOSS_RTOL * want.max(), np.abs(loss - want).max() / want.max()
    nD, nF = hp.M * 4 * hp.filter_len, hp.K * 2 * hp.M * hp.h
    got = {"D": flat[:nD], "F": flat[nD:nD + nF]}
    o = nD + nF
    for name, n in zip(pkg_vec_fields(), pkg_vec_sizes(hp)):
        got[name] = flat[o:o + n]
        o += n
    for n in NAMES:
        w = gm["grad%d_%s" % (G, n)].astype(np.float64)
        if n == "D":
            assert_grad(got[n], w, n)
            continue
        if n != "F":         # the seven 3- / 6-entry vectors: each entry is a sum over G x 453 600 products whose block partials meet in float
            # atomics - 1.1e-6 ... 2.2e-6 of the largest entry from run to run at 64 mini-batches (GRAD_INF there is marginal: VEC_INF = 2x achieved)
            assert rel_inf(got[n], w) <= VEC_INF and rel_elem(got[n], w) <= GRAD_ELEM, (n, rel_inf(got[n], w), rel_elem(got[n], w))
            continue
        e = np.abs(got[n].astype(np.float64) - w.ravel()) / np.abs(w).max()
        assert int((e > GRAD_INF).sum()) <= F_TIE_ENTRIES and e.max() <= F_TIE_INF, (int((e > GRAD_INF).sum()), e.max())
        assert np.quantile(e, 0.999) <= GRAD_INF / 2, np.quantile(e, 0.999)


def test_multi_mini_batch_launch_on_the_float32_matrix_instruction(pkg, tmp_path):
    """MOTIFS_GEMM_F32=1 (read once per process: a fresh one) keeps the four GEMMs of the 64-mini-batch launch on v_mfma_f32_32x32x2_f32 /
    16x16x4_f32 - the forms every launch took before round 4 and small launches still take: the same golden, the same bounds."""
    import subprocess
    import sys
    path = str(tmp_path / "f32.npz")
    e = dict(os.environ)
    e.pop("MOTIFS_GEMM_F16_MIN", None)
    e["MOTIFS_GEMM_F32"] = "1"
    subprocess.run([sys.executable, os.path.join(HERE, "_multi_helper.py"), "64", path], check=True, env=e, timeout=600)
    out = np.load(path)
    gm, hp, _ = multi_golden_state()
    check_multi_launch(gm, hp, 64, out["loss"], out["flat"])


def test_train_step_matches_adabelief_oracle(ctx, pkg):
    hp, codes, cdl_o = tiny(5, G=1)
    cdl = to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
    opt = mo.AdaBelief()
    cdl_ref = cdl_o
    for step in range(3):                 # three reference steps on the same mini-batch (train.jl:41-46)
        val, grads = mo.loss_and_grads(codes, cdl_ref, hp, torch.float64)
        opt.update(cdl_ref, grads)
        loss, l1 = cdl.model.train_step(codes, 1)
        assert abs(loss[0] - val.item()) <= 2 * RTOL * abs(val.item())
        assert abs(l1 - mo.l1_syntax(cdl_ref).item()) <= 1e-4 * l1
    f = cdl.fields()
    for n in mo.PARAM_VECS + ["D", "F"]:
        assert rel_inf(f[n], getattr(cdl_ref, n).detach().numpy()) <= 2e-5, n


def test_groups_are_independent(ctx, pkg):
    """G mini-batches in one launch == the same mini-batches one at a time."""
    hp, codes, cdl_o = tiny(9, G=3)
    B = hp.batch_size
    cdl = to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, 3)
    acc = np.zeros_like(flat, dtype=np.float64)
    for g in range(3):
        l1, f1 = gpu_loss_grad(pkg, ctx, cdl, codes[g * B:(g + 1) * B], 1)
        assert abs(l1[0] - loss[g]) <= 1e-6 * abs(loss[g])     # block partial sums meet in float atomics: last bit may differ
        acc += f1
    assert rel_inf(flat, acc) <= 1e-5


def test_filter_bank_scan_on_base_codes_equals_the_gemm_form(ctx, pkg):
    """a4 (warmup_ZY's conv pair) reads the base codes and gathers fl bank rows per output row once a launch has >= 96 reads
    (k_onehot_bank_scan); below that it is the Toeplitz GEMM on the one-hot image.  The same 32 mini-batches (96 reads, one with an
    all-zero column) in one launch and in four launches of 8 give the same losses and the same summed gradient."""
    G, B, Lbp = 32, 3, 40
    hp = mo.Hyperparam(filter_len=8, M=6, h=3, K=4, q=6, batch_size=B, num_pass_xyz=2, num_pass_df=2)     # 2M = 12 columns: four per lane
    rng = np.random.default_rng(21)
    codes = rng.integers(0, 4, size=(G * B, Lbp)).astype(np.uint8)
    cdl_o = mo.UCDL(hp, rng).to(torch.float64)
    codes[5, 7] = 4
    codes[95, Lbp - 1] = 4
    B = hp.batch_size
    cdl = to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, 32)
    acc = np.zeros_like(flat, dtype=np.float64)
    for g0 in range(0, 32, 8):
        l8, f8 = gpu_loss_grad(pkg, ctx, cdl, codes[g0 * B:(g0 + 8) * B], 8)
        assert np.allclose(l8, loss[g0:g0 + 8], rtol=2e-6, atol=0)
        acc += f8
    assert rel_inf(flat, acc) <= 1e-5


def test_train_ucdl_runs_and_code_retrieval_format(ctx, pkg):
    md = pkg.model
    hp = md.Hyperparam(filter_len=4, M=6, h=3, K=4, q=5, batch_size=3)
    codes = pkg.synth.gen_codes(31, 40, 3, n_plant=2, k=4)
    cdl, _, losses = md.train_ucdl(codes, hp, num_epochs=2, groups_per_step=2, ctx=ctx, arena_bytes=1 << 30,
                                     l1_loss_thresh=0.0)
    assert len(losses) == 2 * 10 and np.all(np.isfinite(losses))
    # the reference stops as soon as sum|F| < 95 (train.jl:47-52); this small bank is below it from the start
    _, _, short = md.train_ucdl(codes, hp, num_epochs=2, groups_per_step=2, ctx=ctx, arena_bytes=1 << 30)
    assert len(short) == 2
    rec = md.code_retrieval(codes, cdl)
    assert rec["seq"].max() <= 30 and rec["seq"].min() >= 1          # remainder (31st read) dropped: partial=false
    key = list(zip(rec["seq"].tolist(), rec["fil"].tolist(), rec["position"].tolist()))
    assert key == sorted(key) and len(key) >= 30 * hp.q // 2         # findall order: position fastest, then fil, then seq
    assert (rec["mag"] > 0).all()


def test_cfg4_shape_runs(ctx, pkg):
    """BASELINE configs[3] shape (500 bp, 512 filters of length 20) with the reference's own init scale (test_cfg3_golden
    holds one mini-batch to the oracle): finite losses, gradient of two mini-batches == sum of the two single ones."""
    md = pkg.model
    hp = md.Hyperparam(filter_len=20, M=512)
    L = 500
    cdl = md.ucdl(hp, L, ctx=ctx, seed=4, arena_bytes=24 << 30)
    codes = pkg.synth.gen_codes(12, L, 4, n_plant=3, k=20)
    loss, flat = gpu_loss_grad(pkg, ctx, cdl, codes, 2)
    assert np.all(np.isfinite(loss)) and np.all(np.isfinite(flat)) and np.abs(flat).max() > 0
    acc = np.zeros_like(flat, dtype=np.float64)
    for g in range(2):
        l1, f1 = gpu_loss_grad(pkg, ctx, cdl, codes[6 * g:6 * g + 6], 1)
        assert abs(l1[0] - loss[g]) <= 1e-5 * abs(loss[g])
        acc += f1
    assert rel_inf(flat, acc) <= 1e-4
    cdl.model.close()


@pytest.mark.parametrize("null_stream", [False, True])
def test_replayed_steps_follow_the_reads(pkg, null_stream):
    """The reference's schedule (train.jl:40-46): one mini-batch per step, every step other reads behind the same pointers.
    From the third step on the launches are replayed from a hipGraph; each replay must give the loss and the gradient of ITS
    reads.  With hipMemsetAsync / hipMemcpyAsync nodes in the captured step (round 2) later replays did not (group 3 of 32
    came back as 35.8199 for 35.7228): twelve one-mini-batch steps against one eager step of twelve (> 8: never captured)."""
    md, sy, lib = pkg.model, pkg.synth, pkg._lib
    c = lib.Context(0)
    if null_stream:
        c.set_stream(0)
    try:
        hp = md.Hyperparam(filter_len=8, M=6, K=4, q=6, h=3, batch_size=3, num_pass_xyz=2, num_pass_df=2)
        L, G = 40, 12
        cdl = md.ucdl(hp, L, ctx=c, seed=5, arena_bytes=1 << 30)
        codes = np.random.default_rng(21).integers(0, 4, size=(G * hp.batch_size, L)).astype(np.uint8)
        nP = cdl.model.nP

        def run(rows, n, dcodes, loss, grad):
            raw = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
            torch.cuda.synchronize()
            c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, rows.shape[0], L, dcodes.data_ptr())
            cdl.model.loss_grad_dev(dcodes.data_ptr(), n, loss.data_ptr(), grad.data_ptr())
            c.synchronize()
            return loss.cpu().numpy().copy(), grad.cpu().numpy().copy()

        big = [torch.zeros(lib.Context.codes_bytes(G * hp.batch_size, L), dtype=torch.uint8, device="cuda"),
               torch.zeros(G, dtype=torch.float32, device="cuda"), torch.zeros(nP, dtype=torch.float32, device="cuda")]
        l_all, g_all = run(codes, G, *big)
        one = [torch.zeros(lib.Context.codes_bytes(hp.batch_size, L), dtype=torch.uint8, device="cuda"),
               torch.zeros(1, dtype=torch.float32, device="cuda"), torch.zeros(nP, dtype=torch.float32, device="cuda")]
        l_one, g_sum = [], np.zeros(nP, dtype=np.float64)
        for g in range(G):
            l, gr = run(codes[g * hp.batch_size:(g + 1) * hp.batch_size], 1, *one)
            l_one.append(l[0])
            g_sum += gr
        assert np.allclose(np.array(l_one), l_all, rtol=2e-6), (l_one, l_all)
        assert np.allclose(g_sum, g_all, rtol=0, atol=1e-5 * np.abs(g_all).max())
        cdl.model.close()
    finally:
        c.close()


def test_step_graph_replay_equals_eager(ctx, pkg):
    """Small steps (<= 8 mini-batches) are captured into a hipGraph on their second call with the same buffers and replayed
    from the third: same losses and gradient as the eager first call (float atomics leave ulp-level noise), and the
    replay follows new parameters and new reads, which it reads from memory."""
    md, sy, lib = pkg.model, pkg.synth, pkg._lib
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    L, G = 60, 3
    cdl = md.ucdl(hp, L, ctx=ctx, seed=77, arena_bytes=1 << 30)
    try:
        S = G * hp.batch_size
        dcodes = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
        loss = torch.zeros(G, dtype=torch.float32, device="cuda")
        grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")

        def run(codes):
            raw = torch.from_numpy(codes).cuda()
            torch.cuda.synchronize()
            ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dcodes.data_ptr())
            cdl.model.loss_grad_dev(dcodes.data_ptr(), G, loss.data_ptr(), grad.data_ptr())
            ctx.synchronize()
            return loss.cpu().numpy().copy(), grad.cpu().numpy().copy()

        c1, c2 = sy.gen_codes(S, L, 1, n_plant=2, k=8), sy.gen_codes(S, L, 2, n_plant=2, k=8)
        l_eager, g_eager = run(c1)          # 1: eager
        l_cap, g_cap = run(c1)              # 2: captured, then launched
        l_rep, g_rep = run(c1)              # 3: replayed
        scale = np.abs(g_eager).max()
        for l, g in ((l_cap, g_cap), (l_rep, g_rep)):
            assert np.allclose(l, l_eager, rtol=2e-6)
            assert np.allclose(g, g_eager, rtol=0, atol=2e-6 * scale)
        l2_rep, _ = run(c2)                 # other reads through the same buffers
        assert not np.allclose(l2_rep, l_eager, rtol=1e-4)
        D, F, w, v = cdl.model.get_params()
        cdl.model.set_params(D * 1.01, F, None, None)
        l3_rep, _ = run(c2)                 # other parameters
        assert not np.allclose(l3_rep, l2_rep, rtol=1e-6)
        fresh = md.ucdl(hp, L, ctx=ctx, seed=77, arena_bytes=1 << 30)
        try:
            fresh.model.set_params(D * 1.01, F, w, v)
            lo2, _ = gpu_loss_grad(pkg, ctx, fresh, c2, G)      # eager (first sight of its buffers)
            assert np.allclose(lo2, l3_rep, rtol=2e-6)
        finally:
            fresh.model.close()
    finally:
        cdl.model.close()
