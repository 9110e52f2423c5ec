"""Pins the scan oracle (oracle/scan_oracle.c).  The reference has no tests or
golden vectors (test/runtests.jl:4-6), so the pins are: numpy's correctly rounded
Float16 arithmetic, a hand-computed known-answer case, the reverse-complement
identity and the committed golden fixture."""
import os

import numpy as np
import pytest

from oracle import scan_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))


def test_soft_fp16_add_matches_numpy():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 1 << 16, size=200000, dtype=np.uint16)
    b = rng.integers(0, 1 << 16, size=200000, dtype=np.uint16)
    fa, fb = a.view(np.float16), b.view(np.float16)
    ok = np.isfinite(fa) & np.isfinite(fb)
    with np.errstate(over="ignore", invalid="ignore"):
        want = (fa + fb).view(np.uint16)
    lib = so.lib()
    got = np.array([lib.oracle_h_add(int(x), int(y)) for x, y in zip(a[ok][:50000], b[ok][:50000])], dtype=np.uint16)
    w = want[ok][:50000]
    nan = np.isnan(w.view(np.float16))
    assert np.array_equal(got[~nan], w[~nan])


def test_f32_to_f16_matches_numpy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)])
    x = np.concatenate([x, np.float32([0, -0.0, 65504, 65519.99, 65520, 2.0**-25, 2.0**-24, 3 * 2.0**-25, 6.1e-5])])
    lib = so.lib()
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([lib.oracle_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_known_answer_by_hand():
    """4x3 PWM on ACGTACGT, both strands, worked by hand (Float16-exact values)."""
    # pwm[a, ind]: rows A,C,G,T
    pwm = np.array([[1.0, -2.0, 0.5], [-1.0, 3.0, -0.5], [0.25, -4.0, 2.0], [-8.0, 0.125, -0.25]], dtype=np.float16)
    bank = np.zeros((3, 4, 1), dtype=np.float16)
    bank[:, :, 0] = pwm.T
    lens = np.array([3])
    codes = np.array([[0, 1, 2, 3, 0, 1, 2, 3]], dtype=np.uint8)  # ACGTACGT
    onehot = np.zeros((1, 32), dtype=np.float32)
    for p, c in enumerate(codes[0]):
        onehot[0, 4 * p + c] = 1
    # forward windows: ACG, CGT, GTA, TAC, ACG, CGT
    # ACG: 1 + 3 + 2 = 6 ; CGT: -1 - 4 - 0.25 < 0 ; GTA: 0.25 + 0.125 + 0.5 = 0.875 ; TAC: -8 ... < 0
    found, score = so.get_pos_scores_arr(bank, lens, onehot, rc=False)
    assert [(int(f["m"]), int(f["n"]), int(f["l"])) for f in found] == [(1, 1, 1), (1, 1, 3), (1, 1, 5)]
    assert score.tolist() == [6.0, 0.875, 6.0]
    # reverse strand uses reverse(pwm) in both dims (:68): rc[a, ind] = pwm[3-a, 2-ind]
    # ACG: rc[A,0]+rc[C,1]+rc[G,2] = pwm[T,2]+pwm[G,1]+pwm[C,0] = -0.25-4-1 < 0
    # CGT: pwm[G,2]+pwm[C,1]+pwm[A,0] = 2+3+1 = 6 ; GTA: pwm[C,2]+pwm[A,1]+pwm[T,0] < 0 ; TAC: pwm[A,2]+pwm[T,1]+pwm[G,0] = 0.5+0.125+0.25
    found, score = so.get_pos_scores_arr(bank, lens, onehot, rc=True)
    assert [(int(f["m"]), int(f["n"]), int(f["l"])) for f in found] == [(1, 1, 2), (1, 1, 4), (1, 1, 6)]
    assert score.tolist() == [6.0, 0.875, 6.0]


def test_sequential_rounding_differs_from_exact_sum():
    """A case where rounding after every add (the reference, :29) is not the rounded exact sum."""
    # 2048 + 1 + 1: sequential fp16 -> 2048 (each +1 is a tie to even), exact sum 2050 is representable
    pwm = np.zeros((3, 4, 1), dtype=np.float16)
    pwm[0, 0, 0], pwm[1, 0, 0], pwm[2, 0, 0] = 2048, 1, 1
    onehot = np.zeros((1, 12), dtype=np.float32)
    onehot[0, [0, 4, 8]] = 1  # AAA
    found, score = so.get_pos_scores_arr(pwm, np.array([3]), onehot)
    assert score.tolist() == [2048.0]
    # and a hit decision that flips: -2048 - 1 + 2049 ... sequential: (-2048-1) = -2048 (tie->even), +2050 = 2
    pwm[0, 0, 0], pwm[1, 0, 0], pwm[2, 0, 0] = -2048, -1, 2050
    found, score = so.get_pos_scores_arr(pwm, np.array([3]), onehot)
    assert score.tolist() == [2.0]  # exact sum is 1


def test_literal_equals_numpy_and_gather(pkg):
    sy = pkg.synth
    codes = sy.gen_codes(9, 37, 3)
    codes[2, 5] = 4  # an all-zero column
    pwms, lens = sy.gen_pwm_bank(7, 4, len_lo=3, len_hi=11)
    bank = sy.pad_bank(pwms, lens)
    oh16 = sy.codes_to_onehot(codes).astype(np.float16)
    lit = so.greedy_search(bank, lens, oh16)
    ref = so.greedy_search_numpy(bank, lens, oh16)
    assert np.array_equal(lit.view(np.uint16), ref.view(np.uint16))
    g = so.scan_gather(bank, lens, codes)
    assert np.array_equal(g.view(np.uint16), lit[: g.shape[0]].view(np.uint16))
    assert not lit[g.shape[0]:].any()  # the rest of the 4L third dim stays zero (:25, :75)


def test_reverse_complement_identity(pkg):
    """scan_rc(seq)[l] == scan_fwd(revcomp(seq))[L - len - l + 2]."""
    sy = pkg.synth
    L = 41
    codes = sy.gen_codes(6, L, 11)
    pwms, lens = sy.gen_pwm_bank(5, 12, len_lo=5, len_hi=9)
    bank = sy.pad_bank(pwms, lens)
    rc_codes = (3 - codes[:, ::-1]).astype(np.uint8)
    f_rc, s_rc = so.get_pos_scores_arr(bank, lens, sy.codes_to_onehot(codes), rc=True)
    f_fw, s_fw = so.get_pos_scores_arr(bank, lens, sy.codes_to_onehot(rc_codes), rc=False)
    # The identity holds for the exact sum; with per-add rounding the order of adds is reversed,
    # so compare hit sets only where the two orders agree on the sign, and require >= 99 % overlap.
    a = {(int(f["m"]), int(f["n"]), int(f["l"])) for f in f_rc}
    b = {(int(f["m"]), int(f["n"]), L - int(lens[f["m"] - 1]) - int(f["l"]) + 2) for f in f_fw}
    assert len(a & b) >= 0.99 * max(len(a), len(b), 1)


def test_record_order_is_batched_column_major(pkg):
    sy = pkg.synth
    codes = sy.gen_codes(23, 30, 5)
    pwms, lens = sy.gen_pwm_bank(6, 6, len_lo=4, len_hi=8, alpha=0.8)
    bank = sy.pad_bank(pwms, lens)
    found, _ = so.get_pos_scores_arr(bank, lens, sy.codes_to_onehot(codes), batch_size=10)
    key = [((int(f["n"]) - 1) // 10, int(f["l"]), int(f["n"]), int(f["m"])) for f in found]
    assert key == sorted(key) and len(key) > 50


def test_golden_fixture(pkg):
    path = os.path.join(HERE, "golden", "scan_small.npz")
    g = np.load(path)
    for rc in (0, 1):
        found, score = so.get_pos_scores_arr(g["bank"], g["lens"], g["onehot"], rc=bool(rc), batch_size=int(g["batch"]))
        assert np.array_equal(found, g[f"found_rc{rc}"].view(so.HIT_DTYPE).reshape(-1))
        assert np.array_equal(score.view(np.uint16), g[f"score_rc{rc}"])


@pytest.mark.parametrize("N,L,K,lo,hi,batch", [(120, 100, 37, 6, 14, 50), (60, 200, 200, 12, 12, 5000), (50, 64, 9, 3, 20, 7),
                                               (30, 150, 20, 33, 64, 11), (40, 500, 64, 20, 20, 5000)])
def test_vectorised_port_equals_the_literal_restatement(N, L, K, lo, hi, batch):
    """oracle_get_pos_scores_arr_fast (AVX2/F16C, 8 PWMs per register, no dense tensor) is what bench.py times as the CPU
    baseline and what the larger GPU parity cases are checked against: it must give the literal loop's records, order
    and binary16 score bits, both strands, all-zero columns and ragged PWM lengths included."""
    from _pkg import load_pkg

    sy = load_pkg().synth
    codes = sy.gen_codes(N, L, 5 + N, n_plant=3, k=10)
    codes[::7, ::5] = 4
    pwms, lens = sy.gen_pwm_bank(K, 3 + K, len_lo=lo, len_hi=hi, alpha=0.5)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)
    for rc in (False, True):
        fast = so.get_pos_scores_arr_fast(bank, lens, codes, rc=rc, batch_size=batch)
        if fast is None:
            pytest.skip("host CPU lacks AVX2/F16C")
        lit = so.get_pos_scores_arr(bank, lens, onehot, rc=rc, batch_size=batch)
        assert len(lit[0]) > 0
        assert np.array_equal(fast[0], lit[0])
        assert np.array_equal(fast[1].view(np.uint16), lit[1].view(np.uint16))
