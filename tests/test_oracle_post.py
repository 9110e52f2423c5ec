"""CPU tests: the FASTA reader of the library (host code, no GPU call) against the restatement of
`reading` (loadfasta/helpers.jl:83-108), and small pins of oracle/post_oracle.py."""
import numpy as np
import pytest

from oracle import post_oracle as po

FASTA = """>s1 first
ACGTAC
GT
>s2
ACGTNCGT
>s3
acgtacgt
>s4 short
ACGT
>s5
TTTTGGGG
"""


def test_fasta_read_matches_reading(pkg, tmp_path):
    p = tmp_path / "a.fa"
    p.write_text(FASTA)
    reads = po.read_fasta(str(p))
    assert reads == ["ACGTACGT", "ACGTACGT", "TTTTGGGG"]          # N dropped, other length dropped, upper-cased
    got = pkg._lib.fasta_read(str(p))
    assert np.array_equal(got, po.reads_to_codes(reads))
    # the 100000 cap is applied before the equal-length filter (helpers.jl:95-98)
    assert np.array_equal(pkg._lib.fasta_read(str(p), max_entries=2), po.reads_to_codes(po.read_fasta(str(p), 2)))
    assert len(po.read_fasta(str(p), 3)) == 2                     # s1, s3, (s4 dropped by length)


def test_fasta_reader_on_a_messy_file(pkg, tmp_path):
    """The threaded reader against the restatement of `reading` on 3000 records: wrapped lines, mixed case, reads with N,
    other lengths, a header without a sequence at the end, the cap before the length filter."""
    import random

    rnd = random.Random(7)
    parts = []
    for i in range(3000):
        s_ = "".join(rnd.choice("ACGTacgt") for _ in range(90))
        if i % 17 == 0:
            s_ = s_[:10] + rnd.choice("Nn") + s_[11:]
        if i % 23 == 0:
            s_ = s_[:30]
        w = rnd.choice([25, 60, 90])
        parts.append(">r%d some text\n" % i + "\n".join(s_[j:j + w] for j in range(0, len(s_), w)) + "\n")
    parts.append(">last_without_sequence")
    p = tmp_path / "messy.fa"
    p.write_text("".join(parts))
    for cap in (100000, 1500, 1):
        want = po.read_fasta(str(p), cap)
        got = pkg._lib.fasta_read(str(p), max_entries=cap)
        if want and len(want[0]) > 0:
            assert np.array_equal(got, po.reads_to_codes(want)), cap
        else:
            assert got.shape[0] == len(want)
    # text before the first '>' is a record too (its first line plays the header): here an empty read, so only empty reads survive
    p.write_text("ACGTACGTAC\n" + "".join(parts))
    assert pkg._lib.fasta_read(str(p)).shape == (len(po.read_fasta(str(p))), 0)
    # an invalid base deep in the file: the first offending read is named
    bad = "".join(parts[:2000]) + ">bad\n" + "ACGT" * 10 + "R" + "ACGT" * 12 + "A\n" + "".join(parts[2000:])
    p.write_text(bad)
    with pytest.raises(pkg._lib.MotifsError) as e:
        pkg._lib.fasta_read(str(p))
    assert "'R' at 41" in str(e.value)


def test_fasta_errors(pkg, tmp_path):
    p = tmp_path / "b.fa"
    p.write_text(">x\nACGR\n")
    with pytest.raises(pkg._lib.MotifsError):
        pkg._lib.fasta_read(str(p))
    with pytest.raises(pkg._lib.MotifsError):
        pkg._lib.fasta_read(str(tmp_path / "missing.fa"))


def test_threshold_sweep_is_float16():
    t = po.threshold_sweep(np.float16(0.1), np.float16(2.2))
    assert t.dtype == np.float16 and t[0] == np.float16(0.1)
    assert np.all(np.diff(t.astype(np.float32)) > 0) and t[-1] < np.float16(2.2) <= np.float16(t[-1] + np.float16(0.5))


def test_countmats_reverse_complement():
    codes = np.array([[0, 1, 2, 3, 0, 0]], dtype=np.uint8)        # ACGTAA
    lens = np.array([3])
    fwd = po.countmats(np.array([1]), np.array([1]), np.array([2]), False, codes, lens, 1, 3)   # window CGT
    assert fwd[0].tolist() == [[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    rc = po.countmats(np.array([1]), np.array([1]), np.array([2]), True, codes, lens, 1, 3)     # revcomp(CGT) = ACG
    assert rc[0].tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]


# ---- src/inference/_2_enumerate.jl -----------------------------------------------------------------------------
def _recs(rows):
    from motifs_jl_amd._lib import CODE_DTYPE  # registered by conftest's pkg loader

    out = np.zeros(len(rows), dtype=CODE_DTYPE)
    for i, (p, f, s, m) in enumerate(rows):
        out[i] = (p, f, s, np.float16(m))
    return out


def test_julia_quantile_known_answers(pkg):
    v = np.array([1.0, 2.0, 3.0, 4.0, 5.0], dtype=np.float16)
    # aleph = (n-1)p + 1: p=0.5 -> 3.0 exactly; p=0.35 -> j=2, gamma=0.4 -> 2 + 0.4*1
    assert po.julia_quantile_f16(v, 0.5) == 3.0
    assert abs(po.julia_quantile_f16(v, 0.35) - 2.4) < 1e-12
    assert po.julia_quantile_f16(v[:1], 0.3) == 1.0
    # same as numpy's default (type-7) quantile when b - a is exact in Float16
    rng = np.random.default_rng(3)
    w = (rng.integers(1, 2000, size=501) / 8.0).astype(np.float16)
    for p in (0.01, 0.05, 0.25, 0.5, 0.75):
        assert abs(po.julia_quantile_f16(w, p) - np.quantile(w.astype(np.float64), p)) < 1e-9


def test_scanning_ranges_quirks_and_vectorised_form(pkg):
    post = pkg.post
    # seq 1,1,2,3,3: pushes (1,2) at i=3, (3,3) at i=4; the last range (4,5) is never pushed
    r = _recs([(5, 1, 1, 1), (9, 2, 1, 1), (3, 1, 2, 1), (4, 1, 3, 1), (8, 2, 3, 1)])
    assert po.scanning_ranges(r) == [(1, 2), (3, 3)]
    # a gap (sequence 2 has no component): the counter lags and single-element ranges follow until it catches up
    g = _recs([(1, 1, 1, 1), (2, 1, 3, 1), (3, 1, 3, 1), (4, 1, 3, 1), (5, 1, 4, 1), (6, 1, 5, 1)])
    assert po.scanning_ranges(g) == [(1, 1), (2, 2), (3, 4), (5, 5)]
    rng = np.random.default_rng(11)
    for trial in range(30):
        n = int(rng.integers(1, 80))
        seq = np.sort(rng.integers(1, 25, size=n)).astype(np.uint32)
        rr = _recs([(1, 1, int(s), 1) for s in seq])
        want = po.scanning_ranges(rr)
        st, ln = post.scanning_ranges(seq)
        assert [(int(a) + 1, int(a) + int(b)) for a, b in zip(st, ln)] == want


def test_enumerate_triplets_by_hand(pkg):
    # one range of four components; stable sort by position puts (3,f2) (5,f1) (5,f3) (9,f1)
    r = _recs([(5, 1, 1, 2.0), (9, 1, 1, 2.0), (3, 2, 1, 2.0), (5, 3, 1, 2.0), (7, 1, 2, 2.0)])
    H = po.enumerate_triplets(r, po.scanning_ranges(r), h=12)
    assert list(H.keys()) == [(2, 1, 3, 2, 2, 14), (2, 1, 1, 2, 6, 18), (2, 3, 1, 2, 6, 18), (1, 3, 1, 0, 4, 16)]
    assert H[(2, 1, 3, 2, 2, 14)] == [(1, 3, False)] and H[(1, 3, 1, 0, 4, 16)] == [(1, 5, False)]
    f, thr = po.filter_code_components_using_quantile(_recs([(1, 1, 1, 1.0), (2, 1, 1, 2.0), (3, 1, 1, 3.0)]), 0.5)
    assert thr == 2.0 and list(f["position"]) == [3]
