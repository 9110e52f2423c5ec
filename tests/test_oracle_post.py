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
