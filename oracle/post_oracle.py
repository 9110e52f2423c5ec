"""CPU restatements of the steps either side of the scan.  TEST INFRASTRUCTURE ONLY.
PARITY UNPINNED (the reference has no fixtures); pinned by tests/test_oracle_post.py.

  reading / read_fasta      src/loadfasta/helpers.jl:83-108
  get_hits / get_min_score / get_max_score / filter_position_by_best_thresh!
                            src/inference/_s2_filter_pos_w_scores.jl:3-35, :116-125
  posdicts2countmats        src/inference/_h6_positions2countmat.jl:40-55 (+ submat_comlement, _3_make_pfms.jl:49-52)
"""
import numpy as np


def read_fasta(path, max_entries=100000):
    text = open(path, "r").read()
    dna_reads = []
    for rec in text.split(">"):
        if rec:                                             # !isempty(i)
            splits = rec.split("\n")
            this_read = "".join(splits[1:])
            if "N" not in this_read and "n" not in this_read:
                dna_reads.append(this_read)
    if len(dna_reads) > max_entries:
        dna_reads = dna_reads[:max_entries]
    dna_reads = [s for s in dna_reads if len(s) == len(dna_reads[0])]
    return [s.upper() for s in dna_reads]


def reads_to_codes(reads):
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    return np.array([[lut[c] for c in r] for r in reads], dtype=np.uint8)


def get_hits(scores_by_motif, thresh):
    """:3-9 for one motif: number of scores > thresh (Float16 comparison)."""
    return int((scores_by_motif.astype(np.float16) > np.float16(thresh)).sum())


def minmax_by_motif(m, scores, K):
    mn = np.full(K, np.inf, dtype=np.float16)
    mx = np.full(K, -np.inf, dtype=np.float16)
    for k in range(K):
        s = scores[m == k + 1]
        if len(s):
            mn[k], mx[k] = s.min(), s.max()
    return mn, mx


def threshold_sweep(min_score, max_score, inc=np.float16(0.5)):
    """The thresholds get_best_thresh visits (:98-112): score_thresh starts at min_score and grows by
    score_thresh_increment in Float16 arithmetic while < max_score."""
    out = []
    t = np.float16(min_score)
    while t < np.float16(max_score):
        out.append(t)
        nt = np.float16(t + inc)
        if nt == t:
            break
        t = nt
    return np.array(out, dtype=np.float16)


def filter_records(m, n, l, scores, thresh):
    keep = scores.astype(np.float16) > thresh[m - 1]
    return m[keep], n[keep], l[keep], scores[keep]


def countmats(m, n, l, comp, codes, lens, K, maxlen):
    """(K, maxlen, 4) counts [numpy order of Julia's (4, maxlen, K)]: every hit adds its one-hot window;
    reverse-strand hits add reverse(window) in both dims."""
    out = np.zeros((K, maxlen, 4), dtype=np.uint32)
    for mi, ni, li in zip(m, n, l):
        ln = int(lens[mi - 1])
        w = codes[ni - 1, li - 1:li - 1 + ln]
        for ind, b in enumerate(w):
            if b > 3:
                continue
            if comp:
                out[mi - 1, ln - 1 - ind, 3 - b] += 1
            else:
                out[mi - 1, ind, b] += 1
    return out
