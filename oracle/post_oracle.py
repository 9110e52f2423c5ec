"""CPU restatements of the steps either side of the scan.  TEST INFRASTRUCTURE ONLY.
PARITY UNPINNED (the reference has no fixtures); pinned by tests/test_oracle_post.py.

  reading / read_fasta      src/loadfasta/helpers.jl:83-108
  get_hits / get_min_score / get_max_score / filter_position_by_best_thresh!
                            src/inference/_s2_filter_pos_w_scores.jl:3-35, :116-125
  posdicts2countmats        src/inference/_h6_positions2countmat.jl:40-55 (+ submat_comlement, _3_make_pfms.jl:49-52)
  filter_code_components_using_quantile!, get_scanning_range_of_filtered_code_components, insert_H!,
  enumerate_triplets        src/inference/_2_enumerate.jl:10-13, :25-35, :37-46, :50-65
                            (Statistics.quantile as in Julia 1.9's stdlib: alpha = beta = 1)
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


# ---- src/inference/_2_enumerate.jl ---------------------------------------------------------------------------
def julia_quantile_f16(mags, p):
    """Statistics.quantile(v::Vector{Float16}, p::Float64) with the default alpha = beta = 1: the sorted copy,
    aleph = n*p + (1 - p), j = clamp(trunc(aleph), 1, n-1), gamma = clamp(aleph - j, 0, 1),
    a + gamma*(b - a) where b - a is a Float16 subtraction and the rest Float64."""
    v = np.sort(np.asarray(mags, dtype=np.float16))
    n = len(v)
    m = 1.0 + p * (1.0 - 1.0 - 1.0)
    aleph = n * p + m
    j = int(min(max(np.trunc(aleph), 1), max(n - 1, 1)))
    g = float(min(max(aleph - j, 0.0), 1.0))
    if n == 1:
        a = b = v[0]
    else:
        a, b = v[j - 1], v[j]
    return float(a) + g * float(np.float16(b - a))


def filter_code_components_using_quantile(recs, p):
    """recs: structured array (position, fil, seq, mag).  Keeps mag > quantile (:12), order preserved."""
    thr = julia_quantile_f16(recs["mag"], p)
    return recs[recs["mag"].astype(np.float64) > thr], thr


def scanning_ranges(recs):
    """get_scanning_range_of_filtered_code_components (:25-35), literally: 1-based inclusive (start, stop) pairs.
    Note the reference's quirks, kept: cur_seq only ever advances by one, and the last range is never pushed."""
    cur_seq, cur_range_start, ranges = 1, 1, []
    seq = recs["seq"]
    for i in range(1, len(recs) + 1):
        if int(seq[i - 1]) != cur_seq:
            ranges.append((cur_range_start, i - 1))
            cur_range_start = i
            cur_seq += 1
    return ranges


def enumerate_triplets(recs, ranges, h):
    """enumerate_triplets (:50-65) + insert_H! (:37-46): dict key -> list of values, both in insertion order.
    key = (f1, f2, f3, d12, d13, len), value = (seq_num = index of the range, pos of the first component, comp=False)."""
    H = {}
    for ind, (lo, hi) in enumerate(ranges, start=1):
        store = recs[lo - 1:hi]
        order = np.argsort(store["position"], kind="stable")       # sort(by = x -> x[1]) is stable
        cs = store[order]
        n = len(cs)
        for i in range(n - 2):
            for j in range(i + 1, n - 1):
                for k in range(j + 1, n):
                    d12 = int(cs["position"][j]) - int(cs["position"][i])
                    d13 = int(cs["position"][k]) - int(cs["position"][i])
                    key = (int(cs["fil"][i]), int(cs["fil"][j]), int(cs["fil"][k]), d12, d13, d13 + h)
                    H.setdefault(key, []).append((ind, int(cs["position"][i]), False))
    return H
