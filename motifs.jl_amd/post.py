"""Host mirror of the steps right after the scan:
`filter_positions_scores_usecomp!`'s Fisher sweep (src/inference/_s2_filter_pos_w_scores.jl:90-139) and
`posdicts2countmats` (src/inference/_h6_positions2countmat.jl:26-55), working on the device-resident record
arrays of `motifs_pwm_scan_hits_dev` instead of Julia Dicts.  torch is plumbing (device buffers)."""
import numpy as np

from . import _lib

score_thresh_increment = np.float16(0.5)      # _0_const.jl:27


def _torch():
    import torch

    return torch


def score_range(ctx, hits_t, scores_t, n, K):
    """get_min_score / get_max_score over one record array -> (min, max) Float16 arrays of K."""
    torch = _torch()
    mn = torch.empty(K, dtype=torch.int16, device=hits_t.device)
    mx = torch.empty(K, dtype=torch.int16, device=hits_t.device)
    _torch_first(mn)
    ctx.hits_minmax_dev(hits_t.data_ptr(), scores_t.data_ptr(), n, K, mn.data_ptr(), mx.data_ptr())
    ctx.synchronize()
    return mn.cpu().numpy().view(np.float16), mx.cpu().numpy().view(np.float16)


def sweep_thresholds(min_scores, max_scores):
    """The thresholds `get_best_thresh` visits per motif (Float16 accumulation, :100-112), padded with +Inf."""
    rows = []
    for lo, hi in zip(min_scores, max_scores):
        t, row = np.float16(lo), []
        while np.isfinite(t) and t < np.float16(hi):
            row.append(t)
            nt = np.float16(t + score_thresh_increment)
            if nt == t:
                break
            t = nt
        rows.append(row)
    T = max(1, max(len(r) for r in rows))
    thr = np.full((len(rows), T), np.inf, dtype=np.float16)
    for i, r in enumerate(rows):
        thr[i, :len(r)] = r
    return thr, [len(r) for r in rows]


def _torch_first(t):
    """The library's kernels run on the context's stream, torch's fills and uploads on torch's: wait for torch's before the
    library reads what it produced (a no-op when the two are the same stream and idle)."""
    _torch().cuda.current_stream(t.device).synchronize()


def sweep_counts(ctx, hits_t, scores_t, n, thr):
    """counts[m][j] = get_hits(scores of motif m, thr[m][j]) for the whole sweep at once."""
    torch = _torch()
    K, T = thr.shape
    thr_t = torch.from_numpy(np.ascontiguousarray(thr).view(np.int16)).to(hits_t.device)
    counts = torch.zeros((K, T), dtype=torch.int64, device=hits_t.device)
    _torch_first(counts)
    ctx.hits_threshold_counts_dev(hits_t.data_ptr(), scores_t.data_ptr(), n, K, thr_t.data_ptr(), T, counts.data_ptr())
    ctx.synchronize()
    return counts.cpu().numpy()


def get_best_thresh_fisher(counts_fg, counts_bg, thr_row, n_thr, asum, min_score):
    """The sweep branch of get_best_thresh (:97-113): the threshold with the smallest right-tail Fisher p-value."""
    from scipy.stats import fisher_exact

    best_thresh, best_p = np.float16(min_score), 1.0
    for j in range(n_thr):
        a, b = int(counts_fg[j]), int(counts_bg[j])
        p = fisher_exact([[a, asum - a], [b, asum - b]], alternative="greater")[1]
        if p < best_p:
            best_p, best_thresh = p, thr_row[j]
    return best_thresh


def filter_by_thresh(ctx, hits_t, scores_t, n, thresh):
    """filter_position_by_best_thresh! (:116-125) on a record array; returns (hits, scores, n_kept) on the device."""
    torch = _torch()
    K = len(thresh)
    th_t = torch.from_numpy(np.ascontiguousarray(thresh, dtype=np.float16).view(np.int16)).to(hits_t.device)
    oh = torch.empty_like(hits_t)
    os_ = torch.empty_like(scores_t)
    _torch_first(th_t)
    kept = ctx.hits_filter_dev(hits_t.data_ptr(), scores_t.data_ptr(), n, K, th_t.data_ptr(), oh.data_ptr(), os_.data_ptr())
    return oh, os_, kept


def posdicts2countmats(ctx, strands, codes_dev_ptr, L, lens, maxlen, n0=0, ps=0.01):
    """_h6:26-37: count matrices (4, len) per motif from the forward and reverse-strand record arrays.
    `strands`: [(hits_t, n_records, comp_flag), ...].  Returns a list of (4, len) Float16 matrices (count + ps)."""
    torch = _torch()
    K = len(lens)
    counts = torch.zeros((K, maxlen, 4), dtype=torch.int32, device="cuda")
    _torch_first(counts)
    for hits_t, n, comp in strands:
        ctx.hits_count_matrices_dev(hits_t.data_ptr(), n, codes_dev_ptr, L, n0, lens, K, maxlen, comp, counts.data_ptr())
    ctx.synchronize()
    c = counts.cpu().numpy().astype(np.float32)
    return [(c[k, : int(lens[k]), :].T + np.float32(ps)).astype(np.float16) for k in range(K)]


# ---- consumers of the code records (src/inference/_2_enumerate.jl) ---------------------------------
def code_quantile(ctx, recs_t, n, p):
    """Statistics.quantile of the Float16 magnitudes (alpha = beta = 1) from the device histogram of their bit
    patterns: the two order statistics it interpolates between, then a + gamma*(b - a) as Julia evaluates it."""
    torch = _torch()
    hist = torch.empty(65536, dtype=torch.int32, device=recs_t.device)
    _torch_first(hist)
    ctx.codes_mag_histogram_dev(recs_t.data_ptr(), n, hist.data_ptr())
    ctx.synchronize()
    hcnt = hist.cpu().numpy().astype(np.int64)
    # ascending value order of the bit patterns: negative values (sign bit set) descend with their pattern
    bits = np.concatenate([np.arange(0xFFFF, 0x7FFF, -1), np.arange(0, 0x8000)]).astype(np.uint16)
    cum = np.cumsum(hcnt[bits])
    aleph = n * p + (1.0 - p)
    j = int(min(max(np.trunc(aleph), 1), max(n - 1, 1)))
    g = float(min(max(aleph - j, 0.0), 1.0))

    def kth(k):                       # k-th smallest, 1-based
        return np.array([bits[int(np.searchsorted(cum, k, side="left"))]], dtype=np.uint16).view(np.float16)[0]

    a = kth(j) if n > 1 else kth(1)
    b = kth(j + 1) if n > 1 else a
    return float(a) + g * float(np.float16(b - a))


def filter_code_components(ctx, recs_t, n, p):
    """filter_code_components_using_quantile! (:10-13) on a device record array; returns (filtered tensor, count, threshold)."""
    torch = _torch()
    thr = code_quantile(ctx, recs_t, n, p)
    out = torch.empty_like(recs_t)
    _torch_first(out)
    m = ctx.codes_filter_dev(recs_t.data_ptr(), n, thr, out.data_ptr())
    return out, m, thr


def scanning_ranges(seq):
    """get_scanning_range_of_filtered_code_components (:25-35) on the `seq` column (host): 0-based starts and
    lengths of the ranges the reference pushes (its counter advances by one per mismatch; the last range is
    never pushed).  c[i+1] = min(seq[i], c[i] + 1), so the counter is a running minimum and the loop vectorises."""
    seq = np.asarray(seq, dtype=np.int64)
    n = len(seq)
    if n == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    i = np.arange(n, dtype=np.int64)
    c_next = np.minimum(np.minimum.accumulate(seq - i) + i, i + 2)          # counter after element i (it starts at 1)
    c = np.concatenate([[1], c_next[:-1]])
    ev = np.nonzero(seq != c)[0]                                            # pushes happen at these elements
    starts = np.concatenate([[0], ev[:-1]]) if len(ev) else np.zeros(0, np.int64)
    lens = ev - starts
    return starts.astype(np.uint32), lens.astype(np.uint32)


def enumerate_triplets(ctx, recs_t, n, h):
    """enumerate_triplets (:50-65) on a (filtered) device record array.  Returns a dict of numpy arrays describing
    the Dictionary the reference builds: keys (U, 6) = (f1, f2, f3, d12, d13, len) in insertion order, counts,
    offsets, and values (seq_num, pos) grouped by key in insertion order."""
    torch = _torch()
    dev = recs_t.device
    seq = recs_t.view(torch.uint8).view(-1, 12)[:n, 4:8].contiguous().view(torch.int32).cpu().numpy().ravel().astype(np.int64)
    starts, lens = scanning_ranges(seq)
    nr = len(starts)
    empty = {"keys": np.zeros((0, 6), np.int64), "counts": np.zeros(0, np.int64), "offsets": np.zeros(1, np.int64),
             "values": np.zeros(0, dtype=_lib.TRIPLET_VAL_DTYPE), "n_triplets": 0, "ranges": (starts, lens)}
    if nr == 0:
        return empty
    st = torch.from_numpy(starts.view(np.int32)).to(dev)
    ln = torch.from_numpy(lens.view(np.int32)).to(dev)
    offs = torch.empty(nr, dtype=torch.int64, device=dev)
    _torch_first(ln)
    total = ctx.triplets_offsets_dev(ln.data_ptr(), nr, offs.data_ptr())
    if total == 0:
        return empty
    keys = torch.empty(total, dtype=torch.int64, device=dev)
    vals = torch.empty(total, dtype=torch.int64, device=dev)
    ctx.triplets_enumerate_dev(recs_t.data_ptr(), st.data_ptr(), ln.data_ptr(), nr, h, offs.data_ptr(), keys.data_ptr(), vals.data_ptr(), total)
    uniq = torch.empty(total, dtype=torch.int64, device=dev)
    first = torch.empty(total, dtype=torch.int64, device=dev)
    counts = torch.empty(total, dtype=torch.int64, device=dev)
    goff = torch.empty(total, dtype=torch.int64, device=dev)
    perm = torch.empty(total, dtype=torch.int64, device=dev)
    U = ctx.triplets_group_dev(keys.data_ptr(), total, uniq.data_ptr(), first.data_ptr(), counts.data_ptr(), goff.data_ptr(), perm.data_ptr())
    uk = uniq[:U].cpu().numpy().view(np.uint64)
    d13 = (uk & 0xFFFF).astype(np.int64)
    kk = np.stack([(uk >> 48) & 0xFF, (uk >> 40) & 0xFF, (uk >> 32) & 0xFF, (uk >> 16) & 0xFFFF, uk & 0xFFFF], axis=1).astype(np.int64)
    kk = np.concatenate([kk, (d13 + h)[:, None]], axis=1)
    cnt = counts[:U].cpu().numpy()
    v = vals[perm].cpu().numpy().view(_lib.TRIPLET_VAL_DTYPE)
    return {"keys": kk, "counts": cnt, "offsets": np.concatenate([[0], np.cumsum(cnt)]), "values": v, "n_triplets": int(total),
            "ranges": (starts, lens)}
