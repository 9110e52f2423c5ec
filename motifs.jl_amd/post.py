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


def sweep_counts(ctx, hits_t, scores_t, n, thr):
    """counts[m][j] = get_hits(scores of motif m, thr[m][j]) for the whole sweep at once."""
    torch = _torch()
    K, T = thr.shape
    thr_t = torch.from_numpy(np.ascontiguousarray(thr).view(np.int16)).to(hits_t.device)
    counts = torch.zeros((K, T), dtype=torch.int64, device=hits_t.device)
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
    kept = ctx.hits_filter_dev(hits_t.data_ptr(), scores_t.data_ptr(), n, K, th_t.data_ptr(), oh.data_ptr(), os_.data_ptr())
    return oh, os_, kept


def posdicts2countmats(ctx, strands, codes_dev_ptr, L, lens, maxlen, n0=0, ps=0.01):
    """_h6:26-37: count matrices (4, len) per motif from the forward and reverse-strand record arrays.
    `strands`: [(hits_t, n_records, comp_flag), ...].  Returns a list of (4, len) Float16 matrices (count + ps)."""
    torch = _torch()
    K = len(lens)
    counts = torch.zeros((K, maxlen, 4), dtype=torch.int32, device="cuda")
    for hits_t, n, comp in strands:
        ctx.hits_count_matrices_dev(hits_t.data_ptr(), n, codes_dev_ptr, L, n0, lens, K, maxlen, comp, counts.data_ptr())
    ctx.synchronize()
    c = counts.cpu().numpy().astype(np.float32)
    return [(c[k, : int(lens[k]), :].T + np.float32(ps)).astype(np.float16) for k in range(K)]
