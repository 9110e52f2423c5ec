"""Synthetic inputs of SURVEY.md §8(d): i.i.d. uniform ACGT reads (optionally
with planted k-mers) and PWM banks built the way the reference builds them
(src/inference/_s1_make_motifs.jl:68-83: pfm = (cnt + 0.01) / (n + 0.04),
pwm = log2(pfm / bg), stored as Float16)."""
import numpy as np

SEED_BASE = 20260101


def gen_codes(N, L, seed, n_plant=0, k=12, frac=0.2):
    """uint8 (N, L) codes 0..3 = A,C,G,T."""
    rng = np.random.Generator(np.random.PCG64(seed))
    codes = rng.integers(0, 4, size=(N, L), dtype=np.uint8)
    if n_plant > 0 and L >= k:
        kmers = rng.integers(0, 4, size=(n_plant, k), dtype=np.uint8)
        pick = np.nonzero(rng.random(N) < frac)[0]
        which = rng.integers(0, n_plant, size=pick.size)
        offs = rng.integers(0, L - k + 1, size=pick.size)
        for r, w, o in zip(pick, which, offs):
            codes[r, o:o + k] = kmers[w]
    return codes


def codes_to_onehot(codes, dtype=np.float32):
    """(N, L) codes -> the bytes of the reference's (4L, 1, N) column-major
    one-hot matrix (loadfasta/helpers.jl:110-139), as a numpy (N, 4L) array."""
    N, L = codes.shape
    out = np.zeros((N, L, 4), dtype=dtype)
    valid = codes < 4
    n_idx, p_idx = np.nonzero(valid)
    out[n_idx, p_idx, codes[n_idx, p_idx]] = 1
    return out.reshape(N, 4 * L)


def gen_count_matrices(K, lens, seed, n_sites=100, alpha=0.3):
    """K random count matrices (4, len) with `n_sites` sites per column; columns
    are drawn from Dirichlet(alpha) so the motifs carry information."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mats = []
    for k in range(K):
        p = rng.dirichlet([alpha] * 4, size=int(lens[k]))          # (len, 4)
        cnt = np.stack([rng.multinomial(n_sites, pi) for pi in p], axis=1).astype(np.float32)  # (4, len)
        mats.append(cnt)
    return mats


def countmat2pwm(cnt, bg=(0.25, 0.25, 0.25, 0.25), ps=0.01):
    """_s1_make_motifs.jl:68-76: pfm in Float16, then log2(pfm ./ bg) in Float16."""
    cnt = np.asarray(cnt, dtype=np.float32)
    pfm = ((cnt + np.float32(ps)) / (cnt.sum(axis=0, keepdims=True) + np.float32(4 * ps))).astype(np.float16)
    bg16 = np.asarray(bg, dtype=np.float16).reshape(4, 1)
    ratio = (pfm.astype(np.float32) / bg16.astype(np.float32)).astype(np.float16)   # Float16 division
    return np.log2(ratio.astype(np.float32)).astype(np.float16)                       # Float16 log2


def gen_pwm_bank(K, seed, len_lo=12, len_hi=12, alpha=0.3):
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    lens = rng.integers(len_lo, len_hi + 1, size=K).astype(np.int64)
    pwms = [countmat2pwm(c) for c in gen_count_matrices(K, lens, seed, alpha=alpha)]
    return pwms, lens


def pad_bank(pwms, lens):
    """List of (4, len) Float16 -> padded bank with the bytes of the reference's
    (K, 4, maxlen) column-major array (_h3_1_alignment.jl:66-69), numpy (maxlen, 4, K)."""
    K = len(pwms)
    maxlen = int(max(lens))
    out = np.zeros((maxlen, 4, K), dtype=np.float16)
    for i, p in enumerate(pwms):
        out[: p.shape[1], :, i] = np.asarray(p, dtype=np.float16).T
    return out
