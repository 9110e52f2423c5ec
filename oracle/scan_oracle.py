"""ctypes front-end of oracle/scan_oracle.c plus a tiny pure-numpy restatement.

TEST INFRASTRUCTURE ONLY — never imported by the product package.
PARITY UNPINNED (the reference has no tests or fixtures; see scan_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle_scan.so")
HIT_DTYPE = np.dtype([("m", "<u4"), ("n", "<u4"), ("l", "<u4")])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        h = C.CDLL(SO)
        p, i64, i = C.c_void_p, C.c_int64, C.c_int
        h.oracle_greedy_search.restype = None
        h.oracle_greedy_search.argtypes = [p, p, p, i, i, i64, i, p]
        h.oracle_get_pos_scores_arr.restype = i64
        h.oracle_get_pos_scores_arr.argtypes = [p, p, i, i, p, i64, i, i, i, p, p, i64]
        h.oracle_get_pos_scores_arr_fast.restype = i64
        h.oracle_get_pos_scores_arr_fast.argtypes = [p, p, i, i, p, i64, i, i, i, p, p, i64]
        h.oracle_scan_gather.restype = None
        h.oracle_scan_gather.argtypes = [p, p, i, p, i64, i, i, p]
        h.oracle_num_threads.restype = i
        h.oracle_set_threads.restype = None
        h.oracle_set_threads.argtypes = [i]
        try:
            h.oracle_set_threads(min(len(os.sched_getaffinity(0)), 64))
        except AttributeError:
            pass
        h.oracle_h_add.restype = C.c_uint16
        h.oracle_h_add.argtypes = [C.c_uint16, C.c_uint16]
        h.oracle_f32_to_f16.restype = C.c_uint16
        h.oracle_f32_to_f16.argtypes = [C.c_float]
        _lib = h
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def greedy_search(pwms, lens, data16):
    """pwms: numpy (maxlen,4,K) f16 [bytes of Julia (K,4,maxlen)]; data16: (N, 4L) f16 one-hot.
    Returns pos_scores as numpy (4L, N, K) f16 [bytes of Julia (K, N, 4L)]."""
    pwms = np.ascontiguousarray(pwms, dtype=np.float16)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    data16 = np.ascontiguousarray(data16, dtype=np.float16)
    maxlen, _, K = pwms.shape
    N, L4 = data16.shape
    out = np.zeros((L4, N, K), dtype=np.float16)
    lib().oracle_greedy_search(_ptr(pwms), _ptr(data16), _ptr(lens), K, maxlen, N, L4, _ptr(out))
    return out


def get_pos_scores_arr(pwms, lens, data_f32, rc=False, batch_size=5000):
    """Restates _h3_1_alignment.jl:57-87 for one strand.  Returns (found_record, score_record)."""
    pwms = np.ascontiguousarray(pwms, dtype=np.float16)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    data_f32 = np.ascontiguousarray(data_f32, dtype=np.float32)
    maxlen, _, K = pwms.shape
    N, L4 = data_f32.shape
    cap = 1 << 16
    while True:
        found = np.zeros(cap, dtype=HIT_DTYPE)
        score = np.zeros(cap, dtype=np.float16)
        n = lib().oracle_get_pos_scores_arr(_ptr(pwms), _ptr(lens), K, maxlen, _ptr(data_f32), N, L4, int(bool(rc)),
                                            int(batch_size), _ptr(found), _ptr(score), cap)
        if n <= cap:
            return found[:n], score[:n]
        cap = int(n)


def get_pos_scores_arr_fast(pwms, lens, codes, rc=False, batch_size=5000, cap_hint=None):
    """The optimised CPU form (AVX2/F16C, 8 PWMs per register, OpenMP over start positions): same records, same order,
    same binary16 rounding sequence as get_pos_scores_arr.  codes: (N, L) uint8.  None if the CPU lacks F16C."""
    pwms = np.ascontiguousarray(pwms, dtype=np.float16)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    maxlen, _, K = pwms.shape
    N, L = codes.shape
    cap = int(cap_hint) if cap_hint else 1 << 16
    while True:
        found = np.zeros(cap, dtype=HIT_DTYPE)
        score = np.zeros(cap, dtype=np.float16)
        n = lib().oracle_get_pos_scores_arr_fast(_ptr(pwms), _ptr(lens), K, maxlen, _ptr(codes), N, L, int(bool(rc)),
                                                 int(batch_size), _ptr(found), _ptr(score), cap)
        if n < 0:
            return None
        if n <= cap:
            return found[:n], score[:n]
        cap = int(n)


def scan_gather(pwms, lens, codes, Lout=None):
    """Gather formulation (optimised CPU leg).  Returns numpy (Lout, N, K) f16."""
    pwms = np.ascontiguousarray(pwms, dtype=np.float16)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    _, _, K = pwms.shape
    N, L = codes.shape
    if Lout is None:
        Lout = L - int(lens.min()) + 1
    out = np.zeros((Lout, N, K), dtype=np.float16)
    lib().oracle_scan_gather(_ptr(pwms), _ptr(lens), K, _ptr(codes), N, L, Lout, _ptr(out))
    return out


def num_threads():
    return lib().oracle_num_threads()


def physical_cores():
    """Distinct physical cores among the CPUs this process may run on (two SMT siblings share one set of vector ports, so
    the thread count overstates what the baseline had)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                seen.add(fh.read().strip())
        except OSError:
            seen.add(str(c))
    return len(seen)


def greedy_search_numpy(pwms, lens, data16):
    """Pure-numpy restatement of greedy_search! (:18-36) with numpy's Float16
    arithmetic (correctly rounded per operation).  Small cases only."""
    pwms = np.asarray(pwms, dtype=np.float16)
    data16 = np.asarray(data16, dtype=np.float16)
    maxlen, _, K = pwms.shape
    N, L4 = data16.shape
    L = L4 // 4
    out = np.zeros((L4, N, K), dtype=np.float16)
    for k in range(K):
        for n in range(N):
            for l in range(1, L4 + 1):
                if not l <= L - lens[k] + 1:
                    continue
                acc = np.float16(0)
                for ind, i in enumerate(range(l, l + int(lens[k]))):
                    for a in range(4):
                        acc = np.float16(acc + np.float16(pwms[ind, a, k] * data16[n, (i - 1) * 4 + a]))
                out[l - 1, n, k] = acc if acc > 0 else np.float16(0)
    return out
