"""ctypes binding of include/motifs_hip.h.

Loading is lazy so that CPU-only tests can import the package; any attempt to
create a context without the shared library or without a gfx950 GPU raises —
there is no CPU fallback behind this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOTIFS_HIP_LIB") or os.path.join(HERE, "libmotifs_hip.so")   # MOTIFS_HIP_LIB: another build (A/B timing)

OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_BUFFER_TOO_SMALL, ERR_NOT_ONEHOT, ERR_NONFINITE, ERR_UNSUPPORTED, ERR_COMM = range(9)
ABI_VERSION = 3
COMM_ID_BYTES = 128
DATA_CODES_U8, DATA_ONEHOT_F32, DATA_ONEHOT_F16 = 0, 1, 2
KS_ENCODE, KS_SCAN_DENSE, KS_SCAN_COUNT, KS_SCAN_OFFSETS, KS_SCAN_FILL, KS_TRAIN_STEP, KS_TRAIN_ISTA_BWD = range(7)
SCAN_BATCH = 5000
SCAN_MAX_LEN = 64

HIT_DTYPE = np.dtype([("m", "<u4"), ("n", "<u4"), ("l", "<u4")])
# stored_code_component_t (_0_const.jl:3-4) with Julia's 12-byte isbits layout
TRIPLET_VAL_DTYPE = np.dtype({"names": ["seq_num", "pos", "comp"], "formats": ["<u4", "<u2", "<u2"], "itemsize": 8})
CODE_DTYPE = np.dtype({"names": ["position", "fil", "seq", "mag"], "formats": ["<u2", "<u2", "<u4", "<f2"],
                       "offsets": [0, 2, 4, 8], "itemsize": 12})


class HParams(C.Structure):
    """motifs_hparams == Hyperparam (model.jl:1-14)."""
    _fields_ = [("filter_len", C.c_int32), ("M", C.c_int32), ("h", C.c_int32), ("K", C.c_int32), ("q", C.c_int32),
                ("batch_size", C.c_int32), ("num_pass_xyz", C.c_int32), ("num_pass_df", C.c_int32),
                ("magnifying_factor", C.c_float), ("gamma", C.c_float)]


class MotifsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmotifs_hip status {code}: {msg}")
        self.code = code


_lib = None

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); mirrors include/motifs_hip.h declaration by declaration
SIGNATURES = {
    "motifs_abi_version": (_int, []),
    "motifs_last_error": (C.c_char_p, []),
    "motifs_ctx_create": (_int, [_int, C.POINTER(_p)]),
    "motifs_ctx_destroy": (None, [_p]),
    "motifs_ctx_set_stream": (_int, [_p, _p]),
    "motifs_ctx_get_stream": (_int, [_p, C.POINTER(_p)]),
    "motifs_ctx_use_private_stream": (_int, [_p]),
    "motifs_dev_alloc": (_int, [_p, C.c_size_t, C.POINTER(_p)]),
    "motifs_dev_free": (_int, [_p, _p]),
    "motifs_dev_upload": (_int, [_p, _p, _p, C.c_size_t]),
    "motifs_dev_download": (_int, [_p, _p, _p, C.c_size_t]),
    "motifs_dev_memset": (_int, [_p, _p, _int, C.c_size_t]),
    "motifs_ctx_set_workspace_limit": (_int, [_p, C.c_size_t]),
    "motifs_ctx_set_records_in_stream_order": (_int, [_p, _int]),
    "motifs_ctx_synchronize": (_int, [_p]),
    "motifs_ctx_enable_timing": (_int, [_p, _int]),
    "motifs_ctx_reset_timing": (_int, [_p]),
    "motifs_ctx_kernel_ms": (_int, [_p, _int, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "motifs_ctx_scan_plan": (_int, [_p, C.POINTER(C.c_int32)]),
    "motifs_model_arena_peak": (_int, [_p, C.POINTER(C.c_size_t)]),
    "motifs_codes_bytes": (C.c_size_t, [_i64, _int]),
    "motifs_codes_pitch": (_int, [_int]),
    "motifs_encode_dev": (_int, [_p, _p, _int, _i64, _int, _p, _p]),
    "motifs_pwm_scan_dense_dev": (_int, [_p, _p, _p, _int, _int, _p, _i64, _int, _p, _i64]),
    "motifs_pwm_scan_hits_dev": (
        _int,
        [_p, _p, _p, _int, _int, _p, _i64, _int, _int, _i64, _int, _p, _p, _i64, C.POINTER(_i64), _p],
    ),
    "motifs_pwm_scan_hits_both_dev": (
        _int,
        [_p, _p, _p, _int, _int, _p, _i64, _int, _i64, _int, _p, _p, _p, _p, _i64, C.POINTER(_i64), _p],
    ),
    "motifs_model_create": (_int, [_p, C.POINTER(HParams), _int, C.c_size_t, C.POINTER(_p)]),
    "motifs_model_destroy": (None, [_p]),
    "motifs_model_sizes": (_int, [_p] + [C.POINTER(_i64)] * 5),
    "motifs_model_set_params": (_int, [_p, _p, _p, _p, _p]),
    "motifs_model_get_params": (_int, [_p, _p, _p, _p, _p]),
    "motifs_model_init_random": (_int, [_p, C.c_uint64]),
    "motifs_model_loss_grad_dev": (_int, [_p, _p, _int, _p, _p, _int]),
    "motifs_model_adabelief_dev": (_int, [_p, _p, C.c_float]),
    "motifs_model_l1_syntax": (_int, [_p, C.POINTER(C.c_float)]),
    "motifs_model_train_step": (_int, [_p, _p, _int, _p, C.POINTER(C.c_float)]),
    "motifs_model_train_step_onehot": (_int, [_p, _p, _int, _p, C.POINTER(C.c_float)]),
    "motifs_comm_unique_id": (_int, [_p]),
    "motifs_comm_create": (_int, [_p, _p, _int, _int, C.POINTER(_p)]),
    "motifs_comm_create_all": (_int, [C.POINTER(_p), _int, C.POINTER(_p)]),
    "motifs_comm_destroy": (None, [_p]),
    "motifs_comm_rank": (_int, [_p, C.POINTER(_int), C.POINTER(_int)]),
    "motifs_comm_group_start": (_int, []),
    "motifs_comm_group_end": (_int, []),
    "motifs_comm_allreduce_sum_f32_dev": (_int, [_p, _p, _i64]),
    "motifs_comm_allreduce_sum_i64_dev": (_int, [_p, _p, _i64]),
    "motifs_comm_allreduce_sum_u32_dev": (_int, [_p, _p, _i64]),
    "motifs_comm_allreduce_sum_f32_to_dev": (_int, [_p, _p, _p, _i64]),
    "motifs_model_allreduce_grad": (_int, [_p, _p, _p]),
    "motifs_model_dp_grad_dev": (_int, [_p, _p, _int, _p, _p]),
    "motifs_model_dp_update_dev": (_int, [_p, _p, _i64]),
    "motifs_model_dp_train_step_all": (_int, [C.POINTER(_p), C.POINTER(_p), _int, C.POINTER(_p), C.POINTER(_int), _i64,
                                              C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "motifs_model_dp_train_step_host": (_int, [C.POINTER(_p), C.POINTER(_p), _int, _p, _int, _int, _p, C.POINTER(C.c_float)]),
    "motifs_pwm_scan_both_sharded": (
        _int,
        [C.POINTER(_p), C.POINTER(_p), _int, _p, _p, _int, _int, _p, _int, _i64, _int, _i64, _p, _p, _p, _p, _i64,
         C.POINTER(_i64), _p, _p],
    ),
    "motifs_hist_allreduce": (_int, [_p, _p, _int, _int]),
    "motifs_model_dp_train_step_dev": (_int, [_p, _p, _p, _int, _i64, _p, _p]),
    "motifs_model_retrieve_codes": (_int, [_p, _p, _int, _i64, _p, _i64, C.POINTER(_i64)]),
    "motifs_model_time_filter_scan": (_int, [_p, _p, _int, _int, C.POINTER(C.c_float)]),
    "motifs_model_time_syntax_conv": (_int, [_p, _p, _int, _int, C.POINTER(C.c_float)]),
    "motifs_model_dump": (_int, [_p, C.c_char_p, _p, _i64, C.POINTER(_i64)]),
    "motifs_fasta_read": (_int, [C.c_char_p, _i64, _p, _i64, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "motifs_hits_minmax_dev": (_int, [_p, _p, _p, _i64, _int, _p, _p]),
    "motifs_hits_threshold_counts_dev": (_int, [_p, _p, _p, _i64, _int, _p, _int, _p]),
    "motifs_hits_filter_dev": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p, C.POINTER(_i64)]),
    "motifs_hits_count_matrices_dev": (_int, [_p, _p, _i64, _p, _int, _i64, _p, _int, _int, _int, _p]),
    "motifs_codes_mag_histogram_dev": (_int, [_p, _p, _i64, _p]),
    "motifs_codes_filter_dev": (_int, [_p, _p, _i64, C.c_double, _p, C.POINTER(_i64)]),
    "motifs_triplets_offsets_dev": (_int, [_p, _p, _i64, _p, C.POINTER(_i64)]),
    "motifs_triplets_enumerate_dev": (_int, [_p, _p, _p, _p, _i64, _int, _p, _p, _p, _i64]),
    "motifs_triplets_group_dev": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, C.POINTER(_i64)]),
    "motifs_pwm_scan": (
        _int,
        [_p, _p, _p, _int, _int, _p, _int, _i64, _int, _int, _p, _p, _i64, C.POINTER(_i64), _p],
    ),
    "motifs_pwm_scan_both": (
        _int,
        [_p, _p, _p, _int, _int, _p, _int, _i64, _int, _p, _p, _p, _p, _i64, C.POINTER(_i64), _p],
    ),
}


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MotifsError(
                ERR_NO_DEVICE,
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)",
            )
        # torch bundles its own libamdhip64.so.7; if it is going to live in this process it
        # must be loaded BEFORE us so that both bind to ONE HIP runtime (the loader
        # de-duplicates by SONAME only in that order).  torch is plumbing here: device
        # memory, streams, torch.distributed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        got = handle.motifs_abi_version()
        if got != ABI_VERSION:      # e.g. NULL in motifs_ctx_set_stream changed meaning between ABI 1 and 2
            raise MotifsError(ERR_INVALID, f"{LIB_PATH} has ABI {got}, this binding is written for ABI {ABI_VERSION}: rebuild the library")
        _lib = handle
    return _lib


def check(code):
    if code != OK:
        raise MotifsError(code, lib().motifs_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


class Context:
    """Owns a motifs_ctx (device, stream, workspaces)."""

    def __init__(self, device=0, stream=None):
        self._h = _p()
        check(lib().motifs_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._models = []          # weak references: models must die before their context
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            for ref in self._models:
                m = ref()
                if m is not None:
                    m.close()
            self._models = []
            lib().motifs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        """Enqueue on the caller's hipStream_t; 0 / None = HIP's null stream (torch's default current stream)."""
        check(lib().motifs_ctx_set_stream(self._h, _p(hip_stream_ptr or None)))

    def get_stream(self):
        """The hipStream_t in use as an integer (0 = the null stream), e.g. for torch.cuda.ExternalStream."""
        out = _p()
        check(lib().motifs_ctx_get_stream(self._h, C.byref(out)))
        return out.value or 0

    def use_private_stream(self):
        """Back to a private non-blocking stream of the library's own (the state of a fresh context)."""
        check(lib().motifs_ctx_use_private_stream(self._h))

    # ---- device memory (for hosts without a GPU array package; tests use it next to torch's allocator) ----
    def dev_alloc(self, nbytes):
        out = _p()
        check(lib().motifs_dev_alloc(self._h, int(nbytes), C.byref(out)))
        return out.value

    def dev_free(self, ptr):
        check(lib().motifs_dev_free(self._h, _p(ptr)))

    def dev_upload(self, dst_ptr, arr):
        arr = np.ascontiguousarray(arr)
        check(lib().motifs_dev_upload(self._h, _p(dst_ptr), _np_ptr(arr), arr.nbytes))

    def dev_download(self, src_ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        check(lib().motifs_dev_download(self._h, _np_ptr(out), _p(src_ptr), out.nbytes))
        return out

    def dev_memset(self, ptr, byte_value, nbytes):
        check(lib().motifs_dev_memset(self._h, _p(ptr), int(byte_value), int(nbytes)))

    def set_workspace_limit(self, nbytes):
        """Bound of the scan's candidate/staging workspace (0 = default 8 GiB); larger scans run in super-batches."""
        check(lib().motifs_ctx_set_workspace_limit(self._h, int(nbytes)))

    def set_records_in_stream_order(self, on=True):
        """pwm_scan_hits_both_dev returns once the totals are known; the records are complete in stream order (see the header)."""
        check(lib().motifs_ctx_set_records_in_stream_order(self._h, int(bool(on))))

    def synchronize(self):
        check(lib().motifs_ctx_synchronize(self._h))

    def enable_timing(self, on=True, slots=None):
        """on: all kernel slots; slots: only these KS_* slots (each timed section costs a few us of stream time)."""
        v = int(bool(on))
        if slots is not None:
            v = 0
            for s in slots:
                v |= 1 << (int(s) + 1)
        check(lib().motifs_ctx_enable_timing(self._h, v))

    def reset_timing(self):
        check(lib().motifs_ctx_reset_timing(self._h))

    def kernel_ms(self, slot):
        ms, n = C.c_double(0), _i64(0)
        check(lib().motifs_ctx_kernel_ms(self._h, int(slot), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def scan_plan(self):
        """Layout of the last hit-record scan: dict(compact, cg_chunks, cg_groups, launches) (motifs_ctx_scan_plan)."""
        v = (C.c_int32 * 4)()
        check(lib().motifs_ctx_scan_plan(self._h, v))
        return {"compact": bool(v[0]), "cg_chunks": int(v[1]), "cg_groups": int(v[2]), "launches": int(v[3])}

    # ---- sequence encoding ----
    @staticmethod
    def codes_bytes(N, L):
        return lib().motifs_codes_bytes(int(N), int(L))

    @staticmethod
    def codes_pitch(L):
        return lib().motifs_codes_pitch(int(L))

    def encode_dev(self, data_ptr, kind, N, L, codes_ptr, bad_flag_ptr=None):
        check(lib().motifs_encode_dev(self._h, _p(data_ptr), int(kind), int(N), int(L), _p(codes_ptr), _p(bad_flag_ptr)))

    # ---- scan ----
    def pwm_scan_dense_dev(self, pwms, lens, codes_ptr, N, L, scores_ptr, ld_l):
        pwms, lens, K, maxlen = _bank(pwms, lens)
        check(
            lib().motifs_pwm_scan_dense_dev(
                self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _p(codes_ptr), int(N), int(L), _p(scores_ptr), int(ld_l)
            )
        )

    def pwm_scan_hits_dev(self, pwms, lens, codes_ptr, N, L, rc, hits_ptr, scores_ptr, cap, n0=0, batch=SCAN_BATCH,
                          counts_ptr=None, allow_small=False):
        """Returns the total number of hits; raises on a too-small buffer unless allow_small."""
        pwms, lens, K, maxlen = _bank(pwms, lens)
        n_out = _i64(0)
        code = lib().motifs_pwm_scan_hits_dev(
            self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _p(codes_ptr), int(N), int(L), int(bool(rc)), int(n0),
            int(batch), _p(hits_ptr), _p(scores_ptr), int(cap), C.byref(n_out), _p(counts_ptr),
        )
        if code == ERR_BUFFER_TOO_SMALL and allow_small:
            return n_out.value
        check(code)
        return n_out.value

    def pwm_scan_hits_both_dev(self, pwms, lens, codes_ptr, N, L, hits_ptrs, scores_ptrs, cap, n0=0, batch=SCAN_BATCH, counts_ptr=None):
        """gpu_scan: both strands in one call (one host wait).  hits_ptrs / scores_ptrs: (forward, reverse) device pointers
        (None, None with cap == 0: count only); counts_ptr: optional 2*K int64 on device.  Returns (n_forward, n_reverse)."""
        pwms, lens, K, maxlen = _bank(pwms, lens)
        n_out = (_i64 * 2)(0, 0)
        hp = hits_ptrs if hits_ptrs is not None else (None, None)
        sp = scores_ptrs if scores_ptrs is not None else (None, None)
        check(lib().motifs_pwm_scan_hits_both_dev(
            self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _p(codes_ptr), int(N), int(L), int(n0), int(batch),
            _p(hp[0]), _p(sp[0]), _p(hp[1]), _p(sp[1]), int(cap), n_out, _p(counts_ptr)))
        return int(n_out[0]), int(n_out[1])

    # ---- consumers of the hit records ----
    def hits_minmax_dev(self, hits_ptr, scores_ptr, n, K, min_ptr, max_ptr):
        check(lib().motifs_hits_minmax_dev(self._h, _p(hits_ptr), _p(scores_ptr), int(n), int(K), _p(min_ptr), _p(max_ptr)))

    def hits_threshold_counts_dev(self, hits_ptr, scores_ptr, n, K, thr_ptr, T, counts_ptr):
        check(lib().motifs_hits_threshold_counts_dev(self._h, _p(hits_ptr), _p(scores_ptr), int(n), int(K), _p(thr_ptr), int(T),
                                                     _p(counts_ptr)))

    def hits_filter_dev(self, hits_ptr, scores_ptr, n, K, thresh_ptr, out_hits_ptr, out_scores_ptr):
        n_out = _i64(0)
        check(lib().motifs_hits_filter_dev(self._h, _p(hits_ptr), _p(scores_ptr), int(n), int(K), _p(thresh_ptr), _p(out_hits_ptr),
                                           _p(out_scores_ptr), C.byref(n_out)))
        return n_out.value

    def hits_count_matrices_dev(self, hits_ptr, n, codes_ptr, L, n0, lens, K, maxlen, comp, counts_ptr):
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        check(lib().motifs_hits_count_matrices_dev(self._h, _p(hits_ptr), int(n), _p(codes_ptr), int(L), int(n0), _np_ptr(lens), int(K),
                                                   int(maxlen), int(bool(comp)), _p(counts_ptr)))

    # ---- consumers of the code records ----
    def codes_mag_histogram_dev(self, recs_ptr, n, hist_ptr):
        check(lib().motifs_codes_mag_histogram_dev(self._h, _p(recs_ptr), int(n), _p(hist_ptr)))

    def codes_filter_dev(self, recs_ptr, n, thresh, out_ptr):
        n_out = _i64(0)
        check(lib().motifs_codes_filter_dev(self._h, _p(recs_ptr), int(n), float(thresh), _p(out_ptr), C.byref(n_out)))
        return n_out.value

    def triplets_offsets_dev(self, range_len_ptr, nranges, offsets_ptr):
        total = _i64(0)
        check(lib().motifs_triplets_offsets_dev(self._h, _p(range_len_ptr), int(nranges), _p(offsets_ptr), C.byref(total)))
        return total.value

    def triplets_enumerate_dev(self, recs_ptr, range_start_ptr, range_len_ptr, nranges, h, offsets_ptr, keys_ptr, vals_ptr, cap):
        check(lib().motifs_triplets_enumerate_dev(self._h, _p(recs_ptr), _p(range_start_ptr), _p(range_len_ptr), int(nranges), int(h),
                                                  _p(offsets_ptr), _p(keys_ptr), _p(vals_ptr), int(cap)))

    def triplets_group_dev(self, keys_ptr, n, uniq_ptr, first_ptr, counts_ptr, group_off_ptr, perm_ptr):
        nu = _i64(0)
        check(lib().motifs_triplets_group_dev(self._h, _p(keys_ptr), int(n), _p(uniq_ptr), _p(first_ptr), _p(counts_ptr), _p(group_off_ptr),
                                              _p(perm_ptr), C.byref(nu)))
        return nu.value

    def pwm_scan(self, pwms, lens, data, kind, N, L, rc, cap=None, want_counts=False):
        """Host-buffer scan (the entry Julia's ccall binds).  cap=None sizes the
        buffers with a count-only first call."""
        pwms, lens, K, maxlen = _bank(pwms, lens)
        data = np.ascontiguousarray(data)
        n_out = _i64(0)
        counts = np.zeros(K, dtype=np.int64) if want_counts else None
        if cap is None:
            code = lib().motifs_pwm_scan(self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind),
                                         int(N), int(L), int(bool(rc)), None, None, 0, C.byref(n_out), None)
            check(code)
            cap = n_out.value
        hits = np.zeros(max(cap, 1), dtype=HIT_DTYPE)
        scores = np.zeros(max(cap, 1), dtype=np.uint16)
        code = lib().motifs_pwm_scan(self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind), int(N),
                                     int(L), int(bool(rc)), _np_ptr(hits) if cap > 0 else None,
                                     _np_ptr(scores) if cap > 0 else None, int(cap), C.byref(n_out), _np_ptr(counts))
        check(code)
        n = n_out.value
        out = (hits[:n], scores[:n].view(np.float16))
        return out + (counts,) if want_counts else out


def _pwm_scan_both(self, pwms, lens, data, kind, N, L, cap=None):
    """gpu_scan on host buffers (one upload, both strands).  Returns ((found_fwd, score_fwd), (found_rc, score_rc))."""
    pwms, lens, K, maxlen = _bank(pwms, lens)
    data = np.ascontiguousarray(data)
    n_out = (_i64 * 2)(0, 0)
    if cap is None:
        check(lib().motifs_pwm_scan_both(self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind), int(N), int(L),
                                         None, None, None, None, 0, n_out, None))
        cap = max(n_out[0], n_out[1])
    hits = [np.zeros(max(cap, 1), dtype=HIT_DTYPE) for _ in range(2)]
    scores = [np.zeros(max(cap, 1), dtype=np.uint16) for _ in range(2)]
    check(lib().motifs_pwm_scan_both(self._h, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind), int(N), int(L),
                                     _np_ptr(hits[0]) if cap else None, _np_ptr(scores[0]) if cap else None,
                                     _np_ptr(hits[1]) if cap else None, _np_ptr(scores[1]) if cap else None, int(cap), n_out, None))
    return tuple((hits[s][: n_out[s]], scores[s][: n_out[s]].view(np.float16)) for s in range(2))


Context.pwm_scan_both = _pwm_scan_both


def fasta_read(path, max_entries=100000):
    """read_fasta (loadfasta/helpers.jl:102-108) + base coding: (N, L) uint8 codes.  Host only, no GPU needed."""
    n, L = _i64(0), C.c_int32(0)
    check(lib().motifs_fasta_read(os.fsencode(path), int(max_entries), None, 0, C.byref(n), C.byref(L)))
    out = np.zeros((n.value, L.value), dtype=np.uint8)
    check(lib().motifs_fasta_read(os.fsencode(path), int(max_entries), _np_ptr(out), out.size, C.byref(n), C.byref(L)))
    return out


def _bank(pwms, lens):
    """pwms: uint16/float16 array with Julia layout (K,4,maxlen) column-major,
    i.e. numpy shape (maxlen, 4, K) C-order."""
    pwms = np.ascontiguousarray(pwms)
    if pwms.dtype == np.float16:
        pwms = pwms.view(np.uint16)
    assert pwms.dtype == np.uint16 and pwms.ndim == 3 and pwms.shape[1] == 4, pwms.shape
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    maxlen, _, K = pwms.shape
    assert lens.shape == (K,)
    return pwms, lens, K, maxlen


class Model:
    """Owns a motifs_model: the `ucdl` state (model.jl:67-137), AdaBelief moments and the engine arena."""

    def __init__(self, ctx, hp, L, arena_bytes=0):
        self.ctx, self.hp, self.L = ctx, hp, int(L)
        self._h = _p()
        check(lib().motifs_model_create(ctx._h, C.byref(hp), int(L), int(arena_bytes), C.byref(self._h)))
        v = [_i64(0) for _ in range(5)]
        check(lib().motifs_model_sizes(self._h, *[C.byref(x) for x in v]))
        self.nD, self.nF, self.nV, self.c, self.l = [x.value for x in v]
        self.nP = self.nD + self.nF + self.nV
        import weakref

        ctx._models.append(weakref.ref(self))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().motifs_model_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, D=None, F=None, warmup3=None, vecs=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in (D, F, warmup3, vecs)]
        for a, n in zip(arrs, (self.nD, self.nF, 3, self.nV)):
            assert a is None or a.size == n, (a.size, n)
        check(lib().motifs_model_set_params(self._h, *[_np_ptr(a) for a in arrs]))

    def get_params(self):
        D, F = np.zeros(self.nD, np.float32), np.zeros(self.nF, np.float32)
        w, v = np.zeros(3, np.float32), np.zeros(self.nV, np.float32)
        check(lib().motifs_model_get_params(self._h, _np_ptr(D), _np_ptr(F), _np_ptr(w), _np_ptr(v)))
        return D, F, w, v

    def init_random(self, seed):
        check(lib().motifs_model_init_random(self._h, int(seed)))

    def loss_grad_dev(self, codes_ptr, n_groups, loss_ptr, grad_ptr, keep=False):
        check(lib().motifs_model_loss_grad_dev(self._h, _p(codes_ptr), int(n_groups), _p(loss_ptr), _p(grad_ptr), int(keep)))

    def adabelief_dev(self, grad_ptr, gscale):
        check(lib().motifs_model_adabelief_dev(self._h, _p(grad_ptr), float(gscale)))

    def l1_syntax(self):
        out = C.c_float(0)
        check(lib().motifs_model_l1_syntax(self._h, C.byref(out)))
        return out.value

    def train_step(self, codes, n_groups, want_l1=True):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.shape == (n_groups * self.hp.batch_size, self.L), codes.shape
        loss = np.zeros(n_groups, np.float32)
        l1 = C.c_float(0)
        check(lib().motifs_model_train_step(self._h, _np_ptr(codes), int(n_groups), _np_ptr(loss),
                                            C.byref(l1) if want_l1 else None))
        return loss, l1.value

    def train_step_onehot(self, S, n_groups, want_l1=True):
        """train.jl:41-52 on the reference's batch format: S = bytes of the (4L, 1, n_groups*batch_size) Float32 array."""
        S = np.ascontiguousarray(S, dtype=np.float32)
        assert S.size == n_groups * self.hp.batch_size * 4 * self.L, S.shape
        loss = np.zeros(n_groups, np.float32)
        l1 = C.c_float(0)
        check(lib().motifs_model_train_step_onehot(self._h, _np_ptr(S), int(n_groups), _np_ptr(loss),
                                                   C.byref(l1) if want_l1 else None))
        return loss, l1.value

    def dp_train_step_dev(self, comm, codes_ptr, n_groups_local, n_groups_total, loss_ptr, grad_ptr):
        """One data-parallel optimiser step (comm: a Comm or None for a single device)."""
        check(lib().motifs_model_dp_train_step_dev(self._h, comm._h if comm is not None else None, _p(codes_ptr), int(n_groups_local),
                                                   int(n_groups_total), _p(loss_ptr), _p(grad_ptr)))

    def dp_grad_dev(self, codes_ptr, n_groups_local, loss_ptr, grad_ptr):
        """Phase 1 of a data-parallel step: this device's summed gradient (zeros for no mini-batch)."""
        check(lib().motifs_model_dp_grad_dev(self._h, _p(codes_ptr), int(n_groups_local), _p(loss_ptr), _p(grad_ptr)))

    def dp_update_dev(self, grad_ptr, n_groups_total):
        """Phase 3: AdaBelief on gradient / n_groups_total."""
        check(lib().motifs_model_dp_update_dev(self._h, _p(grad_ptr), int(n_groups_total)))

    def allreduce_grad(self, comm, grad_ptr):
        check(lib().motifs_model_allreduce_grad(self._h, comm._h, _p(grad_ptr)))

    def retrieve_codes(self, data, kind, N, cap=None):
        data = np.ascontiguousarray(data)
        n_out = _i64(0)
        if cap is None:
            cap = int(N) * max(4 * self.hp.q, 64)
        out = np.zeros(max(cap, 1), dtype=CODE_DTYPE)
        check(lib().motifs_model_retrieve_codes(self._h, _np_ptr(data), int(kind), int(N), _np_ptr(out), int(cap), C.byref(n_out)))
        return out[: n_out.value]

    def time_filter_scan(self, codes_ptr, n_groups, reps=5):
        """Average device ms of a4 (warmup_ZY's filter-bank scan) alone, a measurement hook for bench.py."""
        ms = C.c_float(0)
        check(lib().motifs_model_time_filter_scan(self._h, _p(codes_ptr), int(n_groups), int(reps), C.byref(ms)))
        return ms.value

    def time_syntax_conv(self, codes_ptr, n_groups, reps=5):
        """Average device ms of a7's dense contraction (the syntax-layer analysis GEMM) alone, a measurement hook for bench.py."""
        ms = C.c_float(0)
        check(lib().motifs_model_time_syntax_conv(self._h, _p(codes_ptr), int(n_groups), int(reps), C.byref(ms)))
        return ms.value

    def arena_peak(self):
        """Bytes of the engine arena the steps so far have used at most."""
        v = C.c_size_t(0)
        check(lib().motifs_model_arena_peak(self._h, C.byref(v)))
        return int(v.value)

    def dump(self, name):
        n = _i64(0)
        check(lib().motifs_model_dump(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        check(lib().motifs_model_dump(self._h, name.encode(), _np_ptr(out), n.value, C.byref(n)))
        return out


class Comm:
    """Owns a motifs_comm: one rank of an RCCL communicator on a context's device (collectives run on its stream)."""

    def __init__(self, ctx, uid, nranks, rank, _handle=None):
        self.ctx = ctx
        self._h = _p()
        if _handle is not None:          # a rank of Comm.create_all
            self._h = _p(_handle)
        else:
            uid = np.frombuffer(bytes(uid), dtype=np.uint8)
            assert uid.size == COMM_ID_BYTES
            check(lib().motifs_comm_create(ctx._h, _np_ptr(uid), int(nranks), int(rank), C.byref(self._h)))
        self.rank, self.nranks = int(rank), int(nranks)

    @staticmethod
    def create_all(ctxs):
        """ncclCommInitAll: every rank of a single-process communicator over the devices of `ctxs` (one Comm per context)."""
        n = len(ctxs)
        hs = (_p * n)(*[c._h for c in ctxs])
        out = (_p * n)()
        check(lib().motifs_comm_create_all(hs, n, out))
        return [Comm(ctxs[i], None, n, i, _handle=out[i]) for i in range(n)]

    @staticmethod
    def group_start():
        check(lib().motifs_comm_group_start())

    @staticmethod
    def group_end():
        check(lib().motifs_comm_group_end())

    def allreduce_sum_f32_to(self, send_ptr, recv_ptr, n):
        check(lib().motifs_comm_allreduce_sum_f32_to_dev(self._h, _p(send_ptr), _p(recv_ptr), int(n)))

    @staticmethod
    def unique_id():
        uid = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        check(lib().motifs_comm_unique_id(_np_ptr(uid)))
        return uid.tobytes()

    def allreduce_sum_f32(self, ptr, n):
        check(lib().motifs_comm_allreduce_sum_f32_dev(self._h, _p(ptr), int(n)))

    def allreduce_sum_i64(self, ptr, n):
        check(lib().motifs_comm_allreduce_sum_i64_dev(self._h, _p(ptr), int(n)))

    def allreduce_sum_u32(self, ptr, n):
        check(lib().motifs_comm_allreduce_sum_u32_dev(self._h, _p(ptr), int(n)))

    def hist_allreduce(self, counts_ptr, K, n_strands=1):
        check(lib().motifs_hist_allreduce(self._h, _p(counts_ptr), int(K), int(n_strands)))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().motifs_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _parr(ptrs, n):
    """n pointers (ints / None / objects with ._h) as a C array of void*."""
    vals = []
    for x in ptrs:
        x = getattr(x, "_h", x)
        vals.append(x.value if isinstance(x, _p) else (x or None))
    assert len(vals) == n
    return (_p * n)(*vals)


def dp_train_step_all(models, comms, codes_ptrs, n_groups_local, n_groups_total, loss_ptrs, grad_ptrs, reduced_ptrs=None):
    """One data-parallel optimiser step of len(models) replicas driven by this host thread: every device's gradient,
    then the grouped all-reduces, then every device's AdaBelief (motifs_model_dp_train_step_all)."""
    n = len(models)
    check(lib().motifs_model_dp_train_step_all(
        _parr(models, n), _parr(comms, n) if comms is not None else None, n, _parr(codes_ptrs, n),
        (_int * n)(*[int(g) for g in n_groups_local]), int(n_groups_total), _parr(loss_ptrs, n), _parr(grad_ptrs, n),
        _parr(reduced_ptrs, n) if reduced_ptrs is not None else None))


def dp_train_step_host(models, comms, data, kind, n_groups, want_l1=True):
    """The same step on a host matrix of `kind` holding n_groups * batch_size reads; returns (losses, l1F)."""
    n = len(models)
    data = np.ascontiguousarray(data)
    loss = np.zeros(n_groups, np.float32)
    l1 = C.c_float(0)
    check(lib().motifs_model_dp_train_step_host(_parr(models, n), _parr(comms, n) if comms is not None else None, n, _np_ptr(data),
                                                int(kind), int(n_groups), _np_ptr(loss), C.byref(l1) if want_l1 else None))
    return loss, l1.value


def pwm_scan_both_sharded(ctxs, comms, pwms, lens, data, kind, N, L, shard_align=SCAN_BATCH, cap=None):
    """gpu_scan of a host matrix over the devices of `ctxs` (one process, one host thread per device).  Returns
    ((found_fwd, score_fwd), (found_rc, score_rc), counts[2, K], shard_counts[n_dev, 2])."""
    n = len(ctxs)
    pwms, lens, K, maxlen = _bank(pwms, lens)
    data = np.ascontiguousarray(data)
    n_out = (_i64 * 2)(0, 0)
    counts = np.zeros((2, K), dtype=np.int64)
    shard = np.zeros((n, 2), dtype=np.int64)
    cs, cm = _parr(ctxs, n), (_parr(comms, n) if comms is not None else None)
    if cap is None:
        check(lib().motifs_pwm_scan_both_sharded(cs, cm, n, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind), int(N), int(L),
                                                 int(shard_align), None, None, None, None, 0, n_out, None, None))
        cap = max(n_out[0], n_out[1])
    hits = [np.zeros(max(cap, 1), dtype=HIT_DTYPE) for _ in range(2)]
    scores = [np.zeros(max(cap, 1), dtype=np.uint16) for _ in range(2)]
    check(lib().motifs_pwm_scan_both_sharded(cs, cm, n, _np_ptr(pwms), _np_ptr(lens), K, maxlen, _np_ptr(data), int(kind), int(N), int(L),
                                             int(shard_align), _np_ptr(hits[0]) if cap else None, _np_ptr(scores[0]) if cap else None,
                                             _np_ptr(hits[1]) if cap else None, _np_ptr(scores[1]) if cap else None, int(cap), n_out,
                                             _np_ptr(counts), _np_ptr(shard)))
    fwd, rcs = [(hits[s][: n_out[s]], scores[s][: n_out[s]].view(np.float16)) for s in range(2)]
    return fwd, rcs, counts, shard
