"""Host mirror of the reference's scan driver, src/inference/_h3_1_alignment.jl:38-112.

Same names, argument meaning and output format as the Julia functions; the
numeric work is one call into libmotifs_hip (`motifs_pwm_scan`, the entry the
Julia `ccall` shim binds).  numpy arrays hold the same bytes as the Julia
arrays, so shapes read reversed: `data.data_matrix` (4L,1,N) is numpy (N,1,4L).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .synth import pad_bank

float_type_retrieval = np.float16  # _0_const.jl:1
batch_size_greedy = _lib.SCAN_BATCH  # _h3_1_alignment.jl:12


@dataclass
class Motifs:
    """The fields of `motifs{T,S}` (_s1_make_motifs.jl:1-18) the scan reads and writes."""
    pwms: List[np.ndarray]                      # each (4, len) Float16 log-odds
    lens: np.ndarray                            # (num_motifs,) int64
    num_motifs: int = 0
    positions: Optional[List[Dict[int, list]]] = None
    scores: Optional[List[Dict[int, list]]] = None
    use_comp: Optional[List[Dict[int, list]]] = None
    positions_bg: Optional[List[Dict[int, list]]] = None
    scores_bg: Optional[List[Dict[int, list]]] = None
    use_comp_bg: Optional[List[Dict[int, list]]] = None

    def __post_init__(self):
        self.lens = np.asarray(self.lens, dtype=np.int64)
        self.num_motifs = len(self.pwms)


@dataclass
class FastaData:
    """The fields of FASTA_DNA (loadfasta/fasta.jl:6-57) the scan touches."""
    data_matrix: np.ndarray                     # one-hot Float32, bytes of (4L,1,N)
    data_matrix_bg: Optional[np.ndarray] = None
    data_matrix_test: Optional[np.ndarray] = None
    data_matrix_bg_test: Optional[np.ndarray] = None
    N: int = field(init=False, default=0)
    L: int = field(init=False, default=0)

    def __post_init__(self):
        m = self.data_matrix.reshape(self.data_matrix.shape[0], -1)
        self.N, self.L = m.shape[0], m.shape[1] // 4


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(0)
    return _default_ctx


def data_(data, test=False):       # :54
    return data.data_matrix_test if test else data.data_matrix


def data_bg(data, test=False):     # :55
    return data.data_matrix_bg_test if test else data.data_matrix_bg


def get_pos_scores_arr(ms, data, rc=False, bg=False, test=False, ctx=None):
    """:57-87.  Returns (found_record, score_record): a structured uint32 array
    with fields m, n, l (record_t, 1-based) and the Float16 scores, in the
    reference's order (5000-sequence batches, then column-major findall)."""
    ctx = ctx or default_context()
    data_matrix = data_bg(data, test=test) if bg else data_(data, test=test)
    data_matrix = np.ascontiguousarray(data_matrix, dtype=np.float32)
    N = data_matrix.shape[0]
    data_matrix = data_matrix.reshape(N, -1)
    L4 = data_matrix.shape[1]
    pwms = pad_bank(ms.pwms, ms.lens)           # :65-67 (the rc reverse of :68 happens in the library)
    return ctx.pwm_scan(pwms, ms.lens, data_matrix, _lib.DATA_ONEHOT_F32, N, L4 // 4, rc)


def motifs_prep(ms):               # _s1_make_motifs.jl:185-190
    return ([dict() for _ in range(ms.num_motifs)], [dict() for _ in range(ms.num_motifs)],
            [dict() for _ in range(ms.num_motifs)])


def modify_w_found(found_record, score_record, positions, scores, use_comp, rc=False):
    """:38-52 (`modify_w_found!`): push every record, in record order."""
    if len(found_record) == 0:
        return
    m, n, l = found_record["m"], found_record["n"], found_record["l"]
    order = np.lexsort((n, m))                  # stable: keeps record order inside each (m, n)
    m, n, l, s = m[order], n[order], l[order], np.asarray(score_record)[order]
    cut = np.nonzero((np.diff(m) != 0) | (np.diff(n) != 0))[0] + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [len(m)]))
    for a, b in zip(starts, ends):
        mi, ni = int(m[a]) - 1, int(n[a])
        if ni in positions[mi]:
            positions[mi][ni].extend(l[a:b].tolist())
            scores[mi][ni].extend(s[a:b].tolist())
            use_comp[mi][ni].extend([bool(rc)] * (b - a))
        else:
            positions[mi][ni] = l[a:b].tolist()
            scores[mi][ni] = s[a:b].tolist()
            use_comp[mi][ni] = [bool(rc)] * (b - a)


def gpu_scan(ms, data, bg=False, test=False, ctx=None):
    """:89-99.  Both strands through one library call (`motifs_pwm_scan_both`): the data matrix crosses PCIe once."""
    ctx = ctx or default_context()
    data_matrix = data_bg(data, test=test) if bg else data_(data, test=test)
    data_matrix = np.ascontiguousarray(data_matrix, dtype=np.float32)
    N = data_matrix.shape[0]
    data_matrix = data_matrix.reshape(N, -1)
    (found, score), (found_rc, score_rc) = ctx.pwm_scan_both(pad_bank(ms.pwms, ms.lens), ms.lens, data_matrix, _lib.DATA_ONEHOT_F32, N,
                                                             data_matrix.shape[1] // 4)
    positions, scores, use_comp = motifs_prep(ms)
    modify_w_found(found, score, positions, scores, use_comp, rc=False)
    modify_w_found(found_rc, score_rc, positions, scores, use_comp, rc=True)
    return positions, scores, use_comp


def scan_w_gpu(ms, data, bg=False, ctx=None):
    """:101-112 (`scan_w_gpu!`)."""
    positions, scores, use_comp = gpu_scan(ms, data, bg=bg, ctx=ctx)
    if bg:
        ms.positions_bg, ms.scores_bg, ms.use_comp_bg = positions, scores, use_comp
    else:
        ms.positions, ms.scores, ms.use_comp = positions, scores, use_comp
