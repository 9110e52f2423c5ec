"""motifs.jl_amd — MI355X-native hot path of MOTIFs.jl behind a C ABI.

The directory name carries a dot, so it is not importable by name; load it with
`load_pkg()` from the repo-root helper `_pkg.py` (registers it as
`motifs_jl_amd`).  Submodules:

  _build   compile csrc/*.hip for gfx950 into libmotifs_hip.so (in-tree)
  _lib     ctypes binding of include/motifs_hip.h (fails loudly without the .so / a GPU)
  scan     host mirror of src/inference/_h3_1_alignment.jl:38-112
  synth    synthetic inputs of SURVEY.md §8(d)
"""
from . import _lib, model, parallel, post, scan, synth  # noqa: F401

__all__ = ["_lib", "model", "parallel", "post", "scan", "synth"]
