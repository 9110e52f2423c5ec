"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/motifs_hip.h declares, and refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "motifs_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(motifs_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built(pkg):
    if not os.path.exists(pkg._lib.LIB_PATH):
        from _pkg import load_build

        load_build().build(verbose=False)
    return pkg


def test_header_declares_something():
    names = declared_functions()
    assert "motifs_pwm_scan" in names and "motifs_ctx_create" in names and len(names) >= 10


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(built._lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(handle, n)]
    assert not missing, f"declared in motifs_hip.h but not exported: {missing}"


def test_binding_covers_every_declared_symbol(built):
    assert sorted(built._lib.SIGNATURES) == declared_functions()


def test_abi_version(built):
    assert built._lib.lib().motifs_abi_version() == built._lib.ABI_VERSION == 3


def test_comm_entry_points_fail_cleanly_without_a_device(built):
    """The collectives are part of the ABI; without a context there is nothing to run them on."""
    lib = built._lib.lib()
    assert lib.motifs_comm_allreduce_sum_f32_dev(None, None, 4) == built._lib.ERR_INVALID
    assert lib.motifs_hist_allreduce(None, None, 4, 3) == built._lib.ERR_INVALID
    assert lib.motifs_comm_rank(None, None, None) == built._lib.ERR_INVALID


def test_no_cpu_fallback(built):
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built._lib.MotifsError) as e:
        built._lib.Context(0)
    assert e.value.code == built._lib.ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg_dir = os.path.join(ROOT, "motifs.jl_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_every_switch_of_the_library_is_named_in_a_test():
    """The MOTIFS_* environment switches csrc/ reads (debugging / A-B forms, read once per process) stay few and none of them guards a form that no
    test runs: at most 20, each named in a tests/*.py file (round-4 verdict: 35 switches, a third of them covered by nothing)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "motifs.jl_amd", "csrc", "*.h*")):
        names |= set(re.findall(r'getenv\("(MOTIFS_[A-Z0-9_]+)"\)', open(f).read()))
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "*.py")) if not f.endswith("test_abi.py"))
    assert 0 < len(names) <= 20, sorted(names)
    missing = sorted(n for n in names if n not in tests)
    assert not missing, missing
