"""Build libmotifs_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmotifs_hip.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

# per-file extra flags
EXTRA = {
    # The scan kernel's 4-way scalar branch per base must survive code generation:
    # LLVM structurizes even uniform control flow by default, which turns the
    # rotating accumulators into a v_mov per add (2x the VALU work).
    "scan_kernels.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions"],
    # keep MFMA accumulators in VGPRs: the candidate kernel reads every accumulator with VALU right after the
    # chain, and the AGPR form costs one v_accvgpr_read per register (16 extra VALU per 32x32 tile)
    "scan_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "scan_dense.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}


def _newer(src_paths, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in src_paths)


def build(verbose=True, force=False, defs=None, out=None):
    """defs / out: an experimental build beside the shipped one (A/B timing through MOTIFS_HIP_LIB): extra -D flags, another
    library path; its objects go to their own directory."""
    global OBJ, LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir, lib_path = OBJ, LIB
    if defs or out:
        assert out, "an experimental build needs its own output path"
        lib_path = os.path.abspath(out)
        obj_dir = os.path.join(CSRC, "_obj_" + os.path.basename(lib_path).replace(".so", ""))
        force = True
    return _build(hipcc, obj_dir, lib_path, list(defs or []), verbose, force)


def _build(hipcc, OBJ, LIB, defs, verbose, force):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]   # (.inc: the parts of cdl_engine.hip)
    headers.append(os.path.join(HERE, "..", "include", "motifs_hip.h"))
    objs, cmds = [], []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmds.append([hipcc] + COMMON + defs + EXTRA.get(f, []) + ["-c", src, "-o", obj])
    if cmds:    # the translation units side by side (the largest, scan_mfma.hip, is half of a sequential build)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[build]", " ".join(cmd), file=sys.stderr, flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, cmds))
    if force or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # python motifs.jl_amd/_build.py [--force] [--out path.so -DNAME=1 ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = args[args.index("--out") + 1] if "--out" in args else None
    build(force="--force" in sys.argv, defs=[a for a in args if a.startswith("-D")], out=out)
