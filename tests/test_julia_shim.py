"""julia/MotifsHIP.jl cannot be executed in the build image (no Julia), so it is checked statically: every `ccall`
in it must name a function include/motifs_hip.h declares, with the declared arity, and with argument / return types
of the right class (integers by width, floats by width, pointers as Ptr/Ref/Cstring).  The shim must also bind
every entry point a `discover_motifs` port needs."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "motifs_hip.h")
SHIM = os.path.join(ROOT, "julia", "MotifsHIP.jl")

C_CLASS = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "float": "f32",
           "double": "f64", "void": "void"}
JL_CLASS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "usize", "Cfloat": "f32",
            "Float32": "f32", "Cdouble": "f64", "Float64": "f64", "Cvoid": "void", "Cstring": "ptr"}


def c_class(t):
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    base = t.split()[0] if t else "void"
    return C_CLASS[base]


def header_decls():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(motifs_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        if args in ("", "void"):
            classes = []
        else:
            classes = []
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier), keep pointer stars and array brackets
                mm = re.match(r"(.*?)(\b[A-Za-z_]\w*)(\s*\[[^\]]*\])?$", a)
                typ = (mm.group(1) + (mm.group(3) or "")) if mm and mm.group(1).strip() else a
                classes.append(c_class(typ))
        decls[name] = (c_class(ret), classes)
    return decls


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")):
        return "ptr"
    return JL_CLASS[t]


def shim_ccalls():
    text = open(SHIM).read()
    text = re.sub(r"#.*$", "", text, flags=re.M)
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*lib\),\s*(\w+),\s*\(", text):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = split_top(text[m.end():i - 1])
        # the call's own arguments follow the type tuple
        j, depth = i, 1
        while depth:
            depth += {"(": 1, ")": -1, "[": 1, "]": -1}.get(text[j], 0)
            j += 1
        nargs = len(split_top(text[i:j - 1].lstrip(", \n")))
        calls.append((name, ret, types, nargs))
    return calls


def test_header_parser_sees_every_function():
    decls = header_decls()
    assert decls["motifs_pwm_scan"] == ("i32", ["ptr", "ptr", "ptr", "i32", "i32", "ptr", "i32", "i64", "i32", "i32", "ptr", "ptr",
                                                "i64", "ptr", "ptr"])
    assert decls["motifs_ctx_destroy"] == ("void", ["ptr"])
    assert decls["motifs_codes_bytes"] == ("usize", ["i64", "i32"])
    assert decls["motifs_last_error"] == ("ptr", [])
    assert decls["motifs_comm_unique_id"] == ("i32", ["ptr"])


def test_every_ccall_matches_the_header():
    decls = header_decls()
    calls = shim_ccalls()
    assert len(calls) >= 25
    for name, ret, types, nargs in calls:
        assert name in decls, f"{name} is not declared in motifs_hip.h"
        cret, cargs = decls[name]
        assert jl_class(ret) == cret, f"{name}: return {ret} vs {cret}"
        assert len(types) == len(cargs), f"{name}: {len(types)} argument types, the header has {len(cargs)}"
        assert nargs == len(cargs), f"{name}: {nargs} call arguments, the header has {len(cargs)}"
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            assert jl_class(jt) == ct, f"{name}: argument {k + 1} is {jt}, the header says {ct}"


def test_shim_binds_what_discover_motifs_needs():
    bound = {c[0] for c in shim_ccalls()}
    need = {"motifs_ctx_create", "motifs_ctx_destroy", "motifs_model_create", "motifs_model_destroy", "motifs_model_init_random",
            "motifs_model_get_params", "motifs_model_set_params", "motifs_model_train_step_onehot", "motifs_model_l1_syntax",
            "motifs_model_retrieve_codes", "motifs_pwm_scan", "motifs_comm_create_all", "motifs_comm_create",
            "motifs_model_dp_train_step_dev", "motifs_hist_allreduce", "motifs_model_allreduce_grad",
            # round 3: what a Julia host needs to REACH the device-resident and multi-device paths without CUDA.jl / AMDGPU.jl
            "motifs_abi_version", "motifs_dev_alloc", "motifs_dev_free", "motifs_dev_upload", "motifs_dev_download", "motifs_dev_memset",
            "motifs_encode_dev", "motifs_codes_bytes", "motifs_pwm_scan_hits_both_dev", "motifs_pwm_scan_dense_dev", "motifs_pwm_scan_both",
            "motifs_fasta_read", "motifs_hits_minmax_dev", "motifs_hits_threshold_counts_dev", "motifs_hits_filter_dev",
            "motifs_hits_count_matrices_dev", "motifs_codes_mag_histogram_dev", "motifs_codes_filter_dev", "motifs_triplets_offsets_dev",
            "motifs_triplets_enumerate_dev", "motifs_triplets_group_dev", "motifs_model_dp_train_step_all", "motifs_model_dp_train_step_host",
            "motifs_model_dp_grad_dev", "motifs_model_dp_update_dev", "motifs_comm_allreduce_sum_f32_to_dev", "motifs_pwm_scan_both_sharded",
            "motifs_ctx_use_private_stream", "motifs_comm_group_start", "motifs_comm_group_end"}
    assert need <= bound, sorted(need - bound)
    text = open(SHIM).read()
    for fn in ("function train_ucdl(data;", "function code_retrieval(data, cdl", "function get_pos_scores_arr(ms, data;",
               "function gpu_scan(ms, data;", "function scan_w_gpu!(ms, data;", "function modify_w_found!(",
               "function gpu_scan(ms, reads::DeviceReads", "function gpu_scan_sharded(ms, data, ctxs", "function dp_train_step!(cdls::Vector{ucdl}",
               "__init__() = abi_version() == ABI_VERSION"):
        assert fn in text, fn


def test_every_dev_pointer_in_the_integration_example_has_a_source():
    """INTEGRATION.md's multi-GPU example used codes_dev[d] / loss_dev[d] / grad_dev[d] that came from nowhere and
    wrapped the per-rank step in a group.  The example must only use names the shim defines, and must not group the step."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(SHIM).read()
    blocks = re.findall(r"```julia\n(.*?)```", text, flags=re.S)
    assert blocks
    for b in blocks:
        for name in re.findall(r"(?<![A-Za-z])HIP\.([A-Za-z_!]+)", b):
            assert re.search(r"(function |^|\n)(%s)(?![A-Za-z_!])" % re.escape(name), shim) or ("struct " + name) in shim or ("const " + name) in shim, \
                f"INTEGRATION.md uses HIP.{name}, which julia/MotifsHIP.jl does not define"
        if "group_start" in b:
            inside = b.split("group_start", 1)[1].split("group_end", 1)[0]
            assert "dp_train_step" not in inside, "a whole optimiser step inside an RCCL group runs AdaBelief before the sum"


def test_struct_layouts_match():
    """HParams mirrors motifs_hparams field for field (8 x int32, 2 x float)."""
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    m = re.search(r"typedef struct motifs_hparams \{(.*?)\}", h, flags=re.S)
    cfields = [f.strip() for f in re.sub(r"\b(int32_t|float)\b", "", m.group(1)).replace(";", ",").split(",") if f.strip()]
    j = re.search(r"struct HParams.*?\n(.*?)\nend", open(SHIM).read(), flags=re.S).group(1)
    jfields = re.findall(r"(\w+)::(Int32|Float32)", j)
    assert [f for f, _ in jfields] == cfields
    assert [t for _, t in jfields] == ["Int32"] * 8 + ["Float32"] * 2
