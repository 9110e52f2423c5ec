# MotifsHIP.jl — ccall shim over libmotifs_hip.so (include/motifs_hip.h, ABI 3).
#
# Host code stays in Julia; no CUDA.jl / AMDGPU.jl / Flux / NNlib on the path.  The functions below carry the
# reference's names and argument meaning, so `discover_motifs` (src/wrap.jl:1-11) needs the four substitutions shown
# in INTEGRATION.md and nothing else.  Julia is not installed in the build image: this file is checked there by
# tests/test_julia_shim.py (every ccall against the header: symbol, arity, argument and return types), not executed.
module MotifsHIP

import Random

const lib = get(ENV, "MOTIFS_HIP_LIB", "libmotifs_hip.so")
const float_type = Float32                      # src/MOTIFs.jl:14
const float_type_retrieval = Float16            # src/inference/_0_const.jl:1
const batch_size_greedy = 5000                  # src/inference/_h3_1_alignment.jl:12
const COMM_ID_BYTES = 128

struct HParams           # motifs_hparams == Hyperparam (src/model.jl:1-14)
    filter_len::Int32; M::Int32; h::Int32; K::Int32; q::Int32; batch_size::Int32
    num_pass_xyz::Int32; num_pass_df::Int32; magnifying_factor::Float32; gamma::Float32
end
Hyperparam(; filter_len=8, M=50, h=12, K=24, q=32, batch_size=6, num_pass_xyz=6, num_pass_df=3,
           magnifying_factor=10f0, gamma=0.1f0) =
    HParams(filter_len, M, h, K, q, batch_size, num_pass_xyz, num_pass_df, magnifying_factor, gamma)

# stored_code_component_t (_0_const.jl:3-4): 12-byte isbits records, the layout of motifs_code_rec
const stored_code_component_t = NamedTuple{(:position, :fil, :seq, :mag), Tuple{UInt16, UInt16, UInt32, Float16}}
const record_t = NTuple{3, UInt32}              # _h3_1_alignment.jl:10 == motifs_hit

last_error() = unsafe_string(ccall((:motifs_last_error, lib), Cstring, ()))
check(rc) = rc == 0 || error("libmotifs_hip status $rc: " * last_error())
abi_version() = ccall((:motifs_abi_version, lib), Cint, ())
const ABI_VERSION = 3
# a library of another ABI is refused at load time (between ABI 1 and 2 a NULL stream changed meaning, for one)
__init__() = abi_version() == ABI_VERSION || error("libmotifs_hip has ABI $(abi_version()), MotifsHIP.jl is written for ABI $ABI_VERSION")

# ---- context ---------------------------------------------------------------------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    device::Int
    function Context(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:motifs_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[], device)
        finalizer(close, c)
        c
    end
end
function Base.close(c::Context)
    c.h == C_NULL || ccall((:motifs_ctx_destroy, lib), Cvoid, (Ptr{Cvoid},), c.h)
    c.h = C_NULL
    nothing
end
synchronize(c::Context) = check(ccall((:motifs_ctx_synchronize, lib), Cint, (Ptr{Cvoid},), c.h))
set_stream!(c::Context, stream::Ptr{Cvoid}) = check(ccall((:motifs_ctx_set_stream, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, stream))
function get_stream(c::Context)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:motifs_ctx_get_stream, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), c.h, r))
    r[]
end
use_private_stream!(c::Context) = check(ccall((:motifs_ctx_use_private_stream, lib), Cint, (Ptr{Cvoid},), c.h))
set_workspace_limit!(c::Context, bytes::Integer) =
    check(ccall((:motifs_ctx_set_workspace_limit, lib), Cint, (Ptr{Cvoid}, Csize_t), c.h, bytes))
# gpu_scan over many shards: return once the hit totals are known, the records complete in stream order (see the header)
set_records_in_stream_order!(c::Context, on::Bool=true) =
    check(ccall((:motifs_ctx_set_records_in_stream_order, lib), Cint, (Ptr{Cvoid}, Cint), c.h, on ? 1 : 0))
# diagnostics: how the last hit-record scan was laid out (compact entries, chunks per chunk group, groups, launches)
function scan_plan(c::Context)
    v = zeros(Int32, 4)
    check(ccall((:motifs_ctx_scan_plan, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), c.h, v))
    (compact = v[1] != 0, cg_chunks = Int(v[2]), cg_groups = Int(v[3]), launches = Int(v[4]))
end

const default_context = Ref{Union{Nothing, Context}}(nothing)
context() = (default_context[] === nothing && (default_context[] = Context(0)); default_context[])

# ---- the model: train.jl:29-35 (Hyperparam(), length_info, projectors, ucdl(hp), Flux.params, AdaBelief()) -------------
mutable struct ucdl
    h::Ptr{Cvoid}
    ctx::Context
    hp::HParams
    L::Int
    nD::Int64; nF::Int64; nV::Int64; c::Int64; l::Int64
    function ucdl(hp::HParams, L::Integer; ctx::Context=context(), seed=rand(UInt64), arena_bytes::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:motifs_model_create, lib), Cint, (Ptr{Cvoid}, Ref{HParams}, Cint, Csize_t, Ref{Ptr{Cvoid}}),
                    ctx.h, Ref(hp), L, arena_bytes, r))
        nD = Ref{Int64}(0); nF = Ref{Int64}(0); nV = Ref{Int64}(0); c = Ref{Int64}(0); l = Ref{Int64}(0)
        check(ccall((:motifs_model_sizes, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}),
                    r[], nD, nF, nV, c, l))
        m = new(r[], ctx, hp, L, nD[], nF[], nV[], c[], l[])
        finalizer(close, m)
        check(ccall((:motifs_model_init_random, lib), Cint, (Ptr{Cvoid}, UInt64), m.h, seed))   # ucdl(hp), model.jl:84-100
        m
    end
end
function Base.close(m::ucdl)
    (m.h == C_NULL || m.ctx.h == C_NULL) || ccall((:motifs_model_destroy, lib), Cvoid, (Ptr{Cvoid},), m.h)
    m.h = C_NULL
    nothing
end

# cdl.D (f_len, 1, M) and cdl.F (h, twoM, 1, K) in the reference's layouts (model.jl:84-90); also .warmup (3) and .vecs
function get_params(m::ucdl)
    hp = m.hp
    D = Array{Float32}(undef, 4 * hp.filter_len, 1, hp.M)
    F = Array{Float32}(undef, hp.h, 2 * hp.M, 1, hp.K)
    warm = Vector{Float32}(undef, 3); vecs = Vector{Float32}(undef, m.nV)
    GC.@preserve D F warm vecs check(ccall((:motifs_model_get_params, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), m.h, D, F, warm, vecs))
    (D=D, F=F, warmup=warm, vecs=vecs)
end
function set_params!(m::ucdl; D=nothing, F=nothing, warmup=nothing, vecs=nothing)
    p(x) = x === nothing ? Ptr{Float32}(C_NULL) : pointer(x)
    a = map(x -> x === nothing ? nothing : Array{Float32}(x), (D, F, warmup, vecs))
    GC.@preserve a check(ccall((:motifs_model_set_params, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), m.h, p(a[1]), p(a[2]), p(a[3]), p(a[4])))
end
function Base.getproperty(m::ucdl, s::Symbol)
    s === :D && return get_params(m).D
    s === :F && return get_params(m).F
    getfield(m, s)
end

# bytes of the engine arena the steps so far needed at most: what `arena_bytes` has to cover
function arena_peak(m::ucdl)
    r = Ref{Csize_t}(0)
    check(ccall((:motifs_model_arena_peak, lib), Cint, (Ptr{Cvoid}, Ref{Csize_t}), getfield(m, :h), r))
    Int(r[])
end
function l1_syntax(m::ucdl)                       # sum(abs.(prep_syntax_filters(cdl.F))), train.jl:47
    r = Ref{Float32}(0)
    check(ccall((:motifs_model_l1_syntax, lib), Cint, (Ptr{Cvoid}, Ref{Float32}), getfield(m, :h), r))
    r[]
end

# train.jl:41-52 for one DataLoader batch S = data.data_matrix[:, :, idx]: (4L, 1, n_groups * batch_size) Float32
function train_step!(m::ucdl, S::Array{Float32,3}, n_groups::Integer=1)
    loss = Vector{Float32}(undef, n_groups); l1 = Ref{Float32}(0)
    GC.@preserve S loss check(ccall((:motifs_model_train_step_onehot, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Cint, Ptr{Float32}, Ref{Float32}), getfield(m, :h), S, n_groups, loss, l1))
    loss, l1[]
end
# the same on base codes: (L, n_groups * batch_size) UInt8, 0..3 = A,C,G,T, one read per column
function train_step!(m::ucdl, codes::Matrix{UInt8}, n_groups::Integer=1)
    loss = Vector{Float32}(undef, n_groups); l1 = Ref{Float32}(0)
    GC.@preserve codes loss check(ccall((:motifs_model_train_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ptr{Float32}, Ref{Float32}), getfield(m, :h), codes, n_groups, loss, l1))
    loss, l1[]
end

function setup_num_epochs(number_training_samples)  # train.jl:1-11
    number_training_samples < 1000 && return 25
    number_training_samples < 10000 && return 10
    number_training_samples < 100000 && return 5
    return 3
end

# train.jl:13-58.  Returns (cdl, hp, len, projs) like the reference; len and projs only encode "stride 4" and
# "needed lags" and have no counterpart here (nothing), the callers below do not use them.
# groups_per_step = 1 is the reference's schedule (one AdaBelief step per mini-batch of hp.batch_size reads).
function train_ucdl(data; num_epochs=nothing, l1_loss_thresh=float_type(95.0), hp::HParams=Hyperparam(),
                    groups_per_step::Integer=1, ctx::Context=context(), verbose::Bool=false)
    L4, _, N = size(data.data_matrix)
    cdl = ucdl(hp, L4 ÷ 4; ctx=ctx)
    num_epochs = isnothing(num_epochs) ? setup_num_epochs(N) : num_epochs
    B = Int(hp.batch_size); per_step = B * groups_per_step
    nfull = (N ÷ B) * B                              # DataLoader(partial=false)
    break_condition = false
    for i in 1:num_epochs
        order = Random.randperm(N)                   # DataLoader(shuffle=true)
        for i0 in 1:per_step:nfull
            idx = order[i0:min(i0 + per_step - 1, nfull)]
            g = length(idx) ÷ B
            S = data.data_matrix[:, :, idx[1:g * B]]
            loss, l1_loss = train_step!(cdl, S, g)
            verbose && println("loss $(sum(loss) / g)")   # model.jl:392
            if l1_loss < l1_loss_thresh
                break_condition = true
                break
            end
        end
        break_condition && break
        println("Epoch: $i completed")
    end
    return cdl, hp, nothing, nothing
end

# _1_code_retrieval.jl:33-56 — same signature; data.data_matrix: (4L, 1, N) Float32 one-hot
function code_retrieval(data, cdl::ucdl, hp=nothing, len=nothing, projs=nothing)
    dm = data.data_matrix
    N = size(dm, 3)
    cap = N * max(4 * Int(getfield(cdl, :hp).q), 64)
    out = Vector{stored_code_component_t}(undef, cap); n = Ref{Int64}(0)
    rc = GC.@preserve dm out ccall((:motifs_model_retrieve_codes, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ref{Int64}), getfield(cdl, :h), dm, 1, N, out, cap, n)
    if rc == 4                                       # MOTIFS_ERR_BUFFER_TOO_SMALL: n holds the required count
        cap = n[]; resize!(out, cap)
        rc = GC.@preserve dm out ccall((:motifs_model_retrieve_codes, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ref{Int64}), getfield(cdl, :h), dm, 1, N, out, cap, n)
    end
    check(rc)
    resize!(out, n[])
end

# ---- PWM scan: _h3_1_alignment.jl:38-112 -------------------------------------------------------------------------
data_(data; test=false) = test ? data.data_matrix_test : data.data_matrix
data_bg(data; test=false) = test ? data.data_matrix_bg_test : data.data_matrix_bg

function get_pos_scores_arr(ms, data; rc=false, bg=false, test=false, ctx::Context=context())
    data_matrix = bg ? data_bg(data; test=test) : data_(data; test=test)
    length(size(data_matrix)) == 2 && (data_matrix = reshape(data_matrix, (size(data_matrix, 1), 1, size(data_matrix, 2))))
    data_matrix = Array{Float32}(data_matrix)
    L4, _, N = size(data_matrix)
    K = ms.num_motifs; maxlen = maximum(ms.lens)
    pwms = zeros(float_type_retrieval, K, 4, maxlen)            # :65-67, forward bank; the library applies reverse() for rc
    for i in 1:K; pwms[i, :, 1:ms.lens[i]] = ms.pwms[i]; end
    lens = Int64.(ms.lens); n = Ref{Int64}(0)
    cap = max(1024, (N * (L4 ÷ 4) * K) ÷ 64)                    # first guess; a too-small buffer costs one more call
    found = Vector{record_t}(undef, cap); score = Vector{float_type_retrieval}(undef, cap)
    status = GC.@preserve pwms lens data_matrix found score ccall((:motifs_pwm_scan, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Cint, Ptr{Cvoid}, Ptr{UInt16},
         Int64, Ref{Int64}, Ptr{Int64}),
        ctx.h, pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, rc, found, score, cap, n, C_NULL)
    if status == 4                                              # MOTIFS_ERR_BUFFER_TOO_SMALL
        cap = n[]; resize!(found, cap); resize!(score, cap)
        status = GC.@preserve pwms lens data_matrix found score ccall((:motifs_pwm_scan, lib), Cint,
            (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Cint, Ptr{Cvoid}, Ptr{UInt16},
             Int64, Ref{Int64}, Ptr{Int64}),
            ctx.h, pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, rc, found, score, cap, n, C_NULL)
    end
    check(status)
    resize!(found, n[]); resize!(score, n[])
    return found, score
end

motifs_prep(ms) = ([Dict{Int, Vector{Int}}() for _ in 1:ms.num_motifs], [Dict{Int, Vector{float_type_retrieval}}() for _ in 1:ms.num_motifs],
                   [Dict{Int, Vector{Bool}}() for _ in 1:ms.num_motifs])     # _s1_make_motifs.jl:185-190

function modify_w_found!(found_record, score_record, positions, scores, use_comp; rc=false)   # :38-52, unchanged
    comp = rc ? true : false
    @inbounds for (f, s) in zip(found_record, score_record)
        m, n, l = f[1], f[2], f[3]
        if haskey(positions[m], n)
            push!(positions[m][n], l); push!(scores[m][n], s); push!(use_comp[m][n], comp)
        else
            positions[m][n] = [l]; scores[m][n] = [s]; use_comp[m][n] = [comp]
        end
    end
end

# :89-99.  discover_motifs scans the same reads again and again (data, shuffled background, held-out split: render.jl:70,72,
# pvec_calculations.jl:2-3), so by default the reads of a data matrix are encoded and uploaded ONCE (DeviceReads, kept per matrix object
# in `resident_reads`) and every later gpu_scan of that matrix runs on the resident codes: only the records cross PCIe (the host-matrix
# entry below moves 320 MB of Float32 per 100 000 x 200 bp call: 21.9 ms against 1.2 ms of kernels).  resident=false: the one-call form
# through motifs_pwm_scan_both, the Float32 matrix crossing PCIe once for the two strands.  A matrix that is MUTATED IN PLACE after its
# first scan must be dropped with forget_reads!(matrix) (or forget_reads!() for all).
const resident_reads = IdDict{Any, Any}()
function device_reads(data_matrix; ctx::Context=context())
    get!(resident_reads, data_matrix) do
        DeviceReads(Array{Float32}(data_matrix); ctx=ctx)
    end
end
function forget_reads!(data_matrix=nothing)
    if data_matrix === nothing
        foreach(r -> free!(r.codes), values(resident_reads)); empty!(resident_reads)
    elseif haskey(resident_reads, data_matrix)
        free!(resident_reads[data_matrix].codes); delete!(resident_reads, data_matrix)
    end
    nothing
end
function gpu_scan(ms, data; bg=false, test=false, ctx::Context=context(), resident::Bool=true)
    data_matrix = bg ? data_bg(data; test=test) : data_(data; test=test)
    resident && return gpu_scan(ms, device_reads(data_matrix; ctx=ctx))
    length(size(data_matrix)) == 2 && (data_matrix = reshape(data_matrix, (size(data_matrix, 1), 1, size(data_matrix, 2))))
    data_matrix = Array{Float32}(data_matrix)
    L4, _, N = size(data_matrix)
    K = ms.num_motifs; maxlen = maximum(ms.lens)
    pwms = zeros(float_type_retrieval, K, 4, maxlen)
    for i in 1:K; pwms[i, :, 1:ms.lens[i]] = ms.pwms[i]; end
    lens = Int64.(ms.lens); n2 = zeros(Int64, 2)
    cap = max(1024, (N * (L4 ÷ 4) * K) ÷ 64)
    found = [Vector{record_t}(undef, cap) for _ in 1:2]; score = [Vector{float_type_retrieval}(undef, cap) for _ in 1:2]
    f1 = found[1]; f2 = found[2]; s1 = score[1]; s2 = score[2]
    status = GC.@preserve pwms lens data_matrix f1 f2 s1 s2 n2 ccall((:motifs_pwm_scan_both, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{UInt16}, Ptr{Cvoid}, Ptr{UInt16},
         Int64, Ptr{Int64}, Ptr{Int64}),
        ctx.h, pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, f1, s1, f2, s2, cap, n2, C_NULL)
    if status == 4                                              # MOTIFS_ERR_BUFFER_TOO_SMALL: n2 holds the required counts
        cap = maximum(n2); resize!(f1, cap); resize!(f2, cap); resize!(s1, cap); resize!(s2, cap)
        status = GC.@preserve pwms lens data_matrix f1 f2 s1 s2 n2 ccall((:motifs_pwm_scan_both, lib), Cint,
            (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{UInt16}, Ptr{Cvoid}, Ptr{UInt16},
             Int64, Ptr{Int64}, Ptr{Int64}),
            ctx.h, pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, f1, s1, f2, s2, cap, n2, C_NULL)
    end
    check(status)
    resize!(f1, n2[1]); resize!(s1, n2[1]); resize!(f2, n2[2]); resize!(s2, n2[2])
    positions, scores, use_comp = motifs_prep(ms)
    modify_w_found!(f1, s1, positions, scores, use_comp; rc=false)
    modify_w_found!(f2, s2, positions, scores, use_comp; rc=true)
    return positions, scores, use_comp
end

function scan_w_gpu!(ms, data; bg=false, ctx::Context=context())               # :101-112
    positions, scores, use_comp = gpu_scan(ms, data; bg=bg, ctx=ctx)
    if bg
        ms.positions_bg = positions; ms.scores_bg = scores; ms.use_comp_bg = use_comp
    else
        ms.positions = positions; ms.scores = scores; ms.use_comp = use_comp
    end
end

# ---- device memory without CUDA.jl / AMDGPU.jl: the library allocates, Julia holds the pointer -----------------------------
mutable struct DeviceBuffer
    p::Ptr{Cvoid}
    bytes::Int
    ctx::Context
    function DeviceBuffer(ctx::Context, bytes::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:motifs_dev_alloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.h, bytes, r))
        b = new(r[], bytes, ctx)
        finalizer(free!, b)
        b
    end
end
function free!(b::DeviceBuffer)
    (b.p == C_NULL || b.ctx.h == C_NULL) || ccall((:motifs_dev_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), b.ctx.h, b.p)
    b.p = C_NULL
    nothing
end
Base.pointer(b::DeviceBuffer) = b.p
function upload!(b::DeviceBuffer, a::Array)
    sizeof(a) <= b.bytes || error("upload!: $(sizeof(a)) bytes into a buffer of $(b.bytes)")
    GC.@preserve a check(ccall((:motifs_dev_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), b.ctx.h, b.p, a, sizeof(a)))
    b
end
function download!(a::Array, b::DeviceBuffer)
    sizeof(a) <= b.bytes || error("download!: $(sizeof(a)) bytes from a buffer of $(b.bytes)")
    GC.@preserve a check(ccall((:motifs_dev_download, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), b.ctx.h, a, b.p, sizeof(a)))
    a
end
memset!(b::DeviceBuffer, byte::Integer=0) =
    check(ccall((:motifs_dev_memset, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Csize_t), b.ctx.h, b.p, byte, b.bytes))
device_array(ctx::Context, a::Array) = upload!(DeviceBuffer(ctx, max(sizeof(a), 16)), a)

codes_bytes(N::Integer, L::Integer) = Int(ccall((:motifs_codes_bytes, lib), Csize_t, (Int64, Cint), N, L))
codes_pitch(L::Integer) = Int(ccall((:motifs_codes_pitch, lib), Cint, (Cint,), L))

# loadfasta/helpers.jl:83-139 (`reading` / `read_fasta` + base coding): (L, N) UInt8 codes 0..3, one read per column
function fasta_read(path::AbstractString; max_entries::Integer=100000)
    n = Ref{Int64}(0); L = Ref{Int32}(0)
    check(ccall((:motifs_fasta_read, lib), Cint, (Cstring, Int64, Ptr{UInt8}, Int64, Ref{Int64}, Ref{Int32}), path, max_entries, C_NULL, 0, n, L))
    out = Matrix{UInt8}(undef, Int(L[]), Int(n[]))
    GC.@preserve out check(ccall((:motifs_fasta_read, lib), Cint, (Cstring, Int64, Ptr{UInt8}, Int64, Ref{Int64}, Ref{Int32}),
                                 path, max_entries, out, length(out), n, L))
    out
end

# Reads kept on the device in the library's own layout (1 byte per base): made once, scanned as often as needed
# (data, shuffled background, held-out split: render.jl:70,72, pvec_calculations.jl:2-3 re-scan the same reads).
struct DeviceReads
    codes::DeviceBuffer
    N::Int
    L::Int
    ctx::Context
end
function DeviceReads(data_matrix::Array{Float32}; ctx::Context=context())          # (4L, 1, N) or (4L, N) one-hot
    L4 = size(data_matrix, 1); N = size(data_matrix, ndims(data_matrix)); L = L4 ÷ 4
    raw = device_array(ctx, data_matrix)
    codes = DeviceBuffer(ctx, codes_bytes(N, L)); bad = device_array(ctx, Int32[0])
    check(ccall((:motifs_encode_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx.h, raw.p, 1, N, L, codes.p, bad.p))
    flag = download!(Int32[0], bad)
    free!(raw); free!(bad)
    flag[1] == 0 || error("data matrix has a column that is neither one-hot nor all-zero")
    DeviceReads(codes, N, L, ctx)
end
function DeviceReads(codes::Matrix{UInt8}; ctx::Context=context())                  # (L, N) base codes, e.g. fasta_read
    L, N = size(codes)
    raw = device_array(ctx, codes)
    dc = DeviceBuffer(ctx, codes_bytes(N, L))
    check(ccall((:motifs_encode_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx.h, raw.p, 0, N, L, dc.p, C_NULL))
    synchronize(ctx); free!(raw)
    DeviceReads(dc, N, L, ctx)
end

function padded_bank(ms)                                           # _h3_1_alignment.jl:65-67
    K = ms.num_motifs; maxlen = maximum(ms.lens)
    pwms = zeros(float_type_retrieval, K, 4, maxlen)
    for i in 1:K; pwms[i, :, 1:ms.lens[i]] = ms.pwms[i]; end
    pwms, Int64.(ms.lens), K, maxlen
end

# Device-resident records of one gpu_scan: hit buffers and counts stay on the device for the consumers below.
struct DeviceHits
    hits::Vector{DeviceBuffer}       # forward, reverse: record_t each
    scores::Vector{DeviceBuffer}     # Float16 each
    n::Vector{Int64}                 # records per strand
    counts::DeviceBuffer             # 2K Int64: per-PWM hit counts [forward K][reverse K]
end
# gpu_scan (:89-99) on resident reads through motifs_pwm_scan_hits_both_dev; n0 = global index of the first read minus one
function gpu_scan_dev(ms, reads::DeviceReads; n0::Integer=0, batch::Integer=batch_size_greedy)
    ctx = reads.ctx
    pwms, lens, K, maxlen = padded_bank(ms)
    n2 = zeros(Int64, 2)
    GC.@preserve pwms lens n2 check(ccall((:motifs_pwm_scan_hits_both_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Int64, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Int64, Ptr{Int64}, Ptr{Cvoid}),
        ctx.h, pwms, lens, K, maxlen, reads.codes.p, reads.N, reads.L, n0, batch, C_NULL, C_NULL, C_NULL, C_NULL, 0, n2, C_NULL))
    cap = max(maximum(n2), 1)
    hits = [DeviceBuffer(ctx, cap * sizeof(record_t)) for _ in 1:2]; scores = [DeviceBuffer(ctx, cap * 2) for _ in 1:2]
    counts = DeviceBuffer(ctx, 2 * K * 8)
    GC.@preserve pwms lens n2 check(ccall((:motifs_pwm_scan_hits_both_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Int64, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Int64, Ptr{Int64}, Ptr{Cvoid}),
        ctx.h, pwms, lens, K, maxlen, reads.codes.p, reads.N, reads.L, n0, batch, hits[1].p, scores[1].p, hits[2].p, scores[2].p, cap, n2, counts.p))
    DeviceHits(hits, scores, n2, counts)
end
# ... and as the reference's dictionaries (the records cross PCIe once, the reads not at all)
function gpu_scan(ms, reads::DeviceReads; n0::Integer=0)
    dh = gpu_scan_dev(ms, reads; n0=n0)
    positions, scores, use_comp = motifs_prep(ms)
    for s in 1:2
        f = download!(Vector{record_t}(undef, dh.n[s]), dh.hits[s]); sc = download!(Vector{float_type_retrieval}(undef, dh.n[s]), dh.scores[s])
        modify_w_found!(f, sc, positions, scores, use_comp; rc=(s == 2))
    end
    foreach(free!, dh.hits); foreach(free!, dh.scores); free!(dh.counts)
    return positions, scores, use_comp
end
# greedy_search! (:18-36) as the reference launches it: the dense (K, N, ld_l) Float16 tensor, on the device
function greedy_search_dev(ms, reads::DeviceReads; ld_l::Integer=reads.L - minimum(ms.lens) + 1)
    pwms, lens, K, maxlen = padded_bank(ms)
    out = DeviceBuffer(reads.ctx, K * reads.N * ld_l * 2)
    GC.@preserve pwms lens check(ccall((:motifs_pwm_scan_dense_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Int64),
        reads.ctx.h, pwms, lens, K, maxlen, reads.codes.p, reads.N, reads.L, out.p, ld_l))
    out
end

# ---- consumers of the records on the device (SURVEY.md §8f; _s2_filter_pos_w_scores.jl, _h6_positions2countmat.jl) ---------
# get_min_score / get_max_score (:11-35) of one strand's records: (min, max) Float16 per PWM, +Inf / -Inf where a PWM has none
function hits_minmax(ctx::Context, hits::DeviceBuffer, scores::DeviceBuffer, n::Integer, K::Integer)
    mn = DeviceBuffer(ctx, 2K); mx = DeviceBuffer(ctx, 2K)
    check(ccall((:motifs_hits_minmax_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx.h, hits.p, scores.p, n, K, mn.p, mx.p))
    a = download!(Vector{Float16}(undef, K), mn); b = download!(Vector{Float16}(undef, K), mx)
    free!(mn); free!(mx)
    a, b
end
# get_hits (:3-9) for a whole sweep: thr (T, K) Float16 ascending per PWM (pad with Inf); counts (T, K) += #{score > thr}
function hits_threshold_counts!(counts::DeviceBuffer, ctx::Context, hits::DeviceBuffer, scores::DeviceBuffer, n::Integer, thr::Matrix{Float16})
    T, K = size(thr)
    d = device_array(ctx, thr)
    check(ccall((:motifs_hits_threshold_counts_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}),
                ctx.h, hits.p, scores.p, n, K, d.p, T, counts.p))
    synchronize(ctx); free!(d)
    counts
end
# filter_position_by_best_thresh! (:116-125): records with score > thresh[m], order kept; returns the number kept
function hits_filter!(out_hits::DeviceBuffer, out_scores::DeviceBuffer, ctx::Context, hits::DeviceBuffer, scores::DeviceBuffer, n::Integer,
                      thresh::Vector{Float16})
    d = device_array(ctx, thresh); kept = Ref{Int64}(0)
    check(ccall((:motifs_hits_filter_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}),
                ctx.h, hits.p, scores.p, n, length(thresh), d.p, out_hits.p, out_scores.p, kept))
    free!(d)
    kept[]
end
# posdicts2countmats (_h6:26-55) without the pseudo-count: (4, maxlen, K) UInt32 counts += the one-hot window of every record
function hits_count_matrices!(counts::DeviceBuffer, reads::DeviceReads, hits::DeviceBuffer, n::Integer, lens::Vector{Int64}, maxlen::Integer;
                              n0::Integer=0, comp::Bool=false)
    GC.@preserve lens check(ccall((:motifs_hits_count_matrices_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Cint, Cint, Cint, Ptr{Cvoid}),
        reads.ctx.h, hits.p, n, reads.codes.p, reads.L, n0, lens, length(lens), maxlen, comp, counts.p))
    counts
end

# ---- consumers of the code records (SURVEY.md §8f-4; _2_enumerate.jl) --------------------------------------------------------
codes_mag_histogram!(hist::DeviceBuffer, ctx::Context, recs::DeviceBuffer, n::Integer) =      # 65536 UInt32 bins over the Float16 magnitudes
    check(ccall((:motifs_codes_mag_histogram_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), ctx.h, recs.p, n, hist.p))
function codes_filter!(out::DeviceBuffer, ctx::Context, recs::DeviceBuffer, n::Integer, thresh::Real)    # :10-13, second half
    kept = Ref{Int64}(0)
    check(ccall((:motifs_codes_filter_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Ref{Int64}), ctx.h, recs.p, n, thresh, out.p, kept))
    kept[]
end
function triplets_offsets!(offsets::DeviceBuffer, ctx::Context, range_len::DeviceBuffer, nranges::Integer)      # :50-65: C(len, 3) per range, scanned
    total = Ref{Int64}(0)
    check(ccall((:motifs_triplets_offsets_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ref{Int64}), ctx.h, range_len.p, nranges, offsets.p, total))
    total[]
end
triplets_enumerate!(keys::DeviceBuffer, vals::DeviceBuffer, ctx::Context, recs::DeviceBuffer, range_start::DeviceBuffer, range_len::DeviceBuffer,
                    nranges::Integer, h::Integer, offsets::DeviceBuffer, cap::Integer) =
    check(ccall((:motifs_triplets_enumerate_dev, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                ctx.h, recs.p, range_start.p, range_len.p, nranges, h, offsets.p, keys.p, vals.p, cap))
function triplets_group!(uniq::DeviceBuffer, first::DeviceBuffer, counts::DeviceBuffer, group_off::DeviceBuffer, perm::DeviceBuffer,
                         ctx::Context, keys::DeviceBuffer, n::Integer)                                           # the Dictionary of insert_H! (:37-46)
    nu = Ref{Int64}(0)
    check(ccall((:motifs_triplets_group_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}),
                ctx.h, keys.p, n, uniq.p, first.p, counts.p, group_off.p, perm.p, nu))
    nu[]
end

# ---- multi-GPU: one Context per device, RCCL behind the ABI (no reference counterpart; SURVEY.md §8e) -----------------------
mutable struct Comm
    h::Ptr{Cvoid}
    ctx::Context
end
# one Julia process driving the n devices of a node (ncclCommInitAll)
function comm_create_all(ctxs::Vector{Context})
    hs = [c.h for c in ctxs]; out = fill(Ptr{Cvoid}(C_NULL), length(ctxs))
    GC.@preserve hs out check(ccall((:motifs_comm_create_all, lib), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}), hs, length(ctxs), out))
    [Comm(out[i], ctxs[i]) for i in eachindex(ctxs)]
end
# one process (or task) per device: rank 0 makes the id, the host carries it to the others
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    GC.@preserve id check(ccall((:motifs_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    id
end
function comm_create(ctx::Context, id::Vector{UInt8}, nranks::Integer, rank::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:motifs_comm_create, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}), ctx.h, id, nranks, rank, r))
    Comm(r[], ctx)
end
Base.close(c::Comm) = (c.h == C_NULL || ccall((:motifs_comm_destroy, lib), Cvoid, (Ptr{Cvoid},), c.h); c.h = C_NULL; nothing)
function comm_rank(c::Comm)
    r = Ref{Cint}(0); n = Ref{Cint}(0)
    check(ccall((:motifs_comm_rank, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), c.h, r, n))
    Int(r[]), Int(n[])
end
group_start() = check(ccall((:motifs_comm_group_start, lib), Cint, ()))
group_end() = check(ccall((:motifs_comm_group_end, lib), Cint, ()))
allreduce_sum_f32!(c::Comm, buf_dev::Ptr{Cvoid}, n::Integer) =
    check(ccall((:motifs_comm_allreduce_sum_f32_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), c.h, buf_dev, n))
allreduce_sum_i64!(c::Comm, buf_dev::Ptr{Cvoid}, n::Integer) =
    check(ccall((:motifs_comm_allreduce_sum_i64_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), c.h, buf_dev, n))
# sharded count matrices (posdicts2countmats sums over all reads): UInt32 sums
allreduce_sum_u32!(c::Comm, buf_dev::Ptr{Cvoid}, n::Integer) =
    check(ccall((:motifs_comm_allreduce_sum_u32_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), c.h, buf_dev, n))
allreduce_grad!(m::ucdl, c::Comm, grad_flat_dev::Ptr{Cvoid}) =
    check(ccall((:motifs_model_allreduce_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), getfield(m, :h), c.h, grad_flat_dev))
hist_allreduce!(c::Comm, counts_dev::Ptr{Cvoid}, K::Integer, n_strands::Integer=1) =
    check(ccall((:motifs_hist_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), c.h, counts_dev, K, n_strands))
# one data-parallel optimiser step on device-resident codes of this rank's mini-batches (motifs_encode_dev layout)
dp_train_step!(m::ucdl, c::Union{Comm, Nothing}, codes_dev::Ptr{Cvoid}, n_groups_local::Integer, n_groups_total::Integer,
               loss_dev::Ptr{Cvoid}, grad_flat_dev::Ptr{Cvoid}) =
    check(ccall((:motifs_model_dp_train_step_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                getfield(m, :h), c === nothing ? C_NULL : c.h, codes_dev, n_groups_local, n_groups_total, loss_dev, grad_flat_dev))

# ---- one Julia process, several devices, HOST data: no device pointer crosses into Julia ------------------------------------------
# One optimiser step of the replicas `cdls` (identical parameters, one per device, comms = comm_create_all(ctxs)) on the
# DataLoader batch S (4L, 1, n_groups * batch_size) Float32: mini-batches are dealt to the devices in contiguous blocks; every
# device's gradient, then the grouped all-reduces, then every device's AdaBelief.  Returns (losses in batch order, l1 of replica 1).
function dp_train_step!(cdls::Vector{ucdl}, comms::Union{Vector{Comm}, Nothing}, S::Array{Float32,3}, n_groups::Integer)
    ms = [getfield(m, :h) for m in cdls]; cs = comms === nothing ? Ptr{Cvoid}[] : [c.h for c in comms]
    loss = Vector{Float32}(undef, n_groups); l1 = Ref{Float32}(0)
    GC.@preserve ms cs S loss check(ccall((:motifs_model_dp_train_step_host, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Float32}, Ref{Float32}),
        ms, comms === nothing ? C_NULL : pointer(cs), length(ms), S, 1, n_groups, loss, l1))
    loss, l1[]
end
# The same step on device-resident shards (codes_dev[d]: motifs_encode_dev layout on device d; loss_dev / grad_dev: DeviceBuffers of
# n_groups_local[d] and nD + nF + nV floats on device d).  NOT to be wrapped in group_start / group_end: it opens its own group
# around the all-reduces only.
function dp_train_step_all!(cdls::Vector{ucdl}, comms::Vector{Comm}, codes_dev::Vector{DeviceBuffer}, n_groups_local::Vector{<:Integer},
                            n_groups_total::Integer, loss_dev::Vector{DeviceBuffer}, grad_dev::Vector{DeviceBuffer})
    ms = [getfield(m, :h) for m in cdls]; cs = [c.h for c in comms]; nl = Cint.(n_groups_local)
    cd = [b.p for b in codes_dev]; ld = [b.p for b in loss_dev]; gd = [b.p for b in grad_dev]
    GC.@preserve ms cs nl cd ld gd check(ccall((:motifs_model_dp_train_step_all, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
        ms, cs, length(ms), cd, nl, n_groups_total, ld, gd, C_NULL))
end
# the three phases for a caller that schedules them itself: only allreduce_grad! may sit between group_start() and group_end()
dp_grad!(m::ucdl, codes_dev::Ptr{Cvoid}, n_groups_local::Integer, loss_dev::Ptr{Cvoid}, grad_flat_dev::Ptr{Cvoid}) =
    check(ccall((:motifs_model_dp_grad_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                getfield(m, :h), codes_dev, n_groups_local, loss_dev, grad_flat_dev))
dp_update!(m::ucdl, grad_flat_dev::Ptr{Cvoid}, n_groups_total::Integer) =
    check(ccall((:motifs_model_dp_update_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), getfield(m, :h), grad_flat_dev, n_groups_total))
allreduce_sum_f32_to!(c::Comm, send_dev::Ptr{Cvoid}, recv_dev::Ptr{Cvoid}, n::Integer) =
    check(ccall((:motifs_comm_allreduce_sum_f32_to_dev, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64), c.h, send_dev, recv_dev, n))

# gpu_scan (:89-99) of a host matrix over the devices of ctxs.  shard_align = batch_size_greedy: the two record lists are the
# single-device ones bit for bit (20 ordering batches over 8 devices = 3,3,3,3,2,2,2,2); shard_align = 1: even shards, the lists are
# in sequence-block-major order and the dictionaries below are the single-device ones all the same (:38-52 only depend on the order
# within one (m, n)).
function gpu_scan_sharded(ms, data, ctxs::Vector{Context}; comms::Union{Vector{Comm}, Nothing}=nothing, bg=false, test=false,
                          shard_align::Integer=batch_size_greedy)
    data_matrix = bg ? data_bg(data; test=test) : data_(data; test=test)
    data_matrix = Array{Float32}(data_matrix)
    L4 = size(data_matrix, 1); N = size(data_matrix, ndims(data_matrix))
    pwms, lens, K, maxlen = padded_bank(ms)
    hs = [c.h for c in ctxs]; cs = comms === nothing ? Ptr{Cvoid}[] : [c.h for c in comms]
    n2 = zeros(Int64, 2); counts = zeros(Int64, K, 2)
    cap = max(1024, (N * (L4 ÷ 4) * K) ÷ 64)
    f1 = Vector{record_t}(undef, cap); f2 = Vector{record_t}(undef, cap)
    s1 = Vector{float_type_retrieval}(undef, cap); s2 = Vector{float_type_retrieval}(undef, cap)
    status = GC.@preserve hs cs pwms lens data_matrix f1 f2 s1 s2 n2 counts ccall((:motifs_pwm_scan_both_sharded, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Int64, Ptr{Cvoid}, Ptr{UInt16},
         Ptr{Cvoid}, Ptr{UInt16}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
        hs, comms === nothing ? C_NULL : pointer(cs), length(hs), pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, shard_align, f1, s1, f2, s2, cap, n2,
        counts, C_NULL)
    if status == 4                                              # MOTIFS_ERR_BUFFER_TOO_SMALL: n2 holds the required counts
        cap = maximum(n2); resize!(f1, cap); resize!(f2, cap); resize!(s1, cap); resize!(s2, cap)
        status = GC.@preserve hs cs pwms lens data_matrix f1 f2 s1 s2 n2 counts ccall((:motifs_pwm_scan_both_sharded, lib), Cint,
            (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Int64, Ptr{Cvoid}, Ptr{UInt16},
             Ptr{Cvoid}, Ptr{UInt16}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
            hs, comms === nothing ? C_NULL : pointer(cs), length(hs), pwms, lens, K, maxlen, data_matrix, 1, N, L4 ÷ 4, shard_align, f1, s1, f2, s2, cap, n2,
            counts, C_NULL)
    end
    check(status)
    resize!(f1, n2[1]); resize!(s1, n2[1]); resize!(f2, n2[2]); resize!(s2, n2[2])
    positions, scores, use_comp = motifs_prep(ms)
    modify_w_found!(f1, s1, positions, scores, use_comp; rc=false)
    modify_w_found!(f2, s2, positions, scores, use_comp; rc=true)
    return positions, scores, use_comp, counts
end

end # module
