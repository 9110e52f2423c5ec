# MotifsHIP.jl — thin ccall shim over libmotifs_hip.so (include/motifs_hip.h).
# NOT executed in the build image (Julia is not installed there); it transliterates the four call sites of
# INTEGRATION.md.  No CUDA.jl / AMDGPU.jl / Flux / NNlib on the path.
module MotifsHIP

const lib = get(ENV, "MOTIFS_HIP_LIB", "libmotifs_hip.so")

struct HParams           # motifs_hparams == Hyperparam (src/model.jl:1-14)
    filter_len::Int32; M::Int32; h::Int32; K::Int32; q::Int32; batch_size::Int32
    num_pass_xyz::Int32; num_pass_df::Int32; magnifying_factor::Float32; gamma::Float32
end
HParams(; filter_len=8, M=50, h=12, K=24, q=32, batch_size=6, num_pass_xyz=6, num_pass_df=3,
        magnifying_factor=10f0, gamma=0.1f0) =
    HParams(filter_len, M, h, K, q, batch_size, num_pass_xyz, num_pass_df, magnifying_factor, gamma)

const CodeRec = NamedTuple{(:position, :fil, :seq, :mag), Tuple{UInt16, UInt16, UInt32, Float16}}  # _0_const.jl:3-4

check(rc) = rc == 0 || error(unsafe_string(ccall((:motifs_last_error, lib), Cstring, ())))

function context(device::Integer=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:motifs_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
    h[]
end

function model(ctx, hp::HParams, L::Integer; seed=nothing, arena_bytes=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:motifs_model_create, lib), Cint, (Ptr{Cvoid}, Ref{HParams}, Cint, Csize_t, Ref{Ptr{Cvoid}}),
                ctx, Ref(hp), L, arena_bytes, h))
    seed === nothing || check(ccall((:motifs_model_init_random, lib), Cint, (Ptr{Cvoid}, UInt64), h[], seed))
    h[]
end

# train.jl:40-52 — codes: (L, n_groups*batch_size) UInt8 matrix, bases 0..3, one read per column
function train_step!(m, codes::Matrix{UInt8}, n_groups::Integer)
    loss = Vector{Float32}(undef, n_groups); l1 = Ref{Float32}(0)
    GC.@preserve codes loss check(ccall((:motifs_model_train_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ptr{Float32}, Ref{Float32}), m, codes, n_groups, loss, l1))
    loss, l1[]
end

# _1_code_retrieval.jl:33-56 — data_matrix: (4L, 1, N) Float32 one-hot
function code_retrieval(m, data_matrix::Array{Float32,3}; cap=size(data_matrix, 3) * 128)
    out = Vector{CodeRec}(undef, cap); n = Ref{Int64}(0)
    GC.@preserve data_matrix out check(ccall((:motifs_model_retrieve_codes, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ref{Int64}),
        m, data_matrix, 1, size(data_matrix, 3), out, cap, n))
    resize!(out, n[])
end

# _h3_1_alignment.jl:57-87
function get_pos_scores_arr(ctx, pwm_list::Vector{Matrix{Float16}}, lens::Vector{Int}, data_matrix::Array{Float32,3}; rc=false)
    L4, _, N = size(data_matrix); K = length(pwm_list); maxlen = maximum(lens)
    pwms = zeros(Float16, K, 4, maxlen)
    for i in 1:K; pwms[i, :, 1:lens[i]] = pwm_list[i]; end
    lens64 = Int64.(lens); n = Ref{Int64}(0)
    sig = (Ptr{Cvoid}, Ptr{UInt16}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Cint, Int64, Cint, Cint, Ptr{Cvoid}, Ptr{UInt16},
           Int64, Ref{Int64}, Ptr{Int64})
    GC.@preserve pwms lens64 data_matrix begin
        check(ccall((:motifs_pwm_scan, lib), Cint, sig, ctx, pwms, lens64, K, maxlen, data_matrix, 1, N, L4 ÷ 4, rc,
                    C_NULL, C_NULL, 0, n, C_NULL))
        found = Vector{NTuple{3,UInt32}}(undef, n[]); score = Vector{Float16}(undef, n[])
        check(ccall((:motifs_pwm_scan, lib), Cint, sig, ctx, pwms, lens64, K, maxlen, data_matrix, 1, N, L4 ÷ 4, rc,
                    found, score, n[], n, C_NULL))
        return found, score
    end
end

end # module
