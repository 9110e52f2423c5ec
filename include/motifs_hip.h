/*
 * motifs_hip.h — C ABI of libmotifs_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the motif-scanning hot path of MOTIFs.jl
 * (SURVEY.md §8b).  The reference has no FFI of its own: the entry points
 * below replace ordinary Julia call sites, each cited as
 * `path:line` relative to the reference checkout.  Julia binds them with
 * `ccall` (see INTEGRATION.md and julia/MotifsHIP.jl); tests and bench.py
 * bind the same symbols with ctypes.
 *
 * Conventions
 *   - every function returns an int status (MOTIFS_OK == 0); the text of the
 *     last failure on the calling thread is motifs_last_error();
 *   - no C++ exception, torch type or library-allocated buffer crosses the ABI;
 *   - host buffers are owned by the caller and only read/written during the
 *     call; `*_dev` entry points take device pointers (hipMalloc'ed or
 *     torch-allocated) and enqueue on the context's stream without
 *     synchronising;
 *   - array layouts are the reference's: column-major, 1-based indices inside
 *     records.
 */
#ifndef MOTIFS_HIP_H
#define MOTIFS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOTIFS_ABI_VERSION 3

enum motifs_status {
    MOTIFS_OK = 0,
    MOTIFS_ERR_INVALID = 1,      /* bad argument (shape, null pointer, range)            */
    MOTIFS_ERR_HIP = 2,          /* a HIP runtime call failed                            */
    MOTIFS_ERR_NO_DEVICE = 3,    /* no gfx950 device visible: there is NO CPU fallback   */
    MOTIFS_ERR_BUFFER_TOO_SMALL = 4, /* *n_out holds the required record count           */
    MOTIFS_ERR_NOT_ONEHOT = 5,   /* a data column is not exactly one-hot / all-zero      */
    MOTIFS_ERR_NONFINITE = 6,    /* PWM bank holds NaN/Inf (reference semantics differ)  */
    MOTIFS_ERR_UNSUPPORTED = 7,  /* e.g. the engine arena is too small for n_groups       */
    MOTIFS_ERR_COMM = 8          /* RCCL missing or a collective failed                  */
};

/* Encodings accepted for a sequence matrix. */
enum motifs_data_kind {
    MOTIFS_DATA_CODES_U8 = 0,   /* N rows of L bytes, 0..3 = A,C,G,T, 4 = all-zero column   */
    MOTIFS_DATA_ONEHOT_F32 = 1, /* reference `data.data_matrix` (4L,1,N) Float32,           */
                                /* loadfasta/helpers.jl:110-139, fasta.jl:78-81             */
    MOTIFS_DATA_ONEHOT_F16 = 2  /* what _h3_1_alignment.jl:74 uploads                       */
};

/* Order of hit records. */
enum motifs_hit_order {
    /* exactly the reference's: per 5000-sequence batch, `findall` column-major
     * over (k, n, l): k fastest, then n, then l (_h3_1_alignment.jl:71-84).   */
    MOTIFS_ORDER_REFERENCE = 0
};

#define MOTIFS_SCAN_MAX_LEN 64          /* longest PWM on the compile-time-length kernels.  NOT a cap (the reference has
                                         * none: _h3_1_alignment.jl:25-31; motif length is d13 + h, _2_enumerate.jl:43):
                                         * a longer bank is scanned by run-time-length kernels, same records, slower */
#define MOTIFS_SCAN_BATCH 5000          /* batch_size_greedy, _h3_1_alignment.jl:12  */

typedef struct motifs_ctx motifs_ctx;   /* opaque: device id, stream, workspaces */

/* One hit = the reference's record_t NTuple{3,UInt32} (_h3_1_alignment.jl:10):
 * (m, n, l) = (PWM index, global sequence index, start position), all 1-based. */
typedef struct motifs_hit {
    uint32_t m;
    uint32_t n;
    uint32_t l;
} motifs_hit;

/* ---- context ------------------------------------------------------------ */

int motifs_abi_version(void);
const char* motifs_last_error(void);

/* Fails with MOTIFS_ERR_NO_DEVICE when no GPU is visible. */
int motifs_ctx_create(int device, motifs_ctx** out);
void motifs_ctx_destroy(motifs_ctx* ctx);
/* A context starts on a private non-blocking stream.  motifs_ctx_set_stream makes it enqueue on the caller's
 * hipStream_t instead (NULL = HIP's null stream, which is what torch's default "current stream" is);
 * motifs_ctx_get_stream returns the stream in use, so a host framework can order its own work against it
 * (e.g. torch.cuda.ExternalStream).  Every `*_dev` entry point and every collective below runs on that stream. */
int motifs_ctx_set_stream(motifs_ctx* ctx, void* hip_stream);
int motifs_ctx_get_stream(motifs_ctx* ctx, void** hip_stream_out);
/* Back to a private non-blocking stream of the library's own (the state a fresh context is in); the stream in use is
 * drained first.  (ABI 1 spelled this motifs_ctx_set_stream(ctx, NULL); since ABI 2 that binds HIP's null stream.) */
int motifs_ctx_use_private_stream(motifs_ctx* ctx);
int motifs_ctx_synchronize(motifs_ctx* ctx);
/* Upper bound in bytes for the scan's candidate / staging workspace (0 = the default, 8 GiB).  A scan that needs
 * more walks the reads in super-batches of whole ordering batches; the records do not depend on the bound. */
int motifs_ctx_set_workspace_limit(motifs_ctx* ctx, size_t bytes);
/* Records in stream order (off by default).  on != 0: motifs_pwm_scan_hits_both_dev returns as soon as the hit totals are known -
 * the row scans have run - while the kernel that writes the records may still be running: hits / scores are complete for
 * everything queued on the context's stream afterwards and after motifs_ctx_synchronize, as a hipMemcpyAsync's destination is.
 * A host loop over many shards then prepares its next call under the record writes instead of after them (the single-launch
 * plan only: a scan that needs several super-batches or runs in chunk groups still returns when all of it is done).
 * The wait for the totals is a poll of a pinned host word the row scan writes behind them (no event is recorded in the stream); a host core spins meanwhile.
 * MOTIFS_ERR_BUFFER_TOO_SMALL is returned only after the stream has drained (the first `cap` records of each strand are written,
 * nothing past `cap` is touched), so the caller may release or re-allocate hits / scores at once. */
int motifs_ctx_set_records_in_stream_order(motifs_ctx* ctx, int on);
/* Per-kernel device time, measured with HIP events on the context stream
 * around every launch of that kernel (each timed launch synchronises, so leave
 * timing off outside measurements).  `slot` is a motifs_kernel_slot; *ms is the
 * total since the last reset and *launches the number of timed launches. */
enum motifs_kernel_slot {
    MOTIFS_KS_ENCODE = 0,
    MOTIFS_KS_SCAN_DENSE = 1,
    MOTIFS_KS_SCAN_COUNT = 2,   /* scan_kernel<LEN,MASK>: all windows, `> 0` test, 128-bit hit masks */
    MOTIFS_KS_SCAN_OFFSETS = 3, /* fill_row_sums + fill_row_scan: record offsets                   */
    MOTIFS_KS_SCAN_FILL = 4,    /* fill_records: (m, n, l) + fp16 score per set mask bit           */
    MOTIFS_KS_TRAIN_STEP = 5,   /* the whole forward/backward graph of motifs_model_loss_grad_dev  */
    MOTIFS_KS_TRAIN_ISTA_BWD = 6 /* k_zy_step2_bwd, the VJP of update_ZY's fused ISTA step (model.jl:237-245): the largest kernel
                                  * by time of a many-mini-batch step; stamped only on steps that run outside a captured graph */
};
/* on = 0: off; 1: every slot; otherwise a set of slots, (1 << (slot + 1)) or-ed together (an event pair costs a few
 * microseconds of stream time per timed section: time only what is being reported). */
int motifs_ctx_enable_timing(motifs_ctx* ctx, int on);
int motifs_ctx_reset_timing(motifs_ctx* ctx);
int motifs_ctx_kernel_ms(motifs_ctx* ctx, int slot, double* ms, int64_t* launches);
/* How the last hit-record scan on this context was laid out (diagnostics for tests and benchmarks; the records do not
 * depend on it): plan[0] = 1 when the candidates travelled as compact 16-bit entries, plan[1] = chunks of 128 PWMs per
 * chunk group of the re-scoring (0: the whole table sat in one block's LDS or was gathered from L2), plan[2] = chunk
 * groups, plan[3] = super-batch launches of the last strand scanned. */
int motifs_ctx_scan_plan(motifs_ctx* ctx, int32_t plan[4]);

/* ---- device memory for a host without a GPU array package ----------------- */

/* north_star keeps the host in Julia without CUDA.jl / AMDGPU.jl, so a Julia caller has no way of its own to make the
 * device pointers the `*_dev` entry points take: these five give it one.  Buffers belong to the caller (free them before
 * motifs_ctx_destroy); upload / download block until the bytes have arrived (the copy is enqueued on the context's
 * stream, behind whatever the library queued before); memset is enqueued like a kernel. */
int motifs_dev_alloc(motifs_ctx* ctx, size_t bytes, void** out_dev);
int motifs_dev_free(motifs_ctx* ctx, void* ptr_dev);
int motifs_dev_upload(motifs_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int motifs_dev_download(motifs_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int motifs_dev_memset(motifs_ctx* ctx, void* dst_dev, int byte_value, size_t bytes);

/* ---- sequence encoding (input side of every kernel) ----------------------- */

/* Bytes needed for the internal code matrix of N sequences of length L
 * (rows padded to a multiple of 4 bytes + a guard). */
size_t motifs_codes_bytes(int64_t N, int L);
/* Row pitch in bytes of that matrix. */
int motifs_codes_pitch(int L);
/* Convert a device-resident matrix of `kind` into the internal code matrix.
 * `bad_flag_dev` (int32 on device, may be NULL) is set non-zero if a column is
 * neither one-hot nor all-zero. */
int motifs_encode_dev(motifs_ctx* ctx, const void* data_dev, int kind, int64_t N, int L,
                      uint8_t* codes_dev, int32_t* bad_flag_dev);

/* ---- PWM scan: a17 greedy_search! --------------------------------------- */

/* Replaces the kernel launch at _h3_1_alignment.jl:76-80.
 * pwms_fp16: (K,4,maxlen) column-major IEEE binary16 bits, zero-padded beyond
 *            lens[k] (host pointer; _h3_1_alignment.jl:66-69);
 * codes_dev: internal code matrix of N sequences (motifs_encode_dev);
 * scores_dev: (K, N, ld_l) column-major fp16, ld_l >= L - min(lens) + 1.
 *            Entries with l <= L-lens[k]+1 get max(score,0); entries beyond, up
 *            to ld_l, get +0 (the reference pre-zeroes the tensor, :75, with
 *            ld_l = 4L).
 * Score = sequential fp16 sum over ind = 1..lens[k] of
 *         pwms[k, base(l+ind-1), ind], one rounding per add (:26-31). */
int motifs_pwm_scan_dense_dev(motifs_ctx* ctx, const uint16_t* pwms_fp16, const int64_t* lens,
                              int K, int maxlen, const uint8_t* codes_dev, int64_t N, int L,
                              uint16_t* scores_dev, int64_t ld_l);

/* ---- PWM scan: a18 get_pos_scores_arr ------------------------------------ */

/* Replaces get_pos_scores_arr (_h3_1_alignment.jl:57-87) for one strand:
 * scan + threshold (> 0) + ordered compaction, all on the device.
 * hits_dev / hit_scores_dev: device buffers of `cap` records (may be NULL with
 *            cap == 0: count only);
 * n_out:     host; total number of hits (also when the buffer is too small);
 * per_pwm_counts_dev: optional K int64 on device, hit count per PWM (the
 *            histogram all-reduced across GPUs);
 * n0:        global index of the first sequence minus one (records carry
 *            n + n0, _h3_1_alignment.jl:83), so shards can be concatenated;
 * batch:     sequences per ordering batch (MOTIFS_SCAN_BATCH for the
 *            reference's order).
 * `rc` != 0 scans with reverse(pwm) in both dims (:68-69); pass the forward
 * bank, the library builds the reverse one. */
int motifs_pwm_scan_hits_dev(motifs_ctx* ctx, const uint16_t* pwms_fp16, const int64_t* lens,
                             int K, int maxlen, const uint8_t* codes_dev, int64_t N, int L, int rc,
                             int64_t n0, int batch, motifs_hit* hits_dev, uint16_t* hit_scores_dev,
                             int64_t cap, int64_t* n_out, int64_t* per_pwm_counts_dev);

/* gpu_scan (src/inference/_h3_1_alignment.jl:89-99): both strands of one shard in one call.  Same arguments as
 * motifs_pwm_scan_hits_dev, with one record buffer pair per strand (forward, reverse: `cap` records each), n_out2 =
 * host int64[2] (forward, reverse totals) and per_pwm_counts2_dev = optional 2*K int64 on device ([forward K][reverse K]).
 * The reverse-strand kernels are enqueued behind the forward ones; the host waits once. */
int motifs_pwm_scan_hits_both_dev(motifs_ctx* ctx, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen,
                                  const uint8_t* codes_dev, int64_t N, int L, int64_t n0, int batch,
                                  motifs_hit* hits_fwd_dev, uint16_t* scores_fwd_dev, motifs_hit* hits_rc_dev,
                                  uint16_t* scores_rc_dev, int64_t cap, int64_t* n_out2, int64_t* per_pwm_counts2_dev);

/* There is no `score_mode` argument (SURVEY.md 8b sketched fp32 / MFMA-tolerance modes): the only mode is the
 * reference's own arithmetic, bit for bit - the matrix cores are used as an exact pre-filter behind it, so an approximate
 * mode would not be faster.
 *
 * Host-buffer form of the same call: what Julia's `ccall` binds.  `data` is a
 * host matrix of `kind`; hits / hit_scores / per_pwm_counts are host buffers.
 * hits == NULL or cap too small: returns MOTIFS_ERR_BUFFER_TOO_SMALL (or
 * MOTIFS_OK when hits == NULL && cap == 0) with *n_out = required count. */
int motifs_pwm_scan(motifs_ctx* ctx, const uint16_t* pwms_fp16, const int64_t* lens, int K,
                    int maxlen, const void* data, int kind, int64_t N, int L, int rc,
                    motifs_hit* hits, uint16_t* hit_scores, int64_t cap, int64_t* n_out,
                    int64_t* per_pwm_counts);

/* gpu_scan (_h3_1_alignment.jl:89-99) on host buffers: one upload of `data`, both strands, two record lists of `cap`
 * entries each; n_out2 = {forward, reverse} totals; per_pwm_counts2 = optional 2*K host int64. */
int motifs_pwm_scan_both(motifs_ctx* ctx, const uint16_t* pwms_fp16, const int64_t* lens, int K, int maxlen, const void* data,
                         int kind, int64_t N, int L, motifs_hit* hits_fwd, uint16_t* scores_fwd, motifs_hit* hits_rc,
                         uint16_t* scores_rc, int64_t cap, int64_t* n_out2, int64_t* per_pwm_counts2);

/* ---- convolutional sparse coding: src/model.jl, src/train.jl, _1_code_retrieval.jl ------------ */

/* Hyperparam (model.jl:1-14); f_len = 4*filter_len and twoM = 2*M are derived. */
typedef struct motifs_hparams {
    int32_t filter_len, M, h, K, q, batch_size, num_pass_xyz, num_pass_df;
    float magnifying_factor, gamma;
} motifs_hparams;

/* stored_code_component_t (_0_const.jl:3-4): (position::UInt16, fil::UInt16, seq::UInt32,
 * mag::Float16) with Julia's isbits layout (12 bytes: offsets 0, 2, 4, 8). */
typedef struct motifs_code_rec {
    uint16_t position;
    uint16_t fil;
    uint32_t seq;
    uint16_t mag;   /* IEEE binary16 bits */
    uint16_t pad_;
} motifs_code_rec;

typedef struct motifs_model motifs_model;   /* opaque: the `ucdl` state + AdaBelief moments + engine arena */

/* Replaces train.jl:29-35 (Hyperparam(), length_info, projectors, ucdl(hp), Flux.params, AdaBelief()).
 * L: sequence length in bp.  arena_bytes: device memory for intermediates (0 = 8 GiB); one mini-batch of
 * BASELINE configs[1] needs ~0.5 GiB with gradients. */
int motifs_model_create(motifs_ctx* ctx, const motifs_hparams* hp, int L, size_t arena_bytes, motifs_model** out);
void motifs_model_destroy(motifs_model* m);
/* nD = 4*filter_len*M, nF = h*2M*K, nV = 33 with the default pass counts; c, l as in length_info. */
int motifs_model_sizes(motifs_model* m, int64_t* nD, int64_t* nF, int64_t* nV, int64_t* c, int64_t* l);
/* Filter-bank layouts are the reference's (model.jl:84-90): D (f_len,1,M) and F (h,twoM,1,K), column-major.
 * warmup3 = {lambda_sparsity_warmup, lambda_stepsize_warmup, omega_stepsize_warmup} (plain scalars, not
 * trained: model.jl:68,72,73); vecs = lambda_sparsity | kappa_sparsity | lambda_stepsize | omega_stepsize |
 * kappa_stepsize | penalty_xyz | mu, raw (un-squared) values.  NULL leaves a block untouched. */
int motifs_model_set_params(motifs_model* m, const float* D, const float* F, const float* warmup3, const float* vecs);
int motifs_model_get_params(motifs_model* m, float* D, float* F, float* warmup3, float* vecs);
/* ucdl(hp) initial values (model.jl:84-100) from a seeded stream. */
int motifs_model_init_random(motifs_model* m, uint64_t seed);
/* Replaces `gradient(ps) do forward_pass_return_loss(...) end` (train.jl:42-44) for n_groups independent
 * mini-batches at once.  codes_dev: n_groups*batch_size rows (motifs_encode_dev layout).
 * loss_dev[n_groups]: the reference's loss per mini-batch; grad_flat_dev[nD+nF+nV]: SUM over the
 * mini-batches of the gradient w.r.t. [D | F | vecs] (NULL: forward only). */
int motifs_model_loss_grad_dev(motifs_model* m, const uint8_t* codes_dev, int n_groups, float* loss_dev,
                               float* grad_flat_dev, int keep_intermediates);
/* Replaces Flux.Optimise.update!(opt, ps, gs) with AdaBelief() (train.jl:35,46); uses gscale*grad. */
int motifs_model_adabelief_dev(motifs_model* m, const float* grad_flat_dev, float gscale);
/* sum(abs.(prep_syntax_filters(cdl.F))) (train.jl:47). */
int motifs_model_l1_syntax(motifs_model* m, float* out);
/* Host-buffer form of one loop body of train.jl:40-52 (what Julia's ccall binds): codes = n_groups *
 * batch_size rows of L bytes (0..3).  n_groups == 1 is exactly the reference's step. */
int motifs_model_train_step(motifs_model* m, const uint8_t* codes, int n_groups, float* loss_out, float* l1F_out);
/* The same step on the reference's own batch format: S = `data.data_matrix[:, :, batch]`, the bytes of a
 * (4L, 1, n_groups*batch_size) Float32 one-hot array (train.jl:33,41; loadfasta/helpers.jl:110-139), so a Julia
 * caller passes the DataLoader's batch unchanged.  MOTIFS_ERR_NOT_ONEHOT if a column is neither one-hot nor zero. */
int motifs_model_train_step_onehot(motifs_model* m, const float* S, int n_groups, float* loss_out, float* l1F_out);
/* Replaces code_retrieval (_1_code_retrieval.jl:33-56).  data: host matrix of `kind`, N sequences. */
int motifs_model_retrieve_codes(motifs_model* m, const void* data, int kind, int64_t N, motifs_code_rec* out,
                                int64_t cap, int64_t* n_out);
/* Measurement hook (bench.py, SURVEY.md 8d "conv forward scan"): a4 alone -- warmup_ZY's two convolutions of the reads
 * with the filter bank (model.jl:171-173), one Toeplitz GEMM over the expanded bank -- launched `reps` times for
 * n_groups mini-batches; *ms_out = average device time per launch (HIP events on the context's stream). */
int motifs_model_time_filter_scan(motifs_model* m, const uint8_t* codes_dev, int n_groups, int reps, float* ms_out);
/* Measurement hook (bench.py, SURVEY.md 8d "MFMA fraction is computed on the syntax-layer GEMM"): a7's dense contraction alone --
 * conv(ZY, F, flipped=true) of model.jl:214,251: rows = reads x l, columns = K, reduction = h * 2M, 2 * l * K * h * 2M flop per
 * read -- launched `reps` times for n_groups mini-batches; *ms_out = average device time per launch. */
int motifs_model_time_syntax_conv(motifs_model* m, const uint8_t* codes_dev, int n_groups, int reps, float* ms_out);
/* Bytes of the engine arena the steps so far have used at most (intermediates + tape of the largest step): what `arena_bytes`
 * of motifs_model_create has to cover. */
int motifs_model_arena_peak(motifs_model* m, size_t* bytes);
/* Test hook: a named intermediate of the last loss_grad call made with keep_intermediates != 0. */
int motifs_model_dump(motifs_model* m, const char* name, float* out, int64_t cap, int64_t* n);

/* ---- multi-GPU: RCCL over xGMI behind the ABI (SURVEY.md §5 last row, §8e) ----------------------------- */

/* The reference is single-GPU (src/MOTIFs.jl:4-8 imports no communication package): these entry points are new.
 * Reads shard over devices in contiguous blocks; parameters and the PWM bank are replicated; the only exchanges are
 * one sum of the flat gradient [dD | dF | dvecs] per optimiser step and one sum of the K hit counts per scan.
 * librccl.so is opened on first use (MOTIFS_ERR_COMM if it cannot be); every collective is in place and is
 * enqueued on the stream of the communicator's context, behind the kernels that produced its operand. */
typedef struct motifs_comm motifs_comm;
#define MOTIFS_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 makes the id, the host carries it to the other ranks (any channel). */
int motifs_comm_unique_id(uint8_t id[MOTIFS_COMM_ID_BYTES]);
/* One rank of an nranks communicator on ctx's device (ncclCommInitRank): one process per GPU, or one host thread
 * per device.  Collective: returns when every rank has joined. */
int motifs_comm_create(motifs_ctx* ctx, const uint8_t id[MOTIFS_COMM_ID_BYTES], int nranks, int rank, motifs_comm** out);
/* All ranks of a single-process communicator over the devices of ctxs[0..n_dev) (ncclCommInitAll): the form a
 * single Julia process driving the 8 GPUs of a node uses; bracket the per-device collective calls of one step
 * with motifs_comm_group_start / _end (ncclGroupStart / ncclGroupEnd).  Only collectives belong inside a group: RCCL
 * launches them at the closing _end, so a kernel that reads a sum must be enqueued after it (see
 * motifs_model_dp_train_step_all). */
int motifs_comm_create_all(motifs_ctx* const* ctxs, int n_dev, motifs_comm** out);
void motifs_comm_destroy(motifs_comm* comm);   /* before the context it was made on */
int motifs_comm_rank(motifs_comm* comm, int* rank, int* nranks);
int motifs_comm_group_start(void);
int motifs_comm_group_end(void);
int motifs_comm_allreduce_sum_f32_dev(motifs_comm* comm, float* buf_dev, int64_t n);
int motifs_comm_allreduce_sum_i64_dev(motifs_comm* comm, int64_t* buf_dev, int64_t n);
/* uint32 sums: the count matrices of motifs_hits_count_matrices_dev built per shard (posdicts2countmats,
 * src/inference/_h6_positions2countmat.jl:26-55, sums its windows over ALL reads). */
int motifs_comm_allreduce_sum_u32_dev(motifs_comm* comm, uint32_t* buf_dev, int64_t n);
/* Out of place: recv_dev = sum over ranks of send_dev (send_dev is left as it was). */
int motifs_comm_allreduce_sum_f32_to_dev(motifs_comm* comm, const float* send_dev, float* recv_dev, int64_t n);
/* Sum over ranks of the flat gradient motifs_model_loss_grad_dev wrote (nD + nF + nV floats). */
int motifs_model_allreduce_grad(motifs_model* m, motifs_comm* comm, float* grad_flat_dev);
/* Sum over ranks of the per-PWM hit counts of a scan (K int64 per strand, n_strands = 1 or 2), on the context's stream. */
int motifs_hist_allreduce(motifs_comm* comm, int64_t* per_pwm_counts_dev, int K, int n_strands);
/* One data-parallel optimiser step: gradient of this rank's n_groups_local mini-batches (0 is allowed: the rank
 * contributes zeros and still takes part in the exchange) -> sum over ranks -> AdaBelief with the mean over the
 * n_groups_total mini-batches of all ranks, identical on every rank, so the replicas stay bit-identical.
 * comm == NULL: single device.  grad_flat_dev: nD + nF + nV floats of scratch; loss_dev: n_groups_local floats. */
int motifs_model_dp_train_step_dev(motifs_model* m, motifs_comm* comm, const uint8_t* codes_dev, int n_groups_local,
                                   int64_t n_groups_total, float* loss_dev, float* grad_flat_dev);
/* NOT inside motifs_comm_group_start / _end: between them RCCL only records a collective and launches it at the closing
 * ncclGroupEnd, so the AdaBelief kernel this call enqueues behind its all-reduce would run BEFORE the sum.  The call
 * refuses (MOTIFS_ERR_INVALID) while a group is open on the calling thread.  One host thread driving several devices
 * uses motifs_model_dp_train_step_all below, or the three phases of the step themselves:
 *   motifs_model_dp_grad_dev    this device's summed gradient (zeros for n_groups_local == 0)      -- every device
 *   motifs_comm_group_start();  motifs_model_allreduce_grad per device;  motifs_comm_group_end();  -- the only grouped part
 *   motifs_model_dp_update_dev  AdaBelief on gradient / n_groups_total                             -- every device */
int motifs_model_dp_grad_dev(motifs_model* m, const uint8_t* codes_dev, int n_groups_local, float* loss_dev, float* grad_flat_dev);
int motifs_model_dp_update_dev(motifs_model* m, const float* grad_flat_dev, int64_t n_groups_total);
/* One data-parallel optimiser step of n_dev replicas driven by ONE host thread (the communicators of
 * motifs_comm_create_all): every device's gradient is enqueued, then ncclGroupStart, every device's all-reduce,
 * ncclGroupEnd, then every device's AdaBelief - in that order on each device's stream.  models / comms / codes_dev /
 * n_groups_local / loss_dev / grad_dev: arrays of n_dev entries (device d's model lives on comms[d]'s context);
 * reduced_dev: NULL (sum in place in grad_dev[d]) or n_dev buffers of nD + nF + nV floats that receive the sum while
 * grad_dev[d] keeps device d's own gradient.  comms == NULL is allowed for n_dev == 1. */
int motifs_model_dp_train_step_all(motifs_model* const* models, motifs_comm* const* comms, int n_dev,
                                   const uint8_t* const* codes_dev, const int* n_groups_local, int64_t n_groups_total,
                                   float* const* loss_dev, float* const* grad_dev, float* const* reduced_dev);
/* The same step on HOST data (what a Julia process without device pointers calls): `data` holds n_groups * batch_size
 * reads of `kind` in loader order; mini-batches are dealt to the n_dev replicas in contiguous blocks, uploaded and
 * encoded by one host thread per device, and the step above runs.  loss_out[n_groups] in the order of `data`;
 * l1F_out (optional): sum(abs.(prep_syntax_filters(F))) after the step (train.jl:47), from replica 0. */
int motifs_model_dp_train_step_host(motifs_model* const* models, motifs_comm* const* comms, int n_dev, const void* data,
                                    int kind, int n_groups, float* loss_out, float* l1F_out);
/* gpu_scan (_h3_1_alignment.jl:89-99) of a HOST matrix over the n_dev devices of ctxs: reads shard in contiguous
 * blocks whose edges are multiples of shard_align, device d scans its block with n0 = its first read (one host thread
 * per device: upload + encode, count, fill, download), the record lists are concatenated in device order into the
 * caller's buffers and the 2 x K hit histograms are summed (RCCL all-reduce inside one group when comms != NULL, on the
 * host otherwise).  shard_align = MOTIFS_SCAN_BATCH: the concatenation IS the single-device record list, bit for bit;
 * shard_align = 1 (or any other value): even blocks, global order = sequence-block-major, and for every (m, n) the
 * records still come forward strand first, ascending l within a strand - the order the dictionaries of
 * modify_w_found! (_h3_1_alignment.jl:38-52) depend on.  Other arguments as motifs_pwm_scan_both;
 * shard_counts (optional): n_dev x 2 record counts per device {forward, reverse}. */
int motifs_pwm_scan_both_sharded(motifs_ctx* const* ctxs, motifs_comm* const* comms, int n_dev, const uint16_t* pwms_fp16,
                                 const int64_t* lens, int K, int maxlen, const void* data, int kind, int64_t N, int L,
                                 int64_t shard_align, motifs_hit* hits_fwd, uint16_t* scores_fwd, motifs_hit* hits_rc,
                                 uint16_t* scores_rc, int64_t cap, int64_t* n_out2, int64_t* per_pwm_counts2,
                                 int64_t* shard_counts);

/* ---- either side of the scan ----------------------------------------------------- */

/* `reading`/`read_fasta` + base coding (loadfasta/helpers.jl:83-139): reads with N/n dropped, first
 * max_entries kept (the reference uses 100000, helpers.jl:2), then only reads as long as the first.
 * Host only.  codes_out: n_reads rows of L bytes (0..3); NULL queries *n_reads and *L. */
int motifs_fasta_read(const char* path, int64_t max_entries, uint8_t* codes_out, int64_t cap_bytes, int64_t* n_reads,
                      int32_t* L);
/* get_min_score / get_max_score (_s2_filter_pos_w_scores.jl:11-35) over one record array: binary16 bits per
 * PWM; +Inf / -Inf where a PWM has no record.  Combine data and background on the host (min of mins). */
int motifs_hits_minmax_dev(motifs_ctx* ctx, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n, int K,
                           uint16_t* min_dev, uint16_t* max_dev);
/* get_hits (:3-9) for a whole threshold sweep at once: counts[m][j] += #{records of PWM m with
 * score > thr[m][j]}; thr_dev: K rows of T ascending binary16 thresholds (pad with +Inf). */
int motifs_hits_threshold_counts_dev(motifs_ctx* ctx, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n,
                                     int K, const uint16_t* thr_dev, int T, int64_t* counts_dev);
/* filter_position_by_best_thresh! (:116-125): keep records with score > thresh[m], order preserved. */
int motifs_hits_filter_dev(motifs_ctx* ctx, const motifs_hit* hits_dev, const uint16_t* scores_dev, int64_t n, int K,
                           const uint16_t* thresh_dev, motifs_hit* out_hits_dev, uint16_t* out_scores_dev, int64_t* n_out);
/* posdicts2countmats (_h6_positions2countmat.jl:26-55) without the pseudo-count: counts_dev has the bytes of
 * a (4, maxlen, K) UInt32 array and is incremented by the one-hot window of every record; comp != 0 adds the
 * reverse complement (submat_comlement, _3_make_pfms.jl:49-52).  codes_dev row 0 is global sequence n0 + 1. */
int motifs_hits_count_matrices_dev(motifs_ctx* ctx, const motifs_hit* hits_dev, int64_t n, const uint8_t* codes_dev, int L,
                                   int64_t n0, const int64_t* lens, int K, int maxlen, int comp, uint32_t* counts_dev);

/* ---- consumers of the code records (src/inference/_2_enumerate.jl) --------------------- */

/* value_type (_0_const.jl:9-10): (seq_num::UInt32, pos::UInt16, comp::Bool); comp is always false here
 * (create_value, _2_enumerate.jl:2). */
typedef struct motifs_triplet_val {
    uint32_t seq_num;   /* the index of the scanning range, 1-based (enumerate_triplets passes `ind`, :53) */
    uint16_t pos;       /* position of the first component */
    uint16_t comp;
} motifs_triplet_val;
/* composition_key_type (_0_const.jl:6-7) packed into 64 bits: f1 << 48 | f2 << 40 | f3 << 32 | d12 << 16 | d13
 * (len = d13 + h is implied). */

/* filter_code_components_using_quantile! (:10-13), first half: hist_dev[b] = number of records whose
 * magnitude has the binary16 bit pattern b (65536 counters).  The caller takes the two order statistics
 * Statistics.quantile interpolates between from it. */
int motifs_codes_mag_histogram_dev(motifs_ctx* ctx, const motifs_code_rec* recs_dev, int64_t n, uint32_t* hist_dev);
/* second half: keep the records with Float64(mag) > thresh, order preserved. */
int motifs_codes_filter_dev(motifs_ctx* ctx, const motifs_code_rec* recs_dev, int64_t n, double thresh,
                            motifs_code_rec* out_dev, int64_t* n_out);
/* enumerate_triplets (:50-65): offsets_dev[r] = triplets of the ranges before range r (C(len, 3) each). */
int motifs_triplets_offsets_dev(motifs_ctx* ctx, const uint32_t* range_len_dev, int64_t nranges, int64_t* offsets_dev,
                                int64_t* total);
/* For every scanning range (0-based start, length; at most 256 records) the records are sorted by position
 * (stable, :57) and every i < j < k (:59-63) yields a packed key and a value at offsets_dev[r] + its rank:
 * exactly the order in which insert_H! (:37-46) sees them.  Entries past cap are dropped. */
int motifs_triplets_enumerate_dev(motifs_ctx* ctx, const motifs_code_rec* recs_dev, const uint32_t* range_start_dev,
                                  const uint32_t* range_len_dev, int64_t nranges, int h, const int64_t* offsets_dev,
                                  uint64_t* keys_dev, motifs_triplet_val* vals_dev, int64_t cap);
/* The Dictionary those insertions build: unique keys in first-insertion order (uniq_keys_dev, first_dev =
 * index of the first triplet with the key, counts_dev), group_off_dev = exclusive scan of the counts, and
 * perm_dev[group_off[t] .. +counts[t]) = the triplet indices of key t in insertion order.  All outputs
 * hold n entries at most. */
int motifs_triplets_group_dev(motifs_ctx* ctx, const uint64_t* keys_dev, int64_t n, uint64_t* uniq_keys_dev,
                              int64_t* first_dev, int64_t* counts_dev, int64_t* group_off_dev, int64_t* perm_dev,
                              int64_t* n_unique);

#ifdef __cplusplus
}
#endif
#endif /* MOTIFS_HIP_H */
