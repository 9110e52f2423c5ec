// api_common.h — context object and error plumbing shared by the C-ABI files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include <string>
#include <thread>
#include <vector>

#include "../../include/motifs_hip.h"

namespace motifs {

void set_error(const char* fmt, ...);

// A grow-only device buffer (never shrinks; freed with the context).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return e;
        cap = want;
        // MOTIFS_POISON_WS=1 (debugging aid): a new workspace starts as 0xFF bytes instead of whatever the allocator hands out,
        // so a kernel that reads a cell, entry or slot nothing wrote gives the same wrong answer every time
        static const bool poison = getenv("MOTIFS_POISON_WS") != nullptr;
        if (poison) {
            e = hipMemset(p, 0xFF, want);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// A PWM bank packed for the device, kept between calls: scanning the same bank again (the other strand's
// twin, the next shard, the second call of the count-then-fill protocol) skips packing and upload.
struct BankSlot {
    std::vector<uint8_t> key;   // the caller's bank bytes + shape; empty = nothing cached
    DevBuf tab, lim, afrag, cinit, tabk;
    int tabk_stride = 0;
    int KP = 0, nch = 0, lenp = 0, minlen = 0, maxlen_true = 0, ntiles = 0, uniform_eps = 0;
};

enum KernelSlot {
    KS_ENCODE = 0,
    KS_SCAN_DENSE = 1,
    KS_SCAN_COUNT = 2,
    KS_SCAN_OFFSETS = 3,
    KS_SCAN_FILL = 4,
    KS_TRAIN_STEP = 5,
    KS_TRAIN_ISTA_BWD = 6,
    KS_COUNT_
};

}  // namespace motifs

struct motifs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t timing = 0;            // bit s set: launches of slot s are timed
    size_t ws_limit = (size_t)8 << 30;   // bound of the scan's candidate/staging workspace (motifs_ctx_set_workspace_limit)
    bool scan_valu = false;      // MOTIFS_SCAN_VALU=1: hit records through the all-VALU mask kernel (cross-check path)
    // timing: event pairs are recorded around launches without synchronising and
    // resolved when the totals are read (motifs_ctx_kernel_ms)
    struct TimedSpan {
        int slot;
        hipEvent_t e0, e1;
    };
    std::vector<TimedSpan> pending;
    std::vector<hipEvent_t> free_events;
    double kernel_ms[motifs::KS_COUNT_] = {0};
    int64_t kernel_launches[motifs::KS_COUNT_] = {0};
    // scan workspaces
    motifs::DevBuf tab, lim, cnt, off, tilesum, small, codes, hits_tmp, scores_tmp, pwmcnt, data_tmp, afrag, cinit;
    motifs::DevBuf staging, rowx;   // matrix-core scan: staged hit words, per-row offsets
    motifs::DevBuf dp_scratch;      // host-buffer data-parallel step: flat gradient + losses
    motifs::DevBuf centries;        // matrix-core scan: compact 16-bit entries of the candidate cells (scan_mfma.hip)
    bool compact_cells = true;      // compact entries where the bank allows them (uniform slack, PWMs of up to 20 positions); 128-bit cells otherwise
    motifs::DevBuf cnt2, centries2; // the reverse strand's cells / entries when one candidate launch serves both strands of gpu_scan
    motifs::DevBuf cm_lens;         // motifs_hits_count_matrices_dev: the PWM lengths of the last call (uploaded again only when they change)
    std::vector<int32_t> cm_lens_host;
    int cg_chunks = -1;             // chunk groups of the re-scoring (scan_mfma.hip): -1 = when the table does not fit the LDS; MOTIFS_CG_CHUNKS overrides
    bool dense_fused = true;        // a17's tensor in one kernel (scan_dense.hip) where the bank fits it; candidate kernel + stage_hits<.., 2> otherwise
    int32_t scan_plan[4] = {0, 0, 0, 0};   // motifs_ctx_scan_plan
    bool pair_launches = true;      // both strands in one launch of stage_hits / row scans / emit_records too (per strand: chunk groups, several super-batches)
    bool fuse_strands = true;       // one candidate launch for both strands where both banks take the four-reads kernel with compact entries
    bool records_async = false;     // motifs_ctx_set_records_in_stream_order: the both-strands scan returns once the totals are on the host
    hipEvent_t ev_totals = nullptr; // ... recorded behind the row scans of that call

    bool ev_totals_set = false;     // (this call recorded it)
    int64_t ticket_seq = 0;         // records in stream order, one-launch plan: the row scan writes this call's ticket into pinned memory behind the
    bool ticket_wait = false;       // totals and the host polls for it (this call asked for it)
    motifs::BankSlot bank_slot[2];  // [rc]
    void* pinned = nullptr;  // small pinned host block for totals / flags
    // pinned staging of the host-buffer entries (motifs_pwm_scan*): code rows on the way up, record chunks on the way down
    void* pin_stage = nullptr;
    size_t pin_stage_cap = 0;
};

#define MOTIFS_HIP_CHECK(expr)                                                                   \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            motifs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MOTIFS_ERR_HIP;                                                               \
        }                                                                                        \
    } while (0)

namespace motifs {

// RAII timer around a group of launches on the context stream.  Records two
// events on the stream the kernels run on; nothing waits until the totals are read.
struct KernelTimer {
    motifs_ctx* c;
    int slot;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    static hipEvent_t get(motifs_ctx* c) {
        if (!c->free_events.empty()) {
            hipEvent_t e = c->free_events.back();
            c->free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    bool attached;     // the events are handed to one launch (hipExtLaunchKernelGGL stamps them); nothing is recorded here
    bool stamped = true;   // attached form: cleared by the caller when the launch that should stamp the events failed
    KernelTimer(motifs_ctx* ctx, int s, bool attach = false) : c(ctx), slot(s), attached(attach) {
        if ((c->timing >> slot) & 1u) {
            e0 = get(c);
            e1 = get(c);
            if (!attached) (void)hipEventRecord(e0, c->stream);
        }
    }
    ~KernelTimer() {
        if (e0 && e1) {
            if (!attached) (void)hipEventRecord(e1, c->stream);
            if (stamped) c->pending.push_back({slot, e0, e1});
            else c->free_events.push_back(e0), c->free_events.push_back(e1);   // never recorded: nothing to resolve
        }
    }
};

// shared host-side helpers of the host-buffer entries (scan_api.hip)
int upload_and_encode(motifs_ctx* c, const void* data, int kind, int64_t N, int L);   // host matrix of `kind` -> c->codes
int download_chunked(motifs_ctx* c, void* dst, const void* src_dev, size_t bytes);    // device bytes -> pageable host memory
const char* last_error_text();                                                        // this thread's motifs_last_error()
// ncclGroupStart / ncclGroupEnd nesting depth on this thread (comm_rccl.hip): inside an open group RCCL only records a
// collective and launches it at the closing ncclGroupEnd, so nothing that consumes its result may be enqueued before that
int comm_group_depth();
motifs_ctx* comm_ctx(motifs_comm* c);     // the context a communicator rank was made on (nullptr for nullptr)

// run fn(d) for d in [0, n): on the calling thread for n == 1, else one host thread per device (each binds its own device);
// returns the first non-zero status and leaves that thread's error text as this thread's
template <typename F>
inline int for_each_device(int n, F&& fn) {
    if (n == 1) return fn(0);
    std::vector<int> rc(n, 0);
    std::vector<std::string> why(n);
    std::vector<std::thread> th;
    th.reserve(n);
    for (int d = 0; d < n; d++)
        th.emplace_back([&, d]() {
            rc[d] = fn(d);
            if (rc[d]) why[d] = last_error_text();
        });
    for (auto& t : th) t.join();
    for (int d = 0; d < n; d++)
        if (rc[d]) {
            set_error("device slot %d: %s", d, why[d].c_str());
            return rc[d];
        }
    return MOTIFS_OK;
}

inline void resolve_timing(motifs_ctx* c) {
    for (auto& sp : c->pending) {
        (void)hipEventSynchronize(sp.e1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess) {
            c->kernel_ms[sp.slot] += ms;
            c->kernel_launches[sp.slot] += 1;
        }
        c->free_events.push_back(sp.e0);
        c->free_events.push_back(sp.e1);
    }
    c->pending.clear();
}

}  // namespace motifs
