// scan_kernels.h — argument blocks and launchers of the PWM-scan kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace motifs {

constexpr int SCAN_BLOCK = 512;           // 8 waves
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;
constexpr int SCAN_GUARD_BYTES = 128;     // readable zero bytes behind the last code row
constexpr int FILL_THREADS = 256;
constexpr int FILL_HIST_MAX = 8192;       // LDS histogram bins of fill_records (PWMs, padded)

enum { MODE_DENSE = 0, MODE_MASK = 1 };

struct HitRec {
    uint32_t m, n, l;
};

// scalars of one scan launch (passed by value; pointers travel as separate
// __restrict__ kernel parameters so the sequence rows can use scalar loads)
struct ScanDims {
    int64_t N;
    int L, pitch;
    int K, KP, nch;           // PWMs, padded pairs (nch*64), chunks of 64 pairs
    int Lout, LoutP;          // L - minlen + 1; positions per mask row (see scan_lout_padded)
    int lim_min;              // L - maxlen: starts <= lim_min are valid for every PWM
    int spw;                  // sequences per wave
    int cpb;                  // PWM chunks handled side by side in one block (1, 2, 4 or 8)
    int k_even;
    int batch;                // ordering batch (5000)
};

struct ScanArgs {
    const uint32_t* tab;      // [(ind*4 + b) * KP + kp] half2 {pwm 2kp, pwm 2kp+1}
    const int32_t* lim;       // [2*KP] last valid 0-based start per PWM (L - len), -1 if absent
    const uint8_t* codes;     // N rows of `pitch` bytes
    uint16_t* scores;         // DENSE: (K, N, ld_l) col-major
    uint4* masks;             // MASK: [(batch, p, n-in-batch, chunk)] 128-bit hit masks {lo halves, hi halves}
    ScanDims d;
};

struct FastDivHost {
    uint32_t d, m, s;
};

struct FillArgs {
    const uint16_t* centries; // compact entries of the cells (nullptr: the cells themselves are the candidates), 2 per cell
    const uint4* masks;       // [row = (batch, p)][cell = (n-in-batch, chunk)]
    int64_t nrows;            // batches * LoutP
    uint32_t row_cells;       // batch * nch
    uint32_t* row_sum;        // [nrows]
    int64_t* row_base;        // [nrows] exclusive
    int64_t* total;
    const uint32_t* tab;
    const uint8_t* codes;
    HitRec* hits;
    uint16_t* hit_scores;
    int64_t* pwm_counts;      // optional [2*KP]
    int64_t base0;            // records emitted before this super-batch
    int64_t n0;               // global index offset for records
    int nch, batch, Lout, LoutP, lshift, lenp, KP, pitch;
    int hist_bins;            // 2*KP when the LDS histogram is on, else 0
    const int32_t* lim;       // matrix-core path: last valid start per PWM
    int64_t N;                // reads in this super-batch (cells of later reads are empty)
    int K;
    int L;                    // read length (matrix-core path; lens[k] = L - lim[k])
    int lim_min;              // starts <= lim_min are valid for every PWM
    // matrix-core path (stage_hits / emit_records): a (batch, l) line of cells is split into `parts` rows of `rpr` reads
    int parts, rpr;
    const uint16_t* tabk;         // [K][tabk_stride] binary16 re-scoring table: [ind][5] per PWM, column 4 = +0
    int tabk_stride;              // halves per PWM row (an odd number of dwords)
    uint32_t* staging;            // [nrows][row_slots] staged hit words
    uint16_t* dense;              // mode 2: (K, N, ld_l) score tensor; planes l < Lout are written in full
    int row_slots;
    uint32_t* row_excl;           // [nrows] hits before the row inside its block of 1024 rows
    unsigned long long* blk_base; // [ceil(nrows / 1024)] block totals, then records before the block
    const int64_t* base_in;       // records emitted before this super-batch (nullptr: none, the first one)
    int64_t* total_host;          // optional pinned host copy of *total (the host reads it after its stream wait)
    int64_t* ticket_host;         // optional: pinned word that takes `ticket` once *total_host is written (records in stream order: the host
    int64_t ticket;               // polls it instead of waiting on an event - an event record leaves a 15-20 us hole in the stream)
    int64_t cap;                  // records the output arrays can hold
    // chunk-group mode (banks whose re-scoring table does not fit the LDS; scan_mfma.hip "chunk groups"): cgc = chunks of 128 PWMs
    // per group (1, 2 or 4; 0 = off), ncg = groups.  A row is then rpr = 512 / cgc reads x nch chunks, row_sum / staging are kept
    // per (row, group): row_sum[row * ncg + cg], staging[(row * ncg + cg) * row_slots ...]
    int cgc, ncg;
    struct {
        uint32_t d, m, s;
        __device__ uint32_t div(uint32_t n) const {
            if ((d & (d - 1)) == 0) return n >> __builtin_ctz(d);      // 1, 2, 4, .. chunks: a shift

            const uint32_t t = __umulhi(n, m);
            return (t + ((n - t) >> 1)) >> (s - 1);
        }
    } div_nch;
};

// ---- matrix-core candidate path (scan_mfma.hip) ----
struct CandDims {
    int64_t N;
    int L, pitch, Lout, nch, batch, spw;
    int cgc;                  // chunks per chunk group of the ENTRY / cell layout (= nch: the plain (batch, l, read, chunk) order; smaller with
                              // chunk groups: (batch, group, l, read, chunk in group), so that a group's cells are contiguous for its consumer)
    int ohlen;                // positions in a read's one-hot LDS image (covers every window tile + the PWM length)
    int used_tiles;           // tiles that hold at least one PWM: ceil(K / 32)
    // four-reads kernel only (set by its launcher): a block takes `seg_tiles` window tiles of 8 starts of its reads (grid z = segment) and stages
    // `ohseg` positions of their images - long reads then leave the CU as many blocks as the registers do, and a small shard has
    // enough blocks to even out its last round.  nseg = 1: the whole read (seg_tiles = all tiles, ohseg = ohlen).
    int nseg, seg_tiles, ohseg;
    // four-reads kernel only: `zero_n` 64-bit counters at `zero_ptr` set to zero by the launch's first block (the per-PWM hit counts of the call: the
    // consumers that add to them run behind this kernel in stream order; a hipMemsetAsync in front of it cost ~16 us of stream time, 3.6 of them the fill)
    unsigned long long* zero_ptr;
    int zero_n;
};
struct CandArgs {
    const uint4* afrag;       // [tiles][T][64] PWM fragments (A operand)
    const float* cinit;       // [tiles][2][16] eps per PWM in accumulator order
    const uint8_t* codes;
    uint32_t* cells;          // [(batch, l, n-in-batch, chunk)] x 4 words, bit i of a cell = PWM 128*chunk + i
    uint16_t* centries;       // compact form (nullptr: off): one 16-bit entry per half cell, see scan_mfma.hip "compact entries"
    const uint4* afrag2;      // the other strand's bank in the same launch (nullptr: one strand), with its own cells2 / centries2
    uint32_t* cells2;
    uint16_t* centries2;
    CandDims d;
    int lenp, ntiles;         // padded PWM length (multiple of 4); tiles of 32 PWMs (multiple of 4: whole chunks)
    int uniform_eps;          // afrag holds the bank scaled so that the slack is 4.0 for every PWM (cinit unused)
};
int cand_tile_group(int lenp);
bool cand_compact_ok(const CandArgs& a);
bool cand_two_strands_ok(const CandArgs& a);   // ... and take both strands' banks in one launch   // this launch can write compact entries (four-reads-per-wave kernel, tile groups of 4)
hipError_t launch_cand(const CandArgs& a, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int stage_row_reads(int nch);                                              // reads per row of cells
bool stage_big_lds_ok(int K, int KP, int tabk_stride, bool hist);          // the table fits one 16-wave block's LDS (and not an 8-wave block's 64 KB)
int dense_row_reads(int nch);                                              // the same for the dense tensor (mode 2)
// candidates -> row counts (mode 0), + staged hits (1), or a17's dense tensor, zeros included (2); + histogram
// (b != nullptr: the other strand of gpu_scan in the same launches - same bank shape and geometry, its own buffers)
hipError_t launch_stage_hits(const FillArgs& a, int mode, hipStream_t st, const FillArgs* b = nullptr);
hipError_t launch_row_scan(const FillArgs& a, hipStream_t st, const FillArgs* b = nullptr);             // row counts -> offsets; *total = *base_in + hits
hipError_t launch_emit_records(const FillArgs& a, hipStream_t st, const FillArgs* b = nullptr);         // staged hits -> records
// chunk groups: the largest group (chunks of 128 PWMs: 4, 2, 1) whose slice of the re-scoring table fits a block's LDS beside
// the queues, 0 when the whole table fits (or the bank cannot take the mode); `want` > 0 forces a size (tests, A/B runs)
int stage_cg_chunks(int K, int nch, int lenp, int tabk_stride, int want);
size_t stage_cg_lds_bytes(int cgc, int tabk_stride);
hipError_t launch_fill_scan(const FillArgs& a, hipStream_t st);            // exclusive scan of row_sum

// ---- a17's dense tensor in one kernel (scan_dense.hip)
struct DenseFusedArgs {
    const uint4* afrag;       // [tiles][T][64] PWM fragments of a bank scaled to one slack (pack_mfma, uniform_eps)
    const uint16_t* tabk;     // [K][tabk_stride] binary16 re-scoring table
    const int32_t* lim;       // last valid start per PWM
    const uint8_t* codes;
    uint16_t* out;            // (K, N, ld_l) tensor
    int64_t N;
    int L, pitch, Lout, K, lim_min, ntiles, tabk_stride;
    int ohlen, opitch, cpitch;   // derived by dense_fused_plan
    int l_per_block;             // starts per block (set by the launch: grid.y splits the starts)
    int nwaves;                  // waves per block (8, or 4 when the LDS is short: dense_fused_plan)
};
bool dense_fused_plan(DenseFusedArgs& a, int lenp, int uniform_eps);
hipError_t launch_dense_fused(const DenseFusedArgs& a, int lenp, hipStream_t st);

int scan_len_padded(int maxlen);
// positions per mask row for windows 0..Lout-1 of padded length lenp
int scan_lout_padded(int Lout, int lenp);
hipError_t launch_scan(int mode, int len_padded, const ScanArgs& a, hipStream_t st);
hipError_t launch_fill_sums(const FillArgs& a, hipStream_t st);
hipError_t launch_fill_records(const FillArgs& a, hipStream_t st);
hipError_t launch_mask_histogram(const FillArgs& a, hipStream_t st);
hipError_t launch_encode(int kind, const void* x, int64_t N, int L, int pitch, uint8_t* codes, int32_t* bad,
                         hipStream_t st);

}  // namespace motifs
