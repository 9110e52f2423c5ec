// scan_kernels.h — argument blocks and launchers of the PWM-scan kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace motifs {

constexpr int SCAN_BLOCK = 256;           // 4 waves
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;
constexpr int OFFS_TILE = 64;             // sequences per offset tile

enum { MODE_DENSE = 0, MODE_COUNT = 1, MODE_FILL = 2 };

struct HitRec {
    uint32_t m, n, l;
};

// scalars of one scan launch (passed by value; pointers travel as separate
// __restrict__ kernel parameters so the sequence rows can use scalar loads)
struct ScanDims {
    int64_t N;
    int L, pitch;
    int K, KP, nch;           // PWMs, padded pairs (nch*64), chunks of 64 pairs
    int Lout, LoutP;          // L - minlen + 1; rounded up to 64
    int lim_min;              // L - maxlen: starts <= lim_min are valid for every PWM
    int spw;                  // sequences per wave
    int k_even;
    int batch;                // ordering batch (5000)
    int64_t n0;               // global index offset for records
};

struct ScanArgs {
    const uint32_t* tab;      // [(ind*4 + b) * KP + kp] half2 {pwm 2kp, pwm 2kp+1}
    const int32_t* lim;       // [2*KP] last valid 0-based start per PWM (L - len), -1 if absent
    const uint8_t* codes;     // N rows of `pitch` bytes
    uint16_t* scores;         // DENSE: (K, N, ld_l) col-major
    uint16_t* cnt;            // COUNT: [(n*nch + ch) * LoutP + l]
    const uint32_t* off;      // FILL: same shape
    const int64_t* batch_base;
    HitRec* hits;
    uint16_t* hit_scores;
    int64_t* pwm_counts;      // optional [2*KP]
    ScanDims d;
};

struct OffsArgs {
    const uint16_t* cnt;
    uint32_t* off;
    uint32_t* tilesum;        // [(b*Lout + l) * tiles + t]
    int64_t* batch_base;      // [nbatch]
    int64_t* total;
    int32_t* overflow;
    int64_t N;
    int64_t base0;            // records already emitted before this super-batch
    int Lout, LoutP, nch, batch, tiles, nbatch;
};

int scan_len_padded(int maxlen);
hipError_t launch_scan(int mode, int len_padded, const ScanArgs& a, hipStream_t st);
hipError_t launch_offsets(const OffsArgs& a, hipStream_t st);
hipError_t launch_encode(int kind, const void* x, int64_t N, int L, int pitch, uint8_t* codes, int32_t* bad,
                         hipStream_t st);

}  // namespace motifs
