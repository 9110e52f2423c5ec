// cdl_engine.hip — primitives of the sparse-coding engine (see cdl_engine.h).
// Each forward primitive is a HIP kernel; its VJP is recorded on the tape.
#include "cdl_engine.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace motifs {

// The four GEMMs of a step (syntax-layer analysis, the D-layer's tall / row forms, their filter gradients) run on the binary16 matrix
// instruction with three products per term once a launch has enough tiles; two switches, read once per process, move that choice for tests
// and A/B runs: MOTIFS_GEMM_F32 (set: the float32 matrix instruction for every launch) and MOTIFS_GEMM_F16_MIN=<n> (the bar of all four, in
// jobs / tiles; 1: every launch takes the binary16 forms; defaults 256 / 768 / 768 / 512).
static bool gemm_f32_only() {
    static const bool v = getenv("MOTIFS_GEMM_F32") != nullptr;
    return v;
}
static long gemm_f16_min(long dflt) {
    static const char* s = getenv("MOTIFS_GEMM_F16_MIN");
    return s ? atol(s) : dflt;
}


static inline unsigned nblocks(size_t n, int per = 256, size_t cap = 256 * 32) {
    size_t b = (n + per - 1) / per;
    if (b < 1) b = 1;
    return (unsigned)std::min(b, cap);
}

// Fills and device-to-device copies of the step as plain kernels (no hipMemsetAsync / hipMemcpyAsync): one kind of node for
// the stream and for a captured step alike.  All buffers of the engine are float arrays.
__global__ __launch_bounds__(256) void k_zero4(uint4* __restrict__ p, size_t n16, float* __restrict__ tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.0f;
}
__global__ __launch_bounds__(256) void k_copy1(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
void dev_zero(hipStream_t st, float* p, size_t n) {
    if (!n) return;
    size_t head = (((uintptr_t)p + 15) & ~(uintptr_t)15) - (uintptr_t)p;      // bytes up to 16-byte alignment
    head = std::min(head / 4, n);
    if (head) hipLaunchKernelGGL(k_zero4, dim3(1), dim3(256), 0, st, (uint4*)nullptr, (size_t)0, p, (int)head);
    const size_t rest = n - head, n16 = rest / 4;
    hipLaunchKernelGGL(k_zero4, dim3(nblocks(n16, 256 * 4, 2048)), dim3(256), 0, st, (uint4*)(p + head), n16, p + head + n16 * 4, (int)(rest & 3));
}
void dev_copy(hipStream_t st, float* dst, const float* src, size_t n) {
    if (n) hipLaunchKernelGGL(k_copy1, dim3(nblocks(n, 256 * 4, 2048)), dim3(256), 0, st, src, dst, n);
}

// ---------------------------------------------------------------------------------------------
// engine bookkeeping
// ---------------------------------------------------------------------------------------------
void Engine::reset() {
    for (TNode* t : nodes) delete t;
    nodes.clear();
    tape.clear();
    named.clear();
    arena.reset();
    derived.clear();
    absmax_of.clear();
    zpool = nullptr;
    zleft = 0;
    failed = false;
}

Tensor Engine::make(size_t n, bool needs_grad) {
    TNode* t = new TNode();
    nodes.push_back(t);
    t->n = n;
    t->needs_grad = needs_grad;
    t->v = arena.alloc(n);
    if (!t->v) failed = true;
    return t;
}

Tensor Engine::wrap(float* v, float* g, size_t n, bool needs_grad) {
    TNode* t = new TNode();
    nodes.push_back(t);
    t->v = v;
    t->g = g;
    t->n = n;
    t->needs_grad = needs_grad;
    return t;
}

// Zeroed scratch.  A backward pass asks for a hundred small zeroed buffers (scalar and filter gradients, selection
// workspaces); each fill is a launch, so they are carved from 64 MB chunks that are zeroed once.
float* Engine::zeros(size_t n) {
    constexpr size_t CHUNK = (size_t)16 << 20, SMALL = (size_t)2 << 20;   // floats
    if (n > SMALL) {
        float* p = arena.alloc(n);
        if (!p) {
            failed = true;
            return nullptr;
        }
        dev_zero(st, p, n);
        return p;
    }
    const size_t need = (n + 63) & ~(size_t)63;    // keep 256-byte alignment
    if (need > zleft) {
        zpool = arena.alloc(CHUNK);
        if (!zpool) {
            failed = true;
            zleft = 0;
            return nullptr;
        }
        dev_zero(st, zpool, CHUNK);
        zleft = CHUNK;
    }
    float* p = zpool;
    zpool += need;
    zleft -= need;
    return p;
}

float* Engine::grad(Tensor t) {
    if (!t->g) t->g = zeros(t->n);
    return t->g;
}

// a bank pointer seen with another re-layout kind or other dimensions is another entry
float* Engine::relayout(const float* src, int kind, int d0, int d1, int d2, size_t n, bool& fresh) {
    const RelayoutKey key{(const void*)src, kind, d0, d1, d2, n};
    auto it = derived.find(key);
    fresh = it == derived.end();
    if (!fresh) return it->second;
    float* p = arena.alloc(n);
    if (!p) {
        failed = true;
        return nullptr;
    }
    derived[key] = p;
    return p;
}

// The gradient buffer of t for a kernel that can either overwrite or accumulate: acc = 0 on first use (no zero fill).
float* Engine::grad_first(Tensor t, int& acc) {
    acc = 1;
    if (t->g) return t->g;
    t->g = arena.alloc(t->n);
    if (!t->g) {
        failed = true;
        return nullptr;
    }
    acc = 0;
    return t->g;
}

void Engine::backward() {
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) {
        if (failed) break;
        (*it)();
    }
    tape.clear();
}

#include "cdl_elementwise.inc"
#include "cdl_toeplitz_tiles.inc"
#include "cdl_syntax_gemm.inc"
#include "cdl_dlayer_gemm.inc"
#include "cdl_layouts.inc"
#include "cdl_sparse_syntax.inc"
#include "cdl_selections.inc"

}  // namespace motifs
