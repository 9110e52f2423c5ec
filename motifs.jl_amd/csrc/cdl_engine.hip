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

// ---------------------------------------------------------------------------------------------
// elementwise kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_lin(const float* x, float a, const float* y, float b, float cst, size_t n, size_t yn, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i % yn] : 0.0f) + cst;
}
// The backward kernels below take `acc`: 1 adds into dx, 0 overwrites it (the first contribution to a gradient
// buffer, which then needs no zero fill: Engine::grad_first).
__global__ void k_axpy(const float* go, float a, size_t n, float* dx, int acc) {   // dx (+)= a * go
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (acc ? dx[i] : 0.0f) + a * go[i];
}
__global__ void k_mul(const float* x, const float* y, size_t n, size_t yn, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = x[i] * y[i % yn];
}
__global__ void k_mul_bwd_x(const float* go, const float* y, size_t n, size_t yn, float* dx, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (acc ? dx[i] : 0.0f) + go[i] * y[i % yn];
}
// dy[j] += coef * sum_{i == j mod yn} go[i] * (x ? x[i] : 1): yn == 1 -> block reduction; else one thread per j
__global__ void k_bcast_reduce_all(const float* go, const float* x, size_t n, float coef, float* dy) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += (double)go[i] * (x ? (double)x[i] : 1.0);
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
        atomicAdd(dy, (float)(coef * t));
    }
}
__global__ void k_bcast_reduce_mod(const float* go, const float* x, size_t n, size_t yn, float coef, float* dy) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= yn) return;
    double acc = 0.0;
    for (size_t i = j; i < n; i += yn) acc += (double)go[i] * (x ? (double)x[i] : 1.0);
    dy[j] += (float)(coef * acc);
}
static void bcast_reduce(hipStream_t st, const float* go, const float* x, size_t n, size_t yn, float coef, float* dy) {
    if (yn == n) {
        if (x) hipLaunchKernelGGL(k_mul_bwd_x, dim3(nblocks(n)), dim3(256), 0, st, go, x, n, n, dy, 1);  // coef == 1 there
        else hipLaunchKernelGGL(k_axpy, dim3(nblocks(n)), dim3(256), 0, st, go, coef, n, dy, 1);
    } else if (yn == 1) {
        hipLaunchKernelGGL(k_bcast_reduce_all, dim3(nblocks(n, 256, 1024)), dim3(256), 0, st, go, x, n, coef, dy);
    } else {
        hipLaunchKernelGGL(k_bcast_reduce_mod, dim3((unsigned)((yn + 255) / 256)), dim3(256), 0, st, go, x, n, yn, coef, dy);
    }
}
__global__ void k_relu(const float* x, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = x[i] > 0.0f ? x[i] : 0.0f;
}
__global__ void k_relu_bwd(const float* go, const float* x, size_t n, float* dx, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (acc ? dx[i] : 0.0f) + (x[i] > 0.0f ? go[i] : 0.0f);
}
__global__ void k_maskmul(const float* x, const float* m, float c, size_t n, float* out, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = c * m[i] * x[i];
        out[i] = acc ? out[i] + v : v;
    }
}
__global__ void k_exp(const float* x, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = expf(x[i]);
}
// prep_filters (model.jl:139-146) in one pass: norm4(x .* x + eps) (a mul, a lin and a norm4 before)
__global__ void k_norm4sq(const float* x, float eps, size_t n4, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i];
        const float4 t = make_float4(1.0f * (v.x * v.x) + 0.0f + eps, 1.0f * (v.y * v.y) + 0.0f + eps, 1.0f * (v.z * v.z) + 0.0f + eps, 1.0f * (v.w * v.w) + 0.0f + eps);
        const float s = t.x + t.y + t.z + t.w;
        ((float4*)out)[i] = make_float4(t.x / s, t.y / s, t.z / s, t.w / s);
    }
}
__global__ void k_norm4sq_bwd(const float* go, const float* x, const float* out, float eps, size_t n4, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i], o = ((const float4*)out)[i], g = ((const float4*)go)[i];
        const float s = (1.0f * (v.x * v.x) + 0.0f + eps) + (1.0f * (v.y * v.y) + 0.0f + eps) + (1.0f * (v.z * v.z) + 0.0f + eps) + (1.0f * (v.w * v.w) + 0.0f + eps);
        const float dot = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
        const float4 dq = make_float4((g.x - dot) / s, (g.y - dot) / s, (g.z - dot) / s, (g.w - dot) / s);
        float4 d = ((float4*)dx)[i];
        d.x += dq.x * v.x + dq.x * v.x, d.y += dq.y * v.y + dq.y * v.y, d.z += dq.z * v.z + dq.z * v.z, d.w += dq.w * v.w + dq.w * v.w;
        ((float4*)dx)[i] = d;
    }
}
__global__ void k_norm4(const float* x, size_t n4, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i];
        const float s = v.x + v.y + v.z + v.w;
        ((float4*)out)[i] = make_float4(v.x / s, v.y / s, v.z / s, v.w / s);
    }
}
__global__ void k_norm4_bwd(const float* go, const float* x, const float* out, size_t n4, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i], o = ((const float4*)out)[i], g = ((const float4*)go)[i];
        const float s = v.x + v.y + v.z + v.w;
        const float dot = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
        float4 d = ((float4*)dx)[i];
        d.x += (g.x - dot) / s;
        d.y += (g.y - dot) / s;
        d.z += (g.z - dot) / s;
        d.w += (g.w - dot) / s;
        ((float4*)dx)[i] = d;
    }
}
// one block per segment
// SQ: the segment is x .* x (prep_syntax_filters, model.jl:148-151: the square was a launch of its own)
template <bool SQ>
__global__ void k_norml2(const float* x, int seg, float* out, float* nrm_out) {
    const float* xs = x + (size_t)blockIdx.x * seg;
    double acc = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const float t = SQ ? xs[i] * xs[i] : xs[i];
        acc += (double)t * t;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    __shared__ float nrm;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
        nrm = (float)sqrt(t);
        nrm_out[blockIdx.x] = nrm;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < seg; i += blockDim.x) out[(size_t)blockIdx.x * seg + i] = (SQ ? xs[i] * xs[i] : xs[i]) / nrm;
}
template <bool SQ>
__global__ void k_norml2_bwd(const float* go, const float* out, const float* nrm_in, int seg, float* dx, const float* x) {
    const size_t base = (size_t)blockIdx.x * seg;
    double acc = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) acc += (double)go[base + i] * out[base + i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    __shared__ float dot;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
        dot = (float)t;
    }
    __syncthreads();
    const float nrm = nrm_in[blockIdx.x];
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const float dt = (go[base + i] - out[base + i] * dot) / nrm;
        dx[base + i] += SQ ? dt * x[base + i] + dt * x[base + i] : dt;      // through x .* x: once per factor
    }
}
// out[grp] = coef * sum of squares of the group's slice; one block per (group, chunk) + atomics
__global__ void k_sumsq_groups(const float* x, size_t per_group, float coef, float* out) {
    const int g = blockIdx.y;
    const float* xs = x + (size_t)g * per_group;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_group; i += (size_t)gridDim.x * blockDim.x)
        acc += (double)xs[i] * xs[i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
        atomicAdd(&out[g], (float)(coef * t));
    }
}
__global__ void k_sumsq_groups_bwd(const float* gout, const float* x, size_t per_group, size_t n, float coef2, float* dx, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (acc ? dx[i] : 0.0f) + coef2 * x[i] * gout[i / per_group];
}

// coef * sum over a group of (x + b*[y >= thr]*y)^2: the syntax-layer term of the loss (model.jl:321-323) without the
// residual ever being written; the VJP recomputes it.
__global__ void k_resid_sumsq(const float* x, const float* y, const float* thr, float b, size_t per_group, float coef, float* out) {
    const int g = blockIdx.y;
    const size_t base = (size_t)g * per_group;
    const float t = thr[g];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_group; i += (size_t)gridDim.x * blockDim.x) {
        const float yv = y[base + i];
        const float r = 1.0f * x[base + i] + b * (!(yv >= t) ? 0.0f : yv);
        acc += (double)r * r;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        atomicAdd(&out[g], (float)(coef * s));
    }
}
__global__ void k_resid_sumsq_bwd(const float* gout, const float* x, const float* y, const float* thr, float b, size_t per_group, float coef2,
                                  float* dx, int ax, float* dy, int ay) {
    const int g = blockIdx.y;
    const size_t base = (size_t)g * per_group;
    const float t = thr[g], w = coef2 * gout[g];
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per_group; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float yv = y[i];
        const bool on = yv >= t;
        const float r = 1.0f * x[i] + b * (on ? yv : 0.0f);
        const float gr = w * r;
        if (dx) dx[i] = (ax ? dx[i] : 0.0f) + gr;
        if (dy) dy[i] = (ay ? dy[i] : 0.0f) + b * (on ? gr : 0.0f);
    }
}

// a*x + b*y + c*z in one pass (z optional), one grid row per group of `per` elements.
// ythr (optional): cat_ZY's median mask folded in as a threshold per group, y counts where y >= ythr[group]
// (a constant in the backward, @ignore model.jl:208) - no 0/1 mask is ever written or read.
constexpr size_t AMAX_MIN_N = (size_t)8 << 20;    // floats: below this (~18 mini-batches) a pass of k_absmax per image is cheaper than keeping the maximum here
// amax (optional, pre-zeroed): the bits of the largest |out| - the image this forms is the syntax-layer GEMM's operand, whose binary16
// form (k_ana_f16x3) is scaled by it; one atomic per block, and only when it can raise the maximum
__global__ void k_lin3(const float* x, float a, const float* y, const float* ythr, float b, const float* z, float c, size_t per,
                       float* out, uint32_t* amax) {
    const size_t base = (size_t)blockIdx.y * per;
    const float t = ythr ? ythr[blockIdx.y] : 0.0f;
    // The image's largest magnitude (for the binary16 GEMM's scale) without a pass of its own: a wave leaves its maximum in LDS, the LAST wave of
    // the block to arrive (an LDS counter, no barrier at the end) carries the block's to memory - and only if it beats what the maximum stood
    // at when the block began (asked for at the start: a load at the END of every wave was a trip to L2 that nothing hid, +12-18 us per
    // launch at 64 mini-batches).
    __shared__ uint32_t bmax, bcnt;
    uint32_t seen = 0;
    if (amax) {
        if (threadIdx.x == 0) bmax = 0, bcnt = 0;
        seen = __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    uint32_t m = 0;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float yv = y[i];
        const float o = a * x[i] + b * ((ythr && !(yv >= t)) ? 0.0f : yv) + (z ? c * z[i] : 0.0f);
        out[i] = o;
        m = max(m, __float_as_uint(o) & 0x7fffffffu);
    }
    if (amax) {
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&bmax, m);
            if (atomicAdd(&bcnt, 1u) == (blockDim.x >> 6) - 1) {       // (a wave's LDS operations complete in order: every maximum is in)
                const uint32_t mb = atomicMax(&bmax, 0u);
                if (mb > seen && mb > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, mb);
            }
        }
    }
}
// VJP of k_lin3: d{x,y,z} (+)= {a, b*[y >= thr], c} * go, go read once
__global__ void k_lin3_bwd(const float* go, size_t per, float a, float* dx, int ax, float b, const float* y, const float* ythr, float* dy,
                           int ay, float c, float* dz, int az) {
    const size_t base = (size_t)blockIdx.y * per;
    const float t = ythr ? ythr[blockIdx.y] : 0.0f;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float g = go[i];
        if (dx) dx[i] = (ax ? dx[i] : 0.0f) + a * g;
        if (dy) dy[i] = (ay ? dy[i] : 0.0f) + b * ((ythr && !(y[i] >= t)) ? 0.0f : g);
        if (dz) dz[i] = (az ? dz[i] : 0.0f) + c * g;
    }
}
// c * [x >= thr[group]] .* x and its VJP (src = x forward, src = d out backward; sel = x both times)
__global__ void k_thrmul(const float* src, const float* sel, const float* thr, float c, size_t per, float* out, int acc) {
    const size_t base = (size_t)blockIdx.y * per;
    const float t = thr[blockIdx.y];
    if ((per & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)sel) | ((uintptr_t)out)) & 15) == 0) {       // 16-byte accesses
        for (size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; j < per; j += (size_t)gridDim.x * blockDim.x * 4) {
            const size_t i = base + j;
            const float4 s4 = *(const float4*)(sel + i), x4 = *(const float4*)(src + i);
            float4 o = acc ? *(const float4*)(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float v0 = s4.x >= t ? c * x4.x : 0.0f, v1 = s4.y >= t ? c * x4.y : 0.0f, v2 = s4.z >= t ? c * x4.z : 0.0f, v3 = s4.w >= t ? c * x4.w : 0.0f;
            o.x = acc ? o.x + v0 : v0, o.y = acc ? o.y + v1 : v1, o.z = acc ? o.z + v2 : v2, o.w = acc ? o.w + v3 : v3;
            *(float4*)(out + i) = o;
        }
        return;
    }
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float v = sel[i] >= t ? c * src[i] : 0.0f;
        out[i] = acc ? out[i] + v : v;
    }
}
// The ISTA step of update_ZY (model.jl:240-244) on the compact image, fused:
//   out = relu(ZY - lst * (g1 + pen * (ZY - FX - ab)) - ls * lst)          (ab optional; pen, lst, ls device scalars)
// The first pass of the median select over the new codes (the top 11 bits of the positive entries, create_ZY_mask
// model.jl:194-204) can ride in the kernel that writes them: per thread, runs of equal digits are counted in registers
// and leave as one LDS atomic; the block's histogram is added to hist0[group][ZH_BINS] at the end.
constexpr int ZH_BINS = 2048, ZH_SHIFT = 21;
struct RunHist {
    uint32_t d = 0xffffffffu, c = 0;
    __device__ __forceinline__ void take(float v, uint32_t* h) {
        if (!(v > 0.0f)) return;
        const uint32_t b = __float_as_uint(v) >> ZH_SHIFT;
        if (b == d) {
            c++;
        } else {
            if (c) atomicAdd(&h[d], c);
            d = b, c = 1;
        }
    }
    __device__ __forceinline__ void flush(uint32_t* h) {
        if (c) atomicAdd(&h[d], c);
    }
};
static __device__ __forceinline__ void zh_begin(uint32_t* zh) {
    for (int i = threadIdx.x; i < ZH_BINS; i += blockDim.x) zh[i] = 0;
    __syncthreads();
}
static __device__ __forceinline__ void zh_end(uint32_t* zh, uint32_t* hist0) {
    __syncthreads();
    uint32_t* hg = hist0 + (size_t)blockIdx.y * 2 * ZH_BINS;
    for (int i = threadIdx.x; i < ZH_BINS; i += blockDim.x) {
        const uint32_t c = zh[i];
        if (c) atomicAdd(&hg[i], c);
    }
}
// (one grid row per group of `per` elements; hist0 optional)
// Blocks per group: every block ends with one atomic per occupied bin on its group's histogram - a few dozen addresses -
// so with the 1772 blocks a single mini-batch used to get, the step took 23 us for 1.8 MB; 256 blocks per group at most.
static size_t zy_blocks_cap(int G, bool hist) { return std::max<size_t>(std::min<size_t>(256 * 32 / G, hist ? 256 : 256 * 32), 1); }
__global__ void k_zy_step(const float* ZY, const float* g1, const float* FX, const float* ab, const float* pen, const float* lst,
                          const float* ls, size_t per, float* out, uint32_t* hist0) {
    __shared__ uint32_t zh[ZH_BINS];
    const float p = *pen, s = *lst, l = *ls;
    const size_t base = (size_t)blockIdx.y * per;
    RunHist rh;
    if (hist0) zh_begin(zh);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float inner = ZY[i] - (FX[i] + (ab ? ab[i] : 0.0f));
        const float grad = g1[i] + inner * p;
        const float u = (ZY[i] - grad * s) - l * s;
        const float o = u > 0.0f ? u : 0.0f;
        out[i] = o;
        if (hist0) rh.take(o, zh);
    }
    if (hist0) {
        rh.flush(zh);
        zh_end(zh, hist0);
    }
}
// its VJP: one pass over the image for the four tensor gradients and the three scalar gradients
// (dsc[0..2] = d pen, d lst, d ls; block sums in double, one float atomic each per block)
template <int V>   // V = 4: 16-byte accesses (n % 4 == 0, 16-byte aligned tensors); V = 1: scalar
__global__ void k_zy_step_bwd(const float* go, const float* out, const float* ZY, const float* g1, const float* FX, const float* ab,
                              const float* pen, const float* lst, const float* ls, size_t n, float* dZY, int aZY, float* dg1, int ag1,
                              float* dFX, int aFX, float* dab, int aab, float* dpen, float* dlst, float* dls) {
    struct VF {
        float e[V];
    };
    const float p = *pen, s = *lst, l = *ls;
    double sp = 0, ss = 0, sl = 0;
    auto ld = [&](const float* q, size_t i) {
        VF r;
        if (V == 4) {
            const float4 x = *(const float4*)(q + i);
            r.e[0] = x.x, r.e[1 % V] = x.y, r.e[2 % V] = x.z, r.e[3 % V] = x.w;
        } else {
            r.e[0] = q[i];
        }
        return r;
    };
    auto stv = [&](float* q, size_t i, const VF& r) {
        if (V == 4) *(float4*)(q + i) = make_float4(r.e[0], r.e[1 % V], r.e[2 % V], r.e[3 % V]);
        else q[i] = r.e[0];
    };
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
        VF z{};
        const VF vo = ld(out, i), vgo = ld(go, i), vzy = ld(ZY, i), vfx = ld(FX, i), vab = ab ? ld(ab, i) : z, vg1 = ld(g1, i);
        VF oZY = (dZY && aZY) ? ld(dZY, i) : z, og1 = (dg1 && ag1) ? ld(dg1, i) : z, oFX = (dFX && aFX) ? ld(dFX, i) : z, oab = (dab && aab) ? ld(dab, i) : z;
#pragma unroll
        for (int u = 0; u < V; u++) {
            const float du = vo.e[u] > 0.0f ? vgo.e[u] : 0.0f;
            const float inner = vzy.e[u] - (vfx.e[u] + vab.e[u]);
            const float grad = vg1.e[u] + inner * p;
            oZY.e[u] = oZY.e[u] + du * (1.0f - s * p);
            og1.e[u] = og1.e[u] - s * du;
            const float dfx = s * p * du;
            oFX.e[u] = oFX.e[u] + dfx;
            oab.e[u] = oab.e[u] + dfx;
            sp -= (double)du * (double)(s * inner);
            ss -= (double)du * (double)(grad + l);
            sl -= (double)du * (double)s;
        }
        if (dZY) stv(dZY, i, oZY);
        if (dg1) stv(dg1, i, og1);
        if (dFX) stv(dFX, i, oFX);
        if (dab) stv(dab, i, oab);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sp += __shfl_xor(sp, d);
        ss += __shfl_xor(ss, d);
        sl += __shfl_xor(sl, d);
    }
    __shared__ double red[3][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = sp;
        red[1][wv] = ss;
        red[2][wv] = sl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (dpen) atomicAdd(dpen, (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        if (dlst) atomicAdd(dlst, (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
        if (dls) atomicAdd(dls, (float)(red[2][0] + red[2][1] + red[2][2] + red[2][3]));
    }
}

// The gradient step of update_X (model.jl:252-253) before the projection: out = X - ost * xg (ost a device scalar),
// and its VJP in one pass: dX (+)= go, dxg (+)= -ost * go, d ost += -sum(go * xg).
__global__ void k_x_step(const float* X, const float* xg, const float* ost, size_t n, float* out) {
    const float o = *ost;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = X[i] - xg[i] * o;
}
__global__ void k_x_step_bwd(const float* go, const float* xg, const float* ost, size_t n, float* dX, int aX, float* dxg, int axg,
                             float* dost) {
    const float o = *ost;
    double so = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = go[i];
        if (dX) dX[i] = (aX ? dX[i] : 0.0f) + g;
        if (dxg) dxg[i] = (axg ? dxg[i] : 0.0f) - o * g;
        so -= (double)g * (double)xg[i];
    }
    for (int d = 32; d >= 1; d >>= 1) so += __shfl_xor(so, d);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = so;
    __syncthreads();
    if (threadIdx.x == 0 && dost) atomicAdd(dost, (float)(red[0] + red[1] + red[2] + red[3]));
}

// The proximal step of update_F (model.jl:298-303) before the normalisation, with the arithmetic of the separate
// passes it replaces: out = relu((Fc - sg * Fgrad * kst) - kst * ks); Fc may be one bank shared by all groups
// (n_c < n); sg = +-1 (the gradient may arrive negated), Fgrad may be absent (identically zero).
// VJP in one pass: dt = go * [out > 0]; dFgrad (+)= -sg kst dt; dFc (+)= dt (same size) or dt is written out for the
// reduction over groups; d kst += -sum(dt (sg Fgrad + ks)); d ks += -kst sum(dt).
// sw (h, 2M, K; h == 0: none): Fgrad is still in the layout its kernel wrote, [g][h][2M][K], and is read through swap02's index
// map (element (k, j, i) of the step is element (i, j, k) of Fgrad) - for banks of few mini-batches, where the swap was a launch
struct SwapDims {
    int h, n2, k;
};
static __device__ __forceinline__ size_t fgrad_at(size_t i, const SwapDims& sw) {
    if (!sw.h) return i;
    const size_t per = (size_t)sw.h * sw.n2 * sw.k, gg = i / per, r = i - gg * per;
    const int ii = (int)(r % sw.h), j = (int)((r / sw.h) % sw.n2), k = (int)(r / ((size_t)sw.h * sw.n2));
    return gg * per + ((size_t)ii * sw.n2 + j) * sw.k + k;
}
__global__ void k_f_step(const float* Fc, size_t nc, const float* Fgrad, float sg, const float* kst, const float* ks, size_t n, float* out, SwapDims sw) {
    const float a = *kst, m2 = *kst * *ks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m1 = Fgrad ? (sg * Fgrad[fgrad_at(i, sw)]) * a : 0.0f;
        const float t3 = (Fc[nc == n ? i : i % nc] - m1) - m2;
        out[i] = t3 > 0.0f ? t3 : 0.0f;
    }
}
__global__ void k_f_step_bwd(const float* go, const float* out, const float* Fgrad, float sg, const float* kst, const float* ks, size_t n,
                             float* dFc, int aFc, float* dFg, int aFg, float* dt_out, float* dkst, float* dks, SwapDims sw) {
    const float a = *kst, b = *ks;
    double sk = 0, ss = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float dt = out[i] > 0.0f ? go[i] : 0.0f;
        const size_t gi = fgrad_at(i, sw);
        if (dFc) dFc[i] = (aFc ? dFc[i] : 0.0f) + dt;
        if (dt_out) dt_out[i] = dt;
        if (dFg) dFg[gi] = (aFg ? dFg[gi] : 0.0f) - (sg * a) * dt;
        sk -= (double)dt * (double)((Fgrad ? sg * Fgrad[gi] : 0.0f) + b);
        ss -= (double)dt * (double)a;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sk += __shfl_xor(sk, d);
        ss += __shfl_xor(ss, d);
    }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sk;
        red[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (dkst) atomicAdd(dkst, (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        if (dks) atomicAdd(dks, (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}

// The same step with the dual update of the previous pass folded in (model.jl:263-266 then :240-244): the scaled
// duals [alpha beta] advance by FX - ZY on entry,
//   abn = FX - ZY + abp                         (abp optional: zero duals, :338)
//   out = relu(ZY - lst * (g1 + pen * (ZY - FX - abn)) - ls * lst)
// which saves the separate three-term pass (and its VJP) per ADMM pass.
__global__ void k_zy_step2(const float* ZY, const float* g1, const float* FX, const float* abp, const float* pen, const float* lst,
                           const float* ls, size_t per, float* out, float* abn, uint32_t* hist0) {
    __shared__ uint32_t zh[ZH_BINS];
    const float p = *pen, s = *lst, l = *ls;
    const size_t base = (size_t)blockIdx.y * per;
    RunHist rh;
    if (hist0) zh_begin(zh);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + j;
        const float zy = ZY[i], fx = FX[i];
        const float dual = (fx - zy) + (abp ? abp[i] : 0.0f);
        const float inner = zy - (fx + dual);
        const float grad = g1[i] + inner * p;
        const float u = (zy - grad * s) - l * s;
        const float o = u > 0.0f ? u : 0.0f;
        abn[i] = dual;
        out[i] = o;
        if (hist0) rh.take(o, zh);
    }
    if (hist0) {
        rh.flush(zh);
        zh_end(zh, hist0);
    }
}
// VJP: go = d out (may be null), gab = d abn (may be null).  inner = 2 ZY - 2 FX - abp.
// g3 (optional) = d of the combination img = FX + b3*[out >= thr]*out + abn formed after the step (lin3_zy): its
// three contributions (to d out, d abn and d FX) are folded in here instead of a pass of their own.
template <int V>   // V = 4: 16-byte accesses (per % 4 == 0, 16-byte aligned tensors); V = 8: two of them per lane (per % 8 == 0); V = 1: scalar
__global__ void k_zy_step2_bwd(const float* go, const float* gab, const float* g3, float b3, const float* thr, const float* out,
                               const float* ZY, const float* g1, const float* FX, const float* abp, const float* pen, const float* lst,
                               const float* ls, size_t per, float* dZY, int aZY, float* dg1, int ag1, float* dFX, int aFX, float* dabp,
                               int aabp, float* dpen, float* dlst, float* dls) {
    struct VF {
        float e[V];
    };
    const float p = *pen, s = *lst, l = *ls;
    const size_t base = (size_t)blockIdx.y * per;
    const float t3 = thr ? thr[blockIdx.y] : 0.0f;
    double sp = 0, ss = 0, sl = 0;
    auto ld = [&](const float* q, size_t i) {
        VF r;
        if (V == 8) {                              // two 16-byte pieces 4 KB apart per lane: a wave's two loads of a stream cover 2 x 1 KB
            const float4 x = *(const float4*)(q + i), y = *(const float4*)(q + i + 4);
            r.e[0] = x.x, r.e[1 % V] = x.y, r.e[2 % V] = x.z, r.e[3 % V] = x.w, r.e[4 % V] = y.x, r.e[5 % V] = y.y, r.e[6 % V] = y.z, r.e[7 % V] = y.w;
        } else if (V == 4) {
            const float4 x = *(const float4*)(q + i);
            r.e[0] = x.x, r.e[1 % V] = x.y, r.e[2 % V] = x.z, r.e[3 % V] = x.w;
        } else {
            r.e[0] = q[i];
        }
        return r;
    };
    auto st = [&](float* q, size_t i, const VF& r) {
        if (V == 8) {
            *(float4*)(q + i) = make_float4(r.e[0], r.e[1 % V], r.e[2 % V], r.e[3 % V]);
            *(float4*)(q + i + 4) = make_float4(r.e[4 % V], r.e[5 % V], r.e[6 % V], r.e[7 % V]);
        } else if (V == 4) *(float4*)(q + i) = make_float4(r.e[0], r.e[1 % V], r.e[2 % V], r.e[3 % V]);
        else q[i] = r.e[0];
    };
    for (size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; j < per; j += (size_t)gridDim.x * blockDim.x * V) {
        const size_t i = base + j;
        VF z{};
        const VF vo = ld(out, i), vg3 = g3 ? ld(g3, i) : z, vgo = go ? ld(go, i) : z, vgab = gab ? ld(gab, i) : z;
        const VF vzy = ld(ZY, i), vfx = ld(FX, i), vabp = abp ? ld(abp, i) : z, vg1 = ld(g1, i);
        VF oZY = (dZY && aZY) ? ld(dZY, i) : z, og1 = (dg1 && ag1) ? ld(dg1, i) : z, oFX = (dFX && aFX) ? ld(dFX, i) : z,
           oab = (dabp && aabp) ? ld(dabp, i) : z;
#pragma unroll
        for (int u = 0; u < V; u++) {
            const float o = vo.e[u], g3v = vg3.e[u];
            float gz = vgo.e[u];
            if (g3) gz += b3 * ((thr && !(o >= t3)) ? 0.0f : g3v);
            const float du = o > 0.0f ? gz : 0.0f;
            const float gb = vgab.e[u] + g3v;
            const float zy = vzy.e[u], fx = vfx.e[u];
            const float dual = (fx - zy) + vabp.e[u];
            const float inner = zy - (fx + dual);
            const float grad = vg1.e[u] + inner * p;
            const float t = s * p * du;
            oZY.e[u] = oZY.e[u] + (du - 2.0f * t) - gb;
            og1.e[u] = og1.e[u] - s * du;
            oFX.e[u] = oFX.e[u] + (2.0f * t + gb) + g3v;
            oab.e[u] = oab.e[u] + t + gb;
            sp -= (double)du * (double)(s * inner);
            ss -= (double)du * (double)(grad + l);
            sl -= (double)du * (double)s;
        }
        if (dZY) st(dZY, i, oZY);
        if (dg1) st(dg1, i, og1);
        if (dFX) st(dFX, i, oFX);
        if (dabp) st(dabp, i, oab);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sp += __shfl_xor(sp, d);
        ss += __shfl_xor(ss, d);
        sl += __shfl_xor(sl, d);
    }
    __shared__ double red[3][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = sp;
        red[1][wv] = ss;
        red[2][wv] = sl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (dpen) atomicAdd(dpen, (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        if (dlst) atomicAdd(dlst, (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
        if (dls) atomicAdd(dls, (float)(red[2][0] + red[2][1] + red[2][2] + red[2][3]));
    }
}

#define EW(kern, n, ...) hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(256), 0, st, __VA_ARGS__)

Tensor Engine::lin(Tensor x, float a, Tensor y, float b, float cst) {
    Tensor out = make(x->n, x->needs_grad || (y && y->needs_grad));
    if (failed) return out;
    EW(k_lin, x->n, x->v, a, y ? y->v : nullptr, b, cst, x->n, y ? y->n : (size_t)1, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y, a, b]() {
            if (!out->g) return;
            if (x->needs_grad) {
                int acc;
                float* dx = grad_first(x, acc);
                if (dx) EW(k_axpy, out->n, out->g, a, out->n, dx, acc);
            }
            if (y && y->needs_grad) bcast_reduce(st, out->g, nullptr, out->n, y->n, b, grad(y));
        });
    return out;
}

// relu(a * x + c) in one pass (warmup_ZY's shrinkage, model.jl:176-177: a lin and a relu before), with k_lin's arithmetic
__global__ void k_shrink(const float* x, float a, float c, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float u = a * x[i] + 0.0f + c;
        out[i] = u > 0.0f ? u : 0.0f;
    }
}
__global__ void k_shrink_bwd(const float* go, const float* out, float a, size_t n, float* dx, int acc) {   // dx (+)= a * [out > 0] * go
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = (acc ? dx[i] : 0.0f) + a * (out[i] > 0.0f ? go[i] : 0.0f);
}
Tensor Engine::shrink(Tensor x, float a, float c) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_shrink, x->n, x->v, a, c, x->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, a]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) EW(k_shrink_bwd, out->n, out->g, out->v, a, out->n, dx, acc);
        });
    return out;
}

Tensor Engine::lin3(Tensor x, float a, Tensor y, float b, Tensor z, float c, const float* ythr, int groups) {
    Tensor out = make(x->n, x->needs_grad || y->needs_grad || (z && z->needs_grad));
    if (failed) return out;
    const int G = ythr ? groups : 1;
    const size_t per = x->n / G;
    const dim3 grid(nblocks(per, 256, std::max<size_t>(256 * 32 / G, 1)), G);
    uint32_t* am = out->n >= AMAX_MIN_N ? (uint32_t*)zeros(1) : nullptr;   // (images of small steps never reach the binary16 GEMM)
    if (failed) return out;
    if (am) absmax_of[out->v] = am;
    hipLaunchKernelGGL(k_lin3, grid, dim3(256), 0, st, x->v, a, y->v, ythr, b, z ? z->v : nullptr, c, per, out->v, am);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y, z, a, b, c, ythr, per, grid]() {
            if (!out->g) return;
            if (x == y || x == z || (z && y == z)) {                  // aliased operands: one contribution at a time
                int ax;
                float* dx = x->needs_grad ? grad_first(x, ax) : nullptr;
                if (dx) hipLaunchKernelGGL(k_lin3_bwd, grid, dim3(256), 0, st, out->g, per, a, dx, ax, 0.0f, nullptr, nullptr, nullptr, 1, 0.0f, nullptr, 1);
                int ay;
                float* dy = y->needs_grad ? grad_first(y, ay) : nullptr;
                if (dy) hipLaunchKernelGGL(k_lin3_bwd, grid, dim3(256), 0, st, out->g, per, 0.0f, nullptr, 1, b, y->v, ythr, dy, ay, 0.0f, nullptr, 1);
                int az;
                float* dz = (z && z->needs_grad) ? grad_first(z, az) : nullptr;
                if (dz) hipLaunchKernelGGL(k_lin3_bwd, grid, dim3(256), 0, st, out->g, per, 0.0f, nullptr, 1, 0.0f, nullptr, nullptr, nullptr, 1, c, dz, az);
                return;
            }
            int ax = 1, ay = 1, az = 1;
            float* dx = x->needs_grad ? grad_first(x, ax) : nullptr;
            float* dy = y->needs_grad ? grad_first(y, ay) : nullptr;
            float* dz = (z && z->needs_grad) ? grad_first(z, az) : nullptr;
            if (failed) return;
            hipLaunchKernelGGL(k_lin3_bwd, grid, dim3(256), 0, st, out->g, per, a, dx, ax, b, y->v, ythr, dy, ay, c, dz, az);
        });
    return out;
}

Tensor Engine::thrmul(Tensor x, const float* thr, int groups, float c) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    const size_t per = x->n / groups;
    const dim3 grid(nblocks(per, 256, std::max<size_t>(256 * 32 / groups, 1)), groups);
    hipLaunchKernelGGL(k_thrmul, grid, dim3(256), 0, st, x->v, x->v, thr, c, per, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, thr, c, per, grid]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) hipLaunchKernelGGL(k_thrmul, grid, dim3(256), 0, st, out->g, x->v, thr, c, per, dx, acc);
        });
    return out;
}

Tensor Engine::zy_step(Tensor ZY, Tensor g1, Tensor FX, Tensor ab, Tensor pen, Tensor lst, Tensor ls, uint32_t* hist0, int groups) {
    const bool ng = ZY->needs_grad || g1->needs_grad || FX->needs_grad || (ab && ab->needs_grad) || pen->needs_grad ||
                    lst->needs_grad || ls->needs_grad;
    Tensor out = make(ZY->n, ng);
    if (failed) return out;
    {
        const int G = hist0 ? groups : 1;
        const size_t per = ZY->n / G;
        hipLaunchKernelGGL(k_zy_step, dim3(nblocks(per, 256, zy_blocks_cap(G, hist0 != nullptr)), G), dim3(256), 0, st, ZY->v, g1->v, FX->v,
                           ab ? ab->v : nullptr, pen->v, lst->v, ls->v, per, out->v, hist0);
    }
    if (recording && out->needs_grad)
        tape.push_back([this, out, ZY, g1, FX, ab, pen, lst, ls]() {
            if (!out->g) return;
            int a0 = 1, a1 = 1, a2 = 1, a3 = 1;
            float* d0 = ZY->needs_grad ? grad_first(ZY, a0) : nullptr;
            float* d1 = g1->needs_grad ? grad_first(g1, a1) : nullptr;
            float* d2 = FX->needs_grad ? grad_first(FX, a2) : nullptr;
            float* d3 = (ab && ab->needs_grad) ? grad_first(ab, a3) : nullptr;
            float* dp = pen->needs_grad ? grad(pen) : nullptr;
            float* ds = lst->needs_grad ? grad(lst) : nullptr;
            float* dl = ls->needs_grad ? grad(ls) : nullptr;
            if (failed) return;
            auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
            const bool v4 = (out->n & 3) == 0 && al16(out->g) && al16(out->v) && al16(ZY->v) && al16(g1->v) && al16(FX->v) && al16(ab ? ab->v : nullptr) &&
                            al16(d0) && al16(d1) && al16(d2) && al16(d3);
            // (at most 1024 blocks: each ends with three atomics on three addresses)
            if (v4)
                hipLaunchKernelGGL(k_zy_step_bwd<4>, dim3(nblocks(out->n / 4, 256 * 4, 1024)), dim3(256), 0, st, out->g, out->v, ZY->v, g1->v, FX->v,
                                   ab ? ab->v : nullptr, pen->v, lst->v, ls->v, out->n, d0, a0, d1, a1, d2, a2, d3, a3, dp, ds, dl);
            else
                hipLaunchKernelGGL(k_zy_step_bwd<1>, dim3(nblocks(out->n, 256 * 8, 2048)), dim3(256), 0, st, out->g, out->v, ZY->v, g1->v, FX->v,
                                   ab ? ab->v : nullptr, pen->v, lst->v, ls->v, out->n, d0, a0, d1, a1, d2, a2, d3, a3, dp, ds, dl);
        });
    return out;
}

Tensor Engine::lin3_zy(Tensor FX, Tensor zy, float b, Tensor abn, const float* thr, int groups) {
    Tensor out = make(FX->n, FX->needs_grad || zy->needs_grad || abn->needs_grad);
    if (failed) return out;
    const int G = thr ? groups : 1;
    const size_t per = FX->n / G;
    const dim3 grid(nblocks(per, 256, std::max<size_t>(256 * 32 / G, 1)), G);
    uint32_t* am = out->n >= AMAX_MIN_N ? (uint32_t*)zeros(1) : nullptr;
    if (failed) return out;
    if (am) absmax_of[out->v] = am;
    hipLaunchKernelGGL(k_lin3, grid, dim3(256), 0, st, FX->v, 1.0f, zy->v, thr, b, abn->v, 1.0f, per, out->v, am);
    if (recording && out->needs_grad) {            // no tape entry: zy_step2's VJP (the neighbour on the tape) picks this up
        zy->fl_img = out;
        zy->fl_x = FX;
        zy->fl_b = b;
        zy->fl_thr = thr;
        zy->fl_groups = G;
    }
    return out;
}

// sw_h > 0: Fgrad is [g][h][2M][K] as wgrad_sp wrote it (the step reads it through swap02's map)
Tensor Engine::f_step(Tensor Fc, Tensor Fgrad, float sg, Tensor kst, Tensor ks, size_t n, int sw_h, int sw_n2, int sw_k) {
    Tensor out = make(n, Fc->needs_grad || (Fgrad && Fgrad->needs_grad) || kst->needs_grad || ks->needs_grad);
    if (failed) return out;
    const SwapDims sw{Fgrad ? sw_h : 0, sw_n2, sw_k};
    EW(k_f_step, out->n, Fc->v, Fc->n, Fgrad ? Fgrad->v : nullptr, sg, kst->v, ks->v, out->n, out->v, sw);
    if (recording && out->needs_grad)
        tape.push_back([this, out, Fc, Fgrad, sg, kst, ks, sw]() {
            if (!out->g) return;
            const bool same = Fc->n == out->n;
            int a0 = 1, a1 = 1;
            float* d0 = (Fc->needs_grad && same) ? grad_first(Fc, a0) : nullptr;
            float* d1 = (Fgrad && Fgrad->needs_grad) ? grad_first(Fgrad, a1) : nullptr;
            float* dt = (Fc->needs_grad && !same) ? arena.alloc(out->n) : nullptr;
            float* dk = kst->needs_grad ? grad(kst) : nullptr;
            float* ds = ks->needs_grad ? grad(ks) : nullptr;
            if (failed || (Fc->needs_grad && !same && !dt)) {
                failed = true;
                return;
            }
            hipLaunchKernelGGL(k_f_step_bwd, dim3(nblocks(out->n, 256 * 4, 2048)), dim3(256), 0, st, out->g, out->v, Fgrad ? Fgrad->v : nullptr, sg,
                               kst->v, ks->v, out->n, d0, a0, d1, a1, dt, dk, ds, sw);
            if (dt) bcast_reduce(st, dt, nullptr, out->n, Fc->n, 1.0f, grad(Fc));   // the shared bank: sum over the groups
        });
    return out;
}

Tensor Engine::x_step(Tensor X, Tensor xg, Tensor ost) {
    Tensor out = make(X->n, X->needs_grad || xg->needs_grad || ost->needs_grad);
    if (failed) return out;
    EW(k_x_step, X->n, X->v, xg->v, ost->v, X->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, X, xg, ost]() {
            if (!out->g) return;
            int a0 = 1, a1 = 1;
            float* d0 = X->needs_grad ? grad_first(X, a0) : nullptr;
            float* d1 = xg->needs_grad ? grad_first(xg, a1) : nullptr;
            float* dq = ost->needs_grad ? grad(ost) : nullptr;
            if (failed) return;
            hipLaunchKernelGGL(k_x_step_bwd, dim3(nblocks(out->n, 256, 1024)), dim3(256), 0, st, out->g, xg->v, ost->v, out->n, d0, a0, d1, a1, dq);
        });
    return out;
}

std::pair<Tensor, Tensor> Engine::zy_step2(Tensor ZY, Tensor g1, Tensor FX, Tensor abp, Tensor pen, Tensor lst, Tensor ls, uint32_t* hist0,
                                           int groups) {
    const bool ng = ZY->needs_grad || g1->needs_grad || FX->needs_grad || (abp && abp->needs_grad) || pen->needs_grad ||
                    lst->needs_grad || ls->needs_grad;
    Tensor out = make(ZY->n, ng);
    Tensor abn = make(ZY->n, ng);
    if (failed) return {out, abn};
    {
        const int G = hist0 ? groups : 1;
        const size_t per = ZY->n / G;
        hipLaunchKernelGGL(k_zy_step2, dim3(nblocks(per, 256, zy_blocks_cap(G, hist0 != nullptr)), G), dim3(256), 0, st, ZY->v, g1->v, FX->v,
                           abp ? abp->v : nullptr, pen->v, lst->v, ls->v, per, out->v, abn->v, hist0);
    }
    if (recording && ng)
        tape.push_back([this, out, abn, ZY, g1, FX, abp, pen, lst, ls]() {
            // the combination formed after the step (lin3_zy), if any, and its gradient
            Tensor img = (out->fl_img && out->fl_x == FX) ? out->fl_img : nullptr;
            const float* g3 = img ? img->g : nullptr;
            if (!out->g && !abn->g && !g3) return;
            int a0 = 1, a1 = 1, a2 = 1, a3 = 1;
            float* d0 = ZY->needs_grad ? grad_first(ZY, a0) : nullptr;
            float* d1 = g1->needs_grad ? grad_first(g1, a1) : nullptr;
            float* d2 = FX->needs_grad ? grad_first(FX, a2) : nullptr;
            float* d3 = (abp && abp->needs_grad) ? grad_first(abp, a3) : nullptr;
            float* dp = pen->needs_grad ? grad(pen) : nullptr;
            float* ds = lst->needs_grad ? grad(lst) : nullptr;
            float* dl = ls->needs_grad ? grad(ls) : nullptr;
            if (failed) return;
            const int G = g3 ? out->fl_groups : 1;
            const size_t per = out->n / G;
            auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
            const bool v4 = (per & 3) == 0 && al16(out->g) && al16(abn->g) && al16(g3) && al16(out->v) && al16(ZY->v) && al16(g1->v) &&
                            al16(FX->v) && al16(abp ? abp->v : nullptr) && al16(d0) && al16(d1) && al16(d2) && al16(d3);
            // 32 bytes per lane and stream where the image allows it and the step is large (twelve streams per thread: 276 -> 267 us at 64 mini-batches)
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            if (probe) {
                pe0 = probe->get(), pe1 = probe->get();
                (void)hipEventRecord(pe0, st);
            }
            if (v4 && (per & 7) == 0 && out->n >= ((size_t)8 << 20))
                hipLaunchKernelGGL(k_zy_step2_bwd<8>, dim3(nblocks(per / 8, 256 * 4, std::max<size_t>(2048 / G, 1)), G), dim3(256), 0, st, out->g, abn->g,
                                   g3, out->fl_b, g3 ? out->fl_thr : nullptr, out->v, ZY->v, g1->v, FX->v, abp ? abp->v : nullptr, pen->v, lst->v,
                                   ls->v, per, d0, a0, d1, a1, d2, a2, d3, a3, dp, ds, dl);
            else if (v4)
                hipLaunchKernelGGL(k_zy_step2_bwd<4>, dim3(nblocks(per / 4, 256 * 4, std::max<size_t>(2048 / G, 1)), G), dim3(256), 0, st, out->g, abn->g,
                                   g3, out->fl_b, g3 ? out->fl_thr : nullptr, out->v, ZY->v, g1->v, FX->v, abp ? abp->v : nullptr, pen->v, lst->v,
                                   ls->v, per, d0, a0, d1, a1, d2, a2, d3, a3, dp, ds, dl);
            else
                hipLaunchKernelGGL(k_zy_step2_bwd<1>, dim3(nblocks(per, 256 * 16, std::max<size_t>(2048 / G, 1)), G), dim3(256), 0, st, out->g, abn->g, g3,
                                   out->fl_b, g3 ? out->fl_thr : nullptr, out->v, ZY->v, g1->v, FX->v, abp ? abp->v : nullptr, pen->v, lst->v,
                                   ls->v, per, d0, a0, d1, a1, d2, a2, d3, a3, dp, ds, dl);
            if (probe) {
                (void)hipEventRecord(pe1, st);
                probe->pairs.push_back({pe0, pe1});
            }
        });
    return {out, abn};
}

Tensor Engine::mul(Tensor x, Tensor y) {
    Tensor out = make(x->n, x->needs_grad || y->needs_grad);
    if (failed) return out;
    EW(k_mul, x->n, x->v, y->v, x->n, y->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y]() {
            if (!out->g) return;
            if (x->needs_grad) {
                int acc;
                float* dx = grad_first(x, acc);
                if (dx) EW(k_mul_bwd_x, out->n, out->g, y->v, out->n, y->n, dx, acc);
            }
            if (y->needs_grad) bcast_reduce(st, out->g, x->v, out->n, y->n, 1.0f, grad(y));
        });
    return out;
}

Tensor Engine::relu(Tensor x) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_relu, x->n, x->v, x->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) EW(k_relu_bwd, out->n, out->g, x->v, out->n, dx, acc);
        });
    return out;
}

Tensor Engine::maskmul(Tensor x, const float* mask, float c) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_maskmul, x->n, x->v, mask, c, x->n, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, mask, c]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) EW(k_maskmul, out->n, out->g, mask, c, out->n, dx, acc);
        });
    return out;
}

Tensor Engine::expo(Tensor x) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_exp, x->n, x->v, x->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) EW(k_mul_bwd_x, out->n, out->g, out->v, out->n, out->n, dx, acc);
        });
    return out;
}

Tensor Engine::norm4(Tensor x) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_norm4, x->n / 4, x->v, x->n / 4, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x]() {
            if (out->g) EW(k_norm4_bwd, out->n / 4, out->g, x->v, out->v, out->n / 4, grad(x));
        });
    return out;
}

// update_D's multiplicative step in one pass (model.jl:285-289): out = norm4(exp(-mu * Dgrad) .* Dc), four bases at a time,
// with the arithmetic of the five launches it replaces (lin, mul, exp, mul, norm4) and one kernel for their five VJPs.
// GA (M > 0): Dgrad is read as collapseD of the expanded gradient GA[g][4 fl][2M] it comes from,
//   Dgrad[g][m][ka] = GA[g][ka][m] + GA[g][4 fl - 1 - ka][M + m],
// and its gradient is written back the same way (collapseD and its VJP were launches of their own).
template <bool GA>
static __device__ __forceinline__ void d_step_load(const float* g, size_t i, int M, int fl, float (&gv)[4], size_t (&at)[2][4]) {
    if (!GA) {
        const float4 v = ((const float4*)g)[i];
        gv[0] = v.x, gv[1] = v.y, gv[2] = v.z, gv[3] = v.w;
        return;
    }
    const int Q = 4 * fl;
    const size_t e = 4 * i, perD = (size_t)M * Q, gg = e / perD, r = e - gg * perD;
    const int m = (int)(r / Q), ka0 = (int)(r - (size_t)m * Q);
    const size_t base = gg * (size_t)Q * 2 * M;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        at[0][u] = base + (size_t)(ka0 + u) * 2 * M + m;
        at[1][u] = base + (size_t)(Q - 1 - ka0 - u) * 2 * M + M + m;
        gv[u] = g[at[0][u]] + g[at[1][u]];
    }
}
template <bool GA>
__global__ void k_d_step(const float* __restrict__ g, const float* __restrict__ mu, const float* __restrict__ Dc, size_t n4, float* __restrict__ out, int M,
                         int fl) {
    const float neg = -1.0f * *mu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float gv[4];
        size_t at[2][4];
        d_step_load<GA>(g, i, M, fl, gv, at);
        const float4 dc = ((const float4*)Dc)[i];
        float4 q;
        q.x = expf(gv[0] * neg) * dc.x, q.y = expf(gv[1] * neg) * dc.y, q.z = expf(gv[2] * neg) * dc.z, q.w = expf(gv[3] * neg) * dc.w;
        const float s = q.x + q.y + q.z + q.w;
        ((float4*)out)[i] = make_float4(q.x / s, q.y / s, q.z / s, q.w / s);
    }
}
// go = d out.  dq = (go - <go, out>) / s;  d Dc (+)= dq .* ex;  d Dgrad (+)= -mu * dq .* q;  d mu -= sum(dq .* q .* Dgrad)
template <bool GA>
__global__ void k_d_step_bwd(const float* __restrict__ go, const float* __restrict__ out, const float* __restrict__ g, const float* __restrict__ mu,
                             const float* __restrict__ Dc, size_t n4, float* dg, int ag, float* dDc, int aDc, float* dmu, int M, int fl) {
    const float neg = -1.0f * *mu;
    double sm = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float gg[4];
        size_t at[2][4];
        d_step_load<GA>(g, i, M, fl, gg, at);
        const float4 dc = ((const float4*)Dc)[i], o = ((const float4*)out)[i], gq = ((const float4*)go)[i];
        const float ex[4] = {expf(gg[0] * neg), expf(gg[1] * neg), expf(gg[2] * neg), expf(gg[3] * neg)};
        const float dcv[4] = {dc.x, dc.y, dc.z, dc.w}, ov[4] = {o.x, o.y, o.z, o.w}, gov[4] = {gq.x, gq.y, gq.z, gq.w};
        const float s = (ex[0] * dcv[0] + ex[1] * dcv[1]) + ex[2] * dcv[2] + ex[3] * dcv[3];
        const float dot = gov[0] * ov[0] + gov[1] * ov[1] + gov[2] * ov[2] + gov[3] * ov[3];
        float ddc[4], ddg[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float dq = (gov[u] - dot) / s;
            const float dp = (dq * dcv[u]) * ex[u];                 // through the product with Dc, then through exp
            ddc[u] = dq * ex[u];
            ddg[u] = dp * neg;
            sm += (double)dp * (double)gg[u];
        }
        if (dDc) {
            float4 t = aDc ? ((float4*)dDc)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            ((float4*)dDc)[i] = make_float4(t.x + ddc[0], t.y + ddc[1], t.z + ddc[2], t.w + ddc[3]);
        }
        if (dg && !GA) {
            float4 t = ag ? ((float4*)dg)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            ((float4*)dg)[i] = make_float4(t.x + ddg[0], t.y + ddg[1], t.z + ddg[2], t.w + ddg[3]);
        }
        if (dg && GA) {                                            // every element of d GA is one of these: expandD of d Dgrad
#pragma unroll
            for (int u = 0; u < 4; u++) {
                dg[at[0][u]] = (ag ? dg[at[0][u]] : 0.0f) + ddg[u];
                dg[at[1][u]] = (ag ? dg[at[1][u]] : 0.0f) + ddg[u];
            }
        }
    }
    if (!dmu) return;
    for (int d = 32; d >= 1; d >>= 1) sm += __shfl_xor(sm, d);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dmu, -(float)(red[0] + red[1] + red[2] + red[3]));      // d(-mu) = sum: d mu = -sum
}
// Dgrad: [g][M][4 fl], or with M > 0 the expanded gradient GA [g][4 fl][2M] it is the collapseD of
Tensor Engine::d_step(Tensor Dgrad, Tensor mu, Tensor Dc, int g, int M, int fl) {
    static const bool off = getenv("MOTIFS_NO_D_STEP") != nullptr;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool ga = M > 0;
    const size_t n = ga ? Dgrad->n / 2 : Dgrad->n;
    if (off || Dc->n != n || (n & 3) || !al16(Dgrad->v) || !al16(Dc->v)) {   // a bank shared by the mini-batches: the separate launches
        Tensor Dg = ga ? collapseD(Dgrad, g, M, fl) : Dgrad;
        return norm4(mul(expo(mul(Dg, lin(mu, -1.0f, nullptr, 0.0f, 0.0f))), Dc));
    }
    Tensor out = make(n, Dgrad->needs_grad || mu->needs_grad || Dc->needs_grad);
    if (failed) return out;
    const size_t n4 = n / 4;
    if (ga) hipLaunchKernelGGL(k_d_step<true>, dim3(nblocks(n4, 256, 1024)), dim3(256), 0, st, Dgrad->v, mu->v, Dc->v, n4, out->v, M, fl);
    else hipLaunchKernelGGL(k_d_step<false>, dim3(nblocks(n4, 256, 1024)), dim3(256), 0, st, Dgrad->v, mu->v, Dc->v, n4, out->v, 0, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, Dgrad, mu, Dc, n4, ga, M, fl]() {
            if (!out->g) return;
            int ag = 1, ad = 1;
            float* dg = Dgrad->needs_grad ? grad_first(Dgrad, ag) : nullptr;
            float* dd = Dc->needs_grad ? grad_first(Dc, ad) : nullptr;
            float* dm = mu->needs_grad ? grad(mu) : nullptr;
            if (failed) return;
            if (ga)
                hipLaunchKernelGGL(k_d_step_bwd<true>, dim3(nblocks(n4, 256, 256)), dim3(256), 0, st, out->g, out->v, Dgrad->v, mu->v, Dc->v, n4, dg, ag, dd, ad, dm, M, fl);
            else
                hipLaunchKernelGGL(k_d_step_bwd<false>, dim3(nblocks(n4, 256, 256)), dim3(256), 0, st, out->g, out->v, Dgrad->v, mu->v, Dc->v, n4, dg, ag, dd, ad, dm, 0, 0);
        });
    return out;
}

Tensor Engine::norm4sq(Tensor x, float eps) {
    if ((x->n & 3) || (((uintptr_t)x->v) & 15)) return norm4(lin(mul(x, x), 1.0f, nullptr, 0.0f, eps));
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_norm4sq, x->n / 4, x->v, eps, x->n / 4, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, eps]() {
            if (out->g) EW(k_norm4sq_bwd, out->n / 4, out->g, x->v, out->v, eps, out->n / 4, grad(x));
        });
    return out;
}

Tensor Engine::norml2(Tensor x, int seg, bool squared) {
    Tensor out = make(x->n, x->needs_grad);
    Tensor nrm = make(x->n / seg, false);
    if (failed) return out;
    const unsigned nseg = (unsigned)(x->n / seg);
    if (squared) {
        hipLaunchKernelGGL(k_norml2<true>, dim3(nseg), dim3(nseg >= 256 ? 256 : 1024), 0, st, x->v, seg, out->v, nrm->v);
        if (recording && out->needs_grad)
            tape.push_back([this, out, x, nrm, seg, nseg]() {
                if (out->g) hipLaunchKernelGGL(k_norml2_bwd<true>, dim3(nseg), dim3(nseg >= 256 ? 256 : 1024), 0, st, out->g, out->v, nrm->v, seg, grad(x), x->v);
            });
        return out;
    }
    hipLaunchKernelGGL(k_norml2<false>, dim3(nseg), dim3(nseg >= 256 ? 256 : 1024), 0, st, x->v, seg, out->v, nrm->v);   // few segments: more waves on each
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, nrm, seg, nseg]() {
            if (out->g) hipLaunchKernelGGL(k_norml2_bwd<false>, dim3(nseg), dim3(nseg >= 256 ? 256 : 1024), 0, st, out->g, out->v, nrm->v, seg, grad(x), (const float*)nullptr);
        });
    return out;
}

// update_F's proximal step and the normalisation after it in one kernel per direction (model.jl:298-308; f_step then norml2 before):
//   t = relu((Fc - sg * Fgrad * kst) - kst * ks),  out = t / ||t||_2 per segment (one block per segment, as k_norml2)
// Fgrad through swap02's map (SwapDims), Fc of the same size as the output (a bank per mini-batch, or one mini-batch).
__global__ void k_f_step_norm(const float* Fc, const float* Fgrad, float sg, const float* kst, const float* ks, int seg, float* out, float* nrm_out,
                              SwapDims sw) {
    const size_t base = (size_t)blockIdx.x * seg;
    const float a = *kst, m2 = *kst * *ks;
    double acc = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const float m1 = Fgrad ? (sg * Fgrad[fgrad_at(base + i, sw)]) * a : 0.0f;
        const float t3 = (Fc[base + i] - m1) - m2;
        const float t = t3 > 0.0f ? t3 : 0.0f;
        out[base + i] = t;                                          // kept until the norm is known
        acc += (double)t * t;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[16];
    __shared__ float nrm;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
        nrm = (float)sqrt(t);
        nrm_out[blockIdx.x] = nrm;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < seg; i += blockDim.x) out[base + i] = out[base + i] / nrm;
}
// go = d out.  dt = [out > 0] * (go - out * <go, out>) / nrm;  then f_step's VJP on dt
__global__ void k_f_step_norm_bwd(const float* go, const float* out, const float* nrm_in, const float* Fgrad, float sg, const float* kst, const float* ks,
                                  int seg, float* dFc, int aFc, float* dFg, int aFg, float* dkst, float* dks, SwapDims sw) {
    const size_t base = (size_t)blockIdx.x * seg;
    double acc = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) acc += (double)go[base + i] * out[base + i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ double red[3][16];
    __shared__ float dot;
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[0][i];
        dot = (float)t;
    }
    __syncthreads();
    const float nrm = nrm_in[blockIdx.x], a = *kst, b = *ks;
    double sk = 0, ss = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const float o = out[base + i];
        const float dn = (go[base + i] - o * dot) / nrm;
        const float dt = o > 0.0f ? dn : 0.0f;
        const size_t gi = fgrad_at(base + i, sw);
        if (dFc) dFc[base + i] = (aFc ? dFc[base + i] : 0.0f) + dt;
        if (dFg) dFg[gi] = (aFg ? dFg[gi] : 0.0f) - (sg * a) * dt;
        sk -= (double)dt * (double)((Fgrad ? sg * Fgrad[gi] : 0.0f) + b);
        ss -= (double)dt * (double)a;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sk += __shfl_xor(sk, d);
        ss += __shfl_xor(ss, d);
    }
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = sk, red[2][threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tk = 0, ts = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) tk += red[1][i], ts += red[2][i];
        if (dkst) atomicAdd(dkst, (float)tk);
        if (dks) atomicAdd(dks, (float)ts);
    }
}
Tensor Engine::f_step_norm(Tensor Fc, Tensor Fgrad, float sg, Tensor kst, Tensor ks, size_t n, int seg, int sw_h, int sw_n2, int sw_k) {
    static const bool off = getenv("MOTIFS_NO_F_STEP_NORM") != nullptr;
    if (off || Fc->n != n || n % seg) return norml2(f_step(Fc, Fgrad, sg, kst, ks, n, sw_h, sw_n2, sw_k), seg);   // a shared bank: the separate launches
    Tensor out = make(n, Fc->needs_grad || (Fgrad && Fgrad->needs_grad) || kst->needs_grad || ks->needs_grad);
    Tensor nrm = make(n / seg, false);
    if (failed) return out;
    const SwapDims sw{Fgrad ? sw_h : 0, sw_n2, sw_k};
    const unsigned nseg = (unsigned)(n / seg), thr = nseg >= 256 ? 256 : 1024;
    hipLaunchKernelGGL(k_f_step_norm, dim3(nseg), dim3(thr), 0, st, Fc->v, Fgrad ? Fgrad->v : nullptr, sg, kst->v, ks->v, seg, out->v, nrm->v, sw);
    if (recording && out->needs_grad)
        tape.push_back([this, out, nrm, Fc, Fgrad, sg, kst, ks, seg, nseg, thr, sw]() {
            if (!out->g) return;
            int a0 = 1, a1 = 1;
            float* d0 = Fc->needs_grad ? grad_first(Fc, a0) : nullptr;
            float* d1 = (Fgrad && Fgrad->needs_grad) ? grad_first(Fgrad, a1) : nullptr;
            float* dk = kst->needs_grad ? grad(kst) : nullptr;
            float* ds = ks->needs_grad ? grad(ks) : nullptr;
            if (failed) return;
            hipLaunchKernelGGL(k_f_step_norm_bwd, dim3(nseg), dim3(thr), 0, st, out->g, out->v, nrm->v, Fgrad ? Fgrad->v : nullptr, sg, kst->v, ks->v, seg, d0,
                               a0, d1, a1, dk, ds, sw);
        });
    return out;
}

// into: the per-group sums are ADDED to an existing [groups] tensor (the two terms of the loss meet in one buffer instead of in
// a lin of two; its gradient is read by both VJPs)
Tensor Engine::resid_sumsq_groups(Tensor x, Tensor y, float b, const float* thr, float coef, int groups, Tensor into) {
    Tensor out = into ? into : make(groups, x->needs_grad || y->needs_grad);
    if (failed) return out;
    const size_t per = x->n / groups;
    if (into) into->needs_grad = into->needs_grad || x->needs_grad || y->needs_grad;
    else dev_zero(st, out->v, (size_t)groups);
    hipLaunchKernelGGL(k_resid_sumsq, dim3(nblocks(per, 256, 64), groups), dim3(256), 0, st, x->v, y->v, thr, b, per, coef, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y, b, thr, per, coef, groups]() {
            if (!out->g) return;
            int ax = 1, ay = 1;
            float* dx = x->needs_grad ? grad_first(x, ax) : nullptr;
            float* dy = y->needs_grad ? grad_first(y, ay) : nullptr;
            if (failed) return;
            hipLaunchKernelGGL(k_resid_sumsq_bwd, dim3(nblocks(per, 256, 128), groups), dim3(256), 0, st, out->g, x->v, y->v, thr, b, per, 2.0f * coef,
                               dx, ax, dy, ay);
        });
    return out;
}

Tensor Engine::sumsq_groups(Tensor x, float coef, int groups) {
    Tensor out = make(groups, x->needs_grad);
    if (failed) return out;
    const size_t per = x->n / groups;
    dev_zero(st, out->v, (size_t)groups);
    hipLaunchKernelGGL(k_sumsq_groups, dim3(nblocks(per, 256, 64), groups), dim3(256), 0, st, x->v, per, coef, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, per, coef]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) EW(k_sumsq_groups_bwd, x->n, out->g, x->v, per, x->n, 2.0f * coef, dx, acc);
        });
    return out;
}

// ---------------------------------------------------------------------------------------------
// Toeplitz GEMM: C[s][p][n] (+)= sum_q Aw(s,p,q) * Bm[grp(s)][q][n]
// 64 x BN output tile per block, 16-deep LDS stages, TM x TN = 4 x (BN/16) per thread.
// ---------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256) void k_toep(const float* __restrict__ A, const float* __restrict__ Bm,
                                              float* __restrict__ C, ToepGeom gm, int acc) {
    constexpr int BM = 64, BK = 16, TM = 4, TN = BN / 16;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN];
    const int s = blockIdx.z, p0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    const float* As_g = A + (size_t)s * gm.lda;
    const float* Bg = Bm + (size_t)(s / gm.B) * gm.ldb;
    float accv[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) accv[i][j] = 0.0f;

    for (int q0 = 0; q0 < gm.Q; q0 += BK) {
#pragma unroll
        for (int it = 0; it < (BM * BK) / 256; it++) {
            const int idx = tid + it * 256;
            const int row = idx / BK, qq = idx % BK;
            const int p = p0 + row, q = q0 + qq;
            // unconditional (clamped) load + select: a guarded load would serialise the stage
            const int e = gm.a0 + p * gm.sa + q;
            const bool ok = p < gm.P && q < gm.Q && e >= 0 && e < gm.amax;
            const float v = As_g[ok ? e : 0];
            As[qq][row] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < (BK * BN + 255) / 256; it++) {
            const int idx = tid + it * 256;
            if (idx < BK * BN) {
                const int qq = idx / BN, nn = idx % BN;
                const int q = q0 + qq, n = n0 + nn;
                const bool ok = q < gm.Q && n < gm.N;
                const float v = Bg[ok ? (size_t)q * gm.N + n : 0];
                Bs[qq][nn] = ok ? v : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk++) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = As[kk][ty * TM + i];
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = Bs[kk][tx * TN + j];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) accv[i][j] = fmaf(a[i], b[j], accv[i][j]);
        }
        __syncthreads();
    }
    float* Cs = C + (size_t)s * gm.ldc;
    const bool vec = TN == 4 && (gm.N & 3) == 0 && (gm.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0;   // whole float4 per row
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int p = p0 + ty * TM + i;
        if (p >= gm.P) continue;
        const int nb = n0 + tx * TN;
        if (vec) {
            if (nb < gm.N) {
                float4* o = (float4*)&Cs[(size_t)p * gm.N + nb];
                float4 v = make_float4(accv[i][0], accv[i][1 % TN], accv[i][2 % TN], accv[i][3 % TN]);
                if (acc) {
                    const float4 t = *o;
                    v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                }
                *o = v;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nb + j;
            if (n < gm.N) {
                float* o = &Cs[(size_t)p * gm.N + n];
                *o = acc ? *o + accv[i][j] : accv[i][j];
            }
        }
    }
}

// ---- f32 MFMA forms for narrow outputs (8 < N <= 32): the syntax-layer contraction (model.jl:214, :251) and
// its adjoints, reduction length h*2M.  v_mfma_f32_32x32x2_f32 is an exact f32 fma chain (same numerics as
// the FMA kernel up to summation order).  Block = 2 waves, each a 32-row x 32-column tile; LDS stages of 32.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The same staging split in two, so that the global loads of the next reduction step are in flight while the
// matrix cores work on the current one: load() fills registers, store() moves them to the wave's LDS tile.
template <int ROWS>
struct RowTile {
    static constexpr int NJ = ROWS / 8;           // ROWS*8 float4 per tile, 64 lanes
    const float* base[NJ];                        // sequence start of the lane's row j
    int e0[NJ];                                   // flat offset of the row's window start + the lane's column group
    bool rowok[NJ];
    int cg;
    __device__ __forceinline__ void init(const float* __restrict__ A, const ToepGeom& gm, int g, int loc0, int lane) {
        const int grp_rows = gm.B * gm.P;
        cg = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int row = (lane + j * 64) >> 3;
            const int loc = loc0 + row;
            rowok[j] = loc < grp_rows;
            const int lc = rowok[j] ? loc : grp_rows - 1;
            const int sl = lc / gm.P, p = lc - sl * gm.P;
            base[j] = A + (size_t)(g * gm.B + sl) * gm.lda;
            e0[j] = gm.a0 + p * gm.sa + cg;
        }
    }
    __device__ __forceinline__ void load(const ToepGeom& gm, int q0, float4 (&v)[NJ]) const {
        bool allfast = true;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int e = e0[j] + q0;
            allfast = allfast && q0 + cg + 3 < gm.Q && e >= 0 && e + 3 < gm.amax && (((uintptr_t)(base[j] + e)) & 15) == 0;
        }
        if (__all(allfast)) {                     // wave-uniform: the loads below are unconditional and pipeline
#pragma unroll
            for (int j = 0; j < NJ; j++) v[j] = *(const float4*)(base[j] + e0[j] + q0);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int e = e0[j] + q0, q = q0 + cg;
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool ok = q + u < gm.Q && e + u >= 0 && e + u < gm.amax;
                    const float x = base[j][ok ? e + u : 0];
                    t[u] = ok ? x : 0.0f;
                }
                v[j] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
    __device__ __forceinline__ void store(const float4 (&v)[NJ], float (*As)[33], int lane) const {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int row = (lane + j * 64) >> 3;
            const float4 w = rowok[j] ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            As[row][cg + 0] = w.x;
            As[row][cg + 1] = w.y;
            As[row][cg + 2] = w.z;
            As[row][cg + 3] = w.w;
        }
    }
};

// Stage a [ROWS][32] tile of Toeplitz rows into a wave-private LDS tile (row stride 33).  Rows are
// (group-local) indices loc0.. of group g; lanes walk float4 columns.
template <int ROWS>
static __device__ __forceinline__ void stage_rows(const float* __restrict__ A, const ToepGeom& gm, int g, int loc0, int q0,
                                                  float (*As)[33], int lane) {
    const int grp_rows = gm.B * gm.P;
    constexpr int NJ = ROWS / 8;                  // ROWS*8 float4 per tile, 64 lanes
    const float* src[NJ];
    bool rowok[NJ], fast[NJ];
    bool allfast = true;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int f = lane + j * 64, row = f >> 3, cg = (f & 7) * 4;
        const int loc = loc0 + row;
        rowok[j] = loc < grp_rows;
        const int lc = rowok[j] ? loc : grp_rows - 1;
        const int sl = lc / gm.P, p = lc - sl * gm.P;
        const int q = q0 + cg;
        const int e = gm.a0 + p * gm.sa + q;
        src[j] = A + (size_t)(g * gm.B + sl) * gm.lda + e;
        fast[j] = q + 3 < gm.Q && e >= 0 && e + 3 < gm.amax && (((uintptr_t)src[j]) & 15) == 0;
        allfast = allfast && fast[j];
    }
    float4 v[NJ];
    if (__all(allfast)) {                         // wave-uniform: the loads below are unconditional and pipeline
#pragma unroll
        for (int j = 0; j < NJ; j++) v[j] = *(const float4*)src[j];
    } else {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int f = lane + j * 64, cg = (f & 7) * 4;
            const int q = q0 + cg;
            const int loc = loc0 + (f >> 3);
            const int lc = rowok[j] ? loc : grp_rows - 1;
            const int sl = lc / gm.P, p = lc - sl * gm.P;
            const int e = gm.a0 + p * gm.sa + q;
            const float* base = A + (size_t)(g * gm.B + sl) * gm.lda;
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool ok = q + u < gm.Q && e + u >= 0 && e + u < gm.amax;
                const float x = base[ok ? e + u : 0];
                t[u] = ok ? x : 0.0f;
            }
            v[j] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int f = lane + j * 64, row = f >> 3, cg = (f & 7) * 4;
        const float4 w = rowok[j] ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        As[row][cg + 0] = w.x;
        As[row][cg + 1] = w.y;
        As[row][cg + 2] = w.z;
        As[row][cg + 3] = w.w;
    }
}

// 8 waves per block: 2 row tiles of 32 x 4 interleaved slices of the reduction.  Every wave owns its LDS
// tiles and runs un-synchronised (4 waves per SIMD hide each other's load latency); the four partial
// tiles are summed through LDS at the end.
__global__ __launch_bounds__(512) void k_toep_mfma(const float* __restrict__ A, const float* __restrict__ Bm,
                                                   float* __restrict__ C, ToepGeom gm, int acc) {
    constexpr int BK = 32, KS = 4;
    __shared__ float As[8][32][33];
    __shared__ float Bs[8][BK][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, ks = wave >> 1;
    const int grp_rows = gm.B * gm.P;
    const int g = blockIdx.y;
    const int loc0 = blockIdx.x * 64 + rt * 32;
    const int n0 = blockIdx.z * 32;                // column tile
    const float* Bg = Bm + (size_t)g * gm.ldb;
    f32x16 accv;
#pragma unroll
    for (int i = 0; i < 16; i++) accv[i] = 0.0f;
    RowTile<32> rows;
    rows.init(A, gm, g, loc0, lane);
    float4 va[4];
    float bt[16];
    // B tile [32 k][32 n]: four float4 per lane when the rows allow it (N % 4 == 0, 16-byte aligned bank), else scalars
    const bool bvec = (gm.N & 3) == 0 && (((uintptr_t)Bg) & 15) == 0;
    const int bk = lane >> 3, bn = (lane & 7) * 4;            // this lane's (k, n) corner in the vector form
    const bool bn_ok = n0 + bn + 3 < gm.N;
    const float* bptr = Bg + (size_t)bk * gm.N + n0 + (bn_ok ? bn : 0);
    auto load_b = [&](int q0) {
        if (bvec) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = q0 + bk + 8 * j;
                const bool ok = bn_ok && q < gm.Q;
                const float4 x = *(const float4*)(ok ? bptr + (size_t)(q0 + 8 * j) * gm.N : Bg);
                bt[4 * j + 0] = ok ? x.x : 0.0f;
                bt[4 * j + 1] = ok ? x.y : 0.0f;
                bt[4 * j + 2] = ok ? x.z : 0.0f;
                bt[4 * j + 3] = ok ? x.w : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int f = lane + j * 64, kk = f >> 5, n = f & 31;
                const int q = q0 + kk;
                const bool ok = q < gm.Q && n0 + n < gm.N;
                const float x = Bg[ok ? (size_t)q * gm.N + n0 + n : 0];
                bt[j] = ok ? x : 0.0f;
            }
        }
    };
    if (ks * BK < gm.Q) {
        rows.load(gm, ks * BK, va);
        load_b(ks * BK);
    }
    for (int q0 = ks * BK; q0 < gm.Q; q0 += KS * BK) {
        rows.store(va, As[wave], lane);
        if (bvec) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int u = 0; u < 4; u++) Bs[wave][bk + 8 * j][bn + u] = bt[4 * j + u];
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int f = lane + j * 64;
                Bs[wave][f >> 5][f & 31] = bt[j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (q0 + KS * BK < gm.Q) {                 // next step's global loads fly while the MFMAs below run
            rows.load(gm, q0 + KS * BK, va);
            load_b(q0 + KS * BK);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[wave][lane & 31][kk + (lane >> 5)];
            const float b = Bs[wave][kk + (lane >> 5)][lane & 31];
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accv, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // reduce the KS partial tiles: slices 1..3 park theirs in their A tile region ([16 regs][64 lanes] floats)
    float* park = &As[wave][0][0];               // 32*33 = 1056 floats >= 1024
    __syncthreads();
    if (ks > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) park[r * 64 + lane] = accv[r];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int o = 1; o < KS; o++) {
            const float* src = &As[rt + 2 * o][0][0];
#pragma unroll
            for (int r = 0; r < 16; r++) accv[r] += src[r * 64 + lane];
        }
        const int col = n0 + (lane & 31);
        if (col < gm.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int loc = loc0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (loc < grp_rows) {
                    const int sl = loc / gm.P, p = loc - sl * gm.P;
                    float* o = C + (size_t)(g * gm.B + sl) * gm.ldc + (size_t)p * gm.N + col;
                    *o = acc ? *o + accv[r] : accv[r];
                }
            }
        }
    }
}

// out[i] (+)= part[0][i] + part[1][i] + ... (k partial images of n floats, added in order)
__global__ void k_sum_parts(const float* __restrict__ part, size_t n, int k, float* __restrict__ out, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.0f;
        for (int j0 = 0; j0 < k; j0 += 8) {        // eight loads in flight, added in order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = j0 + u < k ? part[(size_t)(j0 + u) * n + i] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < k) a = (j0 + u == 0) ? v[u] : a + v[u];
        }
        out[i] = acc ? out[i] + a : a;
    }
}

// The syntax-layer analysis (model.jl:251, conv(img, F, flipped)) with the image resident in LDS.  The Toeplitz
// rows of one sequence overlap by (H-1)/H, so a block keeps the 32 + H - 1 image rows of its 32 output rows in
// LDS (CC channels at a time, double buffered) and feeds the matrix cores from there: one ds_read per MFMA for
// the image operand.  The filter operand comes from a copy of the bank in fragment order (k_frag_b): the 64
// lane values of four consecutive MFMA steps lie together, so one 16-byte load per lane (1 KB per wave,
// contiguous) feeds four MFMAs; the bank stays in L2.  4 waves split the reduction (window rows j), partial
// tiles meet in LDS and leave as one contiguous [32][N] span.
//   Bf[g][t][lane][u] = B[g][8t + 2u + (lane >> 5)][min(lane & 31, N-1)]     (columns >= N are never stored)
__global__ void k_frag_b(const float* __restrict__ Bm, int G, int Q, int N, float* __restrict__ out) {
    const size_t per = (size_t)(Q / 8) * 256, total = per * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, r = i - g * per;
        const int u = (int)(r & 3), lane = (int)((r >> 2) & 63);
        const size_t t = r >> 8;
        const size_t q = 8 * t + 2 * u + (lane >> 5);
        out[i] = Bm[g * (size_t)Q * N + q * N + min(lane & 31, N - 1)];
    }
}
template <int H, int CC>
// nsplit > 1 (steps of few reads: 36 blocks on 256 CUs otherwise): the channel chunks are dealt to nsplit blocks per job, each
// leaves its partial tile at C + split * cstride and k_sum_parts adds them.
__global__ __launch_bounds__(256) void k_ana_lds(const float* __restrict__ A, const float* __restrict__ Bf, float* __restrict__ C,
                                                 ToepGeom gm, int acc, int tps, int64_t ldbf, int nsplit, size_t cstride) {
    constexpr int ST = CC + 1, ROWS = 32 + H, JW = H / 4, C4 = CC / 4, NV = (ROWS * C4 + 255) / 256;   // one spare row
    extern __shared__ float lds[];                 // 2 x [ROWS][ST]; at the end 4 x [32][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x / nsplit, split = blockIdx.x - job * nsplit;
    const int s = job / tps, p0 = (job - s * tps) * 32;
    const int W = gm.sa, NCH = W / CC, N = gm.N;
    const int ch0 = split * NCH / nsplit, ch1 = (split + 1) * NCH / nsplit;      // this block's chunks
    C += (size_t)split * cstride;
    const float* img = A + (size_t)s * gm.lda + gm.a0 + (size_t)p0 * W;
    const int lim = gm.amax - gm.a0 - p0 * W;      // valid flat range seen from img
    const float4* Bg = (const float4*)(Bf + (size_t)(s / gm.B) * ldbf) + lane;
    float4 v[NV];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / C4, c4 = idx - row * C4;
            const int flat = row * W + c0 + c4 * 4;
            const bool ok = idx < ROWS * C4 && flat + 3 < lim;
            const float4 x = *(const float4*)(img + (ok ? flat : 0));
            v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / C4, c4 = idx - row * C4;
            if (idx < ROWS * C4) {
                float* d = buf + row * ST + c4 * 4;
                d[0] = v[i].x, d[1] = v[i].y, d[2] = v[i].z, d[3] = v[i].w;
            }
        }
    };
    f32x16 accv;
#pragma unroll
    for (int i = 0; i < 16; i++) accv[i] = 0.0f;
    const int aoff = ((lane & 31) + wave * JW) * ST + (lane >> 5);
    constexpr int NI = CC / 2, FJ = CC / 8;         // MFMAs and filter fragments (4 MFMAs each) per window row and chunk
    constexpr int PF = FJ % 5 == 0 ? 5 : 4;         // fragments in flight
    static_assert(CC % 8 == 0 && FJ % PF == 0, "the fragment ring carries over from window row to window row");
    // fragments of window row j, chunk ch: channels ch*CC + 8 * f .., f < FJ
    auto bbase = [&](int ch, int j) -> const float4* { return Bg + ((((size_t)j * W + (size_t)ch * CC) >> 3) << 6); };
    float4 br[PF];
    gload(ch0 * CC);
    {
        const float4* b0 = bbase(ch0, wave * JW);
#pragma unroll
        for (int i = 0; i < PF; i++) br[i] = b0[i * 64];
    }
    lstore(lds + (ch0 & 1) * (ROWS * ST));
    __syncthreads();
    for (int ch = ch0; ch < ch1; ch++) {
        float* buf = lds + (ch & 1) * (ROWS * ST);
        const bool more = ch + 1 < ch1;
        if (more) gload((ch + 1) * CC);            // in flight under the MFMAs below
        const float* a = buf + aoff;
        float ar[4];
        ar[0] = a[0], ar[1] = a[2], ar[2] = a[4], ar[3] = a[6];
#pragma unroll 1
        for (int jj = 0; jj < JW; jj++) {
            const float4* cur = bbase(ch, wave * JW + jj);
            const float4* nxt = jj + 1 < JW ? bbase(ch, wave * JW + jj + 1) : more ? bbase(ch + 1, wave * JW) : cur;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const float4 bq = br[(i / 4) % PF];
                const float av = ar[i & 3], bv = (i & 3) == 0 ? bq.x : (i & 3) == 1 ? bq.y : (i & 3) == 2 ? bq.z : bq.w;
                // four steps ahead; past the end of the row that is the start of the next one (the spare LDS row after the last)
                ar[i & 3] = i + 4 < NI ? a[2 * (i + 4)] : a[ST + 2 * (i + 4 - NI)];
                accv = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accv, 0, 0, 0);
                if ((i & 3) == 3) {                // the fragment is used up: refill its slot
                    const int f = i / 4 + PF;
                    br[(i / 4) % PF] = f < FJ ? cur[f * 64] : nxt[(f - FJ) * 64];
                }
                __builtin_amdgcn_sched_barrier(0); // keep the prefetch distances as written
            }
            a += ST;
        }
        if (more) lstore(lds + ((ch + 1) & 1) * (ROWS * ST));
        __syncthreads();
    }
    // partial tiles -> LDS [wave][row][32]; register r of lane l is (row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31)
    float* red = lds + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; r++) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = accv[r];
    __syncthreads();
    const int nrow = min(32, gm.P - p0);
    float* Cs = C + (size_t)s * gm.ldc + (size_t)p0 * N;
    const int total = nrow * N;                    // one contiguous span of the output
    const bool vec = (N & 3) == 0 && (((uintptr_t)Cs) & 15) == 0;
    if (vec) {
        for (int e4 = tid; e4 * 4 < total; e4 += 256) {
            const int e = e4 * 4, row = e / N, col = e - row * N;
            const float* q = lds + row * 32 + col;
            float4 o;
            o.x = (q[0] + q[1024]) + (q[2048] + q[3072]);
            o.y = (q[1] + q[1025]) + (q[2049] + q[3073]);
            o.z = (q[2] + q[1026]) + (q[2050] + q[3074]);
            o.w = (q[3] + q[1027]) + (q[2051] + q[3075]);
            float4* dst = (float4*)(Cs + e);
            if (acc) {
                const float4 t = *dst;
                o.x += t.x, o.y += t.y, o.z += t.z, o.w += t.w;
            }
            *dst = o;
        }
    } else {
        for (int e = tid; e < total; e += 256) {
            const int row = e / N, col = e - row * N;
            const float* q = lds + row * 32 + col;
            const float o = (q[0] + q[1024]) + (q[2048] + q[3072]);
            Cs[e] = acc ? Cs[e] + o : o;
        }
    }
}
// the shapes k_ana_lds takes: whole in-bounds windows of H = 4k image rows, up to 32 output channels
template <int H, int CC>
static bool launch_ana_lds(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& gm, int acc) {
    if (gm.sa <= 0 || gm.Q != H * gm.sa || gm.sa % CC != 0 || gm.N < 9 || gm.N > 32) return false;
    if (gm.a0 < 0 || (int64_t)gm.a0 + (int64_t)(gm.P - 1) * gm.sa + gm.Q > gm.amax) return false;
    if ((gm.sa & 7) || (gm.a0 & 3) || (gm.lda & 3) || (((uintptr_t)A) & 15)) return false;
    constexpr int need = 2 * (32 + H) * (CC + 1) * 4 > 16384 ? 2 * (32 + H) * (CC + 1) * 4 : 16384;
    const int tps = (gm.P + 31) / 32;
    const long jobs = (long)gm.S * tps;
    const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
    const size_t perf = (size_t)(gm.Q / 8) * 256;
    bool fresh;
    float* Bf = e.relayout(Bm, 1, gm.Q, gm.N, 0, perf * gB, fresh);
    if (!Bf) return true;
    if (fresh) hipLaunchKernelGGL(k_frag_b, dim3(nblocks(perf * gB)), dim3(256), 0, e.st, Bm, gB, gm.Q, gm.N, Bf);
    // blocks per CU: the count whose rounds x resident waves is smallest (all blocks take the same time)
    int best = 1;
    long cost = -1;
    for (int k = 1; k <= 4 && k * need <= 160 * 1024; k++) {
        const long c = ((jobs + 256L * k - 1) / (256L * k)) * k;
        if (cost < 0 || c <= cost) cost = c, best = k;
    }
    int lds = (160 * 1024 / best) & ~1023;
    if (lds > 64 * 1024) lds = 64 * 1024;
    if (lds < need) lds = need;
    const int nch = gm.sa / CC;
    if (jobs < 256 && nch > 1 && gm.ldc == (int64_t)gm.P * gm.N) {   // few reads: a block per (job, channel chunk), partial tiles, one sum
        const size_t cn = (size_t)gm.S * gm.ldc;
        float* part = e.arena.alloc(cn * nch);
        if (!part) {
            e.failed = true;
            return true;
        }
        hipLaunchKernelGGL((k_ana_lds<H, CC>), dim3((unsigned)(jobs * nch)), dim3(256), (size_t)lds, e.st, A, Bf, part, gm, 0, tps,
                           (int64_t)(gm.ldb == 0 ? 0 : perf), nch, cn);
        hipLaunchKernelGGL(k_sum_parts, dim3(nblocks(cn / 4 + 1, 256, 1024)), dim3(256), 0, e.st, part, cn, nch, C, acc);
        return true;
    }
    hipLaunchKernelGGL((k_ana_lds<H, CC>), dim3((unsigned)jobs), dim3(256), (size_t)lds, e.st, A, Bf, C, gm, acc, tps,
                       (int64_t)(gm.ldb == 0 ? 0 : perf), 1, (size_t)0);
    return true;
}

// ---- the same contraction on the binary16 matrix instruction, three products per term ("f16x3") ----------------------------
// k_ana_lds sits at 81 % matrix-pipe busy on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak): the largest kernel of a 64-mini-batch step.
// v_mfma_f32_32x32x16_f16 runs 16x the flops per instruction cycle.  Each float32 operand x is split as hi = f16(x s), lo = f16(x s - hi)
// (s a power of two that puts the tensor's largest magnitude just under 2^15, so that lo is a normal binary16 number for every entry
// that matters: hi + lo carries 22 bits), and a b = hi_a hi_b + hi_a lo_b + lo_a hi_b to 2^-22, accumulated in float32 as before
// (the two small products in an accumulator of their own).  Emulated on the shapes of this layer: 4e-8 of the largest output against
// 3.5e-7 for a float32 GEMM with float32 accumulation.  Three instructions of 32 cycles replace eight of 64 per 16 reduction terms: the
// kernel stops being bound by the matrix pipe and becomes bound by the filter fragments it pulls from L2, which is why a wave carries
// NT row tiles per fragment (the float32 form re-reads the whole 614 KB bank per 32 output rows).
__global__ void k_absmax(const float* __restrict__ x0, size_t n, size_t stride, int nseg, uint32_t* __restrict__ out) {   // out: pre-zeroed; bits of max |x|
    // nseg segments of n floats, `stride` apart; blocks stride over the segments; ONE atomic per block, and only if it can raise the maximum
    // (an atomic per wave - 15 000 on one address - took 185 us for 116 MB, the streaming itself 20)
    uint32_t m = 0;
    for (int sg = blockIdx.y; sg < nseg; sg += gridDim.y) {
        const float* x = x0 + (size_t)sg * stride;
        const bool al = (((uintptr_t)x) & 15) == 0;
        const size_t n4 = al ? n / 4 : 0;
        const float4* x4 = (const float4*)x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const float4 v = x4[i];
            m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
        if (blockIdx.x == 0)
            for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    }
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}
// the power of two that takes a tensor whose largest magnitude has the bits `mb` into [2^14, 2^15) (1 for an all-zero tensor; Inf / NaN
// inputs give Inf / NaN outputs either way)
static __device__ __forceinline__ int f16x3_scale_exp(uint32_t mb) {
    const int e = (int)((mb >> 23) & 255u);
    if (mb == 0 || e == 255) return 0;
    const int se = 141 - (e ? e : 1);
    return se > 120 ? 120 : se < -120 ? -120 : se;
}
// filter fragments: Bf16[g][q16][plane][lane] = 8 halves, plane 0 = hi, 1 = lo; lane (n = lane & 31, kb = lane >> 5) holds
// B[g][16 q16 + 8 kb + i][min(n, N - 1)] * 2^seB, i < 8
__global__ void k_frag_b_hilo(const float* __restrict__ Bm, int G, int Q, int N, const uint32_t* __restrict__ bmax, uint4* __restrict__ out) {
    const float sB = __uint_as_float((uint32_t)(f16x3_scale_exp(*bmax) + 127) << 23);
    const size_t per = (size_t)(Q / 16) * 128, total = per * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, r = i - g * per;
        const int lane = (int)(r & 63), plane = (int)((r >> 6) & 1);
        const size_t q16 = r >> 7;
        const int n = min(lane & 31, N - 1), kb = lane >> 5;
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t hh[2];
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const size_t q = 16 * q16 + 8 * kb + 2 * u + v;
                const float x = Bm[g * (size_t)Q * N + q * N + n] * sB;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)(x - (float)hi);
                hh[v] = (uint32_t)__builtin_bit_cast(uint16_t, plane ? lo : hi);
            }
            w[u] = hh[0] | (hh[1] << 16);
        }
        out[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int H, int CC, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ana_f16x3(const float* __restrict__ A, const uint4* __restrict__ Bf, float* __restrict__ C,
                                                                                               ToepGeom gm, int acc, int tps, int64_t ldbf,
                                                                                               const uint32_t* __restrict__ amax, const uint32_t* __restrict__ bmax) {
    constexpr int RS = CC + 8, ROWS = 32 * NT + H, JW = H / 4, C4 = CC / 4, NV = (ROWS * C4 + 255) / 256, KT = CC / 16;
    static_assert(CC % 16 == 0 && H % 4 == 0, "whole k-steps of 16 channels, four waves over the window rows");
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsh[];        // 2 buffers x 2 planes x [ROWS][RS] halves; at the end 4 x NT x [32][32] floats
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int s = job / tps, p0 = (job - s * tps) * (32 * NT);
    const int W = gm.sa, NCH = W / CC, N = gm.N;
    const float* img = A + (size_t)s * gm.lda + gm.a0 + (size_t)p0 * W;
    const int lim = gm.amax - gm.a0 - p0 * W;      // valid flat range seen from img
    const int seA = f16x3_scale_exp(*amax), seB = f16x3_scale_exp(*bmax);
    const float sA = __uint_as_float((uint32_t)(seA + 127) << 23);
    const uint4* Bg = Bf + (size_t)(s / gm.B) * ldbf + lane;
    float4 v[NV];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / C4, c4 = idx - row * C4;
            const int flat = row * W + c0 + c4 * 4;
            const bool ok = idx < ROWS * C4 && flat + 3 < lim;
            const float4 x = *(const float4*)(img + (ok ? flat : 0));
            v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](uint16_t* buf) {             // buf: [2 planes][ROWS][RS]
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / C4, c4 = idx - row * C4;
            if (idx < ROWS * C4) {
                const float x[4] = {v[i].x * sA, v[i].y * sA, v[i].z * sA, v[i].w * sA};
                uint16_t hb[4], lb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const _Float16 hi = (_Float16)x[u];
                    const _Float16 lo = (_Float16)(x[u] - (float)hi);
                    hb[u] = __builtin_bit_cast(uint16_t, hi), lb[u] = __builtin_bit_cast(uint16_t, lo);
                }
                uint16_t* d = buf + row * RS + c4 * 4;
                *(uint2*)d = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
                *(uint2*)(d + ROWS * RS) = make_uint2((uint32_t)lb[0] | ((uint32_t)lb[1] << 16), (uint32_t)lb[2] | ((uint32_t)lb[3] << 16));
            }
        }
    };
    f32x16 accM[NT], accS[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) accM[t][i] = 0.0f, accS[t][i] = 0.0f;
    // fragments of (window row j, chunk ch): k-steps (j W + ch CC) / 16 + t, t < KT; per k-step 2 planes x 64 lanes of 16 bytes
    auto bfrag = [&](int ch, int j, int t, int plane) -> uint4 { return Bg[(((size_t)j * W + (size_t)ch * CC) / 16 + t) * 128 + plane * 64]; };
    uint4 bh[KT], bl[KT];
    gload(0);
#pragma unroll
    for (int t = 0; t < KT; t++) bh[t] = bfrag(0, wave * JW, t, 0), bl[t] = bfrag(0, wave * JW, t, 1);
    lstore(ldsh);
    __syncthreads();
    const int arow = (lane & 31), acol = 8 * (lane >> 5);
    for (int ch = 0; ch < NCH; ch++) {
        const uint16_t* buf = ldsh + (ch & 1) * (2 * ROWS * RS);
        const bool more = ch + 1 < NCH;
        if (more) gload((ch + 1) * CC);            // in flight under the MFMAs below
#pragma unroll 1
        for (int jj = 0; jj < JW; jj++) {
            const int j = wave * JW + jj;
            // the next window row's fragments are requested before this one's matrix instructions
            uint4 nh[KT], nl[KT];
            const bool last = jj + 1 == JW;
            if (!last || more) {
                const int nch = last ? ch + 1 : ch, nj = last ? wave * JW : j + 1;
#pragma unroll
                for (int t = 0; t < KT; t++) nh[t] = bfrag(nch, nj, t, 0), nl[t] = bfrag(nch, nj, t, 1);
            }
#pragma unroll
            for (int t = 0; t < KT; t++) {
                const f16x8v Bh = __builtin_bit_cast(f16x8v, bh[t]), Bl = __builtin_bit_cast(f16x8v, bl[t]);
#pragma unroll
                for (int rt = 0; rt < NT; rt++) {
                    const uint16_t* ap = buf + (32 * rt + arow + j) * RS + 16 * t + acol;
                    const f16x8v Ah = __builtin_bit_cast(f16x8v, *(const uint4*)ap);
                    const f16x8v Al = __builtin_bit_cast(f16x8v, *(const uint4*)(ap + ROWS * RS));
                    accM[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, accM[rt], 0, 0, 0);
                    accS[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, accS[rt], 0, 0, 0);
                    accS[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, accS[rt], 0, 0, 0);
                }
            }
            if (!last || more) {
#pragma unroll
                for (int t = 0; t < KT; t++) bh[t] = nh[t], bl[t] = nl[t];
            }
        }
        if (more) lstore(ldsh + ((ch + 1) & 1) * (2 * ROWS * RS));
        __syncthreads();
    }
    // partial tiles -> LDS [wave][tile][row][32], scaled back; register r of lane l is (row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31)
    float* red = (float*)ldsh;
    const float iA = __uint_as_float((uint32_t)(127 - seA) << 23), iB = __uint_as_float((uint32_t)(127 - seB) << 23);
#pragma unroll
    for (int rt = 0; rt < NT; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            red[((wave * NT + rt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = ((accM[rt][r] + accS[rt][r]) * iA) * iB;
    __syncthreads();
    const int nrow = min(32 * NT, gm.P - p0);
    float* Cs = C + (size_t)s * gm.ldc + (size_t)p0 * N;
    const int total = nrow * N;                    // one contiguous span of the output
    for (int e = tid; e < total; e += 256) {
        const int row = e / N, col = e - row * N;
        const float* q = red + row * 32 + col;     // tile row / 32, row % 32: consecutive tiles are 1024 floats apart, as rows are 32
        const float o = (q[0] + q[NT * 1024]) + (q[2 * NT * 1024] + q[3 * NT * 1024]);
        Cs[e] = acc ? Cs[e] + o : o;
    }
}
template <int H, int CC>
static bool launch_ana_f16x3(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& gm, int acc) {
    const bool off = gemm_f32_only();        // A/B: the float32 matrix instruction for every launch
    if (off) return false;
    if (gm.sa <= 0 || gm.Q != H * gm.sa || gm.sa % CC != 0 || gm.N < 9 || gm.N > 32) return false;
    if (gm.a0 < 0 || (int64_t)gm.a0 + (int64_t)(gm.P - 1) * gm.sa + gm.Q > gm.amax) return false;
    if ((gm.sa & 15) || (gm.a0 & 3) || (gm.lda & 3) || (((uintptr_t)A) & 15)) return false;
    // steps of few reads keep the float32 form (its split over channel chunks); MOTIFS_GEMM_F16_MIN lowers the bar (tests: the
    // one-mini-batch goldens through this kernel)
    const long min_jobs = gemm_f16_min(256);      // measured: pays from 8 mini-batches (288 row tiles), costs 4-7 % at 1-4
    if ((long)gm.S * ((gm.P + 31) / 32) < min_jobs) return false;
    const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
    const size_t perf = (size_t)(gm.Q / 16) * 128;                        // uint4 per bank
    // the largest magnitudes (bits): of the bank (cached with its fragments) and of this call's image
    bool fresh;
    float* Bf = e.relayout(Bm, 7, gm.Q, gm.N, 0, perf * gB * 4 + 4, fresh);
    uint32_t* am = (uint32_t*)e.zeros(1);
    if (!Bf || !am) {
        e.failed = true;
        return true;
    }
    uint32_t* bm = (uint32_t*)(Bf + perf * gB * 4);
    if (fresh) {
        dev_zero(e.st, (float*)bm, 1);
        hipLaunchKernelGGL(k_absmax, dim3(nblocks((size_t)gB * gm.Q * gm.N / 4 + 1, 256, 512)), dim3(256), 0, e.st, Bm, (size_t)gB * gm.Q * gm.N, (size_t)0, 1, bm);
        hipLaunchKernelGGL(k_frag_b_hilo, dim3(nblocks(perf * gB)), dim3(256), 0, e.st, Bm, gB, gm.Q, gm.N, bm, (uint4*)Bf);
    }
    // the image's largest magnitude: kept by the kernel that formed it (k_lin3), else one streaming pass over the windows' ranges
    // (sequences are lda apart and use [a0, amax) of each)
    const auto known = e.absmax_of.find((const void*)A);
    if (known != e.absmax_of.end()) {
        am = known->second;
    } else {
        const size_t nper = (size_t)(gm.amax - gm.a0);
        const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>(8, nper / 4 / 1024));
        const unsigned by = (unsigned)std::min<int>(gm.S, std::max<int>(1, 2048 / (int)bx));
        hipLaunchKernelGGL(k_absmax, dim3(bx, by), dim3(256), 0, e.st, A + gm.a0, nper, (size_t)gm.lda, gm.S, am);
    }
    auto go = [&](auto nt) {
        constexpr int NT = decltype(nt)::value;
        const int tps = (gm.P + 32 * NT - 1) / (32 * NT);
        const size_t lds = std::max<size_t>((size_t)2 * 2 * (32 * NT + H) * (CC + 8) * 2, (size_t)4 * NT * 4096);
        auto kern = k_ana_f16x3<H, CC, NT>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)((long)gm.S * tps)), dim3(256), lds, e.st, A, (const uint4*)Bf, C, gm, acc, tps,
                           (int64_t)(gm.ldb == 0 ? 0 : perf), am, bm);
    };
    // row tiles per wave: the fewest padded rows, then the most rows per fragment
    const int pad2 = (gm.P + 63) / 64 * 64, pad3 = (gm.P + 95) / 96 * 96;
    if (pad3 <= pad2) go(std::integral_constant<int, 3>{});
    else go(std::integral_constant<int, 2>{});
    return true;
}

// N <= 4 outputs (D-layer synthesis and its relatives: an image 4 bases wide): one row per lane, 4
// accumulators, 8 waves = 2 row tiles of 64 x 4 slices of the reduction, wave-private LDS as above.
__global__ __launch_bounds__(512) void k_toep_n4(const float* __restrict__ A, const float* __restrict__ Bm,
                                                 float* __restrict__ C, ToepGeom gm, int acc) {
    constexpr int BK = 32, KS = 4;
    __shared__ float As[8][64][33];
    __shared__ float4 Bs[8][BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave & 1, ks = wave >> 1;
    const int grp_rows = gm.B * gm.P;
    const int g = blockIdx.y;
    const int loc0 = blockIdx.x * 128 + rt * 64;
    const float* Bg = Bm + (size_t)g * gm.ldb;
    float4 accv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q0 = ks * BK; q0 < gm.Q; q0 += KS * BK) {
        stage_rows<64>(A, gm, g, loc0, q0, As[wave], lane);
        {
            const int q = q0 + (lane & 31);
            float t[4];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const bool ok = q < gm.Q && n < gm.N;
                const float x = Bg[ok ? (size_t)q * gm.N + n : 0];
                t[n] = ok ? x : 0.0f;
            }
            Bs[wave][lane & 31] = make_float4(t[0], t[1], t[2], t[3]);   // lanes 32..63 rewrite the same values
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < BK; kk++) {
            const float a = As[wave][lane][kk];
            const float4 b = Bs[wave][kk];
            accv.x = fmaf(a, b.x, accv.x);
            accv.y = fmaf(a, b.y, accv.y);
            accv.z = fmaf(a, b.z, accv.z);
            accv.w = fmaf(a, b.w, accv.w);
        }
        __builtin_amdgcn_wave_barrier();
    }
    float4* park = (float4*)&As[wave][0][0];
    __syncthreads();
    if (ks > 0) park[lane] = accv;
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int o = 1; o < KS; o++) {
            const float4 t = ((const float4*)&As[rt + 2 * o][0][0])[lane];
            accv.x += t.x;
            accv.y += t.y;
            accv.z += t.z;
            accv.w += t.w;
        }
        const int loc = loc0 + lane;
        if (loc < grp_rows) {
            const int sl = loc / gm.P, p = loc - sl * gm.P;
            float* o = C + (size_t)(g * gm.B + sl) * gm.ldc + (size_t)p * gm.N;
            const float v[4] = {accv.x, accv.y, accv.z, accv.w};
#pragma unroll
            for (int n = 0; n < 4; n++)
                if (n < gm.N) o[n] = acc ? o[n] + v[n] : v[n];
        }
    }
}

// dB[g][q][n] (+)= sum_{s in g, p} Aw(s,p,q) * C[s][p][n] for N <= 32: 4 waves, each 32 q x 32 n
__global__ __launch_bounds__(256) void k_wgrad_mfma(const float* __restrict__ A, const float* __restrict__ C,
                                                    float* __restrict__ dB, ToepGeom gm, int acc) {
    constexpr int BQ = 128, BK = 32;
    __shared__ float As[BK][BQ + 4];   // [k][q]
    __shared__ float Cs[BK][32 + 1];   // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y, q0 = blockIdx.x * BQ, n0 = blockIdx.z * 32;
    const int KT = gm.B * gm.P;
    f32x16 accv;
#pragma unroll
    for (int i = 0; i < 16; i++) accv[i] = 0.0f;
    for (int k0 = 0; k0 < KT; k0 += BK) {
        // A stage: 32 k-rows x 128 q = 1024 float4, 4 per thread; consecutive threads walk q
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = tid + j * 256, kk = f >> 5, cg = (f & 31) * 4;
            const int k = k0 + kk, q = q0 + cg;
            const int kc = k < KT ? k : 0;
            const int sl = kc / gm.P, p = kc - sl * gm.P;
            const int e = gm.a0 + p * gm.sa + q;
            const float* base = A + (size_t)(g * gm.B + sl) * gm.lda;
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool ok = k < KT && q + u < gm.Q && e + u >= 0 && e + u < gm.amax;
                const float x = base[ok ? e + u : 0];
                t[u] = ok ? x : 0.0f;
            }
            *(float4*)&As[kk][cg] = make_float4(t[0], t[1], t[2], t[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = tid + j * 256, kk = f >> 5, n = f & 31;
            const int k = k0 + kk;
            const bool ok = k < KT && n0 + n < gm.N;
            const int kc = ok ? k : 0;
            const int sl = kc / gm.P, p = kc - sl * gm.P;
            const float x = C[(size_t)(g * gm.B + sl) * gm.ldc + (size_t)p * gm.N + (ok ? n0 + n : 0)];
            Cs[kk][n] = ok ? x : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wave * 32 + (lane & 31)];
            const float b = Cs[kk + (lane >> 5)][lane & 31];
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accv, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + (lane & 31);
    if (col < gm.N) {
        float* out = dB + (size_t)g * gm.Q * gm.N;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int q = q0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (q < gm.Q) {
                float* o = &out[(size_t)q * gm.N + col];
                *o = acc ? *o + accv[r] : accv[r];
            }
        }
    }
}

// ---- "tall" Toeplitz GEMMs: few output channels (N <= 8) over a window of H = Q/sa whole image rows of sa >= 64
// columns (the D-layer synthesis: N = 4 bases, H = filter_len, sa = 2M).  Every image row then serves H*N outputs:
//   W[s][rho][(i',n)] = sum_j A[s][rho][j] * Bm[i'][j][n]          (row GEMM, MFMA)
//   C[s][r][n]        = sum_i' W[s][r + i' + a0/sa][i'][n]          (gather of H terms)
// which reads each image element once per 32 x (H*N) tile instead of once per output row.
__global__ void k_tall_bt(const float* __restrict__ Bm, int g, int H, int W, int N, float* __restrict__ Bt) {
    const size_t per = (size_t)H * W * N, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i % per;          // r indexes Bt [W][H][N]
        const int n = (int)(r % N), ip = (int)((r / N) % H), j = (int)(r / ((size_t)N * H));
        Bt[i] = Bm[gg * per + ((size_t)ip * W + j) * N + n];
    }
}
__global__ void k_tall_bt_T(const float* __restrict__ dBt, int g, int H, int W, int N, float* __restrict__ dBm, int acc) {
    const size_t per = (size_t)H * W * N, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i % per;          // r indexes dBm [H][W][N]
        const int n = (int)(r % N), j = (int)((r / N) % W), ip = (int)(r / ((size_t)N * W));
        const float v = dBt[gg * per + ((size_t)j * H + ip) * N + n];
        dBm[i] = acc ? dBm[i] + v : v;
    }
}
// (y, yb): an image of C's layout added on the way out, C = gather + yb * y - the "- S" / "+ S" that follows every D-layer
// synthesis (model.jl:238, :276, :313), which was a launch of its own
__global__ void k_tall_gather(const float* __restrict__ Wt, float* __restrict__ C, int S, int P, int H, int N, int R, int off,
                              int64_t ldc, int acc, const float* __restrict__ y, float yb) {
    const size_t total = (size_t)S * P * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N), r = (int)((i / N) % P), s = (int)(i / ((size_t)N * P));
        float a = 0.0f;
        for (int i0 = 0; i0 < H; i0 += 8) {        // eight loads in flight, added in order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ip = i0 + u, rho = r + ip + off;
                v[u] = (ip < H && rho >= 0 && rho < R) ? Wt[(((size_t)s * R + rho) * H + ip) * N + n] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ip = i0 + u, rho = r + ip + off;
                if (ip < H && rho >= 0 && rho < R) a += v[u];
            }
        }
        const size_t oi = (size_t)s * ldc + (size_t)r * N + n;
        if (y) a += yb * y[oi];
        C[oi] = acc ? C[oi] + a : a;
    }
}
// dW[s][rho][i'][n] = dC[s][rho - i' - off][n]
__global__ void k_tall_scatter(const float* __restrict__ dC, float* __restrict__ dW, int S, int P, int H, int N, int R, int off,
                               int64_t ldc) {
    const size_t total = (size_t)S * R * H * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N), ip = (int)((i / N) % H);
        const size_t t = i / ((size_t)N * H);
        const int rho = (int)(t % R), s = (int)(t / R);
        const int r = rho - ip - off;
        dW[i] = (r >= 0 && r < P) ? dC[(size_t)s * ldc + (size_t)r * N + n] : 0.0f;
    }
}

// The row GEMM of the tall forms on the matrix cores: C[r][n] = sum_q A[r][q] B[q][n] over contiguous rows of Q
// floats, N <= 64 outputs (the D-layer synthesis: 400 channels -> 12 lags x 4 bases).  HBM-bound (the rows are read
// once), so a block takes 32 rows whole into LDS with 16-byte loads, its 4 waves split the reduction, and
// v_mfma_f32_16x16x4_f32 tiles cover N in 16-column blocks (48 = 3 blocks, nothing padded).  The filter comes in
// fragment order, one 16-byte load per lane and reduction step for all column blocks:
//   Bf[g][ks][lane][cb] = B[g][4 ks + (lane >> 4)][min(16 cb + (lane & 15), N-1)]
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_frag_b16(const float* __restrict__ Bm, int G, int Q, int N, float* __restrict__ out) {
    const size_t per = (size_t)(Q / 4) * 256, total = per * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, r = i - g * per;
        const int cb = (int)(r & 3), lane = (int)((r >> 2) & 63);
        const size_t q = 4 * (r >> 8) + (lane >> 4);
        out[i] = 16 * cb < N ? Bm[g * (size_t)Q * N + q * N + min(16 * cb + (lane & 15), N - 1)] : 0.0f;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void k_rowgemm_lds(const float* __restrict__ A, const float* __restrict__ Bf, float* __restrict__ C,
                                                     int rpg, int tpg, int Q, int N, int64_t ldbf, int acc) {
    constexpr int NV = 15, U = 4;                  // 16-byte loads per thread (Q <= 480); reduction steps per prefetch unit
    extern __shared__ float lds[];                 // [32][Q + 4]; at the end 4 x [32][16 NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x / tpg, r0 = (blockIdx.x - g * tpg) * 32;
    const int nrow = min(32, rpg - r0), ST = Q + 4, Q4 = Q >> 2;   // Q % 16 == 0: rows 4*odd dwords apart, conflict-free operand reads
    const size_t row0 = (size_t)g * rpg + r0;
    const float4* At = (const float4*)(A + row0 * Q);
    const int nf4 = nrow * Q4, tf4 = 32 * Q4;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = tid + i * 256;
        const float4 x = At[idx < nf4 ? idx : 0];
        v[i] = idx < nf4 ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        int row = tid / Q4, c4 = tid - row * Q4;
        const int drow = 256 / Q4, dc = 256 - drow * Q4;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            if (tid + i * 256 < tf4) *(float4*)(lds + row * ST + c4 * 4) = v[i];
            row += drow, c4 += dc;
            if (c4 >= Q4) c4 -= Q4, row++;
        }
    }
    __syncthreads();
    f32x4 accv[2][NT];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < NT; cb++) accv[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = Q >> 4;                        // reduction steps (4 channels each) of this wave
    const float* a0p = lds + (lane & 15) * ST + wave * (Q >> 2) + (lane >> 4);
    const float* a1p = a0p + 16 * ST;
    const float4* bp = (const float4*)(Bf + (size_t)g * ldbf) + (size_t)(wave * nks) * 64 + lane;
    float4 cur[U], nxt[U];
#pragma unroll
    for (int i = 0; i < U; i++) cur[i] = bp[(size_t)min(i, nks - 1) * 64];
    for (int u0 = 0; u0 < nks; u0 += U) {
#pragma unroll
        for (int i = 0; i < U; i++) nxt[i] = bp[(size_t)min(u0 + U + i, nks - 1) * 64];
#pragma unroll
        for (int i = 0; i < U; i++) {
            if (u0 + i < nks) {
                const float a0 = a0p[4 * (u0 + i)], a1 = a1p[4 * (u0 + i)];
                const float bv[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
                for (int cb = 0; cb < NT; cb++) {
                    accv[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[cb], accv[0][cb], 0, 0, 0);
                    accv[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[cb], accv[1][cb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < U; i++) cur[i] = nxt[i];
    }
    __syncthreads();                               // the image rows are done with: partial tiles take their place
    constexpr int RW = 16 * NT;
    float* red = lds + wave * (32 * RW);
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < NT; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(rb * 16 + 4 * (lane >> 4) + r) * RW + cb * 16 + (lane & 15)] = accv[rb][cb][r];
    __syncthreads();
    float* Cs = C + row0 * N;
    const int total = nrow * N;                    // one contiguous span of the output
    const bool vec = (N & 3) == 0 && (((uintptr_t)Cs) & 15) == 0;
    if (vec) {
        for (int e4 = tid; e4 * 4 < total; e4 += 256) {
            const int e = e4 * 4, row = e / N, col = e - row * N;
            const float* q = lds + row * RW + col;
            float4 o;
            o.x = (q[0] + q[32 * RW]) + (q[64 * RW] + q[96 * RW]);
            o.y = (q[1] + q[32 * RW + 1]) + (q[64 * RW + 1] + q[96 * RW + 1]);
            o.z = (q[2] + q[32 * RW + 2]) + (q[64 * RW + 2] + q[96 * RW + 2]);
            o.w = (q[3] + q[32 * RW + 3]) + (q[64 * RW + 3] + q[96 * RW + 3]);
            float4* dst = (float4*)(Cs + e);
            if (acc) {
                const float4 t = *dst;
                o.x += t.x, o.y += t.y, o.z += t.z, o.w += t.w;
            }
            *dst = o;
        }
    } else {
        for (int e = tid; e < total; e += 256) {
            const int row = e / N, col = e - row * N;
            const float* q = lds + row * RW + col;
            const float o = (q[0] + q[32 * RW]) + (q[64 * RW] + q[96 * RW]);
            Cs[e] = acc ? Cs[e] + o : o;
        }
    }
}
// The tall form in ONE launch for steps of few reads (the row GEMM and the gather after it were two launches of ~36 blocks):
// a block owns TR = 33 - H output rows of one read, takes the 32 image rows rho = r0 + off .. r0 + off + 31 they draw on into
// LDS (rows outside the read are zeros), forms their [32][H N] products as k_rowgemm_lds does, and adds the H shifted slices
//     C[s][r][n] (+)= sum_ip W[r + ip + off][ip][n]   (+ yb * y[s][r][n])
// straight from the partial tiles.  The image rows of neighbouring blocks overlap by H - 1: only worth it while the launch,
// not HBM, is the cost.
template <int NT>
__global__ __launch_bounds__(256) void k_tall_fused(const float* __restrict__ A, const float* __restrict__ Bf, float* __restrict__ C, int R, int P,
                                                    int H, int Q, int N, int off, int tps, int B, int64_t ldbf, int64_t ldc, int acc,
                                                    const float* __restrict__ y, float yb) {
    constexpr int NV = 15, U = 4;
    extern __shared__ float lds[];                 // [32][Q + 4]; at the end 4 x [32][16 NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TR = 33 - H;
    const int s = blockIdx.x / tps, r0 = (blockIdx.x - s * tps) * TR;
    const int rho0 = r0 + off, ST = Q + 4, Q4 = Q >> 2;
    const float4* At = (const float4*)(A + (size_t)s * R * Q);
    const int tf4 = 32 * Q4;
    float4 v[NV];
    {
        int row = tid / Q4, c4 = tid - row * Q4;
        const int drow = 256 / Q4, dc = 256 - drow * Q4;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int rho = rho0 + row;
            const bool ok = tid + i * 256 < tf4 && rho >= 0 && rho < R;
            const float4 x = At[ok ? (size_t)rho * Q4 + c4 : 0];
            v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            row += drow, c4 += dc;
            if (c4 >= Q4) c4 -= Q4, row++;
        }
    }
    // the first bank fragments are on their way while the image rows settle in LDS
    const int nks = Q >> 4;                        // reduction steps (4 channels each) of this wave
    const float4* bp = (const float4*)(Bf + (size_t)(s / B) * ldbf) + (size_t)(wave * nks) * 64 + lane;
    float4 cur[U], nxt[U];
#pragma unroll
    for (int i = 0; i < U; i++) cur[i] = bp[(size_t)min(i, nks - 1) * 64];
    {
        int row = tid / Q4, c4 = tid - row * Q4;
        const int drow = 256 / Q4, dc = 256 - drow * Q4;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            if (tid + i * 256 < tf4) *(float4*)(lds + row * ST + c4 * 4) = v[i];
            row += drow, c4 += dc;
            if (c4 >= Q4) c4 -= Q4, row++;
        }
    }
    __syncthreads();
    f32x4 accv[2][NT];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < NT; cb++) accv[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a0p = lds + (lane & 15) * ST + wave * (Q >> 2) + (lane >> 4);
    const float* a1p = a0p + 16 * ST;
    for (int u0 = 0; u0 < nks; u0 += U) {
#pragma unroll
        for (int i = 0; i < U; i++) nxt[i] = bp[(size_t)min(u0 + U + i, nks - 1) * 64];
#pragma unroll
        for (int i = 0; i < U; i++) {
            if (u0 + i < nks) {
                const float a0 = a0p[4 * (u0 + i)], a1 = a1p[4 * (u0 + i)];
                const float bv[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
                for (int cb = 0; cb < NT; cb++) {
                    accv[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[cb], accv[0][cb], 0, 0, 0);
                    accv[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[cb], accv[1][cb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < U; i++) cur[i] = nxt[i];
    }
    __syncthreads();                               // the image rows are done with: partial tiles take their place
    constexpr int RW = 16 * NT;
    float* red = lds + wave * (32 * RW);
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < NT; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(rb * 16 + 4 * (lane >> 4) + r) * RW + cb * 16 + (lane & 15)] = accv[rb][cb][r];
    __syncthreads();
    const int nrow = min(TR, P - r0);
    for (int e = tid; e < nrow * N; e += 256) {
        const int rr = e / N, n = e - rr * N;
        float a = 0.0f;
        for (int ip = 0; ip < H; ip++) {           // W row (local) rr + ip, column ip N + n; rows outside the read were zeros
            const float* q = lds + (rr + ip) * RW + ip * N + n;
            a += (q[0] + q[32 * RW]) + (q[64 * RW] + q[96 * RW]);
        }
        const size_t oi = (size_t)s * ldc + (size_t)(r0 + rr) * N + n;
        if (y) a += yb * y[oi];
        C[oi] = acc ? C[oi] + a : a;
    }
}
static bool launch_tall_fused(Engine& e, const float* A, const float* Bt, float* C, const ToepGeom& gm, const ToepGeom& rg, int acc, const float* y,
                              float yb) {
    static const bool off = getenv("MOTIFS_NO_TALL_FUSED") != nullptr;
    const int H = gm.Q / gm.sa, R = gm.amax / gm.sa;
    if (off || H < 2 || H > 16 || gm.a0 % gm.sa != 0) return false;
    if (rg.a0 != 0 || rg.sa != rg.Q || rg.lda != (int64_t)rg.P * rg.Q) return false;
    if ((rg.Q & 15) || rg.Q > 480 || rg.Q < 64 || rg.N < 9 || rg.N > 64 || (((uintptr_t)A) & 15)) return false;
    const int TR = 33 - H, tps = (gm.P + TR - 1) / TR;
    if ((long)gm.S * tps > 1024) return false;     // many reads: the two-launch form reads every image row once
    const int groups = rg.ldb == 0 ? 1 : rg.S / rg.B;
    const size_t perf = (size_t)(rg.Q / 4) * 256;
    bool fresh;
    float* Bf = e.relayout(Bt, 2, rg.Q, rg.N, 0, perf * groups, fresh);
    if (!Bf) return true;
    if (fresh) hipLaunchKernelGGL(k_frag_b16, dim3(nblocks(perf * groups)), dim3(256), 0, e.st, Bt, groups, rg.Q, rg.N, Bf);
    const int NT = (rg.N + 15) / 16;
    const size_t lds = std::max((size_t)32 * (rg.Q + 4) * 4, (size_t)4 * 32 * 16 * NT * 4);
    const dim3 grid((unsigned)(gm.S * tps));
    const int64_t ldbf = rg.ldb == 0 ? 0 : (int64_t)perf;
#define TALLF(NTV) \
    hipLaunchKernelGGL((k_tall_fused<NTV>), grid, dim3(256), lds, e.st, A, Bf, C, R, gm.P, H, rg.Q, gm.N, gm.a0 / gm.sa, tps, gm.B, ldbf, gm.ldc, acc, y, yb)
    if (NT == 1) TALLF(1);
    else if (NT == 2) TALLF(2);
    else if (NT == 3) TALLF(3);
    else TALLF(4);
#undef TALLF
    return true;
}

static bool launch_rowgemm16(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& rg, int acc, int groups, long rpg, int tpg);
// rows must be contiguous ([S][P][Q] with nothing between sequences), Q a multiple of 16 up to 480, N <= 64
static bool launch_rowgemm_lds(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& rg, int acc) {
    if (rg.a0 != 0 || rg.sa != rg.Q || rg.lda != (int64_t)rg.P * rg.Q || rg.ldc != (int64_t)rg.P * rg.N) return false;
    if ((rg.Q & 15) || rg.Q > 480 || rg.Q < 64 || rg.N < 9 || rg.N > 64 || (((uintptr_t)A) & 15)) return false;
    const int groups = rg.ldb == 0 ? 1 : rg.S / rg.B;
    const long rpg = (long)(rg.S / groups) * rg.P;
    const int tpg = (int)((rpg + 31) / 32);
    if (launch_rowgemm16(e, A, Bm, C, rg, acc, groups, rpg, tpg)) return true;
    const size_t perf = (size_t)(rg.Q / 4) * 256;
    bool fresh;
    float* Bf = e.relayout(Bm, 2, rg.Q, rg.N, 0, perf * groups, fresh);
    if (!Bf) return true;
    if (fresh) hipLaunchKernelGGL(k_frag_b16, dim3(nblocks(perf * groups)), dim3(256), 0, e.st, Bm, groups, rg.Q, rg.N, Bf);
    const int NT = (rg.N + 15) / 16;
    const size_t lds = std::max((size_t)32 * (rg.Q + 4) * 4, (size_t)4 * 32 * 16 * NT * 4);
    const dim3 grid((unsigned)(groups * tpg));
    const int64_t ldbf = rg.ldb == 0 ? 0 : (int64_t)perf;
#define ROWGEMM(NTV) hipLaunchKernelGGL((k_rowgemm_lds<NTV>), grid, dim3(256), lds, e.st, A, Bf, C, (int)rpg, tpg, rg.Q, rg.N, ldbf, acc)
    if (NT == 1) ROWGEMM(1);
    else if (NT == 2) ROWGEMM(2);
    else if (NT == 3) ROWGEMM(3);
    else ROWGEMM(4);
#undef ROWGEMM
    return true;
}

// Short windows, wide outputs (the D-layer analysis, model.jl:238, and the adjoint of its synthesis: Q = 4 fl <= 64
// terms of the signal, N = 2M = 400 outputs): out[s][p][n] (+)= sum_q sig[s][a0 + p sa + q] B[q][n].  The output
// stream bounds it (116 MB at cfg-2).  Every wave works alone: its 32 rows of windows live in registers as MFMA
// operands for the whole job (Q/2 values per lane), the bank comes from L2 in fragment order, one column tile of
// 32 at a time, the next tile's fragments in flight:
//   Bf[g][ct][kg][lane][u] = B[g][8 kg + 2 u + (lane >> 5)][min(32 ct + (lane & 31), N-1)]
__global__ void k_frag_bw(const float* __restrict__ Bm, int G, int Q, int N, float* __restrict__ out) {
    const int KG = Q / 8, NCT = (N + 31) / 32;
    const size_t per = (size_t)NCT * KG * 256, total = per * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, r = i - g * per;
        const int u = (int)(r & 3), lane = (int)((r >> 2) & 63);
        const int kg = (int)((r >> 8) % KG), ct = (int)((r >> 8) / KG);
        const int q = 8 * kg + 2 * u + (lane >> 5);
        out[i] = Bm[g * (size_t)Q * N + (size_t)q * N + min(32 * ct + (lane & 31), N - 1)];
    }
}
template <int KG>
__global__ __launch_bounds__(256) void k_toep_wide(const float* __restrict__ A, const float* __restrict__ Bf, float* __restrict__ C,
                                                   ToepGeom gm, int acc, int tps, int64_t ldbf, int SO, int cs) {
    extern __shared__ float lds[];                 // the block's 32 output rows, [32][SO]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // cs > 1 (few reads): cs blocks share a job's 32 rows, each takes a run of column tiles
    const int job = blockIdx.x / cs, part = blockIdx.x - job * cs;
    const int s = job / tps, p0 = (job - s * tps) * 32;
    const float* sig = A + (size_t)s * gm.lda;
    // this wave's first bank fragments are on their way while the windows are staged
    const float4* bp = (const float4*)(Bf + (size_t)(s / gm.B) * ldbf) + lane;
    const int nct_all = (gm.N + 31) >> 5;
    const int ct_lo = part * nct_all / cs, nct = (part + 1) * nct_all / cs;      // this block's tiles [ct_lo, nct)
    float4 cur[KG], nxt[KG];
    if (ct_lo + wave < nct) {
#pragma unroll
        for (int kg = 0; kg < KG; kg++) cur[kg] = bp[(size_t)((ct_lo + wave) * KG + kg) * 64];
    }
    float a[4 * KG];
    // the block's windows overlap: their span (31 sa + Q floats) goes through LDS once, coalesced, instead of one
    // strided 4-byte gather per operand and wave
    const int span = 31 * gm.sa + 8 * KG;
    if (span <= 32 * SO) {
        const int e0 = gm.a0 + p0 * gm.sa;
        for (int i = tid; i < span; i += 256) {
            const int e = e0 + i;
            const bool ok = e >= 0 && e < gm.amax;
            const float x = sig[ok ? e : 0];
            lds[i] = ok ? x : 0.0f;
        }
        __syncthreads();
        const float* w = lds + (lane & 31) * gm.sa + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 4 * KG; ks++) a[ks] = w[2 * ks];
        __syncthreads();                           // the tile stores below reuse the space
    } else {
        const int base = gm.a0 + (p0 + (lane & 31)) * gm.sa + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 4 * KG; ks++) {
            const int e = base + 2 * ks;
            const bool ok = e >= 0 && e < gm.amax;
            const float x = sig[ok ? e : 0];
            a[ks] = ok ? x : 0.0f;
        }
    }
    for (int ct = ct_lo + wave; ct < nct; ct += 4) {       // wave w: column tiles w, w+4, ..
        const int cn = min(ct + 4, nct - 1);
#pragma unroll
        for (int kg = 0; kg < KG; kg++) nxt[kg] = bp[(size_t)(cn * KG + kg) * 64];
        f32x16 accv;
#pragma unroll
        for (int i = 0; i < 16; i++) accv[i] = 0.0f;
#pragma unroll
        for (int kg = 0; kg < KG; kg++) {
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * kg + 0], cur[kg].x, accv, 0, 0, 0);
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * kg + 1], cur[kg].y, accv, 0, 0, 0);
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * kg + 2], cur[kg].z, accv, 0, 0, 0);
            accv = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * kg + 3], cur[kg].w, accv, 0, 0, 0);
        }
        const int col = ct * 32 + (lane & 31);
        if (col < gm.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) lds[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SO + col] = accv[r];
        }
#pragma unroll
        for (int kg = 0; kg < KG; kg++) cur[kg] = nxt[kg];
    }
    __syncthreads();
    // the rows leave whole: [nrow][N] is one contiguous span of the output
    const int nrow = min(32, gm.P - p0), N4 = gm.N >> 2;
    float* Cs = C + (size_t)s * gm.ldc + (size_t)p0 * gm.N;
    if (cs > 1) {                                  // this block's columns of every row
        const int c_lo = ct_lo * 32, nc = min(gm.N, nct * 32) - c_lo;
        for (int idx = tid; idx < nrow * nc; idx += 256) {
            const int row = idx / nc, col = c_lo + idx - row * nc;
            const float o = lds[row * SO + col];
            float* dst = Cs + (size_t)row * gm.N + col;
            *dst = acc ? *dst + o : o;
        }
    } else if ((gm.N & 3) == 0 && (((uintptr_t)Cs) & 15) == 0) {
        int row = tid / N4, c4 = tid - row * N4;
        const int drow = 256 / N4, dc = 256 - drow * N4;
        for (int idx = tid; idx < nrow * N4; idx += 256) {
            float4 o = *(const float4*)(lds + row * SO + c4 * 4);
            float4* dst = (float4*)Cs + idx;
            if (acc) {
                const float4 t = *dst;
                o.x += t.x, o.y += t.y, o.z += t.z, o.w += t.w;
            }
            *dst = o;
            row += drow, c4 += dc;
            if (c4 >= N4) c4 -= N4, row++;
        }
    } else {
        for (int idx = tid; idx < nrow * gm.N; idx += 256) {
            const int row = idx / gm.N, col = idx - row * gm.N;
            const float o = lds[row * SO + col];
            Cs[idx] = acc ? Cs[idx] + o : o;
        }
    }
}
// ---- k_toep_wide on the binary16 matrix instruction, three products per term (see k_ana_f16x3) ----
// Here both operands are small - the block's windows (172 floats at stride 4) and the bank (48 x 400) - and the reduction (Q <= 64 terms) lies
// inside one block, so the windows are scaled by the largest magnitude of the BLOCK's own span (found while it is staged in LDS; no pass over
// the signal) and the bank by its own (cached with its fragments).  9 instructions of 32 cycles per 32 x 32 tile instead of 24 of 64.
// Bf16[g][ct][t][plane][lane] = 8 halves: B[g][16 t + 8 kb + u][min(32 ct + n, N - 1)] * 2^seB, lane = (n, kb)
// FRAG_SPLIT blocks per group: each finds the group's largest magnitude for itself (its scale; the bank is a few thousand floats in L2), leaves it
// in bmax[g] for the consumers, and writes its share of the fragments - one launch where a zero fill, a maximum pass and the re-layout were three.
constexpr int FRAG_SPLIT = 8;
__global__ __launch_bounds__(256) void k_frag_bw16(const float* __restrict__ Bm, int Q, int N, uint32_t* __restrict__ bmax, uint4* __restrict__ out) {
    __shared__ uint32_t wm[4];
    const int g = blockIdx.x / FRAG_SPLIT, part = blockIdx.x - g * FRAG_SPLIT, tid = threadIdx.x, n = Q * N;
    const float* B = Bm + (size_t)g * n;
    uint32_t m = 0;
    if ((n & 3) == 0 && (((uintptr_t)B) & 15) == 0) {              // eight 16-byte loads in flight (one load per turn was 75 trips to L2 in a row)
        const float4* B4 = (const float4*)B;
        for (int i0 = tid; i0 < n / 4; i0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = B4[min(i0 + 256 * u, n / 4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                m = max(max(m, max(__float_as_uint(v[u].x) & 0x7fffffffu, __float_as_uint(v[u].y) & 0x7fffffffu)),
                        max(__float_as_uint(v[u].z) & 0x7fffffffu, __float_as_uint(v[u].w) & 0x7fffffffu));
        }
    } else {
        for (int i = tid; i < n; i += 256) m = max(m, __float_as_uint(B[i]) & 0x7fffffffu);
    }
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (tid == 0 && part == 0) bmax[g] = m;
    const float sB = __uint_as_float((uint32_t)(f16x3_scale_exp(m) + 127) << 23);
    const int KT = Q / 16, NCT = (N + 31) / 32, per = NCT * KT * 128;
    for (int r = part * 256 + tid; r < per; r += 256 * FRAG_SPLIT) {
        const int lane = r & 63, plane = (r >> 6) & 1;
        const int ctt = r >> 7;
        const int t = ctt % KT, ct = ctt / KT;
        const int col = min(32 * ct + (lane & 31), N - 1), kb = lane >> 5;
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t hh[2];
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const int q = 16 * t + 8 * kb + 2 * u + v;
                const float x = B[(size_t)q * N + col] * sB;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)(x - (float)hi);
                hh[v] = (uint32_t)__builtin_bit_cast(uint16_t, plane ? lo : hi);
            }
            w[u] = hh[0] | (hh[1] << 16);
        }
        out[(size_t)g * per + r] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// ---- k_rowgemm_lds on the same instruction: C[r][n] = sum_q A[r][q] B[q][n], rows of Q floats read once ----
// The float32 form is a block per 32 rows that loads, then multiplies, then writes: co-resident blocks run in step, so HBM idles while the matrix
// pipe works (34 us with the matrix loop removed, 49 with it, 26 at the HBM rate).  Prefetching the next rows from the same waves does not help on
// gfx9: a wave's memory loads return in order (vmcnt), so the first wait for a filter fragment is a wait for the whole prefetch in front of it
// (a persistent form with the next tile in flight and this instruction measured the same 48.6 us).  So the waves take ROLES here, each with its own
// counter: one block of 8 waves per CU for the whole launch; waves 0-3 only move rows (8 rows of a tile each: global -> registers -> hi / lo
// planes in LDS, scaled by the largest magnitude of their 8 rows, each register asked for its piece of the tile after next as soon as it is
// converted), waves 4-7 only multiply (the bank's fragments - k_frag_bw16's [g][ct][t][plane][lane] - sit in their registers for the whole launch,
// every fourth k-step of 16 channels each, partial tiles met in LDS) and write the finished rows.  One block barrier per tile: the row planes and
// the partial tiles are both double buffered.  What the time is (s_memtime per wave and turn, -DRG16_TIMING + tools/rg16_timing.py): a mover and a
// multiplier share each SIMD and their instructions do not overlap - a turn costs the SUM of the conversion (12 vector instructions per 16 bytes),
// the 42 matrix instructions and the tile's write-out, ~4 900 cycles against ~4 500 for a tile at the HBM rate; the first three turns (cold
// fragments, code and pages) cost ~7 000 each.  48.6 -> ~40 us per launch; the depth of the movers' ring (1, 2 or 3 tiles) does not matter.
#ifdef RG16_TIMING
__device__ unsigned long long g_rg16_ts[256][2][16][5];
#define RG16_TS(role, turn, k) \
    if (lane == 0 && (wave & 3) == 0 && (turn) < 16) g_rg16_ts[blockIdx.x][role][turn][k] = __builtin_amdgcn_s_memtime()
#else
#define RG16_TS(role, turn, k)
#endif
template <int NCT, int KS, int NV, int RING = 2>
__global__ __launch_bounds__(512) void k_rowgemm16(const float* __restrict__ A, const uint4* __restrict__ Bf, float* __restrict__ C, int rpg, int tpg,
                                                   int ntiles, int Q, int N, int64_t ldbf, int acc, const uint32_t* __restrict__ bmax) {
    // NV: 16-byte loads per mover lane (8 rows of Q <= 32 NV floats); RW: partial-tile row stride, 4 rows apart = 32 banks apart
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsr[];        // 2 x (hi [32][RS], lo [32][RS]) halves; 2 x 4 x [32][RW] floats
    __shared__ int sexp[2][4];                     // scale exponent of (buffer, 8-row group)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RS = Q + 8, Q4 = Q >> 2, KT = Q >> 4;                         // rows 4 * odd dwords apart: conflict-free 16-byte operand reads
    const int plane = 32 * RS, bufsz = 2 * plane;
    const int RW = ((N + 15) & ~15) + 8;
    float* red0 = (float*)(ldsr + 2 * bufsz);
    // tiles of this block: a run of consecutive ones, t0 + j (with a bank per group a strided deal changed the bank - 100 KB of fragments - at every tile)
    const int tpb = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x, t0 = blockIdx.x * tpb, nmine = max(0, min(ntiles, t0 + tpb) - t0);
    // RING: tiles a mover has in flight (register sets)
    const int nturns = (nmine + RING) / RING * RING;   // nmine + 1 turns, rounded up to the movers' ring
    if (wave < 4) {
        // ---- movers ----
        float4 v[RING][NV];
        const int nf8 = 8 * Q4;
        // where tile j's 8 rows of this wave lie: rows past the end of the group are never written out - their lanes re-read the last valid
        // 16 bytes - and a turn past the block's last tile re-reads the first 16 bytes of A (no branch, no select: the same NV loads on every
        // path, so the wait for one tile's registers leaves the tiles behind it in flight)
        auto rows_of = [&](int j, const float4*& At, int& last) {
            const int tile = t0 + (j < nmine ? j : 0), g = tile / tpg, r0 = (tile - g * tpg) * 32 + 8 * wave;
            const int nf4 = j < nmine ? max(0, min(8, rpg - r0)) * Q4 : 0;
            last = max(nf4, 1) - 1;
            At = (const float4*)(A + (nf4 > 0 ? ((size_t)g * rpg + r0) * Q : (size_t)0));
        };
        // where a lane's pieces go in its wave's 8 rows of a plane (halves); the pieces past the 8 rows go to the 8 spare halves behind row 0
        int lofs[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / Q4, c4 = idx - row * Q4;
            lofs[i] = idx < nf8 ? row * RS + c4 * 4 : Q;
        }
        // one turn: the tile in vs goes to buffer j & 1 as hi / lo planes, and every register is asked for its piece of tile j + RING as soon as
        // it has been converted.  12 vector instructions per 16 bytes: packed multiply, packed conversions, the remainder x s - hi as a packed fma
        auto turn = [&](float4* vs, int j) {
            float mf = 0.0f;
#pragma unroll
            for (int i = 0; i < NV; i++) mf = fmaxf(fmaxf(fmaxf(fmaxf(mf, fabsf(vs[i].x)), fabsf(vs[i].y)), fabsf(vs[i].z)), fabsf(vs[i].w));
            uint32_t m = __float_as_uint(mf);
            for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
            const int seA = f16x3_scale_exp(m);
            RG16_TS(0, j, 1);
            const float sA = __uint_as_float((uint32_t)(seA + 127) << 23);
            uint16_t* buf = ldsr + (j & 1) * bufsz + 8 * wave * RS;
            if (lane == 0) sexp[j & 1][wave] = seA;
            const float4* An;
            int lastn;
            rows_of(j + RING, An, lastn);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                f32x2v p0 = {vs[i].x, vs[i].y}, p1 = {vs[i].z, vs[i].w};
                p0 *= sA, p1 *= sA;
                const f16x2v h0 = __builtin_convertvector(p0, f16x2v), h1 = __builtin_convertvector(p1, f16x2v);
                const f32x2v r0 = p0 - __builtin_convertvector(h0, f32x2v), r1 = p1 - __builtin_convertvector(h1, f32x2v);
                const f16x2v l0 = __builtin_convertvector(r0, f16x2v), l1 = __builtin_convertvector(r1, f16x2v);
                uint16_t* d = buf + lofs[i];
                *(uint2*)d = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
                *(uint2*)(d + plane) = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
                vs[i] = An[min(lane + i * 64, lastn)];
            }
        };
#pragma unroll
        for (int u = 0; u < RING; u++) {
            const float4* At;
            int last;
            rows_of(u, At, last);
#pragma unroll
            for (int i = 0; i < NV; i++) v[u][i] = At[min(lane + i * 64, last)];
        }
        for (int j3 = 0; j3 < nturns; j3 += RING) {   // turn j: tile j into buffer j & 1 (the multipliers are on tile j - 1)
#pragma unroll
            for (int u = 0; u < RING; u++) {
                RG16_TS(0, j3 + u, 0);
                turn(v[u], j3 + u);                // (a turn past the last tile converts what the filler loads brought: finite, never read)
                RG16_TS(0, j3 + u, 2);
                __syncthreads();
                RG16_TS(0, j3 + u, 3);
            }
        }
    } else {
        // ---- multipliers ----
        const int cw = wave - 4, ctid = tid - 256;
        float iB = 1.0f;
        uint4 bfr[KS][2 * NCT];
        auto bload = [&](int g) {
            iB = __uint_as_float((uint32_t)(127 - f16x3_scale_exp(bmax[ldbf ? g : 0])) << 23);
            const uint4* bp = Bf + (size_t)g * ldbf + lane;
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const int t = min(cw + 4 * k, KT - 1);
#pragma unroll
                for (int ct = 0; ct < NCT; ct++)
                    bfr[k][2 * ct] = bp[(size_t)((ct * KT + t) * 2) * 64], bfr[k][2 * ct + 1] = bp[(size_t)((ct * KT + t) * 2 + 1) * 64];
            }
        };
        int cur_g = t0 / tpg;                      // the first tile's bank: on its way while the movers fetch that tile
        bload(cur_g);
        int qofs[2];                               // where this thread's 16-byte pieces of a finished tile start in the partial tiles
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int e = (ctid + 256 * it) * 4, row = e / N;
            qofs[it] = row * RW + (e - row * N);
        }
        const int arow = lane & 31, acol = 8 * (lane >> 5);
        for (int j = 0; j < nturns; j++) {         // turn j: tile j - 1 out of buffer (j - 1) & 1
            const int jt = j - 1;
            const bool work = jt >= 0 && jt < nmine;
            RG16_TS(1, j, 0);
            const int tile = t0 + (work ? jt : 0), g = tile / tpg, r0 = (tile - g * tpg) * 32;
            if (work) {
                if (g != cur_g) {                  // (never when the bank is shared)
                    bload(g);
                    cur_g = g;
                }
                f32x16 accM[NCT], accS[NCT];
                const uint16_t* ap = ldsr + (jt & 1) * bufsz + arow * RS + acol;
                {                                  // the first k-step (cw < 4 <= KT) starts the sums from the constant 0
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const f16x8v Ah = __builtin_bit_cast(f16x8v, *(const uint4*)(ap + 16 * cw));
                    const f16x8v Al = __builtin_bit_cast(f16x8v, *(const uint4*)(ap + 16 * cw + plane));
#pragma unroll
                    for (int ct = 0; ct < NCT; ct++) {
                        const f16x8v Bh = __builtin_bit_cast(f16x8v, bfr[0][2 * ct]), Bl = __builtin_bit_cast(f16x8v, bfr[0][2 * ct + 1]);
                        accM[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, zero, 0, 0, 0);
                        accS[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, zero, 0, 0, 0);
                        accS[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, accS[ct], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int k = 1; k < KS; k++) {
                    const int t = cw + 4 * k;
                    if (t < KT) {
                        const f16x8v Ah = __builtin_bit_cast(f16x8v, *(const uint4*)(ap + 16 * t));
                        const f16x8v Al = __builtin_bit_cast(f16x8v, *(const uint4*)(ap + 16 * t + plane));
#pragma unroll
                        for (int ct = 0; ct < NCT; ct++) {
                            const f16x8v Bh = __builtin_bit_cast(f16x8v, bfr[k][2 * ct]), Bl = __builtin_bit_cast(f16x8v, bfr[k][2 * ct + 1]);
                            accM[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, accM[ct], 0, 0, 0);
                            accS[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, accS[ct], 0, 0, 0);
                            accS[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, accS[ct], 0, 0, 0);
                        }
                    }
                }
                // register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5): rows of 8-row group r >> 2, scaled back by that group's factor
                float iA[4];
#pragma unroll
                for (int q = 0; q < 4; q++) iA[q] = __uint_as_float((uint32_t)(127 - sexp[jt & 1][q]) << 23);
                float* red = red0 + (j & 1) * (128 * RW);
#pragma unroll
                for (int ct = 0; ct < NCT; ct++)
                    if (ct * 32 + (lane & 31) < N) {
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            red[(cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * RW + ct * 32 + (lane & 31)] = ((accM[ct][r] + accS[ct][r]) * iA[r >> 2]) * iB;
                    }
            }
            RG16_TS(1, j, 1);
            __syncthreads();                       // the one barrier of a turn: tile j is in its planes, tile j - 1's partial sums in theirs
            RG16_TS(1, j, 2);
            const float* red = red0 + (j & 1) * (128 * RW);
            if (work) {
                const int nrow = min(32, rpg - r0);
                float* Cs = C + ((size_t)g * rpg + r0) * N;
                const int total = nrow * N;        // one contiguous span of the output
                if ((N & 3) == 0 && (((uintptr_t)Cs) & 15) == 0) {
#pragma unroll
                    for (int it = 0; it < 2; it++) {   // (32 N / 4 <= 512 pieces of 16 bytes)
                        const int e = (ctid + 256 * it) * 4;
                        if (e >= total) break;
                        const float* q = red + qofs[it];
                        const float4 p0 = *(const float4*)q, p1 = *(const float4*)(q + 32 * RW), p2 = *(const float4*)(q + 64 * RW), p3 = *(const float4*)(q + 96 * RW);
                        float4 o = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                                               (p0.w + p1.w) + (p2.w + p3.w));
                        float4* dst = (float4*)(Cs + e);
                        if (acc) {
                            const float4 t = *dst;
                            o.x += t.x, o.y += t.y, o.z += t.z, o.w += t.w;
                        }
                        *dst = o;
                    }
                } else {
                    for (int e = ctid; e < total; e += 256) {
                        const int row = e / N, col = e - row * N;
                        const float* q = red + row * RW + col;
                        const float o = (q[0] + q[32 * RW]) + (q[64 * RW] + q[96 * RW]);
                        Cs[e] = acc ? Cs[e] + o : o;
                    }
                }
            }
            RG16_TS(1, j, 3);
        }
    }
}
#ifdef RG16_TIMING
extern "C" int motifs_debug_rg16_ts(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rg16_ts), sizeof(g_rg16_ts)); }
#endif
static bool launch_rowgemm16(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& rg, int acc, int groups, long rpg, int tpg) {
    const bool f32_only = gemm_f32_only();     // A/B: the float32 matrix instruction
    const long min_tiles = gemm_f16_min(768);       // measured: -1 % at 576 tiles, +1 % at 864
    const long ntiles = (long)groups * tpg;
    if (f32_only || ntiles < min_tiles || ntiles > (1l << 30) || rpg > (1l << 30)) return false;
    const int KT = rg.Q / 16, NCT = (rg.N + 31) / 32;
    const size_t lds = (size_t)2 * 2 * 32 * (rg.Q + 8) * 2 + (size_t)2 * 4 * 32 * (((rg.N + 15) & ~15) + 8) * 4;
    if (lds + 64 > 160 * 1024) return false;
    const size_t perf16 = (size_t)NCT * KT * 128;                     // uint4 per bank
    bool fresh16;
    float* Bf16 = e.relayout(Bm, 8, rg.Q, rg.N, 0, perf16 * groups * 4 + ((groups + 3) & ~3), fresh16);
    if (!Bf16) return true;
    uint32_t* bm = (uint32_t*)(Bf16 + perf16 * groups * 4);            // a scale per group behind the fragments
    if (fresh16) hipLaunchKernelGGL(k_frag_bw16, dim3(groups * FRAG_SPLIT), dim3(256), 0, e.st, Bm, rg.Q, rg.N, bm, (uint4*)Bf16);
    const long tpb = (ntiles + 255) / 256;
    const dim3 grid((unsigned)((ntiles + tpb - 1) / tpb));            // one block per CU, there for the whole launch, each with a run of tpb tiles
    const int64_t ldbf = rg.ldb == 0 ? 0 : (int64_t)perf16;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, e.st, A, (const uint4*)Bf16, C, (int)rpg, tpg, (int)ntiles, rg.Q, rg.N, ldbf, acc, bm);
    };
    const int KS = (KT + 3) / 4;                   // k-steps per multiplier wave; 16-byte loads per mover lane: Q / 32
    if (NCT == 1 && rg.Q <= 416) go(k_rowgemm16<1, 7, 13>);
    else if (NCT == 1) go(k_rowgemm16<1, 8, 15>);
    else if (rg.Q <= 416) go(k_rowgemm16<2, 7, 13>);
    else go(k_rowgemm16<2, 8, 15>);
    (void)KS;
    return true;
}
template <int KT>
__global__ __launch_bounds__(256) void k_toep_wide16(const float* __restrict__ A, const uint4* __restrict__ Bf, float* __restrict__ C, ToepGeom gm,
                                                     int acc, int tps, int64_t ldbf, int SO, const uint32_t* __restrict__ bmax) {
    extern __shared__ float lds[];                 // the block's 32 output rows, [32][SO]
    __shared__ uint32_t wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int job = blockIdx.x;
    const int s = job / tps, p0 = (job - s * tps) * 32;
    const float* sig = A + (size_t)s * gm.lda;
    const uint4* bp = Bf + (size_t)(s / gm.B) * ldbf + lane;
    const int nct = (gm.N + 31) >> 5;
    uint4 cur[2 * KT], nxt[2 * KT];
    if (wave < nct) {
#pragma unroll
        for (int q = 0; q < 2 * KT; q++) cur[q] = bp[(size_t)(wave * 2 * KT + q) * 64];
    }
    // the block's windows: their span (31 sa + Q floats) through LDS once, and its largest magnitude on the way
    const int span = 31 * gm.sa + 16 * KT, e0 = gm.a0 + p0 * gm.sa;
    uint32_t m = 0;
    for (int i = tid; i < span; i += 256) {
        const int e = e0 + i;
        const bool ok = e >= 0 && e < gm.amax;
        const float x = sig[ok ? e : 0];
        lds[i] = ok ? x : 0.0f;
        m = max(m, ok ? __float_as_uint(x) & 0x7fffffffu : 0u);
    }
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    const int seA = f16x3_scale_exp(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]))), seB = f16x3_scale_exp(bmax[ldbf ? s / gm.B : 0]);
    const float sA = __uint_as_float((uint32_t)(seA + 127) << 23);
    f16x8v ah[KT], al[KT];
    {
        const float* w = lds + (lane & 31) * gm.sa + 8 * (lane >> 5);
#pragma unroll
        for (int t = 0; t < KT; t++)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float x = w[16 * t + u] * sA;
                const _Float16 hi = (_Float16)x;
                ah[t][u] = hi;
                al[t][u] = (_Float16)(x - (float)hi);
            }
    }
    __syncthreads();                               // the tile stores below reuse the space
    const float iA = __uint_as_float((uint32_t)(127 - seA) << 23), iB = __uint_as_float((uint32_t)(127 - seB) << 23);
    for (int ct = wave; ct < nct; ct += 4) {       // wave w: column tiles w, w+4, ..
        const int cn = min(ct + 4, nct - 1);
#pragma unroll
        for (int q = 0; q < 2 * KT; q++) nxt[q] = bp[(size_t)(cn * 2 * KT + q) * 64];
        f32x16 accM, accS;
#pragma unroll
        for (int i = 0; i < 16; i++) accM[i] = 0.0f, accS[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < KT; t++) {
            const f16x8v Bh = __builtin_bit_cast(f16x8v, cur[2 * t]), Bl = __builtin_bit_cast(f16x8v, cur[2 * t + 1]);
            accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], Bh, accM, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], Bl, accS, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], Bh, accS, 0, 0, 0);
        }
        const int col = ct * 32 + (lane & 31);
        if (col < gm.N) {
#pragma unroll
            for (int r = 0; r < 16; r++) lds[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SO + col] = ((accM[r] + accS[r]) * iA) * iB;
        }
#pragma unroll
        for (int q = 0; q < 2 * KT; q++) cur[q] = nxt[q];
    }
    __syncthreads();
    // the rows leave whole: [nrow][N] is one contiguous span of the output
    const int nrow = min(32, gm.P - p0), N4 = gm.N >> 2;
    float* Cs = C + (size_t)s * gm.ldc + (size_t)p0 * gm.N;
    if ((gm.N & 3) == 0 && (((uintptr_t)Cs) & 15) == 0) {
        int row = tid / N4, c4 = tid - row * N4;
        const int drow = 256 / N4, dc = 256 - drow * N4;
        for (int idx = tid; idx < nrow * N4; idx += 256) {
            float4 o = *(const float4*)(lds + row * SO + c4 * 4);
            float4* dst = (float4*)Cs + idx;
            if (acc) {
                const float4 t = *dst;
                o.x += t.x, o.y += t.y, o.z += t.z, o.w += t.w;
            }
            *dst = o;
            row += drow, c4 += dc;
            if (c4 >= N4) c4 -= N4, row++;
        }
    } else {
        for (int idx = tid; idx < nrow * gm.N; idx += 256) {
            const int row = idx / gm.N, col = idx - row * gm.N;
            const float o = lds[row * SO + col];
            Cs[idx] = acc ? Cs[idx] + o : o;
        }
    }
}
static bool launch_toep_wide(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& gm, int acc) {
    if (gm.N < 64 || gm.N > 480 || gm.Q > 64 || (gm.Q & 7) || gm.Q < 32) return false;
    const int KG = gm.Q / 8, NCT = (gm.N + 31) / 32;
    const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
    const size_t perf = (size_t)NCT * KG * 256;
    bool fresh;
    float* Bf = e.relayout(Bm, 3, gm.Q, gm.N, 0, perf * gB, fresh);
    if (!Bf) return true;
    if (fresh) hipLaunchKernelGGL(k_frag_bw, dim3(nblocks(perf * gB)), dim3(256), 0, e.st, Bm, gB, gm.Q, gm.N, Bf);
    const int tps = (gm.P + 31) / 32;
    const bool f32_only = gemm_f32_only();   // A/B: the float32 matrix instruction
    const long min_jobs16 = gemm_f16_min(768);   // (tests: 1; measured: -2 % at 576 jobs, +2 % at 864)
    const bool want16 = !f32_only && (long)gm.S * tps >= min_jobs16 && gm.Q % 16 == 0 && gm.sa > 0;
    const int cs = !want16 && (long)gm.S * tps < 256 ? std::min(4, NCT) : 1;      // few reads: four blocks per row tile, a wave per column tile
    const dim3 grid((unsigned)((long)gm.S * tps * cs));
    const int SO = ((gm.N + 15) & ~15) + 8;        // 4 rows apart = 32 banks apart: the two lane halves of a tile store never meet
    const size_t lds = (size_t)32 * SO * 4;
    const int64_t ldbf = gm.ldb == 0 ? 0 : (int64_t)perf;
    if (want16 && 31 * gm.sa + gm.Q <= 32 * SO) {
        const int KT = gm.Q / 16;
        const size_t perf16 = (size_t)NCT * KT * 128;                 // uint4 per bank
        bool fresh16;
        float* Bf16 = e.relayout(Bm, 8, gm.Q, gm.N, 0, perf16 * gB * 4 + ((gB + 3) & ~3), fresh16);
        if (!Bf16) return true;
        uint32_t* bm = (uint32_t*)(Bf16 + perf16 * gB * 4);            // a scale per group behind the fragments
        if (fresh16) hipLaunchKernelGGL(k_frag_bw16, dim3(gB * FRAG_SPLIT), dim3(256), 0, e.st, Bm, gm.Q, gm.N, bm, (uint4*)Bf16);
        const int64_t ldbf16 = gm.ldb == 0 ? 0 : (int64_t)perf16;
#define TOEPWIDE16(K) hipLaunchKernelGGL((k_toep_wide16<K>), grid, dim3(256), lds, e.st, A, (const uint4*)Bf16, C, gm, acc, tps, ldbf16, SO, bm)
        if (KT == 2) TOEPWIDE16(2);
        else if (KT == 3) TOEPWIDE16(3);
        else TOEPWIDE16(4);
#undef TOEPWIDE16
        return true;
    }
#define TOEPWIDE(K) hipLaunchKernelGGL((k_toep_wide<K>), grid, dim3(256), lds, e.st, A, Bf, C, gm, acc, tps, ldbf, SO, cs)
    if (KG == 4) TOEPWIDE(4);
    else if (KG == 5) TOEPWIDE(5);
    else if (KG == 6) TOEPWIDE(6);
    else if (KG == 7) TOEPWIDE(7);
    else TOEPWIDE(8);
#undef TOEPWIDE
    return true;
}

static bool is_tall(const ToepGeom& gm) {
    if (gm.N > 8 || gm.sa < 64 || gm.Q % gm.sa != 0 || gm.amax % gm.sa != 0 || gm.a0 % gm.sa != 0) return false;
    const int H = gm.Q / gm.sa;
    return H >= 2 && H * gm.N <= 64 && H * gm.N > 8;
}
// the row GEMM behind both tall forms: rows = image rows, reduction = one row (sa), outputs = H*N
static ToepGeom tall_row_geom(const ToepGeom& gm) {
    const int H = gm.Q / gm.sa, R = gm.amax / gm.sa;
    ToepGeom r;
    r.S = gm.S;
    r.P = R;
    r.Q = gm.sa;
    r.N = H * gm.N;
    r.sa = gm.sa;
    r.a0 = 0;
    r.amax = gm.amax;
    r.lda = gm.lda;
    r.ldc = (int64_t)R * H * gm.N;
    r.B = gm.B;
    r.ldb = gm.ldb == 0 ? 0 : (int64_t)gm.Q * gm.N;
    return r;
}

static void launch_toep(Engine& e, const float* A, const float* Bm, float* C, const ToepGeom& gm, int acc, const float* y = nullptr, float yb = 0.0f) {
    hipStream_t st = e.st;
    if (is_tall(gm)) {
        const int H = gm.Q / gm.sa, R = gm.amax / gm.sa;
        const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
        const size_t per = (size_t)gm.Q * gm.N;
        bool fresh;
        float* Bt = e.relayout(Bm, 4, gm.sa, gm.N, H, per * gB, fresh);
        float* Wt = e.arena.alloc((size_t)gm.S * R * H * gm.N);
        if (!Bt || !Wt) {
            e.failed = true;
            return;
        }
        if (fresh) hipLaunchKernelGGL(k_tall_bt, dim3(nblocks(per * gB)), dim3(256), 0, st, Bm, gB, H, gm.sa, gm.N, Bt);
        const ToepGeom rg = tall_row_geom(gm);
        if (launch_tall_fused(e, A, Bt, C, gm, rg, acc, y, yb)) return;
        if (!launch_rowgemm_lds(e, A, Bt, Wt, rg, 0)) {
            const int grp_rows = rg.B * rg.P;
            hipLaunchKernelGGL(k_toep_mfma, dim3((unsigned)((grp_rows + 63) / 64), (unsigned)(rg.S / rg.B), (unsigned)((rg.N + 31) / 32)),
                               dim3(512), 0, st, A, Bt, Wt, rg, 0);
        }
        hipLaunchKernelGGL(k_tall_gather, dim3(nblocks((size_t)gm.S * gm.P * gm.N)), dim3(256), 0, st, Wt, C, gm.S, gm.P, H, gm.N, R,
                           gm.a0 / gm.sa, gm.ldc, acc, y, yb);
        return;
    }
    if (launch_ana_f16x3<12, 80>(e, A, Bm, C, gm, acc) || launch_ana_f16x3<12, 64>(e, A, Bm, C, gm, acc) || launch_ana_lds<12, 80>(e, A, Bm, C, gm, acc) || launch_ana_lds<12, 64>(e, A, Bm, C, gm, acc) ||
               launch_ana_lds<12, 32>(e, A, Bm, C, gm, acc) || launch_ana_lds<8, 64>(e, A, Bm, C, gm, acc) ||
               launch_ana_lds<8, 32>(e, A, Bm, C, gm, acc)) {
        return;
    }
    // MFMA form: narrow output, long reduction (tiles never straddle two filter groups)
    if (gm.N > 8 && gm.N <= 64 && gm.Q >= 256) {
        const int grp_rows = gm.B * gm.P;
        hipLaunchKernelGGL(k_toep_mfma, dim3((unsigned)((grp_rows + 63) / 64), (unsigned)(gm.S / gm.B), (unsigned)((gm.N + 31) / 32)),
                           dim3(512), 0, st, A, Bm, C, gm, acc);
        return;
    }
    if (gm.N <= 4 && gm.Q >= 256) {
        const int grp_rows = gm.B * gm.P;
        hipLaunchKernelGGL(k_toep_n4, dim3((unsigned)((grp_rows + 127) / 128), (unsigned)(gm.S / gm.B)), dim3(512), 0, st, A, Bm, C,
                           gm, acc);
        return;
    }
    if (launch_toep_wide(e, A, Bm, C, gm, acc)) return;
    if (gm.N <= 32) {
        dim3 grid((gm.N + 31) / 32, (gm.P + 63) / 64, gm.S);
        hipLaunchKernelGGL(k_toep<32>, grid, dim3(256), 0, st, A, Bm, C, gm, acc);
    } else {
        dim3 grid((gm.N + 63) / 64, (gm.P + 63) / 64, gm.S);
        hipLaunchKernelGGL(k_toep<64>, grid, dim3(256), 0, st, A, Bm, C, gm, acc);
    }
}

// dB[g][q][n] (+)= sum_{s in g} sum_p Aw(s,p,q) * C[s][p][n]
template <int BN>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ A, const float* __restrict__ C,
                                               float* __restrict__ dB, ToepGeom gm, int acc) {
    constexpr int BM = 64, BK = 16, TM = 4, TN = BN / 16;
    __shared__ float As[BK][BM + 4];   // [sp][q]
    __shared__ float Cs[BK][BN];       // [sp][n]
    const int g = blockIdx.z, q0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    const int KT = gm.B * gm.P;        // reduction length: (sequence in group, position)
    float accv[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) accv[i][j] = 0.0f;
    for (int k0 = 0; k0 < KT; k0 += BK) {
#pragma unroll
        for (int it = 0; it < (BM * BK) / 256; it++) {
            const int idx = tid + it * 256;
            const int kk = idx / BM, qq = idx % BM;       // contiguous in q
            const int k = k0 + kk, q = q0 + qq;
            const int kc = k < KT ? k : 0;
            const int sl = kc / gm.P, p = kc - sl * gm.P;
            const int e = gm.a0 + p * gm.sa + q;
            const bool ok = k < KT && q < gm.Q && e >= 0 && e < gm.amax;
            const float v = A[(size_t)(g * gm.B + sl) * gm.lda + (ok ? e : 0)];
            As[kk][qq] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < (BK * BN + 255) / 256; it++) {
            const int idx = tid + it * 256;
            if (idx < BK * BN) {
                const int kk = idx / BN, nn = idx % BN;
                const int k = k0 + kk, n = n0 + nn;
                const bool ok = k < KT && n < gm.N;
                const int kc = ok ? k : 0;
                const int sl = kc / gm.P, p = kc - sl * gm.P;
                const float v = C[(size_t)(g * gm.B + sl) * gm.ldc + (size_t)p * gm.N + (ok ? n : 0)];
                Cs[kk][nn] = ok ? v : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk++) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = As[kk][ty * TM + i];
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = Cs[kk][tx * TN + j];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) accv[i][j] = fmaf(a[i], b[j], accv[i][j]);
        }
        __syncthreads();
    }
    float* out = dB + (size_t)g * gm.Q * gm.N;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int q = q0 + ty * TM + i;
        if (q >= gm.Q) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + tx * TN + j;
            if (n < gm.N) {
                float* o = &out[(size_t)q * gm.N + n];
                *o = acc ? *o + accv[i][j] : accv[i][j];
            }
        }
    }
}

// y[g][j] (+)= sum of the B consecutive slices x[g*B .. g*B + B) (per-sequence partials -> per-group sums)
// The filter gradient of the row GEMM: part[s][q][n] = sum_r A[s][r][q] * C[s][r][n] over the R contiguous rows of
// sequence s (Q floats of A, N <= 48 of C per row).  HBM-bound: A is read once.  A block walks its sequence 32
// rows at a time (rows to LDS with 16-byte accesses, the next 32 already in flight in registers); wave w owns the
// 16-row blocks w, w+4, .. of q for all column blocks (v_mfma_f32_16x16x4_f32, the accumulators stay in registers
// for the whole sequence).  Row strides of both LDS tiles are 16 mod 32 floats: conflict-free operand reads.
// Where the C rows come from.  mode 0: rows of N floats in memory.  mode 1: the scatter of a tall form,
// C[s][rho][ip][n] = src[s][rho - ip - off][n] (n < 4 n4s; zero outside 0 <= row < P), and mode 2: the window matrix of a
// Toeplitz operand, C[s][p][q] = src[s][a0 + p sa + q] (zero outside [0, amax)) - both were kernels of their own
// (k_tall_scatter, k_windows) that wrote what this one then read.
struct RowSrc {
    int mode, n4s, off, P, a0, sa, amax;
    int64_t ld;
};
template <int QW, int NW>
__global__ __launch_bounds__(NW * 64) void k_rowwgrad_lds(const float* __restrict__ A, const float* __restrict__ C, float* __restrict__ part,
                                                      int R, int Q, int N, int ST, int SN, int TS, RowSrc cs) {
    constexpr int NT = 3, TH = NW * 64, NVA = (32 * 120 + TH - 1) / TH, NVC = (32 * 16 + TH - 1) / TH;   // 16-byte loads per thread: 32 x Q (Q <= 480) of A, 32 x N (N <= 64) of C
    extern __shared__ float lds[];                 // A tile [32][ST], C tile [32][SN]
    float* Cs = lds + 32 * ST;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // TS blocks share a sequence (block ts walks the row tiles ts, ts + TS, ..): with the reference's 6 reads per step one
    // block per read leaves the chip idle behind a serial walk of R / 32 tiles; the TS partial banks are summed with the
    // per-read ones
    const int s = blockIdx.x / TS, ts = blockIdx.x - s * TS, Q4 = Q >> 2, N4 = N >> 2, QB = Q >> 4;
    const float4* Ag = (const float4*)(A + (size_t)s * R * Q);
    const float4* Cg = (const float4*)(C + (cs.mode == 0 ? (size_t)s * R * N : (size_t)s * cs.ld));
    float4 va[NVA], vc[NVC];
    auto gload = [&](int r0) {
        const int na = min(32, R - r0) * Q4, nc = min(32, R - r0) * N4;
#pragma unroll
        for (int i = 0; i < NVA; i++) {
            const int idx = tid + i * TH;
            const float4 x = Ag[(size_t)r0 * Q4 + (idx < na ? idx : 0)];
            va[i] = idx < na ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NVC; i++) {
            const int idx = tid + i * TH;
            bool ok = idx < nc;
            size_t at = (size_t)r0 * N4 + (ok ? idx : 0);
            if (cs.mode != 0) {
                const int r = idx / N4, c = idx - r * N4;
                if (cs.mode == 1) {
                    const int ip = c / cs.n4s, row = r0 + r - ip - cs.off;
                    ok = ok && row >= 0 && row < cs.P;
                    at = (size_t)row * cs.n4s + (c - ip * cs.n4s);
                } else {
                    const int e = cs.a0 + (r0 + r) * cs.sa + 4 * c;
                    ok = ok && e >= 0 && e + 3 < cs.amax;
                    at = (size_t)(e >> 2);
                }
                if (!ok) at = 0;
            }
            const float4 x = Cg[at];
            vc[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
        int row = tid / Q4, c4 = tid - row * Q4;
        const int drow = TH / Q4, dc = TH - drow * Q4;
#pragma unroll
        for (int i = 0; i < NVA; i++) {
            if (tid + i * TH < 32 * Q4) *(float4*)(lds + row * ST + c4 * 4) = va[i];
            row += drow, c4 += dc;
            if (c4 >= Q4) c4 -= Q4, row++;
        }
#pragma unroll
        for (int i = 0; i < NVC; i++) {
            const int idx = tid + i * TH;
            if (idx < 32 * N4) {
                const int r = idx / N4, c = idx - r * N4;
                *(float4*)(Cs + r * SN + c * 4) = vc[i];
            }
        }
    };
    f32x4 accv[QW][NT];
#pragma unroll
    for (int i = 0; i < QW; i++)
#pragma unroll
        for (int cb = 0; cb < NT; cb++) accv[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // columns N .. 16 NT - 1 of the C tile stay zero
    for (int i = tid; i < 32 * SN; i += TH) Cs[i] = 0.0f;
    const float* ap = lds + (lane >> 4) * ST + (lane & 15) + wave * 16;     // A'[q][r] = tile[r][q]
    const float* cp = Cs + (lane >> 4) * SN + (lane & 15);
    if (ts * 32 < R) gload(ts * 32);
    for (int r0 = ts * 32; r0 < R; r0 += 32 * TS) {
        __syncthreads();                           // the previous tile has been read
        lstore();
        __syncthreads();
        if (r0 + 32 * TS < R) gload(r0 + 32 * TS);
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            float bv[NT];
#pragma unroll
            for (int cb = 0; cb < NT; cb++) bv[cb] = cp[ks * 4 * SN + cb * 16];
#pragma unroll
            for (int i = 0; i < QW; i++) {
                if (wave + NW * i < QB) {
                    const float av = ap[ks * 4 * ST + i * (16 * NW)];
#pragma unroll
                    for (int cb = 0; cb < NT; cb++) accv[i][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[cb], accv[i][cb], 0, 0, 0);
                }
            }
        }
    }
    float* out = part + ((size_t)s * TS + ts) * Q * N;
#pragma unroll
    for (int i = 0; i < QW; i++) {
        if (wave + NW * i < QB) {
#pragma unroll
            for (int cb = 0; cb < NT; cb++) {
                const int n = cb * 16 + (lane & 15);
                if (n < N) {
#pragma unroll
                    for (int r = 0; r < 4; r++) out[(size_t)((wave + NW * i) * 16 + 4 * (lane >> 4) + r) * N + n] = accv[i][cb][r];
                }
            }
        }
    }
}
// ---- the same filter gradient on the binary16 matrix instruction, three products per term, waves in two roles (see k_rowgemm16) ----
// The float32 kernel above is bound by the matrix pipe where two of its blocks share a CU (384 reads over 256 CUs: 7 200 instructions of 32
// cycles on four pipes = 24 us of its 48).  Here a GROUP's B R rows are cut into tiles of 32 and dealt to nbg blocks, one block of 8 waves per CU:
// waves 0-3 move (8 rows of the A tile and of the C tile each: global -> registers -> hi / lo planes in LDS, ROW-major, so the loads stay
// 16-byte coalesced), waves 4-7 multiply: the operands of part[q][n] = sum_r A[r][q] C[r][n] run down the rows, and ds_read_b64_tr_b16 hands
// a lane 4 rows of one column (tools/ubench/tr_read_probe.hip) - two such reads are the 8 consecutive k of a 16x16x32 operand.  The sums of a
// block stay in its multipliers' registers over all its tiles (25 x 3 tiles of 16 x 16 over four waves) and leave once, as one partial bank
// per block.  Scales: the reduction runs ACROSS rows, so a tile's rows share one power of two per operand - the largest magnitude the block has
// seen so far, pushed a turn ahead (LDS atomic max into the slot of the tile's turn parity, so that the turn's one barrier lies between the
// push and the read); when it grows the multipliers rescale their sums (factors <= 1).  48.7 -> 36 us per launch at 64 mini-batches.
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
template <int QW, int NV>
__global__ __launch_bounds__(512) void k_rowwgrad16(const float* __restrict__ A, const float* __restrict__ C, float* __restrict__ part, int R, int Q, int N,
                                                    int tps, int B, int nbg, int tpb, RowSrc cs) {
    constexpr int CS = 56, NB = 3;                 // C planes' row stride (halves); 16-column blocks of the 48 columns
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsw[];        // 2 x (A hi, lo [32][RS]) halves, 2 x (C hi, lo [32][CS])
    __shared__ uint32_t wmax[2][2];                // (turn parity) -> the largest |A|, |C| this block has met up to that turn's tile (bits)
    __shared__ int sexp[2][2];                     // (buffer) -> scale exponents of A and C
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RS = Q + 8, Q4 = Q >> 2, QB = Q >> 4, N4 = N >> 2;
    const int planeA = 32 * RS, bufA = 2 * planeA, planeC = 32 * CS, bufC = 2 * planeC;
    uint16_t* ldsC = ldsw + 2 * bufA;
    const int g = blockIdx.x / nbg, piece = blockIdx.x - g * nbg;
    const int T = B * tps, t0 = piece * tpb, nmine = max(0, min(T, t0 + tpb) - t0);
    const int nturns = (nmine + 2) / 2 * 2;        // nmine + 1 turns, rounded up to the movers' two register sets
    if (tid < 4) wmax[tid >> 1][tid & 1] = 0;
    __syncthreads();
    if (wave < 4) {
        // ---- movers ----
        float4 v[2][NV], vc[2][2];                 // two tiles on their way (register sets)
        const int nf8 = 8 * Q4, nc8 = 8 * N4;
        int lofs[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / Q4, c4 = idx - row * Q4;
            lofs[i] = idx < nf8 ? row * RS + c4 * 4 : Q;       // (pieces past the 8 rows: the 8 spare halves behind row 0)
        }
        int crow[2], cc4[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = lane + i * 64;
            crow[i] = idx / N4, cc4[i] = idx - crow[i] * N4;
        }
        // the same NV + 2 loads on every path (a turn past the block's last tile re-reads the first 16 bytes of A and of C): the wait for one
        // tile's registers then leaves the other tile's loads in flight
        auto gload = [&](float4* va, float4* vcs, int j) -> int {
            int okm = 0;
            const bool live = j < nmine;
            const int tt = t0 + (live ? j : 0), sl = tt / tps, r0 = (tt - sl * tps) * 32 + 8 * wave;
            const size_t s = (size_t)g * B + sl;
            // A: rows past the read's end re-read its last valid 16 bytes (their C rows are zeros)
            const int nf4 = live ? max(0, min(8, R - r0)) * Q4 : 0, last = max(nf4, 1) - 1;
            const float4* At = (const float4*)(A + (nf4 > 0 ? (s * R + r0) * Q : (size_t)0));
#pragma unroll
            for (int i = 0; i < NV; i++) va[i] = At[min(lane + i * 64, last)];
            const float4* Cg = (const float4*)(C + (cs.mode == 0 ? s * R * N : s * (size_t)cs.ld));
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int r = r0 + crow[i], c = cc4[i];
                bool ok = live && lane + i * 64 < nc8 && r < R;
                size_t at = (size_t)r * N4 + c;
                if (cs.mode == 1) {
                    const int ip = c / cs.n4s, row = r - ip - cs.off;
                    ok = ok && row >= 0 && row < cs.P;
                    at = (size_t)row * cs.n4s + (c - ip * cs.n4s);
                } else if (cs.mode == 2) {
                    const int e = cs.a0 + r * cs.sa + 4 * c;
                    ok = ok && e >= 0 && e + 3 < cs.amax;
                    at = (size_t)(e >> 2);
                }
                vcs[i] = Cg[ok ? at : 0];          // (what is not ok is zeroed when the tile is converted, not here: nothing waits on a load yet)
                okm |= ok ? 1 << i : 0;
            }
            return okm;
        };
        // the largest magnitudes of the tile in a register set, pushed (with everything before it: `sofar`) into the slot of its turn's parity:
        // every mover reads a turn's slot after the barrier in front of that turn and nobody writes it during the turn, so the four waves
        // convert a tile with one scale
        auto push_max = [&](const float4* va, const float4* vcs, int okm, int j, uint32_t sofarA, uint32_t sofarC) {
            float ma = 0.0f, mc = 0.0f;
#pragma unroll
            for (int i = 0; i < NV; i++) ma = fmaxf(fmaxf(fmaxf(fmaxf(ma, fabsf(va[i].x)), fabsf(va[i].y)), fabsf(va[i].z)), fabsf(va[i].w));
#pragma unroll
            for (int i = 0; i < 2; i++)
                if (okm >> i & 1) mc = fmaxf(fmaxf(fmaxf(fmaxf(mc, fabsf(vcs[i].x)), fabsf(vcs[i].y)), fabsf(vcs[i].z)), fabsf(vcs[i].w));
            uint32_t ua = max(__float_as_uint(ma), sofarA), uc = max(__float_as_uint(mc), sofarC);
            for (int d = 32; d >= 1; d >>= 1) ua = max(ua, (uint32_t)__shfl_xor((int)ua, d)), uc = max(uc, (uint32_t)__shfl_xor((int)uc, d));
            if (lane == 0) {
                atomicMax(&wmax[j & 1][0], ua);
                atomicMax(&wmax[j & 1][1], uc);
            }
        };
        int okc[2];
        okc[0] = gload(v[0], vc[0], 0);
        okc[1] = gload(v[1], vc[1], 1);
        if (nmine > 0) push_max(v[0], vc[0], okc[0], 0, 0u, 0u);
        __syncthreads();
        for (int j2 = 0; j2 < nturns; j2 += 2) {   // turn j: tile j into buffer j & 1 (the multipliers are on tile j - 1)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int j = j2 + u;
                float4* va = v[u];
                float4* vcs = vc[u];
                const uint32_t mA = wmax[u][0], mC = wmax[u][1];
                if (j < nmine) {
                    const int seA = f16x3_scale_exp(mA), seC = f16x3_scale_exp(mC);
                    if (tid == 0) sexp[u][0] = seA, sexp[u][1] = seC;
                    const float sA = __uint_as_float((uint32_t)(seA + 127) << 23), sC = __uint_as_float((uint32_t)(seC + 127) << 23);
                    uint16_t* buf = ldsw + u * bufA + 8 * wave * RS;
#pragma unroll
                    for (int i = 0; i < NV; i++) {
                        f32x2v p0 = {va[i].x, va[i].y}, p1 = {va[i].z, va[i].w};
                        p0 *= sA, p1 *= sA;
                        const f16x2v h0 = __builtin_convertvector(p0, f16x2v), h1 = __builtin_convertvector(p1, f16x2v);
                        const f32x2v r0 = p0 - __builtin_convertvector(h0, f32x2v), r1 = p1 - __builtin_convertvector(h1, f32x2v);
                        const f16x2v l0 = __builtin_convertvector(r0, f16x2v), l1 = __builtin_convertvector(r1, f16x2v);
                        uint16_t* d = buf + lofs[i];
                        *(uint2*)d = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
                        *(uint2*)(d + planeA) = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
                    }
                    uint16_t* cbuf = ldsC + u * bufC + 8 * wave * CS;
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        if (lane + i * 64 < nc8) {
                            const float sz = (okc[u] >> i & 1) ? sC : 0.0f;
                            f32x2v p0 = {vcs[i].x, vcs[i].y}, p1 = {vcs[i].z, vcs[i].w};
                            p0 *= sz, p1 *= sz;
                            const f16x2v h0 = __builtin_convertvector(p0, f16x2v), h1 = __builtin_convertvector(p1, f16x2v);
                            const f32x2v r0 = p0 - __builtin_convertvector(h0, f32x2v), r1 = p1 - __builtin_convertvector(h1, f32x2v);
                            const f16x2v l0 = __builtin_convertvector(r0, f16x2v), l1 = __builtin_convertvector(r1, f16x2v);
                            uint16_t* d = cbuf + crow[i] * CS + cc4[i] * 4;
                            *(uint2*)d = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
                            *(uint2*)(d + planeC) = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
                        }
                    }
                }
                okc[u] = gload(va, vcs, j + 2);    // in flight over this barrier and the other tile's turn
                if (j + 1 < nmine) push_max(v[u ^ 1], vc[u ^ 1], okc[u ^ 1], j + 1, mA, mC);    // (requested a turn ago)
                __syncthreads();
            }
        }
    } else {
        // ---- multipliers ----
        const int cw = wave - 4;
        f32x4 acc[QW][NB];
#pragma unroll
        for (int k = 0; k < QW; k++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[k][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        int curA = 0, curC = 0;
        bool scaled = false;
        // a lane's piece of a transposed read: row 8 (lane >> 4) + ((lane & 15) >> 2) of the block, columns 4 (lane & 3) .. + 3
        const int trow = 8 * (lane >> 4) + ((lane & 15) >> 2), tcol = 4 * (lane & 3);
        const int aoff = trow * RS + tcol + 16 * cw, coff = trow * CS + tcol;
        typedef __attribute__((address_space(3))) h16x4* lds_h4;
        auto tr8 = [&](const uint16_t* p, int rowstride) -> f16x8v {      // 8 consecutive rows of one column: two transposed reads
            const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(p)), hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(p + 4 * rowstride));
            const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
            return __builtin_bit_cast(f16x8v, make_uint4(a.x, a.y, b.x, b.y));
        };
        __syncthreads();                           // (the movers' first maxima)
        for (int j = 0; j < nturns; j++) {         // turn j: tile j - 1 out of buffer (j - 1) & 1
            const int jt = j - 1, b = jt & 1;
            const bool work = jt >= 0 && jt < nmine;
            const uint16_t* ap = ldsw + b * bufA + aoff;
            const uint16_t* cp = ldsC + b * bufC + coff;
            f16x8v ch[NB], cl[NB];
            auto qblocks = [&](int k0, int k1) {
#pragma unroll
                for (int k = k0; k < k1; k++) {
                    if (cw + 4 * k < QB) {
                        const f16x8v ah = tr8(ap + 64 * k, RS), al = tr8(ap + 64 * k + planeA, RS);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[k][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ch[nb], acc[k][nb], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[k][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, cl[nb], acc[k][nb], 0, 0, 0);
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[k][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ch[nb], acc[k][nb], 0, 0, 0);
                    }
                }
            };
            if (work) {
                const int eA = sexp[b][0], eC = sexp[b][1];
                if (scaled && (eA != curA || eC != curC)) {        // a larger magnitude has turned up: the sums so far go to the new (smaller) scale
                    const int de = (eA - curA) + (eC - curC);
                    const float f = de < -126 ? 0.0f : __uint_as_float((uint32_t)(de + 127) << 23);
#pragma unroll
                    for (int k = 0; k < QW; k++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) acc[k][nb] *= f;
                }
                curA = eA, curC = eC, scaled = true;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) ch[nb] = tr8(cp + 16 * nb, CS), cl[nb] = tr8(cp + 16 * nb + planeC, CS);
                qblocks(0, QW);
            }
            __syncthreads();
        }
        // one partial bank per block: part[blockIdx.x][q][n]; register r of lane l is (q = 4 (l >> 4) + r, n = l & 15) of its tile
        const float iA = __uint_as_float((uint32_t)(127 - curA) << 23), iC = __uint_as_float((uint32_t)(127 - curC) << 23);
        float* out = part + (size_t)blockIdx.x * Q * N;
#pragma unroll
        for (int k = 0; k < QW; k++) {
            const int qb = cw + 4 * k;
            if (qb < QB) {
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int n = 16 * nb + (lane & 15);
                    if (n < N) {
#pragma unroll
                        for (int r = 0; r < 4; r++) out[(size_t)(16 * qb + 4 * (lane >> 4) + r) * N + n] = (acc[k][nb][r] * iA) * iC;
                    }
                }
            }
        }
    }
}
// blocks per sequence of k_rowwgrad_lds: 1 when the sequences alone fill the chip
static int rowwgrad_split(const ToepGeom& rg) {
    const int tiles = (rg.P + 31) / 32;
    if (rg.S >= 192) return 1;
    return std::max(1, std::min(std::min(8, tiles), (255 + rg.S) / rg.S));     // enough blocks for the chip, at most one per row tile
}
// per-sequence partial banks part[S * TS][Q][N] (TS = rowwgrad_split) of the row GEMM's filter gradient; false when the
// shape is not covered
static bool rowwgrad_lds_ok(const float* A, const ToepGeom& rg) {
    if (rg.a0 != 0 || rg.sa != rg.Q || rg.lda != (int64_t)rg.P * rg.Q || rg.ldc != (int64_t)rg.P * rg.N) return false;
    if ((rg.Q & 15) || rg.Q > 480 || rg.Q < 64 || rg.N < 33 || rg.N > 48 || (rg.N & 3)) return false;
    return (((uintptr_t)A) & 15) == 0;
}
// nparts: partial banks per group that `part` holds afterwards (rg.B * rowwgrad_split(rg) reserved by the caller)
static bool launch_rowwgrad_lds(Engine& e, const float* A, const float* C, float* part, const ToepGeom& rg, const RowSrc* src = nullptr, int* nparts = nullptr) {
    if (!rowwgrad_lds_ok(A, rg) || (((uintptr_t)C) & 15)) return false;
    const RowSrc cs = src ? *src : RowSrc{0, 0, 0, 0, 0, 0, 0, 0};
    if (nparts) *nparts = rg.B * rowwgrad_split(rg);
    {
        const bool f32_only = gemm_f32_only();   // A/B: the float32 matrix instruction
        const long min_tiles = gemm_f16_min(512);       // measured: level at 288 tiles, +1 % at 576, +2 % at 864
        const int tps = (rg.P + 31) / 32, G = rg.S / rg.B, T = rg.B * tps;
        const size_t lds16 = (size_t)2 * 2 * 32 * (rg.Q + 8) * 2 + (size_t)2 * 2 * 32 * 56 * 2;
        if (!f32_only && nparts && (long)G * T >= min_tiles && lds16 + 64 <= 160 * 1024) {
            // one block per CU where the groups allow it, each with a run of a group's tiles and one partial bank
            int nbg = std::max(1, std::min(T, (256 + G / 2) / G));
            nbg = std::min(nbg, *nparts);
            const int tpb = (T + nbg - 1) / nbg;
            nbg = (T + tpb - 1) / tpb;
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
                hipLaunchKernelGGL(kern, dim3((unsigned)(G * nbg)), dim3(512), lds16, e.st, A, C, part, rg.P, rg.Q, rg.N, tps, rg.B, nbg, tpb, cs);
            };
            const int QB = rg.Q / 16;
            if (QB <= 28 && rg.Q <= 416) go(k_rowwgrad16<7, 13>);
            else if (QB <= 28) go(k_rowwgrad16<7, 15>);
            else go(k_rowwgrad16<8, 15>);
            *nparts = nbg;
            return true;
        }
    }
    auto pad16 = [](int x) { return x + ((16 - x % 32) + 32) % 32; };     // smallest stride >= x that is 16 mod 32
    const int ST = pad16(rg.Q), SN = pad16(48);
    const size_t lds = (size_t)32 * (ST + SN) * 4;
    const int QB = rg.Q / 16;
    if (lds > 64 * 1024) return false;
    // 8 waves per block (each owns every 8th block of 16 rows of q): a read is walked twice as fast as by 4, and reads are
    // few (54 -> 48 us; 16 waves: 53 us)
    const int QW = (QB + 7) / 8;
    const int TS = rowwgrad_split(rg);
#define ROWWGRAD(QWV) hipLaunchKernelGGL((k_rowwgrad_lds<QWV, 8>), dim3(rg.S * TS), dim3(512), lds, e.st, A, C, part, rg.P, rg.Q, rg.N, ST, SN, TS, cs)
    if (QW <= 2) ROWWGRAD(2);
    else if (QW == 3) ROWWGRAD(3);
    else ROWWGRAD(4);
#undef ROWWGRAD
    return true;
}

__global__ void k_sum_segments(const float* x, size_t per, int B, size_t total, float* y, int acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, j = i - g * per;
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};                     // four chains of adds, twelve loads in flight (k_sum_segments_T)
        for (int b = 0; b < B; b += 12) {
            float v[12];
#pragma unroll
            for (int u = 0; u < 12; u++) v[u] = b + u < B ? x[(g * B + b + u) * per + j] : 0.0f;
#pragma unroll
            for (int u = 0; u < 12; u++)
                if (b + u < B) c[u & 3] += v[u];
        }
        const float a = (c[0] + c[1]) + (c[2] + c[3]);
        y[i] = acc ? y[i] + a : a;
    }
}

// k_sum_segments followed by k_tall_bt_T in one pass: dBm[g][ip][j][n] (+)= sum_b part[g B + b][j][ip][n].  Threads walk the
// partial banks in their own order (B coalesced streams); the sums leave as N-float pieces to the transposed place.
__global__ void k_sum_segments_btT(const float* __restrict__ x, int B, int G, int H, int W, int N, float* __restrict__ dBm, int acc) {
    const size_t per = (size_t)H * W * N, total = per * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = i / per, r = i - g * per;       // r indexes a partial bank [W][H][N]
        const int n = (int)(r % N), ip = (int)((r / N) % H), j = (int)(r / ((size_t)N * H));
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};                     // four chains of adds, twelve loads in flight (k_sum_segments_T)
        for (int b = 0; b < B; b += 12) {
            float v[12];
#pragma unroll
            for (int u = 0; u < 12; u++) v[u] = b + u < B ? x[(g * B + b + u) * per + r] : 0.0f;
#pragma unroll
            for (int u = 0; u < 12; u++)
                if (b + u < B) c[u & 3] += v[u];
        }
        const float a = (c[0] + c[1]) + (c[2] + c[3]);
        float* o = dBm + g * per + ((size_t)ip * W + j) * N + n;
        *o = acc ? *o + a : a;
    }
}

// The window matrix of a Toeplitz operand, Wn[s][p][q] = A[s][a0 + p sa + q] (zero outside the valid range): with it
// the filter gradient of a short-window, wide-output form is the row GEMM's, transposed.
__global__ void k_windows(const float* __restrict__ A, ToepGeom gm, float* __restrict__ Wn) {
    const size_t total = (size_t)gm.S * gm.P * gm.Q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % gm.Q), p = (int)((i / gm.Q) % gm.P);
        const size_t s = i / ((size_t)gm.Q * gm.P);
        const int e = gm.a0 + p * gm.sa + q;
        Wn[i] = (e >= 0 && e < gm.amax) ? A[s * gm.lda + e] : 0.0f;
    }
}
// dB[g][q][n] (+)= sum_{b < B} part[g B + b][n][q], Q <= 48.  One block per (NT columns n, group): the partial banks
// are read as contiguous [NT][Q] spans, the sums turn in LDS and leave as row pieces of NT floats.  NT = 32 when there are
// groups enough to fill the chip; 4 for the reference's one-mini-batch steps (13 blocks of 32 columns took 21 us there).
template <int NT>
__global__ __launch_bounds__(256) void k_sum_segments_T(const float* __restrict__ part, int Q, int N, int B, float* __restrict__ dB, int acc) {
    __shared__ float t[48][NT + 1];
    const int n0 = blockIdx.x * NT, g = blockIdx.y, tid = threadIdx.x;
    const int nn_max = min(NT, N - n0), cnt = nn_max * Q;
    const size_t per = (size_t)Q * N;
    for (int idx = tid; idx < cnt; idx += 256) {
        const float* src = part + (size_t)g * B * per + (size_t)n0 * Q + idx;
        // four chains of adds (partial b goes to chain b & 3, as ever); the loads come twelve at a time - with four at a time a
        // sum over the 36-48 partial banks of a one-mini-batch step was nine to twelve trips to memory in a row
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < B; b += 12) {
            float v[12];
#pragma unroll
            for (int u = 0; u < 12; u++) v[u] = b + u < B ? src[(size_t)(b + u) * per] : 0.0f;
#pragma unroll
            for (int u = 0; u < 12; u++)
                if (b + u < B) a[u & 3] += v[u];
        }
        // (a tail of 1-3 partials used to go to chain 0 and now goes to chains 0..2: the same terms)
        t[idx % Q][idx / Q] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
    for (int idx = tid; idx < Q * NT; idx += 256) {
        const int q = idx / NT, nn = idx % NT;
        if (nn < nn_max) {
            float* o = dB + (size_t)g * per + (size_t)q * N + n0 + nn;
            *o = acc ? *o + t[q][nn] : t[q][nn];
        }
    }
}

static void launch_wgrad(Engine& e, const float* A, const float* C, float* dB, const ToepGeom& gm, int acc) {
    hipStream_t st = e.st;
    const int G = gm.S / gm.B;
    if (is_tall(gm)) {   // dBt[j][(i',n)] = sum_{s,rho} A[s][rho][j] * dW[s][rho][(i',n)], dW = scatter of C
        const int H = gm.Q / gm.sa, R = gm.amax / gm.sa;
        const size_t per = (size_t)gm.Q * gm.N;
        float* dW = e.arena.alloc((size_t)gm.S * R * H * gm.N);
        float* dBt = e.arena.alloc(per * G);
        if (!dW || !dBt) {
            e.failed = true;
            return;
        }
        const ToepGeom rg = tall_row_geom(gm);
        const int rtiles = ((rg.Q + 127) / 128) * ((rg.N + 31) / 32);
        static const bool no_src = getenv("MOTIFS_NO_ROW_SRC") != nullptr;
        // the row kernel forms the scattered rows itself, from C (one launch and a round trip of dW through memory less:
        // 13.78 -> 13.53 ms per 64-mini-batch step, 2.27 -> 2.23 at one)
        const bool in_kernel = !no_src && rg.B > 1 && rtiles * G < 2048 && (gm.N & 3) == 0 && (gm.ldc & 3) == 0 &&
                               gm.a0 % gm.sa == 0 && rowwgrad_lds_ok(A, rg) && (((uintptr_t)C) & 15) == 0;
        if (!in_kernel)
            hipLaunchKernelGGL(k_tall_scatter, dim3(nblocks((size_t)gm.S * R * H * gm.N)), dim3(256), 0, st, C, dW, gm.S, gm.P, H, gm.N, R,
                               gm.a0 / gm.sa, gm.ldc);
        if (rg.B > 1 && rtiles * G < 2048) {      // per-sequence partial banks, then their sums (see below)
            const int TS = rowwgrad_split(rg);
            float* part = e.arena.alloc(per * gm.S * TS);
            if (!part) {
                e.failed = true;
                return;
            }
            ToepGeom r1 = rg;
            r1.B = 1;
            int nparts = rg.B * TS;                // partial banks per group
            const RowSrc scat{1, gm.N / 4, gm.a0 / gm.sa, gm.P, 0, 0, 0, gm.ldc};
            if (in_kernel) {
                (void)launch_rowwgrad_lds(e, A, C, part, rg, &scat, &nparts);
            } else if (!launch_rowwgrad_lds(e, A, dW, part, rg, nullptr, &nparts)) {
                hipLaunchKernelGGL(k_wgrad_mfma, dim3((rg.Q + 127) / 128, gm.S, (rg.N + 31) / 32), dim3(256), 0, st, A, dW, part, r1, 0);
                nparts = rg.B;
            }
            hipLaunchKernelGGL(k_sum_segments_btT, dim3(nblocks(per * G)), dim3(256), 0, st, part, nparts, G, H, gm.sa, gm.N, dB, acc);
            return;
        }
        hipLaunchKernelGGL(k_wgrad_mfma, dim3((rg.Q + 127) / 128, G, (rg.N + 31) / 32), dim3(256), 0, st, A, dW, dBt, rg, 0);
        hipLaunchKernelGGL(k_tall_bt_T, dim3(nblocks(per * G)), dim3(256), 0, st, dBt, G, H, gm.sa, gm.N, dB, acc);
        return;
    }
    const size_t per = (size_t)gm.Q * gm.N;
    // short windows, wide outputs (the D-layer analysis): the transposed row-GEMM gradient over the window matrix
    if (gm.Q <= 48 && gm.Q > 32 && gm.N >= 64 && gm.ldc == (int64_t)gm.P * gm.N) {
        ToepGeom rg;
        rg.S = gm.S, rg.P = gm.P, rg.Q = gm.N, rg.N = gm.Q, rg.sa = gm.N, rg.a0 = 0, rg.amax = gm.P * gm.N;
        rg.lda = (int64_t)gm.P * gm.N, rg.ldc = (int64_t)gm.P * gm.Q, rg.B = gm.B, rg.ldb = 0;
        float* Wn = e.arena.alloc((size_t)gm.S * gm.P * gm.Q);
        float* part = e.arena.alloc(per * gm.S * rowwgrad_split(rg));
        if (!Wn || !part) {
            e.failed = true;
            return;
        }
        static const bool no_src = getenv("MOTIFS_NO_ROW_SRC") != nullptr;
        // the row kernel reads the windows from the signal itself
        const bool in_kernel = !no_src && (gm.a0 & 3) == 0 && (gm.sa & 3) == 0 && (gm.amax & 3) == 0 && (gm.lda & 3) == 0 &&
                               rowwgrad_lds_ok(C, rg) && (((uintptr_t)A) & 15) == 0;
        const RowSrc win{2, 0, 0, 0, gm.a0, gm.sa, gm.amax, gm.lda};
        if (!in_kernel) hipLaunchKernelGGL(k_windows, dim3(nblocks((size_t)gm.S * gm.P * gm.Q)), dim3(256), 0, st, A, gm, Wn);
        int nparts = 0;
        if (in_kernel ? launch_rowwgrad_lds(e, C, A, part, rg, &win, &nparts) : launch_rowwgrad_lds(e, C, Wn, part, rg, nullptr, &nparts)) {
            if (G * ((gm.N + 31) / 32) >= 128)
                hipLaunchKernelGGL(k_sum_segments_T<32>, dim3((gm.N + 31) / 32, G), dim3(256), 0, st, part, gm.Q, gm.N, nparts, dB, acc);
            else
                hipLaunchKernelGGL(k_sum_segments_T<4>, dim3((gm.N + 3) / 4, G), dim3(256), 0, st, part, gm.Q, gm.N, nparts, dB, acc);
            return;
        }
    }
    // Few groups give few blocks (one per (tile, group)), each with a reduction over all B*P rows of its group:
    // reduce per sequence instead (B times the blocks) and add the B partial banks afterwards.
    const bool mfma = gm.N > 8 && gm.N <= 64 && gm.Q >= 256;
    const int tiles = mfma ? ((gm.Q + 127) / 128) * ((gm.N + 31) / 32) : ((gm.N + 63) / 64) * ((gm.Q + 63) / 64);
    auto kernels = [&](float* out, const ToepGeom& gg, int groups, int accf) {
        if (mfma) {
            hipLaunchKernelGGL(k_wgrad_mfma, dim3((gg.Q + 127) / 128, groups, (gg.N + 31) / 32), dim3(256), 0, st, A, C, out, gg, accf);
        } else if (gg.N <= 32) {
            dim3 grid((gg.N + 31) / 32, (gg.Q + 63) / 64, groups);
            hipLaunchKernelGGL(k_wgrad<32>, grid, dim3(256), 0, st, A, C, out, gg, accf);
        } else {
            dim3 grid((gg.N + 63) / 64, (gg.Q + 63) / 64, groups);
            hipLaunchKernelGGL(k_wgrad<64>, grid, dim3(256), 0, st, A, C, out, gg, accf);
        }
    };
    if (gm.B > 1 && tiles * G < 2048) {
        float* part = e.arena.alloc(per * gm.S);
        if (!part) {
            e.failed = true;
            return;
        }
        ToepGeom g1 = gm;
        g1.B = 1;
        kernels(part, g1, gm.S, 0);
        hipLaunchKernelGGL(k_sum_segments, dim3(nblocks(per * G)), dim3(256), 0, st, part, per, gm.B, per * G, dB, acc);
        return;
    }
    kernels(dB, gm, G, acc);
}

// dA[s][e] += sum_{p,q: a0 + p*sa + q = e} sum_n dC[s][p][n] * Bm[g][q][n]
// (the adjoint of the Toeplitz gather).  One thread per (s, e); the p range is at most ceil(Q/sa).
__global__ void k_toep_bwd_a(const float* __restrict__ dC, const float* __restrict__ Bm, float* __restrict__ dA,
                             ToepGeom gm) {
    const int64_t total = (int64_t)gm.S * gm.amax;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i / gm.amax), e = (int)(i - (int64_t)s * gm.amax);
        const float* Bg = Bm + (size_t)(s / gm.B) * gm.ldb;
        const float* dCs = dC + (size_t)s * gm.ldc;
        // q = e - a0 - p*sa in [0, Q)  ->  p in [ceil((e - a0 - Q + 1)/sa), floor((e - a0)/sa)]
        const int t = e - gm.a0;
        int p_hi = t >= 0 ? t / gm.sa : -1;
        int p_lo = t - gm.Q + 1 > 0 ? (t - gm.Q + 1 + gm.sa - 1) / gm.sa : 0;
        if (p_hi > gm.P - 1) p_hi = gm.P - 1;
        float acc = 0.0f;
        for (int p = p_lo; p <= p_hi; p++) {
            const int q = t - p * gm.sa;
            const float* b = Bg + (size_t)q * gm.N;
            const float* d = dCs + (size_t)p * gm.N;
            for (int n = 0; n < gm.N; n++) acc = fmaf(d[n], b[n], acc);
        }
        dA[(size_t)s * gm.lda + e] += acc;
    }
}

// sum the per-group slices of x into y (shared parameter gradient)
__global__ void k_sum_groups(const float* x, size_t per, int G, float* y) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    double acc = 0;
    int g = 0;
    for (; g + 16 <= G; g += 16) {                   // sixteen loads in flight, added in group order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = x[(size_t)(g + u) * per + j];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
    }
    for (; g < G; g++) acc += x[(size_t)g * per + j];
    y[j] += (float)acc;
}

// per group [H][W][N] -> out[i'][n][j] = in[H-1-i'][j][n]
// out[g][ip][n][j] (+)= x[g][H-1-ip][j][n]: the rows of a bank reversed and each [W][N] slice transposed, through a
// 64 x 64 LDS tile so that both sides move in contiguous runs.
__global__ __launch_bounds__(256) void k_flipT(const float* __restrict__ x, int H, int W, int N, float* __restrict__ out, int acc) {
    __shared__ float tile[64][65];
    const int tn = (N + 63) / 64;
    const int j0 = (blockIdx.x / tn) * 64, n0 = (blockIdx.x % tn) * 64, ip = blockIdx.y;
    const size_t per = (size_t)H * W * N;
    const float* xs = x + (size_t)blockIdx.z * per + (size_t)(H - 1 - ip) * W * N;
    float* os = out + (size_t)blockIdx.z * per + (size_t)ip * N * W;
    const int nj = min(64, W - j0), nn = min(64, N - n0);
    for (int i = threadIdx.x; i < nj * nn; i += 256) {
        const int jj = i / nn, n = i - jj * nn;
        tile[jj][n] = xs[(size_t)(j0 + jj) * N + n0 + n];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nj * nn; i += 256) {
        const int n = i / nj, jj = i - n * nj;
        float* o = &os[(size_t)(n0 + n) * W + j0 + jj];
        *o = acc ? *o + tile[jj][n] : tile[jj][n];
    }
}
static void launch_flipT(hipStream_t st, const float* x, int g, int H, int W, int N, float* out, int acc) {
    hipLaunchKernelGGL(k_flipT, dim3(((W + 63) / 64) * ((N + 63) / 64), H, g), dim3(256), 0, st, x, H, W, N, out, acc);
}

// dA[s][e] += sum_{p,q: a0 + p*sa + q = e} sum_n dC[s][p][n] * Bm[g][q][n]  (adjoint of the Toeplitz
// gather).  With A viewed as rows of W = sa columns and the window as H = Q/W rows, this is again a
// Toeplitz GEMM: over dC (rows of N columns) with the filter flipT(Bm) = [H][N][W], rows reversed.
static bool toep_adjoint_a(Engine& e, const float* dC, const float* Bm, float* dA, const ToepGeom& gm, int acc = 1) {
    const int W = gm.sa, H = gm.Q / gm.sa;
    const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
    const size_t per = (size_t)gm.Q * gm.N;
    bool fresh;
    float* tmp = e.relayout(Bm, 5, W, gm.N, H, per * gB, fresh);
    if (!tmp) return false;
    if (fresh) launch_flipT(e.st, Bm, gB, H, W, gm.N, tmp, 0);
    ToepGeom g2;
    g2.S = gm.S;
    g2.P = gm.amax / W;
    g2.Q = H * gm.N;
    g2.N = W;
    g2.sa = gm.N;
    g2.a0 = -(H - 1) * gm.N - (gm.a0 / W) * gm.N;
    g2.amax = gm.P * gm.N;
    g2.lda = gm.ldc;
    g2.ldc = gm.lda;
    g2.B = gm.B;
    g2.ldb = gm.ldb == 0 ? 0 : (int64_t)per;
    launch_toep(e, dC, tmp, dA, g2, acc);
    return true;
}

// a4 (warmup_ZY's conv(S, D, flipped=true) | conv(S, D), model.jl:171-173) on base codes.  S is one-hot, so the Toeplitz GEMM
// out[s][p][n] = sum_q S[s][4p + q] Bm[q][n] collapses to fl gathered rows of the bank per output row:
// out[s][p][:] = sum_{k < fl} Bm[4k + code[s][p + k]][:] ("fl gathered adds").  The pass is then bound by writing the
// codes image (2 c M floats per read: 302 KB at configs[1]), not by a contraction: the bank sits in LDS (one extra zero row for
// an all-zero column), a block streams one read's rows, every lane adds its four columns of fl rows and stores 16 bytes.
constexpr int OH_TPB = 1024;   // sixteen waves share one copy of the bank: LDS round trips of different rows overlap
// The launch is a fixed number of blocks (two per CU); block b takes the b-th equal share of ALL output rows of the launch, S * P of
// them in (read, position) order - a share may straddle reads - and stages the bank once.  (One block per read left a third of
// the chip idle in the second round at 384 reads: 38.7 us against 55.6 with 256-thread blocks, and against 46 for the GEMM form.)
template <int FL>       // filter length; 0 = run-time
__global__ __launch_bounds__(OH_TPB) void k_onehot_bank_scan(const uint8_t* __restrict__ codes, int pitch, const float* __restrict__ Bm,
                                                          float* __restrict__ out, ToepGeom gm, int fl_rt, int L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int fl = FL ? FL : fl_rt;
    const int N4 = gm.N >> 2, Q = gm.Q;
    float4* bank = (float4*)lds;                           // [Q + 1][N4]; the shared bank (gm.ldb == 0)
    const int tid = threadIdx.x;
    const float4* B4 = (const float4*)Bm;
    for (int i = tid; i < Q * N4; i += OH_TPB) bank[i] = B4[i];
    for (int i = tid; i < N4; i += OH_TPB) bank[(size_t)Q * N4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int rows_par = OH_TPB / N4;                     // rows in flight per block (10 at N = 400)
    const int rs = tid / N4, cg = tid - rs * N4;
    if (rs >= rows_par) return;
    const int64_t total = (int64_t)gm.S * gm.P;
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t r_lo = (int64_t)blockIdx.x * per, r_hi = r_lo + per < total ? r_lo + per : total;
    for (int64_t r = r_lo + rs; r < r_hi; r += rows_par) {
        const int sq = (int)(r / gm.P), p = (int)(r - (int64_t)sq * gm.P);
        const uint8_t* cd = codes + (size_t)sq * pitch + p;        // the row's fl codes: the same bytes for every lane of the row (L1 / scalar-like)
        float4* o4 = (float4*)(out + (size_t)sq * gm.ldc) + (size_t)p * N4 + cg;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FL) {
            // all FL code bytes, then all FL bank rows, in flight before the adds: two round trips per output row, not 2 FL
            int c[FL ? FL : 1];
            {   // the row's FL code bytes as aligned dwords + a funnel shift (the rows carry >= 4 bytes of padding and the matrix a
                // guard behind its last row, so the dword past the window's end is readable): FL/4 + 1 loads instead of FL
                const uintptr_t ad = (uintptr_t)cd;
                const uint32_t* wp = (const uint32_t*)(ad & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(ad & 3);
                constexpr int NWD = (FL - 1) / 4 + 2;                // dwords the funnel shift reads: those that hold bytes p .. p + FL - 1 + 3
                uint32_t wds[NWD];
#pragma unroll
                for (int j = 0; j < NWD; j++) wds[j] = wp[j];
#pragma unroll
                for (int k = 0; k < FL; k++) {
                    const uint32_t al = __builtin_amdgcn_alignbyte(wds[k / 4 + 1], wds[k / 4], sh);
                    const int cc = (int)((al >> (8 * (k % 4))) & 0xffu);
                    c[k] = p + k < L ? cc : 4;
                }
            }
            float4 w[FL ? FL : 1];
#pragma unroll
            for (int k = 0; k < FL; k++) w[k] = bank[(size_t)(c[k] < 4 ? 4 * k + c[k] : Q) * N4 + cg];
#pragma unroll
            for (int k = 0; k < FL; k++) acc.x += w[k].x, acc.y += w[k].y, acc.z += w[k].z, acc.w += w[k].w;
        } else {
            for (int k = 0; k < fl; k++) {
                const int c = p + k < L ? cd[k] : 4;
                const float4 w = bank[(size_t)(c < 4 ? 4 * k + c : Q) * N4 + cg];
                acc.x += w.x, acc.y += w.y, acc.z += w.z, acc.w += w.w;
            }
        }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4{acc.x, acc.y, acc.z, acc.w}, (f32x4*)o4);      // written once, read by the next kernel from HBM anyway
    }
}

Tensor Engine::toep_onehot(Tensor A, const uint8_t* codes, int pitch, Tensor Bm, const ToepGeom& gm) {
    const int fl = gm.Q / 4, L = (int)(gm.lda / 4);
    const size_t lds = ((size_t)(gm.Q + 1) * gm.N * 4 + 15) & ~(size_t)15;
    // few reads (the reference's 6-read step) leave most CUs without a block, and a bank past the LDS has no fast path here: the GEMM form
    static const bool off = getenv("MOTIFS_NO_ONEHOT_SCAN") != nullptr;      // A/B: the GEMM form for every launch
    if (off || !codes || gm.sa != 4 || gm.a0 != 0 || (gm.N & 3) || gm.N > 1024 || gm.Q != 4 * fl || gm.S < 96 || lds > 80 * 1024 || gm.ldb != 0 ||
        gm.P + fl - 1 > L)
        return toep(A, Bm, gm);
    Tensor out = make((size_t)gm.S * gm.ldc, A->needs_grad || Bm->needs_grad);
    if (failed) return out;
    auto go = [&](auto kern) {
        if (!onehot_attr_set) {      // once per engine (an engine lives on one device): the call costs more than the kernel's launch
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            onehot_attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(512), dim3(OH_TPB), lds, st, codes, pitch, Bm->v, out->v, gm, fl, L);    // two blocks per CU
    };
    if (fl == 8) go(k_onehot_bank_scan<8>);
    else if (fl == 12) go(k_onehot_bank_scan<12>);
    else if (fl == 16) go(k_onehot_bank_scan<16>);
    else if (fl == 20) go(k_onehot_bank_scan<20>);
    else go(k_onehot_bank_scan<0>);
    if (recording && out->needs_grad)
        tape.push_back([this, out, A, Bm, gm]() {          // the adjoints of toep(): the image A holds the same one-hot values
            if (!out->g) return;
            if (A->needs_grad) {
                int a = 1;
                const bool whole = gm.amax == gm.lda && gm.amax % gm.sa == 0;
                float* dA = whole ? grad_first(A, a) : grad(A);
                if (dA) toep_adjoint_a(*this, out->g, Bm->v, dA, gm, a);
            }
            if (Bm->needs_grad) {
                float* dB = grad(Bm);
                if (!dB) return;
                const int G = gm.S / gm.B;
                if (gm.ldb != 0 || G == 1) {
                    launch_wgrad(*this, A->v, out->g, dB, gm, 1);
                } else {
                    const size_t per = (size_t)gm.Q * gm.N;
                    float* tmp = arena.alloc(per * G);
                    if (!tmp) {
                        failed = true;
                        return;
                    }
                    launch_wgrad(*this, A->v, out->g, tmp, gm, 0);
                    hipLaunchKernelGGL(k_sum_groups, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, tmp, per, G, dB);
                }
            }
        });
    return out;
}

// toep(A, Bm) + b * y with a constant image y of the output's layout: folded into the tall form's gather (its VJP is toep's)
Tensor Engine::toep_plus(Tensor A, Tensor Bm, const ToepGeom& gm, Tensor y, float b) {
    static const bool off = getenv("MOTIFS_NO_TOEP_PLUS") != nullptr;
    if (off || !is_tall(gm) || y->needs_grad || y->n != (size_t)gm.S * gm.ldc || gm.ldc != (int64_t)gm.P * gm.N)
        return lin(toep(A, Bm, gm), 1.0f, y, b, 0.0f);
    return toep(A, Bm, gm, y->v, b);
}

Tensor Engine::toep(Tensor A, Tensor Bm, const ToepGeom& gm, const float* y, float yb) {
    Tensor out = make((size_t)gm.S * gm.ldc, A->needs_grad || Bm->needs_grad);
    if (failed) return out;
    launch_toep(*this, A->v, Bm->v, out->v, gm, 0, y, yb);
    if (recording && out->needs_grad)
        tape.push_back([this, out, A, Bm, gm]() {
            if (!out->g) return;
            if (A->needs_grad) {
                // the adjoint writes every element of dA when the windows tile whole rows: no zero fill on first use
                int a = 1;
                const bool whole = gm.amax == gm.lda && gm.amax % gm.sa == 0;
                float* dA = whole ? grad_first(A, a) : grad(A);
                if (dA) toep_adjoint_a(*this, out->g, Bm->v, dA, gm, a);
            }
            if (Bm->needs_grad) {
                float* dB = grad(Bm);
                if (!dB) return;
                const int G = gm.S / gm.B;
                if (gm.ldb != 0 || G == 1) {      // one mini-batch (the reference's schedule): its partial IS the sum over groups
                    launch_wgrad(*this, A->v, out->g, dB, gm, 1);
                } else {   // shared filter: per-group partials, then a sum over groups
                    const size_t per = (size_t)gm.Q * gm.N;
                    float* tmp = arena.alloc(per * G);
                    if (!tmp) {
                        failed = true;
                        return;
                    }
                    launch_wgrad(*this, A->v, out->g, tmp, gm, 0);
                    hipLaunchKernelGGL(k_sum_groups, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, tmp, per, G, dB);
                }
            }
        });
    return out;
}

Tensor Engine::wgrad(Tensor A, Tensor C, const ToepGeom& gm) {
    const int G = gm.S / gm.B;
    Tensor out = make((size_t)G * gm.Q * gm.N, A->needs_grad || C->needs_grad);
    if (failed) return out;
    launch_wgrad(*this, A->v, C->v, out->v, gm, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, A, C, gm]() {
            if (!out->g) return;
            ToepGeom g2 = gm;
            g2.ldb = (int64_t)gm.Q * gm.N;   // the "filter" of the adjoints is dOut, one slice per group
            // the D-layer's filter gradient (Q = 4 fl rows of N = 2M): both adjoints' re-layouts of dOut in one launch
            if (A->needs_grad && C->needs_grad && gm.sa == 4 && (gm.Q & 3) == 0 && (gm.N & 1) == 0) prelayout_an(out->g, gm.S / gm.B, gm.N / 2, gm.Q / 4);
            if (failed) return;
            if (A->needs_grad) {
                int a = 1;
                const bool whole = gm.amax == gm.lda && gm.amax % gm.sa == 0;
                float* dA = whole ? grad_first(A, a) : grad(A);
                if (dA) toep_adjoint_a(*this, C->v, out->g, dA, g2, a);
            }
            if (C->needs_grad) {
                int a = 1;
                float* dCc = gm.ldc == (int64_t)gm.P * gm.N ? grad_first(C, a) : grad(C);
                if (dCc) launch_toep(*this, A->v, out->g, dCc, g2, a);
            }
        });
    return out;
}

// ---------------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------------
// DA[g][(k,a)][j]: j < M -> D[g][j][4k+a];  j >= M -> D[g][j-M][4(fl-1-k) + 3-a]   (reverse strand, model.jl:173)
__global__ void k_expandD(const float* D, int g, int M, int fl, float* DA, int acc) {
    const size_t total = (size_t)g * fl * 4 * 2 * M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % (2 * M));
        size_t r = i / (2 * M);
        const int ka = (int)(r % (fl * 4));
        const int gg = (int)(r / (fl * 4));
        const int m = j < M ? j : j - M;
        const int src = j < M ? ka : (fl * 4 - 1 - ka);   // 4(fl-1-k) + (3-a) == 4fl - 1 - (4k+a)
        const float v = D[((size_t)gg * M + m) * (fl * 4) + src];
        DA[i] = acc ? DA[i] + v : v;
    }
}
__global__ void k_collapseD(const float* GA, int g, int M, int fl, float* Dg, int acc) {
    const size_t total = (size_t)g * M * fl * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ka = (int)(i % (fl * 4));
        size_t r = i / (fl * 4);
        const int m = (int)(r % M);
        const int gg = (int)(r / M);
        const float* base = GA + (size_t)gg * fl * 4 * 2 * M;
        const float v = base[(size_t)ka * 2 * M + m] + base[(size_t)(fl * 4 - 1 - ka) * 2 * M + M + m];
        Dg[i] = acc ? Dg[i] + v : v;
    }
}
Tensor Engine::expandD(Tensor D, int g, int M, int fl) {
    Tensor out = make((size_t)g * fl * 4 * 2 * M, D->needs_grad);
    if (failed) return out;
    EW(k_expandD, out->n, D->v, g, M, fl, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, D, g, M, fl]() {
            int acc;
            float* dx = out->g ? grad_first(D, acc) : nullptr;
            if (dx) EW(k_collapseD, D->n, out->g, g, M, fl, dx, acc);
        });
    return out;
}
Tensor Engine::collapseD(Tensor GA, int g, int M, int fl) {
    Tensor out = make((size_t)g * M * fl * 4, GA->needs_grad);
    if (failed) return out;
    EW(k_collapseD, out->n, GA->v, g, M, fl, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, GA, g, M, fl]() {
            int acc;
            float* dx = out->g ? grad_first(GA, acc) : nullptr;
            if (dx) EW(k_expandD, GA->n, out->g, g, M, fl, dx, acc);
        });
    return out;
}

// The D bank in every layout its consumers ask for, in one launch (expandD, flipT, k_tall_bt, k_frag_b16 and k_frag_bw were five
// launches per bank and four banks per step): every output element is one element of D,
//   an[(k,a)][j]   = da(4k + a, j)                      da(q, j) = j < M ? D[j][q] : D[j - M][4 fl - 1 - q]     (expandD)
//   syn[ip][x][y]  = da(4 (fl-1-ip) + y, x)             (flipT of an viewed [fl][4][2M])
//   Bt[j][ip][n]   = da(4 (fl-1-ip) + n, j)             (k_tall_bt of syn viewed [fl][2M][4])
//   Bf16[ks][l][cb] = Bt'[4 ks + (l >> 4)][16 cb + (l & 15)]    (k_frag_b16 of Bt viewed [2M][4 fl], columns clamped / zero)
//   Bfw[ct][kg][l][u] = an[8 kg + 2 u + (l >> 5)][32 ct + (l & 31)]   (k_frag_bw of an, columns clamped)
// src_an != null: the bank is given in its analysis form already (the gradient a wgrad VJP uses as its filter): da(q, j) = src_an[q][j],
// `an` is not written
__global__ void k_bankD(const float* __restrict__ D, int g, int M, int fl, float* __restrict__ an, float* __restrict__ syn, float* __restrict__ Bt,
                        float* __restrict__ Bf16, float* __restrict__ Bfw, const float* __restrict__ src_an) {
    const int Q = 4 * fl, N2 = 2 * M, KG = Q / 8, NCT = (N2 + 31) / 32;
    const size_t per = (size_t)Q * N2, per16 = (size_t)(N2 / 4) * 256, perw = (size_t)NCT * KG * 256;
    const size_t tot_bank = 3 * per + per16 + perw, total = tot_bank * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / tot_bank;
        size_t r = i - gg * tot_bank;
        const float* Dg = src_an ? src_an + gg * per : D + gg * (size_t)M * Q;
        auto da = [&](int q, int j) {
            if (src_an) return Dg[(size_t)q * N2 + j];
            return j < M ? Dg[(size_t)j * Q + q] : Dg[(size_t)(j - M) * Q + (Q - 1 - q)];
        };
        if (r < per) {                                              // an [Q][2M]
            if (!src_an) an[gg * per + r] = da((int)(r / N2), (int)(r % N2));
        } else if ((r -= per) < per) {                              // syn [fl][2M][4]
            const int y = (int)(r % 4), x = (int)((r / 4) % N2), ip = (int)(r / ((size_t)4 * N2));
            syn[gg * per + r] = da(4 * (fl - 1 - ip) + y, x);
        } else if ((r -= per) < per) {                              // Bt [2M][fl][4]
            const int n = (int)(r % 4), ip = (int)((r / 4) % fl), j = (int)(r / ((size_t)4 * fl));
            Bt[gg * per + r] = da(4 * (fl - 1 - ip) + n, j);
        } else if ((r -= per) < per16) {                            // Bf16 [2M/4][64][4]
            const int cb = (int)(r & 3), lane = (int)((r >> 2) & 63);
            const int qq = 4 * (int)(r >> 8) + (lane >> 4);
            float v = 0.0f;
            if (16 * cb < Q) {
                const int c = min(16 * cb + (lane & 15), Q - 1);
                v = da(4 * (fl - 1 - c / 4) + (c & 3), qq);
            }
            Bf16[gg * per16 + r] = v;
        } else {                                                    // Bfw [NCT][KG][64][4]
            r -= per16;
            const int u = (int)(r & 3), lane = (int)((r >> 2) & 63);
            const int kg = (int)((r >> 8) % KG), ct = (int)((r >> 8) / KG);
            Bfw[gg * perw + r] = da(8 * kg + 2 * u + (lane >> 5), min(32 * ct + (lane & 31), N2 - 1));
        }
    }
}
// its VJP: d D[j][q] (+)= g(q, j) + g(4 fl - 1 - q, M + j),  g(q, x) = d an[q][x] + d syn[fl - 1 - q/4][x][q%4]  (either may be absent)
__global__ void k_bankD_bwd(const float* __restrict__ dan, const float* __restrict__ dsyn, int g, int M, int fl, float* __restrict__ dD, int acc) {
    const int Q = 4 * fl, N2 = 2 * M;
    const size_t per = (size_t)Q * N2, total = (size_t)g * M * Q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        const size_t t = i / Q;
        const int j = (int)(t % M);
        const size_t gg = t / M;
        auto gq = [&](int qq, int x) {
            float v = 0.0f;
            if (dan) v += dan[gg * per + (size_t)qq * N2 + x];
            if (dsyn) v += dsyn[gg * per + ((size_t)(fl - 1 - qq / 4) * N2 + x) * 4 + (qq & 3)];
            return v;
        };
        const float v = gq(q, j) + gq(Q - 1 - q, M + j);
        dD[i] = acc ? dD[i] + v : v;
    }
}
std::pair<Tensor, Tensor> Engine::bankD(Tensor D, int g, int M, int fl) {
    static const bool off = getenv("MOTIFS_NO_BANK_FUSION") != nullptr;
    const int Q = 4 * fl, N2 = 2 * M;
    if (off || (fl & 1) || (N2 & 3)) {
        Tensor DA = expandD(D, g, M, fl);
        return {DA, flipT(DA, g, fl, 4, N2)};
    }
    const size_t per = (size_t)Q * N2, per16 = (size_t)(N2 / 4) * 256, perw = (size_t)((N2 + 31) / 32) * (Q / 8) * 256;
    Tensor an = make(per * g, D->needs_grad), syn = make(per * g, D->needs_grad);
    float* Bt = arena.alloc(per * g);
    float* Bf16 = arena.alloc(per16 * g);
    float* Bfw = arena.alloc(perw * g);
    if (failed || !Bt || !Bf16 || !Bfw) {
        failed = true;
        return {an, syn};
    }
    hipLaunchKernelGGL(k_bankD, dim3(nblocks((3 * per + per16 + perw) * g)), dim3(256), 0, st, D->v, g, M, fl, an->v, syn->v, Bt, Bf16, Bfw, (const float*)nullptr);
    // the re-layouts the consumers will ask for (launch_toep's tall form on syn, launch_rowgemm_lds / k_tall_fused on its Bt,
    // launch_toep_wide on an), and the two forms as each other's flip (toep_adjoint_a)
    derived[RelayoutKey{(const void*)syn->v, 4, N2, 4, fl, per * g}] = Bt;
    derived[RelayoutKey{(const void*)Bt, 2, N2, Q, 0, per16 * g}] = Bf16;
    derived[RelayoutKey{(const void*)an->v, 3, Q, N2, 0, perw * g}] = Bfw;
    derived[RelayoutKey{(const void*)syn->v, 5, N2, 4, fl, per * g}] = an->v;
    derived[RelayoutKey{(const void*)an->v, 5, 4, N2, fl, per * g}] = syn->v;
    if (recording && D->needs_grad)
        tape.push_back([this, an, syn, D, g, M, fl]() {
            if (!an->g && !syn->g) return;
            int acc;
            float* dx = grad_first(D, acc);
            if (dx) EW(k_bankD_bwd, D->n, an->g, syn->g, g, M, fl, dx, acc);
        });
    return {an, syn};
}

// The re-layouts a convolution takes of a bank that is given in its analysis form [g][4 fl][2M] and is not a tensor of the graph
// (the gradient a wgrad VJP uses as the filter of its two adjoints): flipped form, tall form, fragment orders - one launch, the
// same keys the consumers look up (was flipT, k_tall_bt, k_frag_b16 and k_frag_bw, per DF pass)
void Engine::prelayout_an(const float* an, int g, int M, int fl) {
    static const bool off = getenv("MOTIFS_NO_BANK_FUSION") != nullptr;
    const int Q = 4 * fl, N2 = 2 * M;
    if (off || (fl & 1) || (N2 & 3)) return;
    const size_t per = (size_t)Q * N2, per16 = (size_t)(N2 / 4) * 256, perw = (size_t)((N2 + 31) / 32) * (Q / 8) * 256;
    if (derived.count(RelayoutKey{(const void*)an, 5, 4, N2, fl, per * g})) return;
    float* syn = arena.alloc(per * g);
    float* Bt = arena.alloc(per * g);
    float* Bf16 = arena.alloc(per16 * g);
    float* Bfw = arena.alloc(perw * g);
    if (!syn || !Bt || !Bf16 || !Bfw) {
        failed = true;
        return;
    }
    hipLaunchKernelGGL(k_bankD, dim3(nblocks((3 * per + per16 + perw) * g)), dim3(256), 0, st, (const float*)nullptr, g, M, fl, (float*)nullptr, syn, Bt, Bf16,
                       Bfw, an);
    derived[RelayoutKey{(const void*)an, 5, 4, N2, fl, per * g}] = syn;
    derived[RelayoutKey{(const void*)syn, 4, N2, 4, fl, per * g}] = Bt;
    derived[RelayoutKey{(const void*)Bt, 2, N2, Q, 0, per16 * g}] = Bf16;
    derived[RelayoutKey{(const void*)an, 3, Q, N2, 0, perw * g}] = Bfw;
}

// The F bank (reference layout F[g][K][2M][h]) in its two GEMM forms in one launch (swap02 then flipT were two):
//   FA[i][j][k] = F[k][j][i]        syn[ip][k][j] = FA[h-1-ip][j][k] = F[k][j][h-1-ip]
// Threads walk F in memory order; both outputs leave as K-strided pieces (the bank is 115 200 floats at configs[1]).
__global__ void k_bankF(const float* __restrict__ F, int g, int K, int N2, int h, float* __restrict__ FA, float* __restrict__ syn) {
    const size_t per = (size_t)K * N2 * h, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i - gg * per;
        const int ii = (int)(r % h), j = (int)((r / h) % N2), k = (int)(r / ((size_t)h * N2));
        const float v = F[i];
        FA[gg * per + ((size_t)ii * N2 + j) * K + k] = v;
        syn[gg * per + ((size_t)(h - 1 - ii) * K + k) * N2 + j] = v;
    }
}
// d F[k][j][i] (+)= d FA[i][j][k] + d syn[h-1-i][k][j]   (either may be absent)
__global__ void k_bankF_bwd(const float* __restrict__ dFA, const float* __restrict__ dsyn, int g, int K, int N2, int h, float* __restrict__ dF, int acc) {
    const size_t per = (size_t)K * N2 * h, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i - gg * per;
        const int ii = (int)(r % h), j = (int)((r / h) % N2), k = (int)(r / ((size_t)h * N2));
        float v = 0.0f;
        if (dFA) v += dFA[gg * per + ((size_t)ii * N2 + j) * K + k];
        if (dsyn) v += dsyn[gg * per + ((size_t)(h - 1 - ii) * K + k) * N2 + j];
        dF[i] = acc ? dF[i] + v : v;
    }
}
std::pair<Tensor, Tensor> Engine::bankF(Tensor F, int g, int K, int N2, int h) {
    static const bool off = getenv("MOTIFS_NO_BANK_FUSION") != nullptr;
    if (off || F->n > ((size_t)1 << 20)) {              // a bank per mini-batch of a large step: the tiled transposes (13.07 against 13.18 ms at 64)
        Tensor FA = swap02(F, g, K, N2, h);
        return {FA, flipT(FA, g, h, N2, K)};
    }
    Tensor FA = make(F->n, F->needs_grad), syn = make(F->n, F->needs_grad);
    if (failed) return {FA, syn};
    EW(k_bankF, F->n, F->v, g, K, N2, h, FA->v, syn->v);
    if (recording && F->needs_grad)
        tape.push_back([this, FA, syn, F, g, K, N2, h]() {
            if (!FA->g && !syn->g) return;
            int acc;
            float* dx = grad_first(F, acc);
            if (dx) EW(k_bankF_bwd, F->n, FA->g, syn->g, g, K, N2, h, dx, acc);
        });
    return {FA, syn};
}

// per group [d0][d1][d2] -> [d2][d1][d0]
__global__ void k_swap02(const float* x, int g, int d0, int d1, int d2, float* out, int acc) {
    const size_t per = (size_t)d0 * d1 * d2, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i % per;      // r indexes the OUTPUT [d2][d1][d0]
        const int i0 = (int)(r % d0), i1 = (int)((r / d0) % d1), i2 = (int)(r / ((size_t)d0 * d1));
        const float v = x[gg * per + ((size_t)i0 * d1 + i1) * d2 + i2];
        out[i] = acc ? out[i] + v : v;
    }
}
// the same through an LDS tile of 16 middle indices: both the reads ([d0][16][d2]: runs of 16*d2 floats) and the writes
// ([d2][16][d0]: runs of 16*d0 floats) are then contiguous instead of 4-byte accesses d1*d2 floats apart
constexpr int SWAP_T1 = 16;
__global__ __launch_bounds__(256) void k_swap02_tiled(const float* __restrict__ x, int d0, int d1, int d2, float* __restrict__ out, int acc) {
    extern __shared__ float tile[];                      // [d0][T1][d2 + 1]
    const int gg = blockIdx.y, j0 = blockIdx.x * SWAP_T1;
    const int nj = d1 - j0 < SWAP_T1 ? d1 - j0 : SWAP_T1;
    const size_t per = (size_t)d0 * d1 * d2;
    const float* xs = x + (size_t)gg * per;
    float* os = out + (size_t)gg * per;
    const int run_in = nj * d2, p2 = d2 + 1;
    for (int r = threadIdx.x; r < run_in; r += 256) {            // x[i0][j0 + jj][i2], contiguous in (jj, i2): one division per thread
        const int jj = r / d2, i2 = r - jj * d2;
        for (int i0 = 0; i0 < d0; i0++) tile[(i0 * SWAP_T1 + jj) * p2 + i2] = xs[((size_t)i0 * d1 + j0) * d2 + r];
    }
    __syncthreads();
    const int run_out = nj * d0;
    for (int r = threadIdx.x; r < run_out; r += 256) {           // out[i2][j0 + jj][i0], contiguous in (jj, i0)
        const int jj = r / d0, i0 = r - jj * d0;
        for (int i2 = 0; i2 < d2; i2++) {
            float* o = &os[((size_t)i2 * d1 + j0) * d0 + r];
            const float v = tile[(i0 * SWAP_T1 + jj) * p2 + i2];
            *o = acc ? *o + v : v;
        }
    }
}
static void launch_swap02(hipStream_t st, const float* x, int g, int d0, int d1, int d2, float* out, int acc) {
    const size_t lds = (size_t)d0 * SWAP_T1 * (d2 + 1) * 4;
    if (lds <= 48 * 1024)
        hipLaunchKernelGGL(k_swap02_tiled, dim3((d1 + SWAP_T1 - 1) / SWAP_T1, g), dim3(256), lds, st, x, d0, d1, d2, out, acc);
    else
        hipLaunchKernelGGL(k_swap02, dim3(nblocks((size_t)g * d0 * d1 * d2)), dim3(256), 0, st, x, g, d0, d1, d2, out, acc);
}

Tensor Engine::swap02(Tensor x, int g, int d0, int d1, int d2) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    launch_swap02(st, x->v, g, d0, d1, d2, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, g, d0, d1, d2]() {
            int acc;
            float* dx = out->g ? grad_first(x, acc) : nullptr;
            if (dx) launch_swap02(st, out->g, g, d2, d1, d0, dx, acc);
        });
    return out;
}

Tensor Engine::flipT(Tensor Bm, int g, int H, int W, int N) {
    Tensor out = make(Bm->n, Bm->needs_grad);
    if (failed) return out;
    launch_flipT(st, Bm->v, g, H, W, N, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, Bm, g, H, W, N]() {
            // adjoint: dIn[i][j][n] += dOut[H-1-i][n][j]  == flipT with the roles of W and N exchanged
            int acc;
            float* dx = out->g ? grad_first(Bm, acc) : nullptr;
            if (dx) launch_flipT(st, out->g, g, H, N, W, dx, acc);
        });
    return out;
}

// ---------------------------------------------------------------------------------------------
// Syntax layer with sparse codes.  T [S][l][K] has ~q non-zeros per read (top-q projection,
// model.jl:181-192), so everything that multiplies by T — or is only consumed where T's mask is set —
// is done per non-zero on a 12 x 2M filter slab instead of as a dense GEMM over h*2M*K.  The kernels
// scan T for non-zeros themselves, so they stay correct (just slower) if T happens to be dense.
// Non-zeros are taken in memory order (deterministic sums).
// ---------------------------------------------------------------------------------------------
// Non-zero list of a [S][n] tensor: one block per read, entries in memory order (deterministic sums).
__global__ __launch_bounds__(1024) void k_build_nz(const float* __restrict__ x, int n, int* __restrict__ cnt,
                                                   uint2* __restrict__ ent) {
    // every wave owns a contiguous slice of the read: count, meet once, then write in ascending order
    // (4 waves per read when the reads fill the chip, 16 for the few reads of a small step)
    __shared__ int wcnt[16];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const float* xs = x + (size_t)s * n;
    uint2* es = ent + (size_t)s * n;
    const int q = (((n + nw - 1) / nw) + 63) & ~63, lo = min(n, wv * q), hi = min(n, lo + q);
    int c = 0;
    for (int e0 = lo; e0 < hi; e0 += 64) {
        const int e = e0 + lane;
        const float v = e < hi ? xs[e] : 0.0f;
        c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v != 0.0f));
    }
    if (lane == 0) wcnt[wv] = c;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wcnt[w];
    for (int e0 = lo; e0 < hi; e0 += 64) {
        const int e = e0 + lane;
        const float v = e < hi ? xs[e] : 0.0f;
        const bool hit = v != 0.0f;
        const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
        if (hit) es[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_uint2((unsigned)e, __float_as_uint(v));
        base += __builtin_popcountll(m);
    }
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < nw; w++) t += wcnt[w];
        cnt[s] = t;
    }
}

// floor(e / K) for the flat code indices e = p*K + k (e < 2^32 / K): one multiply-high instead of a division.  The entry
// lists are wave-uniform, so the division ran on the scalar unit - one per CU - and the sparse kernels were bound by
// it (PMC: 5.4 scalar instructions per vector one in k_sp_wgrad_ana).
static __device__ __forceinline__ uint32_t kmagic(int K) { return 0xFFFFFFFFu / (uint32_t)K + 1u; }
// S1: out[s][r][j] (+)= sum_nz v * FAf[g][p - r + h - 1][k][j]      (block = (read, 128 columns), thread = column j)
// any filter height: read-modify-write of the output rows in memory
__global__ __launch_bounds__(128) void k_sp_syn_any(NzView nz, const float* __restrict__ FAf, float* __restrict__ out,
                                                    SpDims d, int acc) {
    const int s = blockIdx.y, j = blockIdx.x * 128 + threadIdx.x;
    if (j >= d.W) return;
    const float* Fg = FAf + (size_t)(s / d.B) * d.ldf;
    float* os = out + (size_t)s * d.c * d.W;
    if (!acc)
        for (int r = 0; r < d.c; r++) os[(size_t)r * d.W + j] = 0.0f;
    const int cnt = nz.cnt[s];
    const uint2* es = nz.ent + (size_t)s * nz.cap;
    for (int z = 0; z < cnt; z++) {
        const uint2 en = es[z];
        const int p = (int)__umulhi(en.x, kmagic(d.K)), k = (int)(en.x - (unsigned)p * d.K);
        const float v = __uint_as_float(en.y);
        for (int ip = 0; ip < d.h; ip++) {
            const int r = p + d.h - 1 - ip;
            os[(size_t)r * d.W + j] += v * Fg[((size_t)ip * d.K + k) * d.W + j];
        }
    }
}

// The non-zeros come in ascending position p and each touches rows p .. p + h - 1, so the open rows live in a ring
// of 16 LDS slots per column (slot = row & 15, h <= 16); a row is written to HBM once, when the sweep has passed it.
// HT = the filter height when it is the usual 12 (loops unroll without scalar bookkeeping), 0 = read it from d.
template <int HT>
__global__ __launch_bounds__(128) void k_sp_syn(NzView nz, const float* __restrict__ FAf, float* __restrict__ out,
                                                SpDims d, int acc) {
    __shared__ float ring[16][128];
    const int s = blockIdx.y, tx = threadIdx.x, j = blockIdx.x * 128 + tx;
    const int h = HT ? HT : d.h;
    const bool live = j < d.W;
    const int jc = live ? j : d.W - 1;
    const float* Fg = FAf + (size_t)(s / d.B) * d.ldf + jc;
    float* os = out + (size_t)s * d.c * d.W + jc;
#pragma unroll
    for (int i = 0; i < 16; i++) ring[i][tx] = 0.0f;
    int base = 0;                                        // lowest open row
    const int cnt = nz.cnt[s];
    const uint2* es = nz.ent + (size_t)s * nz.cap;
    const size_t kw = (size_t)d.K * d.W;
    for (int z = 0; z <= cnt; z++) {                     // block-uniform control flow; the last trip closes the tail
        int p = d.c, k = 0;
        float v = 0.0f;
        if (z < cnt) {
            const uint2 en = es[z];
            p = (int)__umulhi(en.x, kmagic(d.K));
            k = (int)(en.x - (unsigned)p * d.K);
            v = __uint_as_float(en.y);
        }
        float f[16];                                     // the entry's filter column, in flight while rows close
        if (z < cnt) {
#pragma unroll
            for (int ip = 0; ip < 16; ip++)
                if (ip < h) f[ip] = Fg[(size_t)ip * kw + (size_t)k * d.W];
        }
        for (; base < p; base++) {                       // rows below p are final
            const float r = ring[base & 15][tx];
            ring[base & 15][tx] = 0.0f;
            if (live) {
                float* o = os + (size_t)base * d.W;
                *o = acc ? *o + r : r;
            }
        }
        if (z < cnt) {
#pragma unroll
            for (int ip = 0; ip < 16; ip++)
                if (ip < h) {                            // row p + h - 1 - ip
                    float* slot = &ring[(p + h - 1 - ip) & 15][tx];
                    *slot = fmaf(v, f[ip], *slot);
                }
        }
    }
}

// S2: dFAf[g][ip][k][j] += sum_{s in g} sum_nz v * dOut[s][p + h - 1 - ip][j]     (block = (128 columns, ip, group))
// Each thread owns the K outputs of its (ip, j): they are accumulated in a private LDS column (k is data-dependent)
// and added to dF once, instead of a read-modify-write of global memory per non-zero.
__global__ __launch_bounds__(128) void k_sp_wgrad_syn(NzView nz, const float* __restrict__ dOut, float* __restrict__ dF,
                                                      SpDims d) {
    extern __shared__ float accs[];                      // [K][128]
    const int g = blockIdx.z, ip = blockIdx.y, tx = threadIdx.x, j = blockIdx.x * 128 + tx;
    const int jc = j < d.W ? j : d.W - 1;                // clamp: out-of-range threads compute, do not store
    for (int k = 0; k < d.K; k++) accs[k * 128 + tx] = 0.0f;
    for (int b = 0; b < d.B; b++) {
        const int s = g * d.B + b;
        const float* ds = dOut + (size_t)s * d.c * d.W + (size_t)(d.h - 1 - ip) * d.W + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        for (int z = 0; z < cnt; z++) {                  // block-uniform
            const uint2 en = es[z];
            const int p = (int)__umulhi(en.x, kmagic(d.K)), k = (int)(en.x - (unsigned)p * d.K);
            accs[k * 128 + tx] = fmaf(__uint_as_float(en.y), ds[(size_t)p * d.W], accs[k * 128 + tx]);
        }
    }
    if (j < d.W) {
        float* dFg = dF + (size_t)g * d.h * d.K * d.W + (size_t)ip * d.K * d.W + j;
        for (int k = 0; k < d.K; k++) dFg[(size_t)k * d.W] += accs[k * 128 + tx];
    }
}

// S3: dB[g][i][j][k] (+)= sum_{s in g} sum_nz v * img[s][p + i][j]                 (block = (128 columns, i, group))
// Each thread owns the K outputs of its (i, j); they are accumulated in a private LDS row (the k index is
// data-dependent) and written once.
__global__ __launch_bounds__(128) void k_sp_wgrad_ana(const float* __restrict__ img, NzView nz, float* __restrict__ dB,
                                                      SpDims d, int acc) {
    extern __shared__ float accs[];                      // [128][K + 1]
    const int g = blockIdx.z, i = blockIdx.y, j = blockIdx.x * 128 + threadIdx.x;
    float* my = accs + (size_t)threadIdx.x * (d.K + 1);
    for (int k = 0; k < d.K; k++) my[k] = 0.0f;
    const int jc = j < d.W ? j : d.W - 1;                // clamp: out-of-range threads compute, do not store
    for (int b = 0; b < d.B; b++) {
        const int s = g * d.B + b;
        const float* is = img + (size_t)s * d.c * d.W + (size_t)i * d.W + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        for (int z = 0; z < cnt; z++) {
            const uint2 en = es[z];
            const int p = (int)__umulhi(en.x, kmagic(d.K)), k = (int)(en.x - (unsigned)p * d.K);
            my[k] = fmaf(__uint_as_float(en.y), is[(size_t)p * d.W], my[k]);
        }
    }
    // the block's outputs [128 columns][K] are one contiguous span: written by all threads in address order
    __syncthreads();
    const int j0 = blockIdx.x * 128, nj = min(128, d.W - j0);
    float* span = dB + (size_t)g * d.h * d.W * d.K + ((size_t)i * d.W + j0) * d.K;
    for (int idx = threadIdx.x; idx < nj * d.K; idx += 128) {
        const int jj = idx / d.K, k = idx - jj * d.K;
        const float v = accs[jj * (d.K + 1) + k];
        span[idx] = acc ? span[idx] + v : v;
    }
}

// S2 / S3 for steps of few mini-batches (the reference's schedule: G = 1, 48 blocks on 256 CUs, each walking the entries
// of all B reads one after the other: 26 us).  Here a block has B wave pairs, pair b walks read b's entries into its own LDS
// accumulators, and the B partial banks are added in read order on the way out.
__global__ __launch_bounds__(1024) void k_sp_wgrad_syn_reads(NzView nz, const float* __restrict__ dOut, float* __restrict__ dF, SpDims d) {
    extern __shared__ float accs[];                      // [B][K][128]
    __shared__ uint2 stg[16][64];
    const int g = blockIdx.z, ip = blockIdx.y, tx = threadIdx.x & 127, b = threadIdx.x >> 7, j0 = blockIdx.x * 128;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int jc = min(j0 + tx, d.W - 1);
    float* mine = accs + (size_t)b * d.K * 128;
    for (int k = 0; k < d.K; k++) mine[k * 128 + tx] = 0.0f;
    {
        const int s = g * d.B + b;
        const float* ds = dOut + (size_t)s * d.c * d.W + (size_t)(d.h - 1 - ip) * d.W + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        // 64 entries at a time come to the wave's LDS slot in one coalesced load; the image values they point at are then
        // fetched eight at a time (entry -> value was a chain of two trips to memory per entry)
        for (int z0 = 0; z0 < cnt; z0 += 64) {
            const int n = min(64, cnt - z0);
            if (lane < n) stg[wv][lane] = es[z0 + lane];
            for (int u0 = 0; u0 < n; u0 += 8) {
                uint2 en[8];
                float x[8];
                int kk[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    en[u] = stg[wv][min(u0 + u, n - 1)];
                    const int p = (int)__umulhi(en[u].x, kmagic(d.K));
                    kk[u] = (int)(en[u].x - (unsigned)p * d.K);
                    x[u] = ds[(size_t)p * d.W];
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (u0 + u < n) mine[kk[u] * 128 + tx] = fmaf(__uint_as_float(en[u].y), x[u], mine[kk[u] * 128 + tx]);
            }
        }
    }
    __syncthreads();
    float* dFg = dF + (size_t)g * d.h * d.K * d.W + (size_t)ip * d.K * d.W + j0;
    const int nj = min(128, d.W - j0);
    for (int idx = threadIdx.x; idx < d.K * 128; idx += blockDim.x) {
        const int k = idx >> 7, jj = idx & 127;
        float v = 0.0f;
        for (int bb = 0; bb < d.B; bb++) v += accs[(size_t)bb * d.K * 128 + idx];
        if (jj < nj) dFg[(size_t)k * d.W + jj] += v;
    }
}

__global__ __launch_bounds__(1024) void k_sp_wgrad_ana_reads(const float* __restrict__ img, NzView nz, float* __restrict__ dB, SpDims d, int acc) {
    extern __shared__ float accs[];                      // [B][128][K + 1]
    __shared__ uint2 stg[16][64];
    const int g = blockIdx.z, i = blockIdx.y, tx = threadIdx.x & 127, b = threadIdx.x >> 7, j0 = blockIdx.x * 128;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int jc = min(j0 + tx, d.W - 1);
    const int per = 128 * (d.K + 1);
    float* my = accs + (size_t)b * per + (size_t)tx * (d.K + 1);
    for (int k = 0; k < d.K; k++) my[k] = 0.0f;
    {
        const int s = g * d.B + b;
        const float* is = img + (size_t)s * d.c * d.W + (size_t)i * d.W + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        for (int z0 = 0; z0 < cnt; z0 += 64) {           // as in k_sp_wgrad_syn_reads
            const int n = min(64, cnt - z0);
            if (lane < n) stg[wv][lane] = es[z0 + lane];
            for (int u0 = 0; u0 < n; u0 += 8) {
                uint2 en[8];
                float x[8];
                int kk[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    en[u] = stg[wv][min(u0 + u, n - 1)];
                    const int p = (int)__umulhi(en[u].x, kmagic(d.K));
                    kk[u] = (int)(en[u].x - (unsigned)p * d.K);
                    x[u] = is[(size_t)p * d.W];
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (u0 + u < n) my[kk[u]] = fmaf(__uint_as_float(en[u].y), x[u], my[kk[u]]);
            }
        }
    }
    __syncthreads();
    const int nj = min(128, d.W - j0);
    float* span = dB + (size_t)g * d.h * d.W * d.K + ((size_t)i * d.W + j0) * d.K;
    for (int idx = threadIdx.x; idx < nj * d.K; idx += blockDim.x) {
        const int jj = idx / d.K, k = idx - jj * d.K;
        float v = 0.0f;
        for (int bb = 0; bb < d.B; bb++) v += accs[(size_t)bb * per + jj * (d.K + 1) + k];
        span[idx] = acc ? span[idx] + v : v;
    }
}

// up to 24 mini-batches: the per-read form (5.51 against 5.70 ms per step at 16, 6.93 against 7.02 at 24, even at 32, 13.75 against
// 13.34 at 64); more: a block per (columns, row, mini-batch) fills the chip by itself
static bool sp_wgrad_by_reads(Engine& e, const SpDims& d, int G, size_t lds) {
    constexpr int max_g = 20;      // (against the four-column form: +2 % at 16 mini-batches, -2 % at 24)
    if (G > max_g || d.B < 2 || d.B > 8 || lds > (size_t)150 << 10) return false;
    if (!e.wgrad_reads_attr_set) {
        (void)hipFuncSetAttribute((const void*)k_sp_wgrad_syn_reads, hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
        (void)hipFuncSetAttribute((const void*)k_sp_wgrad_ana_reads, hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10);
        e.wgrad_reads_attr_set = true;
    }
    return true;
}
// S2 / S3 for steps of many mini-batches with FOUR adjacent columns per lane: the one-column forms above spend ~20 instructions per entry and lane
// on one multiply-add (the walk is bound by instruction issue: 50 us at 64 mini-batches with 24 waves per CU); here a lane takes 16 bytes of the
// image row per entry and its LDS accumulators are float4 [K][64] (one 16-byte read-modify-write per entry), eight image values in flight.  A
// wave covers 256 columns; blocks are one wave.  The sums of a column run over the entries in list order, as before.
__global__ __launch_bounds__(64) void k_sp_wgrad_syn4(NzView nz, const float* __restrict__ dOut, float* __restrict__ dF, SpDims d) {
    extern __shared__ float4 acc4[];                     // [K][64]
    const int g = blockIdx.z, ip = blockIdx.y, lane = threadIdx.x, W4 = d.W >> 2, j4 = blockIdx.x * 64 + lane;
    const int jc = min(j4, W4 - 1);                      // clamp: out-of-range lanes compute, do not store
    for (int k = 0; k < d.K; k++) acc4[k * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < d.B; b++) {
        const int s = g * d.B + b;
        const float4* ds = (const float4*)(dOut + (size_t)s * d.c * d.W + (size_t)(d.h - 1 - ip) * d.W) + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        uint2 nx[8];                                     // the NEXT eight entries are asked for while this round's image values are on their way
#pragma unroll
        for (int u = 0; u < 8; u++) nx[u] = es[min(u, max(cnt - 1, 0))];
        for (int z = 0; z < cnt; z += 8) {               // wave-uniform
            uint2 en[8];
            float4 val[8];
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) en[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int p = (int)__umulhi(en[u].x, kmagic(d.K));
                kk[u] = (int)(en[u].x - (unsigned)p * d.K);
                val[u] = ds[(size_t)p * W4];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) nx[u] = es[min(z + 8 + u, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (z + u < cnt) {
                    const float v = __uint_as_float(en[u].y);
                    float4 a = acc4[kk[u] * 64 + lane];
                    a.x = fmaf(v, val[u].x, a.x), a.y = fmaf(v, val[u].y, a.y), a.z = fmaf(v, val[u].z, a.z), a.w = fmaf(v, val[u].w, a.w);
                    acc4[kk[u] * 64 + lane] = a;
                }
        }
    }
    if (j4 < W4) {
        float4* dFg = (float4*)(dF + (size_t)g * d.h * d.K * d.W + (size_t)ip * d.K * d.W) + j4;
        for (int k0 = 0; k0 < d.K; k0 += 8) {            // eight rows of the bank in flight (one at a time: K trips to memory in a row)
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = dFg[(size_t)min(k0 + u, d.K - 1) * W4];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < d.K) {
                    const float4 a = acc4[(k0 + u) * 64 + lane];
                    t[u].x += a.x, t[u].y += a.y, t[u].z += a.z, t[u].w += a.w;
                    dFg[(size_t)(k0 + u) * W4] = t[u];
                }
        }
    }
}
__global__ __launch_bounds__(64) void k_sp_wgrad_ana4(const float* __restrict__ img, NzView nz, float* __restrict__ dB, SpDims d, int acc) {
    extern __shared__ float4 acc4[];                     // [K][64]
    const int g = blockIdx.z, i = blockIdx.y, lane = threadIdx.x, W4 = d.W >> 2, j4 = blockIdx.x * 64 + lane;
    const int jc = min(j4, W4 - 1);
    for (int k = 0; k < d.K; k++) acc4[k * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < d.B; b++) {
        const int s = g * d.B + b;
        const float4* is = (const float4*)(img + (size_t)s * d.c * d.W + (size_t)i * d.W) + jc;
        const int cnt = nz.cnt[s];
        const uint2* es = nz.ent + (size_t)s * nz.cap;
        uint2 nx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) nx[u] = es[min(u, max(cnt - 1, 0))];
        for (int z = 0; z < cnt; z += 8) {
            uint2 en[8];
            float4 val[8];
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) en[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int p = (int)__umulhi(en[u].x, kmagic(d.K));
                kk[u] = (int)(en[u].x - (unsigned)p * d.K);
                val[u] = is[(size_t)p * W4];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) nx[u] = es[min(z + 8 + u, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (z + u < cnt) {
                    const float v = __uint_as_float(en[u].y);
                    float4 a = acc4[kk[u] * 64 + lane];
                    a.x = fmaf(v, val[u].x, a.x), a.y = fmaf(v, val[u].y, a.y), a.z = fmaf(v, val[u].z, a.z), a.w = fmaf(v, val[u].w, a.w);
                    acc4[kk[u] * 64 + lane] = a;
                }
        }
    }
    // the block's outputs [256 columns][K] are one contiguous span of dB[g][i][j][k]: written by all lanes in address order
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // (one wave: the LDS writes above are complete before the reads below)
    const int j0 = blockIdx.x * 256, nj = min(256, d.W - j0);
    float* span = dB + (size_t)g * d.h * d.W * d.K + ((size_t)i * d.W + j0) * d.K;
    const float* accf = (const float*)acc4;
    const int total = nj * d.K;
    for (int idx0 = lane; idx0 < total; idx0 += 64 * 8) {      // eight pieces in flight
        float old8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) old8[u] = (acc && idx0 + 64 * u < total) ? span[idx0 + 64 * u] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = idx0 + 64 * u;
            if (idx < total) {
                const int jj = (int)__umulhi((unsigned)idx, kmagic(d.K)), k = idx - jj * d.K;
                span[idx] = old8[u] + accf[(k * 64 + (jj >> 2)) * 4 + (jj & 3)];
            }
        }
    }
}
static bool sp_wgrad_four_columns(const SpDims& d, const float* a, const float* b) {
    return (d.W & 3) == 0 && (size_t)d.K * 64 * 16 <= 64 * 1024 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
}
static void launch_sp_wgrad_syn(Engine& e, NzView nz, const float* dOut, float* dF, const SpDims& d, int G) {
    const size_t lds = (size_t)d.B * d.K * 128 * 4;
    if (sp_wgrad_by_reads(e, d, G, lds))
        hipLaunchKernelGGL(k_sp_wgrad_syn_reads, dim3((d.W + 127) / 128, d.h, G), dim3(128 * d.B), lds, e.st, nz, dOut, dF, d);
    else if (sp_wgrad_four_columns(d, dOut, dF))
        hipLaunchKernelGGL(k_sp_wgrad_syn4, dim3((d.W + 255) / 256, d.h, G), dim3(64), (size_t)d.K * 64 * 16, e.st, nz, dOut, dF, d);
    else
        hipLaunchKernelGGL(k_sp_wgrad_syn, dim3((d.W + 127) / 128, d.h, G), dim3(128), (size_t)d.K * 128 * 4, e.st, nz, dOut, dF, d);
}
static void launch_sp_wgrad_ana(Engine& e, const float* img, NzView nz, float* dB, const SpDims& d, int G, int acc) {
    const size_t lds = (size_t)d.B * 128 * (d.K + 1) * 4;
    if (sp_wgrad_by_reads(e, d, G, lds))
        hipLaunchKernelGGL(k_sp_wgrad_ana_reads, dim3((d.W + 127) / 128, d.h, G), dim3(128 * d.B), lds, e.st, img, nz, dB, d, acc);
    else if (sp_wgrad_four_columns(d, img, dB))
        hipLaunchKernelGGL(k_sp_wgrad_ana4, dim3((d.W + 255) / 256, d.h, G), dim3(64), (size_t)d.K * 64 * 16, e.st, img, nz, dB, d, acc);
    else
        hipLaunchKernelGGL(k_sp_wgrad_ana, dim3((d.W + 127) / 128, d.h, G), dim3(128), (size_t)128 * (d.K + 1) * 4, e.st, img, nz, dB, d, acc);
}

// S4: out[s][p][k] += sum_{i,j} img[s][p+i][j] * Fk[g][k][j][i]   only at the listed (masked) entries
// (one wave per entry; Fk is the reference layout F (h,2M,1,K): i fastest)
__global__ __launch_bounds__(256) void k_sp_ana_masked(const float* __restrict__ img, const float* __restrict__ Fk,
                                                       NzView nz, float* __restrict__ out, SpDims d) {
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = nz.cnt[s];
    const float* is = img + (size_t)s * d.c * d.W;
    for (int z = blockIdx.x * 4 + (threadIdx.x >> 6); z < cnt; z += gridDim.x * 4) {   // wave-uniform
        const uint2 en = nz.ent[(size_t)s * nz.cap + z];
        const int p = (int)__umulhi(en.x, kmagic(d.K)), k = (int)(en.x - (unsigned)p * d.K);
        const float* fk = Fk + (size_t)(s / d.B) * d.ldf + (size_t)k * d.W * d.h;
        float a = 0.0f;
        if (d.h == 12 && (((uintptr_t)fk) & 15) == 0) {      // the lane's 12 filter taps as three 16-byte loads
            for (int j = lane; j < d.W; j += 64) {
                const float4* f4 = (const float4*)(fk + (size_t)j * 12);
                const float4 f0 = f4[0], f1 = f4[1], f2 = f4[2];
                const float* ip = is + (size_t)p * d.W + j;
                float x[12];
#pragma unroll
                for (int i = 0; i < 12; i++) x[i] = ip[(size_t)i * d.W];
                a = fmaf(x[0], f0.x, a), a = fmaf(x[1], f0.y, a), a = fmaf(x[2], f0.z, a), a = fmaf(x[3], f0.w, a);
                a = fmaf(x[4], f1.x, a), a = fmaf(x[5], f1.y, a), a = fmaf(x[6], f1.z, a), a = fmaf(x[7], f1.w, a);
                a = fmaf(x[8], f2.x, a), a = fmaf(x[9], f2.y, a), a = fmaf(x[10], f2.z, a), a = fmaf(x[11], f2.w, a);
            }
        } else {
            for (int j = lane; j < d.W; j += 64)
                for (int i = 0; i < d.h; i++) a = fmaf(is[(size_t)(p + i) * d.W + j], fk[(size_t)j * d.h + i], a);
        }
        for (int dd = 32; dd >= 1; dd >>= 1) a += __shfl_xor(a, dd);
        if (lane == 0) out[(size_t)s * d.l * d.K + en.x] += a;
    }
}

NzView Engine::nz_build(const float* data, int S, int n_per) {
    NzView v{nullptr, nullptr, n_per};
    int* cnt = (int*)arena.alloc((size_t)S + 64);
    uint2* ent = (uint2*)arena.alloc((size_t)S * n_per * 2);
    if (!cnt || !ent) {
        failed = true;
        return v;
    }
    hipLaunchKernelGGL(k_build_nz, dim3(S), dim3(S >= 1024 ? 256 : 1024), 0, st, data, n_per, cnt, ent);
    v.cnt = cnt;
    v.ent = ent;
    return v;
}
NzView Engine::nz_of(Tensor t, int S) {
    const int n_per = (int)(t->n / S);
    if (!t->nz_cnt) {
        NzView v = nz_build(t->v, S, n_per);
        t->nz_cnt = const_cast<int*>(v.cnt);
        t->nz_ent = const_cast<uint2*>(v.ent);
    }
    return NzView{t->nz_cnt, t->nz_ent, n_per};
}
NzView Engine::nz_of_mask(Tensor t, int S) {
    const int n_per = (int)(t->n / S);
    if (!t->gm_cnt) {
        NzView v = nz_build(t->gmask, S, n_per);
        t->gm_cnt = const_cast<int*>(v.cnt);
        t->gm_ent = const_cast<uint2*>(v.ent);
    }
    return NzView{t->gm_cnt, t->gm_ent, n_per};
}

// 32 waves per read walk its entry list (any length)
static void launch_sp_ana_masked(hipStream_t st, const float* img, const float* Fk, const NzView& nz, float* out,
                                 const SpDims& d, int) {
    hipLaunchKernelGGL(k_sp_ana_masked, dim3(8, d.S), dim3(256), 0, st, img, Fk, nz, out, d);
}

// S1 by output rows: a block owns 32 rows of one read's image and every thread 4 adjacent columns; the entries that
// reach row r are the (position-sorted) run with p in (r - h, r], found once per row by bisection.  The image is
// written (or accumulated) once with 16-byte accesses, the contributions of a row meet in registers in entry order -
// the same sums as the ring of k_sp_syn, without its serial walk over the entries.
// RB rows per block: 32 when there are reads enough to fill the chip, 2 for the reference's 6-read steps (one output per thread)
template <int RB>
__global__ __launch_bounds__(256) void k_sp_syn_rows(NzView nz, const float* __restrict__ FAf, float* __restrict__ out, SpDims d, int acc) {
    constexpr int EC = 256;                        // decoded entries kept in LDS (more: decoded on the fly)
    __shared__ int lo[RB], hi[RB];
    __shared__ int ep[EC], ek[EC];
    __shared__ float ev[EC];
    __shared__ unsigned ex[EC];
    const int s = blockIdx.y, r0 = blockIdx.x * RB, tid = threadIdx.x;
    const uint2* es = nz.ent + (size_t)s * nz.cap;
    // The usual read has a few dozen entries: the whole list comes to LDS in one round of loads issued beside the load of
    // its length, and the bisections run there (six dependent trips to memory per row before).
    uint2 mine = make_uint2(0, 0);
    if (tid < EC && tid < nz.cap) mine = es[tid];
    const int cnt = nz.cnt[s];
    const bool all = cnt <= EC;
    if (all && tid < cnt) {
        const int p = (int)__umulhi(mine.x, kmagic(d.K));
        ex[tid] = mine.x, ep[tid] = p, ek[tid] = (int)(mine.x - (unsigned)p * d.K), ev[tid] = __uint_as_float(mine.y);
    }
    if (all) __syncthreads();
    if (tid < 2 * RB) {                            // first entry with p >= r - h + 1 (lo) / p >= r + 1 (hi)
        const int r = r0 + (tid % RB);
        const long key = ((long)(tid < RB ? r - d.h + 1 : r + 1)) * d.K;
        int a = 0, b = cnt;
        while (a < b) {
            const int m = (a + b) >> 1;
            if ((long)(all ? ex[m] : es[m].x) < key) a = m + 1;
            else b = m;
        }
        (tid < RB ? lo : hi)[tid % RB] = a;
    }
    __syncthreads();
    const int nrow = min(RB, d.c - r0);
    const int zb = all ? 0 : lo[0], ne = hi[nrow - 1] - zb;
    const bool cached = all || ne <= EC;
    if (!all && cached) {
        for (int i = tid; i < ne; i += 256) {
            const uint2 en = es[zb + i];
            const int p = (int)__umulhi(en.x, kmagic(d.K));
            ep[i] = p, ek[i] = (int)(en.x - (unsigned)p * d.K), ev[i] = __uint_as_float(en.y);
        }
    }
    __syncthreads();
    const int W4 = d.W >> 2;
    const float4* F4 = (const float4*)(FAf + (size_t)(s / d.B) * d.ldf);
    float4* o4 = (float4*)(out + ((size_t)s * d.c + r0) * d.W);
    int row = tid / W4, c4 = tid - row * W4;
    const int drow = 256 / W4, dc = 256 - drow * W4;
    for (int idx = tid; idx < nrow * W4; idx += 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = r0 + row;
        const int zhi = hi[row];
        for (int z = lo[row]; z < zhi; z += 4) {   // four filter rows in flight: the loop is a chain of dependent trips to memory otherwise
            float4 f[4];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = 0.0f, f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (z + u < zhi) {
                    int p, k;
                    if (cached) {
                        p = ep[z + u - zb], k = ek[z + u - zb], v[u] = ev[z + u - zb];
                    } else {
                        const uint2 en = es[z + u];
                        p = (int)__umulhi(en.x, kmagic(d.K)), k = (int)(en.x - (unsigned)p * d.K), v[u] = __uint_as_float(en.y);
                    }
                    f[u] = F4[((size_t)(d.h - 1 - (r - p)) * d.K + k) * W4 + c4];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (z + u < zhi) a.x = fmaf(v[u], f[u].x, a.x), a.y = fmaf(v[u], f[u].y, a.y), a.z = fmaf(v[u], f[u].z, a.z), a.w = fmaf(v[u], f[u].w, a.w);
        }
        if (acc) {
            const float4 t = o4[idx];
            a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
        }
        o4[idx] = a;
        row += drow, c4 += dc;
        if (c4 >= W4) c4 -= W4, row++;
    }
}

static void launch_sp_syn(hipStream_t st, const NzView& nz, const float* FAf, float* out, const SpDims& d, int acc) {
    if ((d.W & 3) == 0 && d.W <= 1024 && (d.ldf & 3) == 0 && ((((uintptr_t)FAf) | ((uintptr_t)out)) & 15) == 0) {
        if (d.S >= 192) hipLaunchKernelGGL(k_sp_syn_rows<32>, dim3((d.c + 31) / 32, d.S), dim3(256), 0, st, nz, FAf, out, d, acc);
        else hipLaunchKernelGGL(k_sp_syn_rows<2>, dim3((d.c + 1) / 2, d.S), dim3(256), 0, st, nz, FAf, out, d, acc);
        return;
    }
    if (d.h == 12) hipLaunchKernelGGL(k_sp_syn<12>, dim3((d.W + 127) / 128, d.S), dim3(128), 0, st, nz, FAf, out, d, acc);
    else if (d.h <= 16) hipLaunchKernelGGL(k_sp_syn<0>, dim3((d.W + 127) / 128, d.S), dim3(128), 0, st, nz, FAf, out, d, acc);
    else hipLaunchKernelGGL(k_sp_syn_any, dim3((d.W + 127) / 128, d.S), dim3(128), 0, st, nz, FAf, out, d, acc);
}

Tensor Engine::sp_syn(Tensor T, Tensor FAf, Tensor Fk, const SpDims& d) {
    Tensor out = make((size_t)d.S * d.c * d.W, T->needs_grad || FAf->needs_grad);
    if (failed) return out;
    NzView nz = nz_of(T, d.S);
    if (failed) return out;
    launch_sp_syn(st, nz, FAf->v, out->v, d, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, T, FAf, Fk, d, nz]() {
            if (!out->g) return;
            if (T->needs_grad) {
                float* dT = grad(T);
                if (dT && T->gmask) {
                    NzView mz = nz_of_mask(T, d.S);
                    if (!failed) launch_sp_ana_masked(st, out->g, Fk->v, mz, dT, d, d.l * d.K);
                } else if (dT) {   // no mask known: dense adjoint
                    ToepGeom gm{d.S, d.c, d.h * d.K, d.W, d.K, -(d.h - 1) * d.K, d.l * d.K, (int64_t)d.l * d.K,
                                (int64_t)d.c * d.W, d.B, d.ldf};
                    toep_adjoint_a(*this, out->g, FAf->v, dT, gm);
                }
            }
            if (FAf->needs_grad) {
                float* dF = grad(FAf);
                const int G = d.S / d.B;
                if (!dF) return;
                if (d.ldf != 0 || G == 1) {
                    launch_sp_wgrad_syn(*this, nz, out->g, dF, d, G);
                } else {
                    const size_t per = (size_t)d.h * d.K * d.W;
                    float* tmp = arena.alloc(per * G);
                    if (!tmp) {
                        failed = true;
                        return;
                    }
                    dev_zero(st, tmp, per * G);
                    launch_sp_wgrad_syn(*this, nz, out->g, tmp, d, G);
                    hipLaunchKernelGGL(k_sum_groups, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, tmp, per, G, dF);
                }
            }
        });
    return out;
}

Tensor Engine::ana_sp(Tensor img, Tensor FA, Tensor FAf, const SpDims& d, const ToepGeom& gm) {
    Tensor out = make((size_t)d.S * d.l * d.K, img->needs_grad || FA->needs_grad);
    if (failed) return out;
    launch_toep(*this, img->v, FA->v, out->v, gm, 0);       // dense forward: every position is needed
    if (recording && out->needs_grad)
        tape.push_back([this, out, img, FA, FAf, d]() {
            if (!out->g) return;
            // d(out) is non-zero only where the top-q masks let it through
            NzView gz = nz_build(out->g, d.S, d.l * d.K);
            if (failed) return;
            if (img->needs_grad) {
                int a = 1;
                float* di = grad_first(img, a);        // the synthesis writes every row: no zero fill on first use
                if (di) launch_sp_syn(st, gz, FAf->v, di, d, a);
            }
            if (FA->needs_grad) {
                float* dB = grad(FA);
                const int G = d.S / d.B;
                if (!dB) return;
                if (d.ldf != 0 || G == 1) {
                    launch_sp_wgrad_ana(*this, img->v, gz, dB, d, G, 1);
                } else {
                    const size_t per = (size_t)d.h * d.W * d.K;
                    float* tmp = arena.alloc(per * G);
                    if (!tmp) {
                        failed = true;
                        return;
                    }
                    launch_sp_wgrad_ana(*this, img->v, gz, tmp, d, G, 0);
                    hipLaunchKernelGGL(k_sum_groups, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, tmp, per, G, dB);
                }
            }
        });
    return out;
}

Tensor Engine::wgrad_sp(Tensor img, Tensor T, const SpDims& d) {
    const int G = d.S / d.B;
    Tensor out = make((size_t)G * d.h * d.W * d.K, img->needs_grad || T->needs_grad);
    if (failed) return out;
    NzView nz = nz_of(T, d.S);
    if (failed) return out;
    launch_sp_wgrad_ana(*this, img->v, nz, out->v, d, G, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, img, T, d, G, nz]() {
            if (!out->g) return;
            const size_t per = (size_t)d.h * d.W * d.K;
            SpDims dg = d;
            dg.ldf = (int64_t)per;                    // the "filter" of both adjoints is dOut, one slice per group
            if (img->needs_grad) {                    // dimg[r][j] += sum_nz v * dOut[r - p][j][k]  = S1 with flipT(dOut)
                int a = 1;
                float* di = grad_first(img, a);
                float* tmp = arena.alloc(per * G);
                if (!di || !tmp) {
                    failed = failed || !tmp;
                    return;
                }
                launch_flipT(st, out->g, G, d.h, d.W, d.K, tmp, 0);
                launch_sp_syn(st, nz, tmp, di, dg, a);
            }
            if (T->needs_grad) {
                float* dT = grad(T);
                if (!dT) return;
                if (T->gmask) {                       // dT[p][k] += sum_{i,j} img[p+i][j] dOut[i][j][k], masked
                    float* tk = arena.alloc(per * G);
                    if (!tk) {
                        failed = true;
                        return;
                    }
                    launch_swap02(st, out->g, G, d.h, d.W, d.K, tk, 0);
                    NzView mz = nz_of_mask(T, d.S);
                    if (!failed) launch_sp_ana_masked(st, img->v, tk, mz, dT, dg, d.l * d.K);
                } else {
                    ToepGeom gm{d.S, d.l, d.h * d.W, d.K, d.W, 0, d.c * d.W, (int64_t)d.c * d.W, (int64_t)d.l * d.K, d.B,
                                (int64_t)per};
                    launch_toep(*this, img->v, out->g, dT, gm, 1);
                }
            }
        });
    return out;
}

// ---------------------------------------------------------------------------------------------
// selections
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint32_t fkey(float f) {   // order-preserving float -> uint
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// k-th smallest (0-based) key among the elements accepted by `pred`, by 4 radix passes of 8 bits.
// Block-cooperative; returns the key to every thread.
template <class Pred>
static __device__ uint32_t block_radix_select(const float* x, int n, uint32_t k, Pred pred, uint32_t* hist, uint32_t* sh) {
    uint32_t prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        if (shift == 24) {
            // the leading digit (sign + 7 exponent bits) takes a handful of values over a whole read: one LDS atomic per element
            // would queue thousands of adds on two or three bins, so each wave counts its lanes per distinct digit first
            // (as many ballots as the wave holds distinct digits) and adds the counts
            const int nround = (n + (int)blockDim.x - 1) / (int)blockDim.x;
            for (int rd = 0; rd < nround; rd++) {
                const int i = rd * (int)blockDim.x + (int)threadIdx.x;
                const float v = i < n ? x[i] : 0.0f;
                bool todo = i < n && pred(v);
                const uint32_t d = fkey(v) >> 24;
                while (true) {
                    const unsigned long long live = __ballot(todo);
                    if (!live) break;
                    const uint32_t lead = (uint32_t)__shfl((int)d, __builtin_ctzll(live));
                    const unsigned long long same = __ballot(todo && d == lead);
                    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(live)) atomicAdd(&hist[lead], (uint32_t)__builtin_popcountll(same));
                    if (d == lead) todo = false;
                }
            }
        } else {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const float v = x[i];
                if (!pred(v)) continue;
                const uint32_t key = fkey(v);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                     // first bin whose running count exceeds k: 4 bins per lane, wave scan
            const int l = threadIdx.x;
            const uint32_t c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
            const uint32_t sum = c0 + c1 + c2 + c3;
            uint32_t incl = sum;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d);
                if (l >= d) incl += t;
            }
            const uint32_t before = incl - sum;
            const uint32_t total = __shfl(incl, 63);
            if ((before <= k && k < incl) || (l == 63 && k >= total)) {
                uint32_t run = before;
                int b = 4 * l;
                if (run + c0 <= k) {
                    run += c0, b++;
                    if (run + c1 <= k) {
                        run += c1, b++;
                        if (run + c2 <= k) {
                            run += c2, b++;
                            if (run + c3 <= k) run += c3, b++;   // only when nothing exceeds k (b = 256, as the serial walk)
                        }
                    }
                }
                sh[0] = (uint32_t)b;
                sh[1] = run;
            }
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 0xffu << shift;
        k -= sh[1];
        __syncthreads();
    }
    return prefix;
}
static __device__ __forceinline__ float unkey(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// generate_bitmat (model.jl:181-187): per sequence, q-th largest of its l*K values; bitmat = X >= that value
__global__ __launch_bounds__(1024) void k_topq_mask(const float* X, float* bitmat, int n, int q) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[2];
    const float* xs = X + (size_t)blockIdx.x * n;
    const uint32_t key = block_radix_select(xs, n, (uint32_t)(n - q), [](float) { return true; }, hist, sh);
    const float thr = unkey(key);
    for (int i = threadIdx.x; i < n; i += blockDim.x) bitmat[(size_t)blockIdx.x * n + i] = xs[i] >= thr ? 1.0f : 0.0f;
}
void topq_mask(hipStream_t st, const float* X, float* bitmat, int S, int n_per_seq, int q) {
    // (16 waves per read up to 1024 reads: a block's phases are latency-bound, and 384 blocks of 4 waves left the chip idle - 28.8 -> 16.5 us for k_x_project)
    hipLaunchKernelGGL(k_topq_mask, dim3(S), dim3(S >= 1024 ? 256 : 1024), 0, st, X, bitmat, n_per_seq, q);
}

// update_X's tail in one launch (model.jl:253 then project_X, :181-192): the gradient step Xu = X - ost * xg (xg == null: Xu = X,
// the warm-up's projection), the q-th largest of each read, bitmat = Xu >= that value, the projected codes bitmat .* Xu, and the
// two entry lists every sparse kernel behind them walks - of the codes (values) and of the mask (ones) - which were launches of
// their own (k_x_step, k_topq_mask, k_maskmul, k_build_nz twice).  One block per read, Xu staged in LDS.
__global__ __launch_bounds__(1024) void k_x_project(const float* __restrict__ X, const float* __restrict__ xg, const float* __restrict__ ost, int n,
                                                    int q, float* __restrict__ out, float* __restrict__ bit, int* __restrict__ cnt,
                                                    uint2* __restrict__ ent, int* __restrict__ mcnt, uint2* __restrict__ ment, float scale) {
    extern __shared__ float xs[];                  // [n]
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[2];
    __shared__ int wc[2][16];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const size_t base = (size_t)s * n;
    const float o = xg ? *ost : 0.0f;
    for (int i = tid; i < n; i += blockDim.x) xs[i] = xg ? X[base + i] - xg[base + i] * o : (scale == 1.0f ? X[base + i] : scale * X[base + i] + 0.0f + 0.0f);
    __syncthreads();
    const uint32_t key = block_radix_select(xs, n, (uint32_t)(n - q), [](float) { return true; }, hist, sh);
    const float thr = unkey(key);
    // every wave owns a contiguous slice of the read: values out and counts, meet once, then the entries in ascending order
    const int per = (((n + nw - 1) / nw) + 63) & ~63, lo = min(n, wv * per), hi = min(n, lo + per);
    int c = 0, cm = 0;
    for (int e0 = lo; e0 < hi; e0 += 64) {
        const int e = e0 + lane;
        float v = 0.0f, m = 0.0f;
        if (e < hi) {
            m = xs[e] >= thr ? 1.0f : 0.0f;
            v = m * xs[e];
            out[base + e] = v;
            bit[base + e] = m;
        }
        c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v != 0.0f));
        cm += __builtin_popcountll(__builtin_amdgcn_ballot_w64(m != 0.0f));
    }
    if (lane == 0) wc[0][wv] = c, wc[1][wv] = cm;
    __syncthreads();
    int b0 = 0, b1 = 0;
    for (int w = 0; w < wv; w++) b0 += wc[0][w], b1 += wc[1][w];
    uint2* es = ent + base;
    uint2* ms = ment + base;
    for (int e0 = lo; e0 < hi; e0 += 64) {
        const int e = e0 + lane;
        float v = 0.0f, m = 0.0f;
        if (e < hi) {
            m = xs[e] >= thr ? 1.0f : 0.0f;
            v = m * xs[e];
        }
        const uint64_t hv = __builtin_amdgcn_ballot_w64(v != 0.0f), hm = __builtin_amdgcn_ballot_w64(m != 0.0f);
        const uint64_t below = (1ull << lane) - 1ull;
        if (v != 0.0f) es[b0 + __builtin_popcountll(hv & below)] = make_uint2((unsigned)e, __float_as_uint(v));
        if (m != 0.0f) ms[b1 + __builtin_popcountll(hm & below)] = make_uint2((unsigned)e, __float_as_uint(m));
        b0 += __builtin_popcountll(hv);
        b1 += __builtin_popcountll(hm);
    }
    if (tid == 0) {
        int t0 = 0, t1 = 0;
        for (int w = 0; w < nw; w++) t0 += wc[0][w], t1 += wc[1][w];
        cnt[s] = t0;
        mcnt[s] = t1;
    }
}
// VJP: g = bitmat .* d out;  dX (+)= g;  d xg (+)= -ost * g;  d ost += -sum(g .* xg)
template <int V>   // V = 4: 16-byte accesses (n % 4 == 0, 16-byte aligned tensors); V = 1: scalar
__global__ void k_x_project_bwd(const float* go, const float* bit, const float* xg, const float* ost, size_t n, float* dX, int aX, float* dxg,
                                int axg, float* dost, float scale) {
    struct VF {
        float e[V];
    };
    auto ld = [&](const float* q, size_t i) {
        VF r;
        if (V == 4) {
            const float4 x = *(const float4*)(q + i);
            r.e[0] = x.x, r.e[1 % V] = x.y, r.e[2 % V] = x.z, r.e[3 % V] = x.w;
        } else {
            r.e[0] = q[i];
        }
        return r;
    };
    auto stv = [&](float* q, size_t i, const VF& r) {
        if (V == 4) *(float4*)(q + i) = make_float4(r.e[0], r.e[1 % V], r.e[2 % V], r.e[3 % V]);
        else q[i] = r.e[0];
    };
    const float o = ost ? *ost : 0.0f;
    double so = 0;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
        VF z{};
        const VF vb = ld(bit, i), vg = ld(go, i), vx = xg ? ld(xg, i) : z;
        VF oX = (dX && aX) ? ld(dX, i) : z, oxg = (dxg && axg) ? ld(dxg, i) : z;
#pragma unroll
        for (int u = 0; u < V; u++) {
            const float g = vb.e[u] * vg.e[u];
            oX.e[u] = oX.e[u] + (scale == 1.0f ? g : scale * g);
            oxg.e[u] = oxg.e[u] - o * g;
            if (xg) so -= (double)g * (double)vx.e[u];
        }
        if (dX) stv(dX, i, oX);
        if (dxg) stv(dxg, i, oxg);
    }
    if (!dost) return;
    for (int d = 32; d >= 1; d >>= 1) so += __shfl_xor(so, d);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = so;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dost, (float)(red[0] + red[1] + red[2] + red[3]));
}

Tensor Engine::x_project(Tensor X, Tensor xg, Tensor ost, int S, int q, float scale) {
    const int n = (int)(X->n / S);
    static const bool off = getenv("MOTIFS_NO_X_PROJECT") != nullptr;
    if (off || (size_t)n * 4 > ((size_t)48 << 10)) {          // the separate launches
        Tensor Xu = xg ? x_step(X, xg, ost) : (scale == 1.0f ? X : lin(X, scale, nullptr, 0.0f, 0.0f));
        Tensor bitm = make(Xu->n, false);
        if (failed) return Xu;
        topq_mask(st, Xu->v, bitm->v, S, n, q);
        Tensor P = maskmul(Xu, bitm->v, 1.0f);
        P->gmask = bitm->v;
        return P;
    }
    Tensor out = make(X->n, X->needs_grad || (xg && (xg->needs_grad || ost->needs_grad)));
    Tensor bitm = make(X->n, false);
    int* c = (int*)arena.alloc(2 * ((size_t)S + 64));
    uint2* en = (uint2*)arena.alloc((size_t)S * n * 4);
    if (failed || !c || !en) {
        failed = true;
        return out;
    }
    int* mc = c + S + 64;
    uint2* men = en + (size_t)S * n;
    hipLaunchKernelGGL(k_x_project, dim3(S), dim3(S >= 1024 ? 256 : 1024), (size_t)n * 4, st, X->v, xg ? xg->v : nullptr, xg ? ost->v : nullptr, n, q,
                       out->v, bitm->v, c, en, mc, men, scale);
    out->gmask = bitm->v;          // every gradient into the projected codes passes this mask on its way back
    out->nz_cnt = c, out->nz_ent = en;
    out->gm_cnt = mc, out->gm_ent = men;
    if (recording && out->needs_grad)
        tape.push_back([this, out, X, xg, ost, bitm, scale]() {
            if (!out->g) return;
            int a0 = 1, a1 = 1;
            float* d0 = X->needs_grad ? grad_first(X, a0) : nullptr;
            float* d1 = xg && xg->needs_grad ? grad_first(xg, a1) : nullptr;
            float* dq = xg && ost->needs_grad ? grad(ost) : nullptr;
            if (failed) return;
            auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
            if ((out->n & 3) == 0 && al16(out->g) && al16(bitm->v) && al16(xg ? xg->v : nullptr) && al16(d0) && al16(d1))
                hipLaunchKernelGGL(k_x_project_bwd<4>, dim3(nblocks(out->n / 4, 256, 512)), dim3(256), 0, st, out->g, bitm->v, xg ? xg->v : nullptr,
                                   xg ? ost->v : nullptr, out->n, d0, a0, d1, a1, dq, scale);
            else
                hipLaunchKernelGGL(k_x_project_bwd<1>, dim3(nblocks(out->n, 256, 1024)), dim3(256), 0, st, out->g, bitm->v, xg ? xg->v : nullptr,
                                   xg ? ost->v : nullptr, out->n, d0, a0, d1, a1, dq, scale);
        });
    return out;
}

// create_ZY_mask (model.jl:194-204): median of the strictly positive entries of the whole mini-batch
// (mean of the two middle values for an even count: Statistics.middle(a, b) = a/2 + b/2); mask = ZY >= median.
// No positive entry -> the reference skips the mask (:209); that is mask == 1 here.
// Multi-block radix select: 3 passes of (histogram over all blocks, digit choice per group) over digits of
// 11, 11 and 10 bits (positive floats order like their bit patterns); the lower and the upper middle element are
// tracked side by side.
struct MedState {
    uint32_t cnt, pref[2], k[2];
};
constexpr int MED_BINS = 2048;
static __host__ __device__ __forceinline__ int med_shift(int pass) { return pass == 0 ? 21 : pass == 1 ? 10 : 0; }
static __host__ __device__ __forceinline__ int med_bits(int pass) { return pass == 2 ? 10 : 11; }

// The workspace of one median_threshold call (zeroed by the caller): three MedState arrays [G] - the select's state before pass
// 0 (zeros), after pass 0 and after pass 1 - then three histograms [G][2][MED_BINS], one per pass (never re-zeroed, never reused).
static __host__ __device__ __forceinline__ size_t med_states_bytes(int G) { return (((size_t)3 * G * sizeof(MedState)) + 255) & ~(size_t)255; }

// histogram of digit `pass` over the block's share of group g, for the entries whose higher digits are p0 (ranks' lower middle)
// and p1 (upper middle); added to hist[g] at the end
static __device__ __forceinline__ void med_hist_block(const float* __restrict__ x, int n, uint32_t p0, uint32_t p1, uint32_t* hist, int pass,
                                                      uint32_t (*h)[MED_BINS]) {
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * MED_BINS; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int shift = med_shift(pass);
    const uint32_t dmask = (1u << med_bits(pass)) - 1u;
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + med_bits(pass)));
    const float* xs = x + (size_t)g * n;
    // consecutive entries of a lane often share the leading digit: runs are counted in registers, one atomic per run.  While both middle elements
    // share their higher digits (p0 == p1: nearly always) one histogram serves both (med_select_core reads it twice): half the LDS atomics and
    // half of the flush - the second pass's blocks each leave ~2 400 non-zero counters, 10 M global atomics per launch at 64 mini-batches,
    // which is what made it 65 us against the third pass's 28
    const bool same = pass == 0 || p0 == p1;
    uint32_t run_d = 0xffffffffu, run_c = 0;
    auto take = [&](float v) {
        if (!(v > 0.0f)) return;
        const uint32_t key = __float_as_uint(v);
        const uint32_t d = (key >> shift) & dmask;
        if ((key & mask) == p0) {
            if (d == run_d) {
                run_c++;
            } else {
                if (run_c) atomicAdd(&h[0][run_d], run_c);
                run_d = d, run_c = 1;
            }
        }
        if (pass != 0 && !same && (key & mask) == p1) atomicAdd(&h[1][d], 1u);
    };
    if ((n & 3) == 0 && (((uintptr_t)xs) & 15) == 0) {
        const float4* x4 = (const float4*)xs;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gridDim.x * 256) {
            const float4 v = x4[i];
            take(v.x), take(v.y), take(v.z), take(v.w);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) take(xs[i]);
    }
    if (run_c) atomicAdd(&h[0][run_d], run_c);
    __syncthreads();
    for (int i = threadIdx.x; i < (same ? 1 : 2) * MED_BINS; i += 256) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&hist[(size_t)g * 2 * MED_BINS + i], c);
    }
}

// One digit choice on a group's histogram h ([MED_BINS] shared by both ranks in pass 0, [2][MED_BINS] later): the first bin whose
// running count exceeds k (the last bin if none does); 256 threads x 8 bins.  sst (LDS) holds the state before and after;
// nothing else is written, so every block of the next histogram pass can make the choice for itself.
static __device__ void med_select_core(MedState& sst, const uint32_t* h, int pass, uint32_t* part) {
    const int tid = threadIdx.x, shift = med_shift(pass);
    __syncthreads();                                // sst as the caller left it is visible
    // while the lower and the upper middle element share their higher digits they look at the same entries: one histogram was counted for both
    const bool one_hist = pass == 0 || sst.pref[0] == sst.pref[1];
    __syncthreads();                                // (read before the first selection changes pref[0])
    for (int sel = 0; sel < 2; sel++) {
        const uint32_t* hh = h + (one_hist ? 0 : sel * MED_BINS);
        uint32_t loc[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            loc[j] = hh[tid * 8 + j];
            sum += loc[j];
        }
        // inclusive scan over the 256 threads: inside each wave by shuffles, the four wave totals through LDS
        uint32_t incl = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if ((tid & 63) >= d) incl += v;
        }
        __syncthreads();                            // part[] of the previous selection has been read
        if ((tid & 63) == 63) part[tid >> 6] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < (tid >> 6); w++) woff += part[w];
        const uint32_t total = part[0] + part[1] + part[2] + part[3], before = woff + incl - sum;
        if (pass == 0 && sel == 0 && tid == 0) {
            sst.cnt = total;
            sst.pref[0] = sst.pref[1] = 0;
            sst.k[0] = total ? (total - 1) / 2 : 0;
            sst.k[1] = total / 2;
        }
        __syncthreads();
        const uint32_t k = sst.k[sel];
        const bool any = sst.cnt != 0;
        // owner: the thread whose bins hold the k-th element; thread 255 also takes "none exceeds"
        const bool owner = any && ((before <= k && k < before + sum) || (tid == 255 && k >= total));
        __syncthreads();
        if (owner) {
            uint32_t run = before;
            int b = tid * 8;
            for (int j = 0; j < 8; j++, b++) {
                if (b == MED_BINS - 1 || run + loc[j] > k) break;
                run += loc[j];
            }
            sst.pref[sel] |= (uint32_t)b << shift;
            sst.k[sel] = k - run;
        }
        __syncthreads();
    }
}

// pass 0 on its own (when the kernel that wrote the codes did not count their top digits on the way)
__global__ __launch_bounds__(256) void k_med_hist0(const float* __restrict__ x, int n, uint32_t* hist0) {
    __shared__ uint32_t h[2][MED_BINS];
    med_hist_block(x, n, 0u, 0u, hist0, 0, h);
}
// Passes 1 and 2: every block first makes the digit choice of the pass before from that pass's (complete) histogram - the
// choice used to be a launch of its own between two histogram launches, five launches per median - then counts its share of
// the group.  Block 0 leaves the state for the next launch.
__global__ __launch_bounds__(256) void k_med_pass(const float* __restrict__ x, int n, const MedState* st_in, const uint32_t* hist_prev,
                                                  uint32_t* hist_cur, int pass, MedState* st_out) {
    __shared__ uint32_t h[2][MED_BINS];
    __shared__ uint32_t part[256];
    __shared__ MedState sst;
    const int g = blockIdx.y;
    if (threadIdx.x == 0) sst = st_in[g];
    med_select_core(sst, hist_prev + (size_t)g * 2 * MED_BINS, pass - 1, part);
    const uint32_t p0 = sst.pref[0], p1 = sst.pref[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) st_out[g] = sst;
    med_hist_block(x, n, p0, p1, hist_cur, pass, h);
}
// the last digit choice and the median itself
__global__ __launch_bounds__(256) void k_med_final(const MedState* st_in, const uint32_t* hist2, float* thr) {
    __shared__ uint32_t part[256];
    __shared__ MedState sst;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) sst = st_in[g];
    med_select_core(sst, hist2 + (size_t)g * 2 * MED_BINS, 2, part);
    if (threadIdx.x == 0) {
        float med = -INFINITY;                         // no positive entry: everything passes
        if (sst.cnt) {
            const float lo = __uint_as_float(sst.pref[0]), hi = __uint_as_float(sst.pref[1]);
            med = (sst.cnt & 1u) ? lo : lo / 2 + hi / 2;
        }
        thr[g] = med;
    }
}

// (A pass as ONE launch with the choice AFTER the counts - every block takes a ticket after its atomics and the block that draws
// the last one makes the digit choice - was measured and lost: the fence in front of the ticket is an L2 write-back per block;
// 17.5 us per pass against 7 + 5 for the two launches at one mini-batch, 420 us against 39 at 64.)

static_assert(MED_BINS == ZH_BINS, "the fused first pass (zy_step kernels) fills the same histogram");
uint32_t* median_hist_ptr(void* workspace, int G) { return (uint32_t*)((char*)workspace + med_states_bytes(G)); }   // the pass-0 histogram
void median_threshold(hipStream_t st, const float* ZY, float* thr, int G, int n_per_group, void* workspace, bool have_pass0) {
    MedState* s0 = (MedState*)workspace;               // the workspace arrives zeroed (Engine::zeros)
    MedState *s1 = s0 + G, *s2 = s1 + G;
    const size_t hsz = (size_t)G * 2 * MED_BINS;
    uint32_t* h0 = median_hist_ptr(workspace, G);
    uint32_t *h1 = h0 + hsz, *h2 = h1 + hsz;
    // blocks per group: 64 when the groups fill the chip, up to 256 for a step of few mini-batches
    const unsigned nb = (unsigned)std::min<size_t>((n_per_group + 256 * 16 - 1) / (256 * 16), G >= 4 ? 64 : 256);
    if (!have_pass0) hipLaunchKernelGGL(k_med_hist0, dim3(nb, G), dim3(256), 0, st, ZY, n_per_group, h0);   // else: counted by the kernel that wrote the codes
    // the second digit's pass leaves up to 2048 non-zero counters per block to add to the group's: with the chip full anyway, half the blocks
    // (50.7 -> 37.1 us at 64 mini-batches; a quarter: 41.1)
    const unsigned nb1 = (size_t)G * nb >= 2048 ? nb / 2 : nb;
    hipLaunchKernelGGL(k_med_pass, dim3(nb1, G), dim3(256), 0, st, ZY, n_per_group, s0, h0, h1, 1, s1);
    hipLaunchKernelGGL(k_med_pass, dim3(nb, G), dim3(256), 0, st, ZY, n_per_group, s1, h1, h2, 2, s2);
    hipLaunchKernelGGL(k_med_final, dim3(G), dim3(256), 0, st, s2, h2, thr);
}
size_t median_workspace_bytes(int G) { return med_states_bytes(G) + (size_t)3 * G * 2 * MED_BINS * 4; }

__global__ void k_onehot(const uint8_t* codes, int pitch, float* S, int nseq, int L) {
    const size_t total = (size_t)nseq * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t s = i / L;
        const int p = (int)(i - s * L);
        const int c = codes[s * pitch + p];
        ((float4*)S)[i] = make_float4(c == 0, c == 1, c == 2, c == 3);
    }
}
void onehot_from_codes(hipStream_t st, const uint8_t* codes, int pitch, float* S, int nseq, int L) {
    hipLaunchKernelGGL(k_onehot, dim3(nblocks((size_t)nseq * L)), dim3(256), 0, st, codes, pitch, S, nseq, L);
}

// Flux 0.14 AdaBelief: m, s running moments; the gradient is gscale * grad
__global__ void k_adabelief(float* x, float* m, float* s, const float* grad, size_t n, float gscale, float eta, float b1,
                            float b2, float eps, float c1, float c2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = gscale * grad[i];
        const float mt = b1 * m[i] + (1.0f - b1) * d;
        const float st_ = b2 * s[i] + (1.0f - b2) * (d - mt) * (d - mt) + eps;
        m[i] = mt;
        s[i] = st_;
        x[i] -= eta * mt / c1 / (sqrtf(st_ / c2) + eps);
    }
}
void adabelief_step(hipStream_t st, float* x, float* m, float* s, const float* grad, size_t n, float gscale, float eta,
                    float b1, float b2, float eps, double b1p, double b2p) {
    // Flux keeps the running powers of beta in Float64 and forms 1 - beta^t there (Flux 0.14 src/optimise/optimisers.jl, AdaBelief:
    // `Float64[beta[1], beta[2]]`); in float, 1 - 0.999 is 0.00099998713 instead of 0.001
    const float c1 = (float)(1.0 - b1p), c2 = (float)(1.0 - b2p);
    hipLaunchKernelGGL(k_adabelief, dim3(nblocks(n)), dim3(256), 0, st, x, m, s, grad, n, gscale, eta, b1, b2, eps, c1, c2);
}

}  // namespace motifs
