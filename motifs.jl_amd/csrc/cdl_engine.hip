// cdl_engine.hip — primitives of the sparse-coding engine (see cdl_engine.h).
// Each forward primitive is a HIP kernel; its VJP is recorded on the tape.
#include "cdl_engine.h"

#include <algorithm>
#include <cstdio>

namespace motifs {

static inline unsigned nblocks(size_t n, int per = 256, size_t cap = 256 * 32) {
    size_t b = (n + per - 1) / per;
    if (b < 1) b = 1;
    return (unsigned)std::min(b, cap);
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

float* Engine::grad(Tensor t) {
    if (!t->g) {
        t->g = arena.alloc(t->n);
        if (!t->g) {
            failed = true;
            return nullptr;
        }
        (void)hipMemsetAsync(t->g, 0, t->n * 4, st);
    }
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
__global__ void k_axpy(const float* go, float a, size_t n, float* dx) {   // dx += a * go
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] += a * go[i];
}
__global__ void k_mul(const float* x, const float* y, size_t n, size_t yn, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = x[i] * y[i % yn];
}
__global__ void k_mul_bwd_x(const float* go, const float* y, size_t n, size_t yn, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] += go[i] * y[i % yn];
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
        if (x) hipLaunchKernelGGL(k_mul_bwd_x, dim3(nblocks(n)), dim3(256), 0, st, go, x, n, n, dy);  // coef == 1 there
        else hipLaunchKernelGGL(k_axpy, dim3(nblocks(n)), dim3(256), 0, st, go, coef, n, dy);
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
__global__ void k_relu_bwd(const float* go, const float* x, size_t n, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] += x[i] > 0.0f ? go[i] : 0.0f;
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
__global__ void k_norml2(const float* x, int seg, float* out, float* nrm_out) {
    const float* xs = x + (size_t)blockIdx.x * seg;
    double acc = 0;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) acc += (double)xs[i] * xs[i];
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
    for (int i = threadIdx.x; i < seg; i += blockDim.x) out[(size_t)blockIdx.x * seg + i] = xs[i] / nrm;
}
__global__ void k_norml2_bwd(const float* go, const float* out, const float* nrm_in, int seg, float* dx) {
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
    for (int i = threadIdx.x; i < seg; i += blockDim.x) dx[base + i] += (go[base + i] - out[base + i] * dot) / nrm;
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
__global__ void k_sumsq_groups_bwd(const float* gout, const float* x, size_t per_group, size_t n, float coef2, float* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] += coef2 * x[i] * gout[i / per_group];
}

#define EW(kern, n, ...) hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(256), 0, st, __VA_ARGS__)

Tensor Engine::lin(Tensor x, float a, Tensor y, float b, float cst) {
    Tensor out = make(x->n, x->needs_grad || (y && y->needs_grad));
    if (failed) return out;
    EW(k_lin, x->n, x->v, a, y ? y->v : nullptr, b, cst, x->n, y ? y->n : (size_t)1, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y, a, b]() {
            if (!out->g) return;
            if (x->needs_grad) EW(k_axpy, out->n, out->g, a, out->n, grad(x));
            if (y && y->needs_grad) bcast_reduce(st, out->g, nullptr, out->n, y->n, b, grad(y));
        });
    return out;
}

Tensor Engine::mul(Tensor x, Tensor y) {
    Tensor out = make(x->n, x->needs_grad || y->needs_grad);
    if (failed) return out;
    EW(k_mul, x->n, x->v, y->v, x->n, y->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, y]() {
            if (!out->g) return;
            if (x->needs_grad) EW(k_mul_bwd_x, out->n, out->g, y->v, out->n, y->n, grad(x));
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
            if (out->g) EW(k_relu_bwd, out->n, out->g, x->v, out->n, grad(x));
        });
    return out;
}

Tensor Engine::maskmul(Tensor x, const float* mask, float c) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_maskmul, x->n, x->v, mask, c, x->n, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, mask, c]() {
            if (out->g) EW(k_maskmul, out->n, out->g, mask, c, out->n, grad(x), 1);
        });
    return out;
}

Tensor Engine::expo(Tensor x) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_exp, x->n, x->v, x->n, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x]() {
            if (out->g) EW(k_mul_bwd_x, out->n, out->g, out->v, out->n, out->n, grad(x));
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

Tensor Engine::norml2(Tensor x, int seg) {
    Tensor out = make(x->n, x->needs_grad);
    Tensor nrm = make(x->n / seg, false);
    if (failed) return out;
    const unsigned nseg = (unsigned)(x->n / seg);
    hipLaunchKernelGGL(k_norml2, dim3(nseg), dim3(256), 0, st, x->v, seg, out->v, nrm->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, nrm, seg, nseg]() {
            if (out->g) hipLaunchKernelGGL(k_norml2_bwd, dim3(nseg), dim3(256), 0, st, out->g, out->v, nrm->v, seg, grad(x));
        });
    return out;
}

Tensor Engine::sumsq_groups(Tensor x, float coef, int groups) {
    Tensor out = make(groups, x->needs_grad);
    if (failed) return out;
    const size_t per = x->n / groups;
    (void)hipMemsetAsync(out->v, 0, (size_t)groups * 4, st);
    hipLaunchKernelGGL(k_sumsq_groups, dim3(nblocks(per, 256, 64), groups), dim3(256), 0, st, x->v, per, coef, out->v);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, per, coef]() {
            if (out->g) EW(k_sumsq_groups_bwd, x->n, out->g, x->v, per, x->n, 2.0f * coef, grad(x));
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
            float v = 0.0f;
            if (p < gm.P && q < gm.Q) {
                const int e = gm.a0 + p * gm.sa + q;
                if (e >= 0 && e < gm.amax) v = As_g[e];
            }
            As[qq][row] = v;
        }
#pragma unroll
        for (int it = 0; it < (BK * BN + 255) / 256; it++) {
            const int idx = tid + it * 256;
            if (idx < BK * BN) {
                const int qq = idx / BN, nn = idx % BN;
                const int q = q0 + qq, n = n0 + nn;
                Bs[qq][nn] = (q < gm.Q && n < gm.N) ? Bg[(size_t)q * gm.N + n] : 0.0f;
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
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int p = p0 + ty * TM + i;
        if (p >= gm.P) continue;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + tx * TN + j;
            if (n < gm.N) {
                float* o = &Cs[(size_t)p * gm.N + n];
                *o = acc ? *o + accv[i][j] : accv[i][j];
            }
        }
    }
}

static void launch_toep(hipStream_t st, const float* A, const float* Bm, float* C, const ToepGeom& gm, int acc) {
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
            float v = 0.0f;
            if (k < KT && q < gm.Q) {
                const int sl = k / gm.P, p = k - sl * gm.P;
                const int e = gm.a0 + p * gm.sa + q;
                if (e >= 0 && e < gm.amax) v = A[(size_t)(g * gm.B + sl) * gm.lda + e];
            }
            As[kk][qq] = v;
        }
#pragma unroll
        for (int it = 0; it < (BK * BN + 255) / 256; it++) {
            const int idx = tid + it * 256;
            if (idx < BK * BN) {
                const int kk = idx / BN, nn = idx % BN;
                const int k = k0 + kk, n = n0 + nn;
                float v = 0.0f;
                if (k < KT && n < gm.N) {
                    const int sl = k / gm.P, p = k - sl * gm.P;
                    v = C[(size_t)(g * gm.B + sl) * gm.ldc + (size_t)p * gm.N + n];
                }
                Cs[kk][nn] = v;
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

static void launch_wgrad(hipStream_t st, const float* A, const float* C, float* dB, const ToepGeom& gm, int acc) {
    const int G = gm.S / gm.B;
    if (gm.N <= 32) {
        dim3 grid((gm.N + 31) / 32, (gm.Q + 63) / 64, G);
        hipLaunchKernelGGL(k_wgrad<32>, grid, dim3(256), 0, st, A, C, dB, gm, acc);
    } else {
        dim3 grid((gm.N + 63) / 64, (gm.Q + 63) / 64, G);
        hipLaunchKernelGGL(k_wgrad<64>, grid, dim3(256), 0, st, A, C, dB, gm, acc);
    }
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
    for (int g = 0; g < G; g++) acc += x[(size_t)g * per + j];
    y[j] += (float)acc;
}

// per group [H][W][N] -> out[i'][n][j] = in[H-1-i'][j][n]
__global__ void k_flipT(const float* x, int g, int H, int W, int N, float* out, int acc) {
    const size_t per = (size_t)H * W * N, total = per * g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t gg = i / per, r = i % per;      // r indexes the OUTPUT [H][N][W]
        const int j = (int)(r % W), n = (int)((r / W) % N), ip = (int)(r / ((size_t)W * N));
        const float v = x[gg * per + ((size_t)(H - 1 - ip) * W + j) * N + n];
        out[i] = acc ? out[i] + v : v;
    }
}

// dA[s][e] += sum_{p,q: a0 + p*sa + q = e} sum_n dC[s][p][n] * Bm[g][q][n]  (adjoint of the Toeplitz
// gather).  With A viewed as rows of W = sa columns and the window as H = Q/W rows, this is again a
// Toeplitz GEMM: over dC (rows of N columns) with the filter flipT(Bm) = [H][N][W], rows reversed.
static bool toep_adjoint_a(Engine& e, const float* dC, const float* Bm, float* dA, const ToepGeom& gm) {
    const int W = gm.sa, H = gm.Q / gm.sa;
    const int gB = gm.ldb == 0 ? 1 : gm.S / gm.B;
    const size_t per = (size_t)gm.Q * gm.N;
    float* tmp = e.arena.alloc(per * gB);
    if (!tmp) {
        e.failed = true;
        return false;
    }
    hipLaunchKernelGGL(k_flipT, dim3(nblocks(per * gB)), dim3(256), 0, e.st, Bm, gB, H, W, gm.N, tmp, 0);
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
    launch_toep(e.st, dC, tmp, dA, g2, 1);
    return true;
}

Tensor Engine::toep(Tensor A, Tensor Bm, const ToepGeom& gm) {
    Tensor out = make((size_t)gm.S * gm.ldc, A->needs_grad || Bm->needs_grad);
    if (failed) return out;
    launch_toep(st, A->v, Bm->v, out->v, gm, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, A, Bm, gm]() {
            if (!out->g) return;
            if (A->needs_grad) {
                float* dA = grad(A);
                if (dA) toep_adjoint_a(*this, out->g, Bm->v, dA, gm);
            }
            if (Bm->needs_grad) {
                float* dB = grad(Bm);
                if (!dB) return;
                const int G = gm.S / gm.B;
                if (gm.ldb != 0) {
                    launch_wgrad(st, A->v, out->g, dB, gm, 1);
                } else {   // shared filter: per-group partials, then a sum over groups
                    const size_t per = (size_t)gm.Q * gm.N;
                    float* tmp = arena.alloc(per * G);
                    if (!tmp) {
                        failed = true;
                        return;
                    }
                    launch_wgrad(st, A->v, out->g, tmp, gm, 0);
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
    launch_wgrad(st, A->v, C->v, out->v, gm, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, A, C, gm]() {
            if (!out->g) return;
            ToepGeom g2 = gm;
            g2.ldb = (int64_t)gm.Q * gm.N;   // the "filter" of the adjoints is dOut, one slice per group
            if (A->needs_grad) {
                float* dA = grad(A);
                if (dA) toep_adjoint_a(*this, C->v, out->g, dA, g2);
            }
            if (C->needs_grad) {
                float* dCc = grad(C);
                if (dCc) launch_toep(st, A->v, out->g, dCc, g2, 1);
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
            if (out->g) EW(k_collapseD, D->n, out->g, g, M, fl, grad(D), 1);
        });
    return out;
}
Tensor Engine::collapseD(Tensor GA, int g, int M, int fl) {
    Tensor out = make((size_t)g * M * fl * 4, GA->needs_grad);
    if (failed) return out;
    EW(k_collapseD, out->n, GA->v, g, M, fl, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, GA, g, M, fl]() {
            if (out->g) EW(k_expandD, GA->n, out->g, g, M, fl, grad(GA), 1);
        });
    return out;
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
Tensor Engine::swap02(Tensor x, int g, int d0, int d1, int d2) {
    Tensor out = make(x->n, x->needs_grad);
    if (failed) return out;
    EW(k_swap02, x->n, x->v, g, d0, d1, d2, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, x, g, d0, d1, d2]() {
            if (out->g) EW(k_swap02, x->n, out->g, g, d2, d1, d0, grad(x), 1);
        });
    return out;
}

Tensor Engine::flipT(Tensor Bm, int g, int H, int W, int N) {
    Tensor out = make(Bm->n, Bm->needs_grad);
    if (failed) return out;
    EW(k_flipT, Bm->n, Bm->v, g, H, W, N, out->v, 0);
    if (recording && out->needs_grad)
        tape.push_back([this, out, Bm, g, H, W, N]() {
            // adjoint: dIn[i][j][n] += dOut[H-1-i][n][j]  == flipT with the roles of W and N exchanged
            if (out->g) EW(k_flipT, Bm->n, out->g, g, H, N, W, grad(Bm), 1);
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
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = x[i];
            if (!pred(v)) continue;
            const uint32_t key = fkey(v);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            int b = 0;
            for (; b < 256; b++) {
                if (run + hist[b] > k) break;
                run += hist[b];
            }
            sh[0] = (uint32_t)b;
            sh[1] = run;
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
__global__ __launch_bounds__(256) void k_topq_mask(const float* X, float* bitmat, int n, int q) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[2];
    const float* xs = X + (size_t)blockIdx.x * n;
    const uint32_t key = block_radix_select(xs, n, (uint32_t)(n - q), [](float) { return true; }, hist, sh);
    const float thr = unkey(key);
    for (int i = threadIdx.x; i < n; i += blockDim.x) bitmat[(size_t)blockIdx.x * n + i] = xs[i] >= thr ? 1.0f : 0.0f;
}
void topq_mask(hipStream_t st, const float* X, float* bitmat, int S, int n_per_seq, int q) {
    hipLaunchKernelGGL(k_topq_mask, dim3(S), dim3(256), 0, st, X, bitmat, n_per_seq, q);
}

// create_ZY_mask (model.jl:194-204): median of the strictly positive entries of the whole mini-batch
// (mean of the two middle values for an even count: Statistics.middle(a, b) = a/2 + b/2); mask = ZY >= median.
// No positive entry -> the reference skips the mask (:209); that is mask == 1 here.
__global__ __launch_bounds__(1024) void k_median_mask(const float* ZY, float* mask, int n) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[2];
    __shared__ uint32_t cnt_sh;
    const float* xs = ZY + (size_t)blockIdx.x * n;
    if (threadIdx.x == 0) cnt_sh = 0;
    __syncthreads();
    uint32_t c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c += xs[i] > 0.0f;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_sh, c);
    __syncthreads();
    const uint32_t cnt = cnt_sh;
    float med = -INFINITY;
    if (cnt > 0) {
        auto pos = [](float v) { return v > 0.0f; };
        const float lo = unkey(block_radix_select(xs, n, (cnt - 1) / 2, pos, hist, sh));
        med = lo;
        if ((cnt & 1u) == 0) {
            const float hi = unkey(block_radix_select(xs, n, cnt / 2, pos, hist, sh));
            med = lo / 2 + hi / 2;
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) mask[(size_t)blockIdx.x * n + i] = xs[i] >= med ? 1.0f : 0.0f;
}
void median_mask(hipStream_t st, const float* ZY, float* mask, int G, int n_per_group) {
    hipLaunchKernelGGL(k_median_mask, dim3(G), dim3(1024), 0, st, ZY, mask, n_per_group);
}

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
                            float b2, float eps, float b1p, float b2p) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = gscale * grad[i];
        const float mt = b1 * m[i] + (1.0f - b1) * d;
        const float st_ = b2 * s[i] + (1.0f - b2) * (d - mt) * (d - mt) + eps;
        m[i] = mt;
        s[i] = st_;
        x[i] -= eta * mt / (1.0f - b1p) / (sqrtf(st_ / (1.0f - b2p)) + eps);
    }
}
void adabelief_step(hipStream_t st, float* x, float* m, float* s, const float* grad, size_t n, float gscale, float eta,
                    float b1, float b2, float eps, float b1p, float b2p) {
    hipLaunchKernelGGL(k_adabelief, dim3(nblocks(n)), dim3(256), 0, st, x, m, s, grad, n, gscale, eta, b1, b2, eps, b1p, b2p);
}

}  // namespace motifs
