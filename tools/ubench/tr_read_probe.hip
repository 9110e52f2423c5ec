#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* in, uint2* out) {
    __shared__ uint16_t lds[32 * 64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    auto* ptr = (__attribute__((address_space(3))) h16x4*)(lds + (8 * g + q) * 64 + 4 * p);
    h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(ptr);
    out[l] = __builtin_bit_cast(uint2, v);
}
int main() {
    uint16_t h[32 * 64]; for (int r = 0; r < 32; r++) for (int c = 0; c < 64; c++) h[r * 64 + c] = r * 100 + c;
    uint16_t* d; uint2* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 8); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    uint2 r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) { uint16_t* e = (uint16_t*)&r[l]; printf("lane %2d: %4d %4d %4d %4d\n", l, e[0], e[1], e[2], e[3]); }
}
