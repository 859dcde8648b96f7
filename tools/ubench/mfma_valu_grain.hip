// Micro-benchmark: at which granularity do v_mfma_f32_32x32x16_bf16 and the filter's top-3 update (v_and_or + 2 v_med3 + v_min per
// score) overlap inside ONE wave?  One "pair" = 24 bf16 MFMAs on two interleaved accumulator chains + the top-3 update of the 32
// scores of the previous pair (128 VALU), as in bf_pair (knn_mfma_kernels.hip).  G = MFMAs issued back to back before the VALU
// that belongs to them (G = 1: the kernel's pattern; G = 24: all MFMAs, then all VALU).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int32_t med3(int32_t a, int32_t b, int32_t c) { int32_t r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#ifndef PUSH_MINMAX
#define PUSH_MINMAX 0
#endif
__device__ __forceinline__ void push(int32_t& k0, int32_t& k1, int32_t& k2, float s, uint32_t mask, uint32_t idx) {
    int32_t k = (int32_t)((__float_as_uint(s) & mask) | idx);
#if PUSH_MINMAX
    const int32_t t0 = min(k0, k); k = max(k0, k); k0 = t0;      // two-source instructions only
    const int32_t t1 = min(k1, k); k = max(k1, k); k1 = t1;
    k2 = min(k2, k);
    return;
#endif
    const int32_t n2 = med3(k1, k2, k), n1 = med3(k0, k1, k);
    k0 = min(k0, k); k1 = n1; k2 = n2;
}
// MODE bit 0: MFMAs, bit 1: VALU
template <int G, int MODE, int AUG>
__device__ __forceinline__ void pair(const bf16x8 (&a)[4], const bf16x8 (&b0)[4], const bf16x8 (&b1)[4], f32x16& c0, f32x16& c1, const f32x16& p0,
                                     const f32x16& p1, uint32_t mask, int32_t (&k)[6]) {
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c0 = z; c1 = z;
    if (AUG == 1 && (MODE & 1)) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, mask), p0[0], z, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, mask), p1[0], z, 0, 0, 0);
    }
    if (AUG == 2 && (MODE & 1)) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b1[0], z, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b0[0], z, 0, 0, 0);
    }
    int pushed = 0;
#pragma unroll
    for (int m = 0; m < 24; ++m) {
        if (MODE & 1) {
            if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 3], b1[(m >> 3) & 3], c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 3], b0[(m >> 3) & 3], c0, 0, 0, 0);
        } else if (m == 0) { c0 = p1; c1 = p0; }
        if ((m + 1) % G == 0) {
            const int upto = (m + 1) * 32 / 24;                    // scores due after m + 1 MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 6) == 4) {
#pragma unroll
                for (int s = pushed; s < upto; ++s) asm volatile("" :: "v"(p0[s >> 1]), "v"(p1[s >> 1]));
            }
            if (MODE & 2) {
#pragma unroll
                for (int s = pushed; s < upto; ++s) {
                    if (s & 1) push(k[3], k[4], k[5], p1[s >> 1], mask, (uint32_t)s);
                    else push(k[0], k[1], k[2], p0[s >> 1], mask, (uint32_t)s);
                }
            }
            pushed = upto;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int G, int MODE, int WAVES, int AUG>
__global__ __launch_bounds__(WAVES * 64) void kern(uint32_t* out, int iters, const uint4* __restrict__ src) {
    bf16x8 a[4], a2[4], b0[4], b1[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[threadIdx.x * 4 + i]);
        a2[i] = __builtin_bit_cast(bf16x8, src[12288 + threadIdx.x * 4 + i]);
        b0[i] = __builtin_bit_cast(bf16x8, src[4096 + threadIdx.x * 4 + i]);
        b1[i] = __builtin_bit_cast(bf16x8, src[8192 + threadIdx.x * 4 + i]);
    }
    uint32_t mask;
    asm("v_mov_b32 %0, 0xffffff80" : "=v"(mask));
    int32_t k[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    f32x16 x0, x1, p0, p1;
    for (int r = 0; r < 16; ++r) { p0[r] = (float)(threadIdx.x * 16 + r); p1[r] = (float)(threadIdx.x * 16 + r) * 0.5f; }
    for (int i = 0; i < iters; ++i) {
        pair<G, MODE, AUG>(a, b0, b1, x0, x1, p0, p1, mask, k);
        pair<G, MODE, AUG>(a2, b1, b0, p0, p1, x0, x1, mask, k);
    }
    uint32_t s = 0;
    for (int r = 0; r < 16; ++r) s += __float_as_uint(p0[r]) + __float_as_uint(p1[r]);
    for (int j = 0; j < 6; ++j) s += (uint32_t)k[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int G, int MODE, int WAVES, int AUG = 0>
void run(const char* label) {
    const int blocks = 256, iters = 2000;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4);
    uint4* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<G, MODE, WAVES, AUG><<<blocks, WAVES * 64>>>(out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<G, MODE, WAVES, AUG><<<blocks, WAVES * 64>>>(out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pairs_per_simd = (double)iters * 2 * (WAVES / 4.0);
    printf("%-28s G=%2d waves/SIMD=%d  %.3f ms  %6.0f cycles per pair and SIMD at 2.4 GHz  (24 bf16 MFMAs of 32 cycles: 768)\n", label, G, WAVES / 4, ms,
           ms * 1e-3 * 2.4e9 / pairs_per_simd);
    hipFree(out); hipFree(src);
}
int main() {
    run<24, 2, 4>("VALU only");
    run<24, 5, 4>("MFMA, results kept alive");
    run<1, 3, 4>("MFMA + VALU");
    run<6, 3, 4>("MFMA + VALU");
    run<24, 3, 4>("MFMA + VALU");
    run<24, 5, 4, 1>("MFMA + f32 aug, kept alive");
    run<1, 3, 4, 1>("MFMA + f32 aug + VALU");
    run<1, 3, 4, 2>("MFMA + bf16 aug + VALU");
    run<24, 5, 8>("MFMA, results kept alive");
    run<1, 3, 8>("MFMA + VALU");
    run<1, 3, 8, 1>("MFMA + f32 aug + VALU");
    return 0;
}
