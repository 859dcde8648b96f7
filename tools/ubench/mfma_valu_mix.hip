// Micro-benchmark: how v_mfma_f32_32x32x2_f32 and ordinary VALU work (u32 min/max top-3 insertion, as in the epilogue of
// knn_mfma_filter_kernel) share a SIMD.  Modes: MFMA only, VALU only, both interleaved (4 MFMA : NV VALU), with 1 or 2 waves
// per SIMD.  Prints cycles per "unit" (4 MFMA + NV VALU) per SIMD.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void top3(uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t k) {   // 5 VALU
    const uint32_t t0 = min(b0, k); k = max(b0, k); b0 = t0;
    const uint32_t t1 = min(b1, k); k = max(b1, k); b1 = t1;
    b2 = min(b2, k);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// same with v_mfma_f32_32x32x16_bf16 (8 passes): 8 MFMA per unit = 256 cycles alone
template <int MODE, int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kb(uint32_t* out, int iters, const uint4* __restrict__ src, uint32_t seed) {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(bf16x8, src[threadIdx.x * 4 + i]); b[i] = __builtin_bit_cast(bf16x8, src[4096 + threadIdx.x * 4 + i]); }
    uint32_t b0[4], b1[4], b2[4], key[4];
    for (int j = 0; j < 4; ++j) { b0[j] = b1[j] = b2[j] = ~0u; key[j] = seed * (threadIdx.x + 1 + j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + m) & 3], b[m], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[(u + m) & 3], acc[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) {
#pragma unroll
                for (int v = 0; v < NV / 5; ++v) {
                    const int j = v & 3;
                    key[j] += 0x9E3779B9u;
                    top3(b0[j], b1[j], b2[j], key[j] ^ (key[j] >> 3));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += __float_as_uint(acc[c][r]);
    for (int j = 0; j < 4; ++j) s += b0[j] + b1[j] + b2[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE bit0: MFMA, bit1: VALU.  NV = VALU ops per 4 MFMAs (multiple of 5).  PIN = pin the interleaving with sched_barrier
template <int MODE, int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(uint32_t* out, int iters, const float* __restrict__ src, uint32_t seed) {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[threadIdx.x * 8 + i]; b[i] = src[8192 + threadIdx.x * 8 + i]; }
    uint32_t b0[4], b1[4], b2[4], key[4];
    for (int j = 0; j < 4; ++j) { b0[j] = b1[j] = b2[j] = ~0u; key[j] = seed * (threadIdx.x + 1 + j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 1) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 2) & 7], b[u], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) {
#pragma unroll
                for (int v = 0; v < NV / 5; ++v) {
                    const int j = v & 3;
                    key[j] += 0x9E3779B9u;                              // +2 full-rate VALU per insertion
                    top3(b0[j], b1[j], b2[j], key[j] ^ (key[j] >> 3));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += __float_as_uint(acc[c][r]);
    for (int j = 0; j < 4; ++j) s += b0[j] + b1[j] + b2[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, int WAVES>
void runb(const char* label) {
    const int blocks = 256;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4);
    uint4* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kb<MODE, NV, WAVES><<<blocks, WAVES * 64>>>(out, 16, src, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kb<MODE, NV, WAVES><<<blocks, WAVES * 64>>>(out, iters, src, 3u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double units_per_simd = (double)iters * 8 * (WAVES / 4.0);
    printf("bf16 %-29s waves/SIMD=%d  %.3f ms  %7.1f ns/unit/SIMD  (= %.0f cycles at 2.4 GHz; 8 bf16 MFMA alone would be 256)\n", label, WAVES / 4, ms,
           ms * 1e6 / units_per_simd, ms * 1e-3 * 2.4e9 / units_per_simd);
    hipFree(out); hipFree(src);
}

template <int MODE, int NV, int WAVES>
void run(const char* label) {
    const int blocks = 256;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * WAVES * 64 * 4);
    float* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV, WAVES><<<blocks, WAVES * 64>>>(out, 16, src, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NV, WAVES><<<blocks, WAVES * 64>>>(out, iters, src, 3u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double units_per_simd = (double)iters * 8 * (WAVES / 4.0);
    printf("%-34s waves/SIMD=%d  %.3f ms  %7.1f ns/unit/SIMD  (= %.0f cycles at 2.4 GHz; MFMA alone would be 256)\n", label, WAVES / 4, ms,
           ms * 1e6 / units_per_simd, ms * 1e-3 * 2.4e9 / units_per_simd);
    hipFree(out); hipFree(src);
}

int main() {
    run<1, 20, 4>("MFMA only (4/unit)");
    run<2, 20, 4>("VALU only (28/unit)");
    run<3, 20, 4>("MFMA + 28 VALU");
    run<2, 40, 4>("VALU only (56/unit)");
    run<3, 40, 4>("MFMA + 56 VALU");
    run<1, 20, 8>("MFMA only (4/unit)");
    run<2, 20, 8>("VALU only (28/unit)");
    run<3, 20, 8>("MFMA + 28 VALU");
    run<2, 40, 8>("VALU only (56/unit)");
    run<3, 40, 8>("MFMA + 56 VALU");
    runb<1, 20, 4>("MFMA only (8/unit)");
    runb<2, 20, 4>("VALU only (28/unit)");
    runb<3, 20, 4>("MFMA + 28 VALU");
    runb<3, 40, 4>("MFMA + 56 VALU");
    runb<1, 20, 8>("MFMA only (8/unit)");
    runb<3, 20, 8>("MFMA + 28 VALU");
    runb<3, 40, 8>("MFMA + 56 VALU");
    return 0;
}
