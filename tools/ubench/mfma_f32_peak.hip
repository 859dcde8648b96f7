// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 on this box (1, 2 or 4 independent accumulator chains per wave,
// 1 or 2 waves per SIMD).  Used to calibrate the roofline of knn_mfma_filter_kernel.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same, but every MFMA reads a DIFFERENT pair of operand registers (as a GEMM k-loop does)
template <int CHAINS>
__global__ __launch_bounds__(256) void kv(float* out, int iters, const float* __restrict__ src) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c);
    float a[32], b[CHAINS][32];
    for (int k = 0; k < 32; ++k) { a[k] = src[threadIdx.x * 32 + k]; for (int c = 0; c < CHAINS; ++c) b[c][k] = src[4096 + (threadIdx.x + c) * 32 + k]; }
    for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[c][kk], acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void runv(int blocks, const char* label) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    float* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kv<CHAINS><<<blocks, 256>>>(out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kv<CHAINS><<<blocks, 256>>>(out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * (iters / 4) * 32 * CHAINS;
    const double flops = mfmas * 32 * 32 * 2 * 2;
    printf("%-40s blocks=%4d chains=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", label, blocks, CHAINS, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
}

template <int CHAINS>
void run(int blocks, const char* label) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 8 * CHAINS;
    const double flops = mfmas * 32 * 32 * 2 * 2;
    printf("%-40s blocks=%4d chains=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", label, blocks, CHAINS, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
    hipFree(out);
}
int main() {
    run<1>(256, "1 wave/SIMD, 1 dependent chain");
    run<2>(256, "1 wave/SIMD, 2 chains");
    run<4>(256, "1 wave/SIMD, 4 chains");
    run<1>(512, "2 waves/SIMD, 1 chain each");
    run<2>(512, "2 waves/SIMD, 2 chains each");
    run<1>(1024, "4 waves/SIMD, 1 chain each");
    runv<1>(256, "varying operands, 1 wave/SIMD, 1 group");
    runv<4>(256, "varying operands, 1 wave/SIMD, 4 groups");
    runv<2>(512, "varying operands, 2 waves/SIMD, 2 groups");
    return 0;
}
