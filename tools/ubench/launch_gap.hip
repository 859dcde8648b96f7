// launch_gap.hip -- how long does the GPU idle between two dependent kernels of one stream, by the LDS request of the two kernels?
// (the frame loop alternates a 146 KB-LDS filter launch with small-LDS launches; rocprofv3 shows ~5.6 us in front of every filter launch
// and ~0 in front of the others)   hipcc --offload-arch=gfx950 -O3 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long cycles, int* sink) {
    extern __shared__ int lds[];
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && sink == (int*)1) lds[0] = 1;
}
template <int R> __global__ __launch_bounds__(256) void fat_kernel(long long cycles, float* out) {   // many VGPRs
    float v[R];
    for (int i = 0; i < R; ++i) v[i] = threadIdx.x * 0.5f + i;
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { for (int i = 0; i < R; ++i) v[i] = v[i] * 1.0001f + 0.5f; }
    float s = 0; for (int i = 0; i < R; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

static double run(hipStream_t st, int iters, size_t lds_a, size_t lds_b, int grid_a, int grid_b, long long cyc) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) {
        spin_kernel<<<grid_a, 256, lds_a, st>>>(cyc, nullptr);
        spin_kernel<<<grid_b, 256, lds_b, st>>>(cyc, nullptr);
    }
    hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    const long long cyc = 2000;   // 100 MHz clock64? on gfx9 s_memtime runs at shader clock: ~1 us at 2 GHz
    const size_t sizes[] = {0, 32 * 1024, 64 * 1024, 65 * 1024, 100 * 1024, 146 * 1024};
    printf("us per PAIR of dependent launches (each kernel spins ~%lld cycles), 256 workgroups each\n", cyc);
    for (size_t a : sizes) for (size_t b : sizes) {
        run(st, 50, a, b, 256, 256, cyc);
        printf("  lds %6zu KB -> %6zu KB : %7.2f\n", a / 1024, b / 1024, run(st, 2000, a, b, 256, 256, cyc));
    }
    printf("grid sizes (0 KB LDS): 256/256 %.2f  547/891 %.2f  2048/2048 %.2f\n", run(st, 2000, 0, 0, 256, 256, cyc), run(st, 2000, 0, 0, 547, 891, cyc),
           run(st, 2000, 0, 0, 2048, 2048, cyc));
    printf("grid 547 @146 KB / 891 @0: %.2f\n", run(st, 2000, 146 * 1024, 0, 547, 891, cyc));
    return 0;
}
