// launch_gap.hip -- how long does the GPU idle between dependent kernels of one stream?  Kernels spin for a fixed wall time
// (s_memrealtime, 100 MHz), so the host is always ahead; wall / iteration - sum of spins = idle time per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_a(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (sink == (int*)1) lds[threadIdx.x] = 1;
}
__global__ void spin_b(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (sink == (int*)1) sink[threadIdx.x] = 2;
}
static void host_spin(double us) {
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) { }
}
// one iteration = [spin_a, spin_b] enqueued back to back, then `pace_us` of host time before the next iteration
static double run(hipStream_t st, int iters, size_t lds_a, double spin_us, double pace_us, bool alternate) {
    hipDeviceSynchronize();
    const unsigned long long ticks = (unsigned long long)(spin_us * 100.0);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) {
        spin_a<<<512, 256, lds_a, st>>>(ticks, nullptr);
        if (alternate) spin_b<<<512, 256, 0, st>>>(ticks, nullptr); else spin_a<<<512, 256, lds_a, st>>>(ticks, nullptr);
        if (pace_us > 0) host_spin(pace_us);
    }
    hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters - 2.0 * spin_us;
}

int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    printf("idle GPU time per iteration of two dependent 15-us kernels (us); 512 workgroups x 256 threads\n");
    for (size_t lds : {(size_t)0, (size_t)146 * 1024}) for (int alt = 0; alt < 2; ++alt) for (double pace : {0.0, 5.0, 10.0, 20.0}) {
        run(st, 50, lds, 15.0, pace, alt);
        printf("  lds(a) %3zu KB  %s  host pause %4.1f us : %6.2f\n", lds / 1024, alt ? "a,b (two kernels)" : "a,a (one kernel) ", pace, run(st, 1000, lds, 15.0, pace, alt));
    }
    return 0;
}
