// anyorder.hip -- can two kernels of ONE stream overlap?  hipExtLaunchKernelGGL takes a flags word whose only documented value is
// hipExtAnyOrderLaunch ("the kernel can be launched in any order": the dispatch packet's barrier bit is cleared, so the command processor
// does not wait for the packets in front of it).  A pipelined frame is launch A then launch B; if launch A(t+1) could start while launch
// B(t) still runs -- the packet in front of it having waited for everything older -- the two latency-bound launches would overlap without
// a second stream (every cross-stream hand-off costs ~7 us here, event_cost.hip).  This measures it with kernels that spin for a fixed
// time and record start / end time stamps (s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o anyorder anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(long long ticks, unsigned long long* stamps, int slot) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) stamps[2 * slot] = (unsigned long long)t0;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) stamps[2 * slot + 1] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int iters = 500;
    const long long A = 1500, B = 1300;                    // 15 us and 13 us of spinning (ticks of 10 ns)
    unsigned long long* d_st = nullptr;
    hipMalloc((void**)&d_st, sizeof(unsigned long long) * 4 * iters);
    std::vector<unsigned long long> st(4 * iters);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: A, B, A, B ... all in order.  mode 1: every A launched with hipExtAnyOrderLaunch (B keeps its barrier).
        // mode 2: every launch with the flag.
        hipMemset(d_st, 0, sizeof(unsigned long long) * 4 * iters);
        for (int warm = 0; warm < 20; ++warm) spin_kernel<<<64, 256, 0, s>>>(100, nullptr, 0);
        hipStreamSynchronize(s);
        const double t0 = now_us();
        for (int i = 0; i < iters; ++i) {
            const unsigned fa = (mode >= 1) ? hipExtAnyOrderLaunch : 0u, fb = (mode == 2) ? hipExtAnyOrderLaunch : 0u;
            hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, nullptr, nullptr, fa, A, d_st, 2 * i);
            hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, nullptr, nullptr, fb, B, d_st, 2 * i + 1);
        }
        const double h = now_us() - t0;
        const hipError_t e = hipStreamSynchronize(s);
        const double w = now_us() - t0;
        hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * 4 * iters, hipMemcpyDeviceToHost);
        // overlap: how often did A(i+1) start before B(i) ended?
        int overlapped = 0;
        double gap_sum = 0;
        for (int i = 10; i + 1 < iters; ++i) {
            const long long b_end = (long long)st[2 * (2 * i + 1) + 1], a_next = (long long)st[2 * (2 * i + 2)];
            if (a_next < b_end) overlapped += 1;
            gap_sum += (double)(a_next - b_end) * 0.01;
        }
        printf("mode %d (%s): %s, host %.2f us, wall %.2f us per A+B pair (28 us of spinning); A(i+1) started before B(i) ended in %d of %d pairs, mean start(A[i+1]) - end(B[i]) = %.2f us\n",
               mode, mode == 0 ? "in order" : mode == 1 ? "A any-order" : "A and B any-order", hipGetErrorString(e), h / iters, w / iters, overlapped, iters - 11,
               gap_sum / (iters - 11));
    }
    return 0;
}
