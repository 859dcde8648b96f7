// event_cost.hip -- what does it cost to hand a dependency from one stream to another and back, per frame?
// The pipelined frame runs two fused launches on one stream; its tail would like its own kernel (own register budget, 1024
// threads) on a second stream:   S1: A(t) .. wait(T[t-1]) .. B(t) .. record(Bev[t])      S2: wait(Bev[t-1]) .. tail(t-1) .. record(T[t-1])
// This measures the stream time and the host time of exactly that pattern against the single-stream pattern, with kernels that
// spin for a fixed time.   hipcc --offload-arch=gfx950 -O3 -o event_cost event_cost.hip && ./event_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long ticks) {           // s_memrealtime: 100 MHz
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int iters = 2000;
    const long long A = 2000, B = 1650, T = 1350;          // 20 us filter, 16.5 us launch B, 13.5 us tail (ticks of 10 ns)
    for (int flags_i = 0; flags_i < 2; ++flags_i) {
        const unsigned flags = flags_i ? hipEventDisableTiming : hipEventDefault;
        hipEvent_t evB[2], evT[2];
        for (int i = 0; i < 2; ++i) { hipEventCreateWithFlags(&evB[i], flags); hipEventCreateWithFlags(&evT[i], flags); }
        // ---- single stream: A(filter+tail fused: max(A, T) -> here A + small), B
        hipDeviceSynchronize();
        double t0 = now_us();
        for (int i = 0; i < iters; ++i) { spin_kernel<<<256, 256, 0, s1>>>(2650); spin_kernel<<<256, 256, 0, s1>>>(B); }
        double h1 = now_us() - t0;
        hipStreamSynchronize(s1);
        double d1 = now_us() - t0;
        // ---- two streams with events
        hipDeviceSynchronize();
        hipEventRecord(evB[1], s1);
        t0 = now_us();
        for (int i = 0; i < iters; ++i) {
            const int c = i & 1, p = c ^ 1;
            hipStreamWaitEvent(s2, evB[p], 0);
            spin_kernel<<<1, 1024, 0, s2>>>(T);
            hipEventRecord(evT[p], s2);
            spin_kernel<<<256, 256, 0, s1>>>(A);
            hipStreamWaitEvent(s1, evT[p], 0);
            spin_kernel<<<256, 256, 0, s1>>>(B);
            hipEventRecord(evB[c], s1);
        }
        double h2 = now_us() - t0;
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        double d2 = now_us() - t0;
        printf("%s events:  one stream (26.5 + 16.5 us of kernels): host %.1f us, wall %.1f us per frame | two streams (20 | 13.5, then 16.5): host %.1f us, wall %.1f us per frame\n",
               flags_i ? "disable-timing" : "default", h1 / iters, d1 / iters, h2 / iters, d2 / iters);
    }
    return 0;
}
