// What a grid-stride 16-byte copy into an arena of each allocation kind costs on one MI355X (one process; with an argument N > 0 the
// program forks N - 1 siblings that run the same loop beside it: two processes sharing the GPU, as tools/p2p_bench.py does).
// hipcc --offload-arch=gfx950 -O3 -o uncached_copy uncached_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <sys/wait.h>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* dst, size_t groups) {
    for (size_t g = (size_t)blockIdx.x * 1024 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 1024) {
        uint4 a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (g + k * 256 < groups) a[k] = src[g + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (g + k * 256 < groups) dst[g + k * 256] = a[k];
    }
}
int main(int argc, char** argv) {
    const int procs = argc > 1 ? atoi(argv[1]) : 1;
    int me = 0;
    for (int i = 1; i < procs; ++i) if (fork() == 0) { me = i; break; }
    const size_t bytes = 8u << 20, groups = bytes / 16;
    uint4 *src, *dst[3];
    hipMalloc((void**)&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc((void**)&dst[0], bytes);
    hipExtMallocWithFlags((void**)&dst[1], bytes, hipDeviceMallocFinegrained);
    hipExtMallocWithFlags((void**)&dst[2], bytes, hipDeviceMallocUncached);
    const char* names[3] = {"coarse", "finegrained", "uncached"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind)
        for (int dir = 0; dir < 2; ++dir)
            for (unsigned grid : {64u, 256u, 1024u}) {
                const uint4* s = dir ? dst[kind] : src;
                uint4* d = dir ? src : dst[kind];
                for (int i = 0; i < 10; ++i) copy_kernel<<<grid, 256>>>(s, d, groups);
                hipEventRecord(e0);
                for (int i = 0; i < 200; ++i) copy_kernel<<<grid, 256>>>(s, d, groups);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                printf("proc %d of %d  %-11s %s  grid %4u  %7.2f us per 8 MB copy  %6.2f TB/s (read + write)\n", me, procs, names[kind], dir ? "arena -> buffer" : "buffer -> arena",
                       grid, ms * 1000 / 200, 2.0 * bytes / (ms / 200 * 1e-3) / 1e12);
            }
    if (me == 0) for (int i = 1; i < procs; ++i) wait(nullptr);
    return 0;
}
