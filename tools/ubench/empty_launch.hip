// Micro-benchmark: what an (almost) empty dependent kernel costs in a stream: [work kernel][flag-check kernel] x N vs [work kernel] x N.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void work(float* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = x[i];
    for (int k = 0; k < n; ++k) v = v * 1.0001f + 0.5f;
    x[i] = v;
}
__global__ __launch_bounds__(256) void check(const int* flag, float* out) {
    if (flag[0] <= 0) return;
    out[blockIdx.x * 256 + threadIdx.x] = 1.0f;
}
int main() {
    float* x; hipMalloc(&x, 1 << 24); hipMemset(x, 0, 1 << 24);
    int* flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int grid : {0, 1, 16, 192, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < N; ++i) {
                work<<<1024, 256>>>(x, 2000);
                if (grid) check<<<grid, 256>>>(flag, x);
            }
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("check grid %4d: %.2f us per iteration\n", grid, ms * 1e3 / N);
    }
    return 0;
}
