// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports per byte actually read, by access width (MI355X_MICROARCH.md, HBM section: exactly 1/2 for
// 16 B per lane; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/ubench/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fc -- ./fetch_calib
// Every kernel streams the same 2 GiB once (each byte read once, coalesced over the wave); rows256 reads it as random 256-byte rows, one row per
// wave-instruction at 4 B per lane -- the shape of the scoring kernel's dense count rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <typename T>
__global__ void stream_read(const T* __restrict__ p, size_t n, T* sink, int magic) {
    T acc = T();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        unsigned char* a = reinterpret_cast<unsigned char*>(&acc);
        for (unsigned k = 0; k < sizeof(T); ++k) a[k] ^= b[k];
    }
    if (reinterpret_cast<unsigned char*>(&acc)[0] == (unsigned char)magic) sink[0] = acc;   // (never true: the buffer holds 0x01 bytes; the compiler cannot know)
}
__global__ void rows256(const uint32_t* __restrict__ p, const uint32_t* __restrict__ perm, size_t n_rows, uint32_t* sink, int magic) {
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6; r < n_rows; r += ((size_t)gridDim.x * blockDim.x) >> 6)
        acc ^= p[(size_t)perm[r] * 64 + lane];
    if (acc == (uint32_t)magic) sink[0] = acc;
}
int main() {
    const size_t bytes = 2048ull << 20;   // 8 x the Infinity Cache: a 256 MiB buffer fresh from hipMemset was served without the requests FETCH_SIZE tallies (8.5 KB reported)
    void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
    void* sink; hipMalloc(&sink, 64);
    const size_t n_rows = bytes / 256;
    std::vector<uint32_t> perm(n_rows);
    for (size_t i = 0; i < n_rows; ++i) perm[i] = (uint32_t)i;
    uint64_t s = 88172645463325252ull;
    for (size_t i = n_rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    uint32_t* dperm; hipMalloc(&dperm, n_rows * 4); hipMemcpy(dperm, perm.data(), n_rows * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        stream_read<uint32_t><<<4096, 256>>>((const uint32_t*)d, bytes / 4, (uint32_t*)sink, 0x5a);
        stream_read<uint2><<<4096, 256>>>((const uint2*)d, bytes / 8, (uint2*)sink, 0x5a);
        stream_read<uint4><<<4096, 256>>>((const uint4*)d, bytes / 16, (uint4*)sink, 0x5a);
        rows256<<<4096, 256>>>((const uint32_t*)d, dperm, n_rows, (uint32_t*)sink, 0x5a5a5a5a);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes\n", bytes);
    return 0;
}
