// devbuf.h -- growable device / pinned-host buffers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcd {

// growable device buffer: capacity at least doubles; the first `keep` bytes survive a growth (device-to-device copy on
// `s`, ordered after everything already enqueued there).  The old allocation is freed after the stream drained.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes, size_t keep, hipStream_t s, int64_t* total) {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = cap ? cap * 2 : 256;
        while (ncap < bytes) ncap *= 2;
        void* np = nullptr;
        hipError_t e = hipMalloc(&np, ncap);
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(np, p, keep < cap ? keep : cap, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        if (p) {
            e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
            (void)hipFree(p);
        }
        if (total) *total += (int64_t)ncap - (int64_t)cap;
        p = np;
        cap = ncap;
        return hipSuccess;
    }
    void release(int64_t* total) {
        if (p) (void)hipFree(p);
        if (total) *total -= (int64_t)cap;
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return (T*)p; }
};

struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = cap ? cap * 2 : 4096;
        while (ncap < bytes) ncap *= 2;
        void* np = nullptr;
        hipError_t e = hipHostMalloc(&np, ncap, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (p) (void)hipHostFree(p);
        p = np;
        cap = ncap;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return (T*)p; }
};

}  // namespace lcd
