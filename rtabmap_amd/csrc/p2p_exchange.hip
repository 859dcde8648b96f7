// p2p_exchange.hip -- liblcd_p2p.so (include/lcd_p2p.h): the sharded frame's two exchanges as one-shot direct peer-to-peer kernels over
// arenas the ranks map into each other with hipIpc.  SURVEY.md section 5 / 8e: the messages (16 KB of candidate records, 0.4-8 MB of partial
// likelihood) are latency-bound on a ring; with every peer one xGMI hop away each rank writes all of them at once.  gfx950 only.
//
// Memory model used here: arenas are UNCACHED device memory (hipDeviceMallocUncached, MTYPE_UC: every store is written through and no
// cache of the reading device keeps a line), so publishing data needs no cache maintenance at all -- a writer waits until its stores are
// acknowledged (s_waitcnt vmcnt(0): gfx950 counts stores in vmcnt) and then stores the flag, relaxed, at system scope, into the READER's
// arena; the reader polls with relaxed system-scope loads and makes one acquire fence when the flag is there.  (The compiler's release
// fences, at agent and at system scope alike, write the whole L2 back: with the 8 MB of partial sums a scoring kernel has just left dirty
// there, staging took 55 us instead of 13 -- tools/ubench/uncached_copy.hip, profiles/r06_p2p_exchange.txt.)  Flags hold the number of
// the exchange (monotonic 64-bit epochs, one counter per kind of exchange), so nothing is ever reset between exchanges and a late reader
// cannot mistake an old announcement for a new one.
#include "../../include/lcd_p2p.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unistd.h>

namespace {

constexpr size_t kFlagStride = 64;                                  // one flag per 64-byte line: each line has exactly one writer
constexpr size_t kGatherFlagOff = 0;                                // [peer] the newest all-gather whose block from `peer` lies in my mailbox
constexpr size_t kStageFlagOff = kFlagStride * LCD_P2P_MAX_WORLD;   // [peer] the newest all-reduce whose operand `peer` has staged (A)
constexpr size_t kReduceFlagOff = 2 * kStageFlagOff;                // [peer] the newest all-reduce whose slice `peer` has written back (B)
constexpr size_t kMailboxOff = 4096;
constexpr uint32_t kMagic = 0x4c503201u;                            // "LP2" + ABI 1

struct Peers { unsigned char* base[LCD_P2P_MAX_WORLD]; int conservative; };   // conservative: lcd_p2p_set_conservative_fences

__device__ __forceinline__ void raise_flag(unsigned char* arena, size_t off, int slot, uint64_t epoch, int conservative) {
    uint64_t* f = (uint64_t*)(arena + off + (size_t)slot * kFlagStride);
    if (conservative) __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// polls until the flag has reached `epoch`; gives up after `timeout` ticks of the 100 MHz wall clock and says so in *status (host memory)
__device__ __forceinline__ void await_flag(const unsigned char* arena, size_t off, int slot, uint64_t epoch, long long timeout, uint32_t* status, uint32_t bit) {
    const uint64_t* f = (const uint64_t*)(arena + off + (size_t)slot * kFlagStride);
    const long long t0 = wall_clock64();
    // every workgroup of a launch polls the same line: back off (64 ... 2048 cycles between two loads) so that the pollers of a large
    // launch do not keep the memory channel busy that the awaited store has to pass
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {   // relaxed: an acquire per poll would invalidate caches per poll
        if (spins < 2u) __builtin_amdgcn_s_sleep(1);
        else if (spins < 4u) __builtin_amdgcn_s_sleep(4);
        else if (spins < 8u) __builtin_amdgcn_s_sleep(12);
        else __builtin_amdgcn_s_sleep(32);
        if ((++spins & 255u) == 0u && wall_clock64() - t0 > timeout) {
            // one word per kind of time-out: a plain store to host memory (no PCIe atomic needed)
            __hip_atomic_store(status + (bit == LCD_P2P_TIMEOUT_GATHER ? 0 : bit == LCD_P2P_TIMEOUT_STAGE ? 1 : 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                    // ONE acquire at system scope once the flag is there
}
// true in exactly one workgroup of the launch: the one that finishes last.  Every wave waits for the acknowledgement of its own stores
// before its workgroup counts, so when the last one has counted every store of the launch is in memory: the flag may follow.
__device__ __forceinline__ bool last_workgroup(uint32_t* counter, unsigned n_groups, int conservative) {
    __shared__ int s_last;
    if (conservative) __threadfence_system();                        // the compiler's release: L2 write-back and all (see the head of the file)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = done == n_groups - 1u;
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch on this stream finds it clear
    }
    __syncthreads();
    return s_last != 0;
}

// grid (workgroups per peer, world): group p pushes my block to peer p, announces it, waits for p's block in my mailbox, copies it out
__global__ __launch_bounds__(256) void p2p_all_gather_kernel(Peers P, int rank, int world, const uint4* __restrict__ send, uint4* __restrict__ recv, size_t vec,
                                                             size_t gcap, uint64_t epoch, uint32_t* counters, long long timeout, uint32_t* status) {
    const int p = blockIdx.y;
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if (p == rank) {
        for (size_t i = first; i < vec; i += step) recv[(size_t)rank * vec + i] = send[i];
        return;
    }
    uint4* box = (uint4*)(P.base[p] + kMailboxOff + ((epoch & 1u) * (size_t)world + (size_t)rank) * gcap);
    for (size_t i = first; i < vec; i += step) box[i] = send[i];
    if (last_workgroup(&counters[p], gridDim.x, P.conservative) && threadIdx.x == 0) raise_flag(P.base[p], kGatherFlagOff, rank, epoch, P.conservative);
    if (threadIdx.x == 0) await_flag(P.base[rank], kGatherFlagOff, p, epoch, timeout, status, LCD_P2P_TIMEOUT_GATHER);
    __syncthreads();
    const uint4* mine = (const uint4*)(P.base[rank] + kMailboxOff + ((epoch & 1u) * (size_t)world + (size_t)p) * gcap);
    for (size_t i = first; i < vec; i += step) recv[(size_t)p * vec + i] = mine[i];
}

// The all-reduce's kernels move 16-byte groups (2 integers or 4 floats of the wire) per lane, two groups in flight per lane: arenas are
// uncached, so every access is a round trip to HBM (or across a link) and what a kernel takes is round trips in a row, not bytes.
constexpr int kCopyUnroll = 4;
template <typename W> struct Group;
template <> struct Group<long long> {
    static constexpr int N = 2;
    long long v[2];
    __device__ __forceinline__ void load_operand(const long long* b) { const longlong2 t = *(const longlong2*)b; v[0] = t.x; v[1] = t.y; }
    __device__ __forceinline__ void store_operand(long long* b) const { *(longlong2*)b = make_longlong2(v[0], v[1]); }
    __device__ __forceinline__ void load_wire(const long long* w) { load_operand(w); }
    __device__ __forceinline__ void store_wire(long long* w) const { store_operand(w); }
};
template <> struct Group<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load_operand(const long long* b) {
        const longlong2 t0 = *(const longlong2*)b, t1 = *(const longlong2*)(b + 2);
        v[0] = (float)t0.x; v[1] = (float)t0.y; v[2] = (float)t1.x; v[3] = (float)t1.y;
    }
    __device__ __forceinline__ void store_operand(long long* b) const {
        *(longlong2*)b = make_longlong2(__float2ll_rn(v[0]), __float2ll_rn(v[1]));
        *(longlong2*)(b + 2) = make_longlong2(__float2ll_rn(v[2]), __float2ll_rn(v[3]));
    }
    __device__ __forceinline__ void load_wire(const float* w) { const float4 t = *(const float4*)w; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ __forceinline__ void store_wire(float* w) const { *(float4*)w = make_float4(v[0], v[1], v[2], v[3]); }
};
__device__ __forceinline__ long long wire_to_operand(long long w) { return w; }
__device__ __forceinline__ long long wire_to_operand(float w) { return __float2ll_rn(w); }

// all-reduce, phase 1: the operand into my own arena (as the wire's type), flag A to every peer
template <typename W>
__global__ __launch_bounds__(256) void p2p_stage_kernel(Peers P, int rank, int world, const long long* __restrict__ buf, size_t count, size_t stage_off,
                                                        uint64_t epoch, uint32_t* counter) {
    constexpr int N = Group<W>::N;
    W* s = (W*)(P.base[rank] + stage_off);
    const size_t groups = count / N;
    for (size_t g = (size_t)blockIdx.x * (256 * kCopyUnroll) + threadIdx.x; g < groups; g += (size_t)gridDim.x * (256 * kCopyUnroll)) {
        Group<W> a[kCopyUnroll];
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) if (g + k * 256 < groups) a[k].load_operand(buf + (g + k * 256) * N);
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) if (g + k * 256 < groups) a[k].store_wire(s + (g + k * 256) * N);
    }
    if (blockIdx.x == 0 && groups * N + threadIdx.x < count) s[groups * N + threadIdx.x] = (W)buf[groups * N + threadIdx.x];
    if (last_workgroup(counter, gridDim.x, P.conservative) && (int)threadIdx.x < world) raise_flag(P.base[threadIdx.x], kStageFlagOff, rank, epoch, P.conservative);
}
// phase 2: slice `rank` of every arena summed in rank order and written back into slice `rank` of every arena, flag B to every peer.
// Only this rank touches slice `rank` of any arena between the two flags, so the exchange needs no third barrier.
template <typename W>
__global__ __launch_bounds__(256) void p2p_reduce_kernel(Peers P, int rank, int world, size_t count, size_t chunk, size_t stage_off, uint64_t epoch,
                                                         uint32_t* counter, long long timeout, uint32_t* status) {
    constexpr int N = Group<W>::N;
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) await_flag(P.base[rank], kStageFlagOff, threadIdx.x, epoch, timeout, status, LCD_P2P_TIMEOUT_STAGE);
    __syncthreads();
    const size_t lo = (size_t)rank * chunk < count ? (size_t)rank * chunk : count, hi = lo + chunk < count ? lo + chunk : count;   // lo: a multiple of 4 elements
    const size_t groups = (hi - lo) / N;
    for (size_t g0 = (size_t)blockIdx.x * 512 + threadIdx.x; g0 < groups; g0 += (size_t)gridDim.x * 512) {
        Group<W> v[2][LCD_P2P_MAX_WORLD];
        const bool two = g0 + 256 < groups;
#pragma unroll
        for (int p = 0; p < LCD_P2P_MAX_WORLD; ++p)                                           // every link at once, two groups per lane in flight
            if (p < world) {
                v[0][p].load_wire((const W*)(P.base[p] + stage_off) + lo + g0 * N);
                if (two) v[1][p].load_wire((const W*)(P.base[p] + stage_off) + lo + (g0 + 256) * N);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            Group<W> sum = v[h][0];
#pragma unroll
            for (int p = 1; p < LCD_P2P_MAX_WORLD; ++p)
                if (p < world) {
#pragma unroll
                    for (int k = 0; k < N; ++k) sum.v[k] += v[h][p].v[k];
                }
#pragma unroll
            for (int p = 0; p < LCD_P2P_MAX_WORLD; ++p) if (p < world) sum.store_wire((W*)(P.base[p] + stage_off) + lo + (g0 + h * 256) * N);
        }
    }
    if (blockIdx.x == 0 && lo + groups * N + threadIdx.x < hi) {
        const size_t i = lo + groups * N + threadIdx.x;
        W sum = ((const W*)(P.base[0] + stage_off))[i];
        for (int p = 1; p < world; ++p) sum += ((const W*)(P.base[p] + stage_off))[i];
        for (int p = 0; p < world; ++p) ((W*)(P.base[p] + stage_off))[i] = sum;
    }
    if (last_workgroup(counter, gridDim.x, P.conservative) && (int)threadIdx.x < world) raise_flag(P.base[threadIdx.x], kReduceFlagOff, rank, epoch, P.conservative);
}
// phase 3: every slice of my arena is final once every peer has raised B
template <typename W>
__global__ __launch_bounds__(256) void p2p_collect_kernel(Peers P, int rank, int world, long long* __restrict__ buf, size_t count, size_t stage_off, uint64_t epoch,
                                                          long long timeout, uint32_t* status) {
    constexpr int N = Group<W>::N;
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) await_flag(P.base[rank], kReduceFlagOff, threadIdx.x, epoch, timeout, status, LCD_P2P_TIMEOUT_REDUCE);
    __syncthreads();
    const W* s = (const W*)(P.base[rank] + stage_off);
    const size_t groups = count / N;
    for (size_t g = (size_t)blockIdx.x * (256 * kCopyUnroll) + threadIdx.x; g < groups; g += (size_t)gridDim.x * (256 * kCopyUnroll)) {
        Group<W> a[kCopyUnroll];
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) if (g + k * 256 < groups) a[k].load_wire(s + (g + k * 256) * N);
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) if (g + k * 256 < groups) a[k].store_operand(buf + (g + k * 256) * N);
    }
    if (blockIdx.x == 0 && groups * N + threadIdx.x < count) buf[groups * N + threadIdx.x] = wire_to_operand(s[groups * N + threadIdx.x]);
}

struct Export {                                                      // LCD_P2P_HANDLE_BYTES
    hipIpcMemHandle_t handle;                                        // 64 bytes
    uint64_t ptr, arena_bytes, gather_cap, reduce_count_max;
    int32_t pid, rank, world, device;
    uint32_t magic;
    unsigned char pad[LCD_P2P_HANDLE_BYTES - 64 - 32 - 16 - 4];
};
static_assert(sizeof(Export) == LCD_P2P_HANDLE_BYTES, "export record");

}  // namespace

struct lcd_p2p {
    int rank = 0, world = 1, device = 0;
    size_t gather_cap = 0, reduce_count_max = 0, stage_off = 0, arena_bytes = 0;
    unsigned char* arena = nullptr;
    Peers peers{};
    bool opened[LCD_P2P_MAX_WORLD] = {};
    bool connected = false;
    uint32_t* d_counters = nullptr;                                  // [0 .. world) all-gather groups, [world] stage, [world + 1] reduce
    uint32_t* h_status = nullptr;                                    // pinned, three words (gather / stage / reduce) set by kernels that time out
    uint64_t gather_epoch = 0, reduce_epoch = 0;
    int wire = LCD_P2P_WIRE_I64;
    long long timeout_ticks = 10000LL * 100000LL;                    // 10 s of the 100 MHz wall clock
    std::string err;
    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {
// elements per rank's slice of an all-reduce: the ranks' shares rounded up to 4 elements, so that every slice starts on a 16-byte boundary of
// either wire (slice r = [min(r * chunk, count), min((r + 1) * chunk, count)): the last slices may be short or empty)
inline size_t all_reduce_chunk(size_t count, int world) { return ((count + (size_t)world - 1) / (size_t)world + 3) / 4 * 4; }

template <typename W>
int all_reduce_launch(lcd_p2p* p, long long* buf, size_t count, hipStream_t s) {
    const uint64_t epoch = ++p->reduce_epoch;
    const size_t chunk = all_reduce_chunk(count, p->world);
    const size_t per_group = 16 / sizeof(W);
    unsigned g_copy = (unsigned)((count / per_group + 256 * kCopyUnroll - 1) / (256 * kCopyUnroll)); if (g_copy > 256) g_copy = 256; if (g_copy < 1) g_copy = 1;
    unsigned g_red = (unsigned)((chunk / per_group + 511) / 512); if (g_red > 256) g_red = 256; if (g_red < 1) g_red = 1;   // two groups per lane
    hipLaunchKernelGGL(HIP_KERNEL_NAME(p2p_stage_kernel<W>), dim3(g_copy), dim3(256), 0, s, p->peers, p->rank, p->world, (const long long*)buf, count, p->stage_off,
                       epoch, p->d_counters + LCD_P2P_MAX_WORLD);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(p2p_reduce_kernel<W>), dim3(g_red), dim3(256), 0, s, p->peers, p->rank, p->world, count, chunk, p->stage_off, epoch,
                       p->d_counters + LCD_P2P_MAX_WORLD + 1, p->timeout_ticks, p->h_status);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(p2p_collect_kernel<W>), dim3(g_copy), dim3(256), 0, s, p->peers, p->rank, p->world, buf, count, p->stage_off, epoch,
                       p->timeout_ticks, p->h_status);
    return hipGetLastError() == hipSuccess ? LCD_OK : p->fail(LCD_ERR_HIP, "all-reduce kernel launch");
}
}  // namespace

extern "C" {

int lcd_p2p_create(int rank, int world, size_t gather_bytes_per_rank_max, size_t reduce_count_max, lcd_p2p** out) {
    if (!out) return LCD_ERR_INVALID;
    *out = nullptr;
    if (world < 1 || world > LCD_P2P_MAX_WORLD || rank < 0 || rank >= world) return LCD_ERR_INVALID;
    try {
        lcd_p2p* p = new (std::nothrow) lcd_p2p();
        if (!p) return LCD_ERR_NOMEM;
        p->rank = rank; p->world = world;
        if (hipGetDevice(&p->device) != hipSuccess) { delete p; return LCD_ERR_HIP; }
        p->gather_cap = (gather_bytes_per_rank_max + 255) / 256 * 256;
        p->reduce_count_max = reduce_count_max;
        p->stage_off = kMailboxOff + 2 * (size_t)world * p->gather_cap;
        p->arena_bytes = (p->stage_off + (reduce_count_max + 4) * sizeof(long long) + 4095) / 4096 * 4096;
        // uncached or nothing: a cached arena would work between two processes of ONE GPU (the test box) and read stale lines across xGMI.
        // LCD_P2P_ARENA_KIND=finegrained|coarse is a measurement switch (tools/p2p_bench.py on one GPU), not a mode.
        const char* kind = getenv("LCD_P2P_ARENA_KIND");
        hipError_t me;
        if (kind && !strcmp(kind, "coarse")) me = hipMalloc((void**)&p->arena, p->arena_bytes);
        else if (kind && !strcmp(kind, "finegrained")) me = hipExtMallocWithFlags((void**)&p->arena, p->arena_bytes, hipDeviceMallocFinegrained);
        else me = hipExtMallocWithFlags((void**)&p->arena, p->arena_bytes, hipDeviceMallocUncached);   // (no cached fallback: see the memory model above)
        if (me != hipSuccess) { (void)hipGetLastError(); delete p; return LCD_ERR_NOMEM; }
        if (hipMemset(p->arena, 0, p->arena_bytes) != hipSuccess ||
            hipMalloc((void**)&p->d_counters, (LCD_P2P_MAX_WORLD + 2) * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(p->d_counters, 0, (LCD_P2P_MAX_WORLD + 2) * sizeof(uint32_t)) != hipSuccess ||
            hipHostMalloc((void**)&p->h_status, 4 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { lcd_p2p_destroy(p); return LCD_ERR_HIP; }
        for (int i = 0; i < 4; ++i) p->h_status[i] = 0u;
        p->peers.base[rank] = p->arena;
        if (world == 1) p->connected = true;
        *out = p;
        return LCD_OK;
    } catch (...) { return LCD_ERR_NOMEM; }
}

int lcd_p2p_export(lcd_p2p* p, unsigned char out[LCD_P2P_HANDLE_BYTES]) {
    if (!p || !out) return LCD_ERR_INVALID;
    try {
        Export e;
        std::memset(&e, 0, sizeof(e));
        if (hipSetDevice(p->device) != hipSuccess) return p->fail(LCD_ERR_HIP, "hipSetDevice");
        if (hipIpcGetMemHandle(&e.handle, p->arena) != hipSuccess) {
            (void)hipGetLastError();
            if (p->world > 1) p->err = "hipIpcGetMemHandle failed: ranks of other processes cannot map this arena";   // same-process peers still can
            std::memset(&e.handle, 0, sizeof(e.handle));
        }
        e.ptr = (uint64_t)(uintptr_t)p->arena; e.arena_bytes = p->arena_bytes; e.gather_cap = p->gather_cap; e.reduce_count_max = p->reduce_count_max;
        e.pid = (int32_t)getpid(); e.rank = p->rank; e.world = p->world; e.device = p->device; e.magic = kMagic;
        std::memcpy(out, &e, sizeof(e));
        return LCD_OK;
    } catch (...) { return p->fail(LCD_ERR_STATE, "unexpected exception"); }
}

int lcd_p2p_connect(lcd_p2p* p, const unsigned char* all_exports) {
    if (!p || !all_exports) return LCD_ERR_INVALID;
    try {
        if (p->connected && p->world > 1) return p->fail(LCD_ERR_STATE, "lcd_p2p_connect: already connected");
        if (hipSetDevice(p->device) != hipSuccess) return p->fail(LCD_ERR_HIP, "hipSetDevice");
        for (int r = 0; r < p->world; ++r) {
            Export e;
            std::memcpy(&e, all_exports + (size_t)r * LCD_P2P_HANDLE_BYTES, sizeof(e));
            if (e.magic != kMagic || e.rank != r || e.world != p->world || e.gather_cap != p->gather_cap || e.reduce_count_max != p->reduce_count_max ||
                e.arena_bytes != p->arena_bytes)
                return p->fail(LCD_ERR_INVALID, "lcd_p2p_connect: export of rank " + std::to_string(r) + " does not match this rank's capacities / world");
            if (r == p->rank) continue;
            if (e.pid == (int32_t)getpid()) {                        // a rank of this process: its pointer is valid here; peers on other devices need access
                if (e.device != p->device) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(e.device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return p->fail(LCD_ERR_HIP, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
                p->peers.base[r] = (unsigned char*)(uintptr_t)e.ptr;
            } else {
                void* m = nullptr;
                if (hipIpcOpenMemHandle(&m, e.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                    return p->fail(LCD_ERR_HIP, "hipIpcOpenMemHandle(rank " + std::to_string(r) + ")");
                p->peers.base[r] = (unsigned char*)m;
                p->opened[r] = true;
            }
        }
        p->connected = true;
        return LCD_OK;
    } catch (...) { return p->fail(LCD_ERR_STATE, "unexpected exception"); }
}

void lcd_p2p_destroy(lcd_p2p* p) {
    if (!p) return;
    try {
        (void)hipSetDevice(p->device);
        (void)hipDeviceSynchronize();
        for (int r = 0; r < p->world; ++r) if (p->opened[r]) (void)hipIpcCloseMemHandle(p->peers.base[r]);
        if (p->arena) (void)hipFree(p->arena);
        if (p->d_counters) (void)hipFree(p->d_counters);
        if (p->h_status) (void)hipHostFree(p->h_status);
        delete p;
    } catch (...) { }
}

const char* lcd_p2p_last_error(const lcd_p2p* p) { return p ? p->err.c_str() : "null lcd_p2p"; }

int lcd_p2p_set_wire(lcd_p2p* p, int wire) {
    if (!p) return LCD_ERR_INVALID;
    if (wire != LCD_P2P_WIRE_I64 && wire != LCD_P2P_WIRE_F32) return p->fail(LCD_ERR_INVALID, "lcd_p2p_set_wire: unknown wire");
    p->wire = wire;
    return LCD_OK;
}
int lcd_p2p_set_conservative_fences(lcd_p2p* p, int on) {
    if (!p) return LCD_ERR_INVALID;
    p->peers.conservative = on ? 1 : 0;
    return LCD_OK;
}
int lcd_p2p_set_timeout_ms(lcd_p2p* p, int64_t ms) {
    if (!p) return LCD_ERR_INVALID;
    if (ms <= 0 || ms > 600000) return p->fail(LCD_ERR_INVALID, "lcd_p2p_set_timeout_ms: 1 .. 600 000");
    p->timeout_ticks = (long long)ms * 100000LL;
    return LCD_OK;
}
uint32_t lcd_p2p_status(const lcd_p2p* p) {
    if (!p || !p->h_status) return 0u;
    uint32_t s = 0u;
    for (int i = 0; i < 3; ++i) if (__atomic_load_n(p->h_status + i, __ATOMIC_ACQUIRE)) s |= 1u << i;
    return s;
}
void lcd_p2p_clear_status(lcd_p2p* p) {
    if (p && p->h_status) for (int i = 0; i < 3; ++i) __atomic_store_n(p->h_status + i, 0u, __ATOMIC_RELEASE);
}

int lcd_p2p_all_gather(lcd_p2p* p, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
    if (!p) return LCD_ERR_INVALID;
    try {
        if (!d_send || !d_recv || bytes_per_rank == 0 || bytes_per_rank % 16 != 0 || ((uintptr_t)d_send | (uintptr_t)d_recv) % 16 != 0)
            return p->fail(LCD_ERR_INVALID, "lcd_p2p_all_gather: blocks are multiples of 16 bytes in buffers aligned to 16");
        if (bytes_per_rank > p->gather_cap) return p->fail(LCD_ERR_INVALID, "lcd_p2p_all_gather: block larger than gather_bytes_per_rank_max");
        if (!p->connected) return p->fail(LCD_ERR_STATE, "lcd_p2p_all_gather: not connected");
        if (hipSetDevice(p->device) != hipSuccess) return p->fail(LCD_ERR_HIP, "hipSetDevice");
        hipStream_t s = (hipStream_t)stream;
        if (p->world == 1) {
            if (hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, s) != hipSuccess) return p->fail(LCD_ERR_HIP, "hipMemcpyAsync");
            return LCD_OK;
        }
        const size_t vec = bytes_per_rank / 16;
        unsigned per_peer = (unsigned)((vec + 1023) / 1024);         // 16 KB per workgroup pass
        if (per_peer > 16) per_peer = 16;
        ++p->gather_epoch;
        hipLaunchKernelGGL(p2p_all_gather_kernel, dim3(per_peer, (unsigned)p->world), dim3(256), 0, s, p->peers, p->rank, p->world, (const uint4*)d_send,
                           (uint4*)d_recv, vec, p->gather_cap, p->gather_epoch, p->d_counters, p->timeout_ticks, p->h_status);
        if (hipGetLastError() != hipSuccess) return p->fail(LCD_ERR_HIP, "p2p_all_gather_kernel launch");
        return LCD_OK;
    } catch (...) { return p->fail(LCD_ERR_STATE, "unexpected exception"); }
}


int lcd_p2p_all_reduce_sum_i64(lcd_p2p* p, void* d_buf, size_t count, void* stream) {
    if (!p) return LCD_ERR_INVALID;
    try {
        if (count == 0) return LCD_OK;
        if (!d_buf || (uintptr_t)d_buf % 16 != 0) return p->fail(LCD_ERR_INVALID, "lcd_p2p_all_reduce_sum_i64: the buffer is aligned to 16 bytes");
        if (count > p->reduce_count_max) return p->fail(LCD_ERR_INVALID, "lcd_p2p_all_reduce_sum_i64: count larger than reduce_count_max");
        if (!p->connected) return p->fail(LCD_ERR_STATE, "lcd_p2p_all_reduce_sum_i64: not connected");
        if (p->world == 1) return LCD_OK;
        if (hipSetDevice(p->device) != hipSuccess) return p->fail(LCD_ERR_HIP, "hipSetDevice");
        return p->wire == LCD_P2P_WIRE_F32 ? all_reduce_launch<float>(p, (long long*)d_buf, count, (hipStream_t)stream)
                                           : all_reduce_launch<long long>(p, (long long*)d_buf, count, (hipStream_t)stream);
    } catch (...) { return p->fail(LCD_ERR_STATE, "unexpected exception"); }
}

namespace {
int tr_gather(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
    return lcd_p2p_all_gather((lcd_p2p*)user, d_send, d_recv, bytes_per_rank, stream);
}
int tr_reduce(void* user, void* d_buf, size_t count, void* stream) { return lcd_p2p_all_reduce_sum_i64((lcd_p2p*)user, d_buf, count, stream); }
}  // namespace

/* not part of the ABI (tests/test_p2p_host_logic.py): the slice arithmetic of the all-reduce, callable without a device */
size_t lcd_p2p_debug_chunk(size_t count, int world) { return world >= 1 ? all_reduce_chunk(count, world) : 0; }

int lcd_p2p_transport(lcd_p2p* p, lcd_shard_transport* out) {
    if (!p || !out) return LCD_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->struct_size = (int32_t)sizeof(lcd_shard_transport);
    out->user = p;
    out->all_gather = tr_gather;
    out->all_reduce_sum_i64 = tr_reduce;
    return LCD_OK;
}

}  // extern "C"
