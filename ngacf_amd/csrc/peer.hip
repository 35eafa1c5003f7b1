// Peer-store all-gather for the node-partitioned path (ngacf_amd/peer.py): a rank WRITES its block of a half-table straight into
// its peers' copies of the table (device pointers mapped through hipIpc) and then publishes a counter in each peer's flag word;
// a consumer spins on its own flag words in front of its traversal.  No collective launch, no host on the path; both kernels
// take their epoch from device memory, so a captured hipGraph replays them unchanged.
//
//   push   grid-stride 16-byte copies of (peer, region, chunk) as system-scope write-through stores; every workgroup: vmcnt(0)
//          (its stores are acknowledged), barrier, then a ticket; the LAST workgroup (ticket == grid - 1): epoch = ++*sent, one
//          system-scope store of the epoch per peer.  A flag can therefore only be seen after every byte of every region has
//          been written.  (The counter form of the guide's fence-free hand-off, at system scope: no cache-wide write-back.)
//   wait   one wave: lane r polls flag word r (system-scope loads: served at the coherent level, never by a cache) until it
//          has reached ++*expect, or `timeout_ticks` of the 100 MHz clock have gone by (then *err = 1 and the wave LEAVES: a
//          stalled peer costs a wrong answer that the host reports, not a hung GPU).  The traversal that follows is a new
//          kernel: its start-of-kernel acquire drops whatever its caches held of the table.
// The memory is plain hipMalloc memory owned by this library (ngacf_peer_alloc), exported / imported as hipIpc handles.
#include <hip/hip_runtime.h>
#include <string.h>

#include "ngacf.h"
#include "ngacf_common.hpp"

namespace ngacf {

struct PushArgs {
    ngacf_peer_region reg[NGACF_PEER_MAX_REGIONS];
    int32_t* flag[NGACF_MAX_PEERS];
    int nreg, npeers;
    int32_t* sent;
    int32_t* ticket;
};

// System-scope write-through stores (sc0 sc1): the bytes leave this GPU's caches with the store itself (to a peer's memory they
// are uncached anyway), so publishing needs no cache-wide write-back -- a `buffer_wbl2` here would also flush what the traversal
// running beside the push keeps in L2.  Inline asm: hipcc does not count these in its own vmcnt bookkeeping; the explicit
// s_waitcnt vmcnt(0) below is the only consumer of their completion.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sys(uint4* p, const uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void store_sys(uint32_t* p, const uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void peer_push_kernel(const PushArgs a) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (int r = 0; r < a.nreg; ++r) {
        const ngacf_peer_region& g = a.reg[r];
        const bool wide = ((reinterpret_cast<uintptr_t>(g.src) | g.bytes) & 15) == 0;
        for (int pp = 0; pp < a.npeers; ++pp) {
            const int p = (pp + (int)blockIdx.x) % a.npeers;      // workgroups start on different peers: every link busy at once
            if (wide && (reinterpret_cast<uintptr_t>(g.dst[p]) & 15) == 0) {
                const uint4* s = static_cast<const uint4*>(g.src);
                uint4* d = static_cast<uint4*>(g.dst[p]);
                for (size_t i = tid; i < g.bytes / 16; i += nth) store_sys(d + i, s[i]);
            } else {
                const uint32_t* s = static_cast<const uint32_t*>(g.src);
                uint32_t* d = static_cast<uint32_t*>(g.dst[p]);
                for (size_t i = tid; i < g.bytes / 4; i += nth) store_sys(d + i, s[i]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's stores have been acknowledged ...
    __syncthreads();                                          // ... and so have the workgroup's, before its ticket
    if (threadIdx.x != 0) return;
    const int t = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != (int)gridDim.x - 1) return;
    __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // clean for the next launch
    const int32_t e = *a.sent + 1;
    *a.sent = e;
    for (int p = 0; p < a.npeers; ++p) __hip_atomic_store(a.flag[p], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// No acquire in here: the kernel only has to END after the flags are up -- the traversal behind it is a new dispatch, and its
// start-of-kernel acquire is what drops stale lines of the table (an acquire fence in this wave would invalidate one XCD's L2
// under the kernels running beside it and help nobody).
__global__ __launch_bounds__(64) void peer_wait_kernel(const int32_t* flags, int n, int self, int32_t* expect, int32_t* err,
                                                       unsigned long long timeout_ticks) {
    const int lane = threadIdx.x;
    const int32_t e = *expect + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool late = false;
    if (lane < n && lane != self) {
        for (;;) {
            const int32_t v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)((uint32_t)v - (uint32_t)e) >= 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { late = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (late) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();                                          // (one wave: every lane has read *expect before lane 0 moves it on)
    if (lane == 0) *expect = e;
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_peer_alloc(uint64_t bytes, void** ptr) {
    if (!ptr || bytes == 0) return NGACF_E_ARG;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipMemset(*ptr, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

// The control block (flag words polled by running kernels, written by other agents): fine-grained, uncached device memory --
// coarse-grained hipMalloc memory has no coherence guarantee against a remote agent's writes while a kernel runs.
extern "C" int ngacf_peer_alloc_ctrl(uint64_t bytes, void** ptr) {
    if (!ptr || bytes == 0) return NGACF_E_ARG;
    hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipMemset(*ptr, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

extern "C" int ngacf_peer_free(void* ptr) {
    if (!ptr) return NGACF_E_ARG;
    const hipError_t e = hipFree(ptr);
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

static_assert(sizeof(hipIpcMemHandle_t) == NGACF_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");

extern "C" int ngacf_peer_export(const void* ptr, void* handle) {
    if (!ptr || !handle) return NGACF_E_ARG;
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(ptr));
    if (e != hipSuccess) return static_cast<int>(e);
    memcpy(handle, static_cast<const void*>(&h), sizeof(h));
    return 0;
}

extern "C" int ngacf_peer_import(const void* handle, void** ptr) {
    if (!ptr || !handle) return NGACF_E_ARG;
    hipIpcMemHandle_t h;
    memcpy(static_cast<void*>(&h), handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

extern "C" int ngacf_peer_close(void* ptr) {
    if (!ptr) return NGACF_E_ARG;
    const hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

extern "C" int ngacf_peer_push(const ngacf_peer_region* regions, int nregions, int npeers, int32_t* const* peer_flags,
                               int32_t* sent, int32_t* ticket, void* stream) {
    if (nregions < 0 || nregions > NGACF_PEER_MAX_REGIONS || npeers < 0 || npeers > NGACF_MAX_PEERS) return NGACF_E_ARG;
    if (!sent || !ticket || (nregions > 0 && !regions) || (npeers > 0 && !peer_flags)) return NGACF_E_ARG;
    PushArgs a;
    a.nreg = nregions; a.npeers = npeers; a.sent = sent; a.ticket = ticket;
    uint64_t most = 0;
    for (int r = 0; r < nregions; ++r) {
        a.reg[r] = regions[r];
        if (!regions[r].src || (regions[r].bytes & 3)) return NGACF_E_ARG;
        if (((reinterpret_cast<uintptr_t>(regions[r].src)) & 3) != 0) return NGACF_E_ARG;
        for (int p = 0; p < npeers; ++p)
            if (!regions[r].dst[p] || (reinterpret_cast<uintptr_t>(regions[r].dst[p]) & 3)) return NGACF_E_ARG;
        most = regions[r].bytes > most ? regions[r].bytes : most;
    }
    for (int p = 0; p < npeers; ++p) {
        if (!peer_flags[p]) return NGACF_E_ARG;
        a.flag[p] = peer_flags[p];
    }
    // enough workgroups to keep every xGMI link busy, few enough to leave the CUs to the traversal running beside the push
    uint64_t nb = (most / 16 + 255) / 256;
    if (nb < 1 || npeers == 0) nb = 1;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(peer_push_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_peer_wait(const int32_t* flags, int n, int self_rank, int32_t* expect, int32_t* err, uint64_t timeout_ticks,
                               void* stream) {
    if (n < 0 || n > 64 || !expect || !err || (n > 0 && !flags)) return NGACF_E_ARG;
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), flags, n, self_rank, expect, err,
                       (unsigned long long)timeout_ticks);
    NGACF_CHECK_LAUNCH();
    return 0;
}
