// Device-side all-reduce of a few hundred doubles between the ranks of ONE node, for the SyncBN statistics exchanges
// (/root/reference/models/model.py:25: ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm): 158 blocking, latency-bound
// collectives per training step, each 2c + 1 ... 3c doubles.  A ring all-reduce through the collective library pays its
// launch and protocol latency per call; here every rank owns a MAILBOX in device memory that its peers have mapped through
// HIP IPC (b2m_xchg_alloc -> handle -> b2m_xchg_open on the peers), and ONE launch of one workgroup
//   1. writes this rank's values into slot [rank] of every rank's mailbox, fences, raises flag [rank] = epoch in each;
//   2. waits until every flag of its OWN mailbox shows the epoch;
//   3. adds the slots up in rank order (the same bits on every rank) and writes the result.
// Two slot sets alternate with the epoch's parity: a rank cannot finish exchange e + 1 before every rank has entered it, and a
// rank enters e + 1 only after it has read exchange e, so the set written for e + 2 is never one still being read.
// The wait is BOUNDED (B2M_XCHG_TIMEOUT_S, default 120 s of wall time): a peer that never arrives sets *err = 1, the result is
// NaN, and the kernel ends -- a kernel that spins for ever takes the device down.  Opt-in (B2M_SYNCBN_IPC=1, box2mask_amd/parallel.py): what has run is two processes
// on one GPU (tests/test_gpu_dp.py); across GPUs it needs peer-visible (fine-grained) memory, which the allocation asks for
// and which no lease of the build pool could exercise.
#include "b2m_common.h"
#include <string.h>
#include <stdlib.h>

#define XCHG_SLOT 2048            // doubles per rank and exchange (the paired BatchNorm of a 256-channel block sends 4c + 1 = 1025)
#define XCHG_MAX_RANKS 16
struct XchgMailbox {
    double slots[2][XCHG_MAX_RANKS][XCHG_SLOT];
    unsigned long long flags[2][XCHG_MAX_RANKS];
};

extern "C" int64_t b2m_xchg_size(void) { return (int64_t)sizeof(XchgMailbox); }
extern "C" int32_t b2m_xchg_max_doubles(void) { return XCHG_SLOT; }
extern "C" int32_t b2m_xchg_max_ranks(void) { return XCHG_MAX_RANKS; }

// mailboxes of this process that are fine-grained (peer-coherent) allocations: b2m_xchg_is_finegrained
static void* g_fine[XCHG_MAX_RANKS];
static int g_nfine = 0;

extern "C" int b2m_xchg_alloc(void** buf, void* handle64) {
    B2M_CHECK_ARG(buf && handle64, "NULL argument");
    bool fine = true;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the IPC handle travels as 64 bytes");
    void* p = nullptr;
    // fine-grained device memory: writes of a running kernel on another device become visible without a kernel boundary;
    // plain device memory if the runtime refuses (one GPU: every rank's mailbox lives on it, coherent through its L2)
    if (hipExtMallocWithFlags(&p, sizeof(XchgMailbox), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        B2M_HIP(hipMalloc(&p, sizeof(XchgMailbox)));
        fine = false;
    }
    B2M_HIP(hipMemset(p, 0, sizeof(XchgMailbox)));
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        // (no IPC handle for fine-grained memory on this runtime: fall back to a plain allocation)
        (void)hipGetLastError();
        (void)hipFree(p);
        B2M_HIP(hipMalloc(&p, sizeof(XchgMailbox)));
        B2M_HIP(hipMemset(p, 0, sizeof(XchgMailbox)));
        B2M_HIP(hipIpcGetMemHandle(&h, p));
        fine = false;
    }
    B2M_HIP(hipDeviceSynchronize());
    ::memcpy(handle64, &h, 64);
    *buf = p;
    if (fine && g_nfine < XCHG_MAX_RANKS) g_fine[g_nfine++] = p;
    return B2M_OK;
}
extern "C" int32_t b2m_xchg_is_finegrained(const void* buf) {
    for (int i = 0; i < g_nfine; ++i)
        if (g_fine[i] == buf) return 1;
    return 0;
}
extern "C" int b2m_xchg_open(const void* handle64, void** ptr) {
    B2M_CHECK_ARG(handle64 && ptr, "NULL argument");
    hipIpcMemHandle_t h;
    ::memcpy(&h, handle64, 64);
    B2M_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return B2M_OK;
}
extern "C" int b2m_xchg_close(void* ptr) {
    if (ptr) B2M_HIP(hipIpcCloseMemHandle(ptr));
    return B2M_OK;
}
extern "C" int b2m_xchg_free(void* buf) {
    for (int i = 0; i < g_nfine; ++i)
        if (g_fine[i] == buf) { g_fine[i] = g_fine[--g_nfine]; break; }
    if (buf) B2M_HIP(hipFree(buf));
    return B2M_OK;
}

// The wait is bounded in WALL time (s_memrealtime: 100 MHz whatever the shader clock): B2M_XCHG_TIMEOUT_S seconds, default 120
// -- a rank that is late by a loader start-up, a checkpoint or a validation pass is waited for, as torch.distributed would; a
// rank that is gone ends the wait, sets *err and POISONS the result with NaN, so that nothing trains on the stale slots of
// exchange e - 2 (the host checks *err at every step boundary: parallel.IpcExchange.check_async).
__global__ __launch_bounds__(256) void xchg_allreduce_kernel(const double* vals, int n, XchgMailbox* const* __restrict__ peers,
                                                             int rank, int world, unsigned long long epoch, double* out,
                                                             int* __restrict__ err, unsigned long long timeout_ticks) {
    __shared__ int timed_out;
    const int par = (int)(epoch & 1ull);
    const int tid = threadIdx.x;
    if (tid == 0) timed_out = 0;
    // (vals and out may be the same array: every element is read into the mailboxes here, before anything is written to out)
    for (int p = 0; p < world; ++p) {
        double* dst = peers[p]->slots[par][rank];
        for (int i = tid; i < n; i += 256) dst[i] = vals[i];
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world)
        __hip_atomic_store(&peers[tid]->flags[par][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    XchgMailbox* me = peers[rank];
    if (tid < world) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&me->flags[par][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
            if (wall_clock64() - t0 > timeout_ticks) { timed_out = 1; *err = 1; break; }     // a peer never arrived
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    __threadfence_system();
    const bool bad = timed_out != 0;
    for (int i = tid; i < n; i += 256) {
        double s = 0.0;
        for (int q = 0; q < world; ++q) s += __builtin_nontemporal_load(&me->slots[par][q][i]);
        out[i] = bad ? __builtin_nan("") : s;
    }
}

static unsigned long long xchg_timeout_ticks() {
    static unsigned long long ticks = 0;
    if (!ticks) {
        const char* e = getenv("B2M_XCHG_TIMEOUT_S");
        double sec = e ? atof(e) : 120.0;
        if (!(sec > 0.0)) sec = 120.0;
        ticks = (unsigned long long)(sec * 1e8);            // s_memrealtime counts at 100 MHz
    }
    return ticks;
}

extern "C" int b2m_xchg_allreduce(const double* vals, int32_t n, const void* const* peers_dev, int32_t rank, int32_t world,
                                  uint64_t epoch, double* out, int32_t* err, void* stream) {
    B2M_CHECK_ARG(vals && peers_dev && out && err, "NULL argument");
    B2M_CHECK_ARG(n >= 1 && n <= XCHG_SLOT, "1 <= n <= b2m_xchg_max_doubles()");
    B2M_CHECK_ARG(world >= 1 && world <= XCHG_MAX_RANKS && rank >= 0 && rank < world && epoch >= 1, "bad rank / world / epoch");
    xchg_allreduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(vals, n, (XchgMailbox* const*)peers_dev, rank, world, (unsigned long long)epoch, out, err,
                                                              xchg_timeout_ticks());
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
