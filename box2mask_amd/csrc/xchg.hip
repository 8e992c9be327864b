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
// The wait is BOUNDED (about seven seconds): a peer that never arrives sets *err = 1 and lets the kernel end -- a kernel that
// spins for ever takes the device down.  Opt-in (B2M_SYNCBN_IPC=1, box2mask_amd/parallel.py): what has run is two processes
// on one GPU (tests/test_gpu_dp.py); across GPUs it needs peer-visible (fine-grained) memory, which the allocation asks for
// and which no lease of the build pool could exercise.
#include "b2m_common.h"
#include <string.h>

#define XCHG_SLOT 2048            // doubles per rank and exchange (the paired BatchNorm of a 256-channel block sends 4c + 1 = 1025)
#define XCHG_MAX_RANKS 16
struct XchgMailbox {
    double slots[2][XCHG_MAX_RANKS][XCHG_SLOT];
    unsigned long long flags[2][XCHG_MAX_RANKS];
};

extern "C" int64_t b2m_xchg_size(void) { return (int64_t)sizeof(XchgMailbox); }
extern "C" int32_t b2m_xchg_max_doubles(void) { return XCHG_SLOT; }
extern "C" int32_t b2m_xchg_max_ranks(void) { return XCHG_MAX_RANKS; }

extern "C" int b2m_xchg_alloc(void** buf, void* handle64) {
    B2M_CHECK_ARG(buf && handle64, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the IPC handle travels as 64 bytes");
    void* p = nullptr;
    // fine-grained device memory: writes of a running kernel on another device become visible without a kernel boundary;
    // plain device memory if the runtime refuses (one GPU: every rank's mailbox lives on it, coherent through its L2)
    if (hipExtMallocWithFlags(&p, sizeof(XchgMailbox), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        B2M_HIP(hipMalloc(&p, sizeof(XchgMailbox)));
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
    }
    B2M_HIP(hipDeviceSynchronize());
    ::memcpy(handle64, &h, 64);
    *buf = p;
    return B2M_OK;
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
    if (buf) B2M_HIP(hipFree(buf));
    return B2M_OK;
}

#define XCHG_SPIN_LIMIT (1 << 25)       // x s_sleep 8 (512 cycles): about seven seconds at 2.4 GHz (ranks drift apart in their first steps)
__global__ __launch_bounds__(256) void xchg_allreduce_kernel(const double* __restrict__ vals, int n, XchgMailbox* const* __restrict__ peers,
                                                             int rank, int world, unsigned long long epoch, double* __restrict__ out,
                                                             int* __restrict__ err) {
    const int par = (int)(epoch & 1ull);
    const int tid = threadIdx.x;
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
        int it = 0;
        while (__hip_atomic_load(&me->flags[par][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
            if (++it > XCHG_SPIN_LIMIT) { *err = 1; break; }        // a peer never arrived: leave (results are garbage, the flag says so)
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __threadfence_system();
    for (int i = tid; i < n; i += 256) {
        double s = 0.0;
        for (int q = 0; q < world; ++q) s += __builtin_nontemporal_load(&me->slots[par][q][i]);
        out[i] = s;
    }
}

extern "C" int b2m_xchg_allreduce(const double* vals, int32_t n, const void* const* peers_dev, int32_t rank, int32_t world,
                                  uint64_t epoch, double* out, int32_t* err, void* stream) {
    B2M_CHECK_ARG(vals && peers_dev && out && err, "NULL argument");
    B2M_CHECK_ARG(n >= 1 && n <= XCHG_SLOT, "1 <= n <= b2m_xchg_max_doubles()");
    B2M_CHECK_ARG(world >= 1 && world <= XCHG_MAX_RANKS && rank >= 0 && rank < world && epoch >= 1, "bad rank / world / epoch");
    xchg_allreduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(vals, n, (XchgMailbox* const*)peers_dev, rank, world, (unsigned long long)epoch, out, err);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
