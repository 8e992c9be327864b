// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/b2m.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define B2M_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

void b2m_set_error(const char* fmt, ...);
// integer value of an environment switch, cached per process (coords.hip); `name` must be a string literal
int b2m_env_int(const char* name, int dflt);

#define B2M_CHECK_ARG(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            b2m_set_error("%s: %s", __func__, msg);                \
            return B2M_ERR_ARG;                                    \
        }                                                          \
    } while (0)

#define B2M_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) {                                                   \
            b2m_set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e__)); \
            return B2M_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define B2M_LAUNCH_CHECK()                                                           \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            b2m_set_error("%s: launch failed -> %s", __func__, hipGetErrorString(e__)); \
            return B2M_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
// rulebooks with at least this many tiles carry XCD work boundaries and a dispatch order (b2m_rulebook_balance), and the
// convolutions use them
#define B2M_BALANCE_MIN_TILES 64

__device__ __forceinline__ uint64_t b2m_pack(int b, int x, int y, int z) {
    return ((uint64_t)(uint32_t)b << 48) | ((uint64_t)(uint32_t)x << 32) | ((uint64_t)(uint32_t)y << 16) |
           (uint64_t)(uint32_t)z;
}

// 64-bit finaliser (splitmix64): spreads the structured coordinate keys over the table
__device__ __forceinline__ uint64_t b2m_hash(uint64_t k) {
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27; k *= 0x94d049bb133111ebull;
    k ^= k >> 31;
    return k;
}

// returns the slot of `key` or -1
__device__ __forceinline__ int64_t b2m_find(const uint64_t* __restrict__ keys, int64_t mask, uint64_t key) {
    int64_t s = (int64_t)(b2m_hash(key) & (uint64_t)mask);
    for (;;) {
        uint64_t kk = keys[s];
        if (kk == key) return s;
        if (kk == B2M_EMPTY_KEY) return -1;
        s = (s + 1) & mask;
    }
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// number of set bits of m strictly below this lane
__device__ __forceinline__ int prefix_popc(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
