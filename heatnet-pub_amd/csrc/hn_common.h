// Shared helpers for the gfx950 kernels of libheatnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/heatnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipGetLastError() is per-thread state shared with every other HIP user of the process (PyTorch leaves benign
// codes such as hipErrorNotReady behind), so it is cleared right before each launch and read right after it.
#define HN_LAUNCH(...) \
    do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define HN_CHECK_LAUNCH() \
    do { if (hipGetLastError() != hipSuccess) return HN_ELAUNCH; } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remember it per device ordinal (one bit each),
// so a process that drives several GPUs configures every instantiation once on each of them.  Thread-safe.
static inline int hn_ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HN_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if ((done_mask.load(std::memory_order_acquire) & bit) == 0) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return HN_ELAUNCH;
        done_mask.fetch_or(bit, std::memory_order_release);
    }
    return HN_OK;
}

// bf16x3 weight planes: plane stride in elements for a matrix of n elements (HN_X3_PLANE in the header)
static inline long long hn_x3_plane(long long n) { return HN_X3_PLANE(n); }

static inline bool hn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Remap the linear block id so
// that blocks sharing an XCD walk a CONTIGUOUS range of logical tile ids and re-use each other's
// operand panels from that L2.  Bijective for any block count (speed only, never correctness).
__device__ __forceinline__ int hn_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float hn_apply_act(float v, int act, float slope) {
    if (act == HN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == HN_ACT_PRELU || act == HN_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}
