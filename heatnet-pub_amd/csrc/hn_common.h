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

// ---- f16x2 tensor scales (include/heatnet_hip.h) ----------------------------------------------------------------------
// A tensor's scale slot is HN_SCALE_WORDS 32-bit words, HN_SCALE_STRIDE words apart (one per 128-B line), each the bit
// pattern of the largest magnitude seen by the blocks that hash to it (as an unsigned integer the pattern is monotonic in
// the magnitude; the words spread the producers' atomics over 64 lines).  Consumers
// take the maximum over the words and derive s = 2^(14 - floor(log2 max|x|)), which puts that magnitude into
// [2^14, 2^15) -- below fp16's 65504, and every element within 2^-18 of it keeps a normal `lo` term (22 significant
// bits).  All-zero, infinite or NaN: s = 1.
__device__ __forceinline__ void hn_h2_scales(unsigned bits, float& s, float& inv) {
    const int e = (int)(bits >> 23);  // biased exponent of the maximum (sign bit already cleared)
    s = 1.f;
    inv = 1.f;
    if (e > 0 && e < 255) {
        int se = 127 + 14 - (e - 127);      // biased exponent of s
        se = se < 1 ? 1 : (se > 253 ? 253 : se);
        s = __uint_as_float((unsigned)se << 23);
        inv = __uint_as_float((unsigned)(254 - se) << 23);
    }
}

// wave-uniform: every lane of the calling wave takes part (one word per lane, six cross-lane steps)
__device__ __forceinline__ void hn_h2_scales_of_slot(const float* slot, float& s, float& inv) {
    const int lane = (threadIdx.y * blockDim.x + threadIdx.x) & (HN_SCALE_WORDS - 1);
    unsigned m = reinterpret_cast<const unsigned*>(slot)[lane * HN_SCALE_STRIDE];
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    hn_h2_scales((unsigned)__builtin_amdgcn_readfirstlane((int)m), s, inv);
}

// Order-preserving key of a float (key(a) < key(b) <=> a < b for non-NaN values); 0 is below every key a kernel produces
// ("nothing seen"), so accumulators of atomicMax(key) start from zero-filled memory.
__device__ __forceinline__ unsigned hn_range_key(float f) {
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float hn_range_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// Every thread of a 256-thread block (1-D or 2-D) calls this with the largest |value| bit pattern it produced (0 if none):
// ONE no-return atomic max per block into word (block % HN_SCALE_WORDS).  Device-scope atomics on one word serialise at
// ~11 ns each (MI355X_MICROARCH.md, fan-in): the words sit on 64 different 128-B lines.
__device__ __forceinline__ void hn_absmax_publish(unsigned local, float* slot) {
    __shared__ unsigned red[4];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) local = max(local, (unsigned)__shfl_xor((int)local, o));
    if ((t & 63) == 0) red[t >> 6] = local;
    __syncthreads();
    if (t == 0) {
        const unsigned m = max(max(red[0], red[1]), max(red[2], red[3]));
        const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) & (HN_SCALE_WORDS - 1);
        if (m != 0u) atomicMax(reinterpret_cast<unsigned*>(slot) + w * HN_SCALE_STRIDE, m);
    }
}
