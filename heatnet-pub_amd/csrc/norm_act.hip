// HBM-bound per-channel kernels: BatchNorm apply / backward fused with the activation and the residual
// add, per-channel column reductions, Dropout2d scaling, strided copy / add.
//
// Every tensor is (ptr, ld) NHWC.  Thread layout: threadIdx.x walks the channel axis in 16-B pieces
// (coalesced), threadIdx.y walks pixels; a block is bx x by = 256 threads with bx = the smallest power
// of two covering the channel pieces (<= 64), so narrow tensors (C = 64) still fill their waves.
// Column reductions finish through LDS across threadIdx.y and one global_atomic_add_f32 per channel.
#include <cstdlib>

#include "hn_common.h"

namespace {

template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&v)[V]) {
    if constexpr (V == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        v[0] = *p;
    }
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&v)[V]) {
    if constexpr (V == 4) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    } else {
        *p = v[0];
    }
}

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// Plane tensors (f16x2 operands made by the producing pass, conv_igemm4.hip): element (pix, c) of a tensor t lives as
// hi = fp16(t * s) at planes[pix * ldp + c] and lo = fp16(t * s - hi) at planes[plane + pix * ldp + c], s = the power of two of
// the tensor's scale slot.  t is recovered as (hi + lo) / s (exact sums in fp32).
template <int V>
__device__ __forceinline__ void st_planes(unsigned short* __restrict__ hi, long long plane, const float (&v)[V], float s) {
    if constexpr (V == 4) {
        h16x4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xs = v[i] * s;
            h[i] = (_Float16)xs;
            l[i] = (_Float16)(xs - (float)h[i]);
        }
        *reinterpret_cast<h16x4*>(hi) = h;
        *reinterpret_cast<h16x4*>(hi + plane) = l;
    } else {
        const float xs = v[0] * s;
        const _Float16 h = (_Float16)xs;
        const _Float16 l = (_Float16)(xs - (float)h);
        hi[0] = __builtin_bit_cast(unsigned short, h);
        hi[plane] = __builtin_bit_cast(unsigned short, l);
    }
}
template <int V>
__device__ __forceinline__ void ld_planes(const unsigned short* __restrict__ hi, long long plane, float (&v)[V], float inv) {
    if constexpr (V == 4) {
        const h16x4 h = *reinterpret_cast<const h16x4*>(hi), l = *reinterpret_cast<const h16x4*>(hi + plane);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ((float)h[i] + (float)l[i]) * inv;
    } else {
        v[0] = ((float)__builtin_bit_cast(_Float16, hi[0]) + (float)__builtin_bit_cast(_Float16, hi[plane])) * inv;
    }
}

// One-plane bf16 tensors (the mixed-precision bf16 arithmetic, HN_MATH_BF16): element (pix, c) is bf16(t) at planes[pix * ldp + c];
// no scale (bf16 has fp32's exponent range).  Selected by a NULL scale slot wherever a plane operand is described.
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
template <int V>
__device__ __forceinline__ void st_bf16(unsigned short* __restrict__ p, const float (&v)[V]) {
    if constexpr (V == 4) {
        b16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = (__bf16)v[i];
        *reinterpret_cast<b16x4*>(p) = h;
    } else {
        p[0] = __builtin_bit_cast(unsigned short, (__bf16)v[0]);
    }
}
template <int V>
__device__ __forceinline__ void ld_bf16(const unsigned short* __restrict__ p, float (&v)[V]) {
    if constexpr (V == 4) {
        const b16x4 h = *reinterpret_cast<const b16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)h[i];
    } else {
        v[0] = (float)__builtin_bit_cast(__bf16, p[0]);
    }
}

// y of a BatchNorm pass: fp32, or (YB: one-plane bf16 arithmetic, HN_Y_BF16) bf16 elements behind the same pointer
template <int V, bool YB>
__device__ __forceinline__ void ld_y(const float* __restrict__ y, int64_t off, float (&v)[V]) {
    if constexpr (YB) ld_bf16<V>(reinterpret_cast<const unsigned short*>(y) + off, v);
    else ldv<V>(y + off, v);
}

struct RowLaunch {
    dim3 grid, block;
    int V;
};

// Decide vector width and block shape.  `ptrs`/`lds` list every strided operand.
RowLaunch plan_rows(int64_t npix, int C, std::initializer_list<const void*> ptrs, std::initializer_list<int> lds,
                    int target_blocks = 2048) {
    bool v4 = (C % 4) == 0;
    for (const void* p : ptrs) v4 = v4 && (p == nullptr || hn_aligned16(p));
    for (int ld : lds) v4 = v4 && (ld % 4 == 0);
    RowLaunch r;
    r.V = v4 ? 4 : 1;
    const int pieces = C / r.V;
    int bx = 1;
    while (bx < pieces && bx < 64) bx <<= 1;
    const int by = 256 / bx;
    const int gx = (pieces + bx - 1) / bx;
    int64_t gy = (npix + by - 1) / by;
    const int64_t cap = (target_blocks + gx - 1) / gx;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    r.grid = dim3(gx, (unsigned)gy);
    r.block = dim3(bx, by);
    return r;
}

// du = dz * act'(u)
template <int V>
__device__ __forceinline__ void act_grad(const float (&dz)[V], const float (&z)[V], const float (&y)[V],
                                         const float (&sc)[V], const float (&sh)[V], int act, float slope,
                                         bool zmask, float (&du)[V], float& dslope) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if (act == HN_ACT_RELU) {
            // mask from the saved output, or (no residual, z not kept) from u recomputed exactly as the forward did
            const float m = zmask ? z[i] : y[i] * sc[i] + sh[i];
            du[i] = m > 0.f ? dz[i] : 0.f;
        } else if (act == HN_ACT_PRELU || act == HN_ACT_LEAKY) {
            const float u = y[i] * sc[i] + sh[i];
            du[i] = u > 0.f ? dz[i] : dz[i] * slope;
            dslope += u > 0.f ? 0.f : dz[i] * u;
        } else {
            du[i] = dz[i];
        }
    }
}

// plane-tensor operands of the BatchNorm passes (see st_planes): `slot` is READ (the scale the planes are / were made with)
struct PlaneOut {
    unsigned short* hi;   // NULL: no plane output
    long long plane;
    int ld;
    const float* slot;
};
struct PlaneIn {
    const unsigned short* hi;  // NULL: the fp32 operand is used
    long long plane;
    int ld;
    const float* slot;
};

// ZP: z is (also) written as a plane tensor with the scale of zp.slot (a bound known before the pass, hn_bn_finalize_range);
// the fp32 z is then optional.  RP: the residual is a plane tensor.
template <int V, bool ZP = false, bool RP = false, bool YB = false>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ y, int ldy,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ res, int ldr, float* __restrict__ z,
                                                        int ldz, int64_t npix, int C, int act, const float* slope_p,
                                                        float leaky, const float* __restrict__ mask, int64_t hw,
                                                        float* __restrict__ z_scale, unsigned char* __restrict__ rbits,
                                                        PlaneOut zp = PlaneOut{nullptr, 0, 0, nullptr},
                                                        PlaneIn rp = PlaneIn{nullptr, 0, 0, nullptr}) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    unsigned amax = 0u;  // largest |z| written by this thread (f16x2 scale of the output, optional)
    float zs = 1.f, zi = 1.f, rs = 1.f, ri = 1.f;
    if constexpr (ZP) { if (zp.slot != nullptr) hn_h2_scales_of_slot(zp.slot, zs, zi); }
    if constexpr (RP) { if (rp.slot != nullptr) hn_h2_scales_of_slot(rp.slot, rs, ri); }
    if (c < C) {
        float sc[V], sh[V];
        ldv<V>(scale + c, sc);
        ldv<V>(shift + c, sh);
        const float slope = act == HN_ACT_PRELU ? *slope_p : leaky;
        for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
            float v[V], r[V], o[V], m[V];
            ld_y<V, YB>(y, pix * ldy + c, v);
            if constexpr (RP) {
                if (rp.slot != nullptr) ld_planes<V>(rp.hi + pix * rp.ld + c, rp.plane, r, ri);
                else ld_bf16<V>(rp.hi + pix * rp.ld + c, r);
            } else if (res != nullptr) ldv<V>(res + pix * ldr + c, r);
            if (mask != nullptr) ldv<V>(mask + (pix / hw) * C + c, m);  // Dropout2d keep/(1-p) factor of (image, channel)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float u = v[i] * sc[i] + sh[i];
                if (RP || res != nullptr) u += r[i];
                o[i] = hn_apply_act(u, act, slope);
                if (mask != nullptr) o[i] *= m[i];
                amax = max(amax, __float_as_uint(o[i]) & 0x7fffffffu);
            }
            if constexpr (ZP) {
                if (zp.slot != nullptr) st_planes<V>(zp.hi + pix * zp.ld + c, zp.plane, o, zs);
                else st_bf16<V>(zp.hi + pix * zp.ld + c, o);
                if (z != nullptr) stv<V>(z + pix * ldz + c, o);
            } else {
                stv<V>(z + pix * ldz + c, o);
            }
            if constexpr (V == 4) {
                // one byte per 16-B piece: bit i = (z[c + i] > 0), what the backward passes need of z (a 16th of its bytes)
                if (rbits != nullptr)
                    rbits[pix * (C / 4) + c / 4] = (unsigned char)((o[0] > 0.f) | ((o[1] > 0.f) << 1) | ((o[2] > 0.f) << 2) | ((o[3] > 0.f) << 3));
            }
        }
    }
    if (z_scale != nullptr) hn_absmax_publish(amax, z_scale);
}

// Bound of |act(y * scale + shift [+ r]) * m| over a channel from the range [ymin, ymax] of y, max |r| and max m: the
// affine map and the activations are monotone, so the extremes sit at the ends of the range (exact without a residual).
__device__ __forceinline__ float bn_out_bound(float ymax, float ymin, float sc, float sh, int act, float slope, float rmax,
                                              float mmax) {
    const float a = ymax * sc + sh, b = ymin * sc + sh;
    const float hi = fmaxf(a, b) + rmax, lo = fminf(a, b) - rmax;
    float bound;
    if (act == HN_ACT_RELU) bound = fmaxf(hi, 0.f);
    else if (act == HN_ACT_PRELU || act == HN_ACT_LEAKY) bound = fmaxf(fabsf(hn_apply_act(hi, act, slope)), fabsf(hn_apply_act(lo, act, slope)));
    else bound = fmaxf(fabsf(hi), fabsf(lo));
    return bound * mmax;
}

// largest magnitude a scale slot holds
__device__ __forceinline__ float slot_max(const float* slot) {
    const int lane = (threadIdx.y * blockDim.x + threadIdx.x) & (HN_SCALE_WORDS - 1);
    unsigned m = reinterpret_cast<const unsigned*>(slot)[lane * HN_SCALE_STRIDE];
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)m));
}

// sums[0:C] += sum du ; sums[C:2C] += sum du*xhat ; sums[2C] += dslope   (double accumulation throughout)
// ZMASK / MASK are compile-time so that the pixel loop is straight-line code the compiler can unroll and pipeline
// (loads of the next pixels issued before the arithmetic of the current ones)
template <int V, bool ZMASK, bool MASK, bool YB = false>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dz, int lddz,
                                                               const float* __restrict__ z, int ldz,
                                                               const float* __restrict__ y, int ldy,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int64_t npix, int C, int act, const float* slope_p, float leaky,
                                                               double* __restrict__ sums, const float* __restrict__ mask,
                                                               int64_t hw, const unsigned char* __restrict__ rbits,
                                                               unsigned* __restrict__ gmax = nullptr) {
    __shared__ double red[2 * 256 * 4 + 256];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    float sc[V], sh[V], mu[V], is[V];
    double s1[V], s2[V];
    double dslope = 0.0;
    unsigned gm[V];  // largest |du| of the channel seen by this thread (bound of dy, optional)
#pragma unroll
    for (int i = 0; i < V; ++i) { sc[i] = sh[i] = mu[i] = is[i] = 0.f; s1[i] = s2[i] = 0.0; gm[i] = 0u; }
    if (live) {
        ldv<V>(scale + c, sc);
        ldv<V>(shift + c, sh);
        ldv<V>(mean + c, mu);
        ldv<V>(invstd + c, is);
    }
    const float slope = act == HN_ACT_PRELU ? *slope_p : leaky;
    if (live) {
        const int64_t first = (int64_t)blockIdx.y * blockDim.y + threadIdx.y, step = (int64_t)gridDim.y * blockDim.y;
        const float* pg = dz + first * lddz + c;
        int64_t oy = first * ldy + c;
        const float* pz = (ZMASK && z != nullptr) ? z + first * ldz + c : nullptr;
        const int64_t sg = step * lddz, sy = step * ldy, sz = step * ldz;
#pragma unroll 4
        for (int64_t pix = first; pix < npix; pix += step) {
            float g[V], zz[V], yy[V], du[V];
            float ds = 0.f;
            ldv<V>(pg, g);
            ld_y<V, YB>(y, oy, yy);
            if constexpr (ZMASK) {
                if (V == 4 && rbits != nullptr) {  // the ReLU mask as one byte per 16-B piece instead of z itself
                    const unsigned b = rbits[pix * (C / 4) + c / 4];
#pragma unroll
                    for (int i = 0; i < V; ++i) zz[i] = (b >> i) & 1u ? 1.f : 0.f;
                } else {
                    ldv<V>(pz, zz);
                    pz += sz;
                }
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) zz[i] = 0.f;
            }
            pg += sg;
            oy += sy;
            if constexpr (MASK) {
                float m[V];
                ldv<V>(mask + (pix / hw) * C + c, m);
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] *= m[i];
            }
            act_grad<V>(g, zz, yy, sc, sh, act, slope, ZMASK, du, ds);
            dslope += (double)ds;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                s1[i] += (double)du[i];
                s2[i] += (double)(du[i] * ((yy[i] - mu[i]) * is[i]));
                gm[i] = max(gm[i], __float_as_uint(du[i]) & 0x7fffffffu);
            }
        }
    }
    // reduce over threadIdx.y
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
    double* r1 = red;
    double* r2 = red + 256 * 4;
    double* r3 = red + 2 * 256 * 4;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        r1[(ty * bx + tx) * V + i] = s1[i];
        r2[(ty * bx + tx) * V + i] = s2[i];
    }
    r3[ty * bx + tx] = dslope;
    __syncthreads();
    double* rep = sums + (size_t)(blockIdx.y % HN_STAT_REPLICAS) * (2 * C + 1);
    if (ty == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < by; ++k) {
                a += r1[(k * bx + tx) * V + i];
                b += r2[(k * bx + tx) * V + i];
            }
            unsafeAtomicAdd(rep + c + i, a);
            unsafeAtomicAdd(rep + C + c + i, b);
        }
    }
    if (gmax != nullptr) {  // block maximum per channel through LDS (re-using the reduction buffer), one atomic per channel
        __syncthreads();
        unsigned* rg = reinterpret_cast<unsigned*>(red);
#pragma unroll
        for (int i = 0; i < V; ++i) rg[(ty * bx + tx) * V + i] = gm[i];
        __syncthreads();
        if (ty == 0 && live) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                unsigned m = 0u;
                for (int k = 0; k < by; ++k) m = max(m, rg[(k * bx + tx) * V + i]);
                if (m != 0u) atomicMax(gmax + (size_t)(blockIdx.y % HN_STAT_REPLICAS) * C + c + i, m);
            }
        }
    }
    if (act == HN_ACT_PRELU) {
        if (ty == 0 && tx == 0) {
            double a = 0.0;
            for (int k = 0; k < bx * by; ++k) a += r3[k];
            unsafeAtomicAdd(rep + 2 * C, a);
        }
    }
}

// out[i] = (float) sum over replicas of in[r][i], and the double workspace is zeroed again so that the next
// reduction can accumulate into it
// (bn backward: n = 2C+1 = dbeta | dgamma | dslope; the three optional pointers receive the same values where the
// caller wants them -- the parameters' gradient slots -- so that no copy / accumulation kernel follows)
// `gmax` (optional, with dy_slot): per-channel max |du| replicas from the reduction kernel (consumed and re-zeroed here) ->
// bound of |dy| = |gamma invstd| (max|du| + |dbeta| / count + max|xhat| |dgamma| / count) over the channels, published into
// dy_slot BEFORE the apply pass writes dy as a plane tensor with that scale.
__global__ __launch_bounds__(256) void bn_bwd_finish_range_kernel(
    double* __restrict__ in, float* __restrict__ out, int n, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dslope, int accumulate, unsigned* __restrict__ gmax, const float* __restrict__ gamma,
    const float* __restrict__ invstd, const float* __restrict__ xhat_max, float inv_count, float* __restrict__ dy_slot) {
    // 16 lanes per channel (one per replica); channel index C stands for the PReLU slope sum (entry 2C)
    static_assert(HN_STAT_REPLICAS == 16, "lane layout");
    const int r = threadIdx.x & 15, c = blockIdx.x * 16 + (threadIdx.x >> 4);
    double a = 0.0, b = 0.0;
    unsigned g = 0u;
    if (c < C) {
        double* pi = in + (size_t)r * n + c;
        unsigned* pg = gmax + (size_t)r * C + c;
        a = pi[0]; b = pi[C]; g = pg[0];
        pi[0] = 0.0; pi[C] = 0.0; pg[0] = 0u;
    } else if (c == C) {
        double* pi = in + (size_t)r * n + 2 * C;
        a = pi[0];
        pi[0] = 0.0;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        g = max(g, (unsigned)__shfl_xor((int)g, o));
    }
    unsigned bits = 0u;
    if (r == 0 && c <= C) {
        const float va = (float)a, vb = (float)b;
        if (c < C) {
            out[c] = va;
            out[C + c] = vb;
            if (dbeta != nullptr) dbeta[c] = accumulate ? dbeta[c] + va : va;
            if (dgamma != nullptr) dgamma[c] = accumulate ? dgamma[c] + vb : vb;
            const float bound = fabsf(gamma[c] * invstd[c]) * (__uint_as_float(g) + fabsf(va) * inv_count + xhat_max[c] * fabsf(vb) * inv_count);
            bits = __float_as_uint(bound) & 0x7fffffffu;
        } else {
            out[2 * C] = va;
            if (dslope != nullptr) *dslope = accumulate ? *dslope + va : va;
        }
    }
    hn_absmax_publish(bits, dy_slot);
}

__global__ void f64_to_f32_reset_kernel(double* __restrict__ in, float* __restrict__ out, int n, int C,
                                        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dslope,
                                        int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < HN_STAT_REPLICAS; ++r) {
            a += in[(size_t)r * n + i];
            in[(size_t)r * n + i] = 0.0;
        }
        const float v = (float)a;
        out[i] = v;
        float* dst = i < C ? (dbeta != nullptr ? dbeta + i : nullptr)
                           : (i < 2 * C ? (dgamma != nullptr ? dgamma + (i - C) : nullptr) : dslope);
        if (dst != nullptr) *dst = accumulate ? *dst + v : v;
    }
}

template <int V, bool DP = false, bool YB = false>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ dz, int lddz,
                                                              const float* __restrict__ z, int ldz,
                                                              const float* __restrict__ y, int ldy,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ dparams,
                                                              float inv_count, int training, float* __restrict__ dy, int lddy,
                                                              float* __restrict__ dres, int lddr, int64_t npix, int C, int act,
                                                              const float* slope_p, float leaky, const float* __restrict__ mask,
                                                              int64_t hw, float* __restrict__ dy_scale,
                                                              const unsigned char* __restrict__ rbits,
                                                              PlaneOut dp = PlaneOut{nullptr, 0, 0, nullptr}) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    unsigned amax = 0u;  // largest |dy| written by this thread (f16x2 scale of the output, optional)
    float ds = 1.f, di = 1.f;
    if constexpr (DP) { if (dp.slot != nullptr) hn_h2_scales_of_slot(dp.slot, ds, di); }
    if (c < C) {
    float sc[V], sh[V], mu[V], is[V], ga[V], m1[V], m2[V];
    ldv<V>(scale + c, sc);
    ldv<V>(shift + c, sh);
    ldv<V>(mean + c, mu);
    ldv<V>(invstd + c, is);
    ldv<V>(gamma + c, ga);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        m1[i] = training ? dparams[c + i] * inv_count : 0.f;
        m2[i] = training ? dparams[C + c + i] * inv_count : 0.f;
        ga[i] *= is[i];
    }
    const float slope = act == HN_ACT_PRELU ? *slope_p : leaky;
    const bool zmask = act == HN_ACT_RELU && (z != nullptr || (V == 4 && rbits != nullptr));
    float unused = 0.f;
#pragma unroll 2
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
        float g[V], zz[V], yy[V], du[V], o[V];
        ldv<V>(dz + pix * lddz + c, g);
        ld_y<V, YB>(y, pix * ldy + c, yy);
        if (mask != nullptr) {
            float m[V];
            ldv<V>(mask + (pix / hw) * C + c, m);
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] *= m[i];
        }
        if (zmask) {
            if (V == 4 && rbits != nullptr) {
                const unsigned b = rbits[pix * (C / 4) + c / 4];
#pragma unroll
                for (int i = 0; i < V; ++i) zz[i] = (b >> i) & 1u ? 1.f : 0.f;
            } else {
                ldv<V>(z + pix * ldz + c, zz);
            }
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) zz[i] = 0.f;
        }
        act_grad<V>(g, zz, yy, sc, sh, act, slope, zmask, du, unused);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            o[i] = ga[i] * (du[i] - m1[i] - (yy[i] - mu[i]) * is[i] * m2[i]);
            amax = max(amax, __float_as_uint(o[i]) & 0x7fffffffu);
        }
        if constexpr (DP) {
            if (dp.slot != nullptr) st_planes<V>(dp.hi + pix * dp.ld + c, dp.plane, o, ds);
            else st_bf16<V>(dp.hi + pix * dp.ld + c, o);
        } else {
            stv<V>(dy + pix * lddy + c, o);
        }
        if (dres != nullptr) stv<V>(dres + pix * lddr + c, du);
    }
    }
    if (dy_scale != nullptr) hn_absmax_publish(amax, dy_scale);
}

__global__ void bn_finalize_kernel(double* __restrict__ stats, double inv_count, double unbias,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < HN_STAT_REPLICAS; ++r) {
        s1 += stats[(size_t)r * 2 * C + c];
        s2 += stats[(size_t)r * 2 * C + C + c];
        // consumed: leave the accumulators zeroed for the next convolution epilogue that adds into them
        stats[(size_t)r * 2 * C + c] = 0.0;
        stats[(size_t)r * 2 * C + C + c] = 0.0;
    }
    const double m = s1 * inv_count;
    double var = s2 * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float mf = (float)m;
    mean[c] = mf;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mf * sc;
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
    if (running_var != nullptr) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
}

// hn_bn_finalize + the bound of the normalised tensor: `keys` are the range keys the convolution epilogue accumulated next to
// the statistics (per channel max y and max -y, replicated; consumed and re-zeroed here); the bound of
// |act(bn(y) [+ residual]) * mask| over all channels is published into z_slot and max |xhat| per channel is kept for the
// backward bound.
__global__ __launch_bounds__(256) void bn_finalize_range_kernel(
    double* __restrict__ stats, unsigned* __restrict__ keys, double inv_count, double unbias, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ xhat_max, int C, int act,
    const float* slope_p, float leaky, const float* __restrict__ res_slot, float mask_max, float* __restrict__ z_slot) {
    // 16 lanes per channel, one per replica of the accumulators: one memory round trip, then a 4-step cross-lane reduction
    static_assert(HN_STAT_REPLICAS == 16, "lane layout");
    const int r = threadIdx.x & 15, c = blockIdx.x * 16 + (threadIdx.x >> 4);
    const float rmax = res_slot != nullptr ? slot_max(res_slot) : 0.f;
    double s1 = 0.0, s2 = 0.0;
    unsigned k0 = 0u, k1 = 0u;
    if (c < C) {
        double* ps = stats + (size_t)r * 2 * C + c;
        unsigned* pk = keys + (size_t)r * 2 * C + c;
        s1 = ps[0]; s2 = ps[C]; k0 = pk[0]; k1 = pk[C];
        ps[0] = 0.0; ps[C] = 0.0; pk[0] = 0u; pk[C] = 0u;  // consumed: zero for the next convolution epilogue
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
        k0 = max(k0, (unsigned)__shfl_xor((int)k0, o));
        k1 = max(k1, (unsigned)__shfl_xor((int)k1, o));
    }
    unsigned bits = 0u;
    if (c < C && r == 0) {
        const double m = s1 * inv_count;
        double var = s2 * inv_count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float mf = (float)m;
        mean[c] = mf;
        invstd[c] = is;
        const float sc = gamma[c] * is;
        const float sh = beta[c] - mf * sc;
        scale[c] = sc;
        shift[c] = sh;
        if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        if (running_var != nullptr) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
        // (keys of zero: the channel saw no value -- an empty launch; treat as the single value 0)
        const float ymax = k0 != 0u ? hn_range_unkey(k0) : 0.f, ymin = k1 != 0u ? -hn_range_unkey(k1) : 0.f;
        xhat_max[c] = fmaxf(fabsf(ymax - mf), fabsf(ymin - mf)) * is;
        const float slope = act == HN_ACT_PRELU ? *slope_p : leaky;
        bits = __float_as_uint(bn_out_bound(ymax, ymin, sc, sh, act, slope, rmax, mask_max)) & 0x7fffffffu;
    }
    hn_absmax_publish(bits, z_slot);
}

__global__ void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                      float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    mean[c] = rm[c];
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// out[c] += block partial (summed in double inside the block, one float atomic per block and channel)
template <int V>
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int ld, int64_t npix, int C,
                                                         float* __restrict__ out) {
    __shared__ double red[256 * 4];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    double s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.0;
    if (live) {
        for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix;
             pix += (int64_t)gridDim.y * blockDim.y) {
            float v[V];
            ldv<V>(x + pix * ld + c, v);
#pragma unroll
            for (int i = 0; i < V; ++i) s[i] += (double)v[i];
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < V; ++i) red[(ty * bx + tx) * V + i] = s[i];
    __syncthreads();
    if (ty == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            double a = 0.0;
            for (int k = 0; k < by; ++k) a += red[(k * bx + tx) * V + i];
            atomicAdd(out + c + i, (float)a);
        }
    }
}

// stats[0:C] += sum x ; stats[C:2C] += sum x*x in double (batch statistics when the producer was not our conv)
template <int V>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, int ld, int64_t npix, int C,
                                                           double* __restrict__ out) {
    __shared__ double red[2 * 256 * 4];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    double s[V], q[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = q[i] = 0.0;
    if (live) {
        for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix;
             pix += (int64_t)gridDim.y * blockDim.y) {
            float v[V];
            ldv<V>(x + pix * ld + c, v);
#pragma unroll
            for (int i = 0; i < V; ++i) { s[i] += (double)v[i]; q[i] += (double)v[i] * (double)v[i]; }
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        red[(ty * bx + tx) * V + i] = s[i];
        red[256 * 4 + (ty * bx + tx) * V + i] = q[i];
    }
    __syncthreads();
    if (ty == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < by; ++k) {
                a += red[(k * bx + tx) * V + i];
                b += red[256 * 4 + (k * bx + tx) * V + i];
            }
            double* rep = out + (size_t)(blockIdx.y % HN_STAT_REPLICAS) * 2 * C;
            unsafeAtomicAdd(rep + c + i, a);
            unsafeAtomicAdd(rep + C + c + i, b);
        }
    }
}

// mode 0: dx = dy * act'(out) ; 1: y = a + b ; 2: y = a
template <int V, int MODE>
__global__ __launch_bounds__(256) void rowwise_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                     int ldb, float* __restrict__ o, int ldo, int64_t npix, int C, int act,
                                                     float slope) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
        float va[V], vb[V], vo[V];
        ldv<V>(a + pix * lda + c, va);
        if constexpr (MODE != 2) ldv<V>(b + pix * ldb + c, vb);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if constexpr (MODE == 0) {
                const float k = vb[i] > 0.f ? 1.f : (act == HN_ACT_RELU ? 0.f : (act == HN_ACT_NONE ? 1.f : slope));
                vo[i] = va[i] * k;
            } else if constexpr (MODE == 1) {
                vo[i] = va[i] + vb[i];
            } else {
                vo[i] = va[i];
            }
        }
        stv<V>(o + pix * ldo + c, vo);
    }
}

// dx = dy * act'(out) AND dbias[c] += sum over pixels of dx (the bias gradient of a convolution with a fused activation: the
// critics' conv + LeakyReLU layers, cm/discriminator_model.py:40-58) in one pass over dy -- a separate column-sum pass cost as
// much as this one again (and 46 us of latency on the critics' small tensors).
template <int V>
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ out,
                                                          int ldo, float* __restrict__ dx, int lddx, int64_t npix, int C, int act,
                                                          float slope, float* __restrict__ dbias, float* __restrict__ dx_scale) {
    __shared__ float red[256 * 4];
    unsigned amax = 0u;  // largest |dx| written by this thread (f16x2 scale of the tensor the conv backward kernels read)
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    if (live) {
        for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
            float g[V], o[V], r[V];
            ldv<V>(dy + pix * lddy + c, g);
            ldv<V>(out + pix * ldo + c, o);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float k = o[i] > 0.f ? 1.f : (act == HN_ACT_RELU ? 0.f : (act == HN_ACT_NONE ? 1.f : slope));
                r[i] = g[i] * k;
                s[i] += r[i];
                amax = max(amax, __float_as_uint(r[i]) & 0x7fffffffu);
            }
            stv<V>(dx + pix * lddx + c, r);
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < V; ++i) red[(ty * bx + tx) * V + i] = s[i];
    __syncthreads();
    if (ty == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float a = 0.f;
            for (int k = 0; k < by; ++k) a += red[(k * bx + tx) * V + i];
            atomicAdd(dbias + c + i, a);
        }
    }
    if (dx_scale != nullptr) hn_absmax_publish(amax, dx_scale);
}

template <int V>
__global__ __launch_bounds__(256) void scale_nc_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mask,
                                                      float* __restrict__ y, int ldy, int64_t npix, int64_t hw, int C) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
        const int64_t n = pix / hw;
        float v[V], m[V];
        ldv<V>(x + pix * ldx + c, v);
        ldv<V>(mask + n * C + c, m);
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] *= m[i];
        stv<V>(y + pix * ldy + c, v);
    }
}

// fp32 tensor -> plane tensor with the scale of `slot` (tests, and tensors whose producer cannot write planes itself)
template <int V>
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ slot,
                                                       unsigned short* __restrict__ planes, long long plane, int ldp,
                                                       int64_t npix, int C) {
    float s = 1.f, inv = 1.f;
    if (slot != nullptr) hn_h2_scales_of_slot(slot, s, inv);
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
        float v[V];
        ldv<V>(x + pix * ldx + c, v);
        if (slot != nullptr) st_planes<V>(planes + pix * ldp + c, plane, v, s);
        else st_bf16<V>(planes + pix * ldp + c, v);
    }
}
template <int V>
__global__ __launch_bounds__(256) void from_planes_kernel(const unsigned short* __restrict__ planes, long long plane, int ldp,
                                                         const float* __restrict__ slot, float* __restrict__ x, int ldx,
                                                         int64_t npix, int C) {
    float s = 1.f, inv = 1.f;
    if (slot != nullptr) hn_h2_scales_of_slot(slot, s, inv);
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
        float v[V];
        if (slot != nullptr) ld_planes<V>(planes + pix * ldp + c, plane, v, inv);
        else ld_bf16<V>(planes + pix * ldp + c, v);
        stv<V>(x + pix * ldx + c, v);
    }
}

// f16x2 range census of one operand tensor (fp32 with its slot, or a plane tensor): out[0] += elements that are not zero,
// out[1] += those of them below 2^-18 of the slot's magnitude bound, i.e. |t * s| < 2^-3 -- their `lo` term is an fp16
// subnormal and they carry fewer than 22 significant bits (include/heatnet_hip.h).
template <int V, bool PL>
__global__ __launch_bounds__(256) void range_census_kernel(const float* __restrict__ x, const unsigned short* __restrict__ hi,
                                                          long long plane, int ld, const float* __restrict__ slot, int64_t npix,
                                                          int C, unsigned long long* __restrict__ out) {
    float s, inv;
    hn_h2_scales_of_slot(slot, s, inv);
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    unsigned nz = 0u, low = 0u;
    if (c < C) {
        for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < npix; pix += (int64_t)gridDim.y * blockDim.y) {
            float v[V];
            if constexpr (PL) ld_planes<V>(hi + pix * ld + c, plane, v, 1.f);  // (hi + lo: the scaled value)
            else {
                ldv<V>(x + pix * ld + c, v);
#pragma unroll
                for (int i = 0; i < V; ++i) v[i] *= s;
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float a = fabsf(v[i]);
                nz += a > 0.f;
                low += a > 0.f && a < 0.125f;
            }
        }
    }
    __shared__ unsigned red[2][4];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) {
        nz += __shfl_xor(nz, o);
        low += __shfl_xor(low, o);
    }
    if ((t & 63) == 0) { red[0][t >> 6] = nz; red[1][t >> 6] = low; }
    __syncthreads();
    if (t == 0) {
        atomicAdd(out, (unsigned long long)red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(out + 1, (unsigned long long)red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

bool bad_rows(int64_t npix, int C, std::initializer_list<int> lds) {
    if (npix <= 0 || C <= 0) return true;
    for (int ld : lds)
        if (ld < C) return true;
    return false;
}

}  // namespace

#define HN_LAUNCH_V(plan, kern, ...)                                                                     \
    do {                                                                                                 \
        if ((plan).V == 4) HN_LAUNCH((kern<4>), (plan).grid, (plan).block, 0, s, __VA_ARGS__);  \
        else HN_LAUNCH((kern<1>), (plan).grid, (plan).block, 0, s, __VA_ARGS__);                \
    } while (0)

extern "C" int hn_bn_finalize(double* stats, int64_t count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* mean,
                              float* invstd, float* scale, float* shift, int C, void* stream) {
    if (!stats || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0 || count <= 0) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    HN_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, 1.0 / (double)count,
                       unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                 float* shift, int C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !mean || !invstd || !scale || !shift || C <= 0)
        return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HN_LAUNCH(bn_eval_params_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, running_mean,
                       running_var, eps, mean, invstd, scale, shift, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bn_act_fwd(const float* y, int ldy, const float* scale, const float* shift, const float* residual,
                             int ldr, float* z, int ldz, int64_t npix, int C, int act, const float* slope,
                             float leaky_slope, const float* mask, int64_t hw, float* z_scale, unsigned char* relu_bits,
                             void* stream) {
    if (!y || !scale || !shift || !z || act < 0 || act > 3 || (act == HN_ACT_PRELU && !slope)) return HN_EINVAL;
    if (mask != nullptr && (hw <= 0 || npix % hw != 0)) return HN_EINVAL;
    if (bad_rows(npix, C, {ldy, ldz}) || (residual && ldr < C)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows(npix, C, {y, z, residual, scale, shift, mask}, {ldy, ldz, residual ? ldr : 4});
    if (relu_bits != nullptr && pl.V != 4) return HN_EALIGN;  // the byte mask exists for whole 16-B pieces only
    HN_LAUNCH_V(pl, bn_act_fwd_kernel, y, ldy, scale, shift, residual, ldr, z, ldz, npix, C, act, slope, leaky_slope, mask,
                hw, z_scale, relu_bits);
    HN_CHECK_LAUNCH();
    return HN_OK;
}


namespace {
// vector width a plane operand allows (8-byte accesses of four fp16 values)
bool planes_v4(const void* hi, long long plane, int ld) {
    return hi == nullptr || ((reinterpret_cast<uintptr_t>(hi) & 7u) == 0 && plane % 4 == 0 && ld % 4 == 0);
}
const void* const kUnaligned = reinterpret_cast<const void*>(uintptr_t(1));  // forces plan_rows to the scalar plan
}  // namespace

extern "C" int hn_bn_finalize_range(double* stats, uint32_t* range_keys, int64_t count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                    float* invstd, float* scale, float* shift, float* xhat_max, int C, int act,
                                    const float* slope, float leaky_slope, const float* res_slot, float mask_max, float* z_slot,
                                    void* stream) {
    if (!stats || !range_keys || !gamma || !beta || !mean || !invstd || !scale || !shift || !xhat_max || !z_slot || C <= 0 ||
        count <= 0 || act < 0 || act > 3 || (act == HN_ACT_PRELU && !slope) || !(mask_max > 0.f))
        return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    HN_LAUNCH(bn_finalize_range_kernel, dim3((C + 15) / 16), dim3(256), 0, s, stats, range_keys, 1.0 / (double)count, unbias, gamma,
              beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, xhat_max, C, act, slope, leaky_slope,
              res_slot, mask_max, z_slot);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bn_act_fwd_planes(const float* y, int ldy, const float* scale, const float* shift, const float* residual, int ldr,
                                    const void* res_planes, int64_t res_plane_elems, int ldrp, const float* res_slot, float* z,
                                    int ldz, void* z_planes, int64_t z_plane_elems, int ldzp, const float* z_slot, int64_t npix,
                                    int C, int act, const float* slope, float leaky_slope, const float* mask, int64_t hw,
                                    unsigned char* relu_bits, void* stream) {
    const bool yb = (act & HN_Y_BF16) != 0;  // y holds bf16 elements (one-plane bf16 arithmetic)
    act &= ~HN_Y_BF16;
    if (!y || !scale || !shift || !z_planes || act < 0 || act > 3 || (act == HN_ACT_PRELU && !slope)) return HN_EINVAL;  // (z_slot NULL: bf16)
    if (yb && z_slot != nullptr) return HN_EINVAL;
    if (residual && res_planes) return HN_EINVAL;
    if (res_planes && (res_plane_elems < (npix - 1) * ldrp + C || ldrp < C)) return HN_EINVAL;
    if (mask != nullptr && (hw <= 0 || npix % hw != 0)) return HN_EINVAL;
    if (bad_rows(npix, C, {ldy, ldzp}) || (z && ldz < C) || (residual && ldr < C) || z_plane_elems < (npix - 1) * ldzp + C)
        return HN_EINVAL;
    if ((residual || res_planes) && (act == HN_ACT_PRELU || act == HN_ACT_LEAKY)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = planes_v4(z_planes, z_plane_elems, ldzp) && planes_v4(res_planes, res_plane_elems, ldrp);
    const RowLaunch pl = plan_rows(npix, C, {y, z, residual, scale, shift, mask, v4 ? nullptr : kUnaligned}, {ldy, z ? ldz : 4, residual ? ldr : 4});
    if ((relu_bits != nullptr || yb) && pl.V != 4) return HN_EALIGN;
    const PlaneOut zp{static_cast<unsigned short*>(z_planes), (long long)z_plane_elems, ldzp, z_slot};
    const PlaneIn rp{static_cast<const unsigned short*>(res_planes), (long long)res_plane_elems, ldrp, res_slot};
    if (yb) {
        if (res_planes)
            HN_LAUNCH((bn_act_fwd_kernel<4, true, true, true>), pl.grid, pl.block, 0, s, y, ldy, scale, shift, residual, ldr, z, ldz,
                      npix, C, act, slope, leaky_slope, mask, hw, (float*)nullptr, relu_bits, zp, rp);
        else
            HN_LAUNCH((bn_act_fwd_kernel<4, true, false, true>), pl.grid, pl.block, 0, s, y, ldy, scale, shift, residual, ldr, z, ldz,
                      npix, C, act, slope, leaky_slope, mask, hw, (float*)nullptr, relu_bits, zp, rp);
        HN_CHECK_LAUNCH();
        return HN_OK;
    }
#define HN_LAUNCH_FWDP(RPF)                                                                                                    \
    do {                                                                                                                        \
        if (pl.V == 4)                                                                                                          \
            HN_LAUNCH((bn_act_fwd_kernel<4, true, RPF>), pl.grid, pl.block, 0, s, y, ldy, scale, shift, residual, ldr, z, ldz, npix, \
                      C, act, slope, leaky_slope, mask, hw, (float*)nullptr, relu_bits, zp, rp);                                \
        else                                                                                                                    \
            HN_LAUNCH((bn_act_fwd_kernel<1, true, RPF>), pl.grid, pl.block, 0, s, y, ldy, scale, shift, residual, ldr, z, ldz, npix, \
                      C, act, slope, leaky_slope, mask, hw, (float*)nullptr, relu_bits, zp, rp);                                \
    } while (0)
    if (res_planes) HN_LAUNCH_FWDP(true);
    else HN_LAUNCH_FWDP(false);
#undef HN_LAUNCH_FWDP
    HN_CHECK_LAUNCH();
    return HN_OK;
}

namespace {
int bn_bwd_reduce_impl(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                    const float* mean, const float* invstd, const float* scale, const float* shift,
                                    int64_t npix, int C, int act, const float* slope, float leaky_slope, double* sums,
                                    float* dparams, float* dgamma_out, float* dbeta_out, float* dslope_out,
                                    int out_accumulate, const float* mask, int64_t hw, const unsigned char* relu_bits,
                                    void* stream, unsigned* gmax = nullptr, const float* gamma = nullptr,
                                    const float* xhat_max = nullptr, float inv_count = 0.f, float* dy_slot = nullptr) {
    const bool yb = (act & HN_Y_BF16) != 0;
    act &= ~HN_Y_BF16;
    if (!dz || !y || !mean || !invstd || !scale || !shift || !sums || !dparams || act < 0 || act > 3) return HN_EINVAL;
    if (mask != nullptr && (hw <= 0 || npix % hw != 0)) return HN_EINVAL;
    if (act == HN_ACT_PRELU && !slope) return HN_EINVAL;
    if (yb && gmax != nullptr) return HN_EINVAL;
    if (bad_rows(npix, C, {lddz, ldy}) || (z && ldz < C)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool zmask = act == HN_ACT_RELU && (z != nullptr || relu_bits != nullptr);
    const bool zread = zmask && relu_bits == nullptr;  // z itself is read only when there is no byte mask
    RowLaunch pl = plan_rows(npix, C, {dz, y, zread ? z : nullptr, mean, invstd, scale, shift},
                             {lddz, ldy, zread ? ldz : 4}, 2048);
    if ((relu_bits != nullptr || yb) && pl.V != 4) return HN_EALIGN;
    {
        // every block ends with 2C double atomics: give each thread at least MINPIX pixels so that small tensors are not
        // dominated by them (a [8x40x88, 256] tensor went from 2048 blocks / 1 M atomics to 440 / 225 k)
        // (measured inside the training step, rocprofv3: tensors under ~48 MB 36.7 -> 27.0 us at 8 pixels per thread,
        // the large residual-block tensors 75.2 -> 61.6 us at 32)
        static const int minpix_env = getenv("HN_BN_MINPIX") ? atoi(getenv("HN_BN_MINPIX")) : 0;
        const int minpix = minpix_env > 0 ? minpix_env : ((int64_t)npix * C * 4 >= (48ll << 20) ? 32 : 8);
        const int64_t gy_cap = (npix + (int64_t)pl.block.y * minpix - 1) / ((int64_t)pl.block.y * minpix);
        if ((int64_t)pl.grid.y > gy_cap) pl.grid.y = (unsigned)(gy_cap < 1 ? 1 : gy_cap);
    }
#define HN_LAUNCH_REDUCE(ZM, MK)                                                                                       \
    do {                                                                                                                \
        if (yb)                                                                                                         \
            HN_LAUNCH((bn_act_bwd_reduce_kernel<4, ZM, MK, true>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean,  \
                      invstd, scale, shift, npix, C, act, slope, leaky_slope, sums, mask, hw, relu_bits, gmax);         \
        else if (pl.V == 4)                                                                                             \
            HN_LAUNCH((bn_act_bwd_reduce_kernel<4, ZM, MK>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean, invstd, \
                      scale, shift, npix, C, act, slope, leaky_slope, sums, mask, hw, relu_bits, gmax);                 \
        else                                                                                                            \
            HN_LAUNCH((bn_act_bwd_reduce_kernel<1, ZM, MK>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean, invstd, \
                      scale, shift, npix, C, act, slope, leaky_slope, sums, mask, hw, relu_bits, gmax);                 \
    } while (0)
    if (zmask) {
        if (mask != nullptr) HN_LAUNCH_REDUCE(true, true);
        else HN_LAUNCH_REDUCE(true, false);
    } else {
        if (mask != nullptr) HN_LAUNCH_REDUCE(false, true);
        else HN_LAUNCH_REDUCE(false, false);
    }
#undef HN_LAUNCH_REDUCE
    HN_CHECK_LAUNCH();
    if (gmax != nullptr)
        HN_LAUNCH(bn_bwd_finish_range_kernel, dim3((C + 1 + 15) / 16), dim3(256), 0, s, sums, dparams, 2 * C + 1, C, dgamma_out,
                  dbeta_out, dslope_out, out_accumulate, gmax, gamma, invstd, xhat_max, inv_count, dy_slot);
    else
        HN_LAUNCH(f64_to_f32_reset_kernel, dim3((2 * C + 1 + 255) / 256), dim3(256), 0, s, sums, dparams, 2 * C + 1, C, dgamma_out,
                  dbeta_out, dslope_out, out_accumulate);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
}  // namespace

extern "C" int hn_bn_act_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                    const float* mean, const float* invstd, const float* scale, const float* shift,
                                    int64_t npix, int C, int act, const float* slope, float leaky_slope, double* sums,
                                    float* dparams, float* dgamma_out, float* dbeta_out, float* dslope_out,
                                    int out_accumulate, const float* mask, int64_t hw, const unsigned char* relu_bits,
                                    void* stream) {
    return bn_bwd_reduce_impl(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, npix, C, act, slope, leaky_slope, sums, dparams,
                              dgamma_out, dbeta_out, dslope_out, out_accumulate, mask, hw, relu_bits, stream);
}

extern "C" int hn_bn_act_bwd_reduce_range(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                          const float* mean, const float* invstd, const float* scale, const float* shift,
                                          int64_t npix, int C, int act, const float* slope, float leaky_slope, double* sums,
                                          float* dparams, float* dgamma_out, float* dbeta_out, float* dslope_out,
                                          int out_accumulate, const float* mask, int64_t hw, const unsigned char* relu_bits,
                                          uint32_t* du_max, const float* gamma, const float* xhat_max, int64_t count, int training,
                                          float* dy_slot, void* stream) {
    if (!du_max || !gamma || !xhat_max || !dy_slot || count <= 0) return HN_EINVAL;
    return bn_bwd_reduce_impl(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, npix, C, act, slope, leaky_slope, sums, dparams,
                              dgamma_out, dbeta_out, dslope_out, out_accumulate, mask, hw, relu_bits, stream, du_max, gamma, xhat_max,
                              training ? (float)(1.0 / (double)count) : 0.f, dy_slot);
}

extern "C" int hn_bn_act_bwd_finish(double* sums, float* dparams, int C, float* dgamma_out, float* dbeta_out, float* dslope_out,
                                    int out_accumulate, uint32_t* du_max, const float* gamma, const float* invstd,
                                    const float* xhat_max, int64_t count, int training, float* dy_slot, void* stream) {
    if (!sums || !dparams || C <= 0) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (du_max != nullptr) {
        if (!gamma || !invstd || !xhat_max || !dy_slot || count <= 0) return HN_EINVAL;
        const float inv_count = training ? (float)(1.0 / (double)count) : 0.f;
        HN_LAUNCH(bn_bwd_finish_range_kernel, dim3((C + 1 + 15) / 16), dim3(256), 0, s, sums, dparams, 2 * C + 1, C, dgamma_out,
                  dbeta_out, dslope_out, out_accumulate, du_max, gamma, invstd, xhat_max, inv_count, dy_slot);
    } else {
        HN_LAUNCH(f64_to_f32_reset_kernel, dim3((2 * C + 1 + 255) / 256), dim3(256), 0, s, sums, dparams, 2 * C + 1, C, dgamma_out,
                  dbeta_out, dslope_out, out_accumulate);
    }
    HN_CHECK_LAUNCH();
    return HN_OK;
}

namespace {
int bn_bwd_apply_impl(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                   const float* mean, const float* invstd, const float* gamma, const float* scale,
                                   const float* shift, const float* dparams, int64_t count, int training, float* dy,
                                   int lddy, float* dres, int lddr, int64_t npix, int C, int act, const float* slope,
                                   float leaky_slope, const float* mask, int64_t hw, float* dy_scale,
                                   const unsigned char* relu_bits, void* stream, PlaneOut dp = PlaneOut{nullptr, 0, 0, nullptr}) {
    const bool yb = (act & HN_Y_BF16) != 0;
    act &= ~HN_Y_BF16;
    if (!dz || !y || !mean || !invstd || !gamma || !scale || !shift || (!dy && !dp.hi) || act < 0 || act > 3) return HN_EINVAL;
    if (yb && (dp.hi == nullptr || dp.slot != nullptr)) return HN_EINVAL;
    if (dp.hi != nullptr) { dy = nullptr; lddy = C; }
    if (mask != nullptr && (hw <= 0 || npix % hw != 0)) return HN_EINVAL;
    if ((training && (!dparams || count <= 0)) || (act == HN_ACT_PRELU && !slope)) return HN_EINVAL;
    if (bad_rows(npix, C, {lddz, ldy, lddy}) || (dres && lddr < C) || (z && ldz < C)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool zmask = act == HN_ACT_RELU && z != nullptr && relu_bits == nullptr;  // (z itself is read)
    const RowLaunch pl =
        plan_rows(npix, C, {dz, y, dy, dres, zmask ? z : nullptr, mean, invstd, gamma, scale, shift,
                            planes_v4(dp.hi, dp.plane, dp.ld) ? nullptr : kUnaligned},
                  {lddz, ldy, lddy, dres ? lddr : 4, zmask ? ldz : 4});
    if ((relu_bits != nullptr || yb) && pl.V != 4) return HN_EALIGN;
    const float inv_count = training ? (float)(1.0 / (double)count) : 0.f;
    if (dp.hi != nullptr) {
        if (yb)
            HN_LAUNCH((bn_act_bwd_apply_kernel<4, true, true>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean, invstd, gamma,
                      scale, shift, dparams, inv_count, training, dy, lddy, dres, lddr, npix, C, act, slope, leaky_slope, mask, hw,
                      (float*)nullptr, relu_bits, dp);
        else if (pl.V == 4)
            HN_LAUNCH((bn_act_bwd_apply_kernel<4, true>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale,
                      shift, dparams, inv_count, training, dy, lddy, dres, lddr, npix, C, act, slope, leaky_slope, mask, hw,
                      (float*)nullptr, relu_bits, dp);
        else
            HN_LAUNCH((bn_act_bwd_apply_kernel<1, true>), pl.grid, pl.block, 0, s, dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale,
                      shift, dparams, inv_count, training, dy, lddy, dres, lddr, npix, C, act, slope, leaky_slope, mask, hw,
                      (float*)nullptr, relu_bits, dp);
    } else {
        HN_LAUNCH_V(pl, bn_act_bwd_apply_kernel, dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale, shift, dparams, inv_count,
                    training, dy, lddy, dres, lddr, npix, C, act, slope, leaky_slope, mask, hw, dy_scale, relu_bits);
    }
    HN_CHECK_LAUNCH();
    return HN_OK;
}
}  // namespace

extern "C" int hn_bn_act_bwd_apply(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                   const float* mean, const float* invstd, const float* gamma, const float* scale,
                                   const float* shift, const float* dparams, int64_t count, int training, float* dy,
                                   int lddy, float* dres, int lddr, int64_t npix, int C, int act, const float* slope,
                                   float leaky_slope, const float* mask, int64_t hw, float* dy_scale,
                                   const unsigned char* relu_bits, void* stream) {
    return bn_bwd_apply_impl(dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale, shift, dparams, count, training, dy, lddy, dres,
                             lddr, npix, C, act, slope, leaky_slope, mask, hw, dy_scale, relu_bits, stream);
}

extern "C" int hn_bn_act_bwd_apply_planes(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                          const float* mean, const float* invstd, const float* gamma, const float* scale,
                                          const float* shift, const float* dparams, int64_t count, int training, void* dy_planes,
                                          int64_t dy_plane_elems, int lddp, const float* dy_slot, float* dres, int lddr,
                                          int64_t npix, int C, int act, const float* slope, float leaky_slope, const float* mask,
                                          int64_t hw, const unsigned char* relu_bits, void* stream) {
    if (!dy_planes || lddp < C || dy_plane_elems < (npix - 1) * lddp + C) return HN_EINVAL;  // (dy_slot NULL: bf16)
    return bn_bwd_apply_impl(dz, lddz, z, ldz, y, ldy, mean, invstd, gamma, scale, shift, dparams, count, training, nullptr, C, dres,
                             lddr, npix, C, act, slope, leaky_slope, mask, hw, nullptr, relu_bits, stream,
                             PlaneOut{static_cast<unsigned short*>(dy_planes), (long long)dy_plane_elems, lddp, dy_slot});
}

extern "C" int hn_channel_sum(const float* x, int ld, int64_t npix, int C, float* out, void* stream) {
    if (!x || !out || bad_rows(npix, C, {ld})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    const RowLaunch pl = plan_rows(npix, C, {x}, {ld}, 1024);
    HN_LAUNCH_V(pl, channel_sum_kernel, x, ld, npix, C, out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_channel_stats(const float* x, int ld, int64_t npix, int C, double* stats, void* stream) {
    if (!x || !stats || bad_rows(npix, C, {ld})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows(npix, C, {x}, {ld}, 2048);
    HN_LAUNCH_V(pl, channel_stats_kernel, x, ld, npix, C, stats);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_act_bwd(const float* dy, int lddy, const float* out, int ldo, float* dx, int lddx, int64_t npix, int C,
                          int act, float leaky_slope, void* stream) {
    if (!dy || !out || !dx || act < 0 || act > 3 || bad_rows(npix, C, {lddy, ldo, lddx})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows(npix, C, {dy, out, dx}, {lddy, ldo, lddx});
    if (pl.V == 4) HN_LAUNCH((rowwise_kernel<4, 0>), pl.grid, pl.block, 0, s, dy, lddy, out, ldo, dx, lddx, npix, C, act, leaky_slope);
    else HN_LAUNCH((rowwise_kernel<1, 0>), pl.grid, pl.block, 0, s, dy, lddy, out, ldo, dx, lddx, npix, C, act, leaky_slope);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_act_bwd_bias(const float* dy, int lddy, const float* out, int ldo, float* dx, int lddx, int64_t npix, int C,
                               int act, float leaky_slope, float* dbias, float* dx_scale, void* stream) {
    if (!dy || !out || !dx || !dbias || act < 0 || act > 3 || bad_rows(npix, C, {lddy, ldo, lddx})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(dbias, 0, (size_t)C * sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    const RowLaunch pl = plan_rows(npix, C, {dy, out, dx}, {lddy, ldo, lddx}, 512);
    HN_LAUNCH_V(pl, act_bwd_bias_kernel, dy, lddy, out, ldo, dx, lddx, npix, C, act, leaky_slope, dbias, dx_scale);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t npix, int C,
                      void* stream) {
    if (!a || !b || !y || bad_rows(npix, C, {lda, ldb, ldy})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows(npix, C, {a, b, y}, {lda, ldb, ldy});
    if (pl.V == 4) HN_LAUNCH((rowwise_kernel<4, 1>), pl.grid, pl.block, 0, s, a, lda, b, ldb, y, ldy, npix, C, 0, 0.f);
    else HN_LAUNCH((rowwise_kernel<1, 1>), pl.grid, pl.block, 0, s, a, lda, b, ldb, y, ldy, npix, C, 0, 0.f);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

namespace {
// nn.Dropout2d's per-(image, channel) factors: 0 with probability p, else 1/(1-p).  Counter-based generator (two rounds
// of a 64-bit mix of seed and element index): the mask of a step is a pure function of (seed, index).
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, int n, float p, float keep_scale,
                                                          unsigned long long seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0, 1)
    mask[i] = u < p ? 0.f : keep_scale;
}

// logical NCHW tensor stored channel-planar (torch's default layout) -> NHWC: tiles of 32 pixels x C channels through LDS
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t hw,
                                                          int ldy) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {  // r = channel inside the tile, tx = pixel
        const int c = c0 + r;
        const int64_t p = p0 + tx;
        tile[r][tx] = (c < C && p < hw) ? x[((int64_t)n * C + c) * hw + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {  // r = pixel, tx = channel
        const int c = c0 + tx;
        const int64_t p = p0 + r;
        if (c < C && p < hw) y[((int64_t)n * hw + p) * ldy + c] = tile[tx][r];
    }
}
}  // namespace

extern "C" int hn_dropout_mask(float* mask, int n, float p, uint64_t seed, void* stream) {
    if (!mask || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return HN_EINVAL;
    HN_LAUNCH(dropout_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), mask, n, p,
              1.0f / (1.0f - p), (unsigned long long)seed);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_nchw_to_nhwc(const float* x, float* y, int N, int C, int64_t hw, int ldy, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || hw <= 0 || ldy < C) return HN_EINVAL;
    if ((hw + 31) / 32 > 0x7fffffffll || N > 65535 || (C + 31) / 32 > 65535) return HN_ESHAPE;
    HN_LAUNCH(nchw_to_nhwc_kernel, dim3((unsigned)((hw + 31) / 32), (C + 31) / 32, N), dim3(256), 0,
              static_cast<hipStream_t>(stream), x, y, C, hw, ldy);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_copy(const float* x, int ldx, float* y, int ldy, int64_t npix, int C, void* stream) {
    if (!x || !y || bad_rows(npix, C, {ldx, ldy})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows(npix, C, {x, y}, {ldx, ldy});
    if (pl.V == 4) HN_LAUNCH((rowwise_kernel<4, 2>), pl.grid, pl.block, 0, s, x, ldx, nullptr, 0, y, ldy, npix, C, 0, 0.f);
    else HN_LAUNCH((rowwise_kernel<1, 2>), pl.grid, pl.block, 0, s, x, ldx, nullptr, 0, y, ldy, npix, C, 0, 0.f);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_tensor_to_planes(const float* x, int ldx, int64_t npix, int C, const float* scale, void* planes,
                                   int64_t plane_elems, int ldp, void* stream) {
    if (!x || !planes || bad_rows(npix, C, {ldx, ldp}) || plane_elems < (npix - 1) * ldp + C) return HN_EINVAL;  // (scale NULL: bf16)
    hipStream_t s = static_cast<hipStream_t>(stream);
    RowLaunch pl = plan_rows(npix, C, {x}, {ldx});
    if (pl.V == 4 && (ldp % 4 != 0 || plane_elems % 4 != 0 || (reinterpret_cast<uintptr_t>(planes) & 7u) != 0)) {
        pl = plan_rows(npix, C, {reinterpret_cast<const void*>(uintptr_t(1))}, {ldx});
    }
    unsigned short* ph = static_cast<unsigned short*>(planes);
    HN_LAUNCH_V(pl, to_planes_kernel, x, ldx, scale, ph, (long long)plane_elems, ldp, npix, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_planes_to_tensor(const void* planes, int64_t plane_elems, int ldp, const float* scale, float* x, int ldx,
                                   int64_t npix, int C, void* stream) {
    if (!x || !planes || bad_rows(npix, C, {ldx, ldp}) || plane_elems < (npix - 1) * ldp + C) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    RowLaunch pl = plan_rows(npix, C, {x}, {ldx});
    if (pl.V == 4 && (ldp % 4 != 0 || plane_elems % 4 != 0 || (reinterpret_cast<uintptr_t>(planes) & 7u) != 0))
        pl = plan_rows(npix, C, {reinterpret_cast<const void*>(uintptr_t(1))}, {ldx});
    const unsigned short* ph = static_cast<const unsigned short*>(planes);
    HN_LAUNCH_V(pl, from_planes_kernel, ph, (long long)plane_elems, ldp, scale, x, ldx, npix, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_range_census(const float* x, const void* planes, int64_t plane_elems, int ld, const float* scale, int64_t npix,
                               int C, uint64_t* out, void* stream) {
    if ((x == nullptr) == (planes == nullptr) || !scale || !out || bad_rows(npix, C, {ld})) return HN_EINVAL;
    if (planes && plane_elems < (npix - 1) * ld + C) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned short* ph = static_cast<const unsigned short*>(planes);
    unsigned long long* o = reinterpret_cast<unsigned long long*>(out);
    if (planes) {
        const bool v4 = (reinterpret_cast<uintptr_t>(planes) & 7u) == 0 && plane_elems % 4 == 0 && ld % 4 == 0;
        const RowLaunch pl = plan_rows(npix, C, {v4 ? nullptr : reinterpret_cast<const void*>(uintptr_t(1))}, {4}, 1024);
        if (pl.V == 4) HN_LAUNCH((range_census_kernel<4, true>), pl.grid, pl.block, 0, s, x, ph, (long long)plane_elems, ld, scale, npix, C, o);
        else HN_LAUNCH((range_census_kernel<1, true>), pl.grid, pl.block, 0, s, x, ph, (long long)plane_elems, ld, scale, npix, C, o);
    } else {
        const RowLaunch pl = plan_rows(npix, C, {x}, {ld}, 1024);
        if (pl.V == 4) HN_LAUNCH((range_census_kernel<4, false>), pl.grid, pl.block, 0, s, x, ph, 0ll, ld, scale, npix, C, o);
        else HN_LAUNCH((range_census_kernel<1, false>), pl.grid, pl.block, 0, s, x, ph, 0ll, ld, scale, npix, C, o);
    }
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_scale_nc(const float* x, int ldx, const float* mask, float* y, int ldy, int N, int64_t hw, int C,
                           void* stream) {
    if (!x || !mask || !y || N <= 0 || hw <= 0 || bad_rows((int64_t)N * hw, C, {ldx, ldy})) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)N * hw;
    const RowLaunch pl = plan_rows(npix, C, {x, y, mask}, {ldx, ldy});
    HN_LAUNCH_V(pl, scale_nc_kernel, x, ldx, mask, y, ldy, npix, hw, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
