// HBM-bound spatial operators on NHWC (ptr, ld) tensors: 3x3/s2 max pooling, adaptive average pooling
// (pyramid pooling), bilinear resampling with align_corners=False.  All backward kernels are written in
// GATHER form (one thread owns one input-gradient element and sums every output that touched it), so
// they need neither atomics nor a zero-fill and are bitwise reproducible.
// Thread layout as in norm_act.hip: threadIdx.x walks channels in 16-B pieces, threadIdx.y walks pixels.
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

struct RowLaunch {
    dim3 grid, block;
    int V;
};

RowLaunch plan_rows(int64_t npix, int C, std::initializer_list<const void*> ptrs, std::initializer_list<int> lds) {
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
    const int64_t cap = (4096 + gx - 1) / gx;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    r.grid = dim3(gx, (unsigned)gy);
    r.block = dim3(bx, by);
    return r;
}

#define HN_PIX_LOOP(npix) \
    for (int64_t pix = (int64_t)blockIdx.y * blockDim.y + threadIdx.y; pix < (npix); pix += (int64_t)gridDim.y * blockDim.y)

// ---- max pooling 3x3 stride 2 pad 1 ---------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                         uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * Ho * Wo;
    HN_PIX_LOOP(npix) {
        const int wo = (int)(pix % Wo);
        const int64_t t = pix / Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        float best[V];
        int bi[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { best[i] = -INFINITY; bi[i] = 0; }
        bool first = true;
        for (int kh = 0; kh < 3; ++kh) {
            const int h = 2 * ho - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int w = 2 * wo - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[V];
                ldv<V>(x + ((int64_t)(n * H + h) * W + w) * ldx + c, v);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    if (first || v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; bi[i] = kh * 3 + kw; }
                }
                first = false;
            }
        }
        stv<V>(y + pix * ldy + c, best);
#pragma unroll
        for (int i = 0; i < V; ++i) idx[pix * C + c + i] = (uint8_t)bi[i];
    }
}

template <int V>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
                                                         float* __restrict__ dx, int lddx, int N, int H, int W, int C, int Ho, int Wo) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * H * W;
    HN_PIX_LOOP(npix) {
        const int w = (int)(pix % W);
        const int64_t t = pix / W;
        const int h = (int)(t % H), n = (int)(t / H);
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        const int ho_lo = h / 2, ho_hi = min(Ho - 1, (h + 1) / 2);
        const int wo_lo = w / 2, wo_hi = min(Wo - 1, (w + 1) / 2);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int k = (h - 2 * ho + 1) * 3 + (w - 2 * wo + 1);
                const int64_t op = (int64_t)(n * Ho + ho) * Wo + wo;
                float g[V];
                ldv<V>(dy + op * lddy + c, g);
#pragma unroll
                for (int i = 0; i < V; ++i)
                    if (idx[op * C + c + i] == k) acc[i] += g[i];
            }
        stv<V>(dx + pix * lddx + c, acc);
    }
}

// ---- adaptive average pooling ----------------------------------------------------------------------------
__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

template <int V>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                         int N, int H, int W, int C, int Ho, int Wo) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * Ho * Wo;
    HN_PIX_LOOP(npix) {
        const int j = (int)(pix % Wo);
        const int64_t t = pix / Wo;
        const int i = (int)(t % Ho), n = (int)(t / Ho);
        const int h0 = bin_start(i, H, Ho), h1 = bin_end(i, H, Ho), w0 = bin_start(j, W, Wo), w1 = bin_end(j, W, Wo);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                float v[V];
                ldv<V>(x + ((int64_t)(n * H + h) * W + w) * ldx + c, v);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v[e];
            }
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= inv;
        stv<V>(y + pix * ldy + c, acc);
    }
}

// Large bins (pyramid pooling: up to the whole 40x88 map per output pixel): one block per output pixel and channel
// slab, threadIdx.y strides over the bin's pixels, LDS reduction across threadIdx.y.
template <int V>
__global__ __launch_bounds__(256) void avgpool_fwd_bin_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                             int N, int H, int W, int C, int Ho, int Wo) {
    __shared__ float red[256 * 4];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    const int pix = blockIdx.y;
    const int j = pix % Wo, t = pix / Wo, i = t % Ho, n = t / Ho;
    const int h0 = bin_start(i, H, Ho), h1 = bin_end(i, H, Ho), w0 = bin_start(j, W, Wo), w1 = bin_end(j, W, Wo);
    const int bw = w1 - w0, cnt = (h1 - h0) * bw;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (live) {
        for (int k = threadIdx.y; k < cnt; k += blockDim.y) {
            const int h = h0 + k / bw, w = w0 + k % bw;
            float v[V];
            ldv<V>(x + ((int64_t)(n * H + h) * W + w) * ldx + c, v);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += v[e];
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int e = 0; e < V; ++e) red[(ty * bx + tx) * V + e] = acc[e];
    __syncthreads();
    if (ty == 0 && live) {
        const float inv = 1.f / (float)cnt;
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float a = 0.f;
            for (int k = 0; k < by; ++k) a += red[(k * bx + tx) * V + e];
            o[e] = a * inv;
        }
        stv<V>(y + (int64_t)pix * ldy + c, o);
    }
}

template <int V>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                         int N, int H, int W, int C, int Ho, int Wo) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * H * W;
    HN_PIX_LOOP(npix) {
        const int w = (int)(pix % W);
        const int64_t t = pix / W;
        const int h = (int)(t % H), n = (int)(t / H);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int i = 0; i < Ho; ++i) {
            const int h0 = bin_start(i, H, Ho), h1 = bin_end(i, H, Ho);
            if (h < h0 || h >= h1) continue;
            for (int j = 0; j < Wo; ++j) {
                const int w0 = bin_start(j, W, Wo), w1 = bin_end(j, W, Wo);
                if (w < w0 || w >= w1) continue;
                const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                float g[V];
                ldv<V>(dy + ((int64_t)(n * Ho + i) * Wo + j) * lddy + c, g);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += g[e] * inv;
            }
        }
        stv<V>(dx + pix * lddx + c, acc);
    }
}

// ---- bilinear, align_corners = False ----------------------------------------------------------------------
// PyTorch: src = scale * (dst + 0.5) - 0.5, clamped below at 0; i0 = floor(src); i1 = i0 + (i0 < in-1);
// l1 = src - i0; l0 = 1 - l1.
__device__ __forceinline__ void bilinear_src(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

template <int V>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                          int N, int H, int W, int C, int Ho, int Wo, float sh, float sw) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * Ho * Wo;
    HN_PIX_LOOP(npix) {
        const int wo = (int)(pix % Wo);
        const int64_t t = pix / Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bilinear_src(ho, sh, H, h0, h1, lh0, lh1);
        bilinear_src(wo, sw, W, w0, w1, lw0, lw1);
        const int64_t base = (int64_t)n * H * W;
        float a[V], b[V], d[V], e[V], o[V];
        ldv<V>(x + (base + (int64_t)h0 * W + w0) * ldx + c, a);
        ldv<V>(x + (base + (int64_t)h0 * W + w1) * ldx + c, b);
        ldv<V>(x + (base + (int64_t)h1 * W + w0) * ldx + c, d);
        ldv<V>(x + (base + (int64_t)h1 * W + w1) * ldx + c, e);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = lh0 * (lw0 * a[i] + lw1 * b[i]) + lh1 * (lw0 * d[i] + lw1 * e[i]);
        stv<V>(y + pix * ldy + c, o);
    }
}

// weight with which output index o (of `out`) reads input index i
__device__ __forceinline__ float bilinear_weight(int o, int i, float scale, int in) {
    int i0, i1;
    float l0, l1;
    bilinear_src(o, scale, in, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                          int N, int H, int W, int C, int Ho, int Wo, float sh, float sw) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * H * W;
    HN_PIX_LOOP(npix) {
        const int w = (int)(pix % W);
        const int64_t t = pix / W;
        const int h = (int)(t % H), n = (int)(t / H);
        // outputs whose source coordinate lies in (h-1, h+1); generous by one on both sides, weights decide
        int ho_lo = (int)floorf(((float)h - 0.5f) / sh - 0.5f) - 1, ho_hi = (int)ceilf(((float)h + 1.5f) / sh - 0.5f) + 1;
        int wo_lo = (int)floorf(((float)w - 0.5f) / sw - 0.5f) - 1, wo_hi = (int)ceilf(((float)w + 1.5f) / sw - 0.5f) + 1;
        ho_lo = max(ho_lo, 0); ho_hi = min(ho_hi, Ho - 1);
        wo_lo = max(wo_lo, 0); wo_hi = min(wo_hi, Wo - 1);
        if (h == 0) ho_lo = 0;
        if (w == 0) wo_lo = 0;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            const float wh = bilinear_weight(ho, h, sh, H);
            if (wh == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const float ww = bilinear_weight(wo, w, sw, W);
                if (ww == 0.f) continue;
                float g[V];
                ldv<V>(dy + ((int64_t)(n * Ho + ho) * Wo + wo) * lddy + c, g);
                const float k = wh * ww;
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += k * g[i];
            }
        }
        stv<V>(dx + pix * lddx + c, acc);
    }
}

// ---- exact x2 upsampling (the decoder's F.upsample before every PSPUpsample conv, cm/models/pspnet.py:36) ----------
// An INPUT pixel's 3x3 neighbourhood produces its 2x2 output pixels (2h..2h+1, 2w..2w+1) with the generic weights
// (bilinear_src: the border clamps come out identically).
typedef _Float16 h16x4s __attribute__((ext_vector_type(4)));

// PL: the output is written as a plane tensor (hi / lo fp16 images, include/heatnet_hip.h) with the scale of `slot` -- the
// input's slot: bilinear weights are convex, so the input's magnitude bounds the output's.
template <int V, bool PL = false>
__global__ __launch_bounds__(256) void bilinear_up2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                              int N, int H, int W, int C, unsigned short* __restrict__ yh = nullptr,
                                                              long long yplane = 0, const float* __restrict__ slot = nullptr) {
    float ps = 1.f, pi = 1.f;
    if constexpr (PL) { if (slot != nullptr) hn_h2_scales_of_slot(slot, ps, pi); }
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    // a thread-iteration owns a 2 x 2 block of input pixels: its 4 x 4 (clamped) neighbourhood is read once -- 16 loads for
    // sixteen outputs where the one-pixel form read 9 per four; every output is the same expression of the same operands
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const int64_t nblk = (int64_t)N * Hb * Wb;
    const int Wo = 2 * W;
    HN_PIX_LOOP(nblk) {
        const int wb = (int)(pix % Wb);
        const int64_t t = pix / Wb;
        const int hb = (int)(t % Hb), n = (int)(t / Hb);
        const int hq = 2 * hb, wq = 2 * wb;
        float g[4][4][V];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int hr = min(max(hq - 1 + a, 0), H - 1), wc = min(max(wq - 1 + b, 0), W - 1);
                ldv<V>(x + ((int64_t)(n * H + hr) * W + wc) * ldx + c, g[a][b]);
            }
#pragma unroll
        for (int pi_ = 0; pi_ < 2; ++pi_)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
        const int h = hq + pi_, w = wq + pj;
        if (h >= H || w >= W) continue;
        // output row 2h+dh reads neighbourhood rows (dh, dh+1) = input rows (h-1, h) / (h, h+1); at the borders the clamped
        // row repeats and bilinear_src's weights (1, 0 at the top / left edge) make the sum the same value as the generic
        // kernel's
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            int h0, h1;
            float lh0, lh1;
            bilinear_src(2 * h + dh, 0.5f, H, h0, h1, lh0, lh1);
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                int w0, w1;
                float lw0, lw1, o[V];
                bilinear_src(2 * w + dw, 0.5f, W, w0, w1, lw0, lw1);
#pragma unroll
                for (int i = 0; i < V; ++i)
                    o[i] = lh0 * (lw0 * g[pi_ + dh][pj + dw][i] + lw1 * g[pi_ + dh][pj + dw + 1][i]) +
                           lh1 * (lw0 * g[pi_ + dh + 1][pj + dw][i] + lw1 * g[pi_ + dh + 1][pj + dw + 1][i]);
                const int64_t opix = ((int64_t)(n * 2 * H + 2 * h + dh) * Wo + 2 * w + dw);
                if constexpr (PL) {
                    unsigned short* dst = yh + opix * ldy + c;
                    if (slot == nullptr) {  // one bf16 plane
#pragma unroll
                        for (int i = 0; i < V; ++i) dst[i] = __builtin_bit_cast(unsigned short, (__bf16)o[i]);
                    } else if constexpr (V == 4) {
                        h16x4s hh, ll;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float xs = o[i] * ps;
                            hh[i] = (_Float16)xs;
                            ll[i] = (_Float16)(xs - (float)hh[i]);
                        }
                        *reinterpret_cast<h16x4s*>(dst) = hh;
                        *reinterpret_cast<h16x4s*>(dst + yplane) = ll;
                    } else {
                        const float xs = o[0] * ps;
                        const _Float16 hh = (_Float16)xs;
                        dst[0] = __builtin_bit_cast(unsigned short, hh);
                        dst[yplane] = __builtin_bit_cast(unsigned short, (_Float16)(xs - (float)hh));
                    }
                } else {
                    stv<V>(y + opix * ldy + c, o);
                }
            }
        }
        }
    }
}

// gradient of the exact x2 upsampling: input pixel (h, w) is read by output rows 2h-1 .. 2h+2 and the same four columns.  A
// thread-iteration owns a 2 x 2 block of INPUT pixels: their 4 x 4 windows overlap in a 6 x 6 patch of dy, read once (36 loads
// for four results instead of 64 -- the one-pixel form moved four times the tensor through the L1 and ran at half the HBM rate),
// row by row: the two horizontal sums of a row, then their share in the two rows of results.  Gather form, fixed order.
template <int V>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                              int lddx, int N, int H, int W, int C) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const int64_t nblk = (int64_t)N * Hb * Wb;
    const int Ho = 2 * H, Wo = 2 * W;
    HN_PIX_LOOP(nblk) {
        const int wb = (int)(pix % Wb);
        const int64_t t = pix / Wb;
        const int hb = (int)(t % Hb), n = (int)(t / Hb);
        const int h0 = 2 * hb, w0 = 2 * wb;
        float ww[2][6];  // weight of dy column 2 w0 - 1 + b in result column w0 + j (zero: not a source, outside, no such result)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int wo = 2 * w0 - 1 + k;
                ww[j][k] = (wo >= 0 && wo < Wo && w0 + j < W) ? bilinear_weight(wo, w0 + j, 0.5f, W) : 0.f;
            }
        float acc[2][2][V];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[i][j][e] = 0.f;
        // a real loop over the six rows (six loads in flight per thread): unrolled, the 36 loads are hoisted together, take 144
        // registers and the occupancy an HBM-bound kernel lives on
#pragma unroll 1
        for (int a = 0; a < 6; ++a) {
            const int hu = 2 * h0 - 1 + a;
            const bool rin = hu >= 0 && hu < Ho;
            const float wh0 = rin ? bilinear_weight(hu, h0, 0.5f, H) : 0.f;
            const float wh1 = (rin && h0 + 1 < H) ? bilinear_weight(hu, h0 + 1, 0.5f, H) : 0.f;
            const int ho = min(max(hu, 0), Ho - 1);
            float rs[2][V];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) rs[j][e] = 0.f;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int wo = min(max(2 * w0 - 1 + b, 0), Wo - 1);
                float g[V];
                ldv<V>(dy + ((int64_t)(n * Ho + ho) * Wo + wo) * lddy + c, g);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < V; ++e) rs[j][e] += ww[j][b] * g[e];  // (zero for clamped / foreign columns)
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    acc[0][j][e] += wh0 * rs[j][e];
                    acc[1][j][e] += wh1 * rs[j][e];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (h0 + i < H && w0 + j < W) stv<V>(dx + ((int64_t)(n * H + h0 + i) * W + w0 + j) * lddx + c, acc[i][j]);
    }
}

// Strong upsampling (pyramid priors 1x1..6x6 -> 40x88, critic maps x32): every input pixel gathers from hundreds of
// outputs, so one block owns one input pixel (and channel slab) and threadIdx.y strides over the candidate window.
template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_pix_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                              int lddx, int N, int H, int W, int C, int Ho, int Wo, float sh,
                                                              float sw) {
    __shared__ float red[256 * 4];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    const int pix = blockIdx.y;
    const int w = pix % W, t = pix / W, h = t % H, n = t / H;
    int ho_lo = (int)floorf(((float)h - 0.5f) / sh - 0.5f) - 1, ho_hi = (int)ceilf(((float)h + 1.5f) / sh - 0.5f) + 1;
    int wo_lo = (int)floorf(((float)w - 0.5f) / sw - 0.5f) - 1, wo_hi = (int)ceilf(((float)w + 1.5f) / sw - 0.5f) + 1;
    ho_lo = max(ho_lo, 0); ho_hi = min(ho_hi, Ho - 1);
    wo_lo = max(wo_lo, 0); wo_hi = min(wo_hi, Wo - 1);
    if (h == 0) ho_lo = 0;
    if (w == 0) wo_lo = 0;
    const int bw = wo_hi - wo_lo + 1, cnt = (ho_hi - ho_lo + 1) * bw;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    if (live) {
        for (int k = threadIdx.y; k < cnt; k += blockDim.y) {
            const int ho = ho_lo + k / bw, wo = wo_lo + k % bw;
            const float kw = bilinear_weight(ho, h, sh, H) * bilinear_weight(wo, w, sw, W);
            if (kw == 0.f) continue;
            float g[V];
            ldv<V>(dy + ((int64_t)(n * Ho + ho) * Wo + wo) * lddy + c, g);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += kw * g[i];
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < V; ++i) red[(ty * bx + tx) * V + i] = acc[i];
    __syncthreads();
    if (ty == 0 && live) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float a = 0.f;
            for (int k = 0; k < by; ++k) a += red[(k * bx + tx) * V + i];
            o[i] = a;
        }
        stv<V>(dx + (int64_t)pix * lddx + c, o);
    }
}

// one block per pixel of the small tensor: bx walks channel pieces (<= 16 so that >= 16 threads share the window)
RowLaunch plan_pixel_blocks(int64_t npix, int C, std::initializer_list<const void*> ptrs, std::initializer_list<int> lds) {
    bool v4 = (C % 4) == 0;
    for (const void* p : ptrs) v4 = v4 && (p == nullptr || hn_aligned16(p));
    for (int ld : lds) v4 = v4 && (ld % 4 == 0);
    RowLaunch r;
    r.V = v4 ? 4 : 1;
    const int pieces = C / r.V;
    int bx = 1;
    while (bx < pieces && bx < 16) bx <<= 1;
    r.grid = dim3((pieces + bx - 1) / bx, (unsigned)npix);
    r.block = dim3(bx, 256 / bx);
    return r;
}

// out = act(main + sum_i bilinear_up(prior_i)): the merge step of the pyramid-pooling head once its 1x1 bottleneck
// has been applied per pyramid level at the pooled resolution (a 1x1 convolution commutes with bilinear resampling).
struct PyramidArgs {
    const float* p[8];
    int h[8], w[8], ld[8];
    int count;
};

template <int V>
__global__ __launch_bounds__(256) void pyramid_merge_kernel(const float* __restrict__ mainp, int ldm, PyramidArgs a,
                                                           float* __restrict__ out, int ldo, int N, int H, int W, int C,
                                                           int relu) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * H * W;
    HN_PIX_LOOP(npix) {
        const int wo = (int)(pix % W);
        const int64_t t = pix / W;
        const int ho = (int)(t % H), n = (int)(t / H);
        float acc[V];
        ldv<V>(mainp + pix * ldm + c, acc);
        for (int k = 0; k < a.count; ++k) {
            const int ph = a.h[k], pw = a.w[k], pl = a.ld[k];
            int h0, h1, w0, w1;
            float lh0, lh1, lw0, lw1;
            bilinear_src(ho, (float)ph / (float)H, ph, h0, h1, lh0, lh1);
            bilinear_src(wo, (float)pw / (float)W, pw, w0, w1, lw0, lw1);
            const float* base = a.p[k] + (int64_t)n * ph * pw * pl + c;
            float q0[V], q1[V], q2[V], q3[V];
            ldv<V>(base + ((int64_t)h0 * pw + w0) * pl, q0);
            ldv<V>(base + ((int64_t)h0 * pw + w1) * pl, q1);
            ldv<V>(base + ((int64_t)h1 * pw + w0) * pl, q2);
            ldv<V>(base + ((int64_t)h1 * pw + w1) * pl, q3);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += lh0 * (lw0 * q0[i] + lw1 * q1[i]) + lh1 * (lw0 * q2[i] + lw1 * q3[i]);
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = acc[i] > 0.f ? acc[i] : 0.f;
        }
        stv<V>(out + pix * ldo + c, acc);
    }
}

// dx = sum over pyramid levels of adaptive_avg_pool backward: every level's gradient is scattered back to the
// full-resolution map in ONE pass (one write of dx instead of one per level plus the adds between them).  The bins
// that contain row h at output size s are i in [floor(h*s/H), ceil((h+1)*s/H) - 1] (1 or 2 of them).
// floor(a / b) for 0 <= a < 2^21 and the reciprocal rb = 1.0f / b: once a is shifted by +0.5 the exact quotient is at
// least 0.5 / b away from an integer, and the computed product is within 2^-23 * a / b of it -- smaller for a < 2^22
__device__ __forceinline__ int fdiv(int a, float rb) { return (int)(((float)a + 0.5f) * rb); }

template <int V>
__global__ __launch_bounds__(256) void pyramid_pool_bwd_kernel(PyramidArgs a, float* __restrict__ dx, int lddx, int N, int H,
                                                              int W, int C) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t npix = (int64_t)N * H * W;
    const float rH = 1.0f / (float)H, rW = 1.0f / (float)W;
    HN_PIX_LOOP(npix) {
        const int w = (int)(pix % W);
        const int64_t t = pix / W;
        const int h = (int)(t % H), n = (int)(t / H);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int k = 0; k < a.count; ++k) {
            const int Ho = a.h[k], Wo = a.w[k], ld = a.ld[k];
            const float rHo = 1.0f / (float)Ho, rWo = 1.0f / (float)Wo;
            const int i_lo = fdiv(h * Ho, rH), i_hi = fdiv((h + 1) * Ho + H - 1, rH) - 1;
            const int j_lo = fdiv(w * Wo, rW), j_hi = fdiv((w + 1) * Wo + W - 1, rW) - 1;
            const float* base = a.p[k] + (int64_t)n * Ho * Wo * ld + c;
            for (int i = i_lo; i <= i_hi; ++i) {
                const int bh = fdiv((i + 1) * H + Ho - 1, rHo) - fdiv(i * H, rHo);  // bin_end - bin_start
                for (int j = j_lo; j <= j_hi; ++j) {
                    const int bw = fdiv((j + 1) * W + Wo - 1, rWo) - fdiv(j * W, rWo);
                    const float inv = 1.f / (float)(bh * bw);
                    float g[V];
                    ldv<V>(base + (int64_t)(i * Wo + j) * ld, g);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += g[e] * inv;
                }
            }
        }
        stv<V>(dx + pix * lddx + c, acc);
    }
}

bool bad_geom(int N, int H, int W, int C, int Ho, int Wo, int lda, int ldb) {
    return N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || lda < C || ldb < C;
}

}  // namespace

// ---- backward of the pyramid merge in two passes over ONE read of the gradient -------------------------------------------
// out = relu(main + sum_k up(prior_k)).  The gradient of every prior is a bilinear down-gather of du = dout * [out > 0] over the
// whole map; done level by level (hn_bilinear_bwd: one block per prior pixel) each of the four levels read the 115 MB of du
// again at a quarter of the HBM rate (4 x 84 us in the headline step).  The bilinear weights are separable:
//   d prior_k[n, i, j] = sum_h wh_k(h, i) * ( sum_w ww_k(w, j) * du[n, h, w] ).
// Pass 1 (one thread-iteration per row (n, h) and channel piece): reads dout and out once, writes du, and sums every row into
// the pw_k column bins of every level -- the source column of a level moves on monotonically with w, so two running
// accumulators per level (the current bin and its right neighbour) suffice; S_k[n][h][j][c] goes to a workspace (N H sum(pw_k) C
// floats).  Pass 2: the column of row sums of every prior pixel, weighted (gather form, fixed order: bitwise reproducible).
namespace {
struct PyramidBwdArgs {
    float* dp[8];   // gradient of prior k: [N][h_k][w_k][ld_k]
    int h[8], w[8], ld[8];
    int soff[8];    // column-bin offset of level k inside a row of the workspace ([N][H][sum w_k][C])
    int count, sumw;
};

template <int V>
__global__ __launch_bounds__(256) void pyramid_merge_bwd_rows_kernel(const float* __restrict__ dout, int lddo,
                                                                    const float* __restrict__ out, int ldo, float* __restrict__ du,
                                                                    int lddu, PyramidBwdArgs a, float* __restrict__ S, int N, int H,
                                                                    int W, int C, int nseg, int segw) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t nitems = (int64_t)N * H * nseg;  // a row is cut into nseg segments of segw columns (more threads in flight)
    HN_PIX_LOOP(nitems) {
        const int seg = (int)(pix % nseg);
        const int64_t row = pix / nseg;
        const int w_lo = seg * segw, w_hi = min(W, w_lo + segw);
        float accA[8][V], accB[8][V];
        int cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int j1;
            float l0, l1;
            cur[k] = 0;
            if (k < a.count && w_lo < w_hi) bilinear_src(w_lo, (float)a.w[k] / (float)W, a.w[k], cur[k], j1, l0, l1);
#pragma unroll
            for (int e = 0; e < V; ++e) accA[k][e] = accB[k][e] = 0.f;
        }
        float* srow = S + (row * nseg + seg) * a.sumw * C + c;  // this segment's own slice: [row][seg][bins][C]
        for (int w = w_lo; w < w_hi; ++w) {
            float g[V], o[V], d[V];
            const int64_t at = row * W + w;
            ldv<V>(dout + at * lddo + c, g);
            ldv<V>(out + at * ldo + c, o);
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = o[e] > 0.f ? g[e] : 0.f;
            stv<V>(du + at * lddu + c, d);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k >= a.count) continue;
                int j0, j1;
                float l0, l1;
                bilinear_src(w, (float)a.w[k] / (float)W, a.w[k], j0, j1, l0, l1);
                if (j0 != cur[k]) {  // the source column moved on (by one: W >= w_k): bin cur[k] is complete
                    stv<V>(srow + (a.soff[k] + cur[k]) * C, accA[k]);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        accA[k][e] = accB[k][e];
                        accB[k][e] = 0.f;
                    }
                    cur[k] = j0;
                }
                if (j1 == j0) {  // (clamped at the last column: both weights on the same bin)
#pragma unroll
                    for (int e = 0; e < V; ++e) accA[k][e] += (l0 + l1) * d[e];
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        accA[k][e] += l0 * d[e];
                        accB[k][e] += l1 * d[e];
                    }
                }
            }
        }
        if (w_lo < w_hi) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k >= a.count) continue;
                stv<V>(srow + (a.soff[k] + cur[k]) * C, accA[k]);
                if (cur[k] + 1 < a.w[k]) stv<V>(srow + (a.soff[k] + cur[k] + 1) * C, accB[k]);
            }
        }
    }
}

// Pass 2.  Segment `seg` of a row wrote the bins from j0(first column) to min(j0(last column) + 1, w_k - 1) of level k and
// nothing else: exactly those slices are read.  One block per prior pixel and channel slab, threadIdx.y strides over the
// (segment, row) pairs -- the 1 x 1 level sums 8 x 40 row sums per pixel; one thread doing that alone took 139 us -- and an
// LDS reduction in fixed order closes it.
template <int V>
__global__ __launch_bounds__(256) void pyramid_merge_bwd_cols_kernel(PyramidBwdArgs a, const float* __restrict__ S, int N, int H,
                                                                    int W, int C, int nseg, int segw) {
    __shared__ float red[256 * 4];
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const bool live = c < C;
    int k = 0;
    int64_t rem = blockIdx.y;
    while (rem >= (int64_t)N * a.h[k] * a.w[k]) {
        rem -= (int64_t)N * a.h[k] * a.w[k];
        ++k;
    }
    const int ph = a.h[k], pw = a.w[k];
    const int j = (int)(rem % pw);
    const int64_t t = rem / pw;
    const int i = (int)(t % ph), n = (int)(t / ph);
    const float sh = (float)ph / (float)H, sw = (float)pw / (float)W;
    // rows that can read prior row i: source coordinate (h + 0.5) * sh - 0.5 within (i - 1, i + 1), one row of slack each way
    const int h_lo = max(0, (int)floorf(((float)i - 0.5f) / sh - 0.5f) - 1);
    const int h_hi = min(H - 1, (int)ceilf(((float)i + 1.5f) / sh - 0.5f) + 1);
    const int nh = h_hi - h_lo + 1;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (live) {
        for (int q = threadIdx.y; q < nseg * nh; q += blockDim.y) {
            const int seg = q / nh, h = h_lo + q - seg * nh;
            const int w_lo = seg * segw, w_hi = min(W, w_lo + segw);
            if (w_lo >= w_hi) continue;
            int f0, f1, g0, g1;
            float l0, l1;
            bilinear_src(w_lo, sw, pw, f0, f1, l0, l1);
            bilinear_src(w_hi - 1, sw, pw, g0, g1, l0, l1);
            if (j < f0 || j > min(g0 + 1, pw - 1)) continue;  // this segment never touched bin j
            const float wh = bilinear_weight(h, i, sh, ph);
            if (wh == 0.f) continue;
            float g[V];
            ldv<V>(S + ((((int64_t)n * H + h) * nseg + seg) * a.sumw + a.soff[k] + j) * C + c, g);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += wh * g[e];
        }
    }
    const int bx = blockDim.x, by = blockDim.y, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int e = 0; e < V; ++e) red[(ty * bx + tx) * V + e] = acc[e];
    __syncthreads();
    if (ty == 0 && live) {
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = 0.f;
            for (int q = 0; q < by; ++q) v += red[(q * bx + tx) * V + e];
            o[e] = v;
        }
        stv<V>(a.dp[k] + rem * a.ld[k] + c, o);
    }
}

// ---- the adaptive average poolings of all pyramid levels from ONE read of the map --------------------------------------------
// Level by level (hn_adaptive_avgpool_fwd: one block per bin) the four levels read the 231 MB of the encoder output four times.
// Same two passes as the merge's backward: pass 1 sums every row segment into the column bins of every level (a column lies in
// one bin, or in two where bins overlap -- the bin it FIRST lies in moves on monotonically, so two running sums per level),
// pass 2 adds the rows of a bin and divides by its area.  The PyramidBwdArgs fields are used as: dp = the pooled outputs.
template <int V>
__global__ __launch_bounds__(256) void pyramid_pool_fwd_rows_kernel(const float* __restrict__ x, int ldx, PyramidBwdArgs a,
                                                                   float* __restrict__ S, int N, int H, int W, int C, int nseg,
                                                                   int segw) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    const int64_t nitems = (int64_t)N * H * nseg;
    HN_PIX_LOOP(nitems) {
        const int seg = (int)(pix % nseg);
        const int64_t row = pix / nseg;
        const int w_lo = seg * segw, w_hi = min(W, w_lo + segw);
        if (w_lo >= w_hi) continue;
        float accA[8][V], accB[8][V];
        int cur[8];  // first bin whose end lies behind the current column
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cur[k] = 0;
            if (k < a.count)
                while (bin_end(cur[k], W, a.w[k]) <= w_lo) ++cur[k];
#pragma unroll
            for (int e = 0; e < V; ++e) accA[k][e] = accB[k][e] = 0.f;
        }
        float* srow = S + (row * nseg + seg) * a.sumw * C + c;
        for (int w = w_lo; w < w_hi; ++w) {
            float v[V];
            ldv<V>(x + (row * W + w) * ldx + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k >= a.count) continue;
                if (bin_end(cur[k], W, a.w[k]) <= w) {  // bin cur[k] is complete (bins are at least one column wide: one step)
                    stv<V>(srow + (a.soff[k] + cur[k]) * C, accA[k]);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        accA[k][e] = accB[k][e];
                        accB[k][e] = 0.f;
                    }
                    ++cur[k];
                }
                const bool second = cur[k] + 1 < a.w[k] && bin_start(cur[k] + 1, W, a.w[k]) <= w;  // overlapping neighbour
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    accA[k][e] += v[e];
                    if (second) accB[k][e] += v[e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= a.count) continue;
            stv<V>(srow + (a.soff[k] + cur[k]) * C, accA[k]);
            if (cur[k] + 1 < a.w[k]) stv<V>(srow + (a.soff[k] + cur[k] + 1) * C, accB[k]);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void pyramid_pool_fwd_cols_kernel(PyramidBwdArgs a, const float* __restrict__ S, int N, int H,
                                                                   int W, int C, int nseg, int segw) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= C) return;
    int64_t total = 0;
    for (int k = 0; k < a.count; ++k) total += (int64_t)N * a.h[k] * a.w[k];
    HN_PIX_LOOP(total) {
        int k = 0;
        int64_t rem = pix;
        while (rem >= (int64_t)N * a.h[k] * a.w[k]) {
            rem -= (int64_t)N * a.h[k] * a.w[k];
            ++k;
        }
        const int ph = a.h[k], pw = a.w[k];
        const int j = (int)(rem % pw);
        const int64_t t = rem / pw;
        const int i = (int)(t % ph), n = (int)(t / ph);
        const int h0 = bin_start(i, H, ph), h1 = bin_end(i, H, ph), w0 = bin_start(j, W, pw), w1 = bin_end(j, W, pw);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int seg = 0; seg < nseg; ++seg) {
            const int w_lo = seg * segw, w_hi = min(W, w_lo + segw);
            if (w_lo >= w_hi || w1 <= w_lo || w0 >= w_hi) continue;  // the segment holds no column of bin j
            // (a segment writes the bins from the first one that ends behind its first column to the neighbour of the last one
            //  it completes: every bin with a column in the segment is among them)
            for (int h = h0; h < h1; ++h) {
                float g[V];
                ldv<V>(S + ((((int64_t)n * H + h) * nseg + seg) * a.sumw + a.soff[k] + j) * C + c, g);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += g[e];
            }
        }
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= inv;
        stv<V>(a.dp[k] + rem * a.ld[k] + c, acc);
    }
}

}  // namespace

#define HN_LAUNCH_V(plan, kern, ...)                                                                     \
    do {                                                                                                 \
        if ((plan).V == 4) HN_LAUNCH((kern<4>), (plan).grid, (plan).block, 0, s, __VA_ARGS__);  \
        else HN_LAUNCH((kern<1>), (plan).grid, (plan).block, 0, s, __VA_ARGS__);                \
    } while (0)

extern "C" int hn_maxpool3x3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int N, int H, int W, int C,
                                   int Ho, int Wo, void* stream) {
    if (!x || !y || !idx || bad_geom(N, H, W, C, Ho, Wo, ldx, ldy)) return HN_EINVAL;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return HN_ESHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * Ho * Wo, C, {x, y}, {ldx, ldy});
    HN_LAUNCH_V(pl, maxpool_fwd_kernel, x, ldx, y, ldy, idx, N, H, W, C, Ho, Wo);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_maxpool3x3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int N, int H, int W,
                                   int C, int Ho, int Wo, void* stream) {
    if (!dy || !dx || !idx || bad_geom(N, H, W, C, Ho, Wo, lddx, lddy)) return HN_EINVAL;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return HN_ESHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * H * W, C, {dy, dx}, {lddy, lddx});
    HN_LAUNCH_V(pl, maxpool_bwd_kernel, dy, lddy, idx, dx, lddx, N, H, W, C, Ho, Wo);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int Ho,
                                       int Wo, void* stream) {
    if (!x || !y || bad_geom(N, H, W, C, Ho, Wo, ldx, ldy)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t opix = (int64_t)N * Ho * Wo;
    if ((int64_t)H * W >= 64 * (int64_t)Ho * Wo && opix <= 65535) {  // bins of >= 64 pixels: block per bin
        const RowLaunch pb = plan_pixel_blocks(opix, C, {x, y}, {ldx, ldy});
        HN_LAUNCH_V(pb, avgpool_fwd_bin_kernel, x, ldx, y, ldy, N, H, W, C, Ho, Wo);
        HN_CHECK_LAUNCH();
        return HN_OK;
    }
    const RowLaunch pl = plan_rows(opix, C, {x, y}, {ldx, ldy});
    HN_LAUNCH_V(pl, avgpool_fwd_kernel, x, ldx, y, ldy, N, H, W, C, Ho, Wo);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_adaptive_avgpool_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int Ho,
                                       int Wo, void* stream) {
    if (!dy || !dx || bad_geom(N, H, W, C, Ho, Wo, lddx, lddy)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * H * W, C, {dy, dx}, {lddy, lddx});
    HN_LAUNCH_V(pl, avgpool_bwd_kernel, dy, lddy, dx, lddx, N, H, W, C, Ho, Wo);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hin, int Win, int C, int Hout,
                               int Wout, void* stream) {
    if (!x || !y || bad_geom(N, Hin, Win, C, Hout, Wout, ldx, ldy)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Hout == 2 * Hin && Wout == 2 * Win) {
        const RowLaunch p2 = plan_rows((int64_t)N * ((Hin + 1) / 2) * ((Win + 1) / 2), C, {x, y}, {ldx, ldy});  // per 2 x 2 input block
        HN_LAUNCH_V(p2, bilinear_up2_fwd_kernel, x, ldx, y, ldy, N, Hin, Win, C);
        HN_CHECK_LAUNCH();
        return HN_OK;
    }
    const RowLaunch pl = plan_rows((int64_t)N * Hout * Wout, C, {x, y}, {ldx, ldy});
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    HN_LAUNCH_V(pl, bilinear_fwd_kernel, x, ldx, y, ldy, N, Hin, Win, C, Hout, Wout, sh, sw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bilinear_up2_planes(const float* x, int ldx, const float* slot, void* y_planes, int64_t plane_elems, int ldp,
                                      int N, int Hin, int Win, int C, void* stream) {
    if (!x || !y_planes || bad_geom(N, Hin, Win, C, 2 * Hin, 2 * Win, ldx, ldp)) return HN_EINVAL;  // (slot NULL: bf16)
    if (plane_elems < ((int64_t)N * 4 * Hin * Win - 1) * ldp + C) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = (reinterpret_cast<uintptr_t>(y_planes) & 7u) == 0 && plane_elems % 4 == 0 && ldp % 4 == 0;
    const RowLaunch p2 = plan_rows((int64_t)N * ((Hin + 1) / 2) * ((Win + 1) / 2), C,
                                   {x, v4 ? nullptr : reinterpret_cast<const void*>(uintptr_t(1))}, {ldx});
    unsigned short* yh = static_cast<unsigned short*>(y_planes);
    if (p2.V == 4)
        HN_LAUNCH((bilinear_up2_fwd_kernel<4, true>), p2.grid, p2.block, 0, s, x, ldx, (float*)nullptr, ldp, N, Hin, Win, C, yh,
                  (long long)plane_elems, slot);
    else
        HN_LAUNCH((bilinear_up2_fwd_kernel<1, true>), p2.grid, p2.block, 0, s, x, ldx, (float*)nullptr, ldp, N, Hin, Win, C, yh,
                  (long long)plane_elems, slot);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hin, int Win, int C, int Hout,
                               int Wout, void* stream) {
    if (!dy || !dx || bad_geom(N, Hin, Win, C, Hout, Wout, lddx, lddy)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    const int64_t ipix = (int64_t)N * Hin * Win;
    if ((int64_t)Hout * Wout >= 16 * (int64_t)Hin * Win && ipix <= 65535) {  // strong upsampling: block per input pixel
        const RowLaunch pb = plan_pixel_blocks(ipix, C, {dy, dx}, {lddy, lddx});
        HN_LAUNCH_V(pb, bilinear_bwd_pix_kernel, dy, lddy, dx, lddx, N, Hin, Win, C, Hout, Wout, sh, sw);
        HN_CHECK_LAUNCH();
        return HN_OK;
    }
    if (Hout == 2 * Hin && Wout == 2 * Win) {  // a thread-iteration per 2 x 2 block of input pixels
        const RowLaunch pb2 = plan_rows((int64_t)N * ((Hin + 1) / 2) * ((Win + 1) / 2), C, {dy, dx}, {lddy, lddx});
        HN_LAUNCH_V(pb2, bilinear_up2_bwd_kernel, dy, lddy, dx, lddx, N, Hin, Win, C);
        HN_CHECK_LAUNCH();
        return HN_OK;
    }
    const RowLaunch pl = plan_rows(ipix, C, {dy, dx}, {lddy, lddx});
    HN_LAUNCH_V(pl, bilinear_bwd_kernel, dy, lddy, dx, lddx, N, Hin, Win, C, Hout, Wout, sh, sw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_pyramid_merge_fwd(const float* mainp, int ldm, int n_priors, const float* const* priors, const int* hs,
                                    const int* ws, const int* lds, float* out, int ldo, int N, int H, int W, int C, int relu,
                                    void* stream) {
    if (!mainp || !out || n_priors < 0 || n_priors > 8 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || ldm < C || ldo < C)
        return HN_EINVAL;
    if (n_priors > 0 && (!priors || !hs || !ws || !lds)) return HN_EINVAL;
    PyramidArgs a;
    a.count = n_priors;
    bool aligned = true;
    for (int k = 0; k < 8; ++k) {
        a.p[k] = k < n_priors ? priors[k] : nullptr;
        a.h[k] = k < n_priors ? hs[k] : 1;
        a.w[k] = k < n_priors ? ws[k] : 1;
        a.ld[k] = k < n_priors ? lds[k] : 4;
        if (k < n_priors && (!priors[k] || hs[k] <= 0 || ws[k] <= 0 || lds[k] < C)) return HN_EINVAL;
        if (k < n_priors) aligned = aligned && hn_aligned16(priors[k]) && (lds[k] % 4 == 0);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * H * W, C, {mainp, out, aligned ? nullptr : (const void*)1}, {ldm, ldo});
    HN_LAUNCH_V(pl, pyramid_merge_kernel, mainp, ldm, a, out, ldo, N, H, W, C, relu);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// segments a row is cut into (workspace: N * H * segments * sum(w_k) * C floats)
extern "C" int hn_pyramid_merge_bwd_segments(int N, int H, int W) {
    int nseg = (int)((1536 + (int64_t)N * H - 1) / ((int64_t)N * H));
    nseg = nseg < 1 ? 1 : (nseg > 8 ? 8 : nseg);
    return nseg > W ? W : nseg;
}

extern "C" int hn_pyramid_merge_bwd(const float* dout, int lddo, const float* out, int ldo, float* du, int lddu, int n_priors,
                                    float* const* dps, const int* hs, const int* ws, const int* lds, float* workspace, int N,
                                    int H, int W, int C, void* stream) {
    if (!dout || !out || !du || !workspace || n_priors <= 0 || n_priors > 8 || !dps || !hs || !ws || !lds || N <= 0 || H <= 0 ||
        W <= 0 || C <= 0 || lddo < C || ldo < C || lddu < C)
        return HN_EINVAL;
    PyramidBwdArgs a;
    a.count = n_priors;
    a.sumw = 0;
    bool aligned = hn_aligned16(workspace);
    for (int k = 0; k < 8; ++k) {
        a.dp[k] = k < n_priors ? dps[k] : nullptr;
        a.h[k] = k < n_priors ? hs[k] : 1;
        a.w[k] = k < n_priors ? ws[k] : 1;
        a.ld[k] = k < n_priors ? lds[k] : 4;
        a.soff[k] = a.sumw;
        if (k < n_priors) {
            if (!dps[k] || hs[k] <= 0 || ws[k] <= 0 || lds[k] < C) return HN_EINVAL;
            if (hs[k] > H || ws[k] > W) return HN_ESHAPE;  // priors are upsampled: the running-bin scheme needs w_k <= W
            a.sumw += ws[k];
            aligned = aligned && hn_aligned16(dps[k]) && (lds[k] % 4 == 0);
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nseg = hn_pyramid_merge_bwd_segments(N, H, W);
    const int segw = (W + nseg - 1) / nseg;
    const RowLaunch pr = plan_rows((int64_t)N * H * nseg, C, {dout, out, du, aligned ? nullptr : (const void*)1}, {lddo, ldo, lddu});
    HN_LAUNCH_V(pr, pyramid_merge_bwd_rows_kernel, dout, lddo, out, ldo, du, lddu, a, workspace, N, H, W, C, nseg, segw);
    HN_CHECK_LAUNCH();
    int64_t total = 0;
    for (int k = 0; k < n_priors; ++k) total += (int64_t)N * hs[k] * ws[k];
    if (total > 65535) return HN_ESHAPE;
    RowLaunch pc = plan_pixel_blocks(total, C, {aligned ? nullptr : (const void*)1}, {});  // one block per prior pixel
    pc.V = pr.V;  // (the workspace rows were written with that vector width's alignment assumptions)
    if (pc.V == 1) { pc.grid = dim3((C + 15) / 16, (unsigned)total); pc.block = dim3(16, 16); }
    if (pc.V == 4) HN_LAUNCH((pyramid_merge_bwd_cols_kernel<4>), pc.grid, pc.block, 0, s, a, workspace, N, H, W, C, nseg, segw);
    else HN_LAUNCH((pyramid_merge_bwd_cols_kernel<1>), pc.grid, pc.block, 0, s, a, workspace, N, H, W, C, nseg, segw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_pyramid_pool_fwd(const float* x, int ldx, int n_levels, float* const* ys, const int* hos, const int* wos,
                                   const int* lds, float* workspace, int N, int H, int W, int C, void* stream) {
    if (!x || !workspace || n_levels <= 0 || n_levels > 8 || !ys || !hos || !wos || !lds || N <= 0 || H <= 0 || W <= 0 || C <= 0 ||
        ldx < C)
        return HN_EINVAL;
    PyramidBwdArgs a;
    a.count = n_levels;
    a.sumw = 0;
    bool aligned = hn_aligned16(workspace);
    for (int k = 0; k < 8; ++k) {
        a.dp[k] = k < n_levels ? ys[k] : nullptr;
        a.h[k] = k < n_levels ? hos[k] : 1;
        a.w[k] = k < n_levels ? wos[k] : 1;
        a.ld[k] = k < n_levels ? lds[k] : 4;
        a.soff[k] = a.sumw;
        if (k < n_levels) {
            if (!ys[k] || hos[k] <= 0 || wos[k] <= 0 || lds[k] < C) return HN_EINVAL;
            if (hos[k] > H || wos[k] > W) return HN_ESHAPE;  // pooling: every bin at least one pixel wide
            a.sumw += wos[k];
            aligned = aligned && hn_aligned16(ys[k]) && (lds[k] % 4 == 0);
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nseg = hn_pyramid_merge_bwd_segments(N, H, W);
    const int segw = (W + nseg - 1) / nseg;
    const RowLaunch pr = plan_rows((int64_t)N * H * nseg, C, {x, aligned ? nullptr : (const void*)1}, {ldx});
    HN_LAUNCH_V(pr, pyramid_pool_fwd_rows_kernel, x, ldx, a, workspace, N, H, W, C, nseg, segw);
    HN_CHECK_LAUNCH();
    int64_t total = 0;
    for (int k = 0; k < n_levels; ++k) total += (int64_t)N * hos[k] * wos[k];
    RowLaunch pc = plan_rows(total, C, {aligned ? nullptr : (const void*)1}, {});
    pc.V = pr.V;
    if (pc.V == 4) HN_LAUNCH((pyramid_pool_fwd_cols_kernel<4>), pc.grid, pc.block, 0, s, a, workspace, N, H, W, C, nseg, segw);
    else HN_LAUNCH((pyramid_pool_fwd_cols_kernel<1>), pc.grid, pc.block, 0, s, a, workspace, N, H, W, C, nseg, segw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_pyramid_pool_bwd(int n_levels, const float* const* dys, const int* hos, const int* wos, const int* lds,
                                   float* dx, int lddx, int N, int H, int W, int C, void* stream) {
    if (!dx || n_levels <= 0 || n_levels > 8 || !dys || !hos || !wos || !lds || N <= 0 || H <= 0 || W <= 0 || C <= 0 ||
        lddx < C)
        return HN_EINVAL;
    for (int k = 0; k < n_levels; ++k)  // bin arithmetic of the kernel (fdiv) is exact below 2^21
        if ((int64_t)(H + 1) * (hos[k] + 1) >= (1 << 21) || (int64_t)(W + 1) * (wos[k] + 1) >= (1 << 21)) return HN_ESHAPE;
    PyramidArgs a;
    a.count = n_levels;
    bool aligned = true;
    for (int k = 0; k < 8; ++k) {
        const bool on = k < n_levels;
        if (on && (!dys[k] || hos[k] <= 0 || wos[k] <= 0 || lds[k] < C)) return HN_EINVAL;
        a.p[k] = on ? dys[k] : nullptr;
        a.h[k] = on ? hos[k] : 1;
        a.w[k] = on ? wos[k] : 1;
        a.ld[k] = on ? lds[k] : 4;
        if (on) aligned = aligned && hn_aligned16(dys[k]) && (lds[k] % 4 == 0);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * H * W, C, {dx, aligned ? nullptr : (const void*)1}, {lddx});
    HN_LAUNCH_V(pl, pyramid_pool_bwd_kernel, a, dx, lddx, N, H, W, C);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
