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

bool bad_geom(int N, int H, int W, int C, int Ho, int Wo, int lda, int ldb) {
    return N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || lda < C || ldb < C;
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
    const RowLaunch pl = plan_rows((int64_t)N * Ho * Wo, C, {x, y}, {ldx, ldy});
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
    const RowLaunch pl = plan_rows((int64_t)N * Hout * Wout, C, {x, y}, {ldx, ldy});
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    HN_LAUNCH_V(pl, bilinear_fwd_kernel, x, ldx, y, ldy, N, Hin, Win, C, Hout, Wout, sh, sw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hin, int Win, int C, int Hout,
                               int Wout, void* stream) {
    if (!dy || !dx || bad_geom(N, Hin, Win, C, Hout, Wout, lddx, lddy)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RowLaunch pl = plan_rows((int64_t)N * Hin * Win, C, {dy, dx}, {lddy, lddx});
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    HN_LAUNCH_V(pl, bilinear_bwd_kernel, dy, lddy, dx, lddx, N, Hin, Win, C, Hout, Wout, sh, sw);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
