// Loss kernels (per-pixel cross entropy, adversarial MSE / BCE-with-logits against a constant map),
// argmax + confusion-matrix histogram for mIoU, and the elementwise optimizer updates.  All HBM-bound.
#include "hn_common.h"

namespace {

constexpr int MAXC = 32;  // class count handled in registers by the per-pixel kernels

__device__ __forceinline__ float block_sum(float v, float* red) {
    // 256 threads: wave shuffle, then LDS across the 4 waves
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// `weight` (optional): per-pixel factor on the loss (nn.CrossEntropyLoss(reduction='none') times a certainty map,
// cm/train_trgb_segnet_conf.py:468-477); out[1] then counts ALL pixels (torch.mean over the map), not the non-ignored.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                    int64_t npix, int C, int64_t ignore, float* __restrict__ out,
                                                    const float* __restrict__ weight) {
    __shared__ float red[4];
    float loss = 0.f, cnt = 0.f;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lab = labels[pix];
        if (lab == ignore || lab < 0 || lab >= C) continue;
        const float* row = logits + pix * ld;
        float v[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
        float se = 0.f, xl = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { se += expf(v[c] - mx); if (c == (int)lab) xl = v[c]; }
        loss += ((mx + logf(se)) - xl) * (weight != nullptr ? weight[pix] : 1.f);
        cnt += 1.f;
    }
    if (weight != nullptr) {  // mean over every pixel of the map
        cnt = 0.f;
        for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) cnt += 1.f;
    }
    const float l = block_sum(loss, red);
    const float n = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, l);
        atomicAdd(out + 1, n);
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                    int64_t npix, int C, int64_t ignore, const float* __restrict__ out,
                                                    const float* __restrict__ gout, float* __restrict__ dl, int lddl,
                                                    const float* __restrict__ weight) {
    const float cnt = out[1];
    const float g0 = cnt > 0.f ? gout[0] / cnt : 0.f;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lab = labels[pix];
        float* drow = dl + pix * lddl;
        if (lab == ignore || lab < 0 || lab >= C) {
            for (int c = 0; c < C; ++c) drow[c] = 0.f;
            continue;
        }
        const float g = weight != nullptr ? g0 * weight[pix] : g0;
        const float* row = logits + pix * ld;
        float v[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
        const float inv = g / se;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) drow[c] = v[c] * inv - (c == (int)lab ? g : 0.f);
    }
}

// The same two kernels for DENSE rows (ld == C, lddl == C: the segmentation head's output): a thread's row is C consecutive floats
// 4 C bytes from its neighbour's, so per-thread row loads reach memory as C strided dword instructions -- 135 us for the 112 MB of
// the headline step's logits and labels, 6 x the HBM time.  Here a block stages 256 rows (one contiguous run of 256 C floats)
// through LDS with coalesced dword accesses, both ways; a thread reads its row at stride C floats (any odd C: conflict-free).
__global__ __launch_bounds__(256) void ce_fwd_dense_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int64_t npix, int C, int64_t ignore, float* __restrict__ out,
                                                          const float* __restrict__ weight) {
    __shared__ float red[4];
    __shared__ float tile[256 * MAXC];
    float loss = 0.f, cnt = 0.f;
    const int64_t nchunks = (npix + 255) / 256;
    // The grid is capped (every block ends with two float atomics on the same two words, ~11 ns apiece at device scope: 4096
    // blocks spent 90 us there), so a block walks several chunks: the next chunk's elements are fetched into registers before
    // this one is reduced.
    float nxt[MAXC];
    auto fetch = [&](int64_t chunk) {
        const int64_t p0 = chunk * 256;
        const int n = chunk < nchunks ? (int)min((int64_t)256, npix - p0) * C : 0;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int i = threadIdx.x + 256 * k;
            nxt[k] = (k < C && i < n) ? logits[p0 * C + i] : 0.f;
        }
    };
    fetch(blockIdx.x);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t p0 = chunk * 256;
        const int rows = (int)min((int64_t)256, npix - p0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
            if (k < C) tile[threadIdx.x + 256 * k] = nxt[k];
        __syncthreads();
        fetch(chunk + gridDim.x);
        const int64_t pix = p0 + threadIdx.x;
        if ((int)threadIdx.x >= rows) continue;
        if (weight != nullptr) cnt += 1.f;  // mean over every pixel of the map
        const int64_t lab = labels[pix];
        if (lab == ignore || lab < 0 || lab >= C) continue;
        const float* row = tile + threadIdx.x * C;
        float v[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
        float se = 0.f, xl = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { se += expf(v[c] - mx); if (c == (int)lab) xl = v[c]; }
        loss += ((mx + logf(se)) - xl) * (weight != nullptr ? weight[pix] : 1.f);
        if (weight == nullptr) cnt += 1.f;
    }
    const float l = block_sum(loss, red);
    const float n = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, l);
        atomicAdd(out + 1, n);
    }
}

__global__ __launch_bounds__(256) void ce_bwd_dense_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int64_t npix, int C, int64_t ignore, const float* __restrict__ out,
                                                          const float* __restrict__ gout, float* __restrict__ dl,
                                                          const float* __restrict__ weight) {
    __shared__ float tile[256 * MAXC];
    const float cnt = out[1];
    const float g0 = cnt > 0.f ? gout[0] / cnt : 0.f;
    const int64_t nchunks = (npix + 255) / 256;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t p0 = chunk * 256;
        const int rows = (int)min((int64_t)256, npix - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < rows * C; i += 256) tile[i] = logits[p0 * C + i];
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            const int64_t pix = p0 + threadIdx.x;
            const int64_t lab = labels[pix];
            float* row = tile + threadIdx.x * C;  // (only this thread touches its row between the barriers)
            if (lab == ignore || lab < 0 || lab >= C) {
                for (int c = 0; c < C; ++c) row[c] = 0.f;
            } else {
                const float g = weight != nullptr ? g0 * weight[pix] : g0;
                float v[MAXC];
                float mx = -INFINITY;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
                float se = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
                const float inv = g / se;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) row[c] = v[c] * inv - (c == (int)lab ? g : 0.f);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < rows * C; i += 256) dl[p0 * C + i] = tile[i];
    }
}

// Night-supervision teacher outputs (cm/train_trgb_segnet_conf.py:430-434, 468-472): p = softmax(logits) per pixel;
// label = argmax p (first maximum); cert = max softmax(p) -- the reference applies softmax to the already normalised
// probabilities (`cert = softmax(out_night_ir_only)` where out_night_ir_only is itself a softmax output), restated as is.
__global__ __launch_bounds__(256) void teacher_cert_kernel(const float* __restrict__ logits, int ld, int64_t npix, int C,
                                                          int64_t* __restrict__ label, float* __restrict__ cert) {
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const float* row = logits + pix * ld;
        float v[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = row[c]; mx = fmaxf(mx, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
        float best = -1.f, pmax = 0.f;
        int bi = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                v[c] = v[c] / se;  // p_c
                if (v[c] > best) { best = v[c]; bi = c; }
                pmax = fmaxf(pmax, v[c]);
            }
        // softmax of the probabilities: exp(p_c - pmax) / sum
        float se2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) se2 += expf(v[c] - pmax);
        if (label != nullptr) label[pix] = bi;
        if (cert != nullptr) cert[pix] = 1.f / se2;  // the maximum of softmax(p) belongs to p_max: exp(0) / sum
    }
}

__device__ __forceinline__ float adv_value(float x, float t, int kind) {
    if (kind == HN_LOSS_MSE) return (x - t) * (x - t);
    // BCEWithLogits: max(x,0) - x*t + log1p(exp(-|x|))
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float adv_grad(float x, float t, int kind) {
    if (kind == HN_LOSS_MSE) return 2.f * (x - t);
    return 1.f / (1.f + expf(-x)) - t;
}

__global__ __launch_bounds__(256) void adv_fwd_kernel(const float* __restrict__ x, int64_t n, float t, int kind, float inv_n,
                                                     float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += adv_value(x[i], t, kind);
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, s * inv_n);
}

__global__ __launch_bounds__(256) void adv_bwd_kernel(const float* __restrict__ x, int64_t n, float t, int kind, float inv_n,
                                                     const float* __restrict__ gout, float* __restrict__ dx) {
    const float g = gout[0] * inv_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = g * adv_grad(x[i], t, kind);
}

__global__ __launch_bounds__(256) void argmax_conf_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                         int64_t npix, int C, int K, unsigned long long* __restrict__ conf,
                                                         uint8_t* __restrict__ pred_out) {
    extern __shared__ unsigned int hist[];  // K*K
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const float* row = logits + pix * ld;
        float best = row[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = row[c];
            if (v > best) { best = v; bi = c; }
        }
        if (pred_out != nullptr) pred_out[pix] = (uint8_t)bi;
        if (labels != nullptr) {
            const int64_t lab = labels[pix];
            if (lab >= 0 && lab < K && bi < K) atomicAdd(&hist[(int)lab * K + bi], 1u);
        }
    }
    __syncthreads();
    if (labels != nullptr)
        for (int i = threadIdx.x; i < K * K; i += blockDim.x)
            if (hist[i] != 0u) atomicAdd(conf + i, (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                     int64_t n, float lr, float alpha, float eps, float wd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;
        sq[i] = s;
        p[i] = pi - lr * gi / (sqrtf(s) + eps);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float step_size, float beta1, float beta2,
                                                  float eps, float wd, float inv_sqrt_bc2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

inline int flat_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}


// ---- the certainty branch of the custom architecture (cm/models/trgb_segnet.py:383-387, cm/train_trgb_segnet_conf.py:482-491) ----
__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.f / (1.f + expf(-x[i]));
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * y[i] * (1.f - y[i]);
}

// probability of the labelled class, p_y = softmax(logits)[label]; 0 where the label is outside [0, C)
__device__ __forceinline__ float prob_of_label(const float* row, int C, int64_t lab, float* probs) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { probs[c] = row[c]; mx = fmaxf(mx, probs[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { probs[c] = expf(probs[c] - mx); se += probs[c]; }
    const float inv = 1.f / se;
    float py = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) { probs[c] *= inv; if (c == (int)lab) py = probs[c]; }
    return (lab >= 0 && lab < C) ? py : 0.f;
}

// The trainer's certainty loss as written: target t[m, p] = 1 - p_y (image m, pixel p), prediction a[n, p] (the sigmoid map,
// [N, 1, H, W]); `a - t` broadcasts [N, 1, H, W] against [N, H, W] to [N, N, H, W], so the loss is
//   10 / (N N HW) sum_p sum_n sum_m (a[n, p] - t[m, p])^2 = 10 / (N N HW) sum_p (N A2 + N T2 - 2 A1 T1),
// A1, A2 = sum_n a, a^2 and T1, T2 = sum_m t, t^2 at pixel p.  One thread per pixel p; out[0] += the unscaled sum.
__global__ __launch_bounds__(256) void cert_loss_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ a, int N, int64_t hw, int C,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
        double a1 = 0., a2 = 0., t1 = 0., t2 = 0.;  // (the expanded square cancels: the four sums in double)
        for (int n = 0; n < N; ++n) {
            const int64_t pix = (int64_t)n * hw + p;
            float probs[MAXC];
            const double t = 1.0 - (double)prob_of_label(logits + pix * ld, C, labels[pix], probs);
            const double av = a[pix];
            a1 += av; a2 += av * av; t1 += t; t2 += t * t;
        }
        acc += (float)((double)N * (a2 + t2) - 2.0 * a1 * t1);
    }
    const float l = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, l);
}

// da[n, p] = s 2 (N a[n, p] - T1),  dt[m, p] = s 2 (N t[m, p] - A1),  dlogits[m, p, c] = -dt p_y (delta_cy - p_c),
// s = 10 / (N N HW) * gout
__global__ __launch_bounds__(256) void cert_loss_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ a, int N, int64_t hw, int C,
                                                           const float* __restrict__ gout, float* __restrict__ da,
                                                           float* __restrict__ dl, int lddl) {
    const float s = 20.f / ((float)N * (float)N * (float)hw) * gout[0];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
        float a1 = 0.f, t1 = 0.f;
        for (int n = 0; n < N; ++n) {
            const int64_t pix = (int64_t)n * hw + p;
            float probs[MAXC];
            t1 += 1.f - prob_of_label(logits + pix * ld, C, labels[pix], probs);
            a1 += a[pix];
        }
        for (int n = 0; n < N; ++n) {
            const int64_t pix = (int64_t)n * hw + p;
            float probs[MAXC];
            const int64_t lab = labels[pix];
            const float py = prob_of_label(logits + pix * ld, C, lab, probs);
            if (da != nullptr) da[pix] = s * ((float)N * a[pix] - t1);
            const float dt = s * ((float)N * (1.f - py) - a1);
            float* row = dl + pix * lddl;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) row[c] = -dt * py * ((c == (int)lab ? 1.f : 0.f) - probs[c]);
        }
    }
}
}  // namespace

extern "C" int hn_ce_fwd(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int64_t ignore_index,
                         float* out, void* stream) {
    if (!logits || !labels || !out || npix <= 0 || C <= 0 || C > MAXC || ld < C) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, 2 * sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    if (ld == C)  // (512 blocks: each ends with two float atomics on the same two words, ~11 ns apiece at device scope)
        HN_LAUNCH(ce_fwd_dense_kernel, dim3(min(flat_grid(npix), 1024)), dim3(256), 0, s, logits, labels, npix, C, ignore_index, out,
                  static_cast<const float*>(nullptr));
    else
        HN_LAUNCH(ce_fwd_kernel, dim3(flat_grid(npix)), dim3(256), 0, s, logits, ld, labels, npix, C, ignore_index, out,
                  static_cast<const float*>(nullptr));
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_ce_weighted_fwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t npix, int C,
                                  float* out, void* stream) {
    if (!logits || !labels || !weight || !out || npix <= 0 || C <= 0 || C > MAXC || ld < C) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, 2 * sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    if (ld == C) HN_LAUNCH(ce_fwd_dense_kernel, dim3(min(flat_grid(npix), 1024)), dim3(256), 0, s, logits, labels, npix, C, (int64_t)-100, out, weight);
    else HN_LAUNCH(ce_fwd_kernel, dim3(flat_grid(npix)), dim3(256), 0, s, logits, ld, labels, npix, C, (int64_t)-100, out, weight);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_ce_weighted_bwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t npix, int C,
                                  const float* out, const float* gout, float* dlogits, int lddl, void* stream) {
    if (!logits || !labels || !weight || !out || !gout || !dlogits || npix <= 0 || C <= 0 || C > MAXC || ld < C || lddl < C)
        return HN_EINVAL;
    if (ld == C && lddl == C)
        HN_LAUNCH(ce_bwd_dense_kernel, dim3(flat_grid(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels, npix, C,
                  (int64_t)-100, out, gout, dlogits, weight);
    else
        HN_LAUNCH(ce_bwd_kernel, dim3(flat_grid(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ld, labels, npix, C,
                  (int64_t)-100, out, gout, dlogits, lddl, weight);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_teacher_cert(const float* logits, int ld, int64_t npix, int C, int64_t* label, float* cert, void* stream) {
    if (!logits || (!label && !cert) || npix <= 0 || C <= 0 || C > MAXC || ld < C) return HN_EINVAL;
    HN_LAUNCH(teacher_cert_kernel, dim3(flat_grid(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ld, npix, C,
              label, cert);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_ce_bwd(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int64_t ignore_index,
                         const float* out, const float* gout, float* dlogits, int lddl, void* stream) {
    if (!logits || !labels || !out || !gout || !dlogits || npix <= 0 || C <= 0 || C > MAXC || ld < C || lddl < C)
        return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ld == C && lddl == C)
        HN_LAUNCH(ce_bwd_dense_kernel, dim3(flat_grid(npix)), dim3(256), 0, s, logits, labels, npix, C, ignore_index, out, gout,
                  dlogits, static_cast<const float*>(nullptr));
    else
        HN_LAUNCH(ce_bwd_kernel, dim3(flat_grid(npix)), dim3(256), 0, s, logits, ld, labels, npix, C, ignore_index, out,
                  gout, dlogits, lddl, static_cast<const float*>(nullptr));
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_adv_loss_fwd(const float* x, int64_t n, float target, int kind, float* out, void* stream) {
    if (!x || !out || n <= 0 || (kind != HN_LOSS_MSE && kind != HN_LOSS_BCE)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    int grid = flat_grid(n);
    if (grid > 512) grid = 512;
    HN_LAUNCH(adv_fwd_kernel, dim3(grid), dim3(256), 0, s, x, n, target, kind, (float)(1.0 / (double)n), out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_adv_loss_bwd(const float* x, int64_t n, float target, int kind, const float* gout, float* dx,
                               void* stream) {
    if (!x || !gout || !dx || n <= 0 || (kind != HN_LOSS_MSE && kind != HN_LOSS_BCE)) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HN_LAUNCH(adv_bwd_kernel, dim3(flat_grid(n)), dim3(256), 0, s, x, n, target, kind, (float)(1.0 / (double)n),
                       gout, dx);
    HN_CHECK_LAUNCH();
    return HN_OK;
}


extern "C" int hn_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (x == nullptr || y == nullptr || n <= 0) return HN_EINVAL;
    HN_LAUNCH(sigmoid_fwd_kernel, dim3(flat_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
    if (y == nullptr || dy == nullptr || dx == nullptr || n <= 0) return HN_EINVAL;
    HN_LAUNCH(sigmoid_bwd_kernel, dim3(flat_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), y, dy, dx, n);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_cert_loss_fwd(const float* logits, int ld, const int64_t* labels, const float* cert, int N, int64_t hw, int C,
                                float* out, void* stream) {
    if (logits == nullptr || labels == nullptr || cert == nullptr || out == nullptr || N <= 0 || hw <= 0) return HN_EINVAL;
    if (C <= 0 || C > MAXC || ld < C) return HN_ESHAPE;
    HN_LAUNCH(cert_loss_fwd_kernel, dim3(min(flat_grid(hw), 512)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ld, labels, cert, N, hw,
              C, out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_cert_loss_bwd(const float* logits, int ld, const int64_t* labels, const float* cert, int N, int64_t hw, int C,
                                const float* gout, float* dcert, float* dlogits, int lddl, void* stream) {
    if (logits == nullptr || labels == nullptr || cert == nullptr || gout == nullptr || dlogits == nullptr || N <= 0 || hw <= 0)
        return HN_EINVAL;
    if (C <= 0 || C > MAXC || ld < C || lddl < C) return HN_ESHAPE;
    HN_LAUNCH(cert_loss_bwd_kernel, dim3(flat_grid(hw)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ld, labels, cert, N, hw,
              C, gout, dcert, dlogits, lddl);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_argmax_confusion(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int K,
                                   int64_t* conf, uint8_t* pred_out, void* stream) {
    if (!logits || npix <= 0 || C <= 0 || C > 255 || ld < C || K <= 0 || K > 64) return HN_EINVAL;
    if (labels != nullptr && conf == nullptr) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = flat_grid(npix);
    if (grid > 1024) grid = 1024;
    HN_LAUNCH(argmax_conf_kernel, dim3(grid), dim3(256), (size_t)K * K * sizeof(unsigned int), s, logits, ld, labels,
                       npix, C, K, reinterpret_cast<unsigned long long*>(conf), pred_out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_rmsprop_step(float* param, const float* grad, float* square_avg, int64_t n, float lr, float alpha,
                               float eps, float weight_decay, void* stream) {
    if (!param || !grad || !square_avg || n <= 0) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HN_LAUNCH(rmsprop_kernel, dim3(flat_grid(n)), dim3(256), 0, s, param, grad, square_avg, n, lr, alpha, eps,
                       weight_decay);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return HN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    HN_LAUNCH(adam_kernel, dim3(flat_grid(n)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n,
                       (float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)));
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" const char* hn_strerror(int code) {
    switch (code) {
        case HN_OK: return "ok";
        case HN_EINVAL: return "invalid argument";
        case HN_EALIGN: return "operand not aligned as required";
        case HN_ELAUNCH: return "HIP launch failed";
        case HN_ESHAPE: return "inconsistent shapes";
        default: return "unknown heatnet_hip error";
    }
}

// bumped whenever kernel tilings / variants change (invalidates heatnet-pub_amd/tune_cache.json)
extern "C" int hn_version(void) { return 13; }
