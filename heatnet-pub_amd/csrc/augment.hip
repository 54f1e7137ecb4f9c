// Per-sample augmentation of the thermal data loader on the GPU: one fused pass that replaces
// models/confusion_maximization/thermal_loader.py:642-735 (+ rectDropTensor, train_trgb_segnet_conf.py:82-86) for a batch
// of already resized samples: random crop, 16-bit thermal clip / scale, horizontal flip, rotation (Pillow's bilinear
// affine transform for the images, its 16.16 fixed-point nearest transform for the labels), to_tensor, normalisation to
// [-1, 1] and the modality block drop.  One thread per output pixel; reads 3 + 2 + 1 bytes per tap from uint8 / uint16
// sources and writes the NHWC fp32 tensors the network consumes (HBM-bound: ~40 B per output pixel).
//
// Arithmetic follows Pillow exactly (oracle/thermal_aug_ref.py, pinned against Pillow itself): coordinates and the
// bilinear blend in double, uint8 results truncated before to_tensor, float32 results rounded once.
#include "hn_common.h"

namespace {

constexpr double IR_MIN = 21800.0, IR_SPAN = 3200.0;  // thermal_loader.py:650-660: clip to [21800, 25000], scale to [0, 1]

struct Src {
    const uint8_t* rgb;
    const uint16_t* ir;
    const uint8_t* label;
    int H0, W0;
};

// pixel (ys, xs) of the cropped, flipped sample -> index into the source arrays
__device__ __forceinline__ int64_t src_index(const Src& s, const HnAugSample& a, int n, int ys, int xs, int w) {
    const int xu = a.flip ? (w - 1 - xs) : xs;
    return ((int64_t)n * s.H0 + (a.crop_i + ys)) * s.W0 + (a.crop_j + xu);
}

__device__ __forceinline__ float ir_value(const Src& s, int64_t idx) {
#pragma clang fp contract(off)
    double v = (double)s.ir[idx];
    v = v < IR_MIN ? IR_MIN : (v > IR_MIN + IR_SPAN ? IR_MIN + IR_SPAN : v);
    return (float)((v - IR_MIN) / IR_SPAN);
}

__global__ __launch_bounds__(256) void thermal_augment_kernel(Src s, const HnAugSample* __restrict__ params,
                                                             const float* __restrict__ class_scale, int N, int h, int w,
                                                             float* __restrict__ rgb_out, float* __restrict__ ir_out,
                                                             int64_t* __restrict__ label_out) {
    // Pillow's blend is plain C double arithmetic (multiply, then add, each rounded): a fused multiply-add rounds once and
    // flips the uint8 truncation of values that land within an ulp of an integer (seen: 173 vs 174)
#pragma clang fp contract(off)
    const int64_t total = (int64_t)N * h * w;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % w);
        const int64_t t = p / w;
        const int y = (int)(t % h), n = (int)(t / h);
        const HnAugSample a = params[n];
        float r[3] = {0.f, 0.f, 0.f}, irv = 0.f;  // to_tensor values in [0, 1]
        int64_t lab = 0;
        if (a.mode == 0 || a.mode == 2) {  // no rotation / Pillow's exact 180-degree fast path
            const int ys = a.mode == 2 ? h - 1 - y : y, xs = a.mode == 2 ? w - 1 - x : x;
            const int64_t i = src_index(s, a, n, ys, xs, w);
            for (int c = 0; c < 3; ++c) r[c] = (float)s.rgb[i * 3 + c] / 255.0f;
            irv = ir_value(s, i);
            if (s.label != nullptr) lab = s.label[i];
        } else {
            // bilinear affine transform (Geometry.c: bilinear_filter8 / bilinear_filter32F), zero outside
            double xin = a.m[0] * (x + 0.5) + a.m[1] * (y + 0.5) + a.m[2];
            double yin = a.m[3] * (x + 0.5) + a.m[4] * (y + 0.5) + a.m[5];
            if (xin >= 0.0 && xin < (double)w && yin >= 0.0 && yin < (double)h) {
                xin -= 0.5;
                yin -= 0.5;
                const int xf = (int)floor(xin), yf = (int)floor(yin);
                const double dx = xin - xf, dy = yin - yf;
                const int x0 = min(max(xf, 0), w - 1), x1 = min(max(xf + 1, 0), w - 1), y0 = min(max(yf, 0), h - 1);
                const bool has_y1 = yf + 1 >= 0 && yf + 1 < h;
                const int y1 = has_y1 ? yf + 1 : y0;
                const int64_t i00 = src_index(s, a, n, y0, x0, w), i01 = src_index(s, a, n, y0, x1, w);
                const int64_t i10 = src_index(s, a, n, y1, x0, w), i11 = src_index(s, a, n, y1, x1, w);
                for (int c = 0; c < 3; ++c) {
                    const double p00 = s.rgb[i00 * 3 + c], p01 = s.rgb[i01 * 3 + c];
                    const double p10 = s.rgb[i10 * 3 + c], p11 = s.rgb[i11 * 3 + c];
                    const double v1 = p00 + (p01 - p00) * dx;
                    const double v2 = has_y1 ? p10 + (p11 - p10) * dx : v1;
                    const uint8_t q = (uint8_t)(v1 + (v2 - v1) * dy);  // truncation, as the C cast in Pillow
                    r[c] = (float)q / 255.0f;
                }
                const double p00 = ir_value(s, i00), p01 = ir_value(s, i01), p10 = ir_value(s, i10), p11 = ir_value(s, i11);
                const double v1 = p00 + (p01 - p00) * dx;
                const double v2 = has_y1 ? p10 + (p11 - p10) * dx : v1;
                irv = (float)(v1 + (v2 - v1) * dy);
            }
            if (s.label != nullptr) {  // nearest, 16.16 fixed point (Geometry.c: affine_fixed)
                const long long xx = a.fx[2] + a.fx[1] * y + a.fx[0] * x, yy = a.fx[5] + a.fx[4] * y + a.fx[3] * x;
                const long long xi = xx >> 16, yi = yy >> 16;
                if (xi >= 0 && xi < w && yi >= 0 && yi < h) lab = s.label[src_index(s, a, n, (int)yi, (int)xi, w)];
            }
        }
        // F.normalize(mean 0.5, std 0.5), then the modality block drop (zeros in the normalised tensor)
        const bool in_rect = y >= a.drop_i && y < a.drop_i + a.drop_h && x >= a.drop_j && x < a.drop_j + a.drop_w;
        const bool drop_rgb = in_rect && a.drop_target == 1, drop_ir = in_rect && a.drop_target == 2;
        for (int c = 0; c < 3; ++c) rgb_out[p * 3 + c] = drop_rgb ? 0.f : (r[c] - 0.5f) / 0.5f;
        // then the trainer's thermal intensity augmentations: whole-sample scale and per-class scale (smartAugment)
        float iv = drop_ir ? 0.f : (irv - 0.5f) / 0.5f;
        iv *= a.ir_scale;
        if (class_scale != nullptr) iv *= class_scale[lab < HN_AUG_CLASSES ? lab : HN_AUG_CLASSES - 1];
        ir_out[p] = iv;
        if (label_out != nullptr) label_out[p] = lab;
    }
}

}  // namespace

extern "C" int hn_thermal_augment(const uint8_t* rgb, const uint16_t* ir, const uint8_t* label, int N, int H0, int W0,
                                  const HnAugSample* params, const float* class_scale, int h, int w, float* rgb_out,
                                  float* ir_out, int64_t* label_out, void* stream) {
    if (!rgb || !ir || !params || !rgb_out || !ir_out || N <= 0 || H0 <= 0 || W0 <= 0 || h <= 0 || w <= 0 || h > H0 || w > W0)
        return HN_EINVAL;
    if ((label == nullptr) != (label_out == nullptr)) return HN_EINVAL;
    if (class_scale != nullptr && label == nullptr) return HN_EINVAL;
    const int64_t total = (int64_t)N * h * w;
    const int64_t blocks = (total + 255) / 256;
    Src s{rgb, ir, label, H0, W0};
    HN_LAUNCH(thermal_augment_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
              static_cast<hipStream_t>(stream), s, params, class_scale, N, h, w, rgb_out, ir_out, label_out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}


// ---- decoded image -> 960 x 320 -> column crop (cm/thermal_loader.py:621-640, 783-791) -----------------------------------
// cv2.resize(INTER_LINEAR) for the uint8 RGB image and the uint16 thermal image, cv2.resize(INTER_NEAREST) for the label
// map, followed by the loaders' column crop -- only the cropped columns are computed.  Arithmetic restated from OpenCV 4.x
// (modules/imgproc/src/resize.cpp; oracle/thermal_resize_ref.py states it in numpy): uint8 in 11-bit fixed point, uint16
// in float32 (multiply, then add: no contraction), an exact 2 x 2 shrink through the INTER_AREA fast path.  PARITY
// UNPINNED against cv2 itself (not installed in the build container); bit-exact against the oracle.
namespace {

struct LinTap {
    int s0, s1;
    float frac;
};

__device__ __forceinline__ LinTap lin_tap(int d, int dsize, int ssize) {
    const double scale = (double)ssize / (double)dsize;
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    float frac = f - (float)s;
    if (s < 0) { s = 0; frac = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; frac = 0.f; }
    LinTap t;
    t.s0 = s;
    t.s1 = min(s + 1, ssize - 1);
    t.frac = frac;
    return t;
}

__global__ __launch_bounds__(256) void thermal_resize_kernel(const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ ir,
                                                            const uint8_t* __restrict__ label, int N, int H0, int W0, int dh,
                                                            int dw, int c0, int cw, uint8_t* __restrict__ rgb_out,
                                                            uint16_t* __restrict__ ir_out, uint8_t* __restrict__ label_out) {
#pragma clang fp contract(off)
    const int64_t total = (int64_t)N * dh * cw;
    const bool area2 = (W0 == 2 * dw) && (H0 == 2 * dh);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(idx % cw), yo = (int)((idx / cw) % dh), n = (int)(idx / ((int64_t)cw * dh));
        const int dx = xo + c0;
        const int64_t img = (int64_t)n * H0 * W0;
        if (area2) {
            const int64_t p00 = img + (int64_t)(2 * yo) * W0 + 2 * dx, p10 = p00 + W0;
            if (rgb != nullptr)
                for (int c = 0; c < 3; ++c)
                    rgb_out[idx * 3 + c] = (uint8_t)(((int)rgb[p00 * 3 + c] + rgb[(p00 + 1) * 3 + c] + rgb[p10 * 3 + c] +
                                                      rgb[(p10 + 1) * 3 + c] + 2) >> 2);
            if (ir != nullptr) ir_out[idx] = (uint16_t)(((int)ir[p00] + ir[p00 + 1] + ir[p10] + ir[p10 + 1] + 2) >> 2);
        } else {
            const LinTap tx = lin_tap(dx, dw, W0), ty = lin_tap(yo, dh, H0);
            const int64_t r0 = img + (int64_t)ty.s0 * W0, r1 = img + (int64_t)ty.s1 * W0;
            if (rgb != nullptr) {
                const int a0 = (int)rintf((1.f - tx.frac) * 2048.f), a1 = (int)rintf(tx.frac * 2048.f);
                const int b0 = (int)rintf((1.f - ty.frac) * 2048.f), b1 = (int)rintf(ty.frac * 2048.f);
                for (int c = 0; c < 3; ++c) {
                    const int h0 = rgb[(r0 + tx.s0) * 3 + c] * a0 + rgb[(r0 + tx.s1) * 3 + c] * a1;
                    const int h1 = rgb[(r1 + tx.s0) * 3 + c] * a0 + rgb[(r1 + tx.s1) * 3 + c] * a1;
                    int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    rgb_out[idx * 3 + c] = (uint8_t)min(max(v, 0), 255);
                }
            }
            if (ir != nullptr) {
                const float a0 = 1.f - tx.frac, a1 = tx.frac, b0 = 1.f - ty.frac, b1 = ty.frac;
                const float m00 = (float)ir[r0 + tx.s0] * a0, m01 = (float)ir[r0 + tx.s1] * a1;
                const float m10 = (float)ir[r1 + tx.s0] * a0, m11 = (float)ir[r1 + tx.s1] * a1;
                const float h0 = m00 + m01, h1 = m10 + m11;
                const float v0 = h0 * b0, v1 = h1 * b1;
                const float v = rintf(v0 + v1);
                ir_out[idx] = (uint16_t)fminf(fmaxf(v, 0.f), 65535.f);
            }
        }
        if (label != nullptr) {
            const int sx = min((int)floor((double)dx * ((double)W0 / (double)dw)), W0 - 1);
            const int sy = min((int)floor((double)yo * ((double)H0 / (double)dh)), H0 - 1);
            label_out[idx] = label[img + (int64_t)sy * W0 + sx];
        }
    }
}

}  // namespace

extern "C" int hn_thermal_resize(const uint8_t* rgb, const uint16_t* ir, const uint8_t* label, int N, int H0, int W0, int dh,
                                 int dw, int col0, int col1, uint8_t* rgb_out, uint16_t* ir_out, uint8_t* label_out,
                                 void* stream) {
    if ((rgb == nullptr && ir == nullptr && label == nullptr) || N <= 0 || H0 <= 0 || W0 <= 0 || dh <= 0 || dw <= 0)
        return HN_EINVAL;
    if (col0 < 0 || col1 > dw || col0 >= col1) return HN_ESHAPE;
    if ((rgb != nullptr && rgb_out == nullptr) || (ir != nullptr && ir_out == nullptr) || (label != nullptr && label_out == nullptr))
        return HN_EINVAL;
    const int cw = col1 - col0;
    const int64_t total = (int64_t)N * dh * cw;
    const int64_t blocks = (total + 255) / 256;
    HN_LAUNCH(thermal_resize_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
              static_cast<hipStream_t>(stream), rgb, ir, label, N, H0, W0, dh, dw, col0, cw, rgb_out, ir_out, label_out);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
