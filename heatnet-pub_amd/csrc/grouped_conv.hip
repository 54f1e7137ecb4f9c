// Grouped 3x3 convolution (stride 1, pad = dilation) with few channels per group: the ResNeXt bottleneck's middle convolution
// of the reference's "custom" architecture (cm/models/trgb_segnet.py:106-107: groups = 64, i.e. 2 / 4 / 8 / 16 channels per
// group at 128 / 256 / 512 / 1024 channels).  9 * CG multiply-adds per output element: no matrix-core shape fits a 2..16-wide
// reduction, and the layer is a small part of an optional branch -- direct fp32 kernels, one 16-channel slab (16 / CG groups)
// per block so that every global access is a 64-byte run of one pixel:
//   forward / data gradient: a thread owns one pixel of the slab (16 outputs), the slab's weights sit in LDS;
//   weight gradient: a block stages 64 pixels (dY tile + the nine shifted X tiles) in LDS, a thread owns weight elements.
#include "hn_common.h"

namespace {

struct GArgs {
    const float* x;   // input of the launch (x, or dY for the data gradient) [N][H][W] pixels of ldx floats
    const float* w;   // [C][3][3][CG]
    const float* dy;  // weight gradient only
    float* out;       // y / dx [pixels of ldy floats], or dw
    int N, H, W, C, CG, ldx, ldy, dil;
    int transpose;    // data gradient: taps flipped, in / out roles of the weights swapped
};

// y[p][c0 + g * CG + o] = sum_{tap, i} x[p + tap][c0 + g * CG + i] * w[c0 + g * CG + o][tap][i]
template <int CG>
__global__ __launch_bounds__(256) void grouped_fwd_kernel(const GArgs p) {
    constexpr int GB = 16 / CG;
    __shared__ float wl[9][GB][CG][CG];  // [tap][group][i][o]
    const int c0 = blockIdx.y * 16;
    for (int e = threadIdx.x; e < 9 * GB * CG * CG; e += 256) {
        const int o = e % CG, i = (e / CG) % CG, g = (e / (CG * CG)) % GB, tap = e / (CG * CG * GB);
        // forward: w[co = g * CG + o][tap][i]; data gradient: w[co = g * CG + i][8 - tap][o]
        const int co = c0 + g * CG + (p.transpose ? i : o), ci = p.transpose ? o : i, t = p.transpose ? 8 - tap : tap;
        wl[tap][g][i][o] = p.w[((size_t)co * 9 + t) * CG + ci];
    }
    __syncthreads();
    const long long npix = (long long)p.N * p.H * p.W;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
        // (the weights are re-read from LDS for every pixel: left to itself hipcc hoists all 9 * 16 * CG of them out of this loop
        // -- a thread walks one or two pixels --, 256 registers + 350-630 B of scratch, one wave per SIMD)
        asm volatile("" ::: "memory");
        const int wq = (int)(pix % p.W);
        const long long t = pix / p.W;
        const int hq = (int)(t % p.H);
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hh = hq + (tap / 3 - 1) * p.dil, ww = wq + (tap % 3 - 1) * p.dil;
            if ((unsigned)hh >= (unsigned)p.H || (unsigned)ww >= (unsigned)p.W) continue;
            const float* src = p.x + (pix + (long long)(tap / 3 - 1) * p.dil * p.W + (tap % 3 - 1) * p.dil) * p.ldx + c0;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(src + 4 * q);
                v[4 * q] = f[0]; v[4 * q + 1] = f[1]; v[4 * q + 2] = f[2]; v[4 * q + 3] = f[3];
            }
#pragma unroll
            for (int g = 0; g < GB; ++g)
#pragma unroll
                for (int i = 0; i < CG; ++i)
#pragma unroll
                    for (int o = 0; o < CG; ++o) acc[g * CG + o] += v[g * CG + i] * wl[tap][g][i][o];
        }
        float* dst = p.out + pix * p.ldy + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
}

// dw[c0 + g * CG + o][tap][i] += sum_p dy[p][c0 + g * CG + o] * x[p + tap][c0 + g * CG + i]
template <int CG>
__global__ __launch_bounds__(256) void grouped_wgrad_kernel(const GArgs p) {
    constexpr int GB = 16 / CG, NE = GB * CG * 9 * CG, PER = (NE + 255) / 256, TP = 64;
    __shared__ float gy[TP][16];
    __shared__ float xt[9][TP][16];
    const int c0 = blockIdx.y * 16;
    const long long npix = (long long)p.N * p.H * p.W;
    float acc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] = 0.f;
    for (long long p0 = (long long)blockIdx.x * TP; p0 < npix; p0 += (long long)gridDim.x * TP) {
        __syncthreads();
        for (int e = threadIdx.x; e < TP * 4; e += 256) {  // dY tile: 64 pixels x four 16-byte pieces
            const int px = e >> 2, q = e & 3;
            const long long pix = p0 + px;
            const f32x4 f = pix < npix ? *reinterpret_cast<const f32x4*>(p.dy + pix * p.ldy + c0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&gy[px][4 * q]) = f;
        }
        for (int e = threadIdx.x; e < 9 * TP * 4; e += 256) {  // the nine shifted X tiles (zero outside the image)
            const int q = e & 3, px = (e >> 2) % TP, tap = e / (4 * TP);
            const long long pix = p0 + px;
            f32x4 f = {0.f, 0.f, 0.f, 0.f};
            if (pix < npix) {
                const int wq = (int)(pix % p.W), hq = (int)((pix / p.W) % p.H);
                const int dh = (tap / 3 - 1) * p.dil, dw = (tap % 3 - 1) * p.dil;
                if ((unsigned)(hq + dh) < (unsigned)p.H && (unsigned)(wq + dw) < (unsigned)p.W)
                    f = *reinterpret_cast<const f32x4*>(p.x + (pix + (long long)dh * p.W + dw) * p.ldx + c0 + 4 * q);
            }
            *reinterpret_cast<f32x4*>(&xt[tap][px][4 * q]) = f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < NE) {
                const int i = e % CG, tap = (e / CG) % 9, o = (e / (CG * 9)) % CG, g = e / (CG * 9 * CG);
                float a = 0.f;
                for (int px = 0; px < TP; ++px) a += gy[px][g * CG + o] * xt[tap][px][g * CG + i];
                acc[k] += a;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e < NE) {
            const int i = e % CG, tap = (e / CG) % 9, o = (e / (CG * 9)) % CG, g = e / (CG * 9 * CG);
            atomicAdd(p.out + ((size_t)(c0 + g * CG + o) * 9 + tap) * CG + i, acc[k]);
        }
    }
}

int check(const float* a, const float* b, const float* c, int N, int H, int W, int C, int groups, int ldx, int ldy, int dil) {
    if (!a || !b || !c || N <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || dil <= 0 || ldx < C || ldy < C) return HN_EINVAL;
    if (C % groups != 0) return HN_ESHAPE;
    const int cg = C / groups;
    if (!(cg == 2 || cg == 4 || cg == 8 || cg == 16) || C % 16 != 0 || ldx % 4 != 0 || ldy % 4 != 0) return HN_ESHAPE;
    if (!hn_aligned16(a) || !hn_aligned16(c)) return HN_EALIGN;
    return HN_OK;
}

}  // namespace

#define HN_GROUPED_LAUNCH(kernel, cg, grid, s, a)                                  \
    do {                                                                           \
        if (cg == 2) HN_LAUNCH(kernel<2>, grid, dim3(256), 0, s, a);               \
        else if (cg == 4) HN_LAUNCH(kernel<4>, grid, dim3(256), 0, s, a);          \
        else if (cg == 8) HN_LAUNCH(kernel<8>, grid, dim3(256), 0, s, a);          \
        else HN_LAUNCH(kernel<16>, grid, dim3(256), 0, s, a);                      \
    } while (0)

// y = conv(x, w): x [N][H][W][ldx], w [C][3][3][C / groups], y [N][H][W][ldy]; stride 1, pad = dil
extern "C" int hn_grouped_conv3x3_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int groups, int ldx,
                                      int ldy, int dil, void* stream) {
    const int rc = check(x, w, y, N, H, W, C, groups, ldx, ldy, dil);
    if (rc != HN_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GArgs a{x, w, nullptr, y, N, H, W, C, C / groups, ldx, ldy, dil, 0};
    const long long npix = (long long)N * H * W;
    const long long bx = (npix + 255) / 256;
    const dim3 grid((unsigned)(bx > 4096 ? 4096 : bx), C / 16);
    HN_GROUPED_LAUNCH(grouped_fwd_kernel, a.CG, grid, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// dx = conv_transpose(dy, w): the same kernel with flipped taps and the weights' in / out roles swapped
extern "C" int hn_grouped_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int groups,
                                        int lddy, int lddx, int dil, void* stream) {
    const int rc = check(dy, w, dx, N, H, W, C, groups, lddy, lddx, dil);
    if (rc != HN_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GArgs a{dy, w, nullptr, dx, N, H, W, C, C / groups, lddy, lddx, dil, 1};
    const long long npix = (long long)N * H * W;
    const long long bx = (npix + 255) / 256;
    const dim3 grid((unsigned)(bx > 4096 ? 4096 : bx), C / 16);
    HN_GROUPED_LAUNCH(grouped_fwd_kernel, a.CG, grid, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// dw [C][3][3][C / groups] (+)= sum over pixels; flags as for hn_conv2d_wgrad (float atomics)
extern "C" int hn_grouped_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int groups, int ldx,
                                        int lddy, int dil, int flags, void* stream) {
    const int rc = check(x, dy, dw, N, H, W, C, groups, ldx, lddy, dil);
    if (rc != HN_OK) return rc;
    if (!hn_aligned16(dy)) return HN_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cg = C / groups;
    if (!(flags & (HN_WGRAD_DW_ZEROED | HN_WGRAD_ACCUMULATE)) &&
        hipMemsetAsync(dw, 0, (size_t)C * 9 * cg * sizeof(float), s) != hipSuccess)
        return HN_ELAUNCH;
    GArgs a{x, nullptr, dy, dw, N, H, W, C, cg, ldx, lddy, dil, 0};
    const long long npix = (long long)N * H * W;
    long long bx = (npix + 63) / 64;
    const long long cap = 4096 / (C / 16) > 0 ? 4096 / (C / 16) : 1;  // ~4096 blocks in all
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, C / 16);
    HN_GROUPED_LAUNCH(grouped_wgrad_kernel, a.CG, grid, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
