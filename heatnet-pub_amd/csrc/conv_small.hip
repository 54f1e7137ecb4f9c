// Convolutions with a tiny channel count on one side: the 7x7 / stride-2 stems (3, 1 or 4 input channels -> 64,
// cm/models/extractors.py:108-126) and the 13-class head (1x1, 64 -> 13, cm/models/pspnet.py:56-58).
//
// On the implicit-GEMM kernels these ran on the scalar loaders with 128-wide tiles of which 3/128 or 13/128 were used:
// 8.6-60 TFLOP/s and 0.1-0.35 ms per launch against 0.03-0.11 ms of HBM time (profiles/r02_conv_table_f16x2.txt).  They are
// HBM-bound streams with a little exact-fp32 matrix work per pixel, so they get kernels of their own on
// v_mfma_f32_16x16x4_f32 (exact fp32: the same product / accumulation quality as the fp32 implicit-GEMM kernels):
//   lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15], and holds D[4 (l >> 4) + r][l & 15], r = 0..3.
// The trick that makes every global access a full 256-byte pixel row: a lane loads / stores FOUR consecutive channels of one
// pixel as a float4 and component j feeds MFMA number j, whose 16 rows (or columns) then stand for the channels 4 n + j --
// four MFMAs cover 64 channels, and the four results a lane holds for one pixel are again 4 consecutive channels.
#include <cstdio>

#include "hn_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct SmallArgs {
    const float* x;    // input  [N][H][W] pixels of ldx floats
    const float* w;    // [Cout][KH][KW][Cin]
    const float* dy;   // output-side tensor (y or dy) [N][Ho][Wo] pixels of ldy floats
    float* out;        // y, dx or dw
    double* stats;     // stem forward: [HN_STAT_REPLICAS][2][64] or NULL
    int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy;
    long long M;       // N * Ho * Wo
    int atomic;        // weight gradients: 1 = accumulate with atomics (dw zeroed or accumulate), always 1 here
};

// ---- head, data gradient: dx[p][64] = sum_co dy[p][co] * w[co][64], Cout <= 16 -------------------------------------------
__global__ __launch_bounds__(256) void narrow_dgrad_kernel(const SmallArgs p) {
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    float bw[4][4];  // B_{s, j}[k = kk][n] = w[co = 4 s + kk][ci = 4 n + j]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int co = 4 * s + kk;
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[s][j] = co < p.Cout ? p.w[co * 64 + 4 * n + j] : 0.f;
    }
    const long long groups = (p.M + 15) / 16;
    for (long long g = wave; g < groups; g += nwaves) {
        const long long pa = g * 16 + n;  // this lane's A row (pixel)
        float a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int co = 4 * s + kk;
            a[s] = (pa < p.M && co < p.Cout) ? p.dy[pa * p.ldy + co] : 0.f;
        }
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = mfma4(a[s], bw[s][j], d[j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long po = g * 16 + 4 * kk + r;
            if (po < p.M) *reinterpret_cast<f32x4*>(p.out + po * p.ldx + 4 * n) = f32x4{d[0][r], d[1][r], d[2][r], d[3][r]};
        }
    }
}

// ---- head, weight gradient: dw[co][64] += sum_p dy[p][co] * x[p][64], Cout <= 16 -----------------------------------------
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const SmallArgs p) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4, wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
    f32x4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long groups = (p.M + 3) / 4;
    constexpr int U = 4;  // groups in flight per wave
    for (long long g0 = wave * U; g0 < groups; g0 += nwaves * U) {
        f32x4 xv[U];
        float gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pix = (g0 + u) * 4 + kk;
            const bool ok = pix < p.M;
            xv[u] = ok ? *reinterpret_cast<const f32x4*>(p.x + pix * p.ldx + 4 * n) : f32x4{0.f, 0.f, 0.f, 0.f};
            gv[u] = (ok && n < p.Cout) ? p.dy[pix * p.ldy + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = mfma4(xv[u][j], gv[u], d[j]);  // D_j[m][co] += x[p][4 m + j] * dy[p][co]
    }
    // D_j[row = 4 kk + r][col = n] = dw[co = n][ci = 4 (4 kk + r) + j]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][n][4 * (4 * kk + r) + j] = d[j][r];
    __syncthreads();
    for (int i = threadIdx.x; i < p.Cout * 64; i += 256) {
        const int co = i >> 6, ci = i & 63;
        const float v = red[0][co][ci] + red[1][co][ci] + red[2][co][ci] + red[3][co][ci];
        atomicAdd(p.out + co * 64 + ci, v);
    }
}

// ---- stems: 7x7, stride 2, pad 3, CIN in {1, 3, 4} -> 64 -------------------------------------------------------------------
template <int CIN>
struct Stem {
    static constexpr int K = 49 * CIN, KS = (K + 3) / 4, KB = (K + 15) / 16;  // K steps of 4 (forward), K blocks of 16 (wgrad)
};

// element offset of reduction index k = (kh, kw, c) relative to the window's top-left input pixel, and its (kh, kw)
template <int CIN>
__device__ __forceinline__ void stem_k(int k, int W, int& off, int& kh, int& kw) {
    const int tap = k / CIN, c = k - tap * CIN;
    kh = tap / 7;
    kw = tap - kh * 7;
    off = (kh * W + kw) * CIN + c;
}

// Output tile of a block: TH x TW pixels; the (2 TH + 5) x (2 TW + 5) x CIN input window is staged in LDS once (coalesced row
// segments, zero outside the image) and all 49 taps gather from there -- the first version gathered from global memory (37
// scattered 4-byte loads per 16 pixels) and was slower than the generic kernels.
constexpr int STEM_TH = 4, STEM_TW = 64, STEM_PH = 2 * STEM_TH + 5, STEM_PW = 2 * STEM_TW + 5;

template <int CIN>
__device__ __forceinline__ void stem_load_patch(const SmallArgs& p, float* patch, int img, int ho0, int wo0) {
    const int hb = 2 * ho0 - 3, wb = 2 * wo0 - 3;
    for (int e = threadIdx.x; e < STEM_PH * STEM_PW * CIN; e += 256) {
        const int row = e / (STEM_PW * CIN), rem = e - row * (STEM_PW * CIN);
        const int col = rem / CIN;
        const int h = hb + row, w = wb + col;
        float v = 0.f;
        if ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
            v = p.x[((long long)(img * p.H + h) * p.W + wb) * CIN + rem];  // (row segment: contiguous in rem)
        patch[e] = v;
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const SmallArgs p) {
    using S = Stem<CIN>;
    __shared__ float wl[S::KS * 4][64];                 // w transposed: [k][co], zero rows past K
    __shared__ float patch[STEM_PH * STEM_PW * CIN];
    __shared__ int ktab[S::KS * 4];                     // k -> offset inside the patch of tap k for output pixel (0, 0); past K: 0
    for (int i = threadIdx.x; i < S::KS * 4 * 64; i += 256) {
        const int k = i >> 6, co = i & 63;
        wl[k][co] = k < S::K ? p.w[co * S::K + k] : 0.f;
    }
    for (int k = threadIdx.x; k < S::KS * 4; k += 256) {
        int off, kh, kw;
        stem_k<CIN>(k, STEM_PW, off, kh, kw);
        ktab[k] = k < S::K ? off : 0;  // (past K the weight row is zero: any in-range patch element will do)
    }
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4, wv = threadIdx.x >> 6;
    const int tiles_w = (p.Wo + STEM_TW - 1) / STEM_TW, tiles_h = (p.Ho + STEM_TH - 1) / STEM_TH;
    const int ntiles = p.N * tiles_h * tiles_w;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, img = tile / (tiles_w * tiles_h);
        const int ho0 = th * STEM_TH, wo0 = tw * STEM_TW;
        __syncthreads();  // (previous tile's readers are done with the patch; first pass: the tables are written)
        stem_load_patch<CIN>(p, patch, img, ho0, wo0);
        __syncthreads();
        // wave wv: output row ho0 + wv, four groups of 16 consecutive columns
        const int ho = ho0 + wv;
        for (int grp = 0; grp < STEM_TW / 16; ++grp) {
            const int wo_a = wo0 + 16 * grp + n;                       // this lane's A row (pixel)
            const int pbase = (2 * wv * STEM_PW + 2 * (16 * grp + n)) * CIN;
            f32x4 d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int s = 0; s < S::KS; ++s) {
                const int k = 4 * s + kk;
                const float a = patch[pbase + ktab[k]];
                const f32x4 b = *reinterpret_cast<const f32x4*>(&wl[k][4 * n]);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = mfma4(a, b[j], d[j]);
            }
            (void)wo_a;
            float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wo = wo0 + 16 * grp + 4 * kk + r;
                if (ho < p.Ho && wo < p.Wo) {
                    const long long po = ((long long)img * p.Ho + ho) * p.Wo + wo;
                    *reinterpret_cast<f32x4*>(p.out + po * p.ldy + 4 * n) = f32x4{d[0][r], d[1][r], d[2][r], d[3][r]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        t1[j] += d[j][r];
                        t2[j] += d[j][r] * d[j][r];
                    }
                }
            }
            if (p.stats != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a1 = t1[j], a2 = t2[j];
                    a1 += __shfl_xor(a1, 16);
                    a2 += __shfl_xor(a2, 16);
                    a1 += __shfl_xor(a1, 32);
                    a2 += __shfl_xor(a2, 32);
                    s1[j] += (double)a1;  // (16 pixels summed in fp32, everything beyond in double, as the implicit-GEMM epilogue)
                    s2[j] += (double)a2;
                }
            }
        }
    }
    if (p.stats != nullptr && kk == 0) {
        double* rep = p.stats + (size_t)(blockIdx.x % HN_STAT_REPLICAS) * 2 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsafeAtomicAdd(rep + 4 * n + j, s1[j]);
            unsafeAtomicAdd(rep + 64 + 4 * n + j, s2[j]);
        }
    }
}

// dw[co][k] += sum_p dy[p][co] * patch[p][k]; a block walks output tiles (accumulators persist), one reduction at the end
template <int CIN>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const SmallArgs p) {
    using S = Stem<CIN>;
    constexpr int KP = S::KB * 16;
    extern __shared__ float dyn[];  // patch [PH * PW * CIN] | red [64][KP]
    float* patch = dyn;
    float* red = dyn + STEM_PH * STEM_PW * CIN;
    for (int i = threadIdx.x; i < 64 * KP; i += 256) red[i] = 0.f;
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4, wv = threadIdx.x >> 6;
    int koff[S::KB];  // this lane's reduction index m = 16 b + n of every K block: offset inside the patch (past K: 0, masked below)
    bool kval[S::KB];
#pragma unroll
    for (int b = 0; b < S::KB; ++b) {
        int kh, kw;
        stem_k<CIN>(16 * b + n, STEM_PW, koff[b], kh, kw);
        kval[b] = 16 * b + n < S::K;
        if (!kval[b]) koff[b] = 0;
    }
    f32x4 d[S::KB][4];
#pragma unroll
    for (int b = 0; b < S::KB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tiles_w = (p.Wo + STEM_TW - 1) / STEM_TW, tiles_h = (p.Ho + STEM_TH - 1) / STEM_TH;
    const int ntiles = p.N * tiles_h * tiles_w;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, img = tile / (tiles_w * tiles_h);
        const int ho0 = th * STEM_TH, wo0 = tw * STEM_TW;
        __syncthreads();
        stem_load_patch<CIN>(p, patch, img, ho0, wo0);
        __syncthreads();
        const int ho = ho0 + wv;  // wave wv: output row ho0 + wv, 16 groups of 4 consecutive columns
        // dy of a group is needed by its first MFMA: fetched PF groups ahead (a load in front of every group left the matrix
        // pipe idle for the whole memory latency, 16 times per tile: the kernel ran at a quarter of its arithmetic time)
        constexpr int PF = 4, NG = STEM_TW / 4;
        auto load_dy = [&](int grp) {
            const int wo = wo0 + 4 * grp + kk;
            const bool pok = grp < NG && ho < p.Ho && wo < p.Wo;
            const long long pix = ((long long)img * p.Ho + (ho < p.Ho ? ho : 0)) * p.Wo + (pok ? wo : 0);
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.dy + pix * p.ldy + 4 * n);  // (unconditional, clamped: no queue drain)
            return pok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        f32x4 ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = load_dy(i);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const f32x4 gv = ring[grp % PF];
            ring[grp % PF] = load_dy(grp + PF);
            const int pbase = (2 * wv * STEM_PW + 2 * (4 * grp + kk)) * CIN;
#pragma unroll
            for (int b = 0; b < S::KB; ++b) {
                const float a = kval[b] ? patch[pbase + koff[b]] : 0.f;  // (dy is zero for pixels outside the image)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[b][j] = mfma4(a, gv[j], d[b][j]);  // D_{b,j}[m][n] += patch[p][16 b + m] * dy[p][4 n + j]
            }
        }
    }
    // D_{b,j}[row = 4 kk + r][col = n] = dw[co = 4 n + j][k = 16 b + 4 kk + r]: the block's four waves meet in LDS
    __syncthreads();
#pragma unroll
    for (int b = 0; b < S::KB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[(4 * n + j) * KP + 16 * b + 4 * kk + r], d[b][j][r]);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * S::K; i += 256) {
        const int co = i / S::K, k = i - co * S::K;
        atomicAdd(p.out + i, red[co * KP + k]);
    }
}

int grid_for(long long groups, int per_wave) {
    long long blocks = (groups + 4LL * per_wave - 1) / (4LL * per_wave);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

bool stem_geometry(const HnConvDesc* d) {
    return d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->dil == 1 && d->Cout == 64 &&
           (d->Cin == 1 || d->Cin == 3) && d->ldx == d->Cin && d->ldy % 4 == 0 &&
           d->Hout == (d->Hin - 1) / 2 + 1 && d->Wout == (d->Win - 1) / 2 + 1 &&
           (long long)d->N * d->Hin * d->Win * d->Cin < (1LL << 31) && (6 * d->Win + 6) * d->Cin + d->Cin < (1 << 24);
}

bool narrow_geometry(const HnConvDesc* d) {
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->dil == 1 && d->Cin == 64 && d->Cout >= 1 &&
           d->Cout <= 16 && d->ldx % 4 == 0 && d->Hout == d->Hin && d->Wout == d->Win;
}

SmallArgs small_args(const float* x, const float* w, const float* dy, float* out, double* stats, const HnConvDesc* d) {
    SmallArgs a;
    a.x = x; a.w = w; a.dy = dy; a.out = out; a.stats = stats;
    a.N = d->N; a.H = d->Hin; a.W = d->Win; a.Cin = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.Cout = d->Cout; a.ldy = d->ldy;
    a.M = (long long)d->N * d->Hout * d->Wout;
    a.atomic = 1;
    return a;
}

}  // namespace

extern "C" int hn_conv_stem_fwd(const float* x, const float* w, float* y, double* stats, const HnConvDesc* d, void* stream) {
    if (!x || !w || !y || !d) return HN_EINVAL;
    if (!stem_geometry(d)) return HN_ESHAPE;
    if (!hn_aligned16(y)) return HN_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SmallArgs a = small_args(x, w, y, y, stats, d);
    const int ntiles = d->N * ((d->Hout + STEM_TH - 1) / STEM_TH) * ((d->Wout + STEM_TW - 1) / STEM_TW);
    const dim3 grid(ntiles < 2048 ? ntiles : 2048);
    if (d->Cin == 1) HN_LAUNCH(stem_fwd_kernel<1>, grid, dim3(256), 0, s, a);
    else HN_LAUNCH(stem_fwd_kernel<3>, grid, dim3(256), 0, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_conv_stem_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int flags, void* stream) {
    if (!x || !dy || !dw || !d) return HN_EINVAL;
    if (!stem_geometry(d)) return HN_ESHAPE;
    if (!hn_aligned16(dy)) return HN_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int K = 49 * d->Cin;
    if (!(flags & (HN_WGRAD_DW_ZEROED | HN_WGRAD_ACCUMULATE)) &&
        hipMemsetAsync(dw, 0, (size_t)64 * K * sizeof(float), s) != hipSuccess)
        return HN_ELAUNCH;
    const SmallArgs a = small_args(x, nullptr, dy, dw, nullptr, d);
    const int ntiles = d->N * ((d->Hout + STEM_TH - 1) / STEM_TH) * ((d->Wout + STEM_TW - 1) / STEM_TW);
    const dim3 grid(ntiles < 512 ? ntiles : 512);  // two blocks per CU; every block ends with 64 * K global atomics
    const int kp = ((K + 15) / 16) * 16;
    const int lds = (64 * kp + STEM_PH * STEM_PW * d->Cin) * (int)sizeof(float);
    if (d->Cin == 1) HN_LAUNCH(stem_wgrad_kernel<1>, grid, dim3(256), lds, s, a);
    else HN_LAUNCH(stem_wgrad_kernel<3>, grid, dim3(256), lds, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_conv_narrow_dgrad(const float* dy, const float* w, float* dx, const HnConvDesc* d, void* stream) {
    if (!dy || !w || !dx || !d) return HN_EINVAL;
    if (!narrow_geometry(d)) return HN_ESHAPE;
    if (!hn_aligned16(dx)) return HN_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SmallArgs a = small_args(nullptr, w, dy, dx, nullptr, d);
    HN_LAUNCH(narrow_dgrad_kernel, dim3(grid_for((a.M + 15) / 16, 8)), dim3(256), 0, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_conv_narrow_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int flags, void* stream) {
    if (!x || !dy || !dw || !d) return HN_EINVAL;
    if (!narrow_geometry(d)) return HN_ESHAPE;
    if (!hn_aligned16(x)) return HN_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!(flags & (HN_WGRAD_DW_ZEROED | HN_WGRAD_ACCUMULATE)) &&
        hipMemsetAsync(dw, 0, (size_t)d->Cout * 64 * sizeof(float), s) != hipSuccess)
        return HN_ELAUNCH;
    const SmallArgs a = small_args(x, nullptr, dy, dw, nullptr, d);
    HN_LAUNCH(narrow_wgrad_kernel, dim3(grid_for((a.M + 3) / 4, 64)), dim3(256), 0, s, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
