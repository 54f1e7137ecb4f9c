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

template <int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const SmallArgs p) {
    using S = Stem<CIN>;
    __shared__ float wl[S::KS * 4][64];  // w transposed: [k][co], zero rows past K
    __shared__ int ktab[S::KS * 4];      // k -> element offset in the window (24 bits) | kh << 24 | kw << 28; past K: -1
    for (int i = threadIdx.x; i < S::KS * 4 * 64; i += 256) {
        const int k = i >> 6, co = i & 63;
        wl[k][co] = k < S::K ? p.w[co * S::K + k] : 0.f;
    }
    for (int k = threadIdx.x; k < S::KS * 4; k += 256) {
        int off, kh, kw;
        stem_k<CIN>(k, p.W, off, kh, kw);
        ktab[k] = k < S::K ? (off | (kh << 24) | (kw << 28)) : -1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long groups = (p.M + 15) / 16;
    const int hw = p.Ho * p.Wo;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};  // channels 4 n + j (lanes < 16 after the reduction)
    for (long long g = wave; g < groups; g += nwaves) {
        const long long pa = g * 16 + n;
        const bool pok = pa < p.M;
        const int img = pok ? (int)(pa / hw) : 0, rem = pok ? (int)(pa - (long long)img * hw) : 0;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hb = 2 * ho - 3, wb = 2 * wo - 3;
        const long long base = ((long long)(img * p.H + hb) * p.W + wb) * CIN;
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // K steps in chunks of CH: the chunk's CH gathers are issued together (CH loads in flight per wave), then its MFMAs run
        constexpr int CH = 8;
#pragma unroll 1
        for (int s0 = 0; s0 < S::KS; s0 += CH) {
            float a[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int k = 4 * (s0 + u) + kk;  // (ktab / wl have KS * 4 rows; chunks past KS are guarded below)
                // volatile LDS reads: the tables are loop-invariant across pixel groups and hipcc would otherwise keep all of
                // them -- 150+ registers -- live, leaving one wave per SIMD to hide the gather latency
                const int e = s0 + u < S::KS ? *reinterpret_cast<volatile const int*>(&ktab[k]) : -1;
                const int off = e & 0xFFFFFF, kh = (e >> 24) & 15, kw = (e >> 28) & 7;
                // (e < 0: a reduction index past K -- no tap: never load; its weight row is zero anyway)
                const bool ok = pok && e >= 0 && (unsigned)(hb + kh) < (unsigned)p.H && (unsigned)(wb + kw) < (unsigned)p.W;
                a[u] = ok ? p.x[base + off] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (s0 + u < S::KS) {
                    const volatile float* wr = &wl[4 * (s0 + u) + kk][4 * n];
                    const float b[4] = {wr[0], wr[1], wr[2], wr[3]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = mfma4(a[u], b[j], d[j]);
                }
            }
        }
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long po = g * 16 + 4 * kk + r;
            if (po < p.M) {
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
    if (p.stats != nullptr && kk == 0) {
        double* rep = p.stats + (size_t)(blockIdx.x % HN_STAT_REPLICAS) * 2 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsafeAtomicAdd(rep + 4 * n + j, s1[j]);
            unsafeAtomicAdd(rep + 64 + 4 * n + j, s2[j]);
        }
    }
}

// dw[co][k] += sum_p dy[p][co] * patch[p][k]
template <int CIN>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const SmallArgs p) {
    using S = Stem<CIN>;
    extern __shared__ float red_dyn[];  // [64][KB * 16]
    constexpr int KP = S::KB * 16;
    for (int i = threadIdx.x; i < 64 * KP; i += 256) red_dyn[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long groups = (p.M + 3) / 4;
    const int hw = p.Ho * p.Wo;
    int koff[S::KB], kkh[S::KB], kkw[S::KB];  // this lane's reduction index m = 16 b + n of every K block
#pragma unroll
    for (int b = 0; b < S::KB; ++b) {
        stem_k<CIN>(16 * b + n, p.W, koff[b], kkh[b], kkw[b]);
        if (16 * b + n >= S::K) kkh[b] = 1 << 20;  // past the end: never valid
    }
    f32x4 d[S::KB][4];
#pragma unroll
    for (int b = 0; b < S::KB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long g = wave; g < groups; g += nwaves) {
        const long long pix = g * 4 + kk;
        const bool pok = pix < p.M;
        const int img = pok ? (int)(pix / hw) : 0, rem = pok ? (int)(pix - (long long)img * hw) : 0;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hb = 2 * ho - 3, wb = 2 * wo - 3;
        const long long base = ((long long)(img * p.H + hb) * p.W + wb) * CIN;
        const f32x4 gv = pok ? *reinterpret_cast<const f32x4*>(p.dy + pix * p.ldy + 4 * n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < S::KB; ++b) {
            const bool ok = pok && (unsigned)(hb + kkh[b]) < (unsigned)p.H && (unsigned)(wb + kkw[b]) < (unsigned)p.W;
            const float a = ok ? p.x[base + koff[b]] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[b][j] = mfma4(a, gv[j], d[b][j]);  // D_{b,j}[m][n] += patch[p][16 b + m] * dy[p][4 n + j]
        }
    }
    // D_{b,j}[row = 4 kk + r][col = n] = dw[co = 4 n + j][k = 16 b + 4 kk + r]: the block's four waves meet in LDS
#pragma unroll
    for (int b = 0; b < S::KB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red_dyn[(4 * n + j) * KP + 16 * b + 4 * kk + r], d[b][j][r]);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * S::K; i += 256) {
        const int co = i / S::K, k = i - co * S::K;
        atomicAdd(p.out + i, red_dyn[co * KP + k]);
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
           (d->Cin == 1 || d->Cin == 3 || d->Cin == 4) && d->ldx == d->Cin && d->ldy % 4 == 0 &&
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
    const dim3 grid(grid_for((a.M + 15) / 16, 8));
    if (d->Cin == 1) HN_LAUNCH(stem_fwd_kernel<1>, grid, dim3(256), 0, s, a);
    else if (d->Cin == 3) HN_LAUNCH(stem_fwd_kernel<3>, grid, dim3(256), 0, s, a);
    else HN_LAUNCH(stem_fwd_kernel<4>, grid, dim3(256), 0, s, a);
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
    const dim3 grid(grid_for((a.M + 3) / 4, 64));
    const int kp = ((K + 15) / 16) * 16;
    const int lds = 64 * kp * (int)sizeof(float);
    if (d->Cin == 1) HN_LAUNCH(stem_wgrad_kernel<1>, grid, dim3(256), lds, s, a);
    else if (d->Cin == 3) HN_LAUNCH(stem_wgrad_kernel<3>, grid, dim3(256), lds, s, a);
    else HN_LAUNCH(stem_wgrad_kernel<4>, grid, dim3(256), lds, s, a);
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
