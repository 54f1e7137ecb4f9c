// Implicit-GEMM convolution forward / data-gradient for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// GEMM view: D[m][n] = sum_k A[m][k] * B[n][k]
//   m = output pixel (n_img, ho, wo), n = output channel, k = (kh, kw, ci).
//   A is gathered on the fly from the NHWC input (never materialised), B is the [Cout][KH][KW][Cin]
//   weight matrix, both K-contiguous, so a lane's MFMA fragment for 4 consecutive k-steps is ONE
//   ds_read_b128: lane l (r = l&31, h = l>>5) reads k = 8*kk + 4*h .. +3 of row r and feeds element j to
//   MFMA j; the two lane halves supply the instruction's two k indices.
//
// Workgroup = 256 threads = 4 waves in a 2x2 grid, each wave TM x TN tiles of 32x32 (block tile
// 64*TM x 64*TN), K step 32, LDS rows padded to 36 floats (144 B: 16-B aligned, and 16 distinct rows
// hit 16 distinct 16-B slots of the 256-B bank row -> conflict-free ds_read_b128).  Operands are staged
// global -> registers -> LDS with the next K step's global loads issued before the current step's MFMAs
// and written to the other LDS buffer after them: one barrier per K step.  2 workgroups per CU.
//
// The same kernel computes the data gradient: dX = conv(dY "dilated" by the forward stride, flipped and
// transposed weights) -- `in_dil` below.
#include "hn_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    double* stats;
    int H, W, C, ldx;        // input tensor
    int Ho, Wo, K, ldy;      // output tensor (K = output channels)
    int KH, KW, stride, pad, dil, in_dil;
    int M, Kdim, ntm, ntn;
    int act;
    float slope;
};

template <int TM, int TN, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                        // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;                         // [2][BN][LDK]
    int* rowinfo = reinterpret_cast<int*>(Bs + 2 * BN * LDK);  // [BM][4]: h base, w base, image pixel base

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = logical % p.ntn, tile_m = logical / p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int hb = -0x40000000, wb = 0, nb = 0;  // rows past M: every tap lands out of range
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hb = ho * p.stride - p.pad;
            wb = wo * p.stride - p.pad;
            nb = n * p.H * p.W;
        }
        rowinfo[r * 4 + 0] = hb;
        rowinfo[r * 4 + 1] = wb;
        rowinfo[r * 4 + 2] = nb;
    }
    __syncthreads();

    // ---- operand staging -------------------------------------------------------------------------------
    constexpr int RA = VEC ? BM / 32 : BM / 8;
    constexpr int RB = VEC ? BN / 32 : BN / 8;
    f32x4 va[VEC ? RA : 1], vb[VEC ? RB : 1];
    float sa[VEC ? 1 : RA], sb[VEC ? 1 : RB];
    int ahb[VEC ? RA : 1], awb[VEC ? RA : 1], anb[VEC ? RA : 1];
    const int c4 = tid & 7, vrow = tid >> 3;    // VEC: 8 lanes x 16 B cover one 128-B row of the K step
    const int sk = tid & 31, srow = tid >> 5;   // scalar: 32 lanes cover one row of the K step
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const int r = vrow + 32 * q;
            ahb[q] = rowinfo[r * 4 + 0];
            awb[q] = rowinfo[r * 4 + 1];
            anb[q] = rowinfo[r * 4 + 2];
        }
    }

    auto tap_ok = [&](int& h, int& w) -> bool {
        bool ok = true;
        if (p.in_dil > 1) {
            ok = (h >= 0) & (w >= 0) & (h % p.in_dil == 0) & (w % p.in_dil == 0);
            h /= p.in_dil;
            w /= p.in_dil;
        }
        return ok & ((unsigned)h < (unsigned)p.H) & ((unsigned)w < (unsigned)p.W);
    };

    auto load_global = [&](int kc) {
        const int k0 = kc * BK;
        if constexpr (VEC) {
            const int tap = k0 / p.C, c0 = k0 - tap * p.C;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                int h = ahb[q] + dh, w = awb[q] + dw;
                const bool ok = tap_ok(h, w);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(anb[q] + h * p.W + w) * p.ldx + c0 + c4 * 4);
                va[q] = v;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int co = n0 + vrow + 32 * q;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (co < p.K) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)co * p.Kdim + k0 + c4 * 4);
                vb[q] = v;
            }
        } else {
            const int kg = k0 + sk;
            const bool kvalid = kg < p.Kdim;
            const int tap = kg / p.C, ci = kg - tap * p.C;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int r = srow + 8 * q;
                int h = rowinfo[r * 4 + 0] + dh, w = rowinfo[r * 4 + 1] + dw;
                const int nb = rowinfo[r * 4 + 2];
                const bool ok = tap_ok(h, w) & kvalid;
                sa[q] = ok ? p.x[(size_t)(nb + h * p.W + w) * p.ldx + ci] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int co = n0 + srow + 8 * q;
                sb[q] = (kvalid && co < p.K) ? p.w[(size_t)co * p.Kdim + kg] : 0.f;
            }
        }
    };

    auto store_lds = [&](int buf) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RA; ++q) *reinterpret_cast<f32x4*>(a + (vrow + 32 * q) * LDK + c4 * 4) = va[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) *reinterpret_cast<f32x4*>(b + (vrow + 32 * q) * LDK + c4 * 4) = vb[q];
        } else {
#pragma unroll
            for (int q = 0; q < RA; ++q) a[(srow + 8 * q) * LDK + sk] = sa[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) b[(srow + 8 * q) * LDK + sk] = sb[q];
        }
    };

    // ---- main loop ---------------------------------------------------------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int nchunks = (p.Kdim + BK - 1) / BK;

    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < nchunks;
        if (more) load_global(kc + 1);

        const float* a = As + buf * BM * LDK + (wm * TM * 32 + lr) * LDK + lh * 4;
        const float* b = Bs + buf * BN * LDK + (wn * TN * 32 + lr) * LDK + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }

        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, batch-statistics partials, activation, store -------------------------------------
    // Accumulator map (32x32 MFMA): column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + lr;
        const bool cok = col < p.K;
        const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const float v = acc[i][j][r] + bv;
                if (row < p.M && cok) {
                    s1 += v;
                    s2 += v * v;
                    p.y[(size_t)row * p.ldy + col] = hn_apply_act(v, p.act, p.slope);
                }
            }
        }
        if (p.stats != nullptr) {
            // <= 64 rows per lane were summed in fp32; everything across tiles is accumulated in double
            // (global_atomic_add_f64), like the double accumulators of the CPU reference's BatchNorm.
            double d1 = (double)s1, d2 = (double)s2;
            d1 += __shfl_xor(d1, 32);
            d2 += __shfl_xor(d2, 32);
            if (lh == 0 && cok) {
                unsafeAtomicAdd(p.stats + col, d1);
                unsafeAtomicAdd(p.stats + p.K + col, d2);
            }
        }
    }
}

template <int TM, int TN, bool VEC>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = (2 * (BM + BN) * LDK + BM * 4) * (int)sizeof(float);
    static bool configured = false;  // raising the dynamic-LDS cap is idempotent; the flag only saves calls
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<TM, TN, VEC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return HN_ELAUNCH;
        configured = true;
    }
    ConvArgs p = a;
    p.ntm = (a.M + BM - 1) / BM;
    p.ntn = (a.K + BN - 1) / BN;
    HN_LAUNCH((conv_igemm_kernel<TM, TN, VEC>), dim3(p.ntm * p.ntn), dim3(256), lds, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

int dispatch_conv(const ConvArgs& a, hipStream_t stream) {
    const bool vec = (a.C % BK == 0) && (a.ldx % 4 == 0) && hn_aligned16(a.x) && hn_aligned16(a.w);
    const bool wide = a.K > 64;
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.K + (wide ? 127 : 63)) / (wide ? 128 : 64));
    const bool tall = blocks128 >= 256;  // enough 128-row tiles to cover the chip; else halve the tile
    if (vec) {
        if (tall) return wide ? launch_conv<2, 2, true>(a, stream) : launch_conv<2, 1, true>(a, stream);
        return wide ? launch_conv<1, 2, true>(a, stream) : launch_conv<1, 1, true>(a, stream);
    }
    if (tall) return wide ? launch_conv<2, 2, false>(a, stream) : launch_conv<2, 1, false>(a, stream);
    return wide ? launch_conv<1, 2, false>(a, stream) : launch_conv<1, 1, false>(a, stream);
}

int check_desc(const HnConvDesc* d) {
    if (d == nullptr) return HN_EINVAL;
    if (d->N <= 0 || d->Hin <= 0 || d->Win <= 0 || d->Cin <= 0 || d->Hout <= 0 || d->Wout <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0)
        return HN_EINVAL;
    if (d->ldx < d->Cin || d->ldy < d->Cout) return HN_EINVAL;
    const int ho = (d->Hin + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->Win + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    if (ho != d->Hout || wo != d->Wout) return HN_ESHAPE;
    if ((long)d->N * d->Hin * d->Win * (long)d->ldx >= (1L << 31) * 4 || (long)d->N * d->Hout * d->Wout >= (1L << 31))
        return HN_ESHAPE;
    return HN_OK;
}

// wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci], one 32x32 (co, ci) tile per block through LDS.
__global__ __launch_bounds__(256) void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                   int Cout, int T, int Cin) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * T + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
    const int ftap = T - 1 - tap;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * T + ftap) * Cout + co] = tile[tx][r];
    }
}

}  // namespace

extern "C" int hn_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, double* stats,
                             const HnConvDesc* d, int act, float slope, void* stream) {
    if (x == nullptr || w == nullptr || y == nullptr) return HN_EINVAL;
    if (act < HN_ACT_NONE || act > HN_ACT_LEAKY) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
    a.H = d->Hin; a.W = d->Win; a.C = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.K = d->Cout; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.in_dil = 1;
    a.M = d->N * d->Hout * d->Wout;
    a.Kdim = d->KH * d->KW * d->Cin;
    a.ntm = a.ntn = 0;
    a.act = act; a.slope = slope;
    return dispatch_conv(a, static_cast<hipStream_t>(stream));
}

extern "C" int hn_conv2d_dgrad(const float* dy, const float* wt, float* dx, const HnConvDesc* d, void* stream) {
    if (dy == nullptr || wt == nullptr || dx == nullptr) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    if (d->dil * (d->KH - 1) < d->pad || d->dil * (d->KW - 1) < d->pad) return HN_ESHAPE;
    ConvArgs a;
    a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx; a.stats = nullptr;
    a.H = d->Hout; a.W = d->Wout; a.C = d->Cout; a.ldx = d->ldy;      // gathered tensor: dY
    a.Ho = d->Hin; a.Wo = d->Win; a.K = d->Cin; a.ldy = d->ldx;        // produced tensor: dX
    a.KH = d->KH; a.KW = d->KW; a.stride = 1; a.dil = d->dil; a.in_dil = d->stride;
    a.pad = d->dil * (d->KH - 1) - d->pad;
    a.M = d->N * d->Hin * d->Win;
    a.Kdim = d->KH * d->KW * d->Cout;
    a.ntm = a.ntn = 0;
    a.act = HN_ACT_NONE; a.slope = 0.f;
    // the gather compares against the DILATED extent of dY
    a.H = d->Hout; a.W = d->Wout;
    return dispatch_conv(a, static_cast<hipStream_t>(stream));
}

extern "C" int hn_weight_flip_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    if (w == nullptr || wt == nullptr || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0) return HN_EINVAL;
    const dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, KH * KW);
    HN_LAUNCH(weight_flip_transpose_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, wt, Cout,
                       KH * KW, Cin);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
