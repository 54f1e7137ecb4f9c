// Convolution weight gradient for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// GEMM view: dW[co][(kh,kw,ci)] = sum over output pixels p of dY[p][co] * X[tap-shifted p][ci]
//   m = output channel, n = (tap, ci), reduction over pixels (28 k ... 1.8 M of them), so the output is
//   small and the reduction is split over blockIdx.z; partial tiles meet in dW through
//   global_atomic_add_f32 (dW zeroed first), or are stored directly when there is a single split.
// Both operands are pixel-major in memory (NHWC), i.e. the reduction index is the slow one: tiles are
// staged as [32 pixels][BM or BN channels] and a lane's MFMA operand is one ds_read_b32 -- lanes 0..31
// read 32 consecutive channels of pixel row 2*kk, lanes 32..63 of row 2*kk+1 (conflict-free).
#include "hn_common.h"

namespace {

constexpr int BKP = 32;  // pixels per reduction step

struct WgradArgs {
    const float* x;
    const float* dy;
    float* dw;
    int H, W, C, ldx;
    int Ho, Wo, K, ldy;
    int KH, KW, stride, pad, dil;
    int M;        // N*Ho*Wo reduction length
    int Kdim;     // KH*KW*C
    int ntm, ntn; // tiles over Cout and over (tap, ci)
    int chunks_per_split;
    int atomic;
};

template <int TM, int TN, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BKP][BM]  dY tile
    float* Bs = smem + 2 * BKP * BM;    // [2][BKP][BN]  gathered X tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = p.ntm * p.ntn;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tile_m = tile % p.ntm, tile_n = tile / p.ntm;
    const int co0 = tile_m * BM, n0 = tile_n * BN;
    const int total_chunks = (p.M + BKP - 1) / BKP;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(total_chunks, c_begin + p.chunks_per_split);
    const int hw = p.Ho * p.Wo;

    // VEC: a tile column range lies inside one tap (C % BN == 0)
    constexpr int CA = BM / 4, CB = BN / 4;               // 16-B pieces per pixel row
    constexpr int RA = VEC ? (BKP * CA) / 256 : (BKP * BM) / 256;
    constexpr int RB = VEC ? (BKP * CB) / 256 : (BKP * BN) / 256;
    f32x4 va[VEC ? RA : 1], vb[VEC ? RB : 1];
    float sa[VEC ? 1 : RA], sb[VEC ? 1 : RB];

    int tap = 0, ci0 = 0;
    if constexpr (VEC) {
        tap = n0 / p.C;
        ci0 = n0 - tap * p.C;
    }
    const int vkh = tap / p.KW, vkw = tap - vkh * p.KW;

    // scalar path: this thread's fixed column of each tile
    const int a_col = tid % BM, a_row0 = tid / BM;      // rows a_row0 + (256/BM)*q
    const int b_col = tid % BN, b_row0 = tid / BN;
    int s_kh = 0, s_kw = 0, s_ci = 0;
    bool s_nvalid = false;
    if constexpr (!VEC) {
        const int n = n0 + b_col;
        s_nvalid = n < p.Kdim;
        const int t = n / p.C;
        s_ci = n - t * p.C;
        s_kh = t / p.KW;
        s_kw = t - s_kh * p.KW;
    }

    auto gather_pixel = [&](int pix, int kh, int kw, size_t& off) -> bool {
        if (pix >= p.M) return false;
        const int n = pix / hw, rem = pix - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int h = ho * p.stride - p.pad + kh * p.dil, w = wo * p.stride - p.pad + kw * p.dil;
        if ((unsigned)h >= (unsigned)p.H || (unsigned)w >= (unsigned)p.W) return false;
        off = (size_t)((n * p.H + h) * p.W + w) * p.ldx;
        return true;
    };

    auto load_global = [&](int chunk) {
        const int p0 = chunk * BKP;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int piece = tid + 256 * q;
                const int row = piece / CA, c4 = piece - row * CA;
                const int pix = p0 + row, co = co0 + c4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (pix < p.M && co < p.K) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)pix * p.ldy + co);
                va[q] = v;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int piece = tid + 256 * q;
                const int row = piece / CB, c4 = piece - row * CB;
                size_t off = 0;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gather_pixel(p0 + row, vkh, vkw, off)) v = *reinterpret_cast<const f32x4*>(p.x + off + ci0 + c4 * 4);
                vb[q] = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int pix = p0 + a_row0 + (256 / BM) * q, co = co0 + a_col;
                sa[q] = (pix < p.M && co < p.K) ? p.dy[(size_t)pix * p.ldy + co] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                size_t off = 0;
                const bool ok = s_nvalid && gather_pixel(p0 + b_row0 + (256 / BN) * q, s_kh, s_kw, off);
                sb[q] = ok ? p.x[off + s_ci] : 0.f;
            }
        }
    };

    auto store_lds = [&](int buf) {
        float* a = As + buf * BKP * BM;
        float* b = Bs + buf * BKP * BN;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RA; ++q) *reinterpret_cast<f32x4*>(a + (tid + 256 * q) * 4) = va[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) *reinterpret_cast<f32x4*>(b + (tid + 256 * q) * 4) = vb[q];
        } else {
#pragma unroll
            for (int q = 0; q < RA; ++q) a[(a_row0 + (256 / BM) * q) * BM + a_col] = sa[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) b[(b_row0 + (256 / BN) * q) * BN + b_col] = sb[q];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;

    if (c_begin < c_end) {
        load_global(c_begin);
        store_lds(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            const bool more = c + 1 < c_end;
            if (more) load_global(c + 1);
            const float* a = As + buf * BKP * BM + lh * BM + wm * TM * 32 + lr;
            const float* b = Bs + buf * BKP * BN + lh * BN + wn * TN * 32 + lr;
#pragma unroll
            for (int kk = 0; kk < BKP / 2; ++kk) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = a[kk * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = b[kk * 2 * BN + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }

    // D row = output channel, D column = (tap, ci): column on the lane -> 128-B runs of dW per store.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lr;
        if (n >= p.Kdim) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = co0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = rbase + (r & 3) + 8 * (r >> 2);
                if (co < p.K) {
                    float* dst = p.dw + (size_t)co * p.Kdim + n;
                    if (p.atomic) atomicAdd(dst, acc[i][j][r]);
                    else *dst = acc[i][j][r];
                }
            }
        }
    }
}

template <int TM, int TN, bool VEC>
int launch_wgrad(WgradArgs a, int split_k, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = 2 * BKP * (BM + BN) * (int)sizeof(float);
    a.ntm = (a.K + BM - 1) / BM;
    a.ntn = (a.Kdim + BN - 1) / BN;
    const int tiles = a.ntm * a.ntn;
    const int total_chunks = (a.M + BKP - 1) / BKP;
    if (split_k <= 0) {
        // aim at ~3 blocks per CU, keep at least 8 reduction steps per split
        split_k = (768 + tiles - 1) / tiles;
        const int cap = (total_chunks + 7) / 8;
        if (split_k > cap) split_k = cap;
        if (split_k < 1) split_k = 1;
    }
    if (split_k > total_chunks) split_k = total_chunks;
    a.chunks_per_split = (total_chunks + split_k - 1) / split_k;
    split_k = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.atomic = split_k > 1;
    if (a.atomic) {
        if (hipMemsetAsync(a.dw, 0, (size_t)a.K * a.Kdim * sizeof(float), stream) != hipSuccess) return HN_ELAUNCH;
    }
    HN_LAUNCH((conv_wgrad_kernel<TM, TN, VEC>), dim3(tiles * split_k), dim3(256), lds, stream, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

}  // namespace

extern "C" int hn_conv2d_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int split_k,
                               void* stream) {
    if (x == nullptr || dy == nullptr || dw == nullptr || d == nullptr) return HN_EINVAL;
    if (d->N <= 0 || d->Hin <= 0 || d->Win <= 0 || d->Cin <= 0 || d->Hout <= 0 || d->Wout <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->ldx < d->Cin || d->ldy < d->Cout)
        return HN_EINVAL;
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.H = d->Hin; a.W = d->Win; a.C = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.K = d->Cout; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.M = d->N * d->Hout * d->Wout;
    a.Kdim = d->KH * d->KW * d->Cin;
    a.ntm = a.ntn = a.chunks_per_split = a.atomic = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool aligned = (d->ldx % 4 == 0) && (d->ldy % 4 == 0) && hn_aligned16(x) && hn_aligned16(dy);
    const bool wide_m = d->Cout > 64;
    if (aligned && d->Cout % 4 == 0 && d->Cin % 64 == 0) {
        const bool wide_n = d->Cin % 128 == 0;
        if (wide_m) return wide_n ? launch_wgrad<2, 2, true>(a, split_k, s) : launch_wgrad<2, 1, true>(a, split_k, s);
        return wide_n ? launch_wgrad<1, 2, true>(a, split_k, s) : launch_wgrad<1, 1, true>(a, split_k, s);
    }
    const bool wide_n = a.Kdim > 64;
    if (wide_m) return wide_n ? launch_wgrad<2, 2, false>(a, split_k, s) : launch_wgrad<2, 1, false>(a, split_k, s);
    return wide_n ? launch_wgrad<1, 2, false>(a, split_k, s) : launch_wgrad<1, 1, false>(a, split_k, s);
}
