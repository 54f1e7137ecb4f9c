// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip: fp32-MFMA and the first bf16x3 schedule;
// conv_igemm2.hip: the one-wave-per-SIMD software-pipelined schedule for bf16x3 and plain bf16).
#pragma once
#include <type_traits>

#include "hn_common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace hnconv {

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    double* stats;
    const float* addend;     // optional tensor added to the output (gradient of a branch that by-passed this conv)
    int ldadd;
    int H, W, C, ldx;        // input tensor
    int Ho, Wo, K, ldy;      // output tensor (K = output channels)
    int KH, KW, stride, pad, pad_w, dil, in_dil;  // pad = rows, pad_w = columns
    // output mapping (parity classes of a stride-2 data gradient): output pixel (n, i, j) of this launch is pixel
    // (n, ostep*i + oph, ostep*j + opw) of a tensor with oH x oW pixels per image; omap = 0 -> dense (identity)
    int omap, oH, oW, ostep, oph, opw;
    // split-K (launches with few output tiles but a long reduction): K steps per split, 0 = no split.  Split launches
    // add into a pre-initialised output with global_atomic_add_f32 and leave bias / activation / statistics to a follow-up pass.
    int cps;
    int M, Kdim, ntm, ntn;
    int gm;  // row tiles per group of the tile order (see the kernel)
    int tap_inner = 0;  // conv_igemm4: the K loop walks the filter taps INSIDE a 32-channel chunk (see the kernel)
    int act;
    float slope;
    long long x_bytes;  // extent of the gathered tensor from p.x, for the buffer descriptors
    // bf16x3 / bf16: the weight matrix pre-split into three bf16 planes (hn_split_bf16x3 / hn_weight_flip_transpose),
    // plane p = wx3 + p * wx3_plane elements, each laid out exactly like the fp32 matrix `w`; plane 0 alone is the
    // round-to-nearest bf16 weight of the plain bf16 arithmetic
    const unsigned short* wx3;
    long long wx3_plane;
    // A-operand transform (forward only): the gathered tensor is the PRE-normalisation output of the producing
    // convolution and the loader applies relu(x * in_scale[c] + in_shift[c]) on the fly (BatchNorm + ReLU of a layer
    // whose only consumer is this convolution); taps that fall into the zero padding stay zero.  NULL = plain gather.
    const float* in_scale;
    const float* in_shift;
    // f16x2: the gathered tensor's scale slot (bit pattern of its max |x|, hn_absmax_scale) and the weight planes' {s, 1/s}
    // (stored behind the two f16 planes, hn_split_f16x2); NULL for the other arithmetics
    const float* a_scale;
    const float* w_scale;
    // f16x2, fourth schedule (conv_igemm4.hip): the gathered tensor as two fp16 planes (hi, lo) of xh_plane elements each,
    // pixel stride ldx fp16 elements, made with the scale of a_scale by the pass that produced the tensor; x_bytes is then
    // the extent of ONE plane.  NULL: the fp32 tensor `x` is split inside the kernel.
    const unsigned short* xh = nullptr;
    long long xh_plane = 0;
    // optional [HN_STAT_REPLICAS][2][K] words next to `stats`: order-preserving keys of the per-channel maximum of y and of
    // -y over the launch (zero = nothing seen), from which hn_bn_finalize bounds the normalised tensor's magnitude
    unsigned* minmax = nullptr;
    // one-plane bf16 arithmetic, forward with BatchNorm statistics: y holds bf16 elements (ldy counts elements) -- the
    // statistics still come from the fp32 accumulators.  Honoured by the conv_igemm4 one-plane kernels only.
    int y_bf16 = 0;
    // data gradient of the convolution BEHIND a BatchNorm + ReLU (conv_igemm4 kernels, dense stride-1 launches): this launch's
    // output IS that layer's dz, so its epilogue also takes the first pass of the BatchNorm backward over the tile it holds
    // -- du = dz * (u > 0), sums[0:K] += sum du, sums[K:2K] += sum du * xhat, du_max = max |du| per channel -- reading the
    // layer's pre-normalisation tensor y instead of a stand-alone pass reading dz and y (hn_bn_act_bwd_reduce).  NULL: off.
    const void* br_y = nullptr;          // [pixels][br_ldy] fp32, or bf16 elements when br_ybf
    int br_ldy = 0, br_ybf = 0, br_relu = 0;
    const float* br_scale = nullptr;     // u = y * scale + shift (ReLU mask when there are no byte masks)
    const float* br_shift = nullptr;
    const float* br_mean = nullptr;      // xhat = (y - mean) * invstd
    const float* br_invstd = nullptr;
    const unsigned char* br_bits = nullptr;  // the forward's byte ReLU masks (a residual was added in front of the ReLU)
    double* br_sums = nullptr;           // [HN_STAT_REPLICAS][2K + 1]
    unsigned* br_gmax = nullptr;         // [HN_STAT_REPLICAS][K] or NULL
#ifndef HN_STAMP_TID
#define HN_STAMP_TID 0  // the stamping thread (make STAMPS=1 STAMP_TID=448: the last wave of an eight-wave workgroup)
#endif
#ifdef HN_STAMPS
    long long* stamps = nullptr;         // diagnostic build (make STAMPS=1): 16 words per workgroup, tools/stamp_conv.py
#endif
};
#ifdef HN_STAMPS
extern long long* g_hn_stamps;  // conv_igemm.hip, set by hn_debug_set_stamps
#define HN_STAMP(p, i)                                                                                   \
    do {                                                                                                 \
        if ((p).stamps != nullptr && threadIdx.x == HN_STAMP_TID) (p).stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HN_STAMP(p, i) do { } while (0)
#endif

// bf16x3 LDS image: [split][row][32 k] bf16 with 80-byte rows (64 B of data + 16 B pad: 16 consecutive rows start in
// 16 distinct 16-B slots of the 256-B bank row -> conflict-free ds_read_b128 of a lane's 8 consecutive k)
constexpr int X3_ROW = 80;
// ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59},
// {36-43,48-51,60-63} (MI355X_MICROARCH.md, LDS): with the 16x16x32 operand map (lane -> row lane & 15, k group
// lane >> 4) each group holds every row once, rows {0-3,12-15} at k group g and rows {4-11} at g ^ 1.  Storing the
// 16-B k groups of rows 4-11 with their low index bit flipped makes all 16 lanes of a group address the same physical
// position of 16 different rows: conflict-free with the 80-B row stride (measured before: SQ_LDS_BANK_CONFLICT = 50 %
// of SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_conv_igemm_bf16x3.txt).
__device__ __forceinline__ int x3_kflip(int row) { return ((row >> 2) ^ (row >> 3)) & 1; }

// exact three-way split of four consecutive-k fp32 values into packed bf16 (hi, mid, lo)
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r = v[e] - (float)hh;  // exact
        const __bf16 mm = (__bf16)r;
        const float r2 = r - (float)mm;    // exact
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)r2;
    }
}

// two-way split of four consecutive-k fp32 values, pre-scaled by the power of two `s`, into packed fp16 (hi, lo):
// x * s = hi + lo + e with |e| <= 2^-23 |x * s| while lo stays a normal fp16 number (|x * s| >= 2^-3)
__device__ __forceinline__ void split2h(const f32x4 v, float s, f16x4& h, f16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * s;  // exact
        const _Float16 hh = (_Float16)xs;
        h[e] = hh;
        l[e] = (_Float16)(xs - (float)hh);  // the difference is exact in fp32
    }
}

__device__ __forceinline__ bf16x4 round_bf16(const f32x4 v) {
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return h;
}

__device__ __forceinline__ f32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}


// ---- accumulator <-> tile coordinates ----------------------------------------------------------------------------
// A wave owns TM x TN blocks of 32x32 outputs; element r (0..15) of acc[i][j]:
//   32x32 MFMA (S16 = false):  column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
//   16x16 MFMA (S16 = true):   registers 4s..4s+3 hold sub-block s = 2 * (row half) + (column half):
//                              column = 16 * ((r >> 2) & 1) + (lane & 15), row = 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3)
template <int TM, int TN, bool S16>
struct AccMap {
    static constexpr int NC = S16 ? 2 : 1;  // distinct columns a lane holds per 32-column block
    int wm, wn, lane, n0;
    __device__ __forceinline__ static int c2_of(int r) { return S16 ? (r >> 2) & 1 : 0; }
    __device__ __forceinline__ static constexpr bool dead(int, int) { return false; }  // (a register that holds no output: never here)
    __device__ __forceinline__ int row(int i, int r) const {  // row inside the block tile
        return (wm * TM + i) * 32 + (S16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
    }
    // S16: row = row_base (the same for the whole wave) + row_lane (the same for all of a lane's rows)
    __device__ __forceinline__ int row_base(int i, int r) const { return (wm * TM + i) * 32 + 16 * (r >> 3) + (r & 3); }
    __device__ __forceinline__ int row_lane() const { return 4 * (lane >> 4); }
    __device__ __forceinline__ int col(int j, int c2) const {  // global output channel
        return n0 + (wn * TN + j) * 32 + (S16 ? 16 * c2 + (lane & 15) : (lane & 31));
    }
};

// Accumulators start at zero -- or at the addend (a gradient that reached this tensor through another branch): its
// loads are issued ahead of the operand prologue and their latency disappears behind the first K steps, so the
// epilogue stays a pure store loop.  The addend never overlaps the output (header contract).
// `iscale`, `iscale2` (f16x2: the two operand scales, powers of two): the accumulators live in the scaled domain.  The two
// factors are applied one after the other -- their product can leave the fp32 range (a gradient of ~1e-30 against weights of
// ~0.1: 2^114 * 2^17) and 0 * inf would turn a zero of the addend into NaN.
template <int TM, int TN, bool S16, class AM = AccMap<TM, TN, S16>>
__device__ __forceinline__ void conv_init_acc(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AM& am,
                                              const int* rowinfo, int m0, float iscale = 1.f, float iscale2 = 1.f) {
    if (p.addend != nullptr && p.cps == 0) {
        const float* __restrict__ add = p.addend;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = am.row(i, r), col = am.col(j, am.c2_of(r));
                    float v = 0.f;
                    if (!AM::dead(i, r) && m0 + row < p.M && col < p.K) v = add[(size_t)rowinfo[row * 4 + 3] * p.ldadd + col];
                    acc[i][j][r] = v * iscale * iscale2;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
}

// ---- epilogue: bias, batch-statistics partials, activation, store -------------------------------------------------
// The bias is loaded unconditionally (clamped index) and folded in BEFORE any store is issued, and full tiles take a
// branch-free path: a conditional load inside the store loop makes hipcc wait vmcnt(0) -- i.e. for the previous store --
// in front of every store.
// `oscale`, `oscale2` (f16x2: the inverse operand scales, powers of two, applied one after the other -- see conv_init_acc)
// take the accumulators back to the tensor's own scale first.
template <int TM, int TN, bool S16, int BM = 64 * TM, int BN = 64 * TN, bool SCALED = false, bool YB = false, bool BR = false,
          class AM = AccMap<TM, TN, S16>>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AM& am,
                                              const int* rowinfo, int m0, int tile_m, int split, float oscale = 1.f,
                                              float oscale2 = 1.f) {
    constexpr int NC = AM::NC;
    const int lane = am.lane, n0 = am.n0;
    if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * oscale * oscale2;
    }
    if (p.cps > 0) {  // split-K partial
#pragma unroll
        for (int j = 0; j < TN; ++j) {  // accumulate into the pre-initialised output
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + am.row(i, r), col = am.col(j, am.c2_of(r));
                    if (!AM::dead(i, r) && row < p.M && col < p.K) {
                        const size_t opix = p.omap ? (size_t)rowinfo[(row - m0) * 4 + 3] : (size_t)row;
                        atomicAdd(p.y + opix * p.ldy + col, acc[i][j][r]);
                    }
                }
            }
        }
        return;
    }
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.K);
    float bias_v[TN][NC];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int c2 = 0; c2 < NC; ++c2) bias_v[j][c2] = p.bias != nullptr ? p.bias[min(am.col(j, c2), p.K - 1)] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bias_v[j][am.c2_of(r)];

    if constexpr (BR) {
        if (p.br_y != nullptr) {  // (ConvArgs::br_y) before the stores: its loads would otherwise queue behind them
            // Loads are UNCONDITIONAL (clamped indices) and issued as a batch per column, the arithmetic follows: a load under a
            // condition makes hipcc drain the memory queue in front of every one of them (see the store loop below).
            auto reduce = [&](auto ybf_tag, auto bits_tag) {
                constexpr bool YBF = decltype(ybf_tag)::value, BITS = decltype(bits_tag)::value;
                constexpr int NR = 16 / NC;  // rows of one column a lane holds per 32 x 32 block
                const int kq = p.K >> 2;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) {
                        const int col = am.col(j, c2);
                        const bool cok = col < p.K;
                        const int cc = cok ? col : p.K - 1;
                        const float sc = p.br_scale[cc], sh = p.br_shift[cc], mu = p.br_mean[cc], is = p.br_invstd[cc];
                        float yv[TM][NR];
                        unsigned bb[TM][NR];
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            int q = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (am.c2_of(r) != c2 || AM::dead(i, r)) continue;
                                const int row = min(m0 + am.row(i, r), p.M - 1);
                                const size_t at = (size_t)row * p.br_ldy + cc;
                                if constexpr (YBF) yv[i][q] = (float)reinterpret_cast<const __bf16*>(p.br_y)[at];
                                else yv[i][q] = reinterpret_cast<const float*>(p.br_y)[at];
                                if constexpr (BITS) bb[i][q] = p.br_bits[(size_t)row * kq + (cc >> 2)];
                                else bb[i][q] = 0u;
                                ++q;
                            }
                        }
                        float b1 = 0.f, b2 = 0.f;
                        unsigned bm = 0u;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            int q = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (am.c2_of(r) != c2 || AM::dead(i, r)) continue;
                                const bool in = (m0 + am.row(i, r) < p.M) && cok;
                                bool on;
                                if constexpr (BITS) on = (bb[i][q] >> (cc & 3)) & 1u;
                                else on = !p.br_relu || yv[i][q] * sc + sh > 0.f;
                                const float du = (in && on) ? acc[i][j][r] : 0.f;
                                b1 += du;
                                b2 += du * ((yv[i][q] - mu) * is);
                                bm = max(bm, __float_as_uint(du) & 0x7fffffffu);
                                ++q;
                            }
                        }
                        double d1 = (double)b1, d2 = (double)b2;
                        d1 += __shfl_xor(d1, 32);
                        d2 += __shfl_xor(d2, 32);
                        bm = max(bm, (unsigned)__shfl_xor((int)bm, 32));
                        if constexpr (S16) {
                            d1 += __shfl_xor(d1, 16);
                            d2 += __shfl_xor(d2, 16);
                            bm = max(bm, (unsigned)__shfl_xor((int)bm, 16));
                        }
                        if ((S16 ? (lane >> 4) : (lane >> 5)) == 0 && cok) {
                            const int rep_i = tile_m % HN_STAT_REPLICAS;
                            double* rep = p.br_sums + (size_t)rep_i * (2 * p.K + 1);
                            unsafeAtomicAdd(rep + col, d1);
                            unsafeAtomicAdd(rep + p.K + col, d2);
                            if (p.br_gmax != nullptr && bm != 0u) atomicMax(p.br_gmax + (size_t)rep_i * p.K + col, bm);
                        }
                    }
                }
            };
            if (p.br_ybf) {
                if (p.br_bits != nullptr) reduce(std::true_type{}, std::true_type{});
                else reduce(std::true_type{}, std::false_type{});
            } else {
                if (p.br_bits != nullptr) reduce(std::false_type{}, std::true_type{});
                else reduce(std::false_type{}, std::false_type{});
            }
        }
    }
    // branch-free activation: out = v > 0 ? v : v * neg_slope  (none: 1, relu: 0, leaky/prelu: slope)
    const float neg_slope = p.act == HN_ACT_NONE ? 1.f : (p.act == HN_ACT_RELU ? 0.f : p.slope);
    auto finish = [&](auto full_tag, auto map_tag, auto act_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool OMAP = decltype(map_tag)::value;  // rows land on interleaved pixels (rowinfo[.][3])
        constexpr bool ACT = decltype(act_tag)::value;   // false: HN_ACT_NONE -- the stored value IS the accumulator (the fused conv ->
                                                         // BatchNorm launches: three vector instructions per element less)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2) {
                const int col = am.col(j, c2);
                const bool cok = FULL || col < p.K;
                float s1 = 0.f, s2 = 0.f;
                float vmax = -__builtin_inff(), vmin = __builtin_inff();
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (am.c2_of(r) != c2 || AM::dead(i, r)) continue;
                        const int row = m0 + am.row(i, r);
                        const float v = acc[i][j][r];
                        if (FULL || (row < p.M && cok)) {
                            if constexpr (OMAP) {
                                const size_t opix = (size_t)rowinfo[(row - m0) * 4 + 3];
                                p.y[opix * p.ldy + col] = v;
                            } else {
                                s1 += v;
                                s2 += v * v;
                                vmax = fmaxf(vmax, v);
                                vmin = fminf(vmin, v);
                                const float o = ACT ? (v > 0.f ? v : v * neg_slope) : v;
                                if constexpr (YB) {
                                    if (p.y_bf16) reinterpret_cast<__bf16*>(p.y)[(size_t)row * p.ldy + col] = (__bf16)o;
                                    else p.y[(size_t)row * p.ldy + col] = o;
                                } else {
                                    p.y[(size_t)row * p.ldy + col] = o;
                                }
                            }
                        }
                    }
                }
                if (p.stats != nullptr) {
                    // <= 64 rows per lane were summed in fp32; everything across tiles is accumulated in double
                    // (global_atomic_add_f64), like the double accumulators of the CPU reference's BatchNorm.
                    double d1 = (double)s1, d2 = (double)s2;
                    d1 += __shfl_xor(d1, 32);
                    d2 += __shfl_xor(d2, 32);
                    if constexpr (S16) {
                        d1 += __shfl_xor(d1, 16);
                        d2 += __shfl_xor(d2, 16);
                    }
                    if ((S16 ? (lane >> 4) : (lane >> 5)) == 0 && cok) {
                        double* rep = p.stats + (size_t)(tile_m % HN_STAT_REPLICAS) * 2 * p.K;  // spread same-address atomics
                        unsafeAtomicAdd(rep + col, d1);
                        unsafeAtomicAdd(rep + p.K + col, d2);
                    }
                    if (p.minmax != nullptr) {  // range of the channel: keys of max(y) and max(-y), see hn_range_key
                        vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
                        vmin = fminf(vmin, __shfl_xor(vmin, 32));
                        if constexpr (S16) {
                            vmax = fmaxf(vmax, __shfl_xor(vmax, 16));
                            vmin = fminf(vmin, __shfl_xor(vmin, 16));
                        }
                        if ((S16 ? (lane >> 4) : (lane >> 5)) == 0 && cok) {
                            unsigned* rk = p.minmax + (size_t)(tile_m % HN_STAT_REPLICAS) * 2 * p.K;
                            atomicMax(rk + col, hn_range_key(vmax));
                            atomicMax(rk + p.K + col, hn_range_key(-vmin));
                        }
                    }
                }
            }
        }
    };
    if (p.omap) {  // parity class of a strided data gradient (no activation / statistics on this path)
        finish(std::false_type{}, std::true_type{}, std::false_type{});
    } else if (p.act == HN_ACT_NONE) {
        if (full) finish(std::true_type{}, std::false_type{}, std::false_type{});
        else finish(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        if (full) finish(std::true_type{}, std::false_type{}, std::true_type{});
        else finish(std::false_type{}, std::false_type{}, std::true_type{});
    }
}

// Tile order inside an XCD's contiguous range: groups of `gm` row tiles are walked column by column, so the ~64
// workgroups an XCD runs at a time cover a gm x (64/gm) patch -- each A panel and each B panel in its L2 is shared
// by several of them -- instead of a few rows across every column.  gm = 1: row-major.
__device__ __forceinline__ void conv_tile_of(const ConvArgs& p, int tile, int& tile_m, int& tile_n) {
    const int gm = p.gm, per_group = gm * p.ntn;
    const int g = tile / per_group, in_g = tile - g * per_group;
    const int first = g * gm, rows = min(gm, p.ntm - first);
    tile_n = in_g / rows;
    tile_m = first + (in_g - tile_n * rows);
}

// rowinfo[r] = {h base, w base, image pixel base, output pixel} of block-tile row r
template <int BM>
__device__ __forceinline__ void conv_fill_rowinfo(const ConvArgs& p, int* rowinfo, int m0, int tid) {
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int hb = -0x40000000, wb = 0, nb = 0, opix = 0;  // rows past M: every tap lands out of range
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hb = ho * p.stride - p.pad;
            wb = wo * p.stride - p.pad_w;
            nb = n * p.H * p.W;
            opix = p.omap ? (n * p.oH + p.ostep * ho + p.oph) * p.oW + p.ostep * wo + p.opw : m;
        }
        rowinfo[r * 4 + 0] = hb;
        rowinfo[r * 4 + 1] = wb;
        rowinfo[r * 4 + 2] = nb;
        rowinfo[r * 4 + 3] = opix;
    }
}

// conv_igemm2.hip
enum : int { MATH2_X3 = 3, MATH2_BF16 = 1, MATH2_H2 = 2 };  // = number of 16-bit planes per operand (2: fp16, f16x2)
// runs the v2 schedule if it covers (tall, wide, ns); returns HN_OK / error, or 1 when it does not cover the request
int launch_conv2(const ConvArgs& a, bool tall, bool wide, int ns, hipStream_t stream);
// conv_igemm4.hip: f16x2 with both operands as fp16 planes, everything by LDS-DMA
bool conv4_covers(const ConvArgs& a);
int launch_conv4(const ConvArgs& a, int tile, hipStream_t stream);
const char* conv4_kernel_name(int tile, bool bf16);
// conv_igemm3.hip: bf16x3, 128 x 256 tile, weight operand by LDS-DMA
bool conv3_covers(const ConvArgs& a, int planes, bool wide);
int launch_conv3(const ConvArgs& a, int planes, bool wide, hipStream_t stream);
const char* conv2_kernel_name(bool tall, bool wide, int ns);

}  // namespace hnconv
