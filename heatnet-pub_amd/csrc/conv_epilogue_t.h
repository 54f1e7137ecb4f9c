// Epilogue of the all-LDS-DMA convolution kernels (conv_igemm4.hip) on TRANSPOSED accumulators.
//
// v_mfma_f32_16x16x32 leaves D[4 * (lane >> 4) + r][lane & 15] in register r of a lane.  conv_igemm4 feeds the WEIGHT fragment
// as the instruction's first operand and the pixel fragment as its second, so "D" is the 16 x 16 sub-block transposed: a lane
// holds ONE output pixel (lane & 15) and FOUR CONSECUTIVE output channels (4 * (lane >> 4) + r) -- 16 contiguous bytes of the
// NHWC output.  Everything the epilogue moves is then a 16-byte access per lane and sub-block: the output store, the
// skip-gradient addend, the pre-normalisation tensor of the fused BatchNorm-backward reduction -- a quarter of the vector-memory
// instructions of the one-dword-per-row form (the store tail of a tile is bound by instruction issue, not by bytes:
// MI355X_MICROARCH.md, "attention epilogue store tail").  The products and their accumulation order per output element are
// unchanged, so the results are bitwise those of the untransposed kernels.
//
// The per-channel reductions (batch statistics, range keys, the BatchNorm-backward sums) now run ACROSS the 16 lanes of a DPP
// row instead of down a lane's registers: a reduce-scatter over row_mirror / row_half_mirror / quad_perm steps -- lane k of a
// row ends up with the total of value k of the 16 (8) values every lane carries, 45 (31) vector instructions per quantity --
// and one atomic instruction per quantity then covers 64 different channels.
#pragma once
#include "conv_common.h"

namespace hnconv {

// registers 4 * (2 * rh + ch) .. + 3 of acc[i][j]: pixel row(i, rh), channels col0(j, ch) .. + 3
template <int TM, int TN>
struct AccMapT {
    int wm, wn, lane, n0;
    __device__ __forceinline__ int row(int i, int rh) const { return (wm * TM + i) * 32 + 16 * rh + (lane & 15); }  // inside the tile
    __device__ __forceinline__ int col0(int j, int ch) const { return n0 + (wn * TN + j) * 32 + 16 * ch + 4 * (lane >> 4); }
};

__device__ __forceinline__ bool dev_aligned16(const void* q) { return (reinterpret_cast<size_t>(q) & 15u) == 0; }

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
struct RsAdd {
    __device__ __forceinline__ static float f(float a, int b) { return a + __builtin_bit_cast(float, b); }
};
struct RsMaxF {
    __device__ __forceinline__ static float f(float a, int b) { return fmaxf(a, __builtin_bit_cast(float, b)); }
};
struct RsMinF {
    __device__ __forceinline__ static float f(float a, int b) { return fminf(a, __builtin_bit_cast(float, b)); }
};
struct RsMaxU {
    __device__ __forceinline__ static unsigned f(unsigned a, int b) { return max(a, (unsigned)b); }
};

// Reduce-scatter over the 16 lanes of a DPP row: on return v[0] of lane k (k = lane & 15 for N = 16; lane & 7 for N = 8, lanes
// with bit 3 clear) is op over the row's 16 lanes of their v[k].  Partners: 15 - k (row_mirror), 7 - k within each half
// (row_half_mirror), k ^ 2, k ^ 1 (quad_perm) -- each differs from the lane in exactly the bit that selects which half of the
// remaining values a lane keeps, and the four steps together reach all 16 lanes (the partner offsets {15, 7, 2, 1} generate
// the whole group under xor).
template <int N, class Op, class T>
__device__ __forceinline__ void reduce_scatter16(T (&v)[N], int lane) {
    static_assert(N == 16 || N == 8, "values per lane");
    auto step = [&](auto ctrl_tag, auto half_tag, int bit) {
        constexpr int CTRL = decltype(ctrl_tag)::value, HALF = decltype(half_tag)::value;
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int t = 0; t < HALF; ++t) {
            const T send = up ? v[t] : v[t + HALF];
            const T keep = up ? v[t + HALF] : v[t];
            v[t] = Op::f(keep, dpp_i<CTRL>(__builtin_bit_cast(int, send)));
        }
    };
    if constexpr (N == 16) {
        step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 8>{}, 8);
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = Op::f(v[t], dpp_i<0x140>(__builtin_bit_cast(int, v[t])));
    }
    step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 4>{}, 4);
    step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 2>{}, 2);
    step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 1>{}, 1);
}

// Accumulators start at zero, or at the addend (conv_init_acc in conv_common.h: same contract, same arithmetic).
template <int TM, int TN>
__device__ __forceinline__ void conv_init_acc_t(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AccMapT<TM, TN>& am,
                                                const int* rowinfo, int m0, float iscale, float iscale2) {
    if (p.addend != nullptr) {
        const float* __restrict__ add = p.addend;
        const bool vec = (p.K & 3) == 0 && (p.ldadd & 3) == 0 && dev_aligned16(add);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int col0 = am.col0(j, ch);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        const int row = am.row(i, rh);
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (m0 + row < p.M) {
                            const float* src = add + (size_t)rowinfo[row * 4 + 3] * p.ldadd + col0;
                            if (vec) {
                                if (col0 < p.K) v = *reinterpret_cast<const f32x4*>(src);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (col0 + e < p.K) v[e] = src[e];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * (2 * rh + ch) + e] = v[e] * iscale * iscale2;
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

// bias, the fused BatchNorm-backward reduction (ConvArgs::br_*), activation, store, batch-statistics partials and range keys.
// SCALED: f16x2 (oscale, oscale2 take the accumulators back to the tensor's own scale, one after the other); YB: the output may
// be bf16 (ConvArgs::y_bf16).  No split-K on this path.
template <int TM, int TN, int BM, int BN, bool SCALED, bool YB>
__device__ __forceinline__ void conv_epilogue_t(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AccMapT<TM, TN>& am,
                                                const int* rowinfo, int m0, int tile_m, float oscale, float oscale2) {
    constexpr int NV = 8 * TN;  // channels a lane holds: (j, ch, e) -> value index (2 * j + ch) * 4 + e
    const int lane = am.lane, n0 = am.n0;
    if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * oscale * oscale2;
    }
    // the lane that ends up with value k of a reduce-scatter, and the channel that value belongs to
    const int rs_k = lane & (NV - 1);
    const bool rs_on = NV == 16 || (lane & 8) == 0;
    const int rs_col = n0 + (am.wn * TN + (rs_k >> 3)) * 32 + 16 * ((rs_k >> 2) & 1) + 4 * (lane >> 4) + (rs_k & 3);
    const int rep_i = tile_m % HN_STAT_REPLICAS;  // spread same-address atomics
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.K);
    const float neg_slope = p.act == HN_ACT_NONE ? 1.f : (p.act == HN_ACT_RELU ? 0.f : p.slope);

    auto body = [&](auto vec_tag, auto full_tag, auto map_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;    // 16-byte accesses: K % 4 == 0, ldy % 4 == 0, aligned base
        constexpr bool FULL = decltype(full_tag)::value;  // no row / column of the tile lies outside the output
        constexpr bool OMAP = decltype(map_tag)::value;   // rows land on interleaved pixels (rowinfo[.][3]); no activation / statistics
        // ---- bias (loaded unconditionally, clamped; folded in before anything is stored) ----
        if (p.bias != nullptr) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int col0 = am.col0(j, ch);
                    f32x4 b;
                    if constexpr (VEC) {
                        b = *reinterpret_cast<const f32x4*>(p.bias + min(col0, p.K - 4));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[e] = p.bias[min(col0 + e, p.K - 1)];
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * (2 * rh + ch) + e] += b[e];
                }
        }
        // ---- first pass of the BatchNorm backward of the layer whose dz this tile is (host: only with VEC geometry) ----
        if constexpr (VEC && !OMAP) {
            if (p.br_y != nullptr) {
                auto reduce = [&](auto ybf_tag, auto bits_tag) {
                    constexpr bool YBF = decltype(ybf_tag)::value, BITS = decltype(bits_tag)::value;
                    const int kq = p.K >> 2;
                    float b1[NV], b2[NV];
                    unsigned bm[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        b1[k] = 0.f;
                        b2[k] = 0.f;
                        bm[k] = 0u;
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) {
                            const int col0 = am.col0(j, ch);
                            const bool cok = FULL || col0 < p.K;
                            const int cc = cok ? col0 : p.K - 4;
                            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.br_scale + cc);
                            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.br_shift + cc);
                            const f32x4 mu = *reinterpret_cast<const f32x4*>(p.br_mean + cc);
                            const f32x4 is = *reinterpret_cast<const f32x4*>(p.br_invstd + cc);
                            f32x4 yv[TM][2];
                            unsigned bb[TM][2];
                            // unconditional (clamped) loads, issued as a batch: a load under a condition makes hipcc drain the
                            // memory queue in front of it
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int rh = 0; rh < 2; ++rh) {
                                    const int row = FULL ? m0 + am.row(i, rh) : min(m0 + am.row(i, rh), p.M - 1);
                                    const size_t at = (size_t)row * p.br_ldy + cc;
                                    if constexpr (YBF) {
                                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.br_y) + at);
#pragma unroll
                                        for (int e = 0; e < 4; ++e) yv[i][rh][e] = (float)t[e];
                                    } else {
                                        yv[i][rh] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.br_y) + at);
                                    }
                                    if constexpr (BITS) bb[i][rh] = p.br_bits[(size_t)row * kq + (cc >> 2)];
                                    else bb[i][rh] = 0u;
                                }
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int rh = 0; rh < 2; ++rh) {
                                    const bool in = FULL || ((m0 + am.row(i, rh) < p.M) && cok);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        bool on;
                                        if constexpr (BITS) on = (bb[i][rh] >> e) & 1u;
                                        else on = !p.br_relu || yv[i][rh][e] * sc[e] + sh[e] > 0.f;
                                        const float du = (in && on) ? acc[i][j][4 * (2 * rh + ch) + e] : 0.f;
                                        const int k = (2 * j + ch) * 4 + e;
                                        b1[k] += du;
                                        b2[k] += du * ((yv[i][rh][e] - mu[e]) * is[e]);
                                        bm[k] = max(bm[k], __float_as_uint(du) & 0x7fffffffu);
                                    }
                                }
                        }
                    reduce_scatter16<NV, RsAdd>(b1, lane);
                    reduce_scatter16<NV, RsAdd>(b2, lane);
                    if (p.br_gmax != nullptr) reduce_scatter16<NV, RsMaxU>(bm, lane);
                    if (rs_on && rs_col < p.K) {
                        double* rep = p.br_sums + (size_t)rep_i * (2 * p.K + 1);
                        unsafeAtomicAdd(rep + rs_col, (double)b1[0]);
                        unsafeAtomicAdd(rep + p.K + rs_col, (double)b2[0]);
                        if (p.br_gmax != nullptr && bm[0] != 0u) atomicMax(p.br_gmax + (size_t)rep_i * p.K + rs_col, bm[0]);
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
        // ---- activation + store ----
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int col0 = am.col0(j, ch);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        const int row = am.row(i, rh);
                        if (!FULL && m0 + row >= p.M) continue;
                        const size_t opix = OMAP ? (size_t)rowinfo[row * 4 + 3] : (size_t)(m0 + row);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][j][4 * (2 * rh + ch) + e];
                            o[e] = OMAP ? v : (v > 0.f ? v : v * neg_slope);
                        }
                        if constexpr (VEC) {
                            if (FULL || col0 < p.K) {
                                if (YB && p.y_bf16) {
                                    bf16x4 ob;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) ob[e] = (__bf16)o[e];
                                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.y) + opix * p.ldy + col0) = ob;
                                } else {
                                    *reinterpret_cast<f32x4*>(p.y + opix * p.ldy + col0) = o;
                                }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col0 + e < p.K) {
                                    if (YB && p.y_bf16) reinterpret_cast<__bf16*>(p.y)[opix * p.ldy + col0 + e] = (__bf16)o[e];
                                    else p.y[opix * p.ldy + col0 + e] = o[e];
                                }
                        }
                    }
            }
        // ---- batch statistics and range keys of the PRE-activation values (after the stores are on their way) ----
        if constexpr (!OMAP) {
            if (p.stats != nullptr) {
                float s1[NV], s2[NV], vmx[NV], vmn[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    s1[k] = 0.f;
                    s2[k] = 0.f;
                    vmx[k] = -__builtin_inff();
                    vmn[k] = __builtin_inff();
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int col0 = am.col0(j, ch);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int rh = 0; rh < 2; ++rh) {
                                const bool rok = FULL || m0 + am.row(i, rh) < p.M;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (!FULL && !(rok && col0 + e < p.K)) continue;
                                    const float v = acc[i][j][4 * (2 * rh + ch) + e];
                                    const int k = (2 * j + ch) * 4 + e;
                                    s1[k] += v;
                                    s2[k] += v * v;
                                    vmx[k] = fmaxf(vmx[k], v);
                                    vmn[k] = fminf(vmn[k], v);
                                }
                            }
                    }
                // <= 16 * 2 * TM rows are summed in fp32 (down the registers, then across the row's 16 lanes); everything across
                // tiles is accumulated in double (global_atomic_add_f64), like the double accumulators of the CPU reference
                reduce_scatter16<NV, RsAdd>(s1, lane);
                reduce_scatter16<NV, RsAdd>(s2, lane);
                const bool mine = rs_on && rs_col < p.K;
                if (mine) {
                    double* rep = p.stats + (size_t)rep_i * 2 * p.K;
                    unsafeAtomicAdd(rep + rs_col, (double)s1[0]);
                    unsafeAtomicAdd(rep + p.K + rs_col, (double)s2[0]);
                }
                if (p.minmax != nullptr) {  // keys of max(y) and max(-y), see hn_range_key
                    reduce_scatter16<NV, RsMaxF>(vmx, lane);
                    reduce_scatter16<NV, RsMinF>(vmn, lane);
                    if (mine) {
                        unsigned* rk = p.minmax + (size_t)rep_i * 2 * p.K;
                        atomicMax(rk + rs_col, hn_range_key(vmx[0]));
                        atomicMax(rk + p.K + rs_col, hn_range_key(-vmn[0]));
                    }
                }
            }
        }
    };
    const size_t ybytes = (YB && p.y_bf16) ? 8 : 16;
    const bool vec = (p.K & 3) == 0 && (p.ldy & 3) == 0 && (reinterpret_cast<size_t>(p.y) & (ybytes - 1)) == 0 &&
                     (p.bias == nullptr || dev_aligned16(p.bias));
    if (p.omap) {  // parity class of a strided data gradient
        if (vec) body(std::true_type{}, std::false_type{}, std::true_type{});
        else body(std::false_type{}, std::false_type{}, std::true_type{});
    } else if (vec) {
        if (full) body(std::true_type{}, std::true_type{}, std::false_type{});
        else body(std::true_type{}, std::false_type{}, std::false_type{});
    } else {
        body(std::false_type{}, std::false_type{}, std::false_type{});
    }
}

}  // namespace hnconv
