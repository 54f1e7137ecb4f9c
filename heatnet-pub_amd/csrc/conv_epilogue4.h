// Epilogue of conv_igemm4_kernel (conv_igemm4.hip), ROW by row.
//
// conv_common.h's conv_epilogue walks a lane's outputs column by column and lets the compiler keep one 64-bit address per row
// alive across the columns: 64 rows per lane with TM = 4 (the 224 x 256 tile), next to 128 accumulator registers, in the 256
// registers an eight-wave workgroup has -- the accumulators spilled to scratch and the tile's epilogue took 30-38 us where the
// 128 x 256 tile's takes 2 (tools/stamp_conv.py, profiles/r04_stamps_tile_lifetime.txt).  Here a row's address is
//     scalar base (wave-uniform: tile row + AM::row_base(i, r))  +  ONE 32-bit lane offset for all of a lane's rows
//     (row_lane * ld + the lane's first column)  +  an immediate for the column (j * 32 + c2 * 16)
// so a whole tile needs no per-row address registers and no 64-bit vector arithmetic.  Tiles that hang over M or K clamp the
// lane part per row (three vector instructions).  Per column the rows are visited in the same order as in conv_epilogue: the
// statistics, range keys and BatchNorm-backward partial sums are the same fp32 sums, bit for bit.
// No split-K on this path (conv4_covers).
#pragma once
#include "conv_common.h"

namespace hnconv {

// Accumulators start at zero -- or at the addend (conv_common.h: conv_init_acc), read row by row like everything below: a scalar
// row base, one lane offset, the column as an immediate.  Rows that land on interleaved pixels (ConvArgs::omap: rowinfo[.][3]) keep
// a 64-bit address per row.
template <int TM, int TN, int BM, int BN, class AM>
__device__ __forceinline__ void conv_init_acc4(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AM& am, const int* rowinfo, int m0,
                                               float iscale, float iscale2) {
    constexpr int NC = 2;
    if (p.addend == nullptr) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        return;
    }
    const int col0 = am.col(0, 0), rl = am.row_lane();
    const bool full = (m0 + BM <= p.M) && (am.n0 + BN <= p.K) && !p.omap;
    if (full) {
        const unsigned lo = (unsigned)(rl * p.ldadd + col0) * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row_u = __builtin_amdgcn_readfirstlane(m0 + am.row_base(i, 8 * rh + r4));
                    const char* rp = reinterpret_cast<const char*>(p.addend + (size_t)row_u * p.ldadd);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) {
                            float v = 0.f;
                            if (!AM::dead(i, 8 * rh)) v = *reinterpret_cast<const float*>(rp + (size_t)lo + (32 * j + 16 * c2) * 4);
                            acc[i][j][8 * rh + 4 * c2 + r4] = v * iscale * iscale2;
                        }
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int row = am.row_base(i, 8 * rh + r4) + rl;
                const bool rin = !AM::dead(i, 8 * rh) && m0 + row < p.M;
                const float* rp = p.addend + (size_t)rowinfo[min(row, BM - 1) * 4 + 3] * p.ldadd;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) {
                        const int col = col0 + 32 * j + 16 * c2;
                        float v = 0.f;
                        if (rin && col < p.K) v = rp[col];
                        acc[i][j][8 * rh + 4 * c2 + r4] = v * iscale * iscale2;
                    }
            }
}

template <int TM, int TN, int BM, int BN, bool SCALED, bool YB, bool BR, class AM>
__device__ __forceinline__ void conv_epilogue4(const ConvArgs& p, f32x16 (&acc)[TM][TN], const AM& am, const int* rowinfo,
                                               int m0, int tile_m, float oscale, float oscale2) {
    constexpr int NC = 2;
    const int lane = am.lane, n0 = am.n0;
    if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * oscale * oscale2;
    }
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.K);
    {
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
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bias_v[j][AM::c2_of(r)];
    }
    const int col0 = am.col(0, 0);  // the lane's first column; column (j, c2) = col0 + 32 j + 16 c2
    const int rl = am.row_lane();
    // lane part of a row's element offset in a tensor with `ld` elements per row: whole tiles -- one value; otherwise clamped so
    // that the row stays inside M (ru = the clamped scalar row)
    auto lane_rows = [&](auto full_tag, int ru, int ld) -> unsigned {
        if constexpr (decltype(full_tag)::value) return (unsigned)(rl * ld);
        else return (unsigned)(min(rl, p.M - 1 - ru) * ld);
    };

    // element `imm` (a compile-time constant at every call) past byte offset `off` (32 bits: the lane's part of an address is a
    // few rows) from the scalar `base`: written in bytes so that the instruction is  global_load/store v_off, s[base] offset:imm
    // -- with element indices the compiler widens (index << 2) to 64 bits per access
    auto at = [](auto* base, unsigned off, int imm) {
        using E = std::remove_pointer_t<decltype(base)>;
        using B = std::conditional_t<std::is_const_v<E>, const char, char>;
        return reinterpret_cast<E*>(reinterpret_cast<B*>(base) + (size_t)off + imm * (int)sizeof(E));
    };

    // ---- BatchNorm-backward partial sums of the layer that produced this convolution's input (ConvArgs::br_y) ---------------
    // before the stores: its loads would otherwise queue behind them.  Loads are unconditional (clamped), issued for one 32-row
    // block at a time, the arithmetic follows.
    if constexpr (BR) {  // (the kernel instantiation of the launches with ConvArgs::br_y)
        auto reduce = [&](auto full_tag, auto ybf_tag, auto bits_tag) {
            constexpr bool FULL = decltype(full_tag)::value, YBF = decltype(ybf_tag)::value, BITS = decltype(bits_tag)::value;
            using YT = std::conditional_t<YBF, __bf16, float>;
            const YT* ybase = reinterpret_cast<const YT*>(p.br_y);
            const int kq = p.K >> 2;
            int cc[TN][NC];     // clamped column (only read when !FULL)
            bool cok[TN][NC];
            float sc[TN][NC], sh[TN][NC], mu[TN][NC], is[TN][NC];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) {
                    const int col = col0 + 32 * j + 16 * c2;
                    cok[j][c2] = FULL || col < p.K;
                    cc[j][c2] = FULL ? col : min(col, p.K - 1);
                    sc[j][c2] = p.br_scale[cc[j][c2]];
                    sh[j][c2] = p.br_shift[cc[j][c2]];
                    mu[j][c2] = p.br_mean[cc[j][c2]];
                    is[j][c2] = p.br_invstd[cc[j][c2]];
                }
            float b1[TN][NC], b2[TN][NC];
            unsigned bm[TN][NC];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) { b1[j][c2] = b2[j][c2] = 0.f; bm[j][c2] = 0u; }
            // One block of rows at a time: its loads, then its arithmetic (the compiler is free to start the next block's loads under
            // this block's arithmetic).  A block is 32 rows (8 rows x 4 columns per lane: 32 + 32 loads in flight) -- 16 rows under
            // the 224-row tile, whose 128 accumulator registers leave no room for more.
            constexpr int NB = TM > 2 ? 2 : 1, RH = 2 / NB;
#pragma unroll
            for (int ib = 0; ib < TM * NB; ++ib) {
                const int i = ib / NB, rh0 = (ib % NB) * RH;
                float yv[RH][4][TN][NC];
                unsigned bb[RH][4][TN][NC];
#pragma unroll
                for (int rq = 0; rq < RH; ++rq) {
                    const int rh = rh0 + rq;
                    if (AM::dead(i, 8 * rh)) continue;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row_u = __builtin_amdgcn_readfirstlane(m0 + am.row_base(i, 8 * rh + r4));
                        const int ru = FULL ? row_u : min(row_u, p.M - 1);
                        const YT* yp = ybase + (size_t)ru * p.br_ldy;
                        const unsigned char* bp = BITS ? p.br_bits + (size_t)ru * kq : nullptr;
                        const unsigned lo = lane_rows(full_tag, ru, p.br_ldy);
                        const unsigned lob = BITS ? lane_rows(full_tag, ru, kq) : 0u;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2) {
                                if constexpr (FULL) {
                                    yv[rq][r4][j][c2] = (float)*at(yp, (lo + (unsigned)col0) * (unsigned)sizeof(YT), 32 * j + 16 * c2);
                                    if constexpr (BITS) bb[rq][r4][j][c2] = *at(bp, lob + (unsigned)(col0 >> 2), 8 * j + 4 * c2);
                                } else {
                                    yv[rq][r4][j][c2] = (float)*at(yp, (lo + (unsigned)cc[j][c2]) * (unsigned)sizeof(YT), 0);
                                    if constexpr (BITS) bb[rq][r4][j][c2] = *at(bp, lob + (unsigned)(cc[j][c2] >> 2), 0);
                                }
                                if constexpr (!BITS) bb[rq][r4][j][c2] = 0u;
                            }
                    }
                }
#pragma unroll
                for (int rq = 0; rq < RH; ++rq) {
                    const int rh = rh0 + rq;
                    if (AM::dead(i, 8 * rh)) continue;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row_u = m0 + am.row_base(i, 8 * rh + r4);
                        const bool rin = FULL || row_u + rl < p.M;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2) {
                                const float y = yv[rq][r4][j][c2];
                                bool on;
                                if constexpr (BITS) on = (bb[rq][r4][j][c2] >> (col0 & 3)) & 1u;
                                else on = !p.br_relu || y * sc[j][c2] + sh[j][c2] > 0.f;
                                const float du = (rin && cok[j][c2] && on) ? acc[i][j][8 * rh + 4 * c2 + r4] : 0.f;
                                b1[j][c2] += du;
                                b2[j][c2] += du * ((y - mu[j][c2]) * is[j][c2]);
                                bm[j][c2] = max(bm[j][c2], __float_as_uint(du) & 0x7fffffffu);
                            }
                    }
                }
                if constexpr (TM > 2) __builtin_amdgcn_sched_barrier(0);
            }
            const int rep_i = tile_m % HN_STAT_REPLICAS;
            double* rep = p.br_sums + (size_t)rep_i * (2 * p.K + 1);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int c2 = 0; c2 < NC; ++c2) {
                    double d1 = (double)b1[j][c2], d2 = (double)b2[j][c2];
                    unsigned m = bm[j][c2];
                    d1 += __shfl_xor(d1, 32);
                    d2 += __shfl_xor(d2, 32);
                    m = max(m, (unsigned)__shfl_xor((int)m, 32));
                    d1 += __shfl_xor(d1, 16);
                    d2 += __shfl_xor(d2, 16);
                    m = max(m, (unsigned)__shfl_xor((int)m, 16));
                    if ((lane >> 4) == 0 && cok[j][c2]) {
                        const int col = col0 + 32 * j + 16 * c2;
                        unsafeAtomicAdd(rep + col, d1);
                        unsafeAtomicAdd(rep + p.K + col, d2);
                        if (p.br_gmax != nullptr && m != 0u) atomicMax(p.br_gmax + (size_t)rep_i * p.K + col, m);
                    }
                }
        };
        auto by_format = [&](auto full_tag) {
            if (p.br_ybf) {
                if (p.br_bits != nullptr) reduce(full_tag, std::true_type{}, std::true_type{});
                else reduce(full_tag, std::true_type{}, std::false_type{});
            } else {
                if (p.br_bits != nullptr) reduce(full_tag, std::false_type{}, std::true_type{});
                else reduce(full_tag, std::false_type{}, std::false_type{});
            }
        };
        if (full) by_format(std::true_type{});
        else by_format(std::false_type{});
    }

    // ---- activation, statistics, store ------------------------------------------------------------------------------------------
    // branch-free activation: out = v > 0 ? v : v * neg_slope  (relu: 0, leaky / prelu: slope; none: the value itself, ACT = false)
    const float neg_slope = p.act == HN_ACT_RELU ? 0.f : p.slope;
    auto finish = [&](auto full_tag, auto map_tag, auto act_tag, auto* ybase) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool OMAP = decltype(map_tag)::value;  // rows land on interleaved pixels (rowinfo[.][3]): no activation / statistics
        constexpr bool ACT = decltype(act_tag)::value;
        using OT = std::remove_pointer_t<decltype(ybase)>;
        float s1[TN][NC], s2[TN][NC], vmax[TN][NC], vmin[TN][NC];
        int cc[TN][NC];
        bool cok[TN][NC];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2) {
                s1[j][c2] = s2[j][c2] = 0.f;
                vmax[j][c2] = -__builtin_inff();
                vmin[j][c2] = __builtin_inff();
                const int col = col0 + 32 * j + 16 * c2;
                cok[j][c2] = FULL || col < p.K;
                cc[j][c2] = FULL ? col : min(col, p.K - 1);
            }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                if (AM::dead(i, 8 * rh)) continue;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rb = am.row_base(i, 8 * rh + r4);  // row inside the tile, without the lane's part
                    if constexpr (OMAP) {
                        const int row = rb + rl;
                        const bool rin = m0 + row < p.M;
                        OT* rp = ybase + (size_t)rowinfo[min(row, BM - 1) * 4 + 3] * p.ldy;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2)
                                if (rin && cok[j][c2]) rp[cc[j][c2]] = (OT)acc[i][j][8 * rh + 4 * c2 + r4];
                    } else {
                        const int row_u = __builtin_amdgcn_readfirstlane(m0 + rb);
                        const int ru = FULL ? row_u : min(row_u, p.M - 1);
                        OT* rp = ybase + (size_t)ru * p.ldy;
                        const unsigned lo = lane_rows(full_tag, ru, p.ldy);
                        const bool rin = FULL || row_u + rl < p.M;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2) {
                                const float v = acc[i][j][8 * rh + 4 * c2 + r4];
                                if (FULL || (rin && cok[j][c2])) {
                                    s1[j][c2] += v;
                                    s2[j][c2] += v * v;
                                    vmax[j][c2] = fmaxf(vmax[j][c2], v);
                                    vmin[j][c2] = fminf(vmin[j][c2], v);
                                    const float o = ACT ? (v > 0.f ? v : v * neg_slope) : v;
                                    if constexpr (FULL) *at(rp, (lo + (unsigned)col0) * (unsigned)sizeof(OT), 32 * j + 16 * c2) = (OT)o;
                                    else *at(rp, (lo + (unsigned)cc[j][c2]) * (unsigned)sizeof(OT), 0) = (OT)o;
                                }
                            }
                    }
                }
            }
        }
        if constexpr (!OMAP) {
            if (p.stats != nullptr) {
                // <= 64 rows per lane were summed in fp32; everything across tiles is accumulated in double
                // (global_atomic_add_f64), like the double accumulators of the CPU reference's BatchNorm.
                const int rep_i = tile_m % HN_STAT_REPLICAS;  // spread same-address atomics
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) {
                        const int col = col0 + 32 * j + 16 * c2;
                        double d1 = (double)s1[j][c2], d2 = (double)s2[j][c2];
                        d1 += __shfl_xor(d1, 32);
                        d2 += __shfl_xor(d2, 32);
                        d1 += __shfl_xor(d1, 16);
                        d2 += __shfl_xor(d2, 16);
                        if ((lane >> 4) == 0 && cok[j][c2]) {
                            double* rep = p.stats + (size_t)rep_i * 2 * p.K;
                            unsafeAtomicAdd(rep + col, d1);
                            unsafeAtomicAdd(rep + p.K + col, d2);
                        }
                        if (p.minmax != nullptr) {  // range of the channel: keys of max(y) and max(-y), see hn_range_key
                            float vx = vmax[j][c2], vn = vmin[j][c2];
                            vx = fmaxf(vx, __shfl_xor(vx, 32));
                            vn = fminf(vn, __shfl_xor(vn, 32));
                            vx = fmaxf(vx, __shfl_xor(vx, 16));
                            vn = fminf(vn, __shfl_xor(vn, 16));
                            if ((lane >> 4) == 0 && cok[j][c2]) {
                                unsigned* rk = p.minmax + (size_t)rep_i * 2 * p.K;
                                atomicMax(rk + col, hn_range_key(vx));
                                atomicMax(rk + p.K + col, hn_range_key(-vn));
                            }
                        }
                    }
            }
        }
    };
    auto by_type = [&](auto full_tag, auto map_tag, auto act_tag) {
        if constexpr (YB) {
            if (p.y_bf16) { finish(full_tag, map_tag, act_tag, reinterpret_cast<__bf16*>(p.y)); return; }
        }
        finish(full_tag, map_tag, act_tag, p.y);
    };
    if (p.omap) {  // parity class of a strided data gradient (no activation / statistics on this path; fp32 output)
        finish(std::false_type{}, std::true_type{}, std::false_type{}, p.y);
    } else if (p.act == HN_ACT_NONE) {
        if (full) by_type(std::true_type{}, std::false_type{}, std::false_type{});
        else by_type(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        if (full) by_type(std::true_type{}, std::false_type{}, std::true_type{});
        else by_type(std::false_type{}, std::false_type{}, std::true_type{});
    }
}

}  // namespace hnconv
