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
//
// MATH_X3 ("bf16x3"): the same fp32 product computed on the bf16 matrix cores.  Every fp32 operand is split exactly into
// three bf16 terms (x = hi + mid + lo, round-to-nearest at each level) when it is staged into LDS, and the six products
// whose weight is >= 2^-24 of the result (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi; each exact in fp32) are
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16, smallest terms first.  Measured against fp64 this is as accurate as
// the fp32 MFMA (profiles/r01_bf16x3_probe.txt: rms error 0.38-0.57 x 2^-24 of sum|products| vs 0.44-0.50) at 16/6
// of its matrix-core rate.  Not a reduced-precision mode: nothing of the fp32 operands is dropped above 2^-24.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "conv_common.h"

#ifdef HN_EXP_PIN
#define HN_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define HN_PIN() ((void)0)
#endif

namespace {

// K step (reduction indices per LDS stage): 32 -> 76 KB of LDS per 128x128 workgroup, 2 workgroups per CU;
// 16 -> 43 KB, 3 workgroups per CU (finer-grained tail when the grid is a small non-multiple of the CU count).
template <int BK_>
struct KStep {
    static constexpr int BK = BK_;
    static constexpr int LDK = BK_ + 4;      // LDS row stride in floats: 16-B aligned rows, conflict-free ds_read_b128
    static constexpr int PR = BK_ / 4;       // 16-B pieces per staged row
    static constexpr int VROWS = 256 / PR;   // rows staged by one pass of the 256 threads (vector path)
    static constexpr int SROWS = 256 / BK_;  // same, scalar path
};

using namespace hnconv;

// Loader variants.  BUF fetches through buffer descriptors: one SGPR resource per tile, 32-bit byte offsets, and
// out-of-range taps / rows are zero-filled by the hardware range check instead of by selects.
enum : int { MODE_SCALAR = 0, MODE_SCALAR_STRIDED = 1, MODE_VEC = 2, MODE_VEC_STRIDED = 3, MODE_VEC_BUF = 4 };
enum : int { MATH_F32 = 0, MATH_X3 = 1, MATH_H2 = 2, MATH3_BF16 = 3 };  // (MATH3_BF16: plan code of FAM_V3 only)

// STRIDED = the gathered tensor is "dilated" by in_dil > 1 (data gradient of a strided convolution): taps must
// land on multiples of in_dil.  Kept out of the common instantiation because it costs integer divisions per tap.
// WN = waves along the column (output channel) axis: 2 -> four waves in a 2 x 2 grid, block tile 64*TM x 64*TN, two
// workgroups per CU; 4 (bf16x3 only) -> EIGHT waves in a 2 x 4 grid, block tile 64*TM x 128*TN (128 x 256), ONE workgroup
// per CU whose eight waves run the split/store and MFMA phases in lockstep: the A operand is split once per 256 output
// columns (half the split arithmetic per MFMA) and every SIMD's MFMA phase is two waves long against one staging phase.
template <int TM, int TN, int MODE, int BK, int MATH = MATH_F32, int WN = 2>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& p, int block_id, int nblocks) {
    constexpr int NT = 128 * WN;  // threads
    constexpr int LDK = KStep<BK>::LDK, PR = KStep<BK>::PR, VROWS = NT / PR, SROWS = NT / BK;
    constexpr bool VEC = MODE >= MODE_VEC;
    constexpr bool STRIDED = MODE == MODE_SCALAR_STRIDED || MODE == MODE_VEC_STRIDED;
    constexpr bool BUF = MODE == MODE_VEC_BUF;
    constexpr bool H2 = MATH == MATH_H2;         // f16x2: two fp16 planes per operand, three products
    constexpr bool X3 = MATH == MATH_X3 || H2;   // the split arithmetics share loaders, LDS image and schedule
    constexpr int NS = H2 ? 2 : 3;               // planes per operand
    constexpr bool S16 = X3;  // they run on v_mfma_f32_16x16x32_{bf16,f16}: accumulators hold 16x16 sub-blocks (acc_row / acc_col)
    static_assert(!X3 || (BUF && BK == 32), "the split arithmetics are built for the buffer loaders at K step 32");
    static_assert(WN == 2 || MATH == MATH_X3, "the eight-wave tile exists for the bf16x3 arithmetic only");
    constexpr int BM = 64 * TM, BN = 32 * WN * TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                        // f32: [2][BM][LDK]     split: [NS][BM][X3_ROW bytes]
    float* Bs = smem + (X3 ? NS * BM * X3_ROW / 4 : 2 * BM * LDK);  // f32: [2][BN][LDK]  split: [NS][BN][X3_ROW bytes]
    int* rowinfo = reinterpret_cast<int*>(Bs + (X3 ? NS * BN * X3_ROW / 4 : 2 * BN * LDK));  // [BM][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int logical = hn_xcd_remap(block_id, nblocks);
    const int ntiles = p.ntm * p.ntn;
    const int split = logical / ntiles, tile = logical - split * ntiles;
    int tile_m, tile_n;
    conv_tile_of(p, tile, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    conv_fill_rowinfo<BM>(p, rowinfo, m0, tid);
    __syncthreads();

    // ---- operand staging -------------------------------------------------------------------------------
    constexpr int RA = VEC ? BM / VROWS : BM / SROWS;
    // bf16x3 stages B from the pre-split planes: BN rows x 3 planes x 4 pieces of 16 B (8 bf16) per K step
    constexpr int RB = X3 ? BN * 4 * NS / NT : (VEC ? BN / VROWS : BN / SROWS);
    f32x4 va[VEC ? RA : 1], vb[VEC ? RB : 1];
    f32x4 va1[X3 ? RA : 1], vb1[X3 ? RB : 1];  // bf16x3: second staging set (two K steps of loads in flight)
    float sa[VEC ? 1 : RA], sb[VEC ? 1 : RB];
    int ahb[VEC ? RA : 1], awb[VEC ? RA : 1], anb[VEC ? RA : 1];
    long long aoff[VEC ? RA : 1];  // element offset of the row's (h base, w base) pixel, may be negative
    long long boff[VEC ? RB : 1];
    const int c4 = tid % PR, vrow = tid / PR;   // VEC: PR lanes x 16 B cover one row of the K step
    const int sk = tid % BK, srow = tid / BK;   // scalar: BK lanes cover one row of the K step
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const int r = vrow + VROWS * q;
            ahb[q] = rowinfo[r * 4 + 0];
            awb[q] = rowinfo[r * 4 + 1];
            anb[q] = rowinfo[r * 4 + 2];
            aoff[q] = ((long long)anb[q] + (long long)ahb[q] * p.W + awb[q]) * p.ldx + c4 * 4;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int co = n0 + vrow + VROWS * q;
            boff[q] = co < p.K ? (long long)co * p.Kdim + c4 * 4 : -1;
        }
    }
    // BUF: resources rebased to the tile (first image touched / first weight row) so that offsets fit 32 bits
    int aoff32[BUF ? RA : 1];
    unsigned boff32[BUF ? RB : 1];
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    unsigned a_records = 0;
    if constexpr (BUF) {
        const int nb0 = __builtin_amdgcn_readfirstlane(rowinfo[2]);
        const long long a_base = (long long)nb0 * p.ldx * 4;
        const long long a_left = p.x_bytes - a_base;
        a_records = a_left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)a_left;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long long)nb0 * p.ldx, 0, a_records, 0x00020000);
#pragma unroll
        for (int q = 0; q < RA; ++q)
            aoff32[q] = (((anb[q] - nb0) + ahb[q] * p.W + awb[q]) * p.ldx + c4 * 4) * 4;
        if constexpr (X3) {
            const unsigned b_records = (unsigned)(NS * p.wx3_plane * 2);
            rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3), 0, b_records, 0x00020000);
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int id = tid + NT * q, plane = id / (BN * 4), rem = id % (BN * 4);
                const int co = n0 + (rem >> 2);
                boff32[q] = co < p.K ? (unsigned)((plane * p.wx3_plane + (long long)co * p.Kdim + (rem & 3) * 8) * 2)
                                     : b_records;  // rows past Cout: out of range -> zero-filled
            }
        } else {
            const int rows_b = min(BN, p.K - n0);
            rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (long long)n0 * p.Kdim, 0,
                                                       (unsigned)((long long)rows_b * p.Kdim * 4), 0x00020000);
#pragma unroll
            for (int q = 0; q < RB; ++q) boff32[q] = (unsigned)(((vrow + VROWS * q) * p.Kdim + c4 * 4) * 4);
        }
    }

    // K-step position of the NEXT chunk to fetch, advanced incrementally (VEC: a step never straddles a tap
    // because C % 32 == 0).
    int st_kh = 0, st_kw = 0, st_c0 = 0, st_k0 = 0;

    // Predicated loads are branch-free (the address falls back to the tensor base, the value is zeroed by a
    // select) so that a whole K step stays one basic block the scheduler can interleave with the MFMAs.
    auto load_a = [&](auto& dva) {
        if constexpr (VEC) {
            const int dh = st_kh * p.dil, dw = st_kw * p.dil;
            if constexpr (BUF) {
                const int doff = ((dh * p.W + dw) * p.ldx + st_c0) * 4;
#pragma unroll
                for (int q = 0; q < RA; ++q) {
                    const bool ok = ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
                    dva[q] = buffer_load16(rsrc_a, ok ? (unsigned)(aoff32[q] + doff) : a_records, 0);
                }
            } else if constexpr (!STRIDED) {
                const long long doff = ((long long)dh * p.W + dw) * p.ldx + st_c0;
#pragma unroll
                for (int q = 0; q < RA; ++q) {
                    const bool ok = ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
                    const float* src = ok ? p.x + (aoff[q] + doff) : p.x;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    dva[q] = ok ? v : z;
                }
            } else {
#pragma unroll
                for (int q = 0; q < RA; ++q) {
                    int h = ahb[q] + dh, w = awb[q] + dw;
                    bool ok = (h >= 0) & (w >= 0) & (h % p.in_dil == 0) & (w % p.in_dil == 0);
                    h /= p.in_dil;
                    w /= p.in_dil;
                    ok = ok & (h < p.H) & (w < p.W);
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(anb[q] + h * p.W + w) * p.ldx + st_c0 + c4 * 4);
                    dva[q] = v;
                }
            }
        } else {
            const int kg = st_k0 + sk;
            const bool kvalid = kg < p.Kdim;
            const int tap = kg / p.C, ci = kg - tap * p.C;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int r = srow + SROWS * q;
                int h = rowinfo[r * 4 + 0] + dh, w = rowinfo[r * 4 + 1] + dw;
                const int nb = rowinfo[r * 4 + 2];
                bool ok = kvalid;
                if constexpr (STRIDED) {
                    ok = ok & (h >= 0) & (w >= 0) & (h % p.in_dil == 0) & (w % p.in_dil == 0);
                    h /= p.in_dil;
                    w /= p.in_dil;
                }
                ok = ok & ((unsigned)h < (unsigned)p.H) & ((unsigned)w < (unsigned)p.W);
                sa[q] = ok ? p.x[(size_t)(nb + h * p.W + w) * p.ldx + ci] : 0.f;
            }
        }
    };
    auto load_b = [&](auto& dvb) {  // also advances the fetch position
        if constexpr (BUF) {
            // rows past Cout are beyond the resource's extent: zero-filled by the range check
#pragma unroll
            for (int q = 0; q < RB; ++q) dvb[q] = buffer_load16(rsrc_b, boff32[q], (unsigned)(st_k0 * (X3 ? 2 : 4)));
            st_c0 += BK;
            if (st_c0 >= p.C) {
                st_c0 = 0;
                if (++st_kw == p.KW) { st_kw = 0; ++st_kh; }
            }
        } else if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const bool ok = boff[q] >= 0;
                const float* src = ok ? p.w + (boff[q] + st_k0) : p.w;
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                dvb[q] = ok ? v : z;
            }
            st_c0 += BK;
            if (st_c0 >= p.C) {
                st_c0 = 0;
                if (++st_kw == p.KW) { st_kw = 0; ++st_kh; }
            }
        } else {
            const int kg = st_k0 + sk;
            const bool kvalid = kg < p.Kdim;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int co = n0 + srow + SROWS * q;
                sb[q] = (kvalid && co < p.K) ? p.w[(size_t)co * p.Kdim + kg] : 0.f;
            }
        }
        st_k0 += BK;
    };

    // f16x2: operand scales (powers of two): the gathered tensor is multiplied by a_s as it is split, the weight planes
    // were made from w * w_s; accumulators hold a_s * w_s times the result
    float a_s = 1.f, o_s = 1.f, i_s = 1.f, o_s2 = 1.f, i_s2 = 1.f;
    if constexpr (H2) {
        float a_inv;
        hn_h2_scales_of_slot(p.a_scale, a_s, a_inv);  // from the tensor's max |x| bit patterns
        o_s = a_inv;
        o_s2 = p.w_scale[1];
        i_s = a_s;
        i_s2 = p.w_scale[0];
    }
    auto store_lds = [&](int buf, const auto& sva, const auto& svb) {
        if constexpr (X3) {  // single LDS stage: split each staged 16 B (4 consecutive k) into 3 x 8 B of bf16
            char* a = reinterpret_cast<char*>(As) + (c4 & 1) * 8;  // + the (possibly flipped) 16-B k group of the row
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int row = vrow + VROWS * q;
                const int kg = ((c4 >> 1) ^ x3_kflip(row)) * 16;
                if constexpr (H2) {
                    f16x4 h, l;
                    split2h(sva[q], a_s, h, l);
                    *reinterpret_cast<f16x4*>(a + (0 * BM + row) * X3_ROW + kg) = h;
                    *reinterpret_cast<f16x4*>(a + (1 * BM + row) * X3_ROW + kg) = l;
                } else {
                    bf16x4 h, m, l;
                    split3(sva[q], h, m, l);
                    *reinterpret_cast<bf16x4*>(a + (0 * BM + row) * X3_ROW + kg) = h;
                    *reinterpret_cast<bf16x4*>(a + (1 * BM + row) * X3_ROW + kg) = m;
                    *reinterpret_cast<bf16x4*>(a + (2 * BM + row) * X3_ROW + kg) = l;
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {  // already split: piece id -> (plane, row, 16-B piece), see boff32
                const int id = tid + NT * q;
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(Bs) + (id >> 2) * X3_ROW +
                                          ((id & 3) ^ x3_kflip(id >> 2)) * 16) = svb[q];
            }
            return;
        }
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RA; ++q) *reinterpret_cast<f32x4*>(a + (vrow + VROWS * q) * LDK + c4 * 4) = sva[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) *reinterpret_cast<f32x4*>(b + (vrow + VROWS * q) * LDK + c4 * 4) = svb[q];
        } else {
#pragma unroll
            for (int q = 0; q < RA; ++q) a[(srow + SROWS * q) * LDK + sk] = sa[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) b[(srow + SROWS * q) * LDK + sk] = sb[q];
        }
    };

    // ---- main loop ---------------------------------------------------------------------------------------
    const int lr = lane & 31, lh = lane >> 5;
    const AccMap<TM, TN, S16> am{wm, wn, lane, n0};
    f32x16 acc[TM][TN];
    conv_init_acc<TM, TN, S16>(p, acc, am, rowinfo, m0, i_s, i_s2);
    const int total_chunks = (p.Kdim + BK - 1) / BK;
    const int kc_first = p.cps > 0 ? split * p.cps : 0;
    const int nchunks = p.cps > 0 ? min(p.cps, total_chunks - kc_first) : total_chunks;  // >= 1 by construction
    st_k0 = kc_first * BK;
    if constexpr (VEC) {
        const int tap0 = st_k0 / p.C;
        st_c0 = st_k0 - tap0 * p.C;
        st_kh = tap0 / p.KW;
        st_kw = tap0 - st_kh * p.KW;
    }

    // Fragments are double-buffered in registers: the ds_read_b128s of group g+1 are issued before the 16 MFMAs of
    // group g, so LDS latency is exposed once per K step (after the barrier) instead of once per group.
    struct Frag {
        f32x4 a[TM], b[TN];
    };
    auto read_frag = [&](int buf, int kk, Frag& f) {
        const float* a = As + buf * BM * LDK + (wm * TM * 32 + lr) * LDK + lh * 4 + kk * 8;
        const float* b = Bs + buf * BN * LDK + (wn * TN * 32 + lr) * LDK + lh * 4 + kk * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK);
    };
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };

    // Software pipeline, two LDS buffers: while step kc multiplies buffer kc&1, the staging registers (chunk kc+1,
    // fetched during step kc-1) are written to the other buffer first, then the fetch of chunk kc+2 is issued in
    // two halves between the MFMA groups -- every global load has a full K step to land before it is needed.
    auto k_step = [&](int buf, auto write_tag, auto fetch_tag) {
        constexpr int G = BK / 8;  // MFMA groups per step
        Frag f[2];
        read_frag(buf, 0, f[0]);
#ifndef HN_EXP_NOWRITE
        if constexpr (decltype(write_tag)::value) store_lds(buf ^ 1, va, vb);
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) read_frag(buf, g + 1, f[(g + 1) & 1]);
#ifndef HN_EXP_NOLOAD
            if constexpr (decltype(fetch_tag)::value) {
                if (g == (G >= 4 ? 1 : 0)) load_a(va);
                if (g == (G >= 4 ? 2 : G - 1)) load_b(vb);
            }
#endif
            mfma16(f[g & 1]);
        }
#ifndef HN_EXP_NOBARRIER
        __syncthreads();
#endif
    };

    if constexpr (X3) {
        // One LDS stage, register prefetch: fetch chunk kc+1 | split+store chunk kc | barrier | fragments and
        // 6 * 4*TM*TN MFMAs | barrier.  The two workgroups of a CU run out of phase, so one wave's
        // split/store (VALU, LDS) overlaps the other's MFMAs on the same SIMD.
        // v_mfma_f32_16x16x32_bf16 (measured 2-5 % faster than the 32x32x16 form of the same loop on random data: the chip
        // holds a higher clock on it): one instruction spans the whole K step; the B fragments of the wave's 2*TN column
        // blocks stay in registers while the 2*TM row blocks stream through
        auto x3_k32 = [&]() {
            const int kg = ((lane >> 4) ^ x3_kflip(lane & 15)) * 16;  // every 16-row block starts at a multiple of 16 rows
            const char* a = reinterpret_cast<const char*>(As) + ((wm * TM * 32 + (lane & 15)) * X3_ROW + kg);
            const char* b = reinterpret_cast<const char*>(Bs) + ((wn * TN * 32 + (lane & 15)) * X3_ROW + kg);
            using frag_t = std::conditional_t<H2, f16x8, bf16x8>;
            frag_t fb[2 * TN][NS];
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
                for (int s3 = 0; s3 < NS; ++s3) fb[bj][s3] = *reinterpret_cast<const frag_t*>(b + (s3 * BN + bj * 16) * X3_ROW);
#pragma unroll
            for (int bi = 0; bi < 2 * TM; ++bi) {
                frag_t fa[NS];
#pragma unroll
                for (int s3 = 0; s3 < NS; ++s3) fa[s3] = *reinterpret_cast<const frag_t*>(a + (s3 * BM + bi * 16) * X3_ROW);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj) {
                    const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                    f32x16& blk = acc[bi >> 1][bj >> 1];
                    f32x4 c = {blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
                    if constexpr (H2) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1], fb[bj][0], c, 0, 0, 0);  // lo*hi
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][1], c, 0, 0, 0);  // hi*lo
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][0], c, 0, 0, 0);  // hi*hi
                    } else {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][1], c, 0, 0, 0);  // mid*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][2], c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[bj][0], c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][1], c, 0, 0, 0);  // hi*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][0], c, 0, 0, 0);  // mid*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][0], c, 0, 0, 0);  // hi*hi
                    }
                    blk[sb] = c[0];
                    blk[sb + 1] = c[1];
                    blk[sb + 2] = c[2];
                    blk[sb + 3] = c[3];
                }
            }
        };
        // Two staging sets, fetch distance two K steps: chunk kc is split and stored from its set, and that set is then
        // refilled at once with chunk kc+2 -- so a fetch has the rest of this step and ALL of the next one to land
        // (a fetch issued one step ahead was what the waves waited on most: 39 % of the wave cycles in
        // SQ_WAIT_INST_ANY beyond LDS).
        auto x3_step = [&](bool refill, auto& set_a, auto& set_b) {
#ifndef HN_EXP_NOWRITE
            store_lds(0, set_a, set_b);
#endif
#ifndef HN_EXP_NOLOAD
            if (refill) {
                load_a(set_a);
                load_b(set_b);
            }
#endif
            __syncthreads();
            x3_k32();
            __syncthreads();
        };
        load_a(va);
        load_b(vb);
        if (nchunks > 1) {
            load_a(va1);
            load_b(vb1);
        }
        int kc = 0;
        for (; kc + 1 < nchunks; kc += 2) {
            x3_step(kc + 2 < nchunks, va, vb);
            x3_step(kc + 3 < nchunks, va1, vb1);
        }
        if (kc < nchunks) x3_step(false, va, vb);
    } else {
    load_a(va);
    load_b(vb);
    store_lds(0, va, vb);
    if (nchunks > 1) {
        load_a(va);
        load_b(vb);
    }
    __syncthreads();
    int kc = 0;
    for (; kc + 2 < nchunks; ++kc) k_step(kc & 1, std::true_type{}, std::true_type{});
    if (kc + 1 < nchunks) {
        k_step(kc & 1, std::true_type{}, std::false_type{});
        ++kc;
    }
    k_step(kc & 1, std::false_type{}, std::false_type{});
    }

    conv_epilogue<TM, TN, S16, BM, BN, H2>(p, acc, am, rowinfo, m0, tile_m, split, o_s, o_s2);
}

template <int TM, int TN, int MODE, int BK, int MATH = MATH_F32, int WN = 2>
__global__ __launch_bounds__(128 * WN, 2) void conv_igemm_kernel(const ConvArgs p) {
    conv_igemm_body<TM, TN, MODE, BK, MATH, WN>(p, blockIdx.x, gridDim.x);
}

// The four parity classes of a stride-2 data gradient (conv2d_dgrad_impl) as ONE launch: the classes are independent
// convolutions of the same dY into interleaved pixels of dX, each a few workgroups on the critics' small maps -- four
// launches of 6-8 us each were most of such a layer's time.  Workgroups [first[c], first[c + 1]) run class c.
struct ConvArgs4 {
    ConvArgs a[4];
    int first[5];
};

template <int TM, int TN, int MODE, int BK, int MATH = MATH_F32, int WN = 2>
__global__ __launch_bounds__(128 * WN, 2) void conv_igemm_classes_kernel(const ConvArgs4 q) {
    int c = 0;
    while (c < 3 && (int)blockIdx.x >= q.first[c + 1]) ++c;
    conv_igemm_body<TM, TN, MODE, BK, MATH, WN>(q.a[c], (int)blockIdx.x - q.first[c], q.first[c + 1] - q.first[c]);
}

template <int TM, int TN, int MODE, int BK, int MATH = MATH_F32, int WN = 2>
int finish_launch_args(const ConvArgs& a, ConvArgs& p) {  // tile counts, tile order, split count: the launch's workgroups
    constexpr int BM = 64 * TM, BN = 32 * WN * TN;
    p = a;
    p.ntm = (a.M + BM - 1) / BM;
    p.ntn = (a.K + BN - 1) / BN;
    static const int gm_env = getenv("HN_GROUP_M") ? atoi(getenv("HN_GROUP_M")) : 0;  // experiment knob
    p.gm = gm_env > 0 ? gm_env : (p.ntn >= 8 ? 8 : 1);
    if (p.gm > p.ntm) p.gm = p.ntm;
    int splits = 1;
    if (a.cps > 0) {  // a.cps is expressed in 32-wide K steps; rescale to this instantiation's step
        const int total = (a.Kdim + BK - 1) / BK;
        p.cps = a.cps * 32 / BK;
        splits = (total + p.cps - 1) / p.cps;
    }
    return p.ntm * p.ntn * splits;
}

// f16x2, buffer loaders, 64-row tiles: the instantiations the stride-2 classes of the critics run on
template <int TN>
int launch_conv_classes(const ConvArgs* cls, int n, hipStream_t stream) {
    constexpr int TM = 1, BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = 2 * (BM + BN) * X3_ROW + BM * 16;
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_classes_kernel<TM, TN, MODE_VEC_BUF, 32, MATH_H2, 2>), lds,
                              configured) != HN_OK)
        return HN_ELAUNCH;
    ConvArgs4 q;
    q.first[0] = 0;
    for (int c = 0; c < 4; ++c) {
        if (c < n) q.first[c + 1] = q.first[c] + finish_launch_args<TM, TN, MODE_VEC_BUF, 32, MATH_H2, 2>(cls[c], q.a[c]);
        else { q.a[c] = q.a[0]; q.first[c + 1] = q.first[c]; }
    }
    HN_LAUNCH((conv_igemm_classes_kernel<TM, TN, MODE_VEC_BUF, 32, MATH_H2, 2>), dim3(q.first[4]), dim3(256), lds, stream, q);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

template <int TM, int TN, int MODE, int BK, int MATH = MATH_F32, int WN = 2>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 32 * WN * TN;
    constexpr int lds = MATH != MATH_F32 ? (MATH == MATH_H2 ? 2 : 3) * (BM + BN) * X3_ROW + BM * 16
                                        : (2 * (BM + BN) * KStep<BK>::LDK + BM * 4) * (int)sizeof(float);
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_kernel<TM, TN, MODE, BK, MATH, WN>), lds, configured) != HN_OK)
        return HN_ELAUNCH;
    // (tile order, measured in profiles/r01_tile_order.txt: 8-row groups gain 4-6 % once there are >= 8 column tiles -- the B
    // panels of one group then outgrow an XCD's L2 in row-major order -- and lose 1-2 % below that)
    ConvArgs p;
    const int nblocks = finish_launch_args<TM, TN, MODE, BK, MATH, WN>(a, p);
    HN_LAUNCH((conv_igemm_kernel<TM, TN, MODE, BK, MATH, WN>), dim3(nblocks), dim3(128 * WN), lds, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

template <int MODE, int BK, int MATH = MATH_F32>
int dispatch_tile(const ConvArgs& a, bool tall, bool wide, hipStream_t stream, int wn = 2) {
    if constexpr (MATH == MATH_X3) {
        if (wn == 4) return launch_conv<2, 2, MODE, BK, MATH, 4>(a, stream);  // 128 x 256, eight waves
    }
    if (tall) return wide ? launch_conv<2, 2, MODE, BK, MATH>(a, stream) : launch_conv<2, 1, MODE, BK, MATH>(a, stream);
    return wide ? launch_conv<1, 2, MODE, BK, MATH>(a, stream) : launch_conv<1, 1, MODE, BK, MATH>(a, stream);
}

constexpr int VARIANT_F16X2 = 4 << 6;  // math_sel of the f16x2 arithmetic (variant codes, plan_conv)
bool g_deterministic = false;  // hn_set_deterministic: no split-K float atomics in these kernels
// experiment knob HN_CLASSES_ONE_LAUNCH=0: the parity classes of a stride-2 data gradient as four launches again
const bool g_classes_one_launch = !(getenv("HN_CLASSES_ONE_LAUNCH") && atoi(getenv("HN_CLASSES_ONE_LAUNCH")) == 0);
int g_conv_math = 0;  // HN_MATH_* default for variant codes that leave the math field 0 (hn_set_conv_math)

// What a launch will run: the kernel family, its tile and arithmetic.
enum : int { FAM_V1 = 0, FAM_V2 = 1, FAM_V3 = 2, FAM_V4 = 3 };
struct ConvPlan {
    int family;      // FAM_V1: conv_igemm_kernel (this file); FAM_V2: conv_igemm2_kernel (conv_igemm2.hip); FAM_V3: conv_igemm3_kernel
    bool tall, wide;
    int wn;          // FAM_V1 bf16x3: 4 = the eight-wave 128 x 256 tile (tn_sel 3), else 2
    int mode, bk;    // FAM_V1: loader variant and K step
    int math;        // FAM_V1: MATH_F32 / MATH_X3 / MATH_H2; FAM_V2: bf16 planes per operand (3 = bf16x3, 1 = bf16)
    int epi = 0;     // FAM_V4: the kernel's EPI argument (1: BatchNorm-backward reduction in the epilogue)
};

// variant: 0 = heuristics; otherwise (bk_sel) + 4 * (tm_sel) + 16 * (tn_sel) + 64 * (math_sel).
//   math_sel 1 -> fp32 MFMA, 2 -> bf16x3, 3 -> bf16, 4 -> f16x2 (0: library default, hn_set_conv_math)
//   bk_sel   fp32 MFMA: 1 -> K step 32, 2 -> 16.  bf16x3: 1 -> first schedule (two workgroups per CU, phases
//            separated by barriers), 2 -> second schedule (producer / consumer waves, conv_igemm2.hip), 3 -> third
//            schedule (128 x 256 tile, weights by LDS-DMA, conv_igemm3.hip; where it does not cover: the first); 0 = the first.
//   tm_sel / tn_sel 1 -> 64-wide, 2 -> 128-wide tile side; tn_sel 3 (bf16x3, first schedule): the 128 x 256 eight-wave tile.
// Within one math mode all variants give bitwise identical results: the per-element accumulation order does not depend
// on the tiling or the schedule.
int plan_conv(const ConvArgs& a, int variant, ConvPlan& pl) {
    if (variant < 0 || variant >= 64 * 5) return HN_EINVAL;
    const int bk_sel = variant & 3, tm_sel = (variant >> 2) & 3, tn_sel = (variant >> 4) & 3, math_sel = variant >> 6;
    if (tm_sel == 3 && a.xh == nullptr) return HN_EINVAL;  // (plane-tensor launches: the low four bits are a tile code)
    const bool vec = (a.C % 32 == 0) && (a.ldx % 4 == 0) && hn_aligned16(a.x) && hn_aligned16(a.w);
    bool wide = a.K > 64;
    if (tn_sel) wide = tn_sel >= 2;
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.K + (wide ? 127 : 63)) / (wide ? 128 : 64));
    bool tall = blocks128 >= 256;  // enough 128-row tiles to cover the chip; else halve the tile
    if (tm_sel) tall = tm_sel == 2;
    if (a.cps > 0) tall = false;   // split-K plans are made for 64-row tiles (plan_split_k)
    const bool strided = a.in_dil > 1;
    // buffer path: offsets are 32-bit relative to the first image a tile touches (a tile may reach into the next
    // image) and to the tile's first weight row
    const long long image_bytes = (long long)a.H * a.W * a.ldx * 4;
    const bool buf_ok = 3 * image_bytes < (1ll << 31) && 128ll * a.Kdim * 4 < (1ll << 31) &&
                        (long long)a.H * a.W >= 128;
    // the bf16 kernels need the pre-split weight planes and the buffer loaders; anything else runs on the fp32 MFMA
    const int math = math_sel ? math_sel - 1 : g_conv_math;  // HN_MATH_*
    if (bk_sel == 3 && math == HN_MATH_F32) return HN_EINVAL;  // (the LDS-DMA schedule needs weight planes)
    const bool planes_ok = a.wx3 != nullptr && vec && !strided && buf_ok && hn_aligned16(a.wx3) &&
                           a.wx3_plane % 8 == 0 && 6 * a.wx3_plane < (1ll << 32);
    pl.tall = tall;
    pl.wide = wide;
    pl.wn = 2;
    pl.bk = 32;
    if (a.xh != nullptr) {
        // the gathered operand arrives as fp16 planes: only the all-DMA schedule (conv_igemm4.hip) reads that format.
        // variant & 15 = its tile (0: 128 x 128, or 128 x 64 when there are at most 64 output channels; 1 .. 6: launch_conv4)
        // (fp16 hi / lo planes with scales under f16x2, one bf16 plane without under bf16)
        if (!(math == HN_MATH_F16X2 && a.a_scale != nullptr) && !(math == HN_MATH_BF16 && a.a_scale == nullptr)) return HN_ESHAPE;
        if (!conv4_covers(a)) return HN_ESHAPE;
        pl.family = FAM_V4; pl.math = math == HN_MATH_BF16 ? MATH3_BF16 : MATH_H2; pl.mode = MODE_VEC_BUF;
        const int tile = variant & 15;
        if (tile > 8) return HN_EINVAL;
        pl.wn = tile == 0 ? (a.K <= 64 ? 1 : 0) : tile;
        pl.epi = a.br_y != nullptr ? 1 : 0;
        return HN_OK;
    }
    if (math == HN_MATH_F16X2) {
        // the caller chose the operand format (fp16 planes + scales): there is no other kernel to fall back to
        if (!planes_ok || a.a_scale == nullptr) return HN_ESHAPE;
        pl.family = FAM_V1; pl.math = MATH_H2; pl.mode = MODE_VEC_BUF;
        if (bk_sel == 2) {  // the wave-specialised schedule (conv_igemm2.hip)
            pl.family = FAM_V2; pl.math = MATH2_H2;
            pl.bk = (a.C % 64 == 0 && a.cps == 0) ? 64 : 32;
        } else if (bk_sel == 3 && conv3_covers(a, 2, tn_sel == 3)) {  // weights by LDS-DMA (conv_igemm3.hip)
            pl.family = FAM_V3; pl.math = MATH_H2; pl.tall = true; pl.wide = true; pl.wn = tn_sel == 3 ? 4 : 2;
        }
    } else if (math == HN_MATH_BF16 && planes_ok) {
        pl.family = FAM_V2; pl.math = MATH2_BF16; pl.mode = MODE_VEC_BUF;
        pl.bk = a.C % 64 == 0 ? 64 : 32;
        if (bk_sel == 3 && conv3_covers(a, 1, tn_sel == 3)) {  // weights by LDS-DMA (conv_igemm3.hip)
            pl.family = FAM_V3; pl.math = MATH3_BF16; pl.tall = pl.wide = true; pl.wn = tn_sel == 3 ? 4 : 2; pl.bk = 32;
        }
    } else if (math == HN_MATH_BF16X3 && planes_ok) {
        pl.mode = MODE_VEC_BUF;
        // default: the first schedule -- measured 2-10 % faster than the wave-specialised one on every layer shape of the
        // network (profiles/r02_conv_schedules.txt)
        if (bk_sel == 2) { pl.family = FAM_V2; pl.math = MATH2_X3; }
        else if (bk_sel == 3 && conv3_covers(a, 3, true)) { pl.family = FAM_V3; pl.math = MATH_X3; pl.tall = pl.wide = true; }
        else {
            pl.family = FAM_V1; pl.math = MATH_X3;
            if (tn_sel == 3 && a.cps == 0) { pl.wn = 4; pl.tall = pl.wide = true; }  // 128 x 256 tile, eight waves
        }
    } else {
        pl.family = FAM_V1; pl.math = MATH_F32;
        if (vec && !strided && buf_ok) { pl.mode = MODE_VEC_BUF; pl.bk = bk_sel == 2 ? 16 : 32; }
        else if (vec) pl.mode = strided ? MODE_VEC_STRIDED : MODE_VEC;
        else pl.mode = strided ? MODE_SCALAR_STRIDED : MODE_SCALAR;
    }
    if (a.in_scale != nullptr && pl.family != FAM_V2) return HN_ESHAPE;  // the A-operand transform lives in the v2 loader
    return HN_OK;
}

int run_conv(const ConvArgs& a, const ConvPlan& pl, hipStream_t stream) {
    if (pl.family == FAM_V4) return launch_conv4(a, pl.wn, stream);
    if (pl.family == FAM_V3) {
        const int planes = pl.math == MATH_H2 ? 2 : (pl.math == MATH3_BF16 ? 1 : 3);
        return launch_conv3(a, planes, pl.wn == 4 || planes == 3, stream);
    }
    if (pl.family == FAM_V2) {
        const int rc = launch_conv2(a, pl.tall, pl.wide, pl.math, stream);
        return rc == 1 ? HN_ESHAPE : rc;
    }
    if (pl.math == MATH_X3) return dispatch_tile<MODE_VEC_BUF, 32, MATH_X3>(a, pl.tall, pl.wide, stream, pl.wn);
    if (pl.math == MATH_H2) return dispatch_tile<MODE_VEC_BUF, 32, MATH_H2>(a, pl.tall, pl.wide, stream);
    switch (pl.mode) {
        case MODE_VEC_BUF:
            return pl.bk == 16 ? dispatch_tile<MODE_VEC_BUF, 16>(a, pl.tall, pl.wide, stream)
                               : dispatch_tile<MODE_VEC_BUF, 32>(a, pl.tall, pl.wide, stream);
        case MODE_VEC_STRIDED: return dispatch_tile<MODE_VEC_STRIDED, 32>(a, pl.tall, pl.wide, stream);
        case MODE_VEC: return dispatch_tile<MODE_VEC, 32>(a, pl.tall, pl.wide, stream);
        case MODE_SCALAR_STRIDED: return dispatch_tile<MODE_SCALAR_STRIDED, 32>(a, pl.tall, pl.wide, stream);
        default: return dispatch_tile<MODE_SCALAR, 32>(a, pl.tall, pl.wide, stream);
    }
}

int dispatch_conv(const ConvArgs& a, int variant, hipStream_t stream) {
    ConvPlan pl;
    const int rc = plan_conv(a, variant, pl);
    if (rc != HN_OK) return rc;
    return run_conv(a, pl, stream);
}

// the kernel's name as rocprofv3 prints it (tools/pmc_aggregate.py keys), for per-instantiation accounting
void plan_name(const ConvPlan& pl, char* buf, int n) {
    const int tm = pl.tall ? 2 : 1, tn = pl.wide ? 2 : 1;
    if (pl.family == FAM_V4) snprintf(buf, n, "%s, %d>", conv4_kernel_name(pl.wn, pl.math == MATH3_BF16), pl.epi);
    else if (pl.family == FAM_V3) {
        const int planes = pl.math == MATH_H2 ? 2 : (pl.math == MATH3_BF16 ? 1 : 3);
        const bool small = planes != 3 && pl.wn != 4;  // the 128 x 128 four-wave tile with two A staging sets
        snprintf(buf, n, "conv_igemm3_kernel<%d, %d, %d>", planes, small ? 2 : 4, small ? 2 : 1);
    }
    else if (pl.family == FAM_V2) snprintf(buf, n, "conv_igemm2_kernel<%d, %d, %d, %d, false>", tm, tn, pl.math, pl.bk);
    else if (pl.math == MATH_X3) snprintf(buf, n, "conv_igemm_kernel<%d, %d, %d, %d, 1, %d>", tm, tn, pl.mode, pl.bk, pl.wn);
    else if (pl.math == MATH_H2) snprintf(buf, n, "conv_igemm_kernel<%d, %d, %d, %d, 2, 2>", tm, tn, pl.mode, pl.bk);
    else snprintf(buf, n, "conv_igemm_kernel<%d, %d, %d, %d, 0, 2>", tm, tn, pl.mode, pl.bk);
}

// Split-K plan: K steps (of 32) per split, or 0.  Used when the output has so few tiles that the chip would idle
// (pyramid levels, critic tails: a few hundred pixels) while the reduction is long.
int plan_split_k(int M, int K, int Kdim) {
    const int total = (Kdim + 31) / 32;
    if (total < 8) return 0;
    const int tiles = ((M + 63) / 64) * ((K + (K > 64 ? 127 : 63)) / (K > 64 ? 128 : 64));
    // more rows only when the tiles still leave most of the chip idle under a very long reduction: the critics' first
    // convolutions (4x4 stride 2 over 1024 / 2048-channel features: 110 tiles, 512 / 1024 K steps ran at 43 TFLOP/s unsplit)
    if (M > 2048 && !(tiles <= 192 && total >= 128)) return 0;
    int splits = 512 / tiles;
    if (splits > total / 4) splits = total / 4;
    if (splits < 2) return 0;
    return (total + splits - 1) / splits;
}

// y = act(y + bias[c]) over npix x C (after a split-K accumulation)
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, int ld, const float* __restrict__ bias, int npix,
                                                      int C, float neg_slope) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * C) return;
    const int pix = idx / C, c = idx - pix * C;
    float v = y[(size_t)pix * ld + c] + (bias != nullptr ? bias[c] : 0.f);
    y[(size_t)pix * ld + c] = v > 0.f ? v : v * neg_slope;
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

// Tap table of the data-gradient weight layout.  Stride 1 (or a stride the parity decomposition does not cover):
// one block wt[Cin][T][Cout] with the taps flipped.  Stride 2, odd dilation: four consecutive blocks, one per
// output-parity class (ph, pw) in the order (0,0) (0,1) (1,0) (1,1), block = [Cin][Th*Tw][Cout] holding only the taps
// that reach that class, flipped within the class.
struct TapTable {
    int src[64];   // source tap index kh*KW+kw of destination slot z
    int dst[64];   // float offset of element (ci = 0, slot, co = 0) of that slot
    int row[64];   // floats between consecutive ci rows of the slot's block (T_class * Cout)
    int count;
};

struct ClassGeom {
    int k0, T, c;  // first tap, tap count, (ph + pad - k0*dil) / 2
};

inline ClassGeom class_geom(int K, int pad, int dil, int parity) {
    ClassGeom g{0, 0, 0};
    int first = -1;
    for (int k = 0; k < K; ++k)
        if (((parity + pad - k * dil) & 1) == 0) {
            if (first < 0) first = k;
            ++g.T;
        }
    g.k0 = first < 0 ? 0 : first;
    g.c = first < 0 ? 0 : (parity + pad - g.k0 * dil) / 2;  // exact: numerator is even (may be negative)
    return g;
}

inline bool parity_decomposable(int KH, int KW, int stride, int dil) {
    return stride == 2 && (dil & 1) == 1 && KH * KW <= 64;
}

inline void build_tap_table(TapTable& t, int Cout, int KH, int KW, int Cin, int stride, int pad, int dil) {
    const int T = KH * KW;
    t.count = T;
    if (!parity_decomposable(KH, KW, stride, dil)) {
        for (int z = 0; z < T && z < 64; ++z) {
            t.src[z] = T - 1 - z;
            t.dst[z] = z * Cout;
            t.row[z] = T * Cout;
        }
        return;
    }
    int slot = 0, base = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            const ClassGeom gh = class_geom(KH, pad, dil, ph), gw = class_geom(KW, pad, dil, pw);
            const int Tc = gh.T * gw.T;
            for (int a = 0; a < gh.T; ++a)
                for (int b = 0; b < gw.T; ++b) {  // (a, b) = flipped class-local tap
                    const int kh = gh.k0 + 2 * (gh.T - 1 - a), kw = gw.k0 + 2 * (gw.T - 1 - b);
                    t.src[slot] = kh * KW + kw;
                    t.dst[slot] = base + (a * gw.T + b) * Cout;
                    t.row[slot] = Tc * Cout;
                    ++slot;
                }
            base += Cin * Tc * Cout;
        }
}

__device__ __forceinline__ void split3_scalar(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 hh = (__bf16)v;
    const float r = v - (float)hh;
    const __bf16 mm = (__bf16)r;
    const __bf16 ll = (__bf16)(r - (float)mm);
    h = __builtin_bit_cast(unsigned short, hh);
    m = __builtin_bit_cast(unsigned short, mm);
    l = __builtin_bit_cast(unsigned short, ll);
}

__device__ __forceinline__ void split2h_scalar(float v, float s, unsigned short& h, unsigned short& l) {
    const float xs = v * s;
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)(xs - (float)hh);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}

// ---- f16x2 operand preparation ------------------------------------------------------------------------------------
// Weight planes in the f16x2 format: [hi plane][lo plane][area], plane = HN_X3_PLANE(n) fp16 elements; the area (the
// bf16x3 format's third plane) holds the scale slot {s, 1/s, -, -} and H2_PARTS partial maxima.  Two passes without
// atomics or persistent state: absmax over H2_PARTS chunks, then every block of the splitting kernel reduces the
// partials itself (bitwise reproducible, nothing to zero).
constexpr int H2_PARTS = 64;
constexpr long long H2_MIN_ELEMS = 256;  // the area must hold 4 + H2_PARTS floats: plane * 2 bytes >= 272

__device__ __forceinline__ float* h2_area(unsigned short* planes, long long plane) {
    return reinterpret_cast<float*>(planes + 2 * plane);
}

// chunk `part` (of H2_PARTS) of a matrix of n floats -> area[4 + part] = max |w| as uint bits
__device__ __forceinline__ void h2_absmax_part(const float* __restrict__ w, long long n, int part, float* area) {
    const long long len = ((n + H2_PARTS - 1) / H2_PARTS + 3) & ~3ll;
    const long long lo = part * len, hi = min(n, lo + len);
    unsigned m = 0u;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
    __shared__ unsigned red[4];
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<unsigned*>(area)[4 + part] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// every thread of the block gets the scale; `publish`: this block also writes the slot(s) the convolution kernels read
__device__ __forceinline__ float h2_scale_of_parts(float* area, bool publish, float* also = nullptr) {
    __shared__ float s_sh[2];
    if (threadIdx.x < 64) {
        unsigned m = reinterpret_cast<const unsigned*>(area)[4 + threadIdx.x];
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) {
            hn_h2_scales(m, s_sh[0], s_sh[1]);
            if (publish) {
                area[0] = s_sh[0];
                area[1] = s_sh[1];
                if (also != nullptr && also != area) { also[0] = s_sh[0]; also[1] = s_sh[1]; }
            }
        }
    }
    __syncthreads();
    return s_sh[0];
}

// One 32x32 (co, ci) tile of one tap, transposed through LDS.  `wt` (fp32) and `wt_x3` (planes of `plane` elements each,
// same element order) are both optional; `w_x3` (optional) receives the planes of the UNtransposed matrix -- every source
// element passes through exactly one block, so the forward operand comes for free.  H2: two fp16 planes of w * s instead
// of three bf16 planes.
template <bool H2>
__device__ __forceinline__ void prep_tile(const float* __restrict__ w, unsigned short* __restrict__ w_x3, float* __restrict__ wt,
                                          unsigned short* __restrict__ wt_x3, long long plane, int Cout, int T, int Cin, int src,
                                          int dst, int rowstride, int ci0, int co0, float s) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const size_t o = ((size_t)co * T + src) * Cin + ci;
            v = w[o];
            if (w_x3 != nullptr) {
                if constexpr (H2) split2h_scalar(v, s, w_x3[o], w_x3[plane + o]);
                else split3_scalar(v, w_x3[o], w_x3[plane + o], w_x3[2 * plane + o]);
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (wt == nullptr && wt_x3 == nullptr) return;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) {
            const size_t o = (size_t)dst + (size_t)ci * rowstride + co;
            const float v = tile[tx][r];
            if (wt != nullptr) wt[o] = v;
            if (wt_x3 != nullptr) {
                if constexpr (H2) split2h_scalar(v, s, wt_x3[o], wt_x3[plane + o]);
                else split3_scalar(v, wt_x3[o], wt_x3[plane + o], wt_x3[2 * plane + o]);
            }
        }
    }
}

// the scale area of a layer's f16x2 planes lives in the forward planes when there are any
__device__ __forceinline__ float* h2_layer_area(unsigned short* w_x3, unsigned short* wt_x3, long long plane) {
    return h2_area(w_x3 != nullptr ? w_x3 : wt_x3, plane);
}

__global__ __launch_bounds__(256) void weight_taps_transpose_kernel(const float* __restrict__ w,
                                                                   unsigned short* __restrict__ w_x3, float* __restrict__ wt,
                                                                   unsigned short* __restrict__ wx3, long long plane,
                                                                   int Cout, int T, int Cin, int format, TapTable t) {
    const int z = blockIdx.z;
    if (format == 1) {
        const bool first = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
        const float s = h2_scale_of_parts(h2_layer_area(w_x3, wx3, plane), first, wx3 != nullptr ? h2_area(wx3, plane) : nullptr);
        prep_tile<true>(w, w_x3, wt, wx3, plane, Cout, T, Cin, t.src[z], t.dst[z], t.row[z], blockIdx.x * 32, blockIdx.y * 32, s);
    } else {
        prep_tile<false>(w, w_x3, wt, wx3, plane, Cout, T, Cin, t.src[z], t.dst[z], t.row[z], blockIdx.x * 32, blockIdx.y * 32, 1.f);
    }
}

// All layers of a model in ONE launch: the same tile body, with the layer looked up from the block index (a step prepares
// ~90 weight matrices; as separate launches they cost ~7 us each, mostly latency).
struct PrepLayer {
    const float* w;
    unsigned short* w_x3;
    float* wt;
    unsigned short* wt_x3;
    long long plane;
    int Cout, T, Cin;
    int first_block, gx, gy;  // this layer's blocks are [first_block, first_block + gx * gy * T)
    int format;               // 0: three bf16 planes; 1: two fp16 planes + scale (f16x2)
    TapTable t;
};

__global__ __launch_bounds__(256) void weight_prepare_batch_kernel(const PrepLayer* __restrict__ layers, int n) {
    int lo = 0, hi = n - 1;  // last layer whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (layers[mid].first_block <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const PrepLayer& L = layers[lo];
    const int local = blockIdx.x - L.first_block;
    const int bx = local % L.gx, by = (local / L.gx) % L.gy, z = local / (L.gx * L.gy);
    if (L.format == 1) {
        const float s = h2_scale_of_parts(h2_layer_area(L.w_x3, L.wt_x3, L.plane), local == 0,
                                          L.wt_x3 != nullptr ? h2_area(L.wt_x3, L.plane) : nullptr);
        prep_tile<true>(L.w, L.w_x3, L.wt, L.wt_x3, L.plane, L.Cout, L.T, L.Cin, L.t.src[z], L.t.dst[z], L.t.row[z], bx * 32,
                        by * 32, s);
    } else {
        prep_tile<false>(L.w, L.w_x3, L.wt, L.wt_x3, L.plane, L.Cout, L.T, L.Cin, L.t.src[z], L.t.dst[z], L.t.row[z], bx * 32,
                         by * 32, 1.f);
    }
}

// pass 1 of the f16x2 layers of a batch: H2_PARTS blocks per layer (block = layer * H2_PARTS + part)
__global__ __launch_bounds__(256) void weight_absmax_batch_kernel(const PrepLayer* __restrict__ layers) {
    const PrepLayer& L = layers[blockIdx.x / H2_PARTS];
    if (L.format != 1) return;
    h2_absmax_part(L.w, (long long)L.Cout * L.T * L.Cin, blockIdx.x % H2_PARTS, h2_layer_area(L.w_x3, L.wt_x3, L.plane));
}

__global__ __launch_bounds__(256) void weight_absmax_kernel(const float* __restrict__ w, long long n, float* area) {
    h2_absmax_part(w, n, blockIdx.x, area);
}

__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                         long long n, long long plane) {
    const float s = h2_scale_of_parts(h2_area(dst, plane), blockIdx.x == 0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        split2h_scalar(src[i], s, dst[i], dst[plane + i]);
}

__global__ __launch_bounds__(256) void absmax_scale_kernel(const float* __restrict__ x, int ld, long long npix, int C,
                                                          float* __restrict__ slot) {
    unsigned m = 0u;
    if (ld == C && (C & 3) == 0) {  // dense: 16-B loads over the flat tensor
        const long long n4 = npix * C / 4;
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
        }
    } else {
        const long long n = npix * C;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long pix = i / C;
            m = max(m, __float_as_uint(x[pix * ld + (i - pix * C)]) & 0x7fffffffu);
        }
    }
    hn_absmax_publish(m, slot);
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                          long long n, long long plane) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        split3_scalar(src[i], dst[i], dst[plane + i], dst[2 * plane + i]);
}

}  // namespace

namespace {
// scale slot of an f16x2 plane buffer: behind the two planes
inline const float* h2_scale_of(const void* planes, long long plane) {
    return reinterpret_cast<const float*>(static_cast<const unsigned short*>(planes) + 2 * plane);
}
}  // namespace

namespace {
ConvArgs fwd_args(const float* x, const float* w, const void* w_x3, const float* bias, float* y, double* stats,
                  const HnConvDesc* d, int act, float slope) {
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.addend = nullptr; a.ldadd = 0;
    a.H = d->Hin; a.W = d->Win; a.C = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.K = d->Cout; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = a.pad_w = d->pad; a.dil = d->dil; a.in_dil = 1;
    a.omap = 0; a.oH = a.oW = a.oph = a.opw = 0; a.ostep = 1; a.cps = 0;
    a.M = d->N * d->Hout * d->Wout;
    a.Kdim = d->KH * d->KW * d->Cin;
    a.ntm = a.ntn = 0; a.gm = 1;
    a.act = act; a.slope = slope;
    a.x_bytes = (((long long)d->N * d->Hin * d->Win - 1) * d->ldx + d->Cin) * 4;
    a.wx3 = static_cast<const unsigned short*>(w_x3);
    a.wx3_plane = hn_x3_plane((long long)d->Cout * a.Kdim);
    a.in_scale = a.in_shift = nullptr;
    a.a_scale = a.w_scale = nullptr;
    return a;
}
}  // namespace

extern "C" int hn_set_deterministic(int on) {
    g_deterministic = on != 0;
    return HN_OK;
}

extern "C" int hn_get_deterministic(void) { return g_deterministic ? 1 : 0; }

extern "C" int hn_conv2d_fwd_affine(const float* x, const float* in_scale, const float* in_shift, const float* w,
                                    const void* w_x3, const float* bias, float* y, double* stats, const HnConvDesc* d,
                                    int variant, void* stream) {
    if (x == nullptr || w == nullptr || y == nullptr || in_scale == nullptr || in_shift == nullptr) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    if (!hn_aligned16(in_scale) || !hn_aligned16(in_shift)) return HN_EALIGN;
    ConvArgs a = fwd_args(x, w, w_x3, bias, y, stats, d, HN_ACT_NONE, 0.f);
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    return dispatch_conv(a, variant, static_cast<hipStream_t>(stream));  // HN_ESHAPE when no kernel with the transform covers it
}

extern "C" int hn_conv2d_plan(int kind, const HnConvDesc* d, int has_planes, int variant, char* name, int name_len) {
    if (kind < 0 || kind > 1 || name == nullptr || name_len <= 0) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    // operand pointers only enter the plan through their alignment: stand-ins with the alignment the host guarantees
    const float* al = reinterpret_cast<const float*>(uintptr_t(256));
    ConvArgs a = fwd_args(al, al, has_planes ? al : nullptr, nullptr, const_cast<float*>(al), nullptr, d, HN_ACT_NONE, 0.f);
    if (variant >= VARIANT_F16X2) a.a_scale = a.w_scale = al;
    if (has_planes & 2) {  // the gathered operand as fp16 planes too (hn_conv2d_fwd_planes / hn_conv2d_dgrad_planes)
        a.xh = reinterpret_cast<const unsigned short*>(al);
        a.xh_plane = 1ll << 20;
    }
    if (has_planes & 4) a.br_y = al;  // hn_conv2d_dgrad_planes_bnreduce: the instantiation with the reduction in its epilogue
    bool can_split = d->ldy == d->Cout;
    if (kind == 1) {  // data gradient: the gathered tensor is dY, the produced one dX (first parity class when strided)
        const bool classes = parity_decomposable(d->KH, d->KW, d->stride, d->dil);
        a.H = d->Hout; a.W = d->Wout; a.C = d->Cout; a.ldx = d->ldy; a.K = d->Cin; a.ldy = d->ldx;
        a.stride = 1; a.Ho = d->Hin; a.Wo = d->Win; a.M = d->N * d->Hin * d->Win; a.Kdim = d->KH * d->KW * d->Cout;
        a.in_dil = classes ? 1 : d->stride;
        if (classes) { a.Ho = (d->Hin + 1) / 2; a.Wo = (d->Win + 1) / 2; a.M = d->N * a.Ho * a.Wo; a.Kdim = ((d->KH + 1) / 2) * ((d->KW + 1) / 2) * d->Cout; }
        a.wx3_plane = hn_x3_plane((long long)d->Cin * d->KH * d->KW * d->Cout);
        a.x_bytes = (((long long)d->N * d->Hout * d->Wout - 1) * d->ldy + d->Cout) * 4;
        can_split = d->ldx == d->Cin && (d->stride == 1 || classes);
    }
    a.cps = (can_split && !g_deterministic && !(has_planes & 2)) ? plan_split_k(a.M, a.K, a.Kdim) : 0;
    ConvPlan pl;
    const int rp = plan_conv(a, a.cps > 0 ? (variant & ~3) : variant, pl);
    if (rp != HN_OK) return rp;
    plan_name(pl, name, name_len);
    return pl.family == FAM_V2 ? 1 : 0;
}

namespace {
int conv2d_fwd_impl(const float* x, const float* x_scale, const float* w, const void* w_x3, const float* bias, float* y,
                    double* stats, const HnConvDesc* d, int act, float slope, int variant, void* stream,
                    const void* xh = nullptr, long long xh_plane = 0, unsigned* minmax = nullptr) {
    if ((x == nullptr && xh == nullptr) || (w == nullptr && xh == nullptr) || y == nullptr) return HN_EINVAL;
    const bool y_bf16 = (act & HN_Y_BF16) != 0;  // (one bf16 plane in, bf16 y out: hn_conv2d_fwd_planes only)
    act &= ~HN_Y_BF16;
    if (y_bf16 && (xh == nullptr || x_scale != nullptr || act != HN_ACT_NONE || d == nullptr || d->ldy % 2 != 0)) return HN_EINVAL;
    if (act < HN_ACT_NONE || act > HN_ACT_LEAKY) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    ConvArgs a = fwd_args(x, w, w_x3, bias, y, stats, d, act, slope);
    if (x_scale != nullptr) {
        if (w_x3 == nullptr || variant < 0) return HN_EINVAL;
        a.a_scale = x_scale;
        a.w_scale = h2_scale_of(w_x3, a.wx3_plane);
        variant = (variant & 63) | VARIANT_F16X2;
    } else if (variant >= VARIANT_F16X2) {
        return HN_EINVAL;  // f16x2 needs the operand scales: hn_conv2d_fwd_scaled
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.minmax = minmax;
    if (minmax != nullptr && (stats == nullptr || act != HN_ACT_NONE)) return HN_EINVAL;
    if (xh != nullptr) {  // plane tensor: the all-DMA kernels only (no split-K on that path)
        if (xh_plane <= 0 || w_x3 == nullptr) return HN_EINVAL;
        if (x_scale == nullptr) variant = (variant & 63) | (3 << 6);  // one bf16 plane: the mixed-precision bf16 arithmetic
        a.y_bf16 = y_bf16 ? 1 : 0;
        a.xh = static_cast<const unsigned short*>(xh);
        a.xh_plane = xh_plane;
        a.x_bytes = (((long long)d->N * d->Hin * d->Win - 1) * d->ldx + d->Cin) * 2;
        return dispatch_conv(a, variant, s);
    }
    a.cps = (d->ldy == d->Cout && !g_deterministic) ? plan_split_k(a.M, a.K, a.Kdim) : 0;
    if (a.cps > 0 && minmax != nullptr) a.cps = 0;  // (the range keys come from the kernel epilogue)
    if (a.cps == 0) return dispatch_conv(a, variant, s);
    // split-K: zero y, accumulate partial products, then bias / activation and (for BatchNorm) the statistics
    if (hipMemsetAsync(y, 0, (size_t)a.M * a.K * sizeof(float), s) != hipSuccess) return HN_ELAUNCH;
    a.bias = nullptr; a.stats = nullptr; a.act = HN_ACT_NONE;
    const int rcs = dispatch_conv(a, variant & ~3, s);  // split-K plans are made for the default K step / schedule
    if (rcs != HN_OK) return rcs;
    if (bias != nullptr || act != HN_ACT_NONE) {
        const float neg_slope = act == HN_ACT_NONE ? 1.f : (act == HN_ACT_RELU ? 0.f : slope);
        HN_LAUNCH(bias_act_kernel, dim3((a.M * a.K + 255) / 256), dim3(256), 0, s, y, d->ldy, bias, a.M, a.K, neg_slope);
        HN_CHECK_LAUNCH();
    }
    if (stats != nullptr) {
        if (act != HN_ACT_NONE) return HN_EINVAL;  // statistics are defined on the pre-activation output
        return hn_channel_stats(y, d->ldy, a.M, a.K, stats, stream);
    }
    return HN_OK;
}
}  // namespace

extern "C" int hn_conv2d_fwd(const float* x, const float* w, const void* w_x3, const float* bias, float* y, double* stats,
                             const HnConvDesc* d, int act, float slope, int variant, void* stream) {
    return conv2d_fwd_impl(x, nullptr, w, w_x3, bias, y, stats, d, act, slope, variant, stream);
}

extern "C" int hn_conv2d_fwd_scaled(const float* x, const float* x_scale, const float* w, const void* w_h2, const float* bias,
                                    float* y, double* stats, const HnConvDesc* d, int act, float slope, int variant,
                                    void* stream) {
    if (x_scale == nullptr || w_h2 == nullptr) return HN_EINVAL;
    return conv2d_fwd_impl(x, x_scale, w, w_h2, bias, y, stats, d, act, slope, variant, stream);
}

extern "C" int hn_conv2d_fwd_planes(const void* x_planes, int64_t plane_elems, const float* x_scale, const void* w_h2,
                                    const float* bias, float* y, double* stats, uint32_t* range_keys, const HnConvDesc* d,
                                    int act, float slope, int variant, void* stream) {
    if (x_planes == nullptr || w_h2 == nullptr) return HN_EINVAL;  // (x_scale == NULL: one bf16 plane, w_h2 = bf16x3 planes)
    return conv2d_fwd_impl(nullptr, x_scale, nullptr, w_h2, bias, y, stats, d, act, slope, variant, stream, x_planes, plane_elems,
                           range_keys);
}

extern "C" int hn_conv2d_fwd_scaled_range(const float* x, const float* x_scale, const float* w, const void* w_h2,
                                          const float* bias, float* y, double* stats, uint32_t* range_keys, const HnConvDesc* d,
                                          int variant, void* stream) {
    if (x_scale == nullptr || w_h2 == nullptr || range_keys == nullptr) return HN_EINVAL;
    return conv2d_fwd_impl(x, x_scale, w, w_h2, bias, y, stats, d, HN_ACT_NONE, 0.f, variant, stream, nullptr, 0, range_keys);
}

namespace {
// dX = addend, or zero: what the accumulating (split-K / partially covered) data-gradient paths start from
int init_output(float* dx, int ldx, const float* addend, int ldadd, int64_t npix, int C, void* stream) {
    if (addend != nullptr) return addend == dx ? HN_OK : hn_copy(addend, ldadd, dx, ldx, npix, C, stream);
    if (ldx != C) return HN_ESHAPE;  // zero-filling a channel slice is not needed by any caller
    if (hipMemsetAsync(dx, 0, (size_t)npix * C * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess)
        return HN_ELAUNCH;
    return HN_OK;
}
}  // namespace

namespace {
int conv2d_dgrad_impl(const float* dy, const float* dy_scale, const float* wt, const void* wt_x3, float* dx, const float* addend,
                      int ldadd, const HnConvDesc* d, int variant, void* stream, const void* dyh = nullptr,
                      long long dyh_plane = 0, const HnBnReduceArgs* br = nullptr) {
    if ((dy == nullptr && dyh == nullptr) || (wt == nullptr && dyh == nullptr) || dx == nullptr) return HN_EINVAL;
    if (br != nullptr) {
        // the BatchNorm-backward reduction rides in the epilogue of the dense stride-1 plane-tensor launch only
        if (dyh == nullptr || d == nullptr || d->stride != 1 || d->ldx != d->Cin) return HN_ESHAPE;
        if (!br->y || !br->scale || !br->shift || !br->mean || !br->invstd || !br->sums || br->ldy < d->Cin) return HN_EINVAL;
        if (br->act != HN_ACT_NONE && br->act != HN_ACT_RELU) return HN_EINVAL;
        if (br->relu_bits != nullptr && (d->Cin % 4 != 0 || br->act != HN_ACT_RELU)) return HN_EINVAL;
        // the epilogue reads y and the four per-channel vectors 16 bytes (four channels) per lane (conv_epilogue_t.h)
        if (d->Cin % 4 != 0 || br->ldy % 4 != 0 || !hn_aligned16(dx) || !hn_aligned16(br->scale) || !hn_aligned16(br->shift) ||
            !hn_aligned16(br->mean) || !hn_aligned16(br->invstd) ||
            (reinterpret_cast<uintptr_t>(br->y) & (br->y_bf16 ? 7u : 15u)) != 0)
            return HN_ESHAPE;
    }
    if (dyh != nullptr && (wt_x3 == nullptr || dyh_plane <= 0)) return HN_EINVAL;
    if (dyh != nullptr && dy_scale == nullptr) variant = (variant & 63) | (3 << 6);  // one bf16 plane
    if (dy_scale != nullptr) {
        if (wt_x3 == nullptr || variant < 0) return HN_EINVAL;
        variant = (variant & 63) | VARIANT_F16X2;
    } else if (variant >= VARIANT_F16X2) {
        return HN_EINVAL;
    }
    if (addend != nullptr && (d == nullptr || ldadd < d->Cin)) return HN_EINVAL;
    const int rc = check_desc(d);
    if (rc != HN_OK) return rc;
    if (d->dil * (d->KH - 1) < d->pad || d->dil * (d->KW - 1) < d->pad) return HN_ESHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ConvArgs a;
    a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx; a.stats = nullptr; a.addend = addend; a.ldadd = ldadd;
    a.H = d->Hout; a.W = d->Wout; a.C = d->Cout; a.ldx = d->ldy;      // gathered tensor: dY
    a.K = d->Cin; a.ldy = d->ldx;                                     // produced tensor: dX
    a.stride = 1; a.dil = d->dil;
    a.ntm = a.ntn = 0;
    a.act = HN_ACT_NONE; a.slope = 0.f;
    a.x_bytes = (((long long)d->N * d->Hout * d->Wout - 1) * d->ldy + d->Cout) * 4;
    a.wx3 = static_cast<const unsigned short*>(wt_x3);
    a.wx3_plane = hn_x3_plane((long long)d->Cin * d->KH * d->KW * d->Cout);
    a.in_scale = a.in_shift = nullptr;
    a.a_scale = dy_scale;
    a.w_scale = dy_scale != nullptr && wt_x3 != nullptr ? h2_scale_of(wt_x3, a.wx3_plane) : nullptr;
    a.gm = 1;
    const bool planes_in = dyh != nullptr;
    if (planes_in) {
        a.xh = static_cast<const unsigned short*>(dyh);
        a.xh_plane = dyh_plane;
        a.x_bytes = (((long long)d->N * d->Hout * d->Wout - 1) * d->ldy + d->Cout) * 2;
    }

    if (!parity_decomposable(d->KH, d->KW, d->stride, d->dil)) {
        // stride 1: a plain convolution of dY with the flipped taps; other strides: the same kernel gathering from the
        // zero-dilated gradient (taps must land on multiples of the stride)
        a.Ho = d->Hin; a.Wo = d->Win;
        a.KH = d->KH; a.KW = d->KW; a.in_dil = d->stride;
        a.pad = d->dil * (d->KH - 1) - d->pad;
        a.pad_w = d->dil * (d->KW - 1) - d->pad;
        a.M = d->N * d->Hin * d->Win;
        a.Kdim = d->KH * d->KW * d->Cout;
        a.omap = 0; a.oH = a.oW = a.oph = a.opw = 0; a.ostep = 1;
        a.cps = (d->ldx == d->Cin && d->stride == 1 && !g_deterministic && !planes_in) ? plan_split_k(a.M, a.K, a.Kdim) : 0;
        if (br != nullptr) {
            a.br_y = br->y; a.br_ldy = br->ldy; a.br_ybf = br->y_bf16 ? 1 : 0; a.br_relu = br->act == HN_ACT_RELU ? 1 : 0;
            a.br_scale = br->scale; a.br_shift = br->shift; a.br_mean = br->mean; a.br_invstd = br->invstd;
            a.br_bits = br->relu_bits; a.br_sums = br->sums; a.br_gmax = br->du_max;
        }
        if (a.cps > 0) {  // split-K partials add into dX, which therefore starts as the addend (or zero)
            const int rci = init_output(dx, d->ldx, addend, ldadd, (int64_t)a.M, d->Cin, stream);
            if (rci != HN_OK) return rci;
            a.addend = nullptr;
            variant &= ~3;
        }
        return dispatch_conv(a, variant, s);
    }

    // Stride 2: the pixels of dX split into four parity classes; class (ph, pw) only ever sees the taps with
    // kh = ph + pad (mod 2), kw = pw + pad (mod 2), and over the class's own (Hin/2 x Win/2) grid that is a stride-1
    // convolution of dY -- no multiply-adds against inserted zeros.
    const ClassGeom gh[2] = {class_geom(d->KH, d->pad, d->dil, 0), class_geom(d->KH, d->pad, d->dil, 1)};
    const ClassGeom gw[2] = {class_geom(d->KW, d->pad, d->dil, 0), class_geom(d->KW, d->pad, d->dil, 1)};
    const bool some_empty = gh[0].T == 0 || gh[1].T == 0 || gw[0].T == 0 || gw[1].T == 0;
    // split-K when the classes are tiny (critic tails): every class then ADDS into a pre-initialised dX
    const int Mc = d->N * ((d->Hin + 1) / 2) * ((d->Win + 1) / 2);
    const bool split = d->ldx == d->Cin && !g_deterministic && !planes_in &&
                       plan_split_k(Mc, d->Cin, gh[1].T * gw[1].T * d->Cout) > 0;
    if (some_empty || split) {  // pixels no tap reaches, or accumulation target: dX = addend or 0
        const int rci = init_output(dx, d->ldx, addend, ldadd, (int64_t)d->N * d->Hin * d->Win, d->Cin, stream);
        if (rci != HN_OK) return rci;
    }
    if (split) {
        a.addend = nullptr;
        variant &= ~3;
    }
    long long wofs = 0;
    ConvArgs cls[4];
    int ncls = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            const int Th = gh[ph].T, Tw = gw[pw].T;
            const long long wsize = (long long)d->Cin * Th * Tw * d->Cout;
            const int Hc = (d->Hin - ph + 1) / 2, Wc = (d->Win - pw + 1) / 2;
            if (Th > 0 && Tw > 0 && Hc > 0 && Wc > 0) {
                ConvArgs c = a;
                c.w = wt + wofs;
                c.wx3 = a.wx3 != nullptr ? a.wx3 + wofs : nullptr;
                c.Ho = Hc; c.Wo = Wc;
                c.KH = Th; c.KW = Tw; c.in_dil = 1;
                c.pad = d->dil * (Th - 1) - gh[ph].c;
                c.pad_w = d->dil * (Tw - 1) - gw[pw].c;
                c.M = d->N * Hc * Wc;
                c.Kdim = Th * Tw * d->Cout;
                c.omap = 1; c.oH = d->Hin; c.oW = d->Win; c.ostep = 2; c.oph = ph; c.opw = pw;
                c.cps = 0;
                if (split) {
                    c.cps = plan_split_k(c.M, c.K, c.Kdim);
                    if (c.cps == 0) c.cps = (c.Kdim + 31) / 32;  // one split, but through the accumulating epilogue
                }
                cls[ncls++] = c;
            }
            wofs += wsize;
        }
    // one launch for all classes where they run the same instantiation of the f16x2 64-row kernel (conv_igemm_classes_kernel)
    if (ncls > 1 && g_classes_one_launch) {
        ConvPlan p0;
        bool same = plan_conv(cls[0], variant, p0) == HN_OK && p0.family == FAM_V1 && p0.math == MATH_H2 && !p0.tall &&
                    p0.mode == MODE_VEC_BUF && p0.bk == 32;
        for (int i = 1; i < ncls && same; ++i) {
            ConvPlan pi;
            same = plan_conv(cls[i], variant, pi) == HN_OK && pi.family == p0.family && pi.math == p0.math && pi.tall == p0.tall &&
                   pi.wide == p0.wide && pi.mode == p0.mode && pi.bk == p0.bk;
        }
        if (same) return p0.wide ? launch_conv_classes<2>(cls, ncls, s) : launch_conv_classes<1>(cls, ncls, s);
    }
    for (int i = 0; i < ncls; ++i) {
        const int rcc = dispatch_conv(cls[i], variant, s);
        if (rcc != HN_OK) return rcc;
    }
    return HN_OK;
}
}  // namespace

#ifdef HN_STAMPS
long long* hnconv::g_hn_stamps = nullptr;
extern "C" int hn_debug_set_stamps(long long* buf) { hnconv::g_hn_stamps = buf; return HN_OK; }
#endif

extern "C" int hn_conv2d_dgrad(const float* dy, const float* wt, const void* wt_x3, float* dx, const float* addend, int ldadd,
                               const HnConvDesc* d, int variant, void* stream) {
    return conv2d_dgrad_impl(dy, nullptr, wt, wt_x3, dx, addend, ldadd, d, variant, stream);
}

extern "C" int hn_conv2d_dgrad_scaled(const float* dy, const float* dy_scale, const float* wt, const void* wt_h2, float* dx,
                                      const float* addend, int ldadd, const HnConvDesc* d, int variant, void* stream) {
    if (dy_scale == nullptr || wt_h2 == nullptr) return HN_EINVAL;
    return conv2d_dgrad_impl(dy, dy_scale, wt, wt_h2, dx, addend, ldadd, d, variant, stream);
}

extern "C" int hn_conv2d_dgrad_planes(const void* dy_planes, int64_t plane_elems, const float* dy_scale, const void* wt_h2,
                                      float* dx, const float* addend, int ldadd, const HnConvDesc* d, int variant,
                                      void* stream) {
    if (dy_planes == nullptr || wt_h2 == nullptr) return HN_EINVAL;
    return conv2d_dgrad_impl(nullptr, dy_scale, nullptr, wt_h2, dx, addend, ldadd, d, variant, stream, dy_planes, plane_elems);
}

extern "C" int hn_conv2d_dgrad_planes_bnreduce(const void* dy_planes, int64_t plane_elems, const float* dy_scale, const void* wt_h2,
                                               float* dx, const float* addend, int ldadd, const HnConvDesc* d, int variant,
                                               const HnBnReduceArgs* bn, void* stream) {
    if (dy_planes == nullptr || wt_h2 == nullptr || bn == nullptr) return HN_EINVAL;
    return conv2d_dgrad_impl(nullptr, dy_scale, nullptr, wt_h2, dx, addend, ldadd, d, variant, stream, dy_planes, plane_elems, bn);
}

extern "C" int hn_absmax_scale(const float* x, int ld, int64_t npix, int C, float* scale, void* stream) {
    if (x == nullptr || scale == nullptr || npix <= 0 || C <= 0 || ld < C) return HN_EINVAL;
    const bool dense = ld == C && (C & 3) == 0;
    if (dense && !hn_aligned16(x)) return HN_EALIGN;
    const long long work = dense ? npix * C / 4 : npix * C;
    long long blocks = (work + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    HN_LAUNCH(absmax_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld,
              (long long)npix, C, scale);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_split_f16x2(const float* src, void* dst, int64_t n, void* stream) {
    if (src == nullptr || dst == nullptr || n <= 0) return HN_EINVAL;
    if (n < H2_MIN_ELEMS) return HN_ESHAPE;
    if (!hn_aligned16(dst)) return HN_EALIGN;
    const long long plane = hn_x3_plane(n);
    unsigned short* planes = static_cast<unsigned short*>(dst);
    hipStream_t s = static_cast<hipStream_t>(stream);
    HN_LAUNCH(weight_absmax_kernel, dim3(H2_PARTS), dim3(256), 0, s, src, (long long)n,
              reinterpret_cast<float*>(planes + 2 * plane));
    HN_CHECK_LAUNCH();
    const long long blocks = (n + 255) / 256;
    HN_LAUNCH(split_f16x2_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, src, planes, (long long)n,
              plane);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

namespace {
int weight_flip_transpose_impl(const float* w, void* w_x3, float* wt, void* wt_x3, int Cout, int KH, int KW, int Cin, int stride,
                               int pad, int dil, int format, void* stream) {
    if (w == nullptr || (wt == nullptr && wt_x3 == nullptr) || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0 || stride <= 0 ||
        pad < 0 || dil <= 0)
        return HN_EINVAL;
    if (KH * KW > 64) return HN_ESHAPE;
    const long long n = (long long)Cout * KH * KW * Cin, plane = hn_x3_plane(n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (format == 1) {
        if (w_x3 == nullptr && wt_x3 == nullptr) return HN_EINVAL;
        if (n < H2_MIN_ELEMS) return HN_ESHAPE;
        unsigned short* planes = static_cast<unsigned short*>(w_x3 != nullptr ? w_x3 : wt_x3);
        HN_LAUNCH(weight_absmax_kernel, dim3(H2_PARTS), dim3(256), 0, s, w, n, reinterpret_cast<float*>(planes + 2 * plane));
        HN_CHECK_LAUNCH();
    }
    TapTable t;
    build_tap_table(t, Cout, KH, KW, Cin, stride, pad, dil);
    const dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, KH * KW);
    HN_LAUNCH(weight_taps_transpose_kernel, grid, dim3(256), 0, s, w, static_cast<unsigned short*>(w_x3), wt,
              static_cast<unsigned short*>(wt_x3), plane, Cout, KH * KW, Cin, format, t);
    HN_CHECK_LAUNCH();
    return HN_OK;
}
}  // namespace

extern "C" int hn_weight_flip_transpose(const float* w, void* w_x3, float* wt, void* wt_x3, int Cout, int KH, int KW, int Cin,
                                        int stride, int pad, int dil, void* stream) {
    return weight_flip_transpose_impl(w, w_x3, wt, wt_x3, Cout, KH, KW, Cin, stride, pad, dil, 0, stream);
}

extern "C" int hn_weight_flip_transpose_h2(const float* w, void* w_h2, float* wt, void* wt_h2, int Cout, int KH, int KW, int Cin,
                                           int stride, int pad, int dil, void* stream) {
    return weight_flip_transpose_impl(w, w_h2, wt, wt_h2, Cout, KH, KW, Cin, stride, pad, dil, 1, stream);
}

extern "C" size_t hn_weight_prep_table_bytes(int n) { return (size_t)(n > 0 ? n : 0) * sizeof(PrepLayer); }

extern "C" int hn_weight_prep_table_fill(const HnWeightPrepItem* items, int n, void* table_host, int* total_blocks) {
    if (items == nullptr || table_host == nullptr || total_blocks == nullptr || n <= 0) return HN_EINVAL;
    PrepLayer* out = static_cast<PrepLayer*>(table_host);
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const HnWeightPrepItem& it = items[i];
        if (it.w == nullptr || (it.w_x3 == nullptr && it.wt == nullptr && it.wt_x3 == nullptr) || it.Cout <= 0 || it.KH <= 0 ||
            it.KW <= 0 || it.Cin <= 0 || it.stride <= 0 || it.pad < 0 || it.dil <= 0)
            return HN_EINVAL;
        if (it.KH * it.KW > 64) return HN_ESHAPE;
        if (it.format != 0 && it.format != 1) return HN_EINVAL;
        if (it.format == 1 && (it.w_x3 == nullptr && it.wt_x3 == nullptr)) return HN_EINVAL;
        if (it.format == 1 && (long long)it.Cout * it.KH * it.KW * it.Cin < H2_MIN_ELEMS) return HN_ESHAPE;
        PrepLayer& L = out[i];
        L.format = it.format;
        L.w = it.w;
        L.w_x3 = static_cast<unsigned short*>(it.w_x3);
        L.wt = it.wt;
        L.wt_x3 = static_cast<unsigned short*>(it.wt_x3);
        L.plane = hn_x3_plane((long long)it.Cout * it.KH * it.KW * it.Cin);
        L.Cout = it.Cout; L.T = it.KH * it.KW; L.Cin = it.Cin;
        L.gx = (it.Cin + 31) / 32; L.gy = (it.Cout + 31) / 32;
        L.first_block = (int)blocks;
        build_tap_table(L.t, it.Cout, it.KH, it.KW, it.Cin, it.stride, it.pad, it.dil);
        blocks += (long long)L.gx * L.gy * L.T;
        if (blocks > 0x7fffffffll) return HN_ESHAPE;
    }
    *total_blocks = (int)blocks;
    return HN_OK;
}

extern "C" int hn_weight_prepare_batch(const void* table_dev, int n, int total_blocks, int has_f16x2, void* stream) {
    if (table_dev == nullptr || n <= 0 || total_blocks <= 0) return HN_EINVAL;
    if (has_f16x2) {  // pass 1 of the f16x2 layers: partial maxima of |w|
        HN_LAUNCH(weight_absmax_batch_kernel, dim3((unsigned)n * H2_PARTS), dim3(256), 0, static_cast<hipStream_t>(stream),
                  static_cast<const PrepLayer*>(table_dev));
        HN_CHECK_LAUNCH();
    }
    HN_LAUNCH(weight_prepare_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
              static_cast<const PrepLayer*>(table_dev), n);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_split_bf16x3(const float* src, void* dst, int64_t n, void* stream) {
    if (src == nullptr || dst == nullptr || n <= 0) return HN_EINVAL;
    const long long blocks = (n + 255) / 256;
    HN_LAUNCH(split_bf16x3_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
              static_cast<hipStream_t>(stream), src, static_cast<unsigned short*>(dst), (long long)n, hn_x3_plane(n));
    HN_CHECK_LAUNCH();
    return HN_OK;
}

extern "C" int hn_set_conv_math(int math) {
    if (math != HN_MATH_F32 && math != HN_MATH_BF16X3 && math != HN_MATH_BF16) return HN_EINVAL;
    g_conv_math = math;
    return HN_OK;
}

extern "C" int hn_get_conv_math(void) { return g_conv_math; }
