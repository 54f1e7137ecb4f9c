// Implicit-GEMM convolution forward / data-gradient on the bf16 matrix cores, second schedule: wave-specialised.
//
// Same GEMM view, operand layouts, loaders and epilogue as conv_igemm.hip (MATH_X3); what differs is who does what when.
// There, every wave of a 4-wave workgroup alternates a split/store phase and an MFMA phase around two barriers per K
// step and the CU relies on its second resident workgroup to fill the matrix pipe meanwhile -- measured:
// SQ_VALU_MFMA_BUSY_CYCLES = 52 % of the cycles, 54 % of the wave cycles in SQ_WAIT_INST_ANY
// (profiles/r01_pmc_conv_igemm_bf16x3.txt).  hipcc does not interleave the ~125 vector instructions of a staging step
// with the MFMA chain of ONE wave either (it emits each class in a run, also under sched_group_barrier pipelines), and a
// wave issues in order.  So the two jobs go to different waves and the hardware's issue arbiter does the interleaving:
//
//   workgroup = 8 waves on one CU, two per SIMD, two LDS stages:
//     waves 0-3  CONSUMERS  one 64*TM/2 x 64*TN/2 quarter of the tile each: ds_read_b128 fragments of stage k & 1,
//                           MFMA chain, epilogue.  No global loads, no split arithmetic.
//     waves 4-7  PRODUCERS  fetch chunk k+3 (buffer loads into a staging register set, two steps ahead), split chunk
//                           k+1 into bf16 planes, ds_write it to stage (k+1) & 1.  No MFMAs.
//     one s_barrier per K step.
//
// While a consumer's MFMA occupies the matrix pipe (16 cycles, of which it blocks vector issue for 8) the producer on
// the same SIMD issues its VALU / LDS / VMEM instructions.  Out-of-range chunks are fetched through the buffer range
// check and come back as zeros, so neither loop contains conditional code.
// Arithmetic: NS = 3 -> bf16x3 (six products per fp32 product, identical accumulation order to conv_igemm.hip: results
// are bitwise identical to the first schedule); NS = 1 -> plain bf16 (operands rounded to nearest bf16, fp32 accumulate;
// K step 64 when Cin allows); NS = 2 -> f16x2 (two fp16 planes per operand, per-tensor power-of-two scales, three products:
// the producers' share of a step shrinks from ~200 to ~40 vector instructions per lane and a third of the LDS bytes go, so
// here -- unlike under bf16x3 -- the consumers' MFMA chain is the longest thing in a step).
#include "conv_common.h"

namespace hnconv {
namespace {

// waves per SIMD the register budget is set for: 2 (one 8-wave workgroup per CU, up to 256 registers per lane) for the
// 128x128 tile and wherever the two LDS stages leave room for a single workgroup anyway, 4 otherwise
template <int TM, int TN, int NS, int BK>
constexpr int conv2_waves() { return (TM * TN == 4 || 2 * NS * 64 * (TM + TN) * (2 * BK + 16) > 80 * 1024) ? 2 : 4; }

template <int TM, int TN, int NS, int BK, bool AFF>
__global__ __launch_bounds__(512, (conv2_waves<TM, TN, NS, BK>())) void conv_igemm2_kernel(const ConvArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int ROW = 2 * BK + 16;                 // bytes per LDS row: BK bf16 + 16 B pad (conflict-free, see x3_kflip)
    constexpr int A_PLANE = BM * ROW, B_PLANE = BN * ROW;
    constexpr int STAGE = NS * (A_PLANE + B_PLANE);  // bytes per LDS stage
    constexpr int KH = BK / 32;                      // MFMA k-halves per K step
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    int* rowinfo = reinterpret_cast<int*>(smem2 + 2 * STAGE);  // [BM][4]

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: the role branch below
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3, tid = wave * 64 + lane;                  // index inside the role's four waves
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = p.ntm * p.ntn;
    const int split = logical / ntiles, tile = logical - split * ntiles;
    int tile_m, tile_n;
    conv_tile_of(p, tile, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    conv_fill_rowinfo<BM>(p, rowinfo, m0, tid + (producer ? 256 : 0));  // (rows beyond 256 do not exist: BM <= 128)
    __syncthreads();

    constexpr bool H2 = NS == 2;
    float a_s = 1.f, o_s = 1.f, i_s = 1.f, o_s2 = 1.f, i_s2 = 1.f;  // f16x2 operand scales (conv_igemm.hip)
    if constexpr (H2) {
        float a_inv;
        hn_h2_scales_of_slot(p.a_scale, a_s, a_inv);
        o_s = a_inv;
        o_s2 = p.w_scale[1];
        i_s = a_s;
        i_s2 = p.w_scale[0];
    }
    const int total_chunks = p.Kdim / BK;  // a step never straddles a tap: C % BK == 0
    const int kc_first = p.cps > 0 ? split * p.cps : 0;
    const int nchunks = p.cps > 0 ? min(p.cps, total_chunks - kc_first) : total_chunks;  // >= 1 by construction
    char* stage0 = smem2;
    char* stage1 = smem2 + STAGE;

    if (producer) {
        // ---- producers: global -> registers -> (split) -> LDS ---------------------------------------------
        constexpr int PA = BK / 4;              // 16-B fp32 pieces per A row and K step
        constexpr int RA = BM * PA / 256;       // pieces per thread
        constexpr int AROWS = 256 / PA;         // rows covered by one pass of the 256 producer threads
        constexpr int PB = BK / 8;              // 16-B bf16 pieces per B row, plane and K step
        constexpr int RB = BN * NS * PB / 256;
        const int a_c = tid % PA, a_r = tid / PA;
        struct StageSet {
            f32x4 a[RA];
            f32x4 b[RB];
            f32x4 sc, sh;       // AFF: scale / shift of this thread's 4 channels in this chunk
            unsigned okmask;    // AFF: bit q = piece q came from inside the image
        };
        int ahb[RA], awb[RA], aoff32[RA];
        unsigned boff32[RB];
        const int nb0 = __builtin_amdgcn_readfirstlane(rowinfo[2]);
        const long long a_base = (long long)nb0 * p.ldx * 4;
        const long long a_left = p.x_bytes - a_base;
        const unsigned a_records = a_left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)a_left;
        const __amdgpu_buffer_rsrc_t rsrc_a =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long long)nb0 * p.ldx, 0, a_records, 0x00020000);
        const unsigned b_records = (unsigned)(3 * p.wx3_plane * 2);
        const __amdgpu_buffer_rsrc_t rsrc_b =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3), 0, b_records, 0x00020000);
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const int r = a_r + AROWS * q;
            ahb[q] = rowinfo[r * 4 + 0];
            awb[q] = rowinfo[r * 4 + 1];
            aoff32[q] = (((rowinfo[r * 4 + 2] - nb0) + ahb[q] * p.W + awb[q]) * p.ldx + a_c * 4) * 4;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int id = tid + 256 * q, plane = id / (BN * PB), rem = id % (BN * PB);
            const int co = n0 + rem / PB;
            boff32[q] = co < p.K ? (unsigned)((plane * p.wx3_plane + (long long)co * p.Kdim + (rem % PB) * 8) * 2)
                                 : b_records;  // rows past Cout: out of range -> zero-filled
        }
        __amdgpu_buffer_rsrc_t rsrc_sc, rsrc_sh;
        if constexpr (AFF) {
            rsrc_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_scale), 0, (unsigned)(p.C * 4), 0x00020000);
            rsrc_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_shift), 0, (unsigned)(p.C * 4), 0x00020000);
        }
        // K-step position of the NEXT chunk to fetch, advanced incrementally
        const int kc_end = kc_first + nchunks;
        int st_chunk = kc_first, st_k0 = kc_first * BK;
        int st_kh, st_kw, st_c0;
        {
            const int tap0 = st_k0 / p.C;
            st_c0 = st_k0 - tap0 * p.C;
            st_kh = tap0 / p.KW;
            st_kw = tap0 - st_kh * p.KW;
        }
        auto fetch = [&](StageSet& s) {  // chunk st_chunk -> s (all zeros past the last chunk of this launch); advances
            const bool live = st_chunk < kc_end;
            const int dh = st_kh * p.dil, dw = st_kw * p.dil;
            const int doff = ((dh * p.W + dw) * p.ldx + st_c0) * 4;
            unsigned okm = 0;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const bool ok = live & ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
                s.a[q] = buffer_load16(rsrc_a, ok ? (unsigned)(aoff32[q] + doff) : a_records, 0);
                if constexpr (AFF) okm |= ok ? (1u << q) : 0u;
            }
            if constexpr (AFF) {
                s.okmask = okm;
                s.sc = buffer_load16(rsrc_sc, (unsigned)((st_c0 + a_c * 4) * 4), 0);
                s.sh = buffer_load16(rsrc_sh, (unsigned)((st_c0 + a_c * 4) * 4), 0);
            }
            // (the K advance rides in soffset, which the range check does not see: the out-of-range marker goes in voffset)
#pragma unroll
            for (int q = 0; q < RB; ++q) s.b[q] = buffer_load16(rsrc_b, live ? boff32[q] : b_records, (unsigned)(st_k0 * 2));
            // branch-free position update: a step stays one basic block
            st_c0 += BK;
            const int wrap_c = st_c0 >= p.C ? 1 : 0;
            st_c0 = wrap_c ? 0 : st_c0;
            st_kw += wrap_c;
            const int wrap_w = st_kw == p.KW ? 1 : 0;
            st_kw = wrap_w ? 0 : st_kw;
            st_kh += wrap_w;
            st_k0 += BK;
            ++st_chunk;
        };
        // LDS image of a stage: A planes [NS][BM][ROW], then B planes [NS][BN][ROW]; a row's 16-B k groups are stored
        // with their low index bit flipped for rows 4-11 of every 16 (x3_kflip)
        auto store = [&](char* st, const StageSet& s) {
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int row = a_r + AROWS * q;
                char* dst = st + row * ROW + (((a_c >> 1) ^ x3_kflip(row)) * 16) + (a_c & 1) * 8;
                f32x4 v = s.a[q];
                if constexpr (AFF) {
                    const bool ok = (s.okmask >> q) & 1u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = fmaf(v[e], s.sc[e], s.sh[e]);
                        v[e] = ok ? fmaxf(u, 0.f) : 0.f;
                    }
                }
                if constexpr (NS == 3) {
                    bf16x4 h, m, l;
                    split3(v, h, m, l);
                    *reinterpret_cast<bf16x4*>(dst) = h;
                    *reinterpret_cast<bf16x4*>(dst + A_PLANE) = m;
                    *reinterpret_cast<bf16x4*>(dst + 2 * A_PLANE) = l;
                } else if constexpr (H2) {
                    f16x4 h, l;
                    split2h(v, a_s, h, l);
                    *reinterpret_cast<f16x4*>(dst) = h;
                    *reinterpret_cast<f16x4*>(dst + A_PLANE) = l;
                } else {
                    *reinterpret_cast<bf16x4*>(dst) = round_bf16(v);
                }
            }
            char* bs = st + NS * A_PLANE;
#pragma unroll
            for (int q = 0; q < RB; ++q) {  // already bf16: piece id -> (plane, row, 16-B piece), see boff32
                const int id = tid + 256 * q, plane = id / (BN * PB), rem = id % (BN * PB);
                const int row = rem / PB, piece = rem % PB;
                *reinterpret_cast<f32x4*>(bs + plane * B_PLANE + row * ROW + ((piece ^ x3_kflip(row)) * 16)) = s.b[q];
            }
        };
        // chunk 0 -> stage 0; during consumer step k: chunk k+1 -> stage (k+1) & 1 from the set fetched two steps ago,
        // which is refilled at once with chunk k+3
        StageSet set0, set1;
        fetch(set0);
        fetch(set1);
        store(stage0, set0);
        fetch(set0);
        __syncthreads();
        int kc = 0;
        // (hipcc hoists the second step's split arithmetic above the first barrier, so the pair waits for its loads at
        // the top; pinning the order with sched_barrier(0) restores the two-step fetch distance and measured 5-8 % SLOWER)
        for (; kc + 1 < nchunks; kc += 2) {
            store(stage1, set1);
            fetch(set1);
            __syncthreads();
            store(stage0, set0);
            fetch(set0);
            __syncthreads();
        }
        if (kc < nchunks) __syncthreads();  // (nothing left to stage for the last consumer step)
        return;
    }

    // ---- consumers: LDS -> fragments -> MFMA -> epilogue --------------------------------------------------------
    const AccMap<TM, TN, true> am{wm, wn, lane, n0};
    f32x16 acc[TM][TN];
    conv_init_acc<TM, TN, true>(p, acc, am, rowinfo, m0, i_s, i_s2);

    // lane -> (row lane & 15, k group lane >> 4) of a 16-row block; every block starts at a multiple of 16 rows
    const int frag_off = (lane & 15) * ROW + (((lane >> 4) ^ x3_kflip(lane & 15)) * 16);
    const int a_frag = wm * TM * 32 * ROW + frag_off;
    const int b_frag = NS * A_PLANE + wn * TN * 32 * ROW + frag_off;
    struct FragA { bf16x8 v[NS][KH]; };
    struct FragB { bf16x8 v[2 * TN][NS][KH]; };
    auto read_a = [&](const char* st, int bi, FragA& f) {
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3)
#pragma unroll
            for (int h = 0; h < KH; ++h)
                f.v[s3][h] = *reinterpret_cast<const bf16x8*>(st + a_frag + s3 * A_PLANE + bi * 16 * ROW + h * 64);
    };
    auto read_b = [&](const char* st, FragB& f) {
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3)
#pragma unroll
                for (int h = 0; h < KH; ++h)
                    f.v[bj][s3][h] = *reinterpret_cast<const bf16x8*>(st + b_frag + s3 * B_PLANE + bj * 16 * ROW + h * 64);
    };
    // one 16-row block of the wave's tile against all of its 2*TN column blocks
    auto mfma_block = [&](int bi, const FragA& fa, const FragB& fb) {
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj) {
            const int sb = 4 * (2 * (bi & 1) + (bj & 1));
            f32x16& blk = acc[bi >> 1][bj >> 1];
            f32x4 c = {blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                if constexpr (H2) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, fa.v[0][h]), al = __builtin_bit_cast(f16x8, fa.v[1][h]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, fb.v[bj][0][h]), bl = __builtin_bit_cast(f16x8, fb.v[bj][1][h]);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);  // hi*hi
                } else if constexpr (NS == 3) {  // smallest terms first (same order as conv_igemm.hip)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[1][h], fb.v[bj][1][h], c, 0, 0, 0);  // mid*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[0][h], fb.v[bj][2][h], c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[2][h], fb.v[bj][0][h], c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[0][h], fb.v[bj][1][h], c, 0, 0, 0);  // hi*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[1][h], fb.v[bj][0][h], c, 0, 0, 0);  // mid*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[0][h], fb.v[bj][0][h], c, 0, 0, 0);  // hi*hi
                } else {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[0][h], fb.v[bj][0][h], c, 0, 0, 0);
                }
            }
            blk[sb] = c[0];
            blk[sb + 1] = c[1];
            blk[sb + 2] = c[2];
            blk[sb + 3] = c[3];
        }
    };
    constexpr int NB = 2 * TM;  // 16-row blocks per wave
    auto step = [&](const char* st) {
        FragB fb;
        FragA fa[2];
        read_b(st, fb);
        read_a(st, 0, fa[0]);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            if (bi + 1 < NB) read_a(st, bi + 1, fa[(bi + 1) & 1]);
            mfma_block(bi, fa[bi & 1], fb);
        }
        __syncthreads();
    };
    __syncthreads();  // stage 0 holds chunk 0
    {
        int kc = 0;
        for (; kc + 1 < nchunks; kc += 2) {
            step(stage0);
            step(stage1);
        }
        if (kc < nchunks) step(stage0);
    }
    conv_epilogue<TM, TN, true, 64 * TM, 64 * TN, H2>(p, acc, am, rowinfo, m0, tile_m, split, o_s, o_s2);
}

template <int TM, int TN, int NS, int BK>
int launch2(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = 2 * NS * (BM + BN) * (2 * BK + 16) + BM * 16;
    static std::atomic<unsigned long long> configured[2] = {{0}, {0}};
    const bool aff = a.in_scale != nullptr;
    const void* kern = aff ? reinterpret_cast<const void*>(conv_igemm2_kernel<TM, TN, NS, BK, true>)
                           : reinterpret_cast<const void*>(conv_igemm2_kernel<TM, TN, NS, BK, false>);
    if (hn_ensure_dynamic_lds(kern, lds, configured[aff]) != HN_OK) return HN_ELAUNCH;
    ConvArgs p = a;
    p.ntm = (a.M + BM - 1) / BM;
    p.ntn = (a.K + BN - 1) / BN;
    p.gm = p.ntn >= 8 ? 8 : 1;
    if (p.gm > p.ntm) p.gm = p.ntm;
    int splits = 1;
    if (a.cps > 0) {  // a.cps is expressed in 32-wide K steps; rescale to this instantiation's step
        const int total = a.Kdim / BK;
        p.cps = (a.cps * 32 + BK - 1) / BK;
        splits = (total + p.cps - 1) / p.cps;
    }
    if (aff) HN_LAUNCH((conv_igemm2_kernel<TM, TN, NS, BK, true>), dim3(p.ntm * p.ntn * splits), dim3(512), lds, stream, p);
    else HN_LAUNCH((conv_igemm2_kernel<TM, TN, NS, BK, false>), dim3(p.ntm * p.ntn * splits), dim3(512), lds, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

template <int NS, int BK>
int launch2_tile(const ConvArgs& a, bool tall, bool wide, hipStream_t s) {
    if (tall) return wide ? launch2<2, 2, NS, BK>(a, s) : launch2<2, 1, NS, BK>(a, s);
    return wide ? launch2<1, 2, NS, BK>(a, s) : launch2<1, 1, NS, BK>(a, s);
}

}  // namespace

int launch_conv2(const ConvArgs& a, bool tall, bool wide, int ns, hipStream_t stream) {
    if (ns == MATH2_X3) {
        if (a.C % 32 != 0) return 1;
        return launch2_tile<3, 32>(a, tall, wide, stream);
    }
    if (ns == MATH2_H2) {
        if (a.a_scale == nullptr || a.w_scale == nullptr || a.in_scale != nullptr) return 1;
        if (a.C % 64 == 0 && a.cps == 0) return launch2_tile<2, 64>(a, tall, wide, stream);
        if (a.C % 32 == 0) return launch2_tile<2, 32>(a, tall, wide, stream);
        return 1;
    }
    if (ns == MATH2_BF16) {
        // (K step 128 -- half the barriers -- measured 15-30 % slower: 192 staging registers per producer thread)
        if (a.C % 64 == 0) return launch2_tile<1, 64>(a, tall, wide, stream);
        if (a.C % 32 == 0) return launch2_tile<1, 32>(a, tall, wide, stream);
        return 1;
    }
    return 1;
}

}  // namespace hnconv
