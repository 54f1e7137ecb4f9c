// Implicit-GEMM convolution forward / data-gradient, bf16x3 arithmetic, third schedule: the pre-split WEIGHT operand goes
// global -> LDS by LDS-DMA (buffer_load ... lds), the activation operand through registers.
//
// Why: with three bf16 planes per operand a K step of the first schedule writes 61 KB of LDS per 128 x 128 tile through
// ds_write (VGPR -> LDS transfer, ~75 B/clk/CU): 800-1 200 of a ~3 000-cycle step whichever way the waves are arranged
// (profiles/r02_conv_schedules.txt).  Half of those bytes are the weight planes, which are already
// in their final form in memory.  Here they never touch a VGPR:
//
//   bf16x3: workgroup = 8 waves (2 x 4), block tile 128 pixels x 256 channels, ONE workgroup per CU (155 KB of LDS):
//     A stage   [3][128][80 B]        30 KB   activations: buffer loads -> registers -> exact 3-way split -> ds_write
//     B stage 0 [3][256][80 B]        61 KB   weights: 60 x 1-KiB LDS-DMA pieces per K step (lane-linear: the 80-byte rows
//     B stage 1                        61 KB   are 5 slots of 16 B, the 5th -- the bank padding -- fetched out of range)
//   step k:  vmcnt(0)                  everything issued during step k-1 had a whole MFMA phase to land
//            split chunk k (A), ds_write, lgkmcnt(0), s_barrier
//            issue: A loads of chunk k+1 (ONE staging set), B DMA of chunk k+1 into the other B stage
//            24 ds_read_b128 + 96 MFMAs per wave (two waves per SIMD: 3 072 matrix-pipe cycles per step)
//            lgkmcnt(0), s_barrier
// Raw s_barrier with explicit waits: a __syncthreads() would drain the LDS-DMA just issued (cdna_hip_programming.md,
// "Pipelining across barriers").  Same LDS images, fragment reads, MFMA order and epilogue as conv_igemm.hip: results are
// bitwise identical to the other schedules.
// f16x2 (two fp16 planes): the same kernel with 2 planes -- either the eight-wave 128 x 256 tile (104 KB), or FOUR waves on
// 128 x 128 with 62 KB of LDS, i.e. TWO workgroups per CU as in the first schedule, and two A staging sets (fetch distance
// two steps: the waits count "all but the newest RA loads").  What it removes from the first schedule: the weight planes'
// trip through VGPRs and ds_write_b128 (13 of the ~25 VGPR->LDS transfer cycles a wave spends per K step).
#include <cstdlib>

#include "conv_common.h"

namespace hnconv {
namespace {

__device__ __forceinline__ void dma16_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset,
                                             0, 0);
#endif
}

// NS planes per operand (3: bf16x3, 2: f16x2), WN waves along the channel axis (4: eight waves, 128 x 256 tile, one workgroup
// per CU; 2: four waves, 128 x 128 tile, two workgroups per CU), SETS staging register sets of the A operand (fetch distance
// in K steps).
template <int NS, int WN>
struct T3 {
    static constexpr int BM = 128, BN = 64 * WN, NT = 128 * WN, BK = 32;
    static constexpr int A_PLANE = BM * X3_ROW, B_PLANE = BN * X3_ROW;
    static constexpr int A_BYTES = NS * A_PLANE, B_BYTES = NS * B_PLANE;
    static constexpr int B_CHUNKS = B_BYTES / 1024;  // 1-KiB LDS-DMA pieces per K step
    static constexpr int LDS = A_BYTES + 2 * B_BYTES + BM * 16;
};

template <int NS, int WN, int SETS>
__global__ __launch_bounds__(128 * WN, 2) void conv_igemm3_kernel(const ConvArgs p) {
    using T = T3<NS, WN>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = 2, TN = 2, BK = T::BK;
    constexpr int NW = 2 * WN;  // waves
    constexpr bool H2 = NS == 2;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    char* As = smem3;
    char* Bs0 = smem3 + T::A_BYTES;
    int* rowinfo = reinterpret_cast<int*>(smem3 + T::A_BYTES + 2 * T::B_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    conv_tile_of(p, logical, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    conv_fill_rowinfo<BM>(p, rowinfo, m0, tid);
    __syncthreads();
    float a_s = 1.f, o_s = 1.f, i_s = 1.f, o_s2 = 1.f, i_s2 = 1.f;  // f16x2 operand scales (conv_igemm.hip)
    if constexpr (H2) {
        float a_inv;
        hn_h2_scales_of_slot(p.a_scale, a_s, a_inv);
        o_s = a_inv;
        o_s2 = p.w_scale[1];
        i_s = a_s;
        i_s2 = p.w_scale[0];
    }

    // ---- A operand: 128 rows x 8 pieces of 16 B per K step ------------------------------------------------------
    constexpr int RA = BM * 8 / NT;   // pieces per thread: 2 (eight waves) or 4
    constexpr int AROWS = NT / 8;     // rows per pass
    const int a_c = tid & 7, a_r = tid >> 3;
    int ahb[RA], awb[RA], aoff32[RA];
    const int nb0 = __builtin_amdgcn_readfirstlane(rowinfo[2]);
    const long long a_left = p.x_bytes - (long long)nb0 * p.ldx * 4;
    const unsigned a_records = a_left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)a_left;
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long long)nb0 * p.ldx, 0, a_records, 0x00020000);
#pragma unroll
    for (int q = 0; q < RA; ++q) {
        const int r = a_r + AROWS * q;
        ahb[q] = rowinfo[r * 4 + 0];
        awb[q] = rowinfo[r * 4 + 1];
        aoff32[q] = (((rowinfo[r * 4 + 2] - nb0) + ahb[q] * p.W + awb[q]) * p.ldx + a_c * 4) * 4;
    }
    // ---- B operand: per wave, the source offsets of its LDS-DMA pieces (chunk c = wave + NW j, slot = 64 c + lane) ---
    constexpr int NBC = (T::B_CHUNKS + NW - 1) / NW;
    const unsigned b_records = (unsigned)(NS * p.wx3_plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3), 0, b_records, 0x00020000);
    unsigned bvoff[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) {
        const int slot = 64 * (wave + NW * j) + lane;
        const int plane = slot / (BN * 5), rem = slot % (BN * 5);
        const int row = rem / 5, piece = rem % 5;
        const int co = n0 + row;
        const bool ok = piece < 4 && co < p.K && wave + NW * j < T::B_CHUNKS;
        bvoff[j] = ok ? (unsigned)((plane * p.wx3_plane + (long long)co * p.Kdim + ((piece ^ x3_kflip(row)) * 8)) * 2) : b_records;
    }

    const int nchunks = p.Kdim / BK;  // (no split-K on this tile; C % 32 == 0)
    // A and B are fetched at different distances (A: SETS steps ahead, B: one), so each has its own position
    struct Pos { int kh, kw, c0, k0, chunk; };
    Pos pa{0, 0, 0, 0, 0}, pb{0, 0, 0, 0, 0};
    auto advance = [&](Pos& s) {
        s.c0 += BK;
        const int wrap_c = s.c0 >= p.C ? 1 : 0;
        s.c0 = wrap_c ? 0 : s.c0;
        s.kw += wrap_c;
        const int wrap_w = s.kw == p.KW ? 1 : 0;
        s.kw = wrap_w ? 0 : s.kw;
        s.kh += wrap_w;
        s.k0 += BK;
        ++s.chunk;
    };
    f32x4 va[SETS][RA];
    auto fetch_a = [&](f32x4 (&dst)[RA]) {  // chunk pa.chunk -> dst (zeros past the end); advances
        const bool live = pa.chunk < nchunks;
        const int dh = pa.kh * p.dil, dw = pa.kw * p.dil;
        const int doff = ((dh * p.W + dw) * p.ldx + pa.c0) * 4;
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const bool ok = live & ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
            dst[q] = buffer_load16(rsrc_a, ok ? (unsigned)(aoff32[q] + doff) : a_records, 0);
        }
        advance(pa);
    };
    auto dma_b = [&](char* bs) {  // chunk pb.chunk -> bs (out of range past the end: zero fill); advances
        const bool live = pb.chunk < nchunks;
#pragma unroll
        for (int j = 0; j < NBC; ++j) {
            if (j < NBC - 1 || wave < T::B_CHUNKS - NW * (NBC - 1))  // wave-uniform: a ragged last round
                dma16_lds(rsrc_b, bs + 1024 * (wave + NW * j), live ? bvoff[j] : b_records, (unsigned)(pb.k0 * 2));
        }
        advance(pb);
    };
    auto store_a = [&](const f32x4 (&src)[RA]) {
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const int row = a_r + AROWS * q;
            char* dst = As + row * X3_ROW + (((a_c >> 1) ^ x3_kflip(row)) * 16) + (a_c & 1) * 8;
            if constexpr (H2) {
                f16x4 h, l;
                split2h(src[q], a_s, h, l);
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + T::A_PLANE) = l;
            } else if constexpr (NS == 1) {
                *reinterpret_cast<bf16x4*>(dst) = round_bf16(src[q]);
            } else {
                bf16x4 h, m, l;
                split3(src[q], h, m, l);
                *reinterpret_cast<bf16x4*>(dst) = h;
                *reinterpret_cast<bf16x4*>(dst + T::A_PLANE) = m;
                *reinterpret_cast<bf16x4*>(dst + 2 * T::A_PLANE) = l;
            }
        }
    };

    const AccMap<TM, TN, true> am{wm, wn, lane, n0};
    f32x16 acc[TM][TN];
    conv_init_acc<TM, TN, true>(p, acc, am, rowinfo, m0, i_s, i_s2);

    const int frag = (lane & 15) * X3_ROW + (((lane >> 4) ^ x3_kflip(lane & 15)) * 16);
    const char* a_frag = As + wm * TM * 32 * X3_ROW + frag;
    const int b_frag = wn * TN * 32 * X3_ROW + frag;
    auto mfma_phase = [&](const char* bs) {
        using frag_t = std::conditional_t<H2, f16x8, bf16x8>;
        frag_t fb[2 * TN][NS];
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3)
                fb[bj][s3] = *reinterpret_cast<const frag_t*>(bs + b_frag + (s3 * BN + bj * 16) * X3_ROW);
#pragma unroll
        for (int bi = 0; bi < 2 * TM; ++bi) {
            frag_t fa[NS];
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3) fa[s3] = *reinterpret_cast<const frag_t*>(a_frag + (s3 * BM + bi * 16) * X3_ROW);
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                f32x16& blk = acc[bi >> 1][bj >> 1];
                f32x4 c = {blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
                if constexpr (H2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1], fb[bj][0], c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][1], c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][0], c, 0, 0, 0);  // hi*hi
                } else if constexpr (NS == 1) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][0], c, 0, 0, 0);
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
    // explicit waits + raw barrier (see header); the asm statements also keep hipcc from moving memory operations across.
    // Issue order inside a step is B DMA first, A loads second, so "all but the newest RA" = the B stage of this step and
    // the A set of this step have landed while the A loads issued one step ago (SETS == 2) stay in flight.
    auto wait_step = [] {
        if constexpr (SETS == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto step = [&](int kc, f32x4 (&set)[RA]) {
        char* bs_cur = Bs0 + (kc & 1) * T::B_BYTES;
        char* bs_nxt = Bs0 + ((kc + 1) & 1) * T::B_BYTES;
        wait_step();        // chunk kc: A in registers, B in its stage (this wave's pieces; the barrier covers the others')
        store_a(set);
        barrier();          // #1: A stage and B stage kc & 1 complete for every wave
        dma_b(bs_nxt);      // chunk kc + 1
        fetch_a(set);       // chunk kc + SETS (zeros past the end)
        mfma_phase(bs_cur);
        barrier();          // #2: every wave has read the A stage and B stage kc & 1
    };

    // prologue: B chunk 0, A chunks 0 .. SETS-1 in flight (the first wait_step() must see them in the steady-state order)
    if constexpr (SETS == 2) {
        fetch_a(va[0]);
        dma_b(Bs0);
        fetch_a(va[1]);
        int kc = 0;
        for (; kc + 1 < nchunks; kc += 2) {
            step(kc, va[0]);
            step(kc + 1, va[1]);
        }
        if (kc < nchunks) step(kc, va[0]);
    } else {
        dma_b(Bs0);
        fetch_a(va[0]);
        for (int kc = 0; kc < nchunks; ++kc) step(kc, va[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last, out-of-range fetches)

    conv_epilogue<TM, TN, true, BM, BN, H2>(p, acc, am, rowinfo, m0, tile_m, 0, o_s, o_s2);
}

template <int NS, int WN, int SETS>
int launch3(const ConvArgs& a, hipStream_t stream) {
    using T = T3<NS, WN>;
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm3_kernel<NS, WN, SETS>), T::LDS, configured) != HN_OK)
        return HN_ELAUNCH;
    ConvArgs p = a;
    p.ntm = (a.M + T::BM - 1) / T::BM;
    p.ntn = (a.K + T::BN - 1) / T::BN;
    static const int gm_env = getenv("HN_GROUP_M") ? atoi(getenv("HN_GROUP_M")) : 0;  // experiment knob
    p.gm = gm_env > 0 ? gm_env : (p.ntn >= (WN == 4 ? 4 : 8) ? 8 : 1);
    if (p.gm > p.ntm) p.gm = p.ntm;
    HN_LAUNCH((conv_igemm3_kernel<NS, WN, SETS>), dim3(p.ntm * p.ntn), dim3(T::NT), T::LDS, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

}  // namespace

// planes: 3 = bf16x3 (the 128 x 256 tile), 2 = f16x2, 1 = bf16 (wide: 128 x 256, eight waves; else 128 x 128, four waves, two
// workgroups per CU)
bool conv3_covers(const ConvArgs& a, int planes, bool wide) {
    if (!(a.C % 32 == 0 && a.cps == 0 && a.in_scale == nullptr && a.wx3 != nullptr)) return false;
    if (planes == 2 && (a.a_scale == nullptr || a.w_scale == nullptr)) return false;
    return a.K >= (planes == 3 || wide ? 192 : 96);
}

int launch_conv3(const ConvArgs& a, int planes, bool wide, hipStream_t stream) {
    if (planes == 3) return launch3<3, 4, 1>(a, stream);
    if (planes == 1) return wide ? launch3<1, 4, 1>(a, stream) : launch3<1, 2, 2>(a, stream);
    return wide ? launch3<2, 4, 1>(a, stream) : launch3<2, 2, 2>(a, stream);
}

}  // namespace hnconv
