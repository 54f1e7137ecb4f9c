// Implicit-GEMM convolution forward / data-gradient, f16x2 arithmetic, fourth schedule: BOTH operands arrive as fp16
// planes and go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds); nothing is staged through VGPRs, nothing is split
// or written with ds_write inside the kernel.  The K loop is DMA issue + ds_read_b128 + MFMA.
//
// The gathered operand (activations in the forward, dY in the data gradient) is a "plane tensor": two dense fp16 images
// [pixel][ldx channels], hi = fp16(t * s) and lo = fp16(t * s - hi), written by the pass that PRODUCES the tensor (BatchNorm
// apply forward / backward, bilinear resampling, hn_tensor_to_planes) with the power-of-two scale s of the tensor's scale
// slot -- the same representation conv_igemm.hip / conv_igemm3.hip build in registers for every tile that touches an
// element (nine times per column tile for a 3x3 layer), made once.  Given the same scale the LDS contents, the MFMA order
// and therefore the results are bitwise identical to those kernels.
//
// LDS image of one operand stage: [plane][row][32 k] fp16 with UNPADDED 64-byte rows (a 1-KiB DMA covers 16 rows, lane l ->
// row l >> 2, 16-byte piece l & 3); bank conflicts of the fragment reads are avoided by a swizzle instead of padding:
// piece q of row r is stored at position q ^ hsw((r >> 2) & 3), hsw = {0, 3, 2, 1}.  ds_read_b128 is served in four
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS; conv_common.h): with the 16x16x32
// operand map (lane -> row lane & 15, k group lane >> 4) a group holds every row of a 16-row block once, rows {0-3, 12-15}
// at k group g and rows {4-11} at g ^ 1; rows r, r+4, r+8, r+12 share the 64-B quarter of the 256-B bank row, and
// {hsw(0), 1 ^ hsw(1), 1 ^ hsw(2), hsw(3)} = {0, 2, 3, 1} are four different 16-B positions -> conflict-free.  The swizzle
// is applied on the SOURCE side of the DMA (the lane that fills position q fetches piece q ^ hsw).
//
// Schedule (NST stages of A + B, DMA distance NST - 1 steps, ONE barrier per K step):
//   step k:  s_waitcnt vmcnt(all but the newest NST - 2 steps' DMAs)   chunk k has landed (this wave's pieces)
//            s_barrier                                                 ... everybody's; stage (k - 1) % NST is free
//            DMA chunk k + NST - 1 -> stage (k - 1) % NST
//            16 ds_read_b128 + 48 MFMAs per wave (64 x 64 wave tile)
#include <cstdlib>

#include "conv_common.h"
#include "conv_epilogue4.h"

// Register budget of the four-wave kernels, as "workgroups per CU" for __launch_bounds__: 2 = 256 registers per lane (what the LDS
// allows anyway); 3 = 168 -- still two workgroups per CU (LDS), but a third of the register file stays free, so the HBM-bound
// kernels of the other stream (the second encoder, weight gradients) can share a CU with a convolution instead of waiting for
// one of its workgroups to retire.  The K loop needs ~140 registers (no spills there at 168); the epilogue spills 90-150 dwords
// per tile.  Measured inside the step: -0.14 ms (gpurun_out/r4/ab9: 38.04 38.15 38.09 against 38.15 38.26 38.31).
#ifndef HN_C4_OCC
#define HN_C4_OCC 3
#endif

namespace hnconv {
namespace {

template <class T>
constexpr int BM_BLOCKS() { return T::BM / 16; }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset,
                                             0, 0);
#endif
}

__device__ __forceinline__ int hsw(int t) { return (4 - t) & 3; }  // {0, 3, 2, 1}

// WM x WN waves, each TM x TN blocks of 32 x 32 outputs; NST LDS stages; NPL planes per operand: 2 = f16x2 (hi / lo fp16, three
// MFMAs per product), 1 = the mixed-precision bf16 arithmetic (ONE bf16 image per operand -- activations / gradients stored in
// bf16 by the producing pass, plane 0 of the weight planes --, one v_mfma_f32_16x16x32_bf16 per product, no scales)
// LIVE > 0: only the first LIVE 16-row blocks of the tile's WM * TM * 2 exist (BM = 16 * LIVE rows), dealt to the WM row waves in
// turn (block g belongs to wave g % WM) so that every wave keeps the same number of them -- the 224-row tile below.
template <int WM, int WN, int TM, int TN, int NST, int NPL = 2, int LIVE = 0>
struct T4 {
    static constexpr int BM = LIVE > 0 ? LIVE * 16 : WM * TM * 32, BN = WN * TN * 32, NW = WM * WN, NT = 64 * NW, BK = 32;
    static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;  // bytes
    static constexpr int A_BYTES = NPL * A_PLANE, B_BYTES = NPL * B_PLANE;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int QA = (BM / 16 + NW - 1) / NW, QB = BN / 16 / NW;  // 16-row blocks per wave and stage (per plane)
    static constexpr int SINK = NST * STAGE + BM * 16;  // LIVE tiles: 1 KiB per wave for the DMAs of blocks that do not exist
    static constexpr int LDS = SINK + (LIVE > 0 ? NW * 1024 : 0);
    static_assert(LIVE > 0 || BM % (16 * NW) == 0, "row blocks must divide over the waves");
    static_assert(BN % (16 * NW) == 0, "column blocks must divide over the waves");
    static_assert(LIVE == 0 || (LIVE % WM == 0 && LIVE <= WM * TM * 2 && LIVE >= WM * (TM * 2 - 1)), "LIVE: whole rounds of blocks, at most the last one missing");
    static_assert(NPL * (QA + QB) * (NST - 2) < 64, "vmcnt immediate");
};

// accumulator map of a LIVE tile: 16-row block b of the wave (b = 2 * i + (r >> 3)) is block b * WM + wm of the tile
template <int TM, int TN, int WM, int LIVE>
struct AccMapI {
    static constexpr int NC = 2;
    int wm, wn, lane, n0;
    __device__ __forceinline__ static int c2_of(int r) { return (r >> 2) & 1; }
    __device__ __forceinline__ static constexpr bool dead(int i, int r) { return (2 * i + (r >> 3)) * WM >= LIVE; }
    __device__ __forceinline__ int row(int i, int r) const {
        return ((2 * i + (r >> 3)) * WM + wm) * 16 + 4 * (lane >> 4) + (r & 3);
    }
    // row = row_base (the same for the whole wave) + row_lane (the same for all of a lane's rows)
    __device__ __forceinline__ int row_base(int i, int r) const { return ((2 * i + (r >> 3)) * WM + wm) * 16 + (r & 3); }
    __device__ __forceinline__ int row_lane() const { return 4 * (lane >> 4); }
    __device__ __forceinline__ int col(int j, int c2) const { return n0 + (wn * TN + j) * 32 + 16 * c2 + (lane & 15); }
};

template <int WM, int WN, int TM, int TN, int NST, int NPL = 2, int LIVE = 0, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4) ? ((EPI & 1) && TM * TN >= 4 ? 2 : HN_C4_OCC) : 1) void conv_igemm4_kernel(const ConvArgs p) {
    using T = T4<WM, WN, TM, TN, NST, NPL, LIVE>;
    using AM = std::conditional_t<(LIVE > 0), AccMapI<TM, TN, WM, LIVE>, AccMap<TM, TN, true>>;
    constexpr int NBLK = BM_BLOCKS<T>();  // 16-row blocks of the tile
    constexpr int BM = T::BM, BN = T::BN, NW = T::NW, BK = T::BK, QA = T::QA, QB = T::QB;
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    int* rowinfo = reinterpret_cast<int*>(smem4 + NST * T::STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    conv_tile_of(p, logical, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    HN_STAMP(p, 0);
#ifdef HN_STAMPS
    if (p.stamps != nullptr && tid == HN_STAMP_TID) {
        p.stamps[(size_t)blockIdx.x * 16 + 9] = (long long)__builtin_amdgcn_s_memrealtime();
        p.stamps[(size_t)blockIdx.x * 16 + 11] = (long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_ID
        p.stamps[(size_t)blockIdx.x * 16 + 12] = (long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));  // XCC_ID
    }
#endif
    conv_fill_rowinfo<BM>(p, rowinfo, m0, tid);
    __syncthreads();
    HN_STAMP(p, 1);
    // The scale slot's words are REQUESTED here and reduced where they are first needed: behind the first operand DMAs when the
    // accumulators start at zero (vector-memory operations return in order, so the request, issued ahead of the DMAs, costs them
    // nothing and its latency hides behind their issue), in front of the addend loads otherwise.
    unsigned slot_word = 0;
    if constexpr (NPL == 2) slot_word = reinterpret_cast<const unsigned*>(p.a_scale)[(lane & (HN_SCALE_WORDS - 1)) * HN_SCALE_STRIDE];
    float o_s = 1.f, o_s2 = 1.f, i_s = 1.f, i_s2 = 1.f;
    auto reduce_scales = [&] {
        if constexpr (NPL == 2) {
            float a_s, a_inv;
            for (int o = 32; o > 0; o >>= 1) slot_word = max(slot_word, (unsigned)__shfl_xor((int)slot_word, o));
            hn_h2_scales((unsigned)__builtin_amdgcn_readfirstlane((int)slot_word), a_s, a_inv);
            o_s = a_inv; o_s2 = p.w_scale[1];  // applied one after the other (conv_common.h)
            i_s = a_s; i_s2 = p.w_scale[0];
        }
    };
    const bool has_addend = p.addend != nullptr;
    if (has_addend) reduce_scales();

    HN_STAMP(p, 2);
    // ---- A operand: this lane's rows of the wave's 16-row blocks, source piece after the swizzle ------------------
    const int l_row = lane >> 2;
    const int l_src = (lane & 3) ^ hsw((lane >> 4) & 3);
    const int nb0 = __builtin_amdgcn_readfirstlane(rowinfo[2]);
    const long long a_left = p.x_bytes - (long long)nb0 * p.ldx * 2;  // x_bytes: extent of ONE plane from p.xh, in bytes
    const unsigned a_records = a_left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)a_left;
    const unsigned short* a_base = p.xh + (long long)nb0 * p.ldx;
    const __amdgpu_buffer_rsrc_t rsrc_a0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a_base), 0, a_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a_base + p.xh_plane), 0, a_records, 0x00020000);
    int ahb[QA], awb[QA], aoff[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int blk = wave + NW * q;  // (LIVE tiles: the last round of blocks does not exist for every wave)
        const int r = (blk < NBLK ? blk : 0) * 16 + l_row;
        ahb[q] = blk < NBLK ? rowinfo[r * 4 + 0] : -0x40000000;
        awb[q] = rowinfo[r * 4 + 1];
        aoff[q] = (((rowinfo[r * 4 + 2] - nb0) + ahb[q] * p.W + awb[q]) * p.ldx + l_src * 8) * 2;
    }
    // ---- B operand: weight rows (output channels) of the wave's 16-row blocks --------------------------------------
    const unsigned b_records = (unsigned)(p.wx3_plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_b0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3), 0, b_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3 + p.wx3_plane), 0, b_records, 0x00020000);
    unsigned bvoff[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int co = n0 + (wave + NW * q) * 16 + l_row;
        bvoff[q] = co < p.K ? (unsigned)(((long long)co * p.Kdim + l_src * 8) * 2) : b_records;
    }

    const int nchunks = p.Kdim / BK;  // (C % 32 == 0, no split-K on this path)
    int f_kh = 0, f_kw = 0, f_c0 = 0, f_k0 = 0, f_chunk = 0;  // position of the next chunk to fetch
    auto dma_chunk = [&](char* stage) {  // chunk f_chunk -> stage (out of range past the end: zero fill); advances
        const bool live = f_chunk < nchunks;
        const int dh = f_kh * p.dil, dw = f_kw * p.dil;
        const int doff = ((dh * p.W + dw) * p.ldx + f_c0) * 2;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const bool ok = live & ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
            const unsigned vo = ok ? (unsigned)(aoff[q] + doff) : a_records;
            // (a block past the tile's last -- LIVE tiles -- fetches nothing: an out-of-range DMA still WRITES its zeros, so it
            // goes to a sink of its own; issued all the same, the counted waits assume the same number of DMAs from every wave)
            char* dst = (wave + NW * q < NBLK) ? stage + (wave + NW * q) * 1024 : smem4 + T::SINK + wave * 1024;
            dma16(rsrc_a0, dst, vo, 0);
            if constexpr (NPL == 2) dma16(rsrc_a1, dst + T::A_PLANE, vo, 0);
        }
        // (the scalar offset operand: said to be uniform in so many words, or the tap-inner arithmetic below lands in vector
        // registers and every weight DMA is wrapped in a readfirstlane loop)
        const unsigned ko = (unsigned)__builtin_amdgcn_readfirstlane(f_k0 * 2);
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const unsigned vo = live ? bvoff[q] : b_records;
            char* dst = stage + T::A_BYTES + (wave + NW * q) * 1024;
            dma16(rsrc_b0, dst, vo, ko);
            if constexpr (NPL == 2) dma16(rsrc_b1, dst + T::B_PLANE, vo, ko);
        }
        if (p.tap_inner) {
            // all taps of a 32-channel chunk, then the next chunk: the KH x KW shifted reads of the same rows follow one another
            // while those rows' 64-byte pieces are still in the L2 (channels-outermost walks the whole row footprint, C / 32
            // chunks, between two taps: every tap came from memory again -- 4.9 x the algorithmic bytes on the 3x3 layers)
            ++f_kw;
            const int wrap_w = f_kw == p.KW ? 1 : 0;
            f_kw = wrap_w ? 0 : f_kw;
            f_kh += wrap_w;
            const int wrap_h = f_kh == p.KH ? 1 : 0;
            f_kh = wrap_h ? 0 : f_kh;
            f_c0 += wrap_h ? BK : 0;
            f_k0 = (f_kh * p.KW + f_kw) * p.C + f_c0;
        } else {
            f_c0 += BK;
            const int wrap_c = f_c0 >= p.C ? 1 : 0;
            f_c0 = wrap_c ? 0 : f_c0;
            f_kw += wrap_c;
            const int wrap_w = f_kw == p.KW ? 1 : 0;
            f_kw = wrap_w ? 0 : f_kw;
            f_kh += wrap_w;
            f_k0 += BK;
        }
        ++f_chunk;
    };

    const AM am{wm, wn, lane, n0};
    f32x16 acc[TM][TN];
    conv_init_acc4<TM, TN, BM, BN, AM>(p, acc, am, rowinfo, m0, i_s, i_s2);

    const int frag = (lane & 15) * 64 + (((lane >> 4) ^ hsw((lane >> 2) & 3)) * 16);
    const int a_frag = (LIVE > 0 ? wm * 16 * 64 : wm * TM * 32 * 64) + frag;  // (LIVE: the wave's blocks are WM blocks apart)
    constexpr int A_STEP = (LIVE > 0 ? WM : 1) * 16 * 64, NBI = LIVE > 0 ? LIVE / WM : 2 * TM;
    const int b_frag = T::A_BYTES + wn * TN * 32 * 64 + frag;
    auto mfma_phase = [&](const char* st) {
        using frag_t = std::conditional_t<NPL == 2, f16x8, bf16x8>;
        frag_t fb[2 * TN][NPL];
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s2 = 0; s2 < NPL; ++s2)
                fb[bj][s2] = *reinterpret_cast<const frag_t*>(st + b_frag + s2 * T::B_PLANE + bj * 16 * 64);
#pragma unroll
        for (int bi = 0; bi < NBI; ++bi) {
            frag_t fa[NPL];
#pragma unroll
            for (int s2 = 0; s2 < NPL; ++s2) fa[s2] = *reinterpret_cast<const frag_t*>(st + a_frag + s2 * T::A_PLANE + bi * A_STEP);
            // per accumulator the order is lo*hi, hi*lo, hi*hi (as in conv_igemm.hip); consecutive MFMAs go to different
            // accumulators
            f32x4 c[2 * TN];
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                const f32x16& blk = acc[bi >> 1][bj >> 1];
                c[bj] = f32x4{blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
            }
            if constexpr (NPL == 2) {
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj) c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1], fb[bj][0], c[bj], 0, 0, 0);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj) c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][1], c[bj], 0, 0, 0);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj) c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0], fb[bj][0], c[bj], 0, 0, 0);
            } else {
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj) c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][0], c[bj], 0, 0, 0);
            }
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                f32x16& blk = acc[bi >> 1][bj >> 1];
                blk[sb] = c[bj][0];
                blk[sb + 1] = c[bj][1];
                blk[sb + 2] = c[bj][2];
                blk[sb + 3] = c[bj][3];
            }
        }
    };

    constexpr int PER_STEP = NPL * (QA + QB);  // DMA instructions a wave issues per chunk
    auto wait_landed = [] {  // everything but the newest NST - 2 chunks' DMAs of this wave
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP * (NST - 2)) : "memory");
    };
    auto barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // the addend loads of conv_init_acc are vector-memory operations too: drain them before the counted waits start (without an
    // addend only the scale slot's request is in flight: it is older than every DMA, so the counted waits stay right)
    if (has_addend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();  // rowinfo reads of every wave are done (init) before a DMA can reuse ... (rowinfo is not reused; keeps waves aligned)
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) dma_chunk(smem4 + s * T::STAGE);
    if (!has_addend) reduce_scales();
    int st_cur = 0, st_free = NST - 1;  // stage of chunk kc; stage that becomes free at the barrier of step kc
    HN_STAMP(p, 3);
    for (int kc = 0; kc < nchunks; ++kc) {
#ifdef HN_STAMPS
        if (kc == 3) HN_STAMP(p, 13);  // inside one step (the fourth) of thread 0's wave: 13 -> 14 the counted wait for its own DMAs,
#endif                                 // 14 -> 15 the barrier (= the other waves' DMAs)
        wait_landed();
#ifdef HN_STAMPS
        if (kc == 3) HN_STAMP(p, 14);
#endif
        barrier();
#ifdef HN_STAMPS
        if (kc == 0) HN_STAMP(p, 4);
        if (kc == 3) HN_STAMP(p, 15);
#endif
        dma_chunk(smem4 + st_free * T::STAGE);
        mfma_phase(smem4 + st_cur * T::STAGE);
        st_free = st_cur;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    }
    HN_STAMP(p, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last, out-of-range DMAs)
    HN_STAMP(p, 6);

    conv_epilogue4<TM, TN, BM, BN, NPL == 2, NPL == 1, (EPI & 1) != 0, AM>(p, acc, am, rowinfo, m0, tile_m, o_s, o_s2);
#ifdef HN_STAMPS
    HN_STAMP(p, 7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HN_STAMP(p, 8);
    if (p.stamps != nullptr && tid == HN_STAMP_TID) p.stamps[(size_t)blockIdx.x * 16 + 10] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int WM, int WN, int TM, int TN, int NST, int NPL, int LIVE, int EPI>
int launch4e(const ConvArgs& a, hipStream_t stream) {
    using T = T4<WM, WN, TM, TN, NST, NPL, LIVE>;
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm4_kernel<WM, WN, TM, TN, NST, NPL, LIVE, EPI>), T::LDS, configured) != HN_OK)
        return HN_ELAUNCH;
    ConvArgs p = a;
#ifdef HN_STAMPS
    p.stamps = g_hn_stamps;
#endif
    p.ntm = (a.M + T::BM - 1) / T::BM;
    p.ntn = (a.K + T::BN - 1) / T::BN;
    static const int gm_env = getenv("HN_GROUP_M") ? atoi(getenv("HN_GROUP_M")) : 0;  // experiment knob
    static const int tap_env = getenv("HN_TAP_INNER") ? atoi(getenv("HN_TAP_INNER")) : 1;  // experiment knob (0: channels innermost)
    p.tap_inner = (tap_env && a.KH * a.KW > 1) ? 1 : 0;
    p.gm = gm_env > 0 ? gm_env : (p.ntn >= (T::BN >= 256 ? 4 : 8) ? 8 : 1);
    if (p.gm > p.ntm) p.gm = p.ntm;
    HN_LAUNCH((conv_igemm4_kernel<WM, WN, TM, TN, NST, NPL, LIVE, EPI>), dim3(p.ntm * p.ntn), dim3(T::NT), T::LDS, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// EPI (the kernel's last template argument): 1 = the launch carries the BatchNorm-backward reduction (ConvArgs::br_y) -- an
// instantiation of its own, so that the launches without it do not pay for its registers (conv_epilogue4.h)
template <int WM, int WN, int TM, int TN, int NST, int NPL = 2, int LIVE = 0>
int launch4(const ConvArgs& a, hipStream_t stream) {
    return a.br_y != nullptr ? launch4e<WM, WN, TM, TN, NST, NPL, LIVE, 1>(a, stream) : launch4e<WM, WN, TM, TN, NST, NPL, LIVE, 0>(a, stream);
}

}  // namespace

// (a_scale == NULL: the one-plane bf16 arithmetic)
bool conv4_covers(const ConvArgs& a) {
    return a.xh != nullptr && a.wx3 != nullptr && (a.a_scale == nullptr) == (a.w_scale == nullptr) && a.C % 32 == 0 && a.cps == 0 &&
           a.in_scale == nullptr && a.in_dil == 1 && a.ldx % 8 == 0 && hn_aligned16(a.xh) && a.xh_plane % 8 == 0 &&
           a.wx3_plane * 2 < (1ll << 32) && (long long)a.H * a.W >= 128 && 3ll * a.H * a.W * a.ldx * 2 < (1ll << 31);
}

// tile: 0 = 128 x 128 (four waves, two workgroups per CU), 1 = 128 x 64 (four waves 4 x 1), 2 = 128 x 256 (eight waves, three
// stages, one workgroup per CU), 3 = 256 x 128 (eight waves 4 x 2, three stages); 7 = 224 x 256 (eight waves 2 x 4, each row wave
// SEVEN 16-row blocks of the fourteen, dealt in turn -- T4<.., LIVE = 14> --, two stages of 60 KB).  Two reasons: fewer operand
// bytes per multiply-add than the 128-row tiles (their K loop is bound by what a CU can pull out of its L2,
// tools/probes/dma_line_probe.hip, tools/stamp_conv.py), and the HEIGHT: 8 x 40 x 88 = 28160 = 125.7 x 224 rows and 8 x 80 x 176
// = 502.9 x 224, so the layers on those maps make 126 / 503 row tiles -- with 2, 4 or 8 column tiles a whole number of rounds
// on 256 CUs (0.98, 1.97, 3.94) where 128-row tiles make 1.72 / 3.44 / 6.9; 8 = 112 x 128, its two-workgroups-per-CU sibling (four
// waves side by side, each all seven row blocks of its 32 columns: 252 / 1006 row tiles -- 0.98 .. 3.94 rounds on the 512 slots
// for the layers with 256 .. 2048 output channels that make 0.86 / 1.72 / 3.44 rounds with 128 rows);
// 4 = 128 x 64, 5 = 64 x 128, 6 = 64 x 64 with
// THREE (6: four) stages and two workgroups per CU: the DMA runs two (three) K steps ahead -- for the layers whose K steps
// carry little matrix work per byte (1x1 convolutions with few channels on one side), where a distance of one step leaves
// every step waiting out the memory latency
int launch_conv4(const ConvArgs& a, int tile, hipStream_t stream) {
    if (a.a_scale == nullptr) {  // bf16, one plane: the same tiles with deeper pipelines (a stage is half the size)
        switch (tile) {
            case 1: return launch4<4, 1, 1, 2, 3, 1>(a, stream);
            case 2: return launch4<2, 4, 2, 2, 4, 1>(a, stream);
            case 3: return launch4<4, 2, 2, 2, 4, 1>(a, stream);
            case 4: return launch4<4, 1, 1, 2, 4, 1>(a, stream);
            case 5: return launch4<2, 2, 1, 2, 4, 1>(a, stream);
            case 6: return launch4<2, 2, 1, 1, 4, 1>(a, stream);
            case 7: return launch4<2, 4, 4, 2, 3, 1, 14>(a, stream);
            case 8: return launch4<1, 4, 4, 1, 4, 1, 7>(a, stream);
            default: return launch4<2, 2, 2, 2, 3, 1>(a, stream);
        }
    }
    switch (tile) {
        case 1: return launch4<4, 1, 1, 2, 2>(a, stream);
        case 2: return launch4<2, 4, 2, 2, 3>(a, stream);
        case 3: return launch4<4, 2, 2, 2, 3>(a, stream);
        case 4: return launch4<4, 1, 1, 2, 3>(a, stream);
        case 5: return launch4<2, 2, 1, 2, 3>(a, stream);
        case 6: return launch4<2, 2, 1, 1, 4>(a, stream);
        case 7: return launch4<2, 4, 4, 2, 2, 2, 14>(a, stream);
        case 8: return launch4<1, 4, 4, 1, 2, 2, 7>(a, stream);
        default: return launch4<2, 2, 2, 2, 2>(a, stream);
    }
}

// (without the last template argument and the closing bracket: plan_name appends ", EPI>")
const char* conv4_kernel_name(int tile, bool bf16) {
    if (bf16) {
        switch (tile) {
            case 1: return "conv_igemm4_kernel<4, 1, 1, 2, 3, 1, 0";
            case 2: return "conv_igemm4_kernel<2, 4, 2, 2, 4, 1, 0";
            case 3: return "conv_igemm4_kernel<4, 2, 2, 2, 4, 1, 0";
            case 4: return "conv_igemm4_kernel<4, 1, 1, 2, 4, 1, 0";
            case 5: return "conv_igemm4_kernel<2, 2, 1, 2, 4, 1, 0";
            case 6: return "conv_igemm4_kernel<2, 2, 1, 1, 4, 1, 0";
            case 7: return "conv_igemm4_kernel<2, 4, 4, 2, 3, 1, 14";
            case 8: return "conv_igemm4_kernel<1, 4, 4, 1, 4, 1, 7";
            default: return "conv_igemm4_kernel<2, 2, 2, 2, 3, 1, 0";
        }
    }
    switch (tile) {
        case 1: return "conv_igemm4_kernel<4, 1, 1, 2, 2, 2, 0";
        case 2: return "conv_igemm4_kernel<2, 4, 2, 2, 3, 2, 0";
        case 3: return "conv_igemm4_kernel<4, 2, 2, 2, 3, 2, 0";
        case 4: return "conv_igemm4_kernel<4, 1, 1, 2, 3, 2, 0";
        case 5: return "conv_igemm4_kernel<2, 2, 1, 2, 3, 2, 0";
        case 6: return "conv_igemm4_kernel<2, 2, 1, 1, 4, 2, 0";
        case 7: return "conv_igemm4_kernel<2, 4, 4, 2, 2, 2, 14";
        case 8: return "conv_igemm4_kernel<1, 4, 4, 1, 2, 2, 7";
        default: return "conv_igemm4_kernel<2, 2, 2, 2, 2, 2, 0";
    }
}

}  // namespace hnconv
