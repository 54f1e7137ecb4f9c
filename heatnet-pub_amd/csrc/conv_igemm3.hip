// Implicit-GEMM convolution forward / data-gradient, bf16x3 arithmetic, third schedule: the pre-split WEIGHT operand goes
// global -> LDS by LDS-DMA (buffer_load ... lds), the activation operand through registers.
//
// Why: with three bf16 planes per operand a K step of the first schedule writes 61 KB of LDS per 128 x 128 tile through
// ds_write (VGPR -> LDS transfer, ~75 B/clk/CU): 800-1 200 of a ~3 000-cycle step whichever way the waves are arranged
// (profiles/r02_conv_schedules.txt).  Half of those bytes are the weight planes, which are already
// in their final form in memory.  Here they never touch a VGPR:
//
//   workgroup = 8 waves (2 x 4), block tile 128 pixels x 256 channels, ONE workgroup per CU (155 KB of LDS):
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
#include "conv_common.h"

namespace hnconv {
namespace {

__device__ __forceinline__ void dma16_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset,
                                             0, 0);
#endif
}

constexpr int T3_BM = 128, T3_BN = 256, T3_NT = 512, T3_BK = 32;
constexpr int T3_A_PLANE = T3_BM * X3_ROW, T3_B_PLANE = T3_BN * X3_ROW;
constexpr int T3_A_BYTES = 3 * T3_A_PLANE, T3_B_BYTES = 3 * T3_B_PLANE;  // 30 720 and 61 440
constexpr int T3_B_CHUNKS = T3_B_BYTES / 1024;                           // 60 LDS-DMA pieces per K step
constexpr int T3_LDS = T3_A_BYTES + 2 * T3_B_BYTES + T3_BM * 16;

__global__ __launch_bounds__(T3_NT, 2) void conv_igemm3_kernel(const ConvArgs p) {
    constexpr int BM = T3_BM, BN = T3_BN, TM = 2, TN = 2, BK = T3_BK;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    char* As = smem3;
    char* Bs0 = smem3 + T3_A_BYTES;
    int* rowinfo = reinterpret_cast<int*>(smem3 + T3_A_BYTES + 2 * T3_B_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    conv_tile_of(p, logical, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    conv_fill_rowinfo<BM>(p, rowinfo, m0, tid);
    __syncthreads();

    // ---- A operand: 128 rows x 8 pieces of 16 B per K step = 2 pieces per thread --------------------------------
    constexpr int RA = BM * 8 / T3_NT;  // 2
    constexpr int AROWS = T3_NT / 8;    // 64 rows per pass
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
    // ---- B operand: per wave, the source offsets of its LDS-DMA pieces (chunk c = wave + 8 j, slot = 64 c + lane) ---
    constexpr int NBC = (T3_B_CHUNKS + 7) / 8;  // 8 pieces for waves 0-3, 7 for waves 4-7
    const unsigned b_records = (unsigned)(3 * p.wx3_plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wx3), 0, b_records, 0x00020000);
    unsigned bvoff[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) {
        const int slot = 64 * (wave + 8 * j) + lane;
        const int plane = slot / (BN * 5), rem = slot % (BN * 5);
        const int row = rem / 5, piece = rem % 5;
        const int co = n0 + row;
        const bool ok = piece < 4 && co < p.K && wave + 8 * j < T3_B_CHUNKS;
        bvoff[j] = ok ? (unsigned)((plane * p.wx3_plane + (long long)co * p.Kdim + ((piece ^ x3_kflip(row)) * 8)) * 2) : b_records;
    }

    const int nchunks = p.Kdim / BK;  // (no split-K on this tile; C % 32 == 0)
    int st_kh = 0, st_kw = 0, st_c0 = 0, st_k0 = 0, st_chunk = 0;
    f32x4 va[RA];
    auto fetch_a = [&]() {  // chunk st_chunk -> va (zeros past the end); advances the position
        const bool live = st_chunk < nchunks;
        const int dh = st_kh * p.dil, dw = st_kw * p.dil;
        const int doff = ((dh * p.W + dw) * p.ldx + st_c0) * 4;
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const bool ok = live & ((unsigned)(ahb[q] + dh) < (unsigned)p.H) & ((unsigned)(awb[q] + dw) < (unsigned)p.W);
            va[q] = buffer_load16(rsrc_a, ok ? (unsigned)(aoff32[q] + doff) : a_records, 0);
        }
    };
    auto dma_b = [&](char* bs) {  // chunk st_chunk -> bs (out of range past the end: zero fill)
        const bool live = st_chunk < nchunks;
#pragma unroll
        for (int j = 0; j < NBC; ++j) {
            if (j < NBC - 1 || wave < T3_B_CHUNKS - 8 * (NBC - 1))  // wave-uniform: the last round has 4 pieces
                dma16_lds(rsrc_b, bs + 1024 * (wave + 8 * j), live ? bvoff[j] : b_records, (unsigned)(st_k0 * 2));
        }
    };
    auto advance = [&]() {
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
    auto store_a = [&]() {
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            bf16x4 h, m, l;
            split3(va[q], h, m, l);
            const int row = a_r + AROWS * q;
            char* dst = As + row * X3_ROW + (((a_c >> 1) ^ x3_kflip(row)) * 16) + (a_c & 1) * 8;
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + T3_A_PLANE) = m;
            *reinterpret_cast<bf16x4*>(dst + 2 * T3_A_PLANE) = l;
        }
    };

    const AccMap<TM, TN, true> am{wm, wn, lane, n0};
    f32x16 acc[TM][TN];
    conv_init_acc<TM, TN, true>(p, acc, am, rowinfo, m0);

    const int frag = (lane & 15) * X3_ROW + (((lane >> 4) ^ x3_kflip(lane & 15)) * 16);
    const char* a_frag = As + wm * TM * 32 * X3_ROW + frag;
    const int b_frag = wn * TN * 32 * X3_ROW + frag;
    auto mfma_phase = [&](const char* bs) {
        bf16x8 fb[2 * TN][3];
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
                fb[bj][s3] = *reinterpret_cast<const bf16x8*>(bs + b_frag + (s3 * BN + bj * 16) * X3_ROW);
#pragma unroll
        for (int bi = 0; bi < 2 * TM; ++bi) {
            bf16x8 fa[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) fa[s3] = *reinterpret_cast<const bf16x8*>(a_frag + (s3 * BM + bi * 16) * X3_ROW);
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                f32x16& blk = acc[bi >> 1][bj >> 1];
                f32x4 c = {blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][1], c, 0, 0, 0);  // mid*mid
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][2], c, 0, 0, 0);  // hi*lo
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[bj][0], c, 0, 0, 0);  // lo*hi
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][1], c, 0, 0, 0);  // hi*mid
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][0], c, 0, 0, 0);  // mid*hi
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][0], c, 0, 0, 0);  // hi*hi
                blk[sb] = c[0];
                blk[sb + 1] = c[1];
                blk[sb + 2] = c[2];
                blk[sb + 3] = c[3];
            }
        }
    };
    // explicit waits + raw barrier (see header); the asm statements also keep hipcc from moving memory operations across
    auto wait_vm = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // prologue: chunk 0 in flight (A registers, B stage 0)
    fetch_a();
    dma_b(Bs0);
    advance();
    for (int kc = 0; kc < nchunks; ++kc) {
        char* bs_cur = Bs0 + (kc & 1) * T3_B_BYTES;
        char* bs_nxt = Bs0 + ((kc + 1) & 1) * T3_B_BYTES;
        wait_vm();          // chunk kc: A in registers, B in its stage (this wave's pieces; the barrier covers the others')
        store_a();
        barrier();          // #1: A stage and B stage kc & 1 complete for every wave
        fetch_a();          // chunk kc + 1 (zeros past the end)
        dma_b(bs_nxt);
        advance();
        mfma_phase(bs_cur);
        barrier();          // #2: every wave has read the A stage and B stage kc & 1
    }
    wait_vm();  // (the last, out-of-range fetches)

    conv_epilogue<TM, TN, true, BM, BN>(p, acc, am, rowinfo, m0, tile_m, 0);
}

}  // namespace

bool conv3_covers(const ConvArgs& a) {
    return a.C % 32 == 0 && a.cps == 0 && a.in_scale == nullptr && a.K >= 192 && a.wx3 != nullptr;
}

int launch_conv3(const ConvArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm3_kernel), T3_LDS, configured) != HN_OK) return HN_ELAUNCH;
    ConvArgs p = a;
    p.ntm = (a.M + T3_BM - 1) / T3_BM;
    p.ntn = (a.K + T3_BN - 1) / T3_BN;
    p.gm = p.ntn >= 4 ? 8 : 1;
    if (p.gm > p.ntm) p.gm = p.ntm;
    HN_LAUNCH(conv_igemm3_kernel, dim3(p.ntm * p.ntn), dim3(T3_NT), T3_LDS, stream, p);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

}  // namespace hnconv
