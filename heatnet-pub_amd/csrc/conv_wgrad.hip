// Convolution weight gradient for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// GEMM view: dW[co][(kh,kw,ci)] = sum over output pixels p of dY[p][co] * X[tap-shifted p][ci]
//   m = output channel, n = (tap, ci), reduction over pixels (28 k ... 1.8 M of them), so the output is
//   small and the reduction is split over the grid; partial tiles meet in dW through
//   global_atomic_add_f32 (dW zeroed first), or are stored directly when there is a single split.
// Both operands are pixel-major in memory (NHWC), i.e. the reduction index is the slow one: tiles are
// staged as [32 pixels][BM or BN channels] and a lane's MFMA operand is one ds_read_b32 -- lanes 0..31
// read 32 consecutive channels of pixel row 2*kk, lanes 32..63 of row 2*kk+1 (conflict-free).
//
// Loader variants: WG_VEC_BUF (16-B buffer loads through per-launch SGPR resources, out-of-range taps and
// pixel tails zero-filled by the hardware range check, pixel coordinates advanced incrementally -- no integer
// division in the loop), WG_VEC (same with plain global loads, for tensors beyond 4 GiB), WG_SCALAR (any
// channel count / alignment: 7x7x3 stem, 13- and 1-channel heads).
#include <cstdio>
#include <type_traits>

#include "hn_common.h"

#ifdef HN_EXP_PIN
#define HN_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define HN_PIN() ((void)0)
#endif

namespace {

// pixels per reduction step (template parameter BKP): 32 -> 64 KB of LDS per 128x128 workgroup (2 per CU), 16 -> 32 KB
// (3 per CU with the 146 VGPRs of the register-staged loaders)

enum : int { WG_SCALAR = 0, WG_VEC = 1, WG_VEC_BUF = 2, WG_VEC_DMA = 3 };

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
    int dw_zeroed;  // the caller guarantees dW is all zero on entry: accumulate without a fill
    int accumulate; // dW += (always through atomics, never filled)
    float* slab;    // deterministic mode: split s stores its partial dW to slab + s*K*Kdim (no atomics); NULL otherwise
    long long x_bytes, dy_bytes;
    const float* x_scale;   // f16x2: scale slots of x and of dy (hn_absmax_scale); NULL otherwise
    const float* dy_scale;
    // plane tensors (conv_wgrad4_kernel): hi / lo fp16 images of x and dy, plane strides in elements, pixel strides ldx / ldy
    // in fp16 elements; x_bytes / dy_bytes are then the extent of ONE plane
    const unsigned short* xh = nullptr;
    const unsigned short* dyh = nullptr;
    long long xh_plane = 0, dyh_plane = 0;
};

__device__ __forceinline__ f32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}

// buffer_load_dwordx4 ... offen lds: each lane's 16 B go to (wave-uniform LDS base) + lane * 16.  The builtin only
// exists in the device pass; the host pass just needs the kernel's stub.
__device__ __forceinline__ void buffer_load16_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0, 0);
#endif
}

template <int TM, int TN, int MODE, int BKP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int WG_GROUPS = BKP / 8;  // MFMA groups (8 pixels each) per step
    constexpr bool VEC = MODE != WG_SCALAR;
    constexpr bool DMA = MODE == WG_VEC_DMA;              // buffer loads that write LDS directly (no VGPR staging)
    constexpr bool BUF = MODE == WG_VEC_BUF || DMA;
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
    constexpr int ROWS_A = 256 / CA, ROWS_B = 256 / CB;   // VEC: pixel rows covered by one pass of the 256 threads
    f32x4 va[VEC ? RA : 1], vb[VEC ? RB : 1];
    float sa[VEC ? 1 : RA], sb[VEC ? 1 : RB];

    int tap = 0, ci0 = 0;
    if constexpr (VEC) {
        tap = n0 / p.C;
        ci0 = n0 - tap * p.C;
    }
    const int vkh = tap / p.KW, vkw = tap - vkh * p.KW;

    // ---- VEC state: this thread's piece column and the coordinates of its RB gathered pixels ---------------
    const int a_c4 = tid % CA, a_r0 = tid / CA;
    const int b_c4 = tid % CB, b_r0 = tid / CB;
    const bool a_col_ok = co0 + a_c4 * 4 < p.K;
    int hq[VEC ? RB : 1], wq[VEC ? RB : 1];   // input row / column hit by this tile's tap for pixel (ho, wo)
    int wo_[VEC ? RB : 1], ho_[VEC ? RB : 1];
    long long xoff[VEC ? RB : 1];             // element offset of that input pixel (+ channel piece)
    unsigned a_voff[BUF ? RA : 1];
    __amdgpu_buffer_rsrc_t rsrc_x, rsrc_dy;
    unsigned x_records = 0;
    if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int pix = c_begin * BKP + b_r0 + ROWS_B * q;
            const int n = pix / hw, rem = pix - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            ho_[q] = ho;
            wo_[q] = wo;
            hq[q] = ho * p.stride - p.pad + vkh * p.dil;
            wq[q] = wo * p.stride - p.pad + vkw * p.dil;
            xoff[q] = ((long long)(n * p.H + hq[q]) * p.W + wq[q]) * p.ldx + ci0 + b_c4 * 4;
        }
    }
    if constexpr (BUF) {
        x_records = (unsigned)p.x_bytes;
        rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_records, 0x00020000);
        rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (unsigned)p.dy_bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < RA; ++q)
            a_voff[q] = a_col_ok ? (unsigned)(((a_r0 + ROWS_A * q) * p.ldy + co0 + a_c4 * 4) * 4) : (unsigned)p.dy_bytes;
    }
    // uniform increments of xoff when a pixel advances by one step (32 pixels) and wraps rows / images
    const long long step_w = (long long)p.stride * p.ldx;
    const long long step_h = (long long)p.stride * p.W * p.ldx;
    const long long wrap_w = step_h - (long long)p.Wo * step_w;                        // wo -= Wo, ho += 1
    const long long wrap_h = (long long)p.H * p.W * p.ldx - (long long)p.Ho * step_h;  // ho -= Ho, n += 1

    // scalar path: this thread's fixed column of each tile
    const int sa_col = tid % BM, sa_row0 = tid / BM;      // rows sa_row0 + (256/BM)*q
    const int sb_col = tid % BN, sb_row0 = tid / BN;
    int s_kh = 0, s_kw = 0, s_ci = 0;
    bool s_nvalid = false;
    if constexpr (!VEC) {
        const int n = n0 + sb_col;
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

    int fetch_chunk = c_begin;  // next chunk to fetch

    auto load_a = [&]() {
        const int p0 = fetch_chunk * BKP;
        if constexpr (BUF) {
            // rows past M lie beyond the resource extent and come back as zeros.  The chunk advance is folded into
            // voffset (not soffset) so that the range check sees the full offset.
            const unsigned adv = (unsigned)((long long)p0 * p.ldy * 4);
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const unsigned off = a_voff[q] + adv;
                va[q] = buffer_load16(rsrc_dy, off < a_voff[q] ? 0xFFFFFFFFu : off, 0);
            }
        } else if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int pix = p0 + a_r0 + ROWS_A * q;
                const bool ok = (pix < p.M) & a_col_ok;
                const float* src = ok ? p.dy + ((size_t)pix * p.ldy + co0 + a_c4 * 4) : p.dy;
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                va[q] = ok ? v : z;
            }
        } else {
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int pix = p0 + sa_row0 + (256 / BM) * q, co = co0 + sa_col;
                sa[q] = (pix < p.M && co < p.K) ? p.dy[(size_t)pix * p.ldy + co] : 0.f;
            }
        }
    };

    auto load_b = [&]() {  // also advances the fetch position
        const int p0 = fetch_chunk * BKP;
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int pix = p0 + b_r0 + ROWS_B * q;
                const bool ok = (pix < p.M) & ((unsigned)hq[q] < (unsigned)p.H) & ((unsigned)wq[q] < (unsigned)p.W);
                if constexpr (BUF) {
                    vb[q] = buffer_load16(rsrc_x, ok ? (unsigned)(xoff[q] * 4) : x_records, 0);
                } else {
                    const float* src = ok ? p.x + xoff[q] : p.x;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    vb[q] = ok ? v : z;
                }
                // advance this pixel by one reduction step
                wo_[q] += BKP;
                wq[q] += BKP * p.stride;
                xoff[q] += BKP * step_w;
                while (wo_[q] >= p.Wo) {
                    wo_[q] -= p.Wo;
                    wq[q] -= p.Wo * p.stride;
                    ho_[q] += 1;
                    hq[q] += p.stride;
                    xoff[q] += wrap_w;
                    if (ho_[q] >= p.Ho) {
                        ho_[q] -= p.Ho;
                        hq[q] -= p.Ho * p.stride;
                        xoff[q] += wrap_h;
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                size_t off = 0;
                const bool ok = s_nvalid && gather_pixel(p0 + sb_row0 + (256 / BN) * q, s_kh, s_kw, off);
                sb[q] = ok ? p.x[off + s_ci] : 0.f;
            }
        }
        ++fetch_chunk;
    };

    // LDS-DMA: a wave instruction deposits 64 x 16 B = 1 KiB contiguously at a wave-uniform LDS address, which is
    // exactly this kernel's tile image (piece index = tid + 256 q, lane-linear); out-of-range offsets deposit zeros.
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma_chunk = [&](int buf) {
        if constexpr (DMA) {
            const int p0 = fetch_chunk * BKP;
            const unsigned adv = (unsigned)((long long)p0 * p.ldy * 4);
            float* a = As + buf * BKP * BM + wave_u * 256;
            float* b = Bs + buf * BKP * BN + wave_u * 256;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const unsigned off = a_voff[q] + adv;
                buffer_load16_lds(rsrc_dy, a + 1024 * q, off < a_voff[q] ? 0xFFFFFFFFu : off);
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int pix = p0 + b_r0 + ROWS_B * q;
                const bool ok = (pix < p.M) & ((unsigned)hq[q] < (unsigned)p.H) & ((unsigned)wq[q] < (unsigned)p.W);
                buffer_load16_lds(rsrc_x, b + 1024 * q, ok ? (unsigned)(xoff[q] * 4) : x_records);
                wo_[q] += BKP;
                wq[q] += BKP * p.stride;
                xoff[q] += BKP * step_w;
                while (wo_[q] >= p.Wo) {
                    wo_[q] -= p.Wo;
                    wq[q] -= p.Wo * p.stride;
                    ho_[q] += 1;
                    hq[q] += p.stride;
                    xoff[q] += wrap_w;
                    if (ho_[q] >= p.Ho) {
                        ho_[q] -= p.Ho;
                        hq[q] -= p.Ho * p.stride;
                        xoff[q] += wrap_h;
                    }
                }
            }
            ++fetch_chunk;
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
            for (int q = 0; q < RA; ++q) a[(sa_row0 + (256 / BM) * q) * BM + sa_col] = sa[q];
#pragma unroll
            for (int q = 0; q < RB; ++q) b[(sb_row0 + (256 / BN) * q) * BN + sb_col] = sb[q];
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

    // One group = 8 pixels of the reduction = 4 MFMA k-pairs; its 4*(TM+TN) ds_read_b32 fragments are
    // double-buffered in registers and fetched one group ahead of the MFMAs that consume them.
    struct Frag {
        float a[4][TM], b[4][TN];
    };
    auto read_frag = [&](int buf, int g, Frag& f) {
        const float* a = As + buf * BKP * BM + (g * 8 + lh) * BM + wm * TM * 32 + lr;
        const float* b = Bs + buf * BKP * BN + (g * 8 + lh) * BN + wn * TN * 32 + lr;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[kk][i] = a[kk * 2 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[kk][j] = b[kk * 2 * BN + j * 32];
        }
    };
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk][i], f.b[kk][j], acc[i][j], 0, 0, 0);
    };

    // same software pipeline as the forward kernel: staging registers -> other LDS buffer at the top of a step,
    // the fetch of the chunk after next issued between the MFMA groups
    auto k_step = [&](int buf, auto write_tag, auto fetch_tag) {
        Frag f[2];
        read_frag(buf, 0, f[0]);
#ifndef HN_EXP_NOWRITE
        if constexpr (decltype(write_tag)::value) store_lds(buf ^ 1);
#endif
#pragma unroll
        for (int g = 0; g < WG_GROUPS; ++g) {
            if (g + 1 < WG_GROUPS) read_frag(buf, g + 1, f[(g + 1) & 1]);
#ifndef HN_EXP_NOLOAD
            if constexpr (decltype(fetch_tag)::value) {
                if (g == (WG_GROUPS >= 4 ? 1 : 0)) load_a();
                if (g == (WG_GROUPS >= 4 ? 2 : WG_GROUPS - 1)) load_b();
            }
#endif
            mfma16(f[g & 1]);
        }
        __syncthreads();
    };

    const int nch = c_end - c_begin;
    if constexpr (DMA) {
        // two LDS buffers: the DMA of chunk kc+1 is issued at the top of step kc into the buffer every wave finished
        // reading at the previous barrier, and must have landed (vmcnt(0), emitted with the barrier) by the end of it
        if (nch > 0) {
            dma_chunk(0);
            __syncthreads();
            for (int kc = 0; kc < nch; ++kc) {
                const int buf = kc & 1;
                if (kc + 1 < nch) dma_chunk(buf ^ 1);
                Frag f[2];
                read_frag(buf, 0, f[0]);
#pragma unroll
                for (int g = 0; g < WG_GROUPS; ++g) {
                    if (g + 1 < WG_GROUPS) read_frag(buf, g + 1, f[(g + 1) & 1]);
                    mfma16(f[g & 1]);
                }
                __syncthreads();
            }
        }
    } else if (nch > 0) {
        load_a();
        load_b();
        store_lds(0);
        if (nch > 1) {
            load_a();
            load_b();
        }
        __syncthreads();
        int kc = 0;
        for (; kc + 2 < nch; ++kc) k_step(kc & 1, std::true_type{}, std::true_type{});
        if (kc + 1 < nch) {
            k_step(kc & 1, std::true_type{}, std::false_type{});
            ++kc;
        }
        k_step(kc & 1, std::false_type{}, std::false_type{});
    }

    // D row = output channel, D column = (tap, ci): column on the lane -> 128-B runs of dW per store / atomic.
    const bool full = (co0 + BM <= p.K) && (n0 + BN <= p.Kdim);
    float* out = p.slab != nullptr ? p.slab + (size_t)split * p.K * p.Kdim : p.dw;
    auto finish = [&](auto full_tag, auto atomic_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool ATOMIC = decltype(atomic_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + lr;
            const bool nok = FULL || n < p.Kdim;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = co0 + (wm * TM + i) * 32 + 4 * lh;
                float* dst = out + (size_t)rbase * p.Kdim + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (FULL || (nok && rbase + ro < p.K)) {
                        if constexpr (ATOMIC) atomicAdd(dst + (size_t)ro * p.Kdim, acc[i][j][r]);
                        else dst[(size_t)ro * p.Kdim] = acc[i][j][r];
                    }
                }
            }
        }
    };
#ifdef HN_EXP_NOEPI
    if (p.M >= 0) return;
#endif
    if (p.atomic) {
        if (full) finish(std::true_type{}, std::true_type{});
        else finish(std::false_type{}, std::true_type{});
    } else {
        if (full) finish(std::true_type{}, std::false_type{});
        else finish(std::false_type{}, std::false_type{});
    }
}


// ---- bf16x3 weight gradient ------------------------------------------------------------------------------------------
// Same GEMM on the bf16 matrix cores with fp32-equivalent accuracy (see conv_igemm.hip, MATH_X3): both operands are
// split exactly into three bf16 terms when they are staged, six v_mfma_f32_16x16x32_bf16 per product.  The reduction
// index (pixel) is the slow one of both NHWC operands, so the LDS images stay [pixel][channel] (a thread writes the
// 8 B of its 4 channels of one pixel) and the MFMA fragments -- 8 consecutive pixels of one channel per lane -- are
// fetched with the transposing LDS read ds_read_b64_tr_b16 (16 lanes read a 4-pixel x 16-channel block and receive
// one channel's 4 pixels each).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

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

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// f16x2: x * s = hi + lo (+ 2^-23 relative), s a power of two that puts the tensor's largest magnitude below 2^15
__device__ __forceinline__ void split2h(const f32x4 v, float s, f16x4& h, f16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * s;
        const _Float16 hh = (_Float16)xs;
        h[e] = hh;
        l[e] = (_Float16)(xs - (float)hh);
    }
}

__device__ __forceinline__ bf16x4 round4(const f32x4 v) {
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    return h;
}

__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
#else
    return bf16x4{};
#endif
}

// byte offset of (pixel k, channel c) in a [32][B] bf16 image.  A transposed read of one 32-lane half fetches 32 B (16
// channels) from each of pixel rows q..q+3 and 8+q..8+q+3: the 64-B column blocks are XOR-swizzled with the low pixel
// bits and the 32-B halves inside a block with pixel bit 3, so those 8 pieces fall into 8 different bank groups.
template <int B>
__device__ __forceinline__ int x3_off(int k, int c) {
    const int swz = B >= 128 ? (k & 3) : ((k >> 1) & 1);
    return k * (2 * B) + ((((c >> 5) ^ swz) << 6) | (((c & 31) ^ (((k >> 3) & 1) << 4)) * 2));
}

// NS = 3: bf16x3 as described above.  NS = 1: plain bf16 (HN_MATH_BF16, reduced precision) -- operands rounded to nearest
// bf16 when staged, one MFMA per product.  NS = 2: f16x2 -- both operands scaled by their tensor's power-of-two scale and
// split into two fp16 terms, three products (lo*hi, hi*lo, hi*hi) on v_mfma_f32_16x16x32_f16, result scaled back.
template <int TM, int TN, int NS = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const WgradArgs p) {
    constexpr int BKP = 32;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int PLANE_A = BKP * BM * 2, PLANE_B = BKP * BN * 2;  // bytes per bf16 plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);  // [NS][32][BM] bf16   dY tile
    char* Bs = As + NS * PLANE_A;              // [NS][32][BN] bf16   gathered X tile

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

    constexpr int CA = BM / 4, CB = BN / 4;               // 16-B pieces per pixel row
    constexpr int RA = (BKP * CA) / 256, RB = (BKP * CB) / 256;
    constexpr int ROWS_A = 256 / CA, ROWS_B = 256 / CB;   // pixel rows covered by one pass of the 256 threads
    f32x4 va[2][RA], vb[2][RB];                           // two staging sets: two reduction steps of loads in flight
    float s_dy = 1.f, s_x = 1.f, o_s = 1.f;
    if constexpr (NS == 2) {
        float i_dy, i_x;
        hn_h2_scales_of_slot(p.dy_scale, s_dy, i_dy);  // from the tensors' max |.| bit patterns
        hn_h2_scales_of_slot(p.x_scale, s_x, i_x);
        o_s = i_dy * i_x;
    }

    const int tap = n0 / p.C, ci0 = n0 - tap * p.C;       // a tile's column range lies inside one tap (C % BN == 0)
    const int vkh = tap / p.KW, vkw = tap - vkh * p.KW;
    const int a_c4 = tid % CA, a_r0 = tid / CA;
    const int b_c4 = tid % CB, b_r0 = tid / CB;
    const bool a_col_ok = co0 + a_c4 * 4 < p.K;
    int hq[RB], wq[RB], wo_[RB], ho_[RB];
    long long xoff[RB];
    unsigned a_voff[RA];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int pix = c_begin * BKP + b_r0 + ROWS_B * q;
        const int n = pix / hw, rem = pix - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        ho_[q] = ho;
        wo_[q] = wo;
        hq[q] = ho * p.stride - p.pad + vkh * p.dil;
        wq[q] = wo * p.stride - p.pad + vkw * p.dil;
        xoff[q] = ((long long)(n * p.H + hq[q]) * p.W + wq[q]) * p.ldx + ci0 + b_c4 * 4;
    }
    const unsigned x_records = (unsigned)p.x_bytes;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_dy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (unsigned)p.dy_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < RA; ++q)
        a_voff[q] = a_col_ok ? (unsigned)(((a_r0 + ROWS_A * q) * p.ldy + co0 + a_c4 * 4) * 4) : (unsigned)p.dy_bytes;
    const long long step_w = (long long)p.stride * p.ldx;
    const long long step_h = (long long)p.stride * p.W * p.ldx;
    const long long wrap_w = step_h - (long long)p.Wo * step_w;
    const long long wrap_h = (long long)p.H * p.W * p.ldx - (long long)p.Ho * step_h;

    int fetch_chunk = c_begin;
    auto load_chunk = [&](f32x4 (&da)[RA], f32x4 (&db)[RB]) {
        const int p0 = fetch_chunk * BKP;
        const unsigned adv = (unsigned)((long long)p0 * p.ldy * 4);
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            const unsigned off = a_voff[q] + adv;
            da[q] = buffer_load16(rsrc_dy, off < a_voff[q] ? 0xFFFFFFFFu : off, 0);
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int pix = p0 + b_r0 + ROWS_B * q;
            const bool ok = (pix < p.M) & ((unsigned)hq[q] < (unsigned)p.H) & ((unsigned)wq[q] < (unsigned)p.W);
            db[q] = buffer_load16(rsrc_x, ok ? (unsigned)(xoff[q] * 4) : x_records, 0);
            wo_[q] += BKP;
            wq[q] += BKP * p.stride;
            xoff[q] += BKP * step_w;
            while (wo_[q] >= p.Wo) {
                wo_[q] -= p.Wo;
                wq[q] -= p.Wo * p.stride;
                ho_[q] += 1;
                hq[q] += p.stride;
                xoff[q] += wrap_w;
                if (ho_[q] >= p.Ho) {
                    ho_[q] -= p.Ho;
                    hq[q] -= p.Ho * p.stride;
                    xoff[q] += wrap_h;
                }
            }
        }
        ++fetch_chunk;
    };
    auto store_chunk = [&](const f32x4 (&sa)[RA], const f32x4 (&sb)[RB]) {
#pragma unroll
        for (int q = 0; q < RA; ++q) {
            char* dst = As + x3_off<BM>(a_r0 + ROWS_A * q, a_c4 * 4);
            if constexpr (NS == 3) {
                bf16x4 h, m, l;
                split3(sa[q], h, m, l);
                *reinterpret_cast<bf16x4*>(dst) = h;
                *reinterpret_cast<bf16x4*>(dst + PLANE_A) = m;
                *reinterpret_cast<bf16x4*>(dst + 2 * PLANE_A) = l;
            } else if constexpr (NS == 2) {
                f16x4 h, l;
                split2h(sa[q], s_dy, h, l);
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + PLANE_A) = l;
            } else {
                *reinterpret_cast<bf16x4*>(dst) = round4(sa[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            char* dst = Bs + x3_off<BN>(b_r0 + ROWS_B * q, b_c4 * 4);
            if constexpr (NS == 3) {
                bf16x4 h, m, l;
                split3(sb[q], h, m, l);
                *reinterpret_cast<bf16x4*>(dst) = h;
                *reinterpret_cast<bf16x4*>(dst + PLANE_B) = m;
                *reinterpret_cast<bf16x4*>(dst + 2 * PLANE_B) = l;
            } else if constexpr (NS == 2) {
                f16x4 h, l;
                split2h(sb[q], s_x, h, l);
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + PLANE_B) = l;
            } else {
                *reinterpret_cast<bf16x4*>(dst) = round4(sb[q]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read duty of this lane (tools/probes/tr_read_probe.hip): the 16-lane group g = lane >> 4 fetches pixels
    // 8g .. 8g+7 (the k range of its lanes in the 16x16x32 operand) of one 16-channel block; lane 4q+pp of the group
    // supplies the address of pixel row 8g+q (then +4), channels 4pp..4pp+3, and receives the pixels of channel lane & 15.
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int krow = 8 * tg + tq;
    int offA[2 * TM], offB[2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) offA[i] = x3_off<BM>(krow, (wm * TM * 2 + i) * 16 + 4 * tp);
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) offB[j] = x3_off<BN>(krow, (wn * TN * 2 + j) * 16 + 4 * tp);

    // accumulator element r of acc[i][j]: registers 4s..4s+3 hold the 16x16 sub-block s = 2 * (row half) + (column half);
    // column = 16 * ((r >> 2) & 1) + (lane & 15), row = 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3)
    auto x3_k32 = [&]() {
        bf16x8 fb[2 * TN][NS];
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3) {
                const char* b = Bs + s3 * PLANE_B + offB[bj];
                const bf16x4 lo = lds_read_tr(b), hi = lds_read_tr(b + 4 * (2 * BN));
                fb[bj][s3] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
        for (int bi = 0; bi < 2 * TM; ++bi) {
            bf16x8 fa[NS];
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3) {
                const char* a = As + s3 * PLANE_A + offA[bi];
                const bf16x4 lo = lds_read_tr(a), hi = lds_read_tr(a + 4 * (2 * BM));
                fa[s3] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                f32x16& blk = acc[bi >> 1][bj >> 1];
                f32x4 c = {blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
                if constexpr (NS == 2) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, fa[0]), al = __builtin_bit_cast(f16x8, fa[1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, fb[bj][0]), bl = __builtin_bit_cast(f16x8, fb[bj][1]);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);  // hi*hi
                }
                if constexpr (NS == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][1], c, 0, 0, 0);  // mid*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][2], c, 0, 0, 0);  // hi*lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[bj][0], c, 0, 0, 0);  // lo*hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][1], c, 0, 0, 0);  // hi*mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[bj][0], c, 0, 0, 0);  // mid*hi
                }
                if constexpr (NS != 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[bj][0], c, 0, 0, 0);  // hi*hi
                blk[sb] = c[0];
                blk[sb + 1] = c[1];
                blk[sb + 2] = c[2];
                blk[sb + 3] = c[3];
            }
        }
    };
    // every lane takes part in every transposed read (EXEC all ones): the loop below has no lane-dependent control flow
    // two staging sets, fetch distance two reduction steps: a set is refilled (chunk kc+2) right after it has been
    // split and stored (chunk kc)
    auto x3_step = [&](bool refill, int cur) {
        store_chunk(va[cur], vb[cur]);
        if (refill) load_chunk(va[cur], vb[cur]);
        __syncthreads();
        x3_k32();
        __syncthreads();
    };
    const int nch = c_end - c_begin;
    if (nch > 0) {
        load_chunk(va[0], vb[0]);
        if (nch > 1) load_chunk(va[1], vb[1]);
        int kc = 0;
        for (; kc + 1 < nch; kc += 2) {
            x3_step(kc + 2 < nch, 0);
            x3_step(kc + 3 < nch, 1);
        }
        if (kc < nch) x3_step(false, 0);
    }

    const bool full = (co0 + BM <= p.K) && (n0 + BN <= p.Kdim);
    float* out = p.slab != nullptr ? p.slab + (size_t)split * p.K * p.Kdim : p.dw;
    auto finish = [&](auto full_tag, auto atomic_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool ATOMIC = decltype(atomic_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = co0 + (wm * TM + i) * 32 + 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
                    const int n = n0 + (wn * TN + j) * 32 + 16 * ((r >> 2) & 1) + (lane & 15);
                    if (FULL || (n < p.Kdim && row < p.K)) {
                        float* dst = out + (size_t)row * p.Kdim + n;
                        const float v = NS == 2 ? acc[i][j][r] * o_s : acc[i][j][r];
                        if constexpr (ATOMIC) atomicAdd(dst, v);
                        else *dst = v;
                    }
                }
            }
        }
    };
    if (p.atomic) {
        if (full) finish(std::true_type{}, std::true_type{});
        else finish(std::false_type{}, std::true_type{});
    } else {
        if (full) finish(std::true_type{}, std::false_type{});
        else finish(std::false_type{}, std::false_type{});
    }
}

// ---- f16x2 weight gradient on plane tensors: everything by LDS-DMA ---------------------------------------------------------
// Same GEMM, LDS images ([32 pixels][channels] fp16 per plane, x3_off swizzle), transposed fragment reads, MFMA order and
// epilogue as conv_wgrad_x3_kernel<TM, TN, 2>, but both operands arrive as plane tensors (hi / lo fp16 images written by the
// pass that produced them, include/heatnet_hip.h) and go global -> LDS with buffer_load_dwordx4 ... lds: no staging registers, no
// split arithmetic, no ds_write.  The swizzle is applied on the source side: the lane that fills 16-byte position q of pixel
// row k fetches the channel piece x3_off maps there.  Two LDS stages, one barrier per 32-pixel reduction step.  Given planes
// made with the same scale the result of a launch with the same reduction split is bitwise that of the register kernel.
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0, 0);
#endif
}

// channel piece (8 channels = 16 B) stored at 16-byte position pp of pixel row k of a [32][B] image (inverse of x3_off)
template <int B>
__device__ __forceinline__ int x3_piece_at(int k, int pp) {
    const int swz = B >= 128 ? (k & 3) : ((k >> 1) & 1);
    return (((pp >> 2) ^ swz) << 2) | ((pp & 3) ^ (((k >> 3) & 1) << 1));
}

// NPL = 1: the mixed-precision bf16 arithmetic on one bf16 image per operand (no scales, one MFMA per product).
// WM x WN waves, each TM x TN blocks of 32 x 32: 2 x 2 waves (two workgroups per CU) up to 128 x 128; 2 x 4 waves with 4 x 2 blocks
// = 256 x 256 (one workgroup per CU, two stages of 64 KB): half the operand bytes per multiply-add -- the reduction loop of the
// smaller tiles is bound by what a CU can pull out of its L2, like the forward kernels' (DESIGN.md section 6, round 4).
template <int TM, int TN, int NST = 2, int NPL = 2, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4) ? 2 : 1) void conv_wgrad4_kernel(const WgradArgs p) {
    constexpr int BKP = 32, NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int PLANE_A = BKP * BM * 2, PLANE_B = BKP * BN * 2;  // bytes per 16-bit plane
    constexpr int STAGE = NPL * (PLANE_A + PLANE_B);
    constexpr int ROWS_A = 1024 / (2 * BM), ROWS_B = 1024 / (2 * BN);  // pixel rows one 1-KiB DMA covers
    constexpr int PPR_A = BM / 8, PPR_B = BN / 8;                       // 16-byte pieces per pixel row
    constexpr int QA = BKP / ROWS_A / NW, QB = BKP / ROWS_B / NW;         // DMA row groups per wave, plane and stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = p.ntm * p.ntn;
    const int logical = hn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tile_m = tile % p.ntm, tile_n = tile / p.ntm;
    const int co0 = tile_m * BM, n0 = tile_n * BN;
    const int total_chunks = (p.M + BKP - 1) / BKP;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(total_chunks, c_begin + p.chunks_per_split);
    const int hw = p.Ho * p.Wo;
    float o_s = 1.f, o_s2 = 1.f;
    if constexpr (NPL == 2) {
        float s_dy, s_x;
        hn_h2_scales_of_slot(p.dy_scale, s_dy, o_s);
        hn_h2_scales_of_slot(p.x_scale, s_x, o_s2);
    }

    const int tap = n0 / p.C, ci0 = n0 - tap * p.C;  // a tile's column range lies inside one tap (C % BN == 0)
    const int vkh = tap / p.KW, vkw = tap - vkh * p.KW;

    const unsigned x_records = (unsigned)p.x_bytes, dy_records = (unsigned)p.dy_bytes;
    const __amdgpu_buffer_rsrc_t rsrc_x0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xh), 0, x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xh + p.xh_plane), 0, x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_d0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dyh), 0, dy_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_d1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dyh + p.dyh_plane), 0, dy_records, 0x00020000);

    // ---- A (dY): this lane's pixel rows and channel pieces --------------------------------------------------------
    unsigned a_voff[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int k = (wave + NW * q) * ROWS_A + lane / PPR_A;
        const int cp = x3_piece_at<BM>(k, lane % PPR_A);
        const bool ok = co0 + cp * 8 < p.K;
        a_voff[q] = ok ? (unsigned)((k * p.ldy + co0 + cp * 8) * 2) : dy_records;
    }
    // ---- B (X at this tile's tap): pixel coordinates tracked incrementally (no division in the loop) -----------------
    int hq[QB], wq[QB], wo_[QB], ho_[QB];
    long long xoff[QB];  // fp16-element offset of (pixel, channel piece)
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int k = (wave + NW * q) * ROWS_B + lane / PPR_B;
        const int cp = x3_piece_at<BN>(k, lane % PPR_B);
        const int pix = c_begin * BKP + k;
        const int n = pix / hw, rem = pix - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        ho_[q] = ho;
        wo_[q] = wo;
        hq[q] = ho * p.stride - p.pad + vkh * p.dil;
        wq[q] = wo * p.stride - p.pad + vkw * p.dil;
        xoff[q] = ((long long)(n * p.H + hq[q]) * p.W + wq[q]) * p.ldx + ci0 + cp * 8;
    }
    const long long step_w = (long long)p.stride * p.ldx;
    const long long step_h = (long long)p.stride * p.W * p.ldx;
    const long long wrap_w = step_h - (long long)p.Wo * step_w;
    const long long wrap_h = (long long)p.H * p.W * p.ldx - (long long)p.Ho * step_h;

    int fetch_chunk = c_begin;
    auto dma_chunk = [&](char* st) {  // chunk fetch_chunk -> stage st (zeros past M / outside the image / past this split); advances
        const int p0 = fetch_chunk * BKP;
        const bool live = fetch_chunk < c_end;
        const unsigned adv = (unsigned)((long long)p0 * p.ldy * 2);
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const unsigned off = a_voff[q] + adv;
            const unsigned vo = (off < a_voff[q] || !live) ? 0xFFFFFFFFu : off;
            char* dst = st + (wave + NW * q) * 1024;
            wg_dma16(rsrc_d0, dst, vo);
            if constexpr (NPL == 2) wg_dma16(rsrc_d1, dst + PLANE_A, vo);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int k = (wave + NW * q) * ROWS_B + lane / PPR_B;
            const bool ok = live & (p0 + k < p.M) & ((unsigned)hq[q] < (unsigned)p.H) & ((unsigned)wq[q] < (unsigned)p.W);
            const unsigned vo = ok ? (unsigned)(xoff[q] * 2) : x_records;
            char* dst = st + NPL * PLANE_A + (wave + NW * q) * 1024;
            wg_dma16(rsrc_x0, dst, vo);
            if constexpr (NPL == 2) wg_dma16(rsrc_x1, dst + PLANE_B, vo);
            wo_[q] += BKP;
            wq[q] += BKP * p.stride;
            xoff[q] += BKP * step_w;
            while (wo_[q] >= p.Wo) {
                wo_[q] -= p.Wo;
                wq[q] -= p.Wo * p.stride;
                ho_[q] += 1;
                hq[q] += p.stride;
                xoff[q] += wrap_w;
                if (ho_[q] >= p.Ho) {
                    ho_[q] -= p.Ho;
                    hq[q] -= p.Ho * p.stride;
                    xoff[q] += wrap_h;
                }
            }
        }
        ++fetch_chunk;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read duty of this lane: see conv_wgrad_x3_kernel
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int krow = 8 * tg + tq;
    int offA[2 * TM], offB[2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) offA[i] = x3_off<BM>(krow, (wm * TM * 2 + i) * 16 + 4 * tp);
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) offB[j] = NPL * PLANE_A + x3_off<BN>(krow, (wn * TN * 2 + j) * 16 + 4 * tp);

    auto k32 = [&](const char* st) {
        bf16x8 fb[2 * TN][NPL];  // (fp16 bit patterns under NPL == 2: re-typed at the MFMA)
#pragma unroll
        for (int bj = 0; bj < 2 * TN; ++bj)
#pragma unroll
            for (int s2 = 0; s2 < NPL; ++s2) {
                const char* b = st + s2 * PLANE_B + offB[bj];
                const bf16x4 lo = lds_read_tr(b), hi = lds_read_tr(b + 4 * (2 * BN));
                fb[bj][s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
        for (int bi = 0; bi < 2 * TM; ++bi) {
            bf16x8 fa[NPL];
#pragma unroll
            for (int s2 = 0; s2 < NPL; ++s2) {
                const char* a = st + s2 * PLANE_A + offA[bi];
                const bf16x4 lo = lds_read_tr(a), hi = lds_read_tr(a + 4 * (2 * BM));
                fa[s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            f32x4 c[2 * TN];
#pragma unroll
            for (int bj = 0; bj < 2 * TN; ++bj) {
                const int sb = 4 * (2 * (bi & 1) + (bj & 1));
                const f32x16& blk = acc[bi >> 1][bj >> 1];
                c[bj] = f32x4{blk[sb], blk[sb + 1], blk[sb + 2], blk[sb + 3]};
            }
            // per accumulator: lo*hi, hi*lo, hi*hi (the register kernel's order); consecutive MFMAs hit different accumulators
            if constexpr (NPL == 2) {
                const f16x8 ah = __builtin_bit_cast(f16x8, fa[0]), al = __builtin_bit_cast(f16x8, fa[NPL - 1]);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj)
                    c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(f16x8, fb[bj][0]), c[bj], 0, 0, 0);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj)
                    c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, fb[bj][NPL - 1]), c[bj], 0, 0, 0);
#pragma unroll
                for (int bj = 0; bj < 2 * TN; ++bj)
                    c[bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, fb[bj][0]), c[bj], 0, 0, 0);
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

    auto barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // NST stages, the DMA runs NST - 1 reduction steps ahead (every step issues the same number of DMAs -- out of range past
    // the end of the split -- so that the counted wait below always means "chunk kc has landed")
    constexpr int PER_STEP = NPL * (QA + QB);
    static_assert(PER_STEP * (NST - 2) < 64, "vmcnt immediate");
    const int nch = c_end - c_begin;
    if (nch > 0) {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st) dma_chunk(lds + st * STAGE);
        int st_cur = 0, st_free = NST - 1;
        for (int kc = 0; kc < nch; ++kc) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP * (NST - 2)) : "memory");  // chunk kc has landed (this wave's pieces)
            barrier();                                                                   // ... everybody's; stage st_free is free
            dma_chunk(lds + st_free * STAGE);
            k32(lds + st_cur * STAGE);
            st_free = st_cur;
            st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    const bool full = (co0 + BM <= p.K) && (n0 + BN <= p.Kdim);
    float* out = p.slab != nullptr ? p.slab + (size_t)split * p.K * p.Kdim : p.dw;
    auto finish = [&](auto full_tag, auto atomic_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool ATOMIC = decltype(atomic_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = co0 + (wm * TM + i) * 32 + 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3);
                    const int n = n0 + (wn * TN + j) * 32 + 16 * ((r >> 2) & 1) + (lane & 15);
                    if (FULL || (n < p.Kdim && row < p.K)) {
                        float* dst = out + (size_t)row * p.Kdim + n;
                        const float v = NPL == 2 ? acc[i][j][r] * o_s * o_s2 : acc[i][j][r];
                        if constexpr (ATOMIC) atomicAdd(dst, v);
                        else *dst = v;
                    }
                }
            }
        }
    };
    if (p.atomic) {
        if (full) finish(std::true_type{}, std::true_type{});
        else finish(std::false_type{}, std::true_type{});
    } else {
        if (full) finish(std::true_type{}, std::false_type{});
        else finish(std::false_type{}, std::false_type{});
    }
}

// X3: 0 = fp32 MFMA kernel, 3 = bf16x3, 1 = bf16, 2 = f16x2 (planes per operand of conv_wgrad_x3_kernel)
template <int TM, int TN, int MODE, int BKP, int X3 = 0>
int launch_wgrad(WgradArgs a, int split_k, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    // X3 == 4 / 5: plane tensors, two / three LDS stages of two fp16 planes per operand (conv_wgrad4_kernel); 6 / 7: ONE bf16
    // plane per operand (the mixed-precision bf16 arithmetic), three / four stages
    constexpr bool P4 = X3 >= 4;
    constexpr int NPL4 = X3 >= 6 ? 1 : 2;
    constexpr int NST4 = X3 == 4 ? 2 : (X3 == 7 ? 4 : 3);
    constexpr int lds = P4 ? NST4 * NPL4 * BKP * (BM + BN) * 2
                           : (X3 ? X3 * BKP * (BM + BN) * 2 : 2 * BKP * (BM + BN) * (int)sizeof(float));
    const void* kern = P4 ? reinterpret_cast<const void*>(conv_wgrad4_kernel<TM, TN, NST4, NPL4>)
                          : (X3 ? reinterpret_cast<const void*>(conv_wgrad_x3_kernel<TM, TN, (X3 && !P4) ? X3 : 3>)
                                : reinterpret_cast<const void*>(conv_wgrad_kernel<TM, TN, MODE, BKP>));
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(kern, lds, configured) != HN_OK) return HN_ELAUNCH;
    a.ntm = (a.K + BM - 1) / BM;
    a.ntn = (a.Kdim + BN - 1) / BN;
    const int tiles = a.ntm * a.ntn;
    const int total_chunks = (a.M + BKP - 1) / BKP;
    if (split_k <= 0) {
        // Fill whole rounds of the 512 workgroups the chip holds (2 per CU): the largest split with
        // tiles * split <= 512 * rounds.  0 / -1: two rounds; -2 / -3 / -4 / -5: one / three / four / six rounds.
        const int code = split_k;
        const int rounds = code == -2 ? 1 : code == -3 ? 3 : code == -4 ? 4 : code == -5 ? 6 : 2;
        split_k = (512 * rounds) / tiles;
        const int cap = (total_chunks * BKP / 32 + 3) / 4;  // at least 128 pixels of reduction per split
        if (split_k > cap) split_k = cap;
        if (split_k < 1) split_k = 1;
    }
    if (split_k > total_chunks) split_k = total_chunks;
    a.chunks_per_split = (total_chunks + split_k - 1) / split_k;
    split_k = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.atomic = (split_k > 1 || a.accumulate) && a.slab == nullptr;
    if (a.atomic && !a.dw_zeroed && !a.accumulate) {
        if (hipMemsetAsync(a.dw, 0, (size_t)a.K * a.Kdim * sizeof(float), stream) != hipSuccess) return HN_ELAUNCH;
    }
    if constexpr (P4) HN_LAUNCH((conv_wgrad4_kernel<TM, TN, NST4, NPL4>), dim3(tiles * split_k), dim3(256), lds, stream, a);
    else if constexpr (X3 != 0) HN_LAUNCH((conv_wgrad_x3_kernel<TM, TN, (X3 && !P4) ? X3 : 3>), dim3(tiles * split_k), dim3(256), lds, stream, a);
    else HN_LAUNCH((conv_wgrad_kernel<TM, TN, MODE, BKP>), dim3(tiles * split_k), dim3(256), lds, stream, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// 256 x 256 tile of the plane-tensor kernel (eight waves, one workgroup per CU): Cout % 256 == 0, Cin % 256 == 0
int launch_wgrad4_huge(WgradArgs a, int split_k, hipStream_t stream) {
    constexpr int BM = 256, BN = 256, BKP = 32;
    constexpr int lds = 2 * 2 * BKP * (BM + BN) * 2;
    const void* kern = reinterpret_cast<const void*>(conv_wgrad4_kernel<4, 2, 2, 2, 2, 4>);
    static std::atomic<unsigned long long> configured{0};
    if (hn_ensure_dynamic_lds(kern, lds, configured) != HN_OK) return HN_ELAUNCH;
    a.ntm = (a.K + BM - 1) / BM;
    a.ntn = (a.Kdim + BN - 1) / BN;
    const int tiles = a.ntm * a.ntn;
    const int total_chunks = (a.M + BKP - 1) / BKP;
    if (split_k <= 0) {  // whole rounds of the 256 workgroups the chip holds (as launch_wgrad: 0 / -1 two rounds, -2 one, -3 / -4: 3 / 4)
        const int code = split_k;
        const int rounds = code == -2 ? 1 : code == -3 ? 3 : code == -4 ? 4 : code == -5 ? 6 : 2;
        split_k = (256 * rounds) / tiles;
        const int cap = (total_chunks * BKP / 32 + 3) / 4;
        if (split_k > cap) split_k = cap;
        if (split_k < 1) split_k = 1;
    }
    if (split_k > total_chunks) split_k = total_chunks;
    a.chunks_per_split = (total_chunks + split_k - 1) / split_k;
    split_k = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.atomic = (split_k > 1 || a.accumulate) && a.slab == nullptr;
    if (a.atomic && !a.dw_zeroed && !a.accumulate) {
        if (hipMemsetAsync(a.dw, 0, (size_t)a.K * a.Kdim * sizeof(float), stream) != hipSuccess) return HN_ELAUNCH;
    }
    HN_LAUNCH((conv_wgrad4_kernel<4, 2, 2, 2, 2, 4>), dim3(tiles * split_k), dim3(512), lds, stream, a);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

template <int MODE, int BKP, int X3 = 0>
int dispatch_wgrad(const WgradArgs& a, bool wide_m, bool wide_n, int split_k, hipStream_t s) {
    if (wide_m)
        return wide_n ? launch_wgrad<2, 2, MODE, BKP, X3>(a, split_k, s) : launch_wgrad<2, 1, MODE, BKP, X3>(a, split_k, s);
    return wide_n ? launch_wgrad<1, 2, MODE, BKP, X3>(a, split_k, s) : launch_wgrad<1, 1, MODE, BKP, X3>(a, split_k, s);
}

}  // namespace

// dw = (accumulate ? dw : 0) + slab[0] + slab[1] + ... in that order: the fixed-order second pass of the deterministic mode
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               long long n, int splits, int accumulate) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = accumulate ? dw[i] : 0.f;
        for (int s = 0; s < splits; ++s) v += slab[(long long)s * n + i];
        dw[i] = v;
    }
}

namespace {
int wgrad_impl(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw, const HnConvDesc* d,
               int split_k, int variant, int flags, float* slab, void* stream);

int wgrad_det_impl(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw, const HnConvDesc* d,
                   int split_k, int variant, int flags, float* workspace, void* stream) {
    if (d == nullptr || workspace == nullptr || split_k <= 0) return HN_EINVAL;
    if (d->N <= 0 || d->Hout <= 0 || d->Wout <= 0) return HN_EINVAL;
    const long long total_chunks = ((long long)d->N * d->Hout * d->Wout + 31) / 32;  // reduction steps of 32 pixels
    long long s = split_k < total_chunks ? split_k : total_chunks;
    const long long cps = (total_chunks + s - 1) / s;
    s = (total_chunks + cps - 1) / cps;  // what launch_wgrad arrives at for this request
    const int accumulate = (flags & HN_WGRAD_ACCUMULATE) != 0;
    variant = (variant & ~3) | 1;        // 32-pixel reduction step (the split arithmetic above)
    if (s == 1 && !accumulate) return wgrad_impl(x, x_scale, dy, dy_scale, dw, d, 1, variant, 0, nullptr, stream);  // single store pass
    const int rc = wgrad_impl(x, x_scale, dy, dy_scale, dw, d, (int)s, variant, flags, workspace, stream);
    if (rc != HN_OK) return rc;
    const long long n = (long long)d->Cout * d->KH * d->KW * d->Cin;
    const long long blocks = (n + 255) / 256;
    HN_LAUNCH(wgrad_slab_reduce_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
              static_cast<hipStream_t>(stream), workspace, dw, n, (int)s, accumulate);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

int wgrad_impl(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw, const HnConvDesc* d,
               int split_k, int variant, int flags, float* slab, void* stream) {
    if (x == nullptr || dy == nullptr || dw == nullptr || d == nullptr) return HN_EINVAL;
    if ((x_scale == nullptr) != (dy_scale == nullptr) || variant < 0) return HN_EINVAL;
    if (x_scale != nullptr) variant = (variant & 63) | (4 << 6);  // f16x2
    else if (variant >= (4 << 6)) return HN_EINVAL;                // ... needs the operand scales
    if (d->N <= 0 || d->Hin <= 0 || d->Win <= 0 || d->Cin <= 0 || d->Hout <= 0 || d->Wout <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->ldx < d->Cin || d->ldy < d->Cout)
        return HN_EINVAL;
    // same encoding as hn_conv2d_fwd; the K-step field selects the pixel step (1 -> 32, 2 -> 16; fp32 MFMA only)
    const int bk_sel = variant & 3, tm_sel = (variant >> 2) & 3, tn_sel = (variant >> 4) & 3, math_sel = variant >> 6;
    if (bk_sel == 3 || tm_sel == 3 || tn_sel == 3) return HN_EINVAL;
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.H = d->Hin; a.W = d->Win; a.C = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.K = d->Cout; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.M = d->N * d->Hout * d->Wout;
    a.Kdim = d->KH * d->KW * d->Cin;
    a.ntm = a.ntn = a.chunks_per_split = a.atomic = 0;
    a.slab = slab;
    a.x_scale = x_scale;
    a.dy_scale = dy_scale;
    a.dw_zeroed = (flags & HN_WGRAD_DW_ZEROED) != 0;
    a.accumulate = (flags & HN_WGRAD_ACCUMULATE) != 0;
    a.x_bytes = (((long long)d->N * d->Hin * d->Win - 1) * d->ldx + d->Cin) * 4;
    a.dy_bytes = (((long long)a.M - 1) * d->ldy + d->Cout) * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool aligned = (d->ldx % 4 == 0) && (d->ldy % 4 == 0) && hn_aligned16(x) && hn_aligned16(dy);
    bool wide_m = d->Cout > 64;
    if (tm_sel) wide_m = tm_sel == 2;
    if (aligned && d->Cout % 4 == 0 && d->Cin % 64 == 0) {
        bool wide_n = d->Cin % 128 == 0;
        if (tn_sel == 1) wide_n = false;  // a 128-wide column tile needs Cin % 128 == 0; 64 is always allowed here
        const bool buf_ok = a.x_bytes < 0xFFFF0000ll && a.dy_bytes + 64ll * d->ldy * 4 < 0xFFFF0000ll;
        const bool p16 = bk_sel == 2;
        // bf16x3: tiles are whole 64-channel blocks of dY as well (no partial column pieces in the transposed reads)
        const int math = math_sel ? math_sel - 1 : hn_get_conv_math();  // HN_MATH_*
        if (math != HN_MATH_F32 && buf_ok && d->Cout % 64 == 0) {
            if (d->Cout % 128 != 0) wide_m = false;
            if (math == HN_MATH_F16X2) return dispatch_wgrad<WG_VEC_BUF, 32, 2>(a, wide_m, wide_n, split_k, s);
            return math == HN_MATH_BF16 ? dispatch_wgrad<WG_VEC_BUF, 32, 1>(a, wide_m, wide_n, split_k, s)
                                        : dispatch_wgrad<WG_VEC_BUF, 32, 3>(a, wide_m, wide_n, split_k, s);
        }
        if (math == HN_MATH_F16X2) return HN_ESHAPE;  // the caller chose the operand format: no other kernel takes the scales
#ifdef HN_WG_DMA
        if (buf_ok) return dispatch_wgrad<WG_VEC_DMA, 32>(a, wide_m, wide_n, split_k, s);  // experiment: LDS-DMA staging
#endif
        if (buf_ok)
            return p16 ? dispatch_wgrad<WG_VEC_BUF, 16>(a, wide_m, wide_n, split_k, s)
                       : dispatch_wgrad<WG_VEC_BUF, 32>(a, wide_m, wide_n, split_k, s);
        return dispatch_wgrad<WG_VEC, 32>(a, wide_m, wide_n, split_k, s);
    }
    if (x_scale != nullptr) return HN_ESHAPE;
    bool wide_n = a.Kdim > 64;
    if (tn_sel) wide_n = tn_sel == 2;
    return dispatch_wgrad<WG_SCALAR, 32>(a, wide_m, wide_n, split_k, s);
}

// plane tensors: conv_wgrad4_kernel, or nothing (HN_ESHAPE)
int wgrad_planes_impl(const void* xh, long long xh_plane, const float* x_scale, const void* dyh, long long dyh_plane,
                      const float* dy_scale, float* dw, const HnConvDesc* d, int split_k, int variant, int flags, float* slab,
                      void* stream) {
    if (xh == nullptr || dyh == nullptr || (x_scale == nullptr) != (dy_scale == nullptr) || dw == nullptr || d == nullptr ||
        xh_plane <= 0 || dyh_plane <= 0 || variant < 0)
        return HN_EINVAL;
    const bool bf16 = x_scale == nullptr;  // one bf16 plane per operand, no scales
    if (d->N <= 0 || d->Hin <= 0 || d->Win <= 0 || d->Cin <= 0 || d->Hout <= 0 || d->Wout <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->ldx < d->Cin || d->ldy < d->Cout)
        return HN_EINVAL;
    const int tm_sel = (variant >> 2) & 3, tn_sel = (variant >> 4) & 3;
    const bool huge = tm_sel == 3 && tn_sel == 3;  // the 256 x 256 tile (f16x2 planes only)
    if ((tm_sel == 3 || tn_sel == 3) && !huge) return HN_EINVAL;
    WgradArgs a;
    a.x = nullptr; a.dy = nullptr; a.dw = dw;
    a.xh = static_cast<const unsigned short*>(xh); a.dyh = static_cast<const unsigned short*>(dyh);
    a.xh_plane = xh_plane; a.dyh_plane = dyh_plane;
    a.H = d->Hin; a.W = d->Win; a.C = d->Cin; a.ldx = d->ldx;
    a.Ho = d->Hout; a.Wo = d->Wout; a.K = d->Cout; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.M = d->N * d->Hout * d->Wout;
    a.Kdim = d->KH * d->KW * d->Cin;
    a.ntm = a.ntn = a.chunks_per_split = a.atomic = 0;
    a.slab = slab;
    a.x_scale = x_scale;
    a.dy_scale = dy_scale;
    a.dw_zeroed = (flags & HN_WGRAD_DW_ZEROED) != 0;
    a.accumulate = (flags & HN_WGRAD_ACCUMULATE) != 0;
    a.x_bytes = (((long long)d->N * d->Hin * d->Win - 1) * d->ldx + d->Cin) * 2;
    a.dy_bytes = (((long long)a.M - 1) * d->ldy + d->Cout) * 2;
    if (xh_plane < a.x_bytes / 2 || dyh_plane < a.dy_bytes / 2) return HN_EINVAL;
    const bool ok = d->Cin % 64 == 0 && d->Cout % 64 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && hn_aligned16(xh) &&
                    hn_aligned16(dyh) && xh_plane % 8 == 0 && dyh_plane % 8 == 0 && a.x_bytes < 0xFFFF0000ll &&
                    a.dy_bytes + 64ll * d->ldy * 2 < 0xFFFF0000ll;
    if (!ok) return HN_ESHAPE;
    if (huge) {
        if (bf16 || d->Cout % 256 != 0 || d->Cin % 256 != 0) return HN_ESHAPE;
        return launch_wgrad4_huge(a, split_k, static_cast<hipStream_t>(stream));
    }
    bool wide_m = d->Cout % 128 == 0, wide_n = d->Cin % 128 == 0;
    if (tm_sel == 1) wide_m = false;
    if (tn_sel == 1) wide_n = false;
    // variant & 3 == 2: three LDS stages (the DMA two reduction steps ahead) where the tile leaves room for them with two
    // workgroups per CU, i.e. not on 128 x 128
    if (bf16)  // (a stage is half the size: three stages everywhere, four on request)
        return (variant & 3) == 2 ? dispatch_wgrad<WG_VEC_BUF, 32, 7>(a, wide_m, wide_n, split_k, static_cast<hipStream_t>(stream))
                                  : dispatch_wgrad<WG_VEC_BUF, 32, 6>(a, wide_m, wide_n, split_k, static_cast<hipStream_t>(stream));
    if ((variant & 3) == 2 && !(wide_m && wide_n))
        return dispatch_wgrad<WG_VEC_BUF, 32, 5>(a, wide_m, wide_n, split_k, static_cast<hipStream_t>(stream));
    return dispatch_wgrad<WG_VEC_BUF, 32, 4>(a, wide_m, wide_n, split_k, static_cast<hipStream_t>(stream));
}
}  // namespace

extern "C" int hn_conv2d_wgrad_planes(const void* x_planes, int64_t x_plane_elems, const float* x_scale, const void* dy_planes,
                                      int64_t dy_plane_elems, const float* dy_scale, float* dw, const HnConvDesc* d, int split_k,
                                      int variant, int flags, void* stream) {
    return wgrad_planes_impl(x_planes, x_plane_elems, x_scale, dy_planes, dy_plane_elems, dy_scale, dw, d, split_k, variant, flags,
                             nullptr, stream);
}

extern "C" int hn_conv2d_wgrad_det_planes(const void* x_planes, int64_t x_plane_elems, const float* x_scale,
                                          const void* dy_planes, int64_t dy_plane_elems, const float* dy_scale, float* dw,
                                          const HnConvDesc* d, int split_k, int variant, int flags, float* workspace,
                                          void* stream) {
    if (d == nullptr || workspace == nullptr || split_k <= 0) return HN_EINVAL;
    if (d->N <= 0 || d->Hout <= 0 || d->Wout <= 0) return HN_EINVAL;
    const long long total_chunks = ((long long)d->N * d->Hout * d->Wout + 31) / 32;
    long long s = split_k < total_chunks ? split_k : total_chunks;
    const long long cps = (total_chunks + s - 1) / s;
    s = (total_chunks + cps - 1) / cps;
    const int accumulate = (flags & HN_WGRAD_ACCUMULATE) != 0;
    if (s == 1 && !accumulate)
        return wgrad_planes_impl(x_planes, x_plane_elems, x_scale, dy_planes, dy_plane_elems, dy_scale, dw, d, 1, variant, 0, nullptr,
                                 stream);
    const int rc = wgrad_planes_impl(x_planes, x_plane_elems, x_scale, dy_planes, dy_plane_elems, dy_scale, dw, d, (int)s, variant,
                                     flags, workspace, stream);
    if (rc != HN_OK) return rc;
    const long long n = (long long)d->Cout * d->KH * d->KW * d->Cin;
    const long long blocks = (n + 255) / 256;
    HN_LAUNCH(wgrad_slab_reduce_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
              static_cast<hipStream_t>(stream), workspace, dw, n, (int)s, accumulate);
    HN_CHECK_LAUNCH();
    return HN_OK;
}

// The kernel instantiation a weight-gradient launch runs, as rocprofv3 prints it (per-instantiation accounting in bench.py).
// mode 0: hn_conv2d_wgrad (arithmetic from the variant / library default), 1: hn_conv2d_wgrad_scaled, 2: hn_conv2d_wgrad_planes;
// 16-byte aligned operands assumed.  Mirrors the dispatch in wgrad_impl / wgrad_planes_impl.
extern "C" int hn_conv2d_wgrad_plan(const HnConvDesc* d, int mode, int variant, char* name, int name_len) {
    if (d == nullptr || name == nullptr || name_len <= 0 || mode < 0 || mode > 2 || variant < 0) return HN_EINVAL;
    const int bk_sel = variant & 3, tm_sel = (variant >> 2) & 3, tn_sel = (variant >> 4) & 3, math_sel = variant >> 6;
    if (mode == 2) {
        if (d->Cin % 64 != 0 || d->Cout % 64 != 0) return HN_ESHAPE;
        const bool wm = d->Cout % 128 == 0 && tm_sel != 1, wn = d->Cin % 128 == 0 && tn_sel != 1;
        if (tm_sel == 3 && tn_sel == 3) {
            if (math_sel == 3 || d->Cout % 256 != 0 || d->Cin % 256 != 0) return HN_ESHAPE;
            snprintf(name, name_len, "conv_wgrad4_kernel<4, 2, 2, 2, 2, 4>");
        } else if (math_sel == 3)  // the one-plane bf16 arithmetic
            snprintf(name, name_len, "conv_wgrad4_kernel<%d, %d, %d, 1, 2, 2>", wm ? 2 : 1, wn ? 2 : 1, bk_sel == 2 ? 4 : 3);
        else
            snprintf(name, name_len, "conv_wgrad4_kernel<%d, %d, %d, 2, 2, 2>", wm ? 2 : 1, wn ? 2 : 1, (bk_sel == 2 && !(wm && wn)) ? 3 : 2);
        return HN_OK;
    }
    bool wide_m = d->Cout > 64;
    if (tm_sel) wide_m = tm_sel == 2;
    const long long x_bytes = (((long long)d->N * d->Hin * d->Win - 1) * d->ldx + d->Cin) * 4;
    const long long dy_bytes = (((long long)d->N * d->Hout * d->Wout - 1) * d->ldy + d->Cout) * 4;
    if (d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->Cout % 4 == 0 && d->Cin % 64 == 0) {
        bool wide_n = d->Cin % 128 == 0;
        if (tn_sel == 1) wide_n = false;
        const bool buf_ok = x_bytes < 0xFFFF0000ll && dy_bytes + 64ll * d->ldy * 4 < 0xFFFF0000ll;
        const int math = mode == 1 ? HN_MATH_F16X2 : (math_sel ? math_sel - 1 : hn_get_conv_math());
        if (math != HN_MATH_F32 && buf_ok && d->Cout % 64 == 0) {
            if (d->Cout % 128 != 0) wide_m = false;
            snprintf(name, name_len, "conv_wgrad_x3_kernel<%d, %d, %d>", wide_m ? 2 : 1, wide_n ? 2 : 1,
                     math == HN_MATH_F16X2 ? 2 : (math == HN_MATH_BF16 ? 1 : 3));
            return HN_OK;
        }
        if (math == HN_MATH_F16X2) return HN_ESHAPE;
        snprintf(name, name_len, "conv_wgrad_kernel<%d, %d, %d, %d>", wide_m ? 2 : 1, wide_n ? 2 : 1, buf_ok ? WG_VEC_BUF : WG_VEC,
                 (buf_ok && bk_sel == 2) ? 16 : 32);
        return HN_OK;
    }
    if (mode == 1) return HN_ESHAPE;
    bool wide_n = d->KH * d->KW * d->Cin > 64;
    if (tn_sel) wide_n = tn_sel == 2;
    snprintf(name, name_len, "conv_wgrad_kernel<%d, %d, %d, 32>", wide_m ? 2 : 1, wide_n ? 2 : 1, WG_SCALAR);
    return HN_OK;
}

extern "C" int hn_conv2d_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int split_k, int variant,
                               int flags, void* stream) {
    return wgrad_impl(x, nullptr, dy, nullptr, dw, d, split_k, variant, flags, nullptr, stream);
}

extern "C" int hn_conv2d_wgrad_scaled(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw,
                                      const HnConvDesc* d, int split_k, int variant, int flags, void* stream) {
    if (x_scale == nullptr || dy_scale == nullptr) return HN_EINVAL;
    return wgrad_impl(x, x_scale, dy, dy_scale, dw, d, split_k, variant, flags, nullptr, stream);
}

extern "C" int hn_conv2d_wgrad_det(const float* x, const float* dy, float* dw, const HnConvDesc* d, int split_k, int variant,
                                   int flags, float* workspace, void* stream) {
    return wgrad_det_impl(x, nullptr, dy, nullptr, dw, d, split_k, variant, flags, workspace, stream);
}

extern "C" int hn_conv2d_wgrad_det_scaled(const float* x, const float* x_scale, const float* dy, const float* dy_scale,
                                          float* dw, const HnConvDesc* d, int split_k, int variant, int flags,
                                          float* workspace, void* stream) {
    if (x_scale == nullptr || dy_scale == nullptr) return HN_EINVAL;
    return wgrad_det_impl(x, x_scale, dy, dy_scale, dw, d, split_k, variant, flags, workspace, stream);
}
