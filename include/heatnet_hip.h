/*
 * heatnet_hip.h -- C ABI of libheatnet_hip.so, the MI355X (gfx950) kernels behind the
 * HeatNet training hot path.
 *
 * The reference (jzuern/heatnet-pub) has no native layer: every operator below replaces a
 * `torch.nn` / `torch.nn.functional` call made by the reference's model code, and the
 * citation on each entry point names that call site (paths relative to the reference root,
 * "cm/" = models/confusion_maximization/).  The Python host (heatnet-pub_amd/) binds these
 * symbols with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - All tensors are fp32, caller-owned device memory, NHWC ("channels-last") physical
 *     layout.  A tensor is (pointer, ld): `ld` is the distance in ELEMENTS between two
 *     consecutive pixels (>= channel count), so a channel slice of a wider buffer is a
 *     valid operand and channel concatenation needs no copy.  Pixels are dense:
 *     pixel index = (n*H + h)*W + w.
 *   - Convolution weights are [Cout][KH][KW][Cin] (= an OIHW tensor in channels-last
 *     memory format), dense.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant and allocates
 *     nothing.  The only process-wide state is the two settings hn_set_conv_math / hn_set_deterministic
 *     (defaults for calls that do not choose themselves); they are not synchronised with in-flight
 *     launches -- set them before the first call, one process per GPU.
 *   - Return value: HN_OK or a negative HN_E* code; nothing is thrown across the ABI.
 */
#ifndef HEATNET_HIP_H
#define HEATNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HN_OK 0
#define HN_EINVAL (-1)   /* bad argument (null pointer, non-positive size, bad enum)        */
#define HN_EALIGN (-2)   /* pointer / ld not aligned as the kernel requires                  */
#define HN_ELAUNCH (-3)  /* hipLaunchKernel / hipMemsetAsync reported an error               */
#define HN_ESHAPE (-4)   /* sizes inconsistent with each other (e.g. Hout vs Hin/stride/pad) */

/* activation fused behind an affine/conv epilogue */
#define HN_ACT_NONE 0
#define HN_ACT_RELU 1
#define HN_ACT_PRELU 2   /* one shared slope read from device memory (nn.PReLU())  */
#define HN_ACT_LEAKY 3   /* constant slope (nn.LeakyReLU(0.2))                     */
/* OR-ed into `act` of hn_conv2d_fwd_planes / hn_bn_act_fwd_planes / hn_bn_act_bwd_reduce / hn_bn_act_bwd_apply_planes in the
 * one-plane bf16 arithmetic (scale slots NULL): the pre-normalisation tensor y is stored as bf16 elements (ldy counts
 * elements, C % 4 == 0) -- the convolution still takes the batch statistics from its fp32 accumulators.  Halves the bytes
 * of the tensor every BatchNorm pass reads; not available to the fp32-result arithmetics (HN_EINVAL). */
#define HN_Y_BF16 256

/* Per-channel double accumulators (batch statistics, BatchNorm backward sums) are kept in HN_STAT_REPLICAS
 * interleaved copies so that concurrent workgroups do not serialise on one address; the kernels that consume an
 * accumulator add the replicas up (and zero them again). */
#define HN_STAT_REPLICAS 16

#define HN_LOSS_MSE 0
#define HN_LOSS_BCE 1

typedef struct HnConvDesc {
    int32_t N, Hin, Win, Cin, ldx;    /* input  NHWC, pixel stride ldx  */
    int32_t Hout, Wout, Cout, ldy;    /* output NHWC, pixel stride ldy  */
    int32_t KH, KW, stride, pad, dil; /* square stride/pad/dilation as used by the reference */
} HnConvDesc;

const char* hn_strerror(int code);
int hn_version(void);

/* ---- convolution: nn.Conv2d in cm/models/extractors.py:71-77,111-146, cm/models/pspnet.py:13,18,32,57,
 *      cm/discriminator_model.py:40-44.  Implicit GEMM on v_mfma_f32_32x32x2_f32. ------------------------- */

/* `variant` selects the kernel: 0 = built-in heuristics, else bk + 4*tm + 16*tn + 64*math with tm/tn 1/2 = 64/128-row/
 * -column block tile, math 1/2/3 = fp32 MFMA / bf16x3 / bf16 (see hn_set_conv_math; 4 = f16x2 is set by the *_scaled
 * entry points) and bk = the schedule: fp32 MFMA 1/2 = K step 32/16 (2 or 3 resident workgroups per CU); the plane
 * arithmetics: 0/1 = two workgroups per CU with barrier-separated staging and MFMA phases (csrc/conv_igemm.hip; bf16: as 2),
 * 2 = producer / consumer waves (csrc/conv_igemm2.hip; the home of the bf16 arithmetic), 3 = weight planes by LDS-DMA
 * (csrc/conv_igemm3.hip: a 128 x 128 tile with two workgroups per CU, or with tn = 3 the eight-wave 128 x 256 tile;
 * bf16x3: the latter only); a schedule that does not cover a launch runs the default one.  0 = heuristic / library default
 * for that field.  Within one math mode results are bitwise identical across variants; callers may time them per shape
 * (heatnet-pub_amd/ops.py does).
 *
 * y = act(conv(x, w) + bias).  `bias` may be NULL.  If `stats` is non-NULL it must hold
 * HN_STAT_REPLICAS*2*Cout DOUBLES, zero on entry; the kernel adds per-channel sum and sum of squares of the PRE-activation output
 * (the batch statistics nn.BatchNorm2d needs, cm/models/extractors.py:72,75,77).  Statistics are accumulated
 * in double (global_atomic_add_f64), as the reference's CPU BatchNorm does. */
int hn_conv2d_fwd(const float* x, const float* w, const void* w_x3, const float* bias, float* y, double* stats,
                  const HnConvDesc* d, int act, float slope, int variant, void* stream);

/* Arithmetic of hn_conv2d_fwd / hn_conv2d_dgrad / hn_conv2d_wgrad for variant codes whose math field is 0:
 * HN_MATH_F32 (0, the initial setting): v_mfma_f32_32x32x2_f32 on the fp32 operands.
 * HN_MATH_BF16X3 (1): each fp32 operand is split exactly into three bf16 terms inside the kernel and the six products of
 * weight >= 2^-24 are accumulated in fp32 on v_mfma_f32_16x16x32_bf16 -- fp32-equivalent results (measured against fp64)
 * at 16/6 of the fp32 matrix-core rate.
 * HN_MATH_BF16 (2): REDUCED precision -- operands rounded to nearest bf16 (8 significant bits) as they are staged, one
 * v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation; tensors in memory, BatchNorm statistics, master weights and the
 * optimizer stay fp32 (BASELINE configs[4], "mixed-precision bf16 conv MFMA path").  Never the default.
 * Geometries the bf16 kernels do not cover (Cin % 32 != 0, unaligned tensors, zero-dilated gathers) run on the fp32
 * kernels under either setting.  Process-wide; not synchronised with in-flight launches.
 * HN_MATH_F16X2 (3): fp32 results from two fp16 terms per operand and per-tensor power-of-two scales -- selected per call
 * through the *_scaled entry points (section "f16x2" below), never a process-wide default: it needs the scales. */
#define HN_MATH_F32 0
#define HN_MATH_BF16X3 1
#define HN_MATH_BF16 2
#define HN_MATH_F16X2 3
#define HN_SCALE_WORDS 64
#define HN_SCALE_STRIDE 32 /* 32-bit words between the words of a scale slot: a slot is 64 x 128 B = 8 KB */
#define HN_SCALE_SLOT_FLOATS (HN_SCALE_WORDS * HN_SCALE_STRIDE)
int hn_set_conv_math(int math);
int hn_get_conv_math(void);

/* Deterministic mode: no float atomics in the convolution kernels -- split-K is not used for the forward / data-gradient
 * kernels, and hn_conv2d_wgrad_det writes its reduction splits to a caller-provided workspace and adds them up in a fixed
 * order.  Results are then bitwise reproducible from run to run (the double-precision BatchNorm sums are still collected
 * with global_atomic_add_f64; their order effects are below fp32 resolution).  Off by default. */
int hn_set_deterministic(int on);
int hn_get_deterministic(void);

/* hn_conv2d_fwd with the A operand transformed on the fly: the kernel gathers relu(x * in_scale[c] + in_shift[c]) (zero
 * where a tap falls into the padding) -- the BatchNorm + ReLU of the producing layer applied in this convolution's loader,
 * so the normalised tensor is never written (cm/models/extractors.py:84-94: conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu
 * -> conv3).  in_scale / in_shift: Cin floats each, 16-B aligned.  Only the bf16x3 / bf16 kernels implement it:
 * HN_ESHAPE when the launch would run on another kernel (ask hn_conv2d_plan first). */
int hn_conv2d_fwd_affine(const float* x, const float* in_scale, const float* in_shift, const float* w, const void* w_x3,
                         const float* bias, float* y, double* stats, const HnConvDesc* d, int variant, void* stream);

/* Which kernel hn_conv2d_fwd (kind 0) / hn_conv2d_dgrad (kind 1) would launch for this geometry and variant, assuming
 * 256-B aligned operands and (has_planes != 0) bf16 weight planes: writes the kernel's name as rocprofv3 prints it into
 * `name`; returns 1 if it is a conv_igemm2 kernel (which hn_conv2d_fwd_affine needs), 0 otherwise, < 0 on error.
 * has_planes is a bit set: 1 = weight planes, 2 = the gathered operand is a plane tensor (hn_conv2d_fwd_planes /
 * hn_conv2d_dgrad_planes), 4 = the launch carries a BatchNorm-backward reduction (hn_conv2d_dgrad_planes_bnreduce: the
 * conv_igemm4_kernel instantiation whose last template argument is 1). */
int hn_conv2d_plan(int kind, const HnConvDesc* d, int has_planes, int variant, char* name, int name_len);

/* bf16x3 takes the WEIGHT operand pre-split (the activations / gradients are split inside the kernel): `w_x3` of
 * hn_conv2d_fwd and `wt_x3` of hn_conv2d_dgrad are three consecutive planes (hi, mid, lo) of HN_X3_PLANE(n) bf16 elements
 * each, n = Cout*KH*KW*Cin, every plane in the element order of the fp32 matrix (`w`, resp. `wt`).  hn_split_bf16x3 makes
 * them from any fp32 array; hn_weight_flip_transpose writes them directly.  They may be NULL: the launch then runs on the
 * fp32 MFMA.  The fp32 matrix must always be given (geometries bf16x3 does not cover use it). */
#define HN_X3_PLANE(n) ((((long long)(n)) + 7) & ~7ll)
int hn_split_bf16x3(const float* src, void* dst, int64_t n, void* stream);

/* Operand of hn_conv2d_dgrad for a convolution with this stride / pad / dilation (same element count as w).
 * Stride 1: wt[Cin][KH][KW][Cout] = w[Cout][KH-1-kh][KW-1-kw][Cin].  Stride 2 with odd dilation: four consecutive blocks,
 * one per parity class (ph, pw) of the input pixels, each [Cin][Th*Tw][Cout] with only the taps that reach the class --
 * the data gradient then runs as four small stride-1 convolutions with no arithmetic on inserted zeros.
 * `wt` (fp32) and `wt_x3` (the bf16x3 planes of the same matrix, see above) are each optional, not both NULL.
 * `w_x3` (optional): additionally receives the bf16x3 planes of `w` itself (what hn_split_bf16x3 would produce), so one
 * launch at forward time prepares the weight operands of both the forward and the data-gradient kernels. */
int hn_weight_flip_transpose(const float* w, void* w_x3, float* wt, void* wt_x3, int Cout, int KH, int KW, int Cin, int stride,
                             int pad, int dil, void* stream);

/* The same preparation for MANY convolutions in one launch (a model has ~90; as separate launches they are latency-bound).
 * Fill a host table from the items (hn_weight_prep_table_bytes(n) bytes), copy it to the device once -- it only holds
 * pointers and geometry -- and run hn_weight_prepare_batch whenever the weights have changed.  Outputs may be NULL
 * individually (w_x3: planes of w; wt / wt_x3: the data-gradient operand, fp32 / planes). */
typedef struct HnWeightPrepItem {
    const float* w;
    void* w_x3;
    float* wt;
    void* wt_x3;
    int32_t Cout, KH, KW, Cin, stride, pad, dil;
    int32_t format; /* of w_x3 / wt_x3: 0 = three bf16 planes (bf16x3, bf16), 1 = two fp16 planes + scale (f16x2, below) */
} HnWeightPrepItem;
size_t hn_weight_prep_table_bytes(int n);
int hn_weight_prep_table_fill(const HnWeightPrepItem* items, int n, void* table_host, int* total_blocks);
/* has_f16x2: non-zero when any item has format 1 (adds the pass that finds those layers' max |w|). */
int hn_weight_prepare_batch(const void* table_dev, int n, int total_blocks, int has_f16x2, void* stream);

/* ---- f16x2: fp32 results from TWO fp16 terms per operand -----------------------------------------------------------
 * HN_MATH_F16X2.  Every operand tensor t gets a power-of-two scale s_t = 2^(14 - floor(log2 max|t|)) (its largest magnitude
 * lands in [2^14, 2^15), below fp16's range limit); t * s_t is split into hi = fp16(t * s_t) and lo = fp16(t * s_t - hi) --
 * 22 significant bits for every element within 2^-18 of the tensor's maximum, fewer (gracefully) below that -- and the
 * three products lo*hi, hi*lo, hi*hi are accumulated in fp32 on v_mfma_f32_16x16x32_f16; the result is multiplied by
 * 1 / (s_a * s_b) on the way out (all scalings are exact).  Error against fp64: 0.6-1.05x the fp32 MFMA's for reductions
 * of 64 ... 2304 terms (the products' 2^-23 representation error is below the fp32 accumulation error, and three
 * accumulation steps per 32 k round less than sixteen) at half the matrix-core work of bf16x3 (tests/test_gpu_kernels.py,
 * profiles/r02_f16x2_accuracy.txt).  Heavy-tailed tensors (max / typical magnitude > 2^18) lose bits on their small
 * elements -- bf16x3 has no such range: the caller chooses per layer.
 *
 * Scale slots: HN_SCALE_SLOT_FLOATS floats (8 KB) per tensor, of which every HN_SCALE_STRIDE-th word is used: 64 words on 64
 * different 128-B lines, each the bit pattern of the largest |t| seen by some of the producing blocks (as an unsigned
 * integer it is monotonic in the magnitude; separate lines keep the producers' atomics from serialising).  hn_absmax_scale -- and the passes that take an optional slot (hn_bn_act_fwd,
 * hn_bn_act_bwd_apply) -- ACCUMULATE into it: word = max(word, ...).  Zero the slot before the pass that produces t; the
 * kernels that consume the tensor take the maximum over the words and derive s and 1/s themselves.  A slot may hold a
 * BOUND instead (the slot of a tensor t' with max|t| <= max|t'|: a pooling or resampling of t').
 * Weight planes: [hi plane][lo plane][area] with HN_X3_PLANE(n) fp16 elements per plane -- the buffer has the size of the
 * bf16x3 planes, 3 * HN_X3_PLANE(n) * 2 bytes; the area holds the weight scale and scratch, nothing to initialise.
 * n >= 256 elements.  hn_weight_flip_transpose_h2 / HnWeightPrepItem.format = 1 produce both operands (same scale).
 * The *_scaled calls take the geometry, variant codes (tile fields only) and flags of their plain counterparts and
 * return HN_ESHAPE where the f16x2 kernels do not cover the geometry (Cin % 32 != 0, unaligned, zero-dilated gathers;
 * weight gradient: Cin % 64 != 0 or Cout % 64 != 0) -- the caller then uses another arithmetic for that launch. */
int hn_absmax_scale(const float* x, int ld, int64_t npix, int C, float* scale, void* stream);
int hn_split_f16x2(const float* src, void* dst, int64_t n, void* stream);
int hn_weight_flip_transpose_h2(const float* w, void* w_h2, float* wt, void* wt_h2, int Cout, int KH, int KW, int Cin, int stride,
                                int pad, int dil, void* stream);
int hn_conv2d_fwd_scaled(const float* x, const float* x_scale, const float* w, const void* w_h2, const float* bias, float* y,
                         double* stats, const HnConvDesc* d, int act, float slope, int variant, void* stream);
int hn_conv2d_dgrad_scaled(const float* dy, const float* dy_scale, const float* wt, const void* wt_h2, float* dx,
                           const float* addend, int ldadd, const HnConvDesc* d, int variant, void* stream);
int hn_conv2d_wgrad_scaled(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw,
                           const HnConvDesc* d, int split_k, int variant, int flags, void* stream);
int hn_conv2d_wgrad_det_scaled(const float* x, const float* x_scale, const float* dy, const float* dy_scale, float* dw,
                               const HnConvDesc* d, int split_k, int variant, int flags, float* workspace, void* stream);

/* ---- f16x2 with PLANE TENSORS: the operand split is made once, by the pass that produces the tensor -------------------
 * A plane tensor holds t as two dense fp16 images of `plane_elems` elements each, pixel stride ldp fp16 elements:
 * hi = fp16(t * s) at planes[pix * ldp + c], lo = fp16(t * s - hi) at planes[plane_elems + pix * ldp + c], s the power of two
 * derived from the tensor's scale slot (above) -- the 4 bytes per element of the fp32 tensor, in the form the f16x2
 * kernels otherwise build in registers for every tile that touches the element (nine times per column tile of a 3x3
 * convolution).  t = (hi + lo) / s; elements within 2^-18 of the slot's magnitude keep 22 significant bits.  The slot
 * must be final BEFORE the producing pass runs, so it holds a BOUND of max|t| known beforehand: hn_bn_finalize_range derives
 * one for a BatchNorm output from the per-channel range of the convolution output (the `range_keys` of hn_conv2d_fwd_planes /
 * hn_conv2d_fwd_scaled_range), hn_bn_act_bwd_reduce_range one for the gradient hn_bn_act_bwd_apply_planes writes.
 * The *_planes convolution entry points run csrc/conv_igemm4.hip / conv_wgrad4.hip: both operands global -> LDS by LDS-DMA,
 * no staging registers, no split arithmetic, no ds_write in the K loop; with planes made with the same scale the results
 * are bitwise those of the *_scaled entry points.  They cover Cin % 32 == 0 (weight gradient: Cin % 64 == 0 and
 * Cout % 64 == 0), ldp % 8 == 0, 16-B aligned planes, no split-K; HN_ESHAPE otherwise.
 * cm/models/extractors.py:84-102 (bottleneck: conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3 -> add -> relu). */
int hn_tensor_to_planes(const float* x, int ldx, int64_t npix, int C, const float* scale, void* planes, int64_t plane_elems,
                        int ldp, void* stream);
int hn_planes_to_tensor(const void* planes, int64_t plane_elems, int ldp, const float* scale, float* x, int ldx, int64_t npix,
                        int C, void* stream);
/* d->ldx = pixel stride of the planes (fp16 elements).  `variant`: 0 = heuristic tile, 1 = 128 x 64, 2 = 128 x 256 (eight
 * waves), 3 = 256 x 128 (eight waves); else 128 x 128 with two workgroups per CU.  `range_keys` (optional, needs `stats` and
 * act = none): HN_STAT_REPLICAS * 2 * Cout words, zero on entry: order-preserving keys of the per-channel maximum of y and
 * of -y (see hn_bn_finalize_range, which consumes and re-zeroes them). */
int hn_conv2d_fwd_planes(const void* x_planes, int64_t plane_elems, const float* x_scale, const void* w_h2, const float* bias,
                         float* y, double* stats, uint32_t* range_keys, const HnConvDesc* d, int act, float slope, int variant,
                         void* stream);
/* hn_conv2d_fwd_scaled (fp32 input tensor) that also accumulates the range keys of its output */
int hn_conv2d_fwd_scaled_range(const float* x, const float* x_scale, const float* w, const void* w_h2, const float* bias,
                               float* y, double* stats, uint32_t* range_keys, const HnConvDesc* d, int variant, void* stream);
/* d->ldy = pixel stride of the dY planes */
int hn_conv2d_dgrad_planes(const void* dy_planes, int64_t plane_elems, const float* dy_scale, const void* wt_h2, float* dx,
                           const float* addend, int ldadd, const HnConvDesc* d, int variant, void* stream);
/* hn_conv2d_dgrad_planes whose output dx IS the gradient dz of a BatchNorm (+ ReLU) layer in front of this convolution: the
 * epilogue also takes the FIRST pass of that layer's BatchNorm backward over the tile it holds (what hn_bn_act_bwd_reduce /
 * hn_bn_act_bwd_reduce_range compute from dz and y in a pass of their own), reading the layer's pre-normalisation tensor y;
 * follow with hn_bn_act_bwd_finish.  Dense stride-1 convolutions on plane tensors only (d->ldx == d->Cin; HN_ESHAPE
 * otherwise: use the two stand-alone calls).  act: HN_ACT_NONE or HN_ACT_RELU (mask from relu_bits if given -- the
 * forward's byte masks, Cin % 4 == 0 -- else from y * scale + shift > 0); no Dropout2d mask, no PReLU.
 * cm/models/extractors.py:84-102 (conv -> bn -> relu chains of the bottlenecks). */
typedef struct HnBnReduceArgs {
    const void* y;            /* [pixels of dx][ldy] fp32, or bf16 elements when y_bf16 */
    int32_t ldy, y_bf16, act, reserved;
    const float* scale;       /* [Cin] of this convolution = channels of the BatchNorm layer */
    const float* shift;
    const float* mean;
    const float* invstd;
    const unsigned char* relu_bits;  /* optional */
    double* sums;             /* [HN_STAT_REPLICAS][2 Cin + 1], zero on entry (as for hn_bn_act_bwd_reduce) */
    uint32_t* du_max;         /* optional [HN_STAT_REPLICAS][Cin], zero on entry (as for hn_bn_act_bwd_reduce_range) */
} HnBnReduceArgs;
int hn_conv2d_dgrad_planes_bnreduce(const void* dy_planes, int64_t plane_elems, const float* dy_scale, const void* wt_h2,
                                    float* dx, const float* addend, int ldadd, const HnConvDesc* d, int variant,
                                    const HnBnReduceArgs* bn, void* stream);

/* hn_bn_finalize that also consumes (and re-zeroes) the `range_keys` the producing convolution accumulated, publishes into
 * `z_slot` (zero on entry) the bound of |act(y*scale + shift [+ residual]) * mask| over the tensor -- exact when there is no
 * residual: the extremes of a channel sit at the ends of its range -- and stores max |xhat| per channel in xhat_max [C] (for
 * the backward bound).  res_slot: the residual's scale slot or NULL; mask_max: the largest Dropout2d factor (1 without). */
int hn_bn_finalize_range(double* stats, uint32_t* range_keys, int64_t count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                         float* scale, float* shift, float* xhat_max, int C, int act, const float* slope, float leaky_slope,
                         const float* res_slot, float mask_max, float* z_slot, void* stream);
/* hn_bn_act_fwd writing z as a plane tensor with the scale of z_slot (final before the call: hn_bn_finalize_range); the fp32
 * `z` is optional (NULL: planes only).  The residual may be an fp32 tensor (residual, ldr) or a plane tensor (res_planes,
 * res_plane_elems, ldrp, res_slot), not both. */
int hn_bn_act_fwd_planes(const float* y, int ldy, const float* scale, const float* shift, const float* residual, int ldr,
                         const void* res_planes, int64_t res_plane_elems, int ldrp, const float* res_slot, float* z, int ldz,
                         void* z_planes, int64_t z_plane_elems, int ldzp, const float* z_slot, int64_t npix, int C, int act,
                         const float* slope, float leaky_slope, const float* mask, int64_t hw, unsigned char* relu_bits,
                         void* stream);
/* hn_bn_act_bwd_reduce that also finds max |du| per channel (du_max: HN_STAT_REPLICAS * C words, zero on entry, zero again
 * on return) and publishes into dy_slot (zero on entry) the bound |gamma invstd| (max|du| + |dbeta| / count +
 * max|xhat| |dgamma| / count) of the gradient hn_bn_act_bwd_apply_planes is about to write. */
int hn_bn_act_bwd_reduce_range(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* mean,
                               const float* invstd, const float* scale, const float* shift, int64_t npix, int C, int act,
                               const float* slope, float leaky_slope, double* sums, float* dparams, float* dgamma_out,
                               float* dbeta_out, float* dslope_out, int out_accumulate, const float* mask, int64_t hw,
                               const unsigned char* relu_bits, uint32_t* du_max, const float* gamma, const float* xhat_max,
                               int64_t count, int training, float* dy_slot, void* stream);
/* The closing kernel of hn_bn_act_bwd_reduce / _reduce_range on its own: sums (consumed and re-zeroed) -> dparams and the
 * optional gradient outputs; with du_max != NULL also the bound of |dy| into dy_slot (gamma, xhat_max, count, training as
 * there).  For accumulators filled by hn_conv2d_dgrad_planes_bnreduce. */
int hn_bn_act_bwd_finish(double* sums, float* dparams, int C, float* dgamma_out, float* dbeta_out, float* dslope_out,
                         int out_accumulate, uint32_t* du_max, const float* gamma, const float* invstd, const float* xhat_max,
                         int64_t count, int training, float* dy_slot, void* stream);
/* hn_bn_act_bwd_apply writing dy as a plane tensor with the scale of dy_slot */
int hn_bn_act_bwd_apply_planes(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* mean,
                               const float* invstd, const float* gamma, const float* scale, const float* shift,
                               const float* dparams, int64_t count, int training, void* dy_planes, int64_t dy_plane_elems,
                               int lddp, const float* dy_slot, float* dres, int lddr, int64_t npix, int C, int act,
                               const float* slope, float leaky_slope, const float* mask, int64_t hw,
                               const unsigned char* relu_bits, void* stream);
/* F.upsample(x2, bilinear) (cm/models/pspnet.py:36) written as a plane tensor with the INPUT's scale slot (a bound: the
 * weights are convex). */
int hn_bilinear_up2_planes(const float* x, int ldx, const float* slot, void* y_planes, int64_t plane_elems, int ldp, int N,
                           int Hin, int Win, int C, void* stream);

/* f16x2 range census of one operand: out[0] += number of non-zero elements, out[1] += those of them more than 2^18 below the
 * slot's magnitude (|t * s| < 2^-3: fewer than 22 significant bits survive the hi / lo split).  Give either the fp32 tensor
 * `x` or the plane tensor (`planes`, plane_elems); `ld` is the pixel stride of whichever is given.  Diagnostics
 * (HEATNET_F16X2_GUARD in heatnet-pub_amd/ops.py): layers whose operands exceed a threshold are moved to bf16x3. */
int hn_range_census(const float* x, const void* planes, int64_t plane_elems, int ld, const float* scale, int64_t npix, int C,
                    uint64_t* out, void* stream);

/* Weight gradient with BOTH operands as plane tensors (d->ldx / d->ldy = their pixel strides in fp16 elements); split_k,
 * flags and the deterministic form as for hn_conv2d_wgrad / hn_conv2d_wgrad_det; `variant`: only the tile fields
 * (4 * tm + 16 * tn, 1 = 64-wide, 0 / 2 = 128-wide where the channel counts allow). */
int hn_conv2d_wgrad_planes(const void* x_planes, int64_t x_plane_elems, const float* x_scale, const void* dy_planes,
                           int64_t dy_plane_elems, const float* dy_scale, float* dw, const HnConvDesc* d, int split_k,
                           int variant, int flags, void* stream);
int hn_conv2d_wgrad_det_planes(const void* x_planes, int64_t x_plane_elems, const float* x_scale, const void* dy_planes,
                               int64_t dy_plane_elems, const float* dy_scale, float* dw, const HnConvDesc* d, int split_k,
                               int variant, int flags, float* workspace, void* stream);

/* dx = conv_transpose(dy, w) [+ addend] for the forward geometry `d` (dy is N x Hout x Wout x Cout with pixel stride
 * d->ldy, dx is N x Hin x Win x Cin with pixel stride d->ldx); `wt` from hn_weight_flip_transpose.  `addend`
 * (optional, pixel stride ldadd, must not overlap dx) is a gradient that reached the same tensor through another branch --
 * the residual skip of cm/models/extractors.py:96-99 -- and is added in the epilogue instead of by a separate pass.
 * Overwrites dx. */
int hn_conv2d_dgrad(const float* dy, const float* wt, const void* wt_x3, float* dx, const float* addend, int ldadd,
                    const HnConvDesc* d, int variant, void* stream);

/* dw[Cout][KH][KW][Cin] = sum over pixels of dy (x) x.  Overwrites dw.  split_k > 0: that many reduction splits;
 * 0 / -1: fill two rounds of the chip's 512 resident workgroups; -2 / -3 / -4 / -5: one / three / four / six rounds.
 * `variant`: encoded as for hn_conv2d_fwd (the K-step field selects 32 or 16 pixels per reduction step).
 * `flags`: HN_WGRAD_DW_ZEROED = the caller guarantees dw is all zero on entry (e.g. a slice of a gradient arena cleared
 * once per step), so split launches accumulate into it without a fill of their own; HN_WGRAD_ACCUMULATE = dw += the
 * result (a second use of the same weights in one step adds to the gradient already there). */
#define HN_WGRAD_DW_ZEROED 1
#define HN_WGRAD_ACCUMULATE 2
int hn_conv2d_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int split_k, int variant,
                    int flags, void* stream);

/* Which kernel a weight-gradient launch runs, by its rocprofv3 name (per-instantiation accounting): mode 0 = hn_conv2d_wgrad,
 * 1 = hn_conv2d_wgrad_scaled, 2 = hn_conv2d_wgrad_planes; HN_ESHAPE where that entry point would refuse the geometry. */
int hn_conv2d_wgrad_plan(const HnConvDesc* d, int mode, int variant, char* name, int name_len);

/* Deterministic form (see hn_set_deterministic): the `split_k` (> 0) reduction splits store their partial results to
 * `workspace` (split_k * Cout*KH*KW*Cin floats, caller-owned) with plain stores and a second kernel adds them to dw in a
 * fixed order -- no float atomics, bitwise reproducible.  flags as above (HN_WGRAD_ACCUMULATE: dw += in the same fixed
 * order; HN_WGRAD_DW_ZEROED is not needed). */
int hn_conv2d_wgrad_det(const float* x, const float* dy, float* dw, const HnConvDesc* d, int split_k, int variant, int flags,
                        float* workspace, void* stream);

/* Layers with a tiny channel count on one side (csrc/conv_small.hip): HBM-bound streams on v_mfma_f32_16x16x4_f32 (exact
 * fp32), every global access a full 256-byte pixel row.  Stems: KH = KW = 7, stride 2, pad 3, Cin in {1, 3} (ldx == Cin),
 * Cout = 64, no bias (cm/models/extractors.py:108-126); `stats` as for hn_conv2d_fwd.  Head: 1 x 1, Cin = 64, Cout <= 16
 * (cm/models/pspnet.py:56-58); `w` is the forward weight [Cout][64] (no flip needed), dx is overwritten.  The weight
 * gradients accumulate with float atomics (flags as for hn_conv2d_wgrad); HN_ESHAPE for any other geometry. */
int hn_conv_stem_fwd(const float* x, const float* w, float* y, double* stats, const HnConvDesc* d, void* stream);
int hn_conv_stem_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int flags, void* stream);
int hn_conv_narrow_dgrad(const float* dy, const float* w, float* dx, const HnConvDesc* d, void* stream);
int hn_conv_narrow_wgrad(const float* x, const float* dy, float* dw, const HnConvDesc* d, int flags, void* stream);

/* Grouped 3x3 convolution, stride 1, pad = dil, C / groups in {2, 4, 8, 16} channels per group, C % 16 == 0 (the ResNeXt
 * bottleneck of the reference's "custom" architecture, cm/models/trgb_segnet.py:106-107): direct fp32 kernels
 * (csrc/grouped_conv.hip).  w is [C][3][3][C / groups]; the data gradient takes the FORWARD weights. */
int hn_grouped_conv3x3_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int groups, int ldx, int ldy,
                           int dil, void* stream);
int hn_grouped_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int groups, int lddy, int lddx,
                             int dil, void* stream);
int hn_grouped_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int groups, int ldx, int lddy,
                             int dil, int flags, void* stream);

/* out[c] = sum over npix pixels of x[p*ld + c]  (bias gradient).  `out` is overwritten. */
int hn_channel_sum(const float* x, int ld, int64_t npix, int C, float* out, void* stream);

/* stats[HN_STAT_REPLICAS][{sum[C], sumsq[C]}] (doubles, zero on entry) += over npix pixels: batch statistics of a tensor that did
 * not come out of hn_conv2d_fwd. */
int hn_channel_stats(const float* x, int ld, int64_t npix, int C, double* stats, void* stream);

/* ---- batch norm + activation (+ residual): nn.BatchNorm2d / ReLU / PReLU / `out += residual`,
 *      cm/models/extractors.py:84-100, cm/models/pspnet.py:31-35 ------------------------------------------ */

/* From stats[HN_STAT_REPLICAS][{sum[C], sumsq[C]}] (doubles) over `count` values per channel: mean, invstd = 1/sqrt(var_biased+eps),
 * scale = gamma*invstd, shift = beta - mean*scale; running stats updated in place when non-NULL
 * (momentum 0.1 style, unbiased variance), exactly as nn.BatchNorm2d in training mode.  `stats` is ZEROED after it
 * has been read, so one persistent accumulator per layer needs no per-step fill. */
int hn_bn_finalize(double* stats, int64_t count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift, int C, void* stream);

/* Inference-mode statistics: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale;
 * mean/invstd are filled too so that the backward entry points can be shared. */
int hn_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* mean, float* invstd, float* scale, float* shift, int C, void* stream);

/* z = act(y*scale[c] + shift[c] + residual) [* mask[n*C + c]].  residual may be NULL; `slope` is a device pointer for
 * HN_ACT_PRELU, ignored otherwise (HN_ACT_LEAKY uses leaky_slope).  `mask` (optional, with hw = pixels per image) is a
 * Dropout2d keep/(1-p) factor per (image, channel), cm/models/pspnet.py:55,66-72, applied in the same pass; the two
 * backward entry points take the same mask and fold it into dz.  `z_scale` (optional): a scale slot as hn_absmax_scale
 * takes; the pass then also publishes the f16x2 scale of z, so the convolution that consumes z needs no pass of its own.
 * `relu_bits` (optional; C % 4 == 0 and 16-B aligned rows, else HN_EALIGN): npix * C/4 bytes, one per 16-B piece of z, bit i =
 * (z[c + i] > 0) -- all the two backward passes need of z when the ReLU followed a residual add: they then read 1/16 of
 * the bytes. */
int hn_bn_act_fwd(const float* y, int ldy, const float* scale, const float* shift,
                  const float* residual, int ldr, float* z, int ldz, int64_t npix, int C,
                  int act, const float* slope, float leaky_slope, const float* mask, int64_t hw, float* z_scale,
                  unsigned char* relu_bits, void* stream);

/* First backward pass.  du = dz * act'(u), u = y*scale+shift(+residual).  Accumulates into the double workspace
 * sums[HN_STAT_REPLICAS][2C+1] (must be all zero on entry; it is zero again on return) and writes
 * dparams = {dbeta [C], dgamma [C], dslope [1]} = {sum du, sum du*xhat, ...} in fp32 (xhat = (y-mean)*invstd).
 * HN_ACT_RELU takes its mask from `relu_bits` (the forward's byte mask, see hn_bn_act_fwd) if given, else from z > 0, or
 * -- when both are NULL (allowed only if the forward had no residual) -- from u recomputed from y; PRELU/LEAKY always
 * recompute u from y (residual must have been NULL).
 * dgamma_out / dbeta_out / dslope_out (each optional): the same three results written where the caller keeps the
 * parameters' gradients (added to what is there when out_accumulate != 0), so that no copy or accumulation pass follows. */
int hn_bn_act_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                         const float* mean, const float* invstd, const float* scale, const float* shift,
                         int64_t npix, int C, int act, const float* slope, float leaky_slope,
                         double* sums, float* dparams, float* dgamma_out, float* dbeta_out, float* dslope_out,
                         int out_accumulate, const float* mask, int64_t hw, const unsigned char* relu_bits, void* stream);

/* Second backward pass: dy = gamma*invstd*(du - dbeta/count - xhat*dgamma/count) when `training`,
 * dy = gamma*invstd*du otherwise; dres = du when dres != NULL (gradient of the residual input).
 * `dparams` is the fp32 output of hn_bn_act_bwd_reduce; `z` as there.  `dy_scale` (optional): publishes the f16x2 scale of
 * dy (the gradient operand of the producing convolution's backward kernels), as z_scale above. */
int hn_bn_act_bwd_apply(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                        const float* mean, const float* invstd, const float* gamma, const float* scale,
                        const float* shift, const float* dparams, int64_t count, int training,
                        float* dy, int lddy, float* dres, int lddr, int64_t npix, int C,
                        int act, const float* slope, float leaky_slope, const float* mask, int64_t hw, float* dy_scale,
                        const unsigned char* relu_bits, void* stream);

/* dx = dy * act'(y_out) for an activation applied in a conv epilogue (LeakyReLU / ReLU: the sign of the
 * output equals the sign of the pre-activation).  In place allowed (dx == dy). */
int hn_act_bwd(const float* dy, int lddy, const float* out, int ldo, float* dx, int lddx,
               int64_t npix, int C, int act, float leaky_slope, void* stream);

/* hn_act_bwd that also returns the bias gradient dbias[c] = sum over pixels of dx (overwritten) and, optionally, publishes
 * the f16x2 scale of dx into dx_scale (a zeroed slot): one pass instead of hn_act_bwd + hn_channel_sum + hn_absmax_scale for
 * the critics' conv + LeakyReLU layers (cm/discriminator_model.py:40-58). */
int hn_act_bwd_bias(const float* dy, int lddy, const float* out, int ldo, float* dx, int lddx, int64_t npix, int C, int act,
                    float leaky_slope, float* dbias, float* dx_scale, void* stream);

/* y[n,:,:,c] = x[n,:,:,c] * mask[n*C + c]: nn.Dropout2d forward and backward given a host-drawn
 * keep/(1-p) mask, cm/models/pspnet.py:49,55,63-72. */
int hn_scale_nc(const float* x, int ldx, const float* mask, float* y, int ldy, int N, int64_t hw, int C, void* stream);

/* mask[i] = 0 with probability p, else 1/(1-p), i < n = N*C: the Dropout2d factors hn_bn_act_fwd / hn_scale_nc take, drawn on
 * the device by a counter-based generator (a pure function of seed and i). */
int hn_dropout_mask(float* mask, int n, float p, uint64_t seed, void* stream);

/* Layout conversion at the network boundary: x [N][C][hw] (torch's default contiguous NCHW) -> y [N][hw] pixels of ldy
 * floats (NHWC), what callers hand to the first convolutions (scripts/main.py:238-244 feeds NCHW batches). */
int hn_nchw_to_nhwc(const float* x, float* y, int N, int C, int64_t hw, int ldy, void* stream);

/* y = a + b (elementwise over npix x C): gradient accumulation where two branches meet. */
int hn_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int64_t npix, int C, void* stream);

/* y = x (strided copy, used to place a tensor into a channel slice). */
int hn_copy(const float* x, int ldx, float* y, int ldy, int64_t npix, int C, void* stream);

/* ---- spatial operators ------------------------------------------------------------------------------- */

/* nn.MaxPool2d(3, stride 2, padding 1), cm/models/extractors.py:128.  `idx` (uint8, dense N*Ho*Wo*C)
 * receives the window position kh*3+kw of the first maximum. */
int hn_maxpool3x3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx,
                        int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hn_maxpool3x3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx,
                        int N, int H, int W, int C, int Ho, int Wo, void* stream);

/* nn.AdaptiveAvgPool2d((Ho,Wo)), cm/models/pspnet.py:17: bin i covers [floor(i*H/Ho), ceil((i+1)*H/Ho)). */
int hn_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hn_adaptive_avgpool_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int Ho, int Wo, void* stream);

/* F.upsample(mode='bilinear') == align_corners=False, cm/models/pspnet.py:23,39 and nn.Upsample(32),
 * cm/discriminator_model.py:47,61.  Source coordinate = max(0, (o+0.5)*Hin/Hout - 0.5). */
int hn_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hin, int Win, int C, int Hout, int Wout, void* stream);
int hn_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hin, int Win, int C, int Hout, int Wout, void* stream);

/* out = relu?(main + sum_i bilinear_up(prior_i -> H x W)), up to 8 priors of sizes hs[i] x ws[i] with pixel strides
 * lds[i] (host arrays, read at call time).  The merge step of the pyramid-pooling head (cm/models/pspnet.py:21-25) after
 * the 1x1 bottleneck has been applied per pyramid level at the pooled resolution: a 1x1 convolution commutes with
 * bilinear resampling, so the 5*2048-channel concatenation is never built. */
int hn_pyramid_merge_fwd(const float* main, int ldm, int n_priors, const float* const* priors, const int* hs,
                         const int* ws, const int* lds, float* out, int ldo, int N, int H, int W, int C, int relu,
                         void* stream);

/* Backward of hn_pyramid_merge_fwd (relu = 1) in two passes over ONE read of the gradient: du = dout * [out > 0] is written
 * ([N*H*W][lddu]) and every prior's gradient dps[k] ([N][hs[k]][ws[k]][lds[k]], written) is the bilinear down-gather of du --
 * row sums into column bins first (workspace: N * H * hn_pyramid_merge_bwd_segments(N, H, W) * sum(ws) * C floats, contents
 * irrelevant on entry), then the weighted column of row sums.  Gather form, fixed order: bitwise reproducible.  Priors are
 * upsampled (hs[k] <= H, ws[k] <= W), at most 8. */
int hn_pyramid_merge_bwd_segments(int N, int H, int W);
int hn_pyramid_merge_bwd(const float* dout, int lddo, const float* out, int ldo, float* du, int lddu, int n_priors,
                         float* const* dps, const int* hs, const int* ws, const int* lds, float* workspace, int N, int H, int W,
                         int C, void* stream);

/* The adaptive average poolings of all pyramid levels (nn.AdaptiveAvgPool2d((hos[k], wos[k])), cm/models/pspnet.py:17-18) from ONE
 * read of x: ys[k] ([N][hos[k]][wos[k]][lds[k]], written) = the mean over PyTorch's bins (start = floor(i * H / ho), end =
 * ceil((i + 1) * H / ho)); row sums into column bins first (workspace as for hn_pyramid_merge_bwd: N * H *
 * hn_pyramid_merge_bwd_segments(N, H, W) * sum(wos) * C floats), then the rows of a bin.  hos[k] <= H, wos[k] <= W, at most 8. */
int hn_pyramid_pool_fwd(const float* x, int ldx, int n_levels, float* const* ys, const int* hos, const int* wos, const int* lds,
                        float* workspace, int N, int H, int W, int C, void* stream);

/* dx[n,h,w,c] = sum over levels l of adaptive_avg_pool2d backward of dys[l] ([N, hos[l], wos[l], C], row stride lds[l])
 * at input size H x W: the gradient of the four pyramid poolings of PSPModule (cm/models/pspnet.py:22-24,30) written in
 * one pass.  n_levels <= 8.  dx is overwritten. */
int hn_pyramid_pool_bwd(int n_levels, const float* const* dys, const int* hos, const int* wos, const int* lds,
                        float* dx, int lddx, int N, int H, int W, int C, void* stream);

/* ---- losses ------------------------------------------------------------------------------------------ */

/* nn.CrossEntropyLoss (mean over non-ignored pixels), cm/train_trgb_segnet_conf.py:244,452 and
 * scripts/main.py:223.  out = {sum of -log softmax[label], number of counted pixels} (overwritten). */
int hn_ce_fwd(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int64_t ignore_index,
              float* out, void* stream);
/* dlogits = (*gout / out[1]) * (softmax - onehot), 0 on ignored pixels. */
int hn_ce_bwd(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int64_t ignore_index,
              const float* out, const float* gout, float* dlogits, int lddl, void* stream);

/* Certainty-weighted night supervision (`--weight_ir_sup`, cm/train_trgb_segnet_conf.py:468-477):
 * nn.CrossEntropyLoss(reduction='none')(logits, labels) * weight, then torch.mean over the whole map.
 * out = {sum of weight * -log softmax[label], npix}; dlogits = (*gout / npix) * weight * (softmax - onehot). */
int hn_ce_weighted_fwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t npix, int C,
                       float* out, void* stream);
int hn_ce_weighted_bwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t npix, int C,
                       const float* out, const float* gout, float* dlogits, int lddl, void* stream);

/* Outputs of the frozen night-supervision teacher (:430-434, 468-472): label[p] = argmax softmax(logits[p]) (int64) and
 * cert[p] = max softmax(softmax(logits[p])) -- the reference applies softmax twice, restated as is.  Either output may be
 * NULL. */
int hn_teacher_cert(const float* logits, int ld, int64_t npix, int C, int64_t* label, float* cert, void* stream);

/* nn.MSELoss / nn.BCEWithLogitsLoss against a constant map, mean-reduced,
 * cm/train_trgb_segnet_conf.py:237-240,437-446,536-540.  out[0] = mean loss (overwritten). */
int hn_adv_loss_fwd(const float* x, int64_t n, float target, int kind, float* out, void* stream);
/* dx = (*gout / n) * dloss/dx */
int hn_adv_loss_bwd(const float* x, int64_t n, float target, int kind, const float* gout, float* dx, void* stream);

/* The certainty branch of the custom architecture (cm/models/trgb_segnet.py:344-346, 383-387): y = sigmoid(x) over n floats and
 * its gradient dx = dy * y * (1 - y). */
int hn_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int hn_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);

/* The trainer's certainty loss AS WRITTEN (cm/train_trgb_segnet_conf.py:482-491): the target is t = 1 - softmax(logits)[label]
 * ([N, H, W], NOT detached), the prediction `cert` is the sigmoid map [N, 1, H, W]; `cert - t` broadcasts to [N, N, H, W] and
 * the loss is 10 * mean of its square.  out[0] (zero on entry) += sum over pixels, predictions n and targets m of
 * (cert[n, p] - t[m, p])^2 -- the caller scales by 10 / (N N HW).  hw = H * W; logits [N * hw][ld], labels [N * hw].
 * Backward: gout[0] = gradient of the (scaled) loss; dcert [N * hw] (may be NULL) and dlogits [N * hw][lddl] are WRITTEN. */
int hn_cert_loss_fwd(const float* logits, int ld, const int64_t* labels, const float* cert, int N, int64_t hw, int C, float* out,
                     void* stream);
int hn_cert_loss_bwd(const float* logits, int ld, const int64_t* labels, const float* cert, int N, int64_t hw, int C,
                     const float* gout, float* dcert, float* dlogits, int lddl, void* stream);

/* argmax over C + confusion-matrix histogram: conf[label*K + pred] += 1 (int64, caller-zeroed), labels
 * outside [0,K) skipped.  scripts/iou_eval.py:19-98, cm/utils.py:134-163. */
int hn_argmax_confusion(const float* logits, int ld, const int64_t* labels, int64_t npix, int C, int K,
                        int64_t* conf, uint8_t* pred_out, void* stream);

/* ---- optimizers: torch.optim.RMSprop / Adam as configured by cm/train_trgb_segnet_conf.py:270 and
 *      scripts/main.py:159 (elementwise over a flat parameter range) ------------------------------------ */
int hn_rmsprop_step(float* param, const float* grad, float* square_avg, int64_t n,
                    float lr, float alpha, float eps, float weight_decay, void* stream);
int hn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* ---- data path: per-sample augmentation of cm/thermal_loader.py:642-735 on the device ----
 * One entry of `params` (device memory, N of them) describes what the loader draws for a sample:
 *   crop_i, crop_j   top-left corner of the random crop (transforms.RandomCrop.get_params, :642)
 *   flip             horizontal flip (F.hflip, :690-699)
 *   mode             0 = no rotation, 1 = rotation by the affine matrix below, 2 = Pillow's exact 180-degree fast path
 *   m[6]             PIL.Image.rotate's inverse matrix (a, b, c, d, e, f), double
 *   fx[6]            the same matrix in Pillow's 16.16 fixed point (nearest-neighbour path of the labels):
 *                    FLOOR(v * 65536 + 0.5) of a, b, c + a/2 + b/2, d, e, f + d/2 + e/2
 *   drop_*           modality block drop rectangle (train_trgb_segnet_conf.py:82-86); drop_target 0 none, 1 RGB, 2 IR
 *   ir_scale         factor applied to the normalised thermal tensor (`ir_day = scale * ir_day`, :402-405); 1 = off
 * heatnet-pub_amd/augment.py fills it exactly as oracle/thermal_aug_ref.py does. */
typedef struct HnAugSample {
    int32_t crop_i, crop_j, flip, mode;
    double m[6];
    long long fx[6];
    int32_t drop_i, drop_j, drop_h, drop_w, drop_target;
    float ir_scale;
} HnAugSample;

/* rgb [N][H0][W0][3] uint8, ir [N][H0][W0] uint16 (raw thermal counts), label [N][H0][W0] uint8 or NULL ->
 * rgb_out [N][h][w][3], ir_out [N][h][w] fp32 in [-1, 1] (NHWC), label_out [N][h][w] int64 (NULL iff label is NULL).
 * class_scale (optional, HN_AUG_CLASSES floats, needs labels): per-class factor on the thermal tensor where the
 * (augmented) label equals the class -- smartAugment, train_trgb_segnet_conf.py:101-110. */
#define HN_AUG_CLASSES 32
int hn_thermal_augment(const uint8_t* rgb, const uint16_t* ir, const uint8_t* label, int N, int H0, int W0,
                       const HnAugSample* params, const float* class_scale, int h, int w, float* rgb_out, float* ir_out,
                       int64_t* label_out, void* stream);

/* Decoded sample -> network-resolution arrays: cv2.resize(.., (dw, dh), INTER_LINEAR) of the uint8 RGB image [N][H0][W0][3]
 * and the uint16 thermal image [N][H0][W0], cv2.resize(INTER_NEAREST) of the uint8 label map, then the loaders' column crop
 * [:, col0:col1] (cm/thermal_loader.py:621-640: (960, 320) and 150:850; :783-791: 148:852).  Outputs are
 * [N][dh][col1-col0](x3) in the input types -- what hn_thermal_augment takes.  Any of the three images may be NULL.
 * OpenCV's arithmetic is restated (uint8: 11-bit fixed point; uint16: float32; exact 2x2 shrink: area average); parity
 * against cv2 itself is unpinned (cv2 is not available to the build), bit-exact against oracle/thermal_resize_ref.py. */
int hn_thermal_resize(const uint8_t* rgb, const uint16_t* ir, const uint8_t* label, int N, int H0, int W0, int dh, int dw,
                      int col0, int col1, uint8_t* rgb_out, uint16_t* ir_out, uint8_t* label_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HEATNET_HIP_H */
