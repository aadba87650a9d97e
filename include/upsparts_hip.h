/* upsparts_hip.h -- C ABI of libupsparts_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the part-discovery training hot path of
 * CompVis/unsupervised-part-segmentation.  The reference has NO FFI of its own
 * (it is TensorFlow-1.14 graph code); every entry point below replaces the stock
 * TF op(s) that the cited reference lines execute.  Citations are relative to
 * /root/reference:  M = cub/code/SB_model48i/model.py,  N = cub/code/nn.py.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     (caller-owned, no hidden allocation) unless a parameter says "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = ok, negative = UPS_E_* (no exceptions cross the ABI);
 *   - activations are NHWC, element type UPS_F32, UPS_BF16 or (forward tensors only) UPS_F16, with a physical
 *     channel count that is a multiple of 8 (16-byte rows); accumulation is fp32;
 *   - thread-compatible: one stream per concurrent caller.
 */
#ifndef UPSPARTS_HIP_H
#define UPSPARTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ups_conv_desc grew by out_act / res_act, ups_wgrad_desc by in_f16, UPS_F16 was added (round 3) -- a stale library
 * built against the older structs ignores those fields silently, so the loader also compares ups_struct_sizes().
 * 3: ups_wgrad_desc grew by dout_f8 / dout_f8_scale / in_f8_scale / in_f8_amax (the fp8 weight gradient, round 5). */
#define UPS_ABI_VERSION 4

/* UPS_F16 (IEEE half): element type of FORWARD tensors of precision-critical scopes (the mask decoder): ups_conv_igemm,
 * ups_weight_prep(_batch), ups_bilinear2x_fwd, ups_convert / ups_pad_convert accept it; gradients are never fp16 (range): the
 * input-gradient call of such a layer is a plain UPS_BF16 call (`dact` only has its sign read, which is the same bit in both
 * 16-bit formats), its weight-gradient call sets ups_wgrad_desc.in_f16. */
enum { UPS_F32 = 0, UPS_BF16 = 1, UPS_F16 = 2 };
/* UPS_ACT_ELU (tf.nn.elu, N:757-758; no shipped yaml uses it): accepted by the pointwise entry points only (ups_elu_fwd / _bwd,
 * ups_act_mean_*); the convolution descriptors take NONE / LRELU / RELU -- a scope with `activation: elu` materialises act(x). */
enum { UPS_ACT_NONE = 0, UPS_ACT_LRELU = 1, UPS_ACT_RELU = 2, UPS_ACT_ELU = 3 };
enum { UPS_OK = 0, UPS_E_ARG = -1, UPS_E_UNSUPPORTED = -2, UPS_E_LAUNCH = -3 };

int ups_abi_version(void);
/* sizeof of the descriptor structs AS THE LIBRARY WAS COMPILED: out[0] ups_conv_desc, [1] ups_wgrad_desc, [2] ups_prior_desc,
 * [3] ups_prep_item.  A binding compares them with its own struct sizes at load time. */
void ups_struct_sizes(int64_t out[4]);
/* human readable description of the last error on this thread (host string) */
const char* ups_last_error(void);

/* ---------------------------------------------------------------- convolution engine
 * Implicit-GEMM gather convolution on MFMA (bf16 32x32x16 / exact-f32 32x32x2).
 * One kernel serves tf.nn.conv2d forward (N:617-664, nin N:811-813, downsample
 * N:816-817) and its input gradient (dgrad), for stride 1 and 2 with TF 'SAME'
 * padding, by describing the op as
 *     out[img, i*out_sy+out_oy, j*out_sx+out_ox, c] =
 *         epi( sum_t sum_k in[img, i*in_sy+tap_dy[t], j*in_sx+tap_dx[t], k] * w[tap_w[t]][c][k] )
 * over the lattice i<ho, j<wo.  Out-of-range source pixels read as zero.
 * epi:  v = acc + bias[c] + coord_affine(c) ; v *= act'(dact[pix][c]) ; v += res[pix][c].
 */
typedef struct {
    int32_t dtype;            /* UPS_F32 / UPS_BF16: element type of in, w, res, dact and (unless out_f32) out */
    int32_t n, hi, wi;        /* input tensor [n, hi, wi, ldi] */
    int32_t ci;               /* reduction channels per tap (multiple of 8, <= ldi) */
    int32_t ldi;              /* physical input channels */
    int32_t ho, wo;           /* lattice points per image */
    int32_t co;               /* logical output channels = rows of each weight slice */
    int32_t co_fill;          /* channels [co, co_fill) of out are written as zero (>= co) */
    int32_t ldo;              /* physical channels of out */
    int32_t out_h, out_w;     /* physical output spatial size */
    int32_t out_sy, out_sx, out_oy, out_ox;
    int32_t in_sy, in_sx;
    int32_t ntaps, kh, kw;    /* taps are r-major (t = r*kw + s) when coord_tab is used */
    int32_t tap_dy[9], tap_dx[9], tap_w[9];
    int32_t act_in;           /* UPS_ACT_* applied to `in` while it is staged (fused lrelu/relu-on-load) */
    float   act_slope;        /* leaky slope (0.2: N:755-756) */
    int32_t out_f32;          /* 1: out is float regardless of dtype */
    int32_t dact_kind;        /* UPS_ACT_* of the activation whose derivative multiplies the result (dgrad) */
    int32_t ldr, ldd;         /* physical channels of res / dact */
    const void*  in;
    const void*  w;           /* blocked-K weights [n_slices][ceil(ci/BK)][co][BK], BK = 64 bytes / sizeof(dtype)
                                 (32 bf16 / 16 f32), zero padded in K: produced by ups_weight_prep */
    void*        out;
    const float* bias;        /* [co] or NULL */
    const float* coord_tab;   /* [64][3][co] affine CoordConv table (ups_coord_table) or NULL */
    const void*  res;         /* residual, same pixel lattice as out, or NULL */
    const void*  dact;        /* pre-activation tensor for dact_kind, same lattice as out, or NULL */
    float*       workspace;   /* optional fp32 scratch for the deterministic split-K path of small-M problems (few tiles,
                               * long K loops); NULL or too small = no split */
    size_t       workspace_bytes;
    /* Part-masked input (mask_parts + apply_partwise, M:176-187, N:81-113) fused into the load, so that the
     * [P*B,H,W,C] part tensor never exists (3x3 / stride-1 patch kernels, bf16, 16-aligned images; else UPS_E_UNSUPPORTED):
     * `n` = P*mask_batch logical images in part-major order (image p*mask_batch + b); `in` is the UNMASKED view tensor
     * [mask_batch, hi, wi, ldi]; pixel (b,y,x) of part image p reads as in[b,y,x,:] if bit p of mask_bits[b,y,x] is set, else 0. */
    const uint32_t* mask_bits;   /* [mask_batch, hi, wi] hard-mask bit sets (ups_part_softmax_fwd) or NULL */
    int32_t      mask_batch;
    /* Input gradient of such a convolution reduced straight to the hard mask (the dgrad call of the same layer): instead
     * of out[p*B+b][y][x][c] the kernel writes mask_grad[b][y][x][p] = sum_c out * mask_view[b][y][x][c] (M:185 backward). */
    float*       mask_grad;      /* [mask_batch, out_h, out_w, n / mask_batch] fp32 or NULL (`out` may then be NULL) */
    const float* mask_view;      /* [mask_batch, out_h, out_w, co] fp32 view tensor, required with mask_grad */
    /* fp8 forward (BASELINE config #5: e4m3 MFMA operands, fp32 accumulate, bf16 tensors; 3x3 / stride-1 patch kernels,
     * dtype UPS_BF16, 16-aligned images, ci % 64 == 0, no mask; else UPS_E_UNSUPPORTED).  f8_deq != NULL selects it:
     * `w` then holds e4m3 weights [9][ci/64][co][64] scaled per output channel (ups_weight_prep_f8), f8_deq[c] = 1 / that
     * scale; the staged activations are multiplied by *f8_scale before the conversion (device scalar, delayed scaling) and
     * max |act(in)| of this launch is collected into the 64 slots of f8_amax (atomic max of non-negative float bits). */
    const float* f8_deq;         /* [co] or NULL */
    const float* f8_scale;       /* device scalar */
    float*       f8_amax;        /* [64] */
    int32_t      f8_e5m2;        /* 1: `in` is a gradient -- staged as e5m2 (the input-gradient call of an fp8 layer; `w` from
                                  * ups_weight_prep_f8 with transpose = 1), 0: e4m3 */
    /* fp8 copies handed from layer to layer.  Producer side (any bf16 patch-kernel call with the staged epilogue): out_f8_amax != NULL
     * records max |act(out)| (act = out_f8_act, the consumer's activation-on-load) into its 64 slots; with out_f8 != NULL the epilogue also
     * writes e4m3 (out_f8_e5m2: e5m2) of act(out) * *out_f8_scale to out_f8 [n, out_h, out_w, ldo] bytes, next to the bf16 tensor.
     * Consumer side: in_f8 != NULL (with f8_deq) = that byte tensor [n, hi, wi, ldi]; it is staged as it is (f8_scale = the scale it was
     * written with, act_in is not applied again, f8_amax unused) with the bf16 kernel's register budget (two blocks per CU). */
    const void*  in_f8;
    void*        out_f8;
    const float* out_f8_scale;
    float*       out_f8_amax;
    int32_t      out_f8_act, out_f8_e5m2;
    /* Depth-to-space output (input gradient of a 3x3 / stride-2 convolution as ONE stride-1 convolution over the gradient
     * lattice, N:811-817 backward): d2s = C > 0 (power of two, >= 8) declares co = 4 C GEMM channels ordered (py, px, c); channel c of
     * class (py, px) at lattice pixel (y, x) is written to out[n][2y + py][2x + px][c] of a [n, out_h = 2 ho, out_w = 2 wo, ldo]
     * tensor; res / dact are read on that lattice.  `w` = ups_weight_prep_d2s.  bf16 patch kernel only (else UPS_E_UNSUPPORTED). */
    int32_t      d2s;
    /* Post-activation storage.  A 16-bit (or fp32) tensor whose only convolution consumer applies an activation on load may be
     * STORED as act(x) instead of x: the consumer then stages it untouched (act_in = UPS_ACT_NONE: the halo patch can arrive by
     * LDS-DMA, no staging registers, no VALU), its weight gradient reads the operand as it is, its input gradient still takes
     * the sign for act' from it (sign(act(x)) = sign(x)).
     *   out_act  activation applied to the finished value (after bias, CoordConv, act', residual) before it is stored;
     *   res_act  `res` holds act(x) of a leaky-ReLU (slope > 0): it is inverted (r > 0 ? r : r / slope) before it is added --
     *            the residual stream x + conv(act(x)) of N:1042-1056 with x stored as act(x).  UPS_ACT_LRELU or UPS_ACT_NONE.
     * Gradients are always with respect to the pre-activation values: nothing changes in the backward calls. */
    int32_t      out_act, res_act;
    /* Bit-packed activation signs (ABI 4).  An input-gradient launch needs ONE bit of every element of the layer's forward input --
     * the sign, for act' -- and used to re-read the whole 16-bit tensor for it (`dact`).
     *   sign_out   forward call, 16-bit `out` [n, out_h, out_w, ldo] with ldo % 8 == 0, plain (unstrided, no d2s) output: also
     *              write bits[n][out_h][out_w][ldo / 8] bytes, bit e of byte j = (out[..., 8 j + e] > 0) of the STORED value
     *              (with out_act: of act(value), which has the value's sign).  BEST EFFORT: written by the kernels that have the
     *              output tile in hand (the 16-bit patch kernel); a launch that went to another kernel leaves the buffer untouched
     *              (a pass over `out` would cost the read the bits save) -- ups_conv_sign_out_written() tells, ups_sign_pack packs.
     *   dact_bits  backward call with `dact`: the same bytes for the tensor `dact` points to ([n, out_h, out_w, ldd / 8]); kernels
     *              that can (the 16-bit patch kernel) read them INSTEAD of `dact`, the others ignore them.  `dact` stays mandatory. */
    void*        sign_out;
    const void*  dact_bits;
} ups_conv_desc;

int ups_conv_igemm(const ups_conv_desc* d, void* stream);
/* 1 when the calling thread's last ups_conv_igemm call wrote its ups_conv_desc.sign_out buffer, else 0 */
int ups_conv_sign_out_written(void);

/* Weights of the depth-to-space input gradient of a 3x3 / stride-2 'SAME' convolution with forward variable V [3][3][cin_v][co]:
 * w[t9][k][(py*2+px)*C + c][32] (blocked-K over the co gradient channels, bf16), t9 = (dy+1)*3 + (dx+1) the 3x3 neighbourhood of
 * the gradient lattice, C = ci_log rounded up to a power of two (>= 8), = V[r][s][c][k-chunk] where tap (r, s) reaches class (py, px)
 * from lattice offset (dy, dx) -- r = py + pad_y - 2 dy, s likewise -- and zero elsewhere. */
int ups_weight_prep_d2s(const float* src, int32_t cin_v, int32_t ci_log, int32_t co, int32_t pad_y, int32_t pad_x, int32_t C,
                        void* w, void* stream);

/* e4m3 weights for the fp8 forward: w_f8[tap][k][c][64] = e4m3(V[tap][64 k + j][c] * 448 / amax_c), zero padded in K,
 * deq[c] = amax_c / 448 with amax_c = max |V[:, :ci_log, c]| (the CoordConv rows ci_log.. stay fp32 in ups_coord_table).
 * transpose = 1: the input-gradient operand -- rows = input channels, K = output channels: w_f8[tap][k][ci][64] =
 * e4m3(V[tap][ci][64 k + j] * 448 / amax_ci), deq[ci] = amax_ci / 448.
 * src is the HWIO fp32 variable [ntaps][cin_v][co] (N:644-652). */
int ups_weight_prep_f8(const float* src, int32_t ntaps, int32_t cin_v, int32_t ci_log, int32_t co, int32_t transpose,
                       void* w_f8, float* deq, void* stream);

/* Weight gradient  dV[tap][ci][co] = sum_pix act(in)[src(pix,tap)][ci] * dout[pix][co]
 * (gradient of N:661-663 w.r.t. V), split-K over pixels into fp32 slabs + deterministic reduce.
 * grad layout is the TF variable layout HWIO with `cin_v` input channels (cin_v = ci_log (+2 CoordConv)). */
typedef struct {
    int32_t dtype;
    int32_t n, hi, wi, ci, ldi;       /* forward input, ci = channels to differentiate (multiple of 8) */
    int32_t ci_log;                   /* rows actually stored (logical input channels) */
    int32_t cin_v;                    /* input-channel extent of grad (>= ci_log) */
    int32_t ho, wo, co, ldo;          /* dout tensor [n, ho, wo, ldo], co logical */
    int32_t in_sy, in_sx;
    int32_t ntaps;
    int32_t tap_dy[9], tap_dx[9], tap_w[9];
    int32_t act_in; float act_slope;
    int32_t splitk;                   /* from ups_conv_wgrad_plan */
    const void* in; const void* dout;
    float* grad;                      /* [kh*kw][cin_v][co] fp32 */
    float* grad_bias;                 /* [co] fp32 or NULL: sum_pix dout, reduced from the same dout tiles */
    float* workspace;                 /* bytes from ups_conv_wgrad_plan */
    const uint32_t* mask_bits;        /* part-masked input, same meaning as in the conv descriptor; bf16 3x3 / stride-1 patch kernel only; or NULL */
    int32_t mask_batch;
    int32_t in_f16;                   /* dtype UPS_BF16 only: `in` holds fp16 (the forward tensor of a UPS_F16 layer); it is converted to
                                       * bf16 (after the activation) while it is staged -- dout stays bf16 */
    /* fp8 weight gradient (ABI 3; BASELINE config #5): dout_f8 != NULL selects it for the wide 3x3 / stride-1 layers (ci % 64 == 0,
     * co % 128 == 0, 16-aligned images, forward tap order): e5m2(dout * *dout_f8_scale) as its producer wrote it, [n, ho, wo, ldo]
     * bytes -- the copy the layer's input-gradient launch reads (ups_conv_desc.in_f8) --, `in` quantised to e4m3 with
     * *in_f8_scale while it is staged (max |act(in)| recorded into the 64 slots of in_f8_amax for the next step's scale, or NULL),
     * block-scaled K = 128 MFMA, fp32 accumulation; other shapes ignore these fields and run the bf16 kernels on `dout`. */
    const void*  dout_f8;
    const float* dout_f8_scale;
    const float* in_f8_scale;
    float*       in_f8_amax;
} ups_wgrad_desc;

int ups_conv_wgrad_plan(const ups_wgrad_desc* d, int32_t* splitk, size_t* workspace_bytes);
int ups_conv_wgrad(const ups_wgrad_desc* d, void* stream);

/* fp32 HWIO master weights -> dtype copies in the blocked-K layout [tap][k-chunk][row][BK] (BK = 64 bytes):
 * w_fwd rows = co, K = ci_pad; w_dgrad rows = dgrad_rows, K = dgrad_k.
 * src [ntaps][cin_v][co]; only input channels < ci_log are copied (CoordConv rows are handled by
 * ups_coord_table); rows/cols beyond the logical extent are zero.  Either dst may be NULL. */
int ups_weight_prep(const float* src, int32_t ntaps, int32_t cin_v, int32_t ci_log, int32_t co,
                    int32_t dtype, void* w_fwd, int32_t ci_pad,
                    void* w_dgrad, int32_t dgrad_rows, int32_t dgrad_k, void* stream);

/* Batched form: one launch converts every convolution of a sub-network after its optimizer step (weights in both
 * layouts + the CoordConv table).  `items` is a DEVICE array of n_items descriptors, `block_prefix` a DEVICE array of
 * n_items+1 cumulative 256-thread block counts (ups_prep_item_blocks gives the per-item count). */
typedef struct {
    const float* src;         /* fp32 HWIO variable V [ntaps][cin_v][co] */
    void*  w_fwd;             /* [ntaps][ceil(ci_pad/BK)][co][BK] or NULL */
    void*  w_dgrad;           /* [ntaps][ceil(dgrad_k/BK)][dgrad_rows][BK] or NULL */
    float* ctab;              /* [64][3][co] or NULL (no CoordConv) */
    int32_t ntaps, cin_v, ci_log, co, ci_pad, dgrad_rows, dgrad_k;
    int32_t kh, kw, in_sy, in_sx;
    int32_t dy[3], dx[3];
    float ax, ay;
} ups_prep_item;
int64_t ups_prep_item_blocks(const ups_prep_item* item_host, int32_t dtype);
int ups_weight_prep_batch(const ups_prep_item* items, const int64_t* block_prefix, int32_t n_items, int64_t total_blocks,
                          int32_t dtype, void* stream);

/* CoordConv (N:2123-2154) folded into an affine epilogue: tab[cls][0..2][c] with
 * cls = ymask*8 + xmask (valid-tap bitmasks of the output pixel);
 * contribution = tab[cls][0][c] + j*tab[cls][1][c] + i*tab[cls][2][c].
 * V is the fp32 HWIO variable with ci_log+2 input channels; ax = 2/max(1,H-1), ay = 2/max(1,W-1). */
int ups_coord_table(const float* V, int32_t kh, int32_t kw, int32_t ci_log, int32_t co,
                    const int32_t* tap_dy, const int32_t* tap_dx, int32_t in_sy, int32_t in_sx,
                    float ax, float ay, float* tab, void* stream);
/* gsum[pix][c] = sum_n dout[n][pix][c] (fp32) */
int ups_batch_sum(const void* dout, int32_t dtype, int32_t n, int64_t pix, int32_t co, int32_t ldo,
                  float* gsum, void* stream);
/* gradient of the two CoordConv rows of V (and optionally the bias) from gsum [ho*wo][co];
 * `scratch` holds (kh+1)*2*wo*co floats (separable two-stage reduction: rows first, then columns) */
int ups_coord_wgrad(const float* gsum, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t co,
                    int32_t kh, int32_t kw, const int32_t* tap_dy, const int32_t* tap_dx,
                    int32_t in_sy, int32_t in_sx, float ax, float ay,
                    int32_t ci_log, float* gradV, float* grad_bias, float* scratch, void* stream);
/* grad_bias[c] = sum_rows dout[row][c]; workspace >= 1024*co floats */
int ups_col_sum(const void* dout, int32_t dtype, int64_t rows, int32_t co, int32_t ldo,
                float* out, float* workspace, void* stream);

/* ---------------------------------------------------------------- resampling / pooling
 * Legacy TF-1 bilinear x2 (N:834-847, tf.image.resize_images BILINEAR, no half-pixel centres). */
int ups_bilinear2x_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int ups_bilinear2x_bwd(const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
/* The other up-sampling methods of nn.upsample (N:820-849).  "subpixel" = a convolution to 4 C channels (ups_conv_igemm) followed by
 * tf.depth_to_space(x, 2): y[b, 2i+di, 2j+dj, c] = x[b, i, j, (2 di + dj) C + c]; x is [n,h,w,ldx] (4 C logical channels), y is
 * [n,2h,2w,ldy] (C logical channels, pad channels written as zero).  bwd != 0: src = the gradient w.r.t. y, dst = w.r.t. x.
 * "nearest_neighbor": y[b, 2i+di, 2j+dj, :] = x[b, i, j, :]; bwd != 0: src = gradient [n,2h,2w,c], dst [n,h,w,c] = the sum of the four. */
int ups_depth_to_space(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t C, int32_t ldx, int32_t ldy,
                       int32_t bwd, void* stream);
int ups_nearest2x(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bwd, void* stream);
/* The ho x wo window of x [n,h,w,c] whose corner (oy, ox) = yx_dev[0..1] is read ON THE DEVICE (int32; clamped to the image):
 * the random 224x224 window of edflow VGG19Features(original_scale=True).make_loss_op as `perceptual_input: resize256_crop224`
 * reads it (M:610-618; UNVERIFIED), one window per step for the whole batch.  bwd: gx [n,h,w,c] = gy inside the window, 0
 * elsewhere.  16-bit dtypes are copied as raw bits (UPS_BF16 and UPS_F16 alike). */
int ups_crop_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo,
                 const int32_t* yx_dev, void* stream);
int ups_crop_bwd(const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo,
                 const int32_t* yx_dev, void* stream);
/* y = act(bilinear x2 of x): the up-sampled tensor in post-activation storage (ups_conv_desc.out_act) for a consuming residual block */
int ups_bilinear2x_fwd_act(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act, float slope,
                           void* stream);
/* ... the same, also writing the sign bits of the stored values (ups_conv_desc.sign_out layout: [n][2h][2w][c / 8] bytes);
 * 16-bit dtypes. */
int ups_bilinear2x_fwd_bits(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act, float slope,
                            void* sign_bits, void* stream);
/* bits[i] = sign byte of the 8 elements x[8 i .. 8 i + 7] (16-bit dtype), chunks = elements / 8 */
int ups_sign_pack(const void* x, int32_t dtype, int64_t chunks, void* sign_bits, void* stream);
/* The same (bf16) with an fp8 copy of the result for a consuming fp8 convolution (ups_conv_desc.in_f8): max |act(y)| goes to
 * amax[64]; with y_f8 != NULL also e4m3 (e5m2 != 0: e5m2) of act(y) * *scale, one byte per element, laid out like y. */
int ups_bilinear2x_fwd_f8(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* y_f8, const float* scale,
                          float* amax, int32_t act, float slope, int32_t e5m2, void* stream);
int ups_bilinear2x_bwd_f8(const void* gy, void* gx, int32_t n, int32_t h, int32_t w, int32_t c, void* gx_f8, const float* scale,
                          float* amax, int32_t e5m2, void* stream);
/* activate + global spatial mean (M:50-51): y[n][c] = mean_hw act(x) */
/* y = elu(x) = x > 0 ? x : exp(x) - 1 (nn.py:747-758 `activate(x, "elu")`), gx = gy * (x > 0 ? 1 : exp(x)); n elements of dtype
 * UPS_F32 / UPS_BF16 / UPS_F16 (n a multiple of 8). */
int ups_elu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream);
int ups_elu_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int64_t n, void* stream);
int ups_act_mean_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t hw, int32_t c, int32_t act, float slope, void* stream);
int ups_act_mean_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int32_t n, int32_t hw, int32_t c, int32_t act, float slope, void* stream);
/* 2x2/2 max pool on pre-activations (Keras VGG19 block*_pool); bwd routes to the first maximal element */
int ups_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
/* bf16 form that also hands the pooled tensor to an fp8 convolution (Keras VGG19 pools in front of block{2..5}_conv1; edflow
 * VGG19Features, external): max |act(y)| into amax[64]; with y_f8 != NULL the e4m3 bytes of act(y) * *scale next to y. */
int ups_maxpool2_fwd_f8(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* y_f8, const float* scale,
                        float* amax, int32_t act, float slope, void* stream);
int ups_maxpool2_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
/* copy `c` channels between tensors with different physical widths / offsets (concat for N:1049-1051) */
int ups_copy_channels(const void* src, int32_t lds, void* dst, int32_t ldd, int32_t dtype, int64_t rows, int32_t c, void* stream);
/* dst[row][0..c) += src[row][0..c) */
int ups_add_channels(const void* src, int32_t lds, void* dst, int32_t ldd, int32_t dtype, int64_t rows, int32_t c, void* stream);

/* ---------------------------------------------------------------- perceptual loss pieces (M:607-619, edflow VGG19Features)
 * [-1,1] RGB fp32 [n,h,w,3] or dtype [n,h,w,ldx] -> BGR*255 - mean, dtype [n,h,w,8] (channels 3..7 zero) */
int ups_vgg_preprocess_fwd(const void* x, int32_t x_is_f32, int32_t ldx, void* y, int32_t dtype, int64_t pixels, void* stream);
int ups_vgg_preprocess_bwd(const void* gy, void* gx, int32_t dtype, int32_t ldgx, int64_t pixels, void* stream);
/* partial[block] = sum |act(a) - act(b)| ; loss = scale * sum(partial) is finished by ups_sum_scale */
int ups_l1_fwd(const void* a, const void* b, int32_t dtype, int64_t rows, int32_t c, int32_t ld, int32_t act,
               float* partial, int32_t nblocks, void* stream);
/* gb[row][c] = -scale * sign(act(a)-act(b)) * act'(b), pad channels zero (gradient w.r.t. the 2nd operand) */
int ups_l1_bwd(const void* a, const void* b, void* gb, int32_t dtype, int64_t rows, int32_t c, int32_t ld, int32_t act,
               const float* scale_dev, float scale, void* stream);
/* out[0] (+)= scale * sum(partial[0..n)) */
int ups_sum_scale(const float* partial, int32_t n, float scale, float* out, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------- part path
 * l = mean + eps (N:1427-1433); m = softmax_P(l) (N:58-62); hard = (m == max_P m) (N:134-136);
 * argmax = first maximal index (M:447,470); hard_bits[pixel] = bit set of the hard mask (bit p = hard[pixel][p], P <= 32:
 * the compact form the part-masked convolution reads).  eps, hard, argmax, hard_bits may be NULL.  All fp32, [pixels][P]. */
int ups_part_softmax_fwd(const float* mean, const float* eps, float* l, float* m, float* hard, int64_t* argmax,
                         uint32_t* hard_bits, int64_t pixels, int32_t P, void* stream);
/* The same plus, from the same pass, the spatial soft-max moments of gamma * hard (the one-hot map has them in closed form from
 * per-part integer sums of the hard pixels' coordinates): stats [n][P][8] as ups_spatial_moments(hard, gamma, no rectangle) writes
 * them (M:437-440: the input of the rectangle centres).  mean is [n,h,w,P]; h*w must be a multiple of the kernel's pixel tile
 * = ups_part_softmax_moments_tile(P) (else UPS_E_ARG: call ups_part_softmax_fwd + ups_spatial_moments); scratch:
 * ups_part_softmax_moments_ints(n*h*w, P) int32. */
int32_t ups_part_softmax_moments_tile(int32_t P);
size_t ups_part_softmax_moments_ints(int64_t pixels, int32_t P);
int ups_part_softmax_moments_fwd(const float* mean, const float* eps, float* l, float* m, float* hard, int64_t* argmax,
                                 uint32_t* hard_bits, int32_t n, int32_t h, int32_t w, int32_t P, float gamma, float* stats,
                                 int32_t* scratch, void* stream);
/* spatial soft-max moments (N:65-71, N:1541-1587) of gamma*x per (n,p) over H*W, optionally masked by
 * (1 - rect) with integer rectangle centres `rect_c` [n*P][2] (y,x) and half sizes:
 * stats[n][p] = {max, Z, sum e*k, sum e*k*gy, sum e*k*gx, sum e*k*(gy^2+gx^2), sum e*k*gy^2, 0},  e = exp(gamma*x - max) */
int ups_spatial_moments(const float* x, int32_t n, int32_t h, int32_t w, int32_t P, float gamma,
                        const int32_t* rect_c, int32_t half_h, int32_t half_w, float* stats, void* stream);
/* The same pass also sums x * log(P * x + 1e-20) over the whole map -- categorical_kl of view 1's soft map (M:21-25, 659-665), the
 * only other prior term of that view, which therefore needs no pass of its own: kl_sums16[0] = the sum (slots 1..15 zeroed: the
 * layout of ups_prior_desc.sums). */
int ups_spatial_moments_kl(const float* x, int32_t n, int32_t h, int32_t w, int32_t P, float gamma,
                           const int32_t* rect_c, int32_t half_h, int32_t half_w, float* stats, float* kl_sums16, void* stream);
/* `stats` must hold n*P*8 floats of result followed by scratch; total = ups_spatial_moments_floats(n, P). */
size_t ups_spatial_moments_floats(int32_t n, int32_t P);
/* px[n*P][2] = (row, column) centre of the rectangle tfutils.draw_rect paints for int32(mu*h/2 + h/2) (M:441,459;
   truncation) from the un-masked stats.  xy_order != 0: the helper reads the (y, x) pair as (x, y) -- row centre from mu_x,
   column centre from mu_y (the reading the reference's step-0 patch_loss supports; oracle/ref_model.py draw_rect) */
int ups_moments_to_px(const float* stats, int32_t count, int32_t h, int32_t xy_order, int32_t* px, void* stream);
/* tfutils.draw_rect (external; semantics inferred): out [n,h,w,P] fp32 */
int ups_draw_rect(const int32_t* px, int32_t n, int32_t h, int32_t w, int32_t P, int32_t half_h, int32_t half_w,
                  float* out, void* stream);
/* mask_parts + apply_partwise transpose (M:176-187, N:97-103): out[(p*B+b)][y][x][0..8) = view[b][y][x][c]*hard[b][y][x][p] */
int ups_mask_parts_fwd(const float* view, const float* hard, void* out, int32_t dtype, int32_t B, int64_t hw, int32_t P, void* stream);
/* g_hard[b][pix][p] = sum_c g_out[(p*B+b)][pix][c] * view[b][pix][c] */
int ups_mask_parts_bwd(const float* view, const void* g_out, float* g_hard, int32_t dtype, int32_t B, int64_t hw, int32_t P, void* stream);
/* unpool_features + concat (M:225-249, 482-484): out[b][pix][f] = sum_p hard*feat[b][p][f]; out[..][F+p] = hard; pad zero */
int ups_unpool_fwd(const float* hard, const float* feat, void* out, int32_t dtype, int32_t B, int64_t hw, int32_t P, int32_t F, int32_t ldo, void* stream);
/* g_hard[b][pix][p] = sum_f g[..f]*feat[b][p][f] + g[..F+p];  g_feat_partial[blk][b][p][f] partial sums (blocks_per_image each) */
int ups_unpool_bwd(const float* hard, const float* feat, const void* g, float* g_hard, float* g_feat,
                   int32_t dtype, int32_t B, int64_t hw, int32_t P, int32_t F, int32_t ldo, void* stream);
/* `g_feat` must hold B*P*F floats of result followed by scratch; total = ups_unpool_bwd_floats(B,P,F). */
size_t ups_unpool_bwd_floats(int32_t B, int32_t P, int32_t F);

/* ---------------------------------------------------------------- mask priors (M:652-797), fused
 * One pass over l/m per view producing the partial sums, one fused backward producing dl.
 * See csrc/priors.hip for the slot layout of `sums`. */
typedef struct {
    int32_t n, h, w, P;
    int32_t view;                 /* 0: KL + entropy + mumford-shah + area + patch + gmrf ; 1: KL + variance */
    int32_t entropy_ce;           /* 0: entropy_func "entropy", 1: "cross_entropy" (M:671-680) */
    float gamma;
    int32_t half_h, half_w;
    float ms_alpha, ms_lambda;    /* 1.0, 1e-2 hard-coded at M:744-746 */
    float w_kl, w_entropy, w_ms, w_area, w_patch, w_gmrf, w_var;  /* schedule weights at this step */
    const float* l;               /* mean + eps */
    const float* l_mean;          /* view 0 only (gmrf) */
    const float* m;               /* softmax(l) */
    const float* hard;
    const int32_t* px;            /* rectangle centres [n*P][2] */
    float* per_np;                /* [n][P][8] per-(image,part) sums */
    float* sums;                  /* [16] global sums (zeroed by the call) */
    const float* g_hard;          /* upstream gradient w.r.t. the STE hard mask or NULL (bwd) */
    float* dl;                    /* d total / d l_mean (bwd); l = l_mean + eps, so the gmrf term is fused in */
    int32_t variant;              /* 0: cub/pennaction SB_model48i; 1: deepfashion SB_model48c (DF:719-776): no rectangles
                                   * (px may be NULL), view 0 adds w_ms_logits * mean_b sum min(ms_alpha * g(l_mean), ms_lambda)
                                   * (deepfashion/code/nn.py:1388-1391,1451-1455; its sum is returned in sums[2], the patch slot),
                                   * view 1 variance = sum_p (S00^2 + S11^2) of the renormalised, un-masked spatial soft-max */
    float w_ms_logits;
    float* dl_rec;                /* optional (bwd): d(reconstruction loss alone)/d l, i.e. the result of a second call with every
                                   * weight zero -- the encoder_0 key sees this one, decoder_visualize sees `dl` */
} ups_prior_desc;
/* `sums` must hold 16 floats of result followed by scratch; total = ups_prior_sums_floats(n, P). */
size_t ups_prior_sums_floats(int32_t n, int32_t P);
int ups_prior_fwd(const ups_prior_desc* d, void* stream);
int ups_prior_bwd(const ups_prior_desc* d, void* stream);

/* ---------------------------------------------------------------- noise (tf.random_normal of the sampling ops, N:1187, 1431)
 * out[i] ~ N(0, 1), i < n: Philox4x32-10 keyed by `seed`, counter = offset + i / 4, Box-Muller.  A pure function of (seed, offset,
 * i); the caller advances offset by ceil(n / 4) per call. */
int ups_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* ---------------------------------------------------------------- MI critic head (M:159-173 last line, 524-536, 800-834, 855)
 * h_pi, h_al [2B][ld] (K logical channels): the two 512-d embeddings of discriminator_model; rows [0,B) are the joint pairs,
 * [B,2B) the marginal pairs.  logits[i] = <h_pi[i], h_al[i]>; out4 = {0.5 (mean softplus(-joint) + mean softplus(marg)),
 * accuracy ((#joint > 0) + (#marg < 0)) / 2B, mean joint logit (logit_constraint(real=False)), 0}.  Single block: fixed
 * reduction order.  bwd: g_loss / g_mim are DEVICE scalars (NULL = 0), the upstream gradients of out4[0] / out4[2];
 * g_h_pi = dlogit * h_al, g_h_al = dlogit * h_pi (either may be NULL). */
int ups_critic_head_fwd(const void* h_pi, const void* h_al, int32_t dtype, int32_t B, int32_t K, int32_t ld, float* logits,
                        float* out4, void* stream);
int ups_critic_head_bwd(const void* h_pi, const void* h_al, const float* logits, const float* g_loss, const float* g_mim,
                        int32_t dtype, int32_t B, int32_t K, int32_t ld, void* g_h_pi, void* g_h_al, void* stream);

/* ---------------------------------------------------------------- state update (M:28-35, 829-834, 861-866, 890-909, 921-930)
 * stats[6] = {mean joint logit of critic 0 (mim), of critic 1 (independent mim), accuracy 0, accuracy 1, loss_dis0, loss_dis1}
 * (batch means, already averaged over the ranks); old_state / new_state [9] = {avg_acc0, avg_acc1, avg_acc_error, avg_loss_dis0,
 * avg_loss_dis1, avg_mim, avg_independent_mim, loa, lor}.  EMAs: new = decay * old + gain * value (value of avg_acc_error =
 * acc1 - acc0); loa' = max(loa + loa_lr (mim - loa_target), 0) when update_loa, lor' = clip(lor + lor_lr (ind - lor_target),
 * lor_min, lor_max) when update_lor, else copied.  One launch, every product and sum rounded on its own. */
int ups_state_update(const float* stats, const float* old_state, float* new_state, float decay, float gain,
                     int32_t update_loa, float loa_lr, float loa_target, int32_t update_lor, float lor_lr,
                     float lor_target, float lor_min, float lor_max, void* stream);

/* ---------------------------------------------------------------- the critics' towers as grouped launches (M:159-173, round 5)
 * discriminator_model is two towers of nin -> (L - 2) x residual_block(k = 1) -> nin on [M = 2B, 512] rows; the three critics of a
 * step are six such towers.  ups_towers_fwd runs layer l of EVERY tower in one launch (L launches), ups_towers_bwd the input
 * gradients likewise (L - 1 launches, + 1 where a tower's first input wants its gradient) and EVERY weight / bias gradient of every
 * tower and layer in one more -- instead of T * L generic convolution calls (each a split-K GEMM + epilogue launch) per direction.
 * UPS_BF16 only, leaky-ReLU, post-activation storage as ups_conv_desc.out_act / res_act describe it: acts[t * L + l] holds
 * lrelu(x) for l < L - 1 (the next layer activates its input) and the plain value for the last layer; a residual layer adds the x it
 * recovers from its stored input.  Requirements (else UPS_E_ARG): k % 32 == 0, n % 128 == 0, square residual layers, T <= 8, L <= 6.
 *   layers  [T * L], tower-major.  w_fwd / w_dgrad: ups_weight_prep's blocked-K copies of the 1x1 kernel; bias [n] or NULL;
 *           grad_w [k][n] (HWIO of a 1x1 kernel) / grad_b [n]: OVERWRITTEN by ups_towers_bwd(want_wgrad = 1); grad_w NULL skips the layer
 *   x0, ld0 [T]: the towers' inputs [M][ld0] (plain values);  acts [T * L]: outputs [M][n] of every layer (caller-owned)
 *   g_out   [T]: gradients w.r.t. the towers' outputs [M][n_last]; a NULL entry leaves that tower out of the call
 *   g_ws    [T * L]: scratch [M][n] per layer (entry l receives the gradient w.r.t. layer l's output; the last entry is unused)
 *   g_x0, ldg0 [T] or NULL: where a tower's input gradient [M][ldg0] is wanted (k of its first layer % 128 == 0) */
typedef struct ups_tower_layer {
    const void*  w_fwd;
    const void*  w_dgrad;
    const float* bias;
    float*       grad_w;
    float*       grad_b;
    int32_t      k, n;
} ups_tower_layer;
int ups_towers_fwd(const ups_tower_layer* layers, int32_t T, int32_t L, const void* const* x0, const int32_t* ld0,
                   void* const* acts, int32_t M, float slope, void* stream);
int ups_towers_bwd(const ups_tower_layer* layers, int32_t T, int32_t L, const void* const* x0, const int32_t* ld0,
                   const void* const* acts, const void* const* g_out, void* const* g_ws, void* const* g_x0,
                   const int32_t* ldg0, int32_t want_wgrad, int32_t M, float slope, void* stream);

/* ---------------------------------------------------------------- full-covariance latent (N:1134-1208, util.py:878-995)
 * params [B][dim + dim(dim+1)/2] fp32.  samples[s][b][i] = mean + L (level[s]*eps[s][b]) ; kl_rows[b][i]. */
/* `level` is a HOST array of S noise levels (S <= 12).  kl_rows may be NULL. */
int ups_latent_fwd(const float* params, const float* eps, const float* level, int32_t S, int32_t B, int32_t dim,
                   float* samples, float* kl_rows, void* stream);
/* g_params = J^T g_samples + gk * d(sum_i kl_rows[b][i])/d params, gk = g_kl_scale * (g_kl_dev ? *g_kl_dev : 1) */
int ups_latent_bwd(const float* params, const float* eps, const float* level, const float* g_samples,
                   const float* g_kl_dev, float g_kl_scale,
                   int32_t S, int32_t B, int32_t dim, float* g_params, void* stream);

/* ---------------------------------------------------------------- optimizer (tf.train.AdamOptimizer, Appendix A.12)
 * p -= lr_t * m / (sqrt(v) + eps) over a flat fp32 buffer; lr_t computed by the caller. */
int ups_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float beta1, float beta2, float eps,
             float grad_scale, void* stream);
/* same update with the step size read from a device scalar (a captured HIP graph of the step is replayed with new values) */
int ups_adam_dev(float* p, const float* g, float* m, float* v, int64_t count, const float* lr_t_dev, float beta1, float beta2,
                 float eps, float grad_scale, void* stream);

/* ---------------------------------------------------------------- thin-plate-spline augmentation (M:282-311)
 * Replaces eddata.utils.tps.ThinPlateSpline (un-vendored; "adapted from CompVis/unsupervised-disentangling", Y:188):
 * out[n][y][x][:] = bilinear sample of img[n] at (x_s, y_s) = T[n] @ [1, x, y, phi(|(x,y) - coord[n][i]|^2)...],
 * phi(d2) = d2 log(d2 + 1e-6), (x, y) on linspace(-1, 1); `_interpolate` convention: pixel = (coord + 1) * size / 2,
 * clamped corner indices.  img / out fp32 [n,h,w,c]; T fp32 [n][2][K+3]; coord fp32 [n][K][2] (x, y); K <= 32.
 * Semantics re-derived from the published algorithm: UNVERIFIED (parity unpinned). */
int ups_tps_warp(const float* img, const float* T, const float* coord, float* out, int32_t n, int32_t h, int32_t w,
                 int32_t c, int32_t K, void* stream);

/* ---------------------------------------------------------------- Gaussian renderers (N:1639-1702, N:1976-2021)
 * tf_hm: P_xy [B][K][2], stddev [B][K][2] on the integer pixel grid (x,y order) -> heat [B,h,w,K] */
int ups_gauss_hm(const float* pts, const float* stddev, float* out, int32_t B, int32_t h, int32_t w, int32_t K, void* stream);
/* tf_hm3: mu [B][K][2] (y,x in [-1,1]), L [B][K][2][2] -> density [B,h,w,K] */
int ups_gauss_hm3(const float* mu, const float* L, float* out, int32_t B, int32_t h, int32_t w, int32_t K, void* stream);

/* ---------------------------------------------------------------- misc
 * dtype conversion of a flat buffer (fp32 <-> bf16) */
int ups_convert(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t count, void* stream);
/* dst[row][0..c) = src[row][0..c) (fp32 -> dtype), dst[row][c..ldd) = 0 */
int ups_pad_convert(const float* src, int32_t c, void* dst, int32_t dtype, int32_t ldd, int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UPSPARTS_HIP_H */
