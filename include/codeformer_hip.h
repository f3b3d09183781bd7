/*
 * codeformer_hip.h -- C ABI of libcodeformer_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the CodeFormer aligned-face hot path
 * (reference: basicsr/archs/codeformer_arch.py:223-280 CodeFormer.forward and the
 * VQGAN blocks of basicsr/archs/vqgan_arch.py).  The reference has no FFI on this
 * path: every arithmetic step is a PyTorch ATen call issued from Python.  Each
 * entry point below therefore cites the ATen call site(s) of the reference that
 * it replaces.  The Python host (codeformer_amd/lib.py) binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch.empty allocations);
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises, nothing allocates, there is no global mutable state
 *     besides the thread-local last-error string (the table cf_device_init() walks is
 *     filled while the library loads and constant afterwards);
 *   - cf_device_init() is called once per device (with that device current) before the
 *     first launch there: kernels with more than 64 KB of dynamic LDS fail to launch without it;
 *   - return 0 on success, <0 on error (cf_last_error() describes it);
 *   - activations are fp32, channels-last ("NHWC": [batch][h][w][c]) unless a flag
 *     says NCHW; token matrices are row-major [rows][cols] (the same memory).
 */
#ifndef CODEFORMER_HIP_H
#define CODEFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 22
#define CF_SPLITK_IN_WORKGROUP (-1) /* cf_conv_desc.split_k: see there (ABI v21) */

typedef void* cf_stream_t; /* hipStream_t */

enum cf_status {
  CF_OK = 0,
  CF_ERR_ARG = -1,      /* unsupported shape / null pointer */
  CF_ERR_LAUNCH = -2,   /* HIP launch error */
};

int cf_version(void);
/* sha256 (first 16 hex digits) of the sources this library was compiled from (csrc + this header), passed by the build script
 * as -DCF_BUILD_ID; "unknown" for hand builds.  codeformer_amd/build.py rebuilds when it differs from the tree's hash. */
const char* cf_build_id(void);
const char* cf_last_error(void);
/* number of compute units of the current device (for host-side grid heuristics) */
int cf_device_cu_count(void);
/* Per-device setup (ABI v18): sets hipFuncAttributeMaxDynamicSharedMemorySize -- a per-device property of a function -- for every
 * kernel of the library that launches with more than 64 KB of dynamic LDS, on the CURRENT device.  Idempotent; call it once per
 * device before the first launch there (codeformer_amd/lib.py does, keyed by torch.cuda.current_device()).  Until v17 every launch
 * site kept a function-static bitmap of initialised devices instead -- mutable state this header promises not to have. */
int cf_device_init(void);

/* ---- convolution / linear as implicit GEMM on fp32 MFMA ------------------------------------
 * Replaces: F.conv2d 3x3 s1 p1 (vqgan_arch.py:132,147,149,243,266,292,314; codeformer_arch.py:142-149),
 *           pad(0,1,0,1)+conv 3x3 s2 (vqgan_arch.py:120-126), nearest x2 + conv (vqgan_arch.py:134-138),
 *           conv 1x1 (vqgan_arch.py:151,173-200), nn.Linear (codeformer_arch.py:104,106,183,192 and the
 *           MultiheadAttention in/out projections), with the surrounding elementwise work fused:
 *           GroupNorm-apply + swish (vqgan_arch.py:14-20), LeakyReLU(0.2) (codeformer_arch.py:143,148),
 *           torch.cat([enc,dec]) (codeformer_arch.py:152), residual adds, GELU (codeformer_arch.py:132),
 *           the SFT combine dec + w*(dec*scale+shift) (codeformer_arch.py:155-156).
 */
enum cf_prologue {
  CF_PRO_NONE = 0,
  CF_PRO_AFFINE = 1,        /* x*scale[b][c] + shift[b][c]                (GroupNorm apply)         */
  CF_PRO_AFFINE_SWISH = 2,  /* y = x*scale+shift ; y*sigmoid(y)           (GroupNorm apply + swish) */
  CF_PRO_LEAKY = 3,         /* x>0 ? x : 0.2*x                                                       */
};
enum cf_epilogue {
  CF_EPI_NONE = 0,      /* acc + bias                                      */
  CF_EPI_RESIDUAL = 1,  /* acc + bias + res                                */
  CF_EPI_SFT = 2,       /* res + sft_w*(res*sft_scale + (acc + bias))      */
  CF_EPI_GELU = 3,      /* gelu_erf(acc + bias)                            */
  /* dense-block family (Real-ESRGAN RRDBNet, basicsr/archs/rrdbnet_arch.py:32-62,111-119); alpha = sft_w */
  CF_EPI_LEAKY = 4,     /* leaky_relu_0.2(acc + bias)                      */
  CF_EPI_AXPY = 5,      /* (acc + bias) * alpha + res        (x5 * 0.2 + x, rrdbnet_arch.py:39)          */
  CF_EPI_AXPY2 = 6,     /* ((acc + bias) * alpha + res) * alpha + res2 ; res2 = sft_scale
                           (last conv of an RRDB: both residuals of rrdbnet_arch.py:39 and :62 in one pass) */
};

/* MFMA operand format (cf_conv_desc.bf16_mfma; the field keeps its ABI-v3 name) */
enum cf_operand {
  CF_OPERAND_F32 = 0,   /* v_mfma_f32_32x32x2_f32, exact fp32 */
  CF_OPERAND_BF16 = 1,  /* v_mfma_f32_32x32x16_bf16; weight from cf_pack_conv_weight[_up2x]_bf16 */
  CF_OPERAND_F16 = 2,   /* v_mfma_f32_32x32x16_f16; weight from cf_pack_conv_weight[_up2x]_f16.  The operand format of the
                           reference's half-precision Real-ESRGAN (inference_codeformer.py:23-27,44); same speed as bf16
                           with 3 more mantissa bits (conv inputs here are O(1): no range problem) */
  CF_OPERAND_F16X2 = 3, /* fp32-grade accuracy on the f16 MFMA pipe: every fp32 operand is split x = hi + lo into two IEEE halves
                           (22 significant bits) and a product is hi*hi + hi*lo + lo*hi -- three v_mfma_f32_32x32x16_f16 with fp32
                           accumulation (cf_split.hip).  Weight from cf_pack_conv_weight_f16x2, acc_scale = 1 / its scale.  3x3
                           stride-1 NHWC convolutions (plain or `upsample`), channels % 32 == 0, cout % 64 == 0, 16x16-tile sizes;
                           prologues as the fp32 kernels, epilogues none / residual / SFT, statistics supported.  Since ABI v17 also
                           stride 2 (Downsample, vqgan_arch.py:117-126: pad_lo == 0, one dense input, c0 % 16 == 0, output a multiple
                           of 8x16): a 2x2 convolution of the space-to-depth view of the input, weight packed with form 2 */
};

/* Border handling of the 3x3 gather (general instantiations; CodeFormer itself only uses zero padding) */
enum cf_pad {
  CF_PAD_ZERO = 0,     /* F.conv2d(padding=1) / Downsample's F.pad(0,1,0,1) */
  CF_PAD_REFLECT = 1,  /* nn.ReflectionPad2d(1) before an unpadded conv (facelib/parsing/parsenet.py:99,107-109) */
  CF_PAD_EDGE = 2,     /* with `upsample`: reflection padding of the nearest-x2 image == edge replication of the source,
                          so the folded sub-pixel kernel clamps its 2x2 footprint (parsenet.py:96-97,105-109) */
};

typedef struct cf_conv_desc {
  const float* in0;       /* first input  [batch][hin][win][c0]  (or NCHW when in_nchw) */
  const float* in1;       /* optional second input, channel-concatenated after in0      */
  int32_t c0, c1;         /* channels of in0 / in1 (c1 = 0: none); both multiples of 16 unless in_nchw */
  int32_t batch, hin, win;
  int32_t hout, wout;     /* s1: hin<<upsample ; s2: hin/2 */
  int32_t cout;           /* valid output channels */
  int32_t cout_pad;       /* packed weight rows (multiple of the N tile: 32/64/128) */
  int32_t taps;           /* 1 or 9 */
  int32_t stride;         /* 1 or 2 (2: pad right/bottom only, vqgan_arch.py:123) */
  int32_t upsample;       /* 1: nearest x2 + 3x3 (vqgan_arch.py:134-138) computed as four 2x2 sub-pixel convolutions of the
                             SOURCE tensor (taps folded at pack time: 4 instead of 9 MACs per weight, mathematically
                             identical); `weight` must come from cf_pack_conv_weight_up2x[_bf16] */
  int32_t in_nchw;        /* 1: in0 is NCHW with c0 <= 4 channels (network input) */
  int32_t out_nchw;       /* 1: write out as NCHW (network output) */
  int32_t prologue;       /* enum cf_prologue */
  int32_t epilogue;       /* enum cf_epilogue */
  const float* pro_scale; /* [batch][c0+c1] */
  const float* pro_shift; /* [batch][c0+c1] */
  const float* weight;    /* packed by cf_pack_conv_weight */
  const float* bias;      /* [cout] or NULL */
  const float* res;       /* [batch][hout][wout][cout] (RESIDUAL / SFT: the `dec` tensor) */
  const float* sft_scale; /* [batch][hout][wout][cout] (SFT) */
  float sft_w;
  float* out;
  double* stats_out;      /* optional: fp64 partial (sum, sumsq) of the OUTPUT per (image, group of stats_cpg channels,
                             tile part), layout [batch][cout/stats_cpg][parts][2], parts = cf_conv2d_stats_parts(d);
                             feeds cf_groupnorm_finalize so the next GroupNorm never re-reads the tensor */
  int32_t stats_cpg;      /* channels per statistics group: power of two in [2,32] dividing cout */
  int32_t bf16_mfma;      /* enum cf_operand.  1: `weight` was packed by cf_pack_conv_weight_bf16 and the contraction runs on
                             v_mfma_f32_32x32x16_bf16 (activations rounded to bf16 after the prologue, fp32 accumulate,
                             fp32 tensors in HBM); 3x3 stride-1 NHWC only.  Used by the bf16 configurations for the
                             generator / CFT convolutions -- never for encoder or Transformer (code indices stay exact) */
  /* Channel strides (floats per pixel) of in0 / in1 / out when they are channel SLICES of wider NHWC buffers; 0 = dense
   * (c0 / c1 / cout).  res and res2 share ld_out.  A dense block (rrdbnet_arch.py:32-39) keeps x1..x4 in one 128-channel
   * buffer: conv_k reads cat(x, growth[:, :32(k-1)]) through (in0, in1, ld_in1 = 128) and writes its 32 channels in place
   * at out = growth + 32(k-1), ld_out = 128 -- torch.cat never materialises.  Any stride other than dense, the epilogues
   * >= CF_EPI_LEAKY, and output sizes that are not a multiple of the 16x16 pixel tile (edge tiles are masked)
   * select the general instantiations: 3x3 stride-1 NHWC, cout_pad 32 or 64, no statistics, fp32 or f16 operands. */
  int32_t ld_in0, ld_in1, ld_out;
  int32_t pad_mode;       /* enum cf_pad */
  int32_t pad_lo;         /* stride 2 only: 0 = pad right/bottom only (vqgan_arch.py:123), 1 = one row/column on every side
                             (parsenet.py ConvLayer scale='down': ReflectionPad2d(1) + stride-2 conv); needs cout_pad % 128 == 0 */
  int32_t winograd;       /* 1: Winograd F(2x2,3x3) evaluation of a 3x3 stride-1 convolution: `weight` comes from
                             cf_pack_conv_weight_winograd (U = G g G^T), 16 instead of 36 multiplies per 2x2 outputs, all in fp32 --
                             the same function in a different summation order.  Dense NHWC
                             tensors, zero padding, hout % 8 == 0, wout % 16 == 0, cout_pad % 64 == 0; prologues as the direct
                             kernel, epilogues none / residual / SFT, statistics supported.  Measured against fp64 its error
                             is below the direct kernel's, so the host uses it for every eligible 3x3 stride-1 convolution.
                             With bf16_mfma == CF_OPERAND_F16X2 (`weight` from cf_pack_conv_weight_winograd_f16x2, acc_scale set)
                             the 16 Winograd-domain GEMMs run on split operands: U and V as hi + lo IEEE halves, three f16
                             MFMAs per product, fp32 accumulation -- 4/9 of the split-half MFMA work of the direct form.
                             With CF_OPERAND_F16 / CF_OPERAND_BF16 (cout % 128 == 0, >= 32x32 pixels): single rounded operands,
                             one MFMA per product (weight: see cf_pack_conv_weight_winograd_bf16).
                             2 (ABI v18; v19: 16x16 patches and the weight layout below): Winograd F(4x4,3x3) with CF_OPERAND_F16X2 operands only (`weight` from
                             cf_pack_conv_weight_winograd43_f16x2, acc_scale set): 36 transform-domain GEMMs per 4x4 outputs = 2.25
                             products per output and input channel instead of 4, interpolation points (0, +-1/2, +-2, inf).  Dense
                             NHWC, zero padding, hout % 16 == 0, wout % 16 == 0, cout == cout_pad, cout % 64 == 0, cin % 16 == 0,
                             cin <= 256, an image of any of its tensors below 2^31 bytes; prologues / epilogues / statistics / act_scale as winograd 1, no split_k.  Its error
                             against fp64 is ~5x that of F(2x2,3x3) (the conditioning of the larger transform), far inside the
                             1e-3 pixel gate; the host uses it for generator / CFT convolutions (vqgan_arch.py:296-323,
                             codeformer_arch.py:136-157) and -- behind a measured logit-margin gate, round 5 -- for the encoder's covered layers.
                             ABI v20: also with CF_OPERAND_F32 (`weight` from cf_pack_conv_weight_winograd43; acc_scale / act_scale
                             unused, act_scale must be null): the same kernel with IEEE-fp32 operands on v_mfma_f32_16x16x4_f32 */
  float acc_scale;        /* CF_OPERAND_F16X2 only (direct or winograd): the accumulator is multiplied by this before the bias is added -- the exact
                             inverse of the power-of-two scale given to cf_pack_conv_weight_f16x2 (> 0) */
  /* Deterministic split-K for layers with few output tiles (one face: 16x16 .. 64x64 pixels), where latency is the serial K loop of
   * a workgroup: split_k workgroups share an output tile, each contracting a contiguous K range; partial accumulators meet in
   * `workspace`, the workgroup drawing the last ticket on counters[tile] adds them in split order (bitwise reproducible, independent of
   * arrival order) and runs the epilogue.  0 = off.  Supported: taps == 1 (64x64 tiles; also selects them with split_k == 1),
   * winograd, CF_OPERAND_F16X2.  workspace: cf_conv2d_workspace_bytes(d) bytes; counters: one zero-initialised uint32 per output
   * tile (cf_conv2d_tiles(d)), left at zero by every launch.
   * ABI v21: CF_SPLITK_IN_WORKGROUP (-1), taps == 1 with CF_OPERAND_F16X2 only (token GEMMs of one to a few faces): the K chunks of a
   * 32 x 32 output tile are shared by the four waves of ONE workgroup and added in chunk order through LDS -- the same bits as every
   * other split count, no workspace / counters (K <= 1024, M % 32 == 0; N % 64 == 0 as for every form: the weight packing). */
  int32_t split_k;
  float* workspace;
  uint32_t* counters;
  /* Range scaling of an UN-NORMALISED input for the 16-bit-operand kernels (CF_OPERAND_F16X2 / F16 / BF16; prologue NONE or LEAKY only):
   * [batch][2] floats (s_b, 1 / s_b) written by cf_act_scale_from_stats / cf_act_scale_from_tensor, or NULL.  The gather multiplies
   * image b's activations by s_b after the prologue and the epilogue multiplies the accumulator by 1 / s_b on top of acc_scale -- both
   * powers of two, so the result is the one an unscaled evaluation would give, for inputs of ANY fp32 magnitude: without it an IEEE-half
   * operand overflows above 65504 (16376 in the Winograd domain, whose input transform sums four activations) and loses its lo half
   * below 2^-3.  Inputs that went through a GroupNorm prologue need none (their range is bounded by gamma, beta and the group size; the
   * host checks that bound when it packs the layer).  Ignored by the exact fp32 kernels. */
  const float* act_scale;
  /* ABI v22: bf16 STORAGE of the activations of this launch (BASELINE configs 3 / 5: "bf16 = bf16 storage + fp32 accumulate in generator +
   * CFT", vqgan_arch.py:276-323, codeformer_arch.py:136-157).  0: fp32 tensors (every launch of rounds 1-5).  1: in0, in1, res, sft_scale
   * and out hold bf16 elements (2 bytes, dense NHWC; the float* fields are then typed loosely) -- consumers widen on load (exact),
   * the epilogue rounds to nearest even once, AFTER the GroupNorm partials (stats_out) were taken from the fp32 values; prologue tables,
   * bias, weights, accumulation and statistics are unchanged.  An NCHW output (the network's last conv) stays fp32.  Supported by the
   * kernels the bf16 mode runs from 64x64 pixels up: winograd == 1 with CF_OPERAND_BF16 (cout % 128 == 0), the direct CF_OPERAND_BF16
   * 3x3 (cout_pad 64) and folded upsample (cout_pad % 128 == 0) forms, fp32 1x1 on images (no split_k), the <= 4-channel NCHW head. */
  int32_t io_bf16;
  /* ABI v22: a second token matrix for the output columns >= alt_cout0 (taps == 1 with CF_OPERAND_F16X2 on token matrices only; NULL: none).
   * The q|k projections of nn.MultiheadAttention read LayerNorm(x) + pos and the v projection LayerNorm(x) (codeformer_arch.py:124-126):
   * with in0 = LN(x) + pos, in0_alt = LN(x), alt_cout0 = 2 E and the whole in_proj weight (3 E rows) they are ONE launch instead of two.
   * in0_alt has in0's shape; alt_cout0 is a multiple of 128.  Per output element the arithmetic is that of the separate launches. */
  const float* in0_alt;
  int32_t alt_cout0;
} cf_conv_desc;

int cf_conv2d(const cf_conv_desc* d, cf_stream_t stream);
/* number of statistics partials per (image, group) the launch described by d will write (>0), or <0 on error */
int cf_conv2d_stats_parts(const cf_conv_desc* d);
/* split-K launches: bytes of workspace / number of output tiles (= counters) the launch described by d needs (0 when split_k <= 1) */
int64_t cf_conv2d_workspace_bytes(const cf_conv_desc* d);
int cf_conv2d_tiles(const cf_conv_desc* d);

/* Pack a PyTorch conv/linear weight [cout][cin][kh*kw] (taps = 1 or 9) into the kernel layout
 * [tap][cin_pad/16][cout_pad][16] (zero padded). */
int cf_pack_conv_weight(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad,
                        float* packed, cf_stream_t stream);
int64_t cf_packed_weight_elems(int cin_pad, int taps, int cout_pad);
/* Winograd-domain weights for cf_conv_desc.winograd: [16 positions][cin_pad/16][cout_pad][16] fp32 (16*cin_pad*cout_pad values),
 * U[xi*4+nu] = (G g G^T)[xi][nu] evaluated in fp64 and rounded once; cout_pad % 64 == 0 */
int cf_pack_conv_weight_winograd(const float* w, int cout, int cin, int cout_pad, int cin_pad, float* packed, cf_stream_t stream);
/* The same for winograd + CF_OPERAND_F16X2: scale * U as hi + lo IEEE halves in MFMA-operand order (16*cin_pad*cout_pad 32-bit
 * words); `scale` is a power of two that puts max|scale * U| into [2^14, 2^15) (cf_conv_desc.acc_scale = 1 / scale) */
int cf_pack_conv_weight_winograd_f16x2(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                       cf_stream_t stream);
/* winograd == 2 (F(4x4,3x3)) + CF_OPERAND_F16X2: scale * U', U' = G' g G'^T with the rows of G scaled by (4, 4, 4, 2, 2, 4) (the input
 * transform carries the inverse powers of two), as hi + lo IEEE halves in MFMA-operand order [36 positions][cin_pad/16][cout_pad/16]
 * [64 lanes][hi 2 words | lo 2 words] (ABI v19: the 16x16x16 MFMA's B operand) = 36*cin_pad*cout_pad 32-bit words; cout_pad % 64 == 0; `scale` a power of two that puts
 * max|scale * U'| into [2^14, 2^15) (cf_conv_desc.acc_scale = 1 / scale) */
int cf_pack_conv_weight_winograd43_f16x2(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                         cf_stream_t stream);
/* winograd == 2 + CF_OPERAND_F32 (ABI v20): U' in fp32, same tensor order with a lane's 16 (32) bytes = the four (eight) k groups
 * c = slab * KS + 4 j + (lane >> 4) of output channel block * 16 + (lane & 15) (the 16x16x4 fp32 MFMA's B operand); KS = 32 when
 * cout_pad % 128 == 0 and cin_pad % 32 == 0 (the form cf_conv2d runs for that shape), else 16.  36*cin_pad*cout_pad floats. */
int cf_pack_conv_weight_winograd43(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed, cf_stream_t stream);
/* winograd + SINGLE 16-bit operands (cout % 128 == 0, at least 32x32 pixels per image: the eight-wave kernel of cf_wsplit.hip; the
 * network's 'fp16' / 'bf16' modes, BASELINE configs 3 and 5): one MFMA per transform-domain product.  CF_OPERAND_F16 reads the hi slot of
 * the split packing above as it is; CF_OPERAND_BF16 takes this buffer: bf16(scale * U) in the hi slot of the same layout, zeros in the
 * lo slot (same size; scale a power of two, acc_scale = 1 / scale) */
int cf_pack_conv_weight_winograd_bf16(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                      cf_stream_t stream);
/* taps == 1 + CF_OPERAND_F16X2 (Linear / 1x1 on token matrices, codeformer_arch.py:104-106,126,132,183,192): w[n][k] -> scale * w as
 * hi + lo IEEE halves in MFMA-operand order (n*k 32-bit words); n % 64 == 0, k % 128 == 0.  The launch needs M = batch*hout*wout % 64 == 0,
 * a dense single input, no prologue / statistics, epilogue none | GELU | residual; split_k as for the fp32 GEMM (same bits for every count) */
int cf_pack_linear_weight_f16x2(const float* w, int n, int k, float scale, void* packed, cf_stream_t stream);
/* bf16 layout [tap][cin_pad/32][cout_pad][32] (round-to-nearest-even); cin_pad % 32 == 0, cout_pad % 64 == 0; the buffer
 * holds cf_packed_weight_elems(cin_pad, taps, cout_pad) bf16 values (half the bytes of the fp32 packing). */
int cf_pack_conv_weight_bf16(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad, void* packed,
                             cf_stream_t stream);
/* Folded weights for cf_conv_desc.upsample: [class 4][tap 4][cin_pad/16][cout_pad][16] fp32 (16*cin_pad*cout_pad values), or the
 * bf16 analogue with 32-channel slabs.  class = (oy&1)*2 + (ox&1); each entry is the fp32 sum of the 3x3 taps that read the same
 * source pixel for that output parity. */
int cf_pack_conv_weight_up2x(const float* w, int cout, int cin, int cout_pad, int cin_pad, float* packed, cf_stream_t stream);
int cf_pack_conv_weight_up2x_bf16(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed,
                                  cf_stream_t stream);
/* IEEE-half analogues of the two bf16 packers (same layouts; cout_pad % 32 == 0), for CF_OPERAND_F16 */
int cf_pack_conv_weight_f16(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad, void* packed,
                            cf_stream_t stream);
int cf_pack_conv_weight_up2x_f16(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed,
                                 cf_stream_t stream);
/* Split-half weights for CF_OPERAND_F16X2: [slab][cin_pad/32][cout_pad][hi 32 | lo 32] IEEE halves (4 bytes per weight, slab = 9
 * taps or, with up2x == 1, the 16 folded class x tap slabs of cf_pack_conv_weight_up2x; up2x == 2 (ABI v17) selects the STRIDE-2
 * form for cf_conv_desc.stride == 2: 4 taps x 4*cin channels, W'[ty][tx][(p, q, c)] = w[2ty + p][2tx + q][c] or 0 -- 16 * cin *
 * cout_pad words, cin % 16 == 0, cin_pad == cin); up2x == 3 (ABI v17): a 1x1 weight [cout][cin] for cf_conv_desc.taps == 1 on images of
 * more than 1024 pixels (the streaming 1x1 form of the same kernel: the ResBlock skip convolutions, vqgan_arch.py:150-164; two-pointer
 * concat, act_scale, epilogues none / residual / SFT, no statistics) -- token matrices of at most 1024 rows per image keep
 * cf_pack_linear_weight_f16x2 and the token GEMM.  Each (folded) fp32 weight is multiplied
 * by `scale` -- a power of two, exact; choose it so that max|w*scale| lies in [2^14, 2^15) -- then hi = half(w'), lo = half(w' - hi)
 * (round-to-nearest-even).  cin_pad % 32 == 0, cout_pad % 64 == 0; cf_conv_desc.acc_scale must be 1 / scale. */
int cf_pack_conv_weight_f16x2(const float* w, int cout, int cin, int up2x, int cout_pad, int cin_pad, float scale, void* packed,
                              cf_stream_t stream);

/* ---- GroupNorm statistics (vqgan_arch.py:14-15: 32 groups, eps 1e-6, biased variance) --------
 * Partials are fp64 (sum, sumsq) tables [batch][groups][parts][2] over an NHWC tensor with c channels whose (fine)
 * groups are cpg channels wide.  They come from a conv epilogue (cf_conv_desc.stats_out) or from
 * cf_groupnorm_stats (stand-alone pass, `parts` blocks per image).
 * finalize: fixed-order sum of the partials; `gmerge` adjacent fine groups form one GroupNorm group (the CFT block
 *           normalises cat[enc, dec]: its groups are pairs of each tensor's own 32 groups); writes, for the c
 *           channels of THIS tensor, scale[b*ld + i] = gamma[i]*rstd, shift[b*ld + i] = beta[i] - mean*scale
 *           (callers pre-offset gamma/beta/scale/shift to the tensor's first channel inside a concatenation;
 *           ld = row stride of the tables = total channels).  count = elements per merged group.
 */
int cf_groupnorm_stats(const float* x, int batch, int hw, int c, int cpg, double* partial, int parts,
                       cf_stream_t stream);
/* ---- per-image power-of-two range scale for cf_conv_desc.act_scale ------------------------------------------------------------
 * act[b] = (s_b, 1 / s_b) with s_b = 2^k such that growth * A_b * s_b lies in [2^13, 2^14), A_b >= max |x| over image b
 * (s_b = 1 for an all-zero or non-finite image; k clamped to [-100, 100]).  growth = 4 covers the Winograd input transform.
 * from_stats:  A_b = sqrt(max over the image's `nper` = groups * parts statistics partials of sumsq) -- a rigorous bound on max |x|
 *              (each partial covers at most a few hundred elements, so it is loose by at most ~5 bits: harmless, the split operands
 *              keep 22 bits over 18 binades) that costs no pass over the tensor: the partials are the ones cf_conv_desc.stats_out
 *              of the PRODUCING launch wrote, layout [batch][nper][2] doubles.
 * from_tensor: A_b = max |x| exactly, one pass over x [batch][n_per_image] (small tensors: the 16x16 quantised feature).
 * scratch: batch * 32 floats (32 partial maxima per image: the reduction runs on 32 workgroups per image, then one wave per image). */
int cf_act_scale_from_stats(const double* partial, int batch, int nper, float growth, float* scratch, float* act, cf_stream_t stream);
int cf_act_scale_from_tensor(const float* x, int batch, int64_t n_per_image, float growth, float* scratch, float* act, cf_stream_t stream);
/* The same table in ONE launch (ABI v17), for the statistics partials of one or two tensors -- the halves of a concatenated input share
 * one scale: A_b = the larger bound -- or (partial_a == NULL) for a tensor x.  cells: 2 * batch zero-initialised uint32 that every launch
 * leaves at zero again (an atomic maximum and a ticket per image: the workgroup drawing the last of the 32 tickets writes act[b]). */
int cf_act_scale_fused(const double* partial_a, int nper_a, const double* partial_b, int nper_b, const float* x, int64_t n_per_image, int batch,
                       float growth, uint32_t* cells, float* act, cf_stream_t stream);
int cf_groupnorm_finalize(const double* partial, int batch, int parts, int c, int cpg, int gmerge, int64_t count,
                          const float* gamma, const float* beta, float eps, float* scale, float* shift, int ld,
                          cf_stream_t stream);
/* ABI v22: the same for up to TWO tensors in one launch -- the halves [enc, dec] of a concatenated GroupNorm input (codeformer_arch.py:152;
 * partial_b NULL: one tensor) -- whose tables land side by side (tensor b's channels start at c_a; gamma / beta / scale / shift address the
 * concatenated channel axis, ld >= c_a + c_b), bitwise the tables of two cf_groupnorm_finalize launches.  With `act` set ([batch][2], and
 * `cells` as for cf_act_scale_fused) the launch also writes the range-scale table of the same tensor(s) -- bitwise cf_act_scale_fused's on the
 * same partials: the ResBlock input that feeds both norm1 and the 1x1 skip convolution (vqgan_arch.py:153-164) needs one launch, not three. */
int cf_groupnorm_finalize2(const double* partial_a, int parts_a, int c_a, int cpg_a, int gmerge_a, const double* partial_b, int parts_b, int c_b,
                           int cpg_b, int gmerge_b, int batch, int64_t count, const float* gamma, const float* beta, float eps, float* scale,
                           float* shift, int ld, float act_growth, uint32_t* cells, float* act, cf_stream_t stream);

/* ---- LayerNorm over the last dim (codeformer_arch.py:108-109,124,131,191; eps 1e-5) ----------
 * y = LN(x)*gamma+beta ; optional ypos = y + pos[row % npos]   (with_pos_embed, codeformer_arch.py:113-116) */
int cf_layernorm(const float* x, int rows, int c, const float* gamma, const float* beta, float eps,
                 const float* pos, int npos, float* y, float* ypos, cf_stream_t stream);

/* ---- attention over 256 keys (vqgan_arch.py:207-221 AttnBlock bmm/softmax/bmm, and the
 *      nn.MultiheadAttention core called at codeformer_arch.py:126) --------------------------------
 * q,k,v: row (b*256 + i), columns [h*head_dim, (h+1)*head_dim) of matrices with leading dimensions
 * ldq/ldk/ldv; out likewise with ldo.  softmax over keys of (q.k^T * scale).  head_dim in {64, 512}. */
int cf_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                 int batch, int heads, int head_dim, int nkeys, float scale, cf_stream_t stream);

/* ---- code prediction (codeformer_arch.py:257-258 softmax+topk(1)) ------------------------------
 * argmax over each fp32 row, lowest index wins ties; idx is int64 like the reference's top_idx. */
int cf_argmax_rows(const float* logits, int rows, int n, int64_t* idx, cf_stream_t stream);

/* ---- codebook lookup (+ AdaIN) (vqgan_arch.py:72-84, codeformer_arch.py:12-43,266) -------------
 * out[b][p][:] = codebook[idx[b][p]][:]; when adain != 0 the per-(b,c) statistics over the ntok positions
 * (unbiased variance + eps) of `out` are replaced by those of lq (NHWC [batch][ntok][dim]). */
int cf_codebook_gather_adain(const int64_t* idx, const float* codebook, int codebook_size, const float* lq,
                             int batch, int ntok, int dim, int adain, float eps, float* out, cf_stream_t stream);

/* ---- nearest-code L2 quantisation (vqgan_arch.py:33-49 VectorQuantizer.forward) ----------------
 * The z.E^T product runs on cf_conv2d (taps=1, the packed codebook as weight); these two finish it:
 * cf_row_sqnorm: out[r] = sum_k x[r][k]^2                      ((z**2).sum(1) and (E**2).sum(1), :40)
 * cf_vq_argmin:  idx[r] = argmin_j (zz[r] + ee[j]) - 2*scores[r][j]   (:40-45; lowest index on ties) */
int cf_row_sqnorm(const float* x, int rows, int dim, float* out, cf_stream_t stream);
int cf_vq_argmin(const float* scores, const float* zz, const float* ee, int rows, int ncodes, int64_t* idx,
                 float* dist_min, cf_stream_t stream);

/* ---- layout converters at the NCHW module boundary ---------------------------------------------- */
int cf_nchw_to_nhwc(const float* x, int batch, int c, int hw, float* y, cf_stream_t stream);
int cf_nhwc_to_nchw(const float* x, int batch, int c, int hw, float* y, cf_stream_t stream);

/* pixel-unshuffle into channels-last (basicsr/archs/arch_util.py:190-206 + the NCHW->NHWC change): x [batch][c][h*s][w*s]
 * -> out [batch][h][w][c_pad], channel (ci*s + dy)*s + dx = x[ci][y*s+dy][x*s+dx], channels >= c*s*s zero (c_pad % 16 == 0).
 * s = 1 is a plain NCHW -> zero-padded NHWC conversion. */
int cf_pixel_unshuffle_nhwc(const float* x, int batch, int c, int h, int w, int s, int c_pad, float* out, cf_stream_t stream);

/* ---- tensor boundary (basicsr/utils/img_util.py:9-35 img2tensor + normalize, :38-94 tensor2img) ---
 * u8 HWC BGR [batch][h][w][3] -> fp32 NCHW RGB in [-1,1] ((x/255 - 0.5)/0.5), and back
 * (clamp[-1,1], (x+1)/2*255, round-half-even, RGB->BGR). */
int cf_img_u8_to_tensor(const uint8_t* img, int batch, int h, int w, float* out, cf_stream_t stream);
int cf_tensor_to_img_u8(const float* t, int batch, int h, int w, uint8_t* img, cf_stream_t stream);
/* inpainting composite (inference_inpainting.py:68-74): x, y, out are [batch][3][h][w]; mask = (x0+x1+x2 == 3);
 * out = (1-mask)*x + mask*y */
int cf_mask_composite(const float* x, const float* y, int batch, int h, int w, float* out, cf_stream_t stream);
/* ABI v22: fp32 -> bf16 copy (round to nearest even) of `numel` elements (a multiple of 8): the fp32 activations that enter the
 * bf16-storage part of the generator (cf_conv_desc.io_bf16) -- the decoder feature in front of the first bf16 Upsample
 * (vqgan_arch.py:129-138) and the encoder taps the fusion blocks concatenate (codeformer_arch.py:152, :228-230). */
int cf_f32_to_bf16(const float* x, int64_t numel, void* out, cf_stream_t stream);

/* ---- bundled StyleGAN2 ops of basicsr/ops (unused by the hot path) ---------------
 * cf_fused_bias_act: basicsr/ops/fused_act/src/fused_bias_act_kernel.cu:20-50 forward (act=3, grad=0):
 *    y = leaky_relu(x + bias[(i / hw) % c], slope) * scale                     (x is NCHW)
 * cf_fused_bias_act_ex: every mode of the op (:36-46, the switch on act*10 + grad): act 1 linear | 3 leaky ReLU(alpha); grad 0 forward,
 *    1 first derivative (y = ref > 0 ? x : x*alpha, ref = the forward OUTPUT, fused_act.py:33,47), 2 second derivative (0); bias
 *    and ref may be NULL; dtype 0 float | 1 IEEE half | 2 bf16 for x / bias / ref / y alike (AT_DISPATCH_FLOATING_TYPES_AND_HALF, :80)
 * cf_upfirdn2d: basicsr/ops/upfirdn2d/src/upfirdn2d_kernel.cu:50-208: planes [nplanes][in_h][in_w].  The reference's six tiled
 *    configurations (:251-291: (up, down) in {(1,1), (2,1), (1,2)}, taps <= 4 / <= 3 or 2) run an LDS-tiled kernel, everything else
 *    the general polyphase kernel (:50-106). */
int cf_fused_bias_act(const float* x, const float* bias, int64_t numel, int c, int hw, float slope, float scale,
                      float* y, cf_stream_t stream);
int cf_fused_bias_act_ex(const void* x, const void* bias, const void* ref, int64_t numel, int c, int hw, int act, int grad, float alpha,
                         float scale, int dtype, void* y, cf_stream_t stream);
int cf_upfirdn2d(const float* x, int nplanes, int in_h, int in_w, const float* kernel, int kh, int kw, int up_x,
                 int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float* y,
                 cf_stream_t stream);

/* ---- alignment warp and paste-back of the whole-image / video path ----------------------------------------------------------
 * Replaces the OpenCV calls of facelib/utils/face_restoration_helper.py: cv2.warpAffine of align_warp_face (:343-344) and of
 * paste_faces_to_input_image (:400, :428, :479), cv2.erode (:430-431, :447), cv2.GaussianBlur (:449), np.sum (:433), the soft-mask
 * blend (:485-492), cv2.resize of the background (:380) and the final astype(uint8) (:497).  OpenCV's fixed-point definitions are
 * followed bit for bit as restated in oracle/paste_oracle.py (OpenCV itself is not available to pin them).  Affine arguments are
 * the matrices that map DESTINATION pixels to SOURCE coordinates (what warpAffine obtains by inverting its M, in double).
 * Regions (rx, ry, rw, rh) are windows of the destination frame; *_region buffers are compact [rh][rw] float32 arrays. */
/* n warps in one launch: dst[i] (u8 [dh][dw][3]) region <- src + i*src_stride (u8 [sh][sw][3]; stride 0 = one source) through
 * inv_dev[i] (DEVICE array of n x 6 doubles); taps outside the source take (b0, b1, b2) */
int cf_warp_affine_u8(const uint8_t* src, int64_t src_stride, int sh, int sw, const double* inv_dev, int n, uint8_t* dst, int dh, int dw,
                      int rx, int ry, int rw, int rh, int b0, int b1, int b2, cf_stream_t stream);
/* float32 single-channel warp into a compact region buffer (border 0); inv_host: 6 doubles in HOST memory */
int cf_warp_affine_f32(const float* src, int sh, int sw, const double* inv_host, float* dst_region, int rx, int ry, int rw, int rh,
                       cf_stream_t stream);
/* cv2.erode with a k x k rectangle (anchor k/2; k == 0 -> OpenCV's 3x3 default); samples outside the region do not take part */
int cf_erode_f32(const float* in_region, float* tmp_region, float* out_region, int rh, int rw, int k, cf_stream_t stream);
/* cv2.GaussianBlur(ksize, 0) with the caller's taps (DEVICE, ksize floats), BORDER_REFLECT_101 at the borders of the ch x cw frame
 * the region (rx, ry) lies in; frame positions outside the region count as 0 */
int cf_gaussian_blur_f32(const float* in_region, float* tmp_region, float* out_region, int rh, int rw, int rx, int ry, int ch, int cw,
                         const float* taps_dev, int ksize, cf_stream_t stream);
/* 64 fp64 partial sums of x[0..n) in a fixed order (the caller adds them) */
int cf_sum_f32(const float* x, int64_t n, double* partials64, cf_stream_t stream);
/* canvas (f32 [ch][cw][3]) region <- m * (ero * warp(face)) + (1 - m) * canvas, m = soft (or min(parse, soft) when parse != NULL) */
int cf_paste_blend(float* canvas, int ch, int cw, const uint8_t* face, int fh, int fw, const double* inv_host, const float* ero_region,
                   const float* soft_region, const float* parse_region, int rx, int ry, int rw, int rh, cf_stream_t stream);
/* cv2.resize(src u8 [sh][sw][3], INTER_LINEAR) -> f32 [dh][dw][3] (a plain widening copy when the sizes are equal) */
int cf_resize_linear_u8(const uint8_t* src, int sh, int sw, float* dst, int dh, int dw, cf_stream_t stream);
/* cv2.resize(src u8 [sh][sw][3], (dw, dh), INTER_AREA) -> u8 [dh][dw][3], dh <= sh and dw <= sw: the reduction of a frame to the detector's
 * working size (facelib/utils/face_restoration_helper.py:208-215) */
int cf_resize_area_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, cf_stream_t stream);
/* astype(np.uint8) of a float image in [0, 256): truncation */
int cf_f32_to_u8_trunc(const float* src, int64_t n, uint8_t* dst, cf_stream_t stream);
/* parse-map colouring out[i] = lut[labels[i]] (face_restoration_helper.py:468-471; lut_host: nlut <= 32 floats in HOST memory) */
int cf_label_lut_f32(const int64_t* labels, int64_t n, const float* lut_host, int nlut, float* out, cf_stream_t stream);
/* in place: x * scale inside the frame of `border` pixels, 0 on it (face_restoration_helper.py:476-481), x: [batch][h][w] */
int cf_scale_clear_border_f32(float* x, int batch, int h, int w, int border, float scale, cf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CODEFORMER_HIP_H */
