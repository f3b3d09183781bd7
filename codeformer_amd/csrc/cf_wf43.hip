// Winograd F(4x4,3x3) convolution with split-half operands on 16x16-pixel patches (gfx950), three instantiations of one kernel:
//   <PRO, EPI, 8, 16>   64 output channels per eight-wave workgroup, TWO workgroups per CU (the text below describes this form)
//   <PRO, EPI, 16, 32>  128 output channels per sixteen-wave workgroup on 32-channel slabs (see "The 16-wave form" at the end)
//   <PRO, EPI, 16, 16>  the same on 16-channel slabs (cin % 32 != 0)
//
// Serves 3x3 stride-1 convolutions of generator and fusion (CFT) blocks (vqgan_arch.py:141-164,296-323, codeformer_arch.py:136-157) and,
// since round 5, the encoder's covered layers (vqgan_arch.py:243-262) -- the encoder decides the code indices, so that use sits behind a
// measured logit-margin gate (tests/test_gpu_real_images.py::test_encoder_logit_margin).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d: 6x6 input tile, g: 3x3 kernel, Y: 4x4 outputs
// with the interpolation points (0, +-1/2, +-2, inf) -- the set with the smallest mean error among those tools/winograd_f43_numerics.py
// compares -- and the rows of B^T scaled by D = diag(1/4, 1/4, 1/4, 1/2, 1/2, 1/4) (G by D^-1; powers of two, so every intermediate is
// the unscaled one times a power of two and rounds identically).  With D the largest absolute row sum of B^T is 1.875: a
// transform-domain value is at most 3.52 max|d|, below the 4 max|d| of F(2,3), so the IEEE-half operand range rules of the F(2,3)
// kernels (GroupNorm bound on the host, act_scale for un-normalised inputs, growth 4) hold unchanged.
//   B'^T = [ .25 0 -1.0625 0 .25 0 ;  0 -.5 -1 .125 .25 0 ;  0 .5 -1 -.125 .25 0 ;  0 -.25 -.125 1 .5 0 ;  0 .25 -.125 -1 .5 0 ;
//            0 .25 0 -1.0625 0 .25 ]
//   G'   = [ 4 0 0 ; -32/15 -16/15 -8/15 ; -32/15 16/15 -8/15 ; 1/15 2/15 4/15 ; 1/15 -2/15 4/15 ; 0 0 4 ]
//   A^T  = [ 1 1 1 1 1 0 ; 0 .5 -.5 2 -2 0 ; 0 .25 .25 4 4 0 ; 0 .125 -.125 8 -8 1 ]
// The 36 transform-domain GEMMs  M[xi,nu][tile][n] = sum_c V[xi,nu][tile][c] U[xi,nu][c][n]  run on v_mfma_f32_16x16x16_f16 with U and V
// as hi + lo IEEE halves (hi*hi + lo*hi + hi*lo, fp32 accumulation): 36 positions per 16 outputs = 2.25 products per output pixel and
// input channel instead of the 4 of F(2,3).
//
// Why this shape.  The first two forms of this kernel (one eight-wave workgroup per CU owning a 16x32 patch on 32-row MFMAs, 256
// registers per lane; stage-synchronous, then two wave groups half a slab apart: profiles/r04_f43_sync_ablation.txt) measured 1.16 ms
// on 64 -> 64 @ 512x512 x 16, no better than F(2,3), with the stages adding up -- and the largest of them was the EPILOGUE (0.43 ms):
// a CU moves about 11 bytes per clock to and from HBM whatever the other CUs do, so the 262 KB a patch reads (residual) and writes
// take 24k cycles during which a single resident workgroup computes nothing, and its 43k cycles of compute leave the memory path idle.
// Only a second resident workgroup overlaps the two.  Hence 16 tiles per patch (the 16-row MFMA; its accumulators are 72 registers per
// lane at 64 channels), 128 registers per lane, under 80 KB of LDS: two workgroups per CU, four waves per SIMD, each workgroup's
// memory phases under the other's compute.  The price: the legacy 16x16x16 MFMA issues at half the rate of 32x32x16 (these layers
// keep the matrix pipe under a quarter busy), and a weight fragment now serves 16 instead of 32 tiles (twice the fragment bytes out
// of L2 per output).
//
// Work decomposition (512 threads = 8 waves, 80,896 bytes of LDS):
//   * a workgroup owns a 16x16 output patch of ONE image (4x4 tiles of 4x4 outputs) x 64 channels; K loop over 16-channel slabs, two
//     barrier intervals per slab:     T: waves 4..7: prologue + store of patch(s + 1), request of patch(s + 2);
//                                        waves 0..3: input transform of slab s (patch(s) -> V)
//                                     M: every wave: its MFMAs of slab s (V x weight fragments straight from L2)
//   * gather: the 18x18 halo patch of a slab (324 pixels x 4 channel quads, six float4 items per thread of waves 4..7) is requested
//     one interval pair ahead (registers), passes the GroupNorm-apply / swish or LeakyReLU prologue, zero padding and concat as in the other
//     kernels, and is written to one of TWO patch buffers (slab parity): [324 pixel slots][16 floats]; the 64-byte slot p lives at
//     p ^ ((p >> 2) & 3) (low two bits), which spreads the four tiles a half-wave reads over the four 64-byte windows of a bank row;
//   * input transform (waves 0..3): item = (xi half, tile, channel pair): column pass for three rows of B'^T d (five of the six tile
//     rows are read), row pass for their six nu, split into hi + lo, written to V[36][16 tiles][4 quads: 4 hi halves | 4 lo halves]
//     (64-byte rows; quad q of tile t at q ^ s(t >> 2), s = 0, 2, 3, 1: conflict-free for the lane groups of ds_read_b128);
//   * MFMA interval: wave = (xi half g = wave >> 2, 16-channel block nb = wave & 3): positions 18 g .. 18 g + 17 = 72 accumulator
//     registers; one ds_read_b128 per position is the A fragment (hi | lo), one global_load_dwordx4 the B fragment (hi | lo), ring of four;
//   * epilogue: two passes (tile columns 0, 1 / 2, 3) through LDS: M[36][8 tiles][64 ch] over patch buffers + V; item = (tile, channel
//     pair, output-row half): reads 30 of the tile's 36 transform-domain values, contracts xi for its two output rows and nu for the
//     four columns, applies acc_scale / bias / residual / SFT, stores 8-byte pairs (a half-wave writes 256 contiguous bytes per pixel) and
//     accumulates the GroupNorm statistics of what it wrote (fp32 over four values, then fp64; fixed shuffle order, the waves joined
//     through LDS in wave order: one partial per patch and group).
//
// The 16-wave form (layers with a multiple of 128 output channels): the same patch for 128 channels -- gather, prologue and transform of
// a patch serve both channel halves -- at 1024 threads x 128 registers = the whole register file, one workgroup per CU, 159,744 bytes of
// LDS.  32-channel slabs: v_mfma_f32_16x16x32_f16 at the full MFMA rate and half the barrier intervals per input channel; 128-byte pixel
// slots (the same swizzle on bit 7); waves 0..7 transform (wave = (xi half, tile row), lane = (tile column, 16 channel pairs)), waves
// 8..15 gather in two halves of three items (half 0 requested behind the M interval, half 1 inside the T interval, where no fragment
// ring is live); V rows of 128 bytes in 8-byte units placed by f4_vu (four ds_read_b64 per A fragment without bank conflicts); B
// fragments of 32 bytes per lane, ring of three positions, ONE A register set -- what 128 registers hold beside 72 accumulators (a ring
// of two does not cover L2 latency: 0.79 vs 0.72 ms on 128 -> 128 @ 256x256 x 16).  Its M interval is bound by the weight stream: 36 x 32
// x 128 x 4 B = 590 KB per slab through the CU's 64 B/clk vector-memory path (profiles/r04_f43_k32_stage_timing.txt).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "cf_common.h"




namespace {

constexpr int F4_TH = 16, F4_TW = 16;            // output patch of a workgroup
constexpr int F4_PW = F4_TW + 2;                 // halo patch 18 x 18
constexpr int F4_NPIX = (F4_TH + 2) * F4_PW;     // 324 pixels = 324 slots (rows are not padded: the swizzle works on the linear slot index)
constexpr int F4_SLOTS = F4_NPIX + 4;            // + one aligned group of four: the target of the padding items
constexpr int F4_DUMMY = F4_NPIX;
constexpr int F4_NT = 16;                        // tiles per patch (4 x 4) = one MFMA row tile
// NW = waves per workgroup: 8 -> 64 output channels, two workgroups per CU; 16 -> 128 output channels, one workgroup per CU (the gather,
// prologue and transform of a patch then serve 128 channels).  Waves 0..3 transform, waves 4..NW-1 gather: 6 (NW 8) / 2 (NW 16) items each.
constexpr int F4_PATCH_FLOATS = F4_SLOTS * CF_BK;   // 5248 floats = 20992 bytes per buffer
constexpr int F4_PS = F4_NT * CF_BK;             // 256 floats between positions of V
constexpr int F4_V_FLOATS = 36 * F4_PS;          // 9216
constexpr int F4_M_FLOATS = 36 * 8 * 64;         // 18432: one pass of the epilogue of the 8-wave form (36 positions x 8 tiles x 64 channels)
constexpr int F4_TAB = 256;                      // GroupNorm scale / shift rows of the image (cin <= 256) ...
constexpr int F4_TAB_32 = 512;                   // ... cin <= 512 on 32-channel slabs (round 6: the 512 -> 256 fusion convolution at 64x64 in precision 'fp32'; its LDS has the room)
constexpr int F4_LDS_FLOATS = 2 * F4_PATCH_FLOATS + F4_V_FLOATS + 2 * F4_TAB;   // 80,896 bytes: two workgroups per CU
static_assert(F4_M_FLOATS <= 2 * F4_PATCH_FLOATS + F4_V_FLOATS, "epilogue staging must fit the patch buffers + V");
static_assert(F4_LDS_FLOATS * 4 <= 81920, "LDS budget of two workgroups per CU");
constexpr int F4_LDS_FLOATS_16 = 2 * F4_M_FLOATS;   // 16-wave form: its epilogue pass stages 36 x 8 tiles x 128 channels = 147,456 bytes (the slab loop needs the 80,896 above)
static_assert(F4_LDS_FLOATS_16 >= F4_LDS_FLOATS && F4_LDS_FLOATS_16 * 4 <= 163840, "LDS budget of the 16-wave form");
constexpr int F4_LDS_FLOATS_32 = 2 * F4_SLOTS * 32 + 36 * F4_NT * 32 + 2 * F4_TAB_32;   // 16 waves, 32-channel slabs: 161,792 bytes
static_assert(F4_LDS_FLOATS_32 >= F4_LDS_FLOATS_16 && F4_LDS_FLOATS_32 * 4 <= 163840, "LDS budget of the 32-channel-slab form");

typedef _Float16 f4_f16x4 __attribute__((ext_vector_type(4)));
typedef float f4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned f4_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned f4_u32x2 __attribute__((ext_vector_type(2)));

// Buffer addressing (scalar descriptor + 32-bit lane offset + scalar offset) for every global access of the kernel: with plain
// pointers hipcc keeps one 64-bit lane address per unrolled access alive across the slab loop (18 weight-fragment pointers, 6 gather
// pointers, 24 epilogue pointers) -- at 128 registers per lane that is what spilled.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f4_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 f4_ld128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f4_f32x2 f4_ld64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void f4_st64(f4_f32x2 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f4_u32x2, v), r, (int)voff, (int)soff, 0);
}

struct F4Args {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, h, w;
  int cout;
  int prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [36 pos][nchunks][cout/16][64 lanes][hi 2 words | lo 2 words]  (cf_pack_conv_weight_winograd43_f16x2)
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float acc_scale;
  const float* act_scale;
  float* out;
  double* stats_out;
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
  int nt_out;   // non-temporal output stores (cf_common.h: cf_store16)
};

// quad swizzle of V: tile row ty -> 0, 2, 3, 1
__device__ __forceinline__ int f4_vs(int ty) { return (0x78 >> (2 * ty)) & 3; }

// unit (8 bytes) of a 128-byte V row of the 32-channel-slab form that holds part p (0, 1: hi halves 0..3 / 4..7; 2, 3: lo) of octet lq of
// tile t: a bijection (lq, p) -> 0..15 per tile, and for one p the 16 tiles x 2 octets a half-wave reads with ds_read_b64 cover all banks
__device__ __forceinline__ unsigned f4_vu(unsigned lq, unsigned p, unsigned t) { return (lq & 1u) + 2u * (((t >> 1) + 4u * (lq >> 1) + p) & 7u); }

// KS = channels per slab: 16 (v_mfma_f32_16x16x16_f16; both forms) or 32 (v_mfma_f32_16x16x32_f16 at twice the rate and half the barrier
// intervals per input channel: the 16-wave form only -- its 157 KB of LDS hold two 41 KB patch buffers and a 72 KB V)
// F32 = IEEE-fp32 operands on v_mfma_f32_16x16x4_f32 (precision 'fp32': BASELINE config 2 to the letter) instead of hi + lo halves: the SAME
// data movement -- a lane's A and B fragments are 16 (32) bytes either way: four (eight) fp32 values k = 4 j + (lane >> 4) instead of
// [4 hi halves | 4 lo halves] -- with four (eight) fp32 MFMAs per position instead of three f16 ones; no weight / activation scale.
// (Round 5 also built an overlapped 16-wave form -- one barrier interval per slab, V double-buffered, the transform of slab s + 1 under the
// weight stream of slab s: bitwise the two-interval <., ., 16, 16> form, 4-13 % fewer cycles per patch and the same launch time at the
// power cap, split-half operands in round 5 and fp32 operands in round 6 (profiles/r05_f43_ovl_stage_timing.txt, r06_f43_fp32_stage_timing.txt).
// It is not part of the library; commit beb776c holds its source.)
// UP (round 6; fp32 operands, no prologue / epilogue operand: the Upsample blocks of precision 'fp32', vqgan_arch.py:129-138): the convolution reads
// the NEAREST-x2 UPSAMPLED image without it ever existing -- halo pixel (iy, ix) of the (h x w) grid is source pixel (iy >> 1, ix >> 1) of the
// (h/2 x w/2) tensor; zero padding, patch layout, transforms and everything behind the gather are those of the plain form.  2.25 products per
// output and input channel instead of the 4 of the folded sub-pixel form (cf_igemm.hip TAPS = 4), which the fp32 pipe executes one by one.
template <int PRO, int EPI, int NW, int KS, bool F32, bool UP = false>
__global__ __launch_bounds__(NW * 64, 4) void wf43_kernel(const F4Args a) {
  static_assert(!UP || (F32 && PRO == CF_PRO_NONE && EPI == CF_EPI_NONE), "the upsampling gather: fp32 operands, no prologue, no epilogue operand");
  static_assert(KS == 16 || (KS == 32 && NW == 16), "32-channel slabs need the LDS of the 16-wave form");
  constexpr int F4_THREADS = NW * 64;
  constexpr int F4_BN = NW * 8;                  // output channels per workgroup
  constexpr int TWV = KS / 4;                    // transform waves: 256 (tile, channel pair, xi half) items per 16 channels
  constexpr int QPP = KS / 4;                    // float4 items per halo pixel
  constexpr int GT = F4_THREADS - TWV * 64;      // gather threads (waves TWV..NW-1)
  constexpr int F4_APT = 384 * QPP / GT;         // float4 gather items per gather thread: 6 (8 waves) / 2 (16 waves) / 6 (16 waves, 32-channel slabs)
  constexpr int PSTEP = GT / QPP;                // pixels between the items of a thread (64 / 192 / 64: multiples of 16, the swizzle period)
  constexpr int SLOTB = KS * 4;                  // bytes per pixel slot of a patch buffer
  constexpr int PATCHF = F4_SLOTS * KS;          // floats per patch buffer
  constexpr int PSK = F4_NT * KS;                // floats between positions of V
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const V = smem + 2 * PATCHF;
  constexpr int VBUF = 36 * PSK;                   // floats of V
  constexpr int TABN = KS == 32 ? F4_TAB_32 : F4_TAB;
  float* const tab = V + VBUF;                     // [scale: TABN][shift: TABN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform by construction: keeps what depends on it in SGPRs)

  int bid = blockIdx.x;
  {  // XCD-contiguous tile order (see cf_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ntile = bid % a.ntn;
  const int mt = bid / a.ntn;
  const int n0 = ntile * F4_BN;
  const int b = mt / a.tiles_per_img;
  const int rt = mt - b * a.tiles_per_img;
  const int tyw = rt / a.tiles_x;
  const int y0 = tyw * F4_TH;
  const int x0 = (rt - tyw * a.tiles_x) * F4_TW;
  const int n = a.cin / KS;   // slabs

  constexpr bool affine = PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH;
  if (affine) {  // this image's GroupNorm rows -> LDS, read per slab by the patch store (first use is behind the first barrier)
    for (int i = tid; i < a.cin; i += F4_THREADS) {
      tab[i] = a.pro_scale[(size_t)b * a.cin + i];
      tab[TABN + i] = a.pro_shift[(size_t)b * a.cin + i];
    }
  }
  float act_s = 1.f, act_is = 1.f;
  if (!affine && a.act_scale) {
    act_s = a.act_scale[2 * b];
    act_is = a.act_scale[2 * b + 1];
  }
  const float act_s02 = 0.2f * act_s;  // LeakyReLU slope folded with the scale: fl(y * (0.2 s)) == fl(0.2 y) * s

  // ---- gather (waves 4..7; waves 0..3 spend the same interval on the input transform): item j of a thread is float4 #k4 of halo pixel
  //      p = u + PSTEP j, u = (tid - 256) >> 2.  Everything per item -- the pixel's offset from the patch origin, whether it lies inside the
  //      image, its LDS slot -- is REBUILT per slab from the thread id (about ten integer instructions per item: these waves wait for
  //      the transform of waves 0..3 anyway) instead of being carried through the slab loop in registers. ----
  // p < 324: pixel (hy, hx) = (p / 18, p % 18); padding items (p >= 324, item 5 of most threads) and pixels outside the image are loaded
  // from the patch's centre pixel and stored as zeros (padding items: not stored).  The LDS slot of item j is that of item 0 plus 64 j
  // (the swizzle repeats every 16 slots): one address + immediates.
  const int pix_origin = (y0 - 1) * a.w + (x0 - 1);  // (wave-uniform; negative on the top / left border: only invalid items would use it there)
  auto item = [&](unsigned u, int j, unsigned& rel, bool& valid) __attribute__((always_inline)) {
    const unsigned p = u + (unsigned)(PSTEP * j);
    const unsigned hy = (p * 3641u) >> 16;  // p / 18 for p < 1024
    const unsigned hx = p - 18u * hy;
    const int iy = y0 - 1 + (int)hy, ix = x0 - 1 + (int)hx;
    valid = (int)(p < (unsigned)F4_NPIX) & (int)((unsigned)iy < (unsigned)a.h) & (int)((unsigned)ix < (unsigned)a.w);  // (no short circuit: no branches)
    if constexpr (UP) {   // `rel` = the SOURCE pixel's index in its (h/2 x w/2) image (the patch's centre pixel for padding items)
      const unsigned sy = valid ? (unsigned)iy : (unsigned)(y0 + 8), sx = valid ? (unsigned)ix : (unsigned)(x0 + 8);
      rel = (sy >> 1) * ((unsigned)a.w >> 1) + (sx >> 1);
    } else {
      rel = valid ? hy * (unsigned)a.w + hx : (unsigned)(9 * a.w + 9);
    }
  };
  const size_t img0 = (size_t)b * a.h * a.w;
  const unsigned img_px = (unsigned)(a.h * a.w);
  const size_t img0_in = UP ? img0 / 4 : img0;          // (the source of the upsampling form has a quarter of the pixels)
  const unsigned img_px_in = UP ? img_px / 4 : img_px;
  // one descriptor per concatenated input, based at this image (the launch checks that an image stays below 2^31 bytes)
  const __amdgpu_buffer_rsrc_t rs_in0 = f4_rsrc(a.in0 + img0_in * a.c0, img_px_in * (unsigned)a.c0 * 4u);
  const __amdgpu_buffer_rsrc_t rs_in1 = f4_rsrc(a.c1 ? a.in1 + img0_in * a.c1 : a.in0, img_px_in * (unsigned)a.c1 * 4u);
  constexpr bool HALVES = F4_APT >= 6;                   // six items per gather thread: a slab's items are requested and stored in two halves
  constexpr int RAN = HALVES ? F4_APT / 2 : F4_APT;     // gather registers that cross the M interval
  f32x4 ra[RAN], rb[HALVES ? RAN : 1];                   // (rb: the second half, live inside the T interval only)
  // unconditional loads from clamped addresses; out-of-image items are zeroed at the store (see cf_winograd.hip)
  auto load_A_range = [&](int chunk, auto j0c, auto j1c) __attribute__((always_inline)) {
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
    // a slab lies in ONE of the concatenated inputs (c0 % 16 == 0): descriptor, channel stride and channel offset are wave-uniform
    const int c = chunk * KS;
    const bool first = c < a.c0;
    const unsigned cs = (unsigned)(first ? a.c0 : a.c1);
    const unsigned soff = (unsigned)(first ? c : c - a.c0) * 4u;
    unsigned tl = (unsigned)tid - (unsigned)(TWV * 64);
    asm volatile("" : "+v"(tl));  // opaque per slab (see above)
    const unsigned k4x = (tl % (unsigned)QPP) * 4u;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      unsigned rel;
      bool valid;
      item(tl / (unsigned)QPP, j, rel, valid);
      const unsigned voff = (__umul24(UP ? rel : (unsigned)(pix_origin + (int)rel), cs) + k4x) * 4u;
      (HALVES && j >= RAN ? rb[j % RAN] : ra[j % RAN]) = first ? f4_ld128(rs_in0, voff, soff) : f4_ld128(rs_in1, voff, soff);
    }
  };
  auto load_A = [&](int chunk) __attribute__((always_inline)) { load_A_range(chunk, std::integral_constant<int, 0>{}, std::integral_constant<int, F4_APT>{}); };
  auto store_patch_range = [&](int chunk, auto j0c, auto j1c) __attribute__((always_inline)) {
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
    unsigned tl = (unsigned)tid - (unsigned)(TWV * 64);
    asm volatile("" : "+v"(tl));
    const unsigned k4 = tl % (unsigned)QPP, slot = tl / (unsigned)QPP;
    char* const pb = reinterpret_cast<char*>(smem + (chunk & 1) * PATCHF) + (((slot & ~3u) | ((slot ^ (slot >> 2)) & 3u)) * (unsigned)SLOTB + k4 * 16u);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      sc = *reinterpret_cast<const f32x4*>(tab + chunk * KS + k4 * 4);
      sh = *reinterpret_cast<const f32x4*>(tab + TABN + chunk * KS + k4 * 4);
    }
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      unsigned rel;
      bool valid;
      item(slot, j, rel, valid);
      f32x4 v = HALVES && j >= RAN ? rb[j % RAN] : ra[j % RAN];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = v[e];
        if (PRO == CF_PRO_AFFINE) y = y * sc[e] + sh[e];
        if (PRO == CF_PRO_AFFINE_SWISH) {
          y = y * sc[e] + sh[e];
          y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));  // same hardware exp / rcp swish as the other conv kernels
        }
        if (PRO == CF_PRO_LEAKY) y = y * (y > 0.f ? act_s : act_s02);
        if (PRO == CF_PRO_NONE) y = y * act_s;
        v[e] = valid ? y : 0.f;
      }
      // the last item covers pixels up to 383 of which 0..323 exist (+ four spare slots); the others would land past the buffer: skipped
      if (j < F4_APT - 1 || slot + (unsigned)(PSTEP * j) < (unsigned)F4_NPIX) *reinterpret_cast<f32x4*>(pb + j * (PSTEP * SLOTB)) = v;
    }
  };

  auto store_patch = [&](int chunk) __attribute__((always_inline)) { store_patch_range(chunk, std::integral_constant<int, 0>{}, std::integral_constant<int, F4_APT>{}); };

  // ---- input transform (waves 0..3): item = (xi half tg, tile (ty, tx), channel pair cp): V[18 tg + aa * 6 + nu], aa = 0..2, nu = 0..5 ----
  // Thread constants of the transform are REBUILT per slab from the lane id (a handful of integer instructions) instead of living
  // in registers across the slab loop: at 128 registers hipcc spills such values and every reload is a wait on vmcnt(0).
  const int tg = (wave >> (KS == 32 ? 2 : 1)) & 1;
  // pixel (tile row r, tile column j) is slot 18 (4 ty + r) + 4 tx + j = 4 G + (q & 3) with q = 2 r + j and G = 18 ty + tx + 4 r + (q >> 2);
  // it is stored at low bits (q & 3) ^ (G & 3), G & 3 = (2 ty + tx + (q >> 2)) & 3: byte offset from the buffer =
  //   256 (18 ty + tx) + cp * 8  [t_base]  +  256 (4 r + (q >> 2))  [immediate]  +  (((q & 3) << 6) ^ sw[q >> 2]),  sw[k] = ((2 ty + tx + k) & 3) << 6
  // (t_base has bits 6, 7 clear, so the swizzle of window k folds into it: tk[k] = t_base | sw[k], and a read costs one xor with 64 (q & 3))
  float* t_hi;  // hi word of channel pair cp: quad cp >> 1 (swizzled by the tile row), word cp & 1; lo word two words further (32-channel slabs: f4_vu)
  float* t_lo;
  auto row_pass = [&](const f4_f32x2 (&zz)[6], int pos) __attribute__((always_inline)) {
    f4_f32x2 v[6];
    v[0] = (zz[0] + zz[4]) * 0.25f - zz[2] * 1.0625f;
    const f4_f32x2 e1 = zz[4] * 0.25f - zz[2];
    const f4_f32x2 o1 = zz[3] * 0.125f - zz[1] * 0.5f;
    v[1] = e1 + o1;
    v[2] = e1 - o1;
    const f4_f32x2 e2 = zz[4] * 0.5f - zz[2] * 0.125f;
    const f4_f32x2 o2 = zz[3] - zz[1] * 0.25f;
    v[3] = e2 + o2;
    v[4] = e2 - o2;
    v[5] = (zz[1] + zz[5]) * 0.25f - zz[3] * 1.0625f;
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {  // operand split, store: per (position, tile) four quads of [4 hi halves | 4 lo halves]
      if (F32) {  // fp32 operands: the pair's two channels go to their own k-group slots (t_hi / t_lo point there)
        t_hi[(pos + nu) * PSK] = v[nu][0];
        t_lo[(pos + nu) * PSK] = v[nu][1];
      } else {
        float hi, lo;
        cf_split_pair(v[nu][0], v[nu][1], hi, lo);
        t_hi[(pos + nu) * PSK] = hi;
        t_lo[(pos + nu) * PSK] = lo;
      }
    }
  };
  // Two parts keep the live set small: first the single row of the half (xi 0 from tile rows 0, 2, 4 / xi 5 from rows 1, 3, 5), then
  // the even / odd pair (xi 1, 2 or 3, 4: tile rows 1..4).  `mid` runs between the two row passes of the second part (the slab's first
  // weight fragments are requested there: the first pass's column sums are dead, the live set is at its smallest).
  auto transform = [&](int chunk, auto mid) __attribute__((always_inline)) {
    const char* const pb = reinterpret_cast<const char*>(smem + (chunk & 1) * PATCHF);
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));  // opaque per slab (see above)
    // 16-channel slabs: wave = (xi half, tile half), lane = (tile, 8 channel pairs); 32-channel slabs: wave = (xi half, tile row), lane = (tile column, 16 pairs)
    const unsigned t_tile = KS == 32 ? 4u * (unsigned)(wave & 3) + (ln >> 4) : 8u * (unsigned)(wave & 1) + (ln >> 3);
    const unsigned t_cp = KS == 32 ? ln & 15u : ln & 7u;
    const unsigned t_ty = t_tile >> 2, t_tx = t_tile & 3u;
    constexpr unsigned SH = KS == 32 ? 7 : 6;   // log2 of the slot size
    const unsigned t_base = (4u << SH) * (18u * t_ty + t_tx) + t_cp * 8u;
    const unsigned tb0 = 2u * t_ty + t_tx;
    unsigned tk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tk[k] = t_base | (((tb0 + k) & 3u) << SH);
    if (F32) {
      // fp32 operands: channel c of the slab is k-group j = c >> 2 of lane quad lq = c & 3; the pair (2 cp, 2 cp + 1) shares j = cp >> 1 and
      // sits in quads lq0 = (2 cp) & 3 and lq0 + 1.  16-channel slabs: quad lq of a tile = its 16-byte chunk (word j); 32-channel slabs:
      // unit p = j >> 1 of quad lq (f4_vu), word j & 1
      const unsigned lq0 = (2u * t_cp) & 3u, j = t_cp >> 1;
      if (KS == 32) {
        t_hi = V + (tg * 18) * PSK + t_tile * 32 + f4_vu(lq0, j >> 1, t_tile) * 2u + (j & 1u);
        t_lo = V + (tg * 18) * PSK + t_tile * 32 + f4_vu(lq0 + 1u, j >> 1, t_tile) * 2u + (j & 1u);
      } else {
        t_hi = V + (tg * 18) * PSK + t_tile * CF_BK + ((lq0 ^ (unsigned)f4_vs((int)t_ty)) << 2) + j;
        t_lo = V + (tg * 18) * PSK + t_tile * CF_BK + (((lq0 + 1u) ^ (unsigned)f4_vs((int)t_ty)) << 2) + j;
      }
    } else if (KS == 32) {
      const unsigned lq = t_cp >> 2, e = t_cp & 3u;
      t_hi = V + (tg * 18) * PSK + t_tile * 32 + f4_vu(lq, e >> 1, t_tile) * 2u + (e & 1u);
      t_lo = V + (tg * 18) * PSK + t_tile * 32 + f4_vu(lq, 2u + (e >> 1), t_tile) * 2u + (e & 1u);
    } else {
      t_hi = V + (tg * 18) * PSK + t_tile * CF_BK + (((t_cp >> 1) ^ (unsigned)f4_vs((int)t_ty)) << 2) + (t_cp & 1u);
      t_lo = t_hi + 2;
    }
    auto px = [&](int r, int j) __attribute__((always_inline)) {  // (r, j compile-time after unrolling)
      const int q = 2 * r + j;
      return *reinterpret_cast<const f4_f32x2*>(pb + (4 << SH) * (4 * r + (q >> 2)) + (tk[q >> 2] ^ (((unsigned)q & 3u) << SH)));
    };
    if (tg == 0) {
      {  // xi = 0: .25 (d0 + d4) - 1.0625 d2
        f4_f32x2 z[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) z[j] = (px(0, j) + px(4, j)) * 0.25f - px(2, j) * 1.0625f;
        __builtin_amdgcn_sched_barrier(0);
        row_pass(z, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // xi 1, 2: e = .25 d4 - d2, o = .125 d3 - .5 d1
        f4_f32x2 zp[6], zm[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f4_f32x2 e = px(4, j) * 0.25f - px(2, j);
          const f4_f32x2 o = px(3, j) * 0.125f - px(1, j) * 0.5f;
          zp[j] = e + o;
          zm[j] = e - o;
          if (j == 2) __builtin_amdgcn_sched_barrier(0);  // (keeps the reads of the later columns from being hoisted)
        }
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zp, 6);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zm, 12);
      }
    } else {
      {  // xi = 5: .25 (d1 + d5) - 1.0625 d3   (position 30 = 18 + 12)
        f4_f32x2 z[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) z[j] = (px(1, j) + px(5, j)) * 0.25f - px(3, j) * 1.0625f;
        __builtin_amdgcn_sched_barrier(0);
        row_pass(z, 12);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // xi 3, 4: e = .5 d4 - .125 d2, o = d3 - .25 d1   (positions 18, 24)
        f4_f32x2 zp[6], zm[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f4_f32x2 e = px(4, j) * 0.5f - px(2, j) * 0.125f;
          const f4_f32x2 o = px(3, j) - px(1, j) * 0.25f;
          zp[j] = e + o;
          zm[j] = e - o;
          if (j == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zp, 0);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zm, 6);
      }
    }
  };

  // ---- MFMA interval: wave = (xi half m_g, 16-channel block m_nb) owns positions 18 m_g + i, i = 0..17 ----
  const int m_g = wave / (NW / 2), m_nb = wave % (NW / 2);
  f32x4 acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight fragments: descriptor over the packed tensor, lane offset 16 lane bytes, everything else in the scalar offset
  const __amdgpu_buffer_rsrc_t rs_w = f4_rsrc(a.weight, 36u * (unsigned)a.cin * (unsigned)a.cout * 4u);
  const unsigned w_pos = (unsigned)a.cin * (unsigned)a.cout * 4u;                                        // bytes between positions
  const unsigned w_chunk = (unsigned)(a.cout * KS) * 4u;                                                 // ... between slabs
  const unsigned w_s0 = (unsigned)(18 * m_g) * w_pos + (unsigned)(n0 / 16 + m_nb) * (unsigned)(64 * KS); // (wave-uniform; a fragment: 64 lanes x KS bytes)
  constexpr std::integral_constant<int, 4> nb4{};
  unsigned lane16;   // (rebuilt per slab, as a_base: see the transform)
  f32x4 bq[4];  // ring of weight fragments: hi | lo
  auto load_B = [&](int chunk, int i, auto nb) __attribute__((always_inline)) {
    bq[i % decltype(nb)::value] = f4_ld128(rs_w, lane16, w_s0 + (unsigned)i * w_pos + (unsigned)chunk * w_chunk);
  };
  auto set_lane16 = [&]() __attribute__((always_inline)) {
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));
    lane16 = ln * 16u;
  };
  const float* a_base;  // row = tile l15, channels 4 lq .. + 3
  f32x4 va[4];  // A fragments: hi | lo
  auto mfma = [&](float a0, float a1, float b0, float b1, f32x4& c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f4_f16x4, f4_f32x2{a0, a1}), __builtin_bit_cast(f4_f16x4, f4_f32x2{b0, b1}), c, 0, 0, 0);
  };
  // Two positions at a time: lo*hi + hi*lo + hi*hi (the order of the F(2,3) kernels) into each position's accumulator, the two chains
  // interleaved (an MFMA never waits for its predecessor's result).  B fragments: ring of four, positions 0..3 requested in the T
  // interval.  A fragments: NA = 4 reads the next pair while this one multiplies (waves 0..3); NA = 2 reads it afterwards (waves
  // 4..7, which carry the 24 gather registers through the end of this interval).
  auto mma_stage = [&](int chunk, auto na, auto after_last_B) __attribute__((always_inline)) {
    constexpr int NA = decltype(na)::value;
    {
      unsigned ln = (unsigned)lane;
      asm volatile("" : "+v"(ln));
      const unsigned t15 = ln & 15u;
      a_base = V + (18 * m_g) * PSK + t15 * CF_BK + (((ln >> 4) ^ (unsigned)f4_vs((int)(t15 >> 2))) << 2);
      lane16 = ln * 16u;
    }
    auto read_A = [&](int i) __attribute__((always_inline)) { va[i % NA] = *reinterpret_cast<const f32x4*>(a_base + i * PSK); };
    read_A(0);
    read_A(1);
#pragma unroll
    for (int i = 0; i < 18; i += 2) {
      if (NA == 4 && i + 2 < 18) {
        read_A(i + 2);
        read_A(i + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 a0 = va[i % NA], a1 = va[(i + 1) % NA], b0 = bq[i % 4], b1 = bq[(i + 1) % 4];
      if (F32) {  // k groups 0..3 in order, the two positions interleaved
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[i], 0, 0, 0);
          acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[i + 1], 0, 0, 0);
        }
      } else {
        mfma(a0[2], a0[3], b0[0], b0[1], acc[i]);
        mfma(a1[2], a1[3], b1[0], b1[1], acc[i + 1]);
        mfma(a0[0], a0[1], b0[2], b0[3], acc[i]);
        mfma(a1[0], a1[1], b1[2], b1[3], acc[i + 1]);
        mfma(a0[0], a0[1], b0[0], b0[1], acc[i]);
        mfma(a1[0], a1[1], b1[0], b1[1], acc[i + 1]);
      }
      if (NA == 2 && i + 2 < 18) {
        read_A(i + 2);
        read_A(i + 3);
      }
      if (i + 4 < 18) {  // refill of the two slots just consumed
        load_B(chunk, i + 4, nb4);
        load_B(chunk, i + 5, nb4);
      }
      // the gather request of waves 4..7 goes BEHIND the slab's last weight-fragment request: loads return in order, so a request
      // placed earlier would have to land before the next fragment can be waited for
      if (i + 4 == 16) after_last_B();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 32-channel slabs: one position at a time (an A fragment is four ds_read_b64 = 8 hi + 8 lo halves, a B fragment two 16-byte loads);
  // one A register set, B ring of three positions (what 128 registers hold beside 72 accumulators), positions 0..2 requested in the T interval
  typedef _Float16 f4_f16x8 __attribute__((ext_vector_type(8)));
  f32x4 xb[3][2], xa[1][2];
  // (round 5 also measured a form in which a wave owns 9 positions x 2 channel blocks -- half the A-fragment LDS reads, bitwise the same, flat on
  //  every shape: profiles/r05_f43_pair_ab.txt; the switch lives in tools/experiments/ablation_and_timing_macros.patch)
  const int p_g = 18 * m_g;                     // first position of this wave
  const int p_nb = m_nb;                        // its 16-channel block
  const unsigned w_s32 = (unsigned)p_g * w_pos + (unsigned)(n0 / 16 + p_nb) * (unsigned)(64 * KS);
  auto load_B32 = [&](int chunk, int i) __attribute__((always_inline)) {
    const unsigned so = w_s32 + (unsigned)i * w_pos + (unsigned)chunk * w_chunk;
    xb[i % 3][0] = f4_ld128(rs_w, lane16, so);
    xb[i % 3][1] = f4_ld128(rs_w, lane16 + 16u, so);
  };
  auto mma_stage32 = [&](int chunk) __attribute__((always_inline)) {
    unsigned ao[4];
    {
      unsigned ln = (unsigned)lane;
      asm volatile("" : "+v"(ln));
      const unsigned t15 = ln & 15u, lq = ln >> 4;
#pragma unroll
      for (unsigned pp = 0; pp < 4; ++pp) ao[pp] = ((unsigned)p_g * (unsigned)PSK + t15 * 32u + f4_vu(lq, pp, t15) * 2u) * 4u;   // bytes from V
      lane16 = ln * 32u;
    }
    const char* const vb = reinterpret_cast<const char*>(V);
    auto read_A = [&](int pos) __attribute__((always_inline)) {
      const f4_f32x2 p0 = *reinterpret_cast<const f4_f32x2*>(vb + ao[0] + pos * (PSK * 4)), p1 = *reinterpret_cast<const f4_f32x2*>(vb + ao[1] + pos * (PSK * 4));
      const f4_f32x2 p2 = *reinterpret_cast<const f4_f32x2*>(vb + ao[2] + pos * (PSK * 4)), p3 = *reinterpret_cast<const f4_f32x2*>(vb + ao[3] + pos * (PSK * 4));
      xa[0][0] = f32x4{p0[0], p0[1], p1[0], p1[1]};
      xa[0][1] = f32x4{p2[0], p2[1], p3[0], p3[1]};
    };
    auto mf = [&](f32x4 av, f32x4 bv, f32x4& c) __attribute__((always_inline)) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f4_f16x8, av), __builtin_bit_cast(f4_f16x8, bv), c, 0, 0, 0);
    };
    read_A(0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (F32) {  // eight k groups: xa / xb [0] = j 0..3, [1] = j 4..7
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][j >> 2][j & 3], xb[i % 3][j >> 2][j & 3], acc[i], 0, 0, 0);
        }
      } else {
        mf(xa[0][1], xb[i % 3][0], acc[i]);
        mf(xa[0][0], xb[i % 3][1], acc[i]);
        mf(xa[0][0], xb[i % 3][0], acc[i]);
      }
      // (one register set: the next fragment is read once this position's MFMAs are issued -- 128 registers)
      if (i + 1 < 18) read_A(i + 1);
      if (i + 3 < 18) load_B32(chunk, i + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto set_lane32 = [&]() __attribute__((always_inline)) {
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));
    lane16 = ln * 32u;
  };
  constexpr std::integral_constant<int, 4> na4{};
  constexpr std::integral_constant<int, 4> na_gather{};   // (12 gather registers at most cross the M interval: room for the A prefetch)
  auto feed = [&](int s) __attribute__((always_inline)) {  // waves 4..7: prologue + store of slab s (if any); its successor is requested inside the M interval
    if (s < n) {
      store_patch(s);
    }
  };

  // ---- slab loop: two barrier intervals per slab; one loop per wave role (a common loop would keep the gather registers of waves 4..7
  //      alive through the transform of waves 0..3: the allocator is per kernel, not per wave) ----
  if constexpr (KS == 32) {
    // 32-channel slabs: waves 0..7 transform, waves 8..15 gather; the gather request of slab s + 2 follows the M interval of slab s
    // (its registers would not fit beside the fragment rings) and has the transform interval of the other waves to land
    if (wave < TWV) {
      __syncthreads();
      __syncthreads();  // patch(0) visible
      for (int s = 0; s < n; ++s) {
        transform(s, [&]() __attribute__((always_inline)) {
          set_lane32();
          load_B32(s, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        load_B32(s, 1);
        load_B32(s, 2);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // V(s) and patch(s + 1) visible
        mma_stage32(s);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // V and patch(s) are free
      }
    } else {
      // the gather of a slab in two halves of three items (12 registers): half 0 is requested behind the M interval (12 registers beside the fragment rings), half 1 at the top of the
      // next T interval, where half 0 is stored first -- these waves wait for the transform of the others anyway
      constexpr std::integral_constant<int, 0> h0{};
      constexpr std::integral_constant<int, F4_APT / 2> h1{};
      constexpr std::integral_constant<int, F4_APT> h2{};
      auto feed2 = [&](int s) __attribute__((always_inline)) {  // patch(s): half 0 is in flight
        if (s < n) {
          load_A_range(s, h1, h2);   // (its own registers: no fragment ring is live in the T interval)
          __builtin_amdgcn_sched_barrier(0);
          store_patch_range(s, h0, h1);
          __builtin_amdgcn_sched_barrier(0);
          store_patch_range(s, h1, h2);
        }
      };
      load_A_range(0, h0, h1);
      __syncthreads();  // (the GroupNorm rows are in LDS)
      feed2(0);
      if (n > 1) load_A_range(1, h0, h1);
      __syncthreads();
      for (int s = 0; s < n; ++s) {
        feed2(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        set_lane32();
#pragma unroll
        for (int i = 0; i < 3; ++i) load_B32(s, i);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        mma_stage32(s);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < n) load_A_range(s + 2, h0, h1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }
  } else if (wave < 4) {
    __syncthreads();
    __syncthreads();  // patch(0) visible
    for (int s = 0; s < n; ++s) {
      transform(s, [&]() __attribute__((always_inline)) {
        set_lane16();
        load_B(s, 0, nb4);
        load_B(s, 1, nb4);
      });
      __builtin_amdgcn_sched_barrier(0);
      load_B(s, 2, nb4);
      load_B(s, 3, nb4);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // V(s) and patch(s + 1) visible
      mma_stage(s, na4, []() __attribute__((always_inline)) {});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // V and patch(s) are free
    }
  } else {
    // With six items per thread (the 8-wave form) a slab's gather goes in two halves of three, as in the 32-channel-slab form: half 0
    // behind the slab's last weight-fragment request inside the M interval (12 registers beside the rings), half 1 at the top of the
    // T interval, where half 0 is stored first.
    constexpr std::integral_constant<int, 0> h0{};
    constexpr std::integral_constant<int, (HALVES ? F4_APT / 2 : F4_APT)> h1{};
    constexpr std::integral_constant<int, F4_APT> h2{};
    auto feedh = [&](int s) __attribute__((always_inline)) {  // patch(s): the items [h0, h1) are in flight
      if (s < n) {
        if (HALVES) load_A_range(s, h1, h2);
        __builtin_amdgcn_sched_barrier(0);
        store_patch_range(s, h0, h1);
        __builtin_amdgcn_sched_barrier(0);
        if (HALVES) store_patch_range(s, h1, h2);
      }
    };
    load_A_range(0, h0, h1);
    __syncthreads();  // (the GroupNorm rows are in LDS)
    feedh(0);
    if (n > 1) load_A_range(1, h0, h1);
    __syncthreads();
    // (two loops: with the gather request under a condition inside one loop, hipcc's wait for the weight fragments requested before
    //  it must also be right for the path without the request -- and then waits for the gather as well)
    int s = 0;
    for (; s + 2 < n; ++s) {
      feedh(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      set_lane16();
#pragma unroll
      for (int i = 0; i < 4; ++i) load_B(s, i, nb4);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      mma_stage(s, na_gather, [&]() __attribute__((always_inline)) { load_A_range(s + 2, h0, h1); });
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
    for (; s < n; ++s) {  // the last two slabs: nothing left to request
      feedh(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      set_lane16();
#pragma unroll
      for (int i = 0; i < 4; ++i) load_B(s, i, nb4);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      mma_stage(s, na_gather, []() __attribute__((always_inline)) {});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  }

  // ---- epilogue: two passes (tile columns 2 th, 2 th + 1) through LDS: M[36 positions][8 tiles][64 channels] over the patch buffers + V ----
  // staging: accumulator register r of position i is tile 4 lq + r (tile row lq = lane >> 4, tile column r), channel 16 m_nb + (lane & 15); pass th takes
  // r = 2 th, 2 th + 1 as staging tile tp = 2 lq + (r & 1).  item = (tp = wave, channel pair e_cp, output-row half e_rh).
  float* const Mst = smem;
  const int e_cp = (lane & 31) + 32 * (wave >> 3), e_rh = lane >> 5, e_tp = wave & 7;   // (channel pair within the workgroup's F4_BN / 2)
  const float acc_s = a.acc_scale * act_is;                // (a product of powers of two: exact)
  const int nn = n0 + 2 * e_cp;
  f4_f32x2 bias2 = {0.f, 0.f};
  if (a.bias) bias2 = *reinterpret_cast<const f4_f32x2*>(a.bias + nn);
  const unsigned e_rowc = (unsigned)a.w * (unsigned)a.cout * 4u;   // bytes between image rows
  const unsigned e_px = (unsigned)a.cout * 4u;                       // ... between pixels
  const unsigned e_img = img_px * (unsigned)a.cout * 4u;
  const __amdgpu_buffer_rsrc_t rs_out = f4_rsrc(a.out + img0 * a.cout, e_img);
  const __amdgpu_buffer_rsrc_t rs_res = f4_rsrc(EPI == CF_EPI_NONE ? a.out : a.res + img0 * a.cout, e_img);
  const __amdgpu_buffer_rsrc_t rs_sft = f4_rsrc(EPI == CF_EPI_SFT ? a.sft_scale + img0 * a.cout : a.out, e_img);
  const bool rh1 = e_rh != 0;
  // xi contraction of this lane's two output rows: rows 0, 1 (e_rh = 0): tA = (s1 + s2) + m0, tB = .5 d1 + 2 d2;
  //                                                 rows 2, 3 (e_rh = 1): tA = .25 s1 + 4 s2,  tB = (.125 d1 + 8 d2) + m5
  const float ca1 = rh1 ? 0.25f : 1.f, ca2 = rh1 ? 4.f : 1.f, cb1 = rh1 ? 0.125f : 0.5f, cb2 = rh1 ? 8.f : 2.f;
  double psum = 0.0, psq = 0.0;  // this wave's GroupNorm partial over both passes
#pragma unroll
  for (int th = 0; th < 2; ++th) {  // (unrolled: `th` selects accumulator registers)
    unsigned e_zero = 0;
    asm volatile("" : "+v"(e_zero));  // opaque 0 placed here: keeps hipcc from hoisting this pass's residual / SFT loads above the slab loop / the previous pass
    // lane part: the output-row half and the channel pair; scalar part: the tile of this wave and pass
    const unsigned voff0 = (unsigned)(2 * e_rh) * e_rowc + (unsigned)(2 * e_cp) * 4u + e_zero;
    const unsigned soff0 = (unsigned)(y0 + 4 * (e_tp >> 1)) * e_rowc + (unsigned)(x0 + 4 * (2 * th + (e_tp & 1))) * e_px + (unsigned)n0 * 4u;
    // residual / SFT operands of this pass first: their latency overlaps the staging
    f4_f32x2 r0[2][4], r1[2][4];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        r0[aa][c] = r1[aa][c] = f4_f32x2{0.f, 0.f};
        if (EPI == CF_EPI_RESIDUAL || EPI == CF_EPI_SFT) r0[aa][c] = f4_ld64(rs_res, voff0, soff0 + aa * e_rowc + c * e_px);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (th > 0) __syncthreads();  // the previous pass's reads are complete (first pass: the slab loop ended with a barrier)
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      float* mp = Mst + ((18 * m_g + i) * 8 + 2 * (lane >> 4)) * F4_BN + m_nb * 16 + (lane & 15);
      mp[0] = acc[i][2 * th];
      mp[F4_BN] = acc[i][2 * th + 1];
    }
    if (EPI == CF_EPI_SFT) {  // (requested here, where the staged accumulator registers are free)
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int c = 0; c < 4; ++c) r1[aa][c] = f4_ld64(rs_sft, voff0, soff0 + aa * e_rowc + c * e_px);
    }
    __syncthreads();
    const float* mq = Mst + e_tp * F4_BN + 2 * e_cp;
    auto col = [&](int nu, f4_f32x2& tA, f4_f32x2& tB) __attribute__((always_inline)) {
      auto m = [&](int xi) __attribute__((always_inline)) { return *reinterpret_cast<const f4_f32x2*>(mq + ((xi * 6 + nu) * 8) * F4_BN); };
      const f4_f32x2 m1 = m(1), m2 = m(2), m3 = m(3), m4 = m(4);
      const f4_f32x2 me = *reinterpret_cast<const f4_f32x2*>(mq + (((rh1 ? 5 : 0) * 6 + nu) * 8) * F4_BN);
      const f4_f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
      const f4_f32x2 x = s1 * ca1 + s2 * ca2, y = d1 * cb1 + d2 * cb2;
      const f4_f32x2 xm = x + me, ym = y + me;
      tA = rh1 ? x : xm;
      tB = rh1 ? ym : y;
    };
    f4_f32x2 o[2][4];
    {
      f4_f32x2 t0[2], ta[2], tb[2];
      col(0, t0[0], t0[1]);
      col(1, ta[0], ta[1]);
      col(2, tb[0], tb[1]);
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const f4_f32x2 s = ta[aa] + tb[aa], d = ta[aa] - tb[aa];
        o[aa][0] = t0[aa] + s;
        o[aa][1] = d * 0.5f;
        o[aa][2] = s * 0.25f;
        o[aa][3] = d * 0.125f;
      }
      col(3, ta[0], ta[1]);
      col(4, tb[0], tb[1]);
      col(5, t0[0], t0[1]);
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const f4_f32x2 s = ta[aa] + tb[aa], d = ta[aa] - tb[aa];
        o[aa][0] += s;
        o[aa][1] += d * 2.f;
        o[aa][2] += s * 4.f;
        o[aa][3] += d * 8.f + t0[aa];
      }
    }
    double dsum = 0.0, dsq = 0.0, dsum1 = 0.0, dsq1 = 0.0;  // (channel 0 / 1 of the pair; joined below: the pair lies in one group)
#pragma unroll
    for (int aa = 0; aa < 2; ++aa) {
      f4_f32x2 rs = {0.f, 0.f}, rq = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f4_f32x2 v = o[aa][c] * acc_s + bias2;
        if (EPI == CF_EPI_RESIDUAL) v += r0[aa][c];
        else if (EPI == CF_EPI_SFT) v = r0[aa][c] + a.sft_w * (r0[aa][c] * r1[aa][c] + v);
        o[aa][c] = v;   // (stored below: the eight stores of a pass sit behind ONE wave-uniform choice of their cache policy)
        rs += v;
        rq += v * v;
      }
      dsum += (double)rs[0];
      dsq += (double)rq[0];
      dsum1 += (double)rs[1];
      dsq1 += (double)rq[1];
    }
    // Output stores, non-temporal for outputs no cache keeps (cf_common.h: cf_store16).  The policy bits are an immediate of the store, so
    // there are two copies of the block behind one branch per pass -- a branch per store made hipcc wait for every load in flight at each of
    // them (whole step 444 -> 403 faces/s on one box before this form).
    if (a.nt_out) {
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f4_u32x2, o[aa][c]), rs_out, (int)voff0, (int)(soff0 + aa * e_rowc + c * e_px), 2);
    } else {
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int c = 0; c < 4; ++c) f4_st64(o[aa][c], rs_out, voff0, soff0 + aa * e_rowc + c * e_px);
    }
    if (a.stats_out) {
      // GroupNorm statistics of the values this wave wrote in this pass (one tile x 64 channels): fp64, fixed shuffle order; the two
      // passes of a wave are joined in registers (psum / psq), the eight waves through LDS below: ONE partial per (image, group, patch)
      const int cpg = a.stats_cpg;
      dsum += dsum1;  // (the pair belongs to one group: cpg is even)
      dsq += dsq1;
      dsum += __shfl_xor(dsum, 32, 64);  // the two output-row halves
      dsq += __shfl_xor(dsq, 32, 64);
      for (int o2 = 1; o2 * 2 < cpg; o2 <<= 1) {  // adjacent channel pairs of one group
        dsum += __shfl_xor(dsum, o2, 64);
        dsq += __shfl_xor(dsq, o2, 64);
      }
      psum = th == 0 ? dsum : psum + dsum;
      psq = th == 0 ? dsq : psq + dsq;
    }
  }
  if (a.stats_out) {
    // the waves' partials -> LDS (over the staging area: every wave is past its reads behind this barrier), summed in wave order by
    // the lanes of wave 0 that own a group: a 512x512 image then carries 1024 partials per group instead of 16384 (134 MB less to
    // write and to read back per 64-channel layer at sixteen faces)
    const int cpg = a.stats_cpg;
    double* const sred = reinterpret_cast<double*>(smem);  // [NW waves][32 channel pairs][2]
    __syncthreads();
    if (e_rh == 0) {
      sred[(wave * 32 + (lane & 31)) * 2] = psum;
      sred[(wave * 32 + (lane & 31)) * 2 + 1] = psq;
    }
    __syncthreads();
    if (e_tp == 0 && e_rh == 0 && (nn % cpg) == 0) {  // (wave 0, and wave 8 for the second 64 channels of the 16-wave form)
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) {
        s0 += sred[((wave + wv) * 32 + (lane & 31)) * 2];
        s1 += sred[((wave + wv) * 32 + (lane & 31)) * 2 + 1];
      }
      const int ng = a.cout / cpg;
      double* op = a.stats_out + (((size_t)b * ng + nn / cpg) * a.nparts + rt) * 2;
      op[0] = s0;
      op[1] = s1;
    }
  }
}

// U' = scale * G' g G'^T (fp64, rounded once to fp32) as hi = f16(U'), lo = f16(U' - hi), in MFMA-operand order
// [pos = xi*6 + nu][cin_pad/16][cout_pad/16][lane 64][4 words: hi, hi, lo, lo]; a lane's 16 bytes are the hi and the lo halves of
// U'[n = block*16 + (lane&15)][c = chunk*16 + (lane>>4)*4 + 0..3]  (v_mfma_f32_16x16x16_f16 B operand, hi | lo in one dwordx4).
__global__ void pack_weight_wf43_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, int nchunks, float scale,
                                        unsigned* __restrict__ packed, long total, int k32, int f32) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit word = two halves (or one fp32 value)
  if (i >= total) return;
  if (f32) {
    // fp32 operands (v_mfma_f32_16x16x4_f32): word j of a lane's 16 (32) bytes = U'[n = block*16 + (lane&15)][c = chunk*KS + 4 j + (lane>>4)]
    const int j = k32 ? (int)(i & 7) : (int)(i & 3), ln = k32 ? (int)((i >> 3) & 63) : (int)((i >> 2) & 63);
    long r = k32 ? i >> 9 : i >> 8;
    const int ntiles = cout_pad / 16;
    const int nn = (int)(r % ntiles) * 16 + (ln & 15);
    r /= ntiles;
    const int chunk = (int)(r % nchunks), pos = (int)(r / nchunks);
    const int c = chunk * (k32 ? 32 : CF_BK) + 4 * j + (ln >> 4);
    const int xi = pos / 6, nu = pos % 6;
    const double Gf[6][3] = {{4.0, 0.0, 0.0},           {-32.0 / 15.0, -16.0 / 15.0, -8.0 / 15.0}, {-32.0 / 15.0, 16.0 / 15.0, -8.0 / 15.0},
                             {1.0 / 15.0, 2.0 / 15.0, 4.0 / 15.0}, {1.0 / 15.0, -2.0 / 15.0, 4.0 / 15.0},  {0.0, 0.0, 4.0}};
    float val = 0.f;
    if (nn < cout && c < cin) {
      const float* g = w + ((long)nn * cin + c) * 9;
      double u = 0.0;
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        double rowv = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x) rowv += (double)g[y * 3 + x] * Gf[nu][x];
        u += Gf[xi][y] * rowv;
      }
      val = (float)u;
    }
    packed[i] = __builtin_bit_cast(unsigned, val);
    return;
  }
  // 16-channel slabs: a lane's 16 bytes = [hi k0..3 | lo k0..3]; 32-channel slabs (k32): a lane's 32 bytes = [hi k0..7 | lo k0..7]
  const int e = k32 ? (int)(i & 3) : (int)(i & 1), part = k32 ? (int)((i >> 2) & 1) : (int)((i >> 1) & 1);
  const int ln = k32 ? (int)((i >> 3) & 63) : (int)((i >> 2) & 63);
  long r = k32 ? i >> 9 : i >> 8;
  const int ntiles = cout_pad / 16;
  const int nn = (int)(r % ntiles) * 16 + (ln & 15);
  r /= ntiles;
  const int chunk = (int)(r % nchunks);
  const int pos = (int)(r / nchunks);
  const int xi = pos / 6, nu = pos % 6;
  // rows of G' = D^-1 G for the points (0, 1/2, -1/2, 2, -2, inf), D = diag(1/4, 1/4, 1/4, 1/2, 1/2, 1/4)
  const double Gm[6][3] = {{4.0, 0.0, 0.0},           {-32.0 / 15.0, -16.0 / 15.0, -8.0 / 15.0}, {-32.0 / 15.0, 16.0 / 15.0, -8.0 / 15.0},
                           {1.0 / 15.0, 2.0 / 15.0, 4.0 / 15.0}, {1.0 / 15.0, -2.0 / 15.0, 4.0 / 15.0},  {0.0, 0.0, 4.0}};
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = k32 ? chunk * 32 + (ln >> 4) * 8 + e * 2 + h : chunk * CF_BK + (ln >> 4) * 4 + e * 2 + h;
    float val = 0.f;
    if (nn < cout && c < cin) {
      const float* g = w + ((long)nn * cin + c) * 9;
      double u = 0.0;
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        double rowv = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x) rowv += (double)g[y * 3 + x] * Gm[nu][x];
        u += Gm[xi][y] * rowv;
      }
      val = (float)(u * (double)scale);
    }
    const _Float16 hi = (_Float16)val;
    const _Float16 hv = part ? (_Float16)(val - (float)hi) : hi;
    out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
  }
  packed[i] = out;
}

}  // namespace

// Form of the 16-wave (128-output-channel) workgroup; packing and launch agree through this one function:
//   1  32-channel slabs, two barrier intervals per slab (rounds 4: v_mfma_f32_16x16x32_f16)            CF_F43_WIDE=k32
//   0  16-channel slabs, two intervals (A/B only; also what cin % 32 != 0 runs in mode 1)                CF_F43_WIDE=k16
static int f4_wide_mode() {
  static const int v = [] {
    const char* w = getenv("CF_F43_WIDE");
    if (!w || !strcmp(w, "k32")) return 1;
    if (!strcmp(w, "k16")) return 0;
    // (an A/B switch read once per process: a mistyped value must not silently measure another form)
    fprintf(stderr, "libcodeformer_hip: CF_F43_WIDE=%s is not a form of the 16-wave F(4,3) workgroup (k32 | k16): using k32\n", w);
    return 1;
  }();
  return v;
}
static bool f4_k32_enabled() { return f4_wide_mode() == 1; }

extern "C" int cf_pack_conv_weight_winograd43_f16x2(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                                    cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_winograd43_f16x2: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_winograd43_f16x2: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  int ex = 0;
  CF_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "cf_pack_conv_weight_winograd43_f16x2: scale %g is not a power of two", (double)scale);
  const long total = 36L * cin_pad * cout_pad;  // 32-bit words: hi + lo half per weight
  // the layout follows the form cf_conv2d will run (cf_wf43_form below): 32-channel slabs for the 16-wave form where cin allows
  const int k32 = cout_pad % 128 == 0 && cin_pad % 32 == 0 && f4_k32_enabled();
  hipLaunchKernelGGL(pack_weight_wf43_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     cout_pad, cin_pad / (k32 ? 32 : CF_BK), scale, reinterpret_cast<unsigned*>(packed), total, k32, 0);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_winograd43_f16x2");
  return CF_OK;
}

extern "C" int cf_pack_conv_weight_winograd43(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed, cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_winograd43: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_winograd43: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  const long total = 36L * cin_pad * cout_pad;  // fp32 words
  const int k32 = cout_pad % 128 == 0 && cin_pad % 32 == 0 && f4_k32_enabled();   // (the rule of the split-half packing: the form cf_conv2d will run)
  hipLaunchKernelGGL(pack_weight_wf43_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     cout_pad, cin_pad / (k32 ? 32 : CF_BK), 1.f, reinterpret_cast<unsigned*>(packed), total, k32, 1);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_winograd43");
  return CF_OK;
}

// Called by cf_conv2d (cf_igemm.hip) for descriptors with winograd == 2; the common argument checks have run there.
int cf_wf43_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  CF_REQUIRE(d->taps == 9 && d->stride == 1 && !d->in_nchw && !d->out_nchw &&
                 (d->bf16_mfma == CF_OPERAND_F16X2 || d->bf16_mfma == CF_OPERAND_F32),
             "cf_conv2d(winograd 2): F(4x4,3x3) covers 3x3 stride-1 NHWC convolutions with split-half or fp32 operands");
  const bool f32 = d->bf16_mfma == CF_OPERAND_F32;
  // nearest-x2 + 3x3 (cf_conv_desc.upsample) with fp32 operands: the upsampling gather of the 16-wave form on 32-channel slabs (weights: the
  // plain cf_pack_conv_weight_winograd43 packing of the 3x3 kernel, not the folded one)
  CF_REQUIRE(!d->upsample || (f32 && d->prologue == CF_PRO_NONE && d->epilogue == CF_EPI_NONE && d->c1 == 0 && d->cout % 128 == 0 && d->c0 % 32 == 0 &&
                              f4_k32_enabled()),
             "cf_conv2d(winograd 2, upsample): fp32 operands, one input with c0 %% 32 == 0, cout %% 128 == 0, no prologue / epilogue operand");
  CF_REQUIRE(f32 || d->acc_scale > 0.f, "cf_conv2d(winograd 2): acc_scale must be the inverse of the pack-time weight scale (got %g)", (double)d->acc_scale);
  CF_REQUIRE(!f32 || !d->act_scale, "cf_conv2d(winograd 2): fp32 operands take no activation range scale");
  CF_REQUIRE(d->hout % F4_TH == 0 && d->wout % F4_TW == 0, "cf_conv2d(winograd 2): needs an output of %dx%d multiples (got %dx%d)", F4_TH,
             F4_TW, d->hout, d->wout);
  CF_REQUIRE(d->cout % 64 == 0 && d->cout_pad == d->cout, "cf_conv2d(winograd 2): needs cout == cout_pad, a multiple of 64 (got %d / %d)",
             d->cout, d->cout_pad);
  CF_REQUIRE((long)d->hout * d->wout <= (1L << 21), "cf_conv2d(winograd 2): at most 2^21 pixels per image (got %dx%d)", d->hout, d->wout);
  CF_REQUIRE((long)d->hout * d->wout * (d->c0 > d->cout ? d->c0 : d->cout) * 4 < (1L << 31) && (long)d->hout * d->wout * d->c1 * 4 < (1L << 31),
             "cf_conv2d(winograd 2): an image of a tensor must stay below 2^31 bytes (%dx%d, %d / %d / %d channels)", d->hout, d->wout, d->c0, d->c1, d->cout);
  const int tab_max = (d->cout % 128 == 0 && (d->c0 + d->c1) % 32 == 0 && f4_k32_enabled()) ? F4_TAB_32 : F4_TAB;   // (the 32-channel-slab form holds 512 GroupNorm rows)
  CF_REQUIRE(d->c0 + d->c1 <= tab_max, "cf_conv2d(winograd 2): at most %d input channels in this form (got %d)", tab_max, d->c0 + d->c1);
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT,
             "cf_conv2d(winograd 2): epilogues are none / residual / SFT");
  CF_REQUIRE(d->pad_mode == CF_PAD_ZERO && (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_in1 == 0 || d->ld_in1 == d->c1) &&
                 (d->ld_out == 0 || d->ld_out == d->cout) && d->split_k < 1,
             "cf_conv2d(winograd 2): reads / writes dense tensors with zero padding, no split_k");
  CF_REQUIRE(d->stats_cpg == 0 || (d->stats_cpg <= 64 && (d->stats_cpg & (d->stats_cpg - 1)) == 0 && d->stats_cpg >= 2),
             "cf_conv2d(winograd 2): stats_cpg %d (a power of two, 2..64)", d->stats_cpg);
  F4Args a;
  a.in0 = d->in0;
  a.in1 = d->in1;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.cin = d->c0 + d->c1;
  a.nchunks = a.cin / CF_BK;
  a.batch = d->batch;
  a.h = d->hout;
  a.w = d->wout;
  a.cout = d->cout;
  a.prologue = d->prologue;
  a.epilogue = d->epilogue;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.weight = d->weight;
  a.bias = d->bias;
  a.res = d->res;
  a.sft_scale = d->sft_scale;
  a.sft_w = d->sft_w;
  a.acc_scale = f32 ? 1.f : d->acc_scale;
  a.act_scale = d->act_scale;
  a.out = d->out;
  a.stats_out = d->stats_out;
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 2;
  a.tiles_x = d->wout / F4_TW;
  a.tiles_per_img = a.tiles_x * (d->hout / F4_TH);
  a.nparts = a.tiles_per_img;
  const bool wide = d->cout % 128 == 0;                      // 16 waves x 128 channels where the layer has them ...
  const bool k32 = wide && (d->c0 + d->c1) % 32 == 0 && f4_k32_enabled();   // ... on 32-channel slabs where cin allows: the rule the pack functions lay the weight out by
  CF_REQUIRE(!k32 || d->c0 % 32 == 0, "cf_conv2d(winograd 2): with cout %% 128 == 0 and cin %% 32 == 0 the concat boundary must be a multiple of 32 (c0 = %d)", d->c0);
  a.ntn = d->cout / (wide ? 128 : 64);
  a.nt_out = cf_nt_store((long)d->batch * d->hout * d->wout * d->cout * 4);
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  const size_t lds = (k32 ? F4_LDS_FLOATS_32 : wide ? F4_LDS_FLOATS_16 : F4_LDS_FLOATS) * sizeof(float);
  const dim3 grid(a.tiles_per_img * d->batch * a.ntn), block(wide ? 1024 : 512);
  // (cf_device_init sets the dynamic-LDS attribute of every instantiation on each device)
#define F4_LAUNCH_OP(P, E, O)                                                                                                 \
  do {                                                                                                                        \
    CF_LDS_ATTR((wf43_kernel<P, E, 8, 16, O>), F4_LDS_FLOATS * sizeof(float));                                            \
    CF_LDS_ATTR((wf43_kernel<P, E, 16, 16, O>), F4_LDS_FLOATS_16 * sizeof(float));                                            \
    CF_LDS_ATTR((wf43_kernel<P, E, 16, 32, O>), F4_LDS_FLOATS_32 * sizeof(float));                                            \
    if (k32) hipLaunchKernelGGL((wf43_kernel<P, E, 16, 32, O>), grid, block, lds, stream, a);                                 \
    else if (wide) hipLaunchKernelGGL((wf43_kernel<P, E, 16, 16, O>), grid, block, lds, stream, a);                           \
    else hipLaunchKernelGGL((wf43_kernel<P, E, 8, 16, O>), grid, block, lds, stream, a);                                      \
  } while (0)
#define F4_LAUNCH(P, E)                  \
  do {                                   \
    if (f32) F4_LAUNCH_OP(P, E, true);   \
    else F4_LAUNCH_OP(P, E, false);      \
  } while (0)
#define F4_LAUNCH_EPI(P)                                                           \
  do {                                                                             \
    if (d->epilogue == CF_EPI_RESIDUAL) F4_LAUNCH(P, CF_EPI_RESIDUAL);             \
    else if (d->epilogue == CF_EPI_SFT) F4_LAUNCH(P, CF_EPI_SFT);                  \
    else F4_LAUNCH(P, CF_EPI_NONE);                                                \
  } while (0)
  if (d->upsample) {
    CF_LDS_ATTR((wf43_kernel<CF_PRO_NONE, CF_EPI_NONE, 16, 32, true, true>), F4_LDS_FLOATS_32 * sizeof(float));
    hipLaunchKernelGGL((wf43_kernel<CF_PRO_NONE, CF_EPI_NONE, 16, 32, true, true>), grid, block, lds, stream, a);
    CF_CHECK_LAUNCH("cf_conv2d(winograd F(4,3) fp32, upsampling gather)");
    return CF_OK;
  }
  switch (d->prologue) {
    case CF_PRO_AFFINE: F4_LAUNCH_EPI(CF_PRO_AFFINE); break;
    case CF_PRO_AFFINE_SWISH: F4_LAUNCH_EPI(CF_PRO_AFFINE_SWISH); break;
    case CF_PRO_LEAKY: F4_LAUNCH_EPI(CF_PRO_LEAKY); break;
    default: F4_LAUNCH_EPI(CF_PRO_NONE); break;
  }
#undef F4_LAUNCH_EPI
#undef F4_LAUNCH
#undef F4_LAUNCH_OP
  CF_CHECK_LAUNCH("cf_conv2d(winograd F(4,3) f16x2)");
  return CF_OK;
}
