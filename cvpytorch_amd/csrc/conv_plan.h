// conv_plan.h — host-side planning for the implicit-GEMM convolution kernels (pure arithmetic; the
// CPU test-suite checks it against a numpy gather interpreter without needing a GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace cvhip {

constexpr int kMaxClasses = 16;  // stride_h * stride_w <= 16

struct IgemmClass {
  int TR, TS;            // taps visited by this class
  int dh0, dh_step;      // input row  = oh*in_sh + dh0 + i*dh_step   (i in [0,TR))
  int dw0, dw_step;      // input col  = ow*in_sw + dw0 + j*dw_step   (j in [0,TS))
  int out_oh, out_ow;    // output pixel = (oh*out_sh + out_oh, ow*out_sw + out_ow)
  int OHi, OWi;          // iteration grid of this class
  int M;                 // NB * OHi * OWi
  int tile_begin;        // first logical tile of this class (filled by the launcher)
  int64_t w_off;         // element offset of this class's weight image
  unsigned ts_magic;     // ceil(2^32 / TS) for the branch-free tap decode (0 when TS <= 1)
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;  // fast_div31 constants of OHi*OWi and OWi (pixel index -> image, row, column)
  // provenance of the taps in the original kernel (used by the weight packer)
  int r0, r_step, s0, s_step;
};

// scalar part of the kernel arguments, shared by the host-side plan (up to kMaxClasses classes) and the device-side argument
// struct (kKernelClasses classes: a by-value argument of ~1.4 KB made every launch drag a __amd_rocclr_copyBuffer node along
// under hipGraph replay — ~100 extra 2.4-us nodes per YOLOv5-s step; 4 classes cover stride <= 2, more classes launch in groups)
struct IgemmCommon {
  const h16_t* x;
  const h16_t* w;
  h16_t* y;
  const float* bias;
  int bias_n;  // number of valid bias entries (k_valid)
  float* stats;
  int stats_ld;   // accumulator pitch (stats_acc = 1): channels per accumulator row (>= Nout; a sibling pair's accumulator is wider)
  int stats_acc;  // 0: `stats` = fp32 partial rows [tile][2][Nout]; 1: `stats` is a double* accumulator [kAccShards][2][Nout] (atomics)
  int NB, IH, IW, Cin, x_ld;
  unsigned cin_magic;    // ceil(2^32 / Cin)
  int in_sh, in_sw;
  int Nout, y_ld, OH, OW, out_sh, out_sw;
  int n_tiles, total_tiles, ncls;
  int y_vec_ok;
  int staged_epilogue;  // 1: the implicit GEMM stores its output tile through the LDS (16-byte rows) when the geometry allows
  int interleave;  // 1: logical tile id = spatial tile * ncls + class (classes with equal tile counts: stride-parity dgrad)
                   // 2: the same with the class order rotated from one spatial tile to the next
                   // 3: groups of il_group spatial tiles, inside a group class-major in il_order (heaviest class first)
  int il_group, il_tiles, il_order;  // (3) group size, tiles per class, class order packed 4 bits per position
  const h16_t* res;  // optional addend, same pixel grid and channel count as y (dgrad: the gradient arriving over a skip connection)
  int res_ld;
  // "tail" (dgrad only): this launch produces dz, the gradient at the OUTPUT of a Conv-BN-act layer P (the layer whose activations
  // were this convolution's input). Its epilogue then also folds P's BatchNorm-backward sums (sum du, sum du * xhat with
  // du = dz * act'(scale * y + shift), xhat = (y - mean) * invstd) into P's accumulator (`stats` with stats_acc = 1), so that P's
  // backward needs no reduction pass over (dz, y). tail_y: P's raw convolution output, same pixel grid / channel count as `y` here.
  const h16_t* tail_y;
  int tail_y_ld;
  const float *tail_scale, *tail_shift, *tail_mean, *tail_invstd;
  int tail_act;
  float tail_ap;
  // fused EPILOGUE (every conv kernel): out = act((acc + bias) * ep_scale + ep_shift) — a folded / eval-mode BatchNorm and the
  // layer's activation in the convolution's own pass (conv_module.py:201-214 in eval mode, utils/fuse.py:32-54). ep_scale and
  // ep_shift come together or not at all; never combined with `stats` (training-mode sums are those of the raw accumulators).
  const float *ep_scale, *ep_shift;
  int ep_act;
  float ep_ap;
  int res_pre;  // with a fused epilogue: `res` joins BEFORE the activation — act(conv*scale + shift + res), the ResNet bottleneck tail
                // (relu(bn3(conv3) + identity)) — instead of after it (Darknet shortcut x + act(bn(conv)))
  // fused PROLOGUE (conv_patch.hip only): x is the RAW convolution output of the producing Conv-BN-act layer; the patch loader
  // applies act(pro_scale * x + pro_shift) per input channel on the way into the LDS and (z_out != NULL) stores the activated
  // tensor once, for the weight-gradient pass
  const float *pro_scale, *pro_shift;
  int pro_act;
  float pro_ap;
  h16_t* z_out;
  int z_ld;
  // (round 5) the streaming 1x1 kernel takes the same prologue (no z_out) on the input-channel range [pro_lo, pro_hi): the other
  // channels of x — slices of a concatenation that were materialised — pass through untouched
  int pro_lo, pro_hi;
  // (round 5) SPLIT STORE of the streaming 1x1 kernel (sibling pairs): output channels [y_split, Nout) go to y2 (pitch y2_ld) instead of
  // y — the second sibling's raw output lands straight in its channel slice of the concat buffer its consumer reads lazily
  h16_t* y2;
  int y2_ld, y_split;
  // image stems (conv_stem.hip only): the input is the dataloader's own tensor, fp32 NCHW [NB][x_planes][IH][IW] (x_planes <= 4 real
  // channels), read plane by plane and rounded to 16 bits on the way into the LDS patch — `x` is unused then
  const float* x_image;
  int x_planes;
  // the weight image carries the band kernel's fragment-ordered copy behind the row-major one (band_image_fprop / band_image_dgrad below)
  int band_image;
};

constexpr int kKernelClasses = 4;

template <int NC>
struct IgemmParamsT : IgemmCommon {
  IgemmClass cls[NC];
};
typedef IgemmParamsT<kMaxClasses> IgemmParams;      // host-side plan
typedef IgemmParamsT<kKernelClasses> IgemmKernArgs;  // what the kernels take by value

// kernel arguments for classes [first, first + count) of a plan (tile ranges re-based to the group)
inline IgemmKernArgs narrow_plan(const IgemmParams& p, int first, int count) {
  IgemmKernArgs k;
  static_cast<IgemmCommon&>(k) = static_cast<const IgemmCommon&>(p);
  k.ncls = count;
  for (int i = 0; i < kKernelClasses; ++i) k.cls[i] = p.cls[first + (i < count ? i : 0)];
  return k;
}

inline int conv_out_dim(int in, int pad, int dil, int k, int stride) {
  return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

// ---- band image (conv_band.hip) ----------------------------------------------------------------------------------------------
// The row-band kernel's waves fetch their weight fragments (16 output channels x 32 reduction elements of one tap) straight from
// global memory. In the row-major image Wt[n][tap * cin + c] such a fragment is 16 rows x 64 bytes — sixteen half cache lines per
// wave instruction, and the texture addresser, not the MFMA pipe, bounds the kernel (profiles/r05_band_coalesced_weight_probe.log:
// 256 -> 256 @20x20 44 -> 32 us with the same fetches made contiguous). So stride-1 3x3 layers whose channel counts the band kernel
// accepts get a SECOND copy of their weights, in fragment order, appended to the row-major image (element offset nrows * 9 * cin):
//   16-byte vector v = ((tap * (cin / 32) + c / 32) * (nrows / 16) + n / 16) * 64 + ((c % 32) / 8) * 16 + n % 16
// holds Wt[n][tap * cin + c .. c + 7] (c % 8 == 0): a fragment is 1 KB of contiguous memory, lane l of the MFMA operand at l * 16.
// nrows / cin: output / reduction channels of the GEMM (fprop: K / C of the layer; dgrad: C / K). Pure functions of the descriptor:
// the packers (weights_optim.hip), the allocation size (cvhip_conv2d_weight_image_elems) and the planners agree by construction.
inline bool band_image_shape(int nrows, int cin) { return (cin & 31) == 0 && (nrows == 32 || nrows == 64 || (nrows > 0 && (nrows & 127) == 0)); }
inline bool band_image_layer(const cvhip_conv_desc* d) {
  return d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && (d->k_valid <= 0 || d->k_valid == d->K) &&
         (d->c_valid <= 0 || d->c_valid == d->C);
}
// (round 6, second session) stride-2 3x3 / padding 1 / dilation 1 layers get the forward image too: the row-band kernel runs them with a
// patch of two column planes (conv_band.hip BandArgs::s2); their input gradient stays on the per-tap kernel's parity classes
inline bool band_image_layer_s2(const cvhip_conv_desc* d) {
  return d->R == 3 && d->S == 3 && d->stride_h == 2 && d->stride_w == 2 && d->pad_h == 1 && d->pad_w == 1 && d->dil_h == 1 && d->dil_w == 1 &&
         (d->k_valid <= 0 || d->k_valid == d->K) && (d->c_valid <= 0 || d->c_valid == d->C);
}
inline bool band_image_fprop(const cvhip_conv_desc* d) { return (band_image_layer(d) || band_image_layer_s2(d)) && band_image_shape(d->K, d->C); }
inline bool band_image_dgrad(const cvhip_conv_desc* d) { return band_image_layer(d) && band_image_shape(d->C, d->K); }
// vector v of a band image -> (row n, tap, first reduction channel c0)
__host__ __device__ __forceinline__ void band_image_decode(int64_t v, int nrows, int cin, int* n, int* tap, int* c0) {
  const int lane = (int)(v & 63);
  const int64_t fr = v >> 6;
  const int nf = nrows >> 4;
  const int f = (int)(fr % nf);
  const int st = (int)(fr / nf);
  const int nc = cin >> 5;
  *tap = st / nc;
  *c0 = (st - *tap * nc) * 32 + (lane >> 4) * 8;
  *n = f * 16 + (lane & 15);
}

// exact n / d for n, d < 2^16 as __umulhi(n, magic), magic = ceil(2^32 / d)  (d == 1 -> magic 0: caller returns n)
inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// exact n / d for 0 <= n < 2^31 and any d >= 1:  q = umulhi(n, mul) >> sh  (d == 1: mul = 0, the caller returns n).
// mul = ceil(2^(31 + L) / d), sh = L - 1 with L = ceil(log2 d): the classic round-up multiplier, 32 bits wide because d > 2^(L-1).
inline void div31_consts(int d, unsigned* mul, unsigned* sh) {
  if (d <= 1) {
    *mul = 0;
    *sh = 0;
    return;
  }
  int L = 0;
  while ((1ll << L) < d) ++L;
  const int pw = 31 + L;
  *mul = (unsigned)(((1ull << pw) + (unsigned long long)d - 1) / (unsigned long long)d);
  *sh = (unsigned)(L - 1);
}
__host__ __device__ __forceinline__ unsigned fast_div31(unsigned n, unsigned mul, unsigned sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned q = __umulhi(n, mul) >> sh;
#else
  const unsigned q = (unsigned)(((unsigned long long)n * mul) >> 32) >> sh;
#endif
  return mul ? q : n;
}

inline int gcd_(int a, int b) {
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// 1-D tap progression for dgrad parity `ph`: taps r with (ph + pad - r*dil) % stride == 0.
// Returns count; r = r0 + i*r_step; d(i) = (ph + pad - r*dil)/stride = d0 + i*d_step.
inline int dgrad_taps_1d(int ph, int pad, int dil, int stride, int R, int* r0, int* r_step, int* d0,
                         int* d_step) {
  const int g = gcd_(dil, stride);
  const int step = stride / g;
  int first = -1;
  for (int r = 0; r < R && r < step; ++r) {
    int v = ph + pad - r * dil;
    if (((v % stride) + stride) % stride == 0) {
      first = r;
      break;
    }
  }
  *r_step = step;
  *d_step = -(dil / g);
  if (first < 0) {
    *r0 = 0;
    *d0 = 0;
    return 0;
  }
  *r0 = first;
  // floor division is exact here
  *d0 = (ph + pad - first * dil) / stride;
  return (R - 1 - first) / step + 1;
}

int validate_dense_desc(const cvhip_conv_desc* d);
void plan_fprop(const cvhip_conv_desc* d, IgemmParams* p);
// returns number of classes (<= kMaxClasses) or negative status
int plan_dgrad(const cvhip_conv_desc* d, IgemmParams* p);
int igemm_block_m(int Nout, int64_t M, int Ktot);
int launch_igemm(IgemmParams& p, hipStream_t stream);
// conv1x1_stream.hip: grid size of the streaming 1x1 kernel (0 = the general kernel runs) and its launcher (-1 = not taken)
int stream1x1_blocks(int Nout, int Cin, int64_t M, bool stats);
int try_launch_stream1x1(const IgemmParams& p, hipStream_t stream);
bool stream1x1_prologue_ok(const IgemmParams& p, bool stats);
// conv_stem.hip: direct convolution for 8-channel image stems (grid size, 0 = not taken; launcher, -1 = not taken)
int stem_blocks(int C, int x_ld, int K, int R, int S, int sh, int sw, int dh, int dw, int N, int OH, int OW);
int try_launch_stem(const IgemmParams& p, hipStream_t stream);
// BN + activation backward applied on load by the stem weight-gradient kernel (`dy` is then dz, the gradient at the layer's output)
struct StemWgradBn {
  const void* y;
  int y_ld;
  const float *scale, *shift, *mean, *invstd;
  const double* acc;
  int acc_ld;
  int act;
  float act_param;
  float *dgamma_out, *dbeta_out;
  int accumulate;
};
int try_launch_stem_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream, const float* x_image = nullptr,
                          int x_planes = 0, const StemWgradBn* bn = nullptr);
// conv_patch.hip: patch-resident implicit GEMM for multi-tap convolutions (launcher, -1 = not taken; geometry-only query with the
// number of BatchNorm partial rows its epilogue writes)
int try_launch_patch(const IgemmParams& p, hipStream_t stream);
bool patch_takes(const IgemmParams& p, int* stats_rows);
// per class CVHIP_PATCH_CLASS_INTS int32 of tile / patch geometry (cvhip_conv2d_patch_plan); returns the class count, 0 = not taken
// conv_band.hip: row-band 3x3 stride-1 implicit GEMM with register-resident weight fragments (launcher, -1 = not taken)
int try_launch_band(const IgemmParams& p, hipStream_t stream);
int band_plan_export(const IgemmParams& p, int32_t* out);  // 1 = the band kernel runs this plan (out: CVHIP_BAND_PLAN_INTS values), 0 = another kernel
int patch_plan_export(const IgemmParams& p, int32_t* out, int max_classes, bool any_geometry);

}  // namespace cvhip
