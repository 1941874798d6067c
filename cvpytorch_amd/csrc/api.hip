// api.hip — extern "C" entry points for convolution (+ host planning), error plumbing and the
// hardware-layout probes. See include/cvhip.h for the contract of every symbol.
#include <string.h>

#include <string>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

static thread_local std::string g_last_error = "";

void set_last_error(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(what, e);
    return CVHIP_ERR_LAUNCH;
  }
  return CVHIP_OK;
}

__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* p, size_t n_words, int tail_half) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) p[i] = 0u;
  if (tail_half && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint16_t*>(p)[n_words * 2] = 0;
}

__global__ __launch_bounds__(256) void unpad_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int C, int Cv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t kt = i / Cv;
    const int c = (int)(i - kt * Cv);
    dst[i] += src[kt * C + c];
  }
}

int zero_fill(void* ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return CVHIP_OK;
  if ((((uintptr_t)ptr) & 3) || (bytes & 1)) return CVHIP_ERR_INVALID;  // whole 32-bit words (+ one trailing bf16)
  const size_t n = bytes / 4;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t*)ptr, n, (int)((bytes & 2) != 0));
  return check_launch("zero_fill_kernel");
}

int validate_dense_desc(const cvhip_conv_desc* d) {
  if (!d) return CVHIP_ERR_INVALID;
  if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0) return CVHIP_ERR_INVALID;
  if (d->stride_h <= 0 || d->stride_w <= 0 || d->dil_h <= 0 || d->dil_w <= 0 || d->pad_h < 0 || d->pad_w < 0)
    return CVHIP_ERR_INVALID;
  if (d->x_ld < d->C || d->y_ld < d->K) return CVHIP_ERR_INVALID;
  if (d->k_valid < 0 || d->k_valid > d->K || d->c_valid < 0 || d->c_valid > d->C) return CVHIP_ERR_INVALID;
  if (d->groups != 1) return CVHIP_ERR_UNSUPPORTED;
  // 16-byte channel vectors on the gathered operand: input channels and pitches multiple of 8
  if ((d->C & 7) || (d->x_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  if (d->stride_h * d->stride_w > kMaxClasses) return CVHIP_ERR_UNSUPPORTED;
  const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  if (P <= 0 || Q <= 0) return CVHIP_ERR_INVALID;
  // 32-bit pixel indexing inside the kernels
  if ((int64_t)d->N * d->H * d->W >= (1ll << 31) || (int64_t)d->N * P * Q >= (1ll << 31)) return CVHIP_ERR_UNSUPPORTED;
  return CVHIP_OK;
}

void plan_fprop(const cvhip_conv_desc* d, IgemmParams* p) {
  memset(p, 0, sizeof(*p));
  const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  p->NB = d->N;
  p->IH = d->H;
  p->IW = d->W;
  p->Cin = d->C;
  p->x_ld = d->x_ld;
  p->in_sh = d->stride_h;
  p->in_sw = d->stride_w;
  p->Nout = d->K;
  p->y_ld = d->y_ld;
  p->OH = P;
  p->OW = Q;
  p->out_sh = 1;
  p->out_sw = 1;
  p->ncls = 1;
  IgemmClass& c = p->cls[0];
  c.TR = d->R;
  c.TS = d->S;
  c.dh0 = -d->pad_h;
  c.dh_step = d->dil_h;
  c.dw0 = -d->pad_w;
  c.dw_step = d->dil_w;
  c.out_oh = 0;
  c.out_ow = 0;
  c.OHi = P;
  c.OWi = Q;
  c.M = d->N * P * Q;
  c.w_off = 0;
  c.r0 = 0;
  c.r_step = 1;
  c.s0 = 0;
  c.s_step = 1;
  p->band_image = band_image_fprop(d) ? 1 : 0;
}

int plan_dgrad(const cvhip_conv_desc* d, IgemmParams* p) {
  memset(p, 0, sizeof(*p));
  const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  p->NB = d->N;
  p->IH = P;  // the gathered operand is dy
  p->IW = Q;
  p->Cin = d->K;
  p->x_ld = d->y_ld;
  p->in_sh = 1;
  p->in_sw = 1;
  p->Nout = d->C;
  p->y_ld = d->x_ld;
  p->OH = d->H;
  p->OW = d->W;
  p->out_sh = d->stride_h;
  p->out_sw = d->stride_w;
  int n = 0;
  int64_t woff = 0;
  for (int ph = 0; ph < d->stride_h; ++ph) {
    for (int pw = 0; pw < d->stride_w; ++pw) {
      IgemmClass& c = p->cls[n];
      c.TR = dgrad_taps_1d(ph, d->pad_h, d->dil_h, d->stride_h, d->R, &c.r0, &c.r_step, &c.dh0, &c.dh_step);
      c.TS = dgrad_taps_1d(pw, d->pad_w, d->dil_w, d->stride_w, d->S, &c.s0, &c.s_step, &c.dw0, &c.dw_step);
      if (c.TR == 0 || c.TS == 0) {  // no tap reaches this parity: the class just writes zeros
        c.TR = 0;
        c.TS = 0;
      }
      c.out_oh = ph;
      c.out_ow = pw;
      c.OHi = ph < d->H ? (d->H - ph + d->stride_h - 1) / d->stride_h : 0;
      c.OWi = pw < d->W ? (d->W - pw + d->stride_w - 1) / d->stride_w : 0;
      c.M = d->N * c.OHi * c.OWi;
      c.w_off = woff;
      woff += (int64_t)d->C * c.TR * c.TS * d->K;
      ++n;
    }
  }
  p->ncls = n;
  p->band_image = band_image_dgrad(d) ? 1 : 0;
  return n;
}

int pack_weights(const cvhip_conv_desc* d, const float* master, void* w_fprop, void* w_dgrad, hipStream_t stream);
int launch_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream);
int launch_wgrad_impl(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream, float* det_ws, int64_t* det_ws_floats,
                      int det_accumulate);
int wgband_plan_export(const cvhip_conv_desc* d, int32_t* out);

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_version(void) { return CVHIP_VERSION; }
const char* cvhip_last_error(void) { return g_last_error.c_str(); }

int cvhip_conv2d_out_hw(const cvhip_conv_desc* d, int32_t* P, int32_t* Q) {
  if (!d || !P || !Q) return CVHIP_ERR_INVALID;
  *P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  *Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  return (*P > 0 && *Q > 0) ? CVHIP_OK : CVHIP_ERR_INVALID;
}

int cvhip_conv2d_fprop_stats_rows(const cvhip_conv_desc* d) {
  int st = validate_dense_desc(d);
  if (st) return st;
  const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  {
    const int blocks = stem_blocks(d->C, d->x_ld, d->K, d->R, d->S, d->stride_h, d->stride_w, d->dil_h, d->dil_w, d->N, P, Q);
    if (blocks > 0) return blocks;  // same decision as launch_igemm
  }
  if (d->R == 1 && d->S == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 && (d->x_ld & 7) == 0) {
    const int blocks = stream1x1_blocks(d->K, d->C, (int64_t)d->N * P * Q, true);  // same decision as launch_igemm
    if (blocks > 0) return blocks;
  }
  {
    IgemmParams p;
    plan_fprop(d, &p);
    int rows = 0;
    if (patch_takes(p, &rows)) return rows;  // same decision as launch_igemm (conv_patch.hip: one row per spatial tile)
  }
  return cdiv(d->N * P * Q, igemm_block_m(d->K, (int64_t)d->N * P * Q, d->R * d->S * d->C));
}

int cvhip_conv2d_patch_plan(const cvhip_conv_desc* d, int flags, int32_t* out, int max_classes) {
  int st = validate_dense_desc(d);
  if (st) return st;
  IgemmParams p;
  if (flags & 1) {
    if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
    plan_dgrad(d, &p);
  } else {
    plan_fprop(d, &p);
  }
  return patch_plan_export(p, out, max_classes, (flags & 2) != 0);
}

int cvhip_conv2d_band_plan(const cvhip_conv_desc* d, int flags, int32_t* out) {
  int st = validate_dense_desc(d);
  if (st) return st;
  IgemmParams p;
  if (flags & 1) {
    if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
    plan_dgrad(d, &p);
  } else {
    plan_fprop(d, &p);
  }
  // (operand pointers are not part of the descriptor: the query assumes the 16-byte alignment every arena tensor has; the training
  // form's BatchNorm sums go to the fp64 accumulator, which the band kernel supports)
  return band_plan_export(p, out);
}

int cvhip_conv2d_wgrad_band_plan(const cvhip_conv_desc* d, int32_t* out) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  return wgband_plan_export(d, out);
}

int cvhip_conv_stem_blocks(const cvhip_conv_desc* d) {
  int st = validate_dense_desc(d);
  if (st) return st;
  const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  return stem_blocks(d->C, d->x_ld, d->K, d->R, d->S, d->stride_h, d->stride_w, d->dil_h, d->dil_w, d->N, P, Q);
}

int cvhip_conv1x1_stream_blocks(int nout, int cin, int64_t m, int with_stats) { return stream1x1_blocks(nout, cin, m, with_stats != 0); }

int64_t cvhip_conv2d_dgrad_weight_elems(const cvhip_conv_desc* d) {
  int st = validate_dense_desc(d);
  if (st) return st;
  IgemmParams p;
  const int n = plan_dgrad(d, &p);
  int64_t e = 0;
  for (int i = 0; i < n; ++i) e += (int64_t)d->C * p.cls[i].TR * p.cls[i].TS * d->K;
  return e;
}

int64_t cvhip_conv2d_weight_image_elems(const cvhip_conv_desc* d, int which) {
  int st = validate_dense_desc(d);
  if (st) return st;
  const int64_t krsc = (int64_t)d->K * d->R * d->S * d->C;
  if (which == 0) return krsc + (band_image_fprop(d) ? krsc : 0);
  if (which != 1) return CVHIP_ERR_INVALID;
  const int64_t e = cvhip_conv2d_dgrad_weight_elems(d);
  return e < 0 ? e : e + (band_image_dgrad(d) ? krsc : 0);
}

int cvhip_div31_consts(int32_t d, uint32_t* mul, uint32_t* shift) {
  if (d < 1 || !mul || !shift) return CVHIP_ERR_INVALID;
  div31_consts(d, mul, shift);
  return CVHIP_OK;
}

int cvhip_conv2d_dgrad_plan(const cvhip_conv_desc* d, int32_t* out, int max_classes) {
  int st = validate_dense_desc(d);
  if (st) return st;
  IgemmParams p;
  const int n = plan_dgrad(d, &p);
  if (!out || max_classes < n) return CVHIP_ERR_INVALID;
  for (int i = 0; i < n; ++i) {
    const IgemmClass& c = p.cls[i];
    int32_t* o = out + i * CVHIP_DGRAD_CLASS_INTS;
    o[0] = c.TR;
    o[1] = c.TS;
    o[2] = c.r0;
    o[3] = c.r_step;
    o[4] = c.dh0;
    o[5] = c.dh_step;
    o[6] = c.s0;
    o[7] = c.s_step;
    o[8] = c.dw0;
    o[9] = c.dw_step;
    o[10] = (int32_t)(c.w_off & 0xffffffffll);
    o[11] = (int32_t)(c.w_off >> 32);
  }
  return n;
}

int cvhip_conv2d_prep_weights(const cvhip_conv_desc* d, const float* w_master, void* w_fprop, void* w_dgrad, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!w_master || (!w_fprop && !w_dgrad)) return CVHIP_ERR_INVALID;
  if ((((uintptr_t)w_master) & 3) || (((uintptr_t)w_fprop) & 15) || (((uintptr_t)w_dgrad) & 15)) return CVHIP_ERR_INVALID;
  return pack_weights(d, w_master, w_fprop, w_dgrad, (hipStream_t)stream);
}

int cvhip_conv2d_fprop(const cvhip_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                       float* stats_partial, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!x || !w || !y) return CVHIP_ERR_INVALID;
  if (stats_partial && bias) return CVHIP_ERR_INVALID;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)w) & 15)) return CVHIP_ERR_INVALID;
  IgemmParams p;
  plan_fprop(d, &p);
  p.x = (const h16_t*)x;
  p.w = (const h16_t*)w;
  p.y = (h16_t*)y;
  p.bias = bias;
  p.bias_n = d->k_valid > 0 ? d->k_valid : d->K;
  p.stats = stats_partial;
  p.stats_acc = 0;
  p.stats_ld = d->K;
  p.tail_y = nullptr;
  p.y_vec_ok = ((d->y_ld & 3) == 0) && ((((uintptr_t)y) & 7) == 0);
  return launch_igemm(p, (hipStream_t)stream);
}

int cvhip_conv2d_fprop_acc(const cvhip_conv_desc* d, const void* x, const void* w, void* y, double* bn_acc, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!x || !w || !y || !bn_acc) return CVHIP_ERR_INVALID;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)w) & 15) || (((uintptr_t)bn_acc) & 7)) return CVHIP_ERR_INVALID;
  IgemmParams p;
  plan_fprop(d, &p);
  p.x = (const h16_t*)x;
  p.w = (const h16_t*)w;
  p.y = (h16_t*)y;
  p.bias = nullptr;
  p.bias_n = d->k_valid > 0 ? d->k_valid : d->K;
  p.stats = reinterpret_cast<float*>(bn_acc);
  p.stats_acc = 1;
  p.stats_ld = d->K;
  p.tail_y = nullptr;
  p.y_vec_ok = ((d->y_ld & 3) == 0) && ((((uintptr_t)y) & 7) == 0);
  return launch_igemm(p, (hipStream_t)stream);
}

int cvhip_conv2d_fprop_fused(const cvhip_conv_desc* d, const void* x, const void* w, void* y, const cvhip_conv_fuse* f, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!w || !y || !f || (!x && !f->x_image)) return CVHIP_ERR_INVALID;
  if ((!f->x_image && (((uintptr_t)x) & 15)) || (((uintptr_t)w) & 15)) return CVHIP_ERR_INVALID;
  if (f->x_image && (f->x_image_planes < 1 || f->x_image_planes > 4 || f->pro_scale || f->residual)) return CVHIP_ERR_INVALID;
  if (f->stats_partial && f->bn_acc) return CVHIP_ERR_INVALID;
  if ((f->stats_partial || f->bn_acc) && (f->bias || f->ep_scale || f->ep_act != CVHIP_ACT_NONE)) return CVHIP_ERR_INVALID;
  if ((f->ep_scale == nullptr) != (f->ep_shift == nullptr) || (f->pro_scale == nullptr) != (f->pro_shift == nullptr)) return CVHIP_ERR_INVALID;
  if (f->z_out && !f->pro_scale) return CVHIP_ERR_INVALID;
  if (f->pro_lo || f->pro_hi) {  // a partial prologue range: 16-byte channel vectors inside the input
    const int hi = f->pro_hi > 0 ? f->pro_hi : d->C;
    if (!f->pro_scale || f->pro_lo < 0 || hi > d->C || f->pro_lo >= hi || (f->pro_lo & 7) || (hi & 7)) return CVHIP_ERR_INVALID;
  }
  if (f->bn_acc && (((uintptr_t)f->bn_acc) & 7)) return CVHIP_ERR_INVALID;
  IgemmParams p;
  plan_fprop(d, &p);
  p.x = (const h16_t*)x;
  p.w = (const h16_t*)w;
  p.y = (h16_t*)y;
  p.bias = f->bias;
  p.bias_n = d->k_valid > 0 ? d->k_valid : d->K;
  p.stats = f->bn_acc ? reinterpret_cast<float*>(f->bn_acc) : f->stats_partial;
  p.stats_acc = f->bn_acc ? 1 : 0;
  p.stats_ld = d->K;
  p.tail_y = nullptr;
  p.y_vec_ok = ((d->y_ld & 3) == 0) && ((((uintptr_t)y) & 7) == 0);
  p.ep_scale = f->ep_scale;
  p.ep_shift = f->ep_shift;
  p.ep_act = f->ep_act;
  p.ep_ap = f->ep_act_param;
  p.pro_scale = f->pro_scale;
  p.pro_shift = f->pro_shift;
  p.pro_act = f->pro_act;
  p.pro_ap = f->pro_act_param;
  p.z_out = (h16_t*)f->z_out;
  p.z_ld = f->z_ld;
  p.pro_lo = f->pro_lo;
  p.pro_hi = f->pro_hi > 0 ? f->pro_hi : d->C;
  if (f->y2) {
    if (f->residual || f->z_out || f->x_image) return CVHIP_ERR_INVALID;
    // two non-empty halves made of whole 16-byte channel vectors, both destinations vector-aligned
    if (f->y_split <= 0 || f->y_split >= d->K || (f->y_split & 7) || (d->K & 7) || (f->y2_ld & 7) || f->y2_ld < d->K - f->y_split ||
        (((uintptr_t)f->y2) & 15) || (((uintptr_t)y) & 15) || (d->y_ld & 7))
      return CVHIP_ERR_INVALID;
    p.y2 = (h16_t*)f->y2;
    p.y2_ld = f->y2_ld;
    p.y_split = f->y_split;
  }
  if (f->residual) {
    if (f->residual_ld < d->K) return CVHIP_ERR_INVALID;
    p.res = (const h16_t*)f->residual;
    p.res_ld = f->residual_ld;
    p.res_pre = f->residual_pre ? 1 : 0;
    // (the residual-before-activation form lives in the fused-epilogue instances: request them even for a bare residual add)
    if (p.res_pre && !p.ep_scale && p.ep_act == CVHIP_ACT_NONE) p.res_pre = 0;
  }
  if (f->x_image) {  // fp32 NCHW image input: only the image-stem kernel reads it (cvhip_conv_stem_blocks(d) > 0 says whether it runs this d)
    p.x = nullptr;
    p.x_image = f->x_image;
    p.x_planes = f->x_image_planes;
    const int rc = try_launch_stem(p, (hipStream_t)stream);
    return rc >= 0 || rc < -1 ? rc : CVHIP_ERR_UNSUPPORTED;
  }
  return launch_igemm(p, (hipStream_t)stream);
}

int cvhip_conv2d_wgrad_image(const cvhip_conv_desc* d, const float* x_nchw, int32_t planes, const void* dy, float* dw, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!x_nchw || !dy || !dw || planes < 1 || planes > 4) return CVHIP_ERR_INVALID;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  const int rc = try_launch_stem_wgrad(d, nullptr, dy, dw, (hipStream_t)stream, x_nchw, planes);
  return rc >= 0 || rc < -1 ? rc : CVHIP_ERR_UNSUPPORTED;
}

int cvhip_conv2d_wgrad_stem_bn(const cvhip_conv_desc* d, const void* x, const float* x_nchw, int32_t planes, const void* dz, const void* y,
                               const float* scale, const float* shift, const float* mean, const float* invstd, const double* acc, int32_t acc_ld,
                               float* dgamma_out, float* dbeta_out, int32_t accumulate, int32_t act, float act_param, float* dw, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if ((!x && !x_nchw) || !dz || !y || !dw || !scale || !shift || !mean || !invstd || !acc) return CVHIP_ERR_INVALID;
  if (x_nchw && (planes < 1 || planes > 4)) return CVHIP_ERR_INVALID;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  if (act != CVHIP_ACT_NONE && act != CVHIP_ACT_RELU && act != CVHIP_ACT_LEAKY && act != CVHIP_ACT_SILU) return CVHIP_ERR_UNSUPPORTED;
  StemWgradBn bn{y, d->y_ld, scale, shift, mean, invstd, acc, acc_ld, act, act_param, dgamma_out, dbeta_out, accumulate};
  const int rc = try_launch_stem_wgrad(d, x, dz, dw, (hipStream_t)stream, x_nchw, planes, &bn);
  return rc >= 0 || rc < -1 ? rc : CVHIP_ERR_UNSUPPORTED;
}

int cvhip_conv2d_fprop_prologue_ok(const cvhip_conv_desc* d, int with_z_out) {
  if (validate_dense_desc(d)) return 0;
  IgemmParams p;
  plan_fprop(d, &p);
  static const float one = 1.f;
  p.pro_scale = p.pro_shift = &one;  // "a prologue is requested": geometry decides, not the default speed policy
  if (!patch_takes(p, nullptr) || d->C > 768) return 0;
  if (with_z_out) {
    const int P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
    const int Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
    if (P != d->H || Q != d->W || d->stride_h != 1 || d->stride_w != 1) return 0;
  }
  return 1;
}

int cvhip_conv1x1_stream_prologue_ok(const cvhip_conv_desc* d, int with_stats) {
  if (validate_dense_desc(d)) return 0;
  IgemmParams p;
  plan_fprop(d, &p);
  return stream1x1_prologue_ok(p, with_stats != 0) ? 1 : 0;
}

static int dgrad_impl(const cvhip_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, int addend_ld, void* dx, void* stream,
                      const cvhip_bn_tail* tail = nullptr) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!dy || !w_dgrad || !dx) return CVHIP_ERR_INVALID;
  // the gathered operand is dy: its channel count / pitch must be 16-byte vectorisable
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  if ((((uintptr_t)dy) & 15) || (((uintptr_t)w_dgrad) & 15)) return CVHIP_ERR_INVALID;
  if (addend && addend_ld < d->C) return CVHIP_ERR_INVALID;
  IgemmParams p;
  plan_dgrad(d, &p);
  p.x = (const h16_t*)dy;
  p.w = (const h16_t*)w_dgrad;
  p.y = (h16_t*)dx;
  p.bias = nullptr;
  p.stats = nullptr;
  p.stats_acc = 0;
  p.stats_ld = d->C;
  p.res = (const h16_t*)addend;
  p.res_ld = addend_ld;
  p.y_vec_ok = ((d->x_ld & 3) == 0) && ((((uintptr_t)dx) & 7) == 0);
  p.tail_y = nullptr;
  if (tail) {
    if (!tail->y || !tail->scale || !tail->shift || !tail->mean || !tail->invstd || !tail->acc || tail->acc_ld < d->C) return CVHIP_ERR_INVALID;
    // every kernel's tail code sits on its packed-store path: 8-channel groups, 16-byte aligned rows of dx and y
    if ((d->C & 7) || (d->x_ld & 7) || (tail->y_ld & 7) || (((uintptr_t)dx) & 15) || (((uintptr_t)tail->y) & 15)) return CVHIP_ERR_UNSUPPORTED;
    if (tail->act != CVHIP_ACT_NONE && tail->act != CVHIP_ACT_RELU && tail->act != CVHIP_ACT_LEAKY && tail->act != CVHIP_ACT_SILU) return CVHIP_ERR_UNSUPPORTED;
    p.tail_y = (const h16_t*)tail->y;
    p.tail_y_ld = tail->y_ld;
    p.tail_scale = tail->scale;
    p.tail_shift = tail->shift;
    p.tail_mean = tail->mean;
    p.tail_invstd = tail->invstd;
    p.tail_act = tail->act;
    p.tail_ap = tail->act_param;
    p.stats = reinterpret_cast<float*>(tail->acc);
    p.stats_acc = 1;
    p.stats_ld = tail->acc_ld;
  }
  return launch_igemm(p, (hipStream_t)stream);
}

int cvhip_conv2d_dgrad_tail(const cvhip_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, int32_t addend_ld, void* dx,
                            const cvhip_bn_tail* tail, void* stream) {
  if (!tail) return CVHIP_ERR_INVALID;
  return dgrad_impl(d, dy, w_dgrad, addend, addend_ld, dx, stream, tail);
}

int cvhip_conv2d_dgrad(const cvhip_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, void* stream) {
  return dgrad_impl(d, dy, w_dgrad, nullptr, 0, dx, stream);
}

int cvhip_conv2d_dgrad_add(const cvhip_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, int32_t addend_ld, void* dx,
                           void* stream) {
  if (!addend) return CVHIP_ERR_INVALID;
  return dgrad_impl(d, dy, w_dgrad, addend, addend_ld, dx, stream);
}

int cvhip_conv2d_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!x || !dy || !dw) return CVHIP_ERR_INVALID;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15)) return CVHIP_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    int zs = zero_fill(dw, sizeof(float) * (size_t)d->K * d->R * d->S * d->C, s);
    if (zs) return zs;
  }
  return launch_wgrad(d, x, dy, dw, s);
}

int64_t cvhip_conv2d_wgrad_det_workspace_bytes(const cvhip_conv_desc* d) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  int64_t floats = -1;  // size query: no launch
  float dummy = 0.f;
  st = launch_wgrad_impl(d, &dummy, &dummy, &dummy, nullptr, &dummy, &floats, 0);
  if (st) return st;
  return floats * (int64_t)sizeof(float);
}

int cvhip_conv2d_wgrad_det(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* workspace, int64_t ws_bytes,
                           void* stream) {
  int st = validate_dense_desc(d);
  if (st) return st;
  if (!x || !dy || !dw || !workspace) return CVHIP_ERR_INVALID;
  if ((d->K & 7) || (d->y_ld & 7)) return CVHIP_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return CVHIP_ERR_INVALID;
  int64_t floats = ws_bytes / (int64_t)sizeof(float);
  return launch_wgrad_impl(d, x, dy, dw, (hipStream_t)stream, (float*)workspace, &floats, accumulate ? 1 : 0);
}

int cvhip_zero_fill(void* ptr, int64_t bytes, void* stream) {
  if (!ptr || bytes < 0) return CVHIP_ERR_INVALID;
  return zero_fill(ptr, (size_t)bytes, (hipStream_t)stream);
}

int cvhip_f32_unpad_add(const float* src, float* dst, int32_t K_valid, int32_t T, int32_t C, int32_t C_valid, void* stream) {
  if (!src || !dst || K_valid <= 0 || T <= 0 || C <= 0 || C_valid <= 0 || C_valid > C) return CVHIP_ERR_INVALID;
  const int64_t n = (int64_t)K_valid * T * C_valid;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpad_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n, C, C_valid);
  return check_launch("unpad_add_kernel");
}

}  // extern "C"
