// dwconv.hip — depthwise convolution (groups == C == K), direct form. One multiply-add per tap per
// channel: arithmetic intensity ~ R*S/2 flop/byte, far below machine balance, so these are plain
// HBM-bound streaming kernels (16-B channel vectors, fp32 weights held in registers) — no MFMA.
//
// Reference: src/models/bricks/depthwise_separable_conv_module.py:76-94 as used by
// src/models/heads/seg/deeplabv3plus_head.py:18-30,49-54 (3x3, dilation 1/12/24/36).
#include <string.h>
#include <stdlib.h>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

// fused activation of the depthwise forward kernels (inference: ConvModule's act after a folded BatchNorm, utils/fuse.py:32-54):
// one block-uniform switch per 8-channel vector; act == NONE (every training launch) costs one scalar compare
__device__ __forceinline__ void dw_act8(f32x8& o, int act, float ap) {
  if (act == CVHIP_ACT_NONE) return;
  switch (act) {
    case CVHIP_ACT_RELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = act_fwd(o.v[j], CVHIP_ACT_RELU, ap);
      break;
    case CVHIP_ACT_SILU:
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = act_fwd(o.v[j], CVHIP_ACT_SILU, ap);
      break;
    case CVHIP_ACT_LEAKY:
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = act_fwd(o.v[j], CVHIP_ACT_LEAKY, ap);
      break;
    case CVHIP_ACT_SIGMOID:
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = act_fwd(o.v[j], CVHIP_ACT_SIGMOID, ap);
      break;
    default:
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = act_fwd(o.v[j], CVHIP_ACT_HSWISH, ap);
      break;
  }
}


struct DwParams {
  const h16_t* x;   // fprop: input; dgrad: dy; wgrad: x
  const h16_t* dy;  // wgrad only
  const float* w;    // [C][R][S]
  const float* bias;
  h16_t* y;         // fprop: y; dgrad: dx
  float* dw;
  int N, C, H, W, P, Q, R, S, sh, sw, ph, pw, dh, dw_;
  int x_ld, y_ld;
  int act;    // fprop: activation applied to the result (CVHIP_ACT_NONE in training; memset by fill())
  float ap;
};

__host__ __device__ __forceinline__ bool dw_vec_ok(const DwParams& p) {
  return (p.C & 7) == 0 && (p.x_ld & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.dy)) & 15) == 0;
}

__device__ __forceinline__ f32x8 dw_load8(const h16_t* p, int c, int C, bool vec) {
  if (vec) return unpack8(*reinterpret_cast<const uint4*>(p + c));
  f32x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = (c + j < C) ? (float)p[c + j] : 0.f;
  return r;
}
__device__ __forceinline__ void dw_store8(h16_t* p, int c, int C, bool vec, const f32x8& v) {
  if (vec) {
    *reinterpret_cast<uint4*>(p + c) = pack8(v);
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c + j < C) p[c + j] = (h16_t)v.v[j];
}

// y[n,p,q,c] = bias[c] + sum_{r,s} x[n, p*sh-ph+r*dh, q*sw-pw+s*dw, c] * w[c][r][s]
__global__ __launch_bounds__(256) void dw_fprop_kernel(const DwParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = dw_vec_ok(p);
  const int64_t total = (int64_t)p.N * p.P * p.Q * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, q, pp, n;
    split_index(i, CV, p.Q, p.P, &cv, &q, &pp, &n);
    const int c = cv * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (p.bias && c + j < p.C) ? p.bias[c + j] : 0.f;
    for (int r = 0; r < p.R; ++r) {
      const int ih = pp * p.sh - p.ph + r * p.dh;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int s = 0; s < p.S; ++s) {
        const int iw = q * p.sw - p.pw + s * p.dw_;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        const f32x8 v = dw_load8(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.x_ld, c, p.C, vec);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = c + j < p.C ? c + j : p.C - 1;
          acc[j] += v.v[j] * p.w[(cc * p.R + r) * p.S + s];
        }
      }
    }
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    dw_act8(o, p.act, p.ap);
    dw_store8(p.y + ((int64_t)(n * p.P + pp) * p.Q + q) * p.y_ld, c, p.C, vec, o);
  }
}

// dx[n,h,w,c] = sum_{r,s : (h+ph-r*dh) % sh == 0 ...} dy[n,(h+ph-r*dh)/sh,(w+pw-s*dw)/sw,c] * w[c][r][s]
// here p.x = dy (P x Q, pitch x_ld), p.y = dx (H x W, pitch y_ld)
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = dw_vec_ok(p);
  const int64_t total = (int64_t)p.N * p.H * p.W * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, w, h, n;
    split_index(i, CV, p.W, p.H, &cv, &w, &h, &n);
    const int c = cv * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int r = 0; r < p.R; ++r) {
      const int th = h + p.ph - r * p.dh;
      if (th < 0 || th % p.sh) continue;
      const int oh = th / p.sh;
      if (oh >= p.P) continue;
      for (int s = 0; s < p.S; ++s) {
        const int tw = w + p.pw - s * p.dw_;
        if (tw < 0 || tw % p.sw) continue;
        const int ow = tw / p.sw;
        if (ow >= p.Q) continue;
        const f32x8 g = dw_load8(p.x + ((int64_t)(n * p.P + oh) * p.Q + ow) * p.x_ld, c, p.C, vec);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = c + j < p.C ? c + j : p.C - 1;
          acc[j] += g.v[j] * p.w[(cc * p.R + r) * p.S + s];
        }
      }
    }
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    dw_store8(p.y + ((int64_t)(n * p.H + h) * p.W + w) * p.y_ld, c, p.C, vec, o);
  }
}

// 3x3 / stride 2 / padding 1 / dilation 1 input gradient (round 6: STDC's stride-2 blocks — the depthwise `avd_layer` and the
// AvgPool2d(3, 2, 1) skip, stdcnet.py — ran the generic gather above: run-time tap loops with a division and a modulo per tap and
// eight scalar weight loads per tap and element, 224 us per launch for 167 MB = 0.75 TB/s). An input row / column of parity e takes
// tap 1 of output index i / 2 when even, taps 0 and 2 of (i + 1) / 2 and (i - 1) / 2 when odd: at most four taps, decided by parity.
// A thread owns one channel vector (its 9 x 8 weights in registers) and walks pixels; a block owns a contiguous pixel range.
__global__ __launch_bounds__(256) void dw3x3_s2_dgrad_kernel(const DwParams p, int pix_per_block) {
  const int CV = p.C >> 3;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
  if (ty >= rpp) return;
  const int64_t npix = (int64_t)p.N * p.H * p.W;
  const int64_t p_begin = (int64_t)blockIdx.x * pix_per_block;
  int64_t p_end = p_begin + pix_per_block;
  if (p_end > npix) p_end = npix;
  for (int cv = tx; cv < CV; cv += cols) {
    const int c = cv * 8;
    float w[3][3][8];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[r][s][j] = p.w[(int64_t)(c + j) * 9 + r * 3 + s];
    for (int64_t i = p_begin + ty; i < p_end; i += rpp) {
      const int wx = (int)(i % p.W);
      const int64_t q = i / p.W;
      const int h = (int)(q % p.H);
      const int n = (int)(q / p.H);
      // rows: (tap r, output row) pairs; th = h + 1 - r must be even and th / 2 < P
      int rr[2], oh[2], nr = 0;
      if ((h & 1) == 0) {
        if (h / 2 < p.P) { rr[0] = 1; oh[0] = h / 2; nr = 1; }
      } else {
        if ((h + 1) / 2 < p.P) { rr[nr] = 0; oh[nr] = (h + 1) / 2; ++nr; }
        rr[nr] = 2; oh[nr] = (h - 1) / 2; ++nr;   // (h - 1) / 2 <= P - 1 always: P = floor((H - 1) / 2) + 1
      }
      int ss[2], ow[2], ns = 0;
      if ((wx & 1) == 0) {
        if (wx / 2 < p.Q) { ss[0] = 1; ow[0] = wx / 2; ns = 1; }
      } else {
        if ((wx + 1) / 2 < p.Q) { ss[ns] = 0; ow[ns] = (wx + 1) / 2; ++ns; }
        ss[ns] = 2; ow[ns] = (wx - 1) / 2; ++ns;
      }
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      // same accumulation order as the generic kernel (r ascending, then s ascending)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a >= nr) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b >= ns) continue;
          const f32x8 g = unpack8(*reinterpret_cast<const uint4*>(p.x + ((int64_t)(n * p.P + oh[a]) * p.Q + ow[b]) * p.x_ld + c));
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
              if (r == rr[a] && s == ss[b]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += g.v[j] * w[r][s][j];
              }
        }
      }
      f32x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
      *reinterpret_cast<uint4*>(p.y + i * p.y_ld + c) = pack8(o);
    }
  }
}

// dw[c][r][s] += sum_m dy[m][c] * x[pix(m)+tap][c]; block = chunk of output rows, thread = (row lane,
// channel vector), R*S <= 9 taps accumulated in registers, LDS reduce over row lanes, one fp32
// atomic per (block, channel, tap).
constexpr int kDwMaxTaps = 9;
constexpr int kDwWgradCols = 32;
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwParams p, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int CV = (p.C + 7) >> 3;
  const bool vec = dw_vec_ok(p);
  const int t = threadIdx.x;
  // blockIdx.y = chunk of <= 32 channel vectors (wide layers on small maps — ASPP: 2048 channels @16x32 — would otherwise
  // run 32 blocks with one row lane each); 256 / cols row lanes per block
  const int cols = CV < kDwWgradCols ? CV : kDwWgradCols;
  const int rpp = 256 / cols;
  const int tx = t % cols, ty = t / cols;
  const int64_t M = (int64_t)p.N * p.P * p.Q;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  const int T = p.R * p.S;
  {
    const int cv0 = blockIdx.y * cols;
    const int cv = cv0 + tx;
    const int c = cv * 8;
    float acc[kDwMaxTaps][8];
#pragma unroll
    for (int a = 0; a < kDwMaxTaps; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a][j] = 0.f;
    if (ty < rpp && cv < CV) {
      for (int64_t m = r_begin + ty; m < r_end; m += rpp) {
        const int q = (int)(m % p.Q);
        const int pp = (int)((m / p.Q) % p.P);
        const int n = (int)(m / ((int64_t)p.Q * p.P));
        const f32x8 g = dw_load8(p.dy + m * p.y_ld, c, p.C, vec);
#pragma unroll
        for (int a = 0; a < kDwMaxTaps; ++a) {
          // no break / continue in here: an early exit keeps hipcc from unrolling, `a` becomes a runtime index and the whole
          // accumulator array moves to scratch memory (304 B/lane: every FMA through memory)
          const int r = a / p.S, s = a - r * p.S;
          const int ih = pp * p.sh - p.ph + r * p.dh, iw = q * p.sw - p.pw + s * p.dw_;
          if (a < T && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
            const f32x8 v = dw_load8(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.x_ld, c, p.C, vec);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][j] += g.v[j] * v.v[j];
          }
        }
      }
    }
    for (int a = 0; a < T; ++a) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = 0.f;
#pragma unroll
        for (int aa = 0; aa < kDwMaxTaps; ++aa)
          if (aa == a) v = acc[aa][j];
        red[t * 8 + j] = v;
      }
      __syncthreads();
      for (int idx = t; idx < cols * 8; idx += 256) {
        const int x = idx >> 3, j = idx & 7;
        float s = 0.f;
        for (int yy = 0; yy < rpp; ++yy) s += red[(yy * cols + x) * 8 + j];
        const int cc = (cv0 + x) * 8 + j;
        if (cv0 + x < CV && cc < p.C && s != 0.f) unsafeAtomicAdd(p.dw + (int64_t)cc * T + a, s);
      }
    }
  }
}

// ---- generic depthwise forward / stride-1 input gradient with the taps in registers (round 3) ----------------------------------------------
// out[n][oh][ow][c] = bias[c] + sum_t in[n][oh*sh + offh[t]][ow*sw + offw[t]][c] * w[c][t]            (t < R*S <= 9)
// fprop: offh = r*dh - ph; stride-1 dgrad: in = dy, offh = ph - r*dh (no stride holes). The kernels above re-read 8 scalar weights per tap
// per pixel (72 loads per output vector) and keep every tap load in its own branch: on DeepLabv3+'s ASPP (2048 channels @16x32, dilation
// 12 / 24 / 36: 67 MB per pass) they ran at 0.7 TB/s. Here a thread owns one channel vector — its <= 9 x 8 weights live in registers — and
// walks pixels; all tap loads of a pixel are issued unconditionally (clamped address, weight 0 outside the image) before any arithmetic.
struct DwTapParams {
  const h16_t* in;
  const float* w;
  const float* bias;
  h16_t* out;
  int N, C, IH, IW, OH, OW, sh, sw, T, in_ld, out_ld, ppb;  // ppb: pixels per block
  int offh[kDwMaxTaps], offw[kDwMaxTaps];
  int act;
  float ap;
};

__global__ __launch_bounds__(256) void dw_taps_kernel(const DwTapParams p) {
  const int CV = p.C >> 3;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  const int t = threadIdx.x;
  const int tx = t % cols, ty = t / cols;
  const int cv = blockIdx.y * cols + tx;
  if (ty >= rpp || cv >= CV) return;
  const int c = cv * 8;
  float w[kDwMaxTaps][8], bias[8];
  if (p.T == kDwMaxTaps && ((((uintptr_t)p.w) & 15) == 0)) {  // 3x3: the thread's 8 x 9 weights are 72 consecutive floats (18 x 16 bytes)
    float flat[8 * kDwMaxTaps];
    const float4* src = reinterpret_cast<const float4*>(p.w + (int64_t)c * kDwMaxTaps);
#pragma unroll
    for (int v = 0; v < 2 * kDwMaxTaps; ++v) {
      const float4 f = src[v];
      flat[4 * v] = f.x;
      flat[4 * v + 1] = f.y;
      flat[4 * v + 2] = f.z;
      flat[4 * v + 3] = f.w;
    }
#pragma unroll
    for (int a = 0; a < kDwMaxTaps; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) w[a][j] = flat[j * kDwMaxTaps + a];
  } else {
#pragma unroll
    for (int a = 0; a < kDwMaxTaps; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) w[a][j] = a < p.T ? p.w[(int64_t)(c + j) * p.T + a] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = p.bias ? p.bias[c + j] : 0.f;
  const int npix = p.N * p.OH * p.OW;  // (< 2^31: checked by the host)
  const int p_end = min(npix, ((int)blockIdx.x + 1) * p.ppb);
  for (int pix = (int)blockIdx.x * p.ppb + ty; pix < p_end; pix += rpp) {
    const unsigned ohw = (unsigned)(p.OH * p.OW);
    const unsigned n = (unsigned)pix / ohw;
    const unsigned rem = (unsigned)pix - n * ohw;
    const int oh = (int)(rem / (unsigned)p.OW), ow = (int)(rem - (rem / (unsigned)p.OW) * (unsigned)p.OW);
    const h16_t* const img = p.in + (int64_t)n * p.IH * p.IW * p.in_ld + c;
    uint4 raw[kDwMaxTaps];
    float m[kDwMaxTaps];
#pragma unroll
    for (int a = 0; a < kDwMaxTaps; ++a) {
      const int ih = oh * p.sh + p.offh[a], iw = ow * p.sw + p.offw[a];
      const bool ok = a < p.T && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      m[a] = ok ? 1.f : 0.f;
      const int64_t off = ok ? ((int64_t)ih * p.IW + iw) * p.in_ld : 0;
      raw[a] = *reinterpret_cast<const uint4*>(img + off);
    }
    __builtin_amdgcn_sched_barrier(0);  // every tap load ahead of the arithmetic
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[j];
#pragma unroll
    for (int a = 0; a < kDwMaxTaps; ++a) {
      const f32x8 v = unpack8(raw[a]);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (v.v[j] * m[a]) * w[a][j];
    }
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    dw_act8(o, p.act, p.ap);
    *reinterpret_cast<uint4*>(p.out + (int64_t)pix * p.out_ld + c) = pack8(o);
  }
}

// launch for fprop (dgrad == false) or the stride-1 input gradient; false when the taps form does not apply
static bool dw_taps_launch(const DwParams& p, bool dgrad, hipStream_t s, int* status) {
  const int T = p.R * p.S;
  if (T > kDwMaxTaps || (p.C & 7) || (p.x_ld & 7) || (p.y_ld & 7) || ((((uintptr_t)p.x) | ((uintptr_t)p.y)) & 15)) return false;
  if (dgrad && (p.sh != 1 || p.sw != 1)) return false;
  DwTapParams q{};
  q.in = p.x;   // fprop: x; dgrad: dy (the caller swapped the pitches)
  q.w = p.w;
  q.bias = dgrad ? nullptr : p.bias;
  q.act = dgrad ? CVHIP_ACT_NONE : p.act;
  q.ap = p.ap;
  q.out = p.y;
  q.N = p.N;
  q.C = p.C;
  q.T = T;
  q.in_ld = p.x_ld;
  q.out_ld = p.y_ld;
  if (!dgrad) {
    q.IH = p.H; q.IW = p.W; q.OH = p.P; q.OW = p.Q; q.sh = p.sh; q.sw = p.sw;
    for (int r = 0; r < p.R; ++r)
      for (int c2 = 0; c2 < p.S; ++c2) {
        q.offh[r * p.S + c2] = r * p.dh - p.ph;
        q.offw[r * p.S + c2] = c2 * p.dw_ - p.pw;
      }
  } else {
    q.IH = p.P; q.IW = p.Q; q.OH = p.H; q.OW = p.W; q.sh = 1; q.sw = 1;
    for (int r = 0; r < p.R; ++r)
      for (int c2 = 0; c2 < p.S; ++c2) {
        q.offh[r * p.S + c2] = p.ph - r * p.dh;
        q.offw[r * p.S + c2] = p.pw - c2 * p.dw_;
      }
  }
  const int64_t npix = (int64_t)q.N * q.OH * q.OW;
  if (npix <= 0 || npix >= (1ll << 31) - 65536) return false;
  const int CV = q.C >> 3;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  const int chunks = (CV + cols - 1) / cols;
  // >= ~1024 blocks, >= 8 pixels per pixel lane (the 72 weight loads are paid once per block)
  int64_t ppb = (npix * chunks + 1023) / 1024;
  if (ppb < (int64_t)rpp * 8) ppb = (int64_t)rpp * 8;
  q.ppb = (int)ppb;
  const int64_t bx = (npix + ppb - 1) / ppb;
  hipLaunchKernelGGL(dw_taps_kernel, dim3((unsigned)bx, (unsigned)chunks), dim3(256), 0, s, q);
  *status = check_launch("dw_taps_kernel");
  return true;
}

// ---- 3x3 / stride 1 / dilation 1 fast path -------------------------------------------------------------------------------
// The generic kernels above issue 9 tap loads (+ 72 scalar weight loads) per output vector and walk pixels in flat order:
// on DeepLabv3+'s decoder (304 / 512 channels @128x256, batch 16) every tap missed L2 and the three passes took 1.7-2.2 ms
// each (rocprofv3: 41 % of the step). Here a thread owns one 16-B channel vector, keeps its 3x3 weights in registers and
// WALKS an image row with a 3x3 register window: 3 new 16-B loads per pixel instead of 9; neighbouring row lanes of a block
// take neighbouring image rows, so the halo rows are shared through L1/L2.
struct Dw3Params {
  const h16_t* in;   // fprop: x; dgrad: dy; wgrad: x
  const h16_t* dy;   // wgrad only
  const float* w;     // [C][3][3]
  const float* bias;  // fprop only (may be null)
  h16_t* out;
  float* dw;
  int N, C, IH, IW, OH, OW, ph, pw, in_ld, out_ld, dy_ld, flip, seg_len, rows_per_thread;
  int act;   // MODE 0 (fprop): fused activation
  float ap;
  // MODE 0, LDS form only (round 6, second session): training-mode BatchNorm sums of the fp32 outputs (before the rounding to 16 bits, as
  // the convolution epilogues take them), one partial row [2][C] per (image, row block, strip) — the reduction pass over the stored
  // output (colreduce_kernel<0>: 587 MB per decoder layer of DeepLabv3+) is not run. NULL = none. Never with a fused activation.
  float* stats_partial;
};

__device__ __forceinline__ f32x8 dw3_load(const h16_t* row, bool row_ok, int iw, int IW, int in_ld, int c) {
  if (row_ok && (unsigned)iw < (unsigned)IW) return unpack8(*reinterpret_cast<const uint4*>(row + (int64_t)iw * in_ld + c));
  f32x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z.v[j] = 0.f;
  return z;
}

// MODE 0: out[n,oh,ow,c] = bias + sum_{r,s} in[n, oh-ph+r, ow-pw+s, c] * w[c][r][s]      (flip: w[c][2-r][2-s] — dgrad)
// MODE 1: dw[c][r][s]   += sum_{n,oh,ow} dy[n,oh,ow,c] * in[n, oh-ph+r, ow-pw+s, c]
template <int MODE>
__global__ __launch_bounds__(256) void dw3x3_kernel(const Dw3Params p) {
  __shared__ float red[MODE == 1 ? 256 * 8 : 1];
  const int CV = p.C >> 3;
  const int t = threadIdx.x;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  const int tx = t % cols, ty = t / cols;
  const int nseg = (p.OW + p.seg_len - 1) / p.seg_len;
  const int ncv = (CV + cols - 1) / cols;
  int b = blockIdx.x;
  const int cvc = b % ncv;
  b /= ncv;
  const int seg = b % nseg;
  const int rowblk = b / nseg;
  const int cv = cvc * cols + tx;
  const bool active = ty < rpp && cv < CV;
  const int c = (cv < CV ? cv : CV - 1) * 8;
  float w[3][3][8];
  float acc9[MODE == 1 ? 9 : 1][8];
  if (MODE == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[r][s2][j] = p.w[(int64_t)(c + j) * 9 + (p.flip ? (2 - r) * 3 + (2 - s2) : r * 3 + s2)];
  } else {
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc9[a][j] = 0.f;
  }
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = (MODE == 0 && p.bias) ? p.bias[c + j] : 0.f;
  const int q0 = seg * p.seg_len;
  const int q1 = q0 + p.seg_len < p.OW ? q0 + p.seg_len : p.OW;
  const int total_rows = p.N * p.OH;
  if (active) {
    for (int it = 0; it < p.rows_per_thread; ++it) {
      const int row = (rowblk * p.rows_per_thread + it) * rpp + ty;
      if (row >= total_rows) break;
      const int n = row / p.OH, oh = row - n * p.OH;
      const h16_t* rp[3];
      bool rok[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ih = oh - p.ph + r;
        rok[r] = (unsigned)ih < (unsigned)p.IH;
        rp[r] = p.in + ((int64_t)(n * p.IH + (rok[r] ? ih : 0)) * p.IW) * p.in_ld;
      }
      f32x8 win[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        win[r][0] = dw3_load(rp[r], rok[r], q0 - p.pw, p.IW, p.in_ld, c);
        win[r][1] = dw3_load(rp[r], rok[r], q0 - p.pw + 1, p.IW, p.in_ld, c);
      }
      // software pipeline: the raw 16-B loads of the new window columns (and dy) of the NEXT kDw3Ahead pixels are in flight
      // while this pixel's 72 FMAs run. With weights + window in registers only 2 waves fit a SIMD, so one pixel of lookahead
      // left the walk latency-bound (3 x 16 B per lane per ~2 us round trip: 1.3 TB/s algorithmic on the 319 MB DeepLabv3+
      // decoder tensors); three pixels ahead triple the bytes in flight.
      constexpr int kDw3Ahead = 3;
      uint4 nraw[kDw3Ahead][3], ndy[kDw3Ahead];
      auto issue = [&](int q, uint4 (&raw)[3], uint4& dyv) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int iw = q - p.pw + 2;
          const bool ok = q < q1 && rok[r] && (unsigned)iw < (unsigned)p.IW;
          raw[r] = ok ? *reinterpret_cast<const uint4*>(rp[r] + (int64_t)iw * p.in_ld + c) : uint4{0u, 0u, 0u, 0u};
        }
        if (MODE == 1) dyv = q < q1 ? *reinterpret_cast<const uint4*>(p.dy + ((int64_t)row * p.OW + q) * p.dy_ld + c) : uint4{0u, 0u, 0u, 0u};
      };
#pragma unroll
      for (int a = 0; a < kDw3Ahead; ++a) issue(q0 + a, nraw[a], ndy[a]);
      for (int q = q0; q < q1; ++q) {
#pragma unroll
        for (int r = 0; r < 3; ++r) win[r][2] = unpack8(nraw[0][r]);
        const uint4 cdy = ndy[0];
#pragma unroll
        for (int a = 0; a + 1 < kDw3Ahead; ++a) {  // rotate the ring (register moves: cheap next to 72 FMAs)
#pragma unroll
          for (int r = 0; r < 3; ++r) nraw[a][r] = nraw[a + 1][r];
          ndy[a] = ndy[a + 1];
        }
        issue(q + kDw3Ahead, nraw[kDw3Ahead - 1], ndy[kDw3Ahead - 1]);
        if (MODE == 0) {
          f32x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a = bias[j];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int s2 = 0; s2 < 3; ++s2) a += win[r][s2].v[j] * w[r][s2][j];
            o.v[j] = a;
          }
          dw_act8(o, p.act, p.ap);
          *reinterpret_cast<uint4*>(p.out + ((int64_t)row * p.OW + q) * p.out_ld + c) = pack8(o);
        } else {
          const f32x8 g = unpack8(cdy);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc9[r * 3 + s2][j] += g.v[j] * win[r][s2].v[j];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          win[r][0] = win[r][1];
          win[r][1] = win[r][2];
        }
      }
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[t * 8 + j] = active ? acc9[a][j] : 0.f;
      __syncthreads();
      for (int idx = t; idx < cols * 8; idx += 256) {
        const int x = idx >> 3, j = idx & 7;
        float sum = 0.f;
        for (int yy = 0; yy < rpp; ++yy) sum += red[(yy * cols + x) * 8 + j];
        const int cc = (cvc * cols + x) * 8 + j;
        if (cvc * cols + x < CV && sum != 0.f) unsafeAtomicAdd(p.dw + (int64_t)cc * 9 + a, sum);
      }
    }
  }
}

// geometry test + launch; returns false when the fast path does not apply
static bool dw3_applicable(const DwParams& p, const void* a, const void* b2, const void* c2) {
  return p.R == 3 && p.S == 3 && p.sh == 1 && p.sw == 1 && p.dh == 1 && p.dw_ == 1 && (p.C & 7) == 0 && (p.x_ld & 7) == 0 &&
         (p.y_ld & 7) == 0 && p.ph <= 2 && p.pw <= 2 && ((((uintptr_t)a) | ((uintptr_t)b2) | ((uintptr_t)c2)) & 15) == 0;
}

static void dw3_launch_geom(Dw3Params& q, int* grid, bool wgrad = false) {
  const int CV = q.C >> 3;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  constexpr int seg_env = 0, rpt_env = 0;   // (forced segment length / rows per thread: dev sweeps of round 3, tools/dw_bench.py)
  // row segment one thread walks: 64 columns for fprop / dgrad; the weight-gradient walk keeps 9 x 8 sums in registers and
  // pays an LDS fold + atomics per block, so it takes whole rows up to 256 columns (tools/dw_bench.py: 397 -> 262 us on the
  // DeepLabv3+ decoder's 304-channel tensor)
  q.seg_len = seg_env > 0 ? seg_env : (wgrad ? (q.OW <= 256 ? q.OW : 128) : (q.OW <= 96 ? q.OW : 64));
  const int nseg = (q.OW + q.seg_len - 1) / q.seg_len;
  const int ncv = (CV + cols - 1) / cols;
  const int64_t rows = (int64_t)q.N * q.OH;
  // enough blocks to fill the chip (>= ~2048), at most 4 image rows per thread
  int rpt = 4;
  while (rpt > 1 && cdiv64(rows, (int64_t)rpp * rpt) * nseg * ncv < 2048) rpt >>= 1;
  if (rpt_env > 0) rpt = rpt_env;
  q.rows_per_thread = rpt;
  *grid = (int)(cdiv64(rows, (int64_t)rpp * rpt) * nseg * ncv);
}

// ---- 3x3 / stride 1 / dilation 1 through an LDS ring fed by LDS-DMA (round 3) -----------------------------------------------------------
// The register-window walk above keeps 3 x 16 B per lane in flight behind a ~2 us round trip with 2 waves per SIMD: 1.5 TB/s on the
// DeepLabv3+ decoder tensors (412-453 us per pass over 319 MB in / out), and every input row is fetched three times (by the three
// output rows that use it) through L1/L2. Here a block owns a STRIP — TW output columns x RB output rows of one image, a chunk of
// <= 64 channel vectors — and walks it top to bottom:
//   * input row segments (TW + 2 pixels x chunk) go global -> LDS with global_load_lds_dwordx4 into a ring of NR rows, issued NR - 3
//     rows ahead of their first use: two row segments (24 KB per block, 2 blocks per CU) are in flight while a row is computed, no
//     VGPRs are spent on staging, and each input element is fetched from memory ONCE per strip (halo: 2 columns per TW, 2 rows per RB);
//   * a thread owns one channel vector (weights in registers, as before) and PXT adjacent output pixels of the row: 3 x (PXT + 2)
//     ds_read_b128 per PXT outputs;
//   * the weight gradient reads its dy row segments through a second ring the same way (an ordinary global load beside LDS-DMAs in
//     flight makes hipcc drain the DMA queue with s_waitcnt vmcnt(0)).
// One raw s_barrier per row; the counted s_waitcnt leaves the younger rows in flight. Output stores are not counted in the wait
// (vmcnt is shared with stores on gfx9: counting only the younger LOADS makes the wait at worst stricter, never weaker).
// Out-of-image rows / columns and lanes past the segment read a zero page, so every wave issues the same number of DMAs per row.
__device__ __attribute__((aligned(64))) unsigned int g_dw_zero[16];

#define CVHIP_DW_GLDS16(src, dst)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int N>
__device__ __forceinline__ void dw_wait_vm_barrier() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

struct Dw3LdsGeom {
  int cols, rpp, TW, RB, nstrip, nrowblk, ncv;  // chunk width (channel vectors), pixel lanes, strip width / height, grid factors
};

// KX / KD: DMA instructions per 256 lanes per input / dy row segment (host: ceil((TW + 2) * cols / 256), ceil(TW * cols / 256)).
// Block = kDwLdsCW compute waves + 1 PRODUCER wave. The producer issues every LDS-DMA of the block (4 * KX per input row) and
// is the only wave that waits on them: its VM queue holds nothing but those DMAs, so the counted s_waitcnt is exact. (With the
// compute waves issuing the DMAs themselves, their output stores sit in the same in-order vmcnt queue behind the look-ahead row:
// the wait for "row j+2 landed" then also waits for the look-ahead row j+3 — measured 199 us fprop / 503 us wgrad against 250 / 269
// for the register-window kernel on the 304-channel decoder tensor.)
constexpr int kDwLdsCW = 3;  // compute waves per block (+ 1 producer wave): 2 blocks per CU at ~200 VGPRs need <= 8 waves per CU
template <int MODE, int PXT, int KX, int KD>
__global__ __launch_bounds__((kDwLdsCW + 1) * 64) void dw3x3_lds_kernel(const Dw3Params p, const Dw3LdsGeom g) {
  constexpr int NR = MODE == 1 ? 5 : 6;  // input-row ring: rows j .. j+2 in use, j+3 (and j+4: fprop / dgrad) in flight
  constexpr int ND = MODE == 1 ? 3 : 1;  // dy-row ring (weight gradient): row j in use, j+1 and j+2 in flight (a ring of 2 exposed the
                                         // DMA latency of the dy row every iteration: 421 vs 269 us for the register-window kernel)
  extern __shared__ __attribute__((aligned(1024))) unsigned char dw_smem[];
  const int cols = g.cols, TW = g.TW, TWI = g.TW + 2;
  constexpr int xrow_bytes = KX * 4096, drow_bytes = KD * 4096;  // ring slots are whole DMA groups (256 lanes x 16 B)
  unsigned char* const sX = dw_smem;
  unsigned char* const sD = dw_smem + NR * xrow_bytes;
  const int CV = p.C >> 3;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // The channel chunks of one strip share cache lines wherever a chunk boundary is not 128-byte aligned (560 channels: 35 + 35 vectors
  // at a 1120-byte pixel pitch). Blocks are dealt round-robin to the 8 XCDs, each with its own L2: blocks i and i + 8 of a group of
  // 8 * ncv take the chunks of the same strip, so the shared lines are fetched (and the partial output lines merged) in ONE L2.
  int b = blockIdx.x, cvc;
  {
    const int rest_all = gridDim.x / g.ncv, r8 = rest_all & ~7;
    if (b < r8 * g.ncv) {
      const int lo = b & 7, hi = b >> 3;
      cvc = hi % g.ncv;
      b = (hi / g.ncv) * 8 + lo;
    } else {
      const int tl = b - r8 * g.ncv;
      cvc = tl % g.ncv;
      b = r8 + tl / g.ncv;
    }
  }
  const int strip = b % g.nstrip;
  b /= g.nstrip;
  const int rowblk = b % g.nrowblk;
  const int n = b / g.nrowblk;
  const int q0 = strip * TW;
  const int r0 = rowblk * g.RB;
  const int r1 = min(r0 + g.RB, p.OH);
  const int nrows = r1 - r0;

  constexpr int NCT = kDwLdsCW * 64;  // compute threads
  if (wave == kDwLdsCW) {
    // ================================ producer ================================
    const h16_t* const zero = reinterpret_cast<const h16_t*>(g_dw_zero) + (lane & 3) * 8;
    // element e = k * 64 + lane of a row segment is pixel e / cols, channel vector e % cols; its offset inside an image row
    // (pixels * pitch + channel) or -1 = zero page (halo outside the image, lanes past the segment, channels past C)
    int xoff[4 * KX], doff[KD > 0 ? 4 * KD : 1];
#pragma unroll
    for (int k = 0; k < 4 * KX; ++k) {
      const int e = k * 64 + lane;
      const int px = e / cols, v = e - px * cols;
      const int iw = q0 - p.pw + px;
      const bool ok = px < TWI && (unsigned)iw < (unsigned)p.IW && cvc * cols + v < CV;
      xoff[k] = ok ? iw * p.in_ld + (cvc * cols + v) * 8 : -1;
    }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4 * KD; ++k) {
        const int e = k * 64 + lane;
        const int px = e / cols, v = e - px * cols;
        const int ow = q0 + px;
        const bool ok = px < TW && ow < p.OW && cvc * cols + v < CV;
        doff[k] = ok ? ow * p.dy_ld + (cvc * cols + v) * 8 : -1;
      }
    }
    const int i_first = r0 - p.ph;  // first input row of the strip (may be < 0: zero rows)
    auto issue_x = [&](int j) {     // j-th input row of the strip -> ring slot j % NR
      const int ih = i_first + j;
      const bool rok = (unsigned)ih < (unsigned)p.IH;
      const h16_t* const row = p.in + ((int64_t)(n * p.IH + (rok ? ih : 0)) * p.IW) * p.in_ld;
      unsigned char* const dst = sX + (j % NR) * xrow_bytes;
#pragma unroll
      for (int k = 0; k < 4 * KX; ++k) {
        const h16_t* src = (rok && xoff[k] >= 0) ? row + xoff[k] : zero;
        CVHIP_DW_GLDS16(src, dst + k * 1024);
      }
    };
    auto issue_d = [&](int j) {  // dy row r0 + j -> ring slot j % ND
      const int oh = r0 + j;
      const bool rok = oh < r1;
      const h16_t* const row = p.dy + ((int64_t)(n * p.OH + (rok ? oh : 0)) * p.OW) * p.dy_ld;
      unsigned char* const dst = sD + (j % ND) * drow_bytes;
#pragma unroll
      for (int k = 0; k < 4 * KD; ++k) {
        const h16_t* src = (rok && doff[k] >= 0) ? row + doff[k] : zero;
        CVHIP_DW_GLDS16(src, dst + k * 1024);
      }
    };
    // prologue. Per iteration the dy row is issued BEFORE the input row; the prologue interleaves the same way, so that at every
    // barrier the groups younger than what the compute waves are about to read are exactly the look-ahead of the last NR - 4
    // iterations: (dy row, input row) pairs for the weight gradient (NR 5: one pair), input rows otherwise (NR 6: two rows)
    if (MODE == 1) {
      issue_x(0);
      issue_x(1);
      issue_d(0);
      issue_x(2);
      issue_d(1);
      issue_x(3);
    } else {
#pragma unroll
      for (int j = 0; j < NR - 1; ++j) issue_x(j);
    }
    for (int j = 0; j < nrows; ++j) {
      dw_wait_vm_barrier<(NR - 4) * (4 * KX + (MODE == 1 ? 4 * KD : 0))>();  // input rows j .. j+2 and dy row j have landed
      if (MODE == 1) issue_d(j + 2);  // into the slot of dy row j-1
      issue_x(j + NR - 1);            // into the slot of row j-1: every compute wave has passed the barrier, i.e. finished output row j-1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead rows past the strip must not land after the block retires
    if (MODE == 1) {
      for (int a = 0; a < 18; ++a) __syncthreads();  // the 18 barriers of the compute waves' reduction below
    }
    if (MODE == 0 && p.stats_partial) {
      __syncthreads();  // the 2 barriers of the compute waves' BatchNorm-sum reduction below
      __syncthreads();
    }
    return;
  }

  // ================================ compute waves ================================
  const int tx = t % cols, ty = t / cols;
  const int cv = cvc * cols + tx;
  const bool active = ty < g.rpp && cv < CV;
  const int c = (cv < CV ? cv : CV - 1) * 8;
  float w[3][3][8];
  float acc9[MODE == 1 ? 9 : 1][8];
  float bias[8];
  if (MODE == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[r][s2][j] = p.w[(int64_t)(c + j) * 9 + (p.flip ? (2 - r) * 3 + (2 - s2) : r * 3 + s2)];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = p.bias ? p.bias[c + j] : 0.f;
  } else {
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc9[a][j] = 0.f;
  }
  const int px0 = ty * PXT;  // first output pixel of this thread inside the strip
  float bs1[MODE == 0 ? 8 : 1], bs2[MODE == 0 ? 8 : 1];  // BatchNorm sums of this thread's outputs (stats_partial)
#pragma unroll
  for (int jj = 0; jj < (MODE == 0 ? 8 : 1); ++jj) bs1[jj] = bs2[jj] = 0.f;
  const bool want_stats = MODE == 0 && p.stats_partial != nullptr;
  for (int j = 0; j < nrows; ++j) {
    asm volatile("s_barrier" ::: "memory");  // the producer arrives here after rows j .. j+2 (and dy row j) have landed
    if (active) {
      const int oh = r0 + j;
      f32x8 win[3][PXT + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned char* const row = sX + ((j + r) % NR) * xrow_bytes;
#pragma unroll
        for (int s2 = 0; s2 < PXT + 2; ++s2) win[r][s2] = unpack8(*reinterpret_cast<const uint4*>(row + ((px0 + s2) * cols + tx) * 16));
      }
#pragma unroll
      for (int q = 0; q < PXT; ++q) {
        const int ow = q0 + px0 + q;
        if (MODE == 0) {
          f32x8 o;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            float a = bias[jj];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int s2 = 0; s2 < 3; ++s2) a += win[r][q + s2].v[jj] * w[r][s2][jj];
            o.v[jj] = a;
          }
          dw_act8(o, p.act, p.ap);
          if (ow < p.OW) *reinterpret_cast<uint4*>(p.out + ((int64_t)(n * p.OH + oh) * p.OW + ow) * p.out_ld + c) = pack8(o);
          if constexpr (MODE == 0) {
            if (want_stats && ow < p.OW) {
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) {
                bs1[jj] += o.v[jj];
                bs2[jj] += o.v[jj] * o.v[jj];
              }
            }
          }
        } else {
          // dy of pixels past the image / strip edge was staged as zeros
          const f32x8 gq = unpack8(*reinterpret_cast<const uint4*>(sD + (j % ND) * drow_bytes + ((px0 + q) * cols + tx) * 16));
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) acc9[r * 3 + s2][jj] += gq.v[jj] * win[r][q + s2].v[jj];
        }
      }
    }
  }
  if constexpr (MODE == 0) {
    if (want_stats) {
      __syncthreads();  // (the producer has drained its queue before it joins: the rings are dead)
      float* const red = reinterpret_cast<float*>(dw_smem);  // NCT x 16 floats
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        red[t * 16 + jj] = active ? bs1[jj] : 0.f;
        red[t * 16 + 8 + jj] = active ? bs2[jj] : 0.f;
      }
      __syncthreads();
      const int64_t prow = ((int64_t)n * g.nrowblk + rowblk) * g.nstrip + strip;   // every row is written once by each channel chunk
      for (int idx = t; idx < cols * 16; idx += NCT) {
        const int x = idx >> 4, jj = idx & 15;
        float sum = 0.f;
        for (int yy = 0; yy < g.rpp; ++yy) sum += red[(yy * cols + x) * 16 + jj];
        const int cc = (cvc * cols + x) * 8 + (jj & 7);
        if (cvc * cols + x < CV) p.stats_partial[(prow * 2 + (jj >> 3)) * p.C + cc] = sum;
      }
    }
  }
  if (MODE == 1) {
    __syncthreads();  // (the producer has drained its queue before it joins: the rings are dead)
    float* const red = reinterpret_cast<float*>(dw_smem);  // NCT x 8 floats
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      if (a > 0) __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) red[t * 8 + jj] = active ? acc9[a][jj] : 0.f;
      __syncthreads();
      for (int idx = t; idx < cols * 8; idx += NCT) {
        const int x = idx >> 3, jj = idx & 7;
        float sum = 0.f;
        for (int yy = 0; yy < g.rpp; ++yy) sum += red[(yy * cols + x) * 8 + jj];
        const int cc = (cvc * cols + x) * 8 + jj;
        if (cvc * cols + x < CV && sum != 0.f) unsafeAtomicAdd(p.dw + (int64_t)cc * 9 + a, sum);
      }
    }
  }
}

// strip geometry + launch; false when the LDS form does not apply (then the register-window kernel runs)
// rows_out != NULL: geometry query only — *rows_out = partial rows the stats form writes (N * row blocks * strips); nothing is launched
template <int MODE>
static bool dw3_lds_launch(const Dw3Params& q, hipStream_t s, int* status, int64_t* rows_out = nullptr) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVHIP_DW3_LDS");
    on = e ? atoi(e) : 1;
  }
  if (!on) return false;
  constexpr int PXT = MODE == 1 ? 2 : 3;
  const int CV = q.C >> 3;
  Dw3LdsGeom g;
  // channel chunks of equal width (560 channels = 70 vectors ran as 64 + 6: half of the blocks moved 9 % of the bytes at the full
  // per-row latency), at most kColsMax vectors wide: narrower chunks leave more pixel lanes, i.e. wider strips and less column halo
  // (measured on rotating 560- / 512- / 304- / 256-channel 16 x 128 x 256 tensors, tools/dw_bench_cold.py, profiles/r06_dw_chunks.log:
  // the weight gradient 326 / 309 / 195 / 222 us at 64 -> 255 / 234 / 166 / 192 at 24; fprop / dgrad flat within the noise)
  constexpr int cols_max = MODE == 1 ? 24 : 64;
  g.ncv = (CV + cols_max - 1) / cols_max;
  g.cols = (CV + g.ncv - 1) / g.ncv;
  g.rpp = (kDwLdsCW * 64) / g.cols;
  g.TW = g.rpp * PXT;
  if (q.OW < g.TW || q.OH < 8) return false;  // small maps: the strip would be mostly halo
  g.nstrip = (q.OW + g.TW - 1) / g.TW;
  const int kx = ((g.TW + 2) * g.cols + 255) / 256, kd = (g.TW * g.cols + 255) / 256;
  // strip height: >= ~1024 blocks for the chip, >= 16 rows to amortise the 2 halo rows and the prologue
  // (weight gradient: every block ends with 9 * C atomics onto the SAME 9 * C addresses — 1664 blocks of 32 rows spent more time
  // serialised on them than walking: 407 us; whole-height strips, ~400 blocks, as long as they still cover the CUs)
  const int min_blocks = MODE == 1 ? 384 : 1024;
  int rb = q.OH;
  while (rb > 16 && (int64_t)q.N * g.nstrip * g.ncv * ((q.OH + rb - 1) / rb) < min_blocks) rb = (rb + 1) / 2;
  g.RB = rb;
  g.nrowblk = (q.OH + rb - 1) / rb;
  const int64_t grid = (int64_t)q.N * g.nrowblk * g.nstrip * g.ncv;
  if (grid > 0x7fffffff) return false;
  const int lds = (MODE == 1 ? 5 : 6) * kx * 4096 + (MODE == 1 ? 3 * kd * 4096 : 0);
  if (lds < 256 * 8 * 4 || lds > 96 * 1024) return false;
  if (MODE == 0 && lds < kDwLdsCW * 64 * 16 * 4) return false;   // (the stats reduction's scratch; never the case: >= 6 x 8 KB)
  if (rows_out) {
    const bool inst = (kx == 3 && (kd == 2 || kd == 3)) || (kx == 4 && (kd == 3 || kd == 4)) || (kx == 2 && (kd == 2 || kd == 1));
    if (!inst) return false;
    *rows_out = (int64_t)q.N * g.nrowblk * g.nstrip;
    return true;
  }
#define CVHIP_DW3L(KXV, KDV)                                                                                                 \
  {                                                                                                                          \
    auto kern = dw3x3_lds_kernel<MODE, PXT, KXV, (MODE == 1 ? KDV : 0)>;                                                    \
    static bool attr_done[64] = {};                                                                                          \
    int devid = 0;                                                                                                           \
    (void)hipGetDevice(&devid);                                                                                              \
    if (!attr_done[devid & 63]) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) { \
        *status = CVHIP_ERR_LAUNCH;                                                                                          \
        return true;                                                                                                         \
      }                                                                                                                      \
      attr_done[devid & 63] = true;                                                                                          \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((kDwLdsCW + 1) * 64), lds, s, q, g);                                                 \
    *status = check_launch("dw3x3_lds_kernel");                                                                              \
    return true;                                                                                                             \
  }
  if (kx == 3 && kd == 2) CVHIP_DW3L(3, 2)
  if (kx == 3 && kd == 3) CVHIP_DW3L(3, 3)
  if (kx == 4 && kd == 3) CVHIP_DW3L(4, 3)
  if (kx == 4 && kd == 4) CVHIP_DW3L(4, 4)
  if (kx == 2 && kd == 2) CVHIP_DW3L(2, 2)
  if (kx == 2 && kd == 1) CVHIP_DW3L(2, 1)
#undef CVHIP_DW3L
  return false;
}

static int fill(const cvhip_conv_desc* d, DwParams* p) {
  if (!d) return CVHIP_ERR_INVALID;
  if (d->groups != d->C || d->K != d->C) return CVHIP_ERR_UNSUPPORTED;
  if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->R <= 0 || d->S <= 0 || d->stride_h <= 0 || d->stride_w <= 0 ||
      d->dil_h <= 0 || d->dil_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->x_ld < d->C || d->y_ld < d->C)
    return CVHIP_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  p->N = d->N;
  p->C = d->C;
  p->H = d->H;
  p->W = d->W;
  p->R = d->R;
  p->S = d->S;
  p->sh = d->stride_h;
  p->sw = d->stride_w;
  p->ph = d->pad_h;
  p->pw = d->pad_w;
  p->dh = d->dil_h;
  p->dw_ = d->dil_w;
  p->P = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  p->Q = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  if (p->P <= 0 || p->Q <= 0) return CVHIP_ERR_INVALID;
  return CVHIP_OK;
}

static inline int grid_for(int64_t total) {
  int64_t b = cdiv64(total, 256);
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

static int dw_fprop_impl(const cvhip_conv_desc* d, const void* x, const float* w, const float* bias, int act, float ap, void* y, void* stream) {
  DwParams p;
  int st = fill(d, &p);
  if (st) return st;
  if (!x || !w || !y) return CVHIP_ERR_INVALID;
  p.x = (const h16_t*)x;
  p.w = w;
  p.bias = bias;
  p.act = act;
  p.ap = ap;
  p.y = (h16_t*)y;
  p.x_ld = d->x_ld;
  p.y_ld = d->y_ld;
  if (dw3_applicable(p, x, y, nullptr)) {
    Dw3Params q{};
    q.in = p.x; q.w = w; q.bias = bias; q.out = p.y;
    q.act = act; q.ap = ap;
    q.N = p.N; q.C = p.C; q.IH = p.H; q.IW = p.W; q.OH = p.P; q.OW = p.Q; q.ph = p.ph; q.pw = p.pw;
    q.in_ld = p.x_ld; q.out_ld = p.y_ld; q.flip = 0;
    int grid, lst = CVHIP_OK;
    if (dw3_lds_launch<0>(q, (hipStream_t)stream, &lst)) return lst;
    dw3_launch_geom(q, &grid);
    hipLaunchKernelGGL(dw3x3_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, q);
    return check_launch("dw3x3_kernel<0>");
  }
  {
    int tst = CVHIP_OK;
    if (dw_taps_launch(p, false, (hipStream_t)stream, &tst)) return tst;
  }
  hipLaunchKernelGGL(dw_fprop_kernel, dim3(grid_for((int64_t)p.N * p.P * p.Q * ((p.C + 7) / 8))), dim3(256), 0,
                     (hipStream_t)stream, p);
  return check_launch("dw_fprop_kernel");
}

int cvhip_dwconv2d_fprop(const cvhip_conv_desc* d, const void* x, const float* w, const float* bias, void* y, void* stream) {
  return dw_fprop_impl(d, x, w, bias, CVHIP_ACT_NONE, 0.f, y, stream);
}

// Forward + training-mode BatchNorm sums in one pass (3x3 / stride 1 / dilation 1 problems the LDS strip kernel runs: the decoder
// layers of DeepLabv3+, depthwise_separable_conv_module.py:10-99 in front of their norm layer). rows = 0: not such a problem — the caller
// runs cvhip_dwconv2d_fprop and the reduction pass. `x` / `y` only decide the alignment test of the query.
static int dw_stats_setup(const cvhip_conv_desc* d, const void* x, const void* y, DwParams* p, Dw3Params* q) {
  int st = fill(d, p);
  if (st) return st;
  p->x = (const h16_t*)x;
  p->y = (h16_t*)y;
  p->x_ld = d->x_ld;
  p->y_ld = d->y_ld;
  if (!dw3_applicable(*p, x, y, nullptr)) return CVHIP_ERR_UNSUPPORTED;
  *q = Dw3Params{};
  q->in = p->x; q->out = p->y;
  q->act = CVHIP_ACT_NONE;
  q->N = p->N; q->C = p->C; q->IH = p->H; q->IW = p->W; q->OH = p->P; q->OW = p->Q; q->ph = p->ph; q->pw = p->pw;
  q->in_ld = p->x_ld; q->out_ld = p->y_ld; q->flip = 0;
  return CVHIP_OK;
}

int64_t cvhip_dwconv2d_fprop_stats_rows(const cvhip_conv_desc* d, const void* x, const void* y) {
  DwParams p;
  Dw3Params q;
  const int st = dw_stats_setup(d, x, y, &p, &q);
  if (st == CVHIP_ERR_UNSUPPORTED) return 0;
  if (st) return st;
  int lst = CVHIP_OK;
  int64_t rows = 0;
  if (!dw3_lds_launch<0>(q, nullptr, &lst, &rows)) return 0;
  return rows;
}

int cvhip_dwconv2d_fprop_stats(const cvhip_conv_desc* d, const void* x, const float* w, const float* bias, void* y, float* stats_partial,
                               void* stream) {
  if (!x || !w || !y || !stats_partial) return CVHIP_ERR_INVALID;
  DwParams p;
  Dw3Params q;
  const int st = dw_stats_setup(d, x, y, &p, &q);
  if (st) return st;
  q.w = w;
  q.bias = bias;
  q.stats_partial = stats_partial;
  int lst = CVHIP_OK;
  if (dw3_lds_launch<0>(q, (hipStream_t)stream, &lst)) return lst;
  return CVHIP_ERR_UNSUPPORTED;
}

int cvhip_dwconv2d_fprop_act(const cvhip_conv_desc* d, const void* x, const float* w, const float* bias, int32_t act, float act_param, void* y,
                             void* stream) {
  if (act < CVHIP_ACT_NONE || act > CVHIP_ACT_HSWISH) return CVHIP_ERR_INVALID;
  return dw_fprop_impl(d, x, w, bias, act, act_param, y, stream);
}

int cvhip_dwconv2d_dgrad(const cvhip_conv_desc* d, const void* dy, const float* w, void* dx, void* stream) {
  DwParams p;
  int st = fill(d, &p);
  if (st) return st;
  if (!dy || !w || !dx) return CVHIP_ERR_INVALID;
  p.x = (const h16_t*)dy;
  p.w = w;
  p.y = (h16_t*)dx;
  p.x_ld = d->y_ld;  // pitch of dy
  p.y_ld = d->x_ld;  // pitch of dx
  if (dw3_applicable(p, dy, dx, nullptr)) {
    // stride-1 dgrad == correlation of dy with the flipped kernel and padding 2 - pad
    Dw3Params q{};
    q.in = p.x; q.w = w; q.bias = nullptr; q.out = p.y;
    q.N = p.N; q.C = p.C; q.IH = p.P; q.IW = p.Q; q.OH = p.H; q.OW = p.W; q.ph = 2 - p.ph; q.pw = 2 - p.pw;
    q.in_ld = p.x_ld; q.out_ld = p.y_ld; q.flip = 1;
    int grid, lst = CVHIP_OK;
    if (dw3_lds_launch<0>(q, (hipStream_t)stream, &lst)) return lst;
    dw3_launch_geom(q, &grid);
    hipLaunchKernelGGL(dw3x3_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, q);
    return check_launch("dw3x3_kernel<0>(dgrad)");
  }
  {
    int tst = CVHIP_OK;
    if (dw_taps_launch(p, true, (hipStream_t)stream, &tst)) return tst;
  }
  if (p.R == 3 && p.S == 3 && p.sh == 2 && p.sw == 2 && p.ph == 1 && p.pw == 1 && p.dh == 1 && p.dw_ == 1 && dw_vec_ok(p) &&
      p.P == (p.H - 1) / 2 + 1 && p.Q == (p.W - 1) / 2 + 1) {
    const int CV = p.C >> 3;
    const int rpp = 256 / (CV < 256 ? CV : 256);
    const int64_t npix = (int64_t)p.N * p.H * p.W;
    int ppb = rpp * 16;   // ~16 pixel visits per thread
    while (cdiv64(npix, ppb) > 256 * 32) ppb *= 2;
    hipLaunchKernelGGL(dw3x3_s2_dgrad_kernel, dim3((unsigned)cdiv64(npix, ppb)), dim3(256), 0, (hipStream_t)stream, p, ppb);
    return check_launch("dw3x3_s2_dgrad_kernel");
  }
  hipLaunchKernelGGL(dw_dgrad_kernel, dim3(grid_for((int64_t)p.N * p.H * p.W * ((p.C + 7) / 8))), dim3(256), 0,
                     (hipStream_t)stream, p);
  return check_launch("dw_dgrad_kernel");
}

int cvhip_dwconv2d_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* stream) {
  DwParams p;
  int st = fill(d, &p);
  if (st) return st;
  if (!x || !dy || !dw) return CVHIP_ERR_INVALID;
  if (p.R * p.S > kDwMaxTaps) return CVHIP_ERR_UNSUPPORTED;
  p.x = (const h16_t*)x;
  p.dy = (const h16_t*)dy;
  p.dw = dw;
  p.x_ld = d->x_ld;
  p.y_ld = d->y_ld;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    int zs = zero_fill(dw, sizeof(float) * (size_t)p.C * p.R * p.S, s);
    if (zs) return zs;
  }
  if (dw3_applicable(p, x, dy, nullptr)) {
    Dw3Params q{};
    q.in = p.x; q.dy = p.dy; q.dw = dw;
    q.N = p.N; q.C = p.C; q.IH = p.H; q.IW = p.W; q.OH = p.P; q.OW = p.Q; q.ph = p.ph; q.pw = p.pw;
    q.in_ld = p.x_ld; q.dy_ld = p.y_ld;
    int grid, lst = CVHIP_OK;
    if (dw3_lds_launch<1>(q, s, &lst)) return lst;
    dw3_launch_geom(q, &grid, true);
    hipLaunchKernelGGL(dw3x3_kernel<1>, dim3(grid), dim3(256), 0, s, q);
    return check_launch("dw3x3_kernel<1>");
  }
  const int64_t M = (int64_t)p.N * p.P * p.Q;
  const int CVh = (p.C + 7) / 8;
  const int colsh = CVh < kDwWgradCols ? CVh : kDwWgradCols;
  const int chunks = cdiv(CVh, colsh);
  const int rpp = 256 / colsh;
  int64_t blocks = cdiv64(M, (int64_t)rpp * 16);  // >= 16 row visits per thread
  const int64_t cap = 2048 / chunks > 1 ? 2048 / chunks : 1;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int rows_per_block = (int)cdiv64(M, blocks);
  hipLaunchKernelGGL(dw_wgrad_kernel, dim3((int)blocks, chunks), dim3(256), 0, s, p, rows_per_block);
  return check_launch("dw_wgrad_kernel");
}

}  // extern "C"
