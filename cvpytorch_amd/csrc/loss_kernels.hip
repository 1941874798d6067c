// loss_kernels.hip — loss-side kernels on NHWC bf16 activations.
//
//  * per-pixel softmax cross-entropy with ignore_index (nn.CrossEntropyLoss, reduction='mean'):
//    reference src/losses/seg/cross_entropy_loss.py:32-40 as called from
//    src/models/segmentors/encoder_decoder.py:93-107 (after the bilinear resize to label size).
//  * per-(sample, channel) scaling = Dropout2d apply / backward (src/models/heads/seg/base_seg_head.py:32-37).
#include "common.h"
#include "bilinear_index.h"

namespace cvhip {

constexpr int kCeMaxC = 64;

// logits: [M][ld] bf16 (C valid channels), target: int64 [M]; partial: [gridDim.x][2] fp32 = (sum of -log p_t, #valid)
__global__ __launch_bounds__(256) void seg_ce_fwd_kernel(const h16_t* __restrict__ logits, int ld, const int64_t* __restrict__ target, int64_t M,
                                                         int C, int ignore, float* __restrict__ partial) {
  __shared__ float red[2][256];
  float loss = 0.f, cnt = 0.f;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t t = target[m];
    if (t == ignore || t < 0 || t >= C) continue;
    const h16_t* row = logits + m * ld;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, (float)row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf((float)row[c] - mx);
    loss += (mx + __logf(se)) - (float)row[t];
    cnt += 1.f;
  }
  red[0][threadIdx.x] = loss;
  red[1][threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0];
    partial[2 * blockIdx.x + 1] = red[1][0];
  }
}

// out[0] = mean loss (sum/count, 0 if count==0), out[1] = count
__global__ void seg_ce_finalize_kernel(const float* partial, int rows, float* out) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    a += (double)partial[2 * r];
    b += (double)partial[2 * r + 1];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[1][0] > 0.0 ? (float)(red[0][0] / red[1][0]) : 0.f;
    out[1] = (float)red[1][0];
  }
}

// dlogits[m][c] = gscale[0] * (softmax_c - [c==t]) / count  for valid pixels, 0 otherwise; pad channels zero.
// A block owns 256 consecutive pixels = one contiguous run of 256*ld elements in both tensors: rows are staged through LDS
// so global loads / stores are coalesced whatever the (odd) class count — a thread-per-pixel walk of 38-byte rows ran at
// 0.3 TB/s (1.27 ms for 16x512x1024x19).
constexpr int kCeMaxLd = 32;
__global__ __launch_bounds__(256) void seg_ce_bwd_kernel(const h16_t* __restrict__ logits, int ld, const int64_t* __restrict__ target, int64_t M,
                                                         int C, int ignore, const float* __restrict__ stat, const float* __restrict__ gscale,
                                                         h16_t* __restrict__ dlogits, int ld_d) {
  __shared__ float tile[256 * kCeMaxLd];
  __shared__ h16_t tout[256 * kCeMaxLd];
  const float cnt = stat[1];
  const float g = (cnt > 0.f ? 1.f / cnt : 0.f) * (gscale ? gscale[0] : 1.f);
  const int t = threadIdx.x;
  for (int64_t m0 = (int64_t)blockIdx.x * 256; m0 < M; m0 += (int64_t)gridDim.x * 256) {
    const int rows = (int)(M - m0 < 256 ? M - m0 : 256);
    const int n_in = rows * ld;
    const h16_t* src = logits + m0 * ld;
    for (int i = t; i < n_in; i += 256) tile[i] = (float)src[i];
    __syncthreads();
    if (t < rows) {
      const int64_t tt = target[m0 + t];
      const float* row = tile + t * ld;
      h16_t* orow = tout + t * ld_d;
      if (tt == ignore || tt < 0 || tt >= C) {
        for (int c = 0; c < ld_d; ++c) orow[c] = (h16_t)0.f;
      } else {
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(row[c] - mx);
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) orow[c] = (h16_t)(g * (__expf(row[c] - mx) * inv - (c == (int)tt ? 1.f : 0.f)));
        for (int c = C; c < ld_d; ++c) orow[c] = (h16_t)0.f;
      }
    }
    __syncthreads();
    const int n_out = rows * ld_d;
    h16_t* dst = dlogits + m0 * ld_d;
    for (int i = t; i < n_out; i += 256) dst[i] = tout[i];
    __syncthreads();
  }
}

// generic fallback (pitches above kCeMaxLd)
__global__ __launch_bounds__(256) void seg_ce_bwd_rows_kernel(const h16_t* __restrict__ logits, int ld, const int64_t* __restrict__ target, int64_t M,
                                                              int C, int ignore, const float* __restrict__ stat, const float* __restrict__ gscale,
                                                              h16_t* __restrict__ dlogits, int ld_d) {
  const float cnt = stat[1];
  const float g = (cnt > 0.f ? 1.f / cnt : 0.f) * (gscale ? gscale[0] : 1.f);
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t t = target[m];
    h16_t* drow = dlogits + m * ld_d;
    if (t == ignore || t < 0 || t >= C) {
      for (int c = 0; c < ld_d; ++c) drow[c] = (h16_t)0.f;
      continue;
    }
    const h16_t* row = logits + m * ld;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, (float)row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf((float)row[c] - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
      const float p = __expf((float)row[c] - mx) * inv;
      drow[c] = (h16_t)(g * (p - (c == (int)t ? 1.f : 0.f)));
    }
    for (int c = C; c < ld_d; ++c) drow[c] = (h16_t)0.f;
  }
}

// ---- bilinear resize + cross-entropy in one pass (round 3) --------------------------------------------------------------------------
// encoder_decoder.py:93-107 resizes the 1/4-resolution logits to label size and takes the pixel-wise CE there. Done as two ops, the
// full-resolution logits (16 x 512 x 1024 x 24 bf16 = 400 MB for DeepLabv3+ at batch 16) are written by the resize, read by the CE
// forward, read again and their gradient written by the CE backward, and that is read by the resize backward: 2 GB of traffic and 2.7 ms
// of a 36 ms step for a tensor nobody needs. Here the label-resolution logits exist only in registers:
//   forward : one thread per label pixel interpolates its C logits from the 4 source pixels (L2-resident: the low-resolution
//             tensor is 25 MB), reduces max / sum-exp and adds -log p_target to the block's partial;
//   backward: one block per TH x TW tile of LOW-resolution pixels re-derives softmax - onehot for every label pixel of the tile's
//             footprint (the pixels whose interpolation reads a tile pixel; halo pixels are recomputed by the neighbouring blocks:
//             1.4x the pixels for x4 upsampling) into an LDS tile, then every (pixel, class) of the tile gathers its footprint with
//             the interpolation weights. No atomics: each low-resolution gradient element is written once, in a fixed order.
// The interpolated logits are not rounded to 16 bits in between (the two-op form rounds them and their gradient).
struct CeBilParams {
  const h16_t* x;  // [N][Hi][Wi][ld_x] low-resolution logits
  int ld_x;
  const int64_t* target;  // [N][Ho][Wo]
  int N, C, Hi, Wi, Ho, Wo, align, ignore;
  float sh, sw;
  float* partial;  // forward: [gridDim.x][2] = (sum of -log p_t, #valid)
  const float* stat;    // backward: forward's (mean loss, count)
  const float* gscale;  // backward: upstream gradient of the mean loss (1 element) or NULL
  h16_t* dx;            // backward: [N][Hi][Wi][ld_dx]
  int ld_dx;
  // per-pixel forms (round 6: OHEM, cross_entropy_loss.py:51-69 — the selection of hard pixels happens between the two passes)
  float* loss_px;       // forward: [N][Ho][Wo] -log p_t of every label pixel (0 where ignored); no reduction
  const float* w_px;    // backward: [N][Ho][Wo] weights in [0, 1]: dx = gscale * sum_m w_px[m] * d(-log p_t(m)) / dx (no 1 / #valid)
  // OHEM selection on the device (cvhip_ohem_select): the weight of a pixel follows from its forward loss and the selection record
  const float* sel;     // float[8]: [0] loss value, [1] backward scalar, [2] weight of the tied pixels, [3] threshold branch?, [4] v, [5] thr
  const float* sel_loss;  // [N][Ho][Wo] the forward's per-pixel losses (loss_px of fwd_px)
  float sel_lw;           // loss_weight: the selection compares loss_px * loss_weight
  int tiles_h, tiles_w, fh_max, fw_max;
  int ch_rows;   // backward: footprint rows whose softmax - onehot sit in the LDS at a time (the gather accumulates over the chunks)
};

// interpolated logits of label pixel (oy, ox): z[c] = a0*(b0*v00 + b1*v01) + a1*(b0*v10 + b1*v11) (bilinear_fwd_kernel's expression)
// `img` holds source rows [h_org, ...) x columns [w_org, w_org + pw) with `ld` elements per pixel: the image in global memory
// (h_org = w_org = 0, pw = Wi) or a patch of it staged in the LDS
template <int CMAX>
__device__ __forceinline__ void cebil_logits(const CeBilParams& p, const h16_t* img, int h_org, int w_org, int pw, int ld, bool vec, int oy, int ox,
                                             float (&z)[CMAX]) {
  int h0, h1, w0, w1;
  float lh, lw;
  bil_src(oy, p.sh, p.align, p.Hi, &h0, &h1, &lh);
  bil_src(ox, p.sw, p.align, p.Wi, &w0, &w1, &lw);
  const h16_t* r00 = img + ((int64_t)(h0 - h_org) * pw + (w0 - w_org)) * ld;
  const h16_t* r01 = img + ((int64_t)(h0 - h_org) * pw + (w1 - w_org)) * ld;
  const h16_t* r10 = img + ((int64_t)(h1 - h_org) * pw + (w0 - w_org)) * ld;
  const h16_t* r11 = img + ((int64_t)(h1 - h_org) * pw + (w1 - w_org)) * ld;
  const float a0 = 1.f - lh, a1 = lh, b0 = 1.f - lw, b1 = lw;
#pragma unroll
  for (int v = 0; v < CMAX / 8; ++v) {
    f32x8 q00, q01, q10, q11;
    if (vec) {  // pad lanes carry whatever the buffer holds: never used (c < C below)
      q00 = unpack8(*reinterpret_cast<const uint4*>(r00 + v * 8));
      q01 = unpack8(*reinterpret_cast<const uint4*>(r01 + v * 8));
      q10 = unpack8(*reinterpret_cast<const uint4*>(r10 + v * 8));
      q11 = unpack8(*reinterpret_cast<const uint4*>(r11 + v * 8));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = v * 8 + j < p.C;
        q00.v[j] = ok ? (float)r00[v * 8 + j] : 0.f;
        q01.v[j] = ok ? (float)r01[v * 8 + j] : 0.f;
        q10.v[j] = ok ? (float)r10[v * 8 + j] : 0.f;
        q11.v[j] = ok ? (float)r11[v * 8 + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) z[v * 8 + j] = a0 * (b0 * q00.v[j] + b1 * q01.v[j]) + a1 * (b0 * q10.v[j] + b1 * q11.v[j]);
  }
}

// the same from an fp32 patch in the LDS (rows [h_org, ..) x columns [w_org, w_org + pw), ldp floats per pixel, 16-byte aligned): no
// 16-bit unpacking per tap — 4 taps x CMAX channels of shifts / masks were a quarter of the backward kernel's instructions
template <int CMAX>
__device__ __forceinline__ void cebil_logits_lds(const CeBilParams& p, const float* P, int h_org, int w_org, int pw, int ldp, int oy, int ox,
                                                 float (&z)[CMAX]) {
  int h0, h1, w0, w1;
  float lh, lw;
  bil_src(oy, p.sh, p.align, p.Hi, &h0, &h1, &lh);
  bil_src(ox, p.sw, p.align, p.Wi, &w0, &w1, &lw);
  const float* r00 = P + ((h0 - h_org) * pw + (w0 - w_org)) * ldp;
  const float* r01 = P + ((h0 - h_org) * pw + (w1 - w_org)) * ldp;
  const float* r10 = P + ((h1 - h_org) * pw + (w0 - w_org)) * ldp;
  const float* r11 = P + ((h1 - h_org) * pw + (w1 - w_org)) * ldp;
  const float a0 = 1.f - lh, a1 = lh, b0 = 1.f - lw, b1 = lw;
  const float c00 = a0 * b0, c01 = a0 * b1, c10 = a1 * b0, c11 = a1 * b1;
#pragma unroll
  for (int v = 0; v < CMAX / 4; ++v) {
    const float4 q00 = *reinterpret_cast<const float4*>(r00 + v * 4);
    const float4 q01 = *reinterpret_cast<const float4*>(r01 + v * 4);
    const float4 q10 = *reinterpret_cast<const float4*>(r10 + v * 4);
    const float4 q11 = *reinterpret_cast<const float4*>(r11 + v * 4);
    z[v * 4 + 0] = (c00 * q00.x + c01 * q01.x) + (c10 * q10.x + c11 * q11.x);
    z[v * 4 + 1] = (c00 * q00.y + c01 * q01.y) + (c10 * q10.y + c11 * q11.y);
    z[v * 4 + 2] = (c00 * q00.z + c01 * q01.z) + (c10 * q10.z + c11 * q11.z);
    z[v * 4 + 3] = (c00 * q00.w + c01 * q01.w) + (c10 * q10.w + c11 * q11.w);
  }
}

template <int CMAX>
__global__ __launch_bounds__(256) void seg_ce_bilinear_fwd_kernel(const CeBilParams p) {
  __shared__ float red[2][256];
  float loss = 0.f, cnt = 0.f;
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < total; m += (int64_t)gridDim.x * 256) {
    const int64_t t = p.target[m];
    if (t == p.ignore || t < 0 || t >= p.C) {
      if (p.loss_px) p.loss_px[m] = 0.f;
      continue;
    }
    const int ox = (int)(m % p.Wo);
    const int64_t q = m / p.Wo;
    const int oy = (int)(q % p.Ho);
    const int n = (int)(q / p.Ho);
    float z[CMAX];
    const bool vec = (p.ld_x & 7) == 0 && p.ld_x >= CMAX && ((uintptr_t)p.x & 15) == 0;
    cebil_logits<CMAX>(p, p.x + (int64_t)n * p.Hi * p.Wi * p.ld_x, 0, 0, p.Wi, p.ld_x, vec, oy, ox, z);
    float mx = -INFINITY, zt = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < p.C) {
        mx = fmaxf(mx, z[c]);
        if (c == (int)t) zt = z[c];
      }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < p.C) se += __expf(z[c] - mx);
    const float lm = (mx + __logf(se)) - zt;
    if (p.loss_px) p.loss_px[m] = lm;
    loss += lm;
    cnt += 1.f;
  }
  if (p.loss_px) return;   // (per-pixel form: the caller reduces what it selects)
  red[0][threadIdx.x] = loss;
  red[1][threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p.partial[2 * blockIdx.x] = red[0][0];
    p.partial[2 * blockIdx.x + 1] = red[1][0];
  }
}

// low-resolution tile of a backward block: 4 x 8 source pixels; smaller tiles (round 6) where the label footprint of that tile would not
// fit the LDS — the x8 up-sampling of the STDC heads (64 x 128 logits against 512 x 1024 labels) runs 4 x 4
constexpr int kCeBilTiles[4][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}};

template <int CMAX, int TH, int TW>
__global__ __launch_bounds__(256) void seg_ce_bilinear_bwd_kernel(const CeBilParams p) {
  extern __shared__ __attribute__((aligned(16))) float cebil_smem[];
  // LDS: G[ch_rows * fw_max][C] (fp16: softmax - onehot lies in [-1, 1], 11 significant bits there; fp32 cost half the resident
  // blocks) | WY[fh_max][TH] | WX[fw_max][TW] | P[(TH + 2) * (TW + 2)][ldp] (the source patch, 16-bit).
  // Round 6: G holds a CHUNK of ch_rows footprint rows; the gather below is a sum over footprint rows, so it accumulates chunk after
  // chunk in registers. The footprint no longer has to fit the LDS: x8 / x16 up-sampling (the STDC heads) runs 8 x 8 source tiles whose
  // footprints overlap their neighbours' by a third instead of 2 x 2 tiles that recomputed every label pixel's softmax 2.4 times.
  _Float16* const G = reinterpret_cast<_Float16*>(cebil_smem);
  float* const WY = cebil_smem + ((((size_t)p.ch_rows * p.fw_max * p.C + 1) / 2 + 3) & ~(size_t)3);  // (multiples of 4 floats: P is 16-byte aligned)
  float* const WX = WY + p.fh_max * TH;
  const int ldp = CMAX;  // patch pitch in floats (channels >= ld_x are zero)
  float* const P = WX + (((p.fw_max * TW) + 3) & ~3);
  __shared__ int geo[4 + 2 * TW];  // oy_lo, FH, ox_lo, FW, then per tile column the [first, last] footprint column with a non-zero weight
  const int t = threadIdx.x;
  int b = blockIdx.x;
  const int tj = b % p.tiles_w;
  b /= p.tiles_w;
  const int ti = b % p.tiles_h;
  const int n = b / p.tiles_h;
  const int i0 = ti * TH, j0 = tj * TW;
  const int i1 = min(i0 + TH, p.Hi) - 1, j1 = min(j0 + TW, p.Wi) - 1;  // last tile row / column inside the image

  if (t == 0) {
    int lo, hi, l2, h2, a, bb;
    float l;
    bil_range(i0, p.sh, p.align, p.Ho, &lo, &h2);
    bil_range(i1, p.sh, p.align, p.Ho, &l2, &hi);
    // tighten the conservative bounds to the exact footprint: label rows that read a tile row
    for (; lo < hi; ++lo) {
      bil_src(lo, p.sh, p.align, p.Hi, &a, &bb, &l);
      if (bb >= i0) break;
    }
    for (; hi > lo; --hi) {
      bil_src(hi, p.sh, p.align, p.Hi, &a, &bb, &l);
      if (a <= i1) break;
    }
    geo[0] = lo;
    geo[1] = min(hi - lo + 1, p.fh_max);
  } else if (t == 64) {
    int lo, hi, l2, h2, a, bb;
    float l;
    bil_range(j0, p.sw, p.align, p.Wo, &lo, &h2);
    bil_range(j1, p.sw, p.align, p.Wo, &l2, &hi);
    for (; lo < hi; ++lo) {
      bil_src(lo, p.sw, p.align, p.Wi, &a, &bb, &l);
      if (bb >= j0) break;
    }
    for (; hi > lo; --hi) {
      bil_src(hi, p.sw, p.align, p.Wi, &a, &bb, &l);
      if (a <= j1) break;
    }
    geo[2] = lo;
    geo[3] = min(hi - lo + 1, p.fw_max);
  } else if (t >= 128 && t < 128 + TW) {
    geo[4 + 2 * (t - 128)] = 1 << 30;
    geo[5 + 2 * (t - 128)] = -1;
  }
  // source patch: every footprint pixel interpolates from rows [i0 - 1, i1 + 1] x columns [j0 - 1, j1 + 1] (clamped to the image).
  // Staged once, coalesced: the 4 taps x 3 vectors of a footprint pixel then come from the LDS instead of 12 dependent L2 round
  // trips behind the label load (the unstaged form ran at a third of the forward kernel's pixel rate)
  const int ph0 = max(i0 - 1, 0), pw0 = max(j0 - 1, 0);
  const int prow = min(i1 + 1, p.Hi - 1) - ph0 + 1, pcol = min(j1 + 1, p.Wi - 1) - pw0 + 1;
  {
    const h16_t* img = p.x + (int64_t)n * p.Hi * p.Wi * p.ld_x;
    const bool vsrc = (p.ld_x & 7) == 0 && ((uintptr_t)p.x & 15) == 0;
    constexpr int vpp = CMAX / 8;
    for (int e = t; e < prow * pcol * vpp; e += 256) {
      const int v = e % vpp, px = e / vpp;
      const int pc = px % pcol, pr = px / pcol;
      const h16_t* src = img + ((int64_t)(ph0 + pr) * p.Wi + pw0 + pc) * p.ld_x + v * 8;
      f32x8 q;
      if (vsrc && v * 8 + 8 <= p.ld_x) {
        q = unpack8(*reinterpret_cast<const uint4*>(src));
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) q.v[k] = v * 8 + k < p.ld_x ? (float)src[k] : 0.f;
      }
      float* dst = P + (size_t)px * ldp + v * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(q.v[0], q.v[1], q.v[2], q.v[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(q.v[4], q.v[5], q.v[6], q.v[7]);
    }
  }
  __syncthreads();
  const int oy_lo = geo[0], FH = geo[1], ox_lo = geo[2], FW = geo[3];
  // interpolation weight of footprint row f for tile row li (0 when the row does not read it), and the same for columns
  for (int e = t; e < FH * TH; e += 256) {
    const int f = e / TH, li = e - f * TH;
    int a, bb;
    float l;
    bil_src(oy_lo + f, p.sh, p.align, p.Hi, &a, &bb, &l);
    float w = 0.f;
    if (a == i0 + li) w += 1.f - l;
    if (bb == i0 + li) w += l;
    WY[e] = w;
  }
  for (int e = t; e < FW * TW; e += 256) {
    const int f = e / TW, lj = e - f * TW;
    int a, bb;
    float l;
    bil_src(ox_lo + f, p.sw, p.align, p.Wi, &a, &bb, &l);
    float w = 0.f;
    if (a == j0 + lj) w += 1.f - l;
    if (bb == j0 + lj) w += l;
    WX[e] = w;
    if (w != 0.f) {  // a tile column's footprint columns are contiguous (src is monotone): keep [first, last]
      atomicMin(&geo[4 + 2 * lj], f);
      atomicMax(&geo[5 + 2 * lj], f);
    }
  }
  __syncthreads();   // WY / WX and the column ranges are complete
  // gather ownership: a thread owns (tile column lj, class c) — c fastest, ld_dx entries per pixel, the pad channels are written as
  // zeros — and keeps its TH row sums in registers across the chunks (the launcher guarantees TW * ld_dx <= 256: one output per thread)
  const int per_px = p.ld_dx;
  const int o = t;
  const int oc = o % per_px, olj = o / per_px;
  const bool owner = o < TW * per_px && j0 + olj < p.Wi;
  float acc[TH];
#pragma unroll
  for (int li = 0; li < TH; ++li) acc[li] = 0.f;
  const int fx0 = owner ? geo[4 + 2 * olj] : 0, fx1 = owner ? geo[5 + 2 * olj] : -1;
  const int64_t* tg = p.target + (int64_t)n * p.Ho * p.Wo;
  for (int f0 = 0; f0 < FH; f0 += p.ch_rows) {
    const int f1 = min(f0 + p.ch_rows, FH);
    // softmax - onehot of the chunk's footprint pixels (0 for ignored labels and for pixels the selection dropped)
    for (int f = t; f < (f1 - f0) * FW; f += 256) {
      const int fyl = f / FW, fx = f - fyl * FW;
      const int oy = oy_lo + f0 + fyl, ox = ox_lo + fx;
      const int64_t tt = tg[(int64_t)oy * p.Wo + ox];
      _Float16* const g = G + (size_t)f * p.C;
      float wm = 1.f;
      if (p.w_px) wm = p.w_px[((int64_t)n * p.Ho + oy) * p.Wo + ox];
      else if (p.sel) {
        const float l = p.sel_loss[((int64_t)n * p.Ho + oy) * p.Wo + ox] * p.sel_lw;
        wm = p.sel[3] != 0.f ? (l > p.sel[5] ? 1.f : 0.f) : (l > p.sel[4] ? 1.f : (l == p.sel[4] ? p.sel[2] : 0.f));
      }
      if (tt == p.ignore || tt < 0 || tt >= p.C || wm == 0.f) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < p.C) g[c] = (_Float16)0.f;
        continue;
      }
      float z[CMAX];
      cebil_logits_lds<CMAX>(p, P, ph0, pw0, pcol, ldp, oy, ox, z);
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.C) mx = fmaxf(mx, z[c]);
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.C) {
          z[c] = __expf(z[c] - mx);
          se += z[c];
        }
      const float inv = 1.f / se;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.C) g[c] = (_Float16)((z[c] * inv - (c == (int)tt ? 1.f : 0.f)) * wm);
    }
    __syncthreads();
    // gather over the chunk's rows: the row sums over the footprint columns are shared by the TH tile rows, the <= 2r independent LDS
    // reads of a row are in flight together
    if (owner && oc < p.C) {
      for (int fy = f0; fy < f1; ++fy) {
        const _Float16* const g = G + (size_t)(fy - f0) * FW * p.C + oc;
        float row = 0.f;
#pragma unroll 4
        for (int fx = fx0; fx <= fx1; ++fx) row += WX[fx * TW + olj] * (float)g[(size_t)fx * p.C];
#pragma unroll
        for (int li = 0; li < TH; ++li) acc[li] += WY[fy * TH + li] * row;
      }
    }
    __syncthreads();   // (the next chunk overwrites G)
  }
  const float cnt = (p.w_px || p.sel) ? 1.f : p.stat[1];
  const float gs = (cnt > 0.f ? 1.f / cnt : 0.f) * (p.gscale ? p.gscale[0] : 1.f) * (p.sel ? p.sel[1] : 1.f);
  if (owner) {
#pragma unroll
    for (int li = 0; li < TH; ++li)
      if (i0 + li < p.Hi) p.dx[((int64_t)(n * p.Hi + i0 + li) * p.Wi + j0 + olj) * p.ld_dx + oc] = (h16_t)(acc[li] * gs);
  }
}

// ---- OHEM selection on the device (round 6) ------------------------------------------------------------------------------------------
// cross_entropy_loss.py:51-69 sorts all per-pixel losses and branches on loss_sorted[min_kept] > thresh. Everything it needs is the
// value v = the (min_kept + 1)-th largest loss and five masked sums (segmentors.OhemCrossEntropyLoss2d has the derivation): v by a
// three-pass radix select on the order-preserving integer image of the fp32 losses (12 + 12 + 8 bits: block-local LDS histograms, one
// single-block scan per pass), the sums by one deterministic two-stage reduction. Four reads of the 33-MB loss vector instead of a
// full radix sort (torch.topk) plus ~15 element-wise / reduction launches.
__device__ __forceinline__ unsigned ohem_key(float l) {
  const unsigned b = __float_as_uint(l);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);   // larger float <=> larger key
}
__device__ __forceinline__ float ohem_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

struct OhemWs {
  unsigned hist[4096];
  unsigned prefix;   // key bits fixed so far (aligned at the top)
  unsigned rank;     // descending rank still to descend inside the fixed prefix
  unsigned pad[2];
  double partial[1024][5];
};

template <int PASS>
__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* __restrict__ loss, int64_t M, float lw, OhemWs* ws) {
  __shared__ unsigned h[4096];
  constexpr int NB = PASS == 2 ? 256 : 4096;
  for (int i = threadIdx.x; i < NB; i += 256) h[i] = 0u;
  __syncthreads();
  const unsigned prefix = PASS == 0 ? 0u : ws->prefix;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const unsigned k = ohem_key(loss[m] * lw);
    if (PASS == 0) atomicAdd(&h[k >> 20], 1u);
    else if (PASS == 1) {
      if ((k >> 20) == (prefix >> 20)) atomicAdd(&h[(k >> 8) & 0xfffu], 1u);
    } else {
      if ((k >> 8) == (prefix >> 8)) atomicAdd(&h[k & 0xffu], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += 256)
    if (h[i]) atomicAdd(&ws->hist[i], h[i]);
}

// one block: the bin (descending) in which the running count passes the rank; fixes its bits, leaves the rank inside it, clears the
// histogram for the next pass
template <int PASS>
__global__ __launch_bounds__(256) void ohem_scan_kernel(OhemWs* ws, unsigned rank0) {
  constexpr int NB = PASS == 2 ? 256 : 4096;
  constexpr int PER = NB / 256;
  __shared__ unsigned chunk[256];
  __shared__ unsigned found[2];
  const int t = threadIdx.x;
  const unsigned rank = PASS == 0 ? rank0 : ws->rank;
  // thread t owns bins [NB - 1 - t * PER, ...) descending
  unsigned s = 0;
  for (int j = 0; j < PER; ++j) s += ws->hist[NB - 1 - (t * PER + j)];
  chunk[t] = s;
  __syncthreads();
  if (t == 0) {
    unsigned cum = 0;
    int c = 0;
    for (; c < 256; ++c) {
      if (cum + chunk[c] > rank) break;
      cum += chunk[c];
    }
    if (c == 256) c = 255;   // (rank >= count: cannot happen for rank < M)
    int bin = NB - 1 - c * PER;
    for (int j = 0; j < PER; ++j) {
      const unsigned hcount = ws->hist[NB - 1 - (c * PER + j)];
      bin = NB - 1 - (c * PER + j);
      if (cum + hcount > rank) break;
      cum += hcount;
    }
    found[0] = (unsigned)bin;
    found[1] = rank - cum;
  }
  __syncthreads();
  for (int i = t; i < NB; i += 256) ws->hist[i] = 0u;
  if (t == 0) {
    const unsigned prev = PASS == 0 ? 0u : ws->prefix;
    ws->prefix = PASS == 0 ? (found[0] << 20) : PASS == 1 ? (prev | (found[0] << 8)) : (prev | found[0]);
    ws->rank = found[1];
  }
}

__global__ __launch_bounds__(256) void ohem_stats_kernel(const float* __restrict__ loss, int64_t M, float lw, float thr, OhemWs* ws) {
  __shared__ double red[5][256];
  const float v = ohem_unkey(ws->prefix);
  double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const float l = loss[m] * lw;
    if (l > thr) {
      a[0] += (double)l;
      a[1] += 1.0;
    }
    if (l > v) {
      a[2] += (double)l;
      a[3] += 1.0;
    } else if (l == v) a[4] += 1.0;
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = a[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) ws->partial[blockIdx.x][threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void ohem_finalize_kernel(OhemWs* ws, int nblocks, int min_kept, float thr, float lw, float* sel) {
  __shared__ double red[5][256];
  double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int q = 0; q < 5; ++q) a[q] += ws->partial[b][q];
#pragma unroll
  for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = a[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = ohem_unkey(ws->prefix);
    const double s_above = red[0][0], n_above = red[1][0] > 1.0 ? red[1][0] : 1.0, s_gt = red[2][0], n_gt = red[3][0];
    const double n_ties = red[4][0] > 1.0 ? red[4][0] : 1.0;
    const bool hard = v > thr;
    const double mean_above = s_above / n_above;
    const double mean_top = (s_gt + ((double)min_kept - n_gt) * (double)v) / (double)min_kept;
    double tie = ((double)min_kept - n_gt) / n_ties;
    tie = tie < 0.0 ? 0.0 : (tie > 1.0 ? 1.0 : tie);
    sel[0] = (float)(hard ? mean_above : mean_top);
    sel[1] = (float)((hard ? 1.0 / n_above : 1.0 / (double)min_kept) * (double)lw);
    sel[2] = (float)tie;
    sel[3] = hard ? 1.f : 0.f;
    sel[4] = v;
    sel[5] = thr;
    sel[6] = (float)red[3][0];
    sel[7] = (float)red[4][0];
  }
}

// ---- boundary targets of the STDC detail loss (round 6) ------------------------------------------------------------------------------
// detail_loss.py:37-79: three Laplacian convolutions of the LABEL map (3x3 kernel [-1 .. 8 .. -1], padding 1, strides 1 / 2 / 4,
// clamp(min=0)), the strided ones brought back to label size by nearest up-sampling, each thresholded to {0, 1}; the three planes fused
// by a 1x1 convolution with weights 0.6 / 0.3 / 0.1 and thresholded again. Written with torch ops that was three F.conv2d on
// [N, 1, H, W] fp32 tensors (MIOpen's naive direct kernel: 2 ms each at 16 x 512 x 1024), two F.interpolate, a cat and a fourth conv
// — per step, without a gradient. One thread per label pixel here; the sums are small integers, exact in fp32.
__device__ __forceinline__ float detail_lap(const int64_t* __restrict__ g, int H, int W, int cy, int cx) {
  // sum_{dy, dx} k[dy][dx] * g[cy + dy][cx + dx] with zero padding, k = 8 at the centre and -1 around it
  float s = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int y = cy + dy, x = cx + dx;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
        const float v = (float)g[(int64_t)y * W + x];
        s += (dy == 0 && dx == 0) ? 8.f * v : -v;
      }
    }
  return s;
}

__global__ __launch_bounds__(256) void detail_targets_kernel(const int64_t* __restrict__ labels, int N, int H, int W, float thr, float* __restrict__ out) {
  const unsigned total = (unsigned)N * (unsigned)H * (unsigned)W;
  // strided outputs: floor((H + 2 - 3) / s) + 1 rows; F.interpolate(mode="nearest") reads src = min(floor(dst * in / out), in - 1)
  const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1, H4 = (H - 1) / 4 + 1, W4 = (W - 1) / 4 + 1;
  const float sh2 = (float)H2 / (float)H, sw2 = (float)W2 / (float)W, sh4 = (float)H4 / (float)H, sw4 = (float)W4 / (float)W;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / (unsigned)W, x = i - r * (unsigned)W;
    const unsigned n = r / (unsigned)H, y = r - n * (unsigned)H;
    const int64_t* g = labels + (int64_t)n * H * W;
    const float l1 = fmaxf(detail_lap(g, H, W, (int)y, (int)x), 0.f) > thr ? 1.f : 0.f;
    int y2 = (int)floorf((float)y * sh2), x2 = (int)floorf((float)x * sw2);
    y2 = y2 < H2 - 1 ? y2 : H2 - 1;
    x2 = x2 < W2 - 1 ? x2 : W2 - 1;
    const float l2 = fmaxf(detail_lap(g, H, W, 2 * y2, 2 * x2), 0.f) > thr ? 1.f : 0.f;
    int y4 = (int)floorf((float)y * sh4), x4 = (int)floorf((float)x * sw4);
    y4 = y4 < H4 - 1 ? y4 : H4 - 1;
    x4 = x4 < W4 - 1 ? x4 : W4 - 1;
    const float l4 = fmaxf(detail_lap(g, H, W, 4 * y4, 4 * x4), 0.f) > thr ? 1.f : 0.f;
    const float pyr = (0.6f * l1 + 0.3f * l2) + 0.1f * l4;
    out[i] = pyr > thr ? 1.f : 0.f;
  }
}

// y[n][hw][c] = x[n][hw][c] * s[n][c]   (Dropout2d's per-(image, channel) mask scale, SE / attention gating)
// Round 4: a block owns `chunk` pixel rows of ONE image; a thread owns one 16-byte channel vector, keeps its 8 scales in registers and
// walks the rows four per trip with all loads of a trip issued before the arithmetic (the grid-stride form re-decoded (n, hw, cv) and
// re-read its 8 scales per element with ONE row in flight per lane: 232 us for 268 MB tensors = 2.3 TB/s in the DeepLabv3+ step).
__global__ __launch_bounds__(256) void scale_nc_kernel(const h16_t* __restrict__ x, int ld_x, const float* __restrict__ s, h16_t* __restrict__ y,
                                                       int ld_y, int N, int C, int HW, int chunks) {
  const int CV = (C + 7) >> 3;
  const bool vec = (C & 7) == 0 && (ld_x & 7) == 0 && (ld_y & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int cols = CV < 256 ? CV : 256;
  const int rpp = 256 / cols;
  const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
  if (ty >= rpp || n >= N) return;
  const int rows = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rows;
  const int r1 = r0 + rows < HW ? r0 + rows : HW;
  const h16_t* const xn = x + (int64_t)n * HW * ld_x;
  h16_t* const yn = y + (int64_t)n * HW * ld_y;
  for (int cv = tx; cv < CV; cv += cols) {
    const int c = cv * 8;
    float sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = c + j < C ? s[(int64_t)n * C + c + j] : 0.f;
    int r = r0 + ty;
    if (vec) {
      for (; r < r1; r += 4 * rpp) {
        uint4 u[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rq = r + q * rpp;
          ok[q] = rq < r1;
          u[q] = *reinterpret_cast<const uint4*>(xn + (int64_t)(ok[q] ? rq : r) * ld_x + c);
        }
        __builtin_amdgcn_sched_barrier(0);  // the four loads first
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x8 v = unpack8(u[q]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v.v[j] *= sc[j];
          if (ok[q]) *reinterpret_cast<uint4*>(yn + (int64_t)(r + q * rpp) * ld_y + c) = pack8(v);
        }
      }
    } else {
      for (; r < r1; r += rpp)
        for (int j = 0; j < 8 && c + j < C; ++j) yn[(int64_t)r * ld_y + c + j] = (h16_t)((float)xn[(int64_t)r * ld_x + c + j] * sc[j]);
    }
  }
}

static inline int grid_for(int64_t total) {
  int64_t b = cdiv64(total, 256);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

template <int CMAX, int TH, int TW>
static int cebil_launch_bwd(const CeBilParams& p, int lds, hipStream_t s) {
  auto kern = seg_ce_bilinear_bwd_kernel<CMAX, TH, TW>;
  static bool attr_done[64] = {};  // per instantiation and device
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& done = attr_done[devid & 63];
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(seg_ce_bilinear_bwd_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.N * p.tiles_h * p.tiles_w), dim3(256), lds, s, p);
  return check_launch("seg_ce_bilinear_bwd_kernel");
}


// footprint bound (label rows read by TH consecutive source rows) for the LDS tile: exact footprints are <= this
static int cebil_fbound(int in, int out, int T) {
  const int r = (out + in - 1) / in;  // label pixels per source pixel, rounded up
  int f = (T + 1) * r + 2;
  return f < out ? f : out;
}

static int cebil_lds_bytes_tile(int C, int Hi, int Wi, int Ho, int Wo, int th, int tw, int ch_rows) {
  const int fh = cebil_fbound(Hi, Ho, th), fw = cebil_fbound(Wi, Wo, tw);
  const int ldp = C <= 24 ? 24 : 32;  // the kernel's CMAX
  return ((((ch_rows * fw * C + 1) / 2 + 3) & ~3) + fh * th + ((fw * tw + 3) & ~3)) * 4 + (th + 2) * (tw + 2) * ldp * 4;
}

// the largest backward tile that leaves room for a chunk of >= 8 footprint rows (or the whole footprint) inside the LDS budget — 64 KB:
// two resident blocks per CU — and keeps one gather output per thread (tw * ld_dx <= 256): index into kCeBilTiles, -1 = none
static int cebil_pick_tile(int C, int ld_dx, int Hi, int Wi, int Ho, int Wo, int* lds_out, int* ch_out) {
  constexpr int kBudget = 64 * 1024;
  for (int i = 0; i < 4; ++i) {
    const int th = kCeBilTiles[i][0], tw = kCeBilTiles[i][1];
    if (tw * ld_dx > 256) continue;
    const int fh = cebil_fbound(Hi, Ho, th);
    int ch = fh;
    while (ch > 1 && cebil_lds_bytes_tile(C, Hi, Wi, Ho, Wo, th, tw, ch) > kBudget) ch = (ch + 1) / 2;
    if (ch < 8 && ch < fh) continue;
    const int lds = cebil_lds_bytes_tile(C, Hi, Wi, Ho, Wo, th, tw, ch);
    if (lds > kBudget) continue;
    if (lds_out) *lds_out = lds;
    if (ch_out) *ch_out = ch;
    return i;
  }
  return -1;
}

template <int CMAX>
static int cebil_launch_bwd_tile(CeBilParams& p, hipStream_t s) {
  int lds = 0, ch = 0;
  const int ti = cebil_pick_tile(p.C, p.ld_dx, p.Hi, p.Wi, p.Ho, p.Wo, &lds, &ch);
  if (ti < 0) return CVHIP_ERR_UNSUPPORTED;
  p.ch_rows = ch;
  const int th = kCeBilTiles[ti][0], tw = kCeBilTiles[ti][1];
  p.tiles_h = (p.Hi + th - 1) / th;
  p.tiles_w = (p.Wi + tw - 1) / tw;
  p.fh_max = cebil_fbound(p.Hi, p.Ho, th);
  p.fw_max = cebil_fbound(p.Wi, p.Wo, tw);
  switch (ti) {
    case 0: return cebil_launch_bwd<CMAX, 8, 8>(p, lds, s);
    case 1: return cebil_launch_bwd<CMAX, 4, 8>(p, lds, s);
    case 2: return cebil_launch_bwd<CMAX, 4, 4>(p, lds, s);
    default: return cebil_launch_bwd<CMAX, 2, 4>(p, lds, s);
  }
}


}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_seg_ce_rows(int64_t M) { return grid_for(M); }

int cvhip_seg_ce_fwd(const void* logits, int32_t ld, const int64_t* target, int64_t M, int32_t C, int32_t ignore_index, float* partial,
                     float* out2, void* stream) {
  if (!logits || !target || !partial || !out2 || M <= 0 || C <= 0 || C > kCeMaxC * 64 || ld < C) return CVHIP_ERR_INVALID;
  const int rows = grid_for(M);
  hipLaunchKernelGGL(seg_ce_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const h16_t*)logits, ld, target, M, C, ignore_index,
                     partial);
  int st = check_launch("seg_ce_fwd_kernel");
  if (st) return st;
  hipLaunchKernelGGL(seg_ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, rows, out2);
  return check_launch("seg_ce_finalize_kernel");
}

int cvhip_seg_ce_bwd(const void* logits, int32_t ld, const int64_t* target, int64_t M, int32_t C, int32_t ignore_index, const float* out2,
                     const float* grad_scale, void* dlogits, int32_t ld_d, void* stream) {
  if (!logits || !target || !out2 || !dlogits || M <= 0 || C <= 0 || ld < C || ld_d < C) return CVHIP_ERR_INVALID;
  if (ld <= kCeMaxLd && ld_d <= kCeMaxLd) {
    int64_t blocks = (M + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(seg_ce_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const h16_t*)logits, ld, target, M, C,
                       ignore_index, out2, grad_scale, (h16_t*)dlogits, ld_d);
  } else {
    hipLaunchKernelGGL(seg_ce_bwd_rows_kernel, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)logits, ld, target, M, C,
                       ignore_index, out2, grad_scale, (h16_t*)dlogits, ld_d);
  }
  return check_launch("seg_ce_bwd_kernel");
}

int cvhip_seg_ce_bilinear_ok(int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t align_corners) {
  // align_corners: the backward's footprint bound (cebil_fbound) assumes the half-pixel mapping, whose label rows per source row are
  // out/in; with align_corners they are (out-1)/(in-1) > out/in and part of the gradient would be dropped — the two-op path runs
  if (align_corners) return 0;
  if (C <= 0 || C > 32 || Hi <= 0 || Wi <= 0 || Ho < Hi || Wo < Wi) return 0;  // upsampling only
  return cebil_pick_tile(C, (C + 7) & ~7, Hi, Wi, Ho, Wo, nullptr, nullptr) >= 0 ? 1 : 0;   // (the gradient's usual pitch: C rounded up to 8)
}

static int cebil_fill(CeBilParams& p, const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                      int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index) {
  if (!x || !target || N <= 0 || ld_x < C) return CVHIP_ERR_INVALID;
  if (!cvhip_seg_ce_bilinear_ok(C, Hi, Wi, Ho, Wo, align_corners)) return CVHIP_ERR_UNSUPPORTED;
  p.x = (const h16_t*)x;
  p.ld_x = ld_x;
  p.target = target;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.align = align_corners;
  p.ignore = ignore_index;
  bil_scales(Hi, Wi, Ho, Wo, align_corners, &p.sh, &p.sw);
  return CVHIP_OK;
}

int cvhip_seg_ce_bilinear_fwd(const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                              int32_t Wo, int32_t align_corners, int32_t ignore_index, float* partial, float* out2, void* stream) {
  CeBilParams p{};
  int st = cebil_fill(p, x, ld_x, target, N, C, Hi, Wi, Ho, Wo, align_corners, ignore_index);
  if (st) return st;
  if (!partial || !out2) return CVHIP_ERR_INVALID;
  p.partial = partial;
  const int rows = grid_for((int64_t)N * Ho * Wo);
  if (C <= 24) hipLaunchKernelGGL(seg_ce_bilinear_fwd_kernel<24>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(seg_ce_bilinear_fwd_kernel<32>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
  st = check_launch("seg_ce_bilinear_fwd_kernel");
  if (st) return st;
  hipLaunchKernelGGL(seg_ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, rows, out2);
  return check_launch("seg_ce_finalize_kernel");
}

int cvhip_seg_ce_bilinear_bwd(const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                              int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* out2, const float* grad_scale, void* dx,
                              int32_t ld_dx, void* stream) {
  CeBilParams p{};
  int st = cebil_fill(p, x, ld_x, target, N, C, Hi, Wi, Ho, Wo, align_corners, ignore_index);
  if (st) return st;
  if (!out2 || !dx || ld_dx < C) return CVHIP_ERR_INVALID;
  p.stat = out2;
  p.gscale = grad_scale;
  p.dx = (h16_t*)dx;
  p.ld_dx = ld_dx;
  if (C <= 24) return cebil_launch_bwd_tile<24>(p, (hipStream_t)stream);
  return cebil_launch_bwd_tile<32>(p, (hipStream_t)stream);
}

int cvhip_seg_ce_bilinear_fwd_px(const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                 int32_t Wo, int32_t align_corners, int32_t ignore_index, float* loss_px, void* stream) {
  CeBilParams p{};
  int st = cebil_fill(p, x, ld_x, target, N, C, Hi, Wi, Ho, Wo, align_corners, ignore_index);
  if (st) return st;
  if (!loss_px) return CVHIP_ERR_INVALID;
  p.loss_px = loss_px;
  const int rows = grid_for((int64_t)N * Ho * Wo);
  if (C <= 24) hipLaunchKernelGGL(seg_ce_bilinear_fwd_kernel<24>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(seg_ce_bilinear_fwd_kernel<32>, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("seg_ce_bilinear_fwd_kernel(px)");
}

int cvhip_seg_ce_bilinear_bwd_px(const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                 int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* w_px, const float* grad_scale, void* dx,
                                 int32_t ld_dx, void* stream) {
  CeBilParams p{};
  int st = cebil_fill(p, x, ld_x, target, N, C, Hi, Wi, Ho, Wo, align_corners, ignore_index);
  if (st) return st;
  if (!w_px || !dx || ld_dx < C) return CVHIP_ERR_INVALID;
  p.w_px = w_px;
  p.gscale = grad_scale;
  p.dx = (h16_t*)dx;
  p.ld_dx = ld_dx;
  if (C <= 24) return cebil_launch_bwd_tile<24>(p, (hipStream_t)stream);
  return cebil_launch_bwd_tile<32>(p, (hipStream_t)stream);
}

int64_t cvhip_ohem_select_workspace_bytes(void) { return (int64_t)sizeof(OhemWs); }

int cvhip_ohem_select(const float* loss_px, int64_t M, int32_t min_kept, float thresh_nlog, float loss_weight, void* workspace, float* sel8,
                      void* stream) {
  if (!loss_px || !workspace || !sel8 || M <= 0 || min_kept <= 0 || (int64_t)min_kept >= M || M >= (1ll << 32)) return CVHIP_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  OhemWs* ws = (OhemWs*)workspace;
  int st = zero_fill(ws, sizeof(unsigned) * 4096 + 16, s);
  if (st) return st;
  int64_t nb = cdiv64(M, 256 * 16);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(ohem_hist_kernel<0>, dim3((unsigned)nb), dim3(256), 0, s, loss_px, M, loss_weight, ws);
  hipLaunchKernelGGL(ohem_scan_kernel<0>, dim3(1), dim3(256), 0, s, ws, (unsigned)min_kept);
  hipLaunchKernelGGL(ohem_hist_kernel<1>, dim3((unsigned)nb), dim3(256), 0, s, loss_px, M, loss_weight, ws);
  hipLaunchKernelGGL(ohem_scan_kernel<1>, dim3(1), dim3(256), 0, s, ws, 0u);
  hipLaunchKernelGGL(ohem_hist_kernel<2>, dim3((unsigned)nb), dim3(256), 0, s, loss_px, M, loss_weight, ws);
  hipLaunchKernelGGL(ohem_scan_kernel<2>, dim3(1), dim3(256), 0, s, ws, 0u);
  hipLaunchKernelGGL(ohem_stats_kernel, dim3((unsigned)nb), dim3(256), 0, s, loss_px, M, loss_weight, thresh_nlog, ws);
  hipLaunchKernelGGL(ohem_finalize_kernel, dim3(1), dim3(256), 0, s, ws, (int)nb, min_kept, thresh_nlog, loss_weight, sel8);
  return check_launch("ohem_select");
}

int cvhip_seg_ce_bilinear_bwd_ohem(const void* x, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                   int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* loss_px, float loss_weight,
                                   const float* sel8, const float* grad_scale, void* dx, int32_t ld_dx, void* stream) {
  CeBilParams p{};
  int st = cebil_fill(p, x, ld_x, target, N, C, Hi, Wi, Ho, Wo, align_corners, ignore_index);
  if (st) return st;
  if (!loss_px || !sel8 || !dx || ld_dx < C) return CVHIP_ERR_INVALID;
  p.sel = sel8;
  p.sel_loss = loss_px;
  p.sel_lw = loss_weight;
  p.gscale = grad_scale;
  p.dx = (h16_t*)dx;
  p.ld_dx = ld_dx;
  if (C <= 24) return cebil_launch_bwd_tile<24>(p, (hipStream_t)stream);
  return cebil_launch_bwd_tile<32>(p, (hipStream_t)stream);
}

int cvhip_detail_boundary_targets(const int64_t* labels, int32_t N, int32_t H, int32_t W, float threshold, float* out, void* stream) {
  if (!labels || !out || N <= 0 || H <= 0 || W <= 0) return CVHIP_ERR_INVALID;
  const int64_t total = (int64_t)N * H * W;
  if (total >= (1ll << 31)) return CVHIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(detail_targets_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, labels, N, H, W, threshold, out);
  return check_launch("detail_targets_kernel");
}

int cvhip_scale_nc(const void* x, int32_t ld_x, const float* scale_nc, void* y, int32_t ld_y, int32_t N, int32_t C, int32_t HW, void* stream) {
  if (!x || !scale_nc || !y || N <= 0 || C <= 0 || HW <= 0) return CVHIP_ERR_INVALID;
  const int cv = (C + 7) / 8;
  const int rpp = 256 / (cv < 256 ? cv : 256);
  // ~16 row visits per thread, at most ~4096 blocks over the batch
  int chunks = (HW + rpp * 16 - 1) / (rpp * 16);
  const int cap = 4096 / N > 1 ? 4096 / N : 1;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(scale_nc_kernel, dim3((unsigned)N * (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ld_x, scale_nc,
                     (h16_t*)y, ld_y, N, C, HW, chunks);
  return check_launch("scale_nc_kernel");
}

}  // extern "C"
