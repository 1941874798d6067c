// yolo_loss.hip — YOLOv5 loss on device, fused per detection level: build_targets (anchor-ratio match, +-0.5-cell
// offset expansion, grid-index clamp), CIoU box loss, class BCE, objectness BCE with the IoU-scatter target, and their
// gradients, reading the bf16 NHWC head map directly and writing its bf16 gradient directly (no (N,3,H,W,85) fp32
// materialisation, no boolean-mask indexing, no host sync).
//
// Reference: src/losses/yolov5_loss.py:12-54 (bbox_iou, CIoU :36-49, eps 1e-7), :173-223 (__call__), :225-278 (build_targets:
// ratio test :247-248, offsets :233-259, long-cast + clamp :268-273). Candidate ordinal = (offset*A + anchor)*T + target
// reproduces the reference's row order after its boolean-mask filters; on duplicate cells its (CPU) index_put keeps the
// LAST writer for the objectness target (:199) => the candidate with the largest ordinal wins, while box / class
// gradients of all duplicates are summed (backward of the `pi[b, a, gj, gi]` gather).
//
// Index arithmetic must round exactly like the reference's fp32 tensor ops (t * gain, gxy - off, .long()), so FMA
// contraction is disabled for this translation unit.
#pragma clang fp contract(off)
#include "common.h"
#include "dual4.h"

namespace cvhip {

constexpr int kLossMaxA = 8;
constexpr int kCandStride = 4;  // floats of per-candidate scalars: iou, lbox, lcls, (unused)

struct YoloLossParams {
  const h16_t* raw;
  h16_t* draw;
  const float* targets;  // (T, 6) [img, cls, cx, cy, w, h] normalised; img < 0 = padding row
  const int* assign;     // optional [ncand]: externally decided positives (OTA, ota_assign.hip): matched target row or -1;
                         // the candidate keeps ITS cell (from its generating target) and takes box / class from the matched row
  int ld, N, A, NO, H, W, T, nc, ncand;
  float anchor_t;
  float anchors[kLossMaxA * 2];
  // workspace
  int* winner;     // [N*A*H*W] ordinal+1 of the last (largest-ordinal) valid candidate of the cell, 0 = none
  int* head;       // [N*A*H*W] linked list head (candidate+1), 0 = empty
  int* next;       // [ncand]
  int* cell;       // [ncand]   flat (n, a, gj, gi) index or -1
  float* cand;     // [ncand][kCandStride]
  float* cgrad;    // [ncand][gstride] unscaled d/dlogit: box channels 0-3 = -dCIoU/dlogit, class channels = sigmoid(x) - t
  int gstride;
  float* partial;  // [1024] objectness partial sums
  float* dobj;     // [N*A*H*W] sigmoid(x_obj) - t_obj per cell, kept by the forward objectness pass for the backward fill
  float* sums;     // [4] n, sum(1 - ciou), sum(cls bce), sum(obj bce)
  // backward scales
  const float* gout;  // upstream gradient of the total loss (device scalar) or nullptr (= 1)
  float k_box, k_cls, k_obj;  // hyp_box*bs, hyp_cls*bs/nc, hyp_obj*balance*bs/ncell
};

// CIoU of box1 = (x, y, w, h) [differentiated] and box2 = t (xywh); yolov5_loss.py:12-54 (x1y1x2y2=False, CIoU=True)
__device__ __forceinline__ D4 ciou_xywh(float x, float y, float w, float h, float tx, float ty, float tw, float th) {
  const float eps = 1e-7f;
  const D4 X = var(x, 0), Y = var(y, 1), Wd = var(w, 2), Hd = var(h, 3);
  const D4 b1x1 = X - scale(Wd, 0.5f), b1x2 = X + scale(Wd, 0.5f);
  const D4 b1y1 = Y - scale(Hd, 0.5f), b1y2 = Y + scale(Hd, 0.5f);
  const float b2x1 = tx - tw / 2, b2x2 = tx + tw / 2, b2y1 = ty - th / 2, b2y2 = ty + th / 2;
  const D4 iw = clamp0(dmin(b1x2, cst(b2x2)) - dmax(b1x1, cst(b2x1)));
  const D4 ih = clamp0(dmin(b1y2, cst(b2y2)) - dmax(b1y1, cst(b2y1)));
  const D4 inter = iw * ih;
  const D4 w1 = b1x2 - b1x1, h1 = (b1y2 - b1y1) + cst(eps);
  const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
  const D4 uni = (w1 * h1 + cst(w2 * h2)) - inter + cst(eps);
  const D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, cst(b2x2)) - dmin(b1x1, cst(b2x1));
  const D4 ch = dmax(b1y2, cst(b2y2)) - dmin(b1y1, cst(b2y1));
  const D4 c2 = cw * cw + ch * ch + cst(eps);
  const D4 dx = cst(b2x1 + b2x2) - b1x1 - b1x2;
  const D4 dy = cst(b2y1 + b2y2) - b1y1 - b1y2;
  const D4 rho2 = scale(dx * dx + dy * dy, 0.25f);
  const D4 da = cst(atanf(w2 / h2)) - datan(w1 / h1);
  const D4 v = scale(da * da, 0.40528473456935108578f);  // 4 / pi^2
  const float alpha = v.v / (v.v - iou.v + (1.f + eps));
  return iou - (rho2 / c2 + scale(v, alpha));
}

// ---- stage A: one wave per candidate ------------------------------------------------------------------
__global__ __launch_bounds__(256) void yolo_cand_kernel(const YoloLossParams p) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.ncand) return;
  const int T = p.T, A = p.A;
  const int off = c / (A * T);
  const int a = (c / T) % A;
  const int t = c % T;
  const float* tg = p.targets + (int64_t)t * 6;
  const float img = tg[0];
  const float nx = (float)p.W, ny = (float)p.H;
  const float gx = tg[2] * nx, gy = tg[3] * ny, gw = tg[4] * nx, gh = tg[5] * ny;
  const float aw = p.anchors[a * 2], ah = p.anchors[a * 2 + 1];
  const float rw = gw / aw, rh = gh / ah;
  const float rmax = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
  bool sel = (img >= 0.f) && (rmax < p.anchor_t);
  const float gxi = nx - gx, gyi = ny - gy;
  const float g = 0.5f;
  if (off == 1) sel = sel && ((gx - floorf(gx)) < g) && (gx > 1.f);
  else if (off == 2) sel = sel && ((gy - floorf(gy)) < g) && (gy > 1.f);
  else if (off == 3) sel = sel && ((gxi - floorf(gxi)) < g) && (gxi > 1.f);
  else if (off == 4) sel = sel && ((gyi - floorf(gyi)) < g) && (gyi > 1.f);
  float* cs = p.cand + (int64_t)c * kCandStride;
  const float* tm = tg;  // row that supplies the regression / class target
  if (p.assign) {
    const int m = p.assign[c];
    sel = m >= 0;
    if (sel) tm = p.targets + (int64_t)m * 6;
  }
  if (!sel) {
    if (lane == 0) {
      p.cell[c] = -1;
      cs[0] = 0.f;
      cs[1] = 0.f;
      cs[2] = 0.f;
    }
    return;
  }
  const float ox = (off == 1 ? 0.5f : (off == 3 ? -0.5f : 0.f)), oy = (off == 2 ? 0.5f : (off == 4 ? -0.5f : 0.f));
  int gi = (int)(gx - ox), gj = (int)(gy - oy);  // .long(): truncation toward zero
  gi = min(max(gi, 0), p.W - 1);
  gj = min(max(gj, 0), p.H - 1);
  const int b = (int)img;
  const int cell = ((b * A + a) * p.H + gj) * p.W + gi;
  if (lane == 0) {
    p.cell[c] = cell;
    atomicMax(p.winner + cell, c + 1);
    p.next[c] = atomicExch(p.head + cell, c + 1);
  }
  const h16_t* px = p.raw + ((int64_t)(b * p.H + gj) * p.W + gi) * p.ld + a * p.NO;
  // box (all lanes redundantly: 4 broadcast loads)
  const float r0 = (float)px[0], r1 = (float)px[1], r2 = (float)px[2], r3 = (float)px[3];
  const float s0 = sigmoid_ref(r0), s1 = sigmoid_ref(r1), s2 = sigmoid_ref(r2), s3 = sigmoid_ref(r3);
  const float bx = s0 * 2.f - 0.5f, by = s1 * 2.f - 0.5f;
  const float bw = (s2 * 2.f) * (s2 * 2.f) * aw, bh = (s3 * 2.f) * (s3 * 2.f) * ah;
  const float mx = tm[2] * nx, my = tm[3] * ny, mw = tm[4] * nx, mh = tm[5] * ny;  // == gx, gy, gw, gh without an assignment
  const D4 ci = ciou_xywh(bx, by, bw, bh, mx - (float)gi, my - (float)gj, mw, mh);
  float* gr = p.cgrad + (int64_t)c * p.gstride;
  if (lane == 0) {
    cs[0] = ci.v;
    cs[1] = 1.f - ci.v;
    gr[0] = -ci.d[0] * (2.f * s0 * (1.f - s0));
    gr[1] = -ci.d[1] * (2.f * s1 * (1.f - s1));
    gr[2] = -ci.d[2] * (8.f * s2 * s2 * (1.f - s2) * aw);
    gr[3] = -ci.d[3] * (8.f * s3 * s3 * (1.f - s3) * ah);
    gr[4] = 0.f;
  }
  // classes
  int cls = (int)tm[1];
  cls = min(max(cls, 0), p.nc - 1);
  float lsum = 0.f;
  for (int k = lane; k < p.nc; k += 64) {
    const float x = (float)px[5 + k];
    const float tt = (k == cls) ? 1.f : 0.f;
    lsum += bce_logits(x, tt);
    gr[5 + k] = sigmoid_ref(x) - tt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (lane == 0) cs[2] = p.nc > 1 ? lsum : 0.f;
}

// ---- deterministic single-block reductions --------------------------------------------------------------
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// (a device function since round 6: it runs as block 1 of yolo_obj_reduce_kernel, beside the objectness fold, instead of as a launch of
// its own between the candidate and the objectness passes — nothing before the finalize reads its three sums)
__device__ __forceinline__ void yolo_cand_reduce(const YoloLossParams& p, float* red) {
  float n = 0.f, lb = 0.f, lc = 0.f;
  for (int c0 = threadIdx.x; c0 < p.ncand; c0 += 8 * 1024) {  // eight candidates' loads in flight per thread, folded in candidate order
    int cl[8];
    float b8[8], c8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * 1024;
      cl[u] = c < p.ncand ? p.cell[c] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t c = cl[u] >= 0 ? c0 + u * 1024 : 0;
      b8[u] = p.cand[c * kCandStride + 1];
      c8[u] = p.cand[c * kCandStride + 2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (cl[u] >= 0) {
        n += 1.f;
        lb += b8[u];
        lc += c8[u];
      }
    }
  }
  n = block_sum_1024(n, red);
  lb = block_sum_1024(lb, red);
  lc = block_sum_1024(lc, red);
  if (threadIdx.x == 0) {
    p.sums[0] = n;
    p.sums[1] = lb;
    p.sums[2] = lc;
  }
}

// ---- backward of stage B as ONE streaming pass over the head gradient (round 6) --------------------------------------------------
// The gradient map is zero except the objectness channel of every anchor (dense) and the box / class channels of the matched cells
// (sparse, yolo_cand_bwd_kernel afterwards). Before: a zero-fill of the whole map (275 MB for the three YOLOv5-s levels at batch 64)
// followed by a pass that read the objectness logits and wrote 2-byte gradients at a 170-byte stride. Here a lane owns one 16-byte
// channel vector of one pixel and stores it once — zeros, or zeros with the objectness gradient of the anchor whose channel falls
// inside: the map is written exactly once, fully coalesced; same fp32 formula and rounding as yolo_obj_kernel<true>. The factor
// sigmoid(x) - t of every cell comes from the forward objectness pass (YoloLossParams::dobj): re-deriving it here put a three-deep chain
// of dependent gathers (logit, winner, candidate IoU) into 3 of every 32 lanes and held the whole pass at 1.9 TB/s.
// IT: index type of the flat walks below — unsigned 32-bit whenever the element count allows (the 64-bit divisions by run-time values
// cost ~100 instructions each; every launch of the benchmarked configurations takes the 32-bit instance)
template <typename IT>
__global__ __launch_bounds__(256) void yolo_obj_bwd_fill_kernel(const YoloLossParams p) {
  const IT VP = (IT)(p.ld >> 3);  // 16-byte vectors per pixel
  const IT HW = (IT)p.H * (IT)p.W;
  const IT nvec = (IT)p.N * HW * VP;
  const float go = (p.gout ? p.gout[0] : 1.f) * p.k_obj;
  const IT step = (IT)gridDim.x * 256;
  // the vector's position inside its pixel is thread-invariant when VP is a power of two dividing the grid stride (ld = 256: VP = 32):
  // which anchors' objectness channels fall into it is then decided once, and only those lanes (3 of 32) do any index arithmetic
  const bool inv = (VP & (VP - 1)) == 0 && (step & (VP - 1)) == 0;
  const int vp_shift = inv ? __ffs((int)VP) - 1 : 0;
  auto anchors_of = [&](int c0) {  // bit a set: channel a * NO + 4 lies in [c0, c0 + 8)
    unsigned m = 0;
    for (int a = 0; a < p.A; ++a) {
      const int oc = a * p.NO + 4;
      if (oc >= c0 && oc < c0 + 8) m |= 1u << a;
    }
    return m;
  };
  const unsigned my_mask = inv ? anchors_of((int)(((IT)blockIdx.x * 256 + threadIdx.x) & (VP - 1)) * 8) : 0u;
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < nvec; i += step) {
    IT pix;
    int c0;
    unsigned mask;
    if (inv) {
      pix = i >> vp_shift;
      c0 = (int)(i & (VP - 1)) * 8;
      mask = my_mask;
    } else {
      pix = i / VP;
      c0 = (int)(i - pix * VP) * 8;
      mask = anchors_of(c0);
    }
    h16_t out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (h16_t)0.f;
    if (mask) {
      const IT n = pix / HW;
      const IT cell0 = pix + n * (IT)(p.A - 1) * HW;  // cell (n, a = 0, gj, gi); anchor a adds a * HW
      for (int a = 0; a < p.A; ++a) {
        if (!((mask >> a) & 1u)) continue;
        const int oc = a * p.NO + 4;
        const h16_t gq = (h16_t)(p.dobj[(int64_t)cell0 + (int64_t)a * (int64_t)HW] * go);   // (sigmoid(x) - t): kept by the forward pass
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c0 + j == oc) out[j] = gq;
      }
    }
    *reinterpret_cast<uint4*>(p.draw + (int64_t)pix * p.ld + c0) = *reinterpret_cast<const uint4*>(out);
  }
}

// ---- stage B: objectness over every cell ------------------------------------------------------------------
template <bool BWD, typename IT>
__global__ __launch_bounds__(256) void yolo_obj_kernel(const YoloLossParams p) {
  __shared__ float red[256];
  const IT ncell = (IT)p.N * (IT)p.A * (IT)p.H * (IT)p.W;
  const float go = BWD ? (p.gout ? p.gout[0] : 1.f) * p.k_obj : 0.f;
  float acc = 0.f;
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < ncell; i += (IT)gridDim.x * 256) {
    // cell order (n, a, gj, gi); memory order pixel-major: walk pixels so neighbouring lanes touch neighbouring rows
    IT r = i / (IT)p.W;
    const int gi = (int)(i - r * (IT)p.W);
    const IT r2 = r / (IT)p.H;
    const int gj = (int)(r - r2 * (IT)p.H);
    const int n = (int)(r2 / (IT)p.A);
    const int a = (int)(r2 - (IT)n * (IT)p.A);
    const int64_t o = ((int64_t)(n * p.H + gj) * p.W + gi) * p.ld + a * p.NO + 4;
    const float x = (float)p.raw[o];
    const int w = p.winner[i];
    const float tt = w > 0 ? fmaxf(p.cand[(int64_t)(w - 1) * kCandStride], 0.f) : 0.f;
    if (BWD) p.draw[o] = (h16_t)((sigmoid_ref(x) - tt) * go);
    else {
      acc += bce_logits(x, tt);
      p.dobj[i] = sigmoid_ref(x) - tt;
    }
  }
  if (!BWD) {
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) p.partial[blockIdx.x] = red[0];
  }
}

__global__ __launch_bounds__(1024) void yolo_obj_reduce_kernel(const YoloLossParams p, int nblocks) {
  __shared__ float red[1024];
  if (blockIdx.x == 1) {
    yolo_cand_reduce(p, red);
    return;
  }
  float v = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 1024) v += p.partial[i];
  v = block_sum_1024(v, red);
  if (threadIdx.x == 0) p.sums[3] = v;
}

// ---- stage C (backward): per cell, the winner candidate sums the box/class gradients of all its duplicates -------
__global__ __launch_bounds__(256) void yolo_cand_bwd_kernel(const YoloLossParams p) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.ncand) return;
  const int cell = p.cell[c];
  if (cell < 0 || p.winner[cell] != c + 1) return;
  const float n = p.sums[0];
  const float g = p.gout ? p.gout[0] : 1.f;
  const float kb = g * p.k_box / n, kc = g * p.k_cls / n;
  // walk the cell's list; lane i keeps the i-th entry (first 64), the rest are folded in list order
  int mine = 0x7fffffff;
  int cur = p.head[cell];
  int cnt = 0;
  float acc0 = 0.f, acc1 = 0.f;  // channels lane, lane + 64
  const int ch0 = lane, ch1 = lane + 64;
  while (cur != 0) {
    const int e = cur - 1;
    if (cnt < 64) {
      if (lane == cnt) mine = e;
    } else {
      const float* gr = p.cgrad + (int64_t)e * p.gstride;
      if (ch0 < p.NO) acc0 += gr[ch0];
      if (ch1 < p.NO) acc1 += gr[ch1];
    }
    ++cnt;
    cur = p.next[e];
  }
  const int k = cnt < 64 ? cnt : 64;
  int last = -1;
  for (int i = 0; i < k; ++i) {  // ascending ordinal: deterministic summation order
    int cand_v = mine > last ? mine : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand_v = min(cand_v, __shfl_xor(cand_v, o, 64));
    last = cand_v;
    const float* gr = p.cgrad + (int64_t)cand_v * p.gstride;
    if (ch0 < p.NO) acc0 += gr[ch0];
    if (ch1 < p.NO) acc1 += gr[ch1];
  }
  // cell -> pixel
  const int gi = cell % p.W;
  int r = cell / p.W;
  const int gj = r % p.H;
  r /= p.H;
  const int a = r % p.A;
  const int b = r / p.A;
  h16_t* dst = p.draw + ((int64_t)(b * p.H + gj) * p.W + gi) * p.ld + a * p.NO;
  if (ch0 < p.NO && ch0 != 4) dst[ch0] = (h16_t)(acc0 * (ch0 < 4 ? kb : kc));
  if (ch1 < p.NO) dst[ch1] = (h16_t)(acc1 * kc);
}

// ---- stage D (backward, round 6): column sums of the head gradient from the loss's own compact state -------------------------------
// The detect convolutions carry a bias (yolov5_head.py: nn.Conv2d(ch, na * no, 1)): its gradient is the column sum of the gradient
// map, which the engine used to compute by reading the whole map again (275 MB per YOLOv5-s step, 3 x 52 us) — although the map is
// zero except one objectness channel per anchor and the box / class channels of the matched cells. Here every block sums its slice of
// the candidates' unscaled gradients (cgrad, the rows yolo_cand_bwd_kernel folds into the map) and of the per-cell objectness factors
// (dobj) and writes ONE partial row; cvhip_colsum_finalize folds the CVHIP_YOLO_BIAS_ROWS rows. Fixed partition, fixed orders:
// deterministic. The sums are those of the fp32 values BEFORE their rounding into the 16-bit map.
constexpr int kBiasRows = CVHIP_YOLO_BIAS_ROWS;
template <typename IT>
__global__ __launch_bounds__(256) void yolo_bias_partial_kernel(const YoloLossParams p, float* __restrict__ part) {
  __shared__ float wacc[4][kLossMaxA * 128];
  __shared__ float ored[kLossMaxA][256];
  const int K = p.A * p.NO;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 4 * kLossMaxA * 128; i += 256) (&wacc[0][0])[i] = 0.f;
  __syncthreads();
  // candidates: a contiguous slice per block, groups of 64 dealt to the waves in order; a lane owns channels lane, lane + 64
  const int per_blk = (p.ncand + gridDim.x - 1) / gridDim.x;
  const int c_begin = blockIdx.x * per_blk, c_end = min(c_begin + per_blk, p.ncand);
  for (int g0 = c_begin + wave * 64; g0 < c_end; g0 += 256) {
    const int c = g0 + lane;
    unsigned long long m = __ballot(c < c_end && p.cell[c] >= 0);
    while (m) {  // eight candidates' rows in flight per wave, folded in candidate order
      int cc[8];
      float v0[8], v1[8];
      int nv = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cc[u] = -1;
        if (m) {
          cc[u] = g0 + __ffsll((long long)m) - 1;
          m &= m - 1;
          nv = u + 1;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* gr = p.cgrad + (int64_t)(cc[u] >= 0 ? cc[u] : 0) * p.gstride;
        v0[u] = (cc[u] >= 0 && lane < p.NO && lane != 4) ? gr[lane] : 0.f;
        v1[u] = (cc[u] >= 0 && lane + 64 < p.NO) ? gr[lane + 64] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < nv) {
          const int a = (cc[u] / p.T) % p.A;
          if (lane < p.NO && lane != 4) wacc[wave][a * p.NO + lane] += v0[u];
          if (lane + 64 < p.NO) wacc[wave][a * p.NO + lane + 64] += v1[u];
        }
      }
    }
  }
  // objectness: a contiguous slice of the cells (n, a, gj, gi) per block
  float oa[kLossMaxA];
#pragma unroll
  for (int k = 0; k < kLossMaxA; ++k) oa[k] = 0.f;
  {
    const IT ncell = (IT)p.N * (IT)p.A * (IT)p.H * (IT)p.W, HW = (IT)p.H * (IT)p.W;
    const IT per = (ncell + gridDim.x - 1) / gridDim.x;
    const IT i_begin = (IT)blockIdx.x * per;
    IT i_end = i_begin + per;
    if (i_end > ncell) i_end = ncell;
    for (IT i0 = i_begin + t; i0 < i_end; i0 += 256 * 8) {  // 8 independent loads in flight per lane (64 blocks walk 1.2 M cells)
      float d[8];
      int a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const IT i = i0 + (IT)u * 256;
        const bool ok = i < i_end;
        d[u] = ok ? p.dobj[ok ? i : i_begin] : 0.f;
        a[u] = (int)(((ok ? i : i_begin) / HW) % (IT)p.A);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < kLossMaxA; ++k) oa[k] += k == a[u] ? d[u] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < kLossMaxA; ++k) ored[k][t] = oa[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {  // fixed tree: the same sums on every run
    if (t < st) {
#pragma unroll
      for (int k = 0; k < kLossMaxA; ++k) ored[k][t] += ored[k][t + st];
    }
    __syncthreads();
  }
  const float n = p.sums[0];
  const float g = p.gout ? p.gout[0] : 1.f;
  const float kb = n > 0.f ? g * p.k_box / n : 0.f, kc = n > 0.f ? g * p.k_cls / n : 0.f, go = g * p.k_obj;
  float* const row = part + (int64_t)blockIdx.x * 2 * K;
  for (int ch = t; ch < K; ch += 256) {
    const int a = ch / p.NO, o = ch - a * p.NO;
    const float v = ((wacc[0][ch] + wacc[1][ch]) + wacc[2][ch]) + wacc[3][ch];
    row[ch] = o == 4 ? ored[a][0] * go : v * (o < 4 ? kb : kc);
  }
}

// ---- final scalars ----------------------------------------------------------------------------------------
__global__ void yolo_finalize_kernel(const float* sums, int L, const float* ncell, const float* balance, float hyp_box,
                                     float hyp_obj, float hyp_cls, int nc, float bs, float* total, float* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lbox = 0.f, lobj = 0.f, lcls = 0.f;
  for (int i = 0; i < L; ++i) {
    const float n = sums[i * 4];
    if (n > 0.f) {
      lbox += sums[i * 4 + 1] / n;
      lcls += sums[i * 4 + 2] / (n * (float)nc);
    }
    lobj += sums[i * 4 + 3] / ncell[i] * balance[i];
  }
  lbox *= hyp_box;
  lobj *= hyp_obj;
  lcls *= hyp_cls;
  total[0] = (lbox + lobj + lcls) * bs;
  stats[0] = lbox;
  stats[1] = lobj;
  stats[2] = lcls;
}

static int fill(YoloLossParams& p, const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, float* sums) {
  if (!d || !raw || !targets || !ws || !sums) return CVHIP_ERR_INVALID;
  if (d->N <= 0 || d->A <= 0 || d->A > kLossMaxA || d->NO < 6 || d->NO > 128 || d->H <= 0 || d->W <= 0 || d->T <= 0 || d->ld < d->A * d->NO)
    return CVHIP_ERR_INVALID;
  const int64_t ncell = (int64_t)d->N * d->A * d->H * d->W;
  const int64_t ncand = (int64_t)5 * d->A * d->T;
  if (ncell >= (1ll << 31) || ncand >= (1ll << 30)) return CVHIP_ERR_UNSUPPORTED;
  p.raw = (const h16_t*)raw;
  p.draw = nullptr;
  p.targets = targets;
  p.assign = nullptr;
  p.ld = d->ld;
  p.N = d->N;
  p.A = d->A;
  p.NO = d->NO;
  p.H = d->H;
  p.W = d->W;
  p.T = d->T;
  p.nc = d->NO - 5;
  p.ncand = (int)ncand;
  p.anchor_t = d->anchor_t;
  for (int i = 0; i < d->A * 2; ++i) p.anchors[i] = d->anchors[i];
  p.gstride = (d->NO + 3) / 4 * 4;
  unsigned char* w = (unsigned char*)ws;
  auto take = [&](int64_t bytes) {
    unsigned char* r = w;
    w += (bytes + 255) / 256 * 256;
    return r;
  };
  p.winner = (int*)take(ncell * 4);
  p.head = (int*)take(ncell * 4);
  p.next = (int*)take(ncand * 4);
  p.cell = (int*)take(ncand * 4);
  p.cand = (float*)take(ncand * kCandStride * 4);
  p.cgrad = (float*)take(ncand * p.gstride * 4);
  p.partial = (float*)take(1024 * 4);
  p.dobj = (float*)take(ncell * 4);
  p.sums = sums;
  p.gout = nullptr;
  p.k_box = p.k_cls = p.k_obj = 0.f;
  return CVHIP_OK;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int64_t cvhip_yolov5_loss_workspace_bytes(const cvhip_yolo_loss_desc* d) {
  if (!d || d->N <= 0 || d->A <= 0 || d->NO <= 0 || d->H <= 0 || d->W <= 0 || d->T <= 0) return CVHIP_ERR_INVALID;
  const int64_t ncell = (int64_t)d->N * d->A * d->H * d->W;
  const int64_t ncand = (int64_t)5 * d->A * d->T;
  const int64_t gs = (d->NO + 3) / 4 * 4;
  auto r = [](int64_t b) { return (b + 255) / 256 * 256; };
  return r(ncell * 4) * 2 + r(ncand * 4) * 2 + r(ncand * kCandStride * 4) + r(ncand * gs * 4) + r(1024 * 4) + r(ncell * 4);
}

static int level_fwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, const int32_t* assign, void* ws, float* sums4,
                     void* stream);

int cvhip_yolov5_loss_level_fwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, float* sums4,
                                void* stream) {
  return level_fwd(d, raw, targets, nullptr, ws, sums4, stream);
}

int cvhip_yolov5_loss_level_fwd_assigned(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, const int32_t* assign,
                                         void* ws, float* sums4, void* stream) {
  if (!assign) return CVHIP_ERR_INVALID;
  return level_fwd(d, raw, targets, assign, ws, sums4, stream);
}

static int level_fwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, const int32_t* assign, void* ws, float* sums4,
                     void* stream) {
  YoloLossParams p;
  int rc = fill(p, d, raw, targets, ws, sums4);
  if (rc != CVHIP_OK) return rc;
  p.assign = assign;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ncell = (int64_t)p.N * p.A * p.H * p.W;
  rc = zero_fill(p.winner, (ncell * 4 + 255) / 256 * 256 * 2, st);  // winner + head are adjacent
  if (rc != CVHIP_OK) return rc;
  hipLaunchKernelGGL(yolo_cand_kernel, dim3(cdiv(p.ncand, 4)), dim3(256), 0, st, p);
  const int nb = (int)(cdiv64(ncell, 256) < 1024 ? cdiv64(ncell, 256) : 1024);
  if (ncell + (int64_t)nb * 256 < (1ll << 32)) hipLaunchKernelGGL((yolo_obj_kernel<false, unsigned>), dim3(nb), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((yolo_obj_kernel<false, int64_t>), dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(yolo_obj_reduce_kernel, dim3(2), dim3(1024), 0, st, p, nb);  // block 0: objectness partials, block 1: candidate sums
  return check_launch("yolov5_loss_level_fwd");
}

int cvhip_yolov5_loss_finalize(const float* sums, int32_t levels, const float* ncell, const float* balance, float hyp_box,
                               float hyp_obj, float hyp_cls, int32_t nc, float batch, float* total, float* stats3, void* stream) {
  if (!sums || !ncell || !balance || !total || !stats3 || levels <= 0) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(yolo_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, levels, ncell, balance, hyp_box, hyp_obj,
                     hyp_cls, nc, batch, total, stats3);
  return check_launch("yolov5_loss_finalize");
}

static int level_bwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, const float* sums4, const float* gout,
                     float k_box, float k_cls, float k_obj, void* draw, float* bias_partial, void* stream);

int cvhip_yolov5_loss_level_bwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, const float* sums4,
                                const float* gout, float k_box, float k_cls, float k_obj, void* draw, void* stream) {
  return level_bwd(d, raw, targets, ws, sums4, gout, k_box, k_cls, k_obj, draw, nullptr, stream);
}

int cvhip_yolov5_loss_level_bwd_bias(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, const float* sums4,
                                     const float* gout, float k_box, float k_cls, float k_obj, void* draw, float* bias_partial,
                                     void* stream) {
  if (!bias_partial) return CVHIP_ERR_INVALID;
  return level_bwd(d, raw, targets, ws, sums4, gout, k_box, k_cls, k_obj, draw, bias_partial, stream);
}

static int level_bwd(const cvhip_yolo_loss_desc* d, const void* raw, const float* targets, void* ws, const float* sums4, const float* gout,
                     float k_box, float k_cls, float k_obj, void* draw, float* bias_partial, void* stream) {
  YoloLossParams p;
  int rc = fill(p, d, raw, targets, ws, const_cast<float*>(sums4));
  if (rc != CVHIP_OK) return rc;
  if (!draw) return CVHIP_ERR_INVALID;
  p.draw = (h16_t*)draw;
  p.gout = gout;
  p.k_box = k_box;
  p.k_cls = k_cls;
  p.k_obj = k_obj;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ncell = (int64_t)p.N * p.A * p.H * p.W;
  if ((p.ld & 7) == 0 && (((uintptr_t)p.draw) & 15) == 0) {
    const int64_t nvec = (int64_t)p.N * p.H * p.W * (p.ld >> 3);
    const int nbf = (int)(cdiv64(nvec, 256 * 4) < 8192 ? cdiv64(nvec, 256 * 4) : 8192);
    if (nvec + (int64_t)nbf * 256 < (1ll << 32)) hipLaunchKernelGGL(yolo_obj_bwd_fill_kernel<unsigned>, dim3(nbf > 0 ? nbf : 1), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(yolo_obj_bwd_fill_kernel<int64_t>, dim3(nbf > 0 ? nbf : 1), dim3(256), 0, st, p);
  } else {
    rc = zero_fill(p.draw, (int64_t)p.N * p.H * p.W * p.ld * 2, st);
    if (rc != CVHIP_OK) return rc;
    const int nb = (int)(cdiv64(ncell, 256) < 8192 ? cdiv64(ncell, 256) : 8192);
    if (ncell + (int64_t)nb * 256 < (1ll << 32)) hipLaunchKernelGGL((yolo_obj_kernel<true, unsigned>), dim3(nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((yolo_obj_kernel<true, int64_t>), dim3(nb), dim3(256), 0, st, p);
  }
  hipLaunchKernelGGL(yolo_cand_bwd_kernel, dim3(cdiv(p.ncand, 4)), dim3(256), 0, st, p);
  if (bias_partial) {
    if (ncell < (1ll << 31)) hipLaunchKernelGGL(yolo_bias_partial_kernel<unsigned>, dim3(kBiasRows), dim3(256), 0, st, p, bias_partial);
    else hipLaunchKernelGGL(yolo_bias_partial_kernel<int64_t>, dim3(kBiasRows), dim3(256), 0, st, p, bias_partial);
  }
  return check_launch("yolov5_loss_level_bwd");
}

}  // extern "C"
