// weights_optim.hip — weight operand packing (fp32 KRSC master -> bf16 fprop / dgrad images) and the
// fused SGD-nesterov + EMA update over a flat fp32 parameter arena.
//
// Reference: autocast's per-forward weight cast (trainer.py:179-184), torch.optim.SGD
// (src/optimizers/__init__.py:60-68) and ModelEMA.update (src/utils/ema.py:30-39).
#include <string.h>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

// w_fprop[k][r][s][c] = bf16(master[k][r][s][c]) with zero rows/columns beyond the master's real (Kv, Cv) extent.
// Fast path (no padding): same element order, 8 elements per thread.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, h16_t* __restrict__ dst, int64_t n) {
  const int64_t nv = n >> 3;
  const bool al = (((uintptr_t)src) & 15) == 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    float4 a, b;
    if (al) {
      a = reinterpret_cast<const float4*>(src)[2 * i];
      b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    } else {
      const float* q = src + 8 * i;
      a = make_float4(q[0], q[1], q[2], q[3]);
      b = make_float4(q[4], q[5], q[6], q[7]);
    }
    uint4 u;
    u.x = pack2(a.x, a.y);
    u.y = pack2(a.z, a.w);
    u.z = pack2(b.x, b.y);
    u.w = pack2(b.z, b.w);
    reinterpret_cast<uint4*>(dst)[i] = u;
  }
  for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (h16_t)src[i];
}

__global__ __launch_bounds__(256) void pack_fprop_padded_kernel(const float* __restrict__ src, h16_t* __restrict__ dst, int K, int T, int C,
                                                                int Kv, int Cv) {
  const int64_t n = (int64_t)K * T * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t kt = i / C;
    const int t = (int)(kt % T);
    const int k = (int)(kt / T);
    dst[i] = (k < Kv && c < Cv) ? (h16_t)src[((int64_t)k * T + t) * Cv + c] : (h16_t)0.f;
  }
}

struct DgradPack {
  int ncls;
  int K, R, S, C, Kv, Cv;
  IgemmClass cls[kMaxClasses];
};

// dgrad image of class q: [c][i*TS + j][k] = master[k][r0+i*r_step][s0+j*s_step][c]
// one thread per output element, k fastest (coalesced writes; reads strided by R*S*C — the weight
// tensors are small and L2-resident).
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ master, h16_t* __restrict__ dst, const DgradPack p) {
  const int q = blockIdx.y;
  const IgemmClass& cl = p.cls[q];
  const int T = cl.TR * cl.TS;
  const int64_t total = (int64_t)p.C * T * p.K;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int k = (int)(idx % p.K);
    const int64_t ct = idx / p.K;
    const int tap = (int)(ct % T);
    const int c = (int)(ct / T);
    const int i = tap / cl.TS, j = tap - i * cl.TS;
    const int r = cl.r0 + i * cl.r_step, s = cl.s0 + j * cl.s_step;
    dst[cl.w_off + idx] = (k < p.Kv && c < p.Cv) ? (h16_t)master[(((int64_t)k * p.R + r) * p.S + s) * p.Cv + c] : (h16_t)0.f;
  }
}

// ---- band image (conv_plan.h): the fragment-ordered copy of a stride-1 3x3 layer's weights, straight from the fp32 master ------
// One thread per 16-byte vector of the copy. fprop (rows = K, reduction = C): eight consecutive floats of master[n][tap][c0..];
// dgrad (rows = C, reduction = K): master[k0 .. k0 + 7][r][s][n], a gather with a stride of R*S*C floats (weights are L2-resident).
__device__ __forceinline__ void band_image_store(const float* __restrict__ master, h16_t* __restrict__ dst, int64_t v, int K, int C, bool dgrad,
                                                 int r0, int r_step, int s0, int s_step) {
  int n, tap, c0;
  float f[8];
  if (!dgrad) {
    band_image_decode(v, K, C, &n, &tap, &c0);
    const float4* q = reinterpret_cast<const float4*>(master + ((int64_t)n * 9 + tap) * C + c0);
    const float4 a = q[0], b = q[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    band_image_decode(v, C, K, &n, &tap, &c0);
    const int i = tap / 3, j = tap - i * 3;
    const int r = r0 + i * r_step, s2 = s0 + j * s_step;
    const float* q = master + ((int64_t)c0 * 9 + r * 3 + s2) * C + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = q[(int64_t)e * 9 * C];
  }
  uint4 u;
  u.x = pack2(f[0], f[1]);
  u.y = pack2(f[2], f[3]);
  u.z = pack2(f[4], f[5]);
  u.w = pack2(f[6], f[7]);
  reinterpret_cast<uint4*>(dst)[v] = u;
}

__global__ __launch_bounds__(256) void band_image_kernel(const float* __restrict__ master, h16_t* __restrict__ dst, int K, int C, int dgrad, int r0,
                                                         int r_step, int s0, int s_step) {
  const int64_t nv = (int64_t)K * 9 * C / 8;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256)
    band_image_store(master, dst, v, K, C, dgrad != 0, r0, r_step, s0, s_step);
}

// ---- batched operand preparation ------------------------------------------------------------------------
// One launch packs the bf16 fprop AND dgrad images of every conv layer of a model (a training step otherwise pays one cast
// and one pack launch per layer: ~115 sub-5-us kernels on the step's critical path for YOLOv5-s). The table lives in device
// memory (built once by cvhip_prep_plan_build from the layers' descriptors and fixed operand addresses); block -> item by
// binary search over the items' first-block indices.
struct PrepClass {
  int TR, TS, r0, r_step, s0, s_step;
  int tap_begin, pad_;  // first tap (in class order) of this class
  int64_t w_off, w_end;
};
struct PrepItem {
  const float* master;
  h16_t* wf;
  h16_t* wd;
  int K, R, S, C, Kv, Cv;
  int ncls, blk_begin, nblk_f, nblk_d;
  int nblk_bf, nblk_bd;  // band images (conv_plan.h) behind the fprop / dgrad image: 256 16-byte vectors per block
  int ktiles, ctiles;  // dgrad image: one block per (tap, 64 x 64 tile of the [K][C] slice of that tap)
  PrepClass cls[kMaxClasses];
};
constexpr int kPrepFpropPerBlock = 2048, kPrepTile = 64;

__global__ __launch_bounds__(256) void prep_all_kernel(const PrepItem* __restrict__ items, int n_items) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= items[mid].blk_begin) lo = mid;
    else hi = mid - 1;
  }
  const PrepItem& it = items[lo];
  int b = blockIdx.x - it.blk_begin;
  const int K = it.K, C = it.C, T = it.R * it.S, Kv = it.Kv, Cv = it.Cv;
  const int64_t n = (int64_t)K * T * C;
  const float* __restrict__ src = it.master;
  if (b < it.nblk_f) {
    h16_t* __restrict__ dst = it.wf;
    const int64_t e0 = (int64_t)b * kPrepFpropPerBlock + threadIdx.x * 8;
    if (e0 >= n) return;
    if (Kv == K && Cv == C && e0 + 8 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
      const float4 a = reinterpret_cast<const float4*>(src + e0)[0];
      const float4 c = reinterpret_cast<const float4*>(src + e0)[1];
      uint4 u;
      u.x = pack2(a.x, a.y);
      u.y = pack2(a.z, a.w);
      u.z = pack2(c.x, c.y);
      u.w = pack2(c.z, c.w);
      *reinterpret_cast<uint4*>(dst + e0) = u;
    } else {
      for (int64_t i = e0; i < e0 + 8 && i < n; ++i) {
        const int c = (int)(i % C);
        const int64_t kt = i / C;
        const int t = (int)(kt % T);
        const int k = (int)(kt / T);
        dst[i] = (k < Kv && c < Cv) ? (h16_t)src[((int64_t)k * T + t) * Cv + c] : (h16_t)0.f;
      }
    }
    return;
  }
  // dgrad image: [c][tap][k] per class = the transpose of the tap's [K][C] slice of the master. 64 x 64 tiles through LDS:
  // reads coalesced along c, writes coalesced along k (the one-element-per-thread version read with a stride of R*S*C floats:
  // 0.36 ms per step for the 26 M parameters of DeepLabv3+)
  b -= it.nblk_f;
  if (b >= it.nblk_d) {  // band images, fprop's first
    b -= it.nblk_d;
    const bool dg = b >= it.nblk_bf;
    if (dg) b -= it.nblk_bf;
    const int64_t v = (int64_t)b * 256 + threadIdx.x;
    if (v < n / 8) band_image_store(src, (dg ? it.wd : it.wf) + n, v, K, C, dg, it.cls[0].r0, it.cls[0].r_step, it.cls[0].s0, it.cls[0].s_step);
    return;
  }
  __shared__ float tile[kPrepTile][kPrepTile + 1];
  const int per_tap = it.ktiles * it.ctiles;
  const int tapidx = b / per_tap;
  const int rem = b - tapidx * per_tap;
  const int kt = rem / it.ctiles, ct = rem - kt * it.ctiles;
  int q = 0;
  while (q + 1 < it.ncls && tapidx >= it.cls[q + 1].tap_begin) ++q;
  const PrepClass& cl = it.cls[q];
  const int tap = tapidx - cl.tap_begin;
  const int Tq = cl.TR * cl.TS;
  const int i = tap / cl.TS, jj = tap - i * cl.TS;
  const int r = cl.r0 + i * cl.r_step, s2 = cl.s0 + jj * cl.s_step;
  const int k0 = kt * kPrepTile, c0 = ct * kPrepTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int y = ty; y < kPrepTile; y += 4) {
    const int k = k0 + y, c = c0 + tx;
    tile[y][tx] = (k < Kv && c < Cv) ? src[(((int64_t)k * it.R + r) * it.S + s2) * Cv + c] : 0.f;
  }
  __syncthreads();
  h16_t* __restrict__ dst = it.wd + cl.w_off;
#pragma unroll
  for (int y = ty; y < kPrepTile; y += 4) {
    const int c = c0 + y, k = k0 + tx;
    if (c < C && k < K) dst[((int64_t)c * Tq + tap) * K + k] = (h16_t)tile[tx][y];
  }
}

// ---- fused optimizer ---------------------------------------------------------------------------------
// torch.optim.SGD step for element i in segment g:
//   d = grad*grad_scale + wd_g * p
//   buf = first_step ? d : momentum*buf + d
//   d = nesterov ? d + momentum*buf : buf
//   p -= lr_g * d
//   ema = decay*ema + (1-decay)*p        (if ema != NULL)
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                      float* __restrict__ mom, float* __restrict__ ema, int64_t n,
                                                      const int64_t* __restrict__ seg, const float* __restrict__ seg_lr,
                                                      const float* __restrict__ seg_wd, int nseg, float momentum,
                                                      int nesterov, int first, float decay, float gscale,
                                                      const float* __restrict__ dyn, const float* __restrict__ scaler) {
  float lr_scale = 1.f;
  if (dyn) {  // {ema_decay, lr_scale} read from device memory: values can change between hipGraph replays
    decay = dyn[0];
    lr_scale = dyn[1];
  }
  if (scaler) {  // dynamic loss scaling: {1/scale, skip} written by loss_scale_update_kernel for THIS step
    gscale *= scaler[0];
    if (scaler[1] != 0.f) {
      // non-finite gradients: GradScaler.step() skips optimizer.step() — parameters and momentum stay; ModelEMA.update still
      // runs in the reference (trainer.py:206), on the unchanged parameters
      if (ema)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
          ema[i] = decay * ema[i] + (1.f - decay) * param[i];
      return;
    }
  }
  auto seg_of = [&](int64_t i) {  // binary search the segment containing i (segments are sorted, disjoint, cover [0,n))
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (i >= seg[2 * mid + 1]) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  auto upd = [&](float& pv, float g, float& mv, float& ev, float lr, float wd) {
    float d = g * gscale + wd * pv;
    const float b = first ? d : momentum * mv + d;
    mv = b;
    d = nesterov ? d + momentum * b : b;
    pv -= lr * d;
    if (ema) ev = decay * ev + (1.f - decay) * pv;
  };
  // 16 bytes per lane per array (the arenas are 16-byte aligned; the element-at-a-time walk ran at 2.1 TB/s on DeepLabv3+'s 44 M
  // parameters: 414 us per step); a vector that straddles a segment boundary looks its elements up one by one
  const bool v4 = ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)mom) | ((uintptr_t)ema)) & 15) == 0;
  const int64_t n4 = v4 ? (n >> 2) : 0;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q << 2;
    float4 pv = *reinterpret_cast<const float4*>(param + i);
    const float4 gv = *reinterpret_cast<const float4*>(grad + i);
    float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(mom + i);
    float4 ev = ema ? *reinterpret_cast<const float4*>(ema + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = seg_of(i);
    const bool one = i + 3 < seg[2 * s0 + 1];
    const int s1 = one ? s0 : seg_of(i + 1), s2 = one ? s0 : seg_of(i + 2), s3 = one ? s0 : seg_of(i + 3);
    upd(pv.x, gv.x, mv.x, ev.x, seg_lr[s0] * lr_scale, seg_wd[s0]);
    upd(pv.y, gv.y, mv.y, ev.y, seg_lr[s1] * lr_scale, seg_wd[s1]);
    upd(pv.z, gv.z, mv.z, ev.z, seg_lr[s2] * lr_scale, seg_wd[s2]);
    upd(pv.w, gv.w, mv.w, ev.w, seg_lr[s3] * lr_scale, seg_wd[s3]);
    *reinterpret_cast<float4*>(param + i) = pv;
    *reinterpret_cast<float4*>(mom + i) = mv;
    if (ema) *reinterpret_cast<float4*>(ema + i) = ev;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int lo = seg_of(i);
    float pv = param[i], mv = first ? 0.f : mom[i], ev = ema ? ema[i] : 0.f;
    upd(pv, grad[i], mv, ev, seg_lr[lo] * lr_scale, seg_wd[lo]);
    param[i] = pv;
    mom[i] = mv;
    if (ema) ema[i] = ev;
  }
}

// fused AdamW + ModelEMA over the flat arenas (torch.optim.AdamW semantics, src/optimizers/__init__.py:71-73; config 1 trains with it:
// conf/mini-imagenet.yml:91-99). Decoupled weight decay, amsgrad off, per element:
//   p   *= 1 - lr_g * wd_g
//   m    = b1*m + (1-b1)*g            v = b2*v + (1-b2)*g*g
//   p   -= (lr_g / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   ema  = decay*ema + (1-decay)*p    (if ema != NULL)
// The step count lives in DEVICE memory (step_state[0] = t as float, incremented by adamw_tick_kernel once per step), so a captured
// step replays with the right bias corrections; {ema_decay, lr_scale} come from `dyn` like in the SGD kernel.
__global__ void adamw_tick_kernel(float* step_state, const float* scaler) {
  // (GradScaler.step() does not call optimizer.step() on non-finite gradients: the count stays)
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(scaler && scaler[1] != 0.f)) step_state[0] += 1.f;
}
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m1,
                                                        float* __restrict__ m2, float* __restrict__ ema, int64_t n,
                                                        const int64_t* __restrict__ seg, const float* __restrict__ seg_lr,
                                                        const float* __restrict__ seg_wd, int nseg, float beta1, float beta2, float eps,
                                                        const float* __restrict__ step_state, float decay, float gscale,
                                                        const float* __restrict__ dyn, const float* __restrict__ scaler) {
  float lr_scale = 1.f;
  if (dyn) {
    decay = dyn[0];
    lr_scale = dyn[1];
  }
  if (scaler) {
    gscale *= scaler[0];
    if (scaler[1] != 0.f) {  // non-finite gradients: the step is skipped, ModelEMA.update still runs (trainer.py:199-207)
      if (ema)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
          ema[i] = decay * ema[i] + (1.f - decay) * param[i];
      return;
    }
  }
  const float t = step_state[0];  // >= 1: the tick kernel ran before this one
  // torch computes the corrections in double on the host; fp32 pow of a float t is exact enough only for small t, so use exp2/log2
  // in fp64 here (one thread-uniform evaluation per thread: negligible against the memory pass)
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  const float inv_bc1 = (float)(1.0 / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  auto seg_of = [&](int64_t i) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (i >= seg[2 * mid + 1]) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  auto upd = [&](float& pv, float g, float& a, float& b, float& ev, float lr, float wd) {
    g *= gscale;
    pv *= 1.f - lr * wd;
    a = beta1 * a + (1.f - beta1) * g;
    b = beta2 * b + (1.f - beta2) * g * g;
    const float denom = sqrtf(b) * inv_sqrt_bc2 + eps;
    pv -= (lr * inv_bc1) * (a / denom);
    if (ema) ev = decay * ev + (1.f - decay) * pv;
  };
  const bool v4 = ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)m1) | ((uintptr_t)m2) | ((uintptr_t)ema)) & 15) == 0;
  const int64_t n4 = v4 ? (n >> 2) : 0;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q << 2;
    float4 pv = *reinterpret_cast<const float4*>(param + i);
    const float4 gv = *reinterpret_cast<const float4*>(grad + i);
    float4 av = *reinterpret_cast<const float4*>(m1 + i), bv = *reinterpret_cast<const float4*>(m2 + i);
    float4 ev = ema ? *reinterpret_cast<const float4*>(ema + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = seg_of(i);
    const bool one = i + 3 < seg[2 * s0 + 1];
    const int s1 = one ? s0 : seg_of(i + 1), s2 = one ? s0 : seg_of(i + 2), s3 = one ? s0 : seg_of(i + 3);
    upd(pv.x, gv.x, av.x, bv.x, ev.x, seg_lr[s0] * lr_scale, seg_wd[s0]);
    upd(pv.y, gv.y, av.y, bv.y, ev.y, seg_lr[s1] * lr_scale, seg_wd[s1]);
    upd(pv.z, gv.z, av.z, bv.z, ev.z, seg_lr[s2] * lr_scale, seg_wd[s2]);
    upd(pv.w, gv.w, av.w, bv.w, ev.w, seg_lr[s3] * lr_scale, seg_wd[s3]);
    *reinterpret_cast<float4*>(param + i) = pv;
    *reinterpret_cast<float4*>(m1 + i) = av;
    *reinterpret_cast<float4*>(m2 + i) = bv;
    if (ema) *reinterpret_cast<float4*>(ema + i) = ev;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int lo = seg_of(i);
    float pv = param[i], a = m1[i], b = m2[i], ev = ema ? ema[i] : 0.f;
    upd(pv, grad[i], a, b, ev, seg_lr[lo] * lr_scale, seg_wd[lo]);
    param[i] = pv;
    m1[i] = a;
    m2[i] = b;
    if (ema) ema[i] = ev;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ src, int64_t n, float decay,
                                                  const float* __restrict__ dyn) {
  if (dyn) decay = dyn[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    ema[i] = decay * ema[i] + (1.f - decay) * src[i];
}

// ---- dynamic loss scaling (fp16 storage): torch.cuda.amp.GradScaler's rule on device, no host round trip -------------------
// state = {scale, growth_tracker, found_inf, skipped_steps}
__global__ __launch_bounds__(256) void loss_scale_check_kernel(const float* __restrict__ grad, int64_t n, float* __restrict__ state) {
  bool bad = false;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(grad);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = g4[i];
    // x - x is 0 for finite x and NaN for +-inf / NaN
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad = bad || !(t == 0.f);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = grad[i];
    bad = bad || !((x - x) == 0.f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.f;  // benign race: every writer stores the same value
}

__global__ void loss_scale_update_kernel(float* __restrict__ state, float* __restrict__ scaler, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float scale = state[0];
  const bool inf = state[2] != 0.f;
  scaler[0] = inf ? 0.f : 1.f / scale;   // the gradients of THIS step were produced under `scale`
  scaler[1] = inf ? 1.f : 0.f;
  if (inf) {
    state[0] = fmaxf(scale * backoff, 1.f / 65536.f);
    state[1] = 0.f;
    state[3] += 1.f;
  } else {
    const float t = state[1] + 1.f;
    if (t >= (float)interval) {
      state[0] = fminf(scale * growth, 3.0e38f);
      state[1] = 0.f;
    } else {
      state[1] = t;
    }
  }
  state[2] = 0.f;
}

static inline int grid1d(int64_t n) {
  int64_t b = cdiv64(n, 256);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

int pack_weights(const cvhip_conv_desc* d, const float* master, void* w_fprop, void* w_dgrad, hipStream_t stream) {
  const int64_t n = (int64_t)d->K * d->R * d->S * d->C;
  const int Kv = d->k_valid > 0 ? d->k_valid : d->K, Cv = d->c_valid > 0 ? d->c_valid : d->C;
  if (w_fprop) {
    if (Kv == d->K && Cv == d->C) {
      hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid1d(n / 8 + 1)), dim3(256), 0, stream, master, (h16_t*)w_fprop, n);
    } else {
      hipLaunchKernelGGL(pack_fprop_padded_kernel, dim3(grid1d(n)), dim3(256), 0, stream, master, (h16_t*)w_fprop, d->K, d->R * d->S,
                         d->C, Kv, Cv);
    }
    int st = check_launch("pack_fprop_kernel");
    if (st) return st;
  }
  if (w_dgrad) {
    IgemmParams ip;
    const int ncls = plan_dgrad(d, &ip);
    if (ncls < 0) return ncls;
    DgradPack p;
    p.ncls = ncls;
    p.K = d->K;
    p.R = d->R;
    p.S = d->S;
    p.C = d->C;
    p.Kv = Kv;
    p.Cv = Cv;
    int64_t maxe = 1;
    for (int i = 0; i < ncls; ++i) {
      p.cls[i] = ip.cls[i];
      const int64_t e = (int64_t)d->C * ip.cls[i].TR * ip.cls[i].TS * d->K;
      if (e > maxe) maxe = e;
    }
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid1d(maxe), ncls), dim3(256), 0, stream, master, (h16_t*)w_dgrad, p);
    int st = check_launch("pack_dgrad_kernel");
    if (st) return st;
    if (band_image_dgrad(d)) {  // (stride 1: one class, its image is the whole K*R*S*C)
      hipLaunchKernelGGL(band_image_kernel, dim3(grid1d(n / 8)), dim3(256), 0, stream, master, (h16_t*)w_dgrad + n, d->K, d->C, 1, ip.cls[0].r0,
                         ip.cls[0].r_step, ip.cls[0].s0, ip.cls[0].s_step);
      st = check_launch("band_image_kernel(dgrad)");
      if (st) return st;
    }
  }
  if (w_fprop && band_image_fprop(d)) {
    hipLaunchKernelGGL(band_image_kernel, dim3(grid1d(n / 8)), dim3(256), 0, stream, master, (h16_t*)w_fprop + n, d->K, d->C, 0, 0, 1, 0, 1);
    return check_launch("band_image_kernel(fprop)");
  }
  return CVHIP_OK;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_sgd_nesterov_ema(float* param, const float* grad, float* momentum_buf, float* ema, int64_t n,
                           const int64_t* seg_bounds, const float* seg_lr, const float* seg_wd, int32_t nseg,
                           float momentum, int32_t nesterov, int32_t first_step, float ema_decay, float grad_scale,
                           const float* dyn_decay_lrscale, void* stream) {
  if (!param || !grad || !momentum_buf || n < 0 || !seg_bounds || !seg_lr || !seg_wd || nseg <= 0) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(sgd_ema_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, ema, n,
                     seg_bounds, seg_lr, seg_wd, nseg, momentum, nesterov, first_step, ema_decay, grad_scale, dyn_decay_lrscale,
                     (const float*)nullptr);
  return check_launch("sgd_ema_kernel");
}

int cvhip_sgd_nesterov_ema_scaled(float* param, const float* grad, float* momentum_buf, float* ema, int64_t n,
                                  const int64_t* seg_bounds, const float* seg_lr, const float* seg_wd, int32_t nseg,
                                  float momentum, int32_t nesterov, int32_t first_step, float ema_decay, float grad_scale,
                                  const float* dyn_decay_lrscale, const float* scaler2, void* stream) {
  if (!param || !grad || !momentum_buf || n < 0 || !seg_bounds || !seg_lr || !seg_wd || nseg <= 0 || !scaler2) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(sgd_ema_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, ema, n,
                     seg_bounds, seg_lr, seg_wd, nseg, momentum, nesterov, first_step, ema_decay, grad_scale, dyn_decay_lrscale,
                     scaler2);
  return check_launch("sgd_ema_kernel(scaled)");
}

int cvhip_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t n, const int64_t* seg_bounds,
                    const float* seg_lr, const float* seg_wd, int32_t nseg, float beta1, float beta2, float eps, float* step_state,
                    float ema_decay, float grad_scale, const float* dyn_decay_lrscale, const float* scaler2, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || !seg_bounds || !seg_lr || !seg_wd || nseg <= 0 || !step_state) return CVHIP_ERR_INVALID;
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f)) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_state, scaler2);
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, ema, n, seg_bounds,
                     seg_lr, seg_wd, nseg, beta1, beta2, eps, step_state, ema_decay, grad_scale, dyn_decay_lrscale, scaler2);
  return check_launch("adamw_ema_kernel");
}

int cvhip_loss_scale_check(const float* grad, int64_t n, float* state4, void* stream) {
  if (!grad || !state4 || n < 0 || (((uintptr_t)grad) & 15)) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(loss_scale_check_kernel, dim3(grid1d(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, grad, n, state4);
  return check_launch("loss_scale_check_kernel");
}

int cvhip_loss_scale_update(float* state4, float* scaler2, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream) {
  if (!state4 || !scaler2 || growth_factor < 1.f || backoff_factor <= 0.f || backoff_factor > 1.f || growth_interval < 1) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state4, scaler2, growth_factor, backoff_factor,
                     growth_interval);
  return check_launch("loss_scale_update_kernel");
}

int cvhip_prep_plan_item_bytes(void) { return (int)sizeof(PrepItem); }

int cvhip_prep_plan_build(const cvhip_prep_entry* entries, int32_t n, void* table_host, int32_t* total_blocks) {
  if (!entries || n <= 0 || !table_host || !total_blocks) return CVHIP_ERR_INVALID;
  PrepItem* out = reinterpret_cast<PrepItem*>(table_host);
  int blk = 0;
  for (int e = 0; e < n; ++e) {
    const cvhip_conv_desc* d = &entries[e].desc;
    int st = validate_dense_desc(d);
    if (st) return st;
    if (!entries[e].master || !entries[e].w_fprop) return CVHIP_ERR_INVALID;
    if ((((uintptr_t)entries[e].w_fprop) & 15) || (((uintptr_t)entries[e].w_dgrad) & 15)) return CVHIP_ERR_INVALID;
    PrepItem& it = out[e];
    memset(&it, 0, sizeof(it));
    it.master = entries[e].master;
    it.wf = (h16_t*)entries[e].w_fprop;
    it.wd = (h16_t*)entries[e].w_dgrad;
    it.K = d->K;
    it.R = d->R;
    it.S = d->S;
    it.C = d->C;
    it.Kv = d->k_valid > 0 ? d->k_valid : d->K;
    it.Cv = d->c_valid > 0 ? d->c_valid : d->C;
    const int64_t nel = (int64_t)d->K * d->R * d->S * d->C;
    it.blk_begin = blk;
    it.nblk_f = (int)cdiv64(nel, kPrepFpropPerBlock);
    it.nblk_d = 0;
    if (it.wd) {
      IgemmParams ip;
      const int ncls = plan_dgrad(d, &ip);
      if (ncls < 0) return ncls;
      it.ncls = ncls;
      int64_t end = 0;
      int tapn = 0;
      for (int q = 0; q < ncls; ++q) {
        const IgemmClass& c = ip.cls[q];
        PrepClass& pc = it.cls[q];
        pc.TR = c.TR; pc.TS = c.TS; pc.r0 = c.r0; pc.r_step = c.r_step; pc.s0 = c.s0; pc.s_step = c.s_step;
        pc.tap_begin = tapn;
        tapn += c.TR * c.TS;
        pc.w_off = c.w_off;
        pc.w_end = c.w_off + (int64_t)d->C * c.TR * c.TS * d->K;
        if (pc.w_off != end) return CVHIP_ERR_UNSUPPORTED;  // the class images must tile [0, K*R*S*C) in order
        end = pc.w_end;
      }
      if (end != nel || tapn != d->R * d->S) return CVHIP_ERR_UNSUPPORTED;
      it.ktiles = cdiv(d->K, kPrepTile);
      it.ctiles = cdiv(d->C, kPrepTile);
      it.nblk_d = tapn * it.ktiles * it.ctiles;
    }
    it.nblk_bf = band_image_fprop(d) ? (int)cdiv64(nel / 8, 256) : 0;
    it.nblk_bd = (it.wd && band_image_dgrad(d)) ? (int)cdiv64(nel / 8, 256) : 0;
    blk += it.nblk_f + it.nblk_d + it.nblk_bf + it.nblk_bd;
  }
  *total_blocks = blk;
  return CVHIP_OK;
}

int cvhip_prep_plan_run(const void* table_device, int32_t n, int32_t total_blocks, void* stream) {
  if (!table_device || n <= 0 || total_blocks <= 0) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(prep_all_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)table_device, n);
  return check_launch("prep_all_kernel");
}

extern "C++" {
namespace cvhip {
__global__ void i64_add_kernel(long long* v, int64_t n, long long delta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += delta;
}
}  // namespace cvhip
}

int cvhip_i64_add(int64_t* v, int64_t n, int64_t delta, void* stream) {
  if (!v || n < 0) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(i64_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long*)v, n, (long long)delta);
  return check_launch("i64_add_kernel");
}

int cvhip_ema_update(float* ema, const float* src, int64_t n, float decay, const float* dyn_decay, void* stream) {
  if (!ema || !src || n < 0) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipLaunchKernelGGL(ema_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, ema, src, n, decay, dyn_decay);
  return check_launch("ema_kernel");
}

}  // extern "C"
