// pool_resize.hip — pooling / resampling / layout glue on NHWC bf16 (HBM-bound streaming kernels,
// 16-B channel vectors per thread where C % 8 == 0, scalar fallback otherwise).
//
// Reference call sites: src/models/modules/yolo_modules.py:147,152 (nearest x2 + cat),
// :176-192 (SPPF max-pools), src/models/heads/seg/deeplabv3plus_head.py:56-66 and
// src/models/segmentors/encoder_decoder.py:99 (bilinear), src/models/modules/yolo_modules.py:30-36
// (Focus space-to-depth).
#include "common.h"
#include "bilinear_index.h"

namespace cvhip {

__device__ __forceinline__ bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// generic 8-wide load/store helpers with scalar tail
__device__ __forceinline__ f32x8 load8(const h16_t* p, int c, int C, bool vec) {
  if (vec) return unpack8(*reinterpret_cast<const uint4*>(p + c));
  f32x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = (c + j < C) ? (float)p[c + j] : 0.f;
  return r;
}
__device__ __forceinline__ void store8(h16_t* p, int c, int C, bool vec, const f32x8& v) {
  if (vec) {
    *reinterpret_cast<uint4*>(p + c) = pack8(v);
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c + j < C) p[c + j] = (h16_t)v.v[j];
}

// ---------------------------------------------------------------------------------------------------
// max pool
// ---------------------------------------------------------------------------------------------------
struct PoolParams {
  const h16_t* x;
  h16_t* y;
  uint8_t* idx;
  const h16_t* dy;
  const uint8_t* cidx;
  h16_t* dx;
  int ld_x, ld_y, ld_dy, ld_dx;
  int N, C, H, W, OH, OW, k, s, pad;
  int accumulate;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.C & 7) == 0 && (p.ld_x & 7) == 0 && (p.ld_y & 7) == 0 && aligned16(p.x) && aligned16(p.y);
  const int64_t total = (int64_t)p.N * p.OH * p.OW * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, ow, oh, n;
    split_index(i, CV, p.OW, p.OH, &cv, &ow, &oh, &n);
    const int c = cv * 8;
    const int h0 = oh * p.s - p.pad, w0 = ow * p.s - p.pad;
    float best[8];
    int bi[8];
    bool first = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 0;
    }
    for (int kh = 0; kh < p.k; ++kh) {
      const int ih = h0 + kh;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int kw = 0; kw < p.k; ++kw) {
        const int iw = w0 + kw;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        const f32x8 v = load8(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.ld_x, c, p.C, vec);
        const int off = kh * p.k + kw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // ATen rule: the first in-bounds tap seeds the index (value stays -inf); a later tap wins
          // only if strictly greater, or NaN.
          if (first) bi[j] = off;
          if (v.v[j] > best[j] || v.v[j] != v.v[j]) {
            best[j] = v.v[j];
            bi[j] = off;
          }
        }
        first = false;
      }
    }
    const int64_t opix = ((int64_t)(n * p.OH + oh) * p.OW + ow);
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = best[j];
    store8(p.y + opix * p.ld_y, c, p.C, vec, o);
    if (p.idx) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c + j < p.C) p.idx[opix * p.C + c + j] = (uint8_t)bi[j];
    }
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.C & 7) == 0 && (p.ld_dy & 7) == 0 && (p.ld_dx & 7) == 0 && aligned16(p.dy) && aligned16(p.dx);
  const int64_t total = (int64_t)p.N * p.H * p.W * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, iw, ih, n;
    split_index(i, CV, p.W, p.H, &cv, &iw, &ih, &n);
    const int c = cv * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // outputs whose window covers (ih, iw): oh*s - pad <= ih <= oh*s - pad + k - 1
    int oh_lo = ih + p.pad - p.k + 1;
    oh_lo = oh_lo <= 0 ? 0 : (oh_lo + p.s - 1) / p.s;
    int oh_hi = (ih + p.pad) / p.s;
    if (oh_hi > p.OH - 1) oh_hi = p.OH - 1;
    int ow_lo = iw + p.pad - p.k + 1;
    ow_lo = ow_lo <= 0 ? 0 : (ow_lo + p.s - 1) / p.s;
    int ow_hi = (iw + p.pad) / p.s;
    if (ow_hi > p.OW - 1) ow_hi = p.OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int kh = ih - (oh * p.s - p.pad);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int kw = iw - (ow * p.s - p.pad);
        const int off = kh * p.k + kw;
        const int64_t opix = ((int64_t)(n * p.OH + oh) * p.OW + ow);
        const f32x8 g = load8(p.dy + opix * p.ld_dy, c, p.C, vec);
        const uint8_t* ip = p.cidx + opix * p.C + c;
        if (vec) {  // the 8 arg-max bytes of this channel vector in ONE 8-byte load (they were 8 byte loads per window)
          const uint2 iv = *reinterpret_cast<const uint2*>(ip);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned b = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
            if ((int)b == off) acc[j] += g.v[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c + j < p.C && ip[j] == off) acc[j] += g.v[j];
        }
      }
    }
    h16_t* dst = p.dx + ((int64_t)(n * p.H + ih) * p.W + iw) * p.ld_dx;
    f32x8 o;
    if (p.accumulate) {
      const f32x8 old = load8(dst, c, p.C, vec);
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = old.v[j] + acc[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    }
    store8(dst, c, p.C, vec, o);
  }
}

// ---- stride-1 "same" max pool of small maps through the LDS (round 4: SPPF / SPP pools, 3x3 / 5x5 on maps of <= 1024 pixels) ----------
// The gather kernels above issue k*k 16-byte L2 loads (+ k*k index loads in backward) and k*k compare / select chains per output
// vector: 38.6 / 45.8 us for SPPF's 13 MB tensors. Here a block stages ONE image x 4 channel vectors (32 channels) of the map in the
// LDS with coalesced 64-byte row pieces. Forward is SEPARABLE with ATen's exact (value, arg-max) semantics — first in-bounds tap
// seeds the index, a later tap wins only if strictly greater or NaN: pass 1 gives every pixel its ROW result (value and kw of the
// row's first maximum, or last NaN), pass 2 combines the k row results top to bottom (an earlier row keeps ties): k + k comparisons
// per output instead of k * k, the same answer as the row-major scan (tests/test_gpu_kernels.py compares values AND arg-max bytes
// with max_pool2d_with_indices, ties / -inf / NaN included). Backward sums, per input pixel, the windows that selected it in the
// gather kernel's order (bit-identical sums) from LDS copies of dy and the arg-max bytes. 24.5 / 33.6 us in the YOLOv5-s step.
// Round 6 (second session): k = 9 and 13 as well (the SPP / SPPCSPC pools of YOLOX and YOLOv7, yolo_modules.py SPP / yolov7_modules.py
// SPPCSPC: the gather kernels spent 141 - 179 us per launch on their 81 / 169 taps), and 2 or 1 channel vectors per block where 4 do not
// fit the LDS (YOLOv7-l's 40 x 40 maps: 1600 pixels x 2 vectors x 40 B = 128 KB).
constexpr int kPoolLdsCvbMax = 4;     // channel vectors per block (32 channels: 512 blocks for SPPF's 64 x 256-channel maps)
constexpr int kPoolLdsMaxPix = 4096;  // H * W limit with ONE channel vector per block: 4096 x 40 B = 160 KB of LDS (forward: input + row values + row arg-max)
static inline int pool_lds_cvb(int HW, bool bwd) {   // the widest block that fits the CU's LDS, 0 = none
  for (int cvb = kPoolLdsCvbMax; cvb >= 1; cvb >>= 1)
    if (HW * cvb * (bwd ? 24 : 40) <= 160 * 1024) return cvb;
  return 0;
}

template <int K, int kPoolLdsCvb>
__global__ __launch_bounds__(256) void maxpool_s1_lds_fwd_kernel(const PoolParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pool_smem[];
  const int CV = p.C >> 3;
  const int groups = (CV + kPoolLdsCvb - 1) / kPoolLdsCvb;
  const int n = blockIdx.x / groups, cv0 = (blockIdx.x - n * groups) * kPoolLdsCvb;
  const int cvb = min(kPoolLdsCvb, CV - cv0);
  const int HW = p.H * p.W;
  const int items = HW * kPoolLdsCvb;
  uint4* const xs = reinterpret_cast<uint4*>(pool_smem);                       // [H*W][CVB] input vectors
  uint4* const rvs = xs + items;                                               // [H*W][CVB] ROW results: value of the row's first maximum
  uint2* const rks = reinterpret_cast<uint2*>(rvs + items);                    // ... and its kw, one byte per channel
  const h16_t* const xn = p.x + (int64_t)n * HW * p.ld_x + cv0 * 8;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int pix = i / kPoolLdsCvb, cv = i - pix * kPoolLdsCvb;
    if (cv < cvb) xs[i] = *reinterpret_cast<const uint4*>(xn + (int64_t)pix * p.ld_x + cv * 8);
  }
  __syncthreads();
  constexpr int PAD = K / 2;
  // pass 1: every (pixel, channel vector) scans its ROW window (the values are 16-bit inputs, -inf or NaN: pack8 keeps them exactly)
  for (int i = threadIdx.x; i < items; i += 256) {
    const int pix = i / kPoolLdsCvb, cv = i - pix * kPoolLdsCvb;
    if (cv >= cvb) continue;
    const int h = pix / p.W, w = pix - h * p.W;
    f32x8 ov;
    int ok[8];
    bool first = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ov.v[j] = -INFINITY;
      ok[j] = 0;
    }
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int iw = w + kw - PAD;
      if ((unsigned)iw >= (unsigned)p.W) continue;
      const f32x8 v = unpack8(xs[(h * p.W + iw) * kPoolLdsCvb + cv]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (first) ok[j] = kw;
        if (v.v[j] > ov.v[j] || v.v[j] != v.v[j]) {
          ov.v[j] = v.v[j];
          ok[j] = kw;
        }
      }
      first = false;
    }
    rvs[i] = pack8(ov);
    uint2 kk;
    kk.x = (unsigned)ok[0] | ((unsigned)ok[1] << 8) | ((unsigned)ok[2] << 16) | ((unsigned)ok[3] << 24);
    kk.y = (unsigned)ok[4] | ((unsigned)ok[5] << 8) | ((unsigned)ok[6] << 16) | ((unsigned)ok[7] << 24);
    rks[i] = kk;
  }
  __syncthreads();
  // pass 2: combine the row results top to bottom
  for (int i = threadIdx.x; i < items; i += 256) {
    const int pix = i / kPoolLdsCvb, cv = i - pix * kPoolLdsCvb;
    if (cv >= cvb) continue;
    const int h = pix / p.W, w = pix - h * p.W;
    float best[8];
    int bi[8];
    bool first = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 0;
    }
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int ih = h + kh - PAD;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      const int o = (ih * p.W + w) * kPoolLdsCvb + cv;
      const f32x8 v = unpack8(rvs[o]);
      const uint2 kk = rks[o];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int off = kh * K + (int)(((j < 4 ? kk.x : kk.y) >> (8 * (j & 3))) & 0xffu);
        if (first) bi[j] = off;
        if (v.v[j] > best[j] || v.v[j] != v.v[j]) {
          best[j] = v.v[j];
          bi[j] = off;
        }
      }
      first = false;
    }
    {
      const int64_t opix = (int64_t)n * HW + h * p.W + w;
      f32x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = best[j];
      *reinterpret_cast<uint4*>(p.y + opix * p.ld_y + (cv0 + cv) * 8) = pack8(o);
      if (p.idx) {
        uint2 iv;
        iv.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        iv.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2*>(p.idx + opix * p.C + (cv0 + cv) * 8) = iv;
      }
    }
  }
}

// backward twin: dy and the arg-max bytes of one image x 4 channel vectors in the LDS, a thread per input pixel x channel vector sums the
// windows that selected it (k * k LDS reads instead of k * k L2 gathers)
template <int K, int kPoolLdsCvb>
__global__ __launch_bounds__(256) void maxpool_s1_lds_bwd_kernel(const PoolParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pool_smem[];
  const int HW = p.H * p.W;
  uint4* const gs = reinterpret_cast<uint4*>(pool_smem);                               // [H*W][CVB] dy
  uint2* const is = reinterpret_cast<uint2*>(pool_smem + (size_t)HW * kPoolLdsCvb * 16);  // [H*W][CVB] arg-max bytes
  const int CV = p.C >> 3;
  const int groups = (CV + kPoolLdsCvb - 1) / kPoolLdsCvb;
  const int n = blockIdx.x / groups, cv0 = (blockIdx.x - n * groups) * kPoolLdsCvb;
  const int cvb = min(kPoolLdsCvb, CV - cv0);
  for (int i = threadIdx.x; i < HW * kPoolLdsCvb; i += 256) {
    const int pix = i / kPoolLdsCvb, cv = i - pix * kPoolLdsCvb;
    if (cv < cvb) {
      const int64_t opix = (int64_t)n * HW + pix;
      gs[i] = *reinterpret_cast<const uint4*>(p.dy + opix * p.ld_dy + (cv0 + cv) * 8);
      is[i] = *reinterpret_cast<const uint2*>(p.cidx + opix * p.C + (cv0 + cv) * 8);
    }
  }
  __syncthreads();
  constexpr int PAD = K / 2;
  for (int i = threadIdx.x; i < HW * kPoolLdsCvb; i += 256) {
    const int pix = i / kPoolLdsCvb, cv = i - pix * kPoolLdsCvb;
    if (cv >= cvb) continue;
    const int ih = pix / p.W, iw = pix - ih * p.W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // window (oh, ow) covers (ih, iw) with tap kh = ih - oh + PAD, kw = iw - ow + PAD; same accumulation order as maxpool_bwd_kernel
    // (oh ascending, then ow ascending): bit-identical sums
    const int oh_lo = max(0, ih - PAD), oh_hi = min(p.H - 1, ih + PAD);
    const int ow_lo = max(0, iw - PAD), ow_hi = min(p.W - 1, iw + PAD);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int kh = ih - oh + PAD;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const unsigned off = (unsigned)(kh * K + (iw - ow + PAD));
        const int o = (oh * p.W + ow) * kPoolLdsCvb + cv;
        const f32x8 g = unpack8(gs[o]);
        const uint2 iv = is[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned b = ((j < 4 ? iv.x : iv.y) >> (8 * (j & 3))) & 0xffu;
          if (b == off) acc[j] += g.v[j];
        }
      }
    }
    h16_t* const dst = p.dx + ((int64_t)n * HW + pix) * p.ld_dx + (cv0 + cv) * 8;
    f32x8 o;
    if (p.accumulate) {
      const f32x8 old = unpack8(*reinterpret_cast<const uint4*>(dst));
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = old.v[j] + acc[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    }
    *reinterpret_cast<uint4*>(dst) = pack8(o);
  }
}

static bool pool_lds_ok(const PoolParams& p, bool bwd) {
  auto a16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVHIP_POOL_LDS");  // 0: the gather kernels for every shape (A/B switch)
    on = (e && e[0] == '0') ? 0 : 1;
  }
  if (!on || p.s != 1 || !(p.k & 1) || p.pad != p.k / 2 || p.OH != p.H || p.OW != p.W || (p.k != 3 && p.k != 5 && p.k != 9 && p.k != 13)) return false;
  const int cvb = pool_lds_cvb(p.H * p.W, bwd);
  if ((p.C & 7) || p.H * p.W > kPoolLdsMaxPix || cvb == 0 || (int64_t)p.N * ((p.C / 8 + cvb - 1) / cvb) < 128) return false;
  if (bwd) return (p.ld_dy & 7) == 0 && (p.ld_dx & 7) == 0 && a16(p.dy) && a16(p.dx) && ((((uintptr_t)p.cidx) & 7) == 0);
  return (p.ld_x & 7) == 0 && (p.ld_y & 7) == 0 && a16(p.x) && a16(p.y) && (!p.idx || ((((uintptr_t)p.idx) & 7) == 0));
}

template <int K, int CVB>
static int launch_pool_lds(const PoolParams& p, bool bwd, hipStream_t s) {
  constexpr int kPoolLdsCvb = CVB;
  const int groups = (p.C / 8 + kPoolLdsCvb - 1) / kPoolLdsCvb;
  const int lds = p.H * p.W * kPoolLdsCvb * (bwd ? 24 : 40);
  auto kf = maxpool_s1_lds_fwd_kernel<K, CVB>;
  auto kb = maxpool_s1_lds_bwd_kernel<K, CVB>;
  static bool attr_done[2][64] = {};
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& done = attr_done[bwd ? 1 : 0][devid & 63];
  if (!done) {
    const int cap = 160 * 1024;  // the CU's LDS (pool_lds_ok keeps every launch below it)
    hipError_t e = bwd ? hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, cap)
                       : hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(maxpool_s1_lds)", e);
      return CVHIP_ERR_LAUNCH;
    }
    done = true;
  }
  if (bwd) hipLaunchKernelGGL(kb, dim3(p.N * groups), dim3(256), lds, s, p);
  else hipLaunchKernelGGL(kf, dim3(p.N * groups), dim3(256), lds, s, p);
  return check_launch(bwd ? "maxpool_s1_lds_bwd_kernel" : "maxpool_s1_lds_fwd_kernel");
}

static int try_pool_lds(const PoolParams& p, bool bwd, hipStream_t s) {
  if (!pool_lds_ok(p, bwd)) return -1;
  const int cvb = pool_lds_cvb(p.H * p.W, bwd);
#define CVHIP_POOL_K(KV)                                       \
  if (cvb == 4) return launch_pool_lds<KV, 4>(p, bwd, s);     \
  if (cvb == 2) return launch_pool_lds<KV, 2>(p, bwd, s);     \
  return launch_pool_lds<KV, 1>(p, bwd, s);
  switch (p.k) {
    case 3: CVHIP_POOL_K(3)
    case 5: CVHIP_POOL_K(5)
    case 9: CVHIP_POOL_K(9)
    default: CVHIP_POOL_K(13)
  }
#undef CVHIP_POOL_K
}

// ---------------------------------------------------------------------------------------------------
// nearest x2 upsample + channel concat
// ---------------------------------------------------------------------------------------------------
struct UpParams {
  const h16_t *a, *b;
  h16_t* out;
  int ld_a, ld_b, ld_out, Ca, Cb, N, Ha, Wa;
};

__global__ __launch_bounds__(256) void up2cat_fwd_kernel(const UpParams p) {
  const int Ct = p.Ca + p.Cb;
  const int CVa = (p.Ca + 7) >> 3, CVb = (p.Cb + 7) >> 3;
  const int CV = CVa + CVb;
  const int OH = p.Ha * 2, OW = p.Wa * 2;
  const bool vec = (p.Ca & 7) == 0 && (p.Cb & 7) == 0 && (p.ld_a & 7) == 0 && (p.ld_b & 7) == 0 &&
                   (p.ld_out & 7) == 0 && aligned16(p.a) && aligned16(p.b) && aligned16(p.out);
  (void)Ct;
  const int64_t total = (int64_t)p.N * OH * OW * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, ow, oh, n;
    split_index(i, CV, OW, OH, &cv, &ow, &oh, &n);
    h16_t* orow = p.out + ((int64_t)(n * OH + oh) * OW + ow) * p.ld_out;
    if (cv < CVa) {
      const f32x8 v = load8(p.a + ((int64_t)(n * p.Ha + (oh >> 1)) * p.Wa + (ow >> 1)) * p.ld_a, cv * 8, p.Ca, vec);
      store8(orow, cv * 8, p.Ca, vec, v);
    } else {
      const int cb = (cv - CVa) * 8;
      const f32x8 v = load8(p.b + ((int64_t)(n * OH + oh) * OW + ow) * p.ld_b, cb, p.Cb, vec);
      store8(orow + p.Ca, cb, p.Cb, vec, v);
    }
  }
}

__global__ __launch_bounds__(256) void up2_bwd_kernel(const UpParams p) {
  // p.out = dout (read), p.a = da (write)
  const int CV = (p.Ca + 7) >> 3;
  const int OW = p.Wa * 2, OH = p.Ha * 2;
  const bool vec = (p.Ca & 7) == 0 && (p.ld_a & 7) == 0 && (p.ld_out & 7) == 0 && aligned16(p.a) && aligned16(p.out);
  const int64_t total = (int64_t)p.N * p.Ha * p.Wa * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, w, h, n;
    split_index(i, CV, p.Wa, p.Ha, &cv, &w, &h, &n);
    f32x8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const f32x8 v = load8(p.out + ((int64_t)(n * OH + 2 * h + dy) * OW + 2 * w + dx) * p.ld_out, cv * 8, p.Ca, vec);
#pragma unroll
        for (int j = 0; j < 8; ++j) s.v[j] += v.v[j];
      }
    store8(const_cast<h16_t*>(p.a) + ((int64_t)(n * p.Ha + h) * p.Wa + w) * p.ld_a, cv * 8, p.Ca, vec, s);
  }
}

// ---------------------------------------------------------------------------------------------------
// bilinear resize (ATen upsample_bilinear2d index rule)
// ---------------------------------------------------------------------------------------------------
struct BilParams {
  const h16_t* src;
  h16_t* dst;
  int ld_src, ld_dst, N, C, Hi, Wi, Ho, Wo, align;
  float sh, sw;  // source-index scale
};

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const BilParams p) {
  const int CV = (p.C + 7) >> 3;
  // 16-B vectors also for odd channel counts when both pitches leave room for the padded tail vector (19-class logits in
  // ld = 24 buffers: 3 vector accesses per pixel instead of 19 scalar ones; pad lanes carry pad values through)
  const bool vec = (p.ld_src & 7) == 0 && (p.ld_dst & 7) == 0 && p.ld_src >= ((p.C + 7) & ~7) && p.ld_dst >= ((p.C + 7) & ~7) &&
                   aligned16(p.src) && aligned16(p.dst);
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, ow, oh, n;
    split_index(i, CV, p.Wo, p.Ho, &cv, &ow, &oh, &n);
    int h0, h1, w0, w1;
    float lh, lw;
    bil_src(oh, p.sh, p.align, p.Hi, &h0, &h1, &lh);
    bil_src(ow, p.sw, p.align, p.Wi, &w0, &w1, &lw);
    const h16_t* base = p.src + (int64_t)n * p.Hi * p.Wi * p.ld_src;
    const f32x8 v00 = load8(base + ((int64_t)h0 * p.Wi + w0) * p.ld_src, cv * 8, p.C, vec);
    const f32x8 v01 = load8(base + ((int64_t)h0 * p.Wi + w1) * p.ld_src, cv * 8, p.C, vec);
    const f32x8 v10 = load8(base + ((int64_t)h1 * p.Wi + w0) * p.ld_src, cv * 8, p.C, vec);
    const f32x8 v11 = load8(base + ((int64_t)h1 * p.Wi + w1) * p.ld_src, cv * 8, p.C, vec);
    const float a0 = 1.f - lh, a1 = lh, b0 = 1.f - lw, b1 = lw;
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = a0 * (b0 * v00.v[j] + b1 * v01.v[j]) + a1 * (b0 * v10.v[j] + b1 * v11.v[j]);
    store8(p.dst + ((int64_t)(n * p.Ho + oh) * p.Wo + ow) * p.ld_dst, cv * 8, p.C, vec, o);
  }
}

// ---- nearest-neighbour resize to an arbitrary size (F.interpolate(mode="nearest"): STDC neck stdcnet.py / ffm paths) -----------
// torch's rule: src = min(floor(dst * scale), in - 1), scale = (float)in / out. Forward is an exact copy; backward is a
// deterministic gather: every INPUT pixel sums the (contiguous) run of output pixels that map to it.
__device__ __forceinline__ int nn_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ __launch_bounds__(256) void nearest_fwd_kernel(const BilParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.ld_src & 7) == 0 && (p.ld_dst & 7) == 0 && p.ld_src >= ((p.C + 7) & ~7) && p.ld_dst >= ((p.C + 7) & ~7) &&
                   aligned16(p.src) && aligned16(p.dst);
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, ow, oh, n;
    split_index(i, CV, p.Wo, p.Ho, &cv, &ow, &oh, &n);
    const int ih = nn_src(oh, p.sh, p.Hi), iw = nn_src(ow, p.sw, p.Wi);
    const f32x8 v = load8(p.src + ((int64_t)(n * p.Hi + ih) * p.Wi + iw) * p.ld_src, cv * 8, p.C, vec);
    store8(p.dst + ((int64_t)(n * p.Ho + oh) * p.Wo + ow) * p.ld_dst, cv * 8, p.C, vec, v);
  }
}

// [lo, hi) = outputs o with nn_src(o) == i (monotone in o): start below the analytic position, walk to the exact ends
__device__ __forceinline__ void nn_range(int i, float scale, int in, int out, int* lo, int* hi) {
  int o = (int)floorf((float)i / scale) - 2;
  if (o < 0) o = 0;
  while (o < out && nn_src(o, scale, in) < i) ++o;
  *lo = o;
  while (o < out && nn_src(o, scale, in) == i) ++o;
  *hi = o;
}

// here src = the gradient at the OUTPUT (Ho x Wo), dst = the gradient at the input (Hi x Wi)
__global__ __launch_bounds__(256) void nearest_bwd_kernel(const BilParams p) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.ld_src & 7) == 0 && (p.ld_dst & 7) == 0 && p.ld_src >= ((p.C + 7) & ~7) && p.ld_dst >= ((p.C + 7) & ~7) &&
                   aligned16(p.src) && aligned16(p.dst);
  const int64_t total = (int64_t)p.N * p.Hi * p.Wi * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, iw, ih, n;
    split_index(i, CV, p.Wi, p.Hi, &cv, &iw, &ih, &n);
    int h0, h1, w0, w1;
    nn_range(ih, p.sh, p.Hi, p.Ho, &h0, &h1);
    nn_range(iw, p.sw, p.Wi, p.Wo, &w0, &w1);
    f32x8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
    for (int oh = h0; oh < h1; ++oh)
      for (int ow = w0; ow < w1; ++ow) {
        const f32x8 v = load8(p.src + ((int64_t)(n * p.Ho + oh) * p.Wo + ow) * p.ld_src, cv * 8, p.C, vec);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] += v.v[j];
      }
    store8(p.dst + ((int64_t)(n * p.Hi + ih) * p.Wi + iw) * p.ld_dst, cv * 8, p.C, vec, acc);
  }
}

__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const BilParams p) {
  // p.src = dy (Ho x Wo, read), p.dst = dx (Hi x Wi, write)
  const int CV = (p.C + 7) >> 3;
  // 16-B vectors also for odd channel counts when both pitches leave room for the padded tail vector (19-class logits in
  // ld = 24 buffers: 3 vector accesses per pixel instead of 19 scalar ones; pad lanes carry pad values through)
  const bool vec = (p.ld_src & 7) == 0 && (p.ld_dst & 7) == 0 && p.ld_src >= ((p.C + 7) & ~7) && p.ld_dst >= ((p.C + 7) & ~7) &&
                   aligned16(p.src) && aligned16(p.dst);
  const int64_t total = (int64_t)p.N * p.Hi * p.Wi * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cv, iw, ih, n;
    split_index(i, CV, p.Wi, p.Hi, &cv, &iw, &ih, &n);
    int oh_lo, oh_hi, ow_lo, ow_hi;
    bil_range(ih, p.sh, p.align, p.Ho, &oh_lo, &oh_hi);
    bil_range(iw, p.sw, p.align, p.Wo, &ow_lo, &ow_hi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const h16_t* base = p.src + (int64_t)n * p.Ho * p.Wo * p.ld_src;
    // column weights depend on ow only: computed once per thread, not per row. The conservative range of bil_range is first tightened
    // to the columns that really read this source column (<= 2r of them for an r-fold upsampling: 8 for x4, 16 for x8).
    constexpr int kBilCols = 16;
    float wcol[kBilCols];
    auto colw = [&](int ow) {
      int w0, w1;
      float lw;
      bil_src(ow, p.sw, p.align, p.Wi, &w0, &w1, &lw);
      float ww = 0.f;
      if (w0 == iw) ww += 1.f - lw;
      if (w1 == iw) ww += lw;
      return ww;
    };
    while (ow_lo < ow_hi && colw(ow_lo) == 0.f) ++ow_lo;
    while (ow_hi > ow_lo && colw(ow_hi) == 0.f) --ow_hi;
    const int ncol = ow_hi - ow_lo + 1;
    const bool hoisted = ncol <= kBilCols;
    if (hoisted) {
#pragma unroll
      for (int c = 0; c < kBilCols; ++c) wcol[c] = c < ncol ? colw(ow_lo + c) : 0.f;
    }
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      int h0, h1;
      float lh;
      bil_src(oh, p.sh, p.align, p.Hi, &h0, &h1, &lh);
      float wh = 0.f;
      if (h0 == ih) wh += 1.f - lh;
      if (h1 == ih) wh += lh;
      if (wh == 0.f) continue;
      if (hoisted) {
        // four columns per batch, loaded UNCONDITIONALLY (index clamped, weight 0 past the range): a load guarded by its weight sat
        // in its own branch with an s_waitcnt vmcnt(0) behind it — one 16-byte load in flight per lane (0.8 TB/s on the x8 decoder
        // upsampling of DeepLabv3+)
        const h16_t* const row = base + ((int64_t)oh * p.Wo + ow_lo) * p.ld_src;
#pragma unroll
        for (int c0 = 0; c0 < kBilCols; c0 += 4) {
          if (c0 < ncol) {
            f32x8 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int cc = c0 + u < ncol ? c0 + u : ncol - 1;
              g[u] = load8(row + (int64_t)cc * p.ld_src, cv * 8, p.C, vec);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float wgt = wh * wcol[c0 + u];   // 0 past the range
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += wgt * g[u].v[j];
            }
          }
        }
        continue;
      }
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        int w0, w1;
        float lw;
        bil_src(ow, p.sw, p.align, p.Wi, &w0, &w1, &lw);
        float ww = 0.f;
        if (w0 == iw) ww += 1.f - lw;
        if (w1 == iw) ww += lw;
        if (ww == 0.f) continue;
        const f32x8 g = load8(base + ((int64_t)oh * p.Wo + ow) * p.ld_src, cv * 8, p.C, vec);
        const float wgt = wh * ww;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wgt * g.v[j];
      }
    }
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    store8(p.dst + ((int64_t)(n * p.Hi + ih) * p.Wi + iw) * p.ld_dst, cv * 8, p.C, vec, o);
  }
}

// ---- separable backward for large upsampling ratios (round 3) ----------------------------------------------------------------------------
// The gather above reads every dy element four times (two source rows x two source columns reference it): at r = 8 (DeepLabv3+'s decoder:
// 16x32 -> 128x256, 512 channels, 537 MB of dy) that is 2.1 GB through L2 and 322 us. Bilinear interpolation is separable, so is its
// transpose:  T[n][ih][ow][c] = sum_oh wy(oh -> ih) dy[n][oh][ow][c]     (vertical pass: dy is read exactly ONCE, coalesced)
//             dx[n][ih][iw][c] = sum_ow wx(ow -> iw) T[n][ih][ow][c]     (horizontal pass over the Ho/Hi-times smaller T, fp32)
// Vertical pass: a thread owns (n, ow, channel vector) and walks the label rows top to bottom; src row indices are non-decreasing in oh,
// so two running accumulators (rows cur, cur + 1) suffice and every T row is written once. Four rows of loads in flight per lane.
__global__ __launch_bounds__(256) void bilinear_bwd_v_kernel(const BilParams p, float* __restrict__ T) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.ld_src & 7) == 0 && p.ld_src >= ((p.C + 7) & ~7) && aligned16(p.src);
  const int64_t total = (int64_t)p.N * p.Wo * CV;
  const int Cp = CV * 8;  // T pitch (floats)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t q = i / CV;
    const int ow = (int)(q % p.Wo);
    const int n = (int)(q / p.Wo);
    const h16_t* col = p.src + ((int64_t)n * p.Ho * p.Wo + ow) * p.ld_src;
    float* tcol = T + ((int64_t)n * p.Hi * p.Wo + ow) * Cp + cv * 8;
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
    int cur = 0;
    auto flush = [&]() {  // T[cur] = a0; shift
      float* dst = tcol + (int64_t)cur * p.Wo * Cp;
      *reinterpret_cast<float4*>(dst) = make_float4(a0[0], a0[1], a0[2], a0[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(a0[4], a0[5], a0[6], a0[7]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = a1[j];
        a1[j] = 0.f;
      }
      ++cur;
    };
    for (int oh0 = 0; oh0 < p.Ho; oh0 += 4) {
      f32x8 g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int oh = oh0 + u < p.Ho ? oh0 + u : p.Ho - 1;  // clamped: rows past the end get weight 0 below
        g[u] = load8(col + (int64_t)oh * p.Wo * p.ld_src, cv * 8, p.C, vec);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (oh0 + u < p.Ho) {
          int h0, h1;
          float lh;
          bil_src(oh0 + u, p.sh, p.align, p.Hi, &h0, &h1, &lh);
          while (cur < h0) flush();
          const float w0 = h1 == h0 ? 1.f : 1.f - lh, w1 = h1 == h0 ? 0.f : lh;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a0[j] += w0 * g[u].v[j];
            a1[j] += w1 * g[u].v[j];
          }
        }
      }
    }
    while (cur < p.Hi) flush();  // the last touched rows, and zeros for source rows no label row reads
  }
}

__global__ __launch_bounds__(256) void bilinear_bwd_h_kernel(const BilParams p, const float* __restrict__ T) {
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.ld_dst & 7) == 0 && p.ld_dst >= ((p.C + 7) & ~7) && aligned16(p.dst);
  const int Cp = CV * 8;
  const int64_t total = (int64_t)p.N * p.Hi * p.Wi * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t pix = i / CV;
    const int iw = (int)(pix % p.Wi);
    pix /= p.Wi;  // = n * Hi + ih
    int ow_lo, ow_hi;
    bil_range(iw, p.sw, p.align, p.Wo, &ow_lo, &ow_hi);
    const float* trow = T + (pix * p.Wo) * Cp + cv * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int c0 = ow_lo; c0 <= ow_hi; c0 += 4) {
      float4 ga[4], gb[4];
      float wgt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ow = c0 + u <= ow_hi ? c0 + u : ow_hi;
        ga[u] = *reinterpret_cast<const float4*>(trow + (int64_t)ow * Cp);
        gb[u] = *reinterpret_cast<const float4*>(trow + (int64_t)ow * Cp + 4);
        int w0, w1;
        float lw;
        bil_src(ow, p.sw, p.align, p.Wi, &w0, &w1, &lw);
        float ww = 0.f;
        if (w0 == iw) ww += 1.f - lw;
        if (w1 == iw) ww += lw;
        wgt[u] = c0 + u <= ow_hi ? ww : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0] += wgt[u] * ga[u].x;
        acc[1] += wgt[u] * ga[u].y;
        acc[2] += wgt[u] * ga[u].z;
        acc[3] += wgt[u] * ga[u].w;
        acc[4] += wgt[u] * gb[u].x;
        acc[5] += wgt[u] * gb[u].y;
        acc[6] += wgt[u] * gb[u].z;
        acc[7] += wgt[u] * gb[u].w;
      }
    }
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    store8(p.dst + pix * p.Wi * p.ld_dst + (int64_t)iw * p.ld_dst, cv * 8, p.C, vec, o);
  }
}

// uint8 NHWC image batch -> normalised bf16 NHWC with channels zero-padded to ld: y = x * scale[c] + shift[c]
// (scale = 1 / (255 * std), shift = -mean / std: ToTensor + Normalize of the reference's CPU transforms fused with the
// relayout the stem conv wants). One thread per pixel: C (<= 8) byte loads, one 16-byte store per 8 output channels.
__global__ __launch_bounds__(256) void u8_norm_kernel(const unsigned char* __restrict__ x, h16_t* __restrict__ y, int64_t npix, int C, int ld,
                                                      const float* __restrict__ scale, const float* __restrict__ shift) {
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = j < C ? scale[j] : 0.f;
    sh[j] = j < C ? shift[j] : 0.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const unsigned char* px = x + i * C;
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = j < C ? (float)px[j] * sc[j] + sh[j] : 0.f;
    h16_t* dst = y + i * ld;
    if ((ld & 7) == 0 && aligned16(y)) {
      *reinterpret_cast<uint4*>(dst) = pack8(o);
      for (int c = 8; c < ld; ++c) dst[c] = (h16_t)0.f;
    } else {
      for (int c = 0; c < ld; ++c) dst[c] = (h16_t)(c < 8 ? o.v[c] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// global average pool
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_fwd_kernel(const h16_t* x, int ld_x, h16_t* y, int N, int C, int HW) {
  // block = (image n, 32 channels); 256 threads = 8 row-lanes x 32 channels
  __shared__ float red[8][33];
  const int n = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), ry = threadIdx.x >> 5;
  float s = 0.f;
  if (c < C)
    for (int r = ry; r < HW; r += 8) s += (float)x[((int64_t)n * HW + r) * ld_x + c];
  red[ry][threadIdx.x & 31] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    y[(int64_t)n * C + c] = (h16_t)(t / (float)HW);
  }
}

// Round 6: the same with 1024 threads = 256 row lanes x four 16-byte channel vectors (STDC's attention modules pool 16 x 128..1024
// channels over 2048 / 512 pixels: the form above ran 64 - 512 blocks whose threads walked 256 - 1024 two-byte loads one after the
// other, 88 us per launch; here a lane makes HW / 256 sixteen-byte loads, four in flight)
__global__ __launch_bounds__(1024) void gap_fwd_vec_kernel(const h16_t* __restrict__ x, int ld_x, h16_t* __restrict__ y, int N, int C, int HW) {
  __shared__ float red[256][4][8];
  const int n = blockIdx.y, tx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const int c = (blockIdx.x * 4 + tx) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c < C) {
    const h16_t* base = x + (int64_t)n * HW * ld_x + c;
    int r = ry;
    for (; r + 3 * 256 < HW; r += 4 * 256) {
      uint4 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const uint4*>(base + (int64_t)(r + q * 256) * ld_x);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x8 v = unpack8(u[q]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v.v[j];
      }
    }
    for (; r < HW; r += 256) {
      const f32x8 v = unpack8(*reinterpret_cast<const uint4*>(base + (int64_t)r * ld_x));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ry][tx][j] = acc[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (ry < s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[ry][tx][j] += red[ry + s][tx][j];
    }
    __syncthreads();
  }
  if (ry == 0 && c < C) {
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = red[0][tx][j] / (float)HW;
    *reinterpret_cast<uint4*>(y + (int64_t)n * C + c) = pack8(o);
  }
}

// ds[n][c] = sum over the pixels of dy[n][p][c] * x[n][p][c] (fp32): the gate's gradient of y = x * s[n][c] (attention-refinement / feature-
// fusion modules, stdc_neck.py:53-58,110-114) — the torch form converted both maps to fp32, multiplied and ran a strided reduce_kernel:
// four passes and ~110 us per module; here both maps are read once, 16 bytes per lane, in gap_fwd_vec_kernel's block shape
__global__ __launch_bounds__(1024) void chscale_ds_kernel(const h16_t* __restrict__ dy, int ld_dy, const h16_t* __restrict__ x, int ld_x,
                                                           float* __restrict__ ds, int N, int C, int HW) {
  __shared__ float red[256][4][8];
  const int n = blockIdx.y, tx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const int c = (blockIdx.x * 4 + tx) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c < C) {
    const h16_t* bd = dy + (int64_t)n * HW * ld_dy + c;
    const h16_t* bx = x + (int64_t)n * HW * ld_x + c;
    int r = ry;
    for (; r + 256 < HW; r += 2 * 256) {
      uint4 ud[2], ux[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ud[q] = *reinterpret_cast<const uint4*>(bd + (int64_t)(r + q * 256) * ld_dy);
        ux[q] = *reinterpret_cast<const uint4*>(bx + (int64_t)(r + q * 256) * ld_x);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x8 a = unpack8(ud[q]), b = unpack8(ux[q]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += a.v[j] * b.v[j];
      }
    }
    for (; r < HW; r += 256) {
      const f32x8 a = unpack8(*reinterpret_cast<const uint4*>(bd + (int64_t)r * ld_dy)), b = unpack8(*reinterpret_cast<const uint4*>(bx + (int64_t)r * ld_x));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += a.v[j] * b.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ry][tx][j] = acc[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (ry < s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[ry][tx][j] += red[ry + s][tx][j];
    }
    __syncthreads();
  }
  if (ry == 0 && c < C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ds[(int64_t)n * C + c + j] = red[0][tx][j];
  }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const h16_t* dy, h16_t* dx, int ld_dx, int N, int C, int HW) {
  const int64_t total = (int64_t)N * HW * C;
  const float inv = 1.f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int n = (int)(pix / HW);
    dx[pix * ld_dx + c] = (h16_t)((float)dy[(int64_t)n * C + c] * inv);
  }
}

// ---------------------------------------------------------------------------------------------------
// layout conversions at the torch boundary
// Focus relayout of a CC-channel (<= 4) fp32 NCHW image into NHWC rows of 16 halfwords: see nchw_to_nhwc_kernel's fast path
template <int CC>
__device__ __forceinline__ void focus_small(const float* __restrict__ x, h16_t* __restrict__ y, int H, int W, int OH, int OW, unsigned total) {
  const unsigned uOW = (unsigned)OW, uOH = (unsigned)OH;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / uOW, ow = i - r * uOW;
    const unsigned n = r / uOH, oh = r - n * uOH;
    float o16[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o16[c] = 0.f;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      const float* row0 = x + ((int64_t)(n * CC + cc) * H + 2 * oh) * W + 2 * ow;
      const float2 t = *reinterpret_cast<const float2*>(row0);        // TL, TR
      const float2 bt = *reinterpret_cast<const float2*>(row0 + W);   // BL, BR
      o16[0 * CC + cc] = t.x;    // patch order TL, BL, TR, BR (yolox_csp_darknet.py Focus.forward)
      o16[1 * CC + cc] = bt.x;
      o16[2 * CC + cc] = t.y;
      o16[3 * CC + cc] = bt.y;
    }
    f32x8 lo, hi;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      lo.v[c] = o16[c];
      hi.v[c] = o16[8 + c];
    }
    uint4* dst = reinterpret_cast<uint4*>(y) + (int64_t)i * 2;
    dst[0] = pack8(lo);
    dst[1] = pack8(hi);
  }
}

// ---------------------------------------------------------------------------------------------------
// mode 0: plain NCHW fp32 -> NHWC bf16 (pitch ld, channels >= C zero-filled up to Cfill)
// mode 1: Focus space-to-depth: out (N, H/2, W/2, 4C) channel order TL, BL, TR, BR
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, h16_t* y, int N, int C, int H, int W, int ld,
                                                           int Cfill, int focus) {
  const int OH = focus ? H / 2 : H, OW = focus ? W / 2 : W;
  const int64_t total = (int64_t)N * OH * OW;
  if (!focus && C <= 8 && Cfill == 8 && ld == 8 && ((((uintptr_t)y) & 15) == 0)) {
    // image stem: one pixel per thread — C coalesced plane reads, ONE 16-byte store (pad channels zero)
    const int64_t HW = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t n = i / HW, hw = i - n * HW;
      const float* src = x + n * C * HW + hw;
      f32x8 v;
#pragma unroll
      for (int c = 0; c < 8; ++c) v.v[c] = c < C ? src[c * HW] : 0.f;
      reinterpret_cast<uint4*>(y)[i] = pack8(v);
    }
    return;
  }
  if (focus && 4 * C <= 16 && Cfill == 16 && ld == 16 && ((((uintptr_t)y) & 15) == 0) && ((((uintptr_t)x) & 7) == 0) && total < (1ll << 31)) {
    // Focus stem of a <= 4-channel image (round 6; the generic loop below issued 12 strided scalar loads and 16 two-byte stores per
    // output pixel behind three 64-bit divisions: 341 us for YOLOX-s at batch 64, 1.5 TB/s): one output pixel per thread, the two
    // input rows of every channel as 8-byte loads (consecutive lanes, consecutive pairs), TWO 16-byte stores
    switch (C) {
      case 1: focus_small<1>(x, y, H, W, OH, OW, (unsigned)total); break;
      case 2: focus_small<2>(x, y, H, W, OH, OW, (unsigned)total); break;
      case 3: focus_small<3>(x, y, H, W, OH, OW, (unsigned)total); break;
      default: focus_small<4>(x, y, H, W, OH, OW, (unsigned)total); break;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ow = (int)(i % OW);
    const int oh = (int)((i / OW) % OH);
    const int n = (int)(i / ((int64_t)OW * OH));
    h16_t* dst = y + i * ld;
    const int Cout = focus ? 4 * C : C;
    for (int c = 0; c < Cout; ++c) {
      float v;
      if (focus) {
        const int patch = c / C, cc = c - patch * C;
        const int dh = patch & 1, dw = patch >> 1;  // order TL(0,0) BL(1,0) TR(0,1) BR(1,1)
        v = x[((int64_t)(n * C + cc) * H + (2 * oh + dh)) * W + (2 * ow + dw)];
      } else {
        v = x[((int64_t)(n * C + c) * H + oh) * W + ow];
      }
      dst[c] = (h16_t)v;
    }
    for (int c = Cout; c < Cfill; ++c) dst[c] = (h16_t)0.f;
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const h16_t* x, int ld, float* y, int N, int C, int H, int W) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)N * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / HW, hw = i - n * HW;
    const h16_t* src = x + i * ld;
    for (int c = 0; c < C; ++c) y[(n * C + c) * HW + hw] = (float)src[c];
  }
}


// YOLO head boundary: bf16 NHWC (N,H,W,ld>=A*NO) <-> fp32 (N,A,H,W,NO) contiguous.
// Fuses the reference's x.view(bs,na,no,ny,nx).permute(0,1,3,4,2).contiguous() (+ fp32 cast for the
// loss) — src/models/detects/yolov5_detect.py:43-44.
__global__ __launch_bounds__(256) void head_permute_fwd_kernel(const h16_t* x, int ld, float* y, int N, int A, int NO, int H, int W) {
  const int64_t total = (int64_t)N * A * H * W * NO;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int o = (int)(i % NO);
    int64_t r = i / NO;
    const int xw = (int)(r % W);
    r /= W;
    const int yh = (int)(r % H);
    r /= H;
    const int a = (int)(r % A);
    const int n = (int)(r / A);
    y[i] = (float)x[((int64_t)(n * H + yh) * W + xw) * ld + a * NO + o];
  }
}
__global__ __launch_bounds__(256) void head_permute_bwd_kernel(const float* dy, h16_t* dx, int ld, int N, int A, int NO, int H, int W) {
  // one thread per (pixel, channel<ld); pad channels >= A*NO are zero-filled
  const int64_t total = (int64_t)N * H * W * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % ld);
    int64_t pix = i / ld;
    float v = 0.f;
    if (c < A * NO) {
      const int a = c / NO, o = c - a * NO;
      const int xw = (int)(pix % W);
      const int yh = (int)((pix / W) % H);
      const int n = (int)(pix / ((int64_t)W * H));
      v = dy[((((int64_t)n * A + a) * H + yh) * W + xw) * NO + o];
    }
    dx[i] = (h16_t)v;
  }
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

int cvhip_maxpool2d_fwd(const void* x, int32_t ld_x, void* y, int32_t ld_y, uint8_t* argmax, int32_t N, int32_t C,
                        int32_t H, int32_t W, int32_t k, int32_t stride, int32_t pad, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0) return CVHIP_ERR_INVALID;
  if (k * k > 255 || pad * 2 > k) return CVHIP_ERR_UNSUPPORTED;
  PoolParams p{};
  p.x = (const h16_t*)x;
  p.y = (h16_t*)y;
  p.idx = argmax;
  p.ld_x = ld_x;
  p.ld_y = ld_y;
  p.N = N;
  p.C = C;
  p.H = H;
  p.W = W;
  p.k = k;
  p.s = stride;
  p.pad = pad;
  p.OH = (H + 2 * pad - k) / stride + 1;
  p.OW = (W + 2 * pad - k) / stride + 1;
  {
    const int rc = try_pool_lds(p, false, (hipStream_t)stream);
    if (rc >= 0 || rc < -1) return rc;
  }
  const int64_t total = (int64_t)N * p.OH * p.OW * ((C + 7) / 8);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("maxpool_fwd_kernel");
}

int cvhip_maxpool2d_bwd(const void* dy, int32_t ld_dy, const uint8_t* argmax, void* dx, int32_t ld_dx, int32_t N,
                        int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride, int32_t pad, int accumulate,
                        void* stream) {
  if (!dy || !dx || !argmax || N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return CVHIP_ERR_INVALID;
  PoolParams p{};
  p.dy = (const h16_t*)dy;
  p.cidx = argmax;
  p.dx = (h16_t*)dx;
  p.ld_dy = ld_dy;
  p.ld_dx = ld_dx;
  p.N = N;
  p.C = C;
  p.H = H;
  p.W = W;
  p.k = k;
  p.s = stride;
  p.pad = pad;
  p.OH = (H + 2 * pad - k) / stride + 1;
  p.OW = (W + 2 * pad - k) / stride + 1;
  p.accumulate = accumulate;
  {
    const int rc = try_pool_lds(p, true, (hipStream_t)stream);
    if (rc >= 0 || rc < -1) return rc;
  }
  const int64_t total = (int64_t)N * H * W * ((C + 7) / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("maxpool_bwd_kernel");
}

int cvhip_upsample2x_cat_fwd(const void* a, int32_t ld_a, int32_t Ca, const void* b, int32_t ld_b, int32_t Cb,
                             void* out, int32_t ld_out, int32_t N, int32_t Ha, int32_t Wa, void* stream) {
  if (!a || !out || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || N <= 0 || Ha <= 0 || Wa <= 0) return CVHIP_ERR_INVALID;
  UpParams p{};
  p.a = (const h16_t*)a;
  p.b = (const h16_t*)b;
  p.out = (h16_t*)out;
  p.ld_a = ld_a;
  p.ld_b = ld_b;
  p.ld_out = ld_out;
  p.Ca = Ca;
  p.Cb = Cb;
  p.N = N;
  p.Ha = Ha;
  p.Wa = Wa;
  const int64_t total = (int64_t)N * Ha * 2 * Wa * 2 * ((Ca + 7) / 8 + (Cb + 7) / 8);
  hipLaunchKernelGGL(up2cat_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("up2cat_fwd_kernel");
}

int cvhip_upsample2x_bwd(const void* dout, int32_t ld_dout, void* da, int32_t ld_da, int32_t Ca, int32_t N, int32_t Ha,
                         int32_t Wa, void* stream) {
  if (!dout || !da || Ca <= 0 || N <= 0 || Ha <= 0 || Wa <= 0) return CVHIP_ERR_INVALID;
  UpParams p{};
  p.out = (h16_t*)const_cast<void*>(dout);
  p.a = (const h16_t*)da;
  p.ld_out = ld_dout;
  p.ld_a = ld_da;
  p.Ca = Ca;
  p.N = N;
  p.Ha = Ha;
  p.Wa = Wa;
  const int64_t total = (int64_t)N * Ha * Wa * ((Ca + 7) / 8);
  hipLaunchKernelGGL(up2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("up2_bwd_kernel");
}

int cvhip_resize_nearest_fwd(const void* x, int32_t ld_x, void* y, int32_t ld_y, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CVHIP_ERR_INVALID;
  BilParams p{};
  p.src = (const h16_t*)x;
  p.dst = (h16_t*)y;
  p.ld_src = ld_x;
  p.ld_dst = ld_y;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.sh = (float)Hi / (float)Ho;
  p.sw = (float)Wi / (float)Wo;
  const int64_t total = (int64_t)N * Ho * Wo * ((C + 7) / 8);
  hipLaunchKernelGGL(nearest_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("nearest_fwd_kernel");
}

int cvhip_resize_nearest_bwd(const void* dy, int32_t ld_dy, void* dx, int32_t ld_dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, void* stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CVHIP_ERR_INVALID;
  BilParams p{};
  p.src = (const h16_t*)dy;
  p.dst = (h16_t*)dx;
  p.ld_src = ld_dy;
  p.ld_dst = ld_dx;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.sh = (float)Hi / (float)Ho;
  p.sw = (float)Wi / (float)Wo;
  const int64_t total = (int64_t)N * Hi * Wi * ((C + 7) / 8);
  hipLaunchKernelGGL(nearest_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("nearest_bwd_kernel");
}

int cvhip_resize_bilinear_fwd(const void* x, int32_t ld_x, void* y, int32_t ld_y, int32_t N, int32_t C, int32_t Hi,
                              int32_t Wi, int32_t Ho, int32_t Wo, int32_t align_corners, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CVHIP_ERR_INVALID;
  BilParams p{};
  p.src = (const h16_t*)x;
  p.dst = (h16_t*)y;
  p.ld_src = ld_x;
  p.ld_dst = ld_y;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.align = align_corners;
  bil_scales(Hi, Wi, Ho, Wo, align_corners, &p.sh, &p.sw);
  const int64_t total = (int64_t)N * Ho * Wo * ((C + 7) / 8);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("bilinear_fwd_kernel");
}

int cvhip_resize_bilinear_bwd(const void* dy, int32_t ld_dy, void* dx, int32_t ld_dx, int32_t N, int32_t C, int32_t Hi,
                              int32_t Wi, int32_t Ho, int32_t Wo, int32_t align_corners, void* stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CVHIP_ERR_INVALID;
  BilParams p{};
  p.src = (const h16_t*)dy;
  p.dst = (h16_t*)dx;
  p.ld_src = ld_dy;
  p.ld_dst = ld_dx;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.align = align_corners;
  bil_scales(Hi, Wi, Ho, Wo, align_corners, &p.sh, &p.sw);
  const int64_t total = (int64_t)N * Hi * Wi * ((C + 7) / 8);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("bilinear_bwd_kernel");
}

int64_t cvhip_resize_bilinear_bwd_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo) {
  // the separable form pays when each dy element would otherwise be fetched four times from far away: ratios >= 4 on both axes
  if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho < 4 * Hi || Wo < 4 * Wi) return 0;
  return (int64_t)N * Hi * Wo * ((C + 7) / 8 * 8) * (int64_t)sizeof(float);
}

int cvhip_resize_bilinear_bwd_ws(const void* dy, int32_t ld_dy, void* dx, int32_t ld_dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                 int32_t Wo, int32_t align_corners, void* workspace, int64_t ws_bytes, void* stream) {
  const int64_t need = cvhip_resize_bilinear_bwd_workspace_bytes(N, C, Hi, Wi, Ho, Wo);
  if (need == 0 || !workspace || ws_bytes < need || (((uintptr_t)workspace) & 15))
    return cvhip_resize_bilinear_bwd(dy, ld_dy, dx, ld_dx, N, C, Hi, Wi, Ho, Wo, align_corners, stream);
  if (!dy || !dx) return CVHIP_ERR_INVALID;
  BilParams p{};
  p.src = (const h16_t*)dy;
  p.dst = (h16_t*)dx;
  p.ld_src = ld_dy;
  p.ld_dst = ld_dx;
  p.N = N;
  p.C = C;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.align = align_corners;
  bil_scales(Hi, Wi, Ho, Wo, align_corners, &p.sh, &p.sw);
  const int CV = (C + 7) / 8;
  hipLaunchKernelGGL(bilinear_bwd_v_kernel, dim3(grid_for((int64_t)N * Wo * CV)), dim3(256), 0, (hipStream_t)stream, p, (float*)workspace);
  int st = check_launch("bilinear_bwd_v_kernel");
  if (st) return st;
  hipLaunchKernelGGL(bilinear_bwd_h_kernel, dim3(grid_for((int64_t)N * Hi * Wi * CV)), dim3(256), 0, (hipStream_t)stream, p, (const float*)workspace);
  return check_launch("bilinear_bwd_h_kernel");
}

int cvhip_global_avgpool_fwd(const void* x, int32_t ld_x, void* y, int32_t N, int32_t C, int32_t HW, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return CVHIP_ERR_INVALID;
  if ((C & 7) == 0 && (ld_x & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && HW >= 256) {
    hipLaunchKernelGGL(gap_fwd_vec_kernel, dim3(cdiv(C, 32), N), dim3(1024), 0, (hipStream_t)stream, (const h16_t*)x, ld_x, (h16_t*)y, N, C, HW);
    return check_launch("gap_fwd_vec_kernel");
  }
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ld_x,
                     (h16_t*)y, N, C, HW);
  return check_launch("gap_fwd_kernel");
}

int cvhip_channel_scale_bwd_ds(const void* dy, int32_t ld_dy, const void* x, int32_t ld_x, float* ds, int32_t N, int32_t C, int32_t HW, void* stream) {
  if (!dy || !x || !ds || N <= 0 || C <= 0 || HW <= 0) return CVHIP_ERR_INVALID;
  if ((C & 7) || (ld_dy & 7) || (ld_x & 7) || ((((uintptr_t)dy) | ((uintptr_t)x)) & 15)) return CVHIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(chscale_ds_kernel, dim3(cdiv(C, 32), N), dim3(1024), 0, (hipStream_t)stream, (const h16_t*)dy, ld_dy, (const h16_t*)x, ld_x, ds, N,
                     C, HW);
  return check_launch("chscale_ds_kernel");
}

int cvhip_global_avgpool_bwd(const void* dy, void* dx, int32_t ld_dx, int32_t N, int32_t C, int32_t HW, void* stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || HW <= 0) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for((int64_t)N * HW * C)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)dy, (h16_t*)dx, ld_dx, N, C, HW);
  return check_launch("gap_bwd_kernel");
}

int cvhip_nchw_f32_to_nhwc_bf16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                                void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                     (h16_t*)y, N, C, H, W, Cpad, Cpad, 0);
  return check_launch("nchw_to_nhwc_kernel");
}

int cvhip_focus_nchw_f32_to_nhwc_bf16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                                      void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cpad < 4 * C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)N * H * W / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (h16_t*)y, N, C, H, W, Cpad, Cpad, 1);
  return check_launch("nchw_to_nhwc_kernel(focus)");
}

int cvhip_nhwc_bf16_to_nchw_f32(const void* x, int32_t ld, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                     (const h16_t*)x, ld, y, N, C, H, W);
  return check_launch("nhwc_to_nchw_kernel");
}

int cvhip_nchw_f32_to_nhwc_bf16_ld(const float* x, void* y, int32_t ld, int32_t N, int32_t C, int32_t H, int32_t W,
                                   void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                     (h16_t*)y, N, C, H, W, ld, C, 0);
  return check_launch("nchw_to_nhwc_kernel(ld)");
}

int cvhip_u8_nhwc_to_bf16_norm(const void* x_u8, int64_t npix, int32_t C, void* y_bf16, int32_t ld, const float* scale, const float* shift,
                               void* stream) {
  if (!x_u8 || !y_bf16 || !scale || !shift || npix <= 0 || C <= 0 || C > 8 || ld < C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(u8_norm_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x_u8, (h16_t*)y_bf16,
                     npix, C, ld, scale, shift);
  return check_launch("u8_norm_kernel");
}

int cvhip_head_permute_fwd(const void* x, int32_t ld, float* y, int32_t N, int32_t A, int32_t NO, int32_t H, int32_t W,
                           void* stream) {
  if (!x || !y || N <= 0 || A <= 0 || NO <= 0 || H <= 0 || W <= 0 || ld < A * NO) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(head_permute_fwd_kernel, dim3(grid_for((int64_t)N * A * H * W * NO)), dim3(256), 0,
                     (hipStream_t)stream, (const h16_t*)x, ld, y, N, A, NO, H, W);
  return check_launch("head_permute_fwd_kernel");
}

int cvhip_head_permute_bwd(const float* dy, void* dx, int32_t ld, int32_t N, int32_t A, int32_t NO, int32_t H, int32_t W,
                           void* stream) {
  if (!dy || !dx || N <= 0 || A <= 0 || NO <= 0 || H <= 0 || W <= 0 || ld < A * NO) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(head_permute_bwd_kernel, dim3(grid_for((int64_t)N * H * W * ld)), dim3(256), 0,
                     (hipStream_t)stream, dy, (h16_t*)dx, ld, N, A, NO, H, W);
  return check_launch("head_permute_bwd_kernel");
}

}  // extern "C"
