// common.h — shared device helpers for libcvhip (gfx950 / CDNA4 only; wave = 64 lanes).
//
// STORAGE PRECISION. Activations, operand images and activation gradients are 16-bit floats (`h16_t`), accumulation is fp32.
// Every source that touches them is compiled TWICE into the same library:
//   default      : h16_t = bf16  (v_mfma_f32_16x16x32_bf16)   entry points cvhip_xxx           namespace cvhip
//   -DCVHIP_F16  : h16_t = fp16  (v_mfma_f32_16x16x32_f16)    entry points cvhip_xxx_f16       namespace cvhip_f16
// (reference: torch.cuda.amp.autocast fp16 + GradScaler, trainer.py:179-201; BASELINE config 5 "fp16"). The kernels are written
// once against h16_t / h16x8 / pack8 / unpack8 / CVHIP_MFMA_*; the renaming of the exported symbols happens in f16_names.h
// (generated from the sources by tools/gen_f16_names.py, checked by tests/test_abi_plan.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef CVHIP_F16
#include "f16_names.h"
#endif
#include "../../include/cvhip.h"

#ifdef CVHIP_F16
typedef _Float16 h16_t;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define CVHIP_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define CVHIP_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
typedef __fp16 cvhip_fp16x4_raw __attribute__((ext_vector_type(4)));  // the builtin's element type is __fp16, not _Float16
#define CVHIP_DS_READ_TR16_B64(ptr) \
  __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) cvhip_fp16x4_raw*)(ptr)))
#define cvhip cvhip_f16  /* the C++ namespace of this translation unit */
#else
typedef __bf16 h16_t;
typedef __bf16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 h16x2 __attribute__((ext_vector_type(2)));
#define CVHIP_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define CVHIP_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define CVHIP_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace cvhip {

void set_last_error(const char* what, hipError_t e);
int check_launch(const char* what);
// zero-fill by a kernel (not hipMemsetAsync: memset nodes proved unreliable under hipGraph replay on this stack)
int zero_fill(void* ptr, size_t bytes, hipStream_t stream);

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 8 bf16 <-> 8 fp32
struct f32x8 {
  float v[8];
};

#ifdef CVHIP_F16
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  const h16x2 t = __builtin_bit_cast(h16x2, w);
  lo = (float)t[0];  // v_cvt_f32_f16 (the high half through SDWA)
  hi = (float)t[1];
}
#else
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  // bf16 -> fp32 is a 16-bit left shift
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xffff0000u);
}
#endif

__device__ __forceinline__ f32x8 unpack8(uint4 u) {
  f32x8 r;
  unpack2(u.x, r.v[0], r.v[1]);
  unpack2(u.y, r.v[2], r.v[3]);
  unpack2(u.z, r.v[4], r.v[5]);
  unpack2(u.w, r.v[6], r.v[7]);
  return r;
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  h16x2 t;
  t[0] = (h16_t)lo;  // RNE; lowers to v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950
  t[1] = (h16_t)hi;
  return __builtin_bit_cast(uint32_t, t);
}

__device__ __forceinline__ uint4 pack8(const f32x8& f) {
  uint4 u;
  u.x = pack2(f.v[0], f.v[1]);
  u.y = pack2(f.v[2], f.v[3]);
  u.z = pack2(f.v[4], f.v[5]);
  u.w = pack2(f.v[6], f.v[7]);
  return u;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(h16_t, h); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  h16_t b = (h16_t)f;
  return __builtin_bit_cast(uint16_t, b);
}

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: the BN/act passes are
// HBM-bound only if the per-element math stays this cheap (8 sigmoids per 16-byte vector)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// activation forward / derivative (as function of the pre-activation u)
__device__ __forceinline__ float act_fwd(float u, int act, float ap) {
  switch (act) {
    case CVHIP_ACT_RELU: return u > 0.f ? u : 0.f;
    case CVHIP_ACT_SILU: return u * sigmoidf_(u);
    case CVHIP_ACT_LEAKY: return u > 0.f ? u : u * ap;
    case CVHIP_ACT_SIGMOID: return sigmoidf_(u);
    case CVHIP_ACT_HSWISH: {
      float r = fminf(fmaxf(u + 3.f, 0.f), 6.f);
      return u * r * (1.f / 6.f);
    }
    default: return u;
  }
}
__device__ __forceinline__ float act_bwd(float u, int act, float ap) {
  switch (act) {
    case CVHIP_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case CVHIP_ACT_SILU: {
      float s = sigmoidf_(u);
      return s * (1.f + u * (1.f - s));
    }
    case CVHIP_ACT_LEAKY: return u > 0.f ? 1.f : ap;
    case CVHIP_ACT_SIGMOID: {
      float s = sigmoidf_(u);
      return s * (1.f - s);
    }
    case CVHIP_ACT_HSWISH: {
      if (u <= -3.f) return 0.f;
      if (u >= 3.f) return 1.f;
      return (2.f * u + 3.f) * (1.f / 6.f);
    }
    default: return 1.f;
  }
}

// BN + activation BACKWARD of 8 channels from (dz, y) with per-channel constants (u = sc*y + sh; dy = sc*du + b1*y + c1: the affine
// form of gamma*invstd*(du - dbeta/M - xhat*dgamma/M), see conv1x1_bwd.hip) — the on-load transform of the fused backward kernels
struct f32x8;
template <int ACT>
__device__ __forceinline__ void bnact_bwd8_into(const float (&dz)[8], const float (&y)[8], const float (&sc)[8], const float (&sh)[8],
                                                const float (&b1)[8], const float (&c1)[8], float ap, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float u = y[j] * sc[j] + sh[j];
    const float du = dz[j] * act_bwd(u, ACT, ap);
    o[j] = sc[j] * du + (b1[j] * y[j] + c1[j]);
  }
}

// 8 per-channel constants for the channel vector starting at c: UNCONDITIONAL clamped loads (a per-element
// `ok ? p[c] : dflt` compiles to 8 exec-masked blocks with an s_waitcnt vmcnt(0) at every join: 16-48 serialized
// ~1 us round trips per thread, i.e. a fixed ~17 us per launch — measured, tools/stream_probe.py)
__device__ __forceinline__ void load8c(const float* __restrict__ p, int c, int C, float (&o)[8]) {
  if (c + 8 <= C && ((((uintptr_t)p) | (uintptr_t)(c * 4)) & 15) == 0) {  // wave-uniform in the vector paths (C % 8 == 0)
    const float4 a = *reinterpret_cast<const float4*>(p + c), b = *reinterpret_cast<const float4*>(p + c + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = p[(c + j < C) ? c + j : C - 1];
}
__device__ __forceinline__ void fill8c(float v, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = v;
}

// sum over the 16 lanes of a DPP row (= the 16 pixel rows of an MFMA fragment), result in every lane: 4 VALU DPP steps
// (quad swaps, then half-row and row mirrors) instead of 4 ds_bpermute round trips through the LDS
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// ---- BatchNorm statistic accumulators (fp64, sharded) ---------------------------------------------------------------------------
// Producers (conv epilogues, the BN-backward reduction, dgrad epilogues) fold their per-tile column sums straight into a per-layer
// accumulator acc[kAccShards][2][C] with fp64 atomics instead of writing one partial row per tile for a finalize launch to reduce;
// consumers (the BN+activation apply passes, the fused 1x1 backward) fold the shards in their prologue. fp64 makes the result
// independent of the arrival order to ~1e-16 relative (fp32 tile sums of similar magnitude add EXACTLY in fp64), i.e. the fp32
// statistics derived from it are run-to-run identical for all practical purposes. 16 shards keep the same-address chain short
// (profiles/r03_ceilings_probe.log: 6400 tiles x 128 adds cost 12 us spread over the producing kernel's lifetime).
constexpr int kAccShards = CVHIP_BN_ACC_SHARDS;
__device__ __forceinline__ void acc_add2(double* acc, int shard, int C, int c, float s1, float s2) {
  double* a = acc + (size_t)(shard & (kAccShards - 1)) * 2 * C;
  unsafeAtomicAdd(a + c, (double)s1);
  unsafeAtomicAdd(a + C + c, (double)s2);
}
__device__ __forceinline__ void acc_fold2(const double* acc, int C, int c, double& s1, double& s2) {
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int sh = 0; sh < kAccShards; ++sh) {
    a += acc[(size_t)sh * 2 * C + c];
    b += acc[(size_t)sh * 2 * C + C + c];
  }
  s1 = a;
  s2 = b;
}

// XCD-aware, bijective block-id remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective"):
// hardware places block b on XCD b%8; give each XCD a contiguous chunk of the logical tile space so
// neighbouring tiles (shared im2col halos / shared A panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int NX = 8;
  if (nblk < NX * 2) return bid;
  int xcd = bid % NX;
  int q = nblk / NX, r = nblk % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / NX;
}

// (cv, a, b, n) of a flat index i = ((n * Y + b) * X + a) * CV + cv. Flat indices below 2^31 (every tensor of the workloads: 16 x 128 x
// 256 x 64 vectors = 33.5 M) take the 32-bit path: three 64-bit divisions by run-time values cost ~250 instructions per thread
// iteration, which is what the light gather kernels (pooling, resize, generic depthwise, channel scaling) were spending their time on.
__device__ __forceinline__ void split_index(int64_t i, int CV, int X, int Y, int* cv, int* a, int* b, int* n) {
  if ((((uint64_t)i) >> 31) == 0) {
    uint32_t u = (uint32_t)i;
    uint32_t q = u / (uint32_t)CV;
    *cv = (int)(u - q * (uint32_t)CV);
    u = q;
    q = u / (uint32_t)X;
    *a = (int)(u - q * (uint32_t)X);
    u = q;
    q = u / (uint32_t)Y;
    *b = (int)(u - q * (uint32_t)Y);
    *n = (int)q;
  } else {
    int64_t pix = i / CV;
    *cv = (int)(i - pix * CV);
    int64_t r = pix / X;
    *a = (int)(pix - r * X);
    pix = r;
    r = pix / Y;
    *b = (int)(pix - r * Y);
    *n = (int)r;
  }
}

}  // namespace cvhip
