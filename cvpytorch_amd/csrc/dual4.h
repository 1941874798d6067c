// dual4.h — forward-mode dual numbers over 4 inputs (value + 4 partial derivatives), used by the detection-loss kernels to
// get exact box-loss gradients (CIoU, IoU^2) without hand-derived formulas.
#pragma once
#include "common.h"

namespace cvhip {

struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 cst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 var(float v, int i) {
  D4 r = cst(v);
  r.d[i] = 1.f;
  return r;
}
__device__ __forceinline__ D4 operator+(D4 a, D4 b) {
  D4 r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator-(D4 a, D4 b) {
  D4 r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
  D4 r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  D4 r;
  r.v = a.v / b.v;
  const float inv = 1.f / b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ D4 scale(D4 a, float s) {
  D4 r;
  r.v = a.v * s;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 clamp0(D4 a) { return a.v >= 0.f ? a : cst(0.f); }
__device__ __forceinline__ D4 datan(D4 a) {
  D4 r;
  r.v = atanf(a.v);
  const float g = 1.f / (1.f + a.v * a.v);
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g;
  return r;
}


__device__ __forceinline__ float sigmoid_ref(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) {
  // aten::binary_cross_entropy_with_logits: (1 - t) * x + max(-x, 0) + log(exp(-max) + exp(-x - max))
  const float m = fmaxf(-x, 0.f);
  return (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
}

}  // namespace cvhip
