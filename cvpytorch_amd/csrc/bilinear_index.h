// bilinear_index.h — ATen upsample_bilinear2d index rule, shared by the resize kernels (pool_resize.hip) and the fused
// resize + cross-entropy kernels (loss_kernels.hip).
#pragma once
#include "common.h"

namespace cvhip {

__device__ __forceinline__ void bil_src(int o, float scale, int align, int in, int* i0, int* i1, float* l1) {
  float s;
  if (align) s = scale * o;
  else {
    s = scale * (o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
  }
  int a = (int)s;
  if (a > in - 1) a = in - 1;
  const int b = a + ((a < in - 1) ? 1 : 0);
  *i0 = a;
  *i1 = b;
  *l1 = s - (float)a;
}

// backward as a deterministic gather: each INPUT pixel scans the (small) range of output pixels that
// can reference it and re-derives their interpolation weights.
__device__ __forceinline__ void bil_range(int i, float scale, int align, int out, int* lo, int* hi) {
  // outputs o with floor(src(o)) in {i-1, i}; src is monotone in o. Conservative bounds, then exact test.
  const float inv = scale > 0.f ? 1.f / scale : 0.f;
  float a, b;
  if (align) {
    a = (i - 1) * inv;
    b = (i + 1) * inv;
  } else {
    a = (i - 1 + 0.5f) * inv - 0.5f;
    b = (i + 1 + 0.5f) * inv - 0.5f;
  }
  int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
  if (scale <= 0.f) {
    l = 0;
    h = out - 1;
  }
  if (l < 0) l = 0;
  if (h > out - 1) h = out - 1;
  *lo = l;
  *hi = h;
}

}  // namespace cvhip

static inline void bil_scales(int Hi, int Wi, int Ho, int Wo, int align, float* sh, float* sw) {
  if (align) {
    *sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    *sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  } else {
    *sh = (float)Hi / (float)Ho;
    *sw = (float)Wi / (float)Wo;
  }
}

