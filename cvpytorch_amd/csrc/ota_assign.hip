// ota_assign.hip — YOLOv7's OTA label assignment on the device: find_3_positive candidates of every level pooled per image,
// pairwise IoU + cost against every ground truth of the image, dynamic-k selection, conflict resolution.
//
// Reference: src/losses/yolov7_loss.py:217-365 (build_targets: per-image python loop, boolean-mask compaction, torch.topk and
// per-gt `.item()` loops), candidates from find_3_positive (:367-420, the YOLOv5 anchor-ratio + 0.5-cell offset rule).
//   candidates   : (offset o, anchor a, target t) per level, ordinal c = (o*A + a)*T + t — the ordering of yolo_loss.hip
//   per image    : pooled candidates of the L levels; pred box (sigmoid*2-0.5+grid)*stride, ((sigmoid*2)^2*anchor)*stride
//   per gt g     : IoU(g, c); dynamic_k = max(int(sum of the 20 largest IoU), 1)
//                  cost = [sum_k bce(z_k, 0) - z_cls(g)] + 3 * (-log(IoU + 1e-8)),  z = logit(sqrt(sigmoid(cls) * sigmoid(obj)))
//                  the dynamic_k cheapest candidates are matched to g
//   conflicts    : a candidate matched by several gts goes to the gt of minimum cost (over ALL gts of the image)
// Output: assign[l][c] = matched flat target row or -1. The loss itself is then the YOLOv5-form loss on that assignment
// (cvhip_yolov5_loss_level_fwd_assigned): CIoU / class BCE on the positives, objectness BCE with the IoU target.
//
// One wave per ground truth for the matching (top-k by iterative wave-wide extraction), one wave per candidate for the
// class-cost base. Index / comparison arithmetic follows the reference's fp32 order: FMA contraction off.
#pragma clang fp contract(off)
#include "common.h"
#include "dual4.h"

namespace cvhip {

constexpr int kOtaMaxL = 4;
constexpr int kOtaMaxE = 4096;  // pooled candidates per image (64 per lane: the taken-mask is one 64-bit word per lane)

struct OtaParams {
  const h16_t* raw[kOtaMaxL];
  int ld[kOtaMaxL], H[kOtaMaxL], W[kOtaMaxL];
  float stride[kOtaMaxL];
  float anchors[kOtaMaxL][16];
  int L, N, A, NO, nc, T, G, ncand, E;
  float anchor_t, img_size;
  const float* targets;
  int* first;    // [N] first flat row of the image
  int* count;    // [N] rows of the image
  float4* cbox;  // [L][ncand] predicted box, pixels, xyxy
  int* ccell;    // [L][ncand] element offset of the candidate's anchor block in raw[l], or -1 (not a find_3_positive candidate)
  float* cbase;  // [L][ncand] sum_k bce(z_k, 0)
  int* cnt;      // [L][ncand] gts that picked the candidate
  int* owner;    // [L][ncand] one of them
  int* assign;   // [L][ncand]
  int* overflow; // [1] an image had more than G targets (the extra ones get no candidates)
};

__device__ __forceinline__ float ota_z(float cls_logit, float obj_logit) {
  const float y = sqrtf(sigmoid_ref(cls_logit) * sigmoid_ref(obj_logit));
  return logf(y / (1.f - y));
}

// ---- image ranges of the flat target list (rows of one image are contiguous: collate order) --------------------------------------
__global__ void ota_ranges_kernel(const OtaParams p) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < p.T; t += gridDim.x * blockDim.x) {
    const float img = p.targets[(int64_t)t * 6];
    if (img >= 0.f && img < (float)p.N) {
      const int b = (int)img;
      atomicAdd(&p.count[b], 1);
      atomicMin(&p.first[b], t);
    }
  }
}

// ---- candidates: one wave per (level, ordinal) --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ota_cand_kernel(const OtaParams p) {
  const int lane = threadIdx.x & 63;
  const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gidx >= p.L * p.ncand) return;
  const int l = gidx / p.ncand, c = gidx - l * p.ncand;
  const int T = p.T, A = p.A;
  const int off = c / (A * T);
  const int a = (c / T) % A;
  const int t = c % T;
  const float* tg = p.targets + (int64_t)t * 6;
  const float img = tg[0];
  const float nx = (float)p.W[l], ny = (float)p.H[l];
  const float gx = tg[2] * nx, gy = tg[3] * ny, gw = tg[4] * nx, gh = tg[5] * ny;
  const float aw = p.anchors[l][a * 2], ah = p.anchors[l][a * 2 + 1];
  const float rw = gw / aw, rh = gh / ah;
  const float rmax = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
  bool sel = (img >= 0.f) && (img < (float)p.N) && (rmax < p.anchor_t);
  const float gxi = nx - gx, gyi = ny - gy;
  if (off == 1) sel = sel && ((gx - floorf(gx)) < 0.5f) && (gx > 1.f);
  else if (off == 2) sel = sel && ((gy - floorf(gy)) < 0.5f) && (gy > 1.f);
  else if (off == 3) sel = sel && ((gxi - floorf(gxi)) < 0.5f) && (gxi > 1.f);
  else if (off == 4) sel = sel && ((gyi - floorf(gyi)) < 0.5f) && (gyi > 1.f);
  if (sel) {  // targets beyond the per-image capacity have no candidates
    const int b = (int)img;
    if (t - p.first[b] >= p.G) sel = false;
  }
  const int64_t o = (int64_t)l * p.ncand + c;
  if (!sel) {
    if (lane == 0) {
      p.ccell[o] = -1;
      p.cnt[o] = 0;
    }
    return;
  }
  const float ox = (off == 1 ? 0.5f : (off == 3 ? -0.5f : 0.f)), oy = (off == 2 ? 0.5f : (off == 4 ? -0.5f : 0.f));
  int gi = (int)(gx - ox), gj = (int)(gy - oy);
  gi = min(max(gi, 0), p.W[l] - 1);
  gj = min(max(gj, 0), p.H[l] - 1);
  const int b = (int)img;
  const int64_t eo = ((int64_t)(b * p.H[l] + gj) * p.W[l] + gi) * p.ld[l] + a * p.NO;
  const h16_t* px = p.raw[l] + eo;
  const float s0 = sigmoid_ref((float)px[0]), s1 = sigmoid_ref((float)px[1]), s2 = sigmoid_ref((float)px[2]), s3 = sigmoid_ref((float)px[3]);
  const float st = p.stride[l];
  const float cx = (s0 * 2.f - 0.5f + (float)gi) * st, cy = (s1 * 2.f - 0.5f + (float)gj) * st;
  const float w = (s2 * 2.f) * (s2 * 2.f) * aw * st, h = (s3 * 2.f) * (s3 * 2.f) * ah * st;
  const float obj = (float)px[4];
  float lsum = 0.f;
  for (int k = lane; k < p.nc; k += 64) lsum += bce_logits(ota_z((float)px[5 + k], obj), 0.f);
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) lsum += __shfl_xor(lsum, s, 64);
  if (lane == 0) {
    p.cbox[o] = make_float4(cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2);
    p.ccell[o] = (int)eo;
    p.cbase[o] = lsum;
    p.cnt[o] = 0;
  }
}

// IoU (yolov7_loss.py box_iou: inter / (area1 + area2 - inter)) and cost of (gt row t, pooled candidate (l, c))
__device__ __forceinline__ void ota_pair(const OtaParams& p, const float* tg, int l, int c, float* iou_out, float* cost_out) {
  const int64_t o = (int64_t)l * p.ncand + c;
  const int cell = p.ccell[o];
  if (cell < 0) {
    *iou_out = 0.f;
    *cost_out = __builtin_inff();
    return;
  }
  const float sz = p.img_size;
  const float tx = tg[2] * sz, ty = tg[3] * sz, tw = tg[4] * sz, th = tg[5] * sz;
  const float x1 = tx - tw / 2, y1 = ty - th / 2, x2 = tx + tw / 2, y2 = ty + th / 2;
  const float4 b = p.cbox[o];
  const float a1 = (x2 - x1) * (y2 - y1), a2 = (b.z - b.x) * (b.w - b.y);
  const float iw = fmaxf(fminf(x2, b.z) - fmaxf(x1, b.x), 0.f), ih = fmaxf(fminf(y2, b.w) - fmaxf(y1, b.y), 0.f);
  const float inter = iw * ih;
  const float iou = inter / (a1 + a2 - inter);
  int cls = (int)tg[1];
  cls = min(max(cls, 0), p.nc - 1);
  const h16_t* px = p.raw[l] + cell;
  const float z = ota_z((float)px[5 + cls], (float)px[4]);
  *iou_out = iou;
  *cost_out = (p.cbase[o] + (-z)) + 3.0f * (-logf(iou + 1e-8f));
}

// pooled-candidate enumeration of image b: e -> (level, ordinal)
__device__ __forceinline__ void ota_decode(const OtaParams& p, int e, int first, int cntb, int* l, int* c) {
  const int j = e % cntb;
  int r = e / cntb;
  const int a = r % p.A;
  r /= p.A;
  const int off = r % 5;
  *l = r / 5;
  *c = (off * p.A + a) * p.T + first + j;
}

// ---- matching: one wave (= one block) per ground truth; the IoU / cost rows of the image's pooled candidates live in LDS -----------------
__global__ __launch_bounds__(64) void ota_match_kernel(const OtaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ota_smem[];
  const int lane = threadIdx.x;
  const int t = blockIdx.x;
  if (t >= p.T) return;
  const float* tg = p.targets + (int64_t)t * 6;
  const float img = tg[0];
  if (!(img >= 0.f && img < (float)p.N)) return;
  const int b = (int)img;
  const int first = p.first[b];
  if (t - first >= p.G) return;  // beyond the per-image capacity (flagged)
  const int cntb = min(p.count[b], p.G);
  const int Eb = p.L * 5 * p.A * cntb;
  float* iou = reinterpret_cast<float*>(ota_smem);
  float* cost = iou + p.E;
  for (int e = lane; e < Eb; e += 64) {
    int l, c;
    ota_decode(p, e, first, cntb, &l, &c);
    float v, w;
    ota_pair(p, tg, l, c, &v, &w);
    iou[e] = v;
    cost[e] = w;
  }
  __syncthreads();
  // ---- dynamic_k = max(int(sum of the 20 largest IoU), 1) ----
  unsigned long long taken = 0ull;
  float ksum = 0.f;
  const int kk = Eb < 20 ? Eb : 20;
  for (int r = 0; r < kk; ++r) {
    float bv = -1.f;
    int be = 0x7fffffff;
    for (int e = lane, s = 0; e < Eb; e += 64, ++s) {
      if ((taken >> s) & 1ull) continue;
      const float v = iou[e];
      if (v > bv) {  // first (lowest e) maximum of the lane
        bv = v;
        be = e;
      }
    }
    float wv = bv;
    int we = be;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const float ov = __shfl_xor(wv, s, 64);
      const int oe = __shfl_xor(we, s, 64);
      if (ov > wv || (ov == wv && oe < we)) {
        wv = ov;
        we = oe;
      }
    }
    if (we == 0x7fffffff) break;
    ksum += wv;
    if ((we & 63) == lane) taken |= 1ull << (we >> 6);
  }
  int dyn_k = (int)ksum;
  if (dyn_k < 1) dyn_k = 1;
  if (dyn_k > kk) dyn_k = kk;
  // ---- the dyn_k cheapest usable candidates ----
  taken = 0ull;
  for (int r = 0; r < dyn_k; ++r) {
    float bv = __builtin_inff();
    int be = 0x7fffffff;
    for (int e = lane, s = 0; e < Eb; e += 64, ++s) {
      if ((taken >> s) & 1ull) continue;
      const float v = cost[e];
      if (v < bv) {
        bv = v;
        be = e;
      }
    }
    float wv = bv;
    int we = be;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const float ov = __shfl_xor(wv, s, 64);
      const int oe = __shfl_xor(we, s, 64);
      if (ov < wv || (ov == wv && oe < we)) {
        wv = ov;
        we = oe;
      }
    }
    if (we == 0x7fffffff || !(wv < __builtin_inff())) break;  // only finite costs are matched
    if ((we & 63) == lane) {
      taken |= 1ull << (we >> 6);
      int l, c;
      ota_decode(p, we, first, cntb, &l, &c);
      const int64_t o = (int64_t)l * p.ncand + c;
      atomicAdd(&p.cnt[o], 1);
      p.owner[o] = t;
    }
  }
}

// ---- conflicts: a candidate picked by several gts goes to the cheapest gt of the image --------------------------------------------------
__global__ __launch_bounds__(256) void ota_resolve_kernel(const OtaParams p) {
  const int64_t total = (int64_t)p.L * p.ncand;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int n = p.ccell[o] >= 0 ? p.cnt[o] : 0;
    int res = -1;
    if (n == 1) {
      res = p.owner[o];
    } else if (n > 1) {
      const int l = (int)(o / p.ncand), c = (int)(o - (int64_t)l * p.ncand);
      const int t0 = c % p.T;
      const int b = (int)p.targets[(int64_t)t0 * 6];
      const int first = p.first[b];
      const int cntb = min(p.count[b], p.G);
      float best = __builtin_inff();
      for (int j = 0; j < cntb; ++j) {  // torch.argmin: the first minimum
        float v, w;
        ota_pair(p, p.targets + (int64_t)(first + j) * 6, l, c, &v, &w);
        if (w < best) {
          best = w;
          res = first + j;
        }
      }
      if (res < 0) res = p.owner[o];
    }
    p.assign[o] = res;
  }
}

__global__ void ota_flag_kernel(const OtaParams p) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < p.N; b += gridDim.x * blockDim.x)
    if (p.count[b] > p.G) p.overflow[0] = 1;
}

static int64_t ota_align(int64_t b) { return (b + 255) / 256 * 256; }

static int ota_fill(OtaParams& p, const cvhip_ota_desc* d, const void* const* raws, const float* targets, void* ws, int32_t* assign) {
  if (!d || !raws || !targets || !ws || !assign) return CVHIP_ERR_INVALID;
  if (d->L <= 0 || d->L > kOtaMaxL || d->N <= 0 || d->A <= 0 || d->A > 8 || d->NO < 6 || d->T <= 0 || d->G <= 0) return CVHIP_ERR_INVALID;
  const int64_t ncand = (int64_t)5 * d->A * d->T;
  const int64_t E = (int64_t)d->L * 5 * d->A * d->G;
  if (E > kOtaMaxE || ncand >= (1ll << 28)) return CVHIP_ERR_UNSUPPORTED;
  p.L = d->L;
  p.N = d->N;
  p.A = d->A;
  p.NO = d->NO;
  p.nc = d->NO - 5;
  p.T = d->T;
  p.G = d->G;
  p.ncand = (int)ncand;
  p.E = (int)E;
  p.anchor_t = d->anchor_t;
  p.img_size = d->img_size;
  p.targets = targets;
  for (int l = 0; l < d->L; ++l) {
    if (!raws[l] || d->H[l] <= 0 || d->W[l] <= 0 || d->ld[l] < d->A * d->NO) return CVHIP_ERR_INVALID;
    if ((int64_t)d->N * d->H[l] * d->W[l] * d->ld[l] >= (1ll << 31)) return CVHIP_ERR_UNSUPPORTED;
    p.raw[l] = (const h16_t*)raws[l];
    p.ld[l] = d->ld[l];
    p.H[l] = d->H[l];
    p.W[l] = d->W[l];
    p.stride[l] = d->stride[l];
    for (int i = 0; i < d->A * 2; ++i) p.anchors[l][i] = d->anchors[l][i];
  }
  unsigned char* w = (unsigned char*)ws;
  auto take = [&](int64_t bytes) {
    unsigned char* r = w;
    w += ota_align(bytes);
    return r;
  };
  p.first = (int*)take((int64_t)d->N * 4);
  p.count = (int*)take((int64_t)d->N * 4);
  p.overflow = (int*)take(256);
  p.cbox = (float4*)take(d->L * ncand * 16);
  p.ccell = (int*)take(d->L * ncand * 4);
  p.cbase = (float*)take(d->L * ncand * 4);
  p.cnt = (int*)take(d->L * ncand * 4);
  p.owner = (int*)take(d->L * ncand * 4);
  p.assign = assign;
  return CVHIP_OK;
}

__global__ void ota_init_kernel(const OtaParams p) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < p.N; b += gridDim.x * blockDim.x) {
    p.first[b] = 0x7fffffff;
    p.count[b] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.overflow[0] = 0;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int64_t cvhip_ota_workspace_bytes(const cvhip_ota_desc* d) {
  if (!d || d->L <= 0 || d->L > kOtaMaxL || d->N <= 0 || d->A <= 0 || d->T <= 0 || d->G <= 0) return CVHIP_ERR_INVALID;
  const int64_t ncand = (int64_t)5 * d->A * d->T;
  const int64_t E = (int64_t)d->L * 5 * d->A * d->G;
  if (E > kOtaMaxE) return CVHIP_ERR_UNSUPPORTED;
  return ota_align((int64_t)d->N * 4) * 2 + 256 + ota_align(d->L * ncand * 16) + ota_align(d->L * ncand * 4) * 4;
}

int cvhip_ota_assign(const cvhip_ota_desc* d, const void* const* raws, const float* targets, void* ws, int32_t* assign, void* stream) {
  OtaParams p;
  int rc = ota_fill(p, d, raws, targets, ws, assign);
  if (rc != CVHIP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ota_init_kernel, dim3(cdiv(p.N, 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(ota_ranges_kernel, dim3(cdiv(p.T, 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(ota_flag_kernel, dim3(cdiv(p.N, 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(ota_cand_kernel, dim3(cdiv(p.L * p.ncand, 4)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(ota_match_kernel, dim3(p.T), dim3(64), (size_t)p.E * 8, st, p);
  const int64_t total = (int64_t)p.L * p.ncand;
  hipLaunchKernelGGL(ota_resolve_kernel, dim3((int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096)), dim3(256), 0, st, p);
  return check_launch("ota_assign");
}

/* diagnostics for the tests: overflow flag of the last cvhip_ota_assign on this workspace */
int cvhip_ota_read_overflow(const cvhip_ota_desc* d, const void* ws, int32_t* out, void* stream) {
  if (!d || !ws || !out) return CVHIP_ERR_INVALID;
  const unsigned char* w = (const unsigned char*)ws + ota_align((int64_t)d->N * 4) * 2;
  hipError_t e = hipMemcpyAsync(out, w, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    set_last_error("cvhip_ota_read_overflow", e);
    return CVHIP_ERR_LAUNCH;
  }
  return CVHIP_OK;
}

}  // extern "C"
