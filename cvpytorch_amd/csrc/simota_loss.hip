// simota_loss.hip — YOLOX loss on device: SimOTA label assignment + 5*IoU^2 + objectness + class BCE and their gradients,
// reading the bf16 NHWC head maps (channels [reg4, obj1, cls nc]) directly and writing bf16 gradients directly.
//
// Reference: src/losses/det/yolox_loss.py: get_output_and_grid :138-153, get_in_boxes_info :350-403 (centre radius 2.5),
// bboxes_iou :14-31, cost = cls + 3*(-log IoU) + 1e5*!in_both :335 (cls term = BCE(sqrt(sigmoid(cls)*sigmoid(obj)), one-hot)),
// dynamic_k_matching :405-435 (sum of the 10 largest IoUs -> k, the k smallest costs per gt, conflicts -> argmin over gts),
// IOUloss :34-69 (1 - IoU^2), loss = 5*iou + obj + cls, all / num_fg :279-280.
// The per-image python loop / per-gt `.item()` top-k loop become:   prep (per anchor) -> match (one wave per (image, gt):
// register top-10 lists + wave-shuffle extraction) -> resolve (per anchor) -> loss / gradient (per anchor).
// Cost arithmetic follows the dense torch formulation in cvpytorch_amd/yolox.py (itself identical to the reference's
// assignments on its golden vectors): class cost = per-anchor base (all-negative labels) + correction of the gt's class.
#pragma clang fp contract(off)
#include <string.h>
#include "common.h"
#include "dual4.h"

namespace cvhip {

constexpr int kSimMaxLevels = 4;

struct SimotaParams {
  const h16_t* raw[kSimMaxLevels];
  h16_t* draw[kSimMaxLevels];
  int ld[kSimMaxLevels], H[kSimMaxLevels], W[kSimMaxLevels], off[kSimMaxLevels + 1];
  float stride[kSimMaxLevels];
  int L, B, A, G, nc;
  const float* targets;  // (B, G, 5) [cls, cx, cy, w, h] pixels; all-zero rows = padding
  // workspace
  float* boxes;     // [B*A][4] decoded cx, cy, w, h
  float* base;      // [B*A]    sum_c -log(1 - p_c)
  float* sobj;      // [B*A]    sigmoid(obj)
  unsigned char* cand;  // [B*A]
  int* nlabel;      // [B]
  int* cnt;         // [B*A]    number of gts that selected the anchor
  int* selgt;       // [B*A]    a gt that selected it
  int* matched;     // [B*A]    final gt or -1
  float* miou;      // [B*A]
  int* numfg;       // [1]
  float* partial;   // [1024][3]
  float* sums;      // [8]: iou_loss_sum, obj_sum, cls_sum, num_fg, num_gts
  const float* gout;
};

__device__ __forceinline__ void sim_locate(const SimotaParams& p, int a, int* l, int* y, int* x) {
  int lv = 0;
#pragma unroll
  for (int i = 1; i < kSimMaxLevels; ++i)
    if (i < p.L && a >= p.off[i]) lv = i;
  const int r = a - p.off[lv];
  *l = lv;
  *y = r / p.W[lv];
  *x = r - (*y) * p.W[lv];
}

__device__ __forceinline__ const h16_t* sim_ptr(const SimotaParams& p, int b, int l, int y, int x) {
  return p.raw[l] + ((int64_t)(b * p.H[l] + y) * p.W[l] + x) * p.ld[l];
}

__device__ __forceinline__ float clog(float v) { return fmaxf(logf(v), -100.f); }

__device__ __forceinline__ float pair_iou(float gx, float gy, float gw, float gh, float px, float py, float pw, float ph) {
  const float tlx = fmaxf(gx - gw / 2, px - pw / 2), tly = fmaxf(gy - gh / 2, py - ph / 2);
  const float brx = fminf(gx + gw / 2, px + pw / 2), bry = fminf(gy + gh / 2, py + ph / 2);
  const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
  const float ai = (brx - tlx) * (bry - tly) * en;
  return ai / (gw * gh + pw * ph - ai);
}

__device__ __forceinline__ void in_flags(float xc, float yc, float s, float gx, float gy, float gw, float gh, bool* in_box, bool* in_ctr) {
  const float l = gx - 0.5f * gw, r = gx + 0.5f * gw, t = gy - 0.5f * gh, b = gy + 0.5f * gh;
  *in_box = fminf(fminf(xc - l, yc - t), fminf(r - xc, b - yc)) > 0.f;
  const float rad = 2.5f * s;
  *in_ctr = fminf(fminf(xc - (gx - rad), yc - (gy - rad)), fminf((gx + rad) - xc, (gy + rad) - yc)) > 0.f;
}

// ---- K0: labels per image -------------------------------------------------------------------------------------------
__global__ void sim_nlabel_kernel(const SimotaParams p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  int n = 0;
  for (int g = 0; g < p.G; ++g) {
    const float* t = p.targets + ((int64_t)b * p.G + g) * 5;
    if (((((t[0] + t[1]) + t[2]) + t[3]) + t[4]) > 0.f) ++n;
  }
  p.nlabel[b] = n;
  if (b == 0) p.numfg[0] = 0;
}

// ---- K1: per anchor: decode, class-cost base, candidate flag ------------------------------------------------------------
// Round 6: the 5 + nc logits of the block's 256 anchors are staged through the LDS with coalesced loads (lanes walk a row's consecutive
// halfwords); before, every thread walked its own row with 2-byte loads at a 176-byte lane stride inside the 80-class loop — 612 us for
// YOLOX-s at batch 64, ten times the arithmetic. Same formulas, same order of the class sum.
constexpr int kSimPrepMaxNo = 96;   // rows of up to 96 logits are staged (48 KB); wider heads read global memory as before
__global__ __launch_bounds__(256) void sim_prep_kernel(const SimotaParams p) {
  __shared__ h16_t srow[256 * kSimPrepMaxNo];
  __shared__ const h16_t* sptr[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int NO = 5 + p.nc;
  const bool staged = NO <= kSimPrepMaxNo;
  const bool live = i < p.B * p.A;
  int b = 0, a = 0, l = 0, y = 0, x = 0;
  const h16_t* px = p.raw[0];
  if (live) {
    b = i / p.A;
    a = i - b * p.A;
    sim_locate(p, a, &l, &y, &x);
    px = sim_ptr(p, b, l, y, x);
  }
  if (staged) {
    sptr[threadIdx.x] = px;
    __syncthreads();
    const unsigned inv = (unsigned)(((1ull << 32) + (unsigned)NO - 1) / (unsigned)NO);   // e / NO by multiply-high (e < 256 * 96)
    for (int e = threadIdx.x; e < 256 * NO; e += 256) {
      const int r = (int)__umulhi((unsigned)e, inv);
      srow[e] = sptr[r][e - r * NO];
    }
    __syncthreads();
  }
  if (!live) return;
  const h16_t* const lrow = srow + threadIdx.x * NO;
  auto lg = [&](int c) -> float { return staged ? (float)lrow[c] : (float)px[c]; };   // (block-uniform: ds_read or global_load, never flat)
  const float s = p.stride[l];
  const float bx = (lg(0) + (float)x) * s, by = (lg(1) + (float)y) * s;
  const float bw = expf(lg(2)) * s, bh = expf(lg(3)) * s;
  float* bo = p.boxes + (int64_t)i * 4;
  bo[0] = bx;
  bo[1] = by;
  bo[2] = bw;
  bo[3] = bh;
  const float so = sigmoid_ref(lg(4));
  float base = 0.f;
  for (int c = 0; c < p.nc; ++c) {
    const float pc = sqrtf(sigmoid_ref(lg(5 + c)) * so);
    base += -clog(1.f - pc);
  }
  p.base[i] = base;
  p.sobj[i] = so;
  const float xc = (float)x * s + 0.5f * s, yc = (float)y * s + 0.5f * s;
  bool cand = false;
  const int ng = p.nlabel[b];
  for (int g = 0; g < ng; ++g) {
    const float* t = p.targets + ((int64_t)b * p.G + g) * 5;
    bool ib, ic;
    in_flags(xc, yc, s, t[1], t[2], t[3], t[4], &ib, &ic);
    cand = cand || ib || ic;
  }
  p.cand[i] = cand ? 1 : 0;
  p.cnt[i] = 0;
  p.selgt[i] = -1;
}

// cost of (gt g, anchor a) — shared by match and resolve so both see bit-identical values
__device__ __forceinline__ float sim_cost(const SimotaParams& p, int b, int a, const float* t, int gcls, float* iou_out) {
  int l, y, x;
  sim_locate(p, a, &l, &y, &x);
  const float s = p.stride[l];
  const int64_t i = (int64_t)b * p.A + a;
  const float* bo = p.boxes + i * 4;
  const float iou = pair_iou(t[1], t[2], t[3], t[4], bo[0], bo[1], bo[2], bo[3]);
  *iou_out = iou;
  const float xc = (float)x * s + 0.5f * s, yc = (float)y * s + 0.5f * s;
  bool ib, ic;
  in_flags(xc, yc, s, t[1], t[2], t[3], t[4], &ib, &ic);
  const h16_t* px = sim_ptr(p, b, l, y, x);
  const float pc = sqrtf(sigmoid_ref((float)px[5 + gcls]) * p.sobj[i]);
  const float cls_cost = p.base[i] + (clog(1.f - pc) - clog(pc));
  return (cls_cost + 3.0f * (-logf(iou + 1e-8f))) + 100000.0f * ((ib && ic) ? 0.f : 1.f);
}

// ---- K2: one 4-wave block per (image, gt): dynamic k and the k cheapest candidates ------------------------------------------------
// (round 6: one WAVE per pair scanned the image's 8400 anchors 131 to a lane, 1280 waves for the whole chip: 259 us. Four waves share
// the scan; the arg-max / arg-min rounds go wave shuffle -> four LDS slots -> every thread picks the block's winner with the same
// total order — (value, thread) for the IoU list, (cost, anchor) for the cost list — so the selection does not depend on the split.)
constexpr int kSimMatchThreads = 512;
__global__ __launch_bounds__(kSimMatchThreads) void sim_match_kernel(const SimotaParams p) {
  __shared__ float sbest[kSimMatchThreads / 64];
  __shared__ int swho[kSimMatchThreads / 64];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (g >= p.nlabel[b]) return;
  const float* t = p.targets + ((int64_t)b * p.G + g) * 5;
  int gcls = (int)t[0];
  gcls = min(max(gcls, 0), p.nc - 1);
  float ti[10], tc[10];
  int ta[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    ti[k] = -1.f;          // IoUs are >= 0
    tc[k] = INFINITY;
    ta[k] = 0x7fffffff;
  }
  for (int a = tid; a < p.A; a += kSimMatchThreads) {
    if (!p.cand[(int64_t)b * p.A + a]) continue;
    float iou;
    const float cost = sim_cost(p, b, a, t, gcls, &iou);
    if (iou > ti[9]) {  // insert into the descending IoU list
      float v = iou;
#pragma unroll
      for (int k = 0; k < 10; ++k)
        if (v > ti[k]) {
          const float o = ti[k];
          ti[k] = v;
          v = o;
        }
    }
    if (cost < tc[9] || (cost == tc[9] && a < ta[9])) {  // insert into the ascending (cost, anchor) list
      float v = cost;
      int va = a;
#pragma unroll
      for (int k = 0; k < 10; ++k)
        if (v < tc[k] || (v == tc[k] && va < ta[k])) {
          const float o = tc[k];
          const int oa = ta[k];
          tc[k] = v;
          ta[k] = va;
          v = o;
          va = oa;
        }
    }
  }
  // dynamic k = clamp(int(sum of the 10 largest IoUs), 1): 10 rounds of wave arg-max over the lanes' list heads
  float ksum = 0.f;
  for (int r = 0; r < 10; ++r) {
    float best = ti[0];
    int who = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int ow = __shfl_xor(who, o, 64);
      if (ob > best || (ob == best && ow < who)) {
        best = ob;
        who = ow;
      }
    }
    if (lane == 0) {
      sbest[wave] = best;
      swho[wave] = who;
    }
    __syncthreads();
    best = sbest[0];
    who = swho[0];
#pragma unroll
    for (int w = 1; w < kSimMatchThreads / 64; ++w)
      if (sbest[w] > best || (sbest[w] == best && swho[w] < who)) {
        best = sbest[w];
        who = swho[w];
      }
    __syncthreads();
    if (best < 0.f) break;  // fewer than 10 candidates
    ksum += best;
    if (tid == who) {
#pragma unroll
      for (int k = 0; k < 9; ++k) ti[k] = ti[k + 1];
      ti[9] = -1.f;
    }
  }
  int dynk = (int)ksum;
  if (dynk < 1) dynk = 1;
  for (int r = 0; r < dynk; ++r) {
    float best = tc[0];
    int ba = ta[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(ba, o, 64);
      if (ob < best || (ob == best && oa < ba)) {
        best = ob;
        ba = oa;
      }
    }
    if (lane == 0) {
      sbest[wave] = best;
      swho[wave] = ba;
    }
    __syncthreads();
    best = sbest[0];
    ba = swho[0];
#pragma unroll
    for (int w = 1; w < kSimMatchThreads / 64; ++w)
      if (sbest[w] < best || (sbest[w] == best && swho[w] < ba)) {
        best = sbest[w];
        ba = swho[w];
      }
    __syncthreads();
    if (!(best < INFINITY)) break;
    if (ta[0] == ba) {  // the owning thread consumes its head and publishes the match
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        tc[k] = tc[k + 1];
        ta[k] = ta[k + 1];
      }
      tc[9] = INFINITY;
      ta[9] = 0x7fffffff;
      const int64_t i = (int64_t)b * p.A + ba;
      atomicAdd(p.cnt + i, 1);
      p.selgt[i] = g;
    }
  }
}

// ---- K3: per anchor: final gt (conflicts -> arg-min cost over all gts), matched IoU, foreground count ---------------------------
__global__ __launch_bounds__(256) void sim_resolve_kernel(const SimotaParams p) {
  __shared__ int red[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  int fg = 0;
  if (i < p.B * p.A) {
    const int b = i / p.A, a = i - b * p.A;
    const int c = p.cnt[i];
    int m = -1;
    float miou = 0.f;
    if (c == 1) {
      m = p.selgt[i];
    } else if (c > 1) {
      float bestc = INFINITY;
      const int ng = p.nlabel[b];
      for (int g = 0; g < ng; ++g) {
        const float* t = p.targets + ((int64_t)b * p.G + g) * 5;
        int gcls = (int)t[0];
        gcls = min(max(gcls, 0), p.nc - 1);
        float iou;
        const float cost = sim_cost(p, b, a, t, gcls, &iou);
        if (cost < bestc) {
          bestc = cost;
          m = g;
        }
      }
    }
    if (m >= 0) {
      const float* t = p.targets + ((int64_t)b * p.G + m) * 5;
      const float* bo = p.boxes + (int64_t)i * 4;
      miou = pair_iou(t[1], t[2], t[3], t[4], bo[0], bo[1], bo[2], bo[3]);
      fg = 1;
    }
    p.matched[i] = m;
    p.miou[i] = miou;
  }
  red[threadIdx.x] = fg;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0]) atomicAdd(p.numfg, red[0]);
}

// IoU of the IOUloss (:34-69): pred box differentiated
__device__ __forceinline__ D4 iou_loss_iou(float px, float py, float pw, float ph, float tx, float ty, float tw, float th) {
  const D4 X = var(px, 0), Y = var(py, 1), Wd = var(pw, 2), Hd = var(ph, 3);
  const D4 tlx = dmax(X - scale(Wd, 0.5f), cst(tx - tw / 2)), tly = dmax(Y - scale(Hd, 0.5f), cst(ty - th / 2));
  const D4 brx = dmin(X + scale(Wd, 0.5f), cst(tx + tw / 2)), bry = dmin(Y + scale(Hd, 0.5f), cst(ty + th / 2));
  const D4 area_p = Wd * Hd;
  const float area_g = tw * th;
  const float en = (tlx.v < brx.v && tly.v < bry.v) ? 1.f : 0.f;
  const D4 area_i = scale((brx - tlx) * (bry - tly), en);
  return area_i / (area_p + cst(area_g) - area_i + cst(1e-16f));
}

// ---- K4: losses (BWD = false: per-block partial sums) or gradients (BWD = true) -------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void sim_loss_kernel(const SimotaParams p) {
  __shared__ float red[3][256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f;
  if (i < p.B * p.A) {
    const int b = i / p.A, a = i - b * p.A;
    int l, y, x;
    sim_locate(p, a, &l, &y, &x);
    const h16_t* px = sim_ptr(p, b, l, y, x);
    const int m = p.matched[i];
    const float fgt = m >= 0 ? 1.f : 0.f;
    const float nf = fmaxf((float)p.numfg[0], 1.f);
    const float go = BWD ? (p.gout ? p.gout[0] : 1.f) / nf : 0.f;
    h16_t* dx = BWD ? p.draw[l] + ((int64_t)(b * p.H[l] + y) * p.W[l] + x) * p.ld[l] : nullptr;
    const float xo = (float)px[4];
    if (BWD) dx[4] = (h16_t)((sigmoid_ref(xo) - fgt) * go);
    else l_obj = bce_logits(xo, fgt);
    if (m >= 0) {
      const float* t = p.targets + ((int64_t)b * p.G + m) * 5;
      const float* bo = p.boxes + (int64_t)i * 4;
      const D4 iou = iou_loss_iou(bo[0], bo[1], bo[2], bo[3], t[1], t[2], t[3], t[4]);
      int tcls = (int)t[0];
      tcls = min(max(tcls, 0), p.nc - 1);
      const float mi = p.miou[i];
      if (BWD) {
        const float s = p.stride[l];
        const float k = -2.f * iou.v * 5.0f * go;  // d(5 * (1 - iou^2))
        dx[0] = (h16_t)(k * iou.d[0] * s);
        dx[1] = (h16_t)(k * iou.d[1] * s);
        dx[2] = (h16_t)(k * iou.d[2] * bo[2]);
        dx[3] = (h16_t)(k * iou.d[3] * bo[3]);
        for (int c = 0; c < p.nc; ++c) dx[5 + c] = (h16_t)((sigmoid_ref((float)px[5 + c]) - (c == tcls ? mi : 0.f)) * go);
      } else {
        l_iou = 1.f - iou.v * iou.v;
        for (int c = 0; c < p.nc; ++c) l_cls += bce_logits((float)px[5 + c], c == tcls ? mi : 0.f);
      }
    }
  }
  if (!BWD) {
    red[0][threadIdx.x] = l_iou;
    red[1][threadIdx.x] = l_obj;
    red[2][threadIdx.x] = l_cls;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        red[0][threadIdx.x] += red[0][threadIdx.x + s];
        red[1][threadIdx.x] += red[1][threadIdx.x + s];
        red[2][threadIdx.x] += red[2][threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      p.partial[blockIdx.x * 3 + 0] = red[0][0];
      p.partial[blockIdx.x * 3 + 1] = red[1][0];
      p.partial[blockIdx.x * 3 + 2] = red[2][0];
    }
  }
}

// ---- K5: final scalars: out5 = {loss, conf_loss, cls_loss, 5*iou_loss, num_fg / num_gts} ------------------------------------
__global__ __launch_bounds__(1024) void sim_finalize_kernel(const SimotaParams p, int nblocks, float* out5) {
  __shared__ float red[1024];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nblocks; i += 1024) {
    acc[0] += p.partial[i * 3 + 0];
    acc[1] += p.partial[i * 3 + 1];
    acc[2] += p.partial[i * 3 + 2];
  }
  float tot[3];
  for (int k = 0; k < 3; ++k) {
    red[threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    tot[k] = red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nf = fmaxf((float)p.numfg[0], 1.f);
    int ngts = 0;
    for (int b = 0; b < p.B; ++b) ngts += p.nlabel[b];
    const float li = tot[0] / nf, lo = tot[1] / nf, lc = tot[2] / nf;
    out5[0] = (5.0f * li + lo) + lc;
    out5[1] = lo;
    out5[2] = lc;
    out5[3] = 5.0f * li;
    out5[4] = nf / fmaxf((float)ngts, 1.f);
  }
}

__global__ void sim_copy_assign_kernel(const int* m, const float* u, int* mo, float* uo, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    mo[i] = m[i];
    uo[i] = u[i];
  }
}

static int64_t sim_ws_layout(const cvhip_simota_desc* d, SimotaParams* p, void* ws) {
  const int64_t BA = (int64_t)d->B * d->A;
  unsigned char* w = (unsigned char*)ws;
  int64_t used = 0;
  auto take = [&](int64_t bytes) {
    unsigned char* r = w ? w + used : nullptr;
    used += (bytes + 255) / 256 * 256;
    return r;
  };
  float* boxes = (float*)take(BA * 16);
  float* base = (float*)take(BA * 4);
  float* sobj = (float*)take(BA * 4);
  unsigned char* cand = (unsigned char*)take(BA);
  int* nlabel = (int*)take((int64_t)d->B * 4);
  int* cnt = (int*)take(BA * 4);
  int* selgt = (int*)take(BA * 4);
  int* matched = (int*)take(BA * 4);
  float* miou = (float*)take(BA * 4);
  int* numfg = (int*)take(256);
  float* partial = (float*)take(((BA + 255) / 256) * 3 * 4);
  if (p) {
    p->boxes = boxes; p->base = base; p->sobj = sobj; p->cand = cand; p->nlabel = nlabel; p->cnt = cnt; p->selgt = selgt;
    p->matched = matched; p->miou = miou; p->numfg = numfg; p->partial = partial;
  }
  return used;
}

static int sim_fill(SimotaParams& p, const cvhip_simota_desc* d, const void* const* raws, const float* targets, void* ws) {
  if (!d || !raws || !targets || !ws) return CVHIP_ERR_INVALID;
  if (d->L <= 0 || d->L > kSimMaxLevels || d->B <= 0 || d->G <= 0 || d->nc <= 0 || d->nc > 256) return CVHIP_ERR_INVALID;
  memset(&p, 0, sizeof(p));
  int off = 0;
  for (int i = 0; i < d->L; ++i) {
    if (!raws[i] || d->H[i] <= 0 || d->W[i] <= 0 || d->ld[i] < 5 + d->nc) return CVHIP_ERR_INVALID;
    p.raw[i] = (const h16_t*)raws[i];
    p.ld[i] = d->ld[i];
    p.H[i] = d->H[i];
    p.W[i] = d->W[i];
    p.stride[i] = d->stride[i];
    p.off[i] = off;
    off += d->H[i] * d->W[i];
  }
  p.off[d->L] = off;
  if (off != d->A) return CVHIP_ERR_INVALID;
  if ((int64_t)d->B * d->A >= (1ll << 31)) return CVHIP_ERR_UNSUPPORTED;
  p.L = d->L;
  p.B = d->B;
  p.A = d->A;
  p.G = d->G;
  p.nc = d->nc;
  p.targets = targets;
  sim_ws_layout(d, &p, ws);
  return CVHIP_OK;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int64_t cvhip_simota_workspace_bytes(const cvhip_simota_desc* d) {
  if (!d || d->B <= 0 || d->A <= 0) return CVHIP_ERR_INVALID;
  return sim_ws_layout(d, nullptr, nullptr);
}

int cvhip_simota_loss_fwd(const cvhip_simota_desc* d, const void* const* raws, const float* targets, void* ws, float* out5, void* stream) {
  SimotaParams p;
  int rc = sim_fill(p, d, raws, targets, ws);
  if (rc != CVHIP_OK) return rc;
  if (!out5) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int BA = p.B * p.A;
  const int nb = cdiv(BA, 256);
  hipLaunchKernelGGL(sim_nlabel_kernel, dim3(cdiv(p.B, 64)), dim3(64), 0, st, p);
  hipLaunchKernelGGL(sim_prep_kernel, dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(sim_match_kernel, dim3(p.G, p.B), dim3(kSimMatchThreads), 0, st, p);
  hipLaunchKernelGGL(sim_resolve_kernel, dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(sim_loss_kernel<false>, dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(sim_finalize_kernel, dim3(1), dim3(1024), 0, st, p, nb, out5);
  return check_launch("simota_loss_fwd");
}

int cvhip_simota_loss_bwd(const cvhip_simota_desc* d, const void* const* raws, const float* targets, void* ws, const float* gout,
                          void* const* draws, void* stream) {
  SimotaParams p;
  int rc = sim_fill(p, d, raws, targets, ws);
  if (rc != CVHIP_OK) return rc;
  if (!draws) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < p.L; ++i) {
    if (!draws[i]) return CVHIP_ERR_INVALID;
    p.draw[i] = (h16_t*)draws[i];
    rc = zero_fill(p.draw[i], (int64_t)p.B * p.H[i] * p.W[i] * p.ld[i] * 2, st);
    if (rc != CVHIP_OK) return rc;
  }
  p.gout = gout;
  hipLaunchKernelGGL(sim_loss_kernel<true>, dim3(cdiv(p.B * p.A, 256)), dim3(256), 0, st, p);
  return check_launch("simota_loss_bwd");
}

/* copy of the assignment (matched gt per anchor, -1 = background; its IoU) out of the workspace — tests / diagnostics */
int cvhip_simota_read_assignment(const cvhip_simota_desc* d, void* ws, int32_t* matched_out, float* miou_out, void* stream) {
  if (!d || !ws || !matched_out || !miou_out || d->B <= 0 || d->A <= 0) return CVHIP_ERR_INVALID;
  SimotaParams p;
  memset(&p, 0, sizeof(p));
  sim_ws_layout(d, &p, ws);
  const int n = d->B * d->A;
  hipLaunchKernelGGL(sim_copy_assign_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p.matched, p.miou, matched_out, miou_out, n);
  return check_launch("sim_copy_assign_kernel");
}

}  // extern "C"
