// post_batch.hip — detection post-processing for a whole BATCH in one launch set, no host round trip:
//   candidates (confidence filter) -> top-`cap` selection by score -> class-offset boxes -> greedy NMS -> fixed-capacity outputs.
//
// Replaces the per-image python loops of reference src/models/yolov5.py:62-153 (non_max_suppression: best-class path) and
// src/models/yolox.py:48-68 (yolox_post_process: torchvision.ops.batched_nms), plus torch.sort / boolean-mask indexing /
// torchvision.ops.nms inside them; cvhip_sort_keys_u64 is the device sort the python mirrors of src/models/modules/nms.py
// (multiclass_nms / batched_nms) use instead of torch.sort.
//
// Semantics = the reference's with `max_nms := cap` (the reference keeps the 30000 best candidates per image; here the `cap`
// best, cap <= 8192): identical whenever an image has <= cap candidates; `overflow[b]` says when it had more.
// Order: descending score, ties by ascending row (a stable sort). NMS arithmetic is the fp32 arithmetic of post.hip
// (bit-exact vs the CPU restatement): FMA contraction off.
#pragma clang fp contract(off)
#include "common.h"

namespace cvhip {

constexpr int kPostMaxCap = 8192;  // 64 KB of 64-bit keys in LDS

// orderable key: ascending key order == descending score, ties ascending row
__device__ __forceinline__ unsigned long long post_key(float score, unsigned row) {
  if (score == 0.f) score = 0.f;  // -0.0 compares EQUAL to +0.0 in the reference's sort: one key for both (ties then go by row)
  unsigned u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone in the float order
  return ((unsigned long long)(0xFFFFFFFFu - u) << 32) | row;
}

struct PostParams {
  const float* pred;  // [B][n][5 + nc (+ extra)] decoded rows: cx, cy, w, h, obj, cls...
  int B, n, no, nc;
  float conf_thres, iou_thres, class_offset;
  int mode;           // 0: YOLOv5 non_max_suppression   1: YOLOX yolox_post_process (batched_nms offsets)
  int multi_label;    // mode 0 only: every (row, class) with obj*cls > conf_thres is a candidate (yolov5.py:106-108)
  int cap, max_det, ncol, cand_cap;
  unsigned long long* keys;   // [B][cand_cap] candidate keys (unordered)
  int* ncand;                 // [B]
  int* nsel;                  // [B]
  float4* nms_boxes;          // [B][cap] offset boxes in selection order
  float* sel_rows;            // [B][cap][ncol] output rows in selection order
  unsigned long long* mask;   // [B][cap][cap/64]
  float* dets;                // [B][max_det][ncol]
  int* counts;                // [B]
  int* overflow;              // [B]
};

// ---- 1. candidates ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_candidates_kernel(const PostParams p) {
  const int b = blockIdx.y;
  const float* base = p.pred + (int64_t)b * p.n * p.no;
  unsigned long long* keys = p.keys + (int64_t)b * p.cand_cap;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < p.n; r += gridDim.x * 256) {
    const float* row = base + (int64_t)r * p.no;
    const float obj = row[4];
    if (p.multi_label) {
      // yolov5.py:69,104-108: obj > conf_thres; every class with cls * obj > conf_thres is its own detection, in (row, class) order
      if (!(obj > p.conf_thres)) continue;
      for (int c = 0; c < p.nc; ++c) {
        const float v = row[5 + c] * obj;
        if (v > p.conf_thres) {
          const int slot = atomicAdd(&p.ncand[b], 1);
          if (slot < p.cand_cap) keys[slot] = post_key(v, (unsigned)(r * p.nc + c));
        }
      }
      continue;
    }
    // yolov5.py:104,111: conf = (cls * obj).max()  — the maximum of the PRODUCTS (rounding can merge neighbours);
    // yolox.py:55-57: class_conf = cls.max(), score = obj * class_conf
    const float m0 = p.mode == 0 ? obj : 1.0f;
    float best = row[5] * m0;
    for (int c = 1; c < p.nc; ++c) best = fmaxf(best, row[5 + c] * m0);
    const float score = p.mode == 0 ? best : obj * best;
    // yolov5.py:69,112: obj > conf_thres, then conf > conf_thres;  yolox.py:57: obj * class_conf >= conf_thre
    const bool ok = p.mode == 0 ? (obj > p.conf_thres && score > p.conf_thres) : (score >= p.conf_thres);
    if (ok) {
      const int slot = atomicAdd(&p.ncand[b], 1);
      if (slot < p.cand_cap) keys[slot] = post_key(score, (unsigned)r);
    }
  }
}

// ---- 2. selection + sort + gather (one block per image) ------------------------------------------------------------------------------
// radix select: the key of rank `want` (0-based) among the m keys of the image (keys are unique: they carry the row)
__device__ unsigned long long post_radix_select(const unsigned long long* keys, int m, int want, unsigned* hist /*[256]*/, unsigned long long* sh) {
  unsigned long long prefix = 0ull, pmask = 0ull;
  for (int byte = 7; byte >= 0; --byte) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
      const unsigned long long k = keys[i];
      if ((k & pmask) == prefix) atomicAdd(&hist[(unsigned)(k >> (8 * byte)) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0, d = 0;
      for (; d < 256; ++d) {
        if (acc + (int)hist[d] > want) break;
        acc += (int)hist[d];
      }
      sh[0] = (unsigned long long)d;
      sh[1] = (unsigned long long)acc;
    }
    __syncthreads();
    prefix |= sh[0] << (8 * byte);
    pmask |= 0xFFull << (8 * byte);
    want -= (int)sh[1];
    __syncthreads();
  }
  return prefix;
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* s, int npad) {
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = s[i], c = s[ixj];
          const bool up = (i & k) == 0;
          if ((a > c) == up) {
            s[i] = c;
            s[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void post_select_kernel(const PostParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* s = reinterpret_cast<unsigned long long*>(smem_raw);  // [npad]
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh[2];
  __shared__ float red[1024];
  __shared__ int s_cnt;
  const int b = blockIdx.x;
  const int m_all = p.ncand[b];
  const int m = m_all < p.cand_cap ? m_all : p.cand_cap;  // candidates beyond the buffer were dropped (overflow is flagged)
  const unsigned long long* keys = p.keys + (int64_t)b * p.cand_cap;
  const int nsel = m < p.cap ? m : p.cap;
  int npad = 1;
  while (npad < nsel) npad <<= 1;
  if (npad < 2) npad = 2;
  if (threadIdx.x == 0) {
    p.nsel[b] = nsel;
    p.overflow[b] = m_all > p.cap ? 1 : 0;
    s_cnt = 0;
  }
  __syncthreads();
  if (m <= p.cap) {
    for (int i = threadIdx.x; i < npad; i += blockDim.x) s[i] = i < m ? keys[i] : ~0ull;
  } else {
    // more candidates than the capacity: keep exactly the `cap` best (smallest keys)
    const unsigned long long kth = post_radix_select(keys, m, p.cap - 1, hist, sh);
    for (int i = threadIdx.x; i < npad; i += blockDim.x) s[i] = ~0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
      const unsigned long long k = keys[i];
      if (k <= kth) s[atomicAdd(&s_cnt, 1)] = k;
    }
  }
  __syncthreads();
  bitonic_sort_lds(s, npad);

  // gather rows in selection order; boxes: xywh -> xyxy (yolov5.py:52-59 / yolox.py:40-45: c -/+ wh/2)
  float4* nb = p.nms_boxes + (int64_t)b * p.cap;
  float* out = p.sel_rows + (int64_t)b * p.cap * p.ncol;
  const float* base = p.pred + (int64_t)b * p.n * p.no;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < nsel; i += blockDim.x) {
    const unsigned idx = (unsigned)(s[i] & 0xFFFFFFFFull);
    const unsigned r = p.multi_label ? idx / (unsigned)p.nc : idx;
    const float* row = base + (int64_t)r * p.no;
    const float cx = row[0], cy = row[1], w = row[2], h = row[3], obj = row[4];
    int cls = 0;
    const float m0 = p.mode == 0 ? obj : 1.0f;
    float best = row[5] * m0;
    if (p.multi_label) {
      cls = (int)(idx - r * (unsigned)p.nc);
      best = row[5 + cls] * obj;
    } else {
      for (int c = 1; c < p.nc; ++c) {
        const float v = row[5 + c] * m0;
        if (v > best) {  // first maximum wins (torch.max)
          best = v;
          cls = c;
        }
      }
    }
    float4 bx;
    bx.x = cx - w / 2;
    bx.y = cy - h / 2;
    bx.z = cx + w / 2;
    bx.w = cy + h / 2;
    float* o = out + (int64_t)i * p.ncol;
    o[0] = bx.x;
    o[1] = bx.y;
    o[2] = bx.z;
    o[3] = bx.w;
    if (p.mode == 0) {
      o[4] = best;
      o[5] = (float)cls;
    } else {
      o[4] = obj;
      o[5] = best;
      o[6] = (float)cls;
    }
    nb[i] = bx;
    mx = fmaxf(mx, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
  }
  // class offsets: yolov5.py:139-140 boxes + cls * max_wh;  torchvision batched_nms: boxes + idx * (boxes.max() + 1)
  float off_unit = p.class_offset;
  if (p.mode == 1) {
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
      __syncthreads();
    }
    off_unit = red[0] + 1.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nsel; i += blockDim.x) {
    const float cls = out[(int64_t)i * p.ncol + (p.mode == 0 ? 5 : 6)];
    const float c = cls * off_unit;
    float4 bx = nb[i];
    bx.x = bx.x + c;
    bx.y = bx.y + c;
    bx.z = bx.z + c;
    bx.w = bx.w + c;
    nb[i] = bx;
  }
}

// ---- 3. NMS over the selected boxes of every image (the arithmetic of post.hip) ---------------------------------------------------------
__device__ __forceinline__ bool post_iou_gt(const float4 a, const float4 b, float thr) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float area_a = (a.z - a.x) * (a.w - a.y);
  const float area_b = (b.z - b.x) * (b.w - b.y);
  const float ovr = inter / (area_a + area_b - inter);
  return ovr > thr;
}

__global__ __launch_bounds__(64) void post_nms_mask_kernel(const PostParams p) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = p.nsel[b];
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  const int nblk = p.cap / 64;
  const float4* boxes = p.nms_boxes + (int64_t)b * p.cap;
  unsigned long long* mask = p.mask + (int64_t)b * p.cap * nblk;
  const int lane = threadIdx.x;
  __shared__ float4 rows[64];
  const int ri = rb * 64 + lane, cj = cb * 64 + lane;
  rows[lane] = ri < n ? boxes[ri] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 cbx = cj < n ? boxes[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  unsigned long long mine = 0ull;
  for (int r = 0; r < 64; ++r) {
    const int i = rb * 64 + r;
    const bool hit = (i < n) && (cj < n) && (cj > i) && post_iou_gt(rows[r], cbx, p.iou_thres);
    const unsigned long long word = __ballot(hit);
    if (lane == r) mine = word;
  }
  if (ri < n) mask[(int64_t)ri * nblk + cb] = mine;
}

__global__ __launch_bounds__(64) void post_nms_scan_kernel(const PostParams p) {
  const int b = blockIdx.x;
  const int n = p.nsel[b];
  const int nblk_all = p.cap / 64;
  const int nblk = (n + 63) / 64;
  const unsigned long long* mask = p.mask + (int64_t)b * p.cap * nblk_all;
  const float* rows = p.sel_rows + (int64_t)b * p.cap * p.ncol;
  float* dets = p.dets + (int64_t)b * p.max_det * p.ncol;
  const int lane = threadIdx.x;
  __shared__ unsigned long long removed_ws[kPostMaxCap / 64];
  for (int w = lane; w < nblk; w += 64) removed_ws[w] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int blk = 0; blk < nblk && cnt < p.max_det; ++blk) {
    unsigned long long cur = removed_ws[blk];
    const int lim = min(64, n - blk * 64);
    for (int r = 0; r < lim && cnt < p.max_det; ++r) {
      if ((cur >> r) & 1ull) continue;  // wave-uniform
      const int i = blk * 64 + r;
      if (lane < p.ncol) dets[(int64_t)cnt * p.ncol + lane] = rows[(int64_t)i * p.ncol + lane];
      ++cnt;
      const unsigned long long* row = mask + (int64_t)i * nblk_all;
      cur |= row[blk];
      for (int w = blk + 1 + lane; w < nblk; w += 64) removed_ws[w] |= row[w];
    }
    __syncthreads();
  }
  // zero the unused tail so the fixed-capacity output is fully defined
  for (int i = cnt * p.ncol + lane; i < p.max_det * p.ncol; i += 64) dets[i] = 0.f;
  if (lane == 0) p.counts[b] = cnt;
}

// ---- device sort of 64-bit keys (any n): LDS bitonic blocks + global merge steps ---------------------------------------------------------
constexpr int kSortChunk = 4096;

__global__ __launch_bounds__(1024) void sort_chunk_kernel(unsigned long long* keys, int64_t npad, int k_lo, int k_hi) {
  // one block per 4096-key chunk: runs the bitonic stages k in [k_lo, k_hi] restricted to strides j < 4096 (within the chunk)
  __shared__ unsigned long long s[kSortChunk];
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
  for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) s[i] = keys[base + i];
  __syncthreads();
  for (int64_t k = k_lo; k <= k_hi; k <<= 1) {
    for (int j = (int)(k >> 1 < kSortChunk ? k >> 1 : kSortChunk >> 1); j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = s[i], c = s[ixj];
          const bool up = ((base + i) & k) == 0;
          if ((a > c) == up) {
            s[i] = c;
            s[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) keys[base + i] = s[i];
}

__global__ __launch_bounds__(256) void sort_global_step_kernel(unsigned long long* keys, int64_t npad, int64_t k, int64_t j) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256) {
    const int64_t ixj = i ^ j;
    if (ixj > i) {
      const unsigned long long a = keys[i], c = keys[ixj];
      const bool up = (i & k) == 0;
      if ((a > c) == up) {
        keys[i] = c;
        keys[ixj] = a;
      }
    }
  }
}

__global__ __launch_bounds__(256) void sort_make_keys_kernel(const float* scores, int64_t n, int64_t npad, unsigned long long* keys) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256)
    keys[i] = i < n ? post_key(scores[i], (unsigned)i) : ~0ull;
}

__global__ __launch_bounds__(256) void sort_emit_kernel(const unsigned long long* keys, int64_t n, int64_t* order) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) order[i] = (int64_t)(keys[i] & 0xFFFFFFFFull);
}

static int64_t sort_npad(int64_t n) {
  int64_t p = kSortChunk;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int64_t cvhip_detect_postprocess_workspace_bytes(int32_t B, int32_t cand_cap, int32_t cap) {
  if (B <= 0 || cand_cap <= 0 || cap < 64 || cap > kPostMaxCap || (cap & (cap - 1))) return -1;  // capacity: a power of two in [64, 8192]
  const int64_t keys = (int64_t)B * cand_cap * 8;
  const int64_t ints = (int64_t)B * 2 * 4;
  const int64_t boxes = (int64_t)B * cap * 16;
  const int64_t rows = (int64_t)B * cap * 8 * 4;
  const int64_t mask = (int64_t)B * cap * (cap / 64) * 8;
  return keys + ((ints + 15) & ~15ll) + boxes + rows + mask + 256;
}

int cvhip_detect_postprocess(const float* pred, int32_t B, int32_t n, int32_t no, int32_t nc, float conf_thres, float iou_thres,
                             float class_offset, int32_t mode, int32_t multi_label, int32_t cand_cap, int32_t cap, int32_t max_det,
                             void* workspace, float* dets, int32_t* counts, int32_t* overflow, void* stream) {
  if (!pred || !workspace || !dets || !counts || !overflow) return CVHIP_ERR_INVALID;
  if (B <= 0 || n <= 0 || nc <= 0 || no < 5 + nc || max_det <= 0 || (mode != 0 && mode != 1) || cand_cap <= 0) return CVHIP_ERR_INVALID;
  if (multi_label && (mode != 0 || (int64_t)n * nc >= (1ll << 32))) return CVHIP_ERR_UNSUPPORTED;
  if (!multi_label && cand_cap < n) return CVHIP_ERR_INVALID;
  if (cap < 64 || cap > kPostMaxCap || (cap & (cap - 1)) || (int64_t)n >= (1ll << 31)) return CVHIP_ERR_UNSUPPORTED;
  if ((((uintptr_t)workspace) & 15)) return CVHIP_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  PostParams p{};
  p.pred = pred;
  p.B = B;
  p.n = n;
  p.no = no;
  p.nc = nc;
  p.conf_thres = conf_thres;
  p.iou_thres = iou_thres;
  p.class_offset = class_offset;
  p.mode = mode;
  p.multi_label = multi_label ? 1 : 0;
  p.cand_cap = cand_cap;
  p.cap = cap;
  p.max_det = max_det;
  p.ncol = mode == 0 ? 6 : 7;
  unsigned char* w = (unsigned char*)workspace;
  p.keys = (unsigned long long*)w;
  w += (int64_t)B * cand_cap * 8;
  p.ncand = (int*)w;
  p.nsel = p.ncand + B;
  w += (((int64_t)B * 2 * 4) + 15) & ~15ll;
  p.nms_boxes = (float4*)w;
  w += (int64_t)B * cap * 16;
  p.sel_rows = (float*)w;
  w += (int64_t)B * cap * 8 * 4;
  p.mask = (unsigned long long*)w;
  p.dets = dets;
  p.counts = counts;
  p.overflow = overflow;
  int st = zero_fill(p.ncand, (size_t)B * 2 * 4, s);
  if (st) return st;
  int gx = (n + 255) / 256;
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(post_candidates_kernel, dim3(gx, B), dim3(256), 0, s, p);
  st = check_launch("post_candidates_kernel");
  if (st) return st;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(post_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPostMaxCap * 8);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(post_select_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL(post_select_kernel, dim3(B), dim3(1024), (size_t)cap * 8, s, p);
  st = check_launch("post_select_kernel");
  if (st) return st;
  const int nblk = cap / 64;
  hipLaunchKernelGGL(post_nms_mask_kernel, dim3(nblk, nblk, B), dim3(64), 0, s, p);
  st = check_launch("post_nms_mask_kernel");
  if (st) return st;
  hipLaunchKernelGGL(post_nms_scan_kernel, dim3(B), dim3(64), 0, s, p);
  return check_launch("post_nms_scan_kernel");
}

int64_t cvhip_sort_workspace_bytes(int64_t n) { return n <= 0 ? 64 : sort_npad(n) * 8; }

int cvhip_argsort_desc_f32(const float* scores, int64_t n, void* workspace, int64_t* order, void* stream) {
  if (n < 0 || (n > 0 && (!scores || !workspace || !order)) || n > (1ll << 29)) return CVHIP_ERR_INVALID;
  if (n == 0) return CVHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t npad = sort_npad(n);
  unsigned long long* keys = (unsigned long long*)workspace;
  int g = (int)((npad + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(sort_make_keys_kernel, dim3(g), dim3(256), 0, s, scores, n, npad, keys);
  const int chunks = (int)(npad / kSortChunk);
  // stages k = 2 .. 4096 entirely inside the chunks
  hipLaunchKernelGGL(sort_chunk_kernel, dim3(chunks), dim3(1024), 0, s, keys, npad, 2, kSortChunk);
  for (int64_t k = 2 * kSortChunk; k <= npad; k <<= 1) {
    for (int64_t j = k >> 1; j >= kSortChunk; j >>= 1) hipLaunchKernelGGL(sort_global_step_kernel, dim3(g), dim3(256), 0, s, keys, npad, k, j);
    hipLaunchKernelGGL(sort_chunk_kernel, dim3(chunks), dim3(1024), 0, s, keys, npad, (int)k, (int)k);  // strides < 4096 of stage k
  }
  hipLaunchKernelGGL(sort_emit_kernel, dim3(g), dim3(256), 0, s, keys, n, order);
  return check_launch("argsort_desc_f32");
}

}  // extern "C"
