// post.hip — detection post-processing: YOLOv5 box decode, greedy NMS (64-lane ballot bitmask +
// single-wave scan) and pairwise IoU.
//
// Reference: src/models/detects/yolov5_detect.py:48-55 (decode), src/models/yolov5.py:62-153
// (non_max_suppression -> torchvision.ops.nms, third-party; semantics restated in oracle/torch_ref.py `nms`, pinned by the
// hand-derived known-answer vectors tests/golden/nms_kat.json),
// src/models/yolov5.py:27-49 (box_iou).
//
// Bit-exactness: the IoU predicate must round exactly like the fp32 CPU arithmetic
//   inter = max(0,xx2-xx1)*max(0,yy2-yy1);  ovr = inter / (area_i + area_j - inter);  ovr > thr
// so FMA contraction is disabled for this translation unit.
#pragma clang fp contract(off)
#include "common.h"

namespace cvhip {

// One block per (n, a, y) ROW of a level: the row's W * NO outputs are contiguous in `out` (fp32, fully coalesced stores) and its
// inputs are W pieces of NO consecutive 16-bit logits. Round 6: the first version decomposed a flat 64-bit element index with five
// 64-bit divisions by run-time values per element — 213 us per level launch for 275 MB in / 548 MB out (1.3 TB/s), 16 % of the
// YOLOv5-s inference batch; here the row is decoded once per block (scalar) and the element needs one multiply-high.
__global__ __launch_bounds__(256) void yolov5_decode_kernel(const h16_t* __restrict__ p, int ld, float* __restrict__ out, int N,
                                                            int A, int NO, int H, int W, float stride,
                                                            const float* __restrict__ anchors_px, int64_t img_stride,
                                                            int64_t lvl_off, unsigned inv_no) {
  const int64_t rows = (int64_t)N * A * H;
  const int row_elems = W * NO;
  for (int64_t rowi = blockIdx.x; rowi < rows; rowi += gridDim.x) {
    const int y = (int)(rowi % H);
    const int64_t r = rowi / H;
    const int a = (int)(r % A);
    const int n = (int)(r / A);
    const h16_t* const src = p + ((int64_t)(n * H + y) * W) * ld + a * NO;
    float* const dst = out + (int64_t)n * img_stride + (lvl_off + (int64_t)(a * H + y) * W) * NO;
    const float aw = anchors_px[a * 2], ah = anchors_px[a * 2 + 1];
    for (int t = threadIdx.x; t < row_elems; t += 256) {
      const int x = (int)__umulhi((unsigned)t, inv_no);  // t / NO (inv_no = ceil(2^32 / NO): exact for t * NO < 2^32)
      const int o = t - x * NO;
      const float v = (float)src[(int64_t)x * ld + o];
      const float s = 1.0f / (1.0f + expf(-v));
      float res;
      if (o == 0) res = (s * 2.0f - 0.5f + (float)x) * stride;
      else if (o == 1) res = (s * 2.0f - 0.5f + (float)y) * stride;
      else if (o == 2 || o == 3) {
        const float tt = s * 2.0f;
        res = (tt * tt) * (o == 2 ? aw : ah);
      } else res = s;
      dst[t] = res;
    }
  }
}

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float area_a = (a.z - a.x) * (a.w - a.y);
  const float area_b = (b.z - b.x) * (b.w - b.y);
  const float ovr = inter / (area_a + area_b - inter);
  return ovr > thr;
}

// mask[i][cb] bit j = IoU(box i, box cb*64+j) > thr  (only j > i matters to the scan).
// One wave per (row-block rb, col-block cb >= rb); lane = column box; 64 ballots give the 64 row words.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, int n, float thr,
                                                      unsigned long long* __restrict__ mask, int nblk) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int lane = threadIdx.x;
  __shared__ float4 rows[64];
  const int ri = rb * 64 + lane, cj = cb * 64 + lane;
  rows[lane] = ri < n ? boxes[ri] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 cbx = cj < n ? boxes[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  unsigned long long mine = 0ull;
  for (int r = 0; r < 64; ++r) {
    const int i = rb * 64 + r;
    const bool hit = (i < n) && (cj < n) && (cj > i) && iou_gt(rows[r], cbx, thr);
    const unsigned long long word = __ballot(hit);
    if (lane == r) mine = word;
  }
  if (ri < n) mask[(int64_t)ri * nblk + cb] = mine;
}

// single wave: walk boxes in score order; `removed` bit-vector lives in LDS (nblk words)
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int nblk,
                                                      int* __restrict__ keep, int* __restrict__ keep_count) {
  const int lane = threadIdx.x;
  // removed bit-vector in LDS: n <= 131072 boxes (reference caps at 30000: models/yolov5.py:79)
  __shared__ unsigned long long removed_ws[2048];
  for (int w = lane; w < nblk; w += 64) removed_ws[w] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int b = 0; b < nblk; ++b) {
    // all lanes read the same word (broadcast); bits of this block may be updated by rows of the same block
    unsigned long long cur = removed_ws[b];
    const int lim = min(64, n - b * 64);
    for (int r = 0; r < lim; ++r) {
      if ((cur >> r) & 1ull) continue;  // wave-uniform branch
      const int i = b * 64 + r;
      if (lane == 0) keep[cnt] = i;
      ++cnt;
      const unsigned long long* row = mask + (int64_t)i * nblk;
      // OR this row's upper-triangular words into `removed` (word b.. nblk-1)
      cur |= row[b];
      for (int w = b + 1 + lane; w < nblk; w += 64) removed_ws[w] |= row[w];
    }
    __syncthreads();
  }
  if (lane == 0) *keep_count = cnt;
}

__global__ __launch_bounds__(256) void box_iou_kernel(const float4* __restrict__ a, int n, const float4* __restrict__ b, int m,
                                                      float* __restrict__ out) {
  const int64_t total = (int64_t)n * m;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / m), c = (int)(i - (int64_t)r * m);
    const float4 A = a[r], B = b[c];
    const float area1 = (A.z - A.x) * (A.w - A.y), area2 = (B.z - B.x) * (B.w - B.y);
    const float w = fmaxf(fminf(A.z, B.z) - fmaxf(A.x, B.x), 0.0f);
    const float h = fmaxf(fminf(A.w, B.w) - fmaxf(A.y, B.y), 0.0f);
    const float inter = w * h;
    out[i] = inter / (area1 + area2 - inter);
  }
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_yolov5_decode(const void* p, int32_t ld, float* out, int32_t N, int32_t A, int32_t NO, int32_t H, int32_t W,
                        float stride, const float* anchors_px, int64_t out_image_stride, int64_t out_level_offset,
                        void* stream) {
  if (!p || !out || !anchors_px || N <= 0 || A <= 0 || NO < 5 || H <= 0 || W <= 0 || ld < A * NO) return CVHIP_ERR_INVALID;
  if ((int64_t)W * NO * NO >= (1ll << 32)) return CVHIP_ERR_UNSUPPORTED;   // (the multiply-high division of the row offset)
  int64_t b = (int64_t)N * A * H;
  if (b > 256 * 256) b = 256 * 256;
  const unsigned inv_no = (unsigned)(((1ull << 32) + (unsigned)NO - 1) / (unsigned)NO);
  hipLaunchKernelGGL(yolov5_decode_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, (const h16_t*)p, ld, out, N, A,
                     NO, H, W, stride, anchors_px, out_image_stride, out_level_offset, inv_no);
  return check_launch("yolov5_decode_kernel");
}

int64_t cvhip_nms_workspace_bytes(int32_t n) {
  if (n <= 0) return 64;
  const int64_t nblk = (n + 63) / 64;
  return ((int64_t)n * nblk + nblk) * 8;
}

int cvhip_nms_sorted(const float* boxes, int32_t n, float iou_thr, void* workspace, int32_t* keep_idx,
                     int32_t* keep_count, void* stream) {
  if (n < 0 || !keep_count || (n > 0 && (!boxes || !workspace || !keep_idx))) return CVHIP_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return zero_fill(keep_count, sizeof(int32_t), s);
  if ((((uintptr_t)boxes) & 15) != 0) return CVHIP_ERR_INVALID;
  const int nblk = (n + 63) / 64;
  if (nblk > 2048) return CVHIP_ERR_UNSUPPORTED;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk), dim3(64), 0, s, (const float4*)boxes, n, iou_thr, mask, nblk);
  int st = check_launch("nms_mask_kernel");
  if (st) return st;
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, s, mask, n, nblk, keep_idx, keep_count);
  return check_launch("nms_scan_kernel");
}

int cvhip_box_iou(const float* a, int32_t n, const float* b, int32_t m, float* out, void* stream) {
  if (n < 0 || m < 0 || (n > 0 && m > 0 && (!a || !b || !out))) return CVHIP_ERR_INVALID;
  if (n == 0 || m == 0) return CVHIP_OK;
  if (((((uintptr_t)a) | ((uintptr_t)b)) & 15) != 0) return CVHIP_ERR_INVALID;
  int64_t blocks = cdiv64((int64_t)n * m, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(box_iou_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)a, n,
                     (const float4*)b, m, out);
  return check_launch("box_iou_kernel");
}

}  // extern "C"
