// conv_stem.hip — direct convolution for image stems on gfx950: 8 (zero-padded 3) input channels, <= 32 output channels.
//
//   Y[n][oy][ox][k] = sum_{r,s,c} X[n][oy*st - pad + r][ox*st - pad + s][c] * W[k][r][s][c]
//
// The implicit-GEMM kernel materialises the im2col rows in LDS: for the YOLOv5 stem (k6 s2, 640x640, batch 64) that is 9x the
// input (3.8 GB of global->LDS traffic per launch; the kernel runs at 1.4 TB/s of algorithmic traffic, 3.4x off the HBM
// bound). With 8 channels one tap of one pixel is exactly one 16-byte vector = one lane's 8 reduction elements of a
// v_mfma_f32_16x16x32_bf16 fragment, so the MFMA A fragments can be read STRAIGHT from an input patch held in LDS:
//   * block = 4 waves; tile = 4 output rows x 64 output columns x all K; the (3*st+R) x (63*st+S) input patch is loaded once
//     with coalesced 16-byte loads (zero outside the image), register-prefetched one tile ahead, LDS double-buffered: one
//     barrier per tile, blocks are persistent;
//   * lane (r, g) of k-step j reads tap 4j+g of output pixel ox = 16b + r: one ds_read_b128; stride-2 columns are stored
//     de-interleaved (even columns, then odd) so the 16 lanes of a fragment hit consecutive 16-byte slots (no bank conflicts);
//   * the whole weight tensor (<= 32 x 49 x 8) sits in registers as B fragments, rows permuted so that a lane ends up with 8
//     consecutive output channels of its pixel (one 16-byte store); BatchNorm partial sums stay in registers across tiles.
// Replaces aten::convolution for the first layer (reference src/models/backbones/yolov5_csp_darknet.py:83-91 stem,
// yolov7 backbone stem) — same C-ABI entry (cvhip_conv2d_fprop), picked by launch_igemm.
#include <stdlib.h>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

constexpr int kStemTH = 4, kStemTW = 64;
constexpr int kStemMaxPH = 3 * 2 + 7, kStemMaxPW = 63 * 2 + 7 + 1;  // R, S <= 7, stride <= 2 (PW rounded up to even)
constexpr int kStemPatchBytes = kStemMaxPH * kStemMaxPW * 16;
constexpr int kStemBlocks = 768;   // persistent grid (3 blocks per CU by LDS)

// XCD-aware start tile of a persistent block (CVHIP_STEM_XCD=1; default OFF): hardware places block b on XCD b % 8; with the identity
// mapping the vertically adjacent tiles (t, t + tiles_x) — which share R - stride input rows — run on different XCDs and each fetches
// the shared rows into its own L2 (the stem kernels read 1.3 - 2.0x their algorithmic input bytes: profiles/r05_pmc_summary.txt).
// Giving each XCD a contiguous chunk of the tile space makes them L2 neighbours. MEASURED, round 5 (VERDICT r04 task 7's question):
// no gain — fprop 256 / 260 us -> 266 / 275 us, fused weight gradient 372 / 379 -> 359 / 383 us: the re-fetched halo rows come out of
// the Infinity Cache, they are not what these (latency-bound, 3.3 TB/s) kernels wait for. An LDS row ring across vertically
// consecutive tiles would remove the same bytes and was therefore not built.
__device__ __forceinline__ int stem_first_tile(int xcd) { return xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x; }
static int stem_xcd_mode() { return 0; }   // (XCD-contiguous tile chunks: measured, no gain — DESIGN.md 4.000 "Stem halo")

struct StemParams {
  const h16_t* x;
  const float* xf;  // != NULL: fp32 NCHW image [NB][planes][IH][IW] instead of x (round 4: no layout / precision pass in front of the stem)
  int planes;
  const h16_t* w;  // [K][R*S*8] bf16
  h16_t* y;
  const float* bias;
  float* stats;
  int stats_acc;
  int bias_n;
  const float *ep_scale, *ep_shift;  // fused epilogue (never with stats): out = act((acc + bias) * ep_scale + ep_shift)
  int ep_act;
  float ep_ap;
  int NB, IH, IW, OH, OW, K, y_ld, R, S, pad_h, pad_w;
  int tiles_x, tiles_y, ntiles;
  int xcd;  // XCD-aware start tile (stem_first_tile)
};


// activation of 8 / 4 values with ONE switch (a switch per element multiplied the unrolled epilogue's code size and pushed the
// 256-wide streaming kernel's accumulators into scratch)
template <int NV>
__device__ __forceinline__ void stem_act_vec(float (&v)[NV], int act, float ap) {
  switch (act) {
    case CVHIP_ACT_NONE: break;
    case CVHIP_ACT_RELU:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_RELU, ap);
      break;
    case CVHIP_ACT_SILU:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_SILU, ap);
      break;
    case CVHIP_ACT_LEAKY:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_LEAKY, ap);
      break;
    case CVHIP_ACT_SIGMOID:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_SIGMOID, ap);
      break;
    default:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_HSWISH, ap);
      break;
  }
}

template <int ST, int NSTEP, bool STATS, bool EPI = false>
__global__ __launch_bounds__(256, 2) void stem_fprop_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStemPatchBytes];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int R = p.R, S = p.S, taps = R * S;
  const int PH = (kStemTH - 1) * ST + R;
  const int PW = ((kStemTW - 1) * ST + S + 1) & ~1;  // even, so the de-interleaved halves are equal
  const int HALF = PW >> 1;
  const int nchunk = PH * PW;

  // ---- B fragments (weights) in registers: LDS-free, loaded once. Row i of fragment a <-> channel (i>>2)*8 + a*4 + (i&3)
  h16x8 wb[2][NSTEP];
  int aoff[NSTEP];
  {
    const int i = lane & 15;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ch = (i >> 2) * 8 + a * 4 + (i & 3);
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) {
        const int tap = j * 4 + g;
        h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ch < p.K && tap < taps) v = *reinterpret_cast<const h16x8*>(p.w + ((int64_t)ch * taps + tap) * 8);
        wb[a][j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      int tap = j * 4 + g;
      if (tap >= taps) tap = taps - 1;  // weight fragment is zero there; any finite operand will do
      const int tr = tap / S, ts = tap - tr * S;
      // patch slot of column c: ST == 2: (c & 1) * HALF + (c >> 1); ST == 1: c.   column of (ox, ts) = ox * ST + ts
      aoff[j] = (tr * PW + (ST == 2 ? (ts & 1) * HALF + (ts >> 1) : ts)) * 16;
    }
  }

  // ---- patch loader: thread owns chunks t, t+256, ... of the PH x PW patch (row-major, 16 B per pixel); their patch
  // coordinates and LDS slots do not depend on the tile
  constexpr int LD_IT = (kStemMaxPH * kStemMaxPW + 255) / 256;
  uint4 pre[LD_IT];
  int pk[LD_IT], loff[LD_IT];
#pragma unroll
  for (int i = 0; i < LD_IT; ++i) {
    const int q = t + i * 256;
    const int pr = q / PW, pc = q - pr * PW;
    pk[i] = q < nchunk ? ((pr << 16) | pc) : -1;
    loff[i] = (pr * PW + (ST == 2 ? (pc & 1) * HALF + (pc >> 1) : pc)) * 16;
  }
  auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
    const int tx = tile % p.tiles_x;
    const int rest = tile / p.tiles_x;
    const int ty = rest % p.tiles_y;
    n = rest / p.tiles_y;
    oy0 = ty * kStemTH;
    ox0 = tx * kStemTW;
  };
  auto gload = [&](int tile) {
    int n, oy0, ox0;
    tile_origin(tile, n, oy0, ox0);
    const int iy0 = oy0 * ST - p.pad_h, ix0 = ox0 * ST - p.pad_w;
    if (p.xf) {  // block-uniform: planar fp32 image, one 4-byte load per real channel (adjacent lanes = adjacent columns: coalesced)
      const int64_t plane = (int64_t)p.IH * p.IW;
      const float* img = p.xf + (int64_t)n * p.planes * plane;
#pragma unroll
      for (int i = 0; i < LD_IT; ++i) {
        const int iy = iy0 + (pk[i] >> 16), ix = ix0 + (pk[i] & 0xffff);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        if (pk[i] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
          const float* px = img + (int64_t)iy * p.IW + ix;
          c0 = px[0];
          if (p.planes > 1) c1 = px[plane];
          if (p.planes > 2) c2 = px[2 * plane];
          if (p.planes > 3) c3 = px[3 * plane];
        }
        f32x8 v8;
        v8.v[0] = c0; v8.v[1] = c1; v8.v[2] = c2; v8.v[3] = c3;
        v8.v[4] = v8.v[5] = v8.v[6] = v8.v[7] = 0.f;
        pre[i] = pack8(v8);
      }
    } else {
    const h16_t* img = p.x + (int64_t)n * p.IH * p.IW * 8;
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      const int iy = iy0 + (pk[i] >> 16), ix = ix0 + (pk[i] & 0xffff);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (pk[i] >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
        v = *reinterpret_cast<const uint4*>(img + (int64_t)(iy * p.IW + ix) * 8);
      pre[i] = v;
    }
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* dst = smem + buf * kStemPatchBytes;
#pragma unroll
    for (int i = 0; i < LD_IT; ++i)
      if (pk[i] >= 0) *reinterpret_cast<uint4*>(dst + loff[i]) = pre[i];
  };

  float s1[2][4], s2[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) s1[a][q] = s2[a][q] = 0.f;

  int tile = stem_first_tile(p.xcd);
  if (tile < p.ntiles) {
    gload(tile);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < p.ntiles) gload(nxt);

    // ---- compute: wave = output row of the tile, 4 fragments of 16 output columns
    const unsigned char* base = smem + cur * kStemPatchBytes + ((wave * ST) * PW + r) * 16;
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      h16x8 xa[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) xa[b] = *reinterpret_cast<const h16x8*>(base + aoff[j] + b * 16 * 16);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = CVHIP_MFMA_16X16X32(wb[a][j], xa[b], acc[a][b], 0, 0, 0);
    }

    // ---- epilogue: lane (r, g) holds channels g*8 .. g*8+7 of pixel (oy, ox0 + 16 b + r)
    int n, oy0, ox0;
    tile_origin(tile, n, oy0, ox0);
    const int oy = oy0 + wave;
    const int ch0 = g * 8;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ox = ox0 + b * 16 + r;
      const bool ok = oy < p.OH && ox < p.OW;
      f32x8 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v.v[q] = ok ? acc[0][b][q] : 0.f;
        v.v[4 + q] = ok ? acc[1][b][q] : 0.f;
      }
      if constexpr (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s1[0][q] += v.v[q];
          s2[0][q] += v.v[q] * v.v[q];
          s1[1][q] += v.v[4 + q];
          s2[1][q] += v.v[4 + q] * v.v[4 + q];
        }
      }
      if (ok && ch0 < p.K) {
        if (p.bias) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (ch0 + q < p.bias_n) v.v[q] += p.bias[ch0 + q];
        }
        if constexpr (EPI) {  // K <= 32: the constants are L1-resident
          if (p.ep_scale) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int cq = ch0 + q < p.K ? ch0 + q : p.K - 1;
              v.v[q] = v.v[q] * p.ep_scale[cq] + p.ep_shift[cq];
            }
          }
          stem_act_vec<8>(v.v, p.ep_act, p.ep_ap);
        }
        h16_t* yrow = p.y + ((int64_t)(n * p.OH + oy) * p.OW + ox) * p.y_ld + ch0;
        if (ch0 + 7 < p.K && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0)) {
          *reinterpret_cast<uint4*>(yrow) = pack8(v);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (ch0 + q < p.K) yrow[q] = (h16_t)v.v[q];
        }
      }
    }
    if (nxt < p.ntiles) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (STATS) {
    if (p.stats) {
      float* red = reinterpret_cast<float*>(smem);  // [4 waves][32][2]; every wave passed the loop's final barrier
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float u1 = row16_sum(s1[a][q]), u2 = row16_sum(s2[a][q]);
          if (r == 0) {
            const int lc = g * 8 + a * 4 + q;
            red[(wave * 32 + lc) * 2 + 0] = u1;
            red[(wave * 32 + lc) * 2 + 1] = u2;
          }
        }
      __syncthreads();
      if (t < p.K) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          u1 += red[(w * 32 + t) * 2 + 0];
          u2 += red[(w * 32 + t) * 2 + 1];
        }
        if (p.stats_acc) {
          acc_add2(reinterpret_cast<double*>(p.stats), blockIdx.x, p.K, t, u1, u2);
        } else {
          float* dst = p.stats + (int64_t)blockIdx.x * 2 * p.K;
          dst[t] = u1;
          dst[p.K + t] = u2;
        }
      }
    }
  }
}

// ---- weight gradient -----------------------------------------------------------------------------
//   dW[k][tap][c] += sum over output pixels dY[pix][k] * X[pix*st - pad + tap][c]        (fp32 atomics, [K][R*S][8])
// Same tiles and the same LDS patch as the forward kernel; the reduction runs over pixels, so both MFMA operands are
// gathered with the gfx950 LDS transpose read (ds_read_b64_tr_b16): A = dY^T from a row-major [pixel][32 ch] tile (layout and
// swizzle of conv_wgrad.hip), B = X^T straight from the patch — 16 B columns of a fragment are the 8 channels of TWO taps,
// lanes with (lane & 2) point at the second tap's pixel; rows are 8 consecutive output columns = 8 consecutive 16-byte patch
// slots. A wave owns every 4th column fragment (<= NFW of them) and keeps its slice of dW in registers across all tiles of
// the (persistent) block: one atomic epilogue per block. The general wgrad kernel re-gathers x per tap and re-reads dY per
// 128-column tile (450 us for the YOLOv5-s stem; HBM bound ~170 us).
typedef __attribute__((address_space(3))) h16x4 stem_lds_h16x4;
__device__ __forceinline__ h16x8 stem_tr_read8(const unsigned char* p0, const unsigned char* p1) {
  h16x4 lo = CVHIP_DS_READ_TR16_B64((stem_lds_h16x4*)(p0));
  h16x4 hi = CVHIP_DS_READ_TR16_B64((stem_lds_h16x4*)(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct StemWgradParams {
  const h16_t* x;
  const float* xf;  // as in StemParams
  int planes;
  const h16_t* dy;
  float* dw;  // [K][R*S][8] fp32, accumulated with atomics
  int NB, IH, IW, OH, OW, K, dy_ld, R, S, pad_h, pad_w;
  int tiles_x, tiles_y, ntiles;
  int xcd;  // XCD-aware start tile (stem_first_tile)
  // BNB instances (round 5): `dy` is dz, the gradient at the OUTPUT of the stem's Conv-BN-act layer; the BN + activation backward is
  // applied ON LOAD (dy = sc*du + b1*y + c1 with du = dz*act'(sc*y + sh), rounded to 16 bits as the stand-alone pass stores it), so the
  // apply pass (read dz, read y, write dy) and the dy tensor disappear — the image stem has no input gradient, the weight gradient is
  // dy's only consumer. (sum du, sum du*xhat) come from the layer's fp64 accumulator; block 0 stores dgamma / dbeta.
  const h16_t* y;
  int y_ld;
  const float *scale, *shift, *mean, *invstd;
  const double* acc;
  int acc_ld;
  float inv_count;
  int act;
  float ap;
  float *o_dgamma, *o_dbeta;
  int accumulate;
};

template <int ST, int NFW, bool BNB = false>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemWgradParams p) {
  constexpr int DY_BYTES = kStemTH * kStemTW * 64;  // [256 pixels][32 channels] bf16
  // BNB: the 4 x 32 per-channel constants live in the LDS (in registers they cost the third resident block: 194 VGPRs, 423 us)
  __shared__ __attribute__((aligned(16))) unsigned char smem[kStemPatchBytes + DY_BYTES + (BNB ? 4 * 32 * 4 : 0)];
  unsigned char* const sP = smem;
  unsigned char* const sD = smem + kStemPatchBytes;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, q = (lane >> 2) & 3;
  const int R = p.R, S = p.S, taps = R * S;
  const int nfrag = (taps + 1) >> 1;  // 16-column fragments: two taps x 8 channels
  const int PH = (kStemTH - 1) * ST + R;
  const int PW = ((kStemTW - 1) * ST + S + 1) & ~1;
  const int HALF = PW >> 1;
  const int nchunk = PH * PW;

  // ---- loaders (patch: as in stem_fprop_kernel; dY: thread owns 16-byte chunk (t & 3) of pixels (t >> 2) + 64 i)
  constexpr int LD_IT = (kStemMaxPH * kStemMaxPW + 255) / 256;
  uint4 pre[LD_IT], pred[4];
  uint4 predy[BNB ? 4 : 1];
  unsigned pvalid = 0;  // BNB: which of this thread's 4 dY pixels exist (the transform of a zero-filled pixel is c1, not zero)
  // (row << 16 | column) and LDS byte offset of this thread's i-th patch slot: tables in registers — or, in the BNB instances (which
  // need the registers for the second operand's prefetch), recomputed per use with an exact multiply-shift division by PW
  int pk_t[BNB ? 1 : LD_IT], loff_t[BNB ? 1 : LD_IT];
  const unsigned pw_magic = (unsigned)((0x100000000ull + (unsigned)PW - 1) / (unsigned)PW);  // exact qq / PW for qq < 2^16 (conv_plan.h div_magic)
  auto slot = [&](int i, int& pkv, int& loffv) {
    if constexpr (BNB) {
      const int qq = t + i * 256;
      const int pr = (int)__umulhi((unsigned)qq, pw_magic), pc = qq - pr * PW;
      pkv = qq < nchunk ? ((pr << 16) | pc) : -1;
      loffv = (pr * PW + (ST == 2 ? (pc & 1) * HALF + (pc >> 1) : pc)) * 16;
    } else {
      pkv = pk_t[i];
      loffv = loff_t[i];
    }
  };
  if constexpr (!BNB) {
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      const int qq = t + i * 256;
      const int pr = qq / PW, pc = qq - pr * PW;
      pk_t[i] = qq < nchunk ? ((pr << 16) | pc) : -1;
      loff_t[i] = (pr * PW + (ST == 2 ? (pc & 1) * HALF + (pc >> 1) : pc)) * 16;
    }
  }
  auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
    const int tx = tile % p.tiles_x;
    const int rest = tile / p.tiles_x;
    const int ty = rest % p.tiles_y;
    n = rest / p.tiles_y;
    oy0 = ty * kStemTH;
    ox0 = tx * kStemTW;
  };
  const int dchunk = t & 3;
  auto gload = [&](int tile) {
    int n, oy0, ox0;
    tile_origin(tile, n, oy0, ox0);
    const int iy0 = oy0 * ST - p.pad_h, ix0 = ox0 * ST - p.pad_w;
    if (p.xf) {  // block-uniform: planar fp32 image, one 4-byte load per real channel (adjacent lanes = adjacent columns: coalesced)
      const int64_t plane = (int64_t)p.IH * p.IW;
      const float* img = p.xf + (int64_t)n * p.planes * plane;
#pragma unroll
      for (int i = 0; i < LD_IT; ++i) {
        int pkv, loffv;
        slot(i, pkv, loffv);
        const int iy = iy0 + (pkv >> 16), ix = ix0 + (pkv & 0xffff);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        if (pkv >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
          const float* px = img + (int64_t)iy * p.IW + ix;
          c0 = px[0];
          if (p.planes > 1) c1 = px[plane];
          if (p.planes > 2) c2 = px[2 * plane];
          if (p.planes > 3) c3 = px[3 * plane];
        }
        f32x8 v8;
        v8.v[0] = c0; v8.v[1] = c1; v8.v[2] = c2; v8.v[3] = c3;
        v8.v[4] = v8.v[5] = v8.v[6] = v8.v[7] = 0.f;
        pre[i] = pack8(v8);
      }
    } else {
    const h16_t* img = p.x + (int64_t)n * p.IH * p.IW * 8;
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      int pkv, loffv;
      slot(i, pkv, loffv);
      const int iy = iy0 + (pkv >> 16), ix = ix0 + (pkv & 0xffff);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (pkv >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
        v = *reinterpret_cast<const uint4*>(img + (int64_t)(iy * p.IW + ix) * 8);
      pre[i] = v;
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (t >> 2) + 64 * i;  // tile pixel: row px / 64, column px % 64
      const int oy = oy0 + (px >> 6), ox = ox0 + (px & 63);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      const bool ok = oy < p.OH && ox < p.OW && dchunk * 8 < p.K;
      if (ok) v = *reinterpret_cast<const uint4*>(p.dy + ((int64_t)(n * p.OH + oy) * p.OW + ox) * p.dy_ld + dchunk * 8);
      pred[i] = v;
      if constexpr (BNB) {
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (ok) w = *reinterpret_cast<const uint4*>(p.y + ((int64_t)(n * p.OH + oy) * p.OW + ox) * p.y_ld + dchunk * 8);
        predy[i] = w;
        pvalid = ok ? (pvalid | (1u << i)) : (pvalid & ~(1u << i));
      }
    }
  };
  // BNB: per-channel constants of this thread's channel vector (dchunk: thread-constant), from the layer's backward accumulator
  float* const sK = reinterpret_cast<float*>(smem + kStemPatchBytes + DY_BYTES);  // [4][32]: sc | sh | b1 | c1
  if constexpr (BNB) {
    float bsc[8], bsh[8], bb1[8], bc1[8];
    float* const kst = reinterpret_cast<float*>(sD);  // [2][32]: the dY tile is not in use yet
    if (t < 32) {
      double s1 = 0.0, s2 = 0.0;
      if (t < p.K) acc_fold2(p.acc, p.acc_ld, t, s1, s2);
      kst[t] = (float)s1;
      kst[32 + t] = (float)s2;
      if (blockIdx.x == 0 && t < p.K) {
        if (p.o_dbeta) p.o_dbeta[t] = p.accumulate ? p.o_dbeta[t] + (float)s1 : (float)s1;
        if (p.o_dgamma) p.o_dgamma[t] = p.accumulate ? p.o_dgamma[t] + (float)s2 : (float)s2;
      }
    }
    __syncthreads();
    const int c0 = dchunk * 8 < p.K ? dchunk * 8 : 0;
    float mu[8], is[8];
    load8c(p.scale, c0, p.K, bsc);
    load8c(p.shift, c0, p.K, bsh);
    load8c(p.mean, c0, p.K, mu);
    load8c(p.invstd, c0, p.K, is);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float q2 = bsc[j] * is[j] * (kst[32 + c0 + j] * p.inv_count);
      bb1[j] = -q2;
      bc1[j] = q2 * mu[j] - bsc[j] * (kst[c0 + j] * p.inv_count);
    }
    if (t < 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sK[c0 + j] = bsc[j];
        sK[32 + c0 + j] = bsh[j];
        sK[64 + c0 + j] = bb1[j];
        sK[96 + c0 + j] = bc1[j];
      }
    }
    __syncthreads();  // the constants are published (and the scratch is free) before the first lstore
  }
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
      int pkv, loffv;
      slot(i, pkv, loffv);
      if (pkv >= 0) *reinterpret_cast<uint4*>(sP + loffv) = pre[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (t >> 2) + 64 * i;
      const int h = (px & 3) | ((px >> 1) & 4);  // 32-byte-segment swizzle of conv_wgrad.hip (2 segments per 64-byte row)
      uint4 v = pred[i];
      if constexpr (BNB) {
        // two channel pairs at a time: (dz, y) 32-bit words in, one packed word out — keeps the transform's live registers small
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        const uint32_t wy[4] = {predy[i].x, predy[i].y, predy[i].z, predy[i].w};
        uint32_t wo[4];
        const float* k = sK + dchunk * 8;
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {
          float d0, d1, y0, y1;
          unpack2(wd[h2], d0, d1);
          unpack2(wy[h2], y0, y1);
          const int j = 2 * h2;
          const float u0 = y0 * k[j] + k[32 + j], u1 = y1 * k[j + 1] + k[32 + j + 1];
          float a0, a1;
          switch (p.act) {  // block-uniform
            case CVHIP_ACT_SILU: a0 = act_bwd(u0, CVHIP_ACT_SILU, p.ap); a1 = act_bwd(u1, CVHIP_ACT_SILU, p.ap); break;
            case CVHIP_ACT_RELU: a0 = act_bwd(u0, CVHIP_ACT_RELU, p.ap); a1 = act_bwd(u1, CVHIP_ACT_RELU, p.ap); break;
            case CVHIP_ACT_LEAKY: a0 = act_bwd(u0, CVHIP_ACT_LEAKY, p.ap); a1 = act_bwd(u1, CVHIP_ACT_LEAKY, p.ap); break;
            default: a0 = a1 = 1.f; break;
          }
          const float o0 = k[j] * (d0 * a0) + (k[64 + j] * y0 + k[96 + j]);
          const float o1 = k[j + 1] * (d1 * a1) + (k[64 + j + 1] * y1 + k[96 + j + 1]);
          wo[h2] = pack2(o0, o1);
        }
        v = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        if (!((pvalid >> i) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
      }
      *reinterpret_cast<uint4*>(sD + px * 64 + ((((dchunk >> 1) ^ h) & 1) << 5) + (dchunk & 1) * 16) = v;
    }
  };

  // ---- fragment geometry (lane constants)
  const int hsw = q | ((g & 1) << 2);
  const int px0 = 8 * g + q;                 // pixel row of this lane's first transposed read inside a 32-pixel step
  const int tapsel = (lane >> 1) & 1;        // which of the fragment's two taps this lane's 8-byte piece belongs to
  int boff[NFW];                             // patch byte offset of (tap of fragment f, this lane's piece), without pixel / row terms
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    const int f = wave + 4 * j;
    int tap = 2 * f + tapsel;
    if (tap >= taps) tap = taps - 1;  // odd tap count / fragments past the end: columns never stored
    const int tr = tap / S, ts = tap - tr * S;
    boff[j] = (tr * PW + (ST == 2 ? (ts & 1) * HALF + (ts >> 1) : ts)) * 16 + (lane & 1) * 8;
  }
  f32x4 acc[2][NFW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < NFW; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = stem_first_tile(p.xcd);
  if (tile < p.ntiles) gload(tile);
  for (; tile < p.ntiles; tile += gridDim.x) {
    lstore();
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < p.ntiles) gload(nxt);
#pragma unroll 2
    for (int step = 0; step < (kStemTH * kStemTW) / 32; ++step) {
      const int ty = step >> 1, xh = step & 1;
      h16x8 fd[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const unsigned char* base = sD + step * 32 * 64 + (((a ^ hsw) & 1) << 5) + (lane & 3) * 8;
        fd[a] = stem_tr_read8(base + px0 * 64, base + (px0 + 4) * 64);
      }
      const unsigned char* prow = sP + ((ty * ST) * PW + xh * 32 + px0) * 16;
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        if (wave + 4 * j < nfrag) {
          const h16x8 fx = stem_tr_read8(prow + boff[j], prow + boff[j] + 4 * 16);
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[a][j] = CVHIP_MFMA_16X16X32(fd[a], fx, acc[a][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // everybody is done reading before the next tile's data is stored
  }

  // ---- epilogue: lane holds D[k = a*16 + 4*(lane>>4) + r][column = f*16 + (lane & 15)]; column -> (tap, channel)
  const int Ktot = taps * 8;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = a * 16 + 4 * g + r;
      if (k >= p.K) continue;
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        const int col = (wave + 4 * j) * 16 + (lane & 15);
        if (col < Ktot) unsafeAtomicAdd(p.dw + (int64_t)k * Ktot + col, acc[a][j][r]);
      }
    }
}

// ---- host side ----------------------------------------------------------------------------------

static int stem_mode() {  // CVHIP_STEM: 0 = never, 1 = default
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CVHIP_STEM");
    v = e ? atoi(e) : 1;
  }
  return v;
}

// grid size of the stem kernel for a descriptor-level problem, 0 = not taken. Pure arithmetic (also sizes the BN partial rows).
int stem_blocks(int C, int x_ld, int K, int R, int S, int sh, int sw, int dh, int dw, int N, int OH, int OW) {
  if (stem_mode() == 0) return 0;
  if (C != 8 || x_ld != 8 || K > 32 || (K & 7) || R > 7 || S > 7 || sh != sw || (sh != 1 && sh != 2) || dh != 1 || dw != 1) return 0;
  const int nstep = (R * S + 3) / 4;
  if (nstep != 1 && nstep != 2 && nstep != 3 && nstep != 7 && nstep != 9 && nstep != 13) return 0;  // instantiated step counts
  const int64_t tiles = (int64_t)N * cdiv(OH, kStemTH) * cdiv(OW, kStemTW);
  if (tiles < 512 || tiles >= (1ll << 31)) return 0;  // small problems: the general kernel (and its tests) stay in charge
  const int cap = kStemBlocks;
  return (int)(tiles < cap ? tiles : cap);
}

template <int ST, bool STATS, bool EPI = false>
static int launch_stem_steps(const StemParams& sp, int blocks, int nstep, hipStream_t stream) {
#define CVHIP_STEM_CASE(NS)                                                                                        \
  case NS:                                                                                                         \
    hipLaunchKernelGGL((stem_fprop_kernel<ST, NS, STATS, EPI>), dim3(blocks), dim3(256), 0, stream, sp);           \
    break;
  switch (nstep) {
    CVHIP_STEM_CASE(1)
    CVHIP_STEM_CASE(2)
    CVHIP_STEM_CASE(3)   // 3x3
    CVHIP_STEM_CASE(7)   // 5x5
    CVHIP_STEM_CASE(9)   // 6x6
    CVHIP_STEM_CASE(13)  // 7x7
    default: return -1;
  }
#undef CVHIP_STEM_CASE
  return check_launch("stem_fprop_kernel");
}

// returns -1 when the problem is not taken (caller falls through to the general kernel)
int try_launch_stem(const IgemmParams& p, hipStream_t stream) {
  if (p.y2) return -1;
  if (p.ncls != 1 || p.out_sh != 1 || p.out_sw != 1) return -1;
  const IgemmClass& c = p.cls[0];
  if (c.out_oh != 0 || c.out_ow != 0 || c.OHi != p.OH || c.OWi != p.OW || c.dh0 > 0 || c.dw0 > 0) return -1;
  const int blocks = stem_blocks(p.Cin, p.x_ld, p.Nout, c.TR, c.TS, p.in_sh, p.in_sw, c.dh_step, c.dw_step, p.NB, p.OH, p.OW);
  if (blocks <= 0 || p.res || p.tail_y || (((uintptr_t)p.w) & 15)) return -1;
  if (p.x_image ? (p.x_planes < 1 || p.x_planes > 4 || (((uintptr_t)p.x_image) & 3)) : ((((uintptr_t)p.x) & 15) != 0)) return -1;
  const int nstep = (c.TR * c.TS + 3) / 4;
  StemParams sp;
  sp.x = p.x;
  sp.xf = p.x_image;
  sp.planes = p.x_planes;
  sp.w = p.w;
  sp.y = p.y;
  sp.bias = p.bias;
  sp.bias_n = p.bias_n;
  sp.stats = p.stats;
  sp.stats_acc = p.stats_acc;
  if (p.stats && (p.ep_scale || p.ep_act != CVHIP_ACT_NONE)) return CVHIP_ERR_INVALID;
  sp.ep_scale = p.ep_scale;
  sp.ep_shift = p.ep_shift;
  sp.ep_act = p.ep_act;
  sp.ep_ap = p.ep_ap;
  sp.NB = p.NB; sp.IH = p.IH; sp.IW = p.IW; sp.OH = p.OH; sp.OW = p.OW;
  sp.K = p.Nout; sp.y_ld = p.y_ld; sp.R = c.TR; sp.S = c.TS;
  sp.pad_h = -c.dh0; sp.pad_w = -c.dw0;
  sp.tiles_x = cdiv(p.OW, kStemTW);
  sp.tiles_y = cdiv(p.OH, kStemTH);
  sp.ntiles = p.NB * sp.tiles_x * sp.tiles_y;
  sp.xcd = stem_xcd_mode();
  int rc;
  if (sp.ep_scale || sp.ep_act != CVHIP_ACT_NONE)
    rc = p.in_sh == 2 ? launch_stem_steps<2, false, true>(sp, blocks, nstep, stream) : launch_stem_steps<1, false, true>(sp, blocks, nstep, stream);
  else if (p.in_sh == 2) rc = p.stats ? launch_stem_steps<2, true>(sp, blocks, nstep, stream) : launch_stem_steps<2, false>(sp, blocks, nstep, stream);
  else rc = p.stats ? launch_stem_steps<1, true>(sp, blocks, nstep, stream) : launch_stem_steps<1, false>(sp, blocks, nstep, stream);
  return rc;
}

// wgrad twin of try_launch_stem: same eligibility (descriptor level), dw = [K][R*S][8] fp32 (already zeroed / holding the sum)
int try_launch_stem_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream, const float* x_image, int x_planes,
                          const StemWgradBn* bn) {
  const int OH = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  const int OW = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  const int blocks = stem_blocks(d->C, d->x_ld, d->K, d->R, d->S, d->stride_h, d->stride_w, d->dil_h, d->dil_w, d->N, OH, OW);
  if (blocks <= 0 || (d->y_ld & 7) || (((uintptr_t)dy) & 15)) return -1;
  if (x_image ? (x_planes < 1 || x_planes > 4 || (((uintptr_t)x_image) & 3)) : ((((uintptr_t)x) & 15) != 0)) return -1;
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("CVHIP_STEM_WGRAD");
    off = (e && e[0] == '0') ? 1 : 0;
  }
  if (off) return -1;
  StemWgradParams sp;
  sp.x = (const h16_t*)x;
  sp.xf = x_image;
  sp.planes = x_planes;
  sp.dy = (const h16_t*)dy;
  sp.dw = dw;
  sp.NB = d->N; sp.IH = d->H; sp.IW = d->W; sp.OH = OH; sp.OW = OW;
  sp.K = d->K; sp.dy_ld = d->y_ld; sp.R = d->R; sp.S = d->S; sp.pad_h = d->pad_h; sp.pad_w = d->pad_w;
  sp.tiles_x = cdiv(OW, kStemTW);
  sp.tiles_y = cdiv(OH, kStemTH);
  sp.ntiles = d->N * sp.tiles_x * sp.tiles_y;
  sp.xcd = stem_xcd_mode();
  sp.y = nullptr;
  if (bn) {
    if (!bn->y || !bn->scale || !bn->shift || !bn->mean || !bn->invstd || !bn->acc || bn->acc_ld < d->K || (bn->y_ld & 7) || (((uintptr_t)bn->y) & 15)) return -1;
    sp.y = (const h16_t*)bn->y;
    sp.y_ld = bn->y_ld;
    sp.scale = bn->scale; sp.shift = bn->shift; sp.mean = bn->mean; sp.invstd = bn->invstd;
    sp.acc = bn->acc; sp.acc_ld = bn->acc_ld;
    sp.inv_count = 1.f / (float)((int64_t)d->N * OH * OW);
    sp.act = bn->act; sp.ap = bn->act_param;
    sp.o_dgamma = bn->dgamma_out; sp.o_dbeta = bn->dbeta_out; sp.accumulate = bn->accumulate;
  }
  const int nfrag = (d->R * d->S + 1) / 2;
  const int nfw = cdiv(nfrag, 4);
  const bool s2 = d->stride_h == 2;
  const bool bnb = bn != nullptr;
  int grid = blocks;
  if (bnb) {
    // the on-load instances hold dz AND y of the next tile in registers (~180 VGPRs): two resident blocks per CU, not three — a
    // persistent grid larger than the resident slots would run its last third after everything else. Measured on the YOLOv5-s stem
    // (profiles/r05_stem_bn_*): 768 blocks 423 us, 512 blocks 354 us, 256 blocks 577 us; forcing three blocks per CU
    // (__launch_bounds__(256, 3): 168 VGPRs + 68 B of scratch) 436 us — against 215 us (plain) + 217 us (the apply pass it replaces)
    const int cap = 512;
    if (grid > cap) grid = cap;
  }
#define CVHIP_STEMW_CASE(NF)                                                                                          \
  case NF:                                                                                                            \
    if (s2 && bnb) hipLaunchKernelGGL((stem_wgrad_kernel<2, NF, true>), dim3(grid), dim3(256), 0, stream, sp);        \
    else if (bnb) hipLaunchKernelGGL((stem_wgrad_kernel<1, NF, true>), dim3(grid), dim3(256), 0, stream, sp);         \
    else if (s2) hipLaunchKernelGGL((stem_wgrad_kernel<2, NF>), dim3(blocks), dim3(256), 0, stream, sp);              \
    else hipLaunchKernelGGL((stem_wgrad_kernel<1, NF>), dim3(blocks), dim3(256), 0, stream, sp);                      \
    break;
  switch (nfw) {
    CVHIP_STEMW_CASE(1)
    CVHIP_STEMW_CASE(2)  // 3x3: 5 fragments
    CVHIP_STEMW_CASE(4)  // 5x5: 13
    CVHIP_STEMW_CASE(5)  // 6x6: 18
    CVHIP_STEMW_CASE(7)  // 7x7: 25
    default: return -1;
  }
#undef CVHIP_STEMW_CASE
  return check_launch("stem_wgrad_kernel");
}

}  // namespace cvhip
