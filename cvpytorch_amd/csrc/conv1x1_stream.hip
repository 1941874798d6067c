// conv1x1_stream.hip — streaming 1x1 / stride-1 convolution (fprop and dgrad) for gfx950.
//
//   Out[m][n] = sum_c X[m][c] * Wt[n][c]            (m = pixel row, NHWC; dgrad: X = dy, Wt = W^T image)
//
// The 1x1 layers of the detectors are HBM-bound (arithmetic intensity C*K/(C+K) flop/B << the MFMA ridge): the general
// implicit-GEMM kernel stages A through LDS in 32-deep steps with a barrier each and starts a new block per 128/256 rows —
// prologue/epilogue latency, not bandwidth, sets its time (2-3 TB/s measured, gpurun conv_table). This kernel is built as a
// stream instead:
//   * the whole weight tile (BN x Cin, <= 72 KB) is staged into LDS ONCE per block; blocks are persistent (<= 2 per CU) and
//     walk the pixel rows with a grid stride, so there is no per-tile barrier at all;
//   * A fragments go global -> VGPR directly (a 1x1 conv needs no gather: lane (r, g) of v_mfma_f32_16x16x32_bf16 wants 8
//     consecutive channels of pixel row r = one 16-byte load; 16 rows x 64 contiguous bytes per instruction), register
//     double-buffered one pipeline stage (tile, 128-channel chunk) ahead of the MFMAs;
//   * the LDS weight rows are PERMUTED so that the two fragments of a pair leave every lane with 8 consecutive output channels
//     of its pixel -> one 16-byte store (64 contiguous bytes per pixel row per instruction);
//   * BatchNorm partial sums stay in registers across all tiles of the block and are written once (one partial row per
//     block: <= 512 rows for rows_reduce instead of M/256).
// Replaces the same aten::convolution / convolution_backward(input) calls as conv_igemm.hip (reference
// src/models/bricks/conv_module.py:209) for kernel_size 1, stride 1, padding 0, groups 1.
#include <stdlib.h>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

constexpr int kS1MaxLds = 72 * 1024;   // weight tile incl. row padding (+ 1 KB bias); two blocks per CU fit the 160 KB LDS
constexpr int kS1MaxBlocks = 512;      // persistent grid: 2 blocks per CU
constexpr int kS1MinTiles = 192;       // below this the grid cannot cover the chip: the general kernel's smaller tiles win

// STATS: 0 = none, 1 = BatchNorm sums of the outputs (fprop), 2 = "tail" (dgrad; conv_plan.h IgemmCommon::tail_y): BatchNorm-BACKWARD
// sums of the layer whose output gradient this launch produces, from the stored dz and that layer's y / statistics

// activation of 8 / 4 values with ONE switch (a switch per element multiplied the unrolled epilogue's code size and pushed the
// 256-wide streaming kernel's accumulators into scratch)
template <int NV>
__device__ __forceinline__ void s1_act_vec(float (&v)[NV], int act, float ap) {
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

// EPI: fused epilogue instance (STATS == 0 only): out = act((acc + bias) * ep_scale + ep_shift)
// PRO (round 5, lazy activations): x is the RAW convolution output of the producing Conv-BN-act layer(s); input channels
// [pro_lo, pro_hi) are transformed ON LOAD, z = act(pro_scale[c] * x + pro_shift[c]), in fp32 and rounded to 16 bits exactly as the
// stand-alone BN + activation pass (ew_kernel<0>) would have stored them — the pass, and the activated tensor, do not exist. The
// transform sits between the register prefetch and the MFMAs of a pipeline stage: the kernel is HBM-bound, its VALU is idle.
template <int NF, int MF, int STATS, bool EPI = false, bool PRO = false>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const IgemmKernArgs p, int ntiles, int vec16) {
  constexpr int BN = NF * 16;
  constexpr int RT = 64 * MF;   // pixel rows per block tile: 4 waves x MF fragments x 16
  constexpr int KC = 256 / MF;  // channels per pipeline stage (register budget: MF * KC/32 * 4 VGPRs per buffer)
  constexpr int KS = KC / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int Cin = p.Cin;
  const int M = p.cls[0].M;
  const int cin_pad = (Cin + 31) & ~31;
  const int brow = cin_pad * 2 + 16;  // bytes per LDS weight row; (cin_pad/2 + 4) banks = 4 * odd -> 16 rows hit 64 distinct banks
  const int nkc = (cin_pad + KC - 1) / KC;
  const int n0 = blockIdx.y * BN;

  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nkc;

  h16x8 buf0[MF][KS], buf1[MF][KS];

  // pipeline position of the NEXT load
  int ld_tile = blockIdx.x, ld_kc = 0;
  // Loads are UNCONDITIONAL: rows past M (and whole tiles past the block's last one) read row M-1, channel chunks past Cin read chunk
  // 0 — real, finite data whose products are dropped (rows: never stored, never summed) or meet the zero-padded weight columns
  // (chunks). With `ok ? load : 0` hipcc put every load in its own branch and, unable to count the younger prefetch loads, waited
  // s_waitcnt vmcnt(0) before the first MFMA of every stage: the next stage's loads never overlapped this stage's arithmetic.
  auto load = [&](h16x8 (&dst)[MF][KS]) {
    const int row0 = ld_tile * RT + wave * (MF * 16) + r;
    const int kbase = ld_kc * KC + g * 8;
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      int m = row0 + b * 16;
      m = m < M ? m : M - 1;
      const h16_t* src = p.x + (int64_t)m * p.x_ld;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        int kk = kbase + ks * 32;
        kk = kk < Cin ? kk : 0;
        dst[b][ks] = *reinterpret_cast<const h16x8*>(src + kk);
      }
    }
    if (++ld_kc == nkc) {
      ld_kc = 0;
      ld_tile += gridDim.x;
    }
  };

  load(buf0);   // (a block without tiles loads clamped rows and stores nothing)

  // ---- weight tile -> LDS, rows permuted: LDS row a*16 + i holds channel n0 + (a>>1)*32 + (i>>2)*8 + (a&1)*4 + (i&3)
  {
    const int cpr = cin_pad >> 3;  // 16-byte chunks per row
    const int nchunks = BN * cpr;
    constexpr int U = 8;  // loads in flight per thread (the tile is up to 72 KB = 18 chunks per thread)
    for (int q0 = t; q0 < nchunks; q0 += 256 * U) {
      uint4 v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u * 256;
        const int L = q / cpr;
        const int kq = q - L * cpr;
        const int a = L >> 4, i = L & 15;
        const int ch = n0 + (a >> 1) * 32 + (i >> 2) * 8 + (a & 1) * 4 + (i & 3);
        dst[u] = q < nchunks ? L * brow + kq * 16 : -1;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (q < nchunks && ch < p.Nout && kq * 8 < Cin) v[u] = *reinterpret_cast<const uint4*>(p.w + (int64_t)ch * Cin + kq * 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (dst[u] >= 0) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
    }
  }
  float* const sbias = reinterpret_cast<float*>(smem + BN * brow);  // [BN] fp32 behind the weight tile
  const float* const stail = sbias + BN;                            // [4][BN]: scale | shift | mean | invstd of the tail layer
  if (p.bias && t < BN) sbias[t] = (n0 + t < p.bias_n) ? p.bias[n0 + t] : 0.f;
  // fused epilogue: the constants share the tail rows
  static_assert(!EPI || STATS == 0, "the fused epilogue excludes BatchNorm sums");
  if (EPI && t < BN) {
    const int n = n0 + t < p.Nout ? n0 + t : p.Nout - 1;
    sbias[BN + t] = p.ep_scale ? p.ep_scale[n] : 1.f;
    sbias[2 * BN + t] = p.ep_scale ? p.ep_shift[n] : 0.f;
  }
  if (STATS == 2 && t < BN) {
    const int n = n0 + t < p.Nout ? n0 + t : p.Nout - 1;
    sbias[BN + t] = p.tail_scale[n];
    sbias[2 * BN + t] = p.tail_shift[n];
    sbias[3 * BN + t] = p.tail_mean[n];
    sbias[4 * BN + t] = p.tail_invstd[n];
  }
  float* const spro = sbias + 5 * BN;  // PRO: [2][cin_pad] scale | shift of the input channels, behind the tail rows
  if constexpr (PRO) {
    for (int c = t; c < cin_pad; c += 256) {
      const bool in = c >= p.pro_lo && c < p.pro_hi && c < Cin;
      spro[c] = in ? p.pro_scale[c] : 1.f;
      spro[cin_pad + c] = in ? p.pro_shift[c] : 0.f;
    }
  }
  __syncthreads();

  f32x4 acc[NF][MF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[STATS ? NF : 1][4], s2[STATS ? NF : 1][4];
  if constexpr (STATS) {
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) s1[a][q] = s2[a][q] = 0.f;
  }

  const unsigned char* const wlane = smem + r * brow + g * 16;
  int cp_tile = blockIdx.x, cp_kc = 0;  // pipeline position of the NEXT compute
  auto compute = [&](const h16x8 (&src)[MF][KS]) {
    const unsigned char* wk = wlane + cp_kc * (KC * 2);
    const int ksn = min(KS, (cin_pad - cp_kc * KC) >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < ksn) {
#pragma unroll
        for (int a = 0; a < NF; ++a) {
          const h16x8 wb = *reinterpret_cast<const h16x8*>(wk + a * 16 * brow + ks * 64);
#pragma unroll
          for (int b = 0; b < MF; ++b) acc[a][b] = CVHIP_MFMA_16X16X32(wb, src[b][ks], acc[a][b], 0, 0, 0);
        }
      }
    }
    if (++cp_kc < nkc) return;
    // ---- tile finished: store, fold into the BN sums, reset
    const int row0 = cp_tile * RT + wave * (MF * 16) + r;
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      const int m = row0 + b * 16;
      h16_t* yrow = p.y + (int64_t)m * p.y_ld;
#pragma unroll
      for (int j = 0; j < NF / 2; ++j) {
        const int ch0 = n0 + j * 32 + g * 8;
        f32x8 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v.v[q] = acc[2 * j][b][q];
          v.v[4 + q] = acc[2 * j + 1][b][q];
        }
        if constexpr (STATS == 1) {
          const float keep = m < M ? 1.f : 0.f;  // rows past M were computed from a clamped row: not part of the statistics
#pragma unroll
          for (int q = 0; q < 8; ++q) v.v[q] *= keep;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s1[2 * j][q] += v.v[q];
            s2[2 * j][q] += v.v[q] * v.v[q];
            s1[2 * j + 1][q] += v.v[4 + q];
            s2[2 * j + 1][q] += v.v[4 + q] * v.v[4 + q];
          }
        }
        if (p.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + j * 32 + g * 8);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + j * 32 + g * 8 + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v.v[q] += b0[q];
            v.v[4 + q] += b1[q];
          }
        }
        if constexpr (EPI) {
          const int cl8 = j * 32 + g * 8;
#pragma unroll
          for (int q = 0; q < 8; ++q) v.v[q] = v.v[q] * stail[cl8 + q] + stail[BN + cl8 + q];
          if (p.res && p.res_pre && m < M) {  // residual before the activation (ResNet bottleneck tail)
            const h16_t* rrow = p.res + (int64_t)m * p.res_ld + ch0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (ch0 + q < p.Nout) v.v[q] += (float)rrow[q];
          }
          s1_act_vec<8>(v.v, p.ep_act, p.ep_ap);
        }
        if (p.res && !(EPI && p.res_pre) && m < M) {  // skip-connection gradient folded into the epilogue
          const h16_t* rrow = p.res + (int64_t)m * p.res_ld + ch0;
          if (ch0 + 7 < p.Nout && (p.res_ld & 7) == 0 && ((((uintptr_t)p.res) & 15) == 0)) {
            const f32x8 rv = unpack8(*reinterpret_cast<const uint4*>(rrow));
#pragma unroll
            for (int q = 0; q < 8; ++q) v.v[q] += rv.v[q];
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (ch0 + q < p.Nout) v.v[q] += (float)rrow[q];
          }
        }
        if constexpr (STATS == 2) {
          if (m < M) {  // tail sums over the ROUNDED dz (what the tail layer's backward reads); the host guarantees the vector path
            const uint4 packed = pack8(v);
            const f32x8 dzr = unpack8(packed);
            const f32x8 yv = unpack8(*reinterpret_cast<const uint4*>(p.tail_y + (int64_t)m * p.tail_y_ld + ch0));
            const int cl8 = j * 32 + g * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float sc = stail[cl8 + q], sh = stail[BN + cl8 + q], mu = stail[2 * BN + cl8 + q], is = stail[3 * BN + cl8 + q];
              const float du = dzr.v[q] * act_bwd(yv.v[q] * sc + sh, p.tail_act, p.tail_ap);
              const float dx = du * ((yv.v[q] - mu) * is);
              if (q < 4) {
                s1[2 * j][q] += du;
                s2[2 * j][q] += dx;
              } else {
                s1[2 * j + 1][q - 4] += du;
                s2[2 * j + 1][q - 4] += dx;
              }
            }
          }
        }
        if (m < M) {
          // split store: the channel vectors from y_split on belong to the second destination (8-aligned split: a vector never straddles)
          h16_t* const dst = (p.y2 && ch0 >= p.y_split) ? p.y2 + (int64_t)m * p.y2_ld + (ch0 - p.y_split) : yrow + ch0;
          if (vec16 && ch0 + 7 < p.Nout) {
            *reinterpret_cast<uint4*>(dst) = pack8(v);
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (ch0 + q < p.Nout) dst[q] = (h16_t)v.v[q];
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    cp_kc = 0;
    cp_tile += gridDim.x;
  };

  // PRO: the stage that is about to be multiplied, transformed in its registers (same channel arithmetic as `load`: a chunk past Cin
  // holds chunk 0 again and meets zero weights). Channel vectors outside [pro_lo, pro_hi) — a concatenation's already-activated
  // slices — pass through untouched.
  auto xform = [&](h16x8 (&buf)[MF][KS]) {
    const int kbase = cp_kc * KC + g * 8;
    const int ksn = min(KS, (cin_pad - cp_kc * KC) >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      int kk = kbase + ks * 32;
      kk = kk < Cin ? kk : 0;
      if (ks < ksn && kk >= p.pro_lo && kk < p.pro_hi) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(spro + kk), c1 = *reinterpret_cast<const f32x4*>(spro + kk + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(spro + cin_pad + kk), h1 = *reinterpret_cast<const f32x4*>(spro + cin_pad + kk + 4);
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          f32x8 v = unpack8(__builtin_bit_cast(uint4, buf[b][ks]));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v.v[q] = v.v[q] * c0[q] + h0[q];
            v.v[4 + q] = v.v[4 + q] * c1[q] + h1[q];
          }
          s1_act_vec<8>(v.v, p.pro_act, p.pro_ap);
          buf[b][ks] = __builtin_bit_cast(h16x8, pack8(v));
        }
      }
    }
  };

  // branch-free body: a stage past the end loads clamped rows and computes a tile whose rows are all >= M (nothing stored or summed)
  for (int it = 0; it < total; it += 2) {
    load(buf1);
    if constexpr (PRO) xform(buf0);
    compute(buf0);
    load(buf0);
    if constexpr (PRO) xform(buf1);
    compute(buf1);
  }

  if constexpr (STATS) {
    if (p.stats) {
      __syncthreads();  // every wave is done with the weight tile
      float* red = reinterpret_cast<float*>(smem);  // [4 waves][BN][2]
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float u1 = row16_sum(s1[a][q]), u2 = row16_sum(s2[a][q]);
          if (r == 0) {
            const int lc = (a >> 1) * 32 + g * 8 + (a & 1) * 4 + q;
            red[(wave * BN + lc) * 2 + 0] = u1;
            red[(wave * BN + lc) * 2 + 1] = u2;
          }
        }
      __syncthreads();
      if (t < BN && n0 + t < p.Nout) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          u1 += red[(w * BN + t) * 2 + 0];
          u2 += red[(w * BN + t) * 2 + 1];
        }
        if (p.stats_acc) {
          acc_add2(reinterpret_cast<double*>(p.stats), blockIdx.x, p.stats_ld, n0 + t, u1, u2);
        } else {
          float* dst = p.stats + (int64_t)blockIdx.x * 2 * p.Nout;
          dst[n0 + t] = u1;
          dst[p.Nout + n0 + t] = u2;
        }
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------

static int s1x1_mode() {  // CVHIP_S1X1: 0 = never, 1 = when profitable (default), 2 = whenever structurally possible (tests)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CVHIP_S1X1");
    v = e ? atoi(e) : 1;
  }
  return v;
}

// fragments across the output channels: 32/64/128-wide tiles; 256-wide (detection-head convs: 255 channels, bias, no BN sums ->
// the registers the sums would take hold the second half of the accumulators) when the weight tile still fits
static int s1x1_nf(int Nout, int Cin, bool stats) {
  if (Nout <= 32) return 2;
  if (Nout <= 64) return 4;
  if (Nout > 128 && Nout <= 256 && !stats && 256 * (((Cin + 31) & ~31) * 2 + 16) <= kS1MaxLds) return 16;
  return 8;
}

// grid.x of the streaming kernel for this problem, 0 when the general kernel should run
int stream1x1_blocks(int Nout, int Cin, int64_t M, bool stats) {
  const int mode = s1x1_mode();
  if (mode == 0 || (Cin & 7) || M <= 0 || M >= (1ll << 31)) return 0;
  const int bn = s1x1_nf(Nout, Cin, stats) * 16;
  const int cin_pad = (Cin + 31) & ~31;
  if (bn * (cin_pad * 2 + 16) > kS1MaxLds) return 0;
  const int ntiles = (int)((M + 127) / 128);
  if (mode == 1) {
    if (ntiles < kS1MinTiles) return 0;
    // wider outputs run as several 128-wide column tiles (grid.y) that re-read the pixel rows: only worth it while those rows
    // stay in the Infinity Cache (<= 64 MB) and there are no BN sums (tools/s1x1_bench.py: -12..-29 % on the 40x40 YOLO
    // layers; with larger operands DeepLabv3+ lost 7 % end to end)
    if (Nout > bn && (Nout > 4 * bn || stats || (double)M * Cin * 2.0 > 64e6)) return 0;
  }
  const int max_blocks = kS1MaxBlocks;   // (768 / 1024-block grids: measured, slower — DESIGN.md 4.00)
  const int rounds = cdiv(ntiles, max_blocks);
  return cdiv(ntiles, rounds);  // balanced: every block walks `rounds` (or rounds-1) tiles
}

static bool s1x1_structural(const IgemmParams& p) {
  if (p.ncls != 1) return false;
  const IgemmClass& c = p.cls[0];
  return c.TR == 1 && c.TS == 1 && p.in_sh == 1 && p.in_sw == 1 && p.out_sh == 1 && p.out_sw == 1 && c.dh0 == 0 && c.dw0 == 0 &&
         c.out_oh == 0 && c.out_ow == 0 && c.OHi == p.OH && c.OWi == p.OW && p.IH == p.OH && p.IW == p.OW && (p.x_ld & 7) == 0;
}

// LDS bytes of a launch: weight tile + bias + the tail layer's 4 constant rows (+ the prologue's scale | shift rows)
static int s1x1_lds(int bn, int cin_pad, bool pro) {
  int lds = bn * (cin_pad * 2 + 16) + 5 * bn * (int)sizeof(float) + (pro ? 2 * cin_pad * (int)sizeof(float) : 0);
  if (lds < 4 * bn * 2 * (int)sizeof(float)) lds = 4 * bn * 2 * (int)sizeof(float);
  return lds;
}
constexpr int kS1MaxLdsPro = kS1MaxLds + 4096;  // what hipFuncSetAttribute grants every instance: a prologue must fit under it

template <int NF, int STATS, bool EPI = false, bool PRO = false>
static int launch_s1(const IgemmParams& p, int blocks, int ntiles, hipStream_t stream) {
  constexpr int MF = 2;
  const int bn = NF * 16;
  const int cin_pad = (p.Cin + 31) & ~31;
  const int lds = s1x1_lds(bn, cin_pad, PRO);
  if (lds > kS1MaxLdsPro) return CVHIP_ERR_UNSUPPORTED;
  auto kern = conv1x1_stream_kernel<NF, MF, STATS, EPI, PRO>;
  static bool attr_done[64] = {};  // per instantiation AND device: the attribute is a per-device property of the function
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& attr_set = attr_done[devid & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kS1MaxLds + 4096);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(conv1x1_stream_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int vec16 = ((p.y_ld & 7) == 0) && ((((uintptr_t)p.y) & 15) == 0);
  const IgemmKernArgs k = narrow_plan(p, 0, 1);
  hipLaunchKernelGGL(kern, dim3(blocks, cdiv(p.Nout, bn)), dim3(256), lds, stream, k, ntiles, vec16);
  return check_launch("conv1x1_stream_kernel");
}

// returns -1 when the problem is not taken (caller falls through to the general kernel)
// 1 when the streaming kernel runs this fprop problem AND can apply a prologue to its input (the plan of cvhip_conv2d_fprop[_acc/_fused])
bool stream1x1_prologue_ok(const IgemmParams& p, bool stats) {
  if (!s1x1_structural(p)) return false;
  if (stream1x1_blocks(p.Nout, p.Cin, p.cls[0].M, stats) <= 0) return false;
  const int nf = s1x1_nf(p.Nout, p.Cin, stats);
  if (nf == 16) return false;  // (the 256-wide detection-head instances have no prologue form)
  return s1x1_lds(nf * 16, (p.Cin + 31) & ~31, true) <= kS1MaxLdsPro;
}

int try_launch_stream1x1(const IgemmParams& p, hipStream_t stream) {
  if (!s1x1_structural(p)) return -1;
  const int64_t M = p.cls[0].M;
  const int blocks = stream1x1_blocks(p.Nout, p.Cin, M, p.stats != nullptr);
  if (blocks <= 0) return -1;
  const int ntiles = (int)((M + 127) / 128);
  const int nf = s1x1_nf(p.Nout, p.Cin, p.stats != nullptr);
  if (p.stats && (p.ep_scale || p.ep_act != CVHIP_ACT_NONE)) return CVHIP_ERR_INVALID;
  if (p.z_out) return CVHIP_ERR_UNSUPPORTED;  // (the activated side output exists in the patch-resident kernel only)
  if (p.y2 && ((p.y_split & 7) || p.y_split <= 0 || p.y_split >= p.Nout || (p.y2_ld & 7) || (((uintptr_t)p.y2) & 15) || (p.y_ld & 7) ||
               (((uintptr_t)p.y) & 15) || (p.Nout & 7)))
    return CVHIP_ERR_INVALID;
  if (p.pro_scale) {
    // lazy input: BN scale / shift + activation of the producing layer applied on load (training forms: raw output, optional BN sums)
    if (p.tail_y || p.ep_scale || p.ep_act != CVHIP_ACT_NONE || p.res || !stream1x1_prologue_ok(p, p.stats != nullptr)) return CVHIP_ERR_UNSUPPORTED;
    if (p.pro_lo < 0 || p.pro_hi > p.Cin || p.pro_lo >= p.pro_hi || (p.pro_lo & 7) || (p.pro_hi & 7)) return CVHIP_ERR_INVALID;
    if (p.stats) {
      if (nf == 2) return launch_s1<2, 1, false, true>(p, blocks, ntiles, stream);
      if (nf == 4) return launch_s1<4, 1, false, true>(p, blocks, ntiles, stream);
      return launch_s1<8, 1, false, true>(p, blocks, ntiles, stream);
    }
    if (nf == 2) return launch_s1<2, 0, false, true>(p, blocks, ntiles, stream);
    if (nf == 4) return launch_s1<4, 0, false, true>(p, blocks, ntiles, stream);
    return launch_s1<8, 0, false, true>(p, blocks, ntiles, stream);
  }
  if (p.stats && p.tail_y) {
    if (nf == 2) return launch_s1<2, 2>(p, blocks, ntiles, stream);
    if (nf == 4) return launch_s1<4, 2>(p, blocks, ntiles, stream);
    return launch_s1<8, 2>(p, blocks, ntiles, stream);
  }
  if (p.stats) {
    if (nf == 2) return launch_s1<2, 1>(p, blocks, ntiles, stream);
    if (nf == 4) return launch_s1<4, 1>(p, blocks, ntiles, stream);
    return launch_s1<8, 1>(p, blocks, ntiles, stream);
  }
  if (p.ep_scale || p.ep_act != CVHIP_ACT_NONE) {
    if (nf == 2) return launch_s1<2, 0, true>(p, blocks, ntiles, stream);
    if (nf == 4) return launch_s1<4, 0, true>(p, blocks, ntiles, stream);
    if (nf == 16) return launch_s1<16, 0, true>(p, blocks, ntiles, stream);
    return launch_s1<8, 0, true>(p, blocks, ntiles, stream);
  }
  if (nf == 2) return launch_s1<2, 0>(p, blocks, ntiles, stream);
  if (nf == 4) return launch_s1<4, 0>(p, blocks, ntiles, stream);
  if (nf == 16) return launch_s1<16, 0>(p, blocks, ntiles, stream);
  return launch_s1<8, 0>(p, blocks, ntiles, stream);
}

}  // namespace cvhip
