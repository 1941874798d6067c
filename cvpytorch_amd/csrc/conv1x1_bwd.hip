// conv1x1_bwd.hip — fused backward of a 1x1 / stride-1 Conv-BN-act layer for gfx950: ONE streaming kernel does what
// cvhip_bn_act_bwd_apply + cvhip_conv2d_dgrad + cvhip_conv2d_wgrad do in three passes.
//
//   dy[m][k]  = sc[k] * (dz[m][k] * act'(sc[k]*y[m][k] + sh[k])) + b1[k]*y[m][k] + c1[k]      (BN + activation backward,
//                                                                                               applied ON LOAD: dy never
//                                                                                               exists in HBM)
//   dx[m][c]  = sum_k dy[m][k] * W[k][c]        (+ addend[m][c]: gradient arriving over a skip connection)
//   dW[k][c] += sum_m dy[m][k] * x[m][c]        (fp32)
//
// The 1x1 layers of the detectors are HBM-bound; per layer the three-pass form moves (R dz, R y, W dy) + (R dy, W dx) +
// (R x, R dy) = 7 activation-sized tensors, this kernel moves (R dz, R y, R x, W dx) = 4. Structure (256 threads, persistent
// blocks, 64 pixel rows per trip):
//   * the dgrad weight image W^T [C][K] is staged into LDS once per block (rows permuted so that a lane ends up with 8
//     consecutive dx channels -> 16-byte stores, as in conv1x1_stream.hip);
//   * dz / y / x rows go global -> VGPR (16-byte vectors, next trip's loads in flight during this trip's MFMAs), the
//     BN/act derivative is evaluated in fp32 with per-thread channel constants, dy and x are written to row-major LDS tiles
//     with the 32-byte-segment XOR swizzle of conv_wgrad.hip;
//   * dgrad reads dy fragments with ds_read_b128 (pixel rows = MFMA B operand), wgrad reads dy^T and x^T fragments with the
//     hardware transpose read ds_read_b64_tr_b16; the dW accumulators stay in registers across ALL trips of the block and
//     are flushed once with fp32 atomics (K*C per block instead of K*C per 512..2048 rows).
// K (output channels of the layer = reduction of dgrad) is 32 / 64 / 128 (256 with C <= 128) per block; wider inputs C run as 128-wide column
// slices (grid.y), which re-read dz / y.
//
// Replaces aten::native_batch_norm_backward + silu_backward + convolution_backward reached from trainer.py:189
// (loss.backward()) for reference src/models/bricks/conv_module.py:201-214 layers with kernel_size 1.
#include <stdlib.h>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

struct Bwd1x1Params {
  const h16_t* dz0;  // gradient at the layer output, channels [0, k_split)
  const h16_t* dz1;  // channels [k_split, K) (sibling pairs deliver two tensors); unused when k_split == K
  int dz0_ld, dz1_ld, k_split;
  const h16_t* y;    // raw convolution output (pre-BN)
  int y_ld;
  const h16_t* y1;   // (round 5, split store) channels [k_split, K) of the raw output when they live in another buffer; NULL: inside y
  int y1_ld;
  const h16_t* x;    // layer input
  int x_ld;
  const h16_t* w;    // dgrad image [C][K]
  const h16_t* res;  // optional addend for dx
  int res_ld;
  h16_t* dx;
  int dx_ld;
  float* dw;          // [K][C] fp32, accumulated
  const float *scale, *shift, *mean, *invstd, *dgamma, *dbeta;
  float inv_count;
  int act;
  float ap;
  int M, K, C, ntiles;
  // accumulator form: (sum du, sum du*xhat) arrive as an fp64 accumulator [kAccShards][2][acc_ld] that every block folds in its
  // prologue (common.h acc_fold2) instead of as arrays prepared by a finalize launch; block (0, 0) stores the parameter gradients
  const double* acc;
  int acc_ld;
  float *o_dgamma, *o_dbeta;
  int accumulate;
  // "tail" (conv_plan.h IgemmCommon::tail_y): dx is the output gradient of the Conv-BN-act layer that produced x; fold that
  // layer's BatchNorm-backward sums into ITS accumulator from the stored dx rows and its y / statistics
  const h16_t* tail_y;
  int tail_y_ld;
  const float *tail_scale, *tail_shift, *tail_mean, *tail_invstd;
  int tail_act;
  float tail_ap;
  double* tail_acc;
  int tail_acc_ld;
  // XPRO instances (round 5, lazy activations): `x` is the RAW convolution output of the layer that produced this layer's input; the
  // weight gradient needs the ACTIVATED input, so the x rows are transformed on their way into the LDS tile,
  // x' = act(xs[c] * x + xh[c]) for input channels c in [x_lo, x_hi), rounded to 16 bits as the stand-alone pass would have stored them
  const float *xs, *xh;
  int x_act;
  float x_ap;
  int x_lo, x_hi;
};

typedef __attribute__((address_space(3))) h16x4 lds_h16x4_b;

__device__ __forceinline__ h16x8 tr_read8_b(const unsigned char* p0, const unsigned char* p1) {
  h16x4 lo = CVHIP_DS_READ_TR16_B64((lds_h16x4_b*)(p0));
  h16x4 hi = CVHIP_DS_READ_TR16_B64((lds_h16x4_b*)(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// byte offset of 16-B vector v of pixel row px inside a row-major tile whose 32-B segments are XOR-swizzled by the row
// (conv_wgrad.hip's layout: conflict-free for ds_read_b64_tr_b16 fragment gathers and for row-wise ds_read_b128)
template <int SEGM>
__device__ __forceinline__ int swz_off(int px, int v) {
  const int h = (px & 3) | ((px >> 1) & 4);
  return ((((v >> 1) ^ h) & SEGM) << 5) + (v & 1) * 16;
}

template <int ACT>
__device__ __forceinline__ f32x8 bnact_bwd8(const f32x8& dz, const f32x8& y, const float (&sc)[8], const float (&sh)[8],
                                            const float (&b1)[8], const float (&c1)[8], float ap) {
  f32x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float u = y.v[j] * sc[j] + sh[j];
    const float du = dz.v[j] * act_bwd(u, ACT, ap);
    o.v[j] = sc[j] * du + (b1[j] * y.v[j] + c1[j]);
  }
  return o;
}

// TAIL: the tail-sums form is its own instantiation — its 2 x CB/4 per-lane partial sums are live across the whole trip loop and
// pushed the 128 x 128 configuration to 104 spilled VGPRs (300 B/lane of scratch, 106 -> 195 us per launch) when it was a runtime flag
// KB 256 (round 4: the 64 -> 256 / 128 -> 256 expansion layers of ResNet bottlenecks, C <= 128 = ONE input-channel slice): 128 dW
// accumulator registers + 64 of dz / y prefetch per lane — one block per CU (launch bounds 1: up to 512 registers, AGPRs included)
template <int KB, int CB, bool TAIL, bool XPRO = false>
__global__ __launch_bounds__(256, KB >= 256 ? 1 : 2) void bwd1x1_kernel(const Bwd1x1Params p) {
  static_assert(!(TAIL && XPRO), "a lazy input has no tail form");
  constexpr int RT = 64;
  constexpr int KV = KB / 8, CV = CB / 8;
  constexpr int D_PASS = 256 / KV, D_IT = RT / D_PASS;
  constexpr int X_PASS = 256 / CV, X_IT = RT / X_PASS;
  constexpr int D_ROWB = KB * 2, X_ROWB = CB * 2;
  constexpr int D_SEGM = KB / 16 - 1, X_SEGM = CB / 16 - 1;
  constexpr int W_ROWB = KB * 2 + 16;  // (KB/2 + 4) banks = 4 * odd: 16 weight rows hit 64 distinct banks
  constexpr int W_BYTES = CB * W_ROWB, D_BYTES = RT * D_ROWB;  // + RT * X_ROWB for the x tile
  constexpr int NF = CB / 16;   // dx fragments across the input channels (per wave: 16 pixel rows x CB)
  constexpr int KS = KB / 32;   // dgrad reduction steps
  constexpr int KF = KB / 32;   // dW fragments of a wave along K (wave owns K/2 x C/2)
  constexpr int CF = CB / 32;
  static_assert(D_IT >= 1 && X_IT >= 1, "tile too narrow for 256 threads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // W_BYTES + D_BYTES + X_BYTES (up to 66 KB: dynamic)
  unsigned char* const sW = smem;
  unsigned char* const sD = smem + W_BYTES;
  unsigned char* const sX = sD + D_BYTES;
  float* const sT = reinterpret_cast<float*>(sX + RT * X_ROWB);  // tail: [4][CB] scale | shift | mean | invstd of channels c0 ..

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.y * CB;
  const int M = p.M;

  // ---- staging geometry (thread-constant channel vectors) -----------------------------------------------------------------
  const int kv = t % KV, drow = t / KV;
  const int xv = t % CV, xrow = t / CV;
  const bool seg1 = kv * 8 >= p.k_split;
  const h16_t* const dzp = seg1 ? p.dz1 + (kv * 8 - p.k_split) : p.dz0 + kv * 8;
  const int dz_ld = seg1 ? p.dz1_ld : p.dz0_ld;
  const bool ysplit = seg1 && p.y1 != nullptr;
  const h16_t* const yp = ysplit ? p.y1 + (kv * 8 - p.k_split) : p.y + kv * 8;
  const int yp_ld = ysplit ? p.y1_ld : p.y_ld;
  const h16_t* const xp = p.x + c0 + xv * 8;

  // per-channel constants of this thread's 8 channels: u = sc*y + sh; dy = sc*du + b1*y + c1
  float sc[8], sh[8], b1[8], c1[8];
  fill8c(1.f, sc);
  fill8c(0.f, sh);
  fill8c(0.f, b1);
  fill8c(0.f, c1);
  if (p.scale) {
    load8c(p.scale, kv * 8, p.K, sc);
    load8c(p.shift, kv * 8, p.K, sh);
  }
  if (p.acc) {  // block-uniform
    float* const kst = reinterpret_cast<float*>(sD);  // [2][KB]: the tile buffers are not in use yet
    if (t < p.K) {
      double s1, s2;
      acc_fold2(p.acc, p.acc_ld, t, s1, s2);
      kst[t] = (float)s1;
      kst[KB + t] = (float)s2;
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (p.o_dbeta) p.o_dbeta[t] = p.accumulate ? p.o_dbeta[t] + (float)s1 : (float)s1;
        if (p.o_dgamma) p.o_dgamma[t] = p.accumulate ? p.o_dgamma[t] + (float)s2 : (float)s2;
      }
    }
    __syncthreads();
  }
  if (p.mean) {
    float mu[8], is[8], k1[8], k2[8];
    load8c(p.mean, kv * 8, p.K, mu);
    load8c(p.invstd, kv * 8, p.K, is);
    if (p.acc) {
      const float* const kst = reinterpret_cast<const float*>(sD);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        k1[j] = kst[kv * 8 + j];
        k2[j] = kst[KB + kv * 8 + j];
      }
    } else {
      load8c(p.dbeta, kv * 8, p.K, k1);
      load8c(p.dgamma, kv * 8, p.K, k2);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float q2 = sc[j] * is[j] * (k2[j] * p.inv_count);
      b1[j] = -q2;
      c1[j] = q2 * mu[j] - sc[j] * (k1[j] * p.inv_count);
    }
  }

  if (p.acc) __syncthreads();  // everybody has its constants before the first store_tile overwrites the scratch

  // XPRO: constants of this thread's input-channel vector (thread-constant: xv); vectors outside [x_lo, x_hi) pass through
  float xsc[XPRO ? 8 : 1], xsh[XPRO ? 8 : 1];
  bool xlazy = false;
  if constexpr (XPRO) {
    const int xc = c0 + xv * 8;
    xlazy = xc >= p.x_lo && xc < p.x_hi;
    const int xcc = xlazy ? xc : p.x_lo;
    load8c(p.xs, xcc, p.C, xsc);
    load8c(p.xh, xcc, p.C, xsh);
  }

  uint4 rd[D_IT], ry[D_IT], rx[X_IT];
  // loads are unconditional (rows past M read row 0 and are zeroed when the tile is written): no load sits in a branch
  auto load_tile = [&](int tile) {
    const int m0 = tile * RT;
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      int m = m0 + i * D_PASS + drow;
      m = m < M ? m : 0;
      rd[i] = *reinterpret_cast<const uint4*>(dzp + (int64_t)m * dz_ld);
      ry[i] = *reinterpret_cast<const uint4*>(yp + (int64_t)m * yp_ld);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      int m = m0 + i * X_PASS + xrow;
      m = m < M ? m : 0;
      rx[i] = *reinterpret_cast<const uint4*>(xp + (int64_t)m * p.x_ld);
    }
  };
  auto store_tile = [&](int tile) {
    const int m0 = tile * RT;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int px = i * D_PASS + drow;
      const f32x8 dzv = unpack8(rd[i]), yv = unpack8(ry[i]);
      f32x8 o;
      switch (p.act) {  // block-uniform; hoisted out of the element loop
        case CVHIP_ACT_SILU: o = bnact_bwd8<CVHIP_ACT_SILU>(dzv, yv, sc, sh, b1, c1, p.ap); break;
        case CVHIP_ACT_RELU: o = bnact_bwd8<CVHIP_ACT_RELU>(dzv, yv, sc, sh, b1, c1, p.ap); break;
        case CVHIP_ACT_LEAKY: o = bnact_bwd8<CVHIP_ACT_LEAKY>(dzv, yv, sc, sh, b1, c1, p.ap); break;
        case CVHIP_ACT_SIGMOID: o = bnact_bwd8<CVHIP_ACT_SIGMOID>(dzv, yv, sc, sh, b1, c1, p.ap); break;
        case CVHIP_ACT_HSWISH: o = bnact_bwd8<CVHIP_ACT_HSWISH>(dzv, yv, sc, sh, b1, c1, p.ap); break;
        default: o = bnact_bwd8<CVHIP_ACT_NONE>(dzv, yv, sc, sh, b1, c1, p.ap); break;
      }
      uint4 v = pack8(o);
      if (m0 + px >= M) v = z;
      *reinterpret_cast<uint4*>(sD + px * D_ROWB + swz_off<D_SEGM>(px & 31, kv)) = v;
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int px = i * X_PASS + xrow;
      uint4 v = rx[i];
      if constexpr (XPRO) {
        if (xlazy) {
          f32x8 f = unpack8(v);
#pragma unroll
          for (int j = 0; j < 8; ++j) f.v[j] = f.v[j] * xsc[j] + xsh[j];
          switch (p.x_act) {  // block-uniform
            case CVHIP_ACT_SILU:
#pragma unroll
              for (int j = 0; j < 8; ++j) f.v[j] = act_fwd(f.v[j], CVHIP_ACT_SILU, p.x_ap);
              break;
            case CVHIP_ACT_RELU:
#pragma unroll
              for (int j = 0; j < 8; ++j) f.v[j] = act_fwd(f.v[j], CVHIP_ACT_RELU, p.x_ap);
              break;
            case CVHIP_ACT_LEAKY:
#pragma unroll
              for (int j = 0; j < 8; ++j) f.v[j] = act_fwd(f.v[j], CVHIP_ACT_LEAKY, p.x_ap);
              break;
            default: break;
          }
          v = pack8(f);
        }
      }
      if (m0 + px >= M) v = z;
      *reinterpret_cast<uint4*>(sX + px * X_ROWB + swz_off<X_SEGM>(px & 31, xv)) = v;
    }
  };

  constexpr bool tail = TAIL;
  if (tail && t < CB) {
    sT[t] = p.tail_scale[c0 + t];
    sT[CB + t] = p.tail_shift[c0 + t];
    sT[2 * CB + t] = p.tail_mean[c0 + t];
    sT[3 * CB + t] = p.tail_invstd[c0 + t];
  }
  constexpr int TSN = TAIL ? NF / 2 : 1;
  float ts1[TSN][8], ts2[TSN][8];  // this lane's share of the tail sums for channels c0 + j*32 + g*8 + e
#pragma unroll
  for (int j = 0; j < TSN; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) ts1[j][e] = ts2[j][e] = 0.f;

  int tile = blockIdx.x;
  load_tile(tile);

  // ---- dgrad weight tile -> LDS once: LDS row a*16 + i holds input channel c0 + (a>>1)*32 + (i>>2)*8 + (a&1)*4 + (i&3) -------
  {
    constexpr int NCH = CB * KV;  // 16-byte chunks
    for (int q = t; q < NCH; q += 256) {
      const int L = q / KV, kq = q - L * KV;
      const int a = L >> 4, i = L & 15;
      const int ch = c0 + (a >> 1) * 32 + (i >> 2) * 8 + (a & 1) * 4 + (i & 3);
      *reinterpret_cast<uint4*>(sW + L * W_ROWB + kq * 16) = *reinterpret_cast<const uint4*>(p.w + (int64_t)ch * p.K + kq * 8);
    }
  }

  f32x4 accw[KF][CF];
#pragma unroll
  for (int a = 0; a < KF; ++a)
#pragma unroll
    for (int b = 0; b < CF; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wk2 = wave >> 1, wc2 = wave & 1;
  // transpose-read geometry (conv_wgrad.hip): lane reads pixel rows 8g+q (+4), 8-byte column (lane&3) of the 32-B segment
  const int q4 = (lane >> 2) & 3;
  const int hsw = q4 | ((g & 1) << 2);
  const int px0 = 8 * g + q4;
  const int prow = wave * 16 + r;  // this lane's pixel row of the trip (dgrad B operand / dx row)
  const unsigned char* const wlane = sW + r * W_ROWB + g * 16;

  for (; tile < p.ntiles; tile += gridDim.x) {
    store_tile(tile);
    load_tile(tile + gridDim.x);  // next trip (past the end: clamped rows, never used)
    __syncthreads();

    // ---- dgrad: dx[16 rows of this wave][CB] -------------------------------------------------------------------------------
    f32x4 accx[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) accx[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const h16x8 fb = *reinterpret_cast<const h16x8*>(sD + prow * D_ROWB + swz_off<D_SEGM>(prow & 31, 4 * ks + g));
#pragma unroll
      for (int a = 0; a < NF; ++a) {
        const h16x8 wa = *reinterpret_cast<const h16x8*>(wlane + a * 16 * W_ROWB + ks * 64);
        accx[a] = CVHIP_MFMA_16X16X32(wa, fb, accx[a], 0, 0, 0);
      }
    }
    {
      const int m = tile * RT + prow;
      if (m < M) {
        h16_t* const drow_p = p.dx + (int64_t)m * p.dx_ld + c0 + g * 8;
#pragma unroll
        for (int j = 0; j < NF / 2; ++j) {
          f32x8 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v.v[e] = accx[2 * j][e];
            v.v[4 + e] = accx[2 * j + 1][e];
          }
          if (p.res) {
            const f32x8 rv = unpack8(*reinterpret_cast<const uint4*>(p.res + (int64_t)m * p.res_ld + c0 + j * 32 + g * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] += rv.v[e];
          }
          const uint4 packed = pack8(v);
          *reinterpret_cast<uint4*>(drow_p + j * 32) = packed;
          if constexpr (TAIL) {  // sums over the ROUNDED gradient, i.e. over what the tail layer's backward reads
            const f32x8 dzr = unpack8(packed);
            const f32x8 yv = unpack8(*reinterpret_cast<const uint4*>(p.tail_y + (int64_t)m * p.tail_y_ld + c0 + j * 32 + g * 8));
            const float* const k = sT + j * 32 + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float du = dzr.v[e] * act_bwd(yv.v[e] * k[e] + k[CB + e], p.tail_act, p.tail_ap);
              ts1[j][e] += du;
              ts2[j][e] += du * ((yv.v[e] - k[2 * CB + e]) * k[3 * CB + e]);
            }
          }
        }
      }
    }

    // ---- wgrad: dW[K/2 of this wave][C/2 of this wave] += dy^T x over the trip's 64 pixel rows -------------------------------
#pragma unroll
    for (int sub = 0; sub < RT / 32; ++sub) {
      h16x8 fd[KF], fx[CF];
#pragma unroll
      for (int a = 0; a < KF; ++a) {
        const int seg = (wk2 * (KB / 2) + a * 16) >> 4;
        const unsigned char* base = sD + sub * 32 * D_ROWB + (((seg ^ hsw) & D_SEGM) << 5) + (lane & 3) * 8;
        fd[a] = tr_read8_b(base + px0 * D_ROWB, base + (px0 + 4) * D_ROWB);
      }
#pragma unroll
      for (int b = 0; b < CF; ++b) {
        const int seg = (wc2 * (CB / 2) + b * 16) >> 4;
        const unsigned char* base = sX + sub * 32 * X_ROWB + (((seg ^ hsw) & X_SEGM) << 5) + (lane & 3) * 8;
        fx[b] = tr_read8_b(base + px0 * X_ROWB, base + (px0 + 4) * X_ROWB);
      }
#pragma unroll
      for (int a = 0; a < KF; ++a)
#pragma unroll
        for (int b = 0; b < CF; ++b) accw[a][b] = CVHIP_MFMA_16X16X32(fd[a], fx[b], accw[a][b], 0, 0, 0);
    }
    __syncthreads();  // everybody is done with the tiles before the next trip overwrites them
  }

  // ---- tail sums: channel c0 + j*32 + g*8 + e is shared by the 16 lanes of a DPP row and by the 4 waves ----------------------------
  if constexpr (TAIL) {
    float* const red = reinterpret_cast<float*>(sD);  // [4 waves][CB][2]; the loop's last barrier freed the tiles
#pragma unroll
    for (int j = 0; j < NF / 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float u1 = row16_sum(ts1[j][e]), u2 = row16_sum(ts2[j][e]);
        if (r == 0) {
          const int lc = j * 32 + g * 8 + e;
          red[(wave * CB + lc) * 2 + 0] = u1;
          red[(wave * CB + lc) * 2 + 1] = u2;
        }
      }
    __syncthreads();
    if (t < CB) {
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        u1 += red[(w * CB + t) * 2 + 0];
        u2 += red[(w * CB + t) * 2 + 1];
      }
      acc_add2(p.tail_acc, blockIdx.x, p.tail_acc_ld, c0 + t, u1, u2);
    }
  }

  // ---- flush dW: lane holds D[k = 4g + e][c = lane & 15] per fragment -----------------------------------------------------------
#pragma unroll
  for (int a = 0; a < KF; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = wk2 * (KB / 2) + a * 16 + 4 * g + e;
#pragma unroll
      for (int b = 0; b < CF; ++b) {
        const int c = c0 + wc2 * (CB / 2) + b * 16 + r;
        unsafeAtomicAdd(p.dw + ((int64_t)k * p.C + c), accw[a][b][e]);
      }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static int bwd1x1_mode() {  // CVHIP_BWD1X1: 0 = never (three-pass backward), 1 = when the geometry fits (default)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CVHIP_BWD1X1");
    v = e ? atoi(e) : 1;
  }
  return v;
}

static int bwd1x1_k256() { return 1; }   // K = 256 layers take the fused kernel (31.50 -> 31.02 ms per DeepLabv3+ step, profiles/r04_bwd1x1_k256_ab.log)

static int bwd1x1_cb(int C) { return (C % 128 == 0) ? 128 : (C % 64 == 0) ? 64 : (C % 32 == 0) ? 32 : 0; }

// structural fit: the kernel can run this layer geometry at all
static int bwd1x1_structural(const cvhip_conv_desc* d) {
  if (d->groups != 1 || d->R != 1 || d->S != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 0 || d->pad_w != 0) return 0;
  if (d->K != 32 && d->K != 64 && d->K != 128 && !(d->K == 256 && bwd1x1_k256() && d->C <= (bwd1x1_k256() >= 2 ? 256 : 128))) return 0;
  if (d->k_valid || d->c_valid) return 0;
  if (bwd1x1_cb(d->C) == 0 || d->C > 1024) return 0;
  if ((d->x_ld & 7) || (d->y_ld & 7)) return 0;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  if (M < 1 || M >= (1ll << 31) - 64 * 1024) return 0;
  return 1;
}

// policy: 1 when the fused kernel is also the faster form. Small layers (few 64-row trips per persistent block) expose the
// per-trip load -> LDS -> MFMA latency and the K*C-atomics flush of every block: measured on YOLOv5-s (gpurun conv_table,
// profiles/r02_*): 128->128 @40x40 (1600 trips) 65.8 us fused vs 60.6 us three-pass, everything from 3200 trips up wins 1.2-2x.
int bwd1x1_fits(const cvhip_conv_desc* d) {
  if (bwd1x1_mode() == 0 || !bwd1x1_structural(d)) return 0;
  if (bwd1x1_mode() >= 2) return 1;  // CVHIP_BWD1X1=2: whenever structurally possible (tests)
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const int64_t trips = ((M + 63) / 64) * (d->C / bwd1x1_cb(d->C));
  return trips >= 2400 ? 1 : 0;
}

template <int KB, int CB, bool TAIL, bool XPRO = false>
static int launch_b1t(const Bwd1x1Params& p, int blocks, hipStream_t s) {
  constexpr int LDS = CB * (KB * 2 + 16) + 64 * KB * 2 + 64 * CB * 2 + 4 * CB * (int)sizeof(float);
  auto kern = bwd1x1_kernel<KB, CB, TAIL, XPRO>;
  static bool attr_done[64] = {};  // per instantiation AND device: the attribute is a per-device property of the function
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& attr_set = attr_done[devid & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(bwd1x1_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(blocks, p.C / CB), dim3(256), LDS, s, p);
  return check_launch("bwd1x1_kernel");
}

template <int KB, int CB>
static int launch_b1(const Bwd1x1Params& p, int blocks, hipStream_t s) {
  if (p.xs) {
    if (p.tail_y) return CVHIP_ERR_UNSUPPORTED;
    if constexpr (KB <= 128) return launch_b1t<KB, CB, false, true>(p, blocks, s);
    return CVHIP_ERR_UNSUPPORTED;  // (the K = 256 instances sit at their register budget)
  }
  return p.tail_y ? launch_b1t<KB, CB, true>(p, blocks, s) : launch_b1t<KB, CB, false>(p, blocks, s);
}

int launch_bwd1x1(Bwd1x1Params& p, hipStream_t s) {
  p.ntiles = (p.M + 63) / 64;
  // persistent grid: <= 2 blocks per CU, >= 16 trips per block (the dW flush costs K*C atomics per block; A/B on the 80x80
  // layers: 16 trips 101 us, 8 trips 110 us, <= 256 blocks 122 us)
  const int max_blocks = 512, min_trips = 16;
  const int cb = bwd1x1_cb(p.C);
  const int slices = p.C / cb;
  int cap = max_blocks / slices;
  if (p.K >= 256 && cap > 256 / slices) cap = 256 / slices;  // one resident block per CU (register budget of the K = 256 instances)
  if (cap < 64) cap = 64;
  int blocks = p.ntiles / min_trips;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int rounds = cdiv(p.ntiles, blocks);
  blocks = cdiv(p.ntiles, rounds);
#define CVHIP_B1(KBV)                                        \
  if (cb == 128) return launch_b1<KBV, 128>(p, blocks, s);   \
  if (cb == 64) return launch_b1<KBV, 64>(p, blocks, s);     \
  return launch_b1<KBV, 32>(p, blocks, s);
  if (p.K == 256) { CVHIP_B1(256) }
  if (p.K == 128) { CVHIP_B1(128) }
  if (p.K == 64) { CVHIP_B1(64) }
  CVHIP_B1(32)
#undef CVHIP_B1
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_conv1x1_bwd_fused_ok(const cvhip_conv_desc* d) { return d ? bwd1x1_fits(d) : 0; }

static int bwd1x1_impl(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                       const void* y, const void* x, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* dgamma, const float* dbeta, const double* acc, int32_t acc_ld, float* o_dgamma,
                       float* o_dbeta, int32_t accumulate, int32_t act, float act_param, const void* addend, int32_t addend_ld, void* dx,
                       int32_t dx_ld, float* dw, void* stream, const cvhip_bn_tail* tail = nullptr, const cvhip_lazy_in* xin = nullptr,
                       const void* y1 = nullptr, int32_t y1_ld = 0) {
  if (!d || !dz0 || !y || !x || !w_dgrad || !dx || !dw) return CVHIP_ERR_INVALID;
  if (xin) {
    if (!xin->scale || !xin->shift) return CVHIP_ERR_INVALID;
    if (tail || d->K > 128) return CVHIP_ERR_UNSUPPORTED;
    if (xin->act != CVHIP_ACT_NONE && xin->act != CVHIP_ACT_RELU && xin->act != CVHIP_ACT_LEAKY && xin->act != CVHIP_ACT_SILU) return CVHIP_ERR_UNSUPPORTED;
  }
  if (!bwd1x1_structural(d)) return CVHIP_ERR_UNSUPPORTED;
  if (k_split <= 0 || k_split > d->K || (k_split & 7)) return CVHIP_ERR_INVALID;
  if (k_split < d->K && (!dz1 || (dz1_ld & 7) || (((uintptr_t)dz1) & 15))) return CVHIP_ERR_INVALID;
  if ((dz0_ld & 7) || (dx_ld & 7) || dx_ld < d->C) return CVHIP_ERR_INVALID;
  if ((((uintptr_t)dz0) | ((uintptr_t)y) | ((uintptr_t)x) | ((uintptr_t)w_dgrad) | ((uintptr_t)dx)) & 15) return CVHIP_ERR_INVALID;
  if (addend && ((addend_ld & 7) || addend_ld < d->C || (((uintptr_t)addend) & 15))) return CVHIP_ERR_INVALID;
  if ((scale == nullptr) != (shift == nullptr)) return CVHIP_ERR_INVALID;
  if (mean && (!invstd || !scale || (!acc && (!dgamma || !dbeta)))) return CVHIP_ERR_INVALID;
  if (acc && (!mean || acc_ld < d->K)) return CVHIP_ERR_INVALID;
  Bwd1x1Params p{};
  p.dz0 = (const h16_t*)dz0;
  p.dz1 = (const h16_t*)(k_split < d->K ? dz1 : dz0);
  p.dz0_ld = dz0_ld;
  p.dz1_ld = k_split < d->K ? dz1_ld : dz0_ld;
  p.k_split = k_split;
  p.y = (const h16_t*)y;
  p.y_ld = d->y_ld;
  p.y1 = nullptr;
  p.y1_ld = 0;
  if (y1) {
    if (k_split >= d->K || (y1_ld & 7) || (((uintptr_t)y1) & 15) || tail) return CVHIP_ERR_INVALID;
    p.y1 = (const h16_t*)y1;
    p.y1_ld = y1_ld;
  }
  p.x = (const h16_t*)x;
  p.x_ld = d->x_ld;
  p.w = (const h16_t*)w_dgrad;
  p.res = (const h16_t*)addend;
  p.res_ld = addend_ld;
  p.dx = (h16_t*)dx;
  p.dx_ld = dx_ld;
  p.dw = dw;
  p.scale = scale;
  p.shift = shift;
  p.mean = mean;
  p.invstd = invstd;
  p.dgamma = dgamma;
  p.dbeta = dbeta;
  p.M = d->N * d->H * d->W;
  p.K = d->K;
  p.C = d->C;
  p.inv_count = 1.f / (float)p.M;
  p.act = act;
  p.ap = act_param;
  p.acc = acc;
  p.acc_ld = acc_ld;
  p.o_dgamma = o_dgamma;
  p.o_dbeta = o_dbeta;
  p.accumulate = accumulate;
  if (xin) {
    p.xs = xin->scale;
    p.xh = xin->shift;
    p.x_act = xin->act;
    p.x_ap = xin->act_param;
    p.x_lo = xin->c_lo;
    p.x_hi = xin->c_hi > 0 ? xin->c_hi : d->C;
    if (p.x_lo < 0 || p.x_hi > d->C || p.x_lo >= p.x_hi || (p.x_lo & 7) || (p.x_hi & 7)) return CVHIP_ERR_INVALID;
  }
  p.tail_y = nullptr;
  if (tail) {
    if (!tail->y || !tail->scale || !tail->shift || !tail->mean || !tail->invstd || !tail->acc || tail->acc_ld < d->C) return CVHIP_ERR_INVALID;
    if ((tail->y_ld & 7) || (((uintptr_t)tail->y) & 15)) return CVHIP_ERR_UNSUPPORTED;
    if (tail->act != CVHIP_ACT_NONE && tail->act != CVHIP_ACT_RELU && tail->act != CVHIP_ACT_LEAKY && tail->act != CVHIP_ACT_SILU) return CVHIP_ERR_UNSUPPORTED;
    p.tail_y = (const h16_t*)tail->y;
    p.tail_y_ld = tail->y_ld;
    p.tail_scale = tail->scale;
    p.tail_shift = tail->shift;
    p.tail_mean = tail->mean;
    p.tail_invstd = tail->invstd;
    p.tail_act = tail->act;
    p.tail_ap = tail->act_param;
    p.tail_acc = tail->acc;
    p.tail_acc_ld = tail->acc_ld;
  }
  return launch_bwd1x1(p, (hipStream_t)stream);
}

int cvhip_conv1x1_bwd_fused(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                            const void* y, const void* x, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                            const float* invstd, const float* dgamma, const float* dbeta, int32_t act, float act_param,
                            const void* addend, int32_t addend_ld, void* dx, int32_t dx_ld, float* dw, void* stream) {
  return bwd1x1_impl(d, dz0, dz0_ld, dz1, dz1_ld, k_split, y, x, w_dgrad, scale, shift, mean, invstd, dgamma, dbeta, nullptr, 0, nullptr,
                     nullptr, 0, act, act_param, addend, addend_ld, dx, dx_ld, dw, stream);
}

int cvhip_conv1x1_bwd_fused_acc(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                const void* y, const void* x, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const double* acc, int32_t acc_ld, float* dgamma_out, float* dbeta_out, int32_t accumulate,
                                int32_t act, float act_param, const void* addend, int32_t addend_ld, void* dx, int32_t dx_ld, float* dw,
                                const cvhip_bn_tail* tail, void* stream) {
  if (!acc && mean) return CVHIP_ERR_INVALID;
  return bwd1x1_impl(d, dz0, dz0_ld, dz1, dz1_ld, k_split, y, x, w_dgrad, scale, shift, mean, invstd, nullptr, nullptr, acc, acc_ld, dgamma_out,
                     dbeta_out, accumulate, act, act_param, addend, addend_ld, dx, dx_ld, dw, stream, tail);
}

int cvhip_conv1x1_bwd_fused_split(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                  const void* y, const void* y1, int32_t y1_ld, const void* x, const void* w_dgrad, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, const double* acc, int32_t acc_ld,
                                  float* dgamma_out, float* dbeta_out, int32_t accumulate, int32_t act, float act_param, const void* addend,
                                  int32_t addend_ld, void* dx, int32_t dx_ld, float* dw, const cvhip_lazy_in* xin, void* stream) {
  if (!acc && mean) return CVHIP_ERR_INVALID;
  if (!y1) return CVHIP_ERR_INVALID;
  return bwd1x1_impl(d, dz0, dz0_ld, dz1, dz1_ld, k_split, y, x, w_dgrad, scale, shift, mean, invstd, nullptr, nullptr, acc, acc_ld, dgamma_out,
                     dbeta_out, accumulate, act, act_param, addend, addend_ld, dx, dx_ld, dw, stream, nullptr, xin, y1, y1_ld);
}

int cvhip_conv1x1_bwd_fused_lazy(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                 const void* y, const void* x_raw, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const double* acc, int32_t acc_ld, float* dgamma_out, float* dbeta_out, int32_t accumulate,
                                 int32_t act, float act_param, const void* addend, int32_t addend_ld, void* dx, int32_t dx_ld, float* dw,
                                 const cvhip_lazy_in* xin, void* stream) {
  if (!acc && mean) return CVHIP_ERR_INVALID;
  if (!xin) return CVHIP_ERR_INVALID;
  return bwd1x1_impl(d, dz0, dz0_ld, dz1, dz1_ld, k_split, y, x_raw, w_dgrad, scale, shift, mean, invstd, nullptr, nullptr, acc, acc_ld, dgamma_out,
                     dbeta_out, accumulate, act, act_param, addend, addend_ld, dx, dx_ld, dw, stream, nullptr, xin);
}

}  // extern "C"
