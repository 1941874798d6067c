// bn_act.hip — training-mode BatchNorm + activation, forward and backward, over NHWC bf16 matrices
// [M = N*H*W][C] with row pitch ld. All kernels are HBM-bound streaming passes: 16-B vector
// accesses along the channel axis, fp32 math, two-stage deterministic column reductions.
//
// Replaces aten::native_batch_norm(_backward) + aten::silu_/relu_ (+ residual add) reached from
// reference src/models/bricks/conv_module.py:211-213 and src/models/modules/yolo_modules.py:102.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace cvhip {

constexpr int kRedBlocksMax = 1024;
constexpr int kRedRowsPerBlockMin = 64;

static inline int colreduce_rows_host(int64_t M, int32_t C) {
  (void)C;
  const int min_rows = kRedRowsPerBlockMin;
  int64_t b = cdiv64(M, min_rows);
  if (b > kRedBlocksMax) b = kRedBlocksMax;
  if (b < 1) b = 1;
  return (int)b;
}

// MODE 0: (sum x, sum x^2)          MODE 1: (sum du, sum du*xhat)       MODE 2: (sum x, -)
// MODE 3 (round 5): the residual TAIL z = act(bn(y) + identity) of a ResNet bottleneck (torchvision Bottleneck.forward, reached through
// src/models/backbones/seg/resnet.py:91-94): du = dz * act'(z) from the saved OUTPUT z is computed, STORED (the identity branch's
// gradient) and reduced to the BatchNorm-backward sums (sum du, sum du*xhat) in ONE pass — read dz, read z, read y, write du — instead
// of an apply pass (read dz, read z, write du) followed by a reduction pass (read du, read y)
struct RedParams {
  const h16_t* a;   // x (mode 0/2) or dz (mode 1/3)
  const h16_t* y;   // conv output (mode 1/3)
  const h16_t* z;   // mode 3: the layer's saved output
  h16_t* du;        // mode 3: dz * act'(z), stored
  int ld_z, ld_du;
  int ld_a, ld_y;
  int64_t M;
  int C;
  const float *scale, *shift, *mean, *invstd;
  int act;
  float ap;
  float* partial;  // [gridDim.x][2][C]
  double* acc;     // or: fp64 accumulator [kAccShards][2][acc_ld] (atomics; common.h acc_add2) — no partial rows, no finalize launch
  int acc_ld;
};

// ACT: compile-time activation id for MODE 1 (a runtime switch inside the element loop compiles to a chain of scalar
// branches per element)
template <int MODE, int ACT = 0>
__global__ __launch_bounds__(256) void colreduce_kernel(const RedParams p) {
  __shared__ float red[256 * 16];
  const int t = threadIdx.x;
  const int CV = (p.C + 7) >> 3;  // 16-B column vectors (C % 8 == 0 on the fast path)
  const bool vec = (p.C & 7) == 0 && (p.ld_a & 7) == 0 && ((MODE != 1 && MODE != 3) || (p.ld_y & 7) == 0) &&
                   (MODE != 3 || ((p.ld_z & 7) == 0 && (p.ld_du & 7) == 0 && (((uintptr_t)p.z | (uintptr_t)p.du) & 15) == 0)) &&
                   (((uintptr_t)p.a | (uintptr_t)p.y) & 15) == 0;
  const int cols_per_pass = CV < 256 ? CV : 256;
  const int rows_per_pass = 256 / cols_per_pass;
  const int tx = t % cols_per_pass, ty = t / cols_per_pass;
  const bool active = ty < rows_per_pass;
  const int64_t rows_per_block = cdiv64(p.M, gridDim.x);
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > p.M) r_end = p.M;

  for (int cv0 = 0; cv0 < CV; cv0 += cols_per_pass) {
    const int cv = cv0 + tx;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    float sc[8], sh[8], mu[8], is[8];
    fill8c(1.f, sc);
    fill8c(0.f, sh);
    fill8c(0.f, mu);
    fill8c(1.f, is);
    if (MODE == 1 || MODE == 3) {
      const int cc = (cv < CV ? cv : CV - 1) * 8;
      if (p.scale) {  // uniform: scale/shift come together, mean/invstd come together
        load8c(p.scale, cc, p.C, sc);
        load8c(p.shift, cc, p.C, sh);
      }
      if (p.mean) {
        load8c(p.mean, cc, p.C, mu);
        load8c(p.invstd, cc, p.C, is);
      }
    }
    if (active && cv < CV) {
      auto accum = [&](const f32x8& a, const f32x8& y) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (MODE == 0) {
            s1[j] += a.v[j];
            s2[j] += a.v[j] * a.v[j];
          } else if (MODE == 2) {
            s1[j] += a.v[j];
          } else if (MODE == 3) {
            // a = du (already masked and rounded to 16 bits by the caller below): the BatchNorm branch has no activation of its own
            const float xh = (y.v[j] - mu[j]) * is[j];
            s1[j] += a.v[j];
            s2[j] += a.v[j] * xh;
          } else {
            const float u = y.v[j] * sc[j] + sh[j];
            const float du = a.v[j] * act_bwd(u, ACT, p.ap);
            const float xh = (y.v[j] - mu[j]) * is[j];
            s1[j] += du;
            s2[j] += du * xh;
          }
        }
      };
      int64_t r = r_begin + ty;
      if (vec) {
        // 4 rows per trip: 4-8 independent 16-B loads in flight per lane before any arithmetic (HBM latency hiding). The LAST trip
        // of a block is the same code with its missing rows masked (clamped address, zero contribution): a remainder walked one
        // row per iteration cost up to 3 extra dependent memory round trips per block — on the 26 MB layers (100 rows per block,
        // one 64-row trip + 36 rows of remainder) more than the trip itself (2.4 TB/s against 3.8 for the aligned shapes).
        // (Software-pipelining the trips — next trip's loads issued before this trip's arithmetic — was measured in round 3: +32
        // VGPRs, no gain on the backward passes: they are co-limited by the quarter-rate v_exp_f32 / v_rcp_f32 of the SiLU derivative.)
        const int64_t stp = rows_per_pass;
        for (; r < r_end; r += 4 * stp) {
          uint4 ua[4], uy[4], uz[MODE == 3 ? 4 : 1];
          bool ok[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t rq = r + q * stp;
            ok[q] = rq < r_end;
            const int64_t rc = ok[q] ? rq : r;
            ua[q] = *reinterpret_cast<const uint4*>(p.a + rc * p.ld_a + cv * 8);
            if (MODE == 1 || MODE == 3) uy[q] = *reinterpret_cast<const uint4*>(p.y + rc * p.ld_y + cv * 8);
            if constexpr (MODE == 3) uz[q] = *reinterpret_cast<const uint4*>(p.z + rc * p.ld_z + cv * 8);
          }
          // every load of the trip is issued before any arithmetic: without the fence hipcc's scheduler sinks the loads of rows
          // 1..3 below the arithmetic of row 0 (fewer live registers) and waits vmcnt(0) after each — one row in flight per lane
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (round 6) a row slot that is past the block's last row for EVERY lane of the wave contributes nothing: its arithmetic —
            // the SiLU derivative is ~40 VALU instructions per element, and with 1024 blocks the mid-size layers run 1.56 / 3.125
            // trips per lane, i.e. up to 22 % masked slots — is skipped under a wave-uniform branch. Adding +0 changed no sum before.
            if (__builtin_amdgcn_ballot_w64(ok[q]) == 0) continue;
            if (!ok[q]) ua[q] = make_uint4(0u, 0u, 0u, 0u);  // x = 0 / dz = 0: no contribution to either sum
            if constexpr (MODE == 3) {
              // du = dz * act'(z), rounded to 16 bits (what the stand-alone apply pass stores and the reduction pass then reads)
              f32x8 d = unpack8(ua[q]);
              const f32x8 zz = unpack8(uz[q]);
#pragma unroll
              for (int j = 0; j < 8; ++j) d.v[j] *= act_bwd(zz.v[j], ACT, p.ap);
              const uint4 packed = pack8(d);
              if (ok[q]) *reinterpret_cast<uint4*>(p.du + (r + q * stp) * p.ld_du + cv * 8) = packed;
              accum(unpack8(packed), unpack8(uy[q]));
            } else {
              accum(unpack8(ua[q]), MODE == 1 ? unpack8(uy[q]) : f32x8{});
            }
          }
        }
      }
      for (; r < r_end; r += rows_per_pass) {  // (scalar path only: the vector loop above leaves nothing)
        f32x8 a, y;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = cv * 8 + j;
          a.v[j] = c < p.C ? (float)p.a[r * p.ld_a + c] : 0.f;
          if (MODE == 1 || MODE == 3) y.v[j] = c < p.C ? (float)p.y[r * p.ld_y + c] : 0.f;
          if constexpr (MODE == 3) {
            if (c < p.C) {
              const h16_t dq = (h16_t)(a.v[j] * act_bwd((float)p.z[r * p.ld_z + c], ACT, p.ap));
              p.du[r * p.ld_du + c] = dq;
              a.v[j] = (float)dq;
            }
          }
        }
        accum(a, y);
      }
    }
    // reduce over ty through LDS
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[t * 16 + j] = s1[j];
      red[t * 16 + 8 + j] = s2[j];
    }
    __syncthreads();
    // thread (tx, j16) sums over ty
    for (int idx = t; idx < cols_per_pass * 16; idx += 256) {
      const int x = idx >> 4, j = idx & 15;
      float s = 0.f;
      for (int yy = 0; yy < rows_per_pass; ++yy) s += red[(yy * cols_per_pass + x) * 16 + j];
      const int c = (cv0 + x) * 8 + (j & 7);
      if (cv0 + x < CV && c < p.C) {
        if (p.acc) unsafeAtomicAdd(p.acc + ((size_t)((blockIdx.x & (kAccShards - 1)) * 2 + (j >> 3))) * p.acc_ld + c, (double)s);
        else p.partial[((int64_t)blockIdx.x * 2 + (j >> 3)) * p.C + c] = s;
      }
    }
  }
}

// ---- partial-row pre-reduction -------------------------------------------------------------------
// The conv epilogue emits one partial row per M tile (25,600 rows for the YOLOv5-s stem at batch 64) and
// the streaming reductions up to 1024: reducing those serially in a one-thread-per-channel finalize
// kernel costs hundreds of microseconds of pure latency. When rows > kStage2Rows a fully parallel
// pre-pass folds them into kStage2Rows rows first, written to the scratch rows that every partial
// buffer carries behind its payload (CVHIP_REDUCE_SCRATCH_ROWS). Deterministic (fixed partition).
constexpr int kStage2Rows = CVHIP_REDUCE_SCRATCH_ROWS;
constexpr int kDirectRows = 1024;  // up to here the 64-lane finalize kernels read the partial rows themselves

__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ in, int rows, int Wd, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int t = threadIdx.x;
  const int col = blockIdx.x * 64 + (t & 63), lane = t >> 6;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * chunk;
  const int r1 = min(rows, r0 + chunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < Wd) {
    int r = r0 + lane;
    for (; r + 12 < r1; r += 16) {
      a0 += in[(int64_t)r * Wd + col];
      a1 += in[(int64_t)(r + 4) * Wd + col];
      a2 += in[(int64_t)(r + 8) * Wd + col];
      a3 += in[(int64_t)(r + 12) * Wd + col];
    }
    for (; r < r1; r += 4) a0 += in[(int64_t)r * Wd + col];
  }
  red[lane][t & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (lane == 0 && col < Wd) out[(int64_t)blockIdx.y * Wd + col] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// returns the (possibly pre-reduced) partial pointer and updates *rows
static int direct_rows() { return kDirectRows; }

static const float* prereduce(const float* partial, int* rows, int Wd, hipStream_t s) {
  if (*rows <= direct_rows()) return partial;  // few enough rows for the 16-lane finalize kernels (4 independent chains per lane)
  float* scratch = const_cast<float*>(partial) + (int64_t)(*rows) * Wd;
  hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv(Wd, 64), kStage2Rows), dim3(256), 0, s, partial, *rows, Wd, scratch);
  *rows = kStage2Rows;
  return scratch;
}

// ---- finalize kernels (one thread per channel) -----------------------------------------------------
// 256 threads = CPB channels x LPC row-lanes (LPC = 16: CPB = 16, for <= 64 partial rows; LPC = 64: CPB = 4, for up to
// kDirectRows rows — every lane then sums <= 16 rows with independent loads, i.e. ~2 L2 round trips instead of a
// rows_reduce_kernel launch + a finalize launch). The lanes of a channel are folded through LDS.
template <int LPC>
__device__ __forceinline__ void foldN(const float* partial, int rows, int C, int c, int lane, double* s1, double* s2, bool want2,
                                      double (*red)[LPC][256 / LPC + 1]) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    // 4 independent accumulation chains: the loads of a lane are independent, only the adds chain
    double a1 = 0.0, a2 = 0.0, a3 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int r = lane;
    for (; r + 3 * LPC < rows; r += 4 * LPC) {
      a += (double)partial[((int64_t)r * 2 + 0) * C + c];
      a1 += (double)partial[((int64_t)(r + LPC) * 2 + 0) * C + c];
      a2 += (double)partial[((int64_t)(r + 2 * LPC) * 2 + 0) * C + c];
      a3 += (double)partial[((int64_t)(r + 3 * LPC) * 2 + 0) * C + c];
      if (want2) {
        b += (double)partial[((int64_t)r * 2 + 1) * C + c];
        b1 += (double)partial[((int64_t)(r + LPC) * 2 + 1) * C + c];
        b2 += (double)partial[((int64_t)(r + 2 * LPC) * 2 + 1) * C + c];
        b3 += (double)partial[((int64_t)(r + 3 * LPC) * 2 + 1) * C + c];
      }
    }
    for (; r < rows; r += LPC) {
      a += (double)partial[((int64_t)r * 2 + 0) * C + c];
      if (want2) b += (double)partial[((int64_t)r * 2 + 1) * C + c];
    }
    a = (a + a1) + (a2 + a3);
    b = (b + b1) + (b2 + b3);
  }
  constexpr int CPB = 256 / LPC;
  const int cc = threadIdx.x % CPB;
  red[0][lane][cc] = a;
  red[1][lane][cc] = b;
  __syncthreads();
  a = 0.0;
  b = 0.0;
  if (lane == 0) {
#pragma unroll 16
    for (int l = 0; l < LPC; ++l) {
      a += red[0][l][cc];
      b += red[1][l][cc];
    }
  }
  *s1 = a;
  *s2 = b;
}

template <int LPC>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int rows, int C, double count, const float* gamma,
                                                          const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                                          float* mean, float* invstd, float* scale, float* shift) {
  constexpr int CPB = 256 / LPC;
  __shared__ double red[2][LPC][CPB + 1];
  const int c = blockIdx.x * CPB + (threadIdx.x % CPB), lane = threadIdx.x / CPB;
  double s1, s2;
  foldN<LPC>(partial, rows, C, c, lane, &s1, &s2, true, red);
  if (lane != 0 || c >= C) return;
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  if (mean) mean[c] = (float)m;
  if (invstd) invstd[c] = is;
  const float sc = g * is;
  if (scale) scale[c] = sc;
  if (shift) shift[c] = b - (float)m * sc;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
  if (rvar) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_kernel(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rvar[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - rmean[c] * g * is;
}

// out0[c] = sum_r partial[r][0][c]; out1[c] = sum_r partial[r][1][c]; optionally acc0/acc1 += the same sums
// (accumulate != 0 makes out0/out1 accumulate as well)
template <int LPC>
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partial, int rows, int C, float* out0, float* out1, int accumulate,
                                                           float* acc0, float* acc1) {
  constexpr int CPB = 256 / LPC;
  __shared__ double red[2][LPC][CPB + 1];
  const int c = blockIdx.x * CPB + (threadIdx.x % CPB), lane = threadIdx.x / CPB;
  double s1, s2;
  foldN<LPC>(partial, rows, C, c, lane, &s1, &s2, out1 != nullptr || acc1 != nullptr, red);
  if (lane != 0 || c >= C) return;
  if (out0) out0[c] = accumulate ? out0[c] + (float)s1 : (float)s1;
  if (out1) out1[c] = accumulate ? out1[c] + (float)s2 : (float)s2;
  if (acc0) acc0[c] += (float)s1;
  if (acc1) acc1[c] += (float)s2;
}

// One launch instead of rows_reduce_kernel + a finalize kernel (2 x ~4.7 us of dependent launch latency per BN layer and
// pass: ~230 such pairs per YOLOv5-s step). Grid = (32-channel chunks, kStage2Rows row groups); every block folds its row
// group of its chunk's 64 columns (32 sums + 32 second sums) into the scratch rows, and the LAST block of a chunk to
// arrive (device-scope counter) reduces the chunk's scratch rows and runs the final per-channel math. Deterministic: the
// partition and both summation orders are fixed; only WHICH block does the tail varies.
// The counters are library-global: launches of this kernel must not overlap (they are issued on one stream).
// MEASURED AND LEFT OFF (enable with CVHIP_FUSED_FINALIZE=1): on MI355X the device-scope release/acquire pair the
// last-block hand-off needs (__threadfence => L2 write-back + invalidate across the 8 XCDs, whose L2s are not coherent
// with each other) costs ~20 us per launch — YOLOv5-s 3166 -> 2811 img/s, DeepLabv3+ 392 -> 330 img/s. Two dependent
// ~4.7 us launches are cheaper than one cross-XCD rendezvous.
struct FinParams {
  int mode;  // 0: BN forward statistics, 1: plain sums (out0/out1 [+= ], acc0/acc1 +=)
  double count;
  const float *gamma, *beta;
  float *rmean, *rvar;
  float momentum, eps;
  float *mean, *invstd, *scale, *shift;
  float *out0, *out1, *acc0, *acc1;
  int accumulate;
};
__device__ unsigned int g_fin_counters[256];

__global__ __launch_bounds__(256) void rows_reduce_fin_kernel(const float* __restrict__ in, int rows, int C, float* scratch, const FinParams fp) {
  __shared__ float red[4][64];
  __shared__ double red2[2][8][33];
  __shared__ int s_last;
  const int t = threadIdx.x;
  const int j = t & 63, lane = t >> 6;
  const int c0 = blockIdx.x * 32;
  const int cc = c0 + (j & 31);
  const int Wd = 2 * C;
  const int col = (j >> 5) * C + cc;  // j < 32: first sums, j >= 32: second sums of the same 32 channels
  const bool col_ok = cc < C;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * chunk;
  const int r1 = min(rows, r0 + chunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col_ok) {
    int r = r0 + lane;
    for (; r + 12 < r1; r += 16) {
      a0 += in[(int64_t)r * Wd + col];
      a1 += in[(int64_t)(r + 4) * Wd + col];
      a2 += in[(int64_t)(r + 8) * Wd + col];
      a3 += in[(int64_t)(r + 12) * Wd + col];
    }
    for (; r < r1; r += 4) a0 += in[(int64_t)r * Wd + col];
  }
  red[lane][j] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (lane == 0 && col_ok) scratch[(int64_t)blockIdx.y * Wd + col] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  __threadfence();
  __syncthreads();
  if (t == 0) {
    const unsigned old = atomicAdd(&g_fin_counters[blockIdx.x], 1u);
    const int last = (old == gridDim.y - 1);
    if (last) g_fin_counters[blockIdx.x] = 0u;  // self-cleaning for the next launch
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // tail: 32 channels x 8 row lanes over the gridDim.y scratch rows (written by other CUs: read past the L1)
  const volatile float* sc = scratch;
  const int ch = t & 31, rl = t >> 5;
  const int c = c0 + ch;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int r = rl; r < (int)gridDim.y; r += 8) {
      s1 += (double)sc[(int64_t)r * Wd + c];
      s2 += (double)sc[(int64_t)r * Wd + C + c];
    }
  red2[0][rl][ch] = s1;
  red2[1][rl][ch] = s2;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  s1 = 0.0;
  s2 = 0.0;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    s1 += red2[0][l][ch];
    s2 += red2[1][l][ch];
  }
  if (fp.mode == 0) {
    const double m = s1 / fp.count;
    double var = s2 / fp.count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)fp.eps));
    const float g = fp.gamma ? fp.gamma[c] : 1.f, b = fp.beta ? fp.beta[c] : 0.f;
    if (fp.mean) fp.mean[c] = (float)m;
    if (fp.invstd) fp.invstd[c] = is;
    const float scl = g * is;
    if (fp.scale) fp.scale[c] = scl;
    if (fp.shift) fp.shift[c] = b - (float)m * scl;
    if (fp.rmean) fp.rmean[c] = (1.f - fp.momentum) * fp.rmean[c] + fp.momentum * (float)m;
    if (fp.rvar) {
      const double unb = fp.count > 1.0 ? var * fp.count / (fp.count - 1.0) : var;
      fp.rvar[c] = (1.f - fp.momentum) * fp.rvar[c] + fp.momentum * (float)unb;
    }
  } else {
    if (fp.out0) fp.out0[c] = fp.accumulate ? fp.out0[c] + (float)s1 : (float)s1;
    if (fp.out1) fp.out1[c] = fp.accumulate ? fp.out1[c] + (float)s2 : (float)s2;
    if (fp.acc0) fp.acc0[c] += (float)s1;
    if (fp.acc1) fp.acc1[c] += (float)s2;
  }
}

// launch the fused form; false when it does not apply (few rows: the plain finalize kernels need no pre-reduction)
static bool launch_fused_finalize(const float* partial, int rows, int C, const FinParams& fp, hipStream_t s) {
  constexpr bool on = false;   // (pre-reduction + finalize in one launch: built in round 2, measured no faster than the two launches)
  if (!on || rows <= kStage2Rows || cdiv(C, 32) > 256) return false;
  float* scratch = const_cast<float*>(partial) + (int64_t)rows * 2 * C;
  hipLaunchKernelGGL(rows_reduce_fin_kernel, dim3(cdiv(C, 32), kStage2Rows), dim3(256), 0, s, partial, rows, C, scratch, fp);
  return true;
}

// ---- elementwise passes -----------------------------------------------------------------------------
struct EwParams {
  const h16_t *a, *y, *res;
  h16_t* out;
  int ld_a, ld_y, ld_res, ld_out;
  int64_t M;
  int C;
  const float *scale, *shift, *mean, *invstd, *dgamma, *dbeta;
  int act;
  float ap;
  float inv_count;
  int res_pre;  // MODE 0: residual is added BEFORE the activation (ResNet bottleneck: relu(bn(y) + identity))
  // ACC variants: the per-channel constants are derived in the block's prologue from an fp64 accumulator (common.h acc_fold2)
  // instead of being read from arrays a finalize launch prepared
  const double* acc;   // [kAccShards][2][acc_ld]: MODE 0 (sum y, sum y^2); MODE 1 (sum du, sum du*xhat)
  int acc_ld;
  double count;        // MODE 0: elements per channel
  const float *gamma, *beta;
  float *rmean, *rvar;
  float momentum, eps;
  float *o_mean, *o_invstd, *o_scale, *o_shift;   // MODE 0: block 0 stores the layer's statistics for backward
  float *o_dgamma, *o_dbeta;                       // MODE 1: block 0 stores (accumulate != 0: adds) the parameter gradients
  int accumulate;
  // RLZ instances (round 5, lazy activations): `res` is the RAW convolution output of the layer that produced the residual; the
  // pass applies that layer's BatchNorm scale / shift and the (same) activation to it on load: out = act(a*sc+sh) + act(res*rsc+rsh)
  const float *res_scale, *res_shift;
};
constexpr int kEwAccMaxC = 2048;

// rows a lane loads per trip (all issued before the trip's arithmetic). Round 4: 8 rows for the one-tensor forms (-DCVHIP_EW_RPT0=8)
// measured neutral in the step (14.17-14.20 vs 14.08-14.16 ms) and on rotating tensors (profiles/r04_ew_grid_ab.log): the passes
// are not short of bytes in flight, and more, shorter blocks (CVHIP_EW_ROWS=4/8) LOSE 0.1-0.6 ms to the per-block prologue
// (requesting the first trip's rows AHEAD of the accumulator fold of the ACC forms was built too: 122 -> 186 VGPRs, 14.25 vs 14.07-14.18 ms: no gain)
#ifndef CVHIP_EW_RPT0
#define CVHIP_EW_RPT0 4
#endif
template <int MODE, bool HAS_RES>
struct EW_RPT {
  static constexpr int value = (MODE == 1 || HAS_RES || MODE == 3) ? 4 : CVHIP_EW_RPT0;
};

// MODE 0: out = act(a*scale+shift) (+res)         [a = conv output y]
// MODE 1: out = dy from (a = dz, y)               [BN+act backward apply]
// MODE 2: out = a (copy)       MODE 3: out = a + res
// RLZ (MODE 0 only): the residual operand is a LAZY activation — see EwParams::res_scale
template <int MODE, int ACT = 0, bool ACC = false, bool RLZ = false>
__global__ __launch_bounds__(256) void ew_kernel(const EwParams p) {
  static_assert(!RLZ || MODE == 0, "lazy residuals exist in the forward apply pass only");
  __shared__ float cst[ACC ? 2 * kEwAccMaxC : 1];  // ACC: MODE 0 scale | shift; MODE 1 dbeta/M | dgamma/M
  if constexpr (ACC) {
    for (int c = threadIdx.x; c < p.C; c += 256) {
      double s1, s2;
      acc_fold2(p.acc, p.acc_ld, c, s1, s2);
      if (MODE == 0) {
        const double m = s1 / p.count;
        double var = s2 / p.count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)p.eps));
        const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
        const float sc = g * is, sh = b - (float)m * sc;
        cst[c] = sc;
        cst[kEwAccMaxC + c] = sh;
        if (blockIdx.x == 0) {  // the same arithmetic as bn_finalize_kernel
          p.o_mean[c] = (float)m;
          p.o_invstd[c] = is;
          p.o_scale[c] = sc;
          p.o_shift[c] = sh;
          if (p.rmean) p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)m;
          if (p.rvar) {
            const double unb = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
            p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
          }
        }
      } else {
        cst[c] = (float)s1 * p.inv_count;                 // dbeta / M
        cst[kEwAccMaxC + c] = (float)s2 * p.inv_count;    // dgamma / M
        if (blockIdx.x == 0) {
          if (p.o_dbeta) p.o_dbeta[c] = p.accumulate ? p.o_dbeta[c] + (float)s1 : (float)s1;
          if (p.o_dgamma) p.o_dgamma[c] = p.accumulate ? p.o_dgamma[c] + (float)s2 : (float)s2;
        }
      }
    }
    __syncthreads();
  }
  const int CV = (p.C + 7) >> 3;
  const bool vec = (p.C & 7) == 0 && (p.ld_a & 7) == 0 && (p.ld_out & 7) == 0 &&
                   (MODE != 1 || (p.ld_y & 7) == 0) && (!p.res || (p.ld_res & 7) == 0) &&
                   (((uintptr_t)p.a | (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.out) & 15) == 0;
  // thread owns one 16-B channel vector (tx) and walks rows: per-channel constants live in registers
  const int t = threadIdx.x;
  const int cols_per_pass = CV < 256 ? CV : 256;
  const int rows_per_pass = 256 / cols_per_pass;
  const int tx = t % cols_per_pass, ty = t / cols_per_pass;
  if (ty >= rows_per_pass) return;
  const int64_t rows_per_block = cdiv64(p.M, gridDim.x);
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > p.M) r_end = p.M;

  for (int cv = tx; cv < CV; cv += cols_per_pass) {
    const int c = cv * 8;
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
    fill8c(1.f, sc);
    fill8c(0.f, sh);
    fill8c(0.f, mu);
    fill8c(1.f, is);
    fill8c(0.f, k1);
    fill8c(0.f, k2);
    if (ACC && MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cc = c + j < p.C ? c + j : p.C - 1;
        sc[j] = cst[cc];
        sh[j] = cst[kEwAccMaxC + cc];
      }
    } else if (p.scale) {  // uniform branches: (scale, shift) and (mean, invstd, dgamma, dbeta) come as groups
      load8c(p.scale, c, p.C, sc);
      load8c(p.shift, c, p.C, sh);
    }
    if (MODE == 1 && p.mean) {
      load8c(p.mean, c, p.C, mu);
      load8c(p.invstd, c, p.C, is);
      if (ACC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cc = c + j < p.C ? c + j : p.C - 1;
          k1[j] = cst[cc];
          k2[j] = cst[kEwAccMaxC + cc];
        }
      } else {
        load8c(p.dbeta, c, p.C, k1);
        load8c(p.dgamma, c, p.C, k2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          k1[j] *= p.inv_count;
          k2[j] *= p.inv_count;
        }
      }
    }
    float rsc[RLZ ? 8 : 1], rsh[RLZ ? 8 : 1];
    if constexpr (RLZ) {
      load8c(p.res_scale, c, p.C, rsc);
      load8c(p.res_shift, c, p.C, rsh);
    }
    // the residual mode is block-uniform: one branch-free instance of the row loops per mode (with the runtime tests inside the
    // element loop hipcc emitted a branch per element and no packed fp32 math: ISA of ew_kernel<0, SILU>, round 3)
    auto rows = [&](auto rm) {
      constexpr int RES = decltype(rm)::value;  // 0: no residual, 1: added after the activation (MODE 3: the add itself), 2: before, 3: lazy, after
      auto math = [&](const f32x8& a, const f32x8& y, const f32x8& rs) -> f32x8 {
        f32x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (MODE == 0) {
            float u = a.v[j] * sc[j] + sh[j];
            if (RES == 2) u += rs.v[j];
            float v = act_fwd(u, ACT, p.ap);
            if (RES == 1) v += rs.v[j];
            if constexpr (RES == 3) v += (float)(h16_t)act_fwd(rs.v[j] * rsc[j] + rsh[j], ACT, p.ap);  // rounded as the stand-alone pass stores it
            o.v[j] = v;
          } else if (MODE == 1) {
            const float u = y.v[j] * sc[j] + sh[j];
            const float du = a.v[j] * act_bwd(u, ACT, p.ap);
            if (p.mean) {
              const float xh = (y.v[j] - mu[j]) * is[j];
              o.v[j] = sc[j] * (du - k1[j] - xh * k2[j]);
            } else {
              o.v[j] = sc[j] * du;
            }
          } else if (MODE == 2) {
            o.v[j] = a.v[j];
          } else {
            o.v[j] = a.v[j] + rs.v[j];
          }
        }
        return o;
      };
      constexpr bool has_res = RES != 0;
      int64_t r = r_begin + ty;
      if (vec) {
        // 4 rows per trip: all loads of the trip are issued before the arithmetic (bytes in flight per lane x4); the block's last
        // trip runs the same code with its missing rows masked (clamped loads, no store) instead of a row-at-a-time remainder
        const int64_t stp = rows_per_pass;
        constexpr int RPT = EW_RPT<MODE, has_res>::value;
        for (; r < r_end; r += RPT * stp) {
          uint4 ua[RPT], uy[RPT], ur[RPT];
          bool ok[RPT];
#pragma unroll
          for (int q = 0; q < RPT; ++q) {
            const int64_t rq = r + q * stp;
            ok[q] = rq < r_end;
            const int64_t rc = ok[q] ? rq : r;
            ua[q] = *reinterpret_cast<const uint4*>(p.a + rc * p.ld_a + c);
            if (MODE == 1) uy[q] = *reinterpret_cast<const uint4*>(p.y + rc * p.ld_y + c);
            if (has_res) ur[q] = *reinterpret_cast<const uint4*>(p.res + rc * p.ld_res + c);
          }
          __builtin_amdgcn_sched_barrier(0);  // loads of all rows of the trip first (see colreduce_kernel)
#pragma unroll
          for (int q = 0; q < RPT; ++q) {
            const f32x8 o = math(unpack8(ua[q]), MODE == 1 ? unpack8(uy[q]) : f32x8{}, has_res ? unpack8(ur[q]) : f32x8{});
            if (ok[q]) *reinterpret_cast<uint4*>(p.out + (r + q * stp) * p.ld_out + c) = pack8(o);
          }
        }
      }
      for (; r < r_end; r += rows_per_pass) {  // (scalar path only)
        f32x8 a, y, rs;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool okc = c + j < p.C;
          a.v[j] = okc ? (float)p.a[r * p.ld_a + c + j] : 0.f;
          if (MODE == 1) y.v[j] = okc ? (float)p.y[r * p.ld_y + c + j] : 0.f;
          if (has_res) rs.v[j] = okc ? (float)p.res[r * p.ld_res + c + j] : 0.f;
        }
        const f32x8 o = math(a, y, rs);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c + j < p.C) p.out[r * p.ld_out + c + j] = (h16_t)o.v[j];
      }
    };
    if constexpr (RLZ) {
      rows(std::integral_constant<int, 3>{});
    } else if constexpr (MODE == 3) {
      rows(std::integral_constant<int, 1>{});
    } else if constexpr (MODE == 0) {
      if (!p.res) rows(std::integral_constant<int, 0>{});
      else if (p.res_pre) rows(std::integral_constant<int, 2>{});
      else rows(std::integral_constant<int, 1>{});
    } else {
      rows(std::integral_constant<int, 0>{});
    }
  }
}

// Round 5 (lazy activations): a layer whose BN + activation apply pass is deferred into its consumers' loads still needs its
// statistics finalized — mean | invstd | scale | shift for backward and for the consumers' prologues, running statistics updated.
// The same arithmetic as block 0 of ew_kernel<0, ACT, true>'s prologue, as a launch of its own (C / 256 blocks).
__global__ __launch_bounds__(256) void bn_finalize_acc_kernel(const double* acc, int acc_ld, int C, double count, const float* gamma,
                                                              const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                                              float* o_mean, float* o_invstd, float* o_scale, float* o_shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1, s2;
  acc_fold2(acc, acc_ld, c, s1, s2);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * is, sh = b - (float)m * sc;
  o_mean[c] = (float)m;
  o_invstd[c] = is;
  o_scale[c] = sc;
  o_shift[c] = sh;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
  if (rvar) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// blocks are row chunks; a thread moves 16 B per row visit, so ~32 row-visits per thread keeps
// enough bytes in flight while leaving >> 256 blocks for large activations
// launch KERNEL<MODE, act> for the runtime activation id
#define CVHIP_LAUNCH_ACT_ACC(KERNEL, MODE, ACTV, GRID, STREAM, PARAMS)                                                         \
  switch (ACTV) {                                                                                                              \
    case CVHIP_ACT_RELU: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_RELU, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;        \
    case CVHIP_ACT_SILU: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_SILU, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;        \
    case CVHIP_ACT_LEAKY: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_LEAKY, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;      \
    case CVHIP_ACT_SIGMOID: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_SIGMOID, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;  \
    case CVHIP_ACT_HSWISH: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_HSWISH, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;    \
    default: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_NONE, true>), GRID, dim3(256), 0, STREAM, PARAMS); break;                   \
  }

#define CVHIP_LAUNCH_ACT(KERNEL, MODE, ACTV, GRID, STREAM, PARAMS)                                                       \
  switch (ACTV) {                                                                                                        \
    case CVHIP_ACT_RELU: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_RELU>), GRID, dim3(256), 0, STREAM, PARAMS); break;        \
    case CVHIP_ACT_SILU: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_SILU>), GRID, dim3(256), 0, STREAM, PARAMS); break;        \
    case CVHIP_ACT_LEAKY: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_LEAKY>), GRID, dim3(256), 0, STREAM, PARAMS); break;      \
    case CVHIP_ACT_SIGMOID: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_SIGMOID>), GRID, dim3(256), 0, STREAM, PARAMS); break;  \
    case CVHIP_ACT_HSWISH: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_HSWISH>), GRID, dim3(256), 0, STREAM, PARAMS); break;    \
    default: hipLaunchKernelGGL((KERNEL<MODE, CVHIP_ACT_NONE>), GRID, dim3(256), 0, STREAM, PARAMS); break;                   \
  }

static int ew_rows_per_thread() { return 16; }   // (4 / 8 rows per thread lose 0.1-0.6 ms per step to the per-block prologue: profiles/r04_ew_grid_ab.log)

static inline int ew_grid(int64_t M, int C) {
  const int CV = (C + 7) / 8;
  const int cols = CV < 256 ? CV : 256;
  const int rows_per_pass = 256 / cols;
  int64_t b = cdiv64(M, (int64_t)rows_per_pass * ew_rows_per_thread());
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

int cvhip_colreduce_rows(int64_t M, int32_t C) { return colreduce_rows_host(M, C); }

static int launch_red(int mode, RedParams& p, hipStream_t s) {
  if (!p.a || !p.partial || p.M < 0 || p.C <= 0) return CVHIP_ERR_INVALID;
  const int rows = colreduce_rows_host(p.M, p.C);
  if (mode == 0) hipLaunchKernelGGL(colreduce_kernel<0>, dim3(rows), dim3(256), 0, s, p);
  else if (mode == 1) {
    CVHIP_LAUNCH_ACT(colreduce_kernel, 1, p.act, dim3(rows), s, p)
  }
  else hipLaunchKernelGGL(colreduce_kernel<2>, dim3(rows), dim3(256), 0, s, p);
  return check_launch("colreduce_kernel");
}

int cvhip_bn_stats_partial(const void* x, int64_t M, int32_t C, int32_t ld, float* partial, void* stream) {
  RedParams p{};
  p.a = (const h16_t*)x;
  p.ld_a = ld;
  p.M = M;
  p.C = C;
  p.partial = partial;
  return launch_red(0, p, (hipStream_t)stream);
}

int cvhip_colsum_partial(const void* x, int64_t M, int32_t C, int32_t ld, float* partial, void* stream) {
  RedParams p{};
  p.a = (const h16_t*)x;
  p.ld_a = ld;
  p.M = M;
  p.C = C;
  p.partial = partial;
  return launch_red(2, p, (hipStream_t)stream);
}

int cvhip_bn_act_bwd_partial(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, int64_t M, int32_t C,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             int32_t act, float act_param, float* partial, void* stream) {
  if (!y) return CVHIP_ERR_INVALID;
  RedParams p{};
  p.a = (const h16_t*)dz;
  p.y = (const h16_t*)y;
  p.ld_a = ld_dz;
  p.ld_y = ld_y;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.mean = mean;
  p.invstd = invstd;
  p.act = act;
  p.ap = act_param;
  p.partial = partial;
  return launch_red(1, p, (hipStream_t)stream);
}

int cvhip_bn_finalize(const float* partial, int32_t rows, int32_t C, int64_t count, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      float* mean, float* invstd, float* scale, float* shift, void* stream) {
  if (!partial || rows <= 0 || C <= 0 || count <= 0) return CVHIP_ERR_INVALID;
  {
    FinParams fp{};
    fp.mode = 0;
    fp.count = (double)count;
    fp.gamma = gamma; fp.beta = beta; fp.rmean = running_mean; fp.rvar = running_var;
    fp.momentum = momentum; fp.eps = eps;
    fp.mean = mean; fp.invstd = invstd; fp.scale = scale; fp.shift = shift;
    if (launch_fused_finalize(partial, rows, C, fp, (hipStream_t)stream)) return check_launch("rows_reduce_fin_kernel");
  }
  partial = prereduce(partial, &rows, 2 * C, (hipStream_t)stream);
  if (rows > kStage2Rows)
    hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  return check_launch("bn_finalize_kernel");
}

int cvhip_bn_eval_scale_shift(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, float eps, float* scale, float* shift, void* stream) {
  if (C <= 0 || !running_mean || !running_var || !scale || !shift) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(bn_eval_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  return check_launch("bn_eval_kernel");
}

int cvhip_bn_bwd_finalize(const float* partial, int32_t rows, int32_t C, float* dgamma, float* dbeta, float* acc_dgamma,
                          float* acc_dbeta, void* stream) {
  if (!partial || rows <= 0 || C <= 0) return CVHIP_ERR_INVALID;
  {
    FinParams fp{};
    fp.mode = 1;
    fp.out0 = dbeta; fp.out1 = dgamma; fp.acc0 = acc_dbeta; fp.acc1 = acc_dgamma; fp.accumulate = 0;
    if (launch_fused_finalize(partial, rows, C, fp, (hipStream_t)stream)) return check_launch("rows_reduce_fin_kernel");
  }
  partial = prereduce(partial, &rows, 2 * C, (hipStream_t)stream);
  if (rows > kStage2Rows)
    hipLaunchKernelGGL(sum_partials_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       dbeta, dgamma, 0, acc_dbeta, acc_dgamma);
  else
    hipLaunchKernelGGL(sum_partials_kernel<16>, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       dbeta, dgamma, 0, acc_dbeta, acc_dgamma);
  return check_launch("sum_partials_kernel");
}

int cvhip_colsum_finalize(const float* partial, int32_t rows, int32_t C, float* out, int accumulate, void* stream) {
  if (!partial || rows <= 0 || C <= 0 || !out) return CVHIP_ERR_INVALID;
  partial = prereduce(partial, &rows, 2 * C, (hipStream_t)stream);
  if (rows > kStage2Rows)
    hipLaunchKernelGGL(sum_partials_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       out, (float*)nullptr, accumulate, (float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL(sum_partials_kernel<16>, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, C,
                       out, (float*)nullptr, accumulate, (float*)nullptr, (float*)nullptr);
  return check_launch("sum_partials_kernel");
}

int cvhip_bn_act_fwd(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const float* scale,
                     const float* shift, int32_t act, float act_param, const void* residual, int32_t ld_res,
                     void* stream) {
  if (!y || !z || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)y;
  p.ld_a = ld_y;
  p.out = (h16_t*)z;
  p.ld_out = ld_z;
  p.res = (const h16_t*)residual;
  p.ld_res = ld_res;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.act = act;
  p.ap = act_param;
  CVHIP_LAUNCH_ACT(ew_kernel, 0, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<0>");
}

int cvhip_bn_add_act_fwd(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const float* scale,
                         const float* shift, int32_t act, float act_param, const void* residual, int32_t ld_res, void* stream) {
  if (!y || !z || !residual || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)y;
  p.ld_a = ld_y;
  p.out = (h16_t*)z;
  p.ld_out = ld_z;
  p.res = (const h16_t*)residual;
  p.ld_res = ld_res;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.act = act;
  p.ap = act_param;
  p.res_pre = 1;  // the residual joins BEFORE the activation
  CVHIP_LAUNCH_ACT(ew_kernel, 0, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<0>(bn_add_act)");
}

int cvhip_bn_act_bwd_apply(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, void* dy, int32_t ld_dy,
                           int64_t M, int32_t C, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* dgamma, const float* dbeta, int32_t act,
                           float act_param, void* stream) {
  if (!dz || !y || !dy || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (mean && (!invstd || !dgamma || !dbeta)) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)dz;
  p.ld_a = ld_dz;
  p.y = (const h16_t*)y;
  p.ld_y = ld_y;
  p.out = (h16_t*)dy;
  p.ld_out = ld_dy;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.mean = mean;
  p.invstd = invstd;
  p.dgamma = dgamma;
  p.dbeta = dbeta;
  p.act = act;
  p.ap = act_param;
  p.inv_count = 1.f / (float)M;
  CVHIP_LAUNCH_ACT(ew_kernel, 1, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<1>");
}

int cvhip_add_act_fwd(const void* a, int32_t ld_a, const void* b, int32_t ld_b, void* out, int32_t ld_out, int64_t M, int32_t C,
                      int32_t act, float act_param, void* stream) {
  if (!a || !b || !out || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)a;
  p.ld_a = ld_a;
  p.res = (const h16_t*)b;
  p.ld_res = ld_b;
  p.out = (h16_t*)out;
  p.ld_out = ld_out;
  p.M = M;
  p.C = C;
  p.act = act;
  p.ap = act_param;
  p.res_pre = 1;
  CVHIP_LAUNCH_ACT(ew_kernel, 0, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<0>(add_act)");
}

int cvhip_copy2d(const void* src, int32_t ld_src, void* dst, int32_t ld_dst, int64_t M, int32_t C, void* stream) {
  if (!src || !dst || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)src;
  p.ld_a = ld_src;
  p.out = (h16_t*)dst;
  p.ld_out = ld_dst;
  p.M = M;
  p.C = C;
  hipLaunchKernelGGL(ew_kernel<2>, dim3(ew_grid(M, C)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("ew_kernel<2>");
}

int cvhip_add2d(const void* a, int32_t ld_a, const void* b, int32_t ld_b, void* dst, int32_t ld_dst, int64_t M,
                int32_t C, void* stream) {
  if (!a || !b || !dst || M < 0 || C <= 0) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  EwParams p{};
  p.a = (const h16_t*)a;
  p.ld_a = ld_a;
  p.res = (const h16_t*)b;
  p.ld_res = ld_b;
  p.out = (h16_t*)dst;
  p.ld_out = ld_dst;
  p.M = M;
  p.C = C;
  hipLaunchKernelGGL(ew_kernel<3>, dim3(ew_grid(M, C)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("ew_kernel<3>");
}

int cvhip_bn_acc_shards(void) { return kAccShards; }

int cvhip_bn_act_bwd_sums_acc(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, int64_t M, int32_t C, const float* scale,
                              const float* shift, const float* mean, const float* invstd, int32_t act, float act_param, double* acc,
                              int32_t acc_ld, void* stream) {
  if (!dz || !y || !acc || M < 0 || C <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  if (M == 0) return CVHIP_OK;
  RedParams p{};
  p.a = (const h16_t*)dz;
  p.y = (const h16_t*)y;
  p.ld_a = ld_dz;
  p.ld_y = ld_y;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.mean = mean;
  p.invstd = invstd;
  p.act = act;
  p.ap = act_param;
  p.acc = acc;
  p.acc_ld = acc_ld;
  const int rows = colreduce_rows_host(M, C);
  CVHIP_LAUNCH_ACT(colreduce_kernel, 1, act, dim3(rows), (hipStream_t)stream, p)
  return check_launch("colreduce_kernel(acc)");
}

int cvhip_bn_tail_bwd_sums_acc(const void* dz, int32_t ld_dz, const void* z_out, int32_t ld_z, const void* y, int32_t ld_y, void* du, int32_t ld_du,
                               int64_t M, int32_t C, const float* mean, const float* invstd, int32_t act, float act_param, double* acc,
                               int32_t acc_ld, void* stream) {
  if (!dz || !z_out || !y || !du || !mean || !invstd || !acc || M < 0 || C <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  if (act != CVHIP_ACT_NONE && act != CVHIP_ACT_RELU && act != CVHIP_ACT_LEAKY) return CVHIP_ERR_UNSUPPORTED;  // act'(z) from the OUTPUT's sign
  if (M == 0) return CVHIP_OK;
  RedParams p{};
  p.a = (const h16_t*)dz;
  p.ld_a = ld_dz;
  p.y = (const h16_t*)y;
  p.ld_y = ld_y;
  p.z = (const h16_t*)z_out;
  p.ld_z = ld_z;
  p.du = (h16_t*)du;
  p.ld_du = ld_du;
  p.M = M;
  p.C = C;
  p.mean = mean;
  p.invstd = invstd;
  p.act = act;
  p.ap = act_param;
  p.acc = acc;
  p.acc_ld = acc_ld;
  const int rows = colreduce_rows_host(M, C);
  switch (act) {
    case CVHIP_ACT_RELU: hipLaunchKernelGGL((colreduce_kernel<3, CVHIP_ACT_RELU>), dim3(rows), dim3(256), 0, (hipStream_t)stream, p); break;
    case CVHIP_ACT_LEAKY: hipLaunchKernelGGL((colreduce_kernel<3, CVHIP_ACT_LEAKY>), dim3(rows), dim3(256), 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL((colreduce_kernel<3, CVHIP_ACT_NONE>), dim3(rows), dim3(256), 0, (hipStream_t)stream, p); break;
  }
  return check_launch("colreduce_kernel<3>(acc)");
}

int cvhip_bn_act_fwd_acc(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const double* acc, int32_t acc_ld,
                         int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, float* mean, float* invstd, float* scale, float* shift, int32_t act, float act_param,
                         const void* residual, int32_t ld_res, int32_t res_pre, void* stream) {
  if (!y || !z || !acc || !mean || !invstd || !scale || !shift || M <= 0 || C <= 0 || count <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  if (C > kEwAccMaxC) return CVHIP_ERR_UNSUPPORTED;
  EwParams p{};
  p.a = (const h16_t*)y;
  p.ld_a = ld_y;
  p.out = (h16_t*)z;
  p.ld_out = ld_z;
  p.res = (const h16_t*)residual;
  p.ld_res = ld_res;
  p.res_pre = residual ? res_pre : 0;
  p.M = M;
  p.C = C;
  p.act = act;
  p.ap = act_param;
  p.acc = acc;
  p.acc_ld = acc_ld;
  p.count = (double)count;
  p.gamma = gamma;
  p.beta = beta;
  p.rmean = running_mean;
  p.rvar = running_var;
  p.momentum = momentum;
  p.eps = eps;
  p.o_mean = mean;
  p.o_invstd = invstd;
  p.o_scale = scale;
  p.o_shift = shift;
  CVHIP_LAUNCH_ACT_ACC(ew_kernel, 0, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<0,acc>");
}

int cvhip_bn_finalize_acc(const double* acc, int32_t acc_ld, int32_t C, int64_t count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                          float* shift, void* stream) {
  if (!acc || !mean || !invstd || !scale || !shift || C <= 0 || count <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(bn_finalize_acc_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, acc, acc_ld, C, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  return check_launch("bn_finalize_acc_kernel");
}

int cvhip_bn_act_fwd_acc_lazyres(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const double* acc, int32_t acc_ld,
                                 int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int32_t act,
                                 float act_param, const void* residual_raw, int32_t ld_res, const float* res_scale, const float* res_shift,
                                 void* stream) {
  if (!y || !z || !acc || !mean || !invstd || !scale || !shift || M <= 0 || C <= 0 || count <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  if (!residual_raw || !res_scale || !res_shift) return CVHIP_ERR_INVALID;
  if (C > kEwAccMaxC) return CVHIP_ERR_UNSUPPORTED;
  EwParams p{};
  p.a = (const h16_t*)y;
  p.ld_a = ld_y;
  p.out = (h16_t*)z;
  p.ld_out = ld_z;
  p.res = (const h16_t*)residual_raw;
  p.ld_res = ld_res;
  p.res_scale = res_scale;
  p.res_shift = res_shift;
  p.M = M;
  p.C = C;
  p.act = act;
  p.ap = act_param;
  p.acc = acc;
  p.acc_ld = acc_ld;
  p.count = (double)count;
  p.gamma = gamma;
  p.beta = beta;
  p.rmean = running_mean;
  p.rvar = running_var;
  p.momentum = momentum;
  p.eps = eps;
  p.o_mean = mean;
  p.o_invstd = invstd;
  p.o_scale = scale;
  p.o_shift = shift;
  const dim3 grid(ew_grid(M, C));
  switch (act) {
    case CVHIP_ACT_RELU: hipLaunchKernelGGL((ew_kernel<0, CVHIP_ACT_RELU, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
    case CVHIP_ACT_SILU: hipLaunchKernelGGL((ew_kernel<0, CVHIP_ACT_SILU, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
    case CVHIP_ACT_LEAKY: hipLaunchKernelGGL((ew_kernel<0, CVHIP_ACT_LEAKY, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
    default: return CVHIP_ERR_UNSUPPORTED;
  }
  return check_launch("ew_kernel<0,acc,lazyres>");
}

int cvhip_bn_act_bwd_apply_acc(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, void* dy, int32_t ld_dy, int64_t M, int32_t C,
                               const float* scale, const float* shift, const float* mean, const float* invstd, const double* acc,
                               int32_t acc_ld, float* dgamma, float* dbeta, int32_t accumulate, int32_t act, float act_param, void* stream) {
  if (!dz || !y || !dy || !scale || !shift || !mean || !invstd || !acc || M <= 0 || C <= 0 || acc_ld < C) return CVHIP_ERR_INVALID;
  if (C > kEwAccMaxC) return CVHIP_ERR_UNSUPPORTED;
  EwParams p{};
  p.a = (const h16_t*)dz;
  p.ld_a = ld_dz;
  p.y = (const h16_t*)y;
  p.ld_y = ld_y;
  p.out = (h16_t*)dy;
  p.ld_out = ld_dy;
  p.M = M;
  p.C = C;
  p.scale = scale;
  p.shift = shift;
  p.mean = mean;
  p.invstd = invstd;
  p.act = act;
  p.ap = act_param;
  p.inv_count = 1.f / (float)M;
  p.acc = acc;
  p.acc_ld = acc_ld;
  p.o_dgamma = dgamma;
  p.o_dbeta = dbeta;
  p.accumulate = accumulate;
  CVHIP_LAUNCH_ACT_ACC(ew_kernel, 1, act, dim3(ew_grid(M, C)), (hipStream_t)stream, p)
  return check_launch("ew_kernel<1,acc>");
}

}  // extern "C"
