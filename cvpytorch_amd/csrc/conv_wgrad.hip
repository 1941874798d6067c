// conv_wgrad.hip — weight-gradient implicit GEMM for gfx950 (CDNA4).
//
//   dW[n][t*Cin + c] (+)= sum_m dY[m][n] * X[pix(m) + tap(t)][c]         (fp32, KRSC layout)
//
// The reduction runs over output pixels m = (batch, p, q): in NHWC that is the SLOW axis of both
// operands, so the MFMA fragments (8 consecutive reduction elements per lane) are gathered from
// row-major [pixel][channel] LDS tiles with the gfx950 hardware-transpose read
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10) — no software transpose, coalesced 16-B global
// loads along channels. Split-K over pixels (grid.y) with fp32 atomics fills the chip even when
// the weight tensor is a single tile (128x128 3x3: 9 tiles x splits).
//
// Replaces aten::convolution_backward(weight) reached from trainer.py:189 (loss.backward()).
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "conv_plan.h"

namespace cvhip {

struct WgradParams {
  const h16_t* x;
  const h16_t* dy;
  float* dw;
  int NB, IH, IW, Cin, x_ld;
  int OHi, OWi, in_sh, in_sw;
  int dh0, dh_step, dw0, dw_step, TR, TS;
  int Nout, dy_ld;
  int Ktot, M;
  int n_tiles, k_tiles, m_per_split;
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;  // fast_div31 constants of OHi*OWi and OWi
  int ablate;  // 0 = fp32 atomics into dW; 3 = the DETERMINISTIC mode: every pixel split stores its partial tile into its own slab of the
               // caller's workspace and wgrad_fold_kernel adds the slabs in split order — no floating-point atomics anywhere
  float* scratch;
  int64_t split_stride;
  float* det_ws;      // deterministic mode: caller's workspace (>= splits * Nout * Ktot floats), else NULL
  int64_t det_ws_floats;
  int det_accumulate;
};

// dw[i] (+)= sum over the splits, in split order (fixed order => bit-reproducible), 4 floats per thread, 4 independent loads in flight
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ ws, int splits, int64_t n, float* __restrict__ dw, int accumulate) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n && ((((uintptr_t)ws) | ((uintptr_t)dw) | (uintptr_t)(n * 4)) & 15) == 0) {
    float4 a = accumulate ? *reinterpret_cast<const float4*>(dw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    int sidx = 0;
    for (; sidx + 3 < splits; sidx += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(ws + (int64_t)sidx * n + i);
      const float4 v1 = *reinterpret_cast<const float4*>(ws + (int64_t)(sidx + 1) * n + i);
      const float4 v2 = *reinterpret_cast<const float4*>(ws + (int64_t)(sidx + 2) * n + i);
      const float4 v3 = *reinterpret_cast<const float4*>(ws + (int64_t)(sidx + 3) * n + i);
      a.x = (((a.x + v0.x) + v1.x) + v2.x) + v3.x;
      a.y = (((a.y + v0.y) + v1.y) + v2.y) + v3.y;
      a.z = (((a.z + v0.z) + v1.z) + v2.z) + v3.z;
      a.w = (((a.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; sidx < splits; ++sidx) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)sidx * n + i);
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
    *reinterpret_cast<float4*>(dw + i) = a;
  } else {
    for (int64_t j = i; j < n && j < i + 4; ++j) {
      float a = accumulate ? dw[j] : 0.f;
      for (int sidx = 0; sidx < splits; ++sidx) a += ws[(int64_t)sidx * n + j];
      dw[j] = a;
    }
  }
}

typedef __attribute__((address_space(3))) h16x4 lds_h16x4;

// Operand loads issued through inline asm: hipcc's waitcnt pass does not see them, so it cannot drain them (it put s_waitcnt
// vmcnt(0) at the header of the pipelined loop); wgrad_wait_vm<N> is the only vmcnt wait of the loop and the empty asm statements
// after it name the registers as in/out operands, so no use of a loaded value can be scheduled above the wait.
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;  // a native vector: asm register operands cannot be HIP's uint4 struct
__device__ __forceinline__ void gload16_async(u32x4& dst, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void wgrad_wait_vm() {
  static_assert(N >= 0 && N <= 8, "vmcnt literal table");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// M * tiles threshold below which a block runs two 4-wave groups (per-shape A/B on YOLOv5-s and DeepLabv3+ layers)
constexpr int64_t kWgTwoGroupWork = 600000;

__device__ __forceinline__ h16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
  h16x4 lo = CVHIP_DS_READ_TR16_B64((lds_h16x4*)(p0));
  h16x4 hi = CVHIP_DS_READ_TR16_B64((lds_h16x4*)(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TN: out-channel tile (128/64/32); K-column tile is always 128; reduction step 32 pixels.
// A block is TWO 4-wave groups working on the two halves of the block's pixel range with private LDS rings; their
// accumulators are folded through LDS before the epilogue, which halves the fp32 atomics per MFMA (the atomic epilogue was
// 28 % of wgrad time: profiles/README.md) at unchanged occupancy (1 x 8 waves instead of 2 x 4 per CU).
// PD: prefetch depth in 32-pixel steps. The operands of steps s+1 .. s+PD are in flight (global -> registers) while step s is
// multiplied from the LDS. PD 1 left one step (12-16 KB per block) in flight for ~300 cycles of fragment reads + 8-16 MFMAs, far
// less than the load latency under load: the waves sat in s_waitcnt for 41-59 % of their cycles (profiles/r02_sq_step_summary.txt).
// Loads are unconditional (masked lanes read the tensor's first bytes and are zeroed on the way into the LDS), so hipcc emits
// counted vmcnt waits and the younger steps stay in flight across the store of the oldest one.
// ABL: always 0 in the library (profiling builds of round 2 / 3 used 4 = no fragment reads / MFMAs, 5 = no global
// loads after the first step (LDS writes, fragment reads and MFMAs only)
template <int TN, int WN, int WK, int kWgGroups, int PD = 3, int ABL = 0>
// (launch bounds: PD 1 with a 128-VGPR cap — 4 blocks per CU instead of 3 — was measured in round 3: the 128-wide tile spills
// 12-20 B/lane and loses 15-50 % per launch, profiles/r03_wgrad_ablation.log)
__global__ __launch_bounds__(256 * kWgGroups, kWgGroups == 1 ? 2 : 1) void wgrad_kernel(const WgradParams p) {
  constexpr int TK = 128;
  constexpr int WAVES_K = TK / WK;
  static_assert((TN / WN) * WAVES_K == 4, "4 waves per block");
  constexpr int NF = WN / 16, KF = WK / 16;
  constexpr int DV = TN / 8;             // 16-B vectors per dY row
  constexpr int D_ROWS = 256 / DV;       // dY rows staged per pass
  constexpr int D_IT = (32 + D_ROWS - 1) / D_ROWS;
  constexpr int D_ROWB = TN * 2;         // dY LDS row bytes
  constexpr int D_SEGM = TN / 16 - 1;    // 32-B segment mask
  constexpr int X_ROWB = TK * 2;
  constexpr int D_BYTES = 32 * D_ROWB, X_BYTES = 32 * X_ROWB;

  constexpr int GROUP_BYTES = 2 * D_BYTES + 2 * X_BYTES;
  static_assert(kWgGroups == 1 || kWgGroups * GROUP_BYTES >= (kWgGroups / 2) * 4 * NF * KF * 4 * 64 * 4, "LDS must hold half the groups' accumulators for the fold");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kWgGroups * GROUP_BYTES];
  const int grp = threadIdx.x >> 8;
  unsigned char* const sD = smem + grp * GROUP_BYTES;
  unsigned char* const sX = sD + 2 * D_BYTES;

  const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;

  // 1-D grid, XCD-aware: all (n, k) tiles of one pixel split get consecutive logical ids => the same XCD, so the x / dY
  // slab of the split is fetched into ONE L2 and re-used by the other tiles (PMC: 3.1x over-fetch with the 2-D grid)
  const int tiles = p.n_tiles * p.k_tiles;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / tiles;
  const int tile = lin - split * tiles;
  const int ntile = tile / p.k_tiles, ktile = tile - ntile * p.k_tiles;
  const int n0 = ntile * TN, k0 = ktile * TK;
  const int blk_begin = split * p.m_per_split;
  const int blk_end = min(p.M, blk_begin + p.m_per_split);
  if (blk_end <= blk_begin) return;
  // group g takes rows [blk_begin + g*half, +half); both groups run the same number of steps (block-wide barriers), the
  // shorter one sees masked rows
  const int half = (((blk_end - blk_begin + kWgGroups - 1) / kWgGroups + 31) >> 5) << 5;
  const int m_begin = blk_begin + grp * half;
  const int m_end = min(blk_end, m_begin + half);
  const int nsteps = half >> 5;

  // ---- X staging: thread owns 16-B column (t&15) of rows (t>>4) and (t>>4)+16 -------------------
  const int xv = t & 15;
  const int kcol = k0 + xv * 8;
  const bool k_ok = kcol < p.Ktot;
  int c0 = 0, dh = 0, dw = 0;
  if (k_ok) {
    const int tap = kcol / p.Cin;
    c0 = kcol - tap * p.Cin;
    const int tr = tap / p.TS, ts = tap - tr * p.TS;
    dh = p.dh0 + tr * p.dh_step;
    dw = p.dw0 + ts * p.dw_step;
  }
  const int OHWi = p.OHi * p.OWi;
  // ---- dY staging ---------------------------------------------------------------------------------
  const int dv = t % DV;
  const int drow = t / DV;
  const int dn = n0 + dv * 8;
  const bool dn_ok = dn < p.Nout;  // Nout % 8 == 0 is required

  u32x4 rx[PD][2], rd[PD][D_IT];
  unsigned live[PD];  // bit i: rx[i] is a real element; bit 8+i: rd[i] is

  auto load_step = [&](int step, auto setc) {
    constexpr int S = decltype(setc)::value;
    if (ABL == 5 && step >= PD) return;
    const int mb = m_begin + step * 32;
    unsigned lv = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // pixel -> (image, row, column) by exact multiply-shift division: no data-dependent loop in the pipelined body (a
      // carry loop here made hipcc drain vmcnt to 0 at the loop header)
      const int m = mb + (t >> 4) + 16 * i;
      const int n = (int)fast_div31((unsigned)m, p.ohw_mul, p.ohw_sh);
      const int rem = m - n * OHWi;
      const int oh = (int)fast_div31((unsigned)rem, p.ow_mul, p.ow_sh);
      const int ow = rem - oh * p.OWi;
      const int ih = oh * p.in_sh + dh, iw = ow * p.in_sw + dw;
      const bool ok = k_ok && m < m_end && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      const int64_t off = ok ? ((int64_t)((n * p.IH + ih) * p.IW + iw) * p.x_ld + c0) : 0;
      gload16_async(rx[S][i], p.x + off);
      lv |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int row = drow + i * D_ROWS;
      const int m = mb + row;
      const bool ok = row < 32 && dn_ok && m < m_end;
      const int64_t off = ok ? ((int64_t)m * p.dy_ld + dn) : 0;
      gload16_async(rd[S][i], p.dy + off);
      lv |= (ok ? 1u : 0u) << (8 + i);
    }
    live[S] = lv;
  };
  // swizzle: 32-B segment index ^= h(px), h(px) = (px&3) | ((px>>1)&4)
  auto store_step = [&](int buf, auto setc) {
    constexpr int S = decltype(setc)::value;
    const uint4 z = make_uint4(0, 0, 0, 0);
    // the oldest step in flight has landed when only the PD-1 younger ones are outstanding (2 + D_IT loads per step, every lane)
    if (ABL == 5) wgrad_wait_vm<0>();
    else wgrad_wait_vm<(PD - 1) * (2 + D_IT)>();
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(rx[S][i]));
#pragma unroll
    for (int i = 0; i < D_IT; ++i) asm volatile("" : "+v"(rd[S][i]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = (t >> 4) + 16 * i;
      const int h = (px & 3) | ((px >> 1) & 4);
      const bool ok = (live[S] >> i) & 1u;
      uint4 v;
      v.x = ok ? rx[S][i][0] : z.x;
      v.y = ok ? rx[S][i][1] : z.y;
      v.z = ok ? rx[S][i][2] : z.z;
      v.w = ok ? rx[S][i][3] : z.w;
      *reinterpret_cast<uint4*>(sX + buf * X_BYTES + px * X_ROWB + ((((xv >> 1) ^ h) & 7) << 5) + (xv & 1) * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int px = drow + i * D_ROWS;
      if (px < 32) {
        const int h = (px & 3) | ((px >> 1) & 4);
        const bool ok = (live[S] >> (8 + i)) & 1u;
        uint4 v;
        v.x = ok ? rd[S][i][0] : z.x;
        v.y = ok ? rd[S][i][1] : z.y;
        v.z = ok ? rd[S][i][2] : z.z;
        v.w = ok ? rd[S][i][3] : z.w;
        *reinterpret_cast<uint4*>(sD + buf * D_BYTES + px * D_ROWB + ((((dv >> 1) ^ h) & D_SEGM) << 5) + (dv & 1) * 16) = v;
      }
    }
  };

  f32x4 acc[NF][KF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < KF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read geometry (lane-constant): pixel rows 8g+4j+q, 8-B column (lane&3) of the 32-B seg
  const int g = lane >> 4, q = (lane >> 2) & 3;
  const int hsw = q | ((g & 1) << 2);
  const int px0 = 8 * g + q;
  auto compute = [&](int cur) {
    if (ABL == 4) return;
    h16x8 fd[NF], fx[KF];
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const int seg = (wn * WN + a * 16) >> 4;
      const unsigned char* base = sD + cur * D_BYTES + (((seg ^ hsw) & D_SEGM) << 5) + (lane & 3) * 8;
      fd[a] = tr_read8(base + px0 * D_ROWB, base + (px0 + 4) * D_ROWB);
    }
#pragma unroll
    for (int b = 0; b < KF; ++b) {
      const int seg = (wk * WK + b * 16) >> 4;
      const unsigned char* base = sX + cur * X_BYTES + (((seg ^ hsw) & 7) << 5) + (lane & 3) * 8;
      fx[b] = tr_read8(base + px0 * X_ROWB, base + (px0 + 4) * X_ROWB);
    }
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int b = 0; b < KF; ++b)
        acc[a][b] = CVHIP_MFMA_16X16X32(fd[a], fx[b], acc[a][b], 0, 0, 0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1 % PD>;
  using I2 = std::integral_constant<int, 2 % PD>;
  if constexpr (PD == 3) {
    // register sets rotate with the step: step s lives in set s % 3; the LDS ring stays two deep
    load_step(0, I0{});
    load_step(1, I1{});
    load_step(2, I2{});
    store_step(0, I0{});
    __syncthreads();
    auto body = [&](int step, auto mine, auto next) {
      load_step(step + 3, mine);  // this step's set was emptied into the LDS one step ago; steps past the end are all-masked
      compute(step & 1);
      if (step + 1 < nsteps) store_step((step + 1) & 1, next);
      __syncthreads();
    };
    for (int step = 0; step < nsteps; step += 3) {
      body(step, I0{}, I1{});
      if (step + 1 < nsteps) body(step + 1, I1{}, I2{});
      if (step + 2 < nsteps) body(step + 2, I2{}, I0{});
    }
  } else {
    load_step(0, I0{});
    store_step(0, I0{});
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      const int cur = step & 1;
      if (step + 1 < nsteps) load_step(step + 1, I0{});
      compute(cur);
      if (step + 1 < nsteps) store_step(cur ^ 1, I0{});
      __syncthreads();
    }
  }

  wgrad_wait_vm<0>();  // the all-masked loads issued past the last step must not land in registers the epilogue re-uses

  // fold the other groups into group 0 through LDS (the staging rings are dead after the final barrier of the loop): a
  // binary tree, each round the upper half of the live groups parks its accumulators and the lower half adds them
  if constexpr (kWgGroups >= 2) {
    float* fold = reinterpret_cast<float*>(smem);
    constexpr int ACC_FLOATS = 4 * NF * KF * 4 * 64;  // one group's accumulators
#pragma unroll
    for (int live = kWgGroups; live > 1; live >>= 1) {
      const int hl = live >> 1;
      if (grp >= hl && grp < live) {
        float* dst = fold + (grp - hl) * ACC_FLOATS;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
          for (int b = 0; b < KF; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((wave * NF * KF + a * KF + b) * 4 + r) * 64 + lane] = acc[a][b][r];
      }
      __syncthreads();
      if (grp < hl) {
        const float* src = fold + grp * ACC_FLOATS;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
          for (int b = 0; b < KF; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] += src[((wave * NF * KF + a * KF + b) * 4 + r) * 64 + lane];
      }
      if (live > 2) __syncthreads();  // the next round overwrites the parking area
    }
    if (grp != 0) return;
  }
  // epilogue: lane holds D[n = 4*(lane>>4)+r][kcol = lane&15]
  float* const dwp = p.ablate == 3 ? p.scratch + (int64_t)split * p.split_stride : p.dw;
#pragma unroll
  for (int a = 0; a < NF; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * WN + a * 16 + 4 * (lane >> 4) + r;
      if (n >= p.Nout) continue;
#pragma unroll
      for (int b = 0; b < KF; ++b) {
        const int kc = k0 + wk * WK + b * 16 + (lane & 15);
        if (kc < p.Ktot) {
          if (p.ablate == 3) dwp[(int64_t)n * p.Ktot + kc] = acc[a][b][r];
          else unsafeAtomicAdd(dwp + ((int64_t)n * p.Ktot + kc), acc[a][b][r]);
        }
      }
    }
  }
}

// (The LDS-DMA staged variant of this kernel (round 3) left the library in round 6: hipcc drains an outstanding LDS-DMA before every
// ds_read_b64_tr_b16 builtin it emits — conv_wgrad_band.hip has the measurement — so that form could never overlap staging with the
// gathers; profiles/r03_wgrad_dma_ab.log holds its numbers.)

// Launcher policy (round 4, tools/wgrad_sweep.sh -> profiles/r04_wgrad_sweep.log: 38 configurations x 18 shapes, then step A/Bs): block
// target 768 (256 for the 128-wide tile on short 1x1 pixel ranges), >= 512 pixel rows per 4-wave group, two-group blocks where the atomic
// epilogue dominates, three 32-pixel steps in flight for the 64- and 128-wide tiles, no grid a few blocks over the resident slots.
template <int TN, int WN, int WK>
static int launch_wg(WgradParams& p, hipStream_t stream, int tgt_hint = 0, int groups_hint = 0) {
  p.n_tiles = cdiv(p.Nout, TN);
  p.k_tiles = cdiv(p.Ktot, 128);
  const int tiles = p.n_tiles * p.k_tiles;
  constexpr int kTarget = 768, kMinRows = 512;
  p.ablate = p.det_ws ? 3 : 0;
  p.scratch = nullptr;
  p.split_stride = 0;
  // Two 4-wave groups per block (accumulators folded through LDS, half the atomics) when the atomic epilogue is a large
  // share of the block's work: few pixel rows per (n,k) tile. Otherwise 4-wave blocks (more resident blocks per CU).
  // (the 32-wide output tile always gains: its blocks have the least MFMA work per atomic)
  const int groups = groups_hint ? groups_hint : ((TN == 32 || (int64_t)p.M * tiles <= kWgTwoGroupWork) ? 2 : 1);
  const int rows_min = kMinRows * groups;
  // block target: every block ends by flushing its TN x 128 tile with fp32 atomics, so the atomic volume is blocks x 64 KB whatever the
  // layer; for the 128-wide tile on short pixel ranges (1x1 layers of <= 40 k pixels: ResNet layer3 / layer4 at batch 16, the 20x20 maps
  // of the detectors) 768 blocks flush more bytes than they read — 256 blocks are 15-19 % faster there (profiles/r03_wgrad_ablation.log)
  int tgt = kTarget;
  if (TN == 128 && p.TR == 1 && p.TS == 1 && p.M <= 40000) tgt = 256;
  if (tgt_hint > 0) tgt = tgt_hint;
  int splits = cdiv(tgt, tiles);
  const int max_splits = (p.M + rows_min - 1) / rows_min;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = cdiv(p.M, splits);
  mps = ((mps + 63) / 64) * 64;
  splits = cdiv(p.M, mps);
  {
    // resident block slots of the instance that will run (VGPR / LDS occupancy from the compiler's resource remarks: a one-group block
    // is one wave per SIMD, a two-group block two): a grid a few blocks larger than that runs a second, nearly empty round — rounding
    // the rows per split up to 64 and the splits up to whole numbers overshot it (128 -> 128 3x3 at 64 x 128, batch 16: 774 blocks on 768
    // slots, +11 %; profiles/r04_wgrad_sweep.log)
    const int cap = groups >= 2 ? (TN == 32 ? 768 : 256) : (TN == 128 ? 512 : 1024);
    while (tiles * splits > cap && tiles * splits < 2 * cap - cap / 4 && splits > 1) {
      mps += 64;
      splits = cdiv(p.M, mps);
    }
  }
  p.m_per_split = mps;
  if (p.det_ws) {
    const int64_t n = (int64_t)p.Nout * p.Ktot;
    if (p.det_ws_floats == -1) {  // size query
      p.det_ws_floats = (int64_t)splits * n;
      return CVHIP_OK;
    }
    if (p.det_ws_floats < (int64_t)splits * n) return CVHIP_ERR_INVALID;
    p.scratch = p.det_ws;
    p.split_stride = n;
    // every (n, k) of every split slab is written exactly once
    if (groups >= 2) hipLaunchKernelGGL((wgrad_kernel<TN, WN, WK, 2, 1>), dim3(tiles * splits), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((wgrad_kernel<TN, WN, WK, 1, 1>), dim3(tiles * splits), dim3(256), 0, stream, p);
    int st = check_launch("wgrad_kernel(det)");
    if (st) return st;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, stream, p.det_ws, splits, n, p.dw, p.det_accumulate);
    return check_launch("wgrad_fold_kernel");
  }
  // (four groups per block were measured too: 1024-thread blocks in lockstep lose 10-50 % on every YOLOv5-s layer)
  // Prefetch depth: three steps in flight pay for the 64-wide tile (-10...-17 % per launch, profiles/r02_wgrad_pd.log) and, inside the
  // train step where x is cold, for the 128-wide tile too (177-179 VGPRs = two resident one-group blocks per CU instead of three:
  // -0.4 ms per YOLOv5-s step, -0.5 ms per DeepLabv3+ step, profiles/r04_wgrad_step_ab.log); nothing for the 32-wide one
  constexpr int pd = TN == 32 ? 1 : 3;
  if (groups >= 2) hipLaunchKernelGGL((wgrad_kernel<TN, WN, WK, 2, pd>), dim3(tiles * splits), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((wgrad_kernel<TN, WN, WK, 1, pd>), dim3(tiles * splits), dim3(256), 0, stream, p);
  return check_launch("wgrad_kernel");
}

int try_launch_wgrad_band(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream);  // conv_wgrad_band.hip

int launch_wgrad_impl(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream, float* det_ws, int64_t* det_ws_floats,
                      int det_accumulate) {
  if (!det_ws) {
    const int stem = try_launch_stem_wgrad(d, x, dy, dw, stream);  // 8-channel image stem: patch kernel (conv_stem.hip; atomic epilogue)
    if (stem >= 0) return stem;
    const int band = try_launch_wgrad_band(d, x, dy, dw, stream);  // stride-1 3x3 "same" layers: tap-resident tiles, replicas folded in the LDS
    if (band >= 0) return band;
  }
  WgradParams p;
  p.det_ws = det_ws;
  p.det_ws_floats = det_ws_floats ? *det_ws_floats : 0;
  p.det_accumulate = det_accumulate;
  p.x = (const h16_t*)x;
  p.dy = (const h16_t*)dy;
  p.dw = dw;
  p.NB = d->N;
  p.IH = d->H;
  p.IW = d->W;
  p.Cin = d->C;
  p.x_ld = d->x_ld;
  p.OHi = conv_out_dim(d->H, d->pad_h, d->dil_h, d->R, d->stride_h);
  p.OWi = conv_out_dim(d->W, d->pad_w, d->dil_w, d->S, d->stride_w);
  p.in_sh = d->stride_h;
  p.in_sw = d->stride_w;
  p.dh0 = -d->pad_h;
  p.dh_step = d->dil_h;
  p.dw0 = -d->pad_w;
  p.dw_step = d->dil_w;
  p.TR = d->R;
  p.TS = d->S;
  p.Nout = d->K;
  p.dy_ld = d->y_ld;
  p.Ktot = d->R * d->S * d->C;
  p.M = d->N * p.OHi * p.OWi;
  if (p.M <= 0) return CVHIP_OK;
  div31_consts(p.OHi * p.OWi, &p.ohw_mul, &p.ohw_sh);
  div31_consts(p.OWi, &p.ow_mul, &p.ow_sh);
  // Out-channel tile. The fp32 atomic epilogue is 20-45 % of a launch for the small-M layers (measured in round 1 with the
  // epilogue switched off): narrower tiles mean more (n, k) tiles, hence fewer pixel splits and fewer atomics per MFMA, at
  // the price of more LDS fragment reads per MFMA. Per-shape A/B on the YOLOv5-s layers (tile forced to 32 / 64 / 128)
  // and DeepLabv3+ R50 layers: memory-bound 1x1 layers of modest size (M*K*C <= 7.5e9: the 20x20..80x80 YOLO layers) want
  // the 32-wide tile — the large ResNet 1x1 layers lose up to 2x with it; 3x3 layers with >= 256 outputs the 64-wide one.
  // Round 4 (isolated sweep of tile x block target x groups over 18 layer shapes, tools/wgrad_sweep.sh -> profiles/r04_wgrad_sweep.log,
  // then a step A/B): (a) 1x1 layers with >= 256 channels on both sides used the 32-wide tile, whose 8 x 2 (n, k) tiles re-stage
  // x eight times and dY twice from the L2 (525 MB per launch for 256 -> 256 @40x40: L2->LDS-bound at 45 us) — the 128-wide tile as ONE
  // two-group block per CU (256 blocks, 2 x 2 tiles, 210 MB staged, 16.8 MB of atomics) runs them in 38-43 us instead of 52;
  // (b) 3x3 layers with >= 256 outputs: the 128-wide tile with 512 blocks instead of the 64-wide one with 768 (-7...-14 %).
  int tn = d->K <= 32 ? 32 : d->K <= 64 ? 64 : 128;
  int tgt_hint = 0, groups_hint = 0;
  if (d->R == 1 && d->S == 1) {
    if ((double)p.M * d->K * d->C <= 7.5e9) {
      tn = 32;
      if (d->K >= 256 && d->C >= 256) {
        tn = 128;
        tgt_hint = 256;
        groups_hint = 2;
      }
    }
  } else if (d->K >= 256 && d->C <= 1024) {
    tn = 128;
    tgt_hint = 512;
    groups_hint = 1;
  }
  int rc;
  if (tn == 32) rc = launch_wg<32, 32, 32>(p, stream, tgt_hint, groups_hint);
  else if (tn == 64) rc = launch_wg<64, 32, 64>(p, stream, tgt_hint, groups_hint);
  else rc = launch_wg<128, 64, 64>(p, stream, tgt_hint, groups_hint);
  if (det_ws_floats) *det_ws_floats = p.det_ws_floats;
  return rc;
}

int launch_wgrad(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream) {
  return launch_wgrad_impl(d, x, dy, dw, stream, nullptr, nullptr, 0);
}

}  // namespace cvhip
