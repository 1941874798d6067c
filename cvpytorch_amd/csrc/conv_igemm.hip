// conv_igemm.hip — implicit-GEMM convolution for gfx950 (CDNA4): fprop and dgrad share one kernel.
//
//   Out[m][n] = sum_{t,c} X[pix(m) + tap(t)][c] * Wt[n][t*Cin + c]
//
//   m  : output "iteration pixel" (batch, oh, ow) of one stride-parity class
//   n  : output channel (K for fprop, C for dgrad)
//   t  : tap (r,s) of the class; input pixel = (oh*in_sh + dh0 + i*dh_step, ow*in_sw + dw0 + j*dw_step)
//
// fprop  : one class, in_s = conv stride, out_s = 1.
// dgrad  : one class per (h % stride_h, w % stride_w) parity; each class only visits the taps that
//          actually reach it (k3 s2: 1+2+2+4 = 9 taps over the 4 classes — no wasted MFMA work),
//          in_s = 1, out_s = conv stride, weights pre-packed per class as [C][taps][K].
//
// Replaces aten::convolution / convolution_backward(input) reached from
// reference src/models/bricks/conv_module.py:209 and trainer.py:189 (loss.backward()).
//
// Tiling (256 threads = 4 waves of 64): block tile BM x BN x 32, wave tile WM x WN built from
// v_mfma_f32_16x16x32_bf16 fragments with SWAPPED operands (A-operand = weight rows, B-operand =
// pixel rows) so every lane ends up holding 4 CONSECUTIVE output channels of one pixel -> 8-byte
// packed bf16 stores along the NHWC channel axis. Global->register->LDS staging (zero-fill for the
// conv halo needs the register hop), double-buffered LDS, one barrier per 32-deep K step, next
// tile's global loads issued before the MFMA block (cdna_hip_programming.md T14). LDS rows are
// 64 B (32 bf16) with a 16-B-slot XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free for the 4x16-lane groups of MI355X_MICROARCH.md §LDS.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

// (The register-staged v1 kernel of round 1 and the measured-and-rejected launch forms — 64-deep ring slots, 8-wave blocks, wave
// specialisation, chunk-major K order, the ablation instances — left the product library in round 6: git history and DESIGN.md 4.0 keep
// their measurements.)

// =====================================================================================================
// v2: the same implicit GEMM with LDS-DMA staging (global_load_lds_dwordx4: HBM/L2 -> LDS without a VGPR hop and
// without ds_write instructions), a 3-deep LDS ring and ONE raw s_barrier per K step with a COUNTED s_waitcnt
// vmcnt (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4): while tile kt is consumed, tiles kt+1
// and kt+2 are in flight. The conv halo / K tail is zero-filled by pointing masked lanes at a 64-byte zero page.
// The LDS image is lane-linear (DMA destination = wave base + lane*16), so the 16-B-slot XOR swizzle is applied to
// the SOURCE address (which logical K-slot a lane fetches) and to the fragment reads (rule 21).
// =====================================================================================================
// Masked DMA lanes read zeros from here. FAST staging walks a masked row's pointer along with the live ones (Cin * 2 bytes per tap),
// so the page covers kFastMaxCin channels plus one 64-byte row.
constexpr int kFastMaxCin = 4096;
__device__ __attribute__((aligned(64))) unsigned int g_zero_page[kFastMaxCin / 2 + 16];

#define CVHIP_GLDS16(src, dst)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// counted s_waitcnt vmcnt(N) for a compile-time N (the immediate must be a literal)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 12, "vmcnt literal table");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// ABL: profiling ablation (CVHIP_IGEMM_ABLATE): 0 = the kernel, 1 = staging only (no LDS reads / MFMA), 2 = compute only
// BK : reduction depth per ring slot / barrier. 32: LDS rows of 64 B, a DMA instruction covers 16 rows x 64 B (half cache lines).
//      64: LDS rows of 128 B, a DMA instruction covers 8 rows x 128 B = FULL 128-byte lines of the NHWC channel axis (when
//      Cin >= 64), half the barriers / tap decodes / counted waits per MFMA (cdna_hip_programming.md §5: "x through LDS in
//      full 128-B lines", BK 32 -> 64). The 16-B slots of a 128-B row are XOR-swizzled by (row >> 1) & 7 — conflict-free for
//      the 16-lane groups of ds_read_b128 — on the DMA SOURCE side and on the fragment reads (rule 21).
// FAST: Cin % BK == 0, so a K step lies inside ONE tap and its channel offset advances by BK: the per-row source pointers are
//      kept in registers, bumped by one 64-bit add per K step and re-derived (halo test, pixel address) only when the tap changes —
//      every Cin / BK steps, under a wave-uniform branch. The general path re-decodes (tap, channel) and re-tests the halo for every
//      row at every K step: ~110 VALU instructions and four divergent branches per step, which made ADDRESS GENERATION, not the
//      DMA rate, the cost of staging (profiles/r01_igemm_ablation.log: staging-only 146 us vs 48 us of pure MFMA; halving the
//      steps with BK 64 doubled the rows per step and changed nothing: profiles/r02_ab_bwd1x1_bk64.log).
// A register-double-buffered variant (fragments of tile kt+1 read while tile kt is multiplied, asm-issued ds_read_b128 with one
// explicit lgkmcnt wait per step) was built and measured in round 2: 8-13 % SLOWER per kernel (profiles/r02_igemm_ablation.log) and
// removed — the step is not bound by the latency of its own fragment reads.
// NW : waves per block. 4: one wave per SIMD and block; 8 (512 threads): two — the same block tile with HALF the wave tile, so a block
//      alone on its CU still has a second wave per SIMD to issue MFMAs while the first one sits in its DMA-issue / fragment-read /
//      barrier phase (profiles/r03_ceilings_probe.log: neither the LDS read rate nor the L2->LDS path is the limit of the 4-wave
//      form; its in-order waves are).
// (occupancy bound = the blocks per CU the LDS ring allows, at most 4: keeps the register allocation from dropping a resident block —
// the 256x64 two-slot configuration sits exactly on the 128-VGPR step)
constexpr int igemm_lds_bytes(int BM, int BN, int BK, int NST, int LW) {
  const int rpt = LW * (1024 / (BK * 2));
  return NST * (BM + (BN < rpt ? rpt : BN)) * BK * 2;
}
constexpr int igemm_min_waves(int BM, int BN, int BK, int NST, int NW, int LW, int acc_regs) {
  const int blocks = (160 * 1024) / igemm_lds_bytes(BM, BN, BK, NST, LW);
  const int cap = acc_regs >= 128 ? 2 : 4;  // a 128-register accumulator tile leaves room for two waves per SIMD at most
  return (blocks < 1 ? 1 : blocks > cap ? cap : blocks) * (NW / 4);
}
// WS : wave specialisation (NW = 8, FAST, 3-deep ring). Waves 0-3 are CONSUMERS: fragment reads + MFMA only, one per SIMD, the whole
//      block tile between them (128x64 each for 256x128); waves 4-7 are LOADERS: they issue every LDS DMA of the ring and wait for
//      it, nothing else. Wave w and wave w + 4 share a SIMD (workgroup waves are dealt to the SIMDs cyclically), so every SIMD runs
//      one MFMA stream that never stops to compute addresses / issue DMAs / wait for its own loads, beside one VMEM stream that never
//      competes for the matrix pipe. One s_barrier per ring slot couples the two roles (loaders arrive once the next slot has
//      landed, consumers once they are done with the current one). Why: profiles/r03_igemm_ablation.log — in the symmetric form
//      staging-only takes 126 us, fragment reads + MFMA only 115 us, both together 181 us: the two halves barely overlap because
//      the same in-order waves do both and the blocks of a CU fall into step.
// ORD: K-step order of FAST staging. 0: tap-major (all channel chunks of a tap, then the next tap) — a pixel row's cache lines are
//      touched again only Cin / BK steps later, by the next tap. 1: CHUNK-major (all taps of a channel chunk, then the next chunk):
//      consecutive steps read the same BK-channel segment of (almost) the same pixel rows, shifted by one tap, so the re-reads hit
//      in L2 instead of going back to the Infinity Cache (profiles/r03_igemm_ablation.log: the pixel-tile DMA alone ran at the
//      fabric's ~10 TB/s, 17 B/clk/CU, against 30-50 B/clk/CU for the same pattern from an L2-resident window).

// activation of 8 / 4 values with ONE switch (a switch per element multiplied the unrolled epilogue's code size and pushed the
// 256-wide streaming kernel's accumulators into scratch)
template <int NV>
__device__ __forceinline__ void ig_act_vec(float (&v)[NV], int act, float ap) {
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

// EPI: fused-epilogue instance (out = act((acc + bias) * ep_scale + ep_shift)); the default instances carry none of its code
template <int BM, int BN, int WM, int WN, int ABL = 0, int NST = 3, int BK = 32, bool FAST = false, int NW = 4, bool WS = false, int ORD = 0, bool EPI = false>
__global__ __launch_bounds__(NW * 64, WS ? 2 : igemm_min_waves(BM, BN, BK, NST, NW, NW, WM * WN / 64)) void igemm_dma_kernel(const IgemmKernArgs p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int WAVES_M = BM / WM;
  constexpr int CW = WS ? 4 : NW;       // waves that compute
  constexpr int LW = WS ? NW - 4 : NW;  // waves that stage
  static_assert(!WS || (NW == 8 && FAST && NST == 3), "wave specialisation: 4 consumers + 4 loaders on the FAST 3-deep ring");
  static_assert(WAVES_M * WAVES_N == CW, "one wave tile per computing wave");
  static_assert(BK == 32 || BK == 64, "reduction depth per stage");
  constexpr int MF = WM / 16, NF = WN / 16;
  constexpr int RPI = 1024 / (BK * 2);  // rows per DMA instruction (1 KiB): 16 (BK 32) / 8 (BK 64)
  constexpr int RPT = LW * RPI;         // rows per pass of the staging waves: 64 / 32 (4 waves)
  constexpr int ROWB = BK * 2;          // LDS row bytes
  constexpr int A_IT = BM / RPT;
  constexpr int B_ROWS = BN < RPT ? RPT : BN;  // B tile padded so all 4 waves issue the same DMA count
  constexpr int B_IT = B_ROWS / RPT;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = B_ROWS * ROWB, ST_BYTES = A_BYTES + B_BYTES;
  static_assert(NST == 2 || NST == 3, "LDS ring depth");
  constexpr int PER = A_IT + B_IT;  // DMA instructions per stage per wave

  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool consumer = !WS || wave < 4;           // wave-uniform roles
  const int cwave = WS ? (wave & 3) : wave;        // index among the computing waves (loaders: unused)
  const int swave = WS ? (wave & 3) : wave;        // index among the staging waves (consumers: unused)
  const int st_t = WS ? (t & 255) : t;             // thread index among the staging threads
  const int wm = cwave / WAVES_N, wn = cwave % WAVES_N;

  const int lt = xcd_remap(blockIdx.x, p.total_tiles);
  int ci = 0, local;
  if (p.interleave) {
    // dgrad parity classes with equal tile counts: the classes of one spatial tile run back to back on the same XCD, so the
    // stride-2-interleaved pixels they write (half cache lines each) meet in that L2 before they are written back
    if (p.interleave == 3) {
      const int G = p.il_group, g4 = G * p.ncls;
      const int g = lt / g4, k = lt - g * g4;
      const int nfull = p.il_tiles / G;
      const int Gt = g < nfull ? G : p.il_tiles - nfull * G;   // the last group holds the remainder
      const int o = k / Gt;
      ci = (p.il_order >> (4 * o)) & 15;
      local = g * G + (k - o * Gt);
    } else {
      local = lt / p.ncls;
      ci = lt - local * p.ncls;
      if (p.interleave == 2) ci = (ci + local) % p.ncls;  // rotated class order (launch_group)
    }
  } else {
    int tb = 0;
#pragma unroll
    for (int i = 1; i < kKernelClasses; ++i)
      if (i < p.ncls && lt >= p.cls[i].tile_begin) {
        ci = i;
        tb = p.cls[i].tile_begin;
      }
    local = lt - tb;
  }
  // the class record BY VALUE, picked with compile-time indices: a run-time index into the by-value argument block makes the compiler
  // keep the whole block (600 bytes) in scratch memory as soon as anything stops it from folding the access (measured: the step
  // 30 % slower)
  IgemmClass cl = p.cls[0];
#pragma unroll
  for (int i = 1; i < kKernelClasses; ++i)
    if (ci == i) cl = p.cls[i];
  const int mtile = local / p.n_tiles;
  const int ntile = local - mtile * p.n_tiles;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int TR = cl.TR, TS = cl.TS;
  const int Cin = p.Cin;
  const int Ktot = TR * TS * Cin;
  const int nk = (Ktot + BK - 1) / BK;
  const int M = cl.M;
  const int OWi = cl.OWi, OHWi = cl.OHi * cl.OWi;

  // A class WITHOUT taps (1x1 stride-2 dgrad: three of the four pixel parities — the ResNet projection shortcuts) has nothing to multiply:
  // its pixels are the addend, or zero. Copy them row by row (16 bytes per lane, whole pixel rows) and leave: no accumulators, no LDS
  // tile, no barriers (round 6: DeepLabv3+ 256 -> 512 k1 s2 @128x256, three quarters of whose tiles are of this kind).
  if constexpr (!EPI && ABL == 0) {
    if (nk == 0 && p.staged_epilogue && !p.bias && !p.stats && !p.tail_y && (p.Nout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0) &&
        (!p.res || ((p.res_ld & 7) == 0 && ((((uintptr_t)p.res) & 15) == 0)))) {
      constexpr int CPR0 = BN / 8;
      for (int idx = t; idx < BM * CPR0; idx += NW * 64) {
        const int row = idx / CPR0, ch = idx - row * CPR0;
        const int m = m0 + row;
        if (m >= M || n0 + ch * 8 >= p.Nout) continue;
        const int n_img = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
        const int rem = m - n_img * OHWi;
        const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
        const int ow = rem - oh * OWi;
        const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (p.res) v = *reinterpret_cast<const uint4*>(p.res + opix * p.res_ld + n0 + ch * 8);
        *reinterpret_cast<uint4*>(p.y + opix * p.y_ld + n0 + ch * 8) = v;
      }
      return;
    }
  }

  // staging geometry: this thread's row inside a 4-wave pass and its physical 16-B slot inside the LDS row
  const int srow = BK == 32 ? (st_t >> 2) : (swave * 8 + (lane >> 3));
  int ih0[A_IT], iw0[A_IT], pbase[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + i * RPT + srow;
    if (m < M) {
      const int n = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
      const int rem = m - n * OHWi;
      const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
      const int ow = rem - oh * OWi;
      ih0[i] = oh * p.in_sh + cl.dh0;
      iw0[i] = ow * p.in_sw + cl.dw0;
      pbase[i] = n * p.IH * p.IW;
    } else {
      ih0[i] = -(1 << 28);
      iw0[i] = 0;
      pbase[i] = 0;
    }
  }
  const unsigned cin_magic = p.cin_magic, ts_magic = cl.ts_magic;
  auto fdiv = [](unsigned n, unsigned magic) -> unsigned { return magic ? __umulhi(n, magic) : n; };
  const h16_t* __restrict__ wbase = p.w + cl.w_off;
  const h16_t* const zero = reinterpret_cast<const h16_t*>(g_zero_page);

  // BK 32: physical slot (t&3) of a 64-B row must hold LOGICAL K-slot (t&3)^g(row>>2)
  // BK 64: physical slot (lane&7) of a 128-B row must hold LOGICAL K-slot (lane&7) ^ ((row>>1)&7), row = i*32 + wave*8 + (lane>>3)
  const int lslot = BK == 32 ? ((st_t & 3) ^ ((0x78 >> (2 * ((st_t >> 4) & 3))) & 3)) : ((lane & 7) ^ (((swave & 1) << 2) | (lane >> 4)));
  // fragment reads: lane (r = lane&15, g = lane>>4) reads logical slot 4*ks + g of row base + r
  const int swz_r = BK == 32 ? ((lane >> 4) ^ ((0x78 >> (2 * ((lane >> 2) & 3))) & 3)) : ((lane >> 1) & 7);

  auto stage = [&](int kt, int st) {
    const unsigned k = (unsigned)(kt * BK + lslot * 8);
    const unsigned tap = fdiv(k, cin_magic);
    const int c0 = (int)(k - tap * (unsigned)Cin);
    const unsigned tr = fdiv(tap, ts_magic);
    const int ts = (int)(tap - tr * (unsigned)TS);
    const bool tap_ok = (int)tr < TR;
    const int dh = (int)tr * cl.dh_step, dw = ts * cl.dw_step;
    unsigned char* const sA = smem + st * ST_BYTES;
    unsigned char* const sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int ih = ih0[i] + dh, iw = iw0[i] + dw;
      const bool ok = tap_ok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      const h16_t* src = ok ? (p.x + ((int64_t)(pbase[i] + ih * p.IW + iw) * p.x_ld + c0)) : zero;
      CVHIP_GLDS16(src, sA + (i * RPT + swave * RPI) * ROWB);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = i * RPT + srow;
      const int n = n0 + row;
      const bool ok = row < BN && n < p.Nout && (int)k < Ktot;
      const h16_t* src = ok ? (wbase + ((int64_t)n * Ktot + k)) : zero;
      CVHIP_GLDS16(src, sB + (i * RPT + swave * RPI) * ROWB);
    }
  };

  // ---- FAST staging: running source pointers -----------------------------------------------------------------
  const h16_t* aptr[A_IT];
  const h16_t* bptr[B_IT];
  int f_c = 0, f_tr = 0, f_ts = 0;  // wave-uniform: channel offset inside the current tap, tap coordinates of the NEXT stage
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int n = n0 + i * RPT + srow;  // rows past the tile / past Nout fetch a valid row: their columns are never stored or summed
      n = n < p.Nout ? n : p.Nout - 1;
      bptr[i] = wbase + ((int64_t)n * Ktot + lslot * 8);
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) aptr[i] = zero;
  }
  // chunk-major order (ORD 1): per staged row the address of its tap-(0,0) pixel and one validity bit per tap; a step's source is
  // base + (wave-uniform tap offset + chunk offset) or the zero page
  unsigned amask[A_IT];
  int o_t = 0, o_c = 0;
  const int ntaps = TR * TS;
  if constexpr (FAST && ORD == 1) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      unsigned mk = 0;
      for (int tr = 0, b = 0; tr < TR; ++tr)
        for (int ts = 0; ts < TS; ++ts, ++b) {
          const int ih = ih0[i] + tr * cl.dh_step, iw = iw0[i] + ts * cl.dw_step;
          mk |= (((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) ? 1u : 0u) << b;
        }
      amask[i] = mk;
      // (rows past M carry ih0 = -2^28: no valid tap, the base below is never dereferenced)
      aptr[i] = p.x + ((int64_t)(pbase[i] + ih0[i] * p.IW + iw0[i]) * p.x_ld + lslot * 8);
    }
  }
  auto stage_fast = [&](int st) {
    if constexpr (ORD == 1) {
      const int64_t toff = ((int64_t)(f_tr * cl.dh_step) * p.IW + f_ts * cl.dw_step) * p.x_ld + o_c;  // wave-uniform
      const unsigned bit = 1u << o_t;
      unsigned char* const sA1 = smem + st * ST_BYTES;
      unsigned char* const sB1 = sA1 + A_BYTES;
      if (ABL != 5) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const h16_t* src = (amask[i] & bit) ? aptr[i] + toff : zero + lslot * 8;
          CVHIP_GLDS16(src, sA1 + (i * RPT + swave * RPI) * ROWB);
        }
      }
      const int boff = o_t * Cin + o_c;  // weight column of (tap, chunk)
      if (ABL != 4) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) CVHIP_GLDS16(bptr[i] + boff, sB1 + (i * RPT + swave * RPI) * ROWB);
      }
      ++o_t;
      if (++f_ts == TS) {
        f_ts = 0;
        ++f_tr;
      }
      if (o_t == ntaps) {
        o_t = 0;
        f_tr = f_ts = 0;
        o_c += BK;
      }
      return;
    }
    if (f_c == 0) {  // first K step of a tap: halo test + pixel address, once per Cin / BK steps
      const int dh = f_tr * cl.dh_step, dw = f_ts * cl.dw_step;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int ih = ih0[i] + dh, iw = iw0[i] + dw;
        const bool ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
        const int64_t off = (int64_t)(pbase[i] + ih * p.IW + iw) * p.x_ld;
        aptr[i] = (ok ? p.x + off : zero) + lslot * 8;
      }
    }
    unsigned char* const sA = smem + st * ST_BYTES;
    unsigned char* const sB = sA + A_BYTES;
    // ABL 3 (experiment, wrong results): pixel rows are fetched for 2 of 9 taps only — the DMA volume a patch-in-LDS layout with
    // tap reuse would have; measures how much of the kernel's time is the issue cost of the A-tile DMA pieces
    const bool a_live = (ABL != 3 || ((f_tr * TS + f_ts) % 6 == 0)) && ABL != 5;  // ABL 5: weight tile only
    if (f_c + BK > Cin) {
      // (wave-uniform, rare) the LAST step of a single-tap plan whose channel count is not a multiple of the step (round 6: the
      // 560 -> 512 1x1 of DeepLabv3+'s decoder ran the general staging path): the 16-byte slots past the last channel come from the
      // zero page, pixel tile and weight tile alike. (No early return: it sent the lambda's by-reference state to scratch memory.)
      const bool dead = f_c + lslot * 8 >= Cin;
      if (a_live) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) CVHIP_GLDS16(dead ? zero : aptr[i], sA + (i * RPT + swave * RPI) * ROWB);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        if (ABL != 4) CVHIP_GLDS16(dead ? zero : bptr[i], sB + (i * RPT + swave * RPI) * ROWB);
      }
    } else {
      if (a_live) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) CVHIP_GLDS16(aptr[i], sA + (i * RPT + swave * RPI) * ROWB);
      }
#pragma unroll
      for (int i = 0; i < A_IT; ++i) aptr[i] += BK;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        if (ABL != 4) CVHIP_GLDS16(bptr[i], sB + (i * RPT + swave * RPI) * ROWB);  // ABL 4: pixel tile only
        bptr[i] += BK;
      }
    }
    f_c += BK;
    if (f_c == Cin) {
      f_c = 0;
      if (++f_ts == TS) {
        f_ts = 0;
        ++f_tr;
      }
    }
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_row = wm * WM + (lane & 15);
  const int b_row = wn * WN + (lane & 15);
  auto compute = [&](int st) __attribute__((always_inline)) {
    const unsigned char* const sA = smem + st * ST_BYTES;
    const unsigned char* const sB = sA + A_BYTES;
    if constexpr (WS && BK == 64) {
      // all fragment reads of the slot first: the second half's reads complete behind the first half's MFMAs (a consumer wave is
      // alone on its SIMD's matrix pipe: nobody else hides its read latency)
      h16x8 xa[2][MF], wb[2][NF];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int slot = (4 * ks + (lane >> 4)) ^ swz_r;
#pragma unroll
        for (int a = 0; a < NF; ++a) wb[ks][a] = *reinterpret_cast<const h16x8*>(sB + (b_row + a * 16) * ROWB + slot * 16);
#pragma unroll
        for (int b = 0; b < MF; ++b) xa[ks][b] = *reinterpret_cast<const h16x8*>(sA + (a_row + b * 16) * ROWB + slot * 16);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
          for (int b = 0; b < MF; ++b)
            acc[a][b] = CVHIP_MFMA_16X16X32(wb[ks][a], xa[ks][b], acc[a][b], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int slot = BK == 32 ? swz_r : ((4 * ks + (lane >> 4)) ^ swz_r);
      h16x8 xa[MF], wb[NF];
#pragma unroll
      for (int b = 0; b < MF; ++b) xa[b] = *reinterpret_cast<const h16x8*>(sA + (a_row + b * 16) * ROWB + slot * 16);
#pragma unroll
      for (int a = 0; a < NF; ++a) wb[a] = *reinterpret_cast<const h16x8*>(sB + (b_row + a * 16) * ROWB + slot * 16);
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b)
          acc[a][b] = CVHIP_MFMA_16X16X32(wb[a], xa[b], acc[a][b], 0, 0, 0);
    }
  };

  if constexpr (WS) {
    if (!consumer) {
      // LOADER: keeps two ring slots in flight; arrives at barrier #kt once slot kt has landed, then refills the slot the
      // consumers finished with before they arrived there (slot kt - 1 = kt + 2 mod 3)
      if (ABL != 2) {
        if (nk > 0) stage_fast(0);
        if (nk > 1) stage_fast(1);
      }
      int st_nxt2 = 2;
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<PER>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (ABL != 2 && kt + 2 < nk) stage_fast(st_nxt2);
        st_nxt2 = st_nxt2 == NST - 1 ? 0 : st_nxt2 + 1;
      }
    } else {
      // CONSUMER: one barrier per ring slot, then nothing but fragment reads and MFMAs
      int st_cur = 0;
      for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        if (ABL != 1) compute(st_cur);
        st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
      }
    }
  } else if constexpr (FAST && NST == 3) {
    // no DMA is issued past the last tile (the running weight pointers would leave the array), so the last step waits for
    // everything instead of "all but the next tile"
    // a class without taps (1x1 stride-2 dgrad: three of the four pixel parities) has nk == 0: nothing to stage, zeros are stored
    if (ABL != 2) {
      if (nk > 0) stage_fast(0);
      if (nk > 1) stage_fast(1);
    }
    int st_cur = 0, st_nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      if (ABL == 4) {
        if (kt + 1 < nk) wait_vmcnt<A_IT>();
        else wait_vmcnt<0>();
      } else if (ABL == 5) {
        if (kt + 1 < nk) wait_vmcnt<B_IT>();
        else wait_vmcnt<0>();
      } else {
        if (kt + 1 < nk) wait_vmcnt<PER>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (ABL != 2 && kt + 2 < nk) stage_fast(st_nxt2);
      if (ABL != 1 && ABL != 4 && ABL != 5) compute(st_cur);
      st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
      st_nxt2 = st_nxt2 == NST - 1 ? 0 : st_nxt2 + 1;
    }
  } else if constexpr (FAST) {
    if (nk > 0) stage_fast(0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) stage_fast((kt + 1) & 1);
      compute(kt & 1);
    }
  } else if constexpr (NST == 3) {
    stage(0, 0);
    stage(1, 1);
    int st_cur = 0, st_nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      // this wave's DMAs of tile kt have landed when at most PER (= tile kt+1) remain outstanding ...
      wait_vmcnt<PER>();
      // ... and after the barrier everybody's have; it also proves all waves finished reading ring slot (kt+2)%3
      __builtin_amdgcn_s_barrier();
      if (ABL != 2) stage(kt + 2, st_nxt2);  // past-the-end tiles decode to all-masked lanes (zero page): DMA counts stay uniform
      if (ABL != 1) compute(st_cur);
      st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
      st_nxt2 = st_nxt2 == NST - 1 ? 0 : st_nxt2 + 1;
    }
  } else {
    // 2-deep ring (40 % less LDS => twice the resident blocks for the narrow-output, bandwidth-bound configurations)
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile kt landed everywhere; everybody finished reading slot (kt+1)&1 (= tile kt-1)
      if (ABL != 2) stage(kt + 1, (kt + 1) & 1);
      if (ABL != 1) compute(kt & 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (all-zero) DMAs must land before smem is reused
  __syncthreads();

  // bias (and the tail layer's BN constants) through the LDS: per-element global loads in the epilogue doubled the time of
  // memory-bound layers (tools/s1x1_bench.py)
  const float* const sbias = reinterpret_cast<const float*>(smem);
  const float* const stail = sbias + BN;  // [4][BN]: scale | shift | mean | invstd of the tail layer's channels n0 .. n0 + BN
  const bool tail = p.tail_y != nullptr;
  // fused epilogue: out = act((acc + bias) * ep_scale + ep_shift); its constants share the tail layer's LDS rows (never both)
  constexpr bool ep_on = EPI;  // (the host never combines it with a tail)
  if (p.bias || tail || ep_on) {
    if (t < BN) {
      float* const w = reinterpret_cast<float*>(smem);
      if (p.bias) w[t] = (n0 + t < p.bias_n) ? p.bias[n0 + t] : 0.f;
      if (ep_on) {
        const int n = n0 + t < p.Nout ? n0 + t : p.Nout - 1;
        w[BN + t] = p.ep_scale ? p.ep_scale[n] : 1.f;
        w[2 * BN + t] = p.ep_scale ? p.ep_shift[n] : 0.f;
      }
      if (tail) {
        const int n = n0 + t < p.Nout ? n0 + t : p.Nout - 1;
        w[BN + t] = p.tail_scale[n];
        w[2 * BN + t] = p.tail_shift[n];
        w[3 * BN + t] = p.tail_mean[n];
        w[4 * BN + t] = p.tail_invstd[n];
      }
    }
    __syncthreads();
  }

  const int nq = (lane >> 4) * 4;
  // Nout % 4 == 0 makes every lane's 4-channel group all-valid or all-invalid, so the packed path has no lane-divergent branch
  const bool vec4 = p.y_vec_ok && (p.Nout & 3) == 0;
  const bool rvec = p.res && vec4 && (p.res_ld & 3) == 0 && ((((uintptr_t)p.res) & 7) == 0);
  const int tact = p.tail_act;
  const float tap = p.tail_ap;
  float ts1[NF][4], ts2[NF][4];  // tail: this lane's share of (sum du, sum du * xhat) per channel
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) ts1[a][r] = ts2[a][r] = 0.f;
  // always_inline: an out-of-line instance would take `p` and the accumulators by address — the whole argument block and the
  // accumulator tile then live in scratch memory (measured: 600 B of scratch per lane, the step 30 % slower)
  auto epilogue = [&](auto vec_c) __attribute__((always_inline)) {
    constexpr bool VEC = decltype(vec_c)::value;
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      const int m = m0 + wm * WM + b * 16 + (lane & 15);
      if (m >= M) continue;
      const int n_img = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
      const int rem = m - n_img * OHWi;
      const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
      const int ow = rem - oh * OWi;
      const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
      h16_t* yrow = p.y + opix * p.y_ld;
      const h16_t* rbase = p.res ? p.res + opix * p.res_ld : nullptr;
#pragma unroll
      for (int a = 0; a < NF; ++a) {
        const int n = n0 + wn * WN + a * 16 + nq;
        if (n >= p.Nout) continue;
        float v0 = acc[a][b][0], v1 = acc[a][b][1], v2 = acc[a][b][2], v3 = acc[a][b][3];
        if (p.bias) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + wn * WN + a * 16 + nq);
          v0 += bv[0];
          v1 += bv[1];
          v2 += bv[2];
          v3 += bv[3];
        }
        if constexpr (ep_on) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(stail + wn * WN + a * 16 + nq), tv = *reinterpret_cast<const f32x4*>(stail + BN + wn * WN + a * 16 + nq);
          float ev[4] = {v0 * sv[0] + tv[0], v1 * sv[1] + tv[1], v2 * sv[2] + tv[2], v3 * sv[3] + tv[3]};
          if (p.res && p.res_pre) {  // residual before the activation (ResNet bottleneck tail)
            const h16_t* rrow = rbase + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.Nout) ev[r] += (float)rrow[r];
          }
          ig_act_vec<4>(ev, p.ep_act, p.ep_ap);
          v0 = ev[0];
          v1 = ev[1];
          v2 = ev[2];
          v3 = ev[3];
        }
        if (p.res && !(ep_on && p.res_pre)) {  // skip-connection gradient folded into dgrad's epilogue (replaces autograd's accumulation add)
          const h16_t* rrow = rbase + n;
          if (VEC && rvec) {
            const uint2 u = *reinterpret_cast<const uint2*>(rrow);  // 4 consecutive channels, like the store below
            float r0, r1, r2, r3;
            unpack2(u.x, r0, r1);
            unpack2(u.y, r2, r3);
            v0 += r0;
            v1 += r1;
            v2 += r2;
            v3 += r3;
          } else {
            v0 += (float)rrow[0];
            if (n + 1 < p.Nout) v1 += (float)rrow[1];
            if (n + 2 < p.Nout) v2 += (float)rrow[2];
            if (n + 3 < p.Nout) v3 += (float)rrow[3];
          }
        }
        if (VEC) {
          uint2 u;
          u.x = pack2(v0, v1);
          u.y = pack2(v2, v3);
          *reinterpret_cast<uint2*>(yrow + n) = u;
          if (tail) {  // block-uniform
            // the sums are taken over the values the tail layer's backward will read: the ROUNDED dz just stored
            float d[4], yv[4];
            unpack2(u.x, d[0], d[1]);
            unpack2(u.y, d[2], d[3]);
            const uint2 uy = *reinterpret_cast<const uint2*>(p.tail_y + opix * p.tail_y_ld + n);
            unpack2(uy.x, yv[0], yv[1]);
            unpack2(uy.y, yv[2], yv[3]);
            const int cl4 = wn * WN + a * 16 + nq;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(stail + cl4), sh = *reinterpret_cast<const f32x4*>(stail + BN + cl4);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(stail + 2 * BN + cl4), is = *reinterpret_cast<const f32x4*>(stail + 3 * BN + cl4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float du = d[r] * act_bwd(yv[r] * sc[r] + sh[r], tact, tap);
              ts1[a][r] += du;
              ts2[a][r] += du * ((yv[r] - mu[r]) * is[r]);
            }
          }
        } else {
          yrow[n] = (h16_t)v0;
          if (n + 1 < p.Nout) yrow[n + 1] = (h16_t)v1;
          if (n + 2 < p.Nout) yrow[n + 2] = (h16_t)v2;
          if (n + 3 < p.Nout) yrow[n + 3] = (h16_t)v3;
        }
      }
    }
  };
  // ---- staged epilogue: the output tile goes through the LDS and leaves as 16-byte stores, 16 consecutive lanes per pixel row -----
  // The direct form below stores 8 bytes per lane in the MFMA fragment layout: 32 bytes contiguous per pixel row and instruction,
  // 32 store instructions per lane for a 128x64 wave tile — store-ISSUE bound (profiles/r03_igemm_ablation.log: the kernel without
  // its stores is 17-19 % faster). Here every lane writes its 4-channel groups into a [BM][BN] bf16 tile in the (now idle) ring
  // memory (row pitch BN*2 + 16 bytes: the 16 pixel rows of a fragment land on 16 different 4-bank groups), and after one barrier
  // the tile leaves row by row: one 16-byte store per lane, whole pixel rows contiguous. Same values, same rounding.
  constexpr int EP_PITCH = BN * 2 + 16;
  constexpr int EP_BASE = 5 * BN * (int)sizeof(float);  // behind the bias / tail constants
  constexpr bool CAN_STAGE = EP_BASE + BM * EP_PITCH <= NST * ST_BYTES;
  const bool staged = CAN_STAGE && ABL == 0 && !tail && p.staged_epilogue && (p.Nout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0);
  if (staged) {
    unsigned char* const tile = smem + EP_BASE;
    // Addend through the LDS (round 6): the skip-connection gradient of dgrad_add was read in the MFMA fragment layout — 8 bytes per
    // lane, 32 contiguous bytes per pixel row and instruction, MF * NF dependent-address loads per lane — and cost 1.3 - 2x its own HBM
    // time ON TOP of the plain input gradient (tools/dgrad_add_bench.py). Here the block first copies its BM x BN addend tile into the
    // output tile's LDS rows with the store loop's own coalesced pattern (16 bytes per lane, whole pixel rows, every load of a batch
    // independent), and each lane then takes its 4-channel groups from there: same values, same single rounding of acc + addend.
    constexpr int CPR_R = BN / 8;
    constexpr int RES_IT = BM * CPR_R / (NW * 64);
    const bool res_lds = p.staged_epilogue == 2 && p.res && !ep_on && rvec && (p.res_ld & 7) == 0 && ((((uintptr_t)p.res) & 15) == 0);
    if (res_lds) {
      constexpr int RB = BN >= 128 ? (RES_IT < 8 ? RES_IT : 8) : (RES_IT < 4 ? RES_IT : 4);   // loads in flight per lane (the narrow tiles run at a 128-register cap)
#pragma unroll
      for (int it0 = 0; it0 < RES_IT; it0 += RB) {
        uint4 rv[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int idx = t + (it0 + j) * (NW * 64);
          const int row = idx / CPR_R, ch = idx - row * CPR_R;
          const int m = m0 + row;
          rv[j] = uint4{0u, 0u, 0u, 0u};
          if (m < M && n0 + ch * 8 < p.Nout) {
            const int n_img = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
            const int rem = m - n_img * OHWi;
            const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
            const int ow = rem - oh * OWi;
            const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
            rv[j] = *reinterpret_cast<const uint4*>(p.res + opix * p.res_ld + n0 + ch * 8);
          }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int idx = t + (it0 + j) * (NW * 64);
          const int row = idx / CPR_R, ch = idx - row * CPR_R;
          *reinterpret_cast<uint4*>(tile + row * EP_PITCH + ch * 16) = rv[j];
        }
      }
      __syncthreads();
    }
    if (consumer) {
#pragma unroll
      for (int b = 0; b < MF; ++b) {
        const int row = wm * WM + b * 16 + (lane & 15);
        const int m = m0 + row;
        const h16_t* rbase = nullptr;
        if (p.res && !res_lds && m < M) {
          const int n_img = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
          const int rem = m - n_img * OHWi;
          const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
          const int ow = rem - oh * OWi;
          const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
          rbase = p.res + opix * p.res_ld;
        }
#pragma unroll
        for (int a = 0; a < NF; ++a) {
          const int nl = wn * WN + a * 16 + nq;
          float v0 = acc[a][b][0], v1 = acc[a][b][1], v2 = acc[a][b][2], v3 = acc[a][b][3];
          if (p.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + nl);
            v0 += bv[0];
            v1 += bv[1];
            v2 += bv[2];
            v3 += bv[3];
          }
          if constexpr (ep_on) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(stail + nl), tv = *reinterpret_cast<const f32x4*>(stail + BN + nl);
            float ev[4] = {v0 * sv[0] + tv[0], v1 * sv[1] + tv[1], v2 * sv[2] + tv[2], v3 * sv[3] + tv[3]};
            if (rbase && p.res_pre && n0 + nl < p.Nout) {  // (the staged form needs Nout % 8 == 0: the 4 channels exist)
              const h16_t* rrow = rbase + n0 + nl;
#pragma unroll
              for (int r = 0; r < 4; ++r) ev[r] += (float)rrow[r];
            }
            ig_act_vec<4>(ev, p.ep_act, p.ep_ap);
            v0 = ev[0];
            v1 = ev[1];
            v2 = ev[2];
            v3 = ev[3];
          }
          if (res_lds) {  // (block-uniform) this lane's 4 addend channels from the tile rows filled above; the sum goes back to the same 8 bytes
            const uint2 u = *reinterpret_cast<const uint2*>(tile + row * EP_PITCH + nl * 2);
            float r0, r1, r2, r3;
            unpack2(u.x, r0, r1);
            unpack2(u.y, r2, r3);
            v0 += r0;
            v1 += r1;
            v2 += r2;
            v3 += r3;
          } else if (rbase && !(ep_on && p.res_pre) && n0 + nl < p.Nout) {
            const h16_t* rrow = rbase + n0 + nl;
            if (rvec) {
              const uint2 u = *reinterpret_cast<const uint2*>(rrow);
              float r0, r1, r2, r3;
              unpack2(u.x, r0, r1);
              unpack2(u.y, r2, r3);
              v0 += r0;
              v1 += r1;
              v2 += r2;
              v3 += r3;
            } else {
              v0 += (float)rrow[0];
              v1 += (float)rrow[1];
              v2 += (float)rrow[2];
              v3 += (float)rrow[3];
            }
          }
          uint2 u;
          u.x = pack2(v0, v1);
          u.y = pack2(v2, v3);
          *reinterpret_cast<uint2*>(tile + row * EP_PITCH + nl * 2) = u;
        }
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
    for (int idx = t; idx < BM * CPR; idx += NW * 64) {
      const int row = idx / CPR, ch = idx - row * CPR;
      const int m = m0 + row;
      if (m >= M || n0 + ch * 8 >= p.Nout) continue;
      const int n_img = (int)fast_div31((unsigned)m, cl.ohw_mul, cl.ohw_sh);
      const int rem = m - n_img * OHWi;
      const int oh = (int)fast_div31((unsigned)rem, cl.ow_mul, cl.ow_sh);
      const int ow = rem - oh * OWi;
      const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
      *reinterpret_cast<uint4*>(p.y + opix * p.y_ld + n0 + ch * 8) = *reinterpret_cast<const uint4*>(tile + row * EP_PITCH + ch * 16);
    }
  } else if (ABL == 6) {  // ablation: no output stores (what the epilogue costs); one store keeps the accumulators alive
    float sacc = 0.f;
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int b = 0; b < MF; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sacc == 123.456f) p.y[0] = (h16_t)sacc;
  } else if (consumer) {  // (loaders hold no accumulators; they only take part in the barriers below)
    if (vec4) epilogue(std::true_type{});  // (the host only sets a tail when the packed path applies)
    else epilogue(std::false_type{});
  }

  if (p.stats) {
    if (p.bias || tail || ep_on || staged) __syncthreads();  // the constants / the output tile staged above are dead now
    float* red = reinterpret_cast<float*>(smem);  // [WAVES_M][BN][2]
#pragma unroll
    for (int a = 0; a < NF; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
        if (tail) {
          s1 = ts1[a][r];
          s2 = ts2[a][r];
        } else {
#pragma unroll
          for (int b = 0; b < MF; ++b) {
            const float v = acc[a][b][r];
            s1 += v;
            s2 += v * v;
          }
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (consumer && (lane & 15) == 0) {
          const int nl = wn * WN + a * 16 + nq + r;
          red[(wm * BN + nl) * 2 + 0] = s1;
          red[(wm * BN + nl) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if (t < BN) {
      const int n = n0 + t;
      if (n < p.Nout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          s1 += red[(w * BN + t) * 2 + 0];
          s2 += red[(w * BN + t) * 2 + 1];
        }
        if (p.stats_acc) {
          acc_add2(reinterpret_cast<double*>(p.stats), mtile, p.stats_ld, n, s1, s2);
        } else {
          float* dst = p.stats + (int64_t)mtile * 2 * p.Nout;
          dst[n] = s1;
          dst[p.Nout + n] = s2;
        }
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------

// Launch policy (every value below was an A/B switch while it was being measured; DESIGN.md 4.0 / 4.00 hold the numbers):
//   * LDS ring depth: 2-deep for the narrow-output configurations (BN <= 64: bandwidth / latency-bound, twice the resident blocks:
//     -0.6 ms per YOLOv5-s step), 3-deep elsewhere;
//   * FAST staging (running DMA pointers, one tap decode per tap) wherever Cin % 32 == 0;
//   * staged epilogue (output tile through the LDS, 16-byte row stores);
//   * stride-parity classes of one spatial tile get consecutive tile ids (the half-line writes of interleaved pixels merge in one L2).
template <int BM, int BN, int WM, int WN>
static int launch_group(IgemmKernArgs& p, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < p.ncls; ++i) {
    p.cls[i].tile_begin = total;
    total += cdiv(p.cls[i].M, BM) * p.n_tiles;
  }
  p.total_tiles = total;
  if (total == 0) return CVHIP_OK;
  // 2 = staged, and the addend of dgrad_add goes through the LDS tile too (kernel: res_lds). Measured on rotating operands
  // (profiles/r06_dgrad_add_res_lds.log): dgrad_add 128 -> 256 k3 s2 @80 157 -> 139 us, 32 -> 64 k3 s2 @320 371 -> 302,
  // DeepLabv3+ 256 -> 512 k1 s2 @128x256 302 -> 247, 1024 -> 256 k1 @32x64 60 -> 51. CVHIP_IGEMM_RES_LDS=0 restores the direct read (A/B).
  p.staged_epilogue = 2;
  {
    const char* e = getenv("CVHIP_IGEMM_RES_LDS");
    if (e && atoi(e) == 0) p.staged_epilogue = 1;
  }
  p.interleave = 0;
  if (p.ncls > 1) {
    bool same = true;
    for (int i = 1; i < p.ncls; ++i) same = same && cdiv(p.cls[i].M, BM) == cdiv(p.cls[0].M, BM);
    p.interleave = same ? 1 : 0;
    {
      // Class order (round 6). The hardware hands consecutive workgroups to the CUs of an XCD in turn, so with (spatial tile, class)
      // order the heavy class of a 3x3 stride-2 plan (4 taps against 1 / 2 / 2) — or the ONE class with a tap of a 1x1 stride-2 plan —
      // kept landing on the same quarter of the CUs. Groups of 16 spatial tiles, inside a group class-major with the heaviest class
      // first: every CU gets its share of every class, longest first, and the four classes of a spatial tile still run within
      // 64 tile ids of each other on one XCD (their interleaved half-line writes and their shared dy tile still meet in that L2).
      // profiles/r06_dgrad_class_order.log: dgrad_add 128 -> 256 k3 s2 @80 132 -> 109 us, 256 -> 512 @40 119 -> 95, DeepLabv3+
      // 256 -> 512 k1 s2 @128x256 dgrad 212 -> 107. CVHIP_IGEMM_CLASS_ORDER (A/B): 0 = (tile, class) order, 1 = rotated, N >= 2 = group size.
      const char* e = getenv("CVHIP_IGEMM_CLASS_ORDER");
      const int v = e ? atoi(e) : 16;
      if (same && v == 1) p.interleave = 2;
      if (same && v >= 2) {
        p.interleave = 3;
        p.il_group = v;
        p.il_tiles = cdiv(p.cls[0].M, BM) * p.n_tiles;
        int ord[kKernelClasses];
        for (int i = 0; i < p.ncls; ++i) ord[i] = i;
        for (int i = 0; i < p.ncls; ++i)   // heaviest class (most taps) first
          for (int j = i + 1; j < p.ncls; ++j)
            if (p.cls[ord[j]].TR * p.cls[ord[j]].TS > p.cls[ord[i]].TR * p.cls[ord[i]].TS) {
              const int tmp = ord[i];
              ord[i] = ord[j];
              ord[j] = tmp;
            }
        p.il_order = 0;
        for (int i = 0; i < p.ncls; ++i) p.il_order |= ord[i] << (4 * i);
      }
    }
  }
  constexpr bool nst2 = BN <= 64;
  constexpr int NST = nst2 ? 2 : 3;
  // FAST staging: the reduction step lies inside one tap — any Cin % 32 == 0, or a single-tap plan (1x1) whose last step is partial
  bool one_tap = true;
  for (int i = 0; i < p.ncls; ++i) one_tap = one_tap && p.cls[i].TR * p.cls[i].TS <= 1;
  const bool fast = (p.Cin % 32 == 0 || (one_tap && p.Cin % 8 == 0)) && p.Cin <= kFastMaxCin;
  if (p.ep_scale || p.ep_act != CVHIP_ACT_NONE) {
    // fused epilogue: the EPI instances of the same forms
    if (p.tail_y) return CVHIP_ERR_INVALID;
    if (fast) hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, 0, NST, 32, true, 4, false, 0, true>), dim3(total), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, 0, NST, 32, false, 4, false, 0, true>), dim3(total), dim3(256), 0, stream, p);
    return check_launch("igemm_kernel(fused epilogue)");
  }
  if (fast) hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, 0, NST, 32, true>), dim3(total), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((igemm_dma_kernel<BM, BN, WM, WN, 0, NST>), dim3(total), dim3(256), 0, stream, p);
  return check_launch("igemm_kernel");
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(IgemmParams& p, hipStream_t stream) {
  p.n_tiles = cdiv(p.Nout, BN);
  p.cin_magic = div_magic(p.Cin);
  for (int i = 0; i < p.ncls; ++i) {
    p.cls[i].ts_magic = div_magic(p.cls[i].TS);
    div31_consts(p.cls[i].OHi * p.cls[i].OWi, &p.cls[i].ohw_mul, &p.cls[i].ohw_sh);
    div31_consts(p.cls[i].OWi, &p.cls[i].ow_mul, &p.cls[i].ow_sh);
    if ((int64_t)p.cls[i].TR * p.cls[i].TS * p.Cin >= 65536) return CVHIP_ERR_UNSUPPORTED;  // 16-bit exact fast division
  }
  // the kernels take kKernelClasses classes by value (small argument block); plans with more (stride > 2) launch in groups
  for (int first = 0; first < p.ncls; first += kKernelClasses) {
    const int count = p.ncls - first < kKernelClasses ? p.ncls - first : kKernelClasses;
    IgemmKernArgs k = narrow_plan(p, first, count);
    const int st = launch_group<BM, BN, WM, WN>(k, stream);
    if (st) return st;
  }
  return CVHIP_OK;
}

// Tile choice. Wide outputs (> 64 channels) of large problems use 256x128 block tiles with 128x64 WAVE tiles: LDS fragment
// reads per MFMA drop by 25 % (12 ds_read_b128 per 32 MFMAs instead of 8 per 16) and global->LDS bytes per flop by 25 %
// — the two limits profiles/r01_igemm_ablation.log and r01_lds_read_bw_probe.log measure. Small problems keep 128x128
// so the grid still covers the 256 CUs.
int igemm_block_m(int Nout, int64_t M, int Ktot) {
  if (Nout <= 64) return 256;
  // shallow reductions (1x1 convs with < 512 input channels) are memory-bound: more, smaller blocks hide latency better
  // (measured per shape: gpurun conv_table A/B)
  if (Ktot < 512) return 128;
  const int64_t tiles256 = ((M + 255) / 256) * ((Nout + 127) / 128);
  return tiles256 >= 384 ? 256 : 128;
}

int launch_igemm(IgemmParams& p, hipStream_t stream) {
  if ((p.ep_scale == nullptr) != (p.ep_shift == nullptr)) return CVHIP_ERR_INVALID;
  if (p.stats && (p.ep_scale || p.ep_act != CVHIP_ACT_NONE)) return CVHIP_ERR_INVALID;  // BN sums are those of the raw accumulators
  const int s0 = try_launch_stem(p, stream);  // 8-channel image stem: direct convolution from an LDS patch (conv_stem.hip)
  if (s0 >= 0) return s0;
  const int s1 = try_launch_stream1x1(p, stream);  // 1x1 / stride 1: persistent streaming kernel (conv1x1_stream.hip)
  if (s1 >= 0) return s1;
  const int sb = try_launch_band(p, stream);  // 3x3 stride 1, Cin % 32 == 0: row bands, weights in registers (conv_band.hip)
  if (sb != -1) return sb;
  const int s2 = try_launch_patch(p, stream);  // multi-tap, Cin % 32 == 0: patch-resident implicit GEMM (conv_patch.hip)
  if (s2 != -1) return s2;
  if (p.pro_scale || p.z_out || p.y2) return CVHIP_ERR_UNSUPPORTED;  // a prologue: patch / streaming kernels only; a split store: streaming kernel only
  if (p.Nout <= 32) return launch_cfg<256, 32, 64, 32>(p, stream);
  if (p.Nout <= 64) return launch_cfg<256, 64, 64, 64>(p, stream);
  int64_t M = 0;
  int ktot = 0;
  for (int i = 0; i < p.ncls; ++i) {
    M += p.cls[i].M;
    const int k = p.cls[i].TR * p.cls[i].TS * p.Cin;
    ktot = k > ktot ? k : ktot;
  }
  // stride-parity plans (dgrad of a strided convolution): 128-row tiles — three co-resident blocks per CU overlap the short per-class K
  // loops with the other blocks' store epilogues (profiles/r06_dgrad_class_order.log, columns bm128 / bm256: 128 -> 128 k3 s2 @80
  // 52.3 vs 61.6 us, DeepLabv3+ 256 -> 512 k1 s2 dgrad_add 150.8 vs 161.2, level elsewhere)
  if (p.ncls > 1 && !p.stats) return launch_cfg<128, 128, 64, 64>(p, stream);
  if (igemm_block_m(p.Nout, M, ktot) == 256) return launch_cfg<256, 128, 128, 64>(p, stream);
  return launch_cfg<128, 128, 64, 64>(p, stream);
}

}  // namespace cvhip
