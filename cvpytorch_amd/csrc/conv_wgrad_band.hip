// conv_wgrad_band.hip — tap-resident weight gradient of stride-1 3x3 "same" convolutions (pad == dilation) for gfx950 (CDNA4).
//
//   dW[k][t*Cin + c] += sum_m dY[m][k] * X[pix(m) + tap(t)][c]          (fp32, KRSC layout; m = (image, row, column) of the output)
//
// Why a second weight-gradient kernel (round 6; VERDICT r05 "what's weak" 2: wgrad_kernel<128,64,64> at 0.18 of the MFMA peak). The
// general kernel (conv_wgrad.hip) tiles the GEMM as [K] x [128 columns of (tap, c)]: for a 3x3 layer every one of the nine tap
// tiles stages the SAME dY rows and the same x pixels (shifted) through the LDS again — 472 MB of L2 -> LDS traffic for the 52 MB
// of operands of 128 -> 128 @40x40 — with one block barrier per 32 pixels (16 MFMAs per wave). Making the tile cover all nine taps
// the way the forward band kernel does fails on the OTHER bound of this GEMM: every resident block ends by flushing its whole
// accumulator tile with fp32 atomics (~1.4 TB/s on this chip), so 256 blocks x (64 x 64 x 9) outputs are 38 MB of atomics per launch.
// Here the two are decoupled:
//   * a block owns an output tile of KT (64 or 32) output channels x 32 input channels x ALL NINE taps (73 / 37 KB of fp32) and a
//     contiguous range of output pixels (a pixel split);
//   * its 8 waves = 2 input-channel halves x FOUR PIXEL REPLICAS: the replicas hold the same 16 KF x 16 x 9 accumulator tile
//     (144 registers at KF = 4) and take different 32-pixel steps, and are folded through the LDS (a fixed binary tree) before ONE
//     replica flushes: 256 blocks flush 19 MB instead of 75, while every CU still runs 8 x 36 MFMAs per step;
//   * the pixel range is walked in RANGES of 256 output pixels (8 steps: two per replica): the range's dY rows (256 x KT) and the
//     input patch that its taps reach (whole image rows, one 32-channel chunk, 64-byte pixels) are staged ONCE by LDS-DMA,
//     double-buffered, ONE barrier per range (72 MFMAs per wave), and all nine taps read their fragments from the patch — the
//     staged bytes per launch are (KT + ~1.4 x 32) x 2 per pixel and tile: 157 MB for 128 -> 128 @40x40 instead of 472;
//   * pixels are the REDUCTION axis: both MFMA operands are gathered with ds_read_b64_tr_b16 from pixel-major LDS rows, every lane
//     addressing its own pixel row — so a range is simply 256 consecutive output pixels in (image, row, column) order: it may start
//     mid-row and run across image boundaries. The patch is addressed through a VIRTUAL TALL IMAGE: the batch stacked with G = dil
//     zero rows between images, so that the row above the first / below the last row of an image is a (shared) zero row and one
//     affine map pixel -> patch pixel serves every tap.
// K-element order of a step (the same for both operands, so any order is legal): instruction h of lane group g reads pixels
// 16h + 4g + q (q = 0..3) — lanes 0..31 of one ds_read touch 8 CONSECUTIVE pixels, which the swizzles below make conflict-free.
// LDS images: dY rows of KT x 2 bytes, 32-byte segment s of pixel pl at s ^ ((pl >> 1) & 3) (KT 64) / s ^ ((pl >> 2) & 1) (KT 32);
// x patch = the forward band kernel's image (64-byte pixels, 32-byte half at half ^ ((pp >> 2) & 1), row pitch PW % 8 == 0).
// LDS-DMA writes lane-linearly, so each lane FETCHES the logical chunk that belongs at its physical position (rule 21).
//
// Replaces aten::convolution_backward(weight) reached from trainer.py:189 (loss.backward()) for DarknetBottleneck conv2
// (modules/yolo_modules.py:95-104), the YOLOX head towers and torchvision Bottleneck conv2.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

constexpr int kWbWaves = 8;             // 512 threads: one block per CU, two waves per SIMD
constexpr int kWbReps = 4;              // pixel replicas (waves 2r, 2r + 1 = the two 16-channel halves of replica r)
constexpr int kWbJ = 2;                 // steps per replica and range (the main loop is written for 2)
constexpr int kWbRange = 32 * kWbReps * kWbJ;   // 256 output pixels
constexpr int kWbXPieces = 8;           // patch DMA instructions per wave and range (<= 64 KB of patch per buffer)
constexpr int kWbLdsMax = 159 * 1024;

struct WgBandArgs {
  const h16_t* x;
  const h16_t* dy;
  float* dw;
  int NB, OH, OW, Cin, x_ld;
  int K, dy_ld, Ktot;
  int dil_h, dil_w;
  int G, VP;           // zero rows between images of the virtual tall image; its image pitch OH + G
  int PW;              // patch row pitch (pixels)
  int nxp;             // KB pieces of one patch buffer
  int xbuf, dbuf;      // bytes of one patch / one dY buffer
  int M;
  int c_tiles, tiles, ranges_per_split;
  unsigned ow_mul, ow_sh, oh_mul, oh_sh, vp_mul, vp_sh, pw_mul, pw_sh;
};

__device__ __attribute__((aligned(64))) unsigned int g_wgband_zero[16];

#define CVHIP_WB_GLDS16(src, dst)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// Fragment gathers by hand. hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 it emits for the
// builtin while an LDS-DMA is outstanding (it does not for a plain ds_read_b128: measured on a two-line kernel) — i.e. the next
// range's staging would be drained before the first fragment of this range is read. As asm the reads are outside its bookkeeping;
// the lgkmcnt waits are counted by hand (LDS operations return in order: "at most N outstanding" = all but the N youngest landed)
// and carry the fragment as an in/out operand, so no MFMA that uses it can be scheduled above the wait.
typedef unsigned int wb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wb_tr_read(wb_u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wb_lwait(wb_u32x4& a) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt literal");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
__device__ __forceinline__ wb_u32x4 wb_join(const wb_u32x2& lo, const wb_u32x2& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }

// KF: 16-channel dY fragments per wave (KT = 16 * KF output channels per block tile)
template <int KF>
__global__ __launch_bounds__(kWbWaves * 64, 2) void wgrad_band_kernel(const WgBandArgs p) {
  static_assert(KF == 2 || KF == 4, "output-channel tile of 32 or 64");
  constexpr int KT = 16 * KF;
  constexpr int D_ROWB = KT * 2;                  // dY LDS row bytes
  constexpr int D_RPP = 1024 / D_ROWB;            // dY rows per DMA instruction: 8 / 16
  constexpr int D_PW = kWbRange / D_RPP / kWbWaves;   // dY DMA instructions per wave and range: 4 / 2
  constexpr int D_BYTES = kWbRange * D_ROWB;
  constexpr int ACCN = KF * 9;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ch = wave & 1, rep = wave >> 1;
  const int g = lane >> 4, q4 = (lane >> 2) & 3;

  // all tiles of one pixel split get consecutive logical ids: the same XCD, so the split's x / dY slab is fetched into ONE L2
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / p.tiles;
  const int tile = lin - split * p.tiles;
  const int ktile = tile / p.c_tiles, ctile = tile - ktile * p.c_tiles;
  const int k0 = ktile * KT, c0 = ctile * 32;
  const int m_begin = split * p.ranges_per_split * kWbRange;
  const int m_end = min(p.M, m_begin + p.ranges_per_split * kWbRange);
  if (m_end <= m_begin) return;
  const int nranges = (m_end - m_begin + kWbRange - 1) / kWbRange;
  const int bufstride = D_BYTES + p.xbuf;   // [dY | patch] per buffer
  const h16_t* const zero = reinterpret_cast<const h16_t*>(g_wgband_zero);

  // ---- patch loader geometry (range-independent): DMA instruction k of this wave fills patch pixels (k*8 + wave)*16 .. +15, lane l
  // the PHYSICAL 16-byte slot l & 3 of pixel l >> 2 and fetches the logical slot that belongs there
  const int lsl = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  int prc[kWbXPieces];   // (patch row << 16) | input column, or -1 (padding column / past the buffer)
#pragma unroll
  for (int k = 0; k < kWbXPieces; ++k) {
    const int piece = k * kWbWaves + wave;
    const int pp = piece * 16 + (lane >> 2);
    const int pr = (int)fast_div31((unsigned)pp, p.pw_mul, p.pw_sh);
    const int iw = pp - pr * p.PW - p.dil_w;
    prc[k] = (piece < p.nxp && (unsigned)iw < (unsigned)p.OW) ? ((pr << 16) | iw) : -1;
  }
  const h16_t* const xlane = p.x + c0 + lsl * 8;
  // ---- dY loader geometry: KT 64: 8 rows of 8 chunks per instruction, KT 32: 16 rows of 4 chunks
  const int d_row = KT == 64 ? (lane >> 3) : (lane >> 2);
  const int d_slot = KT == 64 ? (lane & 7) : (lane & 3);
  const int d_f = KT == 64 ? ((lane >> 4) & 3) : ((lane >> 4) & 1);        // f(pl) of this lane's row: piece * D_RPP is a multiple of 8 / 16
  const int d_kf = ((d_slot >> 1) ^ d_f) & (KF - 1);
  const h16_t* const dylane = p.dy + k0 + d_kf * 16 + (d_slot & 1) * 8;

  int vbase_cur = 0;   // v(first output row of the range being multiplied), set by issue_range for the NEXT range and rotated below
  auto range_rows = [&](int q0, int* vbase, int* phr) __attribute__((always_inline)) {
    const int qlast = min(q0 + kWbRange, m_end) - 1;
    const int gr0 = (int)fast_div31((unsigned)q0, p.ow_mul, p.ow_sh);
    const int grl = (int)fast_div31((unsigned)qlast, p.ow_mul, p.ow_sh);
    const int n0 = (int)fast_div31((unsigned)gr0, p.oh_mul, p.oh_sh);
    const int nl = (int)fast_div31((unsigned)grl, p.oh_mul, p.oh_sh);
    *vbase = gr0 + n0 * p.G;
    *phr = (grl + nl * p.G) - *vbase + 1 + 2 * p.dil_h;
  };
  auto issue_range = [&](int rg, unsigned char* buf) __attribute__((always_inline)) {
    const int q0 = m_begin + rg * kWbRange;
    int vbase, phr;
    range_rows(q0, &vbase, &phr);
    const int v0 = vbase - p.dil_h;
    unsigned char* const db = buf;
    unsigned char* const xb = buf + D_BYTES;
#pragma unroll
    for (int i = 0; i < D_PW; ++i) {
      const int piece = i * kWbWaves + wave;
      const int m = q0 + piece * D_RPP + d_row;
      const h16_t* src = m < m_end ? dylane + (int64_t)m * p.dy_ld : zero;
      CVHIP_WB_GLDS16(src, db + piece * 1024);
    }
    const int npix = phr * p.PW;
#pragma unroll
    for (int k = 0; k < kWbXPieces; ++k) {
      const int piece = k * kWbWaves + wave;
      if (piece * 16 < npix) {   // wave-uniform: pieces wholly past this range's patch rows are not fetched (and never read)
        const int e = prc[k];
        const int pr = e >> 16, iw = e & 0xffff;
        const int v = v0 + pr;
        bool ok = e >= 0 && pr < phr && v >= 0;
        const int vv = ok ? v : 0;
        const int n = (int)fast_div31((unsigned)vv, p.vp_mul, p.vp_sh);
        const int ih = vv - n * p.VP;
        ok = ok && ih < p.OH && n < p.NB;
        const h16_t* src = ok ? xlane + (int64_t)((n * p.OH + ih) * p.OW + iw) * p.x_ld : zero;
        CVHIP_WB_GLDS16(src, xb + piece * 1024);
      }
    }
  };

  f32x4 acc[KF][9];
#pragma unroll
  for (int a = 0; a < KF; ++a)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc[a][tp] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane constants of the fragment gathers
  const int fD = KT == 64 ? (2 * (g & 1) + (q4 >> 1)) : (g & 1);   // f(pl) for pl = 32s + 16h + 4g + q4
  const int lane8 = (lane & 3) * 8;
  const int roff = p.dil_h * p.PW * 64;

  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
  // Patch pixel of every output pixel this wave multiplies in a range, computed ONCE per wave: the wave's two steps x two half-steps x
  // 16 pixels are 64 pixels = one per lane (lane = step << 5 | half << 4 | pixel-in-16); the division chain runs once per lane and
  // range, and every lane fetches the four values it gathers from (its pixel-in-16 = 4g + q4) with ds_bpermute_b32
  auto range_pixels = [&](int q0, int vbase, int (&u)[2][2]) __attribute__((always_inline)) {
    const int pl = 32 * (rep + kWbReps * (lane >> 5)) + (lane & 31);
    int m = q0 + pl;
    m = m < m_end ? m : m_end - 1;   // rows past the split: dY is zero there, the patch address only has to be valid
    const int gr = (int)fast_div31((unsigned)m, p.ow_mul, p.ow_sh);
    const int ow = m - gr * p.OW;
    const int n = (int)fast_div31((unsigned)gr, p.oh_mul, p.oh_sh);
    const int ppb = (gr + n * p.G - vbase) * p.PW + ow;
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int h = 0; h < 2; ++h) u[si][h] = __builtin_amdgcn_ds_bpermute(((si << 5) | (h << 4) | (4 * g + q4)) << 2, ppb);
  };
  auto compute_step = [&](int s, const int (&uu)[2], unsigned bufoff) __attribute__((always_inline)) {
    const unsigned db = lds0 + bufoff;
    const unsigned xb = db + D_BYTES;
    unsigned adA[2][KF];
    unsigned aj[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = 32 * s + 16 * h + 4 * g + q4;
#pragma unroll
      for (int a = 0; a < KF; ++a) adA[h][a] = db + pl * D_ROWB + (((a ^ fD) & (KF - 1)) << 5) + lane8;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int u = uu[h] + j * p.dil_w;
        aj[h][j] = xb + (u << 6) + ((ch ^ ((u >> 2) & 1)) << 5) + lane8;
      }
    }
    // reads in flight: the KF dY fragments and the x fragments of three taps; tap t is multiplied once everything older than the
    // reads of taps t + 1, t + 2 has landed, then the reads of tap t + 3 go out (<= 2 * KF + 6 <= 14 outstanding)
    wb_u32x2 alo[KF], ahi[KF];
#pragma unroll
    for (int a = 0; a < KF; ++a) {
      wb_tr_read(alo[a], adA[0][a]);
      wb_tr_read(ahi[a], adA[1][a]);
    }
    wb_u32x2 xlo[3], xhi[3];
    auto xread = [&](int tp, int slot) __attribute__((always_inline)) {
      const int i = tp / 3, j = tp - i * 3;
      wb_tr_read(xlo[slot], aj[0][j] + i * roff);
      wb_tr_read(xhi[slot], aj[1][j] + i * roff);
    };
    xread(0, 0);
    xread(1, 1);
    xread(2, 2);
    h16x8 fd[KF];
    auto tap = [&](auto tc) __attribute__((always_inline)) {
      constexpr int tp = decltype(tc)::value;
      wb_u32x4 fx = wb_join(xlo[tp % 3], xhi[tp % 3]);
      constexpr int YOUNGER = tp + 2 < 9 ? 4 : (tp + 1 < 9 ? 2 : 0);   // reads of the taps issued after this one
      wb_lwait<YOUNGER>(fx);
      if constexpr (tp == 0) {
        // (the dY fragments are older than every x read: they have landed too; naming them here keeps their uses below the wait)
#pragma unroll
        for (int a = 0; a < KF; ++a) {
          wb_u32x4 fa = wb_join(alo[a], ahi[a]);
          asm volatile("" : "+v"(fa));
          fd[a] = __builtin_bit_cast(h16x8, fa);
        }
      }
      const h16x8 fxv = __builtin_bit_cast(h16x8, fx);
#pragma unroll
      for (int a = 0; a < KF; ++a) acc[a][tp] = CVHIP_MFMA_16X16X32(fd[a], fxv, acc[a][tp], 0, 0, 0);
      if constexpr (tp + 3 < 9) xread(tp + 3, tp % 3);
      __builtin_amdgcn_sched_barrier(0);   // (MFMAs and asm statements do not cross: the counted waits stay between the taps)
    };
    __builtin_amdgcn_sched_barrier(0);
    tap(std::integral_constant<int, 0>{});
    tap(std::integral_constant<int, 1>{});
    tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{});
    tap(std::integral_constant<int, 4>{});
    tap(std::integral_constant<int, 5>{});
    tap(std::integral_constant<int, 6>{});
    tap(std::integral_constant<int, 7>{});
    tap(std::integral_constant<int, 8>{});
  };

  // ---- main loop: range rg is multiplied from buffer rg & 1 while range rg + 1 lands in the other ------------------------------------
  issue_range(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int rg = 0; rg < nranges; ++rg) {
    const unsigned cur = (unsigned)((rg & 1) * bufstride);
    const bool more = rg + 1 < nranges;
    // the two waves of a SIMD (replicas r and r + 2) issue their share of the next range's staging at different times — r before its
    // first step, r + 2 between its steps — so that one's address arithmetic and DMA issue run under the other's MFMAs
    const bool late = rep >= 2;
    if (more && !late) issue_range(rg + 1, smem + ((rg + 1) & 1) * bufstride);
    const int q0 = m_begin + rg * kWbRange;
    int phr;
    range_rows(q0, &vbase_cur, &phr);
    int upix[2][2];
    range_pixels(q0, vbase_cur, upix);
    compute_step(rep, upix[0], cur);
    if (more && late) issue_range(rg + 1, smem + ((rg + 1) & 1) * bufstride);
    compute_step(rep + kWbReps, upix[1], cur);
    // the next range has landed (this wave's share; the barrier publishes everybody's) and every wave is done reading `cur`
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- fold the four replicas through the LDS (fixed tree: 2,3 -> 0,1; 1 -> 0), then replica 0 flushes -------------------------------
  f32x4* const park = reinterpret_cast<f32x4*>(smem);
  if (rep >= 2) {
    f32x4* dst = park + ((rep - 2) * 2 + ch) * (ACCN * 64) + lane;
#pragma unroll
    for (int a = 0; a < KF; ++a)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) dst[(a * 9 + tp) * 64] = acc[a][tp];
  }
  __syncthreads();
  if (rep < 2) {
    const f32x4* src = park + (rep * 2 + ch) * (ACCN * 64) + lane;
#pragma unroll
    for (int a = 0; a < KF; ++a)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[a][tp] += src[(a * 9 + tp) * 64];
  }
  __syncthreads();
  if (rep == 1) {
    f32x4* dst = park + ch * (ACCN * 64) + lane;
#pragma unroll
    for (int a = 0; a < KF; ++a)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) dst[(a * 9 + tp) * 64] = acc[a][tp];
  }
  __syncthreads();
  if (rep != 0) return;
  {
    const f32x4* src = park + ch * (ACCN * 64) + lane;
#pragma unroll
    for (int a = 0; a < KF; ++a)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[a][tp] += src[(a * 9 + tp) * 64];
  }
  // lane holds D[k = 4 * (lane >> 4) + r][c = lane & 15] of every (fragment a, tap) tile
  float* const dwl = p.dw + (int64_t)(k0 + 4 * g) * p.Ktot + c0 + ch * 16 + (lane & 15);
#pragma unroll
  for (int a = 0; a < KF; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) unsafeAtomicAdd(dwl + (int64_t)(a * 16 + r) * p.Ktot + tp * p.Cin, acc[a][tp][r]);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------

static int wgband_mode() {  // CVHIP_WGRAD_BAND: 0 = never, 1 = default policy, 2 = wherever the geometry allows (read per launch: in-process A/B)
  const char* e = getenv("CVHIP_WGRAD_BAND");
  return e ? atoi(e) : 1;
}

struct WgBandPlan {
  WgBandArgs a;
  int KF, blocks, lds;
};

static bool wgband_plan(const cvhip_conv_desc* d, WgBandPlan* pl) {
  if (d->R != 3 || d->S != 3 || d->stride_h != 1 || d->stride_w != 1 || d->groups != 1) return false;
  if (d->pad_h != d->dil_h || d->pad_w != d->dil_w || d->dil_h < 1 || d->dil_w < 1) return false;   // "same": OH == IH, OW == IW
  if ((d->C & 31) || (d->K & 31) || (d->x_ld & 7) || (d->y_ld & 7)) return false;
  if (d->W >= 65536 || d->H >= 32768) return false;
  const int64_t M = (int64_t)d->N * d->H * d->W;
  if (M >= (1ll << 30) || M < kWbRange) return false;
  if (wgband_mode() < 2 && M < 4096) return false;   // default policy: small problems stay on the general kernel
  int KF = (d->K & 63) ? 2 : 4;
  int KT = 16 * KF;
  WgBandArgs& a = pl->a;
  memset(&a, 0, sizeof(a));
  a.NB = d->N;
  a.OH = d->H;
  a.OW = d->W;
  a.Cin = d->C;
  a.x_ld = d->x_ld;
  a.K = d->K;
  a.dy_ld = d->y_ld;
  a.Ktot = 9 * d->C;
  a.dil_h = d->dil_h;
  a.dil_w = d->dil_w;
  a.G = d->dil_h;
  a.VP = d->H + a.G;
  a.PW = (d->W + 2 * d->dil_w + 7) & ~7;
  // patch rows a range can need: the output rows 256 consecutive pixels touch, the zero rows of the image boundaries among them,
  // the reach of the taps above and below
  const int rows_max = (kWbRange - 1 + d->W - 1) / d->W + 1;
  const int cross_max = (rows_max + d->H - 1) / d->H;
  const int ph_max = rows_max + cross_max * a.G + 2 * d->dil_h;
  if ((int64_t)ph_max * a.PW >= 32768) return false;
  a.nxp = (ph_max * a.PW + 15) / 16;
  if (a.nxp > kWbXPieces * kWbWaves) return false;
  a.xbuf = a.nxp * 1024;
  a.dbuf = kWbRange * KT * 2;
  if (KF == 4 && 2 * (a.xbuf + a.dbuf) > kWbLdsMax) {
    // wide rows (128 - 200 pixels: the patch of a range is 50 - 63 KB): the 32-channel output tile's dY rows are half as long, and two
    // [dY | patch] buffers fit again — 18 instead of 36 MFMAs per step and wave, still ahead of the general kernel (YOLOv7-l's 160-pixel
    // rows, DeepLabv3+'s 128-pixel rows: profiles/r06_wgrad_band_wide.log) — on the smaller problems only: with many tiles x pixels the
    // 18-MFMA steps lose to the general kernel's 128-wide tile (128 -> 128 @160x160 batch 16: 232 vs 224 us, 256 -> 256: 834 vs 741)
    if (wgband_mode() < 2 && (double)M * (d->K / 32) * (d->C / 32) > 4.0e6) return false;
    KF = 2;
    KT = 32;
    a.dbuf = kWbRange * KT * 2;
  }
  const int fold = 4 * KF * 9 * 64 * 16;
  int lds = 2 * (a.xbuf + a.dbuf);
  if (lds < fold) lds = fold;
  if (lds > kWbLdsMax) return false;
  a.M = (int)M;
  a.c_tiles = d->C / 32;
  a.tiles = (d->K / KT) * a.c_tiles;
  const int total_ranges = (int)((M + kWbRange - 1) / kWbRange);
  int splits = 256 / a.tiles;
  if (splits < 1) splits = 1;
  if (splits > total_ranges) splits = total_ranges;
  a.ranges_per_split = (total_ranges + splits - 1) / splits;
  splits = (total_ranges + a.ranges_per_split - 1) / a.ranges_per_split;
  div31_consts(a.OW, &a.ow_mul, &a.ow_sh);
  div31_consts(a.OH, &a.oh_mul, &a.oh_sh);
  div31_consts(a.VP, &a.vp_mul, &a.vp_sh);
  div31_consts(a.PW, &a.pw_mul, &a.pw_sh);
  pl->KF = KF;
  pl->blocks = a.tiles * splits;
  pl->lds = lds;
  return true;
}

// plan query (api.hip cvhip_conv2d_wgrad_band_plan): {KF, tiles, splits, ranges_per_split, blocks, lds bytes, PW, patch KB pieces}
int wgband_plan_export(const cvhip_conv_desc* d, int32_t* out) {
  if (wgband_mode() == 0) return 0;
  WgBandPlan pl;
  if (!wgband_plan(d, &pl)) return 0;
  if (out) {
    const int32_t v[8] = {pl.KF, pl.a.tiles, pl.blocks / pl.a.tiles, pl.a.ranges_per_split, pl.blocks, pl.lds, pl.a.PW, pl.a.nxp};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
  }
  return 1;
}

template <int KF>
static int wgband_launch(const WgBandPlan& pl, hipStream_t stream) {
  auto kern = wgrad_band_kernel<KF>;
  static bool attr_done[64] = {};
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& attr_set = attr_done[devid & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kWbLdsMax);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(wgrad_band_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(kWbWaves * 64), pl.lds, stream, pl.a);
  return check_launch("wgrad_band_kernel");
}

// -1 = not taken (the caller goes on to the general kernel)
int try_launch_wgrad_band(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, hipStream_t stream) {
  if (wgband_mode() == 0) return -1;
  if ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) return -1;
  WgBandPlan pl;
  if (!wgband_plan(d, &pl)) return -1;
  pl.a.x = (const h16_t*)x;
  pl.a.dy = (const h16_t*)dy;
  pl.a.dw = dw;
  return pl.KF == 4 ? wgband_launch<4>(pl, stream) : wgband_launch<2>(pl, stream);
}

}  // namespace cvhip
