// conv_band.hip — row-band 3x3 implicit-GEMM convolution for gfx950 (CDNA4): stride-1 3x3 (any padding / dilation) fprop and dgrad
// whose input channel count is a multiple of 32 and whose output channel count is 32, 64 or a multiple of 128.
//
//   Out[m][n] = sum_{t,c} X[pix(m) + tap(t)][c] * Wt[n][t*Cin + c]              (same plan / operand images as conv_igemm.hip)
//
// Why a third implicit GEMM (round 5; VERDICT r04 task 2). The per-tap kernel (conv_igemm.hip) and the patch-resident kernel
// (conv_patch.hip) both hand the weight tile of every K step to the waves through an LDS ring: one block-wide barrier per K step (or
// per two), with every wave of the CU arriving in phase — 20 % / 30 % MFMA-pipe busy for four rounds — and both tile the output in
// fixed 256-pixel tiles that quantise badly against the 256 CUs (102 400 output pixels of the dominant 128 -> 128 @40x40 layer = 400
// tiles: 1.56 rounds). Here
//   * a block owns a BAND: TH whole output rows of one image (TH x OW pixels), all output channels of a <= 128-wide tile. Default plan:
//     NW = 4 waves per block, TWO blocks resident per CU (5 rows x 40 = 200 pixels for the dominant layer -> 512 blocks = exactly the
//     512 block slots): the halves are not coupled by barriers, so one block's prologue fetch, chunk barriers and store epilogue run
//     under the other's MFMAs; NW = 8 (one 512-thread block per CU, twice the band) where the small plan does not fit or fill the chip;
//   * the band's input patch ((TH + 2) x (OW + 2) pixels) of one 32-channel chunk is staged into the LDS once by LDS-DMA, pixel-major,
//     double-buffered across chunks; the nine taps read their fragments from it (as conv_patch.hip does);
//   * the WEIGHT fragments never touch the LDS: a wave owns a 16*NF-channel slice of the output channels and loads its NF 16 x 32
//     fragments of a K step straight from global memory into registers, LEAD K steps ahead — so inside a chunk (9 K steps, 234 MFMAs
//     per wave) NO wave waits for any other wave: one barrier per 32-channel chunk instead of one per K step;
//   * those fragments come from the layer's BAND IMAGE (conv_plan.h): a second, fragment-ordered copy of the weights behind the
//     row-major image, 1 KB of contiguous memory per fragment. From the row-major image a wave's fetch was 16 rows x 64 bytes — sixteen
//     half cache lines per instruction — and the texture addresser, not the MFMA pipe or the LDS, bounded the kernel
//     (profiles/r05_band_coalesced_weight_probe.log: 256 -> 256 @20x20 44 -> 32 us with the fetches made contiguous).
// 64 * NW threads = WN (output-channel slices of 16*NF) x WM (pixel parts of <= MFW fragments of 16 pixels) waves; v_mfma_f32_16x16x32
// with swapped operands (a lane ends up with 4 consecutive output channels of one pixel).
//
// Forms (template parameters below; launch_igemm's default is NF = 2, PF = 0; the others are measured options):
// narrow waves (32 channels x 13 or 7 pixel fragments) / wide waves (64 channels x 7 fragments: every pixel fragment read from the LDS
// feeds four MFMAs); compiler-scheduled LDS reads / a hand-counted read-ahead of the next K step's fragments.
//
// LDS image: pixel rows of 64 bytes (one 32-channel chunk), row pitch PW pixels (a multiple of 8, so a fragment whose 16 output
// pixels wrap to the next image row keeps the bank pattern of 16 consecutive pixels), 16-byte slot s of patch pixel pp stored at slot
// s ^ (((pp >> 2) & 1) << 1) — the swizzle of conv_patch.hip (conflict-free ds_read_b128 at any pixel offset). Buffers are whole KBs
// (16 pixels: one DMA instruction of one wave); DMA instructions with nothing to fetch write one shared dummy KB.
//
// VMEM queue discipline. A wave's weight loads are inline-asm global_load_dwordx4 (hipcc drains vmcnt to 0 at the first use of an
// ordinary load's result while an LDS-DMA is in flight: cdna_hip_programming.md "Pipelining across barriers"); the waits are counted
// by hand: at K step k the wave issues B(k + LEAD) and its share of the next chunk's patch DMAs (<= PPS instructions), then waits until
// only the instructions younger than B(k) are outstanding. A patch piece issued at step t is therefore complete at step t + 2 at the
// latest; the pieces of a chunk are issued in its first six steps, so the chunk-end barrier publishes a landed patch. (Spill code would
// only ADD instructions to the queue: a counted wait then waits for more than it needs, never for less.)
//
// Epilogue: raw 16-bit stores (8 bytes per lane: 4 channels of a pixel), optional addend (dgrad: the gradient arriving over a skip
// connection), optional training-mode BatchNorm sums into the layer's fp64 accumulator (common.h acc_add2).
//
// Replaces aten::convolution / convolution_backward(input) reached from reference src/models/bricks/conv_module.py:209 and
// trainer.py:189 for the 3x3 stride-1 layers (DarknetBottleneck conv2: modules/yolo_modules.py:95-104; torchvision Bottleneck conv2).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

constexpr int kBandSteps = 9;        // 3 x 3 taps per chunk
constexpr int kBandPieceSteps = 6;   // the next chunk's patch DMAs go out in the first six K steps of a chunk
constexpr int kBandLdsMax = 156 * 1024;

struct BandArgs {
  const h16_t* x;
  const h16_t* w;
  h16_t* y;
  double* stats;  // fp64 accumulator [kAccShards][2][stats_ld] or NULL
  int stats_ld;
  const h16_t* res;
  int res_ld;
  // fused epilogue (EPI instances, round 6): out = act(acc * ep_scale + ep_shift) with `res` joining before (res_pre) or after the activation
  const float *ep_scale, *ep_shift, *bias;   // (the bias folds into the shift: (acc + b) * s + t = acc * s + (b * s + t))
  int ep_act, res_pre;
  float ep_ap;
  int NB, IH, IW, Cin, x_ld;
  int Nout, y_ld, OH, OW;
  int dh0, dh_step, dw0, dw_step, lo_h, lo_w;
  int TH, bands, PH, PW;
  // stride-2 fprop (round 6): the patch holds rs = 2 input rows per output row and every patch row is stored as TWO column planes —
  // the odd input columns 2k - 1 (k = 0 .. OW: PWo pixels) and then the even ones 2k — so that the 16 consecutive output pixels of a
  // fragment read 16 consecutive LDS pixels for every tap (tap column 0 / 1 / 2 = odd plane k, even plane k, odd plane k + 1)
  int rs, s2, PWo;
  int coffp[3];        // tap column j -> pixel offset inside a patch row
  int nplw;            // patch DMA instructions per wave per chunk
  int dummy_off;       // byte offset of the 1-KB dummy DMA slot
  int buf_bytes;       // one patch buffer
  int n_tiles, total_tiles, Ktot;
  unsigned ow_magic;   // ceil(2^32 / OW)
  unsigned pw_magic;   // ceil(2^32 / PW)
};

#define CVHIP_BGLDS16(src, dst)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __attribute__((aligned(64))) unsigned int g_band_zero[16];

typedef unsigned int band_u32x4 __attribute__((ext_vector_type(4)));

// weight fragment: 16 bytes per lane, global -> VGPRs, outside the compiler's waitcnt bookkeeping (see the header)
// (SGPR base + 32-bit lane offset: the wave-uniform part of the address — channel slice, tap, chunk — is scalar arithmetic, the lane's
// part one VGPR for all fragments)
__device__ __forceinline__ void band_gload16(band_u32x4& dst, const h16_t* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// wait until at most N VMEM instructions of this wave are outstanding; the two fragments are in/out operands so that no use of them
// can be scheduled above the wait. N is a LITERAL and the asm sits in straight-line code: behind a run-time switch the register
// allocator merged the cases with copies of the fragments placed BEFORE the s_waitcnt — copies of registers whose loads had not landed
template <int N>
__device__ __forceinline__ void band_wait(band_u32x4& a, band_u32x4& b) {
  static_assert(N >= 0 && N <= 15, "vmcnt literal");
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void band_wait(band_u32x4& a, band_u32x4& b, band_u32x4& c, band_u32x4& d) {
  static_assert(N >= 0 && N <= 15, "vmcnt literal");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// pixel fragment: 16 bytes per lane, LDS -> VGPRs by hand (PF > 0 forms). hipcc waits for ALL outstanding LDS reads before the first
// MFMA of a K step (s_waitcnt lgkmcnt(0) throughout the compiled loop: round-5 ISA listing), so a read-ahead it schedules itself is
// waited for in the very step that issued it. Here the reads are asm (outside its bookkeeping) and the waits are counted: LDS reads
// return in order, so "at most N outstanding" = everything but the N youngest has landed.
__device__ __forceinline__ void band_lds16(band_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void band_lwait(band_u32x4& a) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt literal");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

__device__ __forceinline__ int band_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// WN : waves along the output channels (16 * NF channels each); WM = 8 / WN pixel parts
// MFW: 16-pixel fragments per wave (7, 10 or 13)    PPS: patch DMA instructions per wave per K step (1 or 2)
// NF : 16-channel weight fragments per wave. 2 = the first form: 4 / 2 / 1 waves across a 128 / 64 / 32-wide tile, 13 (or 7) pixel
//      fragments per wave, weights two K steps ahead. 4 = the WIDE-WAVE form (64 channels x 7 pixel fragments per wave, 2 / 1 waves
//      across a 128 / 64-wide tile): every pixel fragment read from the LDS feeds FOUR MFMAs instead of two — half the ds_read_b128
//      instructions, address adds and LDS waits per MFMA — at twice the weight bytes per wave and K step (four fragments, L2 / L1).
// LEAD: K steps the weight fragments are fetched ahead (2; 1 keeps NF = 4 inside 256 registers: two sets of four fragments live)
// PF : pixel fragments of the NEXT K step read from the LDS before this step's MFMAs (0 = none; MFW = all of them: the step's MFMAs
//      never wait for the LDS inside a chunk). Only inside a chunk: the next chunk's buffer is published by the chunk-end barrier.
// NW : waves per block. 8 = one block per CU (256 VGPRs per wave at two waves per SIMD). 4 = half-height bands in 256-thread blocks, two
//      of them resident per CU (the same registers per wave, <= 78 KB of LDS each): the same per-CU work, but the two blocks are not
//      coupled by barriers — one's prologue, chunk barriers and store epilogue can run under the other's MFMAs.
// EPI: fused-epilogue instance (inference: folded BatchNorm scale / shift + activation + residual in the convolution's own pass:
//      conv_module.py:201-214 in eval mode, utils/fuse.py:32-54); the training instances carry none of its code
__device__ __forceinline__ void band_act4(float (&v)[4], int act, float ap) {
  switch (act) {
    case CVHIP_ACT_NONE: break;
    case CVHIP_ACT_RELU:
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_RELU, ap);
      break;
    case CVHIP_ACT_SILU:
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_SILU, ap);
      break;
    case CVHIP_ACT_LEAKY:
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_LEAKY, ap);
      break;
    case CVHIP_ACT_SIGMOID:
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_SIGMOID, ap);
      break;
    default:
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], CVHIP_ACT_HSWISH, ap);
      break;
  }
}

template <int WN, int MFW, int PPS, int NF = 2, int LEAD = 2, int PF = 0, int NW = 8, bool EPI = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_band_kernel(const BandArgs p) {
  constexpr int kBandWaves = NW;
  static_assert(NW == 4 || NW == 8, "waves per block");
  static_assert(WN <= NW, "channel slices per block");
  static_assert(NF == 2 || NF == 4, "weight fragments per wave");
  static_assert(LEAD == 1 || LEAD == 2, "weight prefetch distance");
  static_assert(PF >= 0 && PF <= MFW, "LDS prefetch: fragments of the next step");
  constexpr int WM = kBandWaves / WN;
  constexpr int BN = WN * 16 * NF;
  constexpr int NPL = PPS * kBandPieceSteps;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int g = lane >> 4;

  // ---- which band ---------------------------------------------------------------------------------------------------
  const int lt = xcd_remap(blockIdx.x, p.total_tiles);
  const int sp = lt / p.n_tiles;
  const int ntile = lt - sp * p.n_tiles;
  const int n_img = sp / p.bands;
  const int band = sp - n_img * p.bands;
  const int oh0 = band * p.TH;
  const int rows = min(p.TH, p.OH - oh0);
  const int npx = rows * p.OW;
  const int nfrag = (npx + 15) >> 4;
  const int n0 = ntile * BN;
  const int f0 = wm * MFW;
  const int nfr = min(max(nfrag - f0, 0), MFW);
  const int Cin = p.Cin;
  const int NC = Cin >> 5;
  const int PW = p.PW;
  unsigned char* const sbuf = smem;

  // ---- patch loader: DMA instruction j of wave w fills the 16 patch pixels (j*8 + w)*16 .. +15, lane l the PHYSICAL slot l % 4 of
  // pixel l / 4 and fetches the LOGICAL slot that belongs there (swizzle on the source side) -----------------------------------
  const int lsl = (lane & 3) ^ (((lane >> 4) & 1) << 1);   // (pp >> 2) & 1 == bit 4 of the lane: the same for every j
  const h16_t* const zsrc = reinterpret_cast<const h16_t*>(g_band_zero) + lsl * 8;
  int poff[NPL];
  const int npix = p.PH * PW;
  {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int pp = ((j * kBandWaves + wave) << 4) + (lane >> 2);
      const int pr = band_div(pp, p.pw_magic);
      const int pc = pp - pr * PW;
      const int ih = oh0 * p.rs + p.lo_h + pr;
      int iw = p.lo_w + pc;
      bool col_ok = true;
      if (p.s2) {  // (block-uniform) column planes: odd input columns first, then the even ones
        const bool odd = pc < p.PWo;
        const int k = odd ? pc : pc - p.PWo;
        iw = 2 * k + (odd ? p.lo_w : p.lo_w + 1);
        col_ok = odd ? k <= p.OW : k < p.OW;
      }
      const bool ok = pp < npix && col_ok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      poff[j] = ok ? ((n_img * p.IH + ih) * p.IW + iw) * (p.x_ld >> 3) : -1;   // in 16-byte units (the planner bounds the tensor)
    }
  }
  // `live` false (no next chunk, or j past the patch): the same instruction fetches the zero page into the block's 1-KB dummy slot, so
  // that every K step carries a compile-time number of VMEM instructions
  unsigned char* const sdummy = sbuf + p.dummy_off;   // ONE dummy KB for all waves: it is written (zeros, any order) and never read
  const h16_t* const xlane = p.x + lsl * 8;
  auto issue_piece = [&](int j, int c, unsigned char* buf, bool live) __attribute__((always_inline)) {
    live = live && j < p.nplw && (((j * kBandWaves + wave) << 4) < npix);   // (a KB wholly past the patch has no room in the buffer)
    // (the offset is made opaque per use: otherwise the compiler hoists the twelve `poff[j] >= 0` lane masks out of the K loop into
    // 24 SGPRs, and the kernel spills scalars into the vector file)
    int po = poff[j];
    asm volatile("" : "+v"(po));
    const h16_t* const src = (live && po >= 0) ? xlane + (((int64_t)po << 3) + c * 32) : zsrc;
    CVHIP_BGLDS16(src, live ? buf + ((j * kBandWaves + wave) << 10) : sdummy);
  };

  // ---- fragment geometry --------------------------------------------------------------------------------------------
  // ab[b]: BYTE address of (this lane's output pixel, logical slot g) in a patch buffer before the tap shift and the swizzle. The
  // column shift of a tap moves the pixel index, so the swizzle (bit 5 ^= bit 8 of the byte address) is applied per read; the row
  // shift is a multiple of PW pixels = of 512 bytes and the buffer base a multiple of 8 KB: neither touches bit 8 relative to bit 5
  int ab[MFW];
#pragma unroll
  for (int b = 0; b < MFW; ++b) {
    int q = ((f0 + b) << 4) + (lane & 15);
    q = q < npx ? q : (npx > 0 ? npx - 1 : 0);
    const int r = band_div(q, p.ow_magic);
    const int c = q - r * p.OW;
    ab[b] = ((r * p.rs * PW + c) << 6) + (g << 4);
  }
  // K steps visit the taps COLUMN-major (step ts: tap column j = ts / 3, tap row i = ts % 3): the swizzled address of a column is
  // computed once (4 VALU per fragment) and serves three steps with one add each — the inner loop is VALU-issue bound otherwise
  // (profiles/r05_band_sq.txt: 3.1 VALU per MFMA with the per-read swizzle)
  int coff[3], roff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) coff[j] = __builtin_amdgcn_readfirstlane(p.coffp[j] * 64);
#pragma unroll
  for (int i = 0; i < 3; ++i) roff[i] = __builtin_amdgcn_readfirstlane((p.dh0 + i * p.dh_step - p.lo_h) * PW * 64);
  int aj[MFW];

  // weight fragments from the layer's BAND IMAGE (conv_plan.h): the fragment of (K step = tap * NC + chunk, 16-channel group f) is the
  // 1 KB at ((step * (Nout / 16) + f) * 64 + lane) * 16 bytes — lane (row n = lane & 15, K group g = lane >> 4) holds Wt[n][k0 + g*8 .. +7]
  // as the MFMA wants it, and the wave's fetch is one contiguous KB instead of sixteen 64-byte row pieces
  const int nfr16 = p.Nout >> 4;
  const h16_t* const wbase = p.w + (int64_t)((n0 >> 4) + wn * NF) * 512;   // wave-uniform
  const unsigned wlane = (unsigned)(lane * 16);

  f32x4 acc[NF][MFW];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MFW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  band_u32x4 wb[3][NF];  // sets named by K step % 3; LEAD + 1 of them are live at a time
  auto load_b = [&](int set, int tt, int cc) __attribute__((always_inline)) {
    cc = cc < NC ? cc : NC - 1;  // (past the end: a valid, unused fetch — the VMEM counts stay uniform)
    const int off = (tt * NC + cc) * nfr16 * 512;
#pragma unroll
    for (int a = 0; a < NF; ++a) band_gload16(wb[set][a], wbase + (off + a * 512), wlane);
  };

  // ---- prologue: the first chunk's patch, the first LEAD K steps' weights --------------------------------------------------
#pragma unroll
  for (int j = 0; j < NPL; ++j)
    if (j < p.nplw) issue_piece(j, 0, sbuf, true);
  load_b(0, 0, 0);                          // step 0: tap (0, 0)
  if constexpr (LEAD == 2) load_b(1, 3, 0);  // step 1: tap (1, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  band_u32x4 xq[PF > 0 ? PF : 1];  // PF > 0: the next K step's first pixel fragments, read one step ahead
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);

  // ---- main loop: chunk-major, nine K steps (taps) inside ---------------------------------------------------------------
  for (int c = 0; c < NC; ++c) {
    const bool more = c + 1 < NC;
    unsigned char* const sNext = sbuf + ((c + 1) & 1) * p.buf_bytes;
    const int cbuf = (c & 1) * p.buf_bytes;
    if constexpr (PF > 0) {
      // (the three tap columns' swizzled addresses of every fragment are loop-invariant; hoisted out of the chunk loop they cost
      // 3 x MFW registers the read-ahead forms do not have — made opaque once per chunk, they are 3 VALU instructions per fragment and column)
#pragma unroll
      for (int b = 0; b < MFW; ++b) asm volatile("" : "+v"(ab[b]));
    }
    auto step = [&](auto tsc) __attribute__((always_inline)) {
      constexpr int ts = decltype(tsc)::value;
      // (1) weights of K step k + LEAD
      constexpr int tsl = (ts + LEAD) % kBandSteps;
      load_b((ts + LEAD) % 3, (tsl % 3) * 3 + tsl / 3, ts + LEAD < kBandSteps ? c : c + 1);
      // (2) this step's share of the next chunk's patch (always PPS instructions in the first six steps)
      if (ts < kBandPieceSteps) {
#pragma unroll
        for (int u = 0; u < PPS; ++u) issue_piece(ts * PPS + u, c + 1, sNext, more);
      }
      // (3) K step k's weights have landed once at most the instructions issued after them are outstanding. LEAD = 2: B(k + 1), the
      // pieces of step k - 1, B(k + 2), the pieces of step k (the pieces of step k - 2 are younger than B(k) too: waiting for them as
      // well keeps the count simple). LEAD = 1: the pieces of step k - 1, B(k + 1), the pieces of step k. Either way a piece issued
      // at step t is complete at step t + 2 at the latest (it is older than B(t + 2) / B(t + 3)).
      constexpr int PPREV = (ts >= 1 && ts - 1 < kBandPieceSteps) ? PPS : 0;
      constexpr int PCUR = ts < kBandPieceSteps ? PPS : 0;
      if constexpr (NF == 2)
        band_wait<LEAD * NF + PPREV + PCUR>(wb[ts % 3][0], wb[ts % 3][1]);
      else
        band_wait<LEAD * NF + PPREV + PCUR>(wb[ts % 3][0], wb[ts % 3][1], wb[ts % 3][2], wb[ts % 3][3]);
      // (4) the tap's fragments from the patch, NF MFMAs per fragment
      if constexpr (ts % 3 == 0 && (PF == 0 || ts == 0)) {
#pragma unroll
        for (int b = 0; b < MFW; ++b) {
          const int u = ab[b] + coff[ts / 3];
          aj[b] = u ^ ((u >> 3) & 32);
        }
      }
      const int tsh = roff[ts % 3] + cbuf;
      h16x8 wf[NF];
#pragma unroll
      for (int a = 0; a < NF; ++a) wf[a] = __builtin_bit_cast(h16x8, wb[ts % 3][a]);
      if constexpr (PF == 0) {
        // (no per-fragment guards: a branch per fragment would fence the scheduler; fragments past the band read a clamped, valid
        // pixel and are never stored or summed — the planner sizes the bands so that few of them exist). Groups of 7 / 5 fragments
        // bound the registers the fragment reads hold.
        constexpr int GF = MFW <= 7 ? 7 : 5;
#pragma unroll
        for (int b0 = 0; b0 < MFW; b0 += GF) {
          h16x8 xa[GF];
#pragma unroll
          for (int b = b0; b < b0 + GF && b < MFW; ++b) {
            xa[b - b0] = *reinterpret_cast<const h16x8*>(sbuf + (aj[b] + tsh));
          }
#pragma unroll
          for (int b = b0; b < b0 + GF && b < MFW; ++b) {
#pragma unroll
            for (int a = 0; a < NF; ++a) acc[a][b] = CVHIP_MFMA_16X16X32(wf[a], xa[b - b0], acc[a][b], 0, 0, 0);
          }
        }
      } else {
        // LDS one step ahead: this step's first PF fragments were read during the previous step (step 0 of a chunk reads its own: the
        // buffer was published by the barrier just passed), the others are read now and land under the MFMAs of the first PF; the
        // next step's first PF reads go out BEFORE this step's MFMAs too. Counted waits (band_lds16): of the NEW reads issued in
        // this step, fragment B0 + i may be used once at most NEW - 1 - i reads are outstanding, the read-ahead ones once at most NEW.
        constexpr int B0 = ts == 0 ? 0 : PF;
        constexpr int NNEXT = ts + 1 < kBandSteps ? PF : 0;
        constexpr int NEW = (MFW - B0) + NNEXT;
        band_u32x4 xa[MFW];
#pragma unroll
        for (int b = 0; b < B0; ++b) xa[b] = xq[b];
#pragma unroll
        for (int b = B0; b < MFW; ++b) band_lds16(xa[b], lds0 + (unsigned)(aj[b] + tsh));
        if constexpr (NNEXT > 0) {
          if constexpr ((ts + 1) % 3 == 0) {
#pragma unroll
            for (int b = 0; b < MFW; ++b) {
              const int u = ab[b] + coff[(ts + 1) / 3];
              aj[b] = u ^ ((u >> 3) & 32);
            }
          }
          const int tsn = roff[(ts + 1) % 3] + cbuf;
#pragma unroll
          for (int b = 0; b < PF; ++b) band_lds16(xq[b], lds0 + (unsigned)(aj[b] + tsn));
        }
        auto mfmas = [&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          if constexpr (b < MFW) {
            // (MFMAs and asm statements do not cross: without the fence hipcc gathers the counted waits into one run ahead of all the
            // step's MFMAs — lgkmcnt(10), (9), ... (4) back to back — which is the blanket wait again)
            if constexpr (b == 0 || b >= B0) __builtin_amdgcn_sched_barrier(0);
            if constexpr (b < B0) {
              if constexpr (b == 0) {
                band_lwait<NEW>(xa[0]);            // everything issued before this step has landed: all B0 read-ahead fragments
                __builtin_amdgcn_sched_barrier(0);  // (only xa[0] is an operand of the wait: the others' MFMAs stay below it too)
              }
            } else {
              band_lwait<NEW - 1 - (b - B0)>(xa[b]);
            }
            const h16x8 xv = __builtin_bit_cast(h16x8, xa[b]);
#pragma unroll
            for (int a = 0; a < NF; ++a) acc[a][b] = CVHIP_MFMA_16X16X32(wf[a], xv, acc[a][b], 0, 0, 0);
            if constexpr (b == MFW - 1) __builtin_amdgcn_sched_barrier(0);
          }
        };
        mfmas(std::integral_constant<int, 0>{});
        mfmas(std::integral_constant<int, 1>{});
        mfmas(std::integral_constant<int, 2>{});
        mfmas(std::integral_constant<int, 3>{});
        mfmas(std::integral_constant<int, 4>{});
        mfmas(std::integral_constant<int, 5>{});
        mfmas(std::integral_constant<int, 6>{});
        static_assert(MFW <= 7, "the read-ahead forms hold at most 7 pixel fragments per wave");
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    // every piece of chunk c + 1 this wave issued has landed (header); the barrier publishes them and frees buffer c & 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // the weight fetches issued past the last K step are still in flight: their destination registers stay LIVE (in/out operands) until
  // they have landed — the compiler sees an asm load as instantaneous and would hand a dead destination to the epilogue's addresses
  if constexpr (NF == 2) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(wb[0][0]), "+v"(wb[0][1]), "+v"(wb[1][0]), "+v"(wb[1][1]), "+v"(wb[2][0]), "+v"(wb[2][1])
                 :
                 : "memory");
  } else {
    // (LEAD = 1: only set 0 — step 0 of the chunk past the end — is in flight; the other sets' last fetches were waited for and used)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wb[0][0]), "+v"(wb[0][1]), "+v"(wb[0][2]), "+v"(wb[0][3]) : : "memory");
    if constexpr (LEAD == 2)
      asm volatile("" : "+v"(wb[1][0]), "+v"(wb[1][1]), "+v"(wb[1][2]), "+v"(wb[1][3]) : : "memory");
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  const int chb = n0 + wn * (16 * NF) + g * 4;  // + a*16: first of the lane's 4 consecutive output channels
  const bool r8 = p.res != nullptr;
  float esc[EPI ? NF : 1][4], esh[EPI ? NF : 1][4];
  if constexpr (EPI) {
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        esc[a][q] = p.ep_scale ? p.ep_scale[chb + a * 16 + q] : 1.f;
        esh[a][q] = p.ep_shift ? p.ep_shift[chb + a * 16 + q] : 0.f;
        if (p.bias) esh[a][q] += p.bias[chb + a * 16 + q] * esc[a][q];
      }
  }
#pragma unroll
  for (int b = 0; b < MFW; ++b) {
    const int q = ((f0 + b) << 4) + (lane & 15);
    if (b < nfr && q < npx) {
      const int r = band_div(q, p.ow_magic);
      const int cc = q - r * p.OW;
      const int64_t opix = ((int64_t)n_img * p.OH + (oh0 + r)) * p.OW + cc;
      h16_t* const yrow = p.y + opix * p.y_ld + chb;
#pragma unroll
      for (int a = 0; a < NF; ++a) {
        float v0 = acc[a][b][0], v1 = acc[a][b][1], v2 = acc[a][b][2], v3 = acc[a][b][3];
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
        if (r8) {
          const uint2 u = *reinterpret_cast<const uint2*>(p.res + opix * p.res_ld + chb + a * 16);
          unpack2(u.x, r0, r1);
          unpack2(u.y, r2, r3);
        }
        if constexpr (EPI) {
          float v[4] = {v0 * esc[a][0] + esh[a][0], v1 * esc[a][1] + esh[a][1], v2 * esc[a][2] + esh[a][2], v3 * esc[a][3] + esh[a][3]};
          if (p.res_pre) {   // residual before the activation (ResNet bottleneck tail)
            v[0] += r0;
            v[1] += r1;
            v[2] += r2;
            v[3] += r3;
          }
          band_act4(v, p.ep_act, p.ep_ap);
          if (!p.res_pre) {  // Darknet shortcut: x + act(bn(conv))
            v[0] += r0;
            v[1] += r1;
            v[2] += r2;
            v[3] += r3;
          }
          v0 = v[0];
          v1 = v[1];
          v2 = v[2];
          v3 = v[3];
        } else {
          v0 += r0;
          v1 += r1;
          v2 += r2;
          v3 += r3;
        }
        uint2 o;
        o.x = pack2(v0, v1);
        o.y = pack2(v2, v3);
        *reinterpret_cast<uint2*>(yrow + a * 16) = o;
      }
    }
  }
  if (p.stats) {  // training-mode BatchNorm sums of the fp32 accumulators (valid output positions only)
    float* const red = reinterpret_cast<float*>(smem);  // [WM][BN][2] (the patch buffers are dead: every wave passed the last barrier)
#pragma unroll
    for (int a = 0; a < NF; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < MFW; ++b) {
          const int q = ((f0 + b) << 4) + (lane & 15);
          const float v = (b < nfr && q < npx) ? acc[a][b][r] : 0.f;
          s1 += v;
          s2 += v * v;
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if ((lane & 15) == 0) {
          const int nl = wn * (16 * NF) + a * 16 + g * 4 + r;
          red[(wm * BN + nl) * 2 + 0] = s1;
          red[(wm * BN + nl) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if (t < BN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s1 += red[(w * BN + t) * 2 + 0];
        s2 += red[(w * BN + t) * 2 + 1];
      }
      acc_add2(p.stats, sp, p.stats_ld, n0 + t, s1, s2);
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------

static int band_mode() {  // CVHIP_BAND: 0 = never, 1 = default policy, 2 = wherever the geometry allows (read per launch: in-process A/B)
  const char* e = getenv("CVHIP_BAND");
  return e ? atoi(e) : 1;
}

struct BandPlan {
  BandArgs a;
  int WN, MFW, PPS, lds, NF, PF, NW;
};

static inline int band_imin(int a, int b) { return a < b ? a : b; }
static inline int band_imax(int a, int b) { return a > b ? a : b; }

static int band_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// pixel-fragment slots per wave of the compiled instances: 7, 10 or 13 (10: rows of 160 / 320 pixels split exactly — YOLOv7-l @1280)
static inline int band_mfw(int per_wave) { return per_wave <= 7 ? 7 : per_wave <= 10 ? 10 : 13; }

// One candidate form (NF weight fragments per wave, at most `cap` pixel fragments per wave): the best band height and its cost in
// rounds of the 256 CUs x fragment units of the busiest wave. false = the geometry does not fit this form.
static bool band_fit(const IgemmParams& p, int NF, int cap, int NW, int EH, int PW, int* th_out, int* nplw_out, int64_t* rounds_out, int RS = 1) {
  const int BN = band_imin(p.Nout, 128);
  if (BN % (16 * NF)) return false;
  const int WN = BN / (16 * NF);
  if (WN > NW) return false;
  const int WM = NW / WN;
  const int threads = NW * 64;
  const int slots = NW == 4 ? 512 : 256;              // resident blocks on the chip
  const int lds_max = NW == 4 ? 79 * 1024 : kBandLdsMax;   // two co-resident blocks share the CU's 160 KB (1 KB each left to the runtime)
  const int NC = p.Cin / 32;
  const int n_tiles = p.Nout / BN;
  int best_th = 0, best_nplw = 0;
  int64_t best_cost = -1;
  constexpr int eth = 0;
  for (int TH = 1; TH <= p.OH; ++TH) {
    if (eth > 0 && TH != eth) continue;
    const int frags = (TH * p.OW + 15) / 16;
    const int per_wave = (frags + WM - 1) / WM;
    if (per_wave > cap) break;
    const int PH = (TH - 1) * RS + EH;
    if ((int64_t)PH * PW >= 65536) break;
    const int nplw = (PH * PW * 4 + threads - 1) / threads;
    if (nplw > 2 * kBandPieceSteps) break;
    const int lds = (NC > 1 ? 2 : 1) * ((PH * PW + 15) / 16) * 1024 + 1024;   // patch buffers in whole KBs (16 pixels) + the dummy KB
    if (lds > lds_max) break;
    const int64_t tiles = (int64_t)p.NB * ((p.OH + TH - 1) / TH) * n_tiles;
    // rounds of the resident block slots x (MFMA work of the busiest wave + prologue / epilogue, in units of NF = 2 fragments)
    const int64_t cost = ((tiles + slots - 1) / slots) * (band_mfw(per_wave) * (NF / 2) * NC + 2 + NC / 2);
    if (best_cost < 0 || cost <= best_cost) {
      best_cost = cost;
      best_th = TH;
      best_nplw = nplw;
    }
  }
  if (!best_th) return false;
  if (eth <= 0) {
    // same number of bands, evenly high: 40 rows in bands of 11 are 11 + 11 + 11 + 7 — four bands of 10 cost the same MFMA slots and stage
    // less patch (a lower band always fits where a higher one did)
    best_th = (p.OH + ((p.OH + best_th - 1) / best_th) - 1) / ((p.OH + best_th - 1) / best_th);
    best_nplw = (((best_th - 1) * RS + EH) * PW * 4 + threads - 1) / threads;
  }
  *th_out = best_th;
  *nplw_out = best_nplw;
  *rounds_out = ((int64_t)p.NB * ((p.OH + best_th - 1) / best_th) * n_tiles + slots - 1) / slots;
  return true;
}

// One candidate: blocks of NW waves. *policy_ok = the default policy would launch it (below).
static bool band_plan_nw(const IgemmParams& p, int NW, BandPlan* pl, bool* policy_ok) {
  const IgemmClass& c = p.cls[0];
  const int BN = band_imin(p.Nout, 128);
  const int h_a = c.dh0, h_b = c.dh0 + 2 * c.dh_step, w_a = c.dw0, w_b = c.dw0 + 2 * c.dw_step;
  const int lo_h = band_imin(h_a, h_b), hi_h = band_imax(h_a, h_b), lo_w = band_imin(w_a, w_b), hi_w = band_imax(w_a, w_b);
  const int EH = hi_h - lo_h + 1, EW = hi_w - lo_w + 1;
  const bool s2 = p.in_sh == 2;   // (band_plan admits stride 2 for 3x3 / pad 1 / dilation 1 forward plans only)
  const int RS = s2 ? 2 : 1;
  const int PWo = s2 ? ((p.OW + 1 + 7) & ~7) : 0;                       // odd input columns 2k - 1, k = 0 .. OW
  const int PW = s2 ? PWo + ((p.OW + 7) & ~7) : ((p.OW - 1 + EW + 7) & ~7);   // + even columns 2k, k = 0 .. OW - 1
  const int NC = p.Cin / 32;
  const int n_tiles = p.Nout / BN;
  // Narrow waves (32 channels x <= 13 pixel fragments). The wide-wave form (64 channels x 7 fragments) and the hand-counted LDS read-ahead
  // were measured in round 5 (profiles/r05_band_forms_bench.log: never ahead once the band image exists) and are no longer instantiated;
  // the kernel template keeps their code paths (NF = 4, PF > 0) for the record.
  int th2 = 0, np2 = 0, th4 = 0, np4 = 0;
  int64_t rounds2 = 0;
  const bool fit4 = false;
  const bool fit2 = band_fit(p, 2, 13, NW, EH, PW, &th2, &np2, &rounds2, RS);
  if (!fit2 && !fit4) return false;
  const bool wide = fit4;
  const int NF = wide ? 4 : 2;
  const int TH = wide ? th4 : th2;
  const int best_nplw = wide ? np4 : np2;
  const int WN = BN / (16 * NF), WM = NW / WN;
  const int frags = (TH * p.OW + 15) / 16;
  const int per_wave = (frags + WM - 1) / WM;
  memset(&pl->a, 0, sizeof(pl->a));
  BandArgs& a = pl->a;
  a.NB = p.NB;
  a.IH = p.IH;
  a.IW = p.IW;
  a.Cin = p.Cin;
  a.x_ld = p.x_ld;
  a.Nout = p.Nout;
  a.y_ld = p.y_ld;
  a.OH = p.OH;
  a.OW = p.OW;
  a.dh0 = c.dh0;
  a.dh_step = c.dh_step;
  a.dw0 = c.dw0;
  a.dw_step = c.dw_step;
  a.lo_h = lo_h;
  a.lo_w = lo_w;
  a.TH = TH;
  a.bands = (p.OH + TH - 1) / TH;
  a.PH = (TH - 1) * RS + EH;
  a.PW = PW;
  a.rs = RS;
  a.s2 = s2 ? 1 : 0;
  a.PWo = PWo;
  for (int j = 0; j < 3; ++j) a.coffp[j] = s2 ? (j == 0 ? 0 : j == 1 ? PWo : 1) : (c.dw0 + j * c.dw_step - lo_w);
  a.nplw = best_nplw;
  a.buf_bytes = ((a.PH * a.PW + 15) / 16) * 1024;
  a.n_tiles = n_tiles;
  const int64_t total = (int64_t)p.NB * a.bands * n_tiles;
  if (total >= (1ll << 30)) return false;
  a.total_tiles = (int)total;
  a.Ktot = 9 * p.Cin;
  a.ow_magic = div_magic(p.OW);
  a.pw_magic = div_magic(PW);
  pl->WN = WN;
  pl->NF = NF;
  pl->MFW = NF == 4 ? 7 : band_mfw(per_wave);
  pl->PPS = best_nplw <= kBandPieceSteps ? 1 : 2;
  pl->PF = 0;
  a.dummy_off = (NC > 1 ? 2 : 1) * a.buf_bytes;
  pl->NW = NW;
  pl->lds = band_imax(a.dummy_off + 1024, WM * BN * 2 * (int)sizeof(float));
  // Default policy = the classes of problems the kernel measured FASTER on than the patch-resident / per-tap kernels, isolated launches on
  // rotating operands AND in the replayed train step (profiles/r05_band_image_bench.log, r05_band_policy_step_ab.log): at most two
  // 128-wide channel tiles, block counts that fill whole rounds of the resident block slots, waves that are nearly full (a
  // single-chunk layer — 32 input channels — is bound by its patch and output traffic, not by MFMA slots: exempt). YOLOv5-s batch 64:
  // 128 -> 128 @40x40 34.5 vs 44.1 us, 256 -> 256 @20x20 32.3 vs 42.1, 32 -> 32 @160x160 71.3 vs 82.1, 64 -> 64 @80x80 55.4 vs 58.6.
  // It LOSES where the bands leave slots or fragment slots idle (DeepLabv3+ batch 16: 128 -> 128 @64x128 59 vs 45.4 us, 64 -> 64
  // @128x256 92 vs 70) and is level on four-tile problems (512 -> 512 dilated: 149 vs 149).
  const int slots = NW == 4 ? 512 : 256;
  const int64_t rounds = (total + slots - 1) / slots;
  if (s2) {
    // stride-2 forward plans (round 6, second session), measured against the per-tap kernel per layer (profiles/r06_band_s2_bench.log):
    // ahead on NARROW maps — 256 -> 256 @40 -> 20 37.4 vs 44.7 us, 256 -> 512 @40 -> 20 68.4 vs 70.8, ResNet layer4 512 -> 512 @32x64 -> 16x32
    // 53.8 vs 56.7 — and behind on wide ones (64 -> 128 @160 -> 80 141 vs 112, 128 -> 256 @80 -> 40 100 vs 88, 256 -> 512 @64x128 122 vs 81): a
    // stride-2 band stages 2.2 - 3 input rows per output row for every channel tile, and at one- or two-row bands the LDS-DMA, not the
    // MFMA pipe, sets the pace (~40 B/clk/CU). Taken up to 32 output columns; CVHIP_BAND_S2=0 / CVHIP_BAND=2 for the A/B.
    // (training forms only: with the fused inference epilogue the two kernels are level — infer 17.75 vs 17.87 k img/s)
    const bool epi = p.bias || p.ep_scale || p.ep_act != CVHIP_ACT_NONE;
    *policy_ok = band_env("CVHIP_BAND_S2", 1) != 0 && !epi && p.OW <= 32 && (int64_t)TH * p.OW * 10 >= (int64_t)pl->MFW * WM * 16 * 6;
    return true;
  }
  *policy_ok = n_tiles <= 2 && total * 10 >= rounds * slots * 9 &&                                 // >= 90 % of the block slots of every round busy
               (NC == 1 || (int64_t)TH * p.OW * 20 >= (int64_t)pl->MFW * WM * 16 * 17);          // >= 85 % of the fragment slots busy
  // Round 6: DEEP reductions (>= 8 chunks of 32 input channels: ResNet layer3 / layer4 conv2, the ASPP / reduce convolutions) are where one
  // barrier per nine K steps and weights straight into registers pay most and where the patch-resident / per-tap kernels are weakest, so
  // the kernel wins with emptier rounds and waves too (DeepLabv3+ batch 16, profiles/r06_band_deep_policy.log: 256 -> 256 @32x64 45.9 vs
  // 54.9 us, 512 -> 512 @16x32 53.9 vs 89.1 (per-tap 63.1), 2560 -> 512 @16x32 225.6 vs 372.7 (278.6); dgrad alike); the shallow
  // 64- / 128-channel layers of the same network still lose (80.9 vs 69.8, 54.0 vs 49.9) and stay out.
  if (!*policy_ok && NC >= 8 && n_tiles <= 32 && total * 10 >= rounds * slots * 7 && (int64_t)TH * p.OW * 4 >= (int64_t)pl->MFW * WM * 16 * 3)
    *policy_ok = true;
  // Round 6, later: SMALL problems (STDC1-Seg at batch 16: 64x128 / 32x64 / 16x32 maps, profiles/r06_band_small_policy.log). The
  // patch-resident kernel's 16x16 tiles leave most of the chip idle there (32 - 128 tiles) and the per-tap kernel's 256-pixel tiles are few;
  // the row bands still make hundreds of blocks: 256 -> 128 @32x64 30.1 vs 47.6 us, 512 -> 256 @16x32 37.9 vs 76.5, 1024 -> 128 @16x32
  // 69.8 vs 146.2, 128 -> 64 @64x128 30.2 vs 40.4, 64 -> 32 @64x128 14.6 vs 19.0. Up to 32 K output pixels for any width, up to 128 K
  // for <= 64 output channels (128 -> 128 @64x128 and 64 -> 64 @128x256 of DeepLabv3+ still lose and stay out).
  const int64_t Mout = (int64_t)p.NB * p.OH * p.OW;
  if (!*policy_ok && (Mout <= 32768 || (Mout <= 131072 && p.Nout <= 64))) *policy_ok = true;
  return true;
}

static bool band_plan(const IgemmParams& p, BandPlan* pl) {
  if (p.ncls != 1 || p.in_sh != p.in_sw || (p.in_sh != 1 && p.in_sh != 2) || p.out_sh != 1 || p.out_sw != 1) return false;
  const IgemmClass& c = p.cls[0];
  if (p.in_sh == 2 && !(c.dh0 == -1 && c.dw0 == -1 && c.dh_step == 1 && c.dw_step == 1)) return false;   // stride 2: 3x3, padding 1, dilation 1
  if (c.TR != 3 || c.TS != 3 || c.out_oh != 0 || c.out_ow != 0 || c.OHi != p.OH || c.OWi != p.OW || c.M <= 0) return false;
  if (p.pro_scale || p.z_out || p.y2 || p.x_image || p.tail_y) return false;
  if ((p.ep_scale == nullptr) != (p.ep_shift == nullptr)) return false;
  if ((p.bias || p.ep_scale || p.ep_act != CVHIP_ACT_NONE) && p.stats) return false;   // (BN sums are those of the raw accumulators: never with an epilogue)
  if (p.stats && !p.stats_acc) return false;
  if (!p.band_image) return false;   // the fragment-ordered weight copy exists for this layer (conv_plan.h: band_image_fprop / _dgrad)
  if ((p.Cin & 31) || (p.x_ld & 7) || (((uintptr_t)p.x) & 15) || (((uintptr_t)p.w) & 15) || (c.w_off & 7)) return false;
  if (!(p.Nout == 32 || p.Nout == 64 || (p.Nout & 127) == 0)) return false;
  if ((p.y_ld & 3) || (((uintptr_t)p.y) & 7)) return false;
  if (p.res && ((p.res_ld & 3) || (((uintptr_t)p.res) & 7))) return false;
  if (p.OW >= 65536 || p.OW < 1 || (int64_t)p.NB * p.IH * p.IW * (p.x_ld >> 3) >= (1ll << 31)) return false;  // 16-byte units in 31 bits
  // CVHIP_BAND_NW (read per launch): 8 = 512-thread blocks only (one per CU), 4 = 256-thread blocks (two co-resident per CU) wherever the
  // geometry fits, 0 = default: two co-resident blocks where the policy accepts that plan, else one 512-thread block
  const int want_nw = band_env("CVHIP_BAND_NW", 0);
  const bool any = band_mode() >= 2;
  for (int k = 0; k < 2; ++k) {
    const int NW = k == 0 ? 4 : 8;
    if (want_nw == 8 && NW == 4) continue;
    bool ok = false;
    if (!band_plan_nw(p, NW, pl, &ok)) continue;
    if (ok || any) return true;
  }
  return false;
}

template <int WN, int MFW, int PPS, int NF, int LEAD, int PF, int NW, bool EPI = false>
static int band_launch(const BandPlan& pl, hipStream_t stream) {
  auto kern = conv_band_kernel<WN, MFW, PPS, NF, LEAD, PF, NW, EPI>;
  static bool attr_done[64] = {};
  int devid = 0;
  (void)hipGetDevice(&devid);
  bool& attr_set = attr_done[devid & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kBandLdsMax);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(conv_band_kernel)", e);
      return CVHIP_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(pl.a.total_tiles), dim3(NW * 64), pl.lds, stream, pl.a);
  return check_launch("conv_band_kernel");
}

template <int WN, int NW, bool EPI>
static int band_launch_narrow(const BandPlan& pl, hipStream_t stream) {  // NF = 2: 32 channels per wave
  if constexpr (WN <= NW) {
    if (pl.MFW == 7) return pl.PPS == 1 ? band_launch<WN, 7, 1, 2, 2, 0, NW, EPI>(pl, stream) : band_launch<WN, 7, 2, 2, 2, 0, NW, EPI>(pl, stream);
    if (pl.MFW == 10) return pl.PPS == 1 ? band_launch<WN, 10, 1, 2, 2, 0, NW, EPI>(pl, stream) : band_launch<WN, 10, 2, 2, 2, 0, NW, EPI>(pl, stream);
    return pl.PPS == 1 ? band_launch<WN, 13, 1, 2, 2, 0, NW, EPI>(pl, stream) : band_launch<WN, 13, 2, 2, 2, 0, NW, EPI>(pl, stream);
  } else {
    return CVHIP_ERR_UNSUPPORTED;
  }
}

template <int NW, bool EPI>
static int band_launch_nw(const BandPlan& pl, hipStream_t stream) {
  if (pl.WN == 4) return band_launch_narrow<4, NW, EPI>(pl, stream);
  if (pl.WN == 2) return band_launch_narrow<2, NW, EPI>(pl, stream);
  return band_launch_narrow<1, NW, EPI>(pl, stream);
}

// plan query (api.hip cvhip_conv2d_band_plan): {NF, WN, MFW, PPS, PF, TH, bands, n_tiles, total_tiles, lds bytes, PH, PW, NW}
int band_plan_export(const IgemmParams& p, int32_t* out) {
  if (band_mode() == 0) return 0;
  BandPlan pl;
  if (!band_plan(p, &pl)) return 0;
  if (out) {
    const int32_t v[CVHIP_BAND_PLAN_INTS] = {pl.NF, pl.WN, pl.MFW, pl.PPS, pl.PF, pl.a.TH, pl.a.bands, pl.a.n_tiles, pl.a.total_tiles, pl.lds, pl.a.PH, pl.a.PW, pl.NW};
    for (int i = 0; i < CVHIP_BAND_PLAN_INTS; ++i) out[i] = v[i];
  }
  return 1;
}

// -1 = not taken (the caller goes on to the patch-resident / per-tap kernels)
int try_launch_band(const IgemmParams& p, hipStream_t stream) {
  if (band_mode() == 0) return -1;
  BandPlan pl;
  if (!band_plan(p, &pl)) return -1;
  BandArgs& a = pl.a;
  a.x = p.x;
  a.w = p.w + p.cls[0].w_off + (int64_t)p.Nout * 9 * p.Cin;   // the band image behind the row-major one
  a.y = p.y;
  a.stats = p.stats ? reinterpret_cast<double*>(p.stats) : nullptr;
  a.stats_ld = p.stats_ld;
  a.res = p.res;
  a.res_ld = p.res_ld;
  a.ep_scale = p.ep_scale;
  a.ep_shift = p.ep_shift;
  a.bias = p.bias;
  a.ep_act = p.ep_act;
  a.ep_ap = p.ep_ap;
  a.res_pre = p.res_pre;
  if (p.bias || p.ep_scale || p.ep_act != CVHIP_ACT_NONE) return pl.NW == 4 ? band_launch_nw<4, true>(pl, stream) : band_launch_nw<8, true>(pl, stream);
  return pl.NW == 4 ? band_launch_nw<4, false>(pl, stream) : band_launch_nw<8, false>(pl, stream);
}

}  // namespace cvhip
