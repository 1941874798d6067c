// conv_patch.hip — patch-resident implicit-GEMM convolution for gfx950 (CDNA4): multi-tap (3x3, dilated 3x3, 2x2 / 1x2 dgrad parity
// classes ...) fprop and dgrad whose gathered operand has a channel count that is a multiple of 32.
//
//   Out[m][n] = sum_{t,c} X[pix(m) + tap(t)][c] * Wt[n][t*Cin + c]              (same plan / operand images as conv_igemm.hip)
//
// Why a second implicit GEMM: conv_igemm.hip stages the pixel (A) tile once PER TAP — 9x for a 3x3 — as 64-byte rows, and sits on the
// L2 -> LDS rate of that pattern (profiles/r03_stage_structure_probe.log: 17-24 B/clk/CU; the dominant kernel ran 1.09 waves per SIMD
// at 21 % MFMA-pipe busy for three rounds). Here the block's INPUT PATCH (tile + halo) of one channel chunk is staged into the LDS
// ONCE (1.3-1.6x the tile instead of 9x), pixel-major, and every tap reads its fragments from it at a wave-uniform pixel offset; only
// the weight tile of a (tap, chunk) K step streams through a small LDS-DMA ring (128-byte rows when Cin % 64 == 0). Because the patch
// passes through registers exactly once per element, the producer layer's BatchNorm scale/shift + activation can be applied on the
// way in (PRO = 1: z = act(scale*y + shift), halo zero AFTER the activation; conv_module.py:201-214 `act(norm(conv(x)))` of the
// layer below), and the activated interior can be written out once for the weight-gradient pass (z_out).
//
// Geometry. Output positions are tiled in a VIRTUAL row space: image n's output rows sit at virtual rows n*vho .. n*vho + OHi - 1
// (vho >= OHi: a few dummy rows per image, computed and not stored), its input rows at virtual input rows n*vho*in_sh + (ih - lo_h).
// A tile is TH virtual output rows x TW output columns (TH*TW <= 256); its patch is PH = (TH-1)*in_sh + EH virtual input rows x
// PWc = (TW-1)*in_sw + EW columns and needs no per-image special case: rows that fall between images are zero rows (halo), tiles
// may span images. Output position ml -> (th, tw) = (ml / TW, ml % TW); lane's patch pixel for tap (i, j) = base(th, tw) + a
// wave-uniform offset. Stride-2 inputs keep even and odd patch columns in separate halves of a patch row so the 16 pixels of an MFMA
// fragment stay consecutive in the LDS.
//
// LDS image: pixel rows of 64 bytes (32 channels per chunk), 16-byte slot s of pixel pp stored at slot s ^ (((pp >> 2) & 1) << 1).
// Unlike the 4-slot table of conv_igemm.hip (conflict-free only for rows aligned to 16) this one keeps the 4x16-lane groups of
// ds_read_b128 conflict-free at ANY pixel offset — taps shift a fragment's 16 consecutive pixels by arbitrary amounts — (exhaustive
// check over offsets and lane groups: DESIGN.md §3) and the 8-lane groups of ds_write_b128 (two whole rows each) conflict-free too.
//
// Round-4 measurements that shaped it (profiles/r04_a_patch_bench.log): the first form — 8 waves x (64 x 64) wave tiles, 64-channel
// chunks, ONE block per CU (147 KB of LDS) — was correct and SLOWER than the per-tap kernel (128->128 3x3 @40x40: 54.8 vs 45.7 us):
// 16 fragment reads per 32 MFMAs and every wave of the CU in lock-step behind one barrier per K step, with the block's prologue and
// epilogue exposed. This form keeps the per-tap kernel's proven compute shape — 256 threads = 4 waves, 128 x 64 wave tiles (12
// fragment reads per 32 MFMAs), 32-deep K steps, two blocks per CU (<= 80 KB of LDS each) whose barriers, prologues and epilogues
// overlap — and replaces only what the round-3 ablations blamed: the per-tap re-staging of the pixel tile.
//
// 256 threads = 4 waves (2 x 2 for 128 output channels, 4 x 1 below), v_mfma_f32_16x16x32 with swapped operands (a lane ends up with
// 4 consecutive output channels of one pixel). The epilogue stages the output tile through the (dead) LDS and
// leaves as 16-byte row stores; it can add a bias, apply a folded / eval-mode BatchNorm scale+shift and an activation, add a residual
// (dgrad skip-connection gradient) and emit training-mode BatchNorm sums.
//
// Replaces aten::convolution / convolution_backward(input) reached from reference src/models/bricks/conv_module.py:209 and
// trainer.py:189; the fused prologue/epilogue replace native_batch_norm + silu/relu of conv_module.py:210-213 and the folded
// conv+act of src/utils/fuse.py:32-54.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_plan.h"

namespace cvhip {

constexpr int kPatchBM = 256;
constexpr int kPatchTabC = 768;   // PRO: scale | shift of up to this many input channels live in the LDS

struct PatchClass {
  int TR, TS, dh0, dh_step, dw0, dw_step;  // taps: input row = oh*in_sh + dh0 + i*dh_step, column likewise
  int out_oh, out_ow;                      // output pixel = (oh*out_sh + out_oh, ow*out_sw + out_ow)
  int OHi, OWi;                            // iteration grid of the class
  int lo_h, lo_w;                          // smallest tap offsets (row / column)
  int TH, TW;                              // tile: virtual output rows x output columns
  int PH, PW, PWh, PWc;                    // patch rows, row pitch (pixels), half pitch (stride-2 de-interleave), valid columns
  int vho;                                 // virtual output rows per image
  int per_image;                           // 1: vho is a multiple of TH — tiles never span images, patch rows outside the image are zero
  int tiles_w;                             // column tiles per row band
  int tile_begin;                          // first logical tile of the class
  int64_t w_off;
};

struct PatchArgs {
  const h16_t* x;
  const h16_t* w;
  h16_t* y;
  const float* bias;
  int bias_n;
  float* stats;
  int stats_ld, stats_acc;
  int NB, IH, IW, Cin, x_ld, in_sh, in_sw;
  int Nout, y_ld, OH, OW, out_sh, out_sw;
  int n_tiles, total_tiles, ncls;
  const h16_t* res;
  int res_ld;
  int res_pre;  // `res` joins before the epilogue's activation
  // prologue (PRO = 1): x holds the RAW convolution output of the producing layer; the patch loader applies act(scale*x + shift)
  const float *pro_scale, *pro_shift;
  float pro_ap;
  h16_t* z_out;  // optional: the activated interior of the patch is stored here (same geometry as x: pitch z_ld)
  int z_ld;
  // epilogue: out = act((acc + bias) * ep_scale + ep_shift) (+ res); ep_scale / ep_shift optional (both or none)
  const float *ep_scale, *ep_shift;
  int ep_act;
  float ep_ap;
  unsigned long long* dbg;  // census (dev tool): per block {HW_ID, XCC_ID, start, end} when non-null
  PatchClass cls[kKernelClasses];
};

#define CVHIP_PGLDS16(src, dst)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// masked patch pixels (halo outside the image, positions past the patch) read zeros from here: every DMA lane always has a source
__device__ __attribute__((aligned(64))) unsigned int g_patch_zero[16];

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a literal): waits until at most min(n, 15) are outstanding
__device__ __forceinline__ void patch_wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  }
}

template <int N>
__device__ __forceinline__ void patch_wait_lit() {
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

constexpr int kPatchThreads = 512;
constexpr int patch_np(int CK) { return 3; }  // loader passes: 3 x 128 pixels = 384 pixels (24 KB per buffer)
constexpr int patch_pix(int CK) { return patch_np(CK) * (kPatchThreads / (CK / 8)); }
constexpr int patch_b_rows(int BN, int CK) {
  const int rpt = (kPatchThreads / 64) * (1024 / (CK * 2));  // rows one pass of the block's waves covers
  return BN < rpt ? rpt : BN;
}
constexpr int patch_lds_bytes(int BN, int CK, int PRO, int PB, int NST, int TPS) {
  const int ring = PB * patch_pix(CK) * CK * 2 + NST * TPS * patch_b_rows(BN, CK) * CK * 2 + (PRO ? 2 * kPatchTabC * 4 : 0);
  const int epi = 5 * BN * 4 + kPatchBM * (BN * 2 + 16);  // constants + staged output tile
  return ring > epi ? ring : epi;
}


// activation of NV values with ONE switch (a switch per element multiplies the unrolled epilogue's code size)
template <int NV>
__device__ __forceinline__ void patch_act_vec(float (&v)[NV], int act, float ap) {
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

// 16-byte slot swizzle of a pixel / weight row
template <int CK>
__device__ __forceinline__ int patch_swz(int row) {
  static_assert(CK == 32, "64-byte rows");
  return ((row >> 2) & 1) << 1;
}

// BN  : output channels per block (128 / 64 / 32)          CK : channels per patch chunk (64 / 32)
// PRO : 0 = x is used as it is; 1 = x is a raw convolution output, act(scale*x + shift) applied in place once a chunk has landed
// PB  : patch buffers (2: the next chunk's patch lands in the other buffer while the current one is multiplied)
// NST : depth of the weight-tile DMA ring
// TPS : taps per K step / barrier (1, or 2: 64-deep steps — half the barriers per MFMA; the weight ring then holds 2-tap slots)
template <int BN, int CK, int PRO, int ACT, int PB, int NST, int TPS = 1>
__global__ __launch_bounds__(kPatchThreads, 4) void conv_patch_kernel(const PatchArgs p) {
  constexpr int NW = kPatchThreads / 64;
  constexpr int WAVES_N = BN == 128 ? 2 : 1, WAVES_M = NW / WAVES_N;
  constexpr int WM = kPatchBM / WAVES_M, WN = BN / WAVES_N;  // 64 x 64 | 32 x 64 | 32 x 32
  constexpr int MF = WM / 16, NF = WN / 16;
  static_assert(NF >= 1 && BN * 0 == 0, "BN >= 32");
  constexpr int SLOTS = CK / 8;
  constexpr int ROWB = CK * 2;
  constexpr int PPP = kPatchThreads / SLOTS;
  constexpr int NPL = patch_np(CK);
  constexpr int PATCH_BYTES = patch_pix(CK) * ROWB;
  constexpr int RPI = 1024 / ROWB;  // weight rows per DMA instruction
  constexpr int B_ROWS = patch_b_rows(BN, CK);
  constexpr int PERB = B_ROWS / (NW * RPI);  // DMA instructions per K step per wave
  constexpr int B_BYTES = B_ROWS * ROWB;     // one tap's weight tile
  constexpr int SLOT_BYTES = TPS * B_BYTES;  // one ring slot = the tiles of one K step
  constexpr int PERS = TPS * PERB;           // DMA instructions per K step per wave
  constexpr int KS = CK / 32;
  static_assert(NST >= 2 && NST <= 4 && PB == 2, "ring depths (the next chunk's patch lands while the current one is multiplied)");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[patch_lds_bytes(BN, CK, PRO, PB, NST, TPS)];
  unsigned char* const sPatch = smem;
  unsigned char* const sB = smem + PB * PATCH_BYTES;
  float* const sTab = reinterpret_cast<float*>(sB + NST * SLOT_BYTES);  // PRO: [scale | shift][kPatchTabC]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const unsigned long long dbg_t0 = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const int g = lane >> 4;

  // ---- which tile ---------------------------------------------------------------------------------------------
  const int lt = xcd_remap(blockIdx.x, p.total_tiles);
  int ci = 0;
#pragma unroll
  for (int i = 1; i < kKernelClasses; ++i)
    if (i < p.ncls && lt >= p.cls[i].tile_begin) ci = i;
  PatchClass cl = p.cls[0];  // by value, picked with compile-time indices (a run-time index would put the argument block in scratch)
#pragma unroll
  for (int i = 1; i < kKernelClasses; ++i)
    if (ci == i) cl = p.cls[i];
  const int local = lt - cl.tile_begin;
  const int sp = local / p.n_tiles;
  const int ntile = local - sp * p.n_tiles;
  const int thi = sp / cl.tiles_w, twi = sp - thi * cl.tiles_w;
  const int n0 = ntile * BN;
  const int Gv0 = thi * cl.TH, ow0 = twi * cl.TW;
  const int V0 = Gv0 * p.in_sh;
  const int col0 = ow0 * p.in_sw + cl.lo_w;
  const int pitch = cl.vho * p.in_sh;
  const int T = cl.TR * cl.TS;
  const int Cin = p.Cin;
  const int NC = Cin / CK;
  const int SPC = (T + TPS - 1) / TPS;  // K steps per chunk
  const int nk = SPC * NC;
  const int Ktot = T * Cin;
  const int PW = cl.PW;

  // ---- patch loader: this thread owns slot lslot of pixels lpix + j*PPP ------------------------------------------
  const int lslot = t % SLOTS, lpix = t / SLOTS;
  int poff[NPL];      // pixel index into x, -1 = outside the image (zero)
  unsigned own = 0;   // bit j: this patch pixel is an output position of the tile (z_out writes it)
  {
    const int npix = cl.PH * PW;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int pp = j * PPP + lpix;
      const int pr = pp / PW;
      const int q = pp - pr * PW;
      const int pc = p.in_sw == 2 ? (q < cl.PWh ? 2 * q : 2 * (q - cl.PWh) + 1) : q;
      const int V = V0 + pr;
      const int n = cl.per_image ? Gv0 / cl.vho : V / pitch;  // (per-image tiles: a row past the bottom must not wrap into image n + 1)
      const int ih = V - n * pitch + cl.lo_h;
      const int iw = col0 + pc;
      const bool ok = pp < npix && pc < cl.PWc && n < p.NB && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      poff[j] = ok ? (n * p.IH + ih) * p.IW + iw : -1;
      if (PRO == 1) {
        const int th = pr + cl.lo_h, tw = pc + cl.lo_w;  // (stride-1 "same" geometry only: the host checks)
        if (ok && ntile == 0 && (unsigned)th < (unsigned)cl.TH && (unsigned)tw < (unsigned)cl.TW) own |= 1u << j;
      }
    }
  }
  // The patch of a chunk goes global -> LDS by DMA (no VGPR hop: the 128 x 64 accumulator tile leaves no registers for a staged
  // copy), lane-linear: DMA instruction j of wave w fills pixels j*PPP + w*16 .. +15, lane l the PHYSICAL slot l % 4 of pixel
  // l / 4 — i.e. exactly the (pixel, slot) this thread owns — and fetches the LOGICAL slot that belongs there (rule 21: swizzle on
  // the source side). Always NPL instructions per wave, so the vmcnt arithmetic of the main loop is shape-independent.
  const int lswz = patch_swz<CK>(lpix);          // (pp >> 2) & 1 does not depend on the pass: PPP % 8 == 0
  const int lch = (lslot ^ lswz) * 8;            // first channel (inside a chunk) of the 16 bytes this thread owns
  const h16_t* const zsrc = reinterpret_cast<const h16_t*>(g_patch_zero) + lch;
  auto issue_patch = [&](int buf, int c, bool live) {
    const h16_t* const base = p.x + (c * CK + lch);
    unsigned char* const dst = sPatch + buf * PATCH_BYTES + wave * (16 * ROWB);
#pragma unroll
    for (int j = 0; j < NPL; ++j) CVHIP_PGLDS16((live && poff[j] >= 0) ? base + (int64_t)poff[j] * p.x_ld : zsrc, dst + j * (PPP * ROWB));
  };
  // PRO = 1: once the chunk's raw values have landed, every thread turns ITS OWN 16-byte pieces into act(scale*y + shift) in place
  // (only this wave's vmcnt orders the reads behind the DMA; no other thread touches these bytes), zeroes the halo again — the
  // activation of the zero page is act(shift), not 0 — and stores the activated interior once for the weight-gradient pass
  auto transform_patch = [&](int buf, int c) {
    if constexpr (PRO == 1) {
      const int ch = c * CK + lch;
      float sc[8], sh[8];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(sTab + ch), a1 = *reinterpret_cast<const f32x4*>(sTab + ch + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sTab + kPatchTabC + ch), b1 = *reinterpret_cast<const f32x4*>(sTab + kPatchTabC + ch + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = a0[e];
        sc[4 + e] = a1[e];
        sh[e] = b0[e];
        sh[4 + e] = b1[e];
      }
      unsigned char* const own_base = sPatch + buf * PATCH_BYTES + lpix * ROWB + lslot * 16;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        unsigned char* const q = own_base + j * (PPP * ROWB);
        f32x8 f = unpack8(*reinterpret_cast<const uint4*>(q));
#pragma unroll
        for (int e = 0; e < 8; ++e) f.v[e] = act_fwd(f.v[e] * sc[e] + sh[e], ACT, p.pro_ap);
        uint4 v = pack8(f);
        const bool ok = poff[j] >= 0;
        v.x = ok ? v.x : 0u;
        v.y = ok ? v.y : 0u;
        v.z = ok ? v.z : 0u;
        v.w = ok ? v.w : 0u;
        *reinterpret_cast<uint4*>(q) = v;
        if (p.z_out && ((own >> j) & 1u)) *reinterpret_cast<uint4*>(p.z_out + (int64_t)poff[j] * p.z_ld + ch) = v;
      }
    }
  };

  // ---- weight-tile DMA: lane fetches the LOGICAL 16-byte slot that belongs at its physical position (rule 21) -----
  const h16_t* bsrc[PERB];
  {
    const h16_t* const wbase = p.w + cl.w_off;
#pragma unroll
    for (int i = 0; i < PERB; ++i) {
      const int row = i * NW * RPI + wave * RPI + lane / SLOTS;
      int n = n0 + row;
      n = n < p.Nout ? n : p.Nout - 1;  // rows past the tile / past Nout fetch a valid row: their columns are never stored or summed
      const int ls = (lane % SLOTS) ^ patch_swz<CK>(row);
      bsrc[i] = wbase + ((int64_t)n * Ktot + ls * 8);
    }
  }
  int nb_c = 0, nb_t = 0;  // (chunk, first tap) of the next K step to stage
  auto issue_b = [&](int st) {
    unsigned char* const dst = sB + st * SLOT_BYTES;
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const int tap = nb_t + u < T ? nb_t + u : nb_t;  // an odd tap count's last step stages its one tap twice (uniform DMA count)
      const int off = tap * Cin + nb_c * CK;
#pragma unroll
      for (int i = 0; i < PERB; ++i) CVHIP_PGLDS16(bsrc[i] + off, dst + u * B_BYTES + (i * NW * RPI + wave * RPI) * ROWB);
    }
    nb_t += TPS;
    if (nb_t >= T) {
      nb_t = 0;
      ++nb_c;
    }
  };

  // ---- fragment geometry ------------------------------------------------------------------------------------------
  int abase[MF];  // BYTE address of (this lane's pixel, logical slot g) before the swizzle
#pragma unroll
  for (int b = 0; b < MF; ++b) {
    const int ml = wm * WM + b * 16 + (lane & 15);
    const int th = ml / cl.TW, tw = ml - th * cl.TW;
    abase[b] = (th < cl.TH ? th * p.in_sh * PW + tw : 0) * ROWB + (g << 4);  // rows past the tile read pixel 0 (never stored or summed)
  }
  int baddr[NF];
#pragma unroll
  for (int a = 0; a < NF; ++a) {
    const int row = wn * WN + a * 16 + (lane & 15);
    baddr[a] = row * ROWB + ((g ^ patch_swz<CK>(row)) << 4);
  }

  f32x4 acc[NF][MF];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int st, int u, int pbuf, int toff) __attribute__((always_inline)) {
    const unsigned char* const sA = sPatch + pbuf * PATCH_BYTES;
    const unsigned char* const sBt = sB + st * SLOT_BYTES + u * B_BYTES;
    // swizzle of 64-byte rows: slot ^= 2 * bit 2 of the pixel index = byte-address bit 5 ^= byte-address bit 8
    int aaddr[MF];
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      const int u = abase[b] + toff * ROWB;
      aaddr[b] = u ^ ((u >> 3) & 32);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      h16x8 xa[MF], wb[NF];
#pragma unroll
      for (int a = 0; a < NF; ++a) wb[a] = *reinterpret_cast<const h16x8*>(sBt + (baddr[a] ^ (ks * 64)));
#pragma unroll
      for (int b = 0; b < MF; ++b) xa[b] = *reinterpret_cast<const h16x8*>(sA + (aaddr[b] ^ (ks * 64)));
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] = CVHIP_MFMA_16X16X32(wb[a], xa[b], acc[a][b], 0, 0, 0);
    }
  };

  // ---- prologue ----------------------------------------------------------------------------------------------------
  if constexpr (PRO == 1) {
    for (int c = t; c < Cin; c += kPatchThreads) {
      sTab[c] = p.pro_scale[c];
      sTab[kPatchTabC + c] = p.pro_shift[c];
    }
  }
  issue_patch(0, 0, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (PRO == 1) {
    __syncthreads();  // the constants are in the LDS (everything this wave issued has landed: a plain barrier)
    transform_patch(0, 0);
  }
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue_b(s);

  // ---- main loop: chunk-major, K steps of TPS taps inside ------------------------------------------------------------
  // Per-wave VMEM queue, oldest first, at the top of K step k (step si of chunk c): [weight tiles of step k] [steps k+1 .. k+NST-2]
  // and, for 1 <= si <= NST-1, the NPL patch DMAs of chunk c+1 (issued in the chunk's first step right after step k+NST-1's tiles).
  // Loads return in order, so "at most (younger instructions) outstanding" means step k's tiles have landed.
  int k = 0;
  int st_cur = 0, st_nxt = NST - 1;
  int tr = 0, ts = 0;
  auto tap_off = [&]() {
    const int cw = cl.dw0 + ts * cl.dw_step - cl.lo_w;
    return (cl.dh0 + tr * cl.dh_step - cl.lo_h) * PW + (p.in_sw == 2 ? ((cw & 1) * cl.PWh + (cw >> 1)) : cw);
  };
  auto next_tap = [&]() {
    if (++ts == cl.TS) {
      ts = 0;
      ++tr;
    }
  };
  auto step_top = [&](int si, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
    // (literal counts behind two uniform branches; near the end of the K loop, where fewer younger steps exist, wait for everything)
    if (k + NST - 2 < nk) {
      if (!FIRST && si <= NST - 1) patch_wait_lit<(NST - 2) * PERS + NPL>();
      else patch_wait_lit<(NST - 2) * PERS>();
    } else {
      patch_wait_lit<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's patch stores (chunk boundary) are in the LDS
    __builtin_amdgcn_s_barrier();  // step k landed everywhere; everybody finished reading ring slot st_nxt (step k - 1)
    if (k + NST - 1 < nk) issue_b(st_nxt);
  };
  auto step_body = [&](int si, int pbuf) __attribute__((always_inline)) {
    compute(st_cur, 0, pbuf, tap_off());
    next_tap();
    if constexpr (TPS == 2) {
      if (si * 2 + 1 < T) {
        compute(st_cur, 1, pbuf, tap_off());
        next_tap();
      }
    }
    st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
    st_nxt = st_nxt == NST - 1 ? 0 : st_nxt + 1;
    ++k;
  };
  for (int c = 0; c < NC; ++c) {
    const bool more = c + 1 < NC;
    const int pbuf = PB == 2 ? (c & 1) : 0;
    tr = ts = 0;
    int pafter = 0;  // DMA instructions this wave issued after the next chunk's patch DMAs
    step_top(0, std::true_type{});
    issue_patch(PB == 2 ? ((c + 1) & 1) : 0, c + 1, more);
    step_body(0, pbuf);
    for (int si = 1; si < SPC; ++si) {
      if (k + NST - 1 < nk) pafter += PERS;
      step_top(si, std::false_type{});
      step_body(si, pbuf);
    }
    if (more) {
      // this wave's patch DMAs have landed once only the DMAs issued after them are outstanding (with fewer than NST steps per
      // chunk the wait for the next weight tiles would not cover them); the next step's barrier publishes them to the other waves
      patch_wait_vm(pafter);
      transform_patch((c + 1) & 1, c + 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  float* const sconst = reinterpret_cast<float*>(smem);  // [bias | ep_scale | ep_shift][BN]
  const bool has_ss = p.ep_scale != nullptr;
  if (t < BN) {
    const int n = n0 + t;
    const int nc = n < p.Nout ? n : p.Nout - 1;
    sconst[t] = (p.bias && n < p.bias_n) ? p.bias[n] : 0.f;
    sconst[BN + t] = has_ss ? p.ep_scale[nc] : 1.f;
    sconst[2 * BN + t] = has_ss ? p.ep_shift[nc] : 0.f;
  }
  __syncthreads();
  constexpr int EP_PITCH = BN * 2 + 16;
  unsigned char* const tile = smem + 5 * BN * (int)sizeof(float);
  const int nq = g * 4;
  const bool plain = !p.bias && !has_ss && p.ep_act == CVHIP_ACT_NONE && !p.res_pre;
  const bool rvec = p.res && (p.res_ld & 3) == 0 && ((((uintptr_t)p.res) & 7) == 0);
  bool rok[MF];
#pragma unroll
  for (int b = 0; b < MF; ++b) {
    const int row = wm * WM + b * 16 + (lane & 15);
    const int th = row / cl.TW, tw = row - th * cl.TW;
    const int Gv = Gv0 + th;
    const int n_img = Gv / cl.vho;
    const int oh = Gv - n_img * cl.vho;
    const int ow = ow0 + tw;
    rok[b] = th < cl.TH && n_img < p.NB && oh < cl.OHi && ow < cl.OWi;
    const h16_t* rbase = nullptr;
    if (p.res && rok[b]) {
      const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
      rbase = p.res + opix * p.res_ld;
    }
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const int nl = wn * WN + a * 16 + nq;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (!plain) {  // block-uniform
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sconst + nl);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sconst + BN + nl), tv = *reinterpret_cast<const f32x4*>(sconst + 2 * BN + nl);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] + bv[r]) * sv[r] + tv[r];
        if (rbase && p.res_pre && n0 + nl < p.Nout) {  // residual before the activation (ResNet bottleneck tail)
          const h16_t* rrow = rbase + n0 + nl;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + nl + r < p.Nout) v[r] += (float)rrow[r];
        }
        patch_act_vec<4>(v, p.ep_act, p.ep_ap);
      }
      if (rbase && !p.res_pre && n0 + nl < p.Nout) {
        const h16_t* rrow = rbase + n0 + nl;
        if (rvec) {
          const uint2 u = *reinterpret_cast<const uint2*>(rrow);
          float r0, r1, r2, r3;
          unpack2(u.x, r0, r1);
          unpack2(u.y, r2, r3);
          v[0] += r0;
          v[1] += r1;
          v[2] += r2;
          v[3] += r3;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + nl + r < p.Nout) v[r] += (float)rrow[r];
        }
      }
      uint2 u;
      u.x = pack2(v[0], v[1]);
      u.y = pack2(v[2], v[3]);
      *reinterpret_cast<uint2*>(tile + row * EP_PITCH + nl * 2) = u;
    }
  }
  __syncthreads();
  {
    constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
    const bool vec_ok = (p.Nout & 7) == 0 && (p.y_ld & 7) == 0 && ((((uintptr_t)p.y) & 15) == 0);
    for (int idx = t; idx < kPatchBM * CPR; idx += kPatchThreads) {
      const int row = idx / CPR, ch = idx - row * CPR;
      const int th = row / cl.TW, tw = row - th * cl.TW;
      const int Gv = Gv0 + th;
      const int n_img = Gv / cl.vho;
      const int oh = Gv - n_img * cl.vho;
      const int ow = ow0 + tw;
      const int n = n0 + ch * 8;
      if (!(th < cl.TH && n_img < p.NB && oh < cl.OHi && ow < cl.OWi) || n >= p.Nout) continue;
      const int64_t opix = ((int64_t)n_img * p.OH + (oh * p.out_sh + cl.out_oh)) * p.OW + (ow * p.out_sw + cl.out_ow);
      h16_t* const yrow = p.y + opix * p.y_ld + n;
      const unsigned char* const src = tile + row * EP_PITCH + ch * 16;
      if (vec_ok) {
        *reinterpret_cast<uint4*>(yrow) = *reinterpret_cast<const uint4*>(src);
      } else {
        const h16_t* const sv = reinterpret_cast<const h16_t*>(src);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.Nout) yrow[e] = sv[e];
      }
    }
  }
  if (p.stats) {  // training-mode BatchNorm sums of the fp32 accumulators (valid output positions only)
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);  // [WAVES_M][BN][2]
#pragma unroll
    for (int a = 0; a < NF; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          const float v = rok[b] ? acc[a][b][r] : 0.f;
          s1 += v;
          s2 += v * v;
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if ((lane & 15) == 0) {
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
          acc_add2(reinterpret_cast<double*>(p.stats), sp, p.stats_ld, n, s1, s2);
        } else {
          float* dst = p.stats + (int64_t)sp * 2 * p.Nout;
          dst[n] = s1;
          dst[p.Nout + n] = s2;
        }
      }
    }
  }
  if (p.dbg && t == 0) {
    unsigned long long* d = p.dbg + (size_t)blockIdx.x * 4;
    d[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    d[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    d[2] = dbg_t0;
    d[3] = __builtin_amdgcn_s_memtime();
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------

static int patch_mode() {  // CVHIP_PATCH: 0 = never (the per-tap implicit GEMM runs), 1 = default policy, 2 = wherever the geometry allows
  const char* e = getenv("CVHIP_PATCH");  // read per call (host-side, once per launch / plan query): the tests switch it
  return e ? atoi(e) : 1;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

// tile search for one class: TH x TW <= 256 with the patch inside the LDS budget; fewest tiles wins, then the smaller patch
static bool patch_plan_class(const IgemmClass& c, int NB, int IH, int in_sh, int in_sw, int max_pix, int n_tiles, PatchClass* o) {
  memset(o, 0, sizeof(*o));
  o->TR = c.TR;
  o->TS = c.TS;
  o->dh0 = c.dh0;
  o->dh_step = c.dh_step;
  o->dw0 = c.dw0;
  o->dw_step = c.dw_step;
  o->out_oh = c.out_oh;
  o->out_ow = c.out_ow;
  o->OHi = c.OHi;
  o->OWi = c.OWi;
  o->w_off = c.w_off;
  if (c.M <= 0 || c.TR <= 0 || c.TS <= 0) return false;
  const int h_a = c.dh0, h_b = c.dh0 + (c.TR - 1) * c.dh_step;
  const int w_a = c.dw0, w_b = c.dw0 + (c.TS - 1) * c.dw_step;
  const int lo_h = imin(h_a, h_b), hi_h = imax(h_a, h_b), lo_w = imin(w_a, w_b), hi_w = imax(w_a, w_b);
  const int EH = hi_h - lo_h + 1, EW = hi_w - lo_w + 1;
  o->lo_h = lo_h;
  o->lo_w = lo_w;
  // virtual input rows per image: real rows keep their place (pitch >= IH - lo_h) and a row past the bottom of image n either stays
  // inside n's slot (invalid) or lands on a row above the top of image n + 1 (pitch > largest row any tap reaches)
  const int ih_max = (c.OHi - 1) * in_sh + hi_h;
  int pitch = imax(imax(IH - lo_h, ih_max + 1), 1);
  if (lo_h > 0) pitch = imax(pitch, IH);  // (rows below lo_h are never read; keep the mapping monotone)
  pitch = (pitch + in_sh - 1) / in_sh * in_sh;
  o->vho = pitch / in_sh;
  if (o->vho < c.OHi) return false;
  const int64_t rows_total = (int64_t)NB * o->vho;
  if (rows_total * in_sh >= (1ll << 30)) return false;
  // Candidates: TH x TW <= 256 whose patch fits, in two row layouts — tiles over the VIRTUAL rows of the whole batch (may span
  // images) or per image (vho rounded up to a multiple of TH). Cost = rounds of the 512 resident-block slots (two 72-KB blocks per
  // CU: a launch of <= 512 blocks takes about one block time whatever its size, 513 take two), +3 % when a fragment's 16 output
  // positions are not 16 consecutive pixels of one row (LDS bank conflicts); ties: fewer blocks, then the smaller patch.
  int best_tw = 0, best_th = 0, best_pix = 0, best_mode = 0;
  int64_t best_tiles = -1, best_cost = -1;
  const int vho_v = o->vho;
  for (int TW = 1; TW <= imin(c.OWi, kPatchBM); ++TW) {
    const int th_max = kPatchBM / TW;
    for (int mode = 0; mode < 2; ++mode) {
      for (int TH = imin(th_max, mode ? c.OHi : (int)imin64(rows_total, th_max)); TH >= 1; --TH) {
        const int PWc = (TW - 1) * in_sw + EW;
        const int PWh = (PWc + 1) / 2;
        const int PW = in_sw == 2 ? 2 * PWh : PWc;
        const int PH = (TH - 1) * in_sh + EH;
        if (PH * PW > max_pix) continue;
        const int64_t row_tiles = mode ? (int64_t)NB * ((c.OHi + TH - 1) / TH) : (rows_total + TH - 1) / TH;
        const int64_t tiles = row_tiles * ((c.OWi + TW - 1) / TW);
        const int64_t blocks = tiles * n_tiles;
        const int64_t cost = ((blocks + 511) / 512) * 512 * (TW % 16 == 0 ? 100 : 103);
        const int pix = PH * PW;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && (tiles < best_tiles || (tiles == best_tiles && pix < best_pix)))) {
          best_cost = cost;
          best_tiles = tiles;
          best_tw = TW;
          best_th = TH;
          best_pix = pix;
          best_mode = mode;
        }
        if (mode == 0 || TH == 1) break;                           // virtual rows: a smaller TH only means more tiles
        if ((c.OHi + TH - 2) / (TH - 1) != (c.OHi + TH - 1) / TH) break;  // per image: stop once the row-tile count would grow
      }
    }
  }
  if (best_tiles < 0) return false;
  o->TH = best_th;
  o->TW = best_tw;
  o->per_image = best_mode;
  if (best_mode) o->vho = ((c.OHi + best_th - 1) / best_th) * best_th;
  (void)vho_v;
  o->PWc = (best_tw - 1) * in_sw + EW;
  o->PWh = (o->PWc + 1) / 2;
  o->PW = in_sw == 2 ? 2 * o->PWh : o->PWc;
  o->PH = (best_th - 1) * in_sh + EH;
  o->tiles_w = (c.OWi + best_tw - 1) / best_tw;
  return true;
}

struct PatchPlan {
  PatchArgs a;
  int BN, CK;
  int64_t useful, padded;  // output positions: real / computed
};

// fills the plan when the patch kernel takes this problem (geometry only; pointers are copied by the launcher)
static bool patch_plan(const IgemmParams& p, PatchPlan* pl, bool any_geometry = false) {
  if (p.ncls < 1 || p.ncls > kKernelClasses) return false;
  if (p.in_sh != p.in_sw || p.in_sh != 1) return false;  // stride-2 inputs: not enabled yet
  if (p.Cin % 32 != 0 || (p.x_ld & 7) != 0) return false;
  if (p.tail_y) return false;
  const int CK = 32;
  const int BN = p.Nout <= 32 ? 32 : p.Nout <= 64 ? 64 : 128;
  memset(&pl->a, 0, sizeof(pl->a));
  PatchArgs& a = pl->a;
  a.NB = p.NB;
  a.IH = p.IH;
  a.IW = p.IW;
  a.Cin = p.Cin;
  a.x_ld = p.x_ld;
  a.in_sh = p.in_sh;
  a.in_sw = p.in_sw;
  a.Nout = p.Nout;
  a.y_ld = p.y_ld;
  a.OH = p.OH;
  a.OW = p.OW;
  a.out_sh = p.out_sh;
  a.out_sw = p.out_sw;
  a.n_tiles = cdiv(p.Nout, BN);
  a.ncls = 0;
  int total = 0;
  int taps_max = 0;
  pl->useful = pl->padded = 0;
  if ((int64_t)p.NB * p.IH * p.IW >= (1ll << 31)) return false;
  for (int i = 0; i < p.ncls; ++i) {
    const IgemmClass& c = p.cls[i];
    if (c.M <= 0) continue;           // empty class (parity row/column past the image)
    if (c.TR * c.TS <= 0) return false;  // a class without taps just writes zeros: the general kernel does that
    PatchClass pc;
    if (!patch_plan_class(c, p.NB, p.IH, p.in_sh, p.in_sw, patch_pix(CK), a.n_tiles, &pc)) return false;
    pc.tile_begin = total;
    const int64_t rows_total = (int64_t)p.NB * pc.vho;
    const int64_t sp = ((rows_total + pc.TH - 1) / pc.TH) * pc.tiles_w;
    if (sp * a.n_tiles + total >= (1ll << 30)) return false;
    total += (int)sp * a.n_tiles;
    pl->useful += c.M;
    pl->padded += sp * kPatchBM;
    taps_max = imax(taps_max, c.TR * c.TS);
    a.cls[a.ncls++] = pc;
  }
  if (a.ncls == 0) return false;
  for (int i = a.ncls; i < kKernelClasses; ++i) a.cls[i] = a.cls[0];
  a.total_tiles = total;
  pl->BN = BN;
  pl->CK = CK;
  if (taps_max < 2 && patch_mode() < 2) return false;  // single-tap problems have nothing to re-use
  if (any_geometry || patch_mode() >= 2) return true;
  // Default policy = where it measured FASTER than the per-tap kernel (profiles/r04_g_patch_bench.log, isolated launches on rotating
  // operands; 8 waves x 64x64 wave tiles, 2-tap K steps): 128-wide output tiles of a single stride-1 class whose blocks all fit
  // the 512 resident slots at once — 128->128 3x3 @40x40 b64 43.8 vs 45.4 us, @64x128 b16 45.2 vs 50.2, 256->256 @32x64 46.8 vs 50.4.
  // It LOSES on 64- / 32-wide outputs (more fragment reads per MFMA: 60.9 vs 58.7, 117 vs 86 us), on the stride-2 dgrad classes
  // (129.7 vs 109.5) and whenever the tile count spills into a second round of slots (512->512 dilated: 592 blocks, 189 vs 154).
  if (BN != 128 || a.ncls != 1) return false;
  if ((int64_t)a.total_tiles > 512) return false;
  if (pl->padded * 2 > pl->useful * 3) return false;  // virtual rows / partial tiles must not waste more than a third of the MFMA work
  return true;
}

template <int BN, int CK, int PRO, int ACT>
static int patch_launch_cfg(const PatchArgs& a, hipStream_t stream) {
  // taps per K step: 2 = 64-deep steps with a 2-deep ring of 2-tap slots (half the barriers; same 80 KB of LDS: two blocks per CU). The
  // prologue form has no room for its constants beside that ring and keeps 1-tap steps with a 3-deep ring.
  const dim3 grid(a.total_tiles), block(kPatchThreads);
  if constexpr (PRO == 0) {
    hipLaunchKernelGGL((conv_patch_kernel<BN, CK, PRO, ACT, 2, 2, 2>), grid, block, 0, stream, a);
    return check_launch("conv_patch_kernel(tps2)");
  }
  hipLaunchKernelGGL((conv_patch_kernel<BN, CK, PRO, ACT, 2, 3, 1>), grid, block, 0, stream, a);
  return check_launch("conv_patch_kernel");
}

template <int BN, int CK>
static int patch_launch_pro(const PatchArgs& a, int pro_act, hipStream_t stream) {
  if (!a.pro_scale) return patch_launch_cfg<BN, CK, 0, CVHIP_ACT_NONE>(a, stream);
  switch (pro_act) {
    case CVHIP_ACT_SILU: return patch_launch_cfg<BN, CK, 1, CVHIP_ACT_SILU>(a, stream);
    case CVHIP_ACT_RELU: return patch_launch_cfg<BN, CK, 1, CVHIP_ACT_RELU>(a, stream);
    case CVHIP_ACT_NONE: return patch_launch_cfg<BN, CK, 1, CVHIP_ACT_NONE>(a, stream);
    default: return CVHIP_ERR_UNSUPPORTED;
  }
}

static unsigned long long* g_patch_dbg = nullptr;
#ifndef CVHIP_F16
extern "C" void cvhip_patch_debug_buffer(void* p) { g_patch_dbg = (unsigned long long*)p; }  // dev tool (tools/patch_census.py; bf16 build)
#endif

// geometry-only query (plan queries of api.hip, the Python host's kernel labels): does the patch kernel take this plan?
bool patch_takes(const IgemmParams& p, int* stats_rows) {
  const bool need = p.pro_scale || p.z_out;
  if (patch_mode() == 0 && !need) return false;
  PatchPlan pl;
  if (!patch_plan(p, &pl, need)) return false;
  if (stats_rows) *stats_rows = pl.a.total_tiles / pl.a.n_tiles;
  return true;
}

int patch_plan_export(const IgemmParams& p, int32_t* out, int max_classes, bool any_geometry) {
  if (patch_mode() == 0) return 0;
  PatchPlan pl;
  if (!patch_plan(p, &pl, any_geometry)) return 0;
  if (!out || max_classes < pl.a.ncls) return CVHIP_ERR_INVALID;
  for (int i = 0; i < pl.a.ncls; ++i) {
    const PatchClass& c = pl.a.cls[i];
    int32_t* o = out + i * CVHIP_PATCH_CLASS_INTS;
    const int32_t v[CVHIP_PATCH_CLASS_INTS] = {c.TR, c.TS, c.dh0, c.dh_step, c.dw0, c.dw_step, c.out_oh, c.out_ow, c.OHi, c.OWi, c.lo_h, c.lo_w,
                                               c.TH, c.TW, c.PH, c.PW, c.PWh, c.PWc, c.vho, c.tiles_w, c.tile_begin,
                                               (int32_t)(c.w_off & 0xffffffffll), (int32_t)(c.w_off >> 32), pl.a.n_tiles, pl.a.total_tiles,
                                               pl.BN, pl.CK, patch_pix(pl.CK), c.per_image, 0};
    for (int j = 0; j < CVHIP_PATCH_CLASS_INTS; ++j) o[j] = v[j];
  }
  return pl.a.ncls;
}

// -1 = not taken (the caller runs the per-tap implicit GEMM). A plan that carries a fused prologue (pro_scale / z_out) can only run
// here: CVHIP_ERR_UNSUPPORTED then, and the caller falls back to separate passes.
int try_launch_patch(const IgemmParams& p, hipStream_t stream) {
  if (p.y2) return -1;  // (split stores exist in the streaming 1x1 kernel only)
  const bool need = p.pro_scale || p.z_out;
  if (patch_mode() == 0 && !need) return -1;
  PatchPlan pl;
  if (!patch_plan(p, &pl, need)) return need ? CVHIP_ERR_UNSUPPORTED : -1;  // (a prologue exists only here: the speed policy does not apply)
  PatchArgs& a = pl.a;
  a.x = p.x;
  a.w = p.w;
  a.y = p.y;
  a.bias = p.bias;
  a.bias_n = p.bias_n;
  a.stats = p.stats;
  a.stats_ld = p.stats_ld;
  a.stats_acc = p.stats_acc;
  a.res = p.res;
  a.res_ld = p.res_ld;
  a.res_pre = (p.res && p.res_pre) ? 1 : 0;
  a.pro_scale = p.pro_scale;
  a.pro_shift = p.pro_shift;
  a.pro_ap = p.pro_ap;
  const int pro_act = p.pro_act;
  a.z_out = p.z_out;
  a.z_ld = p.z_ld;
  a.ep_scale = p.ep_scale;
  a.ep_shift = p.ep_shift;
  a.ep_act = p.ep_act;
  a.ep_ap = p.ep_ap;
  a.dbg = g_patch_dbg;
  if (a.pro_scale) {
    if (!a.pro_shift || p.Cin > kPatchTabC) return CVHIP_ERR_UNSUPPORTED;
    if (a.z_out) {
      // the activated interior is written at the pixels that are output positions: "same" geometry of a single stride-1 class
      const PatchClass& c = a.cls[0];
      if (a.ncls != 1 || p.out_sh != 1 || p.out_sw != 1 || c.OHi != p.IH || c.OWi != p.IW || (a.z_ld & 7) || (((uintptr_t)a.z_out) & 15))
        return CVHIP_ERR_UNSUPPORTED;
    }
  } else if (a.z_out) {
    return CVHIP_ERR_INVALID;
  }
  if ((a.ep_scale == nullptr) != (a.ep_shift == nullptr)) return CVHIP_ERR_INVALID;
  if (a.stats && (a.ep_scale || a.ep_act != CVHIP_ACT_NONE)) return CVHIP_ERR_INVALID;  // the sums are those of the raw accumulators
  if (pl.BN == 128) return patch_launch_pro<128, 32>(a, pro_act, stream);
  if (pl.BN == 64) return patch_launch_pro<64, 32>(a, pro_act, stream);
  return patch_launch_pro<32, 32>(a, pro_act, stream);
}

}  // namespace cvhip
