// probes.hip — machine-ceiling micro-benchmarks for gfx950 (dev tools; tools/ceilings_probe.py prints the table kept in
// profiles/). They answer ONE question each, with the issue pattern MI355X_MICROARCH.md prescribes, so that design arguments in
// DESIGN.md rest on what the box delivers and not on what an earlier kernel happened to reach:
//   lds_read2   : ds_read_b128 / ds_read_b64 rate with >= 16 DS operations per s_waitcnt lgkmcnt(0), 4..16 waves per CU
//   mfma_peak2  : back-to-back MFMA issue (32x32x16 or 16x16x32), 1 or 2 waves per SIMD, zero / small / random operands (DVFS)
//   load_path   : bytes per clock per CU delivered by (0) global_load_lds_dwordx4, (1) global_load_dwordx4 -> VGPR,
//                 (2) global_load_dwordx4 -> VGPR -> ds_write_b128, from an L2-resident or an HBM-sized source
//   atomic_f64  : cost of per-block fp64 atomics into sharded per-channel accumulators (BN statistics without finalize launches)
// Built into libcvhip_probes.so (include/cvhip_probes.h), NOT into the product library; bf16 forms only.
#include <string>

#include "common.h"
#include "../../include/cvhip_probes.h"

namespace cvhip {

// this file is a library of its own (libcvhip_probes.so): its launch check and error string are private to it
static thread_local std::string g_probe_error = "";
void set_last_error(const char* what, hipError_t e) { g_probe_error = std::string(what) + ": " + hipGetErrorString(e); }
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(what, e);
    return CVHIP_ERR_LAUNCH;
  }
  return CVHIP_OK;
}

// ---- LDS read rate ------------------------------------------------------------------------------------------------------
// MODE 0: 16 x ds_read_b128, lane-linear (conflict-free by construction)      -> 16 KiB per wave per trip
// MODE 1: 16 x ds_read_b128 in the implicit GEMM's fragment pattern (64-B rows, row = lane & 15, XOR-swizzled 16-B slot)
// MODE 2: 16 x ds_read_b64, lane-linear                                       ->  8 KiB per wave per trip
template <int MODE>
__global__ __launch_bounds__(1024) void probe_lds2_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[32768];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 32768 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  unsigned addr;
  if (MODE == 1) {
    const int swz = (lane >> 4) ^ ((0x78 >> (2 * ((lane >> 2) & 3))) & 3);
    addr = (unsigned)(uintptr_t)smem + (lane & 15) * 64 + swz * 16;  // + j * 1024 (16 rows of 64 B) per read
  } else if (MODE == 2) {
    addr = (unsigned)(uintptr_t)smem + lane * 8;
  } else {
    addr = (unsigned)(uintptr_t)smem + lane * 16;
  }
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      uint2 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, ra, rb, rc, rd, re, rf;
      asm volatile(
          "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:512\n ds_read_b64 %2, %16 offset:1024\n ds_read_b64 %3, %16 offset:1536\n"
          "ds_read_b64 %4, %16 offset:2048\n ds_read_b64 %5, %16 offset:2560\n ds_read_b64 %6, %16 offset:3072\n ds_read_b64 %7, %16 offset:3584\n"
          "ds_read_b64 %8, %16 offset:4096\n ds_read_b64 %9, %16 offset:4608\n ds_read_b64 %10, %16 offset:5120\n ds_read_b64 %11, %16 offset:5632\n"
          "ds_read_b64 %12, %16 offset:6144\n ds_read_b64 %13, %16 offset:6656\n ds_read_b64 %14, %16 offset:7168\n ds_read_b64 %15, %16 offset:7680\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(ra),
            "=&v"(rb), "=&v"(rc), "=&v"(rd), "=&v"(re), "=&v"(rf)
          : "v"(addr)
          : "memory");
      acc ^= r0.x ^ r5.y ^ rf.x;
    } else {
      uint4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, ra, rb, rc, rd, re, rf;
      asm volatile(
          "ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n"
          "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n"
          "ds_read_b128 %8, %16 offset:8192\n ds_read_b128 %9, %16 offset:9216\n ds_read_b128 %10, %16 offset:10240\n ds_read_b128 %11, %16 offset:11264\n"
          "ds_read_b128 %12, %16 offset:12288\n ds_read_b128 %13, %16 offset:13312\n ds_read_b128 %14, %16 offset:14336\n ds_read_b128 %15, %16 offset:15360\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(ra),
            "=&v"(rb), "=&v"(rc), "=&v"(rd), "=&v"(re), "=&v"(rf)
          : "v"(addr)
          : "memory");
      acc ^= r0.x ^ r5.y ^ rf.w;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

// ---- MFMA issue rate ----------------------------------------------------------------------------------------------------
// SHAPE 0: v_mfma_f32_32x32x16 (8 independent accumulators)   SHAPE 1: v_mfma_f32_16x16x32 (16 independent accumulators)
// data: 0 = zero operands, 1 = small integers, 2 = hashed full-range values in [-1, 1) (what a real kernel's clock sees)
template <int SHAPE>
__global__ __launch_bounds__(512) void probe_mfma2_kernel(float* out, int iters, int data) {
  const int l = threadIdx.x + blockIdx.x * 977;
  h16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    float fa = 0.f, fb = 0.f;
    if (data == 1) {
      fa = (float)((l * 7 + e * 3) % 17 - 8);
      fb = (float)((l * 5 + e) % 13 - 6);
    } else if (data == 2) {
      const unsigned ha = (unsigned)(l * 8 + e) * 2654435761u, hb = (unsigned)(l * 8 + e + 77) * 2246822519u;
      fa = (float)(int)(ha >> 8) * (1.f / 8388608.f) - 1.f;
      fb = (float)(int)(hb >> 8) * (1.f / 8388608.f) - 1.f;
    }
    a[e] = (h16_t)fa;
    b[e] = (h16_t)fb;
  }
  float s = 0.f;
  if (SHAPE == 0) {
    f32x16 c[8];
    for (int j = 0; j < 8; ++j)
      for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = CVHIP_MFMA_32X32X16(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j)
      for (int e = 0; e < 16; ++e) s += c[j][e];
  } else {
    f32x4 c[16];
    for (int j = 0; j < 16; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c[j] = CVHIP_MFMA_16X16X32(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  }
  if (s == 123456.789f) out[blockIdx.x] = s;
}

// ---- global -> on-chip load path ------------------------------------------------------------------------------------------
// Every wave moves `iters` batches of D x 1 KiB (64 lanes x 16 B, full 128-byte lines) and waits for the batch; with 4-16 waves per
// CU the batches of different waves overlap, which is how the implicit GEMM's ring behaves. `span` (bytes, power of two): the
// region a block walks — small (<= 1 MiB, shared by all blocks) = L2-resident after the first touch, large = HBM streaming.
template <int MODE, int D>
__global__ __launch_bounds__(1024) void probe_load_kernel(const unsigned char* __restrict__ src, size_t span, size_t block_stride,
                                                          int iters, float* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * block_stride;
  unsigned char* my = smem + ((wave * D * 1024) & 65535);
  unsigned acc = 0;
  size_t off = (size_t)wave * D * 1024;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const unsigned char* g = base + ((off + d * 1024 + lane * 16) & (span - 1));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = *reinterpret_cast<const uint4*>(base + ((off + d * 1024 + lane * 16) & (span - 1)));
      if (MODE == 2) {
#pragma unroll
        for (int d = 0; d < D; ++d) *reinterpret_cast<uint4*>(my + d * 1024 + lane * 16) = v[d];
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d].x ^ v[d].w;
      }
    }
    off += (size_t)nw * D * 1024;
  }
  if (MODE == 2) {
    __syncthreads();
    acc ^= reinterpret_cast<const unsigned*>(smem)[t];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

// ---- the implicit GEMM's A-tile staging pattern in isolation ---------------------------------------------------------------
// A block owns 256 pixel rows (row pitch `row_stride` bytes = channels x 2) and sweeps their channels in K steps of ROWB bytes
// (64 = BK 32: a DMA instruction covers 16 rows x 64 B, i.e. HALF cache lines; 128 = BK 64: 8 rows x FULL lines); per K step the
// block's waves fetch the whole 256 x ROWB tile into LDS with global_load_lds_dwordx4, `depth` K steps in flight per wave.
// Isolates what the access pattern alone costs on the L2 -> LDS path (the contiguous form of probe_load_kernel reaches
// 50+ B/clk/CU; the implicit GEMM's staging-only ablation ~16).
template <int ROWB>
__global__ __launch_bounds__(512) void probe_gather_kernel(const unsigned char* __restrict__ src, size_t span, int row_stride, int k_bytes,
                                                           int iters, int depth, float* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nw = blockDim.x >> 6;
  constexpr int RPI = 1024 / ROWB;             // rows per DMA instruction
  constexpr int LPR = ROWB / 16;               // lanes per row
  const int pieces = 256 / RPI / nw;           // DMA instructions per wave per K step
  const size_t tile0 = (size_t)blockIdx.x * 256 * (size_t)row_stride;
  int koff = 0;
  for (int it = 0; it < iters; ++it) {
    for (int pc = 0; pc < pieces; ++pc) {
      const int row = (pc * nw + wave) * RPI + lane / LPR;
      const unsigned char* g = src + ((tile0 + (size_t)row * row_stride + koff + (lane % LPR) * 16) & (span - 1));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(smem + (((it * pieces + pc) * nw + wave) & 63) * 1024), 16, 0, 0);
    }
    koff += ROWB;
    if (koff >= k_bytes) koff = 0;
    if ((it + 1) % depth == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<const unsigned*>(smem)[t] == 0x12345678u) out[blockIdx.x] = 1.f;
}

// ---- the implicit GEMM's staging STRUCTURE in isolation ------------------------------------------------------------------------
// probe_gather_kernel plus, one flag at a time, what the real kernel does around the same loads: flags bit 0 = one s_barrier per K
// step (after the wait), bit 1 = ring with counted waits (two K steps stay in flight, vmcnt(PER)) instead of batch + drain,
// bit 2 = a weight tile as well (128 rows x 64 B per K step from ONE 128 x k_bytes*9 matrix shared by every block, pitch 2304 B
// for k_bytes 256), bit 3 = the pixel rows shift by a tap offset every k_bytes / 64 steps (-41 .. +41 rows, as a 3x3 kernel on a
// 40-wide map does). 256 threads, 64-B rows (BK 32): 4 pixel-tile pieces (+ 2 weight pieces) per wave per K step.
__global__ __launch_bounds__(256, 2) void probe_stage_kernel(const unsigned char* __restrict__ src, size_t span, const unsigned char* __restrict__ wsrc,
                                                             int row_stride, int k_bytes, int iters, int flags, float* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * 24576];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool bar = flags & 1, ring = flags & 2, wt = flags & 4, taps = flags & 8;
  const size_t tile0 = (size_t)blockIdx.x * 256 * (size_t)row_stride;
  const int steps_per_tap = k_bytes / 64;
  const int wpitch = steps_per_tap * 64 * 9;
  int koff = 0, tap = 0, kstep = 0;
  auto issue = [&](int it) {
    const int shift = taps ? ((tap / 3 - 1) * 40 + (tap % 3 - 1)) : 0;
    unsigned char* const st = smem + (it % 3) * 24576;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int row = (pc * 4 + wave) * 16 + (lane >> 2) + shift + 41;
      const unsigned char* g = src + ((tile0 + (size_t)row * row_stride + koff + (lane & 3) * 16) & (span - 1));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(st + (pc * 4 + wave) * 1024), 16, 0, 0);
    }
    if (wt) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int row = (pc * 4 + wave) * 16 + (lane >> 2);
        const unsigned char* g = wsrc + (size_t)row * wpitch + (size_t)(tap * steps_per_tap + kstep) * 64 + (lane & 3) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(st + 16384 + (pc * 4 + wave) * 1024), 16, 0, 0);
      }
    }
    koff += 64;
    if (++kstep == steps_per_tap) {
      kstep = 0;
      koff = 0;
      if (++tap == 9) tap = 0;
    }
  };
  if (ring) {
    issue(0);
    issue(1);
    for (int it = 0; it < iters; ++it) {
      if (wt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (bar) __builtin_amdgcn_s_barrier();
      issue(it + 2);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      issue(it);
      if (it & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (bar) __builtin_amdgcn_s_barrier();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<const unsigned*>(smem)[t] == 0x12345678u) out[blockIdx.x] = 1.f;
}

// ---- fp64 atomics into sharded accumulators --------------------------------------------------------------------------------
// block b adds `n` values (n <= 256: threads t < n) into acc[(b % shards) * n + t]: the traffic of a conv epilogue that folds its
// tile's BatchNorm sums (2 x K values) straight into per-layer accumulators. F32 != 0: the same with fp32 atomics.
template <int F32>
__global__ __launch_bounds__(256) void probe_atomic_kernel(void* acc, int shards, int n) {
  const int t = threadIdx.x;
  if (t >= n) return;
  const size_t i = (size_t)(blockIdx.x % shards) * n + t;
  if (F32) unsafeAtomicAdd(reinterpret_cast<float*>(acc) + i, 1.0f + (float)t);
  else unsafeAtomicAdd(reinterpret_cast<double*>(acc) + i, 1.0 + (double)t);
}

// ---- probes ---------------------------------------------------------------------------------------
__global__ void probe_mfma_kernel(const h16_t* a, const h16_t* b, float* d) {
  // a: [16][32] row-major (i,k); b: [32][16] row-major (k,j). Lane l supplies A[i=l&15][k=8*(l>>4)+e],
  // B[k=8*(l>>4)+e][j=l&15]; result reg r -> D[row = 4*(l>>4)+r][col = l&15].
  const int l = threadIdx.x;
  h16x8 fa, fb;
  for (int e = 0; e < 8; ++e) {
    fa[e] = a[(l & 15) * 32 + 8 * (l >> 4) + e];
    fb[e] = b[(8 * (l >> 4) + e) * 16 + (l & 15)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = CVHIP_MFMA_16X16X32(fa, fb, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// LDS read-bandwidth probe: the igemm main loop's ds_read_b128 pattern without MFMA / staging.
//   mode 0: the kernel's swizzled fragment reads   mode 1: same rows, no swizzle   mode 2: linear lane*16 (conflict-free by construction)
//   mode 3: swizzled pattern issued as 2 x ds_read_b64
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe_lds_bw_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[49152];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 49152 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int swz = (MODE == 1) ? (lane >> 4) : ((lane >> 4) ^ ((0x78 >> (2 * ((lane >> 2) & 3))) & 3));
  const int a_row = wm * 64 + (lane & 15), b_row = wn * 64 + (lane & 15);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* sA = smem + (it % 3) * 16384;
    const unsigned char* sB = sA + 8192;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 2) {
        const uint4 u = *reinterpret_cast<const uint4*>(sA + j * 1024 + lane * 16 + wave * 4096 % 8192);
        const uint4 v = *reinterpret_cast<const uint4*>(sB + j * 1024 + lane * 16);
        acc ^= u.x ^ u.w ^ v.y ^ v.z;
      } else if (MODE == 3) {
        const uint2 u0 = *reinterpret_cast<const uint2*>(sA + (a_row + j * 16) * 64 + swz * 16);
        const uint2 u1 = *reinterpret_cast<const uint2*>(sA + (a_row + j * 16) * 64 + swz * 16 + 8);
        const uint2 v0 = *reinterpret_cast<const uint2*>(sB + (b_row + j * 16) * 64 + swz * 16);
        const uint2 v1 = *reinterpret_cast<const uint2*>(sB + (b_row + j * 16) * 64 + swz * 16 + 8);
        acc ^= u0.x ^ u1.y ^ v0.y ^ v1.x;
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(sA + (a_row + j * 16) * 64 + swz * 16);
        const uint4 v = *reinterpret_cast<const uint4*>(sB + (b_row + j * 16) * 64 + swz * 16);
        acc ^= u.x ^ u.w ^ v.y ^ v.z;
      }
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}



// MFMA issue-rate probe (bench.py "attainable peak"): every wave runs `iters` rounds of 8 INDEPENDENT v_mfma_f32_32x32x16_bf16
// (8 accumulator sets: no dependent-accumulator stalls), operands in registers, nothing else. 4 waves per block, one per SIMD.
// flops per launch = blocks * 4 waves * iters * 8 * (2*32*32*16).
__global__ __launch_bounds__(256) void probe_mfma_peak_kernel(float* out, int iters) {
  const int l = threadIdx.x;
  h16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (h16_t)(0.001f * (float)((l * 7 + e * 3) % 17 - 8));
    b[e] = (h16_t)(0.002f * (float)((l * 5 + e) % 13 - 6));
  }
  f32x16 c[8];
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = CVHIP_MFMA_32X32X16(a, b, c[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 16; ++e) s += c[j][e];
  if (s == 123456.789f) out[blockIdx.x] = s;  // keeps the MFMAs live without a store on the normal path
}

typedef __attribute__((address_space(3))) h16x4 lds_h16x4_t;
__global__ void probe_tr16_kernel(const h16_t* in, h16_t* out) {
  // in: 64 lanes x 4 bf16 written linearly to LDS (lane l at byte l*8); every lane then issues
  // ds_read_b64_tr_b16 at its own linear address; out[l][0..3] = what lane l received.
  __shared__ __attribute__((aligned(16))) h16_t lds[256];
  const int l = threadIdx.x;
  for (int e = 0; e < 4; ++e) lds[l * 4 + e] = in[l * 4 + e];
  __syncthreads();
  h16x4 v = CVHIP_DS_READ_TR16_B64((lds_h16x4_t*)(&lds[l * 4]));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

// Grid-barrier probe (tools/barrier_probe.py): is a device-wide barrier inside a persistent kernel cheap enough on MI355X (8 XCDs,
// non-coherent L2s) to fuse reduce-then-apply passes? mode 0: partials by plain stores + __threadfence() both sides;
// mode 1: partials by agent-scope atomic stores / loads (cache-bypassing), no fence. The counter self-resets (generation = target).
__global__ __launch_bounds__(256) void probe_grid_barrier_kernel(int mode, int iters, float* scratch, unsigned* counter, float* out) {
  const int t = threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* part = scratch + (size_t)(it & 1) * gridDim.x;
    const float mine = (float)(blockIdx.x + 1 + it);
    if (t == 0) {
      if (mode == 0) {
        part[blockIdx.x] = mine;
        __threadfence();
      } else {
        __hip_atomic_store(&part[blockIdx.x], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned target = (unsigned)(it + 1) * gridDim.x;
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
      if (mode == 0) __threadfence();
    }
    __syncthreads();
    float s = 0.f;
    for (int i = t; i < (int)gridDim.x; i += 256)
      s += mode == 0 ? part[i] : __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += s;
    __syncthreads();
  }
  __shared__ float red[256];
  red[t] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s += red[i];
    out[blockIdx.x] = s;
  }
}

}  // namespace cvhip

using namespace cvhip;

extern "C" {

const char* cvhip_probes_last_error(void) { return g_probe_error.c_str(); }

int cvhip_probe_mfma_16x16x32(const void* a, const void* b, float* d, void* stream) {
  if (!a || !b || !d) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const h16_t*)a, (const h16_t*)b, d);
  return check_launch("probe_mfma_kernel");
}

int cvhip_probe_ds_read_tr16(const void* in, void* out, void* stream) {
  if (!in || !out) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const h16_t*)in, (h16_t*)out);
  return check_launch("probe_tr16_kernel");
}

int cvhip_probe_grid_barrier(int32_t mode, int32_t iters, int32_t blocks, float* scratch, uint32_t* counter_zeroed, float* out, void* stream) {
  if (!scratch || !counter_zeroed || !out || iters <= 0 || blocks <= 0 || blocks > 1024) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(probe_grid_barrier_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, iters, scratch, counter_zeroed, out);
  return check_launch("probe_grid_barrier_kernel");
}

int cvhip_probe_mfma_peak(int32_t iters, int32_t blocks, float* out, void* stream) {
  if (!out || iters <= 0 || blocks <= 0) return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(probe_mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return check_launch("probe_mfma_peak_kernel");
}

int cvhip_probe_lds_read_bw(int32_t mode, int32_t iters, int32_t blocks, float* out, void* stream) {
  if (!out || iters <= 0 || blocks <= 0) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(probe_lds_bw_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(probe_lds_bw_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(probe_lds_bw_kernel<2>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    case 3: hipLaunchKernelGGL(probe_lds_bw_kernel<3>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    default: return CVHIP_ERR_INVALID;
  }
  return check_launch("probe_lds_bw_kernel");
}


int cvhip_probe_lds_read2(int32_t mode, int32_t iters, int32_t blocks, int32_t threads, float* out, void* stream) {
  if (!out || iters <= 0 || blocks <= 0 || threads < 64 || threads > 1024 || threads % 64) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(probe_lds2_kernel<0>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(probe_lds2_kernel<1>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(probe_lds2_kernel<2>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    default: return CVHIP_ERR_INVALID;
  }
  return check_launch("probe_lds2_kernel");
}

int cvhip_probe_mfma_peak2(int32_t shape, int32_t data, int32_t iters, int32_t blocks, int32_t threads, float* out, void* stream) {
  if (!out || iters <= 0 || blocks <= 0 || threads < 64 || threads > 512 || threads % 64 || data < 0 || data > 2) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (shape == 0) hipLaunchKernelGGL(probe_mfma2_kernel<0>, dim3(blocks), dim3(threads), 0, st, out, iters, data);
  else if (shape == 1) hipLaunchKernelGGL(probe_mfma2_kernel<1>, dim3(blocks), dim3(threads), 0, st, out, iters, data);
  else return CVHIP_ERR_INVALID;
  return check_launch("probe_mfma2_kernel");
}

int cvhip_probe_load_path(int32_t mode, int32_t depth, const void* src, int64_t span, int64_t block_stride, int32_t iters, int32_t blocks,
                          int32_t threads, float* out, void* stream) {
  if (!src || !out || iters <= 0 || blocks <= 0 || threads < 64 || threads > 1024 || threads % 64) return CVHIP_ERR_INVALID;
  if (span < 65536 || (span & (span - 1)) || block_stride < 0) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* s = (const unsigned char*)src;
#define CVHIP_PL(M, D) hipLaunchKernelGGL((probe_load_kernel<M, D>), dim3(blocks), dim3(threads), 0, st, s, (size_t)span, (size_t)block_stride, iters, out)
  if (depth == 4) {
    if (mode == 0) CVHIP_PL(0, 4); else if (mode == 1) CVHIP_PL(1, 4); else if (mode == 2) CVHIP_PL(2, 4); else return CVHIP_ERR_INVALID;
  } else if (depth == 8) {
    if (mode == 0) CVHIP_PL(0, 8); else if (mode == 1) CVHIP_PL(1, 8); else if (mode == 2) CVHIP_PL(2, 8); else return CVHIP_ERR_INVALID;
  } else {
    return CVHIP_ERR_INVALID;
  }
#undef CVHIP_PL
  return check_launch("probe_load_kernel");
}

int cvhip_probe_atomic_add(int32_t f32, void* acc_zeroed, int32_t shards, int32_t n, int32_t blocks, void* stream) {
  if (!acc_zeroed || shards <= 0 || n <= 0 || n > 256 || blocks <= 0) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (f32) hipLaunchKernelGGL(probe_atomic_kernel<1>, dim3(blocks), dim3(256), 0, st, acc_zeroed, shards, n);
  else hipLaunchKernelGGL(probe_atomic_kernel<0>, dim3(blocks), dim3(256), 0, st, acc_zeroed, shards, n);
  return check_launch("probe_atomic_kernel");
}

int cvhip_probe_gather(int32_t row_bytes, const void* src, int64_t span, int32_t row_stride, int32_t k_bytes, int32_t iters, int32_t depth,
                       int32_t blocks, int32_t threads, float* out, void* stream) {
  if (!src || !out || iters <= 0 || depth <= 0 || blocks <= 0 || (threads != 256 && threads != 512)) return CVHIP_ERR_INVALID;
  if (span < 65536 || (span & (span - 1)) || row_stride < row_bytes || k_bytes < row_bytes || k_bytes > row_stride) return CVHIP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* s = (const unsigned char*)src;
  if (row_bytes == 64) hipLaunchKernelGGL(probe_gather_kernel<64>, dim3(blocks), dim3(threads), 0, st, s, (size_t)span, row_stride, k_bytes, iters, depth, out);
  else if (row_bytes == 128) hipLaunchKernelGGL(probe_gather_kernel<128>, dim3(blocks), dim3(threads), 0, st, s, (size_t)span, row_stride, k_bytes, iters, depth, out);
  else return CVHIP_ERR_INVALID;
  return check_launch("probe_gather_kernel");
}

int cvhip_probe_stage(int32_t flags, const void* src, int64_t span, const void* weights, int32_t row_stride, int32_t k_bytes, int32_t iters,
                      int32_t blocks, float* out, void* stream) {
  if (!src || !weights || !out || iters <= 0 || blocks <= 0 || span < 65536 || (span & (span - 1)) || k_bytes < 64 || k_bytes % 64 || k_bytes > row_stride)
    return CVHIP_ERR_INVALID;
  hipLaunchKernelGGL(probe_stage_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, (size_t)span,
                     (const unsigned char*)weights, row_stride, k_bytes, iters, flags, out);
  return check_launch("probe_stage_kernel");
}

}  // extern "C"
