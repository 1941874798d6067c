// Dev probe (not part of the product): fp32 atomic flushes of weight-gradient partial tiles —
//   mode 0: agent-scope atomics into ONE accumulator (what the weight-gradient kernels do today)
//   mode 1: workgroup-scope atomics into the accumulator copy of the block's own XCD (XCC_ID), 8 copies
//   mode 2: agent-scope atomics into the XCD's copy (separates "scope" from "fewer writers per address")
// Each block adds 1.0f to every float of tile (blockIdx % ntiles); afterwards sum over copies must equal blocks / ntiles exactly.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/xcd_atomics tools/probes/xcd_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void flush_kernel(float* acc, int tile_floats, int ntiles, size_t copy_stride) {
  const int tile = blockIdx.x % ntiles;
  int xcd = 0;
  if (MODE != 0) xcd = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;  // HW_REG_XCC_ID
  float* dst = acc + (size_t)xcd * copy_stride + (size_t)tile * tile_floats;
  for (int i = threadIdx.x; i < tile_floats; i += 256) {
    if (MODE == 1) __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void xcd_census(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int copies = 8;
  int census[64];
  int* dc;
  CK(hipMalloc(&dc, sizeof(census)));
  hipLaunchKernelGGL(xcd_census, dim3(64), dim3(64), 0, 0, dc);
  CK(hipMemcpy(census, dc, sizeof(census), hipMemcpyDeviceToHost));
  printf("XCC_ID of blocks 0..23:");
  for (int i = 0; i < 24; ++i) printf(" %d", census[i]);
  printf("\n");
  const int cfgs[][3] = {{16384, 4, 768}, {16384, 16, 768}, {16384, 36, 768}, {9216, 32, 256}, {16384, 4, 256}, {4096, 64, 1024}};
  for (auto& c : cfgs) {
    const int tile_floats = c[0], ntiles = c[1], blocks = c[2];
    const size_t stride = (size_t)tile_floats * ntiles;
    float* acc;
    CK(hipMalloc(&acc, stride * copies * sizeof(float)));
    std::vector<float> h(stride * copies);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      bool ok = true;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(acc, 0, stride * copies * sizeof(float)));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(blocks), dim3(256), 0, 0, acc, tile_floats, ntiles, stride);
        else if (mode == 1) hipLaunchKernelGGL(flush_kernel<1>, dim3(blocks), dim3(256), 0, 0, acc, tile_floats, ntiles, stride);
        else hipLaunchKernelGGL(flush_kernel<2>, dim3(blocks), dim3(256), 0, 0, acc, tile_floats, ntiles, stride);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(h.data(), acc, stride * copies * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < stride; ++i) {
          float s = 0.f;
          for (int k = 0; k < copies; ++k) s += h[k * stride + i];
          const int tile = (int)(i / tile_floats);
          const int expect = blocks / ntiles + (tile < blocks % ntiles ? 1 : 0);
          if (s != (float)expect) {
            if (ok) printf("  MISMATCH mode %d at %zu: %g != %d\n", mode, i, s, expect);
            ok = false;
            break;
          }
        }
      }
      const double mb = (double)blocks * tile_floats * 4 / 1e6;
      printf("tile %6d floats x %2d tiles, %4d blocks (%.1f MB of adds)  mode %d: %7.1f us  %.2f TB/s  %s\n", tile_floats, ntiles, blocks, mb, mode,
             best * 1e3, mb / (best * 1e3), ok ? "exact" : "WRONG");
    }
    CK(hipFree(acc));
  }
  return 0;
}
