/* cvhip_probes.h — C ABI of libcvhip_probes.so: measurement and known-answer kernels for gfx950 (MI355X).
 *
 * NOT part of the product library. libcvhip.so (include/cvhip.h) is what a framework binds; this library is what the test-suite
 * and the dev tools under tools/ use to pin the MFMA / LDS-transpose lane layouts the kernels rely on and to measure machine
 * ceilings (LDS read rate, MFMA issue rate, global->LDS paths, atomics) that DESIGN.md's arguments and bench.py's
 * `peaks_measured` object quote. Built by `python -m cvpytorch_amd.build` from csrc/probes.hip alone.
 * Every function returns 0 or a negative CVHIP_ERR_* code (same values as cvhip.h); cvhip_probes_last_error() names the failure.
 */
#ifndef CVHIP_PROBES_H_
#define CVHIP_PROBES_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* cvhip_probes_last_error(void);

/* Lane-layout known answers (cdna_hip_programming.md §3, T10): out buffers are small arrays. */
int cvhip_probe_mfma_16x16x32(const void* a_bf16_16x32, const void* b_bf16_32x16, float* d_16x16,
                              void* stream);
int cvhip_probe_ds_read_tr16(const void* in_bf16_64x4, void* out_bf16_64x4, void* stream);
/* LDS read-bandwidth probe (dev tool, tools/lds_probe.py): `blocks` x 256 threads issue the implicit-GEMM main loop's
 * ds_read_b128 fragment pattern `iters` times. mode 0 swizzled, 1 unswizzled, 2 linear, 3 as 2 x ds_read_b64. */
/* probe: `iters` device-wide barriers inside one persistent launch of `blocks` (<= resident capacity) blocks; every block then
 * sums the other blocks' per-iteration values (checks visibility across the 8 non-coherent L2s). mode 0 = plain stores +
 * __threadfence, 1 = agent-scope atomic stores/loads without fences. `counter_zeroed`: one zeroed uint32; `scratch`: 2*blocks floats. */
int cvhip_probe_grid_barrier(int32_t mode, int32_t iters, int32_t blocks, float* scratch, uint32_t* counter_zeroed, float* out, void* stream);
int cvhip_probe_lds_read_bw(int32_t mode, int32_t iters, int32_t blocks, float* out, void* stream);
/* MFMA issue-rate probe (bench.py's measured attainable peak beside the nominal 2.5 PFLOP/s): `blocks` x 4 waves run `iters`
 * rounds of 8 independent v_mfma_f32_32x32x16_bf16 on register operands; flops = blocks*4*iters*8*32768. `out`: >= blocks floats. */
int cvhip_probe_mfma_peak(int32_t iters, int32_t blocks, float* out, void* stream);
/* Machine-ceiling probes (csrc/probes.hip, tools/ceilings_probe.py -> profiles/r03_ceilings_probe.log). `out`: >= blocks floats.
 * lds_read2 : mode 0 = 16 lane-linear ds_read_b128 per s_waitcnt lgkmcnt(0), 1 = the implicit GEMM's swizzled fragment pattern,
 *             2 = 16 lane-linear ds_read_b64; bytes per launch = blocks * (threads / 64) * iters * 16 KiB (8 KiB for mode 2).
 * mfma_peak2: shape 0 = v_mfma_f32_32x32x16 x 8 accumulators, 1 = v_mfma_f32_16x16x32 x 16 accumulators per round; data 0 zero /
 *             1 small integers / 2 full-range values; flops = blocks * (threads / 64) * iters * 8 * 32768 (16 * 16384 for shape 1).
 * load_path : mode 0 global_load_lds_dwordx4, 1 global_load_dwordx4 -> VGPR, 2 the same + ds_write_b128; every wave moves `iters`
 *             batches of `depth` (4 | 8) KiB from the `span`-byte (power of two) window at src + block * block_stride.
 * atomic_add: `blocks` blocks each add n (<= 256) values into acc[(block % shards) * n ...] with fp64 (f32 = 0) or fp32 atomics. */
int cvhip_probe_lds_read2(int32_t mode, int32_t iters, int32_t blocks, int32_t threads, float* out, void* stream);
int cvhip_probe_mfma_peak2(int32_t shape, int32_t data, int32_t iters, int32_t blocks, int32_t threads, float* out, void* stream);
int cvhip_probe_load_path(int32_t mode, int32_t depth, const void* src, int64_t span, int64_t block_stride, int32_t iters, int32_t blocks,
                          int32_t threads, float* out, void* stream);
int cvhip_probe_atomic_add(int32_t f32, void* acc_zeroed, int32_t shards, int32_t n, int32_t blocks, void* stream);
/* the implicit GEMM's A-tile staging pattern alone: every block fetches its 256 rows (pitch row_stride bytes) in K steps of row_bytes
 * (64 | 128) bytes per row, sweeping k_bytes per row, `depth` K steps in flight per wave; bytes = blocks * iters * 256 * row_bytes */
/* the implicit GEMM's staging structure: flags bit 0 barrier per K step, 1 ring + counted waits, 2 shared weight tile, 3 tap shifts;
 * 64-byte rows, 256 threads; `weights`: >= 128 * k_bytes * 9 bytes; bytes = blocks * iters * (16 KiB + 8 KiB with bit 2) */
int cvhip_probe_stage(int32_t flags, const void* src, int64_t span, const void* weights, int32_t row_stride, int32_t k_bytes, int32_t iters,
                      int32_t blocks, float* out, void* stream);
int cvhip_probe_gather(int32_t row_bytes, const void* src, int64_t span, int32_t row_stride, int32_t k_bytes, int32_t iters, int32_t depth,
                       int32_t blocks, int32_t threads, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVHIP_PROBES_H_ */
