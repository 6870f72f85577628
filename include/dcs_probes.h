/*
 * dcs_probes.h -- measurement apparatus of the MI355X steering-coefficient generator.
 * NOT part of the product ABI (include/dcs_beamformer.h): nothing a caller of the hot path
 * needs is here.  Library: probes/libdcs_probes.so (`python -m probes.build`), which holds
 *   - the dcs_probe_* entry points below (probes/bf_probes.hip), and
 *   - a second build of the product sources with -DDCS_PROBES, i.e. the whole dcs_bf_* API plus the
 *     measurement knobs of struct dcs_probe_knobs below (store-only skeleton of the generator, sleeps
 *     before each store, the coefficient-reuse beamformer's A/B switches, an injected launch failure for
 *     the error-path tests).  libdcs_beamformer.so has none of them: they are compiled out of its
 *     kernels, argument structs and context (ABI 3; ABI 2 carried two of them in dcs_bf_tuning).
 * Users: tests/ (device sincos sweep, whole-tensor checksum of the full-size configs) and
 * tools/measure.py (the store-pattern studies behind profiles/r01_store_patterns.md).
 */
#ifndef DCS_PROBES_H
#define DCS_PROBES_H

#include "dcs_beamformer.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the libraries are built -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

/* Measurement knobs of a context of THIS library (all 0 = the product's behaviour).  NULL resets them. */
struct dcs_probe_knobs {
    int32_t nomath;         /* generator: addressing and stores only, no arithmetic (the store ceiling of a geometry) */
    int32_t pace;           /* generator: 64-cycle sleeps before each store of the fast loop (0..4096) */
    int32_t fail_at_step;   /* per-time-step launch loops (NAIVE, MULTIPLE_CHANNELS): the launch of step
                             * fail_at_step - 1 reports hipErrorLaunchFailure without being enqueued (0 = never) */
    /* dcs_bf_beamform_accumulated, int8 form (profiles/r02_fused.md, r03_fused.md): */
    int32_t bacc_probe;     /* 1 = stores only, 2 = loads and stores without arithmetic, 3 = stores without
                             * coefficients either, 4 = as 3 with one contiguous KiB per store instruction */
    int32_t bacc_rounds;    /* cap on the rounds of sample blocks per workgroup (0 = the launcher's choice) */
    int32_t bacc_no_share;  /* staged form: every wave makes all its coefficients */
    int32_t bacc_plain;     /* ordinary instead of nontemporal stores */
    int32_t bacc_wg_per_cu; /* staged form: at most this many workgroups resident per CU */
    int32_t bacc_unstaged;  /* <= 64 antennas: operands straight from global memory instead of through LDS */
    int32_t bacc_order;     /* workgroup numbering: 0 = the product's choice, 1 = as dispatched (round 2), 2 = one contiguous
                             * eighth of the order per XCD, 3 = the workgroups sharing a channel's samples always on one XCD */
    int32_t bacc_nbt;       /* staged form: beam tiles per workgroup (1, 2, 4, or 8 with eight-wave workgroups; 0 = the launcher's choice) */
    int32_t bacc_waves;     /* staged form: waves per workgroup (8 or 16; 0 = the launcher's choice, 4) */
};
int dcs_probe_set_knobs(dcs_bf_context *ctx, const struct dcs_probe_knobs *k);

/* Device evaluation of the two sincos forms on n arguments:
 * which = 0 the library's fast path (full polynomials), 1 __ocml_sincos_f32, 2 the fp64
 * slow path, 3 the fast path with the low-degree polynomials (valid below 512), 4 the b16 arithmetic
 * form (dcs_sincos_half2: d_sin receives the packed {cos, sin} half2 words, d_cos is not written). */
int dcs_probe_sincos(int which, const float *d_x, size_t n, float *d_sin, float *d_cos, void *stream);
/* Pure store kernel with the generator's access pattern and no arithmetic: the
 * measured HBM-write ceiling the roofline fraction is read against. */
int dcs_probe_fill(void *d_out, size_t bytes, int nontemporal, void *stream);
/* Store-only kernel over a `rows` x `cols_kib` KiB matrix: each workgroup owns a
 * rectangle of rb rows x qb KiB (tools/measure.py stores --kind pattern maps out which write
 * patterns the HBM system sustains; profiles/r01_store_patterns.md).  xcd_remap is a
 * bit set: 1 = workgroups sharing blockIdx % 8 take consecutive rectangles, 2 = a wave
 * takes consecutive 1-KiB chunks instead of every n-th, bits 4-6 = rotate the rectangle
 * column within groups of 8 (XCD <-> address affinity probe).  `nontemporal` selects the
 * store's cache policy: 0 plain, 1 nt, 2 sc1 (write-through), 3 sc0 sc1, 4 sc1 nt. */
int dcs_probe_store_pattern(void *d_out, uint32_t rows, uint32_t cols_kib, uint32_t qb, uint32_t rb, int order,
                            int xcd_remap, int nontemporal, uint32_t block_threads, void *stream);

/* The leanest store kernels (no loop, no integer division): stores_per_thread = 1 writes
 * the buffer linearly, one 16-byte store per thread; 2..4 writes it as rows of row_bytes with
 * the generator's pattern (a 4-wave workgroup = one 1-KiB tile x 4*stores_per_thread rows).
 * store_mode 0 plain, 1 nontemporal. */
int dcs_probe_one_store(void *d_out, size_t bytes, int store_mode, int stores_per_thread, uint32_t row_bytes, void *stream);

/* Whole-tensor properties of an fp32 coefficient tensor resident on the device:
 * checksum = sum of its 32-bit words mod 2^64 (order independent), and
 * max | re^2 + im^2 - 1 | (inf if any NaN).  Synchronises `stream`. */
int dcs_probe_reduce(const void *d_in, size_t bytes, uint64_t *checksum, float *max_modulus_dev, void *stream);

/* fp32 matrix-core issue-rate probe: `blocks` workgroups of 4 waves, each wave `iters` x 16 MFMAs back to back on
 * register operands.  which = 0: v_mfma_f32_16x16x4_f32, 2 accumulators; 1: 4 accumulators; 2: v_mfma_f32_32x32x2_f32,
 * 2 accumulators; 3: 16x16x4 with an int8 -> fp32 conversion in front of every pair; 4: the beamformer's k-step (A operands from LDS,
 * B operands converted).
 * FLOP per launch = blocks * 4 * iters * 16 * (2048 or 4096). */
int dcs_probe_mfma(int which, uint32_t blocks, uint32_t iters, float *d_out, void *stream);

/* The coefficient-reuse beamformer's XCD-aware workgroup numbering (bf_kernels.h: bf_xcd_grouped), evaluated on the HOST:
 * dispatch number w of a grid of `total` workgroups -> logical workgroup number, `group` consecutive logical workgroups per XCD.
 * tests/test_host_abi.py checks that it is a bijection and that a group's members share w % 8. */
uint32_t dcs_probe_xcd_grouped(uint32_t w, uint32_t total, uint32_t group);

/* Device-to-device copy by the leanest kernel (one 16-byte load and store per thread, `per_thread` of them `stride_kib`
 * KiB apart when > 1; workgroups in address order): the mixed read + write rate the HBM system sustains, which the
 * coefficient-reuse beamformer at 16 beams (as many bytes in as out) is read against.  store_mode 0 plain, 1 nontemporal. */
int dcs_probe_copy(const void *d_in, void *d_out, size_t bytes, int store_mode, int per_thread, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DCS_PROBES_H */
