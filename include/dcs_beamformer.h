/*
 * dcs_beamformer.h -- C-ABI of the MI355X (gfx950) steering-coefficient
 * generator: the drop-in boundary for the one hot path of ska-sa/dc_sand's
 * beamformer_coefficient_generator/.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / scalars,
 * returns an int status (0 = DCS_OK, > 0 = a hipError_t value, < 0 = one of the
 * DCS_ERR_* codes below) and never exits, throws or prints.  All device work is
 * enqueued on the caller's hipStream_t (passed as void*, NULL = the null
 * stream) so the calls can be captured in a hipGraph (dcs_bf_generate /
 * dcs_bf_generate_slab in the default tiled form and the beamformers, for up to
 * 256 time steps per call -- their fDeltaTime values travel in the kernel
 * arguments; the beamformers after a first plain call, which allocates their
 * terms table -- ; longer calls and the rows form stage tables through pinned
 * memory and are not capturable).  A call that cannot be captured says so UP
 * FRONT: on a capturing stream it returns DCS_ERR_UNSUPPORTED before it has
 * enqueued anything, and the capture stays valid (it can be ended, or carried
 * on with capturable calls).  Nothing here starts host threads.
 *
 * Each declaration cites the reference interface it replaces (paths relative
 * to the reference root; "BCT" = beamformer_coefficient_generator/
 * BeamformerCoefficientTest).  INTEGRATION.md shows the reference-side stub.
 *
 * Library: dc_sand_amd/csrc/libdcs_beamformer.so (built by `python -m
 * dc_sand_amd.build` or __graft_entry__.build()).
 */
#ifndef DCS_BEAMFORMER_H
#define DCS_BEAMFORMER_H

#include <stddef.h>
#include <stdint.h>
#include <time.h> /* struct timespec: the reference kernels' time arguments */

#ifdef __cplusplus
extern "C" {
#endif
/* the libraries are built -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

#define DCS_BF_ABI_VERSION 3 /* 3: dcs_bf_tuning without measurement fields; stream ticks from a device-resident table */

/* ---- status codes ------------------------------------------------------- */
#define DCS_OK 0
#define DCS_ERR_INVALID_ARGUMENT (-1)
#define DCS_ERR_UNSUPPORTED (-2)   /* the reference `throw`s here: BCT.cu:40-50 */
#define DCS_ERR_NOT_READY (-3)     /* e.g. generate before a delay table is set */
#define DCS_ERR_OUT_OF_RANGE (-4)
#define DCS_ERR_NO_DEVICE (-5)
#define DCS_ERR_WRONG_DEVICE (-6) /* the context lives on another device than the calling thread's current one */

/* "<HIP Error|dcs error>: <text>", cf. common/Utils.cpp:8-16 (gpu_assert prints
 * and exit()s; here the caller decides). Returns a static string. */
const char *dcs_error_string(int status);
int dcs_abi_version(void);

/* ---- data contract ------------------------------------------------------ */

/* BeamformerParameters.h:61-66 -- 4 x fp32, 16 bytes, no padding; the table is
 * indexed [antenna * nr_beams + beam] (BCT.cu:315; BeamformerKernels.cu:22,79). */
#ifndef DCS_DELAY_VALS_DEFINED
#define DCS_DELAY_VALS_DEFINED
struct dcs_delay_vals {
    float fDelay_s;
    float fDelayRate_sps;
    float fPhase_rad;
    float fPhaseRate_radps;
};
#endif

/* BCT.hpp:19-25 (enum SteeringCoefficientKernel) -- same order, same values. */
enum dcs_bf_kernel {
    DCS_BF_NAIVE = 0,
    DCS_BF_MULTIPLE_CHANNELS = 1,
    DCS_BF_MULTIPLE_CHANNELS_AND_TIMESTAMPS = 2,
    DCS_BF_COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL = 3 /* has its own entry point: dcs_bf_generate_and_beamform */
};

/* BCT.hpp:33-37 (enum SteeringCoefficientBitWidth). */
enum dcs_bf_bitwidth { DCS_BF_B16 = 0, DCS_BF_B32 = 1 };

/* Run-time stand-ins for the compile-time macros of BeamformerParameters.h:4-17.
 * dcs_bf_default_params() fills the header's values (64 chan, 64 ant, 16 beams,
 * 256 samples, 1e-7f, 8192, 1712e6, 256). */
struct dcs_bf_params {
    int32_t nr_channels;                     /* NR_CHANNELS */
    int32_t nr_stations;                     /* NR_STATIONS */
    int32_t nr_beams;                        /* NR_BEAMS (beams held by THIS context) */
    int32_t nr_samples_per_channel;          /* NR_SAMPLES_PER_CHANNEL */
    float sampling_period;                   /* SAMPLING_PERIOD */
    int32_t fft_size;                        /* FFT_SIZE */
    double adc_sample_rate;                  /* ADC_SAMPLE_RATE */
    int32_t accumulations_before_new_coeffs; /* ACCUMULATIONS_BEFORE_NEW_COEFFS */
    int32_t reserved;
};
int dcs_bf_default_params(struct dcs_bf_params *p);

/* Output layout (BCT.cu:331-333; BeamformerKernels.cu:47-49,175-177,181-184):
 *   b32: float  [nt][nr_channels][nr_stations][nr_beams][2]   (re, im)
 *   b16: __half2[nt][nr_channels][nr_stations][nr_beams]      (.x = re, .y = im)
 * Size as BCT.cu:32-38. */
int dcs_bf_output_bytes(const struct dcs_bf_params *p, int bitwidth, uint32_t nt, size_t *bytes);

/* fDeltaTime of the verifier for time indices [t0, t0+nt): BCT.cu:299 (fp32
 * time step in ns, truncated) then ts_diff, BCT.cu:12-18.  Host only.  This is
 * the dt every kernel variant uses. */
int dcs_bf_delta_times(const struct dcs_bf_params *p, uint64_t t0, uint32_t nt, float *dt_out);

/* ts_diff(), BCT.cu:12-18, operation by operation:
 *   (float)last.tv_sec - (float)first.tv_sec + (float)(last.tv_nsec - first.tv_nsec) / 1e9f
 * -- including what that text does not do: no nanosecond carry (the TODOs at BCT.cu:232,244,296 and
 * BeamformerKernels.cu:24,82; harmless, the difference is signed) and seconds converted to fp32 BEFORE
 * the subtraction, so epoch-sized tv_sec (> 2^24) lose their low bits -- use a monotonic clock or a
 * recent reference, as the reference does (CLOCK_MONOTONIC, BCT.cu:59-69).  The reference's kernels
 * subtract the integer seconds first (BeamformerKernels.cu:25-27, 83-85); the two agree whenever both
 * tv_sec < 2^24.  Host only. */
int dcs_bf_ts_diff(const struct timespec *first, const struct timespec *last, float *dt_out);

/* simulate_input(), BCT.cu:185-196: the reference's linear-ramp table for
 * n = nr_stations*nr_beams entries.  Host only. */
int dcs_bf_simulate_input(const struct dcs_bf_params *p, struct dcs_delay_vals *table_out);

/* ---- device plumbing (the pycuda calls of pycuda_example/vector_add.py:14-46) */
int dcs_device_count(int *count);
int dcs_device_set(int device);
int dcs_device_synchronize(void);
int dcs_device_name(int device, char *buf, size_t buflen);
int dcs_malloc(void **dptr, size_t bytes);                 /* cuda.mem_alloc / cudaMalloc BCT.cu:74,78 */
int dcs_free(void *dptr);                                  /* cudaFree BCT.cu:95,99 */
int dcs_host_alloc(void **hptr, size_t bytes);             /* cuda.pagelocked_empty / cudaMallocHost BCT.cu:73,77 */
int dcs_host_free(void *hptr);                             /* cudaFreeHost BCT.cu:96,100 */
int dcs_memcpy_htod(void *dptr, const void *hptr, size_t bytes, void *stream); /* BCT.cu:210 */
int dcs_memcpy_dtoh(void *hptr, const void *dptr, size_t bytes, void *stream); /* BCT.cu:270 */
int dcs_memcpy_dtod(void *dst, const void *src, size_t bytes, void *stream);
/* Strided device->host gather of rows: nrows rows of row_bytes, src pitch src_pitch. */
int dcs_memcpy2d_dtoh(void *hptr, size_t dst_pitch, const void *dptr, size_t src_pitch,
                      size_t row_bytes, size_t nrows, void *stream);
int dcs_memset(void *dptr, int value, size_t bytes, void *stream);
int dcs_stream_create(void **stream);
int dcs_stream_destroy(void *stream);
int dcs_stream_synchronize(void *stream);

/* Event timing of UnitTest::run_test (common/UnitTest.cpp:9-14,34-53). */
int dcs_event_create(void **event);
int dcs_event_destroy(void *event);
int dcs_event_record(void *event, void *stream);
int dcs_event_synchronize(void *event);
int dcs_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- the hot path ------------------------------------------------------- */
typedef struct dcs_bf_context dcs_bf_context;

/* Owns the device delay table (double-buffered) and the per-launch dt slots for
 * the given shape on the current device; replaces the buffer ownership of
 * BeamformerCoeffTest's ctor/dtor (BCT.cu:73-87,93-109).  Output buffers belong
 * to the caller.  Shape guards return DCS_ERR_INVALID_ARGUMENT.
 *
 * STREAM RULE: a context is used from ONE stream at a time.  Its device-side scratch (the two
 * delay-table buffers, the fDeltaTime staging slots of launches longer than 256 time steps, the
 * terms table of the rows form and of the fused kernel) is reused by later calls without
 * cross-stream events, exactly as BeamformerCoeffTest's buffers are used from the null stream
 * only.  To move a context to another stream, synchronise the old stream first; concurrent
 * streams need one context each (a context is a few MiB). */
int dcs_bf_create(const struct dcs_bf_params *p, dcs_bf_context **ctx);
int dcs_bf_destroy(dcs_bf_context *ctx);

/* transfer_HtoD(), BCT.cu:207-216: copy a host table of nr_stations*nr_beams
 * entries into the context (async on `stream`; `table` must stay valid until
 * the stream reaches the copy -- use pinned memory for true overlap). */
int dcs_bf_upload_delays(dcs_bf_context *ctx, const struct dcs_delay_vals *table, void *stream);

/* Multi-GPU entry: take this context's beam slice out of a device-resident
 * GLOBAL table [nr_stations][nr_beams_total] (e.g. just broadcast by RCCL):
 * local beam b is global beam beam_offset + b.  Device-side gather on `stream`. */
int dcs_bf_set_delays_from_global(dcs_bf_context *ctx, const void *d_global_table,
                                  uint32_t nr_beams_total, uint32_t beam_offset, void *stream);

/* run_kernel(), BCT.cu:218-264.  Writes time indices [t0, t0+nt) into d_out
 * (layout above, time index t0 first).  `kernel` keeps the reference's launch
 * shapes: NAIVE and MULTIPLE_CHANNELS issue one launch per time step from a
 * host loop (BCT.cu:230-250), MULTIPLE_CHANNELS_AND_TIMESTAMPS one launch for
 * all of them (BCT.cu:253-257).  NAIVE + B16 returns DCS_ERR_UNSUPPORTED (BCT.cu:40-44), and so does
 * COMBINED here (it needs antenna data: dcs_bf_generate_and_beamform).
 * DCS_ERR_UNSUPPORTED (BCT.cu:40-50).  All variants compute the verifier's
 * arithmetic (BCT.cu:319-328) and agree bit for bit with each other. */
int dcs_bf_generate(dcs_bf_context *ctx, int kernel, int bitwidth, uint64_t t0, uint32_t nt,
                    void *d_out, size_t out_bytes, void *stream);

/* As dcs_bf_generate(MULTIPLE_CHANNELS_AND_TIMESTAMPS) for the channel slab
 * [c0, c0+nc) only; d_out is the slab tensor [nt][nc][stations][beams]. */
int dcs_bf_generate_slab(dcs_bf_context *ctx, int bitwidth, uint64_t t0, uint32_t nt,
                         uint32_t c0, uint32_t nc, void *d_out, size_t out_bytes, void *stream);

/* The same launches with the TIME given by the caller instead of derived from a time index --
 * what the reference's kernels take (struct timespec sCurrentTime, sRefTime by value:
 * BeamformerKernels.cuh:38-42, 81-86; arithmetic BeamformerKernels.cu:25-27, 83-85; the verifier's
 * ts_diff BCT.cu:12-18, :320).  dcs_bf_generate(t0, nt) is the special case
 * cur[i] = ref + (long)((t0+i)*SAMPLING_PERIOD*1e9f*FFT_SIZE) ns (BCT.cu:296-300).
 *   _dt: fDeltaTime of each of the nt time steps by value (host array, consumed before return);
 *   _at: cur[i] (nt entries) and one reference time; fDeltaTime[i] = dcs_bf_ts_diff(ref, &cur[i]).
 * Output layout, kernel / bitwidth rules and capturability as dcs_bf_generate. */
int dcs_bf_generate_dt(dcs_bf_context *ctx, int kernel, int bitwidth, const float *dt, uint32_t nt,
                       void *d_out, size_t out_bytes, void *stream);
int dcs_bf_generate_slab_dt(dcs_bf_context *ctx, int bitwidth, const float *dt, uint32_t nt, uint32_t c0,
                            uint32_t nc, void *d_out, size_t out_bytes, void *stream);
int dcs_bf_generate_at(dcs_bf_context *ctx, int kernel, int bitwidth, const struct timespec *cur,
                       const struct timespec *ref, uint32_t nt, void *d_out, size_t out_bytes, void *stream);

/* run_kernel(), COMBINED_COEFF_GEN_AND_BEAMFORMER_SINGLE_CHANNEL branch (BCT.cu:259-262;
 * kernel BeamformerKernels.cu:192-367, doc BeamformerKernels.cuh:95-162; verifier
 * BCT.cu:363-414) -- SURVEY.md section 8 f1, generalised from the reference's
 * hard-wired 64 antennas x 16 beams to any shape.  For time indices
 * [t0, t0+nt) (both multiples of 16 = INTERNAL_TIME_SAMPLES):
 *   d_antenna: int8  [nr_channels][nt/16][nr_stations][16][2]   (BeamformerKernels.cuh:137-140)
 *   d_beams  : float [nr_channels][nt/16][nr_beams][16][2]      (BeamformerKernels.cuh:141-143)
 *   beams = sum over antennas, in antenna order, of (coeff.re*sample.re, coeff.im*sample.im)
 * -- the reference's element-wise product (BeamformerKernels.cu:315-316), not a
 * complex one.  For this entry point the delay table is indexed
 * [beam * nr_stations + antenna] (BCT.cu:311; BeamformerKernels.cu:252).
 * No coefficient tensor is materialised.  Two launches (terms pre-pass, beamformer) per 256 time steps. */
int dcs_bf_generate_and_beamform(dcs_bf_context *ctx, uint64_t t0, uint32_t nt, const int8_t *d_antenna,
                                 size_t antenna_bytes, float *d_beams, size_t beams_bytes, void *stream);

/* The same with fDeltaTime of each of the nt samples by value (nt a multiple of 16). */
int dcs_bf_generate_and_beamform_dt(dcs_bf_context *ctx, const float *dt, uint32_t nt, const int8_t *d_antenna,
                                    size_t antenna_bytes, float *d_beams, size_t beams_bytes, void *stream);

/* Beamformer with coefficient REUSE -- the "general version" of the fused kernel.  The reference
 * regenerates every coefficient for every sample and only MODELS what a deployed beamformer does: new
 * coefficients every ACCUMULATIONS_BEFORE_NEW_COEFFS time units (BeamformerParameters.h:17; the utilisation
 * model BCT.cu:426-448).  Here the coefficients of ONE time -- time index t_coeff (fDeltaTime as
 * dcs_bf_delta_times gives it), or fDeltaTime by value -- are generated once per (channel, antenna, beam), into
 * registers, never HBM, and applied to nt samples (a multiple of 16; tensors and table ordering exactly as
 * dcs_bf_generate_and_beamform):
 *   beams[c][t/16][b][t%16] = ( sum_a cos(rot[a][b][c]) * re[c][t][a] , sum_a sin(rot[a][b][c]) * im[c][t][a] )
 * Per channel two real contractions over the antennas, on the matrix cores.  Default form: EXACT integer
 * arithmetic on the int8 pipe (v_mfma_i32_16x16x64_i8).  The samples are int8; each coefficient w (|w| <= 1) is
 * taken as the 24-bit fixed-point number rint(w * 8355711) = three signed 8-bit digits, so a plane is three
 * integer contractions whose sums are exact and independent of the antenna order, recombined and divided by
 * 8355711 in fp32 at the end.  Against the exact sum of the fp32 coefficients times the samples the result is
 * within 9e-8 * sum_a |sample_a| + 1.5 ulp -- closer than the verifier's own fp32 loop (BCT.cu:363-414 with the
 * coefficient held: sum += coeff * sample, whose partial sums round at every antenna) -- and within
 * 4e-5 * nr_stations of that loop in the worst case (every sample at full scale and every error aligned: 1 ulp of
 * the coefficient + the quantisation + the roundings of either side, 3e-7 * 128 per antenna; typically 10 x less),
 * against the reference's tolerance of 1e-1 (runBeamformerTests.cpp:15).  A coefficient that is not finite (an infinite or
 * NaN delay value) makes every sample of its beam NaN in the plane concerned, as in the verifier's sum (NaN * 0 = NaN).
 * dcs_bf_tuning.math_mode bit 3 selects the other form: v_mfma_f32_16x16x4_f32, exact fp32 products
 * accumulated as an fma chain in antenna order (differs from the verifier's loop by the chain's single
 * roundings only; same bound; 1/32 of the int8 pipe's rate).  nr_stations <= 256; d_antenna 16-byte aligned.
 * Two launches (terms pre-pass, contraction); capturable once the context's terms table exists (allocated by the
 * first call of this or of dcs_bf_generate_and_beamform: make one call outside the capture). */
int dcs_bf_beamform_accumulated(dcs_bf_context *ctx, uint64_t t_coeff, uint32_t nt, const int8_t *d_antenna,
                                size_t antenna_bytes, float *d_beams, size_t beams_bytes, void *stream);
int dcs_bf_beamform_accumulated_dt(dcs_bf_context *ctx, float dt_coeff, uint32_t nt, const int8_t *d_antenna,
                                   size_t antenna_bytes, float *d_beams, size_t beams_bytes, void *stream);

/* Launch-geometry knobs (all 0 / NULL = the library's shape-aware defaults, DESIGN.md "launch
 * geometry": per launch the library looks at the tiles per row, the workgroups the launch makes and
 * its bytes -- fp32: 1 tile x 12 channels per workgroup and at most 6 workgroups per CU when there is
 * plenty of work (no limit for launches of 256 MiB - 2 GiB), 10 channels without limit for rows of >= 2048 tiles,
 * 2 tiles x 16 channels for launch-bound tensors of <= 32 MiB; launches of >= 2 GiB (terms table): fp32 8 channels x 6 workgroups
 * per CU, fp16 48 x 5 (20 x 6 with the b16 arithmetic form); smaller fp16 launches 128 channels, fewer while the chip would be left
 * under 2048 workgroups; no residency limit when every workgroup is resident at once).  Two forms of the
 * MULTIPLE_CHANNELS_AND_TIMESTAMPS generator exist and give identical bits:
 *   "tiled": a workgroup (4 waves) keeps its pairs' terms in registers and walks chan_per_block
 *           channels, tiles_per_block 1-KiB tiles wide.  The terms (one fp64 chain per pair and time
 *           step) are either computed by every workgroup and staged in LDS (form 1), or read from a
 *           small table a pre-pass kernel writes just before (form 3; at most 8 time steps per launch,
 *           nontemporal stores): the pre-pass costs ~3 us and takes ~12 of 41 VALU operations per fp32
 *           coefficient out of the main kernel.  form 0 (default) picks by size: the table for launches
 *           of >= 2 GiB of output, per-workgroup terms for smaller ones;
 *   form 2 "rows":  the terms table for any number of time steps, then short waves
 *           (waves_per_block adjacent 1-KiB tiles x rows_per_wave channel rows)
 *           stream the tensor in address order.
 * MULTIPLE_CHANNELS always uses the tiled form (the reference's per-time-step shape); from 8 time steps on its
 * launches are spread over four internal streams between a fork and a join on the caller's stream. */
struct dcs_bf_tuning { /* ABI 3: ten int32_t; the measurement knobs of ABI 2 live in include/dcs_probes.h now */
    int32_t form;            /* 0 default (tiled; terms table for large launches), 1 tiled with per-workgroup terms,
                              * 2 rows, 3 tiled with the terms table */
    int32_t nontemporal;     /* -1 default, 0 plain stores, 1 nontemporal */
    int32_t chan_per_block;  /* form 1 */
    int32_t tiles_per_block; /* form 1: 1, 2, 4 */
    int32_t waves_per_block; /* form 2: 4, 8, 16 */
    int32_t rows_per_wave;   /* form 2: 1..4 */
    int32_t xcd_remap;       /* -1 default, 0, 1: workgroups sharing blockIdx % 8 (one XCD) take consecutive work */
    int32_t rows_same_tile;  /* form 2: -1 default, 0 = the waves take adjacent tiles, 1 = they share one tile and
                              * interleave rows */
    int32_t math_mode;       /* arithmetic forms.  Bits 0 and 1 are A/B switches that never change a bit of the output:
                              * bit 0 = keep the 5-op divide even where the 3-op form was verified exact for this
                              * divisor; bit 1 = keep the full-degree polynomials even where the low-degree ones are
                              * proven.  Bit 2 (value 4) OPTS IN to the b16 arithmetic form: where the output is b16
                              * and no pair of a wave needs the slow path (|fRotation| < 32000), sin and cos are evaluated to binary16
                              * accuracy (two-term reduction, degree 5 / 4) and converted once, instead of rounding the
                              * 1-ULP fp32 pair: 21 instead of 28 VALU operations per coefficient.  Every half is within
                              * one binary16 ulp of RN16(correctly rounded value) for EVERY fp32 argument below 32768, and
                              * equal to it for 99.8 % of them (0.9 % differ in [1, 32768); tests/test_numerics.py, by
                              * exhaustion).  The reference rounds whatever __sincosf returned and never checks it
                              * (BeamformerKernels.cu:113-115,182-184; BCT.cu:282-287).  Default off; fp32 output and
                              * the rows form are unaffected.  Bit 3 (value 8): dcs_bf_beamform_accumulated runs its
                              * fp32 fma-chain form instead of the exact fixed-point one. */
    int32_t wg_per_cu;       /* form 1: 0 = default (fp32: 6 for launches that oversubscribe the chip), -1 = no limit, 2..7 = at most this many workgroups resident per CU (the launch
                              * asks for unused dynamic LDS to that end): fewer waves in flight keep the store stream
                              * closer to address order (profiles/r01_store_patterns.md, "Fewer workgroups in flight") */
};
int dcs_bf_set_tuning(dcs_bf_context *ctx, const struct dcs_bf_tuning *t);

/* Measure the tiled form's launch geometries on THIS device for THIS shape (and the delay table
 * currently set) and keep the fastest for large launches of this output width: the optimum is sharp
 * and moves with shape and arithmetic form (profiles/r01_geometry_sweep.md, profiles/r02_autotune.md).
 * Generates channels [0, min(nr_channels, out_bytes / row)) of time index 1 into d_out repeatedly
 * (~25 trial geometries x 2 rounds; each trial first settles ~20 ms on its own geometry, because the
 * first launches after a change of access pattern run slower, then times ~3 ms; the four best meet the
 * library's shape-aware default in a play-off, and a challenger replaces the default only when it is
 * more than 0.7 % faster: ~1 s in all); blocks on events, so it cannot be captured in a graph.  The
 * result is CACHED in the context per output width (b16: per arithmetic form too, math_mode bit 2 selects another kernel) -- a second call returns it at once;
 * dcs_bf_set_tuning(ctx, NULL) forgets it -- and *chosen (may be NULL) reports it.  Knobs set
 * explicitly with dcs_bf_set_tuning keep precedence over it.  Results do not depend on the geometry
 * (every one gives the same bits). */
int dcs_bf_autotune(dcs_bf_context *ctx, int bitwidth, void *d_out, size_t out_bytes, void *stream,
                    struct dcs_bf_tuning *chosen);

/* get_time(), BCT.cu:422-454: the real-time utilisation model, from a kernel
 * duration in ms.  out[0] = per single time unit, out[1] = per
 * ACCUMULATIONS_BEFORE_NEW_COEFFS time units (both x4 as BCT.cu:447-448). */
int dcs_bf_gpu_utilisation(const struct dcs_bf_params *p, float kernel_ms, float out[2]);

/* ---- streaming (BASELINE config 5) -------------------------------------- */
/* A hipGraph with one kernel node -- two for slabs of >= 2 GiB, whose pairs' terms come from a pre-pass
 * kernel (dcs_bf_tuning::form) -- (generate one time step of the channel slab
 * [c0, c0+nc), all (antenna, beam), in place in d_out).  Each tick rewrites the
 * node's arguments in the instantiated graph (fDeltaTime of time index t; the
 * delay-table buffer) and replays it on `stream` -- no host synchronisation.  A
 * non-NULL new_table is staged into the idle table buffer first and used from
 * this tick on (time-varying delay polynomials, double-buffered; the host copy goes
 * through a ring of four pinned buffers, so a tick blocks the host only when four
 * table updates are still in flight). */
typedef struct dcs_bf_stream dcs_bf_stream;
int dcs_bf_stream_begin(dcs_bf_context *ctx, int bitwidth, uint32_t c0, uint32_t nc, void *d_out,
                        size_t out_bytes, void *stream, dcs_bf_stream **s);
int dcs_bf_stream_tick(dcs_bf_stream *s, uint64_t t, const struct dcs_delay_vals *new_table);
/* A tick at an arbitrary model time: fDeltaTime by value, or (current, reference) as the reference's
 * kernels take them (BeamformerKernels.cuh:38-42).  BASELINE configs[4]'s "200 us cadence" is
 * tick_dt(k * 200e-6f): model time and wall time advance together. */
int dcs_bf_stream_tick_dt(dcs_bf_stream *s, float dt, const struct dcs_delay_vals *new_table);
int dcs_bf_stream_tick_at(dcs_bf_stream *s, const struct timespec *cur, const struct timespec *ref,
                          const struct dcs_delay_vals *new_table);
/* The same ticks with the new delay table ALREADY ON THE DEVICE -- the shape of the reference's kernels, whose
 * table is a device pointer argument (BeamformerKernels.cuh:38-42, 81-86), and what the multi-GPU design
 * produces: d_global_table is the GLOBAL table [nr_stations][nr_beams_total] an RCCL broadcast has just
 * landed (16-byte aligned), of which this context owns beams [beam_offset, beam_offset + nr_beams), as in
 * dcs_bf_set_delays_from_global (nr_beams_total = nr_beams, beam_offset = 0: a plain device table).  The
 * slice is gathered into the idle table buffer by a kernel node IN the replayed graph (its arguments are
 * rewritten per tick like the others): no host staging, no event, no synchronisation -- BASELINE
 * configs[3] (beam-sharded, table by broadcast) and configs[4] (streaming) compose without a D2H round
 * trip.  The table must stay unchanged until the stream has passed this tick. */
int dcs_bf_stream_tick_from_global(dcs_bf_stream *s, uint64_t t, const void *d_global_table,
                                   uint32_t nr_beams_total, uint32_t beam_offset);
int dcs_bf_stream_tick_dt_from_global(dcs_bf_stream *s, float dt, const void *d_global_table,
                                      uint32_t nr_beams_total, uint32_t beam_offset);
int dcs_bf_stream_tick_at_from_global(dcs_bf_stream *s, const struct timespec *cur, const struct timespec *ref,
                                      const void *d_global_table, uint32_t nr_beams_total, uint32_t beam_offset);
int dcs_bf_stream_end(dcs_bf_stream *s);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DCS_BEAMFORMER_H */
