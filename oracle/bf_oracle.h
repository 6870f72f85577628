/*
 * bf_oracle.h -- CPU restatement of the dc_sand beamformer steering-coefficient
 * verifier.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (dc_sand_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned".  The reference tree
 * (beamformer_coefficient_generator/) stores no golden vectors and cannot be
 * compiled in this image (it needs cuComplex.h / cuda_runtime_api.h), so the
 * restatement is pinned only procedurally: by the reference's own synthetic
 * input recipe (simulate_input) and by its acceptance rule
 * |kernel - verifier| <= 1e-4 (runBeamformerTests.cpp:30,61), which
 * tests/test_oracle.py re-runs between this file's verifier restatement and its
 * restatement of the reference's device arithmetic.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference root).
 */
#ifndef DCS_BF_ORACLE_H
#define DCS_BF_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* beamformer_coefficient_generator/BeamformerParameters.h:61-66 */
struct dcs_oracle_delay_vals {
    float fDelay_s;
    float fDelayRate_sps;
    float fPhase_rad;
    float fPhaseRate_radps;
};

/* Run-time stand-ins for the compile-time macros of
 * BeamformerParameters.h:7-17 (NR_CHANNELS, NR_STATIONS, NR_BEAMS,
 * SAMPLING_PERIOD, FFT_SIZE). */
struct dcs_oracle_params {
    int32_t nr_channels;
    int32_t nr_stations;
    int32_t nr_beams;
    float sampling_period; /* SAMPLING_PERIOD 1e-7f */
    int32_t fft_size;      /* FFT_SIZE 8192 */
};

/* Defaults of BeamformerParameters.h (64 chan, 64 ant, 16 beams). */
void dcs_oracle_default_params(struct dcs_oracle_params *p);

/* BeamformerCoefficientTest.cu:12-18 */
float dcs_oracle_ts_diff(struct timespec first, struct timespec last);

/* BeamformerCoefficientTest.cu:299 -- verifier's time step in ns. */
long dcs_oracle_time_step_ns(const struct dcs_oracle_params *p, size_t t);

/* BeamformerCoefficientTest.cu:235,247 -- the launch loop's time step in ns
 * (double 1e9); differs from the verifier's at some t.  Reported only. */
long dcs_oracle_time_step_ns_launch_loop(const struct dcs_oracle_params *p, size_t t);

/* BeamformerCoefficientTest.cu:296-300,320 -- fDeltaTime for time index t,
 * with the reference time given explicitly (it cancels). */
float dcs_oracle_delta_time(const struct dcs_oracle_params *p, size_t t,
                            struct timespec ref);

/* BeamformerCoefficientTest.cu:185-196 -- linear-ramp synthetic input for
 * n = nr_stations*nr_beams entries. */
void dcs_oracle_simulate_input(const struct dcs_oracle_params *p,
                               struct dcs_oracle_delay_vals *out);

/* BeamformerCoefficientTest.cu:319-328 -- one coefficient. */
/* 0 (default): (float)cos((double)rot); 1: cosf(rot) of the host libm -- see bf_oracle.c.
 * Process-wide; set it before calling the generate / beamform functions. */
void dcs_oracle_set_trig_reading(int reading);
int dcs_oracle_get_trig_reading(void);

void dcs_oracle_coeff(const struct dcs_oracle_params *p,
                      struct dcs_oracle_delay_vals d, float fDeltaTime,
                      size_t c, float *re, float *im);
/* Same, returning only fRotation (:326). */
float dcs_oracle_rotation(const struct dcs_oracle_params *p,
                          struct dcs_oracle_delay_vals d, float fDeltaTime,
                          size_t c);

/* BeamformerCoefficientTest.cu:294-337 -- expected coefficients for time
 * indices [t0, t0+nt) and channels [c0, c0+nc), written as
 * out[((t-t0)*nc + (c-c0))*A*B + a*B + b][2].  With c0=0, nc=nr_channels this
 * is the reference tensor [t][c][a][b][2] (:331-333).  Returns seconds spent
 * (steady clock, as :289,338). */
double dcs_oracle_generate(const struct dcs_oracle_params *p,
                           const struct dcs_oracle_delay_vals *delays,
                           size_t t0, size_t nt, size_t c0, size_t nc,
                           float *out);

/* BeamformerCoefficientTest.cu:308-333 with fDeltaTime (:320) of each of the nt time steps given
 * by the caller: out[(t*nc + (c-c0))*A*B + a*B + b][2]. */
void dcs_oracle_generate_dt(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays,
                            const float *dt, size_t nt, size_t c0, size_t nc,
                            float *out);
/* ... with fDeltaTime[t] = ts_diff(ref, cur[t]) (:320): the reference kernels' own time arguments
 * (BeamformerKernels.cuh:38-42, 81-86). */
void dcs_oracle_generate_at(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays,
                            const struct timespec *cur, struct timespec ref,
                            size_t nt, size_t c0, size_t nc, float *out);

/* Same loop, no output tensor: returns an order-independent checksum (sum of
 * the fp32 bit patterns, mod 2^64) over the same elements and the seconds
 * spent.  nthreads > 1 splits the channel range over pthreads (the reference
 * is serial; nthreads == 1 is the faithful baseline). */
double dcs_oracle_generate_checksum(const struct dcs_oracle_params *p,
                                    const struct dcs_oracle_delay_vals *delays,
                                    size_t t0, size_t nt, size_t c0, size_t nc,
                                    int nthreads, uint64_t *checksum);

/* BeamformerCoefficientTest.cu:348-359 -- first index where
 * |got - expect| > tol, or -1 (=> m_iResult 1). */
int64_t dcs_oracle_compare(const float *got, const float *expect, size_t n,
                           float tol);

/* Added: ULP distance (ordered-integer difference of the bit patterns);
 * max over n elements, count of elements with distance > limit. */
uint32_t dcs_oracle_ulp_diff(float a, float b);
uint32_t dcs_oracle_max_ulp(const float *got, const float *expect, size_t n,
                            uint32_t limit, uint64_t *n_over, int64_t *first_over);

/* verify_output() without the expected tensor (BeamformerCoefficientTest.cu:294-337 fused with
 * :348-357): every element of got[nt][nc][A*B][2] (channels [c0, c0+nc), fDeltaTime per time step
 * in dt[]) is compared with the coefficient generated on the fly; the channel range is split over
 * nthreads.  reading: 0 = (float)cos((double)x), 1 = cosf(x) (see dcs_oracle_set_trig_reading; the
 * process-wide switch is not touched).  hist[d] = number of fp32 elements at ULP distance d
 * (d = 0, 1, 2; hist[3]: more), *max_ulp, *first_over_1ulp = first flat index with d > 1 or -1.
 * Returns the seconds spent.  This is what lets the full-size configs be compared in EVERY element,
 * as the reference's verifier does. */
double dcs_oracle_compare_generated(const struct dcs_oracle_params *p,
                                    const struct dcs_oracle_delay_vals *delays,
                                    const float *dt, size_t nt, size_t c0, size_t nc,
                                    const float *got, int nthreads, int reading,
                                    uint64_t hist[4], uint32_t *max_ulp, int64_t *first_over_1ulp);
/* The same for the packed binary16 output ([..][A*B] half2 = {re, im}): expectation RN-even(verifier's fp32)
 * (dcs_oracle_f32_to_f16_rn), distances in binary16 ulps.  The reference emits this mode
 * (BeamformerKernels.cu:113-115, 182-184) and never checks it (BeamformerCoefficientTest.cu:282-287). */
double dcs_oracle_compare_generated_f16(const struct dcs_oracle_params *p,
                                        const struct dcs_oracle_delay_vals *delays,
                                        const float *dt, size_t nt, size_t c0, size_t nc,
                                        const uint16_t *got, int nthreads, int reading,
                                        uint64_t hist[4], uint32_t *max_ulp, int64_t *first_over_1ulp);

/* Restatement of the reference's DEVICE arithmetic, kernel a3
 * (BeamformerKernels.cu:153-177): dt = t*Ts*FFT in fp32, integer
 * NR_CHANNELS/2, all-fp32 fDelayN2, and a sincosf stand-in
 * ((float)cos/sin of the double).  Used only to re-run the reference's own
 * acceptance rule against the verifier restatement. */
void dcs_oracle_device_variant_a3(const struct dcs_oracle_params *p,
                                  const struct dcs_oracle_delay_vals *delays,
                                  size_t t0, size_t nt, float *out);

/* ---- fused coefficient generation + beamforming -------------
 * BeamformerCoefficientTest.cu:198-204 -- synthetic antenna data: byte i = (int8_t)i.
 * Layout [chan][time/16][station][16][2] (BeamformerKernels.cuh:137-143). */
void dcs_oracle_simulate_antenna_data(int8_t *out, size_t nbytes);

/* BeamformerCoefficientTest.cu:294-337 with the fused kernel's table ordering
 * iAntBeamOrdering = b*NR_STATIONS + a (:311), then :363-414: for every
 * (chan, time/16, beam, time%16) the fp32 sums over antennas, in antenna
 * order, of coeff.re * sample.re and coeff.im * sample.im (an element-wise
 * product, NOT a complex one -- :391-392).  delays is indexed [b*A + a].
 * out: float [chan][nt/16][beam][16][2]; nt must be a multiple of 16. */
void dcs_oracle_beamform(const struct dcs_oracle_params *p,
                         const struct dcs_oracle_delay_vals *delays, size_t nt,
                         const int8_t *antenna_data, float *out);

void dcs_oracle_beamform_dt(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays, const float *dt, size_t nt,
                            const int8_t *antenna_data, float *out);

/* Channels [c0, c0 + nc) of dcs_oracle_beamform_dt (dt == NULL: time indices 0 .. nt-1); tensors point at the slab. */
void dcs_oracle_beamform_slab(const struct dcs_oracle_params *p,
                              const struct dcs_oracle_delay_vals *delays, const float *dt, size_t nt, size_t c0, size_t nc,
                              const int8_t *antenna_data, float *out);

/* The verifier's beamformer (:363-414) with the coefficient HELD at one fDeltaTime for all nt samples
 * (what ACCUMULATIONS_BEFORE_NEW_COEFFS models; the reference has no kernel for it): the expectation of
 * dcs_bf_beamform_accumulated.  Same tensors and table ordering as dcs_oracle_beamform. */
void dcs_oracle_beamform_accumulated(const struct dcs_oracle_params *p,
                                     const struct dcs_oracle_delay_vals *delays, float dt_coeff, size_t nt,
                                     const int8_t *antenna_data, float *out);
/* Channels [c0, c0 + nc) of it; antenna_data / out point at the slab (first channel c0). */
void dcs_oracle_beamform_accumulated_slab(const struct dcs_oracle_params *p,
                                          const struct dcs_oracle_delay_vals *delays, float dt_coeff, size_t nt,
                                          size_t c0, size_t nc, const int8_t *antenna_data, float *out);

/* fp16 (f2): IEEE binary16 round-to-nearest-even of an fp32, as
 * __floats2half2_rn does per element (BeamformerKernels.cu:113,182). */
uint16_t dcs_oracle_f32_to_f16_rn(float x);

#ifdef __cplusplus
}
#endif
#endif
