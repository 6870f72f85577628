/*
 * bf_oracle.c -- CPU restatement of the dc_sand steering-coefficient verifier.
 * TEST INFRASTRUCTURE ONLY (see bf_oracle.h).  "parity unpinned": the reference
 * holds no golden vectors and is unbuildable here (no nvcc, no libcudart).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math): every
 * fp32 operation below must round once, in the order written, and no
 * multiply-add may be fused.
 *
 * Typing notes (x86-64, FLT_EVAL_METHOD == 0):
 *  - `float * size_t` and `float * int` convert the integer to float.
 *  - `cos(float)`: TWO readings, selected by dcs_oracle_set_trig_reading():
 *      0 (default, canonical for this repo's fixtures): `double cos(double)`, rounded to
 *        float on assignment -- what the text means under ISO C / plain g++ with <cmath>
 *        ("double-then-round"); libm-independent to within a
 *        double's accuracy.
 *      1: `cosf` / `sinf` of the host's float libm -- what the reference's own toolchain
 *        binds to: nvcc force-includes cuda_runtime.h in every .cu, whose
 *        crt/math_functions.h says `using std::cos; using std::sin;` at global scope for
 *        GCC hosts (CUDA 12.8 headers as shipped in this image under
 *        triton/backends/nvidia/include/crt/math_functions.h:4838-4862), so the
 *        unqualified `cos(fRotation)` of BeamformerCoefficientTest.cu:327 resolves to
 *        libstdc++'s `std::cos(float)` = `__builtin_cosf`.  Its value depends on the glibc
 *        the reference was linked against (unpinned by the reference; 2.35 here).
 *    The two differ by exactly 1 ULP in about 1.3 % of samples.
 */
#define _GNU_SOURCE
#include "bf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void dcs_oracle_default_params(struct dcs_oracle_params *p)
{
    /* BeamformerParameters.h:7-17 */
    p->nr_channels = 64;
    p->nr_stations = 64;
    p->nr_beams = 16;
    p->sampling_period = 1e-7f;
    p->fft_size = 8192;
}

/* BeamformerCoefficientTest.cu:12-18 */
float dcs_oracle_ts_diff(struct timespec first, struct timespec last)
{
    float time_difference = (float)last.tv_sec - (float)first.tv_sec;
    long nanosec_difference = last.tv_nsec - first.tv_nsec;
    time_difference += (float)nanosec_difference / 1e9f;
    return time_difference;
}

/* BeamformerCoefficientTest.cu:299
 *   long timeStep = t*SAMPLING_PERIOD*1e9f*FFT_SIZE;   (all fp32, then trunc) */
long dcs_oracle_time_step_ns(const struct dcs_oracle_params *p, size_t t)
{
    long timeStep = t * p->sampling_period * 1e9f * p->fft_size;
    return timeStep;
}

/* BeamformerCoefficientTest.cu:235,247
 *   long lTimeStep = ulTimeIndex*SAMPLING_PERIOD*1e9*FFT_SIZE; (1e9 is double) */
long dcs_oracle_time_step_ns_launch_loop(const struct dcs_oracle_params *p, size_t t)
{
    long lTimeStep = t * p->sampling_period * 1e9 * p->fft_size;
    return lTimeStep;
}

/* BeamformerCoefficientTest.cu:296-300 then :320 */
float dcs_oracle_delta_time(const struct dcs_oracle_params *p, size_t t,
                            struct timespec ref)
{
    struct timespec sCurrentTime_ns;
    sCurrentTime_ns.tv_sec = ref.tv_sec;
    long timeStep = dcs_oracle_time_step_ns(p, t);
    sCurrentTime_ns.tv_nsec = ref.tv_nsec + timeStep;
    return dcs_oracle_ts_diff(ref, sCurrentTime_ns);
}

/* BeamformerCoefficientTest.cu:185-196 */
void dcs_oracle_simulate_input(const struct dcs_oracle_params *p,
                               struct dcs_oracle_delay_vals *out)
{
    size_t ulNumDelayVelays = (size_t)p->nr_stations * (size_t)p->nr_beams;
    const float SAMPLING_PERIOD = p->sampling_period;
    for (size_t i = 0; i < ulNumDelayVelays; i++) {
        out[i].fDelay_s = ((float)i / ((float)ulNumDelayVelays)) * SAMPLING_PERIOD / 3.0;
        out[i].fDelayRate_sps = 2e-6;
        out[i].fPhase_rad = (1 - ((float)i / (float)ulNumDelayVelays)) * SAMPLING_PERIOD / 3.0;
        out[i].fPhaseRate_radps = 3e-6;
    }
}

/* BeamformerCoefficientTest.cu:321-326 */
float dcs_oracle_rotation(const struct dcs_oracle_params *p,
                          struct dcs_oracle_delay_vals sDelayVal,
                          float fDeltaTime, size_t c)
{
    const float SAMPLING_PERIOD = p->sampling_period;
    const int NR_CHANNELS = p->nr_channels;
    float fDeltaDelay = sDelayVal.fDelayRate_sps * fDeltaTime;
    float fDelayN = (sDelayVal.fDelayRate_sps + fDeltaDelay) * c * ((float)M_PI) / (SAMPLING_PERIOD * NR_CHANNELS);
    float fDelayN2 = (sDelayVal.fDelay_s + fDeltaDelay) * (NR_CHANNELS / 2.0) * ((float)M_PI) / (SAMPLING_PERIOD * NR_CHANNELS);
    float fDeltaPhase = sDelayVal.fPhaseRate_radps * fDeltaTime;
    float fPhase0 = sDelayVal.fPhase_rad - fDelayN2 + fDeltaPhase;
    float fRotation = fDelayN + fPhase0;
    return fRotation;
}

/* Which function the verifier's unqualified cos(float) / sin(float) binds to (header comment). */
static int g_trig_reading = 0;
void dcs_oracle_set_trig_reading(int reading) { g_trig_reading = reading == 1 ? 1 : 0; }
int dcs_oracle_get_trig_reading(void) { return g_trig_reading; }

/* BeamformerCoefficientTest.cu:319-328, with the reading of cos(float) passed in (the threaded
 * comparison below evaluates both readings without touching the process-wide switch). */
static void coeff_reading(const struct dcs_oracle_params *p,
                          struct dcs_oracle_delay_vals d, float fDeltaTime,
                          size_t c, int reading, float *re, float *im)
{
    float fRotation = dcs_oracle_rotation(p, d, fDeltaTime, c);
    float fSteeringCoeffCorrectReal, fSteeringCoeffCorrectImag;
    if (reading == 1) {
        fSteeringCoeffCorrectReal = cosf(fRotation);
        fSteeringCoeffCorrectImag = sinf(fRotation);
    } else {
        fSteeringCoeffCorrectReal = cos(fRotation);
        fSteeringCoeffCorrectImag = sin(fRotation);
    }
    *re = fSteeringCoeffCorrectReal;
    *im = fSteeringCoeffCorrectImag;
}

/* BeamformerCoefficientTest.cu:319-328 */
void dcs_oracle_coeff(const struct dcs_oracle_params *p,
                      struct dcs_oracle_delay_vals d, float fDeltaTime,
                      size_t c, float *re, float *im)
{
    coeff_reading(p, d, fDeltaTime, c, g_trig_reading, re, im);
}

static const struct timespec k_ref_zero = {0, 0};

/* BeamformerCoefficientTest.cu:294-337, ordering of :308-316 (non-fused
 * kernels: iAntBeamOrdering = a*NR_BEAMS + b). */
double dcs_oracle_generate(const struct dcs_oracle_params *p,
                           const struct dcs_oracle_delay_vals *delays,
                           size_t t0, size_t nt, size_t c0, size_t nc,
                           float *out)
{
    const size_t NR_STATIONS = (size_t)p->nr_stations;
    const size_t NR_BEAMS = (size_t)p->nr_beams;
    double start = now_s();
    for (size_t t = t0; t < t0 + nt; t++) {
        float fDeltaTime = dcs_oracle_delta_time(p, t, k_ref_zero);
        for (size_t c = c0; c < c0 + nc; c++) {
            for (size_t a = 0; a < NR_STATIONS; a++) {
                for (size_t b = 0; b < NR_BEAMS; b++) {
                    size_t iAntBeamOrdering = a * NR_BEAMS + b;
                    struct dcs_oracle_delay_vals sDelayVal = delays[iAntBeamOrdering];
                    size_t ulCoeffIndex = 2 * (((t - t0) * nc + (c - c0)) * NR_STATIONS * NR_BEAMS + iAntBeamOrdering);
                    dcs_oracle_coeff(p, sDelayVal, fDeltaTime, c,
                                     &out[ulCoeffIndex], &out[ulCoeffIndex + 1]);
                }
            }
        }
    }
    return now_s() - start;
}

/* BeamformerCoefficientTest.cu:308-333 with fDeltaTime (:320) of each time step given by the
 * caller instead of derived from the time index (:296-300): the verifier for a caller that holds
 * real (current, reference) times, as the reference's kernels take them
 * (BeamformerKernels.cuh:38-42, 81-86). */
void dcs_oracle_generate_dt(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays,
                            const float *dt, size_t nt, size_t c0, size_t nc,
                            float *out)
{
    const size_t NR_STATIONS = (size_t)p->nr_stations;
    const size_t NR_BEAMS = (size_t)p->nr_beams;
    for (size_t t = 0; t < nt; t++) {
        float fDeltaTime = dt[t];
        for (size_t c = c0; c < c0 + nc; c++) {
            for (size_t a = 0; a < NR_STATIONS; a++) {
                for (size_t b = 0; b < NR_BEAMS; b++) {
                    size_t iAntBeamOrdering = a * NR_BEAMS + b;
                    size_t ulCoeffIndex = 2 * ((t * nc + (c - c0)) * NR_STATIONS * NR_BEAMS + iAntBeamOrdering);
                    dcs_oracle_coeff(p, delays[iAntBeamOrdering], fDeltaTime, c,
                                     &out[ulCoeffIndex], &out[ulCoeffIndex + 1]);
                }
            }
        }
    }
}

/* The same with fDeltaTime = ts_diff(ref, cur[t]) (BeamformerCoefficientTest.cu:320). */
void dcs_oracle_generate_at(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays,
                            const struct timespec *cur, struct timespec ref,
                            size_t nt, size_t c0, size_t nc, float *out)
{
    const size_t step = 2 * nc * (size_t)p->nr_stations * (size_t)p->nr_beams;
    for (size_t t = 0; t < nt; t++) {
        float fDeltaTime = dcs_oracle_ts_diff(ref, cur[t]);
        dcs_oracle_generate_dt(p, delays, &fDeltaTime, 1, c0, nc, out + t * step);
    }
}

static inline uint32_t f32_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

struct cks_job {
    const struct dcs_oracle_params *p;
    const struct dcs_oracle_delay_vals *delays;
    size_t t0, nt, c0, nc;
    uint64_t sum;
};

static void *cks_worker(void *arg)
{
    struct cks_job *j = (struct cks_job *)arg;
    const size_t n = (size_t)j->p->nr_stations * (size_t)j->p->nr_beams;
    uint64_t s = 0;
    for (size_t t = j->t0; t < j->t0 + j->nt; t++) {
        float fDeltaTime = dcs_oracle_delta_time(j->p, t, k_ref_zero);
        for (size_t c = j->c0; c < j->c0 + j->nc; c++) {
            for (size_t i = 0; i < n; i++) {
                float re, im;
                dcs_oracle_coeff(j->p, j->delays[i], fDeltaTime, c, &re, &im);
                s += (uint64_t)f32_bits(re) + (uint64_t)f32_bits(im);
            }
        }
    }
    j->sum = s;
    return NULL;
}

double dcs_oracle_generate_checksum(const struct dcs_oracle_params *p,
                                    const struct dcs_oracle_delay_vals *delays,
                                    size_t t0, size_t nt, size_t c0, size_t nc,
                                    int nthreads, uint64_t *checksum)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > nc && nc > 0) nthreads = (int)nc;
    struct cks_job *jobs = calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
    double start = now_s();
    size_t per = nc / (size_t)nthreads, rem = nc % (size_t)nthreads, c = c0;
    for (int i = 0; i < nthreads; i++) {
        size_t cnt = per + ((size_t)i < rem ? 1 : 0);
        jobs[i] = (struct cks_job){p, delays, t0, nt, c, cnt, 0};
        c += cnt;
        if (i > 0) pthread_create(&th[i], NULL, cks_worker, &jobs[i]);
    }
    cks_worker(&jobs[0]);
    uint64_t sum = jobs[0].sum;
    for (int i = 1; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        sum += jobs[i].sum;
    }
    double secs = now_s() - start;
    free(jobs);
    free(th);
    if (checksum) *checksum = sum;
    return secs;
}

/* BeamformerCoefficientTest.cu:348-357 */
int64_t dcs_oracle_compare(const float *got, const float *expect, size_t n, float tol)
{
    for (size_t i = 0; i < n; i++) {
        if (fabsf(got[i] - expect[i]) > tol) return (int64_t)i;
    }
    return -1;
}

uint32_t dcs_oracle_ulp_diff(float a, float b)
{
    /* map the sign-magnitude bit pattern onto a monotone integer line */
    int64_t ia = (int64_t)(f32_bits(a) & 0x7fffffffu);
    int64_t ib = (int64_t)(f32_bits(b) & 0x7fffffffu);
    if (f32_bits(a) >> 31) ia = -ia;
    if (f32_bits(b) >> 31) ib = -ib;
    int64_t d = ia - ib;
    if (d < 0) d = -d;
    if (a != a || b != b) return 0xffffffffu;
    return d > 0xfffffffe ? 0xfffffffeu : (uint32_t)d;
}

uint32_t dcs_oracle_max_ulp(const float *got, const float *expect, size_t n,
                            uint32_t limit, uint64_t *n_over, int64_t *first_over)
{
    uint32_t mx = 0;
    uint64_t over = 0;
    int64_t first = -1;
    for (size_t i = 0; i < n; i++) {
        uint32_t d = dcs_oracle_ulp_diff(got[i], expect[i]);
        if (d > mx) mx = d;
        if (d > limit) {
            if (first < 0) first = (int64_t)i;
            over++;
        }
    }
    if (n_over) *n_over = over;
    if (first_over) *first_over = first;
    return mx;
}

/* BeamformerCoefficientTest.cu:294-337 + :348-357 fused: the expected coefficient of every element
 * of got[nt][nc][A*B][2] is generated and compared on the fly (no expected tensor is stored), the
 * channel range split over threads.  hist[d] counts elements at ULP distance d = 0, 1, 2 and
 * hist[3] those beyond. */
struct cmp_job {
    const struct dcs_oracle_params *p;
    const struct dcs_oracle_delay_vals *delays;
    const float *dt;
    size_t nt, c0, nc, cb, ce; /* this thread: channels [cb, ce) of the slab [c0, c0+nc) */
    const void *got; /* float [..][2], or (half = 1) binary16 bit patterns in the same order */
    int reading, half;
    uint64_t hist[4];
    uint32_t max_ulp;
    int64_t first_over;
};

static void *cmp_worker(void *arg)
{
    struct cmp_job *j = (struct cmp_job *)arg;
    const size_t n = (size_t)j->p->nr_stations * (size_t)j->p->nr_beams;
    for (size_t t = 0; t < j->nt; t++) {
        const float fDeltaTime = j->dt[t];
        for (size_t c = j->cb; c < j->ce; c++) {
            const size_t base = 2 * ((t * j->nc + (c - j->c0)) * n);
            for (size_t i = 0; i < n; i++) {
                float e[2];
                coeff_reading(j->p, j->delays[i], fDeltaTime, c, j->reading, &e[0], &e[1]);
                for (int k = 0; k < 2; k++) {
                    uint32_t d;
                    if (j->half) { /* expectation: RN-even binary16 of the verifier's fp32; distance in binary16 ulps */
                        const uint16_t g = ((const uint16_t *)j->got)[base + 2 * i + k], w = dcs_oracle_f32_to_f16_rn(e[k]);
                        const int32_t gi = (g & 0x8000u) ? -(int32_t)(g & 0x7fffu) : (int32_t)(g & 0x7fffu);
                        const int32_t wi = (w & 0x8000u) ? -(int32_t)(w & 0x7fffu) : (int32_t)(w & 0x7fffu);
                        d = (uint32_t)(gi > wi ? gi - wi : wi - gi);
                    } else {
                        d = dcs_oracle_ulp_diff(((const float *)j->got)[base + 2 * i + k], e[k]);
                    }
                    j->hist[d > 3 ? 3 : d]++;
                    if (d > j->max_ulp) j->max_ulp = d;
                    if (d > 1 && j->first_over < 0) j->first_over = (int64_t)(base + 2 * i + k);
                }
            }
        }
    }
    return NULL;
}

static double compare_generated(const struct dcs_oracle_params *p,
                                    const struct dcs_oracle_delay_vals *delays,
                                    const float *dt, size_t nt, size_t c0, size_t nc,
                                    const void *got, int half, int nthreads, int reading,
                                    uint64_t hist[4], uint32_t *max_ulp, int64_t *first_over_1ulp)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > nc && nc > 0) nthreads = (int)nc;
    struct cmp_job *jobs = calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
    double start = now_s();
    size_t per = nc / (size_t)nthreads, rem = nc % (size_t)nthreads, c = c0;
    for (int i = 0; i < nthreads; i++) {
        size_t cnt = per + ((size_t)i < rem ? 1 : 0);
        jobs[i] = (struct cmp_job){p, delays, dt, nt, c0, nc, c, c + cnt, got, reading == 1 ? 1 : 0, half, {0, 0, 0, 0}, 0, -1};
        c += cnt;
        if (i > 0) pthread_create(&th[i], NULL, cmp_worker, &jobs[i]);
    }
    cmp_worker(&jobs[0]);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    uint32_t mx = 0;
    int64_t first = -1;
    uint64_t h[4] = {0, 0, 0, 0};
    for (int i = 0; i < nthreads; i++) {
        for (int k = 0; k < 4; k++) h[k] += jobs[i].hist[k];
        if (jobs[i].max_ulp > mx) mx = jobs[i].max_ulp;
        if (jobs[i].first_over >= 0 && (first < 0 || jobs[i].first_over < first)) first = jobs[i].first_over;
    }
    double secs = now_s() - start;
    free(jobs);
    free(th);
    if (hist) memcpy(hist, h, sizeof(h));
    if (max_ulp) *max_ulp = mx;
    if (first_over_1ulp) *first_over_1ulp = first;
    return secs;
}

double dcs_oracle_compare_generated(const struct dcs_oracle_params *p,
                                    const struct dcs_oracle_delay_vals *delays,
                                    const float *dt, size_t nt, size_t c0, size_t nc,
                                    const float *got, int nthreads, int reading,
                                    uint64_t hist[4], uint32_t *max_ulp, int64_t *first_over_1ulp)
{
    return compare_generated(p, delays, dt, nt, c0, nc, got, 0, nthreads, reading, hist, max_ulp, first_over_1ulp);
}

/* The same for the packed binary16 output (BeamformerKernels.cu:113-115, 182-184: __floats2half2_rn of the pair; the
 * reference never verifies it, BeamformerCoefficientTest.cu:282-287): expectation RN-even(verifier's fp32), distances
 * in binary16 ulps. */
double dcs_oracle_compare_generated_f16(const struct dcs_oracle_params *p,
                                        const struct dcs_oracle_delay_vals *delays,
                                        const float *dt, size_t nt, size_t c0, size_t nc,
                                        const uint16_t *got, int nthreads, int reading,
                                        uint64_t hist[4], uint32_t *max_ulp, int64_t *first_over_1ulp)
{
    return compare_generated(p, delays, dt, nt, c0, nc, got, 1, nthreads, reading, hist, max_ulp, first_over_1ulp);
}

/* BeamformerKernels.cu:153-177 (kernel a3's arithmetic), on the host. */
void dcs_oracle_device_variant_a3(const struct dcs_oracle_params *p,
                                  const struct dcs_oracle_delay_vals *delays,
                                  size_t t0, size_t nt, float *out)
{
    const float SAMPLING_PERIOD = p->sampling_period;
    const int NR_CHANNELS = p->nr_channels;
    const int FFT_SIZE = p->fft_size;
    const size_t n = (size_t)p->nr_stations * (size_t)p->nr_beams;
    for (size_t ti = 0; ti < nt; ti++) {
        int iTimeIndex = (int)(t0 + ti);
        for (size_t i = 0; i < n; i++) {
            struct dcs_oracle_delay_vals sDelayValuesLocal = delays[i];
            float fDeltaTime = iTimeIndex * SAMPLING_PERIOD * FFT_SIZE;
            float fDeltaDelay = sDelayValuesLocal.fDelayRate_sps * fDeltaTime;
            float fDeltaPhase = sDelayValuesLocal.fPhaseRate_radps * fDeltaTime;
            float fDelayN2 = (sDelayValuesLocal.fDelay_s + fDeltaDelay) * (NR_CHANNELS / 2) * ((float)M_PI) / (SAMPLING_PERIOD * NR_CHANNELS);
            for (int iChannelIndex = 0; iChannelIndex < NR_CHANNELS; iChannelIndex++) {
                float fDelayN = (sDelayValuesLocal.fDelayRate_sps + fDeltaDelay) * iChannelIndex * ((float)M_PI) / (SAMPLING_PERIOD * NR_CHANNELS);
                float fPhase0 = sDelayValuesLocal.fPhase_rad - fDelayN2 + fDeltaPhase;
                float fRotation = fDelayN + fPhase0;
                size_t ulOutputIndex = ((size_t)NR_CHANNELS * n * ti + (size_t)iChannelIndex * n + i) * 2;
                out[ulOutputIndex] = (float)cos((double)fRotation);
                out[ulOutputIndex + 1] = (float)sin((double)fRotation);
            }
        }
    }
}

/* BeamformerCoefficientTest.cu:198-204 */
void dcs_oracle_simulate_antenna_data(int8_t *out, size_t nbytes)
{
    for (size_t i = 0; i < nbytes; i++) out[i] = (int8_t)i;
}

/* BeamformerCoefficientTest.cu:294-337 (ordering :311) + :363-414; channels [c0, c0 + nc), the two tensors pointing at
 * that slab (the reference's loop is c0 = 0, nc = NR_CHANNELS) */
static void beamform_impl(const struct dcs_oracle_params *p,
                          const struct dcs_oracle_delay_vals *delays, const float *dt, size_t nt, size_t c0, size_t nc,
                          const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    const size_t NR_CHANNELS = nc;
    const size_t NR_STATIONS = (size_t)p->nr_stations;
    const size_t NR_BEAMS = (size_t)p->nr_beams;
    const size_t NR_SAMPLES_PER_CHANNEL = nt;
    const size_t INTERNAL_TIME_SAMPLES = 16;
    for (size_t c = 0; c < NR_CHANNELS; c++) {
        for (size_t t_ex = 0; t_ex < NR_SAMPLES_PER_CHANNEL / INTERNAL_TIME_SAMPLES; t_ex++) {
            for (size_t t_in = 0; t_in < INTERNAL_TIME_SAMPLES; t_in++) {
                const size_t t = t_ex * INTERNAL_TIME_SAMPLES + t_in;
                float fDeltaTime = dt ? dt[t] : dcs_oracle_delta_time(p, t, k_ref_zero);
                for (size_t b = 0; b < NR_BEAMS; b++) {
                    size_t iBeamIndex = c * NR_SAMPLES_PER_CHANNEL * NR_BEAMS + t_ex * NR_BEAMS * INTERNAL_TIME_SAMPLES + b * INTERNAL_TIME_SAMPLES + t_in;
                    float fBeamSumReal = 0;
                    float fBeamSumImag = 0;
                    for (size_t a = 0; a < NR_STATIONS; a++) {
                        float fRealSteeringCoeff, fImagSteeringCoeff;
                        dcs_oracle_coeff(p, delays[b * NR_STATIONS + a], fDeltaTime, c0 + c,
                                         &fRealSteeringCoeff, &fImagSteeringCoeff);
                        size_t ulAntSampleIndex = 2 * (c * NR_SAMPLES_PER_CHANNEL * NR_STATIONS + t_ex * NR_STATIONS * INTERNAL_TIME_SAMPLES + a * INTERNAL_TIME_SAMPLES + t_in);
                        int8_t iRealAntSample = pi8InAntData[ulAntSampleIndex];
                        int8_t iImagAntSample = pi8InAntData[ulAntSampleIndex + 1];
                        fBeamSumReal += fRealSteeringCoeff * iRealAntSample;
                        fBeamSumImag += fImagSteeringCoeff * iImagAntSample;
                    }
                    pfCorrectBeams[2 * iBeamIndex] = fBeamSumReal;
                    pfCorrectBeams[2 * iBeamIndex + 1] = fBeamSumImag;
                }
            }
        }
    }
}

void dcs_oracle_beamform(const struct dcs_oracle_params *p,
                         const struct dcs_oracle_delay_vals *delays, size_t nt,
                         const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    beamform_impl(p, delays, NULL, nt, 0, (size_t)p->nr_channels, pi8InAntData, pfCorrectBeams);
}

/* The same with fDeltaTime of each of the nt samples given (cf. dcs_oracle_generate_dt). */
void dcs_oracle_beamform_dt(const struct dcs_oracle_params *p,
                            const struct dcs_oracle_delay_vals *delays, const float *dt, size_t nt,
                            const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    beamform_impl(p, delays, dt, nt, 0, (size_t)p->nr_channels, pi8InAntData, pfCorrectBeams);
}

/* Channels [c0, c0 + nc) of dcs_oracle_beamform_dt (dt == NULL: time indices 0 .. nt-1 as dcs_oracle_beamform);
 * pi8InAntData / pfCorrectBeams point at the slab. */
void dcs_oracle_beamform_slab(const struct dcs_oracle_params *p,
                              const struct dcs_oracle_delay_vals *delays, const float *dt, size_t nt, size_t c0, size_t nc,
                              const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    beamform_impl(p, delays, dt, nt, c0, nc, pi8InAntData, pfCorrectBeams);
}

/* BeamformerCoefficientTest.cu:363-414 with the coefficient HELD: the steering coefficient of
 * (channel, beam, antenna) is evaluated once, at fDeltaTime dt_coeff (:319-328, table ordering :311),
 * and applied to all nt samples -- what ACCUMULATIONS_BEFORE_NEW_COEFFS (BeamformerParameters.h:17;
 * the utilisation model :426-448) assumes of a deployed beamformer; the reference has no kernel for it.
 * The sums are the verifier's: antenna order, fBeamSum += coeff * sample (multiply and add rounded
 * separately).
 *
 * _slab: channels [c0, c0 + nc) of it; pi8InAntData / pfCorrectBeams point at the slab (its first channel is c0), so
 * that a few channels of a large problem can be checked without evaluating all of it. */
void dcs_oracle_beamform_accumulated_slab(const struct dcs_oracle_params *p,
                                          const struct dcs_oracle_delay_vals *delays, float dt_coeff, size_t nt,
                                          size_t c0, size_t nc, const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    const size_t NR_STATIONS = (size_t)p->nr_stations;
    const size_t NR_BEAMS = (size_t)p->nr_beams;
    const size_t NR_SAMPLES_PER_CHANNEL = nt;
    const size_t INTERNAL_TIME_SAMPLES = 16;
    float *coeff = malloc(NR_STATIONS * 2 * sizeof(float));
    for (size_t c = 0; c < nc; c++) {
        for (size_t b = 0; b < NR_BEAMS; b++) {
            for (size_t a = 0; a < NR_STATIONS; a++)
                dcs_oracle_coeff(p, delays[b * NR_STATIONS + a], dt_coeff, c0 + c, &coeff[2 * a], &coeff[2 * a + 1]);
            for (size_t t = 0; t < NR_SAMPLES_PER_CHANNEL; t++) {
                const size_t t_ex = t / INTERNAL_TIME_SAMPLES, t_in = t % INTERNAL_TIME_SAMPLES;
                size_t iBeamIndex = c * NR_SAMPLES_PER_CHANNEL * NR_BEAMS + t_ex * NR_BEAMS * INTERNAL_TIME_SAMPLES + b * INTERNAL_TIME_SAMPLES + t_in;
                float fBeamSumReal = 0;
                float fBeamSumImag = 0;
                for (size_t a = 0; a < NR_STATIONS; a++) {
                    size_t ulAntSampleIndex = 2 * (c * NR_SAMPLES_PER_CHANNEL * NR_STATIONS + t_ex * NR_STATIONS * INTERNAL_TIME_SAMPLES + a * INTERNAL_TIME_SAMPLES + t_in);
                    int8_t iRealAntSample = pi8InAntData[ulAntSampleIndex];
                    int8_t iImagAntSample = pi8InAntData[ulAntSampleIndex + 1];
                    fBeamSumReal += coeff[2 * a] * iRealAntSample;
                    fBeamSumImag += coeff[2 * a + 1] * iImagAntSample;
                }
                pfCorrectBeams[2 * iBeamIndex] = fBeamSumReal;
                pfCorrectBeams[2 * iBeamIndex + 1] = fBeamSumImag;
            }
        }
    }
    free(coeff);
}

void dcs_oracle_beamform_accumulated(const struct dcs_oracle_params *p,
                                     const struct dcs_oracle_delay_vals *delays, float dt_coeff, size_t nt,
                                     const int8_t *pi8InAntData, float *pfCorrectBeams)
{
    dcs_oracle_beamform_accumulated_slab(p, delays, dt_coeff, nt, 0, (size_t)p->nr_channels, pi8InAntData, pfCorrectBeams);
}

/* IEEE binary16 round-to-nearest-even of an fp32 (what __floats2half2_rn does
 * per element, BeamformerKernels.cu:113,182). */
uint16_t dcs_oracle_f32_to_f16_rn(float x)
{
    uint32_t u = f32_bits(x);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t abs = u & 0x7fffffffu;
    if (abs >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (abs > 0x7f800000u ? 0x200u | ((abs >> 13) & 0x3ffu) : 0));
    if (abs >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if (abs < 0x33000001u) /* < 2^-25 (or exactly 2^-25, ties to even 0) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(abs >> 23) - 127;
    uint32_t m = (abs & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;                             /* bits dropped from m */
    uint32_t hexp;
    if (e < -14) { /* subnormal half: value = m * 2^(e-23), unit 2^-24 */
        shift = (uint32_t)(-e - 14 + 13);
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    /* normal: q has the implicit bit at 0x400; adding it to hexp<<10 carries
     * correctly into the exponent on overflow of the significand. */
    uint32_t h = (hexp == 0) ? q : (((hexp - 1) << 10) + q);
    return (uint16_t)(sign | h);
}
