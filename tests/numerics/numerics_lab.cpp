/*
 * numerics_lab.c -- host build of dc_sand_amd/csrc/bf_math.h, used ONLY by the
 * tests to sweep the device operation sequences exhaustively on the CPU (an
 * fp32 fma / mul / add / rint gives the same bits on x86-64 and on gfx950).
 * Test infrastructure: never loaded by dc_sand_amd/.
 *
 * Build: g++ -O2 -ffp-contract=off [-mfma] -shared -fPIC (tests/numerics/Makefile)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../dc_sand_amd/csrc/bf_math.h"

static inline uint32_t ulp_diff(float a, float b)
{
    if (a != a || b != b) return 0xffffffffu;
    int64_t ia = (int64_t)(dcs_f32_bits(a) & 0x7fffffffu);
    int64_t ib = (int64_t)(dcs_f32_bits(b) & 0x7fffffffu);
    if (dcs_f32_bits(a) >> 31) ia = -ia;
    if (dcs_f32_bits(b) >> 31) ib = -ib;
    int64_t d = ia - ib;
    if (d < 0) d = -d;
    return d > 0xfffffffe ? 0xfffffffeu : (uint32_t)d;
}

struct sweep_job {
    uint32_t lo, hi;     /* bit patterns [lo, hi) */
    int variant;
    float D, y;          /* div sweep */
    uint32_t max_a, max_b;
    uint64_t over_a, over_b;
    uint32_t worst_a, worst_b;
    /* sincos sweep only: faithfulness (result is one of the two floats bracketing the true
     * value) and distance to this host's float libm (sinf / cosf) */
    uint64_t unfaithful_a, unfaithful_b, over_f_a, over_f_b;
    uint32_t max_f_a, max_f_b;
};

/* r is RD(t) or RU(t) of the true value, taken from its double evaluation td (2^-53 relative:
 * ambiguous only if td is within that of a float, where both neighbours are accepted anyway) */
static inline int faithful(float r, double td)
{
    const float f = (float)td;
    if ((double)f == td) return r == f;
    const float other = (double)f < td ? nextafterf(f, INFINITY) : nextafterf(f, -INFINITY);
    return r == f || r == other;
}

static void *sincos_worker(void *arg)
{
    struct sweep_job *j = (struct sweep_job *)arg;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        float x = dcs_bits_f32(u);
        float s, c;
        if (j->variant == 1)
            dcs_sincos_fast<true>(x, &s, &c);
        else
            dcs_sincos_fast<false>(x, &s, &c);
        float es = (float)sin((double)x), ec = (float)cos((double)x);
        uint32_t ds = ulp_diff(s, es), dc = ulp_diff(c, ec);
        if (ds > j->max_a) { j->max_a = ds; j->worst_a = u; }
        if (dc > j->max_b) { j->max_b = dc; j->worst_b = u; }
        if (ds > 1) j->over_a++;
        if (dc > 1) j->over_b++;
        if (!faithful(s, sin((double)x))) j->unfaithful_a++;
        if (!faithful(c, cos((double)x))) j->unfaithful_b++;
        const uint32_t fs = ulp_diff(s, sinf(x)), fc = ulp_diff(c, cosf(x));
        if (fs > j->max_f_a) j->max_f_a = fs;
        if (fc > j->max_f_b) j->max_f_b = fc;
        if (fs > 1) j->over_f_a++;
        if (fc > 1) j->over_f_b++;
    }
    return NULL;
}

static inline int half_ordered(uint32_t h) { return (h & 0x8000u) ? -(int)(h & 0x7fffu) : (int)(h & 0x7fffu); }

/* dcs_sincos_half2 (the b16 arithmetic form) against RN16 of the correctly rounded fp32 value --
 * the oracle's b16 expectation -- and against RN16 of the fp32 fast path (the default b16 form).
 * max_a / max_b: max binary16-ulp distance of sin / cos to RN16((float)sin((double)x));
 * over_a / over_b: arguments beyond 1; unfaithful_a / _b (reused): arguments where it differs at all;
 * over_f_a / over_f_b (reused): arguments where it differs from RN16(dcs_sincos_fast<true>). */
static void *half_worker(void *arg)
{
    struct sweep_job *j = (struct sweep_job *)arg;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        const float x = dcs_bits_f32(u);
        const uint32_t p = dcs_sincos_half2(x);
        const uint32_t hc = p & 0xffffu, hs = p >> 16;
        const uint32_t es = dcs_f32_to_f16_bits((float)sin((double)x)), ec = dcs_f32_to_f16_bits((float)cos((double)x));
        const uint32_t ds = (uint32_t)abs(half_ordered(hs) - half_ordered(es)), dc = (uint32_t)abs(half_ordered(hc) - half_ordered(ec));
        if (ds > j->max_a) { j->max_a = ds; j->worst_a = u; }
        if (dc > j->max_b) { j->max_b = dc; j->worst_b = u; }
        if (ds > 1) j->over_a++;
        if (dc > 1) j->over_b++;
        if (ds) j->unfaithful_a++;
        if (dc) j->unfaithful_b++;
        float fs, fc;
        dcs_sincos_fast<true>(x, &fs, &fc);
        if (dcs_f32_to_f16_bits(fs) != hs) j->over_f_a++;
        if (dcs_f32_to_f16_bits(fc) != hc) j->over_f_b++;
    }
    return NULL;
}

/* dcs_sincos_fast_half2 (the packed b16 word from the fp32-grade pair, quadrant logic on the packed word) against
 * RN-even of dcs_sincos_fast's fp32 pair, bit for bit: over_a counts the arguments where they differ. */
template <bool LOW>
static void *fast_half2_worker(void *arg)
{
    struct sweep_job *j = (struct sweep_job *)arg;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        for (int neg = 0; neg < 2; neg++) {
            const float x = dcs_bits_f32(u | (neg ? 0x80000000u : 0u));
            float fs, fc;
            dcs_sincos_fast<LOW>(x, &fs, &fc);
            const uint32_t want = dcs_f32_to_f16_bits(fc) | (dcs_f32_to_f16_bits(fs) << 16);
            if (dcs_sincos_fast_half2<LOW>(x) != want) {
                if (j->over_a == 0) j->worst_a = u | (neg ? 0x80000000u : 0u);
                j->over_a++;
            }
        }
    }
    return NULL;
}

static void *div_worker(void *arg)
{
    struct sweep_job *j = (struct sweep_job *)arg;
    for (uint32_t u = j->lo; u < j->hi; u++) {
        float x = dcs_bits_f32(u);
        float q = j->variant == 1 ? dcs_div_const3(x, j->D, j->y) : dcs_div_const(x, j->D, j->y);
        volatile float xd = x, dd = j->D;
        float e = xd / dd;
        if (dcs_f32_bits(q) != dcs_f32_bits(e)) {
            /* +0 vs -0 for x == -0 is tolerated by the callers (ULP metric);
             * count it apart */
            if (q == e) { j->over_b++; continue; }
            j->over_a++;
            if (j->max_a == 0) j->worst_a = u;
            uint32_t d = ulp_diff(q, e);
            if (d > j->max_a) j->max_a = d;
        }
    }
    return NULL;
}

static void run_sweep(void *(*fn)(void *), struct sweep_job *proto, int nthreads,
                      struct sweep_job *out)
{
    if (nthreads < 1) nthreads = 1;
    struct sweep_job *jobs = (struct sweep_job *)calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
    uint64_t span = (uint64_t)proto->hi - proto->lo;
    for (int i = 0; i < nthreads; i++) {
        jobs[i] = *proto;
        jobs[i].lo = proto->lo + (uint32_t)(span * (uint64_t)i / (uint64_t)nthreads);
        jobs[i].hi = proto->lo + (uint32_t)(span * (uint64_t)(i + 1) / (uint64_t)nthreads);
        pthread_create(&th[i], NULL, fn, &jobs[i]);
    }
    *out = *proto;
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        if (jobs[i].max_a > out->max_a) { out->max_a = jobs[i].max_a; out->worst_a = jobs[i].worst_a; }
        if (jobs[i].max_b > out->max_b) { out->max_b = jobs[i].max_b; out->worst_b = jobs[i].worst_b; }
        out->over_a += jobs[i].over_a;
        out->over_b += jobs[i].over_b;
        out->unfaithful_a += jobs[i].unfaithful_a;
        out->unfaithful_b += jobs[i].unfaithful_b;
        out->over_f_a += jobs[i].over_f_a;
        out->over_f_b += jobs[i].over_f_b;
        if (jobs[i].max_f_a > out->max_f_a) out->max_f_a = jobs[i].max_f_a;
        if (jobs[i].max_f_b > out->max_f_b) out->max_f_b = jobs[i].max_f_b;
    }
    free(jobs);
    free(th);
}

template <bool DIV3, bool LOW>
static void coeff_fast_host(float k, float p0, float fc, float D, float y, float *re, float *im)
{
    const float rot = dcs_rotation<DIV3>(k, p0, fc, D, y);
    dcs_sincos_fast<LOW>(rot, im, re);
}

extern "C" {

/* Sweep every fp32 with bit pattern in [lo_bits, hi_bits) (positive floats:
 * monotone in value).  res = {max_ulp_sin, max_ulp_cos, n_sin_over_1,
 * n_cos_over_1, worst_x_bits_sin, worst_x_bits_cos}. */
void lab_sincos_sweep(int lowdeg, uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t *res)
{
    struct sweep_job p = {}, o;
    p.lo = lo_bits;
    p.hi = hi_bits;
    p.variant = lowdeg;
    run_sweep(sincos_worker, &p, nthreads, &o);
    res[0] = o.max_a; res[1] = o.max_b; res[2] = o.over_a; res[3] = o.over_b;
    res[4] = o.worst_a; res[5] = o.worst_b;
}

/* Same sweep, the two further properties: res = {n_unfaithful_sin, n_unfaithful_cos,
 * n_sin_over_1_vs_sinf, n_cos_over_1_vs_cosf, max_ulp_vs_sinf, max_ulp_vs_cosf}. */
void lab_sincos_sweep_faithful(int lowdeg, uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t *res)
{
    struct sweep_job p = {}, o;
    p.lo = lo_bits;
    p.hi = hi_bits;
    p.variant = lowdeg;
    run_sweep(sincos_worker, &p, nthreads, &o);
    res[0] = o.unfaithful_a; res[1] = o.unfaithful_b; res[2] = o.over_f_a; res[3] = o.over_f_b;
    res[4] = o.max_f_a; res[5] = o.max_f_b;
}

/* The b16 arithmetic form over every fp32 with bit pattern in [lo_bits, hi_bits):
 * res = {max_half_ulp_sin, max_half_ulp_cos, n_sin_over_1, n_cos_over_1, n_sin_not_rn16_exact, n_cos_not_rn16_exact,
 *        n_sin_not_rn16_of_fp32_path, n_cos_not_rn16_of_fp32_path, worst_x_bits_sin, worst_x_bits_cos}. */
void lab_half_sweep(uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t *res)
{
    struct sweep_job p = {}, o;
    p.lo = lo_bits;
    p.hi = hi_bits;
    run_sweep(half_worker, &p, nthreads, &o);
    res[0] = o.max_a; res[1] = o.max_b; res[2] = o.over_a; res[3] = o.over_b;
    res[4] = o.unfaithful_a; res[5] = o.unfaithful_b; res[6] = o.over_f_a; res[7] = o.over_f_b;
    res[8] = o.worst_a; res[9] = o.worst_b;
}

/* res = {n_arguments_where dcs_sincos_fast_half2 != RN16(dcs_sincos_fast), first_such_x_bits}; both signs of every
 * fp32 with bit pattern in [lo_bits, hi_bits). */
void lab_fast_half2_sweep(int lowdeg, uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t *res)
{
    struct sweep_job p = {}, o;
    p.lo = lo_bits;
    p.hi = hi_bits;
    run_sweep(lowdeg ? fast_half2_worker<true> : fast_half2_worker<false>, &p, nthreads, &o);
    res[0] = o.over_a;
    res[1] = o.worst_a;
}

void lab_sincos_half2(const float *x, size_t n, uint32_t *packed)
{
    for (size_t i = 0; i < n; i++) packed[i] = dcs_sincos_half2(x[i]);
}

/* res = {n_mismatch, max_ulp, first_bad_x_bits, n_signed_zero_diffs} */
void lab_div_sweep(int three_op, float D, uint32_t lo_bits, uint32_t hi_bits, int nthreads, uint64_t *res)
{
    struct sweep_job p = {}, o;
    p.lo = lo_bits;
    p.hi = hi_bits;
    p.variant = three_op;
    p.D = D;
    p.y = 1.0f / D;
    run_sweep(div_worker, &p, nthreads, &o);
    res[0] = o.over_a; res[1] = o.max_a; res[2] = o.worst_a; res[3] = o.over_b;
}

void lab_sincos(int lowdeg, const float *x, size_t n, float *s, float *c)
{
    for (size_t i = 0; i < n; i++) {
        if (lowdeg)
            dcs_sincos_fast<true>(x[i], &s[i], &c[i]);
        else
            dcs_sincos_fast<false>(x[i], &s[i], &c[i]);
    }
}

/* The device fast path, end to end, for one time step: out[c][i][2].  mode bit 0:
 * 3-op divide (as when dcs_bf_create verified it), bit 1: force the full polynomials. */
void lab_generate_fast(int mode, const struct dcs_delay_vals *delays, size_t n_pairs,
                       int32_t nr_channels, float sampling_period, float fDeltaTime,
                       size_t c0, size_t nc, float *out, uint64_t *n_slow_pairs, uint64_t *n_low_pairs)
{
    const float D = sampling_period * nr_channels;
    const float y = 1.0f / D;
    const double half = nr_channels / 2.0;
    const float scale = (float)((double)(nr_channels - 1) * 3.14159274101257324219 / (double)D * 1.0001);
    uint64_t slow = 0, low = 0;
    for (size_t i = 0; i < n_pairs; i++) {
        float k, p0;
        dcs_pair_terms(delays[i], fDeltaTime, half, (double)D, &k, &p0);
        const uint32_t cls = dcs_pair_class(k, p0, scale, (mode & 2) ? 0.0f : 500.0f);
        if (cls == DCS_CLASS_SLOW) slow++;
        if (cls == DCS_CLASS_FAST_LOW) low++;
        for (size_t c = c0; c < c0 + nc; c++) {
            float s, co;
            const float fc = (float)c;
            if (cls == DCS_CLASS_SLOW) {
                const float rot = dcs_rotation_ieee(k, p0, fc, D);
                s = (float)sin((double)rot);
                co = (float)cos((double)rot);
            } else if (mode & 1) {
                if (cls == DCS_CLASS_FAST_LOW) coeff_fast_host<true, true>(k, p0, fc, D, y, &co, &s);
                else coeff_fast_host<true, false>(k, p0, fc, D, y, &co, &s);
            } else {
                if (cls == DCS_CLASS_FAST_LOW) coeff_fast_host<false, true>(k, p0, fc, D, y, &co, &s);
                else coeff_fast_host<false, false>(k, p0, fc, D, y, &co, &s);
            }
            out[2 * ((c - c0) * n_pairs + i)] = co;
            out[2 * ((c - c0) * n_pairs + i) + 1] = s;
        }
    }
    if (n_slow_pairs) *n_slow_pairs = slow;
    if (n_low_pairs) *n_low_pairs = low;
}

} // extern "C"
