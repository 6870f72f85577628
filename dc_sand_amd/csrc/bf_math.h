// bf_math.h -- the fp32 arithmetic of the steering-coefficient hot path.
//
// One definition, two compilers: hipcc (device code, gfx950) for the kernels in
// bf_kernels.hip, and gcc for tests/numerics (a host build used ONLY to sweep
// these exact operation sequences exhaustively against the oracle; it is not a
// product fallback and nothing in dc_sand_amd/ loads it).
//
// Every operation here is a single IEEE fp32/fp64 operation with one rounding:
// fused multiply-adds are written as dcs_fmaf(), everything else must stay
// unfused -- both builds use -ffp-contract=off.  The sequence to reproduce is
// the reference CPU verifier's (beamformer_coefficient_generator/
// BeamformerCoefficientTest.cu:321-328).
#ifndef DCS_BF_MATH_H
#define DCS_BF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DCS_HD __host__ __device__ __forceinline__
#else
#define DCS_HD static inline __attribute__((always_inline))
#endif

#define dcs_fmaf(a, b, c) __builtin_fmaf((a), (b), (c))

// (float)M_PI, the constant the reference multiplies by (.cu:322-323).
#define DCS_PI_F 3.14159274101257324219f

// 16-byte table entry: beamformer_coefficient_generator/BeamformerParameters.h:61-66
// (also declared, identically, by include/dcs_beamformer.h)
#ifndef DCS_DELAY_VALS_DEFINED
#define DCS_DELAY_VALS_DEFINED
struct dcs_delay_vals {
    float fDelay_s;
    float fDelayRate_sps;
    float fPhase_rad;
    float fPhaseRate_radps;
};
#endif

// Per-launch constants derived on the host from (SAMPLING_PERIOD, NR_CHANNELS).
struct dcs_bf_consts {
    float fDenominator;      // SAMPLING_PERIOD * NR_CHANNELS, fp32 product (.cu:322)
    float fRcpDenominator;   // RN(1 / fDenominator), fp32 divide on the host
    float fRotBoundScale;    // >= pi * (NR_CHANNELS-1) / fDenominator, with margin
    uint32_t uDiv3Exact;     // 1: dcs_div_const3 verified == IEEE divide for THIS fDenominator (bf_capi.hip)
    float fLowDegLimit;      // 500: |fRotation| bound below which the low-degree polynomials are used (0 = never)
    uint32_t uHalfMath;      // 1: b16 output uses dcs_sincos_half2 (binary16-sized sincos) for every wave outside the slow class
    double dHalfChannels;    // NR_CHANNELS / 2.0 (.cu:323)
    double dDenominator;     // (double) fDenominator (.cu:323)
};

// ---------------------------------------------------------------------------
// Per-(antenna, beam, time) terms that do not depend on the channel:
//   fRateTerm = fDelayRate_sps + fDeltaDelay                (.cu:321-322)
//   fPhase0   = fPhase_rad - fDelayN2 + fDeltaPhase         (.cu:323-325)
// fDelayN2 is the one double-precision chain of the verifier: the fp32 sum
// (fDelay_s + fDeltaDelay) is promoted, multiplied by NR_CHANNELS/2.0 and
// (double)(float)M_PI, divided by (double)(SAMPLING_PERIOD*NR_CHANNELS) and
// rounded to fp32 once.
// ---------------------------------------------------------------------------
DCS_HD void dcs_pair_terms(const dcs_delay_vals d, const float fDeltaTime,
                           const double dHalfChannels, const double dDenominator,
                           float *fRateTerm, float *fPhase0)
{
    const float fDeltaDelay = d.fDelayRate_sps * fDeltaTime;
    const float fRate = d.fDelayRate_sps + fDeltaDelay;
    const float fSum = d.fDelay_s + fDeltaDelay;
    const double dN2 = (((double)fSum * dHalfChannels) * (double)DCS_PI_F) / dDenominator;
    const float fDelayN2 = (float)dN2;
    const float fDeltaPhase = d.fPhaseRate_radps * fDeltaTime;
    const float fDiff = d.fPhase_rad - fDelayN2;
    *fRateTerm = fRate;
    *fPhase0 = fDiff + fDeltaPhase;
}

// ---------------------------------------------------------------------------
// Correctly rounded x / D for a launch-constant D, without the hardware divide
// sequence: y = RN(1/D); two Markstein correction steps.  q1 is within half an
// ulp (+2^-47 relative) of x/D, hence faithful; with r1 = x - q1*D exact (fma),
// RN(q1 + r1*y) is the correctly rounded quotient (Markstein 1990, Thm. 8.5 in
// Muller et al., "Handbook of Floating-Point Arithmetic").  Valid while no
// intermediate under/overflows: callers guarantee 2^-60 <= |x| <= 2^90 or
// x == 0 and 2^-40 <= D <= 2^40 (dcs_rate_in_fast_range + host check).
// tests/numerics sweeps it against the IEEE divide.
// ---------------------------------------------------------------------------
DCS_HD float dcs_div_const(const float x, const float D, const float y)
{
    const float q0 = x * y;
    const float r0 = dcs_fmaf(-q0, D, x);
    const float q1 = dcs_fmaf(r0, y, q0);
    const float r1 = dcs_fmaf(-q1, D, x);
    return dcs_fmaf(r1, y, q1);
}

// The first correction step alone: q1 = RN(q0 + r0*y).  For almost every D this is
// already the correctly rounded quotient for ALL x (Brisebarre, Muller, Raina,
// "Accelerating correctly rounded floating-point division when the divisor is
// known in advance", IEEE TC 2004: the exceptions are a few divisors with
// particular significands).  The sequence is invariant under scaling x by powers
// of two, so checking the 2^23 significands of one binade against the IEEE
// divide settles it for a given D: dcs_bf_create does that on the device and
// sets dcs_bf_consts::uDiv3Exact; kernels use this form only then.
DCS_HD float dcs_div_const3(const float x, const float D, const float y)
{
    const float q0 = x * y;
    const float r0 = dcs_fmaf(-q0, D, x);
    return dcs_fmaf(r0, y, q0);
}

// fRotation for channel c (.cu:322,326): three fp32 roundings for fDelayN
// (mul, mul, divide) and one add.
template <bool DIV3 = false>
DCS_HD float dcs_rotation(const float fRateTerm, const float fPhase0, const float fChannel,
                          const float D, const float y)
{
    const float m1 = fRateTerm * fChannel;
    const float m2 = m1 * DCS_PI_F;
    const float fDelayN = DIV3 ? dcs_div_const3(m2, D, y) : dcs_div_const(m2, D, y);
    return fDelayN + fPhase0;
}

// Same with the IEEE divide (slow path: operands outside dcs_div_const's range).
DCS_HD float dcs_rotation_ieee(const float fRateTerm, const float fPhase0, const float fChannel,
                               const float D)
{
    const float m1 = fRateTerm * fChannel;
    const float m2 = m1 * DCS_PI_F;
    const float fDelayN = m2 / D;
    return fDelayN + fPhase0;
}

DCS_HD uint32_t dcs_f32_bits(const float f) { return __builtin_bit_cast(uint32_t, f); }
DCS_HD float dcs_bits_f32(const uint32_t u) { return __builtin_bit_cast(float, u); }

// |fRateTerm| in [2^-60, 2^60] or zero: dcs_div_const is exact for every channel
// index up to 2^24.
DCS_HD bool dcs_rate_in_fast_range(const float fRateTerm)
{
    const uint32_t e = (dcs_f32_bits(fRateTerm) >> 23) & 0xffu;
    return (fRateTerm == 0.0f) || (e >= 127u - 60u && e <= 127u + 60u);
}

// Class of a pair for a whole launch, from a bound on |fRotation| over its
// channels: |fDelayN| <= |fRateTerm| * pi*(C-1)/D * (1 + 2^-21) (fRotBoundScale
// carries a 1e-4 margin).
//   0: bound < fLowDegLimit (500) -> fast path, low-degree polynomials (proven below 512)
//   1: bound < 32000 -> fast path, full polynomials (proven below 32768)
//   2: otherwise, or rate term outside dcs_div_const's range, or NaN/Inf
//      -> slow path (IEEE divide + fp64 sincos)
#define DCS_CLASS_FAST_LOW 0u
#define DCS_CLASS_FAST_HIGH 1u
#define DCS_CLASS_SLOW 2u
DCS_HD uint32_t dcs_pair_class(const float fRateTerm, const float fPhase0, const float fRotBoundScale,
                               const float fLowDegLimit = 500.0f)
{
    const float bound = __builtin_fabsf(fRateTerm) * fRotBoundScale + __builtin_fabsf(fPhase0);
    if (!dcs_rate_in_fast_range(fRateTerm) || !(bound < 32000.0f)) return DCS_CLASS_SLOW;
    return bound < fLowDegLimit ? DCS_CLASS_FAST_LOW : DCS_CLASS_FAST_HIGH;
}
DCS_HD bool dcs_pair_is_fast(const float fRateTerm, const float fPhase0, const float fRotBoundScale)
{
    return dcs_pair_class(fRateTerm, fPhase0, fRotBoundScale) != DCS_CLASS_SLOW;
}

// ---------------------------------------------------------------------------
// sin and cos of an fp32 argument, |x| < DCS_SINCOS_FAST_LIMIT, each within
// 1 ULP of the correctly rounded value ((float)sin((double)x), .cu:327-328).
// tests/test_numerics.py checks EVERY fp32 in the range (proof by exhaustion;
// nothing here is a statistical claim).
//
// Reduction: the quotient n = rint(x * 2/pi) comes out of the low mantissa bits
// of x*(2/pi) + 1.5*2^23 (no float->int convert); r = x - n*pi/2 with pi/2
// split in three fp32 parts (Cody-Waite with fma).  The first step is exact (x
// and n*P1 agree to ~pi/4 and both sit on a 2^-24 grid), the second and third
// round once each, so r carries one half-ulp of error.
// Polynomials on [-pi/4, pi/4]: sin r = r + r*s*Ps(s), cos r = 1 + s*Pc(s),
// s = r*r; the last operation of each adds a small correction to r or to 1.
// Quadrant q = n mod 4 rotates (cos r, sin r) by q quarter turns.
// ---------------------------------------------------------------------------
#define DCS_SINCOS_FAST_LIMIT 32768.0f

#define DCS_TWO_OVER_PI 0.636619746685028076171875f   // RN(2/pi)
#define DCS_RINT_MAGIC 12582912.0f                      // 1.5 * 2^23
#define DCS_PIO2_1 1.57079637050628662109375f          // RN(pi/2)
#define DCS_PIO2_2 -4.37113882867379114032e-08f        // RN(pi/2 - P1)
#define DCS_PIO2_3 -1.71512451733121152534e-15f        // RN(pi/2 - P1 - P2)

// sin(r) ~ r + r*s*(S1 + s*(S2 + s*(S3 + s*S4))),  s = r*r
#define DCS_S1 -1.66666671633720397949e-01f
#define DCS_S2 8.33333376795053482056e-03f
#define DCS_S3 -1.98412701138295233250e-04f
#define DCS_S4 2.75573142971552442759e-06f
// cos(r) ~ 1 + s*(-1/2 + s*(C1 + s*(C2 + s*(C3 + s*C4))))
#define DCS_C1 4.16666679084300994873e-02f
#define DCS_C2 -1.38888892251998186111e-03f
#define DCS_C3 2.48015876422869041562e-05f
#define DCS_C4 -2.75573142971552442759e-07f

// Low-degree set (sin to r^7, cos to r^8): within 1 ULP for every fp32 below 512
// (swept like the full set; first failures appear in [512, 2048)).
#define DCS_SINCOS_LOW_LIMIT 512.0f
#define DCS_LS1 -1.66666552424430847168e-01f
#define DCS_LS2 8.33216123282909393311e-03f
#define DCS_LS3 -1.95152955711819231510e-04f
#define DCS_LC1 4.16666455566883087158e-02f
#define DCS_LC2 -1.38873164542019367218e-03f
#define DCS_LC3 2.44331567955669015646e-05f

// v ^ (m & 0x80000000): one v_bitop3_b32 on gfx950 (truth table 0x6c over (m, v, mask)).
DCS_HD uint32_t dcs_xor_sign_of(const uint32_t v, const uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(m, v, 0x80000000u, 0x6c);
#else
    return v ^ (m & 0x80000000u);
#endif
}

// sin r, cos r of the reduced argument and the word whose low bits are the quadrant number n (mod 2^22): the part of the
// fp32-grade sincos that both finishers below share.
template <bool LOWDEG = false>
DCS_HD void dcs_sincos_core(const float x, float *fSinR, float *fCosR, uint32_t *uQ)
{
    const float nb = dcs_fmaf(x, DCS_TWO_OVER_PI, DCS_RINT_MAGIC);
    const float n = nb - DCS_RINT_MAGIC;
    const uint32_t q = dcs_f32_bits(nb);                   // low bits = n mod 2^22
    float r = dcs_fmaf(-n, DCS_PIO2_1, x);                 // exact
    r = dcs_fmaf(-n, DCS_PIO2_2, r);
    r = dcs_fmaf(-n, DCS_PIO2_3, r);
    const float s = r * r;

    float ps, pc;
    if (LOWDEG) {
        ps = dcs_fmaf(s, DCS_LS3, DCS_LS2);
        ps = dcs_fmaf(ps, s, DCS_LS1);
        pc = dcs_fmaf(s, DCS_LC3, DCS_LC2);
        pc = dcs_fmaf(pc, s, DCS_LC1);
    } else {
        ps = dcs_fmaf(s, DCS_S4, DCS_S3);
        ps = dcs_fmaf(ps, s, DCS_S2);
        ps = dcs_fmaf(ps, s, DCS_S1);
        pc = dcs_fmaf(s, DCS_C4, DCS_C3);
        pc = dcs_fmaf(pc, s, DCS_C2);
        pc = dcs_fmaf(pc, s, DCS_C1);
    }
    pc = dcs_fmaf(pc, s, -0.5f);
    *fSinR = dcs_fmaf(s * r, ps, r);
    *fCosR = dcs_fmaf(s, pc, 1.0f);
    *uQ = q;
}

template <bool LOWDEG = false>
DCS_HD void dcs_sincos_fast(const float x, float *fSin, float *fCos)
{
    float sr, cr;
    uint32_t q;
    dcs_sincos_core<LOWDEG>(x, &sr, &cr, &q);
    // q mod 4:  0: (cr, sr)  1: (-sr, cr)  2: (-cr, -sr)  3: (sr, -cr)
    // sin's sign = bit 1 of q; cos's sign = bit 1 xor bit 0.
    const uint32_t t30 = q << 30, t31 = q << 31;           // bit 1 / bit 0 moved to the sign position
    const bool swap = (int32_t)t31 < 0;
    const uint32_t us = dcs_f32_bits(swap ? cr : sr);
    const uint32_t uc = dcs_f32_bits(swap ? -sr : cr);     // one v_cndmask with a neg modifier (sign flip == xor t31)
    *fSin = dcs_bits_f32(dcs_xor_sign_of(us, t30));
    *fCos = dcs_bits_f32(dcs_xor_sign_of(uc, t30));
}

// The quadrant on a PACKED pair p = {low half = cos r, high half = sin r}, n = q mod 4:
//   0: (cr, sr)   1: (-sr, cr)   2: (-cr, -sr)   3: (sr, -cr)      as (cos x, sin x)
// halves rotated by 16 * (n mod 2) with v_alignbit_b32; both sign bits from ONE 64-bit shift: the constant
// 0x00000000'80008000 shifted left by 16 * (n mod 4) -- the low six bits of the rotate amount, which the hardware's shifter
// takes as they are --, high word:   n = 0: 0    1: 0x00008000 (cos)    2: 0x80008000 (both)    3: 0x80000000 (sin)
// i.e. sin's sign = bit 1 of n, cos's = bit 1 of n + 1.  v_lshlrev_b32, v_alignbit_b32, v_lshlrev_b64, v_xor_b32: four
// operations where round 2 had six (two shifts and two v_bitop3_b32 for the signs): +5 % on the VALU-issue-bound b16
// generator (profiles/r03_fp16.md: 8.7 % fewer operations bought 5.0 %, so the 64-bit shift costs at most two issue slots).
DCS_HD uint32_t dcs_quadrant_half2(const uint32_t p, const uint32_t q)
{
    const uint32_t amt = q << 4; // bit 4 = n mod 2: rotate the halves by 16 when n is odd; bits 4-5 = n mod 4
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t sw = __builtin_amdgcn_alignbit(p, p, amt);
#else
    const uint32_t sw = (amt & 16u) ? ((p >> 16) | (p << 16)) : p;
#endif
    return sw ^ (uint32_t)((0x80008000ull << (amt & 63u)) >> 32);
}

// ---------------------------------------------------------------------------
// sin and cos of |x| < 32768 for a BINARY16 result, packed as the b16 output wants it:
// low half = cos (re), high half = sin (im) (reference: __floats2half2_rn(re, im),
// BeamformerKernels.cu:113,182).  Opt-in (dcs_bf_tuning::math_mode bit 2); the default b16
// path rounds the 1-ULP fp32 pair instead.
//
// Sized for an 11-bit significand: two-term Cody-Waite reduction (the dropped third term is
// < 4e-11 for |n| <= 20861), sin to r^5 and cos to r^4 in fp32 (minimax on [-pi/4, pi/4]:
// relative error 1.9e-6 / 1.5e-5, i.e. < 0.004 / 0.03 of a binary16 ulp), ONE conversion of the
// pair (v_cvt_pk_f16_f32, round to nearest even), then the quadrant on the packed word:
// halves rotated by 16 * (n mod 2) with v_alignbit_b32, sign of sin = bit 1 of n, sign of
// cos = bit 1 of n + 1, both from one v_lshlrev_b64 of a constant and applied by one v_xor_b32.
// 21 VALU operations per coefficient with the rotation's 6 (23 in round 2), against 28-30 for
// the fp32-then-round form.
// tests/test_numerics.py sweeps EVERY fp32 argument below 32768: each half is within one
// binary16 ulp of RN16 of the correctly rounded value (it IS that value for 99.8 % of them,
// 99.1 % in [1, 512)), and the count that differs from RN16(fp32 path) is reported.
// ---------------------------------------------------------------------------
#define DCS_HS1 -1.66633813956683030e-01f
#define DCS_HS2 8.16309870382948000e-03f
#define DCS_HC0 -4.99760307529649600e-01f
#define DCS_HC1 4.04579233108025100e-02f

// IEEE binary16 round-to-nearest-even of an fp32 (what v_cvt_pk_f16_f32 does per element in the
// default rounding mode).  The host form is bit arithmetic; the device form is the instruction.
DCS_HD uint32_t dcs_f32_to_f16_bits(const float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 h = (_Float16)x;
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
#else
    const uint32_t u = dcs_f32_bits(x), sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return sign | 0x7c00u | (a > 0x7f800000u ? 0x200u | ((a >> 13) & 0x3ffu) : 0u);
    if (a >= 0x477ff000u) return sign | 0x7c00u;
    if (a < 0x33000001u) return sign;
    const int32_t e = (int32_t)(a >> 23) - 127;
    const uint32_t m = (a & 0x7fffffu) | 0x800000u;
    const uint32_t shift = e < -14 ? (uint32_t)(-e - 14 + 13) : 13u, hexp = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return sign | (hexp == 0 ? q : (((hexp - 1u) << 10) + q));
#endif
}

DCS_HD uint32_t dcs_pack_half2(const float lo, const float hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef _Float16 dcs_halfx2 __attribute__((ext_vector_type(2)));
    dcs_halfx2 h;
    h.x = (_Float16)lo;
    h.y = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, h); // one v_cvt_pk_f16_f32
#else
    return dcs_f32_to_f16_bits(lo) | (dcs_f32_to_f16_bits(hi) << 16);
#endif
}

// {low half = cos x, high half = sin x}, |x| < 32768 (DCS_SINCOS_FAST_LIMIT).
DCS_HD uint32_t dcs_sincos_half2(const float x)
{
    const float nb = dcs_fmaf(x, DCS_TWO_OVER_PI, DCS_RINT_MAGIC);
    const float n = nb - DCS_RINT_MAGIC;
    const uint32_t q = dcs_f32_bits(nb);
    float r = dcs_fmaf(-n, DCS_PIO2_1, x); // exact
    r = dcs_fmaf(-n, DCS_PIO2_2, r);
    const float s = r * r;
    const float ps = dcs_fmaf(s, DCS_HS2, DCS_HS1);
    const float pc = dcs_fmaf(s, DCS_HC1, DCS_HC0);
    float sr = dcs_fmaf(s * r, ps, r);
    float cr = dcs_fmaf(s, pc, 1.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    // Keep the two fmas fp32 instructions and the conversion ONE v_cvt_pk_f16_f32.  Left alone, hipcc fuses each
    // fma with its conversion into v_fma_mixlo_f16 / v_fma_mixhi_f16, which (measured on MI355X) round the exact
    // fma result straight to binary16 -- 1.3e-4 of the arguments then differ from the fmaf-then-convert sequence
    // the host sweep proves -- and issue so slowly that the 22-instruction loop ran 4 % slower than the
    // 28-instruction fp32-then-round form (profiles/r02_fp16.md).
    asm("" : "+v"(sr), "+v"(cr));
#endif
    return dcs_quadrant_half2(dcs_pack_half2(cr, sr), q);
}

// The fp32-GRADE pair (dcs_sincos_fast: <= 1 ULP in fp32, three-term reduction, full or low-degree polynomials) delivered
// as the packed b16 word: converted ONCE, before the quadrant logic, which then works on the packed word (4 operations)
// instead of on two fp32 values (7 + the conversion).  Bit for bit RN-even(dcs_sincos_fast's fp32 pair): swapping halves
// and flipping signs commute with rounding to nearest even.  tests/test_numerics.py checks that identity for EVERY fp32
// argument of the fast range; the default b16 output (math_mode 0) is this function since round 3 (27 instead of 30
// vector operations per coefficient with the rotation's).
template <bool LOWDEG = false>
DCS_HD uint32_t dcs_sincos_fast_half2(const float x)
{
    float sr, cr;
    uint32_t q;
    dcs_sincos_core<LOWDEG>(x, &sr, &cr, &q);
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(sr), "+v"(cr)); // the conversion stays ONE v_cvt_pk_f16_f32 of the ROUNDED fp32 pair (no v_fma_mix*_f16: see dcs_sincos_half2)
#endif
    return dcs_quadrant_half2(dcs_pack_half2(cr, sr), q);
}

#endif // DCS_BF_MATH_H
