// bf_device.h -- device-side helpers shared by the gfx950 kernel files (bf_kernels.hip,
// bf_beamform_mfma.hip): vector types, the store wrapper, and one coefficient through the fast
// (bf_math.h, swept exhaustively) or slow (IEEE divide + fp64 sincos) path.  Internal.
#ifndef DCS_BF_DEVICE_H
#define DCS_BF_DEVICE_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "bf_math.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256; // 4 waves of 64

template <bool NT, typename T>
__device__ __forceinline__ void store_global(T *p, const T v)
{
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// (cos, sin) of one coefficient, fast path (bf_math.h; swept exhaustively).
//   DIV3   : 3-op divide by the launch constant (only when dcs_bf_create has
//            verified it exact for this D), else the 5-op form
//   LOWDEG : low-degree polynomials (only when every |fRotation| of the wave < 500)
template <bool DIV3, bool LOWDEG>
__device__ __forceinline__ void coeff_fast(const float fRate, const float fPhase0, const float fChan,
                                           const float D, const float y, float &re, float &im)
{
    const float rot = dcs_rotation<DIV3>(fRate, fPhase0, fChan, D, y);
#ifdef DCS_USE_OCML_SINCOS // A/B build only (tools/measure.py sincos with DCS_HIPCC_EXTRA=-DDCS_USE_OCML_SINCOS): __ocml_sincos_f32
    sincosf(rot, &im, &re);
#else
    dcs_sincos_fast<LOWDEG>(rot, &im, &re);
#endif
}

// Run `body(div3, lowdeg)` with the two compile-time switches chosen from
// wave-uniform run-time values (the branch sits outside the channel loop).
template <typename F>
__device__ __forceinline__ void dispatch_fast(const bool div3, const bool lowdeg, F &&body)
{
    if (div3) {
        if (lowdeg)
            body(std::true_type{}, std::true_type{});
        else
            body(std::true_type{}, std::false_type{});
    } else {
        if (lowdeg)
            body(std::false_type{}, std::true_type{});
        else
            body(std::false_type{}, std::false_type{});
    }
}

// Slow path: hardware-sequence IEEE divide and fp64 sincos rounded once to
// fp32, i.e. the verifier's own definition (BeamformerCoefficientTest.cu:327-328).
__device__ __forceinline__ void coeff_slow(const float fRate, const float fPhase0, const float fChan,
                                        const float D, float &re, float &im)
{
    const float rot = dcs_rotation_ieee(fRate, fPhase0, fChan, D);
    double s, c;
    sincos((double)rot, &s, &c);
    re = (float)c;
    im = (float)s;
}

__device__ __forceinline__ uint32_t pack_half2(const float re, const float im)
{
    // v_cvt_f16_f32 rounds to nearest even (default mode) == __floats2half2_rn
    // (reference BeamformerKernels.cu:113,182); .x = re (low half), .y = im.
    halfx2 h;
    h.x = (_Float16)re;
    h.y = (_Float16)im;
    return __builtin_bit_cast(uint32_t, h);
}

} // namespace

#endif // DCS_BF_DEVICE_H
