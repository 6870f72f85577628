// bf_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the steering-coefficient
// generator.  Written for 64-wide wavefronts; no other target is supported.
//
// What is computed (reference: beamformer_coefficient_generator/
// BeamformerCoefficientTest.cu:319-333, the CPU verifier -- NOT the reference's
// device kernels, whose fp32 shortcuts differ from it by ~1e-5):
//   for t, c, (a,b):  out[t][c][a][b] = (cos, sin)(fRotation(delay_vals[a][b], dt[t], c))
//
// Shape of the work: 8 bytes written per coefficient (4 for the fp16 form),
// 16 bytes read per (antenna,beam) per time step -> HBM-write bound.  The
// design follows from that:
//   * a lane owns PPL adjacent (antenna,beam) pairs (2 for fp32, 4 for fp16), so
//     every store is one 16-byte global_store_dwordx4 and a wave writes 1 KiB
//     contiguous per channel (whole 128-byte lines, no read-for-ownership);
//   * a workgroup (4 waves) owns `tiles_per_block` such 1-KiB tiles x a slab of
//     channels; the channel-independent terms of each pair (one fp64 chain) are
//     computed once per workgroup and staged in LDS, then held in 2*PPL VGPRs
//     for the whole channel walk;
//   * the per-channel work is ~35 fp32 VALU operations per coefficient (three
//     roundings of fDelayN with a 5-op exact divide, one add, a 25-op sincos),
//     no divide sequence, no transcendental unit, no double precision;
//   * 64-bit addressing throughout (16 GiB per time step at 64x1024x32768).
// Pairs whose arguments leave the fast path's proven range take a wave-uniform
// slow branch (IEEE divide + fp64 sincos) with the same rounding behaviour.

#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "bf_device.h"

namespace {

// ---------------------------------------------------------------------------
// Tiled generator.
//   OUT16   : packed half2 output (4 pairs per lane) instead of fp32 (2 pairs)
//   TPB     : 1-KiB tiles per workgroup (1, 2, 4); the 4 waves are arranged as
//             TPB tile columns x 4/TPB channel rows
//   NT      : nontemporal stores
//   ALIGNED : n_pairs % PPL == 0 -> 16-byte stores; otherwise per-pair stores
//   NOMATH  : addressing/stores only -- instantiated only in the probes build (-DDCS_PROBES,
//             probes/libdcs_probes.so: the store ceiling of this shape); never in libdcs_beamformer.so
//   TAG     : 0 production; 1 = the same code under another symbol, launched only by
//             dcs_bf_autotune, so that a profiler's per-kernel statistics of the
//             production launches are not mixed with the tuner's trial geometries
// ---------------------------------------------------------------------------
//   INL     : the kernel parameter is bf_tiled_args_inl (fDeltaTime of up to 256 time steps by value)
//   TERMS   : the pairs' channel-independent terms and classes come from the table bf_terms_kernel wrote
//             just before (a.terms / a.flags) instead of being computed and staged in LDS by every
//             workgroup: no fp64, no LDS, no barrier in this kernel.  The per-workgroup set-up is ~70
//             VALU issue slots per lane (the fp64 chain with its divide, the class bound) against the
//             6 (fp32) to 48 (fp16) coefficients a lane then produces on a short walk: 12 of the fp32
//             kernel's 41.6 lane-operations per coefficient.  Large launches take this variant.
//   HALF    : b16 output from the binary16-sized arithmetic (dcs_sincos_half2; math_mode bit 2) -- its own
//             instantiation, so that the fp32 sincos path costs it no registers (60 instead of 84 VGPRs: 8
//             instead of 5 waves per SIMD)
template <bool OUT16, int TPB, bool NT, bool ALIGNED, bool NOMATH, int TAG = 0, bool INL = false, bool TERMS = false, bool HALF = false>
__global__ void __launch_bounds__(kBlock) bf_tiled_kernel(const std::conditional_t<INL, bf_tiled_args_inl, bf_tiled_args> args)
{
    const bf_tiled_args &a = [&]() -> const bf_tiled_args & {
        if constexpr (INL)
            return args.a;
        else
            return args;
    }();
    constexpr int PPL = OUT16 ? 4 : 2;
    constexpr int TILE = 64 * PPL;
    constexpr int ROWS = 4 / TPB;
    constexpr uint32_t EB = OUT16 ? 4u : 8u; // bytes per coefficient
    constexpr int kSlowUnroll = OUT16 ? 1 : PPL; // unrolling of the slow path's pair loop (see there)

    // workgroup -> (tile group, channel block, time step); tile group fastest so
    // that concurrently resident workgroups cover one channel row end to end.
    uint32_t bid = blockIdx.x;
    if (a.xcd_remap) bid = (bid % 8u) * (gridDim.x / 8u) + bid / 8u; // gridDim.x % 8 == 0 (host)
    const uint32_t tg = bid % a.n_tile_groups;
    const uint32_t rest = bid / a.n_tile_groups;
    const uint32_t cb = rest % a.n_cblocks;
    const uint32_t t = rest / a.n_cblocks;
    const uint32_t pair_base = tg * (uint32_t)(TPB * TILE);

    // wave-uniform by construction; the b16 arithmetic form is told so: its channel walk's counter, bound test and fChan
    // then live in scalar registers -- 3 vector instructions per channel step less, 0.75 of 25 per coefficient, +1.3 % at
    // the best geometry and +2.6 % at the default one.  (The other forms measured no gain -- fp32 is store-bound -- or a
    // loss -- fp16 with the fp32-grade arithmetic, -2.8 % -- from the same change and keep the vector counter.)
    const uint32_t wave = HALF ? (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = wave % TPB;
    const uint32_t row = wave / TPB;
    const uint32_t li = tile * TILE + lane * PPL;
    const uint32_t p0 = pair_base + li;

    float fRate[PPL], fPhase0[PPL];
    bool wave_slow, wave_low;
    if constexpr (TERMS) {
        // ---- terms and classes from the pre-pass table ([t][pairs_pad]; pairs past n_pairs hold zeros)
        const uint32_t wave_p0 = pair_base + tile * (uint32_t)TILE; // wave-uniform
        uint32_t cls = DCS_CLASS_FAST_LOW;
        if (wave_p0 < a.pairs_pad) {
            const uint32_t *fl = a.flags + (uint64_t)t * (a.pairs_pad / 64u) + wave_p0 / 64u;
#pragma unroll
            for (int j = 0; j < TILE / 64; j++)
                if (wave_p0 + 64u * (uint32_t)j < a.pairs_pad) cls = max(cls, fl[j]);
        }
        cls = __builtin_amdgcn_readfirstlane(cls);
        wave_slow = false;
        wave_low = cls == DCS_CLASS_FAST_LOW;
        if (cls == DCS_CLASS_SLOW) return; // bf_terms_kernel has written these tiles itself (slow path)
        if (p0 >= a.n_pairs) return; // pairs_pad >= n_pairs: every load below is inside the table
        const floatx4 *tp = reinterpret_cast<const floatx4 *>(a.terms + 2u * ((uint64_t)t * a.pairs_pad + p0));
#pragma unroll
        for (int j = 0; j < PPL; j += 2) {
            const floatx4 v = tp[j / 2];
            fRate[j] = v.x;
            fPhase0[j] = v.y;
            fRate[j + 1] = v.z;
            fPhase0[j + 1] = v.w;
        }
    } else {
        __shared__ __attribute__((aligned(16))) float s_terms[TPB * TILE * 2]; // {fRateTerm, fPhase0}
        float dt;
        if constexpr (INL)
            dt = a.dt_dev ? a.dt_dev[t] : args.dt_inline[t];
        else
            dt = a.dt_dev ? a.dt_dev[t] : a.dt0;

        // ---- stage the channel-independent terms of this workgroup's pairs in LDS
        for (uint32_t i = threadIdx.x; i < (uint32_t)(TPB * TILE); i += kBlock) {
            const uint32_t p = pair_base + i;
            float fR = 0.0f, fP = 0.0f;
            if (p < a.n_pairs) {
                const floatx4 raw = *reinterpret_cast<const floatx4 *>(&a.delays[p]);
                dcs_delay_vals d;
                d.fDelay_s = raw.x;
                d.fDelayRate_sps = raw.y;
                d.fPhase_rad = raw.z;
                d.fPhaseRate_radps = raw.w;
                dcs_pair_terms(d, dt, a.k.dHalfChannels, a.k.dDenominator, &fR, &fP);
            }
            *reinterpret_cast<floatx2 *>(&s_terms[2 * i]) = floatx2{fR, fP};
        }
        __syncthreads();

        uint32_t cls = DCS_CLASS_FAST_LOW;
#pragma unroll
        for (int j = 0; j < PPL; j += 2) {
            const floatx4 v = *reinterpret_cast<const floatx4 *>(&s_terms[2 * (li + j)]);
            fRate[j] = v.x;
            fPhase0[j] = v.y;
            fRate[j + 1] = v.z;
            fPhase0[j + 1] = v.w;
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) cls = max(cls, dcs_pair_class(fRate[j], fPhase0[j], a.k.fRotBoundScale, a.k.fLowDegLimit));
        wave_slow = __builtin_amdgcn_ballot_w64(cls == DCS_CLASS_SLOW) != 0ull;
        wave_low = __builtin_amdgcn_ballot_w64(cls != DCS_CLASS_FAST_LOW) == 0ull;
    }

    const uint32_t cbeg = cb * a.chan_per_block;
    const uint32_t cend = min(cbeg + a.chan_per_block, a.nc);
    if (p0 >= a.n_pairs) return; // after the barrier; whole lane is past the table

    const float D = a.k.fDenominator, y = a.k.fRcpDenominator;
    const uint64_t row_bytes = (uint64_t)a.n_pairs * EB;
    char *dst = reinterpret_cast<char *>(a.out) +
                ((uint64_t)t * a.nc + (cbeg + row)) * row_bytes + (uint64_t)p0 * EB;
    const uint64_t step = (uint64_t)ROWS * row_bytes;

    auto emit = [&](const float (&re)[PPL], const float (&im)[PPL]) {
        if constexpr (ALIGNED) {
            if constexpr (OUT16) {
                uintx4 v;
                v.x = pack_half2(re[0], im[0]);
                v.y = pack_half2(re[1], im[1]);
                v.z = pack_half2(re[2], im[2]);
                v.w = pack_half2(re[3], im[3]);
                store_global<NT>(reinterpret_cast<uintx4 *>(dst), v);
            } else {
                store_global<NT>(reinterpret_cast<floatx4 *>(dst), floatx4{re[0], im[0], re[1], im[1]});
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if (p0 + j < a.n_pairs) {
                    if constexpr (OUT16)
                        store_global<NT>(reinterpret_cast<uint32_t *>(dst) + j, pack_half2(re[j], im[j]));
                    else
                        store_global<NT>(reinterpret_cast<floatx2 *>(dst) + j, floatx2{re[j], im[j]});
                }
            }
        }
    };

    if constexpr (NOMATH) {
        float re[PPL], im[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            re[j] = fRate[j];
            im[j] = fPhase0[j];
        }
        for (uint32_t c = cbeg + row; c < cend; c += ROWS) {
            emit(re, im);
            dst += step;
        }
        return;
    }

    static_assert(!HALF || OUT16, "the binary16-sized arithmetic is for b16 output");
    if (HALF && !wave_slow) {
        if constexpr (HALF) {
            // b16 output, no pair of the wave in the slow class (every |fRotation| < 32000), opted in (math_mode
            // bit 2): the binary16-sized sincos (bf_math.h: dcs_sincos_half2), which yields the packed (re, im) word
            auto walk = [&](auto div3) {
                for (uint32_t c = cbeg + row; c < cend; c += ROWS) {
                    const float fChan = (float)(a.c0 + c);
                    uint32_t w[PPL];
#pragma unroll
                    for (int j = 0; j < PPL; j++)
                        w[j] = dcs_sincos_half2(dcs_rotation<decltype(div3)::value>(fRate[j], fPhase0[j], fChan, D, y));
                    if constexpr (ALIGNED) {
                        store_global<NT>(reinterpret_cast<uintx4 *>(dst), uintx4{w[0], w[1], w[2], w[3]});
                    } else {
#pragma unroll
                        for (int j = 0; j < PPL; j++)
                            if (p0 + j < a.n_pairs) store_global<NT>(reinterpret_cast<uint32_t *>(dst) + j, w[j]);
                    }
                    dst += step;
                }
            };
            if (a.k.uDiv3Exact != 0u)
                walk(std::true_type{});
            else
                walk(std::false_type{});
        }
    } else if (!HALF && !wave_slow) {
        dispatch_fast(a.k.uDiv3Exact != 0u, wave_low, [&](auto div3, auto lowdeg) {
            if constexpr (OUT16) {
                // b16 output from the fp32-grade arithmetic (the default): the pair is converted once and the quadrant
                // logic runs on the packed word (dcs_sincos_fast_half2: bit for bit RN-even of the fp32 pair)
                for (uint32_t c = cbeg + row; c < cend; c += ROWS) {
                    const float fChan = (float)(a.c0 + c);
                    uint32_t w[PPL];
#pragma unroll
                    for (int j = 0; j < PPL; j++)
                        w[j] = dcs_sincos_fast_half2<decltype(lowdeg)::value>(dcs_rotation<decltype(div3)::value>(fRate[j], fPhase0[j], fChan, D, y));
                    if constexpr (ALIGNED) {
                        store_global<NT>(reinterpret_cast<uintx4 *>(dst), uintx4{w[0], w[1], w[2], w[3]});
                    } else {
#pragma unroll
                        for (int j = 0; j < PPL; j++)
                            if (p0 + j < a.n_pairs) store_global<NT>(reinterpret_cast<uint32_t *>(dst) + j, w[j]);
                    }
                    dst += step;
                }
                return;
            }
#pragma unroll 2
            for (uint32_t c = cbeg + row; c < cend; c += ROWS) {
                const float fChan = (float)(a.c0 + c);
                float re[PPL], im[PPL];
#pragma unroll
                for (int j = 0; j < PPL; j++)
                    coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(fRate[j], fPhase0[j], fChan, D, y, re[j], im[j]);
#ifdef DCS_PROBES
                for (uint32_t k = 0; k < a.pace; k++) __builtin_amdgcn_s_sleep(1); // probes build only: 64-cycle units before each store
#endif
                emit(re, im);
                dst += step;
            }
        });
    } else if constexpr (!TERMS) {
        for (uint32_t c = cbeg + row; c < cend; c += ROWS) {
            const float fChan = (float)(a.c0 + c);
            float re[PPL], im[PPL];
            // fp32 (2 pairs per lane): unrolled, so that re[] / im[] stay in registers -- a rolled loop
            // indexes them dynamically and can put them in scratch (12-20 B per lane: +130-230 us on the
            // first launch of a process).  fp16 (4 pairs): rolled, four inlined fp64 sincos would cost
            // the fast loop registers; it has never needed scratch (checked: -Rpass-analysis).
#pragma unroll kSlowUnroll
            for (int j = 0; j < PPL; j++) coeff_slow(fRate[j], fPhase0[j], fChan, D, re[j], im[j]);
            emit(re, im);
            dst += step;
        }
    }
}

// ---------------------------------------------------------------------------
// Row-streaming generator ("rows" form): the form the HBM system likes best.
//
// tools/measure.py stores --kind pattern (profiles/r01_store_patterns.md) shows that the
// write rate MI355X sustains depends on how long a wave keeps storing: waves
// that issue 1 / 2 / 4 stores and retire, dispatched in address order, reach
// 7.1 / 6.9 / 6.6 TB/s; waves that walk hundreds of rows reach 5.5 TB/s.  So
// here a wave owns ONE 1-KiB tile of RPW consecutive channel rows and retires;
// workgroups are numbered column-fastest so the dispatcher itself streams the
// tensor in address order.  The channel-independent terms of every pair come
// from a small table ({fRateTerm, fPhase0} per pair and time step, written by
// bf_terms_kernel just before; L2-resident: 8 B per pair against 8*C B of
// output), one 16-byte load per lane.
//   NW   : waves per workgroup = adjacent 1-KiB tiles of one row (4, 8, 16)
//   RPW  : consecutive channel rows per wave (1, 2, 4)
// ---------------------------------------------------------------------------
template <bool OUT16, int NW, int RPW, bool NT, bool ALIGNED, bool NOMATH>
__global__ void __launch_bounds__(NW * 64) bf_rows_kernel(const bf_rows_args a)
{
    constexpr int PPL = OUT16 ? 4 : 2;
    constexpr int TILE = 64 * PPL;
    constexpr uint32_t EB = OUT16 ? 4u : 8u;

    uint32_t b = blockIdx.x;
    if (a.xcd_remap) b = (b % 8u) * (gridDim.x / 8u) + b / 8u; // gridDim.x % 8 == 0 (host)
    const uint32_t cg = b % a.n_colgroups;
    const uint32_t rg = b / a.n_colgroups;
    const uint32_t t = rg / a.n_rowgroups;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // same_tile == 0: the NW waves take NW adjacent tiles of RPW consecutive rows;
    // same_tile == 1: they share ONE tile and interleave NW*RPW rows (wave w: rows w, w+NW, ...)
    const uint32_t row_step = a.same_tile ? (uint32_t)NW : 1u;
    const uint32_t c = (rg - t * a.n_rowgroups) * (uint32_t)RPW * row_step + (a.same_tile ? wave : 0u); // channel within the slab
    const uint32_t chunk = a.same_tile ? cg : cg * (uint32_t)NW + wave;
    if (chunk * (uint32_t)TILE >= a.n_pairs) return; // wave-uniform
    const uint32_t p0 = chunk * (uint32_t)TILE + lane * PPL;

    // slow-path flags of this wave's pairs: one dword per 64 pairs (scalar loads)
    const uint32_t *fl = a.flags + (uint64_t)t * (a.pairs_pad / 64u) + chunk * (uint32_t)(TILE / 64);
    uint32_t cls = DCS_CLASS_FAST_LOW; // max class over this wave's pairs
#pragma unroll
    for (int j = 0; j < TILE / 64; j++) cls = max(cls, fl[j]);

    float fRate[PPL], fPhase0[PPL];
    const floatx4 *tp = reinterpret_cast<const floatx4 *>(a.terms + 2u * ((uint64_t)t * a.pairs_pad + p0));
#pragma unroll
    for (int j = 0; j < PPL; j += 2) {
        const floatx4 v = tp[j / 2];
        fRate[j] = v.x;
        fPhase0[j] = v.y;
        fRate[j + 1] = v.z;
        fPhase0[j + 1] = v.w;
    }
    if (p0 >= a.n_pairs) return;

    const float D = a.D, y = a.y;
    const uint64_t row_bytes = (uint64_t)a.n_pairs * EB;
    char *dst = reinterpret_cast<char *>(a.out) + ((uint64_t)t * a.nc + c) * row_bytes + (uint64_t)p0 * EB;

    auto emit = [&](const float (&re)[PPL], const float (&im)[PPL]) {
        if constexpr (ALIGNED) {
            if constexpr (OUT16) {
                uintx4 v;
                v.x = pack_half2(re[0], im[0]);
                v.y = pack_half2(re[1], im[1]);
                v.z = pack_half2(re[2], im[2]);
                v.w = pack_half2(re[3], im[3]);
                store_global<NT>(reinterpret_cast<uintx4 *>(dst), v);
            } else {
                store_global<NT>(reinterpret_cast<floatx4 *>(dst), floatx4{re[0], im[0], re[1], im[1]});
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if (p0 + j < a.n_pairs) {
                    if constexpr (OUT16)
                        store_global<NT>(reinterpret_cast<uint32_t *>(dst) + j, pack_half2(re[j], im[j]));
                    else
                        store_global<NT>(reinterpret_cast<floatx2 *>(dst) + j, floatx2{re[j], im[j]});
                }
            }
        }
    };

    if (__builtin_expect(cls != DCS_CLASS_SLOW, 1)) {
        dispatch_fast(a.div3 != 0u, cls == DCS_CLASS_FAST_LOW, [&](auto div3, auto lowdeg) {
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                if (c + r * row_step < a.nc) {
                    const float fChan = (float)(a.c0 + c + r * row_step);
                    float re[PPL], im[PPL];
#pragma unroll
                    for (int j = 0; j < PPL; j++) {
                        if constexpr (NOMATH) {
                            re[j] = fRate[j];
                            im[j] = fPhase0[j] + fChan;
                        } else {
                            coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(fRate[j], fPhase0[j], fChan, D, y, re[j], im[j]);
                        }
                    }
#ifdef DCS_PROBES
                    for (uint32_t k = 0; k < a.pace; k++) __builtin_amdgcn_s_sleep(1); // probes build only, as in the tiled form
#endif
                    emit(re, im);
                    dst += row_bytes * row_step;
                }
            }
        });
    } else {
        for (int r = 0; r < RPW; r++) {
            if (c + r * row_step < a.nc) {
                const float fChan = (float)(a.c0 + c + r * row_step);
                float re[PPL], im[PPL];
#pragma unroll 1
                for (int j = 0; j < PPL; j++) coeff_slow(fRate[j], fPhase0[j], fChan, D, re[j], im[j]);
                emit(re, im);
                dst += row_bytes * row_step;
            }
        }
    }
}

// terms[t][p] = {fRateTerm, fPhase0}; flags[t][p/64] = the highest dcs_pair_class of
// those 64 pairs.  p runs to pairs_pad (a multiple of 256);
// pairs past n_pairs get zeros.  One lane per (t, p); 64 lanes = one flag word.
//
// With a.out set (the tiled form's terms-table variant) this kernel also OWNS the slow path: a workgroup
// that holds a slow-class pair (|fRotation| may reach 32000, or a rate term outside the constant divide's
// range) writes all channels of its 256 pairs itself -- IEEE divide, fp64 sincos, one pair per lane -- and
// marks its four flag words slow, so that the main kernel skips those tiles and carries no fp64 sincos at
// all (its register count falls from 70-84 to 30-51: 8 waves per SIMD instead of 5-6).  Such inputs are
// pathological (a delay rate 10^3 beyond the reference's); their tiles take milliseconds here.
__global__ void __launch_bounds__(kBlock) bf_terms_kernel(const bf_terms_args a)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x; // < pairs_pad (grid exact)
    const uint32_t t = blockIdx.y;
    const float dt = a.dt_dev ? a.dt_dev[t] : a.dt_inline[t];
    float fRate = 0.0f, fPhase0 = 0.0f;
    if (p < a.n_pairs) {
        const floatx4 raw = *reinterpret_cast<const floatx4 *>(&a.delays[p]);
        dcs_delay_vals d;
        d.fDelay_s = raw.x;
        d.fDelayRate_sps = raw.y;
        d.fPhase_rad = raw.z;
        d.fPhaseRate_radps = raw.w;
        dcs_pair_terms(d, dt, a.k.dHalfChannels, a.k.dDenominator, &fRate, &fPhase0);
    }
    const uint32_t cls = dcs_pair_class(fRate, fPhase0, a.k.fRotBoundScale, a.k.fLowDegLimit);
    uint32_t wave_cls = __builtin_amdgcn_ballot_w64(cls == DCS_CLASS_SLOW) != 0ull
                            ? DCS_CLASS_SLOW
                            : (__builtin_amdgcn_ballot_w64(cls == DCS_CLASS_FAST_HIGH) != 0ull ? DCS_CLASS_FAST_HIGH
                                                                                              : DCS_CLASS_FAST_LOW);
    if (a.out != nullptr) {
        if (__syncthreads_or((int)(cls == DCS_CLASS_SLOW))) { // workgroup-uniform
            wave_cls = DCS_CLASS_SLOW;
            if (p < a.n_pairs) {
                const uint64_t eb = a.out16 ? 4u : 8u;
                char *dst = reinterpret_cast<char *>(a.out) + ((uint64_t)t * a.nc * a.n_pairs + p) * eb;
                for (uint32_t c = 0; c < a.nc; c++) {
                    float re, im;
                    coeff_slow(fRate, fPhase0, (float)(a.c0 + c), a.k.fDenominator, re, im);
                    if (a.out16)
                        *reinterpret_cast<uint32_t *>(dst) = pack_half2(re, im);
                    else
                        *reinterpret_cast<floatx2 *>(dst) = floatx2{re, im};
                    dst += (uint64_t)a.n_pairs * eb;
                }
            }
        }
    }
    *reinterpret_cast<floatx2 *>(a.terms + 2u * ((uint64_t)t * a.pairs_pad + p)) = floatx2{fRate, fPhase0};
    if ((threadIdx.x & 63u) == 0u) a.flags[(uint64_t)t * (a.pairs_pad / 64u) + p / 64u] = wave_cls;
}

// ---------------------------------------------------------------------------
// Fused coefficient generation + beamforming (reference
// calculate_beamweights_and_beamform_single_channel, BeamformerKernels.cu:192-367,
// verified by BeamformerCoefficientTest.cu:363-414):
//   beams[c][t/16][b][t%16] = sum over antennas a, IN ANTENNA ORDER, of
//       ( cos(rot) * sample.re , sin(rot) * sample.im )        -- an element-wise
// product, as the reference computes it (:315-316, verifier :391-392), with the
// table indexed [b*A + a].  No coefficient is ever written to HBM: per output
// 8 bytes leave the chip for 2*A sincos evaluations, so this kernel is bound by
// the fp32 VALU rate, not by HBM.
//
// A lane owns one (time, beam) output and walks the antennas sequentially (the
// verifier's summation order, separate multiply and add -- no fma), so the only
// difference from the verifier is the <= 1 ULP of each coefficient.  The
// channel-independent terms come from the table bf_bform_terms_kernel writes,
// laid out [t][a][b] so that a wave's 16 beams x 4 times read four 128-byte
// lines per antenna (L2-resident, reused by every channel); the int8 samples of
// one (channel, 16 times) block are staged in LDS once per channel.
// ---------------------------------------------------------------------------
template <bool INL> // INL: fDeltaTime of up to kDtInline time steps by value, behind the arguments proper
__global__ void __launch_bounds__(kBlock) bf_bform_terms_kernel(const std::conditional_t<INL, bf_bform_terms_args_inl, bf_bform_terms_args> args)
{
    const bf_bform_terms_args &a = [&]() -> const bf_bform_terms_args & {
        if constexpr (INL)
            return args.a;
        else
            return args;
    }();
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x; // table index b*A + a
    const uint32_t t = blockIdx.y;
    uint32_t cls = DCS_CLASS_FAST_LOW;
    if (p < a.n_pairs) {
        const floatx4 raw = *reinterpret_cast<const floatx4 *>(&a.delays[p]);
        dcs_delay_vals d;
        d.fDelay_s = raw.x;
        d.fDelayRate_sps = raw.y;
        d.fPhase_rad = raw.z;
        d.fPhaseRate_radps = raw.w;
        float fRate, fPhase0;
        float dt;
        if constexpr (INL)
            dt = args.dt_inline[t];
        else
            dt = a.dt_dev ? a.dt_dev[t] : a.dt0;
        dcs_pair_terms(d, dt, a.k.dHalfChannels, a.k.dDenominator, &fRate, &fPhase0);
        cls = dcs_pair_class(fRate, fPhase0, a.k.fRotBoundScale, a.k.fLowDegLimit);
        const uint32_t b = p / a.A, ant = p - b * a.A;
        *reinterpret_cast<floatx2 *>(a.terms + 2u * ((uint64_t)t * a.n_pairs + (uint64_t)ant * a.B + b)) =
            floatx2{fRate, fPhase0};
    }
    // flags[t] = (epoch << 2) | highest class at time t: the caller numbers its calls, so a word left by an earlier call
    // (a lower epoch) reads as "nothing above the lowest class yet" and nobody has to zero the words in between
    if (cls != DCS_CLASS_FAST_LOW) atomicMax(&a.flags[t], (a.epoch << 2) | cls);
}

// CH channels per pass: a lane's terms load and the LDS sample reads are shared by CH
// independent coefficient chains (more ILP, fewer loads per product).
constexpr uint32_t kAntChunk = 128; // antennas staged in LDS at a time (as fp32: 16 KiB per channel)
template <int CH>
__global__ void __launch_bounds__(kBlock) bf_beamform_kernel(const bf_beamform_args a)
{
    extern __shared__ __attribute__((aligned(16))) float s_ant[]; // [CH][A][16][2], int8 samples converted once

    const uint32_t bid = blockIdx.x;
    const uint32_t bg = bid % a.n_bgroups;
    const uint32_t rest = bid / a.n_bgroups;
    const uint32_t cb = rest % a.n_cblocks;
    const uint32_t tex = rest / a.n_cblocks; // 16-sample block within this launch

    const uint32_t b_local = threadIdx.x & 15u, t_in = threadIdx.x >> 4;
    const uint32_t b = bg * 16u + b_local;
    const uint32_t t = tex * 16u + t_in; // time index within this launch's terms table
    const bool live = b < a.B;

    // highest pair class over these 16 time steps (bf_bform_terms_kernel)
    const uint32_t fw = a.flags[tex * 16u + (threadIdx.x & 15u)];
    const uint32_t fl = (fw >> 2) == a.epoch ? (fw & 3u) : DCS_CLASS_FAST_LOW; // bf_bform_terms_kernel's epoch-tagged word
    const int slow = __syncthreads_or((int)(fl == DCS_CLASS_SLOW));
    const int high = __syncthreads_or((int)(fl != DCS_CLASS_FAST_LOW));

    const float D = a.k.fDenominator, y = a.k.fRcpDenominator;
    const float *tp = a.terms + 2u * ((uint64_t)t * a.A * a.B + (live ? b : 0u));
    const uint32_t cbeg = cb * a.chan_per_block;
    const uint32_t cend = min(cbeg + a.chan_per_block, a.C);
    const uint32_t tex_g = a.tex0 + tex; // 16-sample block within the whole tensor
    const uint32_t words = a.A * 8u;     // dwords of one [A][16][2] int8 block
    const uint32_t sa = min(kAntChunk, a.A); // antennas per staged chunk = stride of a channel's LDS region

    for (uint32_t c = cbeg; c < cend; c += CH) {
        float fChan[CH], acc_re[CH], acc_im[CH];
#pragma unroll
        for (int h = 0; h < CH; h++) {
            fChan[h] = (float)(c + h);
            acc_re[h] = 0.0f;
            acc_im[h] = 0.0f;
        }
        // antennas in chunks of kAntChunk (the LDS staging buffer); the running sums carry
        // across chunks, so the summation order stays the verifier's (a = 0, 1, 2, ...)
        for (uint32_t a0 = 0; a0 < a.A; a0 += kAntChunk) {
            const uint32_t na = min(kAntChunk, a.A - a0);
            const uint32_t cw = na * 8u; // dwords of this chunk's [na][16][2] int8 block
            __syncthreads();             // previous chunk's readers are done
#pragma unroll
            for (int h = 0; h < CH; h++) {
                if (c + h < cend) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ant) +
                                          ((uint64_t)(c + h) * a.nt16_total + tex_g) * words + (uint64_t)a0 * 8u;
                    for (uint32_t i = threadIdx.x; i < cw; i += kBlock) {
                        const uint32_t w = src[i]; // {re, im, re, im} of two consecutive (antenna, time) samples
                        const floatx4 f = {(float)(int8_t)(w & 0xffu), (float)(int8_t)((w >> 8) & 0xffu),
                                           (float)(int8_t)((w >> 16) & 0xffu), (float)(int8_t)(w >> 24)};
                        *reinterpret_cast<floatx4 *>(&s_ant[((size_t)h * sa * 8u + i) * 4u]) = f;
                    }
                }
            }
            __syncthreads();

            auto sample = [&](int h, uint32_t al, float &sre, float &sim) {
                const floatx2 v = *reinterpret_cast<const floatx2 *>(&s_ant[(((size_t)h * sa + al) * 16u + t_in) * 2u]);
                sre = v.x;
                sim = v.y;
            };
            if (!slow) {
                dispatch_fast(a.k.uDiv3Exact != 0u, !high, [&](auto div3, auto lowdeg) {
                    // terms of antenna al+2 are requested while al is computed (L2 latency >> one step)
                    auto terms_of = [&](uint32_t al) {
                        return *reinterpret_cast<const floatx2 *>(tp + 2u * (uint64_t)(a0 + min(al, na - 1u)) * a.B);
                    };
                    auto products = [&](uint32_t al, const floatx2 kp) {
#pragma unroll
                        for (int h = 0; h < CH; h++) {
                            float re, im, sre, sim;
                            coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(kp.x, kp.y, fChan[h], D, y, re, im);
                            sample(h, al, sre, sim);
                            const float pr = re * sre, pi = im * sim; // product, then sum: two roundings each
                            acc_re[h] = acc_re[h] + pr;
                            acc_im[h] = acc_im[h] + pi;
                        }
                    };
                    // three registers in rotation: step al uses one while al+2 is loaded into the one
                    // step al-1 has just finished with
                    floatx2 qa = terms_of(0), qb = terms_of(1), qc;
                    uint32_t al = 0;
                    for (; al + 2 < na; al += 3) {
                        qc = terms_of(al + 2);
                        products(al, qa);
                        qa = terms_of(al + 3);
                        products(al + 1, qb);
                        qb = terms_of(al + 4);
                        products(al + 2, qc);
                    }
                    if (al < na) products(al, qa);
                    if (al + 1 < na) products(al + 1, qb);
                });
            } else {
                // channel outermost and unrolled (h is a compile-time index: the accumulators stay in
                // registers, nothing goes to scratch), antennas in order inside -- the same sums
#pragma unroll
                for (int h = 0; h < CH; h++) {
                    float are = acc_re[h], aim = acc_im[h];
                    for (uint32_t al = 0; al < na; al++) {
                        const floatx2 kp = *reinterpret_cast<const floatx2 *>(tp + 2u * (uint64_t)(a0 + al) * a.B);
                        float re, im, sre, sim;
                        coeff_slow(kp.x, kp.y, fChan[h], D, re, im);
                        sample(h, al, sre, sim);
                        const float pr = re * sre, pi = im * sim;
                        are = are + pr;
                        aim = aim + pi;
                    }
                    acc_re[h] = are;
                    acc_im[h] = aim;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int h = 0; h < CH; h++) {
                if (c + h < cend) {
                    floatx2 *dst = reinterpret_cast<floatx2 *>(a.beams) +
                                   (((uint64_t)(c + h) * a.nt16_total + tex_g) * a.B + b) * 16u + t_in;
                    *dst = floatx2{acc_re[h], acc_im[h]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// One coefficient per lane (the launch shape of the reference's
// calculate_beamweights_naive, BeamformerKernels.cu:7-52): every lane redoes the
// per-pair terms.  grid = (ceil(n_pairs/256), min(nc, 65535)).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) bf_naive_kernel(const bf_naive_args a)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.n_pairs) return;
    const floatx4 raw = *reinterpret_cast<const floatx4 *>(&a.delays[p]);
    dcs_delay_vals d;
    d.fDelay_s = raw.x;
    d.fDelayRate_sps = raw.y;
    d.fPhase_rad = raw.z;
    d.fPhaseRate_radps = raw.w;
    float fRate, fPhase0;
    dcs_pair_terms(d, a.dt, a.k.dHalfChannels, a.k.dDenominator, &fRate, &fPhase0);
    const uint32_t cls = dcs_pair_class(fRate, fPhase0, a.k.fRotBoundScale, a.k.fLowDegLimit);
    for (uint32_t c = blockIdx.y; c < a.nc; c += gridDim.y) {
        const float fChan = (float)(a.c0 + c);
        float re, im;
        if (cls == DCS_CLASS_SLOW)
            coeff_slow(fRate, fPhase0, fChan, a.k.fDenominator, re, im);
        else if (cls == DCS_CLASS_FAST_LOW)
            coeff_fast<false, true>(fRate, fPhase0, fChan, a.k.fDenominator, a.k.fRcpDenominator, re, im);
        else
            coeff_fast<false, false>(fRate, fPhase0, fChan, a.k.fDenominator, a.k.fRcpDenominator, re, im);
        floatx2 *dst = reinterpret_cast<floatx2 *>(a.out) + ((uint64_t)c * a.n_pairs + p);
        *dst = floatx2{re, im};
    }
}

__global__ void __launch_bounds__(kBlock) bf_gather_beams_kernel(dcs_delay_vals *local,
                                                                 const dcs_delay_vals *global,
                                                                 uint32_t n_ant, uint32_t nb_local,
                                                                 uint32_t nb_total, uint32_t beam_offset)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_ant * nb_local) return;
    const uint32_t ant = i / nb_local, b = i - ant * nb_local;
    const floatx4 v = *reinterpret_cast<const floatx4 *>(&global[(uint64_t)ant * nb_total + beam_offset + b]);
    *reinterpret_cast<floatx4 *>(&local[i]) = v;
}

// Which instantiations exist: {plain, tuner-tagged, inline-dt, terms-table, terms-table tuner-tagged} x
// {fp32-sincos arithmetic, binary16-sized arithmetic (b16 output only)}; the terms-table and the binary16 forms
// with nontemporal stores only (the default policy); the store-only skeleton in the probes build only.
template <bool OUT16, int TPB, bool NT, bool ALIGNED, bool HALF>
const void *tiled_fn_h(bool nomath, bool tuner, bool inl, bool terms)
{
    if (terms) {
        if constexpr (NT) {
            if (inl || nomath) return nullptr;
            return tuner ? reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, false, 1, false, true, HALF>)
                         : reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, false, 0, false, true, HALF>);
        } else {
            return nullptr;
        }
    }
    // (inline-dt launches have no tuner-tagged twin: the tuner's multi-time-step trials run under the production symbol)
    if (inl) return nomath ? nullptr : reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, false, 0, true, false, HALF>);
    if (tuner) return reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, false, 1, false, false, HALF>);
    if constexpr (!HALF) {
#ifdef DCS_PROBES
        if (nomath) return reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, true>);
#endif
    }
    if (nomath) return nullptr; // the store-only skeleton exists in the probes build only
    return reinterpret_cast<const void *>(&bf_tiled_kernel<OUT16, TPB, NT, ALIGNED, false, 0, false, false, HALF>);
}

template <bool OUT16, int TPB, bool NT, bool ALIGNED>
const void *tiled_fn_nm(bool nomath, bool tuner, bool inl, bool terms, bool half)
{
    if (half) {
        if constexpr (OUT16 && NT)
            return tiled_fn_h<OUT16, TPB, NT, ALIGNED, true>(nomath, tuner, inl, terms);
        else
            return nullptr;
    }
    return tiled_fn_h<OUT16, TPB, NT, ALIGNED, false>(nomath, tuner, inl, terms);
}

template <bool OUT16, int TPB>
const void *tiled_fn_t(bool nt, bool aligned, bool nomath, bool tuner, bool inl, bool terms, bool half)
{
    if (nt) return aligned ? tiled_fn_nm<OUT16, TPB, true, true>(nomath, tuner, inl, terms, half) : tiled_fn_nm<OUT16, TPB, true, false>(nomath, tuner, inl, terms, half);
    return aligned ? tiled_fn_nm<OUT16, TPB, false, true>(nomath, tuner, inl, terms, half) : tiled_fn_nm<OUT16, TPB, false, false>(nomath, tuner, inl, terms, half);
}

template <bool OUT16>
const void *tiled_fn_o(int tpb, bool nt, bool aligned, bool nomath, bool tuner, bool inl, bool terms, bool half)
{
    switch (tpb) {
    case 1: return tiled_fn_t<OUT16, 1>(nt, aligned, nomath, tuner, inl, terms, half);
    case 2: return tiled_fn_t<OUT16, 2>(nt, aligned, nomath, tuner, inl, terms, half);
    case 4: return tiled_fn_t<OUT16, 4>(nt, aligned, nomath, tuner, inl, terms, half);
    default: return nullptr;
    }
}

} // namespace

hipError_t bf_prepare_tiled(const bf_tiled_args &a_in, const float *dt_inline, bool out16, int tiles_per_block,
                            bool nontemporal, bf_kernel_launch *out)
{
    bf_tiled_args &a = out->args.a;
    a = a_in;
    out->func = nullptr;
    if (a.n_pairs == 0 || a.nc == 0 || a.nt == 0) return hipSuccess; // nothing to launch
    if (a.chan_per_block == 0) return hipErrorInvalidValue;
    const bool nomath = (tiles_per_block & 0x100) != 0; // probe flag, see bf_capi.hip
    const bool tuner = (tiles_per_block & 0x200) != 0;  // dcs_bf_autotune's trial launches
    tiles_per_block &= 0xff;
    const uint32_t ppl = out16 ? 4u : 2u;
    const uint32_t pairs_per_block = 64u * ppl * (uint32_t)tiles_per_block;
    a.n_tile_groups = (a.n_pairs + pairs_per_block - 1) / pairs_per_block;
    a.n_cblocks = (a.nc + a.chan_per_block - 1) / a.chan_per_block;
    const uint64_t blocks = (uint64_t)a.n_tile_groups * a.n_cblocks * a.nt;
    if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const bool terms = a.terms != nullptr; // the terms-table variant: fDeltaTime went into bf_launch_terms instead
    if (terms && (a.flags == nullptr || a.pairs_pad < a.n_pairs || (a.pairs_pad % 256u))) return hipErrorInvalidValue;
    const bool inl = !terms && a.dt_dev == nullptr && a.nt > 1;
    if (inl && (dt_inline == nullptr || a.nt > kDtInline)) return hipErrorInvalidValue;
    if (inl) std::memcpy(out->args.dt_inline, dt_inline, (size_t)a.nt * sizeof(float));
    if (blocks % 8u) a.xcd_remap = 0; // the renumbering is a bijection only then
    const bool aligned = (a.n_pairs % ppl) == 0 && (reinterpret_cast<uintptr_t>(a.out) % 16u) == 0;
    const bool half = out16 && a.k.uHalfMath != 0u;
    const void *fn = out16 ? tiled_fn_o<true>(tiles_per_block, nontemporal, aligned, nomath, tuner, inl, terms, half)
                           : tiled_fn_o<false>(tiles_per_block, nontemporal, aligned, nomath, tuner, inl, terms, false);
    if (!fn) return hipErrorInvalidValue;
    out->func = fn;
    out->grid = dim3((uint32_t)blocks);
    out->block = dim3(kBlock);
    out->shared = 0;
    return hipSuccess;
}

hipError_t bf_launch_naive(const bf_naive_args &a, hipStream_t stream)
{
    if (a.n_pairs == 0 || a.nc == 0) return hipSuccess;
    const dim3 grid((a.n_pairs + kBlock - 1) / kBlock, a.nc < 65535u ? a.nc : 65535u);
    hipLaunchKernelGGL(bf_naive_kernel, grid, dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t bf_prepare_gather_beams(dcs_delay_vals *local, const dcs_delay_vals *global, uint32_t n_ant, uint32_t n_beams_local,
                                   uint32_t n_beams_total, uint32_t beam_offset, bf_gather_launch *out)
{
    out->func = nullptr;
    out->local = local;
    out->global = global;
    out->n_ant = n_ant;
    out->nb_local = n_beams_local;
    out->nb_total = n_beams_total;
    out->beam_offset = beam_offset;
    const uint64_t n = (uint64_t)n_ant * n_beams_local;
    if (n == 0) return hipSuccess;
    if (n > 0x7fffffffull) return hipErrorInvalidValue;
    out->func = reinterpret_cast<const void *>(&bf_gather_beams_kernel);
    out->grid = dim3((uint32_t)((n + kBlock - 1) / kBlock));
    out->block = dim3(kBlock);
    return hipSuccess;
}

hipError_t bf_launch_gather_beams(dcs_delay_vals *local, const dcs_delay_vals *global, uint32_t n_ant,
                                  uint32_t n_beams_local, uint32_t n_beams_total, uint32_t beam_offset,
                                  hipStream_t stream)
{
    bf_gather_launch g;
    const hipError_t e = bf_prepare_gather_beams(local, global, n_ant, n_beams_local, n_beams_total, beam_offset, &g);
    if (e != hipSuccess || g.func == nullptr) return e;
    void *params[] = {&g.local, &g.global, &g.n_ant, &g.nb_local, &g.nb_total, &g.beam_offset};
    return hipLaunchKernel(g.func, g.grid, g.block, params, 0, stream);
}

namespace {

template <bool OUT16, int NW, int RPW>
hipError_t launch_rows_t(const bf_rows_args &a, bool nt, bool aligned, bool nomath, dim3 grid, hipStream_t stream)
{
#define DCS_ROWS_LAUNCH(NTV, ALV, NMV)                                                                       \
    hipLaunchKernelGGL((bf_rows_kernel<OUT16, NW, RPW, NTV, ALV, NMV>), grid, dim3(NW * 64), a.lds_pad, stream, a)
    if (nomath) {
#ifdef DCS_PROBES
        if (nt) { if (aligned) DCS_ROWS_LAUNCH(true, true, true); else DCS_ROWS_LAUNCH(true, false, true); }
        else    { if (aligned) DCS_ROWS_LAUNCH(false, true, true); else DCS_ROWS_LAUNCH(false, false, true); }
#else
        return hipErrorInvalidValue; // the store-only skeleton exists in the probes build only
#endif
    } else {
        if (nt) { if (aligned) DCS_ROWS_LAUNCH(true, true, false); else DCS_ROWS_LAUNCH(true, false, false); }
        else    { if (aligned) DCS_ROWS_LAUNCH(false, true, false); else DCS_ROWS_LAUNCH(false, false, false); }
    }
#undef DCS_ROWS_LAUNCH
    return hipGetLastError();
}

template <bool OUT16, int NW>
hipError_t launch_rows_w(const bf_rows_args &a, int rpw, bool nt, bool aligned, bool nomath, dim3 grid,
                         hipStream_t stream)
{
    switch (rpw) {
    case 1: return launch_rows_t<OUT16, NW, 1>(a, nt, aligned, nomath, grid, stream);
    case 2: return launch_rows_t<OUT16, NW, 2>(a, nt, aligned, nomath, grid, stream);
    case 3: return launch_rows_t<OUT16, NW, 3>(a, nt, aligned, nomath, grid, stream);
    case 4: return launch_rows_t<OUT16, NW, 4>(a, nt, aligned, nomath, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

template <bool OUT16>
hipError_t launch_rows_o(const bf_rows_args &a, int nw, int rpw, bool nt, bool aligned, bool nomath, dim3 grid,
                         hipStream_t stream)
{
    switch (nw) {
    case 4: return launch_rows_w<OUT16, 4>(a, rpw, nt, aligned, nomath, grid, stream);
    case 8: return launch_rows_w<OUT16, 8>(a, rpw, nt, aligned, nomath, grid, stream);
    case 16: return launch_rows_w<OUT16, 16>(a, rpw, nt, aligned, nomath, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace

hipError_t bf_prepare_terms(const bf_terms_args &a, const void **func, dim3 *grid, dim3 *block)
{
    *func = nullptr;
    if (a.nt == 0 || a.pairs_pad == 0) return hipSuccess;
    if (a.pairs_pad % kBlock || a.nt > 65535u) return hipErrorInvalidValue;
    if (a.dt_dev == nullptr && a.nt > kTermsInline) return hipErrorInvalidValue;
    *func = reinterpret_cast<const void *>(&bf_terms_kernel);
    *grid = dim3(a.pairs_pad / kBlock, a.nt);
    *block = dim3(kBlock);
    return hipSuccess;
}

hipError_t bf_launch_terms(const bf_terms_args &a, hipStream_t stream)
{
    const void *func;
    dim3 grid, block;
    const hipError_t e = bf_prepare_terms(a, &func, &grid, &block);
    if (e != hipSuccess || func == nullptr) return e;
    void *params[] = {const_cast<bf_terms_args *>(&a)};
    return hipLaunchKernel(func, grid, block, params, 0, stream);
}

hipError_t bf_launch_rows(const bf_rows_args &a_in, bool out16, int waves_per_block, int rows_per_wave,
                          bool nontemporal, bool xcd_remap, bool nomath, hipStream_t stream)
{
    bf_rows_args a = a_in;
    if (a.n_pairs == 0 || a.nc == 0 || a.nt == 0) return hipSuccess;
    const uint32_t ppl = out16 ? 4u : 2u;
    const uint32_t cols = 64u * ppl * (a.same_tile ? 1u : (uint32_t)waves_per_block);
    const uint32_t rows = (uint32_t)rows_per_wave * (a.same_tile ? (uint32_t)waves_per_block : 1u);
    a.n_colgroups = (a.n_pairs + cols - 1) / cols;
    a.n_rowgroups = (a.nc + rows - 1) / rows;
    const uint64_t blocks = (uint64_t)a.n_colgroups * a.n_rowgroups * a.nt;
    if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
    a.xcd_remap = (xcd_remap && (blocks % 8u) == 0u) ? 1u : 0u;
    const bool aligned = (a.n_pairs % ppl) == 0 && (reinterpret_cast<uintptr_t>(a.out) % 16u) == 0;
    const dim3 grid((uint32_t)blocks);
    return out16 ? launch_rows_o<true>(a, waves_per_block, rows_per_wave, nontemporal, aligned, nomath, grid, stream)
                 : launch_rows_o<false>(a, waves_per_block, rows_per_wave, nontemporal, aligned, nomath, grid, stream);
}

// Every fp32 significand x in [1, 2): is dcs_div_const3(x, D) the IEEE quotient?
// (scale-invariant in x, so one binade settles this D; see bf_math.h)
__global__ void __launch_bounds__(kBlock) bf_verify_div3_kernel(float D, float y, uint32_t *mismatches)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x; // < 2^23
    const float x = dcs_bits_f32(0x3f800000u + i);
    const float q = dcs_div_const3(x, D, y);
    const float e = x / D; // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the default)
    if (dcs_f32_bits(q) != dcs_f32_bits(e)) atomicAdd(mismatches, 1u);
}

hipError_t bf_launch_verify_div3(float D, float y, uint32_t *d_mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(bf_verify_div3_kernel, dim3((1u << 23) / kBlock), dim3(kBlock), 0, stream, D, y, d_mismatches);
    return hipGetLastError();
}

// Force the code object to load now (first launch otherwise pays ~ms of lazy
// module loading inside the caller's timed region).
hipError_t bf_warm_module()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&bf_gather_beams_kernel));
}

hipError_t bf_launch_bform_terms(const bf_bform_terms_args &a, const float *dt_inline, hipStream_t stream)
{
    if (a.nt == 0 || a.n_pairs == 0) return hipSuccess;
    if (a.nt > 65535u) return hipErrorInvalidValue;
    const dim3 grid((a.n_pairs + kBlock - 1) / kBlock, a.nt);
    if (dt_inline != nullptr && a.nt > 1u) { // the times travel in the kernel arguments: nothing is staged
        if (a.nt > kDtInline) return hipErrorInvalidValue;
        bf_bform_terms_args_inl ai;
        ai.a = a;
        std::memcpy(ai.dt_inline, dt_inline, (size_t)a.nt * sizeof(float));
        hipLaunchKernelGGL(bf_bform_terms_kernel<true>, grid, dim3(kBlock), 0, stream, ai);
        return hipGetLastError();
    }
    bf_bform_terms_args b = a;
    if (dt_inline != nullptr) {
        b.dt_dev = nullptr;
        b.dt0 = dt_inline[0];
    } else if (a.dt_dev == nullptr && a.nt != 1u) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(bf_bform_terms_kernel<false>, grid, dim3(kBlock), 0, stream, b);
    return hipGetLastError();
}

hipError_t bf_launch_beamform(const bf_beamform_args &a_in, hipStream_t stream)
{
    bf_beamform_args a = a_in;
    if (a.A == 0 || a.B == 0 || a.C == 0 || a.nt16 == 0) return hipSuccess;
    if (a.chan_per_block == 0) return hipErrorInvalidValue;
    a.n_bgroups = (a.B + 15u) / 16u;
    a.n_cblocks = (a.C + a.chan_per_block - 1) / a.chan_per_block;
    const uint64_t blocks = (uint64_t)a.n_bgroups * a.n_cblocks * a.nt16;
    // channels per pass: 4 while the staged samples stay within 32 KiB of LDS per workgroup
    // (<= 64 antennas), else 2 (profiles/r01_fused.md)
    const uint32_t na = a.A < kAntChunk ? a.A : kAntChunk;
    const int ch = (a.chan_per_block >= 4 && na <= 64u) ? 4 : (a.chan_per_block >= 2 ? 2 : 1);
    const size_t lds = (size_t)na * 32u * sizeof(float) * (size_t)ch;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (ch == 4)
        hipLaunchKernelGGL(bf_beamform_kernel<4>, dim3((uint32_t)blocks), dim3(kBlock), lds, stream, a);
    else if (ch == 2)
        hipLaunchKernelGGL(bf_beamform_kernel<2>, dim3((uint32_t)blocks), dim3(kBlock), lds, stream, a);
    else
        hipLaunchKernelGGL(bf_beamform_kernel<1>, dim3((uint32_t)blocks), dim3(kBlock), lds, stream, a);
    return hipGetLastError();
}

