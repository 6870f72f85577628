// bf_beamform_mfma.hip -- beamformer with coefficient REUSE on the gfx950 matrix cores
// ("a general version is a contraction over antennas").
//
// The reference regenerates every steering coefficient for every time sample
// (calculate_beamweights_and_beamform_single_channel, BeamformerKernels.cu:192-367) and only MODELS
// what a deployed beamformer does: new coefficients every ACCUMULATIONS_BEFORE_NEW_COEFFS time units
// (BeamformerParameters.h:17; BeamformerCoefficientTest.cu:426-448).  Here the coefficients of ONE time
// are generated once per (channel, antenna, beam) -- into LDS, never HBM -- and applied to a block of
// samples:
//   beams[c][t/16][b][t%16] = ( sum_a cos(rot[a][b][c]) * re[c][t][a] ,  sum_a sin(rot[a][b][c]) * im[c][t][a] )
// (the reference's element-wise product, BeamformerKernels.cu:315-316; table indexed [b*A + a]; layouts
// BeamformerKernels.cuh:137-143).  Per channel that is two real contractions over antennas,
//   Re[beam][t] = Wre[beam][ant] x Sre[ant][t]      Im likewise,
// run on v_mfma_f32_16x16x4_f32: exact-fp32 products, accumulated as an fp32 fma chain IN ANTENNA ORDER
// (the instruction is, bit for bit, a k-ordered fmaf chain), so the result differs from the verifier's
// "sum += coeff * sample" (separate multiply and add) by the roundings of the chain only.
//
// Workgroup = 4 waves = one channel x NBT beam tiles of 16 x a range of 16-sample blocks:
//   wave w: beam tile w % NBT, sample-block slot w / NBT of each round (4 / NBT blocks per round).
// W (all antennas x 16*NBT beams, re and im planes, fp32) is generated once per workgroup into LDS from the
// terms table bf_bform_terms_kernel writes ([a][b]; L2-resident); each round stages its int8 sample blocks
// into LDS as fp32 planes.  Operand fetch is one ds_read_b32 per operand per MFMA, conflict-free:
//   A operand, lane l: W[beam l & 15][antenna 4j + (l >> 4)]   B operand: S[antenna 4j + (l >> 4)][sample l & 15]
// C/D: lane l, register r = beam (l >> 4) * 4 + r, sample l & 15.
// Roofline: int8 samples in (2 B per antenna and sample) + fp32 beams out (8 B per beam and sample) against
// HBM; the fp32 matrix rate (64 FLOP / clk / SIMD) bounds it from ~32 beams per 64 antennas upwards.

#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include "bf_device.h"

namespace {

constexpr uint32_t kKC = 64; // antennas per staged chunk (16 k-steps of 4)

template <int NBT>
__global__ void __launch_bounds__(kBlock) bf_beamform_acc_kernel(const bf_bacc_args a)
{
    constexpr int TPR = 4 / NBT;                                  // 16-sample blocks per round
    constexpr uint32_t WS = 16u * NBT + (NBT > 1 ? 16u : 0u);     // W row stride in floats (padded: no bank conflict)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // lds: Wre[A_pad][WS] | Wim[A_pad][WS] | Sre[TPR][kKC][16] | Sim[TPR][kKC][16]
    const uint32_t A_pad = (a.A + 3u) & ~3u;
    float *Wre = lds;
    float *Wim = Wre + (size_t)A_pad * WS;
    float *Sre = Wim + (size_t)A_pad * WS;
    float *Sim = Sre + (size_t)TPR * kKC * 16u;

    uint32_t bid = blockIdx.x;
    const uint32_t bg = bid % a.n_bgroups;
    bid /= a.n_bgroups;
    const uint32_t tg = bid % a.n_tgroups;
    const uint32_t c = bid / a.n_tgroups;
    const uint32_t b0 = bg * 16u * NBT;            // first beam of this workgroup
    const uint32_t tt0 = tg * a.tiles_per_wg;      // first 16-sample block
    const uint32_t tt1 = min(tt0 + a.tiles_per_wg, a.nT16);

    // ---- W for (channel c, beams [b0, b0 + 16 NBT), all antennas): once per workgroup
    {
        const uint32_t cls = a.flags[0]; // highest pair class of the table (bf_bform_terms_kernel)
        const float fChan = (float)c;
        const float D = a.k.fDenominator, y = a.k.fRcpDenominator;
        const uint32_t nb = 16u * NBT;
        auto fill = [&](auto gen) {
            for (uint32_t i = threadIdx.x; i < A_pad * nb; i += kBlock) {
                const uint32_t ant = i / nb, bl = i - ant * nb, b = b0 + bl;
                float re = 0.0f, im = 0.0f;
                if (ant < a.A && b < a.B) {
                    const floatx2 kp = *reinterpret_cast<const floatx2 *>(a.terms + 2u * ((uint64_t)ant * a.B + b));
                    gen(kp.x, kp.y, re, im);
                }
                Wre[ant * WS + bl] = re;
                Wim[ant * WS + bl] = im;
            }
        };
        if (cls == DCS_CLASS_SLOW) {
            fill([&](float kx, float ky, float &re, float &im) { coeff_slow(kx, ky, fChan, D, re, im); });
        } else {
            dispatch_fast(a.k.uDiv3Exact != 0u, cls == DCS_CLASS_FAST_LOW, [&](auto div3, auto lowdeg) {
                fill([&](float kx, float ky, float &re, float &im) {
                    coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(kx, ky, fChan, D, y, re, im);
                });
            });
        }
    }

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t bt = wave % NBT, slot = wave / NBT;
    const uint32_t lm = lane & 15u, lg = lane >> 4;
    const uint32_t tile_bytes = a.A * 32u; // one [A][16][2] int8 block

    for (uint32_t r0 = tt0; r0 < tt1; r0 += TPR) {
        floatx4 acc_re = {0.0f, 0.0f, 0.0f, 0.0f}, acc_im = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t my_tt = r0 + slot;
        for (uint32_t a0 = 0; a0 < a.A; a0 += kKC) {
            const uint32_t na = min(kKC, a.A - a0);
            __syncthreads(); // W is complete (first pass) / the previous chunk's readers are done
            // ---- stage this round's sample blocks, antennas [a0, a0 + kKC), as fp32 planes; 16 bytes
            //      (8 samples of one antenna) per thread and pass
            for (uint32_t e = threadIdx.x; e < (uint32_t)TPR * kKC * 2u; e += kBlock) {
                const uint32_t r = e / (kKC * 2u), rem = e - r * (kKC * 2u), al = rem >> 1, half = rem & 1u;
                const uint32_t tt = r0 + r;
                floatx4 re0 = {0, 0, 0, 0}, re1 = {0, 0, 0, 0}, im0 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0};
                if (al < na && tt < tt1) {
                    const uintx4 w = *reinterpret_cast<const uintx4 *>(a.ant + ((uint64_t)c * a.nT16 + tt) * tile_bytes +
                                                                       (uint64_t)(a0 + al) * 32u + half * 16u);
                    auto sx = [](uint32_t v, int byte) { return (float)(int8_t)(v >> (8 * byte)); };
                    re0 = floatx4{sx(w.x, 0), sx(w.x, 2), sx(w.y, 0), sx(w.y, 2)};
                    im0 = floatx4{sx(w.x, 1), sx(w.x, 3), sx(w.y, 1), sx(w.y, 3)};
                    re1 = floatx4{sx(w.z, 0), sx(w.z, 2), sx(w.w, 0), sx(w.w, 2)};
                    im1 = floatx4{sx(w.z, 1), sx(w.z, 3), sx(w.w, 1), sx(w.w, 3)};
                }
                float *pr = Sre + ((size_t)r * kKC + al) * 16u + half * 8u;
                float *pi = Sim + ((size_t)r * kKC + al) * 16u + half * 8u;
                *reinterpret_cast<floatx4 *>(pr) = re0;
                *reinterpret_cast<floatx4 *>(pr + 4) = re1;
                *reinterpret_cast<floatx4 *>(pi) = im0;
                *reinterpret_cast<floatx4 *>(pi + 4) = im1;
            }
            __syncthreads();
            // ---- 16 k-steps of 4 antennas: two fma chains (re, im) in antenna order
            const float *wr = Wre + (size_t)(a0 + lg) * WS + bt * 16u + lm;
            const float *wi = Wim + (size_t)(a0 + lg) * WS + bt * 16u + lm;
            const float *sr = Sre + ((size_t)slot * kKC + lg) * 16u + lm;
            const float *si = Sim + ((size_t)slot * kKC + lg) * 16u + lm;
            if (na == kKC) { // a full chunk: 16 k-steps, unrolled (operand reads run ahead of the matrix pipe)
#pragma unroll
                for (uint32_t j = 0; j < kKC / 4u; j++) {
                    acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(size_t)j * 4u * WS], sr[j * 64u], acc_re, 0, 0, 0);
                    acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[(size_t)j * 4u * WS], si[j * 64u], acc_im, 0, 0, 0);
                }
            } else {
                const uint32_t nj = (na + 3u) >> 2; // antennas past A hold W = 0 and S = 0
                for (uint32_t j = 0; j < nj; j++) {
                    acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(size_t)j * 4u * WS], sr[j * 64u], acc_re, 0, 0, 0);
                    acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[(size_t)j * 4u * WS], si[j * 64u], acc_im, 0, 0, 0);
                }
            }
        }
        if (my_tt < tt1) {
            // lane l, register r: beam b0 + 16 bt + 4 (l >> 4) + r, sample l & 15
            floatx2 *dst = reinterpret_cast<floatx2 *>(a.beams) + ((uint64_t)c * a.nT16 + my_tt) * a.B * 16u + lm;
            const uint32_t bb = b0 + bt * 16u + lg * 4u;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (bb + r < a.B) dst[(uint64_t)(bb + r) * 16u] = floatx2{acc_re[r], acc_im[r]};
        }
    }
}

} // namespace

// LDS bytes of one workgroup for NBT beam tiles and A antennas.
static size_t bacc_lds_bytes(int nbt, uint32_t A)
{
    const uint32_t A_pad = (A + 3u) & ~3u;
    const uint32_t ws = 16u * (uint32_t)nbt + (nbt > 1 ? 16u : 0u);
    return ((size_t)A_pad * ws * 2u + (size_t)(4 / nbt) * kKC * 16u * 2u) * sizeof(float);
}

hipError_t bf_launch_beamform_acc(const bf_bacc_args &a_in, hipStream_t stream)
{
    bf_bacc_args a = a_in;
    if (a.A == 0 || a.B == 0 || a.C == 0 || a.nT16 == 0) return hipSuccess;
    // beam tiles per workgroup: as many as the beams need and 64 KiB of LDS hold
    int nbt = a.B > 32u ? 4 : (a.B > 16u ? 2 : 1);
    while (nbt > 1 && bacc_lds_bytes(nbt, a.A) > 64u * 1024u) nbt >>= 1;
    const size_t lds = bacc_lds_bytes(nbt, a.A);
    if (lds > 64u * 1024u) return hipErrorInvalidValue; // more than 256 antennas: not built
    a.n_bgroups = (a.B + 16u * (uint32_t)nbt - 1u) / (16u * (uint32_t)nbt);
    // 16-sample blocks per workgroup: all of them (W is generated once per workgroup), fewer while that leaves
    // the chip under 512 workgroups; whole rounds of 4 / nbt blocks
    const uint32_t tpr = 4u / (uint32_t)nbt;
    uint32_t tiles = (a.nT16 + tpr - 1u) / tpr * tpr;
    if (tiles > 64u * tpr) tiles = 64u * tpr;
    while (tiles > tpr && (uint64_t)a.C * a.n_bgroups * ((a.nT16 + tiles - 1u) / tiles) < 512u) tiles = ((tiles / tpr + 1u) / 2u) * tpr;
    a.tiles_per_wg = tiles;
    a.n_tgroups = (a.nT16 + tiles - 1u) / tiles;
    const uint64_t blocks = (uint64_t)a.C * a.n_bgroups * a.n_tgroups;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)blocks), block(kBlock);
    if (nbt == 4)
        hipLaunchKernelGGL(bf_beamform_acc_kernel<4>, grid, block, lds, stream, a);
    else if (nbt == 2)
        hipLaunchKernelGGL(bf_beamform_acc_kernel<2>, grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL(bf_beamform_acc_kernel<1>, grid, block, lds, stream, a);
    return hipGetLastError();
}
