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
// The waves never wait for each other in the sample loop: a lane reads the (re, im) int8 pair of its own
// B operand straight from global memory (one 2-byte load per k-step; a wave's 64 lanes cover one 128-byte
// line, which the waves of the other beam tiles then find in L1), one stage ahead of the matrix pipe, and
// converts it on the VALU, which idles otherwise.  W (all antennas x 16*NBT beams, re and im planes) is generated
// once per workgroup into LDS from the terms table bf_bform_terms_kernel writes ([a][b]; L2-resident) -- the
// kernel's only barrier -- and read from there, one ds_read_b32 per A operand (a quarter of the LDS bandwidth at
// the full matrix rate).  Few registers, so that 6-8 waves per SIMD hide the memory latencies: loads AND stores
// take 2-3 thousand cycles under this load, three times a wave's matrix phase; versions that held W in registers
// (106-138 VGPRs, 3-4 waves per SIMD) or staged samples through LDS behind barriers all stalled at 45 %.  (Two earlier versions staged
// the samples through LDS -- as fp32 planes, then as int8 -- behind two barriers per round: 45 % of the matrix
// rate, bound first by LDS bandwidth, then by the waves waiting for each other.)
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
    constexpr uint32_t NJ = kKC / 4u;                             // k-steps per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];   // Wre[A_pad][WS] | Wim[A_pad][WS]
    const uint32_t A_pad = (a.A + 3u) & ~3u;

    uint32_t bid = blockIdx.x;
    const uint32_t bg = bid % a.n_bgroups;
    bid /= a.n_bgroups;
    const uint32_t tg = bid % a.n_tgroups;
    const uint32_t c = bid / a.n_tgroups;
    const uint32_t b0 = bg * 16u * NBT;            // first beam of this workgroup
    const uint32_t tt0 = tg * a.tiles_per_wg;      // first 16-sample block
    const uint32_t tt1 = min(tt0 + a.tiles_per_wg, a.nT16);

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t bt = wave % NBT, slot = wave / NBT;
    const uint32_t lm = lane & 15u, lg = lane >> 4;

    const uint32_t cls = a.flags[0]; // highest pair class of the table (bf_bform_terms_kernel)
    const float fChan = (float)c;
    const float D = a.k.fDenominator, y = a.k.fRcpDenominator;
    // one coefficient through the class's path (slow: IEEE divide + fp64 sincos; rare, workgroup-uniform)
    auto with_generator = [&](auto &&body) {
        if (cls == DCS_CLASS_SLOW) {
            body([&](float kx, float ky, float &re, float &im) { coeff_slow(kx, ky, fChan, D, re, im); });
        } else {
            dispatch_fast(a.k.uDiv3Exact != 0u, cls == DCS_CLASS_FAST_LOW, [&](auto div3, auto lowdeg) {
                body([&](float kx, float ky, float &re, float &im) {
                    coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(kx, ky, fChan, D, y, re, im);
                });
            });
        }
    };

    // ---- W for (channel c, beams [b0, b0 + 16 NBT), all antennas) into LDS: once per workgroup, a rolled loop
    //      (a lane computing its own 16 fragments unrolled cost 178 registers and 36 000 lines of code);
    //      batches of 8 terms loads in flight together, unconditional on clamped indices and masked afterwards
    //      (a load under an exec mask makes hipcc wait for it on the spot: one memory latency per load)
    {
        float *Wre = lds, *Wim = lds + (size_t)A_pad * WS;
        const uint32_t nb = 16u * NBT;
        with_generator([&](auto gen) {
            constexpr uint32_t kBatch = 8;
            for (uint32_t i0 = threadIdx.x; i0 < A_pad * nb; i0 += kBlock * kBatch) {
                floatx2 kp[kBatch];
#pragma unroll
                for (uint32_t q = 0; q < kBatch; q++) {
                    const uint32_t i = min(i0 + q * kBlock, A_pad * nb - 1u), ant = i / nb, b = b0 + (i - ant * nb);
                    kp[q] = *reinterpret_cast<const floatx2 *>(a.terms + 2u * ((uint64_t)min(ant, a.A - 1u) * a.B + min(b, a.B - 1u)));
                }
#pragma unroll 1
                for (uint32_t q = 0; q < kBatch; q++) {
                    const uint32_t i = i0 + q * kBlock, ant = i / nb, bl = i - ant * nb;
                    floatx2 t = kp[0];
#pragma unroll
                    for (uint32_t z = 1; z < kBatch; z++) t = (z == q) ? kp[z] : t; // kp[q] without indexing registers
                    float re, im;
                    gen(t.x, t.y, re, im);
                    if (i < A_pad * nb) {
                        const bool live = ant < a.A && b0 + bl < a.B;
                        Wre[ant * WS + bl] = live ? re : 0.0f;
                        Wim[ant * WS + bl] = live ? im : 0.0f;
                    }
                }
            }
        });
        __syncthreads(); // the only barrier of the kernel
    }
    // A "stage" of this wave = (one of its 16-sample blocks, chunk of kKC antennas), block-major.  The (re, im)
    // int8 pairs of stage s + 1 are requested one stage ahead.  Whole chunks (64 antennas) take a path without any
    // per-operand mask or clamp: the vector ALU issues at most ~12 instructions per MFMA on a SIMD, across all its
    // waves, and a version that masked every operand (3-4 VALU instructions each) ran at 45-55 % of the matrix rate
    // with or without its loads and stores.
    const uint32_t n_chunks = (a.A + kKC - 1u) / kKC;
    const uint32_t n_blocks = tt1 > tt0 + slot ? (tt1 - tt0 - slot + TPR - 1u) / TPR : 0u; // this wave's sample blocks
    const uint32_t n_stages = n_blocks * n_chunks;
    const uint16_t *ant16 = reinterpret_cast<const uint16_t *>(a.ant);
    uint32_t cur[NJ], nxt[NJ];
    auto fetch = [&](uint32_t s, uint32_t (&dst)[NJ]) {
        const uint32_t tt = tt0 + (s / n_chunks) * TPR + slot, a0 = (s % n_chunks) * kKC; // tt < tt1 by construction
        const uint16_t *p0 = ant16 + ((uint64_t)c * a.nT16 + tt) * a.A * 16u + lm; // [c][tt][antenna 0][sample] of (re, im)
        if (a0 + kKC <= a.A) { // wave-uniform
            const uint16_t *p = p0 + (size_t)(a0 + lg) * 16u;
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) dst[j] = p[(size_t)j * 64u]; // one base address, immediate offsets
        } else {
            // the last, partial chunk: raw, from a clamped (always valid) antenna index, masked where it is consumed.
            // (A select(cond, load, 0) here becomes a load under an exec mask that hipcc waits for on the spot.)
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) dst[j] = p0[(size_t)min(a0 + lg + 4u * j, a.A - 1u) * 16u];
        }
    };
    floatx4 acc_re = {0.0f, 0.0f, 0.0f, 0.0f}, acc_im = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n_stages) fetch(0, cur);
    for (uint32_t s = 0; s < n_stages; s++) {
        const uint32_t blk = s / n_chunks, chunk = s - blk * n_chunks, a0 = chunk * kKC;
        if (s + 1 < n_stages) fetch(s + 1, nxt);
        // ---- k-steps of 4 antennas: two fma chains (re, im) in antenna order
        const float *wr = lds + (size_t)(a0 + lg) * WS + bt * 16u + lm;
        const float *wi = wr + (size_t)A_pad * WS;
        if (a0 + kKC <= a.A) {
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(size_t)j * 4u * WS], (float)(int8_t)(cur[j] & 0xffu), acc_re, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[(size_t)j * 4u * WS], (float)(int8_t)(cur[j] >> 8), acc_im, 0, 0, 0);
            }
        } else { // antennas past A: W = 0 (rows up to A_pad) and S masked to 0
            const uint32_t nj = (a.A - a0 + 3u) >> 2;
            for (uint32_t j = 0; j < nj; j++) {
                uint32_t v = cur[0];
#pragma unroll
                for (uint32_t z = 1; z < NJ; z++) v = (z == j) ? cur[z] : v; // cur[j] without indexing registers
                v &= 0u - (uint32_t)(a0 + lg + 4u * j < a.A);
                acc_re = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(size_t)j * 4u * WS], (float)(int8_t)(v & 0xffu), acc_re, 0, 0, 0);
                acc_im = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[(size_t)j * 4u * WS], (float)(int8_t)(v >> 8), acc_im, 0, 0, 0);
            }
        }
        if (chunk + 1 == n_chunks) { // the block's last chunk: store, start the next block's sums
            // lane l, register r: beam b0 + 16 bt + 4 (l >> 4) + r, sample l & 15
            floatx2 *dst = reinterpret_cast<floatx2 *>(a.beams) + ((uint64_t)c * a.nT16 + tt0 + blk * TPR + slot) * a.B * 16u + lm;
            const uint32_t bb = b0 + bt * 16u + lg * 4u;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (bb + r < a.B) dst[(uint64_t)(bb + r) * 16u] = floatx2{acc_re[r], acc_im[r]};
            acc_re = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
            acc_im = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (uint32_t j = 0; j < NJ; j++) cur[j] = nxt[j];
    }
}

} // namespace

// LDS bytes of one workgroup for NBT beam tiles and A antennas.
static size_t bacc_lds_bytes(int nbt, uint32_t A)
{
    const uint32_t A_pad = (A + 3u) & ~3u;
    const uint32_t ws = 16u * (uint32_t)nbt + (nbt > 1 ? 16u : 0u);
    return (size_t)A_pad * ws * 2u * sizeof(float);
}

hipError_t bf_launch_beamform_acc(const bf_bacc_args &a_in, hipStream_t stream)
{
    bf_bacc_args a = a_in;
    if (a.A == 0 || a.B == 0 || a.C == 0 || a.nT16 == 0) return hipSuccess;
    // beam tiles per workgroup: as many as the beams need while LDS still admits 6 workgroups per CU (26 KiB each);
    // one tile whatever it takes beyond (256 antennas: 32 KiB)
    int nbt = a.B > 32u ? 4 : (a.B > 16u ? 2 : 1);
    while (nbt > 1 && bacc_lds_bytes(nbt, a.A) > 26u * 1024u) nbt >>= 1;
    const size_t lds = bacc_lds_bytes(nbt, a.A);
    if (lds > 64u * 1024u) return hipErrorInvalidValue; // more than 256 antennas: not built
    a.n_bgroups = (a.B + 16u * (uint32_t)nbt - 1u) / (16u * (uint32_t)nbt);
    // 16-sample blocks per workgroup: all of them (W is generated once per workgroup), fewer while that leaves
    // the chip under 4096 workgroups; whole rounds of 4 / nbt blocks
    const uint32_t tpr = 4u / (uint32_t)nbt;
    // (at most 16 rounds: a CU holds ~5 workgroups, and a launch of only a few thousand long-lived workgroups ends
    // with most of the chip idle behind the last ones -- 4096 workgroups of 64 rounds ran at 80 % of what 16384 of
    // 16 rounds do)
    uint32_t tiles = (a.nT16 + tpr - 1u) / tpr * tpr;
    if (tiles > 16u * tpr) tiles = 16u * tpr;
    while (tiles > tpr && (uint64_t)a.C * a.n_bgroups * ((a.nT16 + tiles - 1u) / tiles) < 4096u) tiles = ((tiles / tpr + 1u) / 2u) * tpr;
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
