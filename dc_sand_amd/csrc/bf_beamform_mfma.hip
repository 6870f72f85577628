// bf_beamform_mfma.hip -- beamformer with coefficient REUSE on the gfx950 matrix cores
// ("a general version is a contraction over antennas").
//
// The reference regenerates every steering coefficient for every time sample
// (calculate_beamweights_and_beamform_single_channel, BeamformerKernels.cu:192-367) and only MODELS
// what a deployed beamformer does: new coefficients every ACCUMULATIONS_BEFORE_NEW_COEFFS time units
// (BeamformerParameters.h:17; BeamformerCoefficientTest.cu:426-448).  Here the coefficients of ONE time
// are generated once per (channel, antenna, beam) -- into registers or LDS, never HBM -- and applied to a
// block of samples:
//   beams[c][t/16][b][t%16] = ( sum_a cos(rot[a][b][c]) * re[c][t][a] ,  sum_a sin(rot[a][b][c]) * im[c][t][a] )
// (the reference's element-wise product, BeamformerKernels.cu:315-316; table indexed [b*A + a]; layouts
// BeamformerKernels.cuh:137-143).  Per channel that is two real contractions over antennas,
//   Re[beam][t] = Wre[beam][ant] x Sre[ant][t]      Im likewise.
// Roofline: int8 samples in (2 B per antenna and sample) + fp32 beams out (8 B per beam and sample) against HBM.
//
// Two forms (bf_bacc_args.fp32_chain):
//
// 1. bf_beamform_i8_kernel (default; second half of this file): the contraction in EXACT integer arithmetic on
//    v_mfma_i32_16x16x64_i8 -- 24-bit fixed-point coefficients as three signed digits -- which takes the matrix
//    work out of the picture (32 x the fp32 pipe's rate) and leaves a kernel bound by the memory system:
//    at 16 beams (as many bytes in as out) 5.0-5.7 TB/s, 80-90 % of what the leanest device copy kernel moves on
//    the same box (6.3 TB/s) and more than hipMemcpyDtoD (5.0); at >= 64 beams (output-dominated) 4.7-4.9 TB/s, the
//    rate this chip gives long-lived waves that each stream 64 stores (profiles/r01_store_patterns.md: 5.4-5.5).
//
// 2. bf_beamform_acc_kernel (first half): v_mfma_f32_16x16x4_f32, exact-fp32 products accumulated as an fp32 fma
//    chain IN ANTENNA ORDER (the instruction is, bit for bit, a k-ordered fmaf chain), so the result differs from
//    the verifier's "sum += coeff * sample" (separate multiply and add) by the roundings of the chain only.
//    Workgroup = 4 waves = one channel x NBT beam tiles of 16 x a range of 16-sample blocks:
//      wave w: beam tile w % NBT, sample-block slot w / NBT of each round (4 / NBT blocks per round).
//    A lane reads the (re, im) int8 pair of its own B operand straight from global memory (one 2-byte load per
//    k-step), one stage ahead of the matrix pipe, and converts it on the VALU.  W (all antennas x 16*NBT beams,
//    re and im planes) is generated once per workgroup into LDS from the terms table bf_bform_terms_kernel
//    writes ([a][b]; L2-resident) -- the kernel's only barrier -- and read from there, one ds_read_b32 per A
//    operand.  0.38-0.53 of the fp32 matrix peak across six structures (profiles/r02_fused.md); kept as the form
//    whose rounding is the verifier's loop with fused multiply-adds.
//      A operand, lane l: W[beam l & 15][antenna 4j + (l >> 4)]   B operand: S[antenna 4j + (l >> 4)][sample l & 15]
//    C/D: lane l, register r = beam (l >> 4) * 4 + r, sample l & 15.

#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include <type_traits>

#include "bf_device.h"

namespace {

// Workgroup numbering, XCD-aware.  The hardware hands workgroup w to XCD w % 8, and every XCD has its own L2.  The
// logical order below is beam group fastest, then sample-block group, then channel: the n_bgroups workgroups that
// read the SAME samples (one channel's blocks) are neighbours in it.  Numbered as dispatched, those neighbours land on
// different XCDs and each L2 fetches the samples again (FETCH_SIZE 4.0 x the algorithmic input at 64 x 256 beams,
// profiles/r02_fused.md -- served by the 256 MiB Infinity Cache behind the L2s, not by HBM, as it turned out).  With
// G = a.xcd_group > 1 the dispatch number w (XCD x = w % 8, that XCD's q-th workgroup, q = w / 8) becomes the logical
// number of member q % G of sharer group (q / G) * 8 + x: the G sharers follow each other on ONE XCD -- all but the
// first hit its L2 (FETCH_SIZE 1.04 x) -- while the eight XCDs still work on eight NEIGHBOURING groups at any moment.
// (Giving every XCD one contiguous eighth of the order instead keeps the sharers together just as well, but sends the
// XCDs to eight far-apart address windows: 6 % slower where there is nothing to share.)  A bijection: on the whole
// multiples of 8 G by construction, identity on the tail; identity altogether for G = 1.  The launcher decides G from
// measurements (profiles/r03_fused.md): the sharers' count for the forms of more than 64 antennas (each workgroup reads
// 4 x what it writes: -18 % time), and for the staged form from 16 sharers on (+3.5 %; at 2 - 8 sharers the grouped
// order measured 1.5 - 7 % SLOWER -- four workgroups missing on the same lines of one L2 at the same moment -- so those
// stay as dispatched).
// (the function itself: bf_kernels.h, bf_xcd_grouped -- host-callable too, so that tests/test_host_abi.py can check the
// bijection on the CPU through probes/libdcs_probes.so)
__device__ __forceinline__ uint32_t xcd_grouped(uint32_t w, uint32_t total, uint32_t G) { return bf_xcd_grouped(w, total, G); }
#ifdef DCS_PROBES
// A/B of the numbering (dcs_probe_knobs.bacc_order): 0 = the launcher's choice, 1 = as dispatched (round 2), 2 = one
// contiguous eighth of the order per XCD, 3 = sharers grouped whatever their number
__device__ __forceinline__ uint32_t probe_order(uint32_t order, uint32_t w, uint32_t total, uint32_t G, uint32_t n_sharers)
{
    if (order == 1u) return w;
    if (order >= 16u && order < 24u) return (w + (order - 16u)) % total; // 16 + r: as dispatched, rotated by r (XCD <-> channel affinity probe)
    if (order == 2u) {
        const uint32_t per = total >> 3, rem = total & 7u, x = w & 7u, q = w >> 3;
        return x * per + min(x, rem) + q;
    }
    return xcd_grouped(w, total, order == 3u ? n_sharers : G);
}
#define BACC_LOGICAL_ID(a) probe_order((a).order, blockIdx.x, gridDim.x, (a).xcd_group, (a).n_bgroups)
#else
#define BACC_LOGICAL_ID(a) xcd_grouped(blockIdx.x, gridDim.x, (a).xcd_group)
#endif

constexpr uint32_t kKC = 64; // antennas per staged chunk (16 k-steps of 4)

template <int NBT>
__global__ void __launch_bounds__(kBlock) bf_beamform_acc_kernel(const bf_bacc_args a)
{
    constexpr int TPR = 4 / NBT;                                  // 16-sample blocks per round
    constexpr uint32_t WS = 16u * NBT + (NBT > 1 ? 16u : 0u);     // W row stride in floats (padded: no bank conflict)
    constexpr uint32_t NJ = kKC / 4u;                             // k-steps per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];   // Wre[A_pad][WS] | Wim[A_pad][WS]
    const uint32_t A_pad = (a.A + 3u) & ~3u;

    uint32_t bid = BACC_LOGICAL_ID(a);
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

    const uint32_t fw = a.flags[0]; // (epoch << 2) | highest pair class of the table (bf_bform_terms_kernel)
    const uint32_t cls = (fw >> 2) == a.epoch ? (fw & 3u) : DCS_CLASS_FAST_LOW;
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


// ---------------------------------------------------------------------------------------------------------------
// The same contraction on the int8 matrix pipe (v_mfma_i32_16x16x64_i8, 32x the fp32 pipe's rate), EXACTLY:
// the samples ARE int8, and a coefficient w in [-1, 1] is taken as the 24-bit fixed-point number
//   F = rint(w * 8355711) = d1 * 65536 + d2 * 256 + d3,   d1, d2, d3 in [-128, 127]   (8355711 = 0x7F7F7F),
// so that  sum_a w_a x_a  ~=  (65536 * sum d1 x  +  256 * sum d2 x  +  sum d3 x) / 8355711:  three integer
// contractions per plane and 64 antennas whose sums are exact (|sum| <= 128 * 128 * 64 = 2^20) -- no rounding
// depends on the order of the antennas; the two low sums are combined in integers, and that and the high sum (and
// the chunks of more than 64 antennas) are scaled by exact powers of two and added in fp32.  What differs from the
// verifier's fp32 "sum += coeff * sample" is the quantisation of each coefficient (|F / 8355711 - w| <=
// 0.75 / 8355711 = 9e-8: the size of the fp32 coefficient's own last place) and the verifier's OWN accumulation
// roundings; against an exact-arithmetic sum of the fp32 coefficients the result is within 9e-8 * sum_a |x_a| + a
// few ulp (tests/test_gpu_parity.py holds it to that).
//
// A wave owns one 16-beam tile of one channel and some of the workgroup's 16-sample blocks.  Its coefficients
// (64 antennas = 6 operands of 4 registers: 3 digits x {re, im}) are made once, in registers (waves that own the
// same tile make a share each and exchange them through LDS).  Up to 64 antennas the workgroup's sample blocks
// (at most 16 = 32 KiB) travel to LDS by LDS-DMA while the coefficients are being made; beyond (kChain, round 3), the
// workgroup's waves make the coefficients of 64 antennas each, put them in LDS, and every wave walks all the antenna
// chunks of its own sample blocks with the matrix instruction's accumulator.  Per pair of blocks and 64 antennas: 16 four-byte operand reads per
// lane, a 4 x 4 byte transpose into four K = 64 operands (slot (lane >> 4, byte p) of BOTH operands is antenna
// 64 ch + 4 p + (lane >> 4): the contraction index may be permuted freely as long as both sides agree), 12 MFMAs,
// ~130 vector instructions to recombine, 4 sixteen-byte stores.  The arithmetic is hidden entirely: with its stores
// alone, no loads and no coefficients, the kernel is as fast (profiles/r02_fused.md).
typedef int intx4 __attribute__((ext_vector_type(4)));
constexpr float kFixScale = 8355711.0f;

// The three signed digits of rint(w * 8355711), one per byte (byte 0 = d3 ... byte 2 = d1; byte 3 unused): with every
// digit biased by 128 the number F + 0x808080 is a plain 24-bit unsigned whose bytes are d + 128, and (d + 128) ^ 0x80
// is d as a signed byte -- no borrows to chase.
__device__ __forceinline__ uint32_t fixed_word(float w)
{
    // a coefficient one ulp above 1 must not carry into a fourth digit
    const float f = __builtin_amdgcn_fmed3f(w * kFixScale, -kFixScale, kFixScale);
    return ((uint32_t)(int)rintf(f) + 0x808080u) ^ 0x808080u;
}

// Three forms of one kernel (FORM):
//   kStaged (nr_stations <= 64): the workgroup's sample blocks (at most 16: 32 KiB) travel to LDS by LDS-DMA, all at
//       once and while the coefficients are being made, and the waves read their operands from there -- every wave of
//       the workgroup needs the same blocks when it owns several beam tiles, and no wave ever waits for a global load
//       inside its loop (a wave that loads its own operands pays one memory latency per trip; with loads and stores
//       but no arithmetic that form ran exactly as fast as with the arithmetic).  Wave w: beam tile w % nbt, blocks
//       w / nbt, w / nbt + 4 / nbt, ...
//   kDirect (nr_stations <= 64; probes and A/B only): the same, every wave loading its own operands.
//   kSplit  (64 < nr_stations <= 256): the CONTRACTION INDEX is split over the waves -- wave w holds the coefficients
//       of antennas [64 w, 64 w + 64) of the workgroup's ONE beam tile (24 registers, as in the other forms, instead of
//       96 in one wave), loads that chunk's operands itself, and the four partial sums of a pair of blocks meet in
//       LDS: each wave adds up, scales and stores the four beams of one result register (wave w: beams w, w + 4 ... of
//       the tile).  Two barriers per trip of two pairs.
//   kChain  (64 < nr_stations <= 256; the product's form there since round 3): the workgroup owns ONE beam tile and its
//       coefficients -- wave w makes those of antennas [64 w, 64 w + 64), as in kSplit -- go to LDS (4 chunks x 6 operands x
//       1 KiB = 24 KiB), from where EVERY wave reads them (one ds_read_b128 per operand and chunk).  Wave w then takes the
//       sample blocks w, w + 4, ... by itself and walks ALL the antenna chunks of a pair of blocks with the matrix
//       instruction's own accumulator: the twelve integer sums (3 digits x 4 planes) run through the chunks in int32,
//       exactly (|sum| <= 128 * 128 * 256 = 2^22), and are recombined ONCE per pair.  Against kSplit: no partial sums in
//       LDS, no barrier inside the loop (kSplit: two per trip, every wave waiting for the slowest), a quarter of the
//       recombination arithmetic, and a result that is one fp32 rounding closer to the exact sum.
// FULL: nr_stations is a multiple of 64 (no antenna masks, immediate load offsets).
enum { kStaged = 0, kDirect = 1, kSplit = 2, kChain = 3 };
// kChain is allocated for 4 waves per SIMD (127 VGPRs, no scratch: the coefficient digits go straight to LDS as they are made,
// and the results are recombined and stored one register at a time) and walks with TWO sample buffers.  It measures the same at 3
// waves (the kernel is issue-bound: profiles/r03_fused.md); tried and not kept: a third sample buffer (168 registers, 2 spills), the
// next step's loads issued before this step's wait (1-4 % slower).
constexpr int kChainWaves = 4;

// NW: waves per workgroup -- 4; 8 (kStaged, eight beam tiles per workgroup) is instantiated in the probes build only: an A/B
// that measured no gain over four tiles (profiles/r03_fused.md)
template <int FORM, bool FULL, int NW = 4>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(FORM == kSplit ? (FULL ? 3 : 2) : (FORM == kChain ? kChainWaves : (FULL || FORM == kStaged ? 4 : 3)))))
bf_beamform_i8_kernel(const bf_bacc_args a)
{
    static_assert(NW == 4 || ((NW == 8 || NW == 16) && FORM == kStaged), "8- and 16-wave workgroups exist for the staged form only");
    constexpr bool STAGED = FORM == kStaged, SPLIT = FORM == kSplit, CHAIN = FORM == kChain;
    extern __shared__ __attribute__((aligned(16))) char staged[]; // kStaged: the sample image (+ the coefficient exchange); kSplit: the partial sums
    uint32_t bid = BACC_LOGICAL_ID(a);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bg = bid % a.n_bgroups;
    bid /= a.n_bgroups;
    const uint32_t tg = bid % a.n_tgroups;
    const uint32_t c = bid / a.n_tgroups;
    // beam tiles per workgroup, sample blocks per round (kSplit: one tile, every wave takes every block)
    const uint32_t nbt_log2 = SPLIT || CHAIN ? 0u : a.nbt_log2;
    const uint32_t nbt = 1u << nbt_log2, tpr = SPLIT ? 1u : (uint32_t)NW >> nbt_log2;
    const uint32_t bt = SPLIT ? 0u : wave & (nbt - 1u), slot = SPLIT ? 0u : wave >> nbt_log2;
    const uint32_t kc = SPLIT || CHAIN ? wave : 0u; // the 64-antenna chunk whose coefficients this wave makes (and, kSplit, contracts)
    const uint32_t lm = lane & 15u, lg = lane >> 4;
    const uint32_t bw = (bg * nbt + bt) * 16u;     // first beam of this wave's tile
    const uint32_t tt0 = tg * a.tiles_per_wg;      // first 16-sample block of the workgroup
    const uint32_t tt1 = min(tt0 + a.tiles_per_wg, a.nT16);
    const uint32_t n_blocks = tt1 > tt0 + slot ? (tt1 - tt0 - slot + tpr - 1u) / tpr : 0u; // this wave's sample blocks
    const bool idle = bw >= a.B || n_blocks == 0u; // wave-uniform (kSplit: workgroup-uniform)
    if (!STAGED && !CHAIN && idle) return;         // (no barrier behind this in the direct form; all waves alike in kSplit)
    const bool has_chunk = !(SPLIT || CHAIN) || 64u * kc < a.A; // kSplit / kChain with <= 192 antennas: the last wave(s) have no chunk of their own
    if (STAGED) { // this wave's share of the workgroup's blocks: global -> LDS, 1 KiB per instruction, same byte order
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const uint32_t bytes = (tt1 - tt0) * a.A * 32u; // a multiple of 32
        const char *src = reinterpret_cast<const char *>(a.ant) + ((uint64_t)c * a.nT16 + tt0) * a.A * 32u;
        for (uint32_t k = wave; k * 1024u < bytes; k += (uint32_t)NW) // a piece that overhangs the end re-reads the last 16 bytes
            __builtin_amdgcn_global_load_lds((glb_void *)(src + min(k * 1024u + lane * 16u, bytes - 16u)), (lds_void *)(staged + k * 1024u),
                                             16, 0, 0);
    }

    const uint32_t fw = a.flags[0]; // (epoch << 2) | highest pair class of the table (bf_bform_terms_kernel)
    const uint32_t cls = (fw >> 2) == a.epoch ? (fw & 3u) : DCS_CLASS_FAST_LOW;
    const float fChan = (float)c;
    const float D = a.k.fDenominator, y = a.k.fRcpDenominator;

    // ---- coefficients: lane (row lm, group lg) holds, in byte p of each of its six operands (3 digits x {re, im}),
    //      antenna 64 kc + 4 p + lg.  16 terms loads in flight together; the fast classes fully unrolled (16 copies of
    //      ~34 instructions, nothing moves), the slow class in four rolled trips.
    // kStaged, fewer than four beam tiles per workgroup: the 4 / nbt waves that own the same tile make a quarter (half)
    // of its coefficient registers each and exchange them through LDS behind the staging barrier (the slow class makes
    // everything everywhere: its rolled loop does not split)
    const bool shared_w = STAGED && a.share_off != 0u && tpr > 1u && cls != DCS_CLASS_SLOW;
    intx4 wre[3], wim[3];
#pragma unroll
    for (int d = 0; d < 3; d++) wre[d] = wim[d] = intx4{0, 0, 0, 0};
    // A coefficient that is not finite (an infinite or NaN delay value: the slow class) has no fixed-point digits; the
    // verifier's sum for that beam and plane is NaN whatever the samples are (NaN * 0 = NaN), and so it is here: the lanes
    // note it, the wave folds the notes into one bit per result row and plane, and the rows are stored as NaN.
    bool bad_re = false, bad_im = false;
    auto make_coefficients = [&]() {
        // row i of the result tile is beam 4 (i & 3) + (i >> 2): the four lane groups of a store instruction then
        // hold four CONSECUTIVE beams (512 contiguous bytes per block) instead of every fourth
        const uint32_t beam = bw + 4u * (lm & 3u) + (lm >> 2);
        const bool beam_live = beam < a.B;
        const float *tp = a.terms + 2u * (uint64_t)min(beam, a.B - 1u);
        // four antennas' words -> one register of each digit plane (a 4 x 4 byte transpose; byte z = antenna z)
        auto planes = [](const uint32_t (&g)[4], uint32_t (&out)[3]) {
            const uint32_t t0 = __builtin_amdgcn_perm(g[1], g[0], 0x05010400u), t1 = __builtin_amdgcn_perm(g[1], g[0], 0x07030602u);
            const uint32_t t2 = __builtin_amdgcn_perm(g[3], g[2], 0x05010400u), t3 = __builtin_amdgcn_perm(g[3], g[2], 0x07030602u);
            out[2] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); // bytes 0: d3
            out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u); // bytes 1: d2
            out[0] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); // bytes 2: d1
        };
        // antennas 64 kc + 4 (4 q + z) + lg, z = 0..3, from their terms: one register of each of the six operands
        auto four = [&](auto gen, auto track, uint32_t q, const floatx2 (&k4)[4], uint32_t (&nr)[3], uint32_t (&ni)[3]) {
            uint32_t gr[4], gi[4];
#pragma unroll
            for (uint32_t z = 0; z < 4; z++) {
                float re, im;
                gen(k4[z].x, k4[z].y, re, im);
                if (decltype(track)::value) { // the slow class only: infinite or NaN delay values end here
                    bad_re |= !(fabsf(re) <= 2.0f);
                    bad_im |= !(fabsf(im) <= 2.0f);
                }
                gr[z] = fixed_word(re), gi[z] = fixed_word(im);
            }
            planes(gr, nr), planes(gi, ni);
            if (!FULL) { // antennas beyond nr_stations: zero digits, byte by byte
                uint32_t mask = 0;
#pragma unroll
                for (uint32_t z = 0; z < 4; z++) mask |= 64u * kc + 4u * (4u * q + z) + lg < a.A ? 0xffu << (8u * z) : 0u;
#pragma unroll
                for (int d = 0; d < 3; d++) nr[d] &= mask, ni[d] &= mask;
            }
        };
        auto generate = [&](auto gen, auto unrolled) {
            floatx2 kp[16];
#pragma unroll
            for (uint32_t z = 0; z < 16; z++) {
                const uint32_t ant = 64u * kc + 4u * z + lg;
                kp[z] = *reinterpret_cast<const floatx2 *>(tp + 2u * (uint64_t)min(ant, a.A - 1u) * a.B);
            }
            if (decltype(unrolled)::value) {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    if (shared_w && (q & (tpr - 1u)) != slot) continue; // a wave sharing its tile makes its own registers only
                    const floatx2 k4[4] = {kp[4 * q], kp[4 * q + 1], kp[4 * q + 2], kp[4 * q + 3]};
                    uint32_t nr[3], ni[3];
                    four(gen, std::false_type{}, q, k4, nr, ni);
                    if constexpr (CHAIN) { // straight to the LDS image (component q of the chunk's six operands): the 24 registers are never held
                        uint32_t *cw = reinterpret_cast<uint32_t *>(staged) + (kc * 6u * 64u + lane) * 4u + q;
#pragma unroll
                        for (int d = 0; d < 3; d++) cw[(uint32_t)d * 256u] = beam_live ? nr[d] : 0u, cw[(3u + (uint32_t)d) * 256u] = beam_live ? ni[d] : 0u;
                    } else {
#pragma unroll
                        for (int d = 0; d < 3; d++) wre[d][q] = (int)nr[d], wim[d][q] = (int)ni[d];
                    }
                }
            } else { // the new register enters at the top while the others, and the loaded terms, move down -- no
                     // register is indexed by a loop variable
#pragma unroll 1
                for (uint32_t q = 0; q < 4; q++) {
                    const floatx2 k4[4] = {kp[0], kp[1], kp[2], kp[3]};
                    uint32_t nr[3], ni[3];
                    four(gen, std::true_type{}, q, k4, nr, ni);
                    if constexpr (CHAIN) {
                        uint32_t *cw = reinterpret_cast<uint32_t *>(staged) + (kc * 6u * 64u + lane) * 4u + q;
#pragma unroll
                        for (int d = 0; d < 3; d++) cw[(uint32_t)d * 256u] = beam_live ? nr[d] : 0u, cw[(3u + (uint32_t)d) * 256u] = beam_live ? ni[d] : 0u;
                    } else {
#pragma unroll
                        for (int d = 0; d < 3; d++) {
                            wre[d] = intx4{wre[d][1], wre[d][2], wre[d][3], (int)nr[d]};
                            wim[d] = intx4{wim[d][1], wim[d][2], wim[d][3], (int)ni[d]};
                        }
                    }
#pragma unroll
                    for (uint32_t z = 0; z < 12; z++) kp[z] = kp[z + 4];
                }
            }
            if (!CHAIN && !beam_live) { // beams beyond nr_beams: zero coefficients (their results are not stored either)
#pragma unroll
                for (int d = 0; d < 3; d++) wre[d] = wim[d] = intx4{0, 0, 0, 0};
            }
        };
        if (cls == DCS_CLASS_SLOW) {
            generate([&](float kx, float ky, float &re, float &im) { coeff_slow(kx, ky, fChan, D, re, im); }, std::false_type{});
        } else {
            dispatch_fast(a.k.uDiv3Exact != 0u, cls == DCS_CLASS_FAST_LOW, [&](auto div3, auto lowdeg) {
                generate([&](float kx, float ky, float &re, float &im) {
                    coeff_fast<decltype(div3)::value, decltype(lowdeg)::value>(kx, ky, fChan, D, y, re, im);
                }, std::true_type{});
            });
        }
    };

    // ---- sample blocks, two at a time.  Column n of the B operand is NOT "sample n of one block": a lane loads the
    //      4 bytes {re, im} x samples (2 m, 2 m + 1), m = n & 7, of block A (n < 8) or block B (n >= 8) of its pair, so
    //      one 4-byte load per antenna serves FOUR contractions -- (even samples, odd samples) x (re, im) -- whose
    //      16 columns are 8 sample pairs of block A and 8 of block B.  A 4 x 4 byte transpose (8 v_perm_b32 per 4
    //      antennas) turns 16 loaded registers into the 4 K = 64 operands; a result lane holds samples 2 m and 2 m + 1 of
    //      its 4 beams: one 16-byte store each.  Half the load and store instructions of a per-block scheme and no
    //      half-word merging (d16 loads do not keep the other half with SRAM-ECC on).
    //      Global addresses (kDirect, kSplit): a wave-uniform base (scalar registers) plus a per-lane byte offset; with
    //      whole chunks (FULL) ONE offset register and the instruction's immediate (antenna 4 p + lg is 128 p bytes
    //      further), otherwise offsets clamped to the last antenna (the coefficient digits are 0 beyond nr_stations).
    //      Two load sets are in flight per wave.  hipcc waits for ALL memory operations at the head of a loop whose
    //      loads cross the back-edge, stores included, so the order inside a trip is: wait, transpose, MFMAs, STORES,
    //      then the next trip's LOADS -- one memory latency per trip, shared by loads and stores (with the loads issued
    //      first, each trip paid the load and the store latency one after the other: 3.2 us per block and wave).
    const char *ant8 = reinterpret_cast<const char *>(a.ant);
    const uint32_t blk_bytes = a.A * 32u;            // one 16-sample block of one channel: <= 8 KiB
    const uint32_t m = lm & 7u;
    const bool second = lm >= 8u;                    // this lane's columns belong to block B of the pair
    const uint32_t last = n_blocks - 1u;
    const uint32_t voff = lg * 32u + m * 4u;
    uint32_t cur[2][16];
    // "every loaded register is needed HERE": keeps the compiler from sinking a load set into the trip that consumes it
    auto arrived = [&](uint32_t (&v)[16]) {
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    };
    auto fetch = [&](uint32_t (&dst)[16], uint32_t blk, uint32_t kc) { // pair (blk, min(blk + 1, last)), antenna chunk kc
        const uint32_t blkA = min(blk, last), blkB = min(blk + 1u, last);
        if (STAGED) { // from the LDS image: block j of the workgroup at j * blk_bytes
            const uint32_t at = ((second ? blkB : blkA) * tpr + slot) * blk_bytes + m * 4u;
#pragma unroll
            for (uint32_t p = 0; p < 16; p++)
                dst[p] = *reinterpret_cast<const uint32_t *>(staged + at + (FULL ? lg + 4u * p : min(lg + 4u * p, a.A - 1u)) * 32u);
            return;
        }
        const char *base = ant8 + ((uint64_t)c * a.nT16 + tt0 + blkA * tpr + slot) * blk_bytes; // wave-uniform
        const uint32_t hop = second ? (blkB - blkA) * tpr * blk_bytes : 0u;
        if (FULL) {
            const char *b2 = base + 2048u * kc;
            const uint32_t vo = hop + voff;
#pragma unroll
            for (uint32_t p = 0; p < 16; p++) dst[p] = *reinterpret_cast<const uint32_t *>(b2 + vo + 128u * p);
        } else {
#pragma unroll
            for (uint32_t p = 0; p < 16; p++)
                dst[p] = *reinterpret_cast<const uint32_t *>(base + (hop + min(64u * kc + lg + 4u * p, a.A - 1u) * 32u + m * 4u));
        }
    };
    // x[0] = re of the even samples, x[1] = im even, x[2] = re odd, x[3] = im odd; byte p of each = antenna 4 p + lg
    auto transpose = [&](const uint32_t (&v)[16], intx4 (&x)[4]) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t t0 = __builtin_amdgcn_perm(v[4 * q + 1], v[4 * q], 0x05010400u);     // {a0.b0, a1.b0, a0.b1, a1.b1}
            const uint32_t t1 = __builtin_amdgcn_perm(v[4 * q + 1], v[4 * q], 0x07030602u);     // {a0.b2, a1.b2, a0.b3, a1.b3}
            const uint32_t t2 = __builtin_amdgcn_perm(v[4 * q + 3], v[4 * q + 2], 0x05010400u);
            const uint32_t t3 = __builtin_amdgcn_perm(v[4 * q + 3], v[4 * q + 2], 0x07030602u);
            x[0][q] = (int)__builtin_amdgcn_perm(t2, t0, 0x05040100u);
            x[1][q] = (int)__builtin_amdgcn_perm(t2, t0, 0x07060302u);
            x[2][q] = (int)__builtin_amdgcn_perm(t3, t1, 0x05040100u);
            x[3][q] = (int)__builtin_amdgcn_perm(t3, t1, 0x07060302u);
        }
    };
    // One load set (64 antennas of a pair of blocks) as fp32 sums f[v], plane v = (re even, im even, re odd, im odd):
    // three integer contractions from zero per plane, each exact (|sum| <= 2^20); the two low digits are combined in
    // integers (s2 * 256 + s3 < 2^29: exact), converted (one rounding, far below the result's last place), and the
    // high digit enters with one fma.
    const intx4 zero = {0, 0, 0, 0};
    auto contract = [&](const intx4 (&x)[4], floatx4 (&f)[4]) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const intx4 s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v & 1) ? wim[2] : wre[2], x[v], zero, 0, 0, 0);
            const intx4 s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v & 1) ? wim[1] : wre[1], x[v], zero, 0, 0, 0);
            const intx4 s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8((v & 1) ? wim[0] : wre[0], x[v], zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) f[v][r] = fmaf((float)s1[r], 65536.0f, (float)(s2[r] * 256 + s3[r]));
        }
    };
    const float inv = 1.0f / kFixScale;
    uint32_t nan_re = 0, nan_im = 0;     // wave-uniform: rows whose re / im plane is NaN (set once the coefficients are made)
    const float fNaN = __builtin_nanf("");
    // register r of this lane is result row 4 lg + r
    auto poison = [&](int r, floatx4 &o) { // o = {re even, im even, re odd, im odd}
        if ((nan_re >> (4u * lg + (uint32_t)r)) & 1u) o[0] = fNaN, o[2] = fNaN;
        if ((nan_im >> (4u * lg + (uint32_t)r)) & 1u) o[1] = fNaN, o[3] = fNaN;
    };
    const uint32_t bb = bw + lg;         // register r of this lane: beam bb + 4 r
    const uint32_t out_blk = a.B * 128u; // bytes per 16-sample block of one channel
    char *out8 = reinterpret_cast<char *>(a.beams);
    // where this lane's 16 bytes {re, im} x samples (2 m, 2 m + 1) of beam bb + 4 r go, pair (blk, blk + 1)
    auto out_of = [&](uint32_t blk, int r) {
        const uint32_t blkA = min(blk, last), blkB = min(blk + 1u, last);
        char *base = out8 + ((uint64_t)c * a.nT16 + tt0 + blkA * tpr + slot) * out_blk; // wave-uniform
        return reinterpret_cast<floatx4 *>(base + ((second ? (blkB - blkA) * tpr * out_blk : 0u) + bb * 128u + m * 16u + 512u * r));
    };
    auto store = [&](floatx4 *dst, const floatx4 o) {
#ifdef DCS_PROBES
        if (a.plain_stores) { // probes build only: the A/B of profiles/r02_fused.md
            *dst = o;
            return;
        }
#endif
        __builtin_nontemporal_store(o, dst); // written once, read by another kernel: do not keep it in L2
    };
    // scale and store: lane l, register r = beam bw + (l >> 4) + 4 r, samples 2 m, 2 m + 1
    auto finish = [&](auto whole, uint32_t blk, const floatx4 (&f)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) { // beam bb + 4 r
            floatx4 o = {f[0][r] * inv, f[1][r] * inv, f[2][r] * inv, f[3][r] * inv};
            if (nan_re | nan_im) poison(r, o);
            if (decltype(whole)::value || bb + 4u * r < a.B) {
                floatx4 *dst = out_of(blk, r);
#ifdef DCS_PROBES
                if (a.probe == 4u) { // same bytes, but each instruction writes ONE contiguous KiB (values land in the wrong places)
                    const uint32_t blkA = min(blk, last), blkB = min(blk + 1u, last);
                    dst = reinterpret_cast<floatx4 *>(out8 + ((uint64_t)c * a.nT16 + tt0 + blkA * tpr + slot) * out_blk +
                                                      ((r >= 2 ? (blkB - blkA) * tpr * out_blk : 0u) + bw * 128u + (r & 1) * 1024u + lane * 16u));
                }
#endif
                store(dst, o);
            }
        }
    };
    // kSplit: the partial sums of two pairs meet in LDS -- [pair h][register r][chunk][lane] x 16 bytes -- and wave w
    // adds up (in chunk order), scales and stores register r = w
    floatx4 *part = reinterpret_cast<floatx4 *>(staged);
    const uint32_t n_chunks = (a.A + 63u) / 64u;
    auto park = [&](int h, const floatx4 (&f)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            floatx4 o = {f[0][r], f[1][r], f[2][r], f[3][r]};
            if (nan_re | nan_im) poison(r, o); // the chunk's NaN rows reach the sum through its partial sums
            part[((h * 4 + r) * 4 + kc) * 64u + lane] = o;
        }
    };
    auto gather = [&](auto whole, int h, uint32_t blk) {
        floatx4 o = part[((h * 4 + wave) * 4 + 0) * 64u + lane];
        for (uint32_t k = 1; k < n_chunks; k++) o = o + part[((h * 4 + wave) * 4 + k) * 64u + lane];
        o = o * inv;
        if (decltype(whole)::value || bb + 4u * wave < a.B) store(out_of(blk, (int)wave), o);
    };
    auto run = [&](auto whole) {
#ifdef DCS_PROBES
        if (!SPLIT && (a.probe == 1u || a.probe == 3u || a.probe == 4u)) { // stores only: what does the memory system make of this store pattern alone?
            floatx4 f[4];
#pragma unroll
            for (int v = 0; v < 4; v++) f[v] = floatx4{(float)wre[0][0], (float)wre[1][1], (float)wim[0][2], (float)wim[2][3]};
            for (uint32_t blk = 0; blk < n_blocks; blk += 2) finish(whole, blk, f);
            return;
        }
        if (FORM == kDirect && a.probe == 2u) { // loads and stores, no arithmetic between them
            for (uint32_t blk = 0; blk < n_blocks; blk += 4) {
                arrived(cur[0]);
                arrived(cur[1]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    floatx4 f[4];
#pragma unroll
                    for (int v = 0; v < 4; v++)
                        f[v] = floatx4{(float)cur[h][4 * v], (float)cur[h][4 * v + 1], (float)cur[h][4 * v + 2], (float)cur[h][4 * v + 3]};
                    finish(whole, blk + 2u * h, f);
                }
                __builtin_amdgcn_sched_barrier(0);
                fetch(cur[0], blk + 4u, kc);
                fetch(cur[1], blk + 6u, kc);
            }
            return;
        }
#endif
        if (STAGED) { // operands from LDS: nothing to wait for but the LDS itself
            for (uint32_t blk = 0; blk < n_blocks; blk += 2) {
                intx4 x[4];
                floatx4 f[4];
                fetch(cur[0], blk, 0u);
                transpose(cur[0], x);
                contract(x, f);
                finish(whole, blk, f);
            }
        } else { // a trip = two pairs of sample blocks (a pair past the end repeats the last block)
            for (uint32_t blk = 0; blk < n_blocks; blk += 4) {
                if (has_chunk) {
                    arrived(cur[0]);
                    arrived(cur[1]);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        intx4 x[4];
                        floatx4 f[4];
                        transpose(cur[h], x);
                        contract(x, f);
                        if (SPLIT)
                            park(h, f);
                        else
                            finish(whole, blk + 2u * h, f);
                    }
                }
                if (SPLIT) {
                    __syncthreads(); // every chunk's partial sums are in LDS
                    gather(whole, 0, blk);
                    gather(whole, 1, blk + 2u);
                    __syncthreads(); // ... and read, before the next trip overwrites them
                }
                __builtin_amdgcn_sched_barrier(0); // the loads stay behind the stores (see above)
                if (has_chunk) {
                    fetch(cur[0], blk + 4u, kc);
                    fetch(cur[1], blk + 6u, kc);
                }
            }
        }
    };
    // the first trip's samples (kStaged: all of them, above) travel while the coefficients are made
    if (!STAGED && !CHAIN && has_chunk) fetch(cur[0], 0, kc);
    if (CHAIN && !idle) fetch(cur[0], 0, 0u); // the first (pair, chunk) of this wave's own blocks
    __builtin_amdgcn_sched_barrier(0);
#ifdef DCS_PROBES
    if (a.probe == 3u || a.probe == 4u) { // no coefficients either: the store pattern alone
#pragma unroll
        for (int d = 0; d < 3; d++) wre[d] = wim[d] = intx4{(int)lane, d, 2, 1};
    } else
#endif
    if ((shared_w || CHAIN ? bw < a.B : !idle) && has_chunk) make_coefficients(); // (kChain: a wave without blocks still makes its chunk)
    if (cls == DCS_CLASS_SLOW) { // bit i: result row i (lanes i, i + 16, i + 32, i + 48 hold its antennas) has a non-finite coefficient
        const uint64_t br = __builtin_amdgcn_ballot_w64(bad_re), bi = __builtin_amdgcn_ballot_w64(bad_im);
        nan_re = (uint32_t)((br | (br >> 16) | (br >> 32) | (br >> 48)) & 0xffffu);
        nan_im = (uint32_t)((bi | (bi >> 16) | (bi >> 32) | (bi >> 48)) & 0xffffu);
    }
    if constexpr (STAGED) {
        uint32_t *wx = reinterpret_cast<uint32_t *>(staged + a.share_off) + bt * (4u * 6u * 64u) + lane; // [tile][q][plane][lane]
        if (shared_w && bw < a.B) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                if ((q & (tpr - 1u)) == slot) {
#pragma unroll
                    for (int d = 0; d < 3; d++) wx[(q * 6u + d) * 64u] = (uint32_t)wre[d][q], wx[(q * 6u + 3u + d) * 64u] = (uint32_t)wim[d][q];
                }
        }
        __syncthreads(); // hipcc drains the LDS-DMA (vmcnt(0)) in front of it
        if (idle) return;
        if (shared_w) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                if ((q & (tpr - 1u)) != slot) {
#pragma unroll
                    for (int d = 0; d < 3; d++) wre[d][q] = (int)wx[(q * 6u + d) * 64u], wim[d][q] = (int)wx[(q * 6u + 3u + d) * 64u];
                }
        }
    } else if constexpr (CHAIN) {
        // ---- the coefficients of this wave's chunk to LDS: [chunk][operand: re d1, d2, d3, im d1, d2, d3][lane] x 16 bytes
        intx4 *coef = reinterpret_cast<intx4 *>(staged);
        uint32_t *nanw = reinterpret_cast<uint32_t *>(staged + 4u * 6u * 64u * 16u); // [chunk][re, im]
        // (make_coefficients has written this wave's chunk: operand o of chunk k at coef[(k * 6 + o) * 64 + lane])
        if (lane == 0u) nanw[2u * wave] = has_chunk ? nan_re : 0u, nanw[2u * wave + 1u] = has_chunk ? nan_im : 0u;
        __syncthreads(); // the kernel's only barrier
        if (idle) return;
        nan_re = nanw[0] | nanw[2] | nanw[4] | nanw[6]; // a non-finite coefficient in ANY chunk poisons the row
        nan_im = nanw[1] | nanw[3] | nanw[5] | nanw[7];
        const uint32_t n_chunks = (a.A + 63u) / 64u;
        auto run_chain = [&](auto whole) {
            // a step = (pair of this wave's blocks, antenna chunk), pair-major; two sample buffers in turn: the samples of step
            // s + 1 are requested before step s is worked on (step 0 was requested before the coefficient making)
            const uint32_t n_steps = ((n_blocks + 1u) >> 1) * n_chunks;
            intx4 acc[4][3];
            uint32_t s = 0, chunk = 0, blk = 0;
            uint32_t f_s = 1u, f_chunk = 1u % n_chunks, f_blk = 2u * (1u / n_chunks); // next step to request
            auto step = [&](uint32_t (&now)[16], uint32_t (&ahead)[16]) {
                arrived(now);
                if (f_s < n_steps) fetch(ahead, f_blk, f_chunk);
                f_s++;
                if (++f_chunk == n_chunks) f_chunk = 0u, f_blk += 2u;
                intx4 x[4];
                transpose(now, x);
                if (chunk == 0u) {
#pragma unroll
                    for (int v = 0; v < 4; v++)
#pragma unroll
                        for (int d = 0; d < 3; d++) acc[v][d] = zero;
                }
#pragma unroll
                for (int half = 0; half < 2; half++) { // re planes (v = 0, 2), then im planes (v = 1, 3)
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        const intx4 w = coef[(chunk * 6u + 3u * (uint32_t)half + (uint32_t)d) * 64u + lane];
#pragma unroll
                        for (int v = half; v < 4; v += 2) acc[v][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, x[v], acc[v][d], 0, 0, 0);
                    }
                }
                if (chunk + 1u == n_chunks) { // all antennas in: digits d1 = acc[v][0], d2 = acc[v][1], d3 = acc[v][2]
                    // one result register (four beams' sixteen bytes) at a time: recombined, scaled, stored -- the sixteen floats
                    // are never all alive beside the accumulators
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        floatx4 o;
#pragma unroll
                        for (int v = 0; v < 4; v++) o[v] = fmaf((float)acc[v][0][r], 65536.0f, (float)(acc[v][1][r] * 256 + acc[v][2][r])) * inv;
                        if (nan_re | nan_im) poison(r, o);
                        if (decltype(whole)::value || bb + 4u * (uint32_t)r < a.B) store(out_of(blk, r), o);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    chunk = 0u, blk += 2u;
                } else {
                    chunk++;
                }
                s++;
            };
            while (s < n_steps) {
                step(cur[0], cur[1]);
                if (s < n_steps) step(cur[1], cur[0]);
            }
        };
        if (bw + 16u <= a.B)
            run_chain(std::true_type{});
        else
            run_chain(std::false_type{});
        return;
    } else if (has_chunk) {
        fetch(cur[1], 2, kc);
    }
    if (bw + 16u <= a.B)
        run(std::true_type{});
    else
        run(std::false_type{});
}

} // namespace

// LDS bytes of one workgroup for NBT beam tiles and A antennas.
static size_t bacc_lds_bytes(int nbt, uint32_t A)
{
    const uint32_t A_pad = (A + 3u) & ~3u;
    const uint32_t ws = 16u * (uint32_t)nbt + (nbt > 1 ? 16u : 0u);
    return (size_t)A_pad * ws * 2u * sizeof(float);
}

// This translation unit is a code object of its own: load it when the context is created, not in the first (timed) call.
hipError_t bf_warm_module_mfma()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&bf_beamform_i8_kernel<kStaged, true>));
}

hipError_t bf_launch_beamform_acc(const bf_bacc_args &a_in, hipStream_t stream)
{
    bf_bacc_args a = a_in;
    if (a.A == 0 || a.B == 0 || a.C == 0 || a.nT16 == 0) return hipSuccess;
    if (a.A > 256u) return hipErrorInvalidValue; // not built
    const bool chain = a.fp32_chain != 0u;
    // beam tiles per workgroup: as many as the beams need; the fp32 form keeps its coefficient planes in LDS and
    // takes as many as still admit 6 workgroups per CU (26 KiB each), one tile whatever it takes beyond
    // more than 64 antennas, int8 form: one beam tile per workgroup; kChain (the product's form) or, probes build only, kSplit
    const bool wide = !chain && a.A > 64u;
    const bool split = wide && BACC_KNOB(a, unstaged) != 0u; // kSplit (round 2): every wave takes every block, partial sums meet in LDS
    const bool staged_form = !chain && a.A <= 64u && !BACC_KNOB(a, unstaged); // at most 16 blocks (32 KiB of LDS) per workgroup
    int nbt = wide ? 1 : (a.B > 32u ? 4 : (a.B > 16u ? 2 : 1));
#ifdef DCS_PROBES
    // A/B: beam tiles per workgroup.  1 / 2 / 4 / 8 (eight-wave workgroups) at 64 x 256 x 4096 x 256: 552 / 502 / 396 / 399 us --
    // four is where the samples' re-staging stops mattering (profiles/r03_fused.md); eight exists in the probes build only
    if (staged_form && (a.nbt_force == 1u || a.nbt_force == 2u || a.nbt_force == 4u || a.nbt_force == 8u)) nbt = (int)a.nbt_force;
#endif
    uint32_t nw = nbt == 8 ? 8u : 4u; // waves per workgroup
#ifdef DCS_PROBES
    if (staged_form && (a.nw_force == 8u || a.nw_force == 16u) && a.nw_force >= (uint32_t)nbt) nw = a.nw_force; // A/B: waves per workgroup
#endif
    while (chain && nbt > 1 && bacc_lds_bytes(nbt, a.A) > 26u * 1024u) nbt >>= 1;
    const size_t lds = chain ? bacc_lds_bytes(nbt, a.A) : 0u;
    a.nbt_log2 = nbt == 8 ? 3u : (nbt == 4 ? 2u : (nbt == 2 ? 1u : 0u));
    a.n_bgroups = (a.B + 16u * (uint32_t)nbt - 1u) / (16u * (uint32_t)nbt);
    // 16-sample blocks per workgroup: whole rounds of 4 / nbt blocks, at most max_rounds (the coefficients are
    // generated once per workgroup; but a launch of only a few thousand long-lived workgroups ends with most of the
    // chip idle behind the last ones), fewer while that leaves the chip under 4096 workgroups
    const uint32_t tpr = split ? 1u : nw / (uint32_t)nbt; // (K-split: every wave takes every block)
    // (64 beams exactly fill one four-tile workgroup per channel: there EIGHT blocks per workgroup measured 3-5 % faster than
    // sixteen on four shapes, while from 128 beams on eight or twelve blocks cost 3-7 %: profiles/r03_fused.md)
    const uint32_t blocks_cap = staged_form && nbt == 4 && a.n_bgroups == 1u ? 8u : 16u;
    const uint32_t max_rounds = BACC_KNOB(a, max_rounds) ? BACC_KNOB(a, max_rounds) : (chain ? 16u : (staged_form || wide ? blocks_cap / tpr : 32u));
    uint32_t tiles = (a.nT16 + tpr - 1u) / tpr * tpr;
    if (tiles > max_rounds * tpr) { // several workgroups per (channel, beam group): equal shares (17 blocks are 9 + 8, not 16 + 1)
        const uint32_t parts = (a.nT16 + max_rounds * tpr - 1u) / (max_rounds * tpr);
        tiles = ((a.nT16 + parts - 1u) / parts + tpr - 1u) / tpr * tpr;
    }
    // (the int8 form makes its coefficients once per wave -- half its arithmetic at 16 blocks -- so it only splits
    // further while the chip, which holds 1280 of its workgroups, would not even be filled once)
    const uint64_t enough = chain ? 4096u : 5120u / nw;
    while (tiles > tpr && (uint64_t)a.C * a.n_bgroups * ((a.nT16 + tiles - 1u) / tiles) < enough) tiles = ((tiles / tpr + 1u) / 2u) * tpr;
    a.tiles_per_wg = tiles;
    a.n_tgroups = (a.nT16 + tiles - 1u) / tiles;
    // the workgroups that share a channel's samples on one XCD? (xcd_grouped above; measured)
    a.xcd_group = wide ? a.n_bgroups : (!chain && a.n_bgroups >= 16u ? a.n_bgroups : 1u);
    const uint64_t blocks = (uint64_t)a.C * a.n_bgroups * a.n_tgroups;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)blocks), block(64u * nw);
    if (chain) {
        if (nbt == 4)
            hipLaunchKernelGGL(bf_beamform_acc_kernel<4>, grid, block, lds, stream, a);
        else if (nbt == 2)
            hipLaunchKernelGGL(bf_beamform_acc_kernel<2>, grid, block, lds, stream, a);
        else
            hipLaunchKernelGGL(bf_beamform_acc_kernel<1>, grid, block, lds, stream, a);
    } else if (staged_form) {
        size_t stage_bytes = ((size_t)a.tiles_per_wg * a.A * 32u + 1023u) / 1024u * 1024u;
        if (tpr > 1u && !BACC_KNOB(a, no_share)) { // waves that own the same tile share the making of its coefficients
            a.share_off = (uint32_t)stage_bytes;
            stage_bytes += (size_t)nbt * 4u * 6u * 64u * sizeof(uint32_t);
        }
#ifdef DCS_PROBES
        if (a.wg_per_cu >= 1u && a.wg_per_cu <= 5u && stage_bytes < 160u * 1024u / a.wg_per_cu) // residency cap: unused LDS
            stage_bytes = (160u * 1024u / a.wg_per_cu) & ~1023u;
#endif
#ifdef DCS_PROBES
        if (nw == 8u) {
            if (a.A == 64u)
                hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, true, 8>), grid, block, stage_bytes, stream, a);
            else
                hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, false, 8>), grid, block, stage_bytes, stream, a);
        } else if (nw == 16u) {
            if (a.A == 64u)
                hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, true, 16>), grid, block, stage_bytes, stream, a);
            else
                hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, false, 16>), grid, block, stage_bytes, stream, a);
        } else
#endif
        if (a.A == 64u)
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, true>), grid, block, stage_bytes, stream, a);
        else
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kStaged, false>), grid, block, stage_bytes, stream, a);
#ifdef DCS_PROBES
    } else if (a.A <= 64u) { // kDirect: the probes build's A/B form only
        if (a.A == 64u)
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kDirect, true>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kDirect, false>), grid, block, 0, stream, a);
#endif
#ifdef DCS_PROBES
    } else if (split) { // 2 pairs x 4 registers x 4 chunks x 64 lanes x 16 bytes of partial sums
        const size_t part_bytes = 2u * 4u * 4u * 64u * 16u;
        if (a.A % 64u == 0u)
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kSplit, true>), grid, block, part_bytes, stream, a);
        else
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kSplit, false>), grid, block, part_bytes, stream, a);
#endif
    } else { // kChain: 4 chunks x 6 operands x 64 lanes x 16 bytes of coefficients + the chunks' NaN-row words
        const size_t coef_bytes = 4u * 6u * 64u * 16u + 8u * sizeof(uint32_t);
        if (a.A % 64u == 0u)
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kChain, true>), grid, block, coef_bytes, stream, a);
        else
            hipLaunchKernelGGL((bf_beamform_i8_kernel<kChain, false>), grid, block, coef_bytes, stream, a);
    }
    return hipGetLastError();
}
