// bf_kernels.h -- host-callable launchers of the gfx950 kernels in
// bf_kernels.hip (internal to libdcs_beamformer.so; the public boundary is
// include/dcs_beamformer.h).
#ifndef DCS_BF_KERNELS_H
#define DCS_BF_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bf_math.h"

// One tiled launch: time steps [0, nt) x channels [c0, c0+nc) x all pairs.
struct bf_tiled_args {
    const dcs_delay_vals *delays; // compact table [n_pairs]
    void *out;                    // [nt][nc][n_pairs] of {re,im} (fp32 pair or half2)
    const float *dt_dev;          // fDeltaTime per time step (device), or nullptr
    float dt0;                    // used when dt_dev == nullptr and nt == 1
    uint32_t n_pairs;
    uint32_t c0, nc;              // channel slab
    uint32_t nt;
    uint32_t chan_per_block;      // channels a workgroup walks
    uint32_t n_tile_groups;       // ceil(n_pairs / pairs per workgroup)
    uint32_t n_cblocks;           // ceil(nc / chan_per_block)
    uint32_t xcd_remap;           // workgroups sharing blockIdx % 8 (one XCD) take consecutive (tile, channel block)s
#ifdef DCS_PROBES
    uint32_t pace;                // probes build only: 64-cycle sleeps before each store of the fast loop
#endif
    dcs_bf_consts k;
    // terms-table variant (nullptr: the workgroups compute and stage their pairs' terms themselves):
    const float *terms;           // [nt][pairs_pad][2], written by bf_launch_terms just before
    const uint32_t *flags;        // [nt][pairs_pad/64]
    uint32_t pairs_pad;
};
// The same with fDeltaTime of up to kDtInline time steps by value (kernel arguments): the reference's
// default tensor (256 time steps, 134 MB) is a 22 us kernel, and a pinned->device copy of the dt table
// in front of it cost another 9 us.  Used when a.dt_dev == nullptr and a.nt > 1; dt_inline[0] == a.dt0.
// One-time-step launches (the per-time-step host loops of NAIVE / MULTIPLE_CHANNELS, every streaming
// tick) take the 0.1 KiB bf_tiled_args alone: those loops are bound by the host's launch rate.
constexpr uint32_t kDtInline = 256;
struct bf_tiled_args_inl {
    bf_tiled_args a;
    float dt_inline[kDtInline];
};

// A resolved launch (kernel instantiation, geometry, final arguments): what
// bf_launch_tiled enqueues, and what a hipGraph kernel node is built from.
struct bf_kernel_launch {
    const void *func; // nullptr: nothing to launch (empty shape)
    dim3 grid, block;
    uint32_t shared; // dynamic LDS bytes requested (occupancy limiter; the kernel does not use them)
    bf_tiled_args_inl args; // the kernel's parameter is `args` (inline form) or `args.a` (same address)
};
// dt_inline: nullptr, or a.nt values (a.nt <= kDtInline) that travel in the kernel arguments
hipError_t bf_prepare_tiled(const bf_tiled_args &a, const float *dt_inline, bool out16, int tiles_per_block,
                            bool nontemporal, bf_kernel_launch *out);

// Row-streaming form: a terms table written by bf_launch_terms, then one short
// wave per (1-KiB tile, rows_per_wave channel rows).
struct bf_terms_args {
    const dcs_delay_vals *delays; // [n_pairs]
    float *terms;                 // [nt][pairs_pad][2]
    uint32_t *flags;              // [nt][pairs_pad/64]
    const float *dt_dev;          // or nullptr -> dt_inline (nt <= kTermsInline; dt_inline[0] == dt0)
    float dt0;
    uint32_t n_pairs, pairs_pad, nt;
    dcs_bf_consts k;
    float dt_inline[8];
    // slow-path duty (tiled form's terms-table variant; nullptr otherwise): the output tensor
    // [nt][nc][n_pairs] of {re,im} and its channel slab, as in bf_tiled_args
    void *out;
    uint32_t c0, nc, out16;
};
constexpr uint32_t kTermsInline = 8;
hipError_t bf_launch_terms(const bf_terms_args &a, hipStream_t stream);
// the same launch resolved but not enqueued (a hipGraph kernel node is built from it); *func == nullptr: nothing to launch
hipError_t bf_prepare_terms(const bf_terms_args &a, const void **func, dim3 *grid, dim3 *block);

struct bf_rows_args {
    const float *terms;
    const uint32_t *flags;
    void *out; // [nt][nc][n_pairs]
    uint32_t n_pairs, pairs_pad;
    uint32_t c0, nc, nt;
    uint32_t n_colgroups, n_rowgroups; // filled by the launcher
    uint32_t xcd_remap;
    uint32_t div3; // dcs_bf_consts::uDiv3Exact
    uint32_t same_tile; // the workgroup's waves share one tile and interleave rows
#ifdef DCS_PROBES
    uint32_t pace;      // probes build only: 64-cycle sleeps before each store
#endif
    uint32_t lds_pad;   // host only: dynamic LDS the launch asks for (occupancy limiter, unused by the kernel)
    float D, y;
};
hipError_t bf_launch_rows(const bf_rows_args &a, bool out16, int waves_per_block, int rows_per_wave,
                          bool nontemporal, bool xcd_remap, bool nomath, hipStream_t stream);

// Fused coefficient generation + beamforming (table indexed [b*A + a]).
struct bf_bform_terms_args {
    const dcs_delay_vals *delays; // [B][A]
    float *terms;                 // [nt][A][B][2]
    uint32_t *flags;              // [nt]: atomicMax of (epoch << 2) | class; words of earlier epochs count as the lowest class
    uint32_t epoch;               // this call's number (1 .. 2^30 - 2), from the context
    const float *dt_dev;          // [nt], or nullptr: nt == 1 and fDeltaTime is dt0 (by value: nothing to stage)
    float dt0;
    uint32_t n_pairs, A, B, nt;
    dcs_bf_consts k;
};
struct bf_bform_terms_args_inl {
    bf_bform_terms_args a;
    float dt_inline[kDtInline];
};
// dt_inline: nullptr (fDeltaTime from a.dt_dev, or a.dt0 when nt == 1), or a.nt <= kDtInline values that travel in the
// kernel arguments
hipError_t bf_launch_bform_terms(const bf_bform_terms_args &a, const float *dt_inline, hipStream_t stream);

struct bf_beamform_args {
    const float *terms;    // [nt16*16][A][B][2] for this launch's time steps
    const uint32_t *flags; // [nt16*16], epoch-tagged (bf_bform_terms_args)
    uint32_t epoch;
    const int8_t *ant;     // [C][nt16_total][A][16][2]
    float *beams;          // [C][nt16_total][B][16][2]
    uint32_t A, B, C;
    uint32_t nt16;         // 16-sample blocks in this launch
    uint32_t tex0;         // first of them within the whole tensor
    uint32_t nt16_total;
    uint32_t chan_per_block;
    uint32_t n_bgroups, n_cblocks; // filled by the launcher
    dcs_bf_consts k;
};
hipError_t bf_launch_beamform(const bf_beamform_args &a, hipStream_t stream);

// Beamformer with coefficient reuse on the matrix cores (bf_beamform_mfma.hip): the coefficients of ONE time
// (terms table [A][B] from bf_launch_bform_terms with nt = 1) applied to nT16 blocks of 16 samples.
struct bf_bacc_args {
    const float *terms;    // [A][B][2]
    const uint32_t *flags; // [1]: (epoch << 2) | highest pair class of the table
    uint32_t epoch;
    const int8_t *ant;     // [C][nT16][A][16][2]
    float *beams;          // [C][nT16][B][16][2]
    uint32_t A, B, C, nT16;
    uint32_t fp32_chain;   // 0: exact fixed-point contraction on the int8 matrix pipe; 1: fp32 fma chain (v_mfma_f32_16x16x4_f32)
    uint32_t share_off;    // staged int8 form: LDS offset of the coefficient exchange (filled by the launcher; 0 = none)
#ifdef DCS_PROBES
    // A/B switches of the measurements in profiles/r02_fused.md / r03_fused.md (include/dcs_probes.h: dcs_probe_knobs);
    // the product build has none of them -- BACC_KNOB() below reads as a constant 0 there
    uint32_t max_rounds;   // cap on the rounds of sample blocks per workgroup (0 = the launcher's choice)
    uint32_t no_share;     // staged int8 form: every wave makes all its coefficients
    uint32_t plain_stores; // int8 form: ordinary instead of nontemporal stores
    uint32_t wg_per_cu;    // staged int8 form: at most this many workgroups resident per CU (0 = as many as fit)
    uint32_t unstaged;     // int8 form, <= 64 antennas: operands straight from global memory instead of through LDS
    uint32_t order;        // workgroup numbering: 0 = the launcher's choice, 1 = as dispatched (round 2), 2 = a contiguous eighth per XCD, 3 = sharers always grouped
    uint32_t nbt_force;    // staged form: beam tiles per workgroup (1, 2, 4, 8; 0 = the launcher's choice)
    uint32_t nw_force;     // staged form: waves per workgroup (8, 16; 0 = the launcher's choice)
    uint32_t probe;        // 1 = stores only, 2 = loads and stores without arithmetic, 3 = stores without coefficients either,
                           // 4 = as 3 with one contiguous KiB per store instruction
#endif
    uint32_t tiles_per_wg, n_bgroups, n_tgroups, nbt_log2, xcd_group; // filled by the launcher
    dcs_bf_consts k;
};
// Dispatch number -> logical workgroup number that puts G consecutive logical workgroups on ONE XCD (the hardware deals
// workgroup w to XCD w % 8) while the eight XCDs work on eight neighbouring groups: XCD x = w % 8's q-th workgroup
// (q = w / 8) is member q % G of group (q / G) * 8 + x.  A bijection of [0, total): by construction on the whole multiples
// of 8 G, identity on the tail and for G <= 1.  (bf_beamform_mfma.hip says when it is used.)
__host__ __device__ inline uint32_t bf_xcd_grouped(uint32_t w, uint32_t total, uint32_t G)
{
    if (G <= 1u) return w;
    const uint32_t full = total - total % (8u * G);
    if (w >= full) return w;
    const uint32_t x = w & 7u, q = w >> 3;
    return ((q / G) * 8u + x) * G + q % G;
}

#ifdef DCS_PROBES
#define BACC_KNOB(a, f) ((a).f)
#else
#define BACC_KNOB(a, f) 0u
#endif
hipError_t bf_launch_beamform_acc(const bf_bacc_args &a, hipStream_t stream);
hipError_t bf_warm_module_mfma();

// One coefficient per lane, one time step (reference kernel a1's shape).
struct bf_naive_args {
    const dcs_delay_vals *delays;
    float *out; // [nc][n_pairs][2]
    float dt;
    uint32_t n_pairs;
    uint32_t c0, nc;
    dcs_bf_consts k;
};
hipError_t bf_launch_naive(const bf_naive_args &a, hipStream_t stream);

// local[a][b] = global[a][beam_offset + b]: the launch resolved but not enqueued (the kernel takes these six
// scalars as its parameters, in this order; a hipGraph kernel node is built from it -- dcs_bf_stream_tick_*_from_global)
struct bf_gather_launch {
    const void *func; // nullptr: nothing to launch
    dim3 grid, block;
    dcs_delay_vals *local;
    const dcs_delay_vals *global;
    uint32_t n_ant, nb_local, nb_total, beam_offset;
};
hipError_t bf_prepare_gather_beams(dcs_delay_vals *local, const dcs_delay_vals *global, uint32_t n_ant, uint32_t n_beams_local,
                                   uint32_t n_beams_total, uint32_t beam_offset, bf_gather_launch *out);
hipError_t bf_launch_gather_beams(dcs_delay_vals *local, const dcs_delay_vals *global,
                                  uint32_t n_ant, uint32_t n_beams_local, uint32_t n_beams_total,
                                  uint32_t beam_offset, hipStream_t stream);

hipError_t bf_warm_module();
hipError_t bf_launch_verify_div3(float D, float y, uint32_t *d_mismatches, hipStream_t stream);

#endif
