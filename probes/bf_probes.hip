// bf_probes.hip -- measurement apparatus, NOT part of the product ABI: the store-pattern,
// fill, sincos and whole-tensor-property probes behind profiles/ and DESIGN.md.  Built only into
// probes/libdcs_probes.so (include/dcs_probes.h), together with a -DDCS_PROBES build of the
// library's own sources in which dcs_bf_tuning::probe_nomath / probe_pace are honoured
// (libdcs_beamformer.so refuses them).  gfx950 only.
#include "../include/dcs_probes.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>

#include "../dc_sand_amd/csrc/bf_kernels.h" // bf_xcd_grouped (host-callable)
#include "../dc_sand_amd/csrc/bf_math.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256; // 4 waves of 64
constexpr int kReduceWaves = 8192;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <bool NT, typename T>
__device__ __forceinline__ void store_global(T *p, const T v)
{
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// Cache-policy variants of the 16-byte store: 0 plain, 1 nt, 2 sc1 (write-through),
// 3 sc0 sc1, 4 sc1 nt.
template <int MODE>
__device__ __forceinline__ void store16_mode(uintx4 *p, const uintx4 v)
{
    if constexpr (MODE == 0) {
        *p = v;
    } else if constexpr (MODE == 1) {
        __builtin_nontemporal_store(v, p);
    } else if constexpr (MODE == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else if constexpr (MODE == 3) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }
}

__global__ void __launch_bounds__(kBlock) bf_probe_sincos_kernel(int which, const float *x, size_t n,
                                                                 float *s, float *c)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float fs, fc;
    if (which == 4) { // the b16 arithmetic form: s[i] receives the packed (cos, sin) half2 word, c[i] is not written
        reinterpret_cast<uint32_t *>(s)[i] = dcs_sincos_half2(v);
        return;
    }
    if (which == 0) {
        dcs_sincos_fast<false>(v, &fs, &fc);
    } else if (which == 3) {
        dcs_sincos_fast<true>(v, &fs, &fc);
    } else if (which == 1) {
        sincosf(v, &fs, &fc); // __ocml_sincos_f32
    } else {
        double ds, dc;
        sincos((double)v, &ds, &dc);
        fs = (float)ds;
        fc = (float)dc;
    }
    s[i] = fs;
    c[i] = fc;
}

// The leanest possible store kernel: one 16-byte store per thread, no loop, no division;
// SPT > 1: each thread stores SPT times, a workgroup's stores KiB-interleaved over SPT rows
// `row16` 16-byte units apart (the generator's pattern without its arithmetic).
template <int MODE, int SPT>
__global__ void __launch_bounds__(kBlock) bf_probe_one_store_kernel(uintx4 *out, uint32_t row16, uint32_t tiles_per_row)
{
    const uintx4 v = {0x3f800000u, blockIdx.x, 0x3f800000u, threadIdx.x};
    if constexpr (SPT == 1) {
        store16_mode<MODE>(out + (size_t)blockIdx.x * kBlock + threadIdx.x, v);
    } else {
        // workgroup b: tile (b % tiles_per_row) of row band (b / tiles_per_row); 4 waves x SPT rows
        const uint32_t tile = blockIdx.x % (tiles_per_row & 0xffffu), band = blockIdx.x / (tiles_per_row & 0xffffu);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        uintx4 *p = out + ((size_t)band * 4u * SPT + wave) * row16 + (size_t)tile * 64u + lane;
        const uint32_t pace = tiles_per_row >> 16; // probe: units of 64 cycles slept before each store
        const uint32_t tiles = tiles_per_row & 0xffffu;
        (void)tiles;
        // pace >= 0x8000: instead of sleeping, wait for the wave's previous store to be
        // acknowledged before issuing the next (at most ONE store in flight per wave)
        const bool self_paced = (pace & 0x8000u) != 0u;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            if (self_paced) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                for (uint32_t k = 0; k < pace; k++) __builtin_amdgcn_s_sleep(1);
            }
            store16_mode<MODE>(p + (size_t)j * 4u * row16, v);
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(kBlock) bf_probe_fill_kernel(uintx4 *out, size_t n16)
{
    const uintx4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock)
        store_global<NT>(out + i, v);
}

// Store-pattern probe: the output seen as `rows` x `cols` chunks of 1 KiB (one
// wave store each).  A workgroup owns a rectangle of RB rows x QB chunks; its 4
// waves take the rectangle's chunks round-robin, chunk k -> (row k / QB,
// col k % QB).  Rectangles are numbered column-fastest (order 0) or row-fastest
// (order 1); xcd != 0 renumbers workgroups so that those sharing b % 8 (one
// XCD under round-robin dispatch) own consecutive rectangles.  No arithmetic.
template <int MODE>
__global__ void __launch_bounds__(1024) bf_probe_pattern_kernel(uintx4 *out, uint32_t rows, uint32_t cols,
                                                                  uint32_t QB, uint32_t RB, uint32_t order,
                                                                  uint32_t xcd)
{
    const uint32_t nq = (cols + QB - 1) / QB, nr = (rows + RB - 1) / RB;
    uint32_t b = blockIdx.x;
    const uint32_t G = gridDim.x;
    const uint32_t rot = xcd >> 4; // probe: rotate the rectangle column within groups of 8 (XCD <-> address affinity)
    if ((xcd & 1u) && (G % 8u) == 0u) b = (b % 8u) * (G / 8u) + b / 8u;
    uint32_t rq, rr;
    if (order == 0) {
        rq = b % nq;
        rr = b / nq;
    } else {
        rr = b % nr;
        rq = b / nr;
    }
    if (rot && (nq % 8u) == 0u) rq = (rq & ~7u) | ((rq + rot) & 7u);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uintx4 v = {0x3f800000u, b, 0x3f800000u, lane};
    const uint32_t nk = QB * RB;
    const uint32_t nwaves = blockDim.x >> 6;
    const bool contiguous = (xcd & 2u) != 0u; // a wave takes consecutive chunks instead of every nwaves-th
    const uint32_t per = (nk + nwaves - 1) / nwaves;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t k = contiguous ? wave * per + j : wave + j * nwaves;
        if (k >= nk) break;
        const uint32_t r = rr * RB + k / QB, q = rq * QB + k % QB;
        if (r < rows && q < cols) store16_mode<MODE>(out + ((uint64_t)r * cols + q) * 64u + lane, v);
    }
}

// Full-tensor property probe: per-wave partials of an order-independent checksum
// (sum of the 32-bit words, mod 2^64) and of max | |z|^2 - 1 | over fp32 (re, im)
// pairs; the host adds the partials.  part[2*w] = checksum, part[2*w+1] = float bits.
__global__ void __launch_bounds__(kBlock) bf_probe_reduce_kernel(const uintx4 *in, size_t n16,
                                                                 unsigned long long *part)
{
    unsigned long long sum = 0;
    float dev = 0.0f;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) {
        const uintx4 u = in[i];
        sum += (unsigned long long)u.x;
        sum += (unsigned long long)u.y;
        sum += (unsigned long long)u.z;
        sum += (unsigned long long)u.w;
        const float re0 = dcs_bits_f32(u.x), im0 = dcs_bits_f32(u.y), re1 = dcs_bits_f32(u.z), im1 = dcs_bits_f32(u.w);
        const float m0 = dcs_fmaf(re0, re0, im0 * im0) - 1.0f, m1 = dcs_fmaf(re1, re1, im1 * im1) - 1.0f;
        dev = fmaxf(dev, fmaxf(fabsf(m0), fabsf(m1)));
        if (m0 != m0 || m1 != m1) dev = INFINITY;
    }
    uint32_t lo = (uint32_t)sum, hi = (uint32_t)(sum >> 32);
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t olo = (uint32_t)__shfl_down((int)lo, o), ohi = (uint32_t)__shfl_down((int)hi, o);
        const unsigned long long t = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)ohi << 32) | olo);
        lo = (uint32_t)t;
        hi = (uint32_t)(t >> 32);
        dev = fmaxf(dev, __shfl_down(dev, o));
    }
    if ((threadIdx.x & 63u) == 0u) {
        const size_t w = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        part[2 * w] = ((unsigned long long)hi << 32) | lo;
        part[2 * w + 1] = (unsigned long long)dcs_f32_bits(dev);
    }
}


hipError_t bf_launch_probe_sincos(int which, const float *x, size_t n, float *s, float *c, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const dim3 grid((uint32_t)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(bf_probe_sincos_kernel, grid, dim3(kBlock), 0, stream, which, x, n, s, c);
    return hipGetLastError();
}

hipError_t bf_launch_probe_fill(void *out, size_t bytes, bool nontemporal, hipStream_t stream)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const dim3 grid(256 * 8);
    if (nontemporal)
        hipLaunchKernelGGL(bf_probe_fill_kernel<true>, grid, dim3(kBlock), 0, stream,
                           reinterpret_cast<uintx4 *>(out), n16);
    else
        hipLaunchKernelGGL(bf_probe_fill_kernel<false>, grid, dim3(kBlock), 0, stream,
                           reinterpret_cast<uintx4 *>(out), n16);
    return hipGetLastError();
}

hipError_t bf_launch_probe_pattern(void *out, uint32_t rows, uint32_t cols, uint32_t QB, uint32_t RB,
                                   uint32_t order, uint32_t xcd, int store_mode, uint32_t block_threads,
                                   hipStream_t stream)
{
    if (!rows || !cols || !QB || !RB) return hipErrorInvalidValue;
    if (block_threads == 0) block_threads = kBlock;
    if (block_threads % 64u || block_threads > 1024u) return hipErrorInvalidValue;
    const uint64_t nblk = (uint64_t)((cols + QB - 1) / QB) * ((rows + RB - 1) / RB);
    if (nblk > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)nblk), block(block_threads);
    uintx4 *o = reinterpret_cast<uintx4 *>(out);
    switch (store_mode) {
    case 0: hipLaunchKernelGGL(bf_probe_pattern_kernel<0>, grid, block, 0, stream, o, rows, cols, QB, RB, order, xcd); break;
    case 1: hipLaunchKernelGGL(bf_probe_pattern_kernel<1>, grid, block, 0, stream, o, rows, cols, QB, RB, order, xcd); break;
    case 2: hipLaunchKernelGGL(bf_probe_pattern_kernel<2>, grid, block, 0, stream, o, rows, cols, QB, RB, order, xcd); break;
    case 3: hipLaunchKernelGGL(bf_probe_pattern_kernel<3>, grid, block, 0, stream, o, rows, cols, QB, RB, order, xcd); break;
    case 4: hipLaunchKernelGGL(bf_probe_pattern_kernel<4>, grid, block, 0, stream, o, rows, cols, QB, RB, order, xcd); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t bf_launch_probe_reduce(const void *in, size_t bytes, unsigned long long *d_part, hipStream_t stream)
{
    // d_part: 2 * kReduceWaves entries
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(bf_probe_reduce_kernel, dim3(kReduceWaves / (kBlock / 64)), dim3(kBlock), 0, stream,
                       reinterpret_cast<const uintx4 *>(in), n16, d_part);
    return hipGetLastError();
}

hipError_t bf_launch_probe_one_store(void *out, size_t bytes, int store_mode_in, int stores_per_thread, uint32_t row_bytes,
                                     hipStream_t stream)
{
    const int store_mode = store_mode_in & 0xff;
    const uint32_t pace = (uint32_t)(store_mode_in >> 8) & 0xffffu; // probe: sleep units before each store
    uintx4 *o = reinterpret_cast<uintx4 *>(out);
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    if (stores_per_thread == 1) {
        const dim3 grid((uint32_t)(n16 / kBlock));
        if (store_mode == 1)
            hipLaunchKernelGGL((bf_probe_one_store_kernel<1, 1>), grid, dim3(kBlock), 0, stream, o, 0u, 1u);
        else
            hipLaunchKernelGGL((bf_probe_one_store_kernel<0, 1>), grid, dim3(kBlock), 0, stream, o, 0u, 1u);
        return hipGetLastError();
    }
    const uint32_t row16 = row_bytes / 16u, ntiles = row_bytes / 1024u;
    const uint32_t tiles = ntiles | (pace << 16);
    const size_t rows = bytes / row_bytes;
#define DCS_ONE_STORE(SPT)                                                                                   \
    {                                                                                                        \
        const dim3 grid((uint32_t)(rows / (4u * SPT) * ntiles));                                             \
        if (store_mode == 1)                                                                                 \
            hipLaunchKernelGGL((bf_probe_one_store_kernel<1, SPT>), grid, dim3(kBlock), 0, stream, o, row16, tiles); \
        else                                                                                                 \
            hipLaunchKernelGGL((bf_probe_one_store_kernel<0, SPT>), grid, dim3(kBlock), 0, stream, o, row16, tiles); \
    }
    switch (stores_per_thread) {
    case 2: DCS_ONE_STORE(2) break;
    case 3: DCS_ONE_STORE(3) break;
    case 4: DCS_ONE_STORE(4) break;
    case 8: DCS_ONE_STORE(8) break;
    case 16: DCS_ONE_STORE(16) break;
    case 64: DCS_ONE_STORE(64) break;
    default: return hipErrorInvalidValue;
    }
#undef DCS_ONE_STORE
    return hipGetLastError();
}

// fp32 matrix-core issue-rate probe: every wave runs `iters` x 16 MFMAs back to back on register operands
// (which = 0: v_mfma_f32_16x16x4_f32 with 2 accumulators, 1: the same with 4, 2: v_mfma_f32_32x32x2_f32 with 2;
//  3: 16x16x4 with a v_cvt between the MFMAs, as the coefficient-reuse beamformer has it).
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int WHICH>
__global__ void __launch_bounds__(kBlock) bf_probe_mfma_kernel(float *out, uint32_t iters, float seed)
{
    const float a0 = seed + (float)threadIdx.x * 0.001f, b0 = 1.0f - (float)(threadIdx.x & 7u) * 0.01f;
    float acc_sum = 0.0f;
    if constexpr (WHICH == 2) {
        floatx16 c0 = {0}, c1 = {0};
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0 + (float)j, b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0 + (float)j, c1, 0, 0, 0);
            }
        }
        acc_sum = c0[0] + c1[3];
    } else {
        floatx4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
        uint32_t v = threadIdx.x * 2654435761u;
        __shared__ float s_w[2 * 64 * 16];
        if constexpr (WHICH == 4) {
            for (uint32_t i = threadIdx.x; i < 2 * 64 * 16; i += kBlock) s_w[i] = seed + 0.001f * (float)i;
            __syncthreads();
        }
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float b = b0;
                if constexpr (WHICH == 4) { // both A operands from LDS, B converted: the beamformer's k-step
                    b = (float)(int8_t)(v & 0xffu);
                    const float bi = (float)(int8_t)((v >> 8) & 0xffu);
                    v = (v >> 16) | (v << 16);
                    const float wr = s_w[(threadIdx.x & 63u) + 64 * (2 * j)], wi = s_w[(threadIdx.x & 63u) + 64 * (2 * j + 1)];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wi, bi, c1, 0, 0, 0);
                    continue;
                }
                if constexpr (WHICH == 3) {
                    b = (float)(int8_t)(v & 0xffu);
                    v = (v >> 8) | (v << 24);
                }
                if (WHICH == 1 && (j & 1)) {
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + (float)j, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + (float)j, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a0, c1, 0, 0, 0);
                }
            }
        }
        acc_sum = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (acc_sum == 12345.678f) out[blockIdx.x] = acc_sum; // keep the chain alive
}

} // namespace

namespace {
template <bool NT, int PER>
__global__ void __launch_bounds__(256) probe_copy_kernel(const floatx4 *in, floatx4 *out, size_t n_vec)
{
    // a workgroup owns PER consecutive 4-KiB pieces; thread t copies vector t of each
    const size_t base = (size_t)blockIdx.x * (256u * PER) + threadIdx.x;
    floatx4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = base + 256u * i < n_vec ? in[base + 256u * i] : floatx4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (base + 256u * i < n_vec) store_global<NT>(out + base + 256u * i, v[i]);
}
} // namespace

extern "C" {

// MFMA issue-rate probe: grid x 4 waves, each `iters` x 16 MFMAs (which: see bf_probe_mfma_kernel).
int dcs_probe_mfma(int which, uint32_t blocks, uint32_t iters, float *d_out, void *stream)
{
    if (which < 0 || which > 4 || !d_out || blocks == 0) return DCS_ERR_INVALID_ARGUMENT;
    const dim3 grid(blocks), block(kBlock);
    switch (which) {
    case 0: hipLaunchKernelGGL(bf_probe_mfma_kernel<0>, grid, block, 0, as_stream(stream), d_out, iters, 1.0f); break;
    case 1: hipLaunchKernelGGL(bf_probe_mfma_kernel<1>, grid, block, 0, as_stream(stream), d_out, iters, 1.0f); break;
    case 2: hipLaunchKernelGGL(bf_probe_mfma_kernel<2>, grid, block, 0, as_stream(stream), d_out, iters, 1.0f); break;
    case 3: hipLaunchKernelGGL(bf_probe_mfma_kernel<3>, grid, block, 0, as_stream(stream), d_out, iters, 1.0f); break;
    default: hipLaunchKernelGGL(bf_probe_mfma_kernel<4>, grid, block, 0, as_stream(stream), d_out, iters, 1.0f); break;
    }
    return (int)hipGetLastError();
}


int dcs_probe_sincos(int which, const float *d_x, size_t n, float *d_sin, float *d_cos, void *stream)
{
    if (which < 0 || which > 4 || (n && (!d_x || !d_sin || !d_cos))) return DCS_ERR_INVALID_ARGUMENT;
    return (int)bf_launch_probe_sincos(which, d_x, n, d_sin, d_cos, as_stream(stream));
}

uint32_t dcs_probe_xcd_grouped(uint32_t w, uint32_t total, uint32_t group) { return bf_xcd_grouped(w, total, group); } // host only

int dcs_probe_copy(const void *d_in, void *d_out, size_t bytes, int store_mode, int per_thread, void *stream)
{
    if (!d_in || !d_out || (bytes % 16u) || (reinterpret_cast<uintptr_t>(d_in) & 15u) || (reinterpret_cast<uintptr_t>(d_out) & 15u))
        return DCS_ERR_INVALID_ARGUMENT;
    const size_t n_vec = bytes / 16u;
    const floatx4 *in = static_cast<const floatx4 *>(d_in);
    floatx4 *out = static_cast<floatx4 *>(d_out);
    hipStream_t s = as_stream(stream);
#define DCS_COPY(PER)                                                                                                  \
    {                                                                                                                  \
        const size_t blocks = (n_vec + 256u * PER - 1u) / (256u * PER);                                                \
        if (blocks > 0x7fffffffull) return DCS_ERR_INVALID_ARGUMENT;                                                   \
        if (store_mode)                                                                                                \
            hipLaunchKernelGGL((probe_copy_kernel<true, PER>), dim3((uint32_t)blocks), dim3(256), 0, s, in, out, n_vec); \
        else                                                                                                           \
            hipLaunchKernelGGL((probe_copy_kernel<false, PER>), dim3((uint32_t)blocks), dim3(256), 0, s, in, out, n_vec); \
    }
    if (per_thread == 1) DCS_COPY(1)
    else if (per_thread == 2) DCS_COPY(2)
    else if (per_thread == 4) DCS_COPY(4)
    else if (per_thread == 8) DCS_COPY(8)
    else return DCS_ERR_INVALID_ARGUMENT;
#undef DCS_COPY
    return (int)hipGetLastError();
}

int dcs_probe_fill(void *d_out, size_t bytes, int nontemporal, void *stream)
{
    if (!d_out && bytes) return DCS_ERR_INVALID_ARGUMENT;
    return (int)bf_launch_probe_fill(d_out, bytes, nontemporal != 0, as_stream(stream));
}

int dcs_probe_store_pattern(void *d_out, uint32_t rows, uint32_t cols_kib, uint32_t qb, uint32_t rb, int order,
                            int xcd_remap, int nontemporal, uint32_t block_threads, void *stream)
{
    if (!d_out) return DCS_ERR_INVALID_ARGUMENT;
    return (int)bf_launch_probe_pattern(d_out, rows, cols_kib, qb, rb, (uint32_t)order, (uint32_t)xcd_remap,
                                        nontemporal, block_threads, as_stream(stream));
}

int dcs_probe_one_store(void *d_out, size_t bytes, int store_mode, int stores_per_thread, uint32_t row_bytes, void *stream)
{
    if (!d_out || (bytes % 4096u) || stores_per_thread < 1 || stores_per_thread > 64) return DCS_ERR_INVALID_ARGUMENT;
    if (stores_per_thread > 1 && (row_bytes == 0 || (row_bytes % 1024u) || bytes % ((size_t)row_bytes * 4u * (size_t)stores_per_thread)))
        return DCS_ERR_INVALID_ARGUMENT;
    return (int)bf_launch_probe_one_store(d_out, bytes, store_mode, stores_per_thread, row_bytes, as_stream(stream));
}

int dcs_probe_reduce(const void *d_in, size_t bytes, uint64_t *checksum, float *max_modulus_dev, void *stream)
{
    if ((!d_in && bytes) || !checksum || !max_modulus_dev || (bytes % 16u)) return DCS_ERR_INVALID_ARGUMENT;
    const size_t n = 2 * (size_t)kReduceWaves;
    unsigned long long *d_part = nullptr;
    unsigned long long *h_part = new (std::nothrow) unsigned long long[n];
    if (!h_part) return (int)hipErrorOutOfMemory;
    hipError_t e = hipMalloc((void **)&d_part, n * sizeof(unsigned long long));
    if (e == hipSuccess) e = bf_launch_probe_reduce(d_in, bytes, d_part, as_stream(stream));
    if (e == hipSuccess) e = hipMemcpyAsync(h_part, d_part, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    if (d_part) (void)hipFree(d_part);
    if (e == hipSuccess) {
        uint64_t sum = 0;
        uint32_t devbits = 0;
        for (size_t w = 0; w < n / 2; w++) {
            sum += h_part[2 * w];
            if ((uint32_t)h_part[2 * w + 1] > devbits) devbits = (uint32_t)h_part[2 * w + 1];
        }
        *checksum = sum;
        std::memcpy(max_modulus_dev, &devbits, sizeof(float));
    }
    delete[] h_part;
    return (int)e;
}

} // extern "C"
