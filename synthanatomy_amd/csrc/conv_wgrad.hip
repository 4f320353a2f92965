// Weight gradient of the implicit-GEMM convolution family on MFMA (gfx950).
//
//   dW[co][ci][tap] += sum_m in[g(m, tap)][ci] * gout[o(m)][co]
//
// A GEMM whose reduction runs over the voxels m.  Block tile: 128 (tap,ci) rows x 128 co columns; the voxel range is
// split across blockIdx.z and combined with fp32 atomics (device scope, hardware global_atomic_add_f32).
// Both operands are stored voxel-major in HBM ([m][c]) but MFMA wants the reduction index contiguous per lane:
//   bf16: tiles stay voxel-major [64 m][128 c] in LDS (coalesced 16-byte loads -> ds_write_b128) and the fragments are
//         fetched with the gfx950 hardware transpose read ds_read_b64_tr_b16: per 16-lane group it turns a [4 m][16 c] block
//         into "lane = channel, 4 consecutive voxels"; two reads make one mfma_f32_16x16x32_bf16 operand.  32-byte chunks
//         are XOR-swizzled with (m & 7) so the 8 rows a half-wave touches cover all 64 banks exactly once;
//   f32 : tiles stay [m][c]; mfma_f32_16x16x4f32 takes one float per lane so fragments are plain ds_read_b32.
#include <stdio.h>
#include <stdlib.h>

#include "sa_common.h"

namespace sa {

struct WgradArgs {
    const void* in;
    const void* gout;
    float* dw;
    sa_conv_geom g;
    FastDiv dW, dH, dD, dTw, dThw, dCin;
    int32_t lut[SA_MAX_TAPS];
    int64_t s_row, s_red;
    uint32_t M, ntaps, ktot;       // ktot = ntaps * Cin
    uint32_t chunks_per_split, nchunks;
    uint32_t nkt, ntiles;
    uint32_t in_bytes, g_bytes;   // non-zero: both operands addressable with 32-bit buffer offsets (LDS-DMA loader)
    float* ws;            // [split][tile][co 128][kidx 128] partial tiles, or NULL -> fp32 atomics straight into dw
    float* db;            // bias gradient fused into the bf16 LDS-DMA kernels: db[co] += sum_m gout[m][co] (NULL: not wanted)
    // fused data gradient of a 1x1x1 / 128-channel layer (sa_conv1x1_backward): dg_out[m][ci] = (in[m][ci] > 0) * sum_co gout[m][co] W[co][ci]
    const void* dg_wpk;   // packed dgrad operand [128 ci][128 co] bf16
    void* dg_out;         // [M][128] bf16
    // halo kernel (3x3x3 stride 1, bf16, Cin = 128): the voxel range is walked in steps of 4 (H) x 16 (W) voxels
    uint32_t HQ, WP, nsteps, steps_per_split, halo;
    FastDiv dWP, dHQ;
};

template <typename T> struct WG;
template <> struct WG<bf16_t> { static constexpr int MK = 64; };
template <> struct WG<float> { static constexpr int MK = 16; };

// bf16 tile [64 m][128 c], 256 B per voxel row; byte offset of channel c (multiple of 4) in row m
__device__ __forceinline__ uint32_t roff(uint32_t m, uint32_t c) { return m * 256u + ((((c >> 4) ^ (m & 7u)) << 5) | ((c & 15u) << 1)); }

typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s_t lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p));
}

// LDS-DMA load the COMPILER DOES NOT SEE (round 4).  The transposing LDS reads of these kernels are intrinsics without alias information, so after a
// __builtin_amdgcn_raw_ptr_buffer_load_lds the waitcnt pass puts s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 -- i.e. the prefetch of step s + 1,
// issued at the top of step s, was WAITED FOR before step s multiplied anything (the "40 % of wave time parked on the per-step wait" of round 2 was this, not DMA
// throughput).  Issued from inline assembly the load stays in flight over the step; the kernel waits for it itself (dma_wait) before the barrier that hands the
// stage over.  Extra loads in flight only make the compiler's own vmcnt waits longer, never shorter.  lds_addr: wave-uniform LDS byte address (M0).
__device__ __forceinline__ void dma16_hidden(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_addr, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

// Column sums of a bf16 gradient tile [ROWS m][128 c] already in LDS (layout `roff`): thread = (channel pair tid & 63, row group tid >> 6).
template <int ROWS, int NGROUPS>
__device__ __forceinline__ void tile_colsum(const unsigned char* pg, uint32_t tid, float& s0, float& s1) {
    const uint32_t cp = tid & 63u, rg = tid >> 6;
#pragma unroll
    for (int m = 0; m < ROWS / NGROUPS + (ROWS % NGROUPS ? 1 : 0); ++m) {
        const uint32_t row = rg + (uint32_t)m * NGROUPS;
        if (ROWS % NGROUPS == 0 || row < (uint32_t)ROWS) {
            const uint32_t v = *(const uint32_t*)(pg + roff(row, 2u * cp));
            s0 += __uint_as_float(v << 16);
            s1 += __uint_as_float(v & 0xffff0000u);
        }
    }
}
// block-level finish: sum the row groups through LDS and add into db (one atomic per channel per block)
template <int NGROUPS>
__device__ __forceinline__ void colsum_finish(float* red /* [NGROUPS][128] */, uint32_t tid, float s0, float s1, float* db, uint32_t c_base, uint32_t c_valid) {
    const uint32_t cp = tid & 63u, rg = tid >> 6;
    red[rg * 128u + 2u * cp] = s0;
    red[rg * 128u + 2u * cp + 1u] = s1;
    __syncthreads();
    if (tid < 128u) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NGROUPS; ++r) t += red[r * 128 + tid];
        if (c_base + tid < c_valid) unsafeAtomicAdd(db + c_base + tid, t);
    }
}

struct RowPos {
    int32_t id, ih, iw;   // input base coordinate (before tap offset)
    int64_t ibase;        // linear input voxel (virtual when out of range)
    int64_t ovox;         // linear output voxel
    bool ok;
};

__device__ __forceinline__ RowPos decode_row(uint32_t m, const WgradArgs& a) {
    RowPos r;
    const sa_conv_geom& g = a.g;
    r.ok = m < a.M;
    const uint32_t mm = r.ok ? m : 0u;
    uint32_t q = fdiv(mm, a.dW);
    const uint32_t wmx = mm - q * g.Wm;
    uint32_t q2 = fdiv(q, a.dH);
    const uint32_t hmx = q - q2 * g.Hm;
    const uint32_t n = fdiv(q2, a.dD);
    const uint32_t dmx = q2 - n * g.Dm;
    r.id = (int32_t)dmx * g.in_mult[0] + g.in_off[0];
    r.ih = (int32_t)hmx * g.in_mult[1] + g.in_off[1];
    r.iw = (int32_t)wmx * g.in_mult[2] + g.in_off[2];
    r.ibase = (((int64_t)n * g.Di + r.id) * g.Hi + r.ih) * g.Wi + r.iw;
    r.ovox = (((int64_t)n * g.Do + (dmx * g.out_mult[0] + g.out_off[0])) * g.Ho + (hmx * g.out_mult[1] + g.out_off[1])) * g.Wo +
             (wmx * g.out_mult[2] + g.out_off[2]);
    return r;
}

struct TapPos {
    int32_t od, oh, ow;
    int64_t off;
    uint32_t c0;
    bool ok;
};

__device__ __forceinline__ TapPos decode_k(uint32_t kidx0, const WgradArgs& a) {
    TapPos t;
    const sa_conv_geom& g = a.g;
    const uint32_t tap = fdiv(kidx0, a.dCin);
    t.c0 = kidx0 - tap * g.Cin;
    t.ok = tap < a.ntaps;
    const uint32_t td = fdiv(tap, a.dThw);
    const uint32_t t2 = tap - td * a.dThw.d;
    const uint32_t th = fdiv(t2, a.dTw);
    const uint32_t tw = t2 - th * a.dTw.d;
    t.od = (int32_t)td * g.tap_step[0];
    t.oh = (int32_t)th * g.tap_step[1];
    t.ow = (int32_t)tw * g.tap_step[2];
    t.off = ((int64_t)t.od * g.Hi + t.oh) * g.Wi + t.ow;
    return t;
}

__device__ __forceinline__ bool in_range(const RowPos& r, const TapPos& t, const sa_conv_geom& g) {
    return r.ok && t.ok && (uint32_t)(r.id + t.od) < (uint32_t)g.Di && (uint32_t)(r.ih + t.oh) < (uint32_t)g.Hi &&
           (uint32_t)(r.iw + t.ow) < (uint32_t)g.Wi;
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int MK = WG<T>::MK;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int TILE_BYTES = IS_BF16 ? 128 * 128 : MK * 128 * 4;  // 16 KB / 8 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                   // 2 x TILE
    unsigned char* sG = smem + 2 * TILE_BYTES;  // 2 x TILE

    const sa_conv_geom& g = a.g;
    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ go = (const T*)a.gout;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 1, wn = wave & 1u;
    // XCD-aware order: block b runs on XCD b % 8 (observed dispatch; speed only).  All (tap, co) tiles of one voxel split are
    // issued back-to-back on ONE XCD so that the X / G voxel rows they all re-read are served by that XCD's L2 instead
    // of being fetched once per tap from HBM.
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const uint32_t tile = seq % a.ntiles, split = (seq / a.ntiles) * 8u + xcd;
    const uint32_t kt = tile % a.nkt, ct = tile / a.nkt;
    const uint32_t chunk0 = split * a.chunks_per_split;
    uint32_t chunk1 = chunk0 + a.chunks_per_split;
    if (chunk1 > a.nchunks) chunk1 = a.nchunks;
    if (chunk0 >= chunk1) return;

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------ loader roles
    // bf16: thread -> 16-byte channel vector v = tid&15 (8 channels) of voxel rows (tid>>4) + 16j (j=0..3), both operands
    // f32 : thread -> voxel rows (tid>>5) + 8j (j=0,1), 16-byte channel vector v = tid&31 of both operands
    TapPos tp[2];
    uint32_t gco[2];
    if constexpr (IS_BF16) {
        const uint32_t v = tid & 15u;
        tp[0] = decode_k(kt * 128u + v * 8u, a);
        if (kt * 128u + v * 8u >= a.ktot) tp[0].ok = false;
        tp[1] = tp[0];
        gco[0] = gco[1] = ct * 128u + v * 8u;
    } else {
        const uint32_t v = tid & 31u;
        tp[0] = decode_k(kt * 128u + v * 4u, a);
        if (kt * 128u + v * 4u >= a.ktot) tp[0].ok = false;
        tp[1] = tp[0];
        gco[0] = gco[1] = ct * 128u + v * 4u;
    }

    u32x4 rx[4], rg[4];
    auto gload = [&](uint32_t chunk) __attribute__((always_inline)) {
        const uint32_t mb = chunk * MK;
        if constexpr (IS_BF16) {
            const bool cok = gco[0] + 8u <= (uint32_t)g.Cout;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                RowPos r = decode_row(mb + (tid >> 4) + 16u * j, a);
                rx[j] = in_range(r, tp[0], g) ? *(const u32x4*)(in + (r.ibase + tp[0].off) * g.Cin + tp[0].c0) : (u32x4){0u, 0u, 0u, 0u};
                rg[j] = (r.ok && cok) ? *(const u32x4*)(go + r.ovox * g.Cout + gco[0]) : (u32x4){0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                RowPos r = decode_row(mb + (tid >> 5) + 8u * j, a);
                rx[j] = in_range(r, tp[0], g) ? *(const u32x4*)(in + (r.ibase + tp[0].off) * g.Cin + tp[0].c0) : (u32x4){0u, 0u, 0u, 0u};
                const bool cok = gco[0] + 4u <= (uint32_t)g.Cout;
                rg[j] = (r.ok && cok) ? *(const u32x4*)(go + r.ovox * g.Cout + gco[0]) : (u32x4){0u, 0u, 0u, 0u};
            }
        }
    };
    auto lstore = [&](uint32_t buf) __attribute__((always_inline)) {
        unsigned char* px = sX + buf * TILE_BYTES;
        unsigned char* pg = sG + buf * TILE_BYTES;
        if constexpr (IS_BF16) {
            const uint32_t v = tid & 15u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t o = roff((tid >> 4) + 16u * j, v * 8u);
                *(u32x4*)(px + o) = rx[j];
                *(u32x4*)(pg + o) = rg[j];
            }
        } else {
            const uint32_t v = tid & 31u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t m = (tid >> 5) + 8u * j;
                const uint32_t o = m * 512u + (((v * 4u) ^ ((m & 1u) << 4)) << 2);
                *(u32x4*)(px + o) = rx[j];
                *(u32x4*)(pg + o) = rg[j];
            }
        }
    };

    gload(chunk0);
    lstore(0);
    __syncthreads();
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    for (uint32_t c = chunk0; c < chunk1; ++c) {
        const uint32_t buf = (c - chunk0) & 1u;
        gload((c + 1) < chunk1 ? c + 1 : c);  // unconditional (clamped) prefetch keeps the staging registers in VGPRs
        const unsigned char* px = sX + buf * TILE_BYTES;
        const unsigned char* pg = sG + buf * TILE_BYTES;
        if constexpr (IS_BF16) {
            // lane (group gq = lane>>4, s = lane&15) addresses voxel row 4*gq + (s>>2) and channels 4*(s&3)..+3 of each 16x(4 m) block
            const uint32_t trow = fq * 4u + (frow >> 2), tcol = (frow & 3u) * 4u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                short8_t xf[4], gf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v4s_t lo = lds_tr16(px + roff(ks * 32 + trow, wm * 64 + i * 16 + tcol));
                    const v4s_t hi = lds_tr16(px + roff(ks * 32 + 16 + trow, wm * 64 + i * 16 + tcol));
                    xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4s_t lo = lds_tr16(pg + roff(ks * 32 + trow, wn * 64 + j * 16 + tcol));
                    const v4s_t hi = lds_tr16(pg + roff(ks * 32 + 16 + trow, wn * 64 + j * 16 + tcol));
                    gf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], gf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t m = kk * 4u + fq;
                const uint32_t sw = (m & 1u) << 4;
                float xf[4], gf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = *(const float*)(px + m * 512u + (((wm * 64 + i * 16 + frow) ^ sw) << 2));
#pragma unroll
                for (int j = 0; j < 4; ++j) gf[j] = *(const float*)(pg + m * 512u + (((wn * 64 + j * 16 + frow) ^ sw) << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[i], gf[j], acc[i][j], 0, 0, 0);
            }
        }
        lstore(buf ^ 1u);
        __syncthreads();
    }

    // ---- epilogue: lane holds rows kidx = .. + fq*4 + r (4 consecutive ci of one tap), column co = .. + frow
    if (a.ws) {
        // partial tile to the workspace (plain 16-byte stores); wgrad_reduce_kernel sums the splits and scatters into dw
        float* wt = a.ws + ((size_t)split * a.ntiles + tile) * (128 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4_t*)(wt + (wn * 64 + j * 16 + frow) * 128 + wm * 64 + i * 16 + fq * 4) = acc[i][j];
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t k0 = kt * 128u + wm * 64u + i * 16u + fq * 4u;
        if (k0 >= a.ktot) continue;
        const uint32_t tap = fdiv(k0, a.dCin);
        const uint32_t c0 = k0 - tap * g.Cin;
        const int64_t tapo = a.lut[tap];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t co = ct * 128u + wn * 64u + j * 16u + frow;
            if (co >= (uint32_t)g.cout_valid) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c0 + r >= (uint32_t)g.cin_valid) continue;
                unsafeAtomicAdd(a.dw + co * a.s_row + (int64_t)(c0 + r) * a.s_red + tapo, acc[i][j][r]);
            }
        }
    }
}

template <typename T, bool DG = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv_wgrad_dma_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // NW = 8 (bf16, no fused data gradient): the same 128 x 128 tile worked by eight waves of 32 x 64 (<= 128 VGPRs: two blocks = four waves per
    // SIMD) -- a chunk is only 64 voxels deep, so a four-wave block issues 32 MFMAs per wave between barriers and needs the extra waves to cover them
    static_assert(NW == 4 || (NW == 8 && sizeof(T) == 2 && !DG), "4 waves, or 8 for the plain bf16 kernel");
    constexpr int NIF = 16 / NW;                 // 16-row fragments of the (tap, ci) dimension per wave
    constexpr int MK = WG<T>::MK;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int TILE_BYTES = IS_BF16 ? 128 * 128 : MK * 128 * 4;  // 16 KB / 8 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                   // 2 x TILE
    unsigned char* sG = smem + 2 * TILE_BYTES;  // 2 x TILE

    const sa_conv_geom& g = a.g;
    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ go = (const T*)a.gout;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 1, wn = wave & 1u;
    // XCD-aware order: block b runs on XCD b % 8 (observed dispatch; speed only).  All (tap, co) tiles of one voxel split are
    // issued back-to-back on ONE XCD so that the X / G voxel rows they all re-read are served by that XCD's L2 instead
    // of being fetched once per tap from HBM.
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const uint32_t tile = seq % a.ntiles, split = (seq / a.ntiles) * 8u + xcd;
    const uint32_t kt = tile % a.nkt, ct = tile / a.nkt;
    const uint32_t chunk0 = split * a.chunks_per_split;
    uint32_t chunk1 = chunk0 + a.chunks_per_split;
    if (chunk1 > a.nchunks) chunk1 = a.nchunks;
    if (chunk0 >= chunk1) return;

    float4_t acc[NIF][4];
#pragma unroll
    for (int i = 0; i < NIF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------ LDS-DMA loader (buffer_load ... lds)
    // A 1 KiB piece is 4 voxel rows x 256 B (bf16) or 2 rows x 512 B (fp32); wave w owns pieces 4w..4w+3 (bf16) / 2w, 2w+1 (fp32)
    // of both tiles.  The DMA writes lane-linearly, so the swizzle is applied to the SOURCE column: bf16 lane l fetches 16-byte
    // vector (((l&15)>>1) ^ (m&7))*2 + (l&1) of row m (two variants: even / odd piece), fp32 lane l fetches (l&31) ^ ((l>>5)<<2).
    constexpr int SZ = sizeof(T);
    constexpr int NPIECE = (IS_BF16 ? 16 : 8) / NW;      // pieces per wave per tile
    constexpr int ROWS_PP = IS_BF16 ? 4 : 2;     // rows per piece
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)a.gout, 0, (int)a.g_bytes, 0x00020000);
    const uint32_t prow = IS_BF16 ? (lane >> 4) : (lane >> 5);
    TapPos tp[2];
    uint32_t gcol[2];
    bool gok[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        uint32_t vec;  // source 16-byte vector of the 128-channel row
        if constexpr (IS_BF16) vec = ((((lane & 15u) >> 1) ^ ((v * 4u + prow) & 7u)) << 1) | (lane & 1u);
        else vec = (lane & 31u) ^ (prow << 2);
        constexpr int VE = 16 / SZ;
        tp[v] = decode_k(kt * 128u + vec * VE, a);
        if (kt * 128u + vec * VE >= a.ktot) tp[v].ok = false;
        gcol[v] = ct * 128u + vec * VE;
        gok[v] = gcol[v] + VE <= (uint32_t)g.Cout;
    }
    const uint32_t sx_addr = __builtin_amdgcn_readfirstlane(lds_address(sX)), sg_addr = __builtin_amdgcn_readfirstlane(lds_address(sG));
    auto issue = [&](uint32_t chunk, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t mb = chunk * MK;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const uint32_t piece = wave * NPIECE + j;
            const int v = IS_BF16 ? (j & 1) : 0;
            RowPos r = decode_row(mb + piece * ROWS_PP + prow, a);
            const bool okx = in_range(r, tp[v], g);
            const uint32_t xoff = okx ? (uint32_t)(r.ibase + tp[v].off) * (uint32_t)(g.Cin * SZ) + tp[v].c0 * SZ : 0xfffffff0u;
            const uint32_t goff = (r.ok && gok[v]) ? (uint32_t)r.ovox * (uint32_t)(g.Cout * SZ) + gcol[v] * SZ : 0xfffffff0u;
            if constexpr (IS_BF16) {   // (invisible to the compiler: see dma16_hidden; the fp32 kernel reads LDS with plain loads, which carry alias information)
                dma16_hidden(rX, sx_addr + buf * TILE_BYTES + piece * 1024, xoff);
                dma16_hidden(rG, sg_addr + buf * TILE_BYTES + piece * 1024, goff);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)(sX + buf * TILE_BYTES + piece * 1024), 16, xoff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (__attribute__((address_space(3))) void*)(sG + buf * TILE_BYTES + piece * 1024), 16, goff, 0, 0, 0);
            }
        }
    };

    issue(chunk0, 0);
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    // DG (1x1x1, 128 channels): the block also produces the data gradient of its rows from the SAME staged gradient tile,
    //   dX[m][ci] = (x[m][ci] > 0) * sum_co g[m][co] W[co][ci],
    // wave (wm, wn) = rows wm*32..+31 x input channels wn*64..+63.  The weights are the MFMA A operand, held in registers for the whole
    // block (16 fragments); the gradient tile is the B operand, read from the transposing-read layout with plain 16-byte reads
    // (channel chunk (c >> 4) ^ (m & 7): the 16 rows x 4 k-groups of a fragment still hit 16 distinct 16-byte bank groups).
    u32x4 dgw[DG ? 4 : 1][DG ? 4 : 1];
    if constexpr (DG) {
        const bf16_t* wp = (const bf16_t*)a.dg_wpk;
#pragma unroll
        for (int ic = 0; ic < 4; ++ic)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dgw[ic][ks] = *(const u32x4*)(wp + (wn * 64 + ic * 16 + frow) * 128 + ks * 32 + fq * 8);
    }
    dma_wait();
    __syncthreads();
    const bool do_db = IS_BF16 && a.db != nullptr && kt == 0;   // the gradient rows of this split pass through exactly one kt = 0 block per co tile
    float bs0 = 0.f, bs1 = 0.f;
    for (uint32_t c = chunk0; c < chunk1; ++c) {
        const uint32_t buf = (c - chunk0) & 1u;
        if (c + 1 < chunk1) issue(c + 1, buf ^ 1u);
        const unsigned char* px = sX + buf * TILE_BYTES;
        const unsigned char* pg = sG + buf * TILE_BYTES;
        if constexpr (IS_BF16) {
            if (do_db) tile_colsum<64, NW>(pg, tid, bs0, bs1);
            if constexpr (DG) {
                float4_t ad[4][2];
#pragma unroll
                for (int ic = 0; ic < 4; ++ic)
#pragma unroll
                    for (int jr = 0; jr < 2; ++jr) ad[ic][jr] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    u32x4 gb[2];
#pragma unroll
                    for (int jr = 0; jr < 2; ++jr) {
                        const uint32_t m = wm * 32 + jr * 16 + frow, cc = ks * 32 + fq * 8;
                        gb[jr] = *(const u32x4*)(pg + m * 256u + ((((cc >> 4) ^ (m & 7u)) << 5) | ((cc & 15u) << 1)));
                    }
#pragma unroll
                    for (int ic = 0; ic < 4; ++ic)
#pragma unroll
                        for (int jr = 0; jr < 2; ++jr)
                            ad[ic][jr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const short8_t*)&dgw[ic][ks], *(const short8_t*)&gb[jr], ad[ic][jr], 0, 0, 0);
                }
                // lane = 4 channels (fq*4 + r) of 4 fragments for row frow -> quarter transpose -> 16 consecutive channels wn*64 + fq*16 ..
#pragma unroll
                for (int jr = 0; jr < 2; ++jr) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t0 = ad[0][jr][r], t1 = ad[1][jr][r], t2 = ad[2][jr][r], t3 = ad[3][jr][r];
                        quarter_transpose(t0, t1, t2, t3);
                        v[r] = t0; v[4 + r] = t1; v[8 + r] = t2; v[12 + r] = t3;
                    }
                    const uint32_t mrow = wm * 32 + jr * 16 + frow, c0 = wn * 64 + fq * 16;
                    const uint32_t gm = c * MK + mrow;
                    // ReLU mask from the staged activation tile: one 32-byte chunk = these 16 channels
                    const unsigned char* xm = px + mrow * 256u + (((c0 >> 4) ^ (mrow & 7u)) << 5);
                    const u32x4 h0 = *(const u32x4*)xm, h1 = *(const u32x4*)(xm + 16);
                    u32x4 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // bf16 > 0  <=>  sign bit clear and not zero
                        const uint32_t ha = h0[e], hb = h1[e];
                        const float a0 = ((ha & 0xffffu) != 0 && !(ha & 0x8000u)) ? v[2 * e] : 0.f, a1 = ((ha >> 16) != 0 && !(ha & 0x80000000u)) ? v[2 * e + 1] : 0.f;
                        const float b0 = ((hb & 0xffffu) != 0 && !(hb & 0x8000u)) ? v[8 + 2 * e] : 0.f, b1 = ((hb >> 16) != 0 && !(hb & 0x80000000u)) ? v[8 + 2 * e + 1] : 0.f;
                        o0[e] = (uint32_t)f32_to_bf16(a0) | ((uint32_t)f32_to_bf16(a1) << 16);
                        o1[e] = (uint32_t)f32_to_bf16(b0) | ((uint32_t)f32_to_bf16(b1) << 16);
                    }
                    if (gm < a.M) {
                        bf16_t* op = (bf16_t*)a.dg_out + (size_t)gm * 128 + c0;
                        *(u32x4*)op = o0;
                        *(u32x4*)(op + 8) = o1;
                    }
                }
            }
            // lane (group gq = lane>>4, s = lane&15) addresses voxel row 4*gq + (s>>2) and channels 4*(s&3)..+3 of each 16x(4 m) block
            const uint32_t trow = fq * 4u + (frow >> 2), tcol = (frow & 3u) * 4u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                short8_t xf[NIF], gf[4];
#pragma unroll
                for (int i = 0; i < NIF; ++i) {
                    const v4s_t lo = lds_tr16(px + roff(ks * 32 + trow, wm * (16 * NIF) + i * 16 + tcol));
                    const v4s_t hi = lds_tr16(px + roff(ks * 32 + 16 + trow, wm * (16 * NIF) + i * 16 + tcol));
                    xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4s_t lo = lds_tr16(pg + roff(ks * 32 + trow, wn * 64 + j * 16 + tcol));
                    const v4s_t hi = lds_tr16(pg + roff(ks * 32 + 16 + trow, wn * 64 + j * 16 + tcol));
                    gf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < NIF; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], gf[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (NW == 4) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t m = kk * 4u + fq;
                const uint32_t sw = (m & 1u) << 4;
                float xf[4], gf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = *(const float*)(px + m * 512u + (((wm * 64 + i * 16 + frow) ^ sw) << 2));
#pragma unroll
                for (int j = 0; j < 4; ++j) gf[j] = *(const float*)(pg + m * 512u + (((wn * 64 + j * 16 + frow) ^ sw) << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[i], gf[j], acc[i][j], 0, 0, 0);
            }
        }
        dma_wait();       // the next chunk's tiles have landed (dma16_hidden: the compiler does not wait for them)
        __syncthreads();  // ... in every wave, and this buffer is free
    }

    if (do_db) colsum_finish<NW>((float*)smem, tid, bs0, bs1, a.db, ct * 128u, (uint32_t)g.cout_valid);   // (the loop's last barrier freed the tiles)
    // ---- epilogue: lane holds rows kidx = .. + fq*4 + r (4 consecutive ci of one tap), column co = .. + frow
    if (a.ws) {
        // partial tile to the workspace (plain 16-byte stores); wgrad_reduce_kernel sums the splits and scatters into dw
        float* wt = a.ws + ((size_t)split * a.ntiles + tile) * (128 * 128);
#pragma unroll
        for (int i = 0; i < NIF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4_t*)(wt + (wn * 64 + j * 16 + frow) * 128 + wm * (16 * NIF) + i * 16 + fq * 4) = acc[i][j];
        return;
    }
#pragma unroll
    for (int i = 0; i < NIF; ++i) {
        const uint32_t k0 = kt * 128u + wm * (uint32_t)(16 * NIF) + i * 16u + fq * 4u;
        if (k0 >= a.ktot) continue;
        const uint32_t tap = fdiv(k0, a.dCin);
        const uint32_t c0 = k0 - tap * g.Cin;
        const int64_t tapo = a.lut[tap];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t co = ct * 128u + wn * 64u + j * 16u + frow;
            if (co >= (uint32_t)g.cout_valid) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c0 + r >= (uint32_t)g.cin_valid) continue;
                unsafeAtomicAdd(a.dw + co * a.s_row + (int64_t)(c0 + r) * a.s_red + tapo, acc[i][j][r]);
            }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Halo weight gradient (3x3x3, stride 1, `same`; bf16; Cin = 128): the im2col-order kernel above reads every activation and
// every output-gradient row once per TAP (27 x), which makes it L2-bound (~10 TB/s of LDS-DMA traffic at 550-650 TFLOP/s).
// Here a block owns one (kd, kh) pair and a range of voxel steps; a step is 4 (H) x 16 (W) voxels of one plane: the gradient tile
// [64][128 co] and the activation halo tile [4 x 18][128 ci] (one extra voxel on either side in W, zero-filled outside the
// volume by the DMA) are staged once and feed the THREE kw taps (PHS = 8: steps of 8 x 16 voxels, half the barriers).  12 waves: wave = (kw, 64-ci half, 64-co half), each a
// 64 x 64 accumulator like the kernel above; the tap only shifts the halo rows a wave transposes out of LDS.
// Operand traffic drops 3 x; the partial tiles go to the same workspace layout (tile = tap) and the same reduce kernel.
template <int PHS>  // voxel rows (H) per step: 4 or 8
__global__ __launch_bounds__(768) void conv_wgrad_halo_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int XP = PHS * 18 / 4, GP = PHS * 4, NP = XP + GP, PPW = (NP + 11) / 12;   // 1 KiB pieces: activation halo, gradient
    constexpr int XT = XP * 1024, GT = GP * 1024, STAGE = XT + GT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const sa_conv_geom& g = a.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tw = wave >> 2, wm = (wave >> 1) & 1u, wn = wave & 1u;
    // block -> (split, kd*3+kh, co tile); an XCD (block b -> XCD b % 8) walks a contiguous range of splits and runs the nine
    // (kd, kh) blocks of a split back to back, so their common rows are L2 hits
    const uint32_t nct = a.ntiles / a.nkt;
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const uint32_t per_split = 9u * nct;
    const uint32_t nsplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
    const uint32_t spx = (nsplit + 7u) >> 3;                       // splits per XCD
    const uint32_t sl = seq / per_split, rem = seq - sl * per_split;
    const uint32_t split = xcd * spx + sl;
    if (sl >= spx || split >= nsplit) return;
    const uint32_t tdth = rem % 9u, ct = rem / 9u;
    const uint32_t td = tdth / 3u, th = tdth - td * 3u;
    const uint32_t step0 = split * a.steps_per_split;
    uint32_t step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)a.gout, 0, (int)a.g_bytes, 0x00020000);
    // NP one-KiB pieces per step (a piece = 4 rows x 256 B): wave w moves pieces w, w+12, w+24, ...
    // Source-side swizzle as in the kernel above: lane l fetches 16-byte vector ((((l&15)>>1) ^ (row&7))<<1 | (l&1)) of its row.
    const uint32_t prow = lane >> 4;
    int32_t p_h[PPW], p_w[PPW];      // row coordinates relative to the step origin (h0, w0) [input coordinates for activation pieces]
    uint32_t p_off[PPW];           // byte offset relative to the step's base voxel, + the lane's vector
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const uint32_t q = wave + 12u * i;
        const bool isx = q < (uint32_t)XP;
        const uint32_t r = (isx ? q : q - (uint32_t)XP) * 4u + prow;     // tile row
        const uint32_t vec = ((((lane & 15u) >> 1) ^ (r & 7u)) << 1) | (lane & 1u);
        if (isx) {
            const uint32_t hh = r / 18u, ww = r - hh * 18u;
            p_h[i] = (int32_t)hh + (int32_t)th + g.in_off[1];
            p_w[i] = (int32_t)ww + g.in_off[2];
            p_off[i] = (uint32_t)((p_h[i] * g.Wi + p_w[i]) * (g.Cin * 2)) + vec * 16u;
        } else {
            p_h[i] = (int32_t)(r >> 4);
            p_w[i] = (int32_t)(r & 15u);
            p_off[i] = (uint32_t)((p_h[i] * g.Wo + p_w[i]) * (g.Cout * 2)) + ct * 256u + vec * 16u;
        }
    }
    auto issue = [&](uint32_t step, uint32_t buf) __attribute__((always_inline)) {
        unsigned char* ps = smem + buf * STAGE;
        uint32_t q1 = fdiv(step, a.dWP);
        const uint32_t wp = step - q1 * a.WP;
        uint32_t q2 = fdiv(q1, a.dHQ);
        const uint32_t hq = q1 - q2 * a.HQ;
        const uint32_t n = fdiv(q2, a.dD);
        const uint32_t d = q2 - n * (uint32_t)g.Dm;
        const int32_t h0 = (int32_t)hq * PHS, w0 = (int32_t)wp * 16;
        const int32_t id = (int32_t)d + g.in_off[0] + (int32_t)td;
        const bool dok = (uint32_t)id < (uint32_t)g.Di;
        const uint32_t xbase = (uint32_t)((((int32_t)n * g.Di + id) * g.Hi + h0) * g.Wi + w0) * (uint32_t)(g.Cin * 2);
        const uint32_t gbase = (uint32_t)((((int32_t)n * g.Do + (int32_t)d) * g.Ho + h0) * g.Wo + w0) * (uint32_t)(g.Cout * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint32_t q = wave + 12u * i;
            if (q >= (uint32_t)NP) break;
            const int32_t hh = h0 + p_h[i], ww = w0 + p_w[i];
            if (q < (uint32_t)XP) {
                const bool ok = dok && (uint32_t)hh < (uint32_t)g.Hi && (uint32_t)ww < (uint32_t)g.Wi;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)(ps + q * 1024), 16, ok ? xbase + p_off[i] : 0xfffffff0u, 0, 0, 0);
            } else {
                const bool ok = (uint32_t)hh < (uint32_t)g.Ho && (uint32_t)ww < (uint32_t)g.Wo;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (__attribute__((address_space(3))) void*)(ps + XT + (q - (uint32_t)XP) * 1024), 16, ok ? gbase + p_off[i] : 0xfffffff0u, 0, 0,
                                                         0);
            }
        }
    };

    issue(step0, 0);
    __syncthreads();
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    // transposing reads: lane (gq = lane>>4, s = lane&15) addresses row 4*gq + (s>>2), channels 4*(s&3)..+3 of a [16 rows][16 ch] block
    const uint32_t trow = fq * 4u + (frow >> 2), tcol = (frow & 3u) * 4u;
    const bool do_db = a.db != nullptr && tdth == 0u;   // one of the nine (kd, kh) blocks of a split also sums the gradient rows
    float bs0 = 0.f, bs1 = 0.f;
    for (uint32_t st = step0; st < step1; ++st) {
        const uint32_t buf = (st - step0) & 1u;
        if (st + 1 < step1) issue(st + 1, buf ^ 1u);
        const unsigned char* px = smem + buf * STAGE;
        const unsigned char* pg = px + XT;
        if (do_db) tile_colsum<PHS * 16, 12>(pg, tid, bs0, bs1);
#pragma unroll
        for (int ks = 0; ks < PHS / 2; ++ks) {
            short8_t xf[4], gf[4];
            // k = voxel (ph, pw) = (2*ks + half, 0..15)  ->  halo row ph*18 + pw + kw, gradient row ph*16 + pw
            const uint32_t xr0 = (uint32_t)(2 * ks) * 18u + tw + trow, xr1 = xr0 + 18u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4s_t lo = lds_tr16(px + roff(xr0, wm * 64 + i * 16 + tcol));
                const v4s_t hi = lds_tr16(px + roff(xr1, wm * 64 + i * 16 + tcol));
                xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4s_t lo = lds_tr16(pg + roff(ks * 32 + trow, wn * 64 + j * 16 + tcol));
                const v4s_t hi = lds_tr16(pg + roff(ks * 32 + 16 + trow, wn * 64 + j * 16 + tcol));
                gf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], gf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // drains the DMA (vmcnt(0)) and frees this buffer
    }
    if (do_db) colsum_finish<12>((float*)smem, tid, bs0, bs1, a.db, ct * 128u, (uint32_t)g.cout_valid);
    // partial tile of tap (kd, kh, kw) -> workspace [split][tile = tap + 27*ct][co 128][ci 128]
    const uint32_t tile = (td * 3u + th) * 3u + tw + a.nkt * ct;
    float* wt = a.ws + ((size_t)split * a.ntiles + tile) * (128 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4_t*)(wt + (wn * 64 + j * 16 + frow) * 128 + wm * 64 + i * 16 + fq * 4) = acc[i][j];
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Halo weight gradient, nine taps per block.  The three-tap kernel above is bound by the operand stream (L2 -> LDS fabric,
// measured: its loads alone take 4.7 of its 5.2 ms on the C = 128 layer).  Here a block owns one kd, one 64-channel half of the input
// channels and a range of voxel steps; a step is 8 (H) x 16 (W) voxels of one plane: the gradient tile [128][128 co] (32 pieces) and
// the activation halo tile [10 x 18][64 ci] (128-byte rows, 23 pieces) feed all NINE (kh, kw) taps.  8 waves: wave = (16-ci fragment,
// 64-co half), 9 taps x 4 co fragments = 36 accumulators (144 VGPRs), 144 MFMAs per step and barrier (32 above).
// Operand bytes per FLOP: 2.9 KB/MFLOP against 5.4 (three taps) and 15.6 (im2col order).
// Measured dead end: touching the lines of step st+2 ahead of time (4-byte LDS-DMA lanes into a scratch slot, counted vmcnt) made it 10 %
// slower -- the 40 % of wave time parked on the per-step wait is DMA throughput, not HBM latency.
// 128-byte rows: 32-byte chunk c of row m sits at chunk c ^ ((m >> 1) & 3), which gives the transposing reads of 8 consecutive rows 8
// distinct 32-byte bank groups (rows of equal parity share a 256-byte bank line).
__device__ __forceinline__ uint32_t roff128(uint32_t m, uint32_t c) { return m * 128u + ((((c >> 4) ^ ((m >> 1) & 3u)) << 5) | ((c & 15u) << 1)); }

// NW = 16: the same step worked by sixteen waves (36 x 2 accumulator fragments each, <= 128 VGPRs -> four waves per SIMD from ONE block: the
// 110 KiB of LDS allow no second block, and two waves per SIMD leave the MFMA pipe idle through every DMA wait / barrier).
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_halo9_kernel(const WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NJ = 32 / NW;                                           // 16-column output-channel fragments per wave (4 or 2)
    constexpr int XP = 23, GP = 32, NP = XP + GP, PPW = (NP + NW - 1) / NW;   // 1 KiB pieces per step, dealt round-robin to the waves
    constexpr int XT = XP * 1024, GT = GP * 1024, STAGE = XT + GT;      // 55 KiB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const sa_conv_geom& g = a.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fi = wave & 3u, wn = wave >> 2;          // 16-ci fragment of this block's 64-channel half, (16 NJ)-co slice
    // block -> (split, kd, ci half, co tile); an XCD walks a contiguous range of splits, the six blocks of a split back to back
    const uint32_t nct = a.ntiles / a.nkt;
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const uint32_t per_split = 6u * nct;
    const uint32_t nsplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
    const uint32_t spx = (nsplit + 7u) >> 3;
    const uint32_t sl = seq / per_split, rem = seq - sl * per_split;
    const uint32_t split = xcd * spx + sl;
    if (sl >= spx || split >= nsplit) return;
    const uint32_t tdc = rem % 6u, ct = rem / 6u;
    const uint32_t td = tdc >> 1, cih = tdc & 1u;
    const uint32_t step0 = split * a.steps_per_split;
    uint32_t step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;

    float4_t acc[9][NJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)a.gout, 0, (int)a.g_bytes, 0x00020000);
    const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
    // piece q = wave + 8 i: q < 23 activation halo (8 rows x 128 B: lane -> row l>>3, 16-byte slot l&7), else gradient (4 rows x 256 B:
    // row l>>4, slot l&15).  The DMA writes lane-linearly, so each lane fetches the SOURCE vector its slot holds after the swizzle.
    int32_t p_h[PPW], p_w[PPW];
    uint32_t p_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const uint32_t q = wave + (uint32_t)NW * i;
        if (q < (uint32_t)XP) {
            const uint32_t r = q * 8u + (lane >> 3), pos = lane & 7u;
            const uint32_t vec = ((((pos >> 1) ^ ((r >> 1) & 3u)) << 1) | (pos & 1u));
            const uint32_t hh = r / 18u, ww = r - hh * 18u;
            p_h[i] = (int32_t)hh + g.in_off[1];
            p_w[i] = (int32_t)ww + g.in_off[2];
            p_off[i] = (uint32_t)((p_h[i] * g.Wi + p_w[i]) * (g.Cin * 2)) + cih * 128u + vec * 16u;
            if (r >= 180u) p_h[i] = -(1 << 20);   // rows 180..183 of the last piece: never valid
        } else {
            const uint32_t r = (q - (uint32_t)XP) * 4u + (lane >> 4);
            const uint32_t vec = ((((lane & 15u) >> 1) ^ (r & 7u)) << 1) | (lane & 1u);
            p_h[i] = (int32_t)(r >> 4);
            p_w[i] = (int32_t)(r & 15u);
            p_off[i] = (uint32_t)((p_h[i] * g.Wo + p_w[i]) * (g.Cout * 2)) + ct * 256u + vec * 16u;
        }
    }
    auto issue = [&](uint32_t step, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t ps = smem_addr + buf * STAGE;
        uint32_t q1 = fdiv(step, a.dWP);
        const uint32_t wp = step - q1 * a.WP;
        uint32_t q2 = fdiv(q1, a.dHQ);
        const uint32_t hq = q1 - q2 * a.HQ;
        const uint32_t n = fdiv(q2, a.dD);
        const uint32_t d = q2 - n * (uint32_t)g.Dm;
        const int32_t h0 = (int32_t)hq * 8, w0 = (int32_t)wp * 16;
        const int32_t id = (int32_t)d + g.in_off[0] + (int32_t)td;
        const bool dok = (uint32_t)id < (uint32_t)g.Di;
        const uint32_t xbase = (uint32_t)((((int32_t)n * g.Di + id) * g.Hi + h0) * g.Wi + w0) * (uint32_t)(g.Cin * 2);
        const uint32_t gbase = (uint32_t)((((int32_t)n * g.Do + (int32_t)d) * g.Ho + h0) * g.Wo + w0) * (uint32_t)(g.Cout * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint32_t q = wave + (uint32_t)NW * i;
            if (q >= (uint32_t)NP) break;
            const int32_t hh = h0 + p_h[i], ww = w0 + p_w[i];
            if (q < (uint32_t)XP) {
                const bool ok = dok && (uint32_t)hh < (uint32_t)g.Hi && (uint32_t)ww < (uint32_t)g.Wi;
                dma16_hidden(rX, ps + q * 1024, ok ? xbase + p_off[i] : 0xfffffff0u);
            } else {
                const bool ok = (uint32_t)hh < (uint32_t)g.Ho && (uint32_t)ww < (uint32_t)g.Wo;
                dma16_hidden(rG, ps + XT + (q - (uint32_t)XP) * 1024, ok ? gbase + p_off[i] : 0xfffffff0u);
            }
        }
    };

    issue(step0, 0);
    dma_wait();
    __syncthreads();
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    const uint32_t trow = fq * 4u + (frow >> 2), tcol = (frow & 3u) * 4u;
    const bool do_db = a.db != nullptr && tdc == 0u;   // one of the six blocks of a split also sums the gradient rows
    float bs0 = 0.f, bs1 = 0.f;
    // Halo addresses of the 9 x 2 transposing reads of K-step 0 (voxel (ph, pw) of tap (kh, kw) reads halo row (ph + kh) * 18 + pw + kw;
    // the low / high half of an operand are patch rows 0 / 1).  K-step ks starts 36 rows further: + 4608 bytes, and the row-dependent
    // chunk swizzle ((m >> 1) & 3) advances by 18 ks = 2 ks mod 4, i.e. byte bit 6 flips on odd ks -- no per-step address arithmetic.
    uint32_t xa[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t xr0 = (uint32_t)(t / 3) * 18u + (uint32_t)(t % 3) + trow;
        xa[t][0] = roff128(xr0, fi * 16 + tcol);
        xa[t][1] = roff128(xr0 + 18u, fi * 16 + tcol);
    }
    uint32_t ga[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ga[j] = roff(trow, wn * (16 * NJ) + j * 16 + tcol);   // (+ 4096 for the high half: 16 rows keep m & 7)
    for (uint32_t st = step0; st < step1; ++st) {
        const uint32_t buf = (st - step0) & 1u;
        if (st + 1 < step1) issue(st + 1, buf ^ 1u);
        const unsigned char* px = smem + buf * STAGE;
        const unsigned char* pg = px + XT;
        if (do_db) tile_colsum<128, NW>(pg, tid, bs0, bs1);
#pragma unroll 1
        for (uint32_t ks = 0; ks < 4; ++ks) {   // 32 voxels = patch rows 2 ks, 2 ks + 1  (rolled: unrolling spills the 144 accumulators)
            short8_t gf[NJ];
            const unsigned char* pgk = pg + ks * 8192u;
            const unsigned char* pxk = px + ks * 4608u;
            const uint32_t kx = (ks & 1u) * 64u;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const v4s_t lo = lds_tr16(pgk + ga[j]);
                const v4s_t hi = lds_tr16(pgk + ga[j] + 4096);
                gf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const v4s_t lo = lds_tr16(pxk + (xa[t][0] ^ kx));
                const v4s_t hi = lds_tr16(pxk + (xa[t][1] ^ kx));
                const short8_t xf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf[j], acc[t][j], 0, 0, 0);
            }
        }
        dma_wait();       // the next step's tiles have landed (the compiler does not know about them: dma16_hidden)
        __syncthreads();  // ... in every wave, and this buffer is free
    }
    if (do_db) colsum_finish<NW>((float*)smem, tid, bs0, bs1, a.db, ct * 128u, (uint32_t)g.cout_valid);
    // partial tiles of the nine taps of this kd -> workspace [split][tile = tap + 27*ct][co 128][ci 128]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* wt = a.ws + ((size_t)split * a.ntiles + (td * 9u + t) + a.nkt * ct) * (128 * 128);
#pragma unroll
        for (int j = 0; j < NJ; ++j) *(float4_t*)(wt + (wn * (16 * NJ) + j * 16 + frow) * 128 + cih * 64 + fi * 16 + fq * 4) = acc[t][j];
    }
#endif
}

// dw[co][ci][tap] += sum_split ws[split][tile][co][kidx]   (no atomics).  One thread owns 4 consecutive kidx of one (tile, co) and
// streams the splits with 8 independent 16-byte loads in flight.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, uint32_t splits) {
    const uint32_t e4 = blockIdx.x * blockDim.x + threadIdx.x;  // (tile*16384 + co_l*128 + k_l) / 4
    if (e4 >= a.ntiles * 4096u) return;
    const uint32_t e = e4 * 4u;
    const uint32_t tile = e >> 14, co_l = (e >> 7) & 127u, k_l = e & 127u;
    const uint32_t kt = tile % a.nkt, ct = tile / a.nkt;
    const uint32_t kidx = kt * 128u + k_l, co = ct * 128u + co_l;
    if (kidx >= a.ktot || co >= (uint32_t)a.g.cout_valid) return;
    float4_t s = (float4_t){0.f, 0.f, 0.f, 0.f};
    const float* p = a.ws + e;
    const size_t stride = (size_t)a.ntiles * 16384u;
    uint32_t sp = 0;
    for (; sp + 8 <= splits; sp += 8) {
        float4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const float4_t*)(p + (size_t)(sp + u) * stride));
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < splits; ++sp) s += *(const float4_t*)(p + (size_t)sp * stride);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t kk = kidx + r;
        if (kk >= a.ktot) break;
        const uint32_t tap = fdiv(kk, a.dCin);
        const uint32_t c = kk - tap * a.g.Cin;
        if (c >= (uint32_t)a.g.cin_valid) continue;
        a.dw[co * a.s_row + (int64_t)c * a.s_red + a.lut[tap]] += s[r];
    }
}

// The same reduction for FEW tiles and MANY splits (the fused 1x1x1 backward: one 128 x 128 tile, 1 024 partial tiles): with one thread per four
// elements only 16 blocks walk 1 024 splits each (59 us, 1.1 ms per VQ-VAE step).  Here a block = 64 element groups (one contiguous KiB per split) x 16
// waves that each take every 16th split; wave 0 adds the 16 partial sums in a fixed order (still no atomics, still deterministic).
__global__ __launch_bounds__(1024) void wgrad_reduce_wide_kernel(const WgradArgs a, uint32_t splits) {
    __shared__ float4_t part[16][64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t e4 = blockIdx.x * 64u + lane;
    const uint32_t e = e4 * 4u;
    const uint32_t tile = e >> 14, co_l = (e >> 7) & 127u, k_l = e & 127u;
    const uint32_t kt = tile % a.nkt, ct = tile / a.nkt;
    const uint32_t kidx = kt * 128u + k_l, co = ct * 128u + co_l;
    const bool live = e4 < a.ntiles * 4096u && kidx < a.ktot && co < (uint32_t)a.g.cout_valid;
    float4_t s = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* p = a.ws + e;
        const size_t stride = (size_t)a.ntiles * 16384u;
        uint32_t sp = wv;
        for (; sp + 7 * 16 < splits; sp += 8 * 16) {
            float4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const float4_t*)(p + (size_t)(sp + 16 * u) * stride));
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < splits; sp += 16) s += *(const float4_t*)(p + (size_t)sp * stride);
    }
    part[wv][lane] = s;
    __syncthreads();
    if (wv != 0 || !live) return;
#pragma unroll
    for (int q = 1; q < 16; ++q) s += part[q][lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t kk = kidx + r;
        if (kk >= a.ktot) break;
        const uint32_t tap = fdiv(kk, a.dCin);
        const uint32_t c = kk - tap * a.g.Cin;
        if (c >= (uint32_t)a.g.cin_valid) continue;
        a.dw[co * a.s_row + (int64_t)c * a.s_red + a.lut[tap]] += s[r];
    }
}

// db[c] += sum over the rows m of a launch geometry of g[o(m)][c]  (fallback of the fused bias gradient for geometries whose rows are a
// strided subset of the output voxels: the transposed convolution's parity classes).  One thread per (row lane, channel).
template <typename T>
__global__ __launch_bounds__(256) void colsum_geom_kernel(const WgradArgs a, int64_t rows_per_block) {
    const T* gp = (const T*)a.gout;
    const int C = a.g.cout_valid, cs = a.g.Cout;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > (int64_t)a.M) r1 = a.M;
    for (int c = threadIdx.x & 63; c < C; c += 64) {
        float s = 0.f;
        for (int64_t m = r0 + (threadIdx.x >> 6); m < r1; m += 4) {
            const RowPos r = decode_row((uint32_t)m, a);
            s += load_as_f32(gp, sizeof(T) == 4 ? SA_F32 : SA_BF16, r.ovox * cs + c);
        }
        unsafeAtomicAdd(a.db + c, s);
    }
}

// db[c] += sum_m g[m][c].  Block = 16 channel-vectors (16 bytes each) x 16 row lanes; 16-byte coalesced loads.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ gp, int64_t M, int C, int cstride, float* __restrict__ db,
                                                     int64_t rows_per_block) {
    constexpr int VEC = DT<T>::VEC;
    const int cv = blockIdx.y * 16 + (threadIdx.x & 15);   // channel vector
    const int rl = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.f;
    if (cv * VEC < cstride) {
        for (int64_t r = r0 + rl; r < r1; r += 16) {
            const u32x4 v = *(const u32x4*)(gp + r * cstride + cv * VEC);
            if constexpr (VEC == 4) {
                s[0] += __uint_as_float(v.x); s[1] += __uint_as_float(v.y); s[2] += __uint_as_float(v.z); s[3] += __uint_as_float(v.w);
            } else {
                s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
                s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
                s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
                s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
            }
        }
    }
    __shared__ float red[16][16 * VEC + 1];
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[rl][(threadIdx.x & 15) * VEC + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 16 * VEC) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        const int c = blockIdx.y * 16 * VEC + threadIdx.x;
        if (c < C) unsafeAtomicAdd(db + c, t);
    }
}

}  // namespace sa

namespace sa {
// Shared launch plan: how the voxel range is split.  ~4096 blocks: short voxel ranges keep the (tap, co) tiles that share a
// range in lockstep inside one XCD L2 (measured 424 -> 556 TFLOP/s from 1024 -> 4096 blocks on the 3x3x3 C=128 layer) and keeps the chip
// full while bounding the partial-tile traffic.
static int plan_wgrad(const sa_conv_geom* g, int dtype, WgradArgs& a, uint32_t& splits) {
    if (dtype != SA_F32 && dtype != SA_BF16) return SA_EUNSUPPORTED;
    const int vec = dtype == SA_F32 ? 4 : 8;
    const int ntaps = g->KT[0] * g->KT[1] * g->KT[2];
    if (g->Cin % vec || g->Cout % vec || ntaps < 1 || ntaps > SA_MAX_TAPS) return SA_EINVAL;
    const int64_t M = (int64_t)g->N * g->Dm * g->Hm * g->Wm;
    if (M <= 0 || M >= (1ll << 31)) return SA_EINVAL;
    a.g = *g;
    a.dW = make_fastdiv(g->Wm);
    a.dH = make_fastdiv(g->Hm);
    a.dD = make_fastdiv(g->Dm);
    a.dTw = make_fastdiv(g->KT[2]);
    a.dThw = make_fastdiv(g->KT[1] * g->KT[2]);
    a.dCin = make_fastdiv(g->Cin);
    a.M = (uint32_t)M;
    a.ntaps = ntaps;
    a.ktot = ntaps * g->Cin;
    const int mk = dtype == SA_F32 ? 16 : 64;
    a.nchunks = (uint32_t)((M + mk - 1) / mk);
    a.nkt = (a.ktot + 127) / 128;
    const uint32_t nct = ((uint32_t)g->cout_valid + 127) / 128;
    a.ntiles = a.nkt * nct;
    // split so that one block streams ~10k voxels (SA_WGRAD_ROWS): short ranges keep the (tap, co) tiles sharing a range in lockstep
    // inside one XCD L2; at least ~1024 blocks to fill the chip, at most 2 GiB of partial tiles
    const int target_rows = g_tunables.wgrad_rows;
    const uint32_t want_cps = (uint32_t)(target_rows / mk) ? (uint32_t)(target_rows / mk) : 1u;
    splits = (a.nchunks + want_cps - 1) / want_cps;
    // dense layers (one tap: the Performer's M = 8 400 rows) do better with ~512 blocks of twice the range -- half the partial tiles to write
    // and reduce (measured 213 -> 191 us per layer over its four weight gradients); the convolutions measured better at 1024
    const uint32_t env_blocks = (uint32_t)g_tunables.wgrad_min_blocks;
    const uint32_t min_blocks = env_blocks ? env_blocks : ((ntaps == 1 && a.nchunks <= 1024u) ? 512u : 1024u);   // (long 1x1x1 reductions: 1024 again)
    const uint32_t min_splits = (min_blocks + a.ntiles - 1) / a.ntiles;
    if (splits < min_splits) splits = min_splits;
    const uint32_t cap = (uint32_t)((2ull << 30) / ((uint64_t)a.ntiles * 65536ull));
    if (splits > cap && cap >= 1) splits = cap;
    const uint32_t min_chunks = dtype == SA_F32 ? 16 : 4;
    uint32_t max_splits = (a.nchunks + min_chunks - 1) / min_chunks;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.chunks_per_split = (a.nchunks + splits - 1) / splits;
    splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    // halo kernel: 3x3x3 / stride 1 / same, bf16, 128 input channels, both operands addressable with 32-bit offsets
    a.halo = 0;
    {
        bool ok = dtype == SA_BF16 && g->Cin == 128 && g->cin_valid == 128 && g->Cout % 128 == 0 && !dbg(SA_DBG_NO_HALO) && !dbg(SA_DBG_NO_DMA);
        for (int d = 0; d < 3 && ok; ++d)
            ok = g->KT[d] == 3 && g->in_mult[d] == 1 && g->tap_step[d] == 1 && g->in_off[d] == -1 && g->out_mult[d] == 1 && g->out_off[d] == 0;
        ok = ok && g->Dm == g->Do && g->Hm == g->Ho && g->Wm == g->Wo && g->Di == g->Do && g->Hi == g->Ho && g->Wi == g->Wo;
        const uint64_t ib = (uint64_t)g->N * g->Di * g->Hi * g->Wi * g->Cin * 2, gb = (uint64_t)g->N * g->Do * g->Ho * g->Wo * g->Cout * 2;
        ok = ok && ib < 0xffffff00ull - (1u << 20) && gb < 0xffffff00ull - (1u << 20);
        if (ok) {
            // nine-tap kernel (steps of 8 x 16 voxels) by default; SA_WGRAD_HALO9=0 -> the three-tap kernel (steps of 4 x 16)
            const bool nine = !dbg(SA_DBG_NO_WGRAD_HALO9);
            const uint32_t phs = nine ? 8u : 4u;
            const uint32_t hq = (uint32_t)(g->Ho + phs - 1) / phs, wp = (uint32_t)(g->Wo + 15) / 16;
            const double eff = (double)g->Ho * g->Wo / ((double)hq * phs * wp * 16);
            const uint64_t nsteps = (uint64_t)g->N * g->Dm * hq * wp;
            if (eff >= 0.8 && nsteps >= 512) {
                a.halo = nine ? 9u : 4u;
                a.HQ = hq;
                a.WP = wp;
                a.nsteps = (uint32_t)nsteps;
                a.dWP = make_fastdiv(wp);
                a.dHQ = make_fastdiv(hq);
                const int want = g_tunables.wgrad_halo_splits ? g_tunables.wgrad_halo_splits : (nine ? 128 : 256);
                uint32_t sp = (uint32_t)want;
                if (sp > a.nsteps / 16) sp = a.nsteps / 16;   // at least 16 steps per block
                if (sp < 1) sp = 1;
                a.steps_per_split = (a.nsteps + sp - 1) / sp;
                splits = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
            }
        }
    }
    return 0;
}
}  // namespace sa

extern "C" int64_t sa_conv_wgrad_workspace_bytes(const sa_conv_geom* g, int dtype) {
    using namespace sa;
    if (!g) return SA_EINVAL;
    WgradArgs a;
    uint32_t splits;
    const int rc = plan_wgrad(g, dtype, a, splits);
    if (rc) return rc;
    return (int64_t)splits * a.ntiles * 128 * 128 * 4;
}

extern "C" int sa_colsum(const void* gp, int dtype, int64_t M, int C, int cstride, float* db, void* stream);

static int conv_wgrad_impl(const sa_conv_geom* g, int dtype, const void* in, const void* gout, float* dw, float* db, const int32_t* tap_lut_host,
                           int64_t s_row, int64_t s_red, void* workspace, int64_t workspace_bytes, const void* dg_wpk, void* dg_out, void* stream) {
    using namespace sa;
    if (!g || !in || !gout || !dw) return SA_EINVAL;
    WgradArgs a;
    uint32_t splits;
    const int rc = plan_wgrad(g, dtype, a, splits);
    if (rc) return rc;
    a.in = in;
    a.gout = gout;
    a.dw = dw;
    const int ntaps = a.ntaps;
    for (int t = 0; t < SA_MAX_TAPS; ++t) a.lut[t] = tap_lut_host ? (t < ntaps ? tap_lut_host[t] : 0) : t;
    a.s_row = s_row;
    a.s_red = s_red;
    const int64_t need = (int64_t)splits * a.ntiles * 128 * 128 * 4;
    a.ws = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
    if (workspace && !a.ws) return SA_EINVAL;  // a workspace was given but is too small
    a.db = nullptr;
    const size_t lds = dtype == SA_F32 ? 4 * 8192 : 4 * 16384;
    const uint32_t splits8 = (splits + 7u) & ~7u;  // blocks of the padded splits exit immediately
    dim3 grid(a.ntiles * splits8);
    hipStream_t st = (hipStream_t)stream;
    {
        const int sz = dtype == SA_F32 ? 4 : 2;
        const uint64_t ib = (uint64_t)g->N * g->Di * g->Hi * g->Wi * g->Cin * sz, gb = (uint64_t)g->N * g->Do * g->Ho * g->Wo * g->Cout * sz;
        const bool fits = ib < 0xffffff00ull && gb < 0xffffff00ull && !dbg(SA_DBG_NO_DMA);
        a.in_bytes = fits ? (uint32_t)ib : 0u;
        a.g_bytes = fits ? (uint32_t)gb : 0u;
    }
    const bool fuse_db = db && dtype == SA_BF16 && !dbg(SA_DBG_NO_FUSED_DB);   // the bf16 LDS-DMA kernels sum the gradient tile they stage
    a.dg_wpk = dg_wpk;
    a.dg_out = dg_out;
    if (dg_out) {  // fused 1x1x1 data gradient: only the bf16 LDS-DMA kernel, one (tap, co) tile, rows = voxels
        if (!dg_wpk || dtype != SA_BF16 || !a.in_bytes || a.ntiles != 1 || a.halo || g->Cin != 128 || g->Cout != 128 || g->cin_valid != 128 || g->cout_valid != 128)
            return SA_EUNSUPPORTED;
        if (fuse_db) a.db = db;
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_dma_kernel<unsigned short, true, 4>"), note_kernel(g_last_conv_kernel));
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<bf16_t, true>), grid, dim3(256), lds, st, a);
    } else if (a.halo && a.ws) {
        if (fuse_db) a.db = db;
        static std::atomic<uint64_t> attr_done{0};   // one bit per device
        configure_once_per_device(attr_done, [] {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_halo_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_halo9_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 55 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_halo9_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 55 * 1024);
        });
        const uint32_t nct = a.ntiles / a.nkt, spx = (splits + 7u) / 8u;
        if (a.halo == 9) {
            // (a sixteen-wave form of this kernel -- NW = 16, 128 VGPRs, four waves per SIMD -- measured 4.74 ms against 4.28 ms: kept as a template
            //  instance for A/B runs, SA_PP_DBG bit 14)
            if (g_tunables.pp_dbg & 16384u) {
                (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_halo9_kernel<16>"), note_kernel(g_last_conv_kernel));
                hipLaunchKernelGGL(conv_wgrad_halo9_kernel<16>, dim3(8u * spx * 6u * nct), dim3(1024), 2 * 55 * 1024, st, a);
            } else {
                (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_halo9_kernel<8>"), note_kernel(g_last_conv_kernel));
                hipLaunchKernelGGL(conv_wgrad_halo9_kernel<8>, dim3(8u * spx * 6u * nct), dim3(512), 2 * 55 * 1024, st, a);
            }
        } else {
            (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_halo_kernel<4>"), note_kernel(g_last_conv_kernel));
            hipLaunchKernelGGL(conv_wgrad_halo_kernel<4>, dim3(8u * spx * 9u * nct), dim3(768), 2 * 34 * 1024, st, a);
        }
    } else if (a.in_bytes) {
        if (fuse_db) a.db = db;
        // bf16: eight waves per block (k4s2 / transposed-conv weight gradients 2.66 -> 2.29 / 2.37 ms at batch 8); SA_DBG_HALO256_4W keeps four
        const bool w8 = dtype == SA_BF16 && !dbg(SA_DBG_HALO256_4W);
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_dma_kernel<%s, false, %d>", dtype == SA_F32 ? "float" : "unsigned short", w8 ? 8 : 4), note_kernel(g_last_conv_kernel));
        if (dtype == SA_F32) hipLaunchKernelGGL(conv_wgrad_dma_kernel<float>, grid, dim3(256), lds, st, a);
        else if (w8) hipLaunchKernelGGL((conv_wgrad_dma_kernel<bf16_t, false, 8>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad_dma_kernel<bf16_t>, grid, dim3(256), lds, st, a);
    } else {
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_wgrad_kernel<%s>", dtype == SA_F32 ? "float" : "unsigned short"), note_kernel(g_last_conv_kernel));
        if (dtype == SA_F32) hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, st, a);
    }
    SA_CHECK_LAUNCH();
    if (a.ws) {
        if (a.ntiles <= 8u && splits >= 128u) hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(a.ntiles * 64u), dim3(1024), 0, st, a, splits);
        else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((a.ntiles * 4096u + 255) / 256), dim3(256), 0, st, a, splits);
        SA_CHECK_LAUNCH();
    }
    if (db && !a.db) {  // not fused (fp32, or operands beyond 32-bit offsets): stand-alone column sums over THIS geometry's rows
        bool dense = g->Dm == g->Do && g->Hm == g->Ho && g->Wm == g->Wo;
        for (int d = 0; d < 3; ++d) dense = dense && g->out_mult[d] == 1 && g->out_off[d] == 0;
        if (dense) return sa_colsum(gout, dtype, (int64_t)g->N * g->Do * g->Ho * g->Wo, g->cout_valid, g->Cout, db, stream);
        a.db = db;
        const int64_t rpb = ((int64_t)a.M + 1023) / 1024 < 64 ? 64 : ((int64_t)a.M + 1023) / 1024;
        const unsigned nb = (unsigned)(((int64_t)a.M + rpb - 1) / rpb);
        if (dtype == SA_F32) hipLaunchKernelGGL(colsum_geom_kernel<float>, dim3(nb), dim3(256), 0, st, a, rpb);
        else hipLaunchKernelGGL(colsum_geom_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, a, rpb);
        SA_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sa_colsum(const void* gp, int dtype, int64_t M, int C, int cstride, float* db, void* stream) {
    using namespace sa;
    if (!gp || !db || M <= 0 || C <= 0) return SA_EINVAL;
    const int vec = dtype == SA_F32 ? 4 : 8;
    if (cstride % vec) return SA_EINVAL;
    int64_t rows_per_block = (M + 511) / 512;
    if (rows_per_block < 64) rows_per_block = 64;
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), (cstride / vec + 15) / 16);
    if (dtype == SA_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)gp, M, C, cstride, db, rows_per_block);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gp, M, C, cstride, db, rows_per_block);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_conv_wgrad(const sa_conv_geom* g, int dtype, const void* in, const void* gout, float* dw, float* db, const int32_t* tap_lut_host,
                             int64_t s_row, int64_t s_red, void* workspace, int64_t workspace_bytes, void* stream) {
    return conv_wgrad_impl(g, dtype, in, gout, dw, db, tap_lut_host, s_row, s_red, workspace, workspace_bytes, nullptr, nullptr, stream);
}

// Backward of a 1x1x1, 128 -> 128 channel convolution whose INPUT is a post-ReLU tensor (the second convolution of the residual block,
// reference baseline.py:150-160), in one launch: weight gradient + bias gradient as sa_conv_wgrad, and the data gradient
//   dx[m][ci] = (in[m][ci] > 0) * sum_co gout[m][co] W[co][ci]
// from the same staged tiles (the two stand-alone kernels read gout twice and `in` twice).  dgrad_wpk = sa_pack_weights operand of the
// layer's data-gradient plan ([128 ci][128 co]).  bf16 only; SA_EUNSUPPORTED otherwise (callers fall back to the two launches).
extern "C" int sa_conv1x1_backward(const sa_conv_geom* g, int dtype, const void* in, const void* gout, float* dw, float* db, int64_t s_row, int64_t s_red,
                                   void* workspace, int64_t workspace_bytes, const void* dgrad_wpk, void* dx, void* stream) {
    if (!g || !dgrad_wpk || !dx) return SA_EINVAL;
    if (g->KT[0] * g->KT[1] * g->KT[2] != 1) return SA_EUNSUPPORTED;
    for (int d = 0; d < 3; ++d)
        if (g->in_mult[d] != 1 || g->in_off[d] != 0 || g->out_mult[d] != 1 || g->out_off[d] != 0) return SA_EUNSUPPORTED;
    if (g->Dm != g->Do || g->Hm != g->Ho || g->Wm != g->Wo || g->Di != g->Do || g->Hi != g->Ho || g->Wi != g->Wo) return SA_EUNSUPPORTED;
    return conv_wgrad_impl(g, dtype, in, gout, dw, db, nullptr, s_row, s_red, workspace, workspace_bytes, dgrad_wpk, dx, stream);
}
