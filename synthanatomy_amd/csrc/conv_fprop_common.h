// Shared by the implicit-GEMM forward / data-gradient translation units (conv_fprop.hip, conv_fprop_f16.hip): the launch arguments, the MFMA slab product,
// the swizzled tile addressing and the epilogues (LDS-staged and register forms).  Everything here is a template or __device__ __forceinline__.
#pragma once
#include <stdlib.h>

#include <stdio.h>

#include "sa_common.h"

namespace sa {

struct FpropArgs {
    const void* in;
    const void* wpk;
    void* out;
    sa_epilogue ep;
    sa_conv_geom g;
    FastDiv dW, dH, dD;   // decode m -> (n, dm, hm, wm)
    FastDiv dTw, dThw;    // decode tap -> (td, th, tw)
    FastDiv dCv;          // k-vector -> (tap, channel-vector): divisor Cin/VEC
    uint32_t M;
    uint32_t ntaps;
    uint32_t nk;          // K-slabs
    uint32_t nblk_m;
    FastDiv dCin;         // element k -> (tap, channel)
    uint32_t in_bytes, w_bytes;
    // fused residual block (bf16, 128 channels): after the main loop  h = relu(acc + bias1)  is kept on chip, optionally stored to
    // `h_out`, and multiplied by the 1x1x1 weights `w2pk` [128][128]; the regular epilogue (bias2 = ep.bias, addend, act) then runs on
    // that second product.
    const void* w2pk;
    const float* bias1;
    void* h_out;
    uint32_t HP, WP;      // halo mainloop: patches per plane along H (8 voxels) and W (16 voxels)
    uint32_t DP;          // cell mainloop: tiles along D (2 planes), HP / WP = tiles of 8 x 8
    uint32_t group_m;     // im2col-order DMA mainloop, dense (1x1x1) layers with several channel tiles: blocks are ordered in groups of `group_m` row tiles x
                          // all channel tiles (0: row tiles fastest, the order in which convolution tiles share their halos)
    // Several launch geometries that differ ONLY in in_off / out_off and their packed weights -- the eight output-parity classes of ConvTranspose3d k4 s2 (and
    // of the strided convolution's data gradient) -- in ONE launch: the grid is ncls x the blocks of one class, class-major; a block takes its class's
    // offsets and operand.  (sa_conv_fprop_classes; ncls <= 1: a.g / a.wpk as they are.)
    uint32_t ncls;
    int32_t cls_in_off[8][3], cls_out_off[8][3];
    const void* cls_wpk[8];
    uint32_t dbg;         // dev only (env SA_PP_DBG): 256 = LDS-staged epilogue instead of the register one; with -DSA_PP_DEBUG_VARIANTS also the
                          // ablation bits (halo: 1 skip halo DMA, 2 skip weight DMA, 64 skip epilogue; im2col-order: 64 / 128 skip activation / weight DMA)
};

template <typename T>
__device__ __forceinline__ void mma_slab(float4_t& acc, const u32x4& wa, const u32x4& xb);

template <>
__device__ __forceinline__ void mma_slab<bf16_t>(float4_t& acc, const u32x4& wa, const u32x4& xb) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const short8_t*)&wa, *(const short8_t*)&xb, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slab<f16_t>(float4_t& acc, const u32x4& wa, const u32x4& xb) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const half8_t*)&wa, *(const half8_t*)&xb, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slab<float>(float4_t& acc, const u32x4& wa, const u32x4& xb) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), __uint_as_float(xb.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), __uint_as_float(xb.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), __uint_as_float(xb.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), __uint_as_float(xb.w), acc, 0, 0, 0);
}

// byte offset of 16-byte vector `vec` (0..7) of row `row` inside a [rows][128 B] swizzled tile
__device__ __forceinline__ uint32_t tile_off(uint32_t row, uint32_t vec) { return row * 128u + ((vec ^ (row & 7u)) << 4); }

// block -> (class, block id inside the class, after the XCD-aware remap); patches the class's offsets / operand into the block's copy of the arguments
// (scalar values: forced into SGPRs -- the kernels that take classes sit at their VGPR line)
__device__ __forceinline__ uint32_t select_class(FpropArgs& a, const FpropArgs& karg) {
    // (the class tables are indexed in the KERNEL ARGUMENT -- scalar loads --, never in the block's copy: a dynamically indexed local copy lives in scratch)
    if (karg.ncls <= 1u) return xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t per = gridDim.x / karg.ncls, cls = blockIdx.x / per;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        a.g.in_off[d] = karg.cls_in_off[cls][d];
        a.g.out_off[d] = karg.cls_out_off[cls][d];
    }
    a.wpk = karg.cls_wpk[cls];
    return xcd_remap(blockIdx.x - cls * per, per);
}

// ---- epilogue shared by both mainloops, staged through LDS so that HBM sees full channel rows
// output voxel (linear index into [N, Do, Ho, Wo]) of GEMM row m of the launch grid, or -1 beyond M
__device__ __forceinline__ long long linear_row_voxel(const FpropArgs& a, uint32_t m) {
    const sa_conv_geom& g = a.g;
    if (m >= a.M) return -1;
    uint32_t q = fdiv(m, a.dW);
    const uint32_t wmx = m - q * g.Wm;
    uint32_t q2 = fdiv(q, a.dH);
    const uint32_t hmx = q - q2 * g.Hm;
    const uint32_t n = fdiv(q2, a.dD);
    const uint32_t dmx = q2 - n * g.Dm;
    return (((long long)n * g.Do + (dmx * g.out_mult[0] + g.out_off[0])) * g.Ho + (hmx * g.out_mult[1] + g.out_off[1])) * g.Wo +
           (wmx * g.out_mult[2] + g.out_off[2]);
}

// `row_ov(row)` -> output voxel of tile row `row` (or -1): the linear launch grid, or the 2-D patch of the halo mainloop
// `parks`: whether this thread holds accumulators of the tile (false for the second K group of a KG = 2 block, which only helps with the write-back)
template <int BM, int BN, int WM, int WN, int MI, int NI, int NT, typename RowOv>
__device__ __forceinline__ void fprop_epilogue_ov(const FpropArgs& a, float4_t (&acc)[NI][MI], unsigned char* smem, uint32_t tid, uint32_t wm, uint32_t wn,
                                                  uint32_t frow, uint32_t fq, uint32_t n_base, RowOv row_ov, bool parks = true) {
    const sa_conv_geom& g = a.g;
    //  A) every lane parks its 4x(acc + bias) for one voxel in an fp32 tile [BM][BN+4] (stride padded: conflict-free b128)
    //  B) the block re-reads the tile voxel-row-wise, 4 channels per thread: addend / activation / mask are applied with
    //     8- or 16-byte coalesced loads and the result leaves as 8-byte (bf16) or 16-byte (fp32) coalesced stores.
    constexpr int LDT = BN + 4;
    float* sT = (float*)smem;
    long long* sOv = (long long*)(smem + BM * LDT * 4);
    const sa_epilogue& ep = a.ep;
    if (parks) {
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const uint32_t row = wm * (MI * 16) + j * 16 + frow;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t col = wn * (NI * 16) + i * 16 + fq * 4;
                float4_t v = acc[i][j];
                if (ep.bias) {
                    const float4_t bv = *(const float4_t*)(ep.bias + n_base + col);
                    v += bv;
                }
                *(float4_t*)(sT + row * LDT + col) = v;
            }
        }
    }
    if (tid < BM) sOv[tid] = row_ov(tid);
    __syncthreads();
    const float alpha = ep.alpha ? *ep.alpha : 1.f;
    const bool vec_ok = (g.Cout & 3) == 0;
    constexpr int NG = BN / 4;           // 4-channel groups per voxel row
    constexpr int RPP = NT / NG;         // rows per pass
    const uint32_t grp = tid % NG, r0 = tid / NG;
    const uint32_t co0 = n_base + grp * 4;
    // Whole tiles of valid channels with 16-byte aligned rows (every dense layer, every 128-channel convolution): phase B in BATCHES of U rows per thread --
    // the U LDS reads, then the U addend / mask loads, then the arithmetic as passes over the batch with each run-time switch (activation, mask mode, operand
    // types) taken ONCE per batch, then the stores.  Same operations per element, in the same order, as the row-at-a-time loop below (bit-identical results);
    // that loop executed ~100 instructions and a dozen scalar branches per 16-byte store, and a store-pattern probe (tools/probes/store_pattern.hip) writes the
    // same tiles 2x faster than the epilogue it models: 35 of the 65 us of the q|k|v projection (103 MB of fp32 output) were this loop.
    if (vec_ok && n_base + BN <= (uint32_t)g.cout_valid) {
        constexpr int ITERS = BM / RPP, U = ITERS % 4 == 0 ? 4 : (ITERS % 3 == 0 ? 3 : (ITERS % 2 == 0 ? 2 : 1));
        const int add_kind = !ep.addend ? 0 : (ep.add_dtype == SA_F32 ? 1 : 2);
        const int mask_kind = ep.mask_mode == SA_MASK_NONE ? 0 : (ep.mask_dtype == SA_F32 ? 1 : 2);
        for (int it0 = 0; it0 < ITERS; it0 += U) {
            long long ov[U];
            float v[U][4], ad[U][4], mk[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t row = r0 + (it0 + u) * RPP;
                ov[u] = sOv[row];
                const float4_t tv = *(const float4_t*)(sT + row * LDT + grp * 4);
                v[u][0] = tv[0]; v[u][1] = tv[1]; v[u][2] = tv[2]; v[u][3] = tv[3];
            }
            int64_t o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) o[u] = (ov[u] < 0 ? 0 : ov[u]) * g.Cout + co0;      // (rows beyond M: a valid address that is never stored to)
            if (add_kind == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4_t t4 = ov[u] >= 0 ? *(const float4_t*)((const float*)ep.addend + o[u]) : (float4_t){0.f, 0.f, 0.f, 0.f};
                    ad[u][0] = t4[0]; ad[u][1] = t4[1]; ad[u][2] = t4[2]; ad[u][3] = t4[3];
                }
            } else if (add_kind == 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint2 t2 = ov[u] >= 0 ? *(const uint2*)((const bf16_t*)ep.addend + o[u]) : make_uint2(0u, 0u);
                    unpack2_dt(ep.add_dtype, t2.x, ad[u][0], ad[u][1]);
                    unpack2_dt(ep.add_dtype, t2.y, ad[u][2], ad[u][3]);
                }
            }
            if (mask_kind == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4_t t4 = ov[u] >= 0 ? *(const float4_t*)((const float*)ep.mask + o[u]) : (float4_t){0.f, 0.f, 0.f, 0.f};
                    mk[u][0] = t4[0]; mk[u][1] = t4[1]; mk[u][2] = t4[2]; mk[u][3] = t4[3];
                }
            } else if (mask_kind == 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint2 t2 = ov[u] >= 0 ? *(const uint2*)((const bf16_t*)ep.mask + o[u]) : make_uint2(0u, 0u);
                    mk[u][0] = __uint_as_float(t2.x << 16); mk[u][1] = __uint_as_float(t2.x & 0xffff0000u);
                    mk[u][2] = __uint_as_float(t2.y << 16); mk[u][3] = __uint_as_float(t2.y & 0xffff0000u);
                }
            }
            if (ep.out_pre) {   // pre-activation copy (bf16): acc + bias
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ov[u] >= 0) {
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(v[u][0]) | ((uint32_t)f32_to_bf16(v[u][1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16(v[u][2]) | ((uint32_t)f32_to_bf16(v[u][3]) << 16);
                        *(uint2*)((bf16_t*)ep.out_pre + o[u]) = pk;
                    }
            }
            if (add_kind && ep.add_before_act) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] += ad[u][r];
            }
            if (ep.act == SA_ACT_RELU) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] = fmaxf(v[u][r], 0.f);
            } else if (ep.act == SA_ACT_LRELU) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] = v[u][r] > 0.f ? v[u][r] : v[u][r] * ep.slope;
            } else if (ep.act == SA_ACT_GELU) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] = gelu_f(v[u][r]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][r] *= alpha;
            if (add_kind && !ep.add_before_act) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] += ad[u][r];
            }
            if (ep.mask_mode == SA_MASK_POS) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] = mk[u][r] > 0.f ? v[u][r] : 0.f;
            } else if (ep.mask_mode == SA_MASK_LRELU) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] = mk[u][r] > 0.f ? v[u][r] : v[u][r] * ep.slope;
            } else if (ep.mask_mode == SA_MASK_GELU) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u][r] *= gelu_grad_f(mk[u][r]);
            }
            if (ep.out_dtype == SA_F32) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ov[u] >= 0) *(float4_t*)((float*)a.out + o[u]) = (float4_t){v[u][0], v[u][1], v[u][2], v[u][3]};
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ov[u] >= 0) {
                        uint2 pk;
                        pk.x = pack2_dt(ep.out_dtype, v[u][0], v[u][1]);
                        pk.y = pack2_dt(ep.out_dtype, v[u][2], v[u][3]);
                        *(uint2*)((bf16_t*)a.out + o[u]) = pk;
                    }
            }
            if (ep.out_lp) {        // bf16 copy of the final value
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ov[u] >= 0) {
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(v[u][0]) | ((uint32_t)f32_to_bf16(v[u][1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16(v[u][2]) | ((uint32_t)f32_to_bf16(v[u][3]) << 16);
                        *(uint2*)((bf16_t*)ep.out_lp + o[u]) = pk;
                    }
            }
        }
        return;
    }
    if (co0 < (uint32_t)g.cout_valid) {
#pragma unroll 4
        for (int it = 0; it < BM / RPP; ++it) {
            const uint32_t row = r0 + it * RPP;
            const long long ov = sOv[row];
            if (ov < 0) continue;
            const int64_t o = ov * g.Cout + co0;
            const float4_t tv = *(const float4_t*)(sT + row * LDT + grp * 4);
            float v[4] = {tv[0], tv[1], tv[2], tv[3]};
            const bool full = vec_ok && co0 + 3 < (uint32_t)g.cout_valid;
            if (ep.out_pre && full) {   // pre-activation copy (bf16): acc + bias
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                *(uint2*)((bf16_t*)ep.out_pre + o) = pk;
            }
            float ad[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (full) {
                if (ep.addend) {
                    if (ep.add_dtype == SA_F32) {
                        const float4_t t4 = *(const float4_t*)((const float*)ep.addend + o);
                        ad[0] = t4[0]; ad[1] = t4[1]; ad[2] = t4[2]; ad[3] = t4[3];
                    } else {
                        const uint2 t2 = *(const uint2*)((const bf16_t*)ep.addend + o);
                        unpack2_dt(ep.add_dtype, t2.x, ad[0], ad[1]);
                        unpack2_dt(ep.add_dtype, t2.y, ad[2], ad[3]);
                    }
                }
                if (ep.mask_mode != SA_MASK_NONE) {
                    if (ep.mask_dtype == SA_F32) {
                        const float4_t t4 = *(const float4_t*)((const float*)ep.mask + o);
                        mk[0] = t4[0]; mk[1] = t4[1]; mk[2] = t4[2]; mk[3] = t4[3];
                    } else {
                        const uint2 t2 = *(const uint2*)((const bf16_t*)ep.mask + o);
                        mk[0] = __uint_as_float(t2.x << 16); mk[1] = __uint_as_float(t2.x & 0xffff0000u);
                        mk[2] = __uint_as_float(t2.y << 16); mk[3] = __uint_as_float(t2.y & 0xffff0000u);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co0 + r >= (uint32_t)g.cout_valid) continue;
                    if (ep.addend) ad[r] = load_as_f32(ep.addend, ep.add_dtype, o + r);
                    if (ep.mask_mode != SA_MASK_NONE) mk[r] = load_as_f32(ep.mask, ep.mask_dtype, o + r);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r];
                if (ep.add_before_act) x += ad[r];
                if (ep.act == SA_ACT_RELU) x = fmaxf(x, 0.f);
                else if (ep.act == SA_ACT_LRELU) x = x > 0.f ? x : x * ep.slope;
                else if (ep.act == SA_ACT_GELU) x = gelu_f(x);
                x *= alpha;
                if (!ep.add_before_act) x += ad[r];
                if (ep.mask_mode == SA_MASK_POS) x = mk[r] > 0.f ? x : 0.f;
                else if (ep.mask_mode == SA_MASK_LRELU) x = mk[r] > 0.f ? x : x * ep.slope;
                else if (ep.mask_mode == SA_MASK_GELU) x *= gelu_grad_f(mk[r]);
                v[r] = x;
            }
            if (full) {
                if (ep.out_dtype == SA_F32) {
                    *(float4_t*)((float*)a.out + o) = (float4_t){v[0], v[1], v[2], v[3]};
                } else {
                    uint2 pk;
                    pk.x = pack2_dt(ep.out_dtype, v[0], v[1]);
                    pk.y = pack2_dt(ep.out_dtype, v[2], v[3]);
                    *(uint2*)((bf16_t*)a.out + o) = pk;
                }
                if (ep.out_lp) {        // bf16 copy of the final value
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *(uint2*)((bf16_t*)ep.out_lp + o) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < (uint32_t)g.cout_valid) store_from_f32(a.out, ep.out_dtype, o + r, v[r]);
            }
        }
    }
}

// ---- register epilogue (no LDS, no barrier): a 4 x 4 transpose across the four 16-lane quarters of the wave (v_permlane32_swap +
// v_permlane16_swap, gfx950) turns "lane = 4 channels of each of 4 column fragments" into "lane = 16 CONSECUTIVE channels" of its voxel
// row, so addend / mask come in and the result leaves as 32-byte (bf16) or 64-byte (fp32) contiguous pieces, 128 / 256 B per row.
// permlane32_swap(a, b) = {[a.q0 a.q1 b.q0 b.q1], [a.q2 a.q3 b.q2 b.q3]};  permlane16_swap(a, b) = {[a.q0 b.q0 a.q2 b.q2], [a.q1 b.q1 a.q3 b.q3]}
// (probed on MI355X).  Requires a full 128-channel tile of valid output channels and 16-byte aligned rows; the caller checks.
__device__ __forceinline__ void load16(const void* base, int dtype, int64_t off, float (&v)[16]) {
    if (dtype == SA_F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4_t t = *(const float4_t*)((const float*)base + off + 4 * k);
            v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32x4 t = *(const u32x4*)((const bf16_t*)base + off + 8 * k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * k + 2 * e] = __uint_as_float(t[e] << 16);
                v[8 * k + 2 * e + 1] = __uint_as_float(t[e] & 0xffff0000u);
            }
        }
    }
}

// The epilogue's addend / mask rows come from HBM (2-4 us under load).  Fetching them row group by row group right before use serialised
// MI round trips per block -- measured with s_memtime on the 16 x 16-patch kernels: 100 k of a block's 242 k cycles sat in the epilogue,
// eight dependent HBM round trips -- so they are fetched in batches of EPI_BATCH row groups: all loads of a batch are issued back to back
// (packed 16-byte pieces, 4 VGPRs each), then the batch is combined and stored.  The fused residual block issues its first batch BEFORE the
// second GEMM, so that round trip hides behind it.
// batch = as many row groups as fit ~64 VGPRs of packed addend / mask pieces (bf16 source: 8 VGPRs per row group, fp32: 16)
template <bool ADD, bool MASK, bool ADD32, bool MASK32, int BUDGET>
constexpr int epi_batch_size() {
    constexpr int regs = (ADD ? (ADD32 ? 16 : 8) : 0) + (MASK ? (MASK32 ? 16 : 8) : 0);
    return regs * 4 <= BUDGET ? 4 : regs * 2 <= BUDGET ? 2 : 1;
}

template <int W, bool F16 = false>
__device__ __forceinline__ void epi_unpack16(const u32x4 (&p)[W], float (&v)[16]) {
    if constexpr (W == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * k + e] = __uint_as_float(p[k][e]);
    } else if constexpr (F16) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * k + 2 * e] = f16_to_f32((unsigned short)(p[k][e] & 0xffffu));
                v[8 * k + 2 * e + 1] = f16_to_f32((unsigned short)(p[k][e] >> 16));
            }
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * k + 2 * e] = __uint_as_float(p[k][e] << 16);
                v[8 * k + 2 * e + 1] = __uint_as_float(p[k][e] & 0xffff0000u);
            }
    }
}

// one batch: row groups JB .. JB + EPI_BATCH - 1 (compile-time indices: a run-time index would demote the accumulators to scratch memory)
// F16IO (the kernels of an f16 forward chain): 16-bit addends and outputs are IEEE halves, and the launch may ask for a bf16 copy of its output
// (sa_epilogue.out_lp); compile-time, because the eight-wave bf16 kernels sit exactly at the 128-VGPR line and must not carry the extra paths
template <int MI, int NI, int JB, bool ADD, bool MASK, bool ADD32, bool MASK32, int BUDGET, bool F16IO, typename RowOv>
__device__ __forceinline__ void epi_batch(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wm, uint32_t frow, uint32_t c0, RowOv row_ov, float alpha) {
    const sa_conv_geom& g = a.g;
    const sa_epilogue& ep = a.ep;
    constexpr int AW = ADD32 ? 4 : 2, MW = MASK32 ? 4 : 2;
    constexpr int EPI_BATCH = epi_batch_size<ADD, MASK, ADD32, MASK32, BUDGET>();
    u32x4 adp[EPI_BATCH][AW], mkp[EPI_BATCH][MW];
    int64_t o[EPI_BATCH];
    bool ok[EPI_BATCH];
#pragma unroll
    for (int jj = 0; jj < EPI_BATCH; ++jj) {
        const long long ov = row_ov(wm * (MI * 16) + (JB + jj) * 16 + frow);
        ok[jj] = ov >= 0;
        o[jj] = (ok[jj] ? ov : 0ll) * g.Cout + c0;   // rows outside the volume read voxel 0 (valid memory) and are not stored
    }
#pragma unroll
    for (int jj = 0; jj < EPI_BATCH; ++jj) {
        if constexpr (ADD) {
#pragma unroll
            for (int k = 0; k < AW; ++k)
                adp[jj][k] = ADD32 ? *(const u32x4*)((const float*)ep.addend + o[jj] + 4 * k) : *(const u32x4*)((const bf16_t*)ep.addend + o[jj] + 8 * k);
        }
        if constexpr (MASK) {
#pragma unroll
            for (int k = 0; k < MW; ++k)
                mkp[jj][k] = MASK32 ? *(const u32x4*)((const float*)ep.mask + o[jj] + 4 * k) : *(const u32x4*)((const bf16_t*)ep.mask + o[jj] + 8 * k);
        }
    }
#pragma unroll
    for (int jj = 0; jj < EPI_BATCH; ++jj) {
        constexpr int j0 = JB;
        float v[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0 = acc[0][j0 + jj][r], t1 = acc[1][j0 + jj][r], t2 = acc[2][j0 + jj][r], t3 = acc[3][j0 + jj][r];
            quarter_transpose(t0, t1, t2, t3);       // t[i'] = channel fq*16 + i'*4 + r
            v[r] = t0; v[4 + r] = t1; v[8 + r] = t2; v[12 + r] = t3;
        }
        float ad[16], mk[16];
        if constexpr (ADD) epi_unpack16<AW, F16IO>(adp[jj], ad);
        if constexpr (MASK) epi_unpack16<MW>(mkp[jj], mk);      // (16-bit masks are bf16: data gradients only)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float x = v[e];
            if constexpr (ADD) { if (ep.add_before_act) x += ad[e]; }
            if (ep.act == SA_ACT_RELU) x = fmaxf(x, 0.f);
            else if (ep.act == SA_ACT_LRELU) x = x > 0.f ? x : x * ep.slope;
            else if (ep.act == SA_ACT_GELU) x = gelu_f(x);
            x *= alpha;
            if constexpr (ADD) { if (!ep.add_before_act) x += ad[e]; }
            if constexpr (MASK) {
                if (ep.mask_mode == SA_MASK_POS) x = mk[e] > 0.f ? x : 0.f;
                else if (ep.mask_mode == SA_MASK_LRELU) x = mk[e] > 0.f ? x : x * ep.slope;
                else if (ep.mask_mode == SA_MASK_GELU) x *= gelu_grad_f(mk[e]);
            }
            v[e] = x;
        }
        if (ok[jj]) {
            if (ep.out_dtype == SA_F32) {
#pragma unroll
                for (int k = 0; k < 4; ++k) *(float4_t*)((float*)a.out + o[jj] + 4 * k) = (float4_t){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            } else if constexpr (F16IO) {
                // the f16 value for the next forward launch and (training) its bf16 copy for the backward pass, packed pair by pair so that the fp32
                // values die as they are consumed
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    u32x4 pk, pl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = pack2<f16_t>(v[8 * k + 2 * e], v[8 * k + 2 * e + 1]);
                        pl[e] = pack2<bf16_t>(v[8 * k + 2 * e], v[8 * k + 2 * e + 1]);
                    }
                    *(u32x4*)((bf16_t*)a.out + o[jj] + 8 * k) = pk;
                    if (ep.out_lp) *(u32x4*)((bf16_t*)ep.out_lp + o[jj] + 8 * k) = pl;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    u32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = pack2<bf16_t>(v[8 * k + 2 * e], v[8 * k + 2 * e + 1]);
                    *(u32x4*)((bf16_t*)a.out + o[jj] + 8 * k) = pk;
                }
            }
        }
    }
}

template <int MI, int NI>
__device__ __forceinline__ void epi_add_bias(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wn, uint32_t fq, uint32_t n_base) {
    if (a.ep.bias) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4_t bv = *(const float4_t*)(a.ep.bias + n_base + wn * 64 + i * 16 + fq * 4);
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] += bv;
        }
    }
}

// all batches of one (ADD, MASK, widths) specialisation
template <int MI, int NI, int JB, bool ADD, bool MASK, bool ADD32, bool MASK32, int BUDGET, bool F16IO, typename RowOv>
__device__ __forceinline__ void epi_run_from(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wm, uint32_t frow, uint32_t c0, RowOv row_ov, float alpha) {
    if constexpr (JB < MI) {
        epi_batch<MI, NI, JB, ADD, MASK, ADD32, MASK32, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        epi_run_from<MI, NI, JB + epi_batch_size<ADD, MASK, ADD32, MASK32, BUDGET>(), ADD, MASK, ADD32, MASK32, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
    }
}
template <int MI, int NI, bool ADD, bool MASK, bool ADD32, bool MASK32, int BUDGET, bool F16IO, typename RowOv>
__device__ __forceinline__ void epi_run(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wm, uint32_t frow, uint32_t c0, RowOv row_ov, float alpha) {
    static_assert(MI % 4 == 0, "row groups per wave must be a multiple of the largest batch");
    epi_run_from<MI, NI, 0, ADD, MASK, ADD32, MASK32, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
}

template <int MI, int NI, int BUDGET = 64, bool F16IO = false, typename RowOv>
__device__ __forceinline__ void fprop_epilogue_regs(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wm, uint32_t wn, uint32_t frow, uint32_t fq,
                                                    uint32_t n_base, RowOv row_ov, bool bias_done = false) {
    static_assert(NI == 4, "4 column fragments per wave");
    const sa_epilogue& ep = a.ep;
    const float alpha = ep.alpha ? *ep.alpha : 1.f;
    if (!bias_done) epi_add_bias<MI, NI>(a, acc, wn, fq, n_base);
    const uint32_t c0 = n_base + wn * 64 + fq * 16;   // this lane's 16 channels after the transpose
    const bool add = ep.addend != nullptr, mask = ep.mask_mode != SA_MASK_NONE;
    const bool a32 = ep.add_dtype == SA_F32, m32 = ep.mask_dtype == SA_F32;
    // block-uniform dispatch to a straight-line specialisation (loads of a batch back to back)
    if (!add && !mask) epi_run<MI, NI, false, false, false, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
    else if (add && !mask) {
        if (a32) epi_run<MI, NI, true, false, true, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        else epi_run<MI, NI, true, false, false, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
    } else if (!add && mask) {
        if (m32) epi_run<MI, NI, false, true, false, true, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        else epi_run<MI, NI, false, true, false, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
    } else {
        if (a32 && m32) epi_run<MI, NI, true, true, true, true, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        else if (!a32 && !m32) epi_run<MI, NI, true, true, false, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        else if (a32) epi_run<MI, NI, true, true, true, false, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
        else epi_run<MI, NI, true, true, false, true, BUDGET, F16IO>(a, acc, wm, frow, c0, row_ov, alpha);
    }
}


template <int BM, int BN, int WM, int WN, int MI, int NI, int NT = 256>
__device__ __forceinline__ void fprop_epilogue(const FpropArgs& a, float4_t (&acc)[NI][MI], unsigned char* smem, uint32_t tid, uint32_t wm, uint32_t wn,
                                               uint32_t frow, uint32_t fq, uint32_t m_base, uint32_t n_base) {
    fprop_epilogue_ov<BM, BN, WM, WN, MI, NI, NT>(a, acc, smem, tid, wm, wn, frow, fq, n_base,
                                                  [&](uint32_t row) __attribute__((always_inline)) { return linear_row_voxel(a, m_base + row); });
}


// 32-bit buffer offset that is out of bounds for every operand: LDS-DMA loads from it deliver zeros (padding taps, rows beyond M)
constexpr uint32_t OOB_OFF = 0xfffffff0u;

}  // namespace sa
