// Implicit-GEMM convolution forward / data-gradient on MFMA (gfx950).
//
//   out[o(m)][co] = epilogue( sum_{tap, ci} in[g(m, tap)][ci] * Wpk[co][tap*Cin + ci] )
//
// One kernel covers Conv3d (k4 s2, k3 s1, k1), ConvTranspose3d k4 s2 (as 8 output-parity classes of 2x2x2 taps),
// their data gradients and nn.Linear: only the gather geometry (sa_conv_geom) differs.  Activations are channels-last so
// a tap contributes one contiguous Cin-vector per voxel.
//
// Tiling: 256 threads = 4 waves.  Block tile BM=128 voxels x BN in {128,64,32,16} channels, K-slab = 128 BYTES per row
// (64 bf16 / 32 f32), double-buffered in LDS with a 16-byte XOR swizzle (conflict-free ds_read_b128 / ds_write_b128).
// The MFMA "A" operand is the WEIGHT tile and "B" the activation tile, so each lane ends up holding 4 consecutive output
// channels of one voxel -> 8/16-byte channels-last stores.
//   bf16: __builtin_amdgcn_mfma_f32_16x16x32_bf16   (one per 64 bytes of K)
//   f32 : __builtin_amdgcn_mfma_f32_16x16x4f32 x4   (exact fp32 fma chain; parity mode)
#pragma once
#include <type_traits>

#include "conv_fprop_common.h"

namespace sa {

template <typename T, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(256) void conv_fprop_kernel(const FpropArgs a) {
    constexpr int BM = WM * MI * 16;
    constexpr int BN = WN * NI * 16;
    static_assert(BM == 128, "A loader assumes 128 rows");
    static_assert(WM * WN == 4, "4 waves");
    constexpr int VEC = DT<T>::VEC;
    constexpr int B_IT = (BN + 31) / 32;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                      // 2 x BM x 128
    unsigned char* sB = smem + 2 * BM * 128;       // 2 x BN x 128

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave / WN, wn = wave % WN;

    const uint32_t nblk = gridDim.x;
    const uint32_t bid = xcd_remap(blockIdx.x, nblk);
    const uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    const uint32_t m_base = bm * BM, n_base = bn * BN;

    const sa_conv_geom& g = a.g;
    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ wpk = (const T*)a.wpk;

    // ---- loader state: this thread owns k-vector column `lv` of 4 activation rows and B_IT weight rows
    const uint32_t lv = tid & 7u;
    const uint32_t lr = tid >> 3;  // 0..31
    int32_t id0[4], ih0[4], iw0[4];
    int64_t rowbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = m_base + lr + 32u * j;
        if (m < a.M) {
            uint32_t q = fdiv(m, a.dW);
            const uint32_t wmx = m - q * g.Wm;
            uint32_t q2 = fdiv(q, a.dH);
            const uint32_t hmx = q - q2 * g.Hm;
            const uint32_t n = fdiv(q2, a.dD);
            const uint32_t dmx = q2 - n * g.Dm;
            id0[j] = (int32_t)dmx * g.in_mult[0] + g.in_off[0];
            ih0[j] = (int32_t)hmx * g.in_mult[1] + g.in_off[1];
            iw0[j] = (int32_t)wmx * g.in_mult[2] + g.in_off[2];
            rowbase[j] = (((int64_t)n * g.Di + id0[j]) * g.Hi + ih0[j]) * g.Wi + iw0[j];
        } else {
            id0[j] = ih0[j] = iw0[j] = -(1 << 28);  // always out of range -> zero rows
            rowbase[j] = 0;
        }
    }
    const T* wrow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) wrow[j] = wpk + (int64_t)(n_base + lr + 32u * j) * g.Kpad + lv * VEC;

    u32x4 ra[4], rb[B_IT];
    auto gload = [&](uint32_t s) __attribute__((always_inline)) {
        const uint32_t kv = s * 8u + lv;               // k index in 16-byte vectors
        const uint32_t tap = fdiv(kv, a.dCv);
        const uint32_t cv = kv - tap * a.dCv.d;
        const uint32_t td = fdiv(tap, a.dThw);
        const uint32_t t2 = tap - td * a.dThw.d;
        const uint32_t th = fdiv(t2, a.dTw);
        const uint32_t tw = t2 - th * a.dTw.d;
        const int32_t od = (int32_t)td * g.tap_step[0], oh = (int32_t)th * g.tap_step[1], ow = (int32_t)tw * g.tap_step[2];
        const int64_t tapoff = ((int64_t)od * g.Hi + oh) * g.Wi + ow;
        const bool tap_ok = tap < a.ntaps;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = tap_ok && (uint32_t)(id0[j] + od) < (uint32_t)g.Di && (uint32_t)(ih0[j] + oh) < (uint32_t)g.Hi &&
                            (uint32_t)(iw0[j] + ow) < (uint32_t)g.Wi;
            if (ok) ra[j] = *(const u32x4*)(in + (rowbase[j] + tapoff) * g.Cin + cv * VEC);
            else ra[j] = (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (BN >= 32 || lr < (uint32_t)BN) rb[j] = *(const u32x4*)(wrow[j] + (int64_t)s * (8 * VEC));
        }
    };
    auto lstore = [&](uint32_t buf) __attribute__((always_inline)) {
        unsigned char* pa = sA + buf * (BM * 128);
        unsigned char* pb = sB + buf * (BN * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)(pa + tile_off(lr + 32u * j, lv)) = ra[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (BN >= 32 || lr < (uint32_t)BN) *(u32x4*)(pb + tile_off(lr + 32u * j, lv)) = rb[j];
        }
    };

    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();

    const uint32_t frow = lane & 15u, fq = lane >> 4;
    for (uint32_t s = 0; s < a.nk; ++s) {
        const uint32_t buf = s & 1u;
        // prefetch the next K-slab unconditionally (clamped: the last iteration re-reads its own slab) so that the
        // staging registers stay in VGPRs instead of a scratch alloca
        gload((s + 1) < a.nk ? s + 1 : s);
        const unsigned char* pa = sA + buf * (BM * 128);
        const unsigned char* pb = sB + buf * (BN * 128);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 xf[MI], wf[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(pa + tile_off(wm * (MI * 16) + j * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + tile_off(wn * (NI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
        }
        lstore(buf ^ 1u);
        __syncthreads();
    }

    fprop_epilogue<BM, BN, WM, WN, MI, NI>(a, acc, smem, tid, wm, wn, frow, fq, m_base, n_base);
}

// Second GEMM of the fused residual block, on chip (bf16, 128 x 128 tile, 4 waves of 64 x 64):
//   out2[m][co] = sum_c relu(acc[m][c] + b1[c]) * W2[co][c]
// W2 (two 128-byte K-slabs) streams into LDS [32 KiB, 64 KiB) while h is converted and parked in [0, 32 KiB); `row_vox(row)` gives
// the voxel a tile row belongs to (for the optional h store), or -1.  Leaves the second product in `acc`; LDS is free on return.
// 8 halves -> 8 bf16 (the hidden activation of an f16 forward chain is saved for the bf16 backward pass)
__device__ __forceinline__ u32x4 cvt8_f16_to_bf16(const u32x4& h) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<bf16_t>(f16_to_f32((unsigned short)(h[e] & 0xffffu)), f16_to_f32((unsigned short)(h[e] >> 16)));
    return o;
}
template <typename T> __device__ __forceinline__ u32x4 hidden_row_for_backward(const u32x4& h) {
    if constexpr (std::is_same<T, f16_t>::value) return cvt8_f16_to_bf16(h);
    else return h;
}

template <typename T, int MI, int NI, typename RowVox>
__device__ __forceinline__ void resblock_second_gemm(const FpropArgs& a, float4_t (&acc)[NI][MI], unsigned char* smem, uint32_t tid, uint32_t wave, uint32_t wm,
                                                     uint32_t wn, uint32_t frow, uint32_t fq, uint32_t prow, uint32_t lv, RowVox row_vox) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 128;
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2pk, 0, 128 * 128 * 2, 0x00020000);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(smem + 2 * BM * 128 + sl * (BN * 128) + (wave * 4 + j) * 1024), 16,
                                                     ((wave * 4 + j) * 8 + prow) * 256u + lv * 16u, sl * 128u, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const uint32_t c0 = wn * (NI * 16) + i * 16 + fq * 4;       // 4 consecutive hidden channels of this lane
        const float4_t b1 = *(const float4_t*)(a.bias1 + c0);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const uint32_t row = wm * (MI * 16) + j * 16 + frow;
            const float4_t v = acc[i][j] + b1;
            uint2 pk;
            pk.x = pack2<T>(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
            pk.y = pack2<T>(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
            *(uint2*)(smem + (c0 >> 6) * (BM * 128) + tile_off(row, (c0 & 63u) >> 3) + (c0 & 7u) * 2) = pk;
            acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();  // h tile complete (and the W2 DMA drained)
    if (a.h_out) {    // training: the hidden activation is needed by the backward pass -> full 256-byte rows
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint32_t row = (tid >> 4) + 16u * it, sl = (tid >> 3) & 1u, vec = tid & 7u;
            const long long vox = row_vox(row);
            if (vox >= 0) *(u32x4*)((bf16_t*)a.h_out + (size_t)vox * 128 + sl * 64 + vec * 8) = hidden_row_for_backward<T>(*(const u32x4*)(smem + sl * (BM * 128) + tile_off(row, vec)));
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const unsigned char* pa = smem + sl * (BM * 128);
        const unsigned char* pb = smem + 2 * BM * 128 + sl * (BN * 128);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 xf[MI], wf[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(pa + tile_off(wm * (MI * 16) + j * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + tile_off(wn * (NI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
        }
    }
    __syncthreads();  // all waves done with the h / W2 tiles before the epilogue reuses the LDS
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Mainloop v2: LDS-DMA staging.  Both tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no
// ds_write pass, no exec-masked branches): out-of-range taps / rows use an out-of-bounds buffer offset, which the hardware
// turns into zeros.  The DMA writes lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the SOURCE:
// lane l of an 8-row x 128-byte piece fetches 16-byte vector (l&7) ^ (l>>3) of row (l>>3).  When a 128-byte K-slab never
// straddles two taps (Cin*sizeof(T) % 128 == 0, UNIFORM) the tap decode is scalar (SALU) work.
// Needs every operand < 4 GiB (32-bit buffer offsets); the register-staged kernel above is the fallback.
// Measured dead end for the narrow tiles (128 x 64, the dense layers of the Performer: 8 slabs of 16 MFMAs per wave): a three-buffer ring with
// two slabs in flight (counted s_waitcnt vmcnt + raw barrier) was 25-35 % SLOWER (512 -> 512 layer 25 -> 33 us, Performer step +10 %): the
// third buffer costs a resident block per CU (72 KiB against 48 KiB), and co-resident blocks hide the DMA round trip better than depth does.

// KG = 2 ("two K groups", round 4): TWICE the waves on the same tile -- waves [0, NW) reduce the first half of the K-slabs, waves [NW, 2 NW) the second half,
// each group through its own pair of stage buffers; group 1 then hands its accumulators to group 0 through LDS and all 2 NW waves write the tile back.
// For launches with about ONE tile per CU and a long reduction (the 512-column dense layers of the Performer at M = 8 400: 264 tiles, K = 1 024 .. 3 072), where a
// lone eight-wave block waits out every DMA round trip: sixteen waves per CU without needing more tiles (128 KiB of LDS: one block per CU).
template <typename T, int WM, int WN, int MI, int NI, bool UNIFORM, bool FUSE = false, int KG = 1>
__global__ __launch_bounds__(WM * WN * 64 * KG) void conv_fprop_dma_kernel(const FpropArgs a_) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins only exist in the device pass; the host pass needs just the stub
    FpropArgs a = a_;
    const uint32_t bid = select_class(a, a_);       // (one class: the launch's own geometry)
    constexpr int BM = WM * MI * 16;
    constexpr int BN = WN * NI * 16;
    constexpr int NW = WM * WN;                           // 4 waves, or 8 (half-size wave tiles: twice the waves per SIMD to cover DMA / LDS latency)
    static_assert((BM == 128 || (BM == 256 && NW == 8 && MI == 4 && NI == 4)) && (NW == 4 || NW == 8) && (!FUSE || NW == 4), "tile");
    static_assert(KG == 1 || (KG == 2 && !FUSE && BM == 128), "two K groups: plain 128-row tiles");
    constexpr int STAGE = 2 * (BM + BN) * 128;            // one group's two stage buffers
    constexpr int A_PER_WAVE = (BM / 8) / NW;              // 1 KiB pieces (8 rows) of the activation tile per wave
    constexpr int SZ = sizeof(T);
    constexpr int BKE = 128 / SZ;
    constexpr int B_PIECES = BN / 8;                       // 1 KiB pieces (8 rows) of the weight tile
    constexpr int B_PER_WAVE = (B_PIECES + NW - 1) / NW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = KG == 1 ? 0u : wave_all / NW;     // K group of this wave
    const uint32_t wave = KG == 1 ? wave_all : wave_all - kg * NW;
    unsigned char* const smem = smem_all + kg * STAGE;
    const uint32_t wm = wave / WN, wn = wave % WN;
    uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    if (a.group_m) {
        // dense layer: the blocks one XCD runs at a time cover group_m row tiles x (up to) all channel tiles, so its L2 fetches group_m activation
        // panels + the weight panels once instead of one activation panel PER BLOCK (row tiles fastest = no activation reuse inside an XCD at all)
        const uint32_t nbn = gridDim.x / a.nblk_m, per = a.group_m * nbn;
        const uint32_t gid = bid / per, first = gid * a.group_m, r = bid - gid * per;
        const uint32_t gsz = a.nblk_m - first < a.group_m ? a.nblk_m - first : a.group_m;
        bm = first + r % gsz;
        bn = r / gsz;
    }
    const uint32_t m_base = bm * BM, n_base = bn * BN;
    const sa_conv_geom& g = a.g;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);

    // ---- this lane's fixed role inside every 8-row piece
    const uint32_t prow = lane >> 3;                       // row within the piece
    const uint32_t lv = (lane & 7u) ^ prow;                // SOURCE 16-byte vector (swizzle on the source side)
    uint32_t rowoff[A_PER_WAVE], vm[A_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const uint32_t m = m_base + (wave * A_PER_WAVE + j) * 8 + prow;
        rowoff[j] = 0;
        vm[j] = 0;
        if (m < a.M) {
            uint32_t q = fdiv(m, a.dW);
            const uint32_t wmx = m - q * g.Wm;
            uint32_t q2 = fdiv(q, a.dH);
            const uint32_t hmx = q - q2 * g.Hm;
            const uint32_t n = fdiv(q2, a.dD);
            const uint32_t dmx = q2 - n * g.Dm;
            const int32_t id0 = (int32_t)dmx * g.in_mult[0] + g.in_off[0];
            const int32_t ih0 = (int32_t)hmx * g.in_mult[1] + g.in_off[1];
            const int32_t iw0 = (int32_t)wmx * g.in_mult[2] + g.in_off[2];
            // modular 32-bit byte offset of the (possibly virtual) base voxel
            rowoff[j] = (uint32_t)((((int32_t)n * g.Di + id0) * g.Hi + ih0) * g.Wi + iw0) * (uint32_t)(g.Cin * SZ);
            uint32_t mk = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < g.KT[0] && (uint32_t)(id0 + t * g.tap_step[0]) < (uint32_t)g.Di) mk |= 1u << t;
                if (t < g.KT[1] && (uint32_t)(ih0 + t * g.tap_step[1]) < (uint32_t)g.Hi) mk |= 16u << t;
                if (t < g.KT[2] && (uint32_t)(iw0 + t * g.tap_step[2]) < (uint32_t)g.Wi) mk |= 256u << t;
            }
            vm[j] = mk;
        }
    }
    uint32_t boff[B_PER_WAVE];
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) boff[j] = (n_base + (wave * B_PER_WAVE + j) * 8 + prow) * (uint32_t)(g.Kpad * SZ) + lv * 16u;

    auto issue = [&](uint32_t s, uint32_t buf) __attribute__((always_inline)) {
        unsigned char* pa = smem + buf * (BM * 128);
        unsigned char* pb = smem + 2 * BM * 128 + buf * (BN * 128);
        uint32_t sel, koff;
        bool tap_ok;
        if constexpr (UNIFORM) {
            const uint32_t ke = s * BKE;                    // scalar: first K element of the slab
            const uint32_t tap = fdiv(ke, a.dCin);
            const uint32_t c0 = ke - tap * g.Cin;
            const uint32_t td = fdiv(tap, a.dThw);
            const uint32_t t2 = tap - td * a.dThw.d;
            const uint32_t th = fdiv(t2, a.dTw);
            const uint32_t tw = t2 - th * a.dTw.d;
            const int32_t vox = (((int32_t)td * g.tap_step[0]) * g.Hi + (int32_t)th * g.tap_step[1]) * g.Wi + (int32_t)tw * g.tap_step[2];
            koff = (uint32_t)vox * (uint32_t)(g.Cin * SZ) + c0 * SZ + lv * 16u;
            sel = (1u << td) | (16u << th) | (256u << tw);
            tap_ok = tap < a.ntaps;
        } else {
            const uint32_t kv = s * 8u + lv;
            const uint32_t tap = fdiv(kv, a.dCv);
            const uint32_t cv = kv - tap * a.dCv.d;
            const uint32_t td = fdiv(tap, a.dThw);
            const uint32_t t2 = tap - td * a.dThw.d;
            const uint32_t th = fdiv(t2, a.dTw);
            const uint32_t tw = t2 - th * a.dTw.d;
            const int32_t vox = (((int32_t)td * g.tap_step[0]) * g.Hi + (int32_t)th * g.tap_step[1]) * g.Wi + (int32_t)tw * g.tap_step[2];
            koff = (uint32_t)vox * (uint32_t)(g.Cin * SZ) + cv * 16u;
            sel = (1u << td) | (16u << th) | (256u << tw);
            tap_ok = tap < a.ntaps;
        }
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j) {
            const bool ok = tap_ok && (vm[j] & sel) == sel;
            const uint32_t voff = ok ? rowoff[j] + koff : OOB_OFF;
#ifdef SA_PP_DEBUG_VARIANTS
            if (a.dbg & 64u) continue;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(pa + (wave * A_PER_WAVE + j) * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PER_WAVE; ++j) {
#ifdef SA_PP_DEBUG_VARIANTS
            if (a.dbg & 128u) continue;
#endif
            if (B_PIECES >= NW || wave * B_PER_WAVE + j < (uint32_t)B_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(pb + (wave * B_PER_WAVE + j) * 1024), 16, boff[j],
                                                         s * 128u, 0, 0);
        }
    };

    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // K-slabs [s0, s1) of this wave's group; every wave runs `trips` barrier intervals (group 1 may idle through the last one when nk is odd)
    const uint32_t nk0 = KG == 1 ? a.nk : (a.nk + 1u) / 2u;
    const uint32_t s0 = kg ? nk0 : 0u, s1 = kg ? a.nk : nk0, trips = nk0;
    if (s0 < s1) issue(s0, 0);
    __syncthreads();
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    for (uint32_t it = 0; it < trips; ++it) {
        const uint32_t s = s0 + it;
        const uint32_t buf = it & 1u;
        if (s + 1 < s1) issue(s + 1, buf ^ 1u);
        const unsigned char* pa = smem + buf * (BM * 128);
        const unsigned char* pb = smem + 2 * BM * 128 + buf * (BN * 128);
        if ((KG == 1 || s < s1)
#ifdef SA_PP_DEBUG_VARIANTS
            && !(a.dbg & 32u)
#endif
        ) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 xf[MI], wf[NI];
#pragma unroll
                for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(pa + tile_off(wm * (MI * 16) + j * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + tile_off(wn * (NI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
            }
        }
        __syncthreads();  // (the DMA in flight makes hipcc drain vmcnt(0) here: next slab landed, this one free)
    }
    if constexpr (KG == 2) {
        // group 1 -> group 0: accumulator fragments through group 1's (now idle) stage buffers, one 16-byte slot per (wave, fragment, lane): both groups
        // hold the same fragment of the tile in the same (wave, lane), so the hand-over is a straight copy + add
        float4_t* const xch = (float4_t*)(smem_all + STAGE);
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) xch[((wave * NI + i) * MI + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] += xch[((wave * NI + i) * MI + j) * 64 + lane];
        }
        __syncthreads();   // (the epilogue's staging tile reaches into group 1's buffers)
        fprop_epilogue_ov<BM, BN, WM, WN, MI, NI, NW * 64 * KG>(a, acc, smem_all, tid, wm, wn, frow, fq, n_base,
                                                                [&](uint32_t row) __attribute__((always_inline)) { return linear_row_voxel(a, m_base + row); }, kg == 0);
        return;
    }
    if constexpr (FUSE) {
        static_assert(!FUSE || (sizeof(T) == 2 && BM == 128 && BN == 128), "fused residual block: bf16, 128 x 128 tile");
        resblock_second_gemm<T, MI, NI>(a, acc, smem, tid, wave, wm, wn, frow, fq, prow, lv,
                                     [&](uint32_t row) __attribute__((always_inline)) { return m_base + row < a.M ? (long long)(m_base + row) : -1ll; });
    }
    if constexpr (BM == 256) {
        // 256 x 128 tile (eight waves of 64 x 64): full tiles of valid channels with 16-byte aligned rows leave through the register epilogue
        const bool regs_ok = n_base + BN <= (uint32_t)g.cout_valid && (g.Cout & 7) == 0 && !(a.dbg & 256u);
        if (regs_ok) {
            fprop_epilogue_regs<MI, NI, 24, std::is_same<T, f16_t>::value>(a, acc, wm, wn, frow, fq, n_base,
                                            [&](uint32_t row) __attribute__((always_inline)) { return linear_row_voxel(a, m_base + row); });
            return;
        }
    }
    fprop_epilogue<BM, BN, WM, WN, MI, NI, NW * 64>(a, acc, smem, tid, wm, wn, frow, fq, m_base, n_base);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Mainloop v5 "halo" (3x3x3, stride 1, `same` geometry; Cin * sizeof(T) a multiple of 128; Cout >= 65): the 27-tap im2col
// re-read of the activations is what bounds v2 (the L2 -> LDS path, not the MFMA), so this loop stages every activation byte
// ONCE per (kd, channel-chunk).  A tile is a 2-D patch of 8 (H) x 16 (W) output voxels of one depth plane; its halo image
// (10 x 18 input voxels x 128 bytes of channels, zero-filled outside the volume by the DMA's out-of-bounds rule) sits in LDS and
// the nine (kh, kw) taps of the group are nine K-slabs that read it at shifted rows.  No boundary masks anywhere: the padding
// is physically in the halo.  K order = (kd, chunk, kh, kw); the packed weights keep their (tap, channel) order and are
// addressed by column.  Per slab a wave issues 4 weight pieces and at most one halo piece of the NEXT group (double-buffered),
// against 4 + 4 in v2.  LDS: 2 x 23 KiB halo + 2 x 16 KiB weights = 78 KiB -> two blocks per CU.
// Measured dead ends (MI355X, C = 128 layer, 1.05-1.1 PFLOP/s here): a persistent 8-wave block on 16 x 16 patches (half the weight
// traffic, cross-tile prefetch, register epilogue) ran at 0.91-0.96 PFLOP/s -- eight waves in lock-step on one barrier lose the overlap two
// independent 4-wave blocks give each other; rotating the K-group order per block (to spread the weight reads over L2) changed nothing.
// A 256-voxel tile with 32-channel slabs (4 waves of 128 x 64, two blocks per CU, half the weight bytes and 3 instead of 5 DMA pieces per
// 32 MFMAs) was 5 % slower as well: what bounds this loop is the barrier interval (32 MFMAs per wave), not the weight bytes.
template <typename T, bool FUSE>
__global__ __launch_bounds__(256, 2) void conv_fprop_halo_kernel(const FpropArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WM = 2, WN = 2, MI = 4, NI = 4;
    constexpr int BM = 128, BN = 128;
    constexpr int PW = 16, HW_ = 18, HROWS = 180, HPIECES = 23;   // patch 8 x 16, halo 10 x 18 = 180 rows in 23 pieces
    constexpr int SZ = sizeof(T);
    constexpr int HALO_BYTES = HPIECES * 1024;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sA = smem;                       // 2 halo images
    unsigned char* const sB = smem + 2 * HALO_BYTES;      // 2 weight slabs
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave / WN, wn = wave % WN;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    const uint32_t n_base = bn * BN;
    const sa_conv_geom& g = a.g;
    // patch -> (n, d, h0, w0).  Tile order = (n, band of 4 patch rows, d, row in band, wp): the ~64 tiles an XCD works on at once then
    // span ~3 consecutive planes of ONE 32-row band (~3.6 MB of activations with their kd neighbours) instead of one whole plane
    // (6.9 MB with its neighbours), so the kd = 0 / 2 re-reads of a plane hit that XCD's 4 MB L2 (measured +3-4 % on the forward).
    const uint32_t per_vol = a.HP * a.WP * (uint32_t)g.Dm, band = 4u * a.WP * (uint32_t)g.Dm;
    const uint32_t pn = bm / per_vol, rv = bm - pn * per_vol;
    const uint32_t bc = rv / band, r2 = rv - bc * band;
    const uint32_t rows_c = a.HP - 4u * bc < 4u ? a.HP - 4u * bc : 4u;
    const uint32_t pd = r2 / (rows_c * a.WP), r3 = r2 - pd * rows_c * a.WP;
    const uint32_t hpi = r3 / a.WP, wp = r3 - hpi * a.WP, hp = 4u * bc + hpi;
    const int32_t h0 = (int32_t)hp * 8, w0 = (int32_t)wp * PW;
    // halo origin and per-dimension tap direction: input = output + in_off + t * tap_step, t = 0..2
    const int32_t oh = g.in_off[1] + (g.tap_step[1] < 0 ? 2 * g.tap_step[1] : 0), ow = g.in_off[2] + (g.tap_step[2] < 0 ? 2 * g.tap_step[2] : 0);

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);

    const uint32_t prow = lane >> 3;
    const uint32_t lv = (lane & 7u) ^ prow;
    // this wave's halo pieces: wave*6 + i, i < 6 (the last wave has 5)
    uint32_t hoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t r = (wave * 6 + i) * 8 + prow;
        const uint32_t hh = r / HW_, ww = r - hh * HW_;
        const int32_t ih = h0 + oh + (int32_t)hh, iw = w0 + ow + (int32_t)ww;
        const bool ok = r < (uint32_t)HROWS && (uint32_t)ih < (uint32_t)g.Hi && (uint32_t)iw < (uint32_t)g.Wi;
        hoff[i] = ok ? (uint32_t)(((int32_t)pn * g.Di * g.Hi + ih) * g.Wi + iw) * (uint32_t)(g.Cin * SZ) + lv * 16u : OOB_OFF;
    }
    uint32_t boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = (n_base + (wave * 4 + j) * 8 + prow) * (uint32_t)(g.Kpad * SZ) + lv * 16u;

    const uint32_t nchunk = (uint32_t)(g.Cin * SZ) / 128u;
    const uint32_t ngroups = 3u * nchunk;
    const uint32_t plane_bytes = (uint32_t)(g.Hi * g.Wi * g.Cin * SZ);

    // halo piece i of group gi -> image gi & 1
    auto issue_halo = [&](uint32_t gi, int i) __attribute__((always_inline)) {
        const uint32_t td = gi / nchunk, ch = gi - td * nchunk;
        const int32_t id = (int32_t)pd + g.in_off[0] + (int32_t)td * g.tap_step[0];
        const bool dok = (uint32_t)id < (uint32_t)g.Di;
        const uint32_t goff = (uint32_t)id * plane_bytes + ch * 128u;
        const uint32_t voff = dok && hoff[i] != OOB_OFF ? hoff[i] + goff : OOB_OFF;
#ifdef SA_PP_DEBUG_VARIANTS
        if (a.dbg & 1u) return;
#endif
        if (wave * 6 + i < (uint32_t)HPIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(sA + (gi & 1u) * HALO_BYTES + (wave * 6 + i) * 1024), 16, voff, 0, 0, 0);
    };
    // weight slab of (group gi, tap t9) -> buffer `buf`
    auto issue_w = [&](uint32_t gi, uint32_t t9, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t td = gi / nchunk, ch = gi - td * nchunk;
        const uint32_t col = ((td * 9u + t9) * (uint32_t)g.Cin) * SZ + ch * 128u;
#ifdef SA_PP_DEBUG_VARIANTS
        if (a.dbg & 2u) return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sB + buf * (BN * 128) + (wave * 4 + j) * 1024), 16, boff[j], col, 0, 0);
    };

    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const uint32_t frow = lane & 15u, fq = lane >> 4;
    // unswizzled byte address of (patch row wm*4 + j, column frow) in a halo image, vector fq
    uint32_t a0[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) a0[j] = ((wm * 4 + j) * HW_ + frow) * 128u + fq * 16u;
    const uint32_t b_off = tile_off(wn * (NI * 16) + frow, fq);
    // halo row offset of tap t along a dimension: (in_off + t*step) - origin = t or 2 - t
    const bool fh = g.tap_step[1] < 0, fw = g.tap_step[2] < 0;

#pragma unroll
    for (int i = 0; i < 6; ++i) issue_halo(0, i);
    issue_w(0, 0, 0);
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t gi = 0; gi < ngroups; ++gi) {
        const unsigned char* pa = sA + (gi & 1u) * HALO_BYTES;
        const bool next_group = gi + 1 < ngroups;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            // ---- prefetch: next weight slab, and one piece of the next group's halo image
            if (t9 < 8) issue_w(gi, t9 + 1, buf ^ 1u);
            else if (next_group) issue_w(gi + 1, 0, buf ^ 1u);
            if (t9 < 6 && next_group) issue_halo(gi + 1, t9);
            // ---- this slab
            const int th = t9 / 3, tw = t9 % 3;
            const uint32_t tapoff = (uint32_t)((fh ? 2 - th : th) * HW_ + (fw ? 2 - tw : tw)) * 128u;
            const unsigned char* pb = sB + buf * (BN * 128);
            uint32_t ax[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const uint32_t ad = a0[j] + tapoff;
                ax[j] = ad ^ (((ad >> 7) & 7u) << 4);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 xf[MI], wf[NI];
#pragma unroll
                for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(pa + (ax[j] ^ (ks * 64u)));
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + ((b_off + i * 2048u) ^ (ks * 64u)));
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
            }
            // The weight pieces of the next slab must have landed; the halo piece issued AFTER them in this slab may stay in flight for
            // one more slab (it comes from HBM, the weights from L2): vmcnt(1) instead of the vmcnt(0) a __syncthreads() would force.
            // It is retired by the next slab's wait, and the last one (t9 = 5) by the vmcnt(0) of t9 = 6.. before the group switch.
            if (t9 < 6 && next_group && wave * 6 + t9 < (uint32_t)HPIECES) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // (only if this wave issued one)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            buf ^= 1u;
        }
    }
    auto row_vox = [&](uint32_t row) __attribute__((always_inline)) {
        const uint32_t h = (uint32_t)h0 + (row >> 4), w = (uint32_t)w0 + (row & 15u);
        return h < (uint32_t)g.Ho && w < (uint32_t)g.Wo ? (((long long)pn * g.Do + pd) * g.Ho + h) * g.Wo + w : -1ll;
    };
#ifdef SA_PP_DEBUG_VARIANTS
    if (a.dbg & 64u) {
        if (acc[0][0][0] == 123.f) *(float*)a.out = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];
        return;
    }
#endif
    if constexpr (FUSE) {
        static_assert(!FUSE || sizeof(T) == 2, "fused residual block: bf16");
        resblock_second_gemm<T, MI, NI>(a, acc, smem, tid, wave, wm, wn, frow, fq, prow, lv, row_vox);
    }
    // full tile of valid channels and 16-byte aligned channel rows -> register epilogue; otherwise the LDS-staged one (block-uniform choice)
    const bool regs_ok = n_base + BN <= (uint32_t)g.cout_valid && (g.Cout & 7) == 0 && !(a.dbg & 256u);
    if (regs_ok) fprop_epilogue_regs<MI, NI, 32, std::is_same<T, f16_t>::value>(a, acc, wm, wn, frow, fq, n_base, row_vox);
    else fprop_epilogue_ov<BM, BN, WM, WN, MI, NI, 256>(a, acc, smem, tid, wm, wn, frow, fq, n_base, row_vox);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Mainloop v8 "halo, 256 voxels": what bounds v5 is the barrier interval (32 MFMAs per wave per weight slab).  Here a tile is a 16 x 16
// patch worked by FOUR waves of 128 x 64 outputs, so a 64-channel weight slab feeds 64 MFMAs per wave between barriers and half as many
// weight bytes per FLOP.  To keep two independent blocks per CU (LDS <= 80 KiB) the halo image (18 x 18 x 128 B = 41 pieces) is
// SINGLE-buffered: it is re-loaded at every (kd, chunk) switch behind a barrier, and that bubble is covered by the other block of the CU
// (the next group's first weight slab is already in flight).  LDS 41 KiB + 2 x 16 KiB = 73 KiB.  Register epilogue only.
// Measured on the C = 128 layer: data gradient 4.52 -> 4.30 ms (+5 %), plain forward +1.5 %; used for the non-fused launches.
#ifdef SA_TIMING
// dev instrumentation (-DSA_TIMING): s_memtime sums of wave 0 of every block: [0] slab compute phase, [1] wait for DMA (vmcnt), [2] wait at the slab
// barrier, [3] halo reload (issue .. barrier), [4] prologue, [5] second GEMM + epilogue, [6] whole kernel, [7] number of blocks
__device__ unsigned long long g_timing[8];
#define SA_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define SA_TACC(i, v) do { if (tid == 0) atomicAdd(&g_timing[i], (unsigned long long)(v)); } while (0)
#else
#define SA_T(var)
#define SA_TACC(i, v)
#endif
// NW = 8: the same 256-voxel tile worked by EIGHT waves of 64 x 64 outputs (<= 128 VGPRs -> two blocks = four waves per SIMD).  A lone
// 4-wave block runs at ~45 % of the MFMA rate (s_memtime: ~1 000 cycles of DMA issue, address arithmetic, LDS latency and barrier per
// 1 024-cycle slab) and its partner block spends half its life in halo reloads / the epilogue; with four waves per SIMD the other three
// cover those cycles.
#ifndef SA_EPI_BUDGET8
#define SA_EPI_BUDGET8 24   // VGPRs of packed addend / mask pieces per epilogue batch of the eight-wave kernels
#endif
// P2 (round 4): the 256-voxel tile as TWO planes of 8 x 16 voxels.  The halo image is two plane slots (10 x 18 rows each); a plane sits in the slot of its
// parity, so that consecutive kd groups -- visited innermost, (chunk, kd) order -- share one plane and a kd switch re-loads ONE plane (23 pieces) instead of the
// whole image (41): 8 plane loads per tile instead of 6 images (180 vs 246 KB), half-size reload bubbles, and 8-row patches waste less at 40 x 56 x 40
// (3 360 instead of 3 840 tiles).
template <typename T, bool FUSE = false, int NW = 4, bool P2 = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv_fprop_halo256_kernel(const FpropArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int MI = 32 / NW, NI = 4;
    constexpr int WPIECES = 16 / NW;          // weight pieces (1 KiB) per wave per slab
    constexpr int BN = 128;
    static_assert(!P2 || NW == 8, "two-plane tiles: eight waves");
    constexpr int HW_ = 18, HROWS = P2 ? 180 : 324, PLPIECES = 23, HPIECES = P2 ? 2 * PLPIECES : 41;   // (P2: rows / pieces per plane slot)
    constexpr int SZ = sizeof(T);
    constexpr int HALO_BYTES = HPIECES * 1024;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sA = smem;                   // 1 halo image
    unsigned char* const sB = smem + HALO_BYTES;      // 2 weight slabs
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 1, wn = wave & 1u;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    const uint32_t n_base = bn * BN;
    const sa_conv_geom& g = a.g;
    // tile order = (n, band of 2 patch rows, d (P2: plane pair), row in band, wp)
    const uint32_t ND = P2 ? a.DP : (uint32_t)g.Dm;
    const uint32_t per_vol = a.HP * a.WP * ND, band = 2u * a.WP * ND;
    const uint32_t pn = bm / per_vol, rv = bm - pn * per_vol;
    const uint32_t bc = rv / band, r2 = rv - bc * band;
    const uint32_t rows_c = a.HP - 2u * bc < 2u ? a.HP - 2u * bc : 2u;
    const uint32_t pd = r2 / (rows_c * a.WP), r3 = r2 - pd * rows_c * a.WP;
    const uint32_t hpi = r3 / a.WP, wp = r3 - hpi * a.WP, hp = 2u * bc + hpi;
    const int32_t h0 = (int32_t)hp * (P2 ? 8 : 16), w0 = (int32_t)wp * 16;
    const int32_t d0 = P2 ? (int32_t)pd * 2 : (int32_t)pd;     // first output plane of the tile
    const int32_t oh = g.in_off[1] + (g.tap_step[1] < 0 ? 2 * g.tap_step[1] : 0), ow = g.in_off[2] + (g.tap_step[2] < 0 ? 2 * g.tap_step[2] : 0);

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);

    const uint32_t prow = lane >> 3;
    const uint32_t lv = (lane & 7u) ^ prow;
    uint32_t boff[WPIECES];
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) boff[j] = (n_base + (wave * WPIECES + j) * 8 + prow) * (uint32_t)(g.Kpad * SZ) + lv * 16u;

    const uint32_t nchunk = (uint32_t)(g.Cin * SZ) / 128u;
    const uint32_t ngroups = 3u * nchunk;
    const uint32_t plane_bytes = (uint32_t)(g.Hi * g.Wi * g.Cin * SZ);
    const uint32_t vox_bytes = (uint32_t)(g.Cin * SZ);
    const uint32_t base_vox = (uint32_t)((int32_t)pn * g.Di * g.Hi * g.Wi);

    // group -> (kd tap, channel chunk): kd-major for the one-plane tile, (chunk, kd) for the two-plane one (kd switches keep a plane)
    auto group_td = [&](uint32_t gi) __attribute__((always_inline)) { return P2 ? gi % 3u : gi / nchunk; };
    auto group_ch = [&](uint32_t gi) __attribute__((always_inline)) { return P2 ? gi / 3u : gi - (gi / nchunk) * nchunk; };
    // one plane (absolute input plane id, channel chunk ch) -> LDS at `dst`: pieces wave, wave + NW, ...; offsets are recomputed here
    // (no registers held across the loop: the 128 accumulators need them)
    auto issue_plane = [&](int32_t id, uint32_t ch, unsigned char* dst, uint32_t npieces) __attribute__((always_inline)) {
        const bool dok = (uint32_t)id < (uint32_t)g.Di;
        const uint32_t goff = (uint32_t)id * plane_bytes + ch * 128u + lv * 16u;
#pragma unroll 1
        for (uint32_t p = wave; p < npieces; p += NW) {
            const uint32_t r = p * 8 + prow;
            const uint32_t hh = r / HW_, ww = r - hh * HW_;
            const int32_t ih = h0 + oh + (int32_t)hh, iw = w0 + ow + (int32_t)ww;
            const bool ok = dok && r < (uint32_t)HROWS && (uint32_t)ih < (uint32_t)g.Hi && (uint32_t)iw < (uint32_t)g.Wi;
            const uint32_t voff = ok ? (base_vox + (uint32_t)(ih * g.Wi + iw)) * vox_bytes + goff : OOB_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, voff, 0, 0, 0);
        }
    };
    // first input plane of group gi's tile planes
    auto group_plane = [&](uint32_t gi) __attribute__((always_inline)) { return d0 + g.in_off[0] + (int32_t)group_td(gi) * g.tap_step[0]; };
    // the halo image of group gi; `only_new` (P2, kd switch inside a chunk): just the plane the previous group did not hold
    auto issue_halo = [&](uint32_t gi, bool only_new) __attribute__((always_inline)) {
        const uint32_t ch = group_ch(gi);
        const int32_t pb = group_plane(gi);
        if constexpr (P2) {
            const int32_t pnew = g.tap_step[0] > 0 ? pb + 1 : pb;
            if (only_new) issue_plane(pnew, ch, sA + ((uint32_t)pnew & 1u) * (PLPIECES * 1024), PLPIECES);
            else {
                issue_plane(pb, ch, sA + ((uint32_t)pb & 1u) * (PLPIECES * 1024), PLPIECES);
                issue_plane(pb + 1, ch, sA + ((uint32_t)(pb + 1) & 1u) * (PLPIECES * 1024), PLPIECES);
            }
        } else {
            issue_plane(pb, ch, sA, HPIECES);
        }
    };
    auto issue_w = [&](uint32_t gi, uint32_t t9, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t td = group_td(gi), ch = group_ch(gi);
        const uint32_t col = ((td * 9u + t9) * (uint32_t)g.Cin) * SZ + ch * 128u;
#pragma unroll
        for (int j = 0; j < WPIECES; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sB + buf * (BN * 128) + (wave * WPIECES + j) * 1024), 16, boff[j], col, 0, 0);
    };

    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

#ifdef SA_MFMA32_PROBE
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    f32x16_t acc32[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
#endif
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    // unswizzled; patch row j adds 18 * 128 j.  P2: the wave's rows are rows (wm & 1) * 4 .. of tile plane wm >> 1, whose slot depends on the group
    const uint32_t a_base = P2 ? (((wm & 1u) * (uint32_t)MI) * HW_ + frow) * 128u + fq * 16u : ((wm * (uint32_t)MI) * HW_ + frow) * 128u + fq * 16u;
    const uint32_t b_off = tile_off(wn * (NI * 16) + frow, fq);
    const bool fh = g.tap_step[1] < 0, fw = g.tap_step[2] < 0;

    SA_T(t_k0);
    issue_halo(0, false);
    issue_w(0, 0, 0);
    __syncthreads();
    SA_T(t_k1);
    SA_TACC(4, t_k1 - t_k0);
#ifdef SA_TIMING
    unsigned long long t_prev = t_k1, s_comp = 0, s_dma = 0, s_bar = 0, s_halo = 0;
#endif
    uint32_t buf = 0;
    for (uint32_t gi = 0; gi < ngroups; ++gi) {
        const bool next_group = gi + 1 < ngroups;
        const uint32_t a_grp = P2 ? a_base + (((uint32_t)(group_plane(gi) + (int32_t)(wm >> 1))) & 1u) * (PLPIECES * 1024) : a_base;   // (slot of this wave's plane)
#pragma unroll 1
        for (uint32_t t9 = 0; t9 < 9; ++t9) {
            const uint32_t th = t9 / 3u, tw = t9 - th * 3u;
            const uint32_t tapoff = ((fh ? 2u - th : th) * (uint32_t)HW_ + (fw ? 2u - tw : tw)) * 128u;
            const unsigned char* pb = sB + buf * (BN * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) {   // (issued after the first half of the slab's MFMAs instead: 2-3 % slower)
                    const bool same = t9 < 8;
                    if (same || next_group) issue_w(same ? gi : gi + 1, same ? t9 + 1 : 0, buf ^ 1u);
                }
                u32x4 xf[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + ((b_off + i * 2048u) ^ (ks * 64u)));
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const uint32_t ad = a_grp + tapoff + (uint32_t)j * (HW_ * 128u);
                    xf[j] = *(const u32x4*)(sA + ((ad ^ (((ad >> 7) & 7u) << 4)) ^ (ks * 64u)));
                }
#ifdef SA_MFMA32_PROBE
                // TIMING PROBE ONLY (wrong results): the same fragments through half as many v_mfma_f32_32x32x16_bf16 -- what the other instruction
                // shape would buy this loop at unchanged LDS traffic
                if constexpr (sizeof(T) == 2 && NW == 8) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                            for (int j2 = 0; j2 < 2; ++j2)
                                acc32[i2][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const short8_t*)&wf[i2 * 2 + kk], *(const short8_t*)&xf[j2 * 2 + kk], acc32[i2][j2], 0, 0, 0);
                } else
#endif
                {
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
                }
            }
#ifdef SA_TIMING
            asm volatile("" ::: "memory");
            SA_T(t_a);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SA_T(t_b);
            __syncthreads();
            SA_T(t_c);
            s_comp += t_a - t_prev; s_dma += t_b - t_a; s_bar += t_c - t_b; t_prev = t_c;
#else
            __syncthreads();   // next weight slab landed (vmcnt(0)), this one free
#endif
            buf ^= 1u;
        }
        if (next_group) {      // every wave is past its last read of the halo image: reload it (the first weight slab of the group is in flight)
            issue_halo(gi + 1, P2 && group_td(gi + 1) != 0u);
            __syncthreads();
#ifdef SA_TIMING
            SA_T(t_h);
            s_halo += t_h - t_prev; t_prev = t_h;
#endif
        }
    }
#ifdef SA_TIMING
    SA_TACC(0, s_comp); SA_TACC(1, s_dma); SA_TACC(2, s_bar); SA_TACC(3, s_halo);
#endif
    auto row_vox = [&](uint32_t row) __attribute__((always_inline)) {
        const uint32_t pr = row >> 4, w = (uint32_t)w0 + (row & 15u);
        const uint32_t h = (uint32_t)h0 + (P2 ? (pr & 7u) : pr), d = (uint32_t)d0 + (P2 ? (pr >> 3) : 0u);
        return d < (uint32_t)g.Do && h < (uint32_t)g.Ho && w < (uint32_t)g.Wo ? (((long long)pn * g.Do + d) * g.Ho + h) * g.Wo + w : -1ll;
    };
#ifdef SA_MFMA32_PROBE
    if constexpr (sizeof(T) == 2 && NW == 8) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = acc32[i >> 1][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + r];
    }
#endif
    if constexpr (FUSE) {
        static_assert(!FUSE || sizeof(T) == 2, "fused residual block: bf16");
        // Second GEMM of the residual block for the 256 rows: h = relu(acc + b1) goes to LDS as two [256][128 B] K-slabs (64 KiB, the
        // ring is free: the loop's last barrier), the 1x1x1 weights are the MFMA A operand held in REGISTERS (16 fragments per wave, read
        // from the packed operand [128 co][128 c]), so nothing but h needs LDS and the block stays at two per CU.
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t c0 = wn * (NI * 16) + i * 16 + fq * 4;
            const float4_t b1 = *(const float4_t*)(a.bias1 + c0);
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const uint32_t row = wm * (MI * 16) + j * 16 + frow;
                const float4_t v = acc[i][j] + b1;
                uint2 pk;
                pk.x = pack2<T>(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
                pk.y = pack2<T>(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                *(uint2*)(smem + (c0 >> 6) * (256 * 128) + tile_off(row, (c0 & 63u) >> 3) + (c0 & 7u) * 2) = pk;
                acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
            }
        }
        constexpr int W2K = NW == 4 ? 4 : 1;   // 4 waves: all 16 fragments up front (registers to spare); 8 waves: 4 per K step (128-VGPR budget)
        u32x4 w2[NI][W2K];
        const bf16_t* const w2p = (const bf16_t*)a.w2pk;
        if constexpr (NW == 4) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) w2[i][ks] = *(const u32x4*)(w2p + (wn * 64 + i * 16 + frow) * 128 + ks * 32 + fq * 8);
        }
        __syncthreads();  // h tile complete
        if (a.h_out) {    // training: the hidden activation is needed by the backward pass -> full 256-byte rows
#pragma unroll
            for (int it = 0; it < 64 / NW; ++it) {
                const uint32_t row = (tid >> 4) + (uint32_t)(NW * 4) * it, sl = (tid >> 3) & 1u, vec = tid & 7u;
                const long long vox = row_vox(row);
                if (vox >= 0) *(u32x4*)((bf16_t*)a.h_out + (size_t)vox * 128 + sl * 64 + vec * 8) = hidden_row_for_backward<T>(*(const u32x4*)(smem + sl * (256 * 128) + tile_off(row, vec)));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 xf[MI];
            if constexpr (NW != 4) {
#pragma unroll
                for (int i = 0; i < NI; ++i) w2[i][0] = *(const u32x4*)(w2p + (wn * 64 + i * 16 + frow) * 128 + ks * 32 + fq * 8);
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(smem + (ks >> 1) * (256 * 128) + tile_off(wm * (MI * 16) + j * 16 + frow, (ks & 1) * 4 + fq));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], w2[i][NW == 4 ? ks : 0], xf[j]);
        }
    }
    fprop_epilogue_regs<MI, NI, (NW == 4 ? 64 : SA_EPI_BUDGET8), std::is_same<T, f16_t>::value>(a, acc, wm, wn, frow, fq, n_base, row_vox);
#ifdef SA_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SA_T(t_end);
    SA_TACC(5, t_end - t_prev); SA_TACC(6, t_end - t_k0); SA_TACC(7, 1);
#endif
#endif
}

#if defined(SA_TIMING) && defined(SA_FPROP_MAIN_TU)
}  // namespace sa
extern "C" int sa_debug_timing(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(sa::g_timing), 64);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(sa::g_timing), z, 64); }
    return 0;
}
namespace sa {
#endif

// ------------------------------------------------------------------------------------------------------------------------
// Mainloop v10 "cells256" (round 5): the stride-2 family in CELL coordinates with 256-voxel tiles.  Conv3d k4 s2 p1 (eight tap-parity classes) and the eight
// output-parity classes of ConvTranspose3d k4 s2 p1 are 2 x 2 x 2 tap groups over a neighbourhood of cells: for the strided convolution the classes are the
// parities (t0) of the taps per axis, taps k = t0 + 2 c (c = 0, 1) read input 2 (m + c) + t0 - 1 = cell m + c of the class's sub-grid.  (Round 4's 128-voxel
// form of this loop, conv_fprop_cells_kernel, tied with the im2col-order kernel and left the tree in round 6: DESIGN Appendix A.)  Ablation of the im2col-order loop on the config-2 down-sampling layer at
// batch 8 (tools/conv_ablate.sh: 1.90 ms in full, 1.17 ms without the activation DMA, 1.43 ms without the weight DMA, 0.98 ms with neither) says the stride-2
// family is bound by what it pulls through L2 -> L1 -> LDS: 2 MB of activation rows AND 2 MB of weight slabs per 128 x 128 tile (23 GB per launch, ~15 TB/s).
// The 128-voxel cell tile cuts the activation half (each staged halo row serves eight taps) and ties; the weight half only shrinks with more voxels per block.
// Here a tile is 4 (D) x 8 (H) x 8 (W) cells = 256 voxels (no padded rows at 40 x 56 x 40), eight waves of 64 voxels x 64 channels (wave = (depth plane, channel
// half), register epilogue) like the 3x3x3 kernels, and the halo is kept as FOUR plane slots of 9 x 9 cells (11 pieces each): tap depth shift 0 reads planes
// 0..3, shift 1 reads planes 1..4 -- plane 4 takes plane 0's slot, so a depth switch re-loads ONE plane.  44 + 2 x 16 KiB = 76 KiB: two blocks per CU.
// Per 256 voxels and (class, chunk): 55 KiB of halo + 128 KiB of weight slabs, against 2 x (31 + 128) KiB for two 128-voxel cell tiles and 2 x 256 KiB in
// im2col order.
template <typename T>
__global__ __launch_bounds__(512, 2) void conv_fprop_cells256_kernel(const FpropArgs a_) {
#if defined(__HIP_DEVICE_COMPILE__)
    FpropArgs a = a_;
    const uint32_t bid = select_class(a, a_);
    constexpr int NW = 8, MI = 4, NI = 4, WPIECES = 16 / NW, BN = 128;
    constexpr int PROWS = 81, PPIECES = 11, PLB = PPIECES * 1024, NSLOT = 4;    // plane: 9 x 9 cells in 11 pieces of 8 rows; slot stride
    constexpr int SZ = sizeof(T);
    constexpr int HALO_BYTES = NSLOT * PLB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sA = smem;                   // 4 plane slots
    unsigned char* const sB = smem + HALO_BYTES;      // 2 weight slabs
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 1, wn = wave & 1u;    // wm = depth plane of the tile
    const uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    const uint32_t n_base = bn * BN;
    const sa_conv_geom& g = a.g;
    // tile -> (n, d quad, hp, wp)
    uint32_t q = bm / a.WP;
    const uint32_t wp = bm - q * a.WP;
    uint32_t q2 = q / a.HP;
    const uint32_t hp = q - q2 * a.HP;
    const uint32_t pn = q2 / a.DP, dp = q2 - pn * a.DP;
    const int32_t d0 = (int32_t)dp * 4, h0 = (int32_t)hp * 8, w0 = (int32_t)wp * 8;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);

    const uint32_t prow = lane >> 3;
    const uint32_t lv = (lane & 7u) ^ prow;
    uint32_t boff[WPIECES];
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) boff[j] = (n_base + (wave * WPIECES + j) * 8 + prow) * (uint32_t)(g.Kpad * SZ) + lv * 16u;

    const uint32_t npar = (uint32_t)g.in_mult[1];                 // 2: strided convolution (eight tap classes); 1: one class
    const uint32_t nchunk = (uint32_t)(g.Cin * SZ) / 128u;
    const uint32_t ngroups = npar * npar * npar * nchunk * 2u;    // group = (class, channel chunk, depth shift): four (kh, kw) slabs
    const uint32_t vox_bytes = (uint32_t)(g.Cin * SZ);

    auto axis_base = [&](int ax, uint32_t pcls) __attribute__((always_inline)) {
        return g.in_off[ax] + (npar == 2u ? (int32_t)pcls * g.tap_step[ax] : (g.tap_step[ax] < 0 ? g.tap_step[ax] : 0));
    };
    auto axis_tap = [&](int ax, uint32_t pcls, uint32_t c) __attribute__((always_inline)) {
        return npar == 2u ? pcls + 2u * c : (g.tap_step[ax] < 0 ? 1u - c : c);
    };
    // halo plane `pl` (0..4, relative to the tile's first depth plane) of (class, chunk) -> slot pl & 3: pieces wave, wave + 8
    auto issue_plane = [&](uint32_t cls, uint32_t ch, uint32_t pl) __attribute__((always_inline)) {
        const int32_t bd = axis_base(0, cls >> 2), bh = axis_base(1, (cls >> 1) & 1u), bw = axis_base(2, cls & 1u);
        const int32_t id = g.in_mult[0] * (d0 + (int32_t)pl) + bd;
        const bool dok = (uint32_t)id < (uint32_t)g.Di;
        unsigned char* const dst = sA + (pl & 3u) * PLB;
#pragma unroll 1
        for (uint32_t p = wave; p < (uint32_t)PPIECES; p += (uint32_t)NW) {
            const uint32_t r = p * 8 + prow;
            const uint32_t hh = r / 9u, ww = r - hh * 9u;
            const int32_t ih = g.in_mult[1] * (h0 + (int32_t)hh) + bh, iw = g.in_mult[2] * (w0 + (int32_t)ww) + bw;
            const bool ok = dok && r < (uint32_t)PROWS && (uint32_t)ih < (uint32_t)g.Hi && (uint32_t)iw < (uint32_t)g.Wi;
            const uint32_t voff = ok ? (uint32_t)((((int32_t)pn * g.Di + id) * g.Hi + ih) * g.Wi + iw) * vox_bytes + ch * 128u + lv * 16u : OOB_OFF;
#ifdef SA_PP_DEBUG_VARIANTS
            if (a.dbg & 64u) continue;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, voff, 0, 0, 0);
        }
    };
    // group -> (class, chunk, depth shift)
    auto g_cls = [&](uint32_t gi) __attribute__((always_inline)) { return (gi >> 1) / nchunk; };
    auto g_ch = [&](uint32_t gi) __attribute__((always_inline)) { return (gi >> 1) - ((gi >> 1) / nchunk) * nchunk; };
    // the planes group gi needs that its predecessor did not leave in the slots: shift 0 = planes 0..3 (a new class / chunk), shift 1 = plane 4 only
    auto issue_halo = [&](uint32_t gi) __attribute__((always_inline)) {
        const uint32_t cls = g_cls(gi), ch = g_ch(gi);
        if (gi & 1u) issue_plane(cls, ch, 4u);
        else {
            issue_plane(cls, ch, 0u);
            issue_plane(cls, ch, 1u);
            issue_plane(cls, ch, 2u);
            issue_plane(cls, ch, 3u);
        }
    };
    // weight slab of (group gi, tap t4 = (kh shift, kw shift)) -> buffer `buf`
    auto issue_w = [&](uint32_t gi, uint32_t t4, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t cls = g_cls(gi), ch = g_ch(gi);
        const uint32_t kd = axis_tap(0, cls >> 2, gi & 1u), kh = axis_tap(1, (cls >> 1) & 1u, (t4 >> 1) & 1u), kw = axis_tap(2, cls & 1u, t4 & 1u);
        const uint32_t col = (((kd * (uint32_t)g.KT[1] + kh) * (uint32_t)g.KT[2] + kw) * (uint32_t)g.Cin) * SZ + ch * 128u;
#ifdef SA_PP_DEBUG_VARIANTS
        if (a.dbg & 128u) return;
#endif
#pragma unroll
        for (int j = 0; j < WPIECES; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(sB + buf * (BN * 128) + (wave * WPIECES + j) * 1024), 16, boff[j], col, 0, 0);
    };

    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const uint32_t frow = lane & 15u, fq = lane >> 4;
    // unswizzled byte address inside a plane slot of tile voxel 16 j + frow of this wave's plane = (h, w) = (v >> 3, v & 7), vector fq
    uint32_t a0[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const uint32_t v = (uint32_t)j * 16u + frow;
        a0[j] = ((v >> 3) * 9u + (v & 7u)) * 128u + fq * 16u;
    }
    const uint32_t b_off = tile_off(wn * (NI * 16) + frow, fq);

    issue_halo(0);
    issue_w(0, 0, 0);
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t gi = 0; gi < ngroups; ++gi) {
        const bool next_group = gi + 1 < ngroups;
        const uint32_t slot = ((wm + (gi & 1u)) & 3u) * PLB;     // this wave's depth plane, shifted by the group's depth tap
#pragma unroll 1
        for (uint32_t t4 = 0; t4 < 4; ++t4) {
            const bool same = t4 < 3;
            if (same || next_group) issue_w(same ? gi : gi + 1, same ? t4 + 1 : 0, buf ^ 1u);
            const uint32_t tapoff = (((t4 >> 1) & 1u) * 9u + (t4 & 1u)) * 128u;
            const unsigned char* pb = sB + buf * (BN * 128);
            uint32_t ax[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const uint32_t ad = a0[j] + tapoff;
                ax[j] = slot + (ad ^ (((ad >> 7) & 7u) << 4));
            }
#ifdef SA_PP_DEBUG_VARIANTS
            if (!(a.dbg & 32u))
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 xf[MI], wf[NI];
#pragma unroll
                for (int j = 0; j < MI; ++j) xf[j] = *(const u32x4*)(sA + (ax[j] ^ (ks * 64u)));
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const u32x4*)(pb + ((b_off + i * 2048u) ^ (ks * 64u)));
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) mma_slab<T>(acc[i][j], wf[i], xf[j]);
            }
            __syncthreads();   // next weight slab landed (vmcnt(0)), this one free
            buf ^= 1u;
        }
        if (next_group) {      // every wave is past its last read of the planes the next group replaces
            issue_halo(gi + 1);
            __syncthreads();
        }
    }
    auto row_vox = [&](uint32_t row) __attribute__((always_inline)) {
        const uint32_t d = (uint32_t)d0 + (row >> 6), h = (uint32_t)h0 + ((row >> 3) & 7u), w = (uint32_t)w0 + (row & 7u);
        if (d >= (uint32_t)g.Dm || h >= (uint32_t)g.Hm || w >= (uint32_t)g.Wm) return -1ll;
        return (((long long)pn * g.Do + (d * g.out_mult[0] + g.out_off[0])) * g.Ho + (h * g.out_mult[1] + g.out_off[1])) * g.Wo + (w * g.out_mult[2] + g.out_off[2]);
    };
    fprop_epilogue_regs<MI, NI, SA_EPI_BUDGET8, std::is_same<T, f16_t>::value>(a, acc, wm, wn, frow, fq, n_base, row_vox);
#endif
}

// compute units of the current device (queried once per device ordinal)
static inline int device_cu_count() {
    static std::atomic<int> cu_count[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int cus = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (cus <= 0) {
        cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cu_count[dev & 63].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

template <typename T, int WM, int WN, int MI, int NI>
static int launch_fprop(const FpropArgs& a, hipStream_t st) {
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    const uint32_t nbn = (uint32_t)a.g.CoutPad / BN;
    // only tiles that contain valid channels
    const uint32_t nbn_valid = ((uint32_t)a.g.cout_valid + BN - 1) / BN;
    (void)nbn;
    const size_t pipe = 2 * (BM + BN) * 128, epi = (size_t)BM * (BN + 4) * 4 + BM * 8;
    const size_t lds = pipe > epi ? pipe : epi;
    const uint32_t ncls = a.ncls > 1u ? a.ncls : 1u;
    dim3 grid(a.nblk_m * nbn_valid * ncls);
    if (ncls > 1u && a.in_bytes == 0) return SA_EUNSUPPORTED;      // (several classes per launch: the LDS-DMA kernels only)
    if (a.in_bytes != 0) {  // every operand addressable with 32-bit buffer offsets -> LDS-DMA mainloop
        const bool uniform = ((size_t)a.g.Cin * sizeof(T)) % 128 == 0;
        if constexpr (BM == 256) {
            static std::atomic<uint64_t> attr_done{0};
            configure_once_per_device(attr_done, [] {
                (void)hipFuncSetAttribute((const void*)conv_fprop_dma_kernel<T, WM, WN, MI, NI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)conv_fprop_dma_kernel<T, WM, WN, MI, NI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            });
        }
        FpropArgs b = a;
        // Dense layers with >= 8 channel tiles (M = 8 400 rows, tools/bench_dense_tiles.py; FETCH_SIZE per launch 203 -> 65 MB for q|k|v): q|k|v forward 77.1 -> 68.0 us,
        // w1 forward 52.8 -> 45.0, w2 data gradient 52.1 -> 43.1, to_out data gradient 30.5 -> 26.3.  With 4 channel tiles (N = 512, one block per CU) the fetch
        // bytes halve as well (139 -> 65 MB) but the time does not move (52.1 -> 54.7 us): those launches wait on the per-slab DMA round trip, not on the fabric.
        // SA_PP_DBG bits 16-23: group size override for A/B runs (255 = off).
        if (b.ntaps == 1 && nbn_valid > 1) b.group_m = (g_tunables.pp_dbg >> 16) ? ((g_tunables.pp_dbg >> 16) & 255u) % 255u : (nbn_valid >= 8 ? 8u : 0u);
        if constexpr (std::is_same<T, bf16_t>::value && WM == 4 && WN == 2 && MI == 2 && NI == 4) {
            // At most one 128 x 128 tile per CU and a long reduction: a lone eight-wave block per CU waits out every DMA round trip with nothing else resident.
            // Two K groups = sixteen waves on the same tile (KG = 2 above), 128 KiB of LDS, i.e. ONE block per CU -- so only grids that fit one round.
            // Measured (tools/bench_dense_tiles.py, K = 1 024 / 2 048 / 3 072 -> 512 columns): M = 8 192 (256 tiles) 25.1 / 37.3 / 48.3 -> 21.7 / 31.6 / 40.9 us;
            // M = 8 400 (264 tiles: a second round of 8) 26.9 / 39.6 / 52.0 -> 34.2 / 50.2 / 70.4 us, which is why the README batch stays on the one-group kernel.
            // SA_NO_KGROUPS / SA_DBG_NO_KGROUPS keeps the one-group kernel everywhere (A/B runs and the equality test).
            if (uniform && b.ntaps == 1 && b.nk >= 8 && ncls == 1u && (int)grid.x <= device_cu_count() && !dbg(SA_DBG_NO_KGROUPS)) {
                static std::atomic<uint64_t> attr2_done{0};
                configure_once_per_device(attr2_done, [] {
                    (void)hipFuncSetAttribute((const void*)conv_fprop_dma_kernel<T, WM, WN, MI, NI, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                });
                (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_dma_kernel<%s, %d, %d, %d, %d, true, false, 2>", tname<T>(), WM, WN, MI, NI), note_kernel(g_last_conv_kernel));
                hipLaunchKernelGGL((conv_fprop_dma_kernel<T, WM, WN, MI, NI, true, false, 2>), grid, dim3(WM * WN * 64 * 2), 2 * pipe, st, b);
                SA_CHECK_LAUNCH();
                return 0;
            }
        }
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_dma_kernel<%s, %d, %d, %d, %d, %s, false, 1>", tname<T>(), WM, WN, MI, NI, uniform ? "true" : "false"), note_kernel(g_last_conv_kernel));
        if (uniform) hipLaunchKernelGGL((conv_fprop_dma_kernel<T, WM, WN, MI, NI, true>), grid, dim3(WM * WN * 64), lds, st, b);
        else hipLaunchKernelGGL((conv_fprop_dma_kernel<T, WM, WN, MI, NI, false>), grid, dim3(WM * WN * 64), lds, st, b);
        SA_CHECK_LAUNCH();
        return 0;
    }
    if constexpr (WM * WN == 4 && !std::is_same<T, f16_t>::value) {   // (f16 forward operands: DMA-addressable operands only)
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_kernel<%s, %d, %d, %d, %d>", tname<T>(), WM, WN, MI, NI), note_kernel(g_last_conv_kernel));
        hipLaunchKernelGGL((conv_fprop_kernel<T, WM, WN, MI, NI>), grid, dim3(256), lds, st, a);
        SA_CHECK_LAUNCH();
        return 0;
    } else {
        return SA_EUNSUPPORTED;   // (the eight-wave tiles are only dispatched with 32-bit addressable operands)
    }
}

// the halo mainloop applies to 3x3x3 / stride 1 / `same` geometry (forward, and the data gradient with the taps reversed)
static bool halo_eligible(const FpropArgs& a, int sz) {
    const sa_conv_geom& g = a.g;
    const bool off = dbg(SA_DBG_NO_HALO);
    if (off || a.in_bytes == 0 || g.cout_valid <= 64 || ((size_t)g.Cin * sz) % 128 != 0) return false;
    for (int d = 0; d < 3; ++d) {
        if (g.KT[d] != 3 || g.in_mult[d] != 1 || g.out_mult[d] != 1 || g.out_off[d] != 0) return false;
        if (g.tap_step[d] != 1 && g.tap_step[d] != -1) return false;
        if (g.in_off[d] + (g.tap_step[d] < 0 ? 2 * g.tap_step[d] : 0) != -1) return false;  // halo origin one voxel before the patch
    }
    if (g.Dm != g.Do || g.Hm != g.Ho || g.Wm != g.Wo || g.Di != g.Do || g.Hi != g.Ho || g.Wi != g.Wo) return false;
    if ((size_t)g.Kpad != (size_t)27 * g.Cin) return false;
    const int hp = (g.Ho + 7) / 8, wp = (g.Wo + 15) / 16;
    const double eff = (double)g.Ho * g.Wo / ((double)hp * 8 * wp * 16);
    return eff >= 0.8 && (int64_t)g.N * g.Dm * hp * wp >= 512;
}

template <typename T, bool FUSE>
static int launch_fprop_halo(FpropArgs a, hipStream_t st) {
    a.HP = (uint32_t)(a.g.Ho + 7) / 8;
    a.WP = (uint32_t)(a.g.Wo + 15) / 16;
    a.nblk_m = (uint32_t)a.g.N * (uint32_t)a.g.Dm * a.HP * a.WP;
    const uint32_t nbn_valid = ((uint32_t)a.g.cout_valid + 127) / 128;
    const size_t pipe = 2 * 23 * 1024 + 2 * 128 * 128, epi = (size_t)128 * (128 + 4) * 4 + 128 * 8;
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (one static per template instance)
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)conv_fprop_halo_kernel<T, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });
    (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_halo_kernel<%s, %s>", tname<T>(), FUSE ? "true" : "false"), note_kernel(g_last_conv_kernel));
    hipLaunchKernelGGL((conv_fprop_halo_kernel<T, FUSE>), dim3(a.nblk_m * nbn_valid), dim3(256), pipe > epi ? pipe : epi, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

// the 256-voxel variant: register epilogue only (full, aligned 128-channel tiles), 16 x 16 patches that tile the plane well
static bool halo256_eligible(const FpropArgs& a, int sz) {
    const sa_conv_geom& g = a.g;
    if (dbg(SA_DBG_NO_HALO256) || !halo_eligible(a, sz)) return false;
    if (g.cout_valid % 128 != 0 || (g.Cout & 7) != 0) return false;
    const int hp = (g.Ho + 15) / 16, wp = (g.Wo + 15) / 16;
    const double eff = (double)g.Ho * g.Wo / ((double)hp * 16 * wp * 16);
    // (0.7: at 40 x 56 planes the 16 x 16 patches waste 27 % of their MFMAs and still beat the 8 x 16-patch kernel, 875 vs 650-735 TFLOP/s effective)
    return eff >= 0.7 && (int64_t)g.N * g.Dm * hp * wp >= 256;
}

template <typename T, bool FUSE, int NW, bool P2 = false>
static int launch_fprop_halo256_impl(const FpropArgs& a, uint32_t nbn, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};   // one bit per device (one static per template instance)
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)conv_fprop_halo256_kernel<T, FUSE, NW, P2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });
    hipLaunchKernelGGL((conv_fprop_halo256_kernel<T, FUSE, NW, P2>), dim3(a.nblk_m * nbn), dim3(NW * 64), lds, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

template <typename T, bool FUSE = false>
static int launch_fprop_halo256(FpropArgs a, hipStream_t st) {
    const uint32_t nbn = (uint32_t)a.g.cout_valid / 128;
    if constexpr (sizeof(T) == 2) {
        // two-plane tiles (2 x 8 x 16 voxels, plane slots shared between consecutive kd groups) when they tile the grid at least as well as 16 x 16 patches
        // (SA_PP_DBG bit 10: never, bit 12: wherever they fit)
        const int hp8 = (a.g.Ho + 7) / 8, hp16 = (a.g.Ho + 15) / 16, wp = (a.g.Wo + 15) / 16, dp = (a.g.Dm + 1) / 2;
        const int64_t tiles1 = (int64_t)a.g.Dm * hp16 * wp, tiles2 = (int64_t)dp * hp8 * wp;
        const bool p2 = !(g_tunables.pp_dbg & 1024u) && ((g_tunables.pp_dbg & 4096u) ? true : tiles2 <= tiles1);
        if (p2) {
            a.HP = (uint32_t)hp8;
            a.WP = (uint32_t)wp;
            a.DP = (uint32_t)dp;
            a.nblk_m = (uint32_t)a.g.N * a.DP * a.HP * a.WP;
            const size_t lds = 46 * 1024 + 2 * 128 * 128;   // 78 KiB: two plane slots + two weight slabs (>= the 64 KiB hidden tile of the fused variant)
            (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_halo256_kernel<%s, %s, 8, true>", tname<T>(), FUSE ? "true" : "false"), note_kernel(g_last_conv_kernel));
            return launch_fprop_halo256_impl<T, FUSE, 8, true>(a, nbn, lds, st);
        }
    }
    a.HP = (uint32_t)(a.g.Ho + 15) / 16;
    a.WP = (uint32_t)(a.g.Wo + 15) / 16;
    a.nblk_m = (uint32_t)a.g.N * (uint32_t)a.g.Dm * a.HP * a.WP;
    const size_t lds = 41 * 1024 + 2 * 128 * 128;   // 73 KiB (>= the 64 KiB hidden tile of the fused variant)
    // bf16: eight waves per block (two blocks = four waves per SIMD): fused block 4.76 -> 4.51 ms, data gradient 4.73 -> 4.29 ms on the C = 128 /
    // 80 x 112 x 80 layer.  fp32 keeps four (its 128 accumulators + wider fragments do not fit 128 VGPRs).  SA_DBG_HALO256_4W selects four for A/B runs.
    // (the four-wave bf16 instances of THIS kernel -- 1 MB of device code each -- left the build after the measurement; SA_DBG_HALO256_4W still selects the
    //  four-wave forms of the im2col-order and weight-gradient kernels)
    if constexpr (sizeof(T) == 2) {
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_halo256_kernel<%s, %s, 8, false>", tname<T>(), FUSE ? "true" : "false"), note_kernel(g_last_conv_kernel));
        return launch_fprop_halo256_impl<T, FUSE, 8>(a, nbn, lds, st);
    } else {
        (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_halo256_kernel<%s, %s, 4, false>", tname<T>(), FUSE ? "true" : "false"), note_kernel(g_last_conv_kernel));
        return launch_fprop_halo256_impl<T, FUSE, 4>(a, nbn, lds, st);
    }
}

// cells256: default for the stride-2 family when the register epilogue applies (whole 128-channel tiles, 16-byte rows, no pre-activation copy, a bf16 copy of
// the output only from f16 launches) and 4 x 8 x 8-cell tiles waste little; SA_DBG_NO_CELLS256 (SA_NO_CELLS256=1) restores the im2col-order kernel.
static bool cells256_eligible(const FpropArgs& a, int sz, bool f16) {
    const sa_conv_geom& g = a.g;
    if (dbg(SA_DBG_NO_CELLS256) || dbg(SA_DBG_NO_HALO) || sz != 2 || a.in_bytes == 0 || ((size_t)g.Cin * sz) % 128 != 0) return false;
    if (g.cout_valid % 128 != 0 || (g.Cout & 7) != 0 || a.ep.out_pre || (a.ep.out_lp && !f16)) return false;
    const bool strided = g.KT[0] == 4;
    for (int d = 0; d < 3; ++d) {
        if (strided ? (g.KT[d] != 4 || g.in_mult[d] != 2 || g.tap_step[d] != 1) : (g.KT[d] != 2 || g.in_mult[d] != 1 || (g.tap_step[d] != 1 && g.tap_step[d] != -1))) return false;
    }
    if ((size_t)g.Kpad != (size_t)g.KT[0] * g.KT[1] * g.KT[2] * g.Cin) return false;
    const int dp = (g.Dm + 3) / 4, hp = (g.Hm + 7) / 8, wp = (g.Wm + 7) / 8;
    const double eff = (double)g.Dm * g.Hm * g.Wm / ((double)dp * 4 * hp * 8 * wp * 8);
    return eff >= 0.85 && (int64_t)g.N * dp * hp * wp >= 256;
}

template <typename T>
static int launch_fprop_cells256(FpropArgs a, hipStream_t st) {
    a.DP = (uint32_t)(a.g.Dm + 3) / 4;
    a.HP = (uint32_t)(a.g.Hm + 7) / 8;
    a.WP = (uint32_t)(a.g.Wm + 7) / 8;
    a.nblk_m = (uint32_t)a.g.N * a.DP * a.HP * a.WP;
    const uint32_t nbn = (uint32_t)a.g.cout_valid / 128u;
    const size_t lds = 4 * 11 * 1024 + 2 * 128 * 128;     // 76 KiB: four plane slots + two weight slabs
    static std::atomic<uint64_t> attr_done{0};
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)conv_fprop_cells256_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });
    (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_cells256_kernel<%s>", tname<T>()), note_kernel(g_last_conv_kernel));
    hipLaunchKernelGGL((conv_fprop_cells256_kernel<T>), dim3(a.nblk_m * nbn * (a.ncls > 1u ? a.ncls : 1u)), dim3(512), lds, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int dispatch_fprop(const FpropArgs& a, hipStream_t st) {
    const int cv = a.g.cout_valid;
    if (a.ep.out_pre || a.ep.out_lp) {
        // out_pre exists in the LDS-staged epilogue of the im2col-order kernels only (dense layers); out_lp there and, for f16 operands, in the register
        // epilogue of the halo kernels too (F16IO); whole 4-channel groups
        const bool halo = halo256_eligible(a, (int)sizeof(T)) || halo_eligible(a, (int)sizeof(T));
        if ((halo && (a.ep.out_pre || !std::is_same<T, f16_t>::value)) || (cv & 3) || (a.g.Cout & 3)) return SA_EUNSUPPORTED;
    }
    const bool multi = a.ncls > 1u;      // several classes in one launch: the cell-256 and im2col-order LDS-DMA kernels take them
    if (multi && (sizeof(T) != 2 || a.in_bytes == 0)) return SA_EUNSUPPORTED;
    if (!multi && halo256_eligible(a, (int)sizeof(T))) return launch_fprop_halo256<T>(a, st);
    if (!multi && halo_eligible(a, (int)sizeof(T))) return launch_fprop_halo<T, false>(a, st);
    if constexpr (sizeof(T) == 2) {
        if (cells256_eligible(a, 2, std::is_same<T, f16_t>::value)) return launch_fprop_cells256<T>(a, st);
    }
    // (measured and removed: a 3-stage ring with 8 waves and counted vmcnt = the 2-stage loop; an 8-wave ping-pong with staggered
    // barriers and s_setprio = -7 %: the L2 -> LDS operand stream bounds this loop, not the barrier structure.  DESIGN.md section 4.1)
    if (cv > 64) {
        // Small grids (the transformer's dense layers: 66 row tiles x N/128): 128 x 128 tiles at two blocks per CU leave the last round almost
        // empty (528 blocks on 512 slots = two rounds).  128 x 64 tiles need 48 KiB of LDS -> three blocks per CU (768 slots) and half the
        // work per block: N = 1024 takes ~1 unit instead of 2.  SA_NO_SMALL_TILES=1 keeps the wide tiles.
        const uint64_t blocks128 = (uint64_t)a.nblk_m * (((uint32_t)cv + 127u) / 128u);
        const bool small_ok = !dbg(SA_DBG_NO_SMALL_TILES);
        // bf16 with DMA-addressable operands: eight waves per block (half-size wave tiles, same 128 x 128 / 128 x 64 block tile and LDS): strided
        // 4x4x4 conv forward 1.83 -> 1.73 ms, its 8-parity data gradient 2.64 -> 2.17 ms, transposed conv forward 2.73 -> 2.15 ms at batch 8
        // (tools/microbench.py); SA_DBG_HALO256_4W keeps four waves for A/B runs
        const bool w8 = sizeof(T) == 2 && a.in_bytes != 0 && !dbg(SA_DBG_HALO256_4W);
        // (256 x 128 tiles -- eight waves of 64 x 64 outputs, 96 KiB of LDS: ONE block per CU, 87 FLOP per staged byte instead of 64 / 43 -- were MEASURED SLOWER in
        // round 3: Performer step 40.2 vs 37.7 ms, VQ-VAE step 62.1 vs 63.3 volumes/s; the instance <T, 4, 2, 4, 4> and its SA_TILE256 switch left the build again:
        // 1.1 MB of device code and a minute of compile time.  The kernel template still accepts BM = 256.)
        // Round 3, per shape (tools/bench_dense_tiles.py, M = 8 400): the narrow tile only pays for SHORT reductions over few output columns
        // (K <= 512 and N <= 2 048: 51.9 vs 51.2, 31.0 vs 31.8 us); with K >= 1 024 the wide tile wins by 14-32 % (w2 forward 52.7 -> 39.3 us,
        // q|k|v data gradient 76.3 -> 51.6 us) and q|k|v forward (N = 3 072) by 15 %.  SA_DENSE_NARROW=1 restores the round-2 rule for A/B runs.
        // (a three-stage 128 x 256 DMA ring for the few-wide-tile dense shapes -- csrc/dense_ring.hip, SA_DENSE_RING -- was measured 15 % SLOWER in round 3 and
        //  left the tree in round 4; history: commit 3edf44f)
        // Round 5: with the batched phase B of the LDS-staged epilogue (conv_fprop_common.h) the wide tile wins these shapes too at M = 8 400 (w1 forward 46.3 vs
        // 37.9 us, w2 data gradient 44.5 vs 36.1, to_out data gradient 27.3 vs 24.7): the narrow tile stays for grids that would not fill the CUs with wide ones.
        const bool narrow_shape = (a.nk <= 8 && cv <= 2048 && blocks128 < 2u * (uint64_t)device_cu_count()) || dbg(SA_DBG_DENSE_NARROW);
        if (small_ok && narrow_shape && blocks128 < 2048 && a.in_bytes != 0) return w8 ? launch_fprop<T, 4, 2, 2, 2>(a, st) : launch_fprop<T, 4, 1, 2, 4>(a, st);
        return w8 ? launch_fprop<T, 4, 2, 2, 4>(a, st) : launch_fprop<T, 2, 2, 4, 4>(a, st);
    }
    if (cv > 32) return launch_fprop<T, 4, 1, 2, 4>(a, st);
    if (cv > 16) return launch_fprop<T, 4, 1, 2, 2>(a, st);
    return launch_fprop<T, 4, 1, 2, 1>(a, st);
}


// fused residual block, by operand type (bf16_t, or f16_t for an f16 forward chain): `a` prepared by sa_resblock_fprop
template <typename T>
static int launch_resblock(const FpropArgs& a, hipStream_t st) {
    if (halo256_eligible(a, 2) && !dbg(SA_DBG_NO_HALO256_FUSE)) return launch_fprop_halo256<T, true>(a, st);
    if (halo_eligible(a, 2)) return launch_fprop_halo<T, true>(a, st);
    const size_t pipe = 2 * (128 + 128) * 128, epi = (size_t)128 * (128 + 4) * 4 + 128 * 8;
    (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "conv_fprop_dma_kernel<%s, 2, 2, 4, 4, true, true, 1>", tname<T>()), note_kernel(g_last_conv_kernel));
    hipLaunchKernelGGL((conv_fprop_dma_kernel<T, 2, 2, 4, 4, true, true>), dim3(a.nblk_m), dim3(256), pipe > epi ? pipe : epi, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

// f16 forward operands live in their own translation unit (conv_fprop_f16.hip: the instances compile in parallel with the bf16 / fp32 ones)
int dispatch_fprop_f16(const FpropArgs& a, hipStream_t st);
int launch_resblock_f16(const FpropArgs& a, hipStream_t st);

}  // namespace sa
