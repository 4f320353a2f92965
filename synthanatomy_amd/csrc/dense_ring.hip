// Dense (1x1x1) layers whose output is only a few channel tiles wide -- the 512-column layers of the Performer (to_out, w2, the data gradients of w1
// and q|k|v: M = batch * N = 8 400 rows, N = 512, K = 1 024 .. 3 072).  On 128 x 128 tiles such a launch is 264 blocks = ONE block per CU, and with one block
// per CU the double-buffered im2col-order loop (conv_fprop_dma_kernel) waits a full L2 / fabric round trip per K-slab: 0.85 us per 64-deep slab against
// 0.21 us of MFMA work (tools/bench_dense_tiles.py: 40 us for K = 2 048 where hipBLASLt needs 23.5).  This mainloop trades blocks for depth:
//   * 128 x 256 tiles (eight waves of 32 x 128: 16 MFMAs per 10 LDS fragment reads, against 8 per 6) -> 132 blocks, one per CU on half the chip;
//   * a THREE-stage LDS-DMA ring (3 x 48 KiB): two K-slabs are in flight while one is multiplied.  The queue is never drained inside the loop:
//     `s_waitcnt vmcnt(P)` (P = this wave's pieces per slab) retires exactly the oldest slab, one raw s_barrier per slab publishes it and frees the
//     buffer the slab after next lands in.  (hipcc adds no vmcnt(0) of its own in front of the ds_reads -- checked in the ISA -- but a
//     __syncthreads() would.)
// MEASURED (round 3, one MI355X, rocprofv3 kernel durations over the four 512-column shapes): 45.0 us mean against 38.9 us for the two-stage kernel
// (hipBLASLt 16-21 us), so this is an opt-in A/B instance (SA_DENSE_RING=1), not the product path.  Ablation with the SA_PP_DBG bits below
// (K = 2 048, us per launch): everything 45.9 | no DMA after the first two slabs 39.7 | no MFMA / LDS reads 30.4 | neither 19.7 | neither and no
// epilogue 12 (the host loop's floor).  I.e. the DMA stream is NOT the bound (198 MB in ~11 us = 18 TB/s from L2): the 128 x 256 tile halves the
// CUs that multiply (20 us of MFMA + fragment reads on 132 CUs), and the LDS-staged epilogue + the write-back of the 17 MB fp32 output cost 8-13 us
// per launch -- as much as half of hipBLASLt's whole kernel.  What this says about the product kernel: the next step for these shapes is the
// epilogue (register form for MI = 2 tiles) and 256 busy CUs, not a deeper ring.
// Same operands, packed weights, epilogue (fprop_epilogue: bias / gate / addend / GELU / mask / pre-activation and bf16 copies) and results as the
// im2col-order kernel: the K order is unchanged, so outputs are bit-identical.  bf16 only.
#include "conv_fprop_common.h"

namespace sa {

constexpr int DR_BM = 128, DR_BN = 256, DR_WM = 4, DR_WN = 2, DR_MI = 2, DR_NI = 8, DR_NW = 8, DR_STAGES = 3;
constexpr int DR_STAGE_BYTES = (DR_BM + DR_BN) * 128;              // 48 KiB: activation rows, then weight rows, 128 bytes (64 bf16) each
constexpr int DR_A_PER_WAVE = (DR_BM / 8) / DR_NW;                 // 1 KiB pieces (8 rows x 128 B) per wave and slab: 2 + 4
constexpr int DR_B_PER_WAVE = (DR_BN / 8) / DR_NW;
constexpr int DR_P = DR_A_PER_WAVE + DR_B_PER_WAVE;

__global__ __launch_bounds__(DR_NW * 64) void dense_ring_kernel(const FpropArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave / DR_WN, wn = wave % DR_WN;
    const uint32_t nbn = gridDim.x / a.nblk_m;
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bm = bid / nbn, bn = bid - bm * nbn;            // channel tiles fastest: the blocks that share an activation panel sit on one XCD
    const uint32_t m_base = bm * DR_BM, n_base = bn * DR_BN;
    const sa_conv_geom& g = a.g;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);
    const uint32_t prow = lane >> 3;                       // row within an 8-row piece
    const uint32_t lv = (lane & 7u) ^ prow;                // SOURCE 16-byte vector: the XOR swizzle is applied on the source side (the DMA writes lane-linearly)
    const uint32_t row_bytes = (uint32_t)g.Cin * 2u;
    uint32_t aoff[DR_A_PER_WAVE], boff[DR_B_PER_WAVE];
#pragma unroll
    for (int j = 0; j < DR_A_PER_WAVE; ++j) {
        const uint32_t m = m_base + (wave * DR_A_PER_WAVE + j) * 8 + prow;
        aoff[j] = m < a.M ? m * row_bytes + lv * 16u : OOB_OFF;    // rows beyond M: zeros
    }
#pragma unroll
    for (int j = 0; j < DR_B_PER_WAVE; ++j) boff[j] = (n_base + (wave * DR_B_PER_WAVE + j) * 8 + prow) * (uint32_t)(g.Kpad * 2) + lv * 16u;

    auto issue = [&](uint32_t s, uint32_t buf) __attribute__((always_inline)) {
        unsigned char* pa = smem + buf * DR_STAGE_BYTES;
        unsigned char* pb = pa + DR_BM * 128;
#pragma unroll
        for (int j = 0; j < DR_A_PER_WAVE; ++j)   // (the slab offset goes into the VGPR offset: the bounds check of a raw buffer load does not see soffset)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(pa + (wave * DR_A_PER_WAVE + j) * 1024), 16,
                                                     aoff[j] == OOB_OFF ? OOB_OFF : aoff[j] + s * 128u, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DR_B_PER_WAVE; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(pb + (wave * DR_B_PER_WAVE + j) * 1024), 16, boff[j], s * 128u, 0, 0);
    };

    float4_t acc[DR_NI][DR_MI];
#pragma unroll
    for (int i = 0; i < DR_NI; ++i)
#pragma unroll
        for (int j = 0; j < DR_MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const uint32_t nk = a.nk;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    uint32_t buf = 0, nxt = 2;                             // slab s is multiplied from `buf`; slab s + 2 lands in `nxt`
    for (uint32_t s = 0; s < nk; ++s) {
        // this wave's pieces of slab s have landed (slab s + 1 may stay in flight); the barrier makes that true for every wave and also says
        // that everybody is done reading slab s - 1, whose buffer (`nxt`) the next issue overwrites
        if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DR_P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nk && !(a.dbg & 64u)) issue(s + 2, nxt);                                                    // (SA_PP_DBG ablation bits: 64 = no DMA after the
        if (a.dbg & 128u) { buf = buf == 2 ? 0 : buf + 1; nxt = nxt == 2 ? 0 : nxt + 1; continue; }             //  prologue, 128 = no products, 32 = no epilogue)
        const unsigned char* pa = smem + buf * DR_STAGE_BYTES;
        const unsigned char* pb = pa + DR_BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 xf[DR_MI], wf[DR_NI];
#pragma unroll
            for (int j = 0; j < DR_MI; ++j) xf[j] = *(const u32x4*)(pa + tile_off(wm * (DR_MI * 16) + j * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < DR_NI; ++i) wf[i] = *(const u32x4*)(pb + tile_off(wn * (DR_NI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < DR_NI; ++i)
#pragma unroll
                for (int j = 0; j < DR_MI; ++j) mma_slab<bf16_t>(acc[i][j], wf[i], xf[j]);
        }
        buf = buf == 2 ? 0 : buf + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    __syncthreads();   // the epilogue stages its tile in the ring's LDS
    if ((a.dbg & 32u) && acc[0][0][0] != 123.456f) return;
    fprop_epilogue<DR_BM, DR_BN, DR_WM, DR_WN, DR_MI, DR_NI, DR_NW * 64>(a, acc, smem, tid, wm, wn, frow, fq, m_base, n_base);
#endif
}

// Qualifies: bf16, 1x1x1 stride 1 (a dense layer over M rows), every output channel valid in whole 256-wide tiles, K in whole 64-element slabs with the
// packed weights unpadded along K, operands addressable with 32-bit offsets, and a launch that would NOT fill the chip with two 128 x 128 blocks per CU.
int launch_dense_ring(const FpropArgs& a, hipStream_t st) {
    const sa_conv_geom& g = a.g;
    if (a.ntaps != 1 || a.in_bytes == 0 || a.w2pk) return SA_EUNSUPPORTED;
    for (int d = 0; d < 3; ++d)
        if (g.KT[d] != 1 || g.in_mult[d] != 1 || g.in_off[d] != 0) return SA_EUNSUPPORTED;
    if (g.Cin % 64 != 0 || g.Kpad != g.Cin || g.cout_valid % DR_BN != 0 || g.cout_valid > g.CoutPad || a.nk < 8) return SA_EUNSUPPORTED;
    if ((uint64_t)a.M * (uint64_t)g.Cin * 2u >= 0xfffffff0ull - 4096u) return SA_EUNSUPPORTED;
    const uint32_t nbm = (a.M + DR_BM - 1) / DR_BM, nbn = (uint32_t)g.cout_valid / DR_BN;
    if ((uint64_t)nbm * ((uint32_t)g.cout_valid / 128u) > 400u) return SA_EUNSUPPORTED;    // >= ~1.5 blocks of 128 x 128 per CU: the two-stage loop hides the round trip itself
    FpropArgs b = a;
    b.nblk_m = nbm;
    constexpr size_t lds = (size_t)DR_STAGES * DR_STAGE_BYTES;                              // 144 KiB >= the epilogue's 128 x 260 fp32 tile + row table
    static_assert(lds >= (size_t)DR_BM * (DR_BN + 4) * 4 + DR_BM * 8, "epilogue tile");
    static std::atomic<uint64_t> attr_done{0};
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)dense_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "dense_ring_kernel"), note_kernel(g_last_conv_kernel));
    hipLaunchKernelGGL(dense_ring_kernel, dim3(nbm * nbn), dim3(DR_NW * 64), lds, st, b);
    SA_CHECK_LAUNCH();
    return 0;
}

}  // namespace sa
