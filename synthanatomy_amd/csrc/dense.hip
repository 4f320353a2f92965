// Dense layers (nn.Linear of the Performer: reference src/networks/transformers/performer.py:194-221, the q|k|v / to_out / feed-forward projections of
// performer-pytorch's SelfAttention and FeedForward) and their data gradients as ONE-TAP implicit GEMMs on MFMA (gfx950), round 5.
//
// OPT-IN (SA_DENSE_RING=1 / SA_DBG_DENSE_RING): parity-exact, but slower than the im2col-order kernel on 7 of the 8 shapes of a Performer layer as long as the
// packed weights are row-major; kept with its ablation tooling (tools/dense_ablate.sh) because the measurements behind it re-shaped
// the product path (batched epilogue, 256-voxel cell tiles).
//
// Why a dedicated mainloop.  The im2col-order kernel (conv_fprop_kernels.h: conv_fprop_dma_kernel) serves these layers with eight waves of 32 x 64 outputs, two
// stage buffers and `s_waitcnt vmcnt(0)` + barrier per 64-element K-slab.  Ablated on the 512-column shapes at M = 8 400 (K = 2 048, rocprofv3 durations):
// 38.6 us in full, 35.1 us without the MFMAs / LDS reads, 32.8 us WITHOUT ANY DMA, 17.8 us with neither -- the compute phase alone runs at ~0.5 us per slab for
// 0.21 us of MFMA issue (each wave reads 12 fragments for 16 MFMAs; the LDS port also takes the DMA's writes), and a deeper ring on the same wave layout was
// slower still (50 us).  This kernel changes the wave layout, not just the depth:
//   * FOUR waves per block, one per SIMD, each owning ALL rows of the tile x 32 output channels (MI x 2 accumulator fragments);
//   * the weight fragments of a wave are private to it, so they come global -> VGPR in MFMA operand layout, S slabs ahead, and never touch LDS; only the
//     activation rows (shared by the four waves) go through a ring of S LDS stage buffers filled by LDS-DMA;
//   * both load streams are issued from inline assembly (invisible to the compiler's waitcnt pass, which would drain vmcnt(0) before every LDS read that
//     follows a builtin LDS-DMA) and counted by the kernel: at the top of slab s the wave waits until slab s + 1 has landed, ONE barrier, issues slab
//     s + S - 1 into the slot slab s - 1 left, multiplies slab s; the loop is unrolled by S so that ring slots and weight registers are compile-time indices;
//   * the fragments of the NEXT K step are requested before the MFMAs of the current one (register double buffer, also across the slab boundary);
//   * the tile height is a template parameter (16 MI rows), chosen per launch so that the grid fills whole rounds of two blocks per CU.
// Epilogue: the LDS-staged one of the convolution kernels (bias, activation, ReZero gate, residual, bf16 copies: conv_fprop_common.h); the register form below
// (a.dbg & 256) measured slower.
#include <type_traits>

#include "conv_fprop_common.h"

namespace sa {

__device__ __forceinline__ void dense_dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void dense_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// Register epilogue: a lane holds 4 consecutive output channels (fq) of row frow of every fragment, so bias / addend / mask come in and the result leaves as
// 16-byte (fp32) or 8-byte (16-bit) pieces straight from the accumulators -- the 16 lanes of a quarter cover 16 rows, the four quarters 64 contiguous bytes (fp32)
// of each row, and the wave's second column fragment the other half of the 128-byte line.  Same arithmetic, in the same order, as the LDS-staged epilogue of the
// convolution kernels (conv_fprop_common.h: fprop_epilogue_ov), which took 13.9 of the 39 us of a 512-column launch (park 72 KiB in LDS, barrier, 18 dependent
// LDS-read -> store passes per thread with one block per CU and nothing to overlap with).  Whole tiles of valid channels with 16-byte aligned rows only.
template <int MI, int NI>
__device__ __forceinline__ void dense_epilogue_direct(const FpropArgs& a, float4_t (&acc)[NI][MI], uint32_t wave, uint32_t frow, uint32_t fq, uint32_t m_base, uint32_t n_base) {
    const sa_epilogue& ep = a.ep;
    const sa_conv_geom& g = a.g;
    const float alpha = ep.alpha ? *ep.alpha : 1.f;
    float4_t bias[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bias[i] = ep.bias ? *(const float4_t*)(ep.bias + n_base + wave * 32u + i * 16u + fq * 4u) : (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const uint32_t m = m_base + j * 16u + frow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t o = (int64_t)m * g.Cout + (n_base + wave * 32u + i * 16u + fq * 4u);
            float v[4] = {acc[i][j][0] + bias[i][0], acc[i][j][1] + bias[i][1], acc[i][j][2] + bias[i][2], acc[i][j][3] + bias[i][3]};
            if (ep.out_pre) {   // pre-activation copy (bf16): acc + bias
                uint2 pk;
                pk.x = pack2<bf16_t>(v[0], v[1]);
                pk.y = pack2<bf16_t>(v[2], v[3]);
                *(uint2*)((bf16_t*)ep.out_pre + o) = pk;
            }
            float ad[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
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
                pk.x = pack2<bf16_t>(v[0], v[1]);
                pk.y = pack2<bf16_t>(v[2], v[3]);
                *(uint2*)((bf16_t*)ep.out_lp + o) = pk;
            }
        }
    }
}

template <int OFF> __device__ __forceinline__ u32x4 dense_load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
    u32x4 r;      // (hidden from the waitcnt pass like the LDS-DMA: the kernel's own counted waits cover it; loads retire in order)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
    return r;
}

// Ablation that shaped this loop (tools/dense_ablate.sh, K = 2 048 -> 512 columns, 144-row tiles, first form: BOTH operands through the LDS ring): DMA stream alone
// 0.37 us per slab, MFMAs + fragment reads beside zero-filling DMAs 0.63 us per slab for 0.24 us of MFMA issue -- the LDS port, which takes the DMA's writes
// (34 KiB per slab at ~64 B/clk) AND the 88 KiB of fragment reads.  A wave owns its 32 output channels alone, so the weight fragments need no LDS at all:
// they come global -> VGPR in MFMA operand layout (lane (frow, fq) = 16 bytes of weight row frow at K offset fq * 16), S slabs ahead, and only the activation
// rows (shared by the four waves) go through the ring: 18 instead of 34 KiB of LDS writes and 72 instead of 88 KiB of reads per slab, 72 KiB of LDS per block
// (two blocks per CU).
template <typename T, int MI, int S, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void dense_gemm_kernel(const FpropArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = MI * 16, BN = 128, NI = 2, SZ = sizeof(T);
    constexpr int PA = BM / 8, NA = (PA + 3) / 4, NOPS = NA + 4;     // per wave and slab: 1 KiB activation pieces by LDS-DMA (padded to whole rounds) + 4 weight fragments
    constexpr int STAGE = BM * 128;
    static_assert(SZ == 2 && S >= 4 && NOPS * (S - 1) <= 63, "16-bit operands; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    uint32_t bm = bid % a.nblk_m, bn = bid / a.nblk_m;
    const uint32_t group_m = a.group_m & 0x7fffffffu;     // (bit 31: register epilogue, set by the launcher)
    if (group_m) {      // blocks one XCD runs at a time: group_m row tiles x all channel tiles (their activation panels + the weight panels once per L2)
        const uint32_t nbn = gridDim.x / a.nblk_m, per = group_m * nbn;
        const uint32_t gid = bid / per, first = gid * group_m, r = bid - gid * per;
        const uint32_t gsz = a.nblk_m - first < group_m ? a.nblk_m - first : group_m;
        bm = first + r % gsz;
        bn = r / gsz;
    }
    const uint32_t m_base = bm * BM, n_base = bn * BN;
    const sa_conv_geom& g = a.g;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)a.w_bytes, 0x00020000);
    const uint32_t prow = lane >> 3, lv = (lane & 7u) ^ prow;     // row inside an 8-row piece; SOURCE 16-byte vector (the XOR swizzle is applied on the source side)
    const uint32_t frow = lane & 15u, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const uint32_t dump = lds0 + S * STAGE;                       // 1 KiB that padding loads (zeros) land in
    uint32_t aoff[NA], alds[NA], woff[NI];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const uint32_t piece = (uint32_t)j * 4u + wave;
        const uint32_t m = m_base + piece * 8u + prow;
        const bool ok = piece < (uint32_t)PA && m < a.M;
        aoff[j] = ok ? m * (uint32_t)(g.Cin * SZ) + lv * 16u : OOB_OFF;                    // rows beyond M: zeros (out-of-bounds offset)
        alds[j] = piece < (uint32_t)PA ? lds0 + piece * 1024u : dump;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) woff[i] = (n_base + wave * 32u + (uint32_t)i * 16u + frow) * (uint32_t)(g.Kpad * SZ) + fq * 16u;
#ifdef SA_PP_DEBUG_VARIANTS
    if (a.dbg & 512u)     // TIMING PROBE ONLY (wrong results): the weight fragments read as if the operand were K-blocked ([K / 8][N][8]): 256 contiguous bytes per 16 lanes
#pragma unroll
        for (int i = 0; i < NI; ++i) woff[i] = (n_base + wave * 32u + (uint32_t)i * 16u) * 16u + frow * 16u + fq * (uint32_t)(g.CoutPad * 16);
#endif
    const uint32_t kbytes = (uint32_t)(g.Cin * SZ);               // bytes of an activation row (slabs beyond it multiply zero columns of the packed weights)
    uint32_t nk = a.nk;
#ifdef SA_PP_DEBUG_VARIANTS
    if (a.dbg & 8u) nk = 0;                 // ablation: no main loop (prologue + epilogue only)
#endif
    u32x4 W[S][2][NI];                       // weight fragments of the slabs in flight: [ring slot][K step][column fragment]
    // every wave issues exactly NOPS loads per slot -- also for slots beyond nk and for padding pieces (out-of-bounds offsets: zeros, no memory traffic) --
    // so that "n slots outstanding" is the same vmcnt value in every wave
    auto issue = [&](uint32_t s, auto slot_c) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        bool live = s < nk, a_ok = live && s * 128u < kbytes;
#ifdef SA_PP_DEBUG_VARIANTS
        if (a.dbg & 64u) a_ok = false;      // ablation: no activation traffic
        if (a.dbg & 128u) live = false;     // ablation: no weight traffic
#endif
#pragma unroll
        for (int j = 0; j < NA; ++j)
            dense_dma16(rA, alds[j] == dump ? dump : alds[j] + (uint32_t)(slot * STAGE), a_ok ? aoff[j] : OOB_OFF, a_ok ? s * 128u : 0u);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#ifdef SA_PP_DEBUG_VARIANTS
            if (a.dbg & 512u) {       // (timing probe, see above: slab s = 8 K-groups of CoutPad x 16 bytes, K step 1 four groups further)
                W[slot][0][i] = dense_load16<0>(rB, live ? woff[i] : OOB_OFF, live ? s * (uint32_t)(g.CoutPad * 128) : 0u);
                W[slot][1][i] = dense_load16<0>(rB, live ? woff[i] + (uint32_t)(g.CoutPad * 64) : OOB_OFF, live ? s * (uint32_t)(g.CoutPad * 128) : 0u);
                continue;
            }
#endif
            W[slot][0][i] = dense_load16<0>(rB, live ? woff[i] : OOB_OFF, live ? s * 128u : 0u);
            W[slot][1][i] = dense_load16<64>(rB, live ? woff[i] : OOB_OFF, live ? s * 128u : 0u);
        }
    };
    float4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // fragment addresses: row j*16 + frow, 16-byte vector ks*4 + fq of a [rows][128 B] swizzled tile = lane offset (ks = 0) ^ 64 (ks = 1) + j * 2 KiB
    const uint32_t l0 = frow * 128u + ((fq ^ (frow & 7u)) << 4), l1 = l0 ^ 64u;
    u32x4 xf[2][MI];
    auto load_frags = [&](int ks, const unsigned char* stage, uint32_t lo) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MI; ++j) xf[ks][j] = *(const u32x4*)(stage + lo + j * 2048);
    };
    auto multiply = [&](int ks, auto slot_c, int j0, int j1) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (j >= j0 && j < j1) mma_slab<T>(acc[i][j], W[slot][ks][i], xf[ks][j]);
    };
    // one slab: the wave waits for slab + 1, ONE barrier, K step 1 of this slab is requested, slab + S - 1 is issued into the slot slab - 1 left (activation rows
    // by DMA, weight fragments into the registers that slab - 1 multiplied from), K step 0 is multiplied while K step 0 of the next slab is requested, then K step 1
    auto slab = [&](uint32_t s, auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value, nx = (j + 1) % S, fr = (j + S - 1) % S;
#ifdef SA_PP_DEBUG_VARIANTS
        if (a.dbg & 32u) {                        // ablation: load streams + barriers only
            dense_wait<NOPS * (S - 3)>();
            __builtin_amdgcn_s_barrier();
            issue(s + S - 1, std::integral_constant<int, fr>{});
            return;
        }
#endif
        dense_wait<NOPS * (S - 3)>();             // slab s + 1 (issued S - 2 slots ago) has landed for this wave: its rows in LDS, its weight fragments in registers
        __builtin_amdgcn_s_barrier();             // ... for every wave; every wave is done with the LDS rows of slab s - 1
        // (the waitcnt pass answers a barrier with lgkmcnt(0) in front of the next LDS-fed MFMA: the first MFMAs run on fragments that landed a slab ago, and the
        //  second K step of this slab is requested behind them, not in front)
        issue(s + S - 1, std::integral_constant<int, fr>{});     // (the scheduler may spread these between the MFMAs below)
        multiply(0, j_c, 0, MI / 2);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(1, smem + j * STAGE, l1);      // second K step of this slab, under the remaining MFMAs of the first
        __builtin_amdgcn_sched_barrier(0);
        multiply(0, j_c, MI / 2, MI);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(0, smem + nx * STAGE, l0);     // first K step of the NEXT slab (beyond nk: zeros, multiplied by zero weights or not at all)
        __builtin_amdgcn_sched_barrier(0);
        multiply(1, j_c, 0, MI);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto for_slots = [&](auto&& f) __attribute__((always_inline)) {
        f(std::integral_constant<int, 0>{});
        f(std::integral_constant<int, 1>{});
        f(std::integral_constant<int, 2>{});
        if constexpr (S > 3) f(std::integral_constant<int, 3>{});
        if constexpr (S > 4) f(std::integral_constant<int, 4>{});
        if constexpr (S > 5) f(std::integral_constant<int, 5>{});
    };
    for_slots([&](auto c) __attribute__((always_inline)) {
        if constexpr (decltype(c)::value < S - 1) issue((uint32_t)decltype(c)::value, c);
    });
    // slab 0 has landed everywhere -> its first K step goes into the register buffer
    dense_wait<NOPS * (S - 2)>();
    __builtin_amdgcn_s_barrier();
    load_frags(0, smem, l0);
    // S slabs per trip, so that ring slots and weight registers are compile-time indices; slabs beyond nk multiply zero weights (nk is a multiple of 4 for every
    // layer of the Performer: no wasted trip)
#pragma clang loop unroll(disable)
    for (uint32_t base = 0; base < nk; base += S)
        for_slots([&](auto c) __attribute__((always_inline)) { slab(base + (uint32_t)decltype(c)::value, c); });
#ifdef SA_PP_DEBUG_VARIANTS
    if (a.dbg & 16u) { dense_wait<0>(); return; }      // ablation: no epilogue
#endif
    dense_wait<0>();         // (padding loads of the last slots still write zeros into stage buffers: retired before the block ends / the epilogue reuses the LDS)
    if (a.group_m >> 31) {   // (flag set by the launcher) whole tile of valid channels, 16-byte aligned identity-mapped rows: straight from the registers
        dense_epilogue_direct<MI, NI>(a, acc, wave, frow, fq, m_base, n_base);
        return;
    }
    __syncthreads();
    fprop_epilogue<BM, BN, 1, 4, MI, NI, 256>(a, acc, smem, tid, 0u, wave, frow, fq, m_base, n_base);
#endif
}

template <typename T, int MI, int S, int OCC>
static int launch_dense_instance(FpropArgs a, hipStream_t st) {
    constexpr int BM = MI * 16;
    a.nblk_m = (a.M + BM - 1) / BM;
    const uint32_t nbn = ((uint32_t)a.g.cout_valid + 127u) / 128u;
    // blocks of an XCD at a time = 8 row tiles x all channel tiles: an activation panel is fetched once per L2 instead of once per channel tile (row tiles fastest
    // puts the channel tiles of one row tile on different XCDs: 4 x 34 MB from the fabric for the 512-column layers; the load streams alone 0.57 -> 0.37 us per
    // slab).  SA_PP_DBG bits 16-23 override (255 = off).
    a.group_m = (g_tunables.pp_dbg >> 16) & 255u ? ((g_tunables.pp_dbg >> 16) & 255u) % 255u : (nbn >= 2 ? 8u : 0u);
    const sa_conv_geom& g = a.g;
    bool direct = (a.dbg & 256u) != 0 && (uint32_t)g.cout_valid % 128u == 0 && (g.Cout & 3) == 0 && g.Do == g.Dm && g.Ho == g.Hm && g.Wo == g.Wm;   // (opt-in: measured slower)
    for (int d = 0; d < 3; ++d) direct = direct && g.out_mult[d] == 1 && g.out_off[d] == 0;
    if (direct) a.group_m |= 0x80000000u;
    const size_t ring = (size_t)S * BM * 128 + 1024, epi = (size_t)BM * (128 + 4) * 4 + BM * 8;
    static std::atomic<uint64_t> attr_done{0};
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)dense_gemm_kernel<T, MI, S, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    (snprintf(g_last_conv_kernel, sizeof g_last_conv_kernel, "dense_gemm_kernel<%s, %d, %d, %d>", tname<T>(), MI, S, OCC), note_kernel(g_last_conv_kernel));
    hipLaunchKernelGGL((dense_gemm_kernel<T, MI, S, OCC>), dim3(a.nblk_m * nbn), dim3(256), ring > epi ? ring : epi, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

static inline int dense_cu_count() {
    static std::atomic<int> cus{0};
    int c = cus.load(std::memory_order_relaxed);
    if (c <= 0) {
        int dev = 0;
        c = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus.store(c, std::memory_order_relaxed);
    }
    return c;
}

// Tile height per launch: the instantiated heights are 16 * {6, 8, 9, 10}; two blocks per CU (72-80 KiB of LDS, <= 256 VGPRs), so a launch costs
// rounds(tiles / (2 CUs)) x (tile height + a fixed per-slab share).  SA_PP_DBG bits 24-27 force a height index + 1 for A/B runs.
template <typename T>
static int launch_dense_t(const FpropArgs& a, hipStream_t st) {
    static const int heights[4] = {6, 8, 9, 10};
    const uint32_t nbn = ((uint32_t)a.g.cout_valid + 127u) / 128u;
    const int slots = 2 * dense_cu_count();
    int best = 0;
    double best_cost = 1e30;
    for (int h = 0; h < 4; ++h) {
        const uint64_t tiles = (uint64_t)((a.M + heights[h] * 16 - 1) / (heights[h] * 16)) * nbn;
        const double rounds = (double)((tiles + slots - 1) / slots);
        const double cost = rounds * (heights[h] * 16 + 32.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = h; }
    }
    const uint32_t force = (g_tunables.pp_dbg >> 24) & 15u;
    if (force >= 1 && force <= 4) best = (int)force - 1;
    switch (best) {
        case 0: return launch_dense_instance<T, 6, 4, 2>(a, st);
        case 1: return launch_dense_instance<T, 8, 4, 2>(a, st);
        case 2: return launch_dense_instance<T, 9, 4, 2>(a, st);
        default: return launch_dense_instance<T, 10, 4, 2>(a, st);
    }
}

// One-tap, stride-1, 16-bit, DMA-addressable, whole 128-byte K-slabs, identity row map: GEMM row m reads input voxel m.
bool dense_gemm_eligible(const FpropArgs& a, int sz) {
    const sa_conv_geom& g = a.g;
    if (!dbg(SA_DBG_DENSE_RING) || sz != 2 || a.ntaps != 1 || a.in_bytes == 0 || a.nk < 4 || ((size_t)g.Cin * sz) % 128 != 0 || g.cout_valid <= 64) return false;
    if (g.Di != g.Dm || g.Hi != g.Hm || g.Wi != g.Wm) return false;
    for (int d = 0; d < 3; ++d)
        if (g.in_mult[d] != 1 || g.in_off[d] != 0) return false;
    return true;
}

int launch_dense_gemm(const FpropArgs& a, int dtype, hipStream_t st) { return dtype == SA_F16 ? launch_dense_t<f16_t>(a, st) : launch_dense_t<bf16_t>(a, st); }

}  // namespace sa
