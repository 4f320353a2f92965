// FAVOR+ causal attention with the random-feature maps RECOMPUTED ON CHIP (throughput mode of the Performer, SURVEY 2.1 K7/K8: "fuse feature map so HBM
// reads only q, k, v [N, 64]").
//
// The unfused chain (performer.hip / favor_proj.hip) materialises dd = x P^T and phi(x) for queries and keys as fp32 [B N 8, 272] tensors -- 73 MB
// each per layer at the README shape -- and walks them ~10 times per training step.  Here a chunk block (64 positions of one (batch, head), the same
// decomposition as the chunked scans: chunk state sums -> exclusive prefix over chunks -> chunk outputs) rebuilds the 64-feature slab it is
// about to use from the chunk's q / k rows (64 floats per position) and a 64 x 64 slab of the projection matrix:
//     dd^T = P_slab x^T  (split-bf16 MFMA, accumulators: 4 features x 1 position per lane)  ->  phi = ratio (exp(dd - rowoff) + eps)
// where rowoff = |x|^2 c^2 / 2 + stabiliser comes from ONE pre-pass over q and k (row maxima of the queries, the global maximum of the keys:
// performer_pytorch.softmax_kernel), 12 bytes per head row instead of 2 x 1 088.  Features that feed a reduction over positions go through an LDS
// tile (hi / lo bf16, the layout of split_bf16.h), features of a wave's own positions go from the accumulators straight into the next MFMA's B
// operand.  The backward kernels apply the feature-map backward and the projection adjoint to the gradient slab while it is still in registers:
//     v = (phi - ratio eps) dphi,  t = sum_f v_f,  dx = sum_f v_f P[f] - [query] t P[argmax] - t c^2 x      (keys: -(sum of all t) P[f*] on the global-max row)
// so d loss / d phi, d loss / d dd never exist in memory either.  Arithmetic = the unfused throughput path (split-bf16 products, fp32 accumulation).
#include <algorithm>

#include "sa_common.h"
#include "split_bf16.h"
#include "local_attn_split.h"

namespace sa {

#ifdef SA_TIMING_FAVOR
// dev instrumentation (-DSA_TIMING_FAVOR): s_memtime sums of wave 0 of every block, read back by sa_debug_timing_favor
// scan B body: [0] prologue, [1] feature maps, [2] wait + tile writes, [3] barrier 2, [4] two GEMMs, [5] VALU + operand split, [6] dx GEMM, [7] epilogue, [8] blocks;
// scan A body: [10] prologue, [11] slab loop, [12] epilogue, [13] blocks;   state body: [15] prologue, [16] slab loop, [17] blocks
__device__ unsigned long long g_ftime[24];
#define FT_T(var) const unsigned long long var = __builtin_readcyclecounter()
// (phase sums are kept in registers and added to g_ftime ONCE per block.  Until round 5 every phase boundary issued its own atomic: it sat in the vector-memory
//  queue in front of the next s_waitcnt vmcnt and was billed to whatever phase waited next -- the "tile writes + wait" and "VALU + split" figures of
//  profiles/r04_favor_phase_timing.txt were mostly that, and the probe ran at 182 k tokens/s instead of 269 k.)
#define FT_DECL unsigned long long ft_sum[24] = {}
#define FT_ACC(i, v) (ft_sum[i] += (unsigned long long)(v))
#define FT_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 24; ++i_) if (ft_sum[i_]) atomicAdd(&g_ftime[i_], ft_sum[i_]); } while (0)
#else
#define FT_T(var)
#define FT_DECL
#define FT_ACC(i, v)
#define FT_FLUSH()
#endif
constexpr int FT_BYTES = 64 * 128;   // one [64 rows][64 bf16] tile
constexpr int FSLAB = 64;
// The last 16 bytes of the 80 KiB projection-tile buffer (row 63 of the fifth slab's lo tile = projection row 319: never an operand row, LDF <= 272) carry a
// flag word written by favor_proj_tiles_kernel: 0 = every lo half-word of the matrix is zero, i.e. the operand is bf16-representable (what the throughput mode's
// layer hands over: its projection operand is a bf16 copy of the folded fp32 matrix, like every dense weight of that mode).  The chunk kernels then skip the
// P_lo * x_hi products, the lo fragment reads and the lo half of every slab transfer -- exact zeros, so the results do not change by a bit (round 6).
constexpr int PT_FLAG_OFF = 5 * 2 * FT_BYTES - 16;
__device__ __forceinline__ bool ptiles_lo_zero(const unsigned char* ptiles) {
    return __builtin_amdgcn_readfirstlane(*(const uint32_t*)(ptiles + PT_FLAG_OFF)) == 0u;
}
#ifndef FUSED_WPS
#define FUSED_WPS 2   // waves per SIMD the chunk kernels are compiled for (measured: 3 forces ~100 spilled VGPRs and is 20 % slower end to end)
#endif

struct FeatSrc {
    const float* x;        // rows of `stride` floats, head g at column g * 64
    const float* rowoff;   // [B * N * G]: |x|^2 c^2 / 2 (+ the row maximum of dd for queries)
    int32_t use_kmax;      // keys: the global maximum of dd (gmax) is subtracted as well
    int32_t pad;
};

struct FusedArgs {
    FeatSrc fa;            // feature map summed over positions j (the "a" operand of the scans)
    FeatSrc fx;            // out A: the per-position ("c") feature map;  out B: the map whose input gradient is produced
    const unsigned char* ptiles;        // [5 slabs][hi 8 KiB | lo 8 KiB]: projection rows as split-bf16 tiles (lroff layout), rows >= m zero
    const float* ps;                    // projection matrix [m][64] (data normaliser folded in), fp32: stabiliser rows
    const unsigned long long* gmax;     // packed (value, flat index) maximum of the keys' dd
    const int32_t* amx;                 // out B, queries: argmax feature of every head row
    const float* b;        // "value" operand rows [B * N][b_stride], head g at column g * 64
    const float* b_scale;  // optional [B * N * G]
    const float* c;        // out B: per-position operand rows [B * N][c_stride]
    const float* c_scale;  // optional [B * N * G]
    const float* ex_scale; // zmode 1 (out B): per-position i factor;  zmode 2: per-position j weights of the running sums
    float* y;              // out A: rows [B * N][y_stride]
    float* dx;             // out B: rows [B * N][stride]
    unsigned short *y_lp, *dx_lp;   // optional bf16 copies of what is written to y / dx (same element strides): the operand of the next dense layer
    float* state;          // [B, G, S][LDF * 64 + LDF]
    float* inv_out;        // out A, zmode 1
    float* tsum;           // out B, keys: per-block partial sums of t [B * G * S]
    int32_t B, N, G, m, LDF, S, stride, b_stride, c_stride, y_stride, reverse, zmode, is_query, accumulate;
    float ratio, reps, c2, den_eps, ex_const;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t f_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float f_ld1(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ u32x4 f_ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ int f_row(const FusedArgs& s, int p) { return s.reverse ? s.N - 1 - p : p; }   // may leave [0, N): buffer loads return zeros

// the x row of one position as the B operand of tile_rows_gemm (natural order d = ks*32 + g4*8 + e) + its feature offset
struct XOperand {
    short8_t h[2], l[2];
    float off;
};
__device__ __forceinline__ void load_x_operand(XOperand& xo, const FeatSrc& f, const FusedArgs& s, int b, int g, int ri, int g4, float kmax) {
    const __amdgpu_buffer_rsrc_t rx = f_rsrc(f.x + (int64_t)b * s.N * s.stride, (int64_t)s.N * s.stride * 4);
    const __amdgpu_buffer_rsrc_t ro = f_rsrc(f.rowoff + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const u32x4 v0 = f_ld4(rx, (uint32_t)(ri * s.stride + g * 64 + ks * 32 + g4 * 8) * 4u), v1 = f_ld4(rx, (uint32_t)(ri * s.stride + g * 64 + ks * 32 + g4 * 8 + 4) * 4u);
        const float xs[8] = {__uint_as_float(v0[0]), __uint_as_float(v0[1]), __uint_as_float(v0[2]), __uint_as_float(v0[3]),
                             __uint_as_float(v1[0]), __uint_as_float(v1[1]), __uint_as_float(v1[2]), __uint_as_float(v1[3])};
        split8(xs, xo.h[ks], xo.l[ks]);
    }
    xo.off = f_ld1(ro, (uint32_t)(ri * s.G + g) * 4u) + (f.use_kmax ? kmax : 0.f);
}

// features [slab0, slab0 + 64) of this lane's position (column lane & 15 of the wave): F[f][r] = feature slab0 + f*16 + g4*4 + r
// nf = 16-feature fragments of the slab that hold features at all (the last slab of m = 266 has ONE: its other three fragments stay zero and cost neither
// MFMAs nor exponentials; block-uniform)
__device__ __forceinline__ int slab_frags(const FusedArgs& s, int slab0) { return min(4, (s.LDF - slab0 + 15) >> 4); }

__device__ __forceinline__ void feat_slab(float4_t (&F)[4], const unsigned char* sPh, const unsigned char* sPl, const XOperand& xo, bool valid, int slab0,
                                          const FusedArgs& s, int fr, int g4, bool plo0 = false) {
    const int nf = slab_frags(s, slab0);
#pragma unroll
    for (int f = 0; f < 4; ++f) F[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    tile_rows_gemm(F, sPh, sPl, xo.h, xo.l, fr, g4, nf, plo0);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        if (f >= nf) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mi = slab0 + f * 16 + g4 * 4 + r;
            const float e = fmaf(s.ratio, __expf(F[f][r] - xo.off), s.reps);
            F[f][r] = (valid && mi < s.m) ? e : 0.f;
        }
    }
}

// two feature maps of the same positions from ONE walk over the projection fragments (half the ds_read_b128 traffic of two feat_slab calls)
__device__ __forceinline__ void feat_slab2(float4_t (&F0)[4], float4_t (&F1)[4], const unsigned char* sPh, const unsigned char* sPl, const XOperand& x0, const XOperand& x1,
                                           bool valid, int slab0, const FusedArgs& s, int fr, int g4, bool plo0 = false) {
    const int nf = slab_frags(s, slab0);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        F0[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        F1[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    // per accumulator the product order of tile_rows_gemm (hi*lo, lo*hi, hi*hi over ks = 0, 1): bit-identical dd to the pre-pass
    auto frag = [&](const int f, const int ks, short8_t& ah, short8_t& al) __attribute__((always_inline)) {
        const uint32_t o = lroff(f * 16 + fr, ks * 32 + g4 * 8);
        ah = *(const short8_t*)(sPh + o);
        if (!plo0) al = *(const short8_t*)(sPl + o);
    };
    if (nf == 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            short8_t ah[4], al[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) frag(f, ks, ah[f], al[f]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], x0.l[ks], F0[f], 0, 0, 0);
                F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], x1.l[ks], F1[f], 0, 0, 0);
            }
            if (!plo0) {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], x0.h[ks], F0[f], 0, 0, 0);
                    F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], x1.h[ks], F1[f], 0, 0, 0);
                }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], x0.h[ks], F0[f], 0, 0, 0);
                F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], x1.h[ks], F1[f], 0, 0, 0);
            }
        }
    } else {   // partial (last) slab: fragment by fragment, the same per-accumulator order
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (f >= nf) break;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                short8_t ah, al;
                frag(f, ks, ah, al);
                F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x0.l[ks], F0[f], 0, 0, 0);
                F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x1.l[ks], F1[f], 0, 0, 0);
                if (!plo0) {
                    F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, x0.h[ks], F0[f], 0, 0, 0);
                    F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, x1.h[ks], F1[f], 0, 0, 0);
                }
                F0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x0.h[ks], F0[f], 0, 0, 0);
                F1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x1.h[ks], F1[f], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        if (f >= nf) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mi = slab0 + f * 16 + g4 * 4 + r;
            const bool in = valid && mi < s.m;
            const float e0 = fmaf(s.ratio, __expf(F0[f][r] - x0.off), s.reps), e1 = fmaf(s.ratio, __expf(F1[f][r] - x1.off), s.reps);
            F0[f][r] = in ? e0 : 0.f;
            F1[f][r] = in ? e1 : 0.f;
        }
    }
}

// this wave's 16 positions x 64 features -> rows w*16 + fr of a [64 positions][64 features] hi / lo tile
__device__ __forceinline__ void feat_to_tile(unsigned char* hi, unsigned char* lo, const float4_t (&F)[4], int w, int fr, int g4) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const uint32_t o = lroff(w * 16 + fr, f * 16 + g4 * 4);
        uint2 h, l;
        split_pair(F[f][0], F[f][1], h.x, l.x);
        split_pair(F[f][2], F[f][3], h.y, l.y);
        *(uint2*)(hi + o) = h;
        *(uint2*)(lo + o) = l;
    }
}

// one projection slab (hi + lo tiles = 16 KiB, already split and swizzled in global memory) a slab ahead in registers
struct PSlabRegs {
    u32x4 v[4];
};
__device__ __forceinline__ void pslab_load(PSlabRegs& r, const unsigned char* ptiles, int slab, int tid) {
    const u32x4* src = (const u32x4*)(ptiles + (size_t)slab * (2 * FT_BYTES));
#pragma unroll
    for (int t = 0; t < 4; ++t) r.v[t] = src[tid + 256 * t];
}
__device__ __forceinline__ void pslab_store(unsigned char* sP, const PSlabRegs& r, int tid) {
#pragma unroll
    for (int t = 0; t < 4; ++t) ((u32x4*)sP)[tid + 256 * t] = r.v[t];
}

// the same 16 KiB straight into LDS (LDS-DMA, sixteen 1 KiB pieces: wave w takes pieces w, w + 4, ...), issued from inline assembly: invisible to the
// compiler's waitcnt pass (which answers a builtin LDS-DMA with s_waitcnt vmcnt(0) in front of the next LDS read) -- the caller waits (pslab_dma_wait) and
// synchronises before the slab is read.  No staging registers.
__device__ __forceinline__ void pslab_dma(unsigned char* sP, const unsigned char* ptiles, int slab, int wave, int lane, bool plo0 = false) {
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(ptiles + (size_t)slab * (2 * FT_BYTES)), 0, 2 * FT_BYTES, 0x00020000);
    const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sP;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= 2 && plo0) break;      // pieces 8 .. 15 are the lo tile: not read when the operand is bf16-representable
        const uint32_t piece = (uint32_t)(wave + 4 * t);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     : : "s"(l0 + piece * 1024u), "v"(piece * 1024u + (uint32_t)lane * 16u), "s"(rp) : "memory", "m0");
    }
}
__device__ __forceinline__ void pslab_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }

// rows [slab0, slab0 + 64) of the chunk's exclusive-prefix state (64 value columns); zeros outside the matrix
__device__ __forceinline__ void tslab_load(u32x4 (&t)[4], __amdgpu_buffer_rsrc_t rt, int slab0, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) t[it] = f_ld4(rt, (uint32_t)((slab0 + (tid >> 4) + 16 * it) * 64 + (tid & 15) * 4) * 4u);
}

// value rows (head block g) of the 64 positions of a chunk, times an optional per-position scale -> swizzled hi / lo tiles
__device__ __forceinline__ void f_stage_values(unsigned char* hi, unsigned char* lo, const float* base, int stride, const float* scale, const FusedArgs& s, int b,
                                               int g, int chunk, int tid) {
    const __amdgpu_buffer_rsrc_t rv = f_rsrc(base + (int64_t)b * s.N * stride, (int64_t)s.N * stride * 4);
    const __amdgpu_buffer_rsrc_t rsc = f_rsrc((scale ? scale : base) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    u32x4 v[4];
    float sc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = (tid >> 4) + 16 * it;
        const int i = f_row(s, chunk * 64 + j);
        v[it] = f_ld4(rv, (uint32_t)(i * stride + g * 64 + (tid & 15) * 4) * 4u);
        sc[it] = scale ? f_ld1(rsc, (uint32_t)(i * s.G + g) * 4u) : 1.f;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t o = lroff((tid >> 4) + 16 * it, (tid & 15) * 4);
        uint2 h, l;
        split_pair(__uint_as_float(v[it][0]) * sc[it], __uint_as_float(v[it][1]) * sc[it], h.x, l.x);
        split_pair(__uint_as_float(v[it][2]) * sc[it], __uint_as_float(v[it][3]) * sc[it], h.y, l.y);
        *(uint2*)(hi + o) = h;
        *(uint2*)(lo + o) = l;
    }
}

__device__ __forceinline__ short8_t f_tr_operand(const unsigned char* t, uint32_t o0, uint32_t o1) {
    return __builtin_shufflevector(lds_tr16_b64(t + o0), lds_tr16_b64(t + o1), 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------ projection tiles
// ps [m][64] fp32 -> 5 slabs of (hi tile | lo tile), tile row rho of slab sl = projection row sl*64 + rho (zero beyond m)
__global__ __launch_bounds__(256) void favor_proj_tiles_kernel(const float* __restrict__ ps, int m, unsigned char* __restrict__ tiles) {
    const int sl = blockIdx.x, tid = threadIdx.x;
    unsigned char* hi = tiles + (size_t)sl * (2 * FT_BYTES);
    unsigned char* lo = hi + FT_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rho = (tid >> 4) + 16 * it, c4 = tid & 15, f = sl * 64 + rho;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < m) v = *(const float4*)(ps + (int64_t)f * 64 + c4 * 4);
        uint2 h, l;
        split_pair(v.x, v.y, h.x, l.x);
        split_pair(v.z, v.w, h.y, l.y);
        const uint32_t o = lroff(rho, c4 * 4);
        *(uint2*)(hi + o) = h;
        if (!(sl == 4 && o >= (uint32_t)(FT_BYTES - 16))) *(uint2*)(lo + o) = l;      // (the flag's sixteen bytes: zeroed by the launcher, see PT_FLAG_OFF)
        if ((l.x | l.y) & 0x7fff7fffu) atomicOr((unsigned int*)(tiles + PT_FLAG_OFF), 1u);
    }
}

// ------------------------------------------------------------------------------------------------ pre-pass
// Head rows of q (blocks [0, nbq)) and k (blocks [nbq, 2 nbq)): dd = x P^T in accumulators only.
//   queries: rowoff = |x|^2 c^2/2 + max_f dd, amx = argmax_f dd (lowest index on ties);   keys: rowoff = |x|^2 c^2/2, gmax = packed global maximum
struct PrepassArgs {
    const float *q, *k;
    const unsigned char* ptiles;   // the projection matrix as split slab tiles (sa_favor_fused_proj_tiles)
    float *offq, *offk;
    int32_t* amq;
    unsigned long long* gmax;
    int64_t rows;           // B * N * G head rows
    int32_t m, LDF, stride, heads, nbq;
    float c2half;
};
__device__ __forceinline__ int64_t head_row_off(int64_t r, int heads, int stride) { return (r / heads) * stride + (r % heads) * 64; }

__global__ __launch_bounds__(256, 2) void favor_prepass_kernel(const PrepassArgs a) {
    // PERSISTENT blocks (two per CU: the projection matrix, 80 KiB of pre-split slab tiles, is staged once per block) walk tiles of 64 head rows
    // (wave = 16 of them) of q (tiles [0, ntq)) and k (tiles [ntq, 2 ntq)): fine-grained tiles keep the last round of the walk short
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // exactly 80 KiB and NO static LDS beside it: two blocks fill the CU's 160 KiB (32 static bytes
                                                                           // for the block maximum made it 81 952 B = ONE block per CU and two rounds of the "persistent" grid)
    const int nfr = a.LDF >> 4;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), qi = lane & 15, g = lane >> 4;
    const bool plo0 = ptiles_lo_zero(a.ptiles);
    {
        const u32x4* src = (const u32x4*)a.ptiles;
        u32x4 v[20];
#pragma unroll
        for (int t = 0; t < 20; ++t) v[t] = src[tid + 256 * t];
#pragma unroll
        for (int t = 0; t < 20; ++t) ((u32x4*)smem)[tid + 256 * t] = v[t];
    }
    __syncthreads();
    unsigned long long best = 0ull;
    for (int tile = blockIdx.x; tile < 2 * a.nbq; tile += gridDim.x) {
        const bool isq = tile < a.nbq;
        const float* X = isq ? a.q : a.k;
        const int64_t r = (int64_t)(isq ? tile : tile - a.nbq) * 64 + w * 16 + qi;
        const bool ok = r < a.rows;
        const float* xr = X + head_row_off(ok ? r : a.rows - 1, a.heads, a.stride);
        short8_t xh[2], xl[2];
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float4 v0 = *(const float4*)(xr + ks * 32 + g * 8), v1 = *(const float4*)(xr + ks * 32 + g * 8 + 4);
            if (!ok) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(xs[e], xs[e], ss);
            split8(xs, xh[ks], xl[ks]);
        }
        float mx = -INFINITY;
        int am = 0x7fffffff;
        // per accumulator the product order of tile_rows_gemm (hi*lo, lo*hi, hi*hi over ks = 0, 1), so the chunk kernels rebuild bit-identical dd values; FOUR feature
        // fragments at a time keep four independent MFMA chains in flight (one chain of six dependent MFMAs per fragment left the pipe idle between issues)
        auto products = [&](int f, int ks, float4_t& c) __attribute__((always_inline)) {
            const unsigned char* sPh = smem + (f >> 2) * (2 * FT_BYTES);
            const unsigned char* sPl = sPh + FT_BYTES;
            const uint32_t o = lroff((f & 3) * 16 + qi, ks * 32 + g * 8);
            const short8_t ah = *(const short8_t*)(sPh + o);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[ks], c, 0, 0, 0);
            if (!plo0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const short8_t*)(sPl + o), xh[ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[ks], c, 0, 0, 0);
        };
        auto take = [&](int f, const float4_t& c) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = f * 16 + g * 4 + rr;
                const bool t0 = (col < a.m) & ((c[rr] > mx) | ((c[rr] == mx) & (col < am)));
                mx = t0 ? c[rr] : mx;
                am = t0 ? col : am;
            }
        };
        int f = 0;
        for (; f + 4 <= nfr; f += 4) {
            float4_t c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < 4; ++u) products(f + u, ks, c[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) take(f + u, c[u]);
        }
        for (; f < nfr; ++f) {
            float4_t c = (float4_t){0.f, 0.f, 0.f, 0.f};
            products(f, 0, c);
            products(f, 1, c);
            take(f, c);
        }
        // a row lives in the four lanes qi, qi + 16, qi + 32, qi + 48
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            const bool t0 = (om > mx) | ((om == mx) & (oa < am));
            mx = t0 ? om : mx;
            am = t0 ? oa : am;
            ss += __shfl_xor(ss, o, 64);
        }
        if (ok) {
            if (isq) {
                if (g == 0) { a.offq[r] = ss * a.c2half + mx; a.amq[r] = am; }
            } else {
                if (g == 0) a.offk[r] = ss * a.c2half;
                const unsigned long long b1 = pack_max(mx, (uint32_t)(r * a.LDF) + (uint32_t)am);
                best = b1 > best ? b1 : best;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ot = __shfl_xor(best, o, 64);
        best = ot > best ? ot : best;
    }
    __syncthreads();   // every wave is done with the projection tiles: their first bytes carry the four wave maxima
    unsigned long long* sbest = (unsigned long long*)smem;
    if (lane == 0) sbest[w] = best;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) best = sbest[q] > best ? sbest[q] : best;
        if (best) atomicMax(a.gmax, best);
    }
}

// ------------------------------------------------------------------------------------------------ chunk state sums
// U_c[m][d] = sum_{j in chunk} phi_a(j)[m] (b_j[d] bs_j),   z[m] = sum_j phi_a(j)[m] w_j     (zmode 1: w = 1, zmode 2: w = ex_scale_j)
// (the three chunk kernels are bodies over a block id and ONE 80 KiB LDS buffer, so that two of them that do not depend on each other can share a launch:
//  favor_fpair_kernel below)
constexpr int FUSED_LDS = 10 * FT_BYTES;
typedef unsigned char FSlab2[2 * FT_BYTES];

__device__ __forceinline__ void favor_fstate_body(const FusedArgs& s, const int bid, unsigned char* const lds) {
    unsigned char* const sBh = lds, * const sBl = lds + FT_BYTES, * const sAh = lds + 2 * FT_BYTES, * const sAl = lds + 3 * FT_BYTES;
    FSlab2* const sP = (FSlab2*)(lds + 4 * FT_BYTES);   // (the projection slab is double-buffered: two barriers per slab)
    float* const sW = (float*)(lds + 8 * FT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = bid % s.S, g = (bid / s.S) % s.G, b = bid / (s.S * s.G);
    const float kmax = unpack_max(*s.gmax);
    const int p = chunk * 64 + w * 16 + fr;
    const bool valid = p < s.N;
    const int ri = f_row(s, p);
    XOperand xa;
    load_x_operand(xa, s.fa, s, b, g, ri, g4, kmax);
    const bool plo0 = ptiles_lo_zero(s.ptiles);
    pslab_dma(sP[0], s.ptiles, 0, w, lane, plo0);      // projection slabs: global -> LDS without staging registers (see pslab_dma)
    f_stage_values(sBh, sBl, s.b, s.b_stride, s.b_scale, s, b, g, chunk, tid);
    if (tid < 64) {
        const int pj = chunk * 64 + tid;
        float wv = pj < s.N ? 1.f : 0.f;
        if (s.zmode == 2) {
            const __amdgpu_buffer_rsrc_t rex = f_rsrc(s.ex_scale + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
            wv = f_ld1(rex, (uint32_t)(f_row(s, pj) * s.G + g) * 4u);
        }
        sW[tid] = wv;
    }
    pslab_dma_wait();
    __syncthreads();
    const uint32_t trow = (uint32_t)g4 * 4u + ((uint32_t)fr >> 2), tcol = (uint32_t)(fr & 3) * 4u;
    short8_t bh[2], bl[2], wh[2], wl[2];   // this wave's 16 value columns d = w*16 + (lane & 15); the weights of the running sums in column 0
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t o0 = lroff(ks * 32 + trow, w * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, w * 16 + tcol);
        bh[ks] = f_tr_operand(sBh, o0, o1);
        bl[ks] = f_tr_operand(sBl, o0, o1);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fr == 0 ? sW[ks * 32 + (e >> 2) * 16 + g4 * 4 + (e & 3)] : 0.f;
        split8(x, wh[ks], wl[ks]);
    }
    const int64_t zs = (int64_t)s.LDF * 64 + s.LDF;
    float* st = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    float* zp = st + (int64_t)s.LDF * 64;
    const int nslab = (s.LDF + FSLAB - 1) / FSLAB;
    for (int sl = 0; sl < nslab; ++sl) {
        const int slab0 = sl * FSLAB;
        const unsigned char* sPc = sP[sl & 1];
        float4_t F[4];
        feat_slab(F, sPc, sPc + FT_BYTES, xa, valid, slab0, s, fr, g4, plo0);
        // (the barrier at the end of the previous trip: its feature tile has been consumed, the other projection buffer has no readers left)
        feat_to_tile(sAh, sAl, F, w, fr, g4);
        if (sl + 1 < nslab) pslab_dma(sP[(sl + 1) & 1], s.ptiles, sl + 1, w, lane, plo0);
        __syncthreads();
        float4_t acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        tile_cols_gemm(acc, sAh, sAl, bh, bl, lane, 2, slab_frags(s, slab0));      // acc[f][r]: feature slab0 + f*16 + g4*4 + r, value column w*16 + fr
        float4_t accz = (float4_t){0.f, 0.f, 0.f, 0.f};  // feature tile w of the slab
        if (s.zmode) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t o0 = lroff(ks * 32 + trow, w * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, w * 16 + tcol);
                accz = mfma3(f_tr_operand(sAh, o0, o1), f_tr_operand(sAl, o0, o1), wh[ks], wl[ks], accz);
            }
        }
        if (sl + 1 < nslab) {      // the next projection slab has landed, for every wave; every wave is done with this trip's tiles (before the stores below:
            pslab_dma_wait();      // the wait counts them too)
            __syncthreads();
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = slab0 + f * 16 + g4 * 4 + r;
                if (m < s.LDF) st[m * 64 + w * 16 + fr] = acc[f][r];
            }
        if (s.zmode && fr == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = slab0 + w * 16 + g4 * 4 + r;
                if (m < s.LDF) zp[m] = accz[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ chunk states, sequential form (round 4)
// The chunk-state launch above leaves U_c per chunk and a second launch (favor_fprefix_kernel) turns them into exclusive prefixes: 74 MB written, 148 MB through the
// prefix pass and 1 056 blocks that each stage the whole projection matrix.  Here a block owns ONE 64-feature slab of one (batch, head) and walks the chunks
// in scan order with the running sum in its accumulators: before chunk c's contribution is added the accumulators ARE the exclusive prefix of chunk c and are
// stored as such -- no prefix launch, one projection slab per block, the next chunk's rows fetched while the current one is multiplied.  B G ceil(m / 64)
// blocks (240 at the README shape); what the consumers read (state[b, g, chunk]) keeps its layout.
struct FSeqStep {
    u32x4 xv[4], vv[4];
    float off, sc[4], wv;
};
__device__ __forceinline__ void fseq_load(FSeqStep& r, const FusedArgs& s, int b, int g, int c, int tid, int w, int fr, int g4, float kmax,
                                          __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ro, __amdgpu_buffer_rsrc_t rv, __amdgpu_buffer_rsrc_t rsc,
                                          __amdgpu_buffer_rsrc_t rex) {
    const int ri = f_row(s, c * 64 + w * 16 + fr);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        r.xv[2 * ks] = f_ld4(rx, (uint32_t)(ri * s.stride + g * 64 + ks * 32 + g4 * 8) * 4u);
        r.xv[2 * ks + 1] = f_ld4(rx, (uint32_t)(ri * s.stride + g * 64 + ks * 32 + g4 * 8 + 4) * 4u);
    }
    r.off = f_ld1(ro, (uint32_t)(ri * s.G + g) * 4u) + (s.fa.use_kmax ? kmax : 0.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = f_row(s, c * 64 + (tid >> 4) + 16 * it);
        r.vv[it] = f_ld4(rv, (uint32_t)(i * s.b_stride + g * 64 + (tid & 15) * 4) * 4u);
        r.sc[it] = s.b_scale ? f_ld1(rsc, (uint32_t)(i * s.G + g) * 4u) : 1.f;
    }
    const int pj = c * 64 + (tid & 63);
    r.wv = pj < s.N ? 1.f : 0.f;
    if (s.zmode == 2) r.wv = f_ld1(rex, (uint32_t)(f_row(s, pj) * s.G + g) * 4u);   // (rows outside [0, N) read as zero)
}

__device__ __forceinline__ void favor_fstate_seq_body(const FusedArgs& s, const int bid, unsigned char* const lds) {
    unsigned char* const sBh = lds, * const sBl = lds + FT_BYTES, * const sAh = lds + 2 * FT_BYTES, * const sAl = lds + 3 * FT_BYTES;
    unsigned char* const sPh = lds + 4 * FT_BYTES, * const sPl = lds + 5 * FT_BYTES;
    float* const sW = (float*)(lds + 6 * FT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int nslab = (s.LDF + FSLAB - 1) / FSLAB;
    const int sl = bid % nslab, g = (bid / nslab) % s.G, b = bid / (nslab * s.G);
    const int slab0 = sl * FSLAB, nf = slab_frags(s, slab0);
    const float kmax = unpack_max(*s.gmax);
    const __amdgpu_buffer_rsrc_t rx = f_rsrc(s.fa.x + (int64_t)b * s.N * s.stride, (int64_t)s.N * s.stride * 4);
    const __amdgpu_buffer_rsrc_t ro = f_rsrc(s.fa.rowoff + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    const __amdgpu_buffer_rsrc_t rv = f_rsrc(s.b + (int64_t)b * s.N * s.b_stride, (int64_t)s.N * s.b_stride * 4);
    const __amdgpu_buffer_rsrc_t rsc = f_rsrc((s.b_scale ? s.b_scale : s.b) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    const __amdgpu_buffer_rsrc_t rex = f_rsrc((s.zmode == 2 ? s.ex_scale : s.b) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    // (round 6: the rows of TWO chunks ahead in flight -- 243 VGPRs, no spills -- measured 282.2 vs 283.6 k tokens/s, alternating: the walk does not wait for its
    //  loads; DESIGN Appendix A)
    FSeqStep nxt;
    fseq_load(nxt, s, b, g, 0, tid, w, fr, g4, kmax, rx, ro, rv, rsc, rex);
    const bool plo0 = ptiles_lo_zero(s.ptiles);
    {
        PSlabRegs pre;
        pslab_load(pre, s.ptiles, sl, tid);
        pslab_store(sPh, pre, tid);     // (hi tile | lo tile, contiguous)
    }
    __syncthreads();    // every thread's pieces of the projection slab are in LDS before chunk 0 builds features from the whole tile
    float4_t acc[4], accz = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const uint32_t trow = (uint32_t)g4 * 4u + ((uint32_t)fr >> 2), tcol = (uint32_t)(fr & 3) * 4u;
    const int64_t zs = (int64_t)s.LDF * 64 + s.LDF;
    float* const st0 = s.state + (((int64_t)b * s.G + g) * s.S) * zs;
    for (int c = 0; c < s.S; ++c) {
        const FSeqStep cur = nxt;
        if (c + 1 < s.S) fseq_load(nxt, s, b, g, c + 1, tid, w, fr, g4, kmax, rx, ro, rv, rsc, rex);
        // the running sums before this chunk = its exclusive prefix
        float* const st = st0 + (int64_t)c * zs;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f >= nf) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(slab0 + f * 16 + g4 * 4 + r) * 64 + w * 16 + fr] = acc[f][r];
        }
        if (s.zmode && fr == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = slab0 + w * 16 + g4 * 4 + r;
                if (m < s.LDF) st[(int64_t)s.LDF * 64 + m] = accz[r];
            }
        }
        // this chunk's operands: key / query rows -> MFMA operand, value rows (times their scale) -> hi / lo tile, weights of the running sums
        XOperand xa;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float xs[8] = {__uint_as_float(cur.xv[2 * ks][0]), __uint_as_float(cur.xv[2 * ks][1]), __uint_as_float(cur.xv[2 * ks][2]), __uint_as_float(cur.xv[2 * ks][3]),
                                 __uint_as_float(cur.xv[2 * ks + 1][0]), __uint_as_float(cur.xv[2 * ks + 1][1]), __uint_as_float(cur.xv[2 * ks + 1][2]),
                                 __uint_as_float(cur.xv[2 * ks + 1][3])};
            split8(xs, xa.h[ks], xa.l[ks]);
        }
        xa.off = cur.off;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t o = lroff((tid >> 4) + 16 * it, (tid & 15) * 4);
            uint2 h, l;
            split_pair(__uint_as_float(cur.vv[it][0]) * cur.sc[it], __uint_as_float(cur.vv[it][1]) * cur.sc[it], h.x, l.x);
            split_pair(__uint_as_float(cur.vv[it][2]) * cur.sc[it], __uint_as_float(cur.vv[it][3]) * cur.sc[it], h.y, l.y);
            *(uint2*)(sBh + o) = h;
            *(uint2*)(sBl + o) = l;
        }
        if (tid < 64) sW[tid] = cur.wv;
        const bool valid = c * 64 + w * 16 + fr < s.N;
        float4_t F[4];
        feat_slab(F, sPh, sPl, xa, valid, slab0, s, fr, g4, plo0);
        feat_to_tile(sAh, sAl, F, w, fr, g4);
        __syncthreads();
        short8_t bh[2], bl[2], wh[2], wl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t o0 = lroff(ks * 32 + trow, w * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, w * 16 + tcol);
            bh[ks] = f_tr_operand(sBh, o0, o1);
            bl[ks] = f_tr_operand(sBl, o0, o1);
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fr == 0 ? sW[ks * 32 + (e >> 2) * 16 + g4 * 4 + (e & 3)] : 0.f;
            split8(x, wh[ks], wl[ks]);
        }
        tile_cols_gemm(acc, sAh, sAl, bh, bl, lane, 2, nf);      // acc[f][r] += sum_j phi(j)[slab0 + f*16 + g4*4 + r] b_j[w*16 + fr]
        if (s.zmode) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t o0 = lroff(ks * 32 + trow, w * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, w * 16 + tcol);
                accz = mfma3(f_tr_operand(sAh, o0, o1), f_tr_operand(sAl, o0, o1), wh[ks], wl[ks], accz);
            }
        }
        __syncthreads();   // the next chunk overwrites the feature / value tiles
    }
}

// state[b, g, chunk] <- sum of the states of the chunks before it (exclusive prefix), four elements per thread.  Loads go out in batches of eight
// before the dependent stores (the array aliases itself, so the compiler would otherwise keep every load behind the previous store: S round trips).
__global__ void favor_fprefix_kernel(float* __restrict__ state, int64_t BG, int S, int64_t elems4) {
    const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= BG * elems4) return;
    const int64_t bg = tix / elems4, e = tix - bg * elems4;
    float4_t* p = (float4_t*)state + bg * S * elems4 + e;
    float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < S; k0 += 8) {
        float4_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k0 + j < S ? p[(k0 + j) * elems4] : (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k0 + j < S) p[(k0 + j) * elems4] = acc;
            acc += v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------ chunk outputs, scan A
// y_i[d] = sum_m T_prev[m][d] phi_x(i)[m] + sum_{j <= i} b_j[d] (phi_a(j) . phi_x(i))      (zmode 1: divided by phi_x(i) . (z_i + eps))
__device__ __forceinline__ void favor_fout_a_body(const FusedArgs& s, const int bid, unsigned char* const lds) {
    unsigned char* const sAh = lds, * const sAl = lds + FT_BYTES;
    FSlab2* const sT = (FSlab2*)(lds + 2 * FT_BYTES);   // state and projection slabs double-buffered
    FSlab2* const sP = (FSlab2*)(lds + 6 * FT_BYTES);
    unsigned char* const sBh = sAh;   // the value rows take the feature tile's place after the slab loop
    unsigned char* const sBl = sAl;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = bid % s.S, g = (bid / s.S) % s.G, b = bid / (s.S * s.G);
    const float kmax = unpack_max(*s.gmax);
    const int64_t zs = (int64_t)s.LDF * 64 + s.LDF;
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const __amdgpu_buffer_rsrc_t rt = f_rsrc(st0, (int64_t)s.LDF * 64 * 4);
    const __amdgpu_buffer_rsrc_t rz = f_rsrc(st0 + (int64_t)s.LDF * 64, s.zmode == 1 ? s.LDF * 4 : 0);
    const int pi = chunk * 64 + w * 16 + fr;
    const bool vi = pi < s.N;
    const int ri = f_row(s, pi);
    const int64_t rowi = ((int64_t)b * s.N + (vi ? ri : 0)) * s.G + g;
    XOperand xa, xc;
    load_x_operand(xa, s.fa, s, b, g, ri, g4, kmax);
    load_x_operand(xc, s.fx, s, b, g, ri, g4, kmax);
    u32x4 pt[4], pz[4];
    const bool plo0 = ptiles_lo_zero(s.ptiles);
    pslab_dma(sP[0], s.ptiles, 0, w, lane, plo0);      // projection slabs: global -> LDS without staging registers (see pslab_dma)
    tslab_load(pt, rt, 0, tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) pz[q] = f_ld4(rz, (uint32_t)(q * 16 + g4 * 4) * 4u);

    float4_t P[4], acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        P[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    float den = 0.f;
    const int nslab = (s.LDF + FSLAB - 1) / FSLAB;
    tile_stage(sT[0], sT[0] + FT_BYTES, pt, tid);
    pslab_dma_wait();
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
        const int slab0 = sl * FSLAB;
        const unsigned char* sPc = sP[sl & 1];
        const unsigned char* sTh = sT[sl & 1];
        const unsigned char* sTl = sTh + FT_BYTES;
        u32x4 zc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) zc[q] = pz[q];
        if (sl + 1 < nslab) {
            tslab_load(pt, rt, slab0 + FSLAB, tid);
#pragma unroll
            for (int q = 0; q < 4; ++q) pz[q] = f_ld4(rz, (uint32_t)(slab0 + FSLAB + q * 16 + g4 * 4) * 4u);
        }
        float4_t Fa[4], F[4];
        feat_slab2(Fa, F, sPc, sPc + FT_BYTES, xa, xc, vi, slab0, s, fr, g4, plo0);
        // (the barrier at the end of the previous trip: its GEMMs are done with the feature tile and with the other slab buffers)
        feat_to_tile(sAh, sAl, Fa, w, fr, g4);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) den = fmaf(F[f][r], __uint_as_float(zc[f][r]) + s.den_eps, den);
        short8_t Ch[2], Cl[2];
        acc_to_operand(Ch, Cl, F);
        if (sl + 1 < nslab) {      // next slab's operands into the other buffers while this slab's GEMMs run
            tile_stage(sT[(sl + 1) & 1], sT[(sl + 1) & 1] + FT_BYTES, pt, tid);
            pslab_dma(sP[(sl + 1) & 1], s.ptiles, sl + 1, w, lane, plo0);
        }
        __syncthreads();
        const int nks = (min(64, s.LDF - slab0) + 31) >> 5;
        tile_rows_gemm_perm(P, sAh, sAl, Ch, Cl, fr, g4, nks);   // pair products phi_a(j) . phi_x(i)
        tile_cols_gemm(acc, sTh, sTl, Ch, Cl, lane, nks);        // inter-chunk: T_prev^T phi_x(i)
        if (sl + 1 < nslab) {      // the next projection slab has landed, for every wave; every wave is done with this trip's tiles
            pslab_dma_wait();
            __syncthreads();
        }
    }
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (jf * 16 + g4 * 4 + r > w * 16 + fr) P[jf][r] = 0.f;
    float inv_n = 1.f;
    if (s.zmode == 1) {
        float part = den;
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) part += (P[jf][0] + P[jf][1]) + (P[jf][2] + P[jf][3]);
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        inv_n = 1.f / part;
        if (vi && g4 == 0 && s.inv_out) s.inv_out[rowi] = inv_n;
    }
    short8_t Ph[2], Pl[2];
    acc_to_operand(Ph, Pl, P);
    __syncthreads();   // every wave is done with the last feature tile
    f_stage_values(sBh, sBl, s.b, s.b_stride, s.b_scale, s, b, g, chunk, tid);
    __syncthreads();
    tile_cols_gemm(acc, sBh, sBl, Ph, Pl, lane);   // intra-chunk: sum_j b_j[d] P[j][i]
    if (!vi) return;
    float* yp = s.y + ((int64_t)b * s.N + ri) * s.y_stride + g * 64;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        float4 o = make_float4(acc[df][0] * inv_n, acc[df][1] * inv_n, acc[df][2] * inv_n, acc[df][3] * inv_n);
        float4* d4 = (float4*)(yp + df * 16 + g4 * 4);
        if (s.accumulate) {
            const float4 old = *d4;
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *d4 = o;
        if (s.y_lp) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
            *(uint2*)(s.y_lp + ((int64_t)b * s.N + ri) * s.y_stride + g * 64 + df * 16 + g4 * 4) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------ chunk outputs, scan B + feature-map backward + projection adjoint
// dphi_i[m] = sum_d T_prev[m][d] c_i[d] + sum_{j <= i} phi_a(j)[m] (b_j . c_i + E[j][i]) + (running-sum terms)      (c_i times c_scale_i)
//   zmode 1: E[j][i] = ex_scale_i,  + ex_scale_i (z_prev[m] + ex_const);     zmode 2: E[j][i] = ex_scale_j,  + z_prev[m]
// v = (phi_x(i) - ratio eps) dphi_i,  t = sum_m v[m],  dx_i = sum_m v[m] P[m] - [query] t P[argmax_i] - t c^2 x_i
__device__ __forceinline__ void favor_fout_b_body(const FusedArgs& s, const int bid, unsigned char* const lds) {
    FSlab2* const sT = (FSlab2*)lds;   // double-buffered slabs
    unsigned char* const sAh = lds + 4 * FT_BYTES, * const sAl = lds + 5 * FT_BYTES;
    FSlab2* const sP = (FSlab2*)(lds + 6 * FT_BYTES);
    float* const sred = (float*)sAh;    // (block reduction of the keys' t after the slab loop: the feature tile is dead by then; 80 KiB exactly -> two blocks per CU)
    unsigned char* const sBh = sT[0];   // the value tile is dead once the pair products exist: the first state slab takes its place
    unsigned char* const sBl = sT[0] + FT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = bid % s.S, g = (bid / s.S) % s.G, b = bid / (s.S * s.G);
    FT_T(ft_start);
    FT_DECL;
    const float kmax = unpack_max(*s.gmax);
    const __amdgpu_buffer_rsrc_t rc = f_rsrc(s.c + (int64_t)b * s.N * s.c_stride, (int64_t)s.N * s.c_stride * 4);
    const __amdgpu_buffer_rsrc_t rcs = f_rsrc((s.c_scale ? s.c_scale : s.c) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    const __amdgpu_buffer_rsrc_t rex = f_rsrc(s.ex_scale + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G * 4);
    const int64_t zs = (int64_t)s.LDF * 64 + s.LDF;
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const __amdgpu_buffer_rsrc_t rt = f_rsrc(st0, (int64_t)s.LDF * 64 * 4);
    const __amdgpu_buffer_rsrc_t rz = f_rsrc(st0 + (int64_t)s.LDF * 64, (int64_t)s.LDF * 4);
    const int pi = chunk * 64 + w * 16 + fr;
    const bool vi = pi < s.N;
    const int ri = f_row(s, pi);
    const int64_t rowi = ((int64_t)b * s.N + (vi ? ri : 0)) * s.G + g;
    XOperand xa, xx;
    load_x_operand(xa, s.fa, s, b, g, ri, g4, kmax);
    load_x_operand(xx, s.fx, s, b, g, ri, g4, kmax);
    u32x4 pt[4];
    const bool plo0 = ptiles_lo_zero(s.ptiles);
    pslab_dma(sP[0], s.ptiles, 0, w, lane, plo0);      // projection slabs go global -> LDS without staging registers (sP is not aliased by the prologue's tiles)
    tslab_load(pt, rt, 0, tid);
    f_stage_values(sBh, sBl, s.b, s.b_stride, s.b_scale, s, b, g, chunk, tid);
    short8_t Ch[2], Cl[2];   // c_i (times c_scale) as B operand, natural order d = ks*32 + g4*8 + e
    {
        const float cs = s.c_scale ? f_ld1(rcs, (uint32_t)(ri * s.G + g) * 4u) : 1.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 v = f_ld4(rc, (uint32_t)(ri * s.c_stride + g * 64 + ks * 32 + g4 * 8 + q * 4) * 4u);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[q * 4 + e] = __uint_as_float(v[e]) * cs;
            }
            split8(x, Ch[ks], Cl[ks]);
        }
    }
    const float exs = vi ? f_ld1(rex, (uint32_t)(ri * s.G + g) * 4u) : 0.f;   // ex_scale_i (zmode 1)
    __syncthreads();
    float4_t P[4];   // P[j][i] = b_j . c_i (+ E), masked to j <= i
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) P[jf] = (float4_t){0.f, 0.f, 0.f, 0.f};
    tile_rows_gemm(P, sBh, sBl, Ch, Cl, fr, g4);
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = jf * 16 + g4 * 4 + r, pj = chunk * 64 + jl;
            float e = exs;
            if (s.zmode == 2) e = f_ld1(rex, (uint32_t)(f_row(s, pj) * s.G + g) * 4u);
            P[jf][r] = (jl > w * 16 + fr || pj >= s.N) ? 0.f : P[jf][r] + e;
        }
    short8_t Ph[2], Pl[2];
    acc_to_operand(Ph, Pl, P);

    const float zf = s.zmode == 1 ? exs : 1.f;
    const float cadd = s.zmode == 1 ? exs * s.ex_const : 0.f;
    float4_t dxa[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) dxa[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float tp = 0.f;
    const int nslab = (s.LDF + FSLAB - 1) / FSLAB;
    __syncthreads();   // the pair products have read the value tile
    tile_stage(sT[0], sT[0] + FT_BYTES, pt, tid);
    pslab_dma_wait();
    __syncthreads();
    FT_T(ft_loop);
    FT_ACC(0, ft_loop - ft_start);
    for (int sl = 0; sl < nslab; ++sl) {
        const int slab0 = sl * FSLAB;
        const unsigned char* sPc = sP[sl & 1];
        const unsigned char* sTh = sT[sl & 1];
        const unsigned char* sTl = sTh + FT_BYTES;
        FT_T(ft0);
        if (sl + 1 < nslab) tslab_load(pt, rt, slab0 + FSLAB, tid);
        u32x4 zc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) zc[q] = f_ld4(rz, (uint32_t)(slab0 + q * 16 + g4 * 4) * 4u);
        float4_t F[4], Fx[4];
        feat_slab2(F, Fx, sPc, sPc + FT_BYTES, xa, xx, vi, slab0, s, fr, g4, plo0);
        FT_T(ft1);
        // (no barrier here: the one at the end of the previous trip already says that its GEMMs are done with the feature tile and with the other slab buffers)
        feat_to_tile(sAh, sAl, F, w, fr, g4);
        if (sl + 1 < nslab) {      // next slab's operands into the other buffers while this slab's GEMMs run
            tile_stage(sT[(sl + 1) & 1], sT[(sl + 1) & 1] + FT_BYTES, pt, tid);
            pslab_dma(sP[(sl + 1) & 1], s.ptiles, sl + 1, w, lane, plo0);
        }
        FT_T(ft2);
        __syncthreads();
        FT_T(ft3);
        FT_ACC(1, ft1 - ft0); FT_ACC(2, ft2 - ft1); FT_ACC(3, ft3 - ft2);
        float4_t acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const int nf = slab_frags(s, slab0);
        tile_rows_gemm(acc, sTh, sTl, Ch, Cl, fr, g4, nf);   // inter-chunk: T_prev c_i
        tile_cols_gemm(acc, sAh, sAl, Ph, Pl, lane, 2, nf);  // intra-chunk: sum_j phi_a(j)[m] P[j][i]
#ifdef SA_TIMING_FAVOR
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
#endif
        FT_T(ft4);
        FT_ACC(4, ft4 - ft3);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mi = slab0 + f * 16 + g4 * 4 + r;
                const float dphi = acc[f][r] + zf * __uint_as_float(zc[f][r]) + cadd;
                const float v = (vi && mi < s.m) ? (Fx[f][r] - s.reps) * dphi : 0.f;
                tp += v;
                acc[f][r] = v;
            }
        short8_t Dh[2], Dl[2];
        acc_to_operand(Dh, Dl, acc);
        FT_T(ft5);
        tile_cols_gemm(dxa, sPc, sPc + FT_BYTES, Dh, Dl, lane, (min(64, s.LDF - slab0) + 31) >> 5, 4, plo0);   // dx^T[d][i] += sum_m P[m][d] v[m][i]
#ifdef SA_TIMING_FAVOR
        asm volatile("s_nop 0" ::"v"(dxa[0][0]), "v"(dxa[1][0]), "v"(dxa[2][0]), "v"(dxa[3][0]));
#endif
        FT_T(ft6);
        FT_ACC(5, ft5 - ft4); FT_ACC(6, ft6 - ft5);
        if (sl + 1 < nslab) {      // the next projection slab has landed (requested a slab's GEMMs ago) -- for every wave; every wave is done with this trip's tiles
            pslab_dma_wait();
            __syncthreads();
        }
    }
    FT_T(ft_ep);
    float t = tp;
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (vi) {
        const int64_t xoff = ((int64_t)b * s.N + ri) * s.stride + g * 64;
        const float* xr = s.fx.x + xoff;
        float* dxr = s.dx + xoff;
        const int am = s.is_query ? s.amx[rowi] : 0;
        const float* pa = s.ps + (int64_t)min(am, s.m - 1) * 64;
        const float ts = s.is_query ? t : 0.f, tc = t * s.c2;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const int d0 = df * 16 + g4 * 4;
            const float4 xv = *(const float4*)(xr + d0), pv = *(const float4*)(pa + d0);
            const float4 o = make_float4(dxa[df][0] - ts * pv.x - tc * xv.x, dxa[df][1] - ts * pv.y - tc * xv.y, dxa[df][2] - ts * pv.z - tc * xv.z,
                                         dxa[df][3] - ts * pv.w - tc * xv.w);
            *(float4*)(dxr + d0) = o;
            if (s.dx_lp) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                *(uint2*)(s.dx_lp + xoff + d0) = pk;
            }
        }
    }
    if (!s.is_query) {   // keys: the sum of t over all rows goes to the row that holds the global maximum (fix-up launch)
        float tb = (vi && g4 == 0) ? t : 0.f;
        tb = wave_sum(tb);
        __syncthreads();   // every wave is past its last read of the feature tile
        if (lane == 0) sred[w] = tb;
        __syncthreads();
        if (tid == 0) s.tsum[bid] = (sred[0] + sred[1]) + (sred[2] + sred[3]);   // one partial per block, summed in a fixed order by the fix-up launch
    }
#ifdef SA_TIMING_FAVOR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FT_T(ft_end);
    FT_ACC(7, ft_end - ft_ep); FT_ACC(8, 1); FT_ACC(9, ft_end - ft_start);
    FT_FLUSH();
#endif
}

__global__ __launch_bounds__(256, 2) void favor_fstate_kernel(const FusedArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    favor_fstate_body(s, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(256, 2) void favor_fstate_seq_kernel(const FusedArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    favor_fstate_seq_body(s, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fout_a_kernel(const FusedArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    favor_fout_a_body(s, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fout_b_kernel(const FusedArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    favor_fout_b_body(s, (int)blockIdx.x, lds);
}
// Two chunk kernels that do not depend on each other in ONE launch (blocks [0, nb) run scan B, the rest the other body): B * G * ceil(N / 64) = 1 056 blocks are
// 2.06 rounds on the 512 resident slots, i.e. every one of these launches ends in a tail of 32 blocks on an otherwise idle chip (measured: N = 1 344 -- 1 008
// blocks -- is 15-18 % faster than N = 1 400); back to back the pair is 4.1 rounds with ONE tail, and the lighter body's blocks fill it.
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_b_state_kernel(const FusedArgs sb, const FusedArgs ss, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nb) favor_fout_b_body(sb, (int)blockIdx.x, lds);
    else favor_fstate_body(ss, (int)blockIdx.x - nb, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_b_a_kernel(const FusedArgs sb, const FusedArgs sa, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nb) favor_fout_b_body(sb, (int)blockIdx.x, lds);
    else favor_fout_a_body(sa, (int)blockIdx.x - nb, lds);
}

// The same launches with the LOCAL-WINDOW heads' blocks appended (sa_local_attn_args): they depend on q | k | v (or d attn) only, like the FAVOR+ chain, and
// each of their launches ends in a tail of its own; here their blocks fill the FAVOR+ launches' tails instead.  [scan A | local forward],
// [scan B dq | reversed states | local dq], [scan B dk | scan A dv | local dk dv].
static_assert(LA_SPLIT_LDS <= FUSED_LDS, "the local-attention bodies run in the FAVOR+ kernels' LDS buffer");
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fout_a_la_kernel(const FusedArgs sa, const LAArgs la, const int nfav) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nfav) favor_fout_a_body(sa, (int)blockIdx.x, lds);
    else local_attn_q_split_body<0>(la, (int)blockIdx.x - nfav, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_b_state_la_kernel(const FusedArgs sb, const FusedArgs ss, const LAArgs la, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nb) favor_fout_b_body(sb, (int)blockIdx.x, lds);
    else if ((int)blockIdx.x < 2 * nb) favor_fstate_body(ss, (int)blockIdx.x - nb, lds);
    else local_attn_q_split_body<1>(la, (int)blockIdx.x - 2 * nb, lds);
}
// with the sequential chunk states (favor_fstate_seq_body: B G ceil(m / 64) long-running blocks): they come FIRST in the grid so that they start at once and the
// short blocks of the other bodies fill the slots beside them
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fseq_la_kernel(const FusedArgs ss, const LAArgs la, const int nst) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nst) favor_fstate_seq_body(ss, (int)xcd_remap(blockIdx.x, (uint32_t)nst), lds);   // (the feature-slab blocks of a (batch, head) walk the SAME k / v rows: one XCD, one L2)
    else local_attn_q_split_body<0>(la, (int)blockIdx.x - nst, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_seq_b_kernel(const FusedArgs ss, const FusedArgs sb, const int nst, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nst) favor_fstate_seq_body(ss, (int)xcd_remap(blockIdx.x, (uint32_t)nst), lds);   // (the feature-slab blocks of a (batch, head) walk the SAME k / v rows: one XCD, one L2)
    else favor_fout_b_body(sb, (int)blockIdx.x - nst, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_seq_b_la_kernel(const FusedArgs ss, const FusedArgs sb, const LAArgs la, const int nst, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nst) favor_fstate_seq_body(ss, (int)xcd_remap(blockIdx.x, (uint32_t)nst), lds);   // (the feature-slab blocks of a (batch, head) walk the SAME k / v rows: one XCD, one L2)
    else if ((int)blockIdx.x < nst + nb) favor_fout_b_body(sb, (int)blockIdx.x - nst, lds);
    else local_attn_q_split_body<1>(la, (int)blockIdx.x - nst - nb, lds);
}
__global__ __launch_bounds__(256, FUSED_WPS) void favor_fpair_b_a_la_kernel(const FusedArgs sb, const FusedArgs sa, const LAArgs la, const int nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[FUSED_LDS];
    if ((int)blockIdx.x < nb) favor_fout_b_body(sb, (int)blockIdx.x, lds);
    else if ((int)blockIdx.x < 2 * nb) favor_fout_a_body(sa, (int)blockIdx.x - nb, lds);
    else local_attn_kv_split_body(la, (int)blockIdx.x - 2 * nb, lds);
}

// keys: the global-max element (head row r*, feature f*) takes -(sum of all t): dk[r*] -= T P[f*];  T = the per-block partials in a fixed order (deterministic)
__global__ __launch_bounds__(64) void favor_fkey_fix_kernel(float* __restrict__ dx, unsigned short* __restrict__ dx_lp, int x_stride, int heads,
                                                            const unsigned long long* __restrict__ gmax, const float* __restrict__ partial, int nblk,
                                                            const float* __restrict__ ps, int LDF) {
    // one wave, up to ~30 partials per lane: the loads of eight trips are issued together (the additions keep their order: bit-identical to the plain loop,
    // whose every trip waited for its own load)
    float t = 0.f;
    int i = threadIdx.x;
    for (; i + 7 * 64 < nblk; i += 8 * 64) {
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = partial[i + u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += p[u];
    }
    for (; i < nblk; i += 64) t += partial[i];
    t = wave_sum(t);
    const uint32_t idx = 0xffffffffu - (uint32_t)(*gmax & 0xffffffffull);
    const int64_t r = idx / (uint32_t)LDF;
    const int f = (int)(idx % (uint32_t)LDF);
    const int64_t o = head_row_off(r, heads, x_stride) + threadIdx.x;
    const float v = dx[o] - t * ps[(int64_t)f * 64 + threadIdx.x];
    dx[o] = v;
    if (dx_lp) dx_lp[o] = f32_to_bf16(v);
}

// dden[row] = -(dout . out) * inv over the head's 64 columns (one wave per head row)
__global__ void favor_fdden_kernel(const float* __restrict__ dout, const float* __restrict__ out, int stride, int G, const float* __restrict__ inv,
                                   float* __restrict__ dden, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t rp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rp >= rows) return;
    const int64_t r = rp / G;
    const int g = (int)(rp - r * G);
    float sv = dout[r * stride + g * 64 + lane] * out[r * stride + g * 64 + lane];
    sv = wave_sum(sv);
    if (lane == 0) dden[rp] = -sv * inv[rp];
}

// the same with 16-byte loads: sixteen lanes per head row, four head rows per wave and trip, four trips per wave (all loads of a wave issued before its first
// reduction); 16-byte aligned rows.  The one-row-per-wave form above launched 67 200 waves of two 4-byte loads each (31 us for 34 MB at N = 1 400, batch 6).
__global__ __launch_bounds__(256) void favor_fdden_v4_kernel(const float* __restrict__ dout, const float* __restrict__ out, int stride, int G, const float* __restrict__ inv,
                                                             float* __restrict__ dden, int64_t rows) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 a[4], b[4];
    int64_t rp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        rp[t] = (wave * 4 + t) * 4 + sub;
        a[t] = b[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rp[t] < rows) {
            const int64_t r = rp[t] / G;
            const int64_t o = r * stride + (rp[t] - r * G) * 64 + l16 * 4;
            a[t] = *(const float4*)(dout + o);
            b[t] = *(const float4*)(out + o);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float sv = fmaf(a[t].w, b[t].w, fmaf(a[t].z, b[t].z, fmaf(a[t].y, b[t].y, a[t].x * b[t].x)));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
        if (l16 == 0 && rp[t] < rows) dden[rp[t]] = -sv * inv[rp[t]];
    }
}

}  // namespace sa

#ifdef SA_TIMING_FAVOR
extern "C" int sa_debug_timing_favor(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(sa::g_ftime), 24 * 8);
    if (reset) { unsigned long long z[24] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(sa::g_ftime), z, 24 * 8); }
    return 0;
}
#endif

using namespace sa;

static int fused_check(int B, int N, int G, int m, int dh, int stride) {
    if (B <= 0 || N <= 0 || G <= 0 || m <= 0) return SA_EINVAL;
    if (dh != 64 || m > 272 || (stride & 3) || stride < G * 64) return SA_EUNSUPPORTED;
    if ((int64_t)N * stride * 4 >= ((int64_t)1 << 31) || (int64_t)B * N * G * 272 >= ((int64_t)1 << 32) - 1) return SA_EUNSUPPORTED;
    return 0;
}
static inline int ldf_of(int m) { return (m + 15) / 16 * 16; }

extern "C" int64_t sa_favor_fused_state_bytes(int B, int N, int G, int m) {
    return (int64_t)B * G * ((N + 63) / 64) * ldf_of(m) * 65 * 4;
}

extern "C" int sa_favor_fused_proj_tiles(const float* ps, int m, void* tiles, void* stream) {
    if (!ps || !tiles || m <= 0 || m > 272) return SA_EINVAL;
    (void)hipMemsetAsync((unsigned char*)tiles + PT_FLAG_OFF, 0, 16, (hipStream_t)stream);   // "every lo half-word is zero" until a block finds one that is not
    SA_CHECK_LAUNCH();
    SA_LAUNCH(favor_proj_tiles_kernel, dim3(5), dim3(256), 0, (hipStream_t)stream, ps, m, (unsigned char*)tiles);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_fused_prepass(const float* q, const float* k, int stride, int G, const void* tiles, float* offq, int32_t* amq, float* offk, void* gmax_ws,
                                      int64_t rows, int m, int dh, void* stream) {
    if (!q || !k || !tiles || !offq || !amq || !offk || !gmax_ws || rows <= 0 || G <= 0) return SA_EINVAL;
    if (dh != 64 || m <= 0 || m > 272 || (stride & 3) || stride < G * 64 || rows * 272 >= ((int64_t)1 << 32) - 1) return SA_EUNSUPPORTED;
    PrepassArgs a = {};
    a.q = q; a.k = k; a.ptiles = (const unsigned char*)tiles; a.offq = offq; a.offk = offk; a.amq = amq; a.gmax = (unsigned long long*)gmax_ws; a.rows = rows; a.m = m;
    a.LDF = ldf_of(m); a.stride = stride; a.heads = G; a.nbq = (int)((rows + 63) / 64);   // tiles of 64 head rows per operand
    const float c = powf((float)dh, -0.25f);
    a.c2half = 0.5f * c * c;
    const size_t lds = (size_t)5 * 2 * FT_BYTES;
    static std::atomic<uint64_t> attr_done{0};
    configure_once_per_device(attr_done, [] { (void)hipFuncSetAttribute((const void*)favor_prepass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 2 * FT_BYTES); });
    hipMemsetAsync(gmax_ws, 0, 8, (hipStream_t)stream);
    static std::atomic<int> cu_count[64];   // per device ordinal, queried once (0 = not yet)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int cus = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (cus <= 0) {
        cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cu_count[dev & 63].store(cus, std::memory_order_relaxed);
    }
    const unsigned nblk = (unsigned)std::min<int64_t>(2 * (int64_t)a.nbq, 2 * (int64_t)cus);   // persistent: two blocks per CU
    SA_LAUNCH(favor_prepass_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

// sa_local_attn_args -> the local-attention launch arguments (split-bf16 path only); nblk = its grid
static int la_from_args(const sa_local_attn_args* p, int B, int N, bool backward, LAArgs& a, unsigned& nblk) {
    if (!p->q || !p->k || !p->v || p->L <= 0 || p->W <= 0) return SA_EINVAL;
    if (la_exact()) return SA_EUNSUPPORTED;            // the exact-fp32 kernels live in local_attn.hip: call sa_local_attn_* separately
    a = LAArgs{};
    if (int rc = fill_la(a, p->q_stride, p->q_off, p->k_stride, p->k_off, p->v_stride, p->v_off, p->o_stride, p->o_off, B, N, p->L, p->W, 64)) return rc;
    a.q = p->q; a.k = p->k; a.v = p->v;
    if (!backward) {
        if (!p->o || !p->lse) return SA_EINVAL;
        a.o = p->o; a.lse_out = p->lse; a.o_lp = (unsigned short*)p->o_lp;
    } else {
        if (!p->out || !p->dout || !p->lse_in || !p->dq || !p->dk || !p->dv || !p->Dbuf) return SA_EINVAL;
        a.out = p->out; a.dout = p->dout; a.lse_in = p->lse_in; a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.Dbuf_out = p->Dbuf; a.Dbuf_in = p->Dbuf;
        a.dv_lp = (unsigned short*)p->dv_lp;
    }
    nblk = (unsigned)(B * p->L * ((N + LT - 1) / LT));
    return 0;
}

static void fused_common(FusedArgs& s, const void* tiles, const float* ps, const void* gmax, int B, int N, int G, int m, int stride) {
    s.ptiles = (const unsigned char*)tiles; s.ps = ps; s.gmax = (const unsigned long long*)gmax;
    s.B = B; s.N = N; s.G = G; s.m = m; s.LDF = ldf_of(m); s.S = (N + 63) / 64; s.stride = stride;
    const float c = powf(64.f, -0.25f);
    s.ratio = 1.f / sqrtf((float)m); s.reps = s.ratio * 1e-4f; s.c2 = c * c;
}
// SA_PP_DBG bit 11 keeps the parallel chunk-state launch + prefix launch (A/B runs; the two forms differ in the association of the fp32 sums only)
// The walk is a chain of S dependent chunk steps (~4 us each) per (batch, head, slab) block, whatever the batch: it pays when there are enough (batch, head) pairs
// to fill the chip with such chains (README shape: 6 x 8 pairs x 5 slabs = 240 blocks, 22 steps), not for ONE long sequence (N = 14 000, batch 1: 40 blocks x 219
// steps = 0.9 ms against 0.16 ms for the parallel states + prefix launches; measured 207 vs 287 k tokens/s).
// SA_PP_DBG bit 11: never (A/B runs); SA_DBG_FAVOR_SEQ_ALWAYS: always (the tests run both forms against the reference at small shapes).
static inline bool fused_seq_states(int B, int G) { return !(g_tunables.pp_dbg & 2048u) && (B * G >= 40 || dbg(SA_DBG_FAVOR_SEQ_ALWAYS)); }
static int fused_prefix(float* state, int B, int G, int S, int LDF, hipStream_t st) {
    const int64_t bg = (int64_t)B * G, e4 = ((int64_t)LDF * 65) / 4;   // LDF % 16 == 0
    SA_LAUNCH(favor_fprefix_kernel, dim3((unsigned)((bg * e4 + 255) / 256)), dim3(256), 0, st, state, bg, S, e4);
    SA_CHECK_LAUNCH();
    return 0;
}

// forward of the global heads: attn_i = (phi_q(i) . sum_{j<=i} phi_k(j) (x) v_j) / (phi_q(i) . (sum_{j<=i} phi_k(j) + eps)); `state` keeps the chunk prefixes
// (sum phi_k (x) v | sum phi_k) for the dq' scan of the backward pass
extern "C" int sa_favor_fused_fwd(const float* q, const float* k, const float* v, int stride, const void* tiles, const float* ps, const float* offq,
                                  const float* offk, const void* gmax_ws, float* attn, int attn_stride, float* inv_out, float den_eps, int B, int N, int G, int m,
                                  float* state, void* attn_lp, const sa_local_attn_args* la, void* stream) {
    if (!q || !k || !v || !tiles || !ps || !offq || !offk || !gmax_ws || !attn || !inv_out || !state) return SA_EINVAL;
    LAArgs laa;
    unsigned nla = 0;
    if (la) {
        if (int rc = la_from_args(la, B, N, false, laa, nla)) return rc;
    }
    if (int rc = fused_check(B, N, G, m, 64, stride)) return rc;
    if ((attn_stride & 3) || (int64_t)N * attn_stride * 4 >= ((int64_t)1 << 31)) return SA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    FusedArgs s = {};
    fused_common(s, tiles, ps, gmax_ws, B, N, G, m, stride);
    s.fa = FeatSrc{k, offk, 1, 0};
    s.fx = FeatSrc{q, offq, 0, 0};
    s.b = v; s.b_stride = stride; s.state = state; s.zmode = 1; s.den_eps = den_eps; s.inv_out = inv_out; s.y = attn; s.y_stride = attn_stride;
    s.y_lp = (unsigned short*)attn_lp;
    const unsigned nblk = (unsigned)((int64_t)B * G * s.S);
    if (fused_seq_states(B, G)) {
        // sequential chunk states (exclusive prefixes written directly: no prefix launch), the local-window heads' forward beside them; then scan A
        const unsigned nst = (unsigned)((int64_t)B * G * ((s.LDF + FSLAB - 1) / FSLAB));
        if (la) SA_LAUNCH(favor_fseq_la_kernel, dim3(nst + nla), dim3(256), 0, st, s, laa, (int)nst);
        else SA_LAUNCH(favor_fstate_seq_kernel, dim3(nst), dim3(256), 0, st, s);
        SA_CHECK_LAUNCH();
        SA_LAUNCH(favor_fout_a_kernel, dim3(nblk), dim3(256), 0, st, s);
        SA_CHECK_LAUNCH();
        return 0;
    }
    SA_LAUNCH(favor_fstate_kernel, dim3(nblk), dim3(256), 0, st, s);
    SA_CHECK_LAUNCH();
    if (int rc = fused_prefix(state, B, G, s.S, s.LDF, st)) return rc;
    if (la) SA_LAUNCH(favor_fout_a_la_kernel, dim3(nblk + nla), dim3(256), 0, st, s, laa, (int)nblk);
    else SA_LAUNCH(favor_fout_a_kernel, dim3(nblk), dim3(256), 0, st, s);
    SA_CHECK_LAUNCH();
    return 0;
}

// backward of the global heads.  state_fwd: the prefixes sa_favor_fused_fwd left (or NULL: rebuilt into state_ws first); state_ws: scratch of
// sa_favor_fused_state_bytes; dden_ws: B*N*G floats; tsum_ws: B*G*ceil(N/64) floats (per-block partial sums).  dq / dk / dv: head blocks (stride `stride`) like q / k / v.
extern "C" int sa_favor_fused_bwd(const float* q, const float* k, const float* v, int stride, const void* tiles, const float* ps, const float* offq,
                                  const int32_t* amq, const float* offk, const void* gmax_ws, const float* dattn, const float* attn, int attn_stride,
                                  const float* inv, float* dq, float* dk, float* dv, int B, int N, int G, int m, const float* state_fwd, float* state_ws,
                                  float* dden_ws, float* tsum_ws, void* dq_lp, void* dk_lp, void* dv_lp, const sa_local_attn_args* la,
                                  void* stream) {
    if (!q || !k || !v || !tiles || !ps || !offq || !amq || !offk || !gmax_ws || !dattn || !attn || !inv || !dq || !dk || !dv || !state_ws || !dden_ws || !tsum_ws)
        return SA_EINVAL;
    if (int rc = fused_check(B, N, G, m, 64, stride)) return rc;
    if ((attn_stride & 3) || (int64_t)N * attn_stride * 4 >= ((int64_t)1 << 31)) return SA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    LAArgs laa;
    unsigned nla = 0;
    if (la) {   // (checked before the first launch: SA_EUNSUPPORTED leaves nothing behind)
        if ((g_tunables.pp_dbg & 512u) || !state_fwd) return SA_EUNSUPPORTED;   // the co-launch rides on the paired launches (forward states kept): otherwise call sa_local_attn_bwd
        if (int rc = la_from_args(la, B, N, true, laa, nla)) return rc;
    }
    const int64_t rows = (int64_t)B * N * G;
    if ((((uintptr_t)dattn | (uintptr_t)attn) & 15) == 0)
        SA_LAUNCH(favor_fdden_v4_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, st, dattn, attn, attn_stride, G, inv, dden_ws, rows);
    else
        SA_LAUNCH(favor_fdden_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dattn, attn, attn_stride, G, inv, dden_ws, rows);
    SA_CHECK_LAUNCH();
    FusedArgs s = {};
    fused_common(s, tiles, ps, gmax_ws, B, N, G, m, stride);
    const unsigned nblk = (unsigned)((int64_t)B * G * s.S);
    // ---- d loss / d q: scan over j <= i of phi_k(j) (x) v_j  (the forward's states), c_i = dattn_i inv_i, running-sum term d den_i (sum_{j<=i} phi_k(j) + eps)
    s.fa = FeatSrc{k, offk, 1, 0};
    s.fx = FeatSrc{q, offq, 0, 0};
    s.b = v; s.b_stride = stride; s.b_scale = nullptr;
    s.c = dattn; s.c_stride = attn_stride; s.c_scale = inv;
    s.ex_scale = dden_ws; s.zmode = 1; s.ex_const = 1e-6f; s.reverse = 0; s.is_query = 1; s.amx = amq; s.dx = dq; s.tsum = tsum_ws;
    s.dx_lp = (unsigned short*)dq_lp;
    // Independent launches share a grid (favor_fpair_*): [scan B for dq | reversed chunk states] when the forward's states were kept, and [scan B for dk | scan A for dv].
    // SA_PP_DBG bit 9 keeps the five separate launches (A/B runs; results are bit-identical either way).
    const bool pair = !(g_tunables.pp_dbg & 512u);
    const bool seq = fused_seq_states(B, G);
    const unsigned nst = (unsigned)((int64_t)B * G * ((s.LDF + FSLAB - 1) / FSLAB));
    if (state_fwd) s.state = (float*)state_fwd;
    else {
        s.state = state_ws;
        if (seq) {
            SA_LAUNCH(favor_fstate_seq_kernel, dim3(nst), dim3(256), 0, st, s);
            SA_CHECK_LAUNCH();
        } else {
            SA_LAUNCH(favor_fstate_kernel, dim3(nblk), dim3(256), 0, st, s);
            SA_CHECK_LAUNCH();
            if (int rc = fused_prefix(state_ws, B, G, s.S, s.LDF, st)) return rc;
        }
    }
    const FusedArgs sq = s;
    // ---- d loss / d k and d loss / d v: reversed scans over i >= j of phi_q(i) (x) (dattn_i inv_i), running sums weighted by d den_i
    s.fa = FeatSrc{q, offq, 0, 0};
    s.fx = FeatSrc{k, offk, 1, 0};
    s.b = dattn; s.b_stride = attn_stride; s.b_scale = inv;
    s.c = v; s.c_stride = stride; s.c_scale = nullptr;
    s.zmode = 2; s.ex_const = 0.f; s.reverse = 1; s.is_query = 0; s.amx = nullptr; s.dx = dk; s.state = state_ws;
    s.dx_lp = (unsigned short*)dk_lp;
    const FusedArgs sk = s;
    if (pair && state_fwd && seq) {   // [reversed chunk states, sequential | scan B for dq | local dq]: the states leave as exclusive prefixes, no prefix launch
        if (la) SA_LAUNCH(favor_fpair_seq_b_la_kernel, dim3(nst + nblk + nla), dim3(256), 0, st, sk, sq, laa, (int)nst, (int)nblk);
        else SA_LAUNCH(favor_fpair_seq_b_kernel, dim3(nst + nblk), dim3(256), 0, st, sk, sq, (int)nst, (int)nblk);
        SA_CHECK_LAUNCH();
    } else {
        if (pair && state_fwd) {
            if (la) SA_LAUNCH(favor_fpair_b_state_la_kernel, dim3(2 * nblk + nla), dim3(256), 0, st, sq, sk, laa, (int)nblk);
            else SA_LAUNCH(favor_fpair_b_state_kernel, dim3(2 * nblk), dim3(256), 0, st, sq, sk, (int)nblk);
            SA_CHECK_LAUNCH();
        } else {
            SA_LAUNCH(favor_fout_b_kernel, dim3(nblk), dim3(256), 0, st, sq);
            SA_CHECK_LAUNCH();
            if (seq) SA_LAUNCH(favor_fstate_seq_kernel, dim3(nst), dim3(256), 0, st, sk);
            else SA_LAUNCH(favor_fstate_kernel, dim3(nblk), dim3(256), 0, st, sk);
            SA_CHECK_LAUNCH();
        }
        if (!(seq && !(pair && state_fwd)))
            if (int rc = fused_prefix(state_ws, B, G, s.S, s.LDF, st)) return rc;
    }
    // dv_j[d] = sum_m phi_k(j)[m] R_j[m][d]: scan A on the same states (a = phi_q, b = dattn inv, reversed), per-position map phi_k, no normaliser
    s.zmode = 0; s.y = dv; s.y_stride = stride; s.inv_out = nullptr; s.accumulate = 0;
    s.y_lp = (unsigned short*)dv_lp;
    if (pair) {
        if (la) SA_LAUNCH(favor_fpair_b_a_la_kernel, dim3(2 * nblk + nla), dim3(256), 0, st, sk, s, laa, (int)nblk);
        else SA_LAUNCH(favor_fpair_b_a_kernel, dim3(2 * nblk), dim3(256), 0, st, sk, s, (int)nblk);
        SA_CHECK_LAUNCH();
    } else {
        SA_LAUNCH(favor_fout_b_kernel, dim3(nblk), dim3(256), 0, st, sk);
        SA_CHECK_LAUNCH();
    }
    SA_LAUNCH(favor_fkey_fix_kernel, dim3(1), dim3(64), 0, st, dk, (unsigned short*)dk_lp, stride, G, (const unsigned long long*)gmax_ws, tsum_ws, (int)nblk, ps, s.LDF);
    SA_CHECK_LAUNCH();
    if (!pair) {
        SA_LAUNCH(favor_fout_a_kernel, dim3(nblk), dim3(256), 0, st, s);
        SA_CHECK_LAUNCH();
    }
    return 0;
}
