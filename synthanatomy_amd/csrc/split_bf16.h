// Split-bf16 arithmetic on mfma_f32_16x16x32_bf16: an fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a product is
// evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation (dropped terms <= 2^-16 |a||b|).  Three 16-cycle MFMAs cover the 32 reduction
// steps that cost eight 32-cycle mfma_f32_16x16x4f32: 5.3 x less matrix time at ~1e-5 relative error.  Shared by local_attn.hip and the
// chunked FAVOR+ scans in performer.hip.
#pragma once
#include "sa_common.h"

namespace sa {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef short v4s_t __attribute__((ext_vector_type(4)));

// two fp32 -> one word of two bf16 (round to nearest even) with the gfx950 conversion instruction (v_cvt_pk_bf16_f32): one VALU op where the
// bit-twiddling f32_to_bf16() of sa_common.h costs ~6 per value
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// two values per call, packed as bf16 pairs
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// eight consecutive reduction steps of one lane as MFMA operand words
__device__ __forceinline__ void split8(const float (&x)[8], short8_t& hi, short8_t& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(x[2 * e], x[2 * e + 1], h[e], l[e]);
    hi = __builtin_bit_cast(short8_t, (u32x4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(short8_t, (u32x4){l[0], l[1], l[2], l[3]});
}

// byte offset of column c (bf16 index, multiple of 4) of row m in a [rows][64] bf16 tile with 128-byte rows; 32-byte chunks are
// XOR-swizzled with the row so that both the row-major ds_read_b128 fragments and the transposing reads are bank-conflict free.
// Round 6: rows m and m + 8 also swap the two 16-byte halves of a chunk.  Row parity x the chunk swizzle give eight distinct 32-byte bank groups to eight
// consecutive rows, but the 8-byte reads of tile_rows_gemm_perm put SIXTEEN rows x 16 bytes into one half-wave pass -- rows m and m + 8 met on the same
// sixteen bytes (2-way conflict: 56 % of scan A's LDS cycles, profiles/r04_performer_pmc_lds.txt); with the half swap they use the two halves of the group.
#ifndef SA_LROFF_NO_HALF_SWAP
__device__ __forceinline__ uint32_t lroff(uint32_t m, uint32_t c) {
    return m * 128u + (((((c >> 4) ^ ((m >> 1) & 3u)) << 5) | ((c & 15u) << 1)) ^ ((m & 8u) << 1));
}
#else
__device__ __forceinline__ uint32_t lroff(uint32_t m, uint32_t c) { return m * 128u + ((((c >> 4) ^ ((m >> 1) & 3u)) << 5) | ((c & 15u) << 1)); }
#endif

// ds_read_b64_tr_b16: per 16-lane group a [4 rows][16 columns] bf16 block -> lane s holds column s of the 4 rows.  Lane (group gq, s)
// addresses row 4*gq + (s >> 2), columns 4*(s & 3)..+3 of the block; two reads 16 rows apart make one MFMA operand whose reduction index
// (gq, e) is row (e / 4) * 16 + 4 * gq + e % 4 of a 32-row block -- the order in which 16x16 accumulator fragments hold their rows.
__device__ __forceinline__ v4s_t lds_tr16_b64(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p));
}

// three-term product into one accumulator
__device__ __forceinline__ float4_t mfma3(short8_t ah, short8_t al, short8_t bh, short8_t bl, float4_t acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// accumulator fragments (f, r) <-> tile row f*16 + g*4 + r  ->  B operands of the next GEMM: k = 32-row block h, rows (2h + e/4)*16 + 4g + e%4
__device__ __forceinline__ void acc_to_operand(short8_t (&hi)[2], short8_t (&lo)[2], const float4_t (&p)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float x[8] = {p[2 * h][0], p[2 * h][1], p[2 * h][2], p[2 * h][3], p[2 * h + 1][0], p[2 * h + 1][1], p[2 * h + 1][2], p[2 * h + 1][3]};
        split8(x, hi[h], lo[h]);
    }
}

// ---- GEMM steps over one [64 rows][64 bf16] hi / lo tile pair in the lroff() layout; four accumulator fragments per call ----

// acc[f] += rows (f*16 + lane&15) of the tile . B operand; reduction over the 64 tile columns in natural order: B operand word e of lane
// (n, g) is column ks*32 + g*8 + e.  Fragments via ds_read_b128.
// a_lo_zero (block-uniform): the tile's lo half is known to hold zeros (a bf16-representable operand: the throughput mode's projection matrix) -- its fragment
// reads and the lo*hi products are skipped; adding their exact zeros would not change a bit of the result.
__device__ __forceinline__ void tile_rows_gemm(float4_t (&acc)[4], const unsigned char* tHi, const unsigned char* tLo, const short8_t (&bh)[2],
                                               const short8_t (&bl)[2], int i16, int g, int nf = 4, bool a_lo_zero = false) {
    if (nf == 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            short8_t ah[4], al[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) ah[f] = *(const short8_t*)(tHi + lroff(f * 16 + i16, ks * 32 + g * 8));
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bl[ks], acc[f], 0, 0, 0);
            if (!a_lo_zero) {
#pragma unroll
                for (int f = 0; f < 4; ++f) al[f] = *(const short8_t*)(tLo + lroff(f * 16 + i16, ks * 32 + g * 8));
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], bh[ks], acc[f], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bh[ks], acc[f], 0, 0, 0);
        }
        return;
    }
    // only the first nf output fragments (block-uniform nf < 4): the same per-accumulator product order
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f >= nf) break;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t o = lroff(f * 16 + i16, ks * 32 + g * 8);
            const short8_t ah = *(const short8_t*)(tHi + o);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ks], acc[f], 0, 0, 0);
            if (!a_lo_zero) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const short8_t*)(tLo + o), bh[ks], acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ks], acc[f], 0, 0, 0);
        }
    }
}

// same product with the reduction index in accumulator-row order: word e of lane (n, g) is column ks*32 + (e/4)*16 + 4g + e%4
// (two ds_read_b64 per fragment), so one B operand can serve this GEMM and a tile_cols_gemm over the same index
__device__ __forceinline__ void tile_rows_gemm_perm(float4_t (&acc)[4], const unsigned char* tHi, const unsigned char* tLo, const short8_t (&bh)[2],
                                                    const short8_t (&bl)[2], int i16, int g, int nks = 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (ks >= nks) break;
        short8_t ah[4], al[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint32_t o0 = lroff(f * 16 + i16, ks * 32 + g * 4), o1 = lroff(f * 16 + i16, ks * 32 + 16 + g * 4);
            ah[f] = __builtin_shufflevector(*(const v4s_t*)(tHi + o0), *(const v4s_t*)(tHi + o1), 0, 1, 2, 3, 4, 5, 6, 7);
            al[f] = __builtin_shufflevector(*(const v4s_t*)(tLo + o0), *(const v4s_t*)(tLo + o1), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bl[ks], acc[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], bh[ks], acc[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bh[ks], acc[f], 0, 0, 0);
    }
}

// acc[f] += tile^T (tile columns f*16 + lane&15) . B operand; reduction over the 64 tile ROWS in accumulator-row order (transposing reads)
__device__ __forceinline__ void tile_cols_gemm(float4_t (&acc)[4], const unsigned char* tHi, const unsigned char* tLo, const short8_t (&bh)[2],
                                               const short8_t (&bl)[2], int lane, int nks = 2, int nf = 4, bool a_lo_zero = false) {
    const uint32_t trow = (uint32_t)(lane >> 4) * 4u + ((uint32_t)(lane & 15) >> 2), tcol = (uint32_t)(lane & 3) * 4u;
    if (nf == 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks >= nks) break;
            short8_t ah[4], al[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const uint32_t o0 = lroff(ks * 32 + trow, f * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, f * 16 + tcol);
                ah[f] = __builtin_shufflevector(lds_tr16_b64(tHi + o0), lds_tr16_b64(tHi + o1), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bl[ks], acc[f], 0, 0, 0);
            if (!a_lo_zero) {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const uint32_t o0 = lroff(ks * 32 + trow, f * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, f * 16 + tcol);
                    al[f] = __builtin_shufflevector(lds_tr16_b64(tLo + o0), lds_tr16_b64(tLo + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], bh[ks], acc[f], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], bh[ks], acc[f], 0, 0, 0);
        }
        return;
    }
    // only the first nf output fragments (tile columns [0, 16 nf)); block-uniform nf < 4, the same per-accumulator product order
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f >= nf) break;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks >= nks) break;
            const uint32_t o0 = lroff(ks * 32 + trow, f * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, f * 16 + tcol);
            const short8_t ah = __builtin_shufflevector(lds_tr16_b64(tHi + o0), lds_tr16_b64(tHi + o1), 0, 1, 2, 3, 4, 5, 6, 7);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ks], acc[f], 0, 0, 0);
            if (!a_lo_zero) {
                const short8_t al = __builtin_shufflevector(lds_tr16_b64(tLo + o0), lds_tr16_b64(tLo + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ks], acc[f], 0, 0, 0);
            }
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ks], acc[f], 0, 0, 0);
        }
    }
}

// four 16-byte pieces per thread of a [64][64] fp32 tile (thread = row tid/16 + 16 it, columns 4 (tid%16)..+3) -> hi / lo tiles
__device__ __forceinline__ void tile_stage(unsigned char* hi, unsigned char* lo, const u32x4 (&v)[4], int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t o = lroff((tid >> 4) + 16 * it, (tid & 15) * 4);
        uint2 h, l;
        split_pair(__uint_as_float(v[it][0]), __uint_as_float(v[it][1]), h.x, l.x);
        split_pair(__uint_as_float(v[it][2]), __uint_as_float(v[it][3]), h.y, l.y);
        *(uint2*)(hi + o) = h;
        *(uint2*)(lo + o) = l;
    }
}

}  // namespace sa
