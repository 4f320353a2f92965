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
// XOR-swizzled with the row so that both the row-major ds_read_b128 fragments and the transposing reads are bank-conflict free
__device__ __forceinline__ uint32_t lroff(uint32_t m, uint32_t c) { return m * 128u + ((((c >> 4) ^ ((m >> 1) & 3u)) << 5) | ((c & 15u) << 1)); }

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

}  // namespace sa
