// Causal local-window attention on the exact-fp32 MFMA (mfma_f32_16x16x4f32) -- replaces local_attention.LocalAttention
// (window W, look_backward = 1, look_forward = 0, causal) behind performer_pytorch's SelfAttention for the local heads
// (reference src/networks/transformers/performer.py:194-219 with local_attn_heads > 0).  q and k arrive already rotated (sa_rotary).
//
// Query i attends keys j with  max(0, (i/W - 1) W) <= j <= i.  Flash-style tiling, 64 queries x 64 keys per step, fp32 throughout:
//   forward   block = 64 queries of one (batch, head); wave = 16 queries.  S^T = K Q^T (keys on MFMA rows, queries on columns) so that
//             every lane owns ONE query: the online softmax is a per-lane max/sum over its 16 scores plus two cross-group shuffles,
//             and the probabilities feed O^T += V^T P^T straight from the accumulator registers (no transposition through LDS).
//   backward  dq: same walk with dP^T = V dO^T, dS = P (dP - D), dQ^T += K^T dS^T.
//             dk/dv: block = 64 keys (wave = 16 keys held in registers), walking the query tiles that can see them:
//             S = Q K^T, dV^T += dO^T P, dK^T += Q^T dS.
// Alg. FLOPs: 4 * 64 per (query, key) pair forward, 10 * 64 backward.
//
// Two arithmetic paths with the same tiling:
//   split-bf16 (default): every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and each product is evaluated as
//             hi*hi + hi*lo + lo*hi on mfma_f32_16x16x32_bf16 with fp32 accumulation (dropped terms <= 2^-16 |a||b|; measured against
//             the fp64 band reference: ~1e-5 relative, well inside the 1e-3 budget of the path).  Three bf16 MFMAs of 16 cycles cover
//             the 32 reduction steps that cost eight 32-cycle fp32 MFMAs: 5.3 x less matrix time.  Tiles are split once while they
//             are staged into LDS ([row][64] bf16, 128-byte rows, 32-byte chunks XOR-swizzled); row-major operands are fetched with
//             ds_read_b128, the transposed ones (V^T, K^T, Q^T, dO^T) with ds_read_b64_tr_b16 whose 4-row groups line up with the
//             4-row groups of the accumulator layout, so P / dS still go from accumulators to the next MFMA's B operand in registers.
//             The next tile's global loads are issued before the current tile's MFMAs (register prefetch).
//   exact fp32 (SA_LOCAL_ATTN_EXACT=1): mfma_f32_16x16x4f32 throughout, bitwise an fmaf chain.
#include <stdlib.h>

#include <algorithm>
#include <vector>


#include "local_attn_split.h"

namespace sa {


// rows [row0, row0+64) x 64 floats of head block (stride, off) -> LDS tile, zero beyond N, optional scale
__device__ __forceinline__ void la_load_tile(float* dst, const float* src, int stride, int off, int64_t rowbase, int row0, int N, float scale, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it, c4 = tid & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < N) v = *(const float4*)(src + (rowbase + row0 + r) * stride + off + c4 * 4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *(float4*)(dst + r * LLD + c4 * 4) = v;
    }
}

__device__ __forceinline__ bool la_allowed(int i, int j, int W, int N) {
    const int lo = max(0, (i / W - 1) * W);
    return j <= i && j >= lo && j < N && i < N;
}


// MODE 0: forward (o, lse)   MODE 1: backward wrt q (dq, D)
template <int MODE>
__global__ __launch_bounds__(256) void local_attn_q_kernel(const LAArgs a) {
    __shared__ __attribute__((aligned(16))) float sK[LT * LLD], sV[LT * LLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int nqt = (a.N + LT - 1) / LT;
    const int qt = blockIdx.x % nqt, h = (blockIdx.x / nqt) % a.L, b = blockIdx.x / (nqt * a.L);
    const int q0 = qt * LT;
    const int iq = q0 + wave * 16 + qi;
    const bool vq = iq < a.N;
    const int64_t rb = (int64_t)b * a.N;
    const int qoff = a.q_off + h * 64, koff = a.k_off + h * 64, voff = a.v_off + h * 64, ooff = a.o_off + h * 64;

    float Qreg[16], Greg[16];  // B operands: scaled q (and dO in backward) of this lane's query, d = dd*4 + g
    float lse = 0.f, Dv = 0.f;
#pragma unroll
    for (int dd = 0; dd < 16; ++dd) {
        Qreg[dd] = vq ? a.q[(rb + iq) * a.q_stride + qoff + dd * 4 + g] * a.scale : 0.f;
        Greg[dd] = 0.f;
    }
    if (MODE == 1) {
        float part = 0.f;
#pragma unroll
        for (int dd = 0; dd < 16; ++dd) {
            const float dov = vq ? a.dout[(rb + iq) * a.o_stride + ooff + dd * 4 + g] : 0.f;
            const float ov = vq ? a.out[(rb + iq) * a.o_stride + ooff + dd * 4 + g] : 0.f;
            Greg[dd] = dov;
            part += dov * ov;
        }
        Dv = group_sum(part);
        lse = vq ? a.lse_in[(rb + iq) * a.L + h] : 0.f;
        if (vq && g == 0) a.Dbuf_out[(rb + iq) * a.L + h] = Dv;
    }
    float m_run = -1e30f, l_run = 0.f;
    float4_t acc[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int kt_lo = max(0, (q0 / a.W - 1) * a.W) / LT;
    const int kt_hi = min(a.N - 1, q0 + LT - 1) / LT;
    for (int kt = kt_lo; kt <= kt_hi; ++kt) {
        __syncthreads();
        la_load_tile(sK, a.k, a.k_stride, koff, rb, kt * LT, a.N, 1.f, tid);
        la_load_tile(sV, a.v, a.v_stride, voff, rb, kt * LT, a.N, 1.f, tid);
        __syncthreads();
        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                s[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(sK[(f * 16 + qi) * LLD + dd * 4 + g], Qreg[dd], s[f], 0, 0, 0);
                if (MODE == 1) dp[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(sV[(f * 16 + qi) * LLD + dd * 4 + g], Greg[dd], dp[f], 0, 0, 0);
            }
        }
        // lane element (f, r) <-> key j = kt*64 + f*16 + g*4 + r, query iq
        float4_t p[4];
        if (MODE == 0) {
            float mx = -1e30f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (la_allowed(iq, kt * LT + f * 16 + g * 4 + r, a.W, a.N)) mx = fmaxf(mx, s[f][r]);
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            float ls = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = la_allowed(iq, kt * LT + f * 16 + g * 4 + r, a.W, a.N) ? __expf(s[f][r] - m_new) : 0.f;
                    p[f][r] = pv;
                    ls += pv;
                }
            ls = group_sum(ls);
            l_run = l_run * alpha + ls;
            m_run = m_new;
#pragma unroll
            for (int df = 0; df < 4; ++df) acc[df] *= alpha;
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = la_allowed(iq, kt * LT + f * 16 + g * 4 + r, a.W, a.N) ? __expf(s[f][r] - lse) : 0.f;
                    p[f][r] = pv * (dp[f][r] - Dv);  // dS
                }
        }
        const float* sR = MODE == 0 ? sV : sK;  // forward: O^T += V^T P^T ; backward: dQ^T += K^T dS^T
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(sR[(f * 16 + g * 4 + r) * LLD + df * 16 + qi], p[f][r], acc[df], 0, 0, 0);
    }
    if (!vq) return;
    if (MODE == 0) {
        const float inv = 1.f / l_run;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const float4 ov4 = make_float4(acc[df][0] * inv, acc[df][1] * inv, acc[df][2] * inv, acc[df][3] * inv);
            *(float4*)(a.o + (rb + iq) * a.o_stride + ooff + df * 16 + g * 4) = ov4;
            if (a.o_lp) la_store_lp(a.o_lp + (rb + iq) * a.o_stride + ooff + df * 16 + g * 4, ov4);
        }
        if (g == 0) a.lse_out[(rb + iq) * a.L + h] = m_run + __logf(l_run);
    } else {
#pragma unroll
        for (int df = 0; df < 4; ++df)
            *(float4*)(a.dq + (rb + iq) * a.q_stride + qoff + df * 16 + g * 4) =
                make_float4(acc[df][0] * a.scale, acc[df][1] * a.scale, acc[df][2] * a.scale, acc[df][3] * a.scale);
    }
}

// dk / dv: block = 64 keys of one (batch, head); wave = 16 keys (K, V rows in registers as MFMA B operands)
__global__ __launch_bounds__(256) void local_attn_kv_kernel(const LAArgs a) {
    __shared__ __attribute__((aligned(16))) float sQ[LT * LLD], sG[LT * LLD];
    __shared__ float sLse[LT], sD[LT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kc = lane & 15, g = lane >> 4;
    const int nkt = (a.N + LT - 1) / LT;
    const int kt = blockIdx.x % nkt, h = (blockIdx.x / nkt) % a.L, b = blockIdx.x / (nkt * a.L);
    const int kj = kt * LT + wave * 16 + kc;
    const bool vk = kj < a.N;
    const int64_t rb = (int64_t)b * a.N;
    const int qoff = a.q_off + h * 64, koff = a.k_off + h * 64, voff = a.v_off + h * 64, ooff = a.o_off + h * 64;
    float Kreg[16], Vreg[16];
#pragma unroll
    for (int dd = 0; dd < 16; ++dd) {
        Kreg[dd] = vk ? a.k[(rb + kj) * a.k_stride + koff + dd * 4 + g] : 0.f;
        Vreg[dd] = vk ? a.v[(rb + kj) * a.v_stride + voff + dd * 4 + g] : 0.f;
    }
    float4_t dka[4], dva[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        dka[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
        dva[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int last_key = min(a.N - 1, kt * LT + LT - 1);
    const int hi = min(a.N - 1, (last_key / a.W + 2) * a.W - 1);  // last query that can see a key of this tile
    for (int qt = kt; qt <= hi / LT; ++qt) {
        __syncthreads();
        la_load_tile(sQ, a.q, a.q_stride, qoff, rb, qt * LT, a.N, a.scale, tid);
        la_load_tile(sG, a.dout, a.o_stride, ooff, rb, qt * LT, a.N, 1.f, tid);
        if (tid < LT) {
            const int i = qt * LT + tid;
            sLse[tid] = i < a.N ? a.lse_in[(rb + i) * a.L + h] : 0.f;
            sD[tid] = i < a.N ? a.Dbuf_in[(rb + i) * a.L + h] : 0.f;
        }
        __syncthreads();
        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                s[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[(f * 16 + kc) * LLD + dd * 4 + g], Kreg[dd], s[f], 0, 0, 0);
                dp[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(sG[(f * 16 + kc) * LLD + dd * 4 + g], Vreg[dd], dp[f], 0, 0, 0);
            }
        }
        // lane element (f, r) <-> query i = qt*64 + f*16 + g*4 + r, key kj
        float4_t p[4], ds[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = f * 16 + g * 4 + r;
                const float pv = la_allowed(qt * LT + il, kj, a.W, a.N) ? __expf(s[f][r] - sLse[il]) : 0.f;
                p[f][r] = pv;
                ds[f][r] = pv * (dp[f][r] - sD[il]);
            }
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (f * 16 + g * 4 + r) * LLD + df * 16 + kc;
                    dva[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(sG[row], p[f][r], dva[df], 0, 0, 0);
                    dka[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[row], ds[f][r], dka[df], 0, 0, 0);
                }
    }
    if (!vk) return;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        *(float4*)(a.dv + (rb + kj) * a.v_stride + voff + df * 16 + g * 4) = make_float4(dva[df][0], dva[df][1], dva[df][2], dva[df][3]);
        if (a.dv_lp) la_store_lp(a.dv_lp + (rb + kj) * a.v_stride + voff + df * 16 + g * 4, make_float4(dva[df][0], dva[df][1], dva[df][2], dva[df][3]));
        *(float4*)(a.dk + (rb + kj) * a.k_stride + koff + df * 16 + g * 4) = make_float4(dka[df][0], dka[df][1], dka[df][2], dka[df][3]);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 3 : 2) void local_attn_q_split_kernel(const LAArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LA_SPLIT_LDS];
    local_attn_q_split_body<MODE>(a, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(256) void local_attn_kv_split_kernel(const LAArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LA_SPLIT_LDS];
    local_attn_kv_split_body(a, (int)blockIdx.x, lds);
}

}  // namespace sa

using namespace sa;

extern "C" int sa_local_attn_fwd(const float* q, int q_stride, int q_off, const float* k, int k_stride, int k_off, const float* v, int v_stride, int v_off,
                                 float* o, int o_stride, int o_off, float* lse, int B, int N, int L, int W, int dh, void* o_lp, void* stream) {
    if (!q || !k || !v || !o || !lse) return SA_EINVAL;
    LAArgs a = {};
    const int rc = fill_la(a, q_stride, q_off, k_stride, k_off, v_stride, v_off, o_stride, o_off, B, N, L, W, dh);
    if (rc) return rc;
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse_out = lse; a.o_lp = (unsigned short*)o_lp;
    const unsigned nblk = (unsigned)(B * L * ((N + LT - 1) / LT));
    if (la_exact()) SA_LAUNCH(local_attn_q_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    else SA_LAUNCH(local_attn_q_split_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_local_attn_bwd(const float* q, int q_stride, int q_off, const float* k, int k_stride, int k_off, const float* v, int v_stride, int v_off,
                                 const float* out, const float* dout, int o_stride, int o_off, const float* lse, float* dq, float* dk, float* dv,
                                 float* Dbuf, int B, int N, int L, int W, int dh, void* dv_lp, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !Dbuf) return SA_EINVAL;
    LAArgs a = {};
    const int rc = fill_la(a, q_stride, q_off, k_stride, k_off, v_stride, v_off, o_stride, o_off, B, N, L, W, dh);
    if (rc) return rc;
    a.q = q; a.k = k; a.v = v; a.out = out; a.dout = dout; a.lse_in = lse; a.dq = dq; a.dk = dk; a.dv = dv; a.Dbuf_out = Dbuf; a.Dbuf_in = Dbuf; a.dv_lp = (unsigned short*)dv_lp;
    const unsigned nblk = (unsigned)(B * L * ((N + LT - 1) / LT));
    const bool exact = la_exact();
    if (exact) SA_LAUNCH(local_attn_q_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    else SA_LAUNCH(local_attn_q_split_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    if (exact) SA_LAUNCH(local_attn_kv_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    else SA_LAUNCH(local_attn_kv_split_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}
