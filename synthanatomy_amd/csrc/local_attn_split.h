// Local-window attention, split-bf16 path: launch arguments, tile helpers and the two kernel BODIES (over a block id and an LDS buffer, so that csrc/favor_fused.hip
// can run them in the same launch as the FAVOR+ chunk kernels they do not depend on); kernels and entry points: csrc/local_attn.hip.
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sa_common.h"
#include "split_bf16.h"

namespace sa {

struct LAArgs {
    const float *q, *k, *v, *out, *dout, *lse_in, *Dbuf_in;
    float *o, *lse_out, *dq, *dk, *dv, *Dbuf_out;
    unsigned short *o_lp, *dv_lp;          // optional bf16 copies of the rows written to o / dv (same strides and offsets)
    int32_t q_stride, q_off, k_stride, k_off, v_stride, v_off, o_stride, o_off;
    int32_t B, N, L, W;
    float scale;
    int32_t use_order;                     // split-bf16 kernels: launch the heaviest tiles first
    uint16_t order_q[256], order_k[256];   // rank -> query tile / key tile
};

__device__ __forceinline__ void la_store_lp(unsigned short* p, const float4 v) {
    uint2 pk;
    pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *(uint2*)p = pk;
}

constexpr int LT = 64;   // tile edge
constexpr int LLD = 68;  // LDS row stride in floats

// ------------------------------------------------------------------------------------------------------------------------------
// split-bf16 path
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

constexpr int LTB = 64 * 128;   // one [64 rows][64 bf16] tile, bytes

// One (batch, head-block) matrix [N rows][stride floats] behind a buffer descriptor: rows beyond N read as zeros (the range check is on
// the per-lane offset), so tile loads need neither clamps nor masks and their addresses are one add per tile.
struct LATile {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[4];     // byte offsets of this thread's four 16-byte pieces of tile 0
    uint32_t step;        // bytes per 64-row tile
    __device__ __forceinline__ void init(const float* src, int stride, int off, int64_t rowbase, int N, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + rowbase * stride), 0, (int)((uint32_t)N * (uint32_t)stride * 4u), 0x00020000);
#pragma unroll
        for (int it = 0; it < 4; ++it) voff[it] = (uint32_t)(((tid >> 4) + 16 * it) * stride + off + (tid & 15) * 4) * 4u;
        step = (uint32_t)stride * 256u;
    }
    __device__ __forceinline__ void load(u32x4 (&v)[4], int tile) const {
        const uint32_t t = (uint32_t)tile * step;
#pragma unroll
        for (int it = 0; it < 4; ++it) v[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it] + t, 0, 0));
    }
};

template <bool SCALE>
__device__ __forceinline__ void la_stage(unsigned char* hi, unsigned char* lo, const u32x4 (&v)[4], float scale, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t o = lroff((tid >> 4) + 16 * it, (tid & 15) * 4);
        float x0 = __uint_as_float(v[it][0]), x1 = __uint_as_float(v[it][1]), x2 = __uint_as_float(v[it][2]), x3 = __uint_as_float(v[it][3]);
        if (SCALE) { x0 *= scale; x1 *= scale; x2 *= scale; x3 *= scale; }
        uint2 h, l;
        split_pair(x0, x1, h.x, l.x);
        split_pair(x2, x3, h.y, l.y);
        *(uint2*)(hi + o) = h;
        *(uint2*)(lo + o) = l;
    }
}

// x or zeros, as a bit mask: a select would let the compiler sink the load under a branch (exec-masked loads join on vmcnt(0))
__device__ __forceinline__ float4 la_keep(float4 x, bool keep) {
    const uint32_t m = keep ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(x.x) & m), __uint_as_float(__float_as_uint(x.y) & m), __uint_as_float(__float_as_uint(x.z) & m),
                       __uint_as_float(__float_as_uint(x.w) & m));
}

// this lane's row of a [rows][64] fp32 matrix as MFMA B operands: d = h2*32 + g*8 .. +7 (row is clamped into the sequence by the caller)
__device__ __forceinline__ void la_row_operand(short8_t (&hi)[2], short8_t (&lo)[2], const float* row, bool valid, float scale, int g) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const float4 x0 = la_keep(*(const float4*)(row + h2 * 32 + g * 8), valid);
        const float4 x1 = la_keep(*(const float4*)(row + h2 * 32 + g * 8 + 4), valid);
        uint32_t h[4], l[4];
        split_pair(x0.x * scale, x0.y * scale, h[0], l[0]);
        split_pair(x0.z * scale, x0.w * scale, h[1], l[1]);
        split_pair(x1.x * scale, x1.y * scale, h[2], l[2]);
        split_pair(x1.z * scale, x1.w * scale, h[3], l[3]);
        hi[h2] = __builtin_bit_cast(short8_t, (u32x4){h[0], h[1], h[2], h[3]});
        lo[h2] = __builtin_bit_cast(short8_t, (u32x4){l[0], l[1], l[2], l[3]});
    }
}

// heaviest tiles first (LAArgs::order): block -> (tile, head, batch)
__device__ __forceinline__ void la_block(const LAArgs& a, const uint16_t* order, int nt, int bid, int& t, int& h, int& b) {
    const int bl = a.B * a.L, rank = bid / bl, bh = bid % bl;
    t = a.use_order ? order[rank] : rank;
    h = bh % a.L;
    b = bh / a.L;
}

// MODE 0: forward (o, lse)   MODE 1: backward wrt q (dq, D)
constexpr int LA_SPLIT_LDS = 4 * LTB + 512;   // bytes of LDS either body needs
template <int MODE>
__device__ __forceinline__ void local_attn_q_split_body(const LAArgs& a, const int bid, unsigned char* const lds) {
    unsigned char* const sKh = lds, * const sKl = lds + LTB, * const sVh = lds + 2 * LTB, * const sVl = lds + 3 * LTB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int nqt = (a.N + LT - 1) / LT;
    int qt, h, b;
    la_block(a, a.order_q, nqt, bid, qt, h, b);
    const int q0 = qt * LT;
    const int iq = q0 + wave * 16 + qi;
    const bool vq = iq < a.N;
    const int64_t rb = (int64_t)b * a.N;
    const int qoff = a.q_off + h * 64, koff = a.k_off + h * 64, voff = a.v_off + h * 64, ooff = a.o_off + h * 64;
    const int64_t qrow = rb + min(iq, a.N - 1);

    const int kt_lo = max(0, (q0 / a.W - 1) * a.W) / LT;
    const int kt_hi = min(a.N - 1, q0 + LT - 1) / LT;
    LATile tK, tV;
    tK.init(a.k, a.k_stride, koff, rb, a.N, tid);
    tV.init(a.v, a.v_stride, voff, rb, a.N, tid);
    u32x4 pk[4], pv[4];
    tK.load(pk, kt_lo);
    tV.load(pv, kt_lo);

    short8_t Qh[2], Ql[2], Gh[2], Gl[2];
    la_row_operand(Qh, Ql, a.q + qrow * a.q_stride + qoff, vq, a.scale, g);
    float lse = 0.f, Dv = 0.f;
    if (MODE == 1) {
        const float* dor = a.dout + qrow * a.o_stride + ooff;
        const float* orow = a.out + qrow * a.o_stride + ooff;
        la_row_operand(Gh, Gl, dor, vq, 1.f, g);
        float part = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const float4 x = *(const float4*)(dor + h2 * 32 + g * 8 + e), y = *(const float4*)(orow + h2 * 32 + g * 8 + e);
                part += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            }
        Dv = vq ? group_sum(part) : 0.f;
        lse = vq ? a.lse_in[qrow * a.L + h] : 0.f;
        if (vq && g == 0) a.Dbuf_out[qrow * a.L + h] = Dv;
    }
    const int lo_i = vq ? max(0, (iq / a.W - 1) * a.W) : a.N;   // invalid queries see no key
    const int j_hi = min(iq, a.N - 1);
    // keys of a tile are visible to every query of this wave (no masks) when they lie between the last query's lower bound and the first query
    const int wq0 = q0 + wave * 16;
    const int free_lo = (wq0 + 15 < a.N) ? max(0, ((wq0 + 15) / a.W - 1) * a.W) : a.N + LT;   // a wave with rows beyond N always masks
    float m_run = -1e30f, l_run = 0.f;
    float4_t acc[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};

    for (int kt = kt_lo; kt <= kt_hi; ++kt) {
        __syncthreads();
        la_stage<false>(sKh, sKl, pk, 1.f, tid);
        la_stage<false>(sVh, sVl, pv, 1.f, tid);
        __syncthreads();
        const int ktn = min(kt + 1, kt_hi);
        tK.load(pk, ktn);
        tV.load(pv, ktn);

        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        tile_rows_gemm(s, sKh, sKl, Qh, Ql, qi, g);                       // S^T = K Q^T
        if (MODE == 1) tile_rows_gemm(dp, sVh, sVl, Gh, Gl, qi, g);       // dP^T = V dO^T
        // lane element (f, r) <-> key j = kt*64 + f*16 + g*4 + r, query iq
        const int jb = kt * LT + g * 4;
        const bool unmasked = kt * LT >= free_lo && kt * LT + LT - 1 <= wq0;   // wave-uniform
        float4_t p[4];
        if (!unmasked) {   // masked scores become -inf: exp() below turns them into exact zeros (the running max is floored at -1e30)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jb + f * 16 + r;
                    s[f][r] = (j >= lo_i && j <= j_hi) ? s[f][r] : -INFINITY;
                }
        }
        if (MODE == 0) {
            float mx = -1e30f;
#pragma unroll
            for (int f = 0; f < 4; ++f) mx = fmaxf(fmaxf(mx, fmaxf(s[f][0], s[f][1])), fmaxf(s[f][2], s[f][3]));
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            float ls = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[f][r] = __expf(s[f][r] - m_new);
                    ls += p[f][r];
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
                for (int r = 0; r < 4; ++r) p[f][r] = __expf(s[f][r] - lse) * (dp[f][r] - Dv);  // dS
        }
        short8_t Ph[2], Pl[2];
        acc_to_operand(Ph, Pl, p);
        if (MODE == 0) tile_cols_gemm(acc, sVh, sVl, Ph, Pl, lane);   // O^T += V^T P^T
        else tile_cols_gemm(acc, sKh, sKl, Ph, Pl, lane);             // dQ^T += K^T dS^T
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

// dk / dv: block = 64 keys of one (batch, head); wave = 16 keys whose K, V rows are the register-resident B operands
__device__ __forceinline__ void local_attn_kv_split_body(const LAArgs& a, const int bid, unsigned char* const lds) {
    unsigned char* const sQh = lds, * const sQl = lds + LTB, * const sGh = lds + 2 * LTB, * const sGl = lds + 3 * LTB;
    float* const sLse = (float*)(lds + 4 * LTB), * const sD = sLse + LT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kc = lane & 15, g = lane >> 4;
    const int nkt = (a.N + LT - 1) / LT;
    int kt, h, b;
    la_block(a, a.order_k, nkt, bid, kt, h, b);
    const int kj = kt * LT + wave * 16 + kc;
    const bool vk = kj < a.N;
    const int64_t rb = (int64_t)b * a.N;
    const int qoff = a.q_off + h * 64, koff = a.k_off + h * 64, voff = a.v_off + h * 64, ooff = a.o_off + h * 64;
    const int64_t krow = rb + min(kj, a.N - 1);

    const int last_key = min(a.N - 1, kt * LT + LT - 1);
    const int hi = min(a.N - 1, (last_key / a.W + 2) * a.W - 1);  // last query that can see a key of this tile
    const int qt_hi = hi / LT;
    LATile tQ, tG;
    tQ.init(a.q, a.q_stride, qoff, rb, a.N, tid);
    tG.init(a.dout, a.o_stride, ooff, rb, a.N, tid);
    u32x4 pq[4], pg[4];
    float plse = 0.f, pD = 0.f;
    const float* lse_p = a.lse_in + rb * a.L + h;
    const float* D_p = a.Dbuf_in + rb * a.L + h;
    auto prefetch = [&](int qt) {
        tQ.load(pq, qt);
        tG.load(pg, qt);
        const int i = min(qt * LT + lane, a.N - 1);
        plse = lse_p[i * a.L];
        pD = D_p[i * a.L];
    };
    prefetch(kt);

    short8_t Kh[2], Kl[2], Vh[2], Vl[2];
    la_row_operand(Kh, Kl, a.k + krow * a.k_stride + koff, vk, 1.f, g);
    la_row_operand(Vh, Vl, a.v + krow * a.v_stride + voff, vk, 1.f, g);
    // query i sees key kj  <=>  kj <= i < (kj / W + 2) W   (and both inside the sequence)
    const int i_lo = vk ? kj : a.N, i_hi = min(a.N - 1, (kj / a.W + 2) * a.W - 1);
    // a query tile needs no masks for this wave when it lies after the wave's last key and before the first key's horizon
    const int wk0 = kt * LT + wave * 16;
    const int free_lo = (wk0 + 15 < a.N) ? wk0 + 15 : a.N + LT;
    const int free_hi = min(a.N - 1, (wk0 / a.W + 2) * a.W - 1);
    float4_t dka[4], dva[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        dka[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
        dva[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    for (int qt = kt; qt <= qt_hi; ++qt) {
        __syncthreads();
        la_stage<true>(sQh, sQl, pq, a.scale, tid);
        la_stage<false>(sGh, sGl, pg, 1.f, tid);
        if (tid < LT) {
            sLse[tid] = plse;
            sD[tid] = pD;
        }
        __syncthreads();
        prefetch(min(qt + 1, qt_hi));

        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        tile_rows_gemm(s, sQh, sQl, Kh, Kl, kc, g);    // S = Q K^T
        tile_rows_gemm(dp, sGh, sGl, Vh, Vl, kc, g);   // dP = dO V^T
        // lane element (f, r) <-> query i = qt*64 + f*16 + g*4 + r, key kj
        const bool unmasked = qt * LT >= free_lo && qt * LT + LT - 1 <= free_hi;   // wave-uniform
        float4_t p[4], ds[4];
        if (!unmasked) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = qt * LT + f * 16 + g * 4 + r;
                    s[f][r] = (i >= i_lo && i <= i_hi) ? s[f][r] : -INFINITY;
                }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float4 l4 = *(const float4*)&sLse[f * 16 + g * 4], d4 = *(const float4*)&sD[f * 16 + g * 4];
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[f][r] = __expf(s[f][r] - lv[r]);
                ds[f][r] = p[f][r] * (dp[f][r] - dv4[r]);
            }
        }
        short8_t Ph[2], Pl[2];
        acc_to_operand(Ph, Pl, p);
        tile_cols_gemm(dva, sGh, sGl, Ph, Pl, lane);   // dV^T += dO^T P
        acc_to_operand(Ph, Pl, ds);
        tile_cols_gemm(dka, sQh, sQl, Ph, Pl, lane);   // dK^T += Q^T dS
    }
    if (!vk) return;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        *(float4*)(a.dv + (rb + kj) * a.v_stride + voff + df * 16 + g * 4) = make_float4(dva[df][0], dva[df][1], dva[df][2], dva[df][3]);
        if (a.dv_lp) la_store_lp(a.dv_lp + (rb + kj) * a.v_stride + voff + df * 16 + g * 4, make_float4(dva[df][0], dva[df][1], dva[df][2], dva[df][3]));
        *(float4*)(a.dk + (rb + kj) * a.k_stride + koff + df * 16 + g * 4) = make_float4(dka[df][0], dka[df][1], dka[df][2], dka[df][3]);
    }
}

static inline bool la_exact() { return dbg(SA_DBG_LOCAL_ATTN_EXACT); }

static inline int fill_la(LAArgs& a, int q_stride, int q_off, int k_stride, int k_off, int v_stride, int v_off, int o_stride, int o_off, int B, int N, int L, int W,
                   int dh) {
    if (dh != 64 || B <= 0 || N <= 0 || L <= 0 || W <= 0) return SA_EUNSUPPORTED;
    if ((q_stride | q_off | k_stride | k_off | v_stride | v_off | o_stride | o_off) & 3) return SA_EINVAL;  // 16-byte rows
    a.q_stride = q_stride; a.q_off = q_off; a.k_stride = k_stride; a.k_off = k_off; a.v_stride = v_stride; a.v_off = v_off; a.o_stride = o_stride; a.o_off = o_off;
    a.B = B; a.N = N; a.L = L; a.W = W;
    a.scale = 1.f / sqrtf((float)dh);
    // buffer descriptors of the split-bf16 kernels address one batch with 32-bit byte offsets
    const int64_t smax = std::max(std::max(q_stride, k_stride), std::max(v_stride, o_stride));
    if ((int64_t)N * smax * 4 >= (int64_t)1 << 31) return SA_EUNSUPPORTED;
    // longest-processing-time-first launch order: blocks walk 1 .. 2W/64 + 1 tiles, the short ones fill the tail
    const int nt = (N + LT - 1) / LT;
    a.use_order = nt <= 256;
    if (a.use_order) {
        std::vector<int> cq(nt), ck(nt);
        for (int t = 0; t < nt; ++t) {
            const int q0 = t * LT;
            cq[t] = std::min(N - 1, q0 + LT - 1) / LT - std::max(0, (q0 / W - 1) * W) / LT + 1;
            const int last_key = std::min(N - 1, q0 + LT - 1);
            ck[t] = std::min(N - 1, (last_key / W + 2) * W - 1) / LT - t + 1;
        }
        std::vector<int> idx(nt);
        for (int t = 0; t < nt; ++t) idx[t] = t;
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return cq[x] > cq[y]; });
        for (int t = 0; t < nt; ++t) a.order_q[t] = (uint16_t)idx[t];
        for (int t = 0; t < nt; ++t) idx[t] = t;
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return ck[x] > ck[y]; });
        for (int t = 0; t < nt; ++t) a.order_k[t] = (uint16_t)idx[t];
    }
    return 0;
}


}  // namespace sa
