// EMA vector quantizer kernels (gfx950) -- replaces Quantizer_impl.forward, reference
// src/networks/vqvae/baseline.py:38-87, and Quantizer.forward's perplexity (:110-120).
//
// sa_vq_assign: one block = 16 latent rows x all K codes.  The 4 waves split the codebook; every wave streams
// 16-code tiles through LDS and evaluates the reference's *expanded* distance  (|x|^2 - 2 x.w) + |w|^2  in fp32 with the
// exact-f32 MFMA (16x16x4: a bitwise fmaf chain in k order), keeping a running (max -d, first index) per lane.  The winner
// is reduced across lanes/waves with a lower-index tie-break (torch.max returns the first maximum).  The same block then
// emits zq_st = (W[idx]-x)+x, the commitment error, and scatter-adds the EMA sufficient statistics
// counts[K], dw[K,D] (fp32 atomics; summed over ranks by one RCCL all-reduce on the host side).
#include "sa_common.h"

namespace sa {

__global__ void vq_wnorm_kernel(const float* __restrict__ cb, int K, int D, float* __restrict__ wn) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int j = 0; j < D; ++j) {
        const float w = cb[(int64_t)k * D + j];
        s = __fadd_rn(s, __fmul_rn(w, w));
    }
    wn[k] = s;
}

// Round 6: the codebook walk is a per-wave software pipeline.  Every wave owns the 16-code tiles wave, wave + 4, ... and a PRIVATE LDS tile, so the loop needs no
// block barrier at all (rounds 1-5 ran two __syncthreads and one exposed L2 round trip per tile: 32 trips x ~3.5 us = 110 us per block, 326 us per launch for
// 1.5 GFLOP); the rows of the next TWO tiles are requested into registers (NV float4 per lane and tile) while the current tile multiplies.  The arithmetic -- one
// exact-f32 MFMA chain in k order per (row, code), the expanded distance, the first-index tie-break -- is unchanged: indices stay bit-identical.
template <int NV>
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ rows, const float* __restrict__ cb, int64_t M, int K, int D,
                                                        int64_t* __restrict__ idx_out, float* __restrict__ zq_st, bf16_t* __restrict__ zq_lp,
                                                        float* __restrict__ counts, float* __restrict__ dw, float* __restrict__ sqerr,
                                                        const float* __restrict__ wnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LD = D + 4;  // padded row stride (floats): conflict-free column reads
    float* sX = (float*)smem;             // [16][LD]
    float* sW = sX + 16 * LD;             // [4 waves][16][LD]
    float* sxx = sW + 4 * 16 * LD;        // [16]
    float* sbest = sxx + 16;              // [4][16]
    int* sbidx = (int*)(sbest + 64);      // [4][16]
    int* sfin = sbidx + 64;               // [16]
    float* sred = (float*)(sfin + 16);    // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int D4 = D >> 2;

    // stage the 16 rows (zero rows beyond M)
    for (int e = tid; e < 16 * D4; e += 256) {
        const int r = e / D4, c4 = e - r * D4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + r < M) v = *(const float4*)(rows + (m0 + r) * D + c4 * 4);
        *(float4*)(sX + r * LD + c4 * 4) = v;
    }
    __syncthreads();
    if (tid < 16) {
        float s = 0.f;
        for (int j = 0; j < D; ++j) s = __fadd_rn(s, __fmul_rn(sX[tid * LD + j], sX[tid * LD + j]));
        sxx[tid] = s;
    }
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    const float xx = sxx[frow];
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    float* myW = sW + wave * 16 * LD;
    const int ntiles = (K + 15) >> 4;
    const int iters = (ntiles + 3) >> 2;
    // this lane's pieces of a tile: float4 e = lane + 64 u of the [16][D4] tile (row e / D4, columns 4 (e % D4) ..)
    int pr[NV], pc[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int e = lane + 64 * u;
        pr[u] = e < 16 * D4 ? e / D4 : -1;
        pc[u] = e < 16 * D4 ? (e - (e / D4) * D4) * 4 : 0;
    }
    auto request = [&](int it, float4 (&v)[NV], float (&wn4)[4]) __attribute__((always_inline)) {
        const int c0 = (it * 4 + wave) * 16;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr[u] >= 0 && c0 + pr[u] < K) v[u] = *(const float4*)(cb + (int64_t)(c0 + pr[u]) * D + pc[u]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wn4[r] = (c0 + fq * 4 + r < K) ? wnorm[c0 + fq * 4 + r] : 0.f;
    };
    auto park = [&](const float4 (&v)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (pr[u] >= 0) *(float4*)(myW + pr[u] * LD + pc[u]) = v[u];
    };
    float4 va[NV], vb[NV];
    float wna[4], wnb[4];
    request(0, va, wna);
    if (iters > 1) request(1, vb, wnb);
    for (int it = 0; it < iters; ++it) {
        // (LDS operations of one wave execute in order: the tile of trip it - 1 has been read before these stores land; no other wave touches myW)
        float wnc[4];
        if (it & 1) {
            park(vb);
#pragma unroll
            for (int r = 0; r < 4; ++r) wnc[r] = wnb[r];
            if (it + 2 < iters) request(it + 2, vb, wnb);
        } else {
            park(va);
#pragma unroll
            for (int r = 0; r < 4; ++r) wnc[r] = wna[r];
            if (it + 2 < iters) request(it + 2, va, wna);
        }
        const int c0 = (it * 4 + wave) * 16;
        if (c0 < K) {
            float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < D4; ++kk) {
                const float a = myW[frow * LD + kk * 4 + fq];
                const float b = sX[frow * LD + kk * 4 + fq];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int code = c0 + fq * 4 + r;
                if (code < K) {
                    const float d = __fadd_rn(__fsub_rn(xx, __fmul_rn(2.f, acc[r])), wnc[r]);
                    const float nd = -d;
                    if (nd > best) {
                        best = nd;
                        bidx = code;
                    }
                }
            }
        }
    }
    // reduce over the 4 lane groups that share a row
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    if (lane < 16) {
        sbest[wave * 16 + lane] = best;
        sbidx[wave * 16 + lane] = bidx;
    }
    __syncthreads();
    if (tid < 16) {
        float b = sbest[tid];
        int bi = sbidx[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = sbest[w * 16 + tid];
            const int oi = sbidx[w * 16 + tid];
            if (ob > b || (ob == b && oi < bi)) {
                b = ob;
                bi = oi;
            }
        }
        if (bi < 0 || bi >= K) bi = 0;  // all-NaN row: torch.max would return the NaN's position; keep in range
        sfin[tid] = (m0 + tid < M) ? bi : -1;
        if (m0 + tid < M) idx_out[m0 + tid] = bi;
    }
    __syncthreads();
    // EMA statistics: the rows of this block that chose the same code are summed HERE and leave as one atomic per (code, dimension) from the first of them (round 6:
    // with a collapsed codebook -- a handful of codes in use, what synthetic volumes train to -- every row of every block added to the same few cache lines, and
    // 11 200 x 32 same-line atomics were ~250 of the launch's 300 us)
    if (tid < 16 && sfin[tid] >= 0) {
        const int bi = sfin[tid];
        bool lead = true;
        for (int r2 = 0; r2 < tid; ++r2) lead = lead && sfin[r2] != bi;
        if (lead) {
            float c = 1.f;
            for (int r2 = tid + 1; r2 < 16; ++r2) c += sfin[r2] == bi ? 1.f : 0.f;
            unsafeAtomicAdd(counts + bi, c);
        }
        sbidx[tid] = lead ? 1 : 0;      // (the wave minima are consumed: their slots carry the leader flags)
    }
    __syncthreads();
    float err = 0.f;
    for (int e = tid; e < 16 * D; e += 256) {
        const int r = e / D, j = e - r * D;
        if (m0 + r < M) {
            const int bi = sfin[r];
            const float x = sX[r * LD + j];
            const float q = cb[(int64_t)bi * D + j];
            const float dqx = __fsub_rn(q, x);
            const int64_t o = (m0 + r) * D + j;
            zq_st[o] = __fadd_rn(dqx, x);
            if (zq_lp) zq_lp[o] = f32_to_bf16(__fadd_rn(dqx, x));
            err += dqx * dqx;
            if (sbidx[r]) {
                float acc = x;
                for (int r2 = r + 1; r2 < 16; ++r2)
                    if (sfin[r2] == bi) acc += sX[r2 * LD + j];
                unsafeAtomicAdd(dw + (int64_t)bi * D + j, acc);
            }
        }
    }
    err = wave_sum(err);
    if (lane == 0) sred[wave] = err;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(sqerr, sred[0] + sred[1] + sred[2] + sred[3]);
}

__global__ __launch_bounds__(1024) void vq_ema_kernel(float* __restrict__ N, float* __restrict__ avg, float* __restrict__ cb,
                                                      const float* __restrict__ counts, const float* __restrict__ dw, int K, int D, float decay,
                                                      float eps) {
    __shared__ float red[16];
    __shared__ float ntot;
    const int tid = threadIdx.x;
    const float omd = 1.f - decay;
    float s = 0.f;
    for (int k = tid; k < K; k += 1024) {
        const float v = __fadd_rn(__fmul_rn(N[k], decay), __fmul_rn(counts[k], omd));
        N[k] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        ntot = t;
    }
    __syncthreads();
    const float n = ntot;
    const float denom = __fadd_rn(n, __fmul_rn((float)K, eps));
    for (int e = tid; e < K * D; e += 1024) {
        const int k = e / D;
        const float wn = __fmul_rn(__fdiv_rn(__fadd_rn(N[k], eps), denom), n);
        const float a = __fadd_rn(__fmul_rn(avg[e], decay), __fmul_rn(dw[e], omd));
        avg[e] = a;
        cb[e] = __fdiv_rn(a, wn);
    }
}

__global__ __launch_bounds__(1024) void vq_perplexity_kernel(const float* __restrict__ counts, int K, float inv_m, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float p = counts[k] * inv_m;
        s += p * logf(p + 1e-10f);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = expf(-t);
    }
}

__global__ void vq_backward_kernel(const float* __restrict__ rows, const float* __restrict__ cb, const int64_t* __restrict__ idx, const void* g_zq,
                                   int g_dtype, const float* __restrict__ g_loss, float coef, int64_t n, int D, void* dz, int dz_dtype) {
    const float gl = g_loss ? g_loss[0] * coef : 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int j = (int)(e - m * D);
        float v = g_zq ? load_as_f32(g_zq, g_dtype, e) : 0.f;
        v += gl * (rows[e] - cb[idx[m] * D + j]);
        store_from_f32(dz, dz_dtype, e, v);
    }
}

__global__ void vq_embed_kernel(const float* __restrict__ cb, const int64_t* __restrict__ idx, int64_t n, int K, int D, void* out, int out_dtype) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int j = (int)(e - m * D);
        int64_t k = idx[m];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        store_from_f32(out, out_dtype, e, cb[k * D + j]);
    }
}

}  // namespace sa

extern "C" int sa_vq_assign(const float* rows, const float* codebook, int64_t M, int K, int D, int64_t* idx, float* zq_st, void* zq_lp,
                            float* counts, float* dw, float* sqerr, float* wnorm, void* stream) {
    using namespace sa;
    if (!rows || !codebook || !idx || !zq_st || !counts || !dw || !sqerr || !wnorm) return SA_EINVAL;
    if (M <= 0 || K <= 0 || D <= 0 || (D & 3) || D > 1024) return SA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SA_LAUNCH(vq_wnorm_kernel, dim3((K + 255) / 256), dim3(256), 0, st, codebook, K, D, wnorm);
    SA_CHECK_LAUNCH();
    const size_t lds = (size_t)(5 * 16 * (D + 4) + 16 + 64 + 64 + 16 + 4) * 4;
    if (lds > 160 * 1024) return SA_EUNSUPPORTED;
    const unsigned nblk = (unsigned)((M + 15) / 16);
    // NV = float4 pieces of a [16 codes][D] tile per lane (two tiles ahead in registers): D <= 16 / 32 / 64 / 128 / 256 / 496
#define SA_VQ_LAUNCH(NV) SA_LAUNCH(vq_assign_kernel<NV>, dim3(nblk), dim3(256), lds, st, rows, codebook, M, K, D, idx, zq_st, (bf16_t*)zq_lp, counts, dw, sqerr, wnorm)
    if (D <= 16) SA_VQ_LAUNCH(1);
    else if (D <= 32) SA_VQ_LAUNCH(2);
    else if (D <= 64) SA_VQ_LAUNCH(4);
    else if (D <= 128) SA_VQ_LAUNCH(8);
    else if (D <= 256) SA_VQ_LAUNCH(16);
    else SA_VQ_LAUNCH(32);      // (D <= 496: wider rows do not fit the 160 KiB of LDS, checked above)
#undef SA_VQ_LAUNCH
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_vq_ema_update(float* N, float* embed_avg, float* codebook, const float* counts, const float* dw, int K, int D, float decay,
                                float eps, void* stream) {
    using namespace sa;
    if (!N || !embed_avg || !codebook || !counts || !dw || K <= 0 || D <= 0) return SA_EINVAL;
    SA_LAUNCH(vq_ema_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, N, embed_avg, codebook, counts, dw, K, D, decay, eps);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_vq_perplexity(const float* counts, int K, int64_t M, float* out, void* stream) {
    using namespace sa;
    if (!counts || !out || K <= 0 || M <= 0) return SA_EINVAL;
    SA_LAUNCH(vq_perplexity_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, K, 1.f / (float)M, out);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_vq_backward(const float* rows, const float* codebook, const int64_t* idx, const void* g_zq, int g_dtype, const float* g_loss,
                              float beta, int64_t M, int D, void* dz, int dz_dtype, void* stream) {
    using namespace sa;
    if (!rows || !codebook || !idx || !dz || M <= 0 || D <= 0) return SA_EINVAL;
    const int64_t n = M * D;
    const float coef = beta * 2.f / (float)n;
    unsigned nblk = (unsigned)((n + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    SA_LAUNCH(vq_backward_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, rows, codebook, idx, g_zq, g_dtype, g_loss, coef, n, D, dz,
                       dz_dtype);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_vq_embed(const float* codebook, const int64_t* idx, int64_t M, int K, int D, void* out, int out_dtype, void* stream) {
    using namespace sa;
    if (!codebook || !idx || !out || M <= 0 || D <= 0) return SA_EINVAL;
    const int64_t n = M * D;
    unsigned nblk = (unsigned)((n + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    SA_LAUNCH(vq_embed_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, codebook, idx, n, K, D, out, out_dtype);
    SA_CHECK_LAUNCH();
    return 0;
}
