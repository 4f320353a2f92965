// Performer hot-path kernels other than the dense projections (those are 1-tap launches of conv_fprop / conv_wgrad).
//
// Replaces, behind reference src/networks/transformers/performer.py:194-221,229-288, the third-party arithmetic of
// performer-pytorch 1.0.11 (softmax_kernel, causal_linear_attention -> fast_transformers CausalDotProduct CUDA kernel,
// ReZero, FeedForward's GELU), local-attention (rotary + banded causal softmax attention) and torch's embedding /
// LayerNorm / cross_entropy kernels.  Everything here is fp32 (the reference runs this path with amp=False and forces
// fp32 around the FAVOR+ kernel).
//
// FAVOR+ causal attention is split into separable running-state scans so that every kernel has >= 192 independent
// blocks (the per-(batch, head) state S = sum k' (x) v is 266 x 64 fp32):
//   scan A  "reduce over features":  T[m][d] += a_i[m] b_i[d];  y_i[d] = sum_m c_i[m] T[m][d]     (block = 16 columns d)
//   scan B  "reduce over columns" :  T[m][d] += a_i[m] b_i[d];  y_i[m] = sum_d T[m][d] c_i[d]     (block = 64 features m)
// forward numerator = A(k', v, q'); dv = A(q', dnum, k') reversed; dq' = B(k', v, dnum); dk' = B(q', dnum, v) reversed.
#include <stdlib.h>

#include "sa_common.h"
#include "split_bf16.h"

namespace sa {

static inline unsigned grid1d(int64_t n, int block = 256, unsigned cap = 8192) {
    int64_t b = (n + block - 1) / block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------ embeddings
struct EmbedArgs {
    const float* table[6];
    const int64_t* idx[6];
    int32_t per_position[6];  // 1: index by position n = r % N (shared across the batch); 0: index by row r
    int32_t ntab, dim, N;
    int64_t R;
};

__global__ void embed_sum_kernel(const EmbedArgs a, float* __restrict__ out) {
    const int64_t total = a.R * a.dim;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / a.dim;
        const int c = (int)(e - r * a.dim);
        float s = 0.f;
        for (int t = 0; t < a.ntab; ++t) {
            const int64_t ix = a.idx[t][a.per_position[t] ? (r % a.N) : r];
            if (ix >= 0) s += a.table[t][ix * a.dim + c];
        }
        out[e] = s;
    }
}

__global__ void embed_scatter_kernel(const float* __restrict__ dy, float* __restrict__ dtable, const int64_t* __restrict__ idx, int per_position, int dim,
                                     int N, int64_t R) {
    const int64_t total = R * dim;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / dim;
        const int c = (int)(e - r * dim);
        const int64_t ix = idx[per_position ? (r % N) : r];
        if (ix >= 0) unsafeAtomicAdd(dtable + ix * dim + c, dy[e]);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm (one wave per row)
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                     void* y_lp, int lp_dtype, float* __restrict__ stats, int64_t R, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* xr = x + r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / C + eps);
    for (int c = lane; c < C; c += 64) {
        const float o = (xr[c] - mean) * rstd * w[c] + b[c];
        y[r * C + c] = o;
        if (y_lp) store_from_f32(y_lp, lp_dtype, r * C + c, o);
    }
    if (lane == 0) {
        stats[2 * r] = mean;
        stats[2 * r + 1] = rstd;
    }
}

__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                     const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int64_t R,
                                     int C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (x[r * C + c] - mean) * rstd;
        const float g = dy[r * C + c] * w[c];
        s1 += g;
        s2 += g * xh;
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (x[r * C + c] - mean) * rstd;
        const float d = dy[r * C + c];
        dx[r * C + c] = (d * w[c] - s1 - xh * s2) * rstd;
        unsafeAtomicAdd(dw + c, d * xh);
        unsafeAtomicAdd(db + c, d);
    }
}

// C <= 512: a block walks 32 rows (8 per wave), the weight / bias gradient partials stay in registers (columns lane + 64 i) and reach dw / db as ONE
// atomic per column and block (the per-element form issued 2 R C atomics onto 2 C addresses: 370 us for 8 400 x 512)
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ dw,
                                                                 float* __restrict__ db, int64_t R, int C) {
    __shared__ float sw[4][512], sb[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float pw[8], pb[8], wc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pw[i] = 0.f;
        pb[i] = 0.f;
        wc[i] = lane + 64 * i < C ? w[lane + 64 * i] : 0.f;
    }
    for (int k = 0; k < 8; ++k) {
        const int64_t r = (int64_t)blockIdx.x * 32 + wv * 8 + k;
        if (r >= R) break;
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float xh[8], d[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            const bool in = c < C;
            xh[i] = in ? (x[r * C + c] - mean) * rstd : 0.f;
            d[i] = in ? dy[r * C + c] : 0.f;
            const float g = d[i] * wc[i];
            s1 += g;
            s2 += g * xh[i];
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < C) dx[r * C + c] = (d[i] * wc[i] - s1 - xh[i] * s2) * rstd;
            pw[i] += d[i] * xh[i];
            pb[i] += d[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sw[wv][lane + 64 * i] = pw[i];
        sb[wv][lane + 64 * i] = pb[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        unsafeAtomicAdd(dw + c, (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]));
        unsafeAtomicAdd(db + c, (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]));
    }
}

// ------------------------------------------------------------------------------------------------ GELU / ReZero
__global__ void gelu_kernel(const void* u, int u_dtype, void* h, int h_dtype, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        store_from_f32(h, h_dtype, e, gelu_f(load_as_f32(u, u_dtype, e)));
}

// bf16 -> bf16, eight elements (16 bytes) per thread and step
__global__ void gelu_bf16x8_kernel(const uint4* __restrict__ u, uint4* __restrict__ h, int64_t n8) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = u[e];
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            out[i] = (uint32_t)f32_to_bf16(gelu_f(__uint_as_float(in[i] << 16))) | ((uint32_t)f32_to_bf16(gelu_f(__uint_as_float(in[i] & 0xffff0000u))) << 16);
        h[e] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// ReZero forward for bf16 F and a bf16 copy of y, four elements per thread and step
__global__ void rezero_fwd_bf16x4_kernel(const float4* __restrict__ x, const uint2* __restrict__ F, const float* __restrict__ g, float4* __restrict__ y,
                                         uint2* __restrict__ y_lp, int64_t n4) {
    const float gv = g[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 xv = x[e];
        const uint2 f = F[e];
        const float4 o = make_float4(xv.x + gv * __uint_as_float(f.x << 16), xv.y + gv * __uint_as_float(f.x & 0xffff0000u), xv.z + gv * __uint_as_float(f.y << 16),
                                     xv.w + gv * __uint_as_float(f.y & 0xffff0000u));
        y[e] = o;
        if (y_lp) {
            uint2 p;
            p.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
            p.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
            y_lp[e] = p;
        }
    }
}

// y = x + g * F   (ReZero, performer_pytorch.ReZero);  optional low-precision copy of y for the next GEMM
__global__ void rezero_fwd_kernel(const float* __restrict__ x, const void* F, int f_dtype, const float* __restrict__ g, float* __restrict__ y, void* y_lp,
                                  int lp_dtype, int64_t n) {
    const float gv = g[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float o = x[e] + gv * load_as_f32(F, f_dtype, e);
        y[e] = o;
        if (y_lp) store_from_f32(y_lp, lp_dtype, e, o);
    }
}

// dF = g * dy ; dg += sum dy * F
__global__ void rezero_bwd_kernel(const float* __restrict__ dy, const void* F, int f_dtype, const float* __restrict__ g, void* dF, int df_dtype,
                                  float* __restrict__ dg, int64_t n) {
    const float gv = g[0];
    float s = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float d = dy[e];
        s += d * load_as_f32(F, f_dtype, e);
        store_from_f32(dF, df_dtype, e, gv * d);
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(dg, red[0] + red[1] + red[2] + red[3]);
}

// the same for bf16 F / dF, four elements per thread and step (16 + 8 bytes in, 8 bytes out)
__global__ void rezero_bwd_bf16x4_kernel(const float4* __restrict__ dy, const uint2* __restrict__ F, const float* __restrict__ g, uint2* __restrict__ dF,
                                         float* __restrict__ dg, int64_t n4) {
    const float gv = g[0];
    float s = 0.f;
    auto one = [&](const float4 d, const uint2 f, int64_t e) __attribute__((always_inline)) {
        s += d.x * __uint_as_float(f.x << 16) + d.y * __uint_as_float(f.x & 0xffff0000u) + d.z * __uint_as_float(f.y << 16) + d.w * __uint_as_float(f.y & 0xffff0000u);
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(gv * d.x) | ((uint32_t)f32_to_bf16(gv * d.y) << 16);
        o.y = (uint32_t)f32_to_bf16(gv * d.z) | ((uint32_t)f32_to_bf16(gv * d.w) << 16);
        dF[e] = o;
    };
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four positions per trip with all eight loads in flight before the first use: one trip per thread at the Performer's size (a plain grid-stride loop made
    // four dependent HBM round trips: 19 us for 34 MB)
    for (; e + 3 * step < n4; e += 4 * step) {
        float4 d[4];
        uint2 f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = dy[e + u * step];
            f[u] = F[e + u * step];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(d[u], f[u], e + u * step);
    }
    for (; e < n4; e += step) one(dy[e], F[e], e);
    s = wave_sum(s);
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per block: same-address fp32 atomics serialise in L2, so the launch uses few, large blocks (1 024 blocks = 10 us of atomics)
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        unsafeAtomicAdd(dg, t);
    }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] += alpha * x[e];
}

// ------------------------------------------------------------------------------------------------ FAVOR+ feature map
// rows r' = r*G + h (r = b*N + n).  dd [R*G, LDF] = x . (c P)^T from the projection GEMM; x = src[r, (h0+h)*dh .. +dh].
__global__ __launch_bounds__(256) void favor_global_max_kernel(const float* __restrict__ dd, int64_t rows, int m, int LDF, unsigned long long* __restrict__ out) {
    // one wave per group of four rows, 16-byte loads (LDF % 4 == 0), the four rows' loads in flight together; the padding columns >= m are
    // skipped; one atomic per block
    __shared__ unsigned long long sbest[4];
    unsigned long long best = 0ull;
    const int lane = threadIdx.x & 63, nv = LDF >> 2;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4; r0 < rows; r0 += nw * 4) {
        for (int v = lane; v < nv; v += 64) {
            float4 x[4];
            int64_t rr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rr[q] = r0 + q < rows ? r0 + q : rows - 1;   // a repeated row repeats (value, index) pairs: harmless for the maximum
                x[q] = ((const float4*)(dd + rr[q] * LDF))[v];
            }
            const int c = v * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i0 = (uint32_t)(rr[q] * LDF) + (uint32_t)c;
                unsigned long long p;
                if (c < m) { p = pack_max(x[q].x, i0); best = p > best ? p : best; }
                if (c + 1 < m) { p = pack_max(x[q].y, i0 + 1); best = p > best ? p : best; }
                if (c + 2 < m) { p = pack_max(x[q].z, i0 + 2); best = p > best ? p : best; }
                if (c + 3 < m) { p = pack_max(x[q].w, i0 + 3); best = p > best ? p : best; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ot = __shfl_xor(best, o, 64);
        best = ot > best ? ot : best;
    }
    if (lane == 0) sbest[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) best = sbest[q] > best ? sbest[q] : best;
        atomicMax(out, best);
    }
}

// one wave per row: feat = ratio * (exp(dd - |x|^2 c^2/2 - stab) + eps); stab = row max (query) or *gmax (key)
__global__ void favor_feat_fwd_kernel(const float* __restrict__ dd, const float* __restrict__ src, int src_stride, int h0, int G, int dh,
                                      const unsigned long long* __restrict__ gmax, float* __restrict__ feat, int64_t rows, int m, int LDF, float c2half,
                                      float ratio, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t rp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rp >= rows) return;
    const int64_t r = rp / G;
    const int h = (int)(rp - r * G);
    const float* x = src + r * src_stride + (h0 + h) * dh;
    float s = 0.f;
    for (int d = lane; d < dh; d += 64) s += x[d] * x[d];
    const float diag = wave_sum(s) * c2half;
    const float* dr = dd + rp * LDF;
    // the row as 16-byte pieces, both in flight before anything is reduced (LDF <= 512: lane l owns pieces l and 64 + l)
    const int nv = LDF >> 2;
    const bool has0 = lane < nv, has1 = 64 + lane < nv;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 r0 = has0 ? *(const float4*)(dr + lane * 4) : z4, r1 = has1 ? *(const float4*)(dr + 256 + lane * 4) : z4;
    const float v0[4] = {r0.x, r0.y, r0.z, r0.w}, v1[4] = {r1.x, r1.y, r1.z, r1.w};
    float stab;
    if (gmax) {
        stab = unpack_max(*gmax);
    } else {
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mx = (has0 && lane * 4 + e < m) ? fmaxf(mx, v0[e]) : mx;
            mx = (has1 && 256 + lane * 4 + e < m) ? fmaxf(mx, v1[e]) : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        stab = mx;
    }
    float o0[4], o1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = lane * 4 + e < m ? ratio * (expf(v0[e] - diag - stab) + eps) : 0.f;
        o1[e] = 256 + lane * 4 + e < m ? ratio * (expf(v1[e] - diag - stab) + eps) : 0.f;
    }
    if (has0) *(float4*)(feat + rp * LDF + lane * 4) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    if (has1) *(float4*)(feat + rp * LDF + 256 + lane * 4) = make_float4(o1[0], o1[1], o1[2], o1[3]);
}

// backward of the feature map: ddd = e*g (minus the stabiliser path), dsrc[head slice] = -(sum e*g) * c^2 * x
__global__ void favor_feat_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat, const float* __restrict__ dd,
                                      const float* __restrict__ src, int src_stride, int h0, int G, int dh, int is_query, float* __restrict__ ddd,
                                      float* __restrict__ dsrc, float* __restrict__ tsum, int64_t rows, int m, int LDF, float c2, float ratio_eps) {
    const int lane = threadIdx.x & 63;
    const int64_t rp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rp >= rows) return;
    const int64_t r = rp / G;
    const int h = (int)(rp - r * G);
    float t = 0.f, mx = -INFINITY;
    int am = 0;
    for (int c = lane; c < LDF; c += 64) {
        float v = 0.f;
        if (c < m) {
            const float e = feat[rp * LDF + c] - ratio_eps;
            v = e * dfeat[rp * LDF + c];
            const float dv = dd[rp * LDF + c];
            if (dv > mx) {
                mx = dv;
                am = c;
            }
        }
        ddd[rp * LDF + c] = v;
        t += v;
    }
    t = wave_sum(t);
    if (is_query) {
        // stab = dd[argmax]: d feat / d stab = -e  ->  ddd[argmax] -= t   (first maximum, like torch.max)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (om > mx || (om == mx && oa < am)) {
                mx = om;
                am = oa;
            }
        }
        if (lane == 0) ddd[rp * LDF + am] -= t;
    } else if (lane == 0) {
        tsum[rp] = t;  // per-row partial; favor_key_stab_kernel reduces them (67k same-address atomics were 10x the kernel)
    }
    const float* x = src + r * src_stride + (h0 + h) * dh;
    float* dx = dsrc + r * src_stride + (h0 + h) * dh;
    for (int d = lane; d < dh; d += 64) dx[d] = -t * c2 * x[d];   // overwritten: the projection adjoint adds its part to this
}

// the global-max element receives -sum_rows t (d feat / d stab = -e for every element of the key tensor)
__global__ __launch_bounds__(1024) void favor_key_stab_kernel(float* __restrict__ ddd, const unsigned long long* __restrict__ gmax,
                                                              const float* __restrict__ trow, int64_t rows) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 1024) s += trow[r];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        const uint32_t idx = 0xffffffffu - (uint32_t)(*gmax & 0xffffffffull);
        ddd[idx] -= t;
    }
}


// ------------------------------------------------------------------------------------------------ FAVOR+ projection redraw
// gaussian_orthogonal_random_matrix(m, d, scaling=0): each d x d Gaussian block is orthonormalised row by row (modified
// Gram-Schmidt = the Q of a QR up to signs) and row r is rescaled by the norm of an independent Gaussian row.  One wave per block.
__global__ __launch_bounds__(64) void favor_projection_kernel(const float* __restrict__ blocks, const float* __restrict__ rows, float* __restrict__ out,
                                                               int m, int d, int nblk) {
    __shared__ float Q[64][65];
    const int t = threadIdx.x, blk = blockIdx.x % nblk, mat = blockIdx.x / nblk;   // `mat`: one projection matrix per layer
    const float* A = blocks + (size_t)blockIdx.x * d * d;
    rows += (size_t)mat * m * d;
    out += (size_t)mat * m * d;
    for (int j = 0; j < d; ++j) {
        float v = t < d ? A[j * d + t] : 0.f;
        for (int i = 0; i < j; ++i) {
            const float qi = Q[i][t];
            const float dot = wave_sum(qi * v);
            v -= dot * qi;
        }
        const float nrm = sqrtf(wave_sum(v * v));
        Q[j][t] = t < d ? v / nrm : 0.f;
        __syncthreads();
    }
    for (int j = 0; j < d; ++j) {
        const int r = blk * d + j;
        if (r >= m) break;
        const float g = t < d ? rows[(size_t)r * d + t] : 0.f;
        const float mult = sqrtf(wave_sum(g * g));
        if (t < d) out[(size_t)r * d + t] = mult * Q[j][t];
    }
}

// ------------------------------------------------------------------------------------------------ FAVOR+ scans
struct ScanArgs {
    const float* a;       // [B,N,G,LDF]
    const float* c_feat;  // scan A: c [B,N,G,LDF]
    const float* b;       // [B*N, b_stride] column block at b_off + g*dv
    const float* c_col;   // scan B: c [B*N, c_stride] column block at c_off + g*dv
    const float* b_scale; // optional [B,N,G] multiplies b_i
    const float* c_scale; // optional [B,N,G] multiplies c_i (scan B)
    float* y;             // scan A: [B*N, y_stride] at y_off + g*dv ; scan B: [B,N,G,LDF]
    const float* y_scale; // scan A optional [B,N,G] multiplies y_i
    const float* ex_scale;  // scan B optional: y_i[m] += ex_scale_i * (ex_vec_i[m] + ex_const)   (ex_scale NULL -> 1)
    const float* ex_vec;    // scan B optional [B,N,G,LDF]
    float ex_const;
    int32_t B, N, G, LDF, dv, b_stride, b_off, c_stride, c_off, y_stride, y_off, reverse, accumulate;
    // segment parallelism: the N positions are cut into S segments scanned by independent blocks.
    // pass 0: one segment, no state buffer.  pass 1: accumulate only the segment's state sum into `state`.
    // (scan_state_prefix_kernel turns the sums into exclusive prefixes.)  pass 2: start from the prefix and emit y.
    float* state;         // [B, G, S, LDF, dv (+1 when zmode)]
    int32_t pass, S, seg_len;
    // chunked MFMA path only: the running column sums z = sum_j a_j w_j ride along as one extra state column, which removes the separate
    // cumsum / normaliser passes:  zmode 1: w = 1;   scan A: y_i /= c_i . (z_i + den_eps)  (inv written to inv_out);
    //                                               scan B: y_i[m] += ex_scale_i * (z_i[m] + ex_const)
    //                             zmode 2: w = ex_scale;  scan B: y_i[m] += z_i[m]
    int32_t zmode;
    int32_t zcol;          // the state buffer has the extra column (stride LDF * dv + LDF); implied by zmode, may also be set alone to READ such a buffer
    int32_t state_ready;   // host side: `state` already holds the exclusive chunk prefixes of exactly this (a, b, b_scale, reverse): skip both passes
    float den_eps;
    float* inv_out;
    int32_t exact;         // host side: exact-fp32 MFMA kernels instead of the split-bf16 ones (state_flags bit 2)
};

constexpr int SCAN_TB = 8;  // positions staged per barrier

// scan A: block = (b, g, 16-column slice); thread = (column dl = t&15, feature group mg = t>>4), features mg + 16 r
__global__ __launch_bounds__(256) void favor_scan_a_kernel(const ScanArgs s) {
    constexpr int NR = 17;  // LDF <= 272
    __shared__ float sa_[SCAN_TB][272], sc_[SCAN_TB][272], sb_[SCAN_TB][16], sp_[SCAN_TB][16][17];
    const int nsl = s.dv / 16;
    const int seg = blockIdx.x % s.S, bx = blockIdx.x / s.S;
    const int sl = bx % nsl, g = (bx / nsl) % s.G, b = bx / (nsl * s.G);
    const int t = threadIdx.x, dl = t & 15, mg = t >> 4;
    float* st = s.state ? s.state + (((int64_t)b * s.G + g) * s.S + seg) * s.LDF * s.dv : nullptr;
    float T[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) T[r] = (s.pass == 2 && mg + 16 * r < s.LDF) ? st[(mg + 16 * r) * s.dv + sl * 16 + dl] : 0.f;
    const int p0 = seg * s.seg_len, p1 = min(s.N, p0 + s.seg_len);
    for (int i0 = p0; i0 < p1; i0 += SCAN_TB) {
        const int nb = min(SCAN_TB, p1 - i0);
        for (int e = t; e < nb * s.LDF; e += 256) {
            const int k = e / s.LDF, c = e - k * s.LDF;
            const int i = s.reverse ? s.N - 1 - (i0 + k) : i0 + k;
            const int64_t row = ((int64_t)b * s.N + i) * s.G + g;
            sa_[k][c] = s.a[row * s.LDF + c];
            if (s.pass != 1) sc_[k][c] = s.c_feat[row * s.LDF + c];
        }
        if (t < nb * 16) {
            const int k = t >> 4, d = t & 15;
            const int i = s.reverse ? s.N - 1 - (i0 + k) : i0 + k;
            const int64_t r = (int64_t)b * s.N + i;
            float v = s.b[r * s.b_stride + s.b_off + g * s.dv + sl * 16 + d];
            if (s.b_scale) v *= s.b_scale[r * s.G + g];
            sb_[k][d] = v;
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const float bv = sb_[k][dl];
            float p = 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int mrow = mg + 16 * r;
                if (mrow < s.LDF) {
                    T[r] = fmaf(sa_[k][mrow], bv, T[r]);
                    p = fmaf(sc_[k][mrow], T[r], p);
                }
            }
            sp_[k][mg][dl] = p;
        }
        __syncthreads();
        if (s.pass != 1 && t < nb * 16) {
            const int k = t >> 4, d = t & 15;
            float y = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) y += sp_[k][q][d];
            const int i = s.reverse ? s.N - 1 - (i0 + k) : i0 + k;
            const int64_t r = (int64_t)b * s.N + i;
            if (s.y_scale) y *= s.y_scale[r * s.G + g];
            float* yp = s.y + r * s.y_stride + s.y_off + g * s.dv + sl * 16 + d;
            if (s.accumulate) *yp += y;
            else *yp = y;
        }
        __syncthreads();
    }
    if (s.pass == 1) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (mg + 16 * r < s.LDF) st[(mg + 16 * r) * s.dv + sl * 16 + dl] = T[r];
    }
}

// scan B: block = (b, g, 64-feature slice); thread = (feature ml = t>>2, column quarter dq = t&3)
__global__ __launch_bounds__(256) void favor_scan_b_kernel(const ScanArgs s) {
    __shared__ float sa_[SCAN_TB][64], sb_[SCAN_TB][64], sc_[SCAN_TB][64];
    const int nsl = (s.LDF + 63) / 64;
    const int seg = blockIdx.x % s.S, bx = blockIdx.x / s.S;
    const int sl = bx % nsl, g = (bx / nsl) % s.G, b = bx / (nsl * s.G);
    const int t = threadIdx.x, ml = t >> 2, dq = t & 3;
    const int mrow = sl * 64 + ml;
    float* st = s.state ? s.state + (((int64_t)b * s.G + g) * s.S + seg) * s.LDF * s.dv : nullptr;
    float T[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) T[j] = (s.pass == 2 && mrow < s.LDF) ? st[mrow * s.dv + dq * 16 + j] : 0.f;
    const int p0 = seg * s.seg_len, p1 = min(s.N, p0 + s.seg_len);
    for (int i0 = p0; i0 < p1; i0 += SCAN_TB) {
        const int nb = min(SCAN_TB, p1 - i0);
        for (int e = t; e < nb * 64; e += 256) {
            const int k = e >> 6, c = e & 63;
            const int i = s.reverse ? s.N - 1 - (i0 + k) : i0 + k;
            const int64_t r = (int64_t)b * s.N + i;
            const int64_t row = r * s.G + g;
            sa_[k][c] = (sl * 64 + c < s.LDF) ? s.a[row * s.LDF + sl * 64 + c] : 0.f;
            float bv = s.b[r * s.b_stride + s.b_off + g * s.dv + c];
            if (s.b_scale) bv *= s.b_scale[row];
            sb_[k][c] = bv;
            float cv = 0.f;
            if (s.pass != 1) {
                cv = s.c_col[r * s.c_stride + s.c_off + g * s.dv + c];
                if (s.c_scale) cv *= s.c_scale[row];
            }
            sc_[k][c] = cv;
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const float av = sa_[k][ml];
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                T[j] = fmaf(av, sb_[k][dq * 16 + j], T[j]);
                p = fmaf(T[j], sc_[k][dq * 16 + j], p);
            }
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            if (s.pass != 1 && dq == 0 && mrow < s.LDF) {
                const int i = s.reverse ? s.N - 1 - (i0 + k) : i0 + k;
                const int64_t row = ((int64_t)b * s.N + i) * s.G + g;
                if (s.ex_vec) p += (s.ex_scale ? s.ex_scale[row] : 1.f) * (s.ex_vec[row * s.LDF + mrow] + s.ex_const);
                s.y[row * s.LDF + mrow] = p;
            }
        }
        __syncthreads();
    }
    if (s.pass == 1 && mrow < s.LDF) {
#pragma unroll
        for (int j = 0; j < 16; ++j) st[mrow * s.dv + dq * 16 + j] = T[j];
    }
}

// ------------------------------------------------------------------------------------------------ FAVOR+ scans on the fp32 MFMA
// Chunked form of the same scans (64 positions per chunk, every chunk an independent block):
//   1) chunk state sums      U_c = A_c^T B_c                    (favor_chunk_state_kernel)
//   2) exclusive prefix over the chunks of one (batch, head)   (scan_state_prefix_kernel)  ->  T_prev(c)
//   3) chunk outputs         scan A: y^T[d][i] = sum_m T_prev[m][d] c_i[m] + sum_{j<=i} b_j[d] (a_j . c_i)
//                            scan B: y^T[m][i] = sum_d T_prev[m][d] c_i[d] + sum_{j<=i} a_j[m] (b_j . c_i)
// Wave w of a block owns positions i = 16 w + (lane & 15) (MFMA columns), so the masked pair products P[j][i] leave the first MFMA
// already in B-operand form for the second one.  All fp32 (mfma_f32_16x16x4f32).
__device__ __forceinline__ int scan_pos(const ScanArgs& s, int p) { return s.reverse ? s.N - 1 - p : p; }

// Loads are unconditional on clamped addresses and masked by a select afterwards: a conditional load becomes a branch whose join
// waits for vmcnt(0), which serialises every load of the loop.  LDF % 16 == 0 (the host checks), so rows are float4-addressable.
// The MFMA k index of a feature contraction is the permutation m = g4 * (LDF/4) + mm, so each lane reads contiguous floats.
__global__ __launch_bounds__(256) void favor_chunk_state_kernel(const ScanArgs s) {
    // wave w owns features [64 w, 64 w + 64) for ALL 64 value columns: the MFMA row index i is the permutation m = 64 w + 4 i + e, so one
    // 16-byte load per position feeds four fragments (e = 0..3) and the feature rows are read once per block (they were read by all four
    // waves, 4 bytes at a time).  The features beyond 256 (LDF = 272: one fragment) stay with the old mapping, value fragment w.
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const int mlo = 64 * w + fr * 4;                       // this lane's 4 features
    const bool has4 = mlo + 3 < s.LDF, hast = 256 + fr < s.LDF;
    float4_t acc[4][4], acct = (float4_t){0.f, 0.f, 0.f, 0.f};
    float4_t accz[4], acczt = (float4_t){0.f, 0.f, 0.f, 0.f};   // zmode: column 0 of an extra value fragment = sum_j a_j w_j
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        accz[e] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[e][df] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 2
    for (int kk = 0; kk < 16; ++kk) {
        const int p = chunk * 64 + kk * 4 + g4;
        const bool ok = p < s.N;
        const int i = scan_pos(s, ok ? p : 0);
        const int64_t r = (int64_t)b * s.N + i, row = r * s.G + g;
        const float bs = ok ? (s.b_scale ? s.b_scale[row] : 1.f) : 0.f;
        const float* bp = s.b + r * s.b_stride + s.b_off + g * s.dv + fr;
        float bd[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) bd[df] = bp[df * 16] * bs;
        const float* ap = s.a + row * s.LDF;
        float4 a4 = *(const float4*)(ap + (has4 ? mlo : 0));
        if (!has4) a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float at = hast ? ap[256 + fr] : 0.f;
        const float ae[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int df = 0; df < 4; ++df) acc[e][df] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], bd[df], acc[e][df], 0, 0, 0);
        acct = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bd[w], acct, 0, 0, 0);
        if (s.zmode) {
            const float wz = (ok && fr == 0) ? (s.zmode == 2 ? s.ex_scale[row] : 1.f) : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) accz[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], wz, accz[e], 0, 0, 0);
            acczt = __builtin_amdgcn_mfma_f32_16x16x4f32(at, wz, acczt, 0, 0, 0);
        }
    }
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    float* st = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    if (s.zmode && fr == 0) {
        float* zp = st + (int64_t)s.LDF * s.dv;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 64 * w + (g4 * 4 + r) * 4 + e;
                if (m < s.LDF && m < 256) zp[m] = accz[e][r];
            }
        if (w == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (256 + g4 * 4 + r < s.LDF) zp[256 + g4 * 4 + r] = acczt[r];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 64 * w + (g4 * 4 + r) * 4 + e;
            if (m < s.LDF && m < 256) {
#pragma unroll
                for (int df = 0; df < 4; ++df) st[m * s.dv + df * 16 + fr] = acc[e][df][r];
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 256 + g4 * 4 + r;
        if (m < s.LDF) st[m * s.dv + w * 16 + fr] = acct[r];
    }
}


// ------------------------------------------------------------------------------------------------ chunked scans on split-bf16 MFMA
// Same three-launch scheme (chunk state sums -> exclusive prefix -> chunk outputs), with every product evaluated as hi*hi + hi*lo + lo*hi
// on mfma_f32_16x16x32_bf16 (split_bf16.h) and every operand staged ONCE per block in LDS as bf16 hi / lo tiles: the fp32 kernels above feed
// their MFMAs straight from global memory, each of the four waves re-reading the whole feature chunk.  Feature matrices are staged in slabs of
// 144 features (9 MFMA fragments, 288-byte rows: the transposing reads of 16 rows land on all 64 banks exactly twice); value tiles use the
// swizzled 128-byte rows of lroff().  Rows outside [0, N) come back as zeros from the buffer descriptor, in both scan directions.
constexpr int SLAB = 144;
constexpr int SLAB_RS = SLAB * 2;            // bytes per staged row
constexpr int SLAB_BYTES = 64 * SLAB_RS;     // one of hi / lo, 64 positions
constexpr int VT_BYTES = 64 * 128;           // [64 positions][64 values] bf16

__device__ __forceinline__ __amdgpu_buffer_rsrc_t scan_rsrc(const float* base, int64_t elems) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(elems * 4), 0x00020000);
}
__device__ __forceinline__ float scan_ld1(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ int scan_row(const ScanArgs& s, int p) { return s.reverse ? s.N - 1 - p : p; }   // may leave [0, N): reads zeros

// features [slab0, slab0 + 144) of the 64 positions of a chunk of one (batch, head) -> hi / lo tiles (columns beyond LDF hold junk nobody reads)
__device__ __forceinline__ void scan_stage_slab(unsigned char* hi, unsigned char* lo, __amdgpu_buffer_rsrc_t rs, const ScanArgs& s, int g, int chunk,
                                                int slab0, int tid) {
    u32x4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int idx = tid + 256 * t, j = idx / 36, c4 = idx - j * 36;
        const int i = scan_row(s, chunk * 64 + j);
        v[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)((i * s.G + g) * s.LDF + slab0 + c4 * 4) * 4u, 0, 0));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int idx = tid + 256 * t, j = idx / 36, c4 = idx - j * 36;
        uint2 h, l;
        split_pair(__uint_as_float(v[t][0]), __uint_as_float(v[t][1]), h.x, l.x);
        split_pair(__uint_as_float(v[t][2]), __uint_as_float(v[t][3]), h.y, l.y);
        *(uint2*)(hi + j * SLAB_RS + c4 * 8) = h;
        *(uint2*)(lo + j * SLAB_RS + c4 * 8) = l;
    }
}

// value rows (column block at off) of the 64 positions of a chunk, times an optional per-position scale -> swizzled hi / lo tiles
__device__ __forceinline__ void scan_stage_values(unsigned char* hi, unsigned char* lo, __amdgpu_buffer_rsrc_t rv, int stride, int off,
                                                  const float* scale, __amdgpu_buffer_rsrc_t rsc, const ScanArgs& s, int g, int chunk, int tid) {
    u32x4 v[4];
    float sc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = (tid >> 4) + 16 * it;
        const int i = scan_row(s, chunk * 64 + j);
        v[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (uint32_t)(i * stride + off + (tid & 15) * 4) * 4u, 0, 0));
        sc[it] = scale ? scan_ld1(rsc, (uint32_t)(i * s.G + g) * 4u) : 1.f;
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

// one MFMA operand (reduction over the 32 tile rows of block ks) for tile columns col0 + lane&15, through the transposing read
__device__ __forceinline__ short8_t scan_tr_operand(const unsigned char* t, uint32_t o0, uint32_t o1) {
    return __builtin_shufflevector(lds_tr16_b64(t + o0), lds_tr16_b64(t + o1), 0, 1, 2, 3, 4, 5, 6, 7);
}

// U_c[m][d] = sum_{j in chunk} a_j[m] (b_j[d] bs_j)   (+ the running-sum column z[m] = sum_j a_j[m] w_j)
__global__ __launch_bounds__(256) void favor_chunk_state_split_kernel(const ScanArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char sAh[SLAB_BYTES], sAl[SLAB_BYTES], sBh[VT_BYTES], sBl[VT_BYTES];
    __shared__ float sW[64];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const __amdgpu_buffer_rsrc_t ra = scan_rsrc(s.a + (int64_t)b * s.N * s.G * s.LDF, (int64_t)s.N * s.G * s.LDF);
    const __amdgpu_buffer_rsrc_t rb = scan_rsrc(s.b + (int64_t)b * s.N * s.b_stride, (int64_t)s.N * s.b_stride);
    const __amdgpu_buffer_rsrc_t rsc = scan_rsrc((s.b_scale ? s.b_scale : s.a) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
    scan_stage_values(sBh, sBl, rb, s.b_stride, s.b_off + g * s.dv, s.b_scale, rsc, s, g, chunk, tid);
    if (s.zmode && tid < 64) {
        const int p = chunk * 64 + tid, i = scan_row(s, p);
        float wv = p < s.N ? 1.f : 0.f;
        if (s.zmode == 2) {
            const __amdgpu_buffer_rsrc_t rex = scan_rsrc(s.ex_scale + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
            wv = scan_ld1(rex, (uint32_t)(i * s.G + g) * 4u);
        }
        sW[tid] = wv;
    }
    __syncthreads();
    const uint32_t trow = (uint32_t)g4 * 4u + ((uint32_t)fr >> 2), tcol = (uint32_t)(fr & 3) * 4u;
    short8_t bh[4][2], bl[4][2], wh[2], wl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const uint32_t o0 = lroff(ks * 32 + trow, df * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, df * 16 + tcol);
            bh[df][ks] = scan_tr_operand(sBh, o0, o1);
            bl[df][ks] = scan_tr_operand(sBl, o0, o1);
        }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (s.zmode && fr == 0) ? sW[ks * 32 + (e >> 2) * 16 + g4 * 4 + (e & 3)] : 0.f;
        split8(x, wh[ks], wl[ks]);
    }
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    float* st = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    float* zp = st + (int64_t)s.LDF * s.dv;
    for (int slab0 = 0; slab0 < s.LDF; slab0 += SLAB) {
        if (slab0) __syncthreads();
        scan_stage_slab(sAh, sAl, ra, s, g, chunk, slab0, tid);
        __syncthreads();
        const int nfr = min(SLAB, s.LDF - slab0) >> 4;
        for (int f = w; f < nfr; f += 4) {
            float4_t acc[4], accz = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t o0 = (ks * 32 + trow) * SLAB_RS + (f * 16 + tcol) * 2, o1 = o0 + 16 * SLAB_RS;
                const short8_t ah = scan_tr_operand(sAh, o0, o1), al = scan_tr_operand(sAl, o0, o1);
#pragma unroll
                for (int df = 0; df < 4; ++df) acc[df] = mfma3(ah, al, bh[df][ks], bl[df][ks], acc[df]);
                if (s.zmode) accz = mfma3(ah, al, wh[ks], wl[ks], accz);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = slab0 + f * 16 + g4 * 4 + r;
#pragma unroll
                for (int df = 0; df < 4; ++df) st[m * s.dv + df * 16 + fr] = acc[df][r];
                if (s.zmode && fr == 0) zp[m] = accz[r];
            }
        }
    }
}


// One 64-wide slab of the chunk's operands, loaded a slab ahead into registers: A = features [slab0, +64) of the 64 positions (rows = positions),
// T = rows [slab0, +64) of the chunk's exclusive-prefix state (64 value columns); both read zeros outside their matrices.
struct ScanSlabRegs {
    u32x4 a[4], t[4];
};
__device__ __forceinline__ void scan_slab_load(ScanSlabRegs& r, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rt, const ScanArgs& s, int g, int chunk,
                                               int slab0, int tid, bool zero_cols) {
    const int col = slab0 + (tid & 15) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = (tid >> 4) + 16 * it;
        const int i = scan_row(s, chunk * 64 + j);
        // feature columns beyond LDF belong to the next row: forced out of range when they take part in a reduction (zero_cols)
        const uint32_t oa = (zero_cols && col >= s.LDF) ? 0xfffffff0u : (uint32_t)((i * s.G + g) * s.LDF + col) * 4u;
        r.a[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, oa, 0, 0));
        r.t[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, (uint32_t)((slab0 + j) * s.dv + (tid & 15) * 4) * 4u, 0, 0));
    }
}

// scan B outputs:  y_i[m] = sum_d T_prev[m][d] c_i[d] + sum_{j <= i} a_j[m] (b_j . c_i + E[j][i])  (+ the extra terms of ScanArgs)
__global__ __launch_bounds__(256, 3) void favor_chunk_out_b_split_kernel(const ScanArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char sBh[VT_BYTES], sBl[VT_BYTES], sAh[VT_BYTES], sAl[VT_BYTES];
    unsigned char* const sTh = sBh;   // the value tile is dead once the pair products exist: the state slabs take its place
    unsigned char* const sTl = sBl;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const __amdgpu_buffer_rsrc_t ra = scan_rsrc(s.a + (int64_t)b * s.N * s.G * s.LDF, (int64_t)s.N * s.G * s.LDF);
    const __amdgpu_buffer_rsrc_t rb = scan_rsrc(s.b + (int64_t)b * s.N * s.b_stride, (int64_t)s.N * s.b_stride);
    const __amdgpu_buffer_rsrc_t rc = scan_rsrc(s.c_col + (int64_t)b * s.N * s.c_stride, (int64_t)s.N * s.c_stride);
    const __amdgpu_buffer_rsrc_t rsc = scan_rsrc((s.b_scale ? s.b_scale : s.a) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
    const __amdgpu_buffer_rsrc_t rcs = scan_rsrc((s.c_scale ? s.c_scale : s.a) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
    const __amdgpu_buffer_rsrc_t rex = scan_rsrc((s.ex_scale ? s.ex_scale : s.a) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const __amdgpu_buffer_rsrc_t rt = scan_rsrc(st0, (int64_t)s.LDF * s.dv);
    ScanSlabRegs pre;
    scan_slab_load(pre, ra, rt, s, g, chunk, 0, tid, false);
    scan_stage_values(sBh, sBl, rb, s.b_stride, s.b_off + g * s.dv, s.b_scale, rsc, s, g, chunk, tid);

    const int pi = chunk * 64 + w * 16 + fr;
    const bool vi = pi < s.N;
    const int ri = scan_row(s, pi);
    const int64_t rowi = ((int64_t)b * s.N + (vi ? ri : 0)) * s.G + g;
    short8_t Ch[2], Cl[2];   // c_i (times c_scale) as B operand, natural order d = ks*32 + g4*8 + e
    {
        const float cs = s.c_scale ? scan_ld1(rcs, (uint32_t)(ri * s.G + g) * 4u) : 1.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (uint32_t)(ri * s.c_stride + s.c_off + g * s.dv + ks * 32 + g4 * 8 + q * 4) * 4u, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) x[q * 4 + e] = __uint_as_float(v[e]) * cs;
            }
            split8(x, Ch[ks], Cl[ks]);
        }
    }
    const float exs = (vi && (s.ex_vec || s.zmode == 1)) ? (s.ex_scale ? s.ex_scale[rowi] : 1.f) : 0.f;
    __syncthreads();
    float4_t P[4];   // P[j][i] = b_j . c_i (+ E), masked to j <= i; rows beyond N are zero in sB but E is not
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) P[jf] = (float4_t){0.f, 0.f, 0.f, 0.f};
    tile_rows_gemm(P, sBh, sBl, Ch, Cl, fr, g4);
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = jf * 16 + g4 * 4 + r, pj = chunk * 64 + jl;
            float e = 0.f;
            if (s.zmode == 1) e = vi ? exs : 0.f;
            else if (s.zmode == 2) e = scan_ld1(rex, (uint32_t)(scan_row(s, pj) * s.G + g) * 4u);
            P[jf][r] = (jl > w * 16 + fr || pj >= s.N) ? 0.f : P[jf][r] + e;
        }
    short8_t Ph[2], Pl[2];
    acc_to_operand(Ph, Pl, P);

    const float* evp = s.ex_vec ? s.ex_vec + rowi * s.LDF + g4 * 4 : nullptr;
    const float* zpp = s.zmode ? st0 + (int64_t)s.LDF * s.dv + g4 * 4 : nullptr;   // running sums of the chunks before this one
    const float zf = s.zmode == 1 ? exs : 1.f;
    const float cadd = s.zmode == 1 ? exs * s.ex_const : 0.f;
    float* yp = s.y + rowi * s.LDF + g4 * 4;
    for (int slab0 = 0; slab0 < s.LDF; slab0 += 64) {
        __syncthreads();   // previous slab consumed (first pass: the pair products have read the value tile)
        tile_stage(sAh, sAl, pre.a, tid);
        tile_stage(sTh, sTl, pre.t, tid);
        __syncthreads();
        if (slab0 + 64 < s.LDF) scan_slab_load(pre, ra, rt, s, g, chunk, slab0 + 64, tid, false);
        float4_t acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        tile_rows_gemm(acc, sTh, sTl, Ch, Cl, fr, g4);   // inter-chunk: T_prev c_i
        tile_cols_gemm(acc, sAh, sAl, Ph, Pl, lane);     // intra-chunk: sum_j a_j[m] P[j][i]
        if (vi) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int m0 = slab0 + f * 16;   // + g4*4 folded into the pointers
                if (m0 < s.LDF) {
                    float4 o = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
                    if (evp) {
                        const float4 ev = *(const float4*)(evp + m0);
                        o.x += exs * (ev.x + s.ex_const); o.y += exs * (ev.y + s.ex_const); o.z += exs * (ev.z + s.ex_const); o.w += exs * (ev.w + s.ex_const);
                    }
                    if (zpp) {
                        const float4 zv = *(const float4*)(zpp + m0);
                        o.x += zf * zv.x + cadd; o.y += zf * zv.y + cadd; o.z += zf * zv.z + cadd; o.w += zf * zv.w + cadd;
                    }
                    *(float4*)(yp + m0) = o;
                }
            }
        }
    }
}

// scan A outputs:  y_i[d] = sum_m T_prev[m][d] c_i[m] + sum_{j <= i} b_j[d] (a_j . c_i)   (zmode 1: divided by c_i . (z_i + eps))
__global__ __launch_bounds__(256, 2) void favor_chunk_out_a_split_kernel(const ScanArgs s) {
    __shared__ __attribute__((aligned(16))) unsigned char sBh[VT_BYTES], sBl[VT_BYTES], sAh[VT_BYTES], sAl[VT_BYTES], sTh[VT_BYTES], sTl[VT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const __amdgpu_buffer_rsrc_t ra = scan_rsrc(s.a + (int64_t)b * s.N * s.G * s.LDF, (int64_t)s.N * s.G * s.LDF);
    const __amdgpu_buffer_rsrc_t rcf = scan_rsrc(s.c_feat + (int64_t)b * s.N * s.G * s.LDF, (int64_t)s.N * s.G * s.LDF);
    const __amdgpu_buffer_rsrc_t rb = scan_rsrc(s.b + (int64_t)b * s.N * s.b_stride, (int64_t)s.N * s.b_stride);
    const __amdgpu_buffer_rsrc_t rsc = scan_rsrc((s.b_scale ? s.b_scale : s.a) + (int64_t)b * s.N * s.G, (int64_t)s.N * s.G);
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const __amdgpu_buffer_rsrc_t rt = scan_rsrc(st0, (int64_t)s.LDF * s.dv);
    const __amdgpu_buffer_rsrc_t rz = scan_rsrc(st0 + (int64_t)s.LDF * s.dv, s.zmode == 1 ? s.LDF : 0);
    const int pi = chunk * 64 + w * 16 + fr;
    const bool vi = pi < s.N;
    const int ri = scan_row(s, pi);
    const int64_t rowi = ((int64_t)b * s.N + (vi ? ri : 0)) * s.G + g;
    // this lane's c_i in accumulator-row order: word e of reduction block ks is feature slab0 + ks*32 + (e/4)*16 + 4 g4 + e%4
    const uint32_t cbase = (uint32_t)((ri * s.G + g) * s.LDF) * 4u;
    u32x4 pc[4], pz[4];
    auto load_c = [&](int slab0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = slab0 + q * 16 + g4 * 4;
            const bool in = m < s.LDF;
            pc[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rcf, in ? cbase + (uint32_t)m * 4u : 0xfffffff0u, 0, 0));
            pz[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (uint32_t)m * 4u, 0, 0));
        }
    };
    ScanSlabRegs pre;
    scan_slab_load(pre, ra, rt, s, g, chunk, 0, tid, true);
    load_c(0);
    scan_stage_values(sBh, sBl, rb, s.b_stride, s.b_off + g * s.dv, s.b_scale, rsc, s, g, chunk, tid);

    float4_t P[4], acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        P[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        acc[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    float den = 0.f;
    for (int slab0 = 0; slab0 < s.LDF; slab0 += 64) {
        if (slab0) __syncthreads();
        tile_stage(sAh, sAl, pre.a, tid);
        tile_stage(sTh, sTl, pre.t, tid);
        short8_t Ch[2], Cl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = __uint_as_float(pc[ks * 2 + (e >> 2)][e & 3]);
                den = fmaf(x[e], __uint_as_float(pz[ks * 2 + (e >> 2)][e & 3]) + s.den_eps, den);
            }
            split8(x, Ch[ks], Cl[ks]);
        }
        __syncthreads();
        const int nks = (min(64, s.LDF - slab0) + 31) >> 5;
        if (slab0 + 64 < s.LDF) {
            scan_slab_load(pre, ra, rt, s, g, chunk, slab0 + 64, tid, true);
            load_c(slab0 + 64);
        }
        tile_rows_gemm_perm(P, sAh, sAl, Ch, Cl, fr, g4, nks);   // pair products a_j . c_i
        tile_cols_gemm(acc, sTh, sTl, Ch, Cl, lane, nks);        // inter-chunk: T_prev^T c_i
    }
    // mask to j <= i (positions beyond N carry zero features)
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (jf * 16 + g4 * 4 + r > w * 16 + fr) P[jf][r] = 0.f;
    float inv_n = 1.f;
    if (s.zmode == 1) {   // den_i = c_i . (z_prev + eps) + sum_{j <= i in chunk} a_j . c_i
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
    tile_cols_gemm(acc, sBh, sBl, Ph, Pl, lane);   // intra-chunk: sum_j b_j[d] P[j][i]
    if (!vi) return;
    const float ys = s.zmode == 1 ? inv_n : (s.y_scale ? s.y_scale[rowi] : 1.f);
    float* yp = s.y + ((int64_t)b * s.N + ri) * s.y_stride + s.y_off + g * s.dv;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        float4 o = make_float4(acc[df][0] * ys, acc[df][1] * ys, acc[df][2] * ys, acc[df][3] * ys);
        float4* d4 = (float4*)(yp + df * 16 + g4 * 4);
        if (s.accumulate) {
            const float4 old = *d4;
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *d4 = o;
    }
}

// b tile of one chunk (with its per-position scale), zero beyond N
__device__ __forceinline__ void scan_load_b_tile(const ScanArgs& s, float (*sB)[68], int b, int g, int chunk, int tid) {
    for (int e = tid; e < 64 * 16; e += 256) {
        const int j = e >> 4, c4 = e & 15;
        const int p = chunk * 64 + j;
        const bool ok = p < s.N;
        const int64_t r = (int64_t)b * s.N + scan_pos(s, ok ? p : 0);
        float4 v = *(const float4*)(s.b + r * s.b_stride + s.b_off + g * s.dv + c4 * 4);
        const float sc = ok ? (s.b_scale ? s.b_scale[r * s.G + g] : 1.f) : 0.f;
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        *(float4*)&sB[j][c4 * 4] = v;
    }
}

__global__ __launch_bounds__(256) void favor_chunk_out_a_kernel(const ScanArgs s) {
    __shared__ float sB[64][68];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const int q4 = s.LDF >> 4;  // float4 groups per lane quarter: features g4*(LDF/4) + [0, LDF/4)
    const int span = s.LDF >> 2;
    scan_load_b_tile(s, sB, b, g, chunk, tid);
    const int pi = chunk * 64 + w * 16 + fr;    // this lane's position (scan order)
    const bool vi = pi < s.N;
    const int64_t ri = (int64_t)b * s.N + scan_pos(s, vi ? pi : 0), rowi = ri * s.G + g;
    float Creg[68];
    {
        const float4* cp = (const float4*)(s.c_feat + rowi * s.LDF + g4 * span);
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            const float4 v = cp[t < q4 ? t : 0];
            const bool k = vi && t < q4;
            Creg[t * 4 + 0] = k ? v.x : 0.f; Creg[t * 4 + 1] = k ? v.y : 0.f; Creg[t * 4 + 2] = k ? v.z : 0.f; Creg[t * 4 + 3] = k ? v.w : 0.f;
        }
    }
    __syncthreads();
    // P[j][i] = a_j . c_i for the 64 positions j of the chunk, masked to j <= i
    float4_t P[4];
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
        P[jf] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const int pj = chunk * 64 + jf * 16 + fr;
        const bool vj = pj < s.N;
        const float4* ap = (const float4*)(s.a + (((int64_t)b * s.N + scan_pos(s, vj ? pj : 0)) * s.G + g) * s.LDF + g4 * span);
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            const float4 v = ap[t < q4 ? t : 0];
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, Creg[t * 4 + 0], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, Creg[t * 4 + 1], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, Creg[t * 4 + 2], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, Creg[t * 4 + 3], P[jf], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (!vj || jf * 16 + g4 * 4 + r > w * 16 + fr) P[jf][r] = 0.f;
    }
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const float* st = st0 + (int64_t)g4 * span * s.dv + fr;
    // zmode 1: normaliser  den_i = c_i . (z_prev + sum_{j <= i in chunk} a_j + den_eps)  from the masked pair products already in P
    float inv_n = 1.f;
    if (s.zmode == 1) {
        const float4* zp = (const float4*)(st0 + (int64_t)s.LDF * s.dv + g4 * span);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            const float4 zv = zp[t < q4 ? t : 0];
            part = fmaf(Creg[t * 4 + 0], zv.x + s.den_eps, fmaf(Creg[t * 4 + 1], zv.y + s.den_eps, fmaf(Creg[t * 4 + 2], zv.z + s.den_eps, fmaf(Creg[t * 4 + 3], zv.w + s.den_eps, part))));
        }
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) part += (P[jf][0] + P[jf][1]) + (P[jf][2] + P[jf][3]);
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        inv_n = 1.f / part;
        if (vi && g4 == 0 && s.inv_out) s.inv_out[rowi] = inv_n;
    }
    float4_t acc[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < 68; ++mm) {  // inter-chunk: T_prev^T c_i   (Creg is zero beyond span, the row index is only clamped)
        float tv[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) tv[df] = st[(mm < span ? mm : 0) * s.dv + df * 16];
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[df], Creg[mm], acc[df], 0, 0, 0);
    }
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int jf = 0; jf < 4; ++jf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[jf * 16 + g4 * 4 + r][df * 16 + fr], P[jf][r], acc[df], 0, 0, 0);
    if (!vi) return;
    const float ys = s.zmode == 1 ? inv_n : (s.y_scale ? s.y_scale[rowi] : 1.f);
    float* yp = s.y + ri * s.y_stride + s.y_off + g * s.dv;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        float4 o = make_float4(acc[df][0] * ys, acc[df][1] * ys, acc[df][2] * ys, acc[df][3] * ys);
        float4* d4 = (float4*)(yp + df * 16 + g4 * 4);
        if (s.accumulate) {
            const float4 old = *d4;
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *d4 = o;
    }
}

__global__ __launch_bounds__(256) void favor_chunk_out_b_kernel(const ScanArgs s) {
    __shared__ float sB[64][68];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, g4 = lane >> 4;
    const int chunk = blockIdx.x % s.S, g = (blockIdx.x / s.S) % s.G, b = blockIdx.x / (s.S * s.G);
    const int nmf = s.LDF >> 4;
    scan_load_b_tile(s, sB, b, g, chunk, tid);
    const int pi = chunk * 64 + w * 16 + fr;
    const bool vi = pi < s.N;
    const int64_t ri = (int64_t)b * s.N + scan_pos(s, vi ? pi : 0), rowi = ri * s.G + g;
    float Creg[16];  // k index of the d contraction: d = g4 * 16 + dd
    {
        const float cs = vi ? (s.c_scale ? s.c_scale[rowi] : 1.f) : 0.f;
        const float4* cp = (const float4*)(s.c_col + ri * s.c_stride + s.c_off + g * s.dv + g4 * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = cp[t];
            Creg[t * 4 + 0] = v.x * cs; Creg[t * 4 + 1] = v.y * cs; Creg[t * 4 + 2] = v.z * cs; Creg[t * 4 + 3] = v.w * cs;
        }
    }
    __syncthreads();
    float4_t P[4];  // P[j][i] = b_j . c_i, masked to j <= i (rows beyond N are zero in sB)
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
        P[jf] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v = *(const float4*)&sB[jf * 16 + fr][g4 * 16 + t * 4];
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, Creg[t * 4 + 0], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, Creg[t * 4 + 1], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, Creg[t * 4 + 2], P[jf], 0, 0, 0);
            P[jf] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, Creg[t * 4 + 3], P[jf], 0, 0, 0);
        }
        // zmode: the running column sums enter as an additive term of the pair products: y_i[m] = sum_{j <= i} a_j[m] (b_j . c_i + E[j][i]),
        // E = ex_scale_i (mode 1: ex_scale_i * cumsum(a)_i) or ex_scale_j (mode 2: cumsum(a * ex_scale)_i)
        if (s.zmode == 1) {
            const float ev = vi ? s.ex_scale[rowi] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) P[jf][r] += ev;
        } else if (s.zmode == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pj = chunk * 64 + jf * 16 + g4 * 4 + r;
                const int64_t rj = ((int64_t)b * s.N + scan_pos(s, pj < s.N ? pj : 0)) * s.G + g;
                P[jf][r] += pj < s.N ? s.ex_scale[rj] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (jf * 16 + g4 * 4 + r > w * 16 + fr) P[jf][r] = 0.f;
    }
    const int64_t zs = (int64_t)s.LDF * s.dv + (s.zcol ? s.LDF : 0);
    const float* st0 = s.state + (((int64_t)b * s.G + g) * s.S + chunk) * zs;
    const float* st = st0 + g4 * 16;
    // a rows of the 16 positions j = jf*16 + g4*4 + r this lane feeds as the MFMA k index (clamped; P is zero for j beyond N)
    const float* arow[16];
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pj = chunk * 64 + jf * 16 + g4 * 4 + r;
            arow[jf * 4 + r] = s.a + (((int64_t)b * s.N + scan_pos(s, pj < s.N ? pj : 0)) * s.G + g) * s.LDF + fr;
            if (pj >= s.N) P[jf][r] = 0.f;
        }
    const float exs = (vi && (s.ex_vec || s.zmode == 1)) ? (s.ex_scale ? s.ex_scale[rowi] : 1.f) : 0.f;
    const float* evp = s.ex_vec ? s.ex_vec + rowi * s.LDF + g4 * 4 : nullptr;
    const float* zpp = s.zmode ? st0 + (int64_t)s.LDF * s.dv + g4 * 4 : nullptr;   // running sums of the chunks before this one
    const float zf = s.zmode == 1 ? exs : 1.f;
    float* yp = s.y + rowi * s.LDF + g4 * 4;
    for (int mf = 0; mf < nmf; ++mf) {
        float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
        float av[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) av[q] = arow[q][mf * 16];
        const float4* tp = (const float4*)(st + (int64_t)(mf * 16 + fr) * s.dv);
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // inter-chunk: T_prev c_i
            const float4 v = tp[t];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, Creg[t * 4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, Creg[t * 4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, Creg[t * 4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, Creg[t * 4 + 3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int jf = 0; jf < 4; ++jf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jf * 4 + r], P[jf][r], acc, 0, 0, 0);
        if (vi) {
            float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (evp) {
                const float4 ev = *(const float4*)(evp + mf * 16);
                o.x += exs * (ev.x + s.ex_const); o.y += exs * (ev.y + s.ex_const); o.z += exs * (ev.z + s.ex_const); o.w += exs * (ev.w + s.ex_const);
            }
            if (zpp) {
                const float4 zv = *(const float4*)(zpp + mf * 16);
                const float cadd = s.zmode == 1 ? exs * s.ex_const : 0.f;
                o.x += zf * zv.x + cadd; o.y += zf * zv.y + cadd; o.z += zf * zv.z + cadd; o.w += zf * zv.w + cadd;
            }
            *(float4*)(yp + mf * 16) = o;
        }
    }
}

// state[b,g,seg] <- sum of the states of the segments before it (exclusive prefix along S), one thread per element
__global__ void scan_state_prefix_kernel(float* __restrict__ state, int64_t BG, int S, int64_t elems) {
    const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= BG * elems) return;
    const int64_t bg = tix / elems, e = tix - bg * elems;
    float* p = state + bg * S * elems + e;
    float acc = 0.f;
    for (int k = 0; k < S; ++k) {
        const float v = p[k * elems];
        p[k * elems] = acc;
        acc += v;
    }
}

// running (or reverse-running) sum along N of x[b,n,g,:] * scale[b,n,g]
// pass 1 (segsum != NULL, out == NULL): per-segment totals.  pass 2: each segment starts from the sum of the totals before it.
__global__ void cumsum_rows_kernel(const float* __restrict__ x, const float* __restrict__ scale, float* __restrict__ out, float* __restrict__ segsum,
                                   int B, int N, int G, int LDF, int reverse, int S, int seg_len) {
    const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= (int64_t)B * G * LDF * S) return;
    const int c = (int)(tix % LDF);
    const int seg = (int)((tix / LDF) % S);
    const int g = (int)((tix / ((int64_t)LDF * S)) % G);
    const int b = (int)(tix / ((int64_t)LDF * S * G));
    float* ss = segsum ? segsum + (((int64_t)b * G + g) * S) * LDF + c : nullptr;
    float acc = 0.f;
    if (out && ss)
        for (int k = 0; k < seg; ++k) acc += ss[(int64_t)k * LDF];
    const int p0 = seg * seg_len, p1 = min(N, p0 + seg_len);
    for (int k = p0; k < p1; ++k) {
        const int i = reverse ? N - 1 - k : k;
        const int64_t row = ((int64_t)b * N + i) * G + g;
        float v = x[row * LDF + c];
        if (scale) v *= scale[row];
        acc += v;
        if (out) out[row * LDF + c] = acc;
    }
    if (!out) ss[(int64_t)seg * LDF] = acc;
}

// den[row] = sum_{c<m} q[row][c] * (z[row][c] + eps) ; inv[row] = 1/den    (one wave per row)
__global__ void favor_den_kernel(const float* __restrict__ q, const float* __restrict__ z, float eps, float* __restrict__ inv, int64_t rows, int m,
                                 int LDF) {
    const int lane = threadIdx.x & 63;
    const int64_t rp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rp >= rows) return;
    float s = 0.f;
    for (int c = lane; c < m; c += 64) s += q[rp * LDF + c] * (z[rp * LDF + c] + eps);
    s = wave_sum(s);
    if (lane == 0) inv[rp] = 1.f / s;
}

// dden[row] = -(dout . out) * invden  over the head's dv columns (one wave per (row, head))
__global__ void favor_dden_kernel(const float* __restrict__ dout, const float* __restrict__ out, int stride, int off, int G, int dv,
                                  const float* __restrict__ inv, float* __restrict__ dden, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t rp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rp >= rows) return;
    const int64_t r = rp / G;
    const int g = (int)(rp - r * G);
    float s = 0.f;
    for (int d = lane; d < dv; d += 64) s += dout[r * stride + off + g * dv + d] * out[r * stride + off + g * dv + d];
    s = wave_sum(s);
    if (lane == 0) dden[rp] = -s * inv[rp];
}

// ------------------------------------------------------------------------------------------------ rotary (local heads)
// x [R, stride] head block at off + h*dh; table [N, dh] (cos | sin).  mode 0: y = x cos + rot(x) sin ; mode 1: transpose
// thread = four consecutive dimensions of the first half of one head row and their partners in the second half (16-byte accesses)
__global__ void rotary_kernel(const float* __restrict__ x, int stride, int off, int L, int dh, const float* __restrict__ cosb,
                              const float* __restrict__ sinb, float* __restrict__ y, int y_stride, int y_off, int N, int64_t R, int mode,
                              int accumulate, int ngroups, int64_t x_goff, int64_t y_goff, unsigned short* __restrict__ y_lp) {
    // ngroups > 1: the same rotation for several operands in one launch (q and k): group gi reads at x + gi * x_goff and writes at y + gi * y_goff
    const int half = dh / 2, q4 = half / 4;
    const int64_t per = R * L * q4, total = per * ngroups;
    const bool small = total < ((int64_t)1 << 31);       // index arithmetic in 32 bits (five run-time divisions per element: the 64-bit forms cost more than the loads)
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += (int64_t)gridDim.x * blockDim.x) {
        int gi, j, h, n;
        int64_t r;
        if (small) {
            const uint32_t u0 = (uint32_t)e0, up = (uint32_t)per;
            const uint32_t ug = u0 / up, ue = u0 - ug * up;
            const uint32_t urh = ue / (uint32_t)q4, uj = ue - urh * (uint32_t)q4;
            const uint32_t ur = urh / (uint32_t)L, uh = urh - ur * (uint32_t)L;
            gi = (int)ug; j = (int)uj; h = (int)uh; r = ur; n = (int)(ur % (uint32_t)N);
        } else {
            gi = (int)(e0 / per);
            const int64_t e = e0 - gi * per;
            j = (int)(e % q4);
            const int64_t rh = e / q4;
            h = (int)(rh % L);
            r = rh / L;
            n = (int)(r % N);
        }
        x += gi * x_goff;
        y += gi * y_goff;
        const int d = j * 4;
        const float* xr = x + r * stride + off + h * dh;
        const float4 xl = *(const float4*)(xr + d), xh = *(const float4*)(xr + d + half);
        const float4 cl = *(const float4*)(cosb + n * dh + d), ch = *(const float4*)(cosb + n * dh + d + half);
        const float4 sl = *(const float4*)(sinb + n * dh + d), sh = *(const float4*)(sinb + n * dh + d + half);
        float4 ol, oh;
        if (mode == 0) {   // y = x cos + rot(x) sin,  rot(x)_d = -x_{d+half} (d < half), x_{d-half} otherwise
            ol = make_float4(xl.x * cl.x - xh.x * sl.x, xl.y * cl.y - xh.y * sl.y, xl.z * cl.z - xh.z * sl.z, xl.w * cl.w - xh.w * sl.w);
            oh = make_float4(xh.x * ch.x + xl.x * sh.x, xh.y * ch.y + xl.y * sh.y, xh.z * ch.z + xl.z * sh.z, xh.w * ch.w + xl.w * sh.w);
        } else {           // adjoint: y_d = g_d cos_d + (rot^T (g sin))_d ;  rot^T(u)_d = u_{d+half} for d < half, -u_{d-half} otherwise
            ol = make_float4(xl.x * cl.x + xh.x * sh.x, xl.y * cl.y + xh.y * sh.y, xl.z * cl.z + xh.z * sh.z, xl.w * cl.w + xh.w * sh.w);
            oh = make_float4(xh.x * ch.x - xl.x * sl.x, xh.y * ch.y - xl.y * sl.y, xh.z * ch.z - xl.z * sl.z, xh.w * ch.w - xl.w * sl.w);
        }
        float* yp = y + r * y_stride + y_off + h * dh + d;
        if (accumulate) {
            const float4 a = *(const float4*)yp, b = *(const float4*)(yp + half);
            ol.x += a.x; ol.y += a.y; ol.z += a.z; ol.w += a.w;
            oh.x += b.x; oh.y += b.y; oh.z += b.z; oh.w += b.w;
        }
        *(float4*)yp = ol;
        *(float4*)(yp + half) = oh;
        if (y_lp) {   // bf16 copy at the same element offsets (the operand of the next dense layer)
            unsigned short* lp = y_lp + (gi * y_goff + r * y_stride + y_off + h * dh + d);
            uint2 pl, ph;
            pl.x = (uint32_t)f32_to_bf16(ol.x) | ((uint32_t)f32_to_bf16(ol.y) << 16);
            pl.y = (uint32_t)f32_to_bf16(ol.z) | ((uint32_t)f32_to_bf16(ol.w) << 16);
            ph.x = (uint32_t)f32_to_bf16(oh.x) | ((uint32_t)f32_to_bf16(oh.y) << 16);
            ph.y = (uint32_t)f32_to_bf16(oh.z) | ((uint32_t)f32_to_bf16(oh.w) << 16);
            *(uint2*)lp = pl;
            *(uint2*)(lp + half) = ph;
        }
        x -= gi * x_goff;
        y -= gi * y_goff;
    }
}

// ================================================================================================ stateful (O(N)) decoding, section 8(f) N3
// One new position per call; `pos` is a DEVICE integer (index of the position being produced) so the whole per-token step can be
// captured once in a HIP graph and replayed.  The results equal the reference's O(N^2) loop (a full forward over the growing prefix,
// transformer.py:58-101) up to fp32 rounding: the FAVOR+ key stabiliser is the running maximum of the prefix -- the keys' global max of
// performer-pytorch 1.0.11 -- and the state keeps the exp part and the +eps part of phi(k) apart so that it can be rescaled when that
// maximum grows:  phi(k_j) = ratio * (exp(a_j - diag_j - s) + eps)  =>  sum_j phi(k_j) v_j = ratio * (E + eps * 1 (x) V1),
//   E = sum_j exp(a_j - diag_j - s) (x) v_j,  Ez = sum_j exp(a_j - diag_j - s),  V1 = sum_j v_j,  and s -> s' multiplies E, Ez by exp(s - s').
__global__ void embed_step_kernel(const EmbedArgs a, const int* __restrict__ pos, int B, float* __restrict__ out) {
    const int p = *pos;
    const int64_t total = (int64_t)B * a.dim;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / a.dim), c = (int)(e % a.dim);
        float acc = 0.f;
        for (int t = 0; t < a.ntab; ++t) {
            const int64_t ix = a.idx[t][a.per_position[t] ? p : b];
            if (ix >= 0) acc += a.table[t][ix * a.dim + c];
        }
        out[e] = acc;
    }
}

// ---- decode step, the decision for ALL rows in one launch (networks/transformers/transformer.choose_next + the sequence update of Performer._sample_stateful;
// reference transformer.py:11-17,42-54): logits / temperature -> optional top-k cut (everything below the k-th largest value of the row becomes -inf; ties with
// it stay, as `out[out < v[:, [-1]]] = -inf` keeps them) -> softmax -> categorical draw by inverse CDF with the caller's uniform u[b]
// (the first token of non-zero probability whose inclusive cdf reaches u * total -- with exact sums this is the torch rule index = #{v : cdf[v] < u * total};
// a target that falls between two threads' differently associated partial sums yields the neighbouring token, never a spurious V - 1 or a masked one) or arg-max (lowest index on ties, clamped into range when the row is all
// NaN); position pos + 1 receives the token unless it belongs to the given prefix; tok[b] = the token the NEXT step embeds; *pos += 1.
// Grid: one block per row when the caller gives a zeroed `ticket` word (the last block to finish advances *pos and re-zeroes the ticket), else one block
// that walks the rows.  u[*pos * u_stride + b]: a table of uniforms drawn once per sample() call (u_stride = B) or one vector per step (u_stride = 0).
__device__ __forceinline__ unsigned f2ukey(float f) { const unsigned i = __float_as_uint(f); return (i & 0x80000000u) ? ~i : (i | 0x80000000u); }   // unsigned order = float order
__device__ __forceinline__ float ukey2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(1024) void sample_step_kernel(const float* __restrict__ logits, int B, int V, float inv_temp, const float* __restrict__ u, int u_stride,
                                                         int do_sample, int top_k, int64_t* __restrict__ seq, int total, int P, int* __restrict__ pos,
                                                         int* __restrict__ ticket, int64_t* __restrict__ tok) {
    __shared__ float sred[16];
    __shared__ int sidx[16];
    __shared__ float sscan[16];
    __shared__ int sfound, slast;
    __shared__ int shist[256];
    __shared__ int swtot[4];
    __shared__ unsigned sprefix;
    __shared__ int skrem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = *pos;
    const int ept = (V + 1023) / 1024;                 // elements per thread, contiguous: thread t owns [t * ept, (t + 1) * ept)
    const bool per_row = gridDim.x > 1;
    const float* const ur = u ? u + (int64_t)p * u_stride : nullptr;
    for (int b = per_row ? (int)blockIdx.x : 0; b < (per_row ? (int)blockIdx.x + 1 : B); ++b) {
        const float* lr = logits + (int64_t)b * V;
        // row maximum (and its lowest index)
        float mx = -INFINITY;
        int am = 0x7fffffff;
        for (int e = 0; e < ept; ++e) {
            const int v = tid * ept + e;
            if (v < V) {
                const float x = lr[v] * inv_temp;
                if (x > mx) { mx = x; am = v; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            const bool t0 = (om > mx) | ((om == mx) & (oa < am));
            mx = t0 ? om : mx;
            am = t0 ? oa : am;
        }
        if (lane == 0) { sred[wv] = mx; sidx[wv] = am; }
        __syncthreads();
        mx = sred[0];
        am = sidx[0];
        for (int w = 1; w < 16; ++w) {
            const bool t0 = (sred[w] > mx) | ((sred[w] == mx) & (sidx[w] < am));
            mx = t0 ? sred[w] : mx;
            am = t0 ? sidx[w] : am;
        }
        int ix = min(am, V - 1);                       // (all-NaN row: no element ever compared greater)
        if (do_sample) {
            // top-k cut: the k-th largest scaled logit by a four-pass radix select over order-preserving keys (256-bin histogram of the keys that share the
            // digits chosen so far; the digit whose suffix count reaches the remaining k is the next one)
            float kth = -INFINITY;
            if (top_k > 0 && top_k < V) {
                if (tid == 0) { sprefix = 0u; skrem = top_k; }
                for (int pass = 0; pass < 4; ++pass) {
                    const int shift = 24 - 8 * pass;
                    if (tid < 256) shist[tid] = 0;
                    __syncthreads();
                    const unsigned pref = sprefix, pmask = pass ? (0xffffffffu << (shift + 8)) : 0u;
                    const int krem = skrem;
                    for (int e = 0; e < ept; ++e) {
                        const int v = tid * ept + e;
                        if (v < V) {
                            const unsigned key = f2ukey(lr[v] * inv_temp);
                            if ((key & pmask) == pref) atomicAdd(&shist[(key >> shift) & 0xffu], 1);
                        }
                    }
                    __syncthreads();
                    int c = 0, inc = 0;
                    if (tid < 256) {
                        c = shist[tid];
                        inc = c;               // inclusive suffix count inside the wave: keys of this pass with digit >= tid
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const int t = __shfl_down(inc, o, 64);
                            if (lane + o < 64) inc += t;
                        }
                        if (lane == 0) swtot[wv] = inc;
                    }
                    __syncthreads();
                    if (tid < 256) {
                        for (int w = wv + 1; w < 4; ++w) inc += swtot[w];
                        if (inc >= krem && inc - c < krem) {        // exactly one digit: the suffix counts are monotone
                            sprefix = pref | ((unsigned)tid << shift);
                            skrem = krem - (inc - c);
                        }
                    }
                    __syncthreads();
                }
                kth = ukey2f(sprefix);
            }
            // prefix of exp(x - max) over the kept elements in index order: per-thread sums -> wave scan -> scan of the 16 wave totals
            float loc = 0.f;
            for (int e = 0; e < ept; ++e) {
                const int v = tid * ept + e;
                if (v < V) {
                    const float x = lr[v] * inv_temp;
                    loc += x < kth ? 0.f : __expf(x - mx);
                }
            }
            float inc = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            __syncthreads();               // (sred / sidx of the maximum have been read by everyone)
            if (lane == 63) sscan[wv] = inc;
            if (tid == 0) { sfound = 0x7fffffff; slast = -1; }
            __syncthreads();
            float wbase = 0.f, tot = 0.f;
            for (int w = 0; w < 16; ++w) {
                if (w < wv) wbase += sscan[w];
                tot += sscan[w];
            }
            const float target = ur[b] * tot;
            float run = wbase + inc - loc;  // cdf before this thread's first element
            // the first token of non-zero probability whose inclusive cdf reaches the target: every thread that owns probability mass walks its elements and
            // offers its first hit; the lowest index wins.  No "is the target inside my interval" test -- neighbouring threads' differently associated sums
            // can leave one-ulp gaps between intervals -- and a token the top-k cut removed (or with exp() = 0) can never be returned.
            int hit = 0x7fffffff, lastnz = -1;
            if (loc > 0.f) {
                for (int e = 0; e < ept; ++e) {
                    const int v = tid * ept + e;
                    if (v < V) {
                        const float x = lr[v] * inv_temp;
                        const float pe = x < kth ? 0.f : __expf(x - mx);
                        run += pe;
                        if (pe > 0.f) {
                            lastnz = v;
                            if (run >= target && hit == 0x7fffffff) hit = v;
                        }
                    }
                }
            }
            if (hit != 0x7fffffff) atomicMin(&sfound, hit);
            __syncthreads();
            if (sfound == 0x7fffffff) {     // rounding left every recomputed cdf value below the target (u close to 1): the last token that carries mass
                if (lastnz >= 0) atomicMax(&slast, lastnz);
                __syncthreads();
                ix = max(slast, 0);
            } else
                ix = sfound;
        }
        if (tid == 0) {
            const int np = p + 1;
            if (np < total) {
                if (np >= P) seq[(int64_t)b * total + np] = ix;
                tok[b] = seq[(int64_t)b * total + np];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (!per_row) *pos = p + 1;
        else {
            __threadfence();
            if (atomicAdd(ticket, 1) == B - 1) {      // every block has read *pos (at its start) and finished its row
                *ticket = 0;
                *pos = p + 1;
            }
        }
    }
}

// Phase A of a global-head step, one block per (batch, head): projections dd[m] = x . P[m] (P carries the data normaliser) of the new
// query and key into scratch, and the maximum of the key projections into the step's slot of the double-buffered atomic maximum
// (order-preserving integer encoding of the float).
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct FavorProjArgs {
    const float *q, *k, *proj;
    int q_stride, q_off, k_stride, k_off, G, dh, m, LDF;
    float* dd;      // [2][B*G][LDF]
    int* kmax;      // [2]
    const int* pos;
    int rows;       // B * G
};

__device__ __forceinline__ void favor_step_proj_body(const FavorProjArgs& a, const int bg) {
    __shared__ float sq[64], sk[64], red[4];
    const float *q = a.q, *k = a.k, *proj = a.proj;
    float* dd = a.dd;
    int* kmax = a.kmax;
    const int* pos = a.pos;
    const int q_stride = a.q_stride, q_off = a.q_off, k_stride = a.k_stride, k_off = a.k_off, G = a.G, dh = a.dh, m = a.m, LDF = a.LDF;
    const int b = bg / G, g = bg % G, tid = threadIdx.x;
    const int64_t rows = a.rows;
    if (tid < dh) {
        sq[tid] = q[(int64_t)b * q_stride + q_off + g * dh + tid];
        sk[tid] = k[(int64_t)b * k_stride + k_off + g * dh + tid];
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int e = tid; e < m; e += 256) {
        // the 16 pieces of the projection row are fetched BEFORE the dependent fma chain (dh = 64, checked by the launcher): one L2 round trip instead of 16 --
        // this loop was most of the launch's 11 us
        const float4* pr = (const float4*)(proj + (int64_t)e * 64);
        float4 pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = pr[i];
        float aq = 0.f, ak = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = 4 * i;
            aq = fmaf(sq[d], pv[i].x, fmaf(sq[d + 1], pv[i].y, fmaf(sq[d + 2], pv[i].z, fmaf(sq[d + 3], pv[i].w, aq))));
            ak = fmaf(sk[d], pv[i].x, fmaf(sk[d + 1], pv[i].y, fmaf(sk[d + 2], pv[i].z, fmaf(sk[d + 3], pv[i].w, ak))));
        }
        dd[(int64_t)bg * LDF + e] = aq;
        dd[(rows + bg) * LDF + e] = ak;
        mx = fmaxf(mx, ak);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) atomicMax(kmax + (*pos & 1), f2ord(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ __launch_bounds__(256) void favor_step_proj_kernel(const FavorProjArgs a) { favor_step_proj_body(a, (int)blockIdx.x); }

struct FavorStepArgs {
    const float *ddq, *ddk;            // [B*G, LDF] projections (data_normalizer folded into the projection matrix)
    const float *q, *k, *v;            // rows of the fused qkv output
    int q_stride, q_off, k_stride, k_off, v_stride, v_off;
    int G, dh, m, LDF;
    float* smax;                       // [2] running key maximum, slot (pos & 1) = before this step, slot ((pos + 1) & 1) = after
    int* kmax;                         // [2] atomic maximum of this step's key projections (ordered-int encoding), slot pos & 1
    float *E, *Ez, *V1;                // state [B*G, LDF, dh], [B*G, LDF], [B*G, dh]
    const int* pos;
    float* out;                        // attention output rows
    int out_stride, out_off;
    float eps_feat, eps_den;
};

__device__ __forceinline__ void favor_step_body(const FavorStepArgs& a, const int bg) {
    // 16 waves: wave w owns features w, w+16, ... (17 of the 266): their 64-float state rows are independent read-modify-writes, so all of
    // a wave's loads are in flight together -- with 4 waves and 67 dependent-looking iterations this kernel took 33 us of pure latency.
    __shared__ float sq[64], sk[64], sv[64], sqf[320], sek[320], red[32], snum[16][64];
    const int b = bg / a.G, g = bg % a.G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dh = a.dh, m = a.m;
    // everything the step reads from global memory that does not depend on another load is requested here (the 68 KB of state rows first), so that the
    // reductions below run under ONE round trip instead of five dependent ones
    float* const E = a.E + (int64_t)bg * a.LDF * dh;
    float ev[17];
#pragma unroll
    for (int it = 0; it < 17; ++it) {
        const int e = wv + 16 * it;
        ev[it] = e < m ? E[e * dh + lane] : 0.f;
    }
    const int p = *a.pos;
    const float ddq_t = tid < m ? a.ddq[(int64_t)bg * a.LDF + tid] : -INFINITY;
    const float ddk_t = tid < m ? a.ddk[(int64_t)bg * a.LDF + tid] : 0.f;
    const float ez_t = tid < m ? a.Ez[(int64_t)bg * a.LDF + tid] : 0.f;
    if (tid < dh) {
        sq[tid] = a.q[(int64_t)b * a.q_stride + a.q_off + g * dh + tid];
        sk[tid] = a.k[(int64_t)b * a.k_stride + a.k_off + g * dh + tid];
        sv[tid] = a.v[(int64_t)b * a.v_stride + a.v_off + g * dh + tid];
    }
    __syncthreads();
    const float ratio = rsqrtf((float)m), nrm = rsqrtf((float)dh) * 0.5f;   // diag = |x|^2 / 2 * dh^-1/2
    float dq = 0.f, dk = 0.f;
    for (int d = 0; d < dh; ++d) {
        dq = fmaf(sq[d], sq[d], dq);
        dk = fmaf(sk[d], sk[d], dk);
    }
    dq *= nrm;
    dk *= nrm;
    // query stabiliser: row maximum of its own projections
    float qm = ddq_t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qm = fmaxf(qm, __shfl_xor(qm, o, 64));
    if (lane == 0) red[wv] = qm;
    __syncthreads();
    qm = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) qm = fmaxf(qm, red[w]);
    const float s_old = a.smax[p & 1], s_new = fmaxf(s_old, ord2f(a.kmax[p & 1]));
    const float scale = s_old == -INFINITY ? 0.f : __expf(s_old - s_new);
    const int cnt = p + 1;   // keys seen including this one
    __syncthreads();         // (red is reused below)
    if (tid == 0) {          // every block writes the same values; the slots read above are not touched
        a.smax[(p + 1) & 1] = s_new;
        a.kmax[(p + 1) & 1] = f2ord(-INFINITY);   // ready for the next step's atomic maximum
    }
    // features: q' and the exp part of k'; denominator sum_m q'[m] * (ratio * (Ez[m] + eps * cnt) + eps_den)
    float den = 0.f, qsum = 0.f;
    if (tid < m) {
        const float qf = ratio * (__expf(ddq_t - dq - qm) + a.eps_feat);
        const float ek = __expf(ddk_t - dk - s_new);
        sqf[tid] = qf;
        sek[tid] = ek;
        const float z = fmaf(ez_t, scale, ek);
        a.Ez[(int64_t)bg * a.LDF + tid] = z;
        den = qf * (ratio * (z + a.eps_feat * (float)cnt) + a.eps_den);
        qsum = qf;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        den += __shfl_xor(den, o, 64);
        qsum += __shfl_xor(qsum, o, 64);
    }
    if (lane == 0) {
        red[wv] = den;
        red[16 + wv] = qsum;
    }
    __syncthreads();
    // state update + numerator: thread = (feature slice wv, value dim lane)
    float num = 0.f;
#pragma unroll
    for (int it = 0; it < 17; ++it) {
        const int e = wv + 16 * it;
        if (e < m) {
            const float x = fmaf(ev[it], scale, sek[e] * sv[lane]);
            E[e * dh + lane] = x;
            num = fmaf(sqf[e], x, num);
        }
    }
    snum[wv][lane] = num;
    __syncthreads();
    if (tid < dh) {
        den = 0.f;
        qsum = 0.f;
        num = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            den += red[w];
            qsum += red[16 + w];
            num += snum[w][tid];
        }
        const float v1 = a.V1[(int64_t)bg * dh + tid] + sv[tid];
        a.V1[(int64_t)bg * dh + tid] = v1;
        a.out[(int64_t)b * a.out_stride + a.out_off + g * dh + tid] = ratio * (num + a.eps_feat * qsum * v1) / den;
    }
}

__global__ __launch_bounds__(1024) void favor_step_kernel(const FavorStepArgs a) { favor_step_body(a, (int)blockIdx.x); }

// ---- small-batch dense layer of the decode step: y[b][o] = epi( sum_i x[b][i] W[o][i] + bias[o] ), B <= 32 rows.
// At B rows the layer is a stream over the weights (HBM bound).  Up to three weight tensors are concatenated along the output dimension
// (q | k | v in one launch).  round_in / round_w / round_out reproduce the bf16 operand and output rounding of the MFMA path when the network computes
// in bf16.  Epilogue: bias, GELU, then optionally y = res + gate * y (ReZero / plain residual with gate = 1).
struct GemvArgs {
    const float* x;
    int x_stride, in, B;
    const float* w[3];
    const float* bias[3];
    int seg[3];            // output columns of each weight tensor
    int nseg, O;
    float* y;
    int y_stride;
    int act;               // 0 none, 1 GELU
    const float* res;      // residual rows (stride res_stride) or NULL
    int res_stride;
    const float* gate;     // device scalar multiplying the layer output before the residual add, or NULL (1)
    int round_in, round_w, round_out;
    int opb;               // output columns per block: 16 (one MFMA tile), or 8 / 4 for narrow layers (more blocks streaming the weights; the other columns of the tile repeat them)
};

__device__ __forceinline__ float bf16_round(float f) { return __uint_as_float((uint32_t)f32_to_bf16(f) << 16); }

// One wave = 16 output columns x up to 16 batch rows on the MFMA: the weights are the A operand straight from global memory (lane =
// (column lane & 15, k-group lane >> 4) reads 8 consecutive k: 4 lanes cover a contiguous 128 bytes of a weight row), the input rows the B
// operand (L1 / L2 resident), and the reduction over k that a VALU dot product would finish with 6 cross-lane steps per value happens
// inside the instruction.  BF16 = operands rounded to bf16 as on the training path (mfma_f32_16x16x32_bf16); otherwise exact fp32
// (mfma_f32_16x16x4f32, K permuted so that a lane still reads 16 contiguous bytes).
// The K range is split over the block's 8 waves (a single wave streaming a whole weight row is pure HBM latency: 34 us for K = 2048);
// their partial 16 x 16 tiles are summed through LDS by wave 0, which runs the epilogue.
template <bool BF16, bool WB = false>   // WB: the weight tensors hold bf16 values (decode copies of the fp32 parameters: half the bytes, no conversion in the loop)
__global__ __launch_bounds__(512) void gemv_rows_kernel(const GemvArgs a, int b0) {
    __shared__ float part[8][64][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, kg = lane >> 4;
    const int o0 = blockIdx.x * a.opb;
    constexpr int KSTEP = BF16 ? 32 : 16;
    const int kspan = (((a.in + 7) >> 3) + KSTEP - 1) / KSTEP * KSTEP, kbeg = wv * kspan, kend = kbeg + kspan < a.in ? kbeg + kspan : a.in;
    // weight row of this lane's output column
    int o = o0 + (fr & (a.opb - 1)), sgi = 0;
    const bool ocol = o < a.O;
    if (!ocol) o = a.O - 1;
    int oo = o;
    while (sgi + 1 < a.nseg && oo >= a.seg[sgi]) oo -= a.seg[sgi++];
    const float* wr = a.w[sgi] + (int64_t)oo * a.in;
    const int b = b0 + fr;
    const bool brow = b < a.B;
    const float* xr = a.x + (int64_t)(brow ? b : 0) * a.x_stride;
    float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
    // epilogue operands of wave 0 (bias, gate, residual row): fetched NOW, so that their HBM round trip runs under the weight stream instead of after the
    // reduction (the two residual layers of a block were 8.4 us against 5.9-6.5 us for the others)
    float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_res[4] = {0.f, 0.f, 0.f, 0.f}, e_gate = 1.f;
    if (wv == 0 && brow) {
        if (a.gate) e_gate = *a.gate;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = o0 + 4 * kg + r;
            if (oc >= a.O || 4 * kg + r >= a.opb) break;
            int sg = 0, ol = oc;
            while (sg + 1 < a.nseg && ol >= a.seg[sg]) ol -= a.seg[sg++];
            if (a.bias[sg]) e_bias[r] = a.bias[sg][ol];
            if (a.res) e_res[r] = a.res[(int64_t)b * a.res_stride + oc];
        }
    }
    // The wave's K range in blocks of eight MFMA steps whose operand loads are ALL issued before the first product (steps beyond the range multiply a zero
    // input fragment): the compiler does not unroll the run-time loop ("loop not unrolled" with #pragma unroll 8), and one step per iteration was one dependent
    // memory round trip per step -- eight per wave for K = 2 048.
    if constexpr (BF16) {
        const bf16_t* wrb = (const bf16_t*)a.w[sgi] + (int64_t)oo * a.in;
        // NS steps from kb: loads first, then the products (steps beyond the range multiply a zero input fragment)
        auto block = [&](auto ns_tag, int kb) __attribute__((always_inline)) {
            constexpr int NS = decltype(ns_tag)::value;
            short8_t wa[NS];
            float4 x0[NS], x1[NS];
            float4 w0[WB ? 1 : NS], w1[WB ? 1 : NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int k0 = kb + 32 * i < kend ? kb + 32 * i : kbeg;      // (a valid address; its product is zeroed below)
                if constexpr (WB) {
                    wa[i] = *(const short8_t*)(wrb + k0 + kg * 8);
                } else {
                    w0[i] = *(const float4*)(wr + k0 + kg * 8);
                    w1[i] = *(const float4*)(wr + k0 + kg * 8 + 4);
                }
                x0[i] = *(const float4*)(xr + k0 + kg * 8);
                x1[i] = *(const float4*)(xr + k0 + kg * 8 + 4);
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if constexpr (!WB) {
                    wa[i][0] = (short)f32_to_bf16(w0[i].x); wa[i][1] = (short)f32_to_bf16(w0[i].y); wa[i][2] = (short)f32_to_bf16(w0[i].z); wa[i][3] = (short)f32_to_bf16(w0[i].w);
                    wa[i][4] = (short)f32_to_bf16(w1[i].x); wa[i][5] = (short)f32_to_bf16(w1[i].y); wa[i][6] = (short)f32_to_bf16(w1[i].z); wa[i][7] = (short)f32_to_bf16(w1[i].w);
                }
                short8_t xb;
                xb[0] = (short)f32_to_bf16(x0[i].x); xb[1] = (short)f32_to_bf16(x0[i].y); xb[2] = (short)f32_to_bf16(x0[i].z); xb[3] = (short)f32_to_bf16(x0[i].w);
                xb[4] = (short)f32_to_bf16(x1[i].x); xb[5] = (short)f32_to_bf16(x1[i].y); xb[6] = (short)f32_to_bf16(x1[i].z); xb[7] = (short)f32_to_bf16(x1[i].w);
                if (!brow || kb + 32 * i >= kend) xb = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[i], xb, acc, 0, 0, 0);
            }
        };
        const int nsteps = (kend - kbeg + 31) / 32;      // (block-uniform up to the last wave: waves take the branch of their own count)
        if (nsteps > 0) {
            if (nsteps <= 2) block(std::integral_constant<int, 2>{}, kbeg);
            else if (nsteps <= 4) block(std::integral_constant<int, 4>{}, kbeg);
            else
                for (int kb = kbeg; kb < kend; kb += 8 * 32) block(std::integral_constant<int, 8>{}, kb);
        }
    } else {
        for (int kb = kbeg; kb < kend; kb += 8 * 16) {   // MFMA e of a group uses k = k0 + kg*4 + e on both operands
            float4 w0[8], x0[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k0 = kb + 16 * i < kend ? kb + 16 * i : kbeg;
                w0[i] = *(const float4*)(wr + k0 + kg * 4);
                x0[i] = *(const float4*)(xr + k0 + kg * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 xv = x0[i];
                if (!brow || kb + 16 * i >= kend) xv = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i].x, xv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i].y, xv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i].z, xv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i].w, xv.w, acc, 0, 0, 0);
            }
        }
    }
    // D: lane holds output columns o0 + 4*kg + r (r = 0..3) of batch row b0 + fr
    *(float4_t*)part[wv][lane] = acc;
    __syncthreads();
    if (wv != 0 || !brow) return;
#pragma unroll
    for (int w = 1; w < 8; ++w) acc += *(const float4_t*)part[w][lane];
    const float gate = e_gate;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oc = o0 + 4 * kg + r;
        if (oc >= a.O || 4 * kg + r >= a.opb) break;
        float v = acc[r] + e_bias[r];
        if (a.round_out) v = bf16_round(v);
        if (a.act == 1) {
            v = gelu_f(v);
            if (a.round_out) v = bf16_round(v);
        }
        if (a.res) v = e_res[r] + gate * v;
        a.y[(int64_t)b * a.y_stride + oc] = v;
    }
}

struct LocalStepArgs {
    const float *q, *k, *v;
    int q_stride, q_off, k_stride, k_off, v_stride, v_off;
    const float *cosb, *sinb;          // rotary tables [N, dh]
    float *kc, *vc;                    // caches [B, L, N, dh] (rotated keys, values)
    const int* pos;
    int N, W, L, dh;
    float* out;
    int out_stride, out_off;
};

// one block per (batch, local head): rotate q_t / k_t, append (k_t, v_t) to the cache, softmax over the keys of the previous and the
// current window up to t (local_attention with look_backward = 1, causal), output row t
__global__ __launch_bounds__(1024) void local_attn_step_kernel(const LocalStepArgs a) {
    extern __shared__ float sc[];      // scores [2 W]
    __shared__ __attribute__((aligned(16))) float sq[64];
    __shared__ float red[16], sacc[16][64];
    const int bl = blockIdx.x, b = bl / a.L, l = bl % a.L, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dh = a.dh, half = dh / 2, t = *a.pos;
    float* kc = a.kc + ((int64_t)bl * a.N) * dh;
    float* vc = a.vc + ((int64_t)bl * a.N) * dh;
    if (tid < dh) {
        const float* qr = a.q + (int64_t)b * a.q_stride + a.q_off + l * dh;
        const float* kr = a.k + (int64_t)b * a.k_stride + a.k_off + l * dh;
        const float cs = a.cosb[t * dh + tid], sn = a.sinb[t * dh + tid];
        const float qrot = tid < half ? -qr[tid + half] : qr[tid - half];
        const float krot = tid < half ? -kr[tid + half] : kr[tid - half];
        sq[tid] = (qr[tid] * cs + qrot * sn) * rsqrtf((float)dh);
        kc[(int64_t)t * dh + tid] = kr[tid] * cs + krot * sn;
        vc[(int64_t)t * dh + tid] = a.v[(int64_t)b * a.v_stride + a.v_off + l * dh + tid];
    }
    __syncthreads();   // (the block re-reads its own global writes below: same block, after the barrier)
    __threadfence_block();
    const int w = t / a.W, lo = (w > 0 ? w - 1 : 0) * a.W, nk = t - lo + 1;
    // scores: one key per thread (at most 2 W <= 1024 keys), its 256-byte row as 16 independent 16-byte loads -- no cross-lane reduction
    for (int j = tid; j < nk; j += 1024) {
        const float4* kj = (const float4*)(kc + (int64_t)(lo + j) * dh);
        float4 kv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) kv[e] = kj[e];
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float4 qv = *(const float4*)(sq + 4 * e);
            d = fmaf(qv.x, kv[e].x, fmaf(qv.y, kv[e].y, fmaf(qv.z, kv[e].z, fmaf(qv.w, kv[e].w, d))));
        }
        sc[j] = d;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += 1024) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < nk; j += 1024) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += red[i];
    float acc = 0.f;   // thread = (key slice wv, value dim lane); 8 value rows in flight per wave
    int j = wv;
    for (; j + 112 < nk; j += 128) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = vc[(int64_t)(lo + j + 16 * u) * dh + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(sc[j + 16 * u], vv[u], acc);
    }
    for (; j < nk; j += 16) acc = fmaf(sc[j], vc[(int64_t)(lo + j) * dh + lane], acc);
    sacc[wv][lane] = acc;
    __syncthreads();
    if (tid < dh) {
        acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += sacc[i][tid];
        a.out[(int64_t)b * a.out_stride + a.out_off + l * dh + tid] = acc / sum;
    }
}


// ---- a layer's attention step in TWO launches (global + local heads present).  The three launches above leave most of the chip idle (48 + 48 + 48 blocks)
// and the local step walks up to 2 W keys in ONE block per (batch, head) (5 -> 29 us as the window fills).  Here:
//   launch A = [projections of the global heads (favor_step_proj_body) | local heads, keys split over LSPLIT blocks each: partial (max, sum, weighted values)]
//   launch B = [FAVOR+ state update + output (favor_step_body)          | combine of the local partials -> attention rows]
// The new key / value row is appended to the cache by split 0; every split keeps it in LDS as well, so no block reads another block's store.
constexpr int LSPLIT = 4;
constexpr int LPART = 66;      // per (batch, head, split): running maximum, sum of exp, 64 weighted value sums

__device__ __forceinline__ void local_step_partial_body(const LocalStepArgs& a, float* __restrict__ part, const int blk) {
    __shared__ __attribute__((aligned(16))) float sq[64], skt[64], svt[64];
    __shared__ float sc[256], red[4], sacc[4][64];
    const int bl = blk / LSPLIT, sp = blk % LSPLIT, b = bl / a.L, l = bl % a.L, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dh = a.dh, half = dh / 2, t = *a.pos;
    float* kc = a.kc + ((int64_t)bl * a.N) * dh;
    float* vc = a.vc + ((int64_t)bl * a.N) * dh;
    const int w = t / a.W, lo = (w > 0 ? w - 1 : 0) * a.W, nk = t - lo + 1;
    const int per = (nk + LSPLIT - 1) / LSPLIT, j0 = sp * per, j1 = min(nk, j0 + per);      // per <= 2 W / 4 <= 256 (checked by the launcher)
    const int cnt = max(0, j1 - j0);
    // Every cache row this block needs is requested NOW -- the key row of this thread (16 pieces) and the value elements of its (key slice wv, dim lane)
    // walk (up to 64) -- so that the whole step waits for ONE memory round trip beside the q / k / v rows; the value walk used to issue eight rows at a time
    // behind the softmax (seven dependent round trips at 210 keys).  Position t itself comes from LDS (this step's k / v rows).
    // (16 lanes per key row with a cross-lane reduction -- coalesced 256-byte requests -- measured SLOWER: 13.8 vs 10.5 us for the launch)
    float4 kv[16];
    if (tid < cnt) {
        const int j = lo + j0 + tid;
        const float4* kj = (const float4*)(kc + (int64_t)(j == t ? lo : j) * dh);     // (j == t: replaced below; any valid row)
#pragma unroll
        for (int e = 0; e < 16; ++e) kv[e] = kj[e];
    }
    // (unconditional loads from a clamped row + a bit mask: a select would put every load under its own branch -- 64 branches, ~1.5 us of issue)
    float vv[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int j = wv + 4 * u, jj = lo + j0 + j;
        const float x = vc[(int64_t)min(jj, a.N - 1) * dh + lane];
        vv[u] = __uint_as_float(__float_as_uint(x) & ((j < cnt && jj != t) ? 0xffffffffu : 0u));
    }
    if (tid < dh) {
        const float* qr = a.q + (int64_t)b * a.q_stride + a.q_off + l * dh;
        const float* kr = a.k + (int64_t)b * a.k_stride + a.k_off + l * dh;
        const float cs = a.cosb[t * dh + tid], sn = a.sinb[t * dh + tid];
        const float qrot = tid < half ? -qr[tid + half] : qr[tid - half];
        const float krot = tid < half ? -kr[tid + half] : kr[tid - half];
        const float kvv = kr[tid] * cs + krot * sn, vvv = a.v[(int64_t)b * a.v_stride + a.v_off + l * dh + tid];
        sq[tid] = (qr[tid] * cs + qrot * sn) * rsqrtf((float)dh);
        skt[tid] = kvv;
        svt[tid] = vvv;
        if (sp == 0) {
            kc[(int64_t)t * dh + tid] = kvv;
            vc[(int64_t)t * dh + tid] = vvv;
        }
    }
    __syncthreads();
    float d = -INFINITY;
    if (tid < cnt) {
        if (lo + j0 + tid == t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) kv[e] = *(const float4*)(skt + 4 * e);
        }
        d = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float4 qv = *(const float4*)(sq + 4 * e);
            d = fmaf(qv.x, kv[e].x, fmaf(qv.y, kv[e].y, fmaf(qv.z, kv[e].z, fmaf(qv.w, kv[e].w, d))));
        }
    }
    float mx = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float pj = tid < cnt ? __expf(d - mx) : 0.f;
    sc[tid] = pj;
    float sum = pj;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __syncthreads();          // red is reused; sc complete
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    float acc = 0.f;          // thread = (key slice wv, value dim lane); keys in ascending order, as before
    const float svl = svt[lane];
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int j = wv + 4 * u;      // (j >= cnt: sc[j] = 0 and vv[u] = 0)
        acc = fmaf(sc[j], lo + j0 + j == t ? svl : vv[u], acc);
    }
    sacc[wv][lane] = acc;
    __syncthreads();
    float* pr = part + (int64_t)blk * LPART;
    if (tid < dh) pr[2 + tid] = (sacc[0][tid] + sacc[1][tid]) + (sacc[2][tid] + sacc[3][tid]);
    if (tid == 0) {
        pr[0] = cnt > 0 ? mx : -INFINITY;
        pr[1] = cnt > 0 ? sum : 0.f;
    }
}

__device__ __forceinline__ void local_step_combine_body(const LocalStepArgs& a, const float* __restrict__ part, const int bl) {
    const int b = bl / a.L, l = bl % a.L, tid = threadIdx.x;
    if (tid >= a.dh) return;
    const float* pr = part + (int64_t)bl * LSPLIT * LPART;
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < LSPLIT; ++s) M = fmaxf(M, pr[s * LPART]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < LSPLIT; ++s) {
        const float ms = pr[s * LPART];
        const float w = ms == -INFINITY ? 0.f : __expf(ms - M);
        num = fmaf(w, pr[s * LPART + 2 + tid], num);
        den = fmaf(w, pr[s * LPART + 1], den);
    }
    a.out[(int64_t)b * a.out_stride + a.out_off + l * a.dh + tid] = num / den;
}

__global__ __launch_bounds__(256) void attn_step_a_kernel(const FavorProjArgs pa, const LocalStepArgs la, float* __restrict__ part, const int nproj) {
    if ((int)blockIdx.x < nproj) favor_step_proj_body(pa, (int)blockIdx.x);
    else local_step_partial_body(la, part, (int)blockIdx.x - nproj);
}
__global__ __launch_bounds__(1024) void attn_step_b_kernel(const FavorStepArgs fa, const LocalStepArgs la, const float* __restrict__ part, const int nfav) {
    if ((int)blockIdx.x < nfav) favor_step_body(fa, (int)blockIdx.x);
    else local_step_combine_body(la, part, (int)blockIdx.x - nfav);
}


// ------------------------------------------------------------------------------------------------ cross entropy (one wave per row)
// The loss terms of a block's rows are summed in LDS and added with ONE atomic per block: one atomic per row (8 400 on one address) serialised in L2 and made
// this 17-MB kernel take 158 us (round 5).
#define CE_ROWS 8
__global__ __launch_bounds__(64 * CE_ROWS) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t R, int V,
                                                         float* __restrict__ loss_sum, void* dlogits, int d_dtype, float gscale) {
    __shared__ float sterm[CE_ROWS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * CE_ROWS + w;
    float term = 0.f;
    if (r < R) {
        const float* lr = logits + r * V;
        float mx = -INFINITY;
        for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += expf(lr[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        const int64_t tg = target[r];
        // a class id outside [0, V) never reads out of bounds: torch's ignore_index (-100) contributes nothing, anything else poisons the loss with a
        // NaN (torch raises a device assert there; a silent wrong loss would be worse than a loud one)
        const bool valid = tg >= 0 && tg < (int64_t)V, ignored = tg == -100;
        term = valid ? lse - lr[tg] : (ignored ? 0.f : __uint_as_float(0x7fc00000u));
        if (dlogits) {
            for (int c = lane; c < V; c += 64) {
                const float p = expf(lr[c] - lse);
                store_from_f32(dlogits, d_dtype, r * V + c, valid ? (p - (c == tg ? 1.f : 0.f)) * gscale : 0.f);
            }
        }
    }
    if (lane == 0) sterm[w] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < CE_ROWS; ++i) t += sterm[i];
        unsafeAtomicAdd(loss_sum, t);
    }
}

}  // namespace sa

using namespace sa;
#define ST(s) ((hipStream_t)(s))

extern "C" int sa_embed_sum(int ntab, const float* const* tables, const int64_t* const* idx, const int32_t* per_position, int dim, int N, int64_t R,
                            float* out, void* stream) {
    if (ntab < 1 || ntab > 6 || !tables || !idx || !per_position || !out || dim <= 0 || R <= 0) return SA_EINVAL;
    EmbedArgs a;
    for (int t = 0; t < ntab; ++t) {
        a.table[t] = tables[t];
        a.idx[t] = idx[t];
        a.per_position[t] = per_position[t];
    }
    a.ntab = ntab;
    a.dim = dim;
    a.N = N;
    a.R = R;
    SA_LAUNCH(embed_sum_kernel, dim3(grid1d(R * dim)), dim3(256), 0, ST(stream), a, out);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_embed_scatter(const float* dy, float* dtable, const int64_t* idx, int per_position, int dim, int N, int64_t R, void* stream) {
    if (!dy || !dtable || !idx || dim <= 0 || R <= 0) return SA_EINVAL;
    SA_LAUNCH(embed_scatter_kernel, dim3(grid1d(R * dim)), dim3(256), 0, ST(stream), dy, dtable, idx, per_position, dim, N, R);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_layernorm_fwd(const float* x, const float* w, const float* b, float* y, void* y_lp, int lp_dtype, float* stats, int64_t R, int C,
                                float eps, void* stream) {
    if (!x || !w || !b || !y || !stats || R <= 0 || C <= 0) return SA_EINVAL;
    SA_LAUNCH(layernorm_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, ST(stream), x, w, b, y, y_lp, lp_dtype, stats, R, C, eps);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_layernorm_bwd(const float* dy, const float* x, const float* w, const float* stats, float* dx, float* dw, float* db, int64_t R, int C,
                                void* stream) {
    if (!dy || !x || !w || !stats || !dx || !dw || !db || R <= 0 || C <= 0) return SA_EINVAL;
    if (C <= 512) SA_LAUNCH(layernorm_bwd_rows_kernel, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, ST(stream), dy, x, w, stats, dx, dw, db, R, C);
    else SA_LAUNCH(layernorm_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, ST(stream), dy, x, w, stats, dx, dw, db, R, C);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_gelu(const void* u, int u_dtype, void* h, int h_dtype, int64_t n, void* stream) {
    if (!u || !h || n <= 0) return SA_EINVAL;
    if (u_dtype == SA_BF16 && h_dtype == SA_BF16 && (n & 7) == 0 && (((uintptr_t)u | (uintptr_t)h) & 15) == 0) {
        SA_LAUNCH(gelu_bf16x8_kernel, dim3(grid1d(n / 8)), dim3(256), 0, ST(stream), (const uint4*)u, (uint4*)h, n / 8);
        SA_CHECK_LAUNCH();
        return 0;
    }
    SA_LAUNCH(gelu_kernel, dim3(grid1d(n)), dim3(256), 0, ST(stream), u, u_dtype, h, h_dtype, n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_rezero_fwd(const float* x, const void* F, int f_dtype, const float* g, float* y, void* y_lp, int lp_dtype, int64_t n, void* stream) {
    if (!x || !F || !g || !y || n <= 0) return SA_EINVAL;
    if (f_dtype == SA_BF16 && (!y_lp || lp_dtype == SA_BF16) && (n & 3) == 0 && (((uintptr_t)x | (uintptr_t)F | (uintptr_t)y | (uintptr_t)y_lp) & 15) == 0) {
        SA_LAUNCH(rezero_fwd_bf16x4_kernel, dim3(grid1d(n / 4)), dim3(256), 0, ST(stream), (const float4*)x, (const uint2*)F, g, (float4*)y, (uint2*)y_lp, n / 4);
        SA_CHECK_LAUNCH();
        return 0;
    }
    SA_LAUNCH(rezero_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, ST(stream), x, F, f_dtype, g, y, y_lp, lp_dtype, n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_rezero_bwd(const float* dy, const void* F, int f_dtype, const float* g, void* dF, int df_dtype, float* dg, int64_t n, void* stream) {
    if (!dy || !F || !g || !dF || !dg || n <= 0) return SA_EINVAL;
    if (f_dtype == SA_BF16 && df_dtype == SA_BF16 && (n & 3) == 0 && (((uintptr_t)dy | (uintptr_t)F | (uintptr_t)dF) & 15) == 0) {
        SA_LAUNCH(rezero_bwd_bf16x4_kernel, dim3(grid1d(n / 4, 1024, 256)), dim3(1024), 0, ST(stream), (const float4*)dy, (const uint2*)F, g, (uint2*)dF, dg,
                           n / 4);
        SA_CHECK_LAUNCH();
        return 0;
    }
    SA_LAUNCH(rezero_bwd_kernel, dim3(grid1d(n, 256, 1024)), dim3(256), 0, ST(stream), dy, F, f_dtype, g, dF, df_dtype, dg, n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_axpy(float* y, const float* x, float alpha, int64_t n, void* stream) {
    if (!y || !x || n <= 0) return SA_EINVAL;
    SA_LAUNCH(axpy_kernel, dim3(grid1d(n)), dim3(256), 0, ST(stream), y, x, alpha, n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_features_fwd(const float* dd, const float* src, int src_stride, int h0, int G, int dh, int is_query, float* feat, void* gmax_ws,
                                     int64_t rows, int m, int LDF, void* stream) {
    if (!dd || !src || !feat || rows <= 0 || m <= 0 || LDF < m || (is_query != 1 && !gmax_ws)) return SA_EINVAL;
    if ((LDF & 3) || LDF > 512) return SA_EUNSUPPORTED;   // rows are read as 16-byte pieces, two per lane
    const float c = powf((float)dh, -0.25f), ratio = 1.f / sqrtf((float)m);
    unsigned long long* gm = nullptr;
    if (is_query == 2) {          // keys, global maximum already in gmax_ws (sa_favor_project computed it from its accumulators)
        gm = (unsigned long long*)gmax_ws;
    } else if (!is_query) {
        gm = (unsigned long long*)gmax_ws;
        hipMemsetAsync(gm, 0, 8, ST(stream));
        SA_LAUNCH(favor_global_max_kernel, dim3(grid1d(rows * 16, 256, 2048)), dim3(256), 0, ST(stream), dd, rows, m, LDF, gm);
        SA_CHECK_LAUNCH();
    }
    SA_LAUNCH(favor_feat_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), dd, src, src_stride, h0, G, dh, gm, feat, rows, m,
                       LDF, 0.5f * c * c, ratio, 1e-4f);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_features_bwd(const float* dfeat, const float* feat, const float* dd, const float* src, int src_stride, int h0, int G, int dh,
                                     int is_query, float* ddd, float* dsrc, const void* gmax_ws, float* tsum_ws, int64_t rows, int m, int LDF,
                                     void* stream) {
    if (!dfeat || !feat || !dd || !src || !ddd || !dsrc || rows <= 0 || (!is_query && (!gmax_ws || !tsum_ws))) return SA_EINVAL;
    const float c = powf((float)dh, -0.25f), ratio = 1.f / sqrtf((float)m);
    SA_LAUNCH(favor_feat_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), dfeat, feat, dd, src, src_stride, h0, G, dh,
                       is_query, ddd, dsrc, tsum_ws, rows, m, LDF, c * c, ratio * 1e-4f);
    SA_CHECK_LAUNCH();
    if (!is_query) {
        SA_LAUNCH(favor_key_stab_kernel, dim3(1), dim3(1024), 0, ST(stream), ddd, (const unsigned long long*)gmax_ws, tsum_ws, rows);
        SA_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sa_favor_projection(const float* blocks, const float* rows, float* out, int nmat, int nblk, int m, int d, void* stream) {
    if (!blocks || !rows || !out || nmat <= 0 || nblk <= 0 || m <= 0 || d <= 0 || d > 64 || nblk * d < m) return SA_EINVAL;
    SA_LAUNCH(favor_projection_kernel, dim3(nmat * nblk), dim3(64), 0, ST(stream), blocks, rows, out, m, d, nblk);
    SA_CHECK_LAUNCH();
    return 0;
}

static int check_scan(int B, int N, int G, int LDF, int dv) { return (B > 0 && N > 0 && G > 0 && LDF > 0 && LDF <= 272 && dv == 64) ? 0 : SA_EUNSUPPORTED; }

static void scan_segments(int N, int& S, int& seg_len, const void* ws) {
    S = ws ? (N + 127) / 128 : 1;      // VALU path: ~128 positions per block
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    seg_len = (N + S - 1) / S;
    S = (N + seg_len - 1) / seg_len;
}

extern "C" int64_t sa_favor_scan_workspace_bytes(int B, int N, int G, int LDF, int dv) {
    return (int64_t)B * G * ((N + 63) / 64) * LDF * (dv + 1) * 4;  // one state (+ its running column sums) per 64-position chunk
}

// which == 0: scan A, 1: scan B.  With a workspace: chunked MFMA path (3 launches); without: one VALU block per (b, g).
template <typename K>
static int run_scan(K valu_kernel, int which, ScanArgs& s, unsigned base_blocks, float* ws, hipStream_t st) {
    const bool no_mfma = dbg(SA_DBG_SCAN_VALU);
    if (!ws) {
        s.S = 1; s.seg_len = s.N; s.pass = 0; s.state = nullptr;
        SA_LAUNCH(valu_kernel, dim3(base_blocks), dim3(256), 0, st, s);
        SA_CHECK_LAUNCH();
        return 0;
    }
    s.state = ws;
    const int64_t bg = (int64_t)s.B * s.G;
    int64_t elems = (int64_t)s.LDF * s.dv;
    const bool mfma_ok = !no_mfma && (s.LDF & 15) == 0 && s.LDF <= 272 && s.dv == 64;
    if (s.zmode) s.zcol = 1;
    if ((s.zcol || s.state_ready) && !mfma_ok) return SA_EUNSUPPORTED;   // the fused running sums / shared states exist on the chunked MFMA path only
    if (mfma_ok) {
        elems += s.zcol ? s.LDF : 0;
        s.S = (s.N + 63) / 64;
        s.seg_len = 64;
        const unsigned nblk = (unsigned)(bg * s.S);
        const int exact = (int)((g_debug_flags.load(std::memory_order_relaxed) >> SA_DBG_SCAN_EXACT_SHIFT) & 7u) | (s.exact ? 7 : 0);   // bit 0: state sums, bit 1: scan A outputs, bit 2: scan B outputs on the exact-fp32 MFMA kernels
        const bool fits32 = (int64_t)s.N * s.G * s.LDF * 4 < ((int64_t)1 << 31) && (int64_t)s.N * std::max(s.b_stride, std::max(s.c_stride, s.y_stride)) * 4 < ((int64_t)1 << 31);
        if (!s.state_ready) {
            if (!(exact & 1) && fits32) SA_LAUNCH(favor_chunk_state_split_kernel, dim3(nblk), dim3(256), 0, st, s);
            else SA_LAUNCH(favor_chunk_state_kernel, dim3(nblk), dim3(256), 0, st, s);
            SA_CHECK_LAUNCH();
            SA_LAUNCH(scan_state_prefix_kernel, dim3((unsigned)((bg * elems + 255) / 256)), dim3(256), 0, st, ws, bg, s.S, elems);
            SA_CHECK_LAUNCH();
        }
        if (which == 0) {
            if (!(exact & 2) && fits32) SA_LAUNCH(favor_chunk_out_a_split_kernel, dim3(nblk), dim3(256), 0, st, s);
            else SA_LAUNCH(favor_chunk_out_a_kernel, dim3(nblk), dim3(256), 0, st, s);
        } else {
            if (!(exact & 4) && fits32) SA_LAUNCH(favor_chunk_out_b_split_kernel, dim3(nblk), dim3(256), 0, st, s);
            else SA_LAUNCH(favor_chunk_out_b_kernel, dim3(nblk), dim3(256), 0, st, s);
        }
        SA_CHECK_LAUNCH();
        return 0;
    }
    scan_segments(s.N, s.S, s.seg_len, ws);
    s.pass = 1;
    SA_LAUNCH(valu_kernel, dim3(base_blocks * s.S), dim3(256), 0, st, s);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(scan_state_prefix_kernel, dim3((unsigned)((bg * elems + 255) / 256)), dim3(256), 0, st, ws, bg, s.S, elems);
    SA_CHECK_LAUNCH();
    s.pass = 2;
    SA_LAUNCH(valu_kernel, dim3(base_blocks * s.S), dim3(256), 0, st, s);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_scan_a(const float* a, const float* c, const float* b, int b_stride, int b_off, const float* b_scale, float* y, int y_stride,
                               int y_off, const float* y_scale, int B, int N, int G, int LDF, int dv, int reverse, int accumulate, float* state_ws,
                               void* stream) {
    if (!a || !c || !b || !y) return SA_EINVAL;
    if (check_scan(B, N, G, LDF, dv)) return SA_EUNSUPPORTED;
    ScanArgs s = {};
    s.a = a; s.c_feat = c; s.b = b; s.b_scale = b_scale; s.y = y; s.y_scale = y_scale;
    s.B = B; s.N = N; s.G = G; s.LDF = LDF; s.dv = dv; s.b_stride = b_stride; s.b_off = b_off; s.y_stride = y_stride; s.y_off = y_off;
    s.reverse = reverse; s.accumulate = accumulate;
    return run_scan(favor_scan_a_kernel, 0, s, (unsigned)(B * G * (dv / 16)), state_ws, ST(stream));
}

extern "C" int sa_favor_scan_b(const float* a, const float* b, int b_stride, int b_off, const float* b_scale, const float* c, int c_stride, int c_off,
                               const float* c_scale, float* y, const float* ex_scale, const float* ex_vec, float ex_const, int B, int N, int G,
                               int LDF, int dv, int reverse, float* state_ws, void* stream) {
    if (!a || !c || !b || !y) return SA_EINVAL;
    if (check_scan(B, N, G, LDF, dv)) return SA_EUNSUPPORTED;
    ScanArgs s = {};
    s.a = a; s.b = b; s.c_col = c; s.b_scale = b_scale; s.c_scale = c_scale; s.y = y; s.ex_scale = ex_scale; s.ex_vec = ex_vec; s.ex_const = ex_const;
    s.B = B; s.N = N; s.G = G; s.LDF = LDF; s.dv = dv; s.b_stride = b_stride; s.b_off = b_off; s.c_stride = c_stride; s.c_off = c_off; s.reverse = reverse;
    return run_scan(favor_scan_b_kernel, 1, s, (unsigned)(B * G * ((LDF + 63) / 64)), state_ws, ST(stream));
}

extern "C" int sa_cumsum_rows(const float* x, const float* scale, float* out, int B, int N, int G, int LDF, int reverse, float* seg_ws, void* stream) {
    if (!x || !out || B <= 0 || N <= 0 || G <= 0 || LDF <= 0) return SA_EINVAL;
    int S, seg_len;
    scan_segments(N, S, seg_len, seg_ws);   // seg_ws: B*G*S*LDF floats (<= 1/dv of the scan workspace)
    const int64_t threads = (int64_t)B * G * LDF * S;
    const unsigned nblk = (unsigned)((threads + 255) / 256);
    if (S > 1) {
        SA_LAUNCH(cumsum_rows_kernel, dim3(nblk), dim3(256), 0, ST(stream), x, scale, (float*)nullptr, seg_ws, B, N, G, LDF, reverse, S, seg_len);
        SA_CHECK_LAUNCH();
    }
    SA_LAUNCH(cumsum_rows_kernel, dim3(nblk), dim3(256), 0, ST(stream), x, scale, out, S > 1 ? seg_ws : (float*)nullptr, B, N, G, LDF, reverse, S, seg_len);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_den(const float* q, const float* z, float eps, float* inv, int64_t rows, int m, int LDF, void* stream) {
    if (!q || !z || !inv || rows <= 0) return SA_EINVAL;
    SA_LAUNCH(favor_den_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), q, z, eps, inv, rows, m, LDF);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_dden(const float* dout, const float* out, int stride, int off, int G, int dv, const float* inv, float* dden, int64_t rows,
                             void* stream) {
    if (!dout || !out || !inv || !dden || rows <= 0) return SA_EINVAL;
    SA_LAUNCH(favor_dden_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), dout, out, stride, off, G, dv, inv, dden, rows);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_rotary(const float* x, int stride, int off, int L, int dh, const float* cosb, const float* sinb, float* y, int y_stride, int y_off,
                         int N, int64_t R, int transpose, int accumulate, void* stream) {
    if (!x || !cosb || !sinb || !y || L <= 0 || dh <= 0 || (dh & 1) || R <= 0) return SA_EINVAL;
    if ((dh & 7) || ((stride | off | y_stride | y_off) & 3)) return SA_EUNSUPPORTED;   // 16-byte accesses on both halves of a head row
    SA_LAUNCH(rotary_kernel, dim3(grid1d(R * L * dh / 8)), dim3(256), 0, ST(stream), x, stride, off, L, dh, cosb, sinb, y, y_stride, y_off, N, R,
                       transpose, accumulate, 1, (int64_t)0, (int64_t)0, (unsigned short*)nullptr);
    SA_CHECK_LAUNCH();
    return 0;
}

// the same for `ngroups` operands in one launch (q and k of a layer): operand gi lives x_goff / y_goff ELEMENTS behind operand 0
extern "C" int sa_rotary_groups(const float* x, int stride, int off, int L, int dh, const float* cosb, const float* sinb, float* y, int y_stride, int y_off,
                                int N, int64_t R, int transpose, int accumulate, int ngroups, int64_t x_goff, int64_t y_goff, void* y_lp, void* stream) {
    if (!x || !cosb || !sinb || !y || L <= 0 || dh <= 0 || (dh & 1) || R <= 0 || ngroups < 1) return SA_EINVAL;
    if ((dh & 7) || ((stride | off | y_stride | y_off) & 3) || ((x_goff | y_goff) & 3)) return SA_EUNSUPPORTED;
    SA_LAUNCH(rotary_kernel, dim3(grid1d(R * L * dh / 8 * ngroups)), dim3(256), 0, ST(stream), x, stride, off, L, dh, cosb, sinb, y, y_stride, y_off, N, R,
                       transpose, accumulate, ngroups, x_goff, y_goff, (unsigned short*)y_lp);
    SA_CHECK_LAUNCH();
    return 0;
}

// rotary embedding of the GLOBAL heads (the wrapper's rotary_position_emb=True; performer_pytorch 1.0.11 apply_rotary_pos_emb): the pair of consecutive
// dimensions (2i, 2i + 1) of every head row is rotated by the angle whose sine / cosine sit in columns i / dh/2 + i of row n of the table [N, dh] (sin | cos
// halves of FixedPositionalEmbedding(dim_head)).  transpose = 1: the adjoint (the rotation by the opposite angle).  Thread = two pairs (one 16-byte access);
// in place (y == x) is fine: a thread reads its four values before it writes them.
__global__ void rotary_pairs_kernel(const float* __restrict__ x, int stride, int off, int L, int dh, const float* __restrict__ sincos, float* __restrict__ y,
                                    int y_stride, int y_off, int N, int64_t R, int transpose, int ngroups, int64_t x_goff, int64_t y_goff) {
    const int q4 = dh / 4, half = dh / 2;
    const int64_t per = R * L * q4, total = per * ngroups;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += (int64_t)gridDim.x * blockDim.x) {
        const int gi = (int)(e0 / per);
        const int64_t e = e0 - gi * per;
        const int j = (int)(e % q4);
        const int64_t rh = e / q4;
        const int h = (int)(rh % L);
        const int64_t r = rh / L;
        const int n = (int)(r % N);
        const float4 v = *(const float4*)(x + gi * x_goff + r * stride + off + h * dh + 4 * j);
        const float2 sn = *(const float2*)(sincos + (int64_t)n * dh + 2 * j), cs = *(const float2*)(sincos + (int64_t)n * dh + half + 2 * j);
        const float s0 = transpose ? -sn.x : sn.x, s1 = transpose ? -sn.y : sn.y;
        float4 o;
        o.x = v.x * cs.x - v.y * s0;      // x cos + rotate_every_two(x) sin: rotate_every_two(x)_{2i} = -x_{2i+1}, _{2i+1} = x_{2i}
        o.y = v.y * cs.x + v.x * s0;
        o.z = v.z * cs.y - v.w * s1;
        o.w = v.w * cs.y + v.z * s1;
        *(float4*)(y + gi * y_goff + r * y_stride + y_off + h * dh + 4 * j) = o;
    }
}

extern "C" int sa_rotary_pairs(const float* x, int stride, int off, int L, int dh, const float* sincos, float* y, int y_stride, int y_off, int N, int64_t R,
                               int transpose, int ngroups, int64_t x_goff, int64_t y_goff, void* stream) {
    if (!x || !sincos || !y || L <= 0 || dh <= 0 || R <= 0 || N <= 0 || ngroups < 1) return SA_EINVAL;
    if ((dh & 3) || ((stride | off | y_stride | y_off) & 3) || ((x_goff | y_goff) & 3)) return SA_EUNSUPPORTED;   // 16-byte accesses
    SA_LAUNCH(rotary_pairs_kernel, dim3(grid1d(R * L * dh / 4 * ngroups)), dim3(256), 0, ST(stream), x, stride, off, L, dh, sincos, y, y_stride, y_off, N, R, transpose,
              ngroups, x_goff, y_goff);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_cross_entropy(const float* logits, const int64_t* target, int64_t R, int V, float* loss_sum, void* dlogits, int d_dtype, float gscale,
                                void* stream) {
    if (!logits || !target || !loss_sum || R <= 0 || V <= 0) return SA_EINVAL;
    SA_LAUNCH(ce_kernel, dim3((unsigned)((R + CE_ROWS - 1) / CE_ROWS)), dim3(64 * CE_ROWS), 0, ST(stream), logits, target, R, V, loss_sum, dlogits, d_dtype, gscale);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_embed_step(int ntab, const float* const* tables, const int64_t* const* idx, const int32_t* per_position, int dim, const int* pos, int B,
                             float* out, void* stream) {
    if (ntab < 1 || ntab > 6 || !tables || !idx || !per_position || !pos || !out || B <= 0) return SA_EINVAL;
    EmbedArgs a;
    a.ntab = ntab;
    a.dim = dim;
    a.N = 1;
    a.R = B;
    for (int t = 0; t < ntab; ++t) {
        a.table[t] = tables[t];
        a.idx[t] = idx[t];
        a.per_position[t] = per_position[t];
    }
    SA_LAUNCH(embed_step_kernel, dim3(grid1d((int64_t)B * dim)), dim3(256), 0, ST(stream), a, pos, B, out);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_sample_step(const float* logits, int B, int V, float temperature, const float* u, int u_stride, int do_sample, int top_k, int64_t* seq, int total,
                              int P, int* pos, int* ticket, int64_t* tok, void* stream) {
    if (!logits || !seq || !pos || !tok || (do_sample && !u) || u_stride < 0 || B <= 0 || V <= 0 || total <= 0 || !(temperature > 0.f)) return SA_EINVAL;
    SA_LAUNCH(sample_step_kernel, dim3(ticket ? B : 1), dim3(1024), 0, ST(stream), logits, B, V, 1.f / temperature, u, u_stride, do_sample, top_k, seq, total, P, pos,
              ticket, tok);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_step(const float* q, int q_stride, int q_off, const float* k, int k_stride, int k_off, const float* v, int v_stride, int v_off,
                             const float* proj, int B, int G, int dh, int m, int LDF, float* smax, int* kmax, float* dd, float* E, float* Ez, float* V1,
                             const int* pos, float* out, int out_stride, int out_off, void* stream) {
    if (!q || !k || !v || !proj || !smax || !kmax || !dd || !E || !Ez || !V1 || !pos || !out) return SA_EINVAL;
    if (dh != 64 || m > 272 || B <= 0 || G <= 0) return SA_EUNSUPPORTED;   // 16 waves x 17 features
    FavorProjArgs pa;
    pa.q = q; pa.k = k; pa.proj = proj; pa.q_stride = q_stride; pa.q_off = q_off; pa.k_stride = k_stride; pa.k_off = k_off; pa.G = G; pa.dh = dh; pa.m = m; pa.LDF = LDF;
    pa.dd = dd; pa.kmax = kmax; pa.pos = pos; pa.rows = B * G;
    SA_LAUNCH(favor_step_proj_kernel, dim3(B * G), dim3(256), 0, ST(stream), pa);
    SA_CHECK_LAUNCH();
    FavorStepArgs a;
    a.ddq = dd; a.ddk = dd + (int64_t)B * G * LDF; a.q = q; a.k = k; a.v = v;
    a.q_stride = q_stride; a.q_off = q_off; a.k_stride = k_stride; a.k_off = k_off; a.v_stride = v_stride; a.v_off = v_off;
    a.G = G; a.dh = dh; a.m = m; a.LDF = LDF; a.smax = smax; a.kmax = kmax; a.E = E; a.Ez = Ez; a.V1 = V1; a.pos = pos;
    a.out = out; a.out_stride = out_stride; a.out_off = out_off;
    a.eps_feat = 1e-4f;
    a.eps_den = 1e-6f;
    SA_LAUNCH(favor_step_kernel, dim3(B * G), dim3(1024), 0, ST(stream), a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_local_attn_step(const float* q, int q_stride, int q_off, const float* k, int k_stride, int k_off, const float* v, int v_stride, int v_off,
                                  const float* cosb, const float* sinb, float* kcache, float* vcache, const int* pos, int B, int N, int L, int W, int dh,
                                  float* out, int out_stride, int out_off, void* stream) {
    if (!q || !k || !v || !cosb || !sinb || !kcache || !vcache || !pos || !out) return SA_EINVAL;
    if (dh != 64 || W <= 0 || 2 * (size_t)W * 4 > 60 * 1024 || B <= 0 || L <= 0) return SA_EUNSUPPORTED;
    LocalStepArgs a;
    a.q = q; a.k = k; a.v = v;
    a.q_stride = q_stride; a.q_off = q_off; a.k_stride = k_stride; a.k_off = k_off; a.v_stride = v_stride; a.v_off = v_off;
    a.cosb = cosb; a.sinb = sinb; a.kc = kcache; a.vc = vcache; a.pos = pos; a.N = N; a.W = W; a.L = L; a.dh = dh;
    a.out = out; a.out_stride = out_stride; a.out_off = out_off;
    SA_LAUNCH(local_attn_step_kernel, dim3(B * L), dim3(1024), 2 * (size_t)W * sizeof(float), ST(stream), a);
    SA_CHECK_LAUNCH();
    return 0;
}

// global + local heads of one layer's decode step in two launches (attn_step_a_kernel / attn_step_b_kernel): arguments of sa_favor_step and sa_local_attn_step,
// plus `part`: B * L * 4 * 66 floats of scratch.  Same results as the two calls up to the summation order of the local heads' softmax (keys in four segments).
extern "C" int sa_attn_step(const float* qkv, int stride, int inner, const float* proj, int B, int G, int L, int dh, int m, int LDF, float* smax, int* kmax, float* dd,
                            float* E, float* Ez, float* V1, const float* cosb, const float* sinb, float* kcache, float* vcache, int N, int W, float* part,
                            const int* pos, float* out, int out_stride, void* stream) {
    if (!qkv || !proj || !smax || !kmax || !dd || !E || !Ez || !V1 || !cosb || !sinb || !kcache || !vcache || !part || !pos || !out) return SA_EINVAL;
    if (dh != 64 || m > 272 || B <= 0 || G <= 0 || L <= 0 || W <= 0 || (2 * W + LSPLIT - 1) / LSPLIT > 256 || inner != (G + L) * dh) return SA_EUNSUPPORTED;
    FavorProjArgs pa;
    pa.q = qkv; pa.k = qkv; pa.proj = proj; pa.q_stride = stride; pa.q_off = 0; pa.k_stride = stride; pa.k_off = inner; pa.G = G; pa.dh = dh; pa.m = m; pa.LDF = LDF;
    pa.dd = dd; pa.kmax = kmax; pa.pos = pos; pa.rows = B * G;
    LocalStepArgs la;
    la.q = qkv; la.k = qkv; la.v = qkv; la.q_stride = stride; la.q_off = G * dh; la.k_stride = stride; la.k_off = inner + G * dh; la.v_stride = stride;
    la.v_off = 2 * inner + G * dh; la.cosb = cosb; la.sinb = sinb; la.kc = kcache; la.vc = vcache; la.pos = pos; la.N = N; la.W = W; la.L = L; la.dh = dh;
    la.out = out; la.out_stride = out_stride; la.out_off = G * dh;
    SA_LAUNCH(attn_step_a_kernel, dim3(B * G + B * L * LSPLIT), dim3(256), 0, ST(stream), pa, la, part, B * G);
    SA_CHECK_LAUNCH();
    FavorStepArgs a;
    a.ddq = dd; a.ddk = dd + (int64_t)B * G * LDF; a.q = qkv; a.k = qkv; a.v = qkv;
    a.q_stride = stride; a.q_off = 0; a.k_stride = stride; a.k_off = inner; a.v_stride = stride; a.v_off = 2 * inner;
    a.G = G; a.dh = dh; a.m = m; a.LDF = LDF; a.smax = smax; a.kmax = kmax; a.E = E; a.Ez = Ez; a.V1 = V1; a.pos = pos;
    a.out = out; a.out_stride = out_stride; a.out_off = 0;
    a.eps_feat = 1e-4f;
    a.eps_den = 1e-6f;
    SA_LAUNCH(attn_step_b_kernel, dim3(B * G + B * L), dim3(1024), 0, ST(stream), a, la, (const float*)part, B * G);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_gemv_rows(const float* x, int x_stride, int in, int B, int nseg, const float* const* w, const float* const* bias, const int32_t* seg_out,
                            float* y, int y_stride, int act, const float* res, int res_stride, const float* gate, int round_in, int round_w, int round_out,
                            void* stream) {
    if (!x || !w || !seg_out || !y || nseg < 1 || nseg > 3 || B <= 0 || in <= 0 || (in & (round_in ? 31 : 15)) || (x_stride & 3)) return SA_EINVAL;   // K steps of 32 (bf16) / 16 (fp32)
    if ((round_in != 0) != (round_w != 0) || round_w > 2) return SA_EUNSUPPORTED;   // both operands in bf16 (MFMA bf16) or both exact; round_w = 2: w holds bf16 tensors
    GemvArgs a;
    a.x = x; a.x_stride = x_stride; a.in = in; a.B = B; a.nseg = nseg; a.O = 0;
    for (int s = 0; s < 3; ++s) {
        a.w[s] = s < nseg ? w[s] : nullptr;
        a.bias[s] = (s < nseg && bias) ? bias[s] : nullptr;
        a.seg[s] = s < nseg ? seg_out[s] : 0;
        a.O += a.seg[s];
    }
    a.y = y; a.y_stride = y_stride; a.act = act; a.res = res; a.res_stride = res_stride; a.gate = gate;
    a.round_in = round_in; a.round_w = round_w; a.round_out = round_out;
    // narrow layers (to_out, w2: 512 outputs = 32 MFMA tiles): 4 (or 8) columns per block -> 128 (64) blocks pull the weight stream instead of 32
    a.opb = (a.O <= 512 && (a.O & 3) == 0) ? 4 : (a.O <= 1024 && (a.O & 7) == 0) ? 8 : 16;
    const unsigned blocks = (unsigned)((a.O + a.opb - 1) / a.opb);
    for (int b0 = 0; b0 < B; b0 += 16) {
        if (round_w == 2) SA_LAUNCH((gemv_rows_kernel<true, true>), dim3(blocks), dim3(512), 0, ST(stream), a, b0);
        else if (round_in) SA_LAUNCH(gemv_rows_kernel<true>, dim3(blocks), dim3(512), 0, ST(stream), a, b0);
        else SA_LAUNCH(gemv_rows_kernel<false>, dim3(blocks), dim3(512), 0, ST(stream), a, b0);
        SA_CHECK_LAUNCH();
    }
    return 0;
}

// scan A with the FAVOR+ normaliser fused: y_i = (sum_{j<=i} (c_i . a_j) b_j) / (c_i . (sum_{j<=i} a_j + den_eps)); inv_out[i] = 1 / that
// denominator (kept for the backward pass).  Replaces sa_cumsum_rows + sa_favor_den + sa_favor_scan_a(y_scale = inv).  Chunked MFMA
// path only (LDF % 16 == 0, LDF <= 272, dv == 64, workspace given): SA_EUNSUPPORTED otherwise.
extern "C" int sa_favor_scan_a_norm(const float* a, const float* c, const float* b, int b_stride, int b_off, float* y, int y_stride, int y_off, float* inv_out,
                                    float den_eps, int B, int N, int G, int LDF, int dv, float* state_ws, int state_flags, void* stream) {
    if (!a || !c || !b || !y || !inv_out || !state_ws) return SA_EINVAL;
    if (check_scan(B, N, G, LDF, dv)) return SA_EUNSUPPORTED;
    ScanArgs s = {};
    s.a = a; s.c_feat = c; s.b = b; s.y = y;
    s.B = B; s.N = N; s.G = G; s.LDF = LDF; s.dv = dv; s.b_stride = b_stride; s.b_off = b_off; s.y_stride = y_stride; s.y_off = y_off;
    s.zmode = 1; s.den_eps = den_eps; s.inv_out = inv_out; s.state_ready = state_flags & 1; s.exact = (state_flags >> 2) & 1;
    return run_scan(favor_scan_a_kernel, 0, s, (unsigned)(B * G * (dv / 16)), state_ws, ST(stream));
}

// scan B with a cumulative extra term computed on the fly (no cumsum pass, no [B,N,G,LDF] operand):
//   ex_mode 1: y_i[m] += ex_scale_i * (sum_{j<=i} a_j[m] + ex_const)      (gradient wrt the query features: ex_scale = d den)
//   ex_mode 2: y_i[m] += sum_{j<=i} a_j[m] ex_scale_j                     (gradient wrt the key features, reversed scan)
// (j <= i in scan order).  Chunked MFMA path only.
extern "C" int sa_favor_scan_b_cum(const float* a, const float* b, int b_stride, int b_off, const float* b_scale, const float* c, int c_stride, int c_off,
                                   const float* c_scale, float* y, const float* ex_scale, int ex_mode, float ex_const, int B, int N, int G, int LDF, int dv,
                                   int reverse, float* state_ws, int state_flags, void* stream) {
    if (!a || !b || !c || !y || !ex_scale || !state_ws || (ex_mode != 1 && ex_mode != 2)) return SA_EINVAL;
    if (check_scan(B, N, G, LDF, dv)) return SA_EUNSUPPORTED;
    ScanArgs s = {};
    s.a = a; s.b = b; s.c_col = c; s.b_scale = b_scale; s.c_scale = c_scale; s.y = y; s.ex_scale = ex_scale; s.ex_vec = nullptr; s.ex_const = ex_const;
    s.B = B; s.N = N; s.G = G; s.LDF = LDF; s.dv = dv; s.b_stride = b_stride; s.b_off = b_off; s.c_stride = c_stride; s.c_off = c_off; s.reverse = reverse;
    s.zmode = ex_mode; s.state_ready = state_flags & 1; s.exact = (state_flags >> 2) & 1;
    return run_scan(favor_scan_b_kernel, 1, s, (unsigned)(B * G * ((LDF + 63) / 64)), state_ws, ST(stream));
}

// sa_favor_scan_a on a state buffer some other scan of the SAME (a, b, b_scale, reverse) already filled: state_flags bit 0 = the exclusive
// chunk prefixes are in state_ws (skip the state and prefix passes), bit 1 = the buffer has the extra running-sum column (it was written
// by sa_favor_scan_a_norm / sa_favor_scan_b_cum).  The three backward scans of a FAVOR+ head need two distinct state sets, not four.
extern "C" int sa_favor_scan_a_state(const float* a, const float* c, const float* b, int b_stride, int b_off, const float* b_scale, float* y, int y_stride,
                                     int y_off, const float* y_scale, int B, int N, int G, int LDF, int dv, int reverse, int accumulate, float* state_ws,
                                     int state_flags, void* stream) {
    if (!a || !c || !b || !y || !state_ws) return SA_EINVAL;
    if (check_scan(B, N, G, LDF, dv)) return SA_EUNSUPPORTED;
    ScanArgs s = {};
    s.a = a; s.c_feat = c; s.b = b; s.b_scale = b_scale; s.y = y; s.y_scale = y_scale;
    s.B = B; s.N = N; s.G = G; s.LDF = LDF; s.dv = dv; s.b_stride = b_stride; s.b_off = b_off; s.y_stride = y_stride; s.y_off = y_off;
    s.reverse = reverse; s.accumulate = accumulate;
    s.state_ready = state_flags & 1;
    s.zcol = (state_flags >> 1) & 1;
    s.exact = (state_flags >> 2) & 1;
    return run_scan(favor_scan_a_kernel, 0, s, (unsigned)(B * G * (dv / 16)), state_ws, ST(stream));
}
