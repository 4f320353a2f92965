// Order-independent-by-construction forms of the reductions that the throughput path accumulates with fp32 atomics (the reference's
// --deterministic flag, run_vqvae.py / src/utils/general.py:336-338: torch.backends.cudnn.deterministic).  Every sum below is taken in a FIXED
// order -- per-block partials in a caller-supplied workspace, then one pass over the partials -- so two runs of the same step give bit-identical
// gradients, codebook statistics and parameters.  Selected by the host when `--deterministic` / SA_DETERMINISTIC is set; slower than the
// atomic forms (extra passes), never used by the benchmarked configuration.
#include "sa_common.h"

namespace sa {

constexpr int CD_BLOCKS = 256;   // row blocks of the two-stage column sums

// stage 1: partial[bx][c] = sum over the block's rows (4 row lanes, combined in a fixed order) of g[r][c]
__global__ __launch_bounds__(256) void colsum_det_stage1_kernel(const void* __restrict__ gp, int dtype, int64_t M, int C, int cstride, float* __restrict__ partial,
                                                                int64_t rows_per_block) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += load_as_f32(gp, dtype, r * cstride + c);
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) partial[(int64_t)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// stage 2: db[c] += sum over the blocks, in block order
__global__ void colsum_det_stage2_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * C + c];
    db[c] += s;
}

// codebook statistics of the EMA quantizer (reference baseline.py:66-69: encodings.sum(0), encodings^T @ flat_inputs) and the commitment error:
// block k walks ALL rows in index order and sums those assigned to code k; thread j owns dimension j.  err[k] = sum over its rows of |w_k - x|^2.
__global__ __launch_bounds__(256) void vq_stats_det_kernel(const float* __restrict__ rows, const float* __restrict__ cb, const int64_t* __restrict__ idx, int64_t M,
                                                           int D, float* __restrict__ counts, float* __restrict__ dw, float* __restrict__ err) {
    __shared__ float red[256];
    const int k = blockIdx.x, t = threadIdx.x;
    float cnt = 0.f, e = 0.f;
    for (int j0 = 0; j0 < D; j0 += 256) {   // (D <= 256 in every configuration: one pass)
        const int j = j0 + t;
        const float w = j < D ? cb[(int64_t)k * D + j] : 0.f;
        float acc = 0.f;
        for (int64_t i = 0; i < M; ++i) {
            if (idx[i] == k) {
                if (j0 == 0 && t == 0) cnt += 1.f;
                if (j < D) {
                    const float x = rows[i * D + j];
                    acc += x;
                    e += (w - x) * (w - x);
                }
            }
        }
        if (j < D) dw[(int64_t)k * D + j] = acc;
    }
    red[t] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {   // fixed tree
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) {
        counts[k] = cnt;
        err[k] = red[0];
    }
}
__global__ __launch_bounds__(1024) void vq_err_det_kernel(const float* __restrict__ err, int K, float* __restrict__ sqerr) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) s += err[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sqerr[0] = red[0];
}

// embedding gradient: block = one table row, thread = dimension; positions are visited in order
__global__ __launch_bounds__(256) void embed_scatter_det_kernel(const float* __restrict__ dy, float* __restrict__ dtable, const int64_t* __restrict__ idx, int per_position,
                                                                int dim, int N, int64_t R) {
    const int64_t row = blockIdx.x;
    const int64_t n_idx = per_position ? N : R;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float acc = 0.f;
        if (per_position) {
            for (int64_t p = 0; p < n_idx; ++p)
                if (idx[p] == row)
                    for (int64_t r = p; r < R; r += N) acc += dy[r * dim + c];   // the same position of every sequence, in batch order
        } else {
            for (int64_t r = 0; r < R; ++r)
                if (idx[r] == row) acc += dy[r * dim + c];
        }
        dtable[row * dim + c] += acc;
    }
}

// ---- Performer: the sums its backward pass accumulates with fp32 atomics (ReZero gate gradient, CE loss, LayerNorm weight gradient), in a fixed order
// x[0 .. n) summed by ONE block: thread t adds elements t, t + 1024, ... in order, then a fixed tree over the 1 024 partials
__global__ __launch_bounds__(1024) void sum_det_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out, int accumulate) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += 1024) s += x[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0];
}

// squared-error partials: block b owns the contiguous range [b * per, (b + 1) * per) and leaves ONE partial; gradient written as mse_kernel does
__global__ __launch_bounds__(256) void mse_det_stage1_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int64_t per,
                                                             float* __restrict__ partial, float* __restrict__ grad, float gcoef) {
    __shared__ float red[256];
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float s = 0.f;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
        const float d = a[e] - b[e];
        s += d * d;
        if (grad) grad[e] = d * gcoef;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// stage 1 of a dot product: block b owns the contiguous range [b * per, (b + 1) * per) and leaves ONE partial (same thread order and tree as above)
__global__ __launch_bounds__(256) void dot_det_stage1_kernel(const float* __restrict__ a, const void* __restrict__ b, int b_dtype, int64_t n, int64_t per,
                                                             float* __restrict__ partial) {
    __shared__ float red[256];
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float s = 0.f;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 256) s += a[e] * load_as_f32(b, b_dtype, e);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// per-row cross entropy (the row losses are summed afterwards by sum_det_kernel); gradient as ce_kernel
__global__ void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t R, int V, float* __restrict__ row_loss, void* dlogits,
                               int d_dtype, float gscale) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
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
    const bool valid = tg >= 0 && tg < (int64_t)V, ignored = tg == -100;   // as ce_kernel: ignore_index contributes nothing, any other bad id poisons the loss
    if (lane == 0) row_loss[r] = valid ? lse - lr[tg] : (ignored ? 0.f : __uint_as_float(0x7fc00000u));
    if (dlogits) {
        for (int c = lane; c < V; c += 64) {
            const float p = expf(lr[c] - lse);
            store_from_f32(dlogits, d_dtype, r * V + c, valid ? (p - (c == tg ? 1.f : 0.f)) * gscale : 0.f);
        }
    }
}

// prod[r][c] = dy[r][c] * (x[r][c] - mean_r) * rstd_r: the summand of the LayerNorm weight gradient (its column sum goes through sa_colsum_det)
__global__ void layernorm_dwprod_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ prod, int64_t R,
                                        int C) {
    const int64_t total = R * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / C;
        prod[e] = dy[e] * (x[e] - stats[2 * r]) * stats[2 * r + 1];
    }
}

}  // namespace sa

using namespace sa;

// out[0] (+)= sum of x[0 .. n) in a fixed order (one block)
extern "C" int sa_sum_det(const float* x, int64_t n, float* out, int accumulate, void* stream) {
    if (!x || !out || n <= 0) return SA_EINVAL;
    SA_LAUNCH(sum_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out, accumulate);
    SA_CHECK_LAUNCH();
    return 0;
}

// out[0] (+)= sum_e a[e] * b[e] in a fixed order (b fp32 or bf16): 1 024 contiguous ranges, then one block over the partials; ws: 1 024 floats
extern "C" int sa_dot_det(const float* a, const void* b, int b_dtype, int64_t n, float* out, int accumulate, float* ws, void* stream) {
    if (!a || !b || !out || !ws || n <= 0) return SA_EINVAL;
    if (b_dtype != SA_F32 && b_dtype != SA_BF16) return SA_EUNSUPPORTED;
    const int64_t per = (n + 1023) / 1024;
    const int nblk = (int)((n + per - 1) / per);
    SA_LAUNCH(dot_det_stage1_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, b_dtype, n, per, ws);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(sum_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)ws, (int64_t)nblk, out, accumulate);
    SA_CHECK_LAUNCH();
    return 0;
}

// sa_mse in a fixed order (the logged loss / validation MSE decide the key-metric checkpoint): 2 048 contiguous ranges, then one block; ws: 2 048 floats
extern "C" int sa_mse_det(const float* a, const float* b, int64_t n, float* loss_sum, float* grad, float gscale, float* ws, void* stream) {
    if (!a || !b || !loss_sum || !ws || n <= 0) return SA_EINVAL;
    const int64_t per = (n + 2047) / 2048;
    const int nblk = (int)((n + per - 1) / per);
    SA_LAUNCH(mse_det_stage1_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, n, per, ws, grad, 2.f * gscale / (float)n);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(sum_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)ws, (int64_t)nblk, loss_sum, 1);
    SA_CHECK_LAUNCH();
    return 0;
}

// sa_cross_entropy with the row losses written to row_loss[R] instead of accumulated atomically (sum them with sa_sum_det)
extern "C" int sa_cross_entropy_rows(const float* logits, const int64_t* target, int64_t R, int V, float* row_loss, void* dlogits, int d_dtype, float gscale,
                                     void* stream) {
    if (!logits || !target || !row_loss || R <= 0 || V <= 0) return SA_EINVAL;
    SA_LAUNCH(ce_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, target, R, V, row_loss, dlogits, d_dtype, gscale);
    SA_CHECK_LAUNCH();
    return 0;
}

// summand matrix of the LayerNorm weight gradient (stats = mean, rstd pairs of sa_layernorm_fwd); its column sums = d weight
extern "C" int sa_layernorm_dwprod(const float* dy, const float* x, const float* stats, float* prod, int64_t R, int C, void* stream) {
    if (!dy || !x || !stats || !prod || R <= 0 || C <= 0) return SA_EINVAL;
    const int64_t total = R * C;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    SA_LAUNCH(layernorm_dwprod_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, prod, R, C);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sa_colsum_det_workspace_bytes(int C) { return (int64_t)CD_BLOCKS * (C > 0 ? C : 1) * 4; }

// db[c] += sum_m g[m][c] in a fixed order (bias gradients in deterministic mode); ws >= sa_colsum_det_workspace_bytes(C)
extern "C" int sa_colsum_det(const void* gp, int dtype, int64_t M, int C, int cstride, float* db, void* ws, int64_t ws_bytes, void* stream) {
    if (!gp || !db || !ws || M <= 0 || C <= 0 || cstride < C) return SA_EINVAL;
    if (dtype != SA_F32 && dtype != SA_BF16) return SA_EUNSUPPORTED;
    if (ws_bytes < sa_colsum_det_workspace_bytes(C)) return SA_EINVAL;
    const int64_t rpb = (M + CD_BLOCKS - 1) / CD_BLOCKS;
    const int nblk = (int)((M + rpb - 1) / rpb);
    SA_LAUNCH(colsum_det_stage1_kernel, dim3(nblk, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, gp, dtype, M, C, cstride, (float*)ws, rpb);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(colsum_det_stage2_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, C, db);
    SA_CHECK_LAUNCH();
    return 0;
}

// counts[K], dw[K][D], sqerr[1] of sa_vq_assign recomputed in a fixed order from its idx output (overwrites them); err_ws: K floats
extern "C" int sa_vq_stats_det(const float* rows, const float* codebook, const int64_t* idx, int64_t M, int K, int D, float* counts, float* dw, float* sqerr,
                               float* err_ws, void* stream) {
    if (!rows || !codebook || !idx || !counts || !dw || !sqerr || !err_ws || M <= 0 || K <= 0 || D <= 0) return SA_EINVAL;
    SA_LAUNCH(vq_stats_det_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, rows, codebook, idx, M, D, counts, dw, err_ws);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(vq_err_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)err_ws, K, sqerr);
    SA_CHECK_LAUNCH();
    return 0;
}

// sa_embed_scatter in a fixed order: dtable [nrows][dim] += gathered rows of dy
extern "C" int sa_embed_scatter_det(const float* dy, float* dtable, const int64_t* idx, int per_position, int dim, int N, int64_t R, int nrows, void* stream) {
    if (!dy || !dtable || !idx || dim <= 0 || R <= 0 || nrows <= 0 || (per_position && N <= 0)) return SA_EINVAL;
    SA_LAUNCH(embed_scatter_det_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, dy, dtable, idx, per_position, dim, N, R);
    SA_CHECK_LAUNCH();
    return 0;
}
