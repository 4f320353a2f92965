// BatchNorm3d (+ fused LeakyReLU) over channels-last [M, C] activations -- the normalisation of the PatchGAN-3D
// discriminator, reference src/networks/discriminator/baseline.py:52-79 (nn.BatchNorm3d + nn.LeakyReLU(0.2)).
// HBM-bound column reductions + elementwise passes; statistics and gradients accumulate in fp32.
#include "sa_common.h"

namespace sa {

// sums[c] += sum_m f(x[m][c]); sums[C + c] += sum_m f2(...)   mode 0: (x, x^2)   mode 1: (g, g * xhat) with xhat from (x, mean, rstd)
__global__ void bn_colstats_kernel(const void* x, const void* g, int dtype, const float* __restrict__ mean, const float* __restrict__ rstd, int64_t M,
                                   int C, float* __restrict__ sums, int64_t rows_per_block, int mode, int partial) {
    const int c = blockIdx.y * 32 + (threadIdx.x & 31);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float mu = mode ? mean[c] : 0.f, rs = mode ? rstd[c] : 0.f;
        for (int64_t r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
            const float xv = load_as_f32(x, dtype, r * C + c);
            if (mode == 0) {
                s1 += xv;
                s2 += xv * xv;
            } else {
                const float gv = load_as_f32(g, dtype, r * C + c);
                s1 += gv;
                s2 += gv * (xv - mu) * rs;
            }
        }
    }
    __shared__ float red[2][256];
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 32 && c < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t1 += red[0][threadIdx.x + 32 * k];
            t2 += red[1][threadIdx.x + 32 * k];
        }
        if (partial) {   // deterministic mode: one slot per row block behind the totals, summed in block order by bn_sum_partials_kernel
            sums[(int64_t)(1 + blockIdx.x) * 2 * C + c] = t1;
            sums[(int64_t)(1 + blockIdx.x) * 2 * C + C + c] = t2;
        } else {
            unsafeAtomicAdd(sums + c, t1);
            unsafeAtomicAdd(sums + C + c, t2);
        }
    }
}

__global__ void bn_sum_partials_kernel(float* __restrict__ sums, int nblk, int C) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * C) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += sums[(int64_t)(1 + b) * 2 * C + e];
    sums[e] = s;
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, int64_t M, int C, float eps, float momentum, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mu = sums[c] / (float)M;
    float var = sums[C + c] / (float)M - mu * mu;
    var = var < 0.f ? 0.f : var;
    mean[c] = mu;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean) {
        const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = rsqrtf(running_var[c] + eps);
}

// y = lrelu((x - mean) rstd w + b)
__global__ void bn_apply_kernel(const void* x, int dtype, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ w,
                                const float* __restrict__ b, void* y, int64_t n, int C, float slope) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float v = (load_as_f32(x, dtype, e) - mean[c]) * rstd[c] * w[c] + b[c];
        v = v > 0.f ? v : v * slope;
        store_from_f32(y, dtype, e, v);
    }
}

// dx = w rstd (g - mean(g) - xhat mean(g xhat)),  sums = [sum g | sum g xhat]  (training) ; eval: dx = w rstd g
__global__ void bn_bwd_apply_kernel(const void* x, const void* g, int dtype, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ w, const float* __restrict__ sums, int64_t M, void* dx, int64_t n, int C, int training) {
    const float invM = 1.f / (float)M;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float gv = load_as_f32(g, dtype, e);
        float v = gv;
        if (training) {
            const float xh = (load_as_f32(x, dtype, e) - mean[c]) * rstd[c];
            v = gv - sums[c] * invM - xh * sums[C + c] * invM;
        }
        store_from_f32(dx, dtype, e, v * w[c] * rstd[c]);
    }
}

__global__ void bn_param_grads_kernel(float* __restrict__ dw, float* __restrict__ db, const float* __restrict__ sums, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        db[c] += sums[c];
        dw[c] += sums[C + c];
    }
}

// g = dy * lrelu'(y)   (y > 0 ? 1 : slope)
__global__ void lrelu_mask_kernel(const void* dy, const void* y, int dtype, void* g, int64_t n, float slope) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float d = load_as_f32(dy, dtype, e);
        store_from_f32(g, dtype, e, load_as_f32(y, dtype, e) > 0.f ? d : d * slope);
    }
}

static inline unsigned grid_e(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace sa

using namespace sa;

// floats of scratch sa_bn_forward / sa_bn_backward need in sums_ws under the CURRENT debug flags (deterministic mode keeps 64 per-block partials per statistic)
extern "C" int64_t sa_bn_sums_ws_floats(int C) { return (int64_t)(dbg(SA_DBG_DETERMINISTIC) ? 65 : 1) * 2 * (C > 0 ? C : 1); }

extern "C" int sa_bn_forward(const void* x, int dtype, int64_t M, int C, const float* w, const float* b, float* running_mean, float* running_var,
                             float momentum, float eps, int training, float slope, void* y, float* mean, float* rstd, float* sums_ws, void* stream) {
    if (!x || !w || !b || !y || !mean || !rstd || !sums_ws || M <= 0 || C <= 0) return SA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (training) {
        hipMemsetAsync(sums_ws, 0, sizeof(float) * 2 * C, st);
        const bool det = dbg(SA_DBG_DETERMINISTIC);   // sums_ws then holds (1 + 64) * 2 * C floats: totals + per-block partials
        int64_t rpb = det ? (M + 63) / 64 : (M + 1023) / 1024;
        if (rpb < 64) rpb = 64;
        dim3 grid((unsigned)((M + rpb - 1) / rpb), (C + 31) / 32);
        SA_LAUNCH(bn_colstats_kernel, grid, dim3(256), 0, st, x, nullptr, dtype, nullptr, nullptr, M, C, sums_ws, rpb, 0, det ? 1 : 0);
        SA_CHECK_LAUNCH();
        if (det) {
            SA_LAUNCH(bn_sum_partials_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, sums_ws, (int)grid.x, C);
            SA_CHECK_LAUNCH();
        }
        SA_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums_ws, M, C, eps, momentum, mean, rstd, running_mean, running_var);
    } else {
        if (!running_mean || !running_var) return SA_EINVAL;
        SA_LAUNCH(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, running_mean, running_var, C, eps, mean, rstd);
    }
    SA_CHECK_LAUNCH();
    SA_LAUNCH(bn_apply_kernel, dim3(grid_e(M * C)), dim3(256), 0, st, x, dtype, mean, rstd, w, b, y, M * C, C, slope);
    SA_CHECK_LAUNCH();
    return 0;
}

// g = gradient wrt the BatchNorm output (activation mask already applied); dw += sum g xhat, db += sum g, dx as above
extern "C" int sa_bn_backward(const void* x, const void* g, int dtype, int64_t M, int C, const float* w, const float* mean, const float* rstd,
                              int training, void* dx, float* dw, float* db, float* sums_ws, void* stream) {
    if (!x || !g || !w || !mean || !rstd || !dx || !dw || !db || !sums_ws || M <= 0 || C <= 0) return SA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(sums_ws, 0, sizeof(float) * 2 * C, st);
    const bool det = dbg(SA_DBG_DETERMINISTIC);
    int64_t rpb = det ? (M + 63) / 64 : (M + 1023) / 1024;
    if (rpb < 64) rpb = 64;
    dim3 grid((unsigned)((M + rpb - 1) / rpb), (C + 31) / 32);
    SA_LAUNCH(bn_colstats_kernel, grid, dim3(256), 0, st, x, g, dtype, mean, rstd, M, C, sums_ws, rpb, 1, det ? 1 : 0);
    SA_CHECK_LAUNCH();
    if (det) {
        SA_LAUNCH(bn_sum_partials_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, sums_ws, (int)grid.x, C);
        SA_CHECK_LAUNCH();
    }
    SA_LAUNCH(bn_bwd_apply_kernel, dim3(grid_e(M * C)), dim3(256), 0, st, x, g, dtype, mean, rstd, w, sums_ws, M, dx, M * C, C, training);
    SA_CHECK_LAUNCH();
    SA_LAUNCH(bn_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dw, db, sums_ws, C);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_lrelu_mask(const void* dy, const void* y, int dtype, void* g, int64_t n, float slope, void* stream) {
    if (!dy || !y || !g || n <= 0) return SA_EINVAL;
    SA_LAUNCH(lrelu_mask_kernel, dim3(grid_e(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dtype, g, n, slope);
    SA_CHECK_LAUNCH();
    return 0;
}
