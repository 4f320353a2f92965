/* ORACLE (test infrastructure only) -- plain-C restatement of the EMA vector quantizer.
 *
 * Follows /root/reference/src/networks/vqvae/baseline.py:38-87 (Quantizer_impl.forward):
 *   :46     flat = x.permute(0,2,3,4,1).view(-1, D)            (caller passes flat rows)
 *   :49-53  d = sum(flat^2) - 2 flat W^T + sum(W^2)^T          (expanded form, fp32)
 *   :56     idx = argmax(-d)  (first maximum wins)
 *   :68-69  counts = onehot.sum(0);  dw = onehot^T flat
 *   :75-80  N <- g N + (1-g) counts; embed_avg <- g embed_avg + (1-g) dw;
 *           n = sum N; W <- embed_avg / ((N+eps)/(n+K eps) n)
 *   :82     loss = beta * mean((W[idx]-x)^2)
 * Parity: PINNED by tests/golden/quantizer.npz (reference outputs).  Built by oracle/Makefile into
 * oracle/libsa_oracle.so; only tests / smoke / bench's cpu_baseline leg may load it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* rows [M,D], codebook [K,D] -> idx[M], counts[K], dw[K,D], *sqerr = sum (W[idx]-x)^2.  gap[M] (optional) receives the
 * margin between best and second-best -d. */
void sa_oracle_vq_assign(const float *rows, const float *codebook, int64_t M, int K, int D, int64_t *idx, float *counts,
                         float *dw, double *sqerr, float *gap) {
    float *wn = (float *)malloc(sizeof(float) * (size_t)K);
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int j = 0; j < D; ++j) s += codebook[(size_t)k * D + j] * codebook[(size_t)k * D + j];
        wn[k] = s;
    }
    memset(counts, 0, sizeof(float) * (size_t)K);
    memset(dw, 0, sizeof(float) * (size_t)K * D);
    double se = 0.0;
    for (int64_t m = 0; m < M; ++m) {
        const float *x = rows + (size_t)m * D;
        float xn = 0.f;
        for (int j = 0; j < D; ++j) xn += x[j] * x[j];
        float best = -INFINITY, second = -INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float *w = codebook + (size_t)k * D;
            float dot = 0.f;
            for (int j = 0; j < D; ++j) dot = fmaf(x[j], w[j], dot);
            float negd = -((xn - 2.f * dot) + wn[k]);
            if (negd > best) { second = best; best = negd; bi = k; }
            else if (negd > second) second = negd;
        }
        idx[m] = bi;
        if (gap) gap[m] = best - second;
        counts[bi] += 1.f;
        const float *w = codebook + (size_t)bi * D;
        for (int j = 0; j < D; ++j) {
            dw[(size_t)bi * D + j] += x[j];
            double e = (double)w[j] - (double)x[j];
            se += e * e;
        }
    }
    *sqerr = se;
    free(wn);
}

void sa_oracle_vq_ema_update(float *N, float *embed_avg, float *codebook, const float *counts, const float *dw, int K, int D,
                             float decay, float eps) {
    float n = 0.f;
    for (int k = 0; k < K; ++k) {
        N[k] = N[k] * decay + counts[k] * (1.f - decay);
        n += N[k];
    }
    for (int k = 0; k < K; ++k) {
        float wn = (N[k] + eps) / (n + (float)K * eps) * n;
        for (int j = 0; j < D; ++j) {
            size_t o = (size_t)k * D + j;
            embed_avg[o] = embed_avg[o] * decay + dw[o] * (1.f - decay);
            codebook[o] = embed_avg[o] / wn;
        }
    }
}
