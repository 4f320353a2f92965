// Random-feature projection of FAVOR+ (performer_pytorch.softmax_kernel: data_dash = data_normalizer * data @ projection^T, and its adjoint)
// as HBM-bound kernels: dd[r][f] = sum_d x[r][d] P[f][d] with d = 64, f < m <= 272.  The generic implicit-GEMM kernel spends its time in
// per-tile prologues / epilogues on this shape (K = 64: two slabs); here a block stages the whole projection matrix ONCE in LDS as split-bf16
// hi / lo tiles (split_bf16.h: hi*hi + hi*lo + lo*hi on mfma_f32_16x16x32_bf16, fp32 accumulate, ~1e-5 relative), streams rows straight from
// global memory into MFMA operands and writes 16-byte pieces of the result rows from the accumulators.  Traffic = read x, write dd (forward);
// read ddd (+ addend), write dx (adjoint).  The fp32 parity mode of the engine keeps the exact-fp32 GEMM.
#include "sa_common.h"
#include "split_bf16.h"

namespace sa {

struct ProjArgs {
    const float* x;        // forward: rows of 64 floats, see heads
    const float* proj;     // [m][64]
    float* out;            // forward: dd [rows][LDF];  adjoint: dx [rows][o_stride]
    const float* g;        // adjoint: ddd [rows][LDF]
    const float* addend;   // adjoint, optional: laid out like out (may be out itself)
    int64_t rows;
    int32_t m, LDF, x_stride, o_stride;
    unsigned long long* gmax;   // forward, optional: packed (value, index) maximum of dd over valid rows and columns < m
    float* feat;                // forward, QFEAT: query feature map [rows][LDF]
    float c2half, ratio, eps;
    int32_t heads;         // x / out rows are head blocks of wider rows: row r lives at (r / heads) * stride + (r % heads) * 64
};

__device__ __forceinline__ int64_t proj_row_off(int64_t r, int heads, int stride) { return (r / heads) * stride + (r % heads) * 64; }

// projection rows [0, nrows) -> hi / lo tiles [nrows][64] bf16 in the lroff() layout (rows >= m are zero)
// PERM: tile row rho holds projection row pi(rho), pi(ks*32 + b*16 + 4g + r) = ks*32 + 8g + 4b + r.  The transposing reads hand lane group g the
// tile rows {4g..4g+3, 16+4g..} of a 32-row block as its eight reduction steps; with pi those are the EIGHT CONSECUTIVE features ks*32 + 8g .. +7,
// so the gradient rows that multiply them are read as 32 contiguous bytes per lane (a full 128-byte line per row and 32-feature block).
template <bool PERM = false>
__device__ __forceinline__ void proj_stage(unsigned char* hi, unsigned char* lo, const float* proj, int m, int nrows, int tid) {
    // all loads first (nrows <= 288: at most 18 pieces of 16 bytes per thread), then the splits: a load -> split -> store loop with a run-time
    // trip count pays the L2 round trip once per iteration (18 us per block, measured as 2/3 of the adjoint kernel)
    float4 v[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        const int idx = tid + 256 * t, rho = min(idx >> 4, nrows - 1), c4 = idx & 15;
        const int f = PERM ? (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3) : rho;
        v[t] = *(const float4*)(proj + (int64_t)min(f, m - 1) * 64 + c4 * 4);
    }
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        const int idx = tid + 256 * t, rho = idx >> 4, c4 = idx & 15;
        if (rho < nrows) {
            const int f = PERM ? (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3) : rho;
            const float k = f < m ? 1.f : 0.f;
            uint2 h, l;
            split_pair(v[t].x * k, v[t].y * k, h.x, l.x);
            split_pair(v[t].z * k, v[t].w * k, h.y, l.y);
            const uint32_t o = lroff(rho, c4 * 4);
            *(uint2*)(hi + o) = h;
            *(uint2*)(lo + o) = l;
        }
    }
}

__device__ __forceinline__ float4 proj_keep(float4 x, bool keep) {
    const uint32_t mk = keep ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(x.x) & mk), __uint_as_float(__float_as_uint(x.y) & mk), __uint_as_float(__float_as_uint(x.z) & mk),
                       __uint_as_float(__float_as_uint(x.w) & mk));
}

// forward: block = 128 rows, wave = 32 rows (two MFMA column sets); D[i = feature][j = row]
// QFEAT: the QUERY feature map phi = ratio (exp(dd - |x|^2 c^2 / 2 - rowmax) + eps) in the same launch: the row maximum is known after one walk over
// the feature fragments, a second walk recomputes them (17 x 12 MFMAs: nothing next to the 2 x 73 MB of results) and writes phi -- the separate
// feature-map launch and its read of dd disappear.
template <bool QFEAT>
__global__ __launch_bounds__(256, 2) void favor_project_fwd_kernel(const ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nfr = a.LDF >> 4;
    unsigned char* const sPh = smem;
    unsigned char* const sPl = smem + nfr * 16 * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), qi = lane & 15, g = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 128 + w * 32;
    short8_t xh[2][2], xl[2][2];
    float ss[2] = {0.f, 0.f};   // QFEAT: |x|^2 of this lane's quarter of the row
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int64_t r = r0 + st * 16 + qi;
        const bool ok = r < a.rows;
        const float* xr = a.x + proj_row_off(ok ? r : a.rows - 1, a.heads, a.x_stride);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4 v0 = proj_keep(*(const float4*)(xr + ks * 32 + g * 8), ok), v1 = proj_keep(*(const float4*)(xr + ks * 32 + g * 8 + 4), ok);
            const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (QFEAT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss[st] = fmaf(xs[e], xs[e], ss[st]);
            }
            split8(xs, xh[st][ks], xl[st][ks]);
        }
    }
    proj_stage(sPh, sPl, a.proj, a.m, nfr * 16, tid);
    __syncthreads();
    float* o0 = a.out + (r0 + qi) * a.LDF + g * 4;
    float* o1 = o0 + (int64_t)16 * a.LDF;
    const bool ok0 = r0 + qi < a.rows, ok1 = r0 + 16 + qi < a.rows;
    float mx = -INFINITY;          // running maximum of this lane's results and its flat index (lowest index on ties)
    uint32_t mi = 0xffffffffu;
    float rm0 = -INFINITY, rm1 = -INFINITY;   // QFEAT: row maxima over the valid features
    const uint32_t i0 = (uint32_t)((r0 + qi) * a.LDF) + (uint32_t)g * 4u, i1 = i0 + 16u * (uint32_t)a.LDF;
    for (int f = 0; f < nfr; ++f) {
        short8_t ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t o = lroff(f * 16 + qi, ks * 32 + g * 8);
            ah[ks] = *(const short8_t*)(sPh + o);
            al[ks] = *(const short8_t*)(sPl + o);
        }
        float4_t c0 = (float4_t){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            c0 = mfma3(ah[ks], al[ks], xh[0][ks], xl[0][ks], c0);
            c1 = mfma3(ah[ks], al[ks], xh[1][ks], xl[1][ks], c1);
        }
        if (ok0) *(float4*)(o0 + f * 16) = make_float4(c0[0], c0[1], c0[2], c0[3]);
        if (ok1) *(float4*)(o1 + f * 16) = make_float4(c1[0], c1[1], c1[2], c1[3]);
        if (QFEAT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = f * 16 + g * 4 + r < a.m;
                rm0 = in ? fmaxf(rm0, c0[r]) : rm0;
                rm1 = in ? fmaxf(rm1, c1[r]) : rm1;
            }
        }
        if (a.gmax) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = f * 16 + g * 4 + r < a.m;
                const uint32_t j0 = i0 + (uint32_t)(f * 16 + r), j1 = i1 + (uint32_t)(f * 16 + r);
                const bool t0 = ok0 & in & ((c0[r] > mx) | ((c0[r] == mx) & (j0 < mi)));
                mx = t0 ? c0[r] : mx;
                mi = t0 ? j0 : mi;
                const bool t1 = ok1 & in & ((c1[r] > mx) | ((c1[r] == mx) & (j1 < mi)));
                mx = t1 ? c1[r] : mx;
                mi = t1 ? j1 : mi;
            }
        }
    }
    if (a.gmax) {   // one atomic per block
        __shared__ unsigned long long sbest[4];
        unsigned long long best = mi != 0xffffffffu ? pack_max(mx, mi) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ot = __shfl_xor(best, o, 64);
            best = ot > best ? ot : best;
        }
        if (lane == 0) sbest[w] = best;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 1; q < 4; ++q) best = sbest[q] > best ? sbest[q] : best;
            atomicMax(a.gmax, best);
        }
    }
    if (QFEAT) {
        // a row lives in the four lanes qi, qi + 16, qi + 32, qi + 48
        rm0 = fmaxf(rm0, __shfl_xor(rm0, 16, 64)); rm0 = fmaxf(rm0, __shfl_xor(rm0, 32, 64));
        rm1 = fmaxf(rm1, __shfl_xor(rm1, 16, 64)); rm1 = fmaxf(rm1, __shfl_xor(rm1, 32, 64));
        ss[0] += __shfl_xor(ss[0], 16, 64); ss[0] += __shfl_xor(ss[0], 32, 64);
        ss[1] += __shfl_xor(ss[1], 16, 64); ss[1] += __shfl_xor(ss[1], 32, 64);
        const float sh0 = ss[0] * a.c2half + rm0, sh1 = ss[1] * a.c2half + rm1;
        float* p0 = a.feat + (r0 + qi) * a.LDF + g * 4;
        float* p1 = p0 + (int64_t)16 * a.LDF;
        for (int f = 0; f < nfr; ++f) {
            short8_t ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t o = lroff(f * 16 + qi, ks * 32 + g * 8);
                ah[ks] = *(const short8_t*)(sPh + o);
                al[ks] = *(const short8_t*)(sPl + o);
            }
            float4_t c0 = (float4_t){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                c0 = mfma3(ah[ks], al[ks], xh[0][ks], xl[0][ks], c0);
                c1 = mfma3(ah[ks], al[ks], xh[1][ks], xl[1][ks], c1);
            }
            float e0[4], e1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = f * 16 + g * 4 + r < a.m;
                e0[r] = in ? a.ratio * (expf(c0[r] - sh0) + a.eps) : 0.f;
                e1[r] = in ? a.ratio * (expf(c1[r] - sh1) + a.eps) : 0.f;
            }
            if (ok0) *(float4*)(p0 + f * 16) = make_float4(e0[0], e0[1], e0[2], e0[3]);
            if (ok1) *(float4*)(p1 + f * 16) = make_float4(e1[0], e1[1], e1[2], e1[3]);
        }
    }
}

// adjoint: dx[r][d] = sum_f g[r][f] P[f][d] (+ addend); block = 128 rows, wave = 2 x 16 rows; D[i = d][j = row], reduction over the
// projection rows in accumulator-row order (transposing reads of the staged tile)
__global__ __launch_bounds__(256, 2) void favor_project_bwd_kernel(const ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nks = (a.LDF + 31) >> 5;
    unsigned char* const sPh = smem;
    unsigned char* const sPl = smem + nks * 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), qi = lane & 15, g = lane >> 4;
    proj_stage<true>(sPh, sPl, a.proj, a.m, nks * 32, tid);
    __syncthreads();
    const uint32_t trow = (uint32_t)g * 4u + ((uint32_t)qi >> 2), tcol = (uint32_t)(qi & 3) * 4u;
    for (int it = 0; it < 2; ++it) {
        const int64_t r = (int64_t)blockIdx.x * 128 + w * 32 + it * 16 + qi;
        const bool ok = r < a.rows;
        const float* gr = a.g + (ok ? r : a.rows - 1) * a.LDF;
        float4_t acc[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
        float4 v[9][2];   // the whole gradient row of this lane's quarter, in flight together (LDF <= 272: nine 32-feature blocks)
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            const int c0 = ks * 32 + g * 8, c1 = c0 + 4;   // (permuted tile rows: see proj_stage)
            v[ks][0] = proj_keep(*(const float4*)(gr + min(c0, a.LDF - 4)), ok && c0 < a.LDF);
            v[ks][1] = proj_keep(*(const float4*)(gr + min(c1, a.LDF - 4)), ok && c1 < a.LDF);
        }
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            if (ks < nks) {
                const float xs[8] = {v[ks][0].x, v[ks][0].y, v[ks][0].z, v[ks][0].w, v[ks][1].x, v[ks][1].y, v[ks][1].z, v[ks][1].w};
                short8_t bh, bl;
                split8(xs, bh, bl);
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const uint32_t o0 = lroff(ks * 32 + trow, df * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, df * 16 + tcol);
                    const short8_t ah = __builtin_shufflevector(lds_tr16_b64(sPh + o0), lds_tr16_b64(sPh + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                    const short8_t al = __builtin_shufflevector(lds_tr16_b64(sPl + o0), lds_tr16_b64(sPl + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[df] = mfma3(ah, al, bh, bl, acc[df]);
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                float4 o = make_float4(acc[df][0], acc[df][1], acc[df][2], acc[df][3]);
                if (a.addend) {
                    const float4 ad = *(const float4*)(a.addend + proj_row_off(r, a.heads, a.o_stride) + df * 16 + g * 4);
                    o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                }
                *(float4*)(a.out + proj_row_off(r, a.heads, a.o_stride) + df * 16 + g * 4) = o;
            }
        }
    }
}

// Feature-map backward fused with the projection adjoint (throughput mode): the intermediate ddd = d loss / d dd [rows][LDF] is never written.
//   v_f = (feat_f - ratio eps) dfeat_f  (f < m),   t = sum_f v_f,   dx = sum_f v_f P[f] - [query] t P[argmax_f dd] - t c^2 x
// (query rows are stabilised by their own maximum, which takes -t; key rows by the GLOBAL maximum: they report t and one fix-up launch applies
// -(sum of all t) P[f*] to the row that holds it).  Same block shape as the adjoint above; the three feature rows stream through in groups of
// three 32-feature blocks (18 loads of 16 bytes in flight per lane).
struct FeatProjArgs {
    const float *dfeat, *feat, *dd, *x, *proj;
    float* dx;
    float* tsum;           // keys: one float, the sum of t over all rows (zeroed by the launcher)
    int64_t rows;
    int32_t m, LDF, x_stride, heads, is_query;
    float c2, ratio_eps;
};

__global__ __launch_bounds__(256, 2) void favor_feat_proj_bwd_kernel(const FeatProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nks = (a.LDF + 31) >> 5;
    unsigned char* const sPh = smem;
    unsigned char* const sPl = smem + nks * 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), qi = lane & 15, g = lane >> 4;
    proj_stage<true>(sPh, sPl, a.proj, a.m, nks * 32, tid);
    __syncthreads();
    const uint32_t trow = (uint32_t)g * 4u + ((uint32_t)qi >> 2), tcol = (uint32_t)(qi & 3) * 4u;
    float tblock = 0.f;
    for (int it = 0; it < 2; ++it) {
        const int64_t r = (int64_t)blockIdx.x * 128 + w * 32 + it * 16 + qi;
        const bool ok = r < a.rows;
        const int64_t rc = ok ? r : a.rows - 1;
        const float* pf = a.feat + rc * a.LDF;
        const float* pg = a.dfeat + rc * a.LDF;
        const float* pd = a.dd + rc * a.LDF;
        float4_t acc[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) acc[df] = (float4_t){0.f, 0.f, 0.f, 0.f};
        float tp = 0.f, mx = -INFINITY;
        int am = 0x7fffffff;
        // group grp + 1 is in flight while group grp is consumed (two register sets)
        float4 vf[2][3][2], vg[2][3][2], vd[2][3][2];
        auto load_group = [&](int grp, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = min((grp * 3 + k) * 32 + g * 8 + q * 4, a.LDF - 4);   // eight consecutive features per lane (permuted tile rows)
                    vf[b][k][q] = *(const float4*)(pf + c);
                    vg[b][k][q] = *(const float4*)(pg + c);
                    vd[b][k][q] = *(const float4*)(pd + c);
                }
        };
        load_group(0, 0);
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            const int b = grp & 1;
            if (grp < 2) load_group(grp + 1, b ^ 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ks = grp * 3 + k;
                if (ks < nks) {
                    float xs[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c0 = ks * 32 + g * 8 + q * 4;
                        const float f4[4] = {vf[b][k][q].x, vf[b][k][q].y, vf[b][k][q].z, vf[b][k][q].w}, g4[4] = {vg[b][k][q].x, vg[b][k][q].y, vg[b][k][q].z, vg[b][k][q].w};
                        const float d4[4] = {vd[b][k][q].x, vd[b][k][q].y, vd[b][k][q].z, vd[b][k][q].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int col = c0 + e;
                            const bool valid = ok & (col < a.m);
                            const float v = valid ? (f4[e] - a.ratio_eps) * g4[e] : 0.f;
                            tp += v;
                            // first maximum, like torch.max -- as selects (short-circuit && / || became 250 exec-mask branches)
                            const bool take = valid & ((d4[e] > mx) | ((d4[e] == mx) & (col < am)));
                            mx = take ? d4[e] : mx;
                            am = take ? col : am;
                            xs[q * 4 + e] = v;
                        }
                    }
                    short8_t bh, bl;
                    split8(xs, bh, bl);
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        const uint32_t o0 = lroff(ks * 32 + trow, df * 16 + tcol), o1 = lroff(ks * 32 + 16 + trow, df * 16 + tcol);
                        const short8_t ah = __builtin_shufflevector(lds_tr16_b64(sPh + o0), lds_tr16_b64(sPh + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                        const short8_t al = __builtin_shufflevector(lds_tr16_b64(sPl + o0), lds_tr16_b64(sPl + o1), 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[df] = mfma3(ah, al, bh, bl, acc[df]);
                    }
                }
            }
        }
        // the row lives in the four lanes qi, qi + 16, qi + 32, qi + 48
        float t = tp;
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            const bool take = (om > mx) | ((om == mx) & (oa < am));
            mx = take ? om : mx;
            am = take ? oa : am;
        }
        if (ok) {
            const float* xr = a.x + proj_row_off(r, a.heads, a.x_stride);
            float* dxr = a.dx + proj_row_off(r, a.heads, a.x_stride);
            const float* pa = a.proj + (int64_t)min(am, a.m - 1) * 64;
            const float ts = a.is_query ? t : 0.f;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const int d0 = df * 16 + g * 4;
                const float4 xv = *(const float4*)(xr + d0), pv = *(const float4*)(pa + d0);
                const float tc = t * a.c2;
                *(float4*)(dxr + d0) = make_float4(acc[df][0] - ts * pv.x - tc * xv.x, acc[df][1] - ts * pv.y - tc * xv.y, acc[df][2] - ts * pv.z - tc * xv.z,
                                                   acc[df][3] - ts * pv.w - tc * xv.w);
            }
        }
        if (!a.is_query && ok && g == 0) tblock += t;
    }
    if (!a.is_query) {   // sum of t over the block's rows -> one atomic (the fix-up launch used to re-read one value per row)
        __shared__ float sred[4];
        tblock = wave_sum(tblock);
        if (lane == 0) sred[w] = tblock;
        __syncthreads();
        if (tid == 0) unsafeAtomicAdd(a.tsum, (sred[0] + sred[1]) + (sred[2] + sred[3]));
    }
}

// keys: the global-max element (row*, f*) of dd takes -(sum_rows t): dx[row*] -= T P[f*]   (T = *total, accumulated by the kernel above)
__global__ __launch_bounds__(64) void favor_key_stab_dx_kernel(float* __restrict__ dx, int x_stride, int heads, const unsigned long long* __restrict__ gmax,
                                                               const float* __restrict__ total, const float* __restrict__ proj, int LDF) {
    const uint32_t idx = 0xffffffffu - (uint32_t)(*gmax & 0xffffffffull);
    const int64_t r = idx / (uint32_t)LDF;
    const int f = (int)(idx % (uint32_t)LDF);
    dx[proj_row_off(r, heads, x_stride) + threadIdx.x] -= total[0] * proj[(int64_t)f * 64 + threadIdx.x];
}

}  // namespace sa

using namespace sa;

extern "C" int sa_favor_features_project_bwd(const float* dfeat, const float* feat, const float* dd, const float* src, int src_stride, int heads,
                                             const float* proj, int is_query, float* dsrc, const void* gmax_ws, float* tsum_ws, int64_t rows, int m, int LDF,
                                             int dh, void* stream) {
    if (!dfeat || !feat || !dd || !src || !proj || !dsrc || rows <= 0 || m <= 0 || heads <= 0 || (!is_query && (!gmax_ws || !tsum_ws))) return SA_EINVAL;
    if (dh != 64 || (LDF & 15) || LDF < m || LDF > 272 || (src_stride & 3) || src_stride < dh * heads) return SA_EUNSUPPORTED;
    FeatProjArgs a = {};
    a.dfeat = dfeat; a.feat = feat; a.dd = dd; a.x = src; a.proj = proj; a.dx = dsrc; a.tsum = tsum_ws; a.rows = rows; a.m = m; a.LDF = LDF;
    a.x_stride = src_stride; a.heads = heads; a.is_query = is_query;
    const float c = powf((float)dh, -0.25f), ratio = 1.f / sqrtf((float)m);
    a.c2 = c * c; a.ratio_eps = ratio * 1e-4f;
    const size_t lds = (size_t)2 * ((LDF + 31) / 32) * 32 * 128;
    hipFuncSetAttribute((const void*)favor_feat_proj_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (!is_query) hipMemsetAsync(tsum_ws, 0, 4, (hipStream_t)stream);   // tsum_ws[0] accumulates the sum over all rows
    SA_LAUNCH(favor_feat_proj_bwd_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    if (!is_query) {
        SA_LAUNCH(favor_key_stab_dx_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dsrc, src_stride, heads, (const unsigned long long*)gmax_ws, tsum_ws,
                           proj, LDF);
        SA_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sa_favor_project(const float* x, int x_stride, int heads, const float* proj, float* dd, void* gmax_ws, int64_t rows, int m, int LDF, int dh,
                                void* stream) {
    if (!x || !proj || !dd || rows <= 0 || m <= 0 || heads <= 0) return SA_EINVAL;
    if (rows * LDF >= ((int64_t)1 << 32) - 1) return SA_EUNSUPPORTED;
    if (dh != 64 || (LDF & 15) || LDF < m || LDF > 272 || (x_stride & 3) || x_stride < dh * heads) return SA_EUNSUPPORTED;
    ProjArgs a = {};
    a.x = x; a.proj = proj; a.out = dd; a.rows = rows; a.m = m; a.LDF = LDF; a.x_stride = x_stride; a.heads = heads;
    a.gmax = (unsigned long long*)gmax_ws;
    if (gmax_ws) hipMemsetAsync(gmax_ws, 0, 8, (hipStream_t)stream);
    const size_t lds = (size_t)2 * LDF * 128;
    hipFuncSetAttribute((const void*)favor_project_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SA_LAUNCH(favor_project_fwd_kernel<false>, dim3((unsigned)((rows + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_project_features(const float* x, int x_stride, int heads, const float* proj, float* dd, float* feat, int64_t rows, int m, int LDF,
                                         int dh, void* stream) {
    if (!x || !proj || !dd || !feat || rows <= 0 || m <= 0 || heads <= 0) return SA_EINVAL;
    if (dh != 64 || (LDF & 15) || LDF < m || LDF > 272 || (x_stride & 3) || x_stride < dh * heads) return SA_EUNSUPPORTED;
    ProjArgs a = {};
    a.x = x; a.proj = proj; a.out = dd; a.feat = feat; a.rows = rows; a.m = m; a.LDF = LDF; a.x_stride = x_stride; a.heads = heads;
    const float c = powf((float)dh, -0.25f);
    a.c2half = 0.5f * c * c; a.ratio = 1.f / sqrtf((float)m); a.eps = 1e-4f;
    const size_t lds = (size_t)2 * LDF * 128;
    hipFuncSetAttribute((const void*)favor_project_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SA_LAUNCH(favor_project_fwd_kernel<true>, dim3((unsigned)((rows + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_favor_project_bwd(const float* ddd, const float* proj, const float* addend, float* dx, int dx_stride, int heads, int64_t rows, int m, int LDF,
                                    int dh, void* stream) {
    if (!ddd || !proj || !dx || rows <= 0 || m <= 0 || heads <= 0) return SA_EINVAL;
    if (dh != 64 || (LDF & 15) || LDF < m || LDF > 272 || (dx_stride & 3) || dx_stride < dh * heads) return SA_EUNSUPPORTED;
    ProjArgs a = {};
    a.g = ddd; a.proj = proj; a.addend = addend; a.out = dx; a.rows = rows; a.m = m; a.LDF = LDF; a.o_stride = dx_stride; a.heads = heads;
    const size_t lds = (size_t)2 * ((LDF + 31) / 32) * 32 * 128;
    hipFuncSetAttribute((const void*)favor_project_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SA_LAUNCH(favor_project_bwd_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}
