// Final decoder layer: nn.ConvTranspose3d(128 -> 1, k4 s2 p1)  (reference src/networks/vqvae/baseline.py:283-293, last level).
//
// With ONE output channel the implicit GEMM wastes 15/16 of every MFMA tile and the 8 output-parity launches re-read the
// 128-channel input 8 times; the layer is HBM-bound (reads 256 B per input voxel, writes 32 B).  These direct kernels read
// every input row once, coalesced, with lanes = channel pairs, and keep the 64 taps x 2 channels of the weight (or of its
// gradient) in registers:
//   fwd   : per input-grid cell m the 27 neighbour rows x[m+d] feed the 8 outputs o = 2m + par  (each of the 64 taps once)
//   dgrad : dx[i][c] = sum_k g[2i-1+k] w[c][k]  (* relu mask)
//   wgrad : dw[c][k] += sum_i x[i][c] g[2i-1+k],  db += sum g
// out[o] = b + sum_{i,k : 2i-1+k = o} x[i] . w[:,k]   per dimension (k = 0..3).
#include <stdlib.h>

#include <atomic>

#include "sa_common.h"

namespace sa {

struct CT1Args {
    const void* x;      // [N, D, H, W, 128] T
    const float* w;     // [128][64] fp32  (ConvTranspose3d weight [Cin, 1, 4,4,4])
    const float* bias;  // [1] or NULL
    float* out;         // fwd: [N, 2D, 2H, 2W] fp32
    const float* g;     // bwd: [N, 2D, 2H, 2W] fp32 gradient wrt out
    void* dx;           // dgrad out [N, D, H, W, 128] T
    const void* mask;   // dgrad: relu mask tensor (same layout as x) or NULL
    float* dw;          // [128][64] fp32 (accumulates)
    float* db;          // [1] (accumulates) or NULL
    int32_t N, D, H, W;
    FastDiv dW_, dH_, dD_;
    uint32_t cells;
};

template <typename T>
__device__ __forceinline__ void load_pair(const T* p, float& a, float& b);
template <>
__device__ __forceinline__ void load_pair<float>(const float* p, float& a, float& b) {
    const float2 v = *(const float2*)p;
    a = v.x;
    b = v.y;
}
template <>
__device__ __forceinline__ void load_pair<bf16_t>(const bf16_t* p, float& a, float& b) {
    const uint32_t v = *(const uint32_t*)p;
    a = __uint_as_float(v << 16);
    b = __uint_as_float(v & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void store_pair(T* p, float a, float b);
template <>
__device__ __forceinline__ void store_pair<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
template <>
__device__ __forceinline__ void store_pair<bf16_t>(bf16_t* p, float a, float b) { *(uint32_t*)p = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

__device__ __forceinline__ void decode_cell(uint32_t m, const CT1Args& a, int& n, int& d, int& h, int& w) {
    uint32_t q = fdiv(m, a.dW_);
    w = (int)(m - q * a.W);
    uint32_t q2 = fdiv(q, a.dH_);
    h = (int)(q - q2 * a.H);
    n = (int)fdiv(q2, a.dD_);
    d = (int)(q2 - (uint32_t)n * a.D);
}

// per dimension tap k reads input offset DLT[k] and feeds output parity PAR[k]:  k=0 -> (+1, 1)  1 -> (0, 0)  2 -> (0, 1)  3 -> (-1, 0)
__device__ __forceinline__ constexpr int tap_dlt(int k) { return k == 0 ? 1 : (k == 3 ? -1 : 0); }
__device__ __forceinline__ constexpr int tap_par(int k) { return (k == 0 || k == 2) ? 1 : 0; }

__device__ __forceinline__ float bcast(float v, int srclane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane)); }

template <typename T>
__global__ __launch_bounds__(256) void convt1_fwd_kernel(const CT1Args a) {
    __shared__ float2 sw[64][64];  // [tap][lane] = weights of channels (2 lane, 2 lane + 1)
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int k = e >> 6, l = e & 63;
        sw[k][l] = make_float2(a.w[(2 * l) * 64 + k], a.w[(2 * l + 1) * 64 + k]);
    }
    __syncthreads();
    const float b = a.bias ? a.bias[0] : 0.f;
    const T* x = (const T*)a.x;
    for (uint32_t m = wave; m < a.cells; m += nwaves) {
        int n, d, h, w;
        decode_cell(m, a, n, d, h, w);
        float xa[27], xb[27];
#pragma unroll
        for (int r = 0; r < 27; ++r) {
            const int id = d + r / 9 - 1, ih = h + (r / 3) % 3 - 1, iw = w + r % 3 - 1;
            const bool ok = (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            // branch-free: a conditional load would make hipcc drain vmcnt at every join and serialise the 27 row fetches
            const int cd = min(max(id, 0), a.D - 1), ch = min(max(ih, 0), a.H - 1), cw = min(max(iw, 0), a.W - 1);
            load_pair<T>(x + ((((int64_t)n * a.D + cd) * a.H + ch) * a.W + cw) * 128 + 2 * lane, xa[r], xb[r]);
            xa[r] = ok ? xa[r] : 0.f;
            xb[r] = ok ? xb[r] : 0.f;
        }
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.f;
#pragma unroll
        for (int kd = 0; kd < 4; ++kd)
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int r = (tap_dlt(kd) + 1) * 9 + (tap_dlt(kh) + 1) * 3 + (tap_dlt(kw) + 1);
                    const int o = (tap_par(kd) * 2 + tap_par(kh)) * 2 + tap_par(kw);
                    const float2 wv = sw[(kd * 4 + kh) * 4 + kw][lane];
                    acc[o] = fmaf(xa[r], wv.x, fmaf(xb[r], wv.y, acc[o]));
                }
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = wave_sum(acc[o]);
        if (lane < 4) {  // lane = (pd, ph): write the two w-parities as one 8-byte store
            const int pd = lane >> 1, ph = lane & 1;
            const int64_t o = (((int64_t)n * 2 * a.D + 2 * d + pd) * 2 * a.H + 2 * h + ph) * 2 * a.W + 2 * w;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane == q) {
                    v0 = acc[q * 2];
                    v1 = acc[q * 2 + 1];
                }
            *(float2*)(a.out + o) = make_float2(v0 + b, v1 + b);
        }
    }
}

// lane k holds g[2i-1+k] of the current input voxel; broadcast with readlane
__device__ __forceinline__ float load_g_tap(const CT1Args& a, int n, int d, int h, int w, int lane) {
    const int kd = lane >> 4, kh = (lane >> 2) & 3, kw = lane & 3;
    const int od = 2 * d - 1 + kd, oh = 2 * h - 1 + kh, ow = 2 * w - 1 + kw;
    const bool ok = (unsigned)od < (unsigned)(2 * a.D) && (unsigned)oh < (unsigned)(2 * a.H) && (unsigned)ow < (unsigned)(2 * a.W);
    const int cd = min(max(od, 0), 2 * a.D - 1), ch = min(max(oh, 0), 2 * a.H - 1), cw = min(max(ow, 0), 2 * a.W - 1);
    const float v = a.g[(((int64_t)n * 2 * a.D + cd) * 2 * a.H + ch) * 2 * a.W + cw];
    return ok ? v : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void convt1_dgrad_kernel(const CT1Args a) {
    __shared__ float2 sw[64][64];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int k = e >> 6, l = e & 63;
        sw[k][l] = make_float2(a.w[(2 * l) * 64 + k], a.w[(2 * l + 1) * 64 + k]);
    }
    __syncthreads();
    T* dx = (T*)a.dx;
    const T* mk = (const T*)a.mask;
    for (uint32_t m = wave; m < a.cells; m += nwaves) {
        int n, d, h, w;
        decode_cell(m, a, n, d, h, w);
        const float gv = load_g_tap(a, n, d, h, w, lane);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float gk = bcast(gv, k);
            const float2 wv = sw[k][lane];
            s0 = fmaf(gk, wv.x, s0);
            s1 = fmaf(gk, wv.y, s1);
        }
        const int64_t o = (int64_t)m * 128 + 2 * lane;
        if (mk) {
            float ma, mb;
            load_pair<T>(mk + o, ma, mb);
            s0 = ma > 0.f ? s0 : 0.f;
            s1 = mb > 0.f ? s1 : 0.f;
        }
        store_pair<T>(dx + o, s0, s1);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void convt1_wgrad_kernel(const CT1Args a) {
    __shared__ float red[64][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    float a0[64], a1[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) a0[k] = a1[k] = 0.f;
    float gsum = 0.f;
    const T* x = (const T*)a.x;
    for (uint32_t m = wave; m < a.cells; m += nwaves) {
        int n, d, h, w;
        decode_cell(m, a, n, d, h, w);
        const float gv = load_g_tap(a, n, d, h, w, lane);
        // every output voxel o = 2i + par is covered exactly once by the taps (kd,kh,kw) in {1,2}^3 of its cell
        const int kd = lane >> 4, kh = (lane >> 2) & 3, kw = lane & 3;
        if ((kd == 1 || kd == 2) && (kh == 1 || kh == 2) && (kw == 1 || kw == 2)) gsum += gv;
        float xa, xb;
        load_pair<T>(x + (int64_t)m * 128 + 2 * lane, xa, xb);
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float gk = bcast(gv, k);
            a0[k] = fmaf(xa, gk, a0[k]);
            a1[k] = fmaf(xb, gk, a1[k]);
        }
    }
    // block reduction over the 4 waves (taking turns on one LDS tile), then one atomic per (channel, tap) per block
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        for (int turn = 0; turn < 4; ++turn) {
            if (wv == turn) {
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    const float v = pass ? a1[k] : a0[k];
                    red[lane][k] = turn == 0 ? v : red[lane][k] + v;
                }
            }
            __syncthreads();
        }
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int l = e >> 6, k = e & 63;
            unsafeAtomicAdd(a.dw + (2 * l + pass) * 64 + k, red[l][k]);
        }
        __syncthreads();
    }
    if (a.db) {
        gsum = wave_sum(gsum);
        if (lane == 0) unsafeAtomicAdd(a.db, gsum);
    }
}

static int fill_ct1(CT1Args& a, int N, int D, int H, int W) {
    const int64_t cells = (int64_t)N * D * H * W;
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || cells >= (1ll << 31)) return SA_EINVAL;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.dW_ = make_fastdiv(W); a.dH_ = make_fastdiv(H); a.dD_ = make_fastdiv(D);
    a.cells = (uint32_t)cells;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// GEMM route for the same layer (the default for bf16/fp32 at scale): the 64 taps become the channel dimension of a 1x1x1
// convolution that runs on the MFMA kernels of conv_fprop.hip / conv_wgrad.hip, and two thin kernels here move between the
// voxel grid of the output and the [cell][tap] matrices:
//   forward : P[cell][tap] = x[cell] . w[:,tap]  (sa_conv_fprop, 128 -> 64, fp32 out);  out[o] = b + sum of its 8 P entries (gather)
//   backward: Gc[cell][tap] = g[2 cell - 1 + tap]  (im2col, + db = sum g);  dx = Gc . w^T (* mask)  (sa_conv_fprop, 64 -> 128);
//             dw[c][tap] = sum_cell x[cell][c] Gc[cell][tap]  (sa_conv_wgrad)
// Every (cell, tap) pair feeds exactly one output voxel, so P / Gc are read / written once.
struct CT1Map {
    const float* p;     // gather: P [cells][64] fp32
    const float* bias;
    float* out;         // [N, 2D, 2H, 2W]
    const float* g;     // im2col: gradient wrt out
    void* gc;           // im2col: [cells][64] T
    float* db;
    int32_t N, D, H, W;
    FastDiv dW_, dH_, dD_, d2H_, d2D_;
    uint32_t cells, pairs;   // pairs = N * 2D * 2H * W  (two outputs along W per thread)
};

// per dimension: output o = 2q + p is fed by (cell, tap) = p ? {(q+1, 0), (q, 2)} : {(q, 1), (q-1, 3)}
// A block = a 3-D tile of 4 (od) x 8 (oh) x 16 (ow) outputs (thread = (od, oh, pair of ow)): the ~180 P rows it touches are touched by
// its own threads only, so every fetched line is used whole (a row-major thread order re-fetched each P row ~5x: 7.5 GB for 1.47 GB).
__global__ __launch_bounds__(256) void convt1_gather_kernel(const CT1Map a, uint32_t nct, uint32_t nht, uint32_t ndt) {
    const float b = a.bias ? a.bias[0] : 0.f;
    uint32_t t = blockIdx.x;
    const uint32_t ct = t % nct; t /= nct;
    const uint32_t ht = t % nht; t /= nht;
    const uint32_t dt = t % ndt;
    const int n = (int)(t / ndt);
    const int c = (int)(ct * 8 + (threadIdx.x & 7u)), oh = (int)(ht * 8 + ((threadIdx.x >> 3) & 7u)), od = (int)(dt * 4 + (threadIdx.x >> 6));
    if (c >= a.W || oh >= 2 * a.H || od >= 2 * a.D) return;
    const int pd = od & 1, qd = od >> 1, ph = oh & 1, qh = oh >> 1;
    float s0 = b, s1 = b;
#pragma unroll
    for (int ud = 0; ud < 2; ++ud) {
        const int id = pd ? qd + 1 - ud : qd - ud, kd = pd ? 2 * ud : 1 + 2 * ud;
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            const int ih = ph ? qh + 1 - uh : qh - uh, kh = ph ? 2 * uh : 1 + 2 * uh;
            const bool ok = (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H;
            const int cd = min(max(id, 0), a.D - 1), chh = min(max(ih, 0), a.H - 1);
            const float* row = a.p + ((((int64_t)n * a.D + cd) * a.H + chh) * a.W) * 64 + (kd * 4 + kh) * 4;
            // even output 2c: (c, kw=1), (c-1, kw=3);  odd output 2c+1: (c+1, kw=0), (c, kw=2)
            const int cm = max(c - 1, 0), cp = min(c + 1, a.W - 1);
            const float e0 = row[c * 64 + 1], e1 = row[cm * 64 + 3], o0 = row[cp * 64 + 0], o1 = row[c * 64 + 2];
            s0 += ok ? e0 + (c > 0 ? e1 : 0.f) : 0.f;
            s1 += ok ? o1 + (c + 1 < a.W ? o0 : 0.f) : 0.f;
        }
    }
    *(float2*)(a.out + ((((int64_t)n * 2 * a.D + od) * 2 * a.H + oh) * a.W + c) * 2) = make_float2(s0, s1);
}

template <typename T>
__global__ __launch_bounds__(256) void convt1_im2col_kernel(const CT1Map a) {
    float gsum = 0.f;
    T* gc = (T*)a.gc;
    const uint32_t total = a.cells * 16u;   // thread = (cell, kd, kh): the four kw taps are 4 consecutive output voxels
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const uint32_t cell = t >> 4, kd = (t >> 2) & 3u, kh = t & 3u;
        uint32_t q = fdiv(cell, a.dW_);
        const int w = (int)(cell - q * (uint32_t)a.W);
        uint32_t q2 = fdiv(q, a.dH_);
        const int h = (int)(q - q2 * (uint32_t)a.H);
        const int n = (int)fdiv(q2, a.dD_);
        const int d = (int)(q2 - (uint32_t)n * (uint32_t)a.D);
        const int od = 2 * d - 1 + (int)kd, oh = 2 * h - 1 + (int)kh;
        const bool ok = (unsigned)od < (unsigned)(2 * a.D) && (unsigned)oh < (unsigned)(2 * a.H);
        const int cd = min(max(od, 0), 2 * a.D - 1), chh = min(max(oh, 0), 2 * a.H - 1);
        const float* row = a.g + (((int64_t)n * 2 * a.D + cd) * 2 * a.H + chh) * 2 * a.W;
        float v[4];
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int ow = 2 * w - 1 + kw;
            const float x = row[min(max(ow, 0), 2 * a.W - 1)];
            v[kw] = ok && (unsigned)ow < (unsigned)(2 * a.W) ? x : 0.f;
        }
        // every output voxel is covered exactly once by the taps {1,2}^3 of its cell
        if ((kd == 1u || kd == 2u) && (kh == 1u || kh == 2u)) gsum += v[1] + v[2];
        if constexpr (sizeof(T) == 4) {
            *(float4*)((float*)gc + (size_t)t * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            *(uint2*)((bf16_t*)gc + (size_t)t * 4) = pk;
        }
    }
    if (a.db) {
        __shared__ float red[4];
        gsum = wave_sum(gsum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsum;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(a.db, red[0] + red[1] + red[2] + red[3]);
    }
}

static int fill_map(CT1Map& a, int N, int D, int H, int W) {
    const int64_t cells = (int64_t)N * D * H * W;
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || cells * 16 >= (1ll << 32)) return SA_EINVAL;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.dW_ = make_fastdiv(W); a.dH_ = make_fastdiv(H); a.dD_ = make_fastdiv(D);
    a.d2H_ = make_fastdiv(2 * H); a.d2D_ = make_fastdiv(2 * D);
    a.cells = (uint32_t)cells;
    a.pairs = (uint32_t)(cells * 4);
    return 0;
}

int conv1_im2col_bf16(const float* g, void* gc, float* db, int N, int D, int H, int W, hipStream_t stream);   // csrc/conv1.hip


// ---- forward in ONE kernel (bf16): no [cells][64] product matrix in HBM -------------------------------------------------------------------------
// P[cell][tap] = x[cell] . W[:, tap] is what the GEMM route writes (1.47 GB fp32 at batch 8) and convt1_gather_kernel reads back.  Here a block owns
// an 8 x 8 patch of cells in (H, W) and walks the depth: per plane it computes P for the patch and its one-cell halo (10 x 10 cells x 64 taps, MFMA:
// the 64 x 128 weight is a register-resident A operand, the activation rows stream from global memory straight into B operands), keeps the planes
// d - 1 and d in LDS and emits the two output planes 2d - 1 and 2d they determine (per dimension output o = 2q + p is fed by
// (cell, tap) = p ? {(q+1, 0), (q, 2)} : {(q, 1), (q-1, 3)}).  Cells outside the volume load as zeros, so only the depth needs validity flags.
// HBM traffic: the input 1.56 x (the halo in H and W), the output once.
constexpr int CF_TH = 8, CF_TW = 8, CF_PH = CF_TH + 2, CF_PW = CF_TW + 2, CF_NP = CF_PH * CF_PW;   // 100 patch cells per plane
constexpr int CF_LD = 68;                                                                            // floats per P row (64 taps + 4: bank spread)

struct CFArgs {
    const bf16_t* x;     // [N, D, H, W, 128]
    const bf16_t* wpk;   // [>= 64 taps][128 channels] bf16 (the taps-as-output-channels operand)
    const float* bias;
    float* out;          // [N, 2D, 2H, 2W]
    int32_t N, D, H, W, nhp, nwp;
    int32_t dsplit, dchunk;   // the depth walk is cut into `dsplit` ranges of `dchunk` steps (a step d emits output planes 2d - 1 and 2d), one block each
};

__global__ __launch_bounds__(256, 3) void convt1_fused_fwd_kernel(const CFArgs a) {
    __shared__ __attribute__((aligned(16))) float sP[2][CF_NP * CF_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    int t = blockIdx.x;
    const int wp = t % a.nwp; t /= a.nwp;
    const int hp = t % a.nhp; t /= a.nhp;
    const int ds = t % a.dsplit;
    const int n = t / a.dsplit;
    const int h0 = hp * CF_TH, w0 = wp * CF_TW;
    // steps [d_lo, d_hi) of the depth walk 0 .. D; a range that starts inside the volume first rebuilds the plane below it (d_lo - 1) without emitting anything
    const int d_lo = ds * a.dchunk, d_hi = min(a.D + 1, d_lo + a.dchunk), d_first = max(d_lo - 1, 0);
    // weights: A operand, tap tile tt (16 taps) x k-step ks (32 channels)
    short8_t wf[4][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[tt][ks] = *(const short8_t*)(a.wpk + (tt * 16 + fr) * 128 + ks * 32 + g * 8);
    // this lane's cells: column tiles ct = w and w + 4 (7 tiles of 16 cover the 100 patch cells)
    int64_t cbase[2];
    bool cok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int cell = (w + 4 * q) * 16 + fr;
        const int ph = cell / CF_PW, pw = cell - ph * CF_PW;
        const int gh = h0 - 1 + ph, gw = w0 - 1 + pw;
        cok[q] = (w + 4 * q) < 7 && cell < CF_NP && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
        cbase[q] = (((int64_t)n * a.D * a.H + (cok[q] ? gh : 0)) * a.W + (cok[q] ? gw : 0)) * 128 + g * 8;   // + d * H * W * 128 + ks * 32
    }
    const int64_t plane = (int64_t)a.H * a.W * 128;
    u32x4 xb[2][4];
    auto load_plane = [&](int d) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 v = (u32x4){0u, 0u, 0u, 0u};
                if (cok[q] && d < a.D) v = *(const u32x4*)(a.x + cbase[q] + d * plane + ks * 32);
                xb[q][ks] = v;
            }
    };
    load_plane(d_first);
    // gather role: thread = output (oh_l, ow_l) of the 16 x 16 output tile; per dimension two (cell, tap) pairs
    const int oh_l = tid >> 4, ow_l = tid & 15;
    const int oh = 2 * h0 + oh_l, ow = 2 * w0 + ow_l;
    const bool o_ok = oh < 2 * a.H && ow < 2 * a.W;
    int prow[2], ptap_h[2], pcol[2], ptap_w[2];
    {
        const int ph_ = oh_l & 1, qh = oh_l >> 1;          // local cell index qh (0..7) -> patch row qh + 1
        prow[0] = ph_ ? qh + 2 : qh + 1; ptap_h[0] = ph_ ? 0 : 1;
        prow[1] = ph_ ? qh + 1 : qh;     ptap_h[1] = ph_ ? 2 : 3;
        const int pw_ = ow_l & 1, qw = ow_l >> 1;
        pcol[0] = pw_ ? qw + 2 : qw + 1; ptap_w[0] = pw_ ? 0 : 1;
        pcol[1] = pw_ ? qw + 1 : qw;     ptap_w[1] = pw_ ? 2 : 3;
    }
    const float bias = a.bias ? a.bias[0] : 0.f;
    for (int d = d_first; d < d_hi; ++d) {
        float* cur = sP[d & 1];
        const float* prev = sP[(d & 1) ^ 1];
        if (d < a.D) {
            // P[cell][tap] of plane d for this wave's column tiles
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (w + 4 * q >= 7) continue;
                float4_t acc[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tt][ks], *(const short8_t*)&xb[q][ks], acc[tt], 0, 0, 0);
                const int cell = (w + 4 * q) * 16 + fr;
                if (cell < CF_NP) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) *(float4_t*)(cur + cell * CF_LD + tt * 16 + g * 4) = acc[tt];
                }
            }
        }
        __syncthreads();
        if (d + 1 < a.D) load_plane(d + 1);     // in flight during the gather
        // outputs od = 2d - 1 (cells d: kd 0, d - 1: kd 2) and od = 2d (cells d: kd 1, d - 1: kd 3)
        const bool cur_ok = d < a.D, prev_ok = d > 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int od = 2 * d - 1 + e;
            if (od < 0 || od >= 2 * a.D || !o_ok || d < d_lo) continue;
            float sum = bias;
#pragma unroll
            for (int uh = 0; uh < 2; ++uh)
#pragma unroll
                for (int uw = 0; uw < 2; ++uw) {
                    const int cell = prow[uh] * CF_PW + pcol[uw];
                    const int tap_hw = ptap_h[uh] * 4 + ptap_w[uw];
                    if (cur_ok) sum += cur[cell * CF_LD + (e ? 1 : 0) * 16 + tap_hw];
                    if (prev_ok) sum += prev[cell * CF_LD + (e ? 3 : 2) * 16 + tap_hw];
                }
            a.out[(((int64_t)n * 2 * a.D + od) * 2 * a.H + oh) * 2 * a.W + ow] = sum;
        }
        __syncthreads();                        // this plane's readers are done before the buffer of plane d - 1 is overwritten
    }
}

}  // namespace sa

using namespace sa;

extern "C" int sa_convt1_fwd(const void* x, int dtype, const float* w, const float* bias, float* out, int N, int D, int H, int W, int C, void* stream) {
    if (!x || !w || !out) return SA_EINVAL;
    if (C != 128 || (dtype != SA_F32 && dtype != SA_BF16)) return SA_EUNSUPPORTED;
    CT1Args a = {};
    if (fill_ct1(a, N, D, H, W)) return SA_EINVAL;
    a.x = x; a.w = w; a.bias = bias; a.out = out;
    const dim3 grid(2048);
    if (dtype == SA_F32) SA_LAUNCH(convt1_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else SA_LAUNCH(convt1_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_convt1_bwd(const void* x, int dtype, const float* w, const float* g, const void* relu_mask, void* dx, float* dw, float* db, int N,
                             int D, int H, int W, int C, void* stream) {
    if (!x || !w || !g || !dw) return SA_EINVAL;
    if (C != 128 || (dtype != SA_F32 && dtype != SA_BF16)) return SA_EUNSUPPORTED;
    CT1Args a = {};
    if (fill_ct1(a, N, D, H, W)) return SA_EINVAL;
    a.x = x; a.w = w; a.g = g; a.dx = dx; a.mask = relu_mask; a.dw = dw; a.db = db;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if (dtype == SA_F32) SA_LAUNCH(convt1_dgrad_kernel<float>, dim3(2048), dim3(256), 0, st, a);
        else SA_LAUNCH(convt1_dgrad_kernel<bf16_t>, dim3(2048), dim3(256), 0, st, a);
        SA_CHECK_LAUNCH();
    }
    if (dtype == SA_F32) SA_LAUNCH(convt1_wgrad_kernel<float>, dim3(512), dim3(256), 0, st, a);
    else SA_LAUNCH(convt1_wgrad_kernel<bf16_t>, dim3(512), dim3(256), 0, st, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_convt1_gather(const float* p, const float* bias, float* out, int N, int D, int H, int W, void* stream) {
    if (!p || !out) return SA_EINVAL;
    CT1Map a = {};
    if (fill_map(a, N, D, H, W)) return SA_EINVAL;
    a.p = p; a.bias = bias; a.out = out;
    const uint32_t nct = ((uint32_t)W + 7u) / 8u, nht = (2u * (uint32_t)H + 7u) / 8u, ndt = (2u * (uint32_t)D + 3u) / 4u;
    const uint64_t blocks = (uint64_t)N * ndt * nht * nct;
    if (blocks >= (1ull << 31)) return SA_EINVAL;
    SA_LAUNCH(convt1_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, nct, nht, ndt);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_convt1_im2col(const float* g, int dtype, void* gc, float* db, int N, int D, int H, int W, void* stream) {
    if (!g || !gc) return SA_EINVAL;
    if (dtype != SA_F32 && dtype != SA_BF16) return SA_EUNSUPPORTED;
    CT1Map a = {};
    if (fill_map(a, N, D, H, W)) return SA_EINVAL;
    a.g = g; a.gc = gc; a.db = db;
    unsigned blocks = (unsigned)(((uint64_t)a.cells * 16u + 255u) / 256u);
    if (blocks > 4096u) blocks = 4096u;   // one atomic per block for db
    if (dtype == SA_BF16 && !dbg(SA_DBG_IM2COL_DIRECT)) {   // gather through an LDS tile, coalesced on both sides (csrc/conv1.hip)
        const int rc = conv1_im2col_bf16(g, gc, db, N, D, H, W, (hipStream_t)stream);
        if (rc != SA_EUNSUPPORTED) return rc;
    }
    if (dtype == SA_F32) SA_LAUNCH(convt1_im2col_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else SA_LAUNCH(convt1_im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

// forward of the last decoder layer in one launch (bf16 activations, 128 input channels): out [N,2D,2H,2W] fp32 = bias + sum_{cell, tap} x[cell] . W[:, tap];
// wpk = the layer's weight as [taps][128 channels] bf16 (sa_pack_weights of the taps-as-output-channels 1x1x1 convolution: rows = 64, red = 128)
extern "C" int sa_convt1_fused_fwd(const void* x, const void* wpk, const float* bias, float* out, int N, int D, int H, int W, void* stream) {
    if (!x || !wpk || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0) return SA_EINVAL;
    CFArgs a = {};
    a.x = (const bf16_t*)x; a.wpk = (const bf16_t*)wpk; a.bias = bias; a.out = out; a.N = N; a.D = D; a.H = H; a.W = W;
    a.nhp = (H + CF_TH - 1) / CF_TH; a.nwp = (W + CF_TW - 1) / CF_TW;
    int64_t blocks = (int64_t)N * a.nhp * a.nwp;
    // Depth ranges: a block walks its patch through D + 1 steps, three blocks per CU.  Config 2 at batch 8 is 1 120 patches = 1.46 rounds of the 768 slots, i.e.
    // two full-length rounds with the second half empty; cut into two ranges (one rebuilt plane each) it is 2.92 rounds of half the length (631 -> 485 us).
    static std::atomic<int> cu_cache{0};
    int cus = cu_cache.load(std::memory_order_relaxed);
    if (cus <= 0) {
        int dev = 0;
        cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cu_cache.store(cus, std::memory_order_relaxed);
    }
    const int64_t slots = 3ll * cus;
    int best = 1;
    int64_t best_cost = -1;
    for (int sp = 1; sp <= 8; ++sp) {
        const int chunk = (D + 1 + sp - 1) / sp;
        if (sp > 1 && chunk < 8) break;
        const int64_t cost = ((blocks * sp + slots - 1) / slots) * (chunk + (sp > 1 ? 1 : 0));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sp; }
    }
    a.dsplit = best;
    a.dchunk = (D + 1 + best - 1) / best;
    blocks *= best;
    if (blocks >= ((int64_t)1 << 31)) return SA_EUNSUPPORTED;
    SA_LAUNCH(convt1_fused_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}
