// First encoder layer nn.Conv3d(1 -> 128, k4 s2 p1) (+ReLU) of the VQ-VAE (reference src/networks/vqvae/baseline.py:218-226, level 0) and its
// weight / bias gradient, bf16, as kernels of their own.  With ONE input channel the 64 taps are the whole reduction: y[cell][co] =
// sum_t W[co][t] x[2 cell - 1 + t].  The generic implicit GEMM pads that reduction 8x (channels travel in 16-byte vectors); routing it
// through an explicit [cells][64] im2col matrix costs a 734 MB write plus a one-slab GEMM whose blocks are all prologue and epilogue
// (0.66 + 1.49 ms at 160x224x160, batch 8).  Here a block gathers the taps of 256 (forward) / 128 (gradient) consecutive cells of one
// output plane straight from the fp32 volume into an LDS tile ([cell][64 taps] bf16, 128-byte rows, lroff() swizzle) and feeds the MFMA from
// it; the weights (16 KiB) are register-resident MFMA operands, outputs leave through the quarter transpose as 32-byte pieces.
// HBM traffic = the output (forward) / the gradient (backward) once; the volume itself is 4 bytes per voxel and is re-read from L2.
#include "sa_common.h"
#define SA_LROFF_NO_HALF_SWAP   // this file's tiles are private to it and read with ds_read_b128 / transposing reads only: the round-5 form of lroff() (one VALU op less per address)
#include "split_bf16.h"

namespace sa {

struct C1Args {
    const float* x;        // [N, 2D, 2H, 2W] fp32
    const bf16_t* wpk;     // forward: [128][64] bf16 (sa_pack_weights operand of the layer as a 1x1x1 convolution over 64 tap channels)
    const float* bias;     // forward: [128] or NULL
    bf16_t* y;             // forward: [cells][128]
    const bf16_t* g;       // backward: [cells][128]
    float* dw;             // backward: [128][64] fp32, accumulated
    float* db;             // backward: [128] or NULL, accumulated
    int32_t N, D, H, W;    // OUTPUT grid
    int32_t act;
    const bf16_t* mask;    // forward, optional [cells][128]: outputs are zeroed where mask <= 0 (the ReLU mask of a data gradient)
    float* sum_out;        // forward, optional: += sum of the whole volume x (every voxel is the centre tap of exactly one cell)
    bf16_t* y_lp;          // f16 forward (conv1_fwd_f16_kernel), optional: a bf16 copy of y for the backward pass
    uint32_t cells;        // N*D*H*W
    FastDiv dW_, dH_, dD_;
};

typedef float float4u_t __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (global_load_dwordx4)

// taps (kd, kh, kw = 0..3) of NCELL consecutive cells starting at cell0 -> tile[cell][64] bf16; cells >= a.cells give zero rows.
// Lanes run along the cells (thread = one cell, 256 / NCELL threads share its 16 (kd, kh) rows): consecutive lanes read overlapping 16-byte
// windows 8 bytes apart of the same input row, i.e. a wave reads one contiguous 0.5 KiB stretch per (kd, kh).
template <int NCELL, bool F16 = false>
__device__ __forceinline__ void c1_gather(unsigned char* tile, const C1Args& a, uint32_t cell0, int tid, float* centre_sum = nullptr) {
    constexpr int PARTS = 256 / NCELL, PER = 16 / PARTS;
    const uint32_t cl = (uint32_t)tid % NCELL, part = (uint32_t)tid / NCELL;
    const uint32_t cell = cell0 + cl;
    const uint32_t cc = min(cell, a.cells - 1u);
    const uint32_t q = fdiv(cc, a.dW_);                  // cc = ((n*D + d)*H + h)*W + w
    const int w = (int)(cc - q * (uint32_t)a.W);
    const uint32_t hq = fdiv(q, a.dH_);
    const int h = (int)(q - hq * (uint32_t)a.H);
    const uint32_t n = fdiv(hq, a.dD_);
    const int d = (int)(hq - n * (uint32_t)a.D);
    // The four kw taps 2w - 1 .. 2w + 2 of a row: ONE 16-byte load per (kd, kh) for every lane, at a position clamped into the row (w = 0 reads from column 0,
    // w = W - 1 from column 2W - 4) and shifted into place afterwards -- no divergent border branch inside the loop, so the loads of a batch are issued back to
    // back and waited for once.  (Round 4 form: `if (inner) 16-byte load else four clamped scalar loads` per (kd, kh): the branch kept every load behind the
    // previous iteration's use -- sixteen dependent L2 round trips per thread, 0.64 of the waves' cycles waiting, 30 us per 256-cell block.)
    const int iw0 = 2 * w - 1;
    const int sh = w == 0 ? 1 : (w == a.W - 1 ? -1 : 0);   // the load window starts `sh` columns right of the first tap (+1: left border, -1: right border)
    const int iwl = iw0 + sh;                              // in [0, 2W - 4] for W >= 2
    constexpr int BATCH = PER < 8 ? PER : 8;
#pragma unroll
    for (int it0 = 0; it0 < PER; it0 += BATCH) {
        float4u_t t[BATCH];
        bool okv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const uint32_t kk = part * PER + it0 + u, kd = kk >> 2, kh = kk & 3u;
            const int id = 2 * d - 1 + (int)kd, ih = 2 * h - 1 + (int)kh;
            okv[u] = cell < a.cells && (unsigned)id < (unsigned)(2 * a.D) && (unsigned)ih < (unsigned)(2 * a.H);
            const int cd = min(max(id, 0), 2 * a.D - 1), chh = min(max(ih, 0), 2 * a.H - 1);
            const float* row = a.x + (((int64_t)n * 2 * a.D + cd) * 2 * a.H + chh) * 2 * a.W;
            t[u] = *(const float4u_t*)(row + iwl);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const uint32_t kk = part * PER + it0 + u, kd = kk >> 2, kh = kk & 3u;
            float v[4];
            // sh = +1: taps (-, t0, t1, t2);  sh = 0: (t0, t1, t2, t3);  sh = -1: (t1, t2, t3, -)
            v[0] = sh > 0 ? 0.f : (sh < 0 ? t[u][1] : t[u][0]);
            v[1] = sh > 0 ? t[u][0] : (sh < 0 ? t[u][2] : t[u][1]);
            v[2] = sh > 0 ? t[u][1] : (sh < 0 ? t[u][3] : t[u][2]);
            v[3] = sh > 0 ? t[u][2] : (sh < 0 ? 0.f : t[u][3]);
            const bool ok = okv[u];
            // every input voxel is covered exactly once by the taps {1,2}^3 of its cell
            if (centre_sum && ok && (kd == 1u || kd == 2u) && (kh == 1u || kh == 2u)) *centre_sum += v[1] + v[2];
            uint2 pk;
            pk.x = ok ? (F16 ? pack2<f16_t>(v[0], v[1]) : pack_bf16x2(v[0], v[1])) : 0u;
            pk.y = ok ? (F16 ? pack2<f16_t>(v[2], v[3]) : pack_bf16x2(v[2], v[3])) : 0u;
            *(uint2*)(tile + lroff(cl, kk * 4u)) = pk;
        }
    }
}

// forward: block = 256 consecutive cells, wave = 64 of them (four MFMA column sets); D[i = channel][j = cell]
// F16: taps, weights and the output are IEEE halves (the forward operand type of an f16 encoder chain), + an optional bf16 copy of the output
template <bool F16>
__device__ __forceinline__ void conv1_fwd_body(const C1Args& a, unsigned char* sX) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const uint32_t cell0 = blockIdx.x * 256u;   // (a persistent variant that fetches the weight operands once per block measured 7 % slower)
    short8_t wf[8][2];   // W[co = f*16 + fr][taps ks*32 + g*8 .. +7]
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[f][ks] = *(const short8_t*)(a.wpk + (f * 16 + fr) * 64 + ks * 32 + g * 8);
    float4_t bv[8];      // bias of this lane's accumulator rows
#pragma unroll
    for (int f = 0; f < 8; ++f) bv[f] = a.bias ? *(const float4_t*)(a.bias + f * 16 + g * 4) : (float4_t){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    c1_gather<256, F16>(sX, a, cell0, tid, a.sum_out ? &csum : nullptr);
    if (a.sum_out) {   // block-uniform
        __shared__ float red[4];
        csum = wave_sum(csum);
        if (lane == 0) red[w] = csum;
        __syncthreads();
        if (tid == 0) unsafeAtomicAdd(a.sum_out, (red[0] + red[1]) + (red[2] + red[3]));
    }
    __syncthreads();
#pragma unroll 2
    for (int cf = 0; cf < 4; ++cf) {
        const uint32_t cl = (uint32_t)w * 64u + cf * 16u + fr;   // this lane's cell (MFMA column, and the output row after the transpose)
        short8_t xb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xb[ks] = *(const short8_t*)(sX + lroff(cl, ks * 32 + g * 8));
        float4_t acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            acc[f] = bv[f];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (F16) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const half8_t*)&wf[f][ks], *(const half8_t*)&xb[ks], acc[f], 0, 0, 0);
                else acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f][ks], xb[ks], acc[f], 0, 0, 0);
            }
        }
        const uint32_t cell = cell0 + cl;
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // channels half*64 + g*16 .. +15 of this lane's cell
            float v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t0 = acc[half * 4 + 0][r], t1 = acc[half * 4 + 1][r], t2 = acc[half * 4 + 2][r], t3 = acc[half * 4 + 3][r];
                quarter_transpose(t0, t1, t2, t3);   // t[i'] = channel half*64 + g*16 + i'*4 + r
                v[r] = t0; v[4 + r] = t1; v[8 + r] = t2; v[12 + r] = t3;
            }
            if (cell < a.cells) {
                uint32_t pk[8], pl[8];
                u32x4 mk[2] = {(u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}};
                if (a.mask) {
                    const u32x4* mp = (const u32x4*)(a.mask + (int64_t)cell * 128 + half * 64 + g * 16);
                    mk[0] = mp[0];
                    mk[1] = mp[1];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x0 = v[2 * e], x1 = v[2 * e + 1];
                    if (a.act == SA_ACT_RELU) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                    const uint32_t mw = mk[e >> 2][e & 3];
                    x0 = __uint_as_float(mw << 16) > 0.f ? x0 : 0.f;
                    x1 = __uint_as_float(mw & 0xffff0000u) > 0.f ? x1 : 0.f;
                    pk[e] = F16 ? pack2<f16_t>(x0, x1) : pack_bf16x2(x0, x1);
                    if constexpr (F16) pl[e] = pack_bf16x2(x0, x1);
                }
                u32x4* o = (u32x4*)(a.y + (int64_t)cell * 128 + half * 64 + g * 16);
                o[0] = (u32x4){pk[0], pk[1], pk[2], pk[3]};
                o[1] = (u32x4){pk[4], pk[5], pk[6], pk[7]};
                if constexpr (F16) {
                    if (a.y_lp) {
                        u32x4* ol = (u32x4*)(a.y_lp + (int64_t)cell * 128 + half * 64 + g * 16);
                        ol[0] = (u32x4){pl[0], pl[1], pl[2], pl[3]};
                        ol[1] = (u32x4){pl[4], pl[5], pl[6], pl[7]};
                    }
                }
            }
        }
    }
}
__global__ __launch_bounds__(256, 3) void conv1_fwd_kernel(const C1Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[256 * 128];
    conv1_fwd_body<false>(a, sX);
}
__global__ __launch_bounds__(256, 3) void conv1_fwd_f16_kernel(const C1Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[256 * 128];
    conv1_fwd_body<true>(a, sX);
}

// gradient tile [128 cells][128 channels] bf16, 256-byte rows; 32-byte chunks XOR-swizzled with the row for the transposing reads
__device__ __forceinline__ uint32_t c1_goff(uint32_t m, uint32_t c) { return m * 256u + ((((c >> 4) ^ (m & 7u)) << 5) | ((c & 15u) << 1)); }

// dW[co][t] += sum_cells g[cell][co] x[2 cell - 1 + t],  db[co] += sum_cells g[cell][co].  Persistent blocks walk 128-cell tiles; wave w owns
// channels [32 w, 32 w + 32) x all 64 taps in registers for the whole walk and adds them to dw once at the end.
__global__ __launch_bounds__(256, 3) void conv1_wgrad_kernel(const C1Args a, uint32_t ntiles) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[128 * 128], sG[128 * 256];
    __shared__ float sDb[16][8];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, g = lane >> 4;
    const uint32_t trow = (uint32_t)g * 4u + ((uint32_t)fr >> 2), tcol = (uint32_t)(fr & 3) * 4u;
    float4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t cell0 = t * 128u;
        u32x4 gv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {   // piece = 8 channels of one cell; this thread's channel octet is tid & 15 throughout
            const uint32_t p = (uint32_t)tid + 256u * it, cl = p >> 4;
            const uint32_t cell = cell0 + cl;
            const u32x4 v = *(const u32x4*)(a.g + (int64_t)min(cell, a.cells - 1u) * 128 + (p & 15u) * 8u);
            const uint32_t mk = cell < a.cells ? 0xffffffffu : 0u;
            gv[it] = (u32x4){v[0] & mk, v[1] & mk, v[2] & mk, v[3] & mk};
        }
        __syncthreads();   // the previous tile has been consumed
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const uint32_t p = (uint32_t)tid + 256u * it, cl = p >> 4;
            *(u32x4*)(sG + c1_goff(cl, (p & 15u) * 8u)) = gv[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bsum[2 * e] += __uint_as_float(gv[it][e] << 16);
                bsum[2 * e + 1] += __uint_as_float(gv[it][e] & 0xffff0000u);
            }
        }
        c1_gather<128>(sX, a, cell0, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            short8_t ga[2], xb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t c = (uint32_t)(w * 32 + i * 16) + tcol;
                ga[i] = __builtin_shufflevector(lds_tr16_b64(sG + c1_goff(ks * 32 + trow, c)), lds_tr16_b64(sG + c1_goff(ks * 32 + 16 + trow, c)), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xb[j] = __builtin_shufflevector(lds_tr16_b64(sX + lroff(ks * 32 + trow, j * 16 + tcol)), lds_tr16_b64(sX + lroff(ks * 32 + 16 + trow, j * 16 + tcol)),
                                                0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[i], xb[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[i = channel][j = tap]: lane (tap fr, g) holds channels 4g .. 4g+3 of each fragment
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(a.dw + (w * 32 + i * 16 + g * 4 + r) * 64 + j * 16 + fr, acc[i][j][r]);
    if (a.db) {
        // threads with equal tid & 15 summed the same channel octet: lanes l, l^16, l^32 of a wave, then the four waves
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bsum[e] += __shfl_xor(bsum[e], 16, 64);
            bsum[e] += __shfl_xor(bsum[e], 32, 64);
        }
        __syncthreads();
        if (tid < 128) ((float*)sDb)[tid] = 0.f;
        __syncthreads();
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&sDb[lane][e], bsum[e]);
        }
        __syncthreads();
        if (tid < 128) unsafeAtomicAdd(a.db + tid, ((float*)sDb)[tid]);
    }
}

// Xc[cell][tap] (bf16) to HBM for the layers that still want the explicit matrix: the same gather, then the tile leaves LDS as linear 16-byte
// pieces (the direct form wrote 8 bytes per thread at 128-byte stride from lanes that each read a different input row)
__global__ __launch_bounds__(256, 3) void conv1_im2col_kernel(const C1Args a, bf16_t* __restrict__ gc, float* __restrict__ sum_out) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[256 * 128];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const uint32_t cell0 = blockIdx.x * 256u;
    float csum = 0.f;
    c1_gather<256>(sX, a, cell0, tid, sum_out ? &csum : nullptr);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const uint32_t p = (uint32_t)tid + 256u * it, cl = p >> 3, piece = p & 7u;   // 8 pieces of 16 bytes per cell
        if (cell0 + cl < a.cells) *(u32x4*)(gc + ((int64_t)(cell0 + cl) * 64 + piece * 8)) = *(const u32x4*)(sX + lroff(cl, piece * 8u));
    }
    if (sum_out) {
        csum = wave_sum(csum);
        if ((tid & 63) == 0) red[tid >> 6] = csum;
        __syncthreads();
        if (tid == 0) unsafeAtomicAdd(sum_out, red[0] + red[1] + red[2] + red[3]);
    }
}

static int c1_fill(C1Args& a, int N, int D, int H, int W, int cout);

// bf16 form of sa_convt1_im2col (csrc/convt1.hip dispatches here)
int conv1_im2col_bf16(const float* g, void* gc, float* db, int N, int D, int H, int W, hipStream_t stream) {
    C1Args a = {};
    const int rc = c1_fill(a, N, D, H, W, 128);
    if (rc) return rc;
    a.x = g;
    SA_LAUNCH(conv1_im2col_kernel, dim3((a.cells + 255u) / 256u), dim3(256), 0, stream, a, (bf16_t*)gc, db);
    SA_CHECK_LAUNCH();
    return 0;
}

static int c1_fill(C1Args& a, int N, int D, int H, int W, int cout) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return SA_EINVAL;
    if (cout != 128 || W < 2 || (int64_t)N * D * H * W >= ((int64_t)1 << 31) - 256) return SA_EUNSUPPORTED;   // (W >= 2: the gather's clamped 16-byte row window)
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.cells = (uint32_t)((int64_t)N * D * H * W);
    a.dW_ = make_fastdiv((uint32_t)W);
    a.dH_ = make_fastdiv((uint32_t)H);
    a.dD_ = make_fastdiv((uint32_t)D);
    return 0;
}

}  // namespace sa

using namespace sa;

extern "C" int sa_conv1_fwd(const float* x, const void* wpk, const float* bias, void* y, int N, int D, int H, int W, int cout, int act, void* stream) {
    if (!x || !wpk || !y) return SA_EINVAL;
    if (act != SA_ACT_NONE && act != SA_ACT_RELU) return SA_EUNSUPPORTED;
    C1Args a = {};
    const int rc = c1_fill(a, N, D, H, W, cout);
    if (rc) return rc;
    a.x = x; a.wpk = (const bf16_t*)wpk; a.bias = bias; a.y = (bf16_t*)y; a.act = act;
    SA_LAUNCH(conv1_fwd_kernel, dim3((a.cells + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

// the same with f16 operands and output (wpk = the weight packed as SA_F16), + an optional bf16 copy y_lp of the output for the backward pass
extern "C" int sa_conv1_fwd_f16(const float* x, const void* wpk, const float* bias, void* y, void* y_lp, int N, int D, int H, int W, int cout, int act, void* stream) {
    if (!x || !wpk || !y) return SA_EINVAL;
    if (act != SA_ACT_NONE && act != SA_ACT_RELU) return SA_EUNSUPPORTED;
    C1Args a = {};
    const int rc = c1_fill(a, N, D, H, W, cout);
    if (rc) return rc;
    a.x = x; a.wpk = (const bf16_t*)wpk; a.bias = bias; a.y = (bf16_t*)y; a.y_lp = (bf16_t*)y_lp; a.act = act;
    SA_LAUNCH(conv1_fwd_f16_kernel, dim3((a.cells + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_conv1_wgrad(const float* x, const void* g, float* dw, float* db, int N, int D, int H, int W, int cout, void* stream) {
    if (!x || !g || !dw) return SA_EINVAL;
    C1Args a = {};
    const int rc = c1_fill(a, N, D, H, W, cout);
    if (rc) return rc;
    a.x = x; a.g = (const bf16_t*)g; a.dw = dw; a.db = db;
    const uint32_t ntiles = (a.cells + 127u) / 128u;
    SA_LAUNCH(conv1_wgrad_kernel, dim3(ntiles < 768u ? ntiles : 768u), dim3(256), 0, (hipStream_t)stream, a, ntiles);
    SA_CHECK_LAUNCH();
    return 0;
}

// Backward of the LAST decoder layer nn.ConvTranspose3d(128 -> 1, k4 s2 p1) on the first layer's kernels: with G = d loss / d output [N,2D,2H,2W],
//   dx[cell][c] = sum_t G[2 cell - 1 + t] W[c][t]   is the first layer's FORWARD on the volume G with the transposed-convolution weight as [128][64], and
//   dW[c][t]    = sum_cells x[cell][c] G[2 cell - 1 + t]   is its WEIGHT GRADIENT with the layer input x in the role of the output gradient;
// db = sum G rides along the gather of the forward launch.  No [cells][64] im2col matrix, no generic one-slab GEMMs (4.0 -> 1.4 ms at batch 8).
extern "C" int sa_convt1_backward(const float* g, const void* x, const void* wpk, int mask_input, void* dx, float* dw, float* db, int N, int D, int H, int W,
                                  void* stream) {
    if (!g || !x || !wpk || !dx || !dw) return SA_EINVAL;
    C1Args a = {};
    const int rc = c1_fill(a, N, D, H, W, 128);
    if (rc) return rc;
    a.x = g; a.wpk = (const bf16_t*)wpk; a.bias = nullptr; a.y = (bf16_t*)dx; a.act = SA_ACT_NONE;
    a.mask = mask_input ? (const bf16_t*)x : nullptr;
    a.sum_out = db;
    SA_LAUNCH(conv1_fwd_kernel, dim3((a.cells + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    C1Args b = {};
    c1_fill(b, N, D, H, W, 128);
    b.x = g; b.g = (const bf16_t*)x; b.dw = dw; b.db = nullptr;
    const uint32_t ntiles = (b.cells + 127u) / 128u;
    SA_LAUNCH(conv1_wgrad_kernel, dim3(ntiles < 768u ? ntiles : 768u), dim3(256), 0, (hipStream_t)stream, b, ntiles);
    SA_CHECK_LAUNCH();
    return 0;
}
