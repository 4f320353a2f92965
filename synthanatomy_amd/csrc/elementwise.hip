// Small HBM-bound kernels: weight packing, dtype/channel-pad casts, MSE loss (+grad), Adam.
#include "sa_common.h"

#include <cstring>
#include <mutex>
#include <string>

namespace sa {

thread_local hipError_t g_last_error = hipSuccess;
thread_local char g_last_conv_kernel[128] = "";
std::atomic<bool> g_kernel_log_on{false};
static std::mutex g_kernel_log_mu;
static std::string* g_kernel_log = nullptr;   // newline-separated, each distinct name once, in first-launch order
void note_kernel_slow(const char* name) {
    std::lock_guard<std::mutex> lk(g_kernel_log_mu);
    if (!g_kernel_log) g_kernel_log = new std::string();
    std::string key(name);
    // "(kernel<args>)" -> "kernel<args>" (template instances are passed to the launch macro in parentheses)
    if (key.size() > 2 && key.front() == '(' && key.back() == ')') key = key.substr(1, key.size() - 2);
    key.push_back('\n');
    if (g_kernel_log->rfind(key, 0) == 0 || g_kernel_log->find("\n" + key) != std::string::npos) return;
    g_kernel_log->append(key);
}

static uint32_t flags_from_env() {
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    uint32_t f = 0;
    if (on("SA_NO_HALO")) f |= SA_DBG_NO_HALO;
    if (on("SA_NO_HALO256")) f |= SA_DBG_NO_HALO256;
    if (on("SA_NO_HALO256_FUSE")) f |= SA_DBG_NO_HALO256_FUSE;
    if (on("SA_NO_DMA")) f |= SA_DBG_NO_DMA;
    if (on("SA_NO_SMALL_TILES")) f |= SA_DBG_NO_SMALL_TILES;
    if (on("SA_NO_FUSED_DB")) f |= SA_DBG_NO_FUSED_DB;
    if (getenv("SA_WGRAD_HALO9") && num("SA_WGRAD_HALO9", 1) == 0) f |= SA_DBG_NO_WGRAD_HALO9;
    if (on("SA_IM2COL_DIRECT")) f |= SA_DBG_IM2COL_DIRECT;
    if (on("SA_HALO256_4W")) f |= SA_DBG_HALO256_4W;
    if (on("SA_DENSE_NARROW")) f |= SA_DBG_DENSE_NARROW;
    if (on("SA_DETERMINISTIC")) f |= SA_DBG_DETERMINISTIC;
    if (on("SA_NO_KGROUPS")) f |= SA_DBG_NO_KGROUPS;
    if (on("SA_FAVOR_SEQ_ALWAYS")) f |= SA_DBG_FAVOR_SEQ_ALWAYS;
    if (on("SA_NO_CELLS256")) f |= SA_DBG_NO_CELLS256;
    if (on("SA_NO_CLASS_LAUNCH")) f |= SA_DBG_NO_CLASS_LAUNCH;
    if (on("SA_SCAN_VALU")) f |= SA_DBG_SCAN_VALU;
    if (num("SA_LOCAL_ATTN_EXACT", 0) == 1) f |= SA_DBG_LOCAL_ATTN_EXACT;
    f |= ((uint32_t)num("SA_SCAN_EXACT", 0) & 7u) << SA_DBG_SCAN_EXACT_SHIFT;
    return f;
}
static Tunables tunables_from_env() {
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    return Tunables{num("SA_WGRAD_ROWS", 10240), num("SA_WGRAD_MIN_BLOCKS", 0), num("SA_WGRAD_HALO_SPLITS", 0), (uint32_t)num("SA_PP_DBG", 0)};
}
std::atomic<uint32_t> g_debug_flags{flags_from_env()};
const Tunables g_tunables = tunables_from_env();

struct PackArgs {
    const float* w;
    void* wpk;
    int32_t lut[SA_MAX_TAPS];
    int64_t s_row, s_red;
    int32_t dtype, rows, red, ntaps, rows_pad, red_stride, Kpad;
};

// wpk[r][t*red_stride + c] = w[r*s_row + c*s_red + lut[t]]  (zero in every padded position)
__global__ void pack_weights_kernel(const PackArgs a) {
    const int64_t total = (int64_t)a.rows_pad * a.Kpad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / a.Kpad);
        const int k = (int)(e - (int64_t)r * a.Kpad);
        const int t = k / a.red_stride;
        const int c = k - t * a.red_stride;
        float v = 0.f;
        if (r < a.rows && t < a.ntaps && c < a.red) v = a.w[r * a.s_row + c * a.s_red + a.lut[t]];
        store_from_f32(a.wpk, a.dtype, e, v);
    }
}

// many packs in one launch: block -> descriptor by binary search in block_first.  Each block walks 64 x 64 tiles of the packed operand through
// LDS so that BOTH sides are coalesced: the data-gradient operand of a Linear is the transpose of the parameter (s_row = 1), which a direct
// gather reads 4 bytes per 2 KiB stride.
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const sa_pack_desc* __restrict__ table, const int32_t* __restrict__ block_first, int n) {
    __shared__ int s_desc;
    __shared__ float tile[64][65];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (block_first[mid] <= (int)blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        s_desc = lo;
    }
    __syncthreads();
    const sa_pack_desc& a = table[s_desc];
    const int b0 = block_first[s_desc], nb = block_first[s_desc + 1] - b0;
    const int tr = (a.rows_pad + 63) >> 6, tk = (a.Kpad + 63) >> 6;
    const bool row_fast = a.s_row < a.s_red;   // consecutive rows are adjacent in the parameter: read with the row index on the lanes
    const int lo6 = threadIdx.x & 63, hi2 = threadIdx.x >> 6;
    // Linear / 1x1x1 operands (one tap, unit stride along one side, everything a multiple of four): 16-byte reads and 8 / 16-byte writes.
    // Same values and the same rounding as the element-wise walk below, which every other operand keeps.
    const bool lin = a.ntaps == 1 && a.tap_lut[0] == 0 && a.red_stride >= a.red && (a.Kpad & 3) == 0 && (((uintptr_t)a.w | (uintptr_t)a.wpk) & 15) == 0 &&
                     ((a.s_red == 1 && (a.s_row & 3) == 0 && (a.red & 3) == 0) || (a.s_row == 1 && (a.s_red & 3) == 0 && (a.rows & 3) == 0));
    if (lin) {
        const int q4 = (threadIdx.x & 15) << 2, hi4 = threadIdx.x >> 4;
        for (int t = (int)blockIdx.x - b0; t < tr * tk; t += nb) {
            const int r0 = (t / tk) << 6, k0 = (t % tk) << 6;
            __syncthreads();
            float4 v[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int o = hi4 + 16 * p;
                const int r = r0 + (row_fast ? q4 : o), k = k0 + (row_fast ? o : q4);
                v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < a.rows && k < a.red) v[p] = *reinterpret_cast<const float4*>(a.w + r * a.s_row + k * a.s_red);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int o = hi4 + 16 * p;
                if (row_fast) {
                    tile[q4][o] = v[p].x, tile[q4 + 1][o] = v[p].y, tile[q4 + 2][o] = v[p].z, tile[q4 + 3][o] = v[p].w;
                } else {
                    tile[o][q4] = v[p].x, tile[o][q4 + 1] = v[p].y, tile[o][q4 + 2] = v[p].z, tile[o][q4 + 3] = v[p].w;
                }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rl = hi4 + 16 * p, r = r0 + rl, k = k0 + q4;
                if (r < a.rows_pad && k < a.Kpad) {
                    const float x0 = tile[rl][q4], x1 = tile[rl][q4 + 1], x2 = tile[rl][q4 + 2], x3 = tile[rl][q4 + 3];
                    const int64_t e = (int64_t)r * a.Kpad + k;
                    if (a.dtype == SA_F32) *reinterpret_cast<float4*>((float*)a.wpk + e) = make_float4(x0, x1, x2, x3);
                    else *reinterpret_cast<uint2*>((bf16_t*)a.wpk + e) = make_uint2(pack2_dt(a.dtype, x0, x1), pack2_dt(a.dtype, x2, x3));
                }
            }
        }
        return;
    }
    for (int t = (int)blockIdx.x - b0; t < tr * tk; t += nb) {
        const int r0 = (t / tk) << 6, k0 = (t % tk) << 6;
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int rl = row_fast ? lo6 : hi2 + 4 * p, kl = row_fast ? hi2 + 4 * p : lo6;
            const int r = r0 + rl, k = k0 + kl;
            const int tp = k / a.red_stride, c = k - tp * a.red_stride;
            float v = 0.f;
            if (r < a.rows && k < a.Kpad && tp < a.ntaps && c < a.red) v = a.w[r * a.s_row + c * a.s_red + a.tap_lut[tp]];
            tile[rl][kl] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int rl = hi2 + 4 * p, r = r0 + rl, k = k0 + lo6;
            if (r < a.rows_pad && k < a.Kpad) store_from_f32(a.wpk, a.dtype, (int64_t)r * a.Kpad + k, tile[rl][lo6]);
        }
    }
}

__global__ void cast_pad_kernel(const void* src, int src_dtype, int src_c, void* dst, int dst_dtype, int dst_stride, int64_t rows) {
    const int64_t total = rows * dst_stride;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / dst_stride;
        const int c = (int)(e - r * dst_stride);
        const float v = c < src_c ? load_as_f32(src, src_dtype, r * src_c + c) : 0.f;
        store_from_f32(dst, dst_dtype, e, v);
    }
}

__global__ void cast_f32_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n8) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = src[2 * e], b = src[2 * e + 1];
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
        o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
        o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
        o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
        dst[e] = o;
    }
}

__global__ void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ loss_sum, float* __restrict__ grad,
                           float gcoef) {
    float s = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[e] - b[e];
        s += d * d;
        if (grad) grad[e] = d * gcoef;
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.Adam (no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__device__ __forceinline__ void adam_one(float& pe, float gr, float& me, float& ve, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                         float gscale) {
    gr *= gscale;
    if (wd != 0.f) gr += wd * pe;
    me = b1 * me + (1.f - b1) * gr;
    ve = b2 * ve + (1.f - b2) * gr * gr;
    const float denom = sqrtf(ve) / bc2_sqrt + eps;
    pe = pe - (lr / bc1) * (me / denom);
}

// HBM-bound (28 bytes per element): 16-byte accesses, `n4` = n / 4 whole vectors (the four arrays 16-byte aligned: the flat parameter buffers are), the
// remainder and unaligned slices by the scalar loop below.  A block trip takes ONE contiguous 16 KiB piece of each array (4 x 256 vectors, all sixteen loads
// requested before the first use) instead of a grid-wide stride: 382 vs 448 us for the Performer's 79 M parameters in tools/probes/adam_bw.hip (5.8 vs 4.9 TB/s).
__global__ __launch_bounds__(256) void adam_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                        float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    constexpr int U = 4;
    for (int64_t base = (int64_t)blockIdx.x * (256 * U); base < n4; base += (int64_t)gridDim.x * (256 * U)) {
        float4 pe[U], ge[U], me[U], ve[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = base + u * 256 + threadIdx.x;
            if (e < n4) { pe[u] = p[e]; ge[u] = g[e]; me[u] = m[e]; ve[u] = v[e]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = base + u * 256 + threadIdx.x;
            if (e >= n4) continue;
            adam_one(pe[u].x, ge[u].x, me[u].x, ve[u].x, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
            adam_one(pe[u].y, ge[u].y, me[u].y, ve[u].y, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
            adam_one(pe[u].z, ge[u].z, me[u].z, ve[u].z, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
            adam_one(pe[u].w, ge[u].w, me[u].w, ve[u].w, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
            m[e] = me[u];
            v[e] = ve[u];
            p[e] = pe[u];
        }
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_one(pe, g[e], me, ve, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
        m[e] = me;
        v[e] = ve;
        p[e] = pe;
    }
}

static inline unsigned grid_for(int64_t n, int block = 256, unsigned cap = 4096) {
    int64_t b = (n + block - 1) / block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace sa

namespace sa {
// Peak probe: every wave issues `iters` x 8 independent v_mfma_f32_32x32x16_bf16 (8 accumulator tiles, no memory traffic) -- the number the
// roofline fractions can be read against on THIS device at ITS clocks (bench.py `roofline.peak_measured`).
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(16))) float f16x;
    typedef __attribute__((ext_vector_type(8))) short s8x;
    f16x acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    s8x a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3c00 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][15];
    if (t == 12345.678f) out[0] = t;   // keeps the loop alive without a store in the common case
#endif
}
// Cross-check of the probe above at other occupancies and with the instruction shape the convolution kernels use: four accumulator tiles per wave
// (64 VGPRs: up to 8 waves per SIMD), `iters` x 4 MFMAs.  SHAPE 0: v_mfma_f32_32x32x16_bf16 (32 768 FLOP), 1: v_mfma_f32_16x16x32_bf16 (16 384 FLOP).
template <int SHAPE>
__global__ __launch_bounds__(1024) void mfma_peak2_kernel(float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(16))) float f16x;
    typedef __attribute__((ext_vector_type(8))) short s8x;
    s8x a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + (threadIdx.x & 127)); b[e] = (short)(0x3c00 + e); }
    float t = 0.f;
    if constexpr (SHAPE == 0) {
        f16x acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][15];
    } else {
        float4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][3];
    }
    if (t == 12345.678f) out[0] = t;
#endif
}
}  // namespace sa

// FLOPs of one call = blocks * (threads / 64) waves * iters * 4 MFMAs * (32 768 | 16 384)
extern "C" int sa_bench_mfma_bf16_ex(float* scratch, int blocks, int threads, int iters, int shape, void* stream) {
    if (!scratch || blocks <= 0 || iters <= 0 || threads < 64 || threads > 1024 || (threads & 63) || (shape != 0 && shape != 1)) return SA_EINVAL;
    if (shape == 0) SA_LAUNCH(sa::mfma_peak2_kernel<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, scratch, iters);
    else SA_LAUNCH(sa::mfma_peak2_kernel<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, scratch, iters);
    SA_CHECK_LAUNCH();
    return 0;
}

// FLOPs of one call = blocks * 4 waves * iters * 8 MFMAs * (2 * 32 * 32 * 16)
extern "C" int sa_bench_mfma_bf16(float* scratch, int blocks, int iters, void* stream) {
    if (!scratch || blocks <= 0 || iters <= 0) return SA_EINVAL;
    SA_LAUNCH(sa::mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_abi_version(void) { return SA_ABI_VERSION; }
extern "C" const char* sa_last_error(void) { return hipGetErrorString(sa::g_last_error); }
extern "C" const char* sa_last_conv_kernel(void) { return sa::g_last_conv_kernel; }
extern "C" void sa_kernel_log_begin(void) {
    std::lock_guard<std::mutex> lk(sa::g_kernel_log_mu);
    if (sa::g_kernel_log) sa::g_kernel_log->clear();
    sa::g_kernel_log_on = true;
}
extern "C" int sa_kernel_log_read(char* buf, int cap, int stop) {
    std::lock_guard<std::mutex> lk(sa::g_kernel_log_mu);
    const std::string empty;
    const std::string& s = sa::g_kernel_log ? *sa::g_kernel_log : empty;
    const int need = (int)s.size() + 1;
    if (buf && cap > 0) {
        const int n = need <= cap ? need - 1 : cap - 1;
        std::memcpy(buf, s.data(), (size_t)n);
        buf[n] = 0;
    }
    if (stop) sa::g_kernel_log_on = false;
    return need;
}
extern "C" uint32_t sa_get_debug_flags(void) { return sa::g_debug_flags.load(); }
extern "C" uint32_t sa_set_debug_flags(uint32_t flags) { return sa::g_debug_flags.exchange(flags); }

extern "C" int sa_pack_weights(const float* w, void* wpk, int dtype, int rows, int red, int ntaps, const int32_t* tap_lut_host, int64_t s_row,
                               int64_t s_red, int rows_pad, int red_stride, int Kpad, void* stream) {
    using namespace sa;
    if (!w || !wpk || rows <= 0 || red <= 0 || ntaps <= 0 || ntaps > SA_MAX_TAPS || rows_pad < rows || red_stride < red ||
        Kpad < ntaps * red_stride)
        return SA_EINVAL;
    if (dtype != SA_F32 && dtype != SA_BF16 && dtype != SA_F16) return SA_EUNSUPPORTED;
    PackArgs a;
    a.w = w;
    a.wpk = wpk;
    for (int t = 0; t < SA_MAX_TAPS; ++t) a.lut[t] = tap_lut_host ? (t < ntaps ? tap_lut_host[t] : 0) : t;
    a.s_row = s_row;
    a.s_red = s_red;
    a.dtype = dtype;
    a.rows = rows;
    a.red = red;
    a.ntaps = ntaps;
    a.rows_pad = rows_pad;
    a.red_stride = red_stride;
    a.Kpad = Kpad;
    SA_LAUNCH(pack_weights_kernel, dim3(grid_for((int64_t)rows_pad * Kpad)), dim3(256), 0, (hipStream_t)stream, a);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_pack_weights_batch(const sa_pack_desc* table, const int32_t* block_first, int n, int total_blocks, void* stream) {
    using namespace sa;
    if (!table || !block_first || n <= 0 || total_blocks < n) return SA_EINVAL;
    SA_LAUNCH(pack_weights_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, block_first, n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_cast_pad(const void* src, int src_dtype, int src_c, void* dst, int dst_dtype, int dst_stride, int64_t rows, void* stream) {
    using namespace sa;
    if (!src || !dst || src_c <= 0 || dst_stride <= 0 || rows <= 0) return SA_EINVAL;
    const int64_t n = rows * dst_stride;
    if (src_c == dst_stride && src_dtype == SA_F32 && dst_dtype == SA_BF16 && (n & 7) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        // no padding: a flat fp32 -> bf16 conversion, 32 bytes in / 16 bytes out per thread and step
        SA_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n / 8, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (uint4*)dst, n / 8);
        SA_CHECK_LAUNCH();
        return 0;
    }
    SA_LAUNCH(cast_pad_kernel, dim3(grid_for(rows * dst_stride)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, src_c, dst, dst_dtype,
                       dst_stride, rows);
    SA_CHECK_LAUNCH();
    return 0;
}

// ---- sub-pixel up-sampling tail (use_subpixel_conv=True, reference baseline.py:274-282: MONAI SubpixelUpsample(3, C, 1, scale_factor=2, apply_pad_pool=True)) ----
// c [N, D, H, W, 8] fp32 = the conv_block output, channel f = (fd*2 + fh)*2 + fw.  pixelshuffle: S[2d+fd, 2h+fh, 2w+fw] = c[d, h, w, f]; ConstantPad3d((1, 0) x 3) +
// AvgPool3d(2, stride 1): out[z, y, x] = 1/8 sum_{dz,dy,dx in {0,1}} S[z-1+dz, y-1+dy, x-1+dx] (zero outside).  HBM-bound gathers; the eight channels of a cell are
// one 32-byte row, so the eight reads of an output voxel touch at most eight rows, all shared with its neighbours through L2.
namespace sa {
__global__ void subpixel_pool_fwd_kernel(const float* __restrict__ c, float* __restrict__ out, int N, int D, int H, int W) {
    const int64_t D2 = 2 * D, H2 = 2 * H, W2 = 2 * W, total = (int64_t)N * D2 * H2 * W2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W2), y = (int)((i / W2) % H2), z = (int)((i / (W2 * H2)) % D2);
        const int64_t n = i / (W2 * H2 * D2);
        float acc = 0.f;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int sz = z - 1 + dz, sy = y - 1 + dy, sx = x - 1 + dx;
                    if (sz < 0 || sy < 0 || sx < 0) continue;
                    acc += c[((((n * D + (sz >> 1)) * H + (sy >> 1)) * W + (sx >> 1)) << 3) + ((sz & 1) * 4 + (sy & 1) * 2 + (sx & 1))];
                }
        out[i] = acc * 0.125f;
    }
}
// adjoint: dc[d, h, w, f] = 1/8 sum over the outputs o in {s, s + 1} per axis (o < 2 D, 2 H, 2 W) of g[o], s = (2d+fd, 2h+fh, 2w+fw); written in the operand type of
// the conv_block weight / data gradient launches that read it
__global__ void subpixel_pool_bwd_kernel(const float* __restrict__ g, void* __restrict__ dc, int dc_dtype, int N, int D, int H, int W) {
    const int64_t D2 = 2 * D, H2 = 2 * H, W2 = 2 * W, total = (int64_t)N * D * H * W * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i & 7);
        const int64_t cell = i >> 3;
        const int w = (int)(cell % W), h = (int)((cell / W) % H), d = (int)((cell / ((int64_t)W * H)) % D);
        const int64_t n = cell / ((int64_t)W * H * D);
        const int sz = 2 * d + (f >> 2), sy = 2 * h + ((f >> 1) & 1), sx = 2 * w + (f & 1);
        float acc = 0.f;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oz = sz + dz, oy = sy + dy, ox = sx + dx;
                    if (oz >= D2 || oy >= H2 || ox >= W2) continue;
                    acc += g[((n * D2 + oz) * H2 + oy) * W2 + ox];
                }
        acc *= 0.125f;
        if (dc_dtype == SA_F32) ((float*)dc)[i] = acc;
        else ((bf16_t*)dc)[i] = f32_to_bf16(acc);
    }
}
}  // namespace sa

extern "C" int sa_subpixel_pool_fwd(const float* c, float* out, int N, int D, int H, int W, void* stream) {
    using namespace sa;
    if (!c || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0) return SA_EINVAL;
    SA_LAUNCH(subpixel_pool_fwd_kernel, dim3(grid_for((int64_t)N * D * H * W * 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream, c, out, N, D, H, W);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_subpixel_pool_bwd(const float* g, void* dc, int dc_dtype, int N, int D, int H, int W, void* stream) {
    using namespace sa;
    if (!g || !dc || N <= 0 || D <= 0 || H <= 0 || W <= 0) return SA_EINVAL;
    if (dc_dtype != SA_F32 && dc_dtype != SA_BF16) return SA_EUNSUPPORTED;
    SA_LAUNCH(subpixel_pool_bwd_kernel, dim3(grid_for((int64_t)N * D * H * W * 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream, g, dc, dc_dtype, N, D, H, W);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_mse(const float* a, const float* b, int64_t n, float* loss_sum, float* grad, float gscale, void* stream) {
    using namespace sa;
    if (!a || !b || !loss_sum || n <= 0) return SA_EINVAL;
    SA_LAUNCH(mse_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a, b, n, loss_sum, grad, 2.f * gscale / (float)n);
    SA_CHECK_LAUNCH();
    return 0;
}

extern "C" int sa_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int step, float grad_scale, void* stream) {
    using namespace sa;
    if (!p || !g || !m || !v || n <= 0 || step < 1) return SA_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const float bc2s = sqrtf(bc2);
    const hipStream_t st = (hipStream_t)stream;
    auto scalar = [&](int64_t o, int64_t cnt) {
        SA_LAUNCH(adam_kernel, dim3(grid_for(cnt)), dim3(256), 0, st, p + o, g + o, m + o, v + o, cnt, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    };
    // slices of the flat buffers (optimizer-in-backward ranges) share one misalignment: scalar head up to the 16-byte boundary, vectors, scalar tail
    const uintptr_t mis = (uintptr_t)p & 15u;
    const bool same = ((uintptr_t)g & 15u) == mis && ((uintptr_t)m & 15u) == mis && ((uintptr_t)v & 15u) == mis && (mis & 3u) == 0;
    int64_t head = same ? (int64_t)((16u - mis) & 15u) / 4 : n;
    if (head > n) head = n;
    const int64_t n4 = (n - head) / 4;
    if (head > 0) scalar(0, head);
    if (n4 > 0)
        SA_LAUNCH(adam_vec4_kernel, dim3(grid_for((n4 + 3) / 4, 256, 16384)), dim3(256), 0, st, (float4*)(p + head), (const float4*)(g + head), (float4*)(m + head),
                  (float4*)(v + head), n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    if (head + n4 * 4 < n) scalar(head + n4 * 4, n - head - n4 * 4);
    SA_CHECK_LAUNCH();
    return 0;
}
