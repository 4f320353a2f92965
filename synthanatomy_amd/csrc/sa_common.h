// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/synthanatomy_hip.h"

namespace sa {

typedef unsigned short bf16_t;  // raw storage
// IEEE half, raw storage.  A distinct type (not a typedef of unsigned short) so that kernel templates select the f16 MFMA and the f16 conversions
// by overload; only the FORWARD operand type of a chain (the reference's AMP dtype, src/engines/trainer.py:161-163): gradients stay bf16.
struct f16_t { unsigned short v; };
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

extern thread_local hipError_t g_last_error;
// Process-wide developer switches (SA_DBG_* bits of include/synthanatomy_hip.h).  Read ONCE from the environment when the library is
// loaded; afterwards only sa_set_debug_flags() changes them.  No launch reads the environment.
extern std::atomic<uint32_t> g_debug_flags;
inline bool dbg(uint32_t bit) { return (g_debug_flags.load(std::memory_order_relaxed) & bit) != 0; }
struct Tunables { int wgrad_rows, wgrad_min_blocks, wgrad_halo_splits; uint32_t pp_dbg; };
extern const Tunables g_tunables;   // numeric developer knobs (SA_WGRAD_ROWS, SA_WGRAD_MIN_BLOCKS, SA_WGRAD_HALO_SPLITS, SA_PP_DBG), read once at load
// hipFuncSetAttribute is per device: `mask` holds one bit per device ordinal that has been configured.  The bit is set AFTER `configure` has
// run, so a second host thread that races the first either sees the bit (attributes are in place) or configures the kernel itself (setting
// an attribute twice is harmless) -- it can never skip the configuration and launch before it happened.
template <typename F> inline void configure_once_per_device(std::atomic<uint64_t>& mask, F&& configure) {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t bit = 1ull << (d & 63);
    if (mask.load(std::memory_order_acquire) & bit) return;
    configure();
    mask.fetch_or(bit, std::memory_order_release);
}
// name of the convolution kernel instance the last sa_conv_fprop / sa_resblock_fprop / sa_conv_wgrad call launched (rocprofv3 spelling)
extern thread_local char g_last_conv_kernel[128];
// Optional process-wide log of the kernel names the launchers dispatched (sa_kernel_log_begin / sa_kernel_log_read): lets a test assert WHICH
// kernels served a network without a profiler (process-wide because autograd runs the backward launches on its own thread).  Off by default:
// one relaxed atomic load per launch.
void note_kernel_slow(const char* name);
extern std::atomic<bool> g_kernel_log_on;
inline void note_kernel(const char* name) {
    if (g_kernel_log_on.load(std::memory_order_relaxed)) note_kernel_slow(name);
}
#define SA_LAUNCH(kern, ...)                     \
    do {                                         \
        sa::note_kernel(#kern);                  \
        hipLaunchKernelGGL(kern, __VA_ARGS__);   \
    } while (0)
template <typename T> inline const char* tname() { return sizeof(T) == 4 ? "float" : "unsigned short"; }
template <> inline const char* tname<f16_t>() { return "f16_t"; }

#define SA_CHECK_LAUNCH()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) {                  \
            sa::g_last_error = e__;               \
            return (int)e__;                      \
        }                                         \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (same as torch's float->bfloat16): the gfx950 conversion instruction (v_cvt_pk_bf16_f32, one VALU op; the
// bit-twiddling form it replaces in round 4 cost ~6 per value and was a visible share of the convolution epilogues: 64.6 -> 66.2 volumes/s)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// round-to-nearest-even; FINITE values beyond the half range saturate to +-65504 instead of becoming infinities (the un-normalised residual stream);
// NaN and +-inf inputs come out as NaN, so a diverged run shows in z and the loss: v_med3_f32 alone returns min3 = -65504 when an input is NaN (a NaN
// activation would turn into zero behind the next ReLU while the bf16 copy of the same value still carried it); f * 0 is 0 for finite f and NaN otherwise,
// which costs one fused multiply-add per value instead of a compare + select.
__device__ __forceinline__ unsigned short f32_to_f16(float f) {
    const _Float16 h = (_Float16)__builtin_fmaf(f, 0.f, __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_to_f32(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
// two values -> one packed 32-bit word of 16-bit storage type T (bf16_t / f16_t)
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
// (gfx950 v_cvt_pk_bf16_f32: one VALU op, round to nearest even like f32_to_bf16(), which costs ~6 per value)
typedef __attribute__((ext_vector_type(2))) float sa_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 sa_bf16x2_t;
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b) {
    const sa_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sa_bf16x2_t));
}
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b) { return (uint32_t)f32_to_f16(a) | ((uint32_t)f32_to_f16(b) << 16); }
// the same by run-time dtype (SA_BF16 / SA_F16)
__device__ __forceinline__ uint32_t pack2_dt(int dtype, float a, float b) { return dtype == SA_F16 ? pack2<f16_t>(a, b) : pack2<bf16_t>(a, b); }
// packed word -> two floats
__device__ __forceinline__ void unpack2_dt(int dtype, uint32_t w, float& a, float& b) {
    if (dtype == SA_F16) { a = f16_to_f32((unsigned short)(w & 0xffffu)); b = f16_to_f32((unsigned short)(w >> 16)); }
    else { a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u); }
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int id = SA_F32;
    static constexpr int VEC = 4;  // elements per 16 bytes
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int id = SA_BF16;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct DT<f16_t> {
    static constexpr int id = SA_F16;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return f16_to_f32(p->v); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = f32_to_f16(v); }
};

__device__ __forceinline__ float load_as_f32(const void* base, int dtype, int64_t off) {
    return dtype == SA_F32 ? ((const float*)base)[off] : dtype == SA_F16 ? f16_to_f32(((const unsigned short*)base)[off]) : bf16_to_f32(((const bf16_t*)base)[off]);
}
__device__ __forceinline__ void store_from_f32(void* base, int dtype, int64_t off, float v) {
    if (dtype == SA_F32) ((float*)base)[off] = v;
    else if (dtype == SA_F16) ((unsigned short*)base)[off] = f32_to_f16(v);
    else ((bf16_t*)base)[off] = f32_to_bf16(v);
}

// Division by a launch-invariant 32-bit divisor via multiply-high (valid for x < 2^31).
struct FastDiv {
    uint32_t mul, shr, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    uint32_t l = 0;
    while ((1u << l) < d) ++l;  // ceil(log2 d)
    uint64_t p = 31 + l;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = (uint32_t)(p - 32);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t x, const FastDiv& f) { return f.d <= 1 ? x : (__umulhi(x, f.mul) >> f.shr); }

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a linear block id: blocks that are adjacent in the remapped order (and so share
// input halos / weight panels) land on the same XCD's L2.  Dispatch places block b on XCD b % 8 (speed only).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// (value, index) maximum as one 64-bit integer for atomicMax: order-preserving float bits above, inverted index below (ties: lowest index wins)
__device__ __forceinline__ unsigned long long pack_max(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - idx);
}
__device__ __forceinline__ float unpack_max(unsigned long long p) {
    uint32_t u = (uint32_t)(p >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// 4 x 4 transpose across the four 16-lane quarters of a wave (gfx950 v_permlane32_swap + v_permlane16_swap; semantics probed on MI355X):
//   permlane32_swap(a, b) = {[a.q0 a.q1 b.q0 b.q1], [a.q2 a.q3 b.q2 b.q3]};  permlane16_swap(a, b) = {[a.q0 b.q0 a.q2 b.q2], [a.q1 b.q1 a.q3 b.q3]}
// in: lane quarter q holds r_i = element (q, i);  out: lane quarter q holds r_i = element (i, q).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void quarter_transpose(float& r0, float& r1, float& r2, float& r3) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x2_t a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
    const u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
    const u32x2_t c = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const u32x2_t d = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    r0 = __uint_as_float(c.x); r1 = __uint_as_float(c.y); r2 = __uint_as_float(d.x); r3 = __uint_as_float(d.y);
#endif
}

}  // namespace sa
