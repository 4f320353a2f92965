// Dev probe: Adam's 28 bytes per parameter (read p, g, m, v; write p, m, v) at the Performer's 79 M parameters, by launch shape.
// hipcc --offload-arch=gfx950 -O3 adam_bw.hip -o adam_bw && ./adam_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void one(float& p, float g, float& m, float& v) {
    m = 0.9f * m + 0.1f * g;
    v = 0.999f * v + 0.001f * g * g;
    p = p - 1e-3f * (m / (sqrtf(v) / 0.9f + 1e-8f));
}
__device__ __forceinline__ void four(float4& p, const float4 g, float4& m, float4& v) { one(p.x, g.x, m.x, v.x); one(p.y, g.y, m.y, v.y); one(p.z, g.z, m.z, v.z); one(p.w, g.w, m.w, v.w); }
// U float4 groups per thread and trip, grid-stride
template <int U>
__global__ __launch_bounds__(256) void k(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride * U) {
        float4 P[U], G[U], M[U], V[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (e + u * stride < n4) { P[u] = p[e + u * stride]; G[u] = g[e + u * stride]; M[u] = m[e + u * stride]; V[u] = v[e + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (e + u * stride < n4) { four(P[u], G[u], M[u], V[u]); m[e + u * stride] = M[u]; v[e + u * stride] = V[u]; p[e + u * stride] = P[u]; }
    }
}
// one contiguous chunk of 256 x U float4 per block trip (blocks walk the array in order)
template <int U>
__global__ __launch_bounds__(256) void kc(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, long n4) {
    for (long base = (long)blockIdx.x * 256 * U; base < n4; base += (long)gridDim.x * 256 * U) {
        float4 P[U], G[U], M[U], V[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long e = base + u * 256 + threadIdx.x; if (e < n4) { P[u] = p[e]; G[u] = g[e]; M[u] = m[e]; V[u] = v[e]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long e = base + u * 256 + threadIdx.x; if (e < n4) { four(P[u], G[u], M[u], V[u]); m[e] = M[u]; v[e] = V[u]; p[e] = P[u]; } }
    }
}
template <typename F> static float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e3f;
}
int main() {
    const long n = 79000000, n4 = n / 4;
    float4 *p, *g, *m, *v;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    const int grids[] = {2048, 4096, 8192, 16384, 32768, 77149};
    for (int gr : grids) {
        const float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(gr), dim3(256), 0, 0, p, g, m, v, n4); });
        const float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(gr), dim3(256), 0, 0, p, g, m, v, n4); });
        const float t4 = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(gr), dim3(256), 0, 0, p, g, m, v, n4); });
        const float c2 = timeit([&] { hipLaunchKernelGGL(kc<2>, dim3(gr), dim3(256), 0, 0, p, g, m, v, n4); });
        const float c4 = timeit([&] { hipLaunchKernelGGL(kc<4>, dim3(gr), dim3(256), 0, 0, p, g, m, v, n4); });
        printf("grid %6d: stride U=1 %6.1f us (%4.2f TB/s)  U=2 %6.1f  U=4 %6.1f | chunk U=2 %6.1f  U=4 %6.1f (%4.2f TB/s)\n", gr, t1, n * 28.0 / t1 / 1e6, t2, t4, c2, c4, n * 28.0 / c4 / 1e6);
    }
    return 0;
}
