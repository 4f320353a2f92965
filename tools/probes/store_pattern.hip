// Dev probe: how fast can 128 x 128 fp32 tiles of an [M][N] matrix be WRITTEN, by store pattern?  (hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float4_t __attribute__((ext_vector_type(4)));
// P: 0 = tile, thread (row = tid/32 + 16*it, 16-byte group tid%32): a wave writes 2 rows x 512 B per store (the staged epilogue's phase B)
//    1 = the same through an LDS tile first (park float4 per lane as the MFMA layout would, barrier, re-read rows) -- the whole staged epilogue
//    2 = MFMA-layout direct: lane (frow, fq) writes 16 B of row frow: a wave store = 16 rows x 64 B
//    3 = contiguous fill (reference)
template <int P>
__global__ __launch_bounds__(512) void k(float* out, int M, int N, int nblk_m) {
    __shared__ float sT[128 * 132];
    const int tid = threadIdx.x, bm = blockIdx.x % nblk_m, bn = blockIdx.x / nblk_m;
    const int m0 = bm * 128, n0 = bn * 128;
    if (P == 3) {
        float4_t* o = (float4_t*)out + (size_t)blockIdx.x * 4096;
        for (int it = 0; it < 8; ++it) o[it * 512 + tid] = (float4_t){1.f, 2.f, 3.f, (float)tid};
        return;
    }
    if (P == 1) {
        const int lane = tid & 63, wave = tid >> 6, frow = lane & 15, fq = lane >> 4, wm = wave / 2, wn = wave % 2;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 4; ++i) *(float4_t*)(sT + (wm * 32 + j * 16 + frow) * 132 + wn * 64 + i * 16 + fq * 4) = (float4_t){1.f, 2.f, (float)i, (float)tid};
        __syncthreads();
    }
    if (P == 0 || P == 1) {
        const int grp = tid % 32, r0 = tid / 32;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int row = r0 + it * 16, m = m0 + row;
            if (m >= M) continue;
            float4_t v = P == 1 ? *(const float4_t*)(sT + row * 132 + grp * 4) : (float4_t){1.f, 2.f, 3.f, (float)tid};
            *(float4_t*)(out + (size_t)m * N + n0 + grp * 4) = v;
        }
    } else {
        const int lane = tid & 63, wave = tid >> 6, frow = lane & 15, fq = lane >> 4, wm = wave / 2, wn = wave % 2;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 32 + j * 16 + frow;
                if (m < M) *(float4_t*)(out + (size_t)m * N + n0 + wn * 64 + i * 16 + fq * 4) = (float4_t){1.f, 2.f, (float)i, (float)tid};
            }
    }
}
template <int P> float run(float* out, int M, int N) {
    const int nbm = (M + 127) / 128, nbn = N / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<P>, dim3(nbm * nbn), dim3(512), 0, 0, out, M, N, nbm);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<P>, dim3(nbm * nbn), dim3(512), 0, 0, out, M, N, nbm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}
int main() {
    const int M = 8400;
    for (int N : {512, 1024, 2048, 3072}) {
        float* out; hipMalloc(&out, (size_t)(M + 128) * N * 4);
        const double mb = (double)M * N * 4 / 1e6;
        const float t0 = run<0>(out, M, N), t1 = run<1>(out, M, N), t2 = run<2>(out, M, N), t3 = run<3>(out, M, N);
        printf("N=%4d (%6.1f MB): rows-direct %6.1f us (%5.2f TB/s) | LDS-staged rows %6.1f us (%5.2f TB/s) | MFMA-layout direct %6.1f us (%5.2f TB/s) | contiguous %6.1f us (%5.2f TB/s)\n", N, mb,
               t0, mb / t0 / 1e6 * 1e6 / 1e6, t1, mb / t1, t2, mb / t2, t3, mb / t3);
        hipFree(out);
    }
    return 0;
}
