#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r.x; p[64 + threadIdx.x] = r.y; p[128 + threadIdx.x] = q.x; p[192 + threadIdx.x] = q.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[4] = {"p32.x", "p32.y", "p16.x", "p16.y"};
    for (int v = 0; v < 4; ++v) { printf("%s:", nm[v]); for (int q = 0; q < 4; ++q) printf(" [%u..%u]", h[v * 64 + q * 16], h[v * 64 + q * 16 + 15]); printf("\n"); }
    return 0;
}
