// development probe: lane mapping of v_permlane32_swap / v_permlane16_swap on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    auto r2 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[threadIdx.x * 4 + 0] = r[0];
    out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = r2[0];
    out[threadIdx.x * 4 + 3] = r2[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64 * 16);
    k<<<1, 64>>>(d);
    unsigned h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 5, 16, 21, 32, 37, 48, 53}) printf("lane %2d: swap32 -> (%2u, %2u)  swap16 -> (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
