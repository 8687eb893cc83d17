// Development micro-benchmark: issue rate of VALU instruction kinds on gfx950 (cycles per wave64
// instruction per SIMD) with CHAINS independent dependency chains per wave and W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int CHAINS>
__global__ void __launch_bounds__(1024) rate(float* out, int iters, float seed) {
    f32x2 a[CHAINS]; double d[CHAINS]; float f[CHAINS];
    for (int c = 0; c < CHAINS; c++) { a[c] = (f32x2){seed + c, seed - c}; d[c] = seed + c; f[c] = seed + c; }
    f32x2 m = (f32x2){1.0001f, 0.9999f}; double md = 1.0001; float mf = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = __builtin_elementwise_fma(a[c], m, m);       // v_pk_fma_f32
                if (KIND == 1) d[c] = __builtin_fma(d[c], md, md);                 // v_fma_f64
                if (KIND == 2) f[c] = __builtin_fmaf(f[c], mf, mf);                // v_fma_f32
                if (KIND == 3) a[c] = a[c] * m;                                    // v_pk_mul_f32
                if (KIND == 4) a[c] = a[c] + m;                                    // v_pk_add_f32
                if (KIND == 5) d[c] = d[c] + md;                                   // v_add_f64
                if (KIND == 6) f[c] = __builtin_fminf(f[c], mf + f[c]);            // add+min f32
            }
    }
    float s = 0; for (int c = 0; c < CHAINS; c++) s += a[c].x + a[c].y + (float)d[c] + f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int CHAINS> void go(const char* name, int threads, float* out) {
    const int iters = 2000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    rate<KIND, CHAINS><<<blocks, threads>>>(out, 10, 1.0f);
    hipEventRecord(a);
    rate<KIND, CHAINS><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_wave = double(iters) * 8 * CHAINS * (KIND == 6 ? 2 : 1);
    double waves_per_simd = threads / 64.0 / 4.0;
    printf("%-14s chains %d waves/SIMD %.0f: %.2f cycles per instruction per SIMD @2.4GHz\n", name, CHAINS, waves_per_simd,
           ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    go<0, 1>("v_pk_fma_f32", 256, out); go<0, 2>("v_pk_fma_f32", 256, out); go<0, 4>("v_pk_fma_f32", 256, out); go<0, 8>("v_pk_fma_f32", 256, out);
    go<0, 1>("v_pk_fma_f32", 512, out); go<0, 2>("v_pk_fma_f32", 512, out); go<0, 4>("v_pk_fma_f32", 512, out); go<0, 4>("v_pk_fma_f32", 1024, out);
    go<3, 4>("v_pk_mul_f32", 512, out); go<4, 4>("v_pk_add_f32", 512, out);
    go<1, 1>("v_fma_f64", 256, out); go<1, 2>("v_fma_f64", 256, out); go<1, 4>("v_fma_f64", 256, out); go<1, 4>("v_fma_f64", 512, out);
    go<5, 4>("v_add_f64", 512, out);
    go<2, 1>("v_fma_f32", 256, out); go<2, 4>("v_fma_f32", 256, out); go<2, 4>("v_fma_f32", 512, out);
    go<6, 4>("add+min f32", 512, out);
    return 0;
}
