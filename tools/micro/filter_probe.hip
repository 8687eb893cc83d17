// Development micro-benchmark for the packed-f32 filter loop of nn_filtered (icpx_pair.h):
// isolates the loop's VALU rate from flush frequency and branch placement.
//   VARIANT 0: branch per query slot (as nn_filtered)     1: no branch, OR-accumulate
//   VARIANT 2: one branch per pair of query slots          3: one branch per block (8 slots)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off filter_probe.hip -o filter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int QPT = 8;
static size_t g_pad = 96 * 1024;

template <int VARIANT>
__global__ void __launch_bounds__(512) probe(const float* __restrict__ tg, const float* __restrict__ qf_in,
                                             const float* __restrict__ thr_in, int m4, int reps, int* out) {
    extern __shared__ float lds[];
    float* tx = lds; float* ty = lds + m4; float* tz = lds + 2 * m4;
    for (int j = threadIdx.x; j < m4; j += blockDim.x) { tx[j] = tg[j * 3]; ty[j] = tg[j * 3 + 1]; tz[j] = tg[j * 3 + 2]; }
    __syncthreads();
    f32x2 q[QPT][3]; float T[QPT]; int pend[QPT]; int cnt = 0;
    for (int k = 0; k < QPT; k++) {
        for (int d = 0; d < 3; d++) { float v = qf_in[(threadIdx.x * QPT + k) * 3 + d]; q[k][d] = (f32x2){v, v}; }
        T[k] = thr_in[threadIdx.x * QPT + k]; pend[k] = -1;
    }
    for (int r = 0; r < reps; r++) {
        for (int jb = 0; jb < m4; jb += 4) {
            f32x4 x = *reinterpret_cast<const f32x4*>(tx + jb);
            f32x4 y = *reinterpret_cast<const f32x4*>(ty + jb);
            f32x4 z = *reinterpret_cast<const f32x4*>(tz + jb);
            bool c[QPT];
#pragma unroll
            for (int k = 0; k < QPT; k++) {
                f32x2 dx0 = x.xy - q[k][0], dx1 = x.zw - q[k][0];
                f32x2 dy0 = y.xy - q[k][1], dy1 = y.zw - q[k][1];
                f32x2 dz0 = z.xy - q[k][2], dz1 = z.zw - q[k][2];
                f32x2 s0 = dx0 * dx0, s1 = dx1 * dx1;
                s0 = __builtin_elementwise_fma(dy0, dy0, s0); s1 = __builtin_elementwise_fma(dy1, dy1, s1);
                s0 = __builtin_elementwise_fma(dz0, dz0, s0); s1 = __builtin_elementwise_fma(dz1, dz1, s1);
                float mn = __builtin_fminf(__builtin_fminf(s0.x, s0.y), __builtin_fminf(s1.x, s1.y));
                c[k] = !(mn > T[k]);
                if (VARIANT == 0) {
                    if (__any(c[k])) { if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; } }
                } else if (VARIANT == 1) {
                    cnt += c[k];
                } else if (VARIANT == 2) {
                    if (k & 1) {
                        if (__any(c[k] || c[k - 1])) {
                            if (c[k - 1]) { cnt += pend[k - 1] >= 0; pend[k - 1] = jb; }
                            if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; }
                        }
                    }
                }
            }
            if (VARIANT == 3) {
                bool any = false;
#pragma unroll
                for (int k = 0; k < QPT; k++) any = any || c[k];
                if (__any(any)) {
#pragma unroll
                    for (int k = 0; k < QPT; k++) if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; }
                }
            }
        }
    }
    int s = cnt; for (int k = 0; k < QPT; k++) s += pend[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// norm expansion in plain f32: v = |t|^2 - 2 t.q  (3 v_fma_f32 per target and query; |t|^2 once per target and wave)
template <int BR>
__global__ void __launch_bounds__(512) probe_ne(const float* __restrict__ tg, const float* __restrict__ qf_in,
                                                const float* __restrict__ thr_in, int m4, int reps, int* out) {
    extern __shared__ float lds[];
    float* tx = lds; float* ty = lds + m4; float* tz = lds + 2 * m4;
    for (int j = threadIdx.x; j < m4; j += blockDim.x) { tx[j] = tg[j * 3]; ty[j] = tg[j * 3 + 1]; tz[j] = tg[j * 3 + 2]; }
    __syncthreads();
    float q2[QPT][3]; float T[QPT]; int pend[QPT]; int cnt = 0;
    for (int k = 0; k < QPT; k++) {
        float qn = 0;
        for (int d = 0; d < 3; d++) { float v = qf_in[(threadIdx.x * QPT + k) * 3 + d]; q2[k][d] = -2.0f * v; qn += v * v; }
        T[k] = thr_in[threadIdx.x * QPT + k] - qn; pend[k] = -1;
    }
    for (int r = 0; r < reps; r++) {
        for (int jb = 0; jb < m4; jb += 4) {
            f32x4 x = *reinterpret_cast<const f32x4*>(tx + jb);
            f32x4 y = *reinterpret_cast<const f32x4*>(ty + jb);
            f32x4 z = *reinterpret_cast<const f32x4*>(tz + jb);
            float tn[4];
#pragma unroll
            for (int u = 0; u < 4; u++) tn[u] = __builtin_fmaf(z[u], z[u], __builtin_fmaf(y[u], y[u], x[u] * x[u]));
            bool c[QPT];
#pragma unroll
            for (int k = 0; k < QPT; k++) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    v[u] = __builtin_fmaf(z[u], q2[k][2], __builtin_fmaf(y[u], q2[k][1], __builtin_fmaf(x[u], q2[k][0], tn[u])));
                float mn = __builtin_fminf(__builtin_fminf(v[0], v[1]), __builtin_fminf(v[2], v[3]));
                c[k] = !(mn > T[k]);
                if (BR == 1) {
                    if (__any(c[k])) { if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; } }
                } else if (BR == 2) {
                    if (k & 1) {
                        if (__any(c[k] || c[k - 1])) {
                            if (c[k - 1]) { cnt += pend[k - 1] >= 0; pend[k - 1] = jb; }
                            if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; }
                        }
                    }
                } else if (BR == 0) cnt += c[k];
            }
            if (BR == 8) {
                bool any = false;
#pragma unroll
                for (int k = 0; k < QPT; k++) any = any || c[k];
                if (__any(any)) {
#pragma unroll
                    for (int k = 0; k < QPT; k++) if (c[k]) { cnt += pend[k] >= 0; pend[k] = jb; }
                }
            }
        }
    }
    int s = cnt; for (int k = 0; k < QPT; k++) s += pend[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int BR> float run_ne(const float* tg, const float* q, const float* thr, int m4, int reps, int* out, int blocks);

static size_t g_pad2 = 0;
template <int V> float run(const float* tg, const float* q, const float* thr, int m4, int reps, int* out, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t lds = sizeof(float) * 3 * m4 + g_pad;  // pad 96 KB: one workgroup per CU like the real kernel
    probe<V><<<blocks, 512, lds>>>(tg, q, thr, m4, 1, out);
    hipEventRecord(a);
    probe<V><<<blocks, 512, lds>>>(tg, q, thr, m4, reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms;
}

template <int BR> float run_ne(const float* tg, const float* q, const float* thr, int m4, int reps, int* out, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_ne<BR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t lds = sizeof(float) * 3 * m4 + g_pad;
    probe_ne<BR><<<blocks, 512, lds>>>(tg, q, thr, m4, 1, out);
    hipEventRecord(a);
    probe_ne<BR><<<blocks, 512, lds>>>(tg, q, thr, m4, reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms;
}
int main(int argc, char** argv) {
    const int m4 = 4096, reps = 8;
    const float thr_val = argc > 1 ? atof(argv[1]) : 0.0f;
    if (argc > 2) g_pad = atoi(argv[2]) * 1024;
    const int blocks = argc > 3 ? atoi(argv[3]) : 256;  // 0: nothing passes; larger: more candidates
    std::vector<float> tg(m4 * 3), q(512 * QPT * 3), thr(512 * QPT, thr_val);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (20.0f / 16777216.0f) - 10.0f; };
    for (auto& v : tg) v = rnd();
    for (auto& v : q) v = rnd();
    float *d_tg, *d_q, *d_thr; int* d_out;
    hipMalloc(&d_tg, tg.size() * 4); hipMalloc(&d_q, q.size() * 4); hipMalloc(&d_thr, thr.size() * 4); hipMalloc(&d_out, blocks * 512 * 4);
    hipMemcpy(d_tg, tg.data(), tg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice);
    float ms[4] = {run<0>(d_tg, d_q, d_thr, m4, reps, d_out, blocks), run<1>(d_tg, d_q, d_thr, m4, reps, d_out, blocks),
                   run<2>(d_tg, d_q, d_thr, m4, reps, d_out, blocks), run<3>(d_tg, d_q, d_thr, m4, reps, d_out, blocks)};
    // model: 15 VALU per (block, slot): 1024*8*15 instr/wave * 4 cycles * 2 waves/SIMD
    for (int v = 0; v < 4; v++)
        printf("pad %zu KB blocks %d variant %d thr %.3g: %.3f ms per scan (4096 queries x 4096 targets per workgroup), %.2f cycles per (block,slot) per wave @2.4GHz\n",
               g_pad / 1024, blocks, v, thr_val, ms[v] / reps, ms[v] / reps * 1e-3 * 2.4e9 / (1024.0 * 8 * 2));
    float ne[4] = {run_ne<0>(d_tg, d_q, d_thr, m4, reps, d_out, blocks), run_ne<1>(d_tg, d_q, d_thr, m4, reps, d_out, blocks),
                   run_ne<2>(d_tg, d_q, d_thr, m4, reps, d_out, blocks), run_ne<8>(d_tg, d_q, d_thr, m4, reps, d_out, blocks)};
    const char* nm[4] = {"no branch", "branch/slot", "branch/2 slots", "branch/block"};
    for (int v = 0; v < 4; v++)
        printf("norm-expansion plain f32, %s: %.3f ms per scan, %.2f cycles per (block,slot) per wave @2.4GHz\n", nm[v], ne[v] / reps,
               ne[v] / reps * 1e-3 * 2.4e9 / (1024.0 * 8 * 2));
    return 0;
}
