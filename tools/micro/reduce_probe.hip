// Development micro-benchmark: 64-lane f64 sum by ds_bpermute butterfly (__shfl_xor) against a DPP
// reduction, and a bit-level check of the DPP version against the tree it is meant to implement
// (pairs at distance 1, then 2, 4, 8, 16, 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ double sum_shfl(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // lanes not written by the DPP move (row_mask / invalid source) keep `old` = 0.0 bits
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// result valid in every lane
__device__ __forceinline__ double sum_dpp(double v) {
    v = v + dpp_move<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]: distance 1
    v = v + dpp_move<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]: distance 2
    v = v + dpp_move<0x141, 0xf>(v);  // row_half_mirror: quads are uniform -> distance 4
    v = v + dpp_move<0x140, 0xf>(v);  // row_mirror: octets are uniform -> distance 8
    v = v + dpp_move<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3: distance 16 (valid in rows 1, 3)
    v = v + dpp_move<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: distance 32 (valid in row 3)
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

template <int KIND>
__global__ void __launch_bounds__(1024) bench(const double* in, double* out, int reps) {
    double v = in[blockIdx.x * blockDim.x + threadIdx.x];
    double acc = 0.0;
    for (int r = 0; r < reps; r++) {
        double s = KIND == 0 ? sum_shfl(v + r) : sum_dpp(v + r);
        acc += s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void check(const double* in, double* out) {
    out[threadIdx.x] = sum_dpp(in[threadIdx.x]);
}

int main() {
    const int blocks = 256, threads = 1024, reps = 2000;
    std::vector<double> h(blocks * threads);
    unsigned s = 99;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (1.0 / 16777216.0) * 1e3 - 300.0 + 1e-7 * (s & 255); }
    double *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, h.size() * 8);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    // correctness: first 64 values
    check<<<1, 64>>>(d_in, d_out);
    double got[64]; hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost);
    double lane[64]; memcpy(lane, h.data(), sizeof(lane));
    for (int d = 1; d < 64; d <<= 1) { double nx[64]; for (int l = 0; l < 64; l++) nx[l] = lane[l] + lane[l ^ d]; memcpy(lane, nx, sizeof(nx)); }
    int bad = 0; for (int l = 0; l < 64; l++) bad += memcmp(&got[l], &lane[0], 8) != 0;
    printf("dpp sum vs distance-1,2,4,...,32 tree: %d lanes differ (got %.17g want %.17g)\n", bad, got[0], lane[0]);
    for (int kind = 0; kind < 2; kind++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (kind == 0) bench<0><<<blocks, threads>>>(d_in, d_out, 10); else bench<1><<<blocks, threads>>>(d_in, d_out, 10);
        hipEventRecord(a);
        if (kind == 0) bench<0><<<blocks, threads>>>(d_in, d_out, reps); else bench<1><<<blocks, threads>>>(d_in, d_out, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.3f ms for %d sums per wave, 16 waves per CU -> %.1f ns per wave-sum per CU-wave (%.0f cycles @2.4GHz per sum per SIMD)\n",
               kind == 0 ? "shfl_xor (ds_bpermute)" : "dpp", ms, reps, ms * 1e6 / reps / 16, ms * 1e-3 * 2.4e9 / reps / 4);
    }
    return 0;
}
