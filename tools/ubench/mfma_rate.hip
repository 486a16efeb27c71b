// throughput of v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 independent accumulator chains per wave, 1 / 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int CHAINS> __global__ __launch_bounds__(256) void k(float* out, float a0, float b0, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 1e-3f + c;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run(float* d, int wgs) {
    const int iters = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<wgs, 256>>>(d, 1e-3f, 1e-3f, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<wgs, 256>>>(d, 1e-3f, 1e-3f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)wgs * 4 / 1024.0 * iters * 8 * CHAINS;
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    const double tf = (double)wgs * 4 * iters * 8 * CHAINS * 4096.0 / (ms * 1e-3) / 1e12;
    printf("chains=%d wgs=%d (%.1f waves/SIMD): %.3f ms  %.1f cycles/MFMA/SIMD at 2.4 GHz  %.1f TFLOP/s\n", CHAINS, wgs, wgs * 4 / 1024.0, ms, cyc, tf);
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    for (int wgs : {256, 512, 1024}) { run<1>(d, wgs); run<2>(d, wgs); run<4>(d, wgs); }
    return 0;
}
