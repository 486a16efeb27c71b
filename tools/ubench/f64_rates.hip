#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
// throughput microbenchmarks: 8 independent chains per lane, ITER iterations, chip filled with 8 waves/SIMD
#define ITER 4096
template <int OP> __device__ __forceinline__ double step(double x, double c) {
    if (OP == 0) return fma(x, c, 0.5);                       // v_fma_f64
    if (OP == 1) return x + c;                                // v_add_f64
    if (OP == 2) return x * c;                                // v_mul_f64
    if (OP == 3) return rint(x) + c;                          // v_rndne_f64 + add
    if (OP == 4) return (double)((float)x) ;                  // cvt f64->f32->f64
    if (OP == 5) return (double)__builtin_amdgcn_sinf((float)x);   // cvt, v_sin_f32, cvt
    if (OP == 6) return trunc(x) + c;                         // v_trunc_f64 + add
    if (OP == 7) return (x > c) ? x : c + 1.0;                // cmp + cndmask x2 (+add)
    if (OP == 8) return x / c;                                // IEEE divide
    return x;
}
template <int OP> __global__ void k(double* out, double c) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = step<OP>(a[i], c);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* d, double c) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, c); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_steps = (double)blocks * threads / 64 * ITER * 8;     // "step" invocations per wave
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_steps;               // cycles per step per SIMD at 2.4 GHz
    printf("%-28s %8.3f ms  %6.2f cycles/step/wave (at 2.4 GHz)\n", name, ms, cyc);
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("fma_f64", d, 1.0000001);
    run<1>("add_f64", d, 1e-9);
    run<2>("mul_f64", d, 1.0000001);
    run<3>("rndne_f64 + add", d, 0.3);
    run<4>("cvt f64->f32->f64", d, 0.0);
    run<5>("cvt, v_sin_f32, cvt", d, 0.0);
    run<6>("trunc_f64 + add", d, 0.3);
    run<7>("cmp+cndmask(+add)", d, 0.5);
    run<8>("div_f64 (IEEE)", d, 1.0000001);
    return 0;
}
