// do MFMA (matrix core) and f64 VALU work from two different waves of one SIMD overlap?
// grid of 512 workgroups x 256 threads = 2 waves per SIMD; role by blockIdx>>8 (workgroups j and j+256 share a CU)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ float mfma_work(int iters, float a, float b) {
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0); }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    return s;
}
__device__ double valu_work(int iters, double c) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], c, 0.5);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    return s;
}
// mode 0: every wave MFMA; 1: every wave VALU; 2: half the workgroups MFMA, half VALU; 3: every wave does both, interleaved phases
__global__ __launch_bounds__(256) void k(float* out, int mode, int mi, int vi) {
    const bool second = (blockIdx.x >> 8) & 1;
    float r = 0;
    if (mode == 0 || (mode == 2 && !second)) r = mfma_work(mi, 1e-3f + threadIdx.x * 1e-6f, 1e-3f);
    else if (mode == 1 || (mode == 2 && second)) r = (float)valu_work(vi, 1.0000001);
    else { for (int p = 0; p < 16; ++p) { r += mfma_work(mi / 16, 1e-3f + p, 1e-3f); r += (float)valu_work(vi / 16, 1.0000001 + p * 1e-9); } }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
float run(float* d, int mode, int mi, int vi) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<512, 256>>>(d, mode, mi, vi); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<512, 256>>>(d, mode, mi, vi);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int mi = 2048, vi = 8192;       // 32768 MFMAs ~ 2.1 M cycles; 262144 f64 FMAs ~ 1.3 M cycles per wave
    printf("all waves MFMA            : %.3f ms\n", run(d, 0, mi, vi));
    printf("all waves f64 VALU        : %.3f ms\n", run(d, 1, mi, vi));
    printf("half MFMA, half VALU      : %.3f ms  (one wave of each kind per SIMD: max = overlap, sum/2 = none)\n", run(d, 2, mi, vi));
    printf("every wave alternates both: %.3f ms  (sum = none)\n", run(d, 3, mi, vi));
    return 0;
}
