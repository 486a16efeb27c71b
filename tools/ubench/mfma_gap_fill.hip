// How many vector instructions of which kind hide behind one MFMA of the same wave?  One wave per SIMD (1024 waves),
// each iteration = 1 MFMA + N independent VALU instructions; reported: cycles per iteration per SIMD (at 2.4 GHz
// accounting) for N = 0 .. 16.  MFMA kinds: f32 32x32x2 (exact f32, 64 cycles) and bf16 32x32x16 (32 cycles);
// VALU kinds: v_fma_f64, v_fma_f32, v_cvt_pk_bf16_f32 + v_sub_f32 pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
#define ITER 4096
template <int MK, int VK, int N>
__global__ __launch_bounds__(256) void k(float* out, float fa, double dc)
{
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(fa + j); bb[j] = (__bf16)(fa - j); }
    double xd[16]; float xf[16];
    for (int i = 0; i < 16; ++i) { xd[i] = threadIdx.x * 1e-3 + i; xf[i] = threadIdx.x * 1e-3f + i; }
    for (int it = 0; it < ITER; ++it) {
        if (MK == 0) { acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fa, acc0, 0, 0, 0); }
        if (MK == 1) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0); }
        if (MK == 0 && (it & 1)) acc1 = acc0;      // (never true at compile time: keeps two chains alive without extra MFMAs)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (VK == 0) xd[i] = fma(xd[i], dc, 0.5);
            if (VK == 1) xf[i] = fmaf(xf[i], fa, 0.5f);
            if (VK == 2) { const __bf16 h = (__bf16)xf[i]; xf[i] = xf[i] - (float)h + 1.0f; }   // cvt + (shift) + sub + add
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 16; ++i) s += (float)xd[i] + xf[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MK, int VK, int N> void run(float* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MK, VK, N><<<256, 256>>>(d, 1e-3f, 1.0000001); (void)hipDeviceSynchronize();
    for (int w = 0; w < 3; ++w) k<MK, VK, N><<<256, 256>>>(d, 1e-3f, 1.0000001);
    (void)hipEventRecord(e0);
    k<MK, VK, N><<<256, 256>>>(d, 1e-3f, 1.0000001);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf(" %6.1f", ms * 1e-3 * 2.4e9 / ITER);
}
template <int MK, int VK> void row(float* d, const char* name)
{
    printf("%-34s", name);
    run<MK, VK, 0>(d); run<MK, VK, 2>(d); run<MK, VK, 4>(d); run<MK, VK, 6>(d); run<MK, VK, 8>(d); run<MK, VK, 12>(d); run<MK, VK, 16>(d);
    printf("\n");
}
int main()
{
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    printf("cycles per iteration (2.4 GHz), N =      0      2      4      6      8     12     16\n");
    row<2, 0>(d, "no MFMA + N v_fma_f64");
    row<2, 1>(d, "no MFMA + N v_fma_f32");
    row<2, 2>(d, "no MFMA + N (cvt bf16, sub, add)");
    row<0, 0>(d, "f32 32x32x2 + N v_fma_f64");
    row<0, 1>(d, "f32 32x32x2 + N v_fma_f32");
    row<1, 0>(d, "bf16 32x32x16 + N v_fma_f64");
    row<1, 1>(d, "bf16 32x32x16 + N v_fma_f32");
    row<1, 2>(d, "bf16 32x32x16 + N (cvt, sub, add)");
    return 0;
}
