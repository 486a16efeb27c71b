// Store-only ceilings for the shape of BASELINE config 5's output: a (65536, 4096) float32 buffer (1 GiB) written by 2048 waves,
// each owning ROWS consecutive rows.  How much of the 6.9 TB/s a plain fill reaches survives when a wave's stores are 256-byte
// row segments 16 KiB apart (what one 64-voice matrix group of fused_steady_mix_kernel writes), and what wider segments buy:
//   seg256  : a wave = one 64-column group, one store instruction = one 256-B row segment        (the kernel's pattern)
//   seg512  : a wave = two adjacent groups, one instruction (8 B per lane) = one 512-B segment
//   seg1024 : a wave = four adjacent groups, one instruction (16 B per lane) = one 1-KiB segment
//   wg1024  : like seg256, but the 4 waves of a workgroup own 4 adjacent groups and walk the same rows (what the kernel's
//             blockIdx -> (group, block span) map gives: 1 KiB per row per workgroup, if the waves stay in step)
//   fill    : grid-stride 16-B stores over the whole buffer
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int LD = 4096;
using f2 = __attribute__((ext_vector_type(2))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
template <typename T, int PACE> __global__ __launch_bounds__(256) void seg(float* __restrict__ out, int rows_per_wave, int waves_per_row) {
    constexpr int W = sizeof(T) / 4;                                  // floats per lane
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int col = (int)(item % waves_per_row) * 64 * W + lane * W;
    const int64_t r0 = (item / waves_per_row) * rows_per_wave;
    T v;
    for (int k = 0; k < W; ++k) ((float*)&v)[k] = (float)lane;
    float* dst = out + r0 * LD + col;
    for (int r = 0; r < rows_per_wave; ++r) {
        *(T*)(dst + (int64_t)r * LD) = v;
        if (PACE) { for (int k = 0; k < PACE; ++k) asm volatile("s_nop 7"); }    // (some work between the rows, like the kernel has)
    }
}
__global__ __launch_bounds__(256) void fill(f4* __restrict__ out, size_t n) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}
int main() {
    const size_t rows = 65536, bytes = rows * LD * 4;
    float* a; (void)hipMalloc(&a, bytes);
    auto rep = [&](const char* name, float ms) { printf("%-40s %7.1f us  %6.0f GB/s\n", name, ms * 1e3, (double)bytes / ms / 1e6); };
    rep("fill, 8192 workgroups", timeit([&] { fill<<<8192, 256>>>((f4*)a, bytes / 16); }));
    // 2048 waves in all the segment variants: waves_per_row x (rows / rows_per_wave) = 2048
    rep("seg256  (64 groups x 32 spans of 2048 rows)", timeit([&] { seg<float, 0><<<512, 256>>>(a, 2048, 64); }));
    rep("seg512  (32 x 64 spans of 1024 rows)", timeit([&] { seg<f2, 0><<<512, 256>>>(a, 1024, 32); }));
    rep("seg1024 (16 x 128 spans of 512 rows)", timeit([&] { seg<f4, 0><<<512, 256>>>(a, 512, 16); }));
    rep("seg256 paced", timeit([&] { seg<float, 4><<<512, 256>>>(a, 2048, 64); }));
    rep("seg512 paced", timeit([&] { seg<f2, 4><<<512, 256>>>(a, 1024, 32); }));
    rep("seg1024 paced", timeit([&] { seg<f4, 4><<<512, 256>>>(a, 512, 16); }));
    // more, shorter waves (4096 / 8192)
    rep("seg256, 8192 waves of 512 rows", timeit([&] { seg<float, 0><<<2048, 256>>>(a, 512, 64); }));
    rep("seg1024, 8192 waves of 128 rows", timeit([&] { seg<f4, 0><<<2048, 256>>>(a, 128, 16); }));
    return 0;
}
