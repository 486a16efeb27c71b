// HBM copy ceilings on this box for the access patterns the per-node kernels use (1 GiB in, 1 GiB out, f32 (rows, 1024)):
//   flat      : grid-stride f4 copy, fully coalesced over the whole buffer
//   rowserial : what biquad.hip does -- a wave owns a 1 KiB column slice (4 voices per lane) of ROWS_PER_WAVE consecutive
//               rows and walks them serially with RING row loads in flight; a workgroup's 4 waves cover one 4 KiB row
//   *_nt      : the same with non-temporal loads and stores
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int V = 1024;
using f4 = __attribute__((ext_vector_type(4))) float;   // the non-temporal builtins want a native vector type
template <bool NT> __global__ __launch_bounds__(256) void flat(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <int RING, bool NT> __global__ __launch_bounds__(256) void rowserial(const float* __restrict__ in, float* __restrict__ out, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r0 = (size_t)blockIdx.x * rows_per_wave;
    const float* src = in + r0 * V + (wave * 64 + lane) * 4;
    float* dst = out + r0 * V + (wave * 64 + lane) * 4;
    f4 ring[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) ring[u] = NT ? __builtin_nontemporal_load((const f4*)(src + (size_t)u * V)) : *(const f4*)(src + (size_t)u * V);
    for (int r = 0; r < rows_per_wave; r += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            f4 v = ring[u];
            const int rn = (r + u + RING < rows_per_wave) ? r + u + RING : rows_per_wave - 1;
            ring[u] = NT ? __builtin_nontemporal_load((const f4*)(src + (size_t)rn * V)) : *(const f4*)(src + (size_t)rn * V);
            v.x *= 1.0001f;
            if (NT) __builtin_nontemporal_store(v, (f4*)(dst + (size_t)(r + u) * V)); else *(f4*)(dst + (size_t)(r + u) * V) = v;
        }
    }
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
    const size_t rows = 262144, bytes = rows * V * 4;
    float *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 0, bytes);
    auto rep = [&](const char* name, float ms) { printf("%-28s %7.1f us  %6.0f GB/s (read + write)\n", name, ms * 1e3, 2.0 * bytes / ms / 1e6); };
    for (int wgs : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "flat wgs=%d", wgs); rep(nm, timeit([&] { flat<false><<<wgs, 256>>>((const f4*)a, (f4*)b, bytes / 16); }));
        snprintf(nm, 64, "flat_nt wgs=%d", wgs); rep(nm, timeit([&] { flat<true><<<wgs, 256>>>((const f4*)a, (f4*)b, bytes / 16); }));
    }
    for (int rpw : {64, 256, 1024}) {
        char nm[64];
        snprintf(nm, 64, "rowserial ring16 rows/wave=%d", rpw); rep(nm, timeit([&] { rowserial<16, false><<<rows / rpw, 256>>>(a, b, rpw); }));
        snprintf(nm, 64, "rowserial ring32 rows/wave=%d", rpw); rep(nm, timeit([&] { rowserial<32, false><<<rows / rpw, 256>>>(a, b, rpw); }));
        snprintf(nm, 64, "rowserial_nt ring16 rows/wave=%d", rpw); rep(nm, timeit([&] { rowserial<16, true><<<rows / rpw, 256>>>(a, b, rpw); }));
    }
    return 0;
}
