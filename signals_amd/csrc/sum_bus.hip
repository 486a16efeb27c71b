// Sum bus for gfx950 (build-defined node; the reference's Flatten crashes, shape.py:32-35):
//   out[n,c] = sum_v gains[c,v] * x[n,v]   (or the plain sum over voices when gains == NULL).
// HBM-bound row reduction: one wave owns kRows consecutive rows; lane l reads voices
// [4l + 256j, 4l + 256j + 4) for j = 0,1,... as 16-B loads (1 KiB per wave-instruction), accumulates in
// f64, then a 6-step xor butterfly across the wave.  The summation order is fixed, so the result is
// bitwise reproducible run to run.
// Algorithmic traffic: 4 B read per voice-sample; the (C,V) f64 gain table is re-read from L2.
#include "sig_common.h"

namespace {

constexpr int kRows = 8;

template <typename T, int C, bool GAINS>
__global__ __launch_bounds__(256) void sum_bus_kernel(int64_t rows, int voices, const T* __restrict__ x, int64_t ld,
                                                      const double* __restrict__ gains, int64_t gld,
                                                      void* __restrict__ out, int64_t out_ld, int out_f64, bool vec4)
{
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kRows;
    if (r0 >= rows) return;
    double acc[kRows][C];
#pragma unroll
    for (int j = 0; j < kRows; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[j][c] = 0.0;

    if (vec4) {
        using V4 = typename sig_vec4<T>::type;
        for (int v = lane * 4; v < voices; v += 256) {
            double g[C][4];
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) g[c][i] = GAINS ? gains[c * gld + v + i] : 1.0;
            V4 xv[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int64_t r = (r0 + j < rows) ? r0 + j : rows - 1;
                xv[j] = *reinterpret_cast<const V4*>(x + r * ld + v);
            }
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const double e[4] = {(double)xv[j].x, (double)xv[j].y, (double)xv[j].z, (double)xv[j].w};
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][c] = GAINS ? fma(g[c][i], e[i], acc[j][c]) : acc[j][c] + e[i];
            }
        }
    } else {
        for (int v = lane; v < voices; v += 64) {
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int64_t r = (r0 + j < rows) ? r0 + j : rows - 1;
                const double e = (double)x[r * ld + v];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[j][c] = GAINS ? fma(gains[c * gld + v], e, acc[j][c]) : acc[j][c] + e;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kRows; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double s = acc[j][c];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) s += sig_shfl_xor_f64(s, m);
            acc[j][c] = s;
        }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            if (r0 + j >= rows) break;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (out_f64) ((double*)out)[(r0 + j) * out_ld + c] = acc[j][c];
                else ((float*)out)[(r0 + j) * out_ld + c] = (float)acc[j][c];
            }
        }
    }
}

// Fast path (f32 in, voices == 256*CHUNKS <= 1024, 16-B aligned rows): the lane's gains for all of its
// voices stay in VGPRs (C*4*CHUNKS doubles), each wave strides over row groups of kFastRows rows, and all
// CHUNKS*kFastRows 16-B loads of a group are issued before the first FMA.  The summation order per
// (row, channel) is the same as the generic kernel's: chunk-major within the lane, then the butterfly.
constexpr int kFastRows = 4;

template <int C, bool GAINS, int CHUNKS>
__global__ __launch_bounds__(256) void sum_bus_fast_kernel(int64_t rows, const float* __restrict__ x, int64_t ld,
                                                           const double* __restrict__ gains, int64_t gld,
                                                           void* __restrict__ out, int64_t out_ld, int out_f64)
{
    const int lane = threadIdx.x & 63;
    double g[CHUNKS][C][4];
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) g[j][c][i] = GAINS ? gains[c * gld + 256 * j + 4 * lane + i] : 1.0;

    const int64_t groups = (rows + kFastRows - 1) / kFastRows;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t r0 = grp * kFastRows;
        float4 xv[kFastRows][CHUNKS];
#pragma unroll
        for (int r = 0; r < kFastRows; ++r) {
            const int64_t row = (r0 + r < rows) ? r0 + r : rows - 1;
#pragma unroll
            for (int j = 0; j < CHUNKS; ++j)
                xv[r][j] = *reinterpret_cast<const float4*>(x + row * ld + 256 * j + 4 * lane);
        }
        double acc[kFastRows][C];
#pragma unroll
        for (int r = 0; r < kFastRows; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) acc[r][c] = 0.0;
#pragma unroll
            for (int j = 0; j < CHUNKS; ++j) {
                const double e[4] = {(double)xv[r][j].x, (double)xv[r][j].y, (double)xv[r][j].z, (double)xv[r][j].w};
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[r][c] = GAINS ? fma(g[j][c][i], e[i], acc[r][c]) : acc[r][c] + e[i];
            }
        }
#pragma unroll
        for (int r = 0; r < kFastRows; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double s = acc[r][c];
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) s += sig_shfl_xor_f64(s, m);
                acc[r][c] = s;
            }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < kFastRows; ++r) {
                if (r0 + r >= rows) break;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (out_f64) ((double*)out)[(r0 + r) * out_ld + c] = acc[r][c];
                    else ((float*)out)[(r0 + r) * out_ld + c] = (float)acc[r][c];
                }
            }
        }
    }
}

template <int C, bool GAINS>
bool launch_bus_fast(int64_t rows, int voices, const float* x, int64_t ld, const double* gains, int64_t gld,
                     void* out, int64_t out_ld, int out_f64, hipStream_t stream)
{
    if (voices % 256 || voices > 1024 || ld % 4 || reinterpret_cast<uintptr_t>(x) % 16) return false;
    const int64_t groups = (rows + kFastRows - 1) / kFastRows;
    int64_t nwg = (groups + 3) / 4;
    if (nwg > 256 * 8) nwg = 256 * 8;
    switch (voices / 256) {
        case 1: sum_bus_fast_kernel<C, GAINS, 1><<<(unsigned)nwg, 256, 0, stream>>>(rows, x, ld, gains, gld, out, out_ld, out_f64); break;
        case 2: sum_bus_fast_kernel<C, GAINS, 2><<<(unsigned)nwg, 256, 0, stream>>>(rows, x, ld, gains, gld, out, out_ld, out_f64); break;
        case 3: sum_bus_fast_kernel<C, GAINS, 3><<<(unsigned)nwg, 256, 0, stream>>>(rows, x, ld, gains, gld, out, out_ld, out_f64); break;
        case 4: sum_bus_fast_kernel<C, GAINS, 4><<<(unsigned)nwg, 256, 0, stream>>>(rows, x, ld, gains, gld, out, out_ld, out_f64); break;
    }
    return true;
}

template <typename T, int C>
int launch_bus(int64_t rows, int voices, const T* x, int64_t ld, const double* gains, int64_t gld,
               void* out, int64_t out_ld, int out_f64, hipStream_t stream)
{
    if (sizeof(T) == 4 && C <= 2) {
        const float* xf = reinterpret_cast<const float*>(x);
        const bool done = gains ? launch_bus_fast<C, true>(rows, voices, xf, ld, gains, gld, out, out_ld, out_f64, stream)
                                : launch_bus_fast<C, false>(rows, voices, xf, ld, gains, gld, out, out_ld, out_f64, stream);
        if (done) return sig_launch_status();
    }
    const bool vec4 = (voices % 4 == 0) && (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % (4 * sizeof(T)) == 0);
    const int64_t nwg = (rows + 4 * kRows - 1) / (4 * kRows);
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (gains)
        sum_bus_kernel<T, C, true><<<(unsigned)nwg, 256, 0, stream>>>(rows, voices, x, ld, gains, gld, out, out_ld, out_f64, vec4);
    else
        sum_bus_kernel<T, C, false><<<(unsigned)nwg, 256, 0, stream>>>(rows, voices, x, ld, gains, gld, out, out_ld, out_f64, vec4);
    return sig_launch_status();
}

template <typename T>
int dispatch_channels(int C, int64_t rows, int voices, const T* x, int64_t ld, const double* gains, int64_t gld,
                      void* out, int64_t out_ld, int out_f64, hipStream_t stream)
{
    switch (C) {
        case 1: return launch_bus<T, 1>(rows, voices, x, ld, gains, gld, out, out_ld, out_f64, stream);
        case 2: return launch_bus<T, 2>(rows, voices, x, ld, gains, gld, out, out_ld, out_f64, stream);
        case 4: return launch_bus<T, 4>(rows, voices, x, ld, gains, gld, out, out_ld, out_f64, stream);
    }
    return (int)hipErrorInvalidValue;   // bus widths other than 1, 2, 4: split the call
}

}  // namespace

extern "C" int sig_sum_bus(int64_t rows, int32_t voices, const void* x, int64_t x_ld, int32_t x_dtype,
                           const double* gains, int64_t gains_ld, int32_t bus_channels,
                           void* out, int64_t out_ld, int32_t out_dtype, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && voices >= 0 && x != nullptr && out != nullptr);
    SIG_CHECK_ARG(x_ld >= voices && out_ld >= bus_channels && bus_channels >= 1);
    SIG_CHECK_ARG(gains != nullptr ? gains_ld >= voices : bus_channels == 1);
    SIG_CHECK_ARG(out_dtype == SIG_F32 || out_dtype == SIG_F64);
    if (rows == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (x_dtype == SIG_F32)
        return dispatch_channels<float>(bus_channels, rows, voices, static_cast<const float*>(x), x_ld, gains, gains_ld,
                                        out, out_ld, out_dtype == SIG_F64, s);
    if (x_dtype == SIG_F64)
        return dispatch_channels<double>(bus_channels, rows, voices, static_cast<const double*>(x), x_ld, gains, gains_ld,
                                         out, out_ld, out_dtype == SIG_F64, s);
    return (int)hipErrorInvalidValue;
}
