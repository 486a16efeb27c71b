// Cold-start 4th-order Butterworth band filters for gfx950: BandPass / BandStop as two DF2T biquad
// sections in series.  The reference's DoubleCritFilter (src/signals/chain/fx.py:132-163)
// raises TypeError at fx.py:99 on every block; its intent is
//   sos = butter(N=2, Wn=[low, high]/(rate/2), btype='bp'|'bs', output='sos');  y = sosfilt(sos, window)
// with the same [<=100 context | block] cold start as LowPass/HighPass.  Same mapping as biquad.hip's plain
// kernel (wave = 64*VPT voices of one block, serial over rows, register ring of row loads); 4 f64 state
// registers and 10 coefficients per voice.  HBM-bound: 8 B per voice-sample, 18 f64 ops.
#include "sig_biquad.h"

namespace {

using sig_biquad::Biquad;
using sig_biquad::design_band2;

template <typename T, int VPT> struct RowVec;
template <> struct RowVec<float, 1> { using type = float; };
template <> struct RowVec<double, 1> { using type = double; };
template <> struct RowVec<float, 4> { using type = float4; };
template <> struct RowVec<double, 4> { using type = double4; };

template <typename V> __device__ __forceinline__ void unpack(const V& v, double (&x)[1]) { x[0] = (double)v; }
__device__ __forceinline__ void unpack(const float4& v, double (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
__device__ __forceinline__ void unpack(const double4& v, double (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
__device__ __forceinline__ void pack(float& v, const double (&y)[1]) { v = (float)y[0]; }
__device__ __forceinline__ void pack(double& v, const double (&y)[1]) { v = y[0]; }
__device__ __forceinline__ void pack(float4& v, const double (&y)[4]) { v = make_float4((float)y[0], (float)y[1], (float)y[2], (float)y[3]); }
__device__ __forceinline__ void pack(double4& v, const double (&y)[4]) { v = make_double4(y[0], y[1], y[2], y[3]); }

constexpr int kRing = 8;

template <typename T, int VPT>
__global__ __launch_bounds__(256) void band_coldstart_kernel(
    int type, double rate, int64_t position, int N, int K, int ctx, int voices,
    const double* __restrict__ low, int ls, const double* __restrict__ high, int hs,
    const T* __restrict__ in, int64_t in_ld, T* __restrict__ out, int64_t out_ld,
    int voice_tiles, int* __restrict__ status)
{
    using Vec = typename RowVec<T, VPT>::type;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int vt = (int)(item % voice_tiles);
    const int64_t b = item / voice_tiles;
    if (b >= K) return;
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live = v0 < voices;
    const int vc = live ? v0 : 0;
    const int64_t p_b = position + b * N;
    const int c = (int)((p_b < (int64_t)ctx) ? p_b : (int64_t)ctx);

    Biquad q1[VPT], q2[VPT];
    double s10[VPT], s11[VPT], s20[VPT], s21[VPT];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = (vc + i < voices) ? vc + i : vc;
        ok &= design_band2(type, low[(int64_t)v * ls], high[(int64_t)v * hs], rate, q1[i], q2[i]);
        s10[i] = s11[i] = s20[i] = s21[i] = 0.0;
    }
    if (!ok && live && status) atomicOr(status, SIG_STATUS_BAD_CUTOFF);

    const int total = c + N;
    const T* src = in + (b * N - c) * in_ld + vc;
    T* dst = out + (b * N - c) * out_ld + vc;
    Vec ring[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
        const int r = (u < total) ? u : total - 1;
        ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)r * in_ld);
    }
    for (int r0 = 0; r0 < total; r0 += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            const int r = r0 + u;
            if (r >= total) break;
            double x[VPT], y[VPT];
            unpack(ring[u], x);
            const int rn = (r + kRing < total) ? r + kRing : total - 1;
            ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)rn * in_ld);
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                // scipy _sosfilt: section by section for each sample, one rounding per op
                const double y1 = q1[i].b0 * x[i] + s10[i];
                s10[i] = q1[i].b1 * x[i] - q1[i].a1 * y1 + s11[i];
                s11[i] = q1[i].b2 * x[i] - q1[i].a2 * y1;
                y[i] = q2[i].b0 * y1 + s20[i];
                s20[i] = q2[i].b1 * y1 - q2[i].a1 * y[i] + s21[i];
                s21[i] = q2[i].b2 * y1 - q2[i].a2 * y[i];
            }
            if (r >= c && live) {
                Vec o; pack(o, y);
                *reinterpret_cast<Vec*>(dst + (int64_t)r * out_ld) = o;
            }
        }
    }
}

template <typename T>
int launch_band(int type, int32_t rate, int64_t position, int32_t N, int32_t K, int32_t ctx, int32_t voices,
                const double* low, int ls, const double* high, int hs,
                const T* in, int64_t in_ld, T* out, int64_t out_ld, int32_t* status, hipStream_t stream)
{
    const bool vec4 = (voices % 4 == 0) && (in_ld % 4 == 0) && (out_ld % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(in) % (4 * sizeof(T)) == 0) &&
                      (reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0);
    const int span = SIG_WAVE * (vec4 ? 4 : 1);
    const int voice_tiles = (voices + span - 1) / span;
    const int64_t nwg = ((int64_t)voice_tiles * K + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (vec4)
        band_coldstart_kernel<T, 4><<<(unsigned)nwg, 256, 0, stream>>>(type, (double)rate, position, N, K, ctx, voices,
                                                                         low, ls, high, hs, in, in_ld, out, out_ld, voice_tiles, status);
    else
        band_coldstart_kernel<T, 1><<<(unsigned)nwg, 256, 0, stream>>>(type, (double)rate, position, N, K, ctx, voices,
                                                                         low, ls, high, hs, in, in_ld, out, out_ld, voice_tiles, status);
    return sig_launch_status();
}

}  // namespace

extern "C" int sig_band_coldstart(int type, int32_t rate, int64_t position,
                                  int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                  const double* low, int32_t low_stride, const double* high, int32_t high_stride,
                                  const void* in, int64_t in_ld, int64_t in_history,
                                  void* out, int64_t out_ld, int32_t dtype,
                                  int32_t* status, void* stream)
{
    SIG_CHECK_ARG(type == SIG_FILT_BANDPASS || type == SIG_FILT_BANDSTOP);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(low && high && in && out && in_ld >= voices && out_ld >= voices);
    SIG_CHECK_ARG((low_stride | 1) == 1 && (high_stride | 1) == 1);
    SIG_CHECK_ARG(in_history >= (position < context ? position : context));
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SIG_F32)
        return launch_band<float>(type, rate, position, block_frames, nblocks, context, voices, low, low_stride, high, high_stride,
                                  static_cast<const float*>(in), in_ld, static_cast<float*>(out), out_ld, status, s);
    if (dtype == SIG_F64)
        return launch_band<double>(type, rate, position, block_frames, nblocks, context, voices, low, low_stride, high, high_stride,
                                   static_cast<const double*>(in), in_ld, static_cast<double*>(out), out_ld, status, s);
    return (int)hipErrorInvalidValue;
}
