// Oscillator bank for gfx950: closed-form in absolute position, f64 phase, f32 (or f64) store.
// Replaces Osc._eval + Sine/Square/Sawtooth/Triangle._osc (reference src/signals/chain/osc.py:26-62).
//
// Mapping: one wave = 64*VEC consecutive voices x 16 consecutive rows; a 256-thread workgroup
// stacks 4 waves in time (64 rows).  Lane l owns voices [VEC*l, VEC*l+VEC) of the wave's span, so a
// row is stored as 64 x 16-B lanes = 1 KiB, fully coalesced.  The per-row quotient n/rate (an IEEE
// f64 divide, the expensive op) is computed ONCE per row by lanes 0..15 and broadcast with
// v_readlane, so the per-sample cost is mul + add + waveform.
//
// Roofline: 4 B written per voice-sample (f32), no reads beyond 2 x 8 B per voice per wave.
#include "sig_common.h"

namespace {

constexpr double kPi = 3.141592653589793115997963468544185161590576171875;   // np.pi
constexpr double kPiTail = 1.2246467991473532e-16;                            // pi - fl(pi)
constexpr double kTwoPiHi = 6.28318530717958623199592693708837032318115234375;
constexpr double kTwoPiLo = 2.4492935982947064e-16;

// sin(x) for |x| <= pi/2 + eps, odd Taylor polynomial through x^21 (remainder < 2e-18).
__device__ __forceinline__ double sin_poly(double x) {
    const double s = x * x;
    double p = -1.9572941063391263e-20;                 // -1/21!
    p = fma(p, s, 8.2206352466243295e-18);              //  1/19!
    p = fma(p, s, -2.8114572543455206e-15);             // -1/17!
    p = fma(p, s, 7.6471637318198164e-13);              //  1/15!
    p = fma(p, s, -1.6059043836821613e-10);             // -1/13!
    p = fma(p, s, 2.5052108385441720e-08);              //  1/11!
    p = fma(p, s, -2.7557319223985893e-06);             // -1/9!
    p = fma(p, s, 1.9841269841269841e-04);              //  1/7!
    p = fma(p, s, -8.3333333333333332e-03);             // -1/5!
    p = fma(p, s, 1.6666666666666666e-01);              //  1/3!   (sign folded below)
    // sin x = x - x^3/6 + ... ; the chain above carries alternating signs starting at +1/3!
    return fma(-(x * s), p, x);
}

// np.sin(t * 2 * np.pi) reproduced including the reference's own argument rounding:
//   a = fl(fl(2t) * fl(pi)) is what numpy hands to libm.  a = 2*pi*t + delta with
//   delta = -(fl(2t)*fl(pi) - a) - 2t*(pi - fl(pi)); the first term is exact via fma.
// So sin(a) = sin(2*pi*frac(t) + delta), evaluated with frac(t) exact in f64.
__device__ __forceinline__ double osc_sine(double t) {
    const double t2 = t * 2.0;
    const double a = t2 * kPi;
    const double e = fma(t2, kPi, -a);
    const double delta = -e - t2 * kPiTail;
    const double r = t - rint(t);                       // exact, |r| <= 0.5
    const double k = rint(r + r);                       // -1, 0, +1: half-turns to remove
    const double rq = fma(k, -0.5, r);                  // exact, |rq| <= 0.25
    const double x = fma(rq, kTwoPiHi, fma(rq, kTwoPiLo, delta));
    const double y = sin_poly(x);
    // sin(theta + pi*k) = -sin(theta) for odd k: flip the sign bit, branch-free
    const int flip = (k != 0.0) ? (int)0x80000000 : 0;
    return __hiloint2double(__double2hiint(y) ^ flip, __double2loint(y));
    // domain: |t| < 2^51 cycles (beyond that f64 has no fraction bits left)
}

__device__ __forceinline__ double osc_square(double t) {       // osc.py:48-49
    return sig_sign(0.5 - sig_npmod_pow2<1>(t));
}

__device__ __forceinline__ double osc_sawtooth(double t) {     // osc.py:54-55
    return 2.0 * sig_npmod_pow2<1>(t - 0.5) - 1.0;
}

__device__ __forceinline__ double osc_triangle(double t) {     // osc.py:60-62
    const double u = t - 0.25;
    return (4.0 * sig_npmod_pow2<2>(u) - 1.0) * sig_sign(sig_npmod_pow2<1>(u) - 0.5);
}

// f32 store path of Sine: same exact phase reduction, then the hardware sine (v_sin_f32 takes
// REVOLUTIONS; measured max |err| 1.07e-7 on [-0.25, 0.25]).  Total error vs the reference
// <= 1.3e-7 (bar 1e-6), at 15 f64-rate ops per sample instead of 27, which is what makes the kernel
// HBM-write-bound instead of f64-VALU-bound.
constexpr double kInvTwoPi = 0.15915494309189535;
__device__ __forceinline__ float osc_sine_f32(double t) {
    const double t2 = t + t;
    const double a = t2 * kPi;
    const double e = fma(t2, kPi, -a);                  // a + e == t2 * fl(pi) exactly
    const double s = fma(t2, kPiTail, e);               // -(delta): how far numpy's argument is from 2*pi*t
    const double r = t - rint(t);
    const double k = rint(r + r);
    const double rq = fma(k, -0.5, r);                  // exact, |rq| <= 0.25
    const float rev = (float)fma(s, -kInvTwoPi, rq);
    const float y = __builtin_amdgcn_sinf(rev);
    return __uint_as_float(__float_as_uint(y) ^ ((k != 0.0) ? 0x80000000u : 0u));
}

template <int KIND, typename OUT> __device__ __forceinline__ OUT osc_wave(double t) {
    if (KIND == SIG_OSC_SINE) {
        if (sizeof(OUT) == 4) return (OUT)osc_sine_f32(t);
        return (OUT)osc_sine(t);
    }
    if (KIND == SIG_OSC_SQUARE) return (OUT)osc_square(t);
    if (KIND == SIG_OSC_SAWTOOTH) return (OUT)osc_sawtooth(t);
    return (OUT)osc_triangle(t);
}

constexpr int kRowsPerWave = 16;
constexpr int kWavesPerWg = 4;

template <int KIND, int VEC, typename OUT>
__global__ __launch_bounds__(256) void osc_bank_kernel(
    int64_t position, double rate, int64_t rows, int voices,
    const double* __restrict__ hertz, int hs, const double* __restrict__ phase, int ps,
    OUT* __restrict__ out, int64_t ld, int voice_tiles)
{
    // 1-D grid: consecutive workgroups cover adjacent voice tiles of the same 64 rows
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vt = blockIdx.x % voice_tiles;
    const int64_t rt = blockIdx.x / voice_tiles;
    const int v0 = (vt * SIG_WAVE + lane) * VEC;
    const int64_t r0 = (rt * kWavesPerWg + wave) * kRowsPerWave;
    if (r0 >= rows) return;                                        // wave-uniform

    // osc.py:32  frame_range / rate : int64 -> f64, IEEE divide, one per row
    const double q_lane = (double)(position + r0 + (lane & (kRowsPerWave - 1))) / rate;

    double hz[VEC], ph[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int v = v0 + i;
        hz[i] = (v < voices) ? hertz[(int64_t)v * hs] : 0.0;
        ph[i] = (v < voices && phase) ? phase[(int64_t)v * ps] : 0.0;
    }

#pragma unroll 2                                                    // keep the loop body inside the I-cache
    for (int j = 0; j < kRowsPerWave; ++j) {
        const int64_t row = r0 + j;
        if (row >= rows) break;                                    // wave-uniform
        const double q = sig_readlane_f64(q_lane, j);
        OUT y[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const double t = q * hz[i] + ph[i];                    // two roundings, like numpy
            y[i] = osc_wave<KIND, OUT>(t);
        }
        OUT* dst = out + row * ld + v0;
        if (VEC == 4) {
            if (v0 < voices) {                                     // voices % 4 == 0 on this path
                typename sig_vec4<OUT>::type o;
                o.x = y[0]; o.y = y[1]; o.z = y[2]; o.w = y[3];
                *reinterpret_cast<typename sig_vec4<OUT>::type*>(dst) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (v0 + i < voices) dst[i] = y[i];
        }
    }
}

template <int KIND, typename OUT>
int launch_osc(int64_t position, int32_t rate, int64_t rows, int32_t voices,
               const double* hertz, int hs, const double* phase, int ps,
               OUT* out, int64_t ld, hipStream_t stream)
{
    const bool vec4 = (voices % 4 == 0) && (ld % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) % (4 * sizeof(OUT))) == 0);
    const int64_t rows_per_wg = (int64_t)kRowsPerWave * kWavesPerWg;
    const int64_t row_tiles = (rows + rows_per_wg - 1) / rows_per_wg;
    const int span = SIG_WAVE * (vec4 ? 4 : 1);
    const int voice_tiles = (voices + span - 1) / span;
    const int64_t nwg = row_tiles * voice_tiles;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (vec4)
        osc_bank_kernel<KIND, 4, OUT><<<(unsigned)nwg, 256, 0, stream>>>(position, (double)rate, rows, voices,
                                                                           hertz, hs, phase, ps, out, ld, voice_tiles);
    else
        osc_bank_kernel<KIND, 1, OUT><<<(unsigned)nwg, 256, 0, stream>>>(position, (double)rate, rows, voices,
                                                                           hertz, hs, phase, ps, out, ld, voice_tiles);
    return sig_launch_status();
}

template <typename OUT>
int dispatch_kind(int kind, int64_t position, int32_t rate, int64_t rows, int32_t voices,
                  const double* hertz, int hs, const double* phase, int ps,
                  OUT* out, int64_t ld, hipStream_t stream)
{
    switch (kind) {
        case SIG_OSC_SINE: return launch_osc<SIG_OSC_SINE, OUT>(position, rate, rows, voices, hertz, hs, phase, ps, out, ld, stream);
        case SIG_OSC_SQUARE: return launch_osc<SIG_OSC_SQUARE, OUT>(position, rate, rows, voices, hertz, hs, phase, ps, out, ld, stream);
        case SIG_OSC_SAWTOOTH: return launch_osc<SIG_OSC_SAWTOOTH, OUT>(position, rate, rows, voices, hertz, hs, phase, ps, out, ld, stream);
        case SIG_OSC_TRIANGLE: return launch_osc<SIG_OSC_TRIANGLE, OUT>(position, rate, rows, voices, hertz, hs, phase, ps, out, ld, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int sig_osc_bank(int kind, int64_t position, int32_t rate, int64_t rows, int32_t voices,
                            const double* hertz, int32_t hertz_stride,
                            const double* phase, int32_t phase_stride,
                            void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && voices >= 0 && rate > 0 && position >= 0);
    SIG_CHECK_ARG(hertz != nullptr && out != nullptr && out_ld >= voices);
    SIG_CHECK_ARG((hertz_stride == 0 || hertz_stride == 1) && (phase_stride == 0 || phase_stride == 1));
    if (rows == 0 || voices == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == SIG_F32)
        return dispatch_kind<float>(kind, position, rate, rows, voices, hertz, hertz_stride, phase, phase_stride,
                                    static_cast<float*>(out), out_ld, s);
    if (out_dtype == SIG_F64)
        return dispatch_kind<double>(kind, position, rate, rows, voices, hertz, hertz_stride, phase, phase_stride,
                                     static_cast<double*>(out), out_ld, s);
    return (int)hipErrorInvalidValue;
}

extern "C" int sig_abi_version(void) { return SIG_ABI_VERSION; }
