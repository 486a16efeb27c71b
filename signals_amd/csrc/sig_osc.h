// Oscillator waveforms shared by osc_bank.hip and fused_voice.hip: `t` (cycles, f64) -> sample.
// Reference: src/signals/chain/osc.py:40-62.  See osc_bank.hip for the precision notes.
#pragma once
#include "sig_common.h"

namespace sig_osc {

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
//   a = fl(fl(2t) * fl(pi)) = fl(t * 2fl(pi)) is what numpy hands to libm;  a = 2*pi*t - s  with
//   s = (t*2fl(pi) - a) + t*2(pi - fl(pi)); the first term is exact via fma.
// So sin(a) = sin(2*pi*(t - K/2) + pi*K - s) with K = rint(2t): rq = t - K/2 is exact in f64, |rq| <= 1/4,
// and an odd K flips the sign.  K and its parity come from one magic-number add (RNE to an integer in the
// low mantissa bits), valid for |t| < 2^50 cycles -- beyond that f64 has no fraction bits left anyway.
constexpr double kRoundMagic = 6755399441055744.0;      // 1.5 * 2^52
constexpr double kTwoPi = 2.0 * kPi;                     // exact doubling of fl(pi)
constexpr double kTwoPiTail = 2.0 * kPiTail;

struct SinePhase { double rq; double s; unsigned flip; };   // quarter-range revolutions, -(delta), sign bit

__device__ __forceinline__ SinePhase sine_phase(double t) {
    SinePhase p;
    const double a = t * kTwoPi;
    const double e = fma(t, kTwoPi, -a);                // a + e == t * 2fl(pi) exactly
    p.s = fma(t, kTwoPiTail, e);
    const double u = fma(t, 2.0, kRoundMagic);          // = fl(2t + M): low mantissa bits = rint(2t)
    const double k = u - kRoundMagic;                   // exact
    p.rq = fma(k, -0.5, t);                             // exact, |rq| <= 0.25
    p.flip = ((unsigned)__double2loint(u) & 1u) << 31;  // parity of K
    return p;
}

// f64 store path (block-rate control values): polynomial on [-pi/2, pi/2], ~1 ulp(f64)
__device__ __forceinline__ double osc_sine(double t) {
    const SinePhase p = sine_phase(t);
    const double x = fma(p.rq, kTwoPiHi, fma(p.rq, kTwoPiLo, -p.s));
    const double y = sin_poly(x);
    return __hiloint2double(__double2hiint(y) ^ (int)p.flip, __double2loint(y));
}

__device__ __forceinline__ double osc_square(double t) {       // osc.py:48-49
    return sig_sign(0.5 - sig_npmod_pow2<1>(t));
}

__device__ __forceinline__ double osc_sawtooth(double t) {     // osc.py:54-55
    return fma(sig_npmod_pow2<1>(t - 0.5), 2.0, -1.0);     // 2m is exact: the same single rounding as 2*m - 1
}

__device__ __forceinline__ double osc_triangle(double t) {     // osc.py:60-62
    const double u = t - 0.25;
    return (4.0 * sig_npmod_pow2<2>(u) - 1.0) * sig_sign(sig_npmod_pow2<1>(u) - 0.5);
}

// f32 store path of Sine: same exact phase reduction, then the hardware sine (v_sin_f32 takes
// REVOLUTIONS; measured max |err| 1.07e-7 on [-0.25, 0.25]).  Total error vs the reference
// <= 1.3e-7 (bar 1e-6), at ~13 f64-rate ops per sample instead of 27, which is what makes the oscillator
// kernel HBM-write-bound instead of f64-VALU-bound.
constexpr double kInvTwoPi = 0.15915494309189535;
__device__ __forceinline__ float osc_sine_f32(double t) {
    const SinePhase p = sine_phase(t);
    const float rev = (float)fma(p.s, -kInvTwoPi, p.rq);
    return __uint_as_float(__float_as_uint(__builtin_amdgcn_sinf(rev)) ^ p.flip);
}

// Fast Sine phase for f64-issue-bound kernels that walk consecutive rows: within a chunk of rows the phase
// advances by d = hertz/rate revolutions per row, so  t_j ~ t_0 + j*d  and only frac(t_0) (exact) and d are
// needed.  Relative to the exact path this drops (a) the tracking of numpy's own argument rounding,
// |fl(t*2pi) - 2pi*t| <= 9.4e-16*|t| rad, and (b) the roundings inside fl(fl(n/rate)*hertz)+phase, <= ulp(t)
// cycles: together <= 3 |t| 2^-53 cycles = 2.1e-15 |t| rad per row of reference rounding noise that the incremental
// phase does not reproduce, plus the same bound once more for the seed (which carries the reference's rounding at the
// span's first row): < 2.9e-7 while |t| < 2^26 cycles, a third of the 1e-6 bar, which the caller checks per wave
// (kSineFastMaxT); beyond that the exact path is used.  f_j = fma(j, d, f0) is a single rounding from exact operands.
constexpr double kSineFastMaxT = 67108864.0;            // 2^26 cycles (10.6 h at 1760 Hz, 56 min at 20 kHz)

__device__ __forceinline__ float osc_sine_f32_fast(double f0, double d, double j) {
    const double f = fma(j, d, f0);                     // |f| <= 0.5 + 64*0.5
    const double u = fma(f, 2.0, kRoundMagic);
    const double k = u - kRoundMagic;
    const float rev = (float)fma(k, -0.5, f);           // |rev| <= 0.25
    const unsigned flip = ((unsigned)__double2loint(u) & 1u) << 31;
    return __uint_as_float(__float_as_uint(__builtin_amdgcn_sinf(rev)) ^ flip);
}

// Sawtooth for the fused kernels, whose oscillator sample feeds an f64 filter: m = t' - floor(t') as ONE instruction
// (v_fract_f64).  It differs from sig_npmod_pow2 only where the rounded subtraction would reach 1.0 (t' in (-2^-54, 0)):
// v_fract_f64 answers the largest double below 1 instead, 2.2e-16 away in the sample; the per-node oscillator kernel, whose
// float32 output is compared bit for bit, keeps the two-instruction form.
__device__ __forceinline__ double osc_sawtooth_fract(double t) {
    return fma(__builtin_amdgcn_fract(t - 0.5), 2.0, -1.0);
}

// Square for the fused kernels: sign(d), d = 0.5 - m, as copysign(1, d) (one v_and_or_b32 on the high word) kept where
// |d| > 0 and d itself (a zero, or a NaN from NaN parameters) elsewhere -- one compare and two 32-bit selects instead of
// np.sign's two compares and four.  Same values as osc_square except in v_fract_f64's corner (see above), where both
// answer -1.
__device__ __forceinline__ double osc_square_fract(double t) {
    const double d = 0.5 - __builtin_amdgcn_fract(t);
    const double one = __hiloint2double((int)(((unsigned)__double2hiint(d) & 0x80000000u) | 0x3ff00000u), 0);
    return (fabs(d) > 0.0) ? one : d;
}

// Triangle for the fused kernels.  The reference's (4 mod(u, 1/2) - 1) * sign(mod(u, 1) - 1/2), u = t - 1/4, is
// 4 |m - 1/2| - 1 with m = mod(u, 1), in one rounding of the same real number on either branch (m - 1/2 and mod(u, 1/2)
// are exact), except AT m = 1/2, where sign(0) makes it a zero: kept.  5 instructions + the zero fix instead of 15.
__device__ __forceinline__ double osc_triangle_fract(double t) {
    const double d = __builtin_amdgcn_fract(t - 0.25) - 0.5;
    const double r = fma(fabs(d), 4.0, -1.0);
    return (d == 0.0) ? 0.0 : r;
}

template <int KIND> __device__ __forceinline__ double osc_wave_fused(double t) {
    if (KIND == SIG_OSC_SQUARE) return osc_square_fract(t);
    if (KIND == SIG_OSC_SAWTOOTH) return osc_sawtooth_fract(t);
    if (KIND == SIG_OSC_TRIANGLE) return osc_triangle_fract(t);
    return osc_sine(t);
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

}  // namespace sig_osc
