// Butterworth biquad design shared by biquad.hip and fused_voice.hip.
#pragma once
#include "sig_common.h"

namespace sig_biquad {

constexpr double kPi = 3.141592653589793115997963468544185161590576171875;
constexpr double kSqrt2 = 1.4142135623730951454746218587388284504413604736328125;

struct Biquad { double b0, b1, b2, a1, a2; };

// Closed form of scipy.signal.butter(2, wn, 'lp'|'hp', output='sos') -- oracle/chain_ref.py:butter2_sos.
// Returns false (and NaN coefficients) where scipy raises: wn <= 0 or wn >= 1 after the clip (fx.py:99-102).
__device__ __forceinline__ bool design_butter2(int type, double cutoff, double rate, Biquad& q) {
    double wn = cutoff / (rate * 0.5);                      // scaled_crit /= rate / 2
    wn = (wn < 0.0) ? 0.0 : ((wn > 1.0) ? 1.0 : wn);        // clip(0, 1); NaN stays NaN
    const bool bad = !(wn > 0.0 && wn < 1.0);               // scipy: `if not (all(Wn > 0) and all(Wn < 1)): raise` -- NaN raises too
    const double k = tan(kPi * wn / 2.0);
    const double k2 = k * k;
    const double nrm = 1.0 / (1.0 + kSqrt2 * k + k2);
    const double nan = __builtin_nan("");
    if (type == SIG_FILT_LOWPASS) { q.b0 = k2 * nrm; q.b1 = 2.0 * k2 * nrm; q.b2 = q.b0; }
    else                          { q.b0 = nrm;      q.b1 = -2.0 * nrm;     q.b2 = nrm;  }
    q.a1 = 2.0 * (k2 - 1.0) * nrm;
    q.a2 = (1.0 - kSqrt2 * k + k2) * nrm;
    if (bad) { q.b0 = q.b1 = q.b2 = q.a1 = q.a2 = nan; }
    return !bad;
}

// ---- 4th-order band filters: butter(2, [lo, hi], 'bp'|'bs', output='sos') = two sections --------------
// Restatement of scipy's chain (buttap -> lp2bp_zpk / lp2bs_zpk -> bilinear_zpk -> zpk2sos, pairing
// 'nearest'), fs = 2:  w = 4 tan(pi Wn / 2), bw = w2 - w1, wo = sqrt(w1 w2); prototype pole p = (-1+j)/sqrt2
//   bp: q = p bw/2 +- sqrt((p bw/2)^2 - wo^2), zeros {+1,+1,-1,-1} after the bilinear map, k = bw^2
//   bs: q = (bw/2)/p +- sqrt(((bw/2)/p)^2 - wo^2), zeros {z0,z0,z0*,z0*}, z0 = (4 + j wo)/(4 - j wo), k = 1
//   P = (4 + q)/(4 - q);  k_z = k prod(4 - z) / prod(4 - q)
// zpk2sos: the pole pair closest to the unit circle goes LAST and takes the nearest zeros (bp: the two
// nearest of the real zeros, one at a time; bs: a conjugate zero pair); the gain multiplies the first
// section.  Checked against scipy over 4000 random bands: coefficients within 6e-13 relative
// (tests/test_oracle_build_defined.py restates the same arithmetic in Python).
struct Cx { double re, im; };
__device__ __forceinline__ Cx cx_mul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cx cx_div(Cx a, Cx b) {
    const double d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ __forceinline__ Cx cx_sqrt(Cx z) {
    const double m = hypot(z.re, z.im);
    if (m == 0.0) return {0.0, 0.0};
    if (z.re >= 0.0) { const double t = sqrt((m + z.re) * 0.5); return {t, z.im / (2.0 * t)}; }
    const double t = sqrt((m - z.re) * 0.5);
    return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}

// type: SIG_FILT_BANDPASS / SIG_FILT_BANDSTOP.  Returns false where scipy raises (Wn outside (0,1) after the
// clip, or lo >= hi); the coefficients are then NaN.
__device__ __forceinline__ bool design_band2(int type, double lo_hz, double hi_hz, double rate, Biquad& first, Biquad& last) {
    auto scaled = [&](double hz) { double w = hz / (rate * 0.5); return (w < 0.0) ? 0.0 : ((w > 1.0) ? 1.0 : w); };
    const double wl = scaled(lo_hz), wh = scaled(hi_hz);
    const bool bad = !(wl > 0.0 && wl < 1.0 && wh > 0.0 && wh < 1.0 && wl < wh);   // NaN fails every comparison: bad
    const double w1 = 4.0 * tan(kPi * wl / 2.0), w2 = 4.0 * tan(kPi * wh / 2.0);
    const double bw = w2 - w1, wo2 = w1 * w2;
    const Cx p = {-0.70710678118654757, 0.70710678118654757};
    Cx c;                                                   // centre: p bw/2 (bp) or (bw/2)/p (bs)
    if (type == SIG_FILT_BANDPASS) c = {p.re * bw * 0.5, p.im * bw * 0.5};
    else c = cx_div({bw * 0.5, 0.0}, p);
    Cx c2 = cx_mul(c, c);
    const Cx s = cx_sqrt({c2.re - wo2, c2.im});
    const Cx qa = {c.re + s.re, c.im + s.im}, qb = {c.re - s.re, c.im - s.im};
    auto bilinear = [](Cx q) { Cx r = cx_div({4.0 + q.re, q.im}, {4.0 - q.re, -q.im}); if (r.im < 0.0) r.im = -r.im; return r; };
    const Cx Pa = bilinear(qa), Pb = bilinear(qb);
    const double den = ((4.0 - qa.re) * (4.0 - qa.re) + qa.im * qa.im) * ((4.0 - qb.re) * (4.0 - qb.re) + qb.im * qb.im);
    const bool a_worst = fabs(1.0 - hypot(Pa.re, Pa.im)) <= fabs(1.0 - hypot(Pb.re, Pb.im));
    const Cx worst = a_worst ? Pa : Pb, other = a_worst ? Pb : Pa;
    double kz, bl1, bl2, bf1, bf2;                          // b = [1, b1, b2] per section (before the gain)
    if (type == SIG_FILT_BANDPASS) {
        kz = bw * bw * 16.0 / den;
        // nearest real zeros to the worst pole, one at a time, from {+1, +1, -1, -1}
        const double dp = hypot(worst.re - 1.0, worst.im), dm = hypot(worst.re + 1.0, worst.im);
        // first pick: nearer of +1 / -1; second pick: nearer of what is left (the same value again is still available)
        const double z1 = (dp <= dm) ? 1.0 : -1.0;
        const double z2 = z1;                               // two copies of each zero exist, so the second pick repeats
        bl1 = -(z1 + z2); bl2 = z1 * z2;
        bf1 = -(-z1 - z2); bf2 = z1 * z2;                   // the other section gets the two zeros of opposite sign
    } else {
        const Cx z0 = cx_div({4.0, sqrt(wo2)}, {4.0, -sqrt(wo2)});
        const double num = (16.0 + wo2) * (16.0 + wo2);
        kz = num / den;
        bl1 = bf1 = -2.0 * z0.re; bl2 = bf2 = 1.0;
    }
    first = {kz, kz * bf1, kz * bf2, -2.0 * other.re, other.re * other.re + other.im * other.im};
    last = {1.0, bl1, bl2, -2.0 * worst.re, worst.re * worst.re + worst.im * worst.im};
    if (bad) {
        const double nan = __builtin_nan("");
        first = {nan, nan, nan, nan, nan};
        last = first;
    }
    return !bad;
}

}  // namespace sig_biquad
