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
    const bool bad = (wn <= 0.0) || (wn >= 1.0);
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

}  // namespace sig_biquad
