// ADSR envelope evaluation shared by adsr.hip (the node) and biquad.hip (filter epilogue).
// Definition: oracle/chain_ref.py:adsr.  f64, contract off.
#pragma once
#include "sig_common.h"

namespace sig_env {

struct AdsrRows { const double* p[6]; int s[6]; };   // attack, decay, sustain, release, gate_on, gate_off

// clip(x, 0, 1) as two full-rate instructions (v_max_f64, v_min_f64) instead of two compares and two 64-bit selects.
// Same value as np.clip for every non-NaN x (a NaN envelope parameter is outside the definition).
__device__ __forceinline__ double clip01(double x) { return fmin(fmax(x, 0.0), 1.0); }

// Per-voice constants.  The "a zero-length stage counts as complete" cases of the definition are folded into
// constants so that the per-row code has no select for them:
//   attack == 0: ia = 0 and the attack value is never chosen (u >= 0 implies v = u - 0 >= 0);
//   decay == 0:  id = 0, d_bias = 1  ->  clip(fma(v, 0, 1)) = 1;   decay > 0: d_bias = 0 and fma(v, id, 0) == v * id;
//   release == 0: ir = 0, rel_bias = 0  ->  clip(0 - w * 0) = 0;   release > 0: rel_bias = 1  ->  clip(1 - w * ir).
struct Voice { double ia, id, sm1, ir, on, off, attack, hold_off, d_bias, rel_bias; };

__device__ __forceinline__ double held(const Voice& p, double t) {
    const double u = t - p.on;
    const double v = u - p.attack;
    const double a = clip01(u * p.ia);
    const double d = clip01(fma(v, p.id, p.d_bias));
    return (u < 0.0) ? 0.0 : ((v < 0.0) ? a : 1.0 + p.sm1 * d);
}

__device__ __forceinline__ Voice load_voice(const AdsrRows& in, int v) {
    Voice p;
    const double attack = in.p[0][(int64_t)v * in.s[0]], decay = in.p[1][(int64_t)v * in.s[1]];
    const double sustain = in.p[2][(int64_t)v * in.s[2]], release = in.p[3][(int64_t)v * in.s[3]];
    p.on = in.p[4][(int64_t)v * in.s[4]];
    p.off = in.p[5][(int64_t)v * in.s[5]];
    p.attack = attack;
    p.ia = (attack > 0.0) ? 1.0 / attack : 0.0;
    p.id = (decay > 0.0) ? 1.0 / decay : 0.0;
    p.ir = (release > 0.0) ? 1.0 / release : 0.0;
    p.d_bias = (decay > 0.0) ? 0.0 : 1.0;
    p.rel_bias = (release > 0.0) ? 1.0 : 0.0;
    p.sm1 = sustain - 1.0;
    p.hold_off = held(p, p.off);
    return p;
}

// envelope level at time t (seconds)
__device__ __forceinline__ double level(const Voice& p, double t) {
    const double w = t - p.off;
    const double rel = clip01(p.rel_bias - w * p.ir);
    return (w < 0.0) ? held(p, t) : p.hold_off * rel;
}

// Piecewise-linear tracking for kernels that walk time forwards and only need the envelope to rounding (the fused
// bus kernel, not the ADSR node itself): within one stage  level(t) = L0 + slope * (t - t0)  exactly up to rounding,
// so a voice carries (t0, L0, slope, end) and re-derives them from the definition only when t reaches `end` -- at
// most five times per voice over a whole stream.  The stage boundaries are those of the definition (gate_on,
// + attack, + decay, gate_off, + release); a boundary computed here may sit one ulp of t away from where the
// definition's comparisons switch, which moves the level by slope * ulp(t) < 1e-9.
struct Segment { double t0, l0, slope, end; };

__device__ __forceinline__ Segment segment_at(const Voice& p, double t) {
    const double inf = __builtin_inf();
    Segment s;
    s.t0 = t;
    s.l0 = level(p, t);
    if (t < p.off) {
        const double u = t - p.on, v = u - p.attack;
        if (u < 0.0) { s.slope = 0.0; s.end = fmin(p.on, p.off); }
        else if (v < 0.0) { s.slope = p.ia; s.end = fmin(p.on + p.attack, p.off); }
        else if (fma(v, p.id, p.d_bias) < 1.0) { s.slope = p.sm1 * p.id; s.end = fmin(p.on + p.attack + 1.0 / p.id, p.off); }
        else { s.slope = 0.0; s.end = p.off; }
    } else {
        const double w = t - p.off;
        if (p.rel_bias - w * p.ir > 0.0) { s.slope = -(p.hold_off * p.ir); s.end = p.off + 1.0 / p.ir; }
        else { s.slope = 0.0; s.end = inf; }
    }
    if (!(s.end > t)) s.end = inf;                      // NaN parameters or a boundary at t itself: never re-enter for this t
    return s;
}

inline bool load_rows(const double* const* params, const int32_t* strides, AdsrRows& in) {
    if (!params || !strides) return false;
    for (int i = 0; i < 6; ++i) {
        if (params[i] == nullptr || (strides[i] != 0 && strides[i] != 1)) return false;
        in.p[i] = params[i];
        in.s[i] = strides[i];
    }
    return true;
}

}  // namespace sig_env
