// ADSR envelope evaluation shared by adsr.hip (the node) and biquad.hip (filter epilogue).
// Definition: oracle/chain_ref.py:adsr.  f64, contract off.
#pragma once
#include "sig_common.h"

namespace sig_env {

struct AdsrRows { const double* p[6]; int s[6]; };   // attack, decay, sustain, release, gate_on, gate_off

__device__ __forceinline__ double clip01(double x) { return (x < 0.0) ? 0.0 : ((x > 1.0) ? 1.0 : x); }

struct Voice { double ia, id, sm1, ir, on, off, attack, hold_off; };

__device__ __forceinline__ double held(const Voice& p, double t) {
    const double u = t - p.on;
    const double v = u - p.attack;
    const double a = (p.ia > 0.0) ? clip01(u * p.ia) : 1.0;
    const double d = (p.id > 0.0) ? clip01(v * p.id) : 1.0;
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
    p.sm1 = sustain - 1.0;
    p.hold_off = held(p, p.off);
    return p;
}

// envelope level at time t (seconds)
__device__ __forceinline__ double level(const Voice& p, double t) {
    const double w = t - p.off;
    const double rel = (p.ir > 0.0) ? clip01(1.0 - w * p.ir) : 0.0;
    return (w < 0.0) ? held(p, t) : p.hold_off * rel;
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
