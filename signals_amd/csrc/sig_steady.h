// Pieces shared by the fused kernels' translation units (fused_voice.hip, fused_mix.hip): the launch arguments, and the
// per-voice constants of the closed form (a sinusoid through an LTI filter: steady state + homogeneous part, see
// fused_voice.hip "closed form").
#pragma once
#include "sig_biquad.h"
#include "sig_osc.h"

namespace sig_fused {

using sig_biquad::Biquad;
using sig_biquad::design_butter2;

struct FusedArgs {
    int type; double rate; int64_t position; int N, K, ctx, voices;
    const double* hertz; int hs; const double* phase; int ps;
    const double* cutoff; int cs; const double* gain; int gs;
    float* out; int64_t out_ld; int voice_tiles; int* status;
    const int64_t* pos_dev = nullptr;        // when set, the position is read from device memory (hipGraph replay)
    const float* mix = nullptr;              // C == -1: the (64, 64) row-major mix matrix
    int span = 1;                            // consecutive blocks per lane (> 1 needs N >= ctx)
    int steady = 0;                          // Sine + bus: waves passing steady_wave() are done by fused_steady_bus_kernel
    const double* steady_consts = nullptr;   // its per-voice constants (steady_prep_kernel)
    double* consts_ext = nullptr;            // caller-held buffer for them (sig_fused_voice_bus_prepared), else the workspace tail
    int consts_ready = 0;                    // the caller vouches that consts_ext already holds them: no prep launch
    int force_walk = 0;                      // sig_fused_voice_bus_walk: never the closed form
    int cutoff_rows = 1, gain_rows = 1;      // > 1: one (1,V)|(1,1) parameter row PER BLOCK (sig_fused_*_rows), row b at + b * (stride ? voices : 1)
    // sig_fused_*_fm: hertz / phase also per block (block-rate FM, osc.py:28-30).  The reference's oscillators keep their
    // previous block (BlockCachingEmitter), so the context rows in front of block b are block b - 1's samples, made with row
    // b - 1; for the launch's first block that row is *_hist (the previous batch's last row on a continuing stream, the
    // controls at position - min(context, position) on a fresh graph); a non-null *_hist is what marks the parameter as modulated
    int hertz_rows = 1, phase_rows = 1;
    const double* hertz_hist = nullptr; const double* phase_hist = nullptr;
    // sig_fused_*_pair: the filter reads Mix(A, B, mix) (pair_op 1) or RingMod(A, B) (pair_op 2) of TWO oscillators
    int pair_op = 0, kind2 = 0;
    const double* hertz2 = nullptr; int hs2 = 0; const double* phase2 = nullptr; int ps2 = 0; const double* mixrow = nullptr; int ms = 0;
};

struct BusArgs {
    const double* pan; int64_t pan_ld; double* partials; int64_t rows;
    float* out = nullptr; int64_t out_ld = 0;   // set: the kernel adds the voice tiles itself (sig_bus::sum_tiles_in_workgroup), no partials_kernel launch
};

// Tuning / test hooks of the fused entry points (fused_voice.hip): one instance, read by both of its translation units
struct Tuning { int vpt = 0, span = 0, steady = -1, scan = -1, tile_sum_kernel = 0, mix_f32 = 0; };     // 0 / -1 = the launch heuristics decide
Tuning& tuning();
// fused_voice_b.hip: the walkers of Square / Sawtooth / Triangle, called by fused_voice.hip's dispatchers
int part_b_rows(int C, int kind, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s);
int part_b_bus(int gain, int kind, int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s);
int part_b_mix(int gain, int kind, const FusedArgs& a, hipStream_t s);
int part_b_chain(int gain, int kind, const FusedArgs& a, hipStream_t s);

// sin(2 pi f) and cos(2 pi f) in f64 (~1 ulp), any |f| < 2^50: quarter-range reduction by the magic-number
// rint of sig_osc.h, true 2 pi as hi + lo
__device__ __forceinline__ double sin2pi(double f) {
    const double u = fma(f, 2.0, sig_osc::kRoundMagic);
    const double k = u - sig_osc::kRoundMagic;
    const double rq = fma(k, -0.5, f);
    const double y = sig_osc::sin_poly(fma(rq, sig_osc::kTwoPiHi, rq * sig_osc::kTwoPiLo));
    return __hiloint2double(__double2hiint(y) ^ (int)(((unsigned)__double2loint(u) & 1u) << 31), __double2loint(y));
}

struct M2 { double a, b, c, d; };                                             // [[a, b], [c, d]]
__device__ __forceinline__ M2 m2_mul(const M2& x, const M2& y) {
    return {fma(x.a, y.a, x.b * y.c), fma(x.a, y.b, x.b * y.d), fma(x.c, y.a, x.d * y.c), fma(x.c, y.b, x.d * y.d)};
}

constexpr double kHomogeneousTol = 1e-9;      // of one voice's full scale (unit-amplitude oscillator, before gain and pan): a bus of V voices is off by
                                              // at most V x 1e-9 x |weight| where every dropped part adds up coherently -- 5e-8 of ITS full scale (~ sqrt(V) weights)

// per-row phase step and whether the closed form applies to a voice for rows [first, last]: |t| < 2^26 cycles over
// the span (as for the walker's Sine recurrence), at most a quarter turn per row, and sin(theta) not tiny (the map
// from (yss, dss) back to the complex amplitude divides by it: below ~8 Hz at 48 kHz the walker is used instead)
__device__ __forceinline__ bool steady_voice_ok(double hz, double ph, double rate, double st, double q_first, double q_last) {
    const double t_first = q_first * hz + ph, t_last = q_last * hz + ph;
    const double d = hz / rate;
    const double dr = d - rint(d);
    return fabs(t_first) < sig_osc::kSineFastMaxT && fabs(t_last) < sig_osc::kSineFastMaxT && fabs(dr) <= 0.25 &&
           fabs(st) >= 1e-3;
}

struct SteadyVoice { bool ok; double na1, na2, scale, k2c, st, ct, hre, him, nd; M2 T, T0; };

// `cutoff`, `gain`: the voice's values (of one block, when they are read per block); `c_first`: the context of the launch's first block
template <bool GAIN>
__device__ __forceinline__ SteadyVoice steady_constants_of(const FusedArgs& a, int v, double cutoff, double gain, int c_first)
{
    using sig_biquad::Cx; using sig_biquad::cx_mul; using sig_biquad::cx_div;
    SteadyVoice r;
    Biquad q;
    r.ok = design_butter2(a.type, cutoff, a.rate, q);
    const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;                // b1 / b0
    const double a1 = q.a1, a2 = q.a2;
    const double d = a.hertz[(int64_t)v * a.hs] / a.rate;
    const double dr = d - rint(d);
    const double st = sin2pi(dr), ct = sin2pi(dr + 0.25), sh = sin2pi(0.5 * dr);
    const Cx z = {ct, -st};                                                    // e^{-j theta}
    const Cx z2 = cx_mul(z, z);
    const Cx H = cx_div({1.0 + s2 * z.re + z2.re, s2 * z.im + z2.im}, {1.0 + a1 * z.re + a2 * z2.re, a1 * z.im + a2 * z2.im});
    const Cx P = {H.re - 1.0, H.im};                                           // z0ss_{n-1} = Im(P u_n)
    const Cx Pe = cx_mul(P, {ct, st});
    const Cx Q = {Pe.re - s2 + a1 * H.re, Pe.im + a1 * H.im};                  // z1ss_{n-1} = Im(Q u_n)
    const double alpha = 2.0 * sh * sh / st, beta = 1.0 / st;                  // wr = alpha yss + beta dss, wi = yss
    auto make_T = [&](int c) {                                                 // T_c = -A^c Mss(c)
        const double cf = (double)c * dr;                                      // c theta in revolutions
        const Cx E = cx_div({sin2pi(cf + 0.25), -sin2pi(cf)}, H);              // e^{-j c theta} / H
        const Cx PE = cx_mul(P, E), QE = cx_mul(Q, E);
        const M2 Mss = {fma(PE.im, alpha, PE.re), PE.im * beta, fma(QE.im, alpha, QE.re), QE.im * beta};
        M2 Ac = {1.0, 0.0, 0.0, 1.0}, Ap = {-a1, 1.0, -a2, 0.0};               // A^c by squaring
        for (int e = c; e > 0; e >>= 1) {
            if (e & 1) Ac = m2_mul(Ac, Ap);
            Ap = m2_mul(Ap, Ap);
        }
        const M2 t = m2_mul(Ac, Mss);
        return M2{-t.a, -t.b, -t.c, -t.d};
    };
    r.T = make_T(a.ctx);
    r.T0 = (c_first == a.ctx) ? r.T : make_T(c_first);
    // Rows after a cold start until the homogeneous part is below kHomogeneousTol of the voice's full scale, for good:
    // in the coordinates S x in which A is a rotation times the pole radius rho = sqrt(a2) the state shrinks by exactly
    // rho per row, so |yh_n| <= cond(S) rho^n |x_0| with x_0 = minus the steady-state DF2T state, |x_0| <= sqrt(|P|^2 + |Q|^2)
    // (b0-normalised, hence the factor b0).  S^-1 = [[1, 0], [a1/2, d]], d = sqrt(a2 - a1^2/4) (eigenvector (1, a1 + lambda));
    // its condition number from the Frobenius norm and the determinant.  NaN or real poles: never (infinity).
    r.nd = __builtin_inf();
    {
        const double d2 = a2 - 0.25 * a1 * a1;
        if (r.ok && d2 > 0.0 && a2 > 0.0 && a2 < 1.0) {
            const double dd = sqrt(d2), f2 = 1.0 + 0.25 * a1 * a1 + d2;
            const double kappa = (f2 + sqrt(fmax(f2 * f2 - 4.0 * d2, 0.0))) / (2.0 * dd);
            const double amp = q.b0 * kappa * sqrt(P.re * P.re + P.im * P.im + Q.re * Q.re + Q.im * Q.im);
            const double rows = (amp > kHomogeneousTol) ? log(kHomogeneousTol / amp) / (0.5 * log(a2)) : 0.0;
            if (rows == rows) r.nd = ceil(rows) + 1.0;
        }
    }
    r.na1 = -a1; r.na2 = -a2;
    r.scale = GAIN ? q.b0 * gain : q.b0;
    r.k2c = 2.0 * ct; r.st = st; r.ct = ct; r.hre = H.re; r.him = H.im;
    return r;
}

template <bool GAIN>
__device__ __forceinline__ SteadyVoice steady_constants(const FusedArgs& a, int v)
{
    const int c0 = (int)((a.position < (int64_t)a.ctx) ? a.position : (int64_t)a.ctx);
    return steady_constants_of<GAIN>(a, v, a.cutoff[(int64_t)v * a.cs], GAIN ? a.gain[(int64_t)v * a.gs] : 1.0, c0);
}

// The closed form's constants for ONE block of a voice whose cutoff (and gain) is read per block (fused_steady_bus_kernel<..,
// CROWS>), in two parts.  What depends on the oscillator only is made once per span (OscPart: e^{j theta}, alpha / beta of the
// map from (yss, dss) to the complex amplitude, and e^{-j c theta} / N(e^{-j theta}) for the block context c, N the numerator of
// H); what depends on the block's filter per block (steady_block_constants): a light tangent (the argument is pi Wn / 2, Wn in
// (0, 1): no Payne-Hanek reduction), H = N / D and 1 / H = D / N from ONE reciprocal, the decay bound through v_log_f32 with
// two rows of margin -- three f64 divisions and ~350 instructions instead of nine and ~1500 (steady_constants_of), the same
// values to rounding (the same tests cover both).
struct OscPart { double ct, st, beta, enr, eni; };                            // beta = 1 / sin(theta); (enr, eni) = e^{-j c theta} / N
struct BlockVoice { bool ok; double na1, na2, scale, hre, him; M2 T; double nd; };

__device__ __forceinline__ OscPart steady_osc_part(int type, double hertz, double rate, int c)
{
    using sig_biquad::Cx; using sig_biquad::cx_mul; using sig_biquad::cx_div;
    const double d = hertz / rate, dr = d - rint(d);
    OscPart o;
    o.st = sin2pi(dr); o.ct = sin2pi(dr + 0.25);
    o.beta = 1.0 / o.st;
    const double s2 = (type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;
    const Cx z = {o.ct, -o.st}, z2 = cx_mul(z, z);
    const Cx N = {1.0 + s2 * z.re + z2.re, s2 * z.im + z2.im};
    const double cf = (double)c * dr;
    const Cx E = cx_div({sin2pi(cf + 0.25), -sin2pi(cf)}, N);
    o.enr = E.re; o.eni = E.im;
    return o;
}

__device__ __forceinline__ BlockVoice steady_block_constants(int type, double rate, double cutoff, double gain, const OscPart& o, int c)
{
    using sig_biquad::Cx; using sig_biquad::cx_mul;
    BlockVoice r;
    double wn = cutoff / (rate * 0.5);
    wn = (wn < 0.0) ? 0.0 : ((wn > 1.0) ? 1.0 : wn);
    r.ok = (wn > 0.0 && wn < 1.0);                                             // scipy raises otherwise (NaN too)
    // k = tan(pi Wn / 2) = sn / cs: with den = cs^2 + sqrt2 sn cs + sn^2 the coefficients need ONE division
    //   nrm = cs^2 / den,  k^2 nrm = sn^2 / den,  a1 = 2 (sn^2 - cs^2) / den,  a2 = (cs^2 - sqrt2 sn cs + sn^2) / den
    const double x = sig_biquad::kPi * wn * 0.5;
    const double xc = (1.5707963267948966 - x) + 6.123233995736766e-17;
    const double sn = sig_osc::sin_poly(x), cs = sig_osc::sin_poly(xc);
    const double sn2 = sn * sn, cs2 = cs * cs, sc2 = sig_biquad::kSqrt2 * sn * cs;
    const double iden = 1.0 / (cs2 + sc2 + sn2);
    const double b0 = (type == SIG_FILT_LOWPASS) ? sn2 * iden : cs2 * iden;
    double a1 = 2.0 * (sn2 - cs2) * iden, a2 = (cs2 - sc2 + sn2) * iden;
    if (!r.ok) { a1 = a2 = __builtin_nan(""); }
    const double s2 = (type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;
    const Cx z = {o.ct, -o.st};                                                // e^{-j theta}
    const Cx z2 = {o.ct * o.ct - o.st * o.st, -2.0 * o.ct * o.st};
    const Cx N = {1.0 + s2 * z.re + z2.re, s2 * z.im + z2.im}, D = {1.0 + a1 * z.re + a2 * z2.re, a1 * z.im + a2 * z2.im};
    const double id = 1.0 / (D.re * D.re + D.im * D.im);
    const Cx H = {(N.re * D.re + N.im * D.im) * id, (N.im * D.re - N.re * D.im) * id};
    const Cx P = {H.re - 1.0, H.im};
    const Cx Pe = cx_mul(P, {o.ct, o.st});
    const Cx Q = {Pe.re - s2 + a1 * H.re, Pe.im + a1 * H.im};
    const Cx E = cx_mul({o.enr, o.eni}, D);                                    // e^{-j c theta} / H = (e^{-j c theta} / N) D
    const Cx PE = cx_mul(P, E), QE = cx_mul(Q, E);
    const double alpha = (1.0 - o.ct) * o.beta;                               // 2 sin^2(theta / 2) / sin(theta)  (theta >= 1e-3: the difference keeps 10 digits)
    const M2 Mss = {fma(PE.im, alpha, PE.re), PE.im * o.beta, fma(QE.im, alpha, QE.re), QE.im * o.beta};
    M2 Ac;                                                                     // A^c by squaring
    const M2 A1 = {-a1, 1.0, -a2, 0.0};
    if (c == 100) {
        // the reference's context (fx.py:82-83).  Cayley-Hamilton: A^2 = tr A - det I (tr = -a1, det = a2), so every power is
        // p A + q I and squaring / multiplying by A act on the two scalars: (p, q)^2 = (p (p tr + 2 q), q^2 - p^2 det),
        // (p, q) A = (p tr + q, -p det).  100 = 1100100b: 6 squarings + 2 multiplications, ~40 operations instead of the 64 of
        // eight 2x2 products
        const double tr = -a1, det = a2;
        double pw = 1.0, qw = 0.0;                                             // A^1
        auto sq = [&]() { const double t = fma(pw, tr, qw + qw), pp = pw * pw; qw = fma(qw, qw, -(pp * det)); pw = pw * t; };
        auto ma = [&]() { const double t = fma(pw, tr, qw); qw = -(pw * det); pw = t; };
        sq(); ma();                                                            // A^3
        sq(); sq(); sq(); ma();                                                // A^25
        sq(); sq();                                                            // A^100
        Ac = M2{fma(pw, -a1, qw), pw, -a2 * pw, qw};
    } else {
        Ac = M2{1.0, 0.0, 0.0, 1.0};
        M2 Ap = A1;
        for (int e = c; e > 0; e >>= 1) {
            if (e & 1) Ac = m2_mul(Ac, Ap);
            Ap = m2_mul(Ap, Ap);
        }
    }
    const M2 t = m2_mul(Ac, Mss);
    r.T = M2{-t.a, -t.b, -t.c, -t.d};
    r.nd = __builtin_inf();
    {
        const double d2 = a2 - 0.25 * a1 * a1;
        if (r.ok && d2 > 0.0 && a2 > 0.0 && a2 < 1.0) {
            const float dd = __builtin_sqrtf((float)d2);
            const float f2 = (float)(1.0 + 0.25 * a1 * a1 + d2);
            const float kappa = (f2 + __builtin_sqrtf(fmaxf(f2 * f2 - 4.0f * (float)d2, 0.0f))) * __builtin_amdgcn_rcpf(2.0f * dd);
            const float amp = (float)b0 * kappa * __builtin_sqrtf((float)(P.re * P.re + P.im * P.im + Q.re * Q.re + Q.im * Q.im)) * 1.001f;
            const float rows = (amp > (float)kHomogeneousTol)
                ? (__log2f((float)kHomogeneousTol) - __log2f(amp)) * __builtin_amdgcn_rcpf(0.5f * __log2f((float)a2)) : 0.0f;
            if (rows == rows) r.nd = (double)(ceilf(rows * 1.0001f) + 3.0f);   // (float roundings, approximate reciprocals: margin)
        }
    }
    r.na1 = -a1; r.na2 = -a2; r.scale = b0 * gain; r.hre = H.re; r.him = H.im;
    return r;
}

__device__ __forceinline__ int wave_max_int(int x) {
#pragma unroll
    for (int d = 1; d < SIG_WAVE; d <<= 1) {
        const int y = __shfl_xor(x, d, SIG_WAVE);
        x = (y > x) ? y : x;
    }
    return __builtin_amdgcn_readfirstlane(x);
}

constexpr int kNeverDrops = 0x3fffffff;

// fused_mix.hip: the closed form into the MixMatrix sink (a.span, a.voice_tiles set by the caller)
int launch_steady_mix(const FusedArgs& a, bool gain, hipStream_t stream);

}  // namespace sig_fused
