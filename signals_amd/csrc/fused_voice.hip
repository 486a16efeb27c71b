// Fused voice chain for gfx950: Osc -> cold-start Butterworth biquad -> [x per-voice gain] -> f32 store or
// -> [pan x gain] -> bus partial sums, K blocks per launch.  Chosen by the batched engine when a LowPass/HighPass
// reads an oscillator nobody else consumes (and, optionally, feeds a Gain and a SumBus nobody else consumes): the
// oscillator samples never touch HBM, so the stage costs 4 B/voice-sample (the store) or ~0.13 B (the bus
// partials) instead of 4 + 8 (+ 8 + 4).
//
// Same design and phase arithmetic as the node kernels (sig_osc.h, sig_biquad.h; reference osc.py:26-62,
// fx.py:85-121, fx.py:51-52): f64 phase, f64 recurrence from zero state over [c context rows | block].  What
// differs from the per-node path, all of it below 1e-9 of the f64 reference and far inside the 1e-6 bar:
//   * the filter input is the oscillator's f64 sample, not its f32-rounded store;
//   * the recurrence runs on the b0-normalised filter  y' = y / b0  (b = [1, +-2, 1] for a Butterworth
//     low/high-pass), 4 fused multiply-adds per row instead of sosfilt's 8 separately rounded operations; b0 is
//     folded into the per-voice output weight (gain, pan), which is applied in f64 before the one f32 rounding;
//   * SPAN WALKER: a lane owns `span` consecutive blocks of its voices.  Block b+1 cold-starts from zero state
//     at its row -c, i.e. inside block b: its warm-up runs as a second recurrence on the oscillator sample the
//     lane has just computed for block b, and becomes the output recurrence at the block boundary.  Every
//     oscillator sample is computed once (not (N+c)/N times); the arithmetic of each chain is unchanged;
//   * Sine, while every |t| of the span is < 2^26 cycles and the voice advances by at most a quarter turn per
//     row (|hertz| <= rate/4 after aliasing): the oscillator is the two-term recurrence in difference (Reinsch)
//     form   x <- x + d;  d <- d - m x,   m = 4 sin^2(theta/2),  d_0 = 2 sin(theta/2) cos(phi_0 + theta/2),
//     seeded once per span from the reference's own t at the span's first row (sin by the f64 polynomial).
//     2 f64 ops per sample, no divide, no conversion, no v_sin_f32; rounding grows like rows x 1e-16 (the
//     difference form has no 1/theta amplification), i.e. ~1e-13 from sin(2 pi t) instead of v_sin_f32's 1e-7.
//     Otherwise (wave-uniform test) the exact per-row phase of sig_osc.h is used, as for the other waveforms.
//
// Mapping: one wave = 64*VPT consecutive voices x `span` consecutive blocks, lanes walk rows serially.  On the
// exact-phase path the per-row quotient n/rate (IEEE f64 divide) is computed 64 rows at a time, one row per
// lane, and broadcast with v_readlane.  f64-VALU-bound: Sine ~ 2 (osc) + 4 (N+c')/N (filter) + C (bus) f64
// ops per voice-sample.
#include <cstdlib>
#include <type_traits>

#include "sig_biquad.h"
#include "sig_bus_tile.h"
#include "sig_mix_tile.h"
#include "sig_osc.h"
#include "sig_steady.h"

// This file is compiled twice (compile time: 240 walker instantiations + the closed form took 4 min 50 s in one unit):
// fused_voice.hip itself -- the Sine kernels (walker, closed form) and the C ABI -- and fused_voice_b.hip, which includes it with
// SIG_FUSED_PART_B -- the walkers of Square, Sawtooth and Triangle behind four plain functions (sig_steady.h: part_b_*).
// Tuning builds (tools/build_variant.sh -DSIG_TUNE_SINE_ONLY) keep one unit with the Sine kernels and a stereo bus only.
#if defined(SIG_TUNE_SINE_ONLY)
#define SIG_FUSED_SPLIT 0
#define SIG_FUSED_HERE_SINE 1
#define SIG_FUSED_HERE_OTHERS 0
#elif defined(SIG_FUSED_PART_B)
#define SIG_FUSED_SPLIT 1
#define SIG_FUSED_HERE_SINE 0
#define SIG_FUSED_HERE_OTHERS 1
#else
#define SIG_FUSED_SPLIT 1
#define SIG_FUSED_HERE_SINE 1
#define SIG_FUSED_HERE_OTHERS 0
#endif

namespace {

using namespace sig_fused;

template <int VPT> struct OutVec;
template <> struct OutVec<1> { using type = float; };
template <> struct OutVec<2> { using type = float2; };
template <> struct OutVec<4> { using type = float4; };

__device__ __forceinline__ void put(float& v, const float (&y)[1]) { v = y[0]; }
__device__ __forceinline__ void put(float2& v, const float (&y)[2]) { v = make_float2(y[0], y[1]); }
__device__ __forceinline__ void put(float4& v, const float (&y)[4]) { v = make_float4(y[0], y[1], y[2], y[3]); }

// Register budget per voices-per-lane variant, as waves per SIMD the compiler must leave room for (0 = its own
// choice): the row groups below are straight-line code with many independent chains, which the scheduler would
// otherwise spread over every register it can get.  Values from tools/sweep_fused.sh.
#ifndef SIG_FUSED_OCC1
#define SIG_FUSED_OCC1 0
#endif
#ifndef SIG_FUSED_OCC2
#define SIG_FUSED_OCC2 0
#endif
#ifndef SIG_FUSED_OCC4
#define SIG_FUSED_OCC4 0
#endif
template <int VPT> struct Occ;
template <> struct Occ<1> { static constexpr int lo = SIG_FUSED_OCC1 ? SIG_FUSED_OCC1 : 1, hi = SIG_FUSED_OCC1 ? SIG_FUSED_OCC1 : 8; };
template <> struct Occ<2> { static constexpr int lo = SIG_FUSED_OCC2 ? SIG_FUSED_OCC2 : 1, hi = SIG_FUSED_OCC2 ? SIG_FUSED_OCC2 : 8; };
template <> struct Occ<4> { static constexpr int lo = SIG_FUSED_OCC4 ? SIG_FUSED_OCC4 : 1, hi = SIG_FUSED_OCC4 ? SIG_FUSED_OCC4 : 8; };

// Bus sums: sig_bus_tile.h (wave-private LDS tile, transposed reduction, per-tile f64 partials + fixed-order tile sum)
using sig_bus::kPairs;
using sig_bus::kTileStride;

__device__ __forceinline__ bool steady_wave(const FusedArgs& a, int v0, int vpt, int64_t p0, int nb);   // below

// C == 0: store (float)(weight * y) to a.out; C > 0: C bus channels into bus.partials; C == -1 (one voice per lane,
// voices a multiple of 64): the MixMatrix sink -- the wave's 64 voices are one matrix group, every 32 rows of
// float32 samples are staged in a wave-private LDS tile and multiplied by the 64 x 64 matrix on the matrix cores
// (sig_mix_tile.h: each float32 as three bfloat16, six bf16 MFMAs per k-block), then stored.  The per-voice rows
// never touch HBM.
// ROWS: cutoff and gain are read per block (the reference reads a control port once per block, at the block's position:
// chain/__init__.py:305-306 -- an LFO on a cutoff, a tremolo); the filter is then designed per block, the next block's
// warm-up chain with the next block's design.  GAIN is ignored (a null gain pointer means 1).
template <int KIND, int VPT, bool GAIN, int C, bool ROWS>
__device__ __forceinline__ void walk_wave(const FusedArgs& a, const BusArgs& bus, double* tile, int lane, int wave)
{
    constexpr bool BUS = C > 0, MIX = C < 0;
    constexpr int CC = BUS ? C : 1;
    constexpr int R = kPairs / CC;         // rows per flush
    static_assert(!MIX || VPT == 1, "the MixMatrix sink maps one matrix group to one wave");
    using Vec = typename OutVec<VPT>::type;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b_first = (item / a.voice_tiles) * a.span;
    if (b_first >= a.K) return;                                               // wave-uniform
    const int nb = (int)((a.K - b_first < (int64_t)a.span) ? a.K - b_first : (int64_t)a.span);
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live0 = v0 < a.voices;
    const int vc = live0 ? v0 : 0;

    const int64_t p0 = (a.pos_dev ? *a.pos_dev : a.position) + b_first * a.N;  // first frame of the span's first block
    const int c0 = (int)((p0 < (int64_t)a.ctx) ? p0 : (int64_t)a.ctx);
    const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;                // b1 / b0

    double na1[VPT], na2[VPT], z0[VPT], z1[VPT], wt[CC][VPT];
    double wna1[ROWS ? VPT : 1], wna2[ROWS ? VPT : 1], wwt[ROWS ? CC : 1][ROWS ? VPT : 1];   // ROWS: the next block's design and weights
    // the filter and the output weights of block b (ROWS: from parameter row b)
    auto design_block = [&](int64_t b, double* n1, double* n2, auto weights) {
        bool ok = true, any_live = false;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const bool live = v0 + i < a.voices;
            const int v = live ? v0 + i : vc;                                  // dead voices shadow a live one ...
            any_live |= live;
            const int64_t crow = (ROWS && a.cutoff_rows > 1) ? b * (int64_t)(a.cs ? a.voices : 1) : 0;
            const int64_t grow = (ROWS && a.gain_rows > 1) ? b * (int64_t)(a.gs ? a.voices : 1) : 0;
            Biquad q;
            ok &= design_butter2(a.type, a.cutoff[crow + (int64_t)v * a.cs], a.rate, q) || !live;
            n1[i] = -q.a1; n2[i] = -q.a2;
            const double scale = (ROWS ? a.gain != nullptr : GAIN) ? q.b0 * a.gain[grow + (int64_t)v * a.gs] : q.b0;
#pragma unroll
            for (int ch = 0; ch < CC; ++ch)                                    // ... with weight exactly 0 on the bus
                weights(ch, i, BUS ? (live ? (bus.pan ? bus.pan[ch * bus.pan_ld + v] * scale : scale) : 0.0) : scale);
        }
        if (!ok && any_live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
    };
    design_block(b_first, na1, na2, [&](int ch, int i, double w) { wt[ch][i] = w; });
#pragma unroll
    for (int i = 0; i < VPT; ++i) z0[i] = z1[i] = 0.0;

    // hertz / phase of the lane's voices (re-read where needed rather than kept live across the row loops)
    // ROWS with hertz_rows / phase_rows > 1: row `blk` of the launch (-1: the row in front of it, *_hist)
    const bool fm = ROWS && (a.hertz_hist || a.phase_hist);                    // (a one-block launch has one row, and still a row in front)
    auto load_hz_ph = [&](double (&hz)[VPT], double (&ph)[VPT], int64_t blk = 0) {
        const double* hp = a.hertz; const double* pp = a.phase;
        if (ROWS && a.hertz_hist) hp = (blk < 0) ? a.hertz_hist : a.hertz + (a.hertz_rows > 1 ? blk * (int64_t)(a.hs ? a.voices : 1) : 0);
        if (ROWS && a.phase_hist) pp = (blk < 0) ? a.phase_hist : a.phase + (a.phase_rows > 1 ? blk * (int64_t)(a.ps ? a.voices : 1) : 0);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = (v0 + i < a.voices) ? v0 + i : vc;
            hz[i] = hp[(int64_t)v * a.hs];
            ph[i] = pp ? pp[(int64_t)v * a.ps] : 0.0;
        }
    };

    // Sine as a two-term recurrence (see the header): seeded at the span's first row; under block-rate FM at every block's
    // first row (and for the span's warm-up rows) with that block's hertz / phase -- the recurrence then never runs longer
    // than a block
    bool fast = false;
    double sx[VPT], sdl[VPT], snm[VPT];                                        // x, d, -m
    int64_t seeded_blk = b_first - 1;                                          // FM: the block whose rows the recurrence was last seeded for
    // seeds for rows from frame n on, made with parameter row `blk`; returns whether every voice qualifies up to frame n_last
    auto seed_sine = [&](int64_t blk, int64_t n, int64_t n_last) {
        double hz[VPT], ph[VPT];
        load_hz_ph(hz, ph, blk);
        const double q_first = (double)n / a.rate;                             // osc.py:32
        const double q_last = (double)n_last / a.rate;
        bool small = true;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const double t_first = q_first * hz[i] + ph[i];
            const double t_last = q_last * hz[i] + ph[i];                      // t is monotonic in the row
            const double d = hz[i] / a.rate;                                   // revolutions per row
            const double dr = d - rint(d);
            small &= fabs(t_first) < sig_osc::kSineFastMaxT && fabs(t_last) < sig_osc::kSineFastMaxT && fabs(dr) <= 0.25;
            const double f0 = t_first - rint(t_first);                         // exact, |f0| <= 0.5
            const double sh = sin2pi(0.5 * dr);                                // sin(theta / 2)
            sx[i] = sin2pi(f0);
            sdl[i] = 2.0 * sh * sin2pi(f0 + 0.5 * dr + 0.25);                  // x_1 - x_0
            snm[i] = -4.0 * sh * sh;
        }
        return small;
    };
    if (KIND == SIG_OSC_SINE) {
        bool small = true;
        if (fm) {                                                              // every block of the span must qualify with its own row
            for (int bi = nb - 1; bi >= 0; --bi)
                small &= seed_sine(b_first + bi, p0 + (int64_t)bi * a.N, p0 + (int64_t)(bi + 1) * a.N - 1);
            small &= seed_sine(b_first - 1, p0 - c0, p0 - 1 >= p0 - c0 ? p0 - 1 : p0 - c0);     // (last: the warm-up rows come first)
        } else {
            small = seed_sine(0, p0 - c0, p0 + (int64_t)nb * a.N - 1);
        }
        fast = __all(small);
    }

    float* dst = (BUS || MIX) ? nullptr : a.out + vc;                          // row index = frame - position
    double* dstp = BUS ? bus.partials + (int64_t)vt * bus.rows * C : nullptr;  // [tile][row][c]
    sig_bus::PipelinedTile<CC> stage(tile, lane, dstp, b_first * a.N);
    int64_t n_cur = p0 - c0;                                                   // absolute frame of the next row

    // MixMatrix sink: rows staged as float32, 32 at a time through the matrix cores (sig_mix_tile.h)
    std::conditional_t<MIX, sig_mix::Sink, int> sink{};
    if constexpr (MIX) sink.init(a.mix, reinterpret_cast<float*>(tile), a.out + (int64_t)vt * 64, a.out_ld, b_first * a.N, lane);

    // one row of the lane's recurrences: y = output of the current block's chain; WARM rows also advance the
    // next block's warm-up chain on the same input
    auto chains = [&](const double (&x)[VPT], double (&y)[VPT], double (&w0)[VPT], double (&w1)[VPT], auto warm_tag) {
        constexpr bool WARM = decltype(warm_tag)::value;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            y[i] = x[i] + z0[i];                                               // DF2T of [1, s2, 1] / [1, a1, a2]
            z0[i] = fma(na1[i], y[i], fma(s2, x[i], z1[i]));
            z1[i] = fma(na2[i], y[i], x[i]);
            if (WARM) {
                const double yw = x[i] + w0[i];
                w0[i] = fma(ROWS ? wna1[i] : na1[i], yw, fma(s2, x[i], w1[i]));
                w1[i] = fma(ROWS ? wna2[i] : na2[i], yw, x[i]);
            }
        }
    };
    auto to_tile = [&](const double (&y)[VPT], double* where, int stride = kTileStride) {
#pragma unroll
        for (int ch = 0; ch < CC; ++ch) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < VPT; ++i) acc = fma(wt[ch][i], y[i], acc);
            where[ch * stride] = acc;
        }
    };
    sig_bus::FoldedGroup<CC> folded(tile, lane, dstp);                         // whole groups of R rows: sums folded in registers
    auto to_out = [&](const double (&y)[VPT], int64_t out_row) {
        if constexpr (MIX) {                                                   // rows arrive in order: stage, multiply every 32
            sink.stage((float)(y[0] * wt[0][0]));
            return;
        }
        float y32[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) y32[i] = (float)(y[i] * wt[0][i]);
        if (live0) {
            Vec o; put(o, y32);
            *reinterpret_cast<Vec*>(dst + out_row * a.out_ld) = o;
        }
    };

    // `count` consecutive rows from n_cur on; OUT rows go to output rows out_row, out_row + 1, ...
    auto walk = [&](int count, int64_t out_row, double (&w0)[VPT], double (&w1)[VPT], auto out_tag, auto warm_tag, auto fast_tag, auto pair_tag, int64_t blk) {
        constexpr bool OUT = decltype(out_tag)::value, FAST = decltype(fast_tag)::value;
        double hz[VPT], ph[VPT], q_lane = 0.0;
        int64_t qbase = 0;
        bool q_valid = false;
        if (!FAST) load_hz_ph(hz, ph, blk);
        if constexpr (FAST && ROWS) {
            if (fm && blk != seeded_blk) {                                     // wave-uniform: a new block's hertz / phase
                seed_sine(blk, n_cur, n_cur);
                seeded_blk = blk;
            }
        }
        // second oscillator of a Mix / RingMod source (ROWS kernels only; its waveform is a wave-uniform run-time switch)
        constexpr bool paired = ROWS && decltype(pair_tag)::value;               // (a compile-time copy of the row code: a run-time test per row cut the groups into pieces)
        double hz2[ROWS ? VPT : 1], ph2[ROWS ? VPT : 1], mx[ROWS ? VPT : 1];
        if constexpr (ROWS) {
            if (paired) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const int v = (v0 + i < a.voices) ? v0 + i : vc;
                    hz2[i] = a.hertz2[(int64_t)v * a.hs2];
                    ph2[i] = a.phase2 ? a.phase2[(int64_t)v * a.ps2] : 0.0;
                    mx[i] = a.mixrow ? a.mixrow[(int64_t)v * a.ms] : 0.0;
                }
            }
        }
        // exact phase: n/rate (IEEE divide) for 64 rows at a time, one row per lane (osc.py:32)
        auto ensure = [&](int rows) {
            if ((!FAST || paired) && (!q_valid || n_cur + rows > qbase + SIG_WAVE)) {     // wave-uniform
                qbase = n_cur;
                q_lane = (double)(qbase + lane) / a.rate;
                q_valid = true;
            }
        };
        auto gen = [&](double (&x)[VPT], int k) {                              // sample of row n_cur + k
            if (FAST) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    x[i] = sx[i];
                    sx[i] = x[i] + sdl[i];
                    sdl[i] = fma(snm[i], sx[i], sdl[i]);
                }
            } else {
                const double t_s = sig_readlane_f64(q_lane, (int)(n_cur - qbase) + k);
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const double t = t_s * hz[i] + ph[i];
                    x[i] = (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t) : sig_osc::osc_wave_fused<KIND>(t);
                }
            }
            if constexpr (ROWS) {
                if (paired) {                                                  // x = mix * A + (1 - mix) * B (fx.py:38-40) or A * B (fx.py:45-46)
                    const double t_s = sig_readlane_f64(q_lane, (int)(n_cur - qbase) + k);
#pragma unroll
                    for (int i = 0; i < VPT; ++i) {
                        const double t = t_s * hz2[i] + ph2[i];
                        double b;
                        switch (a.kind2) {                                     // wave-uniform
                            case SIG_OSC_SINE: b = (double)sig_osc::osc_sine_f32(t); break;
                            case SIG_OSC_SQUARE: b = sig_osc::osc_square_fract(t); break;
                            case SIG_OSC_SAWTOOTH: b = sig_osc::osc_sawtooth_fract(t); break;
                            default: b = sig_osc::osc_triangle_fract(t); break;
                        }
                        x[i] = (a.pair_op == 1) ? mx[i] * x[i] + (1.0 - mx[i]) * b : x[i] * b;
                    }
                }
            }
        };
        int done = 0;
        // the exact-phase Sine path is the rare one (positions beyond 2^26 cycles): rolled loops, so that its
        // register needs do not set the kernel's budget
        constexpr bool GROUPED = FAST || KIND != SIG_OSC_SINE || ROWS;           // (ROWS: under block-rate FM the exact phase IS the Sine path)
        if (!OUT || !BUS) {
            constexpr int U = GROUPED ? 4 : 1;                                 // rows per unrolled step
            for (; done + U <= count; done += U) {
                ensure(U);
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    double x[VPT], y[VPT];
                    gen(x, k);
                    chains(x, y, w0, w1, warm_tag);
                    if (OUT) to_out(y, out_row + done + k);
                }
                n_cur += U;
            }
            for (; done < count; ++done) {
                double x[VPT], y[VPT];
                ensure(1);
                gen(x, 0);
                chains(x, y, w0, w1, warm_tag);
                if (OUT) to_out(y, out_row + done);
                ++n_cur;
            }
            return;
        }
        auto single = [&]() {
            double x[VPT], y[VPT];
            ensure(1);
            gen(x, 0);
            chains(x, y, w0, w1, warm_tag);
            to_tile(y, stage.slot);
            ++n_cur; ++done;
            stage.advance();
        };
        while (stage.staged != 0 && done < count) single();                    // until the tile is empty
        double pend[4];
        int64_t pend_row = 0;
        bool have = false;
        for (; GROUPED && done + R <= count; done += R) {                      // whole groups: sums in registers, folded across lanes
            ensure(R);                                                         // (sig_bus::FoldedGroup), the flush one group behind
            double acc[kPairs];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double x[VPT], y[VPT];
                gen(x, k);
                chains(x, y, w0, w1, warm_tag);
                to_tile(y, acc + k * CC, 1);
#pragma unroll
                for (int g = 0; g < kPairs / 4; ++g)
                    if (4 * g + 3 < (k + 1) * CC && 4 * g + 3 >= k * CC)
                        folded.fold4(g, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                if (k + 1 == R / 2 && have) folded.finish(pend, pend_row, R);
            }
            n_cur += R;
            folded.issue(pend);
            pend_row = stage.first; stage.first += R; have = true;
        }
        if (have) folded.finish(pend, pend_row, R);
        while (done < count) single();
    };
    // `blk`: the block whose hertz / phase rows these rows were made with (only read under block-rate FM)
    auto walk_any = [&](int count, int64_t out_row, double (&w0)[VPT], double (&w1)[VPT], auto out_tag, auto warm_tag, int64_t blk) {
        if constexpr (ROWS) {
            if (a.pair_op != 0) {                                              // wave-uniform
                if (KIND == SIG_OSC_SINE && fast) walk(count, out_row, w0, w1, out_tag, warm_tag, std::true_type{}, std::true_type{}, blk);
                else walk(count, out_row, w0, w1, out_tag, warm_tag, std::false_type{}, std::true_type{}, blk);
                return;
            }
        }
        if (KIND == SIG_OSC_SINE && fast) walk(count, out_row, w0, w1, out_tag, warm_tag, std::true_type{}, std::false_type{}, blk);
        else walk(count, out_row, w0, w1, out_tag, warm_tag, std::false_type{}, std::false_type{}, blk);
    };

    walk_any(c0, 0, z0, z1, std::false_type{}, std::false_type{}, b_first - 1);   // warm-up of the span's first block: the previous block's samples
    for (int bi = 0; bi < nb; ++bi) {
        const int64_t orow = (b_first + bi) * a.N;
        const int tail = (bi + 1 < nb) ? a.ctx : 0;                            // rows that also warm the next block up (N >= ctx)
        walk_any(a.N - tail, orow, z0, z1, std::true_type{}, std::false_type{}, b_first + bi);
        if (tail) {
            double w0[VPT], w1[VPT];                                           // the next block's chain, from zero state
#pragma unroll
            for (int i = 0; i < VPT; ++i) { w0[i] = 0.0; w1[i] = 0.0; }
            if constexpr (ROWS) {
                if (a.cutoff_rows > 1 || a.gain_rows > 1) {                    // (block-rate FM alone: one design for the launch)
                    design_block(b_first + bi + 1, wna1, wna2, [&](int ch, int i, double w) { wwt[ch][i] = w; });
                } else {
#pragma unroll
                    for (int i = 0; i < VPT; ++i) {
                        wna1[i] = na1[i]; wna2[i] = na2[i];
#pragma unroll
                        for (int ch = 0; ch < CC; ++ch) wwt[ch][i] = wt[ch][i];
                    }
                }
            }
            walk_any(tail, orow + a.N - tail, w0, w1, std::true_type{}, std::true_type{}, b_first + bi);
#pragma unroll
            for (int i = 0; i < VPT; ++i) { z0[i] = w0[i]; z1[i] = w1[i]; }
            if constexpr (ROWS) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    na1[i] = wna1[i]; na2[i] = wna2[i];
#pragma unroll
                    for (int ch = 0; ch < CC; ++ch) wt[ch][i] = wwt[ch][i];
                }
            }
        }
    }
    if (BUS && stage.staged) stage.now();
    if constexpr (MIX) sink.finish();
}

template <int KIND, int VPT, bool GAIN, int C, bool ROWS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Occ<VPT>::lo, Occ<VPT>::hi)))
void fused_walk_kernel(FusedArgs a, BusArgs bus)
{
    constexpr bool BUS = C > 0, MIX = C < 0;
    __shared__ __attribute__((aligned(16))) double lds[(BUS || MIX) ? 4 : 1][BUS ? kPairs * kTileStride : (MIX ? sig_mix::kTileRows * sig_mix::kLdsStride / 2 : 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform by construction: block / tile indices in SGPRs
    walk_wave<KIND, VPT, GAIN, C, ROWS>(a, bus, lds[(BUS || MIX) ? wave : 0], lane, wave);
    if constexpr (BUS) {
        if (bus.out) sig_bus::sum_tiles_in_workgroup<C>(bus.partials, a.voice_tiles, bus.rows, a.span, a.K, a.N, bus.out, bus.out_ld, lane, wave);
    }
}

// ---------------------------------------------------------------------------------------------------
// Sine through a cold-started LTI filter in closed form ("steady" kernel).  For x_n = sin(phi_n), phi_n = phi_0 +
// n theta, the filter's response from zero state at row r0 is the steady-state sinusoid plus a decaying
// homogeneous solution:
//     y_n = yss_n + yh_n,     yss_n = Im(H(e^{j theta}) e^{j phi_n}),     (z0h, z1h)_n = A (z0h, z1h)_{n-1},  yh_n = z0h_{n-1}
// with the homogeneous state at r0 - 1 equal to minus the steady-state DF2T state there (so the total state is
// zero, fx.py:104's sosfilt start).  Every ingredient is linear in (yss_n, yss_{n+1} - yss_n), so the homogeneous
// state at a block's first row p = r0 + c is one per-voice 2x2 matrix applied to the steady-state oscillator's
// state at p:   (z0h, z1h)_{p-1} = T_c (yss_p, dss_p),   T_c = -A^c Mss(c)   -- no warm-up rows at all.
// Per stored sample: 1 (yss: the two-term recurrence y_{n+1} = 2 cos(theta) y_n - y_{n-1}, one fma, re-seeded from the
// reference's own t at every span start; its error grows like rows * 2e-16 / sin(theta), < 1e-9 for the voices this
// kernel accepts) + C (bus) + C/VPT (flush), and -- only while the homogeneous part of a voice is still above 1e-11
// of that voice's full scale -- 2 (homogeneous recurrence) + 1 (sum).  The homogeneous part decays like the pole
// radius^n and has already decayed over the c warm-up rows when the block starts: steady_prep_kernel bounds it
// rigorously per voice (rows from the cold start until it is below the tolerance, SC_ND), the kernel takes the wave
// maximum per voice SLOT (the i-th voice of every lane) and runs row groups in variants with only the first M slots
// "live".  A caller that orders its voices so that a slot holds neighbours in cutoff (the engine sorts by cutoff,
// slot-major) gets most row groups at M = 0; any order is correct.  Mathematically identical to the walker; rounding
// differs at 1e-10.  A wave takes this path when every voice of it passes steady_voice_ok(); the rare other waves run
// steady_fallback_span inside the same launch.
// Per-voice constants, computed once per launch by steady_prep_kernel into the tail of the workspace (SoA, kSteadyConsts
// rows of `voices` doubles): the filter, the oscillator step, H(e^{j theta}), T_c for c = ctx and for the launch's first
// block (c = min(ctx, position)), and the decay bound.
enum { SC_NA1, SC_NA2, SC_SCALE, SC_K2C, SC_ST, SC_CT, SC_HRE, SC_HIM, SC_ND, SC_T, SC_T0 = SC_T + 4, kSteadyConsts = SC_T0 + 4 };

// does the steady kernel take the wave of voices [v0, v0 + vpt) x 64 lanes for the span starting at frame p0?
__device__ __forceinline__ bool steady_wave(const FusedArgs& a, int v0, int vpt, int64_t p0, int nb) {
    const double q_first = (double)p0 / a.rate, q_last = (double)(p0 + (int64_t)nb * a.N - 1) / a.rate;
    bool ok = true;
    for (int i = 0; i < vpt; ++i) {
        const int v = (v0 + i < a.voices) ? v0 + i : ((v0 < a.voices) ? v0 : 0);
        ok &= steady_voice_ok(a.hertz[(int64_t)v * a.hs], a.phase ? a.phase[(int64_t)v * a.ps] : 0.0, a.rate,
                              a.steady_consts[(int64_t)SC_ST * a.voices + v], q_first, q_last);
    }
    return __all(ok);
}

// the closed form's per-voice constants (see the enum above), derived from the voice's parameters
template <bool GAIN>
__global__ __launch_bounds__(256) void steady_prep_kernel(FusedArgs a, double* __restrict__ consts)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.voices) return;
    const SteadyVoice c = steady_constants<GAIN>(a, v);
    if (!c.ok && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
    auto put = [&](int k, double x) { consts[(int64_t)k * a.voices + v] = x; };
    put(SC_NA1, c.na1); put(SC_NA2, c.na2); put(SC_SCALE, c.scale);
    put(SC_K2C, c.k2c); put(SC_ST, c.st); put(SC_CT, c.ct);
    put(SC_HRE, c.hre); put(SC_HIM, c.him); put(SC_ND, c.nd);
    put(SC_T + 0, c.T.a); put(SC_T + 1, c.T.b); put(SC_T + 2, c.T.c); put(SC_T + 3, c.T.d);
    put(SC_T0 + 0, c.T0.a); put(SC_T0 + 1, c.T0.b); put(SC_T0 + 2, c.T0.c); put(SC_T0 + 3, c.T0.d);
}

// The rare waves the closed form does not take (a voice below ~8 Hz, above rate/4 or past 2^26 cycles), done inside the
// same launch by the plainest possible code: every block on its own, exact per-row phase (one IEEE divide per row),
// the b0-normalised recurrence from zero state over [c context rows | block], rows staged one at a time.  Rolled
// loops and no row groups, so that this path does not set the kernel's register budget; ~4x slower per voice-sample
// than the closed form, and it saves launching the span walker over every wave just to find nothing to do.
template <int VPT, int C>
__device__ __forceinline__ void steady_fallback_span(const FusedArgs& a, const BusArgs& bus, double* tile, int lane, int vt,
                                                  int64_t b_first, int nb, int v0)
{
    const int vc = (v0 < a.voices) ? v0 : 0;
    const double* sc = a.steady_consts;
    const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;
    sig_bus::PipelinedTile<C> stage(tile, lane, bus.partials + (int64_t)vt * bus.rows * C, b_first * a.N);
    for (int i = 0; i < VPT; ++i)                                              // a rejected design is NaN in the constants
        if (v0 + i < a.voices && a.status && sc[(int64_t)SC_NA1 * a.voices + v0 + i] != sc[(int64_t)SC_NA1 * a.voices + v0 + i])
            atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
#pragma unroll 1
    for (int bi = 0; bi < nb; ++bi) {
        const int64_t p_b = a.position + (b_first + bi) * a.N;
        const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);
        double z0[VPT], z1[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) z0[i] = z1[i] = 0.0;
#pragma unroll 1
        for (int r = -c; r < a.N; ++r) {
            const double q = (double)(p_b + r) / a.rate;                       // osc.py:32
            double acc[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = 0.0;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const bool live = v0 + i < a.voices;
                const int v = live ? v0 + i : vc;
                const double t = q * a.hertz[(int64_t)v * a.hs] + (a.phase ? a.phase[(int64_t)v * a.ps] : 0.0);
                const double x = (double)sig_osc::osc_sine_f32(t);
                double na1 = sc[(int64_t)SC_NA1 * a.voices + v], na2 = sc[(int64_t)SC_NA2 * a.voices + v];
                double scale = live ? sc[(int64_t)SC_SCALE * a.voices + v] : 0.0;
                if (a.cutoff_rows > 1) {                                       // per-block cutoff rows: the block's own design (wave-uniform branch)
                    Biquad qd;
                    design_butter2(a.type, a.cutoff[(b_first + bi) * (int64_t)(a.cs ? a.voices : 1) + (int64_t)v * a.cs], a.rate, qd);
                    na1 = -qd.a1; na2 = -qd.a2;
                    scale = live ? qd.b0 : 0.0;
                    if (a.gain && a.gain_rows == 1) scale *= a.gain[(int64_t)v * a.gs];
                }
                const double y = x + z0[i];
                z0[i] = fma(na1, y, fma(s2, x, z1[i]));
                z1[i] = fma(na2, y, x);
                if (a.gain_rows > 1) scale *= a.gain[(b_first + bi) * (int64_t)(a.gs ? a.voices : 1) + (int64_t)v * a.gs];   // per-block gain rows (the constants then hold b0 only)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[ch] = fma(bus.pan ? bus.pan[ch * bus.pan_ld + v] * scale : scale, y, acc[ch]);
            }
            if (r >= 0) {                                                      // wave-uniform
#pragma unroll
                for (int ch = 0; ch < C; ++ch) stage.slot[ch * kTileStride] = acc[ch];
                stage.advance();
            }
        }
    }
    if (stage.staged) stage.now();
}

// wave-wide maximum of a non-negative int, the same value in every lane

// the row-group variants of fused_steady_bus_kernel: "the first M of the lane's VPT voice slots still carry their
// homogeneous part", largest first
template <int VPT> struct SteadyVariants {
    static constexpr int count = (VPT >= 8) ? 7 : (VPT == 4) ? 4 : (VPT == 2) ? 3 : 2;
    static constexpr int at(int k) {                   // (16 voices per lane: registers for 8 live slots, like 8 per lane)
        constexpr int v8[7] = {8, 6, 4, 3, 2, 1, 0}, v4[4] = {4, 2, 1, 0}, v2[3] = {2, 1, 0}, v1[2] = {1, 0};
        return (VPT >= 8) ? v8[k] : (VPT == 4) ? v4[k] : (VPT == 2) ? v2[k] : v1[k];
    }
};

// Register budget of the closed-form kernel, as waves per SIMD the compiler must leave room for: its row groups are
// straight-line code with many independent chains, which the scheduler otherwise spreads over every register it can
// get (8 voices per lane: 417 registers and scratch, for 210 live values).
#ifndef SIG_STEADY_OCC8
#define SIG_STEADY_OCC8 1
#endif
#ifndef SIG_STEADY_AUTO16
#define SIG_STEADY_AUTO16 0              // 16 voices per lane (live slots capped at 8): 512 registers, AGPR copies and scratch -- 266 us vs 205 with 8: tuning hook only
#endif
#ifndef SIG_STEADY_OCC16
#define SIG_STEADY_OCC16 1
#endif
template <int VPT> struct SteadyOcc { static constexpr int waves = (VPT == 16) ? SIG_STEADY_OCC16 : (VPT == 8) ? SIG_STEADY_OCC8 : 2; };

// GROWS: the gain is read per block (a tremolo: sig_fused_voice_bus_rows with rows for the gain only); the constants then hold
// b0 alone and the bus weights are rebuilt at every block's first row
// CROWS: the cutoff (and possibly the gain) is read per block: the filter, its response H, T_c and the decay bound are derived
// at every block's first row from that block's rows, the steady-state recurrence is re-seeded at every block's first row with that block's H
// (the oscillator itself runs on: the phase of the row is recomputed from the reference's own t, two sines per voice and block)
template <int VPT, int C, bool GROWS, bool CROWS = false>
__device__ __forceinline__ void steady_bus_wave(const FusedArgs& a, const BusArgs& bus, double* tile, int lane, int wave, double* osc_store = nullptr)
{
    constexpr bool OSC_LDS = CROWS && VPT >= 8;                                // the per-span oscillator parts in LDS instead of registers (osc_store)
    constexpr int R = kPairs / C;          // rows per flush
    constexpr int LC = SteadyVariants<VPT>::at(0);     // voice slots that can carry a homogeneous part (all of them up to 8 per lane)
    static_assert(R % 2 == 0, "the two-term recurrence rotates two registers per voice: row groups are even");
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b_first = (item / a.voice_tiles) * a.span;
    if (b_first >= a.K) return;                                               // wave-uniform
    const int nb = (int)((a.K - b_first < (int64_t)a.span) ? a.K - b_first : (int64_t)a.span);
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const int vc = (v0 < a.voices) ? v0 : 0;
    const int64_t p0 = a.position + b_first * a.N;
    if (!steady_wave(a, v0, VPT, p0, nb)) {                                   // wave-uniform, rare
        steady_fallback_span<VPT, C>(a, bus, tile, lane, vt, b_first, nb, v0);
        return;
    }
    const double* sc = a.steady_consts;

    // per voice: the filter (na1, na2), the oscillator step k = 2 cos(theta), the steady-state output at rows p0 - 1
    // and p0 (ya, yb), the bus weights; per voice SLOT (wave-uniform): rows from a cold start after which the
    // homogeneous part is dropped
    double na1[LC], na2[LC], k2c[VPT], ya[VPT], yb[VPT], wt[C][VPT];
    int nd_total[VPT];
    const double q_first = (double)p0 / a.rate;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool live = v0 + i < a.voices;
        const int v = live ? v0 + i : vc;                                      // dead voices shadow a live one ...
        auto cst = [&](int k) { return sc[(int64_t)k * a.voices + v]; };
        const double na1_i = cst(SC_NA1);
        if (i < LC) { na1[i < LC ? i : 0] = na1_i; na2[i < LC ? i : 0] = cst(SC_NA2); }
        k2c[i] = cst(SC_K2C);
        // the design is checked where the constants are made (steady_prep_kernel); a caller that keeps them across
        // calls skips that launch, so every launch that USES a rejected design (NaN coefficients) reports it again
        if (live && na1_i != na1_i && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
        const double scale = cst(SC_SCALE);
#pragma unroll
        for (int ch = 0; ch < C; ++ch)                                         // ... with weight exactly 0 on the bus
            wt[ch][i] = live ? (bus.pan ? bus.pan[ch * bus.pan_ld + v] * scale : scale) : 0.0;
        // steady-state oscillator at the span's first row: w = H e^{j phi}, yss_p0 = Im w, yss_{p0-1} = Im(w e^{-j theta})
        const double hz = a.hertz[(int64_t)v * a.hs], ph = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
        const double t_first = q_first * hz + ph;                              // osc.py:32
        const double f0 = t_first - rint(t_first);                             // exact, |f0| <= 0.5
        const double ur = sin2pi(f0 + 0.25), ui = sin2pi(f0);
        const double hre = cst(SC_HRE), him = cst(SC_HIM);
        const double wr = fma(hre, ur, -(him * ui)), wi = fma(hre, ui, him * ur);
        yb[i] = wi;
        ya[i] = fma(wi, cst(SC_CT), -(wr * cst(SC_ST)));
        const double nd = cst(SC_ND);
        nd_total[i] = (live && nd < (double)kNeverDrops) ? (int)nd : (live ? kNeverDrops : 0);   // NaN: never
    }
    // ... per voice SLOT: the wave maximum, the VPT butterflies side by side (one after the other their cross-lane round trips
    // were a tenth of a one-block span)
#pragma unroll
    for (int d = 1; d < SIG_WAVE; d <<= 1) {
        int other[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) other[i] = __shfl_xor(nd_total[i], d, SIG_WAVE);
#pragma unroll
        for (int i = 0; i < VPT; ++i) nd_total[i] = (other[i] > nd_total[i]) ? other[i] : nd_total[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) nd_total[i] = __builtin_amdgcn_readfirstlane(nd_total[i]);

    double* dstp = bus.partials + (int64_t)vt * bus.rows * C;                  // [tile][row][c]
    sig_bus::PipelinedTile<C> stage(tile, lane, dstp, b_first * a.N);

    // 16 voices per lane: only the first 8 slots have registers for a homogeneous part; the caller vouched (consts_ready
    // bit 1) that the others have none at any block start of this launch -- checked here, a wave it does not hold for takes
    // the plain fallback (correct, slow)
    if constexpr (LC < VPT) {
        const int c_min = (b_first == 0 && a.position < (int64_t)a.ctx) ? (int)a.position : a.ctx;
        bool capped = true;
#pragma unroll
        for (int i = LC; i < VPT; ++i) capped &= nd_total[i] <= c_min;
        if (!capped) {                                                         // wave-uniform
            steady_fallback_span<VPT, C>(a, bus, tile, lane, vt, b_first, nb, v0);
            return;
        }
    }
    [[maybe_unused]] OscPart osc[(CROWS && !OSC_LDS) ? VPT : 1];               // CROWS: what the per-block constants need of the oscillator (once per span)
    auto osc_slot = [&](int field, int i) -> double& { return osc_store[((size_t)field * VPT + i) * SIG_WAVE + lane]; };
    if constexpr (CROWS) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = (v0 + i < a.voices) ? v0 + i : vc;
            const OscPart made = steady_osc_part(a.type, a.hertz[(int64_t)v * a.hs], a.rate, a.ctx);
            if constexpr (OSC_LDS) {
                osc_slot(0, i) = made.ct; osc_slot(1, i) = made.st; osc_slot(2, i) = made.beta; osc_slot(3, i) = made.enr; osc_slot(4, i) = made.eni;
            } else {
                osc[i] = made;
            }
        }
    }
    double z0h[LC], z1h[LC];
    // One row of every voice; the first M slots carry their homogeneous part, the others have dropped it.  The row's C
    // sums over the lane's voices go to `sums` (registers of the group being built, or the LDS slot of the single-row
    // form).  The two-term recurrence runs IN PLACE on two registers per voice: on an even row yb is the sample and ya
    // becomes the one after next, on an odd row the roles are swapped -- no register rotation for the compiler to undo.
    auto row = [&](double* sums, int sums_stride, auto m_tag, auto odd_tag) {
        constexpr int M = decltype(m_tag)::value;
        constexpr bool ODD = decltype(odd_tag)::value;
        double y[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const double ys = ODD ? ya[i] : yb[i];
            if (ODD) yb[i] = fma(k2c[i], ya[i], -yb[i]);
            else ya[i] = fma(k2c[i], yb[i], -ya[i]);
            if (i < M) {
                const int j = i < LC ? i : 0;                                  // (M <= LC: always i itself)
                y[i] = ys + z0h[j];
                const double yh = z0h[j];
                z0h[j] = fma(na1[j], yh, z1h[j]);
                z1h[j] = na2[j] * yh;
            } else {
                y[i] = ys;
            }
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < VPT; ++i) acc = fma(wt[ch][i], y[i], acc);
            sums[ch * sums_stride] = acc;
        }
    };
    // One group of R rows: their kPairs sums stay in registers, are folded across lanes (sig_bus::FoldedGroup) and the
    // LDS reads of the last step are issued at once; they are consumed half-way through the NEXT group, when they (and
    // the stores in front of them: a wave's LDS operations complete in order) have long retired.
    sig_bus::FoldedGroup<C> folded(tile, lane, dstp);
    double pend[4];
    int64_t pend_row = 0;
    bool have = false;
    // Two consecutive rows at once (an even one and an odd one, see `row`), so that their 2 C bus sums are FOUR
    // independent accumulation chains: a lone wave issues an f64 instruction every 4 cycles but a dependent one only
    // every ~10, and two interleaved chains (one row's two channels) ran at 60 % of the issue rate.
    auto rows2 = [&](double* sums, auto m_tag) {
        constexpr int M = decltype(m_tag)::value;
        double y0[VPT], y1[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            y0[i] = yb[i];
            ya[i] = fma(k2c[i], yb[i], -ya[i]);
            y1[i] = ya[i];
            yb[i] = fma(k2c[i], ya[i], -yb[i]);
            if (i < M) {
                const int j = i < LC ? i : 0;                                  // (M <= LC: always i itself)
                const double h0 = z0h[j];
                y0[i] += h0;
                const double h1 = fma(na1[j], h0, z1h[j]);
                y1[i] += h1;
                z0h[j] = fma(na1[j], h1, na2[j] * h0);
                z1h[j] = na2[j] * h1;
            }
        }
        double acc0[C], acc1[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) { acc0[ch] = 0.0; acc1[ch] = 0.0; }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                acc0[ch] = fma(wt[ch][i], y0[i], acc0[ch]);
                acc1[ch] = fma(wt[ch][i], y1[i], acc1[ch]);
            }
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) { sums[ch] = acc0[ch]; sums[C + ch] = acc1[ch]; }
    };
    auto group = [&](auto m_tag) {
        double acc[kPairs];
#pragma unroll
        for (int k = 0; k < R; k += 2) {
            rows2(acc + k * C, m_tag);
#pragma unroll
            for (int q = 0; q < kPairs / 4; ++q)                               // every four sums are folded as soon as they exist
                if (4 * q + 3 < (k + 2) * C && 4 * q + 3 >= k * C)
                    folded.fold4(q, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (k + 2 == R / 2 && have) folded.finish(pend, pend_row, R);
        }
        folded.issue(pend);
        pend_row = stage.first; stage.first += R; have = true;
    };
    // Row groups come in variants "the first M slots live", M from kVariants; within a block the number of live slots only
    // falls, so a block is a sequence of PHASES, one plain loop per variant (a switch per group cost 20-30 %: the
    // variants' registers had to be shuffled into one layout at every merge).  Phase of variant M runs until every slot
    // >= the next smaller variant has dropped.
    using Variants = SteadyVariants<VPT>;

    for (int bi = 0; bi < nb; ++bi) {
        if constexpr (GROWS && !CROWS) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const bool live = v0 + i < a.voices;
                const int v = live ? v0 + i : vc;
                const double scale = sc[(int64_t)SC_SCALE * a.voices + v] * a.gain[(b_first + bi) * (int64_t)(a.gs ? a.voices : 1) + (int64_t)v * a.gs];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) wt[ch][i] = live ? (bus.pan ? bus.pan[ch * bus.pan_ld + v] * scale : scale) : 0.0;
            }
        }
        // homogeneous state at the block's first row; only the launch's very first block can have a short context
        const bool first = (b_first + bi == 0);
        const int tk = first ? SC_T0 : SC_T;
        const int c = first ? (int)((a.position < (int64_t)a.ctx) ? a.position : (int64_t)a.ctx) : a.ctx;
        if constexpr (CROWS) {
            // this block's filter, derived here from cutoff row b (and gain row b): coefficients, bus weights, the steady-state
            // seeds from its H at the voice's frequency, T_c for the block's context and the decay bound per slot.  ~300 f64
            // operations per voice and block (steady_block_constants) against ~1100 for the block's 256 rows -- and no round trip of 80 bytes per
            // (block, voice) through HBM, which a prep launch would cost (measured: 31 us + 29 us per 1024-block batch)
            const int64_t blk = b_first + bi;
            const double q_b = (double)(p0 + (int64_t)bi * a.N) / a.rate;
            // every load of the block first, all voices side by side: the rows of block b come from HBM, and one voice after
            // the other (each voice's constants end in a loop) their latencies added up to 20 us per block
            double cut_i[VPT], gain_i[VPT];                                    // (the block-invariant rows -- pan, hertz, phase -- sit in the caches)
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = (v0 + i < a.voices) ? v0 + i : vc;
                cut_i[i] = a.cutoff[(a.cutoff_rows > 1 ? blk * (int64_t)(a.cs ? a.voices : 1) : 0) + (int64_t)v * a.cs];
                gain_i[i] = a.gain ? a.gain[(a.gain_rows > 1 ? blk * (int64_t)(a.gs ? a.voices : 1) : 0) + (int64_t)v * a.gs] : 1.0;
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const bool live = v0 + i < a.voices;
                const int v = live ? v0 + i : vc;
                const double cutoff = cut_i[i], gain = gain_i[i];
                const double hz_v = a.hertz[(int64_t)v * a.hs], ph_v = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
                // the oscillator's part is kept per voice for c = ctx; the launch's first block may have a shorter context
                OscPart op;
                if constexpr (OSC_LDS) op = OscPart{osc_slot(0, i), osc_slot(1, i), osc_slot(2, i), osc_slot(3, i), osc_slot(4, i)};
                else op = osc[i];
                if (__builtin_expect(c != a.ctx, 0)) op = steady_osc_part(a.type, hz_v, a.rate, c);      // (wave-uniform, the first block of a stream only)
                const double ct_i = op.ct, st_i = op.st;
                const BlockVoice cv = steady_block_constants(a.type, a.rate, cutoff, gain, op, c);
                if (i < LC) { na1[i < LC ? i : 0] = cv.na1; na2[i < LC ? i : 0] = cv.na2; }
                if (live && !cv.ok && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
#pragma unroll
                for (int ch = 0; ch < C; ++ch) wt[ch][i] = live ? (bus.pan ? bus.pan[ch * bus.pan_ld + v] * cv.scale : cv.scale) : 0.0;
                const double t_first = q_b * hz_v + ph_v;                      // osc.py:32
                const double f0 = t_first - rint(t_first);
                const double ur = sin2pi(f0 + 0.25), ui = sin2pi(f0);
                const double wr = fma(cv.hre, ur, -(cv.him * ui)), wi = fma(cv.hre, ui, cv.him * ur);
                yb[i] = wi;
                ya[i] = fma(wi, ct_i, -(wr * st_i));
                if (i < LC) {                                                  // the homogeneous state at the block's first row (zeroed below where the slot has none)
                    const double dss = fma(k2c[i], yb[i], -ya[i]) - yb[i];      // yss_{p+1} - yss_p
                    z0h[i < LC ? i : 0] = fma(cv.T.a, yb[i], cv.T.b * dss);
                    z1h[i < LC ? i : 0] = fma(cv.T.c, yb[i], cv.T.d * dss);
                }
                nd_total[i] = (live && cv.nd < (double)kNeverDrops) ? (int)cv.nd : (live ? kNeverDrops : 0);
            }
#pragma unroll
            for (int d = 1; d < SIG_WAVE; d <<= 1) {
                int other[VPT];
#pragma unroll
                for (int i = 0; i < VPT; ++i) other[i] = __shfl_xor(nd_total[i], d, SIG_WAVE);
#pragma unroll
                for (int i = 0; i < VPT; ++i) nd_total[i] = (other[i] > nd_total[i]) ? other[i] : nd_total[i];
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) nd_total[i] = __builtin_amdgcn_readfirstlane(nd_total[i]);
        }
        int drop_at[LC];                                                       // row of the block from which slot i is dropped
#pragma unroll
        for (int i = 0; i < LC; ++i) {
            drop_at[i] = (nd_total[i] > c) ? nd_total[i] - c : 0;              // wave-uniform
            if (drop_at[i] > 0) {
                const int v = (v0 + i < a.voices) ? v0 + i : vc;
                const double dss = fma(k2c[i], yb[i], -ya[i]) - yb[i];          // yss_{p+1} - yss_p
                if constexpr (!CROWS) {                                       // (CROWS: made with the block's constants above)
                    const double* t = sc + (int64_t)tk * a.voices + v;
                    z0h[i] = fma(t[0], yb[i], t[a.voices] * dss);
                    z1h[i] = fma(t[2 * (int64_t)a.voices], yb[i], t[3 * (int64_t)a.voices] * dss);
                }
            } else {
                z0h[i] = 0.0; z1h[i] = 0.0;
            }
        }
        int done = 0;
        auto single = [&]() {                                                  // (dropped slots carry zeros: the full row is exact)
            row(stage.slot, kTileStride, std::integral_constant<int, LC>{}, std::false_type{});
#pragma unroll
            for (int i = 0; i < VPT; ++i) { const double t = ya[i]; ya[i] = yb[i]; yb[i] = t; }   // back to (previous, current)
            ++done;
            stage.advance();
        };
        while (stage.staged != 0 && done < a.N) single();
        // phases; after single rows `done` is not a multiple of R, the groups simply start there
        const int last_group_row = done + ((a.N - done) / R) * R;
        auto phase = [&](auto k_tag) {
            constexpr int K = decltype(k_tag)::value;
            constexpr int M = Variants::at(K);
            constexpr int lower = (K + 1 < Variants::count) ? Variants::at(K + 1) : 0;
            int until = 0;                                                     // first row at which every slot >= lower has dropped
#pragma unroll
            for (int i = lower; i < LC; ++i) until = (i < M && drop_at[i] > until) ? drop_at[i] : until;
            if (M == 0) until = a.N;
            until = (until < last_group_row) ? until : last_group_row;
            while (done < until) {                                             // (a group that starts before `until` runs whole)
                group(std::integral_constant<int, M>{});
                done += R;
            }
        };
#define SIG_PHASE(K) if constexpr (K < Variants::count) phase(std::integral_constant<int, K>{});
        SIG_PHASE(0) SIG_PHASE(1) SIG_PHASE(2) SIG_PHASE(3) SIG_PHASE(4) SIG_PHASE(5) SIG_PHASE(6) SIG_PHASE(7) SIG_PHASE(8)
#undef SIG_PHASE
        if (done < a.N) {                                                      // rows left over: one at a time, after the pending flush
            if (have) { folded.finish(pend, pend_row, R); have = false; }
#pragma unroll
            for (int i = 0; i < LC; ++i)
                if (drop_at[i] <= done) { z0h[i] = 0.0; z1h[i] = 0.0; }        // dropped slots were not advanced: exact zeros
            while (done < a.N) single();
        }
    }
    if (have) folded.finish(pend, pend_row, R);
    if (stage.staged) stage.now();
}

template <int VPT, int C, bool GROWS = false, bool CROWS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CROWS ? 1 : SteadyOcc<VPT>::waves, 8)))
void fused_steady_bus_kernel(FusedArgs a, BusArgs bus)
{
    __shared__ double lds[4][kPairs * kTileStride];
    // per-block constants at 8 voices per lane: the per-span oscillator parts (5 doubles per voice) do not fit the register file
    // beside the row state -- they live here, [wave][field][voice][lane], 80 KiB per workgroup (one workgroup per CU: the kernel
    // runs one wave per SIMD anyway)
    constexpr bool kOscInLds = CROWS && VPT >= 8;
    __shared__ double osc_lds[kOscInLds ? 4 * 5 * VPT * SIG_WAVE : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform BY CONSTRUCTION: tell the compiler, so that
    steady_bus_wave<VPT, C, GROWS, CROWS>(a, bus, lds[wave], lane, wave,       // everything derived from it lives in SGPRs and branches are scalar
                                          kOscInLds ? osc_lds + (size_t)wave * 5 * VPT * SIG_WAVE : nullptr);
    if (bus.out) sig_bus::sum_tiles_in_workgroup<C>(bus.partials, a.voice_tiles, bus.rows, a.span, a.K, a.N, bus.out, bus.out_ld, lane, wave);
}

// workspace of sig_fused_voice_bus: [tile partials, worst case one tile per 64 voices][steady constants]
int64_t steady_consts_offset(int voices, int64_t rows, int bus_channels) {
    return (int64_t)((voices + SIG_WAVE - 1) / SIG_WAVE) * rows * bus_channels;       // in doubles
}


// Tuning / test hooks.  Product launches read four plain ints; they start from the environment (SIG_FUSED_VPT, _SPAN,
// _STEADY, _SCAN: read ONCE, when the first launch asks) and tests set them through sig_fused_set_tuning.
// (struct Tuning and tuning() are declared in sig_steady.h: ONE instance for both translation units of this file)
#ifndef SIG_FUSED_PART_B
}  // namespace
sig_fused::Tuning& sig_fused::tuning() {
    static Tuning t = [] {
        auto env = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        Tuning u;
        u.vpt = env("SIG_FUSED_VPT", 0); u.span = env("SIG_FUSED_SPAN", 0);
        u.steady = env("SIG_FUSED_STEADY", -1); u.scan = env("SIG_FUSED_SCAN", -1);
        if (u.steady == 2) { u.steady = 1; u.tile_sum_kernel = 1; }
        if (u.steady == 3) { u.steady = 1; u.mix_f32 = 1; }
        return u;
    }();
    return t;
}
namespace {
#endif

// Launch geometry (tools/sweep_fused.sh).  Voices per lane: 4 amortises the per-row work shared by a lane's voices
// (bus staging, loop control) best and leaves room for two waves per SIMD.  Blocks per lane (span): the first
// block of a span pays a c-row warm-up that computes the oscillator only for the filter, so longer spans waste
// less -- but a lane walks its rows serially and one wave per SIMD cannot keep the f64 pipe busy, so the span
// only grows while the launch still has two waves per SIMD; and in the latency regime (one block, few voices)
// the voices are spread over more, shorter waves instead.
constexpr int64_t kWavesWanted = 2048;     // two waves per SIMD

void pick_geometry(const FusedArgs& a, int max_vpt, int& vpt, int& span) {
    const int env_vpt = tuning().vpt, env_span = tuning().span;               // tuning / test hooks
    const int max_span = (a.N >= a.ctx) ? 8 : 1;
    auto waves = [&](int v, int s) { return (int64_t)((a.voices + SIG_WAVE * v - 1) / (SIG_WAVE * v)) * ((a.K + s - 1) / s); };
    vpt = max_vpt; span = max_span;
    while (span > 1 && waves(vpt, span) < kWavesWanted) span >>= 1;
    while (vpt > 1 && waves(vpt, 1) < kWavesWanted / 2) vpt >>= 1;
    if (env_vpt == 1 || env_vpt == 2 || env_vpt == 4) vpt = (env_vpt <= max_vpt) ? env_vpt : max_vpt;
    if (env_span >= 1) span = (env_span <= max_span) ? env_span : max_span;
}

template <int KIND, bool GAIN, int C, bool ROWS = false>
int launch_walk(FusedArgs a, BusArgs bus, int vpt, hipStream_t stream)
{
    a.voice_tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * ((a.K + a.span - 1) / a.span) + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if constexpr (C < 0) {
        fused_walk_kernel<KIND, 1, GAIN, C, ROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus);
    } else {
        switch (vpt) {
            case 1: fused_walk_kernel<KIND, 1, GAIN, C, ROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
            case 2: fused_walk_kernel<KIND, 2, GAIN, C, ROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
            case 4: fused_walk_kernel<KIND, 4, GAIN, C, ROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
            default: return (int)hipErrorInvalidValue;
        }
    }
    return sig_launch_status();
}

struct BusPlan { int vpt, span, steady; };
BusPlan plan_voice_bus(const FusedArgs& a, int kind);
template <bool GAIN, int C, bool GROWS, bool CROWS = false>
int launch_steady(FusedArgs& a, BusArgs& bus, int vpt, float* out, int64_t out_ld, hipStream_t stream);

// the per-block-parameter entry points (sig_fused_osc_biquad_rows, sig_fused_voice_bus_rows, *_fm, *_pair): the walker --
// except a Sine voice whose ONLY per-block parameter is its gain (a tremolo), which keeps the closed form with the bus
// weights rebuilt at every block's first row (fused_steady_bus_kernel<.., GROWS = true>)
template <int KIND, int C>
int launch_rows(FusedArgs a, BusArgs bus, float* out, int64_t out_ld, hipStream_t stream)
{
    if constexpr (KIND == SIG_OSC_SINE && C > 0) {
        const bool gain_only = a.gain && a.gain_rows > 1 && a.cutoff_rows == 1;
        if ((gain_only || a.cutoff_rows > 1) && !a.hertz_hist && !a.phase_hist && a.pair_op == 0) {
            BusPlan plan = plan_voice_bus(a, KIND);
            if (plan.steady) {
                if (plan.vpt > 8) plan.vpt = 8;
                // per-block constants: 8 voices per lane where the launch is big enough for them (one wave per SIMD; the per-span
                // oscillator parts then live in LDS -- in registers the kernel passed 512 and spilled 132, 20 us of scratch round
                // trips per block), else 2 (195 registers: two waves per SIMD under the constants' long dependent chains; 4 need
                // 335 and run one wave, 7 % slower than 2).  Tuning hook: as forced
                if (!gain_only && tuning().vpt == 0) plan.vpt = (plan.vpt >= 8) ? 8 : (plan.vpt > 2 ? 2 : plan.vpt);
                a.span = plan.span;
                a.steady = 1;
                // a swept cutoff (with or without a tremolo): per-(block, voice) filter constants; a tremolo alone: the bus
                // weights rebuilt per block
                const int err = gain_only ? launch_steady<false, C, true>(a, bus, plan.vpt, out, out_ld, stream)
                                          : launch_steady<false, C, true, true>(a, bus, plan.vpt, out, out_ld, stream);
                if (err || bus.out) return err;
                const int tiles_s = (a.voices + SIG_WAVE * plan.vpt - 1) / (SIG_WAVE * plan.vpt);
                return sig_bus::launch_partials<C>(bus.partials, tiles_s, bus.rows, out, out_ld, stream);
            }
        }
    }
    auto ok = [&](int vpt) {
        return C > 0 || ((a.voices % vpt == 0) && (a.out_ld % vpt == 0) && (reinterpret_cast<uintptr_t>(a.out) % (vpt * 4) == 0));
    };
    int max_vpt = 4;
    while (max_vpt > 1 && !ok(max_vpt)) max_vpt >>= 1;
    int vpt;
    pick_geometry(a, max_vpt, vpt, a.span);
    const int tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    if (C > 0 && sig_bus::tiles_sum_in_workgroup(tiles) && tuning().tile_sum_kernel == 0) { bus.out = out; bus.out_ld = out_ld; }
    const int err = launch_walk<KIND, false, C, true>(a, bus, vpt, stream);
    if (err || C == 0 || bus.out) return err;
    return sig_bus::launch_partials<(C > 0 ? C : 1)>(bus.partials, tiles, bus.rows, out, out_ld, stream);
}

template <int C>
int dispatch_rows_kind(int kind, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s)
{
    switch (kind) {
#if SIG_FUSED_HERE_SINE
        case SIG_OSC_SINE: return launch_rows<SIG_OSC_SINE, C>(a, bus, out, out_ld, s);
#endif
#if SIG_FUSED_HERE_OTHERS
        case SIG_OSC_SQUARE: return launch_rows<SIG_OSC_SQUARE, C>(a, bus, out, out_ld, s);
        case SIG_OSC_SAWTOOTH: return launch_rows<SIG_OSC_SAWTOOTH, C>(a, bus, out, out_ld, s);
        case SIG_OSC_TRIANGLE: return launch_rows<SIG_OSC_TRIANGLE, C>(a, bus, out, out_ld, s);
#elif SIG_FUSED_SPLIT
        case SIG_OSC_SQUARE: case SIG_OSC_SAWTOOTH: case SIG_OSC_TRIANGLE: return part_b_rows(C, kind, a, bus, out, out_ld, s);
#endif
    }
    return (int)hipErrorInvalidValue;
}

// What sig_fused_voice_bus launches for this problem: voices per lane, blocks per lane, and whether the Sine closed
// form (fused_steady_bus_kernel) takes the launch.  One decision function for the launcher and for
// sig_fused_voice_bus_plan (tests and bench.py name the kernel they time with it).
BusPlan plan_voice_bus(const FusedArgs& a, int kind) {
    BusPlan p{4, 1, 0};
    pick_geometry(a, 4, p.vpt, p.span);
    if (kind == SIG_OSC_SINE && !a.force_walk && (a.N >= a.ctx || a.position >= a.ctx)) {   // at most the first block has a short context
        p.steady = tuning().steady < 0 ? 1 : tuning().steady;                  // tuning / test hook
        if (p.steady) {
            // the closed form needs few registers per voice: 8 voices per lane (one wave per SIMD, 302 registers) beat
            // 4 (two waves) by 5 % when the launch still has a wave for every SIMD -- half the flushes per sample
            // Better still 16 (the cross-lane flush -- 16 LDS stores and 16 loads per lane per 8 rows, the kernel's real
            // bottleneck: 13 + 8 cycles of the CU's LDS path per pair, shared by four SIMDs -- is paid per LANE and row,
            // so its cost per voice-sample halves), with shorter spans if that is what keeps a wave on every SIMD.
            const int env_vpt = tuning().vpt;                                  // tuning / test hook
            auto waves = [&](int v, int s) { return (int64_t)((a.voices + SIG_WAVE * v - 1) / (SIG_WAVE * v)) * ((a.K + s - 1) / s); };
            if (env_vpt == 8 || (env_vpt == 0 && p.vpt == 4 && waves(8, p.span) >= kWavesWanted / 2)) p.vpt = 8;
            if (env_vpt == 16 || (SIG_STEADY_AUTO16 && env_vpt == 0 && p.vpt == 8 && a.voices >= SIG_WAVE * 16)) {
                int span = p.span;
                while (span > 1 && waves(16, span) < kWavesWanted / 2) span >>= 1;
                if (env_vpt == 16 || waves(16, span) >= kWavesWanted / 2) {
                    p.vpt = 16;
                    if (tuning().span == 0) p.span = span;
                }
            }
        }
    }
    return p;
}

// the closed form: per-voice constants, then one launch (closed form per wave, or its built-in plain fallback
// steady_fallback_span); sets bus.out when the kernel adds the voice tiles itself
template <bool GAIN, int C, bool GROWS, bool CROWS>
int launch_steady(FusedArgs& a, BusArgs& bus, int vpt, float* out, int64_t out_ld, hipStream_t stream)
{
    double* consts = a.consts_ext ? a.consts_ext : bus.partials + steady_consts_offset(a.voices, bus.rows, C);
    a.steady_consts = consts;
    if (!(a.consts_ext && a.consts_ready))
        steady_prep_kernel<GAIN><<<(a.voices + 255) / 256, 256, 0, stream>>>(a, consts);

    a.voice_tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * ((a.K + a.span - 1) / a.span) + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (sig_bus::tiles_sum_in_workgroup(a.voice_tiles) && tuning().tile_sum_kernel == 0) { bus.out = out; bus.out_ld = out_ld; }
    switch (vpt) {
        case 1: fused_steady_bus_kernel<1, C, GROWS, CROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
        case 2: fused_steady_bus_kernel<2, C, GROWS, CROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
        case 8: fused_steady_bus_kernel<8, C, GROWS, CROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
        case 16: if constexpr (!CROWS) { fused_steady_bus_kernel<16, C, GROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break; }
        default: fused_steady_bus_kernel<4, C, GROWS, CROWS><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
    }
    return sig_launch_status();
}

template <int KIND, bool GAIN, int C>
int launch_voice_bus(FusedArgs a, BusArgs bus, float* out, int64_t out_ld, hipStream_t stream)
{
    const BusPlan plan = plan_voice_bus(a, KIND);
    const int vpt = plan.vpt;
    a.span = plan.span;
    a.steady = plan.steady;
    bool done = false;
    if constexpr (KIND == SIG_OSC_SINE) {                                      // (the closed form exists for a sinusoid only: not instantiated for the others)
        if (a.steady) {
            const int e2 = launch_steady<GAIN, C, false>(a, bus, vpt, out, out_ld, stream);
            if (e2 || bus.out) return e2;                                      // (the kernel added the voice tiles itself)
            done = true;
        }
    }
    if (!done) {                                                               // (Sine with the closed form: that launch did every wave)
        const int tiles_w = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
        if (sig_bus::tiles_sum_in_workgroup(tiles_w) && tuning().tile_sum_kernel == 0) { bus.out = out; bus.out_ld = out_ld; }
        const int err = launch_walk<KIND, GAIN, C>(a, bus, vpt, stream);
        if (err || bus.out) return err;                                        // (the kernel added the voice tiles itself)
    }
    const int tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    return sig_bus::launch_partials<C>(bus.partials, tiles, bus.rows, out, out_ld, stream);
}

template <int KIND, bool GAIN>
int dispatch_bus_channels(int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s)
{
    switch (C) {
#ifndef SIG_TUNE_SINE_ONLY
        case 1: return launch_voice_bus<KIND, GAIN, 1>(a, bus, out, out_ld, s);
        case 4: return launch_voice_bus<KIND, GAIN, 4>(a, bus, out, out_ld, s);
#endif
        case 2: return launch_voice_bus<KIND, GAIN, 2>(a, bus, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}

template <bool GAIN>
int dispatch_bus_kind(int kind, int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s)
{
    switch (kind) {
#if SIG_FUSED_HERE_SINE
        case SIG_OSC_SINE: return dispatch_bus_channels<SIG_OSC_SINE, GAIN>(C, a, bus, out, out_ld, s);
#endif
#if SIG_FUSED_HERE_OTHERS
        case SIG_OSC_SQUARE: return dispatch_bus_channels<SIG_OSC_SQUARE, GAIN>(C, a, bus, out, out_ld, s);
        case SIG_OSC_SAWTOOTH: return dispatch_bus_channels<SIG_OSC_SAWTOOTH, GAIN>(C, a, bus, out, out_ld, s);
        case SIG_OSC_TRIANGLE: return dispatch_bus_channels<SIG_OSC_TRIANGLE, GAIN>(C, a, bus, out, out_ld, s);
#elif SIG_FUSED_SPLIT
        case SIG_OSC_SQUARE: case SIG_OSC_SAWTOOTH: case SIG_OSC_TRIANGLE: return part_b_bus(GAIN ? 1 : 0, kind, C, a, bus, out, out_ld, s);
#endif
    }
    return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// Latency mode: wavefront prefix-scan over TIME.  With one block per launch there are only `voices`
// independent chains (16 waves for 1024 voices) and each lane walks c+N rows serially: ~50 us for N=256 on
// an otherwise idle chip.  Here one WAVE owns one (voice, block) and its 64 lanes own consecutive chunks of
// L = ceil((c+N)/64) rows.  The recurrence is affine in the state s = (z0, z1):
//     s_n = A s_{n-1} + B x_n,   y_n = b0 x_n + z0_{n-1},   A = [[-a1, 1], [-a2, 0]]
// so (1) every lane runs its chunk from ZERO state (local outputs + local end state e_l), (2) a 6-step
// Hillis-Steele scan over the lanes with the matrices A^(L 2^k) turns the e_l into true chunk end states,
// (3) every lane adds the homogeneous response of its true start state to its local outputs.
// ~14 serial row steps + 6 scan steps instead of 356.  The scan reassociates the sums, so results match
// the serial kernels to ~1e-13 (f64), not bit for bit.
constexpr int kScanMaxL = 8;                                                  // rows per lane: c + N <= 512


template <int KIND, bool GAIN>
__global__ __launch_bounds__(256) void fused_scan_kernel(FusedArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // one wave = one (voice, block)
    const int v = (int)(item % a.voices);
    const int64_t b = item / a.voices;
    if (b >= a.K) return;
    const int64_t p_b = (a.pos_dev ? *a.pos_dev : a.position) + b * a.N;
    const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);
    const int64_t n0 = p_b - c;
    const int total = c + a.N;
    const int L = (total + SIG_WAVE - 1) / SIG_WAVE;                           // <= kScanMaxL (host-checked)

    Biquad q;
    const bool ok = design_butter2(a.type, a.cutoff[(int64_t)v * a.cs], a.rate, q);
    if (!ok && a.status && lane == 0) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
    const double hz = a.hertz[(int64_t)v * a.hs];
    const double ph = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
    const double g = GAIN ? a.gain[(int64_t)v * a.gs] : 1.0;

    // (1) local pass from zero state
    double yl[kScanMaxL];
    double z0 = 0.0, z1 = 0.0;
#pragma unroll
    for (int k = 0; k < kScanMaxL; ++k) {
        const int r = lane * L + k;
        const bool valid = (k < L) && (r < total);
        const double t = (double)(n0 + r) / a.rate * hz + ph;                  // osc.py:32, same operator order
        double x = (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t) : sig_osc::osc_wave_fused<KIND>(t);
        x = valid ? x : 0.0;
        const double y = fma(q.b0, x, z0);
        const double nz0 = fma(q.b1, x, fma(-q.a1, y, z1));
        const double nz1 = fma(q.b2, x, -q.a2 * y);
        yl[k] = y;
        if (k < L) { z0 = nz0; z1 = nz1; }                                     // rows past the chunk do not exist
    }

    // (2) scan of chunk end states: S_l = M S_{l-1} + e_l,  M = A^L
    const M2 A = {-q.a1, 1.0, -q.a2, 0.0};
    M2 M = A;
    for (int k = 1; k < L; ++k) M = m2_mul(A, M);
    double s0 = z0, s1 = z1;
#pragma unroll
    for (int d = 1; d < SIG_WAVE; d <<= 1) {
        const double p0 = __hiloint2double(__shfl_up(__double2hiint(s0), d, SIG_WAVE), __shfl_up(__double2loint(s0), d, SIG_WAVE));
        const double p1 = __hiloint2double(__shfl_up(__double2hiint(s1), d, SIG_WAVE), __shfl_up(__double2loint(s1), d, SIG_WAVE));
        if (lane >= d) {
            s0 += fma(M.a, p0, M.b * p1);
            s1 += fma(M.c, p0, M.d * p1);
        }
        M = m2_mul(M, M);
    }
    // true start state of this lane's chunk = end state of the previous lane's chunk
    double t0 = __hiloint2double(__shfl_up(__double2hiint(s0), 1, SIG_WAVE), __shfl_up(__double2loint(s0), 1, SIG_WAVE));
    double t1 = __hiloint2double(__shfl_up(__double2hiint(s1), 1, SIG_WAVE), __shfl_up(__double2loint(s1), 1, SIG_WAVE));
    if (lane == 0) { t0 = 0.0; t1 = 0.0; }

    // (3) homogeneous response of the start state, added to the local outputs
    float* dst = a.out + (b * a.N - c) * a.out_ld + v;
#pragma unroll
    for (int k = 0; k < kScanMaxL; ++k) {
        const int r = lane * L + k;
        const double yh = t0;                                                  // y = b0*0 + z0
        const double y = yl[k] + yh;
        const double u0 = fma(-q.a1, yh, t1);
        t1 = -q.a2 * yh;
        t0 = u0;
        if (k < L && r >= c && r < total) dst[(int64_t)r * a.out_ld] = (float)(GAIN ? y * g : y);
    }
}


// chains below which the serial walk leaves most of the chip idle (one wave per SIMD = 65536 lanes)
constexpr int64_t kScanMaxChains = 16384;

template <int KIND, bool GAIN>
int launch_fused(FusedArgs a, hipStream_t stream)
{
    {
        const int scan_env = tuning().scan;                                    // tuning / test hook
        const int64_t chains = (int64_t)a.voices * a.K;
        const bool fits = a.ctx + a.N <= kScanMaxL * SIG_WAVE;
        const bool want = scan_env >= 0 ? scan_env != 0 : chains <= kScanMaxChains;
        if (fits && want) {
            const int64_t nwg = (chains + 3) / 4;
            if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
            fused_scan_kernel<KIND, GAIN><<<(unsigned)nwg, 256, 0, stream>>>(a);
            return sig_launch_status();
        }
    }
    auto ok = [&](int vpt) {
        return (a.voices % vpt == 0) && (a.out_ld % vpt == 0) && (reinterpret_cast<uintptr_t>(a.out) % (vpt * 4) == 0);
    };
    int max_vpt = 4;
    while (max_vpt > 1 && !ok(max_vpt)) max_vpt >>= 1;
    int vpt;
    pick_geometry(a, max_vpt, vpt, a.span);
    return launch_walk<KIND, GAIN, 0>(a, BusArgs{nullptr, 0, nullptr, 0}, vpt, stream);
}

template <int KIND, bool GAIN>
int launch_mix(FusedArgs a, hipStream_t stream)
{
    int vpt;
    pick_geometry(a, 1, vpt, a.span);                                          // one voice per lane: a wave = one matrix group
    if (KIND == SIG_OSC_SINE && (tuning().steady < 0 ? 1 : tuning().steady)) { // closed form per wave (or its built-in plain fallback)
        // (fused_mix.hip; blocks per wave: see launch_steady_mix)
        if (tuning().span == 0) a.span = 0;
        a.voice_tiles = a.voices / SIG_WAVE;
        a.steady = tuning().mix_f32 ? 3 : 1;
        return launch_steady_mix(a, GAIN, stream);
    }
    return launch_walk<KIND, GAIN, -1>(a, BusArgs{nullptr, 0, nullptr, 0}, 1, stream);
}

template <bool GAIN>
int dispatch_mix_kind(int kind, const FusedArgs& a, hipStream_t s)
{
    switch (kind) {
#if SIG_FUSED_HERE_SINE
        case SIG_OSC_SINE: return launch_mix<SIG_OSC_SINE, GAIN>(a, s);
#endif
#if SIG_FUSED_HERE_OTHERS
        case SIG_OSC_SQUARE: return launch_mix<SIG_OSC_SQUARE, GAIN>(a, s);
        case SIG_OSC_SAWTOOTH: return launch_mix<SIG_OSC_SAWTOOTH, GAIN>(a, s);
        case SIG_OSC_TRIANGLE: return launch_mix<SIG_OSC_TRIANGLE, GAIN>(a, s);
#elif SIG_FUSED_SPLIT
        case SIG_OSC_SQUARE: case SIG_OSC_SAWTOOTH: case SIG_OSC_TRIANGLE: return part_b_mix(GAIN ? 1 : 0, kind, a, s);
#endif
    }
    return (int)hipErrorInvalidValue;
}

template <bool GAIN>
int dispatch_kind(int kind, const FusedArgs& a, hipStream_t s)
{
    switch (kind) {
#if SIG_FUSED_HERE_SINE
        case SIG_OSC_SINE: return launch_fused<SIG_OSC_SINE, GAIN>(a, s);
#endif
#if SIG_FUSED_HERE_OTHERS
        case SIG_OSC_SQUARE: return launch_fused<SIG_OSC_SQUARE, GAIN>(a, s);
        case SIG_OSC_SAWTOOTH: return launch_fused<SIG_OSC_SAWTOOTH, GAIN>(a, s);
        case SIG_OSC_TRIANGLE: return launch_fused<SIG_OSC_TRIANGLE, GAIN>(a, s);
#elif SIG_FUSED_SPLIT
        case SIG_OSC_SQUARE: case SIG_OSC_SAWTOOTH: case SIG_OSC_TRIANGLE: return part_b_chain(GAIN ? 1 : 0, kind, a, s);
#endif
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

#ifdef SIG_FUSED_PART_B
// the second translation unit of this file (fused_voice_b.hip): Square, Sawtooth and Triangle; what the first one calls for them
namespace sig_fused {
int part_b_rows(int C, int kind, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s) {
    switch (C) {
        case 0: return dispatch_rows_kind<0>(kind, a, bus, out, out_ld, s);
        case 1: return dispatch_rows_kind<1>(kind, a, bus, out, out_ld, s);
        case 2: return dispatch_rows_kind<2>(kind, a, bus, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}
int part_b_bus(int gain, int kind, int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s) {
    return gain ? dispatch_bus_kind<true>(kind, C, a, bus, out, out_ld, s) : dispatch_bus_kind<false>(kind, C, a, bus, out, out_ld, s);
}
int part_b_mix(int gain, int kind, const FusedArgs& a, hipStream_t s) {
    return gain ? dispatch_mix_kind<true>(kind, a, s) : dispatch_mix_kind<false>(kind, a, s);
}
int part_b_chain(int gain, int kind, const FusedArgs& a, hipStream_t s) {
    return gain ? dispatch_kind<true>(kind, a, s) : dispatch_kind<false>(kind, a, s);
}
}  // namespace sig_fused
#else       // (the C ABI lives in the first translation unit)

extern "C" int sig_fused_osc_biquad(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                    int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                    const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                    const double* cutoff, int32_t cutoff_stride,
                                    const double* gain, int32_t gain_stride,
                                    float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && out_ld >= voices);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, position, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                out, out_ld, 0, status};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return gain ? dispatch_kind<true>(osc_kind, a, s) : dispatch_kind<false>(osc_kind, a, s);
}

namespace {
struct PairSource { int op, kind2; const double* hertz2; int hs2; const double* phase2; int ps2; const double* mix; int ms; };

bool pair_ok(const PairSource& p) {
    return p.op == 0 || ((p.op == 1 || p.op == 2) && p.kind2 >= SIG_OSC_SINE && p.kind2 <= SIG_OSC_TRIANGLE && p.hertz2 &&
                         (p.hs2 | 1) == 1 && (p.ps2 | 1) == 1 && (p.ms | 1) == 1 && (p.op == 2 || p.mix));
}
// block-rate FM (sig_fused_*_fm): hertz / phase rows per block + the row in front of the launch
struct FmSource { int hertz_rows = 1, phase_rows = 1; const double* hertz_hist = nullptr; const double* phase_hist = nullptr; };
bool fm_ok(const FmSource& f, int nblocks, const double* phase, int block_frames, int context) {
    if ((f.hertz_hist || f.phase_hist) && block_frames < context) return false;   // (the context of a shorter block is not the previous block's samples)
    return (f.hertz_rows == 1 || (f.hertz_rows == nblocks && f.hertz_hist)) &&
           (f.phase_rows == 1 || (f.phase_rows == nblocks && f.phase_hist)) && (!f.phase_hist || phase);
}
void set_fm(FusedArgs& a, const FmSource& f) {
    a.hertz_rows = f.hertz_rows; a.phase_rows = f.phase_rows; a.hertz_hist = f.hertz_hist; a.phase_hist = f.phase_hist;
}
void set_pair(FusedArgs& a, const PairSource& p) {
    a.pair_op = p.op; a.kind2 = p.kind2; a.hertz2 = p.hertz2; a.hs2 = p.hs2; a.phase2 = p.phase2; a.ps2 = p.ps2;
    a.mixrow = p.mix; a.ms = p.ms;
}

int fused_chain_general(int osc_kind, int filt_type, int32_t rate, int64_t position, int32_t block_frames, int32_t nblocks,
                        int32_t context, int32_t voices, const double* hertz, int32_t hertz_stride, const double* phase,
                        int32_t phase_stride, const PairSource& pair, const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                        const double* gain, int32_t gain_stride, int32_t gain_rows, float* out, int64_t out_ld, int32_t* status,
                        void* stream, const FmSource& fm = FmSource{})
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && out_ld >= voices && pair_ok(pair) && fm_ok(fm, nblocks, phase, block_frames, context));
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    SIG_CHECK_ARG((cutoff_rows == 1 || cutoff_rows == nblocks) && (gain_rows == 1 || gain_rows == nblocks));
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, position, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                out, out_ld, 0, status};
    a.cutoff_rows = cutoff_rows; a.gain_rows = gain_rows;
    set_pair(a, pair);
    set_fm(a, fm);
    return dispatch_rows_kind<0>(osc_kind, a, BusArgs{nullptr, 0, nullptr, 0}, out, out_ld, static_cast<hipStream_t>(stream));
}

int fused_bus_general(int osc_kind, int filt_type, int32_t rate, int64_t position, int32_t block_frames, int32_t nblocks,
                      int32_t context, int32_t voices, const double* hertz, int32_t hertz_stride, const double* phase,
                      int32_t phase_stride, const PairSource& pair, const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                      const double* gain, int32_t gain_stride, int32_t gain_rows, const double* bus_gains, int64_t bus_gains_ld,
                      int32_t bus_channels, double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream,
                      const FmSource& fm = FmSource{})
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && workspace && out_ld >= bus_channels && pair_ok(pair) && fm_ok(fm, nblocks, phase, block_frames, context));
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    SIG_CHECK_ARG((cutoff_rows == 1 || cutoff_rows == nblocks) && (gain_rows == 1 || gain_rows == nblocks));
    SIG_CHECK_ARG(bus_gains ? bus_gains_ld >= voices : bus_channels == 1);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, position, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                nullptr, 0, 0, status};
    a.cutoff_rows = cutoff_rows; a.gain_rows = gain_rows;
    set_pair(a, pair);
    set_fm(a, fm);
    BusArgs bus{bus_gains, bus_gains_ld, workspace, (int64_t)block_frames * nblocks};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (bus_channels) {
        case 1: return dispatch_rows_kind<1>(osc_kind, a, bus, out, out_ld, s);
        case 2: return dispatch_rows_kind<2>(osc_kind, a, bus, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;                                          // (4-channel buses: the per-node schedule)
}
}  // namespace

extern "C" int sig_fused_osc_pair_biquad(int osc_kind, int osc2_kind, int pair_op, int filt_type, int32_t rate, int64_t position,
                                         int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                         const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                         const double* hertz2, int32_t hertz2_stride, const double* phase2, int32_t phase2_stride,
                                         const double* mix, int32_t mix_stride,
                                         const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                         const double* gain, int32_t gain_stride, int32_t gain_rows,
                                         float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(pair_op == 1 || pair_op == 2);
    return fused_chain_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                               phase_stride, PairSource{pair_op, osc2_kind, hertz2, hertz2_stride, phase2, phase2_stride, mix, mix_stride},
                               cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, out, out_ld, status, stream);
}

extern "C" int sig_fused_voice_pair_bus(int osc_kind, int osc2_kind, int pair_op, int filt_type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                        const double* hertz2, int32_t hertz2_stride, const double* phase2, int32_t phase2_stride,
                                        const double* mix, int32_t mix_stride,
                                        const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                        const double* gain, int32_t gain_stride, int32_t gain_rows,
                                        const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                        double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(pair_op == 1 || pair_op == 2);
    return fused_bus_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                             phase_stride, PairSource{pair_op, osc2_kind, hertz2, hertz2_stride, phase2, phase2_stride, mix, mix_stride},
                             cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, bus_gains, bus_gains_ld, bus_channels,
                             workspace, out, out_ld, status, stream);
}

extern "C" int sig_fused_osc_biquad_rows(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                         int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                         const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                         const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                         const double* gain, int32_t gain_stride, int32_t gain_rows,
                                         float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_chain_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                               phase_stride, PairSource{}, cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, out, out_ld,
                               status, stream);
}

extern "C" int sig_fused_voice_bus_rows(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                        const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                        const double* gain, int32_t gain_stride, int32_t gain_rows,
                                        const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                        double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_bus_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                             phase_stride, PairSource{}, cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, bus_gains,
                             bus_gains_ld, bus_channels, workspace, out, out_ld, status, stream);
}

extern "C" int sig_fused_osc_biquad_fm(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                       int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                       const double* hertz, int32_t hertz_stride, int32_t hertz_rows, const double* hertz_hist,
                                       const double* phase, int32_t phase_stride, int32_t phase_rows, const double* phase_hist,
                                       const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                       const double* gain, int32_t gain_stride, int32_t gain_rows,
                                       float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_chain_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                               phase_stride, PairSource{}, cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, out, out_ld,
                               status, stream, FmSource{hertz_rows, phase_rows, hertz_hist, phase_hist});
}

extern "C" int sig_fused_voice_bus_fm(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                      int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                      const double* hertz, int32_t hertz_stride, int32_t hertz_rows, const double* hertz_hist,
                                      const double* phase, int32_t phase_stride, int32_t phase_rows, const double* phase_hist,
                                      const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                                      const double* gain, int32_t gain_stride, int32_t gain_rows,
                                      const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                      double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_bus_general(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride, phase,
                             phase_stride, PairSource{}, cutoff, cutoff_stride, cutoff_rows, gain, gain_stride, gain_rows, bus_gains,
                             bus_gains_ld, bus_channels, workspace, out, out_ld, status, stream,
                             FmSource{hertz_rows, phase_rows, hertz_hist, phase_hist});
}

extern "C" int sig_fused_osc_biquad_devpos(int osc_kind, int filt_type, int32_t rate, const int64_t* position_dev,
                                           int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                           const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                           const double* cutoff, int32_t cutoff_stride,
                                           const double* gain, int32_t gain_stride,
                                           float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position_dev && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && out_ld >= voices);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, 0, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                out, out_ld, 0, status, position_dev};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return gain ? dispatch_kind<true>(osc_kind, a, s) : dispatch_kind<false>(osc_kind, a, s);
}

namespace {
__global__ void advance_kernel(int64_t* p, int64_t delta) { *p += delta; }
}  // namespace

extern "C" int sig_advance_position(int64_t* position_dev, int64_t delta, void* stream)
{
    SIG_CHECK_ARG(position_dev != nullptr);
    advance_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(position_dev, delta);
    return sig_launch_status();
}

extern "C" int sig_fused_osc_biquad_mix(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                        const double* cutoff, int32_t cutoff_stride,
                                        const double* gain, int32_t gain_stride,
                                        const float* matrix, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && matrix && out && out_ld >= voices && voices % 64 == 0);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, position, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                out, out_ld, 0, status};
    a.mix = matrix;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return gain ? dispatch_mix_kind<true>(osc_kind, a, s) : dispatch_mix_kind<false>(osc_kind, a, s);
}

extern "C" int sig_fused_geometry(int32_t voices, int32_t block_frames, int32_t nblocks, int32_t context,
                                  int32_t* voices_per_lane, int32_t* blocks_per_lane)
{
    SIG_CHECK_ARG(voices >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices_per_lane && blocks_per_lane);
    FusedArgs a{};
    a.N = block_frames; a.K = nblocks; a.ctx = context; a.voices = voices;
    int vpt = 1, span = 1;
    pick_geometry(a, 4, vpt, span);
    *voices_per_lane = vpt;
    *blocks_per_lane = span;
    return 0;
}

extern "C" int sig_fused_voice_bus_plan(int osc_kind, int64_t position, int32_t voices, int32_t block_frames, int32_t nblocks,
                                       int32_t context, int32_t* voices_per_lane, int32_t* blocks_per_lane,
                                       int32_t* closed_form)
{
    SIG_CHECK_ARG(voices >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && position >= 0);
    SIG_CHECK_ARG(voices_per_lane && blocks_per_lane && closed_form);
    FusedArgs a{};
    a.position = position; a.N = block_frames; a.K = nblocks; a.ctx = context; a.voices = voices;
    const BusPlan p = plan_voice_bus(a, osc_kind);
    *voices_per_lane = p.vpt;
    *blocks_per_lane = p.span;
    *closed_form = p.steady;
    return 0;
}

extern "C" int sig_fused_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane, int32_t closed_form, int32_t scan)
{
    SIG_CHECK_ARG(voices_per_lane >= 0 && blocks_per_lane >= 0 && closed_form >= -1 && closed_form <= 3 && scan >= -1);
    Tuning& t = tuning();
    t.vpt = voices_per_lane; t.span = blocks_per_lane; t.scan = scan;
    t.steady = (closed_form == 2 || closed_form == 3) ? 1 : closed_form;       // 2: closed form on, voice tiles added by partials_kernel
    t.tile_sum_kernel = (closed_form == 2) ? 1 : 0;
    t.mix_f32 = (closed_form == 3) ? 1 : 0;                                    // 3: closed form on, the MixMatrix sink on v_mfma_f32_32x32x2_f32
    return 0;
}

extern "C" int64_t sig_fused_voice_bus_workspace(int32_t voices, int64_t rows, int32_t bus_channels)
{
    return (steady_consts_offset(voices, rows, bus_channels) + (int64_t)kSteadyConsts * voices) * (int64_t)sizeof(double);
}

namespace {
int fused_voice_bus_impl(int osc_kind, int filt_type, int32_t rate, int64_t position,
                         int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                         const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                         const double* cutoff, int32_t cutoff_stride, const double* gain, int32_t gain_stride,
                         const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                         double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream,
                         double* consts, int32_t consts_ready, int force_walk = 0)
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && workspace && out_ld >= bus_channels);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    SIG_CHECK_ARG(bus_gains ? bus_gains_ld >= voices : bus_channels == 1);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    FusedArgs a{filt_type, (double)rate, position, block_frames, nblocks, context, voices,
                hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
                nullptr, 0, 0, status};
    a.consts_ext = consts;
    a.consts_ready = consts_ready;
    a.force_walk = force_walk;
    BusArgs bus{bus_gains, bus_gains_ld, workspace, (int64_t)block_frames * nblocks};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return gain ? dispatch_bus_kind<true>(osc_kind, bus_channels, a, bus, out, out_ld, s)
                : dispatch_bus_kind<false>(osc_kind, bus_channels, a, bus, out, out_ld, s);
}
}  // namespace

extern "C" int sig_fused_voice_bus(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                   int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                   const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                   const double* cutoff, int32_t cutoff_stride,
                                   const double* gain, int32_t gain_stride,
                                   const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                   double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_voice_bus_impl(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride,
                                phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride, bus_gains, bus_gains_ld,
                                bus_channels, workspace, out, out_ld, status, stream, nullptr, 0);
}

extern "C" int sig_fused_voice_bus_walk(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                        const double* cutoff, int32_t cutoff_stride,
                                        const double* gain, int32_t gain_stride,
                                        const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                        double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    return fused_voice_bus_impl(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride,
                                phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride, bus_gains, bus_gains_ld,
                                bus_channels, workspace, out, out_ld, status, stream, nullptr, 0, 1);
}

extern "C" int64_t sig_fused_voice_consts_size(int32_t voices)
{
    return (int64_t)kSteadyConsts * voices * (int64_t)sizeof(double);
}

extern "C" int sig_fused_voice_bus_prepared(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                            int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                            const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                            const double* cutoff, int32_t cutoff_stride,
                                            const double* gain, int32_t gain_stride,
                                            const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                            double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream,
                                            double* consts, int32_t consts_ready)
{
    SIG_CHECK_ARG(consts != nullptr);
    return fused_voice_bus_impl(osc_kind, filt_type, rate, position, block_frames, nblocks, context, voices, hertz, hertz_stride,
                                phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride, bus_gains, bus_gains_ld,
                                bus_channels, workspace, out, out_ld, status, stream, consts, consts_ready);
}

// sig_fused_voice_bus_prepared / _walk with everything but the position, the output and the stream in a caller-held block: a
// host binding that marshals every argument per call (ctypes: ~5 us for the 26 of them) pays that once
extern "C" int sig_fused_voice_bus_bound(const sig_fused_voice_bus_call* c, int64_t position, float* out, int32_t consts_ready,
                                         int32_t walk, void* stream)
{
    SIG_CHECK_ARG(c != nullptr && (walk || c->consts != nullptr));
    return fused_voice_bus_impl(c->osc_kind, c->filt_type, c->rate, position, c->block_frames, c->nblocks, c->context, c->voices,
                                c->hertz, c->hertz_stride, c->phase, c->phase_stride, c->cutoff, c->cutoff_stride, c->gain, c->gain_stride,
                                c->bus_gains, c->bus_gains_ld, c->bus_channels, c->workspace, out, c->out_ld, c->status, stream,
                                walk ? nullptr : c->consts, walk ? 0 : consts_ready, walk ? 1 : 0);
}
#endif  // SIG_FUSED_PART_B
