// Fused filter cascade for gfx950:  Osc -> Filter1 -> Filter2 [-> x ADSR envelope] -> [gain x pan] -> bus partial
// sums, K blocks per launch, nothing per-voice through HBM (BASELINE config 3: Sawtooth -> LowPass -> LowPass -> x ADSR
// -> SumBus, 40 algorithmic B per voice-sample on the per-node schedule, 0.03 here).
//
// What makes a cascade different from a single filter is the reference's block cache (chain/__init__.py:431-442,
// SURVEY.md 8a A9): the outer filter's 100 context rows are not a fresh cold start of the inner filter but a SLICE OF
// THE INNER FILTER'S PREVIOUS BLOCK, which was itself cold-started 100 rows before that block (fx.py:85-106).  So for
// block b at frame p:
//     inner_b  = Filter1 from zero state over osc rows [p - 100, p + N),         kept rows [p, p + N)
//     outer_b  = Filter2 from zero state over [inner_{b-1}'s last 100 rows | inner_b], kept rows [p, p + N)
// A lane owns `span` consecutive blocks of its voices and walks time once with ONE pair of recurrences per voice (inner,
// outer).  The cold starts come from linearity: two solutions of  z' = A z + B x  over the same input differ by a
// homogeneous solution, so the chain cold-started at row r is the running chain minus what the running chain's state at
// r has become,
//     z_cold(p) = z_run(p) - A^(p - r) z_run(r),         A = [[-a1, 1], [-a2, 0]]  (b0-normalised DF2T)
// -- for the inner filter of block b+1 (input: the oscillator) and for its outer filter (input over the 100 context
// rows: inner_b's output, which is what the running outer chain consumes there) alike.  So a lane keeps a copy of both
// states as they stand 100 rows before a block's end and, at the boundary, subtracts A^100 times the copy (a 2x2 power
// by squaring per voice and filter: 80 operations per block instead of two more recurrences over 100 rows).  The first
// block of a span needs inner_{b-1}'s last 100 rows: the lane first walks that whole HISTORY BLOCK (oscillator + inner
// filter, cold-started where the reference cold-started it; the outer filter joins, from zero state, for the last 100
// rows).  For the launch's very first block the caller says where the history block starts (`first_history_start`):
// position - N_previous on a continuing stream, position - min(100, position) on a fresh graph (the reference then
// renders [p - 100, p) as a block of its own), position itself at frame 0 (no history).
//
// Arithmetic as in fused_voice.hip: exact per-row phase t = n / rate * hertz + phase (n / rate for 64 rows at a time, one
// row per lane, broadcast by v_readlane), b0-normalised DF2T (4 operations per filter row, both b0 folded into the output
// weight), envelope as the voice's current linear stage (sig_adsr.h: Segment) folded with the bus weight into one fma
// per (voice, channel, row), bus sums folded across lanes by sig_bus::FoldedGroup.  The inner filter's output reaches
// the outer filter in f64 (the per-node path rounds it to f32 on the way): closer to the f64 reference, 1e-6 parity
// asserted against the oracle in the tests.
#include <type_traits>

#include "sig_adsr.h"
#include "sig_biquad.h"
#include "sig_bus_tile.h"
#include "sig_osc.h"

namespace {

using sig_biquad::Biquad;
using sig_biquad::design_butter2;
using sig_bus::kPairs;

struct CascadeArgs {
    int type1, type2; double rate; int64_t position, first_history_start; int N, K, ctx, voices;
    const double* hertz; int hs; const double* phase; int ps;
    const double* cutoff1; int c1s; const double* cutoff2; int c2s; const double* gain; int gs;
    const double* pan; int64_t pan_ld; double* partials; int64_t rows;
    int voice_tiles, span; int* status;
    float* out = nullptr; int64_t out_ld = 0;   // set: the kernel adds the voice tiles itself (sig_bus::sum_tiles_in_workgroup)
};

constexpr int kFoldTileDoubles = kPairs * sig_bus::kFoldStride;

// A^e for A = [[na1, 1], [na2, 0]], by squaring (e wave-uniform, >= 0)
struct Mat2 { double a, b, c, d; };
__device__ __forceinline__ Mat2 mul(const Mat2& x, const Mat2& y) {
    return {fma(x.a, y.a, x.b * y.c), fma(x.a, y.b, x.b * y.d), fma(x.c, y.a, x.d * y.c), fma(x.c, y.b, x.d * y.d)};
}
__device__ __forceinline__ Mat2 transition_power(double na1, double na2, int e) {
    Mat2 r{1.0, 0.0, 0.0, 1.0}, base{na1, 1.0, na2, 0.0};
    for (; e > 0; e >>= 1) {
        if (e & 1) r = mul(r, base);
        if (e > 1) base = mul(base, base);
    }
    return r;
}
// z <- z - A^e s : the chain cold-started e rows ago, from the running chain z and its state s of e rows ago
__device__ __forceinline__ void restart(double na1, double na2, int e, double s0, double s1, double& z0, double& z1) {
    const Mat2 m = transition_power(na1, na2, e);
    z0 -= fma(m.a, s0, m.b * s1);
    z1 -= fma(m.c, s0, m.d * s1);
}

// The envelope stage a voice is in at time q, as a line in t.  Not inlined: it runs five times per voice over a whole stream,
// and inlined into every row of the checked row groups it set the register budget of the whole kernel.
struct StageLine { double end, slope, l; };
__device__ __attribute__((noinline)) StageLine stage_line(const sig_env::AdsrRows& env, int v, double q) {
    const sig_env::Segment s = sig_env::segment_at(sig_env::load_voice(env, v), q);
    return {s.end, s.slope, fma(-s.slope, s.t0, s.l0)};
}

template <int KIND, int VPT, bool ENV, int C>
__device__ __forceinline__ void cascade_wave(const CascadeArgs& a, const sig_env::AdsrRows& env, double* tile, int lane, int wave)
{
    constexpr int R = kPairs / C;                                              // rows per group (the host checked N % R == 0)
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b_first = (item / a.voice_tiles) * a.span;
    if (b_first >= a.K) return;                                               // wave-uniform
    const int nb = (int)((a.K - b_first < (int64_t)a.span) ? a.K - b_first : (int64_t)a.span);
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const int vc = (v0 < a.voices) ? v0 : 0;

    const double s2a = (a.type1 == SIG_FILT_LOWPASS) ? 2.0 : -2.0, s2b = (a.type2 == SIG_FILT_LOWPASS) ? 2.0 : -2.0;   // b1 / b0
    double hz[VPT], ph[VPT], a1a[VPT], a2a[VPT], a1b[VPT], a2b[VPT], wt[C][VPT];
    double za0[VPT], za1[VPT], zb0[VPT], zb1[VPT];                             // the running chains: inner, outer
    double sa0[VPT], sa1[VPT], sb0[VPT], sb1[VPT];                             // their states `ctx` rows before the block's end
    double seg_end[ENV ? VPT : 1], sw[C][ENV ? VPT : 1], lw[C][ENV ? VPT : 1]; // the envelope's current stage, times the weights
    bool ok = true, any_live = false;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool live = v0 + i < a.voices;
        const int v = live ? v0 + i : vc;                                      // dead voices shadow a live one with weight 0
        any_live |= live;
        hz[i] = a.hertz[(int64_t)v * a.hs];
        ph[i] = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
        Biquad q1, q2;
        ok &= design_butter2(a.type1, a.cutoff1[(int64_t)v * a.c1s], a.rate, q1) || !live;
        ok &= design_butter2(a.type2, a.cutoff2[(int64_t)v * a.c2s], a.rate, q2) || !live;
        a1a[i] = -q1.a1; a2a[i] = -q1.a2; a1b[i] = -q2.a1; a2b[i] = -q2.a2;
        const double scale = q1.b0 * q2.b0 * (a.gain ? a.gain[(int64_t)v * a.gs] : 1.0);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) wt[ch][i] = live ? (a.pan ? a.pan[ch * a.pan_ld + v] * scale : scale) : 0.0;
        za0[i] = za1[i] = zb0[i] = zb1[i] = 0.0;
        sa0[i] = sa1[i] = sb0[i] = sb1[i] = 0.0;
        if (ENV) seg_end[i] = -1.0;                                            // derived at the first output row
    }
    if (!ok && any_live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);

    // n / rate (IEEE f64 divide, osc.py:32) for 64 rows at a time, one row per lane
    int64_t qbase = 0;
    double q_lane = 0.0;
    bool q_valid = false;
    auto ensure = [&](int64_t n, int rows) {                                   // wave-uniform
        if (!q_valid || n < qbase || n + rows > qbase + SIG_WAVE) {
            qbase = n;
            q_lane = (double)(qbase + lane) / a.rate;
            q_valid = true;
        }
    };
    auto osc = [&](double q, int i) {
        const double t = q * hz[i] + ph[i];
        return (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t) : sig_osc::osc_wave_fused<KIND>(t);
    };
    // one step of the b0-normalised DF2T of [1, s2, 1] / [1, -na1, -na2]
    auto biquad = [](double x, double s2, double na1, double na2, double& z0, double& z1) {
        const double y = x + z0;
        z0 = fma(na1, y, fma(s2, x, z1));
        z1 = fma(na2, y, x);
        return y;
    };

    const int64_t p_first = a.position + b_first * a.N;                        // first frame of the span's first block
    const int c = (int)((p_first < (int64_t)a.ctx) ? p_first : (int64_t)a.ctx);   // BlockLoc.before: min(ctx, position)

    // ---- history: the inner filter over the block in front of the span, cold-started where the reference cold-started
    // it; over its last c rows the outer filter of the span's first block warms up on it from zero state
    {
        const int64_t h0 = (b_first == 0) ? a.first_history_start : p_first - a.N;
        const int ch = (int)((h0 < (int64_t)a.ctx) ? h0 : (int64_t)a.ctx);
        int64_t n = h0 - ch;
        for (; n + 4 <= p_first - c; n += 4) {                                 // inner filter only
            ensure(n, 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double q = sig_readlane_f64(q_lane, (int)(n - qbase) + k);
#pragma unroll
                for (int i = 0; i < VPT; ++i) biquad(osc(q, i), s2a, a1a[i], a2a[i], za0[i], za1[i]);
            }
        }
        for (; n < p_first - c; ++n) {
            ensure(n, 1);
            const double q = sig_readlane_f64(q_lane, (int)(n - qbase));
#pragma unroll
            for (int i = 0; i < VPT; ++i) biquad(osc(q, i), s2a, a1a[i], a2a[i], za0[i], za1[i]);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) { sa0[i] = za0[i]; sa1[i] = za1[i]; }
        for (; n < p_first; ++n) {
            ensure(n, 1);
            const double q = sig_readlane_f64(q_lane, (int)(n - qbase));
#pragma unroll
            for (int i = 0; i < VPT; ++i)
                biquad(biquad(osc(q, i), s2a, a1a[i], a2a[i], za0[i], za1[i]), s2b, a1b[i], a2b[i], zb0[i], zb1[i]);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) restart(a1a[i], a2a[i], c, sa0[i], sa1[i], za0[i], za1[i]);   // the first block's inner filter
    }

    double* dstp = a.partials + (int64_t)vt * a.rows * C;                      // [tile][row][c]
    sig_bus::FoldedGroup<C> folded(tile, lane, dstp);
    double pend[4];
    int64_t pend_row = 0, out_row = b_first * a.N;
    bool have = false;

    // One group of R output rows starting at frame n.  SNAP: the states are copied before row `snap_at` of the group (the
    // block's row N - ctx).  CHECKED: some voice's envelope stage ends inside the group, so every row re-derives the
    // stages that have ended (rare: five boundaries per voice).
    auto group = [&](int64_t n, int snap_at, auto snap_tag, auto checked_tag) {
        constexpr bool SNAP = decltype(snap_tag)::value, CHECKED = decltype(checked_tag)::value;
        ensure(n, R);
        double acc[kPairs];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double q = sig_readlane_f64(q_lane, (int)(n - qbase) + k);
            if (SNAP && k == snap_at) {                                        // wave-uniform
#pragma unroll
                for (int i = 0; i < VPT; ++i) { sa0[i] = za0[i]; sa1[i] = za1[i]; sb0[i] = zb0[i]; sb1[i] = zb1[i]; }
            }
            double y[VPT];
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                y[i] = biquad(biquad(osc(q, i), s2a, a1a[i], a2a[i], za0[i], za1[i]), s2b, a1b[i], a2b[i], zb0[i], zb1[i]);
                if (ENV && CHECKED && !(q < seg_end[i])) {                     // a stage boundary: per lane
                    const int v = (v0 + i < a.voices) ? v0 + i : vc;
                    const StageLine s = stage_line(env, v, q);
                    seg_end[i] = s.end;
                    const double l = s.l;                                      // level(t) = l + slope * t within the stage
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) { sw[ch][i] = s.slope * wt[ch][i]; lw[ch][i] = l * wt[ch][i]; }
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < VPT; ++i) s = fma(ENV ? fma(sw[ch][i], q, lw[ch][i]) : wt[ch][i], y[i], s);
                acc[k * C + ch] = s;
            }
#pragma unroll
            for (int g = 0; g < kPairs / 4; ++g)                               // every four sums are folded as soon as they exist
                if (4 * g + 3 < (k + 1) * C && 4 * g + 3 >= k * C)
                    folded.fold4(g, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            if (k + 1 == R / 2 && have) folded.finish(pend, pend_row, R);
        }
        folded.issue(pend);
        pend_row = out_row; out_row += R; have = true;
    };
    auto run_group = [&](int64_t n, int snap_at, auto snap_tag) {
        bool settled = true;
        if (ENV) {
            const double q_last = (double)(n + R - 1) / a.rate;
#pragma unroll
            for (int i = 0; i < VPT; ++i) settled &= q_last < seg_end[i];
            settled = __all(settled);
        }
        if (ENV && !settled) group(n, snap_at, snap_tag, std::true_type{});
        else group(n, snap_at, snap_tag, std::false_type{});
    };

    const int snap_group = (a.N - a.ctx) / R, snap_at = (a.N - a.ctx) % R;     // (N > ctx: host-checked)
    for (int bi = 0; bi < nb; ++bi) {
        const int64_t p_b = p_first + (int64_t)bi * a.N;
        const bool more = bi + 1 < nb;
        const bool snap = more && a.ctx > 0;                                   // (no context rows: the next block starts from zero state)
        const int64_t n_snap = snap ? p_b + (int64_t)snap_group * R : p_b + a.N;
        int64_t n = p_b;
        for (; n < n_snap; n += R) run_group(n, 0, std::false_type{});
        if (snap) {
            run_group(n, snap_at, std::true_type{});
            n += R;
        }
        for (; n < p_b + a.N; n += R) run_group(n, 0, std::false_type{});
        if (snap) {                                                            // the next block's chains: cold-started ctx rows ago
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                restart(a1a[i], a2a[i], a.ctx, sa0[i], sa1[i], za0[i], za1[i]);
                restart(a1b[i], a2b[i], a.ctx, sb0[i], sb1[i], zb0[i], zb1[i]);
            }
        } else if (more) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) za0[i] = za1[i] = zb0[i] = zb1[i] = 0.0;
        }
    }
    if (have) folded.finish(pend, pend_row, R);
}

template <int KIND, int VPT, bool ENV, int C>
__global__ __launch_bounds__(256) void fused_cascade_kernel(CascadeArgs a, sig_env::AdsrRows env)
{
    __shared__ double lds[4][kFoldTileDoubles];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform by construction: keep what follows in SGPRs
    cascade_wave<KIND, VPT, ENV, C>(a, env, lds[wave], lane, wave);
    if (a.out) sig_bus::sum_tiles_in_workgroup<C>(a.partials, a.voice_tiles, a.rows, a.span, a.K, a.N, a.out, a.out_ld, lane, wave);
}

// Launch geometry: voices per lane and blocks per lane.  Two costs pull apart: every span walks one extra block of
// oscillator + inner filter (the history: ~10.6 instructions per row against ~17 per output row), and every output row
// costs ~6 instructions per LANE whatever the lane carries (cross-lane fold, n / rate broadcast, loop) -- so per
// voice-sample  6 / vpt + 12 / span  on top of the arithmetic.  The cheapest pair that still puts a wave on every SIMD
// (1024) wins, ties to more voices per lane; when no pair fills the chip, the one with the most waves (ties to fewer voices per lane).  Measured at
// V = 1024, N = 1024 (tools/time_cascade.py): K = 256 (2,2) 287 us against (4,1) 304-317; K = 1024 (4,4) 792 us,
// (2,8) 840, (1,16) 1120; K = 4096 (4,16) 2760 us, (2,16) 2820, (4,8) 2880.
int g_force_vpt = 0, g_force_span = 0;         // tuning / test hook (sig_fused_cascade_set_tuning); 0 = the heuristic below
int g_tile_sum_kernel = 0;                     // 1 (blocks_per_lane given as its negative): voice tiles added by partials_kernel

void cascade_geometry(int voices, int nblocks, int& vpt, int& span) {
    auto waves = [&](int v, int s) { return (int64_t)((voices + SIG_WAVE * v - 1) / (SIG_WAVE * v)) * ((nblocks + s - 1) / s); };
    double best_cost = 0.0;
    int64_t best_waves = -1;
    vpt = 1; span = 1;
    for (int v = 4; v >= 1; v >>= 1)
        for (int s = 64; s >= 1; s >>= 1) {
            if (s > 1 && s / 2 >= nblocks) continue;                           // (a span longer than the stream is the same launch)
            const int64_t w = waves(v, s);
            const double cost = 6.0 / v + 12.0 / s;
            const bool fills = w >= 1024, best_fills = best_waves >= 1024;
            const bool better = best_waves < 0 || (fills != best_fills ? fills : (fills ? cost < best_cost : w >= best_waves));
            if (better) { vpt = v; span = s; best_cost = cost; best_waves = w; }
        }
    if (g_force_vpt == 1 || g_force_vpt == 2 || g_force_vpt == 4) vpt = g_force_vpt;
    if (g_force_span >= 1) span = g_force_span;
}

template <int KIND, bool ENV, int C>
int launch(CascadeArgs a, const sig_env::AdsrRows& env, float* out, int64_t out_ld, hipStream_t stream)
{
    int vpt;
    cascade_geometry(a.voices, a.K, vpt, a.span);
    a.voice_tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * ((a.K + a.span - 1) / a.span) + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (sig_bus::tiles_sum_in_workgroup(a.voice_tiles) && !g_tile_sum_kernel) { a.out = out; a.out_ld = out_ld; }
    switch (vpt) {
        case 1: fused_cascade_kernel<KIND, 1, ENV, C><<<(unsigned)nwg, 256, 0, stream>>>(a, env); break;
        case 2: fused_cascade_kernel<KIND, 2, ENV, C><<<(unsigned)nwg, 256, 0, stream>>>(a, env); break;
        default: fused_cascade_kernel<KIND, 4, ENV, C><<<(unsigned)nwg, 256, 0, stream>>>(a, env); break;
    }
    const int err = sig_launch_status();
    if (err || a.out) return err;                                              // (the kernel added the voice tiles itself)
    return sig_bus::launch_partials<C>(a.partials, a.voice_tiles, a.rows, out, out_ld, stream);
}

template <int KIND, bool ENV>
int dispatch_channels(int C, const CascadeArgs& a, const sig_env::AdsrRows& env, float* out, int64_t out_ld, hipStream_t s)
{
    switch (C) {
        case 1: return launch<KIND, ENV, 1>(a, env, out, out_ld, s);
        case 2: return launch<KIND, ENV, 2>(a, env, out, out_ld, s);
        case 4: return launch<KIND, ENV, 4>(a, env, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}

template <bool ENV>
int dispatch_kind(int kind, int C, const CascadeArgs& a, const sig_env::AdsrRows& env, float* out, int64_t out_ld, hipStream_t s)
{
    switch (kind) {
        case SIG_OSC_SINE: return dispatch_channels<SIG_OSC_SINE, ENV>(C, a, env, out, out_ld, s);
        case SIG_OSC_SQUARE: return dispatch_channels<SIG_OSC_SQUARE, ENV>(C, a, env, out, out_ld, s);
        case SIG_OSC_SAWTOOTH: return dispatch_channels<SIG_OSC_SAWTOOTH, ENV>(C, a, env, out, out_ld, s);
        case SIG_OSC_TRIANGLE: return dispatch_channels<SIG_OSC_TRIANGLE, ENV>(C, a, env, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int sig_fused_cascade_geometry(int32_t voices, int32_t nblocks, int32_t* voices_per_lane, int32_t* blocks_per_lane)
{
    SIG_CHECK_ARG(voices >= 0 && nblocks >= 0 && voices_per_lane && blocks_per_lane);
    int vpt, span;
    cascade_geometry(voices, nblocks, vpt, span);
    *voices_per_lane = vpt;
    *blocks_per_lane = span;
    return 0;
}

extern "C" int sig_fused_cascade_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane)
{
    SIG_CHECK_ARG(voices_per_lane >= 0);
    g_force_vpt = voices_per_lane;
    g_force_span = blocks_per_lane < 0 ? -blocks_per_lane : blocks_per_lane;   // negative: that span, and the voice tiles added by a second launch
    g_tile_sum_kernel = blocks_per_lane < 0;
    return 0;
}

extern "C" int sig_fused_cascade_bus(int osc_kind, int filt1_type, int filt2_type, int32_t rate, int64_t position,
                                     int64_t first_history_start, int32_t block_frames, int32_t nblocks, int32_t context,
                                     int32_t voices,
                                     const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                     const double* cutoff1, int32_t cutoff1_stride, const double* cutoff2, int32_t cutoff2_stride,
                                     const double* gain, int32_t gain_stride,
                                     const double* const* adsr_params, const int32_t* adsr_strides,
                                     const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                     double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(filt1_type == SIG_FILT_LOWPASS || filt1_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(filt2_type == SIG_FILT_LOWPASS || filt2_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(first_history_start >= 0 && first_history_start <= position);
    SIG_CHECK_ARG((position == 0) == (first_history_start == position));      // history exists exactly when there are frames before the stream
    SIG_CHECK_ARG(position - first_history_start >= (position < context ? position : context));
    SIG_CHECK_ARG(hertz && cutoff1 && cutoff2 && out && workspace && out_ld >= bus_channels);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff1_stride | 1) == 1 && (cutoff2_stride | 1) == 1 &&
                  (gain_stride | 1) == 1);
    SIG_CHECK_ARG(bus_gains ? bus_gains_ld >= voices : bus_channels == 1);
    SIG_CHECK_ARG(bus_channels == 1 || bus_channels == 2 || bus_channels == 4);
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    SIG_CHECK_ARG(block_frames > context && block_frames % (16 / bus_channels) == 0);
    sig_env::AdsrRows env{};
    if (adsr_params) SIG_CHECK_ARG(sig_env::load_rows(adsr_params, adsr_strides, env));
    CascadeArgs a{filt1_type, filt2_type, (double)rate, position, first_history_start, block_frames, nblocks, context, voices,
                  hertz, hertz_stride, phase, phase_stride, cutoff1, cutoff1_stride, cutoff2, cutoff2_stride, gain, gain_stride,
                  bus_gains, bus_gains_ld, workspace, (int64_t)block_frames * nblocks, 0, 1, status};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return adsr_params ? dispatch_kind<true>(osc_kind, bus_channels, a, env, out, out_ld, s)
                       : dispatch_kind<false>(osc_kind, bus_channels, a, env, out, out_ld, s);
}
