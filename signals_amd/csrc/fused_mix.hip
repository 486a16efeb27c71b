// The closed form feeding the MixMatrix sink, gfx950 (sig_fused_osc_biquad_mix with a Sine oscillator, BASELINE config 5:
// 4096-voice Sine -> LowPass -> MixMatrix(64 x 64)); the entry point and the row-walker form for the other oscillators are
// in fused_voice.hip.
#include <type_traits>

#include "sig_mix_tile.h"
#include "sig_steady.h"

namespace sig_fused {
namespace {

// One voice per lane, a wave = one 64-voice matrix group over `span` blocks.  Per stored sample 1 (steady two-term
// recurrence; the output scale is folded into its two seeds) + a conversion, plus 3 while the wave's slowest voice still
// carries its homogeneous part, instead of the walker's 10.45; every 32 rows go through sig_mix::Sink (bf16 MFMAs) while
// the next 32 are produced into the wave's second LDS buffer.  Two waves per SIMD (the sink's matrix operands take 96
// VGPRs).  Measured by leaving parts out (config 5): stores ~22 us, MFMAs ~28 us, vector work ~21 us, per-wave set-up
// ~6 us of ~75 -- they add; the stores (3.6-4.4 TB/s) are the part nearest a roof.  The per-voice constants are derived
// by each wave for its own 64 voices (no workspace in this entry point).  Waves with a voice outside the closed form's
// range walk their blocks row by row with the exact phase.
template <bool GAIN, bool F32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fused_steady_mix_kernel(FusedArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2 * sig_mix::kTileRows * sig_mix::kLdsStride];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ftile = lds[wave];
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b_first = (item / a.voice_tiles) * a.span;
    if (b_first >= a.K) return;                                               // wave-uniform
    const int nb = (int)((a.K - b_first < (int64_t)a.span) ? a.K - b_first : (int64_t)a.span);
    const int v = vt * SIG_WAVE + lane;                                       // voices % 64 == 0 (host-checked): every lane is live
    const int64_t p0 = a.position + b_first * a.N;

    const SteadyVoice sv = steady_constants<GAIN>(a, v);
    if (!sv.ok && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
    const double hz = a.hertz[(int64_t)v * a.hs], ph = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;

    // MixMatrix sink: rows staged as float32, 32 at a time through the matrix cores (sig_mix_tile.h)
    sig_mix::SinkT<F32> sink;                                                  // F32: the exact-f32 MFMA (tuning hook; a.steady == 3)
    sink.init(a.mix, ftile, a.out + (int64_t)vt * 64, a.out_ld, b_first * a.N, lane);
    auto stage = [&](double y) { sink.stage((float)y); };

    const double q_first = (double)p0 / a.rate, q_last = (double)(p0 + (int64_t)nb * a.N - 1) / a.rate;
    if (!__all(steady_voice_ok(hz, ph, a.rate, sv.st, q_first, q_last) && (a.N >= a.ctx || p0 >= a.ctx))) {
        // the plain way: every block on its own from zero state over [c context rows | block], exact per-row phase
        const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;
#pragma unroll 1
        for (int bi = 0; bi < nb; ++bi) {
            const int64_t p_b = p0 + (int64_t)bi * a.N;
            const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);
            double z0 = 0.0, z1 = 0.0;
#pragma unroll 1
            for (int r = -c; r < a.N; ++r) {
                const double t = (double)(p_b + r) / a.rate * hz + ph;         // osc.py:32
                const double x = (double)sig_osc::osc_sine_f32(t);
                const double y = x + z0;
                z0 = fma(sv.na1, y, fma(s2, x, z1));
                z1 = fma(sv.na2, y, x);
                if (r >= 0) stage(y * sv.scale);                               // wave-uniform
            }
        }
        sink.finish();
        return;
    }

    // steady-state oscillator at rows p0 - 1 and p0: w = H e^{j phi}, yss_p0 = Im w, yss_{p0-1} = Im(w e^{-j theta})
    double ya, yb;
    {
        const double t_first = q_first * hz + ph;                              // osc.py:32
        const double f0 = t_first - rint(t_first);                             // exact, |f0| <= 0.5
        const double ur = sin2pi(f0 + 0.25), ui = sin2pi(f0);
        const double wr = fma(sv.hre, ur, -(sv.him * ui)), wi = fma(sv.hre, ui, sv.him * ur);
        yb = wi * sv.scale;                                                    // (everything below is linear in the two seeds)
        ya = fma(wi, sv.ct, -(wr * sv.st)) * sv.scale;
    }
    const int nd_total = wave_max_int((sv.nd < (double)kNeverDrops) ? (int)sv.nd : kNeverDrops);   // NaN: never
    double z0h = 0.0, z1h = 0.0;
    // rows of block bi that still carry the homogeneous part (wave-uniform)
    auto live_rows_of = [&](int bi) {
        const bool first = (b_first + bi == 0);
        const int c = first ? (int)((a.position < (int64_t)a.ctx) ? a.position : (int64_t)a.ctx) : a.ctx;
        const int live = (nd_total > c) ? nd_total - c : 0;
        return (live < a.N) ? live : a.N;
    };
    // homogeneous state at a block's first row
    auto reseed = [&](int bi) {
        const M2& t = (b_first + bi == 0) ? sv.T0 : sv.T;
        const double dss = fma(sv.k2c, yb, -ya) - yb;                          // yss_{p+1} - yss_p
        z0h = fma(t.a, yb, t.b * dss);
        z1h = fma(t.c, yb, t.d * dss);
    };
    auto row = [&](auto live_tag) {
        double y = yb;
        if (decltype(live_tag)::value) {
            const double yh = z0h;
            y += yh;
            z0h = fma(sv.na1, yh, z1h);
            z1h = sv.na2 * yh;
        }
        const double nx = fma(sv.k2c, yb, -ya);
        ya = yb; yb = nx;
        return (float)y;
    };

    if (a.N % sig_mix::kTileRows == 0) {
        // Tiles never straddle a block.  Software pipeline: while tile t goes through the matrix cores, the rows of tile
        // t + 1 are produced into the other LDS buffer in the same straight-line code -- a bf16 MFMA holds the vector issue
        // for a quarter of its cycles only, so the recurrences run in its shadow.  A tile that contains the row at which
        // the homogeneous part is dropped simply keeps it (it is the exact solution; dropping is the approximation).
        const int tpb = a.N / sig_mix::kTileRows, tiles = nb * tpb;
        int live_rows = live_rows_of(0);
        if (live_rows > 0) reseed(0);
#pragma unroll
        for (int k = 0; k < sig_mix::kTileRows; ++k) sink.put(k, (live_rows > 0) ? row(std::true_type{}) : row(std::false_type{}));
        for (int t = 0; t + 1 < tiles; ++t) {
            const int bi = (t + 1) / tpb, r = ((t + 1) % tpb) * sig_mix::kTileRows;
            if (r == 0) {
                live_rows = live_rows_of(bi);
                if (live_rows > 0) reseed(bi);
            }
            if (r < live_rows) {
                sink.flush(sig_mix::kTileRows, [&](int kb) {
#pragma unroll
                    for (int k = 8 * kb; k < 8 * kb + 8; ++k) sink.put_next(k, row(std::true_type{}));
                });
            } else {
                sink.flush(sig_mix::kTileRows, [&](int kb) {
#pragma unroll
                    for (int k = 8 * kb; k < 8 * kb + 8; ++k) sink.put_next(k, row(std::false_type{}));
                });
            }
            sink.swap();
        }
        sink.flush(sig_mix::kTileRows);
        return;
    }
    for (int bi = 0; bi < nb; ++bi) {                                          // any block length: one row at a time
        const int live_rows = live_rows_of(bi);
        if (live_rows > 0) reseed(bi);
        for (int r = 0; r < a.N; ++r) sink.stage((r < live_rows) ? row(std::true_type{}) : row(std::false_type{}));
    }
    sink.finish();
}

}  // namespace

// a.span == 0: blocks per wave chosen here
int launch_steady_mix(const FusedArgs& a_, bool gain, hipStream_t stream)
{
    FusedArgs a = a_;
    if (a.span <= 0) {
        // Blocks per wave: every wave derives its 64 voices' constants and splits the matrix first (~7000 cycles, about two
        // 32-row tiles' worth), so spans grow while the launch still has two waves for every SIMD (the kernel's occupancy)
        a.span = 1;
        while (a.span < 8 && (int64_t)a.voice_tiles * ((a.K + 2 * a.span - 1) / (2 * a.span)) >= 2048) a.span *= 2;
    }
    const int64_t nwg = ((int64_t)a.voice_tiles * ((a.K + a.span - 1) / a.span) + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const bool f32 = a.steady == 3;                                            // sig_fused_set_tuning(closed_form = 3)
    if (gain) {
        if (f32) fused_steady_mix_kernel<true, true><<<(unsigned)nwg, 256, 0, stream>>>(a);
        else fused_steady_mix_kernel<true, false><<<(unsigned)nwg, 256, 0, stream>>>(a);
    } else {
        if (f32) fused_steady_mix_kernel<false, true><<<(unsigned)nwg, 256, 0, stream>>>(a);
        else fused_steady_mix_kernel<false, false><<<(unsigned)nwg, 256, 0, stream>>>(a);
    }
    return sig_launch_status();
}

}  // namespace sig_fused
