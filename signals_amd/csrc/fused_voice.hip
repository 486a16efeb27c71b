// Fused voice chain for gfx950: Osc -> cold-start Butterworth biquad -> [x per-voice gain] -> f32 store,
// K blocks per launch.  Chosen by the batched engine when a LowPass/HighPass reads an oscillator nobody
// else consumes (and, optionally, feeds a Gain nobody else consumes): the oscillator samples never touch
// HBM, so the stage costs 4 B/voice-sample (the store) instead of 4 + 8 (+ 8).
//
// Same design and phase arithmetic as the node kernels (sig_osc.h, sig_biquad.h; reference osc.py:26-62,
// fx.py:85-121, fx.py:51-52): f64 phase, f64 recurrence from zero state over [c context rows | block], context
// rows are recomputed (the oscillator is position-pure), the filter input is the oscillator's f64 sample
// rather than its f32-rounded store, the recurrence uses fused multiply-adds (one rounding per FMA instead of
// sosfilt's two), and the gain multiplies the f64 filter output before the single f32 rounding -- i.e.
// closer to the exact f64 recurrence than the materialised path, and 1e-6-parity with the reference.
//
// Mapping: one wave = 64*VPT consecutive voices of ONE block, lanes walk c+N rows serially; the per-row
// quotient n/rate (IEEE f64 divide) is computed 64 rows at a time, one row per lane, and broadcast with
// v_readlane.  f64-VALU-bound: ~(15 osc + 9 filter + 2) x (N+c)/N f64-rate ops per voice-sample.
#include <cstdlib>
#include <type_traits>

#include "sig_biquad.h"
#include "sig_osc.h"

namespace {

using sig_biquad::Biquad;
using sig_biquad::design_butter2;

template <int VPT> struct OutVec;
template <> struct OutVec<1> { using type = float; };
template <> struct OutVec<2> { using type = float2; };
template <> struct OutVec<4> { using type = float4; };

__device__ __forceinline__ void put(float& v, const float (&y)[1]) { v = y[0]; }
__device__ __forceinline__ void put(float2& v, const float (&y)[2]) { v = make_float2(y[0], y[1]); }
__device__ __forceinline__ void put(float4& v, const float (&y)[4]) { v = make_float4(y[0], y[1], y[2], y[3]); }

#ifndef SIG_FUSED_ILP
#define SIG_FUSED_ILP 4
#endif
constexpr int kIlp = SIG_FUSED_ILP;     // rows whose oscillator samples are computed ahead of the recurrence

struct FusedArgs {
    int type; double rate; int64_t position; int N, K, ctx, voices;
    const double* hertz; int hs; const double* phase; int ps;
    const double* cutoff; int cs; const double* gain; int gs;
    float* out; int64_t out_ld; int voice_tiles; int* status;
    const int64_t* pos_dev = nullptr;        // when set, the position is read from device memory (hipGraph replay)
};

template <int KIND, int VPT, bool GAIN>
__global__ __launch_bounds__(256) void fused_osc_biquad_kernel(FusedArgs a)
{
    using Vec = typename OutVec<VPT>::type;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b = item / a.voice_tiles;
    if (b >= a.K) return;                                                     // wave-uniform
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live = v0 < a.voices;
    const int vc = live ? v0 : 0;

    const int64_t p_b = (a.pos_dev ? *a.pos_dev : a.position) + b * a.N;
    const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);
    const int64_t n0 = p_b - c;                                               // absolute frame of row 0
    const int total = c + a.N;

    Biquad q[VPT];
    double z0[VPT], z1[VPT], hz[VPT], ph[VPT], g[VPT], dr[VPT];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = (vc + i < a.voices) ? vc + i : vc;
        ok &= design_butter2(a.type, a.cutoff[(int64_t)v * a.cs], a.rate, q[i]);
        hz[i] = a.hertz[(int64_t)v * a.hs];
        ph[i] = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
        g[i] = GAIN ? a.gain[(int64_t)v * a.gs] : 1.0;
        z0[i] = 0.0; z1[i] = 0.0;
        const double d = hz[i] / a.rate;                                       // revolutions per row
        dr[i] = d - rint(d);
    }
    if (!ok && live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);

    float* dst = a.out + (b * a.N - c) * a.out_ld + vc;                      // rows < c are never stored

    // rows [r_begin, r_end): STORE=false warms the filter up (context rows), STORE=true keeps the block
    auto walk = [&](int r_begin, int r_end, auto store_tag) {
        constexpr bool STORE = decltype(store_tag)::value;
        for (int r0 = r_begin; r0 < r_end; r0 += SIG_WAVE) {
            const double q_lane = (double)(n0 + r0 + lane) / a.rate;         // osc.py:32, one row per lane
            const int lim = (r_end - r0 < SIG_WAVE) ? r_end - r0 : SIG_WAVE;
            // Sine: advance the phase by hertz/rate per row inside the chunk when every |t| of the wave is small
            bool fast = false;
            double f0[VPT];
            if (KIND == SIG_OSC_SINE) {
                bool small = true;
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const double t_first = sig_readlane_f64(q_lane, 0) * hz[i] + ph[i];
                    const double t_last = sig_readlane_f64(q_lane, lim - 1) * hz[i] + ph[i];
                    small &= fabs(t_first) < sig_osc::kSineFastMaxT && fabs(t_last) < sig_osc::kSineFastMaxT;
                    f0[i] = t_first - rint(t_first);                           // exact
                }
                fast = __all(small);
            }
            // oscillator samples of kIlp rows are independent of the filter state: compute them first so their
            // long dependent chains overlap, then run the (serial) recurrence over them
            auto rows = [&](int j, auto count_tag) {
                constexpr int CNT = decltype(count_tag)::value;
                double xs[CNT][VPT];
                if (KIND == SIG_OSC_SINE && fast) {
#pragma unroll
                    for (int u = 0; u < CNT; ++u) {
                        const double jj = (double)(j + u);
#pragma unroll
                        for (int i = 0; i < VPT; ++i) xs[u][i] = (double)sig_osc::osc_sine_f32_fast(f0[i], dr[i], jj);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CNT; ++u) {
                        const double t_s = sig_readlane_f64(q_lane, j + u);
#pragma unroll
                        for (int i = 0; i < VPT; ++i) {
                            const double t = t_s * hz[i] + ph[i];
                            xs[u][i] = (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t)
                                                               : sig_osc::osc_wave<KIND, double>(t);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    float y32[VPT];
#pragma unroll
                    for (int i = 0; i < VPT; ++i) {
                        const double x = xs[u][i];
                        // DF2T with fused multiply-adds: 5 f64 ops instead of sosfilt's 8 separately rounded ones (this
                        // kernel is f64-issue-bound; the per-node biquad kernels keep scipy's exact operation order)
                        const double y = fma(q[i].b0, x, z0[i]);
                        z0[i] = fma(q[i].b1, x, fma(-q[i].a1, y, z1[i]));
                        z1[i] = fma(q[i].b2, x, -q[i].a2 * y);
                        if (STORE) y32[i] = (float)(GAIN ? y * g[i] : y);
                    }
                    if (STORE && live) {
                        Vec o; put(o, y32);
                        *reinterpret_cast<Vec*>(dst + (int64_t)(r0 + j + u) * a.out_ld) = o;
                    }
                }
            };
            int j = 0;
            for (; j + kIlp <= lim; j += kIlp) rows(j, std::integral_constant<int, kIlp>{});
            for (; j < lim; ++j) rows(j, std::integral_constant<int, 1>{});
        }
    };
    walk(0, c, std::false_type{});
    walk(c, total, std::true_type{});
}

// ---------------------------------------------------------------------------------------------------
// Fused voice chain + bus: the same chain, but instead of storing each voice the wave reduces its
// 64*VPT voices into the C bus channels:  partial[tile][row][c] = sum_v pan[c][v] * (gain[v] * y[v]).
// Lanes are voices, so a row's sum is a cross-lane sum; doing it per row with a butterfly would cost as
// much as the chain itself, so rows are staged kGroup at a time in a wave-private LDS tile
// [pair = row*C + c][lane] (row stride 65 doubles: conflict-free for the transposed read) and reduced by
// lane = pair: 16 LDS reads + 2 shuffles per lane per kGroup rows.  A second tiny kernel adds the voice
// tiles in a fixed order (deterministic, no atomics) and rounds to f32.  Nothing but parameters is read
// from HBM and nothing but the bus is written: 8*C B per frame instead of 4 B per voice-sample.
constexpr int kPairs = 16;                 // (row, channel) pairs reduced per flush
constexpr int kTileStride = 65;            // doubles

struct BusArgs { const double* pan; int64_t pan_ld; double* partials; int64_t rows; };

template <int KIND, int VPT, bool GAIN, int C>
__global__ __launch_bounds__(256) void fused_voice_bus_kernel(FusedArgs a, BusArgs bus)
{
    constexpr int R = kPairs / C;          // rows per flush
    __shared__ double lds[4][kPairs * kTileStride];
    const int lane = threadIdx.x & 63;
    double* tile = lds[threadIdx.x >> 6];
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b = item / a.voice_tiles;
    if (b >= a.K) return;                                                     // wave-uniform
    const int v0 = (vt * SIG_WAVE + lane) * VPT;

    const int64_t p_b = (a.pos_dev ? *a.pos_dev : a.position) + b * a.N;
    const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);
    const int64_t n0 = p_b - c;
    const int total = c + a.N;

    Biquad q[VPT];
    double z0[VPT], z1[VPT], hz[VPT], ph[VPT], dr[VPT], w[C][VPT];
    bool ok = true, any_live = false;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool live = v0 + i < a.voices;
        const int v = live ? v0 + i : 0;                                       // dead lanes shadow voice 0 ...
        any_live |= live;
        ok &= design_butter2(a.type, a.cutoff[(int64_t)v * a.cs], a.rate, q[i]) || !live;
        hz[i] = a.hertz[(int64_t)v * a.hs];
        ph[i] = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
        const double gn = GAIN ? a.gain[(int64_t)v * a.gs] : 1.0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch)                                         // ... with weight exactly 0
            w[ch][i] = live ? (bus.pan ? bus.pan[ch * bus.pan_ld + v] * gn : gn) : 0.0;   // pan * gain, once per voice
        z0[i] = 0.0; z1[i] = 0.0;
        const double d = hz[i] / a.rate;                                       // revolutions per row
        dr[i] = d - rint(d);
    }
    if (!ok && any_live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);

    double* dst = bus.partials + ((int64_t)vt * bus.rows + b * a.N) * C;      // [tile][row][c], row 0 = block start
    const int pair = lane & (kPairs - 1), quarter = lane >> 4;

    auto flush = [&](int row_first, int nrows) {                               // rows [row_first, row_first+nrows) of the block
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += tile[pair * kTileStride + quarter * 16 + k];
        s += sig_shfl_xor_f64(s, 16);
        s += sig_shfl_xor_f64(s, 32);
        if (lane < nrows * C) dst[(int64_t)row_first * C + lane] = s;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    auto walk = [&](int r_begin, int r_end, auto store_tag) {
        constexpr bool STORE = decltype(store_tag)::value;
        int staged = 0, first = r_begin - c;
        for (int r0 = r_begin; r0 < r_end; r0 += SIG_WAVE) {
            const double q_lane = (double)(n0 + r0 + lane) / a.rate;         // osc.py:32, one row per lane
            const int lim = (r_end - r0 < SIG_WAVE) ? r_end - r0 : SIG_WAVE;
            // Sine: advance the phase by hertz/rate per row inside the chunk when every |t| of the wave is small
            bool fast = false;
            double f0[VPT];
            if (KIND == SIG_OSC_SINE) {
                bool small = true;
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const double t_first = sig_readlane_f64(q_lane, 0) * hz[i] + ph[i];
                    const double t_last = sig_readlane_f64(q_lane, lim - 1) * hz[i] + ph[i];
                    small &= fabs(t_first) < sig_osc::kSineFastMaxT && fabs(t_last) < sig_osc::kSineFastMaxT;
                    f0[i] = t_first - rint(t_first);                           // exact
                }
                fast = __all(small);
            }
            auto rows = [&](int j, auto count_tag) {
                constexpr int CNT = decltype(count_tag)::value;
                double xs[CNT][VPT];
                if (KIND == SIG_OSC_SINE && fast) {
#pragma unroll
                    for (int u = 0; u < CNT; ++u) {
                        const double jj = (double)(j + u);
#pragma unroll
                        for (int i = 0; i < VPT; ++i) xs[u][i] = (double)sig_osc::osc_sine_f32_fast(f0[i], dr[i], jj);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CNT; ++u) {
                        const double t_s = sig_readlane_f64(q_lane, j + u);
#pragma unroll
                        for (int i = 0; i < VPT; ++i) {
                            const double t = t_s * hz[i] + ph[i];
                            xs[u][i] = (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t)
                                                               : sig_osc::osc_wave<KIND, double>(t);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    double acc[C];
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) acc[ch] = 0.0;
#pragma unroll
                    for (int i = 0; i < VPT; ++i) {
                        const double x = xs[u][i];
                        const double y = fma(q[i].b0, x, z0[i]);              // fused DF2T, see fused_osc_biquad_kernel
                        z0[i] = fma(q[i].b1, x, fma(-q[i].a1, y, z1[i]));
                        z1[i] = fma(q[i].b2, x, -q[i].a2 * y);
                        if (STORE) {
#pragma unroll
                            for (int ch = 0; ch < C; ++ch) acc[ch] = fma(w[ch][i], y, acc[ch]);
                        }
                    }
                    if (STORE) {
#pragma unroll
                        for (int ch = 0; ch < C; ++ch) tile[(staged * C + ch) * kTileStride + lane] = acc[ch];
                        if (++staged == R) { flush(first, R); first += R; staged = 0; }
                    }
                }
            };
            int j = 0;
            for (; j + kIlp <= lim; j += kIlp) rows(j, std::integral_constant<int, kIlp>{});
            for (; j < lim; ++j) rows(j, std::integral_constant<int, 1>{});
        }
        if (STORE && staged) flush(first, staged);
    };
    walk(0, c, std::false_type{});
    walk(c, total, std::true_type{});
}

template <int C>
__global__ __launch_bounds__(256) void bus_partials_kernel(const double* __restrict__ partials, int tiles, int64_t rows,
                                                           float* __restrict__ out, int64_t out_ld)
{
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += partials[(int64_t)t * n + i];       // fixed order
        out[(i / C) * out_ld + (i % C)] = (float)s;
    }
}

// Voices per lane: 4 amortises the per-row scalar work best, but a lane walks its rows serially, so when the
// launch is small (latency mode: one block) spread the voices over more waves instead of fewer, longer ones.
int pick_vpt(int voices, int64_t nblocks) {
    const char* e = getenv("SIG_FUSED_VPT");
    if (e) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) return v; }
    for (int vpt = 4; vpt > 1; vpt >>= 1) {
        const int64_t waves = ((voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt)) * nblocks;
        if (waves >= 1024) return vpt;                                         // one wave per SIMD or more
    }
    return 1;
}

template <int KIND, bool GAIN, int C>
int launch_voice_bus(FusedArgs a, BusArgs bus, float* out, int64_t out_ld, hipStream_t stream)
{
    const int vpt = pick_vpt(a.voices, a.K);
    a.voice_tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * a.K + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    switch (vpt) {
        case 1: fused_voice_bus_kernel<KIND, 1, GAIN, C><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
        case 2: fused_voice_bus_kernel<KIND, 2, GAIN, C><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
        default: fused_voice_bus_kernel<KIND, 4, GAIN, C><<<(unsigned)nwg, 256, 0, stream>>>(a, bus); break;
    }
    int err = sig_launch_status();
    if (err) return err;
    const int64_t n = bus.rows * C;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    bus_partials_kernel<C><<<(unsigned)g, 256, 0, stream>>>(bus.partials, a.voice_tiles, bus.rows, out, out_ld);
    return sig_launch_status();
}

template <int KIND, bool GAIN>
int dispatch_bus_channels(int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s)
{
    switch (C) {
        case 1: return launch_voice_bus<KIND, GAIN, 1>(a, bus, out, out_ld, s);
        case 2: return launch_voice_bus<KIND, GAIN, 2>(a, bus, out, out_ld, s);
        case 4: return launch_voice_bus<KIND, GAIN, 4>(a, bus, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}

template <bool GAIN>
int dispatch_bus_kind(int kind, int C, const FusedArgs& a, const BusArgs& bus, float* out, int64_t out_ld, hipStream_t s)
{
    switch (kind) {
        case SIG_OSC_SINE: return dispatch_bus_channels<SIG_OSC_SINE, GAIN>(C, a, bus, out, out_ld, s);
        case SIG_OSC_SQUARE: return dispatch_bus_channels<SIG_OSC_SQUARE, GAIN>(C, a, bus, out, out_ld, s);
        case SIG_OSC_SAWTOOTH: return dispatch_bus_channels<SIG_OSC_SAWTOOTH, GAIN>(C, a, bus, out, out_ld, s);
        case SIG_OSC_TRIANGLE: return dispatch_bus_channels<SIG_OSC_TRIANGLE, GAIN>(C, a, bus, out, out_ld, s);
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

struct M2 { double a, b, c, d; };                                             // [[a, b], [c, d]]
__device__ __forceinline__ M2 m2_mul(const M2& x, const M2& y) {
    return {fma(x.a, y.a, x.b * y.c), fma(x.a, y.b, x.b * y.d), fma(x.c, y.a, x.d * y.c), fma(x.c, y.b, x.d * y.d)};
}

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
        double x = (KIND == SIG_OSC_SINE) ? (double)sig_osc::osc_sine_f32(t) : sig_osc::osc_wave<KIND, double>(t);
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

int pick_vpt(int voices, int64_t nblocks);

int fused_variant() {
    static int v = [] { const char* e = getenv("SIG_FUSED_VPT"); return e ? atoi(e) : 0; }();
    return v;
}

// chains below which the serial walk leaves most of the chip idle (one wave per SIMD = 65536 lanes)
constexpr int64_t kScanMaxChains = 16384;

template <int KIND, bool GAIN>
int launch_fused(FusedArgs a, hipStream_t stream)
{
    {
        static const int scan_env = [] { const char* e = getenv("SIG_FUSED_SCAN"); return e ? atoi(e) : -1; }();
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
    int vpt = fused_variant() ? fused_variant() : pick_vpt(a.voices, a.K);
    while (vpt > 1 && !ok(vpt)) vpt >>= 1;
    if (!ok(vpt)) vpt = 1;
    const int span = SIG_WAVE * vpt;
    a.voice_tiles = (a.voices + span - 1) / span;
    const int64_t nwg = ((int64_t)a.voice_tiles * a.K + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    switch (vpt) {
        case 1: fused_osc_biquad_kernel<KIND, 1, GAIN><<<(unsigned)nwg, 256, 0, stream>>>(a); break;
        case 2: fused_osc_biquad_kernel<KIND, 2, GAIN><<<(unsigned)nwg, 256, 0, stream>>>(a); break;
        case 4: fused_osc_biquad_kernel<KIND, 4, GAIN><<<(unsigned)nwg, 256, 0, stream>>>(a); break;
        default: return (int)hipErrorInvalidValue;
    }
    return sig_launch_status();
}

template <bool GAIN>
int dispatch_kind(int kind, const FusedArgs& a, hipStream_t s)
{
    switch (kind) {
        case SIG_OSC_SINE: return launch_fused<SIG_OSC_SINE, GAIN>(a, s);
        case SIG_OSC_SQUARE: return launch_fused<SIG_OSC_SQUARE, GAIN>(a, s);
        case SIG_OSC_SAWTOOTH: return launch_fused<SIG_OSC_SAWTOOTH, GAIN>(a, s);
        case SIG_OSC_TRIANGLE: return launch_fused<SIG_OSC_TRIANGLE, GAIN>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

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

extern "C" int64_t sig_fused_voice_bus_workspace(int32_t voices, int64_t rows, int32_t bus_channels)
{
    // worst case: one tile per 64 voices (VPT = 1)
    const int64_t tiles = (voices + SIG_WAVE - 1) / SIG_WAVE;
    return tiles * rows * bus_channels * (int64_t)sizeof(double);
}

extern "C" int sig_fused_voice_bus(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                   int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                   const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                   const double* cutoff, int32_t cutoff_stride,
                                   const double* gain, int32_t gain_stride,
                                   const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                   double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
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
    BusArgs bus{bus_gains, bus_gains_ld, workspace, (int64_t)block_frames * nblocks};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return gain ? dispatch_bus_kind<true>(osc_kind, bus_channels, a, bus, out, out_ld, s)
                : dispatch_bus_kind<false>(osc_kind, bus_channels, a, bus, out, out_ld, s);
}
