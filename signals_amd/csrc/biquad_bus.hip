// Cold-start Butterworth biquad over a materialised input, [x ADSR envelope], summed straight into the bus:
//   out[n, c] = sum_v bus_gains[c, v] * env[n, v] * Filter(x)[n, v]
// i.e. SumBus(Filter(x)) or SumBus(RingMod(Filter(x), ADSR)) (BASELINE config 3's last three nodes) in one pass
// over x: 4 B per voice-sample read, nothing per-voice written, instead of 8 (filter) [+ 8 (envelope x input)] + 4
// (bus read).  Same design and recurrence as biquad.hip (closed-form butter(2), scipy's transposed direct form II
// operation for operation in f64, contract off; reference fx.py:85-121), same envelope as adsr.hip; what is
// skipped relative to the per-node path are the float32 roundings of the filter and RingMod stores, so results
// agree with it to those roundings (and are closer to the f64 reference).  Lanes are voices (1 or 4 per lane),
// rows are walked serially with kRing row loads in flight; the bus sums go through sig_bus::Tile.
#include <type_traits>

#include "sig_adsr.h"
#include "sig_biquad.h"
#include "sig_bus_tile.h"

namespace {

using sig_biquad::Biquad;
using sig_biquad::design_butter2;

struct Args {
    int type; double rate; int64_t position; int N, K, ctx, voices;
    const double* cutoff; int cs, cutoff_blocks;
    const float* in; int64_t in_ld;
    const double* pan; int64_t pan_ld; double* partials; int64_t rows;
    int voice_tiles; int* status;
};

template <int VPT> struct RowVec;
template <> struct RowVec<1> { using type = float; };
template <> struct RowVec<4> { using type = float4; };
__device__ __forceinline__ void unpack(const float& v, double (&x)[1]) { x[0] = v; }
__device__ __forceinline__ void unpack(const float4& v, double (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }

template <int VPT, int kRing, bool ENV, int C>
__global__ __launch_bounds__(256) void biquad_bus_kernel(Args a, sig_env::AdsrRows env)
{
    using Vec = typename RowVec<VPT>::type;
    __shared__ double lds[4][sig_bus::kTileDoubles];
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave = one (voice tile, block)
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b = item / a.voice_tiles;
    if (b >= a.K) return;                                                      // wave-uniform
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const int vc = (v0 < a.voices) ? v0 : 0;                                   // dead lanes shadow voice 0 with weight 0

    const int64_t p_b = a.position + b * a.N;
    const int c = (int)((p_b < (int64_t)a.ctx) ? p_b : (int64_t)a.ctx);        // BlockLoc.before: min(ctx, position)

    const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;                // b1 / b0
    double na1[VPT], na2[VPT], z0[VPT], z1[VPT], w[C][VPT];
    sig_env::Segment seg[ENV ? VPT : 1];                                       // the envelope stage each voice is in
    bool ok = true, any_live = false;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool live = v0 + i < a.voices;
        const int v = live ? v0 + i : vc;
        any_live |= live;
        const double hz = a.cutoff[(a.cutoff_blocks > 1 ? b * (int64_t)(a.cs ? a.voices : 1) : 0) + (int64_t)v * a.cs];
        Biquad q;
        ok &= design_butter2(a.type, hz, a.rate, q) || !live;
        na1[i] = -q.a1; na2[i] = -q.a2;
        z0[i] = 0.0; z1[i] = 0.0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch][i] = live ? (a.pan ? a.pan[ch * a.pan_ld + v] * q.b0 : q.b0) : 0.0;
        if (ENV) seg[i].end = -1.0;                                            // (re)derived at the first output row
    }
    if (!ok && any_live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);

    const int total = c + a.N;
    const float* src = a.in + (b * a.N - c) * a.in_ld + vc;                    // first context row of this block
    sig_bus::Tile<C> bus(lds[threadIdx.x >> 6], lane, a.partials + (int64_t)vt * a.rows * C);
    constexpr int R = sig_bus::Tile<C>::R;
    constexpr int kChunk = (kRing > R) ? kRing : R;                           // rows per unrolled chunk: whole bus groups, whole rings
    static_assert(kChunk % R == 0 && kChunk % kRing == 0, "a chunk of rows is a whole number of bus groups and of ring turns");
    double q_lane = 0.0;

    Vec ring[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
        const int r = (u < total) ? u : total - 1;
        ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)r * a.in_ld);
    }
    // kChunk rows: CHECKED re-derives a voice's envelope stage at the row where it ends (sig_adsr.h: Segment); when no
    // stage of the lane's voices ends inside the chunk (the usual case: a voice has five stage boundaries in its
    // whole life) the unchecked copy runs, whose row is just  y *= l0 + slope * (t - t0)
    auto chunk = [&](int r0, auto checked_tag) {
        constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int r = r0 + u;                                              // rows >= total repeat the last row: never stored
            double x[VPT], y[VPT];
            unpack(ring[u % kRing], x);
            const int rn = (r + kRing < total) ? r + kRing : total - 1;        // refill this slot
            ring[u % kRing] = *reinterpret_cast<const Vec*>(src + (int64_t)rn * a.in_ld);
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                y[i] = x[i] + z0[i];                                           // DF2T of [1, s2, 1] / [1, a1, a2]
                z0[i] = fma(na1[i], y[i], fma(s2, x[i], z1[i]));
                z1[i] = fma(na2[i], y[i], x[i]);
            }
            if (ENV) {
                if ((r & 63) == 0) q_lane = (double)(p_b - c + r + lane) / a.rate;   // n/rate for 64 rows, one per lane
                if (r >= c) {                                                  // wave-uniform: context rows are never stored
                    const double t = sig_readlane_f64(q_lane, r & 63);
#pragma unroll
                    for (int i = 0; i < VPT; ++i) {
                        if (CHECKED && !(t < seg[i].end)) {                    // a stage boundary: rare, per lane
                            const int v = (v0 + i < a.voices) ? v0 + i : vc;
                            seg[i] = sig_env::segment_at(sig_env::load_voice(env, v), t);
                        }
                        y[i] *= fma(seg[i].slope, t - seg[i].t0, seg[i].l0);
                    }
                }
            }
            double acc[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                acc[ch] = 0.0;
#pragma unroll
                for (int i = 0; i < VPT; ++i) acc[ch] = fma(w[ch][i], y[i], acc[ch]);
            }
            bus.put(u % R, acc);                                               // context rows too: their sums are just not stored
            if (u % R == R - 1) {
                const int lo = r - (R - 1);                                    // first row of this group
                if (lo + R > c && lo < total)                                  // wave-uniform
                    bus.flush(b * a.N + lo - c, (c > lo) ? c - lo : 0, (total - lo < R) ? total - lo : R);
            }
        }
    };
    for (int r0 = 0; r0 < total; r0 += kChunk) {
        bool settled = true;
        if (ENV) {
            const double t_last = (double)(p_b - c + r0 + kChunk - 1) / a.rate;
#pragma unroll
            for (int i = 0; i < VPT; ++i) settled &= t_last < seg[i].end;
            settled = __all(settled);
        }
        if (ENV && !settled) chunk(r0, std::true_type{});
        else chunk(r0, std::false_type{});
    }
}

template <bool ENV, int C>
int launch(Args a, const sig_env::AdsrRows& env, float* out, int64_t out_ld, hipStream_t stream)
{
    const bool vec = (a.voices % 4 == 0) && (a.in_ld % 4 == 0) && (reinterpret_cast<uintptr_t>(a.in) % 16 == 0);
    const int vpt = vec ? 4 : 1;
    a.voice_tiles = (a.voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * a.K + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // rows of loads in flight per lane: 16 for the HBM-bound plain form; the envelope form (f64-issue-bound, one wave
    // per SIMD because of its cold stage-derivation code) measured the same with 8 and with 2 voices per lane
    constexpr int kRing = ENV ? 8 : 16;
    if (vec) biquad_bus_kernel<4, kRing, ENV, C><<<(unsigned)nwg, 256, 0, stream>>>(a, env);
    else     biquad_bus_kernel<1, 16, ENV, C><<<(unsigned)nwg, 256, 0, stream>>>(a, env);
    const int err = sig_launch_status();
    if (err) return err;
    return sig_bus::launch_partials<C>(a.partials, a.voice_tiles, a.rows, out, out_ld, stream);
}

template <bool ENV>
int dispatch(int C, const Args& a, const sig_env::AdsrRows& env, float* out, int64_t out_ld, hipStream_t s)
{
    switch (C) {
        case 1: return launch<ENV, 1>(a, env, out, out_ld, s);
        case 2: return launch<ENV, 2>(a, env, out, out_ld, s);
        case 4: return launch<ENV, 4>(a, env, out, out_ld, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int sig_biquad_coldstart_bus(int type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                                        const double* const* adsr_params, const int32_t* adsr_strides,
                                        const float* in, int64_t in_ld, int64_t in_history,
                                        const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                        double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(type == SIG_FILT_LOWPASS || type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(cutoff && in && out && workspace && in_ld >= voices && out_ld >= bus_channels);
    SIG_CHECK_ARG(cutoff_stride == 0 || cutoff_stride == 1);
    SIG_CHECK_ARG(cutoff_blocks == 1 || cutoff_blocks == nblocks);
    SIG_CHECK_ARG(bus_gains ? bus_gains_ld >= voices : bus_channels == 1);
    {   // the caller must have materialised min(context, position) rows in front of `in`
        const int64_t need = position < context ? position : context;
        SIG_CHECK_ARG(in_history >= need);
    }
    sig_env::AdsrRows env{};
    if (adsr_params) SIG_CHECK_ARG(sig_env::load_rows(adsr_params, adsr_strides, env));
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    Args a{type, (double)rate, position, block_frames, nblocks, context, voices, cutoff, cutoff_stride, cutoff_blocks,
           in, in_ld, bus_gains, bus_gains_ld, workspace, (int64_t)block_frames * nblocks, 0, status};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return adsr_params ? dispatch<true>(bus_channels, a, env, out, out_ld, s)
                       : dispatch<false>(bus_channels, a, env, out, out_ld, s);
}
