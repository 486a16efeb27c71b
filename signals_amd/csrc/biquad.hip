// Cold-start Butterworth biquad bank for gfx950.
// Replaces CritFilter._filter / _get_sos (reference src/signals/chain/fx.py:85-121) for LowPass and
// HighPass: per voice, per block, design butter(N=2) and run scipy's sosfilt recurrence from ZERO
// state over [<=100 context frames | block], keeping the block.  The reference re-warms every block
// (SURVEY.md §0-2), so (voice, block) pairs are independent chains: lanes = voices, and blocks go on
// the grid.  Coefficients and both state registers stay in f64 VGPRs (f32 recurrence misses the 1e-6
// bar); HBM storage is f32.
//
// Mapping: one wave = 64*VPT consecutive voices of ONE block; each lane walks the rows serially with a
// U-deep register ring of row loads in flight (each row of a wave is 64 x 4*VPT B contiguous).
// Algorithmic traffic: 4 B read + 4 B written per voice-sample; the (N+c)/N context re-read is the
// previous block's tail and is served by L2/Infinity Cache when the neighbouring wave ran recently.
#include <cstdlib>

#include "sig_adsr.h"
#include "sig_biquad.h"

namespace {

using sig_biquad::Biquad;
using sig_biquad::design_butter2;

template <typename T, int VPT> struct RowVec;
template <> struct RowVec<float, 1> { using type = float; };
template <> struct RowVec<double, 1> { using type = double; };
template <> struct RowVec<float, 2> { using type = float2; };
template <> struct RowVec<double, 2> { using type = double2; };
template <> struct RowVec<float, 4> { using type = float4; };
template <> struct RowVec<double, 4> { using type = double4; };

template <typename T> __device__ __forceinline__ void unpack(const T& v, double (&x)[1]) { x[0] = (double)v; }
__device__ __forceinline__ void unpack(const float2& v, double (&x)[2]) { x[0] = v.x; x[1] = v.y; }
__device__ __forceinline__ void unpack(const double2& v, double (&x)[2]) { x[0] = v.x; x[1] = v.y; }
__device__ __forceinline__ void pack(float2& v, const double (&y)[2]) { v = make_float2((float)y[0], (float)y[1]); }
__device__ __forceinline__ void pack(double2& v, const double (&y)[2]) { v = make_double2(y[0], y[1]); }
__device__ __forceinline__ void unpack(const float4& v, double (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
__device__ __forceinline__ void unpack(const double4& v, double (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
__device__ __forceinline__ void pack(float& v, const double (&y)[1]) { v = (float)y[0]; }
__device__ __forceinline__ void pack(double& v, const double (&y)[1]) { v = y[0]; }
__device__ __forceinline__ void pack(float4& v, const double (&y)[4]) { v = make_float4((float)y[0], (float)y[1], (float)y[2], (float)y[3]); }
__device__ __forceinline__ void pack(double4& v, const double (&y)[4]) { v = make_double4(y[0], y[1], y[2], y[3]); }

// ENV: multiply the stored rows by a per-voice ADSR envelope evaluated at the row's time (the
// RingMod(Filter, ADSR) pair of BASELINE config 3 without a separate pass; sig_adsr.h).  n/rate is computed for
// 64 rows at a time, one row per lane, and broadcast with v_readlane, like in the oscillator kernels.
template <typename T, int VPT, int kRing, bool ENV>   // kRing = rows of loads in flight per lane
__global__ __launch_bounds__(256) void biquad_coldstart_kernel(
    int type, double rate, int64_t position, int N, int K, int ctx, int voices,
    const double* __restrict__ cutoff, int cs, int cutoff_blocks,
    const T* __restrict__ in, int64_t in_ld, T* __restrict__ out, int64_t out_ld,
    int voice_tiles, int* __restrict__ status, sig_env::AdsrRows env)
{
    using Vec = typename RowVec<T, VPT>::type;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave = one item
    const int vt = (int)(item % voice_tiles);
    const int64_t b = item / voice_tiles;
    if (b >= K) return;                                                        // wave-uniform
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live = v0 < voices;                                             // VPT==4 path: voices % 4 == 0
    const int vc = live ? v0 : 0;                                              // clamp: dead lanes shadow voice 0

    const int64_t p_b = position + b * N;
    const int c = (int)((p_b < (int64_t)ctx) ? p_b : (int64_t)ctx);            // BlockLoc.before: min(ctx, position)

    Biquad q[VPT];
    double z0[VPT], z1[VPT];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = (vc + i < voices) ? vc + i : vc;
        const double hz = cutoff[(cutoff_blocks > 1 ? b * (int64_t)(cs ? voices : 1) : 0) + (int64_t)v * cs];
        ok &= design_butter2(type, hz, rate, q[i]);
        z0[i] = 0.0; z1[i] = 0.0;
    }
    if (!ok && live && status) atomicOr(status, SIG_STATUS_BAD_CUTOFF);
    sig_env::Voice ev[ENV ? VPT : 1];
    double q_lane = 0.0;
    if (ENV) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) ev[i] = sig_env::load_voice(env, (vc + i < voices) ? vc + i : vc);
    }

    const int total = c + N;
    const T* src = in + (b * N - c) * in_ld + vc;        // first context row of this block
    T* dst = out + (b * N - c) * out_ld + vc;            // aligned with src; rows < c are never stored

    Vec ring[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
        const int r = (u < total) ? u : total - 1;
        ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)r * in_ld);
    }

    for (int r0 = 0; r0 < total; r0 += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            const int r = r0 + u;
            const bool valid = r < total;                                      // wave-uniform; tail rows are no-ops
            double x[VPT], y[VPT];
            unpack(ring[u], x);
            const int rn = (r + kRing < total) ? r + kRing : total - 1;        // refill this slot
            ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)rn * in_ld);
            if (valid) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    // scipy _sosfilt, transposed direct form II, one rounding per op (contract off)
                    y[i] = q[i].b0 * x[i] + z0[i];
                    z0[i] = q[i].b1 * x[i] - q[i].a1 * y[i] + z1[i];
                    z1[i] = q[i].b2 * x[i] - q[i].a2 * y[i];
                }
            }
            if (ENV && valid) {
                if ((r & 63) == 0) q_lane = (double)(p_b - c + r + lane) / rate;                       // wave-uniform
                if (r >= c) {
                    const double t = sig_readlane_f64(q_lane, r & 63);
#pragma unroll
                    for (int i = 0; i < VPT; ++i) y[i] *= sig_env::level(ev[i], t);
                }
            }
            if (valid && r >= c && live) {
                Vec o; pack(o, y);
                *reinterpret_cast<Vec*>(dst + (int64_t)r * out_ld) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Span walker: one lane owns VPT voices x `span` CONSECUTIVE blocks and reads every input row ONCE.
// The plain kernel re-reads each block's 100 context rows (they are the previous block's tail): measured
// with rocprofv3 FETCH_SIZE the read side is 1.39x the algorithmic bytes at N=256 and none of it is
// absorbed by L2 / Infinity Cache.  Here block m's warm-up (rows [(m+1)N-ctx, (m+1)N) of block m's main
// region) runs as a SECOND chain on the same loaded row while block m's own chain is still producing
// output; at the block boundary the warm chain becomes the output chain.  Arithmetic per chain is
// unchanged (same zero start, same rows, same order), so results are bit-identical to the plain kernel.
// Needs N > ctx (at most two live chains) and one cutoff row for all blocks.
template <typename T, int VPT, int kRing, bool ENV>
__global__ __launch_bounds__(256) void biquad_walk_kernel(
    int type, double rate, int64_t position, int N, int K, int ctx, int voices, int span,
    const double* __restrict__ cutoff, int cs,
    const T* __restrict__ in, int64_t in_ld, T* __restrict__ out, int64_t out_ld,
    int voice_tiles, int* __restrict__ status, sig_env::AdsrRows env)
{
    using Vec = typename RowVec<T, VPT>::type;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int vt = (int)(item % voice_tiles);
    const int64_t b0 = (item / voice_tiles) * span;                           // first block of this lane's span
    if (b0 >= K) return;                                                       // wave-uniform
    const int nb = (K - b0 < span) ? (int)(K - b0) : span;                     // blocks in this span
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live = v0 < voices;
    const int vc = live ? v0 : 0;

    const int64_t p0 = position + b0 * N;
    const int c0 = (int)((p0 < (int64_t)ctx) ? p0 : (int64_t)ctx);             // only the span's first block can be short

    Biquad q[VPT];
    double a0[VPT], a1[VPT], w0[VPT], w1[VPT];                                  // output chain / warm-up chain states
    bool ok = true;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = (vc + i < voices) ? vc + i : vc;
        ok &= design_butter2(type, cutoff[(int64_t)v * cs], rate, q[i]);
        a0[i] = a1[i] = w0[i] = w1[i] = 0.0;
    }
    if (!ok && live && status) atomicOr(status, SIG_STATUS_BAD_CUTOFF);
    sig_env::Voice ev[ENV ? VPT : 1];
    double q_lane = 0.0;
    if (ENV) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) ev[i] = sig_env::load_voice(env, (vc + i < voices) ? vc + i : vc);
    }

    const int total = c0 + nb * N;
    const T* src = in + (b0 * N - c0) * in_ld + vc;
    T* dst = out + (b0 * N - c0) * out_ld + vc;

    Vec ring[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
        const int r = (u < total) ? u : total - 1;
        ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)r * in_ld);
    }

    int in_block = -c0;                 // row index inside the current block; negative = the first block's own warm-up
    int blocks_left = nb;
    for (int r0 = 0; r0 < total; r0 += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            const int r = r0 + u;
            const bool valid = r < total;                                      // wave-uniform; tail rows are no-ops
            double x[VPT], y[VPT];
            unpack(ring[u], x);
            const int rn = (r + kRing < total) ? r + kRing : total - 1;
            ring[u] = *reinterpret_cast<const Vec*>(src + (int64_t)rn * in_ld);
            const bool warm = valid && (blocks_left > 1) && (in_block >= N - ctx);   // next block's context rows
            if (valid) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    y[i] = q[i].b0 * x[i] + a0[i];
                    a0[i] = q[i].b1 * x[i] - q[i].a1 * y[i] + a1[i];
                    a1[i] = q[i].b2 * x[i] - q[i].a2 * y[i];
                }
            }
            if (warm) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const double yw = q[i].b0 * x[i] + w0[i];
                    w0[i] = q[i].b1 * x[i] - q[i].a1 * yw + w1[i];
                    w1[i] = q[i].b2 * x[i] - q[i].a2 * yw;
                }
            }
            if (ENV && valid) {
                if ((r & 63) == 0) q_lane = (double)(p0 - c0 + r + lane) / rate;                       // wave-uniform
                if (in_block >= 0) {
                    const double t = sig_readlane_f64(q_lane, r & 63);
#pragma unroll
                    for (int i = 0; i < VPT; ++i) y[i] *= sig_env::level(ev[i], t);
                }
            }
            if (valid && in_block >= 0 && live) {
                Vec o; pack(o, y);
                *reinterpret_cast<Vec*>(dst + (int64_t)r * out_ld) = o;
            }
            in_block += valid ? 1 : 0;
            if (in_block == N) {                                                // block boundary: warm chain takes over
                in_block = 0;
                --blocks_left;
#pragma unroll
                for (int i = 0; i < VPT; ++i) { a0[i] = w0[i]; a1[i] = w1[i]; w0[i] = 0.0; w1[i] = 0.0; }
            }
        }
    }
}

static int biquad_variant() {
    // tuning hook: SIG_BIQUAD_VARIANT=<vpt><ring> e.g. "416" = 4 voices/lane, 16-row ring
    static int v = [] { const char* e = getenv("SIG_BIQUAD_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}

template <typename T, bool ENV>
int launch_biquad(int type, int32_t rate, int64_t position, int32_t N, int32_t K, int32_t ctx, int32_t voices,
                  const double* cutoff, int32_t cs, int32_t cutoff_blocks,
                  const T* in, int64_t in_ld, T* out, int64_t out_ld, int32_t* status, hipStream_t stream,
                  const sig_env::AdsrRows& env)
{
    auto ok = [&](int vpt) {
        return (voices % vpt == 0) && (in_ld % vpt == 0) && (out_ld % vpt == 0) &&
               (reinterpret_cast<uintptr_t>(in) % (vpt * sizeof(T)) == 0) &&
               (reinterpret_cast<uintptr_t>(out) % (vpt * sizeof(T)) == 0);
    };
    int variant = biquad_variant();
    // span walker: every row read once (see biquad_walk_kernel).  Keep >= ~1024 waves on the chip.
    {
        static const int walk_env = [] { const char* e = getenv("SIG_BIQUAD_WALK"); return e ? atoi(e) : -1; }();
        static const int walk_variant = [] { const char* e = getenv("SIG_WALK_VARIANT"); return e ? atoi(e) : 0; }();
        int wvpt = walk_variant ? walk_variant / 100 : 4;                     // tuning: <vpt><ring>, e.g. 216
        const int wring = walk_variant ? walk_variant % 100 : 16;
        while (wvpt > 1 && !ok(wvpt)) wvpt >>= 1;
        const int tiles = (voices + SIG_WAVE * wvpt - 1) / (SIG_WAVE * wvpt);
        // measured on C2 at K=1024 (vpt x ring x span sweep): 4 voices/lane, 16 rows in flight, span 4 is the best
        // point (5.0 TB/s algorithmic); longer spans starve the chip of waves, shorter ones re-read more context
        int span = (int)(((int64_t)tiles * K) / 1024);
        if (span > 4) span = 4;
        if (walk_env >= 0) span = walk_env;                                    // tuning: 0/1 disables
        if (span >= 2 && N > ctx && cutoff_blocks == 1 && variant == 0) {
            const int64_t items = (int64_t)tiles * ((K + span - 1) / span);
            const int64_t nwg = (items + 3) / 4;
            if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
#define SIG_WALK(V, R) biquad_walk_kernel<T, V, R, ENV><<<(unsigned)nwg, 256, 0, stream>>>( \
                type, (double)rate, position, N, K, ctx, voices, span, cutoff, cs, in, in_ld, out, out_ld, tiles, status, env)
            switch (wvpt * 100 + wring) {
                case 116: SIG_WALK(1, 16); break;
                case 132: SIG_WALK(1, 32); break;
                case 216: SIG_WALK(2, 16); break;
                case 232: SIG_WALK(2, 32); break;
                case 408: SIG_WALK(4, 8); break;
                case 416: SIG_WALK(4, 16); break;
                case 432: SIG_WALK(4, 32); break;
                default: return (int)hipErrorInvalidValue;
            }
#undef SIG_WALK
            return sig_launch_status();
        }
    }
    int vpt = variant ? variant / 100 : 4;
    int ring = variant ? variant % 100 : 8;
    if (!variant) {
        // lanes walk rows serially: with few (voice tile, block) items prefer more, narrower waves and a deeper
        // ring (big blocks x few blocks, e.g. N=1024 K=64, would otherwise leave 3/4 of the SIMDs idle)
        while (vpt > 1 && (int64_t)((voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt)) * K < 2048) vpt >>= 1;
        ring = (vpt == 4) ? 8 : 16;
    }
    while (vpt > 1 && !ok(vpt)) vpt >>= 1;
    if (vpt == 1 && !variant) ring = 16;
    const int span = SIG_WAVE * vpt;
    const int voice_tiles = (voices + span - 1) / span;
    const int64_t items = (int64_t)voice_tiles * K;
    const int64_t nwg = (items + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
#define SIG_BQ(V, R) biquad_coldstart_kernel<T, V, R, ENV><<<(unsigned)nwg, 256, 0, stream>>>( \
        type, (double)rate, position, N, K, ctx, voices, cutoff, cs, cutoff_blocks, in, in_ld, out, out_ld, voice_tiles, status, env)
    switch (vpt * 100 + ring) {
        case 108: SIG_BQ(1, 8); break;
        case 116: SIG_BQ(1, 16); break;
        case 132: SIG_BQ(1, 32); break;
        case 208: SIG_BQ(2, 8); break;
        case 216: SIG_BQ(2, 16); break;
        case 232: SIG_BQ(2, 32); break;
        case 408: SIG_BQ(4, 8); break;
        case 416: SIG_BQ(4, 16); break;
        case 432: SIG_BQ(4, 32); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef SIG_BQ
    return sig_launch_status();
}

}  // namespace

namespace {

int run_biquad(int type, int32_t rate, int64_t position, int32_t block_frames, int32_t nblocks, int32_t context,
               int32_t voices, const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
               const void* in, int64_t in_ld, int64_t in_history, void* out, int64_t out_ld, int32_t dtype,
               int32_t* status, void* stream, const sig_env::AdsrRows* env)
{
    SIG_CHECK_ARG(type == SIG_FILT_LOWPASS || type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(cutoff != nullptr && in != nullptr && out != nullptr);
    SIG_CHECK_ARG(cutoff_stride == 0 || cutoff_stride == 1);
    SIG_CHECK_ARG(cutoff_blocks == 1 || cutoff_blocks == nblocks);
    SIG_CHECK_ARG(in_ld >= voices && out_ld >= voices);
    // block 0 reads min(context, position) rows in front of `in`; later blocks reach back into
    // earlier blocks' rows and, when block_frames < context, also into the history.
    {
        const int64_t c0 = position < context ? position : context;
        SIG_CHECK_ARG(in_history >= c0);
    }
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const sig_env::AdsrRows none{};
    if (dtype == SIG_F32) {
        const float* x = static_cast<const float*>(in);
        float* y = static_cast<float*>(out);
        return env ? launch_biquad<float, true>(type, rate, position, block_frames, nblocks, context, voices, cutoff,
                                                cutoff_stride, cutoff_blocks, x, in_ld, y, out_ld, status, s, *env)
                   : launch_biquad<float, false>(type, rate, position, block_frames, nblocks, context, voices, cutoff,
                                                 cutoff_stride, cutoff_blocks, x, in_ld, y, out_ld, status, s, none);
    }
    if (dtype == SIG_F64 && !env)
        return launch_biquad<double, false>(type, rate, position, block_frames, nblocks, context, voices, cutoff,
                                            cutoff_stride, cutoff_blocks, static_cast<const double*>(in), in_ld,
                                            static_cast<double*>(out), out_ld, status, s, none);
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int sig_biquad_coldstart(int type, int32_t rate, int64_t position,
                                    int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                    const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                                    const void* in, int64_t in_ld, int64_t in_history,
                                    void* out, int64_t out_ld, int32_t dtype,
                                    int32_t* status, void* stream)
{
    return run_biquad(type, rate, position, block_frames, nblocks, context, voices, cutoff, cutoff_stride, cutoff_blocks,
                      in, in_ld, in_history, out, out_ld, dtype, status, stream, nullptr);
}

extern "C" int sig_biquad_coldstart_env(int type, int32_t rate, int64_t position,
                                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                        const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                                        const double* const* adsr_params, const int32_t* adsr_strides,
                                        const float* in, int64_t in_ld, int64_t in_history,
                                        float* out, int64_t out_ld, int32_t* status, void* stream)
{
    sig_env::AdsrRows env;
    SIG_CHECK_ARG(sig_env::load_rows(adsr_params, adsr_strides, env));
    return run_biquad(type, rate, position, block_frames, nblocks, context, voices, cutoff, cutoff_stride, cutoff_blocks,
                      in, in_ld, in_history, out, out_ld, SIG_F32, status, stream, &env);
}
