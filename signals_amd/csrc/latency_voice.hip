// Latency mode of the C2 graph (Sine -> LowPass|HighPass -> [x gain] -> [pan] -> bus) for ONE block per launch: a
// real-time sink pulls 256 frames of 1024 voices at a time, which is far too little work to hide a chain of
// launches behind (fused_voice.hip's scan kernel + sum_bus + position advance: ~25 us per block through a hipGraph).
// Here the whole block is one launch with rows x voices parallelism:
//   * closed form of fused_voice.hip ("steady" kernel): y_n = yss_n + yh_n, steady-state sinusoid + homogeneous
//     solution of the cold start at r0 = p - c.  Both can be seeded at ANY row: yss from the phase of that row
//     (sin by the f64 polynomial), yh = [A^(n - r0) (-sss_{r0-1})]_0 with A^k by squaring.  So a lane takes one voice
//     and one chunk of kRows rows: ~600 f64 operations of set-up, then 7 per row.  voices/64 x N/kRows waves.
//   * lanes whose voice the closed form does not cover (below ~8 Hz, above rate/4, past 2^26 cycles) walk from r0
//     to their chunk with the exact per-row phase and the driven recurrence -- slow, rare, same launch.
//   * per-row voice sums through sig_bus::Tile into per-tile f64 partials; the LAST workgroup to finish (a ticket
//     from one atomic counter) adds the tiles in fixed order, writes the float32 bus, re-arms the counter and, when
//     the position lives in device memory (hipGraph replay), advances it by one block.
// Reference: osc.py:26-43, fx.py:85-121, fx.py:49-52 and the build-defined bus; 1e-6 parity like the other fused paths.
#include "sig_biquad.h"
#include "sig_bus_tile.h"
#include "sig_osc.h"

// Memory order of the arrival ticket (agent scope).  __ATOMIC_ACQ_REL is the textbook hand-off (release of this
// workgroup's partials, acquire of everybody else's); __ATOMIC_RELAXED relies on the write-through / sc1-load form that
// MI355X_MICROARCH.md lists under "valid forms" for inter-workgroup hand-offs.  tools/time_latency.py measures both.
#ifndef SIG_LATENCY_TICKET_ORDER
#define SIG_LATENCY_TICKET_ORDER __ATOMIC_ACQ_REL
#endif

namespace {

using sig_biquad::Biquad;
using sig_biquad::Cx;
using sig_biquad::cx_div;
using sig_biquad::cx_mul;
using sig_biquad::design_butter2;

constexpr int kRows = 16;                  // rows per lane (= one or more whole bus groups: 16 / C rows each)

struct M2 { double a, b, c, d; };                                             // [[a, b], [c, d]]
__device__ __forceinline__ M2 m2_mul(const M2& x, const M2& y) {
    return {fma(x.a, y.a, x.b * y.c), fma(x.a, y.b, x.b * y.d), fma(x.c, y.a, x.d * y.c), fma(x.c, y.b, x.d * y.d)};
}

// sin(2 pi f) in f64 (~1 ulp), any |f| < 2^50 (fused_voice.hip)
__device__ __forceinline__ double sin2pi(double f) {
    const double u = fma(f, 2.0, sig_osc::kRoundMagic);
    const double k = u - sig_osc::kRoundMagic;
    const double rq = fma(k, -0.5, f);
    const double y = sig_osc::sin_poly(fma(rq, sig_osc::kTwoPiHi, rq * sig_osc::kTwoPiLo));
    return __hiloint2double(__double2hiint(y) ^ (int)(((unsigned)__double2loint(u) & 1u) << 31), __double2loint(y));
}

struct Args {
    int type; double rate; int64_t position; const int64_t* pos_dev; int N, ctx, voices;
    const double* hertz; int hs; const double* phase; int ps; const double* cutoff; int cs; const double* gain; int gs;
    const double* pan; int64_t pan_ld; double* partials; unsigned* ticket; float* out; int64_t out_ld;
    int tiles, chunks; int* status; int64_t* pos_advance;
};

template <int C>
__global__ __launch_bounds__(256) void latency_voice_bus_kernel(Args a)
{
    constexpr int R = sig_bus::Tile<C>::R;
    static_assert(kRows % R == 0, "a chunk is a whole number of bus groups");
    __shared__ double lds[4][sig_bus::kTileDoubles];
    __shared__ unsigned last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;                                   // (voice tile, chunk)
    const int64_t p = a.pos_dev ? *a.pos_dev : a.position;
    if (item < a.tiles * a.chunks) {
        const int vt = item % a.tiles, chunk = item / a.tiles;
        const int v0 = vt * SIG_WAVE + lane;
        const bool live = v0 < a.voices;
        const int v = live ? v0 : 0;
        const int c = (int)((p < (int64_t)a.ctx) ? p : (int64_t)a.ctx);
        const int64_t r0 = p - c, n0 = p + (int64_t)chunk * kRows;            // cold start row; first row of this chunk
        const int rows = (a.N - chunk * kRows < kRows) ? a.N - chunk * kRows : kRows;

        Biquad q;
        const bool ok = design_butter2(a.type, a.cutoff[(int64_t)v * a.cs], a.rate, q);
        if (!ok && live && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
        const double s2 = (a.type == SIG_FILT_LOWPASS) ? 2.0 : -2.0;           // b1 / b0
        const double a1 = q.a1, a2 = q.a2;
        const double hz = a.hertz[(int64_t)v * a.hs], ph = a.phase ? a.phase[(int64_t)v * a.ps] : 0.0;
        const double scale = a.gain ? q.b0 * a.gain[(int64_t)v * a.gs] : q.b0;
        double w[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = live ? (a.pan ? a.pan[ch * a.pan_ld + v] * scale : scale) : 0.0;

        const double d = hz / a.rate;
        const double dr = d - rint(d);                                         // revolutions per row
        const double st = sin2pi(dr), ct = sin2pi(dr + 0.25), sh = sin2pi(0.5 * dr);
        const double t_r0 = (double)r0 / a.rate * hz + ph, t_n0 = (double)n0 / a.rate * hz + ph;   // osc.py:32
        const double t_end = (double)(p + a.N - 1) / a.rate * hz + ph;
        const bool closed = fabs(t_r0) < sig_osc::kSineFastMaxT && fabs(t_end) < sig_osc::kSineFastMaxT &&
                            fabs(dr) <= 0.25 && fabs(st) >= 1e-3;

        double yss = 0.0, dss = 0.0, z0 = 0.0, z1 = 0.0;                       // closed: yss/dss + homogeneous (z0, z1); else driven (z0, z1)
        const double nm = -4.0 * sh * sh;
        if (closed) {
            const Cx z = {ct, -st};                                            // e^{-j theta}
            const Cx zz = cx_mul(z, z);
            const Cx H = cx_div({1.0 + s2 * z.re + zz.re, s2 * z.im + zz.im}, {1.0 + a1 * z.re + a2 * zz.re, a1 * z.im + a2 * zz.im});
            // steady-state oscillator at n0
            const double f0 = t_n0 - rint(t_n0);
            const Cx wn = cx_mul(H, {sin2pi(f0 + 0.25), sin2pi(f0)});
            yss = wn.im;
            dss = fma(wn.re, st, wn.im * (-2.0 * sh * sh));
            // homogeneous state at n0 - 1:  A^(n0 - r0) (-sss_{r0-1}),  sss_{r0-1} = (Im(P u), Im(Q u)),  u = e^{j phi(r0)}
            const double fr = t_r0 - rint(t_r0);
            const Cx u = {sin2pi(fr + 0.25), sin2pi(fr)};
            const Cx P = {H.re - 1.0, H.im};
            const Cx Pe = cx_mul(P, {ct, st});
            const Cx Q = {Pe.re - s2 + a1 * H.re, Pe.im + a1 * H.im};
            const double s0 = -cx_mul(P, u).im, s1 = -cx_mul(Q, u).im;
            M2 Ak = {1.0, 0.0, 0.0, 1.0}, Ap = {-a1, 1.0, -a2, 0.0};
            for (int64_t e = n0 - r0; e > 0; e >>= 1) {
                if (e & 1) Ak = m2_mul(Ak, Ap);
                Ap = m2_mul(Ap, Ap);
            }
            z0 = fma(Ak.a, s0, Ak.b * s1);
            z1 = fma(Ak.c, s0, Ak.d * s1);
        } else {
            for (int64_t n = r0; n < n0; ++n) {                                // the plain way, from the cold start to this chunk
                const double x = (double)sig_osc::osc_sine_f32((double)n / a.rate * hz + ph);
                const double y = x + z0;
                z0 = fma(-a1, y, fma(s2, x, z1));
                z1 = fma(-a2, y, x);
            }
        }

        sig_bus::Tile<C> bus(lds[wave], lane, a.partials + (int64_t)vt * a.N * C);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            double y;
            if (closed) {
                y = yss + z0;
                const double yh = z0;
                z0 = fma(-a1, yh, z1);
                z1 = -a2 * yh;
                yss += dss;
                dss = fma(nm, yss, dss);
            } else {
                const double x = (double)sig_osc::osc_sine_f32((double)(n0 + k) / a.rate * hz + ph);
                y = x + z0;
                z0 = fma(-a1, y, fma(s2, x, z1));
                z1 = fma(-a2, y, x);
            }
            double acc[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = w[ch] * y;
            bus.put(k % R, acc);
            if (k % R == R - 1) {
                const int lo = k - (R - 1);
                if (lo < rows) bus.template flush<true>((int64_t)chunk * kRows + lo, 0, (rows - lo < R) ? rows - lo : R);
            }
        }
    }
    // ---- the last workgroup to arrive adds the voice tiles (fixed order) and finishes the block.  Hand-off without an
    // L2 write-back fence (MI355X_MICROARCH.md, valid forms): every partial was stored write-through (sc1) and is drained
    // (vmcnt(0)) before the workgroup's barrier; one lane then takes a ticket with an agent-scope atomic; the last
    // workgroup reads the partials with sc1 loads, which do not look at its CU's L1.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        last_flag = (__hip_atomic_fetch_add(a.ticket, 1u, SIG_LATENCY_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    const int n = a.N * C;
    for (int i = threadIdx.x; i < n; i += 256) {
        double s = 0.0;
        for (int t = 0; t < a.tiles; ++t)                                       // fixed order
            s += __hip_atomic_load(a.partials + (int64_t)t * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.out[(int64_t)(i / C) * a.out_ld + (i % C)] = (float)s;
    }
    if (threadIdx.x == 0) {
        *a.ticket = 0u;                                                        // re-armed for the next launch (stream order)
        if (a.pos_advance) *a.pos_advance = p + a.N;
    }
}

template <int C>
int launch(Args a, hipStream_t stream)
{
    a.tiles = (a.voices + SIG_WAVE - 1) / SIG_WAVE;
    a.chunks = (a.N + kRows - 1) / kRows;
    const int nwg = (a.tiles * a.chunks + 3) / 4;
    latency_voice_bus_kernel<C><<<nwg, 256, 0, stream>>>(a);
    return sig_launch_status();
}

}  // namespace

extern "C" int64_t sig_latency_voice_bus_workspace(int32_t voices, int32_t block_frames, int32_t bus_channels)
{
    const int64_t tiles = (voices + SIG_WAVE - 1) / SIG_WAVE;
    return (tiles * block_frames * bus_channels + 1) * (int64_t)sizeof(double);    // partials + the ticket word
}

extern "C" int sig_latency_voice_bus(int filt_type, int32_t rate, int64_t position, int64_t* position_dev,
                                     int32_t block_frames, int32_t context, int32_t voices,
                                     const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                     const double* cutoff, int32_t cutoff_stride,
                                     const double* gain, int32_t gain_stride,
                                     const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                     double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(filt_type == SIG_FILT_LOWPASS || filt_type == SIG_FILT_HIGHPASS);
    SIG_CHECK_ARG(rate > 0 && (position_dev || position >= 0) && block_frames >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(hertz && cutoff && out && workspace && out_ld >= bus_channels);
    SIG_CHECK_ARG((hertz_stride | 1) == 1 && (phase_stride | 1) == 1 && (cutoff_stride | 1) == 1 && (gain_stride | 1) == 1);
    SIG_CHECK_ARG(bus_gains ? bus_gains_ld >= voices : bus_channels == 1);
    if (block_frames == 0 || voices == 0) return 0;
    const int64_t tiles = (voices + SIG_WAVE - 1) / SIG_WAVE;
    Args a{filt_type, (double)rate, position, position_dev, block_frames, context, voices,
           hertz, hertz_stride, phase, phase_stride, cutoff, cutoff_stride, gain, gain_stride,
           bus_gains, bus_gains_ld, workspace,
           reinterpret_cast<unsigned*>(workspace + tiles * block_frames * bus_channels), out, out_ld, 0, 0, status, position_dev};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (bus_channels) {
        case 1: return launch<1>(a, s);
        case 2: return launch<2>(a, s);
        case 4: return launch<4>(a, s);
    }
    return (int)hipErrorInvalidValue;
}
