// ADSR envelope bank for gfx950 (build-defined node; the reference has only a dead
// sketch, src/signals/sig.py:89-100).  Position-pure piecewise-linear envelope per voice at frame rate:
//   t = n / rate;  u = t - gate_on;  v = u - attack;  w = t - gate_off
//   held(t) = 0                                   (u < 0)
//           = clip(u * (1/attack), 0, 1)          (v < 0)
//           = 1 + (sustain - 1) * clip(v * (1/decay), 0, 1)      (a zero-length stage counts as complete)
//   level(t) = held(t)                            (w < 0)
//            = held(gate_off) * clip(1 - w * (1/release), 0, 1)  (release == 0: 0)
// Definition and operator order are those of oracle/chain_ref.py:adsr (f64, contract off), so the f32 store
// is bit-exact against it.  HBM-write-bound: 4 B per voice-sample, ~12 f64 ops.
#include "sig_adsr.h"

namespace {

using namespace sig_env;

constexpr int kRowsPerWave = 16;

// MUL: out = envelope * x (the RingMod(x, ADSR) pair of BASELINE config 3 in one pass: 8 B per voice-sample
// instead of 4 (envelope store) + 12 (RingMod)); x is f32 audio aligned with `out`.
template <int VEC, typename OUT, bool MUL>
__global__ __launch_bounds__(256) void adsr_kernel(int64_t position, double rate, int64_t rows, int voices, AdsrRows in,
                                                   OUT* __restrict__ out, int64_t ld, int voice_tiles,
                                                   const float* __restrict__ x, int64_t x_ld)
{
    const int lane = threadIdx.x & 63;
    const int vt = blockIdx.x % voice_tiles;
    const int64_t rt = blockIdx.x / voice_tiles;
    const int v0 = (vt * SIG_WAVE + lane) * VEC;
    const int64_t r0 = (rt * 4 + (threadIdx.x >> 6)) * kRowsPerWave;
    if (r0 >= rows) return;
    const double q_lane = (double)(position + r0 + (lane & (kRowsPerWave - 1))) / rate;
    Voice p[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = load_voice(in, (v0 + i < voices) ? v0 + i : 0);
#pragma unroll 2
    for (int j = 0; j < kRowsPerWave; ++j) {
        const int64_t row = r0 + j;
        if (row >= rows) break;
        const double t = sig_readlane_f64(q_lane, j);
        OUT y[VEC];
        float xin[VEC];
        if constexpr (MUL && VEC == 4) {
            const float4 v = (v0 < voices) ? *reinterpret_cast<const float4*>(x + row * x_ld + v0) : make_float4(0, 0, 0, 0);
            xin[0] = v.x; xin[1] = v.y; xin[2] = v.z; xin[3] = v.w;
        } else if constexpr (MUL) {
            xin[0] = (v0 < voices) ? x[row * x_ld + v0] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const double env = level(p[i], t);
            y[i] = (OUT)(MUL ? env * (double)xin[i] : env);
        }
        OUT* dst = out + row * ld + v0;
        if (VEC == 4) {
            if (v0 < voices) {
                typename sig_vec4<OUT>::type o;
                o.x = y[0]; o.y = y[1]; o.z = y[2]; o.w = y[3];
                *reinterpret_cast<typename sig_vec4<OUT>::type*>(dst) = o;
            }
        } else {
            if (v0 < voices) dst[0] = y[0];
        }
    }
}

template <typename OUT>
int launch_adsr(int64_t position, int32_t rate, int64_t rows, int32_t voices, const AdsrRows& in, OUT* out, int64_t ld,
                const float* x, int64_t x_ld, hipStream_t stream)
{
    const bool vec4 = (voices % 4 == 0) && (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % (4 * sizeof(OUT)) == 0) &&
                      (!x || (x_ld % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0));
    const int span = SIG_WAVE * (vec4 ? 4 : 1);
    const int voice_tiles = (voices + span - 1) / span;
    const int64_t nwg = ((rows + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave)) * voice_tiles;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
#define SIG_ADSR(V, M) adsr_kernel<V, OUT, M><<<(unsigned)nwg, 256, 0, stream>>>(position, (double)rate, rows, voices, in, \
                                                                                    out, ld, voice_tiles, x, x_ld)
    if (vec4) { if (x) SIG_ADSR(4, true); else SIG_ADSR(4, false); }
    else      { if (x) SIG_ADSR(1, true); else SIG_ADSR(1, false); }
#undef SIG_ADSR
    return sig_launch_status();
}

}  // namespace

extern "C" int sig_adsr(int64_t position, int32_t rate, int64_t rows, int32_t voices,
                        const double* const* params, const int32_t* strides,
                        void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(position >= 0 && rate > 0 && rows >= 0 && voices >= 0 && params && strides && out && out_ld >= voices);
    AdsrRows in;
    SIG_CHECK_ARG(load_rows(params, strides, in));
    if (rows == 0 || voices == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == SIG_F32) return launch_adsr<float>(position, rate, rows, voices, in, static_cast<float*>(out), out_ld, nullptr, 0, s);
    if (out_dtype == SIG_F64) return launch_adsr<double>(position, rate, rows, voices, in, static_cast<double*>(out), out_ld, nullptr, 0, s);
    return (int)hipErrorInvalidValue;
}

extern "C" int sig_adsr_apply(int64_t position, int32_t rate, int64_t rows, int32_t voices,
                              const double* const* params, const int32_t* strides,
                              const float* x, int64_t x_ld, float* out, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(position >= 0 && rate > 0 && rows >= 0 && voices >= 0 && params && strides && x && out);
    SIG_CHECK_ARG(out_ld >= voices && x_ld >= voices);
    AdsrRows in;
    SIG_CHECK_ARG(load_rows(params, strides, in));
    if (rows == 0 || voices == 0) return 0;
    return launch_adsr<float>(position, rate, rows, voices, in, out, out_ld, x, x_ld, static_cast<hipStream_t>(stream));
}
