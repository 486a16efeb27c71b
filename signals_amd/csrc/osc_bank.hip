// Oscillator bank for gfx950: closed-form in absolute position, f64 phase, f32 (or f64) store.
// Replaces Osc._eval + Sine/Square/Sawtooth/Triangle._osc (reference src/signals/chain/osc.py:26-62).
//
// Mapping: one wave = 64*VEC consecutive voices x 16 consecutive rows; a 256-thread workgroup
// stacks 4 waves in time (64 rows).  Lane l owns voices [VEC*l, VEC*l+VEC) of the wave's span, so a
// row is stored as 64 x 16-B lanes = 1 KiB, fully coalesced.  The per-row quotient n/rate (an IEEE
// f64 divide, the expensive op) is computed ONCE per row by lanes 0..15 and broadcast with
// v_readlane, so the per-sample cost is mul + add + waveform.
//
// Roofline: 4 B written per voice-sample (f32), no reads beyond 2 x 8 B per voice per wave.
#include "sig_osc.h"

namespace {

using sig_osc::osc_wave;

constexpr int kRowsPerWave = 16;
constexpr int kWavesPerWg = 4;

// Parameter rows: hertz/phase are (1|P, V|1) f64.  `rpp` (rows per parameter row) = 0: one row for the whole
// launch; otherwise output row r reads parameter row r / rpp -- rpp = block_frames for an audio-rate launch whose
// control inputs change per block (forward_at_block_rate, osc.py:28-30), rpp = 1 with `step` = block_frames for
// a block-RATE launch (one output row per block: what a control port sees for K consecutive blocks).
struct OscArgs {
    int64_t position, step; double rate; int64_t rows; int voices;
    const double* hertz; int hs; int64_t hrs; const double* phase; int ps; int64_t prs; int rpp;
};

template <int KIND, int VEC, typename OUT>
__global__ __launch_bounds__(256) void osc_bank_kernel(OscArgs a, OUT* __restrict__ out, int64_t ld, int voice_tiles)
{
    // 1-D grid: consecutive workgroups cover adjacent voice tiles of the same 64 rows
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vt = blockIdx.x % voice_tiles;
    const int64_t rt = blockIdx.x / voice_tiles;
    const int v0 = (vt * SIG_WAVE + lane) * VEC;
    const int64_t r0 = (rt * kWavesPerWg + wave) * kRowsPerWave;
    if (r0 >= a.rows) return;                                      // wave-uniform

    // osc.py:32  frame_range / rate : int64 -> f64, IEEE divide, one per row
    const double q_lane = (double)(a.position + (r0 + (lane & (kRowsPerWave - 1))) * a.step) / a.rate;

    double hz[VEC], ph[VEC];
    int64_t loaded = -1;                                           // parameter row currently in registers
#pragma unroll 2                                                    // keep the loop body inside the I-cache
    for (int j = 0; j < kRowsPerWave; ++j) {
        const int64_t row = r0 + j;
        if (row >= a.rows) break;                                  // wave-uniform
        const int64_t prow = a.rpp ? row / a.rpp : 0;              // wave-uniform
        if (prow != loaded) {
            loaded = prow;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int v = v0 + i;
                hz[i] = (v < a.voices) ? a.hertz[prow * a.hrs + (int64_t)v * a.hs] : 0.0;
                ph[i] = (v < a.voices && a.phase) ? a.phase[prow * a.prs + (int64_t)v * a.ps] : 0.0;
            }
        }
        const double q = sig_readlane_f64(q_lane, j);
        OUT y[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const double t = q * hz[i] + ph[i];                    // two roundings, like numpy
            y[i] = osc_wave<KIND, OUT>(t);
        }
        OUT* dst = out + row * ld + v0;
        if (VEC == 4) {
            if (v0 < a.voices) {                                   // voices % 4 == 0 on this path
                typename sig_vec4<OUT>::type o;
                o.x = y[0]; o.y = y[1]; o.z = y[2]; o.w = y[3];
                *reinterpret_cast<typename sig_vec4<OUT>::type*>(dst) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (v0 + i < a.voices) dst[i] = y[i];
        }
    }
}

template <int KIND, typename OUT>
int launch_osc(const OscArgs& a, OUT* out, int64_t ld, hipStream_t stream)
{
    const bool vec4 = (a.voices % 4 == 0) && (ld % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) % (4 * sizeof(OUT))) == 0);
    const int64_t rows_per_wg = (int64_t)kRowsPerWave * kWavesPerWg;
    const int64_t row_tiles = (a.rows + rows_per_wg - 1) / rows_per_wg;
    const int span = SIG_WAVE * (vec4 ? 4 : 1);
    const int voice_tiles = (a.voices + span - 1) / span;
    const int64_t nwg = row_tiles * voice_tiles;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (vec4)
        osc_bank_kernel<KIND, 4, OUT><<<(unsigned)nwg, 256, 0, stream>>>(a, out, ld, voice_tiles);
    else
        osc_bank_kernel<KIND, 1, OUT><<<(unsigned)nwg, 256, 0, stream>>>(a, out, ld, voice_tiles);
    return sig_launch_status();
}

template <typename OUT>
int dispatch_kind(int kind, const OscArgs& a, OUT* out, int64_t ld, hipStream_t stream)
{
    switch (kind) {
        case SIG_OSC_SINE: return launch_osc<SIG_OSC_SINE, OUT>(a, out, ld, stream);
        case SIG_OSC_SQUARE: return launch_osc<SIG_OSC_SQUARE, OUT>(a, out, ld, stream);
        case SIG_OSC_SAWTOOTH: return launch_osc<SIG_OSC_SAWTOOTH, OUT>(a, out, ld, stream);
        case SIG_OSC_TRIANGLE: return launch_osc<SIG_OSC_TRIANGLE, OUT>(a, out, ld, stream);
    }
    return (int)hipErrorInvalidValue;
}

int run_osc(int kind, const OscArgs& a, void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == SIG_F32) return dispatch_kind<float>(kind, a, static_cast<float*>(out), out_ld, s);
    if (out_dtype == SIG_F64) return dispatch_kind<double>(kind, a, static_cast<double*>(out), out_ld, s);
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int sig_osc_bank(int kind, int64_t position, int32_t rate, int64_t rows, int32_t voices,
                            const double* hertz, int32_t hertz_stride,
                            const double* phase, int32_t phase_stride,
                            void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && voices >= 0 && rate > 0 && position >= 0);
    SIG_CHECK_ARG(hertz != nullptr && out != nullptr && out_ld >= voices);
    SIG_CHECK_ARG((hertz_stride == 0 || hertz_stride == 1) && (phase_stride == 0 || phase_stride == 1));
    if (rows == 0 || voices == 0) return 0;
    const OscArgs a{position, 1, (double)rate, rows, voices, hertz, hertz_stride, 0, phase, phase_stride, 0, 0};
    return run_osc(kind, a, out, out_dtype, out_ld, stream);
}

extern "C" int sig_osc_bank_mod(int kind, int64_t position, int64_t position_step, int32_t rate, int64_t rows,
                                int32_t voices, int32_t rows_per_param,
                                const double* hertz, int32_t hertz_stride, int64_t hertz_row_stride,
                                const double* phase, int32_t phase_stride, int64_t phase_row_stride,
                                void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && voices >= 0 && rate > 0 && position >= 0 && position_step >= 1 && rows_per_param >= 0);
    SIG_CHECK_ARG(hertz != nullptr && out != nullptr && out_ld >= voices);
    SIG_CHECK_ARG((hertz_stride == 0 || hertz_stride == 1) && (phase_stride == 0 || phase_stride == 1));
    SIG_CHECK_ARG(hertz_row_stride >= 0 && phase_row_stride >= 0);
    if (rows == 0 || voices == 0) return 0;
    const OscArgs a{position, position_step, (double)rate, rows, voices, hertz, hertz_stride, hertz_row_stride,
                    phase, phase_stride, phase_row_stride, rows_per_param};
    return run_osc(kind, a, out, out_dtype, out_ld, stream);
}

extern "C" int sig_abi_version(void) { return SIG_ABI_VERSION; }
