// A whole frame-rate voice graph in ONE launch, gfx950: the general form of the fused voice kernels.
//
// The fused kernels of fused_voice.hip / fused_cascade.hip each cover one graph shape (Filter(Osc), Filter(Filter(Osc)) ...).
// Everything else -- an Amp or a Mix behind a filter, RingMod of two filtered voices, three filters in series, block-rate
// FM together with a second oscillator, blocks shorter than the filter context -- ran one kernel per node (24+ B per
// voice-sample through HBM) or, for short blocks, the eager pull path (~150 us per block).  Here the per-voice graph is
// compiled on the host into a short straight-line program for an ACCUMULATOR MACHINE and interpreted per row by every
// lane for its own voices (lanes = voices, like the walkers): the accumulator and a few temporaries live in VGPRs, the
// program word of instruction pc sits in lane pc of one VGPR and is fetched with v_readlane (a wave-uniform scalar), the
// dispatch is a scalar compare chain.  State arrays (filter slots, oscillator slots, parameter registers, temporaries)
// are indexed by a wave-uniform slot number through a compile-time switch, so they stay in registers.
//
// What makes one machine enough for every shape is the reference's block structure, reproduced as a sequence of blocks:
//   * a filter answers a block from ZERO state over [<=100 context rows | block] (CritFilter._filter, fx.py:85-106);
//   * with blocks longer than the context, the context rows in front of block j are the input's rows of block j - 1: the
//     input keeps its previous block cached and the context request is a slice of it (BlockCachingEmitter,
//     chain/__init__.py:431-442) -- for a filter input that block was itself cold-started 100 rows before block j - 1;
//   * every node reads its control ports once per block, at the block's position (forward_at_block_rate,
//     chain/__init__.py:305-306).
// So every filter slot carries TWO recurrences: the CURRENT block's chain, and -- over the last min(100, .) rows of a
// block -- the NEXT block's chain, cold-started there on the same input (the current block's values, which is exactly
// what the cache serves as context) with the next block's design.  At a block boundary next becomes current.  A lane
// that starts its span at block b first re-walks the D - 1 blocks in front of it (D = filters in series) plus 100 rows:
// a filter at level l of the cascade is exact from block b - D + l on (induction over levels), so level D is exact from
// block b.  In front of the launch those are the previous render's blocks, or on a fresh graph the virtual 100-row
// blocks the reference's context requests create ([p - 100, p) answered as a block of its own, controls read at p - 100).
// Blocks SHORTER than the context (PortAudio callbacks of 32 or 64 frames): a context request then lies in no single
// cached block, and the reference answers it as a block of its own for EVERY block -- each output block is walked on its
// own behind its virtual block (controls read at max(p - 100, 0)); two filters in series at most in that mode.
//
// Arithmetic: f64 throughout, exact per-row phase t = n / rate * hertz + phase (osc.py:32, one IEEE divide per row shared
// by the wave), Butterworth design per block in-kernel (sig_biquad.h), b0-normalised DF2T recurrence (4 FMAs + 1).  No
// float32 rounding between nodes: closer to the f64 reference than the per-node schedule; 1e-6 parity by test.
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "sig_adsr.h"
#include "sig_biquad.h"
#include "sig_bus_tile.h"
#include "sig_osc.h"

namespace {

using sig_biquad::Biquad;
using sig_bus::kTileStride;

constexpr int kMaxIns = SIG_VP_MAX_INS;
#ifndef SIG_VP_ROWS
#define SIG_VP_ROWS 8
#endif
constexpr int kRowGroup = SIG_VP_ROWS;     // the interpreter runs every instruction for this many consecutive rows at a time: its dispatch
                                           // (fetch, decode, two scalar switches) is paid once per group, and a handler has rows x voices
                                           // independent evaluations to overlap

struct Rows { const double* ptr; int cs; int rows; };      // (rows, V | 1) float64: rows == 1 holds for every block

struct VpArgs {
    double rate; int64_t position; int N, K, ctx, voices;
    int n_ins; uint32_t code[kMaxIns];                       // op | kind << 5 | a << 8 | b << 12 | c << 16
    int n_oscs; Rows hertz[SIG_VP_MAX_OSCS], phase[SIG_VP_MAX_OSCS];
    int n_params; Rows params[SIG_VP_MAX_PARAMS];
    int n_filters; Rows cutoff[SIG_VP_MAX_FILTERS]; int ftype[SIG_VP_MAX_FILTERS]; int flevel[SIG_VP_MAX_FILTERS];
    sig_env::AdsrRows adsr; int has_adsr;
    uint64_t seeds[2];
    int depth, hist, small, blocks_before; int64_t hist_pos[SIG_VP_MAX_HIST];
    int span, voice_tiles;
    float* out; int64_t out_ld;                              // store sink
    const double* pan; int64_t pan_ld; double* partials; int64_t rows; float* bus_out; int64_t bus_out_ld;   // bus sink
    int* status;
};

// f(integral_constant<int, i>) for a wave-uniform i < N: a scalar switch, so arrays indexed inside stay in registers
template <int N, typename F>
__device__ __forceinline__ void with_index(int i, F&& f) {
    switch (i) {
        case 0: if constexpr (N > 0) f(std::integral_constant<int, 0>{}); break;
        case 1: if constexpr (N > 1) f(std::integral_constant<int, 1>{}); break;
        case 2: if constexpr (N > 2) f(std::integral_constant<int, 2>{}); break;
        case 3: if constexpr (N > 3) f(std::integral_constant<int, 3>{}); break;
        case 4: if constexpr (N > 4) f(std::integral_constant<int, 4>{}); break;
        case 5: if constexpr (N > 5) f(std::integral_constant<int, 5>{}); break;
        case 6: if constexpr (N > 6) f(std::integral_constant<int, 6>{}); break;
        case 7: if constexpr (N > 7) f(std::integral_constant<int, 7>{}); break;
        default: break;
    }
}

// An opaque copy of a wave-uniform double.  A handler's arithmetic on the row's n / rate and the slot's block-invariant
// registers does not depend on the instruction counter, so the optimiser hoists it out of the instruction loop and evaluates
// EVERY case of every slot once per row, unconditionally (measured: all four waveforms of all oscillator slots per row, ~10x
// the work of the program itself and twice the registers).  Passing an input through an empty volatile asm pins the work to
// the case that needs it.
__device__ __forceinline__ double vp_pin(double x) {
    asm volatile("" : "+s"(x));
    return x;
}

__device__ __forceinline__ uint64_t vp_mix64(uint64_t z) {                     // noise.hip
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// sig_biquad.h's design_butter2 with a light tangent: libm's tan() carries a Payne-Hanek reduction for huge arguments whose
// temporaries set the register budget of the whole kernel (+110 VGPRs, measured), and the argument here is pi Wn / 2 with
// Wn in (0, 1): tan = sin / cos from the odd polynomial of sig_osc.h on [0, pi/2], cos(x) = sin(pi/2 - x) with pi/2 as hi + lo.
// Coefficients within 1e-15 (relative) of the libm form.
__device__ __forceinline__ bool vp_design(int type, double cutoff, double rate, Biquad& q) {
    double wn = cutoff / (rate * 0.5);                      // scaled_crit /= rate / 2  (fx.py:99-101)
    wn = (wn < 0.0) ? 0.0 : ((wn > 1.0) ? 1.0 : wn);
    const bool bad = !(wn > 0.0 && wn < 1.0);               // scipy raises (NaN too)
    const double x = sig_biquad::kPi * wn / 2.0;
    const double xc = (1.5707963267948966 - x) + 6.123233995736766e-17;
    const double k = sig_osc::sin_poly(x) / sig_osc::sin_poly(xc);
    const double k2 = k * k;
    const double nrm = 1.0 / (1.0 + sig_biquad::kSqrt2 * k + k2);
    q.b0 = (type == SIG_FILT_LOWPASS) ? k2 * nrm : nrm;
    q.a1 = 2.0 * (k2 - 1.0) * nrm;
    q.a2 = (1.0 - sig_biquad::kSqrt2 * k + k2) * nrm;
    if (bad) { q.b0 = q.a1 = q.a2 = __builtin_nan(""); }
    q.b1 = q.b2 = 0.0;                                      // (the recurrence is b0-normalised: b = b0 [1, +-2, 1])
    return !bad;
}

__device__ __noinline__ double vp_amp(double x, double e) { return copysign(pow(x, e), x); }   // fx.py:60 (kept out of line: pow is long)

template <int VPT> struct VpOut;
template <> struct VpOut<1> { using type = float; };
template <> struct VpOut<2> { using type = float2; };
template <> struct VpOut<4> { using type = float4; };
__device__ __forceinline__ void vp_put(float& v, const float (&y)[1]) { v = y[0]; }
__device__ __forceinline__ void vp_put(float2& v, const float (&y)[2]) { v = make_float2(y[0], y[1]); }
__device__ __forceinline__ void vp_put(float4& v, const float (&y)[4]) { v = make_float4(y[0], y[1], y[2], y[3]); }

// Register-file sizes per variant (the host picks the smallest variant a program fits).  SMALL: two filter slots, three
// oscillator slots, four parameter registers, one temporary (a temporary is a whole row group: 8 rows x 2 voices = 32 VGPRs), no Amp / ADSR / White -- the common synthesiser voice; its
// state fits two waves per SIMD at two voices per lane, which the interpreter's scalar dispatch needs to hide its branches.
// The full register file (four filters, four oscillators, eight parameters, four temporaries, every instruction) runs at one.
template <bool SMALL> struct VpLimits {
#ifndef SIG_VP_S_NF
#define SIG_VP_S_NF 2
#define SIG_VP_S_NO 3
#define SIG_VP_S_NP 4
#define SIG_VP_S_NT 1
#endif
    static constexpr int NF = SMALL ? SIG_VP_S_NF : 4, NO = SMALL ? SIG_VP_S_NO : 4, NP = SMALL ? SIG_VP_S_NP : 8, NT = SMALL ? SIG_VP_S_NT : 4;
#ifdef SIG_VP_S_EXT
    static constexpr bool EXT = !SMALL || (SIG_VP_S_EXT != 0);
#else
    static constexpr bool EXT = !SMALL;
#endif
};

// C == 0: store (float) acc to a.out; C > 0: C bus channels into a.partials.  EXT: Amp, ADSR and White instructions.
template <int VPT, bool SMALL, int C>
__device__ __forceinline__ void vp_wave(const VpArgs& a, double* tile, int lane, int wave)
{
    constexpr int RG = SMALL ? kRowGroup : kRowGroup / 2;                      // rows per instruction dispatch (the full register file: four temporaries of a row group each)
    constexpr bool BUS = C > 0;
    constexpr int CC = BUS ? C : 1;
    using L = VpLimits<SMALL>;
    constexpr int NF = L::NF, NO = L::NO, NP = L::NP, NT = L::NT;
    constexpr bool EXT = L::EXT;
    using Vec = typename VpOut<VPT>::type;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int vt = (int)(item % a.voice_tiles);
    const int64_t b_first = (item / a.voice_tiles) * a.span;
    if (b_first >= a.K) return;                                               // wave-uniform
    const int nb = (int)((a.K - b_first < (int64_t)a.span) ? a.K - b_first : (int64_t)a.span);
    const int v0 = (vt * SIG_WAVE + lane) * VPT;
    const bool live0 = v0 < a.voices;
    const int vc = live0 ? v0 : 0;
    auto voice = [&](int i) { return (v0 + i < a.voices) ? v0 + i : vc; };     // dead voices shadow a live one (weight 0 / not stored)

    // the program: word pc in lane pc
    uint32_t codev = 0;
#pragma unroll
    for (int k = 0; k < kMaxIns; ++k) codev = (lane == k) ? a.code[k] : codev;

    double acc[RG][VPT], T[NT > 0 ? NT : 1][RG][VPT], pr[NP][VPT], ohz[NO][VPT], oph[NO][VPT];
    double z0[NF][VPT], z1[NF][VPT], w0[NF][VPT], w1[NF][VPT], na1[NF][VPT], na2[NF][VPT], fb0[NF][VPT], xa1[NF][VPT], xa2[NF][VPT];
    double s2[NF];
    double wt[CC][VPT];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        s2[f] = (f < a.n_filters && a.ftype[f] == SIG_FILT_HIGHPASS) ? -2.0 : 2.0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { z0[f][i] = z1[f][i] = w0[f][i] = w1[f][i] = 0.0; na1[f][i] = na2[f][i] = fb0[f][i] = xa1[f][i] = xa2[f][i] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int i = 0; i < VPT; ++i) T[k][r][i] = 0.0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
#pragma unroll
        for (int r = 0; r < RG; ++r) acc[r][i] = 0.0;
#pragma unroll
        for (int ch = 0; ch < CC; ++ch) wt[ch][i] = BUS ? ((v0 + i < a.voices) ? (a.pan ? a.pan[ch * a.pan_ld + voice(i)] : 1.0) : 0.0) : 1.0;
    }

    // ---- per-block state: parameter registers, oscillator rows, filter designs
    auto row_at = [&](const Rows& r, int64_t cri, int v) {
        return r.ptr[(r.rows > 1 ? cri * (int64_t)(r.cs ? a.voices : 1) : 0) + (int64_t)v * r.cs];
    };
    bool params_loaded = false;
    auto load_params = [&](int64_t cri) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (k < a.n_params && (!params_loaded || a.params[k].rows > 1)) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) pr[k][i] = row_at(a.params[k], cri, voice(i));
            }
#pragma unroll
        for (int k = 0; k < NO; ++k)
            if (k < a.n_oscs) {
                if (!params_loaded || a.hertz[k].rows > 1) {
#pragma unroll
                    for (int i = 0; i < VPT; ++i) ohz[k][i] = row_at(a.hertz[k], cri, voice(i));
                }
                if (!params_loaded || a.phase[k].rows > 1) {
#pragma unroll
                    for (int i = 0; i < VPT; ++i) oph[k][i] = a.phase[k].ptr ? row_at(a.phase[k], cri, voice(i)) : 0.0;
                }
            }
        params_loaded = true;
    };
    bool designed = false;                                                     // block-invariant designs are made once
    auto design = [&](int64_t cri, int min_level, auto next_tag) {
        constexpr bool NEXT = decltype(next_tag)::value;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (f < a.n_filters && a.flevel[f] >= min_level && (!designed || a.cutoff[f].rows > 1)) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    __builtin_amdgcn_sched_barrier(0);                         // one design at a time: four interleaved tan() set the kernel's register budget
                    Biquad q;
                    ok &= vp_design(a.ftype[f], row_at(a.cutoff[f], cri, voice(i)), a.rate, q) || !(v0 + i < a.voices);
                    if (NEXT) { xa1[f][i] = -q.a1; xa2[f][i] = -q.a2; }
                    else { na1[f][i] = -q.a1; na2[f][i] = -q.a2; fb0[f][i] = q.b0; }
                    if (!NEXT && !designed) { xa1[f][i] = -q.a1; xa2[f][i] = -q.a2; }
                }
                if (!ok && a.status) atomicOr(a.status, SIG_STATUS_BAD_CUTOFF);
            }
    };
    auto start_next = [&](int64_t cri, int min_level) {                        // the next block's chains (of filters at that level of a cascade or deeper): zero state, its design
        design(cri, min_level, std::true_type{});
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (f < a.n_filters && a.flevel[f] >= min_level) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) { w0[f][i] = 0.0; w1[f][i] = 0.0; }
            }
    };
    auto enter_block = [&](int64_t cri) {                                      // next becomes current
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < VPT; ++i) { z0[f][i] = w0[f][i]; z1[f][i] = w1[f][i]; }
        design(cri, 0, std::false_type{});
        designed = true;
    };

    // ---- envelope (EXT): the voice's current linear stage, re-derived where a run of rows starts and where a stage ends
    [[maybe_unused]] sig_env::Segment seg[EXT ? VPT : 1];
    auto seed_envelope = [&](double t) {
        if constexpr (EXT) {
            if (a.has_adsr) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) seg[i] = sig_env::segment_at(sig_env::load_voice(a.adsr, voice(i)), t);
            }
        }
    };

    float* dst = BUS ? nullptr : a.out + vc;
    double* dstp = BUS ? a.partials + (int64_t)vt * a.rows * CC : nullptr;     // [tile][row][c]
    sig_bus::PipelinedTile<CC> stage(tile, lane, dstp, b_first * a.N);

    // ---- the block sequence of this span (see the header) as a list of STEPS: [load parameters] [next -> current] [start the
    // next block's chains] then rows [from, to).  One loop, so that the row interpreter exists once in the code.
    const int D = a.depth, H = a.hist;
    const int ctx = (D > 0) ? a.ctx : 0;                                       // no filter: no context rows at all
    // `warm_from`: the row from which the NEXT block's chains run too (their zero state and design are set up when the walk gets
    // there); == to: none in this step
    struct Step { bool load, enter, out, next0; int next_level; int64_t cri_load, cri_enter, cri_next, from, to, warm_from; };   // next0: the next chains start with the step, even an empty one
    auto bpos = [&](int64_t j) -> int64_t { return j >= 0 ? a.position + j * (int64_t)a.N : a.hist_pos[H + j]; };   // j >= -H
    auto cri = [&](int64_t j) -> int64_t { return j + H + 1; };                // control rows: [one in front | H blocks in front | K blocks]
    int64_t j0 = b_first - (D > 1 ? D - 1 : 0);
    if (j0 < -(int64_t)H) j0 = -(int64_t)H;
    const int64_t j_end = b_first + nb;
    int n_steps;
    if (!a.small) n_steps = 1 + (int)(j_end - j0);
    else n_steps = (D >= 2) ? 4 : 2;
    auto make_step = [&](int t) {
        Step st{false, false, false, false, 0, 0, 0, 0, 0, 0, 0};
        if (!a.small) {
            if (t == 0) {                                                      // the rows in front of the first block belong to the block before it
                const int64_t s0 = bpos(j0);
                const int64_t c0 = (s0 < (int64_t)ctx) ? s0 : (int64_t)ctx;
                st.load = true; st.cri_load = cri(j0) - 1;
                st.cri_next = cri(j0); st.next0 = true;
                st.from = s0 - c0; st.to = s0; st.warm_from = st.from;
                return st;
            }
            const int64_t j = j0 + (t - 1);
            const int64_t s0 = bpos(j), e = bpos(j + 1);
            int64_t cn = (j + 1 < j_end) ? ((e < (int64_t)ctx) ? e : (int64_t)ctx) : 0;     // the next block's context, inside this one
            if (cn > e - s0) cn = e - s0;
            st.out = j >= b_first;
            st.enter = true; st.cri_enter = cri(j);
            st.load = true; st.cri_load = cri(j);
            st.cri_next = cri(j + 1);
            st.from = s0; st.to = e; st.warm_from = e - cn;
            return st;
        }
        // Blocks shorter than the context: every block behind its own virtual block [vp, p), vp = max(p - ctx, 0), whose
        // controls are row b of the per-block arrays; the block's own are row K + b.  What feeds the LAST filter over the
        // block's own rows is the oldest cached block of that input containing them: the input's reply to the `after`
        // request made m = min((ctx - N) / N, blocks rendered before) blocks earlier, at q = p - m N (chain/__init__.py:
        // 435-442: the first containing block in insertion order) -- a filter in it was cold-started min(ctx, q) rows before q,
        // and everything in it read its controls at q (the caller evaluates those rows of the K + b group there).
        const int64_t p = a.position + b_first * (int64_t)a.N;
        const int64_t vp = (p > (int64_t)ctx) ? p - ctx : 0;
        if (D < 2) {
            if (t == 0) {                                                      // the virtual block: the whole of it is the block's context
                st.load = true; st.cri_load = b_first;
                st.cri_next = a.K + b_first; st.next0 = true;
                st.from = vp; st.to = p; st.warm_from = vp;
            } else {
                st.enter = true; st.cri_enter = a.K + b_first;
                st.load = true; st.cri_load = a.K + b_first;
                st.from = p; st.to = p + a.N; st.warm_from = st.to; st.out = true;
            }
            return st;
        }
        const int64_t bg = (int64_t)a.blocks_before + b_first;                // blocks of this size rendered since the graph was fresh
        int64_t m = (a.N > 0) ? (ctx - a.N) / a.N : 0;
        if (m > bg - 1) m = bg - 1;
        if (m < 0) m = 0;
        const int64_t q = p - m * a.N;
        const int64_t qc = q - ((q < (int64_t)ctx) ? q : (int64_t)ctx);        // where the inner filter of the block's own rows cold-starts
        const int64_t c0 = (vp < (int64_t)ctx) ? vp : (int64_t)ctx;
        const int64_t r0 = vp - c0;                                            // ... and where the virtual block's inner filter does
        const int64_t r1 = (qc < r0) ? r0 : ((qc > vp) ? vp : qc);
        if (t == 0) {                                                          // the virtual block's inner filter alone (current chains from zero)
            st.enter = true; st.cri_enter = b_first;
            st.load = true; st.cri_load = b_first;
            st.from = r0; st.to = r1; st.warm_from = r1;
        } else if (t == 1) {                                                   // + the inner filter of the block's own rows warms up
            st.next_level = 1; st.cri_next = a.K + b_first; st.next0 = true;
            st.from = r1; st.to = vp; st.warm_from = r1;
        } else if (t == 2) {                                                   // the virtual block: the outer filter cold-starts on it
            st.next_level = 2; st.cri_next = a.K + b_first; st.next0 = true;
            st.from = vp; st.to = p; st.warm_from = vp;
        } else {
            st.enter = true; st.cri_enter = a.K + b_first;
            st.load = true; st.cri_load = a.K + b_first;
            st.from = p; st.to = p + a.N; st.warm_from = st.to; st.out = true;
        }
        return st;
    };

    // rows [from, to): every instruction of the program for a group of R consecutive rows at a time (R = RG, then single rows
    // for what is left of the segment); WARM rows also advance the next block's chains
    double q_lane = 0.0;
    int64_t qbase = 0;
    bool q_valid = false;
    sig_bus::FoldedGroup<CC> folded(tile, lane, dstp);                         // whole groups of the bus sink: sums folded across lanes in registers
    auto group = [&](int64_t n, int warm_r, bool out, auto rows_tag) {        // warm_r: the first row of the group that also advances the next chains (R: none)
        constexpr int R = decltype(rows_tag)::value;
        if (!q_valid || n < qbase || n + R > qbase + SIG_WAVE) {                // n / rate (IEEE divide) for 64 rows at a time, one per lane
            qbase = n;
            q_lane = (double)(qbase + lane) / a.rate;
            q_valid = true;
        }
        double q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = sig_readlane_f64(q_lane, (int)(n - qbase) + r);     // osc.py:32
#ifdef SIG_VP_STATIC_CODE
        // a specialised build: the program is a compile-time constant, the loop is unrolled and every switch below folds away
        constexpr uint32_t kStatic[] = SIG_VP_STATIC_CODE;
        constexpr bool kStaticExt = [] {
            constexpr uint32_t c[] = SIG_VP_STATIC_CODE;
            bool ext = false;
            for (unsigned k = 0; k < sizeof(c) / sizeof(c[0]); ++k) ext |= (c[k] & 31u) >= (uint32_t)SIG_VP_AMP;
            return ext;
        }();
        static_assert(EXT || !kStaticExt, "the program uses Amp / ADSR / White: build with -DSIG_VP_S_EXT=1");
#pragma unroll
        for (int pc = 0; pc < (int)(sizeof(kStatic) / sizeof(kStatic[0])); ++pc) {
            const uint32_t w = kStatic[pc];
#else
        for (int pc = 0; pc < a.n_ins; ++pc) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)codev, pc);
#endif
            const int op = (int)(w & 31u), kind = (int)((w >> 5) & 7u), ia = (int)((w >> 8) & 15u), ib = (int)((w >> 12) & 15u),
                      ic = (int)((w >> 16) & 15u);
            switch (op) {
                case SIG_VP_OSC:
                    with_index<NO>(ia, [&](auto I) {
                        constexpr int S = decltype(I)::value;
                        double t[R][VPT];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const double qr = vp_pin(q[r]);
#pragma unroll
                            for (int i = 0; i < VPT; ++i) t[r][i] = qr * ohz[S][i] + oph[S][i];
                        }
                        switch (kind) {
                            case SIG_OSC_SINE:
#pragma unroll
                                for (int r = 0; r < R; ++r)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) acc[r][i] = (double)sig_osc::osc_sine_f32(t[r][i]);
                                break;
                            case SIG_OSC_SAWTOOTH:
#pragma unroll
                                for (int r = 0; r < R; ++r)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) acc[r][i] = sig_osc::osc_sawtooth_fract(t[r][i]);
                                break;
                            case SIG_OSC_SQUARE:
#pragma unroll
                                for (int r = 0; r < R; ++r)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) acc[r][i] = sig_osc::osc_square_fract(t[r][i]);
                                break;
                            default:
#pragma unroll
                                for (int r = 0; r < R; ++r)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) acc[r][i] = sig_osc::osc_triangle_fract(t[r][i]);
                                break;
                        }
                    });
                    break;
                case SIG_VP_FILTER:
                    with_index<NF>(ia, [&](auto I) {
                        constexpr int F = decltype(I)::value;
                        if (warm_r < R) {                                      // wave-uniform: the next block's chain, on the same input rows
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                if (r >= warm_r) {                             // (wave-uniform too: a group may straddle the first warm row)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) {
                                        const double x = acc[r][i];
                                        const double yw = x + w0[F][i];
                                        w0[F][i] = fma(xa1[F][i], yw, fma(s2[F], x, w1[F][i]));
                                        w1[F][i] = fma(xa2[F][i], yw, x);
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < VPT; ++i) {
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const double x = acc[r][i];
                                const double y = x + z0[F][i];                 // DF2T of [1, s2, 1] / [1, a1, a2]
                                z0[F][i] = fma(na1[F][i], y, fma(s2[F], x, z1[F][i]));
                                z1[F][i] = fma(na2[F][i], y, x);
                                acc[r][i] = fb0[F][i] * y;
                            }
                        }
                    });
                    break;
                case SIG_VP_GAIN:
                    with_index<NP>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) acc[r][i] = acc[r][i] * pr[decltype(I)::value][i];      // fx.py:52
                    });
                    break;
                case SIG_VP_MUL:
                    with_index<NT>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) acc[r][i] = T[decltype(I)::value][r][i] * acc[r][i];     // fx.py:46
                    });
                    break;
                case SIG_VP_SAVE:
                    with_index<NT>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) T[decltype(I)::value][r][i] = acc[r][i];
                    });
                    break;
                case SIG_VP_LOAD:
                    with_index<NT>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) acc[r][i] = T[decltype(I)::value][r][i];
                    });
                    break;
                case SIG_VP_MIX: {                                             // m * L + (1 - m) * R (fx.py:40); ic: the accumulator is L
                    double m[VPT];
                    with_index<NP>(ib, [&](auto I) {
#pragma unroll
                        for (int i = 0; i < VPT; ++i) m[i] = pr[decltype(I)::value][i];
                    });
                    with_index<NT>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) {
                                const double l = ic ? acc[r][i] : T[decltype(I)::value][r][i], rr = ic ? T[decltype(I)::value][r][i] : acc[r][i];
                                acc[r][i] = m[i] * l + (1.0 - m[i]) * rr;
                            }
                    });
                    break;
                }
                case SIG_VP_CONST:
                    with_index<NP>(ia, [&](auto I) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int i = 0; i < VPT; ++i) acc[r][i] = pr[decltype(I)::value][i];
                    });
                    break;
                default:
                    if constexpr (EXT) {
                        if (op == SIG_VP_AMP) {
                            with_index<NP>(ia, [&](auto I) {
#pragma unroll
                                for (int r = 0; r < R; ++r)
#pragma unroll
                                    for (int i = 0; i < VPT; ++i) acc[r][i] = vp_amp(acc[r][i], pr[decltype(I)::value][i]);
                            });
                        } else if (op == SIG_VP_ADSR) {
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const double qr = vp_pin(q[r]);
                                bool stale = false;
#pragma unroll
                                for (int i = 0; i < VPT; ++i) stale |= !(qr < seg[i].end);
                                if (__any(stale)) seed_envelope(qr);           // a stage ended: at most five times per voice and stream
#pragma unroll
                                for (int i = 0; i < VPT; ++i) acc[r][i] = fma(seg[i].slope, qr - seg[i].t0, seg[i].l0);
                            }
                        } else if (op == SIG_VP_NOISE) {
#pragma unroll
                            for (int r = 0; r < R; ++r)
#pragma unroll
                                for (int i = 0; i < VPT; ++i) {
                                    const int ch = v0 + i;
                                    const uint64_t h = vp_mix64(a.seeds[ia & 1] + (uint64_t)(n + r) * 0x9E3779B97F4A7C15ULL +
                                                                (uint64_t)(ch >> 1) * 0xD1B54A32D192ED03ULL);
                                    const uint32_t k = ((ch & 1) ? (uint32_t)(h >> 32) : (uint32_t)h) >> 8;
                                    acc[r][i] = (double)((float)k * 5.9604644775390625e-8f);
                                }
                        }
                    }
                    break;
            }
        }
        if (out) {
            if constexpr (BUS && (R * CC == 16 || R * CC == 8)) {
                // a whole group of the bus: its R x C sums stay in registers, are folded across lanes four at a time (two
                // register-to-register halving steps, sig_bus::FoldedGroup) and only a quarter goes through the LDS
                if (stage.staged) stage.now();                                 // single rows staged before: out first, in order
                double sums[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) sums[k] = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ch = 0; ch < CC; ++ch) {
                        double sum = 0.0;
#pragma unroll
                        for (int i = 0; i < VPT; ++i) sum = fma(wt[ch][i], acc[r][i], sum);
                        sums[r * CC + ch] = sum;
                    }
#pragma unroll
                for (int g4 = 0; g4 < (R * CC) / 4; ++g4) folded.fold4(g4, sums[4 * g4], sums[4 * g4 + 1], sums[4 * g4 + 2], sums[4 * g4 + 3]);
                double pend[4];
                folded.issue(pend);
                folded.finish(pend, stage.first, R);
                stage.first += R;
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if constexpr (BUS) {
#pragma unroll
                        for (int ch = 0; ch < CC; ++ch) {
                            double sum = 0.0;
#pragma unroll
                            for (int i = 0; i < VPT; ++i) sum = fma(wt[ch][i], acc[r][i], sum);
                            stage.slot[ch * kTileStride] = sum;
                        }
                        stage.advance();
                    } else {
                        float y32[VPT];
#pragma unroll
                        for (int i = 0; i < VPT; ++i) y32[i] = (float)acc[r][i];
                        if (live0) {
                            Vec o; vp_put(o, y32);
                            *reinterpret_cast<Vec*>(dst + (n + r - a.position) * a.out_ld) = o;
                        }
                    }
                }
            }
        }
    };
    for (int t = 0; t < n_steps; ++t) {
        const Step st = make_step(t);
        if (st.load) load_params(st.cri_load);
        if (st.enter) enter_block(st.cri_enter);
        bool started = false;                                                  // the next chains of this step
        if (st.next0) { start_next(st.cri_next, st.next_level); started = true; }
        if (st.from >= st.to) continue;
        seed_envelope((double)st.from / a.rate);
        int64_t n = st.from;
        auto warm_rows = [&](int64_t at, int rows) {                           // first warm row of the group at `at` (rows: none), chains started on arrival
            if (st.warm_from >= at + rows) return rows;
            if (!started) { start_next(st.cri_next, st.next_level); started = true; }
            return (st.warm_from <= at) ? 0 : (int)(st.warm_from - at);
        };
        if constexpr (RG > 1) {
            for (; n + RG <= st.to; n += RG) group(n, warm_rows(n, RG), st.out, std::integral_constant<int, RG>{});
        }
        for (; n < st.to; ++n) group(n, warm_rows(n, 1), st.out, std::integral_constant<int, 1>{});
    }
    if (BUS && stage.staged) stage.now();
}

#ifndef SIG_VP_WAVES
#define SIG_VP_WAVES 2
#endif
#ifndef SIG_VP_WAVES1
#define SIG_VP_WAVES1 3
#endif
template <int VPT, bool SMALL, int C>
__device__ __forceinline__ void vp_kernel_body(const VpArgs& a)
{
    constexpr bool BUS = C > 0;
    __shared__ double lds[BUS ? 4 : 1][BUS ? sig_bus::kPairs * kTileStride : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    vp_wave<VPT, SMALL, C>(a, lds[BUS ? wave : 0], lane, wave);
    if constexpr (BUS) {
        if (a.bus_out) sig_bus::sum_tiles_in_workgroup<C>(a.partials, a.voice_tiles, a.rows, a.span, a.K, a.N, a.bus_out, a.bus_out_ld, lane, wave);
    }
}

#ifdef SIG_VP_STATIC_CODE
// A SPECIALISED build of this file (signals_amd/specialise.py: hipcc --genco with the program, the exact register file, the
// voices per lane and the sink as macros): one kernel, the same source as the interpreter with the program a compile-time
// constant -- the dispatch loop unrolls and every switch folds.  Attached to the library with sig_voice_program_attach.
}  // namespace
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SIG_VP_STATIC_WAVES, 8)))
void sig_vp_specialised(VpArgs a) { vp_kernel_body<SIG_VP_STATIC_VPT, true, SIG_VP_STATIC_C>(a); }
// what the attaching library checks before it trusts the image: the argument block's size and the program it was built for
extern "C" __global__ void sig_vp_specialised_info(uint32_t* out)
{
    constexpr uint32_t code[] = SIG_VP_STATIC_CODE;
    out[0] = (uint32_t)sizeof(VpArgs); out[1] = SIG_VP_STATIC_VPT; out[2] = SIG_VP_STATIC_C; out[3] = (uint32_t)(sizeof(code) / sizeof(code[0]));
    for (unsigned k = 0; k < sizeof(code) / sizeof(code[0]); ++k) out[4 + k] = code[k];
}
#else
template <int VPT, bool SMALL, int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMALL ? (VPT == 1 ? SIG_VP_WAVES1 : SIG_VP_WAVES) : 1, 8))) void voice_program_kernel(VpArgs a)
{
    vp_kernel_body<VPT, SMALL, C>(a);
}

struct VpTuning { int vpt = 0, span = 0, attached = 1; };
VpTuning& vp_tuning() { static VpTuning t; return t; }

// Specialised kernels attached at run time (sig_voice_program_attach): this file built once more for ONE program, found again
// by the program's words, the slot counts, the voices per lane and the sink.  Entries are never removed while launches may be
// in flight (detach is for tests and process exit); the list is short, a linear search costs less than the launch.
struct VpSpecial { uint32_t code[kMaxIns]; int n_ins, n_oscs, n_params, n_filters, n_temps, vpt, C; hipModule_t mod; hipFunction_t fn; };
std::vector<VpSpecial>& vp_specials() { static std::vector<VpSpecial> v; return v; }
std::mutex& vp_specials_lock() { static std::mutex m; return m; }

bool vp_encode(const sig_voice_program_t& P, uint32_t* code) {
    for (int k = 0; k < P.n_ins; ++k) {
        const sig_vp_ins& x = P.ins[k];
        if (!(x.op >= SIG_VP_OSC && x.op <= SIG_VP_NOISE && x.kind >= 0 && x.kind <= 7 && x.a >= 0 && x.a <= 15 && x.b >= 0 && x.b <= 15 && x.c >= 0 && x.c <= 15))
            return false;
        code[k] = (uint32_t)x.op | ((uint32_t)x.kind << 5) | ((uint32_t)x.a << 8) | ((uint32_t)x.b << 12) | ((uint32_t)x.c << 16);
    }
    return true;
}

hipFunction_t vp_find_special(const VpArgs& a, const sig_voice_program_t& P, int vpt, int C) {
    if (!vp_tuning().attached) return nullptr;
    std::lock_guard<std::mutex> g(vp_specials_lock());
    for (const VpSpecial& e : vp_specials())
        if (e.n_ins == a.n_ins && e.vpt == vpt && e.C == C && e.n_oscs == P.n_oscs && e.n_params == P.n_params &&
            e.n_filters == P.n_filters && e.n_temps == P.n_temps && memcmp(e.code, a.code, sizeof(uint32_t) * a.n_ins) == 0)
            return e.fn;
    return nullptr;
}

struct VpNeeds { int oscs, params, temps, filters; bool ext; };

bool fits_small(const VpNeeds& n) {
    using L = VpLimits<true>;
    return n.oscs <= L::NO && n.params <= L::NP && n.temps <= L::NT && n.filters <= L::NF && !n.ext;
}

// voices per lane and blocks per lane.  4 voices per lane amortise the interpreter's scalar work best; a lane's span
// re-walks (depth - 1) blocks + the context once, so longer spans waste less -- while the launch still has a wave or two
// for every SIMD
// `four`: a kernel SPECIALISED for four voices per lane is at hand (the interpreter exists for 1 and 2): straight-line code
// keeps a lone wave per SIMD busy, and the bus flush and the row's n / rate are paid per lane -- taken when the store is
// 16-byte aligned (or there is a bus) and the launch still has a wave for every SIMD
void vp_geometry(const VpArgs& a, int store_aligned, bool four, int& vpt, int& span) {
    auto waves = [&](int v, int s) { return (int64_t)((a.voices + SIG_WAVE * v - 1) / (SIG_WAVE * v)) * ((a.K + s - 1) / s); };
    const bool bus = a.partials != nullptr;
    vpt = ((bus || store_aligned >= 2) && waves(2, 1) >= 1024) ? 2 : 1;
    const bool can4 = four && (bus || store_aligned >= 4);
    if (can4 && waves(4, 1) >= 1024) vpt = 4;
    const int t = vp_tuning().vpt;
    if (t == 1 || (t == 2 && (bus || store_aligned >= 2)) || (t == 4 && can4)) vpt = t;
    span = 1;
    if (!a.small) {
        span = 16;
        while (span > 1 && waves(vpt, span) < (vpt == 4 ? 1024 : 2048)) span >>= 1;
        if (vp_tuning().span >= 1) span = vp_tuning().span;
    }
}

template <int VPT, bool SMALL>
int vp_launch_sink(const VpArgs& a, int C, unsigned nwg, hipStream_t s) {
    switch (C) {
        case 0: voice_program_kernel<VPT, SMALL, 0><<<nwg, 256, 0, s>>>(a); break;
        case 1: voice_program_kernel<VPT, SMALL, 1><<<nwg, 256, 0, s>>>(a); break;
        case 2: voice_program_kernel<VPT, SMALL, 2><<<nwg, 256, 0, s>>>(a); break;
        default: return (int)hipErrorInvalidValue;
    }
    return sig_launch_status();
}

}  // namespace

extern "C" int sig_voice_program_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane)
{
    SIG_CHECK_ARG(voices_per_lane >= 0 && blocks_per_lane >= 0);
    vp_tuning().vpt = voices_per_lane;
    vp_tuning().span = blocks_per_lane;
    return 0;
}

extern "C" int sig_voice_program_geometry(int32_t voices, int32_t block_frames, int32_t nblocks, int32_t context, int32_t depth,
                                          int32_t bus_channels, int32_t store_aligned, int32_t specialised,
                                          int32_t* voices_per_lane, int32_t* blocks_per_lane)
{
    SIG_CHECK_ARG(voices >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && depth >= 0 && voices_per_lane && blocks_per_lane);
    VpArgs a{};
    a.voices = voices; a.N = block_frames; a.K = nblocks; a.ctx = context;
    a.small = (depth > 0 && block_frames < context) ? 1 : 0;
    double dummy = 0.0;
    a.partials = bus_channels > 0 ? &dummy : nullptr;                          // (only asked whether there is a bus)
    int vpt = 1, span = 1;
    vp_geometry(a, store_aligned, specialised != 0, vpt, span);
    *voices_per_lane = vpt; *blocks_per_lane = span;
    return 0;
}

extern "C" int64_t sig_voice_program_args_size(void) { return (int64_t)sizeof(VpArgs); }

extern "C" int sig_voice_program_use_attached(int32_t on) { vp_tuning().attached = on ? 1 : 0; return 0; }

extern "C" int sig_voice_program_attach(const sig_voice_program_t* program, int32_t voices_per_lane, int32_t bus_channels,
                                        const void* image)
{
    SIG_CHECK_ARG(program && image && (voices_per_lane == 1 || voices_per_lane == 2 || voices_per_lane == 4) && bus_channels >= 0 && bus_channels <= 2);
    const sig_voice_program_t& P = *program;
    SIG_CHECK_ARG(P.n_ins >= 1 && P.n_ins <= SIG_VP_MAX_INS);
    VpSpecial e{};
    SIG_CHECK_ARG(vp_encode(P, e.code));
    e.n_ins = P.n_ins; e.n_oscs = P.n_oscs; e.n_params = P.n_params; e.n_filters = P.n_filters; e.n_temps = P.n_temps;
    e.vpt = voices_per_lane; e.C = bus_channels;
    hipError_t err = hipModuleLoadData(&e.mod, image);
    if (err != hipSuccess) { (void)hipGetLastError(); return (int)err; }       // (not sticky: the caller's next HIP call must not inherit it)
    hipFunction_t info = nullptr;
    err = hipModuleGetFunction(&e.fn, e.mod, "sig_vp_specialised");
    if (err == hipSuccess) err = hipModuleGetFunction(&info, e.mod, "sig_vp_specialised_info");
    // the image says what it was built for: the argument block's size, voices per lane, sink and program must be THIS library's
    uint32_t* dev = nullptr;
    uint32_t got[4 + kMaxIns] = {0};
    if (err == hipSuccess) err = hipMalloc(&dev, sizeof(got));
    if (err == hipSuccess) {
        void* params[] = {&dev};
        err = hipModuleLaunchKernel(info, 1, 1, 1, 1, 1, 1, 0, nullptr, params, nullptr);
        if (err == hipSuccess) err = hipMemcpy(got, dev, sizeof(got), hipMemcpyDeviceToHost);
        (void)hipFree(dev);
    }
    if (err == hipSuccess &&
        (got[0] != (uint32_t)sizeof(VpArgs) || got[1] != (uint32_t)e.vpt || got[2] != (uint32_t)e.C || got[3] != (uint32_t)e.n_ins ||
         memcmp(got + 4, e.code, sizeof(uint32_t) * e.n_ins) != 0))
        err = hipErrorInvalidImage;
    if (err != hipSuccess) { (void)hipModuleUnload(e.mod); (void)hipGetLastError(); return (int)err; }
    std::lock_guard<std::mutex> g(vp_specials_lock());
    vp_specials().push_back(e);
    return 0;
}

extern "C" int sig_voice_program_detach_all(void)
{
    std::lock_guard<std::mutex> g(vp_specials_lock());
    for (VpSpecial& e : vp_specials()) (void)hipModuleUnload(e.mod);
    vp_specials().clear();
    return 0;
}

extern "C" int sig_voice_program(const sig_voice_program_t* program, int32_t rate, int64_t position, int32_t block_frames,
                                 int32_t nblocks, int32_t context, int32_t voices, int32_t control_rows,
                                 int32_t hist_blocks, const int64_t* hist_positions, int32_t blocks_before,
                                 const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                 double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream)
{
    SIG_CHECK_ARG(program && rate > 0 && position >= 0 && block_frames >= 0 && nblocks >= 0 && context >= 0 && voices >= 0);
    SIG_CHECK_ARG(out && (bus_channels == 0 || bus_channels == 1 || bus_channels == 2));
    SIG_CHECK_ARG(bus_channels == 0 ? out_ld >= voices : (out_ld >= bus_channels && workspace != nullptr));
    SIG_CHECK_ARG(bus_gains ? (bus_channels > 0 && bus_gains_ld >= voices) : bus_channels <= 1);
    const sig_voice_program_t& P = *program;
    SIG_CHECK_ARG(P.n_ins >= 1 && P.n_ins <= SIG_VP_MAX_INS && P.n_oscs >= 0 && P.n_oscs <= SIG_VP_MAX_OSCS);
    SIG_CHECK_ARG(P.n_params >= 0 && P.n_params <= SIG_VP_MAX_PARAMS && P.n_filters >= 0 && P.n_filters <= SIG_VP_MAX_FILTERS);
    SIG_CHECK_ARG(P.n_temps >= 0 && P.n_temps <= SIG_VP_MAX_TEMPS && P.depth >= 0 && P.depth <= P.n_filters);
    SIG_CHECK_ARG(hist_blocks >= 0 && hist_blocks <= SIG_VP_MAX_HIST && (hist_blocks == 0 || hist_positions) && blocks_before >= 0);
    const bool small = P.depth > 0 && block_frames < context;
    SIG_CHECK_ARG(!small || (P.depth <= 2 && block_frames >= 16));            // (deeper cascades, or blocks so short that the reference's 16-entry block cache evicts what they read: the eager path)
    SIG_CHECK_ARG(small || position == 0 || hist_blocks >= (P.depth > 1 ? 1 : 0));
    SIG_CHECK_ARG(control_rows == (small ? 2 * nblocks : hist_blocks + 1 + nblocks));
    if (block_frames == 0 || nblocks == 0 || voices == 0) return 0;
    VpArgs a{};
    a.rate = (double)rate; a.position = position; a.N = block_frames; a.K = nblocks; a.ctx = context; a.voices = voices;
    auto rows_ok = [&](const sig_vp_rows& r, bool optional) {
        if (!r.ptr) return optional;
        return (r.col_stride | 1) == 1 && (r.rows == 1 || r.rows == control_rows);
    };
    VpNeeds need{P.n_oscs, P.n_params, P.n_temps, P.n_filters, false};
    bool has_adsr = false;
    a.n_ins = P.n_ins;
    for (int k = 0; k < P.n_ins; ++k) {
        const sig_vp_ins& x = P.ins[k];
        SIG_CHECK_ARG(x.op >= SIG_VP_OSC && x.op <= SIG_VP_NOISE && x.kind >= 0 && x.kind <= 7 && x.a >= 0 && x.a <= 15 && x.b >= 0 && x.b <= 15 && x.c >= 0 && x.c <= 15);
        switch (x.op) {
            case SIG_VP_OSC: SIG_CHECK_ARG(x.a < P.n_oscs && x.kind <= SIG_OSC_TRIANGLE); break;
            case SIG_VP_FILTER: SIG_CHECK_ARG(x.a < P.n_filters); break;
            case SIG_VP_GAIN: case SIG_VP_CONST: case SIG_VP_AMP: SIG_CHECK_ARG(x.a < P.n_params); break;
            case SIG_VP_MUL: case SIG_VP_SAVE: case SIG_VP_LOAD: SIG_CHECK_ARG(x.a < P.n_temps); break;
            case SIG_VP_MIX: SIG_CHECK_ARG(x.a < P.n_temps && x.b < P.n_params); break;
            case SIG_VP_NOISE: SIG_CHECK_ARG(x.a < 2); break;
            default: break;
        }
        if (x.op == SIG_VP_AMP || x.op == SIG_VP_ADSR || x.op == SIG_VP_NOISE) need.ext = true;
        if (x.op == SIG_VP_ADSR) has_adsr = true;
    }
    SIG_CHECK_ARG(vp_encode(P, a.code));
    a.n_oscs = P.n_oscs; a.n_params = P.n_params; a.n_filters = P.n_filters;
    for (int k = 0; k < P.n_oscs; ++k) {
        SIG_CHECK_ARG(rows_ok(P.hertz[k], false) && rows_ok(P.phase[k], true));
        a.hertz[k] = Rows{P.hertz[k].ptr, P.hertz[k].col_stride, P.hertz[k].rows};
        a.phase[k] = Rows{P.phase[k].ptr, P.phase[k].col_stride, P.phase[k].ptr ? P.phase[k].rows : 1};
    }
    for (int k = 0; k < P.n_params; ++k) {
        SIG_CHECK_ARG(rows_ok(P.params[k], false));
        a.params[k] = Rows{P.params[k].ptr, P.params[k].col_stride, P.params[k].rows};
    }
    for (int k = 0; k < P.n_filters; ++k) {
        SIG_CHECK_ARG(rows_ok(P.cutoff[k], false) && (P.filter_type[k] == SIG_FILT_LOWPASS || P.filter_type[k] == SIG_FILT_HIGHPASS));
        a.cutoff[k] = Rows{P.cutoff[k].ptr, P.cutoff[k].col_stride, P.cutoff[k].rows};
        a.ftype[k] = P.filter_type[k];
        SIG_CHECK_ARG(P.filter_level[k] >= 1 && P.filter_level[k] <= P.depth);
        a.flevel[k] = P.filter_level[k];
    }
    if (has_adsr) {
        SIG_CHECK_ARG(sig_env::load_rows(P.adsr, P.adsr_stride, a.adsr));
        a.has_adsr = 1;
    }
    a.seeds[0] = P.noise_seed[0]; a.seeds[1] = P.noise_seed[1];
    a.depth = P.depth; a.small = small ? 1 : 0; a.blocks_before = blocks_before;
    a.hist = small ? 0 : hist_blocks;
    for (int k = 0; k < a.hist; ++k) {
        SIG_CHECK_ARG(hist_positions[k] >= 0 && hist_positions[k] < (k + 1 < a.hist ? hist_positions[k + 1] : position));
        a.hist_pos[k] = hist_positions[k];
    }
    a.status = status;
    const int64_t rows = (int64_t)block_frames * nblocks;
    if (bus_channels > 0) {
        a.pan = bus_gains; a.pan_ld = bus_gains_ld; a.partials = workspace; a.rows = rows;
    } else {
        a.out = out; a.out_ld = out_ld;
    }
    const int aligned = (voices % 4 == 0 && out_ld % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 4
                      : (voices % 2 == 0 && out_ld % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0) ? 2 : 1;
    int vpt = 1;
    vp_geometry(a, aligned, vp_find_special(a, P, 4, bus_channels) != nullptr, vpt, a.span);
    const bool small_file = fits_small(need);
    a.voice_tiles = (voices + SIG_WAVE * vpt - 1) / (SIG_WAVE * vpt);
    const int64_t nwg = ((int64_t)a.voice_tiles * ((a.K + a.span - 1) / a.span) + 3) / 4;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (bus_channels > 0 && sig_bus::tiles_sum_in_workgroup(a.voice_tiles)) { a.bus_out = out; a.bus_out_ld = out_ld; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int err;
    if (hipFunction_t fn = vp_find_special(a, P, vpt, bus_channels)) {         // this very program, built as straight-line code
        void* params[] = {&a};
        err = (int)hipModuleLaunchKernel(fn, (unsigned)nwg, 1, 1, 256, 1, 1, 0, s, params, nullptr);
    } else if (vpt == 4) {
        return (int)hipErrorInvalidValue;                                      // (forced by the tuning hook after the image was switched off)
    } else if (vpt == 2) err = small_file ? vp_launch_sink<2, true>(a, bus_channels, (unsigned)nwg, s)
                                   : vp_launch_sink<2, false>(a, bus_channels, (unsigned)nwg, s);
    else err = small_file ? vp_launch_sink<1, true>(a, bus_channels, (unsigned)nwg, s)
                          : vp_launch_sink<1, false>(a, bus_channels, (unsigned)nwg, s);
    if (err || bus_channels == 0 || a.bus_out) return err;
    switch (bus_channels) {
        case 1: return sig_bus::launch_partials<1>(a.partials, a.voice_tiles, rows, out, out_ld, s);
        default: return sig_bus::launch_partials<2>(a.partials, a.voice_tiles, rows, out, out_ld, s);
    }
}
#endif  // SIG_VP_STATIC_CODE
