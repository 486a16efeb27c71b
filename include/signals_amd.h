/* signals_amd.h -- C ABI of the MI355X (gfx950) block-render kernels.
 *
 * The reference (noah-aviel-dove/signals) is 100 % Python: its render path has no FFI.
 * The functions a native back end has to replace are the numpy/scipy bodies of the
 * nodes' `_eval` methods; each entry point below names the one it stands in for
 * (paths relative to /root/reference/src/signals/).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - audio buffers: row-major (frames, channels), channel (= voice) index contiguous,
 *    `ld` = elements between consecutive rows.  dtype f32 (SIG_F32) by default; every
 *    entry also accepts f64 (SIG_F64) buffers, which block-rate (frames == 1) control
 *    requests use.
 *  - control rows (hertz, phase, cutoff, gain ...): f64, `stride` 1 = one value per voice,
 *    0 = one scalar broadcast to every voice (a (1,1) reply, chain/__init__.py:59-63).
 *  - every pointer is a DEVICE pointer; the caller owns all memory; nothing is allocated,
 *    freed or synchronised inside (safe to capture into a hipGraph).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - return value: 0 on success, otherwise a hipError_t (1 = hipErrorInvalidValue for
 *    argument errors).  No exceptions cross the boundary.
 */
#ifndef SIGNALS_AMD_H
#define SIGNALS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIG_ABI_VERSION 7

enum { SIG_F32 = 0, SIG_F64 = 1 };

/* oscillator kinds: osc.py:40-62 */
enum { SIG_OSC_SINE = 0, SIG_OSC_SQUARE = 1, SIG_OSC_SAWTOOTH = 2, SIG_OSC_TRIANGLE = 3 };

/* critical-frequency filter types: fx.py:68-72 (only lp/hp are live in the reference, fx.py:142-151) */
enum { SIG_FILT_LOWPASS = 0, SIG_FILT_HIGHPASS = 1, SIG_FILT_BANDPASS = 2, SIG_FILT_BANDSTOP = 3 };

/* element-wise effects: fx.py:35-60 */
enum { SIG_EW_GAIN = 0, SIG_EW_MIX = 1, SIG_EW_RINGMOD = 2, SIG_EW_AMP = 3 };

/* bits OR-ed into the optional device status word by kernels (never cleared by them) */
enum { SIG_STATUS_BAD_CUTOFF = 1 };  /* Wn <= 0 or >= 1: scipy raises ValueError (fx.py:99-102) */

int sig_abi_version(void);

/* Replaces Osc._eval + Sine/Square/Sawtooth/Triangle._osc (osc.py:26-62):
 *   t[n,v] = (position + n) / rate * hertz[v] + phase[v]      (f64, that operator order)
 *   out[n,v] = wave_kind(t[n,v])
 * Position-pure: `rows` may span any number of consecutive blocks.
 * Square/Sawtooth/Triangle are bit-exact in f64 before the store.  Sine: f64 store within 1 ulp(f64)
 * (f64 polynomial); f32 store within 1.3e-7 (exact f64 phase reduction, then v_sin_f32). */
int sig_osc_bank(int kind, int64_t position, int32_t rate, int64_t rows, int32_t voices,
                 const double* hertz, int32_t hertz_stride,
                 const double* phase, int32_t phase_stride,   /* phase may be NULL = unplugged = 0 */
                 void* out, int32_t out_dtype, int64_t out_ld, void* stream);

/* sig_osc_bank for launches whose control inputs change per block (Osc._eval reads hertz and phase through
 * forward_at_block_rate, osc.py:28-30, so in a K-block batch they are K rows) and for block-RATE launches:
 *   output row r is frame position + r * position_step and uses parameter row r / rows_per_param
 *   (rows_per_param == 0: a single parameter row, the *_row_stride arguments are ignored).
 * audio rate, per-block parameters: position_step = 1, rows_per_param = block_frames;
 * block rate (what a control port sees for K blocks): position_step = block_frames, rows_per_param = 1. */
int sig_osc_bank_mod(int kind, int64_t position, int64_t position_step, int32_t rate, int64_t rows,
                     int32_t voices, int32_t rows_per_param,
                     const double* hertz, int32_t hertz_stride, int64_t hertz_row_stride,
                     const double* phase, int32_t phase_stride, int64_t phase_row_stride,
                     void* out, int32_t out_dtype, int64_t out_ld, void* stream);

/* Replaces CritFilter._filter + _get_sos (fx.py:85-121) for LowPass/HighPass (order 2 = one
 * biquad section), batched over `nblocks` consecutive blocks of `block_frames` frames.
 * For block b (p_b = position + b*block_frames, c_b = min(context, p_b)):
 *   design butter(2, clip(cutoff[b or 0, v] / (rate/2), 0, 1)) in closed form,
 *   run the DF2T recurrence in f64 from ZERO state over in rows [b*N - c_b, (b+1)*N),
 *   store rows [b*N, (b+1)*N).
 * `in` points at the row aligned with out row 0; `in_history` rows (>= min(context, position))
 * must precede it in the same allocation (same ld).  The reference's `after` window never
 * reaches the kept samples (sosfilt is causal) and is not read.
 * cutoff: (cutoff_blocks, voices) f64 with cutoff_blocks in {1, nblocks}; stride 0/1 as above.
 * status: optional device int32; SIG_STATUS_BAD_CUTOFF is OR-ed in and the voice's output is NaN. */
int sig_biquad_coldstart(int type, int32_t rate, int64_t position,
                         int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                         const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                         const void* in, int64_t in_ld, int64_t in_history,
                         void* out, int64_t out_ld, int32_t dtype,
                         int32_t* status, void* stream);

/* sig_biquad_coldstart whose stored rows are multiplied by a per-voice ADSR envelope evaluated at the row's
 * time (f32 buffers): RingMod(Filter(x), ADSR) -- fx.py:43-46 over fx.py:85-121 and the envelope of sig_adsr --
 * without the envelope store and the extra read/write pass.  adsr_params / adsr_strides as in sig_adsr. */
int sig_biquad_coldstart_env(int type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                             const double* const* adsr_params, const int32_t* adsr_strides,
                             const float* in, int64_t in_ld, int64_t in_history,
                             float* out, int64_t out_ld, int32_t* status, void* stream);

/* Filter [x envelope] summed straight into the bus: out[n,c] = sum_v bus_gains[c,v] * env[n,v] * Filter(in)[n,v]
 * -- SumBus(Filter(x)) or SumBus(RingMod(Filter(x), ADSR)) (fx.py:85-121, fx.py:43-46, the build-defined bus) in
 * one pass over `in`; nothing per-voice is written.  adsr_params == NULL: no envelope.  bus_gains == NULL:
 * bus_channels == 1, plain sum.  `workspace`: device, at least sig_fused_voice_bus_workspace(voices, rows,
 * bus_channels) bytes (per-voice-tile f64 partials, added in a fixed order by a second launch).  Buffers as in
 * sig_biquad_coldstart (float32 only). */
int sig_biquad_coldstart_bus(int type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* cutoff, int32_t cutoff_stride, int32_t cutoff_blocks,
                             const double* const* adsr_params, const int32_t* adsr_strides,
                             const float* in, int64_t in_ld, int64_t in_history,
                             const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                             double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* BandPass / BandStop done right (the reference's DoubleCritFilter raises TypeError at
 * fx.py:99, its intent being butter(N=2, Wn=[low, high], btype='bp'|'bs', output='sos') + sosfilt):
 * two biquad sections in series, designed in closed form per voice, same cold-start block semantics and
 * buffer conventions as sig_biquad_coldstart.  low/high: f64 rows (1,V)|(1,1).
 * Pinned against scipy.signal.butter + sosfilt (the reference itself has no working output to compare). */
int sig_band_coldstart(int type, int32_t rate, int64_t position,
                       int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                       const double* low, int32_t low_stride, const double* high, int32_t high_stride,
                       const void* in, int64_t in_ld, int64_t in_history,
                       void* out, int64_t out_ld, int32_t dtype,
                       int32_t* status, void* stream);

/* One operand of an element-wise effect.  row_stride / col_stride are in elements; 0 broadcasts
 * that axis (numpy broadcasting of (1,V), (N,1), (1,1) replies). */
typedef struct sig_operand {
    const void* ptr;
    int64_t row_stride;
    int32_t col_stride;
    int32_t dtype;      /* SIG_F32 or SIG_F64 */
    int32_t row_div;    /* > 1: output row r reads operand row r / row_div (a block-rate operand that changes per
                           block in a batched launch: row_div = block_frames); 0 or 1: operand row = r */
    int32_t reserved;
} sig_operand;

/* Replaces Gain/Mix/RingMod/Amp._eval (fx.py:35-60), arithmetic in f64:
 *   GAIN    out = a * b                       (b = right at block rate)
 *   MIX     out = c * a + (1 - c) * b         (c = mix at block rate)
 *   RINGMOD out = a * b
 *   AMP     out = copysign(a ** b, a)         (NaN for a < 0 with fractional b, like numpy) */
int sig_elementwise(int op, int64_t rows, int32_t cols,
                    const sig_operand* a, const sig_operand* b, const sig_operand* c,
                    void* out, int64_t out_ld, int32_t out_dtype, void* stream);

/* Build-defined sum bus (the reference's Flatten/FlattenUnit crash, shape.py:32-41):
 *   gains == NULL : out[n,0] = sum_v x[n,v]                          (bus_channels must be 1)
 *   gains (C,V)   : out[n,c] = sum_v gains[c*gains_ld + v] * x[n,v]
 * f64 accumulation, fixed summation order (lane-strided, then a butterfly), deterministic. */
int sig_sum_bus(int64_t rows, int32_t voices, const void* x, int64_t x_ld, int32_t x_dtype,
                const double* gains, int64_t gains_ld, int32_t bus_channels,
                void* out, int64_t out_ld, int32_t out_dtype, void* stream);

/* Replaces White._eval (noise.py:22-23: np.random.rand(frames, channels), uniform [0,1), global
 * unseeded RNG -- not reproducible, so parity is statistical only).  Here the value of sample
 * (position + n, channel) is a counter-based hash of (seed, frame, channel): position-pure and
 * reproducible; 24 random mantissa bits, u = k * 2^-24, k in [0, 2^24). */
int sig_white_noise(uint64_t seed, int64_t position, int64_t rows, int32_t channels,
                    void* out, int32_t out_dtype, int64_t out_ld, void* stream);

/* Build-defined ADSR envelope bank (the reference has only a dead sketch, sig.py:89-100):
 * position-pure piecewise-linear envelope per voice at frame rate; definition in
 * oracle/chain_ref.py:adsr.  params[6] = device pointers to the f64 rows
 * {attack, decay, sustain, release, gate_on, gate_off} (seconds; sustain is a level), strides[6]
 * their 0/1 strides.  `params` and `strides` themselves are HOST arrays. */
int sig_adsr(int64_t position, int32_t rate, int64_t rows, int32_t voices,
             const double* const* params, const int32_t* strides,
             void* out, int32_t out_dtype, int64_t out_ld, void* stream);

/* out = ADSR envelope * x in one pass: the RingMod(x, ADSR) pair (fx.py:43-46 over the envelope of sig_adsr) without
 * storing the envelope -- 8 B per voice-sample instead of 16.  f32 in / f32 out, product formed in f64. */
int sig_adsr_apply(int64_t position, int32_t rate, int64_t rows, int32_t voices,
                   const double* const* params, const int32_t* strides,
                   const float* x, int64_t x_ld, float* out, int64_t out_ld, void* stream);

/* Build-defined dense mix matrix (BASELINE config 5), f32 in/out, exact-f32 MFMA
 * (v_mfma_f32_32x32x2_f32):  out[n, 64g:64g+64] = x[n, 64g:64g+64] @ matrix,  matrix (64,64)
 * row-major contiguous, voices % 64 == 0, x 16-byte aligned with x_ld % 4 == 0. */
int sig_mix_matrix(int64_t rows, int32_t voices, const float* x, int64_t x_ld,
                   const float* matrix, float* out, int64_t out_ld, void* stream);

/* Fused voice chain, chosen by the batched engine when the intermediate node outputs have no other
 * consumer:  out = [gain *] Filter(Osc)  for `nblocks` cold-started blocks, i.e. sig_osc_bank ->
 * sig_biquad_coldstart -> sig_elementwise(GAIN) without the oscillator / filter stores
 * (osc.py:26-62 + fx.py:85-121 + fx.py:51-52).  Context rows are recomputed from the position-pure
 * oscillator.  f32 store; gain == NULL skips the gain stage; cutoff is one (1,V)|(1,1) row. */
int sig_fused_osc_biquad(int osc_kind, int filt_type, int32_t rate, int64_t position,
                         int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                         const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                         const double* cutoff, int32_t cutoff_stride,
                         const double* gain, int32_t gain_stride,
                         float* out, int64_t out_ld, int32_t* status, void* stream);

/* The fused chain / chain + bus with cutoff and gain read PER BLOCK: the reference reads a control port once per block,
 * at the block's position (forward_at_block_rate, chain/__init__.py:305-306), so an LFO on a cutoff or a tremolo gives
 * every block its own filter design and gain.  cutoff_rows / gain_rows: 1 (one row for the launch) or nblocks (row b
 * for block b, rows contiguous: row b of a per-voice parameter starts at + b * voices, of a broadcast one at + b).
 * The row-by-row span walker (the next block's warm-up chain runs with the next block's design) -- except a Sine voice
 * under a bus (sig_fused_voice_bus_rows), which keeps the closed form of sig_fused_voice_bus: the filter, its response at the
 * voice's frequency, T_c and the decay bound derived per (block, voice) inside the launch, the steady-state recurrence re-seeded
 * at every block's first row.  gain may be NULL; bus_channels 1 or 2.  hertz and phase stay one row here (sig_fused_*_fm below takes rows for them too). */
int sig_fused_osc_biquad_rows(int osc_kind, int filt_type, int32_t rate, int64_t position,
                              int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                              const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                              const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                              const double* gain, int32_t gain_stride, int32_t gain_rows,
                              float* out, int64_t out_ld, int32_t* status, void* stream);
int sig_fused_voice_bus_rows(int osc_kind, int filt_type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                             const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                             const double* gain, int32_t gain_stride, int32_t gain_rows,
                             const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                             double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* ... and with hertz and phase read per block as well: block-rate frequency / phase modulation (Osc._eval reads both
 * ports with forward_at_block_rate, osc.py:28-30).  hertz_rows / phase_rows: 1 or nblocks, rows laid out like cutoff's.
 * The reference's oscillators keep their previous block (BlockCachingEmitter, chain/__init__.py:424-442), so the context
 * rows a filter sees in front of block b are block b - 1's samples (block_frames >= context: with shorter blocks the
 * reference answers the context request as a block of its own -- not this entry point's semantics, callers keep such graphs
 * on the per-node path), made with parameter row b - 1: the walker's
 * warm-up chain already runs on the samples at hand.  For the launch's first block that row is *_hist (one (1,V)|(1,1)
 * row, same stride; non-NULL marks the parameter as modulated -- also in a one-block launch, whose single row still has a
 * different row in front; required when the matching *_rows > 1): the previous batch's last row on a continuing stream, the
 * controls evaluated at position - min(context, position) on a fresh graph.  Sine's incremental phase is re-seeded at every
 * block's first row with that block's hertz / phase. */
int sig_fused_osc_biquad_fm(int osc_kind, int filt_type, int32_t rate, int64_t position,
                            int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                            const double* hertz, int32_t hertz_stride, int32_t hertz_rows, const double* hertz_hist,
                            const double* phase, int32_t phase_stride, int32_t phase_rows, const double* phase_hist,
                            const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                            const double* gain, int32_t gain_stride, int32_t gain_rows,
                            float* out, int64_t out_ld, int32_t* status, void* stream);
int sig_fused_voice_bus_fm(int osc_kind, int filt_type, int32_t rate, int64_t position,
                           int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                           const double* hertz, int32_t hertz_stride, int32_t hertz_rows, const double* hertz_hist,
                           const double* phase, int32_t phase_stride, int32_t phase_rows, const double* phase_hist,
                           const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                           const double* gain, int32_t gain_stride, int32_t gain_rows,
                           const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                           double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* The same two entry points for a filter that reads TWO oscillators through an element-wise node: pair_op 1 =
 * Mix(A, B, mix): mix * A + (1 - mix) * B with mix one (1,V)|(1,1) row (Mix._eval, fx.py:35-40); pair_op 2 =
 * RingMod(A, B): A * B (fx.py:43-46; mix ignored, may be NULL).  A = (osc_kind, hertz, phase), B = (osc2_kind, hertz2,
 * phase2); both evaluated per row in the walker (B always with the exact per-row phase), nothing per-voice stored. */
int sig_fused_osc_pair_biquad(int osc_kind, int osc2_kind, int pair_op, int filt_type, int32_t rate, int64_t position,
                              int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                              const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                              const double* hertz2, int32_t hertz2_stride, const double* phase2, int32_t phase2_stride,
                              const double* mix, int32_t mix_stride,
                              const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                              const double* gain, int32_t gain_stride, int32_t gain_rows,
                              float* out, int64_t out_ld, int32_t* status, void* stream);
int sig_fused_voice_pair_bus(int osc_kind, int osc2_kind, int pair_op, int filt_type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                             const double* hertz2, int32_t hertz2_stride, const double* phase2, int32_t phase2_stride,
                             const double* mix, int32_t mix_stride,
                             const double* cutoff, int32_t cutoff_stride, int32_t cutoff_rows,
                             const double* gain, int32_t gain_stride, int32_t gain_rows,
                             const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                             double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* hipGraph support for the launch-bound latency loop (one block per pull): the same chain with the frame
 * position read from DEVICE memory, plus a one-thread kernel that advances it, so a captured graph
 * [chain(position_dev) -> bus -> position_dev += block_frames*nblocks] replays unchanged block after block. */
int sig_fused_osc_biquad_devpos(int osc_kind, int filt_type, int32_t rate, const int64_t* position_dev,
                                int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                const double* cutoff, int32_t cutoff_stride,
                                const double* gain, int32_t gain_stride,
                                float* out, int64_t out_ld, int32_t* status, void* stream);
int sig_advance_position(int64_t* position_dev, int64_t delta, void* stream);

/* A block-rate control subgraph as one launch.  The reference reads a control port once per block, at the block's position
 * (BoundPort.forward_at_block_rate, chain/__init__.py:305-306); an LFO sweep or a tremolo is an oscillator evaluated at one
 * frame per block (Osc._eval, osc.py:26-62) combined by element-wise nodes (fx.py:35-60).  `program` (device memory, n_ins <=
 * SIG_CTL_MAX_INS = SIG_CTL_MAX_REGS: one register per instruction, the register file is sized by n_ins; a longer program is refused with
 * hipErrorInvalidValue) is that subgraph in evaluation order over registers dst < n_ins; thread (b, v) runs it for
 * block b (frame position + b * step) at column v < cols and writes register outs[k].reg to outs[k].out[b * outs[k].cols + v]
 * for v < outs[k].cols.  `position` may lie before `min_position` (even be negative): blocks whose frame position + b * step is below
 * min_position are evaluated AT min_position (the virtual blocks in front of short blocks read their controls at max(p - context, 0);
 * what feeds the last filter of a short block read them at an earlier block's position, never before the stream's second block).  A register index of -1 reads 0.  Same expressions as sig_osc_bank (f64 store) and sig_elementwise,
 * so the same bits as the node-by-node evaluation. */
enum { SIG_CTL_ROW = 0, SIG_CTL_OSC = 1, SIG_CTL_GAIN = 2, SIG_CTL_MIX = 3, SIG_CTL_RINGMOD = 4, SIG_CTL_AMP = 5 };
enum { SIG_CTL_MAX_REGS = 48, SIG_CTL_MAX_INS = 48 };
typedef struct {
    int32_t op;              /* SIG_CTL_* */
    int32_t kind;            /* OSC: SIG_OSC_* */
    int32_t a, b, c;         /* operand registers.  OSC: a = hertz, b = phase; GAIN / RINGMOD / AMP: a, b; MIX: a, b, c = mix */
    int32_t dst;             /* result register */
    int32_t stride, rows;    /* ROW: a (rows, cols) float64 array, rows 1 | nblocks, rows contiguous, element v * stride (cols 1: stride 0) */
    int32_t cols;            /* EVERY instruction: columns of its result (1: evaluated once per block; the broadcast of its operands' widths otherwise) */
    int32_t reserved;
    const double* row;
} sig_ctl_ins;
typedef struct { int32_t reg; int32_t cols; double* out; double* front; } sig_ctl_out;   /* out: (nblocks, cols) float64, contiguous; front: (1, cols) or NULL */
/* front_position >= 0: the program is evaluated once more, at that frame position, into the outputs' `front` rows (the
 * controls of the block in front of the batch -- sig_fused_*_fm's *_hist -- without a launch of their own); -1: not. */
int sig_control_program(int32_t rate, int64_t position, int32_t step, int32_t nblocks, int32_t cols, int64_t front_position,
                        int64_t min_position,
                        const sig_ctl_ins* program, int32_t n_ins, const sig_ctl_out* outs, int32_t n_outs, void* stream);

/* SPECIALISED control programs: signals_amd/csrc/control_program.hip built as a gfx950 code object for one program STRUCTURE
 * (macros SIG_CTL_STATIC_INS = {{op, kind, a, b, c, dst, wide}, ...} in evaluation order and SIG_CTL_STATIC_OUTS = {{reg, wide},
 * ...}; `hipcc --genco`, signals_amd/specialise.py): the registers are VGPRs instead of an LDS file behind an interpretive loop.
 * `description` = [n_ins, n_outs, the seven words per instruction, the two per output]; the image must describe itself the same
 * way (its sig_ctl_specialised_info kernel) or hipErrorInvalidImage.  A set-up call (allocates, launches, synchronises); the
 * handle (>= 1) is valid for the life of the process.  sig_control_program_attached = sig_control_program through that kernel:
 * the CALLER vouches that `program` / `outs` have the structure the handle was built for (they are device memory, the library
 * cannot look); row pointers, strides and output pointers are read from them as usual.  The same values, bit for bit. */
int sig_control_program_attach(const int32_t* description, int32_t n_words, const void* image, int32_t* handle);
int sig_control_program_attached(int32_t handle, int32_t rate, int64_t position, int32_t step, int32_t nblocks, int32_t cols,
                                 int64_t front_position, int64_t min_position,
                                 const sig_ctl_ins* program, int32_t n_ins, const sig_ctl_out* outs, int32_t n_outs, void* stream);

/* Fused voice chain + dense mix matrix:  out[n, 64g : 64g+64] = ([gain *] Filter(Osc))[n, 64g : 64g+64] @ matrix
 * -- Osc._eval (chain/osc.py:26-62), CritFilter._filter (chain/fx.py:85-121), Gain._eval (chain/fx.py:49-52) and the
 * build-defined MixMatrix, i.e. the chain of sig_fused_osc_biquad feeding sig_mix_matrix (BASELINE config 5) without the per-voice rows going
 * through HBM: every 32 rows of a 64-voice group are staged as float32 in LDS and multiplied on the matrix cores.  The default
 * sink contracts each float32 operand as the exact sum of three bfloat16 (six v_mfma_f32_32x32x16_bf16 products per k-block,
 * float32 accumulators, the three cross terms below 2^-26 of a product dropped; sig_mix_tile.h): WITHIN A FEW float32 ULP of
 * sig_mix_matrix over sig_fused_osc_biquad's output (measured 4-7 ulp of the rows' scale between the two, each 2-5 ulp from
 * the f64 product), not bit for bit.  sig_fused_set_tuning(closed_form = 3) selects v_mfma_f32_32x32x2_f32 (the instruction
 * of sig_mix_matrix) instead.  Non-finite rows: the split forms x - bf16(x), so an Inf sample becomes NaN in the mixed row where
 * the float32 instruction keeps Inf (either way the row is unusable and a rejected design also sets the status bit); residual
 * words below bfloat16's range (|x| < ~1e-33 after two splits) flush to zero, an absolute error below 1e-35.
 * voices % 64 == 0; matrix (64, 64) float32 row-major on the device. */
int sig_fused_osc_biquad_mix(int osc_kind, int filt_type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                             const double* cutoff, int32_t cutoff_stride,
                             const double* gain, int32_t gain_stride,
                             const float* matrix, float* out, int64_t out_ld, int32_t* status, void* stream);

/* Fused voice chain + sum bus:  out[n,c] = sum_v bus_gains[c,v] * ([gain[v] *] Filter(Osc)[n,v])
 * (bus_gains == NULL: bus_channels == 1, plain sum).  Nothing per-voice touches HBM.  Launches inside: the chain
 * kernel writes per-voice-tile f64 partials into `workspace` (device, at least
 * sig_fused_voice_bus_workspace(voices, rows, bus_channels) bytes, rows = block_frames*nblocks), a second kernel adds
 * the tiles in a fixed order and rounds to f32; for a Sine oscillator a one-thread-per-voice kernel first derives the
 * constants of the closed form (steady-state sinusoid + homogeneous transient per cold-started block; voices it does
 * not cover are walked row by row in the same launch) into the tail of the workspace.  Deterministic; no atomics. */
int64_t sig_fused_voice_bus_workspace(int32_t voices, int64_t rows, int32_t bus_channels);
/* sig_fused_voice_bus restricted to the row-by-row span walker (never the Sine closed form).  The closed form, like
 * the walker's incremental Sine phase, holds while |t| = |frame / rate * hertz + phase| < 2^26 cycles (past that the
 * reference's own rounding of t, which neither reproduces, approaches the 1e-6 bar): a wave with a voice beyond it is
 * done inside the closed-form launch by a plain per-row fallback that is correct but ~40x slower, fine for a few
 * waves.  A caller that knows a whole launch lies beyond the limit (max |hertz| and the position are host-side
 * knowledge) calls this entry instead: the walker's exact-phase path runs at about a third of the closed form's rate. */
int sig_fused_voice_bus_walk(int osc_kind, int filt_type, int32_t rate, int64_t position,
                             int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                             const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                             const double* cutoff, int32_t cutoff_stride,
                             const double* gain, int32_t gain_stride,
                             const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                             double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);
/* sig_fused_voice_bus with the per-voice constants of its Sine closed form (filter design, H(e^{j theta}), the
 * block-start matrices: ~5 us of kernel time per call) kept by the caller across calls: `consts` is a device buffer of
 * sig_fused_voice_consts_size(voices) bytes; consts_ready == 0 fills it (and uses it), consts_ready != 0 skips that.
 * The buffer stays valid while hertz, cutoff, gain (their values), filt_type, rate, context, voices and
 * min(context, position) are unchanged; the caller vouches for that.  Other waveforms ignore it. */
int64_t sig_fused_voice_consts_size(int32_t voices);
int sig_fused_voice_bus_prepared(int osc_kind, int filt_type, int32_t rate, int64_t position,
                                 int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                                 const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                                 const double* cutoff, int32_t cutoff_stride,
                                 const double* gain, int32_t gain_stride,
                                 const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                                 double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream,
                                 double* consts, int32_t consts_ready);

/* sig_fused_voice_bus_prepared (walk == 0) / sig_fused_voice_bus_walk (walk != 0) with every argument that does not change from
 * batch to batch in a caller-held block: what a host binding with a per-argument cost (ctypes: ~5 us for 26 arguments, a
 * quarter of a 256-block batch) calls per batch.  `consts` may be NULL when only the walker is ever asked for. */
typedef struct {
    int32_t osc_kind, filt_type, rate, block_frames, nblocks, context, voices;
    int32_t hertz_stride, phase_stride, cutoff_stride, gain_stride, bus_channels;
    const double* hertz; const double* phase; const double* cutoff; const double* gain; const double* bus_gains;
    int64_t bus_gains_ld, out_ld;
    double* workspace; int32_t* status; double* consts;
} sig_fused_voice_bus_call;
int sig_fused_voice_bus_bound(const sig_fused_voice_bus_call* call, int64_t position, float* out, int32_t consts_ready,
                              int32_t walk, void* stream);

/* Fused filter cascade + envelope + sum bus (BASELINE config 3's whole graph in one launch):
 *   out[n,c] = sum_v bus_gains[c,v] * [gain[v] *] [ADSR_v(n) *] Filter2(Filter1(Osc))[n,v]
 * i.e. Osc._eval (chain/osc.py:26-62), two CritFilter._filter (chain/fx.py:85-121) in series with the reference's
 * block-cache semantics between them (chain/__init__.py:431-442, SURVEY.md 8a A9: the outer filter's 100 context rows
 * are the LAST 100 ROWS OF THE INNER FILTER'S PREVIOUS BLOCK, which was cold-started 100 rows before that block),
 * RingMod with the build-defined ADSR (adsr_params == NULL: none; layout as sig_adsr) and the build-defined SumBus.
 * `first_history_start`: the frame at which the block in front of `position` started when the stream was rendered
 * sequentially -- position - previous_block_frames on a continuing stream, position - min(context, position) on a
 * fresh graph (the reference then renders [position - 100, position) as a block of its own), == position exactly when
 * position == 0.  Requires block_frames > context and block_frames % (16 / bus_channels) == 0; cutoffs are one
 * (1,V)|(1,1) row each.  `workspace`: sig_fused_voice_bus_workspace(voices, rows, bus_channels) bytes.
 * Deterministic (fixed summation order, no atomics). */
int sig_fused_cascade_bus(int osc_kind, int filt1_type, int filt2_type, int32_t rate, int64_t position,
                          int64_t first_history_start, int32_t block_frames, int32_t nblocks, int32_t context,
                          int32_t voices,
                          const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                          const double* cutoff1, int32_t cutoff1_stride, const double* cutoff2, int32_t cutoff2_stride,
                          const double* gain, int32_t gain_stride,
                          const double* const* adsr_params, const int32_t* adsr_strides,
                          const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                          double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);
/* Introspection: voices per lane and consecutive blocks per lane sig_fused_cascade_bus uses (every span of blocks walks
 * one extra block of oscillator + inner filter, the history; bench.py's operation count depends on it). */
int sig_fused_cascade_geometry(int32_t voices, int32_t nblocks, int32_t* voices_per_lane, int32_t* blocks_per_lane);
/* Tuning / test hook (process-wide): force voices per lane (1, 2, 4; 0 = heuristic) and blocks per lane (>= 1; 0 = heuristic;
 * given as its negative: that many, and the voice tiles are added by a second launch instead of inside the kernel). */
int sig_fused_cascade_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane);

/* Latency mode of the same graph for a Sine oscillator: ONE block per launch, rows x voices parallelism (closed form
 * seeded per 16-row chunk), the voice tiles added and the float32 bus written by the last workgroup to finish -- a
 * single launch per block.  `workspace`: device, sig_latency_voice_bus_workspace(voices, block_frames, bus_channels)
 * bytes, whose last 8 bytes (the arrival counter) must be zero before the FIRST launch; the kernel re-arms it, so
 * launches that share a workspace must be ordered (one stream, or events between streams).
 * position_dev != NULL: the block's position is read from device memory and advanced by block_frames by the same
 * launch (hipGraph replay with no other node); otherwise `position` is used. */
int64_t sig_latency_voice_bus_workspace(int32_t voices, int32_t block_frames, int32_t bus_channels);
int sig_latency_voice_bus(int filt_type, int32_t rate, int64_t position, int64_t* position_dev,
                          int32_t block_frames, int32_t context, int32_t voices,
                          const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                          const double* cutoff, int32_t cutoff_stride,
                          const double* gain, int32_t gain_stride,
                          const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                          double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* Introspection: the launch geometry the two fused entry points use for this problem size -- voices per lane
 * (1, 2 or 4) and consecutive blocks per lane (the span walker, fused_voice.hip).  For measurement tools (the
 * f64 operation count per voice-sample depends on the span) and tests; no device work. */
int sig_fused_geometry(int32_t voices, int32_t block_frames, int32_t nblocks, int32_t context,
                       int32_t* voices_per_lane, int32_t* blocks_per_lane);
/* Introspection, no device work: what sig_fused_voice_bus launches for this problem -- voices per lane (1, 2, 4 or 8),
 * consecutive blocks per lane, and closed_form = 1 when the Sine closed-form kernel (fused_steady_bus_kernel) takes
 * the launch, 0 for the row-by-row span walker (fused_walk_kernel).  bench.py and the tests name the kernel they
 * time / check with it. */
int sig_fused_voice_bus_plan(int osc_kind, int64_t position, int32_t voices, int32_t block_frames, int32_t nblocks,
                             int32_t context, int32_t* voices_per_lane, int32_t* blocks_per_lane, int32_t* closed_form);
/* Tuning / test hook of the fused entry points: force the voices per lane (0 = heuristic), the blocks per lane
 * (0 = heuristic), the Sine closed form (-1 = heuristic, 0 = off, 1 = on, 2 = on with the voice tiles added by a second
 * launch instead of inside the kernel, 3 = on with the MixMatrix sink on the float32 MFMA instead of three-way bfloat16
 * splits) and the latency-mode prefix-scan kernel
 * (-1 = heuristic, 0 = off, 1 = on).  Process-wide, not thread-safe against concurrent launches; the initial values
 * come from SIG_FUSED_VPT / _SPAN / _STEADY / _SCAN, read once. */
int sig_fused_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane, int32_t closed_form, int32_t scan);
int sig_fused_voice_bus(int osc_kind, int filt_type, int32_t rate, int64_t position,
                        int32_t block_frames, int32_t nblocks, int32_t context, int32_t voices,
                        const double* hertz, int32_t hertz_stride, const double* phase, int32_t phase_stride,
                        const double* cutoff, int32_t cutoff_stride,
                        const double* gain, int32_t gain_stride,
                        const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                        double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);

/* A whole frame-rate voice graph in ONE launch (voice_program.hip): the general form of the fused entry points above, for
 * the graph shapes none of them covers.  Replaces, for every voice of a graph at once, the bodies of
 *   Osc._eval + _osc (osc.py:26-62), CritFilter._filter / _get_sos for LowPass / HighPass (fx.py:85-121), Gain / Mix / RingMod /
 *   Amp._eval (fx.py:35-60), Fixed._eval as an audio operand (fixed.py:38-39), White._eval (noise.py:22-23), the build-defined
 *   ADSR, and the sink: the rows themselves or the build-defined SumBus
 * together with the block structure between them: every node reads its control ports once per block at the block's position
 * (BoundPort.forward_at_block_rate, chain/__init__.py:305-306); a filter answers a block from zero state over
 * [<= context rows | block] (fx.py:93-105, forward_with_context chain/__init__.py:308-315); the context rows in front of block j
 * are the input's rows of block j - 1 as its block cache holds them (BlockCachingEmitter, chain/__init__.py:431-442), or -- on a
 * fresh graph, and for EVERY block when blocks are shorter than the context -- the input rendered as a block of its own
 * [p - context, p) with its controls read at max(p - context, 0).
 *
 * `program` (HOST memory, copied into the launch): the per-voice graph as straight-line code for an accumulator machine.
 *   OSC    acc = wave_kind(n / rate * hertz[a] + phase[a])         oscillator slot a (hertz / phase rows below)
 *   FILTER acc = filter slot a applied to acc                      (cutoff rows / type below; one slot per filter node)
 *   GAIN   acc = acc * params[a]          AMP  acc = copysign(acc ** params[a], acc)          CONST  acc = params[a]
 *   MUL    acc = temp[a] * acc            MIX  acc = m L + (1 - m) R, m = params[b], (L, R) = c ? (acc, temp[a]) : (temp[a], acc)
 *   SAVE   temp[a] = acc                  LOAD acc = temp[a]
 *   ADSR   acc = envelope level at n / rate (the six rows `adsr`)     NOISE acc = White sample (noise_seed[a], frame n, channel)
 * The accumulator after the last instruction is the voice's sample of that row.  Rows (sig_vp_rows) are float64 (rows, voices | 1)
 * arrays, col_stride 1 | 0: rows == 1 holds for every block; otherwise rows == control_rows, one row per block:
 *   block_frames >= context:  [the block in front of the first history block | hist_blocks history blocks | nblocks blocks],
 *                             control_rows = hist_blocks + 1 + nblocks
 *   block_frames <  context:  [nblocks virtual blocks (controls read at max(p_b - context, 0)) | nblocks blocks], control_rows = 2 nblocks.
 *                             There the rows a block's LAST filter reads over the block itself are the oldest cached reply of its input that
 *                             contains them (chain/__init__.py:435-442): the reply to the `after` request made m_b = min((context -
 *                             block_frames) / block_frames, blocks_before + b - 1) blocks earlier, at q_b = p_b - m_b block_frames (block 0 of
 *                             a fresh graph: q = p) -- every node in front of the last filter read its controls THERE, so the caller
 *                             evaluates their rows of the second group at q_b, those of the last filter and behind it at p_b.
 *                             `blocks_before`: blocks of this size rendered contiguously in front of the launch since the graph was fresh.
 *                             16 <= block_frames (below that the reference's 16-entry block cache evicts what a block reads), depth <= 2.
 * `depth` = filters in series on the longest path to the sink.  `hist_positions[hist_blocks]` (ascending, < position): where the
 * depth - 1 blocks in front of the launch start -- the previous render's blocks on a continuing stream, the virtual blocks
 * p - context, p - 2 context ... (clipped at 0) on a fresh graph; ignored when block_frames < context (at most two filters in
 * series there).  bus_channels 0: out (nblocks * block_frames, voices) float32; 1 | 2: out (.., bus_channels) = sum over voices of
 * bus_gains[c, v] * sample (bus_gains NULL: mono sum), float64 accumulation in a fixed order; workspace of
 * sig_fused_voice_bus_workspace(voices, rows, bus_channels) bytes.  f64 arithmetic, no float32 rounding between the nodes.
 * A rejected filter design (fx.py:99-102) gives NaN rows and sets SIG_STATUS_BAD_CUTOFF. */
enum { SIG_VP_OSC = 0, SIG_VP_FILTER = 1, SIG_VP_GAIN = 2, SIG_VP_MUL = 3, SIG_VP_MIX = 4, SIG_VP_SAVE = 5, SIG_VP_LOAD = 6,
       SIG_VP_CONST = 7, SIG_VP_AMP = 8, SIG_VP_ADSR = 9, SIG_VP_NOISE = 10 };
enum { SIG_VP_MAX_INS = 32, SIG_VP_MAX_OSCS = 4, SIG_VP_MAX_PARAMS = 8, SIG_VP_MAX_FILTERS = 4, SIG_VP_MAX_TEMPS = 4, SIG_VP_MAX_HIST = 3 };
typedef struct { int32_t op, kind, a, b, c; } sig_vp_ins;
typedef struct { const double* ptr; int32_t col_stride; int32_t rows; } sig_vp_rows;
typedef struct {
    int32_t n_ins; sig_vp_ins ins[SIG_VP_MAX_INS];
    int32_t n_oscs; sig_vp_rows hertz[SIG_VP_MAX_OSCS], phase[SIG_VP_MAX_OSCS];     /* phase.ptr NULL: unplugged = 0 */
    int32_t n_params; sig_vp_rows params[SIG_VP_MAX_PARAMS];
    int32_t n_filters; sig_vp_rows cutoff[SIG_VP_MAX_FILTERS]; int32_t filter_type[SIG_VP_MAX_FILTERS];
    int32_t filter_level[SIG_VP_MAX_FILTERS];                                       /* 1 + the filters in series in front of this one */
    int32_t n_temps, depth;
    const double* adsr[6]; int32_t adsr_stride[6];                                  /* attack decay sustain release gate_on gate_off */
    uint64_t noise_seed[2];
} sig_voice_program_t;
int sig_voice_program(const sig_voice_program_t* program, int32_t rate, int64_t position, int32_t block_frames,
                      int32_t nblocks, int32_t context, int32_t voices, int32_t control_rows,
                      int32_t hist_blocks, const int64_t* hist_positions, int32_t blocks_before,
                      const double* bus_gains, int64_t bus_gains_ld, int32_t bus_channels,
                      double* workspace, float* out, int64_t out_ld, int32_t* status, void* stream);
/* Tuning / test hook: force the voices per lane (1, 2; 0 = heuristic; ignored where the program does not fit the variant) and
 * the blocks per lane (0 = heuristic) of sig_voice_program.  Process-wide. */
int sig_voice_program_set_tuning(int32_t voices_per_lane, int32_t blocks_per_lane);
/* Introspection, no device work: the voices per lane and blocks per lane sig_voice_program picks for a problem.
 * store_aligned (bus_channels == 0): 4 = `out` 16-byte aligned with voices and out_ld multiples of 4, 2 = 8-byte aligned with
 * even voices and out_ld, else 1.  specialised != 0: as if a kernel specialised for four voices per lane were attached (the
 * interpreter runs one or two; sig_voice_program takes four where such an image is attached, the store is aligned for it or
 * there is a bus, and the launch still has a wave for every SIMD) -- what to build the image for. */
int sig_voice_program_geometry(int32_t voices, int32_t block_frames, int32_t nblocks, int32_t context, int32_t depth,
                               int32_t bus_channels, int32_t store_aligned, int32_t specialised,
                               int32_t* voices_per_lane, int32_t* blocks_per_lane);
/* SPECIALISED kernels.  The interpreter's source (signals_amd/csrc/voice_program.hip) built once more as a gfx950 code object
 * with ONE program as a compile-time constant -- macros SIG_VP_STATIC_CODE={words}, SIG_VP_S_NF / _NO / _NP / _NT / _EXT (the
 * register file: filter, oscillator, parameter, temporary slots; Amp / ADSR / White enabled), SIG_VP_STATIC_VPT, SIG_VP_STATIC_C,
 * SIG_VP_STATIC_WAVES; `hipcc --genco`, see signals_amd/specialise.py -- runs the same arithmetic as straight-line code (the
 * dispatch loop unrolls, every switch folds: 1.5-1.7x the interpreter).  sig_voice_program_attach hands such an image to the
 * library: it is loaded (hipModuleLoadData), asked what it was built for (its sig_vp_specialised_info kernel: the size of
 * the argument block, voices per lane, sink, program words -- anything else than THIS library's and the program given is
 * hipErrorInvalidImage) and from then on launched by sig_voice_program whenever it is called with the same words, slot counts
 * (n_oscs, n_params, n_filters, n_temps of `program`; its row pointers are not read), voices per lane and bus_channels.
 * A set-up call: allocates, launches and synchronises; not for a capturing stream.  sig_voice_program_use_attached(0) makes
 * every launch use the interpreter again (test hook), sig_voice_program_detach_all unloads the images (no launch may be in
 * flight).  sig_voice_program_args_size: sizeof of the kernels' argument block in this build. */
int sig_voice_program_attach(const sig_voice_program_t* program, int32_t voices_per_lane, int32_t bus_channels, const void* image);
int sig_voice_program_detach_all(void);
int sig_voice_program_use_attached(int32_t on);
int64_t sig_voice_program_args_size(void);

#ifdef __cplusplus
}
#endif
#endif /* SIGNALS_AMD_H */
