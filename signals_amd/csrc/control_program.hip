// A whole block-rate control subgraph in ONE launch, gfx950.
//
// The reference reads a control port once per block, at the block's position (BoundPort.forward_at_block_rate,
// chain/__init__.py:305-306): an LFO on a cutoff is an oscillator evaluated at one frame per block, scaled and offset by
// element-wise nodes (osc.py:26-62, fx.py:35-60).  Node by node that is a handful of launches of a few microseconds of
// work each -- a vibrato + cutoff sweep + tremolo voice spent half of its batch time (and all of its host time) in eleven
// of them.  Here the subgraph is a short straight-line program over registers, the same for every (block, column): thread
// (b, v) evaluates it for block b at column v and writes the requested registers to their (nblocks, cols) outputs.
// Arithmetic as in the per-node kernels (osc_bank.hip's f64 store path, elementwise.hip's ew_apply): the same expressions
// under -ffp-contract=off, so the same bits.  Registers live in LDS, [register][thread]: every thread executes the same
// instruction, so a register index is uniform and the accesses are conflict-free (a private array indexed by a run-time
// value would go to scratch).  Instruction k writes register dst < n_ins (the host assigns them in instruction order).
#include <cstring>
#include <mutex>
#include <vector>

#include "sig_osc.h"

#ifdef SIG_CTL_STATIC_INS
// A SPECIALISED build of this file (signals_amd/specialise.py: hipcc --genco with the program's structure as macros): the
// registers are VGPRs, the interpretive loop and its chain of dependent LDS accesses are gone.  SIG_CTL_STATIC_INS = {{op, kind,
// a, b, c, dst, wide}, ...} in evaluation order, SIG_CTL_STATIC_OUTS = {{reg, wide}, ...}; row pointers, strides and output
// pointers are still read from the run-time `program` / `outs` arrays (uniform loads).  The same expressions as the interpreter
// below, so the same bits.  One workgroup per block: the one-column instructions once per thread, the wide ones per chunk of
// 256 columns.
namespace {
struct SIns { int op, kind, a, b, c, dst, wide; };
struct SOut { int reg, wide; };
constexpr SIns kIns[] = SIG_CTL_STATIC_INS;
constexpr SOut kOuts[] = SIG_CTL_STATIC_OUTS;
constexpr int kN = (int)(sizeof(kIns) / sizeof(kIns[0])), kNO = (int)(sizeof(kOuts) / sizeof(kOuts[0]));
constexpr int kSpecThreads = 256;
}  // namespace

extern "C" __global__ __launch_bounds__(kSpecThreads) void sig_ctl_specialised(double rate, int64_t position, int64_t step, int nblocks, int cols,
                                                                               int64_t front_position, int64_t min_position,
                                                                               const sig_ctl_ins* __restrict__ program, int n_ins,
                                                                               const sig_ctl_out* __restrict__ outs, int n_outs)
{
    const bool front = front_position >= 0 && blockIdx.x == (unsigned)nblocks;
    const int64_t b = front ? 0 : blockIdx.x;
    if (front) { position = front_position; step = 0; }
    double reg[kN];
#pragma unroll
    for (int k = 0; k < kN; ++k) reg[k] = 0.0;
    auto run = [&](int v, bool wide_pass) {
#pragma unroll
        for (int k = 0; k < kN; ++k) {
            const SIns I = kIns[k];
            if ((I.wide != 0) != wide_pass) continue;
            auto get = [&](int r) { return r < 0 ? 0.0 : reg[r]; };
            double x;
            switch (I.op) {
                case SIG_CTL_ROW: {
                    const sig_ctl_ins& ins = program[k];
                    x = ins.row[(ins.rows > 1 ? b * (int64_t)(ins.stride ? ins.cols : 1) : 0) + (int64_t)(v < ins.cols ? v : 0) * ins.stride];
                    break;
                }
                case SIG_CTL_OSC: {
                    int64_t frame = position + b * step;
                    if (!front && frame < min_position) frame = min_position;
                    const double t = (double)frame / rate * get(I.a) + get(I.b);          // osc.py:32
                    switch (I.kind) {
                        case SIG_OSC_SINE: x = sig_osc::osc_sine(t); break;
                        case SIG_OSC_SQUARE: x = sig_osc::osc_square(t); break;
                        case SIG_OSC_SAWTOOTH: x = sig_osc::osc_sawtooth(t); break;
                        default: x = sig_osc::osc_triangle(t); break;
                    }
                    break;
                }
                case SIG_CTL_MIX: { const double c = get(I.c); x = c * get(I.a) + (1.0 - c) * get(I.b); break; }   // fx.py:40
                case SIG_CTL_AMP: { const double a = get(I.a); x = copysign(pow(a, get(I.b)), a); break; }           // fx.py:60
                default: x = get(I.a) * get(I.b); break;                                                           // Gain, RingMod: fx.py:46, :52
            }
            reg[I.dst] = x;
        }
    };
    run(0, false);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kNO; ++k)
            if (!kOuts[k].wide) {
                if (!front) outs[k].out[b] = reg[kOuts[k].reg];
                else if (outs[k].front) outs[k].front[0] = reg[kOuts[k].reg];
            }
    }
    for (int v0 = 0; v0 < cols; v0 += kSpecThreads) {
        const int v = v0 + threadIdx.x;
        run(v, true);
#pragma unroll
        for (int k = 0; k < kNO; ++k)
            if (kOuts[k].wide) {
                const sig_ctl_out o = outs[k];
                if (v < o.cols) {
                    if (!front) o.out[b * o.cols + v] = reg[kOuts[k].reg];
                    else if (o.front) o.front[v] = reg[kOuts[k].reg];
                }
            }
    }
}
// what the attaching library checks: the structure the image was built for
extern "C" __global__ void sig_ctl_specialised_info(int32_t* out)
{
    out[0] = kN; out[1] = kNO;
    for (int k = 0; k < kN; ++k) {
        int32_t* w = out + 2 + 7 * k;
        w[0] = kIns[k].op; w[1] = kIns[k].kind; w[2] = kIns[k].a; w[3] = kIns[k].b; w[4] = kIns[k].c; w[5] = kIns[k].dst; w[6] = kIns[k].wide;
    }
    for (int k = 0; k < kNO; ++k) { out[2 + 7 * kN + 2 * k] = kOuts[k].reg; out[2 + 7 * kN + 2 * k + 1] = kOuts[k].wide; }
}
#else

namespace {

constexpr int kMaxRegs = SIG_CTL_MAX_REGS;

constexpr int kThreads = 128;

// One workgroup per block.  The program is copied into LDS once (fetching every instruction from global memory cost a
// dependent scalar load of ~0.5 us per instruction per wave); the register file behind it is sized by the program (n_regs x
// 128 threads x 8 B, dynamic LDS).  Instructions whose result is one column wide (an LFO: oscillator, scale, offset) run
// ONCE per block, before the loop over the columns; only the wide ones (the final products with per-voice rows) run per
// column chunk -- a vibrato + sweep + tremolo program over 1024 blocks x 1024 voices took 65 us with every (block, column)
// thread running all 26 instructions, three f64 sines among them.
__global__ __launch_bounds__(kThreads) void control_program_kernel(double rate, int64_t position, int64_t step, int nblocks, int cols,
                                                                   int64_t front_position, int64_t min_position,
                                                                   const sig_ctl_ins* __restrict__ program, int n_ins,
                                                                   const sig_ctl_out* __restrict__ outs, int n_outs)
{
    __shared__ sig_ctl_ins prog[SIG_CTL_MAX_INS];
    extern __shared__ double regs[];                                           // [register][thread]
    {
        const int words = n_ins * (int)(sizeof(sig_ctl_ins) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(program);
        uint32_t* dst = reinterpret_cast<uint32_t*>(prog);
        for (int w = threadIdx.x; w < words; w += kThreads) dst[w] = src[w];
    }
    __syncthreads();
    // the last workgroup of a launch with a front position evaluates the program ONCE more, at that position, into the
    // outputs' `front` rows (the controls of the block in front of the batch: sig_fused_*_fm's *_hist)
    const bool front = front_position >= 0 && blockIdx.x == (unsigned)nblocks;
    const int64_t b = front ? 0 : blockIdx.x;
    if (front) { position = front_position; step = 0; }
    double* r = regs + threadIdx.x;
    auto get = [&](int reg) { return reg < 0 ? 0.0 : r[reg * kThreads]; };
    auto row_value = [&](const sig_ctl_ins& ins, int v) {
        return ins.row[(ins.rows > 1 ? b * (int64_t)(ins.stride ? ins.cols : 1) : 0) + (int64_t)(v < ins.cols ? v : 0) * ins.stride];
    };
    auto execute = [&](const sig_ctl_ins& ins, int v) {
        double x;
        switch (ins.op) {
            case SIG_CTL_ROW: x = row_value(ins, v); break;
            case SIG_CTL_OSC: {
                int64_t frame = position + b * step;
                if (!front && frame < min_position) frame = min_position;      // (the first blocks of a run that starts before min_position are evaluated there)
                const double t = (double)frame / rate * get(ins.a) + get(ins.b);      // osc.py:32
                switch (ins.kind) {
                    case SIG_OSC_SINE: x = sig_osc::osc_sine(t); break;
                    case SIG_OSC_SQUARE: x = sig_osc::osc_square(t); break;
                    case SIG_OSC_SAWTOOTH: x = sig_osc::osc_sawtooth(t); break;
                    default: x = sig_osc::osc_triangle(t); break;
                }
                break;
            }
            case SIG_CTL_MIX: { const double c = get(ins.c); x = c * get(ins.a) + (1.0 - c) * get(ins.b); break; }   // fx.py:40
            case SIG_CTL_AMP: { const double a = get(ins.a); x = copysign(pow(a, get(ins.b)), a); break; }           // fx.py:60
            default: x = get(ins.a) * get(ins.b); break;                                                           // Gain, RingMod: fx.py:46, :52
        }
        r[ins.dst * kThreads] = x;
    };
    // ---- one column wide: once per block (every thread computes the same value into its own copy of the register).
    // Leading ROW instructions of that kind keep their global loads four in flight.
    int k0 = 0;
    while (k0 < n_ins && prog[k0].op == SIG_CTL_ROW && prog[k0].cols == 1) ++k0;
    for (int k = 0; k < k0; k += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = (k + u < k0) ? row_value(prog[k + u], 0) : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k + u < k0) r[prog[k + u].dst * kThreads] = x[u];
    }
    for (int k = k0; k < n_ins; ++k)
        if (prog[k].cols == 1) execute(prog[k], 0);
    if (threadIdx.x == 0)
        for (int k = 0; k < n_outs; ++k)
            if (outs[k].cols == 1) {
                if (!front) outs[k].out[b] = r[outs[k].reg * kThreads];
                else if (outs[k].front) outs[k].front[0] = r[outs[k].reg * kThreads];
            }
    // ---- wider: per chunk of 128 columns
    for (int v0 = 0; v0 < cols; v0 += kThreads) {
        const int v = v0 + threadIdx.x;
        for (int k = k0; k < n_ins; ++k)
            if (prog[k].cols > 1) execute(prog[k], v);
        for (int k = 0; k < n_outs; ++k) {
            const sig_ctl_out o = outs[k];
            if (o.cols > 1 && v < o.cols) {
                if (!front) o.out[b * o.cols + v] = r[o.reg * kThreads];
                else if (o.front) o.front[v] = r[o.reg * kThreads];
            }
        }
    }
}

}  // namespace

extern "C" int sig_control_program(int32_t rate, int64_t position, int32_t step, int32_t nblocks, int32_t cols,
                                   int64_t front_position, int64_t min_position,
                                   const sig_ctl_ins* program, int32_t n_ins, const sig_ctl_out* outs, int32_t n_outs, void* stream)
{
    SIG_CHECK_ARG(rate > 0 && step >= 0 && nblocks >= 0 && cols >= 1 && n_ins >= 0 && n_outs >= 0 && front_position >= -1 && min_position >= 0);
    SIG_CHECK_ARG((program || n_ins == 0) && (outs || n_outs == 0) && n_ins <= SIG_CTL_MAX_INS);
    if ((nblocks == 0 && front_position < 0) || n_outs == 0) return 0;
    // one register per instruction (dst < n_ins): the program lives in device memory, so register indices cannot be checked
    // here -- the LDS register file is sized by n_ins, and n_ins <= SIG_CTL_MAX_INS == SIG_CTL_MAX_REGS was checked above
    static_assert(SIG_CTL_MAX_INS <= SIG_CTL_MAX_REGS, "the register file is sized by the instruction count");
    const int n_regs = n_ins > 0 ? n_ins : 1;
    control_program_kernel<<<(unsigned)(nblocks + (front_position >= 0 ? 1 : 0)), kThreads, (size_t)n_regs * kThreads * sizeof(double), static_cast<hipStream_t>(stream)>>>((double)rate, position, step, nblocks, cols, front_position, min_position,
                                                                                          program, n_ins, outs, n_outs);
    return sig_launch_status();
}

// ---- specialised builds of this file (see the top): attached at run time, launched through a handle.  The program itself is in
// device memory, so the library cannot match it against the image: the caller keeps the handle with the program it built the
// image from (signals_amd/engine.py: _ControlProgram).
namespace {
struct CtlSpecial { hipModule_t mod; hipFunction_t fn; };
std::vector<CtlSpecial>& ctl_specials() { static std::vector<CtlSpecial> v; return v; }
std::mutex& ctl_specials_lock() { static std::mutex m; return m; }
}  // namespace

extern "C" int sig_control_program_attach(const int32_t* description, int32_t n_words, const void* image, int32_t* handle)
{
    SIG_CHECK_ARG(description && image && handle && n_words >= 2 && n_words <= 2 + 7 * SIG_CTL_MAX_INS + 2 * 64);
    CtlSpecial e{};
    hipError_t err = hipModuleLoadData(&e.mod, image);
    if (err != hipSuccess) { (void)hipGetLastError(); return (int)err; }
    hipFunction_t info = nullptr;
    err = hipModuleGetFunction(&e.fn, e.mod, "sig_ctl_specialised");
    if (err == hipSuccess) err = hipModuleGetFunction(&info, e.mod, "sig_ctl_specialised_info");
    std::vector<int32_t> got((size_t)2 + 7 * SIG_CTL_MAX_INS + 2 * 64, 0);
    int32_t* dev = nullptr;
    if (err == hipSuccess) err = hipMalloc(&dev, got.size() * sizeof(int32_t));
    if (err == hipSuccess) {
        void* params[] = {&dev};
        err = hipModuleLaunchKernel(info, 1, 1, 1, 1, 1, 1, 0, nullptr, params, nullptr);
        if (err == hipSuccess) err = hipMemcpy(got.data(), dev, got.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
        (void)hipFree(dev);
    }
    if (err == hipSuccess && memcmp(got.data(), description, sizeof(int32_t) * (size_t)n_words) != 0) err = hipErrorInvalidImage;
    if (err == hipSuccess && 2 + 7 * got[0] + 2 * got[1] != n_words) err = hipErrorInvalidImage;
    if (err != hipSuccess) { (void)hipModuleUnload(e.mod); (void)hipGetLastError(); return (int)err; }
    std::lock_guard<std::mutex> g(ctl_specials_lock());
    ctl_specials().push_back(e);
    *handle = (int32_t)ctl_specials().size();
    return 0;
}

extern "C" int sig_control_program_attached(int32_t handle, int32_t rate, int64_t position, int32_t step, int32_t nblocks, int32_t cols,
                                            int64_t front_position, int64_t min_position,
                                            const sig_ctl_ins* program, int32_t n_ins, const sig_ctl_out* outs, int32_t n_outs, void* stream)
{
    SIG_CHECK_ARG(rate > 0 && step >= 0 && nblocks >= 0 && cols >= 1 && n_ins >= 0 && n_outs >= 0 && front_position >= -1 && min_position >= 0);
    SIG_CHECK_ARG((program || n_ins == 0) && (outs || n_outs == 0) && n_ins <= SIG_CTL_MAX_INS);
    hipFunction_t fn = nullptr;
    {
        std::lock_guard<std::mutex> g(ctl_specials_lock());
        SIG_CHECK_ARG(handle >= 1 && (size_t)handle <= ctl_specials().size());
        fn = ctl_specials()[(size_t)handle - 1].fn;
    }
    if ((nblocks == 0 && front_position < 0) || n_outs == 0) return 0;
    double rate_d = (double)rate;
    int64_t step64 = step;
    void* params[] = {&rate_d, &position, &step64, &nblocks, &cols, &front_position, &min_position, &program, &n_ins, &outs, &n_outs};
    return (int)hipModuleLaunchKernel(fn, (unsigned)(nblocks + (front_position >= 0 ? 1 : 0)), 1, 1, 256, 1, 1, 0,
                                      static_cast<hipStream_t>(stream), params, nullptr);
}
#endif  // SIG_CTL_STATIC_INS
