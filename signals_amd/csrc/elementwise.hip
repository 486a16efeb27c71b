// Element-wise effects for gfx950: Gain / Mix / RingMod / Amp (reference src/signals/chain/fx.py:35-60).
// HBM-bound streaming kernels: 16-B lanes on the contiguous fast path, arithmetic in f64 so that the
// only error against the f64 reference is the f32 storage rounding of inputs and output.
// numpy broadcasting of (1,V) / (N,1) / (1,1) replies is expressed as zero strides.
#include "sig_common.h"

namespace {

struct Opnd { const void* p; int64_t rs; int32_t cs; int32_t f64; int32_t rd; };   // rd: rows sharing one operand row (block-rate operands)

__device__ __forceinline__ double ld_op(const Opnd& o, int64_t r, int v) {
    const int64_t i = (o.rd > 1 ? r / o.rd : r) * o.rs + (int64_t)v * o.cs;
    return o.f64 ? ((const double*)o.p)[i] : (double)((const float*)o.p)[i];
}

template <int OP>
__device__ __forceinline__ double ew_apply(double a, double b, double c) {
    if (OP == SIG_EW_GAIN || OP == SIG_EW_RINGMOD) return a * b;              // fx.py:46, :52
    if (OP == SIG_EW_MIX) return c * a + (1.0 - c) * b;                        // fx.py:40
    return copysign(pow(a, b), a);                                            // fx.py:60
}

// generic: any strides, any dtype mix
template <int OP, typename OUT>
__global__ __launch_bounds__(256) void ew_generic_kernel(int64_t rows, int cols, Opnd a, Opnd b, Opnd c,
                                                         OUT* __restrict__ out, int64_t out_ld)
{
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int v = (int)(i - r * cols);
        const double x = ld_op(a, r, v);
        const double y = ld_op(b, r, v);
        const double z = (OP == SIG_EW_MIX) ? ld_op(c, r, v) : 0.0;
        out[r * out_ld + v] = (OUT)ew_apply<OP>(x, y, z);
    }
}

// fast path: f32 (rows, cols) operands with unit column stride, per-voice f64 control row(s)
//   GAIN/AMP: a audio, b control row        RINGMOD: a, b audio        MIX: a, b audio, c control row
// thread = 4 consecutive voices x 4 rows (control values stay in registers across the rows)
template <int OP>
__global__ __launch_bounds__(256) void ew_fast_kernel(int64_t rows, int cols, Opnd a, Opnd b, Opnd c,
                                                      float* __restrict__ out, int64_t out_ld, int col_tiles)
{
    constexpr int R = 4;
    const int ct = blockIdx.x % col_tiles;
    const int64_t rt = blockIdx.x / col_tiles;
    const int v = (ct * 64 + (threadIdx.x & 63)) * 4;
    const int64_t r0 = (rt * 4 + (threadIdx.x >> 6)) * R;
    if (v >= cols || r0 >= rows) return;
    double ctl[4] = {0, 0, 0, 0};
    const Opnd& co = (OP == SIG_EW_MIX) ? c : b;
    if (OP != SIG_EW_RINGMOD) {
        // one control row per launch, or per block (rd = block_frames, a multiple of this thread's 4 rows)
        const int64_t crow = (co.rd > 1 ? r0 / co.rd : 0) * co.rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) ctl[i] = ((const double*)co.p)[crow + (int64_t)(v + i) * co.cs];
    }
    float4 xa[R], xb[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int64_t r = (r0 + j < rows) ? r0 + j : rows - 1;
        xa[j] = *reinterpret_cast<const float4*>((const float*)a.p + r * a.rs + v);
        if (OP == SIG_EW_RINGMOD || OP == SIG_EW_MIX)
            xb[j] = *reinterpret_cast<const float4*>((const float*)b.p + r * b.rs + v);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (r0 + j >= rows) break;
        const float av[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
        const float bv[4] = {xb[j].x, xb[j].y, xb[j].z, xb[j].w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double A = av[i];
            const double B = (OP == SIG_EW_GAIN || OP == SIG_EW_AMP) ? ctl[i] : (double)bv[i];
            o[i] = (float)ew_apply<OP>(A, B, ctl[i]);
        }
        *reinterpret_cast<float4*>(out + (r0 + j) * out_ld + v) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

inline bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
inline bool audio_fast(const Opnd& o, int cols) {
    return !o.f64 && o.cs == 1 && o.rs >= cols && o.rs % 4 == 0 && o.rd <= 1 && aligned16(o.p);
}
inline bool ctrl_fast(const Opnd& o) {
    return o.f64 && (o.cs == 0 || o.cs == 1) && (o.rs == 0 || (o.rd >= 4 && o.rd % 4 == 0));
}

template <int OP>
int launch_ew(int64_t rows, int cols, const Opnd& a, const Opnd& b, const Opnd& c,
              void* out, int64_t out_ld, int out_dtype, hipStream_t stream)
{
    bool fast = out_dtype == SIG_F32 && cols % 4 == 0 && out_ld % 4 == 0 && aligned16(out) && audio_fast(a, cols);
    if (OP == SIG_EW_GAIN || OP == SIG_EW_AMP) fast = fast && ctrl_fast(b);
    if (OP == SIG_EW_RINGMOD) fast = fast && audio_fast(b, cols);
    if (OP == SIG_EW_MIX) fast = fast && audio_fast(b, cols) && ctrl_fast(c);
    if (fast) {
        const int col_tiles = (cols + 255) / 256;
        const int64_t row_tiles = (rows + 15) / 16;
        const int64_t nwg = row_tiles * col_tiles;
        if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        ew_fast_kernel<OP><<<(unsigned)nwg, 256, 0, stream>>>(rows, cols, a, b, c, static_cast<float*>(out), out_ld, col_tiles);
    } else {
        const int64_t total = rows * cols;
        int64_t nwg = (total + 255) / 256;
        if (nwg > 2048 * 8) nwg = 2048 * 8;
        if (out_dtype == SIG_F32)
            ew_generic_kernel<OP, float><<<(unsigned)nwg, 256, 0, stream>>>(rows, cols, a, b, c, static_cast<float*>(out), out_ld);
        else
            ew_generic_kernel<OP, double><<<(unsigned)nwg, 256, 0, stream>>>(rows, cols, a, b, c, static_cast<double*>(out), out_ld);
    }
    return sig_launch_status();
}

inline bool load_operand(const sig_operand* s, Opnd& o) {
    if (!s || !s->ptr) return false;
    if (s->dtype != SIG_F32 && s->dtype != SIG_F64) return false;
    if (s->row_stride < 0 || s->col_stride < 0 || s->row_div < 0) return false;
    o.p = s->ptr; o.rs = s->row_stride; o.cs = s->col_stride; o.f64 = (s->dtype == SIG_F64); o.rd = s->row_div;
    return true;
}

}  // namespace

extern "C" int sig_elementwise(int op, int64_t rows, int32_t cols,
                               const sig_operand* a, const sig_operand* b, const sig_operand* c,
                               void* out, int64_t out_ld, int32_t out_dtype, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && cols >= 0 && out != nullptr && out_ld >= cols);
    SIG_CHECK_ARG(out_dtype == SIG_F32 || out_dtype == SIG_F64);
    Opnd A{}, B{}, C{};
    SIG_CHECK_ARG(load_operand(a, A) && load_operand(b, B));
    if (op == SIG_EW_MIX) SIG_CHECK_ARG(load_operand(c, C));
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (op) {
        case SIG_EW_GAIN: return launch_ew<SIG_EW_GAIN>(rows, cols, A, B, C, out, out_ld, out_dtype, s);
        case SIG_EW_MIX: return launch_ew<SIG_EW_MIX>(rows, cols, A, B, C, out, out_ld, out_dtype, s);
        case SIG_EW_RINGMOD: return launch_ew<SIG_EW_RINGMOD>(rows, cols, A, B, C, out, out_ld, out_dtype, s);
        case SIG_EW_AMP: return launch_ew<SIG_EW_AMP>(rows, cols, A, B, C, out, out_ld, out_dtype, s);
    }
    return (int)hipErrorInvalidValue;
}
