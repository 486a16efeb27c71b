// Dense mix-matrix node for gfx950 (build-defined, SURVEY.md §8a A11 / BASELINE config 5):
//   out[n, 64g : 64g+64] = x[n, 64g : 64g+64] @ M,   M (64 x 64) f32, shared by every group g.
// This is the one node on the path that is genuinely a dense contraction, so it runs on the matrix
// cores with the exact-f32 MFMA v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain, no reduced precision).
// 8 B of HBM traffic and 128 flop per voice-sample = 16 flop/B: below the f32-MFMA ridge, still HBM-bound.
//
// One wave = one 32-row x 64-voice tile:
//   1. 8 coalesced global_load_dwordx4 (each covers 4 rows x 256 B) -> wave-private LDS tile, rows
//      padded to 68 floats (272 B) so the 16-lane groups of ds_read_b128 hit 64 distinct banks;
//   2. 8 ds_read_b128 give lane (i = lane&31, h = lane>>5) the half-row x[i][32h .. 32h+31];
//      the contraction index is ordered k(ks, h) = 32h + ks so that register ks IS the A operand of
//      k-step ks (A[i][k] wants one float per lane);
//   3. M's B operands (M[32h+ks][32jt + (lane&31)]) live in 64 VGPRs for the wave's lifetime;
//   4. 2 x 32 MFMAs, then the 32x32 accumulators are stored as 128-B row segments.
#include <cstdlib>

#include "sig_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;     // native vector: loop-carried copies stay in registers
constexpr int kTileRows = 32;
constexpr int kGroup = 64;
constexpr int kLdsStride = 68;                         // floats per LDS row (64 + 4 pad)
constexpr int kWaves = 4;

__global__ __launch_bounds__(256) void mix_matrix_kernel(int64_t rows, int groups, const float* __restrict__ x, int64_t x_ld,
                                                         const float* __restrict__ m, float* __restrict__ out, int64_t out_ld,
                                                         int64_t row_tiles)
{
    __shared__ __attribute__((aligned(16))) float lds[kWaves][kTileRows * kLdsStride];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;

    // B operands: M[32h + ks][32 jt + i]
    float b[2][32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        b[0][ks] = m[(32 * h + ks) * kGroup + i];
        b[1][ks] = m[(32 * h + ks) * kGroup + 32 + i];
    }

    const int items = (int)(row_tiles * groups);
    float* tile = lds[wave];
    // 1. coalesced tile load: instr mm covers rows 4mm .. 4mm+3.  The NEXT tile's loads are issued before this tile's
    //    MFMAs, so their HBM latency (about as long as the 64 MFMAs) is spent under them.
    struct Rows8 { f32x4 r[8]; };
    auto load_tile = [&](int item) {                                           // items < 2^31 (host-checked): 32-bit index math
        Rows8 v;
        const int g = item % groups;
        const int64_t row0 = (int64_t)(item / groups) * kTileRows;
        const int last = (int)((rows - 1 - row0 < kTileRows - 1) ? rows - 1 - row0 : kTileRows - 1);
        const float* src = x + row0 * x_ld + g * kGroup + 4 * (lane & 15);
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
            const int r = (4 * mm + (lane >> 4) < last) ? 4 * mm + (lane >> 4) : last;   // clamp (tail tile)
            v.r[mm] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * x_ld);
        }
        return v;
    };
    const int stride = (int)gridDim.x * kWaves;
    int item = (int)blockIdx.x * kWaves + wave;
    Rows8 v = load_tile(item < items ? item : 0);
    for (; item < items; item += stride) {
        const int g = item % groups;
        const int64_t row0 = (int64_t)(item / groups) * kTileRows;
#pragma unroll
        for (int mm = 0; mm < 8; ++mm)
            *reinterpret_cast<f32x4*>(tile + (4 * mm + (lane >> 4)) * kLdsStride + 4 * (lane & 15)) = v.r[mm];
        // wave-private tile: no barrier, only the LDS counter (the compiler waits before the reads)
        // 2. half-row fragments
        float a[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(tile + i * kLdsStride + 32 * h + 4 * c);
            a[4 * c] = t.x; a[4 * c + 1] = t.y; a[4 * c + 2] = t.z; a[4 * c + 3] = t.w;
        }
        v = load_tile(item < items - stride ? item + stride : item);           // prefetch (v has been copied to LDS; the last one re-reads)
        // 3. out tile = A (32 x 64) @ M (64 x 64): two 32-column halves
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[0][ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[1][ks], acc1, 0, 0, 0);
        }
        // 4. C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        float* dst = out + row0 * out_ld + (int64_t)g * kGroup + i;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row0 + r < rows) {
                dst[(int64_t)r * out_ld] = acc0[reg];
                dst[(int64_t)r * out_ld + 32] = acc1[reg];
            }
        }
    }
}

}  // namespace

extern "C" int sig_mix_matrix(int64_t rows, int32_t voices, const float* x, int64_t x_ld,
                              const float* matrix /* (64,64) row-major, contiguous */,
                              float* out, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(rows >= 0 && voices >= 0 && voices % kGroup == 0 && x && matrix && out);
    SIG_CHECK_ARG(x_ld >= voices && out_ld >= voices && x_ld % 4 == 0);
    SIG_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (rows == 0 || voices == 0) return 0;
    const int groups = voices / kGroup;
    const int64_t row_tiles = (rows + kTileRows - 1) / kTileRows;
    const int64_t items = row_tiles * groups;
    SIG_CHECK_ARG(items < 0x7fffffffLL);
    int64_t nwg = (items + kWaves - 1) / kWaves;
    // persistent: the register budget (M's 64 B-operand VGPRs + A + accumulators) admits 2 workgroups per CU;
    // launch exactly that many so M is fetched once per wave and every wave strides over many tiles
    static const int64_t cap = [] { const char* e = getenv("SIG_MIXMAT_WGS"); return e ? atoll(e) : 512LL; }();
    if (nwg > cap) nwg = cap;
    mix_matrix_kernel<<<(unsigned)nwg, 256, 0, static_cast<hipStream_t>(stream)>>>(rows, groups, x, x_ld, matrix, out, out_ld, row_tiles);
    return sig_launch_status();
}
