// The MixMatrix contraction of the fused kernels (fused_voice.hip, BASELINE config 5): 32 staged rows x 64 voices times
// the 64 x 64 matrix on the matrix cores, float32 in, float32 out.
//
// A float32 is the exact sum of three bfloat16 (8 + 8 + 8 significand bits): x = x0 + x1 + x2, m = m0 + m1 + m2, so
//     x m = x0 m0 + (x0 m1 + x1 m0) + (x0 m2 + x1 m1 + x2 m0) + [x1 m2 + x2 m1 + x2 m2 : below 2^-26 |x m|, dropped]
// with every kept product exact in the float32 accumulator.  Six v_mfma_f32_32x32x16_bf16 per k-block instead of eight
// v_mfma_f32_32x32x2_f32: 48 x 32 cycles per tile instead of 64 x 64 -- and, what matters more in a kernel whose other half
// is f64 vector work, the bf16 instruction holds the SIMD's vector issue for 8 of its 32 cycles only, while the f32 one
// holds it for all 64 (tools/ubench/mfma_gap_fill.hip: f32 MFMA + N v_fma_f64 costs the SUM, bf16 MFMA + N v_fma_f64 the
// MAXIMUM), so the recurrences of the SIMD's other wave run underneath.  Accumulation is float32 either way; within a
// k-block the small terms are added first.  The per-node kernel (mix_matrix.hip) is HBM-bound and keeps the f32 instruction.
#pragma once
#include "sig_common.h"

namespace sig_mix {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kTileRows = 32, kLdsStride = 68;     // rows per MFMA tile, floats per LDS row (64 + 4 pad)

// (a, b) -> three words of two bfloat16 each (a in the low half), a = a0 + a1 + a2 and b likewise
__device__ __forceinline__ void split_pair(float a, float b, unsigned& w0, unsigned& w1, unsigned& w2) {
    f32x2 v = {a, b};
    w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));          // v_cvt_pk_bf16_f32 (RNE)
    v.x -= __uint_as_float(w0 << 16); v.y -= __uint_as_float(w0 & 0xffff0000u);      // exact
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v.x -= __uint_as_float(w1 << 16); v.y -= __uint_as_float(w1 & 0xffff0000u);      // exact
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// eight consecutive k -> the three fragments of one k-block
__device__ __forceinline__ void split8(const float* x, bf16x8 (&f)[3]) {
    unsigned w[3][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) split_pair(x[2 * p], x[2 * p + 1], w[0][p], w[1][p], w[2][p]);
#pragma unroll
    for (int s = 0; s < 3; ++s) f[s] = __builtin_bit_cast(bf16x8, u32x4{w[s][0], w[s][1], w[s][2], w[s][3]});
}

// One wave's sink (its LDS region: two tiles of kTileRows * kLdsStride floats): rows arrive one at a time (lane = voice of the wave's 64-voice group), every 32 go through the matrix.
// Operand maps of v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md): lane l, r = l & 31, h = l >> 5, holds
// A[row r][k = 8 h + j] and B[k = 8 h + j][col r], j = 0 .. 7.  The k of k-block kb is voice 32 h + 8 kb + j -- any
// one-to-one assignment serves as long as A and B agree -- so a lane's A fragments are its row's half 32 h .. 32 h + 31
// (eight ds_read_b128, conflict-free on the padded tile).
// F32 = true: the same sink on v_mfma_f32_32x32x2_f32 (lane l holds A[row l & 31][k = l >> 5] and B[k = l >> 5][col l & 31]; 64
// of them per tile, the matrix as 64 float VGPRs) -- what mix_matrix.hip issues, kept selectable (sig_fused_set_tuning) so
// that BASELINE config 5 can also be timed on the instruction its name carries.
template <bool F32>
struct SinkT {
    bf16x8 bm[F32 ? 1 : 3][F32 ? 1 : 2][F32 ? 1 : 4];     // [part of m][column half][k-block]: 96 VGPRs
    float bmf[F32 ? 2 : 1][F32 ? 32 : 1];                 // F32: M[32 h + ks][32 jt + r]
    float* tile; float* tile_next; float* out; int64_t out_ld, row0; int lane, staged;

    __device__ __forceinline__ void init(const float* mix, float* tile_, float* out_, int64_t out_ld_, int64_t first_row, int lane_) {
        tile = tile_; tile_next = tile_ + kTileRows * kLdsStride; out = out_; out_ld = out_ld_; row0 = first_row; lane = lane_; staged = 0;
        const int r = lane & 31, h = lane >> 5;
        if constexpr (F32) {
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                bmf[0][ks] = mix[(32 * h + ks) * 64 + r];
                bmf[1][ks] = mix[(32 * h + ks) * 64 + 32 + r];
            }
            return;
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float m[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = mix[(32 * h + 8 * kb + j) * 64 + 32 * jt + r];
                bf16x8 f[3];
                split8(m, f);
#pragma unroll
                for (int s = 0; s < 3; ++s) bm[s][jt][kb] = f[s];
            }
    }
    // `between(kb)`: independent work of the caller (the next tile's rows, written with put_next) placed after k-block
    // kb's MFMAs in program order, for the scheduler to fill the MFMAs' shadow with
    template <typename F>
    __device__ __forceinline__ void flush(int nrows, F&& between) {
        const int r = lane & 31, h = lane >> 5;
        f32x16 acc0 = {0}, acc1 = {0};
        constexpr int kTerms[6][2] = {{0, 2}, {1, 1}, {2, 0}, {0, 1}, {1, 0}, {0, 0}};     // (part of x, part of m): small first
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {                                       // one k-block at a time: 12 fragment registers live
            float x[8];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(tile + r * kLdsStride + 32 * h + 8 * kb + 4 * c);
                x[4 * c] = t.x; x[4 * c + 1] = t.y; x[4 * c + 2] = t.z; x[4 * c + 3] = t.w;
            }
            if constexpr (F32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], bmf[0][8 * kb + j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], bmf[1][8 * kb + j], acc1, 0, 0, 0);
                }
            } else {
                bf16x8 ax[3];
                split8(x, ax);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[kTerms[t][0]], bm[kTerms[t][1]][0][kb], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[kTerms[t][0]], bm[kTerms[t][1]][1][kb], acc1, 0, 0, 0);
                }
            }
            between(kb);
        }
        // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  (Results staged back through the
        // tile and stored 16 B per lane measured the same: the stores are paced by HBM, not by their shape.)
        float* d = out + row0 * out_ld + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < nrows) {
                d[(int64_t)row * out_ld] = acc0[reg];
                d[(int64_t)row * out_ld + 32] = acc1[reg];
            }
        }
        row0 += nrows;
        staged = 0;
    }
    __device__ __forceinline__ void flush(int nrows) { flush(nrows, [](int) {}); }
    // double buffering: rows of the NEXT tile go to the other half of the wave's LDS region while this one is multiplied
    __device__ __forceinline__ void put_next(int k, float x) const { tile_next[k * kLdsStride + lane] = x; }
    __device__ __forceinline__ void swap() { float* t = tile; tile = tile_next; tile_next = t; }
    __device__ __forceinline__ void put(int k, float x) const { tile[k * kLdsStride + lane] = x; }   // row k of an empty tile, then flush()
    __device__ __forceinline__ void stage(float x) {
        tile[staged * kLdsStride + lane] = x;
        if (++staged == kTileRows) flush(kTileRows);
    }
    __device__ __forceinline__ void finish() { if (staged) flush(staged); }
};
using Sink = SinkT<false>;

}  // namespace sig_mix
