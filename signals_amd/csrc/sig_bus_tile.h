// Cross-lane bus sums for kernels whose lanes are voices (fused_voice.hip, biquad_bus.hip).
//   partial[tile][row][c] = sum over the wave's voices of weight[c][v] * y[row][v]
// A row's sum is a cross-lane sum; doing it per row with a butterfly would cost as much as the recurrences
// themselves, so rows are staged kPairs/C at a time in a wave-private LDS tile [pair = row*C + c][lane] (row stride
// 65 doubles: conflict-free for the transposed read) and reduced by lane = pair: 16 LDS reads + 2 lane-row swaps per
// lane per flush.  bus_partials_kernel then adds the voice tiles in a fixed order (deterministic, no atomics) and
// rounds to f32.
#pragma once
#include "sig_common.h"

namespace sig_bus {

constexpr int kPairs = 16;                 // (row, channel) pairs reduced per flush
constexpr int kTileStride = 65;            // doubles
constexpr int kTileDoubles = kPairs * kTileStride;

// The 16 lane-group partials of one (row, channel) pair added pairwise (depth 4), in this fixed order: a serial chain of 16
// dependent f64 adds costs a lone wave ~150 cycles per flush (measured: the closed-form bus kernel spent a quarter of
// its time there), the tree a quarter of that, and its independent adds interleave with the next rows' arithmetic.
__device__ __forceinline__ double sum16(const double (&pv)[16]) {
    const double a0 = pv[0] + pv[1], a1 = pv[2] + pv[3], a2 = pv[4] + pv[5], a3 = pv[6] + pv[7];
    const double a4 = pv[8] + pv[9], a5 = pv[10] + pv[11], a6 = pv[12] + pv[13], a7 = pv[14] + pv[15];
    const double b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7;
    return (b0 + b1) + (b2 + b3);
}

// the simple form (HBM-bound callers; the f64-issue-bound fused kernels keep their own software-pipelined variant):
// row k of a group of R = kPairs/C consecutive rows goes to tile row k (k is a compile-time constant in an unrolled
// loop), then one flush per group stores the rows of it that are wanted
template <int C>
struct Tile {
    static constexpr int R = kPairs / C;   // rows per flush
    double* tile; const double* col; double* dstp;
    int lane;

    __device__ __forceinline__ Tile(double* tile_, int lane_, double* dstp_)
        : tile(tile_), col(tile_ + (lane_ & (kPairs - 1)) * kTileStride + (lane_ >> 4) * 16), dstp(dstp_), lane(lane_) {}

    __device__ __forceinline__ void put(int k, const double (&acc)[C]) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) tile[(k * C + ch) * kTileStride + lane] = acc[ch];
    }
    // tile row k holds output row out_row0 + k; rows k_lo <= k < k_hi are stored.  AGENT: write-through (sc1) stores,
    // for partials that another workgroup of the SAME launch reads after an arrival counter (no L2 write-back fence)
    template <bool AGENT = false>
    __device__ __forceinline__ void flush(int64_t out_row0, int k_lo, int k_hi) {
        double pv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) pv[k] = col[k];
        const double s = sig_sum_rows_f64(sum16(pv));
        const int k = lane / C;
        if (lane < kPairs && k >= k_lo && k < k_hi) {
            if (AGENT) __hip_atomic_store(dstp + (out_row0 + k) * C + lane % C, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dstp[(out_row0 + k) * C + lane % C] = s;
        }
    }
};

// the software-pipelined form of the f64-issue-bound fused kernels (fused_voice.hip): rows are staged either one at a
// time (`put`, flushing when the tile is full) or in statically unrolled groups of R (`put_at` + `issue` / `finish`):
// the wave's LDS operations execute in order, so the 16 reads of a flush may be issued right after the group's last
// write and consumed a whole group of rows later, which hides their latency.
template <int C>
struct PipelinedTile {
    static constexpr int R = kPairs / C;   // rows per flush
    double* tile; const double* col; double* slot; double* dstp;
    int lane, staged;                      // rows in the tile (single-row mode)
    int64_t first;                         // output row of the first staged row

    __device__ __forceinline__ PipelinedTile(double* tile_, int lane_, double* dstp_, int64_t first_row)
        : tile(tile_), col(tile_ + (lane_ & (kPairs - 1)) * kTileStride + (lane_ >> 4) * 16), slot(tile_ + lane_),
          dstp(dstp_), lane(lane_), staged(0), first(first_row) {}

    __device__ __forceinline__ void issue(double (&pv)[16]) const {
#pragma unroll
        for (int k = 0; k < 16; ++k) pv[k] = col[k];
    }
    __device__ __forceinline__ void finish(const double (&pv)[16], int64_t row0, int nrows) const {
        const double s = sig_sum_rows_f64(sum16(pv));
        if (lane < nrows * C) dstp[row0 * C + lane] = s;
    }
    __device__ __forceinline__ void now() {                                    // flush what is staged, at once
        double pv[16];
        issue(pv);
        finish(pv, first, staged);
        first += staged;
        staged = 0;
        slot = tile + lane;
    }
    __device__ __forceinline__ double* at(int k) const { return tile + k * C * kTileStride + lane; }   // row k of a group
    __device__ __forceinline__ void advance() {                                // single-row mode: one row was written at `slot`
        slot += C * kTileStride;
        if (++staged == R) now();
    }
};

// The folded form, for kernels that produce a group's kPairs (row, channel) sums in registers (the closed-form bus
// kernel, whose only real cost besides its recurrences was this reduction: 16 LDS stores + 16 loads per lane per group
// take 13 + 8 cycles of the CU's LDS path each, shared by four SIMDs).  The sums are taken four at a time, in the
// order they are produced (pair index n = row * C + channel): two register-to-register halving steps
//     w = sig_fold16(sig_fold32(t0, t1), sig_fold32(t2, t3))
// leave in every lane a sum over the 4 lanes l % 16 == const of ONE of the four -- t0, t2, t1, t3 in the lane rows
// l / 16 = 0, 1, 2, 3 -- so only a quarter of the bytes goes through the LDS: tile[pair][l % 16], 16 x 17 doubles; then
// lane = pair reads quarter l / 16 of its row (4 doubles) and sig_sum_rows_f64 adds the quarters.  Fixed order.
// Shares the PipelinedTile's LDS region (its single-row mode serves the rows left over at a block's end): the wave's LDS
// operations execute in order, so reads issued here are served before any later store of the other form.
constexpr int kFoldStride = 17;            // doubles between pairs: odd, so the 16 lanes of a quarter read 16 different bank pairs

template <int C>
struct FoldedGroup {
    double* put; const double* get; double* dstp; int lane;
    __device__ __forceinline__ FoldedGroup(double* tile, int lane_, double* dstp_)
        : put(tile + ((((lane_ >> 4) & 1) << 1) | (lane_ >> 5)) * kFoldStride + (lane_ & 15)),
          get(tile + (lane_ & 15) * kFoldStride + 4 * (lane_ >> 4)), dstp(dstp_), lane(lane_) {}
    // the sums with pair indices 4 q .. 4 q + 3 of this lane
    __device__ __forceinline__ void fold4(int q, double t0, double t1, double t2, double t3) const {
        put[4 * q * kFoldStride] = sig_fold16(sig_fold32(t0, t1), sig_fold32(t2, t3));
    }
    __device__ __forceinline__ void issue(double (&pv)[4]) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) pv[t] = get[t];
    }
    __device__ __forceinline__ void finish(const double (&pv)[4], int64_t row0, int nrows) const {
        const double s = sig_sum_rows_f64((pv[0] + pv[1]) + (pv[2] + pv[3]));
        if (lane < nrows * C) dstp[row0 * C + lane] = s;
    }
};

// The tile sum inside the producing kernel, for launches whose wave index is  item = 4 * workgroup + wave,  tile =
// item % tiles,  span group = item / tiles  with tiles in {1, 2, 4}: the waves that hold one span group's tiles then sit in
// ONE workgroup, so a workgroup barrier is all the hand-off needed (their partial stores, drained by vmcnt(0) before
// the barrier, have left the CU through its write-through L1) -- no second launch, no ticket.  Wave t of the
// group adds slice t of the group's rows, tiles in the same fixed order as partials_kernel: the same bits.  Every wave of
// the workgroup calls this, also those without a span (ended waves do not count at a hardware barrier, so returning
// early before it would be legal too; they simply find no rows).
constexpr bool tiles_sum_in_workgroup(int tiles) { return tiles == 1 || tiles == 2 || tiles == 4; }

template <int C>
__device__ __forceinline__ void sum_tiles_in_workgroup(const double* partials, int tiles, int64_t rows, int span, int K, int N,
                                                       float* out, int64_t out_ld, int lane, int wave)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int t = (int)(item % tiles);
    const int64_t b_first = (item / tiles) * span;
    if (b_first >= K) return;
    const int nb = (int)((K - b_first < (int64_t)span) ? K - b_first : (int64_t)span);
    const int64_t n = (int64_t)nb * N * C, first = b_first * N * C, per = (n + tiles - 1) / tiles;
    const int64_t lo = first + t * per, hi = (first + n < lo + per) ? first + n : lo + per;
    // (plain loads: the producers are waves of this workgroup, whose stores went through this CU's write-through L1)
    constexpr int U = 8;                                                       // loads in flight per lane and tile
    int64_t i = lo + lane;
    for (; i + (U - 1) * SIG_WAVE < hi; i += U * SIG_WAVE) {
        double s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = 0.0;
        for (int k = 0; k < tiles; ++k) {                                      // fixed order
            const double* src = partials + (int64_t)k * rows * C + i;
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += src[u * SIG_WAVE];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * SIG_WAVE;
            out[(j / C) * out_ld + (j % C)] = (float)s[u];
        }
    }
    for (; i < hi; i += SIG_WAVE) {
        double s = 0.0;
        for (int k = 0; k < tiles; ++k) s += partials[(int64_t)k * rows * C + i];
        out[(i / C) * out_ld + (i % C)] = (float)s;
    }
}

template <int C>
static __global__ __launch_bounds__(256) void partials_kernel(const double* __restrict__ partials, int tiles, int64_t rows,
                                                               float* __restrict__ out, int64_t out_ld)
{
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += partials[(int64_t)t * n + i];       // fixed order
        out[(i / C) * out_ld + (i % C)] = (float)s;
    }
}

template <int C>
static int launch_partials(const double* partials, int tiles, int64_t rows, float* out, int64_t out_ld, hipStream_t stream)
{
    const int64_t n = rows * C;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    partials_kernel<C><<<(unsigned)g, 256, 0, stream>>>(partials, tiles, rows, out, out_ld);
    return sig_launch_status();
}

}  // namespace sig_bus
