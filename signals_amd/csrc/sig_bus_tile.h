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
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += col[k];
        s = sig_sum_rows_f64(s);
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
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += pv[k];
        s = sig_sum_rows_f64(s);
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
