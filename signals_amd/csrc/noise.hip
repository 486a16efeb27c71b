// White noise for gfx950: uniform [0,1) per (frame, channel), replacing White._eval
// (reference src/signals/chain/noise.py:22-23, np.random.rand on the global unseeded RNG).
// Counter-based: one 64-bit mix per PAIR of adjacent channels (high and low words), so a block is
// the same whatever launch geometry or position batching produced it.  HBM-write-bound (4 B/sample).
#include "sig_common.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t noise_bits(uint64_t seed, int64_t frame, int channel) {
    const uint64_t h = mix64(seed + (uint64_t)frame * 0x9E3779B97F4A7C15ULL + (uint64_t)(channel >> 1) * 0xD1B54A32D192ED03ULL);
    return (channel & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
}

template <typename OUT>
__global__ __launch_bounds__(256) void white_kernel(uint64_t seed, int64_t position, int64_t rows, int channels,
                                                    OUT* __restrict__ out, int64_t ld)
{
    const int64_t total = rows * channels;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / channels;
        const int c = (int)(i - r * channels);
        const uint32_t k = noise_bits(seed, position + r, c) >> 8;           // 24 bits
        out[r * ld + c] = (OUT)((float)k * 5.9604644775390625e-8f);          // k * 2^-24, exact in f32
    }
}

}  // namespace

extern "C" int sig_white_noise(uint64_t seed, int64_t position, int64_t rows, int32_t channels,
                               void* out, int32_t out_dtype, int64_t out_ld, void* stream)
{
    SIG_CHECK_ARG(position >= 0 && rows >= 0 && channels >= 0 && out != nullptr && out_ld >= channels);
    if (rows == 0 || channels == 0) return 0;
    const int64_t total = rows * channels;
    int64_t nwg = (total + 255) / 256;
    if (nwg > 2048 * 16) nwg = 2048 * 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == SIG_F32)
        white_kernel<float><<<(unsigned)nwg, 256, 0, s>>>(seed, position, rows, channels, static_cast<float*>(out), out_ld);
    else if (out_dtype == SIG_F64)
        white_kernel<double><<<(unsigned)nwg, 256, 0, s>>>(seed, position, rows, channels, static_cast<double*>(out), out_ld);
    else
        return (int)hipErrorInvalidValue;
    return sig_launch_status();
}
