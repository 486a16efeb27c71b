// Shared device helpers for the gfx950 block-render kernels.
// Built with -ffp-contract=off: every `a*b+c` below is two roundings unless written fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/signals_amd.h"

#define SIG_WAVE 64

#define SIG_CHECK_ARG(cond) \
    do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)

static inline int sig_launch_status() { return (int)hipGetLastError(); }

template <typename T> struct sig_vec4;
template <> struct sig_vec4<float> { using type = float4; };
template <> struct sig_vec4<double> { using type = double4; };

__device__ __forceinline__ double sig_readlane_f64(double x, int lane) {
    // wave-uniform broadcast of one lane's double (lane must be wave-uniform)
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double sig_shfl_xor_f64(double x, int mask) {
    int lo = __shfl_xor(__double2loint(x), mask, SIG_WAVE);
    int hi = __shfl_xor(__double2hiint(x), mask, SIG_WAVE);
    return __hiloint2double(hi, lo);
}

// numpy's float mod for a positive power-of-two divisor (npy_divmod semantics):
//   m = fmod(t, d)  (exact);  if (m != 0 && m < 0) m += d (ROUNDED, like numpy);  if (m == 0) m = +0
// D2 = 1/d must make t*D2 exact, i.e. d in {1, 0.5}.
template <int INV_D>
__device__ __forceinline__ double sig_npmod_pow2(double t) {
    const double d = 1.0 / (double)INV_D;
    double m = t - d * trunc(t * (double)INV_D);   // exact: the result of fmod is representable
    if (m < 0.0) m += d;
    else if (m == 0.0) m = 0.0;                     // -0 -> +0 (copysign(0, d))
    return m;
}

__device__ __forceinline__ double sig_sign(double x) {
    // np.sign: -1, 0, +1, nan
    return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : ((x == 0.0) ? 0.0 : x));
}

template <typename T> __device__ __forceinline__ double sig_ld(const void* p, int64_t i);
template <> __device__ __forceinline__ double sig_ld<float>(const void* p, int64_t i) { return (double)((const float*)p)[i]; }
template <> __device__ __forceinline__ double sig_ld<double>(const void* p, int64_t i) { return ((const double*)p)[i]; }
