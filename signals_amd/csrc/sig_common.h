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

// Sum of a double over the wave's four 16-lane rows, delivered to every lane: what  s += shfl_xor(s, 16); s += shfl_xor(s, 32)
// computes -- the same additions in the same order, so the same bits -- but with gfx950's v_permlane16_swap /
// v_permlane32_swap (register-to-register, VALU latency) instead of two dependent LDS round trips (ds_bpermute).
// v_permlaneN_swap exchanges the odd N-lane groups of its first operand with the even groups of its second: fed two
// copies of x it returns (even groups twice, odd groups twice).
__device__ __forceinline__ double sig_sum_rows_f64(double s) {
    {
        const int lo = __double2loint(s), hi = __double2hiint(s);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        s = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);      // row pairs (0,1) and (2,3)
    }
    {
        const int lo = __double2loint(s), hi = __double2hiint(s);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        s = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);      // halves
    }
    return s;
}

// Halving steps of a cross-lane sum, two values at a time (gfx950's v_permlane32_swap / v_permlane16_swap exchange lane
// groups of two registers in one instruction, so no select is needed):
//   sig_fold32(a, b): lanes 0-31 get a[l] + a[l + 32], lanes 32-63 get b[l - 32] + b[l]
//   sig_fold16(a, b): rows (16-lane groups) R0, R2 get a summed over (R0, R1) resp. (R2, R3); rows R1, R3 the same of b
__device__ __forceinline__ double sig_fold32(double a, double b) {
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double sig_fold16(double a, double b) {
    const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}

// numpy's float mod for a positive power-of-two divisor d = 1/INV_D (npy_divmod semantics):
//   m = fmod(t, d)  (exact);  if (m != 0 && m < 0) m += d (ROUNDED, like numpy);  if (m == 0) m = +0
// Evaluated as  t - d * floor(t * INV_D):  d * floor(..) is exact, and for t < 0 the one rounded subtraction
// rounds the same real number (fmod + d) that numpy's rounded addition rounds, so the bits are the same; an exact
// multiple gives t - t = +0.  Two instructions (v_floor_f64, v_fma_f64 / v_add_f64) and no select.
template <int INV_D>
__device__ __forceinline__ double sig_npmod_pow2(double t) {
    if (INV_D == 1) return t - floor(t);
    return fma(floor(t * (double)INV_D), -1.0 / (double)INV_D, t);   // single rounding of the exact t - d * floor
}

__device__ __forceinline__ double sig_sign(double x) {
    // np.sign: -1, 0, +1, nan
    return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : ((x == 0.0) ? 0.0 : x));
}

template <typename T> __device__ __forceinline__ double sig_ld(const void* p, int64_t i);
template <> __device__ __forceinline__ double sig_ld<float>(const void* p, int64_t i) { return (double)((const float*)p)[i]; }
template <> __device__ __forceinline__ double sig_ld<double>(const void* p, int64_t i) { return ((const double*)p)[i]; }
