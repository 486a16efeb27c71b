// The C ABI without Python or torch: BASELINE config 2 (1024-voice sine -> Butterworth low-pass -> gain -> stereo
// bus, 48 kHz, 256-frame blocks) rendered with hipMalloc'ed buffers and two calls into libsignals_amd.so.
// Prints a checksum of the stereo bus; tests/test_abi_direct.py compares it with the Python engine's.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c2_direct.cpp -L signals_amd/csrc -lsignals_amd \
//         -Wl,-rpath,'$ORIGIN/../signals_amd/csrc' -o examples/c2_direct
//   examples/c2_direct [blocks] [position]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "signals_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// the deterministic parameter rows the test regenerates in numpy (a simple LCG, not numpy's default_rng)
static double lcg(uint64_t& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
    const int V = 1024, N = 256, C = 2, rate = 48000;
    const int K = argc > 1 ? std::atoi(argv[1]) : 64;
    const int64_t position = argc > 2 ? std::atoll(argv[2]) : 0;
    const int64_t rows = (int64_t)N * K;

    std::vector<double> hertz(V), phase(V), cutoff(V), gain(V), pan(C * (size_t)V);
    uint64_t s = 12345;
    for (int v = 0; v < V; ++v) {
        hertz[v] = 55.0 + 1705.0 * lcg(s);
        phase[v] = lcg(s);
        cutoff[v] = 200.0 + 7800.0 * lcg(s);
        gain[v] = lcg(s) / V;
        const double th = 1.5707963267948966 * lcg(s);
        pan[v] = std::cos(th);
        pan[V + v] = std::sin(th);
    }
    double *d_hz, *d_ph, *d_cut, *d_gain, *d_pan, *d_ws;
    float* d_bus;
    int32_t* d_status;
    HIP_OK(hipMalloc(&d_hz, V * 8)); HIP_OK(hipMalloc(&d_ph, V * 8)); HIP_OK(hipMalloc(&d_cut, V * 8));
    HIP_OK(hipMalloc(&d_gain, V * 8)); HIP_OK(hipMalloc(&d_pan, C * V * 8));
    HIP_OK(hipMalloc(&d_ws, sig_fused_voice_bus_workspace(V, rows, C)));
    HIP_OK(hipMalloc(&d_bus, rows * C * 4)); HIP_OK(hipMalloc(&d_status, 4));
    HIP_OK(hipMemcpy(d_hz, hertz.data(), V * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ph, phase.data(), V * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_cut, cutoff.data(), V * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_gain, gain.data(), V * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_pan, pan.data(), C * V * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_status, 0, 4));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    if (sig_abi_version() != SIG_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 3; }
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
    int err = 0;
    for (int rep = 0; rep < 2; ++rep) {                       // second pass is the timed one
        HIP_OK(hipEventRecord(t0, stream));
        err = sig_fused_voice_bus(SIG_OSC_SINE, SIG_FILT_LOWPASS, rate, position, N, K, 100, V,
                                  d_hz, 1, d_ph, 1, d_cut, 1, d_gain, 1, d_pan, V, C, d_ws, d_bus, C, d_status, stream);
        HIP_OK(hipEventRecord(t1, stream));
        if (err) { std::fprintf(stderr, "sig_fused_voice_bus: hipError_t %d\n", err); return 4; }
    }
    HIP_OK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, t0, t1));
    std::vector<float> bus(rows * C);
    int32_t status = 0;
    HIP_OK(hipMemcpy(bus.data(), d_bus, rows * C * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
    double sum = 0, sq = 0, peak = 0;
    for (float x : bus) { sum += x; sq += (double)x * x; peak = std::fmax(peak, std::fabs((double)x)); }
    std::printf("{\"blocks\": %d, \"position\": %lld, \"status\": %d, \"sum\": %.12e, \"sumsq\": %.12e, \"peak\": %.9e, "
                "\"first\": [%.9e, %.9e], \"last\": [%.9e, %.9e], \"ms\": %.4f, \"Msamples_per_s\": %.1f}\n",
                K, (long long)position, status, sum, sq, peak, bus[0], bus[1], bus[rows * C - 2], bus[rows * C - 1],
                ms, (double)V * rows / ms / 1e3);
    return 0;
}
