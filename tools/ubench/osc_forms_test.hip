#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "../../signals_amd/csrc/sig_osc.h"
__global__ void k(const double* t, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sig_osc::osc_square(t[i]); out[n + i] = sig_osc::osc_square_fract(t[i]);
    out[2 * n + i] = sig_osc::osc_triangle(t[i]); out[3 * n + i] = sig_osc::osc_triangle_fract(t[i]);
    out[4 * n + i] = sig_osc::osc_sawtooth(t[i]); out[5 * n + i] = sig_osc::osc_sawtooth_fract(t[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> t(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-3.0, 3.0), big(-1e7, 1e7);
    for (int i = 0; i < n; ++i) t[i] = (i & 1) ? u(g) : big(g);
    // exact quarter / half / eighth points and their neighbours
    int j = 0;
    for (int q = -64; q <= 64; ++q) for (int e = -2; e <= 2; ++e) { double x = q * 0.125; t[j++] = std::nextafter(x, e < 0 ? -1e9 : 1e9) * (e ? 1 : 1); t[j++] = x; t[j++] = x + e * 1e-17; }
    t[j++] = NAN; t[j++] = -0.0; t[j++] = 0.0; t[j++] = 1e15 + 0.5; t[j++] = -1e-20; t[j++] = 0.75; t[j++] = 0.5 - 5.5e-17;
    double *dt, *dout; hipMalloc(&dt, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dt, t.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dt, dout, n);
    std::vector<double> o(6 * (size_t)n); hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* names[3] = {"square", "triangle", "sawtooth"};
    int bad = 0;
    for (int w = 0; w < 3; ++w) {
        long diffbits = 0, diffval = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            double a = o[(2 * w) * (size_t)n + i], b = o[(2 * w + 1) * (size_t)n + i];
            if (std::isnan(a) && std::isnan(b)) continue;
            if (memcmp(&a, &b, 8)) { ++diffbits; if (a != b) { ++diffval; worst = fmax(worst, fabs(a - b)); if (diffval <= 5) printf("  %s t=%.17g ref %.17g fast %.17g\n", names[w], t[i], a, b); } }
        }
        printf("%s: %ld of %d differ in bits, %ld in value (max |diff| %.3g)\n", names[w], diffbits, n, diffval, worst);
        if (worst > 3e-16) bad = 1;
    }
    return bad;
}
