// tools/check_mlp256w.hip -- mlp256w_kernel (two waves per SIMD, 16 tokens per wave) against an fp64 host computation of x + c_proj(GELU_erf(c_fc(LayerNorm(x))))
// (model.py:84-89, 103) on random rows and weights.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../experiments/gpt_kernels_c256w.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;
static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; }
int main()
{
    const int M = 256, C = 256;
    unsigned seed = 1;
    std::vector<float> hx((size_t)M * C), hg(C), hfc((size_t)4 * C * C), hpj((size_t)4 * C * C);
    for (auto &v : hx) v = frand(seed);
    for (auto &v : hg) v = 1.f + 0.1f * frand(seed);
    for (auto &v : hfc) v = 0.05f * frand(seed);
    for (auto &v : hpj) v = 0.05f * frand(seed);
    float *x2, *g, *fc, *pj;
    hipMalloc(&x2, hx.size() * 4); hipMalloc(&g, C * 4); hipMalloc(&fc, hfc.size() * 4); hipMalloc(&pj, hpj.size() * 4);
    hipMemcpy(x2, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(fc, hfc.data(), hfc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(pj, hpj.data(), hpj.size() * 4, hipMemcpyHostToDevice);
    const float sc = 32768.f;
    uint16_t *pk2; hipMalloc(&pk2, (size_t)kM256Steps * 8 * 2 * 512 * 2);
    pack_mlp256w_kernel<F16T, 2><<<(kM256Steps * 8 * 64 + 255) / 256, 256>>>(fc, pj, pk2, sc, sc);
    std::vector<float2> lut(kGeluLutN);
    for (int i = 0; i < kGeluLutN; i++) {
        const double v0 = (i - (double)kGeluLutBias) / kGeluLutScale, v1 = (i + 1 - (double)kGeluLutBias) / kGeluLutScale;
        const float f0 = (float)(0.5 * (1.0 + erf(v0 * 0.70710678118654752440)));
        lut[i] = make_float2(f0, (float)(0.5 * (1.0 + erf(v1 * 0.70710678118654752440)) - (double)f0));
    }
    float2 *dl; hipMalloc(&dl, lut.size() * 8); hipMemcpy(dl, lut.data(), lut.size() * 8, hipMemcpyHostToDevice);
    const int lds2 = 8 * 8 * 2 * 1024 + kGeluLutN * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256w_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    mlp256w_kernel<F16T, 2><<<M / 128, 512, lds2>>>(x2, g, pk2, 1.f / sc, 1.f / sc, dl);
    hipDeviceSynchronize();
    printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
    std::vector<float> b(hx.size());
    hipMemcpy(b.data(), x2, b.size() * 4, hipMemcpyDeviceToHost);
    // fp64 reference
    double mxd = 0, mx = 0;
    std::vector<double> xn(C), hid(4 * C);
    for (int m = 0; m < M; m++) {
        double mean = 0, var = 0;
        for (int c = 0; c < C; c++) mean += hx[(size_t)m * C + c];
        mean /= C;
        for (int c = 0; c < C; c++) { const double d = hx[(size_t)m * C + c] - mean; var += d * d; }
        const double rstd = 1.0 / sqrt(var / C + 1e-5);
        for (int c = 0; c < C; c++) xn[c] = (hx[(size_t)m * C + c] - mean) * rstd * hg[c];
        for (int u = 0; u < 4 * C; u++) {
            double a = 0;
            for (int c = 0; c < C; c++) a += xn[c] * hfc[(size_t)u * C + c];
            hid[u] = 0.5 * a * (1.0 + erf(a * 0.70710678118654752440));
        }
        for (int o = 0; o < C; o++) {
            double a = 0;
            for (int u = 0; u < 4 * C; u++) a += hid[u] * hpj[(size_t)o * 4 * C + u];
            mx = fmax(mx, fabs(a));
            mxd = fmax(mxd, fabs(hx[(size_t)m * C + o] + a - b[(size_t)m * C + o]));
        }
    }
    printf("max |mlp output| %.4f   max |kernel - fp64| %.3e\n", mx, mxd);
    return mxd < 5e-6 ? 0 : 1;
}
