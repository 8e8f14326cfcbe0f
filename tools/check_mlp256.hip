// tools/check_mlp256.hip -- mlp256_kernel against round 1's mlp_pair_kernel (validated against the reference goldens) on the
// same random rows and weights; prints the largest difference per 32-column tile and per 32-token tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../mapf_gpt_amd/csrc/gpt_kernels_c256.h"
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
    float *x1, *x2, *g, *fc, *pj;
    hipMalloc(&x1, hx.size() * 4); hipMalloc(&x2, hx.size() * 4); hipMalloc(&g, C * 4); hipMalloc(&fc, hfc.size() * 4); hipMalloc(&pj, hpj.size() * 4);
    hipMemcpy(x1, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x2, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(fc, hfc.data(), hfc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(pj, hpj.data(), hpj.size() * 4, hipMemcpyHostToDevice);
    const float sc = 32768.f;
    // old kernel
    const size_t frags = C / 16 + 2 * (C / 32), nt = 4 * C / 32;
    uint16_t *pk1; hipMalloc(&pk1, nt * frags * 2 * 512 * 2);
    pack_mlp_kernel<F16T, 2><<<(unsigned)((nt * frags * 64 + 255) / 256), 256>>>(fc, pj, pk1, C, sc, sc);
    const int lds1 = (int)(frags * 2 * 1024 * 2 + 8 * 2048 + 8 * 32 * 4 + 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_pair_kernel<F16T, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    mlp_pair_kernel<F16T, 2, 8><<<M / 128, 512, lds1>>>(x1, g, pk1, 1.f / sc, 1.f / sc);
    // new kernel
    uint16_t *pk2; hipMalloc(&pk2, (size_t)kM256Steps * 8 * 2 * 512 * 2);
    pack_mlp256_kernel<F16T, 2><<<(kM256Steps * 8 * 64 + 255) / 256, 256>>>(fc, pj, pk2, sc, sc);
    std::vector<float2> lut(kGeluLutN);
    for (int i = 0; i < kGeluLutN; i++) {
        const double v0 = (i - (double)kGeluLutBias) / kGeluLutScale, v1 = (i + 1 - (double)kGeluLutBias) / kGeluLutScale;
        const float f0 = (float)(0.5 * (1.0 + erf(v0 * 0.70710678118654752440)));
        lut[i] = make_float2(f0, (float)(0.5 * (1.0 + erf(v1 * 0.70710678118654752440)) - (double)f0));
    }
    float2 *dl; hipMalloc(&dl, lut.size() * 8); hipMemcpy(dl, lut.data(), lut.size() * 8, hipMemcpyHostToDevice);
    const int lds2 = 8 * 8 * 2 * 1024 + kGeluLutN * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    mlp256_kernel<F16T, 2><<<M / 128, 256, lds2>>>(x2, g, pk2, 1.f / sc, 1.f / sc, dl);
    hipDeviceSynchronize();
    printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
    std::vector<float> a(hx.size()), b(hx.size());
    hipMemcpy(a.data(), x1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), x2, b.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, mxd = 0;
    for (size_t i = 0; i < a.size(); i++) { mx = fmax(mx, fabs(a[i] - hx[i])); mxd = fmax(mxd, fabs(a[i] - b[i])); }
    printf("max |mlp output| (old) %.4f   max |new - old| %.3e\n", mx, mxd);
    for (int jt = 0; jt < 8; jt++) { double d = 0; for (int m = 0; m < M; m++) for (int c = 0; c < 32; c++) d = fmax(d, fabs(a[(size_t)m * C + jt * 32 + c] - b[(size_t)m * C + jt * 32 + c])); printf("col tile %d: %.3e\n", jt, d); }
    for (int tt = 0; tt < M / 32; tt++) { double d = 0; for (int m = 0; m < 32; m++) for (int c = 0; c < C; c++) d = fmax(d, fabs(a[(size_t)(tt * 32 + m) * C + c] - b[(size_t)(tt * 32 + m) * C + c])); printf("token tile %d: %.3e\n", tt, d); }
    return 0;
}
