// tools/bench_probes/check_mlp160p.hip -- mlp160p_kernel (persistent producer / consumer MLP block, C = 160) against an fp64 host
// computation of x + c_proj(GELU_erf(c_fc(LayerNorm(x))))  (model.py:84-89, 103), with several blocks per workgroup (grid
// independence bit for bit); then its time per 16384-row launch on realistic operands (f16x3 and bf16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../experiments/gpt_kernels_c160p.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;
static float gauss(uint64_t &st)
{
    auto u = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) + 1) / 9007199254740993.0; };
    return (float)(sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u()));
}
static std::vector<float> retile(const std::vector<float> &a, int C, bool to_tiled)
{
    std::vector<float> o(a.size());
    const size_t M = a.size() / C;
    for (size_t m = 0; m < M; m++)
        for (int n = 0; n < C; n++) {
            const size_t t = (((m >> 5) * (C >> 3) + (n >> 3)) << 8) + ((m & 31) << 3) + (n & 7), p = m * C + n;
            if (to_tiled) o[t] = a[p]; else o[p] = a[t];
        }
    return o;
}
int main()
{
    const int C = 160;
    uint64_t seed = 7;
    std::vector<float> hg(C), hfc((size_t)4 * C * C), hpj((size_t)4 * C * C);
    float mx1 = 0, mx2 = 0;
    for (auto &v : hg) v = 1.f + 0.1f * gauss(seed);
    for (auto &v : hfc) v = 0.02f * gauss(seed);
    for (auto &v : hpj) v = 0.02f * gauss(seed);
    for (size_t i = 0; i < hfc.size(); i++) mx1 = fmaxf(mx1, fabsf(hfc[i] * hg[i % C]));
    for (auto v : hpj) mx2 = fmaxf(mx2, fabsf(v));
    const float sc1 = ldexpf(1.f, (int)floorf(log2f(4096.f / mx1))), sc2 = ldexpf(1.f, (int)floorf(log2f(4096.f / mx2)));
    float *g, *fc, *pj;
    hipMalloc(&g, C * 4); hipMalloc(&fc, hfc.size() * 4); hipMalloc(&pj, hpj.size() * 4);
    hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(fc, hfc.data(), hfc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(pj, hpj.data(), hpj.size() * 4, hipMemcpyHostToDevice);
    uint16_t *pk2, *pk1;
    hipMalloc(&pk2, (size_t)kM5Period * 20 * 2 * 512 * 2); hipMalloc(&pk1, (size_t)kM5Period * 20 * 1 * 512 * 2);
    pack_mlp160p_kernel<F16T, 2><<<(kM5Period * 20 * 64 + 255) / 256, 256>>>(fc, pj, g, pk2, sc1, sc2);
    pack_mlp160p_kernel<BF16T, 1><<<(kM5Period * 20 * 64 + 255) / 256, 256>>>(fc, pj, g, pk1, 1.f, 1.f);
    std::vector<float2> lut(kGeluLutN);
    for (int i = 0; i < kGeluLutN; i++) {
        const double v0 = (i - (double)kGeluLutBias) / kGeluLutScale, v1 = (i + 1 - (double)kGeluLutBias) / kGeluLutScale;
        const float f0 = (float)(0.5 * (1.0 + erf(v0 * 0.70710678118654752440)));
        lut[i] = make_float2(f0, (float)(0.5 * (1.0 + erf(v1 * 0.70710678118654752440)) - (double)f0));
    }
    float2 *dl; hipMalloc(&dl, lut.size() * 8); hipMemcpy(dl, lut.data(), lut.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp160p_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kM5Lds<2>);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp160p_kernel<BF16T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kM5Lds<1>);
    constexpr int L2 = kM5Lds<2>, L1 = kM5Lds<1>;
    std::vector<float> first;
    // ---- correctness: 7 blocks on 1, 2, 3 and 7 workgroups (1 .. 7 blocks per workgroup incl. uneven splits) ----
    for (int mode = 0; mode < 2; mode++)
    for (int grid : {1, 2, 3, 7}) {
        if (mode == 1 && grid != 3) continue;
        const int nb = 7, M = nb * 128;
        std::vector<float> hx((size_t)M * C), b(hx.size());
        uint64_t s2 = 99;
        for (auto &v : hx) v = gauss(s2) + 0.3f;
        float *x; hipMalloc(&x, hx.size() * 4);
        hipMemcpy(x, retile(hx, C, true).data(), hx.size() * 4, hipMemcpyHostToDevice);
        if (mode == 0) mlp160p_kernel<F16T, 2><<<grid, 512, L2>>>(x, pk2, 1.f / sc1, 1.f / sc2, dl, nb);
        else mlp160p_kernel<BF16T, 1><<<grid, 512, L1>>>(x, pk1, 1.f, 1.f, dl, nb);
        hipError_t e = hipDeviceSynchronize();
        printf("%s grid %d: launch status: %s / %s\n", mode ? "bf16" : "f16x3", grid, hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
        hipMemcpy(b.data(), x, b.size() * 4, hipMemcpyDeviceToHost);
        b = retile(b, C, false);
        double mxd = 0, mx = 0; int worst = -1; long nan_count = 0;
        std::vector<double> xn(C), hid(4 * C);
        for (int m = 0; m < M; m++) {
            double mean = 0, var = 0;
            for (int c = 0; c < C; c++) mean += hx[(size_t)m * C + c];
            mean /= C;
            for (int c = 0; c < C; c++) { const double d = hx[(size_t)m * C + c] - mean; var += d * d; }
            const double rstd = 1.0 / sqrt(var / C + 1e-5);
            for (int c = 0; c < C; c++) xn[c] = (hx[(size_t)m * C + c] - mean) * rstd * hg[c];
            for (int u = 0; u < 4 * C; u++) { double a = 0; for (int c = 0; c < C; c++) a += xn[c] * hfc[(size_t)u * C + c]; hid[u] = 0.5 * a * (1.0 + erf(a * 0.70710678118654752440)); }
            for (int o = 0; o < C; o++) {
                double a = 0; for (int u = 0; u < 4 * C; u++) a += hid[u] * hpj[(size_t)o * 4 * C + u];
                const double got = b[(size_t)m * C + o];
                if (got != got) { nan_count++; continue; }
                const double d = fabs(hx[(size_t)m * C + o] + a - got);
                if (d > mxd) { mxd = d; worst = m; }
                mx = fmax(mx, fabs(a));
            }
        }
        printf("%s grid %d: max |mlp output| %.4f   max |kernel - fp64| %.3e (token %d)   NaNs %ld\n", mode ? "bf16" : "f16x3", grid, mx, mxd, worst, nan_count);
        if (mode == 0) {
            if (first.empty()) first = b;
            else { long diff = 0; for (size_t i = 0; i < b.size(); i++) diff += memcmp(&b[i], &first[i], 4) != 0; printf("grid %d vs grid 1: %ld elements differ\n", grid, diff); }
        }
        hipFree(x);
    }
    // ---- time: 16384 rows (32768 blocks) on N(0,1) rows ----
    {
        const int rows = 16384; const size_t M = (size_t)rows * 256;
        std::vector<float> hx(M * C);
        uint64_t s3 = 5; for (size_t i = 0; i < hx.size(); i++) hx[i] = gauss(s3);
        float *x; hipMalloc(&x, hx.size() * 4);
        hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int nb = (int)(M / 128);
        for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) {
                if (mode == 0) mlp160p_kernel<F16T, 2><<<256, 512, L2>>>(x, pk2, 1.f / sc1, 1.f / sc2, dl, nb);
                else mlp160p_kernel<BF16T, 1><<<256, 512, L1>>>(x, pk1, 1.f, 1.f, dl, nb);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 16.0 * C * C * (double)M;
            printf("mlp160p_kernel %s (persistent, 256 workgroups): %.3f ms per 16384-row launch = %.0f TFLOP/s algorithmic  [%s]\n", mode ? "bf16 " : "f16x3", ms / 10,
                   flops / (ms / 10 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
