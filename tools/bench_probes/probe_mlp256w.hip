// tools/probe_mlp256w.hip -- mlp256_kernel (one wave per SIMD, 32 tokens per wave) against mlp256w_kernel (two waves per SIMD,
// 16 tokens per wave) on 4096 rows' worth of tokens, alternating in one process (box-to-box variation is larger than the gap).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../experiments/gpt_kernels_c256w.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;

template <int W, int X = 0>
float run(const char *tag, float *x, const float *gain, const uint16_t *ws, const float2 *lut, int M)
{
    const size_t lds = 8 * 8 * 2 * 1024 + kGeluLutN * 8;
    const int grid = M / 128;
    if (W) hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256w_kernel<F16T, 2, X>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256_kernel<F16T, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    auto launch = [&]() {
        if (W) mlp256w_kernel<F16T, 2, X><<<grid, 512, lds>>>(x, gain, ws, 1e-3f, 1e-3f, lut);
        else mlp256_kernel<F16T, 2, 0><<<grid, 256, lds>>>(x, gain, ws, 1e-3f, 1e-3f, lut);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int i = 0; i < 8; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 8;
    printf("%-52s %7.3f ms  %6.3f us per workgroup step  MFMA-issue %.0f TFLOP/s  [%s]\n", tag, ms, ms * 1e3 / ((M / 128) / 256.0) / kM256Steps,
           3.0 * 16 * 256 * 256 * (double)M / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    return ms;
}
int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int M = 4096 * 256;
    float *x; hipMalloc(&x, (size_t)M * 256 * 4);
    std::vector<float> hx((size_t)M * 256);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    float *gain; hipMalloc(&gain, 1024);
    std::vector<float> hg(256, 1.0f); hipMemcpy(gain, hg.data(), 1024, hipMemcpyHostToDevice);
    const size_t n16 = (size_t)kM256Steps * 8 * 2 * 512;
    uint16_t *ws; hipMalloc(&ws, n16 * 2);
    std::vector<uint16_t> hw(n16);
    for (size_t i = 0; i < n16; i++) { _Float16 v = (_Float16)(((float)((i * 40503u) & 0xfff) / 4096.f - 0.5f) * 0.1f); hw[i] = __builtin_bit_cast(uint16_t, v); }
    hipMemcpy(ws, hw.data(), n16 * 2, hipMemcpyHostToDevice);
    float2 *lut; hipMalloc(&lut, kGeluLutN * 8); hipMemset(lut, 0, kGeluLutN * 8);
    for (int rep = 0; rep < 3; rep++) {
        run<0>("mlp256_kernel  (4 waves x 32 tokens, 1 wave/SIMD)", x, gain, ws, lut, M);
        run<1>("mlp256w_kernel (8 waves x 16 tokens, 2 waves/SIMD)", x, gain, ws, lut, M);
    }
    {   // the same kernels on constant operands (every token row and every weight fragment identical): same instruction stream,
        // far fewer bits toggling -- separates "issue bound" from "what the chip sustains under load"
        std::vector<float> hc((size_t)M * 256);
        for (size_t i = 0; i < hc.size(); i++) hc[i] = (float)(i & 255) / 256.f - 0.5f;        // every row the same ramp
        hipMemcpy(x, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> hwc(n16, (uint16_t)0x2e66);                                         // 0.1 in fp16 everywhere
        hipMemcpy(ws, hwc.data(), n16 * 2, hipMemcpyHostToDevice);
        run<0>("mlp256_kernel,  constant rows and weights", x, gain, ws, lut, M);
        run<1>("mlp256w_kernel, constant rows and weights", x, gain, ws, lut, M);
        run<0>("mlp256_kernel,  constant rows and weights", x, gain, ws, lut, M);
        run<1>("mlp256w_kernel, constant rows and weights", x, gain, ws, lut, M);
        hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ws, hw.data(), n16 * 2, hipMemcpyHostToDevice);
    }
    run<1, 1>("mlp256w: no table gathers", x, gain, ws, lut, M);
    run<1, 2>("mlp256w: MFMA / VALU order left to hipcc", x, gain, ws, lut, M);
    run<1, 4>("mlp256w: no GELU arithmetic", x, gain, ws, lut, M);
    run<1, 8>("mlp256w: no row loads / stores", x, gain, ws, lut, M);
    run<1, 12>("mlp256w: no GELU, no row traffic", x, gain, ws, lut, M);
    run<1>("mlp256w_kernel again", x, gain, ws, lut, M);
    return 0;
}
