// tools/probe_mlp256.hip -- where does a step of mlp256_kernel spend its time?  Times the product kernel (ABL = 0) and
// ablated variants (no weight DMA / no GELU / no barrier / no MFMAs / no fragment reads) on 4096 rows' worth of tokens.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../mapf_gpt_amd/csrc/gpt_kernels_c256.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;

template <int ABL>
void run(const char *tag, float *x, const float *gain, const uint16_t *ws, int M, int grid = 0)
{
    const size_t lds = 8 * 8 * 2 * 1024 + kGeluLutN * 8;
    static float2 *lut = nullptr;
    if (!lut) { hipMalloc(&lut, kGeluLutN * 8); hipMemset(lut, 0, kGeluLutN * 8); }
    if (grid == 0) grid = M / 128;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256_kernel<F16T, 2, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mlp256_kernel<F16T, 2, ABL><<<grid, 256, lds>>>(x, gain, ws, 1e-3f, 1e-3f, lut);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) mlp256_kernel<F16T, 2, ABL><<<grid, 256, lds>>>(x, gain, ws, 1e-3f, 1e-3f, lut);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double blocks_per_cu = (M / 128) / 256.0, us_per_step = ms * 1e3 / blocks_per_cu / kM256Steps;
    printf("%-40s %7.3f ms  %6.3f us per step (24 MFMAs/wave)  MFMA-issue %.0f TFLOP/s\n", tag, ms, us_per_step,
           3.0 * 16 * 256 * 256 * (double)M / (ms * 1e-3) / 1e12);
    (void)hipGetLastError();
}
int main()
{
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
    run<0>("product", x, gain, ws, M);
    run<1>("no weight DMA in the loop", x, gain, ws, M);
    run<2>("no GELU", x, gain, ws, M);
    run<4>("no barrier", x, gain, ws, M);
    run<5>("no DMA, no barrier", x, gain, ws, M);
    run<7>("no DMA, no barrier, no GELU", x, gain, ws, M);
    run<16>("no fragment reads", x, gain, ws, M);
    run<23>("no DMA/barrier/GELU/reads (MFMA only)", x, gain, ws, M);
    run<8>("no MFMAs", x, gain, ws, M);
    run<0>("product again", x, gain, ws, M);
    return 0;
}
