// tools/probe_attn256.hip -- where does attn256_kernel spend its time?  Product kernel and ablated variants on 4096 rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../mapf_gpt_amd/csrc/gpt_kernels_c256.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;
template <int ABL>
void run(const char *tag, const float *x, const float *gain, const uint16_t *ws, uint16_t *y, int rows)
{
    const size_t lds = 5 * 8 * 2 * 1024 + 2 * (256 * 80 + 32 * 528);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&attn256_kernel<F16T, 2, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fprintf(stderr, "running %s\n", tag);
    attn256_kernel<F16T, 2, false, ABL><<<rows, 512, lds>>>(x, gain, ws, 1e-3f, 0.255f, y);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) attn256_kernel<F16T, 2, false, ABL><<<rows, 512, lds>>>(x, gain, ws, 1e-3f, 0.255f, y);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = 3.0 * (6.0 * 256 * 256 * 256 + 4.0 * 256 * 256 * 256) * rows;
    printf("%-44s %7.3f ms  %6.2f us per row-head  MFMA-issue %.0f TFLOP/s  [%s]\n", tag, ms, ms * 1e3 / (rows / 256.0) / 8, flops / (ms * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
}
int main()
{
    const int rows = 4096;
    const size_t M = (size_t)rows * 256;
    float *x; hipMalloc(&x, M * 256 * 4);
    std::vector<float> hx(M * 256);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    float *gain; hipMalloc(&gain, 1024);
    std::vector<float> hg(256, 1.0f); hipMemcpy(gain, hg.data(), 1024, hipMemcpyHostToDevice);
    const size_t n16 = (size_t)8 * kA256StepsPerHead * 8 * 2 * 512;
    uint16_t *ws; hipMalloc(&ws, n16 * 2);
    std::vector<uint16_t> hw(n16);
    for (size_t i = 0; i < n16; i++) { _Float16 v = (_Float16)(((float)((i * 40503u) & 0xfff) / 4096.f - 0.5f) * 40.f); hw[i] = __builtin_bit_cast(uint16_t, v); }
    hipMemcpy(ws, hw.data(), n16 * 2, hipMemcpyHostToDevice);
    uint16_t *y; hipMalloc(&y, M * 256 * 2 * 2);
    run<0>("product", x, gain, ws, y, rows);
    run<1>("no weight DMA in the loop", x, gain, ws, y, rows);
    run<2>("no softmax arithmetic", x, gain, ws, y, rows);
    run<8>("no projection MFMAs", x, gain, ws, y, rows);
    run<16>("no ring barriers", x, gain, ws, y, rows);
    run<17>("no DMA, no ring barriers", x, gain, ws, y, rows);
    run<10>("no softmax, no projection MFMAs", x, gain, ws, y, rows);
    run<27>("no DMA/barriers/softmax/projection MFMAs", x, gain, ws, y, rows);
    run<0>("product again", x, gain, ws, y, rows);
    return 0;
}
