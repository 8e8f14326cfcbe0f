// tools/bench_probes/check_attn256o.hip -- attn256o_kernel (whole attention block, persistent) on realistic operands: time per launch next to
// round 3's attn256_kernel + the out-projection's lower bound, shader clock, and where wave 0's cycles go (STAMPS): prologue, q|k|v
// projection steps, attention phases, tail steps, tail epilogues.  Correctness is the library's parity tests' job (tests/test_gpu_gpt.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../../mapf_gpt_amd/csrc/gpt_kernels_c256b.h"
#ifdef MGPT_CHECK_Q                                      // -DMGPT_CHECK_Q: the same probe on attn256q_kernel (gpt_kernels_c256b.h)
#define attn256o_kernel attn256q_kernel
#define pack_attn256o_kernel pack_attn256q_kernel
#endif
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;
static float gauss(uint64_t &st)
{
    auto u = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) + 1) / 9007199254740993.0; };
    return (float)(sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u()));
}
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rows = argc > 1 ? atoi(argv[1]) : 12288;
    const size_t M = (size_t)rows * 256;
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int n_cu = prop.multiProcessorCount;
    float *x; hipMalloc(&x, M * 256 * 4);
    {
        std::vector<float> hx((size_t)4096 * 256 * 256);
        uint64_t st = 777;
        for (auto &v : hx) v = gauss(st);
        for (size_t o = 0; o < M * 256; o += hx.size()) hipMemcpy(x + o, hx.data(), std::min(hx.size(), M * 256 - o) * 4, hipMemcpyHostToDevice);
    }
    float *gain; hipMalloc(&gain, 1024);
    { std::vector<float> hg(256, 1.0f); hipMemcpy(gain, hg.data(), 1024, hipMemcpyHostToDevice); }
    uint64_t st = 4242;
    std::vector<float> wa((size_t)3 * 256 * 256), wp((size_t)256 * 256);
    float mxa = 0, mxp = 0;
    for (auto &v : wa) { v = 0.02f * gauss(st); mxa = std::max(mxa, fabsf(v)); }
    for (auto &v : wp) { v = 0.02f * gauss(st); mxp = std::max(mxp, fabsf(v)); }
    const float sa = ldexpf(1.f, (int)floorf(log2f(4096.f / mxa))), sp = ldexpf(1.f, (int)floorf(log2f(4096.f / mxp)));
    float *dwa, *dwp; hipMalloc(&dwa, wa.size() * 4); hipMalloc(&dwp, wp.size() * 4);
    hipMemcpy(dwa, wa.data(), wa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    uint16_t *ws; hipMalloc(&ws, (size_t)kA256oPeriod * 8 * 2 * 512 * 2);
    pack_attn256o_kernel<F16T, 2><<<(kA256oPeriod * 8 * 64 + 255) / 256, 256>>>(dwa, gain, dwp, ws, sa, sp);
    unsigned char *spill; hipMalloc(&spill, (size_t)n_cu * kA256oSpillPerWg);
    const size_t lds = 5 * 8 * 2 * 1024 + 2 * (256 * 80 + 32 * 528);
    auto k0 = &attn256o_kernel<F16T, 2, 0>;
    auto k1 = &attn256o_kernel<F16T, 2, 1>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // (x drifts by + y c_proj^T per launch: tiny weights keep it finite over the few hundred launches of this probe)
    const float isa = 1.f / sa, isp = 1.f / sp, sl2 = 0.17677669f * 1.44269504f;
    const int grid = std::min(rows, n_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 40; i++) k0<<<grid, 512, lds>>>(x, ws, isa, sl2, isp * 1e-3f, spill, rows, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("attn256o_kernel: %.3f ms per %d-row launch (40 launches, grid %d)  [%s]\n", ms / 40, rows, grid, hipGetErrorString(hipGetLastError()));
    }
    unsigned long long *stp; hipMalloc(&stp, (size_t)grid * 64);
    const double rows_per_wg = (double)rows / grid;
    for (int which = 0; which < 2; which++) {
    if (which == 0) k1<<<grid, 512, lds>>>(x, ws, isa, sl2, isp * 1e-3f, spill, rows, stp);
    else {
        auto k3 = &attn256o_kernel<F16T, 2, 3>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        k3<<<grid, 512, lds>>>(x, ws, isa, sl2, isp * 1e-3f, spill, rows, stp);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * 8);
    hipMemcpy(h.data(), stp, h.size() * 8, hipMemcpyDeviceToHost);
    double pro = 0, qkv = 0, att = 0, tl = 0, ep = 0, rt = 0, tot = 0;
    for (int b = 0; b < grid; b++) {
        const unsigned long long *s = &h[(size_t)b * 8];
        pro += (double)s[2]; qkv += (double)s[3]; att += (double)s[4]; tl += (double)s[5]; ep += (double)s[6]; rt += (double)(s[7] - s[1]);
    }
    tot = pro + qkv + att + tl + ep;
    printf("stamps (wave %d, mean per row over %d workgroups x %.1f rows): %.0f cycles per row: prologue %.0f | q|k|v projection steps %.0f (%.0f per head) | "
           "attention phases incl. k/v barrier %.0f (%.0f per head) | tail steps %.0f (%.0f per step) | tail epilogues %.0f | shader clock %.3f GHz\n",
           4 * which, grid, rows_per_wg, tot / grid / rows_per_wg, pro / grid / rows_per_wg, qkv / grid / rows_per_wg, qkv / grid / rows_per_wg / 8,
           att / grid / rows_per_wg, att / grid / rows_per_wg / 8, tl / grid / rows_per_wg, tl / grid / rows_per_wg / 16, ep / grid / rows_per_wg,
           tot / rt / 10.0);
    }
    {   // STAMPS == 2: where a q|k-shaped step's cycles go (projection steps 0-3 of every head + the 16 tail steps = 48 per row), waves 0 and 4
        auto k2 = &attn256o_kernel<F16T, 2, 2>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        unsigned long long *st2; hipMalloc(&st2, (size_t)grid * 128);
        k2<<<grid, 512, lds>>>(x, ws, isa, sl2, isp * 1e-3f, spill, rows, st2);
        hipDeviceSynchronize();
        std::vector<unsigned long long> g((size_t)grid * 16);
        hipMemcpy(g.data(), st2, g.size() * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; w++) {
            double a[8] = {0};
            for (int b = 0; b < grid; b++) for (int i = 2; i < 7; i++) a[i] += (double)g[((size_t)b * 2 + w) * 8 + i];
            const double steps = (double)grid * rows_per_wg * 48;
            printf("step phases, wave %d: q|k-shaped steps (48 per row): wait + barrier + piece issue %.0f, four chunks %.0f | v steps (16 per row): %.0f, %.0f  [%s]\n",
                   4 * w, a[2] / steps, a[4] / steps, a[3] / steps * 3, a[5] / steps * 3, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
