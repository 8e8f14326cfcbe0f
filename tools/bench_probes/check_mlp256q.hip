// tools/bench_probes/check_mlp256q.hip -- mlp256q_kernel (mlp256p_kernel on v_mfma_f32_16x16x32, gpt_kernels_c256q.h) against an fp64 host computation of
// x + c_proj(GELU_erf(c_fc(LayerNorm(x))))  (model.py:84-89, 103), with several blocks per workgroup; then its time per
// 4096-row launch on realistic operands next to mlp256_kernel (same process, same rows).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../../mapf_gpt_amd/csrc/gpt_kernels_c256q.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;
static float gauss(uint64_t &st)
{
    auto u = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) + 1) / 9007199254740993.0; };
    return (float)(sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u()));
}
// experiment: zero the low `drop` mantissa bits of every lo-plane value of a packed weight stream (planes alternate per KiB)
__global__ void mask_lo_planes(uint16_t *ws, size_t n16, int drop)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    if ((i >> 9) & 1) ws[i] &= (uint16_t)(0xffffu << drop);
}
// the kernel's residual stream is chunk-major (xt_off): [32-token tile][C / 8][32 tokens][8 floats]
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
int main(int argc, char **argv)
{
    const int C = 256;
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
    uint16_t *pkp, *pko, *pkq;
    hipMalloc(&pkp, (size_t)kMPPeriod * 16 * 2 * 512 * 2); hipMalloc(&pkq, (size_t)kMPPeriod * 16 * 2 * 512 * 2); hipMalloc(&pko, (size_t)kM256Steps * 8 * 2 * 512 * 2);
    pack_mlp256p_kernel<F16T, 2><<<(kMPPeriod * 16 * 64 + 255) / 256, 256>>>(fc, pj, g, pkp, sc1, sc2);
    pack_mlp256q_kernel<F16T, 2><<<(kMPPeriod * 16 * 64 + 255) / 256, 256>>>(fc, pj, g, pkq, sc1, sc2);
    float mxo = 0; for (auto v : hfc) mxo = fmaxf(mxo, fabsf(v));
    const float sco = ldexpf(1.f, (int)floorf(log2f(4096.f / mxo)));
    pack_mlp256_kernel<F16T, 2><<<(kM256Steps * 8 * 64 + 255) / 256, 256>>>(fc, pj, pko, sco, sc2);
    std::vector<float2> lut(kGeluLutN);
    for (int i = 0; i < kGeluLutN; i++) {
        const double v0 = (i - (double)kGeluLutBias) / kGeluLutScale, v1 = (i + 1 - (double)kGeluLutBias) / kGeluLutScale;
        const float f0 = (float)(0.5 * (1.0 + erf(v0 * 0.70710678118654752440)));
        lut[i] = make_float2(f0, (float)(0.5 * (1.0 + erf(v1 * 0.70710678118654752440)) - (double)f0));
    }
    std::vector<float2> lutp = lut;                        // mlp256p_kernel: the table times 1 / scale of the c_fc stream
    for (auto &e : lutp) { e.x *= 1.f / sc1; e.y *= 1.f / sc1; }
    float2 *dlp; hipMalloc(&dlp, lutp.size() * 8); hipMemcpy(dlp, lutp.data(), lutp.size() * 8, hipMemcpyHostToDevice);
    constexpr int LDSP = kMPLds<2>;
    float2 *dl; hipMalloc(&dl, lut.size() * 8); hipMemcpy(dl, lut.data(), lut.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256p_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256q_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256q_kernel<F16T, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256p_kernel<F16T, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
    const int ldso = 8 * 8 * 2 * 1024 + kGeluLutN * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp256_kernel<F16T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ldso);
    int rc = 0;
    std::vector<float> first;                              // output of the first grid: every other grid must reproduce it bit for bit
    // ---- correctness: 7 blocks on 1, 2, 3 and 7 workgroups (1 .. 7 blocks per workgroup incl. uneven splits) ----
    for (int grid : {1, 2, 3, 7}) {
        const int nb = 7, M = nb * 128;
        std::vector<float> hx((size_t)M * C), b(hx.size());
        uint64_t s2 = 99;
        for (auto &v : hx) v = gauss(s2) + 0.3f;
        float *x; hipMalloc(&x, hx.size() * 4);
        hipMemcpy(x, retile(hx, C, true).data(), hx.size() * 4, hipMemcpyHostToDevice);
        mlp256q_kernel<F16T, 2><<<grid, 512, LDSP>>>(x, pkq, 1.f / sc1, 1.f / sc2, dlp, nb);
        hipError_t e = hipDeviceSynchronize();
        printf("grid %d: launch status: %s / %s\n", grid, hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
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
            for (int u = 0; u < 4 * C; u++) {
                double a = 0;
                for (int c = 0; c < C; c++) a += xn[c] * hfc[(size_t)u * C + c];
                hid[u] = 0.5 * a * (1.0 + erf(a * 0.70710678118654752440));
            }
            for (int o = 0; o < C; o++) {
                double a = 0;
                for (int u = 0; u < 4 * C; u++) a += hid[u] * hpj[(size_t)o * 4 * C + u];
                mx = fmax(mx, fabs(a));
                const double d = fabs(hx[(size_t)m * C + o] + a - b[(size_t)m * C + o]);
                if (d != d) { nan_count++; worst = m; }
                else if (d > mxd) { mxd = d; worst = m; }
            }
        }
        printf("grid %d: max |mlp output| %.4f   max |kernel - fp64| %.3e (token %d)\n", grid, mx, mxd, worst);
        if (!(mxd < 5e-6) && grid == 1) {                  // where are the wrong values?
            std::vector<double> ref((size_t)M * C);
            for (int m = 0; m < M; m++) {
                double mean = 0, var = 0;
                for (int c = 0; c < C; c++) mean += hx[(size_t)m * C + c];
                mean /= C;
                for (int c = 0; c < C; c++) { const double d = hx[(size_t)m * C + c] - mean; var += d * d; }
                const double rstd = 1.0 / sqrt(var / C + 1e-5);
                for (int c = 0; c < C; c++) xn[c] = (hx[(size_t)m * C + c] - mean) * rstd * hg[c];
                for (int u = 0; u < 4 * C; u++) { double a = 0; for (int c = 0; c < C; c++) a += xn[c] * hfc[(size_t)u * C + c]; hid[u] = 0.5 * a * (1.0 + erf(a * 0.70710678118654752440)); }
                for (int o = 0; o < C; o++) { double a = 0; for (int u = 0; u < 4 * C; u++) a += hid[u] * hpj[(size_t)o * 4 * C + u]; ref[(size_t)m * C + o] = hx[(size_t)m * C + o] + a; }
            }
            for (int blk = 0; blk < nb; blk++) {
                printf("block %d:", blk);
                for (int w = 0; w < 4; w++) {
                    int bad = 0; double me = 0, mi = 0;
                    for (int t = 0; t < 32; t++) for (int c = 0; c < C; c++) {
                        const size_t i = ((size_t)blk * 128 + w * 32 + t) * C + c;
                        const double e = fabs(ref[i] - b[i]); if (e > 1e-4 || e != e) bad++; me = fmax(me, e); mi = fmax(mi, fabs((double)hx[i] - b[i]));
                    }
                    printf("  pair %d: %d bad, max err %.2e, max |out - in| %.2e;", w, bad, me, mi);
                }
                printf("\n");
            }
            const int m = worst;
            printf("token %d, error per output tile j (max over 32 features) and per feature within tile 0:\n", m);
            for (int j = 0; j < 8; j++) { double e = 0; for (int c = 0; c < 32; c++) e = fmax(e, fabs(ref[(size_t)m * C + 32 * j + c] - b[(size_t)m * C + 32 * j + c])); printf(" %.2e", e); }
            printf("\n");
            for (int c = 0; c < 32; c++) printf(" %.1e", fabs(ref[(size_t)m * C + c] - b[(size_t)m * C + c]));
            printf("\n");
        }
        if (nan_count) printf("grid %d: %ld NaN outputs\n", grid, nan_count);
        if (!(mxd < 5e-6) || nan_count) rc = 1;
        if (first.empty()) first = b;
        else {
            size_t nd = 0, fi = 0;
            for (size_t i = 0; i < b.size(); i++) if (memcmp(&b[i], &first[i], 4) != 0) { if (!nd) fi = i; nd++; }
            printf("grid %d vs grid 1: %zu elements differ", grid, nd);
            if (nd) { printf(" (first: token %zu feature %zu: %.9g vs %.9g)", fi / C, fi % C, b[fi], first[fi]); rc = 1; }
            printf("\n");
        }
        hipFree(x);
    }
    if (argc > 1 && atoi(argv[1]) == 0) return rc;
    // ---- time: 4096 rows, realistic operands, both kernels alternately ----
    const int M = 4096 * 256, nb = M / 128;
    int dev = 0; hipDeviceProp_t prop; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
    const int ncu = prop.multiProcessorCount;
    std::vector<float> hx((size_t)M * C);
    for (auto &v : hx) v = gauss(seed);
    float *x; hipMalloc(&x, hx.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long *st; hipMalloc(&st, (size_t)ncu * 2 * 4 * 8);
    for (int rep = 0; rep < 4; rep++) {
        float ms;
        hipEventRecord(e0);
        for (int i = 0; i < 60; i++) mlp256p_kernel<F16T, 2><<<ncu, 512, LDSP>>>(x, pkp, 1.f / sc1, 1.f / sc2, dlp, nb);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mlp256p_kernel (32x32x16, %d workgroups): %.3f ms per 4096-row launch  [%s]\n", ncu, ms / 60, hipGetErrorString(hipGetLastError()));
        hipEventRecord(e0);
        for (int i = 0; i < 60; i++) mlp256q_kernel<F16T, 2><<<ncu, 512, LDSP>>>(x, pkq, 1.f / sc1, 1.f / sc2, dlp, nb);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mlp256q_kernel (16x16x32, %d workgroups): %.3f ms per 4096-row launch  [%s]\n", ncu, ms / 60, hipGetErrorString(hipGetLastError()));
    }
    for (int which = 0; which < 2; which++) {
        if (which == 0) mlp256p_kernel<F16T, 2, 1><<<ncu, 512, LDSP>>>(x, pkp, 1.f / sc1, 1.f / sc2, dlp, nb, st);
        else mlp256q_kernel<F16T, 2, 1><<<ncu, 512, LDSP>>>(x, pkq, 1.f / sc1, 1.f / sc2, dlp, nb, st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)ncu * 8);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int w = 0; w < ncu; w++) { cyc += (double)(h[8 * w + 2] - h[8 * w]); rt += (double)(h[8 * w + 3] - h[8 * w + 1]); }
        printf("%s stamps: %.0f shader cycles per block = %.1f per 32-KiB stream step; shader clock %.3f GHz\n", which ? "mlp256q" : "mlp256p", cyc / ncu / (nb / (double)ncu),
               cyc / ncu / (nb / (double)ncu) / kMPPeriod, cyc / rt / 10.0);
    }
    return rc;
}
