// tools/bench_probes/probe_gemm_pk.hip -- gemm_pk_kernel (the 85M path's packed-fragment GEMM) by K: time = fixed cost per tile + cost per k-step.
// M = 262144 tokens (1024 rows), N = 3072 (c_fc of the 85M shape) or 768 (c_proj), K = 768 / 1536 / 3072; bf16 and f16x3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../mapf_gpt_amd/csrc/gpt_kernels_fast.h"
namespace mgpt { void set_error(const char *, ...) {} }
using namespace mgpt::fastk;

template <class T, int NP, int EPI, int NWV = 8>
void run(const char *tag, uint16_t *a, uint16_t *w, uint16_t *o, float *x, const float2 *lut, int M, int N, int K, int stagger = 0)
{
    GemmArgs p{};
    p.a_hi = a; p.w_hi = w; p.out_scale = 1.0f; p.M = M; p.N = N; p.K = K; p.n_tiles_n = N / 256; p.x_out = x; p.o_hi = o; p.o_pk = 1; p.gelu_lut = lut;
    p.C = 768; p.n_head = 12; p.hs = 64; p.x_tiled = stagger < 0;
    const size_t lds = (size_t)gemm_pk_lds(NP, NWV, EPI) + (EPI == EPI_GELU ? kGeluLutN * 8 : 0);
    auto kern = &gemm_pk_kernel<T, NP, EPI, NWV>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned n_tiles = (unsigned)((M / (NWV * 32)) * (N / 256));
    int dev = 0, ncu = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const dim3 grid(gemm_pk_persistent(NP, EPI, false) ? std::min(n_tiles, (unsigned)(ncu * (NWV == 8 ? 1 : 2))) : n_tiles);      // persistent workgroups
    for (int i = 0; i < 2; i++) kern<<<grid, NWV * 64, lds>>>(p, nullptr);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) kern<<<grid, NWV * 64, lds>>>(p, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms / 5);
    }
    {   // stamps
        auto dk = &gemm_pk_kernel<T, NP, EPI, NWV, 1>;
        hipFuncSetAttribute(reinterpret_cast<const void *>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        unsigned long long *st; hipMalloc(&st, (size_t)grid.x * 64);
        for (int i = 0; i < 3; i++) dk<<<grid, NWV * 64, lds>>>(p, st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)grid.x * 8);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        double fill = 0, loop = 0, epi = 0, tot = 0, rt = 0, tiles = 0;
        for (unsigned b = 0; b < grid.x; b++) {
            const unsigned long long *q = &h[(size_t)b * 8];
            fill += (double)q[0]; loop += (double)q[1]; epi += (double)q[2]; tiles += (double)q[3]; tot += (double)q[4]; rt += (double)q[5];
        }
        printf("    stamps (wave 0, per tile, mean over %.0f tiles of %u workgroups): wait for stage 0 %.0f | main loop %.0f (%.0f per k-step) | epilogue %.0f | "
               "total %.0f cycles, shader clock %.3f GHz\n",
               tiles, grid.x, fill / tiles, loop / tiles, loop / tiles / (K / 16), epi / tiles, tot / tiles, tot / rt / 10.0);
        hipFree(st);
    }
    const double flops = 2.0 * M * (double)N * K;
    const double tiles_per_cu = (double)n_tiles / 256.0;
    printf("%-18s M=%d N=%4d K=%4d  %8.3f ms  %7.1f TFLOP/s (%4.2f of 2500; x%d MFMA passes)  %7.2f us per tile  [%s]\n", tag, M, N, K, best,
           flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 2.5e15, NP == 2 ? 3 : 1, best * 1e3 / tiles_per_cu, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int M = 262144;
    const size_t na = (size_t)M * 3072 * 2;                    // elements: K up to 3072, two planes
    uint16_t *a, *w, *o; float *x; float2 *lut;
    hipMalloc(&a, na * 2); hipMalloc(&w, (size_t)3072 * 3072 * 2 * 2); hipMalloc(&o, (size_t)M * 3072 * 2 * 2); hipMalloc(&x, (size_t)M * 3072 * 4);
    hipMemset(x, 0, (size_t)M * 3072 * 4);
    {   // operands ~ N(0,1) activations, N(0,0.02)-scaled weights as fp16 / bf16 bit patterns (value distribution matters for the clock)
        std::vector<uint16_t> h((size_t)64 << 20);
        uint64_t st = 99;
        auto u = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) + 1) / 9007199254740993.0; };
        for (size_t i = 0; i < h.size(); i += 2) {
            const double rr = sqrt(-2.0 * log(u())), th = 6.283185307179586 * u();
            const float v0 = (float)(rr * cos(th)), v1 = (float)(rr * sin(th));
            h[i] = (uint16_t)(__builtin_bit_cast(unsigned, v0) >> 16); h[i + 1] = (uint16_t)(__builtin_bit_cast(unsigned, v1) >> 16);   // bf16 pattern
        }
        for (size_t off = 0; off < na; off += h.size()) hipMemcpy(a + off, h.data(), std::min(h.size(), na - off) * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, h.data(), (size_t)3072 * 3072 * 2 * 2, hipMemcpyHostToDevice);
        std::vector<float2> hl(kGeluLutN);
        for (int i = 0; i < kGeluLutN; i++) { const double xx = (i - 1536.0) / 256.0; const double ph = 0.5 * (1 + erf(xx / sqrt(2.0))); hl[i] = make_float2((float)ph, 0.001f); }
        hipMalloc(&lut, hl.size() * 8); hipMemcpy(lut, hl.data(), hl.size() * 8, hipMemcpyHostToDevice);
    }
    for (int K : {768, 1536, 3072}) run<BF16T, 1, EPI_GELU>("bf16 gelu->pk", a, w, o, x, lut, M, 3072, K);
    for (int K : {768, 1536, 3072}) run<BF16T, 1, EPI_RESID>("bf16 resid", a, w, o, x, lut, M, 768, K);
    printf("two 128 x 256 blocks per CU:\n");
    run<BF16T, 1, EPI_GELU, 4>("bf16 gelu->pk 2/CU", a, w, o, x, lut, M, 3072, 768);
    run<BF16T, 1, EPI_RESID, 4>("bf16 resid 2/CU", a, w, o, x, lut, M, 768, 768);
    run<BF16T, 1, EPI_RESID, 4>("bf16 resid 2/CU", a, w, o, x, lut, M, 768, 3072);
    printf("chunk-major residual stream (x as [M/32][N/8][32][8]):\n");
    for (int K : {768, 3072}) run<BF16T, 1, EPI_RESID>("bf16 resid tiled", a, w, o, x, lut, M, 768, K, -1);
    run<F16T, 2, EPI_RESID, 8>("f16x3 resid tiled", a, w, o, x, lut, M, 256, 256, -1);
    run<F16T, 2, EPI_RESID, 8>("f16x3 resid", a, w, o, x, lut, M, 256, 256);
    return 0;
}
