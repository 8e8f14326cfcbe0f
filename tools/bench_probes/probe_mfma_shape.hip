// tools/bench_probes/probe_mfma_shape.hip -- at the package power limit, which MFMA shape delivers more per joule?  v_mfma_f32_32x32x16 (32 768 flops, a
// 1 024-value accumulator read and written) against v_mfma_f32_16x16x32 (16 384 flops, a 256-value accumulator): the same operand bytes per flop, half the
// accumulator traffic per flop.  Register-only loops on pseudo-random operands (four operand sets cycled, so that consecutive MFMAs see different bits),
// two waves per SIMD, every CU, ~150 ms per run so that power management settles; the shader clock comes from wave 0's cycle counter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x16, 1: 16x16x32;  BF 0: f16, 1: bf16;  ZERO: operands all zero (what the data costs)
template <int SHAPE, int BF, int ZERO>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, float scale)
{
    u32x4 A[4], B[4];
    unsigned st = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd16 = [&]() {                                   // a 16-bit pattern of a value ~ scale * U(-1, 1) in the operand type
        st = st * 1664525u + 1013904223u;
        const float v = ZERO ? 0.f : scale * ((float)(st >> 8) * (1.0f / 8388608.0f) - 1.0f);
        if (BF) return (unsigned)(__builtin_bit_cast(unsigned, v) >> 16);
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v);
    };
    for (int s = 0; s < 4; s++)
        for (int e = 0; e < 4; e++) { A[s][e] = rnd16() | (rnd16() << 16); B[s][e] = rnd16() | (rnd16() << 16); }
    f32x16 c32[4] = {};
    f32x4 c16[8] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (SHAPE == 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (BF) c32[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, A[u]), __builtin_bit_cast(b8, B[(u + 1) & 3]), c32[u], 0, 0, 0);
                else c32[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A[u]), __builtin_bit_cast(h8, B[(u + 1) & 3]), c32[u], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (BF) c16[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, A[u & 3]), __builtin_bit_cast(b8, B[(u + 1) & 3]), c16[u], 0, 0, 0);
                else c16[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, A[u & 3]), __builtin_bit_cast(h8, B[(u + 1) & 3]), c16[u], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int u = 0; u < 4; u++) for (int g = 0; g < 16; g++) s += c32[u][g];
    for (int u = 0; u < 8; u++) for (int g = 0; g < 4; g++) s += c16[u][g];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int BF, int ZERO>
void run(const char *tag, float *d, unsigned long long *dc, int ncu)
{
    const int iters = 600000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, BF, ZERO><<<ncu, 512>>>(d, dc, 20000, 1.0f);
    hipEventRecord(e0);
    k<SHAPE, BF, ZERO><<<ncu, 512>>>(d, dc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu);
    hipMemcpy(h.data(), dc, ncu * 8, hipMemcpyDeviceToHost);
    double cy = 0; for (auto v : h) cy += (double)v; cy /= ncu;
    const double flops = 131072.0 * iters * 8.0 * ncu;       // per wave and iteration: 4 x 32 768 or 8 x 16 384 flops; 8 waves per workgroup
    printf("%-28s %8.1f TFLOP/s over %6.1f ms   shader clock %.3f GHz   cycles per 32 768 flops and SIMD %.1f   [%s]\n", tag, flops / ms / 1e9, ms,
           cy / (ms * 1e-3) / 1e9, cy / (iters * 4.0) / 2.0, hipGetErrorString(hipGetLastError()));
}

int main()
{
    int dev = 0, ncu = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    float *d; hipMalloc(&d, (size_t)ncu * 512 * 4);
    unsigned long long *dc; hipMalloc(&dc, ncu * 8);
    for (int rep = 0; rep < 2; rep++) {
        run<0, 0, 0>("f16  32x32x16 random", d, dc, ncu);
        run<1, 0, 0>("f16  16x16x32 random", d, dc, ncu);
        run<0, 1, 0>("bf16 32x32x16 random", d, dc, ncu);
        run<1, 1, 0>("bf16 16x16x32 random", d, dc, ncu);
    }
    run<0, 0, 1>("f16  32x32x16 zeros", d, dc, ncu);
    run<1, 0, 1>("f16  16x16x32 zeros", d, dc, ncu);
    return 0;
}
