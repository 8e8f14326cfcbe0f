// tools/probe_mfma_peak.hip -- sustained MFMA rate of the whole GPU (f16 and bf16 32x32x16, 4 independent accumulators
// per wave, no memory traffic) at 1, 2 and 4 waves per SIMD, for ~20 ms so that power management settles
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int BF>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    h8 x; b8 y;
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(threadIdx.x * 0.001f + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (BF) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a3, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int g = 0; g < 16; g++) s += a0[g] + a1[g] + a2[g] + a3[g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int BF> void run(const char *tag)
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    for (int wps : {1, 2, 4}) {
        const int iters = 40000 / wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<BF><<<256 * wps, 256>>>(d, 1000);
        hipEventRecord(e0);
        k<BF><<<256 * wps, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * (256.0 * wps * 4);
        printf("%-5s %d waves/SIMD: %7.1f TFLOP/s over %.1f ms  (= %.2f GHz if one MFMA per 32 cycles per SIMD)\n", tag, wps,
               flops / ms / 1e9, ms, flops / ms / 1e9 / 1e3 / (1024 * 2.0 * 32 * 32 * 16 / 32) * 1e3);
    }
}
int main() { run<0>("f16"); run<1>("bf16"); return 0; }
