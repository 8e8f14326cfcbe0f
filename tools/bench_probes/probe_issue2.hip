// tools/probe_issue2.hip -- do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// block = 512 threads = 8 waves; waves 0-3 land on SIMD 0-3, waves 4-7 on SIMD 0-3 again (one pair per SIMD)
// MODE 0: all waves MFMA-only; 1: all VALU-only; 2: waves 0-3 MFMA-only, waves 4-7 VALU-only; 3: only 4 waves MFMA; 4: only 4 waves VALU
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b)
{
    const int wave = threadIdx.x >> 6;
    h8 A, B;
    for (int e = 0; e < 8; e++) { A[e] = (_Float16)(threadIdx.x * 0.001f + e); B[e] = (_Float16)(e * 0.5f); }
    f16v acc0, acc1;
    for (int g = 0; g < 16; g++) { acc0[g] = 0.f; acc1[g] = 0.f; }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 4 || (MODE == 2 && wave >= 4);
    if (do_mfma) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc1, 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v0 = __builtin_fmaf(v0, a, b); v1 = __builtin_fmaf(v1, a, b); v2 = __builtin_fmaf(v2, a, b); v3 = __builtin_fmaf(v3, a, b);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int g = 0; g < 16; g++) s += acc0[g] + acc1[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(int threads, const char *tag)
{
    float *d; hipMalloc(&d, 1024 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s: %.1f ns per iteration (8 MFMA and/or 64 FMA per wave)\n", tag, ms * 1e6 / iters);
    hipFree(d);
}
int main()
{
    run<3>(256, "4 waves (1/SIMD) MFMA-only");
    run<4>(256, "4 waves (1/SIMD) VALU-only");
    run<0>(512, "8 waves (2/SIMD) all MFMA-only");
    run<1>(512, "8 waves (2/SIMD) all VALU-only");
    run<2>(512, "8 waves: 4 MFMA-only + 4 VALU-only partners");
    return 0;
}
