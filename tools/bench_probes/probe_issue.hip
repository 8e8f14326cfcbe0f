// tools/probe_issue.hip -- how many VALU instructions hide behind an MFMA on gfx950?  (1 or 2 waves per SIMD)
//   kernel<NV, DEP>: repeat { 3 x v_mfma_f32_32x32x16_f16 (one accumulator) ; NV x v_fma_f32 (DEP: one dependent chain, else 4 chains) }
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, int DEP, int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b)
{
    h8 A, B;
    for (int e = 0; e < 8; e++) { A[e] = (_Float16)(threadIdx.x * 0.001f + e); B[e] = (_Float16)(e * 0.5f); }
    f16v acc[NACC];
    for (int n = 0; n < NACC; n++) for (int g = 0; g < 16; g++) acc[n][g] = 0.f;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 3; m++) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[m % NACC], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; i++) {
            if (DEP) v0 = __builtin_fmaf(v0, a, b);
            else { if ((i & 3) == 0) v0 = __builtin_fmaf(v0, a, b); else if ((i & 3) == 1) v1 = __builtin_fmaf(v1, a, b); else if ((i & 3) == 2) v2 = __builtin_fmaf(v2, a, b); else v3 = __builtin_fmaf(v3, a, b); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = v0 + v1 + v2 + v3;
    for (int n = 0; n < NACC; n++) for (int g = 0; g < 16; g++) s += acc[n][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int DEP, int NACC>
void run(int threads, const char *tag)
{
    float *d; hipMalloc(&d, 1024 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, DEP, NACC><<<256, threads>>>(d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<NV, DEP, NACC><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one block per CU (256 blocks); per SIMD: threads/256 waves
    printf("%s NV=%2d dep=%d nacc=%d waves/SIMD=%d : %.1f ns per group(3 MFMA) per wave-slot -> %.0f cycles@2.0GHz\n", tag, NV, DEP, NACC, threads / 256,
           ms * 1e6 / iters, ms * 1e6 / iters * 2.0);
    hipFree(d);
}
int main()
{
    run<0, 0, 1>(256, "A"); run<6, 0, 1>(256, "A"); run<12, 0, 1>(256, "A"); run<18, 0, 1>(256, "A"); run<24, 0, 1>(256, "A"); run<36, 0, 1>(256, "A");
    run<12, 1, 1>(256, "Adep"); run<24, 1, 1>(256, "Adep");
    run<0, 0, 3>(256, "A3acc"); run<18, 0, 3>(256, "A3acc");
    run<0, 0, 1>(512, "B"); run<6, 0, 1>(512, "B"); run<12, 0, 1>(512, "B"); run<18, 0, 1>(512, "B"); run<24, 0, 1>(512, "B"); run<36, 0, 1>(512, "B");
    run<24, 1, 1>(512, "Bdep"); run<0, 0, 3>(512, "B3acc"); run<18, 0, 3>(512, "B3acc");
    return 0;
}
