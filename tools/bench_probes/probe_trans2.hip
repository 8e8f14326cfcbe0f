// tools/probe_trans2.hip -- do transcendental ops (v_exp_f32) overlap with ordinary VALU ops on a SIMD?  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NF, int NE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    float v[8], e[4];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; i++) e[i] = threadIdx.x * 1e-4f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NF; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a));
#pragma unroll
            for (int i = 0; i < NE; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NF, int NE> void run(float *d)
{
    const int iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NF, NE><<<2048, 256>>>(d, 50, 1.0001f);
    hipEventRecord(e0);
    k<NF, NE><<<2048, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d v_fma + %d v_exp per group: %.2f ns per group per SIMD (8 waves per SIMD)\n", NF, NE, ms * 1e6 / iters / 8 / 8);
}
int main()
{
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    run<8, 0>(d); run<0, 4>(d); run<8, 4>(d); run<8, 2>(d);
    return 0;
}
