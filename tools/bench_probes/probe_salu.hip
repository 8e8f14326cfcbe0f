// tools/probe_salu.hip -- per-SIMD issue time of SALU / VALU / mixed streams, 8 waves per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    float v0 = threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0 || MODE == 2) {
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc");
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) : : "scc");
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) : : "scc");
            }
            if (MODE == 1 || MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(a));
            }
            if (MODE == 3) {   // v_cmp writing an SGPR pair + s_and on it (typical exec-mask idiom)
                unsigned long long m;
                asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(v0), "v"(a) : "scc");
                asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(v1), "v"(a) : "scc");
                asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(v2), "v"(a) : "scc");
                asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(v3), "v"(a) : "scc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + (float)(s0 + s1 + s2 + s3);
}
template <int MODE> void run(const char *tag, int per_iter)
{
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<2048, 256>>>(d, 100, 1.0001f);
    hipEventRecord(e0);
    k<MODE><<<2048, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s: %.2f ns per instruction per SIMD (8 waves per SIMD)\n", tag, ms * 1e6 / iters / per_iter / 8);
}
int main()
{
    run<0>("s_add_u32", 64);
    run<1>("v_fma_f32", 64);
    run<2>("s_add_u32 + v_fma_f32 interleaved", 128);
    run<3>("v_cmp (sgpr dst) + s_and_b64", 128);
    return 0;
}
