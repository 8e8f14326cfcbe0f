// tools/probe_trans.hip -- per-SIMD issue time of transcendental / conversion VALU ops vs v_fma_f32, 2 and 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OP4(str)                                                     \
    asm volatile(str " %0, %0" : "+v"(v0)); asm volatile(str " %0, %0" : "+v"(v1)); \
    asm volatile(str " %0, %0" : "+v"(v2)); asm volatile(str " %0, %0" : "+v"(v3));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(a));
                             asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(a)); }
            if (MODE == 1) { OP4("v_exp_f32") }
            if (MODE == 2) { OP4("v_rcp_f32") }
            if (MODE == 3) { OP4("v_cvt_f16_f32") }
            if (MODE == 4) { OP4("v_cvt_f32_f16") }
            if (MODE == 5) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v0) : "v"(a)); asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v1) : "v"(a));
                             asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v2) : "v"(a)); asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v3) : "v"(a)); }
            if (MODE == 6) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(v0) : "v"(a)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(v1) : "v"(a));
                             asm volatile("v_max_f32 %0, %0, %1" : "+v"(v2) : "v"(a)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(v3) : "v"(a)); }
            if (MODE == 7) { asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v0) : "v"(1)); asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v1) : "v"(1));
                             asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v2) : "v"(1)); asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v3) : "v"(1)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
}
template <int MODE> void run(const char *tag)
{
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    const int iters = 10000;
    for (int wps : {2, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256 * wps, 256>>>(d, 100, 1.0001f);
        hipEventRecord(e0);
        k<MODE><<<256 * wps, 256>>>(d, iters, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-18s %d waves/SIMD: %.2f ns per instruction per SIMD\n", tag, wps, ms * 1e6 / iters / 64 / wps);
    }
}
int main()
{
    run<0>("v_fma_f32"); run<1>("v_exp_f32"); run<2>("v_rcp_f32"); run<3>("v_cvt_f16_f32"); run<4>("v_cvt_f32_f16");
    run<5>("v_cvt_pk_f16_f32"); run<6>("v_max_f32"); run<7>("v_ldexp_f32");
    return 0;
}
