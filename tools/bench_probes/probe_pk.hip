// tools/probe_pk.hip -- throughput of v_fma_f32 vs v_pk_fma_f32 (2 FMAs per lane) on gfx950, eight waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f2 v0 = {(float)threadIdx.x, 1.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    const f2 A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (PK) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(A), "v"(B));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(A), "v"(B));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(A), "v"(B));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(A), "v"(B));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0.x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1.x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2.x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3.x) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
}
template <int PK> void run(const char *tag)
{
    float *d; hipMalloc(&d, 1024 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<PK><<<2048, 256>>>(d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<PK><<<2048, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s: %.2f ns per instruction (8 waves per SIMD, per-SIMD issue time)\n", tag, ms * 1e6 / iters / 64 / 8);
}
int main() { run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); return 0; }
