// tools/probe_mfma_lds2.hip -- the same 256x256x16 block k-step (256 bf16 MFMAs per CU, 16 KiB refill) done by
//   A: 4 waves (one per SIMD), 128x128 per wave: 16 MFMAs + 8 ds_read_b128 + 4 LDS-DMA per wave
//   B: 8 waves (two per SIMD),  64x128 per wave:  8 MFMAs + 6 ds_read_b128 + 2 LDS-DMA per wave
// with one workgroup barrier per k-step.  Does the second wave hide the non-MFMA issue time?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, NWAVE / 4) void k(float *out, const unsigned char *src, int iters)
{
    constexpr int TM = (NWAVE == 4) ? 4 : 2, TN = 4, NDMA = 16 / NWAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 96 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24576; i += NWAVE * 64) reinterpret_cast<unsigned *>(smem)[i] = i * 2654435761u >> 9;
    __syncthreads();
    f32x16 acc[TM * TN];
    for (int a = 0; a < TM * TN; a++) for (int g = 0; g < 16; g++) acc[a][g] = 0.f;
    u32x4 fa[2][TM], fb[2][TN];
    for (int q = 0; q < TM; q++) fa[0][q] = fa[1][q] = *reinterpret_cast<const u32x4 *>(smem + q * 1024 + lane * 16);
    for (int q = 0; q < TN; q++) fb[0][q] = fb[1][q] = *reinterpret_cast<const u32x4 *>(smem + (8 + q) * 1024 + lane * 16);
    const int wm = (NWAVE == 4) ? (wave >> 1) : (wave >> 1), wn = wave & 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const unsigned char *st = smem + ((it * 2 + half) % 6) * 16384 + lane * 16;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < NDMA; i++)
                __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + ((size_t)(blockIdx.x % 64) * 96 + (it * 8 + half * 4 + i) % 96) * 1024 + lane * 16),
                                                 (lds_void_t *)(smem + ((it * 2 + half + 5) % 6) * 16384 + (wave + NWAVE * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < TM; q++) fa[half ^ 1][q] = *reinterpret_cast<const u32x4 *>(st + ((wm * TM + q) & 7) * 1024);
#pragma unroll
            for (int q = 0; q < TN; q++) fb[half ^ 1][q] = *reinterpret_cast<const u32x4 *>(st + (8 + wn * 4 + q) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i * TN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, fb[half][j]), __builtin_bit_cast(b8, fa[half][i]), acc[i * TN + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < TM * TN; a++) for (int g = 0; g < 16; g++) s += acc[a][g];
    out[blockIdx.x * NWAVE * 64 + threadIdx.x] = s;
}
template <int NWAVE> void run(const char *tag, const unsigned char *src, float *d)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NWAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NWAVE><<<256, NWAVE * 64, 98304>>>(d, src, 100);
    hipEventRecord(e0);
    k<NWAVE><<<256, NWAVE * 64, 98304>>>(d, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.1f ns per block k-step (256 MFMAs per CU = %.0f TFLOP/s)\n", tag, ms * 1e6 / (iters * 2), 2.0 * 32 * 32 * 16 * 256 * 256 / (ms * 1e6 / (iters * 2)) / 1e3);
}
int main()
{
    unsigned char *src; hipMalloc(&src, 64 * 96 * 1024); hipMemset(src, 1, 64 * 96 * 1024);
    float *d; hipMalloc(&d, 256 * 512 * 4);
    run<4>("4 waves x (16 MFMA + 8 ds_read + 4 DMA)", src, d);
    run<8>("8 waves x ( 8 MFMA + 6 ds_read + 2 DMA)", src, d);
    return 0;
}
