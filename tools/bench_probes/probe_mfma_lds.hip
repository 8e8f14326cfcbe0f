// tools/probe_mfma_lds.hip -- does a k-step of the packed GEMM (16 independent bf16 MFMAs + 8 ds_read_b128 of the next
// fragments [+ s_barrier] [+ 4 direct-to-LDS loads]) cost more than its 16 MFMAs?  4 waves per CU (one per SIMD), 256 CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <int MODE>     // bit0: ds_reads, bit1: barrier, bit2: DMA, bit3: no MFMA
__global__ __launch_bounds__(256, 1) void k(float *out, const unsigned char *src, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 96 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24576; i += 256) reinterpret_cast<unsigned *>(smem)[i] = i * 2654435761u >> 9;
    __syncthreads();
    f32x16 acc[16];
    for (int a = 0; a < 16; a++) for (int g = 0; g < 16; g++) acc[a][g] = 0.f;
    u32x4 f[2][8];
    for (int q = 0; q < 8; q++) { f[0][q] = *reinterpret_cast<const u32x4 *>(smem + q * 1024 + lane * 16); f[1][q] = f[0][q]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const unsigned char *st = smem + ((it * 2 + half) % 6) * 16384 + lane * 16;
            if (MODE & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            if (MODE & 4) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + ((size_t)(blockIdx.x % 64) * 96 + (it * 8 + half * 4 + i) % 96) * 1024 + lane * 16),
                                                     (lds_void_t *)(smem + ((it * 2 + half + 5) % 6) * 16384 + (wave + 4 * i) * 1024), 16, 0, 0);
            }
            if (MODE & 1) {
#pragma unroll
                for (int q = 0; q < 8; q++) f[half ^ 1][q] = *reinterpret_cast<const u32x4 *>(st + ((q < 4 ? (wave >> 1) * 4 + q : 8 + (wave & 1) * 4 + q - 4)) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(MODE & 8)) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, f[half][4 + j]), __builtin_bit_cast(b8, f[half][i]), acc[i * 4 + j], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(f[half][0]), "v"(f[half][7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float s = 0.f;
    for (int a = 0; a < 16; a++) for (int g = 0; g < 16; g++) s += acc[a][g];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *tag, const unsigned char *src, float *d)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256, 98304>>>(d, src, 100);
    hipEventRecord(e0);
    k<MODE><<<256, 256, 98304>>>(d, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f ns per k-step (16 MFMAs = %.0f TFLOP/s)\n", tag, ms * 1e6 / (iters * 2), 2.0 * 32 * 32 * 16 * 16 * 1024 / (ms * 1e6 / (iters * 2)) / 1e3);
}
int main()
{
    unsigned char *src; hipMalloc(&src, 64 * 96 * 1024); hipMemset(src, 1, 64 * 96 * 1024);
    float *d; hipMalloc(&d, 256 * 256 * 4);
    run<0>("MFMA only", src, d);
    run<1>("MFMA + 8 ds_read_b128", src, d);
    run<3>("MFMA + ds_reads + barrier", src, d);
    run<7>("MFMA + ds_reads + barrier + 4 LDS-DMA", src, d);
    run<5>("MFMA + ds_reads + 4 LDS-DMA", src, d);
    run<4>("MFMA + 4 LDS-DMA", src, d);
    run<9>("ds_reads only", src, d);
    run<12>("LDS-DMA only", src, d);
    return 0;
}
