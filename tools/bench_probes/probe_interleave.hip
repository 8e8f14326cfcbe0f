// tools/probe_interleave.hip -- how many non-MFMA instructions hide in the shadow of a v_mfma_f32_32x32x16_f16 when they
// are INTERLEAVED one MFMA at a time (exact placement via inline asm), as opposed to clumped before/after a run of MFMAs
// (round 1's probe_issue.hip / probe_mfma_lds.hip measured the clumped placement and found that issue times add).
//   V<k>: k x v_fma_f32 after every MFMA          T<k>: k x v_exp_f32          C<k>: k x v_cvt_f16_f32
//   D<k>: k x ds_read_b128 after every MFMA (lgkmcnt(0) once per 4 MFMAs, data unused)
//   G   : one 1-KiB global_load_lds_dwordx4 per 4 MFMAs (+ 3 VALU and 1 ds_read per MFMA): the mix a fused MLP needs
//   L<k>: k 1-KiB global_load_lds_dwordx4 per 4 MFMAs and nothing else     R<k>: k 1-KiB global_load_dwordx4 (to registers) per 4 MFMAs
// 4 independent accumulators; 1 or 2 waves per SIMD; 256 blocks (one per CU).  Output: cycles (s_memtime) per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define REP2(x) x x
#define REP3(x) x x x
#define REP4(x) x x x x
#define REP5(x) x x x x x
#define REP6(x) x x x x x x
#define REP8(x) x x x x x x x x
// operands: 0-3 accumulators, 4-5 ds_read destinations, 6-9 VALU chains, 10 A, 11 B, 12 ca, 13 cb, 14 LDS address
#define MFMA(n) "v_mfma_f32_32x32x16_f16 %" #n ", %10, %11, %" #n "\n"
#define VF "v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"      /* two independent chains */
#define VF1 "v_fma_f32 %8, %8, %12, %13\n"
#define VE "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
#define VC "v_cvt_f16_f32 %8, %6\n v_cvt_f16_f32 %9, %7\n"
#define DS "ds_read_b128 %4, %14\n"
#define DS2 "ds_read_b128 %4, %14\n ds_read_b128 %5, %14 offset:4096\n"

// MODE: 0 V, 1 T, 2 C, 3 D, 4 G, 5 V with ONE accumulator chain (every MFMA depends on the one before), 6 V with TWO alternating chains ; K = fillers per MFMA
template <int MODE, int K, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k(float *out, long long *cyc, const unsigned char *src, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += NT) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    h8 A, B;
    for (int e = 0; e < 8; e++) { A[e] = (_Float16)(threadIdx.x * 0.001f + e); B[e] = (_Float16)(e * 0.5f); }
    f32x16 a0, a1, a2, a3;
    for (int g = 0; g < 16; g++) { a0[g] = 0.f; a1[g] = 0.f; a2[g] = 0.f; a3[g] = 0.f; }
    float v0 = threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    const float ca = 1.0001f, cb = 0.5f;
    f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
    const unsigned laddr = (unsigned)(lane * 16 + wave * 8192);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 7 || MODE == 8) {
            const unsigned char *gp = src + ((size_t)((blockIdx.x * 8 + wave) % 256) * 64 + (it & 63)) * 1024 + lane * 16;
#pragma unroll
            for (int q = 0; q < K; q++) {
                if (MODE == 7) __builtin_amdgcn_global_load_lds((gbl_void_t *)(gp + q * 65536), (lds_void_t *)(smem + 32768 + wave * 1024 + ((it + q) & 1) * 8192), 16, 0, 0);
                else if (q == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(g0) : "v"(gp) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "+v"(g1) : "v"(gp) : "memory");
            }
        }
        if (MODE == 4) {
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + ((size_t)((blockIdx.x * 8 + wave) % 256) * 64 + (it & 63)) * 1024 + lane * 16),
                                             (lds_void_t *)(smem + 32768 + wave * 1024 + (it & 1) * 8192), 16, 0, 0);
        }
#define BODY(F)                                                                                                              \
    asm volatile(MFMA(0) F MFMA(1) F MFMA(2) F MFMA(3) F                                                                     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)          \
                 : "v"(A), "v"(B), "v"(ca), "v"(cb), "v"(laddr))
#define BODY1(F)                                                                                                             \
    asm volatile(MFMA(0) F MFMA(0) F MFMA(0) F MFMA(0) F                                                                     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)          \
                 : "v"(A), "v"(B), "v"(ca), "v"(cb), "v"(laddr))
#define BODY2(F)                                                                                                             \
    asm volatile(MFMA(0) F MFMA(1) F MFMA(0) F MFMA(1) F                                                                     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)          \
                 : "v"(A), "v"(B), "v"(ca), "v"(cb), "v"(laddr))
        if (MODE == 5) {
            if (K == 0) BODY1("");
            else if (K == 4) BODY1(VF VF);
            else if (K == 6) BODY1(VF VF VF);
            else if (K == 8) BODY1(VF VF VF VF);
            else if (K == 12) BODY1(VF VF VF VF VF VF);
        } else if (MODE == 6) {
            if (K == 0) BODY2("");
            else if (K == 4) BODY2(VF VF);
            else if (K == 6) BODY2(VF VF VF);
            else if (K == 8) BODY2(VF VF VF VF);
            else if (K == 12) BODY2(VF VF VF VF VF VF);
        } else if (MODE == 0) {
            if (K == 0) BODY("");
            else if (K == 1) BODY(VF1);
            else if (K == 2) BODY(VF);
            else if (K == 3) BODY(VF VF1);
            else if (K == 4) BODY(VF VF);
            else if (K == 5) BODY(VF VF VF1);
            else if (K == 6) BODY(VF VF VF);
            else if (K == 8) BODY(VF VF VF VF);
            else if (K == 12) BODY(VF VF VF VF VF VF);
        } else if (MODE == 1) {
            if (K == 2) BODY(VE);
            else if (K == 4) BODY(VE VE);
        } else if (MODE == 2) {
            if (K == 2) BODY(VC);
            else if (K == 4) BODY(VC VC);
        } else if (MODE == 3) {
            if (K == 1) BODY(DS);
            else if (K == 2) BODY(DS2);
            else if (K == 3) BODY(DS2 VF VF1);         // 2 ds_read + 3 VALU
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 7 || MODE == 8) {
            BODY("");
            if ((it & 1) == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");      // the loads of the last iteration may stay in flight
        } else if (MODE == 4) {
            BODY(DS VF VF1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((it & 1) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");             // MFMA results settle before compiler code reads them
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1)::"memory");
    float s = v0 + v1 + v2 + v3 + d0[0] + d1[0] + g0[0] + g1[0];
    for (int g = 0; g < 16; g++) s += a0[g] + a1[g] + a2[g] + a3[g];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (NT / 64) + wave] = t1 - t0;
}

template <int MODE, int K, int NT>
void run(const char *tag, float *d, long long *dc, const unsigned char *src)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, K, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, K, NT><<<256, NT, 65536>>>(d, dc, src, 200);
    hipEventRecord(e0);
    k<MODE, K, NT><<<256, NT, 65536>>>(d, dc, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = (double)iters * 4;                       // MFMAs per wave
    const double wps = NT / 256.0;                             // waves per SIMD
    printf("%-34s waves/SIMD=%d : %6.2f ns per MFMA per SIMD  (%5.0f TFLOP/s)  counter ticks per MFMA per wave %.1f\n", tag, (int)wps,
           ms * 1e6 / (nm * wps), 2.0 * 32 * 32 * 16 * nm * wps * 1024 / (ms * 1e-3) / 1e12, (double)h[0] / nm);
}

#define BOTH(M, K_, tag) run<M, K_, 256>(tag, d, dc, src); run<M, K_, 512>(tag, d, dc, src);
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool only_new = argc > 1;
    float *d; hipMalloc(&d, 256 * 512 * 4);
    long long *dc; hipMalloc(&dc, 256 * 8 * 8);
    unsigned char *src; hipMalloc(&src, 256 * 64 * 1024 + (1 << 20)); hipMemset(src, 1, 256 * 64 * 1024 + (1 << 20));
    if (!only_new) {
    BOTH(5, 0, "1 chain: MFMA only");
    BOTH(5, 4, "1 chain: MFMA + 4 v_fma");
    BOTH(5, 6, "1 chain: MFMA + 6 v_fma");
    BOTH(5, 8, "1 chain: MFMA + 8 v_fma");
    BOTH(5, 12, "1 chain: MFMA + 12 v_fma");
    BOTH(6, 0, "2 chains: MFMA only");
    BOTH(6, 6, "2 chains: MFMA + 6 v_fma");
    BOTH(6, 8, "2 chains: MFMA + 8 v_fma");
    BOTH(6, 12, "2 chains: MFMA + 12 v_fma");
    BOTH(0, 0, "MFMA only");
    BOTH(0, 1, "MFMA + 1 v_fma");
    BOTH(0, 2, "MFMA + 2 v_fma");
    BOTH(0, 3, "MFMA + 3 v_fma");
    BOTH(0, 4, "MFMA + 4 v_fma");
    BOTH(0, 5, "MFMA + 5 v_fma");
    BOTH(0, 6, "MFMA + 6 v_fma");
    BOTH(0, 8, "MFMA + 8 v_fma");
    BOTH(0, 12, "MFMA + 12 v_fma");
    BOTH(1, 2, "MFMA + 2 v_exp");
    BOTH(1, 4, "MFMA + 4 v_exp");
    BOTH(2, 2, "MFMA + 2 v_cvt_f16");
    BOTH(2, 4, "MFMA + 4 v_cvt_f16");
    BOTH(3, 1, "MFMA + 1 ds_read_b128");
    BOTH(3, 2, "MFMA + 2 ds_read_b128");
    BOTH(3, 3, "MFMA + 2 ds_read_b128 + 3 v_fma");
    BOTH(4, 0, "MFMA + 1 ds_read + 3 v_fma + DMA/4");
    }
    BOTH(0, 0, "MFMA only");
    BOTH(7, 1, "MFMA + 1 DMA per 4 MFMAs");
    BOTH(7, 2, "MFMA + 2 DMA per 4 MFMAs");
    BOTH(8, 1, "MFMA + 1 global_load_dwordx4 per 4");
    BOTH(8, 2, "MFMA + 2 global_load_dwordx4 per 4");
    return 0;
}
