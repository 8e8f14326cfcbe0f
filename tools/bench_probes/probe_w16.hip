// tools/probe_w16.hip -- would a 16-token-per-wave fused MLP (v_mfma_f32_16x16x32_f16, 8 waves = two per SIMD, 256 registers each)
// beat the 32-token-per-wave one (v_mfma_f32_32x32x16_f16, 4 waves = one per SIMD)?  Both process 128 tokens per workgroup and
// consume the same weight stream: per ring step 16 KiB of direct global->LDS loads shared by the waves, every wave reads all 16
// fragments, 24 MFMAs per wave, the step's share of the GELU arithmetic as plain VALU, one barrier.  The instruction mix per
// step is the product kernel's (gpt_kernels_c256.h); data flow is a stand-in.  Output: time per step of the WORKGROUP.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int STEP = 16384, NSLOT = 8;

template <int W16, int PF = 0>      // W16 = 1: 8 waves x 16 tokens; 0: 4 waves x 32 tokens; PF: fragments requested one chunk ahead
__global__ __launch_bounds__(W16 ? 512 : 256, W16 ? 2 : 1) void k(float *out, const unsigned char *wsrc, int nsteps)
{
    constexpr int NW = W16 ? 8 : 4, PW = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NSLOT * STEP / 4; i += NW * 64) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)smem + lane * 16;
    const unsigned char *wbase = wsrc + (size_t)(blockIdx.x % 64) * 64 * STEP + (size_t)wave * PW * 1024 + lane * 16;
    h8 B0, B1;
    for (int e = 0; e < 8; e++) { B0[e] = (_Float16)(0.001f * (lane + e)); B1[e] = (_Float16)(0.002f * e); }
    float v0 = lane, v1 = lane + 1.f, v2 = 0.5f, v3 = 0.25f;
    auto issue = [&](int s) {
        const unsigned char *src = wbase + (size_t)(s & 63) * STEP;
        unsigned char *dst = smem + (size_t)(s % NSLOT) * STEP + (size_t)wave * PW * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + i * 1024), (lds_void_t *)(dst + i * 1024), 16, 0, 0);
    };
    for (int s = 0; s < NSLOT - 1; s++) issue(s);
    if (W16 && PF) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        u32x4 f[2][4];
        auto req = [&](unsigned base, int buf) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[buf][0]) : "v"(base), "n"(0) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[buf][1]) : "v"(base), "n"(1024) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[buf][2]) : "v"(base), "n"(2048) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[buf][3]) : "v"(base), "n"(3072) : "memory");
        };
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
        __builtin_amdgcn_s_barrier();
        req(lds0, 0);
        for (int s = 0; s < nsteps; s++) {
            if (s > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            issue(s + NSLOT - 1);
            const unsigned base = lds0 + (unsigned)(s % NSLOT) * STEP, nbase = lds0 + (unsigned)((s + 1) % NSLOT) * STEP;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                req(c < 3 ? base + 4096 * (c + 1) : nbase, (c + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                u32x4 *g = f[c & 1];
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[1]), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[3]), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[0]), B1, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[2]), B1, a3, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[0]), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, g[2]), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1;
    } else if (W16) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int s = 0; s < nsteps; s++) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(s + NSLOT - 1);
            const unsigned base = lds0 + (unsigned)(s % NSLOT) * STEP;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                u32x4 f0, f1, f2, f3;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0) : "v"(base), "n"(0) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f1) : "v"(base), "n"(1024) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f2) : "v"(base), "n"(2048) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f3) : "v"(base), "n"(3072) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f1), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f3), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f0), B1, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f2), B1, a3, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f0), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f2), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1;
    } else {
        f32x16 a0, a1;
        for (int g = 0; g < 16; g++) { a0[g] = 0.f; a1[g] = 0.f; }
        for (int s = 0; s < nsteps; s++) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(s + NSLOT - 1);
            const unsigned base = lds0 + (unsigned)(s % NSLOT) * STEP;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                u32x4 f0, f1, f2, f3;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0) : "v"(base), "n"(0) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f1) : "v"(base), "n"(1024) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f2) : "v"(base), "n"(2048) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f3) : "v"(base), "n"(3072) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f1), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f3), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3); v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f0), B1, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f2), B1, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3); v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f0), B0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, f2), B0, a1, 0, 0, 0);
                v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3); v0 = fmaf(v0, v2, v3); v1 = fmaf(v1, v2, v3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + v0 + v1;
    }
}

template <int W16, int PF = 0>
void run(const char *tag, float *d, const unsigned char *w)
{
    const size_t lds = NSLOT * STEP + 24576;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<W16, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nsteps = 132, blocks = 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) k<W16, PF><<<blocks, W16 ? 512 : 256, lds>>>(d, w, nsteps);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<W16, PF><<<blocks, W16 ? 512 : 256, lds>>>(d, w, nsteps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %.3f ms per 8192-workgroup launch of 132 steps  (%.3f us per workgroup step)  [%s]\n", tag, ms,
           ms * 1e3 / (132.0 * blocks / 256), hipGetErrorString(hipGetLastError()));
}

int main()
{
    float *d; hipMalloc(&d, 8192 * 512 * 4);
    unsigned char *w; hipMalloc(&w, (size_t)64 * 64 * STEP); hipMemset(w, 0x3c, (size_t)64 * 64 * STEP);
    run<0>("4 waves x 32 tokens, 32x32x16, 1 wave/SIMD", d, w);
    run<1>("8 waves x 16 tokens, 16x16x32, 2 waves/SIMD", d, w);
    run<0>("4 waves x 32 tokens again", d, w);
    run<1>("8 waves x 16 tokens again", d, w);
    run<1, 1>("8 waves x 16 tokens, fragments one chunk ahead", d, w);
    run<1, 1>("8 waves x 16 tokens, one chunk ahead, again", d, w);
    return 0;
}
