// tools/probe_mfma.hip -- one-off hardware probe (gfx950): operand/result lane layouts of the 16-bit
// 32x32x16 MFMAs as this repo uses them, and subnormal handling of f16 inputs.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o gpurun_out/probe_mfma && gpurun_out/probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A [32][16], B [16][32] as float in global; lane (r = l&31, h = l>>5) takes slots e=0..7 <-> k = 8h+e
template <int MODE>   // 0 f16, 1 bf16
__global__ void probe(const float *A, const float *B, float *D)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16v acc;
    for (int g = 0; g < 16; g++) acc[g] = 0.f;
    if (MODE == 0) {
        h8 a, b;
        for (int e = 0; e < 8; e++) { a[e] = (_Float16)A[r * 16 + 8 * h + e]; b[e] = (_Float16)B[(8 * h + e) * 32 + r]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
        b8 a, b;
        for (int e = 0; e < 8; e++) { a[e] = (__bf16)A[r * 16 + 8 * h + e]; b[e] = (__bf16)B[(8 * h + e) * 32 + r]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int g = 0; g < 16; g++) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];   // assumed C/D map
}

int main()
{
    float hA[512], hB[512], hD[1024], ref[1024];
    srand(1);
    int bad_total = 0;
    for (int mode = 0; mode < 2; mode++) {
        for (int i = 0; i < 512; i++) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 5 - 2); }
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = 0; for (int k = 0; k < 16; k++) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
        float *dA, *dB, *dD;
        hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        if (mode == 0) probe<0><<<1, 64>>>(dA, dB, dD); else probe<1><<<1, 64>>>(dA, dB, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; i++) bad += (hD[i] != ref[i]);
        printf("layout %s: %d / 1024 mismatches\n", mode ? "bf16" : "f16", bad);
        bad_total += bad;
        if (mode == 0) {   // subnormal inputs: A = 2^-20 (f16 subnormal), B = 1 -> 16 * 2^-20 if preserved
            for (int i = 0; i < 512; i++) { hA[i] = ldexpf(1.f, -20); hB[i] = 1.f; }
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
            probe<0><<<1, 64>>>(dA, dB, dD);
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            printf("f16 subnormal A=2^-20: D[0]=%g expected %g -> %s\n", hD[0], 16 * ldexpf(1.f, -20), hD[0] == 16 * ldexpf(1.f, -20) ? "PRESERVED" : "FLUSHED/other");
            for (int i = 0; i < 512; i++) { hA[i] = ldexpf(1.f, -14) * 0.75f; hB[i] = ldexpf(1.f, -20); }   // subnormal B
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
            probe<0><<<1, 64>>>(dA, dB, dD);
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            printf("f16 subnormal A and B product: D[0]=%g expected %g\n", hD[0], 16 * ldexpf(1.f, -14) * 0.75f * ldexpf(1.f, -20));
        }
    }
    return bad_total != 0;
}
