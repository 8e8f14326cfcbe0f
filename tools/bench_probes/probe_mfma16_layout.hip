// tools/bench_probes/probe_mfma16_layout.hip -- operand / result layout of v_mfma_f32_16x16x32_bf16 (and _f16), decoded with one-hot operands:
// which (lane, element) of A is A[i][k], of B is B[k][j], and which (lane, register) of D is D[i][j].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)       // a: [64 lanes][8], b: [64][8] as floats; d: [64][4]
{
    const int l = threadIdx.x;
    b8 A, B;
    for (int e = 0; e < 8; e++) { A[e] = (__bf16)a[l * 8 + e]; B[e] = (__bf16)b[l * 8 + e]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c, 0, 0, 0);
    for (int e = 0; e < 4; e++) d[l * 4 + e] = c[e];
}
int main()
{
    float *da, *db, *dd; hipMalloc(&da, 512 * 4); hipMalloc(&db, 512 * 4); hipMalloc(&dd, 256 * 4);
    // hypothesis: A[i][k] at lane i + 16 (k / 8), element k % 8; B[k][j] at lane j + 16 (k / 8), element k % 8; D[i][j] at lane j + 16 (i / 4), register i % 4
    int bad = 0;
    for (int trial = 0; trial < 64; trial++) {
        const int i = (trial * 7) % 16, j = (trial * 5 + 3) % 16, kk = (trial * 11 + 1) % 32;
        std::vector<float> a(512, 0.f), b(512, 0.f), d(256);
        a[(i + 16 * (kk / 8)) * 8 + kk % 8] = 2.0f;
        b[(j + 16 * (kk / 8)) * 8 + kk % 8] = 3.0f;
        hipMemcpy(da, a.data(), 512 * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512 * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dd);
        hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
        int nz = 0, at = -1;
        for (int x = 0; x < 256; x++) if (d[x] != 0.f) { nz++; at = x; }
        const int want = (j + 16 * (i / 4)) * 4 + i % 4;
        if (nz != 1 || at != want || d[at] != 6.0f) { bad++; printf("trial %d (i=%d j=%d k=%d): %d nonzeros, at lane %d reg %d (want lane %d reg %d)\n", trial, i, j, kk, nz, at / 4, at % 4, want / 4, want % 4); }
    }
    printf("v_mfma_f32_16x16x32_bf16: A[i][k] = lane i + 16 (k / 8), element k %% 8; B[k][j] = lane j + 16 (k / 8), element k %% 8; D[i][j] = lane j + 16 (i / 4), register i %% 4: %s\n",
           bad ? "REFUTED" : "confirmed on 64 one-hot products");
    return bad != 0;
}
