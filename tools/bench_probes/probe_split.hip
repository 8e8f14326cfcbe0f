// tools/probe_split.hip -- does (hi, lo) = (f16(v), f16(v - hi)) reconstruct v?  (gfx950, hipcc -O3)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t cvt(float v) { _Float16 x = (_Float16)v; return __builtin_bit_cast(uint16_t, x); }
__device__ __forceinline__ float back(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__global__ void k(const float* in, u32x2* hi, u32x2* lo, float s, int n4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float v[4] = {in[4*i]*s, in[4*i+1]*s, in[4*i+2]*s, in[4*i+3]*s};
    uint16_t a[4], b[4];
    for (int e = 0; e < 4; e++) { a[e] = cvt(v[e]); b[e] = cvt(v[e] - back(a[e])); }
    u32x2 h, l;
    h[0] = a[0] | (a[1] << 16); h[1] = a[2] | (a[3] << 16);
    l[0] = b[0] | (b[1] << 16); l[1] = b[2] | (b[3] << 16);
    hi[i] = h; lo[i] = l;
}
static float h2f(uint16_t h) { int s = h >> 15, e = (h >> 10) & 31, m = h & 1023; float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m + 1024), e - 25); return s ? -v : v; }
int main() {
    const int n = 1 << 22;
    float* h = (float*)malloc(n * 4); uint16_t *hh = (uint16_t*)malloc(n * 2), *hl = (uint16_t*)malloc(n * 2);
    srand(3);
    for (int i = 0; i < n; i++) h[i] = ((rand() & 0xffffff) / 16777216.0f - 0.5f) * 4.0f;
    // plant exact fp16 ties (after the multiply by s = 0.75): v = (m + 0.5) * 2^-12
    float s = 0.75f;
    float *di; u32x2 *dh, *dl;
    hipMalloc(&di, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2);
    hipMemcpy(di, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 4 / 256, 256>>>(di, dh, dl, s, n / 4);
    hipMemcpy(hh, dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hl, dl, n * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        double e = (double)h[i] * (double)s, got = (double)h2f(hh[i]) + (double)h2f(hl[i]);
        double err = fabs(got - e);
        if (err > 4e-7 * fmax(fabs(e), 1e-3)) {
            if (bad < 8) printf("bad i=%d in=%.9g e=%.10g hi=%.9g lo=%.4g sum=%.10g v32=%.9g\n", i, h[i], e, h2f(hh[i]), h2f(hl[i]), got, (float)(h[i] * s));
            bad++;
        }
    }
    printf("split probe: %d / %d bad\n", bad, n);
    return 0;
}
