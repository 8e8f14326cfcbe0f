// tools/probe_lds.hip -- LDS read bandwidth per CU by access width (fragment-style: lane * width contiguous), 4 or 8 waves per CU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 64 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int f = 0; f < 16; f++) {
            const unsigned char *frag = smem + ((f + wave * 3 + it) & 63) * 1024;      // one 1 KiB fragment
            if (MODE == 0) { u32x4 v = *reinterpret_cast<const u32x4 *>(frag + lane * 16); acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
            if (MODE == 1) { u32x2 a = *reinterpret_cast<const u32x2 *>(frag + lane * 8), b = *reinterpret_cast<const u32x2 *>(frag + 512 + lane * 8);
                             acc += a[0] ^ a[1] ^ b[0] ^ b[1]; }
            if (MODE == 2) { unsigned a = *reinterpret_cast<const unsigned *>(frag + lane * 4), b = *reinterpret_cast<const unsigned *>(frag + 256 + lane * 4),
                                      c = *reinterpret_cast<const unsigned *>(frag + 512 + lane * 4), d = *reinterpret_cast<const unsigned *>(frag + 768 + lane * 4);
                             acc += a ^ b ^ c ^ d; }
            if (MODE == 3) { u32x2 a = *reinterpret_cast<const u32x2 *>(frag + lane * 16), b = *reinterpret_cast<const u32x2 *>(frag + lane * 16 + 8);   // same bytes as b128, as two b64
                             asm volatile("" : "+v"(a), "+v"(b)); acc += a[0] ^ a[1] ^ b[0] ^ b[1]; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char *tag)
{
    unsigned *d; hipMalloc(&d, 256 * 512 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int threads : {256, 512}) {
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256, threads, 65536>>>(d, 10);
        hipEventRecord(e0);
        k<MODE><<<256, threads, 65536>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)iters * 16 * 1024 * (threads / 64);      // per CU (one block per CU)
        printf("%-28s %d waves/CU: %.1f GB/s per CU  (%.1f B/clk at 2.4 GHz)\n", tag, threads / 64, bytes / ms / 1e6, bytes / ms / 1e6 / 2.4);
    }
}
int main()
{
    run<0>("ds_read_b128 lane*16"); run<1>("2 x ds_read_b64 lane*8"); run<2>("4 x ds_read_b32 lane*4"); run<3>("2 x b64 at lane*16 (+8)");
    return 0;
}
