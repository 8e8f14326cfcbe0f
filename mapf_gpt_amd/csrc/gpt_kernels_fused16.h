// gpt_kernels_fused16.h -- mlp_fused_kernel (gpt_kernels_fast.h: the whole MLP block of the C = 64 / 160 shapes in one kernel, hidden activations in registers -- read its
// header first) on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16: at the package power limit the small shape delivers 13-15 % more f16 flops per second (DESIGN
// section 10 fact 5; the 6M block: gpt_kernels_c256q.h).  The kernel's structure does not change -- packets of [c_fc fragments | c_proj fragments] per 32-unit hidden
// tile through a 3-slot ring, counted waits, one barrier per tile -- only the slicing (lane l: t = l % 16, q = l / 16; a wave's 32 tokens are the groups tg = 0, 1):
//   c_fc    fragment (kb, ug) = 16 hidden units x the 32 features of k-block kb, row rho = unit 8 (rho / 4) + 4 ug + rho % 4; operand planes xn[tg * KB + kb] = token
//           16 tg + t, features 32 kb + 8 q .. + 7; D(ug, tg) = registers 4 (2 tg + ug) .. + 3 of the tile's 16: token 16 tg + t, units 8 q + 4 ug + i
//   GELU    a lane's quads (0, tg), (1, tg) are the units 8 q .. 8 q + 7: one K = 32 operand of c_proj per token group, formed in registers
//   c_proj  fragment fg = output features 16 fg + rho x the tile's 32 units; D(fg, tg) = registers 4 (2 (fg % 2) + tg) .. + 3 of acc[fg / 2]
//   x       chunk-major rows: lane (t, q) owns the 32 bytes of token 16 tg + t in chunk 4 kb + q (operand planes) resp. the 16 bytes at half q % 2 of chunk
//           2 fg + q / 2 (residual quads); LayerNorm folds over the four lanes of a token
#pragma once
#include "gpt_kernels_fast.h"

namespace mgpt {
namespace fastk {

template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp16_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                         uint16_t *__restrict__ out, int C, float scale1, float scale2)
{
    // one thread = one (hidden tile t, fragment f, lane): 8 k-slots, both planes.  Packet of tile t: [c_fc fragments (kb, ug): f = 2 kb + ug | c_proj fragments fg]
    const int KB = C / 32, FG = C / 16;
    const int frags = 2 * KB + FG;
    const int NT = 4 * C / 32;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)NT * frags * 64) return;
    const int lane = (int)(gid & 63);
    const int f = (int)((gid >> 6) % frags), t = (int)((gid >> 6) / frags);
    const int rho = lane & 15, qk = lane >> 4;
    float v[8];
    if (f < 2 * KB) {                                      // c_fc: rows = hidden units (permuted: a lane's two result quads are 8 consecutive units), k = features
        const int kb = f >> 1, ug = f & 1, unit = 32 * t + 8 * (rho >> 2) + 4 * ug + (rho & 3);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fc_w[(size_t)unit * C + 32 * kb + 8 * qk + e] * scale1;
    } else {                                               // c_proj: rows = output features 16 fg + rho, k = the tile's 32 units
        const int fg = f - 2 * KB;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = pj_w[(size_t)(16 * fg + rho) * (4 * C) + 32 * t + 8 * qk + e] * scale2;
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)t * frags + f) * NP) * 512 + (size_t)lane * 8;      // [t][f][plane][lane][8 halfs]
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

#if defined(MGPT_AB_MLPF16_CLUMPED)
constexpr bool kMlpF16Placed = false;
#else
constexpr bool kMlpF16Placed = true;
#endif

template <class T, int NP, int CT, int NW = 8, int NFOLD = 0, int NBUF = 3>
__global__ __launch_bounds__(NW * 64, 2) void mlp_fused16_kernel(float *__restrict__ x, const float *__restrict__ gain,
                                                              const uint16_t *__restrict__ wpk, float inv1, float inv2,
                                                              float2 *__restrict__ stats_out, int M,
                                                              const float2 *__restrict__ gelu_lut,
                                                              const float *__restrict__ fold = nullptr, int64_t fold_stride = 0)
{
    constexpr int C = CT * 32, KB = CT, FG = 2 * CT, NT = 4 * CT;
    constexpr int LUT_BYTES = kGeluLutN * 8;               // the Phi table sits behind the ring: [NBUF][PKT][LUT]
    static_assert((LUT_BYTES / 1024) % NW == 0, "every wave stages the same number of table pieces");
    constexpr int FRAGS = 2 * KB + FG;                     // fragments per hidden tile (as many as the 32 x 32 form had)
    constexpr int PKT = FRAGS * NP * 1024;
    constexpr int PER_WAVE = (FRAGS * NP + NW - 1) / NW;
    constexpr int NM = NP == 2 ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NBUF][PKT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t16 = lane & 15, q4 = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * (NW * 32) + wave * 32;        // this wave's first token
    // x is chunk-major (xt_off): chunk c (8 features) of the wave's 32-token tile sits c * 256 floats up, token T at T * 8
    float *xt = x + m0 * C + q4 * 256 + t16 * 8;                           // operand side: chunk 4 kb + q, tokens t (+ 16): 32 bytes each
    float *xq = x + m0 * C + (q4 >> 1) * 256 + t16 * 8 + 4 * (q4 & 1);     // result side: chunk 2 fg + q / 2, half q % 2: 16 bytes

    auto issue = [&](int t) {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)t * PKT;
        unsigned char *dst = smem + (size_t)(t % NBUF) * PKT;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            // every wave issues exactly PER_WAVE pieces so that the counted vmcnt waits below are exact; a wave whose share runs past the packet re-loads the last piece
            const int c = min(wave + NW * i, FRAGS * NP - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)c * 1024 + lane * 16), (lds_void_t *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
    };
    {   // GELU table -> LDS; older than every ring piece, so the first counted wait covers it
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut);
        unsigned char *dst = smem + (size_t)NBUF * PKT;
#pragma unroll
        for (int i = 0; i < LUT_BYTES / 1024 / NW; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(wave + NW * i) * 1024 + lane * 16), (lds_void_t *)(dst + (size_t)(wave + NW * i) * 1024), 16, 0, 0);
    }
    issue(0);
    if (NBUF > 2) issue(1);
    const unsigned lut_addr = (unsigned)(size_t)(smem + (size_t)NBUF * PKT);
    const float lut_scale = inv1 * kGeluLutScale;

    // ---- the wave's 32 x C row block as operand pieces: xr[tg][kb][hf] = features 32 kb + 8 q + 4 hf .. + 3 of token 16 tg + t; LayerNorm over four lanes ----
    f32x4 xr[2][KB][2];
#pragma unroll
    for (int tg = 0; tg < 2; tg++)
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) xr[tg][kb][hf] = *reinterpret_cast<const f32x4 *>(xt + kb * 1024 + tg * 128 + hf * 4);
    if constexpr (NFOLD > 0) {                             // small launches: the heads' partial sums, in x's layout, are added in index order and the sum written back
        const float *fp = fold + (xt - x);
#pragma unroll
        for (int p = 0; p < NFOLD; p++) {
            f32x4 tt[2][KB][2];
#pragma unroll
            for (int tg = 0; tg < 2; tg++)
#pragma unroll
                for (int kb = 0; kb < KB; kb++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) tt[tg][kb][hf] = *reinterpret_cast<const f32x4 *>(fp + (size_t)p * fold_stride + kb * 1024 + tg * 128 + hf * 4);
#pragma unroll
            for (int tg = 0; tg < 2; tg++)
#pragma unroll
                for (int kb = 0; kb < KB; kb++)
#pragma unroll
                    for (int hf = 0; hf < 2; hf++)
#pragma unroll
                        for (int e = 0; e < 4; e++) xr[tg][kb][hf][e] += tt[tg][kb][hf][e];
        }
#pragma unroll
        for (int tg = 0; tg < 2; tg++)
#pragma unroll
            for (int kb = 0; kb < KB; kb++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++) *reinterpret_cast<f32x4 *>(xt + kb * 1024 + tg * 128 + hf * 4) = xr[tg][kb][hf];
    }
    auto fold4 = [&](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };      // a token's features sit in the four lanes t + 16 q
    u32x4 xn[2 * KB][2];                                   // B operand of c_fc: [tg * KB + kb][plane]
#pragma unroll
    for (int tg = 0; tg < 2; tg++) {
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) s += (xr[tg][kb][hf][0] + xr[tg][kb][hf][1]) + (xr[tg][kb][hf][2] + xr[tg][kb][hf][3]);
        const float mean = fold4(s) / (float)C;
        float qv = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; kb++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int e = 0; e < 4; e++) { const float d = xr[tg][kb][hf][e] - mean; qv += d * d; }
        const float rstd = rsqrtf(fold4(qv) / (float)C + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gain + 32 * kb + 8 * q4);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(gain + 32 * kb + 8 * q4 + 4);
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = (xr[tg][kb][0][e] - mean) * rstd * ga[e];
                v1[e] = (xr[tg][kb][1][e] - mean) * rstd * gb[e];
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            xn[tg * KB + kb][0][0] = h0[0]; xn[tg * KB + kb][0][1] = h0[1]; xn[tg * KB + kb][0][2] = h1[0]; xn[tg * KB + kb][0][3] = h1[1];
            xn[tg * KB + kb][1][0] = l0[0]; xn[tg * KB + kb][1][1] = l0[1]; xn[tg * KB + kb][1][2] = l1[0]; xn[tg * KB + kb][1][3] = l1[1];
        }
    }
    f32x16 acc[CT];                                        // output accumulators: acc[fg / 2][4 (2 (fg % 2) + tg) + i]
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
    auto mm16 = [&](const u32x4 &a, const u32x4 &b, f32x16 &blk, auto s_c) {     // one product term on registers 4 S .. 4 S + 3 of a 16-register block
        constexpr int S = decltype(s_c)::value;
        f32x4 c = {blk[4 * S], blk[4 * S + 1], blk[4 * S + 2], blk[4 * S + 3]};
        c = T::mfma16(a, b, c);
        blk[4 * S] = c[0]; blk[4 * S + 1] = c[1]; blk[4 * S + 2] = c[2]; blk[4 * S + 3] = c[3];
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

    // packet 0 must have landed; with 3 buffers packet 1 (the newest PER_WAVE pieces of this wave) may still fly
    if (NBUF > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int t = 0; t < NT; t++) {
        if (t + NBUF - 1 < NT) issue(t + NBUF - 1);       // refill the buffer that was read during tile t-1 (all waves are past the barrier that ended it)
        const unsigned char *pk = smem + (size_t)(t % NBUF) * PKT + lane * 16;
        // ---- c_fc: D(ug, tg) at hd[4 (2 tg + ug) + i]; per k-block the fragments (kb, 0), (kb, 1) against xn[kb], xn[KB + kb]: four chains ----
        f32x16 hd;
#pragma unroll
        for (int g = 0; g < 16; g++) hd[g] = 0.f;
        {
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                u32x4 w[2][2];
#pragma unroll
                for (int ug = 0; ug < 2; ug++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) w[ug][pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((2 * kb + ug) * NP + pl) * 1024);
                if (NP == 2) {
                    mm16(w[0][1], xn[kb][0], hd, S0{}); mm16(w[1][1], xn[kb][0], hd, S1{}); mm16(w[0][1], xn[KB + kb][0], hd, S2{}); mm16(w[1][1], xn[KB + kb][0], hd, S3{});
                    mm16(w[0][0], xn[kb][1], hd, S0{}); mm16(w[1][0], xn[kb][1], hd, S1{}); mm16(w[0][0], xn[KB + kb][1], hd, S2{}); mm16(w[1][0], xn[KB + kb][1], hd, S3{});
                }
                mm16(w[0][0], xn[kb][0], hd, S0{}); mm16(w[1][0], xn[kb][0], hd, S1{}); mm16(w[0][0], xn[KB + kb][0], hd, S2{}); mm16(w[1][0], xn[KB + kb][0], hd, S3{});
            }
            // fragment reads run one k-block ahead of the MFMAs that consume them, one behind each of the block's first MFMAs (-DMGPT_AB_MLPF16_CLUMPED: the
            // four of them in front of the block)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                if (kMlpF16Placed && kb + 1 < KB) {
#pragma unroll
                    for (int n = 0; n < 2 * NP; n++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NM - 2 * NP, 0);
                } else {
                    if (kb + 1 < KB) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GELU by table: all 16 gathers of the tile go out first (asm: invisible to hipcc's LDS-DMA ordering), one wait, then interpolate ----
        float gv[16], gf[16];
        f32x2 gt[16];
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float hv = hd[g];
            gv[g] = hv * inv1;
            const float tt = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
            gf[g] = __builtin_amdgcn_fractf(tt);
            asm volatile("ds_read_b64 %0, %1" : "=v"(gt[g]) : "v"(lut_addr + (unsigned)tt * 8u) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; g++) asm volatile("" : "+v"(gt[g]));
        u32x4 hf[2][2];                                    // [tg][plane]: B operand of c_proj = the tile's units 8 q .. 8 q + 7 of token 16 tg + t
#pragma unroll
        for (int tg = 0; tg < 2; tg++) {
            float v0[4], v1[4];                            // unit group 0, 1: registers 4 (2 tg) .. and 4 (2 tg + 1) ..
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = gv[8 * tg + e] * fmaf(gf[8 * tg + e], gt[8 * tg + e][1], gt[8 * tg + e][0]);
                v1[e] = gv[8 * tg + 4 + e] * fmaf(gf[8 * tg + 4 + e], gt[8 * tg + 4 + e][1], gt[8 * tg + 4 + e][0]);
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            hf[tg][0][0] = h0[0]; hf[tg][0][1] = h0[1]; hf[tg][0][2] = h1[0]; hf[tg][0][3] = h1[1];
            hf[tg][1][0] = l0[0]; hf[tg][1][1] = l0[1]; hf[tg][1][2] = l1[0]; hf[tg][1][3] = l1[1];
        }
        // ---- c_proj: per pair of feature groups (2 j, 2 j + 1) the four quads of acc[j] ----
        {
#pragma unroll
            for (int j = 0; j < CT; j++) {
                u32x4 w[2][2];
#pragma unroll
                for (int fo = 0; fo < 2; fo++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) w[fo][pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((2 * KB + 2 * j + fo) * NP + pl) * 1024);
                if (NP == 2) {
                    mm16(w[0][1], hf[0][0], acc[j], S0{}); mm16(w[0][1], hf[1][0], acc[j], S1{}); mm16(w[1][1], hf[0][0], acc[j], S2{}); mm16(w[1][1], hf[1][0], acc[j], S3{});
                    mm16(w[0][0], hf[0][1], acc[j], S0{}); mm16(w[0][0], hf[1][1], acc[j], S1{}); mm16(w[1][0], hf[0][1], acc[j], S2{}); mm16(w[1][0], hf[1][1], acc[j], S3{});
                }
                mm16(w[0][0], hf[0][0], acc[j], S0{}); mm16(w[0][0], hf[1][0], acc[j], S1{}); mm16(w[1][0], hf[0][0], acc[j], S2{}); mm16(w[1][0], hf[1][0], acc[j], S3{});
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 1);
#pragma unroll
            for (int j = 0; j < CT; j++) {
                if (kMlpF16Placed && j + 1 < CT) {
#pragma unroll
                    for (int n = 0; n < 2 * NP; n++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NM - 2 * NP, 1);
                } else {
                    if (j + 1 < CT) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NM, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // packet t+1 must have landed before anyone reads it; the pieces of packet t+2 issued at the top of this iteration may stay in flight across the barrier
        if (NBUF > 2 && t + NBUF - 1 < NT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- residual add, store, LayerNorm statistics of the new rows: quad (fg, tg) = features 16 fg + 4 q .. + 3 of token 16 tg + t at chunk 2 fg + q / 2, half q % 2 ----
    float *xq2 = xq;
    if (NP == 2) asm volatile("" : "+v"(xq2));            // (addresses formed again from an opaque copy: see mlp_fused_kernel)
    float s2[2] = {0.f, 0.f};
    f32x4 cur[CT][4];
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) cur[j][gq] = *reinterpret_cast<const f32x4 *>(xq2 + (2 * (2 * j + (gq >> 1))) * 256 + (gq & 1) * 128);
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            f32x4 c = cur[j][gq];
#pragma unroll
            for (int e = 0; e < 4; e++) { c[e] += acc[j][4 * gq + e] * inv2; acc[j][4 * gq + e] = c[e]; }
            *reinterpret_cast<f32x4 *>(xq2 + (2 * (2 * j + (gq >> 1))) * 256 + (gq & 1) * 128) = c;
            s2[gq & 1] += (c[0] + c[1]) + (c[2] + c[3]);
        }
    if (stats_out != nullptr) {
#pragma unroll
        for (int tg = 0; tg < 2; tg++) {
            const float mean2 = fold4(s2[tg]) / (float)C;
            float q2 = 0.f;
#pragma unroll
            for (int j = 0; j < CT; j++)
#pragma unroll
                for (int gq = tg; gq < 4; gq += 2)
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float d = acc[j][4 * gq + e] - mean2; q2 += d * d; }
            q2 = fold4(q2);
            if (q4 == 0) stats_out[m0 + 16 * tg + t16] = make_float2(mean2, rsqrtf(q2 / (float)C + 1e-5f));
        }
    }
}

}  // namespace fastk
}  // namespace mgpt
