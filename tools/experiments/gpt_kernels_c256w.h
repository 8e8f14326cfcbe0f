// gpt_kernels_c256w.h -- the fused MLP block for C = 256 with TWO waves per SIMD (16 tokens per wave, v_mfma_f32_16x16x32).
//
// mlp256_kernel (gpt_kernels_c256.h) gives a wave 32 tokens: operand planes (128 registers) + output accumulators (128) force
// one wave per SIMD, and with one wave nothing fills the issue slots its own ds_reads, LDS-DMA pieces and GELU arithmetic
// take between the MFMAs (302 cycles per 6-MFMA chunk against 192 of matrix pipe).  Here a wave owns 16
// tokens: with the 16x16x32 MFMA shape the same operand planes and accumulators are 64 + 64 registers, a wave fits in 256
// registers and a SIMD holds two -- while one wave waits for fragments or issues its loads, the other's MFMAs run.  The
// workgroup is still 128 tokens (8 waves) on the same weight-stream protocol: steps of 8 fragment pairs through an 8-slot
// LDS ring filled by direct global->LDS loads 7 steps ahead, counted vmcnt, one raw s_barrier per step; every in-loop LDS
// access is inline asm (gpt_kernels_c256.h explains why).  Measured stand-in (tools/probe_w16.hip): 0.54 us per workgroup
// step against 0.68 for the one-wave shape under the same instruction mix.
//
// v_mfma_f32_16x16x32_f16 layout:  A 16 x 32: lane l supplies row (l & 15), k-slots 8 (l >> 4) .. + 8;  B 32 x 16: lane l
// supplies column (l & 15), the same k-slots;  D 16 x 16: lane l holds column (l & 15), rows 4 (l >> 4) .. + 4 (4 registers).
// "Swapped" use as everywhere in this library: weights are A (rows = output units), tokens are B / D columns, so
//   * lane (n = l & 15, kg = l >> 4) belongs to token n of the wave and holds features 64 kg .. 64 kg + 63 of its
//     normalised row: k-step ks (8 of them), slot e  <->  feature 64 kg + 8 ks + e          (xn: 8 x 2 planes x 4 registers)
//   * a hidden tile (32 units) is two row tiles ut = 0, 1; the lane's 4 + 4 pre-activations (units 16 ut + 4 kg + 0..3)
//     ARE, after GELU and the split, the 8 k-slots of its group in the B operand of c_proj: slot e < 4 <-> unit 4 kg + e,
//     e >= 4 <-> unit 16 + 4 kg + (e - 4) -- the permutation is baked into the packed c_proj weights, nothing moves
//   * output accumulators: 16 tiles of 16 features, lane holds features 16 j + 4 kg + 0..3 of its token       (64 registers)
// Software pipeline over the 32 hidden tiles exactly as in mlp256_kernel: iteration i = c_fc(i+1) || GELU(i) || c_proj(i-1).
// Results are not bit-identical to mlp256_kernel's (different summation order inside the MFMAs); same 1e-5 logit bar.
#pragma once
#include "../../mapf_gpt_amd/csrc/gpt_kernels_c256.h"

namespace mgpt {
namespace fastk {

template <class T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (std::is_same<T, F16T>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// the weight stream of mlp256w_kernel: kM256Steps steps of 8 fragment pairs, same schedule as pack_mlp256_kernel
//   steps 0, 1: c_fc(tile 0) fragments idx = 8 s + ms;  main step (it, q): ms < 4 c_fc(it + 1) idx = 4 q + ms,
//   ms >= 4 c_proj(it - 1) group 4 q + ms - 4;  last two steps: c_proj(31) group 8 s' + ms
//   c_fc fragment idx: k-step idx >> 1, row tile idx & 1;   c_proj group g: output tile g (one k-step: K = the tile's 32 units)
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp256w_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                           uint16_t *__restrict__ out, float scale1, float scale2)
{
    constexpr int C = 256, NT = 32;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (step, micro-step, lane)
    if (gid >= (int64_t)kM256Steps * 8 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 7), s = (int)(gid >> 9);
    const int i = lane & 15, kg = lane >> 4;
    int kind, t, idx;                                                     // kind 0: zeros, 1: c_fc (t, idx), 2: c_proj (t, group)
    if (s < 2) { kind = 1; t = 0; idx = 8 * s + ms; }
    else if (s < 2 + 4 * NT) {
        const int it = (s - 2) >> 2, q = (s - 2) & 3;
        if (ms < 4) { kind = it + 1 < NT ? 1 : 0; t = it + 1; idx = 4 * q + ms; }
        else { kind = it >= 1 ? 2 : 0; t = it - 1; idx = 4 * q + ms - 4; }
    } else { kind = 2; t = NT - 1; idx = 8 * (s - (2 + 4 * NT)) + ms; }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (kind == 1) {                                                  // A rows = hidden units of row tile ut, k-slots = features
            const int ks = idx >> 1, ut = idx & 1;
            v[e] = fc_w[(size_t)(32 * t + 16 * ut + i) * C + 64 * kg + 8 * ks + e] * scale1;
        } else if (kind == 2) {                                           // A rows = output features of tile idx, k-slots = hidden units
            const int u = 32 * t + (e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4));
            v[e] = pj_w[(size_t)(16 * idx + i) * (4 * C) + u] * scale2;
        } else v[e] = 0.f;
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)s * 8 + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// X (tools/probe_mlp256w.hip only; the library instantiates X = 0): 1 no table gathers, 2 no pinned MFMA / VALU order,
// 4 no GELU arithmetic at all, 8 no row loads / stores -- results are wrong unless X == 0
template <class T, int NP, int X = 0>
__global__ __launch_bounds__(512, 2) void mlp256w_kernel(float *__restrict__ x, const float *__restrict__ gain,
                                                         const uint16_t *__restrict__ wstream, float inv1, float inv2,
                                                         const float2 *__restrict__ gelu_lut)
{
    constexpr int C = 256, KS = 8, NT = 32, NJ = 16, NW = 8;
    constexpr int MS = 8;                                  // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;                   // bytes per step
    constexpr int NSLOT = 8;
    constexpr int LUT_BYTES = kGeluLutN * 8;               // 24 KiB
    constexpr int PW = MS * NP / NW;                       // direct-to-LDS loads per wave per step
    constexpr int NSTEP = kM256Steps;
    static_assert(PW >= 1 && (LUT_BYTES / 1024) % NW == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSLOT][STEP] ring, then the GELU table
    unsigned char *lut = smem + NSLOT * STEP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(wave * PW) * 1024;   // wave-uniform
    int gstep = 0;

    {   // GELU table -> LDS (24 pieces of 1 KiB, 3 per wave); older than every ring piece, so the first counted wait covers it
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut);
#pragma unroll
        for (int i = 0; i < LUT_BYTES / 1024 / NW; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(wave + NW * i) * 1024 + lane16),
                                             (lds_void_t *)(lut + (size_t)(wave + NW * i) * 1024), 16, 0, 0);
    }
    auto issue = [&](int src_step, int slot) {             // this wave moves pieces wave*PW .. +PW of a step
        const unsigned char *src = wbase + (size_t)src_step * STEP;
        unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(wave * PW) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++) dma_piece(src + lane16, dst, std::integral_constant<int, 0>{}, i);
    };
#pragma unroll
    for (int s_ = 0; s_ < NSLOT - 1; s_++) issue(s_, s_);

    // ring protocol (as mlp256_kernel): after the barrier of step s the steps up to s+1 have landed for every wave and the
    // slot of step s-1 is free; it is refilled with step s + NSLOT - 1
    auto sync = [&](int s_) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (s_ + NSLOT - 1 < NSTEP) issue(s_ + NSLOT - 1, (gstep + NSLOT - 1) % NSLOT);
        gstep++;
    };

    // ---- this lane's quarter (64 features) of its token's row; LayerNorm over the 4 lanes of the token (two-pass) ----
    float *xrow = x + ((int64_t)blockIdx.x * 128 + wave * 16 + n) * C;
    u32x4 xn[KS][2];                                       // B operand of c_fc: [k-step][plane]
    {
        float xv[64];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const f32x4 v = (X & 8) ? (f32x4){0.1f * q, 0.2f, -0.3f, 0.01f * lane} : *reinterpret_cast<const f32x4 *>(xrow + 64 * kg + 4 * q);
            xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float qv = 0.f;
#pragma unroll
        for (int e = 0; e < 64; e++) { const float d = xv[e] - mean; qv += d * d; }
        qv += __shfl_xor(qv, 16);
        qv += __shfl_xor(qv, 32);
        const float rstd = rsqrtf(qv / (float)C + 1e-5f);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gain + 64 * kg + 8 * ks);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(gain + 64 * kg + 8 * ks + 4);
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = (xv[8 * ks + e] - mean) * rstd * ga[e];
                v1[e] = (xv[8 * ks + 4 + e] - mean) * rstd * gb[e];
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
            xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
        }
    }
    f32x4 acc[NJ];                                         // output accumulators: features 16 j + 4 kg + 0..3 of the lane's token
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- weight fragments: two register sets of 2 pairs; chunk c uses pairs c and 4 + c, requested one chunk earlier ----
    u32x4 wb[2][2][2];                                     // [set][0: pair c, 1: pair 4+c][plane]
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        else dst[1] = dst[0];
    };
    unsigned cur_addr = 0, nxt_addr = 0;
    auto step_begin = [&](int s_) {
        sync(s_);
        cur_addr = lds0 + (unsigned)((gstep - 1) % NSLOT) * STEP;
        nxt_addr = lds0 + (unsigned)(gstep % NSLOT) * STEP;
    };
    // GATHERS: chunk 0 of a mixed step issues two table gathers after its requests (the two youngest LDS operations at the
    // top of chunk 1: they may keep flying, chunk 2 consumes them)
    auto chunk_begin = [&](auto c_c, bool has_next, bool gathers = false) {
        constexpr int c = decltype(c_c)::value;
        if (gathers && c == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, (c + 1) % 4>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, 4 + (c + 1) % 4>{}, wb[(c + 1) & 1][1]); }
        else if (has_next) { lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]); lds_pair(nxt_addr, std::integral_constant<int, 4>{}, wb[0][1]); }
        __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this chunk's MFMAs
    };
    // one split product step on two independent accumulators, passes interleaved
    auto mma2 = [&](const u32x4 (&wa)[2], const u32x4 (&ba)[2], f32x4 &ca, const u32x4 (&wc)[2], const u32x4 (&bb)[2], f32x4 &cb) {
        if (NP == 2) {
            ca = mfma16<T>(wa[1], ba[0], ca); cb = mfma16<T>(wc[1], bb[0], cb);
            ca = mfma16<T>(wa[0], ba[1], ca); cb = mfma16<T>(wc[0], bb[1], cb);
        }
        ca = mfma16<T>(wa[0], ba[0], ca); cb = mfma16<T>(wc[0], bb[0], cb);
    };
    auto pin = [&](auto n_valu_c) {                        // every MFMA is followed by its share of the chunk's VALU work
        constexpr int n_valu = decltype(n_valu_c)::value;
        if (!(X & 2)) {
#pragma unroll
            for (int k = 0; k < (NP == 2 ? 6 : 2); k++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (n_valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, n_valu, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using VN = std::integral_constant<int, (NP == 2 ? 3 : 8)>;
    using V0 = std::integral_constant<int, 0>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    const float lut_scale = inv1 * kGeluLutScale;
    const unsigned lut_addr = (unsigned)(size_t)lut;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 hA[2] = {zero4, zero4}, hB[2] = {zero4, zero4};  // pre-activations [row tile]: one set accumulates c_fc(i+1), the other feeds GELU(i)

    // ---- steps 0, 1: c_fc of hidden tile 0; pair c -> hA[c & 1], pair 4 + c -> hB[c & 1] (two partial chains per row tile) ----
    auto step_fc0 = [&](int s_, bool first) {
        step_begin(s_);
        if (first) { lds_pair(cur_addr, I0{}, wb[0][0]); lds_pair(cur_addr, std::integral_constant<int, 4>{}, wb[0][1]); }
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, true);
            // idx = 8 s + c and 8 s + 4 + c  ->  k-steps 4 s + (c >> 1) and 4 s + 2 + (c >> 1), row tile c & 1
            mma2(wb[c & 1][0], xn[4 * (first ? 0 : 1) + (c >> 1)], hA[c & 1], wb[c & 1][1], xn[4 * (first ? 0 : 1) + 2 + (c >> 1)], hB[c & 1]);
            pin(V0{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    step_fc0(0, true);
    step_fc0(1, false);
#pragma unroll
    for (int ut = 0; ut < 2; ut++) { hA[ut] += hB[ut]; hB[ut] = zero4; }

    u32x4 hfA[2], hfB[2];                                  // hidden planes [plane]: B operand of c_proj (one k-step = the tile's 32 units)
#pragma unroll
    for (int pl = 0; pl < 2; pl++) { hfA[pl] = (u32x4){0u, 0u, 0u, 0u}; hfB[pl] = (u32x4){0u, 0u, 0u, 0u}; }

    // one step of a pipeline iteration: pairs 0-3 = c_fc(i+1) idx 4q + c, pairs 4-7 = c_proj(i-1) output tile 4q + c;
    // GELU of the two pre-activations that make word q of the hidden planes (row tile q >> 1, registers 2 (q & 1), + 1):
    // chunk 0 forms the table addresses and issues the two gathers, chunk 2 interpolates, multiplies and splits
    auto step_main = [&](int s_, auto q_c, const f32x4 (&hsrc)[2], f32x4 (&hdst)[2], const u32x4 (&hfi)[2], u32x4 (&hfo)[2]) {
        constexpr int q = decltype(q_c)::value;
        step_begin(s_);
        float gvv[2], gfr[2];
        f32x2 gtab[2];
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, true, !(X & 5));
            if (c == 2 && !(X & 4)) asm volatile("" : "+v"(gtab[0]), "+v"(gtab[1]));          // gathers landed (lgkmcnt(0) above)
            if (c == 0 && !(X & 4)) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float hv = hsrc[q >> 1][2 * (q & 1) + e];
                    gvv[e] = hv * inv1;
                    const float t = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                    gfr[e] = __builtin_amdgcn_fractf(t);
                    if (X & 1) gtab[e] = (f32x2){1.f, 0.f};
                    else asm volatile("ds_read_b64 %0, %1" : "=v"(gtab[e]) : "v"(lut_addr + (unsigned)t * 8u) : "memory");
                }
            }
            if (c == 2 && !(X & 4)) {
                const float g0 = gvv[0] * fmaf(gfr[0], gtab[0][1], gtab[0][0]), g1 = gvv[1] * fmaf(gfr[1], gtab[1][1], gtab[1][0]);
                unsigned hi, lo;
                split2p<T, NP>(g0, g1, hi, lo);
                hfo[0][q] = hi;
                hfo[1][q] = lo;
            }
            constexpr int idx = 4 * q + c;                 // c_fc: k-step idx >> 1, row tile idx & 1;  c_proj: output tile idx
            mma2(wb[c & 1][0], xn[idx >> 1], hdst[idx & 1], wb[c & 1][1], hfi, acc[idx]);
            pin(VN{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    auto iteration = [&](int sbase, const f32x4 (&hsrc)[2], f32x4 (&hdst)[2], const u32x4 (&hfi)[2], u32x4 (&hfo)[2]) {
        step_main(sbase + 0, I0{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 1, I1{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 2, I2{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 3, I3{}, hsrc, hdst, hfi, hfo);
    };

#pragma unroll 1
    for (int i = 0; i < NT; i += 2) {
        iteration(2 + 4 * i, hA, hB, hfB, hfA);            // even tile: GELU(hA) -> hfA, c_fc(i+1) -> hB, c_proj(hfB = tile i-1)
        hA[0] = zero4; hA[1] = zero4;
        iteration(2 + 4 * (i + 1), hB, hA, hfA, hfB);      // odd tile
        hB[0] = zero4; hB[1] = zero4;
    }
    // ---- last two steps: c_proj of hidden tile 31 (planes in hfB): pair ms of step s' = output tile 8 s' + ms ----
    auto step_pj31 = [&](int s_, bool last) {
        step_begin(s_);
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, !last);
            mma2(wb[c & 1][0], hfB, acc[8 * (last ? 1 : 0) + c], wb[c & 1][1], hfB, acc[8 * (last ? 1 : 0) + 4 + c]);
            pin(V0{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    step_pj31(2 + 4 * NT, false);
    step_pj31(2 + 4 * NT + 1, true);

    // ---- residual add and store: the 4 lanes of a token cover 64 contiguous bytes per output tile ----
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(xrow + 16 * j + 4 * kg);
        f32x4 cur = (X & 8) ? (f32x4){0.f, 0.f, 0.f, 0.f} : *dst;
#pragma unroll
        for (int e = 0; e < 4; e++) cur[e] += acc[j][e] * inv2;
        if (!(X & 8) || cur[0] == 12345.678f) *dst = cur;
    }
}

}  // namespace fastk
}  // namespace mgpt
