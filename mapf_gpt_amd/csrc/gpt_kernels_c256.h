// gpt_kernels_c256.h -- 16-bit-MFMA kernels for n_embd = 256 (the MAPF-GPT-6M shape), gfx950.
//
// What round 1 got wrong for this shape, measured in round 2 (tools/bench_probes/probe_interleave.hip,
// profiles/r02_probe_interleave.txt): when non-MFMA instructions are INTERLEAVED one v_mfma_f32_32x32x16 at a time, a
// SIMD hides ~6 plain VALU instructions (or ~3 transcendentals, or 1-2 ds_read_b128 whose latency is covered) behind
// every MFMA at no cost in MFMA rate; only when they are clumped before/after a run of MFMAs does their issue time add
// (which is what round 1's probes measured, and what its kernels did: c_fc MFMAs -> GELU -> c_proj MFMAs with all eight
// waves in lock-step between three barriers per hidden tile, 31 % MFMA issue).  The kernels here are therefore written as
// software pipelines in which every MFMA carries its share of the VALU / LDS / DMA work of OTHER pipeline stages.
#pragma once
#include "gpt_kernels_fast.h"

namespace mgpt {
namespace fastk {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// the same for one pair of values -> one hi word and one lo word.  Split-fp16 path: 3 VALU for the pair -- one packed conversion
// for hi, then lo = RNE fp16(v - hi) by v_fma_mixlo / mixhi_f16 (fma(hi as fp16, -1.0, v) rounded once to fp16: v - hi is exact in
// fp32, so this is the same value the conversion chain cvt_f32_f16 / sub / cvt_pk gave, in half the instructions)
template <class T, int NP>
__device__ __forceinline__ void split2p(float v0, float v1, unsigned &hi, unsigned &lo)
{
    asm("" : "+v"(v0));
    asm("" : "+v"(v1));
    if constexpr (NP == 2 && std::is_same<T, F16T>::value) {
        const f16x2 a = __builtin_convertvector((f32x2){v0, v1}, f16x2);
        hi = __builtin_bit_cast(unsigned, a);
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v1));
        lo = l;
    } else if constexpr (NP == 1 && std::is_same<T, BF16T>::value) {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2));
        lo = 0u;
    } else {
        const uint16_t a0 = T::cvt(v0), a1 = T::cvt(v1);
        const uint16_t b0 = (NP == 2) ? T::cvt(v0 - T::back(a0)) : (uint16_t)0, b1 = (NP == 2) ? T::cvt(v1 - T::back(a1)) : (uint16_t)0;
        hi = (unsigned)a0 | ((unsigned)a1 << 16);
        lo = (unsigned)b0 | ((unsigned)b1 << 16);
    }
}

// split of 4 values into hi / lo fp16 words (same results as split4: hi = RNE fp16(v), lo = RNE fp16(v - hi))
template <class T, int NP>
__device__ __forceinline__ void split4p(const float v[4], u32x2 &hi, u32x2 &lo)
{
    if constexpr (NP == 2 && std::is_same<T, F16T>::value) {
        unsigned h0, l0, h1, l1;
        split2p<T, NP>(v[0], v[1], h0, l0);
        split2p<T, NP>(v[2], v[3], h1, l1);
        hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
    } else {
        split4<T, NP>(v, hi, lo);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused MLP block for C = 256:  x <- x + c_proj(GELU(c_fc(LayerNorm(x))))   (model.py:84-89, 103)
//
// Workgroup = 4 waves = 128 tokens, ONE wave per SIMD with the whole 512-register file: a wave owns 32 tokens and keeps
// the normalised rows as MFMA operand planes (128 registers) and the 32 x 256 output accumulators (128 registers) for the
// whole kernel, "swapped" C/D layout as in mlp_fused_kernel (lane = token, registers = features).
// Software pipeline over the 32 hidden tiles (32 hidden units each); iteration i runs, interleaved MFMA by MFMA,
//     c_fc   of tile i+1   (48 MFMAs: 16 k-steps x 3 split passes)      -> pre-activations of the NEXT tile
//     GELU + fp16 split of tile i (~90 VALU per 24 MFMAs, riding in the MFMA shadows)
//     c_proj of tile i-1   (48 MFMAs: 8 output tiles x 2 k-steps x 3)   <- hidden planes of the PREVIOUS tile
// so the matrix pipe never waits for the activation function and no barrier separates the stages.
// Weights arrive as ONE stream in consumption order (pack_mlp256_kernel): "steps" of 8 fragment pairs (16 KiB in the
// split mode) = 24 MFMAs per wave; an 8-slot LDS ring is filled by direct global->LDS loads 7 steps ahead (counted
// vmcnt), one raw s_barrier per step hands a slot over; the first fragments of the next step are read before that barrier.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kM256Steps = 2 + 4 * 32 + 2;     // c_fc(0) | 32 x 4 mixed steps | c_proj(31)

// (GELU table: kGeluLut* in gpt_kernels_fast.h)

// one 1-KiB direct global->LDS piece at byte offset 1024 * i from (src, dst): the instruction's immediate offset applies to
// the global and to the LDS address alike, so the PW pieces of a step share one 64-bit address and one M0 value
__device__ __forceinline__ void dma_piece(const unsigned char *src, unsigned char *dst, std::integral_constant<int, 0>, int i)
{
    switch (i) {                                            // i is a compile-time constant after unrolling
    case 0: __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0); break;
    case 1: __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 1024, 0); break;
    case 2: __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 2048, 0); break;
    default: __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 3072, 0); break;
    }
}

template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp256_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                          uint16_t *__restrict__ out, float scale1, float scale2)
{
    constexpr int C = 256, NT = 32;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (step, micro-step, lane)
    if (gid >= (int64_t)kM256Steps * 8 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 7), s = (int)(gid >> 9);
    const int i = lane & 31, h = lane >> 5;
    int kind, t, idx;                                                     // kind 0: zeros, 1: c_fc (t, k-step), 2: c_proj (t, group)
    if (s < 2) { kind = 1; t = 0; idx = 8 * s + ms; }
    else if (s < 2 + 4 * NT) {
        const int it = (s - 2) >> 2, q = (s - 2) & 3;
        if (ms < 4) { kind = it + 1 < NT ? 1 : 0; t = it + 1; idx = 4 * q + ms; }
        else { kind = it >= 1 ? 2 : 0; t = it - 1; idx = 4 * q + ms - 4; }
    } else { kind = 2; t = NT - 1; idx = 8 * (s - (2 + 4 * NT)) + ms; }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (kind == 1) {                                                  // A rows = hidden units, k-slots = features (tau-permuted)
            const int g = 8 * (idx & 1) + e;
            const int feat = 32 * (idx >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
            v[e] = fc_w[(size_t)(32 * t + i) * C + feat] * scale1;
        } else if (kind == 2) {                                           // A rows = output features of tile j, k-slots = hidden units
            const int kk = idx >> 3, j = idx & 7, g = 8 * kk + e;
            const int u = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
            v[e] = pj_w[(size_t)(32 * j + i) * (4 * C) + u] * scale2;
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

// ABL (tools/bench_probes/probe_mlp256.hip only; the library instantiates ABL = 0): 1 no weight DMA in the loop, 2 no GELU table gather,
// 4 no barrier, 8 no MFMAs, 16 no fragment reads in the loop -- results are wrong unless ABL == 0.  32: wave 0 of every
// block leaves (s_memtime, s_memrealtime) at kernel entry / first ring step / after the last step / exit in stamps[block][8].
//
// Register plan (one wave per SIMD, 512 registers): operand planes xn 128 + hidden planes 2 x 16 + GELU temporaries in the
// arch VGPRs; output accumulators 128 + pre-activation accumulators 2 x 16 + weight fragments 32 in the accumulator file.
// The fragments are read by inline-asm ds_read_b128 with accumulator-file destinations (MFMA takes A/B operands from
// there directly): left to itself hipcc keeps them in arch VGPRs, runs out, and spills INSIDE the loop -- and every scratch
// reload is a VMEM load whose s_waitcnt vmcnt(0) drains the whole weight ring (measured: 3.3 -> 5.0 ms per launch); a
// compiler-visible LDS read has the same effect (hipcc orders it behind every LDS-DMA in flight), hence asm for the
// table gathers as well.  Asm reads are invisible to hipcc's lgkmcnt bookkeeping: every chunk opens with an explicit
// s_waitcnt lgkmcnt + sched_barrier(0) (no MFMA may be hoisted above the wait: cdna guide 5.4 rule 18, 5.7).
// (A persistent variant -- one workgroup per CU walking the blocks with the ring running across block seams, residual rows
//  prefetched into the dead operand-plane registers -- was built and measured: no gain, and hipcc hoists the 128 registers
//  of LayerNorm gains out of the block loop and spills; one block per workgroup it is.)
template <class T, int NP, int ABL = 0>
__global__ __launch_bounds__(256, 1) void mlp256_kernel(float *__restrict__ x, const float *__restrict__ gain,
                                                        const uint16_t *__restrict__ wstream, float inv1, float inv2,
                                                        const float2 *__restrict__ gelu_lut, unsigned long long *stamps = nullptr)
{
    constexpr int C = 256, CT = 8, KS = 16, NT = 32;
    unsigned long long tstamp[8];
    auto stamp = [&](int i) { if constexpr ((ABL & 32) != 0) { tstamp[2 * i] = __builtin_readcyclecounter(); tstamp[2 * i + 1] = wall_clock64(); } };
    stamp(0);
    constexpr int MS = 8;                                  // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;                   // bytes per step
    constexpr int NSLOT = 8;
    constexpr int LUT_BYTES = kGeluLutN * 8;            // 24 KiB
    constexpr int PW = MS * NP / 4;                        // direct-to-LDS loads per wave per step
    constexpr int NSTEP = kM256Steps;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSLOT][STEP] ring, then the GELU table
    unsigned char *lut = smem + NSLOT * STEP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;                 // LDS byte address of this lane's 16 bytes in fragment 0 of slot 0
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(wave * PW) * 1024;   // wave-uniform
    int gstep = 0;                                         // steps done so far (ring position of the NEXT step after sync)

    {   // GELU table -> LDS (48 pieces of 1 KiB, 12 per wave); older than every ring piece, so the first counted wait covers it
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut) + (size_t)wave * (LUT_BYTES / 4);
#pragma unroll
        for (int i = 0; i < LUT_BYTES / 4096; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + i * 1024 + lane16), (lds_void_t *)(lut + wave * (LUT_BYTES / 4) + i * 1024), 16, 0, 0);
    }
    auto issue = [&](int src_step, int slot) {             // this wave moves pieces wave*PW .. +PW of a step
        const unsigned char *src = wbase + (size_t)src_step * STEP;                                              // scalar address math
        unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(wave * PW) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++)                         // one address pair and one M0 value per step: the pieces of a wave are
            dma_piece(src + lane16, dst, std::integral_constant<int, 0>{}, i);   // contiguous on both sides (immediate offsets)
    };
#pragma unroll
    for (int s_ = 0; s_ < NSLOT - 1; s_++) issue(s_, s_);

    // ---- ring protocol: sync(s) at the top of step s ----
    //   after the barrier the steps up to s+1 have landed for every wave (each wave counted its own pieces) and every wave
    //   has finished reading the slot of step s-1, which is refilled with step s+NSLOT-1
    auto sync = [&](int s_) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NSLOT - 3)) : "memory");
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        if (!(ABL & 1) && s_ + NSLOT - 1 < NSTEP) issue(s_ + NSLOT - 1, (gstep + NSLOT - 1) % NSLOT);
        gstep++;
    };

    // ---- 32 x 256 row block in swapped layout, LayerNorm in-lane (two-pass, model.py:19-20) ----
    float *xrow = x + ((int64_t)blockIdx.x * 128 + wave * 32 + r) * C;    // this lane's token
    f32x16 acc[CT];                                        // x now, output accumulators later
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow + 32 * j + 8 * gq + 4 * h);
            acc[j][4 * gq] = v[0]; acc[j][4 * gq + 1] = v[1]; acc[j][4 * gq + 2] = v[2]; acc[j][4 * gq + 3] = v[3];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float qv = 0.f;
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) { const float d = acc[j][g] - mean; qv += d * d; }
    qv += __shfl_xor(qv, 32);
    const float rstd = rsqrtf(qv / (float)C + 1e-5f);
    u32x4 xn[KS][2];                                       // B operand of c_fc: k-step ks <-> registers 8 (ks & 1) .. + 8 of tile ks >> 1
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int j = ks >> 1, g0 = 8 * (ks & 1);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gain + 32 * j + 8 * (g0 >> 2) + 4 * h);
        const f32x4 gb = *reinterpret_cast<const f32x4 *>(gain + 32 * j + 8 * (g0 >> 2) + 8 + 4 * h);
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v0[e] = (acc[j][g0 + e] - mean) * rstd * ga[e];
            v1[e] = (acc[j][g0 + 4 + e] - mean) * rstd * gb[e];
        }
        u32x2 h0, l0, h1, l1;
        split4<T, NP>(v0, h0, l0);
        split4<T, NP>(v1, h1, l1);
        xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
        xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
    }
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) acc[j][g] = 0.f;

    // ---- weight fragments (accumulator file): two small register sets of 2 pairs each.  Every step uses pairs c and 4+c in
    //      chunk c; chunk c requests the pairs of chunk c+1 (chunk 3: pairs 0 and 4 of the NEXT step, whose slot has landed),
    //      so a fragment is requested one chunk = 6 MFMAs (>= 192 cycles, LDS latency is ~130) before its first MFMA and
    //      both register files keep > 50 registers of slack (with 64 fragment registers hipcc started to copy freshly
    //      requested, not yet landed fragments between the files). ----
    u32x4 wb[2][2][2];                                     // [set][0: pair c, 1: pair 4+c][plane]
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {  // fragment pair ms of the slot at LDS address slot_addr
        constexpr int ms = decltype(ms_c)::value;
        if ((ABL & 16) && gstep > 2) return;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        else dst[1] = dst[0];
    };
    unsigned cur_addr = 0, nxt_addr = 0;                   // slots of this step and of the next one
    auto step_begin = [&](int s_) {
        sync(s_);
        cur_addr = lds0 + (unsigned)((gstep - 1) % NSLOT) * STEP;
        nxt_addr = lds0 + (unsigned)(gstep % NSLOT) * STEP;
    };
    // chunk prologue: this chunk's pairs (requested one chunk ago) must have landed; then request the next chunk's.
    // GATHERS: chunk 0 of a mixed step issues four table gathers after its requests; they may still be in flight in chunk 1
    // (consumed in chunks 2 and 3).
    auto chunk_begin = [&](auto c_c, bool has_next, bool gathers = false) {
        constexpr int c = decltype(c_c)::value;
        if (gathers && c == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, (c + 1) % 4>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, 4 + (c + 1) % 4>{}, wb[(c + 1) & 1][1]); }
        else if (has_next) { lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]); lds_pair(nxt_addr, std::integral_constant<int, 4>{}, wb[0][1]); }
        __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this chunk's MFMAs (else hipcc sinks them
    };                                                     // behind the MFMAs, reuses the registers and the next wait eats the LDS latency)
    // one split product step on two independent accumulators, passes interleaved (no back-to-back dependent MFMAs)
    auto mma2 = [&](const u32x4 (&wa)[2], const u32x4 (&ba)[2], f32x16 &ca, const u32x4 (&wb)[2], const u32x4 (&bb)[2], f32x16 &cb) {
        if (ABL & 8) { asm volatile("" :: "a"(wa[0]), "a"(wa[1]), "a"(wb[0]), "a"(wb[1]), "v"(ba[0]), "v"(bb[0]), "v"(ba[1]), "v"(bb[1])); return; }
        if (NP == 2) {
            ca = T::mfma(wa[1], ba[0], ca); cb = T::mfma(wb[1], bb[0], cb);
            ca = T::mfma(wa[0], ba[1], ca); cb = T::mfma(wb[0], bb[1], cb);
        }
        ca = T::mfma(wa[0], ba[0], ca); cb = T::mfma(wb[0], bb[0], cb);
    };
    // inside a chunk every MFMA is followed by its share of the chunk's VALU work
    auto pin = [&](auto n_valu_c) {
        constexpr int n_valu = decltype(n_valu_c)::value;
#pragma unroll
        for (int n = 0; n < (NP == 2 ? 6 : 2); n++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (n_valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, n_valu, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using VN = std::integral_constant<int, (NP == 2 ? 4 : 12)>;
    using V0 = std::integral_constant<int, 0>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    const float lut_scale = inv1 * kGeluLutScale;
    const unsigned lut_addr = (unsigned)(size_t)lut;

    f32x16 hA, hB;                                         // pre-activations: one accumulates c_fc(i+1) while the other feeds GELU(i)
#pragma unroll
    for (int g = 0; g < 16; g++) { hA[g] = 0.f; hB[g] = 0.f; }

    // ---- steps 0, 1: c_fc of hidden tile 0 (two chains, summed) ----
    auto step_fc0 = [&](int s_, bool first) {
        step_begin(s_);
        if (first) { lds_pair(cur_addr, I0{}, wb[0][0]); lds_pair(cur_addr, std::integral_constant<int, 4>{}, wb[0][1]); }   // the very first pairs
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, true);
            mma2(wb[c & 1][0], xn[8 * (first ? 0 : 1) + c], hA, wb[c & 1][1], xn[8 * (first ? 0 : 1) + 4 + c], hB);
            pin(V0{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    stamp(1);
    step_fc0(0, true);
    step_fc0(1, false);
#pragma unroll
    for (int g = 0; g < 16; g++) { hA[g] += hB[g]; hB[g] = 0.f; }

    u32x4 hfA[2][2], hfB[2][2];                            // hidden planes [k-step kk][plane]: B operand of c_proj
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int e = 0; e < 4; e++) { hfA[kk][pl][e] = 0u; hfB[kk][pl][e] = 0u; }

    // one step of a pipeline iteration: fragment pairs 0-3 = c_fc(i+1) k-steps 4q.., 4-7 = c_proj(i-1) groups 4q..
    // chunk c: c_fc k-step 4q+c and c_proj group 4q+c; GELU of pre-activations 4q .. 4q+3 (hidden units tau(4q + e, h) of
    // tile i) by table: chunk 0 forms the four table addresses and issues the gathers, chunks 2 and 3 interpolate,
    // multiply and split one pair each (the gathers have >= 2 chunks = 12 MFMAs to land)
    auto step_main = [&](int s_, auto q_c, const f32x16 &hsrc, f32x16 &hdst, const u32x4 (&hfi)[2][2], u32x4 (&hfo)[2][2]) {
        constexpr int q = decltype(q_c)::value;
        step_begin(s_);
        float gvv[4], gfr[4];
        f32x2 gtab[4];                                     // (Phi, dPhi) pairs, gathered by asm ds_read_b64: a compiler-visible LDS read
        auto chunk = [&](auto c_c) {                       // would make hipcc drain the weight ring (vmcnt(0)) before it
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, true, true);
            if (c == 2) asm volatile("" : "+v"(gtab[0]), "+v"(gtab[1]), "+v"(gtab[2]), "+v"(gtab[3]));   // gathers landed (lgkmcnt(0) above)
            if (c == 0) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float hv = hsrc[4 * q + e];
                    gvv[e] = hv * inv1;
                    const float t = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                    gfr[e] = __builtin_amdgcn_fractf(t);
                    const unsigned idx = (unsigned)t;
                    if (ABL & 2) gtab[e] = (f32x2){1.f, 0.f};
                    else asm volatile("ds_read_b64 %0, %1" : "=v"(gtab[e]) : "v"(lut_addr + idx * 8u) : "memory");
                }
            }
            if (c >= 2) {
                constexpr int e0 = 2 * (c >= 2 ? c - 2 : 0);
                const float g0 = gvv[e0] * fmaf(gfr[e0], gtab[e0][1], gtab[e0][0]), g1 = gvv[e0 + 1] * fmaf(gfr[e0 + 1], gtab[e0 + 1][1], gtab[e0 + 1][0]);
                unsigned hi, lo;
                split2p<T, NP>(g0, g1, hi, lo);
                hfo[q >> 1][0][2 * (q & 1) + (c >= 2 ? c - 2 : 0)] = hi;
                hfo[q >> 1][1][2 * (q & 1) + (c >= 2 ? c - 2 : 0)] = lo;
            }
            constexpr int g = 4 * q + c;                   // c_proj group: k-step g >> 3 of the slice, output tile g & 7
            mma2(wb[c & 1][0], xn[4 * q + c], hdst, wb[c & 1][1], hfi[g >> 3], acc[g & 7]);
            pin(VN{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    // one pipeline iteration = 4 steps: c_fc(i+1) -> hdst, GELU(hsrc) -> hfo, c_proj(hfi) -> acc
    auto iteration = [&](int sbase, const f32x16 &hsrc, f32x16 &hdst, const u32x4 (&hfi)[2][2], u32x4 (&hfo)[2][2]) {
        step_main(sbase + 0, I0{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 1, I1{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 2, I2{}, hsrc, hdst, hfi, hfo);
        step_main(sbase + 3, I3{}, hsrc, hdst, hfi, hfo);
    };

#pragma unroll 1
    for (int i = 0; i < NT; i += 2) {
        iteration(2 + 4 * i, hA, hB, hfB, hfA);            // even tile: GELU(hA) -> hfA, c_fc(i+1) -> hB, c_proj(hfB = tile i-1)
#pragma unroll
        for (int g = 0; g < 16; g++) hA[g] = 0.f;
        iteration(2 + 4 * (i + 1), hB, hA, hfA, hfB);      // odd tile
#pragma unroll
        for (int g = 0; g < 16; g++) hB[g] = 0.f;
    }
    // ---- last two steps: c_proj of hidden tile 31 (its planes are in hfB): pair ms = output tile ms, k-step = step ----
    auto step_pj31 = [&](int s_, bool last) {
        step_begin(s_);
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            chunk_begin(c_c, !last);
            mma2(wb[c & 1][0], hfB[last ? 1 : 0], acc[c], wb[c & 1][1], hfB[last ? 1 : 0], acc[4 + c]);
            pin(V0{});
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
    };
    step_pj31(2 + 4 * NT, false);
    step_pj31(2 + 4 * NT + 1, true);
    stamp(2);

    // ---- residual add and store ----
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(xrow + 32 * j + 8 * gq + 4 * h);
            f32x4 cur = *dst;
#pragma unroll
            for (int e = 0; e < 4; e++) cur[e] += acc[j][4 * gq + e] * inv2;
            *dst = cur;
        }
    if constexpr ((ABL & 32) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        if (tid == 0)
#pragma unroll
            for (int i = 0; i < 8; i++) stamps[(size_t)blockIdx.x * 8 + i] = tstamp[i];
    }
}

// (Fusing the attention out-projection in front of this kernel -- 16 more stream steps on the y operand planes, new residual
//  rows stored for the final add -- was built and measured in round 2: 84.3 ms per step for the fused kernel against
//  67.6 + 13.8 ms for this kernel plus the packed-GEMM out-projection; kept separate.)
// (A two-waves-per-SIMD variant -- every 32-token tile shared by a pair of 256-register waves, c_fc split over K and c_proj
//  over N, partial sums and hidden planes handed over through LDS on the ring barrier -- was built in round 2 and measured at
//  2.82 ms per 4096-row launch against 2.57 ms for this kernel: the second wave per SIMD does not pay for the hand-offs and
//  the 6 % of dummy steps its uniform pipeline needs.  Both sit at ~62 % of the MFMA rate this chip sustains (2.07 PFLOP/s).)

// ---------------------------------------------------------------------------------------------------------------------
// Fused attention front half for C = 256, head size 32:  y = attention(LayerNorm(x))   (model.py:46-68, 102), one row
// (256 tokens) per workgroup, 8 waves x 32 tokens, two waves per SIMD (256 registers each).  q, k, v never touch HBM;
// the only traffic is x in (1 KiB per token) and the y operand planes out (1 KiB per token, packed-fragment layout, which
// the out-projection consumes as they are).  Replaces ln_pack_kernel + 2 x gemm_pk_kernel + attn16_kernel for this shape.
//   per head (8 of them), per wave:
//     steps 0-3  q and k tiles together (both read the same token planes): chunk = k-step ks of both, 6 MFMAs on two
//                independent accumulators; k -> LDS planes sK[plane][key][d], q stays in registers as the B operand of S
//     steps 4-5  v tile ("natural": lane = d) on two chains -> transposed LDS planes sV[plane][d][key]
//     barrier    (every wave contributed its 32 keys)
//     attention  8 key tiles: S^T = K Q^T (6 MFMAs), online softmax in-lane (a lane owns one query), P split in-lane
//                into the B operand of O^T = V^T P^T (6 MFMAs); K / V^T fragments by asm ds_read_b128
//   the register -> row map tau(g, h) = (g & 3) + 8 (g >> 2) + 4 h is the same for "d of a token" (q, k tiles), "token of
//   a d" (v tile) and "key of a query" (S^T tile): register octets are MFMA k-slot groups everywhere, nothing is transposed.
// c_attn.weight arrives as ONE stream in consumption order (pack_attn256_kernel): 6 steps of 8 fragment pairs per head,
// through a 5-slot LDS ring filled by direct global->LDS loads 4 steps ahead (the next head's first steps land during
// the attention), counted vmcnt, one raw s_barrier per step.  Every LDS access inside the head loop is inline asm: a
// compiler-visible LDS access makes hipcc drain the LDS-DMA ring (s_waitcnt vmcnt(0)) in front of it.
// LAST (last layer, model.py:186): all keys and values, but q / attention only for the wave that owns token 255, whose
// output row goes to row b of the compact matrix y (packed-fragment layout over rows instead of tokens).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kA256StepsPerHead = 6;
// The two waves of a SIMD (w and w + 4) leave the head's k / v barrier together and would run the attention phase in phase --
// both in the S / PV MFMAs, then both in the softmax arithmetic, each at half speed.  Waves 4-7 start it STG x 64 cycles late
// (s_sleep), half a key tile, so that one wave's softmax runs under the other's MFMAs (tools/bench_probes/probe_attn256.hip).
constexpr int kA256Stagger = 0;

template <class T, int NP>
__global__ __launch_bounds__(256) void pack_attn256_kernel(const float *__restrict__ w, const float *__restrict__ gain,
                                                           uint16_t *__restrict__ out, float scale)
{
    constexpr int C = 256, NH = 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (global step, pair, lane)
    if (gid >= (int64_t)NH * kA256StepsPerHead * 8 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 7), G = (int)(gid >> 9);
    const int head = G / kA256StepsPerHead, st = G - head * kA256StepsPerHead;
    const int i = lane & 31, h = lane >> 5;
    int which, ks;
    if (st < 4) { which = ms & 1; ks = 4 * st + (ms >> 1); }              // q and k interleaved per k-step
    else { which = 2; ks = 8 * (st - 4) + ms; }
    const float *row = w + (size_t)(which * C + head * 32 + i) * C;        // c_attn.weight row (model.py:50)
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int g = 8 * (ks & 1) + e;
        const int col = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
        v[e] = row[col] * gain[col] * scale;               // ln_1.weight folded in (model.py:19-20, 50): the kernel only normalises
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)G * 8 + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// ABL (tools/bench_probes/probe_attn256.hip only; the library instantiates ABL = 0): 1 no weight DMA in the loop, 2 no softmax arithmetic,
// 4 no attention phase, 8 no projection MFMAs, 16 no ring barriers -- results are wrong unless ABL == 0.  32: wave 0 of every
// block leaves stamps[block][8] = {entry cycles, entry 100-MHz ticks, cycles after LayerNorm, exit cycles, exit ticks,
// cycles in the q|k|v projection steps, cycles waiting at the head's k / v barrier, cycles in the attention phase}.
// HP (head-parallel, round 5: small launches -- one environment's rows on the 6M shape): one workgroup per (row, head), grid = rows * 8.
// The workgroup forms the row's LayerNorm itself and runs ONE head (its six steps of the c_attn stream); the heads' y planes meet in
// the y matrix, and the packed-GEMM out-projection that follows adds the residual.  A row's eight heads then run on eight CUs at
// once instead of one after the other on one (117 -> ~30 us per attention block of a 32-row launch).  Same arithmetic per token as
// the row-per-workgroup form (bit-identical y planes).
template <class T, int NP, bool LAST, int ABL = 0, int STG = kA256Stagger, bool HP = false>
__global__ __launch_bounds__(512, 2) void attn256_kernel(const float *__restrict__ x,
                                                         const uint16_t *__restrict__ wstream, float inv_scale, float scale_log2e,
                                                         uint16_t *__restrict__ y, unsigned long long *stamps = nullptr)
{
    constexpr int C = 256, CT = 8, KS = 16, NH = 8, HS = 32, NW = 8;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = 0;
    auto phase = [&](int i) {                              // ABL & 32: cycles since the previous call go to ts[i]
        if constexpr ((ABL & 32) != 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); ts[i] += t - t_mark; t_mark = t; }
    };
    if constexpr ((ABL & 32) != 0) { ts[0] = __builtin_readcyclecounter(); ts[1] = wall_clock64(); }
    constexpr int MS = 8;                                  // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;
    constexpr int NSLOT = 5;
    constexpr int PW = MS * NP / NW;                       // direct-to-LDS loads per wave per step (2 in the split mode)
    constexpr int SPH = kA256StepsPerHead, NSTEP = NH * SPH;
    constexpr int KROW = 80, VROW = 528;                   // padded LDS rows (bytes): conflict-free b128 reads
    constexpr int NST = 4 * NP;                            // y stores per head per wave (8-byte pieces, coalesced 512 B each)
    static_assert(PW >= 1, "a wave moves at least one piece per step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [NSLOT][STEP] ring | sK [NP][256][KROW] | sV [NP][32][VROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    static_assert(!HP || !LAST, "the head-parallel form serves the full layers");
    const int64_t b = HP ? blockIdx.x / NH : blockIdx.x;
    const int hd_lo = HP ? (int)(blockIdx.x - b * NH) : 0;                                // heads of this workgroup: [hd_lo, hd_hi)
    int hd_hi = HP ? hd_lo + 1 : NH;
    // (opaque: with a trip count of one known at compile time hipcc drops the head loop, schedules the body as straight-line code and
    //  spills 63 dwords -- and scratch traffic counts in vmcnt, which breaks every hand-counted wait of the ring protocol: NaN)
    if constexpr (HP) asm volatile("" : "+s"(hd_hi));
    const int nstep = HP ? SPH : NH * SPH;                 // stream steps of this workgroup, starting at step hd_lo * SPH
    const int tok0 = wave * 32;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned sK = (unsigned)(size_t)smem + NSLOT * STEP, sV = sK + NP * kT * KROW;
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(wave * PW) * 1024 + (size_t)(hd_lo * SPH) * STEP;   // wave-uniform
    const bool full = !LAST || wave == NW - 1;             // wave-uniform: does this wave run the attention?
    int gstep = 0;

    // NSLOT = 5 is not a power of two: the ring position is carried in two scalars (slot of the step about to run, and of
    // the one before it = the slot that is refilled) instead of three `% NSLOT` per step (PMC: 0.9 SALU per MFMA before)
    int slot_cur = 0, slot_prev = NSLOT - 1;
    auto issue = [&](int G, int slot) {                    // global step G -> slot G % NSLOT (passed in)
        const unsigned char *src = wbase + (size_t)G * STEP;
        unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(wave * PW) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++)                         // one address pair and one M0 value per step: the pieces of a wave are
            dma_piece(src + lane16, dst, std::integral_constant<int, 0>{}, i);   // contiguous on both sides (immediate offsets)
    };
#pragma unroll
    for (int G = 0; G < NSLOT - 1; G++) issue(G, G);

    // ---- LayerNorm of this lane's token; operand planes in registers ----
    u32x4 xn[KS][2];
    {
        // x is chunk-major (xt_off in gpt_kernels_fast.h): every load of the wave is 1 KiB contiguous
        const float *xt = x + (b * kT + tok0) * C + r * 8 + 4 * h;
        f32x16 xv[CT];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xt + (4 * j + gq) * 256);
                xv[j][4 * gq] = v[0]; xv[j][4 * gq + 1] = v[1]; xv[j][4 * gq + 2] = v[2]; xv[j][4 * gq + 3] = v[3];
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float qv = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) { const float d = xv[j][g] - mean; xv[j][g] = d; qv += d * d; }   // (the centred row is kept)
        qv += __shfl_xor(qv, 32);
        const float rstd = rsqrtf(qv / (float)C + 1e-5f);
        // (x - mean) * rstd; ln_1.weight is part of the weight stream (pack_attn256_kernel)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int j = ks >> 1, g0 = 8 * (ks & 1);
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = xv[j][g0 + e] * rstd;
                v1[e] = xv[j][g0 + 4 + e] * rstd;
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
            xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
        }
    }

    // ---- ring protocol: sync at the top of global step G.  After the barrier the steps up to G+1 have landed for every wave
    //      and the slot of step G-1 is free; it is refilled with step G+NSLOT-1.  STORES = y stores of this wave that are
    //      younger than the pieces waited for (they retire in issue order behind them). ----
    auto sync = [&](bool stores_younger) {                 // (wave-uniform flag)
        // pieces younger than those of step G + 1: the steps G + 2, G + 3 -- as far as the stream goes.  (Rounds 3-4 allowed two steps'
        // worth of pieces at the end of the stream too, where fewer are in flight: the wait then covered nothing and the last steps'
        // pieces had landed only because they were issued three steps earlier.  Found in round 5, when the head-parallel form made
        // the stream six steps long.)
        const int beyond = nstep - 2 - gstep;              // >= 2: both issued, 1: only G + 2, <= 0: none
        if (beyond >= 2) {
            if (stores_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 2 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 2) : "memory");
        } else if (beyond == 1) {
            if (stores_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        } else {
            if (stores_younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        if constexpr (STG >= 100) { if (wave >= NW / 2) __builtin_amdgcn_s_sleep(STG - 100); }   // probe: phase offset in the projection steps
        if (!(ABL & 1) && gstep + NSLOT - 1 < nstep) issue(gstep + NSLOT - 1, slot_prev);   // (gstep + NSLOT - 1) % NSLOT
        gstep++;
    };
    // (all asm destinations are arch VGPRs here: with no "a" constraint in the kernel hipcc gives the whole 256-register
    //  budget of a two-waves-per-SIMD kernel to the arch VGPRs, MFMA accumulators included; with one it splits 128 / 128
    //  and the 128 registers of operand planes no longer fit either half)
    u32x4 wb[2][2][2];                                     // weight fragments: [set][pair 2c / 2c+1][plane]
    auto lds_frag = [&](unsigned addr, auto off_c, u32x4 &dst) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off_c)::value) : "memory");
    };
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        lds_frag(slot_addr, std::integral_constant<int, ms * NP * 1024>{}, dst[0]);
        if (NP == 2) lds_frag(slot_addr, std::integral_constant<int, (ms * NP + 1) * 1024>{}, dst[1]);
        else dst[1] = dst[0];
    };
    unsigned cur_addr = 0, nxt_addr = 0;
    auto step_begin = [&](bool stores_younger) {
        sync(stores_younger);                              // the step that runs now is gstep - 1, in slot_cur
        const int slot_next = slot_cur + 1 == NSLOT ? 0 : slot_cur + 1;
        cur_addr = lds0 + (unsigned)slot_cur * STEP;
        nxt_addr = lds0 + (unsigned)slot_next * STEP;
        slot_prev = slot_cur;
        slot_cur = slot_next;
    };
    // chunk c of a step uses pairs 2c, 2c+1 (set c & 1), requested one chunk earlier; it requests the pairs of the next chunk
    auto chunk_begin = [&](auto c_c, bool has_next) {
        constexpr int c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, 2 * c + 2>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, 2 * c + 3>{}, wb[(c + 1) & 1][1]); }
        else if (has_next) { lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]); lds_pair(nxt_addr, std::integral_constant<int, 1>{}, wb[0][1]); }
        __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this chunk's MFMAs
    };
    auto pin6 = [&]() {
#pragma unroll
        for (int n = 0; n < (NP == 2 ? 6 : 2); n++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // register octet m (registers 8m .. 8m+7) of a tile -> one 16-byte k-slot group per plane
    auto pack_octet = [&](const f32x16 &v, int m, u32x4 (&dst)[2]) {
#pragma unroll
        for (int wd = 0; wd < 4; wd++) {
            unsigned a, b2;
            split2p<T, NP>(v[8 * m + 2 * wd], v[8 * m + 2 * wd + 1], a, b2);
            dst[0][wd] = a; dst[1][wd] = b2;
        }
    };
    auto lds_write = [&](unsigned addr, const u32x4 &v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // a query's 32 scores of a key tile sit in lanes r and r + 32: exchange by v_permlane32_swap (VALU; a ds_bpermute
    // would join the fragment reads in lgkmcnt)
    // (inline asm: the builtin's second result comes back as a copy of the first with this hipcc -- measured, tools/ history;
    //  s_nop 1 = the two wait states between a VALU write and v_permlane*_swap reading it)
    auto half_swap = [&](float v, float &lower, float &upper) {
        lower = v; upper = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lower), "+v"(upper));
        // lower = value of lane r (lanes >= 32 received it), upper = value of lane r + 32 (lanes < 32 received it)
    };
    auto other_half_max = [&](float v) { float a, b2; half_swap(v, a, b2); return fmaxf(a, b2); };
    auto other_half_sum = [&](float v) { float a, b2; half_swap(v, a, b2); return a + b2; };
    const float sc2 = scale_log2e * inv_scale * inv_scale; // softmax exponent scale for q.k in weight-scaled units
    const unsigned kw_addr = sK + (unsigned)(tok0 + r) * KROW + h * 16;                   // this lane's key row (write side)
    const unsigned vw_addr = sV + (unsigned)r * VROW + (unsigned)wave * 64 + h * 16;      // this lane's d row, this wave's keys
    const unsigned kr_addr = sK + (unsigned)r * KROW + h * 16;                            // read side: key r of a tile
    const unsigned vr_addr = sV + (unsigned)r * VROW + h * 16;                            // read side: d = r

    // the first pairs of the very first step
    step_begin(false);
    lds_pair(cur_addr, I0{}, wb[0][0]);
    lds_pair(cur_addr, I1{}, wb[0][1]);
    if constexpr ((ABL & 32) != 0) { ts[2] = __builtin_readcyclecounter(); t_mark = ts[2]; }

#pragma unroll 1
    for (int hd = hd_lo; hd < hd_hi; hd++) {
        // ---- steps 0-3: q and k tiles (swapped: lane = token, registers = d) ----
        f32x16 qa, ka;
#pragma unroll
        for (int g = 0; g < 16; g++) { qa[g] = 0.f; ka[g] = 0.f; }
        // y stores of the previous head's attention are younger than the pieces that steps 0-2 wait for (see sync)
        const bool st_young = hd > hd_lo && full;
        auto step_qk = [&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            if (j > 0 || hd > hd_lo) step_begin(j < 3 && st_young);
            auto chunk = [&](auto c_c) {
                constexpr int c = decltype(c_c)::value;
                chunk_begin(c_c, true);
                constexpr int ks = 4 * j + c;
                if (ABL & 8) asm volatile("" :: "v"(wb[c & 1][0][0]), "v"(wb[c & 1][0][1]), "v"(wb[c & 1][1][0]), "v"(wb[c & 1][1][1]));
                else if (NP == 2) {
                    qa = T::mfma(wb[c & 1][0][1], xn[ks][0], qa); ka = T::mfma(wb[c & 1][1][1], xn[ks][0], ka);
                    qa = T::mfma(wb[c & 1][0][0], xn[ks][1], qa); ka = T::mfma(wb[c & 1][1][0], xn[ks][1], ka);
                }
                if (!(ABL & 8)) { qa = T::mfma(wb[c & 1][0][0], xn[ks][0], qa); ka = T::mfma(wb[c & 1][1][0], xn[ks][0], ka); }
                pin6();
            };
            chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
        };
        step_qk(I0{});
        step_qk(I1{});
        step_qk(I2{});
        step_qk(I3{});
        u32x4 qf[2][2];                                    // B operand of S^T = K Q^T: [k-step][plane]
#pragma unroll
        for (int ks = 0; ks < 2; ks++) pack_octet(qa, ks, qf[ks]);
        {   // k -> sK[pl][key = tok0 + r][octet ks][half h]   (all waves passed this head's step syncs: the previous head's
            // attention is over everywhere)
            u32x4 kp[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) pack_octet(ka, ks, kp[ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int pl = 0; pl < NP; pl++) lds_write(kw_addr + (unsigned)(pl * kT * KROW + ks * 32), kp[ks][pl]);
        }
        // ---- steps 4-5: v tile (natural: lane = d, registers = tokens), two chains ----
        f32x16 va, vb;
#pragma unroll
        for (int g = 0; g < 16; g++) { va[g] = 0.f; vb[g] = 0.f; }
        auto step_v = [&](auto j_c, bool has_next) {
            constexpr int j = decltype(j_c)::value;
            step_begin(false);
            auto chunk = [&](auto c_c) {
                constexpr int c = decltype(c_c)::value;
                chunk_begin(c_c, has_next);
                constexpr int ks = 8 * j + 2 * c;
                if (ABL & 8) asm volatile("" :: "v"(wb[c & 1][0][0]), "v"(wb[c & 1][0][1]), "v"(wb[c & 1][1][0]), "v"(wb[c & 1][1][1]));
                else if (NP == 2) {
                    va = T::mfma(xn[ks][1], wb[c & 1][0][0], va); vb = T::mfma(xn[ks + 1][1], wb[c & 1][1][0], vb);
                    va = T::mfma(xn[ks][0], wb[c & 1][0][1], va); vb = T::mfma(xn[ks + 1][0], wb[c & 1][1][1], vb);
                }
                if (!(ABL & 8)) { va = T::mfma(xn[ks][0], wb[c & 1][0][0], va); vb = T::mfma(xn[ks + 1][0], wb[c & 1][1][0], vb); }
                pin6();
            };
            chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
        };
        step_v(I0{}, true);
        step_v(I1{}, hd + 1 < hd_hi);
        {   // v^T -> sV[pl][d = r][(wave, octet mm)][half h]
#pragma unroll
            for (int g = 0; g < 16; g++) va[g] += vb[g];
            u32x4 vp[2][2];
#pragma unroll
            for (int mm = 0; mm < 2; mm++) pack_octet(va, mm, vp[mm]);
#pragma unroll
            for (int mm = 0; mm < 2; mm++)
#pragma unroll
                for (int pl = 0; pl < NP; pl++) lds_write(vw_addr + (unsigned)(pl * HS * VROW + mm * 32), vp[mm][pl]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        phase(5);
        __builtin_amdgcn_s_barrier();                      // k, v^T of the head complete
        phase(6);

        // ---- attention of this wave's 32 queries against the 256 keys of the head ----
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 16; g++) o[g] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        if (full && !(ABL & 4)) {
            if constexpr (!LAST && STG > 0 && STG < 100) { if (wave >= NW / 2) __builtin_amdgcn_s_sleep(STG); }
            u32x4 kf[2][2], vf[2][2];
            auto load_k = [&](int kt) {                    // K fragments of key tile kt: [k-step][plane]
                const unsigned a = kr_addr + (unsigned)kt * (32 * KROW);
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][0]) : "v"(a) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(kf[1][0]) : "v"(a) : "memory");
                if (NP == 2) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[0][1]) : "v"(a), "n"(kT * KROW) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][1]) : "v"(a), "n"(kT * KROW + 32) : "memory");
                } else { kf[0][1] = kf[0][0]; kf[1][1] = kf[1][0]; }
            };
            load_k(0);
#pragma unroll 1
            for (int kt = 0; kt < kT / 32; kt++) {
                f32x16 sc;
#pragma unroll
                for (int g = 0; g < 16; g++) sc[g] = 0.f;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ks++) sc = mma<T, NP>(kf[ks], qf[ks], sc);
                __builtin_amdgcn_sched_barrier(0);
                // V^T fragments of this tile, then K of the next one (both land during the softmax arithmetic)
                {
                    const unsigned a = vr_addr + (unsigned)kt * 64;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(vf[0][0]) : "v"(a) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(vf[1][0]) : "v"(a) : "memory");
                    if (NP == 2) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[0][1]) : "v"(a), "n"(HS * VROW) : "memory");
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[1][1]) : "v"(a), "n"(HS * VROW + 32) : "memory");
                    } else { vf[0][1] = vf[0][0]; vf[1][1] = vf[1][0]; }
                }
                if (kt + 1 < kT / 32) load_k(kt + 1);
                // sc[g] = S[query r][key 32 kt + tau(g, h)]  (times 1/inv_scale^2)
                float mx = sc[0];
                if (!(ABL & 2)) {
#pragma unroll
                for (int g = 1; g < 16; g++) mx = fmaxf(mx, sc[g]);
                mx = other_half_max(mx);
                }
                if (!(ABL & 2) && __builtin_amdgcn_ballot_w64(mx > m_run) != 0) {        // some query's running max moved: rescale (wave-uniform branch)
                    const float m_new = fmaxf(m_run, mx);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
                    l_run *= alpha;
#pragma unroll
                    for (int g = 0; g < 16; g++) o[g] *= alpha;
                    m_run = m_new;
                }
                const float nm = -m_run * sc2;
                float psum = 0.f;
                if (!(ABL & 2)) {
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    sc[g] = __builtin_amdgcn_exp2f(fmaf(sc[g], sc2, nm));
                    psum += sc[g];
                }
                l_run += other_half_sum(psum);
                } else l_run = 1.f;
                u32x4 pf[2][2];
#pragma unroll
                for (int mm = 0; mm < 2; mm++) pack_octet(sc, mm, pf[mm]);
                if (kt + 1 < kT / 32) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NP) : "memory");   // v^T fragments landed, K of the next tile may fly
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mm = 0; mm < 2; mm++) o = mma<T, NP>(vf[mm], pf[mm], o);
            }
        }
        // ---- y planes of the head: o[g] = O[query r][d = tau(g, h)] / l, times the v projection's weight scale ----
        if (full) {
            const float inv = inv_scale / l_run;
            const int64_t m = LAST ? b : b * kT + tok0 + r;                // row of the y matrix
            if (!LAST || r == 31) {
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    unsigned h0, l0, h1, l1;
                    split2p<T, NP>(o[4 * gq] * inv, o[4 * gq + 1] * inv, h0, l0);
                    split2p<T, NP>(o[4 * gq + 2] * inv, o[4 * gq + 3] * inv, h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    const int n = hd * HS + 8 * gq + 4 * h;
                    *reinterpret_cast<u32x2 *>(y + pk_off(m, n, 0, C >> 4, NP)) = hi;
                    if (NP == 2) *reinterpret_cast<u32x2 *>(y + pk_off(m, n, 1, C >> 4, NP)) = lo;
                }
            }
        }
        phase(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr ((ABL & 32) != 0) {
        ts[3] = __builtin_readcyclecounter(); ts[4] = wall_clock64();
        if (tid == 0)
#pragma unroll
            for (int i = 0; i < 8; i++) stamps[(size_t)blockIdx.x * 8 + i] = ts[i];
    }
}

}  // namespace fastk
}  // namespace mgpt
