// gpt_kernels_attn_tiles.h -- the attention phase of the persistent attention-block kernels (attn256o_kernel, attn160o_kernel), gfx950:
// one wave, 32 queries (lane (r, h) = query r, half h of a key tile), the head's K planes [plane][key][KROW] and V^T planes
// [plane][d][VROW] in LDS, q as the B operand of S^T = K Q^T in registers (qf), head size 32, 256 keys in eight tiles of 32.
//     attention_tiles        the pipelined key-tile loop (round 5), q and k in EXPONENT units (a score is log2 of its softmax weight up
//                            to the query's reference): o, l_run out; falls back to ...
//     attention_exact_tiles  ... the online-softmax loop with a running maximum per query (rounds 2-4), score scale sc2
// Both leave o = sum_j p_j v_j (times the v projection's weight scale) and l_run = sum_j p_j of the wave's queries; the caller
// normalises.  LDS reads are inline asm with hand-counted lgkmcnt (a compiler-visible LDS access would make hipcc drain the
// callers' direct-to-LDS weight rings with vmcnt(0)).
#pragma once
#include "gpt_kernels_c256p.h"

// VALU instructions placed behind each MFMA of the pipelined loop's sub-blocks B1 / B2 (A/B knobs: tools/bench_probes/check_attn256o.hip)
#ifndef MGPT_ATT_NVB1
#define MGPT_ATT_NVB1 10
#endif
#ifndef MGPT_ATT_NVB2
#define MGPT_ATT_NVB2 6
#endif

namespace mgpt {
namespace fastk {

// waves that threw a head of the pipelined loop away and redid it with the exact loop; read by mgpt_gpt_debug_counter (tests: zero
// on the synthetic N(0, 0.02) checkpoints, non-zero when the scores are made to spread)
__device__ unsigned long long g_attn_fallbacks = 0;

namespace attn_tiles_detail {
template <class T, int NP>
__device__ __forceinline__ void pack_octet(const f32x16 &v, int m, u32x4 (&dst)[2])
{
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
        unsigned a, b2;
        split2p<T, NP>(v[8 * m + 2 * wd], v[8 * m + 2 * wd + 1], a, b2);
        dst[0][wd] = a; dst[1][wd] = b2;
    }
}
__device__ __forceinline__ void half_swap(float v, float &lower, float &upper)
{
    lower = v; upper = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lower), "+v"(upper));
}
__device__ __forceinline__ float other_half_max(float v) { float a, b2; half_swap(v, a, b2); return fmaxf(a, b2); }
__device__ __forceinline__ float other_half_sum(float v) { float a, b2; half_swap(v, a, b2); return a + b2; }
}  // namespace attn_tiles_detail

// The EXACT loop (rounds 2-4): online softmax with a running maximum per query, one key tile after the other (6 S MFMAs, the softmax
// arithmetic, 6 PV MFMAs).  Since round 5 the FALLBACK of attention_tiles (and the whole phase under -DMGPT_AB_ATTN_CLUMPED): a wave
// whose scores outgrow the fp16 range of the P planes redoes its head here.
template <class T, int NP, int KROW, int VROW, int HS>
__device__ __forceinline__ void attention_exact_tiles(unsigned kr_addr, unsigned vr_addr, const u32x4 (&qf)[2][2], float sc2, f32x16 &o, float &l_run)
{
    using namespace attn_tiles_detail;
    auto pack_octet = [&](const f32x16 &v, int m, u32x4 (&dst)[2]) { attn_tiles_detail::pack_octet<T, NP>(v, m, dst); };
    // (the start values come out of an opaque asm: as plain constants hipcc hoisted a zero block and -inf out of the ROW loop
    //  -- this path being cold -- and paid for their 17 registers with spills in the LayerNorm prologue)
    float m_run, zero;
    asm volatile("v_mov_b32 %0, 0xff800000\n\tv_mov_b32 %1, 0" : "=v"(m_run), "=v"(zero));
    l_run = zero;
#pragma unroll
    for (int g = 0; g < 16; g++) o[g] = zero;
{
    u32x4 kf[2][2], vf[2][2];
    auto load_k = [&](int kt) {                // K fragments of key tile kt: [k-step][plane]
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
#pragma unroll
        for (int g = 1; g < 16; g++) mx = fmaxf(mx, sc[g]);
        mx = other_half_max(mx);
        if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {        // some query's running max moved: rescale (wave-uniform branch)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
            l_run *= alpha;
#pragma unroll
            for (int g = 0; g < 16; g++) o[g] *= alpha;
            m_run = m_new;
        }
        const float nm = -m_run * sc2;
        // (Two variants of this loop were built and measured in round 4, A/B in one box, cfg3 attention ms per step:
        //  v_pk_fma_f32 / v_pk_add_f32 on score pairs, 15 instructions fewer per key tile: 54.5 against 53.9;
        //  software pipelining -- the S MFMAs of tile kt + 1 issued before the softmax arithmetic of tile kt, the second
        //  score block in the 16 registers of the next step's prefetched weight fragments, bit-identical results: 54.2
        //  against 54.0.  Neither the count of full-rate VALU instructions nor the MFMA / VALU order inside a wave
        //  bounds this phase: the second wave of the SIMD already fills the gaps, and what is saved in cycles comes
        //  back as a lower clock (HISTORY.md section 12, round 3).)
        float psum = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            sc[g] = __builtin_amdgcn_exp2f(fmaf(sc[g], sc2, nm));
            psum += sc[g];
        }
        l_run += other_half_sum(psum);
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
}

template <class T, int NP, int KROW, int VROW, int HS>
__device__ __forceinline__ void attention_tiles(unsigned kr_addr, unsigned vr_addr, const u32x4 (&qf)[2][2], int lane, f32x16 &o, float &l_run)
{
    using namespace attn_tiles_detail;
    auto pack_octet = [&](const f32x16 &v, int m, u32x4 (&dst)[2]) { attn_tiles_detail::pack_octet<T, NP>(v, m, dst); };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // Round 5: the key-tile loop as a MODULO-SCHEDULED pipeline placed one MFMA at a time, with ONE reference per query.
    // What the hardware does (tools/bench_probes/probe_interleave.hip, profiles/r05_probe_interleave.txt): the two waves of a
    // SIMD TIME-SLICE -- a second wave adds 5-10 % of throughput, an MFMA of one wave never covers the VALU work of the other
    // -- and inside ONE wave about six VALU instructions ride free behind every MFMA when they are placed BETWEEN MFMAs;
    // beyond that every VALU instruction costs its 4 cycles and every MFMA ~15.  Round 4's loop ran a tile as three clumps
    // (6 S MFMAs, ~95 VALU, 6 PV MFMAs: 1 700 - 1 940 cycles per tile pair of a SIMD, the sum of everything).  Two changes:
    // (1) PIPELINE.  Tile kt's softmax arithmetic rides under the MFMAs of its neighbours:
    //         B1  exp2, row sum, split of octet 0     under   PV k-step 1 of tile kt - 1 (3 MFMAs) and S(kt + 1) -> the other
    //                                                         score block (6 MFMAs), two chains alternating
    //         B2  split of octet 1, sums              under   PV k-step 0 of tile kt (3 MFMAs)
    //     (with round 4's running maximum kept, this alone -- bit-identical to round 4 -- took the attention of a cfg3 step from 53.5 to
    //      51.2 ms, profiles/r05_ab.txt)
    // (2) FEWER VALU INSTRUCTIONS: the phase is VALU-issue bound (~95 per tile against 12 MFMAs), so the running maximum
    //     goes (8 v_max3 + half swap + compare + branch per tile): every query takes the maximum of its FIRST key tile as
    //     the reference of the whole head, p = exp2(s - ref) may exceed 1, and the cross-half sums of l are taken once per
    //     head.  Softmax is shift-invariant, fp32 carries p, l and o up to 2^127; the one thing that is not free is the
    //     fp16 range of the P planes (hi = fp16(p) <= 65504): a wave in which a lane's half-row sum reaches 60 000 (some
    //     score more than ~11 nats above its query's first-tile maximum), or is not finite, throws its head away and
    //     redoes it with attention_exact_tiles() (wave-uniform branch; K and V^T stay in LDS until the next head's writes, which
    //     wait for every wave).  Deterministic per row: the decision depends on the wave's own 32 queries only.
    // LDS reads of the phase return in issue order; per tile: [after B1a] V^T k-step 0 of tile kt (NP reads); [B1 end] V^T k-step
    // 1 of tile kt (NP); [B2 end] K of tile kt + 2 (2 NP).  lgkmcnt(N): N = reads issued after the one needed.
    u32x4 kf[2][2], vf[2][2], pf[2][2];
    f32x16 sA, sB;                             // score blocks of the even / odd key tiles
    constexpr int NM = NP == 2 ? 3 : 1;        // MFMAs per k-step
#ifdef MGPT_ABL_ATT                                        // tools/bench_probes/check_attn256o.hip only (results are wrong): 1 = no MFMAs, 2 = no softmax arithmetic in the phase
    auto amfma = [&](u32x4 a, u32x4 b2, f32x16 c) {
        if constexpr ((MGPT_ABL_ATT & 1) != 0) { asm volatile("" : "+v"(c)); return c; }
        else return T::mfma(a, b2, c);
    };
#else
    auto amfma = [&](u32x4 a, u32x4 b2, f32x16 c) { return T::mfma(a, b2, c); };
#endif
    auto amma = [&](const u32x4 (&a)[2], const u32x4 (&b2)[2], f32x16 c) {     // = mma<T, NP>: small terms first
        if (NP == 2) { c = amfma(a[1], b2[0], c); c = amfma(a[0], b2[1], c); }
        return amfma(a[0], b2[0], c);
    };
    auto load_k = [&](auto kt_c) {             // K fragments of key tile kt: [k-step][plane]
        constexpr int off = decltype(kt_c)::value * (32 * KROW);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[0][0]) : "v"(kr_addr), "n"(off) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][0]) : "v"(kr_addr), "n"(off + 32) : "memory");
        if (NP == 2) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[0][1]) : "v"(kr_addr), "n"(off + kT * KROW) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][1]) : "v"(kr_addr), "n"(off + kT * KROW + 32) : "memory");
        } else { kf[0][1] = kf[0][0]; kf[1][1] = kf[1][0]; }
    };
    auto load_v = [&](auto kt_c, auto mm_c) {  // V^T fragments of key tile kt, k-step mm: [plane]
        constexpr int off = decltype(kt_c)::value * 64 + decltype(mm_c)::value * 32, mm = decltype(mm_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[mm][0]) : "v"(vr_addr), "n"(off) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[mm][1]) : "v"(vr_addr), "n"(off + HS * VROW) : "memory");
        else vf[mm][1] = vf[mm][0];
    };
    auto lgkm = [&](auto n_c) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(n_c)::value) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // NMF MFMAs, each followed by NV VALU instructions; what is left of the VALU work goes behind the last one
    auto place = [&](auto nmf_c, auto nv_c) {
#pragma unroll
        for (int n = 0; n < decltype(nmf_c)::value; n++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, decltype(nv_c)::value, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using LK = std::integral_constant<int, 2 * NP>;       // reads of one K tile
    using LV = std::integral_constant<int, NP>;           // reads of one V^T k-step
    load_k(I0{});
    lgkm(I0{});
#pragma unroll
    for (int g = 0; g < 16; g++) sA[g] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) sA = amma(kf[ks], qf[ks], sA);
    __builtin_amdgcn_sched_barrier(0);
    load_k(I1{});
    // the reference of the head: the maximum of the query's first key tile.  sA[g] = S[query r][key tau(g, h)], in exponent
    // units (QK_UNITS).  The scores of tiles 1-7 START from -reference: the first S MFMA of a tile takes the block nmb (the
    // value in all 16 registers) as its C operand, so that exp2 applies to the accumulator as it is -- 16 multiply-adds per
    // tile fewer; tile 0, whose scores exist before the reference does, pays 16 additions once per head.
    f32x16 nmb;
    {
        float mx = sA[0];
#pragma unroll
        for (int g = 1; g < 16; g++) mx = fmaxf(mx, sA[g]);
        const float nm = -other_half_max(mx);
#pragma unroll
        for (int g = 0; g < 16; g++) { nmb[g] = nm; sA[g] += nm; }
    }
    float l_part = 0.f;                        // this lane's half of the row sum (all eight tiles)
    auto tile = [&](auto kt_c, f32x16 &cur, f32x16 &nxt) {
        constexpr int kt = decltype(kt_c)::value;
        constexpr bool FIRSTT = kt == 0, LASTT = kt == kT / 32 - 1, HAS2 = kt + 2 < kT / 32;
        // ---- B1a: the second k-step of the previous tile's PV; the first exponentials ----
        // (read issued after V^T k-step 1 of tile kt - 1: K of tile kt + 1)
#if !defined(MGPT_ABL_ATT) || (MGPT_ABL_ATT & 2) == 0
        if constexpr (!FIRSTT) {
            lgkm(std::integral_constant<int, LASTT ? 0 : LK::value>{});
            o = amma(vf[1], pf[1], o);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                cur[g] = __builtin_amdgcn_exp2f(cur[g]);
                l_part += cur[g];
            }
            place(std::integral_constant<int, NM>{}, std::integral_constant<int, 4>{});
            asm volatile("" : "+v"(o));
        }
#else
        if constexpr (!FIRSTT) { lgkm(std::integral_constant<int, LASTT ? 0 : LK::value>{}); o = amma(vf[1], pf[1], o); __builtin_amdgcn_sched_barrier(0); }
#endif
        // (V^T k-step 0 of THIS tile is requested only here, and K of tile kt + 2 only after B2: requested earlier their registers
        //  were live next to pf[1] / vf[1] above resp. next to the three P planes of B2, and xn paid for them with scratch)
        load_v(kt_c, I0{});
        // ---- B1b: the rest of the exponentials, row sum, split of octet 0 under S(kt + 1) ----
        // (read issued after K of tile kt + 1: V^T k-step 0 of tile kt)
        if constexpr (!LASTT) {
            lgkm(LV{});
            nxt = amma(kf[0], qf[0], nmb);
            nxt = amma(kf[1], qf[1], nxt);
        }
#if !defined(MGPT_ABL_ATT) || (MGPT_ABL_ATT & 2) == 0
#pragma unroll
        for (int g = FIRSTT ? 0 : 4; g < 16; g++) {
            cur[g] = __builtin_amdgcn_exp2f(cur[g]);
            l_part += cur[g];
        }
        pack_octet(cur, 0, pf[0]);
#else
        l_part += cur[3];
#pragma unroll
        for (int e = 0; e < 4; e++) { pf[0][0][e] = __builtin_bit_cast(unsigned, cur[e]); pf[0][1][e] = __builtin_bit_cast(unsigned, cur[4 + e]); }
#endif
        place(std::integral_constant<int, LASTT ? 0 : 2 * NM>{}, std::integral_constant<int, MGPT_ATT_NVB1>{});
        if constexpr (!LASTT) asm volatile("" : "+v"(nxt));
        load_v(kt_c, I1{});
        // ---- B2: split of octet 1 under the first k-step of this tile's PV ----
        // (read issued after V^T k-step 0 of tile kt: V^T k-step 1 of tile kt)
        lgkm(LV{});
        if constexpr (FIRSTT) {                // (o starts here: a zero block held across the first tile cost 16 registers -- hipcc spilled it)
#pragma unroll
            for (int g = 0; g < 16; g++) o[g] = 0.f;
        }
        o = amma(vf[0], pf[0], o);
#if !defined(MGPT_ABL_ATT) || (MGPT_ABL_ATT & 2) == 0
        pack_octet(cur, 1, pf[1]);
#else
#pragma unroll
        for (int e = 0; e < 4; e++) { pf[1][0][e] = __builtin_bit_cast(unsigned, cur[8 + e]); pf[1][1][e] = __builtin_bit_cast(unsigned, cur[12 + e]); }
#endif
        place(std::integral_constant<int, NM>{}, std::integral_constant<int, MGPT_ATT_NVB2>{});
        asm volatile("" : "+v"(o));
        if constexpr (HAS2) load_k(std::integral_constant<int, HAS2 ? kt + 2 : 0>{});
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>; using K5 = std::integral_constant<int, 5>; using K6 = std::integral_constant<int, 6>; using K7 = std::integral_constant<int, 7>;
    static_assert(kT / 32 == 8, "eight key tiles");
    tile(K0{}, sA, sB); tile(K1{}, sB, sA); tile(K2{}, sA, sB); tile(K3{}, sB, sA);
    tile(K4{}, sA, sB); tile(K5{}, sB, sA); tile(K6{}, sA, sB); tile(K7{}, sB, sA);
    // the second k-step of the last tile's PV
    lgkm(I0{});
    o = amma(vf[1], pf[1], o);
    __builtin_amdgcn_sched_barrier(0);
    l_run = other_half_sum(l_part);
    // every p is positive, so a half-row sum below 60 000 bounds every p of the lane; !(a < b) is also true for NaN
    if (__builtin_amdgcn_ballot_w64(!(l_part < 60000.0f)) != 0) {
        if (lane == 0) atomicAdd(&g_attn_fallbacks, 1ull);
        attention_exact_tiles<T, NP, KROW, VROW, HS>(kr_addr, vr_addr, qf, 1.0f, o, l_run);
    }
}

}  // namespace fastk
}  // namespace mgpt
