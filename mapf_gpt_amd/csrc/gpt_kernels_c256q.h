// gpt_kernels_c256q.h -- mlp256p_kernel (gpt_kernels_c256p.h: roles, ring, period, counted waits -- read its header first) on v_mfma_f32_16x16x32 instead of
// v_mfma_f32_32x32x16.  At the package power limit the small shape delivers 13-15 % more f16 flops per second (DESIGN section 10 fact 5,
// profiles/r05_probe_mfma_shape.txt); the forward runs at that limit.  Nothing of the protocol changes -- a step is still 16 fragment pairs of 2 KiB, a chunk two of
// them, the same reads, pieces, waits and barrier -- only what a fragment holds and which registers an MFMA touches (lane l: t = l % 16, q = l / 16; a wave's 32 tokens
// are the two groups tg = 0, 1 of tokens 16 tg + t):
//   c_fc     A = weights: fragment (kb, ug) = rows (hidden units) x 32 features of k-block kb; row rho of unit group ug is hidden unit 8 (rho / 4) + 4 ug + rho % 4
//            B = operand planes xn[tg * 8 + kb]: token 16 tg + t, features 32 kb + 8 q .. + 7        D(ug, tg): token 16 tg + t, units 8 q + 4 ug + i
//            step `half` holds k-blocks 4 half .. + 3 (pairs ms = 2 (kb % 4) + ug); chunk c = k-block 4 half + c: 2 ug x 2 tg x 3 products = 12 MFMAs
//   GELU     a lane's two quads D(0, tg), D(1, tg) are the units 8 q .. 8 q + 7: ONE K = 32 operand of c_proj per token group (hand-off slot tg, where the
//            32 x 32 form had the tile's two k-steps) -- the hand-off code does not change, the tile's 16 pre-activations sit at 4 (2 tg + ug) + i
//   c_proj   A = weights: fragment fg = output features 16 fg + rho x the tile's 32 units (8 q + e); step kk holds fg = 8 kk .. + 7
//            B = hidden planes hf[tg]                                                            D(fg, tg): token 16 tg + t, features 16 fg + 4 q + i
//            at acc[fg / 2][4 (2 (fg % 2) + tg) + i]; chunk c = fg 8 kk + 2 c, + 1: 12 MFMAs, the tg = 0 products first -- a tile's SECOND hidden slot is complete only
//            one step before the tile's first c_proj step (it is requested at that step's top and used six MFMAs later), the NEXT tile's first slot is requested
//            behind the last tg = 0 product of the tile's last chunk
//   x        chunk-major rows: lane (t, q) owns the 32 bytes of token 16 tg + t in chunk 4 kb + q (operand planes) resp. the 16 bytes at half q % 2 of chunk
//            2 fg + q / 2 (residual quads); LayerNorm folds its sums over the FOUR lanes of a token (v_permlane16_swap, v_permlane32_swap)
#pragma once
#include "gpt_kernels_c256p.h"

namespace mgpt {
namespace fastk {

#ifdef MGPT_ABL_MLPQ
constexpr int kMQAbl = MGPT_ABL_MLPQ;
#else
constexpr int kMQAbl = 0;
#endif

// weight stream: [period step R][pair ms][plane][lane][8]; pairs 0-7 = c_fc (gain folded in), 8-15 = c_proj
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp256q_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                           const float *__restrict__ gain, uint16_t *__restrict__ out,
                                                           float scale1, float scale2)
{
    constexpr int C = 256;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (step, pair, lane)
    if (gid >= (int64_t)kMPPeriod * 16 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 15), R = (int)(gid >> 10);
    const int rho = lane & 15, qk = lane >> 4;                            // operand row, k group of 8
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.f;
    if (ms < 8) {                                                         // c_fc(tile t): k-block kb, unit group ug
        if (R >= kMPPause) {
            const int rr = R - kMPPause, t = rr >> 1, kb = 4 * (rr & 1) + (ms >> 1), ug = ms & 1;
            const int unit = 32 * t + 8 * (rho >> 2) + 4 * ug + (rho & 3);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int feat = 32 * kb + 8 * qk + e;
                v[e] = fc_w[(size_t)unit * C + feat] * gain[feat] * scale1;   // LayerNorm weight folded in (model.py:19-20, 86)
            }
        }
    } else {                                                              // c_proj(tile t), step kk: output features 16 fg + rho, fg = 8 kk + (ms - 8)
        int t = -1, kk = 0;
        if (R < 4) { t = 30 + (R >> 1); kk = R & 1; }                     // the previous block's last two tiles
        else if (R >= kMPPause + 4) { t = (R - kMPPause - 4) >> 1; kk = (R - kMPPause - 4) & 1; }
        if (t >= 0) {
            const int fg = 8 * kk + (ms - 8);
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] = pj_w[(size_t)(16 * fg + rho) * (4 * C) + 32 * t + 8 * qk + e] * scale2;
        }
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)R * 16 + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// gelu_lut: the Phi table TIMES inv1 (GELU(v) = v_stream * (Phi(v) * inv1), the same bits as (v_stream * inv1) * Phi(v)).
// STAMPS (tools/bench_probes/check_mlp256p.hip only): 1 = wave 0 and wave 4 of every workgroup leave s_memtime / s_memrealtime at entry and
// exit; 2 = cycles spent in wait + barrier instead of the exit wall clock; 3 = cycles per phase of a step: {wait + barrier,
// DMA issue + slot bookkeeping, chunks 0-2, chunk 3 (up to the next step's top)}.
// NPAIR (round 5): producer / consumer pairs per workgroup = 32-token tiles per block.  4 = the throughput form above (128-token blocks,
// two waves per SIMD, the producers fill the ring).  2 = small launches (one environment's 8 192 tokens are 64 blocks of 128 -- a
// quarter of the chip): 64-token blocks, four waves with a SIMD each, and ALL four issue ring pieces (8 per wave and step, as the
// producers of the 4-pair form) -- with only the two producers issuing, 16 pieces per wave and step would cost more than the block
// gains.  Same arithmetic per token (the block a token falls in never enters it).
template <class T, int NP, int STAMPS = 0, int NPAIR = 4>
__global__ __launch_bounds__(NPAIR * 128, 2) void mlp256q_kernel(float *__restrict__ x, const uint16_t *__restrict__ wstream, float inv1,
                                                         float inv2, const float2 *__restrict__ gelu_lut, int n_blocks,
                                                         unsigned long long *stamps = nullptr)
{
    constexpr int C = 256;
    constexpr int MS = 16;                                 // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;                   // bytes per stream step
    constexpr int NSLOT = kMPSlots;
    static_assert(NPAIR == 4 || NPAIR == 2, "four or two pairs");
    constexpr int NW = 2 * NPAIR;
    constexpr bool ISSUE_ALL = NPAIR < 4;                  // every wave issues ring pieces (small-launch form)
    constexpr int PWP = MS * NP / (ISSUE_ALL ? NW : NPAIR);   // direct-to-LDS pieces per ISSUING wave per step.  NPAIR = 4: the producers only; the consumers issue none: measured
                                                           // (STAMPS = 3), the consumer is the longer chain of a step (4 x 416 cycles of MFMA
                                                           // chunks + 290 of piece issue vs 4 x 309 + 79 with 650 cycles of barrier wait)
    constexpr int LUT_BYTES = kGeluLutN * 8;
    constexpr int NM = (NP == 2 ? 12 : 4);                 // MFMAs per chunk (two fragment pairs x two token groups)
    static_assert(NSLOT == 3, "the counted waits below assume that exactly the next step's pieces are in flight");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPAIR;                    // wave-uniform
    // (round 4: s_setprio for the consumers -- the longer chain of a step -- makes the kernel 3.6 % SLOWER, 57.6 -> 59.7 ms per cfg3
    //  step at priority 1 or 3; for the producers it changes nothing: oldest-first issue, i.e. the producers ahead, is what works)
    const int pair = wave % NPAIR;
    const int issuer = ISSUE_ALL ? wave : pair;            // which PWP-piece share of a step this wave moves
    const int t16 = lane & 15, q4 = lane >> 4;             // token inside a 16-token group, k / row group
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned lut_addr = (unsigned)(size_t)smem + NSLOT * STEP;
    const unsigned hand0 = lut_addr + LUT_BYTES + (unsigned)pair * (2 * 2 * NP * 1024) + lane16;    // this pair's hand-off, this lane
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(issuer * PWP) * 1024 + lane16;
    const int n_mine = n_blocks > (int)blockIdx.x ? (n_blocks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    unsigned long long t_in[2] = {0, 0}, t_sync = 0, t_ph[6] = {0, 0, 0, 0, 0, 0}, t_last = 0;
    auto mark = [&](int ph) {                              // STAMPS == 3: cycles since the previous mark go to phase ph
        if constexpr (STAMPS == 3) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_ph[ph] += t - t_last; t_last = t;
        }
    };
    if constexpr (STAMPS != 0) { t_in[0] = __builtin_readcyclecounter(); t_in[1] = wall_clock64(); }
    if (n_mine == 0) return;

    // ---- ring: slot of step R = R % 3.  Top of step R: this wave's pieces of step R + 1 (issued in step R - 1) have landed,
    //      every LDS access of this wave is done, barrier; then the slot of step R - 1 is refilled with step R + 2.
    //      PENDING = vector-memory operations of this wave other than ring pieces issued since (they are younger than the pieces
    //      waited for, and vector-memory operations retire in issue order). ----
    int r_issue = 0;                                       // stream step (mod period) of the next DMA
    int slot_cur = 0, slot_prev = NSLOT - 1;
    unsigned cur_addr = 0, nxt_addr = 0;
    auto issue = [&](int slot) {
        if (producer || ISSUE_ALL) {                       // wave-uniform
            const unsigned char *src = wbase + (size_t)r_issue * STEP;
            unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(issuer * PWP) * 1024;
#pragma unroll
            for (int i = 0; i < PWP; i++) dma_piece(src + (i >> 2) * 4096, dst + (i >> 2) * 4096, std::integral_constant<int, 0>{}, i & 3);
        }
        r_issue = r_issue + 1 == kMPPeriod ? 0 : r_issue + 1;
    };
    {   // Phi table -> LDS (24 pieces of 1 KiB, 24 / NW per wave); older than every ring piece
        static_assert(LUT_BYTES % (NW * 1024) == 0, "whole pieces per wave");
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut) + (size_t)wave * (LUT_BYTES / NW) + lane16;
#pragma unroll
        for (int i = 0; i < LUT_BYTES / (NW * 1024); i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + i * 1024), (lds_void_t *)(smem + NSLOT * STEP + wave * (LUT_BYTES / NW) + i * 1024), 16, 0, 0);
    }
    issue(0);
    issue(1);
    if (!producer) {                                       // hidden hand-off starts as zeros (the first block has no predecessor)
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 2 * 2 * NP; i++) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(hand0), "v"(z), "n"(i * 1024) : "memory");
    }
    auto sync = [&](auto pending_c) {
        unsigned long long t0 = 0;
        if constexpr (STAMPS == 2) t0 = __builtin_readcyclecounter();
        mark(5);
        vm_wait<decltype(pending_c)::value>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (STAMPS != 4) __builtin_amdgcn_s_barrier();       // STAMPS == 4: timing experiment without the barrier (results are wrong)
        if constexpr (STAMPS == 2) t_sync += __builtin_readcyclecounter() - t0;
        mark(0);
        issue(slot_prev);                                  // always: the stream is cyclic
        const int slot_next = slot_cur + 1 == NSLOT ? 0 : slot_cur + 1;
        cur_addr = lds0 + (unsigned)slot_cur * STEP;
        nxt_addr = lds0 + (unsigned)slot_next * STEP;
        slot_prev = slot_cur;
        slot_cur = slot_next;
    };
    using E0 = std::integral_constant<int, 0>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    u32x4 wb[2][2][2];                                     // weight fragments [set = chunk & 1][pair of the chunk][plane]
    constexpr int kS1 = (kMQAbl & 1) ? 0 : 1;              // (ablation 1: the second pair of a chunk IS the first)
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        else dst[1] = dst[0];
        if (kMQAbl & 4) {                                  // every fragment read issued twice (same bytes to the same registers)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
            if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        }
    };
    // chunk c (0 .. 3) of a step works on pairs MB + 2c, MB + 2c + 1 (set c & 1), requested one chunk earlier; it requests the
    // pairs of the next chunk (chunk 3: the first pairs of the next step, whose slot has landed) in front of its MFMAs
    auto chunk_begin = [&](auto mb_c, auto c_c, bool next_step_has_work) {
        constexpr int MB = decltype(mb_c)::value, c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 2>{}, wb[(c + 1) & 1][0]); if (!(kMQAbl & 1)) lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 3>{}, wb[(c + 1) & 1][1]); }
        else if (next_step_has_work) { lds_pair(nxt_addr, std::integral_constant<int, MB>{}, wb[0][0]); if (!(kMQAbl & 1)) lds_pair(nxt_addr, std::integral_constant<int, MB + 1>{}, wb[0][1]); }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the same requests one at a time (the consumer's placed chunks: each behind one of the chunk's first MFMAs, the matrix pipe being busy for 16 cycles per MFMA --
    // measured on attn256q_kernel, gpt_kernels_c256b.h; -DMGPT_AB_MLPQ_CLUMPED: all in front as in the producer)
#if defined(MGPT_AB_MLPQ_CLUMPED)
    constexpr bool PLACED = false;
#else
    constexpr bool PLACED = (kMQAbl == 0);
#endif
    auto chunk_read = [&](auto mb_c, auto c_c, bool next_step_has_work, auto n_c) {
        constexpr int MB = decltype(mb_c)::value, c = decltype(c_c)::value, n = decltype(n_c)::value;
        if constexpr (n < 2 * NP) {
            constexpr int fr = n / NP, pl = n % NP;
            if (c < 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wb[(c + 1) & 1][fr][pl]) : "v"(cur_addr), "n"(((MB + 2 * c + 2 + fr) * NP + pl) * 1024) : "memory");
            else if (next_step_has_work) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wb[0][fr][pl]) : "v"(nxt_addr), "n"(((MB + fr) * NP + pl) * 1024) : "memory");
            if (NP == 1) wb[(c + 1) & 1][fr][1] = wb[(c + 1) & 1][fr][0];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pin_rest = [&](auto n_c) {                         // the chunk's MFMAs behind the placed ones
#pragma unroll
        for (int n = 0; n < decltype(n_c)::value; n++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pin = [&](auto n_valu_c) {
        constexpr int n_valu = decltype(n_valu_c)::value;
#pragma unroll
        for (int n = 0; n < NM; n++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (n_valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, n_valu, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // one product term on slice S (registers 4 S .. 4 S + 3) of a 16-register block
    auto mm16 = [&](const u32x4 &a, const u32x4 &b, f32x16 &blk, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        f32x4 c = {blk[4 * S], blk[4 * S + 1], blk[4 * S + 2], blk[4 * S + 3]};
        c = T::mfma16(a, b, c);
        blk[4 * S] = c[0]; blk[4 * S + 1] = c[1]; blk[4 * S + 2] = c[2]; blk[4 * S + 3] = c[3];
    };
    // the first term of a slice: C = 0 as the MFMA's inline constant (a slice zeroed by assignment costs four v_mov per slice: 16 x 16 x 32 has no 16-register
    // destination to start from zero at once, and hipcc does not fold the assignments into the first MFMA of the slice)
    auto mm16z = [&](const u32x4 &a, const u32x4 &b, f32x16 &blk, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = T::mfma16(a, b, c);
        blk[4 * S] = c[0]; blk[4 * S + 1] = c[1]; blk[4 * S + 2] = c[2]; blk[4 * S + 3] = c[3];
    };

    if (producer) {
        // =============================================== producer ===============================================
        using MB = std::integral_constant<int, 0>;         // first pair of this role in a step
        u32x4 xn[16][2];                                   // operand planes of this lane's token: [k-step][plane]
        f32x16 hA, hB;                                     // pre-activations of the even / odd hidden tile
        const float lut_scale = inv1 * kGeluLutScale;
        float gvv[4], gfr[4];
        f32x2 gtab[4];
        unsigned hw[2][4];                                 // hidden words of one k-step: [plane][word]
        // chunk: unit groups w0 (ug = 0), w1 (ug = 1) of one k-block against the two token groups x0, x1; slice 2 tg + ug.  Small terms first, the four
        // accumulator chains interleaved
        // FIRST: the tile's first chunk -- the slices start from zero
        auto fc_mma = [&](const u32x4 (&w0)[2], const u32x4 (&w1)[2], const u32x4 (&x0)[2], const u32x4 (&x1)[2], f32x16 &hd, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            auto head4 = [&](const u32x4 &a0, const u32x4 &a1, const u32x4 &b0, const u32x4 &b1) {
                if constexpr (FIRST) { mm16z(a0, b0, hd, I0{}); mm16z(a1, b0, hd, I1{}); mm16z(a0, b1, hd, I2{}); mm16z(a1, b1, hd, I3{}); }
                else { mm16(a0, b0, hd, I0{}); mm16(a1, b0, hd, I1{}); mm16(a0, b1, hd, I2{}); mm16(a1, b1, hd, I3{}); }
            };
            if (NP == 2) {
                head4(w0[1], w1[1], x0[0], x1[0]);
                mm16(w0[0], x0[1], hd, I0{}); mm16(w1[0], x0[1], hd, I1{}); mm16(w0[0], x1[1], hd, I2{}); mm16(w1[0], x1[1], hd, I3{});
                mm16(w0[0], x0[0], hd, I0{}); mm16(w1[0], x0[0], hd, I1{}); mm16(w0[0], x1[0], hd, I2{}); mm16(w1[0], x1[0], hd, I3{});
            } else head4(w0[0], w1[0], x0[0], x1[0]);
        };
        // GELU of pre-activations 4q .. 4q+3 of hsrc (hidden units tau(4q + e, h)): part 0 forms the table addresses and issues
        // the gathers, part 1 (after the next lgkmcnt(0)) interpolates, multiplies, splits; after q = 1 and q = 3 the finished
        // k-step of hidden planes goes to the hand-off buffer of parity par
        auto gelu0 = [&](auto q_c, const f32x16 &hsrc) {
            constexpr int q = decltype(q_c)::value;
            f32x2 *gt = gtab;                              // (names used only inside asm operands of a generic lambda are not captured)
            const unsigned la = lut_addr;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float hv = hsrc[4 * q + e];
                gvv[e] = hv;                               // (stream units: the table entries carry the power-of-two 1 / scale)
                const float t = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                gfr[e] = __builtin_amdgcn_fractf(t);
                const unsigned idx = (unsigned)t;
                if (kMQAbl & 2) { gt[e][0] = __builtin_bit_cast(float, idx); gt[e][1] = hv; (void)la; }
                else asm volatile("ds_read_b64 %0, %1" : "=v"(gt[e]) : "v"(la + idx * 8u) : "memory");
            }
        };
        auto gelu1 = [&](auto q_c, int par) {
            constexpr int q = decltype(q_c)::value;
            asm volatile("" : "+v"(gtab[0]), "+v"(gtab[1]), "+v"(gtab[2]), "+v"(gtab[3]));   // gathers landed (lgkmcnt(0) before)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float g0 = gvv[2 * p] * fmaf(gfr[2 * p], gtab[2 * p][1], gtab[2 * p][0]);
                const float g1 = gvv[2 * p + 1] * fmaf(gfr[2 * p + 1], gtab[2 * p + 1][1], gtab[2 * p + 1][0]);
                unsigned hi, lo;
                split2p<T, NP>(g0, g1, hi, lo);
                hw[0][2 * (q & 1) + p] = hi; hw[1][2 * (q & 1) + p] = lo;
            }
            if constexpr ((q & 1) == 1) {
                const unsigned a = hand0 + (unsigned)par * (2 * NP * 1024);
                const u32x4 ph = {hw[0][0], hw[0][1], hw[0][2], hw[0][3]};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(ph), "n"((q >> 1) * NP * 1024) : "memory");
                if (NP == 2) {
                    const u32x4 pl = {hw[1][0], hw[1][1], hw[1][2], hw[1][3]};
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(pl), "n"((q >> 1) * NP * 1024 + 1024) : "memory");
                }
            }
        };
        // one fc step: k-steps 8 half .. 8 half + 7 of the tile accumulating in hdst; the GELU of k-step `half` of hsrc's hidden
        // planes (pre-activations 8 half .. + 7) rides in the MFMA shadows
        auto step_fc = [&](auto half_c, f32x16 &hdst, const f32x16 &hsrc, int par, bool with_gelu, bool next_step_has_fc) {
            constexpr int half = decltype(half_c)::value;
            using VN = std::integral_constant<int, (NP == 2 ? 2 : 6)>;
            sync(E0{});
            mark(1);
            chunk_begin(MB{}, I0{}, true);
            if (with_gelu) gelu0(std::integral_constant<int, 2 * half>{}, hsrc);
            fc_mma(wb[0][0], wb[0][kS1], xn[4 * half], xn[8 + 4 * half], hdst, std::integral_constant<bool, half == 0>{});
            pin(VN{});
            mark(2);
            chunk_begin(MB{}, I1{}, true);
            if (with_gelu) gelu1(std::integral_constant<int, 2 * half>{}, par);
            fc_mma(wb[1][0], wb[1][kS1], xn[4 * half + 1], xn[8 + 4 * half + 1], hdst, std::false_type{});
            pin(VN{});
            mark(3);
            chunk_begin(MB{}, I2{}, true);
            if (with_gelu) gelu0(std::integral_constant<int, 2 * half + 1>{}, hsrc);
            fc_mma(wb[0][0], wb[0][kS1], xn[4 * half + 2], xn[8 + 4 * half + 2], hdst, std::false_type{});
            pin(VN{});
            mark(4);
            chunk_begin(MB{}, I3{}, next_step_has_fc);
            if (with_gelu) gelu1(std::integral_constant<int, 2 * half + 1>{}, par);
            fc_mma(wb[1][0], wb[1][kS1], xn[4 * half + 3], xn[8 + 4 * half + 3], hdst, std::false_type{});
            pin(VN{});
        };
        auto tile_fc = [&](f32x16 &hdst, const f32x16 &hsrc, int par, bool with_gelu, bool last_of_block) {
            step_fc(I0{}, hdst, hsrc, par, with_gelu, true);  // (its first chunk starts hdst from zero)
            step_fc(I1{}, hdst, hsrc, par, with_gelu, !last_of_block);
        };
        // GELU of one k-step of hidden planes in a step without MFMAs
        auto gelu_only = [&](auto half_c, const f32x16 &hsrc, int par) {
            constexpr int half = decltype(half_c)::value;
            gelu0(std::integral_constant<int, 2 * half>{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(std::integral_constant<int, 2 * half>{}, par);
            gelu0(std::integral_constant<int, 2 * half + 1>{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(std::integral_constant<int, 2 * half + 1>{}, par);
        };
#pragma unroll
        for (int g = 0; g < 16; g++) { hA[g] = 0.f; hB[g] = 0.f; }

#pragma unroll 1
        for (int k = 0; k < n_mine; k++) {
            // (k == 0: hB is zero, its GELU writes zero hidden planes -- what the consumer's first steps expect)
            const int64_t blk = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
            // x is chunk-major (xt_off): [32-token tile][C / 8 chunks][32 tokens][8 floats]; lane (t, q) owns the 32 bytes of tokens t and 16 + t in chunk
            // 4 kb + q: 16 lanes of a load cover 512 contiguous bytes
            const float *xrow = x + (blk * (32 * NPAIR) + pair * 32) * C + q4 * 256 + t16 * 8;
            f32x4 xr[32];                                  // raw row pieces: xr[2 (8 tg + kb) + hf] = features 32 kb + 8 q + 4 hf .. + 3 of token 16 tg + t
            // ---- step 0: GELU(tile 31), first k-step; row loads ----
            sync(E0{});
            gelu_only(I0{}, hB, 1);
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const float *xq = xrow + (i >> 2) * 1024;  // k-block i / 4 (13-bit immediate offsets: one base per k-block); i % 4 = (tg, hf)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xr[2 * (8 * ((i >> 1) & 1) + (i >> 2)) + (i & 1)]) : "v"(xq), "n"(((i >> 1) & 1) * 512 + (i & 1) * 16) : "memory");
            }
            // ---- step 1: GELU(tile 31), second k-step; rows landed; LayerNorm statistics (two-pass, model.py:19-20) ----
            sync(std::integral_constant<int, 32>{});
            gelu_only(I1{}, hB, 1);
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) : [n] "n"(PWP) : "memory");
            asm volatile("" : "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
            asm volatile("" : "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]), "+v"(xr[20]), "+v"(xr[21]), "+v"(xr[22]), "+v"(xr[23]));
            asm volatile("" : "+v"(xr[24]), "+v"(xr[25]), "+v"(xr[26]), "+v"(xr[27]), "+v"(xr[28]), "+v"(xr[29]), "+v"(xr[30]), "+v"(xr[31]));
            // a token's 256 features sit in the four lanes t + 16 q: v_permlane16_swap folds rows 0|1 and 2|3 of 16 lanes, v_permlane32_swap the halves
            auto fold4 = [&](float v) {
                float a = v, b = v;
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                float c = a + b, d = c;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
                return c + d;
            };
            float mean[2], rstd[2];
#pragma unroll
            for (int tg = 0; tg < 2; tg++) {
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) sm += (xr[16 * tg + i][0] + xr[16 * tg + i][1]) + (xr[16 * tg + i][2] + xr[16 * tg + i][3]);
                mean[tg] = fold4(sm) * (1.0f / (float)C);
                float qv = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float d = xr[16 * tg + i][e] - mean[tg]; qv = fmaf(d, d, qv); }
                rstd[tg] = rsqrtf(fold4(qv) * (1.0f / (float)C) + 1e-5f);
            }
            // ---- steps 2, 3: normalise and split the 16 k-steps ----
            auto norm = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { v0[e] = (xr[2 * ks][e] - mean[ks >> 3]) * rstd[ks >> 3]; v1[e] = (xr[2 * ks + 1][e] - mean[ks >> 3]) * rstd[ks >> 3]; }   // ks = 8 tg + kb
                u32x2 h0, l0, h1, l1;
                split4p<T, NP>(v0, h0, l0);
                split4p<T, NP>(v1, h1, l1);
                xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
                xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
            };
            sync(E0{});                                                                  // step 2
            norm(std::integral_constant<int, 0>{}); norm(std::integral_constant<int, 1>{}); norm(std::integral_constant<int, 2>{}); norm(std::integral_constant<int, 3>{});
            norm(std::integral_constant<int, 4>{}); norm(std::integral_constant<int, 5>{}); norm(std::integral_constant<int, 6>{}); norm(std::integral_constant<int, 7>{});
            sync(E0{});                                                                  // step 3
            norm(std::integral_constant<int, 8>{}); norm(std::integral_constant<int, 9>{}); norm(std::integral_constant<int, 10>{}); norm(std::integral_constant<int, 11>{});
            norm(std::integral_constant<int, 12>{}); norm(std::integral_constant<int, 13>{}); norm(std::integral_constant<int, 14>{}); norm(std::integral_constant<int, 15>{});
            // the first fragments of step 4 (its slot has landed for every wave: step 3's barrier)
            lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]);
            lds_pair(nxt_addr, std::integral_constant<int, 1>{}, wb[0][1]);
            // ---- steps 4 .. 67: c_fc of tiles 0 .. 31, GELU one tile behind ----
            tile_fc(hA, hB, 1, false, false);              // tile 0 (tile 31's GELU ran in steps 0, 1)
            tile_fc(hB, hA, 0, true, false);               // tile 1, GELU(tile 0) -> parity 0
#pragma unroll 1
            for (int t = 2; t < 32; t += 2) {
                tile_fc(hA, hB, 1, true, false);           // even tile, GELU(odd tile before it) -> parity 1
                tile_fc(hB, hA, 0, true, t == 30);         // odd tile, GELU(even tile) -> parity 0
            }
        }
        // ---- drain: GELU of the last block's tile 31 (steps 0, 1); the consumer finishes during steps 2 .. 7 ----
        sync(E0{}); gelu_only(I0{}, hB, 1);
        sync(E0{}); gelu_only(I1{}, hB, 1);
#pragma unroll 1
        for (int s_ = 2; s_ < 2 * kMPPause; s_++) sync(E0{});
    } else {
        // =============================================== consumer ===============================================
        using MB = std::integral_constant<int, 8>;
        f32x16 acc[8];                                     // 32 tokens x 256 output features, swapped layout
        u32x4 hf[2][2];                                    // hidden planes: [k-step kk][plane]
        f32x4 xs[4][4];                                    // residual row pieces in flight (write-back)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) { hf[a][b] = (u32x4){0u, 0u, 0u, 0u}; wb[0][a][b] = (u32x4){0u, 0u, 0u, 0u}; }   // (the very first chunk runs on these)
        // half a chunk: the feature groups w0 (even fg), w1 (odd fg) of the chunk against the hidden planes of ONE token group; slices 2 (fg % 2) + tg of blk
        auto pj_half_b = [&](const u32x4 (&w0)[2], const u32x4 (&w1)[2], const u32x4 (&hb)[2], f32x16 &blk, auto tg_c, auto &&behind) {
            constexpr int tg = decltype(tg_c)::value;
            using S0 = std::integral_constant<int, tg>;
            using S1 = std::integral_constant<int, 2 + tg>;
            if (NP == 2) {
                mm16(w0[1], hb[0], blk, S0{}); behind(I0{}); mm16(w1[1], hb[0], blk, S1{}); behind(I1{});
                mm16(w0[0], hb[1], blk, S0{}); behind(I2{}); mm16(w1[0], hb[1], blk, S1{}); behind(I3{});
                mm16(w0[0], hb[0], blk, S0{}); mm16(w1[0], hb[0], blk, S1{});
            } else {
                mm16(w0[0], hb[0], blk, S0{}); behind(I0{}); mm16(w1[0], hb[0], blk, S1{}); behind(I1{});
            }
        };
        auto pj_half = [&](const u32x4 (&w0)[2], const u32x4 (&w1)[2], const u32x4 (&hb)[2], f32x16 &blk, auto tg_c) {
            pj_half_b(w0, w1, hb, blk, tg_c, [&](auto) {});
        };
        // a whole chunk (both token groups) with its requests placed behind the first MFMAs
        // (pre: requests in front of the chunk's MFMAs; between: between its two halves -- 2 NP fragment reads are younger than anything `pre` requested)
        auto pj_chunk = [&](auto c_c, bool next_has_work, const u32x4 (&w0)[2], const u32x4 (&w1)[2], f32x16 &blk, auto &&pre, auto &&between) {
            if constexpr (PLACED) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pre();
                __builtin_amdgcn_sched_barrier(0);
                pj_half_b(w0, w1, hf[0], blk, I0{}, [&](auto n_c) { __builtin_amdgcn_sched_barrier(0); chunk_read(MB{}, c_c, next_has_work, n_c); });
                between();
                pj_half(w0, w1, hf[1], blk, I1{});
                pin_rest(std::integral_constant<int, (NP == 2 ? 8 : 2)>{});
            } else {
                chunk_begin(MB{}, c_c, next_has_work);
                pre();
                pj_half(w0, w1, hf[0], blk, I0{});
                between();
                pj_half(w0, w1, hf[1], blk, I1{});
                pin(E0{});
            }
        };
        // hidden planes of token group kk (slot kk of the hand-off) of the tile with parity par
        auto load_hidden = [&](auto kk_c, int par) {
            constexpr int kk = decltype(kk_c)::value;
            const unsigned a = hand0 + (unsigned)par * (2 * NP * 1024);
            u32x4 (&hk)[2] = hf[kk];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hk[0]) : "v"(a), "n"(kk * NP * 1024) : "memory");
            if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hk[1]) : "v"(a), "n"(kk * NP * 1024 + 1024) : "memory");
            else hk[1] = hk[0];
        };
        // one c_proj step: feature groups 8 kk .. 8 kk + 7 (two per chunk, accumulators acc[4 kk + chunk]) of the hidden tile with parity par, both token groups.
        // The kk = 0 step requests the tile's SECOND hand-off slot at its top (complete since the step before, a barrier in between) and uses it six MFMAs later;
        // the kk = 1 step requests the first slot of the NEXT tile behind its last use of this tile's first slot
        auto step_pj = [&](auto kk_c, int par, bool next_step_has_pj, bool prefetch_next_tile, auto pending_c) {
            constexpr int kk = decltype(kk_c)::value;
            sync(pending_c);
            mark(1);
            auto none = [&]() {};
            // (LDS requests return in issue order: the second slot's planes are older than the chunk's placed fragment reads)
            pj_chunk(I0{}, true, wb[0][0], wb[0][kS1], acc[4 * kk], [&]() { if (kk == 0) load_hidden(I1{}, par); },
                     [&]() {
                         u32x4 (&h1)[2] = hf[1];           // (names used only inside asm operands of a generic lambda are not captured)
                         if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(h1[0]), "+v"(h1[1]) : [n] "n"(PLACED ? 2 * NP : 0) : "memory");
                     });
            mark(2);
            pj_chunk(I1{}, true, wb[1][0], wb[1][kS1], acc[4 * kk + 1], none, none);
            mark(3);
            pj_chunk(I2{}, true, wb[0][0], wb[0][kS1], acc[4 * kk + 2], none, none);
            mark(4);
            pj_chunk(I3{}, next_step_has_pj, wb[1][0], wb[1][kS1], acc[4 * kk + 3], none, [&]() { if (kk == 1 && prefetch_next_tile) load_hidden(I0{}, par ^ 1); });
        };

        // steps 0 .. 7 of a period for the consumer: the block blk_prev is finished (c_proj of its tiles 30, 31, then the
        // residual add + store, acc = 0)
        auto finish_block = [&](int64_t blk_prev) {
            // chunk-major: the quad (fg, tg) of lane (t, q) = features 16 fg + 4 q .. + 3 of token 16 tg + t sits in chunk 2 fg + q / 2, half q % 2; "tile" j =
            // feature groups 2 j, 2 j + 1, piece gq = 2 (fg % 2) + tg
            float *xrow = x + (blk_prev * (32 * NPAIR) + pair * 32) * C + (q4 >> 1) * 256 + t16 * 8 + 4 * (q4 & 1);
            auto ld = [&](auto j_c) {                      // residual pieces of output tile j -> xs[j % 4]
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j % 4];
                float *xp = xrow + j * 1024;
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xj[gq]) : "v"(xp), "n"((gq >> 1) * 2048 + (gq & 1) * 512) : "memory");
            };
            auto st = [&](auto j_c, auto younger_c) {      // x = x + acc[j] * inv2 for output tile j; YOUNGER = operations issued after its loads
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j % 4];
                float *xp = xrow + j * 1024;
                asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]) : [n] "n"(decltype(younger_c)::value) : "memory");
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = fmaf(acc[j][4 * gq + e], inv2, xj[gq][e]);
                    // (s_nop: a store of more than 8 bytes reads its data registers after issue; hipcc pads a VALU write of them for
                    //  its own stores, but it cannot see through inline asm -- without this the next piece's FMAs clobbered the data)
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(xp), "v"(o), "n"((gq >> 1) * 2048 + (gq & 1) * 512) : "memory");
                }
#pragma unroll
                for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
            };
            using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
            using J4 = std::integral_constant<int, 4>; using J5 = std::integral_constant<int, 5>; using J6 = std::integral_constant<int, 6>; using J7 = std::integral_constant<int, 7>;
            // ---- steps 0 .. 3: c_proj of the previous block's tiles 30 (parity 0) and 31 (parity 1) ----
            // (the first fragments and the first k-step of tile 30's planes were requested in step 67)
            step_pj(I0{}, 0, true, false, E0{});
            step_pj(I1{}, 0, true, true, E0{});
            step_pj(I0{}, 1, true, false, E0{});
            step_pj(I1{}, 1, false, false, E0{});          // step 3: every output tile is final after it
            ld(J0{}); ld(J1{});
            // ---- steps 4 .. 7: residual add + store, two output tiles per step, loads one step ahead ----
            // vector-memory operations of a consumer per step: 8 loads (two tiles) | 8 stores (two tiles)   (no ring pieces)
            constexpr int PW = ISSUE_ALL ? PWP : 0;        // ring pieces this consumer issues at the top of a step
            sync(std::integral_constant<int, 8>{});                                       // step 4 (pending: L0 L1)
            ld(J2{}); ld(J3{});
            st(J0{}, std::integral_constant<int, PW + 8>{}); st(J1{}, std::integral_constant<int, PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                                      // step 5 (pending: L2 L3 S0 S1)
            ld(J4{}); ld(J5{});
            st(J2{}, std::integral_constant<int, 8 + PW + 8>{}); st(J3{}, std::integral_constant<int, 8 + PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                                      // step 6
            ld(J6{}); ld(J7{});
            st(J4{}, std::integral_constant<int, 8 + PW + 8>{}); st(J5{}, std::integral_constant<int, 8 + PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                                      // step 7 (pending: L6 L7 S4 S5)
            st(J6{}, std::integral_constant<int, 8 + PW>{}); st(J7{}, std::integral_constant<int, 8 + PW + 4>{});
        };
#pragma unroll 1
        for (int k = 0; k < n_mine; k++) {
            // k == 0: nothing to finish -- the same sequence runs on this block's own rows with acc == 0 and zero hidden planes
            // (x + 0 is written back unchanged), which keeps the loop free of branches and the step / wait counts uniform
            finish_block((int64_t)blockIdx.x + (int64_t)(k > 0 ? k - 1 : 0) * gridDim.x);
            // the first fragments of step 8 and the first k-step of tile 0's hidden planes (complete since step 6)
            lds_pair(nxt_addr, std::integral_constant<int, 8>{}, wb[0][0]);
            lds_pair(nxt_addr, std::integral_constant<int, 9>{}, wb[0][1]);
            load_hidden(I0{}, 0);
            // ---- steps 8 .. 67: c_proj of tiles 0 .. 29 ----
            step_pj(I0{}, 0, true, false, std::integral_constant<int, 8>{});             // step 8 (pending: S6 S7)
            step_pj(I1{}, 0, true, true, E0{});
            step_pj(I0{}, 1, true, false, E0{});
            step_pj(I1{}, 1, true, true, E0{});
#pragma unroll 1
            for (int t = 2; t < 30; t += 2) {
                step_pj(I0{}, 0, true, false, E0{});
                step_pj(I1{}, 0, true, true, E0{});
                step_pj(I0{}, 1, true, false, E0{});
                step_pj(I1{}, 1, true, true, E0{});        // t + 1 == 29: the next tile is tile 30, worked on in the next period
            }
        }
        finish_block((int64_t)blockIdx.x + (int64_t)(n_mine - 1) * gridDim.x);      // drain
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no direct-to-LDS load may outlive the workgroup
    if constexpr (STAMPS != 0) {
        if ((wave == 0 || wave == NPAIR) && lane == 0) {
            unsigned long long *o = stamps + ((size_t)blockIdx.x * 2 + (wave >= NPAIR ? 1 : 0)) * 4;
            if constexpr (STAMPS == 3) { o[0] = t_ph[0]; o[1] = t_ph[1]; o[2] = t_ph[2] + t_ph[3] + t_ph[4]; o[3] = t_ph[5]; }
            else { o[0] = t_in[0]; o[1] = t_in[1]; o[2] = __builtin_readcyclecounter(); o[3] = STAMPS == 2 ? t_sync : wall_clock64(); }
        }
    }
}

}  // namespace fastk
}  // namespace mgpt
