// gpt_kernels_c160p.h -- persistent, role-specialised MLP block for n_embd = 160 (MAPF-GPT-2M), gfx950.
//
// Where it is used (round 4): SMALL launches only (<= 128 rows: one environment, BASELINE cfg1).  Built in round 3 as an experiment for
// the big-batch path, where it ties with mlp_fused_kernel (5.17 vs 5.1 ms per 16 384 rows: both run at the package power limit).  A small
// launch is not power-limited but latency-bound: mlp_fused_kernel spreads cfg1's 8192 tokens as 2-wave workgroups (one wave per SIMD,
// 52 us per launch); here a 128-token block gets two waves per SIMD -- the producer's c_fc / GELU under the consumer's c_proj.
// NFOLD > 0: the block's rows are x + the NFOLD partial sums a head-parallel attn_block_kernel left (fold, fold_stride floats apart,
// x's layout), added in index order by the producer and written back before anything else reads them; the consumer's residual
// loads bypass the vector L1 (sc1), so that they see the producer's write-back whatever this CU's L1 holds.
//
// The C = 256 kernel of gpt_kernels_c256p.h re-cut for C = 160 (CT = 5 feature tiles, KS = 10 k-steps, 20 hidden tiles):
// a PERSISTENT workgroup (grid = number of CUs, blocks of 128 tokens dealt round-robin) of 8 waves in two roles,
//     producer p (waves 0-3)   operand planes of 32 tokens (80 registers): LayerNorm, c_fc MFMAs, GELU (Phi table), hidden
//                              planes of every 32-unit tile handed to its consumer through LDS (4 KiB per tile);
//     consumer p (waves 4-7)   the 32 x 160 output accumulators (80 registers): c_proj MFMAs, residual add and store.
// One stream step = ONE hidden tile for each role: 10 c_fc fragment pairs (the tile's 10 k-steps, ln_2.weight folded in) +
// 10 c_proj fragment pairs (5 output tiles x 2 k-steps of a hidden tile) = 40 KiB in the split mode, 30 MFMAs per wave,
// five chunks of two pairs.  Two LDS slots (80 KiB + 24 KiB Phi table + 32 KiB hand-off = 136 KiB): the step after the
// current one is in flight while this one is consumed; one s_barrier per step.  The stream is CYCLIC, period 22:
//     period step R   producer                                     consumer
//     0               GELU(tile 19 of the previous block); x rows  c_proj(tile 18 of the previous block)
//     1               rows landed: LayerNorm, split                 c_proj(tile 19 of the previous block)
//     2               c_fc(tile 0)                                  residual rows of the previous block requested
//     3               c_fc(tile 1), GELU(0)                         residual add + store, acc = 0
//     4 .. 21         c_fc(tile R - 2), GELU(R - 3)                 c_proj(tile R - 4)
// Every LDS / global access inside the block loop is inline asm with hand-counted s_waitcnt (see gpt_kernels_c256p.h).
// x is chunk-major (xt_off).  Results per token do not depend on the grid or on the block a token falls in.
// (Round 4: touching every line of the stream up front -- all first-touch misses of the block in flight at once -- makes the small
//  launch SLOWER, 0.50 against 0.46 ms per cfg1 step under the profiler: the 1.6 us per step are not L2 misses.)
#pragma once
#include "gpt_kernels_c256p.h"

namespace mgpt {
namespace fastk {

constexpr int kM5Period = 22;                   // stream steps per block
constexpr int kM5Slots = 2;                     // LDS ring depth (steps)

// NPAIR = producer / consumer wave pairs of a workgroup = 32-token tiles of a block: 4 (128-token blocks, two waves per SIMD), 2 (64-token
// blocks, one wave per SIMD) or 1 (32-token blocks) for launches so small that 128-token blocks would leave CUs idle: one environment
// (cfg1, 8192 tokens) 36.7 -> ~30 us per launch with NPAIR = 2, a little less with 1 (a step is then latency, not matrix-pipe time)
template <int NP, int NPAIR = 4>
constexpr int kM5Lds = kM5Slots * 20 * NP * 1024 + kGeluLutN * 8 + NPAIR * 2 * 2 * NP * 1024;   // ring | Phi table | hidden hand-off

// weight stream: [period step R][pair ms][plane][lane][8]; pairs 0-9 = c_fc k-steps (gain folded in), 10 + 2 j + kk = c_proj
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp160p_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                           const float *__restrict__ gain, uint16_t *__restrict__ out,
                                                           float scale1, float scale2)
{
    constexpr int C = 160;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (step, pair, lane)
    if (gid >= (int64_t)kM5Period * 20 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) % 20), R = (int)((gid >> 6) / 20);
    const int i = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.f;
    if (ms < 10) {                                                        // c_fc(tile t), k-step ms: A rows = hidden units, k-slots = features
        if (R >= 2) {
            const int t = R - 2, ks = ms;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int g = 8 * (ks & 1) + e;
                const int feat = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
                v[e] = fc_w[(size_t)(32 * t + i) * C + feat] * gain[feat] * scale1;   // LayerNorm weight folded in (model.py:19-20, 86)
            }
        }
    } else {                                                              // c_proj: hidden tile t, output tile j, k-step kk
        const int t = (R < 2) ? 18 + R : (R >= 4 ? R - 4 : -1);           // the previous block's last two tiles come first
        if (t >= 0) {
            const int j = (ms - 10) >> 1, kk = (ms - 10) & 1;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int g = 8 * kk + e;
                const int u = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
                v[e] = pj_w[(size_t)(32 * j + i) * (4 * C) + u] * scale2;
            }
        }
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)R * 20 + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

template <class T, int NP, int NFOLD = 0, int NPAIR = 4>
__global__ __launch_bounds__(128 * NPAIR, 2) void mlp160p_kernel(float *__restrict__ x, const uint16_t *__restrict__ wstream, float inv1,
                                                         float inv2, const float2 *__restrict__ gelu_lut, int n_blocks,
                                                         const float *__restrict__ fold = nullptr, int64_t fold_stride = 0)
{
    constexpr int C = 160, CT = 5, KS = 10;
    constexpr int MS = 20;                                 // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;                   // bytes per stream step
    constexpr int NSLOT = kM5Slots;
    static_assert(NPAIR == 4 || NPAIR == 2 || NPAIR == 1, "wave pairs");
    constexpr int PWP = MS * NP / NPAIR;                   // direct-to-LDS pieces per PRODUCER wave per step (the consumers issue none)
    constexpr int BLK = 32 * NPAIR;                        // tokens per block
    constexpr int LUT_BYTES = kGeluLutN * 8;
    constexpr int NM = (NP == 2 ? 6 : 2);                  // MFMAs per chunk (two fragment pairs)
    constexpr int NX = C / 8;                              // 16-byte row pieces per lane (chunks of the chunk-major row)
    static_assert(NSLOT == 2, "the counted waits below assume that exactly the next step's pieces are in flight");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPAIR;                    // wave-uniform
    const int pair = producer ? wave : wave - NPAIR;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned lut_addr = (unsigned)(size_t)smem + NSLOT * STEP;
    const unsigned hand0 = lut_addr + LUT_BYTES + (unsigned)pair * (2 * 2 * NP * 1024) + lane16;    // this pair's hand-off, this lane
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(pair * PWP) * 1024 + lane16;
    const int n_mine = n_blocks > (int)blockIdx.x ? (n_blocks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_mine == 0) return;

    // ---- ring: slot of step R = R & 1.  Top of step R: this wave's pieces of step R (issued at the top of step R - 1) have
    //      landed, every LDS access of this wave is done, barrier; then the slot of step R - 1 is refilled with step R + 1.
    //      PENDING = vector-memory operations of this wave issued after those pieces (they retire in issue order). ----
    int r_issue = 0;                                       // stream step (mod period) of the next DMA
    int slot_cur = 0;
    unsigned cur_addr = 0;
    auto issue = [&](int slot) {
        if (producer) {                                    // wave-uniform
            const unsigned char *src = wbase + (size_t)r_issue * STEP;
            unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(pair * PWP) * 1024;
#pragma unroll
            for (int i = 0; i < PWP; i++) dma_piece(src + (i >> 2) * 4096, dst + (i >> 2) * 4096, std::integral_constant<int, 0>{}, i & 3);
        }
        r_issue = r_issue + 1 == kM5Period ? 0 : r_issue + 1;
    };
    {   // Phi table -> LDS (24 pieces of 1 KiB, 3 per wave); older than every ring piece
        constexpr int LW = LUT_BYTES / (2 * NPAIR);        // bytes of the table per wave
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut) + (size_t)wave * LW + lane16;
#pragma unroll
        for (int i = 0; i < LW / 1024; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + i * 1024), (lds_void_t *)(smem + NSLOT * STEP + wave * LW + i * 1024), 16, 0, 0);
    }
    issue(0);
    if (!producer) {                                       // hidden hand-off starts as zeros (the first block has no predecessor)
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 2 * 2 * NP; i++) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(hand0), "v"(z), "n"(i * 1024) : "memory");
    }
    auto sync = [&](auto pending_c) {
        vm_wait<decltype(pending_c)::value>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(slot_cur ^ 1);                               // always: the stream is cyclic
        cur_addr = lds0 + (unsigned)slot_cur * STEP;
        slot_cur ^= 1;
    };
    using E0 = std::integral_constant<int, 0>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    u32x4 wb[2][2][2];                                     // weight fragments [set = chunk & 1][pair of the chunk][plane]
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        else dst[1] = dst[0];
    };
    // the first pairs of a step, right after its barrier
    auto step_first = [&](auto mb_c) {
        constexpr int MB = decltype(mb_c)::value;
        lds_pair(cur_addr, std::integral_constant<int, MB>{}, wb[0][0]);
        lds_pair(cur_addr, std::integral_constant<int, MB + 1>{}, wb[0][1]);
    };
    // chunk c (0 .. 4) of a step works on pairs MB + 2c, MB + 2c + 1 (set c & 1), requested one chunk earlier; it requests the
    // pairs of the next chunk in front of its MFMAs
    auto chunk_begin = [&](auto mb_c, auto c_c) {
        constexpr int MB = decltype(mb_c)::value, c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 4) { lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 2>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 3>{}, wb[(c + 1) & 1][1]); }
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
    // two fragment pairs (two k-steps) into ONE accumulator: weight lo plane first (lo.hi, hi.lo, hi.hi)
    auto mma2 = [&](const u32x4 (&wa)[2], const u32x4 (&xa)[2], const u32x4 (&wc)[2], const u32x4 (&xc)[2], f32x16 &hd) {
        if (NP == 2) {
            hd = T::mfma(wa[1], xa[0], hd); hd = T::mfma(wc[1], xc[0], hd);
            hd = T::mfma(wa[0], xa[1], hd); hd = T::mfma(wc[0], xc[1], hd);
        }
        hd = T::mfma(wa[0], xa[0], hd); hd = T::mfma(wc[0], xc[0], hd);
    };

    if (producer) {
        // =============================================== producer ===============================================
        using MB = std::integral_constant<int, 0>;         // first pair of this role in a step
        u32x4 xn[KS][2];                                   // operand planes of this lane's token: [k-step][plane]
        f32x16 hA, hB;                                     // pre-activations of the even / odd hidden tile
        const float lut_scale = inv1 * kGeluLutScale;
        float gvv[4], gfr[4];
        f32x2 gtab[4];
        unsigned hw[2][4];                                 // hidden words of one k-step: [plane][word]
        // GELU of pre-activations 4q .. 4q+3 of hsrc: part 0 forms the table addresses and issues the gathers, part 1 (after the
        // next lgkmcnt(0)) interpolates, multiplies, splits; after q = 1 and q = 3 the finished k-step of hidden planes goes to
        // the hand-off buffer of parity par
        auto gelu0 = [&](auto q_c, const f32x16 &hsrc) {
            constexpr int q = decltype(q_c)::value;
            f32x2 *gt = gtab;                              // (names used only inside asm operands of a generic lambda are not captured)
            const unsigned la = lut_addr;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float hv = hsrc[4 * q + e];
                gvv[e] = hv * inv1;
                const float t = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                gfr[e] = __builtin_amdgcn_fractf(t);
                const unsigned idx = (unsigned)t;
                asm volatile("ds_read_b64 %0, %1" : "=v"(gt[e]) : "v"(la + idx * 8u) : "memory");
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
        // one step = c_fc of one hidden tile (10 k-steps) accumulating in hdst; the GELU of the tile before it (hsrc, parity
        // par) rides in the MFMA shadows
        auto tile_fc = [&](f32x16 &hdst, const f32x16 &hsrc, int par, bool with_gelu) {
            using VN = std::integral_constant<int, (NP == 2 ? 6 : 16)>;
#pragma unroll
            for (int g = 0; g < 16; g++) hdst[g] = 0.f;
            sync(E0{});
            step_first(MB{});
            chunk_begin(MB{}, I0{});
            if (with_gelu) gelu0(I0{}, hsrc);
            mma2(wb[0][0], xn[0], wb[0][1], xn[1], hdst);
            pin(VN{});
            chunk_begin(MB{}, I1{});
            if (with_gelu) { gelu1(I0{}, par); gelu0(I1{}, hsrc); }
            mma2(wb[1][0], xn[2], wb[1][1], xn[3], hdst);
            pin(VN{});
            chunk_begin(MB{}, I2{});
            if (with_gelu) { gelu1(I1{}, par); gelu0(I2{}, hsrc); }
            mma2(wb[0][0], xn[4], wb[0][1], xn[5], hdst);
            pin(VN{});
            chunk_begin(MB{}, I3{});
            if (with_gelu) { gelu1(I2{}, par); gelu0(I3{}, hsrc); }
            mma2(wb[1][0], xn[6], wb[1][1], xn[7], hdst);
            pin(VN{});
            chunk_begin(MB{}, I4{});
            if (with_gelu) gelu1(I3{}, par);
            mma2(wb[0][0], xn[8], wb[0][1], xn[9], hdst);
            pin(VN{});
        };
        // GELU of a whole tile in a step without MFMAs
        auto gelu_only = [&](const f32x16 &hsrc, int par) {
            gelu0(I0{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(I0{}, par);
            gelu0(I1{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(I1{}, par);
            gelu0(I2{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(I2{}, par);
            gelu0(I3{}, hsrc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gelu1(I3{}, par);
        };
#pragma unroll
        for (int g = 0; g < 16; g++) { hA[g] = 0.f; hB[g] = 0.f; }

#pragma unroll 1
        for (int k = 0; k < n_mine; k++) {
            // (k == 0: hB is zero, its GELU writes zero hidden planes -- what the consumer's first steps expect)
            const int64_t blk = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
            float *xrow = x + (blk * BLK + pair * 32) * C + r * 8 + 4 * h;            // chunk-major: chunk c at xrow + c * 256
            f32x4 xr[NX];                                  // raw row pieces: xr[c] = features 8 c + 4 h .. + 3
            // ---- step 0: GELU(tile 19 of the previous block); row loads ----
            sync(E0{});
            gelu_only(hB, 1);
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const float *xq = xrow + (i >> 2) * 1024;  // (13-bit immediate offsets: four chunks per address)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xr[i]) : "v"(xq), "n"((i & 3) * 1024) : "memory");
            }
            if constexpr (NFOLD > 0) {
                // small launches: row = x + partial sums, in index order; one memory round trip per partial sum (its 20 pieces fly together)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) : : "memory");
                asm volatile("" : "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
                asm volatile("" : "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]));
                const float *frow = fold + (xrow - x);
#pragma unroll 1
                for (int p = 0; p < NFOLD; p++, frow += fold_stride) {
                    f32x4 tp[NX];
#pragma unroll
                    for (int i = 0; i < NX; i++) {
                        const float *fq = frow + (i >> 2) * 1024;
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(tp[i]) : "v"(fq), "n"((i & 3) * 1024) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tp[0]), "+v"(tp[1]), "+v"(tp[2]), "+v"(tp[3]), "+v"(tp[4]), "+v"(tp[5]), "+v"(tp[6]), "+v"(tp[7]) : : "memory");
                    asm volatile("" : "+v"(tp[8]), "+v"(tp[9]), "+v"(tp[10]), "+v"(tp[11]), "+v"(tp[12]), "+v"(tp[13]), "+v"(tp[14]), "+v"(tp[15]));
                    asm volatile("" : "+v"(tp[16]), "+v"(tp[17]), "+v"(tp[18]), "+v"(tp[19]));
#pragma unroll
                    for (int i = 0; i < NX; i++) xr[i] += tp[i];
                }
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    float *xq = xrow + (i >> 2) * 1024;
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(xq), "v"(xr[i]), "n"((i & 3) * 1024) : "memory");
                }
            }
            // ---- step 1: rows landed; LayerNorm (two-pass, model.py:19-20), normalise, split ----
            // (NFOLD: everything before is complete but the NX write-back stores, which are younger than the pieces waited for as well)
            sync(std::integral_constant<int, NX>{});
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) : [n] "n"(PWP) : "memory");
            asm volatile("" : "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
            asm volatile("" : "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]));
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++) s += (xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]);
            {   // the other half of the token sits in lane r + 32 (r - 32): v_permlane32_swap (VALU, no LDS traffic)
                float a = s, b = s;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                s = a + b;
            }
            const float mean = s * (1.0f / (float)C);
            float qv = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) { const float d = xr[i][e] - mean; xr[i][e] = d; qv = fmaf(d, d, qv); }   // (the centred row is kept)
            {
                float a = qv, b = qv;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                qv = a + b;
            }
            const float rstd = rsqrtf(qv * (1.0f / (float)C) + 1e-5f);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { v0[e] = xr[2 * ks][e] * rstd; v1[e] = xr[2 * ks + 1][e] * rstd; }
                u32x2 h0, l0, h1, l1;
                split4p<T, NP>(v0, h0, l0);
                split4p<T, NP>(v1, h1, l1);
                xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
                xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
            }
            // ---- steps 2 .. 21: c_fc of tiles 0 .. 19, GELU one tile behind ----
            tile_fc(hA, hB, 1, false);                     // tile 0 (tile 19's GELU ran in step 0)
            tile_fc(hB, hA, 0, true);                      // tile 1, GELU(tile 0) -> parity 0
#pragma unroll 1
            for (int t = 2; t < 20; t += 2) {
                tile_fc(hA, hB, 1, true);                  // even tile, GELU(odd tile before it) -> parity 1
                tile_fc(hB, hA, 0, true);                  // odd tile, GELU(even tile) -> parity 0
            }
        }
        // ---- drain: GELU of the last block's tile 19 (step 0); the consumer finishes during steps 1 .. 3 ----
        sync(E0{}); gelu_only(hB, 1);
        sync(E0{}); sync(E0{}); sync(E0{});
    } else {
        // =============================================== consumer ===============================================
        using MB = std::integral_constant<int, 10>;
        f32x16 acc[CT];                                    // 32 tokens x 160 output features, swapped layout
        u32x4 hf[2][2];                                    // hidden planes: [k-step kk][plane]
        f32x4 xs[CT][4];                                   // residual row pieces in flight (write-back)
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
        // hidden planes of the tile with parity par: both k-steps
        auto load_hidden = [&](int par) {
            const unsigned a = hand0 + (unsigned)par * (2 * NP * 1024);
            u32x4 (&h0)[2] = hf[0];
            u32x4 (&h1)[2] = hf[1];
            asm volatile("ds_read_b128 %0, %1" : "=v"(h0[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(h1[0]) : "v"(a), "n"(NP * 1024) : "memory");
            if (NP == 2) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(h0[1]) : "v"(a), "n"(1024) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(h1[1]) : "v"(a), "n"(NP * 1024 + 1024) : "memory");
            } else { h0[1] = h0[0]; h1[1] = h1[0]; }
        };
        // one c_proj step: the hidden tile with parity par against the 5 output tiles (chunk c = output tile c, both k-steps)
        auto step_pj = [&](int par, auto pending_c) {
            sync(pending_c);
            step_first(MB{});
            load_hidden(par);
            chunk_begin(MB{}, I0{});
            mma2(wb[0][0], hf[0], wb[0][1], hf[1], acc[0]);
            pin(E0{});
            chunk_begin(MB{}, I1{});
            mma2(wb[1][0], hf[0], wb[1][1], hf[1], acc[1]);
            pin(E0{});
            chunk_begin(MB{}, I2{});
            mma2(wb[0][0], hf[0], wb[0][1], hf[1], acc[2]);
            pin(E0{});
            chunk_begin(MB{}, I3{});
            mma2(wb[1][0], hf[0], wb[1][1], hf[1], acc[3]);
            pin(E0{});
            chunk_begin(MB{}, I4{});
            mma2(wb[0][0], hf[0], wb[0][1], hf[1], acc[4]);
            pin(E0{});
        };
        // steps 0 .. 3 of a period for the consumer: the block blk_prev is finished (c_proj of its tiles 18, 19, then the
        // residual add + store, acc = 0)
        auto finish_block = [&](int64_t blk_prev, auto first_pending_c) {
            float *xrow = x + (blk_prev * BLK + pair * 32) * C + r * 8 + 4 * h;    // chunk-major, as in the producer
            step_pj(0, first_pending_c);                   // step 0: tile 18
            step_pj(1, E0{});                              // step 1: tile 19; every output tile is final after it
            sync(E0{});                                    // step 2: the residual rows are requested ...
#pragma unroll
            for (int j = 0; j < CT; j++) {
                f32x4 (&xj)[4] = xs[j];
                float *xp = xrow + j * 1024;
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(xj[gq]) : "v"(xp), "n"(gq * 1024) : "memory");
            }
            sync(std::integral_constant<int, 4 * CT>{});   // step 3: ... and added and stored (in-order retirement: tile j's loads are
                                                           // followed by 4 (CT - 1 - j) loads and 4 j stores = 16 younger operations)
#pragma unroll
            for (int j = 0; j < CT; j++) {
                f32x4 (&xj)[4] = xs[j];
                float *xp = xrow + j * 1024;
                asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]) : [n] "n"(4 * (CT - 1)) : "memory");
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = fmaf(acc[j][4 * gq + e], inv2, xj[gq][e]);
                    // (s_nop: a store of more than 8 bytes reads its data registers after issue; see gpt_kernels_c256p.h)
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(xp), "v"(o), "n"(gq * 1024) : "memory");
                }
#pragma unroll
                for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
            }
        };
#pragma unroll 1
        for (int k = 0; k < n_mine; k++) {
            // k == 0: nothing to finish -- the same sequence runs on this block's own rows with acc == 0 and zero hidden planes
            // (x + 0 is written back unchanged), which keeps the loop free of branches and the step / wait counts uniform.
            // (pending at its first step: the 18th c_proj step of the block before left nothing in flight)
            finish_block((int64_t)blockIdx.x + (int64_t)(k > 0 ? k - 1 : 0) * gridDim.x, E0{});
            // ---- steps 4 .. 21: c_proj of tiles 0 .. 17 ----
            step_pj(0, std::integral_constant<int, 4 * CT>{});                           // step 4 (pending: the 20 stores)
            step_pj(1, E0{});
#pragma unroll 1
            for (int t = 2; t < 18; t += 2) {
                step_pj(0, E0{});
                step_pj(1, E0{});
            }
        }
        finish_block((int64_t)blockIdx.x + (int64_t)(n_mine - 1) * gridDim.x, E0{});      // drain
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no direct-to-LDS load may outlive the workgroup
}

}  // namespace fastk
}  // namespace mgpt
