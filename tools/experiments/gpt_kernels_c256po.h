// gpt_kernels_c256po.h -- mlp256p_kernel (gpt_kernels_c256p.h) with the attention output projection fused in front:
//     x <- x + y . c_proj^T (model.py:71, 102);  x <- x + c_proj(GELU(c_fc(LayerNorm(x)))) (model.py:84-89, 103)
// in ONE persistent kernel, so that the HBM-bound out-projection GEMM (gemm_pk_kernel<EPI_RESID>: 3.2 GB per 4096-row launch
// at 5.6 TB/s with the matrix pipe idle, 14 ms of a cfg3 step) disappears: the producer wave accumulates y . W^T for its 32
// tokens in the 128 registers that afterwards hold the rows LayerNorm reads (16 stream steps of 8 fragment pairs, y planes
// straight from the attention kernel's packed-fragment output as B operands, two k-steps ahead), adds the residual rows,
// stores them (the consumer's final residual add reads them back ~70 steps later) and goes on exactly as in mlp256p_kernel;
// the consumer keeps working on the previous block during the first 8 of those steps and waits during the rest.
//     period step R   producer                                              consumer
//     0 .. 15         y . c_proj^T (k-step R); GELU(tile 31 of block b-1)   c_proj(tiles 30, 31) | write-back of block b-1 | -
//     16 .. 21        x rows in, + , store; LayerNorm; split                -
//     22 .. 85        c_fc(tile t = (R - 22) / 2), GELU(t - 1)              c_proj(tile t - 2) from step 26
#pragma once
#include "../../mapf_gpt_amd/csrc/gpt_kernels_c256p.h"

namespace mgpt {
namespace fastk {

constexpr int kMQOut = 16;                      // out-projection steps (one k-step of y each: 8 output tiles)
constexpr int kMQLn = 6;                        // residual add + store (4), LayerNorm statistics, normalise + split (2)
constexpr int kMQR0 = kMQOut + kMQLn;           // period step of the first c_fc step
constexpr int kMQPause = 4;                     // the consumer's write-back steps
constexpr int kMQPeriod = kMQR0 + 64;           // stream steps per block
constexpr int kMQSlots = 3;                     // LDS ring depth (steps)

template <int NP>
constexpr int kMQLds = kMQSlots * 16 * NP * 1024 + kGeluLutN * 8 + 4 * 2 * 2 * NP * 1024;   // ring | Phi table | hidden hand-off

// weight stream: [period step R][pair ms][plane][lane][8]; pairs 0-7 = the producer's (attention c_proj in steps 0 .. 15,
// c_fc with the gain folded in from step kMQR0 on), 8-15 = the consumer's (MLP c_proj)
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp256po_kernel(const float *__restrict__ ao_w, const float *__restrict__ fc_w,
                                                            const float *__restrict__ pj_w, const float *__restrict__ gain,
                                                            uint16_t *__restrict__ out, float scale_o, float scale1, float scale2)
{
    constexpr int C = 256;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (step, pair, lane)
    if (gid >= (int64_t)kMQPeriod * 16 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 15), R = (int)(gid >> 10);
    const int i = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.f;
    if (ms < 8) {
        if (R < kMQOut) {                                                 // attention c_proj (model.py:71), k-step R, output tile ms:
#pragma unroll                                                            // natural k order = the y planes' (pk_off)
            for (int e = 0; e < 8; e++) v[e] = ao_w[(size_t)(32 * ms + i) * C + 16 * R + 8 * h + e] * scale_o;
        } else if (R >= kMQR0) {                                          // c_fc(tile t): A rows = hidden units, k-slots = features
            const int rr = R - kMQR0, t = rr >> 1, ks = 8 * (rr & 1) + ms;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int g = 8 * (ks & 1) + e;
                const int feat = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
                v[e] = fc_w[(size_t)(32 * t + i) * C + feat] * gain[feat] * scale1;   // LayerNorm weight folded in (model.py:19-20, 86)
            }
        }
    } else {                                                              // c_proj(tile t), k-step kk: A rows = output features of tile j
        int t = -1, kk = 0;
        if (R < 4) { t = 30 + (R >> 1); kk = R & 1; }                     // the previous block's last two tiles
        else if (R >= kMQR0 + 4) { t = (R - kMQR0 - 4) >> 1; kk = (R - kMQR0 - 4) & 1; }
        if (t >= 0) {
            const int j = ms - 8;
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
    uint16_t *dst = out + (((size_t)R * 16 + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// STAMPS (tools/check_mlp256p.hip only): 1 = wave 0 and wave 4 of every workgroup leave s_memtime / s_memrealtime at entry and
// exit; 2 = cycles spent in wait + barrier instead of the exit wall clock.
template <class T, int NP, int STAMPS = 0>
__global__ __launch_bounds__(512, 2) void mlp256po_kernel(float *__restrict__ x, const uint16_t *__restrict__ y,
                                                          const uint16_t *__restrict__ wstream, float inv_o, float inv1, float inv2,
                                                          const float2 *__restrict__ gelu_lut, int n_blocks,
                                                          unsigned long long *stamps = nullptr)
{
    constexpr int C = 256;
    constexpr int MS = 16;                                 // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;                   // bytes per stream step
    constexpr int NSLOT = kMQSlots;
    constexpr int PW = MS * NP / 8;                        // direct-to-LDS pieces per wave per step
    constexpr int LUT_BYTES = kGeluLutN * 8;
    constexpr int NM = (NP == 2 ? 6 : 2);                  // MFMAs per chunk (two fragment pairs)
    static_assert(NSLOT == 3, "the counted waits below assume that exactly the next step's pieces are in flight");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;                        // wave-uniform
    const int pair = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned lut_addr = (unsigned)(size_t)smem + NSLOT * STEP;
    const unsigned hand0 = lut_addr + LUT_BYTES + (unsigned)pair * (2 * 2 * NP * 1024) + lane16;    // this pair's hand-off, this lane
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(wave * PW) * 1024 + lane16;
    const int n_mine = n_blocks > (int)blockIdx.x ? (n_blocks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    unsigned long long t_in[2] = {0, 0}, t_sync = 0;
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
        const unsigned char *src = wbase + (size_t)r_issue * STEP;
        unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(wave * PW) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++) dma_piece(src, dst, std::integral_constant<int, 0>{}, i);
        r_issue = r_issue + 1 == kMQPeriod ? 0 : r_issue + 1;
    };
    {   // Phi table -> LDS (24 pieces of 1 KiB, 3 per wave); older than every ring piece
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut) + (size_t)wave * (LUT_BYTES / 8) + lane16;
#pragma unroll
        for (int i = 0; i < LUT_BYTES / 8192; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + i * 1024), (lds_void_t *)(smem + NSLOT * STEP + wave * (LUT_BYTES / 8) + i * 1024), 16, 0, 0);
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
        vm_wait<decltype(pending_c)::value>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMPS == 2) t_sync += __builtin_readcyclecounter() - t0;
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
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(slot_addr), "n"(ms * NP * 1024) : "memory");
        if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(slot_addr), "n"((ms * NP + 1) * 1024) : "memory");
        else dst[1] = dst[0];
    };
    // chunk c (0 .. 3) of a step works on pairs MB + 2c, MB + 2c + 1 (set c & 1), requested one chunk earlier; it requests the
    // pairs of the next chunk (chunk 3: the first pairs of the next step, whose slot has landed) in front of its MFMAs
    auto chunk_begin = [&](auto mb_c, auto c_c, bool next_step_has_work) {
        constexpr int MB = decltype(mb_c)::value, c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 2>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, MB + 2 * c + 3>{}, wb[(c + 1) & 1][1]); }
        else if (next_step_has_work) { lds_pair(nxt_addr, std::integral_constant<int, MB>{}, wb[0][0]); lds_pair(nxt_addr, std::integral_constant<int, MB + 1>{}, wb[0][1]); }
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

    if (producer) {
        // =============================================== producer ===============================================
        using MB = std::integral_constant<int, 0>;         // first pair of this role in a step
        u32x4 xn[16][2];                                   // operand planes of this lane's token: [k-step][plane]
        f32x16 hA, hB;                                     // pre-activations of the even / odd hidden tile
        const float lut_scale = inv1 * kGeluLutScale;
        float gvv[4], gfr[4];
        f32x2 gtab[4];
        unsigned hw[2][4];                                 // hidden words of one k-step: [plane][word]
        auto fc_mma = [&](const u32x4 (&wa)[2], const u32x4 (&xa)[2], const u32x4 (&wc)[2], const u32x4 (&xc)[2], f32x16 &hd) {
            if (NP == 2) {
                hd = T::mfma(wa[1], xa[0], hd); hd = T::mfma(wc[1], xc[0], hd);
                hd = T::mfma(wa[0], xa[1], hd); hd = T::mfma(wc[0], xc[1], hd);
            }
            hd = T::mfma(wa[0], xa[0], hd); hd = T::mfma(wc[0], xc[0], hd);
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
        // one fc step: k-steps 8 half .. 8 half + 7 of the tile accumulating in hdst; the GELU of k-step `half` of hsrc's hidden
        // planes (pre-activations 8 half .. + 7) rides in the MFMA shadows
        auto step_fc = [&](auto half_c, f32x16 &hdst, const f32x16 &hsrc, int par, bool with_gelu, bool next_step_has_fc) {
            constexpr int half = decltype(half_c)::value;
            using VN = std::integral_constant<int, (NP == 2 ? 4 : 12)>;
            sync(E0{});
            chunk_begin(MB{}, I0{}, true);
            if (with_gelu) gelu0(std::integral_constant<int, 2 * half>{}, hsrc);
            fc_mma(wb[0][0], xn[8 * half], wb[0][1], xn[8 * half + 1], hdst);
            pin(VN{});
            chunk_begin(MB{}, I1{}, true);
            if (with_gelu) gelu1(std::integral_constant<int, 2 * half>{}, par);
            fc_mma(wb[1][0], xn[8 * half + 2], wb[1][1], xn[8 * half + 3], hdst);
            pin(VN{});
            chunk_begin(MB{}, I2{}, true);
            if (with_gelu) gelu0(std::integral_constant<int, 2 * half + 1>{}, hsrc);
            fc_mma(wb[0][0], xn[8 * half + 4], wb[0][1], xn[8 * half + 5], hdst);
            pin(VN{});
            chunk_begin(MB{}, I3{}, next_step_has_fc);
            if (with_gelu) gelu1(std::integral_constant<int, 2 * half + 1>{}, par);
            fc_mma(wb[1][0], xn[8 * half + 6], wb[1][1], xn[8 * half + 7], hdst);
            pin(VN{});
        };
        auto tile_fc = [&](f32x16 &hdst, const f32x16 &hsrc, int par, bool with_gelu, bool last_of_block) {
#pragma unroll
            for (int g = 0; g < 16; g++) hdst[g] = 0.f;
            step_fc(I0{}, hdst, hsrc, par, with_gelu, true);
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

        // attention-output planes of this pair's 32-token tile in block blk: [k-step][plane][lane][8] (pk_off), 1 KiB per fragment
        auto ytile = [&](int64_t blk) { return y + (((size_t)(blk * 4 + pair) * 16) * NP << 9) + (size_t)lane * 8; };
        u32x4 yb[4][2];                                    // y planes of up to three k-steps in flight: [k-step & 3][plane]
        auto y_load = [&](const uint16_t *yt, auto ko_c) {
            constexpr int ko = decltype(ko_c)::value;
            u32x4 (&d)[2] = yb[ko & 3];
            const uint16_t *yk = yt + ko * NP * 512;       // (the instruction's immediate offset is 13 bits)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[0]) : "v"(yk) : "memory");
            if (NP == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(d[1]) : "v"(yk) : "memory");
        };
        constexpr int YL = NP;                             // vector-memory operations per y_load
        f32x16 oa[8];                                      // out-projection accumulators, then the new residual rows
        auto op_mma = [&](const u32x4 (&wa)[2], const u32x4 (&wc)[2], const u32x4 (&yk)[2], f32x16 &ca, f32x16 &cb) {
            if (NP == 2) {
                ca = T::mfma(wa[1], yk[0], ca); cb = T::mfma(wc[1], yk[0], cb);
                ca = T::mfma(wa[0], yk[1], ca); cb = T::mfma(wc[0], yk[1], cb);
            }
            ca = T::mfma(wa[0], yk[0], ca); cb = T::mfma(wc[0], yk[0], cb);
        };
        // one out-projection step: k-step ko of y against the 8 output tiles (2 per chunk).  PENDING as in sync; YOUNGER = the
        // vector-memory operations issued after the loads of y(ko); the loads of y(ko + 2) follow the wait.  Steps 0 and 1 carry
        // the GELU of the previous block's tile 31.
        auto step_op = [&](auto ko_c, const uint16_t *yt, bool gelu, auto pending_c, auto younger_c) {
            constexpr int ko = decltype(ko_c)::value;
            using VN = std::integral_constant<int, (NP == 2 ? 4 : 12)>;
            sync(pending_c);
            {
                u32x4 (&yk)[2] = yb[ko & 3];
                if (NP == 2) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]), "+v"(yk[1]) : [n] "n"(decltype(younger_c)::value) : "memory");
                else { asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]) : [n] "n"(decltype(younger_c)::value) : "memory"); yk[1] = yk[0]; }
            }
            if constexpr (ko + 2 < kMQOut) y_load(yt, std::integral_constant<int, ko + 2>{});
            chunk_begin(MB{}, I0{}, true);
            if (gelu) gelu0(std::integral_constant<int, 2 * (ko & 1)>{}, hB);
            op_mma(wb[0][0], wb[0][1], yb[ko & 3], oa[0], oa[1]);
            if (gelu) pin(VN{}); else pin(E0{});
            chunk_begin(MB{}, I1{}, true);
            if (gelu) gelu1(std::integral_constant<int, 2 * (ko & 1)>{}, 1);
            op_mma(wb[1][0], wb[1][1], yb[ko & 3], oa[2], oa[3]);
            if (gelu) pin(VN{}); else pin(E0{});
            chunk_begin(MB{}, I2{}, true);
            if (gelu) gelu0(std::integral_constant<int, 2 * (ko & 1) + 1>{}, hB);
            op_mma(wb[0][0], wb[0][1], yb[ko & 3], oa[4], oa[5]);
            if (gelu) pin(VN{}); else pin(E0{});
            chunk_begin(MB{}, I3{}, ko + 1 < kMQOut);
            if (gelu) gelu1(std::integral_constant<int, 2 * (ko & 1) + 1>{}, 1);
            op_mma(wb[1][0], wb[1][1], yb[ko & 3], oa[6], oa[7]);
            if (gelu) pin(VN{}); else pin(E0{});
        };
        // the first two k-steps of the first block's y planes; the first fragments of step 0 are requested after its barrier
        y_load(ytile((int64_t)blockIdx.x), I0{});
        y_load(ytile((int64_t)blockIdx.x), I1{});

#pragma unroll 1
        for (int k = 0; k < n_mine; k++) {
            // (k == 0: hB is zero, its GELU writes zero hidden planes -- what the consumer's first steps expect)
            const int64_t blk = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
            const uint16_t *yt = ytile(blk);
            float *xrow = x + (blk * 128 + pair * 32 + r) * C + 4 * h;         // this lane's token, its half of every octet
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int g = 0; g < 16; g++) oa[j][g] = 0.f;
            // ---- steps 0 .. 15: y . c_proj^T into oa (model.py:71); GELU(tile 31 of the previous block) in steps 0, 1 ----
            // vector-memory operations per step: PW ring pieces | wait y(ko) | YL loads of y(ko + 2)
            {   // step 0: its first fragments are requested here (the previous step was the last c_fc step / nothing)
                using P0 = std::integral_constant<int, 2 * YL>;            // y(0), y(1) were issued in the previous step
                sync(P0{});
                lds_pair(cur_addr, std::integral_constant<int, 0>{}, wb[0][0]);
                lds_pair(cur_addr, std::integral_constant<int, 1>{}, wb[0][1]);
            }
            {   // (step 0's body without a second sync: same as step_op<0> after its sync)
                using VN = std::integral_constant<int, (NP == 2 ? 4 : 12)>;
                {
                    u32x4 (&yk)[2] = yb[0];
                    if (NP == 2) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]), "+v"(yk[1]) : [n] "n"(PW + YL) : "memory");
                    else { asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]) : [n] "n"(PW + YL) : "memory"); yk[1] = yk[0]; }
                }
                y_load(yt, I2{});
                chunk_begin(MB{}, I0{}, true);
                gelu0(I0{}, hB);
                op_mma(wb[0][0], wb[0][1], yb[0], oa[0], oa[1]);
                pin(VN{});
                chunk_begin(MB{}, I1{}, true);
                gelu1(I0{}, 1);
                op_mma(wb[1][0], wb[1][1], yb[0], oa[2], oa[3]);
                pin(VN{});
                chunk_begin(MB{}, I2{}, true);
                gelu0(I1{}, hB);
                op_mma(wb[0][0], wb[0][1], yb[0], oa[4], oa[5]);
                pin(VN{});
                chunk_begin(MB{}, I3{}, true);
                gelu1(I1{}, 1);
                op_mma(wb[1][0], wb[1][1], yb[0], oa[6], oa[7]);
                pin(VN{});
            }
            using PY = std::integral_constant<int, YL>;                  // the previous step's y loads
            using YY = std::integral_constant<int, 2 * PW + YL>;         // after y(ko): y(ko + 1) | PW | PW
            {
                // step 1 carries the second half of GELU(tile 31): q = 2, 3
                using VN = std::integral_constant<int, (NP == 2 ? 4 : 12)>;
                sync(PY{});
                {
                    u32x4 (&yk)[2] = yb[1];
                    if (NP == 2) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]), "+v"(yk[1]) : [n] "n"(2 * PW + YL) : "memory");
                    else { asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(yk[0]) : [n] "n"(2 * PW + YL) : "memory"); yk[1] = yk[0]; }
                }
                y_load(yt, I3{});
                chunk_begin(MB{}, I0{}, true);
                gelu0(I2{}, hB);
                op_mma(wb[0][0], wb[0][1], yb[1], oa[0], oa[1]);
                pin(VN{});
                chunk_begin(MB{}, I1{}, true);
                gelu1(I2{}, 1);
                op_mma(wb[1][0], wb[1][1], yb[1], oa[2], oa[3]);
                pin(VN{});
                chunk_begin(MB{}, I2{}, true);
                gelu0(I3{}, hB);
                op_mma(wb[0][0], wb[0][1], yb[1], oa[4], oa[5]);
                pin(VN{});
                chunk_begin(MB{}, I3{}, true);
                gelu1(I3{}, 1);
                op_mma(wb[1][0], wb[1][1], yb[1], oa[6], oa[7]);
                pin(VN{});
            }
            step_op(std::integral_constant<int, 2>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 3>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 4>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 5>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 6>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 7>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 8>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 9>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 10>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 11>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 12>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 13>{}, yt, false, PY{}, YY{});
            step_op(std::integral_constant<int, 14>{}, yt, false, PY{}, std::integral_constant<int, 2 * PW + YL>{});   // y(15) was issued in step 13
            step_op(std::integral_constant<int, 15>{}, yt, false, E0{}, std::integral_constant<int, 2 * PW>{});        // steps 14, 15 issue no loads
            // ---- steps 16 .. 19: x_mid = x + oa * inv_o (model.py:102), two output tiles per step, stored for the consumer's final
            //      residual add; oa becomes the rows LayerNorm reads.  Loads one step ahead: L(j) = 4 pieces of tile j. ----
            f32x4 xs[4][4];
            auto xl = [&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j & 3];
                float *xp = xrow;
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xj[gq]) : "v"(xp), "n"((32 * j + 8 * gq) * 4) : "memory");
            };
            float srow = 0.f;
            auto xm = [&](auto j_c, auto younger_c) {      // tile j: x_mid, store, running row sum
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j & 3];
                float *xp = xrow;
                asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]) : [n] "n"(decltype(younger_c)::value) : "memory");
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) { o[e] = fmaf(oa[j][4 * gq + e], inv_o, xj[gq][e]); oa[j][4 * gq + e] = o[e]; }
                    srow += (o[0] + o[1]) + (o[2] + o[3]);
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(xp), "v"(o), "n"((32 * j + 8 * gq) * 4) : "memory");
                }
            };
            using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
            using J4 = std::integral_constant<int, 4>; using J5 = std::integral_constant<int, 5>; using J6 = std::integral_constant<int, 6>; using J7 = std::integral_constant<int, 7>;
            xl(J0{}); xl(J1{});                                            // (in step 15, after its MFMAs)
            sync(std::integral_constant<int, 8>{});                        // step 16 (pending: L0 L1)
            xl(J2{}); xl(J3{});
            xm(J0{}, std::integral_constant<int, PW + 8>{}); xm(J1{}, std::integral_constant<int, PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                       // step 17 (pending: L2 L3 S0 S1)
            xl(J4{}); xl(J5{});
            xm(J2{}, std::integral_constant<int, 8 + PW + 8>{}); xm(J3{}, std::integral_constant<int, 8 + PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                       // step 18
            xl(J6{}); xl(J7{});
            xm(J4{}, std::integral_constant<int, 8 + PW + 8>{}); xm(J5{}, std::integral_constant<int, 8 + PW + 8 + 4>{});
            sync(std::integral_constant<int, 16>{});                       // step 19 (pending: L6 L7 S4 S5)
            xm(J6{}, std::integral_constant<int, 8 + PW>{}); xm(J7{}, std::integral_constant<int, 8 + PW + 4>{});
            // LayerNorm statistics of the new rows (two-pass, model.py:19-20)
            {   // the other half of the token sits in lane r + 32 (r - 32): v_permlane32_swap (VALU, no LDS traffic)
                float a = srow, b = srow;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                srow = a + b;
            }
            const float mean = srow * (1.0f / (float)C);
            float qv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int g = 0; g < 16; g++) { const float d = oa[j][g] - mean; qv = fmaf(d, d, qv); }
            {
                float a = qv, b = qv;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                qv = a + b;
            }
            const float rstd = rsqrtf(qv * (1.0f / (float)C) + 1e-5f);
            // ---- steps 20, 21: normalise and split the 16 k-steps (k-step ks = registers 8 (ks & 1) .. + 7 of tile ks >> 1) ----
            auto norm = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int j = ks >> 1, g0 = 8 * (ks & 1);
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { v0[e] = (oa[j][g0 + e] - mean) * rstd; v1[e] = (oa[j][g0 + 4 + e] - mean) * rstd; }
                u32x2 h0, l0, h1, l1;
                split4p<T, NP>(v0, h0, l0);
                split4p<T, NP>(v1, h1, l1);
                xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
                xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
            };
            sync(std::integral_constant<int, 8>{});                                      // step 20 (pending: S6 S7)
            norm(std::integral_constant<int, 0>{}); norm(std::integral_constant<int, 1>{}); norm(std::integral_constant<int, 2>{}); norm(std::integral_constant<int, 3>{});
            norm(std::integral_constant<int, 4>{}); norm(std::integral_constant<int, 5>{}); norm(std::integral_constant<int, 6>{}); norm(std::integral_constant<int, 7>{});
            sync(E0{});                                                                  // step 21
            norm(std::integral_constant<int, 8>{}); norm(std::integral_constant<int, 9>{}); norm(std::integral_constant<int, 10>{}); norm(std::integral_constant<int, 11>{});
            norm(std::integral_constant<int, 12>{}); norm(std::integral_constant<int, 13>{}); norm(std::integral_constant<int, 14>{}); norm(std::integral_constant<int, 15>{});
            // the first fragments of step 22 (its slot has landed for every wave: step 21's barrier)
            lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]);
            lds_pair(nxt_addr, std::integral_constant<int, 1>{}, wb[0][1]);
            // ---- steps 22 .. 85: c_fc of tiles 0 .. 31, GELU one tile behind ----
            tile_fc(hA, hB, 1, false, false);              // tile 0 (tile 31's GELU ran in steps 0, 1)
            tile_fc(hB, hA, 0, true, false);               // tile 1, GELU(tile 0) -> parity 0
#pragma unroll 1
            for (int t = 2; t < 32; t += 2) {
                tile_fc(hA, hB, 1, true, false);           // even tile, GELU(odd tile before it) -> parity 1
                tile_fc(hB, hA, 0, true, t == 30);         // odd tile, GELU(even tile) -> parity 0
            }
            if (k + 1 < n_mine) {                          // the next block's first two k-steps of y (still inside step 85)
                const uint16_t *yn = ytile(blk + gridDim.x);
                y_load(yn, I0{});
                y_load(yn, I1{});
            }
        }
        // ---- drain: GELU of the last block's tile 31 (steps 0, 1); the consumer finishes during steps 2 .. 7
        //      (no y loads were issued after the last block: nothing pending) ----
        sync(E0{}); gelu_only(I0{}, hB, 1);
        sync(E0{}); gelu_only(I1{}, hB, 1);
#pragma unroll 1
        for (int s_ = 2; s_ < 2 * kMQPause; s_++) sync(E0{});
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
        auto pj_mma = [&](const u32x4 (&wa)[2], const u32x4 (&wc)[2], const u32x4 (&hb)[2], f32x16 &ca, f32x16 &cb) {
            if (NP == 2) {
                ca = T::mfma(wa[1], hb[0], ca); cb = T::mfma(wc[1], hb[0], cb);
                ca = T::mfma(wa[0], hb[1], ca); cb = T::mfma(wc[0], hb[1], cb);
            }
            ca = T::mfma(wa[0], hb[0], ca); cb = T::mfma(wc[0], hb[0], cb);
        };
        // hidden planes of k-step kk of the tile with parity par
        auto load_hidden = [&](auto kk_c, int par) {
            constexpr int kk = decltype(kk_c)::value;
            const unsigned a = hand0 + (unsigned)par * (2 * NP * 1024);
            u32x4 (&hk)[2] = hf[kk];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hk[0]) : "v"(a), "n"(kk * NP * 1024) : "memory");
            if (NP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hk[1]) : "v"(a), "n"(kk * NP * 1024 + 1024) : "memory");
            else hk[1] = hk[0];
        };
        // one c_proj step: k-step kk of the hidden tile with parity par against all 8 output tiles (2 per chunk).
        // The kk = 0 step requests the tile's second k-step of hidden planes, the kk = 1 step the first k-step of the NEXT tile
        // (both are complete and visible by then: written one step earlier, a barrier in between)
        auto step_pj = [&](auto kk_c, int par, bool next_step_has_pj, bool prefetch_next_tile, auto pending_c) {
            constexpr int kk = decltype(kk_c)::value;
            sync(pending_c);
            chunk_begin(MB{}, I0{}, true);
            if (kk == 0) load_hidden(I1{}, par);
            if (kk == 1 && prefetch_next_tile) load_hidden(I0{}, par ^ 1);
            pj_mma(wb[0][0], wb[0][1], hf[kk], acc[0], acc[1]);
            pin(E0{});
            chunk_begin(MB{}, I1{}, true);
            pj_mma(wb[1][0], wb[1][1], hf[kk], acc[2], acc[3]);
            pin(E0{});
            chunk_begin(MB{}, I2{}, true);
            pj_mma(wb[0][0], wb[0][1], hf[kk], acc[4], acc[5]);
            pin(E0{});
            chunk_begin(MB{}, I3{}, next_step_has_pj);
            pj_mma(wb[1][0], wb[1][1], hf[kk], acc[6], acc[7]);
            pin(E0{});
        };

        // steps 0 .. 7 of a period for the consumer: the block blk_prev is finished (c_proj of its tiles 30, 31, then the
        // residual add + store, acc = 0)
        auto finish_block = [&](int64_t blk_prev) {
            float *xrow = x + (blk_prev * 128 + pair * 32 + r) * C + 4 * h;
            auto ld = [&](auto j_c) {                      // residual pieces of output tile j -> xs[j % 4]
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j % 4];
                float *xp = xrow;
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xj[gq]) : "v"(xp), "n"((32 * j + 8 * gq) * 4) : "memory");
            };
            auto st = [&](auto j_c, auto younger_c) {      // x = x + acc[j] * inv2 for output tile j; YOUNGER = operations issued after its loads
                constexpr int j = decltype(j_c)::value;
                f32x4 (&xj)[4] = xs[j % 4];
                float *xp = xrow;
                asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]) : [n] "n"(decltype(younger_c)::value) : "memory");
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = fmaf(acc[j][4 * gq + e], inv2, xj[gq][e]);
                    // (s_nop: a store of more than 8 bytes reads its data registers after issue; hipcc pads a VALU write of them for
                    //  its own stores, but it cannot see through inline asm -- without this the next piece's FMAs clobbered the data)
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(xp), "v"(o), "n"((32 * j + 8 * gq) * 4) : "memory");
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
            // vector-memory operations per step: PW ring pieces | 8 loads (two tiles) | 8 stores (two tiles)
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
            // ---- steps 8 .. 25: the producer's out-projection / LayerNorm / first c_fc tiles: nothing to consume ----
            sync(std::integral_constant<int, 8>{});                                        // step 8 (pending: S6 S7)
#pragma unroll 1
            for (int s_ = 9; s_ < kMQR0 + 4; s_++) sync(E0{});
            // the first fragments of step 26 and the first k-step of tile 0's hidden planes (complete since step 24)
            lds_pair(nxt_addr, std::integral_constant<int, 8>{}, wb[0][0]);
            lds_pair(nxt_addr, std::integral_constant<int, 9>{}, wb[0][1]);
            load_hidden(I0{}, 0);
            // ---- steps 26 .. 85: c_proj of tiles 0 .. 29 ----
            step_pj(I0{}, 0, true, false, E0{});
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
        if ((wave == 0 || wave == 4) && lane == 0) {
            unsigned long long *o = stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 4;
            o[0] = t_in[0]; o[1] = t_in[1]; o[2] = __builtin_readcyclecounter(); o[3] = STAMPS == 2 ? t_sync : wall_clock64();
        }
    }
}

}  // namespace fastk
}  // namespace mgpt
