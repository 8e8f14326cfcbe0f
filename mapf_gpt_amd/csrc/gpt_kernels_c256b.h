// gpt_kernels_c256b.h -- attn256o_kernel (gpt_kernels_c256a.h: the whole attention block of the 6M shape, persistent -- read its header first) with its q|k|v projection
// steps and its out-projection tail on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16 (two thirds of the kernel's MFMAs; DESIGN section 10 fact 5: 13-15 % more f16
// flops per second at the package power limit).  The ATTENTION PHASE (gpt_kernels_attn_tiles.h) is untouched: its inputs and outputs keep their 32 x 32 layouts, and the
// two layouts meet through one v_permlane16_swap per register pair (lane l: t = l % 16, q = l / 16 = 2 b5 + b4; a wave's 32 tokens are the groups tg = 0, 1):
//   q|k steps  fragment (kb, ug) = 16 dims x the 32 features of k-block kb; result quads D(ug, tg) = registers 4 (2 tg + ug) .. of qa / ka: token 16 tg + t, dims
//              8 blk + 4 ug + i.  k: blk = q -- a lane's eight dims are one 16-byte piece of the key's row in LDS (layout unchanged).  q: blk = 2 b4 + b5 -- then the
//              swap of the token-group index (a register index) with lane bit 4 turns the packed quads into the B operand of S^T = K Q^T as the attention phase
//              wants it: lane (b5 = k half, b4 t = query), register = k-step
//   v steps    natural orientation (rows = tokens, columns = dims): a lane's quad = four consecutive keys of one dim = 8 bytes of a V^T row, at the position the
//              attention phase's key order gives them (bits 2 and 3 of the key swapped)
//   y          the phase's output block (query r = lane % 32; registers = dims tau(g, h)) becomes the out-projection's operand (token 16 tg + t in lanes t + 16 q)
//              by the same swap between registers g and 8 + g; the order of the head's dims in a lane's eight values is a fixed permutation, baked into c_proj's stream
//   tail       pseudo-head = 64 output features = four groups of 16: qa holds groups 0, 1, ka groups 2, 3 (quad 2 (fg % 2) + tg); head hd is ONE k-block
//   x          as in gpt_kernels_c256q.h: lane (t, q) owns the 32 bytes of tokens t, 16 + t in chunk 4 kb + q (operand side) resp. the 16 bytes at half q % 2 of chunk
//              2 fg + q / 2 (residual quads); LayerNorm folds over the four lanes of a token
#pragma once
#include "gpt_kernels_c256a.h"

namespace mgpt {
namespace fastk {

// weight stream: as pack_attn256o_kernel -- [period step G][fragment ms][plane][lane][8], same sizes -- with 16 x 32 fragments (lane = row rho + 16 qk, eight k values):
//   q|k steps (st < 4)   chunk cc = 4 st + ms / 2: k-block cc / 2, q (cc even) or k (cc odd); fragment ms % 2 = unit group ug: dim 8 blk + 4 ug + rho % 4 with
//                        blk = rho / 4 for k, its two bits swapped for q (header)
//   v steps (st = 4, 5)  chunk cc = 4 (st - 4) + ms / 2 = k-block; fragment ms % 2 = dims 16 dg + rho
//   tail                 pseudo-head tt, step j: chunk cc = 4 j + ms / 2: k-block (= head) cc / 2, output features 64 tt + 16 (2 (cc % 2) + ms % 2) + rho; k slot (qk, e) =
//                        dim tau(8 (qk % 2) + e, qk / 2) of the head: the order the y planes come in (header)

template <class T, int NP>
__global__ __launch_bounds__(256) void pack_attn256q_kernel(const float *__restrict__ w_attn, const float *__restrict__ gain,
                                                            const float *__restrict__ w_proj, uint16_t *__restrict__ out,
                                                            float scale_a, float scale_p)
{
    constexpr int C = 256, NH = 8, QKV = NH * kA256StepsPerHead;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (period step, pair, lane)
    if (gid >= (int64_t)kA256oPeriod * 8 * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) & 7), G = (int)(gid >> 9);
    const int rho = lane & 15, qk = lane >> 4;
    float v[8];
    if (G < QKV) {
        const int head = G / kA256StepsPerHead, st = G - head * kA256StepsPerHead;
        int which, kb, dim;
        if (st < 4) {
            const int cc = 4 * st + (ms >> 1), ug = ms & 1, rq = rho >> 2;
            kb = cc >> 1; which = cc & 1;
            const int blk = which == 0 ? 2 * (rq & 1) + (rq >> 1) : rq;
            dim = 8 * blk + 4 * ug + (rho & 3);
        } else { which = 2; kb = 4 * (st - 4) + (ms >> 1); dim = 16 * (ms & 1) + rho; }
        const float *row = w_attn + (size_t)(which * C + head * 32 + dim) * C;    // c_attn.weight row = output feature (model.py:46-72)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int col = 32 * kb + 8 * qk + e;
            v[e] = row[col] * gain[col] * scale_a;                               // LayerNorm weight folded in (model.py:19-20)
        }
    } else {
        const int s = G - QKV, tt = s >> 2, j = s & 3;
        const int cc = 4 * j + (ms >> 1), kb = cc >> 1, fgl = 2 * (cc & 1) + (ms & 1);
        const float *row = w_proj + (size_t)(64 * tt + 16 * fgl + rho) * C;      // c_proj.weight row = output feature (model.py:31, 70)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int col = 32 * kb + (e & 3) + 8 * (2 * (qk & 1) + (e >> 2)) + 4 * (qk >> 1);
            v[e] = row[col] * scale_p;
        }
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

// STAMPS (tools/bench_probes/check_attn256o.hip only): wave 0 of every workgroup leaves {entry cycles, entry 100-MHz ticks, cycles in
// the prologues, in the q|k|v projection steps, in the attention phases (k / v barrier included), in the tail steps, in the
// tail epilogues, exit ticks} summed over its rows.  STAMPS == 2: waves 0 and 4 leave the step-phase cycles (sphase below).
// EMB (layer 0 of a forward, round 6): the rows do not exist yet -- the prologue takes them from the (position, token) embedding table
// etab[256 positions][67 tokens][C] = wpe[position] + wte[token] (model.py:171-175; built once per checkpoint, 17.5 MB: L2 / memory-side cache
// resident) through the SAME 32 loads per lane with other addresses, and writes them to x from there (the tail's residual read finds them:
// the same wave's stores, one attention phase earlier).  embed_tiled_kernel (0.75 ms per 12 288 rows: a 3.2-GB write) and the prologue's
// 3.2-GB read of what it wrote are gone.
template <class T, int NP, int STAMPS = 0, bool EMB = false>
__global__ __launch_bounds__(512, 2) void attn256q_kernel(float *__restrict__ x, const uint16_t *__restrict__ wstream, float inv_scale,
                                                          float scale_log2e, float inv_proj, unsigned char *__restrict__ spill,
                                                          int n_rows, unsigned long long *stamps = nullptr,
                                                          const unsigned char *__restrict__ tokens = nullptr, const float *__restrict__ etab = nullptr)
{
    constexpr int C = 256, KS = 16, NH = 8, HS = 32, NW = 8;
    static_assert(NP == 2 || NP == 1, "planes");
    constexpr int MS = 8;                                  // fragment pairs per step
    constexpr int STEP = MS * NP * 1024;
    constexpr int NSLOT = 5;
    constexpr int PW = MS * NP / NW;                       // direct-to-LDS loads per wave per step (2 in the split mode)
    constexpr int KROW = 80, VROW = 528;                   // padded LDS rows (bytes): conflict-free b128 reads
    constexpr int NSPILL = 2 * NP;                         // spill stores per head per wave (16 bytes per lane each)
    constexpr int NYLD = 14 * NP;                          // spill loads per row per wave
    static_assert(PW >= 1, "a wave moves at least one piece per step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [NSLOT][STEP] ring | sK [NP][256][KROW] | sV [NP][32][VROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok0 = wave * 32;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned sK = (unsigned)(size_t)smem + NSLOT * STEP, sV = sK + NP * kT * KROW;
    // Global addresses are (wave-uniform 64-bit base in SGPRs) + (one of two 32-bit lane offsets) + immediate: 64-bit per-lane
    // pointers for the x rows, the spill slab and the stream cost ~40 registers that this kernel does not have (first build:
    // hipcc hoisted them out of the row loop and spilled 242 dwords per lane to scratch)
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream) + (size_t)(wave * PW) * 1024;   // wave-uniform
    unsigned char *sp_wave = spill + ((size_t)blockIdx.x * NW + wave) * (size_t)(14 * NP * 1024);              // this wave's slab (uniform)
    // chunk-major x ([32-token tile][C / 8 chunks][32 tokens][8 floats]).  Operand side: lane (t, q) owns the 32 bytes of tokens t, 16 + t in chunk 4 kb + q;
    // residual side: the 16 bytes at half q % 2 of chunk 2 fg + q / 2 (header)
    // (both lane offsets are recomputed from lane16 through an opaque copy where they are used, as the K / V^T addresses below: as kernel-scope values they are two
    //  registers held across the attention phases)
    const int n_mine = n_rows > (int)blockIdx.x ? (n_rows - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = 0;
    auto phase = [&](int i) {                              // STAMPS == 1: cycles since the previous call go to ts[i]
        if constexpr (STAMPS == 1 || STAMPS == 3) { const unsigned long long t = __builtin_readcyclecounter(); ts[i] += t - t_mark; t_mark = t; }
    };
    // STAMPS == 3: as 1, left by wave 4 (the younger wave of SIMD 0) instead of wave 0.  STAMPS == 2, waves 0 AND 4: ts[2] / ts[4] = wait + barrier + piece issue / the four chunks of the q|k-shaped steps (projection
    // steps 0-3 of every head and the 16 tail steps: 48 per row), ts[3] / ts[5] = the same of the v steps (16 per row)
    auto smark = [&]() { if constexpr (STAMPS == 2) t_mark = __builtin_readcyclecounter(); };
    auto sphase = [&](int i) {
        if constexpr (STAMPS == 2) { const unsigned long long t = __builtin_readcyclecounter(); ts[i] += t - t_mark; t_mark = t; }
    };
    if constexpr (STAMPS != 0) { ts[0] = __builtin_readcyclecounter(); ts[1] = wall_clock64(); t_mark = ts[0]; }
    if (n_mine == 0) return;

    // ---- ring: slot of stream step G = G % 5, carried in two scalars; the source is cyclic with period 64 ----
    int r_issue = 0;
    int slot_cur = 0, slot_prev = NSLOT - 1;
    unsigned cur_addr = 0, nxt_addr = 0;
    auto issue = [&](int slot) {
        const unsigned char *src = wbase + (size_t)r_issue * STEP;
        unsigned char *dst = smem + (size_t)slot * STEP + (size_t)(wave * PW) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++) dma_piece(src + lane16, dst, std::integral_constant<int, 0>{}, i);
        r_issue = r_issue + 1 == kA256oPeriod ? 0 : r_issue + 1;
    };
#pragma unroll
    for (int G = 0; G < NSLOT - 1; G++) issue(G);
    // top of stream step G, part 1: this wave's pieces of step G + 1 have landed (PENDING = vector-memory operations of this wave
    // issued after them), every LDS access of the step before is done, barrier
    // (EMB: the first steps of a row's first head find the prologue's 32 row stores in flight and retire them with their counted waits.  Allowing
    //  them to stay in flight -- + 32 on the first three steps' counts -- was built and measured: no difference, profiles/r06_ab.txt visit D)
    auto sync_wait = [&](auto pending_c) {
        vm_wait<decltype(pending_c)::value>();
        __builtin_amdgcn_s_barrier();
    };
    // part 2: the slot of step G - 1 is refilled with step G + 4; addresses of this step's and the next step's slots
    int slot_refill = 0;                                   // (-DMGPT_AB_ATTNQ_DMA_PLACED: the refill is issued from inside the step's first chunk)
    auto sync_issue = [&]() {
#if defined(MGPT_AB_ATTNQ_DMA_PLACED) && !defined(MGPT_AB_ATTNQ_CLUMPED)
        slot_refill = slot_prev;
#else
        issue(slot_prev);
#endif
        const int slot_next = slot_cur + 1 == NSLOT ? 0 : slot_cur + 1;
        cur_addr = lds0 + (unsigned)slot_cur * STEP;
        nxt_addr = lds0 + (unsigned)slot_next * STEP;
        slot_prev = slot_cur;
        slot_cur = slot_next;
    };
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
    // chunk c of a step uses pairs 2c, 2c+1 (set c & 1), requested one chunk earlier; it requests the pairs of the next chunk
    // (chunk 3: the first pairs of the NEXT step, whose slot has landed -- the stream is cyclic, there always is one; NEXT = false
    //  only in a row's last step: the first pairs of the next row's first step are requested after its prologue instead, so that
    //  their 16 registers are not live across the 128-register LayerNorm)
    auto chunk_begin = [&](auto c_c, auto next_c) {
        constexpr int c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { lds_pair(cur_addr, std::integral_constant<int, 2 * c + 2>{}, wb[(c + 1) & 1][0]); lds_pair(cur_addr, std::integral_constant<int, 2 * c + 3>{}, wb[(c + 1) & 1][1]); }
        else if (decltype(next_c)::value) { lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]); lds_pair(nxt_addr, std::integral_constant<int, 1>{}, wb[0][1]); }
        __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this chunk's MFMAs
    };
    // The same requests, one at a time: read N of chunk c's list (fragment 2c+2 plane 0 [, plane 1], fragment 2c+3 ...).  Default build: each rides behind one of the
    // chunk's first MFMAs (the matrix pipe is busy for 16 cycles per MFMA, the LDS request issues in its shadow; -DMGPT_AB_ATTNQ_CLUMPED: all in front, as attn256o_kernel)
#if defined(MGPT_AB_ATTNQ_CLUMPED)
    constexpr bool PLACED = false;
#else
    constexpr bool PLACED = true;
#endif
    auto chunk_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk_read = [&](auto c_c, auto next_c, auto n_c) {
        constexpr int c = decltype(c_c)::value, n = decltype(n_c)::value;
        if constexpr (n < 2 * NP) {
            constexpr int fr = n / NP, pl = n % NP;
            if constexpr (c < 3) lds_frag(cur_addr, std::integral_constant<int, ((2 * c + 2 + fr) * NP + pl) * 1024>{}, wb[(c + 1) & 1][fr][pl]);
            else if constexpr (decltype(next_c)::value) lds_frag(nxt_addr, std::integral_constant<int, (fr * NP + pl) * 1024>{}, wb[0][fr][pl]);
            if constexpr (NP == 1 && (c < 3 || decltype(next_c)::value)) wb[(c + 1) & 1][fr][1] = wb[(c + 1) & 1][fr][0];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pin6 = [&]() {
#pragma unroll
        for (int n = 0; n < (NP == 2 ? 12 : 4); n++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pack_octet = [&](const f32x16 &v, int m, u32x4 (&dst)[2]) {
#pragma unroll
        for (int wd = 0; wd < 4; wd++) {
            unsigned a, b2;
            split2p<T, NP>(v[8 * m + 2 * wd], v[8 * m + 2 * wd + 1], a, b2);
            dst[0][wd] = a; dst[1][wd] = b2;
        }
    };
    // one product term on slice S (registers 4 S .. 4 S + 3) of a 16-register block
    auto mm16 = [&](const u32x4 &a, const u32x4 &b2, f32x16 &blk, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        f32x4 c = {blk[4 * S], blk[4 * S + 1], blk[4 * S + 2], blk[4 * S + 3]};
        c = T::mfma16(a, b2, c);
        blk[4 * S] = c[0]; blk[4 * S + 1] = c[1]; blk[4 * S + 2] = c[2]; blk[4 * S + 3] = c[3];
    };
    // the first term of a slice: C = 0 as the MFMA's inline constant (zeroing a slice by assignment costs four v_mov, see gpt_kernels_c256q.h)
    auto mm16z = [&](const u32x4 &a, const u32x4 &b2, f32x16 &blk, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = T::mfma16(a, b2, c);
        blk[4 * S] = c[0]; blk[4 * S + 1] = c[1]; blk[4 * S + 2] = c[2]; blk[4 * S + 3] = c[3];
    };
    // (LDS writes and global accesses below are written out with immediate offsets: no per-offset address registers.
    //  Global accesses use the saddr form: vdata, voffset (32-bit lane offset), saddr (uniform base), immediate.
    //  s_nop after a store: a store of more than 8 bytes reads its data registers after issue, and hipcc cannot see through asm.)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // fragments w0 (group 0), w1 (group 1) of one k-block against the two token groups x0, x1: slices 2 tg + group of blk.  SWAP: the token planes are the A operand
    // (rows = tokens: the v steps).  Small terms first, the four accumulator chains interleaved
    // behind(n): called after the chunk's MFMA n = 0 .. 5 (the placed LDS requests and, in a step's first chunk, the ring refill)
    // FIRST: the block's first chunk -- its slices start from zero
    auto mma4 = [&](const u32x4 (&w0)[2], const u32x4 (&w1)[2], const u32x4 (&x0)[2], const u32x4 (&x1)[2], f32x16 &blk, auto swap_c, auto first_c, auto &&behind) {
        constexpr bool SW = decltype(swap_c)::value, FIRST = decltype(first_c)::value;
        auto one = [&](const u32x4 &w, const u32x4 &xx, auto s_c) {
            if constexpr (SW) mm16(xx, w, blk, s_c); else mm16(w, xx, blk, s_c);
        };
        auto onez = [&](const u32x4 &w, const u32x4 &xx, auto s_c) {
            if constexpr (!FIRST) one(w, xx, s_c);
            else if constexpr (SW) mm16z(xx, w, blk, s_c);
            else mm16z(w, xx, blk, s_c);
        };
        if (NP == 2) {
            onez(w0[1], x0[0], I0{}); behind(I0{}); onez(w1[1], x0[0], I1{}); behind(I1{}); onez(w0[1], x1[0], I2{}); behind(I2{}); onez(w1[1], x1[0], I3{}); behind(I3{});
            one(w0[0], x0[1], I0{}); behind(std::integral_constant<int, 4>{}); one(w1[0], x0[1], I1{}); behind(std::integral_constant<int, 5>{});
            one(w0[0], x1[1], I2{}); one(w1[0], x1[1], I3{});
            one(w0[0], x0[0], I0{}); one(w1[0], x0[0], I1{}); one(w0[0], x1[0], I2{}); one(w1[0], x1[0], I3{});
        } else {
            onez(w0[0], x0[0], I0{}); behind(I0{}); onez(w1[0], x0[0], I1{}); behind(I1{}); onez(w0[0], x1[0], I2{}); behind(I2{}); onez(w1[0], x1[0], I3{});
        }
    };
    // the MFMAs of a chunk with its requests: `mm` runs mma4 with the hook it is given
    auto chunk_body = [&](auto c_c, auto next_c, auto &&mm) {
        if constexpr (PLACED) {
            chunk_wait();
            // -DMGPT_AB_ATTNQ_DMA_PLACED: a step's first chunk also issues the ring refill (PW direct-to-LDS pieces) two MFMAs behind the last read instead of at the
            // step's top.  Built and measured (profiles/r05_ab.txt, visit N): SLOWER, attention 48.0 -> 48.2 ms per cfg3 step -- the refill stays at the top
#if defined(MGPT_AB_ATTNQ_DMA_PLACED)
            constexpr bool REFILL = decltype(c_c)::value == 0;
#else
            constexpr bool REFILL = false;
#endif
            constexpr int N_REFILL = NP == 2 ? 5 : 2, N_LAST = REFILL ? N_REFILL : 2 * NP - 1, NMF = NP == 2 ? 12 : 4;
            mm([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                if constexpr (n < 2 * NP) { __builtin_amdgcn_sched_barrier(0); chunk_read(c_c, next_c, n_c); }
                else if constexpr (REFILL && n == N_REFILL) { __builtin_amdgcn_sched_barrier(0); issue(slot_refill); __builtin_amdgcn_sched_barrier(0); }
            });
#pragma unroll
            for (int n = 0; n < NMF - 1 - N_LAST; n++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            chunk_begin(c_c, next_c);
            mm([&](auto) {});
            pin6();
        }
    };
    // a token's 256 features sit in the four lanes t + 16 q: v_permlane16_swap folds rows 0|1 and 2|3 of 16 lanes, v_permlane32_swap the halves
    auto fold4 = [&](float v) {
        float a = v, b2 = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        float c = a + b2, d = c;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
        return c + d;
    };
    auto row_swap = [&](unsigned &a, unsigned &b2) {      // rows 1, 3 of a <-> rows 0, 2 of b2 (rows of 16 lanes)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
    };
    auto half_swap = [&](float v, float &lower, float &upper) {
        lower = v; upper = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lower), "+v"(upper));
    };
    auto other_half_max = [&](float v) { float a, b2; half_swap(v, a, b2); return fmaxf(a, b2); };
    auto other_half_sum = [&](float v) { float a, b2; half_swap(v, a, b2); return a + b2; };
    // Units of q and k.  Round 4 (and the -DMGPT_AB_ATTN_CLUMPED build): the planes carry the raw accumulators, i.e. q and k
    // times the weight stream's power-of-two scale, and the softmax multiplies every score by sc2.  Default build: the accumulators are
    // brought to q * log2(e) / sqrt(hs) and to k (true units) before they are split -- 32 multiplies per head -- so that a score IS
    // the exponent and the per-score multiply-add of the key-tile loop goes (see "one reference per query" below): sc2 = 1.
#if defined(MGPT_AB_ATTN_CLUMPED)
    constexpr bool QK_UNITS = false;
    const float sc2 = scale_log2e * inv_scale * inv_scale; // softmax exponent scale for q.k in weight-scaled units
#else
    constexpr bool QK_UNITS = true;
    const float sc2 = 1.0f;
#endif
    const float q_units = scale_log2e * inv_scale, k_units = inv_scale;
    // (the four K / V^T lane addresses are recomputed at the top of every head from lane16 through an opaque copy: as row-loop
    //  invariants they cost four registers that this kernel does not have -- one variant of it kept one in scratch, and the
    //  reload put a compiler vmcnt(0), which drains the ring, in front of every attention phase)

    u32x4 xn[KS][2];                                       // operand planes: LayerNorm(x) during the heads, y during the tail

    // one q|k-shaped step: chunks 4j .. 4j+3 (k-blocks 2j, 2j+1) of the planes in xn (xn[8 tg + kb]) against fragments (2c, 2c+1) -> slices 2 tg + group of qa | ka
    // (qa / ka live at kernel scope and are captured directly: handed to this lambda as reference PARAMETERS, hipcc sank the
    //  whole second chain -- 48 MFMAs -- behind the four steps and kept its 128 registers of weight fragments alive in scratch)
    f32x16 qa, ka;
    auto step_pair = [&](auto j_c, auto pending_c, auto &&after_barrier, auto next_c) {
        constexpr int j = decltype(j_c)::value;
        smark();
        sync_wait(pending_c);
        after_barrier();
        sync_issue();
        sphase(2);
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            constexpr int cc = 4 * j + c, kb = cc >> 1;    // even chunks -> qa, odd chunks -> ka
            chunk_body(c_c, next_c, [&](auto &&behind) {
                using F = std::integral_constant<bool, (cc < 2)>;      // chunks 0, 1 of a head's (pseudo-head's) first step start qa, ka from zero
                if constexpr ((cc & 1) == 0) mma4(wb[c & 1][0], wb[c & 1][1], xn[kb], xn[8 + kb], qa, std::false_type{}, F{}, behind);
                else mma4(wb[c & 1][0], wb[c & 1][1], xn[kb], xn[8 + kb], ka, std::false_type{}, F{}, behind);
            });
            asm volatile("" : "+v"(qa), "+v"(ka));         // both chains are pinned to this chunk (hipcc otherwise sinks a whole chain -- and
                                                           // the weight fragments it needs -- to the chain's first use, see above)
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
        sphase(4);
    };
    auto nothing = [&]() {};
    // every step but a row's first finds its first pairs requested by chunk 3 of the step before; a row's first step reads them after
    // the prologue from nxt_addr (the slot of the step about to run).  For the very first row that is slot 0: all priming pieces
    // landed for every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nxt_addr = lds0;
    using P4 = std::integral_constant<int, PW * (NSLOT - 3)>;               // only the pieces of the two later steps may be in flight
    using P4S = std::integral_constant<int, PW * (NSLOT - 3) + NSPILL>;     // ... and the spill stores of the head before

#pragma unroll 1
    for (int k = 0; k < n_mine; k++) {
        const int64_t b = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
        unsigned char *xw = reinterpret_cast<unsigned char *>(x + (b * kT + tok0) * C);   // this wave's 32-token tile (uniform), 32 KiB
        // ---- prologue: this lane's two tokens, LayerNorm (two-pass, model.py:19-20), operand planes ----
        {
            unsigned l16p = lane16;
            asm volatile("" : "+v"(l16p));
            const unsigned xoff_p = (l16p >> 8) * 1024 + ((l16p >> 4) & 15u) * 32;
            f32x4 xr[32];                                  // xr[2 (8 tg + kb) + hf] = features 32 kb + 8 q + 4 hf .. + 3 of token 16 tg + t
            if constexpr (EMB) {
                // ids of this lane's two tokens (tok0 + t, tok0 + 16 + t) -> byte offsets of their (position, token) table rows
                const unsigned char *tk = tokens + b * kT + tok0;                                      // uniform
                const unsigned tt_p = (l16p >> 4) & 15u;
                unsigned id0, id1;
                asm volatile("global_load_ubyte %0, %1, %2" : "=v"(id0) : "v"(tt_p), "s"(tk) : "memory");
                asm volatile("global_load_ubyte %0, %1, %2 offset:16" : "=v"(id1) : "v"(tt_p), "s"(tk) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(id0), "+v"(id1) : : "memory");
                const unsigned q32 = (l16p >> 8) * 32u;
                const unsigned ve0 = (((unsigned)tok0 + tt_p) * 67u + id0) * 1024u + q32;
                const unsigned ve1 = (((unsigned)tok0 + 16u + tt_p) * 67u + id1) * 1024u + q32;
                const unsigned char *eb = reinterpret_cast<const unsigned char *>(etab);
#pragma unroll
                for (int i = 0; i < 32; i++) {             // k-block i / 4: features 32 kb + 8 q + 4 hf of the row -> byte kb * 128 + q * 32 + hf * 16
                    if (((i >> 1) & 1) == 0)
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[2 * (i >> 2) + (i & 1)]) : "v"(ve0), "s"(eb), "n"((i >> 2) * 128 + (i & 1) * 16) : "memory");
                    else
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[2 * (8 + (i >> 2)) + (i & 1)]) : "v"(ve1), "s"(eb), "n"((i >> 2) * 128 + (i & 1) * 16) : "memory");
                }
            } else {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const unsigned char *xq = xw + (i >> 2) * 4096;   // k-block i / 4 (13-bit immediate offsets: one base per k-block); i % 4 = (tg, hf)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[2 * (8 * ((i >> 1) & 1) + (i >> 2)) + (i & 1)]) : "v"(xoff_p), "s"(xq), "n"(((i >> 1) & 1) * 512 + (i & 1) * 16) : "memory");
            }
            }
            // everything older (ring pieces, the previous row's last stores) retires with them: vmcnt(0)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) : : "memory");
            asm volatile("" : "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
            asm volatile("" : "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]), "+v"(xr[20]), "+v"(xr[21]), "+v"(xr[22]), "+v"(xr[23]));
            asm volatile("" : "+v"(xr[24]), "+v"(xr[25]), "+v"(xr[26]), "+v"(xr[27]), "+v"(xr[28]), "+v"(xr[29]), "+v"(xr[30]), "+v"(xr[31]));
            if constexpr (EMB) {
                // the rows' first appearance in x (chunk-major, the addresses the loads above would have had).  The 32 stores are younger than
                // every ring piece in flight: the first steps' counted waits simply retire them too (vector-memory operations retire in order)
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    const unsigned char *xq = xw + (i >> 2) * 4096;
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(xoff_p), "v"(xr[2 * (8 * ((i >> 1) & 1) + (i >> 2)) + (i & 1)]), "s"(xq), "n"(((i >> 1) & 1) * 512 + (i & 1) * 16) : "memory");
                }
            }
            float rstd[2];
#pragma unroll
            for (int tg = 0; tg < 2; tg++) {
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) sm += (xr[16 * tg + i][0] + xr[16 * tg + i][1]) + (xr[16 * tg + i][2] + xr[16 * tg + i][3]);
                const float mean = fold4(sm) * (1.0f / (float)C);
                float qv = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float d = xr[16 * tg + i][e] - mean; xr[16 * tg + i][e] = d; qv = fmaf(d, d, qv); }   // (the centred row is kept)
                rstd[tg] = rsqrtf(fold4(qv) * (1.0f / (float)C) + 1e-5f);
            }
            // (x - mean) * rstd; ln_1.weight is part of the weight stream (pack_attn256q_kernel)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {              // ks = 8 tg + kb
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { v0[e] = xr[2 * ks][e] * rstd[ks >> 3]; v1[e] = xr[2 * ks + 1][e] * rstd[ks >> 3]; }
                u32x2 h0, l0, h1, l1;
                split4p<T, NP>(v0, h0, l0);
                split4p<T, NP>(v1, h1, l1);
                xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
                xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
                // (k-step by k-step: left to itself the scheduler runs the multiplies of several k-steps ahead of their splits, and with
                //  the attention phase at 254 registers the allocator then spilled four quads of xn here)
                if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the first pairs of the row's first step (its slot landed for every wave before the barrier of the step before)
        lds_pair(nxt_addr, I0{}, wb[0][0]);
        lds_pair(nxt_addr, I1{}, wb[0][1]);
        phase(2);

        // ---- one head; LASTH (head 7): its output planes stay in registers (xn[14], xn[15]) and the planes of heads 0-6 are
        //      requested from the spill slab before its attention phase (xn is dead after the v steps) ----
        auto head = [&](int hd, auto last_c) {
            constexpr bool LASTH = decltype(last_c)::value;
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const unsigned rr = (l16 >> 4) & 31u, hh = l16 >> 9;                                   // r, h of this lane
            const unsigned kr_addr = sK + rr * KROW + hh * 16;                                    // read side: key r of a tile
            const unsigned vr_addr = sV + rr * VROW + hh * 16;                                    // read side: d = r
            const unsigned tt = (l16 >> 4) & 15u, qq = l16 >> 8;                                  // t, q of this lane
            const unsigned kw_addr = sK + ((unsigned)tok0 + tt) * KROW + qq * 16;                // write side: key tok0 + 16 tg + t, dims 8 q .. 8 q + 7
            const unsigned vw_addr = sV + tt * VROW + (unsigned)wave * 64 + (qq & 1u) * 16 + (qq >> 1) * 8;   // d = 16 dg + t, this wave's keys 16 tg + 4 q .. + 3 at their slots
            // ---- steps 0-3: q and k quads (lane = token 16 tg + t, registers 4 (2 tg + ug) + i = dim 8 blk + 4 ug + i) ----
            asm volatile("" : "=v"(qa), "=v"(ka));         // (no instruction: the old values end here -- not zeroed by assignment any more, they would stay live across the prologue)
            // (steps 0-2: the spill stores of the head before are younger than the pieces waited for; for head 0 nothing is in
            //  flight at all after the prologue's vmcnt(0), so the larger count is safe there too)
            step_pair(I0{}, P4S{}, nothing, std::true_type{});
            step_pair(I1{}, P4S{}, nothing, std::true_type{});
            step_pair(I2{}, P4S{}, nothing, std::true_type{});
            step_pair(I3{}, P4{}, nothing, std::true_type{});
            u32x4 qf[2][2];                                // B operand of S^T = K Q^T: [k-step][plane]
            if constexpr (QK_UNITS) {
#pragma unroll
                for (int g = 0; g < 16; g++) { qa[g] *= q_units; ka[g] *= k_units; }
            }
#pragma unroll
            for (int tg = 0; tg < 2; tg++) pack_octet(qa, tg, qf[tg]);         // token group tg, dims block blk(q) ...
#pragma unroll
            for (int pl = 0; pl < NP; pl++)                                     // ... -> lane (query t + 16 b4, half b5), k-step (header)
#pragma unroll
                for (int wd = 0; wd < 4; wd++) { unsigned a = qf[0][pl][wd], b2 = qf[1][pl][wd]; row_swap(a, b2); qf[0][pl][wd] = a; qf[1][pl][wd] = b2; }
            if (NP == 1) { qf[0][1] = qf[0][0]; qf[1][1] = qf[1][0]; }
            {   // k -> sK[pl][key = tok0 + 16 tg + t][dims 8 q ..]   (all waves passed this head's step syncs: the head before -- or the
                // row before -- has finished its attention everywhere)
                u32x4 kp[2][2];
#pragma unroll
                for (int tg = 0; tg < 2; tg++) pack_octet(ka, tg, kp[tg]);
#pragma unroll
                for (int tg = 0; tg < 2; tg++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(kw_addr), "v"(kp[tg][pl]), "n"(pl * kT * KROW + tg * 16 * KROW) : "memory");
            }
            // ---- steps 4-5: v quads (natural: lane = d = 16 dg + t, registers 4 (2 tg + dg) + i = token 16 tg + 4 q + i) ----
            f32x16 va;                                     // (started from zero by the first chunk)
            auto step_v = [&](auto j_c, auto next_c) {
                constexpr int j = decltype(j_c)::value;
                smark();
                sync_wait(P4{});
                sync_issue();
                sphase(3);
                auto chunk = [&](auto c_c) {
                    constexpr int c = decltype(c_c)::value;
                    constexpr int kb = 4 * j + c;
                    chunk_body(c_c, next_c, [&](auto &&behind) { mma4(wb[c & 1][0], wb[c & 1][1], xn[kb], xn[8 + kb], va, std::true_type{}, std::integral_constant<bool, kb == 0>{}, behind); });
                    asm volatile("" : "+v"(va));
                };
                chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{});
                sphase(5);
            };
            step_v(I0{}, std::true_type{});
#ifdef MGPT_AB_ATTN_CLUMPED
            step_v(I1{}, std::true_type{});
#else
            step_v(I1{}, std::false_type{});               // (the next step's first pairs are requested at the end of the attention phase)
#endif
            if constexpr (LASTH) {
                // the normalised rows are dead: their registers take the y planes of heads 0-6 back (this wave's own stores,
                // complete since the step waits above; L2-resident).  Needed at the first tail step, one attention phase away.
#pragma unroll
                for (int hh2 = 0; hh2 < 7; hh2++)
#pragma unroll
                    for (int tg = 0; tg < 2; tg++)
#pragma unroll
                        for (int pl = 0; pl < NP; pl++) {
                            const unsigned char *p = sp_wave + (size_t)hh2 * (size_t)(2 * NP * 1024);   // head hh2 (13-bit immediate offsets)
                            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xn[8 * tg + hh2][pl]) : "v"(lane16), "s"(p), "n"((tg * NP + pl) * 1024) : "memory");
                        }
            }
            {   // v^T -> sV[pl][d = 16 dg + t][wave's 32 keys]: keys 16 tg + 4 q .. + 3 sit at slots 16 tg + 8 (q % 2) + 4 (q / 2) .. (the phase's key order)
#pragma unroll
                for (int sl = 0; sl < 4; sl++) {           // slice 2 tg + dg
                    u32x2 vp[2];
                    unsigned a0, b0, a1, b1;
                    split2p<T, NP>(va[4 * sl], va[4 * sl + 1], a0, b0);
                    split2p<T, NP>(va[4 * sl + 2], va[4 * sl + 3], a1, b1);
                    vp[0][0] = a0; vp[0][1] = a1; vp[1][0] = b0; vp[1][1] = b1;
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(vw_addr), "v"(vp[pl]), "n"(pl * HS * VROW + (sl & 1) * 16 * VROW + (sl >> 1) * 32) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            phase(3);
            __builtin_amdgcn_s_barrier();                  // k, v^T of the head complete

            // ---- attention of this wave's 32 queries against the 256 keys of the head (gpt_kernels_attn_tiles.h) ----
            f32x16 o;
            float l_run = 0.f;
#if defined(MGPT_AB_ATTN_CLUMPED)
            attention_exact_tiles<T, NP, KROW, VROW, HS>(kr_addr, vr_addr, qf, sc2, o, l_run);
#else
            attention_tiles<T, NP, KROW, VROW, HS>(kr_addr, vr_addr, qf, lane, o, l_run);
            // the first pairs of the next stream step (the step after this phase; its slot landed for every wave before the last
            // v step's barrier) -- round 4 requested them in that step's chunk 3 and held their 16 registers across the whole phase
            lds_pair(nxt_addr, I0{}, wb[0][0]);
            lds_pair(nxt_addr, I1{}, wb[0][1]);
#endif
            // ---- y planes of the head: o[g] = O[query r][d = tau(g, h)] / l, times the v projection's weight scale; rows swapped between
            //      registers g and 8 + g: register octet tg = token group tg of k-block hd of the out-projection's B operand (header) ----
            {
                const float inv = inv_scale / l_run;
#pragma unroll
                for (int g = 0; g < 16; g++) o[g] *= inv;
#pragma unroll
                for (int g = 0; g < 8; g++) { float a = o[g], b2 = o[8 + g]; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2)); o[g] = a; o[8 + g] = b2; }
                if constexpr (LASTH) {
                    pack_octet(o, 0, xn[7]);
                    pack_octet(o, 1, xn[15]);
                } else {
                    u32x4 yp[2][2];
                    pack_octet(o, 0, yp[0]);
                    pack_octet(o, 1, yp[1]);
                    unsigned char *p = sp_wave + (size_t)hd * (size_t)(2 * NP * 1024);
#pragma unroll
                    for (int kk = 0; kk < 2; kk++)
#pragma unroll
                        for (int pl = 0; pl < NP; pl++)
                            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(lane16), "v"(yp[kk][pl]), "s"(p), "n"((kk * NP + pl) * 1024) : "memory");
                }
            }
            phase(4);
        };
#pragma unroll 1
        for (int hd = 0; hd < NH - 1; hd++) head(hd, std::false_type{});
        head(NH - 1, std::true_type{});

        // ---- tail: x <- x + y c_proj^T.  Pseudo-head t = output features 64 t .. 64 t + 63 (four groups of 16) over K = 256 in
        //      four q|k-shaped steps; their residual quads are requested at its first step ----
        // (Touching the next row's lines from here -- one dword per 128-byte line, so that the prologue's loads would come from L2 -- was
        //  built and measured in round 4: the lines do not survive in the 4-MiB L2 next to 7 MiB of spill slab per XCD, the counters show
        //  the row fetched twice (+3.2 GB per launch), and the kernel is 2 % SLOWER with the touches: 57.7 vs 56.5 ms per cfg3 step.)
        auto tail = [&](int t, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LASTT = decltype(last_c)::value;
            asm volatile("" : "=v"(qa), "=v"(ka));
            // (qa, ka are started from zero by the first step's first chunks.  Zeroed by assignment, hipcc built a 16-register zero block for the pseudo-head loop BEFORE the
            //  first pseudo-head and kept it there -- with it the allocator spilled residual quads that were still in flight)
            unsigned l16t = lane16;
            asm volatile("" : "+v"(l16t));
            const unsigned xoff_t = (l16t >> 9) * 1024 + ((l16t >> 4) & 15u) * 32 + ((l16t >> 8) & 1u) * 16;
            f32x4 xs[4][2];                                // residual quads: [feature group fgl][token group]
            unsigned char *xp = xw + (size_t)t * 8192;     // feature group 4 t + fgl: chunks 8 t + 2 fgl, + 1
            constexpr int OTHERS = 8;                      // vector-memory operations of this pseudo-head's first step besides ring pieces
            auto requests = [&]() {
#pragma unroll
                for (int fgl = 0; fgl < 4; fgl++)
#pragma unroll
                    for (int tg = 0; tg < 2; tg++)
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xs[fgl][tg]) : "v"(xoff_t), "s"(xp + (fgl >> 1) * 4096), "n"((fgl & 1) * 2048 + tg * 512) : "memory");
            };
            // step 0: pieces waited for are followed by 2 x PW pieces and [FIRST: the 28 spill loads | else: the 8 stores of the pseudo-head before]
            step_pair(I0{}, std::integral_constant<int, 2 * PW + (FIRST ? NYLD : 8)>{}, requests, std::true_type{});
            step_pair(I1{}, std::integral_constant<int, 2 * PW + 8 + OTHERS>{}, nothing, std::true_type{});
            step_pair(I2{}, std::integral_constant<int, 2 * PW + 8 + OTHERS>{}, nothing, std::true_type{});
            step_pair(I3{}, P4{}, nothing, std::integral_constant<bool, !LASTT>{});
            phase(5);
            // ---- epilogue of the four feature groups: x + acc / scale (the residual loads are older than this pseudo-head's ring pieces) ----
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1]), "+v"(xs[2][0]), "+v"(xs[2][1]), "+v"(xs[3][0]), "+v"(xs[3][1])
                         : [n] "n"(4 * PW) : "memory");
#pragma unroll
            for (int fgl = 0; fgl < 4; fgl++)                 // slices 2 tg + fgl % 2 of qa (fgl < 2) | ka
#pragma unroll
                for (int tg = 0; tg < 2; tg++) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaf((fgl < 2 ? qa : ka)[4 * (2 * tg + (fgl & 1)) + e], inv_proj, xs[fgl][tg][e]);
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(xoff_t), "v"(v), "s"(xp + (fgl >> 1) * 4096), "n"((fgl & 1) * 2048 + tg * 512) : "memory");
                }
            phase(6);
        };
        // the planes of heads 0-6 must have landed before the first tail MFMA reads them.  Everything in flight here -- three steps
        // of ring pieces and the 28 spill loads -- was issued at least one attention phase ago, so vmcnt(0) costs nothing and makes
        // the first pseudo-head's step waits trivially safe.
        {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn[0][0]), "+v"(xn[0][1]), "+v"(xn[1][0]), "+v"(xn[1][1]), "+v"(xn[2][0]), "+v"(xn[2][1]), "+v"(xn[3][0]), "+v"(xn[3][1]),
                         "+v"(xn[4][0]), "+v"(xn[4][1]), "+v"(xn[5][0]), "+v"(xn[5][1]), "+v"(xn[6][0]), "+v"(xn[6][1]) : : "memory");
            asm volatile("" : "+v"(xn[8][0]), "+v"(xn[8][1]), "+v"(xn[9][0]), "+v"(xn[9][1]), "+v"(xn[10][0]), "+v"(xn[10][1]), "+v"(xn[11][0]), "+v"(xn[11][1]),
                         "+v"(xn[12][0]), "+v"(xn[12][1]), "+v"(xn[13][0]), "+v"(xn[13][1]), "+v"(xn[14][0]), "+v"(xn[14][1]));
            // (head 7's planes too: left to float, their 16 accumulator registers stay live into the first tail step and the allocator spills residual quads that are still
            //  in flight -- a spill store of a register an asynchronous load has not yet written)
            asm volatile("" : "+v"(xn[7][0]), "+v"(xn[7][1]), "+v"(xn[15][0]), "+v"(xn[15][1]));
        }
        tail(0, std::true_type{}, std::false_type{});
#pragma unroll 1
        for (int t = 1; t < 3; t++) tail(t, std::false_type{}, std::false_type{});
        tail(3, std::false_type{}, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no direct-to-LDS load may outlive the workgroup
    if constexpr (STAMPS != 0) {
        if (lane == 0 && ((wave == 0 && STAMPS != 3) || (STAMPS == 2 && wave == 4) || (STAMPS == 3 && wave == 4))) {
            ts[7] = wall_clock64();
#pragma unroll
            for (int i = 0; i < 8; i++) stamps[((size_t)blockIdx.x * (STAMPS == 2 ? 2 : 1) + (STAMPS == 2 ? (wave >> 2) : 0)) * 8 + i] = ts[i];
        }
    }
}

}  // namespace fastk
}  // namespace mgpt
