// gpt_kernels_c160a.h -- the whole attention block for n_embd = 160, 5 heads of 32 (MAPF-GPT-2M), gfx950:
//     x <- x + c_proj(attention(LayerNorm(x)))                         (model.py:46-72, 102)
// the structure of attn256o_kernel (gpt_kernels_c256a.h) at this shape: PERSISTENT (grid = number of CUs, a workgroup walks
// rows b = blockIdx.x, + gridDim.x, ...), a wave owns 32 tokens and keeps their normalised rows as MFMA operand planes
// (10 k-steps: 80 registers), q, k, v, the scores and y never leave the chip, and the out-projection is a TAIL over the y
// planes (heads 0-3 parked in a per-workgroup spill slab, head 4 kept in registers) so that the residual stream is read
// twice and written once per row by this kernel alone.  It replaces attn_block_kernel (compiler-scheduled, one workgroup
// per row, weight packets fetched per head) on every layer but the first (which gathers the embedding) and the last
// (attn_last1_kernel) of launches above kSmallRows rows.
//
// Stream: "steps" of 10 weight fragments (32 rows x 16 k, hi and lo plane: 20 KiB in the split mode), CYCLIC with period 20:
//     head hd: steps 3 hd, 3 hd + 1 = q|k (k-steps 0-4, 5-9: fragment 2 c = q tile, 2 c + 1 = k tile of k-step 5 j + c),
//              step 3 hd + 2 = v (fragment m = k-step m)                      -- c_attn.weight * ln_1.weight
//     tail:    steps 15, 16 = output tiles 0, 1; 17, 18 = tiles 2, 3 (as q|k); 19 = tile 4 (fragment m = k-step m) -- c_proj.weight
// through a 4-slot LDS ring (direct global->LDS loads three steps ahead, one raw s_barrier per step).  A step is five chunks of
// two fragments = 6 MFMAs on two accumulator chains; the fragments of a chunk are read from LDS one chunk ahead into one of
// three register sets (pattern 0 1 0 1 2, so that every step starts on set 0).  A step has 20 pieces of 1 KiB for 8 waves:
// every wave issues 3 (1-plane mode: 2 of 10), the surplus re-fetches the last piece (same bytes to the same place).
// All vector-memory operations of the row loop are inline asm with hand-counted s_waitcnt (they retire in issue order:
// "at most N outstanding" with N = the operations issued after the one needed is exact; a smaller N is always safe).
// Numerics: the same products as attn_block_kernel except that ln_1.weight rides on the stream (as in the 6M kernels) and the
// heads' c_proj contributions are summed inside one accumulator chain per output tile (k = head-major d); results per token
// do not depend on the grid.
#pragma once
#include "gpt_kernels_c256a.h"

namespace mgpt {
namespace fastk {

constexpr int kA160oPeriod = 20;                // stream steps per row
constexpr int kA160oFrags = 10;                 // fragments per step

template <int NP>
constexpr int kA160oLds = 4 * kA160oFrags * NP * 1024 + NP * (256 * 80 + 32 * 528);     // 4-slot ring | K | V^T planes of a head
template <int NP>
constexpr int kA160oSpillPerWg = 8 * 8 * NP * 1024;                                      // 8 waves x 8 k-steps (heads 0-3) x NP planes x 1 KiB

template <class T, int NP>
__global__ __launch_bounds__(256) void pack_attn160o_kernel(const float *__restrict__ w_attn, const float *__restrict__ gain,
                                                            const float *__restrict__ w_proj, uint16_t *__restrict__ out,
                                                            float scale_a, float scale_p)
{
    constexpr int C = 160, MS = kA160oFrags;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (period step, fragment, lane)
    if (gid >= (int64_t)kA160oPeriod * MS * 64) return;
    const int lane = (int)(gid & 63), ms = (int)((gid >> 6) % MS), G = (int)((gid >> 6) / MS);
    const int i = lane & 31, h = lane >> 5;
    const float *row;
    int ks;
    bool attn = true;
    if (G < 15) {
        const int head = G / 3, st = G - 3 * head;
        int which;
        if (st < 2) { which = ms & 1; ks = 5 * st + (ms >> 1); }
        else { which = 2; ks = ms; }
        row = w_attn + (size_t)(which * C + head * 32 + i) * C;
    } else {
        const int s = G - 15;
        attn = false;
        if (s < 4) { const int t = s >> 1, j = s & 1; ks = 5 * j + (ms >> 1); row = w_proj + (size_t)(32 * (2 * t + (ms & 1)) + i) * C; }
        else { ks = ms; row = w_proj + (size_t)(32 * 4 + i) * C; }
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int g = 8 * (ks & 1) + e;
        const int col = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;      // the operand planes' k-slot order (attn256o_kernel)
        v[e] = attn ? row[col] * gain[col] * scale_a : row[col] * scale_p;
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)G * MS + ms) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// EMB (layer 0, round 6; as attn256q_kernel<.., EMB>, gpt_kernels_c256b.h): the prologue takes the rows from the (position, token) embedding table
// etab[256][67][C] = wpe + wte and writes them to x itself; no embedding kernel, no first read of x
template <class T, int NP, bool EMB = false>
__global__ __launch_bounds__(512, 2) void attn160o_kernel(float *__restrict__ x, const uint16_t *__restrict__ wstream, float inv_scale,
                                                          float scale_log2e, float inv_proj, unsigned char *__restrict__ spill, int n_rows,
                                                          const unsigned char *__restrict__ tokens = nullptr, const float *__restrict__ etab = nullptr)
{
    constexpr int C = 160, KS = 10, NH = 5, HS = 32, NW = 8;
    static_assert(NP == 2 || NP == 1, "planes");
    constexpr int MS = kA160oFrags;                        // fragments per step
    constexpr int NPIECE = MS * NP;                        // 1-KiB pieces per step
    constexpr int STEP = NPIECE * 1024;
    constexpr int NSLOT = 4;
    constexpr int PW = (NPIECE + NW - 1) / NW;             // direct-to-LDS loads per wave per step (3 in the split mode, the surplus repeats the last piece)
    constexpr int KROW = 80, VROW = 528;                   // padded LDS rows (bytes): conflict-free b128 reads
    constexpr int NSPILL = 2 * NP;                         // spill stores per head per wave (16 bytes per lane each)
    constexpr int NYLD = 8 * NP;                           // spill loads per row per wave
    constexpr int XT = 32 * C * 4;                         // bytes of a 32-token tile of x
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [NSLOT][STEP] ring | sK [NP][256][KROW] | sV [NP][32][VROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tok0 = wave * 32;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)smem + lane16;
    const unsigned sK = (unsigned)(size_t)smem + NSLOT * STEP, sV = sK + NP * kT * KROW;
    // global addresses: wave-uniform 64-bit base in SGPRs + a 32-bit lane offset + immediate (see attn256o_kernel)
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wstream);
    unsigned char *sp_wave = spill + ((size_t)blockIdx.x * NW + wave) * (size_t)(8 * NP * 1024);              // this wave's slab (uniform)
    const unsigned xoff = (unsigned)(r * 32 + h * 16);     // chunk-major x: lane (r, h) owns the 16 bytes at r * 32 + h * 16 of every 1-KiB chunk
    const int n_mine = n_rows > (int)blockIdx.x ? (n_rows - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_mine == 0) return;
    // this wave's pieces of a step: wave, wave + 8, wave + 16 -- clamped to the last one
    int pc[PW];
#pragma unroll
    for (int i = 0; i < PW; i++) pc[i] = (wave + NW * i < NPIECE ? wave + NW * i : NPIECE - 1) * 1024;

    // ---- ring: slot of stream step G = G % 4, carried in two scalars; the source is cyclic with period 20 ----
    int r_issue = 0;
    int slot_cur = 0, slot_prev = NSLOT - 1;
    unsigned cur_addr = 0, nxt_addr = 0;
    auto issue = [&](int slot) {
        const unsigned char *src = wbase + (size_t)r_issue * STEP + lane16;
        unsigned char *dst = smem + (size_t)slot * STEP;
#pragma unroll
        for (int i = 0; i < PW; i++) dma_piece(src + pc[i], dst + pc[i], std::integral_constant<int, 0>{}, 0);
        r_issue = r_issue + 1 == kA160oPeriod ? 0 : r_issue + 1;
    };
#pragma unroll
    for (int G = 0; G < NSLOT - 1; G++) issue(G);
    // top of stream step G, part 1: this wave's pieces of step G + 1 have landed (PENDING = vector-memory operations of this wave
    // issued after them), every LDS access of the step before is done, barrier
    auto sync_wait = [&](auto pending_c) {
        vm_wait<decltype(pending_c)::value>();
        __builtin_amdgcn_s_barrier();
    };
    // part 2: the slot of step G - 1 is refilled with step G + 3; addresses of this step's and the next step's slots
    auto sync_issue = [&]() {
        issue(slot_prev);
        const int slot_next = slot_cur + 1 == NSLOT ? 0 : slot_cur + 1;
        cur_addr = lds0 + (unsigned)slot_cur * STEP;
        nxt_addr = lds0 + (unsigned)slot_next * STEP;
        slot_prev = slot_cur;
        slot_cur = slot_next;
    };
    u32x4 wb[3][2][2];                                     // weight fragments: [set][fragment 2c / 2c+1][plane]
    auto lds_frag = [&](unsigned addr, auto off_c, u32x4 &dst) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off_c)::value) : "memory");
    };
    auto lds_pair = [&](unsigned slot_addr, auto ms_c, u32x4 (&dst)[2]) {
        constexpr int ms = decltype(ms_c)::value;
        lds_frag(slot_addr, std::integral_constant<int, ms * NP * 1024>{}, dst[0]);
        if (NP == 2) lds_frag(slot_addr, std::integral_constant<int, (ms * NP + 1) * 1024>{}, dst[1]);
        else dst[1] = dst[0];
    };
    // chunk c of a step uses fragments 2c, 2c+1 in register set {0, 1, 0, 1, 2}[c], requested one chunk earlier; it requests the
    // fragments of the next chunk (chunk 4: the first fragments of the NEXT step into set 0, whose slot has landed -- the stream
    // is cyclic, there always is one; NEXT = false only in a row's last step: the next row's first fragments are requested after
    // its prologue instead, so that their registers are not live across the LayerNorm)
    auto chunk_begin = [&](auto c_c, auto next_c) {
        constexpr int c = decltype(c_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (c < 4) {
            constexpr int sn = (c + 1 == 4) ? 2 : ((c + 1) & 1);
            lds_pair(cur_addr, std::integral_constant<int, 2 * c + 2>{}, wb[sn][0]);
            lds_pair(cur_addr, std::integral_constant<int, 2 * c + 3>{}, wb[sn][1]);
        } else if constexpr (decltype(next_c)::value) {
            lds_pair(nxt_addr, std::integral_constant<int, 0>{}, wb[0][0]);
            lds_pair(nxt_addr, std::integral_constant<int, 1>{}, wb[0][1]);
        }
        __builtin_amdgcn_sched_barrier(0);                 // the requests stay in front of this chunk's MFMAs
    };
    // the same requests one at a time (default build: read n rides behind the chunk's MFMA n -- DESIGN 11.8; -DMGPT_AB_ATTN160_CLUMPED: all in front)
#if defined(MGPT_AB_ATTN160_CLUMPED)
    constexpr bool PLACED = false;
#else
    constexpr bool PLACED = true;
#endif
    auto chunk_read = [&](auto c_c, auto next_c, auto n_c) {
        constexpr int c = decltype(c_c)::value, n = decltype(n_c)::value;
        if constexpr (n < 2 * NP) {
            constexpr int fr = n / NP, pl = n % NP;
            constexpr int sn = (c + 1 == 4) ? 2 : (c == 4 ? 0 : ((c + 1) & 1));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (c < 4) lds_frag(cur_addr, std::integral_constant<int, ((2 * c + 2 + fr) * NP + pl) * 1024>{}, wb[sn][fr][pl]);
            else if constexpr (decltype(next_c)::value) lds_frag(nxt_addr, std::integral_constant<int, (fr * NP + pl) * 1024>{}, wb[0][fr][pl]);
            if constexpr (NP == 1 && (c < 4 || decltype(next_c)::value)) wb[sn][fr][1] = wb[sn][fr][0];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pin6 = [&]() {
#pragma unroll
        for (int n = 0; n < (NP == 2 ? 6 : 2); n++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
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
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    auto half_swap = [&](float v, float &lower, float &upper) {
        lower = v; upper = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lower), "+v"(upper));
    };
    auto other_half_max = [&](float v) { float a, b2; half_swap(v, a, b2); return fmaxf(a, b2); };
    auto other_half_sum = [&](float v) { float a, b2; half_swap(v, a, b2); return a + b2; };
    // units of q and k: see attn256o_kernel (default build: exponent units, 32 multiplies per head before the split)
#if defined(MGPT_AB_ATTN_CLUMPED)
    constexpr bool QK_UNITS = false;
    const float sc2 = scale_log2e * inv_scale * inv_scale; // softmax exponent scale for q.k in weight-scaled units
#else
    constexpr bool QK_UNITS = true;
#endif
    const float q_units = scale_log2e * inv_scale, k_units = inv_scale;

    u32x4 xn[KS][2];                                       // operand planes: LayerNorm(x) during the heads, y during the tail

    // One step = five chunks on the two accumulator chains qa / ka (kernel scope, captured directly: see attn256o_kernel).
    //   MODE 0 (q|k, tail tile pairs): chain 0 += W[2c] xn[5j+c], chain 1 += W[2c+1] xn[5j+c]   (swapped: lane = token)
    //   MODE 1 (tail, single tile):    chain 0 += W[2c] xn[2c],   chain 1 += W[2c+1] xn[2c+1]   (swapped)
    //   MODE 2 (v):                    chain 0 += xn[2c] W[2c],   chain 1 += xn[2c+1] W[2c+1]   (natural: lane = d)
    f32x16 qa, ka;
    auto step = [&](auto mode_c, auto j_c, auto pending_c, auto &&after_barrier, auto next_c) {
        constexpr int MODE = decltype(mode_c)::value, j = decltype(j_c)::value;
        sync_wait(pending_c);
        after_barrier();
        sync_issue();
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            constexpr int s = (c == 4) ? 2 : (c & 1);
            constexpr int k0 = MODE == 0 ? 5 * j + c : 2 * c, k1 = MODE == 0 ? 5 * j + c : 2 * c + 1;
            if constexpr (PLACED) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else chunk_begin(c_c, next_c);
            auto behind = [&](auto n_c) { if constexpr (PLACED) chunk_read(c_c, next_c, n_c); };
            using N0 = std::integral_constant<int, 0>; using N1 = std::integral_constant<int, 1>;
            using N2 = std::integral_constant<int, 2>; using N3 = std::integral_constant<int, 3>;
            if (MODE == 2) {
                if (NP == 2) {
                    qa = T::mfma(xn[k0][1], wb[s][0][0], qa); behind(N0{}); ka = T::mfma(xn[k1][1], wb[s][1][0], ka); behind(N1{});
                    qa = T::mfma(xn[k0][0], wb[s][0][1], qa); behind(N2{}); ka = T::mfma(xn[k1][0], wb[s][1][1], ka); behind(N3{});
                    qa = T::mfma(xn[k0][0], wb[s][0][0], qa); ka = T::mfma(xn[k1][0], wb[s][1][0], ka);
                } else {
                    qa = T::mfma(xn[k0][0], wb[s][0][0], qa); behind(N0{}); ka = T::mfma(xn[k1][0], wb[s][1][0], ka); behind(N1{});
                }
            } else {
                if (NP == 2) {
                    qa = T::mfma(wb[s][0][1], xn[k0][0], qa); behind(N0{}); ka = T::mfma(wb[s][1][1], xn[k1][0], ka); behind(N1{});
                    qa = T::mfma(wb[s][0][0], xn[k0][1], qa); behind(N2{}); ka = T::mfma(wb[s][1][0], xn[k1][1], ka); behind(N3{});
                    qa = T::mfma(wb[s][0][0], xn[k0][0], qa); ka = T::mfma(wb[s][1][0], xn[k1][0], ka);
                } else {
                    qa = T::mfma(wb[s][0][0], xn[k0][0], qa); behind(N0{}); ka = T::mfma(wb[s][1][0], xn[k1][0], ka); behind(N1{});
                }
            }
            if constexpr (PLACED) {
                if (NP == 2) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            } else pin6();
            asm volatile("" : "+v"(qa), "+v"(ka));         // both chains are pinned to this chunk
        };
        chunk(I0{}); chunk(I1{}); chunk(I2{}); chunk(I3{}); chunk(I4{});
    };
    auto nothing = [&]() {};
    auto zero_chains = [&]() {
#pragma unroll
        for (int g = 0; g < 16; g++) { qa[g] = 0.f; ka[g] = 0.f; }
    };
    // every step but a row's first finds its first fragments requested by chunk 4 of the step before; a row's first step reads
    // them after the prologue from nxt_addr (the slot of the step about to run).  For the very first row that is slot 0: all
    // priming pieces landed for every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nxt_addr = lds0;
    using P4 = std::integral_constant<int, PW * (NSLOT - 3)>;               // only the pieces of the one later step may be in flight
    using P4S = std::integral_constant<int, PW * (NSLOT - 3) + NSPILL>;     // ... and the spill stores of the head before

#pragma unroll 1
    for (int k = 0; k < n_mine; k++) {
        const int64_t b = (int64_t)blockIdx.x + (int64_t)k * gridDim.x;
        // x is chunk-major (xt_off): [32-token tile][C / 8 chunks][32 tokens][8 floats]; lane (r, h) owns the 16 bytes at
        // r * 32 + h * 16 of every 1-KiB chunk, so that a wave's load or store is 1 KiB contiguous
        unsigned char *xw = reinterpret_cast<unsigned char *>(x + (b * kT + tok0) * C);   // this wave's 32-token tile (uniform), 20 KiB
        // ---- prologue: this lane's token, LayerNorm (two-pass, model.py:19-20), operand planes ----
        {
            f32x4 xr[2 * KS];                              // xr[i] = features 8 i + 4 h .. + 3 (chunk i)
            if constexpr (EMB) {
                unsigned xoff_e = xoff;
                asm volatile("" : "+v"(xoff_e));
                const unsigned rr_e = xoff_e >> 5, h16 = xoff_e & 16u;                                  // this lane's token r and half h
                const unsigned char *tk = tokens + b * kT + tok0;                                        // uniform
                unsigned id;
                asm volatile("global_load_ubyte %0, %1, %2" : "=v"(id) : "v"(rr_e), "s"(tk) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(id) : : "memory");
                const unsigned ve = (((unsigned)tok0 + rr_e) * 67u + id) * (unsigned)(C * 4) + h16;     // the (position, token) row, this lane's half of a chunk
                const unsigned char *eb = reinterpret_cast<const unsigned char *>(etab);
#pragma unroll
                for (int i = 0; i < 2 * KS; i++)
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[i]) : "v"(ve), "s"(eb), "n"(i * 32) : "memory");
            } else {
#pragma unroll
            for (int i = 0; i < 2 * KS; i++) {
                const unsigned char *xq = xw + (i >> 2) * 4096;   // (13-bit immediate offsets: four chunks per base)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[i]) : "v"(xoff), "s"(xq), "n"((i & 3) * 1024) : "memory");
            }
            }
            // everything older (ring pieces, the previous row's last stores) retires with them: vmcnt(0)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]), "+v"(xr[8]), "+v"(xr[9]) : : "memory");
            asm volatile("" : "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]), "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]));
            if constexpr (EMB) {                           // the rows' first appearance in x (they retire with the first steps' counted waits)
#pragma unroll
                for (int i = 0; i < 2 * KS; i++) {
                    const unsigned char *xq = xw + (i >> 2) * 4096;
                    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(xoff), "v"(xr[i]), "s"(xq), "n"((i & 3) * 1024) : "memory");
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * KS; i++) s += (xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]);
            s = other_half_sum(s);
            const float mean = s * (1.0f / (float)C);
            float qv = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * KS; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) { const float d = xr[i][e] - mean; xr[i][e] = d; qv = fmaf(d, d, qv); }   // (the centred row is kept)
            qv = other_half_sum(qv);
            const float rstd = rsqrtf(qv * (1.0f / (float)C) + 1e-5f);
            // (x - mean) * rstd; ln_1.weight is part of the weight stream (pack_attn160o_kernel)
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
        }
        // the first fragments of the row's first step (its slot landed for every wave before the barrier of the step before)
        lds_pair(nxt_addr, I0{}, wb[0][0]);
        lds_pair(nxt_addr, I1{}, wb[0][1]);

        // ---- one head; LASTH (head 4): its output planes stay in registers (xn[8], xn[9]) and the planes of heads 0-3 are
        //      requested from the spill slab before its attention phase (xn is dead after the v step) ----
        auto head = [&](int hd, auto last_c) {
            constexpr bool LASTH = decltype(last_c)::value;
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const unsigned rr = (l16 >> 4) & 31u, hh = l16 >> 9;                                   // r, h of this lane
            const unsigned kr_addr = sK + rr * KROW + hh * 16;                                    // read side: key r of a tile
            const unsigned vr_addr = sV + rr * VROW + hh * 16;                                    // read side: d = r
            const unsigned kw_addr = kr_addr + (unsigned)tok0 * KROW;                             // this lane's key row (write side)
            const unsigned vw_addr = vr_addr + (unsigned)wave * 64;                               // this lane's d row, this wave's keys
            // ---- steps 0-1: q and k tiles (swapped: lane = token, registers = d) ----
            zero_chains();
            // (the spill stores of the head before are younger than the pieces waited for; for head 0 nothing is in flight at all
            //  after the prologue's vmcnt(0), so the larger count is safe there too)
            step(I0{}, I0{}, P4S{}, nothing, std::true_type{});
            step(I0{}, I1{}, P4S{}, nothing, std::true_type{});
            u32x4 qf[2][2];                                // B operand of S^T = K Q^T: [k-step][plane]
            if constexpr (QK_UNITS) {
#pragma unroll
                for (int g = 0; g < 16; g++) { qa[g] *= q_units; ka[g] *= k_units; }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) pack_octet(qa, ks, qf[ks]);
            {   // k -> sK[pl][key = tok0 + r][octet ks][half h]   (all waves passed this head's step syncs: the head before -- or the
                // row before -- has finished its attention everywhere)
                u32x4 kp[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ks++) pack_octet(ka, ks, kp[ks]);
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(kw_addr), "v"(kp[ks][pl]), "n"(pl * kT * KROW + ks * 32) : "memory");
            }
            // ---- step 2: v tile (natural: lane = d, registers = tokens), two chains ----
            zero_chains();
            step(I2{}, I0{}, P4{}, nothing, std::true_type{});
            if constexpr (LASTH) {
                // the normalised rows are dead: their registers take the y planes of heads 0-3 back (this wave's own stores,
                // complete since the step waits above; L2-resident).  Needed at the first tail step, one attention phase away.
#pragma unroll
                for (int ks = 0; ks < 8; ks++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) {
                        const unsigned char *p = sp_wave + (size_t)(ks >> 1) * (size_t)(2 * NP * 1024);   // head ks >> 1 (13-bit immediate offsets)
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xn[ks][pl]) : "v"(lane16), "s"(p), "n"(((ks & 1) * NP + pl) * 1024) : "memory");
                    }
            }
            {   // v^T -> sV[pl][d = r][(wave, octet mm)][half h]
#pragma unroll
                for (int g = 0; g < 16; g++) qa[g] += ka[g];
                u32x4 vp[2][2];
#pragma unroll
                for (int mm = 0; mm < 2; mm++) pack_octet(qa, mm, vp[mm]);
#pragma unroll
                for (int mm = 0; mm < 2; mm++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(vw_addr), "v"(vp[mm][pl]), "n"(pl * HS * VROW + mm * 32) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // k, v^T of the head complete

            // ---- attention of this wave's 32 queries against the 256 keys of the head (model.py:58-60: no mask): the pipelined
            //      key-tile loop of attn256o_kernel (round 5, gpt_kernels_attn_tiles.h; exact running-maximum loop as its fallback) ----
            f32x16 o;
            float l_run = 0.f;
#if defined(MGPT_AB_ATTN_CLUMPED)
            attention_exact_tiles<T, NP, KROW, VROW, HS>(kr_addr, vr_addr, qf, sc2, o, l_run);
#else
            attention_tiles<T, NP, KROW, VROW, HS>(kr_addr, vr_addr, qf, lane, o, l_run);
#endif
            // ---- y planes of the head: o[g] = O[query r][d = tau(g, h)] / l, times the v projection's weight scale: register
            //      octet kk = k-step 2 hd + kk of the out-projection's B operand ----
            {
                const float inv = inv_scale / l_run;
#pragma unroll
                for (int g = 0; g < 16; g++) o[g] *= inv;
                if constexpr (LASTH) {
                    pack_octet(o, 0, xn[8]);
                    pack_octet(o, 1, xn[9]);
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
        };
#pragma unroll 1
        for (int hd = 0; hd < NH - 1; hd++) head(hd, std::false_type{});
        head(NH - 1, std::true_type{});

        // the planes of heads 0-3 must have landed before the first tail MFMA reads them.  Everything in flight here -- two steps
        // of ring pieces and the spill loads -- was issued one attention phase ago, so vmcnt(0) costs nothing and makes the first
        // tail steps' waits trivially safe.
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn[0][0]), "+v"(xn[0][1]), "+v"(xn[1][0]), "+v"(xn[1][1]), "+v"(xn[2][0]), "+v"(xn[2][1]), "+v"(xn[3][0]), "+v"(xn[3][1]),
                     "+v"(xn[4][0]), "+v"(xn[4][1]), "+v"(xn[5][0]), "+v"(xn[5][1]), "+v"(xn[6][0]), "+v"(xn[6][1]), "+v"(xn[7][0]), "+v"(xn[7][1]) : : "memory");

        // ---- tail: x <- x + y c_proj^T.  Output tile pairs (0, 1), (2, 3) over K = 160 in two q|k-shaped steps each, then tile 4 in
        //      one step (its two chains take the even / the odd k-steps); the residual rows are requested at a pseudo-head's first
        //      step.  Vector-memory operations in issue order: [8 loads | PW] [PW] 8 stores [8 loads | PW] [PW] 8 stores [4 loads | PW] 4 stores
        f32x4 xs[2][4];                                    // residual pieces: [tile of the pair][gq]
        auto request_tile = [&](unsigned char *xp, f32x4 (&dst)[4]) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[gq]) : "v"(xoff), "s"(xp), "n"(gq * 1024) : "memory");
        };
        auto store_tile = [&](unsigned char *xp, const f32x16 &acc, int w2) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(acc[4 * gq + e], inv_proj, xs[w2][gq][e]);
                asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(xoff), "v"(v), "s"(xp), "n"(gq * 1024) : "memory");
            }
        };
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            unsigned char *xp = xw + (size_t)t * 8192;     // tile 2t: chunks 8t .. 8t+3, tile 2t+1: chunks 8t+4 .. 8t+7 (+4 KiB)
            zero_chains();
            // first step: younger than the pieces waited for are one step's pieces and, for t = 1, the 8 stores of t = 0
            // (t = 0: nothing is in flight after the vmcnt(0) above)
            step(I0{}, I0{}, std::integral_constant<int, PW + 8>{}, [&]() { request_tile(xp, xs[0]); request_tile(xp + 4096, xs[1]); }, std::true_type{});
            // second step: ... the 8 residual loads, one step's pieces (and for t = 1 the 8 stores of t = 0, older than the loads)
            step(I0{}, I1{}, std::integral_constant<int, 2 * 8 + PW>{}, nothing, std::true_type{});
            // epilogue of the two tiles: x + acc / scale (the residual loads are older than this pseudo-head's two steps of pieces)
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]), "+v"(xs[1][0]), "+v"(xs[1][1]), "+v"(xs[1][2]), "+v"(xs[1][3])
                         : [n] "n"(2 * PW) : "memory");
            store_tile(xp, qa, 0);
            store_tile(xp + 4096, ka, 1);
        }
        {
            unsigned char *xp = xw + 16384;                // tile 4: chunks 16 .. 19
            zero_chains();
            // younger than the pieces waited for: one step's pieces and the 8 stores of the pair before
            step(I1{}, I0{}, std::integral_constant<int, PW + 8>{}, [&]() { request_tile(xp, xs[0]); }, std::false_type{});
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]) : [n] "n"(PW) : "memory");
#pragma unroll
            for (int g = 0; g < 16; g++) qa[g] += ka[g];
            store_tile(xp, qa, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no direct-to-LDS load may outlive the workgroup
}

}  // namespace fastk
}  // namespace mgpt
