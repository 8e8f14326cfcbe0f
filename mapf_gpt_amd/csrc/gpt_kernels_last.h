// The last transformer layer's attention block for the one token the head reads (token 255 of every row, model.py:186).
//
// Only the new row of token 255 leaves the last layer, and its attention needs K and V of all 256 tokens -- but never as
// matrices.  With xn_t = LayerNorm(x_t) and q = W_q xn_255 (head h: q_h, 32 numbers),
//     score_h[t] = q_h . (W_k,h xn_t) = (W_k,h^T q_h) . xn_t        =: u_h . xn_t
//     y_h        = sum_t p_h[t] (W_v,h xn_t) = W_v,h (sum_t p_h[t] xn_t)   =: W_v,h z_h
// so per row the work is four matrix-vector products (q, u, y, out-projection: 4 C^2 multiply-adds) plus the two
// 256 x C x n_head contractions u . xn_t and sum_t p xn_t -- ~1.3 M multiply-adds at C = 256 against the 34 M of the K and V
// projections that attn256_kernel<LAST> spends.  That is little enough for fp32 FMAs on the vector ALU: no operand
// planes, no rounding of the operands, and the result is closer to the fp32 reference than the split-fp16 path.
//
// One workgroup takes R = 4 rows.  The four matrix-vector products are done for the R rows together (every weight is
// loaded once per workgroup and used R times, so the loads' latency is paid once per R rows); the token phases run row by
// row: a token's normalised row lives in the registers of S threads (C / S features each; 256 S threads per workgroup, so
// that every SIMD has S waves to hide LDS latency with), the transposition to "thread = feature" for z goes through LDS,
// 64 tokens at a time.
// Weights are the fp32 masters: w_k = rows C..2C-1 of c_attn.weight as PyTorch keeps them ([out][in], model.py:52-58) and
// w_t = the transposes ([in][out]) of its q rows, its v rows and c_proj.weight, made once by transpose_kernel -- in both
// cases a thread reads 4 consecutive outputs of one weight row, so every load of a wave is 1 KiB contiguous.
// x is chunk-major (xt_off); x_last is the compact [rows_pad][C] buffer in the same layout: row b = new x of token 255 of
// row b; rows >= n_rows are written as zeros (the MLP kernel that follows works on whole blocks).  grid = rows_pad / R.
#pragma once
#include "gpt_kernels_fast.h"

namespace mgpt {
namespace fastk {

constexpr int kLast1R = 4;                      // rows per workgroup
// feature split: a token's normalised row is held by S threads (C / S registers each), the workgroup has 256 S threads
constexpr int last1_split(int C) { return C >= 128 ? 2 : 1; }   // (4 for C = 256 spills under the 128-register cap of 16 waves and is 0.6 ms slower)

// workgroup barrier for LDS traffic only: __syncthreads() also waits for every global load in flight (vmcnt(0)) -- here that would be the next
// row's tokens, requested ahead on purpose
__device__ __forceinline__ void last1_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int C, int R = kLast1R, bool TAIL = false>
constexpr int kLast1Lds = (64 * (C + 4) + R * (C / 32) * C + last1_split(C) * (C / 32) * 256 + 3 * R * C + 2 * last1_split(C) * 256 + (TAIL ? 7 * C : 0)) * 4;

// out[r] = sum_c wT[c][i] * vec_r[c] for the R rows and output i = tid (tid < C) (vec_r = vec + r * VS, + the head offset
// of output i when PER_HEAD).  Thread (ig = tid & 63, cq = tid >> 6) accumulates outputs 4 ig .. 4 ig + 3 over the cq-th
// of NQ slices of c; the NQ partial sums per output meet in `part` ([NQ][R][C] floats of LDS), thread i adds them in index
// order.  Ends with a barrier after which `part` may be reused.
template <int C, int R, int HS, int NQ, bool PER_HEAD>
__device__ __forceinline__ void last1_matvec(const float *__restrict__ wT, const float *vec, int VS, float *part, int tid, float (&out)[R])
{
    constexpr int CQ = C / NQ;
    static_assert(CQ % 4 == 0, "slices of whole float4s");
    const int ig = tid & 63, cq = tid >> 6;
    if (4 * ig < C) {
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; r++) { acc[r][0] = 0.0f; acc[r][1] = 0.0f; acc[r][2] = 0.0f; acc[r][3] = 0.0f; }
        const float *w = wT + (cq * CQ) * C + 4 * ig;
        const float *v = vec + cq * CQ + (PER_HEAD ? ((4 * ig) / HS) * C : 0);
#pragma unroll
        for (int c4 = 0; c4 < CQ / 4; c4++) {
            f32x4 wv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) wv[k] = *reinterpret_cast<const f32x4 *>(w + (4 * c4 + k) * C);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const f32x4 x4 = *reinterpret_cast<const f32x4 *>(v + r * VS + 4 * c4);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    acc[r][0] = fmaf(wv[k][0], x4[k], acc[r][0]); acc[r][1] = fmaf(wv[k][1], x4[k], acc[r][1]);
                    acc[r][2] = fmaf(wv[k][2], x4[k], acc[r][2]); acc[r][3] = fmaf(wv[k][3], x4[k], acc[r][3]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            f32x4 a; a[0] = acc[r][0]; a[1] = acc[r][1]; a[2] = acc[r][2]; a[3] = acc[r][3];
            *reinterpret_cast<f32x4 *>(part + (cq * R + r) * C + 4 * ig) = a;
        }
    }
    __syncthreads();
    if (tid < C) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            float a = 0.0f;
#pragma unroll
            for (int q = 0; q < NQ; q++) a += part[(q * R + r) * C + tid];
            out[r] = a;
        }
    }
    __syncthreads();
}

// One row: dst[o] = sum_k wT[k][o] * vec[k], o < O (wT row-major [K][O], vec and dst in LDS).  Wave cq takes the cq-th of NQ slices of
// k, lane ig the outputs 256 p + 4 ig .. + 3 of pass p; the NQ partial sums meet in `part` ([NQ][O] floats of LDS) and are added in
// index order.  Ends with a barrier: dst is complete, `part` may be reused.
template <int K, int O, int NQ>
__device__ __forceinline__ void last1_matvec1(const float *__restrict__ wT, const float *vec, float *part, float *dst, int tid)
{
    constexpr int KQ = K / NQ;
    static_assert(KQ % 4 == 0 && O % 4 == 0, "slices of whole float4s");
    const int ig = tid & 63, cq = tid >> 6;
#pragma unroll 1
    for (int o0 = 4 * ig; o0 < O; o0 += 256) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        const float *w = wT + (cq * KQ) * O + o0;
        const float *v = vec + cq * KQ;
#pragma unroll 4
        for (int k4 = 0; k4 < KQ / 4; k4++) {
            f32x4 wv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) wv[k] = *reinterpret_cast<const f32x4 *>(w + (4 * k4 + k) * O);
            const f32x4 x4 = *reinterpret_cast<const f32x4 *>(v + 4 * k4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a0 = fmaf(wv[k][0], x4[k], a0); a1 = fmaf(wv[k][1], x4[k], a1);
                a2 = fmaf(wv[k][2], x4[k], a2); a3 = fmaf(wv[k][3], x4[k], a3);
            }
        }
        f32x4 a; a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3;
        *reinterpret_cast<f32x4 *>(part + cq * O + o0) = a;
    }
    __syncthreads();
    for (int o = tid; o < O; o += 64 * NQ) {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; q++) a += part[q * O + o];
        dst[o] = a;
    }
    __syncthreads();
}

// LayerNorm of an LDS-resident row by wave 0 (two-pass, eps 1e-5, gain, no bias: model.py:20); ends with a barrier
template <int C>
__device__ __forceinline__ void last1_ln_row(const float *src, const float *__restrict__ gain, float *dst, int tid)
{
    if (tid < 64) {
        float v[(C + 63) / 64], s = 0.0f;
#pragma unroll
        for (int k = 0; k < (C + 63) / 64; k++) { v[k] = tid + 64 * k < C ? src[tid + 64 * k] : 0.0f; s += v[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / C);
        float q2 = 0.0f;
#pragma unroll
        for (int k = 0; k < (C + 63) / 64; k++) { v[k] -= mean; if (tid + 64 * k < C) q2 = fmaf(v[k], v[k], q2); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
        const float rstd = rsqrtf(q2 * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int k = 0; k < (C + 63) / 64; k++)
            if (tid + 64 * k < C) dst[tid + 64 * k] = v[k] * rstd * gain[tid + 64 * k];
    }
    __syncthreads();
}

// R rows per workgroup.  TAIL (needs R == 1; small launches, one environment): the workgroup goes on with the rest of the network for
// its row -- the last layer's MLP block (LayerNorm, c_fc, exact-erf GELU, c_proj, residual: model.py:84-89, 103) as fp32 matrix-vector
// products, ln_f and the tied lm_head (model.py:178, 186) -- and writes logits[row][V]: the last layer and the head are ONE launch.
// w_t then continues with the transposes of c_fc.weight ([C][4C]) and of mlp.c_proj.weight ([4C][C]).
template <int C, int HS, int R = kLast1R, bool TAIL = false>
__global__ __launch_bounds__(256 * last1_split(C))
void attn_last1_kernel(const float *__restrict__ x, const float *__restrict__ gain, const float *__restrict__ w_k,
                       const float *__restrict__ w_t, float *__restrict__ x_last, int n_rows, float scale_log2e,
                       const float *__restrict__ gain2 = nullptr, const float *__restrict__ gainf = nullptr,
                       const float *__restrict__ wte = nullptr, float *__restrict__ logits = nullptr, int V = 0)
{
    static_assert(C % 32 == 0 && C <= 256 && HS == 32, "one thread per feature, 32-wide heads");
    constexpr int S = last1_split(C), NW = 4 * S, CS = C / S;          // waves; features per thread in the token phases
    constexpr int NH = C / HS, T = 256, TQ = 64, XS = C + 4;
    static_assert(!TAIL || R == 1, "the tail runs one row per workgroup");
    static_assert(NW * R * C <= TQ * XS && (!TAIL || NW * 4 * C <= TQ * XS) && R <= 4 && CS % 8 == 0 && TQ % (4 * S) == 0, "the partial sums alias xT; one wave per row in phase 0");
    extern __shared__ float sm_last1[];
    float *xT = sm_last1;               // [TQ][XS]: normalised rows of 64 tokens, padded so that a wave's b128 row writes spread over the banks
    float *part = xT;                   // [NW][R][C]: partial sums of the matrix-vector products (never live together with xT)
    float *uS = xT + TQ * XS;           // [R][NH][C]: u_h of every row; row r's slice becomes its z_h
    float *pS = uS + R * NH * C;        // [S][NH][T]: partial scores; slice 0 then holds the probabilities; later the partial z
    float *v1 = pS + S * NH * T;        // [R][C]: xn_255
    float *qS = v1 + R * C;             // [R][C]: q * scale * log2(e)
    float *yS = qS + R * C;             // [R][C]: y
    float *lnS = yS + R * C;            // [2][S][T]: LayerNorm partial sums
    float *tl = lnS + 2 * S * T;        // TAIL: x_last row [C] | LN2 row [C] | hidden [4C] | final row [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = tid & 255, sp = tid >> 8;                          // token phases: token, feature slice [sp CS, sp CS + CS)
    const int64_t b0 = (int64_t)blockIdx.x * R;
    if (b0 >= n_rows) {
        if (!TAIL && tid < C) {
#pragma unroll
            for (int r = 0; r < R; r++) x_last[xt_off(b0 + r, tid, C)] = 0.0f;
        }
        return;
    }
    const int nr = (n_rows - b0 < R) ? (int)(n_rows - b0) : R;
    // The token phases keep a row in registers (thread = (token, feature slice): CS floats), and a row's 256 x C floats are the kernel's whole
    // memory traffic -- so the loads of row r + 1 are issued AHEAD of their use, by every wave
    // as soon as the z phase of row r has put the wave's tokens into LDS (round 6: issued at the top of a row's token phases they were a
    // serial 0.8 ms of the 1.7-ms launch at 12 288 rows -- profiles/r06_ab.txt, visit F).
    float xn[CS];
    auto load_row = [&](int r) {
        const int64_t m = (b0 + r) * T + tok;
        const float *xp = x + (((m >> 5) * (C >> 3) + sp * (CS / 8)) << 8) + ((m & 31) << 3);
#pragma unroll
        for (int c8 = 0; c8 < CS / 8; c8++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(xp + c8 * 256), a1 = *reinterpret_cast<const f32x4 *>(xp + c8 * 256 + 4);
#pragma unroll
            for (int i = 0; i < 4; i++) { xn[8 * c8 + i] = a0[i]; xn[8 * c8 + 4 + i] = a1[i]; }
        }
    };
    const float gain_tid = tid < C ? gain[tid] : 0.0f;                 // (read here: a load at the end of a row would wait for the row requested ahead)
    // ---- phase 0: xn_255 of the R rows, one wave per row (two-pass LayerNorm, eps 1e-5, gain, no bias: model.py:20) ----
    if (wave < R) {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        const bool on = 4 * lane < C && wave < nr;
        if (on) v = *reinterpret_cast<const f32x4 *>(x + xt_off((b0 + wave) * T + T - 1, 4 * lane, C));
        float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / C);
        float q2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] -= mean; q2 = fmaf(v[i], v[i], q2); }
        if (4 * lane >= C) q2 = 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
        const float rstd = rsqrtf(q2 * (1.0f / C) + 1e-5f);
        if (4 * lane < C) {
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gain + 4 * lane);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = on ? v[i] * rstd * g[i] : 0.0f;
            *reinterpret_cast<f32x4 *>(v1 + wave * C + 4 * lane) = v;
        }
    }
    __syncthreads();
    // ---- q[r][i] = W_q[i] . xn_255[r], scaled for exp2 ----
    {
        float q[R];
        last1_matvec<C, R, HS, NW, false>(w_t, v1, C, part, tid, q);
        if (tid < C) {
#pragma unroll
            for (int r = 0; r < R; r++) qS[r * C + tid] = q[r] * scale_log2e;
        }
    }
    __syncthreads();
    // ---- u[r][h][c] = sum_d q[r][h*HS + d] W_k[h*HS + d][c]: lane -> 4 consecutive c, wave -> heads wave, wave + NW .. ----
    if (4 * lane < C) {
#pragma unroll 1
        for (int h = wave; h < NH; h += NW) {
            float acc[R][4];
#pragma unroll
            for (int r = 0; r < R; r++) { acc[r][0] = 0.0f; acc[r][1] = 0.0f; acc[r][2] = 0.0f; acc[r][3] = 0.0f; }
            const float *w = w_k + (h * HS) * C + 4 * lane;
#pragma unroll 2
            for (int d4 = 0; d4 < HS / 4; d4++) {
                f32x4 wv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) wv[k] = *reinterpret_cast<const f32x4 *>(w + (4 * d4 + k) * C);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const f32x4 q4 = *reinterpret_cast<const f32x4 *>(qS + r * C + h * HS + 4 * d4);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        acc[r][0] = fmaf(wv[k][0], q4[k], acc[r][0]); acc[r][1] = fmaf(wv[k][1], q4[k], acc[r][1]);
                        acc[r][2] = fmaf(wv[k][2], q4[k], acc[r][2]); acc[r][3] = fmaf(wv[k][3], q4[k], acc[r][3]);
                    }
                }
            }
            // xn_t = xhat_t * gain enters the scores only through u . xn_t: the gain is folded into u
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gain + 4 * lane);
#pragma unroll
            for (int r = 0; r < R; r++) {
                f32x4 a; a[0] = acc[r][0] * g[0]; a[1] = acc[r][1] * g[1]; a[2] = acc[r][2] * g[2]; a[3] = acc[r][3] * g[3];
                *reinterpret_cast<f32x4 *>(uS + (r * NH + h) * C + 4 * lane) = a;
            }
        }
    }
    __syncthreads();
    // ---- the token phases, row by row: thread = (token tok, feature slice sp); xn below is the normalised row WITHOUT the gain ----
    load_row(0);            // (before the matrix-vector products above it costs them their registers: 424 bytes of scratch per lane at C = 256)
#pragma unroll 1
    for (int r = 0; r < nr; r++) {
        float *ur = uS + r * NH * C;
        // x_t -> xn_t (this thread's CS features of it; the raw row was requested ahead: load_row above)
        {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < CS; c++) s += xn[c];
            if (S > 1) {
                lnS[sp * T + tok] = s;
                __syncthreads();
                s = 0.0f;
#pragma unroll
                for (int q = 0; q < S; q++) s += lnS[q * T + tok];
            }
            const float mean = s * (1.0f / C);
            float q2 = 0.0f;
#pragma unroll
            for (int c = 0; c < CS; c++) { xn[c] -= mean; q2 = fmaf(xn[c], xn[c], q2); }
            if (S > 1) {
                lnS[(S + sp) * T + tok] = q2;
                __syncthreads();
                q2 = 0.0f;
#pragma unroll
                for (int q = 0; q < S; q++) q2 += lnS[(S + q) * T + tok];
            }
            const float rstd = rsqrtf(q2 * (1.0f / C) + 1e-5f);
#pragma unroll
            for (int c = 0; c < CS; c++) xn[c] *= rstd;             // ln_1's gain rides on u and on z (below)
        }
        // partial score_h[tok] over this thread's features
        {
            float sc[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) sc[h] = 0.0f;
            const float *up = ur + sp * CS;
#pragma unroll
            for (int c4 = 0; c4 < CS / 4; c4++) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(up + h * C + 4 * c4);
                    sc[h] = fmaf(u[0], xn[4 * c4], sc[h]); sc[h] = fmaf(u[1], xn[4 * c4 + 1], sc[h]);
                    sc[h] = fmaf(u[2], xn[4 * c4 + 2], sc[h]); sc[h] = fmaf(u[3], xn[4 * c4 + 3], sc[h]);
                }
            }
#pragma unroll
            for (int h = 0; h < NH; h++) pS[(sp * NH + h) * T + tok] = sc[h];
        }
        __syncthreads();
        // softmax over the 256 tokens, one head per wave at a time (token 255 sees every token: no mask)
        for (int h = wave; h < NH; h += NW) {
            float v[4], mx;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = 0.0f;
#pragma unroll
                for (int q = 0; q < S; q++) v[k] += pS[(q * NH + h) * T + lane + 64 * k];
            }
            mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = exp2f(v[k] - mx); sum += v[k]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int k = 0; k < 4; k++) pS[h * T + lane + 64 * k] = v[k] * inv;
        }
        __syncthreads();
        // z_h[c] = sum_t p_h[t] xn_t[c]: tokens -> features through LDS, TQ tokens per pass.  In a pass thread (pair, slice zs, head half hh)
        // takes the features 2 pair, 2 pair + 1, the zs-th TQ / S tokens and the heads of half hh: one 8-byte read per token, one 16-byte read
        // of p_h per four tokens and head, packed FMAs on the feature pair (round 6; until then one feature and all heads per thread: twice
        // the probability reads -- which bound the phase -- and twice the vector instructions per product).  Every (feature, head) keeps its
        // chain of products in the order it always had (tokens ascending inside a slice of TQ / S per pass, the S slices added in index
        // order): the logits do not change by a bit.
        constexpr int TG = TQ / S, NPAIR = C / 2, NHH = (NH + 1) / 2;
        static_assert(2 * S * NPAIR <= 256 * S && TG % 4 == 0 && S * NH * C <= TQ * XS, "thread map of the z phase; the partial z alias xT");
        const int zp = tid % NPAIR, zg = tid / NPAIR, zs = zg % S, h0 = (zg / S) * NHH;
        const bool zon = zg < 2 * S;
        f32x2 z2[NHH];
#pragma unroll
        for (int h = 0; h < NHH; h++) z2[h] = (f32x2){0.0f, 0.0f};
#pragma unroll 1
        for (int pass = 0; pass < T / TQ; pass++) {
            if ((wave & 3) == pass) {
                float *row = xT + lane * XS + sp * CS;
#pragma unroll
                for (int c4 = 0; c4 < CS / 4; c4++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = xn[4 * c4 + i];
                    *reinterpret_cast<f32x4 *>(row + 4 * c4) = v;
                }
                if (r + 1 < nr) load_row(r + 1);                    // this wave's registers are free: the next row's tokens, ahead of their use
            }
            last1_lds_barrier();
            if (zon) {
                const float *pp = pS + pass * TQ + zs * TG + h0 * T;
                const float *xr = xT + (zs * TG) * XS + 2 * zp;
#pragma unroll
                for (int t4 = 0; t4 < TG / 4; t4++) {
                    f32x2 xv[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) xv[i] = *reinterpret_cast<const f32x2 *>(xr + (4 * t4 + i) * XS);
#pragma unroll
                    for (int h = 0; h < NHH; h++) {
                        if (h0 + h < NH) {                          // (uniform per wave half at most: NH odd)
                            const f32x4 p = *reinterpret_cast<const f32x4 *>(pp + h * T + 4 * t4);
#pragma unroll
                            for (int i = 0; i < 4; i++) z2[h] = __builtin_elementwise_fma(xv[i], (f32x2){p[i], p[i]}, z2[h]);
                        }
                    }
                }
            }
            last1_lds_barrier();
        }
        // the S partial z of every head meet in LDS (xT is spent); u of this row is spent too: its slice now holds z
        {
            float *zpart = xT;                                      // [S][NH][C]
            if (zon) {
#pragma unroll
                for (int h = 0; h < NHH; h++)
                    if (h0 + h < NH) *reinterpret_cast<f32x2 *>(zpart + (zs * NH + h0 + h) * C + 2 * zp) = z2[h];
            }
            last1_lds_barrier();
            if (tid < C) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    float a = 0.0f;
#pragma unroll
                    for (int q = 0; q < S; q++) a += zpart[(q * NH + h) * C + tid];
                    ur[h * C + tid] = a * gain_tid;                 // z of xhat -> z of xn
                }
            }
            last1_lds_barrier();
        }
    }
    __syncthreads();
    // ---- y[r][i] = W_v[i] . z[r][head(i)] ----
    {
        float y[R];
        last1_matvec<C, R, HS, NW, true>(w_t + C * C, uS, NH * C, part, tid, y);
        if (tid < C) {
#pragma unroll
            for (int r = 0; r < R; r++) yS[r * C + tid] = y[r];
        }
    }
    __syncthreads();
    // ---- out-projection + residual: x_last[b][i] = x_255[i] + W_proj[i] . y (model.py:99-101, 129) ----
    {
        float o[R];
        last1_matvec<C, R, HS, NW, false>(w_t + 2 * C * C, yS, C, part, tid, o);
        if constexpr (!TAIL) {
            if (tid < C) {
                float xres[R];                              // the R residual values in flight together (rows past the end re-read the last one)
#pragma unroll
                for (int r = 0; r < R; r++) xres[r] = x[xt_off((b0 + (r < nr ? r : nr - 1)) * T + T - 1, tid, C)];
#pragma unroll
                for (int r = 0; r < R; r++) x_last[xt_off(b0 + r, tid, C)] = r < nr ? xres[r] + o[r] : 0.0f;
            }
        } else {
            if (tid < C) tl[tid] = x[xt_off(b0 * T + T - 1, tid, C)] + o[0];
        }
    }
    if constexpr (TAIL) {
        float *xl = tl, *xn2 = tl + C, *hS = tl + 2 * C, *xf = tl + 6 * C;
        __syncthreads();
        // ---- MLP block of the row: x + c_proj(GELU(c_fc(LayerNorm(x)))) ----
        last1_ln_row<C>(xl, gain2, xn2, tid);
        last1_matvec1<C, 4 * C, NW>(w_t + 3 * C * C, xn2, part, hS, tid);
        for (int j = tid; j < 4 * C; j += 64 * NW) { const float v = hS[j]; hS[j] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
        __syncthreads();
        last1_matvec1<4 * C, C, NW>(w_t + 3 * C * C + 4 * C * C, hS, part, xf, tid);
        if (tid < C) xf[tid] += xl[tid];
        __syncthreads();
        // ---- ln_f and the tied head ----
        last1_ln_row<C>(xf, gainf, xn2, tid);
        for (int v = tid; v < V; v += 64 * NW) {
            const float *wr = wte + (size_t)v * C;
            float acc = 0.0f;
#pragma unroll 8
            for (int c = 0; c < C; c++) acc = fmaf(wr[c], xn2[c], acc);
            logits[(size_t)b0 * V + v] = acc;
        }
    }
}

// dst[c][r] = src[r][c] for a rows x cols fp32 matrix (model build, once)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, int rows, int cols)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < rows * cols) dst[(i % cols) * rows + i / cols] = src[i];
}

}  // namespace fastk
}  // namespace mgpt
