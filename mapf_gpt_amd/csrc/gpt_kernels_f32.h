// gpt_kernels_f32.h -- exact-fp32 policy forward kernels (MGPT_PREC_F32), gfx950.
//
// This is the parity path: every matrix product runs on v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate; bit-for-bit an fmaf chain, 157 TFLOP/s peak on MI355X), so logits land within fp32
// round-off of the reference's PyTorch forward (model.py:167-189).  Layout decisions:
//   x, xn, y, h : [rows*256, C] token-major fp32 (residual stream stays fp32 in HBM)
//   q, k, v     : [rows, n_head, 256, hs] head-major, written that way by the QKV epilogue so the
//                 attention kernel streams whole [256, hs] panels contiguously
// MFMA operand convention used everywhere (wave64, 32x32x2): lane l = (r = l & 31, h = l >> 5)
// supplies A[i = r][k-slot h] and B[k-slot h][j = r]; the two k-slots of one instruction may be ANY
// two distinct k indices as long as A and B agree, which lets each half-wave read a contiguous run
// of k (16-byte vector loads) instead of the interleaved k, k+1 pairs.
// C/D: lane (r, h), register g holds D[row = (g & 3) + 8 * (g >> 2) + 4 * h][col = r].
#pragma once
#include "common.h"

namespace mgpt {
namespace f32k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 256;   // tokens per row (block_size, experiment_setup/config-*.py:13)
// T (kernel argument below): tokens per row of THIS call -- 256 on the hot path; GPT.forward accepts any T <= block_size (model.py:167-175), which
// mgpt_gpt_forward_t serves with these kernels.  For T = 256 every kernel executes the instructions it executed before T was an argument.

// ----- embedding: x = wte[idx] + wpe[pos]  (model.py:171-175) -----
__global__ __launch_bounds__(256) void embed_kernel(const uint8_t *__restrict__ tokens, const float *__restrict__ wte,
                                                    const float *__restrict__ wpe, float *__restrict__ x, int64_t n_tok,
                                                    int C, int T = kT)
{
    const int c4n = C >> 2;
    const int64_t total = n_tok * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = i / c4n;
        const int c4 = (int)(i - tok * c4n);
        const int t = T == kT ? (int)(tok & (kT - 1)) : (int)((unsigned)tok % (unsigned)T);
        const int id = tokens[tok];
        const float4 a = reinterpret_cast<const float4 *>(wte + (size_t)id * C)[c4];
        const float4 b = reinterpret_cast<const float4 *>(wpe + (size_t)t * C)[c4];
        reinterpret_cast<float4 *>(x)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// ----- LayerNorm over C, eps 1e-5, gain (+ bias when the checkpoint has one: `b` may be NULL; model.py:14-20): one wavefront per token -----
template <int kMaxV4>   // float4 slots per lane: C <= 256 * kMaxV4
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float *__restrict__ y, int64_t n_tok, int C, int64_t in_stride,
                                                        int64_t out_stride, const float *__restrict__ b = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const int c4n = C >> 2;
    const float4 *px = reinterpret_cast<const float4 *>(x + tok * in_stride);
    float4 v[kMaxV4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        v[k] = (i < c4n) ? px[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + 1e-5f);
    float4 *py = reinterpret_cast<float4 *>(y + tok * out_stride);
    const float4 *pw = reinterpret_cast<const float4 *>(w);
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) {
            const float4 g = pw[i];
            float4 o = make_float4((v[k].x - mean) * rstd * g.x, (v[k].y - mean) * rstd * g.y,
                                   (v[k].z - mean) * rstd * g.z, (v[k].w - mean) * rstd * g.w);
            if (b != nullptr) {                                  // (uniform) F.layer_norm(..., weight, bias, 1e-5)
                const float4 bb = reinterpret_cast<const float4 *>(b)[i];
                o = make_float4(o.x + bb.x, o.y + bb.y, o.z + bb.z, o.w + bb.w);
            }
            py[i] = o;
        }
    }
}

// ----- GEMM: out[M,N] = epilogue(A[M,K] @ W[N,K]^T), W = nn.Linear weight as stored (model.py:29,31,79,81) -----
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_QKV = 3 };

struct EpiArgs {
    int C, n_head, hs;      // EPI_QKV: scatter into [3][rows][n_head][256][hs]
    int64_t plane;          // EPI_QKV: elements per q/k/v plane = M * C
    const float *bias = nullptr;   // nn.Linear bias [N] of a bias = True checkpoint (model.py:29,31,79,81), else NULL
    int T = kT;                    // EPI_QKV: tokens per row
    int64_t m_valid = INT64_MAX;   // EPI_QKV: tokens of the call (M is padded to the 128-token tile when rows * T is not a multiple of it: no scatter from the padding)
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, const float *__restrict__ Wt,
                                                       float *__restrict__ out, int M, int N, int K, int n_tiles_n,
                                                       EpiArgs ep)
{
    constexpr int BM = 128, BK = 32, LDS = 36;                 // LDS row stride in floats (pad 4: conflict-free b128 reads)
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_V4 = BM * (BK / 4) / 256;                  // float4 per thread for the A tile (4)
    constexpr int B_V4 = BN * (BK / 4) / 256;                  // (2, 4 or 5)
    static_assert(WM * WN == 4 && TM * WM * 32 == BM && TN * WN * 32 == BN, "tile config");
    __shared__ __attribute__((aligned(16))) float sA[BM * LDS];
    __shared__ __attribute__((aligned(16))) float sB[BN * LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int mt = blockIdx.x / n_tiles_n, nt = blockIdx.x - mt * n_tiles_n;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;

    f32x4 ra[A_V4], rb[B_V4];
    const float *a_src = A + (m0 + (tid >> 3)) * K + (tid & 7) * 4;        // + i*32 rows, + kt*BK
    const float *b_src = Wt + (size_t)(n0 + (tid >> 3)) * K + (tid & 7) * 4;
#define MGPT_LOAD_TILES(kt_)                                                                        \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < A_V4; i++)                                            \
            ra[i] = *reinterpret_cast<const f32x4 *>(a_src + (size_t)(32 * i) * K + (kt_) * BK);   \
        _Pragma("unroll") for (int i = 0; i < B_V4; i++)                                            \
            rb[i] = *reinterpret_cast<const f32x4 *>(b_src + (size_t)(32 * i) * K + (kt_) * BK);   \
    }
    const int nk = K / BK;
    MGPT_LOAD_TILES(0);
    for (int kt = 0; kt < nk; kt++) {
        __syncthreads();                                        // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < A_V4; i++) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<f32x4 *>(&sA[row * LDS + c4 * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_V4; i++) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            *reinterpret_cast<f32x4 *>(&sB[row * LDS + c4 * 4]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) MGPT_LOAD_TILES(kt + 1);               // in flight during the MFMAs below
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++)
                a[i] = *reinterpret_cast<const f32x4 *>(&sA[((wm * TM + i) * 32 + r) * LDS + 16 * h + 4 * qd]);
#pragma unroll
            for (int j = 0; j < TN; j++)
                b[j] = *reinterpret_cast<const f32x4 *>(&sB[((wn * TN + j) * 32 + r) * LDS + 16 * h + 4 * qd]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

#undef MGPT_LOAD_TILES
    // epilogue: for a fixed register the 32 lanes of a half-wave hold 32 consecutive columns of one row
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            const float bn = ep.bias != nullptr ? ep.bias[n] : 0.f;
            int64_t qkv_col = 0;                                // EPI_QKV: column-only part of the scatter offset
            if (EPI == EPI_QKV) {                               // model.py:50-53
                const int which = n / ep.C, cc = n - which * ep.C;
                const int head = cc / ep.hs, d = cc - head * ep.hs;
                qkv_col = (int64_t)which * ep.plane + (int64_t)head * ep.T * ep.hs + d;
            }
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int64_t m = m0 + (wm * TM + i) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                float v = acc[i][j][g];
                if (ep.bias != nullptr) v += bn;                 // (uniform)
                if (EPI == EPI_STORE) {
                    out[m * N + n] = v;
                } else if (EPI == EPI_RESID) {
                    out[m * N + n] += v;                        // x <- x + proj(...), model.py:102-103
                } else if (EPI == EPI_GELU) {
                    out[m * N + n] = gelu_erf(v);               // nn.GELU() exact erf, model.py:80,86
                } else {
                    int64_t b;
                    int t;
                    if (ep.T == kT) { b = m >> 8; t = (int)(m & (kT - 1)); }                     // (uniform) the hot shape: no division
                    else { const unsigned bu = (unsigned)m / (unsigned)ep.T; b = bu; t = (int)((unsigned)m - bu * (unsigned)ep.T); }   // (token counts fit 32 bits: the workspaces do)
                    if (m < ep.m_valid) out[qkv_col + (b * ep.n_head * ep.T + t) * ep.hs] = v;
                }
            }
        }
}

// ----- non-causal attention (model.py:58-60, is_causal=False), one workgroup per (row, head) -----
// S^T = K Q^T per 32-query tile so that a lane owns ONE query column: the softmax reduction over the
// keys is in-lane (+1 exchange with lane^32) and the probabilities already sit in the B-operand
// position for O^T = V^T P^T -- nothing moves across lanes and S never leaves registers.
// Keys are walked in tiles of 32 with a running (max, sum) pair (exact softmax, flash-style
// rescaling), which keeps the kernel at ~100 VGPRs instead of holding all 256x32 scores.
// T tokens per row (256 on the hot path).  T % 32 != 0: the last tile's missing keys and queries are read from row T - 1 (in bounds, finite), the missing
// keys' scores are -inf before the running maximum (probability exactly 0, as if they were not there), the missing queries are not stored.
template <int HS>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                       const float *__restrict__ v, float *__restrict__ y, int n_head,
                                                       float scale, int T = kT)
{
    constexpr int HH = HS / 2, DT = HS / 32;
    const int KT = (T + 31) >> 5;
    const bool ragged = (T & 31) != 0;                          // (uniform)
    const int bh = blockIdx.x;
    const int b = bh / n_head, head = bh - b * n_head;
    const int C = n_head * HS;
    const float *Q = q + (size_t)bh * T * HS, *Kp = k + (size_t)bh * T * HS, *V = v + (size_t)bh * T * HS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;

    for (int qt = wave; qt < KT; qt += 4) {
        float4 qf[HH / 4];                                      // Q[query r][h*HH .. h*HH+HH)
        const int qi = min(qt * 32 + r, T - 1);
#pragma unroll
        for (int i = 0; i < HH / 4; i++)
            qf[i] = *reinterpret_cast<const float4 *>(Q + (size_t)qi * HS + h * HH + 4 * i);
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int g = 0; g < 16; g++) o[dt][g] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
        for (int kt = 0; kt < KT; kt++) {
            f32x16 s;
#pragma unroll
            for (int g = 0; g < 16; g++) s[g] = 0.f;
            const float *krow = Kp + (size_t)min(kt * 32 + r, T - 1) * HS + h * HH;
#pragma unroll
            for (int i = 0; i < HH / 4; i++) {
                const float4 kf = *reinterpret_cast<const float4 *>(krow + 4 * i);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[i].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[i].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[i].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[i].w, s, 0, 0, 0);
            }
            // s[g] = S[query r][key = kt*32 + (g&3) + 8*(g>>2) + 4*h]
            const bool last_ragged = ragged && kt == KT - 1;     // (uniform)
            if (last_ragged) {
#pragma unroll
                for (int g = 0; g < 16; g++)
                    if (kt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h >= T) s[g] = -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int g = 1; g < 16; g++) mx = fmaxf(mx, s[g]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf((m_run - m_new) * scale);   // 0 on the first tile (m_run = -inf)
            float psum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                s[g] = expf((s[g] - m_new) * scale);
                psum += s[g];
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) o[dt][g] *= alpha;
            const float *vbase = V + (size_t)(kt * 32 + 4 * h) * HS + r;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int key = (g & 3) + 8 * (g >> 2);         // + kt*32 + 4*h folded into vbase
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    float vv;
                    if (last_ragged) vv = V[(size_t)min(kt * 32 + 4 * h + key, T - 1) * HS + r + dt * 32];   // (its probability is 0)
                    else vv = vbase[(size_t)key * HS + dt * 32];
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[g], o[dt], 0, 0, 0);
                }
            }
        }
        const float inv = 1.0f / l_run;
        // o[dt][g] = O[query r][d = dt*32 + (g&3) + 8*(g>>2) + 4*h]  -> y[b, t, head*HS + d]  (model.py:68)
        if (qt * 32 + r >= T) continue;                         // (a query the row does not have: nothing to store; no barrier below)
        float *yrow = y + ((size_t)b * T + qt * 32 + r) * C + head * HS;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                const float4 w4 = make_float4(o[dt][4 * gg] * inv, o[dt][4 * gg + 1] * inv, o[dt][4 * gg + 2] * inv,
                                              o[dt][4 * gg + 3] * inv);
                *reinterpret_cast<float4 *>(yrow + dt * 32 + 8 * gg + 4 * h) = w4;
            }
    }
}

// ----- final LayerNorm + tied lm_head on the last position only (model.py:178,186) -----
// row i reads C floats at x + i * row_stride + row_offset  (full residual stream: stride 256*C, offset 255*C;
// compact last-token buffer: stride C, offset 0)
__global__ __launch_bounds__(64) void head_kernel(const float *__restrict__ x, const float *__restrict__ lnf,
                                                  const float *__restrict__ wte, float *__restrict__ logits, int C,
                                                  int V, int64_t row_stride, int64_t row_offset, const float *__restrict__ lnf_b = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xn = reinterpret_cast<float *>(smem);
    const int row = blockIdx.x, lane = threadIdx.x;
    const float *px = x + (int64_t)row * row_stride + row_offset;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += px[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float qv = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = px[c] - mean; qv += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qv += __shfl_xor(qv, o);
    const float rstd = rsqrtf(qv / (float)C + 1e-5f);
    for (int c = lane; c < C; c += 64) {
        float o = (px[c] - mean) * rstd * lnf[c];
        if (lnf_b != nullptr) o += lnf_b[c];                    // ln_f.bias of a bias = True checkpoint (lm_head itself never has one, model.py:131)
        xn[c] = o;
    }
    __syncthreads();
    for (int vtok = lane; vtok < V; vtok += 64) {
        const float *wr = wte + (size_t)vtok * C;
        float acc = 0.f;
        for (int c = 0; c < C; c++) acc = fmaf(wr[c], xn[c], acc);
        logits[(size_t)row * V + vtok] = acc;
    }
}

}  // namespace f32k
}  // namespace mgpt
