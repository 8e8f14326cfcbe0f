// gpt_kernels_fast.h -- 16-bit-MFMA policy forward kernels (MGPT_PREC_F16X3 and MGPT_PREC_BF16), gfx950.
//
// Split precision (F16X3): every fp32 operand v is carried as two fp16 planes, hi = fp16(v) and
// lo = fp16(v - hi) (22 significand bits together); a product is the 3-term expansion
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (dropped a_lo*b_lo ~ 2^-22 |ab|)
// issued as three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator.  fp16 subnormal inputs are
// honoured by the matrix core on gfx950 (probed: tools/bench_probes/probe_mfma.hip, profiles/r01_probe_mfma.txt), and
// weights are pre-scaled by a power of two so their lo parts stay normal.  That buys ~fp32 accuracy
// (logit error ~1e-6, tested against the reference goldens at 1e-5) at 1/3 of the 2.5 PFLOP/s fp16 rate
// instead of the 157 TFLOP/s fp32-MFMA rate.  BF16 mode = one plane, one pass (the reference's autocast mode).
//
// MFMA conventions (wave64, 32x32x16): lane l = (r = l & 31, h = l >> 5) supplies 8 consecutive
// 16-bit k-slots of row r for BOTH operands; C/D: lane (r, h), register g holds
// D[(g & 3) + 8 * (g >> 2) + 4 * h][r].  Which matrix is passed as the first operand decides whether
// tokens run along registers ("natural") or along lanes ("swapped"); every epilogue picks the
// orientation that makes its stores 8 or 16 bytes wide.
#pragma once
#include "common.h"

namespace mgpt {
namespace fastk {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 256;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct F16T {
    static __device__ __forceinline__ uint16_t cvt(float v) { _Float16 x = (_Float16)v; return __builtin_bit_cast(uint16_t, x); }
    static __device__ __forceinline__ float back(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)      // (probe only: the 16 x 16 x 32 form)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    }
};
struct BF16T {
    static __device__ __forceinline__ uint16_t cvt(float v)
    {   // round-to-nearest-even
        unsigned u = __builtin_bit_cast(unsigned, v);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    static __device__ __forceinline__ float back(uint16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
};

// hi/lo split of 4 consecutive values -> two 8-byte packets
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <class T, int NP>
__device__ __forceinline__ void split4(const float v[4], u32x2 &hi, u32x2 &lo)
{
    if constexpr (NP == 1 && std::is_same<T, BF16T>::value) {
        // single bf16 plane: the hardware's packed round-to-nearest-even conversion (same rounding as BF16T::cvt)
        hi[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2));
        hi[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2));
        lo[0] = 0u; lo[1] = 0u;
        return;
    }
    if constexpr (NP == 2 && std::is_same<T, F16T>::value) {
        // split-fp16: 6 VALU for the four values -- two packed conversions for hi, then lo = RNE fp16(v - hi) by
        // v_fma_mixlo / mixhi_f16 (fma(hi as fp16, -1.0, v) rounded once to fp16; v - hi is exact in fp32, so this is the value
        // the chain below gives, in half the instructions).  The fence keeps each value ONE rounded fp32 first (see below).
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { x[i] = v[i]; asm("" : "+v"(x[i])); }
        hi[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x[0], x[1]}, h2));
        hi[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x[2], x[3]}, h2));
        unsigned l0, l1;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi[0]), "v"(x[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi[0]), "v"(x[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi[1]), "v"(x[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi[1]), "v"(x[3]));
        lo[0] = l0; lo[1] = l1;
        return;
    }
    uint16_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // The value must exist as ONE rounded fp32 before it is split.  Without this fence hipcc contracts the
        // caller's last multiply into the conversions (v_fma_mixlo_f16: fp16(a*b) from the unrounded product) for
        // the residual but not for the stored hi part; when fp32(a*b) sits exactly on an fp16 tie (1 in 2^13
        // values) the two roundings pick different neighbours and hi+lo is off by a whole fp16 ulp.
        float x = v[i];
        asm("" : "+v"(x));   // not volatile: must stay freely schedulable
        a[i] = T::cvt(x);
        b[i] = (NP == 2) ? T::cvt(x - T::back(a[i])) : (uint16_t)0;
    }
    hi[0] = (unsigned)a[0] | ((unsigned)a[1] << 16); hi[1] = (unsigned)a[2] | ((unsigned)a[3] << 16);
    lo[0] = (unsigned)b[0] | ((unsigned)b[1] << 16); lo[1] = (unsigned)b[2] | ((unsigned)b[3] << 16);
}

// acc += A*B with NP-plane operands (NP == 2: 3-term split product, NP == 1: single pass)
template <class T, int NP>
__device__ __forceinline__ f32x16 mma(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c)
{
    if (NP == 2) {
        c = T::mfma(a[1], b[0], c);     // small terms first, then the leading one
        c = T::mfma(a[0], b[1], c);
    }
    return T::mfma(a[0], b[0], c);
}

// ---------------------------------------------------------------------------------------------
// weight packing: fp32 W[N][K] -> NP planes of 16-bit [N][K] (K contiguous), multiplied by `scale`
// ---------------------------------------------------------------------------------------------
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float *__restrict__ w, uint16_t *__restrict__ hi,
                                                          uint16_t *__restrict__ lo, int64_t n4, float scale)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4 *>(w)[i];
    const float s[4] = {v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
    u32x2 a, b;
    split4<T, NP>(s, a, b);
    reinterpret_cast<u32x2 *>(hi)[i] = a;
    if (NP == 2) reinterpret_cast<u32x2 *>(lo)[i] = b;
}

// ---------------------------------------------------------------------------------------------
// embedding + LayerNorm statistics of the fresh residual rows: x = wte[idx] + wpe[pos] (model.py:171-175),
// stats[tok] = (mean, rstd) over C with eps 1e-5 (model.py:20).  One wavefront per token.
// ---------------------------------------------------------------------------------------------
template <int kMaxV4>
__global__ __launch_bounds__(256) void embed_stats_kernel(const uint8_t *__restrict__ tokens, const float *__restrict__ wte,
                                                          const float *__restrict__ wpe, float *__restrict__ x,
                                                          float2 *__restrict__ stats, int64_t n_tok, int C)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const int c4n = C >> 2, t = (int)(tok & (kT - 1)), id = tokens[tok];
    const f32x4 *pa = reinterpret_cast<const f32x4 *>(wte + (size_t)id * C);
    const f32x4 *pb = reinterpret_cast<const f32x4 *>(wpe + (size_t)t * C);
    f32x4 *px = reinterpret_cast<f32x4 *>(x + tok * C);
    f32x4 v[kMaxV4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) {
            v[k] = pa[i] + pb[i];
            px[i] = v[k];
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) {
            const f32x4 d = v[k] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) stats[tok] = make_float2(mean, rsqrtf(q / (float)C + 1e-5f));
}

// stats only (used where the producing GEMM cannot see whole rows)
template <int kMaxV4>
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ x, float2 *__restrict__ stats,
                                                        int64_t n_tok, int C)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tok) return;
    const int c4n = C >> 2;
    const f32x4 *px = reinterpret_cast<const f32x4 *>(x + tok * C);
    f32x4 v[kMaxV4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) { v[k] = px[i]; s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV4; k++) {
        const int i = lane + 64 * k;
        if (i < c4n) { const f32x4 d = v[k] - mean; q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) stats[tok] = make_float2(mean, rsqrtf(q / (float)C + 1e-5f));
}

// last-layer shortcut of the packed-GEMM path: compact residual rows x_last[b] = x[b, 255, :] for b < rows, zeros for the
// padding rows up to rows_pad (they go through the compact GEMMs and the MLP and must stay finite whatever ran before)
// (tiled: both x and x_last are chunk-major, xt_off below)
__device__ __forceinline__ int64_t xt_off(int64_t m, int n, int C);
// fold (small launches): n_fold compact partial sums (the last-token contributions of a head-parallel attn_block_kernel<LAST>, in
// x_last's own layout, fold_stride floats apart) are added in index order
__global__ __launch_bounds__(256) void gather_last_kernel(const float *__restrict__ x, float *__restrict__ x_last, int rows,
                                                          int rows_pad, int C, int tiled, const float *__restrict__ fold = nullptr,
                                                          int n_fold = 0, int64_t fold_stride = 0)
{
    const int c4n = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows_pad * c4n) return;
    const int b = (int)(i / c4n), c4 = (int)(i - (int64_t)b * c4n);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int64_t m = (int64_t)b * kT + kT - 1;
    const int64_t o = tiled ? xt_off(b, 4 * c4, C) : (int64_t)b * C + 4 * c4;
    if (b < rows) {
        v = *reinterpret_cast<const f32x4 *>(x + (tiled ? xt_off(m, 4 * c4, C) : m * C + 4 * c4));
        for (int p = 0; p < n_fold; p++) v += *reinterpret_cast<const f32x4 *>(fold + (size_t)p * fold_stride + o);
    }
    *reinterpret_cast<f32x4 *>(x_last + o) = v;
}

// chunk-major rows -> plain rows (the compact last-token matrix in front of the head kernel)
__global__ __launch_bounds__(256) void untile_rows_kernel(const float *__restrict__ xt, float *__restrict__ out, int rows, int C)
{
    const int c4n = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * c4n) return;
    const int b = (int)(i / c4n), c4 = (int)(i - (int64_t)b * c4n);
    *reinterpret_cast<f32x4 *>(out + (int64_t)b * C + 4 * c4) = *reinterpret_cast<const f32x4 *>(xt + xt_off(b, 4 * c4, C));
}

// etab[position][token][C] = wpe[position] + wte[token] (model.py:171-175: the same fp32 addition, done once per checkpoint): the rows
// attn256q_kernel<.., EMB> starts layer 0 from
__global__ __launch_bounds__(256) void embed_table_kernel(const float *__restrict__ wte, const float *__restrict__ wpe, float *__restrict__ etab, int C, int n_tok)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // (position, token, 4 features)
    const int c4n = C >> 2;
    if (i >= (int64_t)kT * n_tok * c4n) return;
    const int c4 = (int)(i % c4n), id = (int)((i / c4n) % n_tok), t = (int)(i / ((int64_t)c4n * n_tok));
    *reinterpret_cast<f32x4 *>(etab + i * 4) = *reinterpret_cast<const f32x4 *>(wte + (size_t)id * C + 4 * c4) + *reinterpret_cast<const f32x4 *>(wpe + (size_t)t * C + 4 * c4);
}

// x = wte[token] + wpe[position] (model.py:171-175) written chunk-major: one workgroup per 32-token tile, a wave writes
// whole 1-KiB chunks (lane (r, h): token r, columns 8 c + 4 h .. + 3); the embedding rows come from L2
__global__ __launch_bounds__(256) void embed_tiled_kernel(const uint8_t *__restrict__ tokens, const float *__restrict__ wte,
                                                          const float *__restrict__ wpe, float *__restrict__ x, int C)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int64_t tok = (int64_t)blockIdx.x * 32 + r;
    const int t = (int)(tok & (kT - 1)), id = tokens[tok];
    const float *pa = wte + (size_t)id * C + 4 * h, *pb = wpe + (size_t)t * C + 4 * h;
    float *px = x + (int64_t)blockIdx.x * 32 * C + r * 8 + 4 * h;
    // eight chunks per wave in flight (C = 256: the whole share of the wave), streaming stores (0.84 -> 0.75 ms per 12 288 rows, round 4)
#pragma unroll 8
    for (int c = wave; c < (C >> 3); c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(pa + 8 * c) + *reinterpret_cast<const f32x4 *>(pb + 8 * c);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(px + c * 256));
    }
}

// ---------------------------------------------------------------------------------------------
// GEMM: D[M,N] = A[M,K] @ W[N,K]^T on 16-bit MFMA planes, 128 x BN output tile per workgroup,
// K walked in tiles of 32 staged through LDS (register prefetch of the next tile during the MFMAs).
//   PRO_LN     : A = LayerNorm(x) built on the fly from fp32 x, per-row (mean, rstd) and the gain vector
//   PRO_PLANES : A given as NP 16-bit planes
//   EPI_*      : see below; all but EPI_VT run "swapped" (tokens along lanes) so each lane owns 4
//                consecutive output columns per register quad
// ---------------------------------------------------------------------------------------------
enum { PRO_LN = 0, PRO_PLANES = 1 };
enum { EPI_QK = 0, EPI_VT = 1, EPI_RESID = 2, EPI_GELU = 3 };

// GELU by table: gelu(v) = v Phi(v), Phi = standard normal CDF = (1 + erf(v / sqrt 2)) / 2 (model.py:86, exact-erf GELU).
// Phi is tabulated on [-6, 6) in steps of 1/256 as pairs (Phi(v_i), Phi(v_i+1) - Phi(v_i)) and interpolated linearly:
// |error in Phi| <= h^2 / 8 max|Phi''| = 4.6e-7, i.e. <= 4.6e-7 |v| in gelu (the rational approximation used elsewhere has
// 1.6e-6); beyond +-6 the end entries apply (Phi = 1e-9 / 1 - 1e-9).  24 KiB: the LDS goes to a deeper weight ring.  8 VALU + one 8-byte LDS gather per value instead of
// 19 VALU: on this kernel the VALU port, which the MFMAs share, is the scarce resource (section 3 of DESIGN.md).
constexpr int kGeluLutN = 3072;                // entries (float2 each: 24 KiB of LDS)
constexpr float kGeluLutScale = 256.0f, kGeluLutBias = 1536.0f;

struct GemmArgs {
    // A
    const float *x; const float2 *stats; const float *gain;         // PRO_LN
    const uint16_t *a_hi, *a_lo;                                    // PRO_PLANES, [M][K]
    // W planes [N][K] (pre-scaled), result multiplier = 1/scale
    const uint16_t *w_hi, *w_lo;
    float out_scale;
    int M, N, K, n_tiles_n, n_base;                                  // n_base: first output column of this launch
    // outputs
    float *x_out; float2 *stats_out;                                 // EPI_RESID: x_out[m][n] += d ; optional new row stats
    uint16_t *o_hi, *o_lo;                                           // EPI_QK: q|k planes, EPI_VT: v^T planes, EPI_GELU: h planes
    int C, n_head, hs;
    int64_t plane;                                                   // EPI_QK: elements between the q and the k plane (= M*C)
    int o_pk;                                                        // EPI_GELU: write the hidden planes in PK layout (o_hi = base)
    int chunk_major;                                                 // EPI_QK / EPI_VT: chunk-major q|k and v^T planes (below)
    const float2 *gelu_lut;                                          // EPI_GELU in gemm_pk_kernel: the Phi table (kGeluLutN pairs) or NULL
    int x_tiled;                                                     // EPI_RESID: x_out is chunk-major (xt_off) instead of row-major
    // LayerNorm folded into the GEMM chain (one-plane mode, C > 256; gpt_fast.hip `ln_fold`): the A operand is the RAW residual row
    // in operand planes and W carries ln.weight, so acc[m][n] = sum_k x[m][k] W'[n][k]; the consumer's epilogue forms
    // rstd[m] * (acc - mean[m] * colsum[n]) = sum_k LayerNorm(x)[m][k] W[n][k].  No kernel reads x to normalise it.
    const float2 *ln_stats;                                          // consumers (EPI_QK / EPI_VT / EPI_GELU): (mean, rstd) per row, or NULL
    const float *colsum;                                             // consumers: sum over k of the packed (rounded, scaled) W'[n][k]
    uint16_t *raw_out;                                               // EPI_RESID: operand planes (PK, K = N) of the new residual rows, or NULL
    float2 *rsum_out;                                                // EPI_RESID: (sum, sum of squares) of the new rows per 128-column block: [N / 128][M]
    // The planes carry x - shift[row]: the row's mean at the LayerNorm BEFORE (ln_finalize_kernel keeps it), so that what is rounded to
    // the operand type is centred up to the drift of the mean across one residual update, not the raw value (a residual stream with
    // |mean| >> std would otherwise lose |mean| / std of its bits); the consumer's mean term uses mean - shift.
    const float *shift; int shift_stride, shift_offset;              // EPI_RESID: shift of row m = shift[m * shift_stride + shift_offset]
};

// Chunk-major residual stream: x[M][C] stored as [M / 32][C / 8][32 tokens][8 floats].  A wave whose lane (r, h) owns token r
// and the 4 floats at columns 8c + 4h reads or writes 64 x 16 B = 1 KiB CONTIGUOUS per instruction (8 full cache lines)
// instead of 32 row pieces of 32 B in 32 different lines -- the access shape of every kernel that keeps "lane = token".
__device__ __forceinline__ int64_t xt_off(int64_t m, int n, int C) { return (((m >> 5) * (C >> 3) + (n >> 3)) << 8) + ((m & 31) << 3) + (n & 7); }

// erf(x) ~= x P(x^2) / Q(x^2) on [-4, 4] (|erf| = 1 - 1.5e-8 beyond): max abs error 4.5e-7 in fp32 arithmetic
// (checked against scipy over [-6, 6]); ~15 instructions instead of ocml erff's ~40 -- the GELU sits in GEMM epilogues.
__device__ __forceinline__ float erf_fast(float x)
{
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    p *= x;
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return p * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f)); }

// value of the other half-wave's lane (r <-> r + 32) combined with this lane's: v_permlane32_swap is a VALU instruction; the
// __shfl_xor(v, 32) it replaces compiles to ds_bpermute_b32, which joins the K / V fragment reads in lgkmcnt and waits for
// them.  Inline asm: the builtin's second result comes back as a copy of the first with this hipcc; s_nop 1 = the two wait
// states between a VALU write and v_permlane*_swap reading it.
__device__ __forceinline__ void half_swap32(float v, float &lower, float &upper)
{
    lower = v; upper = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lower), "+v"(upper));
}
__device__ __forceinline__ float half_max32(float v) { float a, b; half_swap32(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum32(float v) { float a, b; half_swap32(v, a, b); return a + b; }

// GELU with the 1/sqrt(2) and the powers of two folded into the rational's coefficients:
// gelu(v) = hv + hv * erf(v / sqrt 2), hv = v / 2, erf(u / sqrt 2) ~= u N(u^2) / D(u^2) for |u| <= 4 sqrt 2 (1 beyond).
// max abs error 1.6e-6 over [-9, 9] in fp32 arithmetic; 17 VALU instructions.
__device__ __forceinline__ float gelu_folded(float v)
{
    const float u = __builtin_amdgcn_fmed3f(v, -5.6568542f, 5.6568542f);
    const float z = u * u;
    float p = -3.011990121e-12f;
    p = fmaf(p, z, 6.122398825e-10f);
    p = fmaf(p, z, -9.285302079e-08f);
    p = fmaf(p, z, -5.031512342e-06f);
    p = fmaf(p, z, -1.299292147e-04f);
    p = fmaf(p, z, -1.044608780e-03f);
    p = fmaf(p, z, -1.138161432e-02f);
    p *= u;
    float q = -9.103794904e-07f;
    q = fmaf(q, z, -2.667175691e-05f);
    q = fmaf(q, z, -4.207067436e-04f);
    q = fmaf(q, z, -3.686664584e-03f);
    q = fmaf(q, z, -1.426473905e-02f);
    const float e = p * __builtin_amdgcn_rcpf(q);
    const float hv = 0.5f * v;
    return fmaf(hv, e, hv);
}


// gelu_folded on two values with packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32): the same operations in the same order,
// so the same results.  Pays in the one-wave-per-SIMD GEMM epilogue, where an instruction costs ~5 cycles whatever it does
// (tools/bench_probes/probe_pk.hip); the register-tight fused kernels keep the scalar form (its constants are inline literals).
__device__ __forceinline__ f32x2 gelu_folded2(f32x2 v)
{
    const f32x2 u = {__builtin_amdgcn_fmed3f(v[0], -5.6568542f, 5.6568542f), __builtin_amdgcn_fmed3f(v[1], -5.6568542f, 5.6568542f)};
    const f32x2 z = u * u;
    f32x2 p = (f32x2)(-3.011990121e-12f);
    p = __builtin_elementwise_fma(p, z, (f32x2)(6.122398825e-10f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-9.285302079e-08f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-5.031512342e-06f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-1.299292147e-04f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-1.044608780e-03f));
    p = __builtin_elementwise_fma(p, z, (f32x2)(-1.138161432e-02f));
    p *= u;
    f32x2 q = (f32x2)(-9.103794904e-07f);
    q = __builtin_elementwise_fma(q, z, (f32x2)(-2.667175691e-05f));
    q = __builtin_elementwise_fma(q, z, (f32x2)(-4.207067436e-04f));
    q = __builtin_elementwise_fma(q, z, (f32x2)(-3.686664584e-03f));
    q = __builtin_elementwise_fma(q, z, (f32x2)(-1.426473905e-02f));
    const f32x2 e = p * (f32x2){__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
    const f32x2 hv = 0.5f * v;
    return __builtin_elementwise_fma(hv, e, hv);
}


// Packed ("PK") operand layout used by gemm_pk_kernel: a matrix X[R][K] as MFMA fragments
//   [row tile R/32][k-step K/16][plane][lane = (row % 32) + 32 * ((k % 16) / 8)][8 halves = k % 8]
// i.e. every (tile, k-step, plane) is the 1 KiB register image of one 32x16 operand: one direct global->LDS
// instruction moves it, one conflict-free ds_read_b128 per lane loads it.
__device__ __forceinline__ size_t pk_off(int64_t m, int k, int pl, int KS, int NP)
{
    return ((((size_t)(m >> 5) * KS + (k >> 4)) * NP + pl) << 9) + ((size_t)((m & 31) + ((k & 8) << 2)) << 3) + (k & 7);
}

// Epilogues shared by gemm16_kernel and gemm_pk_kernel.  acc[i][j] is the 32x32 tile (row tile wm*TM+i, column
// tile wn*TN+j) of the block; swapped orientation (EPI != EPI_VT): lane = token, registers = output columns.
// LNF (gemm_pk_kernel, folded LayerNorm: GemmArgs): cs_addr = LDS address of this block's 256 column sums, followed by the (mean, rstd)
// pairs of its 256 rows.
template <class T, int NP, int EPI, int TM, int TN, int WN, bool LNF = false>
__device__ __forceinline__ void gemm16_epilogue(const GemmArgs &p, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn,
                                                int r, int h, unsigned lut_addr = 0u, unsigned cs_addr = 0u)
{
    const float os = p.out_scale;
    if (EPI == EPI_VT) {
        // natural: lane = output column n (-> head, d), registers = tokens; 4 consecutive tokens per register quad
        // v^T planes [rows][n_head][hs][256]
        const int64_t b = m0 >> 8;
        const int tb = (int)(m0 & (kT - 1));
        if constexpr (LNF) {                                                   // folded LayerNorm (GemmArgs): registers = tokens here
            float cs[TN];
#pragma unroll
            for (int j = 0; j < TN; j++) asm volatile("ds_read_b32 %0, %1" : "=v"(cs[j]) : "v"(cs_addr + (unsigned)((wn * TN + j) * 32 + r) * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 s01, s23;                                            // (mean, rstd) of 4 consecutive tokens, from the block's LDS copy
                    const unsigned sa = cs_addr + 1024u + (unsigned)((wm * TM + i) * 32 + 8 * gq + 4 * h) * 8u;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(s01) : "v"(sa) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(s23) : "v"(sa) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s01), "+v"(s23) : : "memory");
                    const float mean[4] = {s01[0], s01[2], s23[0], s23[2]}, rstd[4] = {s01[1], s01[3], s23[1], s23[3]};
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[i][j][4 * gq + e] = rstd[e] * fmaf(-mean[e], cs[j], acc[i][j][4 * gq + e]);
                }
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + (wn * TN + j) * 32 + r;                         // column inside V (n_base handled by the W pointer)
            const int head = n / p.hs, d = n - head * p.hs;
            const int64_t rowbase = ((b * p.n_head + head) * p.hs + d) * kT;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int t = tb + (wm * TM + i) * 32 + 8 * gq + 4 * h;
                    const float v[4] = {acc[i][j][4 * gq] * os, acc[i][j][4 * gq + 1] * os, acc[i][j][4 * gq + 2] * os,
                                        acc[i][j][4 * gq + 3] * os};
                    u32x2 hi, lo;
                    split4<T, NP>(v, hi, lo);
                    // chunk-major v^T planes [rows][n_head][256/8][hs][8 tokens]: the 32 lanes (d) of a store are 512 contiguous bytes
                    const int64_t off = p.chunk_major ? (((b * p.n_head + head) * (kT / 8) + (t >> 3)) * p.hs + d) * 8 + (t & 7) : rowbase + t;
                    *reinterpret_cast<u32x2 *>(p.o_hi + off) = hi;
                    if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_lo + off) = lo;
                }
        }
    } else {
        // swapped: lane = token m, registers = 4 consecutive output columns per quad
        float rsum[TM], rsq[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) { rsum[i] = 0.f; rsq[i] = 0.f; }
        f32x2 st[TM];                                                          // LNF: (mean, rstd) of this lane's tokens
#pragma unroll
        for (int i = 0; i < TM; i++) st[i] = (f32x2){0.f, 1.f};
        if constexpr (LNF && EPI != EPI_RESID) {                               // folded LayerNorm (GemmArgs): lane = token
            // the block's column sums sit in LDS (one 1-KiB piece fetched at kernel start): a column tile's four quads per round trip;
            // acc <- acc - mean * colsum here, the factor rstd rides on the output scale below
#pragma unroll
            for (int i = 0; i < TM; i++)
                asm volatile("ds_read_b64 %0, %1" : "=v"(st[i]) : "v"(cs_addr + 1024u + (unsigned)((wm * TM + i) * 32 + r) * 8u) : "memory");
#pragma unroll
            for (int j = 0; j < TN; j++) {
                f32x4 cs[4];
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(cs[gq]) : "v"(cs_addr + (unsigned)((wn * TN + j) * 32 + 8 * gq + 4 * h) * 4u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]) : : "memory");
                if (j == 0) asm volatile("" : "+v"(st[0]), "+v"(st[TM - 1]));
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int g = 0; g < 16; g++) acc[i][j][g] = fmaf(-st[i][0], cs[g >> 2][g & 3], acc[i][j][g]);
                // (volatile: the tile's arithmetic stays in front of the next tile's reads -- hipcc otherwise reads all 16 quads first and spills)
                static_assert(TM == 2, "the pin below names both row tiles");
                asm volatile("" : "+v"(acc[0][j]), "+v"(acc[1][j]));
            }
        }
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int64_t m = m0 + (wm * TM + i) * 32 + r;
            const float osi = (LNF && EPI != EPI_RESID) ? os * st[i][1] : os;  // output scale of this lane's row
            if (EPI == EPI_RESID) {
                const float sh = (p.raw_out != nullptr && p.shift != nullptr) ? p.shift[m * p.shift_stride + p.shift_offset] : 0.f;
                // read-modify-write of the residual rows: the reads of two column tiles (8 x 16 B per lane) are all in flight
                // before the first store -- one memory round trip per pair of tiles instead of one per 16-byte piece
                // (the stores to x_out would otherwise keep the compiler from moving the next read up)
                constexpr int JB = TN >= 2 ? 2 : 1;
#pragma unroll
                for (int jb = 0; jb < TN; jb += JB) {
                    f32x4 cur[JB][4];
#pragma unroll
                    for (int jj = 0; jj < JB; jj++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++)
                            cur[jj][gq] = *reinterpret_cast<const f32x4 *>(p.x_out + (p.x_tiled ? xt_off(m, n0 + (wn * TN + jb + jj) * 32 + 8 * gq + 4 * h, p.N)
                                                                                                    : m * p.N + n0 + (wn * TN + jb + jj) * 32 + 8 * gq + 4 * h));
#pragma unroll
                    for (int jj = 0; jj < JB; jj++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) {
                            const int j = jb + jj;
                            f32x4 c = cur[jj][gq];
#pragma unroll
                            for (int e = 0; e < 4; e++) { c[e] += acc[i][j][4 * gq + e] * os; acc[i][j][4 * gq + e] = c[e]; }   // keep the new row for the stats
                            *reinterpret_cast<f32x4 *>(p.x_out + (p.x_tiled ? xt_off(m, n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h, p.N)
                                                                             : m * p.N + n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h)) = c;
                            if (p.raw_out == nullptr) rsum[i] += (c[0] + c[1]) + (c[2] + c[3]);
                            if (p.raw_out != nullptr) {                        // the next GEMM's A operand: the raw row in operand planes
                                const int n = n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h;
                                const float v[4] = {c[0] - sh, c[1] - sh, c[2] - sh, c[3] - sh};
                                // (sum, sum of squares) of the SHIFTED row: the shift is the row's mean at the LayerNorm before, so
                                // E[v^2] - E[v]^2 in ln_finalize_kernel is a variance about a nearly centred value -- on the raw row its
                                // relative error grew with (mean / std)^2
                                rsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                                rsq[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                                u32x2 hi, lo;
                                split4<T, NP>(v, hi, lo);
                                *reinterpret_cast<u32x2 *>(p.raw_out + pk_off(m, n, 0, p.N >> 4, NP)) = hi;
                                if (NP == 2) *reinterpret_cast<u32x2 *>(p.raw_out + pk_off(m, n, 1, p.N >> 4, NP)) = lo;
                            }
                        }
                }
                if (p.rsum_out != nullptr) {                                   // this wave's 128 columns of the row: one partial, fixed place
                    const float s1 = rsum[i] + __shfl_xor(rsum[i], 32), s2 = rsq[i] + __shfl_xor(rsq[i], 32);
                    if (h == 0) p.rsum_out[(size_t)((n0 + wn * TN * 32) >> 7) * (size_t)p.M + m] = make_float2(s1, s2);
                }
                continue;
            }
            if (EPI == EPI_GELU && lut_addr != 0u) {
                // Phi table in LDS (gemm_pk_kernel): 8 VALU + one gather per value; the 16 gathers of a column tile are in
                // flight together (one LDS round trip per tile instead of one per 4 values)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    float v[16], fr[16];
                    f32x2 tb[16];
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        v[g] = acc[i][j][g] * osi;
                        const float tt = __builtin_amdgcn_fmed3f(fmaf(v[g], kGeluLutScale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                        fr[g] = __builtin_amdgcn_fractf(tt);
                        asm volatile("ds_read_b64 %0, %1" : "=v"(tb[g]) : "v"(lut_addr + (unsigned)tt * 8u) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        asm volatile("" : "+v"(tb[g]));
                        v[g] *= fmaf(fr[g], tb[g][1], tb[g][0]);
                    }
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int n = n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h;
                        u32x2 hi, lo;
                        split4<T, NP>(v + 4 * gq, hi, lo);
                        if (p.o_pk) {                       // hidden planes feed gemm_pk_kernel next
                            *reinterpret_cast<u32x2 *>(p.o_hi + pk_off(m, n, 0, p.N >> 4, NP)) = hi;
                            if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_hi + pk_off(m, n, 1, p.N >> 4, NP)) = lo;
                        } else {
                            *reinterpret_cast<u32x2 *>(p.o_hi + m * p.N + n) = hi;
                            if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_lo + m * p.N + n) = lo;
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int n = n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h;   // first of 4 consecutive columns
                    float v[4] = {acc[i][j][4 * gq] * osi, acc[i][j][4 * gq + 1] * osi, acc[i][j][4 * gq + 2] * osi,
                                  acc[i][j][4 * gq + 3] * osi};
                    if (EPI == EPI_GELU) {
                        if (TM * TN >= 16) {          // one-wave-per-SIMD kernel: packed math
                            const f32x2 g0 = gelu_folded2((f32x2){v[0], v[1]}), g1 = gelu_folded2((f32x2){v[2], v[3]});
                            v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = gelu_folded(v[e]);
                        }
                        u32x2 hi, lo;
                        split4<T, NP>(v, hi, lo);
                        if (p.o_pk) {                       // hidden planes feed gemm_pk_kernel next
                            *reinterpret_cast<u32x2 *>(p.o_hi + pk_off(m, n, 0, p.N >> 4, NP)) = hi;
                            if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_hi + pk_off(m, n, 1, p.N >> 4, NP)) = lo;
                        } else {
                            *reinterpret_cast<u32x2 *>(p.o_hi + m * p.N + n) = hi;
                            if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_lo + m * p.N + n) = lo;
                        }
                    } else {   // EPI_QK: q|k planes [which][rows][n_head][256][hs]
                        const int which = n / p.C, cc = n - which * p.C;
                        const int head = cc / p.hs, d = cc - head * p.hs;
                        const int64_t bb = m >> 8;
                        const int t = (int)(m & (kT - 1));
                        // chunk-major q|k planes [which][rows][n_head][hs/8][256][8 d]: the 32 lanes (tokens) x 2 halves of a store
                        // are 512 contiguous bytes instead of 32 pieces of 16 B
                        const int64_t off = (int64_t)which * p.plane +
                                            (p.chunk_major ? (((bb * p.n_head + head) * (p.hs >> 3) + (d >> 3)) * kT + t) * 8 + (d & 7)
                                                           : ((bb * p.n_head + head) * kT + t) * p.hs + d);
                        u32x2 hi, lo;
                        split4<T, NP>(v, hi, lo);
                        *reinterpret_cast<u32x2 *>(p.o_hi + off) = hi;
                        if (NP == 2) *reinterpret_cast<u32x2 *>(p.o_lo + off) = lo;
                    }
                }
        }
        if (EPI == EPI_RESID && WN == 1 && p.stats_out != nullptr) {
            // this wave holds complete rows (BN == N): LayerNorm statistics of the NEW residual rows for the next kernel
#pragma unroll
            for (int i = 0; i < TM; i++) {
                float s = rsum[i] + __shfl_xor(rsum[i], 32);
                const float mean = s / (float)p.N;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int g = 0; g < 16; g++) { const float d = acc[i][j][g] - mean; q += d * d; }
                q += __shfl_xor(q, 32);
                if (h == 0) p.stats_out[m0 + (wm * TM + i) * 32 + r] = make_float2(mean, rsqrtf(q / (float)p.N + 1e-5f));
            }
        }
    }
}

template <class T, int NP, int BN, int WM, int WN, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmArgs p)
{
    constexpr int BM = 128, BK = 32;
    constexpr int RS = (BK + 8) * 2;                       // LDS row stride in bytes (80): conflict-free ds_read_b128
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr bool SWAP = (EPI != EPI_VT);
    static_assert(WM * WN == 4 && TM * WM * 32 == BM && TN * WN * 32 == BN, "tile config");
    static_assert(EPI != EPI_RESID || true, "");
    __shared__ __attribute__((aligned(16))) unsigned char sA[NP][BM * RS];
    __shared__ __attribute__((aligned(16))) unsigned char sB[NP][BN * RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int mt = blockIdx.x / p.n_tiles_n, nt = blockIdx.x - mt * p.n_tiles_n;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;                                // row offset into the W planes of this launch
    const int K = p.K;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;

    // ---- staging registers ----
    constexpr int A_LN_V = 4;                              // PRO_LN: 4 float4 per thread (128 rows x 8 float4)
    constexpr int A_PL_TOT = NP * BM * 4;                  // PRO_PLANES: 16-byte chunks per tile
    constexpr int A_PL_V = (A_PL_TOT + 255) / 256;
    constexpr int B_TOT = NP * BN * 4;
    constexpr int B_V = (B_TOT + 255) / 256;
    f32x4 ra_f[PRO == PRO_LN ? A_LN_V : 1];
    u32x4 ra_p[PRO == PRO_PLANES ? A_PL_V : 1];
    u32x4 rb[B_V];
    float my_mean[4], my_rstd[4];
    if (PRO == PRO_LN) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float2 st = p.stats[m0 + (tid >> 3) + 32 * i];
            my_mean[i] = st.x; my_rstd[i] = st.y;
        }
    }

#define MGPT_G16_LOAD(kt_)                                                                                          \
    {                                                                                                               \
        const int k0_ = (kt_) * BK;                                                                                 \
        if (PRO == PRO_LN) {                                                                                        \
            _Pragma("unroll") for (int i = 0; i < A_LN_V; i++)                                                      \
                ra_f[i] = *reinterpret_cast<const f32x4 *>(p.x + (m0 + (tid >> 3) + 32 * i) * K + k0_ + (tid & 7) * 4); \
        } else {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < A_PL_V; i++) {                                                    \
                const int idx = tid + 256 * i;                                                                      \
                if (A_PL_TOT % 256 == 0 || idx < A_PL_TOT) {                                                        \
                    const int pl = idx / (BM * 4), rem = idx - pl * (BM * 4), row = rem >> 2, c = rem & 3;          \
                    const uint16_t *src = (pl == 0 ? p.a_hi : p.a_lo) + (m0 + row) * K + k0_ + c * 8;               \
                    ra_p[i] = *reinterpret_cast<const u32x4 *>(src);                                                \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < B_V; i++) {                                                           \
            const int idx = tid + 256 * i;                                                                          \
            if (B_TOT % 256 == 0 || idx < B_TOT) {                                                                  \
                const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4), row = rem >> 2, c = rem & 3;              \
                const uint16_t *src = (pl == 0 ? p.w_hi : p.w_lo) + (size_t)(n0 + row) * K + k0_ + c * 8;           \
                rb[i] = *reinterpret_cast<const u32x4 *>(src);                                                      \
            }                                                                                                       \
        }                                                                                                           \
    }

    const int nk = K / BK;
    MGPT_G16_LOAD(0);
    for (int kt = 0; kt < nk; kt++) {
        __syncthreads();                                   // previous tile fully consumed
        if (PRO == PRO_LN) {
            const f32x4 gn = *reinterpret_cast<const f32x4 *>(p.gain + kt * BK + (tid & 7) * 4);
#pragma unroll
            for (int i = 0; i < A_LN_V; i++) {
                const int row = (tid >> 3) + 32 * i;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (ra_f[i][e] - my_mean[i]) * my_rstd[i] * gn[e];
                u32x2 hi, lo;
                split4<T, NP>(v, hi, lo);
                *reinterpret_cast<u32x2 *>(&sA[0][row * RS + (tid & 7) * 8]) = hi;
                if (NP == 2) *reinterpret_cast<u32x2 *>(&sA[NP - 1][row * RS + (tid & 7) * 8]) = lo;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PL_V; i++) {
                const int idx = tid + 256 * i;
                if (A_PL_TOT % 256 == 0 || idx < A_PL_TOT) {
                    const int pl = idx / (BM * 4), rem = idx - pl * (BM * 4), row = rem >> 2, c = rem & 3;
                    *reinterpret_cast<u32x4 *>(&sA[pl][row * RS + c * 16]) = ra_p[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_V; i++) {
            const int idx = tid + 256 * i;
            if (B_TOT % 256 == 0 || idx < B_TOT) {
                const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4), row = rem >> 2, c = rem & 3;
                *reinterpret_cast<u32x4 *>(&sB[pl][row * RS + c * 16]) = rb[i];
            }
        }
        __syncthreads();
        if (kt + 1 < nk) MGPT_G16_LOAD(kt + 1);            // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            u32x4 a[TM][2], b[TN][2];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    a[i][pl] = *reinterpret_cast<const u32x4 *>(&sA[pl][((wm * TM + i) * 32 + r) * RS + ks * 32 + h * 16]);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    b[j][pl] = *reinterpret_cast<const u32x4 *>(&sB[pl][((wn * TN + j) * 32 + r) * RS + ks * 32 + h * 16]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = SWAP ? mma<T, NP>(b[j], a[i], acc[i][j]) : mma<T, NP>(a[i], b[j], acc[i][j]);
        }
    }
#undef MGPT_G16_LOAD

    gemm16_epilogue<T, NP, EPI, TM, TN, WN>(p, acc, m0, n0, wm, wn, r, h);
}

// ---------------------------------------------------------------------------------------------
// PK GEMM (C = 256, 768: 6M and 85M shapes): both operands pre-packed as MFMA fragments (pk_off above), so
//   * one k-step of a 256 x 256 block tile = 8 + 8 fragments (x NP planes) that go HBM/L2 -> LDS by direct
//     global->LDS loads, 1 KiB contiguous each, through a ring of NST stages with counted vmcnt + raw s_barrier;
//   * a wave owns a 128 x 128 sub-tile (4 x 4 MFMA tiles, 256 accumulator registers): every fragment it reads from
//     LDS feeds 4 MFMAs (x3 in the split-fp16 mode), i.e. 64 B/clk of LDS reads per CU at full matrix rate -- the
//     128 x BN x 32 kernel above (2 x 2 tiles, two __syncthreads per 24 MFMAs) ran at 20 % of the MFMA peak;
//   * one wave per SIMD (512 registers), so the schedule is explicit: the next k-step's fragments are requested
//     before this k-step's MFMAs, and the 16 accumulators are visited round-robin (3 rounds in the split mode).
// Weights are packed once at finalize (pack_pk_kernel); activations are produced in PK layout by ln_pack_kernel
// (LayerNorm statistics + normalise + split in one pass over x), attn16_kernel (y) and the GELU epilogue (hidden planes).
// ---------------------------------------------------------------------------------------------
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_pk_kernel(const float *__restrict__ w, uint16_t *__restrict__ out, int R, int K,
                                                      float scale, const float *__restrict__ gain = nullptr)
{
    const int KS = K >> 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (row tile, k-step, lane)
    if (gid >= (int64_t)(R >> 5) * KS * 64) return;
    const int lane = (int)(gid & 63), ks = (int)((gid >> 6) % KS), rt = (int)((gid >> 6) / KS);
    const float *src = w + (size_t)(rt * 32 + (lane & 31)) * K + ks * 16 + (lane >> 5) * 8;
    float v0[4], v1[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { v0[e] = src[e] * scale; v1[e] = src[4 + e] * scale; }
    if (gain != nullptr) {                                                // W * ln.weight (folded LayerNorm, GemmArgs)
        const float *gk = gain + ks * 16 + (lane >> 5) * 8;
#pragma unroll
        for (int e = 0; e < 4; e++) { v0[e] = src[e] * gk[e] * scale; v1[e] = src[4 + e] * gk[4 + e] * scale; }
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v0, h0, l0);
    split4<T, NP>(v1, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)rt * KS + ks) * NP << 9) + lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// colsum[n] = sum_k of the values pack_pk_kernel stored for row n (W[n][k] * gain[k] * scale rounded to the operand type, hi + lo):
// what sum_k x[m][k] W'[n][k] yields for a row of ones -- the mean term of the folded LayerNorm (GemmArgs).  One thread per row,
// k in index order.
template <class T, int NP>
__global__ __launch_bounds__(256) void colsum_pk_kernel(const float *__restrict__ w, const float *__restrict__ gain, float *__restrict__ colsum,
                                                        int R, int K, float scale)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= R) return;
    const float *src = w + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; k++) {
        const float v = src[k] * gain[k] * scale;
        const uint16_t hi = T::cvt(v);
        float r = T::back(hi);
        if (NP == 2) r += T::back(T::cvt(v - r));
        s += r;
    }
    colsum[n] = s;
}

// (mean, rstd) of every row from the per-128-column partial sums an EPI_RESID epilogue left (GemmArgs::rsum_out), added in block order
// mean_buf (GemmArgs::shift): on entry the shift the planes of these rows were written with (row m: mean_buf[m * stride + offset]), on exit
// -- keep != 0 -- the rows' new mean, i.e. the shift of the next residual epilogue.  stats.x = mean - shift: what the consumer subtracts.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2 *__restrict__ parts, int n_parts, int64_t M, int C, float2 *__restrict__ stats,
                                                          float *__restrict__ mean_buf, int stride, int offset, int keep)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < n_parts; q++) { const float2 v = parts[(size_t)q * M + m]; s1 += v.x; s2 += v.y; }
    // the partial sums are those of x - shift (shift = the row's mean at the LayerNorm before): mean and variance about the shift
    const float dm = s1 / (float)C;                        // mean - shift
    const float var = fmaxf(s2 / (float)C - dm * dm, 0.f);
    const float sh = mean_buf[m * stride + offset];
    stats[m] = make_float2(dm, rsqrtf(var + 1e-5f));
    if (keep) mean_buf[m * stride + offset] = sh + dm;
}

// LayerNorm + split into PK operand planes, statistics computed here: a workgroup owns one 32-token tile, every lane
// keeps its share of the row (C/8 floats) in registers between the mean, the centred variance (two-pass, as
// model.py:19-20 / F.layer_norm) and the normalisation, so x is read exactly once; wave w writes the whole 1 KiB
// fragments of k-steps w, w+4, ...
template <class T, int NP, int KSW>                        // KSW = k-steps per wave = C / 64
// raw_stats != NULL (folded LayerNorm, GemmArgs: used once per forward, for the embedding rows): the planes carry x - mean (no rstd, no
// gain), raw_stats gets (0, rstd) and raw_mean the mean (= the shift of these planes, GemmArgs::shift).
__global__ __launch_bounds__(256) void ln_pack_kernel(const float *__restrict__ x, const float *__restrict__ gain,
                                                      uint16_t *__restrict__ out, int C, int tiled, float2 *__restrict__ raw_stats = nullptr,
                                                      float *__restrict__ raw_mean = nullptr)
{
    __shared__ float red[2][8][32];
    const int KS = C >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t rt = blockIdx.x, m = rt * 32 + r;
    // tiled: x is chunk-major (xt_off); lane (r, h) of k-step ks owns chunk 2 ks + h of token r
    const float *xr = tiled ? x + rt * 32 * C + h * 256 + r * 8 : x + m * C + h * 8;
    const int kstride = tiled ? 512 : 16;
    f32x4 va[KSW], vb[KSW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KSW; i++) {
        const int ks = wave + 4 * i;
        va[i] = *reinterpret_cast<const f32x4 *>(xr + ks * kstride);
        vb[i] = *reinterpret_cast<const f32x4 *>(xr + ks * kstride + 4);
        s += ((va[i][0] + va[i][1]) + (va[i][2] + va[i][3])) + ((vb[i][0] + vb[i][1]) + (vb[i][2] + vb[i][3]));
    }
    red[0][wave * 2 + h][r] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) tot += red[0][k][r];
    const float mean = tot / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < KSW; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float da = va[i][e] - mean, db = vb[i][e] - mean;
            q += da * da + db * db;
        }
    red[1][wave * 2 + h][r] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) qt += red[1][k][r];
    const float rstd = rsqrtf(qt / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < KSW; i++) {
        const int ks = wave + 4 * i;
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gain + ks * 16 + h * 8), gb = *reinterpret_cast<const f32x4 *>(gain + ks * 16 + h * 8 + 4);
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { v0[e] = (va[i][e] - mean) * rstd * ga[e]; v1[e] = (vb[i][e] - mean) * rstd * gb[e]; }
        if (raw_stats != nullptr) {                        // x - mean: shifted by the row's own mean (GemmArgs::shift), no rstd, no gain
#pragma unroll
            for (int e = 0; e < 4; e++) { v0[e] = va[i][e] - mean; v1[e] = vb[i][e] - mean; }
        }
        u32x2 h0, l0, h1, l1;
        split4<T, NP>(v0, h0, l0);
        split4<T, NP>(v1, h1, l1);
        u32x4 hi, lo;
        hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
        lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
        uint16_t *dst = out + (((size_t)rt * KS + ks) * NP << 9) + lane * 8;
        *reinterpret_cast<u32x4 *>(dst) = hi;
        if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
    }
    if (raw_stats != nullptr && wave == 0 && h == 0) { raw_stats[m] = make_float2(0.f, rstd); raw_mean[m] = mean; }
}

// 8 waves of 64 x 128 (2 x 4 MFMA tiles, 256 registers, two waves per SIMD: the second wave on the SIMD covers part of the
// issue time of the ds_reads, the refill loads and the epilogue's VALU work -- tools/bench_probes/probe_mfma_lds*.hip).
// One ring stage = KPS k-steps of the 256 x 256 block tile (16 * KPS * NP fragments of 1 KiB, fetched by direct
// global->LDS loads NST - 1 stages ahead), one s_barrier per stage.  KPS = 2 in the one-plane (bf16) mode (16 MFMAs per
// wave between barriers) was measured and changes nothing (860 vs 888 cycles per k-step, tools/bench_probes/probe_gemm_pk.hip);
// the library runs KPS = 1.
// DBG (probe only): 1 = wave 0 leaves stamps[workgroup][8] = {cycles waiting at the tops of its tiles, in their k loops, in their epilogues, tiles, total cycles, total 100-MHz ticks}.
// NWV = 8: one 256 x 256 block per CU.  NWV = 4: 128 x 256 blocks, two per CU, each with its own ring and barrier (small launches: more
// tiles than the 256-row form; at full size it is no faster, profiles/r03_probe_gemm_pk.txt).
#ifdef MGPT_AB_GEMM_CLUMPED
constexpr bool kGemmPkPlace = false;
#else
constexpr bool kGemmPkPlace = true;
#endif
#ifdef MGPT_ABL_GEMM_16X16
constexpr int kGemmPkMfmaPerTile = 2;       // (timing experiment: two 16 x 16 x 32 MFMAs in the place of every 32 x 32 x 16 one, see `round`)
#else
constexpr int kGemmPkMfmaPerTile = 1;
#endif
#ifdef MGPT_AB_GEMM_DMA_TOP
constexpr bool kGemmPkDmaPlace = false;
#else
constexpr bool kGemmPkDmaPlace = true;
#endif
constexpr int gemm_pk_kps(int NP) { return 1; }
// instances compiled WITH the tile loop (see gemm_pk_kernel): the one-plane GELU epilogue, where it measured faster; the split-mode and the
// natural-orientation (v^T) instances would spill with the loop's state carried through their epilogues, and the residual epilogue at K = 3072
// got 4.5 % SLOWER compiled with the loop and launched one workgroup per tile (profiles/r05_ab.txt, visit H; 1.3-1.6 % of that is the faster
// c_fc in front of it taking from the shared power budget, visit I)
constexpr bool gemm_pk_persistent(int NP, int EPI, bool LNF) { return NP == 1 && EPI == EPI_GELU; }
constexpr int gemm_pk_nst(int NP, int NWV = 8, int EPI = 0) { return NWV == 8 ? (NP == 2 ? 4 : 6) : (NP == 2 ? 3 : (EPI == EPI_GELU ? 4 : 6)); }
constexpr int gemm_pk_lds(int NP, int NWV = 8, int EPI = 0) { return gemm_pk_nst(NP, NWV, EPI) * (NWV + 8) * gemm_pk_kps(NP) * NP * 1024; }   // + the Phi table when used

// LNF (folded LayerNorm, GemmArgs): the block's 256 column sums and the (mean, rstd) of its 256 rows are three more 1-KiB pieces in LDS,
// behind the ring and the Phi table.
// PERSISTENT form (round 5, gemm_pk_persistent instances): a grid SMALLER than the number of tiles makes workgroup v walk tiles v, v + gridDim.x,
// ...; the first NST - 1 stages of the NEXT tile are requested before the epilogue of this one (the ring is idle by then) and land under it.
// A/B in one box on the 85M chain (profiles/r05_ab.txt, visits G and H): c_fc (GELU epilogue) -3 to -4 %; attention out-projection -2.6 %, q|k
// unchanged, c_proj (K = 3072) +4 % -- only the GELU instance keeps the loop.
template <class T, int NP, int EPI, int NWV, int DBG = 0, bool LNF = false>
__global__ __launch_bounds__(NWV * 64, 2) void gemm_pk_kernel(GemmArgs p, unsigned long long *stamps = nullptr)
{
    static_assert(NWV == 8 || NWV == 4, "8 or 4 waves of 64 x 128");
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};         // DBG: {entry cycles, entry ticks, cycles waiting for stage 0, main loops, epilogues, tiles}
    unsigned long long t_mark = 0;
    if constexpr (DBG != 0) { ts[0] = __builtin_readcyclecounter(); ts[1] = wall_clock64(); }
    constexpr int TM = 2, TN = 4;                          // MFMA tiles per wave; waves are (NWV / 2) x 2
    constexpr int AF = NWV;                                // A fragments per k-step = block rows / 32
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int KPS = gemm_pk_kps(NP);                   // k-steps per ring stage
    constexpr int NST = gemm_pk_nst(NP, NWV, EPI);         // ring depth, in stages
    constexpr int STAGE = (AF + 8) * KPS * NP * 1024;      // AF A fragments + 8 B fragments, KPS k-steps, NP planes each
    constexpr int PER_WAVE = (AF + 8) * KPS * NP / NWV;    // direct-to-LDS loads a wave issues per stage
    static_assert((AF + 8) * KPS * NP % NWV == 0 && NST >= 3, "ring shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NST][STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // Tile map, XCD-aware: hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own 4 MiB L2) and tile
    // v + k * gridDim.x stays on workgroup v's XCD (the grid is a multiple of 8), so logical ids are made contiguous per XCD, and inside an
    // XCD the ~32 tiles in flight form bands of 4 token tiles x all column tiles (token tile fastest): every A tile is fetched once
    // into that L2 and hit by the other column tiles, every weight tile by the 4 token tiles.
    const int ntn = p.n_tiles_n, mtn = p.M / (AF * 32), nb = mtn * ntn;
    const int KS = p.K >> 4, NSTG = KS / KPS;
    int mt = 0, nt = 0;
    const unsigned char *abase = nullptr, *bbase = nullptr;
    auto set_tile = [&](int v) {
        int id = v;
        if ((nb & 7) == 0 && (gridDim.x & 7) == 0) id = (id & 7) * (nb >> 3) + (id >> 3);
        constexpr int GM = 32 / AF;                        // 1024 token rows per band
        const int band = id / (GM * ntn);
        const int gm = min(GM, mtn - band * GM);
        const int rem = id - band * GM * ntn;
        nt = rem / gm; mt = band * GM + (rem - nt * gm);
        abase = reinterpret_cast<const unsigned char *>(p.a_hi) + (size_t)mt * AF * KS * NP * 1024 + lane * 16;
        bbase = reinterpret_cast<const unsigned char *>(p.w_hi) + (size_t)nt * 8 * KS * NP * 1024 + lane * 16;
    };

    // stage S -> LDS [fragment f][k-step kk of the stage][plane]: the KPS * NP pieces of a fragment are contiguous on both sides
    auto issue = [&](int S) {
        unsigned char *dst = smem + (size_t)(S % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            const int c = wave + NWV * i;                  // piece c = (f * KPS + kk) * NP + pl
            const int f = c / (KPS * NP), q = c - f * (KPS * NP);
            const unsigned char *src = (f < AF) ? abase + ((size_t)(f * KS + S * KPS) * NP + q) * 1024
                                                : bbase + ((size_t)((f - AF) * KS + S * KPS) * NP + q) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
    };
    unsigned lut_addr = 0u;
    if (EPI == EPI_GELU && p.gelu_lut != nullptr) {        // (uniform) Phi table behind the ring, older than every ring piece
        unsigned char *dst = smem + (size_t)NST * STAGE;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(p.gelu_lut);
#pragma unroll
        for (int i = 0; i < kGeluLutN * 8 / 1024 / NWV; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(wave + NWV * i) * 1024 + lane * 16),
                                             (lds_void_t *)(dst + (size_t)(wave + NWV * i) * 1024), 16, 0, 0);
        lut_addr = (unsigned)(size_t)dst;
    }
    unsigned cs_addr = 0u;
    if constexpr (LNF) cs_addr = (unsigned)(size_t)(smem + (size_t)NST * STAGE + (EPI == EPI_GELU ? kGeluLutN * 8 : 0));
    int vt = blockIdx.x;                                   // the launcher keeps gridDim.x <= number of tiles
    set_tile(vt);
#pragma unroll
    for (int S = 0; S < NST - 1; S++) issue(S);            // K >= 16 * KPS * NST is checked by the launcher

    f32x16 acc[TM][TN];
    u32x4 fa[2][TM][NP], fb[2][TN][NP];                    // [buffer][tile][plane]
    auto fetch = [&](int S, int kk, int buf) {             // fragments of k-step kk of stage S
        const unsigned char *st = smem + (size_t)(S % NST) * STAGE + (size_t)kk * NP * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < NP; pl++) {
#pragma unroll
            for (int i = 0; i < TM; i++) fa[buf][i][pl] = *reinterpret_cast<const u32x4 *>(st + (size_t)((wm * TM + i) * KPS * NP + pl) * 1024);
#pragma unroll
            for (int j = 0; j < TN; j++) fb[buf][j][pl] = *reinterpret_cast<const u32x4 *>(st + (size_t)((AF + wn * TN + j) * KPS * NP + pl) * 1024);
        }
    };
    auto round = [&](int buf, int pa, int pb) {            // one MFMA on each of the TM x TN accumulators
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
#ifdef MGPT_ABL_GEMM_16X16
                // timing experiment (tools/bench_probes/probe_gemm_pk.hip, results are WRONG): every 32 x 32 x 16 MFMA replaced by two 16 x 16 x 32 MFMAs on the
                // same operand registers (the same flops, the same fragment reads), accumulating into two quads of the tile's registers
                f32x4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                c0 = T::mfma16(fb[buf][j][pb], fa[buf][i][pa], c0);
                c1 = T::mfma16(fa[buf][i][pa], fb[buf][j][pb], c1);
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[i][j][e] = c0[e]; acc[i][j][4 + e] = c1[e]; }
#else
                acc[i][j] = SWAP ? T::mfma(fb[buf][j][pb], fa[buf][i][pa], acc[i][j]) : T::mfma(fa[buf][i][pa], fb[buf][j][pb], acc[i][j]);
#endif
            }
        if (!kGemmPkPlace) __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](int buf) {
        if (NP == 2) {
            // weight operand = B side when SWAP (fb), its lo plane first: lo.hi, hi.lo, hi.hi
            round(buf, SWAP ? 0 : 1, SWAP ? 1 : 0);
            round(buf, SWAP ? 1 : 0, SWAP ? 0 : 1);
        }
        round(buf, 0, 0);
    };
    // global k-step t = S * KPS + kk uses fragment buffer t & 1; its fragments were requested one k-step earlier
    auto stage = [&](int S, int buf0) {
        if (S + 1 < NSTG) {
            // stage S+1 landed for everyone; everyone is done reading the slot that gets refilled below
            if (S + NST - 2 < NSTG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 3) * PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (S + NST - 1 < NSTG) issue(S + NST - 1);
        }
#pragma unroll
        for (int kk = 0; kk < KPS; kk++) {
            const int buf = (buf0 + kk) & 1;
            if (kk + 1 < KPS) fetch(S, kk + 1, buf ^ 1);
            else if (kGemmPkPlace || S + 1 < NSTG) fetch(S + 1, 0, buf ^ 1);   // (placed form: unconditional, so that reads and MFMAs share a basic block; after the last
                                                                               //  stage the fragments read are stale ring contents that nothing uses)
            if (!kGemmPkPlace) __builtin_amdgcn_sched_barrier(0);
            mfmas(buf);
            if (kGemmPkPlace) {
                // round 5: the (TM + TN) NP fragment reads of the NEXT k-step go out one at a time BEHIND the MFMAs of this one --
                // clumped in front of them (rounds 1-4) their issue time added to the MFMAs' on both waves of the SIMD
#pragma unroll
                for (int n = 0; n < TM * TN * (NP == 2 ? 3 : 1); n++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (n < (TM + TN) * NP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // a stage of the main part of the loop (S + NST - 1 < NSTG): the refill of the slot freed by stage S - 1 is unconditional, so that its direct-to-LDS
    // loads share the basic block of the MFMAs and go out BEHIND the first of them -- issued between the barrier and the first MFMA (rounds 1-4)
    // they held the matrix pipe idle at the top of every stage, on both waves of the SIMD
    constexpr bool PERSIST = gemm_pk_persistent(NP, EPI, LNF);
    auto stage_main = [&](int S, int buf) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 3) * PER_WAVE) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        fetch(S + 1, 0, buf ^ 1);
        issue(S + NST - 1);
        mfmas(buf);
        // (the refill is a store to LDS as far as hipcc knows: the fragment reads, earlier in program order, stay in front of it -- so the reads go
        //  behind the first MFMAs and the refill pieces behind the ones that follow)
        constexpr int NMF = kGemmPkMfmaPerTile * TM * TN * (NP == 2 ? 3 : 1), NRD = (TM + TN) * NP, GAP = (NMF - NRD) / PER_WAVE > 0 ? (NMF - NRD) / PER_WAVE : 1;
#pragma unroll
        for (int n = 0; n < NMF; n++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifndef MGPT_AB_GEMM_SALU_FRONT
            // slot / address arithmetic behind the first MFMA instead of between the barrier and it (one-plane mode: 731 -> 690 cycles per k-step; in the
            // split mode the group size does not fit the second stage of the unrolled pair and the placement falls apart -- left to hipcc there)
            if (n == 0 && NP == 1) __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
#endif
            if (n < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if ((n - NRD) % GAP == 0 && (n - NRD) / GAP < PER_WAVE) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x010, PER_WAVE, 0);      // (pieces that found no MFMA to stand behind: NWV = 4)
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll 1
    for (;;) {
        if constexpr (DBG != 0) t_mark = __builtin_readcyclecounter();
        // ---- top of a tile: its stage 0 has landed for everyone (requested by the prologue above, or before the previous tile's epilogue: then the
        //      (NST - 2) PER_WAVE operations a wave may leave in flight are that epilogue's last stores and every piece is older.  Counting on more of
        //      the epilogue's operations -- measured with 24 -- changes nothing: what the top of a tile waits for is the barrier, i.e. the slower wave
        //      of each SIMD finishing its epilogue, 4 k cycles behind the faster one) ----
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (LNF) {                               // this tile's column sums and row statistics (read by its epilogue; the barrier above
            unsigned char *dst = smem + (size_t)NST * STAGE + (EPI == EPI_GELU ? kGeluLutN * 8 : 0);   // is behind the previous epilogue's reads)
            if (wave == 0)
                __builtin_amdgcn_global_load_lds((gbl_void_t *)(reinterpret_cast<const unsigned char *>(p.colsum + nt * 256) + lane * 16), (lds_void_t *)dst, 16, 0, 0);
            if (wave == 1 || (wave == 2 && AF == 8))       // (mean, rstd) of the block's AF x 32 rows: 2 KiB (1 KiB for the 128-row form)
                __builtin_amdgcn_global_load_lds((gbl_void_t *)(reinterpret_cast<const unsigned char *>(p.ln_stats + (size_t)mt * (AF * 32)) + (wave - 1) * 1024 + lane * 16),
                                                 (lds_void_t *)(dst + wave * 1024), 16, 0, 0);
            // (extra operations YOUNGER than a piece only make a counted wait for that piece stricter; the tail of the k loop waits for everything)
        }
        if constexpr (DBG != 0) { const unsigned long long t = __builtin_readcyclecounter(); ts[2] += t - t_mark; t_mark = t; }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;
        fetch(0, 0, 0);
        if (KPS == 2) {
#pragma unroll 1
            for (int S = 0; S < NSTG; S++) stage(S, 0);
        } else {
            int S = 0;
            if (kGemmPkPlace && kGemmPkDmaPlace) {
#pragma unroll 1
                for (; S + NST < NSTG; S += 2) {
                    stage_main(S, 0);
                    stage_main(S + 1, 1);
                }
            }
#pragma unroll 1
            for (; S < NSTG; S += 2) {
                stage(S, 0);
                stage(S + 1, 1);
            }
        }
        if constexpr (DBG != 0) { asm volatile("s_nop 0" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); ts[3] += t - t_mark; t_mark = t; }
        // ---- the next tile's first stages go out before this tile's epilogue: every wave has read its last fragments (barrier), nothing of the
        //      ring is in flight (the loop's tail waited for everything) ----
        const int64_t m0e = (int64_t)mt * (AF * 32);
        const int n0e = nt * 256;
        const int vnext = vt + (int)gridDim.x;
        const bool more = PERSIST && vnext < nb;           // uniform
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            set_tile(vnext);
#pragma unroll
            for (int S = 0; S < NST - 1; S++) issue(S);
        }
        gemm16_epilogue<T, NP, EPI, TM, TN, 2, LNF>(p, acc, m0e, n0e, wm, wn, r, h, lut_addr, cs_addr);
        if constexpr (DBG != 0) { const unsigned long long t = __builtin_readcyclecounter(); ts[4] += t - t_mark; ts[5] += 1; }
        if (!PERSIST || !more) break;
        // (the tile's coordinates and operand addresses are formed again from an opaque copy of the index: carried through the epilogue they
        //  cost it registers it does not have in the natural-orientation and split instances -- and scratch traffic counts in vmcnt)
        vt = vnext;
        asm volatile("" : "+s"(vt));
        set_tile(vt);
    }
    if constexpr (DBG != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
            o[0] = ts[2]; o[1] = ts[3]; o[2] = ts[4]; o[3] = ts[5];
            o[4] = __builtin_readcyclecounter() - ts[0]; o[5] = wall_clock64() - ts[1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_pk16_kernel (round 5): the ONE-PLANE packed GEMM on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16.  At the package power limit the small shape delivers
// 10 % (bf16) to 15 % (f16) more flops per second (profiles/r05_probe_mfma_shape.txt: the same operand bytes per flop, half the accumulator traffic), in this kernel's
// own loop -6...8 % (same file, "in situ").  Memory layouts do not change (PK fragments in, the epilogues' outputs); what changes is the slicing:
//   * a ring PAIR = two stages = K 32.  A 16 x 16 x 32 operand (16 rows x 32 k) comes out of the two stages' 32-row fragments with ONE ds_read_b128 per lane: lane
//     (rho = lane % 16, q = lane / 16) takes row rho (+ 16 for the odd 16-row group) and the 8 k of half (q & 1) from stage 2 p + (q >> 1) -- 16 lanes read 256
//     contiguous bytes; the two stages of a pair sit in consecutive ring slots (the ring depth is even), so the stage is a per-lane constant offset;
//   * wave tile 64 tokens x 128 columns = 4 x 8 groups of 16: 32 MFMAs and 12 operand reads per pair (as many reads as two k-steps had), one barrier per PAIR;
//   * D(ng, mg): a lane holds token 16 mg + rho and the columns 16 ng + 4 q .. + 3 (swapped orientation: q|k, GELU, residual) resp. the tokens
//     16 mg + 4 q .. + 3 and column 16 ng + rho (v^T) -- a quad of four consecutive columns (tokens) per register quad, as in the 32 x 32 layout, so the
//     epilogues store the same 8- / 16-byte pieces; a token now sits in FOUR lanes (rho + 16 q), the row sums of the residual epilogue fold over them.
// Weight fragments are single-buffered (a group's fragment is re-read for the next pair as soon as its four MFMAs are out), token fragments double-buffered.
// ---------------------------------------------------------------------------------------------
template <class T, int EPI, bool LNF>
__device__ __forceinline__ void gemm16x16_epilogue(const GemmArgs &p, f32x4 (&c)[8][4], int64_t m0w, int n0w, int mloc0, int nloc0, int rho, int q,
                                                   unsigned lut_addr, unsigned cs_addr)
{
    // m0w / n0w: first token / column of this wave's 64 x 128 tile; mloc0 / nloc0: the same inside the block (rows of the LNF statistics / column sums in LDS)
    constexpr int NP = 1;
    const float os = p.out_scale;
    if constexpr (EPI == EPI_VT) {
        // natural orientation: lane = column n (-> head, d), registers = 4 consecutive tokens; v^T planes [rows][n_head][hs][256]
        const int64_t b = m0w >> 8;
        const int tb = (int)(m0w & (kT - 1));
#pragma unroll
        for (int ng = 0; ng < 8; ng++) {
            const int n = n0w + 16 * ng + rho;
            const int head = n / p.hs, d = n - head * p.hs;
            const int64_t rowbase = ((b * p.n_head + head) * p.hs + d) * kT;
            float cs = 0.f;
            if constexpr (LNF) asm volatile("ds_read_b32 %0, %1" : "=v"(cs) : "v"(cs_addr + (unsigned)(nloc0 + 16 * ng + rho) * 4u) : "memory");
#pragma unroll
            for (int mg = 0; mg < 4; mg++) {
                float v[4] = {c[ng][mg][0], c[ng][mg][1], c[ng][mg][2], c[ng][mg][3]};
                if constexpr (LNF) {                                       // (mean, rstd) of the 4 consecutive tokens, from the block's LDS copy
                    f32x4 s01, s23;
                    const unsigned sa = cs_addr + 1024u + (unsigned)(mloc0 + 16 * mg + 4 * q) * 8u;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(s01) : "v"(sa) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(s23) : "v"(sa) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s01), "+v"(s23), "+v"(cs) : : "memory");
                    const float mean[4] = {s01[0], s01[2], s23[0], s23[2]}, rstd[4] = {s01[1], s01[3], s23[1], s23[3]};
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = rstd[e] * fmaf(-mean[e], cs, v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] *= os;
                const int t = tb + 16 * mg + 4 * q;
                u32x2 hi, lo;
                split4<T, NP>(v, hi, lo);
                const int64_t off = p.chunk_major ? (((b * p.n_head + head) * (kT / 8) + (t >> 3)) * p.hs + d) * 8 + (t & 7) : rowbase + t;
                *reinterpret_cast<u32x2 *>(p.o_hi + off) = hi;
            }
        }
    } else {
        // swapped: lane = token, registers = 4 consecutive output columns
#pragma unroll
        for (int mg = 0; mg < 4; mg++) {
            const int64_t m = m0w + 16 * mg + rho;
            f32x2 st = {0.f, 1.f};
            if constexpr (LNF && EPI != EPI_RESID) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(st) : "v"(cs_addr + 1024u + (unsigned)(mloc0 + 16 * mg + rho) * 8u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st) : : "memory");
            }
            const float osi = (LNF && EPI != EPI_RESID) ? os * st[1] : os;
            if constexpr (LNF && EPI != EPI_RESID) {                       // acc <- acc - mean * colsum; rstd rides on the output scale
#pragma unroll
                for (int ng = 0; ng < 8; ng++) {
                    f32x4 cs;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(cs) : "v"(cs_addr + (unsigned)(nloc0 + 16 * ng + 4 * q) * 4u) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cs) : : "memory");
#pragma unroll
                    for (int e = 0; e < 4; e++) c[ng][mg][e] = fmaf(-st[0], cs[e], c[ng][mg][e]);
                }
            }
            if constexpr (EPI == EPI_RESID) {
                const float sh = (p.raw_out != nullptr && p.shift != nullptr) ? p.shift[m * p.shift_stride + p.shift_offset] : 0.f;
                float rsum = 0.f, rsq = 0.f;
                f32x4 cur[8];                                              // the token's eight quads of this wave's 128 columns: all reads in flight before the first store
#pragma unroll
                for (int ng = 0; ng < 8; ng++)
                    cur[ng] = *reinterpret_cast<const f32x4 *>(p.x_out + (p.x_tiled ? xt_off(m, n0w + 16 * ng + 4 * q, p.N) : m * p.N + n0w + 16 * ng + 4 * q));
#pragma unroll
                for (int ng = 0; ng < 8; ng++) {
                    const int n = n0w + 16 * ng + 4 * q;
                    f32x4 x4 = cur[ng];
#pragma unroll
                    for (int e = 0; e < 4; e++) x4[e] += c[ng][mg][e] * os;
                    *reinterpret_cast<f32x4 *>(p.x_out + (p.x_tiled ? xt_off(m, n, p.N) : m * p.N + n)) = x4;
                    if (p.raw_out != nullptr) {                            // the next GEMM's A operand: the raw row in operand planes (see gemm16_epilogue)
                        const float v[4] = {x4[0] - sh, x4[1] - sh, x4[2] - sh, x4[3] - sh};
                        rsum += (v[0] + v[1]) + (v[2] + v[3]);
                        rsq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        u32x2 hi, lo;
                        split4<T, NP>(v, hi, lo);
                        *reinterpret_cast<u32x2 *>(p.raw_out + pk_off(m, n, 0, p.N >> 4, NP)) = hi;
                    } else rsum += (x4[0] + x4[1]) + (x4[2] + x4[3]);
                }
                if (p.rsum_out != nullptr) {                               // this wave's 128 columns of the row: the token sits in lanes rho + 16 q
                    rsum += __shfl_xor(rsum, 16); rsq += __shfl_xor(rsq, 16);
                    rsum += __shfl_xor(rsum, 32); rsq += __shfl_xor(rsq, 32);
                    if (q == 0) p.rsum_out[(size_t)(n0w >> 7) * (size_t)p.M + m] = make_float2(rsum, rsq);
                }
                continue;
            }
            if constexpr (EPI == EPI_GELU) {
                // Phi table in LDS: 16 gathers (four quads) in flight per LDS round trip
#pragma unroll
                for (int nb = 0; nb < 8; nb += 4) {
                    float v[16], fr[16];
                    f32x2 tb[16];
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        v[g] = c[nb + (g >> 2)][mg][g & 3] * osi;
                        const float tt = __builtin_amdgcn_fmed3f(fmaf(v[g], kGeluLutScale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
                        fr[g] = __builtin_amdgcn_fractf(tt);
                        asm volatile("ds_read_b64 %0, %1" : "=v"(tb[g]) : "v"(lut_addr + (unsigned)tt * 8u) : "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        asm volatile("" : "+v"(tb[g]));
                        v[g] *= fmaf(fr[g], tb[g][1], tb[g][0]);
                    }
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {
                        const int n = n0w + 16 * (nb + k4) + 4 * q;
                        u32x2 hi, lo;
                        split4<T, NP>(v + 4 * k4, hi, lo);
                        if (p.o_pk) *reinterpret_cast<u32x2 *>(p.o_hi + pk_off(m, n, 0, p.N >> 4, NP)) = hi;
                        else *reinterpret_cast<u32x2 *>(p.o_hi + m * p.N + n) = hi;
                    }
                }
                continue;
            }
            // EPI_QK: q|k planes [which][rows][n_head][256][hs] (chunk-major: [which][rows][n_head][hs/8][256][8 d])
#pragma unroll
            for (int ng = 0; ng < 8; ng++) {
                const int n = n0w + 16 * ng + 4 * q;
                const float v[4] = {c[ng][mg][0] * osi, c[ng][mg][1] * osi, c[ng][mg][2] * osi, c[ng][mg][3] * osi};
                const int which = n / p.C, cc = n - which * p.C;
                const int head = cc / p.hs, d = cc - head * p.hs;
                const int64_t bb = m >> 8;
                const int t = (int)(m & (kT - 1));
                const int64_t off = (int64_t)which * p.plane + (p.chunk_major ? (((bb * p.n_head + head) * (p.hs >> 3) + (d >> 3)) * kT + t) * 8 + (d & 7)
                                                                             : ((bb * p.n_head + head) * kT + t) * p.hs + d);
                u32x2 hi, lo;
                split4<T, NP>(v, hi, lo);
                *reinterpret_cast<u32x2 *>(p.o_hi + off) = hi;
            }
        }
    }
}

#ifndef MGPT_PK16_SALU
#define MGPT_PK16_SALU 12
#endif
constexpr int kPk16SaluBehind = MGPT_PK16_SALU;
template <class T, int EPI, int NWV, bool LNF = false>
__global__ __launch_bounds__(NWV * 64, 2) void gemm_pk16_kernel(GemmArgs p)
{
    static_assert(NWV == 8 || NWV == 4, "8 or 4 waves of 64 x 128");
    constexpr int NP = 1;
    constexpr int AF = NWV;                                // A (token) fragments per k-step = block rows / 32
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int NST = gemm_pk_nst(NP, NWV, EPI);         // ring depth in stages: even, so that a pair's two stages are consecutive slots
    static_assert(NST % 2 == 0 && NST >= 4, "ring of whole pairs");
    constexpr int STAGE = (AF + 8) * 1024;
    constexpr int PER_WAVE = (AF + 8) / NWV;
    static_assert((AF + 8) % NWV == 0, "ring shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NST][STAGE] | Phi table | LNF pieces
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> tile map: as gemm_pk_kernel (XCD-aware bands), one workgroup per tile
    const int nb = gridDim.x, ntn = p.n_tiles_n, mtn = nb / ntn;
    int id = blockIdx.x;
    if ((nb & 7) == 0) id = (id & 7) * (nb >> 3) + (id >> 3);
    constexpr int GM = 32 / AF;
    const int band = id / (GM * ntn);
    const int gm = min(GM, mtn - band * GM);
    const int rem = id - band * GM * ntn;
    const int nt = rem / gm, mt = band * GM + (rem - nt * gm);
    const int KS = p.K >> 4, NSTG = KS, NP2 = NSTG / 2;
    const unsigned char *abase = reinterpret_cast<const unsigned char *>(p.a_hi) + (size_t)mt * AF * KS * 1024 + lane * 16;
    const unsigned char *bbase = reinterpret_cast<const unsigned char *>(p.w_hi) + (size_t)nt * 8 * KS * 1024 + lane * 16;
    // stage S -> ring slot `slot` (= S % NST; the k loop carries the slot of its pair instead of dividing)
    auto issue = [&](int S, int slot) {
        unsigned char *dst = smem + (size_t)slot * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            const int f = wave + NWV * i;
            const unsigned char *src = (f < AF) ? abase + ((size_t)(f * KS + S)) * 1024 : bbase + ((size_t)((f - AF) * KS + S)) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)(dst + (size_t)f * 1024), 16, 0, 0);
        }
    };
    unsigned lut_addr = 0u;
    if (EPI == EPI_GELU && p.gelu_lut != nullptr) {        // (uniform) Phi table behind the ring, older than every ring piece
        unsigned char *dst = smem + (size_t)NST * STAGE;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(p.gelu_lut);
#pragma unroll
        for (int i = 0; i < kGeluLutN * 8 / 1024 / NWV; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(wave + NWV * i) * 1024 + lane * 16),
                                             (lds_void_t *)(dst + (size_t)(wave + NWV * i) * 1024), 16, 0, 0);
        lut_addr = (unsigned)(size_t)dst;
    }
    unsigned cs_addr = 0u;
    if constexpr (LNF) {                                   // column sums + (mean, rstd) of the block's rows: older than every ring piece
        unsigned char *dst = smem + (size_t)NST * STAGE + (EPI == EPI_GELU ? kGeluLutN * 8 : 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(reinterpret_cast<const unsigned char *>(p.colsum + nt * 256) + lane * 16), (lds_void_t *)dst, 16, 0, 0);
        if (wave == 1 || (wave == 2 && AF == 8))
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(reinterpret_cast<const unsigned char *>(p.ln_stats + (size_t)mt * (AF * 32)) + (wave - 1) * 1024 + lane * 16),
                                             (lds_void_t *)(dst + wave * 1024), 16, 0, 0);
        cs_addr = (unsigned)(size_t)dst;
    }
#pragma unroll
    for (int S = 0; S < NST; S++) issue(S, S);             // K >= 128 and K % 64 == 0 are checked by the launcher

    f32x4 c[8][4];
#pragma unroll
    for (int ng = 0; ng < 8; ng++)
#pragma unroll
        for (int mg = 0; mg < 4; mg++) c[ng][mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's 16 bytes inside a 32-row fragment: row rho of the even 16-row group, k half q & 1; + (q >> 1) stages
    const unsigned char *lbase = smem + (size_t)(q >> 1) * STAGE + (size_t)(rho + 32 * (q & 1)) * 16;
    const unsigned char *wfr = lbase + (size_t)(AF + wn * 4) * 1024;       // weight fragments of this wave's 128 columns (4 fragments)
    const unsigned char *tfr = lbase + (size_t)(wm * 2) * 1024;            // token fragments of this wave's 64 rows (2 fragments)
    u32x4 wa[8], ta[2][4];
    auto rd_w = [&](int ps, int ng) { return *reinterpret_cast<const u32x4 *>(wfr + (size_t)(2 * ps) * STAGE + (ng >> 1) * 1024 + (ng & 1) * 256); };
    auto rd_t = [&](int ps, int mg) { return *reinterpret_cast<const u32x4 *>(tfr + (size_t)(2 * ps) * STAGE + (mg >> 1) * 1024 + (mg & 1) * 256); };
    // one pair (its stages sit in ring slots 2 ps, 2 ps + 1): 32 MFMAs.  FETCH: the operands of the next pair (slots 2 psn ..) are read behind them, a weight
    // group's fragment right after the group's four MFMAs.  ISSUE: this pair's slots -- read during the pair before -- are refilled with the stages NST ahead;
    // the refill stands in front of the last two groups' MFMAs in program order (to hipcc it is a store to LDS: the ten reads before it stay before it, the two
    // after it stay after) and goes out one piece behind each of their first MFMAs
    auto pair = [&](int pr, int buf, int ps, int psn, auto fetch_c, auto issue_c) {
        constexpr bool FETCH = decltype(fetch_c)::value, ISSUE = decltype(issue_c)::value;
        constexpr int NPIECE = ISSUE ? 2 * PER_WAVE : 0;
#pragma unroll
        for (int ng = 0; ng < 8; ng++) {
            if (ng == 6) { if constexpr (ISSUE) { issue(2 * pr + NST, 2 * ps); issue(2 * pr + NST + 1, 2 * ps + 1); } }
#pragma unroll
            for (int mg = 0; mg < 4; mg++)
                c[ng][mg] = SWAP ? T::mfma16(wa[ng], ta[buf][mg], c[ng][mg]) : T::mfma16(ta[buf][mg], wa[ng], c[ng][mg]);
            if constexpr (FETCH) {
                wa[ng] = rd_w(psn, ng);
                if (ng < 4) ta[buf ^ 1][ng] = rd_t(psn, ng);
            }
        }
        if constexpr (FETCH || ISSUE) {
#pragma unroll
            for (int ng = 0; ng < 8; ng++) {
                if (ng < 6) {
                    if (ng == 0 && kPk16SaluBehind > 0) {                 // the pair's slot / address arithmetic behind the first MFMA (as in gemm_pk_kernel)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x004, kPk16SaluBehind, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    } else
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if constexpr (FETCH) { if (ng < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                } else {
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (4 * (ng - 6) + n < NPIECE) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                    if constexpr (FETCH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Y = std::true_type; using N_ = std::false_type;

    // stages 0, 1 have landed for everyone; the operands of pair 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ng = 0; ng < 8; ng++) wa[ng] = rd_w(0, ng);
#pragma unroll
    for (int mg = 0; mg < 4; mg++) ta[0][mg] = rd_t(0, mg);
    constexpr int NPS = NST / 2;                            // ring slots, in pairs
    int ps = 0;                                            // pair slot of the pair about to run
    auto next_ps = [&](int x) { return x + 1 == NPS ? 0 : x + 1; };
    // main part: pair pr + 1's stages have landed (the NST - 4 stages issued after them may be in flight), everyone holds pair pr's operands in registers.
    // The last four pairs are written out (which of them still refill the ring is a compile-time fact): K % 64 == 0 and K >= 128 are checked by the launcher.
    constexpr int TAILP = 4;
    static_assert(NST / 2 <= TAILP, "every pair that no longer refills is in the written-out tail");
    int pr = 0;
#pragma unroll 1
    for (; pr < NP2 - TAILP; pr += 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 4) * PER_WAVE) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int ps1 = next_ps(ps), ps2 = next_ps(ps1);
        pair(pr, 0, ps, ps1, Y{}, Y{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 4) * PER_WAVE) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pair(pr + 1, 1, ps1, ps2, Y{}, Y{});
        ps = ps2;
    }
    // tail pair t = 0 .. 3 (pr = NP2 - 4 + t): refills while t < 4 - NST / 2, fetches while t < 3; everything issued has to land (vmcnt(0): a counted
    // form would need a count per pair)
    auto tail_pair = [&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (t < TAILP - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int psn = next_ps(ps);
        pair(pr + t, t & 1, ps, psn, std::integral_constant<bool, (t < TAILP - 1)>{}, std::integral_constant<bool, (t < TAILP - NST / 2)>{});
        ps = psn;
    };
    tail_pair(std::integral_constant<int, 0>{}); tail_pair(std::integral_constant<int, 1>{});
    tail_pair(std::integral_constant<int, 2>{}); tail_pair(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (nothing is in flight; LNF pieces and the Phi table have landed)
    gemm16x16_epilogue<T, EPI, LNF>(p, c, (int64_t)mt * (AF * 32) + wm * 64, nt * 256 + wn * 128, wm * 64, wn * 128, rho, q, lut_addr, cs_addr);
}

// ---------------------------------------------------------------------------------------------
// Non-causal attention on planes (model.py:58-60): one workgroup per (row, head); K and V^T planes
// staged once in LDS, each wave owns 2 query tiles of 32.  S^T = K Q^T so a lane owns one query:
// running (max, sum) softmax is in-lane (+1 exchange with lane^32), P converts in-lane into the
// B-operand of O^T = V^T P^T.  Keys inside a 32-key tile are visited in the order `bits 2<->3 swapped`
// so that the 8 keys a lane holds for one PV MFMA are contiguous in V^T (one 16-byte LDS read).
//   q, k planes [rows][n_head][256][HS]; v^T planes [rows][n_head][HS][256]; y planes [rows*256][C]
// ---------------------------------------------------------------------------------------------
// NW waves per (row, head) workgroup: 8 (one query tile each) keeps two waves on every SIMD even when K and V^T of the
// head fill most of LDS (hs = 64 in the split mode: 141 KiB, one workgroup per CU).
template <class T, int NP, int HS, int NW = 8>
__global__ __launch_bounds__(NW * 64) void attn16_kernel(const uint16_t *__restrict__ q_hi, const uint16_t *__restrict__ q_lo,
                                                     const uint16_t *__restrict__ k_hi, const uint16_t *__restrict__ k_lo,
                                                     const uint16_t *__restrict__ vt_hi, const uint16_t *__restrict__ vt_lo,
                                                     uint16_t *__restrict__ y_hi, uint16_t *__restrict__ y_lo, int n_head,
                                                     float scale_log2e, int y_pk, int chunk_major, int last_only)
{
    // last_only (last layer, model.py:186 reads position 255 only): all keys and values, but only the query tile that holds
    // token 255, and only that token's output row, written to row b of a compact [rows][C] matrix (y_pk layout)
    constexpr int KRS = (HS + 8) * 2;           // K row stride in bytes  (80 for HS = 32)
    constexpr int VRS = (kT + 8) * 2;           // V^T row stride in bytes (528)
    constexpr int KS = HS / 16;                 // k-steps of the S product
    constexpr int DT = HS / 32;                 // d tiles of the output
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem;                                   // [NP][256][KRS]
    unsigned char *sV = smem + NP * kT * KRS;                   // [NP][HS][VRS]

    const int bh = blockIdx.x;
    const int b = bh / n_head, head = bh - b * n_head;
    const int C = n_head * HS;
    const size_t base = (size_t)bh * kT * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int qt_first = last_only ? kT / 32 - 1 + (wave == 0 ? 0 : kT) : wave;
    u32x4 qf[KS][2];                                             // B operand: Q[query r][16 ks + 8 h ..]
    auto load_q = [&](int qt, u32x4 (&dst)[KS][2]) {
        if (qt >= kT / 32) return;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                dst[ks][pl] = *reinterpret_cast<const u32x4 *>((pl == 0 ? q_hi : q_lo) + base +
                                                              (chunk_major ? ((size_t)(ks * 2 + h) * kT + qt * 32 + r) * 8
                                                                           : (size_t)(qt * 32 + r) * HS + ks * 16 + h * 8));
    };
    // stage K [256][HS] and V^T [HS][256] planes (16-byte chunks).  ALL of a thread's loads are issued before the first LDS store (round 5; rounds 1-4: a
    // `for (idx = tid; idx < N; idx += threads)` loop, which hipcc cannot unroll -- it ran load, s_waitcnt vmcnt(0), ds_write eight times in sequence, eight
    // memory round trips per workgroup before its first MFMA: the kernel sat at 0.51 of the HBM rate with 0.22 of the MFMA rate; -DMGPT_AB_ATTN16_SERIAL_STAGE)
    {
        constexpr int NCH = NP * kT * (HS / 8);             // 16-byte chunks of the K planes (= of the V^T planes)
        constexpr int NIT = NCH / (NW * 64);
        static_assert(NIT * NW * 64 == NCH, "whole chunks per thread");
        // chunk idx of the K planes: source element offset and LDS byte offset
        auto k_src = [&](int idx, const uint16_t *&src, unsigned &dst) {
            const int pl = idx / (kT * (HS / 8)), rem = idx - pl * (kT * (HS / 8));
            int row, c;
            if (chunk_major) { c = rem / kT; row = rem - c * kT; src = (pl == 0 ? k_hi : k_lo) + base + (size_t)rem * 8; }   // planes written by the packed GEMM: [hs/8][256][8]
            else { row = rem / (HS / 8); c = rem - row * (HS / 8); src = (pl == 0 ? k_hi : k_lo) + base + (size_t)row * HS + c * 8; }
            dst = (unsigned)(pl * kT * KRS + row * KRS + c * 16);
        };
        auto v_src = [&](int idx, const uint16_t *&src, unsigned &dst) {
            const int pl = idx / (HS * (kT / 8)), rem = idx - pl * (HS * (kT / 8));
            int row, c;
            if (chunk_major) { c = rem / HS; row = rem - c * HS; src = (pl == 0 ? vt_hi : vt_lo) + base + (size_t)rem * 8; }   // [256/8][hs][8]
            else { row = rem / (kT / 8); c = rem - row * (kT / 8); src = (pl == 0 ? vt_hi : vt_lo) + base + (size_t)row * kT + c * 8; }
            dst = (unsigned)(pl * HS * VRS + row * VRS + c * 16);
        };
#if defined(MGPT_AB_ATTN16_SERIAL_STAGE)
        for (int idx = tid; idx < NCH; idx += NW * 64) {
            const uint16_t *src; unsigned dst;
            k_src(idx, src, dst);
            *reinterpret_cast<u32x4 *>(sK + dst) = *reinterpret_cast<const u32x4 *>(src);
        }
        for (int idx = tid; idx < NCH; idx += NW * 64) {
            const uint16_t *src; unsigned dst;
            v_src(idx, src, dst);
            *reinterpret_cast<u32x4 *>(sV + dst) = *reinterpret_cast<const u32x4 *>(src);
        }
#else
        u32x4 kreg[NIT], vreg[NIT];
        unsigned kdst[NIT], vdst[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) { const uint16_t *src; k_src(tid + it * NW * 64, src, kdst[it]); kreg[it] = *reinterpret_cast<const u32x4 *>(src); }
#pragma unroll
        for (int it = 0; it < NIT; it++) { const uint16_t *src; v_src(tid + it * NW * 64, src, vdst[it]); vreg[it] = *reinterpret_cast<const u32x4 *>(src); }
        load_q(qt_first, qf);                               // this wave's first query tile rides with the staging loads
#pragma unroll
        for (int it = 0; it < NIT; it++) *reinterpret_cast<u32x4 *>(sK + kdst[it]) = kreg[it];
#pragma unroll
        for (int it = 0; it < NIT; it++) *reinterpret_cast<u32x4 *>(sV + vdst[it]) = vreg[it];
#endif
    }
    __syncthreads();

    // the S^T tile row this lane feeds as A-operand is key `kperm` of the tile (bits 2 and 3 of r swapped)
    const int kperm = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);

    for (int qt = qt_first; qt < kT / 32; qt += NW) {
#if defined(MGPT_AB_ATTN16_SERIAL_STAGE)
        load_q(qt, qf);
#else
        if (qt != qt_first) load_q(qt, qf);
#endif
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int g = 0; g < 16; g++) o[dt][g] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
        for (int kt = 0; kt < kT / 32; kt++) {
            f32x16 s;
#pragma unroll
            for (int g = 0; g < 16; g++) s[g] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                u32x4 kf[2];
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    kf[pl] = *reinterpret_cast<const u32x4 *>(sK + (size_t)pl * kT * KRS + (kt * 32 + kperm) * KRS + ks * 32 + h * 16);
                s = mma<T, NP>(kf, qf[ks], s);
            }
            // s[g] = S[query r][key kt*32 + 16*(g>>3) + 8*h + (g&7)]   (after the bit-swap permutation)
            float mx = s[0];
#pragma unroll
            for (int g = 1; g < 16; g++) mx = fmaxf(mx, s[g]);
            mx = half_max32(mx);
            // The kernel runs at the SUM of its MFMA and VALU issue time, so the softmax arithmetic is kept short: the rescale of o and l
            // (16 DT + 3 instructions) only when some query's running maximum moved (wave-uniform branch; alpha = 1 exactly otherwise), and one fma per
            // score in front of the exp2 (-DMGPT_AB_ATTN16_PLAIN: round 1's form, rescale every tile and (s - m) * scale)
#if defined(MGPT_AB_ATTN16_PLAIN)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) o[dt][g] *= alpha;
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                s[g] = __builtin_amdgcn_exp2f((s[g] - m_new) * scale_log2e);
                psum += s[g];
            }
#else
            if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
#pragma unroll
                    for (int g = 0; g < 16; g++) o[dt][g] *= alpha;
                m_run = m_new;
            }
            const float nm = -m_run * scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                s[g] = __builtin_amdgcn_exp2f(fmaf(s[g], scale_log2e, nm));
                psum += s[g];
            }
#endif
            psum = half_sum32(psum);
            l_run += psum;
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {                      // two PV MFMAs of 16 keys each
                const float pv0[4] = {s[8 * mm], s[8 * mm + 1], s[8 * mm + 2], s[8 * mm + 3]};
                const float pv1[4] = {s[8 * mm + 4], s[8 * mm + 5], s[8 * mm + 6], s[8 * mm + 7]};
                u32x2 h0, l0, h1, l1;
                split4<T, NP>(pv0, h0, l0);
                split4<T, NP>(pv1, h1, l1);
                u32x4 pf[2];
                pf[0][0] = h0[0]; pf[0][1] = h0[1]; pf[0][2] = h1[0]; pf[0][3] = h1[1];
                pf[1][0] = l0[0]; pf[1][1] = l0[1]; pf[1][2] = l1[0]; pf[1][3] = l1[1];
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    u32x4 vf[2];                                  // A operand: V^T[d = dt*32 + r][keys kt*32 + 16 mm + 8 h ..]
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        vf[pl] = *reinterpret_cast<const u32x4 *>(sV + (size_t)pl * HS * VRS + (dt * 32 + r) * VRS + (kt * 32 + 16 * mm + 8 * h) * 2);
                    o[dt] = mma<T, NP>(vf, pf, o[dt]);
                }
            }
        }
        const float inv = 1.0f / l_run;
        // o[dt][g] = O[query r][d = dt*32 + (g&3) + 8*(g>>2) + 4*h] -> y planes [b*256 + t][head*HS + d]  (model.py:68)
        const size_t yrow = ((size_t)b * kT + qt * 32 + r) * C + head * HS;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const float v[4] = {o[dt][4 * gq] * inv, o[dt][4 * gq + 1] * inv, o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv};
                u32x2 hi, lo;
                split4<T, NP>(v, hi, lo);
                if (last_only && r != 31) continue;
                if (y_pk) {                                 // y feeds gemm_pk_kernel: PK layout, y_hi = base
                    const int64_t m = last_only ? (int64_t)b : (int64_t)b * kT + qt * 32 + r;
                    const int n = head * HS + dt * 32 + 8 * gq + 4 * h;
                    *reinterpret_cast<u32x2 *>(y_hi + pk_off(m, n, 0, C >> 4, NP)) = hi;
                    if (NP == 2) *reinterpret_cast<u32x2 *>(y_hi + pk_off(m, n, 1, C >> 4, NP)) = lo;
                } else {
                    *reinterpret_cast<u32x2 *>(y_hi + yrow + dt * 32 + 8 * gq + 4 * h) = hi;
                    if (NP == 2) *reinterpret_cast<u32x2 *>(y_lo + yrow + dt * 32 + 8 * gq + 4 * h) = lo;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused MLP block: x <- x + c_proj(GELU(c_fc(LayerNorm(x))))   (model.py:84-89, 103), hidden never leaves the CU.
// A wave owns 32 tokens for the whole kernel and keeps every activation in registers in the "swapped"
// C/D layout (lane = token r (+32 h), register g of tile j = feature 32 j + (g & 3) + 8 (g >> 2) + 4 h):
//   * LayerNorm statistics are in-lane sums (+ one exchange with lane ^ 32);
//   * the normalised row, split into fp16 planes, IS the B operand of the c_fc MFMAs -- the k-slot -> feature
//     permutation this implies is baked into the packed weights (pack_mlp_kernel), so nothing is transposed;
//   * each 32-wide hidden tile comes out of the MFMA in the same layout, goes through GELU and is again
//     directly the B operand of the c_proj MFMAs that accumulate the 32 x C output tile in registers.
// Weights stream through LDS in per-hidden-tile packets ([c_fc fragments | c_proj fragments], each fragment-plane
// 1 KiB = 64 lanes x 16 B), double-buffered with direct global->LDS loads, shared by the 4 waves of the workgroup.
// HBM traffic: x read + x write (+ 8 B/token stats) -- 1.3 KB per token instead of ~6.4 KB for the unfused pair.
// ---------------------------------------------------------------------------------------------
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_mlp_kernel(const float *__restrict__ fc_w, const float *__restrict__ pj_w,
                                                       uint16_t *__restrict__ out, int C, float scale1, float scale2)
{
    // one thread = one (hidden tile t, fragment f, lane): 8 k-slots, both planes
    const int CT = C / 32, KS = C / 16;
    const int frags = KS + 2 * CT;                         // per hidden tile: c_fc k-steps, then c_proj (j, kk)
    const int NT = 4 * C / 32;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)NT * frags * 64) return;
    const int lane = (int)(gid & 63);
    const int f = (int)((gid >> 6) % frags), t = (int)((gid >> 6) / frags);
    const int i = lane & 31, h = lane >> 5;
    float v[8];
    if (f < KS) {                                          // c_fc: A rows = hidden units, k-slots = features
        const int ks = f, u = 32 * t + i;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int g = 8 * (ks & 1) + e;
            const int feat = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
            v[e] = fc_w[(size_t)u * C + feat] * scale1;
        }
    } else {                                               // c_proj: A rows = output features, k-slots = hidden units
        const int j = (f - KS) >> 1, kk = (f - KS) & 1, o = 32 * j + i;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int g = 8 * kk + e;
            const int u = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
            v[e] = pj_w[(size_t)o * (4 * C) + u] * scale2;
        }
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    // layout: [t][f][plane][lane][8 halfs]
    uint16_t *dst = out + (((size_t)t * frags + f) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}


// (C = 64, 160: the 2M and tiny shapes; C = 256 has its own pipelined kernel in gpt_kernels_c256.h)
template <class T, int NP, int CT, int NW = 8, int NFOLD = 0, int NBUF = 3, int NCH = 2>
__global__ __launch_bounds__(NW * 64, 2) void mlp_fused_kernel(float *__restrict__ x, const float *__restrict__ gain,
                                                            const uint16_t *__restrict__ wpk, float inv1, float inv2,
                                                            float2 *__restrict__ stats_out, int M,
                                                            const float2 *__restrict__ gelu_lut,
                                                            const float *__restrict__ fold = nullptr, int64_t fold_stride = 0)
{
    // NFOLD > 0 (small launches, after a head-parallel attn_block_kernel): NFOLD partial sums in x's layout, fold_stride floats
    // apart, are added to the row in index order before anything else and the sum is written back (a token has ONE owner here).
    // A compile-time count: every load of the fold is in flight at once (as a run-time loop hipcc serialised them, +23 us per launch)
    constexpr int C = CT * 32, KS = C / 16, NT = 4 * CT;
    constexpr int LUT_BYTES = kGeluLutN * 8;               // the Phi table sits behind the ring: [NBUF][PKT][LUT]
    static_assert((LUT_BYTES / 1024) % NW == 0, "every wave stages the same number of table pieces");
    constexpr int FRAGS = KS + 2 * CT;                     // fragments per hidden tile
    constexpr int PKT = FRAGS * NP * 1024;                 // bytes per hidden-tile packet
    constexpr int PER_WAVE = (FRAGS * NP + NW - 1) / NW;   // DMA instructions a wave issues per packet
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NBUF][PKT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m = (int64_t)blockIdx.x * (NW * 32) + wave * 32 + r;     // this lane's token
    // x is chunk-major (xt_off): a wave's load / store is 1 KiB contiguous; chunk c = 4 j + gq sits c * 256 floats up
    float *xt = x + (m - r) * C + r * 8 + 4 * h;

    // ---- stream helper: packet t -> LDS buffer (t & 1); every wave moves FRAGS*NP/4 fragment-planes of 1 KiB ----
    auto issue = [&](int t) {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)t * PKT;
        unsigned char *dst = smem + (size_t)(t % NBUF) * PKT;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            // every wave issues exactly PER_WAVE pieces so that the counted vmcnt waits below are exact; a wave whose
            // share runs past the packet re-loads the last piece (same bytes to the same place: harmless)
            const int c = min(wave + NW * i, FRAGS * NP - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)c * 1024 + lane * 16),
                                             (lds_void_t *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
    };
    // GELU table -> LDS by the same direct loads; older than every ring piece, so the first counted wait covers it
    {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(gelu_lut);
        unsigned char *dst = smem + (size_t)NBUF * PKT;
#pragma unroll
        for (int i = 0; i < LUT_BYTES / 1024 / NW; i++)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(wave + NW * i) * 1024 + lane * 16),
                                             (lds_void_t *)(dst + (size_t)(wave + NW * i) * 1024), 16, 0, 0);
    }
    issue(0);
    if (NBUF > 2) issue(1);
    const unsigned lut_addr = (unsigned)(size_t)(smem + (size_t)NBUF * PKT);
    const float lut_scale = inv1 * kGeluLutScale;

    // ---- load the 32 x C row block in swapped layout, LayerNorm in-lane ----
    f32x16 acc[CT];                                        // x now, output accumulator later
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xt + (4 * j + gq) * 256);
            acc[j][4 * gq] = v[0]; acc[j][4 * gq + 1] = v[1]; acc[j][4 * gq + 2] = v[2]; acc[j][4 * gq + 3] = v[3];
        }
    if constexpr (NFOLD > 0) {
        const float *fp = fold + (xt - x);
#pragma unroll
        for (int p = 0; p < NFOLD; p++) {
            f32x4 t[4 * CT];
#pragma unroll
            for (int c = 0; c < 4 * CT; c++) t[c] = *reinterpret_cast<const f32x4 *>(fp + (size_t)p * fold_stride + c * 256);
#pragma unroll
            for (int c = 0; c < 4 * CT; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[c >> 2][4 * (c & 3) + e] += t[c][e];
        }
#pragma unroll
        for (int c = 0; c < 4 * CT; c++) {
            const f32x4 v = {acc[c >> 2][4 * (c & 3)], acc[c >> 2][4 * (c & 3) + 1], acc[c >> 2][4 * (c & 3) + 2], acc[c >> 2][4 * (c & 3) + 3]};
            *reinterpret_cast<f32x4 *>(xt + c * 256) = v;
        }
    }
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) s += (acc[j][4 * gq] + acc[j][4 * gq + 1]) + (acc[j][4 * gq + 2] + acc[j][4 * gq + 3]);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) { const float d = acc[j][g] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q / (float)C + 1e-5f);
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

    // packet 0 must have landed; with 3 buffers packet 1 (the newest PER_WAVE pieces of this wave) may still fly
    if (NBUF > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int t = 0; t < NT; t++) {
        // refill the buffer that was read during tile t-1 (all waves are past the barrier that ended it)
        if (t + NBUF - 1 < NT) issue(t + NBUF - 1);
        const unsigned char *pk = smem + (size_t)(t % NBUF) * PKT + lane * 16;
        // ---- hidden tile: c_fc output for (token r, hidden 32 t + (g&3) + 8 (g>>2) + 4 h) ----
        // Two partial accumulators (even / odd k-steps) with their MFMA passes interleaved: a 32x32x16 MFMA
        // that reads the previous one's result as C waits for its full latency (~2x the issue interval), so a
        // single dependent chain runs the matrix pipe at half rate (measured: tools/abl_mlp.sh).
        f32x16 hch[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int g = 0; g < 16; g++) hch[c][g] = 0.f;
        {
#pragma unroll
            for (int ks = 0; ks < KS; ks += NCH) {
                u32x4 w[NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; c++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        w[c][pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((ks + c) * NP + pl) * 1024);
                if (NP == 2) {
#pragma unroll
                    for (int c = 0; c < NCH; c++) hch[c] = T::mfma(w[c][1], xn[ks + c][0], hch[c]);
#pragma unroll
                    for (int c = 0; c < NCH; c++) hch[c] = T::mfma(w[c][0], xn[ks + c][1], hch[c]);
                }
#pragma unroll
                for (int c = 0; c < NCH; c++) hch[c] = T::mfma(w[c][0], xn[ks + c][0], hch[c]);
            }
            // fragment reads run one group of k-steps ahead of the MFMAs that consume them
            __builtin_amdgcn_sched_group_barrier(0x100, NCH * NP, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ks += NCH) {
                if (ks + NCH < KS) __builtin_amdgcn_sched_group_barrier(0x100, NCH * NP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NCH * (NP == 2 ? 3 : 1), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 hacc = hch[0];
#pragma unroll
        for (int c = 1; c < NCH; c++)
#pragma unroll
            for (int g = 0; g < 16; g++) hacc[g] += hch[c][g];
        // GELU by table (8 VALU + one 8-byte LDS gather per value; the rational form costs 17): all 16 gathers of the tile
        // go out first (asm: invisible to hipcc's LDS-DMA ordering), one wait, then interpolate
        float gv[16], gf[16];
        f32x2 gt[16];
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float hv = hacc[g];
            gv[g] = hv * inv1;
            const float tt = __builtin_amdgcn_fmed3f(fmaf(hv, lut_scale, kGeluLutBias), 0.0f, (float)kGeluLutN - 0.002f);
            gf[g] = __builtin_amdgcn_fractf(tt);
            asm volatile("ds_read_b64 %0, %1" : "=v"(gt[g]) : "v"(lut_addr + (unsigned)tt * 8u) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; g++) asm volatile("" : "+v"(gt[g]));
        u32x4 hf[2][2];                                    // [kk][plane]: B operand of c_proj
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = gv[8 * kk + e] * fmaf(gf[8 * kk + e], gt[8 * kk + e][1], gt[8 * kk + e][0]);
                v1[e] = gv[8 * kk + 4 + e] * fmaf(gf[8 * kk + 4 + e], gt[8 * kk + 4 + e][1], gt[8 * kk + 4 + e][0]);
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            hf[kk][0][0] = h0[0]; hf[kk][0][1] = h0[1]; hf[kk][0][2] = h1[0]; hf[kk][0][3] = h1[1];
            hf[kk][1][0] = l0[0]; hf[kk][1][1] = l0[1]; hf[kk][1][2] = l1[0]; hf[kk][1][3] = l1[1];
        }
        // ---- c_proj: for each kk the CT output tiles are independent accumulators; interleave them pairwise ----
        constexpr int NG = 2 * CT;                         // (kk, j) groups, visited kk-major so neighbours differ in j
        static_assert(NG % NCH == 0 && KS % NCH == 0, "chain count must divide the k-steps and the output groups");
        {
#pragma unroll
            for (int gi = 0; gi < NG; gi += NCH) {
                u32x4 w[NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; c++)
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        w[c][pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((KS + 2 * ((gi + c) % CT) + (gi + c) / CT) * NP + pl) * 1024);
                if (NP == 2) {
#pragma unroll
                    for (int c = 0; c < NCH; c++) acc[(gi + c) % CT] = T::mfma(w[c][1], hf[(gi + c) / CT][0], acc[(gi + c) % CT]);
#pragma unroll
                    for (int c = 0; c < NCH; c++) acc[(gi + c) % CT] = T::mfma(w[c][0], hf[(gi + c) / CT][1], acc[(gi + c) % CT]);
                }
#pragma unroll
                for (int c = 0; c < NCH; c++) acc[(gi + c) % CT] = T::mfma(w[c][0], hf[(gi + c) / CT][0], acc[(gi + c) % CT]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, NCH * NP, 1);
#pragma unroll
            for (int gi = 0; gi < NG; gi += NCH) {
                if (gi + NCH < NG) __builtin_amdgcn_sched_group_barrier(0x100, NCH * NP, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, NCH * (NP == 2 ? 3 : 1), 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // packet t+1 must have landed before anyone reads it; the pieces of packet t+2 issued at the top of this
        // iteration (the newest PER_WAVE of this wave) may stay in flight across the barrier.
        // (raw s_barrier: __syncthreads() would drain every outstanding LDS-DMA, cdna guide section 5)
        if (NBUF > 2 && t + NBUF - 1 < NT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... everyone's share landed; and packet t is no longer read
    }

    // ---- residual add, store, LayerNorm statistics of the new row for the next kernel ----
    // (the row's addresses are formed again here from an opaque copy of xt: as values shared with the prologue's loads hipcc kept five 64-bit
    //  bases alive across the tile loop in scratch and reloaded them between the residual loads, one memory round trip per reload; and
    //  all residual reads are in flight before the first store -- the stores to x would otherwise keep the next read from moving up)
    //  (split mode; the one-plane variant never spilled and is 2.6 % faster with the plain read-modify-write loop: A/B, round 4)
    float *xt2 = xt;
    if (NP == 2) asm volatile("" : "+v"(xt2));
    float s2 = 0.f;
    f32x4 cur[CT][4];
    if (NP == 2) {
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) cur[j][gq] = *reinterpret_cast<const f32x4 *>(xt2 + (4 * j + gq) * 256);
    }
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            f32x4 c = NP == 2 ? cur[j][gq] : *reinterpret_cast<const f32x4 *>(xt2 + (4 * j + gq) * 256);
#pragma unroll
            for (int e = 0; e < 4; e++) { c[e] += acc[j][4 * gq + e] * inv2; acc[j][4 * gq + e] = c[e]; }
            *reinterpret_cast<f32x4 *>(xt2 + (4 * j + gq) * 256) = c;
            s2 += (c[0] + c[1]) + (c[2] + c[3]);
        }
    if (stats_out != nullptr) {
        s2 += __shfl_xor(s2, 32);
        const float mean2 = s2 / (float)C;
        float q2 = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) { const float d = acc[j][g] - mean2; q2 += d * d; }
        q2 += __shfl_xor(q2, 32);
        if (h == 0) stats_out[m] = make_float2(mean2, rsqrtf(q2 / (float)C + 1e-5f));
    }
}

// ---------------------------------------------------------------------------------------------
// c_attn.weight for attn_block_kernel: [tile][k-step][plane][lane][8]  (rows 32 tile .. +32 of c_attn.weight, k-slots
// permuted like the MLP's: the normalised tokens are MFMA operand planes in the swapped C/D register layout)
// ---------------------------------------------------------------------------------------------
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_rows_perm_kernel(const float *__restrict__ w, uint16_t *__restrict__ out,
                                                             int n_tiles, int C, float scale)
{
    const int KS = C / 16;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n_tiles * KS * 64) return;
    const int lane = (int)(gid & 63), ks = (int)((gid >> 6) % KS), t = (int)((gid >> 6) / KS);
    const int i = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int g = 8 * (ks & 1) + e;
        const int feat = 32 * (ks >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h;
        v[e] = w[(size_t)(32 * t + i) * C + feat] * scale;
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)t * KS + ks) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// out-projection weights for attn_block_kernel: [head t][out tile j][kk][plane][lane][8]; A rows = output features
// 32 j + i, k-slots = head features 32 t + tau(8 kk + e, h)  (w is [C_out][K_total] row-major)
template <class T, int NP>
__global__ __launch_bounds__(256) void pack_cols_perm_kernel(const float *__restrict__ w, uint16_t *__restrict__ out,
                                                             int n_ktiles, int C_out, int K_total, float scale)
{
    const int CT = C_out / 32;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n_ktiles * CT * 2 * 64) return;
    const int lane = (int)(gid & 63), f = (int)((gid >> 6) % (2 * CT)), t = (int)((gid >> 6) / (2 * CT));
    const int j = f >> 1, kk = f & 1, i = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int g = 8 * kk + e;
        v[e] = w[(size_t)(32 * j + i) * K_total + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h] * scale;
    }
    u32x2 h0, l0, h1, l1;
    split4<T, NP>(v, h0, l0);
    split4<T, NP>(v + 4, h1, l1);
    u32x4 hi, lo;
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
    uint16_t *dst = out + (((size_t)t * 2 * CT + f) * NP) * 512 + (size_t)lane * 8;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    if (NP == 2) *reinterpret_cast<u32x4 *>(dst + 512) = lo;
}

// ---------------------------------------------------------------------------------------------
// Fused attention block front half: LayerNorm + QKV projection + non-causal attention for ONE row per
// workgroup (8 waves x 32 tokens), heads visited one after the other; q, k, v never touch HBM.
//   per head:  q,k (swapped) and v (natural) tiles from the wave's register-resident normalised tokens and the
//              DMA-streamed c_attn fragments  ->  k, v^T planes of the head into LDS (all 8 waves contribute
//              their 32 keys), q stays in registers as the B operand  ->  barrier  ->  every wave runs its 32
//              queries against the 256 keys (online softmax as in attn16_kernel)  ->  y planes of the head.
// Layout trick that makes this transposition-free: the C/D register -> row map tau(g,h) = (g&3)+8(g>>2)+4h is
// the same for "d of a token" (swapped q/k tiles), "token of a d" (natural v tile) and "key of a query" (S^T
// tile), so register octets [8m, 8m+8) are directly MFMA k-slot groups everywhere, and the LDS images are written
// and read with the same (row, octet, half) address.
// ---------------------------------------------------------------------------------------------
// PROJ: also apply the output projection per head (c_proj fragments are the 4th packet of every head) and the
// residual add: x <- x + c_proj(attention), LayerNorm statistics of the new row to stats_out; y planes unused.
// LAST (needs PROJ): last layer -- only position 255 feeds ln_f and the head (model.py:186), so only K and V are
// needed for all tokens; q, the attention, c_proj and the residual run for the wave that owns token 255 only, and
// the new row of token 255 goes to the compact buffer x_last[row][C] (the MLP and the head then run on that).
// EMBED (first layer): the residual row is not read from x but formed here as wte[token] + wpe[position]
// (model.py:171-175), which removes the embedding kernel's write and this kernel's first read of x.
// HP (head-parallel, small launches: needs PROJ, excludes EMBED): one workgroup per (row, head) instead of per row -- with a few dozen
// rows (one environment: BASELINE cfg1) a row's five heads run on five CUs at once instead of one after the other on one.  The
// workgroup forms the row's LayerNorm itself, runs its head and writes that head's c_proj contribution (times the projection's
// 1 / scale) to part_out + head * part_stride in x's own layout (LAST: the compact last-token layout); x is NOT touched -- the
// next kernel (mlp_fused_kernel's / gather_last_kernel's fold arguments) adds the partial sums in head order.
template <class T, int NP, int CT, bool PROJ, bool LAST = false, bool EMBED = false, bool HP = false>
__global__ __launch_bounds__(512, 2) void attn_block_kernel(float *__restrict__ x, const float *__restrict__ gain,
                                                             const uint16_t *__restrict__ wpk, float inv_scale,
                                                             uint16_t *__restrict__ y_hi, uint16_t *__restrict__ y_lo,
                                                             int n_head, float scale_log2e,
                                                             const uint16_t *__restrict__ ppk, float inv_scale_p,
                                                             float2 *__restrict__ stats_out, float *__restrict__ x_last,
                                                             const uint8_t *__restrict__ tokens, const float *__restrict__ wte,
                                                             const float *__restrict__ wpe, float *__restrict__ part_out = nullptr,
                                                             int64_t part_stride = 0)
{
    static_assert(!EMBED || (PROJ && !LAST), "EMBED is the first layer of a model with more than one layer");
    static_assert(!LAST || PROJ, "LAST implies PROJ");
    static_assert(!HP || (PROJ && !EMBED), "HP writes c_proj partial sums and reads x");
    constexpr int C = CT * 32, KS = C / 16, NW = 8, HS = 32;
    constexpr int F = KS * NP, PKT = F * 1024, PER_WAVE = (F + NW - 1) / NW;
    constexpr int KROW = 80, VROW = 528;                                  // padded LDS rows (bytes): conflict-free b128 reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sK = smem;                                             // [NP][256][KROW]
    unsigned char *sV = sK + NP * kT * KROW;                              // [NP][32][VROW]
    unsigned char *sW = sV + NP * HS * VROW;                              // [4][PKT]: q, k, v packets of the head, c_proj slice
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t b = HP ? blockIdx.x / n_head : blockIdx.x;
    const int hd_lo = HP ? (int)(blockIdx.x - b * n_head) : 0, hd_hi = HP ? hd_lo + 1 : n_head;     // heads of this workgroup
    const int tok0 = wave * 32;
    float *xt = x + (b * kT + tok0) * C + r * 8 + 4 * h;                 // chunk-major x (xt_off): chunk c = 4 j + gq at xt + c * 256
    const float *erow = EMBED ? wte + (size_t)tokens[b * kT + tok0 + r] * C : nullptr;   // token embedding row of this lane's token
    const float *prow = EMBED ? wpe + (size_t)(tok0 + r) * C : nullptr;                   // position embedding row
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(wpk);
    const unsigned char *psrc = reinterpret_cast<const unsigned char *>(ppk);
    static_assert(2 * CT * NP == F, "c_proj slice packet has the same size as a c_attn tile packet");
    const bool full = !LAST || wave == NW - 1;                            // wave-uniform: does this wave run q / attention / c_proj?

    // Two barriers per head:
    //   A(hd): q|k|v packets of head hd landed (issued right after B(hd-1)), K/V^T and the c_proj slot are free
    //          -> issue c_proj(hd); project q, k, v; publish k, v^T
    //   B(hd): k, v^T of the head complete, c_proj(hd) landed, q|k|v slots free -> issue q|k|v(hd+1); attention; c_proj
    auto dma = [&](const unsigned char *src, int slot_) {
        unsigned char *dst = sW + (size_t)slot_ * PKT;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            const int c = min(wave + NW * i, F - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)c * 1024 + lane * 16),
                                             (lds_void_t *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
    };
    auto issue_qkv = [&](int hd_) {                                       // hs == 32: one tile per (which, head)
#pragma unroll
        for (int which = 0; which < 3; which++) dma(wsrc + (size_t)(which * CT + hd_) * PKT, which);
    };
    issue_qkv(hd_lo);

    // ---- LayerNorm of this lane's token, operand planes in registers ----
    u32x4 xn[KS][2];
    {
        f32x16 xv[CT];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int off = 32 * j + 8 * gq + 4 * h;
                const f32x4 v = EMBED ? *reinterpret_cast<const f32x4 *>(erow + off) + *reinterpret_cast<const f32x4 *>(prow + off)
                                      : *reinterpret_cast<const f32x4 *>(xt + (4 * j + gq) * 256);
                xv[j][4 * gq] = v[0]; xv[j][4 * gq + 1] = v[1]; xv[j][4 * gq + 2] = v[2]; xv[j][4 * gq + 3] = v[3];
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) { const float d = xv[j][g] - mean; q += d * d; }
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q / (float)C + 1e-5f);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int j = ks >> 1, g0 = 8 * (ks & 1);
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gain + 32 * j + 8 * (g0 >> 2) + 4 * h);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(gain + 32 * j + 8 * (g0 >> 2) + 8 + 4 * h);
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v0[e] = (xv[j][g0 + e] - mean) * rstd * ga[e];
                v1[e] = (xv[j][g0 + 4 + e] - mean) * rstd * gb[e];
            }
            u32x2 h0, l0, h1, l1;
            split4<T, NP>(v0, h0, l0);
            split4<T, NP>(v1, h1, l1);
            xn[ks][0][0] = h0[0]; xn[ks][0][1] = h0[1]; xn[ks][0][2] = h1[0]; xn[ks][0][3] = h1[1];
            xn[ks][1][0] = l0[0]; xn[ks][1][1] = l0[1]; xn[ks][1][2] = l1[0]; xn[ks][1][3] = l1[1];
        }
    }

    // one projection tile from packet slot `slot_`; swapped: lane = token, registers = features (else the transpose).
    // The result stays in the weights' power-of-two scale (1/inv_scale); the scale is folded into the softmax
    // exponent (q, k) and into the final 1/l (v) instead of costing 16 multiplies per tile.
    auto project = [&](int slot_, bool swapped, f32x16 &out) {
        const unsigned char *pk = sW + (size_t)slot_ * PKT + lane * 16;
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 16; g++) { a0[g] = 0.f; a1[g] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            u32x4 w0[2], w1[2];
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                w0[pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)(ks * NP + pl) * 1024);
                w1[pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((ks + 1) * NP + pl) * 1024);
            }
            if (swapped) { a0 = mma<T, NP>(w0, xn[ks], a0); a1 = mma<T, NP>(w1, xn[ks + 1], a1); }
            else { a0 = mma<T, NP>(xn[ks], w0, a0); a1 = mma<T, NP>(xn[ks + 1], w1, a1); }
        }
#pragma unroll
        for (int g = 0; g < 16; g++) out[g] = a0[g] + a1[g];
    };
    // register octet m (registers 8m .. 8m+7) of a tile -> one 16-byte k-slot group per plane
    auto pack_octet = [&](const f32x16 &v, int m, u32x4 (&dst)[2]) {
        const float v0[4] = {v[8 * m], v[8 * m + 1], v[8 * m + 2], v[8 * m + 3]};
        const float v1[4] = {v[8 * m + 4], v[8 * m + 5], v[8 * m + 6], v[8 * m + 7]};
        u32x2 h0, l0, h1, l1;
        split4<T, NP>(v0, h0, l0);
        split4<T, NP>(v1, h1, l1);
        dst[0][0] = h0[0]; dst[0][1] = h0[1]; dst[0][2] = h1[0]; dst[0][3] = h1[1];
        dst[1][0] = l0[0]; dst[1][1] = l0[1]; dst[1][2] = l1[0]; dst[1][3] = l1[1];
    };
    auto sync_all = [&]() {                                                // this wave's DMA pieces landed + LDS writes visible, then barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 pacc[PROJ ? CT : 1];                                            // c_proj output accumulators (swapped layout)
#pragma unroll
    for (int j = 0; j < (PROJ ? CT : 1); j++)
#pragma unroll
        for (int g = 0; g < 16; g++) pacc[j][g] = 0.f;
    const float sc2 = scale_log2e * inv_scale * inv_scale;                // softmax exponent scale for q.k in weight-scaled units

    sync_all();                                                            // A(0)
#pragma unroll 1
    for (int hd = hd_lo; hd < hd_hi; hd++) {
        if (PROJ) dma(psrc + (size_t)hd * PKT, 3);                         // c_proj slice of this head (needed after the attention)
        f32x16 tile;
        u32x4 qf[2][2];                                                   // B operand of S^T = K Q^T: [k-step][plane]
        if (full) {
            project(0, true, tile);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) pack_octet(tile, ks, qf[ks]);
        }
        // ---- k -> sK[pl][key = tok0 + r][octet ks][half h] ----
        project(1, true, tile);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            u32x4 kp[2];
            pack_octet(tile, ks, kp);
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                *reinterpret_cast<u32x4 *>(sK + (size_t)pl * kT * KROW + (tok0 + r) * KROW + ks * 32 + h * 16) = kp[pl];
        }
        // ---- v (natural: lane = d, registers = tokens) -> sV[pl][d = r][(wave, octet mm)][half h] ----
        project(2, false, tile);
#pragma unroll
        for (int mm = 0; mm < 2; mm++) {
            u32x4 vp[2];
            pack_octet(tile, mm, vp);
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                *reinterpret_cast<u32x4 *>(sV + (size_t)pl * HS * VROW + r * VROW + (wave * 2 + mm) * 32 + h * 16) = vp[pl];
        }
        sync_all();                                                        // B(hd)
        if (hd + 1 < hd_hi) issue_qkv(hd + 1);                             // flies during the attention below

        // ---- attention of this wave's 32 queries against the 256 keys of the head ----
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 16; g++) o[g] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < (full ? kT / 32 : 0); kt++) {
            f32x16 sc;
#pragma unroll
            for (int g = 0; g < 16; g++) sc[g] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                u32x4 kf[2];
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    kf[pl] = *reinterpret_cast<const u32x4 *>(sK + (size_t)pl * kT * KROW + (kt * 32 + r) * KROW + ks * 32 + h * 16);
                sc = mma<T, NP>(kf, qf[ks], sc);
            }
            // sc[g] = S[query r][key 32 kt + tau(g, h)]  (times 1/inv_scale^2)
            float mx = sc[0];
#pragma unroll
            for (int g = 1; g < 16; g++) mx = fmaxf(mx, sc[g]);
            mx = half_max32(mx);
            if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {            // some query's running max moved: rescale (wave-uniform branch)
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
                l_run *= alpha;
#pragma unroll
                for (int g = 0; g < 16; g++) o[g] *= alpha;
                m_run = m_new;
            }
            const float nm = -m_run * sc2;
            float psum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                sc[g] = __builtin_amdgcn_exp2f(fmaf(sc[g], sc2, nm));
                psum += sc[g];
            }
            psum = half_sum32(psum);
            l_run += psum;
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {
                u32x4 pf[2], vf[2];
                pack_octet(sc, mm, pf);
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    vf[pl] = *reinterpret_cast<const u32x4 *>(sV + (size_t)pl * HS * VROW + r * VROW + (kt * 2 + mm) * 32 + h * 16);
                o = mma<T, NP>(vf, pf, o);
            }
        }
        const float inv = full ? inv_scale / l_run : 0.f;                  // 1/l and the v projection's weight scale
#pragma unroll
        for (int g = 0; g < 16; g++) o[g] *= inv;
        // o[g] = O[query r][d = tau(g, h)]
        if (!PROJ) {                                                      // -> y planes [b*256 + tok0 + r][hd*32 + d]
            const size_t yrow = ((size_t)b * kT + tok0 + r) * C + hd * HS;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const float v[4] = {o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]};
                u32x2 hi, lo;
                split4<T, NP>(v, hi, lo);
                *reinterpret_cast<u32x2 *>(y_hi + yrow + 8 * gq + 4 * h) = hi;
                if (NP == 2) *reinterpret_cast<u32x2 *>(y_lo + yrow + 8 * gq + 4 * h) = lo;
            }
        } else if (full) {
            // the head's output is, as it stands, the B operand of its slice of c_proj: pacc += Wp[:, head] y_head
            u32x4 yf[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; kk++) pack_octet(o, kk, yf[kk]);
            const unsigned char *pk = sW + (size_t)3 * PKT + lane * 16;
#pragma unroll
            for (int j = 0; j < CT; j++)
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    u32x4 wf[2];
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) wf[pl] = *reinterpret_cast<const u32x4 *>(pk + (size_t)((2 * j + kk) * NP + pl) * 1024);
                    pacc[j] = mma<T, NP>(wf, yf[kk], pacc[j]);
                }
        }
        sync_all();                                                        // A(hd+1)
    }
    if (PROJ && full) {
        // ---- residual add, store, LayerNorm statistics of the new row (as the GEMM / MLP epilogues) ----
        // LAST: only token 255 (lanes 31 and 63 of the last wave) is kept, in the compact buffer
        const bool keep = !LAST || r == 31;
        if constexpr (HP) {
            // this head's c_proj contribution, in true units, where the residual row would go (in the head's own partial buffer)
            float *prow = part_out + (size_t)hd_lo * part_stride +
                          (LAST ? (b >> 5) * 32 * C + (b & 31) * 8 + 4 * h : (b * kT + tok0) * C + r * 8 + 4 * h);
#pragma unroll
            for (int j = 0; j < CT; j++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = pacc[j][4 * gq + e] * inv_scale_p;
                    if (keep) *reinterpret_cast<f32x4 *>(prow + (4 * j + gq) * 256) = v;
                }
            return;
        }
        // (LAST: row b of the compact matrix, chunk-major as well: tile b / 32, token b % 32)
        float *orow = LAST ? x_last + (b >> 5) * 32 * C + (b & 31) * 8 + 4 * h : xt;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int off = 32 * j + 8 * gq + 4 * h;
                f32x4 cur = EMBED ? *reinterpret_cast<const f32x4 *>(erow + off) + *reinterpret_cast<const f32x4 *>(prow + off)
                                  : *reinterpret_cast<const f32x4 *>(xt + (4 * j + gq) * 256);
#pragma unroll
                for (int e = 0; e < 4; e++) { cur[e] += pacc[j][4 * gq + e] * inv_scale_p; pacc[j][4 * gq + e] = cur[e]; }
                if (keep) *reinterpret_cast<f32x4 *>(orow + (4 * j + gq) * 256) = cur;
                s2 += (cur[0] + cur[1]) + (cur[2] + cur[3]);
            }
        if (stats_out != nullptr && !LAST) {
            s2 += __shfl_xor(s2, 32);
            const float mean2 = s2 / (float)C;
            float q2 = 0.f;
#pragma unroll
            for (int j = 0; j < CT; j++)
#pragma unroll
                for (int g = 0; g < 16; g++) { const float d = pacc[j][g] - mean2; q2 += d * d; }
            q2 += __shfl_xor(q2, 32);
            if (h == 0) stats_out[b * kT + tok0 + r] = make_float2(mean2, rsqrtf(q2 / (float)C + 1e-5f));
        }
    }
}

}  // namespace fastk
}  // namespace mgpt
