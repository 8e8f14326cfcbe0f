// gpt_fast.hip -- host side of the 16-bit-MFMA forward (MGPT_PREC_F16X3 / MGPT_PREC_BF16):
// weight plane packing at finalize, workspace, and the per-chunk launch sequence
//   embed+stats -> L x { LN1+QK gemm, LN1+V^T gemm, attention, proj+residual(+stats),
//                        LN2+FC+GELU gemm, proj2+residual(+stats) } -> ln_f + head (fp32 kernel)
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gpt_ctx.h"
#include "gpt_kernels_fast.h"
#include "gpt_kernels_c256.h"
#include "gpt_kernels_c256q.h"      // (includes gpt_kernels_c256p.h)
#include "gpt_kernels_fused16.h"
#include "gpt_kernels_c256a.h"
#include "gpt_kernels_c256b.h"      // attn256q_kernel
#include "gpt_kernels_c160p.h"
#include "gpt_kernels_c160a.h"
#include "gpt_kernels_last.h"

using namespace mgpt;

namespace {

constexpr int kT = 256;
constexpr int kV = MGPT_VOCAB;
// A/B builds (tools/ab_lib.sh): -DMGPT_AB_ATTN_UNFUSED keeps round 3's attn256_kernel + packed-GEMM out-projection
#ifdef MGPT_AB_ATTN_UNFUSED
constexpr bool kAttn256Fused = false;
#else
constexpr bool kAttn256Fused = true;
#endif
// -DMGPT_AB_NO_LAST1 keeps the last layer on the full attention kernels (attn256_kernel<LAST> / attn_block_kernel<LAST>)
// -DMGPT_AB_NO_ATTN160O keeps the 2M shape's middle layers on attn_block_kernel
#ifdef MGPT_AB_NO_ATTN160O
constexpr bool kAttn160o = false;
#else
constexpr bool kAttn160o = true;
#endif
// -DMGPT_AB_NO_LN_FOLD keeps ln_pack_kernel in the one-plane packed-GEMM chain
#ifdef MGPT_AB_NO_LN_FOLD
constexpr bool kLnFold = false;
#else
constexpr bool kLnFold = true;
#endif
// -DMGPT_AB_NO_MLP160_HALF keeps 128-token blocks in mlp160p_kernel for every small launch
// (-DMGPT_AB_NO_MLP160_QUARTER: no 32-token blocks)
#ifdef MGPT_AB_NO_MLP160_QUARTER
constexpr bool kMlp160Quarter = false;
#else
constexpr bool kMlp160Quarter = true;
#endif
#ifdef MGPT_AB_NO_MLP160_HALF
constexpr bool kMlp160Half = false;
#else
constexpr bool kMlp160Half = true;
#endif
// -DMGPT_AB_NO_SMALL256 keeps small launches of the 6M shape on the row-per-workgroup kernels
#ifdef MGPT_AB_NO_SMALL256
constexpr bool kSmall256 = false;
#else
constexpr bool kSmall256 = true;
#endif
#ifdef MGPT_AB_NO_LAST1_TAIL
constexpr bool kLast1Tail = false;
#else
constexpr bool kLast1Tail = true;
#endif
#ifdef MGPT_AB_NO_LAST1
constexpr bool kLast1 = false;
#else
constexpr bool kLast1 = true;
#endif

// kSmallRows (gpt_ctx.h): rows up to which the C = 64 / 160 attention block runs head-parallel: beyond ~164 rows of 5 heads the (row, head)
// workgroups need more rounds on 256 CUs than one workgroup per row takes (25 us against 80 us per workgroup, profiles/r02_cfg1_step_trace.txt)

struct PlaneSet {           // one weight matrix [N][K] as 16-bit planes
    uint16_t *hi = nullptr, *lo = nullptr;
    float inv_scale = 1.f;
};

struct ModeState {          // one precision mode
    bool built = false;
    int np = 0;             // planes per operand
    int n_cu = 256;         // compute units of the model's device (grid of the persistent kernels)
    std::vector<PlaneSet> attn, proj, fc, proj2;
    // workspace (sized for max_rows)
    float2 *stats = nullptr;
    uint16_t *qk[2] = {nullptr, nullptr};      // [2 (q,k)][M][C] per plane
    uint16_t *vt[2] = {nullptr, nullptr};      // [M][C]
    uint16_t *y[2] = {nullptr, nullptr};       // [M][C]
    uint16_t *hbuf[2] = {nullptr, nullptr};    // [M][4C]
    // fused MLP (C = 64 / 160): per layer one packed stream [hidden tile][fragment][plane][lane][8]
    std::vector<uint16_t *> mlp_pk;
    std::vector<uint16_t *> mlp16_pk;           // the same packets in the fragment layout of mlp_fused16_kernel (16 x 16 x 32 MFMA: large calls)
    bool mlp_fused = false;
    // C = 256 (6M): mlp256p_kernel's cyclic weight stream in consumption order (LayerNorm gain folded into c_fc), per layer
    // [period step][pair][plane][lane][8], and the scale c_fc * gain was packed with
    std::vector<uint16_t *> mlp256_pk;
    std::vector<uint16_t *> mlp256q_pk;         // the same stream in the fragment layout of mlp256q_kernel (16 x 16 x 32 MFMA: large calls)
    std::vector<float> mlp256_inv1;
    std::vector<float> attn256_inv;            // 1 / scale of the attn256 weight stream (c_attn.weight * ln_1.weight)
    float2 *gelu_lut = nullptr;                // the Phi table of the fused MLP kernels (kGeluLutN pairs)
    std::vector<float2 *> mlp256_lut;          // per layer: the same table times 1 / scale of the layer's c_fc stream (mlp256p_kernel)
    // C = 256, head size 32 (6M): attn256_kernel's c_attn stream in consumption order, per layer
    std::vector<uint16_t *> attn256_pk;
    bool attn256 = false;
    // attn256o_kernel (whole attention block, persistent): c_attn + c_proj stream per layer, and the per-workgroup spill slab
    std::vector<uint16_t *> attn256o_pk;
    std::vector<uint16_t *> attn256q_pk;   // the same stream in attn256q_kernel's fragment order (gpt_kernels_c256b.h)
    unsigned char *attn256o_spill = nullptr;
    // register-resident LN+QKV (C = 64 / 160): per layer [tile][k-step][plane][lane][8] of c_attn.weight
    std::vector<uint16_t *> qkv_pk;
    bool qkv_fused = false;
    // out-projection slices per head for attn_block_kernel<PROJ>: [head][out tile][kk][plane][lane][8]
    std::vector<uint16_t *> proj_pk;
    // last-layer shortcut: new residual rows of token 255 only, [round_up(max_rows, 256)][C] fp32
    float *x_last = nullptr;
    float *x_head = nullptr;                   // x_tiled: the last-token rows in plain row-major order for the head kernel
    bool x_tiled = false;                      // residual stream chunk-major (fastk::xt_off): the C = 256 kernels
    uint16_t *y_last = nullptr;                // packed-GEMM path: attention output of token 255 of every row, PK planes [rows_pad][C]
    // small launches of the register-resident path (rows <= kSmallRows: one environment, BASELINE cfg1): attn_block_kernel<HP> runs one
    // workgroup per (row, head) and leaves the heads' c_proj contributions here, [n_head][kSmallRows * 256 * C] fp32 in x's layout
    float *head_parts = nullptr;
    std::vector<uint16_t *> attn160o_pk;        // C = 160: attn160o_kernel's stream per layer (c_attn * ln_1 | c_proj), its 1 / scale, spill slab
    std::vector<float> attn160_inv;
    unsigned char *attn160o_spill = nullptr;
    float *last1_wt = nullptr;                 // last layer, attn_last1_kernel: transposes of W_q, W_v, c_proj.weight (fp32)
    // ... and the MLP block of those launches (C = 160): mlp160p_kernel's cyclic stream per layer and the scale c_fc * ln_2 was packed with
    std::vector<uint16_t *> mlp160_pk;
    std::vector<float> mlp160_inv1;
    // PK GEMM path (C % 256 == 0: 6M, 85M): weights as MFMA-fragment streams, activations produced in the same layout
    bool pk_gemm = false;
    std::vector<uint16_t *> attn_pk2, proj_pk2, fc_pk2, proj2_pk2;   // [row tile][k-step][plane][lane][8]
    uint16_t *apk = nullptr;                   // LayerNorm'ed rows in PK layout, [M/32][C/16][NP][512]
    // folded LayerNorm (GemmArgs in gpt_kernels_fast.h; one-plane mode of the packed-GEMM chain): apk holds the RAW rows, written by the
    // residual epilogues; W * ln.weight packings, their column sums, the epilogues' partial row sums
    bool ln_fold = false;
    std::vector<uint16_t *> attn_pk2g, fc_pk2g;
    std::vector<float *> attn_cs, fc_cs;
    float2 *ln_parts = nullptr;                // [C / 128][M]
    float *ln_mean = nullptr;                  // [M]: the mean of every row at its latest LayerNorm = the shift of its operand planes (GemmArgs::shift)
};

struct FastState {
    ModeState mode[3];      // indexed by MGPT_PREC_* (slot 0 unused)
};

template <class T, int NP>
int pack_matrix(const float *d_w, size_t n_elem, float scale, PlaneSet *out, hipStream_t s)
{
    MGPT_HIP(hipMalloc(&out->hi, n_elem * sizeof(uint16_t)));
    if (NP == 2) MGPT_HIP(hipMalloc(&out->lo, n_elem * sizeof(uint16_t)));
    out->inv_scale = 1.0f / scale;
    const int64_t n4 = (int64_t)(n_elem / 4);
    ProfScope ps(P_PACK, s);
    hipLaunchKernelGGL((fastk::pack_planes_kernel<T, NP>), dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, s, d_w, out->hi,
                       out->lo, n4, scale);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

float pick_scale(const float *h_w, size_t n, bool f16)
{
    if (!f16) return 1.0f;                       // bf16 has the fp32 exponent range
    float mx = 0.f;
    for (size_t i = 0; i < n; i++) mx = std::max(mx, fabsf(h_w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e = (int)floorf(log2f(4096.0f / mx));    // largest |w| lands in [2048, 4096]: hi AND lo parts stay normal fp16
    e = std::max(-8, std::min(e, 20));
    return ldexpf(1.0f, e);
}

template <int NP>
constexpr int kA256Lds = 5 * 8 * NP * 1024 + NP * (256 * 80 + 32 * 528);    // attn256_kernel: 5-slot weight ring + K and V^T planes of a head

template <class T, int NP>
int build_mode(mgpt_gpt *g, ModeState *m, bool f16)
{
    const size_t C = g->C;
    g->generation++;                                     // a step graph captured before this build must not be replayed as is
    std::vector<float> host(g->n_params);
    MGPT_HIP(hipMemcpy(host.data(), g->params, g->n_params * sizeof(float), hipMemcpyDeviceToHost));
    m->np = NP;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        MGPT_HIP(hipGetDevice(&dev));
        MGPT_HIP(hipGetDeviceProperties(&prop, dev));
        m->n_cu = std::max(1, prop.multiProcessorCount);
    }
    m->attn.resize(g->L); m->proj.resize(g->L); m->fc.resize(g->L); m->proj2.resize(g->L);
    int rc;
    for (int l = 0; l < g->L; l++) {
        const LayerOff &lo = g->layers[l];
        struct { size_t off, n; PlaneSet *dst; } mats[4] = {{lo.attn_w, 3 * C * C, &m->attn[l]}, {lo.proj_w, C * C, &m->proj[l]},
                                                           {lo.fc_w, 4 * C * C, &m->fc[l]}, {lo.proj2_w, 4 * C * C, &m->proj2[l]}};
        for (auto &mt : mats) {
            const float sc = pick_scale(host.data() + mt.off, mt.n, f16);
            if ((rc = pack_matrix<T, NP>(g->params + mt.off, mt.n, sc, mt.dst, nullptr)) != MGPT_OK) return rc;
        }
    }
    // (C = 256: the fused kernels are the 6M shape's -- 8 heads of 32 -- and share the chunk-major residual stream; any other
    //  C = 256 model takes the packed-fragment GEMM chain)
    m->mlp_fused = (C == 160 || C == 64 || (C == 256 && g->hs == 32 && g->nh == 8));
    {   // Phi(v) = (1 + erf(v / sqrt 2)) / 2 on [-6, 6) in steps of 1/256, as (value, forward difference) pairs
        std::vector<float2> lut(fastk::kGeluLutN);
        auto phi = [](double v) { return 0.5 * (1.0 + erf(v * 0.70710678118654752440)); };
        for (int i = 0; i < fastk::kGeluLutN; i++) {
            const double v0 = (i - (double)fastk::kGeluLutBias) / fastk::kGeluLutScale, v1 = (i + 1 - (double)fastk::kGeluLutBias) / fastk::kGeluLutScale;
            const float f0 = (float)phi(v0);
            lut[i] = make_float2(f0, (float)(phi(v1) - (double)f0));
        }
        MGPT_HIP(hipMalloc(&m->gelu_lut, lut.size() * sizeof(float2)));
        MGPT_HIP(hipMemcpy(m->gelu_lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (C == 256 && m->mlp_fused) {
        const size_t n16 = (size_t)fastk::kMPPeriod * 16 * NP * 512;
        m->mlp256_pk.assign(g->L, nullptr);
        m->mlp256q_pk.assign(g->L, nullptr);
        m->mlp256_inv1.assign(g->L, 1.f);
        m->mlp256_lut.assign(g->L, nullptr);
        for (int l = 0; l < g->L; l++) {
            MGPT_HIP(hipMalloc(&m->mlp256_pk[l], n16 * sizeof(uint16_t)));
            MGPT_HIP(hipMalloc(&m->mlp256q_pk[l], n16 * sizeof(uint16_t)));
            const LayerOff &lo = g->layers[l];
            // the stream carries c_fc.weight * ln_2.weight (model.py:19-20, 86): its own power-of-two scale
            std::vector<float> wg(4 * C * C);
            for (size_t i = 0; i < wg.size(); i++) wg[i] = host[lo.fc_w + i] * host[lo.ln2 + i % C];
            const float sc1 = pick_scale(wg.data(), wg.size(), f16);
            m->mlp256_inv1[l] = 1.0f / sc1;
            {   // Phi table pre-multiplied by the (power-of-two) 1 / scale: GELU = pre-activation in stream units x entry, the same bits
                std::vector<float2> lt(fastk::kGeluLutN);
                MGPT_HIP(hipMemcpy(lt.data(), m->gelu_lut, lt.size() * sizeof(float2), hipMemcpyDeviceToHost));
                for (auto &e : lt) { e.x *= m->mlp256_inv1[l]; e.y *= m->mlp256_inv1[l]; }
                MGPT_HIP(hipMalloc(&m->mlp256_lut[l], lt.size() * sizeof(float2)));
                MGPT_HIP(hipMemcpy(m->mlp256_lut[l], lt.data(), lt.size() * sizeof(float2), hipMemcpyHostToDevice));
            }
            ProfScope ps(P_PACK, nullptr);
            hipLaunchKernelGGL((fastk::pack_mlp256p_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kMPPeriod * 16 * 64, 256)), dim3(256), 0,
                               nullptr, g->params + lo.fc_w, g->params + lo.proj2_w, g->params + lo.ln2, m->mlp256_pk[l], sc1,
                               1.0f / m->proj2[l].inv_scale);
            MGPT_LAUNCH_CHECK();
            hipLaunchKernelGGL((fastk::pack_mlp256q_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kMPPeriod * 16 * 64, 256)), dim3(256), 0,
                               nullptr, g->params + lo.fc_w, g->params + lo.proj2_w, g->params + lo.ln2, m->mlp256q_pk[l], sc1,
                               1.0f / m->proj2[l].inv_scale);
            MGPT_LAUNCH_CHECK();
        }
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp256p_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     fastk::kMPLds<NP>));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp256p_kernel<T, NP, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     fastk::kMPLds<NP>));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp256q_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     fastk::kMPLds<NP>));
        m->attn256 = (g->hs == 32 && g->nh == 8);
        if (m->attn256) {
            const size_t n16 = (size_t)8 * fastk::kA256StepsPerHead * 8 * NP * 512;
            m->attn256_pk.assign(g->L, nullptr);
            m->attn256_inv.assign(g->L, 1.f);
            for (int l = 0; l < g->L; l++) {
                MGPT_HIP(hipMalloc(&m->attn256_pk[l], n16 * sizeof(uint16_t)));
                const LayerOff &lo = g->layers[l];
                // the stream carries c_attn.weight * ln_1.weight (model.py:19-20, 50): its own power-of-two scale
                std::vector<float> wg(3 * C * C);
                for (size_t i = 0; i < wg.size(); i++) wg[i] = host[lo.attn_w + i] * host[lo.ln1 + i % C];
                const float sc = pick_scale(wg.data(), wg.size(), f16);
                m->attn256_inv[l] = 1.0f / sc;
                ProfScope ps(P_PACK, nullptr);
                hipLaunchKernelGGL((fastk::pack_attn256_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)8 * fastk::kA256StepsPerHead * 8 * 64, 256)), dim3(256), 0,
                                   nullptr, g->params + lo.attn_w, g->params + lo.ln1, m->attn256_pk[l], sc);
                MGPT_LAUNCH_CHECK();
            }
            {   // the whole attention block in one persistent kernel: 48 steps of c_attn * ln_1 followed by 16 steps of c_proj
                const size_t n16o = (size_t)fastk::kA256oPeriod * 8 * NP * 512;
                m->attn256o_pk.assign(g->L, nullptr);
                m->attn256q_pk.assign(g->L, nullptr);
                for (int l = 0; l < g->L; l++) {
                    MGPT_HIP(hipMalloc(&m->attn256o_pk[l], n16o * sizeof(uint16_t)));
                    const LayerOff &lo = g->layers[l];
                    ProfScope ps(P_PACK, nullptr);
                    hipLaunchKernelGGL((fastk::pack_attn256o_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kA256oPeriod * 8 * 64, 256)), dim3(256), 0,
                                       nullptr, g->params + lo.attn_w, g->params + lo.ln1, g->params + lo.proj_w, m->attn256o_pk[l],
                                       1.0f / m->attn256_inv[l], 1.0f / m->proj[l].inv_scale);
                    MGPT_LAUNCH_CHECK();
                    MGPT_HIP(hipMalloc(&m->attn256q_pk[l], n16o * sizeof(uint16_t)));
                    hipLaunchKernelGGL((fastk::pack_attn256q_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kA256oPeriod * 8 * 64, 256)), dim3(256), 0,
                                       nullptr, g->params + lo.attn_w, g->params + lo.ln1, g->params + lo.proj_w, m->attn256q_pk[l],
                                       1.0f / m->attn256_inv[l], 1.0f / m->proj[l].inv_scale);
                    MGPT_LAUNCH_CHECK();
                }
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256q_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256q_kernel<T, NP, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
                if (g->embed_table == nullptr) {       // (position, token) rows for layer 0 (attn256q_kernel<.., EMB>); one per checkpoint, shared by the modes
                    MGPT_HIP(hipMalloc(&g->embed_table, (size_t)kT * kV * C * sizeof(float)));
                    hipLaunchKernelGGL(fastk::embed_table_kernel, dim3((unsigned)cdiv64((int64_t)kT * kV * (C / 4), 256)), dim3(256), 0, nullptr,
                                       g->params + g->off_wte, g->params + g->off_wpe, g->embed_table, C, kV);
                    MGPT_LAUNCH_CHECK();
                }
                MGPT_HIP(hipMalloc(&m->attn256o_spill, (size_t)m->n_cu * 8 * 14 * NP * 1024));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256o_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
            }
            MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256_kernel<T, NP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
            MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256_kernel<T, NP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
            MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn256_kernel<T, NP, false, 0, fastk::kA256Stagger, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kA256Lds<NP>));
        }
    } else if (m->mlp_fused) {
        const size_t frags = C / 16 + 2 * (C / 32), nt = 4 * C / 32;
        const size_t n16 = nt * frags * NP * 512;
        m->mlp_pk.assign(g->L, nullptr);
        m->mlp16_pk.assign(g->L, nullptr);
        for (int l = 0; l < g->L; l++) {
            MGPT_HIP(hipMalloc(&m->mlp_pk[l], n16 * sizeof(uint16_t)));
            MGPT_HIP(hipMalloc(&m->mlp16_pk[l], n16 * sizeof(uint16_t)));
            const LayerOff &lo = g->layers[l];
            ProfScope ps(P_PACK, nullptr);
            hipLaunchKernelGGL((fastk::pack_mlp_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(nt * frags * 64), 256)), dim3(256), 0,
                               nullptr, g->params + lo.fc_w, g->params + lo.proj2_w, m->mlp_pk[l], (int)C,
                               1.0f / m->fc[l].inv_scale, 1.0f / m->proj2[l].inv_scale);
            MGPT_LAUNCH_CHECK();
            hipLaunchKernelGGL((fastk::pack_mlp16_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(nt * frags * 64), 256)), dim3(256), 0,
                               nullptr, g->params + lo.fc_w, g->params + lo.proj2_w, m->mlp16_pk[l], (int)C,
                               1.0f / m->fc[l].inv_scale, 1.0f / m->proj2[l].inv_scale);
            MGPT_LAUNCH_CHECK();
        }
        const int pkt = (int)(frags * NP * 1024 * 3) + fastk::kGeluLutN * 8;
#define MGPT_MLP_ATTR(CT_, NW_) \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp_fused_kernel<T, NP, CT_, NW_, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, pkt)); \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp_fused_kernel<T, NP, CT_, NW_, CT_>), hipFuncAttributeMaxDynamicSharedMemorySize, pkt)); \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp_fused16_kernel<T, NP, CT_, NW_, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, pkt))
        if (C == 160) { MGPT_MLP_ATTR(5, 8); MGPT_MLP_ATTR(5, 4); MGPT_MLP_ATTR(5, 2); }
        else { MGPT_MLP_ATTR(2, 8); MGPT_MLP_ATTR(2, 4); MGPT_MLP_ATTR(2, 2); }
#undef MGPT_MLP_ATTR
    }
    m->qkv_fused = (C == 160 || C == 64);
    if (m->qkv_fused) {
        const size_t ks = C / 16, ntile = 3 * C / 32;
        m->qkv_pk.assign(g->L, nullptr);
        for (int l = 0; l < g->L; l++) {
            MGPT_HIP(hipMalloc(&m->qkv_pk[l], ntile * ks * NP * 512 * sizeof(uint16_t)));
            ProfScope ps(P_PACK, nullptr);
            hipLaunchKernelGGL((fastk::pack_rows_perm_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(ntile * ks * 64), 256)), dim3(256), 0,
                               nullptr, g->params + g->layers[l].attn_w, m->qkv_pk[l], (int)ntile, (int)C, 1.0f / m->attn[l].inv_scale);
            MGPT_LAUNCH_CHECK();
        }
        if (g->hs == 32) {
            const size_t ct = C / 32;
            m->proj_pk.assign(g->L, nullptr);
            for (int l = 0; l < g->L; l++) {
                MGPT_HIP(hipMalloc(&m->proj_pk[l], (size_t)g->nh * 2 * ct * NP * 512 * sizeof(uint16_t)));
                ProfScope ps(P_PACK, nullptr);
                hipLaunchKernelGGL((fastk::pack_cols_perm_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(g->nh * 2 * ct * 64), 256)), dim3(256), 0,
                                   nullptr, g->params + g->layers[l].proj_w, m->proj_pk[l], g->nh, (int)C, (int)C, 1.0f / m->proj[l].inv_scale);
                MGPT_LAUNCH_CHECK();
            }
            // attn_block_kernel's dynamic LDS limit is a per-device function attribute: set it for this model's device
            const int lds = (int)((size_t)NP * (kT * 80 + 32 * 528) + (size_t)(C / 16) * NP * 1024 * 4);
#define MGPT_ATTN_LDS(CT_, LAST_, EMB_, HP_)                                                                                      \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn_block_kernel<T, NP, CT_, true, LAST_, EMB_, HP_>), \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds))
            if (C == 160) { MGPT_ATTN_LDS(5, false, false, false); MGPT_ATTN_LDS(5, true, false, false); MGPT_ATTN_LDS(5, false, true, false);
                            MGPT_ATTN_LDS(5, false, false, true); MGPT_ATTN_LDS(5, true, false, true); }
            else { MGPT_ATTN_LDS(2, false, false, false); MGPT_ATTN_LDS(2, true, false, false); MGPT_ATTN_LDS(2, false, true, false);
                   MGPT_ATTN_LDS(2, false, false, true); MGPT_ATTN_LDS(2, true, false, true); }
#undef MGPT_ATTN_LDS
            // (head_parts -- the heads' partial sums of small launches -- is allocated by the first small launch: forward_chunk)
            if (m->mlp_fused && C == 160) {
                const size_t n16 = (size_t)fastk::kM5Period * 20 * NP * 512;
                m->mlp160_pk.assign(g->L, nullptr);
                m->mlp160_inv1.assign(g->L, 1.f);
                for (int l = 0; l < g->L; l++) {
                    MGPT_HIP(hipMalloc(&m->mlp160_pk[l], n16 * sizeof(uint16_t)));
                    const LayerOff &lo = g->layers[l];
                    std::vector<float> wg(4 * C * C);                       // the stream carries c_fc.weight * ln_2.weight: its own power-of-two scale
                    for (size_t i = 0; i < wg.size(); i++) wg[i] = host[lo.fc_w + i] * host[lo.ln2 + i % C];
                    const float sc1 = pick_scale(wg.data(), wg.size(), f16);
                    m->mlp160_inv1[l] = 1.0f / sc1;
                    ProfScope ps(P_PACK, nullptr);
                    hipLaunchKernelGGL((fastk::pack_mlp160p_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kM5Period * 20 * 64, 256)), dim3(256), 0,
                                       nullptr, g->params + lo.fc_w, g->params + lo.proj2_w, g->params + lo.ln2, m->mlp160_pk[l], sc1,
                                       1.0f / m->proj2[l].inv_scale);
                    MGPT_LAUNCH_CHECK();
                }
                // the attention block as one persistent kernel (layers that neither gather the embedding nor are the last)
                m->attn160o_pk.assign(g->L, nullptr);
                m->attn160_inv.assign(g->L, 1.f);
                for (int l = 0; l < g->L; l++) {
                    MGPT_HIP(hipMalloc(&m->attn160o_pk[l], (size_t)fastk::kA160oPeriod * fastk::kA160oFrags * NP * 512 * sizeof(uint16_t)));
                    const LayerOff &lo = g->layers[l];
                    std::vector<float> wg(3 * C * C);                       // c_attn.weight * ln_1.weight: its own power-of-two scale
                    for (size_t i = 0; i < wg.size(); i++) wg[i] = host[lo.attn_w + i] * host[lo.ln1 + i % C];
                    const float sca = pick_scale(wg.data(), wg.size(), f16);
                    m->attn160_inv[l] = 1.0f / sca;
                    ProfScope ps(P_PACK, nullptr);
                    hipLaunchKernelGGL((fastk::pack_attn160o_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)fastk::kA160oPeriod * fastk::kA160oFrags * 64, 256)),
                                       dim3(256), 0, nullptr, g->params + lo.attn_w, g->params + lo.ln1, g->params + lo.proj_w, m->attn160o_pk[l], sca,
                                       1.0f / m->proj[l].inv_scale);
                    MGPT_LAUNCH_CHECK();
                }
                MGPT_HIP(hipMalloc(&m->attn160o_spill, (size_t)m->n_cu * fastk::kA160oSpillPerWg<NP>));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn160o_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             fastk::kA160oLds<NP>));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn160o_kernel<T, NP, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             fastk::kA160oLds<NP>));
                if (g->embed_table == nullptr) {       // (position, token) rows for layer 0 (attn160o_kernel<.., EMB>); one per checkpoint, shared by the modes
                    MGPT_HIP(hipMalloc(&g->embed_table, (size_t)kT * kV * C * sizeof(float)));
                    hipLaunchKernelGGL(fastk::embed_table_kernel, dim3((unsigned)cdiv64((int64_t)kT * kV * (C / 4), 256)), dim3(256), 0, nullptr,
                                       g->params + g->off_wte, g->params + g->off_wpe, g->embed_table, C, kV);
                    MGPT_LAUNCH_CHECK();
                }
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp160p_kernel<T, NP, 5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             fastk::kM5Lds<NP>));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp160p_kernel<T, NP, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             fastk::kM5Lds<NP>));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp160p_kernel<T, NP, 5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (fastk::kM5Lds<NP, 2>)));
                MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::mlp160p_kernel<T, NP, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (fastk::kM5Lds<NP, 1>)));
            }
        }
    }
    if (g->hs == 32 && (C == 256 || C == 160 || C == 64) && m->mlp_fused) {
        // the last layer's attention block for token 255 alone (attn_last1_kernel): fp32 transposes of W_q, W_v and c_proj.weight
        const LayerOff &lo = g->layers[g->L - 1];
        // (also c_fc.weight and mlp.c_proj.weight transposed, for the one-launch last layer + head of small launches; C = 256 since round 5)
        const bool tail = true;
        MGPT_HIP(hipMalloc(&m->last1_wt, (size_t)(tail ? 11 : 3) * C * C * sizeof(float)));
        struct { size_t off; int rows, cols; size_t dst; } mats[5] = {
            {lo.attn_w, (int)C, (int)C, 0}, {lo.attn_w + (size_t)2 * C * C, (int)C, (int)C, (size_t)C * C}, {lo.proj_w, (int)C, (int)C, (size_t)2 * C * C},
            {lo.fc_w, (int)(4 * C), (int)C, (size_t)3 * C * C}, {lo.proj2_w, (int)C, (int)(4 * C), (size_t)7 * C * C}};
        ProfScope ps(P_PACK, nullptr);
        for (int i = 0; i < (tail ? 5 : 3); i++) {
            hipLaunchKernelGGL(fastk::transpose_kernel, dim3((unsigned)cdiv64((int64_t)mats[i].rows * mats[i].cols, 256)), dim3(256), 0, nullptr,
                               g->params + mats[i].off, m->last1_wt + mats[i].dst, mats[i].rows, mats[i].cols);
            MGPT_LAUNCH_CHECK();
        }
#define MGPT_LAST1_LDS(C_, R_, TAIL_)                                                                                                          \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn_last1_kernel<C_, 32, R_, TAIL_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 fastk::kLast1Lds<C_, R_, TAIL_>))
        if (C == 256) { MGPT_LAST1_LDS(256, fastk::kLast1R, false); MGPT_LAST1_LDS(256, 1, false); MGPT_LAST1_LDS(256, 1, true); }
        else if (C == 160) { MGPT_LAST1_LDS(160, fastk::kLast1R, false); MGPT_LAST1_LDS(160, 1, true); }
        else { MGPT_LAST1_LDS(64, fastk::kLast1R, false); MGPT_LAST1_LDS(64, 1, true); }
#undef MGPT_LAST1_LDS
    }
    m->pk_gemm = (C == 256 || C == 512 || C == 768 || C == 1024);
    if (m->pk_gemm) {
        auto pack = [&](std::vector<uint16_t *> &dst, size_t off, size_t R, size_t K, float scale, int l) -> int {
            MGPT_HIP(hipMalloc(&dst[l], R * K * NP * sizeof(uint16_t)));
            ProfScope ps(P_PACK, nullptr);
            hipLaunchKernelGGL((fastk::pack_pk_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(R / 32) * (K / 16) * 64, 256)), dim3(256), 0,
                               nullptr, g->params + off, dst[l], (int)R, (int)K, scale);
            MGPT_LAUNCH_CHECK();
            return MGPT_OK;
        };
        m->attn_pk2.assign(g->L, nullptr); m->proj_pk2.assign(g->L, nullptr); m->fc_pk2.assign(g->L, nullptr); m->proj2_pk2.assign(g->L, nullptr);
        for (int l = 0; l < g->L; l++) {
            const LayerOff &lo = g->layers[l];
            if ((rc = pack(m->attn_pk2, lo.attn_w, 3 * C, C, 1.0f / m->attn[l].inv_scale, l)) != MGPT_OK) return rc;
            if ((rc = pack(m->proj_pk2, lo.proj_w, C, C, 1.0f / m->proj[l].inv_scale, l)) != MGPT_OK) return rc;
            if (!m->mlp_fused) {
                if ((rc = pack(m->fc_pk2, lo.fc_w, 4 * C, C, 1.0f / m->fc[l].inv_scale, l)) != MGPT_OK) return rc;
                if ((rc = pack(m->proj2_pk2, lo.proj2_w, C, 4 * C, 1.0f / m->proj2[l].inv_scale, l)) != MGPT_OK) return rc;
            }
        }
        const int lds = fastk::gemm_pk_lds(NP);
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_QK, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_VT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_RESID, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_RESID, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     fastk::gemm_pk_lds(NP, 4, fastk::EPI_RESID)));
        // the 128-row forms of the other epilogues (small calls of the C = 768 chain)
#define MGPT_PK4(EPI_, LNF_, EXTRA_)                                                                                                                  \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_, 4, 0, LNF_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 fastk::gemm_pk_lds(NP, 4, fastk::EPI_) + (EXTRA_)))
        MGPT_PK4(EPI_QK, false, 0); MGPT_PK4(EPI_VT, false, 0); MGPT_PK4(EPI_GELU, false, fastk::kGeluLutN * 8);
        MGPT_PK4(EPI_QK, true, 3072); MGPT_PK4(EPI_VT, true, 3072); MGPT_PK4(EPI_GELU, true, fastk::kGeluLutN * 8 + 3072);
#undef MGPT_PK4
        if constexpr (NP == 1) {
#define MGPT_PK16(EPI_, NWV_, LNF_, EXTRA_)                                                                                                            \
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk16_kernel<T, fastk::EPI_, NWV_, LNF_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 fastk::gemm_pk_lds(NP, NWV_, fastk::EPI_) + (EXTRA_)))
            MGPT_PK16(EPI_QK, 8, false, 0); MGPT_PK16(EPI_VT, 8, false, 0); MGPT_PK16(EPI_RESID, 8, false, 0); MGPT_PK16(EPI_GELU, 8, false, fastk::kGeluLutN * 8);
            MGPT_PK16(EPI_QK, 4, false, 0); MGPT_PK16(EPI_VT, 4, false, 0); MGPT_PK16(EPI_RESID, 4, false, 0); MGPT_PK16(EPI_GELU, 4, false, fastk::kGeluLutN * 8);
            MGPT_PK16(EPI_QK, 8, true, 3072); MGPT_PK16(EPI_VT, 8, true, 3072); MGPT_PK16(EPI_GELU, 8, true, fastk::kGeluLutN * 8 + 3072);
            MGPT_PK16(EPI_QK, 4, true, 3072); MGPT_PK16(EPI_VT, 4, true, 3072); MGPT_PK16(EPI_GELU, 4, true, fastk::kGeluLutN * 8 + 3072);
#undef MGPT_PK16
        }
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_GELU, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     lds + fastk::kGeluLutN * 8));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_QK, 8, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds + 3072));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_VT, 8, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds + 3072));
        MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::gemm_pk_kernel<T, NP, fastk::EPI_GELU, 8, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     lds + fastk::kGeluLutN * 8 + 3072));
        if (!(C == 256 && m->mlp_fused && g->hs == 32 && g->nh == 8))      // LayerNorm planes of the GEMM chain (the 6M kernels normalise in registers)
            MGPT_HIP(hipMalloc(&m->apk, (size_t)g->max_rows * kT * C * NP * sizeof(uint16_t)));
        // one-plane mode of the full chain (no fused kernels): LayerNorm folded into the GEMMs (GemmArgs) -- ln_pack_kernel, which
        // re-reads every residual row to normalise it (8 % of an 85M step), runs once per forward instead of twice per layer
        // (MGPT_LN_FOLD=0 in the environment keeps the normalised planes: the raw planes round x itself to bf16, so a residual stream whose
        //  per-token mean is many times its standard deviation loses precision that LayerNorm-then-round keeps)
        const char *fold_env = getenv("MGPT_LN_FOLD");
        m->ln_fold = kLnFold && NP == 1 && !m->mlp_fused && m->apk != nullptr && !(fold_env != nullptr && fold_env[0] == '0');
        if (m->ln_fold) {
            m->attn_pk2g.assign(g->L, nullptr); m->fc_pk2g.assign(g->L, nullptr); m->attn_cs.assign(g->L, nullptr); m->fc_cs.assign(g->L, nullptr);
            auto packg = [&](std::vector<uint16_t *> &dst, std::vector<float *> &cs, size_t off, size_t goff, size_t R, size_t K, float scale, int l) -> int {
                MGPT_HIP(hipMalloc(&dst[l], R * K * NP * sizeof(uint16_t)));
                MGPT_HIP(hipMalloc(&cs[l], R * sizeof(float)));
                ProfScope ps(P_PACK, nullptr);
                hipLaunchKernelGGL((fastk::pack_pk_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)(R / 32) * (K / 16) * 64, 256)), dim3(256), 0,
                                   nullptr, g->params + off, dst[l], (int)R, (int)K, scale, g->params + goff);
                MGPT_LAUNCH_CHECK();
                hipLaunchKernelGGL((fastk::colsum_pk_kernel<T, NP>), dim3((unsigned)cdiv64((int64_t)R, 256)), dim3(256), 0, nullptr, g->params + off,
                                   g->params + goff, cs[l], (int)R, (int)K, scale);
                MGPT_LAUNCH_CHECK();
                return MGPT_OK;
            };
            for (int l = 0; l < g->L; l++) {
                const LayerOff &lo = g->layers[l];
                if ((rc = packg(m->attn_pk2g, m->attn_cs, lo.attn_w, lo.ln1, 3 * C, C, 1.0f / m->attn[l].inv_scale, l)) != MGPT_OK) return rc;
                if ((rc = packg(m->fc_pk2g, m->fc_cs, lo.fc_w, lo.ln2, 4 * C, C, 1.0f / m->fc[l].inv_scale, l)) != MGPT_OK) return rc;
            }
            MGPT_HIP(hipMalloc(&m->ln_parts, (size_t)(C / 128) * g->max_rows * kT * sizeof(float2)));
            // (the last layer's compact launch is padded to 256 rows and addresses token 255 of each: room and finite values for all of them)
            const size_t n_mean = (size_t)((g->max_rows + 255) / 256) * 256 * kT;
            MGPT_HIP(hipMalloc(&m->ln_mean, n_mean * sizeof(float)));
            MGPT_HIP(hipMemset(m->ln_mean, 0, n_mean * sizeof(float)));
        }
    }
    {
        const size_t nl = (size_t)((g->max_rows + 255) / 256) * 256 * C;
        MGPT_HIP(hipMalloc(&m->x_last, nl * sizeof(float)));
        MGPT_HIP(hipMemset(m->x_last, 0, nl * sizeof(float)));           // padding rows stay finite
        // chunk-major residual stream: the fused kernels (6M shape; 2M / tiny shapes with heads of 32) and the packed-fragment GEMM chain
        m->x_tiled = m->pk_gemm || (m->qkv_fused && g->hs == 32 && m->mlp_fused);
        if (m->x_tiled) MGPT_HIP(hipMalloc(&m->x_head, nl * sizeof(float)));
        if (m->pk_gemm) {
            MGPT_HIP(hipMalloc(&m->y_last, nl * NP * sizeof(uint16_t)));
            MGPT_HIP(hipMemset(m->y_last, 0, nl * NP * sizeof(uint16_t)));
        }
    }
    const size_t M = (size_t)g->max_rows * kT;
    MGPT_HIP(hipMalloc(&m->stats, M * sizeof(float2)));
    // q|k and v^T planes exist in HBM only where the projection is a GEMM of its own: the fused attention kernels keep them on chip
    const bool qkv_on_chip = m->attn256 || (m->qkv_fused && g->hs == 32 && m->mlp_fused);
    for (int p = 0; p < NP; p++) {
        if (!qkv_on_chip) {
            MGPT_HIP(hipMalloc(&m->qk[p], 2 * M * C * sizeof(uint16_t)));
            MGPT_HIP(hipMalloc(&m->vt[p], M * C * sizeof(uint16_t)));
        }
        if (m->pk_gemm && p > 0) continue;                                   // PK buffers interleave the planes in [0]
        MGPT_HIP(hipMalloc(&m->y[p], M * C * (m->pk_gemm ? NP : 1) * sizeof(uint16_t)));
        if (!m->mlp_fused) MGPT_HIP(hipMalloc(&m->hbuf[p], 4 * M * C * (m->pk_gemm ? NP : 1) * sizeof(uint16_t)));
    }
    MGPT_HIP(hipDeviceSynchronize());
    m->built = true;
    return MGPT_OK;
}

void free_mode(mgpt_gpt *g, ModeState *m)
{
    if (m->built) g->generation++;                       // captured step graphs hold these pointers
    auto fr = [](std::vector<PlaneSet> &v) { for (auto &p : v) { (void)hipFree(p.hi); (void)hipFree(p.lo); } v.clear(); };
    fr(m->attn); fr(m->proj); fr(m->fc); fr(m->proj2);
    for (auto *p : m->mlp_pk) (void)hipFree(p);
    for (auto *p : m->mlp16_pk) (void)hipFree(p);
    for (auto *p : m->mlp256_pk) (void)hipFree(p);
    for (auto *p : m->mlp256q_pk) (void)hipFree(p);
    for (auto *p : m->mlp256_lut) (void)hipFree(p);
    for (auto *p : m->attn256_pk) (void)hipFree(p);
    for (auto *p : m->attn256o_pk) (void)hipFree(p);
    for (auto *p : m->attn256q_pk) (void)hipFree(p);
    (void)hipFree(m->attn256o_spill);
    (void)hipFree(m->gelu_lut);
    for (auto *p : m->qkv_pk) (void)hipFree(p);
    for (auto *p : m->proj_pk) (void)hipFree(p);
    for (auto *v : {&m->attn_pk2, &m->proj_pk2, &m->fc_pk2, &m->proj2_pk2}) for (auto *p : *v) (void)hipFree(p);
    (void)hipFree(m->apk);
    (void)hipFree(m->stats);
    (void)hipFree(m->x_last);
    (void)hipFree(m->x_head);
    (void)hipFree(m->y_last);
    (void)hipFree(m->head_parts);
    (void)hipFree(m->last1_wt);
    for (auto p : m->attn_pk2g) (void)hipFree(p);
    for (auto p : m->fc_pk2g) (void)hipFree(p);
    for (auto p : m->attn_cs) (void)hipFree(p);
    for (auto p : m->fc_cs) (void)hipFree(p);
    (void)hipFree(m->ln_parts);
    (void)hipFree(m->ln_mean);
    for (auto p : m->attn160o_pk) (void)hipFree(p);
    (void)hipFree(m->attn160o_spill);
    for (auto *p : m->mlp160_pk) (void)hipFree(p);
    for (int p = 0; p < 2; p++) { (void)hipFree(m->qk[p]); (void)hipFree(m->vt[p]); (void)hipFree(m->y[p]); (void)hipFree(m->hbuf[p]); }
    *m = ModeState();
}

template <class T, int NP, int PRO, int EPI>
int launch_gemm16(fastk::GemmArgs a, int C, hipStream_t s)
{
    MGPT_REQUIRE(a.M % 128 == 0 && a.K % 32 == 0, MGPT_ERR_UNSUPPORTED, "gemm16 shape M=%d K=%d", a.M, a.K);
    const int mt = a.M / 128;
    if (C == 160 && a.N % 160 == 0) {
        a.n_tiles_n = a.N / 160;
        hipLaunchKernelGGL((fastk::gemm16_kernel<T, NP, 160, 4, 1, PRO, EPI>), dim3(mt * a.n_tiles_n), dim3(256), 0, s, a);
    } else if (C == 64 && a.N % 64 == 0) {
        a.n_tiles_n = a.N / 64;
        hipLaunchKernelGGL((fastk::gemm16_kernel<T, NP, 64, 4, 1, PRO, EPI>), dim3(mt * a.n_tiles_n), dim3(256), 0, s, a);
    } else if (a.N % 128 == 0) {
        a.n_tiles_n = a.N / 128;
        if (EPI == fastk::EPI_RESID) a.stats_out = nullptr;          // rows span two waves: stats come from row_stats_kernel
        hipLaunchKernelGGL((fastk::gemm16_kernel<T, NP, 128, 2, 2, PRO, EPI>), dim3(mt * a.n_tiles_n), dim3(256), 0, s, a);
    } else {
        set_error("gemm16: N=%d unsupported for C=%d", a.N, C);
        return MGPT_ERR_UNSUPPORTED;
    }
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

#ifdef MGPT_AB_MLPF_32X32
constexpr bool kMlpFused16 = false;     // A/B: the fused MLP block of the C = 64 / 160 shapes on the 32 x 32 x 16 MFMA in large calls too
#else
constexpr bool kMlpFused16 = true;
#endif
#ifdef MGPT_AB_ATTN_32X32
constexpr bool kAttn256Q = false;       // A/B: the 6M attention block's projections and tail on the 32 x 32 x 16 MFMA (attn256o_kernel) as in rounds 4-5
#else
constexpr bool kAttn256Q = true;
#endif
#ifdef MGPT_AB_MLP_32X32
constexpr bool kMlp256Q = false;        // A/B: the 6M MLP block of large calls on mlp256p_kernel (32 x 32 x 16 MFMA) as in rounds 3-4
#else
constexpr bool kMlp256Q = true;
#endif
#ifdef MGPT_AB_GEMM_32X32
constexpr bool kGemmPk16 = false;       // A/B: the one-plane packed GEMM on the 32 x 32 x 16 MFMA as in rounds 1-4
#else
constexpr bool kGemmPk16 = true;
#endif

// compute units of the current device (grid of the persistent packed GEMM); asked once
int gemm_pk_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

template <class T, int NP, int EPI>
int launch_gemm_pk(fastk::GemmArgs a, hipStream_t s, bool half_tiles = false, bool big_call = true)
{
    constexpr int NST = fastk::gemm_pk_nst(NP, 8, EPI), KPS = fastk::gemm_pk_kps(NP);
    MGPT_REQUIRE(a.M % 256 == 0 && a.N % 256 == 0 && a.K % 32 == 0 && a.K >= 16 * KPS * NST, MGPT_ERR_UNSUPPORTED,
                 "gemm_pk shape M=%d N=%d K=%d", a.M, a.N, a.K);
    a.n_tiles_n = a.N / 256;
    if (EPI == fastk::EPI_RESID) a.stats_out = nullptr;               // rows span two waves: stats come from row_stats_kernel
    const bool lut = EPI == fastk::EPI_GELU && a.gelu_lut != nullptr;
    const int n_cu = gemm_pk_cus();
    const bool lnf = EPI != fastk::EPI_RESID && a.ln_stats != nullptr;
    const bool persist = fastk::gemm_pk_persistent(NP, EPI, lnf);    // one workgroup per CU walking the tiles (gpt_kernels_fast.h: gemm_pk_kernel)
    const unsigned grid8 = (unsigned)(persist ? std::min((a.M / 256) * a.n_tiles_n, n_cu) : (a.M / 256) * a.n_tiles_n);
    const unsigned grid4 = (unsigned)(persist ? std::min((a.M / 128) * a.n_tiles_n, 2 * n_cu) : (a.M / 128) * a.n_tiles_n);
    // half_tiles (small launches: one environment's rows are 32 tiles of 256 rows per column tile, which leaves most CUs idle): 128-row tiles, 4 waves,
    // two workgroups per CU -- same arithmetic per token (a wave's 64 x 128 sub-tile and its k order do not change)
    const size_t lut_b = lut ? (size_t)fastk::kGeluLutN * 8 : 0;
    if constexpr (NP == 1 && kGemmPk16) {
        // one-plane mode: the same GEMM on v_mfma_f32_16x16x32 (gpt_kernels_fast.h: gemm_pk16_kernel), one workgroup per tile
        // -- in LARGE calls only: a 32-row forward is not at the power limit, and there the 12 % more cycles per flop of the small shape show (2.49 -> 2.71 ms);
        // the choice is a property of the call (as for the other small-launch kernels), so every chunk of a call runs the same arithmetic
        if (big_call && a.K % 64 == 0 && a.K >= 128 && (EPI != fastk::EPI_GELU || lut)) {
            const unsigned g8 = (unsigned)((a.M / 256) * a.n_tiles_n), g4 = (unsigned)((a.M / 128) * a.n_tiles_n);
            if (lnf) {
                if (half_tiles) hipLaunchKernelGGL((fastk::gemm_pk16_kernel<T, EPI, 4, true>), dim3(g4), dim3(256), (size_t)fastk::gemm_pk_lds(NP, 4, EPI) + lut_b + 3072, s, a);
                else hipLaunchKernelGGL((fastk::gemm_pk16_kernel<T, EPI, 8, true>), dim3(g8), dim3(512), (size_t)fastk::gemm_pk_lds(NP) + lut_b + 3072, s, a);
            } else {
                if (half_tiles) hipLaunchKernelGGL((fastk::gemm_pk16_kernel<T, EPI, 4, false>), dim3(g4), dim3(256), (size_t)fastk::gemm_pk_lds(NP, 4, EPI) + lut_b, s, a);
                else hipLaunchKernelGGL((fastk::gemm_pk16_kernel<T, EPI, 8, false>), dim3(g8), dim3(512), (size_t)fastk::gemm_pk_lds(NP) + lut_b, s, a);
            }
            MGPT_LAUNCH_CHECK();
            return MGPT_OK;
        }
    }
    if (lnf) {                                                       // folded LayerNorm (GemmArgs): the Phi table slot is always reserved
        MGPT_REQUIRE(EPI != fastk::EPI_GELU || lut, MGPT_ERR_STATE, "%s", "folded LayerNorm: the GELU epilogue needs the Phi table");
        if (half_tiles)
            hipLaunchKernelGGL((fastk::gemm_pk_kernel<T, NP, EPI, 4, 0, true>), dim3(grid4), dim3(256),
                               (size_t)fastk::gemm_pk_lds(NP, 4, EPI) + lut_b + 3072, s, a, (unsigned long long *)nullptr);
        else
            hipLaunchKernelGGL((fastk::gemm_pk_kernel<T, NP, EPI, 8, 0, true>), dim3(grid8), dim3(512),
                               (size_t)fastk::gemm_pk_lds(NP) + lut_b + 3072, s, a, (unsigned long long *)nullptr);
    } else if (half_tiles) {
        hipLaunchKernelGGL((fastk::gemm_pk_kernel<T, NP, EPI, 4>), dim3(grid4), dim3(256),
                           (size_t)fastk::gemm_pk_lds(NP, 4, EPI) + lut_b, s, a, (unsigned long long *)nullptr);
    } else {
        hipLaunchKernelGGL((fastk::gemm_pk_kernel<T, NP, EPI, 8>), dim3(grid8), dim3(512),
                           (size_t)fastk::gemm_pk_lds(NP) + lut_b, s, a, (unsigned long long *)nullptr);
    }
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

template <class T, int NP>
int launch_ln_pack(const float *x, const float *gain, uint16_t *out, int64_t M, int C, int tiled, hipStream_t s)
{
    ProfScope ps(P_LAYERNORM, s);
    const dim3 grid((unsigned)(M / 32));
    if (C == 256) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 4>), grid, dim3(256), 0, s, x, gain, out, C, tiled);
    else if (C == 512) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 8>), grid, dim3(256), 0, s, x, gain, out, C, tiled);
    else if (C == 768) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 12>), grid, dim3(256), 0, s, x, gain, out, C, tiled);
    else if (C == 1024) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 16>), grid, dim3(256), 0, s, x, gain, out, C, tiled);
    else { set_error("ln_pack: C=%d unsupported", C); return MGPT_ERR_UNSUPPORTED; }
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

bool fused_stats(int C) { return C == 160 || C == 64; }

int launch_row_stats(const float *x, float2 *stats, int64_t n_tok, int C, hipStream_t s)
{
    ProfScope ps(P_LAYERNORM, s);
    const dim3 grid((unsigned)cdiv64(n_tok, 4));
    if (C <= 256) hipLaunchKernelGGL((fastk::row_stats_kernel<1>), grid, dim3(256), 0, s, x, stats, n_tok, C);
    else if (C <= 512) hipLaunchKernelGGL((fastk::row_stats_kernel<2>), grid, dim3(256), 0, s, x, stats, n_tok, C);
    else if (C <= 768) hipLaunchKernelGGL((fastk::row_stats_kernel<3>), grid, dim3(256), 0, s, x, stats, n_tok, C);
    else hipLaunchKernelGGL((fastk::row_stats_kernel<4>), grid, dim3(256), 0, s, x, stats, n_tok, C);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

template <class T, int NP>
int forward_chunk(mgpt_gpt *g, ModeState *m, const uint8_t *d_tokens, int rows, float *d_logits, hipStream_t s, int call_rows)
{
    const int C = g->C;
    const int64_t M = (int64_t)rows * kT;
    const float *P = g->params;
    int rc;
    // register-resident path (C = 64, 160; head size 32): a layer is attn_block_kernel + mlp_fused_kernel
    const bool attn_block = m->qkv_fused && g->hs == 32 && m->mlp_fused;
    // small launch: the heads of a row run on different CUs (attn_block_kernel<HP>), their partial sums are folded by the next kernel
    // (decided by the CALL's row count, not the chunk's: the remainder chunk of a large call stays on the large-launch kernels)
    const bool head_par = attn_block && call_rows <= kSmallRows && rows <= kSmallRows;
    const int small_cap = std::min(g->max_rows, kSmallRows);                 // rows a small launch of this context can have
    const int64_t part_stride = (int64_t)small_cap * kT * C;
    if (head_par && m->head_parts == nullptr) {
        // first small launch of this context (105 MB for the 2M shape were allocated per precision mode and per adapter whether
        // or not the context ever served one environment).  New device memory: captured step graphs of this context are stale
        MGPT_HIP(hipMalloc(&m->head_parts, (size_t)g->nh * (size_t)part_stride * sizeof(float)));
        g->generation++;
    }
    // the first attention block forms x = wte[token] + wpe[position] itself (no embedding kernel, no first read of x)
    // ... unless the persistent block kernel of the 2M shape serves the layer (round 6, profiles/r06_ab.txt visit B): embed_tiled_kernel (0.6 ms per 16 384 rows)
    // + attn160o_kernel (4.5 ms) beat attn_block_kernel<EMBED> (5.8 ms), the round-1 kernel the first layer had stayed on because it gathers the embedding
#if defined(MGPT_AB_EMBED_FUSED_160)
    const bool persistent160 = false;
#else
    const bool persistent160 = attn_block && C == 160 && !head_par && m->attn160o_spill != nullptr && kAttn160o && m->x_tiled;
#endif
    const bool embed_fused = attn_block && g->L > 1 && !head_par && !persistent160;
    // ... and so does the 6M shape's persistent block kernel in a large call (attn256q_kernel<.., EMB>: from the (position, token) table, round 6)
#if defined(MGPT_AB_EMBED_KERNEL_256)
    const bool embed256 = false;
#else
    const bool embed256 = m->attn256 && kAttn256Fused && kAttn256Q && g->L > 1 && g->embed_table != nullptr && m->x_tiled &&
                          !(call_rows <= kSmallRows && rows <= kSmallRows && kSmall256);
#endif
#if defined(MGPT_AB_EMBED_KERNEL_160)
    const bool embed160 = false;
#else
    const bool embed160 = persistent160 && g->L > 1 && g->embed_table != nullptr;      // ... and the 2M shape's (attn160o_kernel<.., EMB>)
#endif
    if (embed256 || embed160) {
    } else if (m->x_tiled && !embed_fused) {
        ProfScope ps(P_EMBED, s);
        hipLaunchKernelGGL(fastk::embed_tiled_kernel, dim3((unsigned)(M / 32)), dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, C);
        MGPT_LAUNCH_CHECK();
    } else if (!embed_fused) {
        ProfScope ps(P_EMBED, s);
        const dim3 grid((unsigned)cdiv64(M, 4));
        if (C <= 256) hipLaunchKernelGGL((fastk::embed_stats_kernel<1>), grid, dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, m->stats, M, C);
        else if (C <= 512) hipLaunchKernelGGL((fastk::embed_stats_kernel<2>), grid, dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, m->stats, M, C);
        else if (C <= 768) hipLaunchKernelGGL((fastk::embed_stats_kernel<3>), grid, dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, m->stats, M, C);
        else hipLaunchKernelGGL((fastk::embed_stats_kernel<4>), grid, dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, m->stats, M, C);
        MGPT_LAUNCH_CHECK();
    }
    // folded LayerNorm: (mean, rstd) of the rows the next GEMM normalises live in m->stats, their raw operand planes in m->apk
    // (compact = the last layer's rows: token 255 of every row; their means are not needed again)
    auto ln_finalize = [&](int64_t Mrows, bool compact) -> int {
        ProfScope ps(P_LAYERNORM, s);
        hipLaunchKernelGGL(fastk::ln_finalize_kernel, dim3((unsigned)cdiv64(Mrows, 256)), dim3(256), 0, s, m->ln_parts, C / 128, Mrows, C, m->stats,
                           m->ln_mean, compact ? kT : 1, compact ? kT - 1 : 0, compact ? 0 : 1);
        MGPT_LAUNCH_CHECK();
        return MGPT_OK;
    };
    if (m->ln_fold) {                                          // the embedding rows: the one ln_pack_kernel launch of the forward (raw mode)
        ProfScope ps(P_LAYERNORM, s);
        const dim3 grid((unsigned)(M / 32));
        const float *g0 = P + g->layers[0].ln1;
        if (C == 256) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 4>), grid, dim3(256), 0, s, g->x, g0, m->apk, C, m->x_tiled ? 1 : 0, m->stats, m->ln_mean);
        else if (C == 512) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 8>), grid, dim3(256), 0, s, g->x, g0, m->apk, C, m->x_tiled ? 1 : 0, m->stats, m->ln_mean);
        else if (C == 768) hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 12>), grid, dim3(256), 0, s, g->x, g0, m->apk, C, m->x_tiled ? 1 : 0, m->stats, m->ln_mean);
        else hipLaunchKernelGGL((fastk::ln_pack_kernel<T, NP, 16>), grid, dim3(256), 0, s, g->x, g0, m->apk, C, m->x_tiled ? 1 : 0, m->stats, m->ln_mean);
        MGPT_LAUNCH_CHECK();
    }
    const float scale_log2e = (1.0f / sqrtf((float)g->hs)) * 1.44269504088896340736f;
    const size_t attn_lds = (size_t)NP * (kT * (g->hs + 8) * 2 + g->hs * (kT + 8) * 2);
    for (int l = 0; l < g->L; l++) {
        const LayerOff &lo = g->layers[l];
        fastk::GemmArgs a = {};
        a.M = (int)M; a.C = C; a.n_head = g->nh; a.hs = g->hs; a.plane = M * C;
        a.x = g->x; a.stats = m->stats; a.gain = P + lo.ln1; a.K = C;
        a.w_hi = m->attn[l].hi; a.w_lo = m->attn[l].lo; a.out_scale = m->attn[l].inv_scale;
        a.N = 2 * C; a.o_hi = m->qk[0]; a.o_lo = m->qk[1];
        // last layer: only token 255 is needed downstream (model.py:186) -> compact buffer, MLP and head on `rows` tokens
        const bool last_short = (attn_block || m->pk_gemm) && l == g->L - 1;
        const int rows_pad = ((rows + 255) / 256) * 256;
        // 6M shape, every layer but the last: attn256o_kernel does the out-projection and the residual add itself
        // ... unless the CALL is small (one environment, rows <= kSmallRows): then a row's eight heads run on eight CUs at once
        // (attn256_kernel<.., HP>, one workgroup per (row, head), y planes) and the packed GEMM projects them (round 5)
        const bool small256 = m->attn256 && !last_short && call_rows <= kSmallRows && rows <= kSmallRows && kSmall256;
        const bool proj_fused = m->attn256 && !last_short && kAttn256Fused && !small256;
        // the packed-GEMM chain (C = 768) in a small call: its residual GEMMs (N = 768: 3 column tiles) are 96 tiles of 256 rows for 32 rows -- 128-row
        // tiles put them on twice the CUs (same arithmetic per token: a wave's 64 x 128 sub-tile and its k order do not change).  One-plane mode: 32-row
        // forward 2.77 -> 2.49 ms (c_proj 70 -> 56 us, out-projection 30 -> 25, q|k + v^T 51 -> 47); in the split mode the 4-wave form has a 3-stage ring and
        // loses (5.8 -> 6.4)
        const bool small_pk = NP == 1 && m->pk_gemm && !m->attn256 && call_rows <= kSmallRows && rows <= kSmallRows;
        const bool big_call = call_rows > kSmallRows;                      // the one-plane packed GEMMs run on the 16 x 16 x 32 MFMA (launch_gemm_pk)
        // last layer of a launch that fills the chip: the attention block of token 255 alone, without K and V (attn_last1_kernel)
        const bool last1 = last_short && m->last1_wt != nullptr && m->x_tiled && !head_par && kLast1;
        // small launch (one environment): the last layer's attention block, its MLP block, ln_f and the head are ONE launch, one
        // workgroup per row (attn_last1_kernel<.., 1, TAIL>), fp32 throughout; logits come straight out of it
        const bool small_call = head_par || (m->attn256 && call_rows <= kSmallRows && rows <= kSmallRows && kSmall256);
        const bool last_tail = last_short && small_call && m->last1_wt != nullptr && kLast1 && kLast1Tail;
        if (last_tail) {
            ProfScope ps(P_ATTN_LAST, s);
            const float *wk = P + lo.attn_w + (size_t)C * C;
#define MGPT_LAST1T(C_)                                                                                                                     \
    hipLaunchKernelGGL((fastk::attn_last1_kernel<C_, 32, 1, true>), dim3((unsigned)rows), dim3(256 * fastk::last1_split(C_)),              \
                       (size_t)(fastk::kLast1Lds<C_, 1, true>), s, g->x, P + lo.ln1, wk, m->last1_wt, (float *)nullptr, rows, scale_log2e,    \
                       P + lo.ln2, P + g->off_lnf, P + g->off_wte, d_logits, kV)
            if (C == 256) MGPT_LAST1T(256); else if (C == 160) MGPT_LAST1T(160); else MGPT_LAST1T(64);
#undef MGPT_LAST1T
            MGPT_LAUNCH_CHECK();
            return MGPT_OK;
        }
        if (last1) {
            ProfScope ps(P_ATTN_LAST, s);
            const float *wk = P + lo.attn_w + (size_t)C * C;
#define MGPT_LAST1(C_)                                                                                                                      \
    hipLaunchKernelGGL((fastk::attn_last1_kernel<C_, 32>), dim3((unsigned)(rows_pad / fastk::kLast1R)), dim3(256 * fastk::last1_split(C_)), (size_t)fastk::kLast1Lds<C_>, s, g->x, P + lo.ln1, \
                       wk, m->last1_wt, m->x_last, rows, scale_log2e)
            if (C == 256 && rows <= 2 * m->n_cu)
                // few rows (one environment on the 6M shape): one row per workgroup, so that they spread over the CUs
                hipLaunchKernelGGL((fastk::attn_last1_kernel<256, 32, 1, false>), dim3((unsigned)rows_pad), dim3(256 * fastk::last1_split(256)),
                                   (size_t)(fastk::kLast1Lds<256, 1, false>), s, g->x, P + lo.ln1, wk, m->last1_wt, m->x_last, rows, scale_log2e);
            else if (C == 256) MGPT_LAST1(256); else if (C == 160) MGPT_LAST1(160); else MGPT_LAST1(64);
#undef MGPT_LAST1
            MGPT_LAUNCH_CHECK();
        } else if (attn_block && C == 160 && !head_par && !last_short && !(embed_fused && l == 0) && m->attn160o_spill != nullptr && kAttn160o) {
            // ---- the 2M shape's attention block as one persistent kernel (attn160o_kernel) ----
            ProfScope ps(P_ATTN, s);
            if (embed160 && l == 0)
                hipLaunchKernelGGL((fastk::attn160o_kernel<T, NP, true>), dim3((unsigned)std::min(rows, m->n_cu)), dim3(512), (size_t)fastk::kA160oLds<NP>, s, g->x,
                                   m->attn160o_pk[l], m->attn160_inv[l], scale_log2e, m->proj[l].inv_scale, m->attn160o_spill, rows, d_tokens, g->embed_table);
            else
                hipLaunchKernelGGL((fastk::attn160o_kernel<T, NP>), dim3((unsigned)std::min(rows, m->n_cu)), dim3(512), (size_t)fastk::kA160oLds<NP>, s, g->x,
                                   m->attn160o_pk[l], m->attn160_inv[l], scale_log2e, m->proj[l].inv_scale, m->attn160o_spill, rows);
            MGPT_LAUNCH_CHECK();
        } else if (attn_block) {
            // ---- LN1 + QKV + attention + out-projection + residual in one kernel: q, k, v, y stay on chip ----
            ProfScope ps(last_short ? P_ATTN_LAST : P_ATTN, s);
            const size_t lds = (size_t)NP * (kT * 80 + 32 * 528) + (size_t)(C / 16) * NP * 1024 * 4;   // K, V^T planes + 4 weight packet slots
#define MGPT_ATTN_BLOCK(CT_, LAST_, EMB_, HP_)                                                                                              \
    hipLaunchKernelGGL((fastk::attn_block_kernel<T, NP, CT_, true, LAST_, EMB_, HP_>), dim3((unsigned)(HP_ ? rows * g->nh : rows)), dim3(512), \
                       lds, s, g->x, P + lo.ln1, m->qkv_pk[l], m->attn[l].inv_scale, m->y[0], m->y[1], g->nh, scale_log2e, m->proj_pk[l],     \
                       m->proj[l].inv_scale, m->stats, m->x_last, d_tokens, P + g->off_wte, P + g->off_wpe, m->head_parts, part_stride)
            const bool emb = embed_fused && l == 0;
            if (head_par) {
                if (C == 160) { if (last_short) MGPT_ATTN_BLOCK(5, true, false, true); else MGPT_ATTN_BLOCK(5, false, false, true); }
                else { if (last_short) MGPT_ATTN_BLOCK(2, true, false, true); else MGPT_ATTN_BLOCK(2, false, false, true); }
            } else if (C == 160) { if (last_short) MGPT_ATTN_BLOCK(5, true, false, false); else if (emb) MGPT_ATTN_BLOCK(5, false, true, false); else MGPT_ATTN_BLOCK(5, false, false, false); }
            else { if (last_short) MGPT_ATTN_BLOCK(2, true, false, false); else if (emb) MGPT_ATTN_BLOCK(2, false, true, false); else MGPT_ATTN_BLOCK(2, false, false, false); }
#undef MGPT_ATTN_BLOCK
            MGPT_LAUNCH_CHECK();
            if (head_par && last_short) {
                // new row of token 255 = its residual row + the heads' contributions (compact partial sums), padding rows zero
                hipLaunchKernelGGL(fastk::gather_last_kernel, dim3((unsigned)cdiv64((int64_t)rows_pad * (C / 4), 256)), dim3(256), 0, s, g->x, m->x_last, rows,
                                   rows_pad, C, 1, m->head_parts, g->nh, part_stride);
                MGPT_LAUNCH_CHECK();
            }
        } else if (m->attn256 && proj_fused) {
            // ---- the whole attention block (LN1, QKV, attention, out-projection, residual) in one persistent kernel: q, k, v, y stay on chip ----
            ProfScope ps(P_ATTN, s);
            // (gpt_kernels_c256b.h: the projection steps and the tail on v_mfma_f32_16x16x32, the attention phase as it was)
            if (kAttn256Q && embed256 && l == 0)
                hipLaunchKernelGGL((fastk::attn256q_kernel<T, NP, 0, true>), dim3((unsigned)std::min(rows, m->n_cu)), dim3(512), (size_t)kA256Lds<NP>, s, g->x,
                                   m->attn256q_pk[l], m->attn256_inv[l], scale_log2e, m->proj[l].inv_scale, m->attn256o_spill, rows,
                                   (unsigned long long *)nullptr, d_tokens, g->embed_table);
            else if (kAttn256Q)
                hipLaunchKernelGGL((fastk::attn256q_kernel<T, NP>), dim3((unsigned)std::min(rows, m->n_cu)), dim3(512), (size_t)kA256Lds<NP>, s, g->x,
                                   m->attn256q_pk[l], m->attn256_inv[l], scale_log2e, m->proj[l].inv_scale, m->attn256o_spill, rows,
                                   (unsigned long long *)nullptr);
            else
                hipLaunchKernelGGL((fastk::attn256o_kernel<T, NP>), dim3((unsigned)std::min(rows, m->n_cu)), dim3(512), (size_t)kA256Lds<NP>, s, g->x,
                                   m->attn256o_pk[l], m->attn256_inv[l], scale_log2e, m->proj[l].inv_scale, m->attn256o_spill, rows,
                                   (unsigned long long *)nullptr);
            MGPT_LAUNCH_CHECK();
        } else if (m->attn256) {
            // ---- LN1 + QKV + attention in one kernel (q, k, v stay on chip) -> y operand planes for the out-projection ----
            ProfScope ps(last_short ? P_ATTN_LAST : P_ATTN, s);
            if (last_short)
                hipLaunchKernelGGL((fastk::attn256_kernel<T, NP, true>), dim3((unsigned)rows), dim3(512), (size_t)kA256Lds<NP>, s, g->x,
                                   m->attn256_pk[l], m->attn256_inv[l], scale_log2e, m->y_last, (unsigned long long *)nullptr);
            else if (small256)
                hipLaunchKernelGGL((fastk::attn256_kernel<T, NP, false, 0, fastk::kA256Stagger, true>), dim3((unsigned)rows * 8), dim3(512), (size_t)kA256Lds<NP>, s,
                                   g->x, m->attn256_pk[l], m->attn256_inv[l], scale_log2e, m->y[0], (unsigned long long *)nullptr);
            else
                hipLaunchKernelGGL((fastk::attn256_kernel<T, NP, false>), dim3((unsigned)rows), dim3(512), (size_t)kA256Lds<NP>, s, g->x,
                                   m->attn256_pk[l], m->attn256_inv[l], scale_log2e, m->y[0], (unsigned long long *)nullptr);
            MGPT_LAUNCH_CHECK();
        } else if (m->pk_gemm) {
            // (folded LayerNorm: m->apk already holds the raw rows -- from the embedding or from the residual epilogue before -- and
            //  m->stats their (mean, rstd); the W * ln_1.weight packing and its column sums do the rest in the epilogue)
            if (!m->ln_fold && (rc = launch_ln_pack<T, NP>(g->x, P + lo.ln1, m->apk, M, C, m->x_tiled ? 1 : 0, s)) != MGPT_OK) return rc;
            ProfScope ps(P_GEMM_QKV, s);
            const size_t tile_halves = (size_t)(C / 16) * NP * 512;          // one 32-row tile of a PK matrix with K = C
            const uint16_t *wq = m->ln_fold ? m->attn_pk2g[l] : m->attn_pk2[l];
            a.a_hi = m->apk; a.w_hi = wq; a.chunk_major = 1;
            if (m->ln_fold) { a.ln_stats = m->stats; a.colsum = m->attn_cs[l]; }
            if ((rc = launch_gemm_pk<T, NP, fastk::EPI_QK>(a, s, small_pk, big_call)) != MGPT_OK) return rc;
            a.w_hi = wq + (size_t)(2 * C / 32) * tile_halves;                // rows 2C.. of c_attn.weight: V
            if (m->ln_fold) a.colsum = m->attn_cs[l] + 2 * C;
            a.N = C; a.o_hi = m->vt[0]; a.o_lo = m->vt[1];
            if ((rc = launch_gemm_pk<T, NP, fastk::EPI_VT>(a, s, small_pk, big_call)) != MGPT_OK) return rc;
            a.ln_stats = nullptr; a.colsum = nullptr;
        } else {
            ProfScope ps(P_GEMM_QKV, s);
            if ((rc = launch_gemm16<T, NP, fastk::PRO_LN, fastk::EPI_QK>(a, C, s)) != MGPT_OK) return rc;
            // ---- LN1 + V projection -> v^T planes ----
            a.w_hi = m->attn[l].hi + (size_t)2 * C * C; a.w_lo = (NP == 2) ? m->attn[l].lo + (size_t)2 * C * C : nullptr;
            a.N = C; a.o_hi = m->vt[0]; a.o_lo = m->vt[1];
            if ((rc = launch_gemm16<T, NP, fastk::PRO_LN, fastk::EPI_VT>(a, C, s)) != MGPT_OK) return rc;
        }
        if (!attn_block && !proj_fused && !last1) {
            const bool ls = last_short && m->pk_gemm;       // only token 255 of every row from here on
            if (!m->attn256) {
                ProfScope ps(ls ? P_ATTN_LAST : P_ATTN, s);
                const uint16_t *qh = m->qk[0], *ql = m->qk[1], *kh = m->qk[0] + M * C, *kl = (NP == 2) ? m->qk[1] + M * C : nullptr;
                uint16_t *yh = ls ? m->y_last : m->y[0];
                if (g->hs == 32)
                    hipLaunchKernelGGL((fastk::attn16_kernel<T, NP, 32>), dim3(rows * g->nh), dim3(512), attn_lds, s, qh, ql, kh, kl, m->vt[0], m->vt[1], yh, m->y[1], g->nh, scale_log2e, m->pk_gemm ? 1 : 0, m->pk_gemm ? 1 : 0, ls ? 1 : 0);
                else
                    hipLaunchKernelGGL((fastk::attn16_kernel<T, NP, 64>), dim3(rows * g->nh), dim3(512), attn_lds, s, qh, ql, kh, kl, m->vt[0], m->vt[1], yh, m->y[1], g->nh, scale_log2e, m->pk_gemm ? 1 : 0, m->pk_gemm ? 1 : 0, ls ? 1 : 0);
                MGPT_LAUNCH_CHECK();
            }
            if (ls) {
                ProfScope ps(P_EMBED, s);
                hipLaunchKernelGGL(fastk::gather_last_kernel, dim3((unsigned)cdiv64((int64_t)rows_pad * (C / 4), 256)), dim3(256), 0, s, g->x, m->x_last, rows,
                                   rows_pad, C, m->x_tiled ? 1 : 0);
                MGPT_LAUNCH_CHECK();
            }
            // ---- attention output projection + residual (+ stats of the new rows) ----
            a.a_hi = ls ? m->y_last : m->y[0]; a.a_lo = m->y[1]; a.K = C; a.N = C;
            a.w_hi = m->proj[l].hi; a.w_lo = m->proj[l].lo; a.out_scale = m->proj[l].inv_scale;
            a.x_out = ls ? m->x_last : g->x; a.stats_out = m->stats; a.x_tiled = m->x_tiled ? 1 : 0;
            if (ls) a.M = rows_pad;
            ProfScope ps(ls ? P_GEMM_PROJ_LAST : P_GEMM_PROJ, s);
            if (m->pk_gemm) {
                a.w_hi = m->proj_pk2[l];
                if (m->ln_fold) {                                                      // the rows ln_2 normalises: planes + partial sums
                    a.raw_out = m->apk; a.rsum_out = m->ln_parts;
                    a.shift = m->ln_mean; a.shift_stride = ls ? kT : 1; a.shift_offset = ls ? kT - 1 : 0;
                }
                if ((rc = launch_gemm_pk<T, NP, fastk::EPI_RESID>(a, s, small256 || (small_pk && !ls), big_call)) != MGPT_OK) return rc;
                a.raw_out = nullptr; a.rsum_out = nullptr; a.shift = nullptr;
            } else if ((rc = launch_gemm16<T, NP, fastk::PRO_PLANES, fastk::EPI_RESID>(a, C, s)) != MGPT_OK) return rc;
        }
        if (!fused_stats(C) && !m->pk_gemm && (rc = launch_row_stats(g->x, m->stats, M, C, s)) != MGPT_OK) return rc;
        float *mlp_x = last_short ? m->x_last : g->x;
        const int64_t mlp_M = last_short ? (int64_t)rows_pad : M;
        if (m->mlp_fused) {
            // ---- whole MLP block in one kernel (hidden stays in registers) ----
            ProfScope ps(last_short ? P_MLP_FUSED_LAST : P_MLP_FUSED, s);
            if (C == 256) {
                // persistent: one workgroup per CU walks the 128-token blocks round-robin (results do not depend on the grid)
                const int n_blocks = (int)(mlp_M / 128);
                if (small256 && 2 * n_blocks <= m->n_cu)
                    // small launch: 64-token blocks, four waves with a SIMD each (mlp256p_kernel<.., NPAIR = 2>): twice the workgroups
                    hipLaunchKernelGGL((fastk::mlp256p_kernel<T, NP, 0, 2>), dim3((unsigned)std::min(2 * n_blocks, m->n_cu)), dim3(256), (size_t)fastk::kMPLds<NP>, s,
                                       mlp_x, m->mlp256_pk[l], m->mlp256_inv1[l], m->proj2[l].inv_scale, m->mlp256_lut[l], 2 * n_blocks, (unsigned long long *)nullptr);
                else if (small256 || !kMlp256Q)
                hipLaunchKernelGGL((fastk::mlp256p_kernel<T, NP>), dim3((unsigned)std::min(n_blocks, m->n_cu)), dim3(512), (size_t)fastk::kMPLds<NP>, s,
                                   mlp_x, m->mlp256_pk[l], m->mlp256_inv1[l], m->proj2[l].inv_scale, m->mlp256_lut[l], n_blocks, (unsigned long long *)nullptr);
                else
                    // large calls: the same block on v_mfma_f32_16x16x32 (gpt_kernels_c256q.h: 13-15 % more f16 flops per second at the power limit; 2.66 -> 2.45 ms
                    // per 4096-row launch).  Small calls are not power-limited and keep the 32 x 32 x 16 kernel; the choice is a property of the call
                    hipLaunchKernelGGL((fastk::mlp256q_kernel<T, NP>), dim3((unsigned)std::min(n_blocks, m->n_cu)), dim3(512), (size_t)fastk::kMPLds<NP>, s,
                                       mlp_x, m->mlp256q_pk[l], m->mlp256_inv1[l], m->proj2[l].inv_scale, m->mlp256_lut[l], n_blocks, (unsigned long long *)nullptr);
                if (!m->pk_gemm && l + 1 < g->L) {                       // this kernel leaves no LayerNorm statistics behind
                    MGPT_LAUNCH_CHECK();
                    if ((rc = launch_row_stats(g->x, m->stats, M, C, s)) != MGPT_OK) return rc;
                }
            } else if (head_par && C == 160) {
                // small launch: persistent producer / consumer blocks of 128 tokens (two waves per SIMD); the heads' partial sums are folded
                // in (the last layer's compact rows were folded by gather_last_kernel)
                const int n_blocks = (int)(mlp_M / 128);
                if (last_short)
                    hipLaunchKernelGGL((fastk::mlp160p_kernel<T, NP, 0>), dim3((unsigned)std::min(n_blocks, m->n_cu)), dim3(512), (size_t)fastk::kM5Lds<NP>, s,
                                       mlp_x, m->mlp160_pk[l], m->mlp160_inv1[l], m->proj2[l].inv_scale, m->gelu_lut, n_blocks, (const float *)nullptr, (int64_t)0);
                else if (mlp_M <= (int64_t)32 * m->n_cu && kMlp160Half && kMlp160Quarter)
                    // ... or 32-token blocks, one producer and one consumer wave (cfg1's 8192 tokens: one block per CU)
                    hipLaunchKernelGGL((fastk::mlp160p_kernel<T, NP, 5, 1>), dim3((unsigned)std::min((int)(mlp_M / 32), m->n_cu)), dim3(128), (size_t)(fastk::kM5Lds<NP, 1>), s,
                                       mlp_x, m->mlp160_pk[l], m->mlp160_inv1[l], m->proj2[l].inv_scale, m->gelu_lut, (int)(mlp_M / 32), m->head_parts, part_stride);
                else if (mlp_M <= (int64_t)64 * m->n_cu && kMlp160Half)
                    // so few tokens that 128-token blocks would leave CUs idle (one environment): 64-token blocks, one wave per SIMD
                    hipLaunchKernelGGL((fastk::mlp160p_kernel<T, NP, 5, 2>), dim3((unsigned)std::min((int)(mlp_M / 64), m->n_cu)), dim3(256), (size_t)(fastk::kM5Lds<NP, 2>), s,
                                       mlp_x, m->mlp160_pk[l], m->mlp160_inv1[l], m->proj2[l].inv_scale, m->gelu_lut, (int)(mlp_M / 64), m->head_parts, part_stride);
                else
                    hipLaunchKernelGGL((fastk::mlp160p_kernel<T, NP, 5>), dim3((unsigned)std::min(n_blocks, m->n_cu)), dim3(512), (size_t)fastk::kM5Lds<NP>, s,
                                       mlp_x, m->mlp160_pk[l], m->mlp160_inv1[l], m->proj2[l].inv_scale, m->gelu_lut, n_blocks, m->head_parts, part_stride);
            } else {
                const size_t lds = (size_t)(C / 16 + 2 * (C / 32)) * NP * 1024 * 3 + fastk::kGeluLutN * 8;
                // 32 tokens per wave whatever the block size: small launches (cfg1: 32 rows = 32 blocks of 256 tokens) take
                // fewer waves per block so that the tokens spread over more CUs; results do not depend on the choice
#define MGPT_MLP_(CT_, NW_, NF_)                                                                                                     \
    hipLaunchKernelGGL((fastk::mlp_fused_kernel<T, NP, CT_, NW_, NF_>), dim3((unsigned)(mlp_M / (32 * NW_))), dim3(64 * NW_), lds, s, mlp_x, \
                       P + lo.ln2, m->mlp_pk[l], m->fc[l].inv_scale, m->proj2[l].inv_scale, last_short ? nullptr : m->stats, (int)mlp_M, m->gelu_lut, \
                       m->head_parts, part_stride)
    // large calls: the same block on v_mfma_f32_16x16x32 (gpt_kernels_fused16.h); small calls (head_par) are not power-limited and keep the 32 x 32 x 16 kernel
#define MGPT_MLP16_(CT_, NW_)                                                                                                        \
    hipLaunchKernelGGL((fastk::mlp_fused16_kernel<T, NP, CT_, NW_, 0>), dim3((unsigned)(mlp_M / (32 * NW_))), dim3(64 * NW_), lds, s, mlp_x, \
                       P + lo.ln2, m->mlp16_pk[l], m->fc[l].inv_scale, m->proj2[l].inv_scale, last_short ? nullptr : m->stats, (int)mlp_M, m->gelu_lut, \
                       (const float *)nullptr, (int64_t)0)
#define MGPT_MLP(CT_, NW_) do { if (fold_parts) MGPT_MLP_(CT_, NW_, CT_); else if (!head_par && kMlpFused16) MGPT_MLP16_(CT_, NW_); else MGPT_MLP_(CT_, NW_, 0); } while (0)
                // (one wave per workgroup -- 256 workgroups for cfg1's 8192 tokens -- is slower: 63 us per launch against 52, round 4)
                const int nw = (mlp_M >= (int64_t)256 * m->n_cu) ? 8 : (mlp_M >= (int64_t)128 * m->n_cu ? 4 : 2);
                // (the heads' partial sums = n_head buffers, and n_head == C / 32 for the shapes of this path;
                //  the last layer's compact rows were folded by gather_last_kernel)
                const bool fold_parts = head_par && !last_short;
                if (C == 160) { if (nw == 8) MGPT_MLP(5, 8); else if (nw == 4) MGPT_MLP(5, 4); else MGPT_MLP(5, 2); }
                else { if (nw == 8) MGPT_MLP(2, 8); else if (nw == 4) MGPT_MLP(2, 4); else MGPT_MLP(2, 2); }
#undef MGPT_MLP
#undef MGPT_MLP16_
#undef MGPT_MLP_
            }
            MGPT_LAUNCH_CHECK();
            continue;
        }
        if (m->pk_gemm) {
            // ---- LN2 -> PK planes; FC + GELU -> hidden PK planes; proj2 + residual ----
            if (m->ln_fold) { if ((rc = ln_finalize(mlp_M, last_short)) != MGPT_OK) return rc; }
            else if ((rc = launch_ln_pack<T, NP>(mlp_x, P + lo.ln2, m->apk, mlp_M, C, m->x_tiled ? 1 : 0, s)) != MGPT_OK) return rc;
            a.M = (int)mlp_M;
            a.a_hi = m->apk; a.K = C; a.N = 4 * C; a.w_hi = m->ln_fold ? m->fc_pk2g[l] : m->fc_pk2[l]; a.out_scale = m->fc[l].inv_scale;
            a.o_hi = m->hbuf[0]; a.o_pk = 1; a.gelu_lut = m->gelu_lut;
            if (m->ln_fold) { a.ln_stats = m->stats; a.colsum = m->fc_cs[l]; }
            {
                ProfScope ps(P_GEMM_FC, s);
                // (c_fc keeps its 256-row tiles in small calls too: 384 of them already cover the CUs, and two 4-wave workgroups per CU move 1.5 x the ring
                //  pieces per MFMA of one 8-wave workgroup -- 54.6 vs 56.5 us per launch at 32 rows)
                if ((rc = launch_gemm_pk<T, NP, fastk::EPI_GELU>(a, s, false, big_call)) != MGPT_OK) return rc;
            }
            a.ln_stats = nullptr; a.colsum = nullptr;
            a.a_hi = m->hbuf[0]; a.K = 4 * C; a.N = C; a.w_hi = m->proj2_pk2[l]; a.out_scale = m->proj2[l].inv_scale;
            a.x_out = mlp_x; a.stats_out = nullptr;
            const bool feeds_next = m->ln_fold && l + 1 < g->L;            // the rows the next layer's ln_1 normalises
            if (feeds_next) { a.raw_out = m->apk; a.rsum_out = m->ln_parts; a.shift = m->ln_mean; a.shift_stride = 1; a.shift_offset = 0; }
            {
                ProfScope ps(P_GEMM_PROJ2, s);
                if ((rc = launch_gemm_pk<T, NP, fastk::EPI_RESID>(a, s, small_pk && !last_short, big_call)) != MGPT_OK) return rc;
            }
            if (feeds_next && (rc = ln_finalize(mlp_M, false)) != MGPT_OK) return rc;
            continue;
        }
        // ---- LN2 + FC + GELU -> hidden planes ----
        a.x = g->x; a.stats = m->stats; a.gain = P + lo.ln2; a.K = C; a.N = 4 * C;
        a.w_hi = m->fc[l].hi; a.w_lo = m->fc[l].lo; a.out_scale = m->fc[l].inv_scale;
        a.o_hi = m->hbuf[0]; a.o_lo = m->hbuf[1];
        {
            ProfScope ps(P_GEMM_FC, s);
            if ((rc = launch_gemm16<T, NP, fastk::PRO_LN, fastk::EPI_GELU>(a, C, s)) != MGPT_OK) return rc;
        }
        // ---- MLP output projection + residual (+ stats) ----
        a.a_hi = m->hbuf[0]; a.a_lo = m->hbuf[1]; a.K = 4 * C; a.N = C;
        a.w_hi = m->proj2[l].hi; a.w_lo = m->proj2[l].lo; a.out_scale = m->proj2[l].inv_scale;
        a.x_out = g->x; a.stats_out = m->stats;
        {
            ProfScope ps(P_GEMM_PROJ2, s);
            if ((rc = launch_gemm16<T, NP, fastk::PRO_PLANES, fastk::EPI_RESID>(a, C, s)) != MGPT_OK) return rc;
        }
        if (!fused_stats(C) && l + 1 < g->L && (rc = launch_row_stats(g->x, m->stats, M, C, s)) != MGPT_OK) return rc;
    }
    if (m->x_tiled) {
        {
            ProfScope ps(P_HEAD, s);
            hipLaunchKernelGGL(fastk::untile_rows_kernel, dim3((unsigned)cdiv64((int64_t)rows * (C / 4), 256)), dim3(256), 0, s, m->x_last, m->x_head, rows, C);
            MGPT_LAUNCH_CHECK();
        }
        return gpt_launch_head_at(g, m->x_head, (int64_t)C, 0, rows, d_logits, s);
    }
    if (attn_block || m->pk_gemm) return gpt_launch_head_at(g, m->x_last, (int64_t)C, 0, rows, d_logits, s);
    return gpt_launch_head(g, rows, d_logits, s);
}

template <class T, int NP, int HS>
int raise_attn_lds(size_t bytes)
{
    MGPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fastk::attn16_kernel<T, NP, HS>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MGPT_OK;
}

}  // namespace

// ---- entry points used by gpt.hip ----
int gpt_fast_finalize(mgpt_gpt *g)
{
    FastState *f = static_cast<FastState *>(g->fast);
    if (f) {                                             // parameters changed: planes are rebuilt lazily
        for (auto &m : f->mode) free_mode(g, &m);
        (void)hipFree(g->embed_table);                   // ... and so is the (position, token) table of layer 0
        g->embed_table = nullptr;
    } else {
        g->fast = new FastState();
    }
    return MGPT_OK;
}

void gpt_fast_destroy(mgpt_gpt *g)
{
    FastState *f = static_cast<FastState *>(g->fast);
    if (!f) return;
    for (auto &m : f->mode) free_mode(g, &m);
    delete f;
    g->fast = nullptr;
    (void)hipFree(g->embed_table);
    g->embed_table = nullptr;
}

int gpt_fast_forward(mgpt_gpt *g, const uint8_t *d_tokens, int rows, float *d_logits, int precision, hipStream_t s, int call_rows)
{
    MGPT_REQUIRE(precision == MGPT_PREC_F16X3 || precision == MGPT_PREC_BF16, MGPT_ERR_ARG, "unknown precision %d", precision);
    FastState *f = static_cast<FastState *>(g->fast);
    MGPT_REQUIRE(f, MGPT_ERR_STATE, "mgpt_gpt_finalize must precede forward");
    ModeState *m = &f->mode[precision];
    const size_t attn_lds = (size_t)(precision == MGPT_PREC_F16X3 ? 2 : 1) * (kT * (g->hs + 8) * 2 + g->hs * (kT + 8) * 2);
    if (!m->built) {                                     // first use of this precision: pack planes, size workspace
        int rc;
        if (precision == MGPT_PREC_F16X3) {
            rc = build_mode<fastk::F16T, 2>(g, m, true);
            if (rc == MGPT_OK) rc = (g->hs == 32) ? raise_attn_lds<fastk::F16T, 2, 32>(attn_lds) : raise_attn_lds<fastk::F16T, 2, 64>(attn_lds);
        } else {
            rc = build_mode<fastk::BF16T, 1>(g, m, false);
            if (rc == MGPT_OK) rc = (g->hs == 32) ? raise_attn_lds<fastk::BF16T, 1, 32>(attn_lds) : raise_attn_lds<fastk::BF16T, 1, 64>(attn_lds);
        }
        if (rc != MGPT_OK) { free_mode(g, m); return rc; }         // no half-built planes survive a failed build (e.g. out of memory)
    }
    if (precision == MGPT_PREC_F16X3) return forward_chunk<fastk::F16T, 2>(g, m, d_tokens, rows, d_logits, s, call_rows);
    return forward_chunk<fastk::BF16T, 1>(g, m, d_tokens, rows, d_logits, s, call_rows);
}

// test/debug: raw copy of a fast-path workspace buffer of the given precision
//   which: 0 stats(float2[M]) 1 qk hi 2 qk lo 3 vT hi 4 vT lo 5 y hi 6 y lo 7 hidden hi 8 hidden lo
int gpt_fast_debug_copy(mgpt_gpt *g, int precision, int which, void *d_out, int64_t nbytes, hipStream_t s)
{
    FastState *f = static_cast<FastState *>(g->fast);
    MGPT_REQUIRE(f && precision >= 1 && precision <= 2 && f->mode[precision].built, MGPT_ERR_STATE, "precision %d not built", precision);
    ModeState *m = &f->mode[precision];
    const void *src = nullptr;
    switch (which) {
        case 0: src = m->stats; break;
        case 1: src = m->qk[0]; break;
        case 2: src = m->qk[1]; break;
        case 3: src = m->vt[0]; break;
        case 4: src = m->vt[1]; break;
        case 5: src = m->y[0]; break;
        case 6: src = m->y[1]; break;
        case 7: src = m->hbuf[0]; break;
        case 8: src = m->hbuf[1]; break;
        default: break;
    }
    MGPT_REQUIRE(src, MGPT_ERR_ARG, "which=%d has no buffer in this mode", which);
    MGPT_HIP(hipMemcpyAsync(d_out, src, (size_t)nbytes, hipMemcpyDeviceToDevice, s));
    return MGPT_OK;
}

// test/debug: event counters of the fast-path kernels (include/mapf_gpt_amd.h)
extern "C" int mgpt_gpt_debug_counter(int which, uint64_t *value, int reset)
{
    MGPT_REQUIRE(value && which == 0, MGPT_ERR_ARG, "debug counter %d does not exist", which);
    MGPT_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    MGPT_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(mgpt::fastk::g_attn_fallbacks), sizeof(v)));
    *value = (uint64_t)v;
    if (reset) {
        const unsigned long long z = 0;
        MGPT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mgpt::fastk::g_attn_fallbacks), &z, sizeof(z)));
    }
    return MGPT_OK;
}
