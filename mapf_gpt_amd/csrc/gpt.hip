// gpt.hip -- policy context, parameter store, forward drivers and the action sampler
// (replaces mapf_gpt/model.py GPT.forward / GPT.act for inference).
#include <string>

#include "common.h"
#include "gpt_ctx.h"
#include "gpt_kernels_f32.h"

using namespace mgpt;

uint64_t mgpt::gpt_generation(const mgpt_gpt *g) { return g->generation; }

namespace {

constexpr int kT = 256;
constexpr int kV = MGPT_VOCAB;

// ----- action sampling, model.py:250-259 -----
// Counter-based RNG: splitmix64 finaliser over (seed, step, row) -> 24-bit uniform.  The same
// arithmetic is restated in mapf_gpt_amd/sampling.py for host-side verification.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t step, uint64_t row)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (step + 1ull);
    z ^= row * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void sample_kernel(const float *__restrict__ logits, int rows, int32_t *__restrict__ actions,
                                                     int do_sample, uint64_t seed, uint64_t step, uint64_t row0,
                                                     const uint64_t *__restrict__ d_step)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    if (d_step) step = *d_step;                  // step counter kept on the device (mgpt_step_run's graph)
    const float *l = logits + (size_t)row * kV;
    float v[MGPT_NUM_ACTIONS];
    float mx = -INFINITY;
    int best = 0;
#pragma unroll
    for (int i = 0; i < MGPT_NUM_ACTIONS; i++) {
        v[i] = l[i];
        if (v[i] > mx) { mx = v[i]; best = i; }     // first maximum, as torch.topk(k=1) on ties
    }
    if (!do_sample) { actions[row] = best; return; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MGPT_NUM_ACTIONS; i++) { v[i] = expf(v[i] - mx); sum += v[i]; }
    const float u = uniform01(seed, step, row0 + (uint64_t)row) * sum;
    float c = 0.f;
    int a = MGPT_NUM_ACTIONS - 1;
#pragma unroll
    for (int i = 0; i < MGPT_NUM_ACTIONS; i++) {
        c += v[i];
        if (u < c) { a = i; break; }
    }
    actions[row] = a;
}

}  // namespace

static size_t n_tensors(const mgpt_gpt *g) { return 3 + (size_t)g->L * 6; }

extern "C" int mgpt_gpt_create(mgpt_gpt **out, int n_layer, int n_head, int n_embd, int block_size, int max_rows)
{
    MGPT_REQUIRE(out, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(n_layer > 0 && n_head > 0 && n_embd > 0 && max_rows > 0, MGPT_ERR_ARG, "bad sizes");
    // block_size 256 in every released config (config-*.py:13); a shorter one is a model whose rows are T <= block_size tokens: mgpt_gpt_forward_t
    MGPT_REQUIRE(block_size >= 1 && block_size <= kT, MGPT_ERR_UNSUPPORTED, "block_size must be in 1 .. 256, got %d", block_size);
    MGPT_REQUIRE(n_embd % n_head == 0, MGPT_ERR_ARG, "n_embd %% n_head != 0 (model.py:27)");
    const int hs = n_embd / n_head;
    MGPT_REQUIRE(hs == 32 || hs == 64, MGPT_ERR_UNSUPPORTED, "head size %d: kernels exist for 32 and 64", hs);
    MGPT_REQUIRE(n_embd % 32 == 0 && n_embd <= 1024, MGPT_ERR_UNSUPPORTED, "n_embd must be a multiple of 32, <= 1024");
    mgpt_gpt *g = new mgpt_gpt();
    g->L = n_layer; g->nh = n_head; g->C = n_embd; g->hs = hs; g->block = block_size; g->max_rows = max_rows;
    const size_t C = n_embd;
    size_t off = 0;
    g->off_wte = off; off += (size_t)kV * C;
    g->off_wpe = off; off += (size_t)block_size * C;
    g->off_lnf = off; off += C;
    for (int l = 0; l < n_layer; l++) {
        LayerOff lo;
        lo.ln1 = off; off += C;
        lo.attn_w = off; off += 3 * C * C;
        lo.proj_w = off; off += C * C;
        lo.ln2 = off; off += C;
        lo.fc_w = off; off += 4 * C * C;
        lo.proj2_w = off; off += 4 * C * C;
        g->layers.push_back(lo);
    }
    g->n_params = off;
    g->is_set.assign(n_tensors(g), 0);
    {   // bias vectors of a bias = True checkpoint (allocated when the first one arrives)
        size_t bo = 0;
        g->off_lnf_b = bo; bo += C;
        for (int l = 0; l < n_layer; l++) {
            BiasOff b;
            b.ln1 = bo; bo += C;
            b.attn = bo; bo += 3 * C;
            b.proj = bo; bo += C;
            b.ln2 = bo; bo += C;
            b.fc = bo; bo += 4 * C;
            b.proj2 = bo; bo += C;
            g->bias_layers.push_back(b);
        }
        g->bias_set.assign(1 + (size_t)n_layer * 6, 0);
    }
    const size_t M = (size_t)max_rows * kT;
    hipError_t e = hipMalloc(&g->params, off * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&g->x, M * C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&g->xn, M * C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&g->qkv, 3 * M * C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&g->hbuf, 4 * M * C * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&g->logits_tmp, (size_t)max_rows * kV * sizeof(float));
    if (e != hipSuccess) {
        set_error("hipMalloc failed in mgpt_gpt_create: %s", hipGetErrorString(e));
        mgpt_gpt_destroy(g);
        return MGPT_ERR_HIP;
    }
    *out = g;
    return MGPT_OK;
}

extern "C" int mgpt_gpt_destroy(mgpt_gpt *g)
{
    if (!g) return MGPT_OK;
    gpt_fast_destroy(g);
    (void)hipFree(g->params); (void)hipFree(g->x); (void)hipFree(g->xn); (void)hipFree(g->qkv);
    (void)hipFree(g->hbuf); (void)hipFree(g->logits_tmp); (void)hipFree(g->bias);
    delete g;
    return MGPT_OK;
}

static size_t n_bias_elems(const mgpt_gpt *g) { return (size_t)g->C * (1 + 11 * (size_t)g->L); }

// reference state_dict key of a bias vector (GPTConfig.bias = True) -> (bias tensor index, offset into g->bias, element count)
static bool locate_bias(const mgpt_gpt *g, const char *name_in, size_t *idx, size_t *off, size_t *count)
{
    std::string name(name_in);
    const std::string pre = "_orig_mod.";
    if (name.compare(0, pre.size(), pre) == 0) name = name.substr(pre.size());
    const size_t C = g->C;
    if (name == "transformer.ln_f.bias") { *idx = 0; *off = g->off_lnf_b; *count = C; return true; }
    const std::string hp = "transformer.h.";
    if (name.compare(0, hp.size(), hp) != 0) return false;
    size_t p = hp.size(), l = 0;
    if (p >= name.size() || name[p] < '0' || name[p] > '9') return false;
    while (p < name.size() && name[p] >= '0' && name[p] <= '9') { l = l * 10 + (size_t)(name[p] - '0'); p++; }
    if (l >= (size_t)g->L || p >= name.size() || name[p] != '.') return false;
    const std::string rest = name.substr(p + 1);
    const BiasOff &b = g->bias_layers[l];
    const size_t base = 1 + l * 6;
    if (rest == "ln_1.bias") { *idx = base + 0; *off = b.ln1; *count = C; return true; }
    if (rest == "attn.c_attn.bias") { *idx = base + 1; *off = b.attn; *count = 3 * C; return true; }
    if (rest == "attn.c_proj.bias") { *idx = base + 2; *off = b.proj; *count = C; return true; }
    if (rest == "ln_2.bias") { *idx = base + 3; *off = b.ln2; *count = C; return true; }
    if (rest == "mlp.c_fc.bias") { *idx = base + 4; *off = b.fc; *count = 4 * C; return true; }
    if (rest == "mlp.c_proj.bias") { *idx = base + 5; *off = b.proj2; *count = C; return true; }
    return false;
}

// reference state_dict key -> (tensor index, offset, element count)
static bool locate_param(const mgpt_gpt *g, const char *name_in, size_t *idx, size_t *off, size_t *count)
{
    std::string name(name_in);
    const std::string pre = "_orig_mod.";                       // inference.py:33-44
    if (name.compare(0, pre.size(), pre) == 0) name = name.substr(pre.size());
    const size_t C = g->C;
    if (name == "transformer.wte.weight" || name == "lm_head.weight") { *idx = 0; *off = g->off_wte; *count = kV * C; return true; }
    if (name == "transformer.wpe.weight") { *idx = 1; *off = g->off_wpe; *count = (size_t)g->block * C; return true; }
    if (name == "transformer.ln_f.weight") { *idx = 2; *off = g->off_lnf; *count = C; return true; }
    const std::string hp = "transformer.h.";
    if (name.compare(0, hp.size(), hp) != 0) return false;
    size_t p = hp.size(), l = 0;
    if (p >= name.size() || name[p] < '0' || name[p] > '9') return false;
    while (p < name.size() && name[p] >= '0' && name[p] <= '9') { l = l * 10 + (size_t)(name[p] - '0'); p++; }
    if (l >= (size_t)g->L || p >= name.size() || name[p] != '.') return false;
    const std::string rest = name.substr(p + 1);
    const LayerOff &lo = g->layers[l];
    const size_t base = 3 + l * 6;
    if (rest == "ln_1.weight") { *idx = base + 0; *off = lo.ln1; *count = C; return true; }
    if (rest == "attn.c_attn.weight") { *idx = base + 1; *off = lo.attn_w; *count = 3 * C * C; return true; }
    if (rest == "attn.c_proj.weight") { *idx = base + 2; *off = lo.proj_w; *count = C * C; return true; }
    if (rest == "ln_2.weight") { *idx = base + 3; *off = lo.ln2; *count = C; return true; }
    if (rest == "mlp.c_fc.weight") { *idx = base + 4; *off = lo.fc_w; *count = 4 * C * C; return true; }
    if (rest == "mlp.c_proj.weight") { *idx = base + 5; *off = lo.proj2_w; *count = 4 * C * C; return true; }
    return false;
}

extern "C" int mgpt_gpt_set_param(mgpt_gpt *g, const char *name, const float *data, int64_t n_elem, int is_device)
{
    MGPT_REQUIRE(g && name && data, MGPT_ERR_ARG, "NULL argument");
    size_t idx, off, count;
    if (locate_bias(g, name, &idx, &off, &count)) {        // a bias = True checkpoint (model.py:115): see mgpt_gpt_forward for which kernels carry it
        MGPT_REQUIRE((size_t)n_elem == count, MGPT_ERR_ARG, "parameter '%s': got %lld elements, expected %zu", name, (long long)n_elem, count);
        if (!g->bias) {
            MGPT_HIP(hipMalloc(&g->bias, n_bias_elems(g) * sizeof(float)));
            MGPT_HIP(hipMemset(g->bias, 0, n_bias_elems(g) * sizeof(float)));
        }
        MGPT_HIP(hipMemcpy(g->bias + off, data, count * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        g->bias_set[idx] = 1;
        g->has_bias = true;
        g->finalized = false;
        return MGPT_OK;
    }
    MGPT_REQUIRE(locate_param(g, name, &idx, &off, &count), MGPT_ERR_ARG, "unknown parameter '%s'", name);
    MGPT_REQUIRE((size_t)n_elem == count, MGPT_ERR_ARG, "parameter '%s': got %lld elements, expected %zu", name,
                 (long long)n_elem, count);
    MGPT_HIP(hipMemcpy(g->params + off, data, count * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    g->is_set[idx] = 1;
    g->finalized = false;
    return MGPT_OK;
}

// ----- precision envelope of the split-fp16 mode (include/mapf_gpt_amd.h: MGPT_ENVELOPE_*) -----
// max |w| and sum of squares of one matrix: block-level reduction, then one atomic per block (max on the bit pattern of a non-negative float)
__global__ __launch_bounds__(256) void wstats_kernel(const float *__restrict__ w, int64_t n, unsigned *__restrict__ max_bits, float *__restrict__ sumsq)
{
    float mx = 0.f, sq = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = w[i];
        mx = fmaxf(mx, fabsf(v));
        sq = fmaf(v, v, sq);
    }
    __shared__ float smx[256], ssq[256];
    smx[threadIdx.x] = mx; ssq[threadIdx.x] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]); ssq[threadIdx.x] += ssq[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicMax(max_bits, __float_as_uint(smx[0])); atomicAdd(sumsq, ssq[0]); }
}

static int envelope_stats(mgpt_gpt *g)
{
    const size_t C = g->C;
    float *d = nullptr;                                    // [0] max bits, [1] sum of squares
    MGPT_HIP(hipMalloc(&d, 2 * sizeof(float)));
    g->env_max_w = 0.f; g->env_max_rms = 0.f;
    int rc = MGPT_OK;
    for (int l = 0; l < g->L && rc == MGPT_OK; l++) {
        const LayerOff &lo = g->layers[l];
        const size_t offs[4] = {lo.attn_w, lo.proj_w, lo.fc_w, lo.proj2_w}, cnt[4] = {3 * C * C, C * C, 4 * C * C, 4 * C * C};
        for (int k = 0; k < 4; k++) {
            if (hipMemset(d, 0, 2 * sizeof(float)) != hipSuccess) { rc = MGPT_ERR_HIP; break; }
            hipLaunchKernelGGL(wstats_kernel, dim3(64), dim3(256), 0, nullptr, g->params + offs[k], (int64_t)cnt[k], reinterpret_cast<unsigned *>(d), d + 1);
            float h[2];
            if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { rc = MGPT_ERR_HIP; break; }
            g->env_max_w = std::max(g->env_max_w, h[0]);
            g->env_max_rms = std::max(g->env_max_rms, sqrtf(h[1] / (float)cnt[k]));
        }
    }
    (void)hipFree(d);
    if (rc != MGPT_OK) set_error("envelope statistics failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

extern "C" int mgpt_gpt_finalize(mgpt_gpt *g)
{
    MGPT_REQUIRE(g, MGPT_ERR_ARG, "NULL argument");
    for (size_t i = 0; i < g->is_set.size(); i++)
        MGPT_REQUIRE(g->is_set[i], MGPT_ERR_STATE, "parameter tensor #%zu was never set (see mgpt_gpt_set_param)", i);
    if (g->has_bias)                                       // nn.Linear(bias=True) / LayerNorm(bias=True) come together (model.py:126-131)
        for (size_t i = 0; i < g->bias_set.size(); i++)
            MGPT_REQUIRE(g->bias_set[i], MGPT_ERR_STATE, "the checkpoint has bias vectors, but bias tensor #%zu (0 = ln_f, then six per layer) was never set", i);
    int rc = MGPT_OK;
    if (g->block == kT && (rc = gpt_fast_finalize(g)) != MGPT_OK) return rc;      // (the 16-bit kernels and their operand planes exist for 256-token rows only)
    if ((rc = envelope_stats(g)) != MGPT_OK) return rc;
    // (bias vectors: outside by construction -- the 16-bit kernels have no bias terms, MGPT_PREC_F16X3 requests follow the envelope policy)
    g->env_state = g->has_bias ? 2 : 0; g->env_probe_err = g->env_probe_err_small = g->env_probe_err_large = -1.f; g->env_logged = false;
    g->env_probe_tol = 0.f; g->env_probe_max_logit = 0.f;
    g->finalized = true;
    return MGPT_OK;
}

extern "C" int mgpt_gpt_set_envelope_policy(mgpt_gpt *g, int policy)
{
    MGPT_REQUIRE(g && policy >= MGPT_ENVELOPE_FALLBACK && policy <= MGPT_ENVELOPE_IGNORE, MGPT_ERR_ARG, "envelope policy %d", policy);
    g->env_policy = policy;
    return MGPT_OK;
}

extern "C" int mgpt_gpt_envelope(mgpt_gpt *g, float *out3, int *state)
{
    MGPT_REQUIRE(g && out3 && state, MGPT_ERR_ARG, "NULL argument");
    out3[0] = g->env_max_w; out3[1] = g->env_max_rms; out3[2] = g->env_probe_err;
    *state = g->env_state;
    return MGPT_OK;
}

extern "C" int mgpt_gpt_envelope_probe(mgpt_gpt *g, float *out4)
{
    MGPT_REQUIRE(g && out4, MGPT_ERR_ARG, "NULL argument");
    out4[0] = g->env_probe_err_small; out4[1] = g->env_probe_err_large; out4[2] = g->env_probe_tol; out4[3] = g->env_probe_max_logit;
    return MGPT_OK;
}

int gpt_launch_head_at(mgpt_gpt *g, const float *xsrc, int64_t row_stride, int64_t row_offset, int rows, float *d_logits,
                       hipStream_t s)
{
    ProfScope ps(P_HEAD, s);
    hipLaunchKernelGGL(f32k::head_kernel, dim3(rows), dim3(64), (size_t)g->C * sizeof(float), s, xsrc, g->params + g->off_lnf,
                       g->params + g->off_wte, d_logits, g->C, kV, row_stride, row_offset,
                       g->has_bias ? (const float *)(g->bias + g->off_lnf_b) : (const float *)nullptr);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

int gpt_launch_head(mgpt_gpt *g, int rows, float *d_logits, hipStream_t s)
{
    return gpt_launch_head_at(g, g->x, (int64_t)kT * g->C, (int64_t)(kT - 1) * g->C, rows, d_logits, s);
}

// ----- fp32 forward -----
template <int EPI>
static int launch_gemm(const float *A, const float *W, float *out, int64_t M, int N, int K, f32k::EpiArgs ep, int C,
                       hipStream_t s)
{
    MGPT_REQUIRE(M % 128 == 0 && K % 32 == 0, MGPT_ERR_UNSUPPORTED, "gemm shape M=%lld K=%d", (long long)M, K);
    const int mt = (int)(M / 128);
    if (C == 160 && N % 160 == 0) {
        const int ntn = N / 160;
        hipLaunchKernelGGL((f32k::gemm_f32_kernel<160, 4, 1, EPI>), dim3(mt * ntn), dim3(256), 0, s, A, W, out, (int)M, N, K, ntn, ep);
    } else if (N % 128 == 0) {
        const int ntn = N / 128;
        hipLaunchKernelGGL((f32k::gemm_f32_kernel<128, 2, 2, EPI>), dim3(mt * ntn), dim3(256), 0, s, A, W, out, (int)M, N, K, ntn, ep);
    } else if (N % 64 == 0) {
        const int ntn = N / 64;
        hipLaunchKernelGGL((f32k::gemm_f32_kernel<64, 4, 1, EPI>), dim3(mt * ntn), dim3(256), 0, s, A, W, out, (int)M, N, K, ntn, ep);
    } else {
        set_error("gemm N=%d is not a multiple of 64", N);
        return MGPT_ERR_UNSUPPORTED;
    }
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

static int launch_layernorm(const float *x, const float *w, const float *b, float *y, int64_t n_tok, int C, hipStream_t s)
{
    ProfScope ps(P_LAYERNORM, s);
    const dim3 grid((unsigned)cdiv64(n_tok, 4));
    if (C <= 256) hipLaunchKernelGGL((f32k::layernorm_kernel<1>), grid, dim3(256), 0, s, x, w, y, n_tok, C, (int64_t)C, (int64_t)C, b);
    else if (C <= 512) hipLaunchKernelGGL((f32k::layernorm_kernel<2>), grid, dim3(256), 0, s, x, w, y, n_tok, C, (int64_t)C, (int64_t)C, b);
    else if (C <= 768) hipLaunchKernelGGL((f32k::layernorm_kernel<3>), grid, dim3(256), 0, s, x, w, y, n_tok, C, (int64_t)C, (int64_t)C, b);
    else hipLaunchKernelGGL((f32k::layernorm_kernel<4>), grid, dim3(256), 0, s, x, w, y, n_tok, C, (int64_t)C, (int64_t)C, b);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

// T = tokens per row (kT on the hot path; mgpt_gpt_forward_t: any T <= block_size).  M = rows * T tokens; the GEMMs run on Mp = M rounded up to
// their 128-token tile -- the padding rows of the workspaces hold whatever an earlier call left there, every token's arithmetic is its own, and
// the one kernel that mixes tokens (attention) and the q|k|v scatter look at the first M only
static int forward_f32_chunk(mgpt_gpt *g, const uint8_t *d_tokens, int rows, float *d_logits, hipStream_t s, int T = kT)
{
    const int C = g->C;
    const int64_t M = (int64_t)rows * T, Mp = (M + 127) / 128 * 128;
    const float *P = g->params;
    int rc;
    {
        ProfScope ps(P_EMBED, s);
        const int64_t total = M * (C / 4);
        const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 64);
        hipLaunchKernelGGL(f32k::embed_kernel, dim3(blocks), dim3(256), 0, s, d_tokens, P + g->off_wte, P + g->off_wpe, g->x, M, C, T);
        MGPT_LAUNCH_CHECK();
    }
    f32k::EpiArgs ep;
    ep.C = C; ep.n_head = g->nh; ep.hs = g->hs; ep.plane = M * C; ep.T = T; ep.m_valid = M;
    const float scale = 1.0f / sqrtf((float)g->hs);
    float *q = g->qkv, *k = g->qkv + M * C, *v = g->qkv + 2 * M * C;
    const float *Bv = g->has_bias ? g->bias : nullptr;       // bias = True checkpoints (model.py:14-17,29,31,79,81)
    auto bias_at = [&](size_t off) -> const float * { return Bv ? Bv + off : nullptr; };
    for (int l = 0; l < g->L; l++) {
        const LayerOff &lo = g->layers[l];
        const BiasOff &bo = g->bias_layers[l];
        if ((rc = launch_layernorm(g->x, P + lo.ln1, bias_at(bo.ln1), g->xn, M, C, s)) != MGPT_OK) return rc;
        {
            ProfScope ps(P_GEMM_QKV, s);
            ep.bias = bias_at(bo.attn);
            if ((rc = launch_gemm<f32k::EPI_QKV>(g->xn, P + lo.attn_w, g->qkv, Mp, 3 * C, C, ep, C, s)) != MGPT_OK) return rc;
        }
        {
            ProfScope ps(P_ATTN, s);
            if (g->hs == 32) hipLaunchKernelGGL((f32k::attn_f32_kernel<32>), dim3(rows * g->nh), dim3(256), 0, s, q, k, v, g->xn, g->nh, scale, T);
            else hipLaunchKernelGGL((f32k::attn_f32_kernel<64>), dim3(rows * g->nh), dim3(256), 0, s, q, k, v, g->xn, g->nh, scale, T);
            MGPT_LAUNCH_CHECK();
        }
        {
            ProfScope ps(P_GEMM_PROJ, s);
            ep.bias = bias_at(bo.proj);
            if ((rc = launch_gemm<f32k::EPI_RESID>(g->xn, P + lo.proj_w, g->x, Mp, C, C, ep, C, s)) != MGPT_OK) return rc;
        }
        if ((rc = launch_layernorm(g->x, P + lo.ln2, bias_at(bo.ln2), g->xn, M, C, s)) != MGPT_OK) return rc;
        {
            ProfScope ps(P_GEMM_FC, s);
            ep.bias = bias_at(bo.fc);
            if ((rc = launch_gemm<f32k::EPI_GELU>(g->xn, P + lo.fc_w, g->hbuf, Mp, 4 * C, C, ep, C, s)) != MGPT_OK) return rc;
        }
        {
            ProfScope ps(P_GEMM_PROJ2, s);
            ep.bias = bias_at(bo.proj2);
            if ((rc = launch_gemm<f32k::EPI_RESID>(g->hbuf, P + lo.proj2_w, g->x, Mp, C, 4 * C, ep, C, s)) != MGPT_OK) return rc;
        }
    }
    if (T == kT) return gpt_launch_head(g, rows, d_logits, s);
    return gpt_launch_head_at(g, g->x, (int64_t)T * C, (int64_t)(T - 1) * C, rows, d_logits, s);      // the last position of a T-token row (model.py:186)
}

// debugging aid for the GPU parity tests: copy an fp32-path workspace buffer out after a forward
// which: 0 = x (residual stream), 1 = xn (last LayerNorm output), 2 = qkv planes, 3 = MLP hidden
extern "C" int mgpt_gpt_debug_copy(mgpt_gpt *g, int which, float *d_out, int64_t n_elem, void *stream)
{
    MGPT_REQUIRE(g && d_out && n_elem > 0, MGPT_ERR_ARG, "bad argument");
    const int64_t M = (int64_t)g->max_rows * kT, C = g->C;
    const float *src = which == 0 ? g->x : which == 1 ? g->xn : which == 2 ? g->qkv : which == 3 ? g->hbuf : nullptr;
    const int64_t cap = which == 2 ? 3 * M * C : which == 3 ? 4 * M * C : M * C;
    MGPT_REQUIRE(src && n_elem <= cap, MGPT_ERR_ARG, "which=%d n_elem=%lld", which, (long long)n_elem);
    MGPT_HIP(hipMemcpyAsync(d_out, src, (size_t)n_elem * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MGPT_OK;
}

extern "C" int mgpt_gpt_debug_copy_raw(mgpt_gpt *g, int precision, int which, void *d_out, int64_t nbytes, void *stream)
{
    MGPT_REQUIRE(g && d_out && nbytes > 0, MGPT_ERR_ARG, "bad argument");
    return gpt_fast_debug_copy(g, precision, which, d_out, nbytes, (hipStream_t)stream);
}

// Decides once per finalized checkpoint whether MGPT_PREC_F16X3 requests are served by the split path: the weight statistics of
// mgpt_gpt_finalize against the validated range, then MGPT_ENVELOPE_PROBE_ROWS fixed pseudo-random rows through both paths.
// (Synchronises the stream: the first split-mode forward of a checkpoint must not sit inside a stream capture -- mgpt_step_run's
// first step is eager for this and other first-use work.)
static int envelope_decide(mgpt_gpt *g, hipStream_t s)
{
    const int n = std::min(g->max_rows, MGPT_ENVELOPE_PROBE_ROWS);
    bool inside = g->env_max_w <= MGPT_ENVELOPE_MAX_W && g->env_max_rms <= MGPT_ENVELOPE_MAX_RMS;
    {   // the probe allocates, copies back and synchronises: it cannot run inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("the first MGPT_PREC_F16X3 forward of a checkpoint probes its precision envelope and cannot be captured: "
                      "run one eager forward first (or mgpt_gpt_set_envelope_policy(MGPT_ENVELOPE_IGNORE))");
            return MGPT_ERR_STATE;
        }
    }
    if (inside) {
        std::vector<uint8_t> h_tok((size_t)n * kT);
        uint64_t st = 0x9e3779b97f4a7c15ull;
        for (auto &t : h_tok) { st = st * 6364136223846793005ull + 1442695040888963407ull; t = (uint8_t)((st >> 33) % (uint64_t)kV); }
        uint8_t *d_tok = nullptr; float *d_lg = nullptr;
        MGPT_HIP(hipMalloc(&d_tok, h_tok.size()));
        if (hipMalloc(&d_lg, (size_t)3 * n * kV * sizeof(float)) != hipSuccess) { (void)hipFree(d_tok); set_error("hipMalloc failed in the envelope probe"); return MGPT_ERR_HIP; }
        int rc = MGPT_OK;
        // BOTH call regimes: the same rows once as a call of n rows (the small-call kernels: head-parallel attention, 32 x 32 x 16 MLP
        // block, fp32 last layer) and once as a chunk of a call of more than kSmallRows rows (attn256q / attn160o, mlp256q /
        // mlp_fused16, gemm_pk16, attn_last1: other arithmetic) -- ADVICE r05: the probe used to see the small-call kernels only
        std::vector<float> h_lg((size_t)3 * n * kV);
        if (hipMemcpyAsync(d_tok, h_tok.data(), h_tok.size(), hipMemcpyHostToDevice, s) != hipSuccess) rc = MGPT_ERR_HIP;
        if (rc == MGPT_OK) rc = forward_f32_chunk(g, d_tok, n, d_lg, s);
        if (rc == MGPT_OK) rc = gpt_fast_forward(g, d_tok, n, d_lg + (size_t)n * kV, MGPT_PREC_F16X3, s, n);
        if (rc == MGPT_OK) rc = gpt_fast_forward(g, d_tok, n, d_lg + (size_t)2 * n * kV, MGPT_PREC_F16X3, s, kSmallRows + 1);
        if (rc == MGPT_OK && hipMemcpyAsync(h_lg.data(), d_lg, h_lg.size() * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) rc = MGPT_ERR_HIP;
        if (rc == MGPT_OK && hipStreamSynchronize(s) != hipSuccess) rc = MGPT_ERR_HIP;
        (void)hipFree(d_tok); (void)hipFree(d_lg);
        if (rc != MGPT_OK) { if (rc == MGPT_ERR_HIP) set_error("envelope probe failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
        float err[2] = {0.f, 0.f}, max_logit = 0.f;
        for (size_t i = 0; i < (size_t)n * kV; i++) {
            max_logit = std::max(max_logit, fabsf(h_lg[i]));
            for (int r = 0; r < 2; r++) {
                const float d = fabsf(h_lg[i] - h_lg[(size_t)(r + 1) * n * kV + i]);
                err[r] = (d == d) ? std::max(err[r], d) : INFINITY;  // NaN counts as outside
            }
        }
        // The bar: MGPT_ENVELOPE_PROBE_TOL absolute while the logits are of order one (where the 1e-5 of the north star was
        // established); relative to the largest fp32 logit beyond that -- fp32 itself moves by ~eps * |logit| * sqrt(depth) between
        // summation orders (3.6e-5 at |logits| ~ 4), so an absolute bar would send every trained checkpoint with
        // logits of 10 to the fp32 kernels on rounding noise alone
        g->env_probe_err_small = err[0]; g->env_probe_err_large = err[1];
        g->env_probe_err = std::max(err[0], err[1]);
        g->env_probe_max_logit = (max_logit == max_logit) ? max_logit : INFINITY;
        g->env_probe_tol = std::max(MGPT_ENVELOPE_PROBE_TOL, MGPT_ENVELOPE_PROBE_REL * g->env_probe_max_logit);
        inside = g->env_probe_err <= g->env_probe_tol;
    }
    g->env_state = inside ? 1 : 2;
    return MGPT_OK;
}

// call_rows: rows of the C-ABI call these rows belong to (mgpt_gpt_act chunks its rows itself), see gpt_ctx.h
static int gpt_forward_impl(mgpt_gpt *g, const uint8_t *d_tokens, int rows, float *d_logits, int precision, void *stream, int call_rows)
{
    MGPT_REQUIRE(g && d_tokens && d_logits, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(rows > 0, MGPT_ERR_ARG, "rows=%d", rows);
    MGPT_REQUIRE(g->finalized, MGPT_ERR_STATE, "mgpt_gpt_finalize must precede forward");
    MGPT_REQUIRE(g->block == kT, MGPT_ERR_UNSUPPORTED, "this entry point takes 256-token rows and the model's block_size is %d: mgpt_gpt_forward_t", g->block);
    hipStream_t s = (hipStream_t)stream;
    // bias = True checkpoints: only the exact-fp32 kernels carry the bias terms.  MGPT_PREC_F16X3 under the fallback policy is served by them
    // (the checkpoint counts as outside the envelope, mgpt_gpt_finalize); every other 16-bit request is refused -- it would be wrong, not imprecise
    MGPT_REQUIRE(!g->has_bias || precision == MGPT_PREC_F32 || (precision == MGPT_PREC_F16X3 && g->env_policy == MGPT_ENVELOPE_FALLBACK),
                 MGPT_ERR_UNSUPPORTED, "the checkpoint has Linear / LayerNorm bias vectors (GPTConfig.bias = True): they are carried by the MGPT_PREC_F32 kernels only");
    if (precision == MGPT_PREC_F16X3 && g->env_policy != MGPT_ENVELOPE_IGNORE) {
        if (g->env_state == 0) {
            const int rc = envelope_decide(g, s);
            if (rc != MGPT_OK) return rc;
        }
        if (g->env_state == 2) {
            char probe[160];
            if (g->has_bias) snprintf(probe, sizeof(probe), "not probed: the checkpoint has bias vectors, which the 16-bit kernels do not carry");
            else if (g->env_probe_err < 0.f) snprintf(probe, sizeof(probe), "not probed: the weight statistics decide");
            else snprintf(probe, sizeof(probe), "probe |f16x3 - f32| %.3g (small calls) / %.3g (large calls) of %.3g at max |logit| %.3g",
                          g->env_probe_err_small, g->env_probe_err_large, g->env_probe_tol, g->env_probe_max_logit);
            MGPT_REQUIRE(g->env_policy != MGPT_ENVELOPE_REFUSE, MGPT_ERR_UNSUPPORTED,
                         "checkpoint outside the validated f16x3 envelope (max|w| %.3g, rms %.3g; %s)", g->env_max_w, g->env_max_rms, probe);
            if (!g->env_logged) {
                fprintf(stderr, "mapf_gpt_amd: checkpoint outside the validated f16x3 envelope (max|w| %.3g of %.3g, rms %.3g of %.3g; %s): "
                                "MGPT_PREC_F16X3 requests run the exact-fp32 kernels (mgpt_gpt_set_envelope_policy to refuse or to ignore)\n",
                        g->env_max_w, MGPT_ENVELOPE_MAX_W, g->env_max_rms, MGPT_ENVELOPE_MAX_RMS, probe);
                g->env_logged = true;
            }
            precision = MGPT_PREC_F32;
        }
    }
    for (int r0 = 0; r0 < rows; r0 += g->max_rows) {
        const int n = std::min(g->max_rows, rows - r0);
        int rc;
        if (precision == MGPT_PREC_F32) rc = forward_f32_chunk(g, d_tokens + (size_t)r0 * kT, n, d_logits + (size_t)r0 * kV, s);
        else rc = gpt_fast_forward(g, d_tokens + (size_t)r0 * kT, n, d_logits + (size_t)r0 * kV, precision, s, call_rows);
        if (rc != MGPT_OK) return rc;
    }
    return MGPT_OK;
}

extern "C" int mgpt_gpt_forward(mgpt_gpt *g, const uint8_t *d_tokens, int rows, float *d_logits, int precision, void *stream)
{
    return gpt_forward_impl(g, d_tokens, rows, d_logits, precision, stream, rows);
}

// = GPT.forward(idx) for idx of T <= block_size tokens per row (model.py:167-175: positions 0 .. T - 1, attention over the T tokens, logits of position
// T - 1).  The hot path never does this -- the tokenizer emits 256-token rows, inference.py:145 -- so it is served by the exact-fp32 kernels alone.
extern "C" int mgpt_gpt_forward_t(mgpt_gpt *g, const uint8_t *d_tokens, int rows, int T, float *d_logits, void *stream)
{
    MGPT_REQUIRE(g && d_tokens && d_logits, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(rows > 0, MGPT_ERR_ARG, "rows=%d", rows);
    MGPT_REQUIRE(g->finalized, MGPT_ERR_STATE, "mgpt_gpt_finalize must precede forward");
    MGPT_REQUIRE(T >= 1 && T <= g->block, MGPT_ERR_ARG, "cannot forward sequence of length %d, block size is only %d (model.py:170)", T, g->block);
    hipStream_t s = (hipStream_t)stream;
    for (int r0 = 0; r0 < rows; r0 += g->max_rows) {
        const int n = std::min(g->max_rows, rows - r0);
        const int rc = forward_f32_chunk(g, d_tokens + (size_t)r0 * T, n, d_logits + (size_t)r0 * kV, s, T);
        if (rc != MGPT_OK) return rc;
    }
    return MGPT_OK;
}

extern "C" int mgpt_sample_actions(const float *d_logits, int rows, int32_t *d_actions, int do_sample, uint64_t seed,
                                   uint64_t step, uint64_t row0, void *stream)
{
    MGPT_REQUIRE(d_logits && d_actions && rows > 0, MGPT_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(P_SAMPLE, s);
    hipLaunchKernelGGL(sample_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, d_logits, rows, d_actions, do_sample, seed, step, row0, (const uint64_t *)nullptr);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

static int gpt_act_impl(mgpt_gpt *g, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits, int do_sample,
                        uint64_t seed, uint64_t step, const uint64_t *d_step, uint64_t row0, int precision, void *stream)
{
    MGPT_REQUIRE(g && d_tokens && d_actions, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(rows > 0, MGPT_ERR_ARG, "rows=%d", rows);
    MGPT_REQUIRE(g->finalized, MGPT_ERR_STATE, "mgpt_gpt_finalize must precede act");
    hipStream_t s = (hipStream_t)stream;
    // row index inside the RNG key is the GLOBAL row (row0 + row of this call), independent of workspace chunking and sharding
    for (int r0 = 0; r0 < rows; r0 += g->max_rows) {
        const int n = std::min(g->max_rows, rows - r0);
        float *lg = d_logits ? d_logits + (size_t)r0 * kV : g->logits_tmp;
        int rc = gpt_forward_impl(g, d_tokens + (size_t)r0 * kT, n, lg, precision, stream, rows);
        if (rc != MGPT_OK) return rc;
        ProfScope ps(P_SAMPLE, s);
        hipLaunchKernelGGL(sample_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, lg, n, d_actions + r0, do_sample,
                           seed, step, row0 + (uint64_t)r0, d_step);
        MGPT_LAUNCH_CHECK();
    }
    return MGPT_OK;
}

extern "C" int mgpt_gpt_act(mgpt_gpt *g, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits,
                            int do_sample, uint64_t seed, uint64_t step, uint64_t row0, int precision, void *stream)
{
    return gpt_act_impl(g, d_tokens, rows, d_actions, d_logits, do_sample, seed, step, nullptr, row0, precision, stream);
}

// the same with the RNG step counter read from device memory at run time (step.hip: a captured graph cannot carry a
// per-step scalar argument)
extern "C" int mgpt_gpt_act_dev(mgpt_gpt *g, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits, int do_sample,
                                uint64_t seed, const uint64_t *d_step, uint64_t row0, int precision, void *stream)
{
    MGPT_REQUIRE(d_step, MGPT_ERR_ARG, "NULL step counter");
    return gpt_act_impl(g, d_tokens, rows, d_actions, d_logits, do_sample, seed, 0, d_step, row0, precision, stream);
}
