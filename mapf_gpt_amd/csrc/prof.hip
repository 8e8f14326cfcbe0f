// prof.hip -- error string, ABI version, device count and the event-timing hooks.
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace mgpt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static const char *kProfNames[P_COUNT] = {
    "bfs_distance_field", "tok_update_agents", "tok_next_action", "tok_generate_observations",
    "env_step", "env_metrics", "gpt_embed", "gpt_layernorm", "gpt_gemm_qkv", "gpt_attention",
    "gpt_gemm_attn_proj", "gpt_gemm_mlp_fc", "gpt_gemm_mlp_proj", "gpt_mlp_fused", "gpt_head",
    "gpt_sample", "gpt_pack_weights", "gpt_ln_qkv_fused",
    "gpt_attention_last", "gpt_gemm_attn_proj_last", "gpt_mlp_fused_last"};

struct ProfState {
    std::mutex mu;
    bool on = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;       // in use
    std::vector<Rec> pool;       // reusable event pairs
    double total[P_COUNT] = {0};
    int64_t launches[P_COUNT] = {0};
};
static ProfState g_prof;
bool prof_is_enabled() { return g_prof.on; }
static const size_t kMaxRecs = 1u << 17;

ProfScope::ProfScope(ProfId id, hipStream_t stream) : slot(-1), s(stream)
{
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.recs.size() >= kMaxRecs) return;
    ProfState::Rec r;
    if (!g_prof.pool.empty()) {
        r = g_prof.pool.back();
        g_prof.pool.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    }
    r.id = id;
    (void)hipEventRecord(r.a, s);
    g_prof.recs.push_back(r);
    slot = (int)g_prof.recs.size() - 1;
}

ProfScope::~ProfScope()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    (void)hipEventRecord(g_prof.recs[slot].b, s);
}

static void prof_drain()   // caller holds the lock; device already synchronised
{
    for (auto &r : g_prof.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof.total[r.id] += ms;
            g_prof.launches[r.id] += 1;
        }
        g_prof.pool.push_back(r);
    }
    g_prof.recs.clear();
}

}  // namespace mgpt

using namespace mgpt;

extern "C" const char *mgpt_last_error(void) { return g_err; }

extern "C" int mgpt_abi_version(void) { return 1002; }

extern "C" int mgpt_device_count(int *count)
{
    MGPT_REQUIRE(count, MGPT_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return MGPT_OK;
}

extern "C" int mgpt_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return MGPT_OK;
}

extern "C" int mgpt_prof_reset(void)
{
    MGPT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain();
    for (int i = 0; i < P_COUNT; i++) { g_prof.total[i] = 0; g_prof.launches[i] = 0; }
    return MGPT_OK;
}

extern "C" int mgpt_prof_read(const char **names, float *total_ms, int64_t *launches, int *n)
{
    MGPT_REQUIRE(names && total_ms && launches && n, MGPT_ERR_ARG, "NULL argument");
    MGPT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain();
    int cap = *n, used = 0;
    for (int i = 0; i < P_COUNT && used < cap; i++) {
        if (g_prof.launches[i] == 0) continue;
        names[used] = kProfNames[i];
        total_ms[used] = (float)g_prof.total[i];
        launches[used] = g_prof.launches[i];
        used++;
    }
    *n = used;
    return MGPT_OK;
}
