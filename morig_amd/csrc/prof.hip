// Status strings, device info and the live HIP-event accounting used by bench.py's `roofline`.
#include "common.h"
#include <mutex>
#include <vector>
#include <string.h>

namespace morig {

static int g_last_hip = 0;
void set_hip_error(hipError_t e) { g_last_hip = (int)e; }

struct Slot { hipEvent_t a, b; int kind; double flops, bytes; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Slot> g_slots;       // recorded this epoch
static std::vector<Slot> g_free;        // recycled event pairs
static double g_ms[K_COUNT], g_fl[K_COUNT], g_by[K_COUNT];
static long long g_n[K_COUNT];

static thread_local ProfScope* t_scope = nullptr;

ProfScope::ProfScope(int kind, hipStream_t s, double flops, double bytes) : slot(-1), stream(s), outer(t_scope) {
    t_scope = this;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    Slot sl;
    if (!g_free.empty()) { sl = g_free.back(); g_free.pop_back(); }
    else {
        if (hipEventCreate(&sl.a) != hipSuccess) return;
        if (hipEventCreate(&sl.b) != hipSuccess) { (void)hipEventDestroy(sl.a); return; }
    }
    sl.kind = kind; sl.flops = flops; sl.bytes = bytes;
    (void)hipEventRecord(sl.a, s);
    g_slots.push_back(sl);
    slot = (int)g_slots.size() - 1;
}
ProfScope::~ProfScope() {
    t_scope = outer;
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < (int)g_slots.size()) (void)hipEventRecord(g_slots[slot].b, stream);
}

void prof_retag(int kind) {
    ProfScope* p = t_scope;
    if (!p || p->slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (p->slot < (int)g_slots.size()) g_slots[p->slot].kind = kind;
}

static void drain_locked() {
    for (auto& s : g_slots) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            g_ms[s.kind] += ms; g_fl[s.kind] += s.flops; g_by[s.kind] += s.bytes; g_n[s.kind] += 1;
        }
        g_free.push_back(s);
    }
    g_slots.clear();
}

static const char* kNames[K_COUNT] = {
    "gemm_f32_bn128", "gemm_f32_bn64", "gemm_f32_bn32", "gemm_f32_pool",
    "edgeconv_h16", "edgeconv_h32", "edgeconv_h64", "edgeconv_h128", "edgeconv_h256",
    "csr_build", "copy", "rownorm", "cls_attention", "misc",
    "fps", "ball_query", "pointconv", "knn_interpolate", "cosine_nn",
    "gemm_f16x3_bn128", "gemm_f16x3_bn64", "gemm_f16x3_bn32", "gemm_f16x3_pool",
    "edgeconv_f16x3_h32", "edgeconv_f16x3_h64", "edgeconv_f16x3_h128", "edgeconv_f16x3_h256", "pointconv_f16x3", "gemm_f16x3_dma",
    "cosine_knn", "flow_vote", "joint_extraction",
    "gemm_f16x3_dmap", "gemm_f16x3_dma128", "edgeconv_f16x3_h256_pp", "edgeconv_f16x3_h128_ws", "edgeconv_f16x3_pc", "geo_graph", "edgeconv_f16x3_x3", "edgeconv_x3", "edgeconv_f16x3_x3_persistent", "edgeconv_f16x3_h128_rl",
};
// kinds whose launches all run ONE kernel: the symbol as rocprofv3 prints it (prefix up to the template arguments that matter: the tile
// engine's sixth argument -- the guard-free FAST form of a dense store GEMM -- is chosen per launch from the shape)
static const char* kSymbols[K_COUNT] = {
    "tile_kernel<128, 32, 0, 0, 0,", "tile_kernel<64, 32, 0, 0, 0,", "tile_kernel<32, 32, 0, 0, 0,", "tile_kernel<128, 32, 0, 1, 0,",
    "tile_kernel<32, 16, 1, 2, 0,", "tile_kernel<32, 32, 1, 2, 0,", "tile_kernel<64, 32, 1, 2, 0,", "tile_kernel<128, 32, 1, 2, 0,", "tile_kernel<256, 16, 1, 2, 0,",
    nullptr, nullptr, "rownorm", "cls_attention_kernel", nullptr,
    nullptr, nullptr, nullptr, nullptr, nullptr,
    "tile_kernel<128, 32, 0, 0, 1,", "tile_kernel<64, 32, 0, 0, 1,", "tile_kernel<32, 32, 0, 0, 1,", "gemm16_dmap_kernel<true>",
    "tile_kernel<32, 32, 1, 2, 1,", "tile_kernel<64, 32, 1, 2, 1,", "edge_pp_kernel<128", "edge_ws_kernel<256", nullptr, "gemm16_dma_kernel<256, 256, 4, 2",
    nullptr, nullptr, nullptr,
    "gemm16_dmap_kernel<false>", "gemm16_dma_kernel<128, 128, 2, 2", "edge_pp_kernel<256", "edge_ws_kernel<128", "edge_pc_kernel", "geo_ball_graph_kernel", "tile_kernel<32, 32, 2, 2, 1,", "tile_kernel<32, 32, 2, 2, 0,", "edge_x3_kernel", "edge_rl128_kernel",
};

}  // namespace morig

using namespace morig;

extern "C" {

int morig_abi_version(void) { return MORIG_ABI_VERSION; }

int morig_reserve_cus(int n) { set_reserved_cus(n); return MORIG_OK; }

const char* morig_strerror(int st) {
    switch (st) {
        case MORIG_OK: return "ok";
        case MORIG_E_INVALID: return "invalid argument (null pointer, negative size or misaligned leading dimension)";
        case MORIG_E_UNSUPPORTED: return "unsupported width/shape for the instantiated gfx950 kernels";
        case MORIG_E_HIP: return "HIP runtime error (see morig_last_hip_error)";
        case MORIG_E_NODEVICE: return "no gfx950 device visible";
        default: return "unknown status";
    }
}

int morig_last_hip_error(void) { return g_last_hip; }

int morig_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* arch, int arch_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MORIG_E_NODEVICE;
    int dev = 0;
    MORIG_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    MORIG_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (clock_khz) *clock_khz = p.clockRate;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? MORIG_OK : MORIG_E_NODEVICE;
}

int morig_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    int prev = g_on ? 1 : 0;
    g_on = on != 0;
    return prev;
}

int morig_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    for (int k = 0; k < K_COUNT; ++k) { g_ms[k] = g_fl[k] = g_by[k] = 0.0; g_n[k] = 0; }
    return MORIG_OK;
}

const char* morig_prof_name(int kind) { return (kind >= 0 && kind < K_COUNT) ? kNames[kind] : nullptr; }
const char* morig_prof_symbol(int kind) { return (kind >= 0 && kind < K_COUNT) ? kSymbols[kind] : nullptr; }

int morig_prof_collect(int kind, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    if (kind < 0 || kind >= K_COUNT) return MORIG_E_INVALID;
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    if (launches) *launches = g_n[kind];
    if (total_ms) *total_ms = g_ms[kind];
    if (flops) *flops = g_fl[kind];
    if (bytes) *bytes = g_by[kind];
    return MORIG_OK;
}

}  // extern "C"
