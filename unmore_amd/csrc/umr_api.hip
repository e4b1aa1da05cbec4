// Host-side plumbing shared by every entry point: version + last-error text.
#include "umr_common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int umr_set_error(int code, const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

extern "C" int umr_version(void) { return 100; }
extern "C" const char* umr_last_error_string(void) { return g_err; }

// ---- f32 product mode (include/umr.h): process-wide, switchable at run time (tests A/B it within one process)
#include <atomic>
#include <stdlib.h>
static std::atomic<int> g_f32_mode{-1};
int umr_f32_mode_now() {
    int m = g_f32_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("UMR_F32_X3");
        m = (e && atoi(e) == 0) ? UMR_F32_EXACT : UMR_F32_X3;
        g_f32_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" int umr_set_f32_mode(int mode) {
    if (mode != UMR_F32_EXACT && mode != UMR_F32_X3 && mode != UMR_F32_X3_FAST) return umr_set_error(UMR_ERR_INVALID, "umr_set_f32_mode: unknown mode");
    g_f32_mode.store(mode, std::memory_order_relaxed);
    return UMR_OK;
}
extern "C" int umr_get_f32_mode(void) { return umr_f32_mode_now(); }

// ---- CU budget of the persistent GEMM grids (include/umr.h): process-wide, run-time
static std::atomic<int> g_cu_budget{-1};
int umr_cu_budget_now() {
    int b = g_cu_budget.load(std::memory_order_relaxed);
    if (b < 0) {
        const char* e = getenv("UMR_CU_BUDGET");
        b = e ? atoi(e) : 0;
        if (b < 0) b = 0;
        g_cu_budget.store(b, std::memory_order_relaxed);
    }
    return b;
}
extern "C" int umr_set_cu_budget(int cus) {
    if (cus < 0) return umr_set_error(UMR_ERR_INVALID, "umr_set_cu_budget: negative budget");
    g_cu_budget.store(cus, std::memory_order_relaxed);
    return UMR_OK;
}
extern "C" int umr_get_cu_budget(void) { return umr_cu_budget_now(); }

// ---- debug / A-B options (umr_common.h: umr_opt; include/umr.h: umr_set_debug_option)
static const char* const g_opt_names[UMR_OPT_COUNT] = {
    "UMR_GEMM_TILE", "UMR_NT_SPLITK", "UMR_SPLITK_FENCE", "UMR_NT_ORDER", "UMR_NT256_PERSIST", "UMR_NT256_BM",
    "UMR_X3_TRACE", "UMR_NT256_PH2", "UMR_ATTN_BWD_FUSED", "UMR_BILINEAR_GY", "UMR_HEAD_OUT_BWD_GENERIC"};
static std::atomic<int> g_opt[UMR_OPT_COUNT];
static int opt_parse(const char* v) {
    if (!v) return UMR_OPT_UNSET;
    if ((v[0] >= '0' && v[0] <= '9') || v[0] == '-' || v[0] == '+') return atoi(v);
    return (int)(unsigned char)v[0];          // letter options (UMR_NT_ORDER=n|m); an empty string reads as 0
}
// read from the environment exactly once, before any entry point can run (a static initialiser of the shared object: dlopen runs it
// on the loading thread)
static const bool g_opt_loaded = [] {
    for (int i = 0; i < UMR_OPT_COUNT; ++i) g_opt[i].store(opt_parse(getenv(g_opt_names[i])), std::memory_order_relaxed);
    return true;
}();
int umr_opt(int id) { return g_opt[id].load(std::memory_order_relaxed); }
extern "C" int umr_set_debug_option(const char* name, const char* value) {
    if (!name) return umr_set_error(UMR_ERR_INVALID, "umr_set_debug_option: null name");
    for (int i = 0; i < UMR_OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) {
            g_opt[i].store(opt_parse(value), std::memory_order_relaxed);
            return UMR_OK;
        }
    return umr_set_error(UMR_ERR_INVALID, "umr_set_debug_option: unknown option");
}
extern "C" int umr_get_debug_option(const char* name, int* value, int* is_set) {
    if (!name) return umr_set_error(UMR_ERR_INVALID, "umr_get_debug_option: null name");
    for (int i = 0; i < UMR_OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) {
            const int v = g_opt[i].load(std::memory_order_relaxed);
            if (is_set) *is_set = v != UMR_OPT_UNSET;
            if (value) *value = v == UMR_OPT_UNSET ? 0 : v;
            return UMR_OK;
        }
    return umr_set_error(UMR_ERR_INVALID, "umr_get_debug_option: unknown option");
}
