// Library-wide state of libpy4cast_hip.so: thread-local error message, device queries.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"

namespace p4c {

static thread_local char g_err[512] = "";

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    // per device (one process may drive several): filled on first use under a lock, read lock-free afterwards
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cus[dev].load(std::memory_order_acquire);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;  // MI355X
    cus[dev].store(n, std::memory_order_release);
    return n;
}

int diag_skip(int bit) {
    static const int mask = [] { const char* e = diag_env("P4C_DIAG"); return e ? atoi(e) : 0; }();
    if (!(mask & bit)) return 0;
    static std::atomic<int> calls[32];
    int idx = 0;
    while ((1 << idx) < bit) ++idx;
    return calls[idx].fetch_add(1) > 400;
}

int ensure_dyn_smem(const void* kernel, int bytes) {
    struct Key { const void* k; int dev; };
    static std::mutex mu;
    static std::vector<std::pair<Key, int>> done;   // a few dozen entries at most: linear scan
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : done)
        if (e.first.k == kernel && e.first.dev == dev) {
            if (e.second >= bytes) return P4C_OK;
            P4C_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            e.second = bytes;
            return P4C_OK;
        }
    P4C_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.push_back({Key{kernel, dev}, bytes});
    return P4C_OK;
}

// ---- kernel timing (disabled by default)
struct ProfRec { hipEvent_t a, b; int tag; int64_t units; bool open; };
static std::vector<ProfRec> g_prof;          // records in use
static std::vector<hipEvent_t> g_prof_pool;  // events created once by p4c_prof_enable (none are created in the timed region)
static size_t g_prof_pool_next = 0;
static int g_prof_mask = 0;
static int64_t g_prof_min_units = 0;
static size_t g_prof_cap = 0;
static std::mutex g_prof_mu;

static thread_local bool g_prof_bwd = false;
void prof_set_backward(bool on) { g_prof_bwd = on; }
static inline int phase_tag(int tag) { return (tag == P4C_PROF_CONV3X3_C64 && g_prof_bwd) ? P4C_PROF_CONV3X3_C64_BWD : tag; }

void prof_begin(int tag, int64_t units, hipStream_t stream) {
    tag = phase_tag(tag);
    if (!(g_prof_mask & tag) || units < g_prof_min_units) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() >= g_prof_cap || g_prof_pool_next + 2 > g_prof_pool.size()) return;
    ProfRec r{g_prof_pool[g_prof_pool_next], g_prof_pool[g_prof_pool_next + 1], tag, units, true};
    g_prof_pool_next += 2;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
}

void prof_end(int tag, hipStream_t stream) {
    tag = phase_tag(tag);
    if (!(g_prof_mask & tag)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof.size(); i-- > 0;)
        if (g_prof[i].tag == tag && g_prof[i].open) {
            (void)hipEventRecord(g_prof[i].b, stream);
            g_prof[i].open = false;
            return;
        }
}

}  // namespace p4c

extern "C" int p4c_prof_enable(int tag_mask, int max_records) {
    std::lock_guard<std::mutex> lk(p4c::g_prof_mu);
    p4c::g_prof.clear();
    p4c::g_prof_pool_next = 0;
    p4c::g_prof_mask = tag_mask;
    p4c::g_prof_min_units = 0;
    p4c::g_prof_cap = (tag_mask && max_records > 0) ? (size_t)max_records : 0;
    while (p4c::g_prof_pool.size() < 2 * p4c::g_prof_cap) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return p4c::fail(P4C_ERR_RUNTIME, "p4c_prof_enable: hipEventCreate failed");
        p4c::g_prof_pool.push_back(e);
    }
    return P4C_OK;
}

extern "C" int p4c_prof_filter(int64_t min_units) {
    std::lock_guard<std::mutex> lk(p4c::g_prof_mu);
    p4c::g_prof_min_units = min_units;
    return P4C_OK;
}

extern "C" int p4c_prof_collect(int tag, int64_t min_units, double* total_ms, int* count, double* total_units) {
    std::lock_guard<std::mutex> lk(p4c::g_prof_mu);
    double ms = 0.0, units = 0.0;
    int n = 0;
    for (auto& r : p4c::g_prof) {
        if (r.tag != tag || r.open || r.units < min_units) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms += t; units += (double)r.units; ++n; }
    }
    if (total_ms) *total_ms = ms;
    if (count) *count = n;
    if (total_units) *total_units = units;
    return P4C_OK;
}

extern "C" int p4c_version(void) { return P4C_VERSION; }
extern "C" const char* p4c_last_error(void) { return p4c::err_buf(); }
extern "C" int p4c_num_cus(void) { return p4c::num_cus(); }
