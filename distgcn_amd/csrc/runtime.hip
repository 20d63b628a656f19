// Error reporting, version and per-kernel-family event timing for libdgcn.so.
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "options.h"

namespace dgcn {

static thread_local char g_err[512] = "";
thread_local DoneHook g_done_hook;
thread_local CompactHook g_compact_hook;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------------------
// Options (options.h): one atomic word per switch.
struct OptionRow { const char* key; int64_t dflt; };
static const OptionRow kOptionRows[OPT_COUNT] = {
#define DGCN_OPT_ROW(e, k, d) {k, d},
    DGCN_OPTION_LIST(DGCN_OPT_ROW)
#undef DGCN_OPT_ROW
};
static std::atomic<int64_t> g_options[OPT_COUNT] = {
#define DGCN_OPT_INIT(e, k, d) {d},
    DGCN_OPTION_LIST(DGCN_OPT_INIT)
#undef DGCN_OPT_INIT
};
int64_t opt64(Opt o) { return g_options[o].load(std::memory_order_relaxed); }
void opt_store(Opt o, int64_t v) { g_options[o].store(v, std::memory_order_relaxed); }
static int option_index(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(kOptionRows[i].key, key) == 0) return i;
    return -1;
}

// ---------------------------------------------------------------------------------------------
// Timing: when enabled, each TimedLaunch owns a (start, stop) event pair that the launch itself stamps
// (DGCN_LAUNCH -> hipExtLaunchKernelGGL).  Nothing is synchronised until dgcn_timing_read().
// Event pairs belong to the device they were created on: the slots are kept PER DEVICE, and a launch takes its pair from the
// device that is current when it is issued (one process driving several devices with timing on).
struct TimingSlot {
    std::string family;
    hipEvent_t start, stop;
};
struct DeviceSlots {
    std::vector<TimingSlot> slots;
    size_t live = 0;  // slots [0, live) carry a recorded pair
};
constexpr int kMaxDevices = 64;
static std::mutex g_tmu;
static DeviceSlots g_dev[kMaxDevices];
static std::atomic<bool> g_timing{false};
static int g_timing_every = 1;      // every N-th launch carries an event pair (dgcn_timing_enable(N)); 1 = every launch
static unsigned long long g_timing_seq = 0;

// The pairs only measure time: hipEventDisableSystemFence ("events that are only being used to measure timing": no system-scope
// release - cache write-back and invalidation - when the event completes, so an instrumented launch neither pays for flushing the
// L2 behind it nor makes the next launch start on a cold one).  Nobody reads memory on the strength of these events: results are
// fetched behind the caller's own stream synchronisation.
static bool create_pair(TimingSlot& t) {
    if (hipEventCreateWithFlags(&t.start, hipEventDisableSystemFence) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&t.stop, hipEventDisableSystemFence) != hipSuccess) { (void)hipEventDestroy(t.start); return false; }
    return true;
}

static int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev < 0 || dev >= kMaxDevices ? 0 : dev;
}

TimedLaunch::TimedLaunch(const char* family, hipStream_t s) : slot(-1), device(0), stream(s) {
    (void)hipGetLastError();  // drop any stale error so check_launch() reports this launch only
    if (!g_timing.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    if (!g_timing.load(std::memory_order_relaxed)) return;  // (switched off while this thread waited for the lock)
    if (g_timing_every > 1 && (g_timing_seq++ % (unsigned long long)g_timing_every) != 0) return;  // (a sample: this launch goes out plain)
    device = current_device();
    DeviceSlots& d = g_dev[device];
    if (d.live == d.slots.size()) {
        TimingSlot t;
        if (!create_pair(t)) return;
        d.slots.push_back(t);
    }
    slot = (int)d.live++;
    d.slots[slot].family = family;
}

hipEvent_t TimedLaunch::start_ev() const {
    std::lock_guard<std::mutex> lk(g_tmu);
    return g_dev[device].slots[slot].start;
}

hipEvent_t TimedLaunch::stop_ev() const {
    std::lock_guard<std::mutex> lk(g_tmu);
    return g_dev[device].slots[slot].stop;
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_version(void) { return DGCN_VERSION; }
extern "C" const char* dgcn_last_error(void) { return g_err; }

extern "C" int dgcn_set_option(const char* key, int64_t value) {
    const int i = option_index(key);
    if (i < 0) return fail(DGCN_ERR_ARG, "dgcn_set_option: unknown option '%s'", key ? key : "(null)");
    g_options[i].store(value, std::memory_order_relaxed);
    return DGCN_OK;
}
extern "C" int dgcn_get_option(const char* key, int64_t* value) {
    const int i = option_index(key);
    if (i < 0 || !value) return fail(DGCN_ERR_ARG, "dgcn_get_option: unknown option '%s' (or null output)", key ? key : "(null)");
    *value = g_options[i].load(std::memory_order_relaxed);
    return DGCN_OK;
}
extern "C" int dgcn_option_count(void) { return OPT_COUNT; }
extern "C" const char* dgcn_option_name(int32_t index, int64_t* default_value) {
    if (index < 0 || index >= OPT_COUNT) return nullptr;
    if (default_value) *default_value = kOptionRows[index].dflt;
    return kOptionRows[index].key;
}

extern "C" int dgcn_timing_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing.store(on != 0, std::memory_order_release);
    if (on != 0) g_timing_every = on > 1 ? on : 1;  // (switching off keeps the factor of the records that stay readable)
    g_timing_seq = 0;
    // The event pairs of the first few thousand launches exist before anything is timed: created on demand they would be
    // created INSIDE the caller's timed region (two hipEventCreate per launch, once per slot), which is not the workload's time.
    // (On the device that is current here; another device's pairs are created on demand.)
    constexpr size_t kPrimed = 4096;
    DeviceSlots& d = g_dev[current_device()];
    if (on != 0 && d.slots.size() < kPrimed) {
        d.slots.reserve(kPrimed);
        while (d.slots.size() < kPrimed) {
            TimingSlot t;
            if (!create_pair(t)) break;  // (on demand then, as before)
            d.slots.push_back(t);
        }
    }
    return DGCN_OK;
}

extern "C" int dgcn_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (int i = 0; i < kMaxDevices; ++i) g_dev[i].live = 0;
    return DGCN_OK;
}

extern "C" int dgcn_timing_read(const char* kernel, double* total_ms, int64_t* launches) {
    if (!kernel || !total_ms || !launches) return fail(DGCN_ERR_ARG, "dgcn_timing_read: null argument");
    std::lock_guard<std::mutex> lk(g_tmu);
    double ms = 0.0;
    int64_t n = 0;
    for (int dv = 0; dv < kMaxDevices; ++dv) {
        const DeviceSlots& d = g_dev[dv];
        for (size_t i = 0; i < d.live; ++i) {
            if (d.slots[i].family != kernel) continue;
            if (hipEventSynchronize(d.slots[i].stop) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "dgcn_timing_read: event sync failed");
            float t = 0.f;
            if (hipEventElapsedTime(&t, d.slots[i].start, d.slots[i].stop) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "dgcn_timing_read: elapsed failed");
            ms += t;
            ++n;
        }
    }
    *total_ms = ms;
    *launches = n;
    return DGCN_OK;
}

extern "C" int32_t dgcn_timing_sampling(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    return g_timing_every;
}
