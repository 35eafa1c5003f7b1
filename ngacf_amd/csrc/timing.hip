// Opt-in per-kernel timing from the dispatch packets' own timestamps (hipExtLaunchKernel start / stop events): the duration of
// the KERNEL, without the host's launch gaps that an event pair around a launch call includes -- the number rocprofv3
// --kernel-trace reports.  bench.py's roofline leg brackets eager passes with ngacf_timing_begin / ngacf_timing_end.
// State is per process, behind a mutex: the launches of ANY host thread are recorded between begin and end (autograd runs a
// backward's launches on its own worker thread); nothing is recorded, created or synchronised unless somebody asked for it.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "ngacf_common.hpp"

namespace ngacf {

struct TimedLaunch { hipEvent_t a, b; int tag; };
struct Timing {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<TimedLaunch> v;
};
static Timing& timing() {
    static Timing t;
    return t;
}

bool timing_on() { return timing().on.load(std::memory_order_relaxed); }

int timing_events(hipEvent_t* a, hipEvent_t* b) {
    hipError_t e = hipEventCreate(a);
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipEventCreate(b);
    if (e != hipSuccess) { (void)hipEventDestroy(*a); return static_cast<int>(e); }
    return 0;
}

// AFTER the launch that records the pair has been issued: a concurrent ngacf_timing_end (another host thread) must never meet
// an event that no launch has been attached to yet.  (A pair recorded after that end stays in the list until the next begin.)
void timing_record(int tag, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(timing().mu);
    timing().v.push_back({a, b, tag});
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_timing_begin(void) {
    Timing& t = timing();
    std::lock_guard<std::mutex> lk(t.mu);
    for (auto& l : t.v) { (void)hipEventDestroy(l.a); (void)hipEventDestroy(l.b); }
    t.v.clear();
    t.on = true;
    return 0;
}

extern "C" int ngacf_timing_end(float* ms, int32_t* tags, int cap) {
    Timing& t = timing();
    t.on = false;
    std::lock_guard<std::mutex> lk(t.mu);
    int n = 0, rc = 0;
    for (auto& l : t.v) {
        if (rc == 0) {
            hipError_t e = hipEventSynchronize(l.b);
            float d = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&d, l.a, l.b);
            if (e != hipSuccess) rc = static_cast<int>(e);
            else if (n < cap && ms && tags) { ms[n] = d; tags[n] = l.tag; ++n; }
        }
        (void)hipEventDestroy(l.a);
        (void)hipEventDestroy(l.b);
    }
    t.v.clear();
    return rc != 0 ? -1000 - rc : n;
}
