// roctx ranges around the launches SURVEY.md section 5 names (K8 blend products, K14 Chamfer NN, K22 optimiser step) and
// around every iteration's backward, for `rocprofv3 --kernel-trace --marker-trace`.  Off unless FDCAP_ROCTX=1: then the
// marker library is bound at run time (librocprofiler-sdk-roctx.so.1, the one rocprofv3 intercepts; libroctx64.so.4 as
// the older spelling) -- no link-time dependency, nothing on the launch path when off but one predictable branch.
// Host-side ranges: they bracket the ENQUEUE of the launches (the kernels run later); the profiler pairs a range with its
// kernels through the dispatches' correlation ids.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>

namespace fdc {

struct RoctxApi {
    bool enabled = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {                                             // (runs once, under the function-local static's lock)
        const char* e = getenv("FDCAP_ROCTX");
        if (!e || e[0] != '1') return;
        const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char* n : names) {
            void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
            pop = (int (*)())dlsym(lib, "roctxRangePop");
            if (push && pop) { enabled = true; return; }
        }
    }
    bool on() const { return enabled; }
};
inline RoctxApi& roctx() { static RoctxApi api; return api; }

// scope guard: TraceRange r("fdcap:chamfer_nn");
struct TraceRange {
    bool live;
    explicit TraceRange(const char* name) : live(roctx().on()) { if (live) roctx().push(name); }
    ~TraceRange() { if (live) roctx().pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace fdc
