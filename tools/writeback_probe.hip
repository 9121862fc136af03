// What does the data a kernel WRITES cost at the kernel's end, and can the store's cache policy move that cost into the kernel's body?
// (tools/launch_overhead_probe.hip: ~0.2 us per MB a kernel leaves dirty in L2.)  Back-to-back launches of a kernel whose workgroups
// first store their share of `mb` MB and THEN stay busy for 4 us: whatever the launch costs beyond 4 us + the fixed 1.2 is the
// write-back.  Store forms: plain; __builtin_nontemporal_store (global_store ... nt); system-scope relaxed atomic store (sc0 sc1).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wbp tools/writeback_probe.hip && /tmp/wbp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int MODE, bool FIRST>
__global__ void body(int ticks, float* sink, f4v* out, int out_f4_per_wg) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    auto writes = [&]() {
        for (int i = threadIdx.x; i < out_f4_per_wg; i += blockDim.x) {
            f4v* p = out + (size_t)blockIdx.x * out_f4_per_wg + i;
            const f4v v = {1.f, 2.f, 3.f, (float)i};
            if (MODE == 0) *p = v;
            else if (MODE == 1) __builtin_nontemporal_store(v, p);
            else {
                float* q = (float*)p;
#pragma unroll
                for (int e = 0; e < 4; ++e) __hip_atomic_store(q + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (FIRST) writes();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (!FIRST) writes();
    if (lds[(threadIdx.x * 7) & 63] == -1.f) *sink = 0.f;
}
template <int MODE, bool FIRST>
static void run(const char* what, int grid, int block, int lds_bytes, float us, double out_mb) {
    float* sink; hipMalloc(&sink, 4);
    const int f4 = (int)(out_mb * 1e6 / 16 / grid);
    f4v* out; hipMalloc(&out, (size_t)grid * (f4 + 1) * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 300, ticks = (int)(us * 100);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((body<MODE, FIRST>), dim3(grid), dim3(block), lds_bytes, 0, ticks, sink, out, f4);
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((body<MODE, FIRST>), dim3(grid), dim3(block), lds_bytes, 0, ticks, sink, out, f4);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %-22s %4.1f MB: %.2f us per launch = body + %.2f\n", what, FIRST ? "stores, then 4 us busy" : "4 us busy, then stores", out_mb,
           ms * 1e3 / n, ms * 1e3 / n - us);
    hipFree(out); hipFree(sink);
}
int main() {
    for (double mb : {0.0, 6.0, 12.0}) {
        run<0, true>("plain stores", 1024, 256, 32 * 1024, 4.f, mb);
        run<0, false>("plain stores", 1024, 256, 32 * 1024, 4.f, mb);
        run<1, true>("nontemporal stores (nt)", 1024, 256, 32 * 1024, 4.f, mb);
        run<1, false>("nontemporal stores (nt)", 1024, 256, 32 * 1024, 4.f, mb);
        run<2, true>("system-scope stores (sc0 sc1)", 1024, 256, 32 * 1024, 4.f, mb);
        run<2, false>("system-scope stores (sc0 sc1)", 1024, 256, 32 * 1024, 4.f, mb);
    }
    return 0;
}
