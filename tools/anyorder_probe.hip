// Does a kernel launched with hipExtAnyOrderLaunch start before its predecessor in the SAME stream has finished on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx boards; measured rather than assumed.)  Kernel A: one workgroup busy ~200 us.
// Kernel B: stamps its start.  Printed: B.start - A.start and A.end - A.start on the 100 MHz clock, for an ordinary launch of B and
// for an any-order launch.   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder_probe tools/anyorder_probe.hip && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void busy(unsigned long long* t, int ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
__global__ void stamp(unsigned long long* t) { if (threadIdx.x == 0) t[2] = wall_clock64(); }
int main() {
    unsigned long long* t; hipMalloc(&t, 64); hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(t, 0, 64, s); hipStreamSynchronize(s);
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, s, t, 20000);              // 200 us
            if (mode == 0) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, t);
            else hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t);
            hipStreamSynchronize(s);
            unsigned long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
            printf("%s launch of B: A ran %.1f us, B started %.1f us after A started\n", mode ? "any-order" : "ordinary ",
                   (h[1] - h[0]) * 0.01, ((long long)h[2] - (long long)h[0]) * 0.01);
        }
    return 0;
}
