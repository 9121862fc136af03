// What does a kernel cost beyond its workgroups' own lifetime?  (The pose / VPoser / skinning kernels of the loop run 9-12 us each while
// their workgroups live 4-9 us by s_memtime stamps.)  Back-to-back launches in one stream of a kernel whose every workgroup is busy for
// exactly T us (wall clock), in the launch shapes of the loop's kernels; printed: time per launch minus T.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lop tools/launch_overhead_probe.hip && /tmp/lop
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void body(int ticks, float* sink, float4* out, int out_f4_per_wg) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    // what the workgroup leaves behind: out_f4_per_wg float4 stores (dirty lines the end of the kernel has to write back)
    for (int i = threadIdx.x; i < out_f4_per_wg; i += blockDim.x) out[(size_t)blockIdx.x * out_f4_per_wg + i] = make_float4(1.f, 2.f, 3.f, (float)i);
    if (lds[(threadIdx.x * 7) & 63] == -1.f) *sink = 0.f;
}
static void run(const char* what, int grid, int block, int lds_bytes, float us, double out_mb = 0.0) {
    float* sink; hipMalloc(&sink, 4);
    const int f4 = (int)(out_mb * 1e6 / 16 / grid);
    float4* out; hipMalloc(&out, (size_t)grid * (f4 + 1) * 16);
    hipFuncSetAttribute((const void*)body, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 300, ticks = (int)(us * 100);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(body, dim3(grid), dim3(block), lds_bytes, 0, ticks, sink, out, f4);
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(body, dim3(grid), dim3(block), lds_bytes, 0, ticks, sink, out, f4);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s grid %5d x %4d threads, %3d KB LDS, body %.1f us, %4.1f MB written: %.2f us per launch = body + %.2f\n", what, grid, block,
           lds_bytes / 1024, us, out_mb, ms * 1e3 / n, ms * 1e3 / n - us);
    hipFree(out); hipFree(sink);
}
// The same chain of dependent launches captured once into a hipGraph and replayed: is the per-kernel cost any lower than in a stream?
static void run_graph(const char* what, int grid, int block, int lds_bytes, float us, double out_mb = 0.0) {
    float* sink; hipMalloc(&sink, 4);
    const int f4 = (int)(out_mb * 1e6 / 16 / grid);
    float4* out; hipMalloc(&out, (size_t)grid * (f4 + 1) * 16);
    hipStream_t st; hipStreamCreate(&st);
    const int n = 300, ticks = (int)(us * 100);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(body, dim3(grid), dim3(block), lds_bytes, st, ticks, sink, out, f4);
    if (hipStreamEndCapture(st, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture failed\n"); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s grid %5d x %4d threads, %3d KB LDS, body %.1f us, %4.1f MB written: %.2f us per node   = body + %.2f   (hipGraph of %d nodes)\n", what, grid,
           block, lds_bytes / 1024, us, out_mb, ms * 1e3 / n, ms * 1e3 / n - us, n);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st); hipFree(out); hipFree(sink);
}
int main() {
    for (float us : {0.f, 4.f}) {
        run("one workgroup", 1, 64, 1024, us);
        run("pose kernels' shape (1024 frames)", 1024, 256, 32 * 1024, us);
        run("... one wave per frame", 1024, 64, 32 * 1024, us);
        run("skinning backward's shape", 1024, 256, 36 * 1024, us);
        run("VPoser kernels' shape", 256, 512, 82 * 1024, us);
        run("blend product's shape", 256, 768, 95 * 1024, us);
        run("Chamfer search's shape (two generations of one-wave WGs)", 16000, 64, 4864, us);
        run("skinning forward's shape", 2048, 256, 3 * 1024, us);
    }
    for (double mb : {1.0, 2.0, 5.0, 10.0, 20.0}) run("pose kernels' shape, writing", 1024, 256, 32 * 1024, 4.f, mb);
    for (double mb : {2.0, 6.0}) run("VPoser kernels' shape, writing", 256, 512, 82 * 1024, 4.f, mb);
    for (float us : {0.f, 4.f}) {
        run_graph("graph: pose kernels' shape (1024 frames)", 1024, 256, 32 * 1024, us);
        run_graph("graph: VPoser kernels' shape", 256, 512, 82 * 1024, us);
    }
    run_graph("graph: pose kernels' shape, writing", 1024, 256, 32 * 1024, 4.f, 5.0);
    return 0;
}
