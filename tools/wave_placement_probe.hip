// Where do the waves of a 4-wave workgroup land?  pose_fwd / pose_bwd keep ONE wave per workgroup for the arithmetic (the other three
// only issue staging copies): if wave 0 of every workgroup sits on the same SIMD of its CU, four frames' arithmetic shares one SIMD.
// Prints, over 1024 workgroups of 256 threads (the pose kernels' launch shape), the histogram of the SIMD id of wave w, w = 0..3,
// and how many distinct SIMDs the wave-0s of one CU occupy.   hipcc --offload-arch=gfx950 -O2 -o /tmp/wpp tools/wave_placement_probe.hip && /tmp/wpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    __shared__ float pad[3800];                                    // ~15 KB like the pose kernels: 4+ workgroups per CU
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);      // stay resident so all 1024 coexist
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = ((xcc & 15) << 16) | (hw & 0xffff);   // simd [5:4] cu [11:8] sh [12] se [15:13]
    }
    if (pad[(threadIdx.x * 7) % 3800] == -1.f) out[0] = 0;
}
int main() {
    unsigned* d; hipMalloc(&d, 1024 * 4 * 4);
    probe<<<1024, 256>>>(d, 2000); hipDeviceSynchronize();
    std::vector<unsigned> h(4096); hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w) {
        int hist[4] = {0, 0, 0, 0};
        for (int b = 0; b < 1024; ++b) hist[(h[b * 4 + w] >> 4) & 3]++;
        printf("wave %d of a workgroup: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[0], hist[1], hist[2], hist[3]);
    }
    std::map<unsigned, std::vector<int>> cu;                       // physical CU -> SIMD of each resident workgroup's wave 0
    for (int b = 0; b < 1024; ++b) cu[h[b * 4] & 0xfff00u].push_back((h[b * 4] >> 4) & 3);
    int dist[5] = {0, 0, 0, 0, 0}; size_t wg = 0;
    for (auto& kv : cu) { std::set<int> s(kv.second.begin(), kv.second.end()); dist[s.size()]++; wg += kv.second.size(); }
    printf("%zu CUs hold %.1f workgroups each; distinct SIMDs used by their wave-0s: 1 -> %d CUs, 2 -> %d, 3 -> %d, 4 -> %d\n", cu.size(),
           (double)wg / cu.size(), dist[1], dist[2], dist[3], dist[4]);
    return 0;
}
