// VALU issue rate of a gfx950 SIMD, measured: independent streams of one instruction, ~1..8 waves per SIMD, s_memtime
// around the loop.  Every wave records its physical SIMD (HW_REG_HW_ID, HW_REG_XCC_ID); per SIMD: cycles per wave64 instruction =
// (last end - first start) / instructions its waves issued -- whatever the dispatcher's placement was.  Printed: the median over
// the SIMDs with at least W waves, the mean waves per SIMD seen, and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip && /tmp/valu_issue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#define R8(x) x x x x x x x x
#define R4(x) x x x x
template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long* t, float* sink, int iters) {
    float a[8], b = 1.0001f, c = 0.5f;
    double p[4];                                                  // (register pairs for the packed / 64-bit forms)
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
    for (int k = 0; k < 4; ++k) p[k] = threadIdx.x + k;
    const unsigned long long r0 = wall_clock64();                  // 100 MHz
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {                             // 32 instructions per trip, 8 (4) independent dependency chains
        if (OP == 0) { R4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 1) { R4(asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 2) { R4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 3) { R4(asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 4) { R8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[0]));) }
#define TWO(op) R4(asm volatile(op " %0, %0, %8\n " op " %1, %1, %8\n " op " %2, %2, %8\n " op " %3, %3, %8\n " op " %4, %4, %8\n " op " %5, %5, %8\n " op " %6, %6, %8\n " op " %7, %7, %8" \
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
#define THREE(op) R4(asm volatile(op " %0, %0, %8, %9\n " op " %1, %1, %8, %9\n " op " %2, %2, %8, %9\n " op " %3, %3, %8, %9\n " op " %4, %4, %8, %9\n " op " %5, %5, %8, %9\n " op " %6, %6, %8, %9\n " op " %7, %7, %8, %9" \
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
        if (OP == 6) { TWO("v_add_f32") }
        if (OP == 7) { TWO("v_mul_f32") }
        if (OP == 8) { TWO("v_sub_f32") }
        if (OP == 9) { TWO("v_min_f32") }
        if (OP == 10) { TWO("v_and_b32") }
        if (OP == 11) { THREE("v_med3_f32") }
        if (OP == 12) { TWO("v_add_u32") }
        if (OP == 14) { TWO("v_min_u32") }
        if (OP == 15) { THREE("v_min3_u32") }
        if (OP == 16) { TWO("v_min_i32") }
        if (OP == 17) { TWO("v_max_f32") }
        if (OP == 18) { TWO("v_lshlrev_b32") }
        if (OP == 19) { TWO("v_xor_b32") }
        if (OP == 20) { THREE("v_bfe_u32") }
        if (OP == 21) { THREE("v_perm_b32") }
        if (OP == 22) { THREE("v_mad_u32_u24") }
        if (OP == 23) { THREE("v_min3_i32") }
        if (OP == 24) { THREE("v_and_or_b32") }
        if (OP == 25) { TWO("v_or_b32") }
        if (OP == 26) { THREE("v_or3_b32") }
        if (OP == 27) { THREE("v_add3_u32") }
        if (OP == 28) { THREE("v_lshl_or_b32") }
        if (OP == 29) { THREE("v_bfi_b32") }
        if (OP == 30) { TWO("v_sub_u32") }
        if (OP == 31) { TWO("v_fmac_f32") }
        if (OP == 32) { THREE("v_xad_u32") }
        if (OP == 33) { TWO("v_mul_u32_u24") }
        if (OP == 34) { TWO("v_max_u32") }
        if (OP == 35) { TWO("v_subrev_f32") }
        if (OP == 5) { R8(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3"
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* r = t + 4 * (size_t)((blockIdx.x * 256 + threadIdx.x) >> 6);
        r[0] = t0; r[1] = t1; r[2] = ((unsigned long long)(xcc & 15) << 32) | (hw & 0xfff0u);   // simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13]
        r[3] = wall_clock64() - r0;
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k]; for (int k = 0; k < 4; ++k) s += (float)p[k];
    if (s == 12345.678f) *sink = s;
}
template <int OP> void run(const char* name) {
    const int iters = 4000, ninst = iters * 32;
    unsigned long long* t; float* sink; hipMalloc(&t, 8 * 256 * 8 * 4 * 4); hipMalloc(&sink, 4);
    printf("%-16s", name);
    for (int W = 1; W <= 8; ++W) {                                 // 256 CUs x W workgroups of 4 waves: W waves on every SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<OP><<<256 * W, 256>>>(t, sink, 10);                  // (warm-up)
        hipEventRecord(e0); probe<OP><<<256 * W, 256>>>(t, sink, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * W * 4 * 4); hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        std::map<unsigned long long, std::vector<size_t>> simd;                       // physical SIMD -> its waves
        for (size_t w = 0; w < h.size() / 4; ++w) simd[h[4 * w + 2]].push_back(w);
        std::vector<double> cpi;
        double nw = 0, ck = 0;
        for (size_t w = 0; w < h.size() / 4; ++w) ck += (double)(h[4 * w + 1] - h[4 * w]) / (double)h[4 * w + 3] * 0.1;    // GHz
        ck /= (double)(h.size() / 4);
        for (auto& kv : simd) {
            unsigned long long lo = ~0ull, hi = 0;
            for (size_t w : kv.second) { lo = std::min(lo, h[4 * w]); hi = std::max(hi, h[4 * w + 1]); }
            nw += kv.second.size();
            if ((int)kv.second.size() >= W) cpi.push_back((double)(hi - lo) / ((double)kv.second.size() * ninst));
        }
        std::sort(cpi.begin(), cpi.end());
        printf("  W=%d %.2f (%.1f w/SIMD on %zu SIMDs, %.2f GHz)", W, cpi.empty() ? 0.0 : cpi[cpi.size() / 2], nw / simd.size(), simd.size(), ck); (void)ms;
    }
    printf("\n");
}
int main() {
    printf("shader cycles per wave64 instruction per physical SIMD (median over SIMDs holding >= W waves); launch = 256 x W workgroups of 4 waves\n");
    run<0>("v_fma_f32"); run<1>("v_min3_f32"); run<2>("v_cmp_lt_f32"); run<3>("v_alignbit_b32"); run<4>("v_pk_fma_f32"); run<5>("v_lshrrev_b64");
    run<6>("v_add_f32"); run<7>("v_mul_f32"); run<8>("v_sub_f32"); run<9>("v_min_f32"); run<10>("v_and_b32"); run<11>("v_med3_f32"); run<12>("v_add_u32");
    run<14>("v_min_u32"); run<15>("v_min3_u32"); run<16>("v_min_i32"); run<17>("v_max_f32"); run<18>("v_lshlrev_b32"); run<19>("v_xor_b32");
    run<20>("v_bfe_u32"); run<21>("v_perm_b32"); run<22>("v_mad_u32_u24"); run<23>("v_min3_i32"); run<24>("v_and_or_b32");
    run<25>("v_or_b32"); run<26>("v_or3_b32"); run<27>("v_add3_u32"); run<28>("v_lshl_or_b32"); run<29>("v_bfi_b32"); run<30>("v_sub_u32");
    run<31>("v_fmac_f32"); run<32>("v_xad_u32"); run<33>("v_mul_u32_u24"); run<34>("v_max_u32"); run<35>("v_subrev_f32");
    return 0;
}
