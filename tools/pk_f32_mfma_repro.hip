// Stand-alone reproducer attempt for DESIGN.md section 7: does a packed fp32 instruction return a wrong element
// while ANOTHER kernel's MFMAs share the CU?  No library code: kernel A = one-wave workgroups, every lane runs chains of
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 and, from the same inputs, the two scalar instructions each packed one stands for,
// and compares the bits in place; kernel B = MFMA loops on a second stream.  Phases: A alone, A next to B, A next to a VALU-only B.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip && /tmp/pk_repro [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));       // a 64-bit register pair {lo, hi}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// every float operation outside the three packed instructions under test is a scalar instruction by construction (hipcc would
// otherwise fuse pairs of them into packed ones itself)
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// rec[0] = mismatches; then up to 15 records {lane, chain step, which (0 fma 1 mul 2 add), element, packed bits, scalar bits, block, 0}
__global__ __launch_bounds__(64) void pk_chain_kernel(unsigned* rec, int steps, unsigned seed, int use_lds) {
    __shared__ float sh[64 * 2];
    const int lane = threadIdx.x;
    // lane-dependent, block-dependent operands (the library's failing lanes were 48-63 of a one-wave workgroup)
    float a0 = 1.0f + 0.001f * lane + 1e-6f * (blockIdx.x & 255), a1 = 0.75f - 0.002f * lane;
    float b0 = 0.999f + 1e-4f * ((seed + lane) & 31), b1 = 1.001f - 1e-4f * ((seed >> 3) & 31);
    float c0 = 0.01f * lane, c1 = -0.02f * lane;
    for (int s = 0; s < steps; ++s) {
        const f32x2 pa = {a0, a1}, pb = {b0, b1}, pc = {c0, c1};
        f32x2 pr;
        float q0, q1;
        const int which = s % 3;
        if (which == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pr) : "v"(pa), "v"(pb), "v"(pc));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(a0), "v"(b0), "v"(c0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(a1), "v"(b1), "v"(c1));
        } else if (which == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(pa), "v"(pb));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(a0), "v"(b0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(a1), "v"(b1));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pr) : "v"(pa), "v"(pc));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q0) : "v"(a0), "v"(c0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(a1), "v"(c1));
        }
        const float r0 = pr.x, r1 = pr.y;
        if (__float_as_uint(r0) != __float_as_uint(q0) || __float_as_uint(r1) != __float_as_uint(q1)) {
            const unsigned k = atomicAdd(&rec[0], 1u);
            if (k < 15) {
                unsigned* r = rec + 8 * (k + 1);
                const int el = __float_as_uint(r0) != __float_as_uint(q0) ? 0 : 1;
                r[0] = lane; r[1] = s; r[2] = which; r[3] = el; r[4] = __float_as_uint(el ? r1 : r0); r[5] = __float_as_uint(el ? q1 : q0); r[6] = blockIdx.x; r[7] = 0;
            }
        }
        a0 = sfma(q0, 0.5f, 0.6f); a1 = sfma(q1, 0.5f, 0.4f);   // keep the chain bounded and data-dependent
        if (use_lds) {                                        // the library's kernel hands values between lanes through LDS between its packed groups
            sh[lane] = a0; sh[64 + lane] = a1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a0 = sfma(sh[(lane + 1) & 63], 0.25f, sfma(a0, 0.75f, 0.f)); a1 = sfma(sh[64 + ((lane + 63) & 63)], 0.25f, sfma(a1, 0.75f, 0.f));
        }
        c0 = sfma(c0, 1.f, 0.001f); c1 = sfma(c1, 1.f, -0.001f);
    }
    if (a0 == 123.f) rec[127] = 1;
}
#ifdef PK_VICTIM_LIB
// tools/pk_bisect.py: the victim alone as a shared library, launched next to the real library's kernels
extern "C" int pk_chain_launch(unsigned* rec, int steps, unsigned seed, int use_lds, void* stream) {
    pk_chain_kernel<<<1028, 64, 0, (hipStream_t)stream>>>(rec, steps, seed, use_lds);
    return (int)hipGetLastError();
}
#endif
__global__ __launch_bounds__(256) void mfma_load_kernel(float* sink, int iters) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (threadIdx.x + k)); b[k] = (__bf16)(0.5f - 0.001f * k); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i)
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) *sink = 1.f;
}
__global__ __launch_bounds__(256) void valu_load_kernel(float* sink, int iters) {
    float x = threadIdx.x, y = 1.0001f;
    for (int i = 0; i < iters * 16; ++i) x = fmaf(x, y, 0.5f);
    if (x == 12345.f) *sink = x;
}
static unsigned phase(const char* name, int rounds, int neighbour, int use_lds, unsigned* rec, float* sink, hipStream_t sa, hipStream_t sb) {
    (void)hipMemsetAsync(rec, 0, 128 * 4, sa); (void)hipStreamSynchronize(sa);
    for (int r = 0; r < rounds; ++r) {
        if (neighbour == 1) mfma_load_kernel<<<512, 256, 0, sb>>>(sink, 40000);      // ~ms of matrix-pipe work on 512 x 4 waves
        if (neighbour == 2) valu_load_kernel<<<512, 256, 0, sb>>>(sink, 40000);
        for (int k = 0; k < 16; ++k) pk_chain_kernel<<<1028, 64, 0, sa>>>(rec, 600, 17u * r + k, use_lds);
        (void)hipStreamSynchronize(sa); (void)hipStreamSynchronize(sb);
    }
    unsigned h[128]; (void)hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s launches %5d  packed results checked %.2e  mismatches %u\n", name, rounds * 16, (double)rounds * 16 * 1028 * 64 * 600, h[0]);
    for (unsigned k = 0; k < (h[0] < 15 ? h[0] : 15); ++k) {
        const unsigned* r = h + 8 * (k + 1);
        printf("    lane %2u step %3u %s element %u packed %08x scalar %08x block %u\n", r[0], r[1], r[2] == 0 ? "v_pk_fma_f32" : r[2] == 1 ? "v_pk_mul_f32" : "v_pk_add_f32", r[3], r[4], r[5], r[6]);
    }
    return h[0];
}
#ifndef PK_VICTIM_LIB
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    unsigned* rec; float* sink; hipStream_t sa, sb;
    (void)hipMalloc(&rec, 128 * 4); (void)hipMalloc(&sink, 4); (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    unsigned bad = 0;
    for (int lds = 0; lds < 2; ++lds) {
        bad += phase(lds ? "alone, LDS hand-overs" : "alone", rounds, 0, lds, rec, sink, sa, sb);
        bad += phase(lds ? "next to MFMA kernel, LDS hand-overs" : "next to MFMA kernel", rounds, 1, lds, rec, sink, sa, sb);
        bad += phase(lds ? "next to VALU-only kernel, LDS hand-overs" : "next to VALU-only kernel", rounds, 2, lds, rec, sink, sa, sb);
    }
    printf("total mismatches %u\n", bad);
    return 0;
}
#endif
