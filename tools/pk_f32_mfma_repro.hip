// Stand-alone reproducer of NOTES.md section 6: on the MI355X boxes of this pool a packed fp32 instruction whose
// op_sel modifier makes the LOW result take the HIGH half of src2,
//     v_pk_fma_f32 K, B, B, R op_sel:[0,0,1] op_sel_hi:[1,1,0]        (K.lo = B.lo * B.lo + R.hi,  K.hi = B.hi * B.hi + R.lo)
// returns K.lo = B.lo * B.lo + 0 -- the addend is dropped -- in lanes 48-63 of the wave, now and then, WHILE ANOTHER KERNEL'S MFMAs
// run on the same CU.  Alone, or next to a VALU-only kernel: never.  No library code here: kernel A runs the sequence hipcc emits for
// the library's Rodrigues formula (k.k of the rotation axis; pose_forward, csrc/fdc_frame.h, built with packed fp32) as inline asm
// and checks every result against scalar instructions on the same inputs; kernel B keeps the matrix pipe (and LDS) busy on a
// second stream.  s_nop 7 in front of the instruction does not help; replacing the v_pk_add that follows by scalar adds does not
// help; replacing THIS instruction by two v_fma_f32 does.  The library is therefore built without packed fp32 instructions
// (-Xclang -target-feature -Xclang -packed-fp32-ops: hipcc forms op_sel swizzles freely and offers no finer switch).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip && /tmp/pk_repro [rounds]
// Measured (profiles/r4_pk_f32_repro.txt): alone 0 of 1.3e10; next to the MFMA kernel 2-6 million, all in lanes 48-63, all K.lo, all equal
// to the result with the addend dropped; next to the VALU-only kernel 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));       // a 64-bit register pair {lo, hi}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// scalar reference arithmetic as explicit instructions (hipcc would fuse pairs of C operations into packed ones itself)
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// rec[0] mismatches, rec[1] of them in lanes 48-63, rec[2] of them K.lo, rec[3] of them equal to "addend dropped";
// then up to 15 records {lane, step, element (0-1 A, 2-3 B, 4-5 R, 6-7 K), got, want, block}
// FIX = 1: the swizzled v_pk_fma_f32 replaced by two v_fma_f32
template <int FIX>
__global__ __launch_bounds__(64) void pk_swz_kernel(unsigned* rec, int steps, unsigned seed) {
    const int lane = threadIdx.x;
    float a0 = 0.31f + 0.011f * lane, a1 = -0.27f + 0.007f * lane, b0 = 0.12f + 0.003f * ((seed + lane) & 31), b1 = 0.45f - 0.004f * lane;
    for (int s = 0; s < steps; ++s) {
        const float r = 1.0f / (0.8f + 0.01f * ((s + lane) & 15));          // (the 1 / theta of the library's code)
        f32x2 vA = {a0, a1}, vB = {b0, b1}, vR = {r, -7.f}, vS, vK;
        asm volatile("v_pk_mul_f32 %[A], %[A], %[R] op_sel_hi:[1,0]\n"     // A *= R.lo
                     "v_pk_mul_f32 %[B], %[B], %[R] op_sel_hi:[1,0]\n"     // B *= R.lo
                     "v_pk_mul_f32 %[R], %[A], %[A]\n"                     // R = A^2
                     "v_pk_mul_f32 %[S], %[B], %[B]\n"                     // S = B^2
                     : [A] "+v"(vA), [B] "+v"(vB), [R] "+v"(vR), [S] "=&v"(vS));
        if (FIX) vK = f32x2{sfma(vB.x, vB.x, vR.y), sfma(vB.y, vB.y, vR.x)};
        else asm volatile("v_pk_fma_f32 %[K], %[B], %[B], %[R] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n" : [K] "=&v"(vK) : [B] "v"(vB), [R] "v"(vR));
        asm volatile("v_pk_add_f32 %[K], %[S], %[K] op_sel:[1,0] op_sel_hi:[0,1]\n" : [K] "+v"(vK) : [S] "v"(vS));     // K = {S.hi + K.lo, S.lo + K.hi}
        const float A0 = smul(a0, r), A1 = smul(a1, r), B0 = smul(b0, r), B1 = smul(b1, r);
        const float R0 = smul(A0, A0), R1 = smul(A1, A1), S0 = smul(B0, B0), S1 = smul(B1, B1);
        const float want[8] = {A0, A1, B0, B1, R0, R1, sadd(S1, sfma(B0, B0, R1)), sadd(S0, sfma(B1, B1, R0))};
        const float got[8] = {vA.x, vA.y, vB.x, vB.y, vR.x, vR.y, vK.x, vK.y};
        for (int k = 0; k < 8; ++k)
            if (__float_as_uint(want[k]) != __float_as_uint(got[k])) {
                const unsigned n = atomicAdd(&rec[0], 1u);
                if (lane >= 48) atomicAdd(&rec[1], 1u);
                if (k == 6) atomicAdd(&rec[2], 1u);
                if (k == 6 && __float_as_uint(got[k]) == __float_as_uint(sadd(S1, sfma(B0, B0, 0.f)))) atomicAdd(&rec[3], 1u);
                if (n < 15) { unsigned* q = rec + 8 * (n + 1); q[0] = lane; q[1] = s; q[2] = k; q[3] = __float_as_uint(got[k]); q[4] = __float_as_uint(want[k]); q[5] = blockIdx.x; }
            }
        a0 = sfma(want[6], 0.01f, smul(a0, 0.99f)); a1 = sfma(want[7], -0.01f, smul(a1, 0.99f));
    }
    if (a0 == 123.f) rec[127] = 1;
}
// the neighbours: MFMA loops with some LDS traffic (what the library's panel products look like to the CU) / VALU only
__global__ __launch_bounds__(256) void mfma_load_kernel(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float4 lbuf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lbuf[i] = make_float4(0.001f * i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 lacc = make_float4(0.f, 0.f, 0.f, 0.f);
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (threadIdx.x + k)); b[k] = (__bf16)(0.5f - 0.001f * k); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        const float4 v0 = lbuf[(threadIdx.x + 17 * i) & 2047];
        lacc.x += v0.x; lacc.y += v0.z;
        if ((i & 63) == 0) { lbuf[(threadIdx.x + i) & 2047] = lacc; __syncthreads(); }
    }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + lacc.x + lacc.y == 12345.f) *sink = 1.f;
}
__global__ __launch_bounds__(256) void valu_load_kernel(float* sink, int iters) {
    float x = threadIdx.x, y = 1.0001f;
    for (int i = 0; i < iters * 16; ++i) x = fmaf(x, y, 0.5f);
    if (x == 12345.f) *sink = x;
}
template <int FIX>
static unsigned phase(const char* name, int rounds, int neighbour, unsigned* rec, float* sink, hipStream_t sa, hipStream_t sb) {
    (void)hipMemsetAsync(rec, 0, 128 * 4, sa); (void)hipStreamSynchronize(sa);
    for (int r = 0; r < rounds; ++r) {
        if (neighbour == 1) mfma_load_kernel<<<512, 256, 0, sb>>>(sink, 40000);       // ~ms of matrix-pipe work, two workgroups per CU
        if (neighbour == 2) valu_load_kernel<<<512, 256, 0, sb>>>(sink, 40000);
        for (int k = 0; k < 16; ++k) pk_swz_kernel<FIX><<<1028, 64, 0, sa>>>(rec, 600, 17u * r + k);
        (void)hipStreamSynchronize(sa); (void)hipStreamSynchronize(sb);
    }
    unsigned h[128]; (void)hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-62s results checked %.2e  mismatches %u (lanes 48-63: %u, K.lo: %u, = addend dropped: %u)\n", name, (double)rounds * 16 * 1028 * 64 * 600 * 8,
           h[0], h[1], h[2], h[3]);
    for (unsigned k = 0; k < (h[0] < 4 ? h[0] : 4); ++k) {
        const unsigned* q = h + 8 * (k + 1);
        printf("    lane %2u step %3u element %u got %08x want %08x block %u\n", q[0], q[1], q[2], q[3], q[4], q[5]);
    }
    return h[0];
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    unsigned* rec; float* sink; hipStream_t sa, sb;
    (void)hipMalloc(&rec, 128 * 4); (void)hipMalloc(&sink, 4); (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    phase<0>("swizzled v_pk_fma_f32, alone", rounds, 0, rec, sink, sa, sb);
    phase<0>("swizzled v_pk_fma_f32, next to the MFMA kernel", rounds, 1, rec, sink, sa, sb);
    phase<0>("swizzled v_pk_fma_f32, next to the VALU-only kernel", rounds, 2, rec, sink, sa, sb);
    phase<1>("two v_fma_f32 in its place, next to the MFMA kernel", rounds, 1, rec, sink, sa, sb);
    return 0;
}
