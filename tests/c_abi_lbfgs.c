/* The batched L-BFGS of the C-ABI (fdcap_lbfgs_*) from a plain C host: NP independent problems, objective evaluated by the caller --
 * here on the host, f_p(x) = sum_i ( a_i (x_i - c_pi)^2 + (x_i - c_pi)^4 ), minimum at x = c_p.  Rounds of (x to the host, f and g
 * back, fdcap_lbfgs_advance) until no problem wants another evaluation.  tests/test_gpu_c_host.py compiles this with gcc as C11 and
 * runs it; exit code 0 = every problem ended within 1e-3 of its minimum.  Prints rounds and the worst distance. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "fdcap.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
#define FD(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s -> %d\n", #x, e_); exit(4); } } while (0)
enum { NP = 9, DIM = 37, STRIDE = 40 };

int main(void) {
    static float x[NP * STRIDE], g[NP * STRIDE], f[NP], c[NP * DIM], a[DIM];
    unsigned s = 12345u;
    for (int i = 0; i < DIM; ++i) a[i] = 1.0f + 0.5f * (float)i;
    for (int p = 0; p < NP; ++p)
        for (int i = 0; i < DIM; ++i) {
            s = s * 1664525u + 1013904223u;
            c[p * DIM + i] = (float)(s >> 8) / 16777216.0f * 4.0f - 2.0f;
            x[p * STRIDE + i] = 0.0f;
        }
    fdcap_lbfgs_config cf = {DIM, 20, 100, 0, 1, 25, 1.0f, 1e-7f, 1e-9f, 0.0f, 0.0f};
    fdcap_lbfgs* opt = NULL;
    FD(fdcap_lbfgs_create(NP, &cf, &opt));
    float *x_d, *g_d, *f_d;
    int32_t* act_d;
    HIP(hipMalloc((void**)&x_d, sizeof x)); HIP(hipMalloc((void**)&g_d, sizeof g)); HIP(hipMalloc((void**)&f_d, sizeof f));
    HIP(hipMalloc((void**)&act_d, sizeof(int32_t)));
    HIP(hipMemcpy(x_d, x, sizeof x, hipMemcpyHostToDevice));
    int rounds = 0;
    int32_t active = 1;
    while (active && rounds < 2000) {
        HIP(hipMemcpy(x, x_d, sizeof x, hipMemcpyDeviceToHost));
        for (int p = 0; p < NP; ++p) {
            double fp = 0.0;
            for (int i = 0; i < DIM; ++i) {
                const float d = x[p * STRIDE + i] - c[p * DIM + i];
                fp += (double)(a[i] * d * d + d * d * d * d);
                g[p * STRIDE + i] = 2.0f * a[i] * d + 4.0f * d * d * d;
            }
            f[p] = (float)fp;
        }
        HIP(hipMemcpy(g_d, g, sizeof g, hipMemcpyHostToDevice));
        HIP(hipMemcpy(f_d, f, sizeof f, hipMemcpyHostToDevice));
        FD(fdcap_lbfgs_advance(opt, x_d, STRIDE, f_d, g_d, STRIDE, act_d, NULL));
        HIP(hipMemcpy(&active, act_d, sizeof active, hipMemcpyDeviceToHost));
        ++rounds;
    }
    HIP(hipMemcpy(x, x_d, sizeof x, hipMemcpyDeviceToHost));
    float worst = 0.0f;
    for (int p = 0; p < NP; ++p)
        for (int i = 0; i < DIM; ++i) worst = fmaxf(worst, fabsf(x[p * STRIDE + i] - c[p * DIM + i]));
    printf("rounds %d active %d worst |x - c| %.3g\n", rounds, (int)active, worst);
    fdcap_lbfgs_destroy(opt);
    return (active == 0 && worst < 1e-3f) ? 0 : 1;
}
