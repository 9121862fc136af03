/* The optimiser through the C-ABI from a plain C host -- no Python, no torch in this process (tests/test_gpu_c_host.py compiles it
 * with gcc against include/fdcap.h and libfdcap_hip.so, runs it on files it wrote, and compares the results bit for bit with
 * FittingOP's on the same inputs).  What a non-Python caller of the reference's hot path does:
 *   fdcap_ctx_create -> fdcap_set_scene / fdcap_set_contact_ids -> fdcap_opt_create (registers ITS buffers) -> fdcap_opt_set_inputs
 *   -> fdcap_opt_run (the loop :560-593, one call; logged loss terms into a device-side history) -> fdcap_opt_get_results.
 * usage: c_abi_fit <dir>      reads <dir>/dims.txt and the *.bin arrays, writes <dir>/out_*.bin */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fdcap.h"

static void* slurp(const char* dir, const char* name, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void* p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read (%zu bytes wanted)\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}
static void dump(const char* dir, const char* name, const void* p, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
#define FD(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s -> %d\n", #x, e_); exit(4); } } while (0)
static void* to_device(const void* h, size_t bytes) {
    void* d; HIP(hipMalloc(&d, bytes ? bytes : 4)); HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return d;
}
static void* zeros_device(size_t bytes) { void* d; HIP(hipMalloc(&d, bytes)); HIP(hipMemset(d, 0, bytes)); return d; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
    const char* dir = argv[1];
    char path[1024];
    snprintf(path, sizeof path, "%s/dims.txt", dir);
    FILE* f = fopen(path, "r");
    int V, num_shape, ns, nc, N, num_iter, P, log_every;
    if (!f || fscanf(f, "%d %d %d %d %d %d %d %d", &V, &num_shape, &ns, &nc, &N, &num_iter, &P, &log_every) != 8) { fprintf(stderr, "dims.txt\n"); return 2; }
    fclose(f);

    fdcap_model_desc md;
    memset(&md, 0, sizeof md);
    md.num_verts = V; md.num_shape = num_shape;
    md.v_template = slurp(dir, "v_template.bin", (size_t)V * 3 * 4);
    md.shapedirs = slurp(dir, "shapedirs.bin", (size_t)V * 3 * num_shape * 4);
    md.posedirs = slurp(dir, "posedirs.bin", (size_t)486 * V * 3 * 4);
    md.J_regressor = slurp(dir, "J_regressor.bin", (size_t)55 * V * 4);
    md.parents = slurp(dir, "parents.bin", 55 * 4);
    md.lbs_weights = slurp(dir, "lbs_weights.bin", (size_t)V * 55 * 4);
    md.hands_componentsl = slurp(dir, "hcl.bin", 12 * 45 * 4); md.hands_componentsr = slurp(dir, "hcr.bin", 12 * 45 * 4);
    md.hands_meanl = slurp(dir, "hml.bin", 45 * 4); md.hands_meanr = slurp(dir, "hmr.bin", 45 * 4);
    md.vp_fc1_w = slurp(dir, "w1.bin", 512 * 32 * 4); md.vp_fc1_b = slurp(dir, "b1.bin", 512 * 4);
    md.vp_fc2_w = slurp(dir, "w2.bin", 512 * 512 * 4); md.vp_fc2_b = slurp(dir, "b2.bin", 512 * 4);
    md.vp_out_w = slurp(dir, "w3.bin", 126 * 512 * 4); md.vp_out_b = slurp(dir, "b3.bin", 126 * 4);

    fdcap_ctx* ctx = NULL;
    FD(fdcap_ctx_create(&md, &ctx));
    printf("%s (%s)\n", fdcap_version(), fdcap_build_info());
    float* scene = slurp(dir, "scene.bin", (size_t)ns * 3 * 4);
    int64_t* vid = slurp(dir, "vid.bin", (size_t)nc * 8);
    FD(fdcap_set_scene(ctx, scene, ns));
    FD(fdcap_set_contact_ids(ctx, vid, nc));

    /* the optimiser's state lives in the CALLER's device memory (the reference's nn.Parameters, :179-182) */
    float* rows_x = zeros_device((size_t)(N + 4) * FDCAP_XDIM * 4);
    float* rows_cam = zeros_device((size_t)(N + 4) * 16 * 4);
    float* scale = zeros_device(4); float* dscale = zeros_device(4);
    double* losses = zeros_device(FDCAP_NUM_LOSSES * 8);
    fdcap_opt_config* cfg = slurp(dir, "cfg.bin", sizeof(fdcap_opt_config));
    if (cfg->n_total != N || cfg->n_local != N || cfg->frame0 != 0) { fprintf(stderr, "cfg.bin does not describe a whole clip of %d frames\n", N); return 2; }
    FD(fdcap_opt_create(ctx, cfg, rows_x, rows_cam, scale, dscale, losses));
    float* data78 = to_device(slurp(dir, "data78.bin", (size_t)N * 78 * 4), (size_t)N * 78 * 4);
    float* init78 = to_device(slurp(dir, "init78.bin", (size_t)N * 78 * 4), (size_t)N * 78 * 4);
    float* mask = to_device(slurp(dir, "mask.bin", (size_t)N * 4), (size_t)N * 4);
    float* cam = to_device(slurp(dir, "cam.bin", (size_t)N * 16 * 4), (size_t)N * 16 * 4);
    FD(fdcap_opt_set_inputs(ctx, data78, init78, mask, cam, NULL));

    /* the loop: one call; every log_every-th iteration (and the last) leaves its loss partial sums in the history */
    int n_rows = 0;
    for (int ii = 0; ii < num_iter; ++ii) n_rows += log_every > 0 && (ii % log_every == 0 || ii == num_iter - 1);
    double* hist = zeros_device((size_t)(n_rows ? n_rows : 1) * FDCAP_NUM_LOSSES * 8);
    int32_t n_logged = -1;
    FD(fdcap_opt_run(ctx, 0, num_iter, num_iter, P, log_every, hist, n_rows, 0, &n_logged, NULL));
    if (n_logged != n_rows) { fprintf(stderr, "logged %d rows, expected %d\n", n_logged, n_rows); return 5; }

    float* body = zeros_device((size_t)N * FDCAP_PDIM * 4); float* scale_out = zeros_device(4); float* cam_out = zeros_device((size_t)N * 16 * 4);
    FD(fdcap_opt_get_results(ctx, body, scale_out, cam_out, NULL));
    HIP(hipDeviceSynchronize());
    {
        float* h = malloc((size_t)N * FDCAP_PDIM * 4); HIP(hipMemcpy(h, body, (size_t)N * FDCAP_PDIM * 4, hipMemcpyDeviceToHost));
        dump(dir, "out_body.bin", h, (size_t)N * FDCAP_PDIM * 4);
        float s; HIP(hipMemcpy(&s, scale_out, 4, hipMemcpyDeviceToHost)); dump(dir, "out_scale.bin", &s, 4);
        float* c = malloc((size_t)N * 16 * 4); HIP(hipMemcpy(c, cam_out, (size_t)N * 16 * 4, hipMemcpyDeviceToHost));
        dump(dir, "out_cam.bin", c, (size_t)N * 16 * 4);
        double* hh = malloc((size_t)(n_rows ? n_rows : 1) * FDCAP_NUM_LOSSES * 8);
        HIP(hipMemcpy(hh, hist, (size_t)n_rows * FDCAP_NUM_LOSSES * 8, hipMemcpyDeviceToHost));
        dump(dir, "out_hist.bin", hh, (size_t)n_rows * FDCAP_NUM_LOSSES * 8);
        printf("scale %.6f after %d iterations, %d logged rows\n", s, num_iter, n_rows);
    }
    fdcap_opt_destroy(ctx);
    fdcap_ctx_destroy(ctx);
    return 0;
}
