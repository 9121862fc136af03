// TEST INFRASTRUCTURE ONLY: stand-alone self-test of the kernels' math headers, built by
// tests/test_sanitizers.py with -fsanitize=address,undefined (GPU sanitizers are not available on
// the pool, so memory / UB checking happens on this host build of the same source).
// Checks every hand-derived backward against central finite differences in double-rounded fp32.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../4dcapture-fpv_amd/csrc/fdc_frame.h"
#include "../../4dcapture-fpv_amd/csrc/fdc_host_setup.h"
#include "../../4dcapture-fpv_amd/csrc/fdc_loss.h"
#include "../../4dcapture-fpv_amd/csrc/fdc_math.h"
#include "../../4dcapture-fpv_amd/csrc/fdc_skin.h"

using namespace fdc;
struct NoSync { void operator()() const {} };

static unsigned long long rng_state = 88172645463325252ull;
static float urand() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f;
}
static int fails = 0;
static void expect(bool ok, const char* what, double a, double b) {
    if (!ok) { printf("FAIL %s: %g vs %g\n", what, a, b); ++fails; }
}

static const int PARENTS[NJ] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                                20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                                21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53};

int main() {
    // --- Gram-Schmidt and Rodrigues: backward vs finite differences of a random linear functional
    for (int trial = 0; trial < 50; ++trial) {
        float s6[6], w[9];
        for (float& v : s6) v = urand();
        for (float& v : w) v = urand();
        GsCache c;
        gs_forward(s6, 1, &c);
        M3 dR; for (int e = 0; e < 9; ++e) dR.m[e] = w[e];
        float g[6];
        gs_backward(c, dR, g, 1);
        for (int k = 0; k < 6; ++k) {
            float h = 1e-3f, sp[6], sm[6];
            for (int e = 0; e < 6; ++e) { sp[e] = s6[e]; sm[e] = s6[e]; }
            sp[k] += h; sm[k] -= h;
            M3 Rp = gs_forward(sp, 1, nullptr), Rm = gs_forward(sm, 1, nullptr);
            double fd = 0;
            for (int e = 0; e < 9; ++e) fd += (double)w[e] * ((double)Rp.m[e] - Rm.m[e]) / (2.0 * h);
            expect(fabs(fd - g[k]) < 2e-2 * (1.0 + fabs(fd)), "gs_backward", fd, g[k]);
        }
        V3 r = v3(urand() * 2, urand() * 2, urand() * 2);
        V3 gr = rodrigues_backward(r, dR);
        const float gg[3] = {gr.x, gr.y, gr.z};
        for (int k = 0; k < 3; ++k) {
            float h = 1e-3f;
            V3 rp = r, rm = r;
            (k == 0 ? rp.x : k == 1 ? rp.y : rp.z) += h;
            (k == 0 ? rm.x : k == 1 ? rm.y : rm.z) -= h;
            M3 Rp = rodrigues_forward(rp), Rm = rodrigues_forward(rm);
            double fd = 0;
            for (int e = 0; e < 9; ++e) fd += (double)w[e] * ((double)Rp.m[e] - Rm.m[e]) / (2.0 * h);
            expect(fabs(fd - gg[k]) < 2e-2 * (1.0 + fabs(fd)), "rodrigues_backward", fd, gg[k]);
        }
        // tgm round trip aa -> R -> aa
        V3 aa = v3(urand() * 1.5f, urand() * 1.5f, urand() * 1.5f);
        V3 back = tgm_rotmat_to_aa(tgm_aa_to_rotmat(aa));
        expect(fabs(back.x - aa.x) + fabs(back.y - aa.y) + fabs(back.z - aa.z) < 2e-5, "tgm round trip", back.x, aa.x);
    }
    // --- whole frame: pose_forward / pose_backward with a random model, directional derivative of
    //     L = <wA, A> + <wJ, Jw> wrt x, o, camera_ext, scale
    const int V = 64;
    std::vector<float> vt(V * 3), S10(V * 30), Jreg(NJ * V, 0.f), hc(2 * 12 * 45), hm(90);
    for (float& v : vt) v = urand();
    for (float& v : S10) v = 0.05f * urand();
    for (int j = 0; j < NJ; ++j) { Jreg[j * V + (j % V)] = 0.6f; Jreg[j * V + ((j * 7 + 3) % V)] = 0.4f; }
    for (float& v : hc) v = 0.1f * urand();
    for (float& v : hm) v = 0.1f * urand();
    HostPoseSetup hs;
    if (!host_pose_setup(V, vt.data(), S10.data(), Jreg.data(), PARENTS, &hs)) { printf("FAIL setup\n"); return 1; }
    PoseModel pm;
    pm.Jt = hs.Jt.data(); pm.Jd = hs.Jd.data(); pm.parents = hs.parents.data(); pm.order = hs.order.data();
    pm.level_start = hs.level_start.data(); pm.child_start = hs.child_start.data(); pm.child_list = hs.child_list.data();
    pm.hand_comp = hc.data(); pm.hand_mean = hm.data(); pm.nlevels = hs.nlevels;
    std::vector<float> x(XDIM), o(ODIM), cam(16), wA(NJ * 12), wJ(NJW * 3);
    for (float& v : x) v = 0.5f * urand();
    for (int j = 0; j < 21; ++j) { const float id6[6] = {1, 0, 0, 1, 0, 0}; for (int e = 0; e < 6; ++e) o[6 * j + e] = id6[e] + 0.3f * urand(); }
    x[3] = 1; x[4] = 0.1f; x[5] = 0.05f; x[6] = 0.9f; x[7] = -0.1f; x[8] = 0.2f;
    for (float& v : cam) v = urand();
    for (float& v : wA) v = urand();
    for (float& v : wJ) v = urand();
    float scale = 1.7f;
    auto L = [&](const std::vector<float>& xx, const std::vector<float>& oo, const std::vector<float>& cc, float ss) {
        static PoseScratch sc;
        std::vector<float> Rm(NJ * 9), PF(NPF), Jr(NJ * 3), G(NJ * 12), A(NJ * 12), M(12), Jw(NJW * 3);
        pose_forward(pm, xx.data(), oo.data(), cc.data(), ss, sc, Rm.data(), PF.data(), Jr.data(), G.data(), A.data(), M.data(),
                     Jw.data(), 0, 1, NoSync());
        double l = 0;
        for (int i = 0; i < NJ * 12; ++i) l += (double)wA[i] * A[i];
        for (int i = 0; i < NJW * 3; ++i) l += (double)wJ[i] * Jw[i];
        return l;
    };
    {
        static PoseScratch sc;
        std::vector<float> Rm(NJ * 9), PF(NPF), Jr(NJ * 3), G(NJ * 12), A(NJ * 12), M(12), Jw(NJW * 3);
        pose_forward(pm, x.data(), o.data(), cam.data(), scale, sc, Rm.data(), PF.data(), Jr.data(), G.data(), A.data(), M.data(),
                     Jw.data(), 0, 1, NoSync());
        std::vector<float> dx(XDIM, 0.f), dO(ODIM, 0.f), dcam(16, 0.f);
        float dscale = 0.f;
        pose_backward(pm, x.data(), o.data(), cam.data(), scale, Rm.data(), Jr.data(), G.data(), wA.data(), nullptr, wJ.data(),
                      nullptr, nullptr, nullptr, nullptr, sc, dx.data(), dO.data(), dcam.data(), &dscale, 0, 1, NoSync());
        const float h = 2e-3f;
        for (int k = 0; k < XDIM; ++k) {
            if (k >= X_LATENT && k < X_LATENT + 32) continue;      // latent enters through o only
            auto xp = x, xm = x; xp[k] += h; xm[k] -= h;
            double fd = (L(xp, o, cam, scale) - L(xm, o, cam, scale)) / (2.0 * h);
            expect(fabs(fd - dx[k]) < 3e-2 * (1.0 + fabs(fd)), "pose_backward dx", fd, dx[k]);
        }
        for (int k = 0; k < ODIM; k += 5) {
            auto op = o, om = o; op[k] += h; om[k] -= h;
            double fd = (L(x, op, cam, scale) - L(x, om, cam, scale)) / (2.0 * h);
            expect(fabs(fd - dO[k]) < 3e-2 * (1.0 + fabs(fd)), "pose_backward dO", fd, dO[k]);
        }
        for (int k = 0; k < 12; ++k) {
            auto cp = cam, cm = cam; cp[k] += h; cm[k] -= h;
            double fd = (L(x, o, cp, scale) - L(x, o, cm, scale)) / (2.0 * h);
            expect(fabs(fd - dcam[k]) < 3e-2 * (1.0 + fabs(fd)), "pose_backward dcam", fd, dcam[k]);
        }
        double fd = (L(x, o, cam, scale + h) - L(x, o, cam, scale - h)) / (2.0 * h);
        expect(fabs(fd - dscale) < 3e-2 * (1.0 + fabs(fd)), "pose_backward dscale", fd, dscale);
    }
    // --- Adam against its textbook form
    {
        float p = 0.3f, m = 0.f, v = 0.f;
        double pd = 0.3, md = 0, vd = 0;
        for (int t = 1; t <= 20; ++t) {
            float g = urand();
            adam_update(p, m, v, g, adam_scalars(0.005, t));
            md = 0.9 * md + 0.1 * g; vd = 0.999 * vd + 0.001 * (double)g * g;
            pd -= 0.005 * (md / (1 - pow(0.9, t))) / (sqrt(vd / (1 - pow(0.999, t))) + 1e-8);
        }
        expect(fabs(p - pd) < 1e-6, "adam", p, pd);
    }
    printf(fails ? "selftest: %d failures\n" : "selftest: ok\n", fails);
    return fails ? 1 : 0;
}
