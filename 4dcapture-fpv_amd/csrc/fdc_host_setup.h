// Host-side preprocessing of the body-model constants (runs once per context): the collapsed
// joint regressor and the kinematic-tree bookkeeping the per-frame kernels walk.
#pragma once
#include <algorithm>
#include <vector>

#include "fdc_frame.h"

namespace fdc {

struct HostPoseSetup {
    std::vector<float> Jt, Jd;                       // [55,3], [55,3,10]
    std::vector<int> parents, order, level_start, child_start, child_list, depth;
    int nlevels = 0;
};

// J = J_regressor @ (v_template + shapedirs beta) collapsed to Jt + Jd beta,
// accumulated in double.  shapedirs10: [V,3,10].
inline bool host_pose_setup(int V, const float* v_template, const float* shapedirs10, const float* J_regressor,
                            const int* parents_in, HostPoseSetup* out) {
    out->Jt.assign(NJ * 3, 0.f);
    out->Jd.assign(NJ * 3 * NBETA, 0.f);
    for (int j = 0; j < NJ; ++j)
        for (int k = 0; k < 3; ++k) {
            double a = 0.0, d[NBETA] = {0};
            for (int v = 0; v < V; ++v) {
                double w = J_regressor[(size_t)j * V + v];
                if (w == 0.0) continue;
                a += w * v_template[(size_t)3 * v + k];
                for (int l = 0; l < NBETA; ++l) d[l] += w * shapedirs10[((size_t)3 * v + k) * NBETA + l];
            }
            out->Jt[3 * j + k] = (float)a;
            for (int l = 0; l < NBETA; ++l) out->Jd[(3 * j + k) * NBETA + l] = (float)d[l];
        }
    out->parents.assign(parents_in, parents_in + NJ);
    std::vector<int> depth(NJ, 0);
    for (int j = 0; j < NJ; ++j) {
        int d = 0, p = out->parents[j];
        while (p >= 0 && d <= NJ) { ++d; p = out->parents[p]; }
        if (d > NJ) return false;                    // cycle
        depth[j] = d;
    }
    out->depth = depth;
    out->order.resize(NJ);
    for (int j = 0; j < NJ; ++j) out->order[j] = j;
    std::stable_sort(out->order.begin(), out->order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
    if (depth[out->order[0]] != 0 || (NJ > 1 && depth[out->order[1]] == 0)) return false;   // exactly one root
    out->nlevels = depth[out->order[NJ - 1]] + 1;
    out->level_start.assign(out->nlevels + 1, 0);
    for (int j = 0; j < NJ; ++j) out->level_start[depth[j] + 1]++;
    for (int l = 0; l < out->nlevels; ++l) out->level_start[l + 1] += out->level_start[l];
    out->child_start.assign(NJ + 1, 0);
    out->child_list.clear();
    for (int p = 0; p < NJ; ++p) {
        out->child_start[p] = (int)out->child_list.size();
        for (int j = 0; j < NJ; ++j)
            if (out->parents[j] == p) out->child_list.push_back(j);
    }
    out->child_start[NJ] = (int)out->child_list.size();
    out->child_list.push_back(0);                    // keep the array non-empty
    return true;
}

}  // namespace fdc
