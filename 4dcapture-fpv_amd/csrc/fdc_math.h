// Small fixed-size linear algebra and rotation parametrisations, forward and hand-derived
// backward.  Everything here is `FDC_HD` so the same source is compiled by hipcc into the
// gfx950 kernels and by g++ into the test-only host harness (tests/cpu_harness), which checks
// each backward against the oracle's autograd.  No product path runs the host build.
//
// Reference semantics restated (file:line in /root/reference):
//   gs_forward        cvae.py:62-72  (ContinousRotReprDecoder.decode, basis vectors as columns)
//   rodrigues_forward smplx.lbs.batch_rodrigues (absent package)
//   tgm_aa_to_rotmat  torchgeometry.angle_axis_to_rotation_matrix (cvae.py:92; Appendix A.1)
//   tgm_rotmat_to_aa  torchgeometry.rotation_matrix_to_angle_axis (cvae.py:83; Appendix A.1)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define FDC_HD __host__ __device__ __forceinline__
#else
#define FDC_HD inline
#endif

namespace fdc {

// Diagnosis (fdcap_debug_kernel_forms): the name of every kernel FORM the host side has launched since the last reset -- several
// stages pick among forms by row count / set size, and tests that mean to cover one form must be able to see that it ran.
struct FormLog { const char* name[48]; int n; };
inline FormLog& form_log() { static FormLog f{}; return f; }
inline void note_form(const char* name) {                  // (string literals: compared by address first, by content across headers)
    FormLog& f = form_log();
    for (int i = 0; i < f.n; ++i) if (f.name[i] == name || !__builtin_strcmp(f.name[i], name)) return;
    if (f.n < 48) f.name[f.n++] = name;
}

#if defined(__HIPCC__)
// Sum over the 64 lanes of a wavefront, the same bits in every lane, fixed order.  Four DPP steps (quad_perm x2,
// row_half_mirror, row_mirror: no LDS traffic) leave every 16-lane row with its row sum, then ((r0 + r1) + r2) + r3 down the rows.  The obvious __shfl_xor butterfly compiles to six ds_bpermute_b32 (LDS crossbar round trips) per sum.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum64(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    // ((r0 + r1) + r2) + r3 down the rows (late r4): row_bcast15 hands lane 15 of row k - 1 to row k; one row at a time (row_mask), so
    // row 1 becomes r1 + r0, row 2 r2 + (r0 + r1), row 3 r3 + ((r0 + r1) + r2) -- the sums of the four-readlane form bit for bit
    // (an unselected row adds +0.0), in three DPP adds and one v_readlane instead of four readlanes and three adds on scalars.
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0x2, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0x4, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0x8, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min64(float v) {
    v = fminf(v, dpp_move<0xB1>(v)); v = fminf(v, dpp_move<0x4E>(v)); v = fminf(v, dpp_move<0x141>(v)); v = fminf(v, dpp_move<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_max64(float v) { return -wave_min64(-v); }
#endif

struct V3 { float x, y, z; };
struct M3 { float m[9]; };   // row-major m[3*r+c]

FDC_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
FDC_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
FDC_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
FDC_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
FDC_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FDC_HD V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
FDC_HD M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.f; return r; }
FDC_HD M3 m3_identity() { M3 r = m3_zero(); r.m[0] = r.m[4] = r.m[8] = 1.f; return r; }
FDC_HD M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
// a * b^T
FDC_HD M3 m3_mul_bt(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
    return r;
}
// a^T * b
FDC_HD M3 m3_mul_at(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
    return r;
}
FDC_HD V3 m3_vec(const M3& a, V3 v) {
    return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z,
              a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
              a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
FDC_HD V3 m3t_vec(const M3& a, V3 v) {
    return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z,
              a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
              a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
// r += a (x) b   (outer product)
FDC_HD void m3_add_outer(M3& r, V3 a, V3 b) {
    r.m[0] += a.x * b.x; r.m[1] += a.x * b.y; r.m[2] += a.x * b.z;
    r.m[3] += a.y * b.x; r.m[4] += a.y * b.y; r.m[5] += a.y * b.z;
    r.m[6] += a.z * b.x; r.m[7] += a.z * b.y; r.m[8] += a.z * b.z;
}
FDC_HD void m3_add(M3& r, const M3& a) { for (int i = 0; i < 9; ++i) r.m[i] += a.m[i]; }

// ---- 6D continuous representation (Gram-Schmidt) ---------------------------------------
// Input layout is the reference's view(-1,3,2): a1 = (s0,s2,s4), a2 = (s1,s3,s5).
// F.normalize semantics: v / max(||v||, 1e-12).
struct GsCache { V3 b1, b2, a2; float n1, n2, d; };

FDC_HD M3 gs_forward(const float* s, int stride, GsCache* c) {
    V3 a1 = v3(s[0], s[2 * stride], s[4 * stride]);
    V3 a2 = v3(s[stride], s[3 * stride], s[5 * stride]);
    float n1 = fmaxf(sqrtf(dot(a1, a1)), 1e-12f);
    V3 b1 = (1.f / n1) * a1;
    float d = dot(b1, a2);
    V3 u = a2 - d * b1;
    float n2 = fmaxf(sqrtf(dot(u, u)), 1e-12f);
    V3 b2 = (1.f / n2) * u;
    V3 b3 = cross(b1, b2);
    M3 R;
    R.m[0] = b1.x; R.m[1] = b2.x; R.m[2] = b3.x;
    R.m[3] = b1.y; R.m[4] = b2.y; R.m[5] = b3.y;
    R.m[6] = b1.z; R.m[7] = b2.z; R.m[8] = b3.z;
    if (c) { c->b1 = b1; c->b2 = b2; c->a2 = a2; c->n1 = n1; c->n2 = n2; c->d = d; }
    return R;
}

// dR (row-major gradient wrt R) -> ds[6] in the same interleaved layout as the input.
FDC_HD void gs_backward(const GsCache& c, const M3& dR, float* ds, int stride) {
    V3 db1 = v3(dR.m[0], dR.m[3], dR.m[6]);
    V3 db2 = v3(dR.m[1], dR.m[4], dR.m[7]);
    V3 db3 = v3(dR.m[2], dR.m[5], dR.m[8]);
    // b3 = b1 x b2
    db1 = db1 + cross(c.b2, db3);
    db2 = db2 + cross(db3, c.b1);
    // b2 = u / n2
    V3 du = (1.f / c.n2) * (db2 - dot(c.b2, db2) * c.b2);
    // u = a2 - d b1 ; d = b1 . a2
    V3 da2 = du;
    float dd = -dot(du, c.b1);
    db1 = db1 - c.d * du;
    db1 = db1 + dd * c.a2;
    da2 = da2 + dd * c.b1;
    // b1 = a1 / n1
    V3 da1 = (1.f / c.n1) * (db1 - dot(c.b1, db1) * c.b1);
    ds[0] = da1.x; ds[2 * stride] = da1.y; ds[4 * stride] = da1.z;
    ds[stride] = da2.x; ds[3 * stride] = da2.y; ds[5 * stride] = da2.z;
}

// ---- Rodrigues as smplx writes it ---------------------------------------------------------
FDC_HD M3 rodrigues_forward(V3 r) {
    V3 rp = v3(r.x + 1e-8f, r.y + 1e-8f, r.z + 1e-8f);
    float th = sqrtf(dot(rp, rp));
    V3 k = (1.f / th) * r;
    float s = sinf(th), c = cosf(th);
    float oc = 1.f - c;
    float kk = dot(k, k);
    M3 R;
    R.m[0] = 1.f + oc * (k.x * k.x - kk); R.m[1] = -s * k.z + oc * k.x * k.y;  R.m[2] = s * k.y + oc * k.x * k.z;
    R.m[3] = s * k.z + oc * k.y * k.x;    R.m[4] = 1.f + oc * (k.y * k.y - kk); R.m[5] = -s * k.x + oc * k.y * k.z;
    R.m[6] = -s * k.y + oc * k.z * k.x;   R.m[7] = s * k.x + oc * k.z * k.y;   R.m[8] = 1.f + oc * (k.z * k.z - kk);
    return R;
}

FDC_HD V3 rodrigues_backward(V3 r, const M3& dR) {
    V3 rp = v3(r.x + 1e-8f, r.y + 1e-8f, r.z + 1e-8f);
    float th = sqrtf(dot(rp, rp));
    float ith = 1.f / th;
    V3 k = ith * r;
    float s = sinf(th), c = cosf(th);
    float oc = 1.f - c;
    float kk = dot(k, k);
    float tr = dR.m[0] + dR.m[4] + dR.m[8];
    // <dR, K> with K = [k]x ;  <dR, k k^T - kk I>
    V3 vee = v3(dR.m[7] - dR.m[5], dR.m[2] - dR.m[6], dR.m[3] - dR.m[1]);
    float dRK = dot(vee, k);
    V3 symk = v3((dR.m[0] + dR.m[0]) * k.x + (dR.m[1] + dR.m[3]) * k.y + (dR.m[2] + dR.m[6]) * k.z,
                 (dR.m[3] + dR.m[1]) * k.x + (dR.m[4] + dR.m[4]) * k.y + (dR.m[5] + dR.m[7]) * k.z,
                 (dR.m[6] + dR.m[2]) * k.x + (dR.m[7] + dR.m[5]) * k.y + (dR.m[8] + dR.m[8]) * k.z);
    float dRKK = 0.5f * dot(symk, k) - kk * tr;
    float dth = c * dRK + s * dRKK;
    V3 dk = s * vee + oc * (symk - (2.f * tr) * k);
    // k = r / th
    V3 dr = ith * dk;
    dth -= dot(dk, k) * ith;
    // th = |r + 1e-8|
    dr = dr + (dth * ith) * rp;
    return dr;
}

// ---- torchgeometry 0.1.2 conversions (used outside the loop: input 75->78, output 78->75) --
FDC_HD M3 tgm_aa_to_rotmat(V3 aa) {
    float th2 = dot(aa, aa);
    M3 R;
    if (th2 > 1e-6f) {
        float th = sqrtf(th2);
        V3 w = (1.f / (th + 1e-6f)) * aa;
        float c = cosf(th), s = sinf(th), oc = 1.f - c;
        R.m[0] = c + w.x * w.x * oc;        R.m[1] = w.x * w.y * oc - w.z * s;  R.m[2] = w.y * s + w.x * w.z * oc;
        R.m[3] = w.z * s + w.x * w.y * oc;  R.m[4] = c + w.y * w.y * oc;        R.m[5] = -w.x * s + w.y * w.z * oc;
        R.m[6] = -w.y * s + w.x * w.z * oc; R.m[7] = w.x * s + w.y * w.z * oc;  R.m[8] = c + w.z * w.z * oc;
    } else {
        R.m[0] = 1.f;   R.m[1] = -aa.z; R.m[2] = aa.y;
        R.m[3] = aa.z;  R.m[4] = 1.f;   R.m[5] = -aa.x;
        R.m[6] = -aa.y; R.m[7] = aa.x;  R.m[8] = 1.f;
    }
    return R;
}

FDC_HD V3 tgm_rotmat_to_aa(const M3& R) {
    // the library transposes first: rt[i][j] = R[j][i]
    const float r00 = R.m[0], r11 = R.m[4], r22 = R.m[8];
    const float t01 = R.m[3], t10 = R.m[1], t02 = R.m[6], t20 = R.m[2], t12 = R.m[7], t21 = R.m[5];
    float qw, qx, qy, qz, t;
    if (r22 < 1e-6f) {
        if (r00 > r11) { t = 1.f + r00 - r11 - r22; qw = t12 - t21; qx = t; qy = t01 + t10; qz = t20 + t02; }
        else           { t = 1.f - r00 + r11 - r22; qw = t20 - t02; qx = t01 + t10; qy = t; qz = t12 + t21; }
    } else {
        if (r00 < -r11) { t = 1.f - r00 - r11 + r22; qw = t01 - t10; qx = t20 + t02; qy = t12 + t21; qz = t; }
        else            { t = 1.f + r00 + r11 + r22; qw = t; qx = t12 - t21; qy = t20 - t02; qz = t01 - t10; }
    }
    // q = (q / sqrt(t)) * 0.5 in the library; same value up to one rounding
    qw = qw / sqrtf(t) * 0.5f; qx = qx / sqrtf(t) * 0.5f; qy = qy / sqrtf(t) * 0.5f; qz = qz / sqrtf(t) * 0.5f;
    float s2 = qx * qx + qy * qy + qz * qz;
    float s = sqrtf(s2);
    float two_theta = 2.f * (qw < 0.f ? atan2f(-s, -qw) : atan2f(s, qw));
    float k = s2 > 0.f ? two_theta / s : 2.f;
    return v3(qx * k, qy * k, qz * k);
}

// Gradient of tgm_rotmat_to_aa: g = d loss / d aa -> d loss / d R (row-major), through the branch the forward selects
// (what autograd of the library's masked blend gives: the unselected candidates are multiplied by 0).  At s2 == 0
// (R == I) torch's `where` leaves 0/0 = NaN in the gradient; here the selected branch k = 2 is differentiated instead
// (finite).  Needed by the operator-level VPoser backward only (decode(..., 'aa')); the optimiser loop never goes R -> aa.
FDC_HD M3 tgm_rotmat_to_aa_backward(const M3& R, V3 g) {
    // rt[i][j] = R[j][i]
    float rt[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rt[3 * i + j] = R.m[3 * j + i];
    int br;
    if (rt[8] < 1e-6f) br = rt[0] > rt[4] ? 0 : 1; else br = rt[0] < -rt[4] ? 2 : 3;
    float t, q[4];                      // q = raw candidate (w, x, y, z)
    switch (br) {
    case 0: t = 1.f + rt[0] - rt[4] - rt[8]; q[0] = rt[5] - rt[7]; q[1] = t; q[2] = rt[1] + rt[3]; q[3] = rt[6] + rt[2]; break;
    case 1: t = 1.f - rt[0] + rt[4] - rt[8]; q[0] = rt[6] - rt[2]; q[1] = rt[1] + rt[3]; q[2] = t; q[3] = rt[5] + rt[7]; break;
    case 2: t = 1.f - rt[0] - rt[4] + rt[8]; q[0] = rt[1] - rt[3]; q[1] = rt[6] + rt[2]; q[2] = rt[5] + rt[7]; q[3] = t; break;
    default: t = 1.f + rt[0] + rt[4] + rt[8]; q[0] = t; q[1] = rt[5] - rt[7]; q[2] = rt[6] - rt[2]; q[3] = rt[1] - rt[3]; break;
    }
    const float c = 0.5f / sqrtf(t);
    const float w = q[0] * c, x = q[1] * c, y = q[2] * c, z = q[3] * c;
    const float s2 = x * x + y * y + z * z, s = sqrtf(s2);
    float dw = 0.f, dx, dy, dz;
    if (s2 > 0.f) {
        const float tt = 2.f * (w < 0.f ? atan2f(-s, -w) : atan2f(s, w));
        const float k = tt / s;
        const float dk = g.x * x + g.y * y + g.z * z;
        const float dtt = dk / s;
        float ds = -dk * tt / s2;
        const float den = w * w + s2;
        ds += dtt * 2.f * w / den;
        dw = -dtt * 2.f * s / den;
        const float ds2 = ds / (2.f * s);
        dx = k * g.x + 2.f * x * ds2; dy = k * g.y + 2.f * y * ds2; dz = k * g.z + 2.f * z * ds2;
    } else {
        dx = 2.f * g.x; dy = 2.f * g.y; dz = 2.f * g.z;
    }
    // q_scaled = q_raw * c, c = 0.5 t^-1/2
    const float dq[4] = {dw * c, dx * c, dy * c, dz * c};
    const float dc = dw * q[0] + dx * q[1] + dy * q[2] + dz * q[3];
    float dt = dc * (-0.25f) / (t * sqrtf(t));
    float d[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // d loss / d rt
    switch (br) {
    case 0: d[5] += dq[0]; d[7] -= dq[0]; dt += dq[1]; d[1] += dq[2]; d[3] += dq[2]; d[6] += dq[3]; d[2] += dq[3];
            d[0] += dt; d[4] -= dt; d[8] -= dt; break;
    case 1: d[6] += dq[0]; d[2] -= dq[0]; d[1] += dq[1]; d[3] += dq[1]; dt += dq[2]; d[5] += dq[3]; d[7] += dq[3];
            d[0] -= dt; d[4] += dt; d[8] -= dt; break;
    case 2: d[1] += dq[0]; d[3] -= dq[0]; d[6] += dq[1]; d[2] += dq[1]; d[5] += dq[2]; d[7] += dq[2]; dt += dq[3];
            d[0] -= dt; d[4] -= dt; d[8] += dt; break;
    default: dt += dq[0]; d[5] += dq[1]; d[7] -= dq[1]; d[6] += dq[2]; d[2] -= dq[2]; d[1] += dq[3]; d[3] -= dq[3];
            d[0] += dt; d[4] += dt; d[8] += dt; break;
    }
    M3 dR;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dR.m[3 * j + i] = d[3 * i + j];
    return dR;
}

}  // namespace fdc
