// Batched L-BFGS with a strong-Wolfe line search: one INDEPENDENT problem per workgroup (four waves stage its state in LDS, one
// runs it), all problems advanced by one launch between two evaluations of the caller's objective (SMPLify-X fits every frame with L-BFGS +
// strong Wolfe; round 3 shipped the inner fit with Adam and listed this as its deviation).
//
// Not in the reference repository (the per-frame fit is the external SMPLify-X step, /root/reference/README.md:14-17).
// Restated from the published algorithm of torch.optim.LBFGS(line_search_fn="strong_wolfe") -- two-loop recursion over a
// bounded history, first step min(1, 1/|g|_1)·lr, bracketing by cubic extrapolation, zoom by cubic interpolation with the
// 10 %-of-bracket safeguard, sufficient decrease c1 = 1e-4, curvature c2 = 0.9, at most 25 line-search evaluations -- plus
// SMPLify-X's outer loop around `optimizer.step` (stop when the relative change of the loss at two successive step() calls
// is <= ftol, or every gradient entry is below gtol).  tests/test_gpu_lbfgs.py runs torch.optim.LBFGS itself (CPU) on the
// same objectives next to this kernel, problem by problem.
//
// A problem is a resumable state machine: `lbfgs_advance_kernel` consumes the objective's value and gradient at the point it
// asked for last time, runs until it needs the next evaluation (writes that point into the caller's x row) or is finished,
// and stores its scalars.  Problems that finished stay where they are while the others go on.
#pragma once
#include "fdc_math.h"

namespace fdc {

constexpr int LB_DPAD = 128;     // floats per stored vector: dim <= 128 (an optimiser row is 78)
constexpr int LB_HMAX = 128;     // history entries at most (torch's default is 100)
constexpr int LB_NT = 256;       // threads of a workgroup: four waves load the history, the first runs the state machine
FDC_HD size_t lbfgs_lds_bytes(int hist) { return (size_t)(2 * hist + 7) * LB_DPAD * sizeof(float); }    // + LV_NUM work vectors

struct LbfgsCfg {
    int dim, hist, max_iter, max_eval, max_steps, max_ls;
    float lr, tol_grad, tol_change, ftol, gtol;
};

enum { LB_INIT = 0, LB_WAIT_BRACKET = 1, LB_WAIT_ZOOM = 2, LB_DONE = 3 };

// Optional: columns [col0, col0 + n) of every gradient row still lack an addend that lies in FOUR partial arrays (the VPoser
// backward's quarter workgroups leave the latent gradient that way; the clip optimiser's Adam folds them itself, and so does this
// kernel for the inner fit -- one launch less per evaluation): g[p][col0 + c] += (part0 + part1) + (part2 + part3) at [p * n + c],
// the partials `stride` floats apart.  part = nullptr: nothing to add.
struct LbfgsFold { const float* part = nullptr; size_t stride = 0; int col0 = 0, n = 0; };

struct LbfgsScalars {
    int phase, step, n_inner, n_iter_total, evals_step, evals_total, ls_iter, ls_evals, nh, h0, low, insuf;
    float loss, orig_loss, prev_orig, prev_loss, t, gtd, d_norm, H_diag;
    float t_prev, f_prev, gtd_prev;
    float b0, b1, bf0, bf1, bgtd0, bgtd1;
};

// vector slots of a problem's workspace, followed by `hist` y vectors and `hist` s vectors
enum { LV_G = 0, LV_PREVG, LV_D, LV_XINIT, LV_GPREV, LV_BG0, LV_BG1, LV_NUM };
static_assert(LV_NUM == 7, "lbfgs_lds_bytes counts seven work vectors");
FDC_HD size_t lbfgs_ws_floats(int hist) { return (size_t)(LV_NUM + 2 * hist) * LB_DPAD; }

#if defined(__HIPCC__)
struct LV { float a, b; };       // a lane's two elements of a 128-padded vector: [lane], [lane + 64]
__device__ __forceinline__ LV lv_ld(const float* v) { return {v[threadIdx.x], v[threadIdx.x + 64]}; }
__device__ __forceinline__ void lv_st(float* v, LV x) { v[threadIdx.x] = x.a; v[threadIdx.x + 64] = x.b; }
// wave_sum64's sum, bit for bit (((r0 + r1) + r2) + r3 over the four rows), with the row sums read out side by side: this
// kernel's cost is a lone wave's chain of DEPENDENT instructions, and four readlanes + three adds are four deep where the three
// row_bcast steps + readlane are seven (the other kernels count issued instructions instead and use wave_sum64)
__device__ __forceinline__ float lb_sum(float v) {
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ float lv_dot(LV x, LV y) { return lb_sum(fmaf(x.a, y.a, x.b * y.b)); }
__device__ __forceinline__ float lv_absmax(LV x) {
    float m = fmaxf(fabsf(x.a), fabsf(x.b));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return m;
}
__device__ __forceinline__ float lv_abssum(LV x) { return lb_sum(fabsf(x.a) + fabsf(x.b)); }

// minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clipped to [lo, hi] (to the two abscissae without bounds)
__device__ __forceinline__ float lb_cubic(float x1, float f1, float g1, float x2, float f2, float g2, bool bounded, float lo, float hi) {
    if (!bounded) { lo = fminf(x1, x2); hi = fmaxf(x1, x2); }
    const float d1 = g1 + g2 - 3.f * (f1 - f2) / (x1 - x2);
    const float d2s = d1 * d1 - g1 * g2;
    if (d2s >= 0.f) {
        const float d2 = sqrtf(d2s);
        const float mp = (x1 <= x2) ? x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.f * d2))
                                    : x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.f * d2));
        return fminf(fmaxf(mp, lo), hi);
    }
    return 0.5f * (lo + hi);
}

// n_active: += 1 per problem that wants another round; n_active_next: zeroed (the two alternate from round to round, so that
// counting needs no launch of its own).
// X [nprob rows of x_stride]: the caller's parameters (read for x_init, written with the next trial point / the accepted point);
// F [nprob], G [nprob rows of g_stride]: the objective and its gradient at X as the previous call left it.
__global__ __launch_bounds__(LB_NT) void lbfgs_advance_kernel(LbfgsCfg cf, LbfgsScalars* __restrict__ S, float* __restrict__ W,
                                                           float* __restrict__ RO, float* __restrict__ X, int x_stride,
                                                           const float* __restrict__ F, const float* __restrict__ G, int g_stride,
                                                           int* __restrict__ n_active, int* __restrict__ n_active_next, LbfgsFold fold) {
    __shared__ float s_ro[LB_HMAX], s_al[LB_HMAX], s_x[LB_DPAD];
    extern __shared__ float4 s_hist4[];                      // the problem's history: [hist] y vectors, [hist] s vectors; then the work vectors
    float* const s_hist = (float*)s_hist4;
    const int p = blockIdx.x;
    if (p == 0 && threadIdx.x == 0 && n_active_next) *n_active_next = 0;      // the NEXT round's counter (nobody adds to it in this launch)
    LbfgsScalars s = S[p];
    if (s.phase == LB_DONE) return;
    float* const w = W + (size_t)p * lbfgs_ws_floats(cf.hist);
    float* const ro = RO + (size_t)p * LB_HMAX;
    // All four waves bring the history into LDS in one burst (whether this round ends a line search and needs it is only known
    // later; read pair by pair when needed, it was 26 dependent round trips of a lone wave).  The ring is unrolled on the way: LDS
    // position i holds the i-th oldest pair, so that the recursion below walks positions and does no ring arithmetic.
    const int nh0 = __builtin_amdgcn_readfirstlane(s.nh), h00 = __builtin_amdgcn_readfirstlane(s.h0);
    {
        const float4* const gy = (const float4*)(w + (size_t)LV_NUM * LB_DPAD);
        const float4* const gs = (const float4*)(w + (size_t)(LV_NUM + cf.hist) * LB_DPAD);
        const int n4 = nh0 * (LB_DPAD / 4), so = cf.hist * (LB_DPAD / 4);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * LB_NT) {
            float4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(i0 + k * LB_NT, n4 - 1);
                int slot = h00 + (i >> 5);
                slot = slot >= cf.hist ? slot - cf.hist : slot;
                a[k] = gy[slot * (LB_DPAD / 4) + (i & 31)]; b[k] = gs[slot * (LB_DPAD / 4) + (i & 31)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = i0 + k * LB_NT; if (i < n4) { s_hist4[i] = a[k]; s_hist4[so + i] = b[k]; } }
        }
        if ((int)threadIdx.x < nh0) { const int slot = h00 + threadIdx.x; s_ro[threadIdx.x] = ro[slot >= cf.hist ? slot - cf.hist : slot]; }
        // ... and the work vectors and the x row with it: read one by one along the state machine they were another six to
        // eight dependent round trips
        if (threadIdx.x < LV_NUM * (LB_DPAD / 4)) s_hist4[2 * so + threadIdx.x] = ((const float4*)w)[threadIdx.x];
        if (threadIdx.x < LB_DPAD) s_x[threadIdx.x] = (int)threadIdx.x < cf.dim ? X[(size_t)p * x_stride + threadIdx.x] : 0.f;
    }
    const int lane = threadIdx.x & 63;
    const float* const gr = G + (size_t)p * g_stride;
    const float f_new = F[p];
    LV g_new = {lane < cf.dim ? gr[lane] : 0.f, lane + 64 < cf.dim ? gr[lane + 64] : 0.f};
    if (fold.part) {
        auto addend = [&](int col) -> float {
            const int c = col - fold.col0;
            if (c < 0 || c >= fold.n) return 0.f;
            const float* const e = fold.part + (size_t)p * fold.n + c;
            return (e[0] + e[fold.stride]) + (e[2 * fold.stride] + e[3 * fold.stride]);
        };
        if (lane >= fold.col0 - 64 && lane < fold.col0 + fold.n) {       // (the few lanes that hold such a column)
            const float a0 = addend(lane), a1 = addend(lane + 64);
            if (lane - fold.col0 >= 0 && lane - fold.col0 < fold.n) g_new.a += a0;
            if (lane + 64 - fold.col0 >= 0 && lane + 64 - fold.col0 < fold.n) g_new.b += a1;
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;                           // the state machine itself is one wave's work
    float* const x = X + (size_t)p * x_stride;
    const float c1 = 1e-4f, c2 = 0.9f;
    float* const s_work = s_hist + (size_t)2 * cf.hist * LB_DPAD;        // the LV_NUM work vectors' copies in LDS
    auto wld = [&](int slot) -> LV { return {s_work[slot * LB_DPAD + lane], s_work[slot * LB_DPAD + lane + 64]}; };
    auto wst = [&](int slot, LV v) {                                     // to HBM for the next launch, to LDS for this one
        lv_st(w + slot * LB_DPAD, v);
        s_work[slot * LB_DPAD + lane] = v.a; s_work[slot * LB_DPAD + lane + 64] = v.b;
    };

    enum { STEP_BEGIN, ITER_BEGIN, BRACKET_EVALD, ZOOM_ENTER, ZOOM_LOOP, ZOOM_EVALD, ZOOM_EXIT, LS_DONE, STEP_END };
    int st;
    LV g, d;                                                 // the accepted point's gradient; the search direction
    if (s.phase == LB_INIT) {
        s.loss = f_new; g = g_new; wst(LV_G, g);
        d = {0.f, 0.f};
        s.evals_total = 1; s.step = 0; s.n_iter_total = 0; s.nh = 0; s.h0 = 0; s.prev_orig = f_new;
        st = STEP_BEGIN;
    } else {
        g = wld(LV_G);
        d = wld(LV_D);
        st = s.phase == LB_WAIT_BRACKET ? BRACKET_EVALD : ZOOM_EVALD;
    }
    auto write_point = [&](float t) {                        // x = x_init + t d
        const LV xi = wld(LV_XINIT);
        const float xa = fmaf(t, d.a, xi.a), xb = fmaf(t, d.b, xi.b);
        if (lane < cf.dim) x[lane] = xa;
        if (lane + 64 < cf.dim) x[lane + 64] = xb;
        s_x[lane] = xa; s_x[lane + 64] = xb;                 // (the next line search of this launch starts from here)
    };
    auto set_low_high = [&]() { s.low = s.bf0 <= s.bf1 ? 0 : 1; };
    auto ring = [&](int i) { return i >= cf.hist ? i - cf.hist : i; };      // slot of ring position i < 2 hist (no integer division)

    // the two-loop recursion's pairs as LDS positions: lo .. hi in age order, and (once the ring is full) the pair pushed in this
    // launch at position pnew = 0, where the oldest was
    int lo = 0, hi = nh0 - 1, pnew = -1;
    bool running = true;
    while (running) {
        switch (st) {
        case STEP_BEGIN: {                                   // one optimizer.step(): the closure's value here is the accepted loss
            s.orig_loss = s.loss; s.n_inner = 0; s.evals_step = 1;
            st = lv_absmax(g) <= cf.tol_grad ? STEP_END : ITER_BEGIN;
        } break;
        case ITER_BEGIN: {
            if (s.n_inner >= cf.max_iter) { st = STEP_END; break; }
            s.n_inner++; s.n_iter_total++;
            if (s.n_iter_total == 1) {
                d = {-g.a, -g.b}; s.nh = 0; s.h0 = 0; s.H_diag = 1.f;
            } else {
                const LV pg = wld(LV_PREVG);
                const LV y = {g.a - pg.a, g.b - pg.b}, sv = {d.a * s.t, d.b * s.t};
                const float ys = lv_dot(y, sv);
                if (ys > 1e-10f) {                                       // (at most once per launch: a second direction here has y = 0)
                    const bool full = s.nh == cf.hist;
                    if (full) { s.h0 = ring(s.h0 + 1); s.nh--; }          // the oldest pair leaves
                    const int slot = ring(s.h0 + s.nh);                   // where the pair lives in HBM
                    lv_st(w + (size_t)(LV_NUM + slot) * LB_DPAD, y);
                    lv_st(w + (size_t)(LV_NUM + cf.hist + slot) * LB_DPAD, sv);
                    int pos;                                              // ... and in LDS
                    if (full) { pos = 0; pnew = 0; lo = 1; } else { pos = ++hi; }
                    s_hist[pos * LB_DPAD + lane] = y.a; s_hist[pos * LB_DPAD + lane + 64] = y.b;
                    s_hist[(cf.hist + pos) * LB_DPAD + lane] = sv.a; s_hist[(cf.hist + pos) * LB_DPAD + lane + 64] = sv.b;
                    const float r = 1.f / ys;
                    s_ro[pos] = r;
                    if (lane == 0) ro[slot] = r;
                    s.nh++;
                    s.H_diag = ys / lv_dot(y, y);
                }
                // The two-loop recursion: a chain of 2 nh dependent dot products, newest pair to oldest and back.  A lone wave issues
                // an instruction every ~5 cycles whatever its kind, so what counts is how few there are per pair: positions, no ring.
                lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi); pnew = __builtin_amdgcn_readfirstlane(pnew);
                LV q = {-g.a, -g.b};
                auto down = [&](int pos) {
                    const LV yy = {s_hist[pos * LB_DPAD + lane], s_hist[pos * LB_DPAD + lane + 64]};
                    const LV sy = {s_hist[(cf.hist + pos) * LB_DPAD + lane], s_hist[(cf.hist + pos) * LB_DPAD + lane + 64]};
                    const float al = lv_dot(sy, q) * s_ro[pos];
                    s_al[pos] = al;
                    q = {fmaf(-al, yy.a, q.a), fmaf(-al, yy.b, q.b)};
                };
                if (pnew >= 0) down(pnew);
                for (int pos = hi; pos >= lo; --pos) down(pos);          // (reading a pair's LDS words one pair ahead: measured slower)
                LV r = {q.a * s.H_diag, q.b * s.H_diag};
                auto up = [&](int pos) {
                    const LV yy = {s_hist[pos * LB_DPAD + lane], s_hist[pos * LB_DPAD + lane + 64]};
                    const LV sy = {s_hist[(cf.hist + pos) * LB_DPAD + lane], s_hist[(cf.hist + pos) * LB_DPAD + lane + 64]};
                    const float be = lv_dot(yy, r) * s_ro[pos];
                    const float c = s_al[pos] - be;
                    r = {fmaf(c, sy.a, r.a), fmaf(c, sy.b, r.b)};
                };
                for (int pos = lo; pos <= hi; ++pos) up(pos);
                if (pnew >= 0) up(pnew);
                d = r;
            }
            wst(LV_PREVG, g);
            s.prev_loss = s.loss;
            s.t = s.n_iter_total == 1 ? fminf(1.f, 1.f / lv_abssum(g)) * cf.lr : cf.lr;
            s.gtd = lv_dot(g, d);
            wst(LV_D, d);
            if (s.gtd > -cf.tol_change) { st = STEP_END; break; }
            // the line search starts: remember where from
            wst(LV_XINIT, LV{s_x[lane], s_x[lane + 64]});
            s.d_norm = lv_absmax(d);
            s.t_prev = 0.f; s.f_prev = s.loss; s.gtd_prev = s.gtd;
            wst(LV_GPREV, g);
            s.ls_iter = 0; s.ls_evals = 0;
            write_point(s.t);
            s.phase = LB_WAIT_BRACKET; running = false;
        } break;
        case BRACKET_EVALD: {
            s.ls_evals++; s.evals_total++;
            const float gtd_new = lv_dot(g_new, d);
            if (s.ls_iter >= cf.max_ls) {                    // out of evaluations while still extrapolating
                s.b0 = 0.f; s.b1 = s.t; s.bf0 = s.loss; s.bf1 = f_new; s.bgtd0 = s.gtd; s.bgtd1 = gtd_new;
                wst(LV_BG0, g); wst(LV_BG1, g_new);
                st = ZOOM_ENTER; break;
            }
            const bool up = f_new > s.loss + (c1 * s.t) * s.gtd || (s.ls_iter > 1 && f_new >= s.f_prev);
            const bool wolfe = !up && fabsf(gtd_new) <= -c2 * s.gtd;
            const bool rising = !up && !wolfe && gtd_new >= 0.f;
            if (up || rising) {
                s.b0 = s.t_prev; s.b1 = s.t; s.bf0 = s.f_prev; s.bf1 = f_new; s.bgtd0 = s.gtd_prev; s.bgtd1 = gtd_new;
                wst(LV_BG0, wld(LV_GPREV)); wst(LV_BG1, g_new);
                st = ZOOM_ENTER;
            } else if (wolfe) {                              // accepted where it stands
                s.loss = f_new; g = g_new; wst(LV_G, g);
                st = LS_DONE;
            } else {
                const float min_step = s.t + 0.01f * (s.t - s.t_prev), max_step = s.t * 10.f;
                const float tn = lb_cubic(s.t_prev, s.f_prev, s.gtd_prev, s.t, f_new, gtd_new, true, min_step, max_step);
                s.t_prev = s.t; s.f_prev = f_new; s.gtd_prev = gtd_new;
                wst(LV_GPREV, g_new);
                s.t = tn; s.ls_iter++;
                write_point(s.t);
                s.phase = LB_WAIT_BRACKET; running = false;
            }
        } break;
        case ZOOM_ENTER: {
            s.insuf = 0; set_low_high();
            st = ZOOM_LOOP;
        } break;
        case ZOOM_LOOP: {
            if (s.ls_iter >= cf.max_ls || fabsf(s.b1 - s.b0) * s.d_norm < cf.tol_change) { st = ZOOM_EXIT; break; }
            float t = lb_cubic(s.b0, s.bf0, s.bgtd0, s.b1, s.bf1, s.bgtd1, false, 0.f, 0.f);
            const float bmax = fmaxf(s.b0, s.b1), bmin = fminf(s.b0, s.b1), eps = 0.1f * (bmax - bmin);
            if (fminf(bmax - t, t - bmin) < eps) {           // too close to an end of the bracket
                if (s.insuf || t >= bmax || t <= bmin) {
                    t = fabsf(t - bmax) < fabsf(t - bmin) ? bmax - eps : bmin + eps;
                    s.insuf = 0;
                } else s.insuf = 1;
            } else s.insuf = 0;
            s.t = t; s.ls_iter++;
            write_point(t);
            s.phase = LB_WAIT_ZOOM; running = false;
        } break;
        case ZOOM_EVALD: {
            s.ls_evals++; s.evals_total++;
            const float gtd_new = lv_dot(g_new, d), t = s.t;
            const float b_low = s.low ? s.b1 : s.b0, bf_low = s.low ? s.bf1 : s.bf0, bgtd_low = s.low ? s.bgtd1 : s.bgtd0;
            const float b_high = s.low ? s.b0 : s.b1;
            const int BGlow = s.low ? LV_BG1 : LV_BG0, BGhigh = s.low ? LV_BG0 : LV_BG1;
            if (f_new > s.loss + (c1 * t) * s.gtd || f_new >= bf_low) {            // the new point replaces the high end
                if (s.low) { s.b0 = t; s.bf0 = f_new; s.bgtd0 = gtd_new; } else { s.b1 = t; s.bf1 = f_new; s.bgtd1 = gtd_new; }
                wst(BGhigh, g_new);
                set_low_high();
                st = ZOOM_LOOP;
            } else {
                const bool done = fabsf(gtd_new) <= -c2 * s.gtd;
                if (!done && gtd_new * (b_high - b_low) >= 0.f) {                   // the old low end becomes the high end
                    if (s.low) { s.b0 = b_low; s.bf0 = bf_low; s.bgtd0 = bgtd_low; } else { s.b1 = b_low; s.bf1 = bf_low; s.bgtd1 = bgtd_low; }
                    wst(BGhigh, wld(BGlow));
                }
                if (s.low) { s.b1 = t; s.bf1 = f_new; s.bgtd1 = gtd_new; } else { s.b0 = t; s.bf0 = f_new; s.bgtd0 = gtd_new; }
                wst(BGlow, g_new);
                st = done ? ZOOM_EXIT : ZOOM_LOOP;
            }
        } break;
        case ZOOM_EXIT: {                                    // the low end of the bracket is the step taken
            s.t = s.low ? s.b1 : s.b0;
            s.loss = s.low ? s.bf1 : s.bf0;
            g = wld(s.low ? LV_BG1 : LV_BG0);
            wst(LV_G, g);
            write_point(s.t);
            st = LS_DONE;
        } break;
        case LS_DONE: {
            s.evals_step += s.ls_evals;
            const bool stop = s.n_inner == cf.max_iter || s.evals_step >= cf.max_eval || lv_absmax(g) <= cf.tol_grad ||
                              s.d_norm * fabsf(s.t) <= cf.tol_change || fabsf(s.loss - s.prev_loss) < cf.tol_change;
            st = stop ? STEP_END : ITER_BEGIN;
        } break;
        case STEP_END: {                                     // SMPLify-X's loop around optimizer.step()
            bool stop = !(fabsf(s.orig_loss) <= 3.0e38f);    // NaN or infinite
            if (!stop && s.step > 0 && cf.ftol > 0.f) {
                const float rel = fabsf(s.prev_orig - s.orig_loss) / fmaxf(fmaxf(fabsf(s.prev_orig), fabsf(s.orig_loss)), 1.f);
                stop = rel <= cf.ftol;
            }
            if (!stop && lv_absmax(g) < cf.gtol) stop = true;
            s.prev_orig = s.orig_loss; s.step++;
            if (stop || s.step >= cf.max_steps) { s.phase = LB_DONE; running = false; }
            else st = STEP_BEGIN;
        } break;
        }
    }
    if (lane == 0) {
        S[p] = s;
        if (s.phase != LB_DONE && n_active) atomicAdd(n_active, 1);
    }
}

// A caller that stops asking before every problem has finished (a round budget): a problem still inside a line search has a TRIAL
// point in its x row (bracketing extrapolates up to 10x per step), not a point it accepted.  Put the line search's starting point
// back -- the point `loss` in the scalars belongs to -- and mark the problem finished; n_unfinished += 1 per problem touched.
__global__ __launch_bounds__(LB_DPAD) void lbfgs_finalize_kernel(LbfgsCfg cf, LbfgsScalars* __restrict__ S, const float* __restrict__ W,
                                                               float* __restrict__ X, int x_stride, int* __restrict__ n_unfinished) {
    const int p = blockIdx.x;
    const int phase = S[p].phase;
    if (phase == LB_DONE) return;
    if (phase == LB_WAIT_BRACKET || phase == LB_WAIT_ZOOM) {
        const float* xi = W + (size_t)p * lbfgs_ws_floats(cf.hist) + (size_t)LV_XINIT * LB_DPAD;
        if ((int)threadIdx.x < cf.dim) X[(size_t)p * x_stride + threadIdx.x] = xi[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S[p].phase = LB_DONE;
        if (n_unfinished) atomicAdd(n_unfinished, 1);
    }
}
#endif

}  // namespace fdc
