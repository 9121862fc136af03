// Scene registration on the device (fdcap_set_scene, r6): the k-d cell order, the boxes of every level, the chunk-centred bf16
// MFMA fragments and the inverse permutation that the in-loop Chamfer search (fdc_chamfer.h, nn_stream4_kernel) prunes with.
// Replaces what the reference does with `scene.repeat(N,1,1)` at /root/reference/global_optimization.py:173-176 -- the scene is
// stored once, sorted into cells.  Until r5 this was a single-threaded std::nth_element recursion on the host (0.16 s at 500 k
// points, 1.0 s at 2 M: more than the fits it served).  Results of the search never depend on the order (pruning only).
//
// The order, as a specification (the host restatement `scene_order_host` below follows the same rules; tests compare the two):
//   * a node of n > 32 points is cut along the LONGEST axis of its bounding box (ties: lower axis) into the nleft points that come
//     first by (coordinate, original index) and the rest; nleft = min(n - 1, (units / 2) * unit) with unit = 512 points (one cell)
//     while n > 512 and 32 points (one MFMA tile) below, units = ceil(n / unit);
//   * inside a 32-point tile points stay in input order.
// Device algorithm: three index lists sorted once by (x, i), (y, i), (z, i) (LSD radix sort) + the list in input order; every
// level of the tree is one stable partition of all four lists by a per-point side flag, nodes side by side (a node's box = first
// and last entry of its part of each sorted list; the flag = rank in the chosen axis' list against nleft).  O(n) per level, no
// data-dependent host work: the tree's SHAPE (starts, sizes, nleft) depends on n alone and is tabulated on the host.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "fdc_chamfer.h"

namespace fdc {

constexpr int SC_RS_ROUNDS = 16;                 // radix pass: a 256-thread block scatters 16 rounds of 256 keys
constexpr int SC_RS_TILE = 256 * SC_RS_ROUNDS;
constexpr int SC_PT_TILE = 1024;                 // partition pass: 256 threads x 4 consecutive positions
constexpr int SC_SUPER = 16;                     // cells per super-cell box (= ST4_SUPER of the search; asserted where both are visible)

// (coordinate, index) order as unsigned keys: -0 == +0 (the host comparator's `==`), negative values reversed
__device__ __forceinline__ unsigned sc_key(float v) {
    unsigned u = __float_as_uint(v);
    if ((u << 1) == 0u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void sc_keys_kernel(const float* __restrict__ xyz, int n, int axis, unsigned* __restrict__ keys,
                                                      int* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = sc_key(xyz[3 * (size_t)i + axis]); vals[i] = i; }
}

// ---- LSD radix sort of (key, value) pairs, 8 bits per pass, stable -----------------------------------------------------------
__global__ __launch_bounds__(256) void sc_rs_hist_kernel(const unsigned* __restrict__ keys, int n, int shift, unsigned* __restrict__ hist, int nb) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SC_RS_TILE;
    for (int r = 0; r < SC_RS_ROUNDS; ++r) {
        const int i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of m unsigned entries in place, one 1024-thread workgroup per segment (blockIdx.x: segment of stride m)
__global__ __launch_bounds__(1024) void sc_scan_kernel(unsigned* __restrict__ a, int m) {
    __shared__ unsigned part[1024];
    a += (size_t)blockIdx.x * m;
    const int per = (m + 1023) / 1024;
    const int lo = min(m, (int)threadIdx.x * per), hi = min(m, lo + per);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        unsigned t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) { unsigned v = a[i]; a[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void sc_rs_scatter_kernel(const unsigned* __restrict__ kin, const int* __restrict__ vin,
                                                            unsigned* __restrict__ kout, int* __restrict__ vout, int n, int shift,
                                                            const unsigned* __restrict__ hist, int nb) {
    __shared__ unsigned base[256];
    __shared__ unsigned wcnt[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    base[tid] = hist[(size_t)tid * nb + blockIdx.x];
    const int tile0 = blockIdx.x * SC_RS_TILE;
    for (int r = 0; r < SC_RS_ROUNDS; ++r) {
        const int i = tile0 + r * 256 + tid;
        const bool valid = i < n;
        const unsigned key = valid ? kin[i] : 0u;
        const int val = valid ? vin[i] : 0;
        const unsigned d = (key >> shift) & 255u;
        unsigned long long m = __ballot(valid);                       // lanes of this wave with the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        for (int k = 0; k < 4; ++k) wcnt[k][tid] = 0;
        __syncthreads();
        if (valid && rank == 0) wcnt[wave][d] = (unsigned)__popcll(m);
        __syncthreads();
        if (valid) {
            unsigned off = base[d];
            for (int k = 0; k < wave; ++k) off += wcnt[k][d];
            kout[off + rank] = key;
            vout[off + rank] = val;
        }
        __syncthreads();
        base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
        __syncthreads();
    }
}

// ---- one level of the tree ----------------------------------------------------------------------------------------------------
// node table of a level, ascending by start and covering [0, n): {start, size, nleft, side-1 points of all earlier nodes};
// a leaf has nleft == size (everything stays)
__device__ __forceinline__ int sc_find_node(const int4* __restrict__ nodes, int nn, int p) {
    int lo = 0, hi = nn - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (nodes[mid].x <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sc_axis_kernel(const int4* __restrict__ nodes, int nn, const float* __restrict__ xyz,
                                                      const int* __restrict__ Lx, const int* __restrict__ Ly, const int* __restrict__ Lz,
                                                      unsigned char* __restrict__ axis) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const int4 nd = nodes[i];
    int ax = 0;
    if (nd.z < nd.y) {
        const int a = nd.x, b = nd.x + nd.y - 1;
        const float ex = xyz[3 * (size_t)Lx[b]] - xyz[3 * (size_t)Lx[a]];
        const float ey = xyz[3 * (size_t)Ly[b] + 1] - xyz[3 * (size_t)Ly[a] + 1];
        const float ez = xyz[3 * (size_t)Lz[b] + 2] - xyz[3 * (size_t)Lz[a] + 2];
        float best = ex;
        if (ey > best) { ax = 1; best = ey; }
        if (ez > best) { ax = 2; }
    }
    axis[i] = (unsigned char)ax;
}

__global__ __launch_bounds__(256) void sc_side_kernel(const int4* __restrict__ nodes, int nn, const unsigned char* __restrict__ axis,
                                                      const int* __restrict__ Lx, const int* __restrict__ Ly, const int* __restrict__ Lz,
                                                      int n, unsigned char* __restrict__ side) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int nd = sc_find_node(nodes, nn, p);
    const int4 d = nodes[nd];
    const int ax = axis[nd];
    const int id = (ax == 0 ? Lx : ax == 1 ? Ly : Lz)[p];
    side[id] = (unsigned char)((p - d.x) >= d.z);
}

struct ScLists { const int* in[4]; int* out[4]; };

__device__ __forceinline__ int sc_wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}

// side-1 points per SC_PT_TILE positions of every list: cnt[l * nb + block]
__global__ __launch_bounds__(256) void sc_part_count_kernel(ScLists L, const unsigned char* __restrict__ side, int n, unsigned* __restrict__ cnt, int nb) {
    __shared__ unsigned tot[4];
    if (threadIdx.x < 4) tot[threadIdx.x] = 0;
    __syncthreads();
    const int p0 = blockIdx.x * SC_PT_TILE + threadIdx.x * 4;
    for (int l = 0; l < 4; ++l) {
        int c = 0;
        for (int k = 0; k < 4; ++k) if (p0 + k < n) c += side[L.in[l][p0 + k]];
        const int s = sc_wave_incl_scan(c);
        if ((threadIdx.x & 63) == 63) atomicAdd(&tot[l], (unsigned)s);
    }
    __syncthreads();
    if (threadIdx.x < 4) cnt[(size_t)threadIdx.x * nb + blockIdx.x] = tot[threadIdx.x];
}

// stable partition of every list inside every node: side-0 points keep their order in [start, start + nleft), side-1 behind them
__global__ __launch_bounds__(256) void sc_part_scatter_kernel(ScLists L, const unsigned char* __restrict__ side, int n,
                                                              const int4* __restrict__ nodes, int nn, const unsigned* __restrict__ cnt, int nb) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int p0 = blockIdx.x * SC_PT_TILE + tid * 4;
    int4 nd[4];
    for (int k = 0; k < 4; ++k) nd[k] = nodes[sc_find_node(nodes, nn, min(p0 + k, n - 1))];
    for (int l = 0; l < 4; ++l) {
        int id[4], s[4], c = 0;
        for (int k = 0; k < 4; ++k) {
            id[k] = p0 + k < n ? L.in[l][p0 + k] : 0;
            s[k] = p0 + k < n ? side[id[k]] : 0;
            c += s[k];
        }
        const int inc = sc_wave_incl_scan(c);
        __syncthreads();
        if ((tid & 63) == 63) wsum[wave] = inc;
        __syncthreads();
        int ones = (int)cnt[(size_t)l * nb + blockIdx.x] + inc - c;          // side-1 points of this list before position p0
        for (int k = 0; k < wave; ++k) ones += wsum[k];
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k;
            if (p < n) {
                const int in_node = ones - nd[k].w;                            // side-1 points of the node before p
                const int dst = s[k] ? nd[k].x + nd[k].z + in_node : nd[k].x + (p - nd[k].x - in_node);
                L.out[l][dst] = id[k];
            }
            ones += s[k];
        }
    }
}

// ---- tables of the sorted scene -------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sc_bf16(float f) {                       // round to nearest even, as the host build did
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float sc_bf16f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float sc_pad(float lo, float hi) {                // 1e-6 + 1e-6 max(|lo|, |hi|), no contraction
    return __fadd_rn(1e-6f, __fmul_rn(1e-6f, fmaxf(fabsf(lo), fabsf(hi))));
}

// one workgroup per cell of MF_CH points: sorted points, inverse permutation, input-order copy, the cell's and its quarters' boxes,
// centre + radius, fragments (see nn_stream4_kernel for the layout)
__global__ __launch_bounds__(MF_CH) void sc_finalize_kernel(const float* __restrict__ xyz, const int* __restrict__ order, int n,
                                                            float4* __restrict__ orig, float4* __restrict__ sorted, int* __restrict__ inv,
                                                            float4* __restrict__ bounds, float4* __restrict__ qbounds,
                                                            uint4* __restrict__ frags, float4* __restrict__ centers) {
    static_assert(MF_CH % 256 == 0 && MF_CH / 4 % 64 == 0, "a quarter cell is a whole number of waves");
    constexpr int NW = MF_CH / 64, WPQ = NW / 4;
    __shared__ float wlo[NW][3], whi[NW][3], wr2[NW];
    __shared__ float cen[3];
    const int ch = blockIdx.x, j = threadIdx.x, wave = j >> 6, lane = j & 63;
    const int p = ch * MF_CH + j;
    const bool valid = p < n;
    float x = 0.f, y = 0.f, z = 0.f;
    int idx = 0;
    if (valid) {
        idx = order[p];
        x = xyz[3 * (size_t)idx]; y = xyz[3 * (size_t)idx + 1]; z = xyz[3 * (size_t)idx + 2];
        const float4 pt = make_float4(x, y, z, __int_as_float(idx));
        sorted[p] = pt;
        orig[idx] = pt;
        inv[idx] = p;
    }
    float lo[3] = {valid ? x : INFINITY, valid ? y : INFINITY, valid ? z : INFINITY};
    float hi[3] = {valid ? x : -INFINITY, valid ? y : -INFINITY, valid ? z : -INFINITY};
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
    if (lane == 0) for (int k = 0; k < 3; ++k) { wlo[wave][k] = lo[k]; whi[wave][k] = hi[k]; }
    __syncthreads();
    if (j < 4) {                                                              // quarter boxes; an empty quarter: (+inf, +inf)
        float ql[3], qh[3];
        for (int k = 0; k < 3; ++k) {
            ql[k] = INFINITY; qh[k] = -INFINITY;
            for (int w = 0; w < WPQ; ++w) { ql[k] = fminf(ql[k], wlo[j * WPQ + w][k]); qh[k] = fmaxf(qh[k], whi[j * WPQ + w][k]); }
        }
        if (ch * MF_CH + j * (MF_CH / 4) < n)
            for (int k = 0; k < 3; ++k) { const float pad = sc_pad(ql[k], qh[k]); ql[k] -= pad; qh[k] += pad; }
        else
            for (int k = 0; k < 3; ++k) qh[k] = INFINITY;
        qbounds[2 * ((size_t)ch * 4 + j)] = make_float4(ql[0], ql[1], ql[2], 0.f);
        qbounds[2 * ((size_t)ch * 4 + j) + 1] = make_float4(qh[0], qh[1], qh[2], 0.f);
    }
    if (j == 0) {                                                             // the cell's box (never empty) and its centre
        float cl[3], chh[3];
        for (int k = 0; k < 3; ++k) {
            cl[k] = INFINITY; chh[k] = -INFINITY;
            for (int w = 0; w < NW; ++w) { cl[k] = fminf(cl[k], wlo[w][k]); chh[k] = fmaxf(chh[k], whi[w][k]); }
            const float pad = sc_pad(cl[k], chh[k]);
            cl[k] -= pad; chh[k] += pad;
            cen[k] = __fmul_rn(0.5f, __fadd_rn(cl[k], chh[k]));
        }
        bounds[2 * (size_t)ch] = make_float4(cl[0], cl[1], cl[2], 0.f);
        bounds[2 * (size_t)ch + 1] = make_float4(chh[0], chh[1], chh[2], 0.f);
    }
    __syncthreads();
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    float yx = 0.f, yy = 0.f, yz = 0.f, n2 = 1e30f, r2 = 0.f;                  // padding rows: score 1e30
    if (valid) {
        yx = x - cx; yy = y - cy; yz = z - cz;
        n2 = __fadd_rn(__fmul_rn(yz, yz), __fadd_rn(__fmul_rn(yy, yy), __fmul_rn(yx, yx)));
        r2 = n2;
    }
    unsigned hx = sc_bf16(yx), hy = sc_bf16(yy), hz = sc_bf16(yz);
    unsigned lx = sc_bf16(yx - sc_bf16f(hx)), ly = sc_bf16(yy - sc_bf16f(hy)), lz = sc_bf16(yz - sc_bf16f(hz));
    // the score's factor -2 (|y|^2 - 2 x.y) rides on the static side: exact in bf16
    hx = sc_bf16(-2.f * sc_bf16f(hx)); hy = sc_bf16(-2.f * sc_bf16f(hy)); hz = sc_bf16(-2.f * sc_bf16f(hz));
    lx = sc_bf16(-2.f * sc_bf16f(lx)); ly = sc_bf16(-2.f * sc_bf16f(ly)); lz = sc_bf16(-2.f * sc_bf16f(lz));
    const unsigned nh = sc_bf16(n2);
    const float r1 = n2 - sc_bf16f(nh);
    const unsigned nm = sc_bf16(r1), nl = sc_bf16(r1 - sc_bf16f(nm));
    uint4* t = frags + ((size_t)ch * (MF_CH / 32) + (j >> 5)) * 64;
    t[j & 31] = make_uint4(hx | (hx << 16), lx | (lx << 16), hy | (hy << 16), ly | (ly << 16));
    t[32 + (j & 31)] = make_uint4(hz | (hz << 16), lz | (lz << 16), nh | (nm << 16), nl);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
    if (lane == 0) wr2[wave] = r2;
    __syncthreads();
    if (j == 0) {
        float m = 0.f;
        for (int w = 0; w < NW; ++w) m = fmaxf(m, wr2[w]);
        centers[ch] = make_float4(cx, cy, cz, __fadd_rn(__fmul_rn(__fsqrt_rn(m), 1.00001f), 1e-6f));
    }
}

// boxes of SC_SUPER consecutive cells
__global__ __launch_bounds__(256) void sc_super_kernel(const float4* __restrict__ bounds, int nchunk, float4* __restrict__ sb, int nsuper) {
    const int su = blockIdx.x * 256 + threadIdx.x;
    if (su >= nsuper) return;
    float4 lo = make_float4(1e30f, 1e30f, 1e30f, 0.f), hi = make_float4(-1e30f, -1e30f, -1e30f, 0.f);
    for (int ch = su * SC_SUPER; ch < min(nchunk, (su + 1) * SC_SUPER); ++ch) {
        const float4 a = bounds[2 * (size_t)ch], b = bounds[2 * (size_t)ch + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
    }
    sb[2 * (size_t)su] = lo; sb[2 * (size_t)su + 1] = hi;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
inline void sc_split(int64_t n, int64_t* nleft) {                             // the cut of a node of n > 32 points
    const int64_t unit = n > MF_CH ? MF_CH : 32;
    const int64_t units = (n + unit - 1) / unit;
    *nleft = std::min(n - 1, (units / 2) * unit);
}

// the tree's levels as node tables (shape depends on n alone): level l = the nodes at depth l, leaves carried down unchanged
inline void sc_level_tables(int64_t ns, std::vector<int4>* nodes, std::vector<int>* level_off) {
    nodes->clear(); level_off->clear();
    std::vector<std::pair<int64_t, int64_t>> cur, nxt;                         // (start, size)
    if (ns > 0) cur.push_back({0, ns});
    for (;;) {
        bool any = false;
        for (auto& nd : cur) any = any || nd.second > 32;
        if (!any) break;
        level_off->push_back((int)nodes->size());
        int64_t ones = 0;
        nxt.clear();
        for (auto& nd : cur) {
            const int64_t a = nd.first, n = nd.second;
            int64_t nl = n;
            if (n > 32) sc_split(n, &nl);
            nodes->push_back(make_int4((int)a, (int)n, (int)nl, (int)ones));
            ones += n - nl;
            if (nl < n) { nxt.push_back({a, nl}); nxt.push_back({a + nl, n - nl}); }
            else nxt.push_back({a, n});
        }
        cur.swap(nxt);
    }
    level_off->push_back((int)nodes->size());
}

// The same order on the host (the r5 build, kept as the specification the device build is tested against; FDCAP_SCENE_BUILD=host)
inline void scene_order_host(const float* xyz, int64_t ns, std::vector<int>& order) {
    order.resize((size_t)ns);
    for (int64_t i = 0; i < ns; ++i) order[i] = (int)i;
    std::vector<std::pair<int64_t, int64_t>> stack;
    if (ns > 0) stack.push_back({0, ns});
    while (!stack.empty()) {
        const int64_t a = stack.back().first, b = stack.back().second;
        stack.pop_back();
        const int64_t n = b - a;
        if (n <= 32) { std::sort(order.begin() + a, order.begin() + b); continue; }   // a tile: input order
        int64_t nleft;
        sc_split(n, &nleft);
        float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
        for (int64_t p = a; p < b; ++p)
            for (int k = 0; k < 3; ++k) { float v = xyz[3 * (int64_t)order[p] + k]; blo[k] = std::min(blo[k], v); bhi[k] = std::max(bhi[k], v); }
        int ax = 0;
        for (int k = 1; k < 3; ++k) if (bhi[k] - blo[k] > bhi[ax] - blo[ax]) ax = k;
        std::nth_element(order.begin() + a, order.begin() + a + nleft, order.begin() + b, [&](int i, int j) {
            const float vi = xyz[3 * (int64_t)i + ax], vj = xyz[3 * (int64_t)j + ax];
            return vi < vj || (vi == vj && i < j);
        });
        stack.push_back({a, a + nleft});
        stack.push_back({a + nleft, b});
    }
}

struct SceneTables {            // device pointers, sized by the caller: ns points, nchunk = ceil(ns / MF_CH) cells
    float4* orig; float4* sorted; int* inv; float4* bounds; float4* qbounds; float4* sbounds; uint4* frags; float4* centers;
};

// `order_host` != nullptr: take this order (the host build's) instead of sorting on the device.  Synchronous.
inline hipError_t scene_build_device(const float* xyz_host, int64_t ns64, const SceneTables& T, const int* order_host) {
#define SC_TRY(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { if (arena) (void)hipFree(arena); return e_; } } while (0)
    char* arena = nullptr;
    const int ns = (int)ns64;
    if (ns <= 0) return hipSuccess;
    const int nchunk = (ns + MF_CH - 1) / MF_CH, nsuper = (nchunk + SC_SUPER - 1) / SC_SUPER;
    const int nb_rs = (ns + SC_RS_TILE - 1) / SC_RS_TILE, nb_pt = (ns + SC_PT_TILE - 1) / SC_PT_TILE;
    std::vector<int4> nodes;
    std::vector<int> level_off;
    if (!order_host) sc_level_tables(ns, &nodes, &level_off);
    const int nlev = order_host ? 0 : (int)level_off.size() - 1;
    int max_nodes = 1;
    for (int l = 0; l < nlev; ++l) max_nodes = std::max(max_nodes, level_off[l + 1] - level_off[l]);
    // one allocation: xyz | 8 lists | 2 key arrays | side | radix histogram | partition counts | node tables | axis
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_xyz = 0, o_list = o_xyz + al((size_t)ns * 12), o_key = o_list + 8 * al((size_t)ns * 4),
                 o_side = o_key + 2 * al((size_t)ns * 4), o_hist = o_side + al((size_t)ns),
                 o_cnt = o_hist + al((size_t)256 * nb_rs * 4), o_nodes = o_cnt + al((size_t)4 * nb_pt * 4),
                 o_axis = o_nodes + al(std::max<size_t>(nodes.size(), 1) * sizeof(int4)), total = o_axis + al((size_t)max_nodes);
    SC_TRY(hipMalloc((void**)&arena, total));
    float* xyz = (float*)(arena + o_xyz);
    int* list[4][2];
    for (int l = 0; l < 4; ++l) for (int b = 0; b < 2; ++b) list[l][b] = (int*)(arena + o_list + (size_t)(2 * l + b) * al((size_t)ns * 4));
    unsigned* key[2] = {(unsigned*)(arena + o_key), (unsigned*)(arena + o_key + al((size_t)ns * 4))};
    unsigned char* side = (unsigned char*)(arena + o_side);
    unsigned* hist = (unsigned*)(arena + o_hist);
    unsigned* cnt = (unsigned*)(arena + o_cnt);
    int4* d_nodes = (int4*)(arena + o_nodes);
    unsigned char* axis = (unsigned char*)(arena + o_axis);
    hipStream_t st = 0;
    SC_TRY(hipMemcpyAsync(xyz, xyz_host, (size_t)ns * 12, hipMemcpyHostToDevice, st));
    const int* order = nullptr;
    if (order_host) {
        SC_TRY(hipMemcpyAsync(list[3][0], order_host, (size_t)ns * 4, hipMemcpyHostToDevice, st));
        order = list[3][0];
    } else {
        if (!nodes.empty()) SC_TRY(hipMemcpyAsync(d_nodes, nodes.data(), nodes.size() * sizeof(int4), hipMemcpyHostToDevice, st));
        const dim3 g256((ns + 255) / 256);
        for (int ax = 0; ax < 3; ++ax) {                                       // list[ax][0] <- indices by (coordinate ax, index)
            hipLaunchKernelGGL(sc_keys_kernel, g256, dim3(256), 0, st, xyz, ns, ax, key[0], list[ax][0]);
            for (int pass = 0; pass < 4; ++pass) {
                const int a = pass & 1, b = a ^ 1;
                hipLaunchKernelGGL(sc_rs_hist_kernel, dim3(nb_rs), dim3(256), 0, st, key[a], ns, 8 * pass, hist, nb_rs);
                hipLaunchKernelGGL(sc_scan_kernel, dim3(1), dim3(1024), 0, st, hist, 256 * nb_rs);
                hipLaunchKernelGGL(sc_rs_scatter_kernel, dim3(nb_rs), dim3(256), 0, st, key[a], list[ax][a], key[b], list[ax][b], ns, 8 * pass, hist, nb_rs);
            }
        }
        hipLaunchKernelGGL(sc_keys_kernel, g256, dim3(256), 0, st, xyz, ns, 0, key[0], list[3][0]);      // (input order; keys unused)
        int cur = 0;
        for (int l = 0; l < nlev; ++l) {
            const int4* nd = d_nodes + level_off[l];
            const int nn = level_off[l + 1] - level_off[l];
            hipLaunchKernelGGL(sc_axis_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, nd, nn, xyz, list[0][cur], list[1][cur], list[2][cur], axis);
            hipLaunchKernelGGL(sc_side_kernel, g256, dim3(256), 0, st, nd, nn, axis, list[0][cur], list[1][cur], list[2][cur], ns, side);
            ScLists L;
            for (int k = 0; k < 4; ++k) { L.in[k] = list[k][cur]; L.out[k] = list[k][cur ^ 1]; }
            hipLaunchKernelGGL(sc_part_count_kernel, dim3(nb_pt), dim3(256), 0, st, L, side, ns, cnt, nb_pt);
            hipLaunchKernelGGL(sc_scan_kernel, dim3(4), dim3(1024), 0, st, cnt, nb_pt);
            hipLaunchKernelGGL(sc_part_scatter_kernel, dim3(nb_pt), dim3(256), 0, st, L, side, ns, nd, nn, cnt, nb_pt);
            cur ^= 1;
        }
        order = list[3][cur];
    }
    hipLaunchKernelGGL(sc_finalize_kernel, dim3(nchunk), dim3(MF_CH), 0, st, xyz, order, ns, T.orig, T.sorted, T.inv, T.bounds, T.qbounds,
                       T.frags, T.centers);
    hipLaunchKernelGGL(sc_super_kernel, dim3((nsuper + 255) / 256), dim3(256), 0, st, T.bounds, nchunk, T.sbounds, nsuper);
    SC_TRY(hipGetLastError());
    SC_TRY(hipStreamSynchronize(st));
    (void)hipFree(arena);
    return hipSuccess;
#undef SC_TRY
}

}  // namespace fdc
