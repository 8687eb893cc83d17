// icpx_kd_lds.h — the kd-tree tie rule (icpx_kd.h) inside the fused pair kernels: the tree of a pair's target cloud
// built by the pair's own workgroup in LDS, and the wave-cooperative resolution of flagged queries against the
// target cloud staged in LDS.  Both run only for pairs in which a search flagged a tie (rare).
#pragma once

#include "icpx_filter.h"
#include "icpx_kd.h"

namespace icpx {

// LDS scratch of kd_build_lds for a cloud of m points: the D coordinate orders (u16), then the larger of the
// presort's (key, value) arrays and the level passes' arrays
inline __host__ __device__ size_t kd_build_bytes(int dim, int64_t m) {
    size_t npad = 2;
    while (static_cast<int64_t>(npad) < m) npad <<= 1;
    const size_t ord = (2 * static_cast<size_t>(dim) * static_cast<size_t>(m) + 15) & ~static_cast<size_t>(15);
    const size_t sort = 12 * npad, level = 8 * static_cast<size_t>(m) + 256;
    return ord + (sort > level ? sort : level);
}

// in-LDS bitonic sort of npad (power of two) (key, val) pairs, ascending lexicographic
ICPX_DEV void kd_bitonic_sort(uint64_t* key, uint32_t* val, int npad, int T) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t ka = key[i], kb = key[ixj];
                    const uint32_t va = val[i], vb = val[ixj];
                    const bool a_gt_b = ka > kb || (ka == kb && va > vb);
                    const bool up = (i & k) == 0;
                    if (a_gt_b == up) {
                        key[i] = kb;
                        key[ixj] = ka;
                        val[i] = vb;
                        val[ixj] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// segment of in-order position p after `levels` splits of [0, m): [lo, hi); a position that became a node on the way
// is its own segment
ICPX_DEV void kd_segment(const int p, const int m, const int levels, int& lo, int& hi) {
    lo = 0;
    hi = m;
    for (int l = 0; l < levels; l++) {
        const int mid = lo + ((hi - lo - 1) >> 1);
        if (p == mid) {
            lo = mid;
            hi = mid + 1;
            break;
        }
        if (p < mid) hi = mid;
        else lo = mid + 1;
    }
}

// Levels the passes run for m points: until every segment holds <= 3 points.  A segment of <= 3 points sorted along
// its level's axis IS its subtree in order (3: left, node, right; 2: node, right), so the last two levels need no pass.
ICPX_DEV int kd_pass_levels(const int m) {
    int k = 0;
    while ((m >> k) > 3) k++;
    return k;
}

// The level passes of the build: ord = the D coordinate orders of the m points ([D][m], u16 ids, each sorted by
// (coordinate, id)); level k splits along axis (axis0 + k) % D.  u: 8 m + 256 bytes of LDS.  Returns the axis whose
// order is the in-order layout of the subtree afterwards.  Workgroup-wide; ends with a barrier.
// seg (optional, [2][m] u16 of LDS): every position's segment bounds carried from level to level; without it they are
// recomputed by descent from the root (k steps per position and pass: the fused pair kernels, where LDS is scarce).
template <int D>
ICPX_DEV int kd_level_passes(uint16_t* ord, const int m, char* u, const int axis0, uint16_t* seg = nullptr) {
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* E = reinterpret_cast<uint32_t*>(u);          // [m] exclusive prefix of (left | median << 16) flags
    uint16_t* posA = reinterpret_cast<uint16_t*>(E + m);   // [m] position of a point in the split axis' order
    uint16_t* tmp = posA + m;                              // [m]
    int* wtot = reinterpret_cast<int*>(u + 8 * static_cast<size_t>(m));  // [waves]
    const int levels = kd_pass_levels(m);
    const int C = (m + T - 1) / T;
    const int p0 = min(tid * C, m), p1 = min(p0 + C, m);
    if (seg) {
        for (int p = p0; p < p1; p++) {
            seg[p] = 0;
            seg[m + p] = static_cast<uint16_t>(m);  // (m <= 65535)
        }
    }
    int a = axis0 % D;
    for (int k = 0; k < levels; k++, a = a + 1 == D ? 0 : a + 1) {
        for (int p = tid; p < m; p += T) posA[ord[a * m + p]] = static_cast<uint16_t>(p);
        __syncthreads();
        for (int b = 0; b < D; b++) {
            if (b == a) continue;
            int local = 0;
            for (int p = p0; p < p1; p++) {
                int lo, hi;
                if (seg) {
                    lo = seg[p];
                    hi = seg[m + p];
                } else {
                    kd_segment(p, m, k, lo, hi);
                }
                const int mid = lo + ((hi - lo - 1) >> 1);
                const int pa = posA[ord[b * m + p]];
                E[p] = static_cast<uint32_t>(local);
                local += (pa < mid ? 1 : 0) + (pa == mid ? 0x10000 : 0);
            }
            int incl = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                incl += lane >= o ? up : 0;
            }
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            int base = incl - local;
            for (int w = 0; w < wave; w++) base += wtot[w];
            for (int p = p0; p < p1; p++) E[p] += static_cast<uint32_t>(base);
            __syncthreads();
            for (int p = p0; p < p1; p++) {
                int lo, hi;
                if (seg) {
                    lo = seg[p];
                    hi = seg[m + p];
                } else {
                    kd_segment(p, m, k, lo, hi);
                }
                const int mid = lo + ((hi - lo - 1) >> 1);
                const int t = ord[b * m + p];
                const int pa = posA[t];
                const uint32_t e = E[p] - E[lo];
                const int nl = static_cast<int>(e & 0xffffu), nm = static_cast<int>(e >> 16);
                const int np = pa < mid ? lo + nl : (pa == mid ? mid : mid + 1 + (p - lo - nl - nm));
                tmp[np] = static_cast<uint16_t>(t);
            }
            __syncthreads();
            for (int p = tid; p < m; p += T) ord[b * m + p] = tmp[p];
            __syncthreads();
        }
        if (seg) {  // one more split of every own position's segment (own entries only: no barrier)
            for (int p = p0; p < p1; p++) {
                const int lo = seg[p], hi = seg[m + p];
                const int mid = lo + ((hi - lo - 1) >> 1);
                if (p == mid) {
                    seg[p] = static_cast<uint16_t>(mid);
                    seg[m + p] = static_cast<uint16_t>(mid + 1);
                } else if (p < mid) {
                    seg[m + p] = static_cast<uint16_t>(mid);
                } else {
                    seg[p] = static_cast<uint16_t>(mid + 1);
                }
            }
        }
    }
    return a;
}

// The tree of one cloud (m >= 1 points, m < 65536; AoS in global memory), built by the calling workgroup.
//   1. per axis: the cloud's order by (coordinate, index) — bitonic sort of (kd_ord_key, index);
//   2. level by level (axis = level % D): in the split axis' order every segment [lo, hi) is sorted, its median sits
//      at mid = lo + (hi - lo - 1) / 2, the points before it form the left child's segment [lo, mid), the points after
//      it the right child's; the other axes' orders are partitioned the same way, stably (a prefix count over the
//      whole array, taken relative to the segment's start), so that every segment stays sorted in every axis.
//      Segments are ranges of in-order positions and depend on m only (kd_segment).
// The passes stop when every segment holds <= 3 points (kd_pass_levels): the order along that level's axis then is the
// in-order layout.
// scratch: kd_build_bytes(D, m) of LDS, 16-byte aligned.  Workgroup-wide (barriers); ends with a barrier behind the
// global stores (node, pos).
template <int D, typename I>
ICPX_DEV void kd_build_lds(const double* __restrict__ pts, const int m, char* scratch, I* __restrict__ node,
                           I* __restrict__ pos, uint16_t* seg = nullptr) {
    const int T = blockDim.x, tid = threadIdx.x;
    int npad = 2;
    while (npad < m) npad <<= 1;
    uint16_t* ord = reinterpret_cast<uint16_t*>(scratch);  // [D][m]
    char* u = scratch + ((2 * static_cast<size_t>(D) * static_cast<size_t>(m) + 15) & ~static_cast<size_t>(15));
    uint64_t* skey = reinterpret_cast<uint64_t*>(u);
    uint32_t* sval = reinterpret_cast<uint32_t*>(skey + npad);
    for (int a = 0; a < D; a++) {
        for (int i = tid; i < npad; i += T) {
            skey[i] = i < m ? kd_ord_key(pts[static_cast<int64_t>(i) * D + a]) : ~0ull;
            sval[i] = i < m ? static_cast<uint32_t>(i) : 0xffffffffu;
        }
        __syncthreads();
        kd_bitonic_sort(skey, sval, npad, T);
        for (int p = tid; p < m; p += T) ord[a * m + p] = static_cast<uint16_t>(sval[p]);
        __syncthreads();
    }
    const int fa = kd_level_passes<D>(ord, m, u, 0, seg);  // (the sort arrays are free now)
    for (int p = tid; p < m; p += T) {
        const int t = ord[fa * m + p];
        node[p] = static_cast<I>(t);
        pos[t] = static_cast<I>(p);
    }
    __threadfence();
    __syncthreads();
}

// For every lane whose flag is set: the wave takes that lane's query (q, best = its minimal d2) and scans the m
// targets of the LDS view together (slot j = target perm[j] of the cloud, or L.j0 + j): tied set by sqrt(d2),
// smallest pre-order key, then the lowest cloud index among the targets coincident with that one.  slot: out, the
// LDS slot of the answer.
template <int D, typename I>
ICPX_DEV void kd_resolve_wave(const bool flag, const double (&q)[D], const double best, const LdsView<D>& L, const int m,
                              const uint16_t* __restrict__ perm, const KdView<I>& kd, int& slot) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(flag && best < __builtin_inf());
    while (todo) {
        const int src = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = __shfl(q[d], src);
        const double bb = __shfl(best, src);
        const double s = __builtin_sqrt(bb);
        unsigned long long mk = ~0ull;
        int ms = -1;
        for (int j = lane; j < m; j += 64) {
            double df = L.t[0][j] - qq[0];
            double acc = df * df;
#pragma unroll
            for (int d = 1; d < D; d++) {
                df = L.t[d][j] - qq[d];
                acc = acc + df * df;
            }
            if (kd_near(acc, bb) && __builtin_sqrt(acc) == s) {
                const int t = perm ? static_cast<int>(perm[j]) : L.j0 + j;
                const unsigned long long k = kd_key<D, I>(kd, qq, t);
                if (k < mk) {
                    mk = k;
                    ms = j;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long ok = __shfl_xor(mk, o);
            const int os = __shfl_xor(ms, o);
            const bool take = ok < mk;
            mk = take ? ok : mk;
            ms = take ? os : ms;
        }
        if (ms < 0) continue;  // (uniform; the minimum itself is in the tied set)
        double w[D];
#pragma unroll
        for (int d = 0; d < D; d++) w[d] = L.t[d][ms];
        int bi = 0x7fffffff, bs = ms;
        for (int j = lane; j < m; j += 64) {
            bool same = true;
#pragma unroll
            for (int d = 0; d < D; d++) same = same && L.t[d][j] == w[d];
            if (same) {
                const int t = perm ? static_cast<int>(perm[j]) : L.j0 + j;
                if (t < bi) {
                    bi = t;
                    bs = j;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int oi = __shfl_xor(bi, o), os = __shfl_xor(bs, o);
            const bool take = oi < bi;
            bi = take ? oi : bi;
            bs = take ? os : bs;
        }
        if (lane == src) slot = bs;
    }
}

}  // namespace icpx
