// icpx_pair_pruned.h — pair_body of the PRUNE kernels (batches: icpx_converge_batch): the same converge loop
// as pair_body (icpx_pair.h), bit for bit, laid out for few registers and little code:
//   * per-slot data (running cloud / source points, neighbour slots, costs) sit in register VECTORS that the
//     slot loops address with a wave-uniform runtime index, so the search exists once per kernel;
//   * nothing is kept that can be recomputed with the same bits (transform * a, matched target coordinates
//     re-read from LDS, squared distances of 3D matches);
//   * the pair's PairState lives in LDS for the visit and goes back to global memory once, at the end.
// Reference lines: as in icpx_pair.h.
#pragma once

#include "icpx_pair.h"

namespace icpx {

template <int D, int METHOD, int QPT>
ICPX_DEV int pair_body_pruned(const PairBatch& pb, const int p_in, const int do_setup_in, const uint64_t max_steps_in,
                              double* lds_raw) {
    // the scheduler hands these over through LDS: tell the compiler they are wave-uniform (scalar registers)
    const int p = uniform_int(p_in), do_setup = uniform_int(do_setup_in);
    const uint64_t max_steps = (static_cast<uint64_t>(static_cast<uint32_t>(uniform_int(static_cast<int>(max_steps_in >> 32)))) << 32) |
                               static_cast<uint32_t>(uniform_int(static_cast<int>(max_steps_in & 0xffffffffu)));
    static_assert(!(D == 2 && METHOD == ICPX_VANILLA), "2D vanilla searches once per begin: no pruned kernel");
    constexpr int L1 = D + 1;
    constexpr int XN = L1 * L1;
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int64_t s0 = pb.src_off[p], d0 = pb.dst_off[p];
    const int n = static_cast<int>(pb.src_off[p + 1] - s0);
    const int m = static_cast<int>(pb.dst_off[p + 1] - d0);
    const double* __restrict__ src = pb.src + s0 * D;
    const double* __restrict__ dst = pb.dst + d0 * D;
    PairState* st = pb.states + p;
    LdsView<D> L = carve<D>(lds_raw, pb.lds_target_cap, pb.lds_sort_cap, false);
    PairState* ls = reinterpret_cast<PairState*>(L.bc + 48);  // this visit's copy of the state (tid 0 works on it)
    const bool empty = (n == 0 || m == 0);
    char* const aux = reinterpret_cast<char*>(lds_raw) + pair_lds_filter_offset(D, pb.lds_target_cap, pb.lds_sort_cap);
    uint16_t* const table = reinterpret_cast<uint16_t*>(aux);                      // the pair's cell start table
    char* const stage_base = aux + sizeof(uint16_t) * kGridTable;                  // per-wave staging of the search
    const CoopStage<D> CS = coop_carve<D>(stage_base + static_cast<size_t>(uniform_int(tid >> 6)) * coop_wave_bytes(D));
    const bool coop = pb.coop != 0;
    const int m4 = (m + 3) & ~3;
    GridView P;
    P.cstart = table;
    P.tperm = pb.tperm + d0;
    P.m = m;
    if (do_setup && pb.prepare) {
        // first visit: the pre-pass of this target cloud, here (no separate launch for the batch to wait for); the
        // cloud lands in LDS in cell order on the way.  Later visits, by any workgroup, read the results back
        if (m > 0) {
            P.g = prepare_target_grid<D>(dst, m, pb.tperm + d0, pb.cstart + static_cast<int64_t>(p) * kGridTable, table,
                                         pb.hcert ? pb.hcert + d0 : nullptr, L.gs, reinterpret_cast<int*>(stage_base), L.t[0],
                                         pb.lds_target_cap, m4);
        } else {
            PairGrid g{};
            g.G = 1;
            P.g = g;
            for (int c = tid; c < kGridTable; c += T) {
                table[c] = 0;
                pb.cstart[static_cast<int64_t>(p) * kGridTable + c] = 0;
            }
        }
        if (tid == 0) pb.grids[p] = P.g;
        P.g.G = uniform_int(P.g.G);
        P.g.bad = uniform_int(P.g.bad);
#pragma unroll
        for (int d = 0; d < D; d++) {
            P.g.lo[d] = uniform_double(P.g.lo[d]);
            P.g.scale[d] = uniform_double(P.g.scale[d]);
        }
        if (do_setup && pb.defer_sparse && (P.g.bad & 2) != 0) {
            // this cloud's grid is a poor index: the brute-force kernel launched behind runs the pair, from the beginning
            if (tid == 0) {
                st->kd_flags = 4;
                atomicAdd(pb.kd_paused, 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            return 1;
        }
    } else {
        P.g = pb.grids[p];
        if (do_setup && pb.defer_sparse && (P.g.bad & 2) != 0) {  // (grids from the separate pre-pass launch)
            if (tid == 0) {
                st->kd_flags = 4;
                atomicAdd(pb.kd_paused, 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            return 1;
        }
        int ncell = P.g.G;
        for (int d = 1; d < D; d++) ncell *= P.g.G;
        for (int c = tid; c <= ncell; c += T) table[c] = pb.cstart[static_cast<int64_t>(p) * kGridTable + c];
        // ---- target cloud into LDS (SoA, in cell order) ----
        (void)stage_targets<D>(dst, m, m4, T, L, false, P.tperm);
    }

    // ---- per-thread point slots ----
    SlotVec<QPT> av[D];  // 2D: source points a; 3D: running cloud c
    SlotVec<QPT> costv;  // 2D: match cost (d2 of the last search)
    SlotVecI<QPT> pv;    // LDS slot of the last neighbour
    int vmask = 0;       // bit k: point tid + k*T exists
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int i = tid + k * T;
        const bool v = i < n;
        vmask |= v ? (1 << k) : 0;
        costv.d[k] = 0.0;
        pv.d[k] = 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            double x = 0.0;
            if (v) {
                if (D == 3 && !do_setup) x = pb.cur[(s0 + i) * D + d];
                else x = src[static_cast<int64_t>(i) * D + d];
            }
            av[d].d[k] = x;
        }
    }

    // ---- state ----
    if (tid == 0) {
        if (do_setup) {
            // ICP::begin, icp.h:120-133
            for (int e = 0; e < 16; e++) {
                double g = 0.0;
                if (e < XN) g = pb.guesses ? pb.guesses[static_cast<int64_t>(p) * XN + e] : ((e % (L1 + 1)) == 0 ? 1.0 : 0.0);
                ls->transform[e] = g;
                ls->last_transform[e] = g;
            }
            ls->iteration_count = 0;
            ls->have_last = 0;
            ls->done = 0;
            ls->cost = 0.0;
            ls->last_cost = 0.0;
            ls->start_clock = wall_clock64();
            ls->kd_flags = 0;
        } else {
            *ls = *st;
        }
        for (int e = 0; e < XN; e++) L.bc[e] = ls->transform[e];
        reinterpret_cast<int*>(L.bc + 41)[0] = static_cast<int>(ls->iteration_count > 0);
    }
    __syncthreads();
    // seeds: the previous nearest neighbours (slots).  Trimmed ICP2 has them from setup on; ICP3 after its
    // first iterate; a resumed visit reloads them.
    bool have_seed = false;
    if (!do_setup && pb.seeds && !empty && (D == 2 || reinterpret_cast<int*>(L.bc + 41)[0] != 0)) {
#pragma unroll
        for (int k = 0; k < QPT; k++) pv.d[k] = ((vmask >> k) & 1) ? pb.seeds[s0 + tid + k * T] : 0;
        have_seed = true;
    }
    bool sorted_matches = false;

    // current transform in registers (uniform)
    Xf<D> X;
#pragma unroll
    for (int e = 0; e < XN; e++) X.m[e] = uniform_double(L.bc[e]);

    // The correspondence step: for every slot k (runtime loop, one copy of the code) the query is
    // c[k] (3D, vanilla_3d.cpp:30-45) or X * a[k] (2D, trimmed.cpp:41,95-109); results into pv / costv.
    // selm: 3D trimmed — bit k set when the neighbour is within max_distance (trimmed_3d.cpp:98).
    constexpr bool TIES = D == 3;  // the kd-tree methods: detect the tie rule's cases (icpx_kd.h); 2D here is trimmed (brute-force rule)
    int kd_tie = 0;
    auto search = [&](const bool seeded, int& selm) __attribute__((always_inline)) {
        const XfRows<D> xr = xf_rows<D>(X);
        selm = 0;
        // seed certificate (icpx_prune.h): the radius of every slot's seed, fetched ahead of the slot loop
        SlotVecI<QPT> hv;  // float bit patterns
#pragma unroll
        for (int k = 0; k < QPT; k++)
            hv.d[k] = (seeded && pb.hcert && m > 0 && ((vmask >> k) & 1))
                          ? __float_as_int(pb.hcert[d0 + min(max(pv.d[k], 0), m - 1)]) : 0;
#pragma unroll 1
        for (int k = 0; k < QPT; k++) {  // slot k sits at index 0: the arrays rotate by one slot per trip
            double a1[D], q1[1][D];
#pragma unroll
            for (int d = 0; d < D; d++) a1[d] = av[d].d[0];
            if (D == 2) xf_apply<D>(xr, a1, q1[0]);
            else {
#pragma unroll
                for (int d = 0; d < D; d++) q1[0][d] = a1[d];
            }
            const bool v = ((vmask >> k) & 1) != 0;
            double b1[1] = {0.0};
            int s1[1] = {pv.d[0]};
            if (seeded) seed_bound<D, 1>(q1, L, m, b1, s1);
            // certified: twice the distance to the seed is below the seed's distance to every other target, so no
            // other target can be as close (seed_certified); the seed is the answer
            const bool todo = v && !(seeded && seed_certified(b1[0], static_cast<double>(__int_as_float(hv.d[0]))));
            if ((threadIdx.x >> 6) == 0) ICPX_STAT(25, __popcll(__ballot(v && !todo)));
            ICPX_STAT(29, __popcll(__ballot(v)));           // every wave: queries answered ...
            ICPX_STAT(30, __popcll(__ballot(v && !todo)));  // ... of which by a certified seed (no search)
            ICPX_STAT(31, __popcll(__ballot(v && seeded))); // ... seed evaluations (one distance each)
            if (__any(todo)) {
                if (coop) nn_coop_slot<D, TIES>(q1[0], todo, seeded, L, P, CS, b1[0], s1[0], &kd_tie);
                else nn_grid_slot<D, TIES>(q1[0], todo, seeded, L, P, b1[0], s1[0], &kd_tie);
            }
            pv.d[0] = v ? s1[0] : 0;
            if (D == 2) {
                costv.d[0] = v ? b1[0] : 0.0;  // trimmed.cpp:104
            } else if (METHOD == ICPX_TRIMMED || pb.m_cost) {
                const double dist = __builtin_sqrt(__builtin_sqrt(b1[0]));  // vanilla_3d.cpp:41 on kdtree.h:126
                if (METHOD == ICPX_TRIMMED && v && dist <= pb.max_distance) selm |= 1 << k;
                if (pb.m_cost && v) pb.m_cost[s0 + tid + k * T] = dist;  // Neighbors of the last iterate (handles)
            }
#pragma unroll
            for (int d = 0; d < D; d++) av[d].rotate();
            pv.rotate();
            hv.rotate();
            if (D == 2) costv.rotate();
        }
    };
    // cost sum of ICP::calculate_cost, icp.h:149-155 (the 3D methods never fill `matches`: zero)
    auto cost_of = [&](int k, int) __attribute__((always_inline)) -> double { return (D == 2 && ((vmask >> k) & 1)) ? costv.d[k] : 0.0; };

    double state_cost = 0.0;
    if (do_setup) {
        if (D == 2 && !empty) {
            int dummy;
#ifdef ICPX_PROBE
            if (!(pb.coop & 256))
#endif
            search(false, dummy);
            have_seed = true;
        }
        double tot[1];
        block_sum<1, QPT>(cost_of, n, T, L.gs, tot);
        state_cost = __builtin_sqrt(tot[0] / static_cast<double>(n));
        if (tid == 0) ls->cost = state_cost;
    }

    // ---- ICPDriver::converge loop, driver.h:63-69 ----
    uint64_t steps = 0;
    int exit_code = 0;
    while (true) {
        if (tid == 0) {
            int stop = 0;
            if (pb.check) {
                double el = pb.host_elapsed_s + static_cast<double>(wall_clock64() - ls->start_clock) * pb.clock_to_s;
                if (should_terminate<D>(pb.crit, *ls, el)) {
                    stop = 1;
                    ls->done = 1;
                }
            }
            if (!stop && steps >= max_steps) stop = 2;
            if (!stop) {
                // last_state = state
                for (int e = 0; e < XN; e++) ls->last_transform[e] = ls->transform[e];
                ls->last_cost = ls->cost;
                ls->have_last = 1;
            }
            reinterpret_cast<int*>(L.bc + 40)[0] = stop;
            reinterpret_cast<int*>(L.bc + 42)[0] = 0;  // block_any's word
        }
        __syncthreads();
        const int stop = reinterpret_cast<int*>(L.bc + 40)[0];
        __syncthreads();
        exit_code = stop;
        if (stop) break;
        steps++;

        if (!empty) {
            // ================= iterate() =================
#ifdef ICPX_STATS
            uint64_t ph0 = wall_clock64();
#endif
            int selm = 0, k_sel = n;
#ifdef ICPX_PROBE
            if (!(pb.coop & 16) || !have_seed)
#endif
            search(have_seed, selm);
            have_seed = true;
            if (TIES) {
                if (block_any(kd_tie != 0, reinterpret_cast<int*>(L.bc + 42))) {
                    // a query met the kd rule's tie case, which this kernel only detects: the pair stops here, as it stood
                    // before this iteration (nothing of it has been stored), and the KD kernel behind takes it over
                    if (tid == 0) {
                        ls->kd_flags |= 2;
                        atomicAdd(pb.kd_paused, 1);
                    }
                    exit_code = 1;
                    break;
                }
            }
            const XfRows<D> xr = xf_rows<D>(X);
            if (D == 2) {
                // trimming step, trimmed.cpp:55-58: in the order (cost, point) keep the first new_n; only the
                // new_n-th pair is needed to mark them (radix_select)
                sorted_matches = true;
                long long new_n = static_cast<long long>(pb.overlap_rate * static_cast<double>(n));
                if (new_n < 1) new_n = 1;
                if (new_n > n) new_n = n;
                k_sel = static_cast<int>(new_n);
                uint64_t skeys[QPT];
                bool valid[QPT];
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    skeys[k] = static_cast<uint64_t>(__double_as_longlong(costv.d[k]));
                    valid[k] = ((vmask >> k) & 1) != 0;
                }
                uint64_t kth;
                uint32_t vth;
                ICPX_PHASE(8);   // correspondence search
#ifdef ICPX_PROBE
                if (pb.coop & 128) {
                    kth = ~0ull;
                    vth = ~0u;
                } else
#endif
                radix_select<QPT>(skeys, valid, T, k_sel, reinterpret_cast<int*>(L.gs), kth, vth);
                ICPX_PHASE(9);   // selection
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    const uint32_t vv = static_cast<uint32_t>(tid + k * T);
                    if (valid[k] && (skeys[k] < kth || (skeys[k] == kth && vv <= vth))) selm |= 1 << k;
                }
            } else {
                if (METHOD == ICPX_VANILLA) selm = vmask;
                else {
                    int flags[QPT];
#pragma unroll
                    for (int k = 0; k < QPT; k++) flags[k] = (selm >> k) & 1;
                    k_sel = block_count<QPT>(flags, T, L.gs);
                }
                ICPX_PHASE(8);
            }
            // the query of slot k as the search saw it, and its matched target (both recomputed: same bits)
            auto cur_of = [&](int k, int d) __attribute__((always_inline)) -> double {
                if (D == 3) return av[d].d[k];
                double a1[D], q1[D];
#pragma unroll
                for (int e = 0; e < D; e++) a1[e] = av[e].d[k];
                xf_apply<D>(xr, a1, q1);
                return q1[d];
            };
            auto bp_of = [&](int k, int d) __attribute__((always_inline)) -> double { return L.t[d][pv.d[k]]; };

            // pass 1: centroids (+ cost sum for 2D)
            constexpr int NV1 = D == 2 ? 2 * D + 1 : 2 * D;
            double s1[NV1];
#ifdef ICPX_PROBE
            if (pb.coop & 32) {
                for (int v = 0; v < NV1; v++) s1[v] = 1.0 + v;
            } else
#endif
            block_sum<NV1, QPT>(
                [&](int k, int v) __attribute__((always_inline)) -> double {
                    const bool s = ((selm >> k) & 1) != 0;
                    if (v < D) return s ? cur_of(k, v) : 0.0;
                    if (v < 2 * D) return s ? bp_of(k, v - D) : 0.0;
                    return cost_of(k, 0);
                },
                n, T, L.gs, s1);
            double ca[D], cb[D];
            const double kd = static_cast<double>(k_sel);
#pragma unroll
            for (int d = 0; d < D; d++) {
                ca[d] = uniform_double(s1[d] / kd);
                cb[d] = uniform_double(s1[D + d] / kd);
            }
            // pass 2: cross-covariance N = sum (a - ca)(b - cb)^T
            double s2[D * D];
#ifdef ICPX_PROBE
            if (pb.coop & 32) {
                for (int v = 0; v < D * D; v++) s2[v] = (v % (D + 1)) == 0 ? 1.0 : 0.01 * v;
            } else
#endif
            block_sum<D * D, QPT>(
                [&](int k, int v) __attribute__((always_inline)) -> double {
                    const int r = v % D, c = v / D;
                    return ((selm >> k) & 1) ? (cur_of(k, r) - ca[r]) * (bp_of(k, c) - cb[c]) : 0.0;
                },
                n, T, L.gs, s2);
            ICPX_PHASE(10);  // sums
            if (tid == 0) {
                Mat<D> N;
#pragma unroll
                for (int e = 0; e < D * D; e++) N.a[e] = s2[e];
#ifdef ICPX_PROBE
                Xf<D> step;
                if (pb.coop & 64) step.set_identity();
                else step = kabsch<D>(N, ca, cb);
#else
                Xf<D> step = kabsch<D>(N, ca, cb);
#endif
                Xf<D> nx = xf_mul<D>(step, X);
                for (int e = 0; e < XN; e++) {
                    L.bc[e] = nx.m[e];
                    L.bc[16 + e] = step.m[e];
                    ls->transform[e] = nx.m[e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < XN; e++) X.m[e] = uniform_double(L.bc[e]);
            if (D == 3) {
                // c = T * c, vanilla_3d.cpp:91
                Xf<D> S;
#pragma unroll
                for (int e = 0; e < XN; e++) S.m[e] = uniform_double(L.bc[16 + e]);
                const XfRows<D> sr = xf_rows<D>(S);
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    double c1[D], c2[D];
#pragma unroll
                    for (int d = 0; d < D; d++) c1[d] = av[d].d[k];
                    xf_apply<D>(sr, c1, c2);
#pragma unroll
                    for (int d = 0; d < D; d++) av[d].d[k] = c2[d];
                }
            }
            if (D == 2) state_cost = __builtin_sqrt(s1[NV1 - 1] / static_cast<double>(n));
            __syncthreads();
            ICPX_PHASE(11);  // Kabsch + broadcast
        }
        if (tid == 0) {
            ls->iteration_count += 1;
            ls->cost = state_cost;
        }
    }

    // ---- publish ----
    // (a converged pair of a batch is never visited again: its running cloud and seeds stay where they are)
    const bool revisit = !(ls->done != 0 && pb.m_point == nullptr);
    if (pb.seeds && revisit) {
#pragma unroll
        for (int k = 0; k < QPT; k++)
            if ((vmask >> k) & 1) pb.seeds[s0 + tid + k * T] = pv.d[k];
    }
    if (D == 3 && pb.cur && revisit) {
#pragma unroll
        for (int k = 0; k < QPT; k++)
            if ((vmask >> k) & 1)
#pragma unroll
                for (int d = 0; d < D; d++) pb.cur[(s0 + tid + k * T) * D + d] = av[d].d[k];
    }
    if (pb.m_point) {
        if (D == 3) {
            // Neighbors of the last iterate (debug view; distances were stored by the search); ICP3 matches stay zero (Q4)
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if ((vmask >> k) & 1) {
                    pb.m_pair[s0 + tid + k * T] = m > 0 ? static_cast<int>(P.tperm[pv.d[k]]) : 0;
                    pb.m_point[s0 + tid + k * T] = tid + k * T;
                    if (steps == 0) pb.m_cost[s0 + tid + k * T] = 0.0;  // no iterate in this visit (begin)
                }
        } else if (!sorted_matches) {
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if ((vmask >> k) & 1) {
                    pb.m_point[s0 + tid + k * T] = tid + k * T;
                    pb.m_pair[s0 + tid + k * T] = static_cast<int>(P.tperm[pv.d[k]]);
                    pb.m_cost[s0 + tid + k * T] = costv.d[k];
                }
        } else {
            // Trimmed leaves matches sorted by cost (trimmed.cpp:55-56): sort (cost, point) now
            const int npad = pb.lds_sort_cap;
            for (int i = tid; i < npad; i += T) {
                L.skey[i] = ~0ULL;
                L.sval[i] = static_cast<uint32_t>(i);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if ((vmask >> k) & 1) {
                    L.skey[tid + k * T] = static_cast<uint64_t>(__double_as_longlong(costv.d[k]));
                    pb.pair_tmp[s0 + tid + k * T] = static_cast<int>(P.tperm[pv.d[k]]);
                }
            __syncthreads();
            bitonic_sort(L.skey, L.sval, npad, T);
            for (int r = tid; r < n; r += T) {
                uint32_t pt = L.sval[r];
                pb.m_point[s0 + r] = pt;
                pb.m_pair[s0 + r] = pb.pair_tmp[s0 + pt];
                pb.m_cost[s0 + r] = __longlong_as_double(static_cast<long long>(L.skey[r]));
            }
        }
    }
    if (tid == 0) {
        *st = *ls;
        if (pb.out_states) {
            icpx_state* o = pb.out_states + p;
            o->cost = ls->cost;
            o->iteration_count = ls->iteration_count;
            for (int e = 0; e < 16; e++) o->transform[e] = e < XN ? ls->transform[e] : 0.0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's stores are out before the hand-off
    __syncthreads();                                   // and LDS may be restaged by the next task
    return exit_code;
}

}  // namespace icpx
