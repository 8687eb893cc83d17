// icpx_pair_inst.hip — the fused pair kernels (icpx_pair.h), instantiated per (dimension, pruned or
// not) in a translation unit of their own: this file is compiled four times (Makefile) with
// -DICPX_INST_DIM=2|3 -DICPX_INST_PRUNE=0|1, so the many (method, geometry) variants build in
// parallel.  Each object exports one launcher, launch_pairs_<dim>d_<plain|prune>.
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "icpx_host.h"
#include "icpx_pair.h"
#include "icpx_pair_pruned.h"

#if !defined(ICPX_INST_DIM) || !defined(ICPX_INST_PRUNE)
#error "compile with -DICPX_INST_DIM=2|3 -DICPX_INST_PRUNE=0|1"
#endif

namespace icpx {
namespace {

template <int D, int METHOD, int QPT, int TMAX, bool PRUNE, bool KD = (D == 2 && METHOD == ICPX_VANILLA)>
int launch_one(const PairBatch& pb_in, int64_t n_pairs, const PairLaunch& g, hipStream_t stream) {
    auto kern = pair_kernel<D, METHOD, QPT, TMAX, PRUNE, KD>;
    static std::mutex mu;
    static std::unordered_map<int, bool> attr_set;  // per device
    int dev = 0;
    ICPX_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(kPairMaxLds)));
            attr_set[dev] = true;
        }
    }
    PairBatch pb = pb_in;
    int64_t blocks = n_pairs;
    if (pb.sched_state) {
        // iteration-granular scheduling pays when the pairs outnumber the resident workgroups only
        // by a small factor (the tail of a pair-per-workgroup launch is then a large share)
        int per_cu = 0;
        ICPX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, g.threads, g.lds));
        const int64_t resident = static_cast<int64_t>(per_cu > 0 ? per_cu : 1) * g.cus;
        const int force = knobs().sched;  // development knob ICPX_SCHED: 0 = never, 1 = always
        bool use = n_pairs > resident && n_pairs <= 16 * resident && n_pairs < (1 << 20) &&
                   !(D == 2 && METHOD == ICPX_VANILLA) && !pb.kd_takeover;
        if (force >= 0) use = force == 1 && n_pairs < (1 << 20) && !(D == 2 && METHOD == ICPX_VANILLA) && !pb.kd_takeover;
        if (use) {
            blocks = n_pairs < resident ? n_pairs : resident;
            pb.n_pairs = static_cast<int32_t>(n_pairs);
            // iterations per task: 16 = most pairs are set up and converged in one visit.  Every visit ends with the
            // pair's running cloud and seeds written back and an agent-scope release; with one iteration per task that
            // was 2/3 of the launch's HBM traffic (710 -> 225 MB per 512-pair 3D launch, PMC) and, for batches of many times
            // the resident workgroups, 20 % of the time (8192 pairs of config 5: 11.0 -> 8.8 ms)
            pb.sched_steps = 16;
            if (const int v = knobs().sched_steps) pb.sched_steps = v;  // development knob ICPX_SCHED_STEPS
            // (whole 16-byte units: one fill kernel instead of two; the callers reserve twice the words)
            ICPX_HIP(hipMemsetAsync(pb.sched_state, 0, (sizeof(int32_t) * (n_pairs + 1) + 15) & ~size_t(15), stream));
            pb.sched_done = pb.sched_state + n_pairs;
        } else {
            pb.sched_state = nullptr;
        }
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(g.threads), g.lds, stream, pb);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

template <int D, int METHOD, bool PRUNE>
int launch_dm(const PairBatch& pb, int64_t n_pairs, const PairLaunch& g, hipStream_t stream) {
    const int key = g.qpt * 10000 + g.threads;
    switch (key) {
        case 10256: return launch_one<D, METHOD, 1, 256, PRUNE>(pb, n_pairs, g, stream);
        case 20256: return launch_one<D, METHOD, 2, 256, PRUNE>(pb, n_pairs, g, stream);
        case 40256: return launch_one<D, METHOD, 4, 256, PRUNE>(pb, n_pairs, g, stream);
        case 80256: return launch_one<D, METHOD, 8, 256, PRUNE>(pb, n_pairs, g, stream);
        case 20512: return launch_one<D, METHOD, 2, 512, PRUNE>(pb, n_pairs, g, stream);
        case 40512: return launch_one<D, METHOD, 4, 512, PRUNE>(pb, n_pairs, g, stream);
        case 80512: return launch_one<D, METHOD, 8, 512, PRUNE>(pb, n_pairs, g, stream);
        default: break;
    }
    return fail(ICPX_ERR_INVALID, "no fused kernel for %d queries per lane x %d threads", g.qpt, g.threads);
}

}  // namespace

#define ICPX_CAT3(a, b, c) a##b##c
#define ICPX_INST_NAME2(d, p) ICPX_CAT3(launch_pairs_d, d, p)
#if ICPX_INST_PRUNE
#define ICPX_INST_NAME(d) ICPX_INST_NAME2(d, _prune)
#else
#define ICPX_INST_NAME(d) ICPX_INST_NAME2(d, _plain)
#endif

int ICPX_INST_NAME(ICPX_INST_DIM)(const PairBatch& pb, int64_t n_pairs, const PairLaunch& g, hipStream_t stream) {
    constexpr int D = ICPX_INST_DIM;
    constexpr bool PRUNE = ICPX_INST_PRUNE != 0;
    if (pb.kd_takeover) {
        // the kernel behind a launch that detects the kd rule's ties (3D: two geometries, n <= 1024 and n <= 4096) and / or hands
        // over the pairs its grid does not suit (2D trimmed: the brute-force filter kernel of the run's geometry)
        if constexpr (D == 3 && !PRUNE) {
            if (g.qpt == 4 && g.threads == 256) {  // (sources of up to 1024 points: the batch kernels' own geometry)
                if (pb.method == ICPX_VANILLA) return launch_one<3, ICPX_VANILLA, 4, 256, false, true>(pb, n_pairs, g, stream);
                return launch_one<3, ICPX_TRIMMED, 4, 256, false, true>(pb, n_pairs, g, stream);
            }
            if (g.qpt != 8 || g.threads != 512) return fail(ICPX_ERR_INVALID, "the kd take-over kernel runs 4 x 256 or 8 x 512 (queries per lane x threads)");
            if (pb.method == ICPX_VANILLA) return launch_one<3, ICPX_VANILLA, 8, 512, false, true>(pb, n_pairs, g, stream);
            return launch_one<3, ICPX_TRIMMED, 8, 512, false, true>(pb, n_pairs, g, stream);
        } else if constexpr (D == 2 && !PRUNE) {
            if (pb.method != ICPX_TRIMMED) return fail(ICPX_ERR_INVALID, "no take-over kernel for this method");
            return launch_dm<2, ICPX_TRIMMED, false>(pb, n_pairs, g, stream);
        } else {
            return fail(ICPX_ERR_INVALID, "no take-over kernel in this object");
        }
    }
    if (pb.method == ICPX_VANILLA) {
        // 2D vanilla searches once per begin: it has no filtered (hence no pruned) search
        if constexpr (D == 2 && PRUNE) return fail(ICPX_ERR_INVALID, "no pruned kernel for 2D vanilla");
        else return launch_dm<D, ICPX_VANILLA, PRUNE>(pb, n_pairs, g, stream);
    }
    return launch_dm<D, ICPX_TRIMMED, PRUNE>(pb, n_pairs, g, stream);
}

#ifdef ICPX_STATS
#define ICPX_STATS_NAME2(d, p) ICPX_CAT3(stats_read_d, d, p)
#if ICPX_INST_PRUNE
#define ICPX_STATS_NAME(d) ICPX_STATS_NAME2(d, _prune)
#else
#define ICPX_STATS_NAME(d) ICPX_STATS_NAME2(d, _plain)
#endif
// adds this object's counters to out[32]; optionally clears them
int ICPX_STATS_NAME(ICPX_INST_DIM)(unsigned long long* out, int reset) {
    unsigned long long v[32];
    ICPX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(icpx::g_icpx_stats), sizeof(v)));
    for (int i = 0; i < 32; i++) out[i] += v[i];
    if (reset) {
        unsigned long long z[32] = {};
        ICPX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(icpx::g_icpx_stats), z, sizeof(z)));
    }
    return ICPX_OK;
}
#endif

}  // namespace icpx
