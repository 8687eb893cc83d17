// icpx_capi.hip — implementation of include/icpx/icpx.h.
//
// Host side of the drop-in boundary: argument checks, device buffers, path selection and kernel
// launches.  There is no CPU compute path in here: when no gfx950 device is present every entry
// point fails with ICPX_ERR_NO_DEVICE.
//
// Paths
//   fused   n <= 4096 and the target cloud fits LDS: one workgroup per scan pair runs the whole
//           converge loop (icpx_pair.h).  Also what icpx_converge_batch launches, one workgroup
//           per pair.
//   large   everything else: exact uniform-grid (or tiled brute-force) NN + multi-workgroup
//           TREE64 reductions + a one-lane Kabsch/termination kernel per iteration, chained on
//           the stream with a device-side done flag (icpx_large.h).

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <algorithm>
#include <thread>
#include <unordered_map>
#include <vector>

#include "icpx_host.h"
#include "icpx_large.h"
#include "icpx_nn.h"
#include "icpx_pair.h"

struct icpx_ctx;

namespace icpx {

// the fused kernels live in four objects built from icpx_pair_inst.hip
int launch_pairs_d2_plain(const PairBatch&, int64_t, const PairLaunch&, hipStream_t);
int launch_pairs_d2_prune(const PairBatch&, int64_t, const PairLaunch&, hipStream_t);
int launch_pairs_d3_plain(const PairBatch&, int64_t, const PairLaunch&, hipStream_t);
int launch_pairs_d3_prune(const PairBatch&, int64_t, const PairLaunch&, hipStream_t);
#ifdef ICPX_STATS
int stats_read_d2_plain(unsigned long long*, int);
int stats_read_d2_prune(unsigned long long*, int);
int stats_read_d3_plain(unsigned long long*, int);
int stats_read_d3_prune(unsigned long long*, int);
#endif

std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}

namespace {
// "ICPX_X" -> knob value; value == nullptr restores the built-in default
bool set_knob(Knobs& k, const char* name, const char* value) {
    const std::string n = name ? name : "";
    const int iv = value ? std::atoi(value) : 0;
    if (n == "ICPX_PRUNE") k.prune = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_FILTER") k.filter = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_SCHED") k.sched = value ? (value[0] == '1' ? 1 : 0) : -1;
    else if (n == "ICPX_SCHED_STEPS") k.sched_steps = (value && iv >= 1 && iv <= 64) ? iv : 0;
    else if (n == "ICPX_PRUNE_HANDLES") k.prune_handles = value ? (value[0] == '1') : 0;
    else if (n == "ICPX_NO_GROUP_SUMS") k.no_group_sums = value ? 1 : 0;
    else if (n == "ICPX_COOP") k.coop = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_GRID_COOP") k.grid_coop = value ? (value[0] == '2' ? 2 : (value[0] != '0')) : 1;
    else if (n == "ICPX_POLL_DONE") k.poll_done = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_GRID_SEED") k.grid_seed = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_DEFER_SPARSE") k.defer_sparse = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_CHUNK_SKIP") k.chunk_skip = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_ZERO_COPY") k.zero_copy = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_GRID_DENSITY") k.grid_density = (value && iv >= 1 && iv <= 1000) ? iv : 0;
    else if (n == "ICPX_GRID_SD") k.grid_sd = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_GRID_FILTER") k.grid_filter = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_KD_EAGER") k.kd_eager = value ? (value[0] == '1' ? 1 : (value[0] == '0' ? 0 : -1)) : -1;
    else if (n == "ICPX_TILE_CHUNK") k.tile_chunk = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_TILE_MIN") k.tile_min = (value && iv >= 64 && iv <= 3072) ? (iv & ~3) : 0;
    else if (n == "ICPX_TILE_WGS") k.tile_wgs = (value && iv >= 64 && iv <= 65536) ? iv : 0;
    else if (n == "ICPX_FUSE_PREPASS") k.fuse_prepass = value ? (value[0] != '0') : 1;
    else if (n == "ICPX_FUSED_SINGLE_MAX") k.fused_single_max = value ? std::atof(value) : -1.0;
    else if (n == "ICPX_GEOM") {
        int q = 0, t = 0;
        if (value && std::sscanf(value, "%d,%d", &q, &t) == 2) {
            k.geom_qpt = q;
            k.geom_threads = t;
        } else {
            k.geom_qpt = 0;
            k.geom_threads = 0;
        }
    } else {
        return false;
    }
    return true;
}
}  // namespace

Knobs& knobs() {
    static Knobs k;
    static const bool from_env = [] {
        for (const char* n : {"ICPX_PRUNE", "ICPX_FILTER", "ICPX_SCHED", "ICPX_SCHED_STEPS", "ICPX_PRUNE_HANDLES",
                              "ICPX_NO_GROUP_SUMS", "ICPX_COOP", "ICPX_GRID_COOP", "ICPX_GRID_SEED", "ICPX_GRID_SD", "ICPX_GRID_FILTER", "ICPX_TILE_CHUNK", "ICPX_KD_EAGER", "ICPX_DEFER_SPARSE", "ICPX_CHUNK_SKIP", "ICPX_ZERO_COPY", "ICPX_GRID_DENSITY", "ICPX_TILE_WGS", "ICPX_TILE_MIN", "ICPX_POLL_DONE", "ICPX_FUSE_PREPASS", "ICPX_FUSED_SINGLE_MAX", "ICPX_GEOM"})
            if (const char* v = std::getenv(n)) set_knob(k, n, v);
        return true;
    }();
    (void)from_env;
    return k;
}

namespace {

constexpr size_t kMaxLds = kPairMaxLds;

struct DeviceInfo {
    bool ok = false;
    double clock_to_s = 1e-8;
    int cus = 256;
    std::string arch;
};

std::mutex g_mu;
std::unordered_map<int, DeviceInfo> g_devinfo;

int device_info(int device, DeviceInfo** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_devinfo.find(device);
    if (it == g_devinfo.end()) {
        DeviceInfo di;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess)
            return fail(ICPX_ERR_NO_DEVICE, "hipGetDeviceProperties(%d) failed: %s", device,
                        hipGetErrorString(e));
        di.arch = prop.gcnArchName;
        if (di.arch.rfind("gfx950", 0) != 0)
            return fail(ICPX_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                        device, di.arch.c_str());
        int khz = 0;
        e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device);
        if (e != hipSuccess || khz <= 0) khz = 100000;
        di.clock_to_s = 1.0 / (static_cast<double>(khz) * 1e3);
        di.cus = prop.multiProcessorCount;
        di.ok = true;
        it = g_devinfo.emplace(device, di).first;
    }
    *out = &it->second;
    return ICPX_OK;
}

int resolve_device(int requested, int* device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(ICPX_ERR_NO_DEVICE, "no HIP device visible (%s); icpx has no CPU path",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    }
    if (requested < 0) {
        ICPX_HIP(hipGetDevice(device));
    } else {
        if (requested >= count) return fail(ICPX_ERR_INVALID, "device %d out of range (%d visible)", requested, count);
        *device = requested;
    }
    return ICPX_OK;
}

int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

constexpr double kFusedSingleMaxWork = 1.1e6;  // measured crossover (n = m = 1024), see DESIGN.md 4.4

struct PairGeom {
    int qpt, threads, tcap, scap;
    int64_t m;    // largest target cloud the geometry was sized for
    bool kd;      // KD kernel (2D vanilla): the kd rule's ties are resolved in the kernel (icpx_kd.h)
    bool filter;  // f32 target copy reserved in LDS (nn_filtered)
    bool prune;   // grid-pruned search (icpx_prune.h): targets ordered by cell, cell table in LDS
    bool chunk;   // trimmed ICP2, brute-force kernels: chunk-skipping exact search (nn_lds_chunked) instead of the filter
    size_t lds;
};

// does a pair of this size fit the fused kernel?
bool fused_fits(int dim, int method, int64_t n, int64_t m, PairGeom* g, bool want_matches = true,
                bool allow_prune = false) {
    if (n > kPairMaxN || m > (1 << 20)) return false;
    PairGeom r;
    if (n <= 256) { r.qpt = 1; r.threads = 256; }
    else if (n <= 512) { r.qpt = 2; r.threads = 256; }
    else if (n <= 1024) {
        // batches: (4,256) has the higher throughput (several workgroups share a CU); a single pair
        // owns its CU and iterates 1.4x faster with twice the threads (59 vs 85 us at n = m = 1024)
        if (want_matches) { r.qpt = 2; r.threads = 512; } else { r.qpt = 4; r.threads = 256; }
    }
    else if (n <= 2048) { if (dim == 2) { r.qpt = 8; r.threads = 256; } else { r.qpt = 4; r.threads = 512; } }
    else { r.qpt = 8; r.threads = 512; }
    // development knob ICPX_GEOM="qpt,threads" (must cover n)
    {
        const int q = knobs().geom_qpt, t = knobs().geom_threads;
        if (q > 0 && t > 0 && static_cast<int64_t>(q) * t >= n) {
            r.qpt = q;
            r.threads = t;
        }
    }
    r.tcap = static_cast<int>((m + 3) & ~int64_t(3));  // targets are consumed four at a time
    if (r.tcap < 4) r.tcap = 4;
    // sort area: only the sorted matches of trimmed ICP2 (published through a handle) need it
    r.scap = (dim == 2 && method == ICPX_TRIMMED && want_matches) ? next_pow2(static_cast<int>(n)) : 0;
    // the packed-f32 candidate filter needs 4 more bytes of LDS per target coordinate; 2D vanilla
    // searches once per begin and never uses it.  Development knob: ICPX_FILTER=0
    r.filter = !(dim == 2 && method == ICPX_VANILLA) && knobs().filter != 0 &&
               pair_lds_bytes(dim, r.tcap, r.scap, true) <= kMaxLds;
    r.prune = false;
    if (allow_prune && !(dim == 2 && method == ICPX_VANILLA) && m >= 128 && m < 65536) {
        // grid-pruned search (icpx_prune.h); development knob: ICPX_PRUNE=0
        if (knobs().prune != 0 && pair_lds_bytes(dim, r.tcap, r.scap, false, true, r.threads) <= kMaxLds) {
            r.prune = true;
            r.filter = false;  // no f32 copy in LDS
        }
    }
    r.lds = pair_lds_bytes(dim, r.tcap, r.scap, r.filter, r.prune, r.threads);
    // trimmed ICP2 without the grid: the exact scan skips the chunks of 32 consecutive targets whose box no query of
    // the wave can reach (icpx_filter.h) — the reference's lidar scans come in scan order.  Development knob: ICPX_CHUNK_SKIP=0
    r.chunk = false;
    if (dim == 2 && method == ICPX_TRIMMED && !r.prune && knobs().chunk_skip != 0 && m >= 2 * kChunk &&
        pair_lds_bytes_chunk(2, r.tcap, r.scap) <= kMaxLds) {
        r.chunk = true;
        r.filter = false;
        r.lds = pair_lds_bytes_chunk(2, r.tcap, r.scap);
    }
    r.m = m;
    r.kd = dim == 2 && method == ICPX_VANILLA;
    // the kd rule (icpx_kd.h): a pair's tree is built in the kernel's LDS on its first tie — by the 2D vanilla kernels
    // themselves, by the take-over kernel for the 3D methods; both need room for it, and 16-bit positions
    if (dim == 3 || r.kd) {
        if (m >= 65536) return false;
        const size_t kd_lds = pair_lds_bytes_kd(dim, r.tcap, r.kd ? r.scap : 0, m);
        if (kd_lds > kMaxLds) return false;
        if (r.kd) r.lds = kd_lds;
    }
    if (r.lds > kMaxLds) return false;
    if (g) *g = r;
    return true;
}

int launch_pairs(PairBatch& pb, int64_t n_pairs, const PairGeom& g, hipStream_t stream) {
    if (n_pairs <= 0) return ICPX_OK;
    pb.lds_target_cap = g.tcap;
    pb.lds_sort_cap = g.scap;
    pb.use_filter = g.filter ? 1 : 0;
    pb.chunk_skip = g.chunk ? 1 : 0;
    pb.coop = knobs().coop != 0 ? 1 : 0;
#ifdef ICPX_PROBE
    if (const char* pe = std::getenv("ICPX_PROBE_MASK")) pb.coop |= std::atoi(pe);
#endif
    int dev = 0;
    ICPX_HIP(hipGetDevice(&dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    PairLaunch pl;
    pl.qpt = g.qpt;
    pl.threads = g.threads;
    pl.lds = g.lds;
    pl.cus = di->cus;
    const bool prune = g.prune && pb.tperm && pb.seeds && pb.grids && pb.cstart;
    if (g.prune && !prune) return fail(ICPX_ERR_INVALID, "pruned launch without its permutation buffers");
    const bool kd_rule = pb.dim == 3 || pb.method == ICPX_VANILLA;
    if (kd_rule && (!pb.kdnode || !pb.kdpos || !pb.kd_paused)) return fail(ICPX_ERR_INVALID, "launch without the kd rule's buffers");
    // pruned kernels hand the pairs whose grid is a poor index (lidar scans: points on curves) to a brute-force kernel
    // launched behind them (PairGrid::bad bit 1, icpx_pair_pruned.h); development knob ICPX_DEFER_SPARSE=0
    const bool defer = prune && pb.kd_paused != nullptr && knobs().defer_sparse != 0;
    pb.defer_sparse = defer ? 1 : 0;
    if (pb.dim == 2) {
        if (!defer) return prune ? launch_pairs_d2_prune(pb, n_pairs, pl, stream) : launch_pairs_d2_plain(pb, n_pairs, pl, stream);
        PairGeom g2;
        if (!fused_fits(2, pb.method, static_cast<int64_t>(g.qpt) * g.threads, g.m, &g2, g.scap > 0, false)) {
            pb.defer_sparse = 0;  // (no brute-force geometry for these sizes: every pair stays with the pruned kernel)
            return launch_pairs_d2_prune(pb, n_pairs, pl, stream);
        }
        ICPX_HIP(hipMemsetAsync(pb.kd_paused, 0, sizeof(int32_t), stream));
        ICPX_TRY(launch_pairs_d2_prune(pb, n_pairs, pl, stream));
        PairBatch tb = pb;
        tb.kd_takeover = 1;
        tb.defer_sparse = 0;
        tb.sched_state = nullptr;
        tb.lds_target_cap = g2.tcap;
        tb.lds_sort_cap = g2.scap;
        tb.use_filter = g2.filter ? 1 : 0;
        tb.chunk_skip = g2.chunk ? 1 : 0;
        PairLaunch tl;
        tl.qpt = g2.qpt;
        tl.threads = g2.threads;
        tl.lds = g2.lds;
        tl.cus = di->cus;
        return launch_pairs_d2_plain(tb, n_pairs, tl, stream);
    }
    // 3D: the kernels only detect the kd rule's ties; a pair that meets one stops there and the KD kernel enqueued
    // behind takes it over (a launch that finds nothing to do when no pair stopped)
    ICPX_HIP(hipMemsetAsync(pb.kd_paused, 0, sizeof(int32_t), stream));
    ICPX_TRY(prune ? launch_pairs_d3_prune(pb, n_pairs, pl, stream) : launch_pairs_d3_plain(pb, n_pairs, pl, stream));
    PairBatch tb = pb;
    tb.kd_takeover = 1;
    tb.defer_sparse = 0;
    tb.sched_state = nullptr;
    tb.use_filter = 0;
    tb.chunk_skip = 0;
    tb.lds_sort_cap = 0;
    tb.do_setup = 0;
    PairLaunch tl;
    const bool small = static_cast<int64_t>(g.qpt) * g.threads <= 1024;
    tl.qpt = small ? 4 : 8;
    tl.threads = small ? 256 : 512;
    tl.lds = pair_lds_bytes_kd(3, g.tcap, 0, g.m);
    tl.cus = di->cus;
    return launch_pairs_d3_plain(tb, n_pairs, tl, stream);
}

// launch of the pre-pass kernel: its dynamic LDS holds the cloud in cell order for the certificate radii; clouds
// too large for that get no certificates (hcert stays unused: *use_cert = false)
int launch_prepass(int dim, int64_t n_pairs, int64_t m_max, const double* d_dst, const int64_t* d_doff, uint16_t* tperm,
                   PairGrid* grids, uint16_t* cstart, float* hcert, bool* use_cert, hipStream_t stream) {
    static std::mutex mu;
    static std::unordered_map<int, bool> attr_set;  // per device
    int dev = 0;
    ICPX_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_perm_kernel<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPairMaxLds)));
            ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_perm_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPairMaxLds)));
            attr_set[dev] = true;
        }
    }
    size_t lds = pair_perm_lds_bytes(dim, m_max);
    *use_cert = hcert != nullptr && lds <= kPairMaxLds;
    if (!*use_cert) lds = pair_perm_lds_bytes(dim, 0);
    auto pk = dim == 2 ? pair_perm_kernel<2> : pair_perm_kernel<3>;
    hipLaunchKernelGGL(pk, dim3(static_cast<unsigned>(n_pairs)), dim3(m_max >= 2048 ? 512 : 256), lds, stream, d_dst, d_doff,
                       tperm, grids, cstart, *use_cert ? hcert : nullptr);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// pre-pass of the grid-pruned search (icpx_prune.h): grid, cell order and cell table of every target cloud
int enqueue_prepass(int dim, int64_t n_pairs, int64_t m_max, int64_t m_total, const double* d_dst, const int64_t* d_doff,
                    DevBuf& tperm, DevBuf& grids, DevBuf& cstart, DevBuf& hcert, bool* cert_out, hipStream_t stream) {
    ICPX_TRY(tperm.reserve(sizeof(uint16_t) * (m_total > 0 ? m_total : 1)));
    ICPX_TRY(hcert.reserve(sizeof(float) * (m_total > 0 ? m_total : 1)));
    ICPX_TRY(grids.reserve(sizeof(PairGrid) * n_pairs));
    ICPX_TRY(cstart.reserve(sizeof(uint16_t) * kGridTable * n_pairs));
    bool use_cert = false;
    ICPX_TRY(launch_prepass(dim, n_pairs, m_max, d_dst, d_doff, tperm.as<uint16_t>(), grids.as<PairGrid>(), cstart.as<uint16_t>(),
                            hcert.as<float>(), &use_cert, stream));
    if (cert_out) *cert_out = use_cert;
    return ICPX_OK;
}

// The deferred sort of trimmed ICP2's matches (PairBatch::lazy_sort): one workgroup, pair 0 of a handle.
// Dynamic LDS: 12 bytes x npad (npad = the power of two >= n).
__global__ void __launch_bounds__(1024) pair_sort_matches_kernel(const int n, const int npad, PairState* __restrict__ st,
                                                                 int64_t* __restrict__ m_point, int64_t* __restrict__ m_pair,
                                                                 double* __restrict__ m_cost, const int32_t* __restrict__ pair_tmp) {
    extern __shared__ __attribute__((aligned(16))) double sort_lds[];
    if ((st->kd_flags & 16) == 0) return;  // (uniform: already sorted, or the matches of setup, which stay in point order)
    uint64_t* skey = reinterpret_cast<uint64_t*>(sort_lds);
    uint32_t* sval = reinterpret_cast<uint32_t*>(skey + npad);
    const int T = blockDim.x, tid = threadIdx.x;
    for (int i = tid; i < npad; i += T) {
        skey[i] = i < n ? static_cast<uint64_t>(__double_as_longlong(m_cost[i])) : ~0ULL;
        sval[i] = static_cast<uint32_t>(i);
    }
    __syncthreads();
    bitonic_sort(skey, sval, npad, T);
    for (int r = tid; r < n; r += T) {
        const uint32_t pt = sval[r];
        m_point[r] = pt;
        m_pair[r] = pair_tmp[pt];
        m_cost[r] = __longlong_as_double(static_cast<long long>(skey[r]));
    }
    __syncthreads();
    if (tid == 0) st->kd_flags &= ~static_cast<uint64_t>(16);
}

// [kdnode m][kdpos m] (u16), then 16 bytes whose first word counts the pairs a launch stopped at a tie
size_t kd_buffer_bytes(int64_t m_total) {
    const size_t pts = static_cast<size_t>(m_total > 0 ? m_total : 1);
    return ((4 * pts + 15) & ~static_cast<size_t>(15)) + 16;
}

icpx_criteria no_criteria() {
    icpx_criteria c;
    std::memset(&c, 0, sizeof(c));
    return c;
}

// Grow-only workspace of the handle-free entry points, one per (device, stream): calls on different
// streams of one device run concurrently (bench.py overlaps its 2D and its 3D batch this way); calls on
// the same stream serialise on `mu`.
struct Workspace {
    std::mutex mu;
    DevBuf src, dst, soff, doff, guesses, states, out, part_d2, part_idx, q, t, idx, d2, sched, cur, seeds;
    DevBuf tperm, grids, cstart, hcert;  // grid-pruned search of the batch entry point
    DevBuf kd, kd_words;                 // kd rule: trees of the target clouds; one counter (16 bytes) per fused run
    DevBuf kd_list;                      // kd rule of the stand-alone query: counter (16 bytes) + flagged queries
    GridIndex* grid = nullptr;    // grow-only uniform-grid index of the stand-alone queries
    BvhIndex* bvh = nullptr;      // likewise, ICPX_NN_BVH
    // pinned staging of the offset arrays, so that a stream-ordered call can return before they are consumed
    int64_t* h_off = nullptr;
    size_t h_off_cap = 0;
    hipEvent_t h_off_free = nullptr;  // recorded behind the last copy out of h_off
    // the offsets the device copy (soff) holds, when it was filled from h_off by a copy on this workspace's stream:
    // a call with the same offsets (a serving loop over one batch layout) skips the upload
    size_t off_cached = 0;
    const void* off_cached_dev = nullptr;
    icpx_ctx* fallback = nullptr;     // handle that runs the pairs of a batch the fused kernel cannot take
    int fb_dim = 0, fb_method = -1;
    int stage_offsets(int64_t entries) {
        if (h_off_free) ICPX_HIP(hipEventSynchronize(h_off_free));
        if (static_cast<size_t>(entries) > h_off_cap) {
            if (h_off) ICPX_HIP(hipHostFree(h_off));
            h_off = nullptr;
            h_off_cap = 0;
            const size_t want = static_cast<size_t>(entries) + static_cast<size_t>(entries) / 4 + 64;
            ICPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_off), sizeof(int64_t) * want, hipHostMallocDefault));
            h_off_cap = want;
        }
        if (!h_off_free) ICPX_HIP(hipEventCreateWithFlags(&h_off_free, hipEventDisableTiming));
        return ICPX_OK;
    }
};
struct WsKey {
    int device;
    hipStream_t stream;
    bool operator==(const WsKey& o) const { return device == o.device && stream == o.stream; }
};
struct WsKeyHash {
    size_t operator()(const WsKey& k) const {
        return std::hash<const void*>()(static_cast<const void*>(k.stream)) * 31u + static_cast<size_t>(k.device);
    }
};
std::mutex g_ws_mu;
std::unordered_map<WsKey, Workspace*, WsKeyHash> g_ws;
Workspace& workspace(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    const WsKey key{device, stream};
    auto it = g_ws.find(key);
    if (it == g_ws.end()) it = g_ws.emplace(key, new Workspace()).first;
    return *it->second;
}

}  // namespace
}  // namespace icpx

using namespace icpx;

struct icpx_ctx {
    int dim = 0, method = 0;
    icpx_params prm{};
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0, m = 0;
    bool begun = false;
    bool large = false;
    PairGeom geom{};
    DevBuf src, dst, cur, off, guess, m_point, m_pair, m_cost, pair_tmp, seeds, state, out;
    DevBuf tperm, grids, cstart, hcert;  // grid-pruned search (geom.prune)
    DevBuf kd;                           // kd rule: [kdnode m][kdpos m] (u16) and the paused-pairs word behind them
    bool use_cert = false;
    LargeIcp* lg = nullptr;
    // Host inputs of a fused pair are not copied to the device: they go into one pinned, device-visible arena
    // ([offsets][guess][source][target][result]) that the kernels read and write across the bus — a scan pair is 30 KB, and
    // the four staged copies, their stream synchronisation and the copy back were 50 us of a 140 us call.
    char* stage = nullptr;
    size_t stage_cap = 0;
    bool staged = false;                 // this begin's inputs live in the arena (else: in src / dst / off / guess / out)
    const double* p_src = nullptr;
    const double* p_dst = nullptr;
    const double* p_guess = nullptr;
    const int64_t* p_off = nullptr;
    icpx_state* p_out = nullptr;
};

namespace {

int check_handle(icpx_handle h) {
    if (!h) return fail(ICPX_ERR_INVALID, "null handle");
    return ICPX_OK;
}

// plain host memory (or pinned / registered host memory)?  Device and managed pointers keep the device-side copies.
bool is_host_pointer(const void* p) {
    if (!p) return true;
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (unregistered host memory reports an error)
        return true;
    }
    return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered;
}

constexpr size_t kStageMaxBytes = size_t(8) << 20;

int upload_pair(icpx_ctx* h, const double* source, int64_t n, const double* target, int64_t m,
                const double* guess) {
    const int D = h->dim, XN = (D + 1) * (D + 1);
    ICPX_TRY(h->cur.reserve(sizeof(double) * D * (n > 0 ? n : 1)));
    ICPX_TRY(h->m_point.reserve(sizeof(int64_t) * (n > 0 ? n : 1)));
    ICPX_TRY(h->m_pair.reserve(sizeof(int64_t) * (n > 0 ? n : 1)));
    ICPX_TRY(h->m_cost.reserve(sizeof(double) * (n > 0 ? n : 1)));
    ICPX_TRY(h->pair_tmp.reserve(sizeof(int32_t) * (n > 0 ? n : 1)));
    ICPX_TRY(h->seeds.reserve(sizeof(int32_t) * (n > 0 ? n : 1)));
    ICPX_TRY(h->state.reserve(sizeof(PairState)));
    ICPX_TRY(h->kd.reserve(kd_buffer_bytes(m)));
    const size_t src_bytes = sizeof(double) * D * static_cast<size_t>(n), dst_bytes = sizeof(double) * D * static_cast<size_t>(m);
    const size_t o_off = 0, o_guess = 32, o_out = o_guess + 128, o_src = (o_out + sizeof(icpx_state) + 63) & ~size_t(63);
    const size_t o_dst = (o_src + src_bytes + 63) & ~size_t(63), total = o_dst + dst_bytes + 64;
    const int64_t off[4] = {0, n, 0, m};
    h->staged = knobs().zero_copy != 0 && total <= kStageMaxBytes && is_host_pointer(source) && is_host_pointer(target) &&
                is_host_pointer(guess);
    if (h->staged) {
        // (kernels of an earlier begin may still be reading the arena: begin / iterate do not wait for theirs)
        ICPX_HIP(hipStreamSynchronize(h->stream));
        if (total > h->stage_cap) {
            if (h->stage) ICPX_HIP(hipHostFree(h->stage));
            h->stage = nullptr;
            h->stage_cap = 0;
            const size_t want = total + total / 4 + 4096;
            ICPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->stage), want, hipHostMallocPortable | hipHostMallocMapped));
            h->stage_cap = want;
        }
        std::memcpy(h->stage + o_off, off, sizeof(off));
        std::memcpy(h->stage + o_guess, guess, sizeof(double) * XN);
        if (n > 0) std::memcpy(h->stage + o_src, source, src_bytes);
        if (m > 0) std::memcpy(h->stage + o_dst, target, dst_bytes);
        h->p_off = reinterpret_cast<const int64_t*>(h->stage + o_off);
        h->p_guess = reinterpret_cast<const double*>(h->stage + o_guess);
        h->p_out = reinterpret_cast<icpx_state*>(h->stage + o_out);
        h->p_src = reinterpret_cast<const double*>(h->stage + o_src);
        h->p_dst = reinterpret_cast<const double*>(h->stage + o_dst);
    } else {
        ICPX_TRY(h->src.reserve(sizeof(double) * D * (n > 0 ? n : 1)));
        ICPX_TRY(h->dst.reserve(sizeof(double) * D * (m > 0 ? m : 1)));
        ICPX_TRY(h->off.reserve(sizeof(int64_t) * 4));
        ICPX_TRY(h->guess.reserve(sizeof(double) * 16));
        ICPX_TRY(h->out.reserve(sizeof(icpx_state)));
        if (n > 0) ICPX_HIP(hipMemcpyAsync(h->src.p, source, src_bytes, hipMemcpyDefault, h->stream));
        if (m > 0) ICPX_HIP(hipMemcpyAsync(h->dst.p, target, dst_bytes, hipMemcpyDefault, h->stream));
        ICPX_HIP(hipMemcpyAsync(h->off.p, off, sizeof(off), hipMemcpyHostToDevice, h->stream));
        ICPX_HIP(hipMemcpyAsync(h->guess.p, guess, sizeof(double) * XN, hipMemcpyDefault, h->stream));
        // the pageable host staging above must be consumed before `off` goes out of scope
        ICPX_HIP(hipStreamSynchronize(h->stream));
        h->p_off = h->off.as<int64_t>();
        h->p_guess = h->guess.as<double>();
        h->p_out = h->out.as<icpx_state>();
        h->p_src = h->src.as<double>();
        h->p_dst = h->dst.as<double>();
    }
    h->n = n;
    h->m = m;
    if (h->geom.prune)
        ICPX_TRY(enqueue_prepass(D, 1, m, m, h->p_dst, h->p_off + 2, h->tperm, h->grids, h->cstart,
                                 h->hcert, &h->use_cert, h->stream));
    return ICPX_OK;
}

void fill_batch(icpx_ctx* h, PairBatch& pb, const DeviceInfo& di) {
    std::memset(&pb, 0, sizeof(pb));
    pb.dim = h->dim;
    pb.method = h->method;
    pb.src = h->p_src;
    pb.src_off = h->p_off;
    pb.dst = h->p_dst;
    pb.dst_off = h->p_off + 2;
    pb.guesses = h->p_guess;
    pb.cur = h->cur.as<double>();
    pb.m_point = h->m_point.as<int64_t>();
    pb.m_pair = h->m_pair.as<int64_t>();
    pb.m_cost = h->m_cost.as<double>();
    pb.pair_tmp = h->pair_tmp.as<int32_t>();
    pb.seeds = h->seeds.as<int32_t>();
    pb.states = h->state.as<PairState>();
    pb.out_states = h->p_out;
    pb.crit = no_criteria();
    pb.overlap_rate = h->prm.overlap_rate;
    pb.max_distance = h->prm.max_distance;
    pb.clock_to_s = di.clock_to_s;
    pb.lazy_sort = 1;
    pb.kdnode = h->kd.as<uint16_t>();
    pb.kdpos = pb.kdnode + (h->m > 0 ? h->m : 1);
    pb.kd_paused = reinterpret_cast<int32_t*>(h->kd.as<char>() + kd_buffer_bytes(h->m) - 16);
    if (h->geom.prune) {
        pb.tperm = h->tperm.as<uint16_t>();
        pb.grids = h->grids.as<PairGrid>();
        pb.cstart = h->cstart.as<uint16_t>();
        pb.hcert = h->use_cert ? h->hcert.as<float>() : nullptr;
    }
}

int begin_common(icpx_ctx* h, const double* source, int64_t n, const double* target, int64_t m,
                 const double* guess) {
    if (n < 0 || m < 0) return fail(ICPX_ERR_INVALID, "negative cloud size");
    if ((n > 0 && !source) || (m > 0 && !target) || !guess)
        return fail(ICPX_ERR_INVALID, "null source/target/guess");
    if (n > INT32_MAX / 4 || m > INT32_MAX / 4) return fail(ICPX_ERR_UNSUPPORTED, "cloud too large");
    h->begun = false;
    // A single pair in the fused kernel runs on ONE compute unit; beyond a modest size the
    // multi-kernel path (tiled correspondence over the whole chip) has the lower latency.
    // Development override: ICPX_FUSED_SINGLE_MAX = largest n*m that stays fused
    double single_max = knobs().fused_single_max;
    if (single_max < 0.0) single_max = kFusedSingleMaxWork;
    // Single pairs keep the brute-force filter kernels (and the measured crossover to the multi-kernel path).  The
    // grid-pruned kernels of the batch path are 1.5 - 2.7x faster per iteration on UNIFORM clouds of every size the fused
    // kernel takes (round 3, ICPX_PRUNE_HANDLES=1: 2D trimmed 1000 points 48 -> 18 us, 3D 1024 68 -> 34, 3D 4096
    // 107 -> 70 us) but 6x SLOWER on the reference's lidar scans (scan1, 22 iterations: 1.42 -> 8.6 ms): the points of a
    // scan lie on curves, most cells of a uniform grid are empty and the occupied ones hold dozens of points, so the
    // boxes overflow the candidate list and fall back to whole-wave scans.  Handles — the reference's use — take the
    // robust path; ICPX_PRUNE_HANDLES=1 selects the pruned kernels (then for every size that fits; tests run both).
    const bool prune_handles = knobs().prune_handles != 0;
    const bool fits = fused_fits(h->dim, h->method, n, m, &h->geom, true, prune_handles);
    bool fused = h->prm.nn_mode != ICPX_NN_GRID && h->prm.nn_mode != ICPX_NN_GRID_F32 && h->prm.nn_mode != ICPX_NN_BVH && fits &&
                 (h->geom.prune || static_cast<double>(n) * static_cast<double>(m) <= single_max) &&
                 h->method != ICPX_FEATURE_AWARE &&  // feature_aware and the opt-in corrected modes run on
                 h->prm.flags == 0;                  // the multi-kernel path only
    // (nn = brute asks for the brute-force search: no pruned kernel)
    if (fused && h->geom.prune && h->prm.nn_mode == ICPX_NN_BRUTE) {
        fused = fused_fits(h->dim, h->method, n, m, &h->geom, true, false) &&
                static_cast<double>(n) * static_cast<double>(m) <= single_max;
    }
    // the tiled brute force of the large path is only chosen explicitly
    h->large = !fused;
    if (h->large && (n == 0 || m == 0)) {
        // empty clouds carry no work: run them through the fused kernel's no-op path
        h->large = !fused_fits(h->dim, h->method, n, m, &h->geom, true, prune_handles);
    }
    return ICPX_OK;
}

// Nothing may throw across the C ABI (icpx.h): std::string / std::vector / unordered_map / new inside the
// entry points can, so every entry point runs its body through this.
template <typename F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { return fail(ICPX_ERR_NOMEM, "out of host memory"); } catch (...) { return ICPX_ERR_NOMEM; }
    } catch (const std::exception& e) {
        try { return fail(ICPX_ERR_HIP, "unexpected exception: %s", e.what()); } catch (...) { return ICPX_ERR_HIP; }
    } catch (...) {
        return ICPX_ERR_HIP;
    }
}

// ICPDriver's loop is a HOST loop in the reference (driver.h:63) and can be interrupted; here it runs
// inside one kernel, so criteria under which it can never end are rejected up front.
int check_criteria(const icpx_criteria* c) {
    if (!(c->has & (ICPX_HAS_MAX_ITER | ICPX_HAS_TIME_LIMIT)))
        return fail(ICPX_ERR_INVALID,
                    "criteria need max_iterations or a time limit: the converge loop runs on the device and could not be "
                    "interrupted (the reference's loop without them never ends either, driver.h:43-44)");
    if ((c->has & ICPX_HAS_MIN_ITER) && (c->has & ICPX_HAS_MAX_ITER) && c->min_iterations > c->max_iterations)
        return fail(ICPX_ERR_INVALID, "min_iterations > max_iterations (driver.h:81,91 assert this)");
    if ((c->has & ICPX_HAS_TIME_LIMIT) && !(c->time_limit_s == c->time_limit_s))
        return fail(ICPX_ERR_INVALID, "time limit is NaN");
    return ICPX_OK;
}

int converge_impl(icpx_handle h, const double* source, int64_t n, const double* target, int64_t m,
                  const double* guess, const icpx_criteria* criteria, icpx_state* out) {
    auto t0 = std::chrono::steady_clock::now();  // driver.h:53
    ICPX_TRY(check_handle(h));
    if (!criteria || !out) return fail(ICPX_ERR_INVALID, "null criteria/out");
    ICPX_TRY(check_criteria(criteria));
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(h->device, &di));
    ICPX_TRY(begin_common(h, source, n, target, m, guess));
    if (h->large) {
        if (!h->lg) ICPX_TRY(large_create(h->dim, h->method, h->prm, di->clock_to_s, h->stream, &h->lg));
        ICPX_TRY(large_begin(h->lg, source, n, target, m, guess));
        h->n = n;
        h->m = m;
        h->begun = true;
        return large_converge(h->lg, criteria, t0, out);
    }
    ICPX_TRY(upload_pair(h, source, n, target, m, guess));
    PairBatch pb;
    fill_batch(h, pb, *di);
    pb.crit = *criteria;
    pb.do_setup = 1;
    pb.check = 1;
    pb.max_steps = UINT64_MAX;
    pb.host_elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ICPX_TRY(launch_pairs(pb, 1, h->geom, h->stream));
    if (h->staged) {
        ICPX_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(out, h->p_out, sizeof(icpx_state));  // (the kernel wrote it into the arena)
    } else {
        ICPX_HIP(hipMemcpyAsync(out, h->p_out, sizeof(icpx_state), hipMemcpyDefault, h->stream));
        ICPX_HIP(hipStreamSynchronize(h->stream));
    }
    h->begun = true;
    return ICPX_OK;
}


// ---- batches ----------------------------------------------------------------------------------

// device memory a fused batch launch needs beyond its inputs and outputs
struct BatchScratch {
    int64_t* soff;      // n_pairs + 1 source offsets
    int64_t* doff;      // n_pairs + 1 target offsets
    PairState* states;  // n_pairs
    int32_t* sched;     // 2 * (n_pairs + 1) scheduler words (every fused run takes its pairs + 1)
    double* cur;        // 3D: n_total running clouds
    int32_t* seeds;     // n_total previous neighbours
    uint16_t* tperm;    // m_total cell order of the targets
    PairGrid* grids;    // n_pairs
    uint16_t* cstart;   // n_pairs x kGridTable
    float* hcert;       // m_total certificate radii
    uint16_t* kdnode;   // m_total: kd rule, the target clouds' trees (in-order layout) ...
    uint16_t* kdpos;    // m_total: ... and their inverse
    int32_t* kd_paused; // one word per fused run: pairs stopped at a tie
};

struct BatchCall {
    int dim, method;
    icpx_params prm;
    int64_t n_pairs;
    const double* d_src;    // device
    const double* d_dst;    // device
    const double* d_guess;  // device or null
    const int64_t* h_soff;  // host
    const int64_t* h_doff;  // host
    icpx_criteria crit;
    icpx_state* d_out;      // device
};

// pairs [lo, hi) of the call through ONE fused launch (they all fit `geom`)
int enqueue_fused_run(const BatchCall& c, int64_t lo, int64_t hi, int64_t run_index, const PairGeom& geom,
                      const BatchScratch& sc, const DeviceInfo& di, hipStream_t stream,
                      std::chrono::steady_clock::time_point t0) {
    const int XN = (c.dim + 1) * (c.dim + 1);
    const int64_t cnt = hi - lo;
    int64_t m_max = 0;
    for (int64_t p = lo; p < hi; p++) m_max = std::max(m_max, c.h_doff[p + 1] - c.h_doff[p]);
    PairBatch pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.dim = c.dim;
    pb.method = c.method;
    pb.src = c.d_src;
    pb.src_off = sc.soff + lo;
    pb.dst = c.d_dst;
    pb.dst_off = sc.doff + lo;
    pb.guesses = c.d_guess ? c.d_guess + lo * XN : nullptr;
    pb.states = sc.states + lo;
    pb.out_states = c.d_out + lo;
    pb.crit = c.crit;
    pb.overlap_rate = c.prm.overlap_rate;
    pb.max_distance = c.prm.max_distance;
    pb.clock_to_s = di.clock_to_s;
    pb.do_setup = 1;
    pb.check = 1;
    pb.max_steps = UINT64_MAX;
    // scheduler words (launch_one decides whether to use them) and, for 3D, the running clouds
    // that a pair carries from one workgroup to the next
    pb.sched_state = sc.sched + lo + run_index;
    if (c.dim == 3) pb.cur = sc.cur;
    if (geom.filter || geom.prune) pb.seeds = sc.seeds;  // the seeded search's previous neighbours travel with the pair
    if (geom.prune) {
        // the pre-pass of a pair's target cloud runs inside the pair kernel's setup visit when the kernel's per-wave
        // staging area can hold its histogram (the launch then depends on nothing before it and can share the chip
        // with another batch from its first workgroup on); as a launch of its own otherwise
        const bool fuse = knobs().fuse_prepass != 0 &&
                          prepare_hist_bytes(c.dim, m_max) <= static_cast<size_t>(geom.threads / 64) * coop_wave_bytes(c.dim);
        if (fuse) {
            pb.prepare = 1;
            pb.hcert = sc.hcert;
        } else {
            bool use_cert = false;
            ICPX_TRY(launch_prepass(c.dim, cnt, m_max, c.d_dst, sc.doff + lo, sc.tperm, sc.grids + lo, sc.cstart + lo * kGridTable,
                                    sc.hcert, &use_cert, stream));
            pb.hcert = use_cert ? sc.hcert : nullptr;
        }
        pb.tperm = sc.tperm;
        pb.grids = sc.grids + lo;
        pb.cstart = sc.cstart + lo * kGridTable;
    }
    pb.kdnode = sc.kdnode;
    pb.kdpos = sc.kdpos;
    pb.kd_paused = sc.kd_paused + 4 * run_index;
    pb.host_elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    PairGeom g = geom;
    return launch_pairs(pb, cnt, g, stream);
}

// Splits the pairs of a call into maximal runs that fit ONE fused geometry (each run = one launch) and
// single pairs that do not (too many points for the fused kernel, feature_aware, opt-in flags): those go
// through `fallback` (a handle on the multi-kernel path), or fail the call when there is none.
struct BatchRun {
    int64_t lo, hi;
    bool fused;
    PairGeom geom;
};
template <typename Runs>
void plan_batch(const BatchCall& c, Runs& runs) {
    const bool never_fused = c.method == ICPX_FEATURE_AWARE || c.prm.flags != 0;
    // geometry of the whole call first: most batches are one run
    int64_t p = 0;
    while (p < c.n_pairs) {
        PairGeom g1;
        const bool fits = !never_fused && fused_fits(c.dim, c.method, c.h_soff[p + 1] - c.h_soff[p],
                                                     c.h_doff[p + 1] - c.h_doff[p], &g1, false, true);
        int64_t q = p + 1;
        if (!fits) {
            runs.push_back(BatchRun{p, q, false, PairGeom{}});
            p = q;
            continue;
        }
        int64_t n_max = c.h_soff[p + 1] - c.h_soff[p], m_max = c.h_doff[p + 1] - c.h_doff[p];
        PairGeom g = g1;
        while (q < c.n_pairs) {
            const int64_t n = c.h_soff[q + 1] - c.h_soff[q], m = c.h_doff[q + 1] - c.h_doff[q];
            PairGeom gq;
            if (!fused_fits(c.dim, c.method, n, m, &gq, false, true)) break;
            PairGeom gj;
            if (!fused_fits(c.dim, c.method, std::max(n_max, n), std::max(m_max, m), &gj, false, true)) break;
            n_max = std::max(n_max, n);
            m_max = std::max(m_max, m);
            g = gj;
            q++;
        }
        runs.push_back(BatchRun{p, q, true, g});
        p = q;
    }
}

int validate_batch_args(int dim, int method, const icpx_params* params, int64_t n_pairs, const double* source,
                        const int64_t* source_offsets, const double* target, const int64_t* target_offsets,
                        const icpx_criteria* criteria, icpx_state* out_states) {
    if (dim != 2 && dim != 3) return fail(ICPX_ERR_INVALID, "dim must be 2 or 3 (got %d)", dim);
    if (method != ICPX_VANILLA && method != ICPX_TRIMMED && method != ICPX_FEATURE_AWARE)
        return fail(ICPX_ERR_INVALID, "unknown method %d", method);
    if (method == ICPX_FEATURE_AWARE && dim != 2)
        return fail(ICPX_ERR_INVALID, "feature_aware is an ICP2 method (lib/icp/icp.cpp:17-21)");
    if (params && (params->flags < 0 || params->flags > 7)) return fail(ICPX_ERR_INVALID, "unknown flags %d", params->flags);
    if (n_pairs < 0) return fail(ICPX_ERR_INVALID, "negative n_pairs");
    if (n_pairs == 0) return ICPX_OK;
    if (!source_offsets || !target_offsets || !criteria || !out_states)
        return fail(ICPX_ERR_INVALID, "null offsets/criteria/out_states");
    ICPX_TRY(check_criteria(criteria));
    if (n_pairs > (1 << 30)) return fail(ICPX_ERR_UNSUPPORTED, "too many pairs");
    for (int64_t p = 0; p < n_pairs; p++) {
        if (source_offsets[p + 1] < source_offsets[p] || target_offsets[p + 1] < target_offsets[p])
            return fail(ICPX_ERR_INVALID, "offsets must be non-decreasing (pair %lld)", (long long)p);
    }
    if (source_offsets[0] < 0 || target_offsets[0] < 0) return fail(ICPX_ERR_INVALID, "negative offset");
    if ((source_offsets[n_pairs] > source_offsets[0] && !source) || (target_offsets[n_pairs] > target_offsets[0] && !target))
        return fail(ICPX_ERR_INVALID, "null source/target");
    return ICPX_OK;
}

// one device, one stream; see icpx_converge_batch in icpx.h
int converge_batch_impl(int dim, int method, const icpx_params* params, int64_t n_pairs, const double* source,
                        const int64_t* source_offsets, const double* target, const int64_t* target_offsets,
                        const double* guesses, const icpx_criteria* criteria, icpx_state* out_states) {
    auto t0 = std::chrono::steady_clock::now();
    ICPX_TRY(validate_batch_args(dim, method, params, n_pairs, source, source_offsets, target, target_offsets, criteria,
                                 out_states));
    if (n_pairs == 0) return ICPX_OK;
    icpx_params prm;
    icpx_default_params(&prm);
    if (params) prm = *params;
    int dev = 0;
    ICPX_TRY(resolve_device(prm.device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    hipStream_t stream = static_cast<hipStream_t>(prm.stream);
    const int64_t n_total = source_offsets[n_pairs], m_total = target_offsets[n_pairs];
    const int XN = (dim + 1) * (dim + 1);

    Workspace& ws = workspace(dev, stream);
    std::lock_guard<std::mutex> lk(ws.mu);
    BatchCall c;
    c.dim = dim;
    c.method = method;
    c.prm = prm;
    c.n_pairs = n_pairs;
    c.h_soff = source_offsets;
    c.h_doff = target_offsets;
    c.crit = *criteria;
    std::vector<BatchRun> runs;
    plan_batch(c, runs);
    bool any_fused = false, any_fallback = false;
    for (const BatchRun& r : runs) (r.fused ? any_fused : any_fallback) = true;

    // a call whose arrays all live on the device is stream-ordered (returns once the work is enqueued)
    bool must_sync = any_fallback;
    c.d_src = source;
    c.d_dst = target;
    c.d_guess = guesses;
    c.d_out = out_states;
    if (any_fused) {
        if (n_total > 0 && !is_device_ptr(source)) {
            ICPX_TRY(ws.src.reserve(sizeof(double) * dim * n_total));
            ICPX_HIP(hipMemcpyAsync(ws.src.p, source, sizeof(double) * dim * n_total, hipMemcpyHostToDevice, stream));
            c.d_src = ws.src.as<double>();
            must_sync = true;
        }
        if (m_total > 0 && !is_device_ptr(target)) {
            ICPX_TRY(ws.dst.reserve(sizeof(double) * dim * m_total));
            ICPX_HIP(hipMemcpyAsync(ws.dst.p, target, sizeof(double) * dim * m_total, hipMemcpyHostToDevice, stream));
            c.d_dst = ws.dst.as<double>();
            must_sync = true;
        }
        if (guesses && !is_device_ptr(guesses)) {
            ICPX_TRY(ws.guesses.reserve(sizeof(double) * XN * n_pairs));
            ICPX_HIP(hipMemcpyAsync(ws.guesses.p, guesses, sizeof(double) * XN * n_pairs, hipMemcpyHostToDevice, stream));
            c.d_guess = ws.guesses.as<double>();
            must_sync = true;
        }
    }
    const bool out_on_device = is_device_ptr(out_states);
    if (!out_on_device) {
        ICPX_TRY(ws.out.reserve(sizeof(icpx_state) * n_pairs));
        c.d_out = ws.out.as<icpx_state>();
        must_sync = true;
    }
    if (any_fused) {
        // offsets: through pinned staging, so the caller's arrays are free when the call returns
        const size_t entries = 2 * static_cast<size_t>(n_pairs + 1);
        ICPX_TRY(ws.soff.reserve(sizeof(int64_t) * entries));
        const bool same = ws.off_cached == entries && ws.off_cached_dev == ws.soff.p && ws.h_off != nullptr &&
                          std::memcmp(ws.h_off, source_offsets, sizeof(int64_t) * (n_pairs + 1)) == 0 &&
                          std::memcmp(ws.h_off + n_pairs + 1, target_offsets, sizeof(int64_t) * (n_pairs + 1)) == 0;
        if (!same) {
            ws.off_cached = 0;
            ICPX_TRY(ws.stage_offsets(static_cast<int64_t>(entries)));
            std::memcpy(ws.h_off, source_offsets, sizeof(int64_t) * (n_pairs + 1));
            std::memcpy(ws.h_off + n_pairs + 1, target_offsets, sizeof(int64_t) * (n_pairs + 1));
            ICPX_HIP(hipMemcpyAsync(ws.soff.p, ws.h_off, sizeof(int64_t) * entries, hipMemcpyHostToDevice, stream));
            ICPX_HIP(hipEventRecord(ws.h_off_free, stream));
            ws.off_cached = entries;
            ws.off_cached_dev = ws.soff.p;
        }
        ICPX_TRY(ws.states.reserve(sizeof(PairState) * n_pairs));
        ICPX_TRY(ws.sched.reserve(sizeof(int32_t) * (2 * (n_pairs + 1) + 4)));  // (+4: launch_one clears whole 16-byte units)
        if (dim == 3) ICPX_TRY(ws.cur.reserve(sizeof(double) * 3 * (n_total > 0 ? n_total : 1)));
        ICPX_TRY(ws.seeds.reserve(sizeof(int32_t) * (n_total > 0 ? n_total : 1)));
        ICPX_TRY(ws.tperm.reserve(sizeof(uint16_t) * (m_total > 0 ? m_total : 1)));
        ICPX_TRY(ws.grids.reserve(sizeof(PairGrid) * n_pairs));
        ICPX_TRY(ws.cstart.reserve(sizeof(uint16_t) * kGridTable * n_pairs));
        ICPX_TRY(ws.hcert.reserve(sizeof(float) * (m_total > 0 ? m_total : 1)));
        ICPX_TRY(ws.kd.reserve(kd_buffer_bytes(m_total)));
        ICPX_TRY(ws.kd_words.reserve(16 * runs.size()));
    }
    BatchScratch sc;
    sc.soff = ws.soff.as<int64_t>();
    sc.doff = sc.soff ? sc.soff + n_pairs + 1 : nullptr;
    sc.states = ws.states.as<PairState>();
    sc.sched = ws.sched.as<int32_t>();
    sc.cur = ws.cur.as<double>();
    sc.seeds = ws.seeds.as<int32_t>();
    sc.tperm = ws.tperm.as<uint16_t>();
    sc.grids = ws.grids.as<PairGrid>();
    sc.cstart = ws.cstart.as<uint16_t>();
    sc.hcert = ws.hcert.as<float>();
    sc.kdnode = ws.kd.as<uint16_t>();
    sc.kdpos = sc.kdnode ? sc.kdnode + (m_total > 0 ? m_total : 1) : nullptr;
    sc.kd_paused = ws.kd_words.as<int32_t>();

    int64_t run_index = 0;
    for (const BatchRun& r : runs) {
        if (r.fused) {
            ICPX_TRY(enqueue_fused_run(c, r.lo, r.hi, run_index++, r.geom, sc, *di, stream, t0));
            continue;
        }
        // a pair the fused kernel cannot take: the multi-kernel path, through a handle of this workspace
        if (!ws.fallback || ws.fb_dim != dim || ws.fb_method != method || ws.fallback->prm.flags != prm.flags ||
            ws.fallback->prm.overlap_rate != prm.overlap_rate || ws.fallback->prm.max_distance != prm.max_distance ||
            ws.fallback->prm.feature_weight != prm.feature_weight ||
            ws.fallback->prm.symmetric_neighbors != prm.symmetric_neighbors || ws.fallback->prm.nn_mode != prm.nn_mode) {
            if (ws.fallback) (void)icpx_destroy(ws.fallback);
            ws.fallback = nullptr;
            icpx_params fp = prm;
            fp.device = dev;
            ICPX_TRY(icpx_create(dim, method, &fp, &ws.fallback));
            ws.fb_dim = dim;
            ws.fb_method = method;
        }
        const int64_t p = r.lo;
        double ident[16] = {0};
        for (int e = 0; e <= dim; e++) ident[e * (dim + 2)] = 1.0;
        double gbuf[16];
        const double* gp = ident;
        if (guesses) {
            ICPX_HIP(hipMemcpy(gbuf, guesses + p * XN, sizeof(double) * XN, hipMemcpyDefault));
            gp = gbuf;
        }
        icpx_state st;
        std::memset(&st, 0, sizeof(st));
        ICPX_TRY(converge_impl(ws.fallback, source ? source + source_offsets[p] * dim : nullptr,
                               source_offsets[p + 1] - source_offsets[p], target ? target + target_offsets[p] * dim : nullptr,
                               target_offsets[p + 1] - target_offsets[p], gp, criteria, &st));
        ICPX_HIP(hipMemcpy(c.d_out + p, &st, sizeof(st), hipMemcpyDefault));
    }
    if (!out_on_device)
        ICPX_HIP(hipMemcpyAsync(out_states, c.d_out, sizeof(icpx_state) * n_pairs, hipMemcpyDeviceToHost, stream));
    if (must_sync) ICPX_HIP(hipStreamSynchronize(stream));
    return ICPX_OK;
}


// ---- several devices --------------------------------------------------------------------------

std::mutex g_multi_mu;
std::unordered_map<long long, hipStream_t> g_multi_streams;  // (device, shard slot) -> stream of that shard
int multi_stream(int dev, int slot, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_multi_mu);
    const long long key = static_cast<long long>(dev) * 4096 + slot;
    auto it = g_multi_streams.find(key);
    if (it == g_multi_streams.end()) {
        DeviceGuard g;
        ICPX_TRY(g.enter(dev));
        hipStream_t st = nullptr;
        ICPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        it = g_multi_streams.emplace(key, st).first;
    }
    *out = it->second;
    return ICPX_OK;
}

// contiguous block of pairs of shard r out of `shards`: pair p -> shard floor(p * shards / n_pairs)
// (icp_amd/shard.py computes the same bounds for the one-process-per-GPU launch)
void shard_bounds(int64_t n_pairs, int r, int shards, int64_t* lo, int64_t* hi) {
    *lo = (n_pairs * r + shards - 1) / shards;
    *hi = (n_pairs * (r + 1) + shards - 1) / shards;
}

int converge_batch_multi_impl(int dim, int method, const icpx_params* params, int device_count, const int* devices,
                              int64_t n_pairs, const double* source, const int64_t* source_offsets, const double* target,
                              const int64_t* target_offsets, const double* guesses, const icpx_criteria* criteria,
                              icpx_state* out_states) {
    if (device_count < 1 || device_count > 4096) return fail(ICPX_ERR_INVALID, "device_count must be in [1, 4096]");
    ICPX_TRY(validate_batch_args(dim, method, params, n_pairs, source, source_offsets, target, target_offsets, criteria,
                                 out_states));
    if (n_pairs == 0) return ICPX_OK;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return fail(ICPX_ERR_NO_DEVICE, "no HIP device visible; icpx has no CPU path");
    }
    std::vector<int> devs(device_count);
    std::vector<int> slot(device_count, 0);
    for (int r = 0; r < device_count; r++) {
        devs[r] = devices ? devices[r] : r;
        if (devs[r] < 0 || devs[r] >= visible)
            return fail(ICPX_ERR_INVALID, "device %d out of range (%d visible)", devs[r], visible);
        for (int q = 0; q < r; q++) slot[r] += devs[q] == devs[r];  // a device listed twice gets two streams
    }
    icpx_params base;
    icpx_default_params(&base);
    if (params) base = *params;
    const int XN = (dim + 1) * (dim + 1);
    std::vector<int> rc(device_count, ICPX_OK);
    std::vector<std::string> err(device_count);
    auto work = [&](int r) {
        rc[r] = guarded([&]() -> int {
            int64_t lo, hi;
            shard_bounds(n_pairs, r, device_count, &lo, &hi);
            if (hi <= lo) return ICPX_OK;
            icpx_params prm = base;
            prm.device = devs[r];
            hipStream_t st = nullptr;
            ICPX_TRY(multi_stream(devs[r], slot[r], &st));
            prm.stream = st;
            // the shard's offsets count from its first point
            std::vector<int64_t> so(hi - lo + 1), to(hi - lo + 1);
            for (int64_t p = lo; p <= hi; p++) {
                so[p - lo] = source_offsets[p] - source_offsets[lo];
                to[p - lo] = target_offsets[p] - target_offsets[lo];
            }
            int rcode = converge_batch_impl(dim, method, &prm, hi - lo, source ? source + source_offsets[lo] * dim : nullptr,
                                            so.data(), target ? target + target_offsets[lo] * dim : nullptr, to.data(),
                                            guesses ? guesses + lo * XN : nullptr, criteria, out_states + lo);
            if (rcode != ICPX_OK) return rcode;
            // the gather: every shard's records are in the caller's array when its stream has drained
            DeviceGuard g;
            ICPX_TRY(g.enter(devs[r]));
            ICPX_HIP(hipStreamSynchronize(st));
            return ICPX_OK;
        });
        if (rc[r] != ICPX_OK) err[r] = last_error_ref();
    };
    // one host thread per shard (the calling thread takes shard 0)
    std::vector<std::thread> threads;
    for (int r = 1; r < device_count; r++) threads.emplace_back(work, r);
    work(0);
    for (auto& t : threads) t.join();
    for (int r = 0; r < device_count; r++)
        if (rc[r] != ICPX_OK) return fail(rc[r], "shard %d (device %d): %s", r, devs[r], err[r].c_str());
    return ICPX_OK;
}

// ---- the entry point that allocates nothing ------------------------------------------------------

constexpr int64_t kNoAllocMaxPairs = ICPX_NOALLOC_MAX_PAIRS;
constexpr int64_t kNoAllocMaxPoints = ICPX_NOALLOC_MAX_POINTS;
constexpr size_t align256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
constexpr size_t kScrOff = 0;
constexpr size_t kScrStates = kScrOff + align256(sizeof(int64_t) * 2 * (kNoAllocMaxPairs + 1));
constexpr size_t kScrSched = kScrStates + align256(sizeof(PairState) * kNoAllocMaxPairs);
constexpr size_t kScrCur = kScrSched + align256(sizeof(int32_t) * 2 * (kNoAllocMaxPairs + 1));
constexpr size_t kScrSeeds = kScrCur + align256(sizeof(double) * 3 * kNoAllocMaxPoints);
constexpr size_t kScrTperm = kScrSeeds + align256(sizeof(int32_t) * kNoAllocMaxPoints);
constexpr size_t kScrGrids = kScrTperm + align256(sizeof(uint16_t) * kNoAllocMaxPoints);
constexpr size_t kScrCstart = kScrGrids + align256(sizeof(PairGrid) * kNoAllocMaxPairs);
constexpr size_t kScrHcert = kScrCstart + align256(sizeof(uint16_t) * kGridTable * kNoAllocMaxPairs);
constexpr size_t kScrKd = kScrHcert + align256(sizeof(float) * kNoAllocMaxPoints);
constexpr size_t kScrKdWords = kScrKd + align256(sizeof(uint16_t) * 2 * kNoAllocMaxPoints);
constexpr size_t kScratchBytes = kScrKdWords + align256(16 * kNoAllocMaxPairs);

// Scratch of icpx_converge_batch_noalloc: every GPU the library runs on holds its own copy of this array
// (about 20 MB) until the process ends.  In host code the NAME is only a stand-in; the device address
// comes from hipGetSymbolAddress (scratch_for_device), once per device, and only that pointer is used.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

std::mutex g_scratch_mu;
std::unordered_map<int, unsigned char*> g_scratch_ptr;  // per device
std::unordered_map<int, std::mutex*> g_scratch_call_mu;
// `dev` must be the current device
int scratch_for_device(int dev, unsigned char** out, std::mutex** call_mu) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto it = g_scratch_ptr.find(dev);
    if (it == g_scratch_ptr.end()) {
        void* p = nullptr;
        ICPX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(ICPX_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", dev);
        it = g_scratch_ptr.emplace(dev, static_cast<unsigned char*>(p)).first;
        g_scratch_call_mu.emplace(dev, new std::mutex());
    }
    *out = it->second;
    if (call_mu) *call_mu = g_scratch_call_mu[dev];
    return ICPX_OK;
}

struct FixedRuns {
    BatchRun v[ICPX_NOALLOC_MAX_PAIRS];
    int64_t count = 0;
    void push_back(const BatchRun& r) { v[count++] = r; }  // plan_batch emits at most one run per pair
};

int converge_batch_noalloc_impl(int dim, int method, const icpx_params* params, int64_t n_pairs, const double* source,
                                const int64_t* source_offsets, const double* target, const int64_t* target_offsets,
                                const double* guesses, const icpx_criteria* criteria, icpx_state* out_states) {
    auto t0 = std::chrono::steady_clock::now();
    ICPX_TRY(validate_batch_args(dim, method, params, n_pairs, source, source_offsets, target, target_offsets, criteria,
                                 out_states));
    if (n_pairs == 0) return ICPX_OK;
    if (method == ICPX_FEATURE_AWARE || (params && params->flags != 0))
        return fail(ICPX_ERR_UNSUPPORTED, "icpx_converge_batch_noalloc: vanilla and trimmed without flags only");
    if (n_pairs > kNoAllocMaxPairs)
        return fail(ICPX_ERR_UNSUPPORTED, "icpx_converge_batch_noalloc: %lld pairs exceed the limit of %lld",
                    (long long)n_pairs, (long long)kNoAllocMaxPairs);
    if (source_offsets[0] != 0 || target_offsets[0] != 0)
        return fail(ICPX_ERR_INVALID, "icpx_converge_batch_noalloc: offsets must start at 0");
    const int64_t n_total = source_offsets[n_pairs], m_total = target_offsets[n_pairs];
    if (n_total > kNoAllocMaxPoints || m_total > kNoAllocMaxPoints)
        return fail(ICPX_ERR_UNSUPPORTED,
                    "icpx_converge_batch_noalloc: %lld source / %lld target points exceed the limit of %lld per side",
                    (long long)n_total, (long long)m_total, (long long)kNoAllocMaxPoints);
    icpx_params prm;
    icpx_default_params(&prm);
    if (params) prm = *params;
    int dev = 0;
    ICPX_TRY(resolve_device(prm.device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    hipStream_t stream = static_cast<hipStream_t>(prm.stream);
    if ((n_total > 0 && !is_device_ptr(source)) || (m_total > 0 && !is_device_ptr(target)) ||
        (guesses && !is_device_ptr(guesses)) || !is_device_ptr(out_states))
        return fail(ICPX_ERR_INVALID,
                    "icpx_converge_batch_noalloc stages nothing: source, target, guesses and out_states must be device memory");
    BatchCall c;
    c.dim = dim;
    c.method = method;
    c.prm = prm;
    c.n_pairs = n_pairs;
    c.d_src = source;
    c.d_dst = target;
    c.d_guess = guesses;
    c.h_soff = source_offsets;
    c.h_doff = target_offsets;
    c.crit = *criteria;
    c.d_out = out_states;
    FixedRuns runs;
    plan_batch(c, runs);
    for (int64_t i = 0; i < runs.count; i++)
        if (!runs.v[i].fused)
            return fail(ICPX_ERR_UNSUPPORTED,
                        "icpx_converge_batch_noalloc: pair %lld (%lld x %lld points) exceeds the fused kernel's limits",
                        (long long)runs.v[i].lo, (long long)(source_offsets[runs.v[i].lo + 1] - source_offsets[runs.v[i].lo]),
                        (long long)(target_offsets[runs.v[i].lo + 1] - target_offsets[runs.v[i].lo]));
    unsigned char* ptr = nullptr;
    std::mutex* call_mu = nullptr;
    ICPX_TRY(scratch_for_device(dev, &ptr, &call_mu));
    std::lock_guard<std::mutex> lk(*call_mu);  // host side only: overlapping calls on other streams still race on the GPU
    BatchScratch sc;
    sc.soff = reinterpret_cast<int64_t*>(ptr + kScrOff);
    sc.doff = sc.soff + n_pairs + 1;
    sc.states = reinterpret_cast<PairState*>(ptr + kScrStates);
    sc.sched = reinterpret_cast<int32_t*>(ptr + kScrSched);
    sc.cur = reinterpret_cast<double*>(ptr + kScrCur);
    sc.seeds = reinterpret_cast<int32_t*>(ptr + kScrSeeds);
    sc.tperm = reinterpret_cast<uint16_t*>(ptr + kScrTperm);
    sc.grids = reinterpret_cast<PairGrid*>(ptr + kScrGrids);
    sc.cstart = reinterpret_cast<uint16_t*>(ptr + kScrCstart);
    sc.hcert = reinterpret_cast<float*>(ptr + kScrHcert);
    sc.kdnode = reinterpret_cast<uint16_t*>(ptr + kScrKd);
    sc.kdpos = sc.kdnode + kNoAllocMaxPoints;
    sc.kd_paused = reinterpret_cast<int32_t*>(ptr + kScrKdWords);
    // everything the kernels expect to start from zero is cleared by them or here: PairState, running
    // clouds, seeds, grids and tables are written before they are read (setup / pre-pass); the scheduler
    // words are cleared by launch_one with hipMemsetAsync
    ICPX_HIP(hipMemcpyAsync(sc.soff, source_offsets, sizeof(int64_t) * (n_pairs + 1), hipMemcpyHostToDevice, stream));
    ICPX_HIP(hipMemcpyAsync(sc.doff, target_offsets, sizeof(int64_t) * (n_pairs + 1), hipMemcpyHostToDevice, stream));
    for (int64_t i = 0; i < runs.count; i++)
        ICPX_TRY(enqueue_fused_run(c, runs.v[i].lo, runs.v[i].hi, i, runs.v[i].geom, sc, *di, stream, t0));
    // the offset arrays are the caller's host memory: finish the copies before returning
    ICPX_HIP(hipStreamSynchronize(stream));
    return ICPX_OK;
}

}  // namespace

extern "C" {

int icpx_version(void) { return ICPX_VERSION; }

const char* icpx_last_error(void) { return last_error_ref().c_str(); }

void icpx_default_params(icpx_params* p) {
    if (!p) return;
    p->overlap_rate = 0.9;  // trimmed.cpp:28
    p->max_distance = 1.0;  // trimmed_3d.cpp:27
    p->nn_mode = ICPX_NN_AUTO;
    p->feature_weight = 0.7;       // feature_aware.cpp:23
    p->symmetric_neighbors = 10;   // feature_aware.cpp:24
    p->flags = 0;
    p->device = -1;
    p->stream = nullptr;
}

int icpx_init(int device) {
    return guarded([&]() -> int {
    int dev = 0;
    ICPX_TRY(resolve_device(device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    ICPX_HIP(hipFree(nullptr));
    unsigned char* scr = nullptr;
    ICPX_TRY(scratch_for_device(dev, &scr, nullptr));  // device address of the no-alloc entry point's scratch
    return ICPX_OK;
    });
}

int icpx_create(int dim, int method, const icpx_params* params, icpx_handle* out) {
    return guarded([&]() -> int {
    if (!out) return fail(ICPX_ERR_INVALID, "null out handle");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(ICPX_ERR_INVALID, "dim must be 2 or 3 (got %d)", dim);
    if (method != ICPX_VANILLA && method != ICPX_TRIMMED && method != ICPX_FEATURE_AWARE)
        return fail(ICPX_ERR_INVALID, "unknown method %d", method);
    if (method == ICPX_FEATURE_AWARE && dim != 2)
        return fail(ICPX_ERR_INVALID, "feature_aware is an ICP2 method (lib/icp/icp.cpp:17-21)");
    icpx_params prm;
    icpx_default_params(&prm);
    if (params) prm = *params;
    if (prm.nn_mode < ICPX_NN_AUTO || prm.nn_mode > ICPX_NN_GRID_F32)
        return fail(ICPX_ERR_INVALID, "unknown nn_mode %d", prm.nn_mode);
    if (prm.flags < 0 || prm.flags > 7) return fail(ICPX_ERR_INVALID, "unknown flags %d", prm.flags);
    int dev = 0;
    ICPX_TRY(resolve_device(prm.device, &dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    icpx_ctx* h = new (std::nothrow) icpx_ctx();
    if (!h) return fail(ICPX_ERR_NOMEM, "out of host memory");
    h->dim = dim;
    h->method = method;
    h->prm = prm;
    h->device = dev;
    h->stream = static_cast<hipStream_t>(prm.stream);
    *out = h;
    return ICPX_OK;
    });
}

int icpx_destroy(icpx_handle h) {
    return guarded([&]() -> int {
    if (!h) return ICPX_OK;
    DeviceGuard g;
    (void)g.enter(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (DevBuf* b : {&h->src, &h->dst, &h->cur, &h->off, &h->guess, &h->m_point, &h->m_pair,
                      &h->m_cost, &h->pair_tmp, &h->seeds, &h->state, &h->out, &h->tperm, &h->grids, &h->cstart, &h->hcert, &h->kd})
        b->release();
    if (h->lg) large_destroy(h->lg);
    if (h->stage) (void)hipHostFree(h->stage);
    delete h;
    return ICPX_OK;
    });
}

int icpx_begin(icpx_handle h, const double* source, int64_t n, const double* target, int64_t m,
               const double* guess) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(h->device, &di));
    ICPX_TRY(begin_common(h, source, n, target, m, guess));
    if (h->large) {
        if (!h->lg) ICPX_TRY(large_create(h->dim, h->method, h->prm, di->clock_to_s, h->stream, &h->lg));
        ICPX_TRY(large_begin(h->lg, source, n, target, m, guess));
        h->n = n;
        h->m = m;
    } else {
        ICPX_TRY(upload_pair(h, source, n, target, m, guess));
        PairBatch pb;
        fill_batch(h, pb, *di);
        pb.do_setup = 1;
        pb.check = 0;
        pb.max_steps = 0;
        ICPX_TRY(launch_pairs(pb, 1, h->geom, h->stream));
    }
    h->begun = true;
    return ICPX_OK;
    });
}

int icpx_iterate(icpx_handle h) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (!h->begun) return fail(ICPX_ERR_STATE, "icpx_iterate before icpx_begin");
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(h->device, &di));
    if (h->large) return large_iterate(h->lg);
    PairBatch pb;
    fill_batch(h, pb, *di);
    pb.do_setup = 0;
    pb.check = 0;
    pb.max_steps = 1;
    return launch_pairs(pb, 1, h->geom, h->stream);
    });
}

int icpx_converge(icpx_handle h, const double* source, int64_t n, const double* target, int64_t m,
                  const double* guess, const icpx_criteria* criteria, icpx_state* out) {
    return guarded([&]() -> int { return converge_impl(h, source, n, target, m, guess, criteria, out); });
}

int icpx_cost(icpx_handle h, double* out) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (!out) return fail(ICPX_ERR_INVALID, "null out");
    if (!h->begun) return fail(ICPX_ERR_STATE, "icpx_cost before icpx_begin");
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    if (h->large) return large_cost(h->lg, out);
    ICPX_HIP(hipMemcpyAsync(out, &h->state.as<PairState>()->cost, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    ICPX_HIP(hipStreamSynchronize(h->stream));
    return ICPX_OK;
    });
}

int icpx_transform(icpx_handle h, double* out) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (!out) return fail(ICPX_ERR_INVALID, "null out");
    if (!h->begun) return fail(ICPX_ERR_STATE, "icpx_transform before icpx_begin");
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    const int XN = (h->dim + 1) * (h->dim + 1);
    if (h->large) return large_transform(h->lg, out);
    ICPX_HIP(hipMemcpyAsync(out, h->state.as<PairState>()->transform, sizeof(double) * XN, hipMemcpyDeviceToHost, h->stream));
    ICPX_HIP(hipStreamSynchronize(h->stream));
    return ICPX_OK;
    });
}

int icpx_sizes(icpx_handle h, int64_t* n, int64_t* m) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (n) *n = h->n;
    if (m) *m = h->m;
    return ICPX_OK;
    });
}

int icpx_matches(icpx_handle h, int64_t* point, int64_t* pair, double* cost) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (!h->begun) return fail(ICPX_ERR_STATE, "icpx_matches before icpx_begin");
    const int64_t n = h->n;
    if (n == 0) return ICPX_OK;
    if (h->dim == 3 && h->large && (h->prm.flags & ICPX_FLAG_FILL_MATCHES)) {
        DeviceGuard g3;
        ICPX_TRY(g3.enter(h->device));
        return large_matches(h->lg, point, pair, cost);  // opt-in: (point, neighbour, distance)
    }
    if (h->dim == 3) {
        // Vanilla3d / Trimmed3d never touch `matches` (icp.h:129 value-initialises them)
        if (point) std::memset(point, 0, sizeof(int64_t) * n);
        if (pair) std::memset(pair, 0, sizeof(int64_t) * n);
        if (cost) std::memset(cost, 0, sizeof(double) * n);
        return ICPX_OK;
    }
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    if (h->large) return large_matches(h->lg, point, pair, cost);
    if (h->method == ICPX_TRIMMED) {
        // the kernels leave trimmed ICP2's matches in point order and marked; sorted by (cost, point) here (trimmed.cpp:55-56)
        int npad = 2;
        while (npad < n) npad <<= 1;
        const size_t lds = 12 * static_cast<size_t>(npad);
        static std::mutex mu;
        static std::unordered_map<int, bool> attr_set;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!attr_set[h->device]) {
                ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_sort_matches_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 2 * kPairMaxN));
                attr_set[h->device] = true;
            }
        }
        hipLaunchKernelGGL(pair_sort_matches_kernel, dim3(1), dim3(1024), lds, h->stream, static_cast<int>(n), npad,
                           h->state.as<PairState>(), h->m_point.as<int64_t>(), h->m_pair.as<int64_t>(), h->m_cost.as<double>(),
                           h->pair_tmp.as<int32_t>());
        ICPX_HIP(hipGetLastError());
    }
    if (point) ICPX_HIP(hipMemcpyAsync(point, h->m_point.p, sizeof(int64_t) * n, hipMemcpyDefault, h->stream));
    if (pair) ICPX_HIP(hipMemcpyAsync(pair, h->m_pair.p, sizeof(int64_t) * n, hipMemcpyDefault, h->stream));
    if (cost) ICPX_HIP(hipMemcpyAsync(cost, h->m_cost.p, sizeof(double) * n, hipMemcpyDefault, h->stream));
    ICPX_HIP(hipStreamSynchronize(h->stream));
    return ICPX_OK;
    });
}

int icpx_debug_neighbors(icpx_handle h, int64_t* indices, double* distances) {
    return guarded([&]() -> int {
    ICPX_TRY(check_handle(h));
    if (!h->begun) return fail(ICPX_ERR_STATE, "icpx_debug_neighbors before icpx_begin");
    if (h->dim != 3) return fail(ICPX_ERR_INVALID, "neighbors exist for the 3D methods only");
    const int64_t n = h->n;
    if (n == 0) return ICPX_OK;
    DeviceGuard g;
    ICPX_TRY(g.enter(h->device));
    if (h->large) return large_neighbors(h->lg, indices, distances);
    if (indices) ICPX_HIP(hipMemcpyAsync(indices, h->m_pair.p, sizeof(int64_t) * n, hipMemcpyDefault, h->stream));
    if (distances) ICPX_HIP(hipMemcpyAsync(distances, h->m_cost.p, sizeof(double) * n, hipMemcpyDefault, h->stream));
    ICPX_HIP(hipStreamSynchronize(h->stream));
    return ICPX_OK;
    });
}

int icpx_converge_batch(int dim, int method, const icpx_params* params, int64_t n_pairs,
                        const double* source, const int64_t* source_offsets,
                        const double* target, const int64_t* target_offsets,
                        const double* guesses, const icpx_criteria* criteria,
                        icpx_state* out_states) {
    return guarded([&]() -> int {
        return converge_batch_impl(dim, method, params, n_pairs, source, source_offsets, target, target_offsets, guesses,
                                   criteria, out_states);
    });
}

int icpx_converge_batch_multi(int dim, int method, const icpx_params* params, int device_count, const int* devices,
                              int64_t n_pairs, const double* source, const int64_t* source_offsets,
                              const double* target, const int64_t* target_offsets, const double* guesses,
                              const icpx_criteria* criteria, icpx_state* out_states) {
    return guarded([&]() -> int {
        return converge_batch_multi_impl(dim, method, params, device_count, devices, n_pairs, source, source_offsets,
                                         target, target_offsets, guesses, criteria, out_states);
    });
}

int icpx_converge_batch_noalloc(int dim, int method, const icpx_params* params, int64_t n_pairs,
                                const double* source, const int64_t* source_offsets,
                                const double* target, const int64_t* target_offsets,
                                const double* guesses, const icpx_criteria* criteria,
                                icpx_state* out_states) {
    return guarded([&]() -> int {
        return converge_batch_noalloc_impl(dim, method, params, n_pairs, source, source_offsets, target, target_offsets,
                                           guesses, criteria, out_states);
    });
}

int icpx_nn_grid_profile(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
                         int device, void* stream_, double* out) {
    return guarded([&]() -> int {
    if (dim != 2 && dim != 3) return fail(ICPX_ERR_INVALID, "dim must be 2 or 3 (got %d)", dim);
    if (n <= 0 || m <= 0 || !queries || !targets || !out) return fail(ICPX_ERR_INVALID, "null or empty input");
    if (n > INT32_MAX / 4 || m > INT32_MAX / 4) return fail(ICPX_ERR_UNSUPPORTED, "cloud too large");
    int dev = 0;
    ICPX_TRY(resolve_device(device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    if (!is_device_ptr(queries) || !is_device_ptr(targets))
        return fail(ICPX_ERR_INVALID, "icpx_nn_grid_profile takes device pointers");
    Workspace& ws = workspace(dev, static_cast<hipStream_t>(stream_));
    std::lock_guard<std::mutex> lk(ws.mu);
    if (!ws.grid) ICPX_TRY(grid_create(&ws.grid));
    grid_set_f32(ws.grid, false);
    return grid_profile(ws.grid, dim, queries, n, targets, m, static_cast<hipStream_t>(stream_), out);
    });
}

#ifdef ICPX_STATS
// development builds only (-DICPX_STATS): read / reset the kernel statistics counters
int icpx_debug_stats(unsigned long long* out, int reset) {
    return guarded([&]() -> int {
    ICPX_HIP(hipDeviceSynchronize());
    unsigned long long acc[32] = {0};
    ICPX_TRY(stats_read_d2_plain(acc, reset));
    ICPX_TRY(stats_read_d2_prune(acc, reset));
    ICPX_TRY(stats_read_d3_plain(acc, reset));
    ICPX_TRY(stats_read_d3_prune(acc, reset));
    if (out) std::memcpy(out, acc, sizeof(acc));
    return ICPX_OK;
    });
}
#endif

int icpx_debug_knob(const char* name, const char* value) {
    return guarded([&]() -> int {
    if (!name) return fail(ICPX_ERR_INVALID, "null knob name");
    if (!set_knob(knobs(), name, value)) return fail(ICPX_ERR_INVALID, "unknown knob %s", name);
    return ICPX_OK;
    });
}

int icpx_debug_kd_tree(int dim, const double* targets, int64_t m, uint32_t* node_out, int device) {
    return guarded([&]() -> int {
    if (dim != 2 && dim != 3) return fail(ICPX_ERR_INVALID, "dim must be 2 or 3 (got %d)", dim);
    if (m <= 0 || !targets || !node_out) return fail(ICPX_ERR_INVALID, "icpx_debug_kd_tree: empty cloud or null pointer");
    int dev = 0;
    ICPX_TRY(resolve_device(device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DevBuf t;
    ICPX_TRY(t.reserve(sizeof(double) * dim * m));
    int rc = ICPX_OK;
    hipError_t e = hipMemcpy(t.p, targets, sizeof(double) * dim * m, hipMemcpyDefault);
    if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "icpx_debug_kd_tree: %s", hipGetErrorString(e));
    if (rc == ICPX_OK) rc = kd_debug_tree(dim, t.as<double>(), m, node_out, nullptr);
    t.release();
    return rc;
    });
}

// icpx_nn (kd = false) and icpx_nn_kdtree (kd = true: the kd-tree's tie rule, `d2` receives Euclidean distances)
static int nn_impl(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
                   int64_t* idx, double* d2, int mode, int device, void* stream_, const bool kd) {
    return guarded([&]() -> int {
    if (dim != 2 && dim != 3) return fail(ICPX_ERR_INVALID, "dim must be 2 or 3 (got %d)", dim);
    if (n < 0 || m < 0) return fail(ICPX_ERR_INVALID, "negative cloud size");
    if (mode < ICPX_NN_AUTO || mode > ICPX_NN_GRID_F32) return fail(ICPX_ERR_INVALID, "unknown nn mode %d", mode);
    if (kd && mode == ICPX_NN_GRID_F32)
        return fail(ICPX_ERR_UNSUPPORTED, "the f32 grid mode gives up index parity: it has no kd-tree tie rule");
    if (n == 0) return ICPX_OK;
    if (!queries || (m > 0 && !targets)) return fail(ICPX_ERR_INVALID, "null queries/targets");
    if (n > INT32_MAX / 4 || m > INT32_MAX / 4) return fail(ICPX_ERR_UNSUPPORTED, "cloud too large");
    int dev = 0;
    ICPX_TRY(resolve_device(device, &dev));
    DeviceGuard g;
    ICPX_TRY(g.enter(dev));
    DeviceInfo* di = nullptr;
    ICPX_TRY(device_info(dev, &di));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Workspace& ws = workspace(dev, stream);
    std::lock_guard<std::mutex> lk(ws.mu);

    const double* d_q = queries;
    const double* d_t = targets;
    if (!is_device_ptr(queries)) {
        ICPX_TRY(ws.q.reserve(sizeof(double) * dim * n));
        ICPX_HIP(hipMemcpyAsync(ws.q.p, queries, sizeof(double) * dim * n, hipMemcpyHostToDevice, stream));
        d_q = ws.q.as<double>();
    }
    if (m > 0 && !is_device_ptr(targets)) {
        ICPX_TRY(ws.t.reserve(sizeof(double) * dim * m));
        ICPX_HIP(hipMemcpyAsync(ws.t.p, targets, sizeof(double) * dim * m, hipMemcpyHostToDevice, stream));
        d_t = ws.t.as<double>();
    }
    const bool idx_dev = is_device_ptr(idx), d2_dev = is_device_ptr(d2);
    int64_t* d_idx = idx;
    double* d_d2 = d2;
    if (idx && !idx_dev) {
        ICPX_TRY(ws.idx.reserve(sizeof(int64_t) * n));
        d_idx = ws.idx.as<int64_t>();
    }
    if (d2 && !d2_dev) {
        ICPX_TRY(ws.d2.reserve(sizeof(double) * n));
        d_d2 = ws.d2.as<double>();
    }
    // kd rule (icpx_kd.h): the searches flag the queries that have a tie partner; they are resolved afterwards
    KdTies ties;
    if (kd && m > 0) {
        if (!d_idx) {  // the resolution corrects indices: it needs them even when the caller does not
            ICPX_TRY(ws.idx.reserve(sizeof(int64_t) * n));
            d_idx = ws.idx.as<int64_t>();
        }
        ICPX_TRY(ws.kd_list.reserve(sizeof(uint32_t) * n + 16));
        ties.cloud = d_t;
        ties.count = ws.kd_list.as<int>();
        ties.list = ws.kd_list.as<uint32_t>() + 4;
        ICPX_HIP(hipMemsetAsync(ties.count, 0, sizeof(int), stream));
    }
    const KdTies* tp = (kd && m > 0) ? &ties : nullptr;

    bool use_grid = mode == ICPX_NN_GRID || mode == ICPX_NN_GRID_F32 ||
                    (mode == ICPX_NN_AUTO && static_cast<double>(n) * static_cast<double>(m) > 512.0 * 1024 * 1024);  // measured crossover of one build + query: ~3e4 x 3e4
    if (m == 0) use_grid = false;
    if (m == 0) {
        hipLaunchKernelGGL(nn_no_targets_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, n, d_idx, d_d2);
        ICPX_HIP(hipGetLastError());
    } else if (mode == ICPX_NN_BVH) {
        if (!ws.bvh) ICPX_TRY(bvh_create(&ws.bvh));
        ICPX_TRY(bvh_nn_oneshot(ws.bvh, dim, d_q, n, d_t, m, d_idx, d_d2, stream, tp));
    } else if (use_grid) {
        if (!ws.grid) ICPX_TRY(grid_create(&ws.grid));
        grid_set_f32(ws.grid, mode == ICPX_NN_GRID_F32);
        if (mode == ICPX_NN_AUTO && !ws.bvh) ICPX_TRY(bvh_create(&ws.bvh));
        ICPX_TRY(grid_nn_oneshot(ws.grid, dim, d_q, n, d_t, m, d_idx, d_d2, stream, mode == ICPX_NN_AUTO ? ws.bvh : nullptr, tp));
    } else {
        // (query tiles) x (target splits) so that the launch fills the chip; f32 filter + exact
        // refinement (nn_tile_kernel).  ICPX_FILTER=0 (development) selects the plain f64 kernel.
        const bool plain = knobs().filter == 0 && !kd;  // (the plain kernel has no tie detection)
        const int per_tile = plain ? kNnThreads * kNnQpt : kNnThreads * kNnTileQpt;
        const int qblocks = static_cast<int>((n + per_tile - 1) / per_tile);
        int splits = (4 * di->cus + qblocks - 1) / qblocks;
        const int64_t min_len = plain ? kNnTile : kNnMinSplit;
        int max_splits = static_cast<int>((m + min_len - 1) / min_len);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        int64_t split_len = (m + splits - 1) / splits;
        if (plain) {
            split_len = (split_len + kNnTile - 1) / kNnTile * kNnTile;
            if (split_len < kNnTile) split_len = kNnTile;
        } else {
            split_len = (split_len + 3) & ~int64_t(3);
            if (split_len < 4) split_len = 4;
            if (split_len > kNnMaxSplit) split_len = kNnMaxSplit;
        }
        splits = m > 0 ? static_cast<int>((m + split_len - 1) / split_len) : 1;
        if (splits > 65535) return fail(ICPX_ERR_UNSUPPORTED, "target cloud too large for the brute-force search; use the grid");
        ICPX_TRY(ws.part_d2.reserve(sizeof(double) * n * splits));
        ICPX_TRY(ws.part_idx.reserve(sizeof(int32_t) * n * splits));
        dim3 grid(qblocks, splits);
        if (plain) {
            if (dim == 2)
                hipLaunchKernelGGL(nn_brute_kernel<2>, grid, dim3(kNnThreads), 0, stream, d_q, n, d_t, m, split_len,
                                   ws.part_d2.as<double>(), ws.part_idx.as<int32_t>());
            else
                hipLaunchKernelGGL(nn_brute_kernel<3>, grid, dim3(kNnThreads), 0, stream, d_q, n, d_t, m, split_len,
                                   ws.part_d2.as<double>(), ws.part_idx.as<int32_t>());
        } else {
            const size_t lds = nn_tile_lds_bytes(dim, static_cast<int>(split_len));
            static std::mutex amu;
            static std::unordered_map<int, bool> aset;
            {
                std::lock_guard<std::mutex> alk(amu);
                if (!aset[dev]) {
                    ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nn_tile_kernel<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(nn_tile_lds_bytes(3, kNnMaxSplit))));
                    ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nn_tile_kernel<3>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(nn_tile_lds_bytes(3, kNnMaxSplit))));
                    aset[dev] = true;
                }
            }
            auto tk = dim == 2 ? (kd ? nn_tile_kernel<2, true> : nn_tile_kernel<2, false>)
                               : (kd ? nn_tile_kernel<3, true> : nn_tile_kernel<3, false>);
            hipLaunchKernelGGL(tk, grid, dim3(kNnThreads), lds, stream, d_q, n, d_t, m, static_cast<int>(split_len),
                               static_cast<int>(split_len), ws.part_d2.as<double>(), ws.part_idx.as<int32_t>());
        }
        ICPX_HIP(hipGetLastError());
        hipLaunchKernelGGL(nn_merge_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, n,
                           splits, ws.part_d2.as<double>(), ws.part_idx.as<int32_t>(), d_idx, d_d2, dim, tp ? tp->cloud : nullptr,
                           tp ? tp->list : nullptr, tp ? tp->count : nullptr);
        ICPX_HIP(hipGetLastError());
    }
    if (tp) {
        ICPX_TRY(kd_resolve_queries(dim, d_q, n, d_t, m, ties, d_idx, stream));
        if (d_d2) {
            hipLaunchKernelGGL(nn_sqrt_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, n, d_d2);
            ICPX_HIP(hipGetLastError());
        }
    }
    if (idx && !idx_dev) ICPX_HIP(hipMemcpyAsync(idx, d_idx, sizeof(int64_t) * n, hipMemcpyDeviceToHost, stream));
    if (d2 && !d2_dev) ICPX_HIP(hipMemcpyAsync(d2, d_d2, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    ICPX_HIP(hipStreamSynchronize(stream));
    return ICPX_OK;
    });
}

int icpx_nn(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
            int64_t* idx, double* d2, int mode, int device, void* stream_) {
    return nn_impl(dim, queries, n, targets, m, idx, d2, mode, device, stream_, false);
}

int icpx_nn_kdtree(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
                   int64_t* idx, double* dist, int mode, int device, void* stream_) {
    return nn_impl(dim, queries, n, targets, m, idx, dist, mode, device, stream_, true);
}

}  // extern "C"
