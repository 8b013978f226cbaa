// vdet_capi.hip -- C-ABI of libvdet_hip.so (see include/vdet_hip.h) and the host-side
// orchestration of the gfx950 kernels.  Built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off is load-bearing: the f32 operation order of the IoU predicate is the spec).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vdet_hip.h"
#include "nms_kernels.hpp"
#include "binsort_kernels.hpp"
#include "small_kernels.hpp"
#include "detnms_kernels.hpp"
#include "fused_kernels.hpp"
#include "adjrows_kernels.hpp"
#include "graphlists_kernels.hpp"
#include "temporal_kernels.hpp"
#include "tubelet_kernels.hpp"
#include "track_kernels.hpp"
#include "batch_kernels.hpp"
#include "gemm_kernels.hpp"

using namespace vdet;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {  // retry exact
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

enum Stage { ST_IOU_BITS = 0, ST_ADJ = 1, ST_SORTK = 2, ST_WALK = 3, ST_TEMPORAL = 4, ST_SORT = 5, ST_IOU_GEN = 6, ST_OTHER = 7,
             ST_TRANSPOSE = 8, ST_TPICK = 9, ST_TLINK = 10, ST_TSUPP = 11, ST_RSPATIAL = 12, ST_RSERIES = 13, ST_SORTFB = 14, ST_TLOOP = 15, ST_COUNT = 16 };

struct Counters {            // one small device block
    // sticky until vdet_sync (or a synchronous graph build) reads and clears them
    int status;
    int eindex;              // latched "IndexError" of the rescoring kernels
    unsigned long long pool_max;   // largest pool_used of the asynchronous builds since the last vdet_sync
    // per graph build (kPerBuildOff .. end): cleared when a build starts
    unsigned int glob_cnt;
    int irregular;           // frames that are not "regular" (frame_flags_kernel), counted per graph build
    unsigned long long pool_used;
};
constexpr size_t kPerBuildOff = 16;

// an asynchronous build is about to clear the per-build counters: keep the largest pool demand of the window
__global__ void fold_pool_kernel(Counters *cnt)
{
    if (cnt->pool_used > cnt->pool_max) cnt->pool_max = cnt->pool_used;
}

struct NmsPlan {
    std::vector<GroupDesc> groups;   // bits_off is batch-local
    std::vector<TileDesc> tiles;     // ordered by batch
    std::vector<TilePair> pairs;     // upper-triangle 256x256 tile pairs, ordered by batch
    std::vector<std::pair<int, int>> batch_tiles;  // [t0, t1) per batch
    std::vector<std::pair<int, int>> batch_pairs;  // [p0, p1) per batch
    std::vector<ListItem> ditems;    // direct lists (graph_lists_kernel): one work item per (row tile, width quartile)
    std::vector<std::pair<int, int>> batch_ditems;
    size_t bits_words_max = 0;       // largest batch
    int64_t ntot = 0;
    int nmax = 0;
};

}  // namespace

struct vdet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;
    Counters *d_cnt = nullptr;
    int latched = 0;              // first failure latched by async entry points
    size_t bits_budget = (size_t)1 << 30;
    size_t max_lds = 160 * 1024;
    int n_cu = 256;
    // scratch
    DevBuf boxes, scores, keys, excl, frames, groups, tiles, bits, rowz, rowmeta, groupz, adj, comp, origidx,
        out64, trk_frames, trk_boxes, b1, b2, iou_out, order, ncand, keepidx, keepcnt, gflags, pairs, tkeys, tstate, visited, heads, xkeys, xord, wmeta, wmeta16, reachtab, rowperm, qreach, ditems, striptot, stripoff, xncand, xbox, xbox16, xord16, xcum, xinfo, tmp[8];
    // timing
    bool timing = false;
    bool timing_accumulate = false;   // vdet_set_timing(ctx, 2): keep events across calls until read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_stage;
    size_t ev_used = 0;
    float last_ms[ST_COUNT] = {0};
    int last_launches[ST_COUNT] = {0};
    bool sort_attr_set = false;
    // opt-in reuse of the per-video preparation (graph + sorted lists) between d_* calls
    bool cache_enabled = false;
    struct PrepKey { const void *boxes = nullptr, *scores = nullptr; int64_t F = 0, B = 0, C = 0; float t32 = 0; int layout = -1, use_thr = 0; float thr = 0; int topk = 0; } prep;
    bool graph_valid = false, lists_valid = false;
    // class-major sort keys left in c->tkeys by vdet_volume_pass (reused like the other prep, cache on)
    struct KeySrc { const void *scores = nullptr; int64_t F = 0, B = 0, C = 0; int use_thr = 0; float thr = 0; } keysrc;
    bool keys_valid = false;
    bool vpass_attr_set = false;
    bool index_valid = false; const void *index_boxes = nullptr; int64_t index_F = 0, index_B = 0;
    bool all_regular = false;     // last graph build: every frame regular
    bool wave_transpose = false;  // wave_transpose64 verified on this device (vdet_create); VDET_WAVE_TRANSPOSE=0 disables
    // Diagnostic switches (read once at vdet_create; DESIGN.md "Knobs").  Each forces a FALLBACK path the library takes anyway
    // on some inputs or devices, so that the tests can run it on every input: no tuning variants live here.
    bool no_lazy = false;         // VDET_NO_LAZY=1: eager track_det_nms of every crossed list (the irregular-frame path)
    bool no_index = false;        // VDET_NO_INDEX=1: no x-sorted proposal index (the path of frames too large for it)
    bool force_general = false;   // VDET_FORCE_GENERAL=1: the general predicate kernel K1 on every frame (the irregular-frame path)
    bool atomic_rank = false;     // LDS returning atomics serve same-address lanes in lane order (probed; VDET_ATOMIC_RANK=0: ballot match)
    bool small_lists = true;      // VDET_SMALL_LISTS=0: frames of <= 384 boxes through the large-list sort and walk too (small_kernels.hpp)
    bool adj_rows = true;         // VDET_ADJ_ROWS=0: adj_build_kernel (a lane per row) also for the regular frames of large volumes
    bool direct_lists = true;     // VDET_DIRECT_LISTS=0: regular frames through the bit matrix + adj_rows_kernel (also taken, for good,
                                  // once a row had more neighbours than a direct slot holds)
    uint32_t direct_cap = 384;    // entries per row slot of the direct lists (a multiple of 8)
    int64_t direct_ntot = 0;      // rows / fixed part of the pool of the last direct build (the failure path sizes the pool from them)
    unsigned long long direct_fixed = 0;
    bool no_fused = false;        // VDET_NO_FUSED=1: the host-buffer calls of <= 640 rows through the general kernel chain too (the path of larger inputs)
    bool binsort = true;          // VDET_BINSORT=0: the LSD radix kernel (the fallback of tied / thresholded columns) for every column
    const std::vector<GroupDesc> *host_groups = nullptr;   // group table of the call in flight (mode 2)
    bool sym_built = false;       // the last graph build ran K0 + frame index + K1s (regular-frame fast path)
    // volume geometry whose group / tile / pair tables are resident in c->groups / c->tiles / c->pairs
    // (uploaded once per geometry: the host copy lives here so no call has to wait for the copies)
    NmsPlan vplan;
    bool vplan_valid = false;     // ... and the device tables still hold it
    int64_t vplan_F = 0, vplan_B = 0;
    size_t vplan_budget = 0;
    // asynchronous video step (vdet_set_async): no host synchronisation inside the d_* entry points
    long long n_host_syncs = 0;   // hipStreamSynchronize calls made by this context so far
    bool async_enabled = false;
    unsigned long long pool_hint = 0;   // adjacency entries used by the largest graph built so far
    bool topk_attr_set = false;
    float gt32 = 0.f;             // threshold of the last graph build (the packed walk's in-group test)
    bool wmeta16_built = false;   // ... and adj_rows_kernel the 16-byte form of the records of integer frames
    bool wmeta_built = false;     // ... which also wrote the packed walk's records (WalkMeta) of the regular frames
    bool last_sort_binned = false;   // the last per-(frame, class) sort went through binsort_kernel (vdet_query 9)
    DevBuf vidtab;                // batched videos: {first frame, frames} per video
    std::vector<VidDesc> h_vids, h_vids_stage;   // the resident table (empty: none) / the source of the copy in flight
    DevBuf segtab;                // batched videos: per-frame {first, one past last} frame of its video
    std::vector<int2> h_seg;
    std::vector<int64_t> h_seg_off;   // the offsets h_seg / segtab were built from
    DevBuf sortctl;               // binsort_kernel's work counter + the list of problems it handed to the LSD kernel
    DevBuf nover;                 // vdet_det_nms_volume: candidates per list before the topk cut
    DevBuf ordncand;              // vdet_nms_volume_ordered: the caller's counts after check_order_kernel
    DevBuf linkmemo, linkstats, linkwarm, linkorder, linkchains, linknodes, tracknode, rtodo;
    // which proposal every row of the last tracking call's tracks is (written by the link kernels; vdet_rescore_tracks
    // then finds a tubelet box's overlapping detections among that proposal's graph neighbours)
    struct NodeKey { const void *tracks = nullptr, *boxes = nullptr; int64_t F = 0, B = 0, C = 0; int T = 0; double nms_thres = 0; } nodekey;
    bool nodes_valid = false;
    // single-launch drop-in calls (vdet_nms_f32 / vdet_track_det_nms_f32 on <= kFusedMax rows): host-mapped staging
    void *fused_in = nullptr, *fused_out = nullptr;          // host addresses
    void *fused_in_dev = nullptr, *fused_out_dev = nullptr;  // ... and what the device calls them
    void *fbatch_in = nullptr, *fbatch_out = nullptr, *fbatch_in_dev = nullptr, *fbatch_out_dev = nullptr;   // vdet_track_det_nms_batch (grown on demand)
    size_t fbatch_in_cap = 0, fbatch_out_cap = 0;
    bool fbatch_attr = false;
    size_t dyn_lds_max = 0;
};

namespace {

// every host wait of the library goes through here (counted: vdet_query(ctx, 8); the asynchronous video step is
// tested to add none)
hipError_t host_sync(vdet_ctx *c)
{
    ++c->n_host_syncs;
    return hipStreamSynchronize(c->stream);
}

int fail(vdet_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail((c), VDET_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct StageTimer {
    vdet_ctx *c;
    size_t slot = (size_t)-1;
    StageTimer(vdet_ctx *ctx, int stage) : c(ctx)
    {
        if (!c->timing) return;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            c->ev_pool.push_back({a, b});
            c->ev_stage.push_back(0);
        }
        slot = c->ev_used++;
        c->ev_stage[slot] = stage;
        (void)hipEventRecord(c->ev_pool[slot].first, c->stream);
    }
    ~StageTimer()
    {
        if (slot != (size_t)-1) (void)hipEventRecord(c->ev_pool[slot].second, c->stream);
    }
};

void timing_reset(vdet_ctx *c) { if (!c->timing_accumulate) c->ev_used = 0; }

// smallest float32 f with (double)f >= thresh: "(double)ovr_f32 >= thresh" <=> "ovr_f32 >= f"
float thresh_to_f32(double thresh)
{
    if (thresh != thresh) return NAN;
    float f = (float)thresh;  // round to nearest
    if ((double)f < thresh) f = nextafterf(f, INFINITY);
    return f;
}

uint32_t pow2ceil(uint32_t x)
{
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

int translate_status(vdet_ctx *c, int st)
{
    if (st & kStPoolAsync) return fail(c, VDET_EAGAIN, "adjacency pool overflow in an asynchronous graph build: run the calls again");
    if (st & kStBadOrder) return fail(c, VDET_EINVAL, "a caller-supplied candidate list holds a count or a box index out of range");
    if (st & kStDivZero) return fail(c, VDET_EDIVZERO, "float division (zero union)");
    if (st & kStCap) return fail(c, VDET_ECAP, "more survivors than the output capacity");
    if (st & kStPool) return fail(c, VDET_EHIP, "internal: adjacency pool overflow");
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
// NMS orchestration
// ---------------------------------------------------------------------------------------------
int make_plan(vdet_ctx *c, NmsPlan &pl)
{
    const size_t budget_words = std::max<size_t>(c->bits_budget / 8, 1);
    size_t cur = 0;
    int t0 = 0, p0 = 0, d0 = 0;
    pl.nmax = 0;
    pl.ntot = 0;
    for (size_t g = 0; g < pl.groups.size(); ++g) {
        GroupDesc &gd = pl.groups[g];
        pl.ntot = std::max<int64_t>(pl.ntot, (int64_t)gd.box_off + gd.nbox);
        pl.nmax = std::max(pl.nmax, gd.nbox);
        if (gd.nbox < 2) { gd.bits_off = 0; continue; }       // singletons: no graph
        const size_t words = (size_t)bit_words_of_group(gd.nbox);
        if (cur && cur + words > budget_words) {
            pl.batch_tiles.push_back({t0, (int)pl.tiles.size()});
            pl.batch_pairs.push_back({p0, (int)pl.pairs.size()});
            pl.batch_ditems.push_back({d0, (int)pl.ditems.size()});
            t0 = (int)pl.tiles.size();
            p0 = (int)pl.pairs.size();
            d0 = (int)pl.ditems.size();
            pl.bits_words_max = std::max(pl.bits_words_max, cur);
            cur = 0;
        }
        gd.bits_off = (int64_t)cur;
        cur += words;
        const int nrt = (gd.nbox + kRowsPerTile - 1) / kRowsPerTile;
        for (int rt = 0; rt < nrt; ++rt) pl.tiles.push_back({(int32_t)g, rt});
        for (int rt = 0; rt < nrt; ++rt)
            for (int ct = rt; ct < nrt; ++ct) pl.pairs.push_back({(int32_t)g, (int16_t)rt, (int16_t)ct});
        // (the widest quartile of every tile first: its items run longest)
        for (int q = 3; q >= 0; --q)
            for (int rt = 0; rt < nrt; ++rt) pl.ditems.push_back({(int32_t)g, (int32_t)(4 * rt + q)});
    }
    if ((int)pl.tiles.size() > t0) {
        pl.batch_tiles.push_back({t0, (int)pl.tiles.size()});
        pl.batch_pairs.push_back({p0, (int)pl.pairs.size()});
        pl.batch_ditems.push_back({d0, (int)pl.ditems.size()});
    }
    pl.bits_words_max = std::max(pl.bits_words_max, cur);
    (void)c;
    return VDET_OK;
}

int sort_groups_by_keys(vdet_ctx *c, int G, int nmax, int64_t ntot);

// x1-sorted proposal index of every group (frame), see nms_kernels.hpp.  Needs c->groups uploaded.
// Used by the fast K1 (rank space + tile skipping) and by the LINK kernels (IoU windows).
int build_frame_index(vdet_ctx *c, const float4 *d_boxes, int64_t ntot, int64_t G, int nmax)
{
    if (c->cache_enabled && c->index_valid && c->index_boxes == d_boxes && c->index_F == G && c->index_B == ntot) return VDET_OK;
    c->index_valid = false;
    HIPCHK(c, c->xkeys.reserve((size_t)ntot * 4));
    HIPCHK(c, c->xord.reserve((size_t)ntot * 2));
    HIPCHK(c, c->xncand.reserve((size_t)G * 4));
    HIPCHK(c, c->xbox.reserve((size_t)ntot * 16));
    // compact copy for frames of integer pixel coordinates (kFlagU16): every group at an even position
    HIPCHK(c, c->xbox16.reserve((size_t)(ntot + G + 4) * 8));
    HIPCHK(c, c->xord16.reserve((size_t)(ntot + G + 4) * 2));
    HIPCHK(c, c->xcum.reserve((size_t)G * 257 * 4));
    HIPCHK(c, c->xinfo.reserve((size_t)G * 16));
    StageTimer tm(c, ST_OTHER);
    hipLaunchKernelGGL(xkey_kernel, dim3((unsigned)((ntot + 255) / 256)), dim3(256), 0, c->stream, d_boxes,
                       c->xkeys.as<uint32_t>(), ntot);
    int rc = sort_groups_by_keys(c, (int)G, nmax, ntot);
    if (rc) return rc;
    hipLaunchKernelGGL(frame_index_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, d_boxes, c->groups.as<GroupDesc>(),
                       c->xord.as<uint16_t>(), c->xbox.as<float4>(), c->xcum.as<uint32_t>(), c->xinfo.as<float>(),
                       c->gflags.as<uint32_t>(), c->xbox16.as<uint2>(), c->xord16.as<uint16_t>());
    HIPCHK(c, hipGetLastError());
    c->index_valid = true; c->index_boxes = d_boxes; c->index_F = G; c->index_B = ntot;
    return VDET_OK;
}

FrameIndex frame_index_of(vdet_ctx *c)
{
    return FrameIndex{c->xbox.as<float4>(), c->xord.as<uint16_t>(), c->xcum.as<uint32_t>(), c->xinfo.as<float>(),
                      c->xbox16.as<uint2>(), c->xord16.as<uint16_t>(), 0};
}

// The plan of a regular volume (one group of B boxes per frame).  Built on the host once per geometry
// and kept in the context: the host tables then outlive every asynchronous copy made from them, and a
// video with the geometry of the previous one finds the device tables already in place.
NmsPlan &volume_plan(vdet_ctx *c, int64_t F, int64_t B)
{
    if (c->vplan_F == F && c->vplan_B == B && c->vplan_budget == c->bits_budget && !c->vplan.groups.empty()) return c->vplan;
    (void)host_sync(c);     // (rare: geometry change) copies from the old tables may be in flight
    // the resident group / tile / pair tables are about to describe another geometry: whatever the cache holds (graph, sorted
    // lists, x-index, recorded track nodes) was built -- and is decoded -- with the old tables
    c->graph_valid = c->lists_valid = c->index_valid = c->nodes_valid = false;
    c->vplan = NmsPlan();
    c->vplan.groups.resize((size_t)F);
    for (int64_t f = 0; f < F; ++f) c->vplan.groups[(size_t)f] = {(int32_t)(f * B), (int32_t)B, 0};
    make_plan(c, c->vplan);
    c->vplan_F = F; c->vplan_B = B; c->vplan_budget = c->bits_budget;
    c->vplan_valid = false;
    return c->vplan;
}

// host-buffer entry points: their group table dies with the call
struct HostGroupsGuard {
    vdet_ctx *c;
    explicit HostGroupsGuard(vdet_ctx *ctx) : c(ctx) {}
    ~HostGroupsGuard() { c->host_groups = nullptr; }
};


// K0 + index + K1(s) + K2 for every batch.
//   volume == false (host-buffer entry points, plan owned by the caller): synchronous, retries once
//     with a larger adjacency pool, clears the device status block (after picking up a failure
//     latched by earlier asynchronous calls).
//   volume == true  (pl == c->vplan): the tables are uploaded once per geometry.  With
//     vdet_set_async and a pool size known from an earlier build there is NO host synchronisation:
//     the status words stay sticky until vdet_sync, which also reports a pool overflow.
int build_graph(vdet_ctx *c, const float4 *d_boxes, NmsPlan &pl, float t32, double thresh, bool volume)
{
    const float one_minus_t = (float)std::max(0.0, 1.0 - (thresh == thresh ? thresh : 0.0));
    const size_t G = pl.groups.size();
    c->host_groups = &pl.groups;
    c->sym_built = false;
    c->gt32 = t32;
    c->wmeta_built = false;
    c->wmeta16_built = false;
    c->nodes_valid = false;      // the adjacency lists the recorded track nodes point into are rewritten
    HIPCHK(c, c->groups.reserve(G * sizeof(GroupDesc)));
    HIPCHK(c, c->tiles.reserve(std::max<size_t>(pl.tiles.size(), 1) * sizeof(TileDesc)));
    HIPCHK(c, c->bits.reserve(std::max<size_t>(pl.bits_words_max, 1) * 8));
    HIPCHK(c, c->rowz.reserve((size_t)pl.ntot * 4));
    HIPCHK(c, c->rowmeta.reserve((size_t)pl.ntot * 8));
    HIPCHK(c, c->groupz.reserve(G * 4));
    HIPCHK(c, c->gflags.reserve(G * 4));
    HIPCHK(c, c->pairs.reserve(std::max<size_t>(pl.pairs.size(), 1) * sizeof(TilePair)));
    HIPCHK(c, c->ditems.reserve(std::max<size_t>(pl.ditems.size(), 1) * sizeof(ListItem)));
    if (!(volume && c->vplan_valid)) {
        if (!pl.pairs.empty())
            HIPCHK(c, hipMemcpyAsync(c->pairs.p, pl.pairs.data(), pl.pairs.size() * sizeof(TilePair),
                                     hipMemcpyHostToDevice, c->stream));
        if (!pl.ditems.empty())
            HIPCHK(c, hipMemcpyAsync(c->ditems.p, pl.ditems.data(), pl.ditems.size() * sizeof(ListItem),
                                     hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->groups.p, pl.groups.data(), G * sizeof(GroupDesc), hipMemcpyHostToDevice, c->stream));
        if (!pl.tiles.empty())
            HIPCHK(c, hipMemcpyAsync(c->tiles.p, pl.tiles.data(), pl.tiles.size() * sizeof(TileDesc),
                                     hipMemcpyHostToDevice, c->stream));
        c->vplan_valid = volume;         // (a host-call plan overwrites the volume tables)
    }
    // direct lists (iou_bits_sym_kernel<true, true>): regular frames of large volumes get a fixed slot per row at the start of
    // the pool, the dynamic part (irregular frames) follows
    auto direct_now = [&]() {
        return c->direct_lists && c->adj_rows && c->wave_transpose && pl.nmax > 384 && t32 > 1e-30f && t32 < INFINITY && !c->force_general &&
               (size_t)8 * pl.nmax + 24 * 1024 <= c->max_lds && ((unsigned long long)pl.ntot + 1ull) * c->direct_cap * 2ull + 512ull <= 0xFFFFFFFFull;     // (byte offsets of the entries in 32 bits)
    };
    // (one slot more than rows: where the entries of a column that outgrew its slot are dumped)
    auto fixed_pool = [&]() { return direct_now() ? (unsigned long long)pl.ntot * c->direct_cap + std::max<unsigned long long>(c->direct_cap, 128) : 0ull; };
    unsigned long long min_pool = fixed_pool() + (unsigned long long)pl.ntot * (direct_now() ? 1 : 32);
    const bool async = volume && c->async_enabled && c->pool_hint > 0;
    if (async) {
        const unsigned long long dyn_hint = c->pool_hint > fixed_pool() ? c->pool_hint - fixed_pool() : 0ull;
        const unsigned long long want = std::max(min_pool, fixed_pool() + dyn_hint + dyn_hint / 2);
        if (c->adj.cap < want * 2) HIPCHK(c, c->adj.reserve((size_t)want * 2 + 4096));
    } else {
        // the caller's host vectors must outlive the async copies; also pick up a failure latched by an
        // earlier asynchronous call before the status word is cleared below
        Counters h0;
        HIPCHK(c, hipMemcpyAsync(&h0, c->d_cnt, sizeof h0, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, host_sync(c));
        if (h0.status && !c->latched) c->latched = translate_status(c, h0.status);
        if (h0.eindex && !c->latched) c->latched = fail(c, VDET_EINDEX, "list index out of range");
        if (c->adj.cap < min_pool * 2) HIPCHK(c, c->adj.reserve((size_t)min_pool * 2 + 4096));
    }

    for (int attempt = 0; attempt < 3; ++attempt) {
        const bool direct = direct_now();
        min_pool = fixed_pool() + (unsigned long long)pl.ntot * (direct ? 1 : 32);
        if (!async && c->adj.cap < min_pool * 2) HIPCHK(c, c->adj.reserve((size_t)min_pool * 2 + 4096));
        HIPCHK(c, hipMemsetAsync(c->rowz.p, 0, (size_t)pl.ntot * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(c->rowmeta.p, 0, (size_t)pl.ntot * 8, c->stream));
        HIPCHK(c, hipMemsetAsync(c->groupz.p, 0, G * 4, c->stream));
        if (async) {
            hipLaunchKernelGGL(fold_pool_kernel, dim3(1), dim3(1), 0, c->stream, c->d_cnt);
            HIPCHK(c, hipMemsetAsync((char *)c->d_cnt + kPerBuildOff, 0, sizeof(Counters) - kPerBuildOff, c->stream));
        }
        else HIPCHK(c, hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
        c->direct_ntot = direct ? pl.ntot : 0;
        c->direct_fixed = fixed_pool();
        if (direct) hipLaunchKernelGGL(pool_start_kernel, dim3(1), dim3(1), 0, c->stream, &c->d_cnt->pool_used, fixed_pool());
        const int pool_bits = async ? (kStPool | kStPoolAsync) : kStPool;
        const unsigned long long pool_cap = (c->adj.cap - 2048) / 2;     // (the packed walk reads up to 256 B past a list)
        // fast symmetric kernel for regular frames needs 0 < t32 < inf (exact divide-free test)
        // ... and the x1 index, whose sort must fit the LDS (8 B per box + tables)
        const bool use_sym = t32 > 1e-30f && t32 < INFINITY && !c->force_general &&
                             (size_t)8 * pl.nmax + 24 * 1024 <= c->max_lds;
        c->sym_built = use_sym;
        c->wmeta_built = use_sym;
        if (c->wmeta_built) HIPCHK(c, c->wmeta.reserve((size_t)pl.ntot * sizeof(WalkMeta)));
        if (c->wmeta_built && c->adj_rows && pl.nmax > 384) HIPCHK(c, c->wmeta16.reserve((size_t)pl.ntot * sizeof(uint4)));
        if (use_sym) {
            {
                StageTimer tm(c, ST_OTHER);
                hipLaunchKernelGGL(frame_flags_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, d_boxes,
                                   c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(), &c->d_cnt->irregular);
            }
            // (the index cache is keyed on the boxes pointer: never valid for the context's own scratch)
            if (!volume) c->index_valid = false;
            const int rci = build_frame_index(c, d_boxes, pl.ntot, (int64_t)G, pl.nmax);
            if (rci) return rci;
            // which 64 x 64 blocks of the predicate matrix can hold a set bit (K1s writes, K2 reads only those)
            HIPCHK(c, c->reachtab.reserve((size_t)(pl.ntot / 64 + (int64_t)G + 2) * sizeof(float2)));
            {
                StageTimer tm(c, ST_OTHER);
                hipLaunchKernelGGL(reach_table_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, c->xbox.as<float4>(),
                                   c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(), one_minus_t, c->reachtab.as<float2>());
                if (direct) {      // the tiles' rows by width: one quartile per wave of the predicate kernel
                    HIPCHK(c, c->rowperm.reserve((size_t)pl.ntot * 2 + 1024));
                    HIPCHK(c, c->qreach.reserve((size_t)(pl.ntot / 256 + (int64_t)G + 2) * 4 * sizeof(float)));
                    hipLaunchKernelGGL(row_classes_kernel, dim3((unsigned)G, (unsigned)((pl.nmax + 255) / 256)), dim3(256), 0, c->stream,
                                       c->xbox.as<float4>(), c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(), one_minus_t,
                                       c->rowperm.as<uint16_t>(), c->qreach.as<float>());
                }
            }
        } else {
            c->index_valid = false;      // gflags / the x-index describe some earlier boxes
        }
        if (direct && !pl.ditems.empty()) {
            // ONE launch for every regular frame of the plan (no bit matrix: nothing ties the launch to the batches below): a
            // work item is a wave that runs 90-250 us, and four launches of two rounds of items each were a quarter tail
            StageTimer tm(c, ST_IOU_BITS);
            GraphListsParams gp;
            gp.xbox = c->xbox.as<float4>(); gp.xord = c->xord.as<uint16_t>(); gp.groups = c->groups.as<GroupDesc>();
            gp.group_flags = c->gflags.as<uint32_t>(); gp.items = c->ditems.as<ListItem>(); gp.nitems = (int)pl.ditems.size();
            gp.t32 = t32; gp.one_minus_t = one_minus_t; gp.row_deg = c->rowz.as<uint32_t>(); gp.reach_table = c->reachtab.as<float2>();
            gp.rowperm = c->rowperm.as<uint16_t>(); gp.qreach = c->qreach.as<float>(); gp.adj = c->adj.as<uint16_t>();
            gp.slot_cap = c->direct_cap; gp.status = &c->d_cnt->status; gp.over_bits = pool_bits | kStDirect;
            gp.trash_off = (uint32_t)((unsigned long long)pl.ntot * c->direct_cap * 2ull);      // (the spare slot behind the rows')
            hipLaunchKernelGGL(graph_lists_kernel, dim3((gp.nitems + 7) & ~7), dim3(64), 0, c->stream, gp);
        }
        for (size_t bi = 0; bi < pl.batch_tiles.size(); ++bi) {
            const auto bt = pl.batch_tiles[bi];
            const auto bp = pl.batch_pairs[bi];
            const int nt = bt.second - bt.first;
            if (nt <= 0) continue;
            uint64_t *bits_b = c->bits.as<uint64_t>();
            if (use_sym && !direct && bp.second > bp.first) {       // (the direct lists: one launch for all frames, above)
                StageTimer tm(c, ST_IOU_BITS);
                if (c->wave_transpose)
                    hipLaunchKernelGGL(iou_bits_sym_kernel<true>, dim3(bp.second - bp.first), dim3(256), 0, c->stream,
                                       c->xbox.as<float4>(), c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(),
                                       c->pairs.as<TilePair>() + bp.first, t32, one_minus_t, bits_b, c->rowz.as<uint32_t>(),
                                       c->reachtab.as<float2>());
                else
                    hipLaunchKernelGGL(iou_bits_sym_kernel<false>, dim3(bp.second - bp.first), dim3(256), 0, c->stream,
                                       c->xbox.as<float4>(), c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(),
                                       c->pairs.as<TilePair>() + bp.first, t32, one_minus_t, bits_b, c->rowz.as<uint32_t>(),
                                       c->reachtab.as<float2>());
            }
            // enough column splits to fill the chip when there are few row tiles
            int splits = 1;
            if (nt < 4 * c->n_cu) {
                int wmax = 1;
                for (int t = bt.first; t < bt.second; ++t)
                    wmax = std::max(wmax, (pl.groups[pl.tiles[t].group].nbox + 63) / 64);
                splits = std::min((4 * c->n_cu + nt - 1) / nt, (wmax + 3) / 4);
                splits = std::max(1, std::min(splits, 65535));
            }
            {
                StageTimer tm(c, ST_IOU_GEN);
                hipLaunchKernelGGL(iou_bits_kernel, dim3(nt, splits), dim3(256), 0, c->stream, d_boxes,
                                   c->groups.as<GroupDesc>(), c->tiles.as<TileDesc>() + bt.first, t32,
                                   bits_b, c->rowz.as<uint32_t>(), c->groupz.as<uint32_t>(),
                                   use_sym ? c->gflags.as<uint32_t>() : (const uint32_t *)nullptr);
            }
            {
                StageTimer tm(c, ST_ADJ);
                const bool k2_tile = pl.nmax <= 384;      // small frames: one block per tile
                const bool rows_path = use_sym && c->adj_rows && !k2_tile;
                hipLaunchKernelGGL(k2_tile ? adj_build_kernel<kRowsPerTile> : adj_build_kernel<kAdjRows>, dim3(k2_tile ? nt : 2 * nt),
                                   dim3(k2_tile ? kRowsPerTile : kAdjRows), 0, c->stream, d_boxes,
                                   c->groups.as<GroupDesc>(), c->tiles.as<TileDesc>() + bt.first,
                                   bits_b, c->rowz.as<uint32_t>(), c->rowmeta.as<uint2>(),
                                   c->adj.as<uint16_t>(), &c->d_cnt->pool_used, pool_cap,
                                   &c->d_cnt->status, use_sym ? c->gflags.as<uint32_t>() : (const uint32_t *)nullptr,
                                   use_sym ? frame_index_of(c) : FrameIndex{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}, one_minus_t,
                                   async ? (kStPool | kStPoolAsync) : kStPool,
                                   c->wmeta_built ? c->wmeta.as<WalkMeta>() : (WalkMeta *)nullptr,
                                   use_sym ? c->reachtab.as<float2>() : (const float2 *)nullptr, rows_path ? 1 : 0);
                // regular groups of large frames: one workgroup per 64-row strip, half a wave per row (adjrows_kernels.hpp)
                if (rows_path && direct) {
                    // (one launch for all tiles, behind the first batch: the lists of every regular frame are complete by then)
                    if (bi == 0)
                    hipLaunchKernelGGL(adj_finish_kernel, dim3((unsigned)pl.tiles.size()), dim3(256), 0, c->stream, c->groups.as<GroupDesc>(),
                                       c->tiles.as<TileDesc>(), c->rowz.as<uint32_t>(), c->rowmeta.as<uint2>(),
                                       c->adj.as<uint16_t>(), c->direct_cap, &c->d_cnt->status, c->gflags.as<uint32_t>(),
                                       c->xbox.as<float4>(), c->xord.as<uint16_t>(), pool_bits | kStDirect,
                                       c->wmeta_built ? c->wmeta.as<WalkMeta>() : (WalkMeta *)nullptr,
                                       c->wmeta_built ? c->wmeta16.as<uint4>() : (uint4 *)nullptr);
                    c->wmeta16_built = c->wmeta_built;
                } else if (rows_path) {
                    // every strip's slab offset first (two small launches instead of an atomic per strip on one address)
                    HIPCHK(c, c->striptot.reserve((size_t)4 * nt * 4));
                    HIPCHK(c, c->stripoff.reserve((size_t)4 * nt * 8));
                    hipLaunchKernelGGL(strip_totals_kernel, dim3(nt), dim3(256), 0, c->stream, c->groups.as<GroupDesc>(),
                                       c->tiles.as<TileDesc>() + bt.first, 4 * nt, c->rowz.as<uint32_t>(), c->gflags.as<uint32_t>(),
                                       c->striptot.as<uint32_t>());
                    hipLaunchKernelGGL(strip_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->striptot.as<uint32_t>(), 4 * nt,
                                       c->stripoff.as<unsigned long long>(), &c->d_cnt->pool_used);
                    hipLaunchKernelGGL(adj_rows_kernel, dim3(4 * nt), dim3(256), 0, c->stream, c->groups.as<GroupDesc>(),
                                       c->tiles.as<TileDesc>() + bt.first, bits_b, c->rowz.as<uint32_t>(), c->rowmeta.as<uint2>(),
                                       c->adj.as<uint16_t>(), c->stripoff.as<unsigned long long>(), pool_cap, &c->d_cnt->status, c->gflags.as<uint32_t>(),
                                       c->xbox.as<float4>(), c->xord.as<uint16_t>(), async ? (kStPool | kStPoolAsync) : kStPool,
                                       c->wmeta_built ? c->wmeta.as<WalkMeta>() : (WalkMeta *)nullptr, c->reachtab.as<float2>(),
                                       c->wmeta_built ? c->wmeta16.as<uint4>() : (uint4 *)nullptr);
                    c->wmeta16_built = c->wmeta_built;
                }
            }
        }
        HIPCHK(c, hipGetLastError());
        if (async) {
            c->all_regular = false;      // not known on the host: the tracking loop asks the device (n_irregular)
            return VDET_OK;
        }
        Counters h;
        HIPCHK(c, hipMemcpyAsync(&h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, host_sync(c));
        c->all_regular = use_sym && h.irregular == 0;
        if (!(h.status & kStPool)) {
            c->pool_hint = std::max(c->pool_hint, h.pool_used);
            return VDET_OK;
        }
        if (h.status & kStDirect) {     // a row with more neighbours than a direct slot holds: through the bit matrix from now on
            c->direct_lists = false;
            continue;
        }
        if (h.pool_used > 0xFFFFFFFFull) return fail(c, VDET_ENOMEM, "suppression graph has more than 2^32 edges");
        if (attempt == 2) break;
        HIPCHK(c, c->adj.reserve((size_t)h.pool_used * 2 + 4096));
    }
    return fail(c, VDET_EHIP, "internal: adjacency pool overflow after regrow");
}

// K3 + K4 for P problems.  d_order / d_ncand / keep buffers are caller-provided device pointers.
struct SortWalkArgs {
    bool sort_only = false;       // tracking: only the per-problem lists are wanted
    bool walk_only = false;       // the lists of a previous call are still valid
    uint16_t *order_out = nullptr;   // default: ctx scratch (c->order / c->ncand)
    int32_t *ncand_out = nullptr;
    const uint16_t *order_in = nullptr;   // walk_only: the caller's own lists instead of the context's
    const int32_t *ncand_in = nullptr;
    int mode, P, B, C;
    const float *scores;
    const uint32_t *keys;
    const uint8_t *excl;
    int use_thr;
    float thr;
    int topk = 0;
    int32_t *nover_out = nullptr; // candidates of every list before the topk cut
    int32_t *keep_idx;
    int32_t *keep_cnt;
    int64_t cap;
};

int sort_comp_desc(vdet_ctx *c, uint32_t n);

int launch_sort_walk(vdet_ctx *c, const SortWalkArgs &a, int nmax, int64_t order_elems)
{
    if (a.P <= 0) return VDET_OK;
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    if (!a.order_out) {
        HIPCHK(c, c->order.reserve((size_t)std::max<int64_t>(order_elems, 1) * 2));
        HIPCHK(c, c->ncand.reserve((size_t)a.P * 4));
    }
    const int block = nmax > 1024 ? 1024 : 256;
    const int nw = block / 64;
    const int nchunks = (std::max(nmax, 1) + 63) / 64;
    const int need_cpw = (nchunks + nw - 1) / nw;
    // instantiated (BLOCK, CPW) pairs; CPW = chunks of 64 keys per wave
    struct Variant { int block, cpw; const void *fn[2]; const void *fn_list[2]; };   // [0]: ballot match, [1]: atomic rank
#define VDET_SV(BL, CP) {BL, CP, {reinterpret_cast<const void *>(sort_kernel<BL, CP, false>), reinterpret_cast<const void *>(sort_kernel<BL, CP, true>)}, \
                         {reinterpret_cast<const void *>(sort_list_kernel<BL, CP, false>), reinterpret_cast<const void *>(sort_list_kernel<BL, CP, true>)}}
    static const Variant variants[] = {VDET_SV(256, 1), VDET_SV(256, 2), VDET_SV(256, 4), VDET_SV(1024, 2), VDET_SV(1024, 4),
                                       VDET_SV(1024, 6), VDET_SV(1024, 8), VDET_SV(1024, 10), VDET_SV(1024, 12),
                                       VDET_SV(1024, 16), VDET_SV(1024, 18)};
#undef VDET_SV
    if (!c->sort_attr_set) {
        size_t stat = 0;
        std::vector<const void *> fns;
        for (const Variant &v : variants) { fns.push_back(v.fn[0]); fns.push_back(v.fn[1]); fns.push_back(v.fn_list[0]); fns.push_back(v.fn_list[1]); }
        fns.push_back(reinterpret_cast<const void *>(walk_kernel));
        for (const void *fn : {reinterpret_cast<const void *>(binsort_kernel<8, false>), reinterpret_cast<const void *>(binsort_kernel<8, true>),
                               reinterpret_cast<const void *>(binsort_kernel<20, false>), reinterpret_cast<const void *>(binsort_kernel<20, true>),
                               reinterpret_cast<const void *>(binsort_kernel<36, false>), reinterpret_cast<const void *>(binsort_kernel<36, true>)})
            fns.push_back(fn);
        for (const void *fn : fns) {
            hipFuncAttributes fa;
            HIPCHK(c, hipFuncGetAttributes(&fa, fn));
            stat = std::max(stat, (size_t)fa.sharedSizeBytes);
        }
        c->dyn_lds_max = c->max_lds - r16(stat);
        for (const void *fn : fns)
            HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dyn_lds_max));
        c->sort_attr_set = true;
    }
    const Variant *var = nullptr;
    for (const Variant &v : variants)
        if (v.block == block && v.cpw >= need_cpw) { var = &v; break; }
    SortParams sp{};
    sp.mode = a.mode; sp.P = a.P; sp.B = a.B; sp.C = a.C;
    sp.scores = a.scores; sp.keys = a.keys; sp.excl = a.excl; sp.use_thr = a.use_thr; sp.thr = a.thr;
    if (a.mode == 0 && !a.keys && !a.walk_only) {
        // class-innermost volume: one coalesced transpose to [F,C,B] keys, then the sort reads rows
        const int64_t F = a.P / a.C;
        const bool have_keys = c->cache_enabled && c->keys_valid && c->keysrc.scores == a.scores && c->keysrc.F == F &&
                               c->keysrc.B == a.B && c->keysrc.C == a.C && c->keysrc.use_thr == (a.use_thr ? 1 : 0) &&
                               (!a.use_thr || c->keysrc.thr == a.thr);     // left there by vdet_volume_pass
        if (!have_keys) c->keys_valid = false;
        HIPCHK(c, c->tkeys.reserve((size_t)a.P * a.B * 4));
        if (!have_keys) {
            StageTimer tm(c, ST_TRANSPOSE);
            hipLaunchKernelGGL(transpose_keys_kernel, dim3((a.B + 63) / 64, (a.C + 63) / 64, (unsigned)F), dim3(256), 0,
                               c->stream, a.scores, c->tkeys.as<uint32_t>(), a.B, a.C, a.use_thr, a.thr);
        }
        HIPCHK(c, hipGetLastError());
        sp.mode = 3;             // keys laid out [P,B] (decode like mode 1)
        sp.keys = c->tkeys.as<uint32_t>();
        sp.scores = nullptr;
        sp.use_thr = 0;
    }
    sp.groups = c->groups.as<GroupDesc>();
    sp.order = a.order_out ? a.order_out : c->order.as<uint16_t>();
    sp.ncand = a.order_out ? a.ncand_out : c->ncand.as<int32_t>();
    sp.topk = a.topk;
    sp.nover = a.nover_out;
    const size_t keysB = r16((size_t)2 * std::max(nmax, 1));     // 16 key bits at a time (see sort_kernel)
    const size_t idxB = r16((size_t)2 * std::max(nmax, 1));
    sp.lds_idxa_off = (int)keysB;
    sp.lds_idxb_off = (int)(keysB + idxB);
    sp.lds_base_off = (int)(keysB + 2 * idxB);
    const size_t lds = keysB + 2 * idxB + (size_t)4 * (nw * 256 + 256 + 4);
    const bool big = !var || lds > c->dyn_lds_max;
    if (big && a.mode != 2)
        return fail(c, VDET_EINVAL, "a frame with %d boxes needs %zu B of LDS for the in-LDS sort; the limit is %zu B "
                                    "(about 18000 boxes per frame)", nmax, lds, c->dyn_lds_max);
    if (big && !a.walk_only) {
        // flat groups (host-buffer entry points) beyond the LDS limit: one global bitonic sort per
        // group.  Slow path for rare, very large single problems (<= 32767 boxes).
        if (!c->host_groups || (int)c->host_groups->size() != a.P) return fail(c, VDET_EINVAL, "internal: group table missing");
        HIPCHK(c, hipMemsetAsync(sp.ncand, 0, (size_t)a.P * 4, c->stream));
        StageTimer tm(c, ST_SORTK);
        for (int g = 0; g < a.P; ++g) {
            const GroupDesc &gd = (*c->host_groups)[(size_t)g];
            const int n = gd.nbox;
            if (n <= 0) continue;
            const uint32_t n2 = pow2ceil((uint32_t)n);
            HIPCHK(c, c->comp.reserve((size_t)n2 * 8));
            hipLaunchKernelGGL(fill_comp_kernel, dim3((n2 + 255) / 256), dim3(256), 0, c->stream, a.scores, a.keys, a.excl,
                               a.use_thr, a.thr, (int64_t)gd.box_off, n, n2, c->comp.as<unsigned long long>(), sp.ncand + g);
            const int rc = sort_comp_desc(c, n2);
            if (rc) return rc;
            hipLaunchKernelGGL(comp_to_order_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                               c->comp.as<unsigned long long>(), n, sp.order + gd.box_off);
        }
        HIPCHK(c, hipGetLastError());
    } else if (!a.walk_only) {
        // volumes of untied scores: the equalised counting sort (binsort_kernels.hpp); what it can not spread -- decided per
        // problem on the device -- lands on a list the LSD kernel works off afterwards (normally empty)
        const int bin_cpw = nmax <= 4096 ? 8 : nmax <= 10240 ? 20 : 36;      // keys per thread, 512 threads
        const size_t bin_lds = binsort_lds_bytes(std::max(nmax, 1), bin_cpw);
        // (threshold / top-k / exclusion lists change ncand: those go to the LSD kernel; a NaN inside an otherwise plain list is
        //  caught per list on the device, binsort_kernels.hpp phase 1)
        const bool use_bin = c->binsort && (sp.mode == 1 || sp.mode == 3) && a.topk == 0 && !a.use_thr && !a.excl && block == 1024 &&
                             nmax <= 512 * bin_cpw && bin_lds + 4096 <= c->dyn_lds_max;
        // frames of at most 384 boxes: one WAVE per list (small_kernels.hpp: the same stable LSD passes without a workgroup)
        const bool use_small = c->small_lists && c->atomic_rank && a.mode != 2 && nmax <= kSmallMax && a.topk == 0 && !a.excl && !a.nover_out;
        StageTimer tm(c, ST_SORTK);
        if (a.mode != 2) c->last_sort_binned = use_bin && !use_small;
        if (use_small) {
            const int kpl = std::max(1, (nmax + 63) / 64);
            const int grid = (((a.P + 3) / 4) + 7) & ~7;
            void *args[] = {&sp};
            const void *fn = kpl == 1 ? reinterpret_cast<const void *>(small_sort_kernel<1>) : kpl == 2 ? reinterpret_cast<const void *>(small_sort_kernel<2>)
                           : kpl == 3 ? reinterpret_cast<const void *>(small_sort_kernel<3>) : kpl == 4 ? reinterpret_cast<const void *>(small_sort_kernel<4>)
                           : kpl == 5 ? reinterpret_cast<const void *>(small_sort_kernel<5>) : reinterpret_cast<const void *>(small_sort_kernel<6>);
            HIPCHK(c, hipLaunchKernel(fn, dim3(grid), dim3(256), args, 0, c->stream));
        } else if (use_bin) {
            HIPCHK(c, c->sortctl.reserve(sizeof(BinSortCtl) + (size_t)a.P * 4));
            HIPCHK(c, hipMemsetAsync(c->sortctl.p, 0, sizeof(BinSortCtl), c->stream));
            BinSortParams bp{};
            const bool floats = sp.keys == nullptr;
            bp.raw = floats ? reinterpret_cast<const uint32_t *>(sp.scores) : sp.keys;
            bp.P = a.P; bp.N = a.B;
            bp.order = sp.order; bp.ncand = sp.ncand;
            bp.ctl = c->sortctl.as<BinSortCtl>();
            bp.fail_list = reinterpret_cast<int32_t *>(c->sortctl.as<char>() + sizeof(BinSortCtl));
            const int per_cu = std::max<int>(1, (int)(c->max_lds / (bin_lds + 4096)));
            const int grid = std::min(a.P, per_cu * c->n_cu);
#define VDET_BSK(CP) (floats ? reinterpret_cast<const void *>(binsort_kernel<CP, true>) : reinterpret_cast<const void *>(binsort_kernel<CP, false>))
            const void *bfn = bin_cpw == 8 ? VDET_BSK(8) : bin_cpw == 20 ? VDET_BSK(20) : VDET_BSK(36);
#undef VDET_BSK
            void *bargs[] = {&bp};
            HIPCHK(c, hipLaunchKernel(bfn, dim3(grid), dim3(512), bargs, bin_lds, c->stream));
            const int32_t *fl = bp.fail_list;
            const int *fc = &bp.ctl->nfail;
            void *largs[] = {&sp, (void *)&fl, (void *)&fc};
            StageTimer tm2(c, ST_SORTFB);
            HIPCHK(c, hipLaunchKernel(var->fn_list[c->atomic_rank ? 1 : 0], dim3(std::min(a.P, 2 * c->n_cu)), dim3(block), largs, lds, c->stream));
        } else {
            const int grid = (a.P + 7) & ~7;
            void *args[] = {&sp};
            HIPCHK(c, hipLaunchKernel(var->fn[c->atomic_rank ? 1 : 0], dim3(grid), dim3(block), args, lds, c->stream));
        }
    }
    HIPCHK(c, hipGetLastError());
    if (a.sort_only) return VDET_OK;
    WalkParams wp{};
    wp.mode = a.mode; wp.P = a.P; wp.B = a.B; wp.C = a.C;
    wp.groups = c->groups.as<GroupDesc>();
    wp.order = a.order_in ? a.order_in : c->order.as<uint16_t>();
    wp.ncand = a.order_in ? a.ncand_in : c->ncand.as<int32_t>();
    wp.row_meta = c->rowmeta.as<uint2>();
    wp.adj = c->adj.as<uint16_t>();
    wp.group_z = c->groupz.as<uint32_t>();
    wp.keep_idx = a.keep_idx;
    wp.keep_cnt = a.keep_cnt;
    wp.cap = a.cap;
    wp.status = &c->d_cnt->status;
    wp.mask_words = (int)(r16((size_t)4 * ((std::max(nmax, 1) + 31) / 32)) / 4);
    wp.group_flags = c->sym_built ? c->gflags.as<uint32_t>() : nullptr;   // frames whose graph is symmetric
    wp.packed = (wp.group_flags && c->wmeta_built) ? 1 : 0;               // regular frames: eight candidates per pass
    wp.wave_words = wp.mask_words + (wp.packed ? 8 * kPackRing : 0);      // + the ring of alive candidates
    wp.wmeta = c->wmeta.as<WalkMeta>();
    wp.t32 = c->gt32;
    wp.wmeta16 = (wp.packed && c->wmeta16_built) ? c->wmeta16.as<uint4>() : nullptr;
    wp.adj32 = c->adj.cap <= 0xFFFFFF00ull ? 1 : 0;
    // frames of at most 384 boxes: one LANE per list, the frames' rows in LDS (small_kernels.hpp); the lists of irregular frames
    // are left to the general walk (walk_rest_kernel: normally nothing)
    const bool small_walk = c->small_lists && a.mode != 2 && nmax <= kSmallMax && wp.group_flags && a.C > 0 && a.P % a.C == 0;
    if (small_walk) {
        const int G = a.P / a.C;
        const int nq = nmax <= 128 ? 1 : nmax <= 256 ? 2 : 3;
        const int nm = std::max(nmax, 1);
        const size_t row_bytes = (size_t)nm * 4 * nq * 4;
        const int fpw = a.C > 32 ? 1 : (int)std::max<size_t>(1, std::min<size_t>((size_t)(64 / a.C), (size_t)(44 * 1024) / row_bytes));
        const size_t lds_bytes = (size_t)fpw * row_bytes;
        // a list's candidates four per 8-byte load: rows of B % 4 == 0 entries from an 8-byte aligned base (the caller's lists of
        // vdet_nms_volume_ordered may sit anywhere)
        const int vec4 = (a.B % 4 == 0 && ((uintptr_t)wp.order & 7) == 0) ? 1 : 0;
        const int grid = (G + fpw - 1) / fpw;
        StageTimer tm(c, ST_WALK);
#define VDET_SMALLW(NQ_, V_) hipLaunchKernelGGL((small_walk_kernel<NQ_, V_>), dim3(grid), dim3(64), lds_bytes, c->stream, wp, G, fpw, nm)
        if (vec4) { if (nq == 1) VDET_SMALLW(1, true); else if (nq == 2) VDET_SMALLW(2, true); else VDET_SMALLW(3, true); }
        else { if (nq == 1) VDET_SMALLW(1, false); else if (nq == 2) VDET_SMALLW(2, false); else VDET_SMALLW(3, false); }
#undef VDET_SMALLW
        if (!c->all_regular)      // (asynchronous build: not known on the host -- the kernel looks at the frames' flags)
            hipLaunchKernelGGL(walk_rest_kernel, dim3(std::min(G, 4 * c->n_cu)), dim3(256), (size_t)wp.wave_words * 4 * 4, c->stream, wp, G);
    } else {
        const int nblk = (((a.P + 3) / 4) + 7) & ~7;
        StageTimer tm(c, ST_WALK);
        hipLaunchKernelGGL(walk_kernel, dim3(nblk), dim3(256), (size_t)wp.wave_words * 4 * 4, c->stream, wp);
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

int sort_groups_by_keys(vdet_ctx *c, int G, int nmax, int64_t ntot)
{
    SortWalkArgs a{};
    a.sort_only = true;
    a.mode = 2; a.P = G;
    a.keys = c->xkeys.as<uint32_t>();
    a.order_out = c->xord.as<uint16_t>();
    a.ncand_out = c->xncand.as<int32_t>();
    const bool st = c->timing;
    c->timing = false;                       // keep the per-(frame,class) sort stage clean
    const int rc = launch_sort_walk(c, a, nmax, ntot);
    c->timing = st;
    return rc;
}

// descending sort of n composites held in c->comp (zero padded to a power of two)
int sort_comp_desc(vdet_ctx *c, uint32_t n)
{
    if (n <= 1) return VDET_OK;
    const uint32_t n2 = pow2ceil(n);
    unsigned long long *d = c->comp.as<unsigned long long>();
    StageTimer tm(c, ST_SORT);
    const uint32_t nb = (n2 + 2047) / 2048;
    hipLaunchKernelGGL(bitonic_lds_kernel, dim3(nb), dim3(1024), 0, c->stream, d, n2, 2u, std::min(n2, 2048u));
    for (uint32_t k = 4096; k <= n2 && k; k <<= 1) {
        for (uint32_t j = k >> 1; j >= 2048; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_step, dim3((n2 + 255) / 256), dim3(256), 0, c->stream, d, n2, j, k);
        hipLaunchKernelGGL(bitonic_lds_kernel, dim3(nb), dim3(1024), 0, c->stream, d, n2, k, k);
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

// Shared tail of nms / vid_nms / track_det_nms on grouped, uploaded data (graph already built):
//   sort -> walk -> survivors of all groups as composites -> global descending sort -> indices.
int nms_grouped_tail(vdet_ctx *c, NmsPlan &pl, const float *d_scores, const uint32_t *d_keys,
                     const uint8_t *d_excl, int64_t *h_keep, int64_t *n_keep)
{
    const int P = (int)pl.groups.size();
    HIPCHK(c, c->keepidx.reserve((size_t)pl.ntot * 4));
    HIPCHK(c, c->keepcnt.reserve((size_t)P * 4));
    SortWalkArgs a{};
    a.mode = 2; a.P = P;
    a.scores = d_scores; a.keys = d_keys; a.excl = d_excl;
    a.keep_idx = c->keepidx.as<int32_t>();
    a.keep_cnt = c->keepcnt.as<int32_t>();
    a.cap = 0;
    int rc = launch_sort_walk(c, a, pl.nmax, pl.ntot);
    if (rc) return rc;
    const uint32_t n2 = pow2ceil((uint32_t)std::max<int64_t>(pl.ntot, 1));
    HIPCHK(c, c->comp.reserve((size_t)n2 * 8));
    HIPCHK(c, hipMemsetAsync(c->comp.p, 0, (size_t)n2 * 8, c->stream));
    hipLaunchKernelGGL(gather_comp_kernel, dim3(P), dim3(256), 0, c->stream, c->groups.as<GroupDesc>(), P,
                       c->keepidx.as<int32_t>(), c->keepcnt.as<int32_t>(), d_scores, d_keys,
                       c->origidx.as<uint32_t>(), c->comp.as<unsigned long long>(), &c->d_cnt->glob_cnt);
    HIPCHK(c, hipGetLastError());
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    rc = translate_status(c, h.status);
    if (rc) return rc;
    const uint32_t nk = h.glob_cnt;
    rc = sort_comp_desc(c, nk);
    if (rc) return rc;
    if (nk) {
        HIPCHK(c, c->out64.reserve((size_t)nk * 8));
        hipLaunchKernelGGL(comp_to_index_kernel, dim3((nk + 255) / 256), dim3(256), 0, c->stream,
                           c->comp.as<unsigned long long>(), nk, c->out64.as<int64_t>());
        HIPCHK(c, hipMemcpyAsync(h_keep, c->out64.p, (size_t)nk * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, host_sync(c));
    }
    *n_keep = nk;
    return VDET_OK;
}

// Group rows by frame value with float32 equality semantics (utils/nms.pyx:111: f_idx[i] != f_idx[j]):
// -0.0 == +0.0; a NaN frame equals nothing, not even itself (singleton groups).
// perm = original indices in grouped order.
int group_by_frame(vdet_ctx *c, const float *frames, int64_t n, int64_t ld, std::vector<int64_t> &perm,
                   std::vector<GroupDesc> &groups)
{
    perm.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = i;
    auto key = [&](int64_t i) { return frames[i * ld] + 0.0f; };
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
        const float fa = key(a), fb = key(b);
        const bool na = fa != fa, nb = fb != fb;
        if (na || nb) return !na && nb;   // NaNs last
        return fa < fb;
    });
    groups.clear();
    int64_t s = 0;
    while (s < n) {
        const float f = key(perm[(size_t)s]);
        int64_t e = s + 1;
        if (f == f)
            while (e < n && key(perm[(size_t)e]) == f) ++e;
        if (e - s > 32767)
            return fail(c, VDET_EINVAL, "%lld detections on one frame; the limit is 32767", (long long)(e - s));
        groups.push_back({(int32_t)s, (int32_t)(e - s), 0});
        s = e;
    }
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
// The drop-in calls on <= kFusedMax rows: ONE launch (fused_kernels.hpp).  The rows are packed into host-mapped memory the
// kernel reads directly and the kept list comes back the same way: a call is one launch and one host wait -- no staging
// copies, no scratch shared with the volume entry points (the context's cache stays valid).
// ---------------------------------------------------------------------------------------------
constexpr size_t kFusedInBytes = (size_t)kFusedMax * 6 * 4 + (size_t)kFusedMax * 4 + (size_t)kFusedMaxTracks * 5 * 4;
constexpr size_t kFusedOutBytes = (size_t)(2 + kFusedMax) * 4;

// Wait for a single-launch call by POLLING its "done" words in host-mapped memory (the kernels publish the kept count last,
// release / system scope) instead of hipStreamSynchronize: at T-CNN's call sizes the kernel runs ~15 us and the runtime's
// wake-up was a third of the call (measured through the python module: nms of 100 rows 0.050 -> see INTEGRATION.md).  The done
// words are set to -1 before the launch; hipStreamQuery pushes the launch out; after 2 ms of polling (a long kernel, a busy
// device) the wait falls back to the stream synchronisation, which is also what reports a failed launch.
hipError_t fused_wait(vdet_ctx *c, volatile int32_t *hdr, int64_t K)
{
    ++c->n_host_syncs;
    (void)hipStreamQuery(c->stream);
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t k = 0; k < K; ++k) {
        unsigned spins = 0;
        while (hdr[2 * k + 1] == -1) {
            __builtin_ia32_pause();
            if ((++spins & 255u) == 0u &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                return hipStreamSynchronize(c->stream);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return hipSuccess;
}

int fused_call(vdet_ctx *c, const float *h_rows, int64_t n, int64_t ld, int ncols, double thresh, const int64_t *h_order,
               const float *h_tracks, int64_t t, int64_t ldt, int64_t *h_keep, int64_t *n_keep)
{
    if (!c->fused_in) {
        void *pi = nullptr, *po = nullptr;
        if (hipHostMalloc(&pi, kFusedInBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc(&po, kFusedOutBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            if (pi) (void)hipHostFree(pi);
            return fail(c, VDET_ENOMEM, "host-mapped staging memory for the single-launch calls");
        }
        for (const void *fn : {reinterpret_cast<const void *>(fused_nms_kernel<256>), reinterpret_cast<const void *>(fused_nms_kernel<1024>)}) {
            hipFuncAttributes fa;       // (the dynamic limit is what the kernel's static LDS leaves of the CU's 160 KiB)
            hipError_t e = hipFuncGetAttributes(&fa, fn);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->max_lds - (((size_t)fa.sharedSizeBytes + 15) & ~(size_t)15)));
            if (e != hipSuccess) {
                (void)hipHostFree(pi); (void)hipHostFree(po);
                return fail(c, VDET_EHIP, "hipFuncSetAttribute(fused_nms_kernel) failed: %s", hipGetErrorString(e));
            }
        }
        void *di = nullptr, *dq = nullptr;
        if (hipHostGetDevicePointer(&di, pi, 0) != hipSuccess || hipHostGetDevicePointer(&dq, po, 0) != hipSuccess) {
            (void)hipHostFree(pi); (void)hipHostFree(po);
            return fail(c, VDET_EHIP, "hipHostGetDevicePointer failed for the single-launch staging memory");
        }
        c->fused_in = pi; c->fused_out = po; c->fused_in_dev = di; c->fused_out_dev = dq;
    }
    float *rows = static_cast<float *>(c->fused_in);
    int32_t *rank = reinterpret_cast<int32_t *>(rows + (size_t)kFusedMax * 6);
    float *trk = reinterpret_cast<float *>(rank + kFusedMax);
    volatile int32_t *out = static_cast<volatile int32_t *>(c->fused_out);
    if (ld == ncols) memcpy(rows, h_rows, (size_t)n * ncols * 4);
    else for (int64_t i = 0; i < n; ++i) memcpy(rows + i * ncols, h_rows + i * ld, (size_t)ncols * 4);
    if (h_order) {   // caller-supplied order: priority = position (earlier = higher)
        for (int64_t i = 0; i < n; ++i) rank[i] = 0;
        for (int64_t pos = 0; pos < n; ++pos) {
            const int64_t i = h_order[pos];
            if (i < 0 || i >= n || rank[i]) return fail(c, VDET_EINVAL, "order is not a permutation of 0..n-1");
            rank[i] = (int32_t)(n - pos);
        }
    }
    for (int64_t j = 0; j < t; ++j) memcpy(trk + j * 5, h_tracks + j * ldt, 20);
    FusedParams fp{};
    fp.rows = static_cast<const float *>(c->fused_in_dev);
    fp.rank = h_order ? reinterpret_cast<const int32_t *>(fp.rows + (size_t)kFusedMax * 6) : nullptr;
    fp.tracks = h_tracks ? reinterpret_cast<const float *>(reinterpret_cast<const int32_t *>(fp.rows + (size_t)kFusedMax * 6) + kFusedMax) : nullptr;
    fp.n = (int)n; fp.ncols = ncols; fp.t = (int)t;
    fp.t32 = thresh_to_f32(thresh);
    fp.hdr = static_cast<int32_t *>(c->fused_out_dev);
    fp.kept = fp.hdr + 2;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const size_t lds = fused_lds_bytes((int)n, n2);
    out[1] = -1;                      // the "done" word: the kernel stores the kept count there last
    {
        StageTimer tm(c, ST_WALK);
        if (n <= 128) hipLaunchKernelGGL(fused_nms_kernel<256>, dim3(1), dim3(256), lds, c->stream, fp);
        else hipLaunchKernelGGL(fused_nms_kernel<1024>, dim3(1), dim3(1024), lds, c->stream, fp);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, fused_wait(c, out, 1));
    if (out[0] & kStDivZero) return fail(c, VDET_EDIVZERO, "float division (zero union)");
    const int64_t nk = out[1];
    if (nk < 0 || nk > n) return fail(c, VDET_EHIP, "internal: single-launch NMS returned %lld of %lld rows", (long long)nk, (long long)n);
    for (int64_t k = 0; k < nk; ++k) h_keep[k] = out[2 + k];
    *n_keep = nk;
    return VDET_OK;
}

// K independent track_det_nms problems (vdet/track.py:236-250: one per tracked box, each on its own frame's still-kept
// detections) in ONE launch of K workgroups and ONE host wait.  Host-mapped staging, grown on demand:
//   in : rows [M,6] f32 | tracks [T,5] f32 | off [K+1] i32 | toff [K+1] i32         out: hdr [K,2] i32 | kept [M] i32
int fused_batch_call(vdet_ctx *c, const float *h_tracks, const int64_t *h_toff, int64_t ldt, const float *h_dets, const int64_t *h_off,
                     int64_t K, int64_t ldd, double thresh, int64_t *h_keep, int64_t *h_nkeep, int64_t max_m)
{
    const int64_t M = h_off[K], T = h_toff ? h_toff[K] : K;
    const size_t in_bytes = (((size_t)M * 24 + (size_t)T * 20 + 15) & ~(size_t)15) + 2 * (size_t)(K + 1) * 4;
    const size_t out_bytes = (size_t)K * 8 + (size_t)M * 4;
    if (in_bytes > c->fbatch_in_cap || out_bytes > c->fbatch_out_cap) {
        if (c->fbatch_in) (void)hipHostFree(c->fbatch_in);
        if (c->fbatch_out) (void)hipHostFree(c->fbatch_out);
        c->fbatch_in = c->fbatch_out = c->fbatch_in_dev = c->fbatch_out_dev = nullptr;
        c->fbatch_in_cap = c->fbatch_out_cap = 0;
        const size_t ci = std::max<size_t>(in_bytes + in_bytes / 2, (size_t)1 << 20), co = std::max<size_t>(out_bytes + out_bytes / 2, (size_t)1 << 18);
        void *pi = nullptr, *po = nullptr, *di = nullptr, *dq = nullptr;
        if (hipHostMalloc(&pi, ci, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc(&po, co, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            if (pi) (void)hipHostFree(pi);
            return fail(c, VDET_ENOMEM, "host-mapped staging memory for vdet_track_det_nms_batch");
        }
        if (hipHostGetDevicePointer(&di, pi, 0) != hipSuccess || hipHostGetDevicePointer(&dq, po, 0) != hipSuccess) {
            (void)hipHostFree(pi); (void)hipHostFree(po);
            return fail(c, VDET_EHIP, "hipHostGetDevicePointer failed for the batch staging memory");
        }
        c->fbatch_in = pi; c->fbatch_out = po; c->fbatch_in_dev = di; c->fbatch_out_dev = dq;
        c->fbatch_in_cap = ci; c->fbatch_out_cap = co;
    }
    if (!c->fbatch_attr) {
        for (const void *fn : {reinterpret_cast<const void *>(fused_nms_batch_kernel<256>), reinterpret_cast<const void *>(fused_nms_batch_kernel<1024>)}) {
            hipFuncAttributes fa;
            hipError_t e = hipFuncGetAttributes(&fa, fn);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->max_lds - (((size_t)fa.sharedSizeBytes + 15) & ~(size_t)15)));
            if (e != hipSuccess) return fail(c, VDET_EHIP, "hipFuncSetAttribute(fused_nms_batch_kernel) failed: %s", hipGetErrorString(e));
        }
        c->fbatch_attr = true;
    }
    float *rows = static_cast<float *>(c->fbatch_in);
    float *trk = rows + (size_t)M * 6;
    const size_t tab = ((size_t)M * 24 + (size_t)T * 20 + 15) & ~(size_t)15;
    int32_t *off = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(c->fbatch_in) + tab);
    int32_t *toff = off + (K + 1);
    if (ldd == 6) memcpy(rows, h_dets, (size_t)M * 24);
    else for (int64_t i = 0; i < M; ++i) memcpy(rows + i * 6, h_dets + i * ldd, 24);
    for (int64_t j = 0; j < T; ++j) memcpy(trk + j * 5, h_tracks + j * ldt, 20);
    for (int64_t k = 0; k <= K; ++k) { off[k] = (int32_t)h_off[k]; toff[k] = (int32_t)(h_toff ? h_toff[k] : k); }
    volatile int32_t *hdr = static_cast<volatile int32_t *>(c->fbatch_out);
    volatile int32_t *kept = hdr + 2 * K;
    FusedBatchParams bp{};
    const unsigned char *din = static_cast<const unsigned char *>(c->fbatch_in_dev);
    bp.rows = reinterpret_cast<const float *>(din);
    bp.tracks = bp.rows + (size_t)M * 6;
    bp.off = reinterpret_cast<const int32_t *>(din + tab);
    bp.toff = bp.off + (K + 1);
    bp.t32 = thresh_to_f32(thresh);
    bp.hdr = static_cast<int32_t *>(c->fbatch_out_dev);
    bp.kept = bp.hdr + 2 * K;
    int n2 = 64;
    while (n2 < max_m) n2 <<= 1;
    const size_t lds = fused_lds_bytes((int)std::max<int64_t>(max_m, 1), n2);
    for (int64_t k = 0; k < K; ++k) hdr[2 * k + 1] = -1;      // the problems' "done" words
    {
        StageTimer tm(c, ST_WALK);
        if (max_m <= 128) hipLaunchKernelGGL(fused_nms_batch_kernel<256>, dim3((unsigned)K), dim3(256), lds, c->stream, bp);
        else hipLaunchKernelGGL(fused_nms_batch_kernel<1024>, dim3((unsigned)K), dim3(1024), lds, c->stream, bp);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, fused_wait(c, hdr, K));
    bool divz = false;
    for (int64_t k = 0; k < K; ++k) {
        const int64_t m = h_off[k + 1] - h_off[k], nk = hdr[2 * k + 1];
        if (hdr[2 * k] & kStDivZero) divz = true;
        if (nk < 0 || nk > m) return fail(c, VDET_EHIP, "internal: batched single-launch NMS returned %lld of %lld rows", (long long)nk, (long long)m);
        h_nkeep[k] = nk;
        for (int64_t q = 0; q < nk; ++q) h_keep[h_off[k] + q] = kept[h_off[k] + q];
    }
    if (divz) return fail(c, VDET_EDIVZERO, "float division (zero union)");
    return VDET_OK;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

const char *vdet_version(void) { return "vdet_hip 0.1 gfx950"; }

int vdet_create(vdet_ctx **out, int device)
{
    if (!out) return VDET_EINVAL;
    *out = nullptr;
    vdet_ctx *c = new (std::nothrow) vdet_ctx;
    if (!c) return VDET_ENOMEM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete c;
        return VDET_EHIP;   // fail loudly: there is NO CPU fallback in this library
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) { delete c; return VDET_EHIP; }
    }
    if (device >= ndev || hipSetDevice(device) != hipSuccess) { delete c; return VDET_EINVAL; }
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // gfx950: one workgroup may own the CU's whole 160 KiB LDS (opt-in via hipFuncSetAttribute)
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) c->max_lds = 64 * 1024;
    }
    if (const char *e = getenv("VDET_FORCE_GENERAL")) c->force_general = atoi(e) != 0;
    if (const char *e = getenv("VDET_BITS_BUDGET_MB")) {
        const long mb = atol(e);
        if (mb > 0) c->bits_budget = (size_t)mb << 20;
    }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&c->d_cnt, sizeof(Counters)) != hipSuccess) {
        delete c;
        return VDET_EHIP;
    }
    c->stream = c->own_stream;
    (void)hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream);
    if (const char *e = getenv("VDET_NO_INDEX")) c->no_index = atoi(e) != 0;
    if (const char *e = getenv("VDET_NO_LAZY")) c->no_lazy = atoi(e) != 0;
    if (const char *e = getenv("VDET_NO_FUSED")) c->no_fused = atoi(e) != 0;
    if (const char *e = getenv("VDET_ADJ_ROWS")) c->adj_rows = atoi(e) != 0;
    if (const char *e = getenv("VDET_DIRECT_LISTS")) c->direct_lists = atoi(e) != 0;
    if (const char *e = getenv("VDET_DIRECT_CAP")) { const int v = atoi(e); if (v >= 8 && v <= 32760) c->direct_cap = (uint32_t)(v & ~7); }
    if (const char *e = getenv("VDET_BINSORT")) c->binsort = atoi(e) != 0;
    if (const char *e = getenv("VDET_SMALL_LISTS")) c->small_lists = atoi(e) != 0;
    {   // probe: do returning LDS atomics resolve same-address lanes in ascending lane order?
        const int npat = 4096;
        std::vector<uint8_t> pats((size_t)npat * 64);
        uint32_t rs = 12345u;
        auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return rs >> 8; };
        for (int t = 0; t < npat; ++t) {
            const int kind = t % 8;
            const int nd = kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 3 : kind == 3 ? 8 : kind == 4 ? 32 : kind == 5 ? 64 : kind == 6 ? 200 : 254;
            for (int l = 0; l < 64; ++l) {
                uint8_t d = (uint8_t)(rnd() % nd);
                if (kind == 7 && (rnd() & 3) == 0) d = 255;          // inactive lanes
                if (t % 16 == 9) d = (uint8_t)((l * (1 + t % 7)) % nd); // structured strides
                pats[(size_t)t * 64 + l] = d;
            }
        }
        bool ok = false;
        DevBuf pb;
        if (pb.reserve(pats.size()) == hipSuccess &&
            hipMemcpyAsync(pb.p, pats.data(), pats.size(), hipMemcpyHostToDevice, c->stream) == hipSuccess) {
            hipLaunchKernelGGL(lds_atomic_order_probe, dim3(256), dim3(64), 0, c->stream, pb.as<uint8_t>(), npat,
                               &c->d_cnt->status);
            Counters h;
            if (hipMemcpyAsync(&h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                host_sync(c) == hipSuccess)
                ok = (h.status == 0);
        }
        pb.release();
        (void)hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream);
        c->atomic_rank = ok;
        if (const char *e = getenv("VDET_ATOMIC_RANK")) c->atomic_rank = c->atomic_rank && atoi(e) != 0;
    }
    {   // does wave_transpose64 (K1s's transposed emission) behave as derived on this device?
        const int n = 64;
        std::vector<uint64_t> hin((size_t)n * 64), hout((size_t)n * 64);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < hin.size(); ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            hin[i] = (i / 64) == 0 ? (1ull << (i % 64)) : ((i / 64) == 1 ? (uint64_t)(i % 64 == 5 ? ~0ull : 0ull) : x);
        }
        bool ok = false;
        DevBuf bi, bo;
        if (bi.reserve(hin.size() * 8) == hipSuccess && bo.reserve(hin.size() * 8) == hipSuccess &&
            hipMemcpyAsync(bi.p, hin.data(), hin.size() * 8, hipMemcpyHostToDevice, c->stream) == hipSuccess) {
            hipLaunchKernelGGL(wave_transpose_probe, dim3(16), dim3(64), 0, c->stream, bi.as<uint64_t>(), bo.as<uint64_t>(), n);
            if (hipMemcpyAsync(hout.data(), bo.p, hout.size() * 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                host_sync(c) == hipSuccess) {
                ok = true;
                for (int t = 0; t < n && ok; ++t)
                    for (int r = 0; r < 64 && ok; ++r)
                        for (int k = 0; k < 64; ++k)
                            if (((hin[(size_t)t * 64 + r] >> k) & 1ull) != ((hout[(size_t)t * 64 + k] >> r) & 1ull)) { ok = false; break; }
            }
        }
        bi.release(); bo.release();
        c->wave_transpose = ok;
        if (const char *e = getenv("VDET_WAVE_TRANSPOSE")) c->wave_transpose = c->wave_transpose && atoi(e) != 0;
    }
    *out = c;
    return VDET_OK;
}

int vdet_destroy(vdet_ctx *c)
{
    if (!c) return VDET_OK;
    (void)hipSetDevice(c->device);
    (void)host_sync(c);
    DevBuf *bufs[] = {&c->boxes, &c->scores, &c->keys, &c->excl, &c->frames, &c->groups, &c->tiles, &c->bits,
                      &c->rowz, &c->rowmeta, &c->groupz, &c->adj, &c->comp, &c->origidx, &c->out64,
                      &c->trk_frames, &c->trk_boxes, &c->b1, &c->b2, &c->iou_out, &c->order, &c->ncand, &c->keepidx,
                      &c->keepcnt, &c->gflags, &c->pairs, &c->tkeys, &c->tstate, &c->visited, &c->heads, &c->xkeys, &c->xord, &c->xncand, &c->linkmemo, &c->linkstats, &c->linkwarm, &c->linkorder, &c->linkchains, &c->linknodes, &c->tracknode, &c->rtodo,
                      &c->xbox, &c->xbox16, &c->xord16, &c->xcum, &c->xinfo, &c->wmeta, &c->wmeta16, &c->reachtab, &c->rowperm, &c->qreach, &c->ditems, &c->striptot, &c->stripoff, &c->sortctl, &c->segtab, &c->vidtab, &c->nover, &c->ordncand};
    for (DevBuf *b : bufs) b->release();
    for (DevBuf &b : c->tmp) b.release();
    for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->fused_in) (void)hipHostFree(c->fused_in);
    if (c->fused_out) (void)hipHostFree(c->fused_out);
    if (c->fbatch_in) (void)hipHostFree(c->fbatch_in);
    if (c->fbatch_out) (void)hipHostFree(c->fbatch_out);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return VDET_OK;
}

int vdet_set_stream(vdet_ctx *c, void *s)
{
    if (!c) return VDET_EINVAL;
    c->stream = (hipStream_t)s;        // verbatim: NULL is HIP's null stream (torch's default stream)
    return VDET_OK;
}

int vdet_reset_stream(vdet_ctx *c)
{
    if (!c) return VDET_EINVAL;
    c->stream = c->own_stream;
    return VDET_OK;
}

const char *vdet_last_error(vdet_ctx *c) { return c ? c->err.c_str() : "null context"; }

int vdet_set_cache(vdet_ctx *c, int enable)
{
    if (!c) return VDET_EINVAL;
    c->cache_enabled = enable != 0;
    c->graph_valid = c->lists_valid = c->index_valid = c->keys_valid = c->nodes_valid = false;
    return VDET_OK;
}

int vdet_invalidate(vdet_ctx *c)
{
    if (!c) return VDET_EINVAL;
    c->graph_valid = c->lists_valid = c->index_valid = c->keys_valid = c->nodes_valid = false;
    return VDET_OK;
}

int vdet_query(vdet_ctx *c, int what)
{
    if (!c) return VDET_EINVAL;
    if (what == 0) return c->atomic_rank ? 1 : 0;
    if (what == 1) return c->n_cu;
    if (what == 2) return c->all_regular ? 1 : 0;
    if (what == 3) return c->wave_transpose ? 1 : 0;
    if (what == 8) return (int)std::min<long long>(c->n_host_syncs, 0x7FFFFFFF);
    if (what == 9) {   // problems the last volume sort's counting kernel handed to the LSD kernel (-1: it did not run)
        if (!c->last_sort_binned || !c->sortctl.p) return -1;
        BinSortCtl h{};
        if (hipMemcpyAsync(&h, c->sortctl.p, sizeof h, hipMemcpyDeviceToHost, c->stream) != hipSuccess || host_sync(c) != hipSuccess) return VDET_EHIP;
        return h.nfail;
    }
    if (what >= 4 && what <= 7) {     // link steps of the last tracking call served by the memo (4) / scanned (5); 6 / 7: the warm-up's
        unsigned int h[4] = {0, 0, 0, 0};
        if (!c->linkstats.p) return 0;
        if (hipMemcpyAsync(h, c->linkstats.p, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            host_sync(c) != hipSuccess) return VDET_EHIP;
        return (int)std::min<unsigned int>(h[what - 4], 0x7FFFFFFFu);
    }
    return VDET_EINVAL;
}

int vdet_set_timing(vdet_ctx *c, int enable)
{
    if (!c) return VDET_EINVAL;
    c->timing = enable != 0;
    c->timing_accumulate = enable == 2;
    c->ev_used = 0;
    return VDET_OK;
}

int vdet_last_timing_ms(vdet_ctx *c, float *out16)
{
    float *out8 = out16;
    if (!c || !out8) return VDET_EINVAL;
    HIPCHK(c, host_sync(c));
    for (int i = 0; i < ST_COUNT; ++i) { c->last_ms[i] = 0; c->last_launches[i] = 0; }
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second) == hipSuccess) {
            c->last_ms[c->ev_stage[i]] += ms;
            c->last_launches[c->ev_stage[i]] += 1;
        }
    }
    for (int i = 0; i < ST_COUNT; ++i) out8[i] = c->last_ms[i];
    c->ev_used = 0;
    return VDET_OK;
}

int vdet_last_launches(vdet_ctx *c, int *out16)
{
    if (!c || !out16) return VDET_EINVAL;
    for (int i = 0; i < ST_COUNT; ++i) out16[i] = c->last_launches[i];
    return VDET_OK;
}

int vdet_sync(vdet_ctx *c)
{
    if (!c) return VDET_EINVAL;
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    HIPCHK(c, hipMemsetAsync(&c->d_cnt->status, 0, kPerBuildOff, c->stream));     // status, eindex, pool_max
    h.pool_used = std::max(h.pool_used, h.pool_max);
    const int l = c->latched;
    c->latched = 0;
    if ((h.status & kStPoolAsync) || l == VDET_EAGAIN) {
        // an asynchronous graph build (vdet_set_async) ran out of adjacency pool: everything enqueued
        // since is invalid.  The pool is enlarged here, so running the same calls again succeeds.
        // (handled before a latched error of the same window is reported: the truncated graph must not be reused)
        c->graph_valid = c->lists_valid = c->nodes_valid = false;
        if (h.status & kStDirect) {      // a direct slot overflowed: the bit-matrix path from now on, its pool sized from the cursors
            c->direct_lists = false;
            unsigned long long need = 0ull;
            if (c->direct_ntot > 0 && c->striptot.reserve(8) == hipSuccess &&
                hipMemsetAsync(c->striptot.p, 0, 8, c->stream) == hipSuccess) {
                hipLaunchKernelGGL(deg_sum_kernel, dim3(1024), dim3(256), 0, c->stream, c->rowz.as<uint32_t>(), c->direct_ntot,
                                   c->striptot.as<unsigned long long>());
                if (hipMemcpyAsync(&need, c->striptot.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || host_sync(c) != hipSuccess) need = 0ull;
            }
            const unsigned long long dyn = h.pool_used > c->direct_fixed ? h.pool_used - c->direct_fixed : 0ull;
            h.pool_used = need + dyn + 4096;
            c->pool_hint = 0;            // (the hint of the direct builds counted the fixed slots)
        }
        c->pool_hint = std::max(c->pool_hint, h.pool_used);
        if (h.pool_used <= 0xFFFFFFFFull) (void)c->adj.reserve((size_t)(h.pool_used + h.pool_used / 2) * 2 + 4096);
        if (l && l != VDET_EAGAIN) return l;
        return fail(c, VDET_EAGAIN, "adjacency pool overflow in an asynchronous graph build (%llu entries needed): the "
                                    "results since the last vdet_sync are invalid; the pool has been enlarged, run the calls again",
                    (unsigned long long)h.pool_used);
    }
    if (l) return l;
    c->pool_hint = std::max(c->pool_hint, h.pool_used);
    if (h.eindex) return fail(c, VDET_EINDEX, "list index out of range");
    return translate_status(c, h.status);
}

int vdet_set_async(vdet_ctx *c, int enable)
{
    if (!c) return VDET_EINVAL;
    c->async_enabled = enable != 0;
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
int vdet_nms_f32(vdet_ctx *c, const float *h_dets, int64_t n, int64_t ld, int ncols, double thresh,
                 const int64_t *h_order, int64_t *h_keep, int64_t *n_keep)
{
    if (!c || !n_keep) return VDET_EINVAL;
    *n_keep = 0;
    if (n < 0 || (ncols != 5 && ncols != 6) || (n > 0 && (!h_dets || !h_keep || ld < ncols)))
        return fail(c, VDET_EINVAL, "dets must be float32 [n,%d]", ncols);
    if (n == 0) return VDET_OK;
    if (n > 0x7FFFFFFF) return fail(c, VDET_EINVAL, "too many detections");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (n <= kFusedMax && !c->no_fused)      // the T-CNN call sizes: one launch (fused_kernels.hpp)
        return fused_call(c, h_dets, n, ld, ncols, thresh, h_order, nullptr, 0, 0, h_keep, n_keep);
    const int o = ncols == 6 ? 1 : 0;

    NmsPlan pl;
    std::vector<int64_t> perm;
    if (o) {
        int rc = group_by_frame(c, h_dets, n, ld, perm, pl.groups);
        if (rc) return rc;
    } else {
        if (n > 32767) return fail(c, VDET_EINVAL, "%lld boxes in one image; the limit is 32767", (long long)n);
        perm.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = i;
        pl.groups.push_back({0, (int32_t)n, 0});
    }
    make_plan(c, pl);

    std::vector<float> hb((size_t)n * 4), hs((size_t)n);
    std::vector<uint32_t> hidx((size_t)n), hkeys;
    for (int64_t r = 0; r < n; ++r) {
        const float *row = h_dets + perm[(size_t)r] * ld + o;
        memcpy(&hb[(size_t)r * 4], row, 16);
        hs[(size_t)r] = row[4];
        hidx[(size_t)r] = (uint32_t)perm[(size_t)r];
    }
    if (h_order) {   // caller-supplied order: priority = position (earlier = higher)
        std::vector<uint32_t> rank((size_t)n, 0);
        for (int64_t pos = 0; pos < n; ++pos) {
            const int64_t i = h_order[pos];
            if (i < 0 || i >= n || rank[(size_t)i]) return fail(c, VDET_EINVAL, "order is not a permutation of 0..n-1");
            rank[(size_t)i] = (uint32_t)(n - pos);
        }
        hkeys.resize((size_t)n);
        for (int64_t r = 0; r < n; ++r) hkeys[(size_t)r] = rank[(size_t)perm[(size_t)r]];
    }
    HIPCHK(c, c->boxes.reserve((size_t)n * 16));
    HIPCHK(c, c->scores.reserve((size_t)n * 4));
    HIPCHK(c, c->origidx.reserve((size_t)n * 4));
    HIPCHK(c, hipMemcpyAsync(c->boxes.p, hb.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->scores.p, hs.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->origidx.p, hidx.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    const uint32_t *d_keys = nullptr;
    if (h_order) {
        HIPCHK(c, c->keys.reserve((size_t)n * 4));
        HIPCHK(c, hipMemcpyAsync(c->keys.p, hkeys.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        d_keys = c->keys.as<uint32_t>();
    }
    HIPCHK(c, host_sync(c));
    // the graph / lists / index of an earlier d_* call are overwritten below (shared scratch)
    c->graph_valid = c->lists_valid = c->index_valid = false;
    HostGroupsGuard guard(c);
    int rc = build_graph(c, c->boxes.as<float4>(), pl, thresh_to_f32(thresh), thresh, false);
    if (rc) return rc;
    return nms_grouped_tail(c, pl, c->scores.as<float>(), d_keys, nullptr, h_keep, n_keep);
}

int vdet_track_det_nms_f32(vdet_ctx *c, const float *h_tracks, int64_t t, int64_t ldt, const float *h_dets,
                           int64_t m, int64_t ldd, double thresh, int64_t *h_keep, int64_t *n_keep)
{
    if (!c || !n_keep) return VDET_EINVAL;
    *n_keep = 0;
    if (t < 0 || m < 0 || (t > 0 && (!h_tracks || ldt < 5)) || (m > 0 && (!h_dets || !h_keep || ldd < 6)))
        return fail(c, VDET_EINVAL, "tracks must be float32 [t,5], dets float32 [m,6]");
    if (m == 0) return VDET_OK;
    if (m > 0x7FFFFFFF || t > 0x7FFFFFFF) return fail(c, VDET_EINVAL, "too many rows");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (m <= kFusedMax && t <= kFusedMaxTracks && !c->no_fused)      // the T-CNN call sizes: one launch (fused_kernels.hpp)
        return fused_call(c, h_dets, m, ldd, 6, thresh, nullptr, t > 0 ? h_tracks : nullptr, t, ldt, h_keep, n_keep);
    NmsPlan pl;
    std::vector<int64_t> perm;
    int rc = group_by_frame(c, h_dets, m, ldd, perm, pl.groups);
    if (rc) return rc;
    make_plan(c, pl);
    std::vector<float> hb((size_t)m * 4), hs((size_t)m), hf((size_t)m);
    std::vector<uint32_t> hidx((size_t)m);
    for (int64_t r = 0; r < m; ++r) {
        const float *row = h_dets + perm[(size_t)r] * ldd;
        hf[(size_t)r] = row[0];
        memcpy(&hb[(size_t)r * 4], row + 1, 16);
        hs[(size_t)r] = row[5];
        hidx[(size_t)r] = (uint32_t)perm[(size_t)r];
    }
    std::vector<float> tb((size_t)std::max<int64_t>(t, 1) * 4), tf((size_t)std::max<int64_t>(t, 1));
    for (int64_t j = 0; j < t; ++j) {
        tf[(size_t)j] = h_tracks[j * ldt];
        memcpy(&tb[(size_t)j * 4], h_tracks + j * ldt + 1, 16);
    }
    const float t32 = thresh_to_f32(thresh);
    HIPCHK(c, c->boxes.reserve((size_t)m * 16));
    HIPCHK(c, c->scores.reserve((size_t)m * 4));
    HIPCHK(c, c->frames.reserve((size_t)m * 4));
    HIPCHK(c, c->origidx.reserve((size_t)m * 4));
    HIPCHK(c, c->excl.reserve((size_t)m));
    HIPCHK(c, c->trk_boxes.reserve(tb.size() * 4));
    HIPCHK(c, c->trk_frames.reserve(tf.size() * 4));
    HIPCHK(c, hipMemcpyAsync(c->boxes.p, hb.data(), (size_t)m * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->scores.p, hs.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->frames.p, hf.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->origidx.p, hidx.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->trk_boxes.p, tb.data(), tb.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->trk_frames.p, tf.data(), tf.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, host_sync(c));
    c->graph_valid = c->lists_valid = c->index_valid = false;     // shared scratch is overwritten below
    HostGroupsGuard guard(c);
    // build_graph clears the status word, so round 1 runs after it (inside the same stream order):
    rc = build_graph(c, c->boxes.as<float4>(), pl, t32, thresh, false);
    if (rc) return rc;
    {
        StageTimer tm(c, ST_OTHER);
        hipLaunchKernelGGL(track_round1_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->frames.as<float>(), c->boxes.as<float4>(), (int)m, c->trk_frames.as<float>(),
                           c->trk_boxes.as<float4>(), (int)t, t32, c->excl.as<uint8_t>(), &c->d_cnt->status);
    }
    HIPCHK(c, hipGetLastError());
    return nms_grouped_tail(c, pl, c->scores.as<float>(), nullptr, c->excl.as<uint8_t>(), h_keep, n_keep);
}


int vdet_track_det_nms_batch(vdet_ctx *c, const float *h_tracks, const int64_t *h_toff, int64_t ldt, const float *h_dets,
                             const int64_t *h_off, int64_t K, int64_t ldd, double thresh, int64_t *h_keep, int64_t *h_nkeep)
{
    if (!c) return VDET_EINVAL;
    if (K < 0 || (K > 0 && (!h_off || !h_nkeep))) return fail(c, VDET_EINVAL, "need K >= 0 problems with offsets and a count per problem");
    if (K == 0) return VDET_OK;
    if (K > 0x3FFFFFFF || h_off[0] != 0 || (h_toff && h_toff[0] != 0)) return fail(c, VDET_EINVAL, "offsets must start at 0");
    int64_t max_m = 0, max_t = 0;
    for (int64_t k = 0; k < K; ++k) {
        const int64_t m = h_off[k + 1] - h_off[k], t = h_toff ? h_toff[k + 1] - h_toff[k] : 1;
        if (m < 0 || t < 0) return fail(c, VDET_EINVAL, "offsets must not decrease");
        max_m = std::max(max_m, m); max_t = std::max(max_t, t);
        h_nkeep[k] = 0;
    }
    const int64_t M = h_off[K], T = h_toff ? h_toff[K] : K;
    if (M > 0x7FFFFFFF || T > 0x7FFFFFFF) return fail(c, VDET_EINVAL, "too many rows");
    if ((T > 0 && (!h_tracks || ldt < 5)) || (M > 0 && (!h_dets || !h_keep || ldd < 6)))
        return fail(c, VDET_EINVAL, "tracks must be float32 [T,5], dets float32 [M,6]");
    if (M == 0) return VDET_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (max_m <= kFusedMax && max_t <= kFusedMaxTracks && !c->no_fused) {
        timing_reset(c);
        return fused_batch_call(c, h_tracks, h_toff, ldt, h_dets, h_off, K, ldd, thresh, h_keep, h_nkeep, max_m);
    }
    // a problem beyond the single-launch sizes (or VDET_NO_FUSED=1): the problems one by one through the general chain
    for (int64_t k = 0; k < K; ++k) {
        const int64_t t0 = h_toff ? h_toff[k] : k, t = h_toff ? h_toff[k + 1] - h_toff[k] : 1;
        const int rc = vdet_track_det_nms_f32(c, h_tracks ? h_tracks + t0 * ldt : nullptr, t, ldt, h_dets + h_off[k] * ldd, h_off[k + 1] - h_off[k], ldd, thresh,
                                              h_keep + h_off[k], &h_nkeep[k]);
        if (rc) return rc;
    }
    return VDET_OK;
}

int vdet_iou_f64(vdet_ctx *c, const double *h_b1, int64_t n1, const double *h_b2, int64_t n2, double *h_out)
{
    if (!c || n1 < 0 || n2 < 0) return VDET_EINVAL;
    if (n1 == 0 || n2 == 0) return VDET_OK;
    if (!h_b1 || !h_b2 || !h_out) return fail(c, VDET_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    HIPCHK(c, c->b1.reserve((size_t)n1 * 32));
    HIPCHK(c, c->b2.reserve((size_t)n2 * 32));
    HIPCHK(c, c->iou_out.reserve((size_t)n1 * n2 * 8));
    HIPCHK(c, hipMemcpyAsync(c->b1.p, h_b1, (size_t)n1 * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b2.p, h_b2, (size_t)n2 * 32, hipMemcpyHostToDevice, c->stream));
    {
        StageTimer tm(c, ST_OTHER);
        hipLaunchKernelGGL(iou_f64_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)std::min<int64_t>(n1, 4096)),
                           dim3(256), 0, c->stream, c->b1.as<double>(), n1, c->b2.as<double>(), n2,
                           c->iou_out.as<double>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, c->iou_out.p, (size_t)n1 * n2 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
int vdet_nms_volume(vdet_ctx *c, const float *d_boxes, const float *d_scores, int layout, int64_t F, int64_t B,
                    int64_t C, double thresh, int use_score_thresh, float score_thresh, int32_t *d_keep_idx,
                    int32_t *d_keep_cnt, int64_t cap)
{
    return vdet_nms_volume_topk(c, d_boxes, d_scores, layout, F, B, C, thresh, use_score_thresh, score_thresh, 0,
                                d_keep_idx, d_keep_cnt, cap);
}

int vdet_nms_volume_topk(vdet_ctx *c, const float *d_boxes, const float *d_scores, int layout, int64_t F, int64_t B,
                         int64_t C, double thresh, int use_score_thresh, float score_thresh, int topk,
                         int32_t *d_keep_idx, int32_t *d_keep_cnt, int64_t cap)
{
    if (!c) return VDET_EINVAL;
    if (F < 0 || B < 0 || C < 0 || cap < 0 || topk < 0 || (layout != VDET_LAYOUT_FBC && layout != VDET_LAYOUT_FCB))
        return fail(c, VDET_EINVAL, "bad shape/layout");
    if (F == 0 || C == 0) return VDET_OK;
    if (!d_keep_cnt || (cap > 0 && !d_keep_idx)) return fail(c, VDET_EINVAL, "null output");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    if (((uintptr_t)d_boxes & 15) != 0) return fail(c, VDET_EINVAL, "d_boxes must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (B == 0) {
        HIPCHK(c, hipMemsetAsync(d_keep_cnt, 0, (size_t)(F * C) * 4, c->stream));
        return VDET_OK;
    }
    const float t32 = thresh_to_f32(thresh);
    const bool same_geo = c->cache_enabled && c->graph_valid && c->prep.boxes == d_boxes && c->prep.F == F &&
                          c->prep.B == B && memcmp(&c->prep.t32, &t32, 4) == 0;
    const bool same_lists = same_geo && c->lists_valid && c->prep.scores == d_scores && c->prep.C == C &&
                            c->prep.layout == layout && c->prep.use_thr == use_score_thresh &&
                            (!use_score_thresh || c->prep.thr == score_thresh) && c->prep.topk == topk;
    int rc;
    if (!same_geo) {
        c->graph_valid = c->lists_valid = false;
        rc = build_graph(c, reinterpret_cast<const float4 *>(d_boxes), volume_plan(c, F, B), t32, thresh, true);
        if (rc) return rc;
        c->prep.boxes = d_boxes; c->prep.F = F; c->prep.B = B; c->prep.t32 = t32;
        c->graph_valid = true;
    }
    SortWalkArgs a{};
    a.walk_only = same_lists;
    a.mode = layout; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
    a.scores = d_scores;
    a.use_thr = use_score_thresh; a.thr = score_thresh; a.topk = topk;
    a.keep_idx = d_keep_idx; a.keep_cnt = d_keep_cnt; a.cap = cap;
    rc = launch_sort_walk(c, a, (int)B, F * C * B);
    if (rc) return rc;
    c->prep.scores = d_scores; c->prep.C = C; c->prep.layout = layout; c->prep.use_thr = use_score_thresh;
    c->prep.thr = score_thresh; c->prep.topk = topk;
    c->lists_valid = true;
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
// The Fast R-CNN per-class flow (vdet/video_det.py:89-99 + vdet/image_det.py:117-123): every class suppresses its OWN
// regressed boxes.  Selection (score > thresh, best topk) = the LSD sort's threshold + top-k on the [F,B,K] score
// volume; then one wave per (frame, class) on its <= 128 selected boxes (detnms_kernels.hpp).
int vdet_det_nms_volume(vdet_ctx *c, const float *d_boxes, const float *d_scores, int64_t F, int64_t B, int64_t K, int class0,
                        int use_score_thresh, float score_thresh, int topk, double nms_thresh, float *d_dets, int32_t *d_sel_idx,
                        int32_t *d_det_cnt, int32_t *d_keep, int32_t *d_keep_cnt)
{
    if (!c) return VDET_EINVAL;
    if (F < 0 || B < 0 || K < 0 || class0 < 0) return fail(c, VDET_EINVAL, "bad shape");
    if (topk < 1 || topk > kDetMax) return fail(c, VDET_EINVAL, "topk = %d; the per-class NMS takes 1..%d boxes per (frame, class)", topk, kDetMax);
    if (F == 0 || K == 0) return VDET_OK;
    if (!d_det_cnt || !d_keep || !d_keep_cnt) return fail(c, VDET_EINVAL, "null output");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (F * K > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (B == 0) {
        HIPCHK(c, hipMemsetAsync(d_det_cnt, 0, (size_t)(F * K) * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d_keep_cnt, 0, (size_t)(F * K) * 4, c->stream));
        return VDET_OK;
    }
    if (!d_boxes || !d_scores) return fail(c, VDET_EINVAL, "null buffer");
    if (((uintptr_t)d_boxes & 15) != 0) return fail(c, VDET_EINVAL, "d_boxes must be 16-byte aligned");
    // only the group table (one group of B boxes per frame) is needed: no suppression graph is shared between classes here
    NmsPlan &pl = volume_plan(c, F, B);
    c->host_groups = &pl.groups;
    HIPCHK(c, c->groups.reserve((size_t)F * sizeof(GroupDesc)));
    if (!c->vplan_valid)
        HIPCHK(c, hipMemcpyAsync(c->groups.p, pl.groups.data(), (size_t)F * sizeof(GroupDesc), hipMemcpyHostToDevice, c->stream));
    c->lists_valid = false;          // (the context's lists become the selections)
    HIPCHK(c, c->nover.reserve((size_t)(F * K) * 4));
    SortWalkArgs a{};
    a.sort_only = true;
    a.mode = 0; a.P = (int)(F * K); a.B = (int)B; a.C = (int)K;
    a.scores = d_scores;
    a.use_thr = use_score_thresh ? 1 : 0; a.thr = score_thresh; a.topk = topk;
    a.nover_out = c->nover.as<int32_t>();
    const int rc = launch_sort_walk(c, a, (int)B, F * K * B);
    if (rc) return rc;
    DetNmsParams dp{};
    dp.boxes = reinterpret_cast<const float4 *>(d_boxes); dp.scores = d_scores;
    dp.F = (int)F; dp.B = (int)B; dp.K = (int)K; dp.class0 = class0;
    dp.order = c->order.as<uint16_t>(); dp.ncand = c->ncand.as<int32_t>(); dp.nover = c->nover.as<int32_t>();
    dp.topk = topk; dp.t32 = thresh_to_f32(nms_thresh);
    dp.dets = d_dets; dp.sel_idx = d_sel_idx; dp.det_cnt = d_det_cnt; dp.keep = d_keep; dp.keep_cnt = d_keep_cnt;
    dp.status = &c->d_cnt->status;
    {
        StageTimer tm(c, ST_WALK);
        hipLaunchKernelGGL(det_nms_kernel, dim3((unsigned)((F * K + 3) / 4)), dim3(256), 0, c->stream, dp);
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
int vdet_nms_volume_ordered(vdet_ctx *c, const float *d_boxes, const uint16_t *d_order, const int32_t *d_ncand, int64_t F, int64_t B,
                            int64_t C, double thresh, int32_t *d_keep_idx, int32_t *d_keep_cnt, int64_t cap)
{
    if (!c) return VDET_EINVAL;
    if (F < 0 || B < 0 || C < 0 || cap < 0) return fail(c, VDET_EINVAL, "bad shape");
    if (F == 0 || C == 0) return VDET_OK;
    if (!d_keep_cnt || (cap > 0 && !d_keep_idx)) return fail(c, VDET_EINVAL, "null output");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (B == 0) {
        HIPCHK(c, hipMemsetAsync(d_keep_cnt, 0, (size_t)(F * C) * 4, c->stream));
        return VDET_OK;
    }
    if (!d_boxes || !d_order || !d_ncand) return fail(c, VDET_EINVAL, "null buffer");
    if (((uintptr_t)d_boxes & 15) != 0) return fail(c, VDET_EINVAL, "d_boxes must be 16-byte aligned");
    const float t32 = thresh_to_f32(thresh);
    const bool same_geo = c->cache_enabled && c->graph_valid && c->prep.boxes == d_boxes && c->prep.F == F &&
                          c->prep.B == B && memcmp(&c->prep.t32, &t32, 4) == 0;
    if (!same_geo) {
        c->graph_valid = c->lists_valid = false;
        const int rc = build_graph(c, reinterpret_cast<const float4 *>(d_boxes), volume_plan(c, F, B), t32, thresh, true);
        if (rc) return rc;
        c->prep.boxes = d_boxes; c->prep.F = F; c->prep.B = B; c->prep.t32 = t32;
        c->graph_valid = true;
    }
    // the lists are the caller's: counts / indices out of range latch VDET_EINVAL (vdet_sync) and the list is walked as empty
    HIPCHK(c, c->ordncand.reserve((size_t)(F * C) * 4));
    hipLaunchKernelGGL(check_order_kernel, dim3((unsigned)((F * C + 3) / 4)), dim3(256), 0, c->stream, d_order, d_ncand, (int)(F * C), (int)B,
                       c->ordncand.as<int32_t>(), &c->d_cnt->status);
    SortWalkArgs a{};
    a.walk_only = true;
    a.mode = 1; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
    a.order_in = d_order; a.ncand_in = c->ordncand.as<int32_t>();
    a.keep_idx = d_keep_idx; a.keep_cnt = d_keep_cnt; a.cap = cap;
    return launch_sort_walk(c, a, (int)B, F * C * B);
}

// ---------------------------------------------------------------------------------------------
int vdet_argsort_volume(vdet_ctx *c, const float *d_scores, int layout, int64_t F, int64_t B, int64_t C,
                        int use_score_thresh, float score_thresh, uint16_t *d_order, int32_t *d_ncand)
{
    if (!c) return VDET_EINVAL;
    if (F < 0 || B < 0 || C < 0 || (layout != VDET_LAYOUT_FBC && layout != VDET_LAYOUT_FCB)) return fail(c, VDET_EINVAL, "bad shape/layout");
    if (F == 0 || C == 0) return VDET_OK;
    if (!d_ncand || (B > 0 && (!d_scores || !d_order))) return fail(c, VDET_EINVAL, "null buffer");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (B == 0) {
        HIPCHK(c, hipMemsetAsync(d_ncand, 0, (size_t)(F * C) * 4, c->stream));
        return VDET_OK;
    }
    // only the group table (one group of B boxes per frame) is needed; tiles / pairs follow with the next graph build
    NmsPlan &pl = volume_plan(c, F, B);
    c->host_groups = &pl.groups;
    HIPCHK(c, c->groups.reserve((size_t)F * sizeof(GroupDesc)));
    if (!c->vplan_valid)
        HIPCHK(c, hipMemcpyAsync(c->groups.p, pl.groups.data(), (size_t)F * sizeof(GroupDesc), hipMemcpyHostToDevice, c->stream));
    c->lists_valid = false;          // (c->tkeys, which the context's own lists are read with, is rewritten)
    SortWalkArgs a{};
    a.sort_only = true;
    a.mode = layout; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
    a.scores = d_scores;
    a.use_thr = use_score_thresh; a.thr = score_thresh;
    a.order_out = d_order; a.ncand_out = d_ncand;
    return launch_sort_walk(c, a, (int)B, F * C * B);
}

// ---------------------------------------------------------------------------------------------
int vdet_track_volume(vdet_ctx *c, const float *d_boxes, const float *d_scores, int64_t F, int64_t B, int64_t C,
                      double nms_thres, double thres, int max_tracks, double link_thres, int max_frames,
                      float *d_tracks, float *d_anchors, int32_t *d_ntracks)
{
    return vdet_nms_track_volume(c, d_boxes, d_scores, F, B, C, nms_thres, thres, max_tracks, link_thres, max_frames,
                                 d_tracks, d_anchors, d_ntracks, 0, nullptr, nullptr);
}

int vdet_nms_track_volume(vdet_ctx *c, const float *d_boxes, const float *d_scores, int64_t F, int64_t B, int64_t C,
                          double nms_thres, double thres, int max_tracks, double link_thres, int max_frames,
                          float *d_tracks, float *d_anchors, int32_t *d_ntracks, int64_t cap, int32_t *d_keep_idx,
                          int32_t *d_keep_cnt)
{
    if (!c) return VDET_EINVAL;
    const bool want_nms = d_keep_cnt != nullptr;
    if (want_nms && (cap < 0 || (cap > 0 && !d_keep_idx))) return fail(c, VDET_EINVAL, "null output");
    if (F <= 0 || B <= 0 || C <= 0 || max_tracks < 0) return fail(c, VDET_EINVAL, "bad shape");
    if (!d_boxes || !d_scores || !d_ntracks || (max_tracks > 0 && (!d_tracks || !d_anchors)))
        return fail(c, VDET_EINVAL, "null buffer");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    if (((uintptr_t)d_boxes & 15) != 0) return fail(c, VDET_EINVAL, "d_boxes must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    const float t32 = thresh_to_f32(nms_thres);
    const bool same_geo = c->cache_enabled && c->graph_valid && c->prep.boxes == d_boxes && c->prep.F == F &&
                          c->prep.B == B && memcmp(&c->prep.t32, &t32, 4) == 0;
    const bool same_lists = same_geo && c->lists_valid && c->prep.scores == d_scores && c->prep.C == C &&
                            c->prep.layout == VDET_LAYOUT_FBC && c->prep.use_thr == 0 && c->prep.topk == 0;
    int rc;
    if (!same_geo) {
        c->graph_valid = c->lists_valid = false;
        rc = build_graph(c, reinterpret_cast<const float4 *>(d_boxes), volume_plan(c, F, B), t32, nms_thres, true);
        if (rc) return rc;
        c->prep.boxes = d_boxes; c->prep.F = F; c->prep.B = B; c->prep.t32 = t32;
        c->graph_valid = true;
    }
    if (!same_lists) {
        // descending lists per (frame, class): always through the transposed keys (pick needs them)
        SortWalkArgs a{};
        a.sort_only = true;
        a.mode = 0; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
        a.scores = d_scores;
        rc = launch_sort_walk(c, a, (int)B, F * C * B);
        if (rc) return rc;
    }
    // regular-frame fast paths (lazy lists, x-window link): only when THIS graph build (or the cached
    // one being reused) ran K0 + the frame index -- gflags / the index are stale otherwise
    const bool regular_ok = c->sym_built;
    const float link_t32 = thresh_to_f32(link_thres);
    const int reach = max_frames > 0 ? (int)std::ceil((max_frames + 1) / 2.0) - 1 : (int)F;
    const uint32_t *w_flags = regular_ok ? c->gflags.as<uint32_t>() : nullptr;
    FrameIndex w_ix{nullptr, nullptr, nullptr, nullptr};
    if (w_flags && c->index_valid && !c->no_index) w_ix = frame_index_of(c);
    c->nodes_valid = false;
    HIPCHK(c, c->tracknode.reserve((size_t)std::max<int64_t>(C * max_tracks * F, 1) * 4));
    HIPCHK(c, hipMemsetAsync(c->tracknode.p, 0xFF, (size_t)std::max<int64_t>(C * max_tracks * F, 1) * 4, c->stream));
    int materialized = 0;            // warm chains per class whose tubelets are written out (0: none)
    // one link memo per call: a link step depends on the video's boxes and link_thres only
    HIPCHK(c, c->linkmemo.reserve((size_t)2 * F * B * 8));
    HIPCHK(c, c->linkstats.reserve(16));
    HIPCHK(c, hipMemsetAsync(c->linkmemo.p, 0, (size_t)2 * F * B * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->linkstats.p, 0, 16, c->stream));
    if (max_tracks > 0) {
        // small frames: the whole link table up front (every node's window scan, chip-filling) -- then no step of any chain
        // is ever scanned again, whatever the anchors turn out to be
        const bool filled = B <= kLinkFillMax && w_ix.xbox != nullptr;
        // warm the memo otherwise: the chains of every class's likely anchors, all at once (the chip is full instead of
        // running 2 C latency-bound blocks per track); the tracking loop below then mostly walks known steps.  + slots for
        // the anchors of COHERENT videos (track_warm_anchors_body: filled only when a class's raw candidates repeat each
        // other's objects across frames; empty -- and free -- otherwise)
        const int wm_raw = std::min(max_tracks + 6, 24);       // (measured: 16 of 10 tracks)
        const bool coherent_slots = F <= 512 && regular_ok && !filled;
        const int wm = coherent_slots ? std::min(wm_raw + max_tracks, 32) : wm_raw;
        StageTimer tm(c, ST_TLINK);
        if (filled)
            hipLaunchKernelGGL(link_fill_frame_kernel, dim3((unsigned)F, 2), dim3((unsigned)(64 * ((B + 63) / 64))), link_fill_lds_bytes((int)B), c->stream,
                               reinterpret_cast<const float4 *>(d_boxes), (int)F, (int)B, link_t32, w_flags, w_ix, link_thres,
                               c->linkmemo.as<unsigned long long>());
        HIPCHK(c, c->linkwarm.reserve((size_t)C * wm * 4));
        hipLaunchKernelGGL(track_warm_anchors_kernel, dim3((unsigned)C), dim3(256), 0, c->stream, c->tkeys.as<uint32_t>(),
                           c->order.as<uint16_t>(), c->ncand.as<int32_t>(), (int)F, (int)B, (int)C, d_scores, thres, wm,
                           c->linkwarm.as<int32_t>(),
                           WarmExtra{coherent_slots ? reinterpret_cast<const float4 *>(d_boxes) : nullptr, t32, wm_raw, max_tracks});
        if (!filled) {
            // longest chains first, then every (chain, direction) as one block of the warm-up launch
            HIPCHK(c, c->linkorder.reserve((size_t)C * wm * 2 * 4));
            hipLaunchKernelGGL(warm_order_kernel, dim3(1), dim3(1024), 0, c->stream, c->linkwarm.as<int32_t>(), (int)(C * wm), (int)F, (int)B,
                               reach, c->linkorder.as<int32_t>());
            hipLaunchKernelGGL((track_link_memo_kernel<256, 1, 8>), dim3((unsigned)(C * wm), 2), dim3(256), 0, c->stream,
                               reinterpret_cast<const float4 *>(d_boxes), (int)F, (int)B, max_tracks, link_t32, reach,
                               (const TrackState *)nullptr, (float *)nullptr, w_flags, w_ix, link_thres,
                               c->linkmemo.as<unsigned long long>(), c->linkstats.as<unsigned int>(), c->linkwarm.as<int32_t>(),
                               (int32_t *)nullptr, (const int32_t *)c->linkorder.as<int32_t>());
        }
        // every step of the warm chains is known now: write each predicted anchor's tubelet ONCE (one wave walks a chain; all
        // of them side by side), for the tracking loop to copy
        HIPCHK(c, c->linkchains.reserve((size_t)C * wm * F * 5 * 4));
        HIPCHK(c, c->linknodes.reserve((size_t)C * wm * F * 4));
        HIPCHK(c, hipMemsetAsync(c->linknodes.p, 0xFF, (size_t)C * wm * F * 4, c->stream));
        hipLaunchKernelGGL((track_link_memo_kernel<64, 2, 8>), dim3((unsigned)(C * wm), 2), dim3(64), 0, c->stream,
                           reinterpret_cast<const float4 *>(d_boxes), (int)F, (int)B, max_tracks, link_t32, reach,
                           (const TrackState *)nullptr, c->linkchains.as<float>(), w_flags, w_ix, link_thres,
                           c->linkmemo.as<unsigned long long>(), (unsigned int *)nullptr, c->linkwarm.as<int32_t>(),
                           c->linknodes.as<int32_t>(), (const int32_t *)nullptr);
        materialized = wm;
    }
    if (want_nms) {                  // the NMS survivors: one walk over the lists, before they are consumed
        SortWalkArgs a{};
        a.walk_only = true;
        a.mode = 0; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
        a.scores = d_scores;
        a.keep_idx = d_keep_idx; a.keep_cnt = d_keep_cnt; a.cap = cap;
        rc = launch_sort_walk(c, a, (int)B, F * C * B);
        if (rc) return rc;
    }
    c->lists_valid = false;          // the lists are consumed (compacted in place) below
    HIPCHK(c, c->tstate.reserve((size_t)C * sizeof(TrackState)));
    TrackState *st = c->tstate.as<TrackState>();
    const unsigned cg = (unsigned)((C + 63) / 64);
    hipLaunchKernelGGL(track_init_kernel, dim3(cg), dim3(64), 0, c->stream, st, (int)C);
    HIPCHK(c, hipMemsetAsync(d_ntracks, 0, (size_t)C * 4, c->stream));
    if (max_tracks == 0) {          // nothing to track: report a launch error of the walk / init above by THIS call
        HIPCHK(c, hipGetLastError());
        return VDET_OK;
    }
    SuppressParams sp{};
    sp.boxes = reinterpret_cast<const float4 *>(d_boxes);
    sp.F = (int)F; sp.B = (int)B; sp.C = (int)C; sp.max_tracks = max_tracks;
    sp.groups = c->groups.as<GroupDesc>();
    sp.row_meta = c->rowmeta.as<uint2>();
    sp.adj = c->adj.as<uint16_t>();
    sp.group_z = c->groupz.as<uint32_t>();
    sp.group_flags = regular_ok ? c->gflags.as<uint32_t>() : nullptr;
    sp.ix = FrameIndex{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (sp.group_flags && c->index_valid && !c->no_index) sp.ix = frame_index_of(c);   // built by build_graph
    sp.thres = nms_thres;
    sp.lists = c->order.as<uint16_t>();
    sp.cnt = c->ncand.as<int32_t>();
    HIPCHK(c, c->visited.reserve((size_t)(F * C)));
    HIPCHK(c, hipMemsetAsync(c->visited.p, 0, (size_t)(F * C), c->stream));
    sp.visited = c->visited.as<uint8_t>();
    sp.st = st;
    sp.tracks = d_tracks;
    sp.t32 = t32;
    sp.status = &c->d_cnt->status;
    sp.mask_words = (int)((((size_t)4 * ((B + 31) / 32) + 15) & ~(size_t)15) / 4);
    sp.lazy = c->no_lazy ? 0 : 1;
    // lazy-list state of the pick: t1 | head | nkp | pos, [F*C] int32 each
    HIPCHK(c, c->heads.reserve((size_t)(F * C) * 16));
    HIPCHK(c, hipMemsetAsync(c->heads.p, 0, (size_t)(F * C) * 16, c->stream));
    LazyLists lz{};
    lz.boxes = sp.boxes; lz.tracks = d_tracks; lz.t32 = t32;
    lz.t1 = c->heads.as<int32_t>(); lz.head = lz.t1 + F * C; lz.nkp = lz.head + F * C; lz.pos = lz.nkp + F * C;
    lz.group_flags = sp.lazy ? sp.group_flags : nullptr;
    // the eager track_det_nms pass is only needed for the lists the pick does not maintain; when the
    // host does not know whether every frame is regular (asynchronous build) the kernel asks the device
    const bool need_suppress = !sp.lazy || !sp.group_flags || !c->all_regular;
    sp.n_irregular = &c->d_cnt->irregular;
    // predicted anchors: their tubelets exist already, the loop copies them
    ResolveArgs rv{c->linkwarm.as<int32_t>(), materialized, c->linkchains.as<float>(), c->linknodes.as<int32_t>(), d_tracks,
                   c->tracknode.as<int32_t>()};
    {
        // the whole tracking loop in one launch: one persistent block per class (track_loop_kernel)
        LoopArgs la{};
        la.keys = c->tkeys.as<uint32_t>(); la.lists = c->order.as<uint16_t>(); la.cnt = c->ncand.as<int32_t>();
        la.scores = d_scores; la.thres = thres; la.link_thres = link_thres; la.anchors = d_anchors;
        la.link_t32 = link_t32; la.reach = reach;
        la.memo = c->linkmemo.as<unsigned long long>(); la.stats = c->linkstats.as<unsigned int>();
        la.nodes = c->tracknode.as<int32_t>(); la.ntracks_out = d_ntracks;
        la.need_suppress = need_suppress ? 1 : 0;
        StageTimer tm(c, ST_TLOOP);
        hipLaunchKernelGGL(track_loop_kernel, dim3((unsigned)C), dim3(256), (size_t)sp.mask_words * 16, c->stream, la, lz, rv, sp);
    }
    HIPCHK(c, hipGetLastError());
    c->nodekey.tracks = d_tracks; c->nodekey.boxes = d_boxes; c->nodekey.F = F; c->nodekey.B = B; c->nodekey.C = C;
    c->nodekey.T = max_tracks; c->nodekey.nms_thres = nms_thres;
    c->nodes_valid = true;
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
// V small videos in one call (BASELINE configs[0] / [4] shapes: hundreds of frames x <= 300 boxes x 30 classes, where
// one video alone is launch-bound at ~60 launches): the frames of all videos are concatenated along F -- per-frame NMS
// problems do not know which video they belong to, so suppression graph, sort and walk are ONE launch sequence for the
// whole batch -- and only the stages that follow a video in time (tracking, re-scoring) run per video, on sub-ranges
// of the same buffers, with the video as a grid dimension (batch_kernels.hpp): one launch per stage for all videos instead of
// ~60 launches per video, no host synchronisation in between.
// ---------------------------------------------------------------------------------------------
int vdet_video_batch(vdet_ctx *c, const float *d_boxes, const float *d_scores, const int64_t *h_frame_off, int64_t V, int64_t B,
                     int64_t C, double nms_thres, double thres, int max_tracks, double link_thres, int max_frames,
                     float *d_tracks, float *d_anchors, int32_t *d_ntracks, int64_t cap, int32_t *d_keep_idx, int32_t *d_keep_cnt,
                     double overlap_thres, int window, double *d_det_score, double *d_pooled, float *d_boxes_out)
{
    if (!c) return VDET_EINVAL;
    const bool want_nms = d_keep_cnt != nullptr, want_rescore = d_pooled != nullptr;
    if (want_nms && (cap < 0 || (cap > 0 && !d_keep_idx))) return fail(c, VDET_EINVAL, "null output");
    if (V <= 0 || B <= 0 || C <= 0 || max_tracks < 0 || !h_frame_off) return fail(c, VDET_EINVAL, "bad shape");
    if (!d_boxes || !d_scores || !d_ntracks || (max_tracks > 0 && (!d_tracks || !d_anchors))) return fail(c, VDET_EINVAL, "null buffer");
    if (want_rescore && (!d_det_score || !d_boxes_out)) return fail(c, VDET_EINVAL, "null buffer");
    if (want_rescore && (window < 1 || window % 2 != 1)) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (B > 32767) return fail(c, VDET_EINVAL, "B = %lld boxes per frame; the limit is 32767", (long long)B);
    if (((uintptr_t)d_boxes & 15) != 0) return fail(c, VDET_EINVAL, "d_boxes must be 16-byte aligned");
    if (h_frame_off[0] != 0) return fail(c, VDET_EINVAL, "frame offsets must start at 0");
    int64_t Fmax = 0;
    for (int64_t v = 0; v < V; ++v) {
        if (h_frame_off[v + 1] <= h_frame_off[v]) return fail(c, VDET_EINVAL, "every video needs at least one frame");
        Fmax = std::max(Fmax, h_frame_off[v + 1] - h_frame_off[v]);
    }
    const int64_t F = h_frame_off[V];
    if (F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll || V * C > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    const float t32 = thresh_to_f32(nms_thres);
    int rc;
    // ---- the batch as one volume of F frames: graph, sorted lists, NMS survivors
    c->graph_valid = c->lists_valid = c->nodes_valid = false;
    rc = build_graph(c, reinterpret_cast<const float4 *>(d_boxes), volume_plan(c, F, B), t32, nms_thres, true);
    if (rc) return rc;
    c->prep.boxes = d_boxes; c->prep.F = F; c->prep.B = B; c->prep.t32 = t32;
    c->graph_valid = true;
    {
        SortWalkArgs a{};
        a.sort_only = true;
        a.mode = 0; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
        a.scores = d_scores;
        rc = launch_sort_walk(c, a, (int)B, F * C * B);
        if (rc) return rc;
    }
    if (want_nms) {
        SortWalkArgs a{};
        a.walk_only = true;
        a.mode = 0; a.P = (int)(F * C); a.B = (int)B; a.C = (int)C;
        a.scores = d_scores;
        a.keep_idx = d_keep_idx; a.keep_cnt = d_keep_cnt; a.cap = cap;
        rc = launch_sort_walk(c, a, (int)B, F * C * B);
        if (rc) return rc;
    }
    c->lists_valid = false;          // consumed by the tracking below
    HIPCHK(c, hipMemsetAsync(d_ntracks, 0, (size_t)(V * C) * 4, c->stream));
    if (max_tracks == 0) {
        HIPCHK(c, hipGetLastError());
        return VDET_OK;
    }
    // ---- per video: tracking (+ re-scoring) on its frame range
    const bool regular_ok = c->sym_built;
    const float link_t32 = thresh_to_f32(link_thres);
    const uint32_t *g_flags = regular_ok ? c->gflags.as<uint32_t>() : nullptr;
    const bool have_ix = g_flags && c->index_valid && !c->no_index;
    const int T = max_tracks;
    const int wm = std::min(T + 6, 24);
    HIPCHK(c, c->tracknode.reserve((size_t)(C * T * F) * 4));
    HIPCHK(c, hipMemsetAsync(c->tracknode.p, 0xFF, (size_t)(C * T * F) * 4, c->stream));
    HIPCHK(c, c->linkmemo.reserve((size_t)2 * F * B * 8));
    HIPCHK(c, c->linkstats.reserve(16));
    HIPCHK(c, hipMemsetAsync(c->linkmemo.p, 0, (size_t)2 * F * B * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->linkstats.p, 0, 16, c->stream));
    HIPCHK(c, c->linkwarm.reserve((size_t)V * C * wm * 4));
    HIPCHK(c, c->linkchains.reserve((size_t)C * wm * F * 5 * 4));
    HIPCHK(c, c->linknodes.reserve((size_t)C * wm * F * 4));
    HIPCHK(c, hipMemsetAsync(c->linknodes.p, 0xFF, (size_t)C * wm * F * 4, c->stream));
    HIPCHK(c, c->tstate.reserve((size_t)(V * C) * sizeof(TrackState)));
    hipLaunchKernelGGL(track_init_kernel, dim3((unsigned)((V * C + 63) / 64)), dim3(64), 0, c->stream, c->tstate.as<TrackState>(), (int)(V * C));
    HIPCHK(c, c->visited.reserve((size_t)(F * C)));
    HIPCHK(c, hipMemsetAsync(c->visited.p, 0, (size_t)(F * C), c->stream));
    HIPCHK(c, c->heads.reserve((size_t)(F * C) * 16));
    HIPCHK(c, hipMemsetAsync(c->heads.p, 0, (size_t)(F * C) * 16, c->stream));
    const bool need_suppress = c->no_lazy || !g_flags || !c->all_regular;
    const int mask_words = (int)((((size_t)4 * ((B + 31) / 32) + 15) & ~(size_t)15) / 4);
    // ---- all videos side by side: one launch per stage, the video is a grid dimension (batch_kernels.hpp)
    {
        // the {first frame, frames} table: uploaded only when it differs from the resident one (the host copy must not be
        // rewritten under a copy in flight, so a change waits for the stream once; the same offsets again cost nothing)
        bool same_tab = c->vidtab.p != nullptr && c->h_vids.size() == (size_t)V;
        for (int64_t v = 0; same_tab && v < V; ++v)
            same_tab = c->h_vids[(size_t)v].f0 == (int32_t)h_frame_off[v] && c->h_vids[(size_t)v].F == (int32_t)(h_frame_off[v + 1] - h_frame_off[v]);
        if (!same_tab) {
            (void)host_sync(c);
            c->h_vids.clear();           // (committed below, once the copy is enqueued: a failure must not leave a "resident" table)
            std::vector<VidDesc> tab((size_t)V);
            for (int64_t v = 0; v < V; ++v) tab[(size_t)v] = VidDesc{(int32_t)h_frame_off[v], (int32_t)(h_frame_off[v + 1] - h_frame_off[v])};
            HIPCHK(c, c->vidtab.reserve((size_t)V * sizeof(VidDesc)));
            c->h_vids_stage.swap(tab);   // the copy's source must outlive it
            HIPCHK(c, hipMemcpyAsync(c->vidtab.p, c->h_vids_stage.data(), (size_t)V * sizeof(VidDesc), hipMemcpyHostToDevice, c->stream));
            c->h_vids = c->h_vids_stage;
        }
    }
    BatchTrack bt{};
    bt.vids = c->vidtab.as<VidDesc>();
    bt.Ftot = (int)F; bt.B = (int)B; bt.C = (int)C; bt.T = T; bt.wm = wm;
    bt.boxes = reinterpret_cast<const float4 *>(d_boxes); bt.scores = d_scores;
    bt.keys = c->tkeys.as<uint32_t>(); bt.lists = c->order.as<uint16_t>(); bt.cnt = c->ncand.as<int32_t>();
    bt.group_flags = g_flags;
    bt.ix = have_ix ? frame_index_of(c) : FrameIndex{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    bt.memo = c->linkmemo.as<unsigned long long>(); bt.stats = c->linkstats.as<unsigned int>();
    bt.warm = c->linkwarm.as<int32_t>(); bt.chains = c->linkchains.as<float>(); bt.chain_nodes = c->linknodes.as<int32_t>();
    bt.track_nodes = c->tracknode.as<int32_t>();
    bt.tracks = d_tracks; bt.anchors = d_anchors; bt.ntracks = d_ntracks;
    bt.st = c->tstate.as<TrackState>();
    bt.heads = c->heads.as<int32_t>(); bt.visited = c->visited.as<uint8_t>();
    bt.groups = c->groups.as<GroupDesc>(); bt.row_meta = c->rowmeta.as<uint2>(); bt.adj = c->adj.as<uint16_t>();
    bt.group_z = c->groupz.as<uint32_t>();
    bt.thres = thres; bt.link_thres = link_thres; bt.nms_thres = nms_thres;
    bt.link_t32 = link_t32; bt.t32 = t32;
    bt.reach_all = max_frames > 0 ? (int)std::ceil((max_frames + 1) / 2.0) - 1 : -1;
    bt.need_suppress = need_suppress ? 1 : 0; bt.lazy = c->no_lazy ? 0 : 1; bt.mask_words = mask_words;
    bt.status = &c->d_cnt->status; bt.n_irregular = &c->d_cnt->irregular;
    {
        StageTimer tm(c, ST_TLINK);
        if (B <= kLinkFillMax && have_ix) {
            // the whole link table of every video: no chain ever scans, whatever its anchor -- so nothing is predicted or
            // materialised either
            hipLaunchKernelGGL(batch_link_fill_frame_kernel, dim3((unsigned)Fmax, 2, (unsigned)V), dim3((unsigned)(64 * ((B + 63) / 64))),
                               link_fill_lds_bytes((int)B), c->stream, bt);
            bt.wm = 0;
        } else {
            hipLaunchKernelGGL(batch_warm_anchors_kernel, dim3((unsigned)C, (unsigned)V), dim3(256), 0, c->stream, bt);
            hipLaunchKernelGGL((batch_link_kernel<256, 1>), dim3((unsigned)(C * wm), 2, (unsigned)V), dim3(256), 0, c->stream, bt);
            hipLaunchKernelGGL((batch_link_kernel<64, 2>), dim3((unsigned)(C * wm), 2, (unsigned)V), dim3(64), 0, c->stream, bt);
        }
    }
    {
        StageTimer tm(c, ST_TLOOP);
        hipLaunchKernelGGL(batch_loop_kernel, dim3((unsigned)C, (unsigned)V), dim3(256), (size_t)mask_words * 16, c->stream, bt);
    }
    if (want_rescore) {
        {
            StageTimer tm(c, ST_RSPATIAL);
            // candidates from the suppression graph wherever the tracking loop recorded the proposal a tubelet box came from (as
            // vdet_rescore_tracks does; needs overlap_thres well above the graph's threshold), the window scan for the rest
            const bool use_adj = g_flags && overlap_thres - nms_thres > 0.05 && nms_thres > 0.0 && overlap_thres < 1.0;
            if (use_adj) {
                const int64_t nb = F * C * T;
                HIPCHK(c, c->rtodo.reserve((size_t)nb * 8 + 16));
                unsigned int *cnt = reinterpret_cast<unsigned int *>(c->rtodo.as<char>() + (size_t)nb * 8);
                HIPCHK(c, hipMemsetAsync(cnt, 0, 4, c->stream));
                hipLaunchKernelGGL(batch_rescore_adj_kernel, dim3((unsigned)((Fmax * C * T + 15) / 16), (unsigned)V), dim3(256), 0, c->stream, bt,
                                   overlap_thres, 1.0 - (overlap_thres - nms_thres) + 0.02, d_det_score, d_boxes_out, c->rtodo.as<int2>(), cnt);
                hipLaunchKernelGGL(batch_rescore_todo_kernel, dim3((unsigned)std::min<int64_t>((nb + 3) / 4, 8 * c->n_cu)), dim3(256), 0, c->stream, bt,
                                   overlap_thres, d_det_score, d_boxes_out, c->rtodo.as<int2>(), cnt);
            } else
            hipLaunchKernelGGL(batch_rescore_spatial_kernel, dim3((unsigned)((Fmax * C * T + 3) / 4), (unsigned)V), dim3(256), 0, c->stream, bt,
                               overlap_thres, d_det_score, d_boxes_out);
        }
        StageTimer tm(c, ST_RSERIES);
        // one wave per series with the series in LDS; videos too long for that stage take the one-thread-per-series kernel
        const int stride = (int)(((size_t)std::min<int64_t>(Fmax, kSeriesWaveMaxF) * 9 + 15) & ~(size_t)15);
        hipLaunchKernelGGL(batch_rescore_series_kernel, dim3((unsigned)((C * T + 3) / 4), (unsigned)V), dim3(256), (size_t)stride * 4, c->stream, bt,
                           d_det_score, d_pooled, window, &c->d_cnt->eindex, stride);
        if (Fmax > kSeriesWaveMaxF)
            hipLaunchKernelGGL(batch_rescore_series_long_kernel, dim3((unsigned)((C * T + 63) / 64), (unsigned)V), dim3(64), 0, c->stream, bt,
                               d_det_score, d_pooled, window, &c->d_cnt->eindex);
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

int vdet_rescore_tracks(vdet_ctx *c, const float *d_tracks, const int32_t *d_ntracks, const float *d_boxes,
                        const float *d_scores, int64_t F, int64_t B, int64_t C, int max_tracks, double overlap_thres,
                        int window, double *d_det_score, double *d_pooled, float *d_boxes_out)
{
    if (!c) return VDET_EINVAL;
    if (F <= 0 || B <= 0 || C <= 0 || max_tracks < 0) return fail(c, VDET_EINVAL, "bad shape");
    if (window < 1 || window % 2 != 1) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (max_tracks == 0) return VDET_OK;
    if (!d_tracks || !d_ntracks || !d_boxes || !d_scores || !d_det_score || !d_pooled || !d_boxes_out)
        return fail(c, VDET_EINVAL, "null buffer");
    const int64_t nb = C * max_tracks * F;
    if (nb > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "too many tubelet boxes");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    FrameIndex ix{nullptr, nullptr, nullptr, nullptr};
    const uint32_t *flags = nullptr;
    if (!c->no_index && !c->force_general && (size_t)8 * B + 24 * 1024 <= c->max_lds) {   // (the x1 sort must fit the LDS)
        // per-frame regular flags + x-sorted index: reused from the graph build of the same boxes
        // when the cache is on, else rebuilt here (cheap: one 3 M-key sort)
        const bool have = c->cache_enabled && c->graph_valid && c->index_valid && c->prep.boxes == d_boxes &&
                          c->prep.F == F && c->prep.B == B && c->index_boxes == d_boxes;
        if (!have) {
            c->graph_valid = c->lists_valid = c->index_valid = false;       // gflags / the index are rewritten
            NmsPlan &pl = volume_plan(c, F, B);
            c->host_groups = &pl.groups;
            HIPCHK(c, c->groups.reserve((size_t)F * sizeof(GroupDesc)));
            HIPCHK(c, c->gflags.reserve((size_t)F * 4));
            if (!c->vplan_valid) {
                // (only the group table is needed here; tiles / pairs follow with the next graph build)
                HIPCHK(c, hipMemcpyAsync(c->groups.p, pl.groups.data(), (size_t)F * sizeof(GroupDesc), hipMemcpyHostToDevice, c->stream));
            }
            c->sym_built = false;
            hipLaunchKernelGGL(frame_flags_kernel, dim3((unsigned)F), dim3(256), 0, c->stream,
                               reinterpret_cast<const float4 *>(d_boxes), c->groups.as<GroupDesc>(), c->gflags.as<uint32_t>(),
                               &c->d_cnt->irregular);
            const int rc = build_frame_index(c, reinterpret_cast<const float4 *>(d_boxes), F * B, F, (int)B);
            if (rc) return rc;
        }
        ix = frame_index_of(c);
        flags = c->gflags.as<uint32_t>();
    }
    {
        StageTimer tm(c, ST_RSPATIAL);
        // the tracks of the last tracking call on this context, same boxes, graph still in place (cache contract):
        // candidates from the suppression graph (see the kernel); needs overlap_thres well above the graph's threshold
        const bool use_adj = flags && c->cache_enabled && c->nodes_valid && c->graph_valid &&
                             c->nodekey.tracks == d_tracks && c->nodekey.boxes == d_boxes && c->prep.boxes == d_boxes &&
                             c->nodekey.F == F && c->nodekey.B == B && c->nodekey.C == C && c->nodekey.T == max_tracks &&
                             c->prep.F == F && c->prep.B == B && overlap_thres - c->nodekey.nms_thres > 0.05 &&
                             c->nodekey.nms_thres > 0.0 && overlap_thres < 1.0 &&
                             // ... and the resident graph is the one the tracks were made on (same threshold)
                             [&] { const float nt = thresh_to_f32(c->nodekey.nms_thres); return memcmp(&c->prep.t32, &nt, 4) == 0; }();
        const double min_self = use_adj ? 1.0 - (overlap_thres - c->nodekey.nms_thres) + 0.02 : 2.0;
        const int32_t *todo = nullptr;
        const unsigned int *todo_cnt = nullptr;
        unsigned scan_grid = (unsigned)((nb + 3) / 4);
        if (use_adj) {
            HIPCHK(c, c->rtodo.reserve((size_t)nb * 4 + 16));
            unsigned int *cnt = reinterpret_cast<unsigned int *>(c->rtodo.as<char>() + (size_t)nb * 4);
            HIPCHK(c, hipMemsetAsync(cnt, 0, 4, c->stream));
            hipLaunchKernelGGL(rescore_adj_kernel, dim3((unsigned)((((nb + 15) / 16) + 7) & ~(int64_t)7)), dim3(256), 0, c->stream, d_tracks, d_ntracks,
                               reinterpret_cast<const float4 *>(d_boxes), d_scores, (int)F, (int)B, (int)C, max_tracks,
                               overlap_thres, d_det_score, d_boxes_out, flags, c->tracknode.as<int32_t>(), c->rowmeta.as<uint2>(),
                               c->adj.as<uint16_t>(), min_self, c->rtodo.as<int32_t>(), cnt);
            todo = c->rtodo.as<int32_t>();
            todo_cnt = cnt;
            scan_grid = (unsigned)std::min<int64_t>((nb + 3) / 4, 8 * c->n_cu);
        }
        if (todo)
            hipLaunchKernelGGL(rescore_spatial_kernel<true>, dim3(scan_grid), dim3(256), 0, c->stream, d_tracks, d_ntracks,
                               reinterpret_cast<const float4 *>(d_boxes), d_scores, (int)F, (int)B, (int)C, max_tracks,
                               overlap_thres, d_det_score, d_boxes_out, ix, flags, todo, todo_cnt);
        else
            hipLaunchKernelGGL(rescore_spatial_kernel<false>, dim3(scan_grid), dim3(256), 0, c->stream, d_tracks, d_ntracks,
                               reinterpret_cast<const float4 *>(d_boxes), d_scores, (int)F, (int)B, (int)C, max_tracks,
                               overlap_thres, d_det_score, d_boxes_out, ix, flags, todo, todo_cnt);
    }
    {
        StageTimer tm(c, ST_RSERIES);
        const int n = (int)(C * max_tracks);
        if (F <= kSeriesWaveMaxF) {      // one wave per series, the series in LDS
            const int stride = (int)(((size_t)F * 9 + 15) & ~(size_t)15);
            hipLaunchKernelGGL(rescore_series_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)stride * 4, c->stream,
                               d_det_score, d_pooled, d_ntracks, (int)F, (int)C, max_tracks, window, &c->d_cnt->eindex, stride);
        } else {
            hipLaunchKernelGGL(rescore_series_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, d_det_score,
                               d_pooled, d_ntracks, (int)F, (int)C, max_tracks, window, &c->d_cnt->eindex);
        }
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
// tubelet re-scoring cores (host buffers)
// ---------------------------------------------------------------------------------------------
static int upload(vdet_ctx *c, DevBuf &b, const void *h, size_t bytes)
{
    HIPCHK(c, b.reserve(std::max<size_t>(bytes, 16)));
    if (bytes) HIPCHK(c, hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, c->stream));
    return VDET_OK;
}

static int status_word(vdet_ctx *c, int *out)
{
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    *out = h.status;
    return VDET_OK;
}

int vdet_spatial_maxpool_f64(vdet_ctx *c, const double *h_tub_boxes, const int32_t *h_tub_group, int64_t T,
                             const double *h_det_boxes, const double *h_det_scores, const int64_t *h_group_off,
                             int64_t G, double thres, int64_t *h_out_idx, double *h_out_score)
{
    if (!c || T < 0 || G < 0) return VDET_EINVAL;
    if (T == 0) return VDET_OK;
    if (!h_tub_boxes || !h_tub_group || !h_group_off || !h_out_idx || !h_out_score || G == 0)
        return fail(c, VDET_EINVAL, "null buffer");
    const int64_t M = h_group_off[G];
    for (int64_t t = 0; t < T; ++t)
        if (h_tub_group[t] < 0 || h_tub_group[t] >= G) return fail(c, VDET_EINVAL, "tubelet frame slot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_tub_boxes, (size_t)T * 32))) return rc;
    if ((rc = upload(c, c->tmp[1], h_tub_group, (size_t)T * 4))) return rc;
    if ((rc = upload(c, c->tmp[2], h_det_boxes, (size_t)M * 32))) return rc;
    if ((rc = upload(c, c->tmp[3], h_det_scores, (size_t)M * 8))) return rc;
    if ((rc = upload(c, c->tmp[4], h_group_off, (size_t)(G + 1) * 8))) return rc;
    HIPCHK(c, c->tmp[5].reserve((size_t)T * 8));
    HIPCHK(c, c->tmp[6].reserve((size_t)T * 8));
    {
        StageTimer tm(c, ST_OTHER);
        hipLaunchKernelGGL(spatial_maxpool_kernel, dim3((unsigned)T), dim3(256), 0, c->stream, c->tmp[0].as<double>(),
                           c->tmp[1].as<int32_t>(), c->tmp[2].as<double>(), c->tmp[3].as<double>(),
                           c->tmp[4].as<int64_t>(), thres, c->tmp[5].as<int64_t>(), c->tmp[6].as<double>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out_idx, c->tmp[5].p, (size_t)T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_out_score, c->tmp[6].p, (size_t)T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

int vdet_series_completion_f64(vdet_ctx *c, double *h_vals, const int64_t *h_off, int64_t T)
{
    if (!c || T < 0) return VDET_EINVAL;
    if (T == 0) return VDET_OK;
    if (!h_off) return fail(c, VDET_EINVAL, "null buffer");
    const int64_t n = h_off[T];
    if (n > 0 && !h_vals) return fail(c, VDET_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_vals, (size_t)n * 8))) return rc;
    if ((rc = upload(c, c->tmp[1], h_off, (size_t)(T + 1) * 8))) return rc;
    HIPCHK(c, hipMemsetAsync(&c->d_cnt->status, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(series_completion_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, c->stream,
                       c->tmp[0].as<double>(), c->tmp[1].as<int64_t>(), T, &c->d_cnt->status);
    HIPCHK(c, hipGetLastError());
    if (n) HIPCHK(c, hipMemcpyAsync(h_vals, c->tmp[0].p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    int st = 0;
    if ((rc = status_word(c, &st))) return rc;
    HIPCHK(c, hipMemsetAsync(&c->d_cnt->status, 0, sizeof(int), c->stream));
    if (st) return fail(c, VDET_EINDEX, "list index out of range");
    return VDET_OK;
}

int vdet_series_maxpool_f64(vdet_ctx *c, const double *h_in, double *h_out, const int64_t *h_off, int64_t T,
                            int window, double pad)
{
    if (!c || T < 0) return VDET_EINVAL;
    if (window < 1 || window % 2 != 1) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (T == 0) return VDET_OK;
    if (!h_off) return fail(c, VDET_EINVAL, "null buffer");
    const int64_t n = h_off[T];
    if (n == 0) return VDET_OK;
    if (!h_in || !h_out) return fail(c, VDET_EINVAL, "null buffer");
    std::vector<int32_t> es((size_t)n);
    for (int64_t t = 0; t < T; ++t)
        for (int64_t e = h_off[t]; e < h_off[t + 1]; ++e) es[(size_t)e] = (int32_t)t;
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_in, (size_t)n * 8))) return rc;
    if ((rc = upload(c, c->tmp[1], h_off, (size_t)(T + 1) * 8))) return rc;
    if ((rc = upload(c, c->tmp[2], es.data(), (size_t)n * 4))) return rc;
    HIPCHK(c, c->tmp[3].reserve((size_t)n * 8));
    hipLaunchKernelGGL(series_maxpool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       c->tmp[0].as<double>(), c->tmp[3].as<double>(), c->tmp[1].as<int64_t>(),
                       c->tmp[2].as<int32_t>(), n, window, pad);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, c->tmp[3].p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

int vdet_series_interp_f64(vdet_ctx *c, const double *h_x, const double *h_y, const int64_t *h_koff,
                           const double *h_q, const int64_t *h_qoff, int64_t T, int K, double *h_out)
{
    if (!c || T < 0 || K < 1) return VDET_EINVAL;
    if (T == 0) return VDET_OK;
    if (!h_x || !h_y || !h_koff || !h_q || !h_qoff || !h_out) return fail(c, VDET_EINVAL, "null buffer");
    const int64_t nk = h_koff[T], nq = h_qoff[T];
    for (int64_t t = 0; t < T; ++t)
        if (h_koff[t + 1] - h_koff[t] < 2) return fail(c, VDET_EINVAL, "interpolation needs at least 2 knots");
    if (nq == 0) return VDET_OK;
    std::vector<int32_t> qs((size_t)nq);
    for (int64_t t = 0; t < T; ++t)
        for (int64_t e = h_qoff[t]; e < h_qoff[t + 1]; ++e) qs[(size_t)e] = (int32_t)t;
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_x, (size_t)nk * 8))) return rc;
    if ((rc = upload(c, c->tmp[1], h_y, (size_t)nk * K * 8))) return rc;
    if ((rc = upload(c, c->tmp[2], h_koff, (size_t)(T + 1) * 8))) return rc;
    if ((rc = upload(c, c->tmp[3], h_q, (size_t)nq * 8))) return rc;
    if ((rc = upload(c, c->tmp[4], h_qoff, (size_t)(T + 1) * 8))) return rc;
    if ((rc = upload(c, c->tmp[5], qs.data(), (size_t)nq * 4))) return rc;
    HIPCHK(c, c->tmp[6].reserve((size_t)nq * K * 8));
    hipLaunchKernelGGL(series_interp_kernel, dim3((unsigned)((nq * K + 255) / 256)), dim3(256), 0, c->stream,
                       c->tmp[0].as<double>(), c->tmp[1].as<double>(), c->tmp[2].as<int64_t>(), c->tmp[3].as<double>(),
                       c->tmp[4].as<int64_t>(), c->tmp[5].as<int32_t>(), nq, K, c->tmp[6].as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, c->tmp[6].p, (size_t)nq * K * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

int vdet_threshold_topk(vdet_ctx *c, const void *h_scores, int is_f64, int64_t B, int64_t ld, int col0, int ncls,
                        double thresh, int k, int32_t *h_idx, int32_t *h_cnt)
{
    if (!c || B < 0 || ncls < 0 || k < 0 || col0 < 0 || ld < col0 + ncls) return VDET_EINVAL;
    if (ncls == 0) return VDET_OK;
    if (!h_cnt || (k > 0 && !h_idx)) return fail(c, VDET_EINVAL, "null buffer");
    if (B == 0) { memset(h_cnt, 0, (size_t)ncls * 4); return VDET_OK; }
    if (!h_scores) return fail(c, VDET_EINVAL, "null buffer");
    const size_t es = is_f64 ? 8 : 4;
    const size_t lds = ((es * B + 15) & ~(size_t)15) + (((size_t)4 * B + 15) & ~(size_t)15) + 4 * 260;
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    if (!c->topk_attr_set) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(threshold_topk_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->max_lds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(threshold_topk_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->max_lds));
        c->topk_attr_set = true;
    }
    if (lds > c->max_lds) return fail(c, VDET_EINVAL, "too many boxes per frame for threshold_topk (%lld)", (long long)B);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_scores, (size_t)B * ld * es))) return rc;
    HIPCHK(c, c->tmp[1].reserve((size_t)ncls * std::max(k, 1) * 4));
    HIPCHK(c, c->tmp[2].reserve((size_t)ncls * 4));
    if (is_f64)
        hipLaunchKernelGGL(threshold_topk_kernel<double>, dim3(ncls), dim3(256), lds, c->stream, c->tmp[0].as<double>(),
                           B, ld, col0, thresh, k, c->tmp[1].as<int32_t>(), c->tmp[2].as<int32_t>());
    else
        hipLaunchKernelGGL(threshold_topk_kernel<float>, dim3(ncls), dim3(256), lds, c->stream, c->tmp[0].as<float>(),
                           B, ld, col0, thresh, k, c->tmp[1].as<int32_t>(), c->tmp[2].as<int32_t>());
    HIPCHK(c, hipGetLastError());
    if (k > 0) HIPCHK(c, hipMemcpyAsync(h_idx, c->tmp[1].p, (size_t)ncls * k * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_cnt, c->tmp[2].p, (size_t)ncls * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

int vdet_conv1d_f32(vdet_ctx *c, const float *h_in, int Cin, int L, const float *h_w, const float *h_b, int Cout, int K,
                    int act, float *h_out)
{
    if (!c || Cin < 1 || Cout < 1 || L < 0 || K < 1 || K % 2 != 1 || act < 0 || act > 2)
        return c ? fail(c, VDET_EINVAL, "bad conv1d arguments") : VDET_EINVAL;
    if (L == 0) return VDET_OK;
    if (!h_in || !h_w || !h_b || !h_out) return fail(c, VDET_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_in, (size_t)Cin * L * 4))) return rc;
    if ((rc = upload(c, c->tmp[1], h_w, (size_t)Cout * Cin * K * 4))) return rc;
    if ((rc = upload(c, c->tmp[2], h_b, (size_t)Cout * 4))) return rc;
    HIPCHK(c, c->tmp[3].reserve((size_t)Cout * L * 4));
    HIPCHK(c, c->tmp[4].reserve((size_t)Cout * L * 4));
    const int n = Cout * L;
    hipLaunchKernelGGL(conv1d_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->tmp[0].as<float>(), Cin, L,
                       c->tmp[1].as<float>(), c->tmp[2].as<float>(), Cout, K, act == 1 ? 1 : 0, c->tmp[3].as<float>());
    float *res = c->tmp[3].as<float>();
    if (act == 2) {
        hipLaunchKernelGGL(softmax_channels_kernel, dim3((L + 255) / 256), dim3(256), 0, c->stream, c->tmp[3].as<float>(),
                           Cout, L, c->tmp[4].as<float>());
        res = c->tmp[4].as<float>();
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, res, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
static int temporal_launch(vdet_ctx *c, int mode, const float *d_in, float *d_out, int64_t F, int64_t S, int W,
                           float pad, float bias, const Taps &taps, const int2 *d_seg = nullptr)
{
    if (F == 0 || S == 0) return VDET_OK;
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    StageTimer tm(c, ST_TEMPORAL);
    const bool vec = (S % 4 == 0) && (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0 && (W == 3 || W == 5 || W == 7 || W == 9);
    if (vec) {
        const int64_t S4 = S / 4;
        const unsigned gx = (unsigned)((S4 + 255) / 256);
        // few series (tubelet tracks): split the frame axis so the chip is still filled
        int64_t chunks = 1;
        if ((int64_t)gx < 4 * c->n_cu) chunks = std::min<int64_t>(std::max<int64_t>(F / 16, 1), (4 * c->n_cu + gx - 1) / gx);
        chunks = std::min<int64_t>(chunks, 65535);
        const int64_t fchunk = (F + chunks - 1) / chunks;
        const dim3 grid(gx, (unsigned)((F + fchunk - 1) / fchunk));
        const float4 *in4 = reinterpret_cast<const float4 *>(d_in);
        float4 *out4 = reinterpret_cast<float4 *>(d_out);
#define VDET_TL(WW, MM) hipLaunchKernelGGL((temporal_vec4_kernel<WW, MM>), grid, dim3(256), 0, c->stream, in4, out4, F, S4, fchunk, pad, bias, taps, d_seg)
        if (mode == 0) {
            if (W == 3) VDET_TL(3, 0); else if (W == 5) VDET_TL(5, 0); else if (W == 7) VDET_TL(7, 0); else VDET_TL(9, 0);
        } else {
            if (W == 3) VDET_TL(3, 1); else if (W == 5) VDET_TL(5, 1); else if (W == 7) VDET_TL(7, 1); else VDET_TL(9, 1);
        }
#undef VDET_TL
    } else {
        const int64_t n = F * S;
        const dim3 grid((unsigned)((n + 255) / 256));
        if (mode == 0) hipLaunchKernelGGL(temporal_scalar_kernel<0>, grid, dim3(256), 0, c->stream, d_in, d_out, F, S, W, pad, bias, taps, d_seg);
        else hipLaunchKernelGGL(temporal_scalar_kernel<1>, grid, dim3(256), 0, c->stream, d_in, d_out, F, S, W, pad, bias, taps, d_seg);
    }
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

int vdet_temporal_maxpool_f32(vdet_ctx *c, const float *d_in, float *d_out, int64_t F, int64_t S, int window, float pad)
{
    if (!c) return VDET_EINVAL;
    if (window < 1 || window % 2 != 1) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (F < 0 || S < 0 || F * S > ((int64_t)1 << 40) || (F * S > 0 && (!d_in || !d_out || d_in == d_out)))
        return fail(c, VDET_EINVAL, "bad temporal_maxpool arguments");
    Taps taps{};
    return temporal_launch(c, 0, d_in, d_out, F, S, window, pad, 0.0f, taps);
}

int vdet_temporal_conv_f32(vdet_ctx *c, const float *d_in, float *d_out, int64_t F, int64_t S, const float *h_taps,
                           int K, float bias, float pad)
{
    if (!c) return VDET_EINVAL;
    if (K < 1 || K % 2 != 1 || K > 31 || !h_taps) return fail(c, VDET_EINVAL, "taps: K must be odd and <= 31");
    if (F < 0 || S < 0 || F * S > ((int64_t)1 << 40) || (F * S > 0 && (!d_in || !d_out || d_in == d_out)))
        return fail(c, VDET_EINVAL, "bad temporal_conv arguments");
    Taps taps{};
    for (int k = 0; k < K; ++k) taps.w[k] = h_taps[k];
    return temporal_launch(c, 1, d_in, d_out, F, S, K, pad, bias, taps);
}

static int temporal_maxpool_conv_impl(vdet_ctx *c, const float *d_in, float *d_out_max, float *d_out_conv, int64_t F, int64_t S,
                                      int window, float pad_max, const float *h_taps, float bias, float pad_conv, const int2 *d_seg);

int vdet_temporal_maxpool_conv_f32(vdet_ctx *c, const float *d_in, float *d_out_max, float *d_out_conv, int64_t F, int64_t S,
                                   int window, float pad_max, const float *h_taps, float bias, float pad_conv)
{
    return temporal_maxpool_conv_impl(c, d_in, d_out_max, d_out_conv, F, S, window, pad_max, h_taps, bias, pad_conv, nullptr);
}

static int temporal_maxpool_conv_impl(vdet_ctx *c, const float *d_in, float *d_out_max, float *d_out_conv, int64_t F, int64_t S,
                                      int window, float pad_max, const float *h_taps, float bias, float pad_conv, const int2 *d_seg)
{
    if (!c) return VDET_EINVAL;
    if (window < 1 || window % 2 != 1) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (window > 31 || !h_taps) return fail(c, VDET_EINVAL, "taps: K must be odd and <= 31");
    if (F < 0 || S < 0 || F * S > ((int64_t)1 << 40) ||
        (F * S > 0 && (!d_in || !d_out_max || !d_out_conv || d_in == d_out_max || d_in == d_out_conv || d_out_max == d_out_conv)))
        return fail(c, VDET_EINVAL, "bad temporal_maxpool_conv arguments");
    if (F == 0 || S == 0) return VDET_OK;
    const bool vec = (S % 4 == 0) && (((uintptr_t)d_in | (uintptr_t)d_out_max | (uintptr_t)d_out_conv) & 15) == 0 &&
                     (window == 3 || window == 5 || window == 7);
    if (!vec) {   // shapes the fused kernel does not cover: the two passes
        Taps t0{};
        int rc = temporal_launch(c, 0, d_in, d_out_max, F, S, window, pad_max, 0.0f, t0, d_seg);
        if (rc) return rc;
        for (int k = 0; k < window; ++k) t0.w[k] = h_taps[k];
        return temporal_launch(c, 1, d_in, d_out_conv, F, S, window, pad_conv, bias, t0, d_seg);
    }
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    Taps taps{};
    for (int k = 0; k < window; ++k) taps.w[k] = h_taps[k];
    StageTimer tm(c, ST_TEMPORAL);
    const int64_t S4 = S / 4;
    const unsigned gx = (unsigned)((S4 + 255) / 256);
    int64_t chunks = 1;
    if ((int64_t)gx < 4 * c->n_cu) chunks = std::min<int64_t>(std::max<int64_t>(F / 16, 1), (4 * c->n_cu + gx - 1) / gx);
    chunks = std::min<int64_t>(chunks, 65535);
    const int64_t fchunk = (F + chunks - 1) / chunks;
    const dim3 grid(gx, (unsigned)((F + fchunk - 1) / fchunk));
    const float4 *in4 = reinterpret_cast<const float4 *>(d_in);
    float4 *om = reinterpret_cast<float4 *>(d_out_max), *oc = reinterpret_cast<float4 *>(d_out_conv);
#define VDET_TB(WW) hipLaunchKernelGGL((temporal_both_vec4_kernel<WW>), grid, dim3(256), 0, c->stream, in4, om, oc, F, S4, fchunk, pad_max, pad_conv, bias, taps, d_seg)
    if (window == 3) VDET_TB(3); else if (window == 5) VDET_TB(5); else VDET_TB(7);
#undef VDET_TB
    HIPCHK(c, hipGetLastError());
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
static int volume_pass_impl(vdet_ctx *c, const float *d_scores, int64_t F, int64_t B, int64_t C, int window, float pad_max,
                            const float *h_taps, float bias, float pad_conv, float *d_out_max, float *d_out_conv,
                            int use_score_thresh, float score_thresh, const int2 *d_seg);

int vdet_volume_pass(vdet_ctx *c, const float *d_scores, int64_t F, int64_t B, int64_t C, int window, float pad_max,
                     const float *h_taps, float bias, float pad_conv, float *d_out_max, float *d_out_conv,
                     int use_score_thresh, float score_thresh)
{
    return volume_pass_impl(c, d_scores, F, B, C, window, pad_max, h_taps, bias, pad_conv, d_out_max, d_out_conv, use_score_thresh,
                            score_thresh, nullptr);
}

// frame offsets of V concatenated videos -> per-frame {first, one past last} table on the device (kept in the context)
static int upload_segments(vdet_ctx *c, const int64_t *h_frame_off, int64_t V, int64_t *Ftot)
{
    if (!h_frame_off || V <= 0 || h_frame_off[0] != 0) return fail(c, VDET_EINVAL, "frame offsets must start at 0");
    for (int64_t v = 0; v < V; ++v)
        if (h_frame_off[v + 1] < h_frame_off[v]) return fail(c, VDET_EINVAL, "frame offsets must not decrease");
    const int64_t F = h_frame_off[V];
    if (F > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "too many frames");
    *Ftot = F;
    // the per-frame {first, one past last} table: rebuilt and uploaded only when the offsets differ from the resident ones
    if (c->segtab.p && c->h_seg_off.size() == (size_t)V + 1 && std::equal(c->h_seg_off.begin(), c->h_seg_off.end(), h_frame_off)) return VDET_OK;
    (void)host_sync(c);      // (an earlier copy from the host table may be in flight)
    c->h_seg_off.clear();    // (committed below, once the copy is enqueued: a failure must not leave a "resident" table)
    c->h_seg.resize((size_t)std::max<int64_t>(F, 1));
    for (int64_t v = 0; v < V; ++v)
        for (int64_t f = h_frame_off[v]; f < h_frame_off[v + 1]; ++f) c->h_seg[(size_t)f] = make_int2((int)h_frame_off[v], (int)h_frame_off[v + 1]);
    HIPCHK(c, c->segtab.reserve(c->h_seg.size() * sizeof(int2)));
    if (F) HIPCHK(c, hipMemcpyAsync(c->segtab.p, c->h_seg.data(), (size_t)F * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    c->h_seg_off.assign(h_frame_off, h_frame_off + V + 1);
    return VDET_OK;
}

int vdet_volume_pass_batch(vdet_ctx *c, const float *d_scores, const int64_t *h_frame_off, int64_t V, int64_t B, int64_t C,
                           int window, float pad_max, const float *h_taps, float bias, float pad_conv, float *d_out_max,
                           float *d_out_conv, int use_score_thresh, float score_thresh)
{
    if (!c) return VDET_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int64_t F = 0;
    const int rc = upload_segments(c, h_frame_off, V, &F);
    if (rc) return rc;
    return volume_pass_impl(c, d_scores, F, B, C, window, pad_max, h_taps, bias, pad_conv, d_out_max, d_out_conv, use_score_thresh,
                            score_thresh, c->segtab.as<int2>());
}

static int volume_pass_impl(vdet_ctx *c, const float *d_scores, int64_t F, int64_t B, int64_t C, int window, float pad_max,
                            const float *h_taps, float bias, float pad_conv, float *d_out_max, float *d_out_conv,
                            int use_score_thresh, float score_thresh, const int2 *d_seg)
{
    if (!c) return VDET_EINVAL;
    if (window < 1 || window % 2 != 1) return fail(c, VDET_EINVAL, "Window size must be odd!");
    if (window > 31) return fail(c, VDET_EINVAL, "taps: K must be odd and <= 31");
    if (F < 0 || B < 0 || C < 0 || F * B * C > ((int64_t)1 << 40)) return fail(c, VDET_EINVAL, "bad shape");
    if (F == 0 || B == 0 || C == 0) return VDET_OK;
    const bool conv = h_taps != nullptr;
    if (!d_scores || !d_out_max || (conv && !d_out_conv) || d_scores == d_out_max || (conv && (d_scores == d_out_conv || d_out_max == d_out_conv)))
        return fail(c, VDET_EINVAL, "bad volume_pass arguments");
    if (B > 32767 || F * C > 0x7FFFFFF0ll || F * B > 0x7FFFFFF0ll) return fail(c, VDET_EINVAL, "volume too large");
    c->keys_valid = false;
    // tile: TB boxes (a power of two, >= 16 so that a key row segment is >= 64 B) x C classes in <= NT * ITEMS float4;
    // 4 items per thread (the window of every item lives in registers: more would cost occupancy), 256 or 512 threads
    const int64_t C4 = C / 4;
    int items = 0, TB = 0, NT = 0;
    if (C % 4 == 0 && (window == 3 || window == 5) &&
        (((uintptr_t)d_scores | (uintptr_t)d_out_max | (uintptr_t)(conv ? d_out_conv : d_out_max)) & 15) == 0) {
        for (int nt : {256, 512}) {
            int tb = 64;
            while (tb >= 16 && (int64_t)tb * C4 > (int64_t)nt * 4) tb >>= 1;
            if (tb >= 16 && (nt == 512 || tb >= 32)) { items = 4; TB = tb; NT = nt; break; }
        }
        if (items && (size_t)2 * C4 * TB * 16 > c->max_lds / 2) items = 0;
    }
    if (!items) {
        // shapes the fused kernel does not cover: the temporal pass(es) now, the key transpose with the sort
        if (conv) return temporal_maxpool_conv_impl(c, d_scores, d_out_max, d_out_conv, F, B * C, window, pad_max, h_taps, bias, pad_conv, d_seg);
        Taps t0{};
        return temporal_launch(c, 0, d_scores, d_out_max, F, B * C, window, pad_max, 0.0f, t0, d_seg);
    }
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    // c->tkeys is rewritten (and possibly reallocated) below: sorted lists stay valid only if they are the lists of
    // exactly these keys (the tracking kernels read keys and lists together)
    if (!(c->prep.scores == d_scores && c->prep.F == F && c->prep.B == B && c->prep.C == C && c->prep.layout == VDET_LAYOUT_FBC &&
          c->prep.topk == 0 && c->prep.use_thr == (use_score_thresh ? 1 : 0) && (!use_score_thresh || c->prep.thr == score_thresh)))
        c->lists_valid = false;
    HIPCHK(c, c->tkeys.reserve((size_t)(F * C * B) * 4));
    Taps taps{};
    if (conv) for (int k = 0; k < window; ++k) taps.w[k] = h_taps[k];
    const int ntiles = (int)((B + TB - 1) / TB);
    int64_t chunks = std::min<int64_t>(std::max<int64_t>(F / 8, 1), (8 * c->n_cu + ntiles - 1) / ntiles);
    chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, 65535));
    const int fchunk = (int)((F + chunks - 1) / chunks);
    const dim3 grid((unsigned)ntiles, (unsigned)((F + fchunk - 1) / fchunk));
    const size_t lds = (size_t)2 * C4 * TB * 16;
    int tb_shift = 0;
    while ((1 << tb_shift) < TB) ++tb_shift;
    const void *fn = nullptr;
#define VDET_VP(WW, CV, NTV) reinterpret_cast<const void *>(volume_pass_kernel<WW, 4, CV, NTV>)
    if (window == 3) fn = NT == 256 ? (conv ? VDET_VP(3, true, 256) : VDET_VP(3, false, 256)) : (conv ? VDET_VP(3, true, 512) : VDET_VP(3, false, 512));
    else fn = NT == 256 ? (conv ? VDET_VP(5, true, 256) : VDET_VP(5, false, 256)) : (conv ? VDET_VP(5, true, 512) : VDET_VP(5, false, 512));
    if (!c->vpass_attr_set) {
        for (const void *f : {VDET_VP(3, true, 256), VDET_VP(3, false, 256), VDET_VP(3, true, 512), VDET_VP(3, false, 512),
                              VDET_VP(5, true, 256), VDET_VP(5, false, 256), VDET_VP(5, true, 512), VDET_VP(5, false, 512)})
            HIPCHK(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->max_lds));
        c->vpass_attr_set = true;
    }
#undef VDET_VP
    const float4 *in4 = reinterpret_cast<const float4 *>(d_scores);
    float4 *om = reinterpret_cast<float4 *>(d_out_max), *oc = reinterpret_cast<float4 *>(d_out_conv);
    uint32_t *keys = c->tkeys.as<uint32_t>();
    int Fi = (int)F, Bi = (int)B, C4i = (int)C4;
    void *args[] = {&in4, &om, &oc, &keys, &Fi, &Bi, &C4i, &TB, &tb_shift, (void *)&fchunk, &pad_max, &pad_conv, &bias, &taps,
                    &use_score_thresh, &score_thresh, (void *)&d_seg};
    {
        StageTimer tm(c, ST_TEMPORAL);
        HIPCHK(c, hipLaunchKernel(fn, grid, dim3((unsigned)NT), args, lds, c->stream));
    }
    HIPCHK(c, hipGetLastError());
    c->keysrc.scores = d_scores; c->keysrc.F = F; c->keysrc.B = B; c->keysrc.C = C;
    c->keysrc.use_thr = use_score_thresh ? 1 : 0; c->keysrc.thr = score_thresh;
    c->keys_valid = true;
    return VDET_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C++" {
template <typename T>
static int svm_scores_impl(vdet_ctx *c, const T *h_feat, int64_t n, int64_t k, const T *h_W, const T *h_B, int64_t m, T *h_out)
{
    if (!c || n < 0 || k < 0 || m < 0) return VDET_EINVAL;
    if (n == 0 || m == 0) return VDET_OK;
    if ((k > 0 && (!h_feat || !h_W)) || !h_out) return fail(c, VDET_EINVAL, "null buffer");
    if (n * k > ((int64_t)1 << 34) || k * m > ((int64_t)1 << 34) || n * m > ((int64_t)1 << 34)) return fail(c, VDET_EINVAL, "matrix too large");
    HIPCHK(c, hipSetDevice(c->device));
    timing_reset(c);
    int rc;
    if ((rc = upload(c, c->tmp[0], h_feat, (size_t)(n * k) * sizeof(T)))) return rc;
    if ((rc = upload(c, c->tmp[1], h_W, (size_t)(k * m) * sizeof(T)))) return rc;
    if (h_B && (rc = upload(c, c->tmp[2], h_B, (size_t)m * sizeof(T)))) return rc;
    HIPCHK(c, c->tmp[3].reserve((size_t)(n * m) * sizeof(T)));
    {
        StageTimer tm(c, ST_OTHER);
        hipLaunchKernelGGL(svm_scores_kernel<T>, dim3((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64)), dim3(256), 0, c->stream,
                           c->tmp[0].as<T>(), c->tmp[1].as<T>(), h_B ? c->tmp[2].as<T>() : (const T *)nullptr, n, k, m, c->tmp[3].as<T>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, c->tmp[3].p, (size_t)(n * m) * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, host_sync(c));
    return VDET_OK;
}
}  // extern "C++"

int vdet_svm_scores_f64(vdet_ctx *c, const double *h_feat, int64_t n, int64_t k, const double *h_W, const double *h_B, int64_t m,
                        double *h_out)
{
    return svm_scores_impl<double>(c, h_feat, n, k, h_W, h_B, m, h_out);
}

int vdet_svm_scores_f32(vdet_ctx *c, const float *h_feat, int64_t n, int64_t k, const float *h_W, const float *h_B, int64_t m,
                        float *h_out)
{
    return svm_scores_impl<float>(c, h_feat, n, k, h_W, h_B, m, h_out);
}

}  // extern "C"
