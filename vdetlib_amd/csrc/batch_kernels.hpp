// batch_kernels.hpp -- the per-video stages of vdet_video_batch for ALL videos of a batch in one launch each (round 3).
//
// vdet_video_batch concatenates the frames of V small videos; graph, sorts and NMS walks never look across frames and
// run once for the whole batch.  Tracking and re-scoring follow ONE video in time: every kernel here takes the video
// from its grid (blockIdx.y or .z), cuts that video's views out of the batch-wide buffers -- pure pointer arithmetic
// on a {first frame, frames} table -- and runs the single-video device function of track_kernels.hpp on them.  Same
// code, same results per video as the single-video entry points (tests/test_batch_gpu.py), V times the parallelism:
// the single-video launches are latency-bound grids of C (or C x warm chains) blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "track_kernels.hpp"

namespace vdet {

struct VidDesc { int32_t f0, F; };       // frames [f0, f0 + F) of the concatenated volume

struct BatchTrack {                      // batch-wide bases (element pointers) + the options of the call
    const VidDesc *vids;
    int Ftot, B, C, T, wm;
    const float4 *boxes;                 // [Ftot*B]
    const float *scores;                 // [Ftot,B,C]
    const uint32_t *keys;                // [Ftot*C, B]
    uint16_t *lists;                     // [Ftot*C, B]
    int32_t *cnt;                        // [Ftot*C]
    const uint32_t *group_flags;         // [Ftot] or null
    FrameIndex ix;                       // xbox == null: none
    unsigned long long *memo;            // video v: [2, F_v, B] at 2 * f0 * B
    unsigned int *stats;
    int32_t *warm;                       // [V, C*wm]
    float *chains;                       // video v: [C*wm, F_v, 5] at C*wm*5*f0
    int32_t *chain_nodes;                // video v: [C*wm, F_v]    at C*wm*f0
    int32_t *track_nodes;                // video v: [C, T, F_v]    at C*T*f0
    float *tracks;                       // video v: [C, T, F_v, 5] at C*T*5*f0
    float *anchors;                      // [V, C, T, 3]
    int32_t *ntracks;                    // [V, C]
    TrackState *st;                      // [V, C]
    int32_t *heads;                      // t1 | head | nkp | pos, [Ftot*C] each
    uint8_t *visited;                    // [Ftot*C]
    const GroupDesc *groups;             // [Ftot], absolute row offsets
    const uint2 *row_meta;               // [Ftot*B]
    const uint16_t *adj;
    const uint32_t *group_z;             // [Ftot]
    double thres, link_thres, nms_thres;
    float link_t32, t32;
    int reach_all;                       // < 0: the whole video
    int need_suppress, lazy, mask_words;
    int *status;
    const int *n_irregular;
};

struct VidView {                         // one video's slice of everything
    int f0, F, reach;
    const float4 *boxes; const float *scores; const uint32_t *keys; uint16_t *lists; int32_t *cnt;
    const uint32_t *group_flags; FrameIndex ix; unsigned long long *memo;
    int32_t *warm; float *chains; int32_t *chain_nodes; int32_t *track_nodes; float *tracks; float *anchors; int32_t *ntracks; TrackState *st;
};

__device__ __forceinline__ VidView vid_view(const BatchTrack &bt, const int v)
{
    const VidDesc d = bt.vids[v];
    const int64_t f0 = d.f0, B = bt.B, C = bt.C, T = bt.T, wm = bt.wm;
    VidView w;
    w.f0 = d.f0; w.F = d.F; w.reach = bt.reach_all >= 0 ? bt.reach_all : d.F;
    w.boxes = bt.boxes + f0 * B;
    w.scores = bt.scores + f0 * B * C;
    w.keys = bt.keys + f0 * C * B;
    w.lists = bt.lists + f0 * C * B;
    w.cnt = bt.cnt + f0 * C;
    w.group_flags = bt.group_flags ? bt.group_flags + f0 : nullptr;
    w.ix = FrameIndex{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (bt.ix.xbox) w.ix = FrameIndex{bt.ix.xbox + f0 * B, bt.ix.xord + f0 * B, bt.ix.cum + f0 * 257, bt.ix.info + f0 * 4,
                                      bt.ix.xbox16, bt.ix.xord16, bt.ix.bias16 + f0 * (B + 1)};
    w.memo = bt.memo + 2 * f0 * B;
    w.warm = bt.warm + (int64_t)v * C * wm;
    w.chains = bt.chains + C * wm * 5 * f0;
    w.chain_nodes = bt.chain_nodes + C * wm * f0;
    w.track_nodes = bt.track_nodes + C * T * f0;
    w.tracks = bt.tracks + C * T * 5 * f0;
    w.anchors = bt.anchors + (int64_t)v * C * T * 3;
    w.ntracks = bt.ntracks + (int64_t)v * C;
    w.st = bt.st + (int64_t)v * C;
    return w;
}

// grid (C, V)
__global__ __launch_bounds__(256) void batch_warm_anchors_kernel(const BatchTrack bt)
{
    const VidView w = vid_view(bt, blockIdx.y);
    track_warm_anchors_body(blockIdx.x, w.keys, w.lists, w.cnt, w.F, bt.B, bt.C, w.scores, bt.thres, bt.wm, w.warm, WarmExtra{nullptr, 0.f, 0, 0});
}

// grid (Fmax, 2, V), block = 64 * ceil(B / 64): the same table with the neighbour frame's index staged in LDS (link_fill_frame)
__global__ __launch_bounds__(1024) void batch_link_fill_frame_kernel(const BatchTrack bt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fill_smem[];
    const VidView w = vid_view(bt, blockIdx.z);
    if ((int)blockIdx.x >= w.F) return;
    link_fill_frame(blockIdx.x, blockIdx.y == 0 ? 1 : -1, w.boxes, w.F, bt.B, bt.link_t32, w.group_flags, w.ix, bt.link_thres, w.memo, fill_smem);
}

// grid (C * wm, 2, V); MODE 1: memo warm-up, MODE 2: materialise the warm chains
template <int LT, int MODE>
__global__ __launch_bounds__(LT, (MODE == 1 && LT == 256) ? 5 : 1) void batch_link_kernel(const BatchTrack bt)
{
    const VidView w = vid_view(bt, blockIdx.z);
    track_link_memo_body<LT, MODE, 8>(blockIdx.x, blockIdx.y == 0 ? 1 : -1, w.boxes, w.F, bt.B, bt.T, bt.link_t32, w.reach,
                                      (const TrackState *)nullptr, MODE == 2 ? w.chains : (float *)nullptr, w.group_flags, w.ix,
                                      bt.link_thres, w.memo, MODE == 1 ? bt.stats : (unsigned int *)nullptr, w.warm,
                                      MODE == 2 ? w.chain_nodes : (int32_t *)nullptr);
}

// grid (C, V); dynamic LDS = 4 dead masks of bt.mask_words words
__global__ __launch_bounds__(256) void batch_loop_kernel(const BatchTrack bt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int v = blockIdx.y;
    const VidView w = vid_view(bt, v);
    const int64_t fc = (int64_t)w.f0 * bt.C, FC = (int64_t)bt.Ftot * bt.C;
    SuppressParams sp{};
    sp.boxes = w.boxes;
    sp.F = w.F; sp.B = bt.B; sp.C = bt.C; sp.max_tracks = bt.T;
    sp.groups = bt.groups + w.f0;
    sp.rb_bias = w.f0 * bt.B;
    sp.row_meta = bt.row_meta + (int64_t)w.f0 * bt.B;
    sp.adj = bt.adj;
    sp.group_z = bt.group_z + w.f0;
    sp.group_flags = w.group_flags;
    sp.ix = w.ix;
    sp.thres = bt.nms_thres;
    sp.lists = w.lists;
    sp.cnt = w.cnt;
    sp.visited = bt.visited + fc;
    sp.st = w.st;
    sp.tracks = w.tracks;
    sp.t32 = bt.t32;
    sp.status = bt.status;
    sp.mask_words = bt.mask_words;
    sp.lazy = bt.lazy;
    sp.n_irregular = bt.n_irregular;
    LazyLists lz{};
    lz.boxes = w.boxes; lz.tracks = w.tracks; lz.t32 = bt.t32;
    lz.t1 = bt.heads + fc; lz.head = bt.heads + FC + fc; lz.nkp = bt.heads + 2 * FC + fc; lz.pos = bt.heads + 3 * FC + fc;
    lz.group_flags = bt.lazy ? w.group_flags : nullptr;
    // (bt.wm == 0: the whole link table is known -- no predicted chains to copy, every tubelet is walked here, by pointer chasing)
    const ResolveArgs rv{w.warm, bt.wm, bt.wm > 0 ? w.chains : (float *)nullptr, w.chain_nodes, w.tracks, w.track_nodes};
    LoopArgs la{};
    la.keys = w.keys; la.lists = w.lists; la.cnt = w.cnt;
    la.scores = w.scores; la.thres = bt.thres; la.link_thres = bt.link_thres; la.anchors = w.anchors;
    la.link_t32 = bt.link_t32; la.reach = w.reach;
    la.memo = w.memo; la.stats = bt.stats;
    la.nodes = w.track_nodes; la.ntracks_out = w.ntracks;
    la.need_suppress = bt.need_suppress;
    track_loop_body(blockIdx.x, la, lz, rv, sp, smem);
}

// raw_dets_spatial_max_pooling of every tubelet box: grid (ceil(Fmax * C * T / 4), V), one wave per (class, track, frame)
__global__ __launch_bounds__(256) void batch_rescore_spatial_kernel(const BatchTrack bt, double thres, double *__restrict__ out_score,
                                                                    float *__restrict__ out_box)
{
    const VidView w = vid_view(bt, blockIdx.y);
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (int64_t)w.F * bt.C * bt.T) return;
    const int64_t o = (int64_t)bt.C * bt.T * w.f0;
    rescore_one_scan(wv, threadIdx.x & 63, w.tracks, w.ntracks, w.boxes, w.scores, w.F, bt.B, bt.C, bt.T, thres, out_score + o,
                     out_box + 4 * o, w.ix, w.group_flags);
}

// The same with the candidates of a tubelet box taken from the suppression graph (rescore_adj_one: the neighbours of the proposal
// the box came from, 16 lanes per box): grid (ceil(Fmax * C * T / 16), V); what the graph can not serve goes on `todo` as
// {video, box} for batch_rescore_todo_kernel (the window scan, grid-stride)
__global__ __launch_bounds__(256) void batch_rescore_adj_kernel(const BatchTrack bt, double thres, double min_self_iou, double *__restrict__ out_score,
                                                                float *__restrict__ out_box, int2 *__restrict__ todo, unsigned int *__restrict__ todo_cnt)
{
    const VidView w = vid_view(bt, blockIdx.y);
    const int l = threadIdx.x & 15;
    const int64_t wv = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (wv >= (int64_t)w.F * bt.C * bt.T) return;
    const int64_t o = (int64_t)bt.C * bt.T * w.f0;
    if (!rescore_adj_one(wv, l, w.tracks, w.ntracks, w.boxes, w.scores, w.F, bt.B, bt.C, bt.T, thres, out_score + o, out_box + 4 * o,
                         w.group_flags, w.track_nodes, bt.row_meta + (int64_t)w.f0 * bt.B, bt.adj, min_self_iou) && l == 0)
        todo[atomicAdd(todo_cnt, 1u)] = make_int2((int)blockIdx.y, (int)wv);
}

__global__ __launch_bounds__(256) void batch_rescore_todo_kernel(const BatchTrack bt, double thres, double *__restrict__ out_score,
                                                                 float *__restrict__ out_box, const int2 *__restrict__ todo,
                                                                 const unsigned int *__restrict__ todo_cnt)
{
    const int64_t n = (int64_t)*todo_cnt;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int2 t = todo[i];
        const VidView w = vid_view(bt, t.x);
        const int64_t o = (int64_t)bt.C * bt.T * w.f0;
        rescore_one_scan(t.y, threadIdx.x & 63, w.tracks, w.ntracks, w.boxes, w.scores, w.F, bt.B, bt.C, bt.T, thres, out_score + o,
                         out_box + 4 * o, w.ix, w.group_flags);
    }
}

// do_score_completion + temporal max-pool of every tubelet series: grid (ceil(C * T / 4), V); dynamic LDS 4 * stride_bytes
__global__ __launch_bounds__(256) void batch_rescore_series_kernel(const BatchTrack bt, double *__restrict__ sc, double *__restrict__ out2,
                                                                   int window, int *__restrict__ err, int stride_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char series_smem[];
    const VidView w = vid_view(bt, blockIdx.y);
    if (w.F > kSeriesWaveMaxF) return;       // (a long video: batch_rescore_series_long_kernel)
    const int64_t o = (int64_t)bt.C * bt.T * w.f0;
    rescore_series_wave_body(blockIdx.x, series_smem, sc + o, out2 + o, w.ntracks, w.F, bt.C, bt.T, window, err, stride_bytes);
}

// ... of the videos with more than kSeriesWaveMaxF frames (the LDS stage of the wave form does not hold their series): one
// thread per series, like the single-video entry point's rescore_series_kernel.  grid (ceil(C * T / 64), V), 64 threads
__global__ __launch_bounds__(64) void batch_rescore_series_long_kernel(const BatchTrack bt, double *__restrict__ sc, double *__restrict__ out2,
                                                                       int window, int *__restrict__ err)
{
    const VidView w = vid_view(bt, blockIdx.y);
    if (w.F <= kSeriesWaveMaxF) return;
    const int64_t o = (int64_t)bt.C * bt.T * w.f0;
    rescore_series_serial_body(blockIdx.x * 64 + threadIdx.x, sc + o, out2 + o, w.ntracks, w.F, bt.C, bt.T, window, err);
}

}  // namespace vdet
