/*
 * vdet_hip.h -- C-ABI of libvdet_hip.so: the MI355X (gfx950) implementation of vdetlib's
 * per-frame scoring / tubelet post-processing hot path.
 *
 * This is the drop-in boundary.  The reference's only native component is the Cython module
 * utils/cython_nms (built from utils/nms.pyx by setup.py:8-14); its three entry points are what
 * vdet_nms_f32 / vdet_track_det_nms_f32 replace.  The remaining entry points are the array forms
 * of the numeric cores of vdet/video_det.py, vdet/track.py, vdet/tubelet_cls.py and
 * utils/common.py:iou, so that T-CNN style host code (vdetlib_amd/, a py3 mirror of the reference's
 * python API) never computes on the CPU.  Reference citations are file:line in /root/reference.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions across the boundary.
 *   - The caller owns every buffer.  "h_" parameters are host pointers, "d_" parameters are device
 *     pointers on the context's GPU.  The library never frees caller memory; device scratch is
 *     owned by the vdet_ctx.
 *   - Every function returns VDET_OK (0) or a negative vdet_status.  vdet_last_error() gives text.
 *   - h_* entry points are synchronous.  d_* entry points enqueue on the context's stream and
 *     return; data-dependent failures (capacity overflow, zero union) are latched on the device
 *     and reported by vdet_sync().
 *   - A context is not thread-safe: one per thread / per GPU (reference is single-threaded, GIL
 *     held throughout).  Results are deterministic (no float atomics on any result path).
 *   - Boxes are (x1,y1,x2,y2), inclusive pixel coordinates, +1 convention (utils/nms.pyx:24,61-62).
 *   - Sort order wherever the reference says scores.argsort()[::-1] (utils/nms.pyx:25,80 -- numpy's
 *     default UNSTABLE sort): descending score, ties by DESCENDING original index
 *     (= argsort(kind='stable')[::-1]); -0.0 == +0.0; NaN scores sort first.  A caller-supplied
 *     `order` reproduces any other tie order exactly.
 *
 * Limits (the reference has none of them: its lists and loops are python objects; every limit is an error code, never a
 * crash, and nothing is written out of bounds):
 *   what                                   limit                      beyond it
 *   boxes per frame, every entry point     B <= 32767                 VDET_EINVAL (u16 box indices, bit 15 = zero-union tag)
 *   boxes per frame, d_* volume calls      B <= ~18000                VDET_EINVAL (a (frame, class) argsort lives in the CU's
 *                                                                     160 KiB LDS: 6 B / box + tables); the h_* calls fall
 *                                                                     back to a global bitonic sort up to 32767
 *   regular-frame fast kernels             B <= 17408                 same results through the general kernels (slower)
 *   rows of a volume                       F*C, F*B < 2^31 - 16       VDET_EINVAL
 *   rows of an h_* call                    n < 2^31, <= 32767 / frame VDET_EINVAL
 *   edges of one suppression graph         < 2^32                     VDET_ENOMEM
 *   survivors per (frame, class)           cap (caller's choice)      VDET_ECAP latched, count still written
 *   vdet_det_nms_volume top-k              1 <= topk <= 128           VDET_EINVAL
 *   temporal window / taps                 odd, <= 31                 VDET_EINVAL (the one-pass volume kernel: 3 or 5, other
 *                                                                     windows run the separate kernels)
 *   frames per video, vdet_video_batch     none                       videos of more than 1536 frames re-score their tubelet
 *                                                                     series one thread per series (slower, same results)
 *   single-launch h_* calls                n <= 640 rows, t <= 256    larger inputs take the general kernel chain (same results)
 *   link table up front                    B <= 1024                  larger frames scan link steps on demand (same results)
 */
#ifndef VDET_HIP_H
#define VDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdet_ctx vdet_ctx;

typedef enum vdet_status {
    VDET_OK = 0,
    VDET_EINVAL = -1,    /* bad shape / argument  (python: ValueError) */
    VDET_ECAP = -2,      /* output capacity too small; nothing written past capacity (ValueError) */
    VDET_EHIP = -3,      /* HIP runtime failure (RuntimeError) */
    VDET_EDIVZERO = -4,  /* zero union where the reference raises ZeroDivisionError
                            (Cython cdivision=False, utils/nms.pyx:64,122,180) */
    VDET_ENOMEM = -5,
    VDET_EINDEX = -6,    /* where the reference raises IndexError (do_score_completion on a tubelet
                            without any valid score, vdet/tubelet_cls.py:293-295) */
    VDET_EAGAIN = -7     /* vdet_sync only, asynchronous mode (vdet_set_async): a suppression-graph build
                            ran out of scratch; the results since the last vdet_sync are invalid, the
                            scratch has been enlarged -- enqueue the same calls again (RuntimeError) */
} vdet_status;

/* ---- context ------------------------------------------------------------------------------- */

/* device < 0: use the calling thread's current HIP device. */
int vdet_create(vdet_ctx **out, int device);
int vdet_destroy(vdet_ctx *ctx);
/* Enqueue on an existing hipStream_t, used verbatim (e.g. torch's current stream; NULL = HIP's null
 * stream, which is torch's default stream). */
int vdet_set_stream(vdet_ctx *ctx, void *hip_stream);
/* Back to the context's own (non-blocking) stream, the default after vdet_create. */
int vdet_reset_stream(vdet_ctx *ctx);
/* Wait for the context's stream; return (and clear) the first latched device-side failure. */
int vdet_sync(vdet_ctx *ctx);
const char *vdet_last_error(vdet_ctx *ctx);
/* "vdet_hip <version> gfx950" */
const char *vdet_version(void);
/* Wall-clock (ms, HIP events on the context's stream) of the kernels enqueued since the events were
 * last read, summed by stage; used by bench.py for the roofline object.  out[16]: 0 iou_bits_sym
 * (K1s), 1 adj_build (K2), 2 sort (K3), 3 walk (K4), 4 temporal, 5 merge sort, 6 iou_bits general
 * (K1), 7 other, 8 transpose_keys, 9 track_pick, 10 track_link, 11 track_suppress,
 * 12 rescore_spatial, 13 rescore_series, 14-15 unused. */
int vdet_last_timing_ms(vdet_ctx *ctx, float *out16);
/* Number of timed launches per stage behind the sums of vdet_last_timing_ms (call it first). */
int vdet_last_launches(vdet_ctx *ctx, int *out16);
/* Opt-in reuse of the per-video preparation between d_* calls: with the cache enabled,
 * vdet_nms_volume / vdet_track_volume skip the suppression-graph build (same d_boxes pointer, shape
 * and threshold) and the per-(frame,class) sort (same d_scores pointer/layout) of the previous call.
 * The CALLER promises the buffers' contents did not change; vdet_invalidate() drops the cache (call
 * it whenever a buffer was rewritten in place).  Default: disabled. */
int vdet_set_cache(vdet_ctx *ctx, int enable);
int vdet_invalidate(vdet_ctx *ctx);
/* Asynchronous video step (default: disabled).  When enabled, the d_* volume entry points
 * (vdet_nms_volume, vdet_track_volume, vdet_nms_track_volume, vdet_rescore_tracks, vdet_volume_pass)
 * never wait for the device once the context has built one suppression graph: the per-geometry launch
 * tables stay resident, the device status words stay latched until vdet_sync, and the scratch for the
 * adjacency lists is sized from the largest graph seen so far (+50 %).  A graph that still outgrows
 * it makes the next vdet_sync return VDET_EAGAIN (nothing is written out of bounds). */
int vdet_set_async(vdet_ctx *ctx, int enable);
/* Introspection: what = 0 -> 1 if the per-(frame,class) sort uses the returning-LDS-atomic rank
 * (selected by a hardware self-test at vdet_create), 0 if it uses the ballot match;
 * what = 1 -> number of compute units;
 * what = 2 -> 1 if every frame of the last SYNCHRONOUS suppression-graph build was "regular" (finite
 * boxes, positive width / height / area); 0 after an asynchronous build (not known on the host);
 * what = 3 -> 1 if the 64x64 in-wave bit transpose passed its self-test at vdet_create;
 * what = 4 / 5 -> link steps of the last tracking call that were served by the link memo / that scanned their
 * frame (synchronises the stream); 6 / 7 -> the same for the memo warm-up launch;
 * what = 8 -> number of host waits (hipStreamSynchronize) this context has made so far: the asynchronous video
 * step (vdet_set_async) adds none between the entry and the return of the volume entry points;
 * what = 9 -> (frame, class) columns of the last volume sort that the equalised counting sort handed to the LSD
 * radix kernel (tied / quantised / thresholded columns; synchronises the stream), -1 if that sort did not use it.
 *
 * Environment switches, read once by vdet_create (diagnostics: each FORCES a fallback path the library takes anyway on some
 * inputs or devices, with identical results; none selects a tuning variant):
 *   VDET_FORCE_GENERAL=1  the all-pairs predicate kernel + one-survivor walk on every frame (what irregular frames take)
 *   VDET_NO_INDEX=1       no x-sorted proposal index (what frames too large for it take)
 *   VDET_NO_LAZY=1        eager track_det_nms of every crossed list (what irregular frames take)
 *   VDET_NO_FUSED=1       h_* calls of <= 640 rows through the general kernel chain (what larger inputs take)
 *   VDET_BINSORT=0        the LSD radix sort for every column (what tied / thresholded columns take)
 *   VDET_SMALL_LISTS=0    frames of <= 384 boxes through the large-list sort and walk
 *   VDET_DIRECT_LISTS=0   the suppression graph of regular frames of > 384 boxes through the bit matrix (what a context takes for
 *                         good once a row had more neighbours than a direct list slot holds); VDET_DIRECT_CAP=n entries per slot
 *   VDET_ADJ_ROWS=0       (bit-matrix path) the lane-per-row adjacency kernel on every frame (what irregular / small frames take)
 *   VDET_ATOMIC_RANK=0 / VDET_WAVE_TRANSPOSE=0   the variants selected when the start-up hardware probes fail
 *   VDET_BITS_BUDGET_MB=n bytes of bit-matrix scratch per graph-build batch (default 1024) */
int vdet_query(vdet_ctx *ctx, int what);
/* Per-stage HIP-event timing: 0 off (default), 1 on (events of the most recent call), 2 on and
 * accumulating over calls until vdet_last_timing_ms reads them. */
int vdet_set_timing(vdet_ctx *ctx, int enable);

/* ---- utils/cython_nms replacements (host buffers, synchronous) ------------------------------- */

/*
 * nms      (utils/nms.pyx:17-68)   ncols == 5, rows (x1,y1,x2,y2,score)
 * vid_nms  (utils/nms.pyx:71-125)  ncols == 6, rows (frame,x1,y1,x2,y2,score); detections on
 *                                  different frames (float32 equality, :111) never suppress.
 * h_dets: float32, n rows, row stride `ld` elements (>= ncols).  thresh is the reference's boxed
 * python float: suppression iff (double)ovr_f32 >= thresh.  h_order: NULL, or the n indices the
 * reference's argsort()[::-1] produced (tie reproduction).  h_keep: capacity n; receives indices
 * in descending score order; *n_keep their count.  VDET_EDIVZERO mirrors ZeroDivisionError.
 */
int vdet_nms_f32(vdet_ctx *ctx, const float *h_dets, int64_t n, int64_t ld, int ncols,
                 double thresh, const int64_t *h_order, int64_t *h_keep, int64_t *n_keep);

/*
 * track_det_nms (utils/nms.pyx:128-189): h_tracks t rows (frame,x1,y1,x2,y2) stride ldt;
 * h_dets m rows (frame,x1,y1,x2,y2,score) stride ldd.  Round 1: a det is dropped when it overlaps
 * (IoU >= thresh, det as the "i" box) a same-frame track; round 2: vid_nms among the survivors.
 * h_keep (capacity m): indices into dets, descending score.
 */
int vdet_track_det_nms_f32(vdet_ctx *ctx, const float *h_tracks, int64_t t, int64_t ldt,
                           const float *h_dets, int64_t m, int64_t ldd, double thresh,
                           int64_t *h_keep, int64_t *n_keep);

/*
 * The reference's per-tracked-box pattern (vdet/track.py:236-250 and :170-184: one track_det_nms call per box of every new
 * tracklet, each against the still-kept detections of that box's frame) as ONE call: K independent problems, problem k =
 * track_det_nms(h_tracks[h_toff[k] .. h_toff[k+1]), h_dets[h_off[k] .. h_off[k+1]), thresh).  h_toff == NULL: exactly one
 * track row per problem (row k).  h_keep (capacity h_off[K]): problem k's kept indices -- positions inside ITS rows,
 * descending score -- start at h_keep[h_off[k]]; h_nkeep[k] their count.  Problems of <= 640 rows run as one launch of K
 * workgroups and one host wait; otherwise the problems are taken one after the other.  The caller guarantees the problems
 * are independent (the boxes of one tracklet sit on different frames); a frame that repeats belongs in the next call.
 */
int vdet_track_det_nms_batch(vdet_ctx *ctx, const float *h_tracks, const int64_t *h_toff, int64_t ldt,
                             const float *h_dets, const int64_t *h_off, int64_t K, int64_t ldd, double thresh,
                             int64_t *h_keep, int64_t *h_nkeep);

/* iou (utils/common.py:451-468): float64 IoU matrix out[n1,n2] of boxes1[n1,4] x boxes2[n2,4]. */
int vdet_iou_f64(vdet_ctx *ctx, const double *h_boxes1, int64_t n1, const double *h_boxes2,
                 int64_t n2, double *h_out);

/*
 * svm_scores (vdet/image_det.py:109-114), the path's one dense contraction:
 *     out[n, m] = feat[n, k] . W[k, m] + B[m]          (B may be NULL)
 * as a hand-written MFMA kernel (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32): the reference computes it in
 * numpy's result dtype -- float64 with its .mat SVM models, float32 when features and W are float32.  The feature
 * scaling `features * (20 / feat_norm_mean)` stays with the caller (it is rounded in the features' dtype first, :112).
 * Row-major host buffers; per output element the products are accumulated in ascending k, one fma each.
 */
int vdet_svm_scores_f64(vdet_ctx *ctx, const double *h_feat, int64_t n, int64_t k, const double *h_W,
                        const double *h_B, int64_t m, double *h_out);
int vdet_svm_scores_f32(vdet_ctx *ctx, const float *h_feat, int64_t n, int64_t k, const float *h_W,
                        const float *h_B, int64_t m, float *h_out);

/* ---- tubelet re-scoring cores (host buffers, synchronous, float64 like the reference) --------- */

/*
 * Spatial max-pooling core of raw_dets_spatial_max_pooling / dets_spatial_max_pooling
 * (vdet/tubelet_cls.py:514-532, :327-347).  T tubelet boxes h_tub_boxes[T,4]; box t lives on frame
 * slot h_tub_group[t] whose detections are rows [h_group_off[g], h_group_off[g+1]) of
 * h_det_boxes[M,4] / h_det_scores[M] (the class column).  Among the detections with
 * iou > thres (strict, float64, utils/common.py:451-468) the first arg-max of the score (np.argmax:
 * first occurrence, NaN wins): h_out_idx[t] = its index within the frame (-1: none overlaps),
 * h_out_score[t] = its score (-1e5 when none, :529).
 */
int vdet_spatial_maxpool_f64(vdet_ctx *ctx, const double *h_tub_boxes, const int32_t *h_tub_group, int64_t T,
                             const double *h_det_boxes, const double *h_det_scores,
                             const int64_t *h_group_off, int64_t G, double thres, int64_t *h_out_idx,
                             double *h_out_score);

/*
 * do_score_completion (vdet/tubelet_cls.py:284-303) on T ragged series h_vals[h_off[t]..h_off[t+1]),
 * in place.  VDET_EINDEX where the reference raises IndexError (whole series <= -10).
 */
int vdet_series_completion_f64(vdet_ctx *ctx, double *h_vals, const int64_t *h_off, int64_t T);

/* score_proto_temporal_maxpool core (vdet/tubelet_cls.py:399-412) on T ragged float64 series. */
int vdet_series_maxpool_f64(vdet_ctx *ctx, const double *h_in, double *h_out, const int64_t *h_off,
                            int64_t T, int window, double pad);

/*
 * score_proto_interpolation core (vdet/tubelet_cls.py:453-487; scipy interp1d linear ==
 * numpy.interp, plus extrap1d :416-428).  Per tubelet t: knots h_x[h_koff[t]..h_koff[t+1])
 * (ascending, >= 2), K fields stored field-major h_y[h_koff[t]*K + f*L + k]; queries
 * h_q[h_qoff[t]..h_qoff[t+1]); results h_out[h_qoff[t]*K + f*Lq + n].
 */
int vdet_series_interp_f64(vdet_ctx *ctx, const double *h_x, const double *h_y, const int64_t *h_koff,
                           const double *h_q, const int64_t *h_qoff, int64_t T, int K, double *h_out);

/*
 * Per-class threshold + top-k of ONE frame (vdet/video_det.py:89-99).  h_scores [B, ld] float32
 * (is_f64 = 0) or float64 (1); class columns col0 .. col0+ncls-1.  Per class: inds = rows with
 * score > thresh; if more than k: the k best by argsort(-score) (stable), in that order; else all
 * of them in ascending row order.  h_idx [ncls, k] int32, h_cnt [ncls] int32.
 */
int vdet_threshold_topk(vdet_ctx *ctx, const void *h_scores, int is_f64, int64_t B, int64_t ld, int col0,
                        int ncls, double thresh, int k, int32_t *h_idx, int32_t *h_cnt);

/*
 * One temporal-convolution layer of the tubelet TCN (the build's stand-in for the external Caffe
 * net that score_conv_cls feeds, vdet/tubelet_cls.py:15-51; parity unpinned):
 *   out[co,l] = act(b[co] + sum_ci sum_k w[co,ci,k] * in[ci, l+k-K/2]), zero padded, f32,
 * accumulated ci-outer / k-inner without contraction.  h_in [Cin,L], h_w [Cout,Cin,K], h_b [Cout],
 * h_out [Cout,L]; K odd.  act: 0 none, 1 ReLU, 2 softmax over the Cout channels (applied after the
 * affine part, like Caffe's SoftmaxLayer).
 */
int vdet_conv1d_f32(vdet_ctx *ctx, const float *h_in, int Cin, int L, const float *h_w, const float *h_b,
                    int Cout, int K, int act, float *h_out);

/* ---- device-resident array forms (asynchronous; see vdet_sync) ------------------------------- */

#define VDET_LAYOUT_FBC 0 /* scores [F,B,C], class innermost (zs[B,C], utils/protocol.py:538) */
#define VDET_LAYOUT_FCB 1 /* scores [F,C,B] */

/*
 * Per-(frame,class) greedy NMS over a whole video: apply_image_nms (vdet/image_det.py:117-123)
 * for every frame and class of fast_rcnn_det_vid's per-class loop (vdet/video_det.py:89-99),
 * == vid_nms (utils/nms.pyx:71-125) of each class decomposed per frame.
 *   d_boxes  [F,B,4] f32    d_scores [F,B,C] or [F,C,B] f32
 *   use_score_thresh != 0: only boxes with score > score_thresh are candidates (video_det.py:90)
 *   d_keep_idx [F,C,cap] int32: kept box indices (0..B-1), descending score; entries >= count
 *                               are left untouched.   d_keep_cnt [F,C] int32.
 *   A (frame,class) with more than cap survivors latches VDET_ECAP (its count is still written).
 * Limits: see the table at the top of this file (B <= ~18000).
 */
int vdet_nms_volume(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, int layout,
                    int64_t F, int64_t B, int64_t C, double thresh, int use_score_thresh,
                    float score_thresh, int32_t *d_keep_idx, int32_t *d_keep_cnt, int64_t cap);

/*
 * vdet_nms_volume preceded, on the device, by the per-class candidate selection of
 * fast_rcnn_det_vid (vdet/video_det.py:89-99): per (frame, class) the candidates are the boxes with
 * score > score_thresh (use_score_thresh; float32 compare like numpy's), and when more than `topk`
 * (> 0; the reference's max_per_image = 100) remain only the topk best -- argsort(-scores)[:topk],
 * i.e. ties at the cut go to the LOWEST indices -- enter the NMS.  The reference's pipeline order
 * (threshold -> top-k -> apply_image_nms, vdet/image_det.py:117-123) without the scores ever leaving
 * HBM.  topk == 0: no cut (== vdet_nms_volume).
 */
int vdet_nms_volume_topk(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, int layout,
                         int64_t F, int64_t B, int64_t C, double thresh, int use_score_thresh,
                         float score_thresh, int topk, int32_t *d_keep_idx, int32_t *d_keep_cnt,
                         int64_t cap);

/*
 * The Fast R-CNN per-class flow on the device: fast_rcnn_det_vid's per-class loop (vdet/video_det.py:89-99) followed by
 * apply_image_nms (vdet/image_det.py:117-123 -> utils/nms.pyx:17-68) for every frame and class, where -- unlike the
 * vdet_nms_volume family -- every class suppresses ITS OWN regressed boxes (boxes[inds, 4j:4j+4], video_det.py:92).
 *   d_boxes  [F,B,K,4] f32 (== the reference's [B, 4K] box array per frame), 16-byte aligned
 *   d_scores [F,B,K]   f32;  classes j < class0 are skipped (class0 = 1: the background column)
 *   candidates of (frame, j): score > score_thresh (use_score_thresh; float32 compare); more than topk (the
 *   reference's max_per_image, <= 128) -> the topk best, argsort(-score)[:topk] (ties at the cut: lowest indices)
 *   d_dets    [F,K,topk,5] f32 or NULL: the rows (x1,y1,x2,y2,score) in the REFERENCE's row order -- ascending box
 *             index, or descending score when the cut applied (video_det.py:93-97)
 *   d_sel_idx [F,K,topk] int32 or NULL: the box index of every row;  d_det_cnt [F,K] rows per (frame, class)
 *   d_keep    [F,K,topk] int32: the kept ROW positions in descending score order (apply_image_nms's list),
 *   d_keep_cnt [F,K].  Entries behind the counts are left untouched.
 * A zero-union pair that the reference would evaluate latches VDET_EDIVZERO (reported by vdet_sync).
 */
int vdet_det_nms_volume(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, int64_t F, int64_t B, int64_t K,
                        int class0, int use_score_thresh, float score_thresh, int topk, double nms_thresh,
                        float *d_dets, int32_t *d_sel_idx, int32_t *d_det_cnt, int32_t *d_keep, int32_t *d_keep_cnt);

/*
 * vdet_nms_volume with the CALLER's order instead of the build's: d_order [F,C,B] uint16 lists every (frame, class)
 * column's candidates in the order the greedy loop of utils/nms.pyx:26-66 is to visit them (the first d_ncand[f,c]
 * entries; the rest is ignored), e.g. vdet_argsort_volume's lists with ties rearranged the way a particular machine's
 * unstable `scores.argsort()[::-1]` (utils/nms.pyx:25) left them -- the volume-scale form of vdet_nms_f32's h_order.
 * An entry must be a box index < B and appear once per list.
 */
int vdet_nms_volume_ordered(vdet_ctx *ctx, const float *d_boxes, const uint16_t *d_order, const int32_t *d_ncand,
                            int64_t F, int64_t B, int64_t C, double thresh, int32_t *d_keep_idx,
                            int32_t *d_keep_cnt, int64_t cap);

/*
 * Batched small videos (BASELINE configs[0] / [4] shapes: hundreds of frames x <= 300 boxes x 30 classes -- a single
 * such video is launch-bound).  The frames of V videos are concatenated along F; h_frame_off [V+1] (host, starts at 0,
 * strictly increasing) gives every video its frame range.  What does not look across frames -- the suppression graph,
 * the per-(frame, class) sorts and NMS walks (vdet/video_det.py:79-106 over utils/nms.pyx) -- runs ONCE for the whole
 * batch; tracking and re-scoring (vdet/track.py:189-252, vdet/tubelet_cls.py:493-535, :284-303, :386-414) run per
 * video on its frame range with the video as a grid dimension (one launch per stage for ALL videos).  Results are what vdet_nms_track_volume + vdet_rescore_tracks return
 * for each video on its own, laid out video after video:
 *   d_tracks [sum_v C*T*F_v*5] (video v at C*T*5*h_frame_off[v]), d_anchors [V,C,T,3], d_ntracks [V,C],
 *   d_keep_idx [F,C,cap] / d_keep_cnt [F,C] over the concatenated frames (d_keep_cnt null: no NMS output),
 *   d_det_score / d_pooled [sum_v C*T*F_v] f64, d_boxes_out [sum_v C*T*F_v*4] (d_pooled null: no re-scoring).
 */
int vdet_video_batch(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, const int64_t *h_frame_off,
                     int64_t V, int64_t B, int64_t C, double nms_thres, double thres, int max_tracks,
                     double link_thres, int max_frames, float *d_tracks, float *d_anchors, int32_t *d_ntracks,
                     int64_t cap, int32_t *d_keep_idx, int32_t *d_keep_cnt, double overlap_thres, int window,
                     double *d_det_score, double *d_pooled, float *d_boxes_out);

/*
 * vdet_volume_pass over V concatenated videos: a temporal window stops at its video's first / last frame (frames of
 * another video count as padding, like frames outside [0, F) of a single video).  Same outputs and layouts.
 */
int vdet_volume_pass_batch(vdet_ctx *ctx, const float *d_scores, const int64_t *h_frame_off, int64_t V, int64_t B,
                           int64_t C, int window, float pad_max, const float *h_taps, float bias, float pad_conv,
                           float *d_out_max, float *d_out_conv, int use_score_thresh, float score_thresh);

/*
 * Descending argsort of every (frame, class) score column of a volume: the order utils/nms.pyx:25
 * (`scores.argsort()[::-1]`) and vdet/video_det.py:93 (`argsort(-cls_scores)`) walk, with the build's
 * deterministic tie rule (equal scores by DESCENDING index, -0.0 == +0.0, NaN first).
 *   d_scores [F,B,C] or [F,C,B] f32 (layout)   use_score_thresh != 0: boxes with score <= score_thresh are
 *   no candidates and go to the tail.   d_order [F,C,B] uint16: box indices;   d_ncand [F,C] int32: candidates.
 * The same lists vdet_nms_volume / vdet_track_volume build internally (equalised counting sort, LSD radix
 * sort for tied / thresholded columns).  B <= ~18000.
 */
int vdet_argsort_volume(vdet_ctx *ctx, const float *d_scores, int layout, int64_t F, int64_t B, int64_t C,
                        int use_score_thresh, float score_thresh, uint16_t *d_order, int32_t *d_ncand);

/*
 * Centred sliding temporal max over series laid out [F,S] (series s = in[f*S+s]); the array form
 * of score_proto_temporal_maxpool (vdet/tubelet_cls.py:386-414): out[f] = max(in[f-h..f+h]),
 * out-of-range samples = pad (-1e5 in the reference, :402); NaN propagates (np.max).  window must
 * be odd (else VDET_EINVAL, the reference's ValueError :389-390).  d_in != d_out.
 * For a score volume [F,B,C] pass S = B*C.
 */
int vdet_temporal_maxpool_f32(vdet_ctx *ctx, const float *d_in, float *d_out, int64_t F, int64_t S,
                              int window, float pad);

/*
 * Single-channel temporal convolution over [F,S] series: out[f] = bias + sum_k taps[k] *
 * in[f+k-K/2] (out-of-range = pad), accumulated left to right in f32, no FMA contraction.
 * Stands in for the external Caffe TCN of score_conv_cls (vdet/tubelet_cls.py:15-51), whose
 * prototxt/weights are not part of the reference tree (parity unpinned).
 * h_taps: K (odd, <= 31) host floats.
 */
int vdet_temporal_conv_f32(vdet_ctx *ctx, const float *d_in, float *d_out, int64_t F, int64_t S,
                           const float *h_taps, int K, float bias, float pad);

/* Both temporal operators of one volume in one pass (the volume is read once): d_out_max as
 * vdet_temporal_maxpool_f32(window, pad_max), d_out_conv as vdet_temporal_conv_f32(h_taps[window],
 * bias, pad_conv).  Bit-identical to the two separate calls. */
int vdet_temporal_maxpool_conv_f32(vdet_ctx *ctx, const float *d_in, float *d_out_max, float *d_out_conv, int64_t F,
                                   int64_t S, int window, float pad_max, const float *h_taps, float bias, float pad_conv);

/*
 * The one pass over a class-innermost score volume d_scores [F,B,C] (zs[B,C] per frame,
 * utils/protocol.py:538): every score is read ONCE and produces
 *   d_out_max  [F,B,C]  = vdet_temporal_maxpool_f32(window, pad_max)        (vdet/tubelet_cls.py:386-414)
 *   d_out_conv [F,B,C]  = vdet_temporal_conv_f32(h_taps[window], bias, pad_conv); h_taps NULL: none
 *   and, inside the context, the class-major sort keys of all F*C per-(frame, class) problems
 *   (score > score_thresh candidates only when use_score_thresh), which the next
 *   vdet_nms_volume[_topk] (layout FBC) / vdet_track_volume / vdet_nms_track_volume call on the SAME
 *   d_scores, shape and score threshold then uses instead of reading the volume again -- under the
 *   vdet_set_cache contract (cache enabled, buffers unchanged in between; vdet_invalidate drops them).
 * Bit-identical to the separate calls.  Shapes the fused kernel does not cover (C % 4 != 0, window
 * other than 3 / 5, unaligned pointers) run the separate temporal kernels and leave no keys.
 */
int vdet_volume_pass(vdet_ctx *ctx, const float *d_scores, int64_t F, int64_t B, int64_t C, int window,
                     float pad_max, const float *h_taps, float bias, float pad_conv, float *d_out_max,
                     float *d_out_conv, int use_score_thresh, float score_thresh);

/*
 * Greedy tubelet generation for every class of a score volume, device-resident: the array form of
 * greedily_track_from_raw_dets (vdet/track.py:189-252) with the built-in IoU-linking tracker as
 * track_method (the reference's trackers are external MATLAB code).  Per class, up to max_tracks
 * times: anchor = best still-kept detection of the video (stop when its score < thres, :218);
 * link it frame by frame to the proposal with the highest float32 IoU with the current (int-
 * truncated) box while that IoU >= link_thres, at most ceil((max_frames+1)/2) frames per side
 * (max_frames <= 0: no limit; vdet/track.py:64-78); then, for every tracked box, track_det_nms
 * (utils/nms.pyx:128-189, threshold nms_thres) prunes that frame's still-kept detections.
 *   d_boxes [F,B,4] f32, d_scores [F,B,C] f32 (class innermost)
 *   d_tracks  [C,max_tracks,F,5] f32 rows (x1,y1,x2,y2,score), NaN where a track has no box
 *   d_anchors [C,max_tracks,3] f32 (1-based frame, box index, score);  d_ntracks [C] int32
 * Asynchronous after the graph build; failures are latched for vdet_sync.
 */
int vdet_track_volume(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, int64_t F, int64_t B,
                      int64_t C, double nms_thres, double thres, int max_tracks, double link_thres,
                      int max_frames, float *d_tracks, float *d_anchors, int32_t *d_ntracks);

/*
 * vdet_track_volume that also returns the per-(frame, class) NMS survivors of vdet_nms_volume
 * (layout FBC, no score threshold, capacity `cap`: d_keep_idx [F,C,cap] int32 in descending score
 * order, d_keep_cnt [F,C]; VDET_ECAP latched like vdet_nms_volume) -- the detections
 * apply_image_nms (vdet/image_det.py:117-123) keeps for every frame next to the tubelets
 * greedily_track_from_raw_dets (vdet/track.py:189-252) builds from the same raw detections.
 * One call because both consume the same per-(frame, class) sorted lists and the same suppression
 * graph, which are built once.  Results are bit-identical to
 * calling vdet_nms_volume and vdet_track_volume separately.  d_keep_cnt == NULL: no NMS output.
 */
int vdet_nms_track_volume(vdet_ctx *ctx, const float *d_boxes, const float *d_scores, int64_t F, int64_t B,
                          int64_t C, double nms_thres, double thres, int max_tracks, double link_thres,
                          int max_frames, float *d_tracks, float *d_anchors, int32_t *d_ntracks, int64_t cap,
                          int32_t *d_keep_idx, int32_t *d_keep_cnt);

/*
 * Re-scoring of the device tracks: raw_dets_spatial_max_pooling (vdet/tubelet_cls.py:493-535: for
 * every tubelet box the best-scoring detection of its frame with float64 iou > overlap_thres gives
 * det_score and replaces the box -- the "box regression"), do_score_completion (:284-303) and
 * score_proto_temporal_maxpool(window) (:386-414; window 1 = none).
 *   d_tracks [C,T,F,5], d_ntracks [C] as produced by vdet_track_volume
 *   d_det_score [C,T,F] f64: completed spatial-max-pool score;  d_pooled [C,T,F] f64: after the
 *   temporal max-pool;  d_boxes_out [C,T,F,4] f32;  NaN where a track has no box.
 * Latches VDET_EINDEX where the reference raises IndexError (a tubelet without any overlapping
 * detection).
 */
int vdet_rescore_tracks(vdet_ctx *ctx, const float *d_tracks, const int32_t *d_ntracks, const float *d_boxes,
                        const float *d_scores, int64_t F, int64_t B, int64_t C, int max_tracks,
                        double overlap_thres, int window, double *d_det_score, double *d_pooled,
                        float *d_boxes_out);

#ifdef __cplusplus
}
#endif
#endif /* VDET_HIP_H */
