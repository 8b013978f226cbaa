/*
 * vdet_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's native hot path
 * (vdetlib utils/nms.pyx) plus array-form restatements of the numeric cores of
 * vdet/tubelet_cls.py and utils/common.py:iou.  It exists to CHECK the HIP
 * product path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under vdetlib_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function in
 * this file against golden vectors produced by the reference itself
 * (tests/golden/make_golden.py imports the reference from /root/reference in
 * the build container and records its outputs).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, no fast-math: the f32
 * operation order below IS the specification).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORACLE_OK        0
#define ORACLE_EINVAL   -1
#define ORACLE_EDIVZERO -4

/* utils/nms.pyx:11-15 -- the module-level inline max/min: "a if a >= b else b".
 * NOT fmaxf: with a NaN operand the SECOND argument wins. */
static inline float ref_max(float a, float b) { return a >= b ? a : b; }
static inline float ref_min(float a, float b) { return a <= b ? a : b; }

/* numpy float32 '<' used by argsort: NaNs sort to the end (ascending). */
static inline int np_lt(float a, float b) { return a < b || (b != b && a == a); }

/* Descending order used when the caller does not inject one:
 * scores.argsort(kind='stable')[::-1]  (utils/nms.pyx:25,80 use the default,
 * unstable kind; on tie-free input both give the same permutation -- SURVEY
 * section 7 "sort tie order").  Stable ascending merge sort, then reversed:
 * ties come out by DESCENDING original index. */
static void merge_sort_idx(const float *s, int64_t stride, int64_t *idx, int64_t *tmp, int64_t n)
{
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, k = lo;
            while (a < mid && b < hi) {
                /* stable: take from the right run only if strictly less */
                if (np_lt(s[idx[b] * stride], s[idx[a] * stride])) tmp[k++] = idx[b++];
                else tmp[k++] = idx[a++];
            }
            while (a < mid) tmp[k++] = idx[a++];
            while (b < hi) tmp[k++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

int oracle_argsort_desc(const float *scores, int64_t n, int64_t stride, int64_t *order)
{
    if (n < 0) return ORACLE_EINVAL;
    if (n == 0) return ORACLE_OK;
    int64_t *tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    if (!tmp) return ORACLE_EINVAL;
    for (int64_t i = 0; i < n; ++i) order[i] = i;
    merge_sort_idx(scores, stride, order, tmp, n);
    for (int64_t i = 0; i < n / 2; ++i) { int64_t t = order[i]; order[i] = order[n - 1 - i]; order[n - 1 - i] = t; }
    free(tmp);
    return ORACLE_OK;
}

/*
 * nms (utils/nms.pyx:17-68) when ncols == 5: rows (x1,y1,x2,y2,score);
 * vid_nms (utils/nms.pyx:71-125) when ncols == 6: rows (frame,x1,y1,x2,y2,score),
 * the frame test comes BEFORE the suppressed test (:111-114).
 * ld = row stride in elements.  order may be NULL (see oracle_argsort_desc).
 * keep receives indices in descending-score order; returns ORACLE_EDIVZERO where
 * the reference raises ZeroDivisionError (Cython cdivision=False).
 */
int oracle_nms(const float *dets, int64_t n, int64_t ld, int ncols, double thresh,
               const int64_t *order_in, int64_t *keep, int64_t *n_keep)
{
    if (n < 0 || (ncols != 5 && ncols != 6) || ld < ncols) return ORACLE_EINVAL;
    *n_keep = 0;
    if (n == 0) return ORACLE_OK;
    const int o = ncols == 6 ? 1 : 0;
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    unsigned char *suppressed = (unsigned char *)calloc((size_t)n, 1);
    int rc = ORACLE_OK;
    for (int64_t i = 0; i < n; ++i) {
        const float *r = dets + i * ld + o;
        /* :24 / :79  areas = (x2 - x1 + 1) * (y2 - y1 + 1) in numpy float32 */
        float w = (r[2] - r[0]) + 1.0f, h = (r[3] - r[1]) + 1.0f;
        areas[i] = w * h;
    }
    if (order_in) memcpy(order, order_in, (size_t)n * sizeof(int64_t));
    else oracle_argsort_desc(dets + o + 4, n, ld, order);

    int64_t nk = 0;
    for (int64_t _i = 0; _i < n && rc == ORACLE_OK; ++_i) {
        const int64_t i = order[_i];
        if (suppressed[i]) continue;
        keep[nk++] = i;
        const float *ri = dets + i * ld + o;
        const float ix1 = ri[0], iy1 = ri[1], ix2 = ri[2], iy2 = ri[3], iarea = areas[i];
        const float fi = o ? dets[i * ld] : 0.0f;
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j];
            if (o && fi != dets[j * ld]) continue;          /* :111 */
            if (suppressed[j]) continue;
            const float *rj = dets + j * ld + o;
            const float xx1 = ref_max(ix1, rj[0]);
            const float yy1 = ref_max(iy1, rj[1]);
            const float xx2 = ref_min(ix2, rj[2]);
            const float yy2 = ref_min(iy2, rj[3]);
            /* :61-62  the "+ 1" is a double add rounded back to f32 in the
             * generated C == a plain f32 add (double rounding is innocuous). */
            const float w = ref_max(0.0f, (float)((double)(xx2 - xx1) + 1.0));
            const float h = ref_max(0.0f, (float)((double)(yy2 - yy1) + 1.0));
            const float inter = w * h;
            const float uni = (iarea + areas[j]) - inter;
            if (uni == 0.0f) { rc = ORACLE_EDIVZERO; break; }
            const float ovr = inter / uni;
            if ((double)ovr >= thresh) suppressed[j] = 1;   /* :65 f64 compare */
        }
    }
    *n_keep = nk;
    free(areas); free(order); free(suppressed);
    return rc;
}

/*
 * track_det_nms (utils/nms.pyx:128-189).  tracks rows (frame,x1,y1,x2,y2),
 * dets rows (frame,x1,y1,x2,y2,score).  Round 1: det i is suppressed by the
 * first same-frame track with IoU >= thresh (the DET is the "i" box).
 * Round 2: vid_nms on the survivors; returns indices into dets.
 */
int oracle_track_det_nms(const float *tracks, int64_t t, int64_t ldt,
                         const float *dets, int64_t m, int64_t ldd, double thresh,
                         int64_t *keep, int64_t *n_keep)
{
    if (t < 0 || m < 0 || ldt < 5 || ldd < 6) return ORACLE_EINVAL;
    *n_keep = 0;
    float *t_areas = (float *)malloc((size_t)(t ? t : 1) * sizeof(float));
    int64_t *remain = (int64_t *)malloc((size_t)(m ? m : 1) * sizeof(int64_t));
    float *sub = (float *)malloc((size_t)(m ? m : 1) * 6 * sizeof(float));
    int64_t *k2 = (int64_t *)malloc((size_t)(m ? m : 1) * sizeof(int64_t));
    int rc = ORACLE_OK;
    for (int64_t j = 0; j < t; ++j) {
        const float *r = tracks + j * ldt;
        t_areas[j] = ((r[3] - r[1]) + 1.0f) * ((r[4] - r[2]) + 1.0f);
    }
    int64_t nr = 0;
    for (int64_t i = 0; i < m && rc == ORACLE_OK; ++i) {
        const float *r = dets + i * ldd;
        const float ix1 = r[1], iy1 = r[2], ix2 = r[3], iy2 = r[4];
        const float iarea = ((ix2 - ix1) + 1.0f) * ((iy2 - iy1) + 1.0f);
        int sup = 0;
        for (int64_t j = 0; j < t; ++j) {
            const float *q = tracks + j * ldt;
            if (r[0] != q[0]) continue;                     /* :170 */
            const float xx1 = ref_max(ix1, q[1]);
            const float yy1 = ref_max(iy1, q[2]);
            const float xx2 = ref_min(ix2, q[3]);
            const float yy2 = ref_min(iy2, q[4]);
            const float w = ref_max(0.0f, (float)((double)(xx2 - xx1) + 1.0));
            const float h = ref_max(0.0f, (float)((double)(yy2 - yy1) + 1.0));
            const float inter = w * h;
            const float uni = (iarea + t_areas[j]) - inter;
            if (uni == 0.0f) { rc = ORACLE_EDIVZERO; break; }
            const float ovr = inter / uni;
            if ((double)ovr >= thresh) { sup = 1; break; }  /* :181-183 */
        }
        if (!sup && rc == ORACLE_OK) remain[nr++] = i;
    }
    if (rc == ORACLE_OK) {
        for (int64_t k = 0; k < nr; ++k) memcpy(sub + k * 6, dets + remain[k] * ldd, 6 * sizeof(float));
        int64_t nk = 0;
        rc = oracle_nms(sub, nr, 6, 6, thresh, NULL, k2, &nk);  /* :187 */
        if (rc == ORACLE_OK) {
            for (int64_t k = 0; k < nk; ++k) keep[k] = remain[k2[k]];
            *n_keep = nk;
        }
    }
    free(t_areas); free(remain); free(sub); free(k2);
    return rc;
}

/* utils/common.py:451-468  iou(boxes1[n1,4], boxes2[n2,4]) -> [n1,n2], float64,
 * +1 convention, "1.*inter/(areas1 + areas2.T - inter)". */
void oracle_iou_f64(const double *b1, int64_t n1, const double *b2, int64_t n2, double *out)
{
    for (int64_t i = 0; i < n1; ++i) {
        const double *p = b1 + 4 * i;
        const double a1 = ((p[2] - p[0]) + 1) * ((p[3] - p[1]) + 1);
        for (int64_t j = 0; j < n2; ++j) {
            const double *q = b2 + 4 * j;
            /* np.maximum/minimum propagate NaN; fmax would not */
            const double ix1 = (p[0] != p[0] || q[0] != q[0]) ? NAN : (p[0] > q[0] ? p[0] : q[0]);
            const double ix2 = (p[2] != p[2] || q[2] != q[2]) ? NAN : (p[2] < q[2] ? p[2] : q[2]);
            const double iy1 = (p[1] != p[1] || q[1] != q[1]) ? NAN : (p[1] > q[1] ? p[1] : q[1]);
            const double iy2 = (p[3] != p[3] || q[3] != q[3]) ? NAN : (p[3] < q[3] ? p[3] : q[3]);
            double iw = (ix2 - ix1) + 1, ih = (iy2 - iy1) + 1;
            iw = (iw != iw) ? NAN : (iw > 0 ? iw : 0);
            ih = (ih != ih) ? NAN : (ih > 0 ? ih : 0);
            const double a2 = ((q[2] - q[0]) + 1) * ((q[3] - q[1]) + 1);
            const double inter = iw * ih;
            out[i * n2 + j] = 1. * inter / ((a1 + a2) - inter);
        }
    }
}

/*
 * Batched per-(frame,class) greedy NMS over a score volume: the array form of
 * "apply_image_nms for every frame and class" (vdet/image_det.py:117-123 over
 * vdet/video_det.py:89-99's per-class loop).  boxes [F,B,4], scores [F,B,C]
 * (class innermost, as zs[B,C] -- utils/protocol.py:538).  With use_score_thresh the
 * candidates are the boxes with score > score_thresh (video_det.py:90), else all boxes.
 * keep_idx [F,C,cap] (descending score), keep_cnt [F,C].  frames/classes give
 * the half-open sub-ranges to process so bench.py can time a bounded sample.
 */
int oracle_nms_volume(const float *boxes, const float *scores, int64_t F, int64_t B, int64_t C,
                      int64_t f0, int64_t f1, int64_t c0, int64_t c1,
                      double thresh, int use_score_thresh, float score_thresh, int32_t *keep_idx,
                      int32_t *keep_cnt, int64_t cap)
{
    (void)F;
    float *d = (float *)malloc((size_t)(B ? B : 1) * 5 * sizeof(float));
    int64_t *map = (int64_t *)malloc((size_t)(B ? B : 1) * sizeof(int64_t));
    int64_t *k = (int64_t *)malloc((size_t)(B ? B : 1) * sizeof(int64_t));
    int rc = ORACLE_OK;
    for (int64_t f = f0; f < f1 && rc == ORACLE_OK; ++f)
        for (int64_t c = c0; c < c1 && rc == ORACLE_OK; ++c) {
            int64_t n = 0;
            for (int64_t b = 0; b < B; ++b) {
                const float s = scores[(f * B + b) * C + c];
                if (use_score_thresh && !(s > score_thresh)) continue;
                memcpy(d + n * 5, boxes + (f * B + b) * 4, 4 * sizeof(float));
                d[n * 5 + 4] = s; map[n++] = b;
            }
            int64_t nk = 0;
            rc = oracle_nms(d, n, 5, 5, thresh, NULL, k, &nk);
            if (rc != ORACLE_OK) break;
            keep_cnt[f * C + c] = (int32_t)nk;
            for (int64_t i = 0; i < nk && i < cap; ++i) keep_idx[(f * C + c) * cap + i] = (int32_t)map[k[i]];
        }
    free(d); free(map); free(k);
    return rc;
}

/*
 * The same per-(frame, class) loop on `nthreads` host threads (the problems are independent; the
 * reference itself is single-threaded, an 8-process multiprocessing run is how SURVEY 8d quotes
 * "all cores").  Only bench.py's all-core CPU baseline uses it.  Results == oracle_nms_volume.
 */
#include <pthread.h>
typedef struct {
    const float *boxes, *scores;
    int64_t B, C, f0, c0, nc, nprob, cap;
    double thresh; int use_thr; float thr;
    int32_t *keep_idx, *keep_cnt;
    int64_t next;             /* shared problem counter */
    int rc;
    pthread_mutex_t mu;
} mt_job_t;

static void *mt_worker(void *arg)
{
    mt_job_t *j = (mt_job_t *)arg;
    const int64_t B = j->B, C = j->C;
    float *d = (float *)malloc((size_t)(B ? B : 1) * 5 * sizeof(float));
    int64_t *map = (int64_t *)malloc((size_t)(B ? B : 1) * sizeof(int64_t));
    int64_t *k = (int64_t *)malloc((size_t)(B ? B : 1) * sizeof(int64_t));
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const int64_t i = j->next++;
        const int stop = j->rc != ORACLE_OK;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->nprob || stop) break;
        const int64_t f = j->f0 + i / j->nc, c = j->c0 + i % j->nc;
        int64_t n = 0;
        for (int64_t b = 0; b < B; ++b) {
            const float s = j->scores[(f * B + b) * C + c];
            if (j->use_thr && !(s > j->thr)) continue;
            memcpy(d + n * 5, j->boxes + (f * B + b) * 4, 4 * sizeof(float));
            d[n * 5 + 4] = s; map[n++] = b;
        }
        int64_t nk = 0;
        const int rc = oracle_nms(d, n, 5, 5, j->thresh, NULL, k, &nk);
        if (rc != ORACLE_OK) { pthread_mutex_lock(&j->mu); j->rc = rc; pthread_mutex_unlock(&j->mu); break; }
        j->keep_cnt[f * C + c] = (int32_t)nk;
        for (int64_t q = 0; q < nk && q < j->cap; ++q) j->keep_idx[(f * C + c) * j->cap + q] = (int32_t)map[k[q]];
    }
    free(d); free(map); free(k);
    return NULL;
}

int oracle_nms_volume_mt(const float *boxes, const float *scores, int64_t F, int64_t B, int64_t C,
                         int64_t f0, int64_t f1, int64_t c0, int64_t c1,
                         double thresh, int use_score_thresh, float score_thresh, int32_t *keep_idx,
                         int32_t *keep_cnt, int64_t cap, int nthreads)
{
    (void)F;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    mt_job_t j;
    j.boxes = boxes; j.scores = scores; j.B = B; j.C = C; j.f0 = f0; j.c0 = c0; j.nc = c1 - c0;
    j.nprob = (f1 - f0) * (c1 - c0); j.cap = cap; j.thresh = thresh; j.use_thr = use_score_thresh; j.thr = score_thresh;
    j.keep_idx = keep_idx; j.keep_cnt = keep_cnt; j.next = 0; j.rc = ORACLE_OK;
    if (j.nc <= 0 || j.nprob <= 0) return ORACLE_OK;
    pthread_mutex_init(&j.mu, NULL);
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    int started = 0;
    for (int t = 0; t < nthreads; ++t) { if (pthread_create(&th[t], NULL, mt_worker, &j) != 0) break; ++started; }
    if (started == 0) mt_worker(&j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&j.mu);
    return j.rc;
}

/*
 * Centred sliding temporal max over series laid out [F, S] (series s is
 * in[f*S + s]): the array form of score_proto_temporal_maxpool
 * (vdet/tubelet_cls.py:386-414): out[f] = max(in[f-h .. f+h]), out-of-range = pad
 * (-1e5 in the reference, :402).  window must be odd.
 */
int oracle_temporal_maxpool_f32(const float *in, float *out, int64_t F, int64_t S, int window, float pad)
{
    if (window < 1 || window % 2 != 1) return ORACLE_EINVAL;
    const int h = window / 2;
    for (int64_t f = 0; f < F; ++f)
        for (int64_t s = 0; s < S; ++s) {
            /* np.max over the rolled stack (:404-409): NaN propagates */
            float m = 0; int first = 1, nan = 0;
            for (int d = -h; d <= h; ++d) {
                const int64_t g = f + d;
                const float v = (g < 0 || g >= F) ? pad : in[g * S + s];
                if (v != v) nan = 1;
                if (first || v > m) { m = v; first = 0; }
            }
            out[f * S + s] = nan ? NAN : m;
        }
    return ORACLE_OK;
}

/*
 * Single-channel temporal convolution over [F,S] series (the build's own op
 * standing in for the external TCN of score_conv_cls, vdet/tubelet_cls.py:15-51
 * -- parity UNPINNED for this one, the net is not in the reference tree):
 * out[f] = bias + sum_{k=0..K-1} taps[k] * in[f + k - K/2], out-of-range = pad,
 * accumulated left to right in f32 without contraction.
 */
int oracle_temporal_conv_f32(const float *in, float *out, int64_t F, int64_t S,
                             const float *taps, int K, float bias, float pad)
{
    if (K < 1 || K % 2 != 1) return ORACLE_EINVAL;
    const int h = K / 2;
    for (int64_t f = 0; f < F; ++f)
        for (int64_t s = 0; s < S; ++s) {
            float acc = bias;
            for (int k = 0; k < K; ++k) {
                const int64_t g = f + k - h;
                const float v = (g < 0 || g >= F) ? pad : in[g * S + s];
                const float p = taps[k] * v;
                acc = acc + p;
            }
            out[f * S + s] = acc;
        }
    return ORACLE_OK;
}
