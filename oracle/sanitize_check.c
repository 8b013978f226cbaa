/* TEST INFRASTRUCTURE ONLY -- a driver that runs every entry point of vdet_oracle.c under
 * AddressSanitizer + UndefinedBehaviorSanitizer (the oracle is the checker, so it gets checked):
 *     make -C oracle sanitize        (gcc -fsanitize=address,undefined; exit code 0 = clean)
 * Inputs: seeded pseudo-random boxes incl. the edge cases the tests use (n = 0 / 1, degenerate boxes, NaN
 * scores, caps smaller than the survivor count, ragged strides). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_argsort_desc(const float *scores, int64_t n, int64_t stride, int64_t *order);
int oracle_nms(const float *dets, int64_t n, int64_t ld, int ncols, double thresh, const int64_t *order_in, int64_t *keep, int64_t *n_keep);
int oracle_track_det_nms(const float *tracks, int64_t t, int64_t ldt, const float *dets, int64_t m, int64_t ldd, double thresh, int64_t *keep, int64_t *n_keep);
void oracle_iou_f64(const double *b1, int64_t n1, const double *b2, int64_t n2, double *out);
int oracle_nms_volume(const float *boxes, const float *scores, int64_t F, int64_t B, int64_t C, int64_t f0, int64_t f1, int64_t c0, int64_t c1,
                      double thresh, int use_score_thresh, float score_thresh, int32_t *keep_idx, int32_t *keep_cnt, int64_t cap);
int oracle_nms_volume_mt(const float *boxes, const float *scores, int64_t F, int64_t B, int64_t C, int64_t f0, int64_t f1, int64_t c0, int64_t c1,
                         double thresh, int use_score_thresh, float score_thresh, int32_t *keep_idx, int32_t *keep_cnt, int64_t cap, int nthreads);
int oracle_temporal_maxpool_f32(const float *in, float *out, int64_t F, int64_t S, int window, float pad);
int oracle_temporal_conv_f32(const float *in, float *out, int64_t F, int64_t S, const float *taps, int K, float bias, float pad);

static uint32_t rs = 2463534242u;
static float rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return (float)(rs >> 8) / 16777216.0f; }

static void fill_box(float *b)
{
    const float x = floorf(rnd() * 1230), y = floorf(rnd() * 670);
    b[0] = x; b[1] = y; b[2] = fminf(x + 10 + floorf(rnd() * 290), 1279); b[3] = fminf(y + 10 + floorf(rnd() * 290), 719);
}

int main(void)
{
    int bad = 0;
    const int sizes[] = {0, 1, 2, 65, 700};
    for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; ++si) {
        const int n = sizes[si];
        const int ld5 = 7, ld6 = 6;      /* a padded stride and a tight one */
        float *d5 = malloc(sizeof(float) * (size_t)(n ? n : 1) * ld5), *d6 = malloc(sizeof(float) * (size_t)(n ? n : 1) * ld6);
        for (int i = 0; i < n; ++i) {
            fill_box(d5 + (size_t)i * ld5); d5[(size_t)i * ld5 + 4] = rnd(); d5[(size_t)i * ld5 + 5] = d5[(size_t)i * ld5 + 6] = NAN;
            d6[(size_t)i * ld6] = (float)(1 + (int)(rnd() * 3)); fill_box(d6 + (size_t)i * ld6 + 1); d6[(size_t)i * ld6 + 5] = rnd();
        }
        if (n > 3) { d5[4] = NAN; d5[ld5 + 4] = d5[2 * ld5 + 4]; }                       /* NaN score, a tie */
        int64_t *keep = malloc(sizeof(int64_t) * (size_t)(n ? n : 1)), nk = -1, *order = malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        bad |= oracle_nms(d5, n, ld5, 5, 0.3, NULL, keep, &nk) != 0 || nk < 0 || nk > n;
        bad |= oracle_nms(d6, n, ld6, 6, 0.5, NULL, keep, &nk) != 0 || nk > n;
        float *sc = malloc(sizeof(float) * (size_t)(n ? n : 1));
        for (int i = 0; i < n; ++i) sc[i] = d5[(size_t)i * ld5 + 4];
        bad |= oracle_argsort_desc(sc, n, 1, order) != 0;
        bad |= oracle_nms(d5, n, ld5, 5, 0.3, order, keep, &nk) != 0;
        float tr[2 * 5] = {1, 100, 100, 300, 300, 2, 50, 60, 400, 500};
        bad |= oracle_track_det_nms(tr, 2, 5, d6, n, ld6, 0.3, keep, &nk) != 0 || nk > n;
        bad |= oracle_track_det_nms(tr, 0, 5, d6, n, ld6, 0.3, keep, &nk) != 0;
        if (n == 2) {            /* zero union: x2 = x1 - 1 on both boxes -> the reference's ZeroDivisionError */
            float z[10] = {5, 5, 4, 9, 0.9f, 5, 5, 4, 9, 0.8f};
            bad |= oracle_nms(z, 2, 5, 5, 0.3, NULL, keep, &nk) == 0;
        }
        free(d5); free(d6); free(keep); free(order); free(sc);
    }
    {
        const int64_t F = 4, B = 300, C = 5, cap = 40;
        float *bx = malloc(sizeof(float) * F * B * 4), *sc = malloc(sizeof(float) * F * B * C);
        for (int64_t i = 0; i < F * B; ++i) fill_box(bx + i * 4);
        for (int64_t i = 0; i < F * B * C; ++i) sc[i] = rnd();
        int32_t *ki = malloc(sizeof(int32_t) * F * C * cap), *kc = malloc(sizeof(int32_t) * F * C);
        int32_t *ki2 = malloc(sizeof(int32_t) * F * C * cap), *kc2 = malloc(sizeof(int32_t) * F * C);
        memset(ki, 0xff, sizeof(int32_t) * F * C * cap); memset(ki2, 0xff, sizeof(int32_t) * F * C * cap);
        memset(kc, 0, sizeof(int32_t) * F * C); memset(kc2, 0, sizeof(int32_t) * F * C);
        bad |= oracle_nms_volume(bx, sc, F, B, C, 0, F, 0, C, 0.3, 1, 0.25f, ki, kc, cap) != 0;     /* cap < survivors: truncation path */
        bad |= oracle_nms_volume_mt(bx, sc, F, B, C, 0, F, 0, C, 0.3, 1, 0.25f, ki2, kc2, cap, 3) != 0;
        bad |= memcmp(ki, ki2, sizeof(int32_t) * F * C * cap) != 0 || memcmp(kc, kc2, sizeof(int32_t) * F * C) != 0;
        bad |= oracle_nms_volume(bx, sc, F, B, C, 1, 3, 2, 4, 0.5, 0, 0.0f, ki, kc, cap) != 0;          /* sub-ranges */
        float *o1 = malloc(sizeof(float) * F * B * C);
        const float taps[5] = {0.1f, 0.2f, 0.4f, 0.2f, 0.1f};
        for (int w = 1; w <= 9; w += 2) bad |= oracle_temporal_maxpool_f32(sc, o1, F, B * C, w, -1e5f) != 0;
        bad |= oracle_temporal_maxpool_f32(sc, o1, F, B * C, 4, -1e5f) == 0;                           /* even window: error */
        bad |= oracle_temporal_conv_f32(sc, o1, F, B * C, taps, 5, 0.5f, 0.0f) != 0;
        bad |= oracle_temporal_conv_f32(sc, o1, 1, 7, taps, 3, 0.0f, 1.0f) != 0;
        double b1[8] = {0, 0, 10, 10, 5, 5, 20, 20}, b2[12] = {0, 0, 10, 10, 100, 100, 120, 130, 5, 5, 5, 5}, io[6];
        oracle_iou_f64(b1, 2, b2, 3, io);
        bad |= !(io[0] == 1.0) || !(io[1] == 0.0);
        free(bx); free(sc); free(ki); free(kc); free(ki2); free(kc2); free(o1);
    }
    printf(bad ? "sanitize_check: FAILED\n" : "sanitize_check: ok\n");
    return bad ? 1 : 0;
}
