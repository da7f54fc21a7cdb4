/*
 * brisk_oracle.c -- see brisk_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 *
 * A sequential restatement, in the reference's own evaluation order, including the lazily filled per-layer score
 * cache that isMax2D reads raw (brisk.cpp:840-842, 1685-1694): which pixels hold a score at the moment a tie is
 * broken depends on everything processed before, so the order is part of the result.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off (see Makefile): no FMA contraction, FLT_EVAL_METHOD 0.
 */
#include "brisk_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ segment tests (OAST 9/16, AGAST 5/8) */
/* Bresenham circles in the order of init_pattern(): oast9_16.h:74-92, agast5_8.h:66-76 */
static const int C16[16][2] = {{-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}, {0, -3}, {1, -3}, {2, -2}, {3, -1},
                               {3, 0},  {3, 1},   {2, 2},   {1, 3},   {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}};
static const int C8[8][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}, {1, 0}, {1, 1}, {0, 1}, {-1, 1}};

static int segment_test(const uint8_t *p, int stride, int b, const int (*circle)[2], int n, int arc)
{
    const int cb = *p + b, c_b = *p - b; /* oast9_16.cc:86-87 */
    int v[16];
    for (int k = 0; k < n; ++k) v[k] = p[circle[k][0] + circle[k][1] * stride];
    for (int s = 0; s < n; ++s) {
        int brighter = 1, darker = 1;
        for (int k = 0; k < arc; ++k) {
            const int q = v[(s + k) % n];
            if (!(q > cb)) brighter = 0;
            if (!(q < c_b)) darker = 0;
        }
        if (brighter || darker) return 1;
    }
    return 0;
}

int mo_oast9_16_is_corner(const uint8_t *p, int stride, int b) { return segment_test(p, stride, b, C16, 16, 9); }

/* oast9_16_nms.cc:42-2116 / agast5_8_nms.cc:42-402: bisection between b and 255 around the tree */
static int bisect_score(const uint8_t *p, int stride, int b, const int (*circle)[2], int n, int arc)
{
    int bmin = b, bmax = 255;
    int b_test = (bmax + bmin) / 2;
    for (;;) {
        if (segment_test(p, stride, b_test, circle, n, arc))
            bmin = b_test;
        else
            bmax = b_test;
        if (bmin == bmax - 1 || bmin == bmax) return bmin;
        b_test = (bmin + bmax) / 2;
    }
}

int mo_oast9_16_score(const uint8_t *p, int stride, int bmin) { return bisect_score(p, stride, bmin, C16, 16, 9); }
int mo_agast5_8_score(const uint8_t *p, int stride, int bmin) { return bisect_score(p, stride, bmin, C8, 8, 5); }

/* oast9_16.cc:46-2141: y in [3, ysize-3), x in [3, xsize-4], raster order */
int mo_oast9_16_detect(const uint8_t *img, int w, int h, int b, int32_t *xy, int cap)
{
    int n = 0;
    const int xsizeB = w - 4, ysizeB = h - 3;
    for (int y = 3; y < ysizeB; ++y)
        for (int x = 3; x <= xsizeB; ++x)
            if (mo_oast9_16_is_corner(img + (size_t)y * w + x, w, b)) {
                if (n < cap) {
                    xy[2 * n] = x;
                    xy[2 * n + 1] = y;
                }
                ++n;
            }
    return n;
}

/* ------------------------------------------------------------------ pyramid resampling */
static inline int avg_u8(int a, int b) { return (a + b + 1) >> 1; } /* _mm_avg_epu8 */

/* brisk.cpp:1840-1972.  The _mm_adds_epu8(upper, ones) results are overwritten before use (:1885-1886): dead. */
void mo_brisk_halfsample(const uint8_t *src, int w, int h, uint8_t *dst)
{
    const int dw = w / 2, dh = h / 2;
    const int leftover_cols = (w % 16) / 2;
    const int noleftover = (w % 16) == 0;
    const int hsize = w / 16, end = hsize / 2, half_end = hsize % 2;
    for (int r = 0; r < dh; ++r) {
        const uint8_t *u = src + (size_t)(2 * r) * w, *l = u + w;
        uint8_t *d = dst + (size_t)r * dw;
        int c = 0;
        for (; c < 16 * end; ++c) /* two 16-byte blocks -> 16 outputs: avg of the vertical averages (:1880-1907) */
            d[c] = (uint8_t)avg_u8(avg_u8(u[2 * c], l[2 * c]), avg_u8(u[2 * c + 1], l[2 * c + 1]));
        if (half_end) /* one block left: the horizontal step is a truncating mean (:1929-1933) */
            for (int j = 0; j < 8; ++j, ++c) d[c] = (uint8_t)((avg_u8(u[2 * c], l[2 * c]) + avg_u8(u[2 * c + 1], l[2 * c + 1])) / 2);
        if (!noleftover) { /* :1949-1956: reads columns k and k+1 (not 2k, 2k+1) after the last whole block */
            const uint8_t *p1 = u + 16 * hsize, *p2 = l + 16 * hsize;
            for (int k = 0; k < leftover_cols; ++k, ++c) d[c] = (uint8_t)((p1[k] + p1[k + 1] + p2[k] + p2[k + 1]) / 4);
        }
    }
}

/* brisk.cpp:1974-2065 */
void mo_brisk_twothirdsample(const uint8_t *src, int w, int h, uint8_t *dst)
{
    static const int T1[10] = {1, 1, 4, 4, 7, 7, 10, 10, 12, 12}; /* mask1 | mask2 (:1982-1983): the last pair is 12, not 13 */
    static const int T2[10] = {0, 2, 3, 5, 6, 8, 9, 11, 12, 14};   /* mask (:1984) */
    const int dw = (w / 3) * 2;
    const int leftover_cols = ((w / 3) * 3) % 15;
    const int hsize = w / 15;
    for (int r = 0; r < h / 3; ++r) {
        const uint8_t *p1 = src + (size_t)(3 * r) * w, *p2 = p1 + w, *p3 = p2 + w;
        uint8_t *d1 = dst + (size_t)(2 * r) * dw, *d2 = d1 + dw;
        for (int i = 0; i < hsize; ++i) {
            int up[15], lo[15];
            for (int k = 0; k < 15; ++k) {
                up[k] = avg_u8(avg_u8(p1[15 * i + k], p2[15 * i + k]), p1[15 * i + k]); /* :2007 */
                lo[k] = avg_u8(avg_u8(p3[15 * i + k], p2[15 * i + k]), p3[15 * i + k]); /* :2013 */
            }
            for (int m = 0; m < 10; ++m) {
                d1[10 * i + m] = (uint8_t)avg_u8(avg_u8(up[T2[m]], up[T1[m]]), up[T2[m]]);
                d2[10 * i + m] = (uint8_t)avg_u8(avg_u8(lo[T2[m]], lo[T1[m]]), lo[T2[m]]);
            }
        }
        const uint8_t *a = p1 + 15 * hsize, *b = p2 + 15 * hsize, *c = p3 + 15 * hsize;
        uint8_t *e1 = d1 + 10 * hsize, *e2 = d2 + 10 * hsize;
        for (int j = 0; j < leftover_cols; j += 3) { /* :2036-2052 */
            const int A1 = a[j], A2 = a[j + 1], A3 = a[j + 2];
            const int B1 = b[j], B2 = b[j + 1], B3 = b[j + 2];
            const int C1 = c[j], C2 = c[j + 1], C3 = c[j + 2];
            *e1++ = (uint8_t)(((4 * A1 + 2 * (A2 + B1) + B2) / 9) & 0xff);
            *e1++ = (uint8_t)(((4 * A3 + 2 * (A2 + B3) + B2) / 9) & 0xff);
            *e2++ = (uint8_t)(((4 * C1 + 2 * (C2 + B1) + B2) / 9) & 0xff);
            *e2++ = (uint8_t)(((4 * C3 + 2 * (C2 + B3) + B2) / 9) & 0xff);
        }
    }
}

/* ------------------------------------------------------------------ BriskLayer */
typedef struct {
    uint8_t *img, *scores;
    int w, h;
    float scale, offset;
    int32_t *pts; /* OAST points of the last getKeypoints call */
    int n_pts;
} layer_t;

struct mo_brisk {
    int layers;
    layer_t L[16];
    int threshold, safe_threshold;
};

/* brisk.cpp:1685-1694.  `score` is a reference into scores_: whatever cornerScore returns is written to the cache. */
static int layer_score(layer_t *l, int x, int y, int threshold)
{
    if (x < 3 || y < 3) return 0;
    if (x >= l->w - 3 || y >= l->h - 3) return 0;
    uint8_t *score = l->scores + x + (size_t)y * l->w;
    if (*score > 2) return *score;
    *score = (uint8_t)mo_oast9_16_score(l->img + x + (size_t)y * l->w, l->w, threshold - 1);
    if (*score < threshold) *score = 0;
    return *score;
}

/* brisk.cpp:1696-1703: not cached */
static int layer_score_5_8(const layer_t *l, int x, int y, int threshold)
{
    if (x < 2 || y < 2) return 0;
    if (x >= l->w - 2 || y >= l->h - 2) return 0;
    int score = (uint8_t)mo_agast5_8_score(l->img + x + (size_t)y * l->w, l->w, threshold - 1);
    if (score < threshold) score = 0;
    return score;
}

/* What a C `float` expression means.  The reference is a 32-bit Visual Studio 2010 project (README.md:13, MoFREAK.vcxproj:
 * Win32 configurations, no /arch option): that compiler emits x87 code, the CRT runs the FPU at 53-bit precision, and
 * under its default /fp:precise an expression's intermediates stay in FPU registers; values are rounded to float where
 * they are assigned to a float, cast, passed or returned.  MO_BRISK_FP_X87 (the default) restates that; MO_BRISK_FP_SSE is
 * the other reading (every float operation rounds to float: what /arch:SSE2 or a 64-bit build would do).  Every
 * operation below is done in double and OP() rounds its result to float or not: rounding a double +, -, *, / of two
 * floats to float equals the float operation (53 >= 2 * 24 + 2 bits), so the SSE reading is exact as well. */
static int g_fp_model = MO_BRISK_FP_X87;
void mo_brisk_set_fp_model(int model) { g_fp_model = model == MO_BRISK_FP_SSE ? MO_BRISK_FP_SSE : MO_BRISK_FP_X87; }
int mo_brisk_get_fp_model(void) { return g_fp_model; }
static inline double OP(double v) { return g_fp_model == MO_BRISK_FP_X87 ? v : (double)(float)v; }
#define FMUL(a, b) OP((double)(a) * (double)(b))
#define FADD(a, b) OP((double)(a) + (double)(b))
#define FSUB(a, b) OP((double)(a) - (double)(b))
#define FDIV(a, b) OP((double)(a) / (double)(b))

/* brisk.cpp:1705-1738 with scale == 1.0f (every call site passes the default): bilinear inside the layer,
 * returned through uint8_t (truncation).  The scale > 1 branch (value()) is never reached by the detector. */
static int layer_score_f(layer_t *l, float xf, float yf, int threshold)
{
    const int x = (int)xf;
    const float rx1 = xf - (float)x;
    const float rx = 1.0f - rx1;
    const int y = (int)yf;
    const float ry1 = yf - (float)y;
    const float ry = 1.0f - ry1;
    /* operands are evaluated left to right here; the four calls only differ in which cache cells they fill */
    const float s00 = (float)layer_score(l, x, y, threshold);
    const float s10 = (float)layer_score(l, x + 1, y, threshold);
    const float s01 = (float)layer_score(l, x, y + 1, threshold);
    const float s11 = (float)layer_score(l, x + 1, y + 1, threshold);
    /* one expression, converted to uint8_t at the return */
    const double v = FADD(FADD(FADD(FMUL(FMUL(rx, ry), s00), FMUL(FMUL(rx1, ry), s10)), FMUL(FMUL(rx, ry1), s01)), FMUL(FMUL(rx1, ry1), s11));
    return (uint8_t)v;
}

static void layer_free(layer_t *l)
{
    free(l->img);
    free(l->scores);
    free(l->pts);
    memset(l, 0, sizeof(*l));
}

/* ------------------------------------------------------------------ BriskScaleSpace */
mo_brisk *mo_brisk_create(const uint8_t *img, int stride, int w, int h, int octaves)
{
    mo_brisk *b = (mo_brisk *)calloc(1, sizeof(*b));
    b->layers = octaves == 0 ? 1 : 2 * octaves; /* :562-567 */
    if (b->layers > 16) b->layers = 16;
    layer_t *L = b->L;
    L[0].w = w;
    L[0].h = h;
    L[0].scale = 1.0f;
    L[0].offset = 0.0f;
    L[0].img = (uint8_t *)malloc((size_t)w * h + 64);
    for (int y = 0; y < h; ++y) memcpy(L[0].img + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
    /* :572-588: layer 1 = 2/3 of layer 0, then every layer i >= 2 is half of layer i-2 */
    for (int i = 1; i < b->layers; ++i) {
        const layer_t *s = i == 1 ? &L[0] : &L[i - 2];
        if (i == 1) { /* :1665-1670 */
            L[i].w = 2 * (s->w / 3);
            L[i].h = 2 * (s->h / 3);
            L[i].scale = (float)((double)s->scale * 1.5);
        } else { /* :1659-1664 */
            L[i].w = s->w / 2;
            L[i].h = s->h / 2;
            L[i].scale = s->scale * 2;
        }
        L[i].offset = (float)(0.5 * (double)L[i].scale - 0.5);
        L[i].img = (uint8_t *)calloc((size_t)L[i].w * L[i].h + 64, 1);
        if (i == 1)
            mo_brisk_twothirdsample(s->img, s->w, s->h, L[i].img);
        else
            mo_brisk_halfsample(s->img, s->w, s->h, L[i].img);
    }
    for (int i = 0; i < b->layers; ++i) L[i].scores = (uint8_t *)calloc((size_t)L[i].w * L[i].h + 64, 1);
    return b;
}

void mo_brisk_destroy(mo_brisk *b)
{
    if (!b) return;
    for (int i = 0; i < b->layers; ++i) layer_free(&b->L[i]);
    free(b);
}

int mo_brisk_layers(const mo_brisk *b) { return b->layers; }

void mo_brisk_layer_info(const mo_brisk *b, int layer, int *w, int *h, float *scale, float *offset)
{
    if (w) *w = b->L[layer].w;
    if (h) *h = b->L[layer].h;
    if (scale) *scale = b->L[layer].scale;
    if (offset) *offset = b->L[layer].offset;
}

const uint8_t *mo_brisk_layer_image(const mo_brisk *b, int layer) { return b->L[layer].img; }
const uint8_t *mo_brisk_layer_scores(const mo_brisk *b, int layer) { return b->L[layer].scores; }

int mo_brisk_layer_points(const mo_brisk *b, int layer, int32_t *xy, int cap)
{
    const layer_t *l = &b->L[layer];
    const int n = l->n_pts < cap ? l->n_pts : cap;
    if (xy && n > 0) memcpy(xy, l->pts, sizeof(int32_t) * 2 * (size_t)n);
    return l->n_pts;
}

/* brisk.cpp:1535-1644.  Shifts of negative ints are written as the multiplications they perform. */
float mo_brisk_subpixel2d(const int s[9], float *delta_x, float *delta_y)
{
    const int s_0_0 = s[0], s_0_1 = s[1], s_0_2 = s[2], s_1_0 = s[3], s_1_1 = s[4], s_1_2 = s[5], s_2_0 = s[6], s_2_1 = s[7],
              s_2_2 = s[8];
    const int tmp1 = s_0_0 + s_0_2 - 2 * s_1_1 + s_2_0 + s_2_2;
    const int coeff1 = 3 * (tmp1 + s_0_1 - ((s_1_0 + s_1_2) * 2) + s_2_1);
    const int coeff2 = 3 * (tmp1 - ((s_0_1 + s_2_1) * 2) + s_1_0 + s_1_2);
    const int tmp2 = s_0_2 - s_2_0;
    const int tmp3 = (s_0_0 + tmp2 - s_2_2);
    const int tmp4 = tmp3 - 2 * tmp2;
    const int coeff3 = -3 * (tmp3 + s_0_1 - s_2_1);
    const int coeff4 = -3 * (tmp4 + s_1_0 - s_1_2);
    const int coeff5 = (s_0_0 - s_0_2 - s_2_0 + s_2_2) * 4;
    const int coeff6 = (-(s_0_0 + s_0_2 - ((s_1_0 + s_0_1 + s_1_2 + s_2_1) * 2) - 5 * s_1_1 + s_2_0 + s_2_2)) * 2;

    const int H_det = 4 * coeff1 * coeff2 - coeff5 * coeff5;
    if (H_det == 0) {
        *delta_x = 0.0f;
        *delta_y = 0.0f;
        return (float)((double)(float)coeff6 / 18.0);
    }
    if (!(H_det > 0 && coeff1 < 0)) { /* the maximum is at one of the four patch corners */
        int tmp_max = coeff3 + coeff4 + coeff5;
        *delta_x = 1.0f;
        *delta_y = 1.0f;
        int tmp = -coeff3 + coeff4 - coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            *delta_x = -1.0f;
            *delta_y = 1.0f;
        }
        tmp = coeff3 - coeff4 - coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            *delta_x = 1.0f;
            *delta_y = -1.0f;
        }
        tmp = -coeff3 - coeff4 + coeff5;
        if (tmp > tmp_max) {
            tmp_max = tmp;
            *delta_x = -1.0f;
            *delta_y = -1.0f;
        }
        return (float)((double)(float)(tmp_max + coeff1 + coeff2 + coeff6) / 18.0);
    }
    float dx = (float)FDIV((float)(2 * coeff2 * coeff3 - coeff4 * coeff5), (float)(-H_det));
    float dy = (float)FDIV((float)(2 * coeff1 * coeff4 - coeff3 * coeff5), (float)(-H_det));
    int tx = 0, tx_ = 0, ty = 0, ty_ = 0;
    if ((double)dx > 1.0)
        tx = 1;
    else if ((double)dx < -1.0)
        tx_ = 1;
    if ((double)dy > 1.0) ty = 1;
    if ((double)dy < -1.0) ty_ = 1;
    if (tx || tx_ || ty || ty_) {
        float dx1 = 0.0f, dx2 = 0.0f, dy1 = 0.0f, dy2 = 0.0f;
        if (tx) {
            dx1 = 1.0f;
            dy1 = (float)FDIV(-(float)(coeff4 + coeff5), (float)(2 * coeff2));
            if ((double)dy1 > 1.0) dy1 = 1.0f; else if ((double)dy1 < -1.0) dy1 = -1.0f;
        } else if (tx_) {
            dx1 = -1.0f;
            dy1 = (float)FDIV(-(float)(coeff4 - coeff5), (float)(2 * coeff2));
            if ((double)dy1 > 1.0) dy1 = 1.0f; else if ((double)dy1 < -1.0) dy1 = -1.0f;
        }
        if (ty) {
            dy2 = 1.0f;
            dx2 = (float)FDIV(-(float)(coeff3 + coeff5), (float)(2 * coeff1));
            if ((double)dx2 > 1.0) dx2 = 1.0f; else if ((double)dx2 < -1.0) dx2 = -1.0f;
        } else if (ty_) {
            dy2 = -1.0f;
            dx2 = (float)FDIV(-(float)(coeff3 - coeff5), (float)(2 * coeff1));
            if ((double)dx2 > 1.0) dx2 = 1.0f; else if ((double)dx2 < -1.0) dx2 = -1.0f;
        }
        /* int * float products, summed left to right as floats, divided in double (:1619-1626) */
#define QUAD(X, Y)                                                                                                                    \
    FADD(FADD(FADD(FADD(FADD(FMUL(FMUL((float)coeff1, X), X), FMUL(FMUL((float)coeff2, Y), Y)), FMUL((float)coeff3, X)), FMUL((float)coeff4, Y)), \
              FMUL(FMUL((float)coeff5, X), Y)),                                                                                       \
         (float)coeff6)
        const float max1 = (float)(QUAD(dx1, dy1) / 18.0);
        const float max2 = (float)(QUAD(dx2, dy2) / 18.0);
        if (max1 > max2) {
            *delta_x = dx1;
            *delta_y = dx1; /* sic (:1629) */
            return max1;
        }
        *delta_x = dx2;
        *delta_y = dx2; /* sic (:1634) */
        return max2;
    }
    *delta_x = dx;
    *delta_y = dy;
    return (float)(QUAD(dx, dy) / 18.0);
#undef QUAD
}

/* brisk.cpp:1418-1457 (variant 0), :1459-1497 (1), :1499-1533 (2) */
float mo_brisk_refine1d(int variant, float s_05, float s0, float s05, float *max)
{
    const int i_05 = (int)(1024.0 * (double)s_05 + 0.5);
    const int i0 = (int)(1024.0 * (double)s0 + 0.5);
    const int i05 = (int)(1024.0 * (double)s05 + 0.5);
    int a, b, c;
    double lo_d, hi_d, div;
    if (variant == 0) {
        a = 16 * i_05 - 24 * i0 + 8 * i05;
        b = -40 * i_05 + 54 * i0 - 14 * i05;
        c = +24 * i_05 - 27 * i0 + 6 * i05;
        lo_d = 0.75;
        hi_d = 1.5;
        div = 3072.0;
    } else if (variant == 1) {
        a = 9 * i_05 - 18 * i0 + 9 * i05;
        b = -21 * i_05 + 36 * i0 - 15 * i05;
        c = +12 * i_05 - 16 * i0 + 6 * i05;
        lo_d = 0.6666666666666666666666666667;
        hi_d = 1.33333333333333333333333333;
        div = 2048.0;
    } else {
        a = 2 * i_05 - 4 * i0 + 2 * i05;
        b = -5 * i_05 + 8 * i0 - 3 * i05;
        c = +3 * i_05 - 3 * i0 + 1 * i05;
        lo_d = 0.7;
        hi_d = 1.5;
        div = 1024.0;
    }
    if (a >= 0) { /* second derivative must be negative */
        if (s0 >= s_05 && s0 >= s05) {
            *max = s0;
            return 1.0f;
        }
        if (s_05 >= s0 && s_05 >= s05) {
            *max = s_05;
            return (float)lo_d;
        }
        if (s05 >= s0 && s05 >= s_05) {
            *max = s05;
            return (float)(variant == 1 ? 1.3333333333333333333333333333 : 1.5);
        }
    }
    float ret_val = (float)FDIV(-(float)b, (float)(2 * a));
    if ((double)ret_val < lo_d)
        ret_val = (float)lo_d;
    else if ((double)ret_val > hi_d)
        ret_val = (float)hi_d;
    float m = (float)FADD(FADD((float)c, FMUL(FMUL((float)a, ret_val), ret_val)), FMUL((float)b, ret_val));
    if (variant == 2)
        m = m / (float)1024; /* max/=1024 (:1531): int divisor, float division */
    else
        m = (float)((double)m / div);
    *max = m;
    return ret_val;
}

/* brisk.cpp:838-934 */
static int is_max_2d(const mo_brisk *b, int layer, int x_layer, int y_layer)
{
    const layer_t *l = &b->L[layer];
    const int cols = l->w;
    const uint8_t *sc = l->scores;
    const uint8_t *data = sc + (size_t)y_layer * cols + x_layer;
    const int center = data[0];
    const int s_10 = data[-1];
    if (center < s_10) return 0;
    const int s10 = data[1];
    if (center < s10) return 0;
    const int s0_1 = data[-cols];
    if (center < s0_1) return 0;
    const int s01 = data[cols];
    if (center < s01) return 0;
    const int s_11 = data[cols - 1];
    if (center < s_11) return 0;
    const int s11 = data[cols + 1];
    if (center < s11) return 0;
    const int s1_1 = data[-cols + 1];
    if (center < s1_1) return 0;
    const int s_1_1 = data[-cols - 1];
    if (center < s_1_1) return 0;

    int delta[16], nd = 0;
    if (center == s_1_1) { delta[nd++] = -1; delta[nd++] = -1; }
    if (center == s0_1) { delta[nd++] = 0; delta[nd++] = -1; }
    if (center == s1_1) { delta[nd++] = 1; delta[nd++] = -1; }
    if (center == s_10) { delta[nd++] = -1; delta[nd++] = 0; }
    if (center == s10) { delta[nd++] = 1; delta[nd++] = 0; }
    if (center == s_11) { delta[nd++] = -1; delta[nd++] = 1; }
    if (center == s01) { delta[nd++] = 0; delta[nd++] = 1; }
    if (center == s11) { delta[nd++] = 1; delta[nd++] = 1; }
    if (nd != 0) {
        const int smoothedcenter = 4 * center + 2 * (s_10 + s10 + s0_1 + s01) + s_1_1 + s1_1 + s_11 + s11;
        for (int i = 0; i < nd; i += 2) {
            const uint8_t *d = sc + (size_t)(y_layer - 1 + delta[i + 1]) * cols + x_layer + delta[i] - 1;
            int other = d[0] + 2 * d[1] + d[2];
            d += cols;
            other += 2 * d[0] + 4 * d[1] + 2 * d[2];
            d += cols;
            other += d[0] + 2 * d[1] + d[2];
            if (other > smoothedcenter) return 0;
        }
    }
    return 1;
}

static float patch_subpixel(layer_t *l, int x, int y, float *dx, float *dy)
{
    /* evaluation order of the nine getAgastScore calls as written at every call site (:614-622 etc.) */
    int s[9];
    s[0] = layer_score(l, x - 1, y - 1, 1); /* s_0_0 */
    s[3] = layer_score(l, x, y - 1, 1);     /* s_1_0 */
    s[6] = layer_score(l, x + 1, y - 1, 1); /* s_2_0 */
    s[7] = layer_score(l, x + 1, y, 1);     /* s_2_1 */
    s[4] = layer_score(l, x, y, 1);         /* s_1_1 */
    s[1] = layer_score(l, x - 1, y, 1);     /* s_0_1 */
    s[2] = layer_score(l, x - 1, y + 1, 1); /* s_0_2 */
    s[5] = layer_score(l, x, y + 1, 1);     /* s_1_2 */
    s[8] = layer_score(l, x + 1, y + 1, 1); /* s_2_2 */
    return mo_brisk_subpixel2d(s, dx, dy);
}

/* brisk.cpp:1106-1249 (above = 1) and :1251-1416 (above = 0): the two differ in the sampling geometry, in the
 * tie rule of the middle rows (below only) and in the final coordinate mapping. */
static float score_max_neighbour_layer(mo_brisk *b, int layer, int x_layer, int y_layer, int threshold, int *ismax, float *dx, float *dy,
                                       int above)
{
    *ismax = 0;
    float x_1, x1, y_1, y1;
    layer_t *lay = above ? &b->L[layer + 1] : &b->L[layer - 1];
    if (above) {
        if (layer % 2 == 0) { /* octave: double division */
            x_1 = (float)((double)(float)(4 * x_layer - 1 - 2) / 6.0);
            x1 = (float)((double)(float)(4 * x_layer - 1 + 2) / 6.0);
            y_1 = (float)((double)(float)(4 * y_layer - 1 - 2) / 6.0);
            y1 = (float)((double)(float)(4 * y_layer - 1 + 2) / 6.0);
        } else { /* intra: float division */
            x_1 = (float)(6 * x_layer - 1 - 3) / 8.0f;
            x1 = (float)(6 * x_layer - 1 + 3) / 8.0f;
            y_1 = (float)(6 * y_layer - 1 - 3) / 8.0f;
            y1 = (float)(6 * y_layer - 1 + 3) / 8.0f;
        }
    } else {
        if (layer % 2 == 0) {
            x_1 = (float)((double)(float)(8 * x_layer + 1 - 4) / 6.0);
            x1 = (float)((double)(float)(8 * x_layer + 1 + 4) / 6.0);
            y_1 = (float)((double)(float)(8 * y_layer + 1 - 4) / 6.0);
            y1 = (float)((double)(float)(8 * y_layer + 1 + 4) / 6.0);
        } else {
            x_1 = (float)((double)(float)(6 * x_layer + 1 - 3) / 4.0);
            x1 = (float)((double)(float)(6 * x_layer + 1 + 3) / 4.0);
            y_1 = (float)((double)(float)(6 * y_layer + 1 - 3) / 4.0);
            y1 = (float)((double)(float)(6 * y_layer + 1 + 3) / 4.0);
        }
    }

    /* first row */
    int max_x = (int)FADD(x_1, 1.0f);
    int max_y = (int)FADD(y_1, 1.0f);
    float tmp_max;
    float max = (float)layer_score_f(lay, x_1, y_1, 1);
    if (max > (float)threshold) return 0;
    for (int x = (int)FADD(x_1, 1.0f); x <= (int)x1; x++) {
        tmp_max = (float)layer_score_f(lay, (float)x, y_1, 1);
        if (tmp_max > (float)threshold) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = x;
        }
    }
    tmp_max = (float)layer_score_f(lay, x1, y_1, 1);
    if (tmp_max > (float)threshold) return 0;
    if (tmp_max > max) {
        max = tmp_max;
        max_x = (int)x1;
    }

    /* middle rows */
    for (int y = (int)FADD(y_1, 1.0f); y <= (int)y1; y++) {
        tmp_max = (float)layer_score_f(lay, x_1, (float)y, 1);
        if (tmp_max > (float)threshold) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = (int)FADD(x_1, 1.0f);
            max_y = y;
        }
        for (int x = (int)FADD(x_1, 1.0f); x <= (int)x1; x++) {
            tmp_max = (float)layer_score(lay, x, y, 1);
            if (tmp_max > (float)threshold) return 0;
            if (!above && tmp_max == max) { /* :1321-1344 */
                const int t1 = 2 * (layer_score(lay, x - 1, y, 1) + layer_score(lay, x + 1, y, 1) + layer_score(lay, x, y + 1, 1) +
                                    layer_score(lay, x, y - 1, 1)) +
                               (layer_score(lay, x + 1, y + 1, 1) + layer_score(lay, x - 1, y + 1, 1) + layer_score(lay, x + 1, y - 1, 1) +
                                layer_score(lay, x - 1, y - 1, 1));
                const int t2 = 2 * (layer_score(lay, max_x - 1, max_y, 1) + layer_score(lay, max_x + 1, max_y, 1) +
                                    layer_score(lay, max_x, max_y + 1, 1) + layer_score(lay, max_x, max_y - 1, 1)) +
                               (layer_score(lay, max_x + 1, max_y + 1, 1) + layer_score(lay, max_x - 1, max_y + 1, 1) +
                                layer_score(lay, max_x + 1, max_y - 1, 1) + layer_score(lay, max_x - 1, max_y - 1, 1));
                if (t1 > t2) {
                    max_x = x;
                    max_y = y;
                }
            }
            if (tmp_max > max) {
                max = tmp_max;
                max_x = x;
                max_y = y;
            }
        }
        tmp_max = (float)layer_score_f(lay, x1, (float)y, 1);
        if (tmp_max > (float)threshold) return 0;
        if (tmp_max > max) {
            max = tmp_max;
            max_x = (int)x1;
            max_y = y;
        }
    }

    /* bottom row: no early exit (:1185-1205) */
    tmp_max = (float)layer_score_f(lay, x_1, y1, 1);
    if (tmp_max > max) {
        max = tmp_max;
        max_x = (int)FADD(x_1, 1.0f);
        max_y = (int)y1;
    }
    for (int x = (int)FADD(x_1, 1.0f); x <= (int)x1; x++) {
        tmp_max = (float)layer_score_f(lay, (float)x, y1, 1);
        if (tmp_max > max) {
            max = tmp_max;
            max_x = x;
            max_y = (int)y1;
        }
    }
    tmp_max = (float)layer_score_f(lay, x1, y1, 1);
    if (tmp_max > max) {
        max = tmp_max;
        max_x = (int)x1;
        max_y = (int)y1;
    }

    float dx_1, dy_1;
    const float refined_max = patch_subpixel(lay, max_x, max_y, &dx_1, &dy_1);

    const float real_x = (float)FADD((float)max_x, dx_1);
    const float real_y = (float)FADD((float)max_y, dy_1);
    int returnrefined = 1;
    if (above) {
        if (layer % 2 == 0) { /* float arithmetic (:1228-1229) */
            *dx = (float)FSUB(FDIV(FADD(FMUL(real_x, 6.0f), 1.0f), 4.0f), (float)x_layer);
            *dy = (float)FSUB(FDIV(FADD(FMUL(real_y, 6.0f), 1.0f), 4.0f), (float)y_layer);
        } else { /* double arithmetic (:1232-1233) */
            *dx = (float)(((double)real_x * 8.0 + 1.0) / 6.0 - (double)(float)x_layer);
            *dy = (float)(((double)real_y * 8.0 + 1.0) / 6.0 - (double)(float)y_layer);
        }
    } else {
        if (layer % 2 == 0) {
            *dx = (float)(((double)real_x * 6.0 + 1.0) / 8.0 - (double)(float)x_layer);
            *dy = (float)(((double)real_y * 6.0 + 1.0) / 8.0 - (double)(float)y_layer);
        } else {
            *dx = (float)(((double)real_x * 4.0 - 1.0) / 6.0 - (double)(float)x_layer);
            *dy = (float)(((double)real_y * 4.0 - 1.0) / 6.0 - (double)(float)y_layer);
        }
    }
    if (*dx > 1.0f) { *dx = 1.0f; returnrefined = 0; }
    if (*dx < -1.0f) { *dx = -1.0f; returnrefined = 0; }
    if (*dy > 1.0f) { *dy = 1.0f; returnrefined = 0; }
    if (*dy < -1.0f) { *dy = -1.0f; returnrefined = 0; }

    *ismax = 1;
    if (returnrefined) return refined_max > max ? refined_max : max; /* std::max(refined_max, max) */
    return max;
}

/* brisk.cpp:937-1103 */
/* (r0 * delta_layer + r1 * delta_other + float(c)) [* scale + offset]: ONE expression each in the reference (:1022-1092) */
#define BLEND(R0, DL, R1, DO, C) FADD(FADD(FMUL(R0, DL), FMUL(R1, DO)), (float)(C))
#define PLACE(R0, DL, R1, DO, C) ((float)FADD(FMUL(BLEND(R0, DL, R1, DO, C), thisLayer->scale), thisLayer->offset))
static float refine_3d(mo_brisk *b, int layer, int x_layer, int y_layer, float *x, float *y, float *scale, int *ismax)
{
    *ismax = 1;
    layer_t *thisLayer = &b->L[layer];
    const int center = layer_score(thisLayer, x_layer, y_layer, 1);

    float delta_x_above, delta_y_above;
    const float max_above = score_max_neighbour_layer(b, layer, x_layer, y_layer, center, ismax, &delta_x_above, &delta_y_above, 1);
    if (!*ismax) return 0.0f;

    float max;
    float delta_x_below, delta_y_below, delta_x_layer, delta_y_layer;
    if (layer % 2 == 0) { /* on octave */
        float max_below_float;
        if (layer == 0) { /* guess the lower intra octave with the 5/8 mask (:959-989) */
            int s[9];
            int max_below_uchar;
            s[0] = layer_score_5_8(thisLayer, x_layer - 1, y_layer - 1, 1);
            max_below_uchar = s[0];
            s[3] = layer_score_5_8(thisLayer, x_layer, y_layer - 1, 1);
            if (s[3] > max_below_uchar) max_below_uchar = s[3];
            s[6] = layer_score_5_8(thisLayer, x_layer + 1, y_layer - 1, 1);
            if (s[6] > max_below_uchar) max_below_uchar = s[6];
            s[7] = layer_score_5_8(thisLayer, x_layer + 1, y_layer, 1);
            if (s[7] > max_below_uchar) max_below_uchar = s[7];
            s[4] = layer_score_5_8(thisLayer, x_layer, y_layer, 1);
            if (s[4] > max_below_uchar) max_below_uchar = s[4];
            s[1] = layer_score_5_8(thisLayer, x_layer - 1, y_layer, 1);
            if (s[1] > max_below_uchar) max_below_uchar = s[1];
            s[2] = layer_score_5_8(thisLayer, x_layer - 1, y_layer + 1, 1);
            if (s[2] > max_below_uchar) max_below_uchar = s[2];
            s[5] = layer_score_5_8(thisLayer, x_layer, y_layer + 1, 1);
            if (s[5] > max_below_uchar) max_below_uchar = s[5];
            s[8] = layer_score_5_8(thisLayer, x_layer + 1, y_layer + 1, 1);
            if (s[8] > max_below_uchar) max_below_uchar = s[8];
            (void)mo_brisk_subpixel2d(s, &delta_x_below, &delta_y_below);
            max_below_float = (float)max_below_uchar;
        } else {
            max_below_float = score_max_neighbour_layer(b, layer, x_layer, y_layer, center, ismax, &delta_x_below, &delta_y_below, 0);
            if (!*ismax) return 0;
        }
        const float max_layer = patch_subpixel(thisLayer, x_layer, y_layer, &delta_x_layer, &delta_y_layer);
        const float s0m = ((float)center < max_layer) ? max_layer : (float)center; /* std::max(float(center), max_layer) */
        if (layer == 0)
            *scale = mo_brisk_refine1d(2, max_below_float, s0m, max_above, &max);
        else
            *scale = mo_brisk_refine1d(0, max_below_float, s0m, max_above, &max);

        if ((double)*scale > 1.0) {
            const float r0 = (float)((1.5 - (double)*scale) / .5);
            const float r1 = (float)(1.0 - (double)r0);
            *x = PLACE(r0, delta_x_layer, r1, delta_x_above, x_layer);
            *y = PLACE(r0, delta_y_layer, r1, delta_y_above, y_layer);
        } else if (layer == 0) {
            const float r0 = (float)(((double)*scale - 0.5) / 0.5);
            const float r_1 = (float)(1.0 - (double)r0);
            *x = (float)BLEND(r0, delta_x_layer, r_1, delta_x_below, x_layer);
            *y = (float)BLEND(r0, delta_y_layer, r_1, delta_y_below, y_layer);
        } else {
            const float r0 = (float)(((double)*scale - 0.75) / 0.25);
            const float r_1 = (float)(1.0 - (double)r0);
            *x = PLACE(r0, delta_x_layer, r_1, delta_x_below, x_layer);
            *y = PLACE(r0, delta_y_layer, r_1, delta_y_below, y_layer);
        }
    } else { /* on intra */
        const float max_below = score_max_neighbour_layer(b, layer, x_layer, y_layer, center, ismax, &delta_x_below, &delta_y_below, 0);
        if (!*ismax) return 0.0f;
        const float max_layer = patch_subpixel(thisLayer, x_layer, y_layer, &delta_x_layer, &delta_y_layer);
        const float s0m = ((float)center < max_layer) ? max_layer : (float)center;
        *scale = mo_brisk_refine1d(1, max_below, s0m, max_above, &max);
        if ((double)*scale > 1.0) {
            const float r0 = (float)(4.0 - (double)*scale * 3.0);
            const float r1 = (float)(1.0 - (double)r0);
            *x = PLACE(r0, delta_x_layer, r1, delta_x_above, x_layer);
            *y = PLACE(r0, delta_y_layer, r1, delta_y_above, y_layer);
        } else {
            const float r0 = (float)((double)*scale * 3.0 - 2.0);
            const float r_1 = (float)(1.0 - (double)r0);
            *x = PLACE(r0, delta_x_layer, r_1, delta_x_below, x_layer);
            *y = PLACE(r0, delta_y_layer, r_1, delta_y_below, y_layer);
        }
    }
    *scale *= thisLayer->scale;
    return max;
}

static void emit(mo_brisk_keypoint *out, int cap, int *n, float x, float y, float size, float response, int layer)
{
    if (*n < cap) {
        out[*n].x = x;
        out[*n].y = y;
        out[*n].size = size;
        out[*n].response = response;
        out[*n].layer = layer;
    }
    ++*n;
}

/* brisk.cpp:590-704 */
int mo_brisk_get_keypoints(mo_brisk *b, int threshold, mo_brisk_keypoint *out, int cap)
{
    static const float basicSize = 12.0f, safetyFactor = 1.0f; /* :58-59 */
    int n = 0;
    b->threshold = threshold & 0xff;
    b->safe_threshold = (uint8_t)((float)b->threshold * safetyFactor);

    /* getAgastPoints on every layer (:600-607, 1676-1689): detect at the safe threshold, cache those scores */
    for (int i = 0; i < b->layers; ++i) {
        layer_t *l = &b->L[i];
        free(l->pts);
        const int np = mo_oast9_16_detect(l->img, l->w, l->h, b->safe_threshold, NULL, 0);
        l->pts = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(np > 0 ? np : 1));
        l->n_pts = mo_oast9_16_detect(l->img, l->w, l->h, b->safe_threshold, l->pts, np);
        for (int k = 0; k < l->n_pts; ++k) {
            const size_t offs = (size_t)l->pts[2 * k] + (size_t)l->pts[2 * k + 1] * l->w;
            l->scores[offs] = (uint8_t)mo_oast9_16_score(l->img + offs, l->w, b->safe_threshold);
        }
    }

    if (b->layers == 1) { /* :609-638 */
        layer_t *l = &b->L[0];
        for (int k = 0; k < l->n_pts; ++k) {
            const int px = l->pts[2 * k], py = l->pts[2 * k + 1];
            if (!is_max_2d(b, 0, px, py)) continue;
            float dx, dy;
            const float max = patch_subpixel(l, px, py, &dx, &dy);
            emit(out, cap, &n, (float)px + dx, (float)py + dy, basicSize, max, 0);
        }
        return n;
    }

    for (int i = 0; i < b->layers; ++i) {
        layer_t *l = &b->L[i];
        for (int k = 0; k < l->n_pts; ++k) {
            const int px = l->pts[2 * k], py = l->pts[2 * k + 1];
            if (!is_max_2d(b, i, px, py)) continue;
            if (i == b->layers - 1) { /* :644-679 */
                int ismax;
                float dx, dy;
                (void)score_max_neighbour_layer(b, i, px, py, layer_score(l, px, py, b->safe_threshold), &ismax, &dx, &dy, 0);
                if (!ismax) continue;
                float delta_x, delta_y;
                const float max = patch_subpixel(l, px, py, &delta_x, &delta_y);
                emit(out, cap, &n, (float)FADD(FMUL(FADD((float)px, delta_x), l->scale), l->offset),
                     (float)FADD(FMUL(FADD((float)py, delta_y), l->scale), l->offset), basicSize * l->scale, max, i);
            } else { /* :681-701 */
                int ismax;
                float x, y, scale;
                const float score = refine_3d(b, i, px, py, &x, &y, &scale, &ismax);
                if (!ismax) continue;
                if (score > (float)b->threshold) emit(out, cap, &n, x, y, basicSize * scale, score, i);
            }
        }
    }
    return n;
}

int mo_brisk_detect(const uint8_t *img, int stride, int w, int h, int threshold, int octaves, mo_brisk_keypoint *out, int cap)
{
    mo_brisk *b = mo_brisk_create(img, stride, w, h, octaves);
    const int n = mo_brisk_get_keypoints(b, threshold, out, cap);
    mo_brisk_destroy(b);
    return n;
}
