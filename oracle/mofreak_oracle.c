/*
 * mofreak_oracle.c -- CPU restatement of the MoFREAK descriptor path.  TEST INFRASTRUCTURE ONLY
 * (see mofreak_oracle.h: PARITY UNPINNED, never linked or called by the product).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Every float/double step below is written in the evaluation order of the reference expression it
 * restates; do not "simplify" the arithmetic.
 */
#include "mofreak_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* OpenCV's CV_PI and freak.cpp's FREAK_LOG2. */
#define MO_PI 3.1415926535897932384626433832795
#define MO_LOG2 0.693147180559945

/* FREAK_DEF_PAIRS of OpenCV 2.4.x features2d/src/freak.cpp (indices into the 903 (i>j) pairs). */
static const int MO_DEF_PAIRS[MO_NB_PAIRS] = {
    404,431,818,511,181,52,311,874,774,543,719,230,417,205,11,
    560,149,265,39,306,165,857,250,8,61,15,55,717,44,412,
    592,134,761,695,660,782,625,487,549,516,271,665,762,392,178,
    796,773,31,672,845,548,794,677,654,241,831,225,238,849,83,
    691,484,826,707,122,517,583,731,328,339,571,475,394,472,580,
    381,137,93,380,327,619,729,808,218,213,459,141,806,341,95,
    382,568,124,750,193,749,706,843,79,199,317,329,768,198,100,
    466,613,78,562,783,689,136,838,94,142,164,679,219,419,366,
    418,423,77,89,523,259,683,312,555,20,470,684,123,458,453,833,
    72,113,253,108,313,25,153,648,411,607,618,128,305,232,301,84,
    56,264,371,46,407,360,38,99,176,710,114,578,66,372,653,
    129,359,424,159,821,10,323,393,5,340,891,9,790,47,0,175,346,
    236,26,172,147,574,561,32,294,429,724,755,398,787,288,299,
    769,565,767,722,757,224,465,723,498,467,235,127,802,446,233,
    544,482,800,318,16,532,801,441,554,173,60,530,713,469,30,
    212,630,899,170,266,799,88,49,512,399,23,500,107,524,90,
    194,143,135,192,206,345,148,71,119,101,563,870,158,254,214,
    276,464,332,725,188,385,24,476,40,231,620,171,258,67,109,
    844,244,187,388,701,690,50,7,850,479,48,522,22,154,12,659,
    736,655,577,737,830,811,174,21,237,335,353,234,53,270,62,
    182,45,177,245,812,673,355,556,612,166,204,54,248,365,226,
    242,452,700,685,573,14,842,481,468,781,564,416,179,405,35,
    819,608,624,367,98,643,448,2,460,676,440,240,130,146,184,
    185,430,65,807,377,82,121,708,239,310,138,596,730,575,477,
    851,797,247,27,85,586,307,779,326,494,856,324,827,96,748,
    13,397,125,688,702,92,293,716,277,140,112,4,80,855,839,1,
    413,347,584,493,289,696,19,751,379,76,73,115,6,590,183,734,
    197,483,217,344,330,400,186,243,587,220,780,200,793,246,824,
    41,735,579,81,703,322,760,720,139,480,490,91,814,813,163,
    152,488,763,263,425,410,576,120,319,668,150,160,302,491,515,
    260,145,428,97,251,395,272,252,18,106,358,854,485,144,550,
    131,133,378,68,102,104,58,361,275,209,697,582,338,742,589,
    325,408,229,28,304,191,189,110,126,486,211,547,533,70,215,
    670,249,36,581,389,605,331,518,442,822
};

/* orientationPairs of freak.cpp buildPattern(). */
static const uint8_t MO_ORIENT_IJ[MO_NB_ORIENPAIRS][2] = {
    {0,3},{1,4},{2,5},{0,2},{1,3},{2,4},{3,5},{4,0},{5,1},
    {6,9},{7,10},{8,11},{6,8},{7,9},{8,10},{9,11},{10,6},{11,7},
    {12,15},{13,16},{14,17},{12,14},{13,15},{14,16},{15,17},{16,12},{17,13},
    {18,21},{19,22},{20,23},{18,20},{19,21},{20,22},{21,23},{22,18},{23,19},
    {24,27},{25,28},{26,29},{30,33},{31,34},{32,35},{36,39},{37,40},{38,41}
};

/* ---------------------------------------------------------------- FREAK::buildPattern */
mo_freak *mo_freak_create(float pattern_scale, int n_octaves, int orientation_normalized,
                          int scale_normalized, int bit_mode)
{
    mo_freak *f = (mo_freak *)calloc(1, sizeof(mo_freak));
    if (!f) return NULL;
    f->pattern_scale = pattern_scale;
    f->n_octaves = n_octaves;
    f->orientation_normalized = orientation_normalized;
    f->scale_normalized = scale_normalized;
    f->bit_mode = bit_mode;
    f->lut = (mo_pattern_point *)malloc(sizeof(mo_pattern_point) * MO_NB_SCALES * MO_NB_ORIENTATION * MO_NB_POINTS);
    if (!f->lut) { free(f); return NULL; }

    const double scaleStep = pow(2.0, (double)(n_octaves) / MO_NB_SCALES);
    const int n[8] = {6, 6, 6, 6, 6, 6, 6, 1};
    const double bigR = 2.0 / 3.0;
    const double smallR = 2.0 / 24.0;
    const double unitSpace = (bigR - smallR) / 21.0;
    const double radius[8] = {bigR, bigR - 6 * unitSpace, bigR - 11 * unitSpace, bigR - 15 * unitSpace,
                              bigR - 18 * unitSpace, bigR - 20 * unitSpace, smallR, 0.0};
    const double sigma[8] = {radius[0] / 2.0, radius[1] / 2.0, radius[2] / 2.0, radius[3] / 2.0,
                             radius[4] / 2.0, radius[5] / 2.0, radius[6] / 2.0, radius[6] / 2.0};

    for (int scaleIdx = 0; scaleIdx < MO_NB_SCALES; ++scaleIdx) {
        f->pattern_sizes[scaleIdx] = 0;
        const double scalingFactor = pow(scaleStep, scaleIdx);
        for (int orientationIdx = 0; orientationIdx < MO_NB_ORIENTATION; ++orientationIdx) {
            const double theta = (double)orientationIdx * 2 * MO_PI / (double)MO_NB_ORIENTATION;
            int pointIdx = 0;
            for (int i = 0; i < 8; ++i) {
                for (int k = 0; k < n[i]; ++k) {
                    const double beta = M_PI / n[i] * (i % 2);
                    const double alpha = (double)k * 2 * M_PI / (double)n[i] + beta + theta;
                    mo_pattern_point *p = &f->lut[(scaleIdx * MO_NB_ORIENTATION + orientationIdx) * MO_NB_POINTS + pointIdx];
                    p->x = (float)(radius[i] * cos(alpha) * scalingFactor * pattern_scale);
                    p->y = (float)(radius[i] * sin(alpha) * scalingFactor * pattern_scale);
                    p->sigma = (float)(sigma[i] * scalingFactor * pattern_scale);
                    const int sizeMax = (int)ceil((radius[i] + sigma[i]) * scalingFactor * pattern_scale) + 1;
                    if (f->pattern_sizes[scaleIdx] < sizeMax) f->pattern_sizes[scaleIdx] = sizeMax;
                    ++pointIdx;
                }
            }
        }
    }

    for (int m = MO_NB_ORIENPAIRS; m--;) {
        f->orient[m].i = MO_ORIENT_IJ[m][0];
        f->orient[m].j = MO_ORIENT_IJ[m][1];
        const float dx = f->lut[f->orient[m].i].x - f->lut[f->orient[m].j].x;
        const float dy = f->lut[f->orient[m].i].y - f->lut[f->orient[m].j].y;
        const float norm_sq = (dx * dx + dy * dy);
        f->orient[m].weight_dx = (int)((dx / (norm_sq)) * 4096.0 + 0.5);
        f->orient[m].weight_dy = (int)((dy / (norm_sq)) * 4096.0 + 0.5);
    }

    /* allPairs: for i in 1..42, j in 0..i-1; descriptionPairs[m] = allPairs[DEF_PAIRS[m]]. */
    mo_desc_pair all[903];
    int cnt = 0;
    for (unsigned i = 1; i < MO_NB_POINTS; ++i)
        for (unsigned j = 0; j < i; ++j) { all[cnt].i = (uint8_t)i; all[cnt].j = (uint8_t)j; ++cnt; }
    for (int m = 0; m < MO_NB_PAIRS; ++m) f->pairs[m] = all[MO_DEF_PAIRS[m]];
    return f;
}

void mo_freak_get_pairs(const mo_freak *f, uint8_t *out_ij)
{
    for (int m = 0; m < MO_NB_PAIRS; ++m) { out_ij[2 * m] = f->pairs[m].i; out_ij[2 * m + 1] = f->pairs[m].j; }
}

void mo_freak_get_orientation(const mo_freak *f, int *out)
{
    for (int m = 0; m < MO_NB_ORIENPAIRS; ++m) {
        out[4 * m] = f->orient[m].i; out[4 * m + 1] = f->orient[m].j;
        out[4 * m + 2] = f->orient[m].weight_dx; out[4 * m + 3] = f->orient[m].weight_dy;
    }
}

void mo_freak_get_pattern(const mo_freak *f, int scale, int rot, float *out)
{
    const mo_pattern_point *p = &f->lut[(scale * MO_NB_ORIENTATION + rot) * MO_NB_POINTS];
    for (int i = 0; i < MO_NB_POINTS; ++i) { out[3 * i] = p[i].x; out[3 * i + 1] = p[i].y; out[3 * i + 2] = p[i].sigma; }
}

void mo_freak_destroy(mo_freak *f)
{
    if (!f) return;
    free(f->lut);
    free(f);
}

/* ---------------------------------------------------------------- frame prep */
void mo_bgr2gray(const uint8_t *bgr, int W, int H, uint8_t *gray)
{
    /* cv::cvtColor(frame, frame, CV_BGR2GRAY) on 8UC3 (MoFREAKUtilities.cpp:395, :410), OpenCV 2.4.x
     * imgproc/src/color.cpp RGB2Gray<uchar>: fixed point, B2Y = 1868, G2Y = 9617, R2Y = 4899, yuv_shift = 14,
     * CV_DESCALE(x, n) = (x + (1 << (n-1))) >> n.  [UPSTREAM: not under /root/reference; parity unpinned] */
    for (long i = 0; i < (long)W * H; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}

void mo_absdiff(const uint8_t *a, const uint8_t *b, uint8_t *d, int W, int H)
{
    for (long i = 0; i < (long)W * H; ++i) d[i] = (uint8_t)(a[i] > b[i] ? a[i] - b[i] : b[i] - a[i]);
}

void mo_integral(const uint8_t *img, int W, int H, int32_t *integ)
{
    const int S = W + 1;
    for (int x = 0; x <= W; ++x) integ[x] = 0;
    for (int y = 0; y < H; ++y) {
        int32_t rowsum = 0;
        integ[(y + 1) * S] = 0;
        for (int x = 0; x < W; ++x) {
            rowsum += img[y * W + x];
            integ[(y + 1) * S + x + 1] = integ[y * S + x + 1] + rowsum;
        }
    }
}

/* ---------------------------------------------------------------- FREAK::computeImpl pieces */
int mo_freak_scale_index(const mo_freak *f, float size)
{
    /* kpScaleIdx = max((int)(log(size/FREAK_SMALLEST_KP_SIZE)*sizeCst+0.5), 0), clamped to 63.
     * log(float) is the float overload = (float)log((double)) on the reference's MSVC x86 CRT. */
    const float sizeCst = (float)(MO_NB_SCALES / (MO_LOG2 * f->n_octaves));
    const float ratio = size / MO_SMALLEST_KP_SIZE;
    const float lg = (float)log((double)ratio);
    int idx = (int)(lg * sizeCst + 0.5);
    if (idx < 0) idx = 0;
    if (idx >= MO_NB_SCALES) idx = MO_NB_SCALES - 1;
    return idx;
}

static int theta_from_angle(float angle)
{
    int thetaIdx = (int)(MO_NB_ORIENTATION * angle * (1 / 360.0) + 0.5);
    if (thetaIdx < 0) thetaIdx += MO_NB_ORIENTATION;
    if (thetaIdx >= MO_NB_ORIENTATION) thetaIdx -= MO_NB_ORIENTATION;
    return thetaIdx;
}

int mo_freak_theta_index(int direction0, int direction1)
{
    /* angle = static_cast<float>(atan2((float)direction1,(float)direction0)*(180.0/CV_PI));
     * atan2(float,float) is the float overload; on the reference's x86 MSVC CRT that is
     * (float)atan2((double),(double)), i.e. the double result rounded once to float. */
    const float a = (float)atan2((double)(float)direction1, (double)(float)direction0);
    const float angle = (float)(a * (180.0 / MO_PI));
    return theta_from_angle(angle);
}

int mo_freak_theta_index_atan2f(int direction0, int direction1)
{
    const float a = atan2f((float)direction1, (float)direction0);
    const float angle = (float)(a * (180.0 / MO_PI));
    return theta_from_angle(angle);
}

uint8_t mo_freak_mean_intensity(const mo_freak *f, const uint8_t *img, const int32_t *integ, int W, int H,
                                float kp_x, float kp_y, unsigned scale, unsigned rot, unsigned point)
{
    (void)H;
    const mo_pattern_point *P = &f->lut[(scale * MO_NB_ORIENTATION + rot) * MO_NB_POINTS + point];
    const float xf = P->x + kp_x;
    const float yf = P->y + kp_y;
    const int x = (int)xf;
    const int y = (int)yf;
    const int imagecols = W;
    const float radius = P->sigma;

    if (radius < 0.5) {
        /* never taken with the default patternScale 22 (all sigmas >= 0.917 px) */
        const int r_x = (int)((xf - x) * 1024);
        const int r_y = (int)((yf - y) * 1024);
        const int r_x_1 = (1024 - r_x);
        const int r_y_1 = (1024 - r_y);
        const uint8_t *ptr = img + x + y * imagecols;
        unsigned int ret_val;
        ret_val = (unsigned)(r_x_1 * r_y_1 * (int)(*ptr));
        ptr++;
        ret_val += (unsigned)(r_x * r_y_1 * (int)(*ptr));
        ptr += imagecols;
        ret_val += (unsigned)(r_x * r_y * (int)(*ptr));
        ptr--;
        ret_val += (unsigned)(r_x_1 * r_y * (int)(*ptr));
        ret_val += 2 * 1024 * 1024;
        return (uint8_t)(ret_val / (4 * 1024 * 1024));
    }

    const int x_left = (int)(xf - radius + 0.5);
    const int y_top = (int)(yf - radius + 0.5);
    const int x_right = (int)(xf + radius + 1.5);
    const int y_bottom = (int)(yf + radius + 1.5);
    const int S = W + 1;
    int ret_val;
    ret_val = integ[y_bottom * S + x_right];
    ret_val -= integ[y_bottom * S + x_left];
    ret_val += integ[y_top * S + x_left];
    ret_val -= integ[y_top * S + x_right];
    ret_val = ret_val / ((x_right - x_left) * (y_bottom - y_top));
    return (uint8_t)ret_val;
}

void mo_freak_compute(const mo_freak *f, const uint8_t *img, int W, int H, const float *kps, int n,
                      uint8_t *valid, uint8_t *desc64, int *theta_out, int *dir_out)
{
    int32_t *integ = (int32_t *)malloc(sizeof(int32_t) * (size_t)(W + 1) * (H + 1));
    mo_integral(img, W, H, integ);
    memset(desc64, 0, (size_t)n * 64);

    for (int k = 0; k < n; ++k) {
        const float kx = kps[3 * k], ky = kps[3 * k + 1], size = kps[3 * k + 2];
        valid[k] = 0;
        if (theta_out) theta_out[k] = -1;
        if (dir_out) { dir_out[2 * k] = 0; dir_out[2 * k + 1] = 0; }
        /* DescriptorExtractor::compute: runByImageBorder(.., 0) is a no-op; runByKeypointSize drops size < eps. */
        if (size < FLT_EPSILON) continue;
        /* non-finite coordinates or sizes are undefined behaviour in the reference ((int) of a NaN);
         * both the oracle and the product treat them as erased */
        if (!isfinite(size) || !isfinite(kx) || !isfinite(ky)) continue;
        int idx;
        if (f->scale_normalized) {
            idx = mo_freak_scale_index(f, size);
        } else {
            const float sizeCst = (float)(MO_NB_SCALES / (MO_LOG2 * f->n_octaves));
            const int scIdx = (int)(1.0986122886681 * sizeCst + 0.5);
            idx = scIdx < 0 ? 0 : scIdx;
            if (idx >= MO_NB_SCALES) idx = MO_NB_SCALES - 1;
        }
        const int ps = f->pattern_sizes[idx];
        if (kx <= ps || ky <= ps || kx >= W - ps || ky >= H - ps) continue;
        valid[k] = 1;

        uint8_t v[MO_NB_POINTS];
        int thetaIdx = 0;
        if (f->orientation_normalized) {
            for (int i = MO_NB_POINTS; i--;)
                v[i] = mo_freak_mean_intensity(f, img, integ, W, H, kx, ky, (unsigned)idx, 0, (unsigned)i);
            int direction0 = 0, direction1 = 0;
            for (int m = MO_NB_ORIENPAIRS; m--;) {
                const int delta = (v[f->orient[m].i] - v[f->orient[m].j]);
                direction0 += delta * (f->orient[m].weight_dx) / 2048;
                direction1 += delta * (f->orient[m].weight_dy) / 2048;
            }
            thetaIdx = mo_freak_theta_index(direction0, direction1);
            if (dir_out) { dir_out[2 * k] = direction0; dir_out[2 * k + 1] = direction1; }
        }
        if (theta_out) theta_out[k] = thetaIdx;
        for (int i = MO_NB_POINTS; i--;)
            v[i] = mo_freak_mean_intensity(f, img, integ, W, H, kx, ky, (unsigned)idx, (unsigned)thetaIdx, (unsigned)i);

        uint8_t *d = desc64 + (size_t)k * 64;
        if (f->bit_mode == MO_BITS_NATURAL) {
            for (int m = 0; m < MO_NB_PAIRS; ++m)
                if (v[f->pairs[m].i] > v[f->pairs[m].j]) d[m >> 3] |= (uint8_t)(1u << (m & 7));
        } else {
            /* SSE layout: for 128-pair block nb, 16-pair group s (mask 0x8080 >> (7-s) = bit s),
             * _mm_set_epi8 puts pair cnt+t into byte 15-t. */
            int cnt = 0;
            for (int nb = 0; nb < MO_NB_PAIRS / 128; ++nb) {
                for (int s = 0; s < 8; ++s, cnt += 16) {
                    for (int t = 0; t < 16; ++t) {
                        const uint8_t a = v[f->pairs[cnt + t].i], b = v[f->pairs[cnt + t].j];
                        const int bit = (f->bit_mode == MO_BITS_SSE_SIGNED) ? ((int8_t)a > (int8_t)b) : (a >= b);
                        if (bit) d[16 * nb + (15 - t)] |= (uint8_t)(1u << s);
                    }
                }
            }
        }
    }
    free(integ);
}

/* ---------------------------------------------------------------- cv::resize INTER_LINEAR 8UC1 */
static int round_half_even_f(float v) { return (int)lrintf(v); } /* cvRound under the default rounding mode */

int mo_resize_axis_table(int ssize, int dsize, int is_x, int *ofs, short *coef)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    int dmax = dsize;
    for (int d = 0; d < dsize; ++d) {
        float fr = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floor((double)fr);
        fr -= s;
        if (is_x) {
            if (s < 0) { fr = 0; s = 0; }
            if (s + 1 >= ssize) {
                if (dmax > d) dmax = d;
                if (s >= ssize - 1) { fr = 0; s = ssize - 1; }
            }
        }
        ofs[d] = s;
        const float c0 = 1.f - fr, c1 = fr;
        int i0 = round_half_even_f(c0 * 2048), i1 = round_half_even_f(c1 * 2048);
        coef[2 * d] = (short)(i0 > 32767 ? 32767 : (i0 < -32768 ? -32768 : i0));
        coef[2 * d + 1] = (short)(i1 > 32767 ? 32767 : (i1 < -32768 ? -32768 : i1));
    }
    return dmax;
}

void mo_resize_linear_8u(const uint8_t *src, int sstride, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw), *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
    int *T0 = (int *)malloc(sizeof(int) * dw), *T1 = (int *)malloc(sizeof(int) * dw);
    const int xmax = mo_resize_axis_table(sw, dw, 1, xofs, ialpha);
    mo_resize_axis_table(sh, dh, 0, yofs, ibeta);

    for (int dy = 0; dy < dh; ++dy) {
        int r0 = yofs[dy], r1 = yofs[dy] + 1;
        r0 = r0 >= 0 ? (r0 < sh ? r0 : sh - 1) : 0;
        r1 = r1 >= 0 ? (r1 < sh ? r1 : sh - 1) : 0;
        const uint8_t *S0 = src + (size_t)r0 * sstride, *S1 = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            if (dx < xmax) {
                const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
                T0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
                T1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
            } else {
                T0[dx] = S0[sx] * 2048;
                T1[dx] = S1[sx] * 2048;
            }
        }
        const short b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        for (int x = 0; x < dw; ++x)
            dst[dy * dw + x] = (uint8_t)((((b0 * (T0[x] >> 4)) >> 16) + ((b1 * (T1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(ialpha); free(ibeta); free(T0); free(T1);
}

/* ---------------------------------------------------------------- MIP (in-tree reference code) */
unsigned mo_mip(const uint8_t *cur19, const uint8_t *prev19, int x, int y)
{
    /* MoFREAKUtilities.cpp:46-99.  patch_t.data / it->data point at the ROI's top-left inside the
     * 19-byte-stride parent buffer and are advanced 9 times with p++: 9 CONTIGUOUS bytes, not 3x3. */
    static const int OFF[8][2] = {{-4, 0}, {-3, 3}, {0, 4}, {3, 3}, {4, 0}, {3, -3}, {0, -4}, {-3, -3}};
    const int THETA = 288;
    const uint8_t *pt = cur19 + (y - 1) * 19 + (x - 1);
    unsigned bit = 1, descriptor = 0;
    for (int i = 0; i < 8; ++i) {
        const uint8_t *p = pt;
        const uint8_t *p2 = prev19 + ((y + OFF[i][1]) - 1) * 19 + ((x + OFF[i][0]) - 1);
        int ssd = 0;
        for (int k = 0; k < 9; ++k) {
            ssd += (int)powf((float)((*p) - (*p2)), 2);
            p++;
            p2++;
        }
        if (ssd > THETA) descriptor |= bit;
        bit <<= 1;
    }
    return descriptor;
}

int mo_mip_descriptor(const uint8_t *cur, const uint8_t *prev, int W, int H, float size, int x, int y,
                      uint8_t out[8])
{
    /* MoFREAKUtilities.cpp:288-325 */
    static const int CENTERS[8][2] = {{5, 5}, {5, 9}, {5, 13}, {9, 5}, {9, 13}, {13, 5}, {13, 9}, {13, 13}};
    const int tl_x = x - (int)size / 2;
    const int tl_y = y - (int)size / 2;
    const int L = (int)ceil(size);
    if (tl_x < 0 || tl_y < 0 || L <= 0 || tl_x + L > W || tl_y + L > H) return -1;
    uint8_t frame_t[19 * 19], frame_t_minus_1[19 * 19];
    mo_resize_linear_8u(cur + (size_t)tl_y * W + tl_x, W, L, L, frame_t, 19, 19);
    mo_resize_linear_8u(prev + (size_t)tl_y * W + tl_x, W, L, L, frame_t_minus_1, 19, 19);
    for (int c = 0; c < 8; ++c) out[c] = (uint8_t)mo_mip(frame_t, frame_t_minus_1, CENTERS[c][0], CENTERS[c][1]);
    return 0;
}

/* ---------------------------------------------------------------- the composed path */
void mo_extract_pair(const mo_freak *f, const uint8_t *cur, const uint8_t *prev, int W, int H,
                     const float *kps, int n, uint8_t *desc16, uint8_t *valid)
{
    /* MoFREAKUtilities.cpp:413-483 for one frame */
    uint8_t *diff = (uint8_t *)malloc((size_t)W * H);
    uint8_t *d64 = (uint8_t *)malloc((size_t)n * 64 + 1);
    mo_absdiff(cur, prev, diff, W, H);
    mo_freak_compute(f, diff, W, H, kps, n, valid, d64, NULL, NULL);
    memset(desc16, 0, (size_t)n * 16);
    for (int k = 0; k < n; ++k) {
        if (!valid[k]) continue;
        memcpy(desc16 + (size_t)k * 16, d64 + (size_t)k * 64, 8);
        /* :460 passes keypt->pt.x / pt.y (float) to int parameters: truncation */
        if (mo_mip_descriptor(cur, prev, W, H, kps[3 * k + 2], (int)kps[3 * k], (int)kps[3 * k + 1],
                              desc16 + (size_t)k * 16 + 8) != 0) {
            valid[k] = 0; /* the reference would have thrown out of cv::Mat::operator()(Rect) */
            memset(desc16 + (size_t)k * 16, 0, 16);
        }
    }
    free(diff);
    free(d64);
}

long mo_extract_stream(const mo_freak *f, const uint8_t *frames, int T, int W, int H, int gap,
                       const float *kps, const long *kp_offsets, mo_row *rows, long max_rows)
{
    /* MoFREAKUtilities.cpp:391-489: prev is the frame `gap` earlier; frame_num starts at gap-1 for
     * the first processed frame (index gap) and is incremented after each frame. */
    long n_rows = 0;
    const size_t fsz = (size_t)W * H;
    unsigned frame_num = (unsigned)gap - 1;
    for (int t = gap; t < T; ++t, ++frame_num) {
        const long p = t - gap;
        const long n = kp_offsets[p + 1] - kp_offsets[p];
        const float *kp = kps + 3 * kp_offsets[p];
        uint8_t *d16 = (uint8_t *)malloc((size_t)n * 16 + 1), *valid = (uint8_t *)malloc((size_t)n + 1);
        mo_extract_pair(f, frames + (size_t)t * fsz, frames + (size_t)p * fsz, W, H, kp, (int)n, d16, valid);
        for (long k = 0; k < n; ++k) {
            if (!valid[k]) continue;
            if (n_rows < max_rows) {
                mo_row *r = &rows[n_rows];
                r->x = kp[3 * k];
                r->y = kp[3 * k + 1];
                r->frame_number = (int32_t)frame_num;
                r->scale = kp[3 * k + 2];
                memcpy(r->appearance, d16 + 16 * k, 8);
                memcpy(r->motion, d16 + 16 * k + 8, 8);
            }
            ++n_rows;
        }
        free(d16);
        free(valid);
    }
    return n_rows;
}

/* ---------------------------------------------------------------- bag-of-words assignment */
static unsigned bow_hamming_byte(unsigned char a, unsigned char b)
{
    /* BagOfWordsRepresentation.cpp:54-72, bit loop kept as written */
    unsigned int hamming_distance = 0;
    unsigned int bit = 1;
    unsigned int xor_result = a ^ b;
    for (bit = 1; bit != 0; bit <<= 1) {
        if ((xor_result & bit) != 0) hamming_distance++;
    }
    return hamming_distance;
}

int mo_bow_match(const uint8_t *feature, const uint8_t *codebook, int n_codewords, int dim)
{
    /* bruteForceMatch (BagOfWordsRepresentation.cpp:22-37): strict <, so the FIRST minimum wins */
    int shortest_distance = 0x7fffffff;
    int shortest_index = -1;
    for (int i = 0; i < n_codewords; i++) {
        unsigned int dist = 0;
        for (int col = 0; col < dim; ++col) dist += bow_hamming_byte(feature[col], codebook[(size_t)i * dim + col]);
        if ((long long)dist < (long long)shortest_distance) {
            shortest_distance = (int)dist;
            shortest_index = i;
        }
    }
    return shortest_index;
}

int mo_bow_histogram(const uint8_t *desc, long n, const uint8_t *codebook, int n_codewords, int dim, float *hist)
{
    /* buildHistogram (BagOfWordsRepresentation.cpp:74-138) after the text parse: +1 per feature in float,
     * then every bin divided by the float sum; returns `success` (0 when there was no feature: bins stay 0) */
    int success = 0;
    for (int c = 0; c < n_codewords; ++c) hist[c] = 0;
    for (long k = 0; k < n; ++k) {
        const int best_match = mo_bow_match(desc + (size_t)k * dim, codebook, n_codewords, dim);
        hist[best_match] = hist[best_match] + 1;
        success = 1;
    }
    if (!success) return 0;
    float histogram_sum = 0;
    for (int c = 0; c < n_codewords; ++c) histogram_sum += hist[c];
    for (int c = 0; c < n_codewords; ++c) hist[c] = hist[c] / histogram_sum;
    return 1;
}

int mo_format_row(const mo_row *r, char *buf, size_t cap)
{
    /* MoFREAKUtilities.cpp:698-715; ostream<<float with default flags == printf("%g"); motion_x/y are 0 (:476-477) */
    int n = snprintf(buf, cap, "%g %g %d %g %g %g ", (double)r->x, (double)r->y, r->frame_number,
                     (double)r->scale, 0.0, 0.0);
    for (int i = 0; i < 8; ++i) n += snprintf(buf + n, cap > (size_t)n ? cap - n : 0, "%u ", (unsigned)r->appearance[i]);
    for (int i = 0; i < 8; ++i) n += snprintf(buf + n, cap > (size_t)n ? cap - n : 0, "%d ", (int)r->motion[i]);
    n += snprintf(buf + n, cap > (size_t)n ? cap - n : 0, "\n");
    return n;
}
