/*
 * brisk_oracle.h -- CPU restatement of the keypoint detector in front of the MoFREAK descriptor path
 * (SURVEY.md 8(f) row 1).  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it; the product never does.
 *
 * What it restates (reference = ChrisWhiten/MoFREAK, vendored BRISK + AGAST):
 *   MoFREAKUtilities.cpp:420-423        BriskFeatureDetector(30).detect(diff_img)
 *   brisk.cpp:549-559, 561-588          detectImpl, BriskScaleSpace ctor, constructPyramid
 *   brisk.cpp:590-704                   getKeypoints
 *   brisk.cpp:838-934                   isMax2D (reads the lazily filled score cache -- order dependent, kept)
 *   brisk.cpp:937-1103                  refine3D
 *   brisk.cpp:1106-1416                 getScoreMaxAbove / getScoreMaxBelow
 *   brisk.cpp:1418-1533, 1535-1644      refine1D, refine1D_1, refine1D_2, subpixel2D
 *   brisk.cpp:1646-1722                 BriskLayer ctor / getAgastPoints / getAgastScore (int and float) / _5_8
 *   brisk.cpp:1840-1972, 1974-2065      halfsample (SSE2), twothirdsample (SSSE3) incl. their scalar tails
 *   oast9_16.cc:46, oast9_16_nms.cc:42  OAST 9/16 detect + bisection corner score
 *   agast5_8_nms.cc:42                  AGAST 5/8 bisection corner score
 *
 * PARITY UNPINNED.  The reference has no tests or golden vectors for the detector.  brisk.cpp needs OpenCV 2.4.2
 * (cv::Mat, cv::KeyPoint, cv::Ptr) and the AGAST sources include opencv2/opencv.hpp through cvWrapper.h:34-37, so
 * neither builds in this image without stand-in headers; no oracle/_ref exists for them.  The machine-generated
 * OAST/AGAST decision trees (about 7000 lines) are restated by the predicate they evaluate -- the segment test
 * of the AGAST paper: N contiguous pixels of the Bresenham circle all brighter than centre+b or all darker than
 * centre-b, N = 9 of 16 (radius 3) and 5 of 8 (radius 1) -- with the reference's own bisection around it.
 * Floating point follows C's usual arithmetic conversions exactly as the reference's expressions are written
 * (float vs double literals), evaluated in IEEE single/double without contraction.
 */
#ifndef BRISK_ORACLE_H
#define BRISK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float x, y;      /* cv::KeyPoint::pt */
    float size;      /* basicSize_ (12) * refined scale */
    float response;  /* refined score */
    int32_t layer;   /* cv::KeyPoint::octave as the reference fills it: the pyramid layer index */
} mo_brisk_keypoint;

typedef struct mo_brisk mo_brisk;

/* BriskScaleSpace(octaves) + constructPyramid(image): layers = octaves ? 2*octaves : 1 */
mo_brisk *mo_brisk_create(const uint8_t *img, int stride, int w, int h, int octaves);
void mo_brisk_destroy(mo_brisk *b);
int mo_brisk_layers(const mo_brisk *b);
void mo_brisk_layer_info(const mo_brisk *b, int layer, int *w, int *h, float *scale, float *offset);
const uint8_t *mo_brisk_layer_image(const mo_brisk *b, int layer);
/* the score cache as it stands (0 where nothing has been computed yet) */
const uint8_t *mo_brisk_layer_scores(const mo_brisk *b, int layer);
/* getKeypoints(threshold): returns the number of keypoints the reference would emit, writes min(n, cap) */
int mo_brisk_get_keypoints(mo_brisk *b, int threshold, mo_brisk_keypoint *out, int cap);
/* the per-layer OAST points of the last mo_brisk_get_keypoints call (x, y interleaved); returns their number */
int mo_brisk_layer_points(const mo_brisk *b, int layer, int32_t *xy, int cap);

/* one-shot: BriskFeatureDetector(threshold, octaves).detect(img) */
int mo_brisk_detect(const uint8_t *img, int stride, int w, int h, int threshold, int octaves, mo_brisk_keypoint *out, int cap);

/* What a float expression of brisk.cpp means (see brisk_oracle.c): MO_BRISK_FP_X87, the default, is the reference as its
 * README and project file say it was built (Visual Studio 2010, Win32, no /arch: x87 code, intermediates at the FPU's 53 bits,
 * rounded to float at assignments); MO_BRISK_FP_SSE rounds every float operation.  Process-wide; set it before detecting. */
#define MO_BRISK_FP_X87 0
#define MO_BRISK_FP_SSE 1
void mo_brisk_set_fp_model(int model);
int mo_brisk_get_fp_model(void);

/* pieces, for component tests */
void mo_brisk_halfsample(const uint8_t *src, int w, int h, uint8_t *dst);       /* dst: (w/2) x (h/2) */
void mo_brisk_twothirdsample(const uint8_t *src, int w, int h, uint8_t *dst);   /* dst: 2*(w/3) x 2*(h/3) */
int mo_oast9_16_is_corner(const uint8_t *p, int stride, int b);
int mo_oast9_16_score(const uint8_t *p, int stride, int bmin);                   /* cornerScore with b = bmin */
int mo_agast5_8_score(const uint8_t *p, int stride, int bmin);
int mo_oast9_16_detect(const uint8_t *img, int w, int h, int b, int32_t *xy, int cap);
float mo_brisk_subpixel2d(const int s[9], float *dx, float *dy);                 /* s = s_0_0,s_0_1,s_0_2,s_1_0,...,s_2_2 */
float mo_brisk_refine1d(int variant, float s_05, float s0, float s05, float *max);  /* variant 0: refine1D, 1: _1, 2: _2 */

#ifdef __cplusplus
}
#endif
#endif
