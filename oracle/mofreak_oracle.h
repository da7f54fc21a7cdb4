/*
 * mofreak_oracle.h -- CPU restatement of the MoFREAK descriptor path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the HIP implementation in mofreak_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product never does.
 *
 * PARITY UNPINNED: the reference (ChrisWhiten/MoFREAK) ships no tests, fixtures or golden vectors,
 * and the part of the path that lives in OpenCV 2.4.2 (cv::FREAK, cv::resize, cv::absdiff,
 * cv::integral; pinned by reference README.md:13) is not in /root/reference and cannot be built here.
 * Those parts are restated from OpenCV 2.4.x's published algorithm (features2d/src/freak.cpp,
 * imgproc/src/imgwarp.cpp); the in-tree parts follow MoFREAKUtilities.cpp line by line.
 * The oracle is pinned only by the hand-derivable known-answer tests in tests/test_oracle_kat.py.
 */
#ifndef MOFREAK_ORACLE_H
#define MOFREAK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FREAK constants (OpenCV 2.4.x freak.cpp). */
#define MO_NB_SCALES 64
#define MO_NB_ORIENTATION 256
#define MO_NB_POINTS 43
#define MO_NB_PAIRS 512
#define MO_NB_ORIENPAIRS 45
#define MO_SMALLEST_KP_SIZE 7

/* How the 512 pair bits land in the 64 descriptor bytes. */
enum {
    MO_BITS_SSE = 0,        /* v[i] >= v[j] unsigned, SSE byte order (every OpenCV >= 2.4.3, and 2.4.2 with CV_SSE2) */
    MO_BITS_NATURAL = 1,    /* v[i] >  v[j], std::bitset order (2.4.2 built without CV_SSE2) */
    MO_BITS_SSE_SIGNED = 2  /* (int8)v[i] > (int8)v[j], SSE byte order (early SSE path with _mm_cmpgt_epi8) */
};

typedef struct { float x, y, sigma; } mo_pattern_point;
typedef struct { uint8_t i, j; int weight_dx, weight_dy; } mo_orient_pair;
typedef struct { uint8_t i, j; } mo_desc_pair;

typedef struct mo_freak {
    float pattern_scale;
    int n_octaves;
    int orientation_normalized;
    int scale_normalized;
    int bit_mode;
    mo_pattern_point *lut;               /* [64][256][43] */
    int pattern_sizes[MO_NB_SCALES];
    mo_orient_pair orient[MO_NB_ORIENPAIRS];
    mo_desc_pair pairs[MO_NB_PAIRS];
} mo_freak;

/* 32-byte binary row: what one line of a .mofreak file carries (MoFREAKUtilities.h:23-53). */
typedef struct {
    float x, y;
    int32_t frame_number;
    float scale;
    uint8_t appearance[8];
    uint8_t motion[8];
} mo_row;

mo_freak *mo_freak_create(float pattern_scale, int n_octaves, int orientation_normalized,
                          int scale_normalized, int bit_mode);
void mo_freak_destroy(mo_freak *f);
/* table read-back for tests: 512 (i,j) byte pairs; 45 x (i, j, weight_dx, weight_dy); 43 x (x, y, sigma) */
void mo_freak_get_pairs(const mo_freak *f, uint8_t *out_ij);
void mo_freak_get_orientation(const mo_freak *f, int *out);
void mo_freak_get_pattern(const mo_freak *f, int scale, int rot, float *out);

/* cv::cvtColor(BGR2GRAY) on 8UC3 (MoFREAKUtilities.cpp:395, :410). */
void mo_bgr2gray(const uint8_t *bgr, int W, int H, uint8_t *gray);
/* cv::absdiff on 8U (MoFREAKUtilities.cpp:414). */
void mo_absdiff(const uint8_t *a, const uint8_t *b, uint8_t *d, int W, int H);
/* cv::integral 8U -> 32S, (H+1)x(W+1), first row/col zero. */
void mo_integral(const uint8_t *img, int W, int H, int32_t *integ);

int mo_freak_scale_index(const mo_freak *f, float size);
/* thetaIdx from the two integer direction sums (freak.cpp computeImpl). */
int mo_freak_theta_index(int direction0, int direction1);
/* Same with glibc atanf2 instead of double atan2 rounded to float; for the disagreement-rate test. */
int mo_freak_theta_index_atan2f(int direction0, int direction1);
uint8_t mo_freak_mean_intensity(const mo_freak *f, const uint8_t *img, const int32_t *integ, int W, int H,
                                float kp_x, float kp_y, unsigned scale, unsigned rot, unsigned point);
/*
 * cv::FREAK::compute on n keypoints (x,y,size triples).  valid[k]=0 where DescriptorExtractor::compute /
 * FREAK::computeImpl would have erased keypoint k.  desc64 is n x 64 (rows of erased keypoints zero),
 * theta_out (optional) receives thetaIdx, dir_out (optional) receives direction0, direction1.
 */
void mo_freak_compute(const mo_freak *f, const uint8_t *img, int W, int H, const float *kps, int n,
                      uint8_t *valid, uint8_t *desc64, int *theta_out, int *dir_out);

/* cv::resize(src 8UC1 -> dw x dh, INTER_LINEAR) (MoFREAKUtilities.cpp:303-304). */
void mo_resize_linear_8u(const uint8_t *src, int sstride, int sw, int sh, uint8_t *dst, int dw, int dh);
/* Per-axis tables of that resize: ofs[d], coef[2d], coef[2d+1]; returns xmax (first d served by pure copy). */
int mo_resize_axis_table(int ssize, int dsize, int is_x, int *ofs, short *coef);

/* MoFREAKUtilities::motionInterchangePattern (MoFREAKUtilities.cpp:46-99) on 19x19 buffers. */
unsigned mo_mip(const uint8_t *cur19, const uint8_t *prev19, int x, int y);
/* MoFREAKUtilities::extractMotionByMotionInterchangePatterns (:288-325).  Returns 0, or -1 if the
 * ROI leaves the image (the reference would throw there). */
int mo_mip_descriptor(const uint8_t *cur, const uint8_t *prev, int W, int H, float size, int x, int y,
                      uint8_t out[8]);

/* One frame pair: FREAK on |cur-prev| + MIP on (cur, prev) -> n x 16 bytes (appearance, motion). */
void mo_extract_pair(const mo_freak *f, const uint8_t *cur, const uint8_t *prev, int W, int H,
                     const float *kps, int n, uint8_t *desc16, uint8_t *valid);

/*
 * computeMoFREAKFromFile's frame loop (:374-498) on a T x H x W gray stack with one keypoint list per
 * processed frame (kp_offsets has T-gap+1 entries).  Writes at most max_rows rows; returns the row count.
 */
long mo_extract_stream(const mo_freak *f, const uint8_t *frames, int T, int W, int H, int gap,
                       const float *kps, const long *kp_offsets, mo_row *rows, long max_rows);

/* BagOfWordsRepresentation::bruteForceMatch (BagOfWordsRepresentation.cpp:22-72): index of the nearest codeword
 * by bitwise Hamming distance over `dim` bytes, first minimum on ties. */
int mo_bow_match(const uint8_t *feature, const uint8_t *codebook, int n_codewords, int dim);
/* BagOfWordsRepresentation::buildHistogram (:74-138) on n in-memory descriptors; returns success (0/1). */
int mo_bow_histogram(const uint8_t *desc, long n, const uint8_t *codebook, int n_codewords, int dim, float *hist);

/* writeMoFREAKFeaturesToFile (:691-719): one text row; returns bytes written (excl. NUL). */
int mo_format_row(const mo_row *r, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
