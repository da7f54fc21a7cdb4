/*
 * mofreak_hip.h -- C ABI of libmofreak_hip.so, the MI355X (gfx950) implementation of the MoFREAK
 * descriptor-extraction path of ChrisWhiten/MoFREAK.
 *
 * The reference exposes this path only as methods of the C++ class MoFREAKUtilities (there is no C ABI
 * or plugin interface upstream).  Each entry point below names the reference code it replaces
 * (paths relative to the reference's src/MoFREAK/).  INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - every function returns an int status (MOFREAK_OK or a negative MOFREAK_ERR_*); nothing throws or
 *    exits across this boundary; mofreak_last_error() returns the message of the last failure;
 *  - one context per host thread / HIP stream; contexts are independent;
 *  - buffers are caller-owned.  `flags` says whether the pointers of a call are device pointers
 *    (MOFREAK_MEM_DEVICE, the fast path: nothing is copied) or host pointers (MOFREAK_MEM_HOST: the
 *    library stages them through its own device buffers and synchronises before returning);
 *  - device-pointer calls are asynchronous on the context's stream unless stated otherwise.
 */
#ifndef MOFREAK_HIP_H
#define MOFREAK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOFREAK_ABI_VERSION 3

/* status codes */
#define MOFREAK_OK 0
#define MOFREAK_ERR_BAD_ARG (-1)
#define MOFREAK_ERR_HIP (-2)
#define MOFREAK_ERR_OOM (-3)
#define MOFREAK_ERR_UNSUPPORTED (-4)
#define MOFREAK_ERR_NO_DEVICE (-5)
#define MOFREAK_ERR_ROI (-6)       /* a keypoint's MIP ROI left the image (reference: cv::Exception) */
#define MOFREAK_ERR_CAPACITY (-7)  /* output buffer too small */

/* device_id for mofreak_create that builds the host tables only (no GPU needed): the table accessors
 * below work on such a context, every compute entry point returns MOFREAK_ERR_NO_DEVICE. */
#define MOFREAK_TABLES_ONLY (-1)

/* flags */
#define MOFREAK_MEM_DEVICE 0u
#define MOFREAK_MEM_HOST 1u
#define MOFREAK_ROWS_DEVICE 2u /* mofreak_extract_clips: rows_out is a device pointer (frames and keypoints stay host pointers) */

/* FREAK pair-bit layout; OpenCV 2.4.2's binary is not available to pin it */
#define MOFREAK_BITS_SSE 0        /* v[i] >= v[j], SSE byte order: OpenCV >= 2.4.3, and 2.4.2 built with CV_SSE2 */
#define MOFREAK_BITS_NATURAL 1    /* v[i] >  v[j], std::bitset order: 2.4.2 built without CV_SSE2 */
#define MOFREAK_BITS_SSE_SIGNED 2 /* (int8)v[i] > (int8)v[j], SSE byte order */

/* What a C `float` expression of the vendored BRISK detector (brisk.cpp, compiled inside the reference's own project) means.
 * The reference is a 32-bit Visual Studio 2010 project (README.md:13; MoFREAK.vcxproj: Win32 configurations, no /arch option):
 * that compiler emits x87 code, the CRT runs the FPU at 53-bit precision, and under the default /fp:precise the intermediates
 * of an expression stay in FPU registers -- values are rounded to float only where they are assigned, cast, passed or
 * returned.  MOFREAK_FP_X87 restates that and is the default; MOFREAK_FP_SSE rounds every float operation to float (what
 * /arch:SSE2, a later Visual Studio or a 64-bit build would do).  On tie-heavy and moving-object test images the two
 * readings give the same NUMBER of keypoints, differ in the last bit of x, y or response for about 30 % of them and by
 * more than that (up to 2 px, or a different size) for 0.07 %.  OpenCV's own code (cv::FREAK,
 * cv::resize) lives in the prebuilt OpenCV 2.4.2 libraries, which are SSE2 builds: nothing else depends on this choice. */
#define MOFREAK_FP_X87 0
#define MOFREAK_FP_SSE 1

#define MOFREAK_APPEARANCE_BYTES 8 /* MoFREAKUtilities.h:21 */
#define MOFREAK_MOTION_BYTES 8     /* MoFREAKUtilities.h:20 */
#define MOFREAK_DESC_BYTES 16

typedef struct mofreak_ctx mofreak_ctx;

/* Algorithm constants; mofreak_default_params() fills in the reference's values. */
typedef struct mofreak_params {
    int32_t struct_size;                  /* sizeof(mofreak_params), set by mofreak_default_params */
    int32_t gap_for_frame_difference;     /* 5      MoFREAKUtilities.cpp:378 */
    int32_t mip_theta;                    /* 288    MoFREAKUtilities.cpp:48 */
    float freak_pattern_scale;            /* 22.0f  cv::FREAK default ctor, MoFREAKUtilities.cpp:427 */
    int32_t freak_n_octaves;              /* 4 */
    int32_t freak_orientation_normalized; /* 1 */
    int32_t freak_scale_normalized;       /* 1 */
    int32_t freak_bit_mode;               /* MOFREAK_BITS_SSE */
    int32_t brisk_fp_model;               /* MOFREAK_FP_X87: keypoint detector only (mofreak_detect_pairs, mofreak_compute_stream, streams) */
} mofreak_params;

/* cv::KeyPoint fields the path reads (pt.x, pt.y, size). */
typedef struct mofreak_keypoint {
    float x, y, size;
} mofreak_keypoint;

/* One .mofreak row in binary: the fields of struct MoFREAKFeature (MoFREAKUtilities.h:23-53) that
 * writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719) prints; motion_x/motion_y are always 0. */
typedef struct mofreak_row {
    float x, y;
    int32_t frame_number;
    float scale;
    uint8_t appearance[MOFREAK_APPEARANCE_BYTES];
    uint8_t motion[MOFREAK_MOTION_BYTES];
} mofreak_row;

/* ------------------------------------------------------------------ context */
int mofreak_abi_version(void);
/* Bit 0: this is the bounds-checking debug build (every tile-kernel LDS access and descriptor store checked; a violation
 * comes back from mofreak_check_status as MOFREAK_ERR_HIP).  The product build returns 0. */
int mofreak_build_flags(void);
int mofreak_default_params(mofreak_params *p);

/* Replaces MoFREAKUtilities::MoFREAKUtilities (MoFREAKUtilities.cpp:5-8) plus the per-frame
 * `cv::FREAK extractor;` construction (:427): builds the FREAK pattern LUT, orientation weights, pair
 * table and the 19x19 resize coefficient tables ON THE HOST once and uploads them to `device_id`. */
int mofreak_create(int device_id, const mofreak_params *params, mofreak_ctx **out);
void mofreak_destroy(mofreak_ctx *ctx);
/* Message of the last failure on ctx (ctx == NULL: of the last failed mofreak_create on this thread). */
const char *mofreak_last_error(const mofreak_ctx *ctx);
/* Run the context's work on an existing hipStream_t (e.g. torch's current stream); NULL = own stream. */
int mofreak_set_stream(mofreak_ctx *ctx, void *hip_stream);
int mofreak_synchronize(mofreak_ctx *ctx);
/* Pre-size the internal workspace (integral images of one chunk of pairs) so that later calls do not
 * allocate.  chunk_pairs <= 0 keeps the default.  Optional. */
int mofreak_reserve(mofreak_ctx *ctx, int W, int H, int chunk_pairs);
/* Synchronise and report per-keypoint anomalies seen by the kernels since the last call:
 * MOFREAK_ERR_ROI / MOFREAK_ERR_UNSUPPORTED, else MOFREAK_OK.  Affected keypoints have out_valid = 0. */
int mofreak_check_status(mofreak_ctx *ctx);

/* Which kernels describe the keypoints.  AUTO: the fused tile kernel for every keypoint whose FREAK pattern fits a
 * tile's 40-px halo (size < ~12.56), the gather path (global integral + one wavefront per keypoint) for the rest.
 * GATHER: the gather path for everything (the round-1 v1 kernels; kept for large keypoints and for A/B tests). */
#define MOFREAK_PATH_AUTO 0
#define MOFREAK_PATH_GATHER 1
int mofreak_set_path(mofreak_ctx *ctx, int path);
/* Per-call device timing, measured with HIP events on the context's stream around the binning kernels, the tile
 * kernel and the gather path of every extract call.  Off by default. */
typedef struct mofreak_profile {
    double bin_ms;       /* keypoint binning (bin_count / bin_scan / bin_scatter) */
    double tile_ms;      /* tile_kernel: the fused kernel that describes everything that fits a tile's halo */
    double gather_ms;    /* gather path: integral kernels + describe_kernel (large keypoints; ~0 when there are none) */
    int64_t calls;       /* extract calls timed; each launches tile_kernel once (per 32768 pairs) */
    int64_t pairs;       /* frame pairs those calls covered */
    int64_t descriptors; /* keypoint instances those calls covered */
} mofreak_profile;
int mofreak_set_profiling(mofreak_ctx *ctx, int enable);
/* Synchronises, folds the recorded events into the running totals and returns them (reset != 0 clears them). */
int mofreak_get_profile(mofreak_ctx *ctx, mofreak_profile *out, int reset);

/* Diagnostics: per-phase s_memtime tick sums of the instrumented tile kernel (one thread per workgroup, summed over
 * workgroups), available only on a context created with MOFREAK_TILE_STAMPS=1 in the environment; such a context
 * runs the instrumented kernel, whose run time must not be quoted.  Slots: see tile_kernel.hip TILE_STAMP. */
int mofreak_get_tile_stamps(mofreak_ctx *ctx, uint64_t *out, int n, int reset);

/* ------------------------------------------------------------------ the hot path */
/*
 * Descriptors for n_pairs frame pairs.  Replaces, per frame of computeMoFREAKFromFile's loop:
 * cv::absdiff (MoFREAKUtilities.cpp:413-414), cv::FREAK::compute on the difference image (:427-428)
 * with the copy of descriptor bytes 0..7 (:453-456), and extractMotionByMotionInterchangePatterns
 * (:460 -> :288-325 -> motionInterchangePattern :46-99) on (current, previous).
 *
 *   cur, prev    gray u8 frames; pair p at cur + p*pair_stride, prev + p*pair_stride; rows row_stride apart
 *                (a T-frame stack is the n_pairs = T-gap case with cur = frames + gap*H*W, prev = frames)
 *                limits (MOFREAK_ERR_UNSUPPORTED beyond them): W <= 10232, row_stride < 2^23, H*row_stride < 2^31
 *   kps          keypoints, in the order the detector produced them
 *   kp_offsets   NULL: the same n_kp keypoints are described in every pair (dense grid);
 *                else n_pairs+1 int64 CSR offsets into kps (same memory space as kps), n_kp = total
 *   out_desc16   n_out x 16 bytes (appearance[8], motion[8]); n_out = n_pairs*n_kp (shared list) or n_kp (CSR)
 *   out_valid    n_out bytes: 0 where DescriptorExtractor::compute / FREAK::computeImpl would have ERASED the
 *                keypoint (size < FLT_EPSILON, or within patternSizes[scale] of the border); its
 *                descriptor bytes are zero.  No compaction here: see mofreak_compact_rows.
 */
int mofreak_extract_pairs(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H,
                          int64_t row_stride, int64_t pair_stride, int n_pairs,
                          const mofreak_keypoint *kps, const int64_t *kp_offsets, int64_t n_kp,
                          uint8_t *out_desc16, uint8_t *out_valid, unsigned flags);

/*
 * Stable (order-preserving) compaction of the valid keypoints into 32-byte rows: what the keypoint loop
 * (MoFREAKUtilities.cpp:436-483) pushes into `features`.  Pair p gets frame_number first_frame_number + p
 * (:401, :448, :488).  rows_out needs room for rows_capacity rows; *n_rows_out (host) receives the count.
 * Synchronises the stream.
 */
int mofreak_compact_rows(mofreak_ctx *ctx, const mofreak_keypoint *kps, const int64_t *kp_offsets,
                         int64_t n_kp, int n_pairs, int first_frame_number, const uint8_t *desc16,
                         const uint8_t *valid, mofreak_row *rows_out, int64_t rows_capacity,
                         int64_t *n_rows_out, unsigned flags);

/*
 * A whole gray frame stack: the frame loop of MoFREAKUtilities::computeMoFREAKFromFile
 * (MoFREAKUtilities.cpp:391-489) after decoding -- prev is the frame `gap` earlier, the first processed
 * frame (index gap) is labelled gap-1.  kp_offsets: NULL (same n_kp keypoints for each of the T-gap
 * processed frames) or T-gap+1 CSR offsets.  T <= gap yields zero rows.  Synchronises the stream.
 */
int mofreak_extract_stream(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H,
                           const mofreak_keypoint *kps, const int64_t *kp_offsets, int64_t n_kp,
                           mofreak_row *rows_out, int64_t rows_capacity, int64_t *n_rows_out,
                           unsigned flags);

/*
 * The same frame loop for a long host-resident stream (BASELINE config 5: hour-long 720x576 streams), pipelined:
 * the stack is walked in chunks of chunk_frames frames (consecutive chunks overlap by `gap` frames, the depth of the
 * reference's frame queue, MoFREAKUtilities.cpp:391-399, 485-487), and chunk k+1's host-to-device copy runs on its
 * own HIP stream under chunk k's kernels while chunk k-1's rows travel back on a third.  frames, kps (one shared
 * list of n_kp keypoints for every processed frame) and rows_out are HOST pointers; frames / rows_out in memory
 * from mofreak_host_alloc (page-locked) are copied by DMA straight from / to the caller's buffer, any other host
 * memory goes through the library's own page-locked staging buffers (one more host copy).  Rows are those of
 * mofreak_extract_stream on the whole stack, byte for byte.  chunk_frames <= gap selects a default (256).  Returns only
 * after every copy from / to the caller's buffers has finished (on errors too); on MOFREAK_ERR_CAPACITY *n_rows_out holds
 * the number of rows a retry needs.
 */
int mofreak_extract_stream_pipelined(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H, int chunk_frames,
                                     const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out,
                                     int64_t rows_capacity, int64_t *n_rows_out);
/*
 * Many videos in one call: the body of the dataset loop of computeMoFREAKFiles (main.cpp:862-921), which hands one video
 * at a time to computeMoFREAKFromFile (MoFREAKUtilities.cpp:374-498), for n_clips decoded videos at once.  Clip c is
 * clip_n_frames[c] contiguous W x H gray frames at clip_frames[c] (HOST pointers; page-locked memory from
 * mofreak_host_alloc is copied by DMA straight from the caller's buffer, other memory through the library's page-locked
 * staging).  Every clip is treated exactly as mofreak_extract_stream treats a stack -- pairs never cross a clip boundary,
 * frame numbers restart at gap - 1 in every clip (:401, :488), clips of <= gap frames yield no rows -- but the clips
 * share the launches and the three-stream pipeline of mofreak_extract_stream_pipelined (copies of the next window of
 * frames under the kernels of this one, rows of the previous one on their way back), so that short clips cost bandwidth,
 * not one synchronous round trip each.  kps: one shared list of n_kp keypoints for every processed frame (host pointer).
 *
 *   rows_out               rows of clip 0, then clip 1, ... (inside a clip: frames ascending, keypoints in input order):
 *                          a HOST pointer, or with MOFREAK_ROWS_DEVICE in flags a DEVICE pointer (rows stay in HBM, e.g.
 *                          for the RCCL gather to the root rank)
 *   clip_row_offsets_out   optional, n_clips + 1 host integers: rows of clip c are [offsets[c], offsets[c + 1])
 *   n_rows_out             total rows; more than rows_capacity: MOFREAK_ERR_CAPACITY, nothing beyond the capacity is
 *                          written and *n_rows_out holds the size a retry needs
 *   chunk_frames <= gap    selects a default window (about 96 MiB of frames)
 * Returns only after every copy from / to the caller's buffers has finished, on errors too.
 */
int mofreak_extract_clips(mofreak_ctx *ctx, const uint8_t *const *clip_frames, const int32_t *clip_n_frames, int n_clips,
                          int W, int H, int chunk_frames, const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out,
                          int64_t rows_capacity, int64_t *clip_row_offsets_out, int64_t *n_rows_out, unsigned flags);
/* The same pipeline with the reference's own keypoint source: BriskFeatureDetector(threshold, octaves) on every pair's
 * difference image (MoFREAKUtilities.cpp:420-423) instead of a caller's list -- computeMoFREAKFromFile's whole frame loop
 * (:374-498) for MANY clips in one pass.  Windows are detector batches: detector, descriptors of the window's keypoints, rows
 * (what mofreak_compute_stream does for one resident stack), the next window's frames copied down meanwhile, the previous
 * window's rows on their way back.  Rows are those of one mofreak_compute_stream call per clip, clip after clip.  The number
 * of rows is not known up front: MOFREAK_ERR_CAPACITY with the size in *n_rows_out, as in mofreak_extract_clips.
 * n_keypoints_out (optional): keypoints detected, erased ones included.  flags: MOFREAK_ROWS_DEVICE as above. */
int mofreak_compute_clips(mofreak_ctx *ctx, const uint8_t *const *clip_frames, const int32_t *clip_n_frames, int n_clips, int W, int H,
                          int chunk_frames, int threshold, int octaves, mofreak_row *rows_out, int64_t rows_capacity,
                          int64_t *clip_row_offsets_out, int64_t *n_rows_out, int64_t *n_keypoints_out, unsigned flags);
/* Device memory for rows that stay in HBM (MOFREAK_ROWS_DEVICE; the RCCL gather of include/mofreak_dist.h) for callers that do
 * not link the HIP runtime themselves: hipMalloc / hipFree on the context's device, and a synchronous device-to-host copy. */
int mofreak_device_alloc(mofreak_ctx *ctx, size_t bytes, void **out);
int mofreak_device_free(mofreak_ctx *ctx, void *ptr);
int mofreak_copy_to_host(mofreak_ctx *ctx, void *host_dst, const void *device_src, size_t bytes);
/* Page-locked host memory for frames decoded by the caller and for rows (hipHostMalloc / hipHostFree). */
int mofreak_host_alloc(mofreak_ctx *ctx, size_t bytes, void **out);
int mofreak_host_free(mofreak_ctx *ctx, void *ptr); /* ctx may be NULL: the memory may outlive its context */

/* ------------------------------------------------------------------ frame preparation */
/* cv::cvtColor(frame, frame, CV_BGR2GRAY) on 8UC3 frames (MoFREAKUtilities.cpp:395, :410): interleaved B,G,R bytes,
 * rows row_stride bytes apart, frames frame_stride bytes apart -> n_frames contiguous W x H gray frames, ready for
 * mofreak_extract_stream.  OpenCV 2.4.x fixed point: (1868 B + 9617 G + 4899 R + 8192) >> 14. */
int mofreak_bgr_to_gray(mofreak_ctx *ctx, const uint8_t *bgr, int W, int H, int64_t row_stride, int64_t frame_stride,
                        int n_frames, uint8_t *gray_out, unsigned flags);

/* ------------------------------------------------------------------ keypoint detector */
/*
 * BriskFeatureDetector(threshold, octaves).detect(|cur - prev|) for n_pairs frame pairs: what
 * computeMoFREAKFromFile does at MoFREAKUtilities.cpp:413-423 before it describes the keypoints -- cv::absdiff, then
 * BriskScaleSpace::constructPyramid + getKeypoints (brisk.cpp:549-704) over the OAST 9/16 corner detector
 * (oast9_16.cc:46, oast9_16_nms.cc:42).  The reference uses threshold 30 and octaves 3 (MoFREAKUtilities.cpp:420,
 * brisk.h:228).  prev == NULL: search cur itself (any gray image).
 *
 *   out_kps       x, y, size of every keypoint, pair after pair, inside a pair in the order of the reference's
 *                 keypoint vector (layer by layer, raster order inside a layer) -- ready to be handed to
 *                 mofreak_extract_pairs together with out_offsets
 *   out_offsets   n_pairs + 1 CSR offsets into out_kps (same memory space as out_kps)
 *   out_response  optional: cv::KeyPoint::response (the refined score); out_layer: optional cv::KeyPoint::octave
 *                 as the reference fills it (the pyramid layer index)
 *   n_out         total number of keypoints (host pointer).  More than `capacity`: MOFREAK_ERR_CAPACITY, nothing
 *                 beyond capacity is written.  octaves: 0..4; W, H <= 65535.  Synchronises the stream.
 */
int mofreak_detect_pairs(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H, int64_t row_stride,
                         int64_t pair_stride, int n_pairs, int threshold, int octaves, mofreak_keypoint *out_kps,
                         int64_t capacity, int64_t *out_offsets, float *out_response, int32_t *out_layer, int64_t *n_out,
                         unsigned flags);
/*
 * The whole frame loop of MoFREAKUtilities::computeMoFREAKFromFile (MoFREAKUtilities.cpp:391-491) for a T-frame gray
 * stack already in memory: for every frame from index gap on, BRISK keypoints on |frame - frame[-gap]|
 * (mofreak_detect_pairs), their MoFREAK descriptors (mofreak_extract_pairs) and the rows FREAK did not erase, in the
 * order the reference appends them (mofreak_compact_rows).  Equivalent to those three calls; frames cross the host
 * boundary once.  Synchronises the stream.
 */
int mofreak_compute_stream(mofreak_ctx *ctx, const uint8_t *frames, int T, int W, int H, int threshold, int octaves,
                           mofreak_row *rows_out, int64_t rows_capacity, int64_t *n_rows_out, int64_t *n_keypoints_out,
                           unsigned flags);
/* ------------------------------------------------------------------ frame at a time */
/*
 * The reference's loop as it is written: `capture >> current_frame` one frame at a time against a queue of the last
 * gap frames (MoFREAKUtilities.cpp:391-401, 402-411, 485-488).  The ring of gap + 1 gray frames lives on the device;
 * a pushed frame is converted (channels == 3: interleaved BGR, cv::cvtColor(BGR2GRAY) :395,410) or copied
 * (channels == 1) into it, and from the (gap + 1)-th frame on its rows come back: keypoints from the BRISK detector
 * on |frame - frame[-gap]| (use_detector != 0, :420-423) or from the caller (kps, n_kp; ignored with the detector),
 * described against the frame gap pushes ago, labelled gap - 1, gap, ... (:401, :488).  For throughput use
 * mofreak_compute_stream / mofreak_extract_stream on whole stacks; this interface is for callers that decode as
 * they go.  frame / kps / rows_out are host or device pointers according to flags; *n_rows_out is a host integer.
 * A stream holds a pointer to its context: close it (mofreak_stream_close) before mofreak_destroy.
 */
typedef struct mofreak_stream mofreak_stream;
int mofreak_stream_open(mofreak_ctx *ctx, int W, int H, int use_detector, int threshold, int octaves,
                        mofreak_stream **out);
int mofreak_stream_push(mofreak_stream *s, const uint8_t *frame, int channels, int64_t row_stride,
                        const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out, int64_t rows_capacity,
                        int64_t *n_rows_out, unsigned flags);
/*
 * Many frames at once, bounded memory: the continuation of the same stream by n_frames contiguous W x H gray frames in HOST
 * memory (page-locked memory from mofreak_host_alloc goes down by DMA in place).  The gap frames before the chunk are
 * still in the device ring, so pairs run on across pushes and frame numbers keep counting (MoFREAKUtilities.cpp:391-401,
 * 485-488 is O(1) in the length of the video, and so is this: nothing but the ring survives a push).  Inside the call the
 * chunk goes through the three-stream pipeline of mofreak_extract_stream_pipelined in windows of chunk_frames frames
 * (<= gap: a default of about 96 MiB): the next window's copy under this window's kernels, the previous one's rows on
 * their way back.  kps: ONE keypoint list (host) for every frame of the chunk; a stream opened with use_detector = 1 finds
 * its keypoints on the device window by window (kps and n_kp are ignored), as mofreak_compute_clips does.
 * rows_out (host; page-locked: DMA in place) receives the chunk's rows -- those of mofreak_extract_stream on the whole
 * stream, piece by piece.  A caller with two chunk buffers refills one while the other is being pushed.  On
 * MOFREAK_ERR_CAPACITY the frames are consumed all the same and *n_rows_out holds the number of rows the chunk has.
 * Frame-at-a-time pushes and chunk pushes may be mixed on one stream.
 */
int mofreak_stream_push_frames(mofreak_stream *s, const uint8_t *frames, int n_frames, int chunk_frames,
                               const mofreak_keypoint *kps, int64_t n_kp, mofreak_row *rows_out, int64_t rows_capacity,
                               int64_t *n_rows_out);
int64_t mofreak_stream_frames(const mofreak_stream *s); /* frames pushed so far */
void mofreak_stream_close(mofreak_stream *s);

/* Candidates (corners of a pair's pyramid) per pair the detector reserves room for; more corners than that in one pair make
 * mofreak_detect_pairs return MOFREAK_ERR_CAPACITY.  By default (and again after candidates_per_pair = 0) the room follows the
 * frames: a call starts from an eighth of the frame's pixels and, when a pair has more, runs again with four times that (kept
 * for the context's later calls), up to 131072 -- small frames do not pay for a full-HD pair's lists.  A number names the room
 * outright (256 .. 2^24). */
int mofreak_detect_set_capacity(mofreak_ctx *ctx, int candidates_per_pair);

/* Components, for tests and for callers that want the pyramid: the layers of BriskScaleSpace::constructPyramid over
 * one gray image (BriskLayer::halfsample / twothirdsample, brisk.cpp:1840-2065) and, with scores_out, the OAST 9/16
 * corner score of every pixel of every layer (BriskLayer::getAgastScore(x, y, 1), :1685-1694; 0 inside the 3-pixel
 * border).  Layers are written one after the other, w*h bytes each, rows packed; dims_out gets w, h per layer
 * (2 * n_layers int32), scale_offset_out scale, offset per layer.  Any output may be NULL.  Returns the number of
 * layers through n_layers_out. */
int mofreak_brisk_pyramid(mofreak_ctx *ctx, const uint8_t *img, int W, int H, int64_t row_stride, int octaves,
                          uint8_t *layers_out, uint8_t *scores_out, int32_t *dims_out, float *scale_offset_out,
                          int *n_layers_out, unsigned flags);

/* ------------------------------------------------------------------ bag-of-words assignment */
/* BagOfWordsRepresentation::bruteForceMatch (BagOfWordsRepresentation.cpp:22-37, hammingDistance :39-72): for each of
 * n 16-byte descriptors the index of the nearest of n_codewords 16-byte codewords by bitwise Hamming distance, the
 * FIRST minimum on ties.  valid may be NULL; descriptors with valid[k] == 0 get index -1.  n_codewords <= 10240. */
int mofreak_bow_assign(mofreak_ctx *ctx, const uint8_t *desc16, const uint8_t *valid, int64_t n, const uint8_t *codebook16,
                       int n_codewords, int32_t *out_index, unsigned flags);
/* BagOfWordsRepresentation::buildHistogram (:74-138) on descriptors already in memory: one count per (valid)
 * descriptor at its codeword, every bin divided by the sum in float.  *success_out (host, optional) is the reference's
 * `success` flag (0: no descriptor, bins are 0).  Synchronises the stream. */
int mofreak_bow_histogram(mofreak_ctx *ctx, const uint8_t *desc16, const uint8_t *valid, int64_t n, const uint8_t *codebook16,
                          int n_codewords, float *hist_out, int32_t *success_out, unsigned flags);

/* ------------------------------------------------------------------ .mofreak text */
/* MoFREAKUtilities::writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719), byte for byte.  Writes at
 * most cap bytes to buf (may be NULL) and always stores the full length in *needed. */
int mofreak_format_rows(const mofreak_row *rows, int64_t n_rows, char *buf, size_t cap, size_t *needed);
/* The same text made ON THE DEVICE from rows in device memory (the rows mofreak_extract_* leave there with MOFREAK_ROWS_DEVICE,
 * or the root's gathered rows): writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719) byte for byte, `ostream << float`
 * (printf "%g") included.  `text` (cap bytes; device memory, or page-locked host memory from mofreak_host_alloc, which the
 * device writes directly; may be NULL to size) receives it, *needed its length.  One call serves many files: row_starts
 * (host, n_segments ascending row indices; may be NULL with n_segments 0) are the first rows of the videos, and
 * segment_offsets_out[i] (host, n_segments + 1 entries) the byte offset of row_starts[i]'s text, [n_segments] the total.
 * A row with a float %g would print with an exponent (outside [1e-4, 1e6)), negative or not finite is not formatted here:
 * MOFREAK_ERR_UNSUPPORTED, and the caller uses mofreak_format_rows (keypoint coordinates and sizes are never such values).
 * MOFREAK_ERR_CAPACITY (with *needed set) when cap is too small.  Synchronises the context's stream. */
int mofreak_format_rows_device(mofreak_ctx *ctx, const mofreak_row *d_rows, int64_t n_rows, char *text, size_t cap, size_t *needed,
                               const int64_t *row_starts, int n_segments, size_t *segment_offsets_out);
/* MoFREAKUtilities::readMoFREAKFeatures' row parser (MoFREAKUtilities.cpp:1146-1190), in FILE order (the
 * reference then stores them reversed, :1206-1210).  rows may be NULL to count. */
int mofreak_parse_rows(const char *text, size_t len, mofreak_row *rows, int64_t rows_capacity,
                       int64_t *n_rows_out);

/* ------------------------------------------------------------------ component entry points (parity tests) */
/* cv::absdiff + cv::integral of the difference image: out is n_pairs x (H+1) x (W+1) int32. */
int mofreak_diff_integral(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H,
                          int64_t row_stride, int64_t pair_stride, int n_pairs, int32_t *out,
                          unsigned flags);
/* motionInterchangePattern at the 8 patch centres on n given 19x19 buffer pairs (361 bytes each). */
int mofreak_mip19(mofreak_ctx *ctx, const uint8_t *cur19, const uint8_t *prev19, int64_t n,
                  uint8_t *out_motion8, unsigned flags);
/* The two cv::resize(ROI -> 19x19) of extractMotionByMotionInterchangePatterns for one pair:
 * out is n_kp x 2 x 361 bytes (current ROI, previous ROI); keypoints whose ROI leaves the image get zeros. */
int mofreak_roi19(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H,
                  const mofreak_keypoint *kps, int64_t n_kp, uint8_t *out, unsigned flags);
/* FREAK internals per keypoint of one pair: out_info[k] = {scaleIdx, thetaIdx, direction0, direction1}
 * (thetaIdx = -1 for erased keypoints). */
int mofreak_freak_info(mofreak_ctx *ctx, const uint8_t *cur, const uint8_t *prev, int W, int H,
                       const mofreak_keypoint *kps, int64_t n_kp, int32_t *out_info, unsigned flags);
/* thetaIdx for n (direction0, direction1) int32 pairs. */
int mofreak_theta_index(mofreak_ctx *ctx, const int32_t *dirs, int64_t n, int32_t *out, unsigned flags);
/* Host-side tables the context was built with: patternSizes[64]; scale index of a keypoint size. */
int mofreak_pattern_sizes(const mofreak_ctx *ctx, int32_t out[64]);
int mofreak_scale_index(const mofreak_ctx *ctx, float size, int32_t *out);
/* patternLookup[scale][rot][0..42] as (x, y, sigma) triples. */
int mofreak_table_pattern(const mofreak_ctx *ctx, int scale, int rot, float out[43 * 3]);
/* orientationPairs as (i, j, weight_dx, weight_dy). */
int mofreak_table_orientation(const mofreak_ctx *ctx, int32_t out[45 * 4]);
/* The point pair (i, j) behind each of the 64 bits of descriptor bytes 0..7 (bit b of byte B at index 8B+b). */
int mofreak_table_bit_pairs(const mofreak_ctx *ctx, uint8_t out[128]);
/* The tile kernel's MIP sampling order for ROI side L (1..16): out[64 * u + lane], u < 4, = the position (frame * 368 +
 * row * 19 + col of the (cur19 | prev19) buffer pair) lane resamples in pass u -- the four bytes of one aligned dword --
 * the remaining entries the last pass, byte by byte; *n_out entries in all.  Which lane takes which dword is chosen per L
 * so that a pass's LDS reads spread over the banks (mofreak_amd/tools/mip_lane_order.py). */
int mofreak_table_mip_positions(const mofreak_ctx *ctx, int L, uint16_t out[320], int32_t *n_out);
/* cv::resize(L -> 19) taps, x axis then y axis: (ofs, ofs1, c0, c1) per output index. */
int mofreak_table_resize(const mofreak_ctx *ctx, int L, int16_t out[2 * 19 * 4]);

#ifdef __cplusplus
}
#endif
#endif /* MOFREAK_HIP_H */
