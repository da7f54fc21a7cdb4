// MoFREAKUtilities -- the reference's class interface for the extraction path, implemented over the C ABI of
// libmofreak_hip.so (include/mofreak_hip.h).  No OpenCV, no Boost.
//
// Mirrors src/MoFREAK/MoFREAKUtilities.h:23-104 of ChrisWhiten/MoFREAK: same struct, same public method names,
// argument meaning and side effects, so code written against the reference class compiles against this one.
// Differences, all forced by the environment:
//   * "video files" are raw gray frame stacks (NumPy .npy, uint8, shape (T, H, W)): there is no video decoder
//     on either box (the reference uses cv::VideoCapture + BGR2GRAY, MoFREAKUtilities.cpp:380-410);
//   * keypoints: useBriskDetector() runs the reference's detector, BriskFeatureDetector(30) on the difference image
//     (:420-423), on the GPU (mofreak_compute_stream); the default is a KeypointProvider with a dense grid, the
//     configuration the benchmark is quoted on;
//   * the dead MoSIFT code path (buildMoFREAKFeaturesFromMoSIFT, :598-663, never called) is not provided.
#ifndef MOFREAK_HOST_MOFREAKUTILITIES_H
#define MOFREAK_HOST_MOFREAKUTILITIES_H

#include <cstdint>
#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mofreak_hip.h"

#define MOTION_BYTES 8
#define APPEARANCE_BYTES 8

// MoSIFTUtilities.h:14-20 (only needed for label parsing)
enum KTH_action { BOXING, HANDCLAPPING, HANDWAVING, JOGGING, RUNNING, WALKING };
enum HMDB_action {
    BRUSH_HAIR, CARTWHEEL, CATCH, CHEW, CLAP, CLIMB, CLIMB_STAIRS, DIVE, DRAW_SWORD, DRIBBLE, DRINK, EAT, FALL_FLOOR,
    FENCING, FLIC_FLAC, GOLF, HANDSTAND, HIT, HUG, JUMP, KICK, KICK_BALL, KISS, LAUGH, PICK, POUR, PULLUP, PUNCH, PUSH,
    PUSHUP, RIDE_BIKE, RIDE_HORSE, RUN, SHAKE_HANDS, SHOOT_BALL, SHOOT_BOW, SHOOT_GUN, SIT, SITUP, SMILE, SMOKE,
    SOMERSAULT, STAND, SWING_BASEBALL, SWORD, SWORD_EXERCISE, TALK, THROW, TURN, WALK, WAVE
};

struct MoFREAKFeature {  // MoFREAKUtilities.h:23-53
    MoFREAKFeature(int motion_bytes, int appearance_bytes)
        : using_image_difference(false), x(0), y(0), scale(0), motion_x(0), motion_y(0), frame_number(0),
          motion(motion_bytes, 0u), appearance(appearance_bytes, 0u), action(0), video_number(0), person(0)
    {
    }
    bool using_image_difference;
    float x, y, scale, motion_x, motion_y;
    int frame_number;
    std::vector<unsigned int> motion;
    std::vector<unsigned int> appearance;
    int action, video_number, person;
};

class MoFREAKUtilities {
public:
    enum datasets { KTH, TRECVID, HOLLYWOOD, UTI1, UTI2, HMDB51, UCF101 };  // MoFREAKUtilities.h:103

    explicit MoFREAKUtilities(int dset);
    ~MoFREAKUtilities();
    MoFREAKUtilities(const MoFREAKUtilities &) = delete;
    MoFREAKUtilities &operator=(const MoFREAKUtilities &) = delete;

    // ---- the reference's public interface (MoFREAKUtilities.h:58-73)
    void readMoFREAKFeatures(std::string filename, int num_to_sample = 0);
    std::deque<MoFREAKFeature> getMoFREAKFeatures();
    void clearFeatures();
    void writeMoFREAKFeaturesToFile(std::string output_file);
    void computeMoFREAKFromFile(std::string video_filename, std::string mofreak_filename,
                                bool clear_features_after_computation);
    void setAllFeaturesToLabel(int label);
    void setCurrentAction(std::string folder_name);
    int current_action;
    std::unordered_map<std::string, int> actions;
    static const int NUMBER_OF_BYTES_FOR_APPEARANCE = APPEARANCE_BYTES;
    static const int NUMBER_OF_BYTES_FOR_MOTION = MOTION_BYTES;

    // ---- the north-star spelling of the same operations (BASELINE.json); thin aliases
    void computeMoFREAKFeatures(std::string video_filename, std::string mofreak_filename,
                                bool clear_features_after_computation)
    {
        computeMoFREAKFromFile(video_filename, mofreak_filename, clear_features_after_computation);
    }
    // One keypoint of one gray frame pair; returns false if cv::FREAK would have erased the keypoint.
    bool buildMoFREAKFeature(const uint8_t *current_frame, const uint8_t *prev_frame, int W, int H, float x, float y,
                             float size, int frame_number, MoFREAKFeature &out);

    // ---- what replaces cv::VideoCapture and the BRISK detector here
    // (frame index, W, H) -> keypoints of that frame, in detector order
    typedef std::function<std::vector<mofreak_keypoint>(int, int, int)> KeypointProvider;
    void setKeypointProvider(KeypointProvider provider, bool same_for_every_frame);
    void setDenseGrid(int step, float size, int lo);  // x = step*i, y = step*j, lo < x < W-lo, lo < y < H-lo
    void useBriskDetector(int threshold = 30, int octaves = 3);  // cv::BriskFeatureDetector(30) (:420-421), brisk.h:293
    void setDevice(int device_id);                    // before the first computation; default 0
    void setParams(const mofreak_params &p);          // before the first computation
    // The frame loop of computeMoFREAKFromFile on frames already in memory (T x H x W gray).
    void computeMoFREAKFromFrames(const uint8_t *frames, int T, int W, int H, const std::string &video_filename);
    // The body of computeMoFREAKFiles' loops (main.cpp:862-921) for many videos at once: what
    //     for (i...) computeMoFREAKFromFile(video_filenames[i], mofreak_filenames[i], true);
    // does -- every video's features written to its own file, nothing kept -- with consecutive videos of one frame size
    // going through ONE mofreak_extract_clips call (shared launches, copies under kernels) per batch of at most
    // setBatchBytes() of frames when the keypoints are a shared list (dense grid); other keypoint sources fall back to the
    // loop above.  A batch is loaded, extracted, written and freed before the next one is read: memory and the work a
    // crash loses are bounded by the batch.  Files come out byte-identical to the loop's.
    void computeMoFREAKFromFiles(const std::vector<std::string> &video_filenames, const std::vector<std::string> &mofreak_filenames);
    void setBatchBytes(size_t bytes) { batch_bytes_ = bytes > 0 ? bytes : 1; }  // default 2 GiB of frames per batch
    // The same over N GPUs, one process per GPU (the reference's loop is sequential): every rank calls this
    // with the same lists and its communicator (include/mofreak_dist.h).  Videos go to ranks longest-processing-time
    // first (file sizes); the ranks walk their shares in rounds of at most setBatchBytes() of frames: a rank's clips of a
    // round through mofreak_extract_clips with the rows left in HBM, the per-video row counts summed over the ranks, the
    // rows gathered to rank 0 over RCCL (ncclAllGather of the counts, grouped ncclSend / ncclRecv peer -> root), and rank 0
    // writes the round's files -- the bytes computeMoFREAKFromFiles writes.  Dense-grid keypoints, or the BRISK detector
    // (useBriskDetector(): mofreak_compute_clips, the detector window by window inside the pipelined pass).
    // setFilesWrittenByTheirRanks(true) (the default; one node = one file system): the files are the output, so every rank
    // writes those of ITS OWN videos -- text made on the device from the rows in HBM (mofreak_format_rows_device, one call per
    // round), file by file through <name>.tmp + fsync + rename -- and only the counts are exchanged: no row crosses a link,
    // rank 0 formats nothing.  false: the rows are gathered to rank 0 over RCCL and rank 0 writes everything (above).
    void setFilesWrittenByTheirRanks(bool on) { files_by_ranks_ = on; }
    void computeMoFREAKFromFilesSharded(const std::vector<std::string> &video_filenames, const std::vector<std::string> &mofreak_filenames,
                                        struct mofreak_comm *comm);

private:
    void readMetadata(const std::string &filename, int &action, int &video_number, int &person);
    void appendRows(const mofreak_row *rows, int64_t n_rows, const std::string &video_filename);
    static void writeTextToFile(const std::string &output_file, const char *text, size_t len);  // <file>.tmp + fsync + rename
    mofreak_ctx *context();

    std::deque<MoFREAKFeature> features;
    int dataset;
    int device_;
    mofreak_params params_;
    mofreak_ctx *ctx_;
    KeypointProvider provider_;
    bool provider_shared_;
    size_t batch_bytes_ = (size_t)2 << 30;
    bool files_by_ranks_ = true;
    int64_t brisk_rows_per_pair_ = 8192;  // first estimate of a detector stream's rows per frame pair (grown when a call needs more)
    bool use_brisk_;
    int brisk_threshold_, brisk_octaves_;
};

#endif
