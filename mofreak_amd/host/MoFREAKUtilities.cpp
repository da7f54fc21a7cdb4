// MoFREAKUtilities over the C ABI of libmofreak_hip.so.  See MoFREAKUtilities.h.
// Reference lines cited are src/MoFREAK/MoFREAKUtilities.cpp of ChrisWhiten/MoFREAK.
#include "MoFREAKUtilities.h"

#include <sys/stat.h>
#include <unistd.h>

#include "mofreak_dist.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

using std::cout;
using std::endl;
using std::string;

namespace {

// Gray frame stack in NumPy .npy format (v1/v2), dtype uint8, C order, shape (T, H, W).
// the header of a .npy stack of u8 frames: its shape, and the file positioned at the first frame byte
bool open_npy_u8_3d(std::ifstream &f, const string &path, int &T, int &H, int &W)
{
    f.open(path.c_str(), std::ios::binary);
    if (!f) return false;
    char magic[8];
    f.read(magic, 8);
    if (!f || std::memcmp(magic, "\x93NUMPY", 6) != 0) return false;
    size_t hlen = 0;
    if (magic[6] == 1) {
        unsigned char b[2];
        f.read(reinterpret_cast<char *>(b), 2);
        hlen = b[0] | (b[1] << 8);
    } else {
        unsigned char b[4];
        f.read(reinterpret_cast<char *>(b), 4);
        hlen = b[0] | (b[1] << 8) | (b[2] << 16) | ((size_t)b[3] << 24);
    }
    string header(hlen, '\0');
    f.read(&header[0], (std::streamsize)hlen);
    if (!f) return false;
    if (header.find("'|u1'") == string::npos && header.find("'u1'") == string::npos && header.find("'<u1'") == string::npos)
        return false;
    if (header.find("'fortran_order': False") == string::npos) return false;
    const size_t sp = header.find("'shape':");
    if (sp == string::npos) return false;
    const size_t lp = header.find('(', sp), rp = header.find(')', sp);
    if (lp == string::npos || rp == string::npos) return false;
    long dims[3] = {0, 0, 0};
    int nd = 0;
    std::stringstream ss(header.substr(lp + 1, rp - lp - 1));
    string item;
    while (std::getline(ss, item, ',')) {
        std::stringstream is(item);
        long v;
        if (is >> v) {
            if (nd >= 3) return false;
            dims[nd++] = v;
        }
    }
    if (nd != 3 || dims[0] < 0 || dims[1] <= 0 || dims[2] <= 0) return false;
    T = (int)dims[0];
    H = (int)dims[1];
    W = (int)dims[2];
    return true;
}

bool load_npy_u8_3d(const string &path, std::vector<uint8_t> &data, int &T, int &H, int &W)
{
    std::ifstream f;
    if (!open_npy_u8_3d(f, path, T, H, W)) return false;
    data.resize((size_t)T * H * W);
    f.read(reinterpret_cast<char *>(data.data()), (std::streamsize)data.size());
    return (size_t)f.gcount() == data.size();
}

string file_name_of(const string &path)
{
    const size_t p = path.find_last_of("/\\");
    return p == string::npos ? path : path.substr(p + 1);
}

std::vector<string> split(const string &s, char delim)
{
    std::vector<string> out;
    std::stringstream ss(s);
    string item;
    while (std::getline(ss, item, delim)) out.push_back(item);
    return out;
}

const char *const kHmdbFolders[] = {  // enum HMDB_action order (MoSIFTUtilities.h:15-20)
    "brush_hair", "cartwheel", "catch", "chew", "clap", "climb", "climb_stairs", "dive", "draw_sword", "dribble",
    "drink", "eat", "fall_floor", "fencing", "flic_flac", "golf", "handstand", "hit", "hug", "jump", "kick",
    "kick_ball", "kiss", "laugh", "pick", "pour", "pullup", "punch", "push", "pushup", "ride_bike", "ride_horse",
    "run", "shake_hands", "shoot_ball", "shoot_bow", "shoot_gun", "sit", "situp", "smile", "smoke", "somersault",
    "stand", "swing_baseball", "sword", "sword_exercise", "talk", "throw", "turn", "walk", "wave"};

void check(mofreak_ctx *ctx, int rc, const char *what)
{
    if (rc == MOFREAK_OK) return;
    std::ostringstream m;
    m << what << " failed (" << rc << "): " << mofreak_last_error(ctx);
    throw std::runtime_error(m.str());
}

}  // namespace

MoFREAKUtilities::MoFREAKUtilities(int dset)
    : current_action(0), dataset(dset), device_(0), ctx_(nullptr), provider_shared_(true), use_brisk_(false), brisk_threshold_(30),
      brisk_octaves_(3)
{
    mofreak_default_params(&params_);
    setDenseGrid(16, 12.0f, 38);
}

MoFREAKUtilities::~MoFREAKUtilities()
{
    if (ctx_) mofreak_destroy(ctx_);
}

void MoFREAKUtilities::setDevice(int device_id) { device_ = device_id; }
void MoFREAKUtilities::setParams(const mofreak_params &p) { params_ = p; }

void MoFREAKUtilities::setKeypointProvider(KeypointProvider provider, bool same_for_every_frame)
{
    provider_ = provider;
    provider_shared_ = same_for_every_frame;
}

void MoFREAKUtilities::setDenseGrid(int step, float size, int lo)
{
    provider_ = [step, size, lo](int, int W, int H) {
        std::vector<mofreak_keypoint> k;
        for (int y = 0; y < H; y += step)
            if (y > lo && y < H - lo)
                for (int x = 0; x < W; x += step)
                    if (x > lo && x < W - lo) k.push_back(mofreak_keypoint{(float)x, (float)y, size});
        return k;
    };
    provider_shared_ = true;
    use_brisk_ = false;
}

void MoFREAKUtilities::useBriskDetector(int threshold, int octaves)
{
    use_brisk_ = true;
    brisk_threshold_ = threshold;
    brisk_octaves_ = octaves;
}

mofreak_ctx *MoFREAKUtilities::context()
{
    if (!ctx_) {
        mofreak_ctx *c = nullptr;
        const int rc = mofreak_create(device_, &params_, &c);
        if (rc != MOFREAK_OK) {
            std::ostringstream m;
            m << "mofreak_create failed (" << rc << "): " << mofreak_last_error(nullptr);
            throw std::runtime_error(m.str());
        }
        ctx_ = c;
    }
    return ctx_;
}

// ---------------------------------------------------------------------------------------------- the hot path
void MoFREAKUtilities::computeMoFREAKFromFile(std::string video_filename, std::string mofreak_filename,
                                              bool clear_features_after_computation)
{
    std::vector<uint8_t> frames;
    int T = 0, H = 0, W = 0;
    if (!load_npy_u8_3d(video_filename, frames, T, H, W)) {
        cout << "Could not open file: " << video_filename << endl;  // :383-386
        return;  // the reference carries on and dies on the empty frame; nothing sensible to do here
    }
    computeMoFREAKFromFrames(frames.data(), T, W, H, video_filename);

    // in the end, print the mofreak file and reset the features for a new file (:491-496)
    cout << "Writing this mofreak file: " << mofreak_filename << endl;
    writeMoFREAKFeaturesToFile(mofreak_filename);
    if (clear_features_after_computation) features.clear();
}

void MoFREAKUtilities::computeMoFREAKFromFiles(const std::vector<std::string> &video_filenames, const std::vector<std::string> &mofreak_filenames)
{
    if (video_filenames.size() != mofreak_filenames.size()) throw std::runtime_error("computeMoFREAKFromFiles: one output name per video");
    if (!use_brisk_ && !provider_shared_) {  // per-frame keypoints from a caller's provider: the plain loop
        for (size_t i = 0; i < video_filenames.size(); ++i) computeMoFREAKFromFile(video_filenames[i], mofreak_filenames[i], true);
        return;
    }
    // Bounded batches: videos are loaded in the order given until the batch holds batch_bytes_ of frames (or the frame size
    // changes), handed to ONE mofreak_extract_clips call, their files written, their frames let go -- host memory and the
    // work a crash can lose are bounded by the batch, not by the dataset (the per-video file is the checkpoint).
    struct Clip {
        std::vector<uint8_t> frames;
        size_t index = 0;
        int T = 0;
    };
    mofreak_ctx *ctx = context();
    const int gap = params_.gap_for_frame_difference;
    std::vector<Clip> batch;
    int bW = 0, bH = 0;
    size_t bytes = 0;
    auto flush = [&]() {
        if (batch.empty()) return;
        std::vector<const uint8_t *> ptr(batch.size());
        std::vector<int32_t> len(batch.size());
        const std::vector<mofreak_keypoint> kps = use_brisk_ ? std::vector<mofreak_keypoint>() : provider_(gap, bW, bH);
        int64_t capacity = 0;
        for (size_t k = 0; k < batch.size(); ++k) {
            ptr[k] = batch[k].frames.data();
            len[k] = batch[k].T;
            // (the detector's rows are counted afterwards: room for 8192 a pair to begin with, the call says what it needs)
            capacity += (int64_t)std::max(len[k] - gap, 0) * (use_brisk_ ? (int64_t)8192 : (int64_t)kps.size());
        }
        std::vector<mofreak_row> rows((size_t)std::max<int64_t>(capacity, 1));
        std::vector<int64_t> offs(batch.size() + 1, 0);
        int64_t n_rows = 0;
        if (use_brisk_) {  // the reference's keypoint source (:420-423), window by window inside the same pipelined pass
            int rc = mofreak_compute_clips(ctx, ptr.data(), len.data(), (int)batch.size(), bW, bH, 0, brisk_threshold_, brisk_octaves_, rows.data(), capacity, offs.data(),
                                           &n_rows, nullptr, 0);
            if (rc == MOFREAK_ERR_CAPACITY && n_rows > capacity) {
                capacity = n_rows;
                rows.resize((size_t)capacity);
                rc = mofreak_compute_clips(ctx, ptr.data(), len.data(), (int)batch.size(), bW, bH, 0, brisk_threshold_, brisk_octaves_, rows.data(), capacity, offs.data(),
                                           &n_rows, nullptr, 0);
            }
            check(ctx, rc, "mofreak_compute_clips");
        } else {
            check(ctx,
                  mofreak_extract_clips(ctx, ptr.data(), len.data(), (int)batch.size(), bW, bH, /*chunk_frames*/ 0, kps.data(), (int64_t)kps.size(),
                                        rows.data(), capacity, offs.data(), &n_rows, 0),
                  "mofreak_extract_clips");
        }
        for (size_t k = 0; k < batch.size(); ++k) {
            const size_t i = batch[k].index;
            appendRows(rows.data() + offs[k], offs[k + 1] - offs[k], video_filenames[i]);  // behind whatever is there (:374-498 appends)
            cout << "Writing this mofreak file: " << mofreak_filenames[i] << endl;
            writeMoFREAKFeaturesToFile(mofreak_filenames[i]);
            features.clear();  // clear_features_after_computation = true
        }
        batch.clear();
        bytes = 0;
    };
    for (size_t i = 0; i < video_filenames.size(); ++i) {
        Clip c;
        int W = 0, H = 0;
        if (!load_npy_u8_3d(video_filenames[i], c.frames, c.T, H, W)) {
            cout << "Could not open file: " << video_filenames[i] << endl;  // :383-386
            continue;
        }
        c.index = i;
        if (!batch.empty() && (W != bW || H != bH || bytes + c.frames.size() > batch_bytes_)) flush();
        bW = W;
        bH = H;
        bytes += c.frames.size();
        batch.push_back(std::move(c));
    }
    flush();
}

void MoFREAKUtilities::computeMoFREAKFromFilesSharded(const std::vector<std::string> &video_filenames, const std::vector<std::string> &mofreak_filenames,
                                                      mofreak_comm *comm)
{
    if (video_filenames.size() != mofreak_filenames.size()) throw std::runtime_error("computeMoFREAKFromFilesSharded: one output name per video");
    if (!comm) throw std::runtime_error("computeMoFREAKFromFilesSharded: no communicator");
    if (!use_brisk_ && !provider_shared_) throw std::runtime_error("computeMoFREAKFromFilesSharded: a shared keypoint list (dense grid) or the BRISK detector is required");
    auto dist_check = [](int rc, const char *what) {
        if (rc != MOFREAK_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + mofreak_dist_last_error());
    };
    const int rank = mofreak_comm_rank(comm), world = mofreak_comm_world(comm), n = (int)video_filenames.size();
    // the same plan on every rank: costs = file sizes, LPT shard, a rank's share cut into rounds of batch_bytes_
    // (the sizes rank 0 sees, handed to everyone: ranks whose own stat() of a file differed -- a file that appears or
    // goes away, attribute caches of a network file system -- would build different plans and hang in the exchange; a file
    // rank 0 cannot stat costs nothing and is reported when its owner fails to open it)
    std::vector<int64_t> cost((size_t)n, 0);
    if (rank == 0)
        for (int i = 0; i < n; ++i) {
            struct stat st;
            if (stat(video_filenames[i].c_str(), &st) == 0) cost[i] = (int64_t)st.st_size;
        }
    if (n > 0) dist_check(mofreak_allreduce_sum_i64(comm, cost.data(), n), "mofreak_allreduce_sum_i64 (file sizes)");
    std::vector<int32_t> rank_of((size_t)n, 0);
    dist_check(mofreak_shard_lpt(cost.data(), n, world, rank_of.data()), "mofreak_shard_lpt");
    std::vector<std::vector<std::vector<int>>> plan((size_t)world);  // rank -> round -> videos (ascending)
    for (int r = 0; r < world; ++r) {
        std::vector<int> cur;
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) {
            if (rank_of[i] != r) continue;
            if (!cur.empty() && acc + cost[i] > (int64_t)batch_bytes_) {
                plan[r].push_back(cur);
                cur.clear();
                acc = 0;
            }
            cur.push_back(i);
            acc += cost[i];
        }
        if (!cur.empty()) plan[r].push_back(cur);
    }
    size_t n_rounds = 0;
    for (auto &p : plan) n_rounds = std::max(n_rounds, p.size());
    mofreak_ctx *ctx = context();
    const int gap = params_.gap_for_frame_difference;
    struct DeviceRows {  // hipMalloc'd through the C ABI, freed on every way out
        mofreak_ctx *ctx;
        void *p = nullptr;
        ~DeviceRows()
        {
            if (p) (void)mofreak_device_free(ctx, p);
        }
    };
    for (size_t round = 0; round < n_rounds; ++round) {
        std::vector<int> ids;  // the round's videos in gathered order: rank after rank, ascending inside a rank
        for (int r = 0; r < world; ++r)
            if (round < plan[r].size()) ids.insert(ids.end(), plan[r][round].begin(), plan[r][round].end());
        const std::vector<int> mine = round < plan[rank].size() ? plan[rank][round] : std::vector<int>();
        // my clips of the round: loaded, then run by run of one frame size through mofreak_extract_clips, rows left in HBM
        struct Clip {  // frames in page-locked memory of the library (mofreak_host_alloc): mofreak_extract_clips copies them down by
            uint8_t *frames = nullptr;  // DMA straight from here, pageable memory would go through its staging buffer first
            int T = 0, H = 0, W = 0;
            bool ok = false;
            Clip() = default;
            Clip(const Clip &) = delete;
            Clip &operator=(const Clip &) = delete;
            ~Clip()
            {
                if (frames) (void)mofreak_host_free(nullptr, frames);
            }
        };
        std::vector<Clip> clips(mine.size());
        int64_t capacity = 0;
        bool redo = false;
        for (size_t k = 0; k < mine.size(); ++k) {
            Clip &c = clips[k];
            std::ifstream f;
            c.ok = open_npy_u8_3d(f, video_filenames[mine[k]], c.T, c.H, c.W);
            if (c.ok) {
                const size_t bytes = (size_t)c.T * c.H * c.W;
                void *mem = nullptr;
                check(ctx, mofreak_host_alloc(ctx, std::max<size_t>(bytes, 1), &mem), "mofreak_host_alloc");
                c.frames = static_cast<uint8_t *>(mem);
                f.read(reinterpret_cast<char *>(c.frames), (std::streamsize)bytes);
                c.ok = (size_t)f.gcount() == bytes;
            }
            if (!c.ok) {
                cout << "Could not open file: " << video_filenames[mine[k]] << endl;  // :383-386
                continue;
            }
            capacity += (int64_t)std::max(c.T - gap, 0) * (use_brisk_ ? brisk_rows_per_pair_ : (int64_t)provider_(gap, c.W, c.H).size());
        }
        DeviceRows mine_rows{ctx}, all_rows{ctx};
        check(ctx, mofreak_device_alloc(ctx, (size_t)std::max<int64_t>(capacity, 1) * sizeof(mofreak_row), &mine_rows.p), "mofreak_device_alloc");
        std::vector<int64_t> count_of((size_t)mine.size(), 0);
        int64_t n_mine = 0;
        for (size_t k0 = 0; k0 < clips.size();) {
            if (!clips[k0].ok) {
                count_of[k0++] = -1;  // (no file is written for it: the sum over the ranks stays negative)
                continue;
            }
            size_t k1 = k0 + 1;
            while (k1 < clips.size() && clips[k1].ok && clips[k1].W == clips[k0].W && clips[k1].H == clips[k0].H) ++k1;
            std::vector<const uint8_t *> ptr;
            std::vector<int32_t> len;
            for (size_t k = k0; k < k1; ++k) {
                ptr.push_back(clips[k].frames);
                len.push_back(clips[k].T);
            }
            std::vector<int64_t> offs(ptr.size() + 1, 0);
            int64_t got = 0;
            if (use_brisk_) {
                int rc = mofreak_compute_clips(ctx, ptr.data(), len.data(), (int)ptr.size(), clips[k0].W, clips[k0].H, 0, brisk_threshold_, brisk_octaves_,
                                               static_cast<mofreak_row *>(mine_rows.p) + n_mine, capacity - n_mine, offs.data(), &got, nullptr, MOFREAK_ROWS_DEVICE);
                if (rc == MOFREAK_ERR_CAPACITY) {  // more rows than the estimate: the round is done again with room for them
                    brisk_rows_per_pair_ *= 4;
                    redo = true;
                    break;
                }
                check(ctx, rc, "mofreak_compute_clips");
            } else {
                const std::vector<mofreak_keypoint> kps = provider_(gap, clips[k0].W, clips[k0].H);
                check(ctx,
                      mofreak_extract_clips(ctx, ptr.data(), len.data(), (int)ptr.size(), clips[k0].W, clips[k0].H, /*chunk_frames*/ 0, kps.data(), (int64_t)kps.size(),
                                            static_cast<mofreak_row *>(mine_rows.p) + n_mine, capacity - n_mine, offs.data(), &got, MOFREAK_ROWS_DEVICE),
                      "mofreak_extract_clips");
            }
            for (size_t k = k0; k < k1; ++k) count_of[k] = offs[k - k0 + 1] - offs[k - k0];
            n_mine += got;
            k0 = k1;
        }
        if (redo) {  // (every rank walks the same rounds: the rank that repeats one simply arrives later at its exchange)
            --round;
            continue;
        }
        clips.clear();
        // the exchange: per-video counts (every video belongs to one rank: the sum is its count), per-rank counts, rows
        std::vector<int64_t> video_count(ids.size(), 0);
        for (size_t j = 0; j < ids.size(); ++j)
            for (size_t k = 0; k < mine.size(); ++k)
                if (ids[j] == mine[k]) video_count[j] = count_of[k];
        dist_check(mofreak_allreduce_sum_i64(comm, video_count.data(), (int)video_count.size()), "mofreak_allreduce_sum_i64");
        std::vector<int64_t> rank_count((size_t)world, 0);
        dist_check(mofreak_gather_counts(comm, n_mine, rank_count.data()), "mofreak_gather_counts");
        int64_t total = 0;
        for (int64_t c : rank_count) total += c;
        {  // what every rank knows of the round
            int64_t sum = 0;
            for (int64_t c : video_count) sum += std::max<int64_t>(c, 0);
            if (sum != total) throw std::runtime_error("computeMoFREAKFromFilesSharded: per-video and per-rank counts disagree");
        }
        if (files_by_ranks_) {
            // my videos' files, from text made where the rows are: one device call for the round, a segment per video
            check(ctx, mofreak_synchronize(ctx), "mofreak_synchronize");
            std::vector<int64_t> starts;
            std::vector<size_t> which;
            int64_t at = 0;
            for (size_t k = 0; k < mine.size(); ++k) {
                if (count_of[k] < 0) continue;  // could not be opened: no file (as the gathered route)
                starts.push_back(at);
                which.push_back(k);
                at += count_of[k];
            }
            std::vector<size_t> seg(starts.size() + 1, 0);
            size_t need = 0;
            struct HostText {
                void *p = nullptr;
                ~HostText()
                {
                    if (p) (void)mofreak_host_free(nullptr, p);
                }
            } text;
            bool on_device = true;
            int rc = mofreak_format_rows_device(ctx, static_cast<const mofreak_row *>(mine_rows.p), n_mine, nullptr, 0, &need, starts.data(), (int)starts.size(), seg.data());
            if (rc == MOFREAK_OK) {
                check(ctx, mofreak_host_alloc(ctx, std::max<size_t>(need, 1), &text.p), "mofreak_host_alloc");
                rc = mofreak_format_rows_device(ctx, static_cast<const mofreak_row *>(mine_rows.p), n_mine, static_cast<char *>(text.p), need, &need, starts.data(),
                                                (int)starts.size(), seg.data());
            }
            if (rc == MOFREAK_ERR_UNSUPPORTED) {
                on_device = false;  // a float the device leaves to the host formatter: rows to the host, text there
            } else {
                check(ctx, rc, "mofreak_format_rows_device");
            }
            std::vector<mofreak_row> host_rows;
            if (!on_device) {
                host_rows.resize((size_t)std::max<int64_t>(n_mine, 1));
                check(ctx, mofreak_copy_to_host(ctx, host_rows.data(), mine_rows.p, (size_t)n_mine * sizeof(mofreak_row)), "mofreak_copy_to_host");
            }
            for (size_t j = 0; j < which.size(); ++j) {
                const std::string &name = mofreak_filenames[mine[which[j]]];
                cout << "Writing this mofreak file: " << name << endl;
                if (on_device) {
                    writeTextToFile(name, static_cast<const char *>(text.p) + seg[j], seg[j + 1] - seg[j]);
                } else {
                    std::string t;
                    size_t len = 0;
                    (void)mofreak_format_rows(host_rows.data() + starts[j], count_of[which[j]], nullptr, 0, &len);
                    t.resize(len);
                    (void)mofreak_format_rows(host_rows.data() + starts[j], count_of[which[j]], &t[0], len, &len);
                    writeTextToFile(name, t.data(), t.size());
                }
            }
            continue;
        }
        if (rank == 0) check(ctx, mofreak_device_alloc(ctx, (size_t)std::max<int64_t>(total, 1) * sizeof(mofreak_row), &all_rows.p), "mofreak_device_alloc");
        check(ctx, mofreak_synchronize(ctx), "mofreak_synchronize");  // my rows are in place before the exchange stream reads them
        dist_check(mofreak_gather_rows(comm, static_cast<const mofreak_row *>(mine_rows.p), rank_count.data(), 0, static_cast<mofreak_row *>(all_rows.p)),
                   "mofreak_gather_rows");
        if (rank != 0) continue;
        std::vector<mofreak_row> rows((size_t)std::max<int64_t>(total, 1));
        check(ctx, mofreak_copy_to_host(ctx, rows.data(), all_rows.p, (size_t)total * sizeof(mofreak_row)), "mofreak_copy_to_host");  // the root's one copy to the host
        int64_t at = 0;
        for (size_t j = 0; j < ids.size(); ++j) {
            if (video_count[j] < 0) continue;  // could not be opened on its rank
            appendRows(rows.data() + at, video_count[j], video_filenames[ids[j]]);
            at += video_count[j];
            cout << "Writing this mofreak file: " << mofreak_filenames[ids[j]] << endl;
            writeMoFREAKFeaturesToFile(mofreak_filenames[ids[j]]);
            features.clear();
        }
        if (at != total) throw std::runtime_error("computeMoFREAKFromFilesSharded: gathered rows and per-video counts disagree");
    }
}

void MoFREAKUtilities::computeMoFREAKFromFrames(const uint8_t *frames, int T, int W, int H,
                                                const std::string &video_filename)
{
    const int gap = params_.gap_for_frame_difference;  // GAP_FOR_FRAME_DIFFERENCE (:378)
    const int n_pairs = T - gap;
    if (n_pairs <= 0) return;
    mofreak_ctx *ctx = context();
    std::vector<mofreak_row> rows;
    int64_t n_rows = 0;
    if (use_brisk_) {  // detector + descriptors in one call; room for the rows is grown until they fit
        int64_t capacity = (int64_t)n_pairs * 8192;
        for (;;) {
            rows.resize((size_t)capacity);
            const int rc = mofreak_compute_stream(ctx, frames, T, W, H, brisk_threshold_, brisk_octaves_, rows.data(), capacity, &n_rows,
                                                  nullptr, MOFREAK_MEM_HOST);
            if (rc == MOFREAK_ERR_CAPACITY && n_rows > capacity) {
                capacity = n_rows;
                continue;
            }
            check(ctx, rc, "mofreak_compute_stream");
            break;
        }
        appendRows(rows.data(), n_rows, video_filename);
        return;
    }

    // keypoints of every processed frame (frame index gap .. T-1), detector order
    std::vector<mofreak_keypoint> kps;
    std::vector<int64_t> offsets;
    if (provider_shared_) {
        kps = provider_(gap, W, H);
    } else {
        offsets.push_back(0);
        for (int t = gap; t < T; ++t) {
            const std::vector<mofreak_keypoint> k = provider_(t, W, H);
            kps.insert(kps.end(), k.begin(), k.end());
            offsets.push_back((int64_t)kps.size());
        }
    }
    const int64_t capacity = provider_shared_ ? (int64_t)n_pairs * (int64_t)kps.size() : (int64_t)kps.size();
    if (capacity == 0) return;
    rows.resize((size_t)capacity);
    check(ctx,
          mofreak_extract_stream(ctx, frames, T, W, H, kps.data(), provider_shared_ ? nullptr : offsets.data(),
                                 (int64_t)kps.size(), rows.data(), capacity, &n_rows, MOFREAK_MEM_HOST),
          "mofreak_extract_stream");
    appendRows(rows.data(), n_rows, video_filename);
}

void MoFREAKUtilities::appendRows(const mofreak_row *rows, int64_t n_rows, const std::string &video_filename)
{
    int action = 0, person = 0, video_number = 0;
    readMetadata(video_filename, action, video_number, person);  // the reference re-parses this per keypoint (:469)
    for (int64_t i = 0; i < n_rows; ++i) {
        const mofreak_row &r = rows[i];
        MoFREAKFeature ftr(NUMBER_OF_BYTES_FOR_MOTION, NUMBER_OF_BYTES_FOR_APPEARANCE);
        ftr.frame_number = r.frame_number;
        ftr.scale = r.scale;
        ftr.x = r.x;
        ftr.y = r.y;
        for (int b = 0; b < NUMBER_OF_BYTES_FOR_APPEARANCE; ++b) ftr.appearance[b] = r.appearance[b];
        for (int b = 0; b < NUMBER_OF_BYTES_FOR_MOTION; ++b) ftr.motion[b] = r.motion[b];
        ftr.action = action;
        ftr.video_number = video_number;
        ftr.person = person;
        ftr.motion_x = 0;  // :476-477
        ftr.motion_y = 0;
        features.push_back(ftr);
    }
}

bool MoFREAKUtilities::buildMoFREAKFeature(const uint8_t *current_frame, const uint8_t *prev_frame, int W, int H,
                                           float x, float y, float size, int frame_number, MoFREAKFeature &out)
{
    mofreak_ctx *ctx = context();
    const mofreak_keypoint kp = {x, y, size};
    uint8_t desc[MOFREAK_DESC_BYTES];
    uint8_t valid = 0;
    check(ctx,
          mofreak_extract_pairs(ctx, current_frame, prev_frame, W, H, W, (int64_t)W * H, 1, &kp, nullptr, 1, desc, &valid,
                                MOFREAK_MEM_HOST),
          "mofreak_extract_pairs");
    if (!valid) return false;
    out = MoFREAKFeature(NUMBER_OF_BYTES_FOR_MOTION, NUMBER_OF_BYTES_FOR_APPEARANCE);
    out.x = x;
    out.y = y;
    out.scale = size;
    out.frame_number = frame_number;
    for (int b = 0; b < 8; ++b) out.appearance[b] = desc[b];
    for (int b = 0; b < 8; ++b) out.motion[b] = desc[8 + b];
    return true;
}

// ---------------------------------------------------------------------------------------------- files
// Both directions go through the library's row text (mofreak_format_rows / mofreak_parse_rows, the byte-exact
// counterparts of :691-719 and :1146-1190): the facade only converts between MoFREAKFeature and the 32-byte row.
namespace {

mofreak_row row_of(const MoFREAKFeature &f)
{
    mofreak_row r;
    r.x = f.x;
    r.y = f.y;
    r.frame_number = f.frame_number;
    r.scale = f.scale;  // motion_x / motion_y are always 0 on this path (:476-477) and are printed as such
    for (int b = 0; b < MOFREAK_APPEARANCE_BYTES; ++b) r.appearance[b] = static_cast<uint8_t>(f.appearance[b]);
    for (int b = 0; b < MOFREAK_MOTION_BYTES; ++b) r.motion[b] = static_cast<uint8_t>(f.motion[b]);
    return r;
}

}  // namespace

void MoFREAKUtilities::writeTextToFile(const std::string &output_file, const char *text, size_t len)
{
    const std::string tmp = output_file + ".tmp";
    FILE *out = std::fopen(tmp.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot write " + tmp);
    const bool written = len == 0 || std::fwrite(text, 1, len, out) == len;
    const bool synced = written && std::fflush(out) == 0 && fsync(fileno(out)) == 0;
    if (std::fclose(out) != 0 || !synced || std::rename(tmp.c_str(), output_file.c_str()) != 0) throw std::runtime_error("cannot finish " + output_file);
}

void MoFREAKUtilities::writeMoFREAKFeaturesToFile(string output_file)
{
    // The text goes to <output_file>.tmp and is renamed over the target once it is complete: a run killed while writing
    // leaves no truncated .mofreak file behind (the per-video file is what a resumed run takes for "done").
    const std::string tmp = output_file + ".tmp";
    FILE *out = std::fopen(tmp.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot write " + tmp);
    const size_t batch = 1 << 16;
    std::vector<mofreak_row> rows;
    std::vector<char> text;
    for (size_t at = 0; at < features.size(); at += batch) {
        const size_t n = std::min(batch, features.size() - at);
        rows.resize(n);
        for (size_t k = 0; k < n; ++k) rows[k] = row_of(features[at + k]);
        size_t need = 0;
        check(nullptr, mofreak_format_rows(rows.data(), (int64_t)n, nullptr, 0, &need), "mofreak_format_rows");
        text.resize(need);
        check(nullptr, mofreak_format_rows(rows.data(), (int64_t)n, text.data(), text.size(), &need), "mofreak_format_rows");
        if (std::fwrite(text.data(), 1, need, out) != need) {
            std::fclose(out);
            throw std::runtime_error("short write to " + tmp);
        }
    }
    // on disk before the name exists: the same durability as the Python mirror's write_atomic (flush, fsync, rename)
    const bool synced = std::fflush(out) == 0 && fsync(fileno(out)) == 0;
    if (std::fclose(out) != 0 || !synced || std::rename(tmp.c_str(), output_file.c_str()) != 0)
        throw std::runtime_error("cannot finish " + output_file);
}

void MoFREAKUtilities::readMoFREAKFeatures(std::string filename, int num_to_sample)
{
    int action = 0, video_number = 0, person = 0;
    readMetadata(filename, action, video_number, person);

    std::string text;
    if (FILE *in = std::fopen(filename.c_str(), "rb")) {
        char buf[1 << 16];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, in)) > 0) text.append(buf, got);
        std::fclose(in);
    }  // a file that does not open reads as no features, like the reference's stream that is never good()
    int64_t n = 0;
    check(nullptr, mofreak_parse_rows(text.data(), text.size(), nullptr, 0, &n), "mofreak_parse_rows");
    std::vector<mofreak_row> rows((size_t)n);
    if (n) check(nullptr, mofreak_parse_rows(text.data(), text.size(), rows.data(), n, &n), "mofreak_parse_rows");

    // file order -> the order the reference leaves in `features`: it collects the file in a scratch deque and moves
    // it over back to front (:1206-1210), or, when sampling, shuffles the scratch deque and moves num_to_sample
    // entries over from its back (:1194-1203)
    std::vector<size_t> order(rows.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    size_t take = order.size();
    if (num_to_sample && (int)order.size() > num_to_sample) {
        std::random_shuffle(order.begin(), order.end());
        take = (size_t)num_to_sample;
    }
    for (size_t k = 0; k < take; ++k) {
        const mofreak_row &r = rows[order[order.size() - 1 - k]];
        MoFREAKFeature ftr(NUMBER_OF_BYTES_FOR_MOTION, NUMBER_OF_BYTES_FOR_APPEARANCE);
        ftr.x = r.x;
        ftr.y = r.y;
        ftr.frame_number = r.frame_number;
        ftr.scale = r.scale;
        for (int b = 0; b < MOFREAK_APPEARANCE_BYTES; ++b) ftr.appearance[b] = r.appearance[b];
        for (int b = 0; b < MOFREAK_MOTION_BYTES; ++b) ftr.motion[b] = r.motion[b];
        ftr.action = action;
        ftr.video_number = video_number;
        ftr.person = person;
        features.push_back(ftr);
    }
}

std::deque<MoFREAKFeature> MoFREAKUtilities::getMoFREAKFeatures() { return features; }

void MoFREAKUtilities::setAllFeaturesToLabel(int label)
{
    for (unsigned i = 0; i < features.size(); ++i) features[i].action = label;
}

void MoFREAKUtilities::clearFeatures() { features.clear(); }

// ---------------------------------------------------------------------------------------------- labels
void MoFREAKUtilities::setCurrentAction(string folder_name)
{
    // :782-1060
    if (dataset == HMDB51) {
        for (int i = 0; i < (int)(sizeof(kHmdbFolders) / sizeof(kHmdbFolders[0])); ++i)
            if (folder_name == kHmdbFolders[i]) {
                current_action = i;
                return;
            }
        // the reference prints, system("PAUSE")s and exit(1)s here (:1041-1047); a library must not
        current_action = BRUSH_HAIR;
        cout << "****Didn't find action" << endl;
        throw std::runtime_error("MoFREAKUtilities::setCurrentAction: unknown HMDB51 folder '" + folder_name + "'");
    } else if (dataset == UCF101) {
        if (actions.find(folder_name) == actions.end()) {
            const int id = (int)actions.size();  // (the reference's actions[f] = actions.size() is order-ambiguous)
            actions[folder_name] = id;
        }
        current_action = actions[folder_name];
    }
}

void MoFREAKUtilities::readMetadata(const std::string &filename, int &action, int &video_number, int &person)
{
    // :1062-1134 (the KTH branch is disabled upstream with `if (false)`)
    const string file_name_str = file_name_of(filename);
    if (dataset == HMDB51) {
        video_number = 0;
        person = 0;
        action = current_action;
    } else if (dataset == UTI2) {
        const std::vector<string> parts = split(file_name_str, '_');
        if (parts.size() >= 3) {
            std::stringstream(parts[1]) >> person;
            std::stringstream(parts[2].substr(0, 1)) >> action;
            std::stringstream(parts[0]) >> video_number;
        }
    }
}
