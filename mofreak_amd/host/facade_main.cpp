// Small driver over the C++ MoFREAKUtilities facade, used by tests/test_facade.py and as a usage example:
//   facade_main extract <video.npy> <out.mofreak> [grid_step grid_size grid_lo | brisk]   (needs a GPU)
//   facade_main files <video_dir> <mofreak_dir>      computeMoFREAKFiles() of main.cpp:854-924 for *.npy (needs a GPU)
//   facade_main files-rank <rank> <N> <id_file> <video_dir> <mofreak_dir>   one rank of the same over N GPUs: every rank writes the
//                                                     files of its own videos from text made on the device; with
//                                                     MOFREAK_GATHER_TO_ROOT=1 the rows are gathered to rank 0 over RCCL and
//                                                     rank 0 writes; started N times by `facade_ranks N <video_dir> <mofreak_dir>`
//                                                     (facade_ranks.c: a launcher that loads no GPU library at all)
//   facade_main roundtrip <in.mofreak> <out.mofreak>  read (reversed, as the reference) + write (no GPU)
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <iostream>
#include <string>
#include <vector>

#include "MoFREAKUtilities.h"
#include "mofreak_dist.h"

static bool ends_with(const std::string &s, const std::string &suffix)
{
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

static std::vector<std::string> list_dir(const std::string &dir)
{
    std::vector<std::string> out;
    if (DIR *d = opendir(dir.c_str())) {
        while (dirent *e = readdir(d)) {
            const std::string n = e->d_name;
            if (n != "." && n != "..") out.push_back(n);
        }
        closedir(d);
    }
    std::sort(out.begin(), out.end());
    return out;
}

static bool is_dir(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// main.cpp:854-924: files directly in VIDEO_PATH, and one level of per-action folders (the same walk on every rank)
static void walk_dataset(MoFREAKUtilities &mofreak, const std::string &video_path, const std::string &mofreak_path, bool make_dirs,
                         std::vector<std::string> &videos, std::vector<std::string> &outputs)
{
    for (const std::string &name : list_dir(video_path)) {
        const std::string p = video_path + "/" + name;
        if (!is_dir(p)) {
            if (ends_with(name, "npy")) {
                videos.push_back(p);
                outputs.push_back(mofreak_path + "/" + name + ".mofreak");
            }
        } else {
            if (make_dirs) std::cout << "action: " << name << std::endl;
            mofreak.setCurrentAction(name);
            if (make_dirs) mkdir((mofreak_path + "/" + name).c_str(), 0777);
            for (const std::string &v : list_dir(p))
                if (ends_with(v, "npy")) {
                    videos.push_back(p + "/" + v);
                    outputs.push_back(mofreak_path + "/" + name + "/" + v + ".mofreak");
                }
        }
    }
}

static int run_rank(int rank, int world, const std::string &id_file, const std::string &video_path, const std::string &mofreak_path)
{
    unsigned char id[MOFREAK_UNIQUE_ID_BYTES];
    if (rank == 0) {
        if (mofreak_comm_unique_id(id) != MOFREAK_OK) throw std::runtime_error(std::string("mofreak_comm_unique_id: ") + mofreak_dist_last_error());
        const std::string tmp = id_file + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id || std::fclose(f) != 0 || std::rename(tmp.c_str(), id_file.c_str()) != 0)
            throw std::runtime_error("cannot write " + id_file);
    } else {
        bool got = false;
        for (int tries = 0; tries < 1200 && !got; ++tries) {  // two minutes
            if (FILE *f = std::fopen(id_file.c_str(), "rb")) {
                got = std::fread(id, 1, sizeof id, f) == sizeof id;
                std::fclose(f);
            }
            if (!got) usleep(100000);
        }
        if (!got) throw std::runtime_error("rank " + std::to_string(rank) + ": no RCCL id from rank 0 in " + id_file);
    }
    mofreak_comm *comm = nullptr;
    if (mofreak_comm_create(id, rank, world, /*device*/ rank, &comm) != MOFREAK_OK)
        throw std::runtime_error(std::string("mofreak_comm_create: ") + mofreak_dist_last_error());
    int rc = 0;
    try {
        if (world == 1 && mofreak_comm_self_exchange(comm, 1 << 20) != MOFREAK_OK)  // the point-to-point path, on a box with one GPU
            throw std::runtime_error(std::string("mofreak_comm_self_exchange: ") + mofreak_dist_last_error());
        MoFREAKUtilities mofreak(MoFREAKUtilities::UCF101);
        mofreak.setDevice(rank);
        mofreak.setDenseGrid(16, 7.0f, 23);
        if (const char *b = std::getenv("MOFREAK_BATCH_BYTES")) mofreak.setBatchBytes((size_t)std::atoll(b));
        if (const char *g = std::getenv("MOFREAK_GATHER_TO_ROOT")) mofreak.setFilesWrittenByTheirRanks(std::atoi(g) == 0);  // 1: rows to rank 0 over RCCL, rank 0 writes
        if (const char *k = std::getenv("MOFREAK_USE_BRISK")) {
            if (std::atoi(k)) mofreak.useBriskDetector();  // the reference's own keypoint source instead of the dense grid
        }
        std::vector<std::string> videos, outputs;
        walk_dataset(mofreak, video_path, mofreak_path, rank == 0, videos, outputs);
        mofreak.computeMoFREAKFromFilesSharded(videos, outputs, comm);
        if (rank == 0) std::cout << "ranks " << world << " videos " << videos.size() << std::endl;
    } catch (...) {
        mofreak_comm_destroy(comm);
        throw;
    }
    mofreak_comm_destroy(comm);
    return rc;
}

int main(int argc, char **argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "files-rank" && argc >= 7) return run_rank(std::atoi(argv[2]), std::atoi(argv[3]), argv[4], argv[5], argv[6]);
        if (mode == "extract" && argc >= 4) {
            MoFREAKUtilities mofreak(MoFREAKUtilities::KTH);
            if (argc >= 7) mofreak.setDenseGrid(std::atoi(argv[4]), (float)std::atof(argv[5]), std::atoi(argv[6]));
            if (argc == 5 && std::string(argv[4]) == "brisk") mofreak.useBriskDetector();  // the reference's own keypoint source
            mofreak.computeMoFREAKFromFile(argv[2], argv[3], false);
            std::cout << mofreak.getMoFREAKFeatures().size() << " features" << std::endl;
            return 0;
        }
        if (mode == "files" && argc >= 4) {
            const std::string video_path = argv[2], mofreak_path = argv[3];
            MoFREAKUtilities mofreak(MoFREAKUtilities::UCF101);
            mofreak.setDenseGrid(16, 7.0f, 23);
            if (const char *bb = std::getenv("MOFREAK_BATCH_BYTES")) mofreak.setBatchBytes((size_t)std::atoll(bb));
            if (const char *k = std::getenv("MOFREAK_USE_BRISK")) {
                if (std::atoi(k)) mofreak.useBriskDetector();
            }
            // The walk is the reference's; the videos it finds are handed over together (pipelined calls per batch and
            // frame size instead of one synchronous call per video), which writes the same files.
            std::vector<std::string> videos, outputs;
            walk_dataset(mofreak, video_path, mofreak_path, true, videos, outputs);
            mofreak.computeMoFREAKFromFiles(videos, outputs);
            return 0;
        }
        if (mode == "roundtrip" && argc >= 4) {
            MoFREAKUtilities mofreak(MoFREAKUtilities::KTH);
            mofreak.readMoFREAKFeatures(argv[2]);
            mofreak.writeMoFREAKFeaturesToFile(argv[3]);
            std::cout << mofreak.getMoFREAKFeatures().size() << " features" << std::endl;
            return 0;
        }
        std::cerr << "usage: facade_main extract|files|roundtrip ..." << std::endl;
        return 2;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
}
