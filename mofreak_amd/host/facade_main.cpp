// Small driver over the C++ MoFREAKUtilities facade, used by tests/test_facade.py and as a usage example:
//   facade_main extract <video.npy> <out.mofreak> [grid_step grid_size grid_lo | brisk]   (needs a GPU)
//   facade_main files <video_dir> <mofreak_dir>      computeMoFREAKFiles() of main.cpp:854-924 for *.npy (needs a GPU)
//   facade_main roundtrip <in.mofreak> <out.mofreak>  read (reversed, as the reference) + write (no GPU)
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "MoFREAKUtilities.h"

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

int main(int argc, char **argv)
{
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "extract" && argc >= 4) {
            MoFREAKUtilities mofreak(MoFREAKUtilities::KTH);
            if (argc >= 7) mofreak.setDenseGrid(std::atoi(argv[4]), (float)std::atof(argv[5]), std::atoi(argv[6]));
            if (argc == 5 && std::string(argv[4]) == "brisk") mofreak.useBriskDetector();  // the reference's own keypoint source
            mofreak.computeMoFREAKFromFile(argv[2], argv[3], false);
            std::cout << mofreak.getMoFREAKFeatures().size() << " features" << std::endl;
            return 0;
        }
        if (mode == "files" && argc >= 4) {
            // main.cpp:854-924: files directly in VIDEO_PATH, and one level of per-action folders
            const std::string video_path = argv[2], mofreak_path = argv[3];
            MoFREAKUtilities mofreak(MoFREAKUtilities::UCF101);
            mofreak.setDenseGrid(16, 7.0f, 23);
            // The walk is the reference's; the videos it finds are handed over together (one pipelined call per frame
            // size instead of one synchronous call per video), which writes the same files.
            std::vector<std::string> videos, outputs;
            for (const std::string &name : list_dir(video_path)) {
                const std::string p = video_path + "/" + name;
                if (!is_dir(p)) {
                    if (ends_with(name, "npy")) {
                        videos.push_back(p);
                        outputs.push_back(mofreak_path + "/" + name + ".mofreak");
                    }
                } else {
                    std::cout << "action: " << name << std::endl;
                    mofreak.setCurrentAction(name);
                    mkdir((mofreak_path + "/" + name).c_str(), 0777);
                    for (const std::string &v : list_dir(p))
                        if (ends_with(v, "npy")) {
                            videos.push_back(p + "/" + v);
                            outputs.push_back(mofreak_path + "/" + name + "/" + v + ".mofreak");
                        }
                }
            }
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
