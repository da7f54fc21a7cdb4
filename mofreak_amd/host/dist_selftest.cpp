// The N-GPU exchange logic (csrc/dist_gather.h) over an in-process stand-in for the communicator: `world` threads, host
// memory for "device" buffers, blocking mailboxes for send / recv.  What runs over RCCL on the GPUs runs here unchanged:
// the LPT shard and the gather (root copies its own rows, one recv per peer with rows, one send per peer).
//   dist_selftest            -> "ok" and exit code 0
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "../csrc/dist_gather.h"

namespace {

struct Mailboxes {  // (src, dst) -> bytes in flight
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::vector<char>> box;
};

struct FakeTransport {
    Mailboxes *mb;
    int rank;
    int sends = 0, recvs = 0, groups = 0;
    int group_start()
    {
        ++groups;
        return 0;
    }
    int group_end() { return 0; }
    int send(const void *p, int64_t bytes, int peer)
    {
        ++sends;
        std::lock_guard<std::mutex> l(mb->m);
        mb->box[{rank, peer}].assign(static_cast<const char *>(p), static_cast<const char *>(p) + bytes);
        mb->cv.notify_all();
        return 0;
    }
    int recv(void *p, int64_t bytes, int peer)
    {
        ++recvs;
        std::unique_lock<std::mutex> l(mb->m);
        mb->cv.wait(l, [&] { return mb->box.count({peer, rank}) > 0; });
        std::vector<char> &v = mb->box[{peer, rank}];
        if ((int64_t)v.size() != bytes) return 7;
        std::memcpy(p, v.data(), (size_t)bytes);
        mb->box.erase({peer, rank});
        return 0;
    }
    int copy_local(void *dst, const void *src, int64_t bytes)
    {
        std::memcpy(dst, src, (size_t)bytes);
        return 0;
    }
    int sync() { return 0; }
};

#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                  \
        }                                                              \
    } while (0)

int gather_case(const std::vector<int64_t> &counts, int root)
{
    const int world = (int)counts.size();
    const int64_t row = 32;
    Mailboxes mb;
    std::vector<std::vector<char>> rows((size_t)world);
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        rows[r].resize((size_t)(counts[r] * row) + 5);  // (capacity > count, like the device buffers)
        for (size_t i = 0; i < rows[r].size(); ++i) rows[r][i] = (char)(r * 37 + i * 11 + 3);
        total += counts[r];
    }
    std::vector<char> out((size_t)(total * row) + 1, (char)0x5a);
    std::vector<int> rc((size_t)world, -1), sends((size_t)world, 0), recvs((size_t)world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            FakeTransport t{&mb, r};
            rc[r] = mofreak_dist::gather_rows(t, r, world, rows[r].data(), counts.data(), root, r == root ? out.data() : nullptr, row);
            sends[r] = t.sends;
            recvs[r] = t.recvs;
        });
    for (auto &t : th) t.join();
    int64_t at = 0;
    for (int r = 0; r < world; ++r) {
        CHECK(rc[r] == 0);
        CHECK(std::memcmp(out.data() + at * row, rows[r].data(), (size_t)(counts[r] * row)) == 0);  // rank order = concatenation
        at += counts[r];
        CHECK(sends[r] == (r != root && counts[r] ? 1 : 0));  // every peer with rows: one send, straight to the root
        if (r != root) CHECK(recvs[r] == 0);
    }
    int with_rows = 0;
    for (int r = 0; r < world; ++r) with_rows += r != root && counts[r] ? 1 : 0;
    CHECK(recvs[root] == with_rows);
    CHECK(out.back() == (char)0x5a && mb.box.empty());
    return 0;
}

}  // namespace

int main()
{
    // LPT: a partition, deterministic, balanced to within the largest item
    {
        std::vector<int64_t> cost;
        unsigned long long s = 12345;
        for (int i = 0; i < 6766; ++i) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            cost.push_back(20 + (int64_t)((s >> 33) % 630));
        }
        for (int world : {1, 2, 3, 8}) {
            std::vector<int32_t> a(cost.size()), b(cost.size());
            mofreak_dist::shard_lpt(cost.data(), (int)cost.size(), world, a.data());
            mofreak_dist::shard_lpt(cost.data(), (int)cost.size(), world, b.data());
            CHECK(a == b);
            std::vector<int64_t> load((size_t)world, 0);
            for (size_t i = 0; i < cost.size(); ++i) {
                CHECK(a[i] >= 0 && a[i] < world);
                load[a[i]] += cost[i];
            }
            CHECK(*std::max_element(load.begin(), load.end()) - *std::min_element(load.begin(), load.end()) <= 650);
        }
        const int64_t two[2] = {5, 1};
        int32_t r2[2];
        mofreak_dist::shard_lpt(two, 2, 4, r2);
        CHECK(r2[0] == 0 && r2[1] == 1);
    }
    if (gather_case({300, 45}, 0)) return 1;
    if (gather_case({0, 17}, 0)) return 1;
    if (gather_case({64, 0}, 0)) return 1;
    if (gather_case({5, 0, 9}, 0)) return 1;
    if (gather_case({7, 1, 2, 3, 0, 4, 5, 6}, 0)) return 1;
    if (gather_case({4, 3, 2}, 2)) return 1;
    if (gather_case({11}, 0)) return 1;
    std::puts("ok");
    return 0;
}
