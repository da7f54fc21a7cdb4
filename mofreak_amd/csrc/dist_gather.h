// The exchange step of the N-GPU run, written against a small transport interface so that the same code
// runs over RCCL (dist_rccl.cpp) and over an in-process stand-in on the CPU (host/dist_selftest.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace mofreak_dist {

// Longest-processing-time-first: videos by descending cost (ties: lower index first) to the least loaded rank (ties: lower rank).
inline void shard_lpt(const int64_t *costs, int n, int world, int32_t *rank_of)
{
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return costs[a] > costs[b]; });
    std::vector<int64_t> load((size_t)world, 0);
    for (int i : order) {
        int r = 0;
        for (int k = 1; k < world; ++k)
            if (load[k] < load[r]) r = k;
        rank_of[i] = r;
        load[r] += costs[i];
    }
}

// Transport: group_start / group_end bracket point-to-point operations that have to progress together; send / recv move
// `bytes` bytes of the transport's ("device") memory; copy_local is a copy within the rank; all are ordered on the
// transport's stream and complete at sync().  Returns 0 or an error code of the transport's.
template <class T>
int gather_rows(T &t, int rank, int world, const void *rows, const int64_t *counts, int root, void *out, int64_t row_bytes)
{
    int rc = 0;
    if (rank == root) {
        std::vector<int64_t> at((size_t)world + 1, 0);
        for (int r = 0; r < world; ++r) at[r + 1] = at[r] + counts[r];
        if (counts[rank] && (rc = t.copy_local(static_cast<char *>(out) + at[rank] * row_bytes, rows, counts[rank] * row_bytes))) return rc;
        if ((rc = t.group_start())) return rc;
        for (int r = 0; r < world && !rc; ++r)
            if (r != rank && counts[r]) rc = t.recv(static_cast<char *>(out) + at[r] * row_bytes, counts[r] * row_bytes, r);
        const int rc2 = t.group_end();
        if (rc || rc2) return rc ? rc : rc2;
    } else if (counts[rank]) {
        if ((rc = t.group_start())) return rc;
        rc = t.send(rows, counts[rank] * row_bytes, root);
        const int rc2 = t.group_end();
        if (rc || rc2) return rc ? rc : rc2;
    }
    return t.sync();
}

}  // namespace mofreak_dist
