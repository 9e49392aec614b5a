// oracle/ (test infrastructure): the reference's own CPU decomposition for the cpu_baseline leg of bench.py —
// OpenCV runs calcOpticalFlowPyrLK's points in parallel (parallel_for_ in modules/video/src/lkpyramid.cpp) and the reference runs the
// detection blocks in tbb::parallel_for (tracking/tracking.cc:656).  ICG_ORACLE_INNER_THREADS=T (read at every call) splits those two
// loops over T threads; every item writes its own outputs, so results are identical to the serial loops.  Default: serial.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

inline int orc_inner_threads() {
    const char *e = getenv("ICG_ORACLE_INNER_THREADS");
    return e ? std::max(1, atoi(e)) : 1;
}

// f(begin, end) over contiguous chunks of [0, n)
template <typename F> inline void orc_parallel_chunks(int n, F &&f) {
    const int t = std::min(orc_inner_threads(), n);
    if (t <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int per = (n + t - 1) / t;
    for (int k = 1; k < t; k++) {
        const int b = k * per, e = std::min(n, b + per);
        if (b < e) th.emplace_back([&f, b, e] { f(b, e); });
    }
    f(0, std::min(n, per));
    for (auto &x : th) x.join();
}
