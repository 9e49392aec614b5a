// Section timers for the host layer (enabled with ICG_HOST_PROF=1; two clock reads per section otherwise skipped).
// Accumulators are process-wide atomics: sections are coarse (per stream per stage), so contention is negligible.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <ctime>

namespace icg {
namespace hostprof {

enum Section {
    BEGIN_FRAME = 0, ON_PREPROCESS, ON_DETECT_A, ON_LK, ON_RANSAC, ON_TRIANGULATE, ON_DETECT_B, DIGEST, KEEPER,
    DEV_PREPROCESS, DEV_DETECT, DEV_LK, DEV_RANSAC, DEV_TRIANGULATE, GATHER, SCATTER, STEP_TOTAL, LK_MAP_FEATURES, LK_MAP_PARALLAX, LK_REF, KEEP_INSERT, KEEP_REMOVE, DET_INTEGRATE, QUEUE_MAP, QUEUE_REF, N_SECTIONS
};

inline std::atomic<uint64_t> *ns() {
    static std::atomic<uint64_t> a[N_SECTIONS];
    return a;
}
inline std::atomic<uint64_t> *calls() {
    static std::atomic<uint64_t> a[N_SECTIONS];
    return a;
}
inline bool enabled() {
    static const bool on = getenv("ICG_HOST_PROF") != nullptr;
    return on;
}
// ICG_HOST_PROF=cpu: sections accumulate the calling thread's CPU time (what the host cores are actually spent on: a thread that
// sleeps in a poll wait or is descheduled does not count) instead of wall time
inline bool cpu_clock() {
    static const bool on = getenv("ICG_HOST_PROF") != nullptr && getenv("ICG_HOST_PROF")[0] == 'c';
    return on;
}
inline uint64_t now_ns() {
    if (cpu_clock()) {
        struct timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (uint64_t) ts.tv_sec * 1000000000ull + (uint64_t) ts.tv_nsec;
    }
    return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Scope {
    int id;
    uint64_t t0;
    explicit Scope(int s) : id(s), t0(enabled() ? now_ns() : 0) {}
    ~Scope() {
        if (t0) {
            ns()[id].fetch_add(now_ns() - t0, std::memory_order_relaxed);
            calls()[id].fetch_add(1, std::memory_order_relaxed);
        }
    }
};
inline const char *name(int s) {
    static const char *n[N_SECTIONS] = {"begin_frame", "on_preprocess", "on_detect_a", "on_lk", "on_ransac", "on_triangulate",
                                        "on_detect_b", "digest", "keeper", "dev_preprocess", "dev_detect", "dev_lk",
                                        "dev_ransac", "dev_triangulate", "gather", "scatter", "step_total", "lk_map_features", "lk_map_parallax", "lk_ref", "keep_insert", "keep_remove",
                                        "det_integrate", "queue_map", "queue_ref"};
    return n[s];
}

} // namespace hostprof
} // namespace icg
