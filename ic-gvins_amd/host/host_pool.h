// A small persistent thread pool for the host phases that run between device launches (per-stream tracker stages of a TrackingBatch
// group, per-window phases of a WindowSolverBatch): creating threads per phase costs more than the phases themselves.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace icg {

// Persistent helpers for the per-stream host stages of one group: parallelFor(n, f) runs f(0..n-1) on the caller plus the
// helper threads (dynamic index claiming).  Helpers spin briefly between dispatches (stages follow each other every
// ~100 us) and sleep on a condition variable when the group is idle.
class HostPool {
public:
    explicit HostPool(int n_threads);
    ~HostPool();
    void parallelFor(int n, const std::function<void(int)> &f);
    int threads() const { return (int) helpers_.size() + 1; }

private:
    void helperLoop();
    void drain();
    std::vector<std::thread> helpers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> next_{0}, acks_{0}, sleepers_{0};
    int n_{0};
    const std::function<void(int)> *fn_{nullptr};
    std::atomic<bool> stop_{false};
    std::string error_;
};

// One persistent helper thread that runs a call beside the caller (WindowSolverBatch: the device half of a linearization while the caller
// drives the pool through the host half).  start(f) hands f over, wait() returns when it is through; creating a std::thread per call cost
// up to 100 us in a process with many threads and mappings (the bench), more than the hand-over was meant to save.
class SideThread {
public:
    SideThread() : t_([this] { loop(); }) {}
    ~SideThread() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        t_.join();
    }
    SideThread(const SideThread &) = delete;
    SideThread &operator=(const SideThread &) = delete;
    void start(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lock(m_);
            fn_   = std::move(f);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return !busy_; });
    }

private:
    void loop() {
        std::unique_lock<std::mutex> lock(m_);
        for (;;) {
            cv_.wait(lock, [&] { return stop_ || busy_; });
            if (stop_) return;
            std::function<void()> f = std::move(fn_);
            lock.unlock();
            f();
            lock.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> fn_;
    bool busy_{false}, stop_{false};
    std::thread t_; // (last: the loop uses the members above)
};

} // namespace icg
