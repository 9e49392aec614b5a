// ORACLE / TEST INFRASTRUCTURE ONLY — serial stand-in for the one tbb::parallel_for of the reference tracker
// (tracking.cc:656: independent per-block detection jobs; serial execution gives the same results)
#pragma once
namespace tbb {
template <typename T> class blocked_range {
public:
    blocked_range(T b, T e) : b_(b), e_(e) {}
    T begin() const { return b_; }
    T end() const { return e_; }

private:
    T b_, e_;
};
template <typename R, typename F> void parallel_for(const R &range, const F &f) { f(range); }
} // namespace tbb
