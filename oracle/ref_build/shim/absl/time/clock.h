// ORACLE / TEST INFRASTRUCTURE ONLY — absl::Now / Time / Duration as used by the reference's common/timecost.h
#pragma once
#include <chrono>
namespace absl {
struct Duration {
    double s = 0;
};
struct Time {
    double t = 0;
};
inline Duration operator-(const Time &a, const Time &b) { return Duration{a.t - b.t}; }
inline Time Now() { return Time{std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()}; }
inline double ToDoubleSeconds(Duration d) { return d.s; }
inline double ToDoubleMilliseconds(Duration d) { return d.s * 1e3; }
} // namespace absl
