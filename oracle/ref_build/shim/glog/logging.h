// ORACLE / TEST INFRASTRUCTURE ONLY — enough of the glog surface for the reference's common/logging.h to parse: all logging
// goes to a null sink.
#pragma once
#include <cstdlib>
#include <iostream>
#include <string>
#define DCHECK_IS_ON() 0
namespace google {
struct NullStream { // discards unless REF_LOG is set in the environment (then: stderr)
    bool on = getenv("REF_LOG") != nullptr;
    template <typename T> NullStream &operator<<(const T &v) { if (on) std::cerr << v; return *this; }
    NullStream &operator<<(std::ostream &(*f) (std::ostream &) ) { if (on) std::cerr << f; return *this; }
};
struct LogLine { // one per LOG(...) statement: appends the newline glog would
    NullStream &s;
    ~LogLine() { if (s.on) std::cerr << "\n"; }
    template <typename T> LogLine &operator<<(const T &v) { s << v; return *this; }
};
struct LogLine;
struct LogMessageVoidify {
    template <typename T> void operator&(const T &) {}
};
inline void InitGoogleLogging(const char *) {}
inline void ShutdownGoogleLogging() {}
} // namespace google
static bool FLAGS_alsologtostderr = false, FLAGS_logtostderr = false, FLAGS_colorlogtostderr = false;
inline google::NullStream &icg_null_log() {
    static google::NullStream s;
    return s;
}
#define LOG(severity) google::LogLine{icg_null_log()}
