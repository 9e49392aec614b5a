// ORACLE / TEST INFRASTRUCTURE ONLY — enough of the glog surface for the reference's common/logging.h to parse: all logging
// goes to a null sink.
#pragma once
#include <iostream>
#include <string>
#define DCHECK_IS_ON() 0
namespace google {
struct NullStream {
    template <typename T> NullStream &operator<<(const T &) { return *this; }
    NullStream &operator<<(std::ostream &(*) (std::ostream &) ) { return *this; }
};
struct LogMessageVoidify {
    void operator&(NullStream &) {}
};
inline void InitGoogleLogging(const char *) {}
inline void ShutdownGoogleLogging() {}
} // namespace google
static bool FLAGS_alsologtostderr = false, FLAGS_logtostderr = false, FLAGS_colorlogtostderr = false;
inline google::NullStream &icg_null_log() {
    static google::NullStream s;
    return s;
}
#define LOG(severity) icg_null_log()
