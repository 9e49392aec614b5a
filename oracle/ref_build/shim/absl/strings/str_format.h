// ORACLE / TEST INFRASTRUCTURE ONLY — absl::StrFormat / StrAppendFormat over snprintf (printf-style formats only, as used by
// the reference's common/logging.h, common/timecost.h, fileio/filesaver.cc)
#pragma once
#include <cstdio>
#include <string>
#include <string_view>
namespace absl {
template <typename... A> std::string StrFormat(std::string_view fmt, A... a) {
    const std::string f(fmt);
    char buf[512];
    snprintf(buf, sizeof buf, f.c_str(), a...);
    return buf;
}
template <typename... A> void StrAppendFormat(std::string *dst, std::string_view fmt, A... a) { *dst += StrFormat(fmt, a...); }
} // namespace absl
