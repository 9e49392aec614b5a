// ORACLE / TEST INFRASTRUCTURE ONLY — absl::StrFormat for the single "%0.6lf" use in the reference's common/logging.h
#pragma once
#include <cstdio>
#include <string>
namespace absl {
template <typename... A> std::string StrFormat(const char *fmt, A... a) {
    char buf[256];
    snprintf(buf, sizeof buf, fmt, a...);
    return buf;
}
} // namespace absl
