#pragma once
#include <string_view>
namespace absl {
using string_view = std::string_view;
}
