// errors.h -- thread-local error reporting shared by every translation unit of libalfa_hot.
#pragma once
#include <string>

#include "../../include/alfa_hot.h"

namespace alfa {
// records `msg` as the calling thread's last error and returns `code`
int fail(int code, const std::string& msg);
}  // namespace alfa
