#include "errors.h"

namespace {
thread_local std::string g_last_error;
}

namespace alfa {
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace alfa

extern "C" const char* alfa_last_error(void) { return g_last_error.c_str(); }
