// util/string.h -- what the reference's arch sources expect under this name (reference src/util/string.h): the ASSERT macro and a
// printf-style message helper. This backend reports violated preconditions as visp::exception (the C ABI turns it into rc 0 + message);
// it never aborts the host process.
#pragma once

#include <cstdio>
#include <string>

#include "../visp/vision.h"

namespace visp {

[[noreturn]] inline void precondition_failed(char const* file, int line, char const* text) {
    throw exception(std::string("precondition failed at ") + file + ":" + std::to_string(line) + ": " + text);
}

template <typename... Args>
inline exception except(char const* fmt, Args... args) {
    char buf[256];
    std::snprintf(buf, sizeof buf, fmt, args...);
    return exception(buf);
}

} // namespace visp

#ifndef ASSERT
#    define ASSERT(cond, ...) ((cond) ? (void)0 : ::visp::precondition_failed(__FILE__, __LINE__, #cond))
#endif
