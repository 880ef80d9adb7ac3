// util/string.h -- what the reference's arch sources expect under this name (reference src/util/string.h): the ASSERT macro and a
// "{}"-style message helper. This backend reports violated preconditions as visp::exception (the C ABI turns it into rc 0 + message);
// it never aborts the host process.
#pragma once

#include <cstdio>
#include <sstream>
#include <string>

#include "../visp/vision.h"

namespace visp {

[[noreturn]] inline void precondition_failed(char const* file, int line, char const* text) {
    throw exception(std::string("precondition failed at ") + file + ":" + std::to_string(line) + ": " + text);
}

// message with "{}" placeholders filled from the arguments in order (the reference formats its messages with fmt)
namespace detail {
inline void format_into(std::ostringstream& o, char const* fmt) { o << fmt; }
template <typename T, typename... Rest>
inline void format_into(std::ostringstream& o, char const* fmt, T const& v, Rest const&... rest) {
    for (; *fmt; ++fmt) {
        if (fmt[0] == '{' && fmt[1] == '}') {
            o << v;
            return format_into(o, fmt + 2, rest...);
        }
        o << *fmt;
    }
}
} // namespace detail
template <typename... Args>
inline exception except(char const* fmt, Args const&... args) {
    std::ostringstream o;
    detail::format_into(o, fmt, args...);
    return exception(o.str());
}

} // namespace visp

#ifndef ASSERT
#    define ASSERT(cond, ...) ((cond) ? (void)0 : ::visp::precondition_failed(__FILE__, __LINE__, #cond))
#endif
