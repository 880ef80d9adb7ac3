// util/common.h -- range helpers the reference's arch sources use (reference src/util/common.h): span / byte in namespace visp,
// membership test on a span.
#pragma once

#include <algorithm>
#include <span>
#include <vector>

#include "string.h"

namespace visp {

template <typename T, size_t Extent, typename U>
constexpr bool contains(std::span<T, Extent> range, U const& value) {
    for (auto const& item : range)
        if (item == value) return true;
    return false;
}

template <typename T>
inline std::span<std::byte const> as_bytes(std::span<T> values) {
    return std::as_bytes(values);
}

} // namespace visp
