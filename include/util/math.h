// util/math.h -- lane-wise arithmetic on the small fixed-size vectors of the public headers (reference src/util/math.h): f32x4 and
// i64x2 are std::array here, i32x2 is the extent type of visp/vision.h. One generic definition per operator instead of one per type.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>

#include "../visp/vision.h"

namespace visp {
using std::clamp;
using i64x2 = std::array<int64_t, 2>;
using i64x4 = std::array<int64_t, 4>;

constexpr int32_t div_ceil(int32_t a, int32_t b) { return (a + b - 1) / b; }
constexpr int64_t div_ceil(int64_t a, int64_t b) { return (a + b - 1) / b; }
constexpr int32_t next_multiple(int32_t x, int32_t m) { return div_ceil(x, m) * m; }
constexpr float sqr(float x) { return x * x; }

namespace detail {
template <typename T, size_t N, typename F>
constexpr std::array<T, N> lanes(std::array<T, N> const& a, std::array<T, N> const& b, F f) {
    std::array<T, N> r{};
    for (size_t i = 0; i < N; ++i) r[i] = f(a[i], b[i]);
    return r;
}
template <typename T, size_t N>
constexpr std::array<T, N> splat(T v) {
    std::array<T, N> r{};
    for (size_t i = 0; i < N; ++i) r[i] = v;
    return r;
}
} // namespace detail

// std::array<T, N> (f32x4, i64x2): vector (op) vector, vector (op) scalar, scalar (op) vector, unary minus
#define VISP_LANE_OP(op)                                                                                                               \
    template <typename T, size_t N>                                                                                                    \
    constexpr std::array<T, N> operator op(std::array<T, N> const& a, std::array<T, N> const& b) {                                     \
        return detail::lanes(a, b, [](T x, T y) { return T(x op y); });                                                                \
    }                                                                                                                                  \
    template <typename T, size_t N, typename S>                                                                                        \
        requires std::is_arithmetic_v<S>                                                                                               \
    constexpr std::array<T, N> operator op(std::array<T, N> const& a, S b) { return a op detail::splat<T, N>(T(b)); }                  \
    template <typename T, size_t N, typename S>                                                                                        \
        requires std::is_arithmetic_v<S>                                                                                               \
    constexpr std::array<T, N> operator op(S a, std::array<T, N> const& b) { return detail::splat<T, N>(T(a)) op b; }
VISP_LANE_OP(+)
VISP_LANE_OP(-)
VISP_LANE_OP(*)
VISP_LANE_OP(/)
#undef VISP_LANE_OP
template <typename T, size_t N>
constexpr std::array<T, N> operator-(std::array<T, N> const& a) { return detail::splat<T, N>(T(0)) - a; }
constexpr float dot(f32x4 const& a, f32x4 const& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// i32x2: extents
constexpr i32x2 operator+(i32x2 a, i32x2 b) { return {a[0] + b[0], a[1] + b[1]}; }
constexpr i32x2 operator-(i32x2 a, i32x2 b) { return {a[0] - b[0], a[1] - b[1]}; }
constexpr i32x2 operator*(i32x2 a, i32x2 b) { return {a[0] * b[0], a[1] * b[1]}; }
constexpr i32x2 operator/(i32x2 a, i32x2 b) { return {a[0] / b[0], a[1] / b[1]}; }
constexpr i32x2 operator*(i32x2 a, int32_t s) { return a * i32x2(s, s); }
constexpr i32x2 operator/(i32x2 a, int32_t s) { return a / i32x2(s, s); }
constexpr i32x2 div_ceil(i32x2 a, int32_t b) { return {div_ceil(a[0], b), div_ceil(a[1], b)}; }
constexpr i32x2 next_multiple(i32x2 x, int32_t m) { return div_ceil(x, m) * m; }
constexpr i32x2 min(i32x2 a, i32x2 b) { return {std::min(a[0], b[0]), std::min(a[1], b[1])}; }

} // namespace visp
