// Small host-side utilities (mirrors the role of the reference's include/visp/util.h and
// src/util/string.h: a library exception type with a bounded message, printf-style format).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <string>

namespace visp {

struct exception : std::exception {
    char message[256];
    explicit exception(const char* msg) { snprintf(message, sizeof message, "%s", msg); }
    const char* what() const noexcept override { return message; }
};

inline exception except(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
inline exception except(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return exception(buf);
}

#define VISP_ASSERT(cond)                                                                  \
    do {                                                                                   \
        if (!(cond)) throw ::visp::except("Assertion failed at %s:%d: %s", __FILE__, __LINE__, #cond); \
    } while (0)

constexpr int32_t div_ceil(int32_t a, int32_t b) { return (a + b - 1) / b; }
constexpr int64_t div_ceil(int64_t a, int64_t b) { return (a + b - 1) / b; }
constexpr int32_t next_multiple(int32_t x, int32_t m) { return div_ceil(x, m) * m; }

} // namespace visp

// IEEE binary16 <-> binary32 (round to nearest even); host-side weight packing
namespace visp {
inline float f16_to_f32(uint16_t h) {
    uint32_t sign = uint32_t(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            bits = sign | uint32_t(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112) << 23 | man << 13;
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    __builtin_memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return uint16_t(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
    if (ax >= 0x477ff000u) return uint16_t(sign | 0x7c00u);
    if (ax < 0x33000001u) return uint16_t(sign);
    int e = int(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t q = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    uint32_t hv = e < -14 ? q : (uint32_t(e + 14) << 10) + q;
    return uint16_t(sign | hv);
}
} // namespace visp
