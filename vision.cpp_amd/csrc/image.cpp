#include "image.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "visp_util.h"

namespace visp {

int n_channels(image_format f) {
    switch (f) {
        case image_format::rgba_u8: case image_format::bgra_u8: case image_format::argb_u8: case image_format::rgba_f32: return 4;
        case image_format::rgb_u8: case image_format::rgb_f32: return 3;
        case image_format::alpha_u8: case image_format::alpha_f32: return 1;
    }
    throw except("Unsupported image format [%d]", int(f));
}
bool is_float(image_format f) { return int(f) >= int(image_format::rgba_f32); }
int n_bytes(image_format f) { return n_channels(f) * (is_float(f) ? 4 : 1); }

image_data image_alloc(i32x2 extent, image_format format) {
    size_t size = (size_t)extent[0] * extent[1] * n_bytes(format);
    return image_data{extent, format, std::unique_ptr<uint8_t[]>(new uint8_t[size ? size : 1])};
}
image_view view_of(image_data const& d) {
    return image_view{d.extent, d.extent[0] * n_bytes(d.format), d.format, d.data.get()};
}
size_t n_bytes(image_view const& v) { return (size_t)v.extent[0] * v.extent[1] * n_bytes(v.format); }

namespace {
void channel_map(image_format f, int m[4]) { // image.cpp get_channel_map
    m[0] = 0; m[1] = 1; m[2] = 2; m[3] = 3;
    if (f == image_format::bgra_u8) { m[0] = 2; m[1] = 1; m[2] = 0; m[3] = 3; }
    if (f == image_format::argb_u8) { m[0] = 1; m[1] = 2; m[2] = 3; m[3] = 0; }
}
void load_u8(const uint8_t* p, int ch, const int map[4], float v[4]) { // image-impl.h:17-34
    if (ch == 1) { v[0] = v[1] = v[2] = v[3] = float(p[0]) / 255.0f; }
    else if (ch == 3) { v[0] = float(p[0]) / 255.0f; v[1] = float(p[1]) / 255.0f; v[2] = float(p[2]) / 255.0f; v[3] = 1.0f / 255.0f; }
    else { for (int c = 0; c < 4; ++c) v[c] = float(p[map[c]]) / 255.0f; }
}
} // namespace

image_data image_u8_to_f32(image_view const& src, image_format format, const float offset[4], const float scale[4]) {
    VISP_ASSERT(!is_float(src.format) && is_float(format));
    int sch = n_channels(src.format), dch = n_channels(format);
    if ((dch == 1) != (sch == 1)) throw except("Number of channels in source and destination are not compatible");
    image_data dst = image_alloc(src.extent, format);
    float* d = reinterpret_cast<float*>(dst.data.get());
    int map[4];
    channel_map(src.format, map);
    const uint8_t* s = static_cast<const uint8_t*>(src.data);
    for (int y = 0; y < src.extent[1]; ++y)
        for (int x = 0; x < src.extent[0]; ++x) {
            float v[4];
            load_u8(s + (size_t)y * src.stride + (size_t)x * sch, sch, map, v);
            float* o = d + ((size_t)y * src.extent[0] + x) * dch;
            for (int c = 0; c < dch; ++c) o[c] = (v[c] + offset[c]) * scale[c];
        }
    return dst;
}

image_data image_f32_to_u8(image_view const& src, image_format format, float scale, float offset) {
    VISP_ASSERT(is_float(src.format) && !is_float(format));
    int sch = n_channels(src.format), dch = n_channels(format);
    if (!((dch == 1 && sch == 1) || (dch == 4 && sch >= 3)))
        throw except("Number of channels in source and destination are not compatible");
    if (format != image_format::alpha_u8 && format != image_format::rgba_u8)
        throw except("Unsupported image format [%d]", int(format));
    image_data dst = image_alloc(src.extent, format);
    const float* s = static_cast<const float*>(src.data);
    size_t srow = (size_t)src.stride / 4;
    for (int y = 0; y < src.extent[1]; ++y)
        for (int x = 0; x < src.extent[0]; ++x) {
            const float* p = s + (size_t)y * srow + (size_t)x * sch;
            float v[4] = {p[0], sch > 1 ? p[1] : p[0], sch > 1 ? p[2] : p[0], sch == 4 ? p[3] : (sch == 3 ? 1.0f : p[0])};
            uint8_t* o = dst.data.get() + ((size_t)y * src.extent[0] + x) * dch;
            for (int c = 0; c < dch; ++c) o[c] = uint8_t(std::clamp(v[c] * scale + offset, 0.0f, 1.0f) * 255.0f);
        }
    return dst;
}

image_data image_normalize(image_view const& src, float min, float max) {
    VISP_ASSERT(is_float(src.format) && min < max);
    int ch = n_channels(src.format);
    image_data dst = image_alloc(src.extent, src.format);
    const float* s = static_cast<const float*>(src.data);
    float* d = reinterpret_cast<float*>(dst.data.get());
    size_t srow = (size_t)src.stride / 4, drow = (size_t)src.extent[0] * ch;
    float mn[4] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX}, mx[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int y = 0; y < src.extent[1]; ++y)
        for (int x = 0; x < src.extent[0]; ++x)
            for (int c = 0; c < ch; ++c) {
                float v = s[y * srow + (size_t)x * ch + c];
                mn[c] = std::min(mn[c], v);
                mx[c] = std::max(mx[c], v);
            }
    float scale[4], offset[4];
    for (int c = 0; c < ch; ++c) {
        float delta = mx[c] - mn[c];
        delta = delta < 1e-5f ? 1.0f : delta;
        scale[c] = (max - min) / delta;
        offset[c] = -mn[c] * scale[c] + min;
    }
    for (int y = 0; y < src.extent[1]; ++y)
        for (int x = 0; x < src.extent[0]; ++x)
            for (int c = 0; c < ch; ++c) d[y * drow + (size_t)x * ch + c] = s[y * srow + (size_t)x * ch + c] * scale[c] + offset[c];
    return dst;
}

image_data image_to_rgb_u8(image_view const& img) {
    if (is_float(img.format)) throw except("Unsupported image format [%d]", int(img.format));
    int sch = n_channels(img.format);
    int map[4];
    channel_map(img.format, map);
    image_data dst = image_alloc(img.extent, image_format::rgb_u8);
    const uint8_t* s = static_cast<const uint8_t*>(img.data);
    for (int y = 0; y < img.extent[1]; ++y)
        for (int x = 0; x < img.extent[0]; ++x) {
            const uint8_t* p = s + (size_t)y * img.stride + (size_t)x * sch;
            uint8_t* o = dst.data.get() + ((size_t)y * img.extent[0] + x) * 3;
            if (sch == 1) { o[0] = o[1] = o[2] = p[0]; }
            else if (sch == 3) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
            else { o[0] = p[map[0]]; o[1] = p[map[1]]; o[2] = p[map[2]]; }
        }
    return dst;
}

} // namespace visp
