// Host image types and algorithms with the reference's semantics
// (include/visp/image.h, src/visp/image.cpp, src/visp/image-impl.h). Only what the
// Depth-Anything path and its C ABI need; file IO (stb) is out of scope.
#pragma once
#include <cstdint>
#include <memory>

namespace visp {

enum class image_format : int32_t { rgba_u8, bgra_u8, argb_u8, rgb_u8, alpha_u8, rgba_f32, rgb_f32, alpha_f32 };

int n_channels(image_format);
int n_bytes(image_format);
bool is_float(image_format);

struct i32x2 {
    int32_t v[2];
    int32_t& operator[](int i) { return v[i]; }
    int32_t operator[](int i) const { return v[i]; }
    bool operator==(i32x2 const& o) const { return v[0] == o.v[0] && v[1] == o.v[1]; }
    bool operator!=(i32x2 const& o) const { return !(*this == o); }
};

// ABI-identical to the reference's image_view (include/visp/image.h:37-41) and to visp_image_view
struct image_view {
    i32x2 extent{};
    int32_t stride = 0;
    image_format format = image_format::rgba_u8;
    void const* data = nullptr;
};

struct image_data {
    i32x2 extent{};
    image_format format = image_format::rgba_u8;
    std::unique_ptr<uint8_t[]> data;
};

image_data image_alloc(i32x2 extent, image_format format);
image_view view_of(image_data const&);
size_t n_bytes(image_view const&);

// image.cpp:215-255: dst = (src/255 + offset) * scale, any u8 format -> float format
image_data image_u8_to_f32(image_view const& src, image_format format, const float offset[4], const float scale[4]);
// image.cpp:257-288: uint8(clamp(src*scale + offset, 0, 1) * 255)
image_data image_f32_to_u8(image_view const& src, image_format format, float scale = 1, float offset = 0);
// image.cpp:537-582
image_data image_normalize(image_view const& img, float min = 0, float max = 1);
// image.cpp:328-356: stb_image_resize v1 semantics (Catmull-Rom enlarging, Mitchell reducing, sRGB-correct for u8, alpha-weighted,
// edge clamp), restated in image_resize.cpp
image_data image_scale(image_view const& img, i32x2 target);
// converts any supported u8 format to tightly packed rgb_u8 (channel map of image.cpp get_channel_map)
image_data image_to_rgb_u8(image_view const& img);

} // namespace visp
