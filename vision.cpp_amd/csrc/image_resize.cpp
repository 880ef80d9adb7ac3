// image_scale with the semantics of the call the reference makes (src/visp/image.cpp:328-356):
//   stbir_resize_{uint8,float}_generic(..., alpha_channel(format), flags 0, STBIR_EDGE_CLAMP, STBIR_FILTER_DEFAULT,
//                                      STBIR_COLORSPACE_SRGB for u8 / _LINEAR for float)
// stb is not part of the reference tree (depend/stb fetches nothings/stb @ 5736b15f, stb_image_resize.h v0.97); this is
// a restatement of that file's published algorithm, anchored on the reference's own vector (tests/test-image.cpp:186-203,
// tests/test_oracle_golden.py):
//   * per axis: Catmull-Rom when the axis is enlarged (scale > 1), Mitchell-Netravali (B = C = 1/3) otherwise -- also at
//     scale 1 -- with the kernel stretched by 1/scale when reducing; pixel centres at n + 0.5; reads outside the image clamp;
//   * reducing: every input pixel (including a margin of clamped ones) spreads kernel(x) * scale over the output pixels in
//     reach, then each output pixel's weights are normalised to sum 1; enlarging: every output pixel gathers normalised weights;
//   * u8: colour channels through the sRGB -> linear table and back through the fp32 -> sRGB8 table conversion stb uses
//     (a monotonic mapping that round-trips all 256 codes and differs from exact rounding by one code on ~1 % of inputs),
//     alpha linear (v / 255, round half up); images with an alpha channel are filtered with colours premultiplied by
//     (alpha + 2^-80) and divided again afterwards; float images are filtered as they are (premultiplied likewise).
// Horizontal pass first, then vertical, f32 accumulation in ascending source order (stb's order for both of its loop forms).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "image.h"
#include "visp_util.h"

namespace visp {
namespace {

float catmull_rom(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return 1 - x * x * (2.5f - 1.5f * x);
    if (x < 2.0f) return 2 - x * (4 + x * (0.5f * x - 2.5f));
    return 0.0f;
}
float mitchell(float x) {
    x = std::fabs(x);
    if (x < 1.0f) return (16 + x * x * (21 * x - 36)) / 18;
    if (x < 2.0f) return (32 + x * (-60 + x * (36 - 7 * x))) / 18;
    return 0.0f;
}

// one output sample = sum of w[i] * src[clamp(first + i)]
struct taps { int first = 0; std::vector<float> w; };

std::vector<taps> axis_taps(int in_size, int out_size) {
    const float scale = (float)out_size / (float)in_size;
    const float support = 2.0f; // both kernels
    std::vector<taps> out((size_t)out_size);
    if (scale > 1.0f) { // enlarge: gather, Catmull-Rom
        const float out_radius = support * scale;
        for (int n = 0; n < out_size; ++n) {
            const float centre = (float)n + 0.5f;
            const float lo = (centre - out_radius) / scale, hi = (centre + out_radius) / scale;
            const float in_centre = centre / scale;
            int first = (int)std::floor(lo + 0.5f), last = (int)std::floor(hi - 0.5f);
            taps& t = out[(size_t)n];
            float total = 0;
            std::vector<float> w;
            for (int p = first; p <= last; ++p) {
                float c = catmull_rom(in_centre - ((float)p + 0.5f));
                if (w.empty() && c == 0.0f) { ++first; continue; } // stb skips leading zeros
                w.push_back(c);
                total += c;
            }
            while (!w.empty() && w.back() == 0.0f) w.pop_back();
            const float norm = 1 / total;
            for (float& c : w) c *= norm;
            t.first = first;
            t.w = std::move(w);
        }
        return out;
    }
    // reduce (or copy): scatter from every input pixel within the margin, Mitchell stretched by 1 / scale
    const int width = (int)std::ceil(support * 2 / scale);
    const int margin = width / 2;
    const float in_radius = support / scale;
    struct contrib { int n0, n1; std::vector<float> c; };
    std::vector<contrib> cs;
    cs.reserve((size_t)in_size + 2 * margin);
    for (int n = -margin; n < in_size + margin; ++n) {
        const float centre = (float)n + 0.5f;
        const float lo = (centre - in_radius) * scale, hi = (centre + in_radius) * scale;
        const float out_centre = centre * scale;
        contrib c;
        c.n0 = (int)std::floor(lo + 0.5f);
        c.n1 = (int)std::floor(hi - 0.5f);
        for (int k = c.n0; k <= c.n1; ++k) c.c.push_back(mitchell(((float)k + 0.5f) - out_centre) * scale);
        cs.push_back(std::move(c));
    }
    for (int i = 0; i < out_size; ++i) { // normalise per output pixel, contributors in ascending input order
        float total = 0;
        for (contrib& c : cs) {
            if (i >= c.n0 && i <= c.n1) total += c.c[(size_t)(i - c.n0)];
            else if (i < c.n0) break;
        }
        const float norm = 1 / total;
        for (contrib& c : cs) {
            if (i >= c.n0 && i <= c.n1) c.c[(size_t)(i - c.n0)] *= norm;
            else if (i < c.n0) break;
        }
    }
    // transpose into gather form: ascending input index per output pixel
    std::vector<int> last_in((size_t)out_size, 0);
    for (int i = 0; i < out_size; ++i) out[(size_t)i].first = INT32_MAX;
    for (size_t j = 0; j < cs.size(); ++j) {
        const int n = (int)j - margin;
        for (int k = std::max(cs[j].n0, 0); k <= std::min(cs[j].n1, out_size - 1); ++k) {
            const float c = cs[j].c[(size_t)(k - cs[j].n0)];
            taps& t = out[(size_t)k];
            if (t.first == INT32_MAX) {
                if (c == 0.0f) continue; // leading zeros are skipped by stb as well
                t.first = n;
            }
            t.w.resize((size_t)(n - t.first) + 1, 0.0f);
            t.w[(size_t)(n - t.first)] = c;
        }
    }
    for (taps& t : out)
        if (t.first == INT32_MAX) { t.first = 0; t.w.assign(1, 0.0f); }
    return out;
}

const float* srgb_to_linear_table() {
    static float table[256];
    static bool ready = false;
    if (!ready) {
        for (int i = 0; i < 256; ++i) {
            double v = i / 255.0;
            table[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
        }
        ready = true;
    }
    return table;
}

// fp32 -> sRGB8 as stb_image_resize.h does it (table-driven piecewise-linear conversion over 13 binades x 8 segments)
uint8_t linear_to_srgb_u8(float in) {
    static const uint32_t tab[104] = {
        0x0073000d, 0x007a000d, 0x0080000d, 0x0087000d, 0x008d000d, 0x0094000d, 0x009a000d, 0x00a1000d, 0x00a7001a, 0x00b4001a, 0x00c1001a,
        0x00ce001a, 0x00da001a, 0x00e7001a, 0x00f4001a, 0x0101001a, 0x010e0033, 0x01280033, 0x01410033, 0x015b0033, 0x01750033, 0x018f0033,
        0x01a80033, 0x01c20033, 0x01dc0067, 0x020f0067, 0x02430067, 0x02760067, 0x02aa0067, 0x02dd0067, 0x03110067, 0x03440067, 0x037800ce,
        0x03df00ce, 0x044600ce, 0x04ad00ce, 0x051400ce, 0x057b00c5, 0x05dd00bc, 0x063b00b5, 0x06970158, 0x07420142, 0x07e30130, 0x087b0120,
        0x090b0112, 0x09940106, 0x0a1700fc, 0x0a9500f2, 0x0b0f01cb, 0x0bf401ae, 0x0ccb0195, 0x0d950180, 0x0e56016e, 0x0f0d015e, 0x0fbc0150,
        0x10630143, 0x11070264, 0x1238023e, 0x1357021d, 0x14660201, 0x156601e9, 0x165a01d3, 0x174401c0, 0x182401af, 0x18fe0331, 0x1a9602fe,
        0x1c1502d2, 0x1d7e02ad, 0x1ed4028d, 0x201a0270, 0x21520256, 0x227d0240, 0x239f0443, 0x25c003fe, 0x27bf03c4, 0x29a10392, 0x2b6a0367,
        0x2d1d0341, 0x2ebe031f, 0x304d0300, 0x31d105b0, 0x34a80555, 0x37520507, 0x39d504c5, 0x3c37048b, 0x3e7c0458, 0x40a8042a, 0x42bd0401,
        0x44c20798, 0x488e071e, 0x4c1c06b6, 0x4f76065d, 0x52a50610, 0x55ac05cc, 0x5892058f, 0x5b590559, 0x5e0c0a23, 0x631c0980, 0x67db08f6,
        0x6c55087f, 0x70940818, 0x74a007bd, 0x787d076c, 0x7c330723};
    const uint32_t min_u = (127u - 13u) << 23, almost_one_u = 0x3f7fffffu;
    float min_f, almost_one;
    memcpy(&min_f, &min_u, 4);
    memcpy(&almost_one, &almost_one_u, 4);
    if (!(in > min_f)) in = min_f;       // NaN -> 0 as well
    if (in > almost_one) in = almost_one;
    uint32_t u;
    memcpy(&u, &in, 4);
    const uint32_t t = tab[(u - min_u) >> 20];
    const uint32_t bias = (t >> 16) << 9, scale = t & 0xffff;
    return (uint8_t)((bias + scale * ((u >> 12) & 0xff)) >> 16);
}

int alpha_channel(image_format f) { // reference src/visp/image.cpp:57-67
    switch (f) {
        case image_format::bgra_u8: return 3;
        case image_format::argb_u8: return 0;
        case image_format::alpha_u8:
        case image_format::alpha_f32: return 0;
        case image_format::rgb_u8:
        case image_format::rgb_f32: return -1;
        default: return 3;
    }
}

} // namespace

image_data image_scale(image_view const& img, i32x2 target) {
    const int ch = n_channels(img.format), ac = alpha_channel(img.format);
    const bool fl = is_float(img.format);
    const int iw = img.extent[0], ih = img.extent[1], ow = target[0], oh = target[1];
    if (iw <= 0 || ih <= 0 || ow <= 0 || oh <= 0) throw except("Failed to resize image %dx%d to %dx%d", iw, ih, ow, oh);
    image_data dst = image_alloc(target, img.format);
    const std::vector<taps> tx = axis_taps(iw, ow), ty = axis_taps(ih, oh);
    const float* lut = srgb_to_linear_table();
    const float alpha_eps = std::ldexp(1.0f, -80);

    // decode + horizontal pass: rows of the source at the target width, linear light, premultiplied
    std::vector<float> row((size_t)iw * ch), mid((size_t)ih * ow * ch);
    for (int y = 0; y < ih; ++y) {
        const uint8_t* src = static_cast<const uint8_t*>(img.data) + (size_t)y * img.stride;
        for (int x = 0; x < iw; ++x) {
            float* p = row.data() + (size_t)x * ch;
            for (int c = 0; c < ch; ++c) {
                if (fl) p[c] = reinterpret_cast<const float*>(src)[x * ch + c];
                else p[c] = c == ac ? (float)src[x * ch + c] / 255.0f : lut[src[x * ch + c]];
            }
            if (ac >= 0) {
                float a = p[ac];
                if (!fl) { a += alpha_eps; p[ac] = a; }
                for (int c = 0; c < ch; ++c)
                    if (c != ac) p[c] *= a;
            }
        }
        float* m = mid.data() + (size_t)y * ow * ch;
        for (int x = 0; x < ow; ++x) {
            const taps& t = tx[(size_t)x];
            for (int c = 0; c < ch; ++c) {
                float acc = 0;
                for (size_t i = 0; i < t.w.size(); ++i) acc += row[(size_t)std::clamp(t.first + (int)i, 0, iw - 1) * ch + c] * t.w[i];
                m[(size_t)x * ch + c] = acc;
            }
        }
    }
    // vertical pass + encode
    std::vector<float> out((size_t)ow * ch);
    for (int y = 0; y < oh; ++y) {
        const taps& t = ty[(size_t)y];
        std::fill(out.begin(), out.end(), 0.0f);
        for (size_t i = 0; i < t.w.size(); ++i) {
            const float* m = mid.data() + (size_t)std::clamp(t.first + (int)i, 0, ih - 1) * ow * ch;
            const float w = t.w[i];
            for (size_t j = 0; j < out.size(); ++j) out[j] += m[j] * w;
        }
        uint8_t* drow = dst.data.get() + (size_t)y * ow * n_bytes(img.format);
        for (int x = 0; x < ow; ++x) {
            float* p = out.data() + (size_t)x * ch;
            if (ac >= 0) {
                const float a = p[ac], ra = a != 0.0f ? 1.0f / a : 0.0f;
                for (int c = 0; c < ch; ++c)
                    if (c != ac) p[c] *= ra;
            }
            for (int c = 0; c < ch; ++c) {
                if (fl) reinterpret_cast<float*>(drow)[x * ch + c] = p[c];
                else if (c == ac) drow[x * ch + c] = (uint8_t)(int)(std::min(std::max(p[c], 0.0f), 1.0f) * 255.0f + 0.5f);
                else drow[x * ch + c] = linear_to_srgb_u8(p[c]);
            }
        }
    }
    return dst;
}

} // namespace visp
