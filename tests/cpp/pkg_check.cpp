// Package smoke test of the public C++ header (include/visp/vision.h) as an installed-package user would write it: find a GPU, refuse the CPU
// backend, load a Depth-Anything GGUF, run images of two extents and look at the results. Exit code 0 = every check held.
// Built by __graft_entry__.build(), run by tests/test_gpu_model.py on the GPU box:   pkg_check <depth-anything.gguf>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include <visp/vision.h>

namespace {

int failures = 0;
void expect(bool ok, char const* what) {
    if (!ok) {
        std::fprintf(stderr, "pkg_check: FAILED: %s\n", what);
        ++failures;
    }
}

visp::image_data gradient_image(int w, int h) { // a smooth, non-constant rgb_u8 picture
    visp::image_data img = visp::image_alloc({w, h}, visp::image_format::rgb_u8);
    uint8_t* px = img.data.get();
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x, px += 3) {
            px[0] = uint8_t(255 * x / std::max(1, w - 1));
            px[1] = uint8_t(255 * y / std::max(1, h - 1));
            px[2] = uint8_t((x ^ y) & 0xff);
        }
    return img;
}

struct depth_stats { float lo = 1e30f, hi = -1e30f; bool finite = true; };
depth_stats stats_of(visp::image_data const& depth) {
    depth_stats s;
    for (float v : visp::image_view(depth).as_floats()) {
        s.finite = s.finite && v == v && v - v == 0.f;
        s.lo = std::min(s.lo, v);
        s.hi = std::max(s.hi, v);
    }
    return s;
}

} // namespace

int main(int argc, char** argv) {
    if (argc != 2) {
        std::fprintf(stderr, "usage: pkg_check <depth-anything.gguf>\n");
        return 2;
    }
    try {
        bool cpu_refused = false;
        try {
            visp::backend_device cpu = visp::backend_init(visp::backend_type::cpu);
        } catch (visp::exception const&) {
            cpu_refused = true; // this build has no CPU backend and says so instead of falling back
        }
        expect(cpu_refused, "backend_init(cpu) must throw");
        expect(visp::backend_is_available(visp::backend_type::gpu), "gpu backend available");

        visp::backend_device gpu = visp::backend_init(visp::backend_type::gpu);
        expect(gpu.type() == visp::backend_type::gpu, "device type");
        visp::depthany_model model = visp::depthany_load_model(argv[1], gpu);

        const int extents[2][2] = {{96, 64}, {70, 126}};
        for (auto const& e : extents) {
            visp::image_data image = gradient_image(e[0], e[1]);
            visp::image_data depth = visp::depthany_compute(model, image);
            expect(depth.extent == image.extent, "depth has the input's extent");
            expect(depth.format == visp::image_format::alpha_f32, "depth is alpha_f32");
            depth_stats s = stats_of(depth);
            expect(s.finite, "depth is finite");
            // normalised to [0, 1] at the model's extent, then scaled back to the caller's (a cubic filter: small over- / undershoot)
            expect(s.lo > -0.1f && s.lo < 0.2f && s.hi > 0.8f && s.hi < 1.1f, "depth spans the normalised range");
            visp::image_data again = visp::depthany_compute(model, image); // same model, same image: same bits
            expect(std::memcmp(depth.data.get(), again.data.get(), size_t(e[0]) * size_t(e[1]) * 4) == 0, "repeatable result");
        }

        visp::image_data flat = visp::image_alloc({48, 48}, visp::image_format::rgb_u8);
        visp::image_clear(flat);
        expect(stats_of(visp::depthany_compute(model, flat)).finite, "a constant image gives a finite map");
        visp::image_data small = visp::image_scale(gradient_image(64, 32), {16, 8});
        expect(small.extent == visp::i32x2(16, 8) && small.format == visp::image_format::rgb_u8, "image_scale extent / format");

        if (failures == 0) std::printf("pkg_check ok on %s\n", gpu.description());
        return failures == 0 ? 0 : 1;
    } catch (std::exception const& e) {
        std::fprintf(stderr, "pkg_check: exception: %s\n", e.what());
        return 1;
    }
}
