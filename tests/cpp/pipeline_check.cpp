// The Depth-Anything pipeline pieces of the public C++ header (include/visp/vision.h; reference include/visp/vision.h:236-252,
// src/visp/arch/depth-anything.cpp:112-149): parameters of a loaded model, the model extent of a caller's extent, input
// pre-processing (image_scale + ImageNet normalisation) and output post-processing (min-max normalise + image_scale back).
// Exit code 0 = ok. Built by __graft_entry__.build(); run by tests/test_gpu_model.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <vector>

#include <visp/vision.h>

using namespace visp;

#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::fprintf(stderr, "pipeline-check: %s failed (line %d)\n", #cond, __LINE__); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    try {
        backend_device backend = backend_init();
        depthany_model model = depthany_load_model(argv[1], backend);

        depthany_params p = depthany_detect_params(model, {640, 480});
        EXPECT(p.image_size == 518 && p.image_multiple == 14 && p.dino.patch_size == 14 && p.dino.embed_dim == 384);
        EXPECT((p.image_extent == i32x2{700, 518}));
        EXPECT((depthany_image_extent({518, 518}, p) == i32x2{518, 518}));
        EXPECT((depthany_image_extent({1000, 2000}, p) == i32x2{1008, 2016}));       // short side above image_size: rounded up to 14

        image_data img = image_alloc({640, 480}, image_format::rgb_u8);
        for (int i = 0; i < 640 * 480 * 3; ++i) img.data[size_t(i)] = uint8_t(128);   // mid grey
        image_data in = depthany_process_input(img, p);
        EXPECT((in.extent == p.image_extent) && in.format == image_format::rgb_f32);
        float const* f = reinterpret_cast<float const*>(in.data.get());
        EXPECT(std::fabs(f[0] - (128.f / 255.f - 0.485f) / 0.229f) < 2e-3f);          // (a constant image stays constant under image_scale)
        EXPECT(std::fabs(f[2] - (128.f / 255.f - 0.406f) / 0.225f) < 2e-3f);

        std::vector<float> raw(size_t(700) * 518);
        for (size_t i = 0; i < raw.size(); ++i) raw[i] = 3.f + 0.001f * float(i % 700);
        image_data out = depthany_process_output(raw, {640, 480}, p);
        EXPECT((out.extent == i32x2{640, 480}) && out.format == image_format::alpha_f32);
        float lo = 1e9f, hi = -1e9f;
        for (float v : image_view(out).as_floats()) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        EXPECT(lo >= -1e-3f && lo < 0.01f && hi > 0.99f && hi <= 1.001f);
        image_data same = depthany_process_output(raw, {700, 518}, p);               // no resize: exactly [0, 1]
        EXPECT(image_view(same).as_floats()[0] == 0.f);

        // the pieces compose to depthany_compute's extents
        image_data depth = depthany_compute(model, img);
        EXPECT((depth.extent == i32x2{640, 480}) && depth.format == image_format::alpha_f32);
        std::printf("pipeline-check ok\n");
        return 0;
    } catch (std::exception const& ex) {
        std::fprintf(stderr, "pipeline-check failed: %s\n", ex.what());
        return 1;
    }
}
