// A caller of the graph layer shaped like the reference's own depthany_compute (src/visp/vision.cpp:137-167): load the weights, build the
// graph once with depthany_predict on a model_ref, allocate, then per image process_input -> transfer_to_backend -> compute ->
// transfer_from_backend -> process_output. The result is checked against the hand-scheduled depthany_compute of the same library on
// the same file and image. Built by __graft_entry__.build(), run by tests/test_gpu_graph.py on the GPU box.
//   graph_check <model.gguf> <width> <height>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define VISP_GGML_NAMES // the ggml_* spellings of visp/nn.h compile too
#include "visp/arch/depth-anything.h"

using namespace visp;

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: graph_check <depth-anything.gguf> <width> <height>\n");
        return 2;
    }
    try {
        const i32x2 extent(std::atoi(argv[2]), std::atoi(argv[3]));
        backend_device dev = backend_init();
        depthany_model fast = depthany_load_model(argv[1], dev);
        depthany_params p = depthany_detect_params(fast, extent);

        model_weights weights = model_load_weights(argv[1]);
        compute_graph graph = compute_graph_init(weights);
        model_ref m(graph);
        tensor input = compute_graph_input(m, GGML_TYPE_F32, {3, p.image_extent[0], p.image_extent[1], 1});
        tensor output = depthany_predict(m, input, p);
        compute_graph_allocate(graph, dev);

        image_data image = image_alloc(extent, image_format::rgb_u8);
        for (int y = 0; y < extent[1]; ++y)
            for (int x = 0; x < extent[0]; ++x)
                for (int c = 0; c < 3; ++c) image.data[size_t(y * extent[0] + x) * 3 + size_t(c)] = uint8_t((x * (3 + c) + y * (5 - c) + (x * y) / 7) & 255);

        image_data img_data = depthany_process_input(image, p);
        transfer_to_backend(input, img_data);
        compute(graph, dev);
        tensor_data out = transfer_from_backend(output);
        image_data depth = depthany_process_output(out.as_f32(), extent, p);

        image_data want = depthany_compute(fast, image);
        if (depth.extent != want.extent || depth.format != image_format::alpha_f32) {
            std::fprintf(stderr, "graph_check: extent / format mismatch\n");
            return 1;
        }
        const size_t n = size_t(extent[0]) * size_t(extent[1]);
        const float* a = reinterpret_cast<float const*>(depth.data.get());
        const float* b = reinterpret_cast<float const*>(want.data.get());
        double sum = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!std::isfinite(a[i])) { std::fprintf(stderr, "graph_check: non-finite depth\n"); return 1; }
            sum += std::fabs(double(a[i]) - double(b[i]));
        }
        std::string text = compute_graph_describe(graph);
        std::printf("graph_check ok: %dx%d (model extent %dx%d), mean |graph - depthany_compute| = %.3e, %s", extent[0], extent[1], p.image_extent[0], p.image_extent[1],
                    sum / double(n), text.substr(text.rfind("launches=")).c_str());
        return sum / double(n) < 1e-3 ? 0 : 1;
    } catch (std::exception const& e) {
        std::fprintf(stderr, "graph_check: %s\n", e.what());
        return 1;
    }
}
