// Driver of tests/test_reference_sources_compile.py. It is linked with objects compiled from the REFERENCE's own, unmodified
// src/visp/arch/dino.cpp and depth-anything.cpp (read where they lie under /root/reference, objects in a temporary directory; nothing of
// them is stored in this repository): every visp:: builder called below -- depthany_detect_params(model_file), depthany_image_extent,
// depthany_predict, depthany_process_input / _output -- is the reference's definition running on this backend's ml.h / nn.h / C ABI.
// No GPU: the graph is lowered and planned only; the image steps are host code in both implementations.
//   arch_source_driver <depth-anything.gguf> <width> <height>
// prints: the detected parameters, the output shape, check sums of process_input / process_output, then the launch list.
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "visp/arch/depth-anything.h"

using namespace visp;

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    try {
        const i32x2 asked(std::atoi(argv[2]), std::atoi(argv[3]));
        model_file file = model_load(argv[1]);
        depthany_params p = depthany_detect_params(file, asked);
        std::printf("arch=%s tensors=%lld\n", file.arch().c_str(), (long long)file.n_tensors());
        std::printf("params patch=%d dim=%d layers=%d heads=%d size=%d taps=%d,%d,%d,%d extent=%dx%d\n", p.dino.patch_size, p.dino.embed_dim, p.dino.n_layers,
                    p.dino.n_heads, p.image_size, p.feature_layers[0], p.feature_layers[1], p.feature_layers[2], p.feature_layers[3], p.image_extent[0], p.image_extent[1]);

        model_weights weights = model_init(size_t(file.n_tensors()));
        model_transfer(file, weights);
        compute_graph graph = compute_graph_init(weights);
        model_ref m(graph);
        tensor input = compute_graph_input(m, GGML_TYPE_F32, {3, p.image_extent[0], p.image_extent[1], 1});
        tensor depth = depthany_predict(m, input, p);
        std::printf("output ne=%lld,%lld,%lld,%lld\n", (long long)depth->ne[0], (long long)depth->ne[1], (long long)depth->ne[2], (long long)depth->ne[3]);
        for (int layer : p.feature_layers) { // the taps are kept under the names the reference gives them
            char name[32];
            std::snprintf(name, sizeof name, "dino_layer_%d", layer);
            tensor tap = get_tensor(m, name);
            std::printf("tap %s ne=%lld,%lld,%lld\n", name, tap ? (long long)tap->ne[0] : -1LL, tap ? (long long)tap->ne[1] : -1LL, tap ? (long long)tap->ne[2] : -1LL);
        }

        // host steps around the graph, on an image of the asked extent
        image_data image = image_alloc(asked, image_format::rgb_u8);
        for (int i = 0; i < asked[0] * asked[1] * 3; ++i) image.data[size_t(i)] = uint8_t((i * 37 + (i >> 5) * 11) & 255);
        image_data in = depthany_process_input(image, p);
        std::span<float const> fin = image_view(in).as_floats();
        std::printf("process_input %dx%d sum=%.6f\n", in.extent[0], in.extent[1], std::accumulate(fin.begin(), fin.end(), 0.0));
        std::vector<float> raw(size_t(p.image_extent[0]) * size_t(p.image_extent[1]));
        for (size_t i = 0; i < raw.size(); ++i) raw[i] = float((i * 13) % 1009) * 0.25f + 3.0f;
        image_data out = depthany_process_output(raw, asked, p);
        std::span<float const> fout = image_view(out).as_floats();
        std::printf("process_output %dx%d sum=%.6f\n", out.extent[0], out.extent[1], std::accumulate(fout.begin(), fout.end(), 0.0));

        compute_graph_plan(graph);
        std::printf("%s\n", compute_graph_describe(graph).c_str());
        return 0;
    } catch (std::exception const& e) {
        std::fprintf(stderr, "arch_source_driver: %s\n", e.what());
        return 1;
    }
}
