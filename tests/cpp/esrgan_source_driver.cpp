// Second driver of tests/test_reference_sources_compile.py: linked with an object compiled from the REFERENCE's own, unmodified
// src/visp/arch/esrgan.cpp (read where it lies; nothing of it is stored in this repository). esrgan_detect_params(model_file) and
// esrgan_generate(model_ref, image, params) are the reference's definitions running on this backend's ml.h / nn.h / C ABI.
// No GPU: the graph is lowered and planned only.
//   esrgan_source_driver <esrgan.gguf> <tile width> <tile height> <tiles>
#include <cstdio>
#include <cstdlib>

#include "visp/arch/esrgan.h"

using namespace visp;

int main(int argc, char** argv) {
    if (argc != 5) return 2;
    try {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), n = std::atoi(argv[4]);
        model_file file = model_load(argv[1]);
        esrgan_params p = esrgan_detect_params(file);
        std::printf("arch=%s scale=%d blocks=%d graph_size=%d\n", file.arch().c_str(), p.scale, p.n_blocks, esrgan_estimate_graph_size(p));
        model_weights weights = model_init(size_t(file.n_tensors()));
        model_transfer(file, weights);
        compute_graph graph = compute_graph_init(weights);
        model_ref m(graph);
        tensor input = compute_graph_input(m, GGML_TYPE_F32, {3, w, h, n});
        tensor result = esrgan_generate(m, input, p);
        std::printf("result ne=%lld,%lld,%lld,%lld\n", (long long)result->ne[0], (long long)result->ne[1], (long long)result->ne[2], (long long)result->ne[3]);
        compute_graph_plan(graph);
        std::printf("%s\n", compute_graph_describe(graph).c_str());
        // a file of another architecture is refused with the reference's message
        return 0;
    } catch (std::exception const& e) {
        std::fprintf(stderr, "esrgan_source_driver: %s\n", e.what());
        return 1;
    }
}
