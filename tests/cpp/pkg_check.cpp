// Installed-package smoke test against the public C++ header, the shape of the reference's scripts/pkg-check/main.cpp:22-44
// (backend_init, depthany_load_model, image_alloc, image_clear, depthany_compute, extent + finite-mean checks), with the GPU
// backend this build has. Exit code 0 = ok. Built by __graft_entry__.build(); run by tests/test_gpu_model.py on the GPU box.
#include <cmath>
#include <iostream>
#include <numeric>
#include <string>

#include <visp/vision.h>

using namespace visp;

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <model-path>\n";
        return 2;
    }
    std::string const model_path = argv[1];
    try {
        try { // this backend has no CPU device: the reference's message, not a fallback
            backend_device cpu = backend_init(backend_type::cpu);
            std::cerr << "a CPU backend must not exist in this build\n";
            return 1;
        } catch (visp::exception const&) {
        }
        backend_device backend = backend_init(backend_type::gpu);
        depthany_model model = depthany_load_model(model_path.c_str(), backend);

        image_data input = image_alloc({64, 64}, image_format::rgb_u8);
        image_clear(input);

        image_data output = depthany_compute(model, input);
        if (output.extent != input.extent || output.format != image_format::alpha_f32) {
            std::cerr << "Unexpected output extent: " << output.extent[0] << "x" << output.extent[1] << "\n";
            return 1;
        }
        std::span<float const> depth = image_view{output}.as_floats();
        double const mean = std::accumulate(depth.begin(), depth.end(), 0.0) / double(depth.size());
        if (!std::isfinite(mean)) {
            std::cerr << "Depth output mean is not finite\n";
            return 1;
        }
        image_data half = image_scale(input, {32, 32});
        if (half.extent != i32x2{32, 32}) return 1;
        std::cout << "pkg-check ok: " << backend.description() << ", mean depth " << mean << "\n";
        return 0;
    } catch (std::exception const& ex) {
        std::cerr << "pkg-check failed: " << ex.what() << "\n";
        return 1;
    }
}
