// ESRGAN / Real-ESRGAN (RRDBNet) on the MI355X backend: model load (GGUF -> weight store), the generator as a graph of the
// reference's nodes that csrc/graph.cpp lowers onto the planar dense-block conv schedule, and the tiled, batched executor around it.
// Host C++ only; all device work goes through the vx_* C ABI (include/visp_hip_kernels.h).
//
// Mirrors the reference's API for this family (include/visp/vision.h:284-304, 361-369;
// src/visp/vision.cpp:208-253; src/visp/arch/esrgan.cpp; tiling: src/visp/image.cpp:612-693).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "depthany.h"

namespace visp {

struct esrgan_params { // vision.h:296-299
    int scale = 4;
    int n_blocks = 23;
};
esrgan_params esrgan_detect_params(model_file const&); // esrgan.cpp:81-97

// tile_layout (image.h:163-181, image.cpp:612-651)
struct tile_layout {
    i32x2 image_extent{}, overlap{}, n_tiles{}, tile_size{};
    tile_layout() = default;
    tile_layout(i32x2 extent, int max_tile_size, int overlap, int align = 16);
    int total() const { return n_tiles[0] * n_tiles[1]; }
    i32x2 coord(int index) const { return {{index % n_tiles[0], index / n_tiles[0]}}; }
    i32x2 start(i32x2 coord) const { return {{coord[0] * (tile_size[0] - overlap[0]), coord[1] * (tile_size[1] - overlap[1])}}; }
};
tile_layout tile_scale(tile_layout const&, int scale);

struct graph;
struct weight_store;

struct esrgan_workspace {
    int group = 0, tile_w = 0, tile_h = 0, scale = 0; // sized for `group` tiles of this extent
    size_t img_in = 0, img_out = 0;                   // bytes reserved for the u8 in/out images
    device_buffer arena;
    void *in_u8 = nullptr, *out_u8 = nullptr, *x0 = nullptr, *tiles_out = nullptr; // whole call: u8 images, f32 rgb tiles in and out
    int lanes = 1;
};

// the generator's graph for n tiles of one extent (csrc/graph.h), lowered and allocated; input / output bound to slices of the call's workspace
struct esrgan_step {
    int n = 0, w = 0, h = 0;
    std::unique_ptr<graph> g;
    int in = -1, out = -1;
    esrgan_step();
    ~esrgan_step();
};

struct esrgan_model : model_base { // vision.h:361-369 counterpart
    esrgan_model() : model_base(family_esrgan) {}
    backend_device const* backend = nullptr;
    esrgan_params params;
    int nf = 64, gc = 32;                    // filters / growth channels of the file
    std::shared_ptr<weight_store> store;     // the file's tensors + their packed device images (one arena: what a load-time broadcast moves)
    device_buffer weight_arena;              // = the store's arena
    std::vector<std::unique_ptr<esrgan_step>> steps[2]; // per lane: graphs of the most recent tile-group shapes
    bool weights_uploaded = false;
    esrgan_workspace ws;
    int tile_group = 64;  // upper bound of tiles pushed through the network together (bounds the workspace)
    // Tile groups are independent, so two of them run concurrently on two streams: a conv launch is 10.25 rounds of
    // tiles on 256 CUs, and the second group's blocks take the CUs the first one's last round leaves idle
    // (measured +6 %). 1 = everything on the caller's stream.
    int streams = 2;
    void* aux_stream = nullptr;
    void *fork_event = nullptr, *join_event = nullptr;
    bool timing = false;
    std::vector<timing_entry> last_timing;
    ~esrgan_model();
};

esrgan_model* esrgan_load_model(char const* filepath, backend_device const& dev, int flags = load_default);
void esrgan_weights_ready(esrgan_model&);

// B images of one extent, any u8 colour format, already on the device -> rgba_u8 [B, h*scale, w*scale, 4] on the
// device. Tiling exactly as esrgan_compute (224 max, overlap 16, align 16); all tiles of all images are batched.
void esrgan_compute_batch_device(esrgan_model&, void const* img_dev, int batch, int w, int h, image_format format, void* out_rgba_dev,
                                 void* stream);
void esrgan_compute_batch_host(esrgan_model&, uint8_t const* img, int batch, int w, int h, image_format format, uint8_t* out_rgba);
// reference API (vision.cpp:220-253): one image view -> rgba_u8 image_data at extent*scale
image_data esrgan_compute(esrgan_model&, image_view image);
// the network on raw rgb f32 tiles [n, h, w, 3] (host) -> [n, h*scale, w*scale, 3] f32: what esrgan_generate
// (esrgan.cpp:55-79) computes per tile; used by the parity tests
void esrgan_generate_host(esrgan_model&, float const* rgb, int n, int w, int h, float* out);

} // namespace visp
