// The drop-in C ABI (include/visp_c_api.h). Part 1 restates the reference's
// src/visp/c-api.cpp:145-253 symbol for symbol on top of this backend; part 2 is the batched
// extension. Error convention: return 1 on success, 0 on error + thread-local message
// (reference c-api.cpp:6-21).
#include "../../include/visp_c_api.h"

#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/visp_hip_kernels.h"
#include "depthany.h"
#include "esrgan.h"
#include "tinyvit.h"
#include "swin.h"
#include "birefnet.h"
#include "graph.h"
#include "visp_util.h"

using namespace visp;

struct visp_image_data : image_data {};
struct visp_device : backend_device {};
struct visp_graph : graph {};
struct visp_weights { std::shared_ptr<weight_store> store; };
struct visp_file { model_file file; };
// visp_model stays opaque: handles are depthany_model* (any_model in the reference, c-api.cpp:193)

namespace {

thread_local char g_error[512] = {0};

void set_error(const char* msg) { snprintf(g_error, sizeof g_error, "%s", msg); }

template <typename F>
int32_t handle_errors(F&& f) {
    try {
        f();
    } catch (std::exception const& e) {
        set_error(e.what());
        return 0;
    } catch (...) {
        set_error("unknown error");
        return 0;
    }
    return 1;
}

int32_t family_of(visp_model const* m) {
    if (!m) throw except("model handle is null");
    return reinterpret_cast<model_base const*>(m)->family;
}
depthany_model& as_depthany(visp_model* m) {
    if (family_of(m) != VISP_DEPTH_ANYTHING) throw except("model handle is not a depth_anything model (family %d)", family_of(m));
    return *static_cast<depthany_model*>(reinterpret_cast<model_base*>(m));
}
depthany_model const& as_depthany(visp_model const* m) {
    if (family_of(m) != VISP_DEPTH_ANYTHING) throw except("model handle is not a depth_anything model (family %d)", family_of(m));
    return *static_cast<depthany_model const*>(reinterpret_cast<model_base const*>(m));
}
esrgan_model& as_esrgan(visp_model* m) {
    if (family_of(m) != VISP_ESRGAN) throw except("model handle is not an esrgan model (family %d)", family_of(m));
    return *static_cast<esrgan_model*>(reinterpret_cast<model_base*>(m));
}
sam_model& as_sam(visp_model* m) {
    if (family_of(m) != VISP_SAM) throw except("model handle is not a sam model (family %d)", family_of(m));
    return *static_cast<sam_model*>(reinterpret_cast<model_base*>(m));
}
swin_model& as_swin(visp_model* m) { // the encoder half of the birefnet family (visp_swin_load)
    if (family_of(m) != VISP_BIREFNET) throw except("model handle is not a swin encoder (family %d)", family_of(m));
    return *static_cast<swin_model*>(reinterpret_cast<model_base*>(m));
}
birefnet_model& as_birefnet(visp_model* m) {
    swin_model& s = as_swin(m);
    if (!s.full) throw except("model handle is a swin encoder (visp_swin_load), not a birefnet model");
    return static_cast<birefnet_model&>(s);
}
visp_model* handle_of(model_base* m) { return reinterpret_cast<visp_model*>(m); }

int32_t detect_family(model_file const& file) { // reference vision.cpp:7-21
    std::string_view arch = file.arch();
    if (arch == "mobile-sam") return VISP_SAM;
    if (arch == "birefnet") return VISP_BIREFNET;
    if (arch == "depthanything") return VISP_DEPTH_ANYTHING;
    if (arch == "migan") return VISP_MIGAN;
    if (arch == "esrgan") return VISP_ESRGAN;
    return VISP_FAMILY_COUNT;
}

void require_built(int32_t family) {
    if (family < 0 || family >= VISP_FAMILY_COUNT) throw except("Unsupported model family");
    if (family != VISP_DEPTH_ANYTHING && family != VISP_ESRGAN && family != VISP_SAM && family != VISP_BIREFNET)
        throw except("Model family %d is not built in this backend (MI355X backend implements depth_anything, esrgan, sam and birefnet)", family);
}

void read_capture(backend_device const& dev, std::map<std::string, capture_entry> const& bufs, char const* name, float* host_out,
                  int64_t capacity, int64_t* n_written, int64_t shape[4]) {
    auto it = bufs.find(name);
    if (it == bufs.end()) throw except("capture '%s' not available (enable captures before compute)", name);
    capture_entry const& c = it->second;
    int64_t n = c.shape[0] * c.shape[1] * c.shape[2] * c.shape[3];
    *n_written = n;
    if (shape) memcpy(shape, c.shape, sizeof(int64_t) * 4);
    if (n > capacity) return; // caller learns the size and retries
    if (!vx_set_device(dev.index)) throw except("%s", vx_last_error());
    if (c.f16) {
        std::unique_ptr<uint16_t[]> tmp(new uint16_t[(size_t)n]);
        if (!vx_memcpy_d2h(tmp.get(), c.dev, (size_t)n * 2, dev.stream)) throw except("%s", vx_last_error());
        for (int64_t i = 0; i < n; ++i) host_out[i] = f16_to_f32(tmp[(size_t)i]);
    } else {
        if (!vx_memcpy_d2h(host_out, c.dev, (size_t)n * 4, dev.stream)) throw except("%s", vx_last_error());
    }
}

void return_image(image_data&& img, visp_image_view* out_image, visp_image_data** out_data) {
    auto* owned = new visp_image_data;
    static_cast<image_data&>(*owned) = std::move(img);
    *out_data = owned;
    out_image->width = owned->extent[0];
    out_image->height = owned->extent[1];
    out_image->stride = owned->extent[0] * n_bytes(owned->format);
    out_image->format = int32_t(owned->format);
    out_image->data = owned->data.get();
}

} // namespace

extern "C" {

char const* visp_get_last_error(void) { return g_error; }

void visp_image_destroy(visp_image_data* img) { delete img; }

// host-only: the reference's image_scale (src/visp/image.cpp:328-356, what depthany_compute / sam_encode apply to inputs whose
// extent is not the model's); no device involved
int32_t visp_image_scale(visp_image_view const* src, int32_t width, int32_t height, visp_image_view* out_image, visp_image_data** out_data) {
    return handle_errors([&]() {
        if (!src || !src->data || !out_image || !out_data) throw except("visp_image_scale: null argument");
        if (src->format < 0 || src->format > int32_t(image_format::alpha_f32)) throw except("Unsupported image format [%d]", src->format);
        image_view v{i32x2{{src->width, src->height}}, src->stride, image_format(src->format), src->data};
        return_image(image_scale(v, i32x2{{width, height}}), out_image, out_data);
    });
}

int32_t visp_image_u8_to_f32(visp_image_view const* src, int32_t format, float const offset[4], float const scale[4], visp_image_view* out_image,
                             visp_image_data** out_data) {
    return handle_errors([&]() {
        if (!src || !src->data || !offset || !scale || !out_image || !out_data) throw except("visp_image_u8_to_f32: null argument");
        if (src->format < 0 || src->format > int32_t(image_format::alpha_f32) || format < 0 || format > int32_t(image_format::alpha_f32))
            throw except("Unsupported image format [%d]", src->format);
        image_view v{i32x2{{src->width, src->height}}, src->stride, image_format(src->format), src->data};
        return_image(image_u8_to_f32(v, image_format(format), offset, scale), out_image, out_data);
    });
}

int32_t visp_image_normalize(visp_image_view const* src, float min, float max, visp_image_view* out_image, visp_image_data** out_data) {
    return handle_errors([&]() {
        if (!src || !src->data || !out_image || !out_data) throw except("visp_image_normalize: null argument");
        if (src->format != int32_t(image_format::alpha_f32)) throw except("visp_image_normalize: expected an alpha_f32 image, got format %d", src->format);
        image_view v{i32x2{{src->width, src->height}}, src->stride, image_format(src->format), src->data};
        return_image(image_normalize(v, min, max), out_image, out_data);
    });
}

int32_t visp_backend_load_all(char const*) { return 1; } // single built-in backend, nothing to load

int32_t visp_device_init(int32_t type, visp_device** out_device) {
    return handle_errors([&]() {
        if (type == VISP_BACKEND_CPU)
            throw except("Failed to initialize backend, no suitable device available (this build has no CPU backend)");
        if (type != VISP_BACKEND_AUTO && type != VISP_BACKEND_GPU)
            throw except("Failed to initialize backend, backend type %d is not available in this build", type);
        *out_device = static_cast<visp_device*>(backend_init(0));
    });
}

int32_t visp_hip_device_init(int32_t device_index, visp_device** out_device) {
    return handle_errors([&]() { *out_device = static_cast<visp_device*>(backend_init(device_index)); });
}

void visp_device_destroy(visp_device* d) { delete static_cast<backend_device*>(d); }

int32_t visp_device_type(visp_device const* d) { return int32_t(d->type()); }
char const* visp_device_name(visp_device const* d) { return d->name.c_str(); }
char const* visp_device_description(visp_device const* d) { return d->description.c_str(); }

int32_t visp_model_detect_family(char const* filepath, int32_t* out_family) {
    return handle_errors([&]() {
        model_file file = model_load(filepath, /*header_only=*/true);
        *out_family = detect_family(file);
    });
}

// host-only: parses the whole file (header AND tensor data ranges) without touching a device; what visp_model_load checks first
int32_t visp_gguf_validate(char const* filepath, int32_t* out_n_tensors) {
    return handle_errors([&]() {
        model_file file = model_load(filepath, /*header_only=*/false);
        if (out_n_tensors) *out_n_tensors = (int32_t)file.tensors.size();
    });
}

int32_t visp_model_load_ex(char const* filepath, visp_device const* dev, int32_t arch, int32_t flags, visp_model** out) {
    return handle_errors([&]() {
        if (!dev) throw except("device handle is null");
        int32_t family = arch;
        if (family == VISP_FAMILY_COUNT) {
            model_file file = model_load(filepath, true);
            family = detect_family(file);
        }
        require_built(family);
        if (family == VISP_ESRGAN) *out = handle_of(esrgan_load_model(filepath, *dev, flags));
        else if (family == VISP_SAM) *out = handle_of(sam_load_model(filepath, *dev, flags));
        else if (family == VISP_BIREFNET) {
            if (flags & VISP_LOAD_NO_UPLOAD) throw except("birefnet: load without upload is not supported (every rank reads the file)");
            *out = handle_of(birefnet_load_model(filepath, *dev));
        }
        else *out = handle_of(depthany_load_model(filepath, *dev, flags));
    });
}

int32_t visp_model_load(char const* filepath, visp_device const* dev, int32_t arch, visp_model** out) {
    return visp_model_load_ex(filepath, dev, arch, VISP_LOAD_DEFAULT, out);
}

void visp_model_destroy(visp_model* model, int32_t arch) {
    if (!model) return;
    // the handle carries its own family; `arch` is what the reference's destroy switch uses (c-api.cpp:224-229)
    model_base* base = reinterpret_cast<model_base*>(model);
    (void)arch;
    if (base->family == VISP_DEPTH_ANYTHING) delete static_cast<depthany_model*>(base);
    else if (base->family == VISP_ESRGAN) delete static_cast<esrgan_model*>(base);
    else if (base->family == VISP_SAM) delete static_cast<sam_model*>(base);
    else if (base->family == VISP_BIREFNET) delete static_cast<swin_model*>(base); // swin encoder or full birefnet model (virtual destructor)
}

int32_t visp_model_compute(visp_model* model, int32_t family, visp_image_view* inputs, int32_t n_inputs, int32_t* args, int32_t n_args,
                           visp_image_view* out_image, visp_image_data** out_data) {
    return handle_errors([&]() {
        require_built(family);
        if (family_of(model) != family) throw except("model handle belongs to family %d, not %d", family_of(model), family);
        if (n_inputs != 1) throw except("Expected %d input images, but got %d.", 1, n_inputs);
        image_view in;
        in.extent = {{inputs[0].width, inputs[0].height}};
        in.stride = inputs[0].stride;
        in.format = image_format(inputs[0].format);
        in.data = inputs[0].data;
        if (family == VISP_SAM) { // model_funcs<sam>::compute = sam_encode + sam_compute (reference c-api.cpp:34-52)
            if (n_args != 2 && n_args != 4) throw except("sam: bad number of arguments (%d), must be 2 or 4", n_args);
            sam_model& sm = as_sam(model);
            sam_encode(sm, in);
            return_image(sam_compute(sm, args, n_args), out_image, out_data);
            return;
        }
        if (family == VISP_BIREFNET) { // model_funcs<birefnet>::compute (reference c-api.cpp:59-62)
            return_image(birefnet_compute(as_birefnet(model), in), out_image, out_data);
            return;
        }
        if (family == VISP_ESRGAN) { // model_funcs<esrgan>::compute (reference c-api.cpp:103-106)
            return_image(esrgan_compute(as_esrgan(model), in), out_image, out_data);
            return;
        }
        // model_funcs<depth_anything>::compute (reference c-api.cpp:72-77)
        image_data result_f32 = depthany_compute(as_depthany(model), in);
        image_data normalized = image_normalize(view_of(result_f32));
        return_image(image_f32_to_u8(view_of(normalized), image_format::alpha_u8), out_image, out_data);
    });
}

// ---- extension ---------------------------------------------------------------------------

int32_t visp_depthany_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        *device_ptr = dm.weight_arena.ptr;
        *n_bytes = dm.weight_arena.bytes;
    });
}

int32_t visp_depthany_weights_ready(visp_model* m) {
    return handle_errors([&]() { depthany_weights_ready(as_depthany(m)); });
}

int32_t visp_depthany_get_info(visp_model const* m, visp_depthany_info* out) {
    return handle_errors([&]() {
        depthany_params const& p = as_depthany(m).params;
        out->patch_size = p.dino.patch_size;
        out->embed_dim = p.dino.embed_dim;
        out->n_layers = p.dino.n_layers;
        out->n_heads = p.dino.n_heads;
        out->image_size = p.image_size;
        out->image_multiple = p.image_multiple;
        for (int i = 0; i < 4; ++i) out->feature_layers[i] = p.feature_layers[i];
        out->max_depth = p.max_depth;
    });
}

int32_t visp_depthany_image_extent(visp_model const* m, int32_t w, int32_t h, int32_t* out_w, int32_t* out_h) {
    return handle_errors([&]() {
        if (w <= 0 || h <= 0) throw except("invalid extent %dx%d", w, h);
        i32x2 e = depthany_image_extent(i32x2{{w, h}}, as_depthany(m).params);
        *out_w = e[0];
        *out_h = e[1];
    });
}

int32_t visp_depthany_reserve(visp_model* m, int32_t batch, int32_t w, int32_t h) {
    return handle_errors([&]() { depthany_reserve(as_depthany(m), batch, w, h); });
}

int32_t visp_depthany_compute_batch_device(visp_model* m, void const* rgb, int32_t batch, int32_t w, int32_t h, void* out,
                                           void* raw_out, void* stream) {
    return handle_errors([&]() {
        if (!rgb || !out) throw except("depthany: null input/output pointer");
        depthany_compute_batch_device(as_depthany(m), rgb, batch, w, h, out, raw_out, stream);
    });
}

int32_t visp_depthany_compute_f32(visp_model* m, visp_image_view const* image, visp_image_view* out_image, visp_image_data** out_data) {
    return handle_errors([&]() {
        if (!image || !image->data || !out_image || !out_data) throw except("depthany: null argument");
        image_view in{i32x2{{image->width, image->height}}, image->stride, image_format(image->format), image->data};
        return_image(depthany_compute(as_depthany(m), in), out_image, out_data);
    });
}

int32_t visp_depthany_compute_batch_host(visp_model* m, uint8_t const* rgb, int32_t batch, int32_t w, int32_t h, float* out,
                                         float* raw_out) {
    return handle_errors([&]() {
        if (!rgb || !out) throw except("depthany: null input/output pointer");
        depthany_compute_batch_host(as_depthany(m), rgb, batch, w, h, out, raw_out);
    });
}

// Multi-GPU without Python: one model per device (each loaded on its own visp_hip_device_init(i) device), one host thread per
// model, contiguous image shards whose sizes differ by at most one (the partitioning of vision.cpp_amd/dist.py::shard_range),
// no data-path collective -- images are independent units (image_normalize is per image, reference image.cpp:537-576). Each shard
// goes through its model's overlapped host pipeline in chunks of 32 (pinned staging, H2D / forward / D2H on three streams).
int32_t visp_depthany_compute_sharded(visp_model* const* models, int32_t n_models, uint8_t const* rgb, int32_t batch, int32_t w, int32_t h,
                                      float* out) {
    return handle_errors([&]() {
        if (!models || n_models <= 0 || !rgb || !out) throw except("depthany sharded: null / empty argument");
        if (batch <= 0 || w <= 0 || h <= 0) throw except("depthany sharded: invalid batch/extent %d x %dx%d", batch, w, h);
        std::vector<depthany_model*> ms;
        for (int i = 0; i < n_models; ++i) {
            ms.push_back(&as_depthany(models[i]));
            for (int j = 0; j < i; ++j)
                if (ms[(size_t)j] == ms[(size_t)i]) throw except("depthany sharded: model %d is passed twice (one model = one thread at a time)", i);
        }
        std::vector<std::string> errors((size_t)n_models);
        std::vector<std::thread> threads;
        // models that sit on the same backend_device share its compute stream (a one-GPU rehearsal of the multi-device call): every call
        // that touches that stream holds the device's turn (device_turn, csrc/depthany.h), so the shards interleave chunk by chunk and a
        // hipGraph capture by one thread is never crossed by another thread's launches
        const int base = batch / n_models, rem = batch % n_models;
        for (int i = 0; i < n_models; ++i) {
            const int begin = i * base + std::min(i, rem), count = base + (i < rem ? 1 : 0);
            if (count == 0) continue;
            threads.emplace_back([&, i, begin, count]() {
                try {
                    depthany_compute_shard_host(*ms[(size_t)i], rgb + (size_t)begin * h * w * 3, count, w, h, out + (size_t)begin * h * w);
                } catch (std::exception const& e) {
                    errors[(size_t)i] = e.what();
                } catch (...) {
                    errors[(size_t)i] = "unknown error";
                }
            });
        }
        for (auto& t : threads) t.join();
        for (int i = 0; i < n_models; ++i)
            if (!errors[(size_t)i].empty()) throw except("depthany sharded: shard %d failed: %s", i, errors[(size_t)i].c_str());
    });
}

int32_t visp_depthany_pipeline_create(visp_model* m, int32_t batch, int32_t w, int32_t h, int32_t n_slots, visp_depthany_pipeline** out) {
    return handle_errors([&]() { *out = reinterpret_cast<visp_depthany_pipeline*>(depthany_pipeline_create(as_depthany(m), batch, w, h, n_slots)); });
}
void visp_depthany_pipeline_destroy(visp_depthany_pipeline* p) { delete reinterpret_cast<depthany_pipeline*>(p); }
int32_t visp_depthany_pipeline_input(visp_depthany_pipeline* p, uint8_t** out_pinned) {
    return handle_errors([&]() {
        if (!p || !out_pinned) throw except("depthany pipeline: null argument");
        *out_pinned = depthany_pipeline_input(*reinterpret_cast<depthany_pipeline*>(p));
    });
}
int32_t visp_depthany_pipeline_submit(visp_depthany_pipeline* p, uint8_t const* rgb_u8, int32_t* out_ticket) {
    return handle_errors([&]() {
        if (!p || !out_ticket) throw except("depthany pipeline: null argument");
        *out_ticket = depthany_pipeline_submit(*reinterpret_cast<depthany_pipeline*>(p), rgb_u8);
    });
}
int32_t visp_depthany_pipeline_wait(visp_depthany_pipeline* p, int32_t ticket, float const** out_pinned) {
    return handle_errors([&]() {
        if (!p || !out_pinned) throw except("depthany pipeline: null argument");
        *out_pinned = depthany_pipeline_wait(*reinterpret_cast<depthany_pipeline*>(p), ticket);
    });
}

int32_t visp_depthany_use_graph(visp_model* m, int32_t enable) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        dm.use_graph = enable != 0;
        if (!enable) depthany_drop_captured_steps(dm);
    });
}

int32_t visp_depthany_set_schedule(visp_model* m, int32_t schedule) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        if (schedule < -1 || schedule > 1) throw except("visp_depthany_set_schedule: unknown schedule %d (-1 = auto, 0 = GEMM launches, 1 = token-stationary block kernel)", schedule);
        if (schedule == 1 && !dm.block_shape) throw except("visp_depthany_set_schedule: the block kernel is built for embed dim 384 / mlp 1536 / head dim 64 only");
        dm.schedule = schedule;
        depthany_drop_captured_steps(dm); // (steps are keyed by the schedule: the other one is lowered on first use)
    });
}

int32_t visp_depthany_set_split(visp_model* m, int32_t n) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        if (n < 0 || n > 4) throw except("visp_depthany_set_split: %d sub-batches (0 = automatic, 1 = none, at most 4)", n);
        dm.split = n;
    });
}

int32_t visp_depthany_enable_captures(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_depthany(m).captures = enable != 0; });
}

int32_t visp_depthany_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity, int64_t* n_written,
                                   int64_t shape[4]) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        read_capture(*dm.backend, dm.capture_bufs, name, host_out, capacity, n_written, shape);
    });
}

int32_t visp_depthany_enable_timing(visp_model* m, int32_t enable) {
    // 1 = the whole batch on one stream; 2 = the step's own launch shapes (sub-batches on parallel streams, every launch timed on its stream)
    return handle_errors([&]() { as_depthany(m).timing = enable != 0; as_depthany(m).timing_split = enable == 2; });
}

int32_t visp_depthany_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n) {
    return handle_errors([&]() {
        depthany_model& dm = as_depthany(m);
        int32_t count = 0;
        for (timing_entry const& t : dm.last_timing) {
            if (count >= cap) break;
            snprintf(out[count].name, sizeof out[count].name, "%s", t.name.c_str());
            out[count].ms = t.ms;
            out[count].launches = t.launches;
            out[count].flops = t.flops;
            out[count].bytes = t.bytes;
            ++count;
        }
        *n = count;
    });
}

// ---- ESRGAN extension ----------------------------------------------------------------------

int32_t visp_esrgan_get_info(visp_model const* m, visp_esrgan_info* out) {
    return handle_errors([&]() {
        esrgan_model const& em = as_esrgan(const_cast<visp_model*>(m));
        out->scale = em.params.scale;
        out->n_blocks = em.params.n_blocks;
        out->n_filters = em.nf;
        out->growth = em.gc;
        out->tile_group = em.tile_group;
    });
}

int32_t visp_esrgan_set_tile_group(visp_model* m, int32_t tiles) {
    return handle_errors([&]() {
        if (tiles < 1) throw except("esrgan: tile group must be >= 1");
        as_esrgan(m).tile_group = tiles;
    });
}

int32_t visp_esrgan_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes) {
    return handle_errors([&]() {
        esrgan_model& em = as_esrgan(m);
        *device_ptr = em.weight_arena.ptr;
        *n_bytes = em.weight_arena.bytes;
    });
}

int32_t visp_esrgan_weights_ready(visp_model* m) {
    return handle_errors([&]() { esrgan_weights_ready(as_esrgan(m)); });
}

int32_t visp_esrgan_tile_layout(int32_t w, int32_t h, int32_t scale, int32_t out8[8]) {
    return handle_errors([&]() {
        tile_layout t = tile_scale(tile_layout({{w, h}}, 224, 16), scale);
        int32_t v[8] = {t.image_extent[0], t.image_extent[1], t.overlap[0], t.overlap[1], t.n_tiles[0], t.n_tiles[1], t.tile_size[0], t.tile_size[1]};
        memcpy(out8, v, sizeof v);
    });
}

int32_t visp_esrgan_compute_batch_device(visp_model* m, void const* img, int32_t batch, int32_t w, int32_t h, int32_t format, void* out_rgba,
                                         void* stream) {
    return handle_errors([&]() {
        if (!img || !out_rgba) throw except("esrgan: null input/output pointer");
        esrgan_compute_batch_device(as_esrgan(m), img, batch, w, h, image_format(format), out_rgba, stream);
    });
}

int32_t visp_esrgan_compute_batch_host(visp_model* m, uint8_t const* img, int32_t batch, int32_t w, int32_t h, int32_t format, uint8_t* out_rgba) {
    return handle_errors([&]() {
        if (!img || !out_rgba) throw except("esrgan: null input/output pointer");
        esrgan_compute_batch_host(as_esrgan(m), img, batch, w, h, image_format(format), out_rgba);
    });
}

int32_t visp_esrgan_generate_host(visp_model* m, float const* rgb, int32_t n, int32_t w, int32_t h, float* out) {
    return handle_errors([&]() {
        if (!rgb || !out) throw except("esrgan: null input/output pointer");
        esrgan_generate_host(as_esrgan(m), rgb, n, w, h, out);
    });
}

int32_t visp_esrgan_enable_timing(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_esrgan(m).timing = enable != 0; });
}

int32_t visp_esrgan_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n) {
    return handle_errors([&]() {
        esrgan_model& em = as_esrgan(m);
        int32_t count = 0;
        for (timing_entry const& t : em.last_timing) {
            if (count >= cap) break;
            snprintf(out[count].name, sizeof out[count].name, "%s", t.name.c_str());
            out[count].ms = t.ms;
            out[count].launches = t.launches;
            out[count].flops = t.flops;
            out[count].bytes = t.bytes;
            ++count;
        }
        *n = count;
    });
}

// ---- MobileSAM image encoder (TinyViT) ----

int32_t visp_sam_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes) {
    return handle_errors([&]() {
        sam_model& sm = as_sam(m);
        *device_ptr = sm.weight_arena.ptr;
        *n_bytes = sm.weight_arena.bytes;
    });
}

int32_t visp_sam_weights_ready(visp_model* m) {
    return handle_errors([&]() { sam_weights_ready(as_sam(m)); });
}

int32_t visp_sam_encode(visp_model* m, visp_image_view const* image) {
    return handle_errors([&]() {
        if (!image) throw except("sam: null image");
        image_view in;
        in.extent = {{image->width, image->height}};
        in.stride = image->stride;
        in.format = image_format(image->format);
        in.data = image->data;
        sam_encode(as_sam(m), in);
    });
}

int32_t visp_sam_read_embedding(visp_model* m, float* host_out, int64_t capacity, int64_t shape[3]) {
    return handle_errors([&]() {
        sam_model& sm = as_sam(m);
        if (!sm.embed.ptr || sm.image_extent[0] <= 0) throw except("Missing image embeds, call sam_encode() first");
        const int R = sm.params.layers[3].resolution;
        shape[0] = R; shape[1] = R; shape[2] = 256;
        if (capacity < (int64_t)R * R * 256) throw except("sam: embedding buffer too small (%lld < %d)", (long long)capacity, R * R * 256);
        if (!vx_memcpy_d2h(host_out, sm.embed.ptr, (size_t)R * R * 256 * 4, sm.backend->stream) || !vx_stream_sync(sm.backend->stream))
            throw except("%s", vx_last_error());
    });
}

int32_t visp_sam_compute(visp_model* m, int32_t const* prompt, int32_t n_prompt, visp_image_view* out_image, visp_image_data** out_data) {
    return handle_errors([&]() {
        if (!prompt || !out_image || !out_data) throw except("sam: null argument");
        return_image(sam_compute(as_sam(m), prompt, n_prompt), out_image, out_data);
    });
}

int32_t visp_sam_read_masks(visp_model* m, float* masks, int64_t capacity, float iou[4]) {
    return handle_errors([&]() {
        sam_model& sm = as_sam(m);
        if (sm.last_masks.empty()) throw except("sam: no mask logits kept: enable captures, then call sam_compute()");
        if (capacity < (int64_t)sm.last_masks.size()) throw except("sam: mask buffer too small (%lld < %zu)", (long long)capacity, sm.last_masks.size());
        memcpy(masks, sm.last_masks.data(), sm.last_masks.size() * sizeof(float));
        memcpy(iou, sm.last_iou, sizeof sm.last_iou);
    });
}

int32_t visp_sam_encode_batch_device(visp_model* m, void const* rgb, int32_t batch, void* out, void* stream) {
    return handle_errors([&]() { sam_encode_batch_device(as_sam(m), rgb, batch, out, stream); });
}

int32_t visp_sam_encode_batch_host(visp_model* m, uint8_t const* rgb, int32_t batch, float* out) {
    return handle_errors([&]() { sam_encode_batch_host(as_sam(m), rgb, batch, out); });
}

int32_t visp_sam_set_fp8_mlp(visp_model* m, int32_t enable) {
    return handle_errors([&]() { sam_set_fp8_mlp(as_sam(m), enable != 0); });
}

int32_t visp_sam_enable_captures(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_sam(m).captures = enable != 0; });
}

int32_t visp_sam_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity, int64_t* n_written, int64_t shape[4]) {
    return handle_errors([&]() {
        sam_model& sm = as_sam(m);
        read_capture(*sm.backend, sm.capture_bufs, name, host_out, capacity, n_written, shape);
    });
}

int32_t visp_sam_enable_timing(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_sam(m).timing = enable != 0; });
}

int32_t visp_sam_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n) {
    return handle_errors([&]() {
        sam_model& sm = as_sam(m);
        int32_t count = 0;
        for (timing_entry const& t : sm.last_timing) {
            if (count >= cap) break;
            snprintf(out[count].name, sizeof out[count].name, "%s", t.name.c_str());
            out[count].ms = t.ms;
            out[count].launches = t.launches;
            out[count].flops = t.flops;
            out[count].bytes = t.bytes;
            ++count;
        }
        *n = count;
    });
}

// ---- SWIN encoder (BiRefNet backbone) ----------------------------------------------------------------------------------

int32_t visp_swin_load(char const* filepath, visp_device const* dev, visp_model** out) {
    return handle_errors([&]() {
        if (!filepath || !dev || !out) throw except("visp_swin_load: null argument");
        *out = handle_of(swin_load_model(filepath, *reinterpret_cast<backend_device const*>(dev)));
    });
}

int32_t visp_birefnet_compute_batch_device(visp_model* m, void const* rgb_u8, int32_t batch, int32_t w, int32_t h, void* mask_f32, void* stream) {
    return handle_errors([&]() { birefnet_compute_batch_device(as_birefnet(m), rgb_u8, batch, w, h, mask_f32, stream); });
}

int32_t visp_birefnet_compute_batch_host(visp_model* m, uint8_t const* rgb_u8, int32_t batch, int32_t w, int32_t h, float* mask) {
    return handle_errors([&]() { birefnet_compute_batch_host(as_birefnet(m), rgb_u8, batch, w, h, mask); });
}

int32_t visp_birefnet_image_extent(visp_model* m, int32_t w, int32_t h, int32_t* out_w, int32_t* out_h) {
    return handle_errors([&]() {
        i32x2 e = birefnet_image_extent({{w, h}}, as_birefnet(m).bparams);
        *out_w = e[0];
        *out_h = e[1];
    });
}

int32_t visp_swin_output_dims(visp_model* m, int32_t w, int32_t h, int32_t dims[12]) {
    return handle_errors([&]() {
        int d[4][3];
        swin_output_dims(as_swin(m), w, h, d);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) dims[i * 3 + j] = d[i][j];
    });
}

int32_t visp_swin_encode_batch_device(visp_model* m, void const* rgb_u8, int32_t batch, int32_t w, int32_t h, void* const outs[4], void* stream) {
    return handle_errors([&]() { swin_encode_batch_device(as_swin(m), rgb_u8, batch, w, h, outs, stream); });
}

int32_t visp_swin_encode_batch_host(visp_model* m, uint8_t const* rgb_u8, int32_t batch, int32_t w, int32_t h, float* const outs[4]) {
    return handle_errors([&]() { swin_encode_batch_host(as_swin(m), rgb_u8, batch, w, h, outs); });
}

int32_t visp_swin_enable_captures(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_swin(m).captures = enable != 0; });
}

int32_t visp_swin_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity, int64_t* n_written, int64_t shape[4]) {
    return handle_errors([&]() {
        swin_model& sm = as_swin(m);
        read_capture(*sm.backend, sm.capture_bufs, name, host_out, capacity, n_written, shape);
    });
}

int32_t visp_swin_set_mask_mode(visp_model* m, int32_t shifted_only) {
    return handle_errors([&]() { as_swin(m).mask_shifted_only = shifted_only != 0; });
}

int32_t visp_swin_enable_timing(visp_model* m, int32_t enable) {
    return handle_errors([&]() { as_swin(m).timing = enable != 0; });
}

int32_t visp_swin_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n) {
    return handle_errors([&]() {
        int32_t count = 0;
        for (timing_entry const& t : as_swin(m).last_timing) {
            if (count >= cap) break;
            snprintf(out[count].name, sizeof out[count].name, "%s", t.name.c_str());
            out[count].ms = t.ms;
            out[count].launches = t.launches;
            out[count].flops = t.flops;
            out[count].bytes = t.bytes;
            ++count;
        }
        *n = count;
    });
}

// ---- model file: the GGUF key/value surface of the reference's model_file (ml.h:85-103, ml.cpp:206-281) ----

static model_file const& as_file(visp_file const* f) {
    if (!f) throw except("model file handle is null");
    return f->file;
}
int32_t visp_file_load(char const* gguf_path, visp_file** out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_file_load: null out pointer");
        if (!gguf_path) throw except("visp_file_load: null path");
        *out = new visp_file{model_load(gguf_path)};
    });
}
void visp_file_destroy(visp_file* f) { delete f; }
int32_t visp_file_n_tensors(visp_file const* f, int64_t* out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_file_n_tensors: null out pointer");
        *out = as_file(f).n_tensors();
    });
}
int32_t visp_file_get_int(visp_file const* f, char const* key, int32_t* out) {
    return handle_errors([&]() {
        if (!key || !out) throw except("visp_file_get_int: null argument");
        *out = as_file(f).get_int(key);
    });
}
int32_t visp_file_get_int_array(visp_file const* f, char const* key, int32_t* out, int64_t n) {
    return handle_errors([&]() {
        if (!key || !out || n < 0) throw except("visp_file_get_int_array: bad argument");
        as_file(f).get_array(key, out, (size_t)n);
    });
}
int32_t visp_file_get_string(visp_file const* f, char const* key, char* out, int64_t capacity, int64_t* needed) {
    return handle_errors([&]() {
        if (!key) throw except("visp_file_get_string: null key");
        std::string_view v = as_file(f).get_string(key);
        if (needed) *needed = (int64_t)v.size() + 1;
        if (out && capacity > 0) snprintf(out, (size_t)capacity, "%.*s", (int)v.size(), v.data());
    });
}
int32_t visp_weights_from_file(visp_file const* f, visp_weights** out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_weights_from_file: null out pointer");
        *out = new visp_weights{weights_from_file(as_file(f))};
    });
}

// ---- graph layer (graph.h) ----

int32_t visp_weights_load(char const* gguf_path, visp_weights** out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_weights_load: null out pointer");
        *out = new visp_weights{weights_load(gguf_path)};
    });
}
int32_t visp_weights_create(visp_weights** out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_weights_create: null out pointer");
        *out = new visp_weights{weights_create()};
    });
}
int32_t visp_weights_add(visp_weights* w, char const* name, int32_t dtype, int64_t const ne[4], float const* data) {
    return handle_errors([&]() {
        if (!w) throw except("weights handle is null");
        weights_add(*w->store, name, dtype, ne, data);
    });
}
void visp_weights_destroy(visp_weights* w) { delete w; }

int32_t visp_graph_create(visp_weights* weights, visp_graph** out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_graph_create: null out pointer");
        *out = static_cast<visp_graph*>(graph_create(weights ? weights->store : nullptr));
    });
}
void visp_graph_destroy(visp_graph* g) { delete static_cast<graph*>(g); }

static graph& as_graph(visp_graph* g) {
    if (!g) throw except("graph handle is null");
    return *g;
}
static graph const& as_cgraph(visp_graph const* g) {
    if (!g) throw except("graph handle is null");
    return *g;
}

int32_t visp_graph_add_weight(visp_graph* g, char const* name, int32_t dtype, int64_t const ne[4], float const* data, int32_t* out) {
    return handle_errors([&]() {
        const int t = graph_add_weight(as_graph(g), name, dtype, ne, data);
        if (out) *out = t;
    });
}
int32_t visp_graph_find_weight(visp_graph* g, char const* name, int32_t* out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_graph_find_weight: null out pointer");
        *out = graph_find_weight(as_graph(g), name);
    });
}
int32_t visp_graph_input(visp_graph* g, int32_t dtype, int64_t const ne[4], char const* name, int32_t* out) {
    return handle_errors([&]() {
        if (!out || !ne) throw except("visp_graph_input: null pointer");
        *out = graph_input(as_graph(g), dtype, ne, name);
    });
}
int32_t visp_graph_op(visp_graph* g, int32_t op, int32_t const* src, int32_t n_src, int64_t const* iparams, int32_t n_iparams, float const* fparams,
                      int32_t n_fparams, int32_t* out) {
    return handle_errors([&]() {
        if (!out || !src) throw except("visp_graph_op: null pointer");
        *out = graph_add(as_graph(g), op, src, n_src, iparams, n_iparams, fparams, n_fparams);
    });
}
int32_t visp_graph_set_name(visp_graph* g, int32_t tensor, char const* name) {
    return handle_errors([&]() { graph_set_name(as_graph(g), tensor, name); });
}
int32_t visp_graph_get_tensor(visp_graph* g, char const* name, int32_t* out) {
    return handle_errors([&]() {
        if (!out) throw except("visp_graph_get_tensor: null out pointer");
        *out = graph_get_tensor(as_graph(g), name);
    });
}
int32_t visp_graph_output(visp_graph* g, int32_t tensor, char const* name) {
    return handle_errors([&]() { graph_output(as_graph(g), tensor, name); });
}
int32_t visp_graph_tensor_info(visp_graph const* g, int32_t tensor, int32_t* dtype, int64_t ne[4], int32_t* is_constant) {
    return handle_errors([&]() {
        graph const& gr = as_cgraph(g);
        if (tensor < 0 || tensor >= (int)gr.nodes.size()) throw except("visp_graph_tensor_info: tensor handle %d is not part of this graph", tensor);
        graph_node const& n = gr.nodes[tensor];
        if (dtype) *dtype = n.dtype;
        if (ne) for (int i = 0; i < 4; ++i) ne[i] = n.ne[i];
        if (is_constant) *is_constant = n.constant;
    });
}
int32_t visp_graph_read_constant(visp_graph const* g, int32_t tensor, float* out, int64_t capacity) {
    return handle_errors([&]() {
        graph const& gr = as_cgraph(g);
        if (tensor < 0 || tensor >= (int)gr.nodes.size()) throw except("visp_graph_read_constant: tensor handle %d is not part of this graph", tensor);
        graph_node const& n = gr.nodes[tensor];
        if (!out) throw except("visp_graph_read_constant: null out pointer");
        if (!n.constant) throw except("visp_graph_read_constant: tensor %d is not a constant", tensor);
        if (capacity < n.n_elements()) throw except("visp_graph_read_constant: capacity %lld < %lld elements", (long long)capacity, (long long)n.n_elements());
        memcpy(out, n.values(), (size_t)n.n_elements() * 4);
    });
}
int32_t visp_graph_allocate(visp_graph* g, visp_device const* dev) {
    return handle_errors([&]() { graph_allocate(as_graph(g), dev); });
}
int32_t visp_graph_use_hip_graph(visp_graph* g, int32_t enable) {
    return handle_errors([&]() { as_graph(g).use_hip_graph = enable != 0; });
}
int32_t visp_graph_set_fused_models(visp_graph* g, int32_t enable) {
    return handle_errors([&]() {
        graph& gr = as_graph(g);
        if (gr.allocated) throw except("visp_graph_set_fused_models: the graph is already allocated");
        gr.fused_models = enable != 0;
    });
}
int32_t visp_graph_compute(visp_graph* g) {
    return handle_errors([&]() { graph_compute(as_graph(g)); });
}
int32_t visp_graph_tensor_set(visp_graph* g, int32_t tensor, void const* data, size_t n_bytes) {
    return handle_errors([&]() { graph_tensor_set(as_graph(g), tensor, data, n_bytes); });
}
int32_t visp_graph_tensor_get(visp_graph* g, int32_t tensor, void* data, size_t n_bytes, int32_t as_f32) {
    return handle_errors([&]() { graph_tensor_get(as_graph(g), tensor, data, n_bytes, as_f32 != 0); });
}
int32_t visp_graph_describe(visp_graph const* g, char* out, int64_t capacity, int64_t* needed) {
    return handle_errors([&]() {
        std::string s = graph_describe(as_cgraph(g));
        if (needed) *needed = (int64_t)s.size() + 1;
        if (out && capacity > 0) snprintf(out, (size_t)capacity, "%s", s.c_str());
    });
}

} // extern "C"
