#include "esrgan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "../../include/visp_hip_kernels.h"
#include "graph.h"
#include "visp_util.h"

namespace visp {

#define VX(call)                                        \
    do {                                                \
        if (!(call)) throw except("%s", vx_last_error()); \
    } while (0)

namespace {

template <typename T>
T round_up(T x, T m) { return (x + m - 1) / m * m; }

constexpr int esrgan_default_tile_size = 224; // vision.cpp:208
constexpr int esrgan_tile_overlap = 16;       // vision.cpp:221

} // namespace

//
// parameters and tiling (reference esrgan.cpp:81-97, image.cpp:612-651)

esrgan_params esrgan_detect_params(model_file const& f) {
    if (std::string_view arch = f.arch(); arch != "esrgan")
        throw except("Architecture expected to be 'esrgan', but was '%.*s' (%s)", (int)arch.size(), arch.data(), f.path.c_str());
    esrgan_params p;
    p.scale = f.get_int("esrgan.scale");
    p.n_blocks = f.get_int("esrgan.block_count");
    if (p.scale < 1 || p.scale > 8) throw except("ESRGAN: unsupported scale: %d", p.scale);
    if (p.n_blocks < 1 || p.n_blocks > 23) throw except("ESRGAN: invalid number of blocks: %d", p.n_blocks);
    return p;
}

tile_layout::tile_layout(i32x2 extent, int max_tile_size, int ov, int align) {
    image_extent = extent;
    overlap = {{ov, ov}};
    for (int i = 0; i < 2; ++i) {
        n_tiles[i] = div_ceil(extent[i], max_tile_size);
        int img_extent_overlap = extent[i] + (n_tiles[i] - 1) * ov;
        tile_size[i] = div_ceil(div_ceil(img_extent_overlap, n_tiles[i]), align) * align;
    }
}

tile_layout tile_scale(tile_layout const& o, int scale) {
    tile_layout s;
    for (int i = 0; i < 2; ++i) {
        s.image_extent[i] = o.image_extent[i] * scale;
        s.overlap[i] = o.overlap[i] * scale;
        s.tile_size[i] = o.tile_size[i] * scale;
        s.n_tiles[i] = o.n_tiles[i];
    }
    return s;
}

namespace {

vx_tile_layout to_vx(tile_layout const& t) {
    return {t.image_extent[0], t.image_extent[1], t.overlap[0], t.overlap[1], t.n_tiles[0], t.n_tiles[1], t.tile_size[0], t.tile_size[1]};
}

int log2_floor(int x) { // src/util/math.h:24-31 (integer log2 of the scale)
    int r = 0;
    while (x > 1) { x >>= 1; ++r; }
    return r;
}

//
// the generator as a graph: the nodes of the reference's esrgan_generate (src/visp/arch/esrgan.cpp:13-79), one node per call it makes. csrc/graph.cpp
// lowers them onto the LDS-ring conv in its planar layout -- the f32 image as a value | residue plane, a dense block as six planes of one buffer (concat is
// no launch), LeakyReLU / * 0.2 + x / (.) * 0.2 + rrdb input as epilogues, the x2 resize in the up-conv's loader, the last conv straight to f32 RGB -- which
// is the schedule this file issued by hand in rounds 1-3 (tests/test_graph_cpu.py holds the list).

struct generator_builder {
    graph& g;
    int weight(std::string const& name) const {
        const int t = graph_find_weight(g, name.c_str());
        if (t < 0) throw except("tensor not found: %s", name.c_str());
        return t;
    }
    int node(int32_t op, std::vector<int> const& src, std::vector<int64_t> const& ip = {}, std::vector<float> const& fp = {}) const {
        return graph_add(g, op, src.data(), (int)src.size(), ip.data(), (int)ip.size(), fp.data(), (int)fp.size());
    }
    int conv(std::string const& mod, int x) const { // conv_2d(m, x, 1, 1): 3x3, stride 1, pad 1, bias if the file has one
        const int b = graph_find_weight(g, (mod + ".bias").c_str());
        return b >= 0 ? node(gop_conv_2d, {x, weight(mod + ".weight"), b}, {1, 1}) : node(gop_conv_2d, {x, weight(mod + ".weight")}, {1, 1});
    }
    int lrelu(int x) const { return node(gop_leaky_relu, {x}, {}, {0.2f}); }
    int conv_block(std::string const& mod, int x) const { return lrelu(conv(mod + ".0", x)); }                  // esrgan.cpp:21-25
    int dense_block(std::string const& mod, int x) const {                                                      // esrgan.cpp:27-41
        int c = x;
        for (int k = 1; k <= 4; ++k) c = node(gop_concat, {c, conv_block(mod + ".conv" + std::to_string(k), c)}, {0});
        const int x5 = node(gop_scale, {conv(mod + ".conv5.0", c)}, {}, {0.2f});
        return node(gop_add, {x, x5});
    }
    int rrdb(std::string const& mod, int x) const {                                                             // esrgan.cpp:43-51
        int y = x;
        for (int r = 1; r <= 3; ++r) y = dense_block(mod + ".RDB" + std::to_string(r), y);
        return node(gop_add, {node(gop_scale, {y}, {}, {0.2f}), x});
    }
    int build(int image, esrgan_params const& P) const {                                                        // esrgan.cpp:55-79
        int x = conv("model.0", image);
        int sub = x;
        for (int i = 0; i < P.n_blocks; ++i) sub = rrdb("model.1.sub." + std::to_string(i), sub);
        sub = conv("model.1.sub." + std::to_string(P.n_blocks), sub);
        x = node(gop_add, {x, sub});
        int seq = 2;
        for (int i = 0; i < log2_floor(P.scale); ++i) {                                                         // esrgan::upsample, esrgan.cpp:13-19
            x = node(gop_interpolate, {x}, {g.nodes[x].ne[1] * 2, g.nodes[x].ne[2] * 2, 0});
            x = lrelu(conv("model." + std::to_string(seq + 1), x));
            seq += 3;
        }
        x = lrelu(conv("model." + std::to_string(seq), x));
        return conv("model." + std::to_string(seq + 2), x);
    }
};

std::unique_ptr<graph> generator_graph(esrgan_model& m, int n, int w, int h, int* in, int* out) {
    std::unique_ptr<graph> g(graph_create(m.store));
    const int64_t ne[4] = {3, w, h, n};
    *in = graph_input(*g, gdt_f32, ne, "image");
    *out = generator_builder{*g}.build(*in, m.params);
    graph_output(*g, *out, "result");
    return g;
}

} // namespace

esrgan_step::esrgan_step() = default;
esrgan_step::~esrgan_step() = default;

esrgan_model* esrgan_load_model(char const* filepath, backend_device const& dev, int flags) {
    const bool with_data = !(flags & load_no_upload);
    model_file file = model_load(filepath, /*header_only=*/!with_data);
    auto model = std::make_unique<esrgan_model>();
    model->backend = &dev;
    model->params = esrgan_detect_params(file);
    esrgan_params const& P = model->params;
    if (P.scale != 1 && P.scale != 2 && P.scale != 4 && P.scale != 8)
        throw except("ESRGAN: scale %d is not built in this backend (powers of two only)", P.scale);
    model->store = weights_from_file(file); // a header-only read gives zero-filled tensors: shapes, and an arena layout to receive the broadcast into

    // the shapes the planar conv schedule is built for (64 filters, growth 32); everything else about the file is checked by the graph's shape inference
    auto shape = [&](std::string const& name) -> weight_store::entry const& {
        auto it = model->store->tensors.find(name);
        if (it == model->store->tensors.end()) throw except("tensor not found: %s", name.c_str());
        return it->second;
    };
    weight_store::entry const& first = shape("model.0.weight"); // ne [Cin, kw, kh, Cout]
    if (first.ne[0] != 3) throw except("ESRGAN: model.0 takes %d input channels, expected 3", (int)first.ne[0]);
    model->nf = (int)first.ne[3];
    model->gc = (int)shape("model.1.sub.0.RDB1.conv1.0.weight").ne[3];
    if (model->nf != 64 || model->gc != 32)
        throw except("ESRGAN: %d filters / %d growth channels are not built in this backend (64 / 32 only)", model->nf, model->gc);
    for (int i = 0; i < P.n_blocks; ++i)
        for (int r = 1; r <= 3; ++r)
            for (int k = 1; k <= 5; ++k) {
                weight_store::entry const& t = shape("model.1.sub." + std::to_string(i) + ".RDB" + std::to_string(r) + ".conv" + std::to_string(k) + ".0.weight");
                if (t.ne[0] != model->nf + (k - 1) * model->gc || t.ne[3] != (k < 5 ? model->gc : model->nf))
                    throw except("ESRGAN: dense block conv%d has shape %d -> %d", k, (int)t.ne[0], (int)t.ne[3]);
            }
    {
        const int seq = 2 + 3 * log2_floor(P.scale);
        if (shape("model." + std::to_string(seq + 2) + ".weight").ne[3] != 3) throw except("ESRGAN: the last conv has %d outputs, expected 3", (int)shape("model." + std::to_string(seq + 2) + ".weight").ne[3]);
    }

    device_turn turn(dev);
    VX(vx_dconv_prepare());
    // The weight arena: one planning pass of the generator's graph says how many bytes of operand images the lowering makes of the weights; the second
    // pass packs them into one allocation in lowering order -- the same order on every rank (what the RCCL broadcast at load time moves).
    size_t need = 0;
    {
        int in, out;
        std::unique_ptr<graph> plan = generator_graph(*model, 1, 32, 32, &in, &out);
        graph_allocate(*plan, nullptr);
        need = plan->plan_store_bytes;
    }
    weight_store& ws = *model->store;
    ws.arena.bytes = need + (1u << 20);
    VX(vx_malloc(&ws.arena.ptr, ws.arena.bytes));
    {
        int in, out;
        std::unique_ptr<graph> warm = generator_graph(*model, 1, 32, 32, &in, &out);
        graph_allocate(*warm, &dev);
    }
    model->weight_arena.ptr = ws.arena.ptr;
    model->weight_arena.bytes = round_up<size_t>(ws.arena_used, 256);
    model->weights_uploaded = with_data;
    if (const char* e = getenv("VISP_ESRGAN_TILE_GROUP")) model->tile_group = std::max(1, atoi(e));
    if (const char* e = getenv("VISP_ESRGAN_STREAMS")) model->streams = std::max(1, atoi(e));
    VX(vx_stream_create(&model->aux_stream));
    VX(vx_event_create(&model->fork_event));
    VX(vx_event_create(&model->join_event));
    return model.release();
}

void esrgan_weights_ready(esrgan_model& m) {
    m.store->no_data = true; // every operand image this rank has is the one in the arena the broadcast filled
    m.weights_uploaded = true;
}

esrgan_model::~esrgan_model() {
    if (backend) vx_set_device(backend->index);
    if (aux_stream) {
        vx_stream_sync(aux_stream);
        vx_stream_destroy(aux_stream);
    }
    if (fork_event) vx_event_destroy(fork_event);
    if (join_event) vx_event_destroy(join_event);
    for (auto& lane : steps) lane.clear();
    vx_free(ws.arena.ptr);
    weight_arena = {}; // (the store owns it)
}

//
// workspace + executor

namespace {

// ws.img_in / ws.img_out (capacity for the host entry point's u8 staging images) are kept, everything else is sized for this call. The activations
// of a tile group live in that group's graph arena (liveness-planned by the graph layer).
void reserve(esrgan_model& m, int n_tiles_total, int tw, int th) {
    esrgan_workspace& ws = m.ws;
    const size_t img_in_bytes = ws.img_in, img_out_bytes = ws.img_out;
    // two concurrent lanes when there is enough work to split (timing runs keep one lane: events on one stream)
    const int lanes = (m.streams >= 2 && !m.timing && n_tiles_total >= 8) ? 2 : 1;
    const int s = m.params.scale;
    const size_t px = (size_t)tw * th;
    // the conv kernel addresses a map's planes through a 32-bit buffer descriptor: keep two planes of the largest
    // (up-sampled) map of a group below 2 GiB
    const int addr_cap = (int)std::max<size_t>(1, ((size_t)1 << 30) / (px * s * s * 64));
    const int group = std::min({(n_tiles_total + lanes - 1) / lanes, m.tile_group, addr_cap});
    struct item { void** p; size_t bytes; };
    std::vector<item> items = {
        {&ws.in_u8, img_in_bytes}, {&ws.out_u8, img_out_bytes},
        {&ws.x0, (size_t)n_tiles_total * px * 3 * 4}, {&ws.tiles_out, (size_t)n_tiles_total * px * s * s * 3 * 4}};
    size_t total = 0;
    for (item& it : items) total += round_up<size_t>(it.bytes, 256);
    if (total > ws.arena.bytes) {
        VX(vx_stream_sync(m.backend->stream));
        if (m.aux_stream) VX(vx_stream_sync(m.aux_stream));
        vx_free(ws.arena.ptr);
        ws.arena = {};
        VX(vx_malloc(&ws.arena.ptr, total));
        ws.arena.bytes = total;
    }
    uint8_t* p = static_cast<uint8_t*>(ws.arena.ptr);
    for (item& it : items) {
        *it.p = p;
        p += round_up<size_t>(it.bytes, 256);
    }
    ws.group = group; ws.lanes = lanes; ws.tile_w = tw; ws.tile_h = th; ws.scale = s;
}

// the graph of `n` tiles of this extent on `lane` (two lanes run concurrently: each has graphs -- and arenas -- of its own); the two most recent shapes
// per lane are kept
esrgan_step& step_for(esrgan_model& m, int lane, int n, int w, int h) {
    auto& cache = m.steps[lane];
    for (size_t i = 0; i < cache.size(); ++i)
        if (cache[i]->n == n && cache[i]->w == w && cache[i]->h == h) {
            if (i + 1 != cache.size()) std::rotate(cache.begin() + (long)i, cache.begin() + (long)i + 1, cache.end());
            return *cache.back();
        }
    if (cache.size() >= 3) { // the oldest one's launches may still be queued
        VX(vx_stream_sync(m.backend->stream));
        if (m.aux_stream) VX(vx_stream_sync(m.aux_stream));
        cache.erase(cache.begin());
    }
    auto st = std::make_unique<esrgan_step>();
    st->n = n; st->w = w; st->h = h;
    st->g = generator_graph(m, n, w, h, &st->in, &st->out);
    graph_allocate(*st->g, m.backend);
    cache.push_back(std::move(st));
    return *cache.back();
}

struct exec {
    esrgan_model& m;
    void* stream;
    std::vector<std::pair<std::string, void*>> marks;
    std::vector<timing_entry> acc;

    void mark(const char* name, double flops, double bytes) {
        if (!m.timing) return;
        void* ev = nullptr;
        VX(vx_event_create(&ev));
        VX(vx_event_record(ev, stream));
        marks.push_back({name, ev});
        acc.push_back({name, 0, 1, flops, bytes});
    }
    void finish_timing() {
        if (!m.timing) return;
        void* ev = nullptr;
        VX(vx_event_create(&ev));
        VX(vx_event_record(ev, stream));
        marks.push_back({"end", ev});
        std::map<std::string, timing_entry> by;
        std::vector<std::string> order;
        for (size_t i = 0; i + 1 < marks.size(); ++i) {
            float ms = 0;
            VX(vx_event_elapsed_ms(marks[i].second, marks[i + 1].second, &ms));
            auto it = by.find(marks[i].first);
            if (it == by.end()) { order.push_back(marks[i].first); it = by.emplace(marks[i].first, timing_entry{marks[i].first, 0, 0, 0, 0}).first; }
            it->second.ms += ms;
            it->second.launches += acc[i].launches;
            it->second.flops += acc[i].flops;
            it->second.bytes += acc[i].bytes;
        }
        m.last_timing.clear();
        for (auto& n : order) m.last_timing.push_back(by[n]);
        for (auto& mk : marks) vx_event_destroy(mk.second);
        marks.clear();
    }

    // esrgan_generate (esrgan.cpp:55-79) on n tiles: rgb f32 [n, h, w, 3] -> rgb f32 [n, h*s, w*s, 3], both in the call's workspace
    void generate(int lane, const float* x0, int n, int w, int h, float* out) {
        esrgan_step& st = step_for(m, lane, n, w, h);
        graph_bind_external(*st.g, st.in, const_cast<float*>(x0));
        graph_bind_external(*st.g, st.out, out);
        for (graph_launch const& l : st.g->launches) {
            mark(l.group.empty() ? "other" : l.group.c_str(), l.flops, l.bytes);
            l.run(stream);
        }
    }
};

void run_tiles(esrgan_model& m, exec& ex, int n_total, int tw, int th) {
    const int s = m.params.scale;
    const size_t px = (size_t)tw * th;
    void* const main_stream = ex.stream;
    const bool two = m.ws.lanes == 2 && n_total > m.ws.group;
    if (two) { // fork: the second lane starts after what is already queued on the caller's stream (tiles_in)
        VX(vx_event_record(m.fork_event, main_stream));
        VX(vx_stream_wait_event(m.aux_stream, m.fork_event));
    }
    int gi = 0;
    for (int t0 = 0; t0 < n_total; t0 += m.ws.group, ++gi) {
        const int n = std::min(m.ws.group, n_total - t0);
        const int lane = two ? gi & 1 : 0;
        ex.stream = lane ? m.aux_stream : main_stream;
        ex.generate(lane, static_cast<const float*>(m.ws.x0) + (size_t)t0 * px * 3, n, tw, th, static_cast<float*>(m.ws.tiles_out) + (size_t)t0 * px * s * s * 3);
    }
    ex.stream = main_stream;
    if (two) { // join
        VX(vx_event_record(m.join_event, m.aux_stream));
        VX(vx_stream_wait_event(main_stream, m.join_event));
    }
}

} // namespace

void esrgan_compute_batch_device(esrgan_model& m, void const* img_dev, int batch, int w, int h, image_format format, void* out_rgba_dev,
                                 void* stream) {
    if (!m.weights_uploaded) throw except("esrgan: weights have not been uploaded (load_no_upload without weights_ready)");
    if (batch < 1 || w < 1 || h < 1) throw except("esrgan: empty input (%d images of %dx%d)", batch, w, h);
    if (is_float(format) || n_channels(format) < 3) throw except("esrgan: unsupported input image format [%d], expected an 8-bit colour image", int(format));
    device_turn turn(*m.backend);
    void* s = stream ? stream : m.backend->stream;
    tile_layout tiles({{w, h}}, esrgan_default_tile_size, esrgan_tile_overlap);
    tile_layout tiles_out = tile_scale(tiles, m.params.scale);
    const int n_total = batch * tiles.total();
    reserve(m, n_total, tiles.tile_size[0], tiles.tile_size[1]);
    exec ex{m, s, {}, {}};
    vx_tile_layout tin = to_vx(tiles), tout = to_vx(tiles_out);
    ex.mark("tiles_in", 0, (double)n_total * tiles.tile_size[0] * tiles.tile_size[1] * 15);
    VX(vx_esrgan_tiles_in_f32(static_cast<const uint8_t*>(img_dev), batch, w, h, int(format), &tin, static_cast<float*>(m.ws.x0), s));
    run_tiles(m, ex, n_total, tiles.tile_size[0], tiles.tile_size[1]);
    ex.mark("tiles_out", 0, (double)batch * tout.image_w * tout.image_h * 20);
    VX(vx_esrgan_tiles_out(static_cast<const float*>(m.ws.tiles_out), batch, &tout, nullptr, static_cast<uint8_t*>(out_rgba_dev), s));
    ex.finish_timing();
    if (!stream) VX(vx_stream_sync(s));
}

void esrgan_compute_batch_host(esrgan_model& m, uint8_t const* img, int batch, int w, int h, image_format format, uint8_t* out_rgba) {
    if (batch < 1 || w < 1 || h < 1) throw except("esrgan: empty input (%d images of %dx%d)", batch, w, h);
    if (is_float(format) || n_channels(format) < 3) throw except("esrgan: unsupported input image format [%d], expected an 8-bit colour image", int(format));
    device_turn turn(*m.backend);
    tile_layout tiles({{w, h}}, esrgan_default_tile_size, esrgan_tile_overlap);
    const int sc = m.params.scale;
    const size_t in_bytes = (size_t)batch * w * h * n_channels(format), out_bytes = (size_t)batch * w * sc * h * sc * 4;
    m.ws.img_in = std::max(m.ws.img_in, in_bytes);
    m.ws.img_out = std::max(m.ws.img_out, out_bytes);
    reserve(m, batch * tiles.total(), tiles.tile_size[0], tiles.tile_size[1]); // same layout as the call below computes
    void* s = m.backend->stream;
    VX(vx_memcpy_h2d(m.ws.in_u8, img, in_bytes, s));
    esrgan_compute_batch_device(m, m.ws.in_u8, batch, w, h, format, m.ws.out_u8, s);
    VX(vx_memcpy_d2h(out_rgba, m.ws.out_u8, out_bytes, s));
    VX(vx_stream_sync(s));
}

// reference src/visp/vision.cpp:220-253
image_data esrgan_compute(esrgan_model& m, image_view image) {
    if (is_float(image.format) || n_channels(image.format) < 3)
        throw except("esrgan: unsupported input image format [%d], expected an 8-bit colour image", int(image.format));
    const int ch = n_channels(image.format);
    const int w = image.extent[0], h = image.extent[1];
    std::vector<uint8_t> packed;
    const uint8_t* src = static_cast<const uint8_t*>(image.data);
    if (image.stride != 0 && image.stride != w * ch) { // drop row padding
        packed.resize((size_t)w * h * ch);
        for (int y = 0; y < h; ++y) memcpy(packed.data() + (size_t)y * w * ch, src + (size_t)y * image.stride, (size_t)w * ch);
        src = packed.data();
    }
    image_data out = image_alloc({{w * m.params.scale, h * m.params.scale}}, image_format::rgba_u8);
    esrgan_compute_batch_host(m, src, 1, w, h, image.format, out.data.get());
    return out;
}

void esrgan_generate_host(esrgan_model& m, float const* rgb, int n, int w, int h, float* out) {
    if (!m.weights_uploaded) throw except("esrgan: weights have not been uploaded");
    if (n < 1 || w < 1 || h < 1) throw except("esrgan: empty input");
    device_turn turn(*m.backend);
    reserve(m, n, w, h);
    const size_t px = (size_t)n * w * h;
    void* s = m.backend->stream;
    VX(vx_memcpy_h2d(m.ws.x0, rgb, px * 3 * 4, s));
    exec ex{m, s, {}, {}};
    run_tiles(m, ex, n, w, h);
    ex.finish_timing();
    const int sc = m.params.scale;
    VX(vx_memcpy_d2h(out, m.ws.tiles_out, px * sc * sc * 3 * 4, s));
    VX(vx_stream_sync(s));
}

} // namespace visp
