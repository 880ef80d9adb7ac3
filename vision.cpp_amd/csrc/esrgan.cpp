#include "esrgan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "../../include/visp_hip_kernels.h"
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
// weight packing

struct arena_builder {
    std::vector<uint8_t> data;
    size_t alloc(size_t bytes) {
        size_t off = round_up<size_t>(data.size(), 256);
        data.resize(off + bytes, 0);
        return off;
    }
};

float tensor_at(gguf_tensor const& t, size_t i) {
    if (t.type == GGML_F32) return reinterpret_cast<const float*>(t.data)[i];
    return f16_to_f32(reinterpret_cast<const uint16_t*>(t.data)[i]);
}

struct packer {
    model_file const& file;
    arena_builder& ab;
    bool with_data;
    bool file_cwhn;

    // conv kernel `name`.weight -> slabs [cin_pad/32][9][cout_pad][32] f16, the four 16-byte groups of every
    // (tap, n) row stored at g ^ ((n >> 2) & 3) so that the kernel's linear LDS-DMA copy lands bank-conflict free
    // (kernels_dconv.hip). dup_in > 0: input channels [dup_in, 2*dup_in) repeat [0, dup_in) (the first conv reads
    // the image as value + f16 rounding residue).
    packed_dconv conv(std::string const& name, int dup_in = 0) {
        gguf_tensor const& t = file.tensor(name + ".weight");
        if (t.type != GGML_F32 && t.type != GGML_F16) throw except("tensor %s: unsupported type %d", t.name.c_str(), t.type);
        // whcn file: ne = [kw, kh, Cin, Cout] (torch OIHW); cwhn file: ne = [Cin, kw, kh, Cout] (OHWI, convert.py:120-125)
        int kw, kh, cin, cout = (int)t.ne[3];
        if (file_cwhn) { cin = (int)t.ne[0]; kw = (int)t.ne[1]; kh = (int)t.ne[2]; }
        else { kw = (int)t.ne[0]; kh = (int)t.ne[1]; cin = (int)t.ne[2]; }
        if (kw != 3 || kh != 3) throw except("tensor %s: expected a 3x3 kernel, got %dx%d", t.name.c_str(), kw, kh);
        packed_dconv g;
        g.cin_real = cin;
        g.cout_real = cout;
        g.cin = round_up(dup_in ? 2 * dup_in : cin, 32);
        g.cout = round_up(cout, 32);
        g.w = ab.alloc((size_t)g.cin * 9 * g.cout * 2);
        g.b = ab.alloc((size_t)g.cout * 4);
        if (!with_data) return g;
        if (!t.data) throw except("tensor %s has no data", t.name.c_str());
        uint16_t* dst = reinterpret_cast<uint16_t*>(ab.data.data() + g.w);
        const int n_in = dup_in ? 2 * dup_in : cin;
        for (int c = 0; c < n_in; ++c) {
            const int cs = dup_in ? c % dup_in : c; // source input channel
            const int chunk = c / 32, grp = (c % 32) / 8, e = c % 8;
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                for (int n = 0; n < cout; ++n) {
                    const size_t src = file_cwhn ? (((size_t)n * 3 + ky) * 3 + kx) * cin + cs : (((size_t)n * cin + cs) * 3 + ky) * 3 + kx;
                    const size_t row = ((size_t)chunk * 9 + tap) * g.cout + n;
                    dst[row * 32 + (size_t)(grp ^ ((n >> 2) & 3)) * 8 + e] = f32_to_f16(tensor_at(t, src));
                }
            }
        }
        if (const gguf_tensor* bt = file.find(name + ".bias")) {
            if ((int)bt->n_elements() != cout) throw except("tensor %s: %d elements, expected %d", bt->name.c_str(), (int)bt->n_elements(), cout);
            float* bd = reinterpret_cast<float*>(ab.data.data() + g.b);
            for (int n = 0; n < cout; ++n) bd[n] = tensor_at(*bt, n);
        }
        return g;
    }
};

} // namespace

esrgan_model* esrgan_load_model(char const* filepath, backend_device const& dev, int flags) {
    const bool with_data = !(flags & load_no_upload);
    model_file file = model_load(filepath, /*header_only=*/!with_data);
    auto model = std::make_unique<esrgan_model>();
    model->backend = &dev;
    model->params = esrgan_detect_params(file);
    esrgan_params const& P = model->params;
    if (P.scale != 1 && P.scale != 2 && P.scale != 4 && P.scale != 8)
        throw except("ESRGAN: scale %d is not built in this backend (powers of two only)", P.scale);

    arena_builder ab;
    packer pk{file, ab, with_data, file.tensor_layout() == layout_cwhn};
    esrgan_weights& Wt = model->weights;
    Wt.first = pk.conv("model.0", 3);
    if (Wt.first.cin_real != 3) throw except("ESRGAN: model.0 takes %d input channels, expected 3", Wt.first.cin_real);
    Wt.nf = Wt.first.cout_real;
    Wt.rdb.resize(P.n_blocks);
    for (int i = 0; i < P.n_blocks; ++i)
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 5; ++k)
                Wt.rdb[i][r][k] = pk.conv("model.1.sub." + std::to_string(i) + ".RDB" + std::to_string(r + 1) + ".conv" + std::to_string(k + 1) + ".0");
    Wt.gc = Wt.rdb[0][0][0].cout_real;
    if (Wt.nf != 64 || Wt.gc != 32)
        throw except("ESRGAN: %d filters / %d growth channels are not built in this backend (64 / 32 only)", Wt.nf, Wt.gc);
    for (auto& blk : Wt.rdb)
        for (auto& rd : blk)
            for (int k = 0; k < 5; ++k)
                if (rd[k].cin_real != Wt.nf + k * Wt.gc || rd[k].cout_real != (k < 4 ? Wt.gc : Wt.nf))
                    throw except("ESRGAN: dense block conv%d has shape %d -> %d", k + 1, rd[k].cin_real, rd[k].cout_real);
    Wt.trunk = pk.conv("model.1.sub." + std::to_string(P.n_blocks));
    int seq = 2;
    for (int i = 0; i < log2_floor(P.scale); ++i) {
        Wt.up.push_back(pk.conv("model." + std::to_string(seq + 1)));
        seq += 3;
    }
    Wt.hr = pk.conv("model." + std::to_string(seq));
    Wt.last = pk.conv("model." + std::to_string(seq + 2));
    if (Wt.last.cout_real != 3) throw except("ESRGAN: the last conv has %d outputs, expected 3", Wt.last.cout_real);
    for (packed_dconv const* g : {&Wt.trunk, &Wt.hr})
        if (g->cin_real != Wt.nf || g->cout_real != Wt.nf) throw except("ESRGAN: trunk/HR conv has shape %d -> %d", g->cin_real, g->cout_real);

    device_turn turn(dev);
    VX(vx_dconv_prepare());
    model->weight_arena.bytes = round_up<size_t>(ab.data.size(), 256);
    VX(vx_malloc(&model->weight_arena.ptr, model->weight_arena.bytes));
    if (with_data) {
        VX(vx_memcpy_h2d(model->weight_arena.ptr, ab.data.data(), ab.data.size(), dev.stream));
        VX(vx_stream_sync(dev.stream));
        model->weights_uploaded = true;
    }
    if (const char* e = getenv("VISP_ESRGAN_TILE_GROUP")) model->tile_group = std::max(1, atoi(e));
    if (const char* e = getenv("VISP_ESRGAN_STREAMS")) model->streams = std::max(1, atoi(e));
    VX(vx_stream_create(&model->aux_stream));
    VX(vx_event_create(&model->fork_event));
    VX(vx_event_create(&model->join_event));
    return model.release();
}

void esrgan_weights_ready(esrgan_model& m) { m.weights_uploaded = true; }

esrgan_model::~esrgan_model() {
    if (aux_stream) {
        vx_stream_sync(aux_stream);
        vx_stream_destroy(aux_stream);
    }
    if (fork_event) vx_event_destroy(fork_event);
    if (join_event) vx_event_destroy(join_event);
    vx_free(ws.arena.ptr);
    vx_free(weight_arena.ptr);
}

//
// workspace + executor

namespace {

// ws.img_in / ws.img_out (capacity for the host entry point's u8 staging images) are kept, everything else is sized
// for this call
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
    const size_t hr_bytes = m.weights.up.empty() ? (size_t)group * px * 64 * 2 : (size_t)group * px * s * s * 64 * 2;
    std::vector<item> items = {
        {&ws.in_u8, img_in_bytes}, {&ws.out_u8, img_out_bytes},
        {&ws.x0, (size_t)n_tiles_total * px * 32 * 2}, {&ws.tiles_out, (size_t)n_tiles_total * px * s * s * 3 * 4}};
    for (int l = 0; l < lanes; ++l) {
        esrgan_workspace::lane_buffers& L = ws.lane[l];
        items.push_back({&L.fea, (size_t)group * px * 64 * 2});
        for (int k = 0; k < 3; ++k) items.push_back({&L.d[k], (size_t)group * px * 192 * 2});
        items.push_back({&L.tr, (size_t)group * px * 64 * 2});
        items.push_back({&L.hr_a, hr_bytes});
        items.push_back({&L.hr_b, hr_bytes});
    }
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

struct exec {
    esrgan_model& m;
    void* stream;
    const uint8_t* wa;
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

    struct opts {
        bool up2 = false, lrelu = false, rgb = false, x_residual = false;
        float s1 = 1, s2 = 1;
        const void* res1 = nullptr; int64_t res1_plane = 0;
        const void* res2 = nullptr; int64_t res2_plane = 0;
    };
    // x: cin/32 planes of [n, H(/2), W(/2), 32] f16, x_plane elements apart; out: planes of [n, H, W, 32] or f32 rgb
    void conv(packed_dconv const& g, const void* x, int64_t x_plane, int cin, int n, int H, int W, void* out, int64_t out_plane, opts const& o,
              const char* group) {
        vx_dconv_args a;
        memset(&a, 0, sizeof a);
        a.x = x; a.x_plane = x_plane; a.cin = cin; a.up2 = o.up2;
        a.B = n; a.H = H; a.W = W;
        a.w = wa + g.w; a.bias = reinterpret_cast<const float*>(wa + g.b); a.cout = g.cout;
        a.epi = o.rgb ? VX_DC_RGB_F32 : VX_DC_F16;
        a.act = o.lrelu;
        a.s1 = o.s1; a.res1 = o.res1; a.res1_plane = o.res1_plane;
        a.s2 = o.s2; a.res2 = o.res2; a.res2_plane = o.res2_plane;
        a.out = out; a.out_plane = out_plane;
        a.x_residual = o.x_residual;
        const double px = (double)n * H * W;
        mark(group, 2.0 * px * 9 * g.cin_real * g.cout_real,
             px * (o.up2 ? 0.25 : 1.0) * cin * 2 + px * (o.rgb ? 12 : g.cout * 2) + (o.res1 ? px * g.cout * 2 : 0) + (o.res2 ? px * g.cout * 2 : 0));
        VX(vx_dconv3x3_f16(&a, stream));
    }

    // esrgan_generate (esrgan.cpp:55-79) on n tiles: x0 [n,h,w,32] f16 -> rgb f32 [n, h*s, w*s, 3]
    void generate(esrgan_workspace::lane_buffers const& lb, const void* x0, int n, int w, int h, float* out) {
        esrgan_weights const& Wt = m.weights;
        esrgan_workspace const& ws = m.ws;
        // every activation buffer is planar (32 channels per plane); plane strides are those of the full tile group
        const int64_t PL = (int64_t)ws.group * w * h * 32;             // low-resolution plane
        const int64_t PH = PL * ws.scale * ws.scale;                    // plane of the up-sampled maps
        auto plane = [](void* base, int64_t stride, int k) { return static_cast<void*>(static_cast<uint16_t*>(base) + stride * k); };
        conv(Wt.first, x0, 0, 32, n, h, w, lb.fea, PL, {}, "first");
        conv(Wt.first, x0, 0, 32, n, h, w, lb.d[0], PL, {}, "first");
        int a = 0;
        for (auto const& blk : Wt.rdb) { // rrdb, esrgan.cpp:43-51
            void *A = lb.d[a], *B = lb.d[(a + 1) % 3], *C = lb.d[(a + 2) % 3];
            void* src[3] = {A, B, C};
            void* dst[3] = {B, C, B};
            for (int r = 0; r < 3; ++r) { // risidual_dense_block, esrgan.cpp:27-41
                for (int k = 0; k < 4; ++k) {
                    opts o;
                    o.lrelu = true;
                    static const char* const names[4] = {"rdb_conv1", "rdb_conv2", "rdb_conv3", "rdb_conv4"};
                    conv(blk[r][k], src[r], PL, 64 + 32 * k, n, h, w, plane(src[r], PL, 2 + k), PL, o, names[k]);
                }
                opts o;
                o.s1 = 0.2f; o.x_residual = true; // x5*0.2 + x, x = planes 0,1 of the halo the conv already holds
                if (r == 2) { o.s2 = 0.2f; o.res2 = A; o.res2_plane = PL; }
                conv(blk[r][4], src[r], PL, 192, n, h, w, dst[r], PL, o, "rdb_conv5");
            }
            a = (a + 1) % 3;
        }
        {
            opts o;
            o.res1 = lb.fea; o.res1_plane = PL;
            conv(Wt.trunk, lb.d[a], PL, 64, n, h, w, lb.tr, PL, o, "trunk");
        }
        const void* cur = lb.tr;
        int64_t cur_plane = PL;
        void* nxt[2] = {lb.hr_a, lb.hr_b};
        int flip = 0, cw = w, ch = h;
        for (packed_dconv const& u : Wt.up) { // esrgan::upsample, esrgan.cpp:13-19
            cw *= 2; ch *= 2;
            opts o;
            o.up2 = true; o.lrelu = true;
            conv(u, cur, cur_plane, 64, n, ch, cw, nxt[flip], PH, o, "upconv");
            cur = nxt[flip];
            cur_plane = PH;
            flip ^= 1;
        }
        {
            opts o;
            o.lrelu = true;
            conv(Wt.hr, cur, cur_plane, 64, n, ch, cw, nxt[flip], PH, o, "hrconv");
            opts l;
            l.rgb = true;
            conv(Wt.last, nxt[flip], PH, 64, n, ch, cw, out, 0, l, "last");
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
        ex.generate(m.ws.lane[lane], static_cast<const uint16_t*>(m.ws.x0) + (size_t)t0 * px * 32, n, tw, th,
                    static_cast<float*>(m.ws.tiles_out) + (size_t)t0 * px * s * s * 3);
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
    exec ex{m, s, static_cast<const uint8_t*>(m.weight_arena.ptr), {}, {}};
    vx_tile_layout tin = to_vx(tiles), tout = to_vx(tiles_out);
    ex.mark("tiles_in", 0, (double)n_total * tiles.tile_size[0] * tiles.tile_size[1] * 67);
    VX(vx_esrgan_tiles_in(static_cast<const uint8_t*>(img_dev), batch, w, h, int(format), &tin, m.ws.x0, s));
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
    std::vector<uint16_t> x0(px * 32, 0);
    for (size_t i = 0; i < px; ++i)
        for (int c = 0; c < 3; ++c) {
            const float v = rgb[i * 3 + c];
            const uint16_t hi = f32_to_f16(v);
            x0[i * 32 + c] = hi;
            x0[i * 32 + 3 + c] = f32_to_f16(v - f16_to_f32(hi));
        }
    void* s = m.backend->stream;
    VX(vx_memcpy_h2d(m.ws.x0, x0.data(), x0.size() * 2, s));
    exec ex{m, s, static_cast<const uint8_t*>(m.weight_arena.ptr), {}, {}};
    run_tiles(m, ex, n, w, h);
    ex.finish_timing();
    const int sc = m.params.scale;
    VX(vx_memcpy_d2h(out, m.ws.tiles_out, px * sc * sc * 3 * 4, s));
    VX(vx_stream_sync(s));
}

} // namespace visp
