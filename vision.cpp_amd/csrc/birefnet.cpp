#include "birefnet.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/visp_hip_kernels.h"
#include "packer.h"
#include "visp_util.h"

namespace visp {

birefnet_model::~birefnet_model() {
    vx_free(dws.ptr);
    vx_free(dec_arena.ptr);
}

i32x2 birefnet_image_extent(i32x2 input_extent, birefnet_params const& p) { // birefnet.cpp:283-301
    if (p.image_size != -1) return {{p.image_size, p.image_size}};
    if (input_extent[0] <= 0 || input_extent[1] <= 0) throw except("birefnet: dynamic models need a positive input extent");
    auto next_multiple = [](int x, int m) { return (x + m - 1) / m * m; };
    return {{next_multiple(input_extent[0], p.image_multiple), next_multiple(input_extent[1], p.image_multiple)}};
}

namespace {

// one kernel of a conv tensor as stored in the file: OIHW when listed in conv2d_weights of a whcn file, else O H W I
struct conv_src {
    gguf_tensor const* w;
    bool oihw;
    int kw, kh, cin, cout;
    float at(int co, int ky, int kx, int c) const {
        const size_t i = oihw ? (((size_t)co * cin + c) * kh + ky) * kw + kx : (((size_t)co * kh + ky) * kw + kx) * cin + c;
        return tensor_at(*w, i);
    }
};
conv_src conv_of(packer& pk, std::string const& name) {
    gguf_tensor const& w = pk.get(name);
    conv_src s;
    s.w = &w;
    s.oihw = pk.file_whcn && pk.listed(name);
    s.cout = (int)w.ne[3];
    if (s.oihw) { s.kw = (int)w.ne[0]; s.kh = (int)w.ne[1]; s.cin = (int)w.ne[2]; }
    else { s.cin = (int)w.ne[0]; s.kw = (int)w.ne[1]; s.kh = (int)w.ne[2]; }
    if (s.kw != s.kh) throw except("tensor %s: non-square kernel", name.c_str());
    return s;
}
void set_bias(arena_builder& ab, packed_gemm& g, std::function<float(int)> at) {
    g.b = ab.alloc((size_t)g.N * 4);
    float* d = reinterpret_cast<float*>(ab.data.data() + g.b);
    for (int n = 0; n < g.n_real; ++n) d[n] = at(n);
}

// deformable_conv_2d + the branch's fused batch norm (birefnet.cpp:83-92, 110-115): offset and modulator convs become one GEMM
// (rows 0 .. 2k^2 - 1 offsets, then k^2 modulator logits); the kernel weights carry bn.weight, the GEMM bias is bn.bias
bf_deform_weights pack_deform(packer& pk, arena_builder& ab, std::string const& p) {
    bf_deform_weights d;
    conv_src off = conv_of(pk, p + ".conv.offset.weight"), mod = conv_of(pk, p + ".conv.modulator.weight"), cw = conv_of(pk, p + ".conv.conv.weight");
    const int k = cw.kw, taps = k * k, C = cw.cin;
    if (off.kw != k || mod.kw != k || off.cout != 2 * taps || mod.cout != taps || off.cin != C || mod.cin != C || !(k & 1) || k > 7)
        throw except("%s: deformable conv shapes do not match (kernel %d, offsets %d, modulator %d)", p.c_str(), k, off.cout, mod.cout);
    d.k = k;
    d.offmod = pk.matrix(3 * taps, taps * C, [&](int n, int kk) {
        const int t = kk / C, c = kk % C;
        return n < 2 * taps ? off.at(n, t / k, t % k, c) : mod.at(n - 2 * taps, t / k, t % k, c);
    }, nullptr);
    gguf_tensor const& ob = pk.get(p + ".conv.offset.bias");
    gguf_tensor const& mb = pk.get(p + ".conv.modulator.bias");
    set_bias(ab, d.offmod, [&](int n) { return n < 2 * taps ? tensor_at(ob, n) : tensor_at(mb, n - 2 * taps); });
    gguf_tensor const& bw = pk.get(p + ".bn.weight");
    gguf_tensor const& bb = pk.get(p + ".bn.bias");
    if (bw.n_elements() != cw.cout || bb.n_elements() != cw.cout) throw except("%s.bn: expected %d channels", p.c_str(), cw.cout);
    d.conv = pk.matrix(cw.cout, taps * C, [&](int n, int kk) { const int t = kk / C; return cw.at(n, t / k, t % k, kk % C) * tensor_at(bw, n); }, nullptr);
    set_bias(ab, d.conv, [&](int n) { return tensor_at(bb, n); });
    return d;
}

bf_block_weights pack_block(packer& pk, arena_builder& ab, std::string const& p) {
    bf_block_weights b;
    int k, cin;
    b.conv_in = pk.conv(p + ".conv_in", &k, &cin);
    if (k != 3) throw except("%s.conv_in: expected a 3x3 kernel", p.c_str());
    b.cin = cin;
    b.inter = b.conv_in.n_real;
    b.aspp[0] = pack_deform(pk, ab, p + ".dec_att.aspp1");
    for (int i = 0; i < 3; ++i) b.aspp[1 + i] = pack_deform(pk, ab, p + ".dec_att.aspp_deforms." + std::to_string(i));
    b.planes = b.aspp[0].conv.n_real;
    const int ks[4] = {1, 1, 3, 7};
    for (int i = 0; i < 4; ++i)
        if (b.aspp[i].k != ks[i] || b.aspp[i].conv.n_real != b.planes || b.aspp[i].conv.k_real != ks[i] * ks[i] * b.inter)
            throw except("%s.dec_att: branch %d has kernel %d / %d planes, expected %d / %d", p.c_str(), i, b.aspp[i].k, b.aspp[i].conv.n_real, ks[i], b.planes);
    b.gap = pk.conv(p + ".dec_att.global_avg_pool.1", &k, &cin);
    if (k != 1 || cin != b.inter || b.gap.n_real != b.planes) throw except("%s.dec_att.global_avg_pool.1: unexpected shape", p.c_str());
    b.conv1 = pk.conv(p + ".dec_att.conv1", &k, &cin);
    if (k != 1 || cin != 5 * b.planes || b.conv1.n_real != b.inter) throw except("%s.dec_att.conv1: unexpected shape", p.c_str());
    b.conv_out = pk.conv(p + ".conv_out", &k, &cin);
    if (k != 3 || cin != b.inter) throw except("%s.conv_out: unexpected shape", p.c_str());
    b.cout = b.conv_out.n_real;
    if (b.inter % 8 || b.planes % 8 || b.cin % 8 || b.cout % 8) throw except("%s: channel counts must be multiples of 8", p.c_str());
    return b;
}

size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

// launches of one pass; with dry = true nothing is launched and only the workspace high-water mark is recorded
struct bf_exec {
    birefnet_model& m;
    void* stream;
    const uint8_t* wa;
    bool dry;
    uint8_t* base = nullptr;
    size_t cur = 0, peak = 0;
    timing_marks tm;
    void* splitk = nullptr; // scratch of split-K convs (f32 partial sums)
    size_t splitk_bytes = 0;

    void* take(size_t bytes) {
        const size_t off = cur;
        cur = align_up(cur + bytes + 256); // slack: GEMM A rows are read up to K padded to 64
        peak = std::max(peak, cur);
        return dry ? nullptr : base + off;
    }
    void mark(const char* name, double flops, double bytes) { if (m.timing && !dry) tm.mark(name, flops, bytes, stream); }
    static void* off16(void* p, size_t elems) { return p ? static_cast<uint8_t*>(p) + elems * 2 : nullptr; } // f16 element offset
    void capture(const std::string& name, const void* src, int B, int h, int w, int C) { // parity-test hook: contiguous f16 [B, h, w, C]
        if (!m.captures || dry) return;
        capture_entry& c = m.capture_bufs[name];
        const size_t bytes = (size_t)B * h * w * C * 2;
        vx_free(c.dev);
        c.dev = nullptr;
        VX(vx_malloc(&c.dev, bytes));
        c.shape[0] = B; c.shape[1] = h; c.shape[2] = w; c.shape[3] = C;
        c.f16 = true;
        VX(vx_memcpy_d2d(c.dev, src, bytes, stream));
    }

    void gemm(packed_gemm const& g, const void* A, long M, int lda, void* out, int ldo, int epi, const void* res1, const char* group) {
        if (dry) return;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.lda = lda;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = (int)M; a.N = g.N; a.K = g.K; a.n_valid = (g.n_real + 7) / 8 * 8; // 1-, 3-, 27-, 147-column outputs: the pad columns (zero weights) land in the row's slack
        a.epi = epi; a.out = out; a.ldo = ldo; a.res1 = res1;
        mark(group, 2.0 * M * g.n_real * g.k_real, (double)M * (g.k_real + g.n_real) * 2);
        VX(vx_gemm_f16(&a, stream));
    }
    void conv(packed_gemm const& g, const void* x, int B, int H, int W, int Cpix, int k, int pad, void* out, int ldo, int epi, const char* group) {
        if (dry) return;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = x;
        a.conv_kh = a.conv_kw = k; a.conv_stride = 1; a.conv_pad = pad;
        a.conv_H = H; a.conv_W = W; a.conv_Cin = Cpix; a.conv_OH = H; a.conv_OW = W;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = B * H * W; a.N = g.N; a.K = g.K; a.n_valid = (g.n_real + 7) / 8 * 8; // 1-, 3-, 27-, 147-column outputs: the pad columns (zero weights) land in the row's slack
        a.epi = epi; a.out = out; a.ldo = ldo;
        // 3x3 convs over thousands of channels on 32 x 32 maps (squeeze / block4 conv_in: K = 25920 / 29160, 64 workgroups): split-K.
        // The split is chosen from the shape of ONE image, so a result never depends on the batch it was computed in.
        const int ks = vx_gemm_pick_k_splits(8 * H * W, g.N, g.K);
        if (ks > 1 && splitk && (size_t)ks * a.M * g.N * 4 <= splitk_bytes && (epi == VX_EPI_F16 || epi == VX_EPI_F16_RELU)) {
            a.k_splits = ks;
            a.k_partial = static_cast<float*>(splitk);
        }
        mark(group, 2.0 * a.M * g.n_real * g.k_real, (double)B * H * W * Cpix * 2 + (double)a.M * g.n_real * 2);
        VX(vx_gemm_f16(&a, stream));
    }
};

// basic_decoder_block (birefnet.cpp:144-150) on x [B, h, w, cin] (row stride = cin) -> y [B*h*w, ldy] (first cout channels)
void decoder_block(bf_exec& ex, bf_block_weights const& bw, const void* x, int B, int h, int w, void* y, int ldy) {
    const long M = (long)B * h * w;
    const size_t mark0 = ex.cur;
    void* a = ex.take((size_t)M * bw.inter * 2);
    ex.conv(bw.conv_in, x, B, h, w, bw.cin, 3, 1, a, bw.inter, VX_EPI_F16_RELU, "dec_conv3x3");
    const int ldc = 5 * bw.planes;
    void* cat = ex.take((size_t)M * ldc * 2);
    for (int j = 0; j < 4; ++j) { // aspp_module_deformable x 4 (birefnet.cpp:110-127)
        bf_deform_weights const& d = bw.aspp[j];
        const int k = d.k, taps = k * k, ldom = (3 * taps + 7) / 8 * 8;
        const size_t mark1 = ex.cur;
        void* om = ex.take((size_t)M * ldom * 2);
        if (k == 1) ex.gemm(d.offmod, a, M, bw.inter, om, ldom, VX_EPI_F16, nullptr, "deform_offsets");
        else ex.conv(d.offmod, a, B, h, w, bw.inter, k, k / 2, om, ldom, VX_EPI_F16, "deform_offsets");
        void* cols = ex.take((size_t)M * taps * bw.inter * 2);
        if (!ex.dry) {
            ex.mark("deform_sample", 0, (double)M * taps * bw.inter * 2 * 2);
            VX(vx_bf_deform_cols_f16(a, om, ldom, cols, B, h, w, bw.inter, k, ex.stream));
        }
        ex.gemm(d.conv, cols, M, taps * bw.inter, bf_exec::off16(cat, (size_t)j * bw.planes), ldc, VX_EPI_F16_RELU, nullptr, "deform_gemm");
        ex.cur = mark1;
    }
    { // global_avg_pool (birefnet.cpp:94-108, 128-133)
        void* mean = ex.take((size_t)std::max(B, 8) * bw.inter * 2 + 4096);
        void* g = ex.take((size_t)std::max(B, 8) * bw.planes * 2 + 4096);
        void* acc = ex.take((size_t)B * bw.inter * 4);
        if (!ex.dry) {
            ex.mark("dec_mean", 0, (double)M * bw.inter * 2);
            VX(vx_bf_mean_f16(a, bw.inter, mean, static_cast<float*>(acc), B, (int64_t)h * w, bw.inter, ex.stream));
        }
        ex.gemm(bw.gap, mean, B, bw.inter, g, bw.planes, VX_EPI_F16_RELU, nullptr, "dec_conv1x1");
        if (!ex.dry) {
            ex.mark("dec_broadcast", 0, (double)M * bw.planes * 2);
            VX(vx_bf_broadcast_f16(g, bw.planes, bf_exec::off16(cat, (size_t)4 * bw.planes), ldc, B, (int64_t)h * w, bw.planes, ex.stream));
        }
    }
    void* d = ex.take((size_t)M * bw.inter * 2);
    ex.gemm(bw.conv1, cat, M, ldc, d, bw.inter, VX_EPI_F16_RELU, nullptr, "dec_conv1x1");
    ex.conv(bw.conv_out, d, B, h, w, bw.inter, 3, 1, y, ldy, VX_EPI_F16, "dec_conv3x3");
    ex.cur = mark0;
}

} // namespace

birefnet_model* birefnet_load_model(char const* filepath, backend_device const& dev) {
    model_file file = model_load(filepath, /*header_only=*/false);
    if (file.arch() != "birefnet")
        throw except("Architecture expected to be 'birefnet', but was '%.*s' (%s)", (int)file.arch().size(), file.arch().data(), filepath); // birefnet.cpp:313-315
    auto model = std::make_unique<birefnet_model>();
    swin_load_into(*model, file, dev, "bb");
    model->bparams.image_size = file.get_int("birefnet.image_size");
    model->bparams.image_multiple = file.get_int("birefnet.image_multiple");
    model->bparams.image_extent = birefnet_image_extent({{1024, 1024}}, model->bparams); // vision.cpp:102
    arena_builder ab;
    packer pk{file, ab, true, file.tensor_layout() != layout_cwhn, file.conv2d_weights()};
    birefnet_weights& D = model->dec;
    const int C0 = model->params.embed_dim;
    D.squeeze = pack_block(pk, ab, "squeeze_module.0");
    if (D.squeeze.cin != 30 * C0 || D.squeeze.cout != 16 * C0) throw except("birefnet: squeeze block maps %d -> %d channels, expected %d -> %d", D.squeeze.cin, D.squeeze.cout, 30 * C0, 16 * C0);
    const std::string d = "decoder.";
    int carried = D.squeeze.cout; // channels of the map entering level 4, 3, 2, 1
    for (int i = 0; i < 5; ++i) {
        const int lvl = 5 - i, grid = i < 4 ? (32 >> i) : 1;
        const std::string p = d + "ipt_blk" + std::to_string(lvl);
        int k, cin;
        D.ipt[i].conv1 = pk.conv(p + ".conv1", &k, &cin, /*dup_in=*/lvl == 1);
        if (k != 3 || cin != 3 * grid * grid) throw except("%s.conv1: expected a 3x3 kernel over %d channels", p.c_str(), 3 * grid * grid);
        D.ipt[i].conv_out = pk.conv(p + ".conv_out", &k, &cin);
        if (k != 3 || cin != D.ipt[i].conv1.n_real) throw except("%s.conv_out: unexpected shape", p.c_str());
        D.ipt[i].cout = D.ipt[i].conv_out.n_real;
        if (D.ipt[i].cout % 8 || D.ipt[i].conv1.n_real % 8) throw except("%s: channel counts must be multiples of 8", p.c_str());
    }
    for (int i = 0; i < 4; ++i) {
        const int lvl = 4 - i;
        D.block[i] = pack_block(pk, ab, d + "block" + std::to_string(lvl));
        if (D.block[i].cin != carried + D.ipt[i].cout) throw except("birefnet: decoder.block%d takes %d channels, the graph supplies %d + %d", lvl, D.block[i].cin, carried, D.ipt[i].cout);
        carried = D.block[i].cout;
        if (i < 3) {
            int k, cin;
            D.lateral[i] = pk.conv(d + "lateral_block" + std::to_string(lvl) + ".conv", &k, &cin);
            if (k != 1 || D.lateral[i].n_real != carried || cin != (16 * C0 >> (i + 1))) throw except("birefnet: decoder.lateral_block%d has an unexpected shape", lvl);
            D.gdt[i] = pk.conv(d + "gdt_convs_" + std::to_string(lvl) + ".0", &k, &cin);
            if (k != 3 || cin != carried || D.gdt[i].n_real % 8) throw except("birefnet: decoder.gdt_convs_%d.0 has an unexpected shape", lvl);
            D.gdt_attn[i] = pk.conv(d + "gdt_convs_attn_" + std::to_string(lvl) + ".0", &k, &cin);
            if (k != 1 || cin != D.gdt[i].n_real || D.gdt_attn[i].n_real != 1) throw except("birefnet: decoder.gdt_convs_attn_%d.0 has an unexpected shape", lvl);
        }
    }
    {
        int k, cin;
        D.conv_out1 = pk.conv(d + "conv_out1.0", &k, &cin);
        if (k != 1 || D.conv_out1.n_real != 1 || cin != carried + D.ipt[4].cout) throw except("birefnet: decoder.conv_out1.0 has an unexpected shape");
    }
    device_turn turn(dev);
    model->dec_arena.bytes = round_up<size_t>(ab.data.size(), 256) + 4096;
    VX(vx_malloc(&model->dec_arena.ptr, model->dec_arena.bytes));
    VX(vx_memcpy_h2d(model->dec_arena.ptr, ab.data.data(), ab.data.size(), dev.stream));
    VX(vx_stream_sync(dev.stream));
    return model.release();
}

void birefnet_compute_batch_device(birefnet_model& m, void const* rgb_dev, int B, int w, int h, void* mask_dev, void* stream) {
    if (B < 1 || !rgb_dev || !mask_dev) throw except("birefnet: empty batch or null pointer");
    // the half-size pass needs extents that are multiples of 32 after the division by 2 (swin: patch 4, three even merges)
    if (w < 64 || h < 64 || w % 64 || h % 64) throw except("birefnet: image extent %dx%d must be a positive multiple of 64", w, h);
    device_turn turn(*m.backend);
    void* s = stream ? stream : m.backend->stream;
    birefnet_weights const& D = m.dec;
    const int C0 = m.params.embed_dim;
    int dims[4][3];
    swin_output_dims(m, w, h, dims);
    const int ccat[4] = {2 * C0, 4 * C0, 8 * C0, 30 * C0};
    const uint8_t* rgb = static_cast<const uint8_t*>(rgb_dev);
    std::vector<timing_entry> timing;

    auto pass = [&](bool dry) -> size_t {
        bf_exec ex{m, s, static_cast<const uint8_t*>(m.dec_arena.ptr), dry, nullptr, 0, 0, {}};
        ex.base = static_cast<uint8_t*>(m.dws.ptr);
        long Mi[4];
        void* F[4];
        ex.splitk_bytes = (size_t)8 * B * dims[2][0] * dims[2][1] * 128 * 4; // up to 8 ranges of a stage-2-sized map, 128 columns
        ex.splitk = ex.take(ex.splitk_bytes);
        for (int i = 0; i < 4; ++i) {
            Mi[i] = (long)B * dims[i][0] * dims[i][1];
            F[i] = ex.take((size_t)Mi[i] * ccat[i] * 2);
        }
        { // ---- birefnet::encode (birefnet.cpp:43-73)
            const size_t mark0 = ex.cur;
            void* low8 = ex.take((size_t)B * (w / 2) * (h / 2) * 8 * 2);
            void* L[4];
            for (int i = 0; i < 4; ++i) L[i] = ex.take((size_t)Mi[i] / 4 * dims[i][2] * 2);
            if (!dry) {
                swin_out so[4];
                // full resolution: stage i -> channels [0, C_i) of F[i]; for stage 3 behind the three downscaled finer stages
                const int off3 = ccat[0] + ccat[1] + ccat[2];
                for (int i = 0; i < 4; ++i) so[i] = {bf_exec::off16(F[i], i == 3 ? (size_t)off3 : 0), ccat[i], false};
                swin_encode_pixels(m, nullptr, B, w, h, so, s, rgb);
                if (m.timing) timing = m.last_timing;
                ex.mark("preprocess", 0, (double)B * w * h * 7);
                VX(vx_bf_preprocess_half(rgb, low8, B, h, w, s));
                for (int i = 0; i < 4; ++i) so[i] = {L[i], dims[i][2], false};
                if (m.timing) ex.tm.finish(s, timing, true);
                swin_encode_pixels(m, low8, B, w / 2, h / 2, so, s);
                if (m.timing)
                    for (timing_entry const& t : m.last_timing) {
                        auto it = std::find_if(timing.begin(), timing.end(), [&](timing_entry const& e) { return e.name == t.name; });
                        if (it == timing.end()) timing.push_back(t);
                        else { it->ms += t.ms; it->launches += t.launches; it->flops += t.flops; it->bytes += t.bytes; }
                    }
                // encode_concat: upscaled half-resolution features behind the full-resolution ones, then stages 0..2 scaled down into stage 3
                for (int i = 0; i < 4; ++i) {
                    ex.mark("resize", 0, (double)Mi[i] * dims[i][2] * 2.5);
                    VX(vx_bf_resize_f16(L[i], dims[i][2], bf_exec::off16(F[i], (size_t)(i == 3 ? off3 : 0) + dims[i][2]), ccat[i], B, dims[i][1] / 2, dims[i][0] / 2, dims[i][2],
                                        dims[i][1], dims[i][0], s));
                }
                int off = 0;
                for (int i = 0; i < 3; ++i) {
                    ex.mark("resize", 0, (double)Mi[i] * ccat[i] * 2);
                    VX(vx_bf_resize_f16(F[i], ccat[i], bf_exec::off16(F[3], (size_t)off), ccat[3], B, dims[i][1], dims[i][0], ccat[i], dims[3][1], dims[3][0], s));
                    off += ccat[i];
                }
            }
            ex.cur = mark0;
        }
        for (int i = 0; i < 4; ++i) ex.capture("feature_" + std::to_string(i), F[i], B, dims[i][1], dims[i][0], ccat[i]);
        // ---- squeeze block, decoder (birefnet.cpp:170-250)
        void* carried = ex.take((size_t)Mi[3] * D.squeeze.cout * 2);
        decoder_block(ex, D.squeeze, F[3], B, dims[3][1], dims[3][0], carried, D.squeeze.cout);
        ex.capture("squeeze", carried, B, dims[3][1], dims[3][0], D.squeeze.cout);
        int cc = D.squeeze.cout, pw = dims[3][0], ph = dims[3][1]; // carried map: channels and extent
        for (int i = 0; i < 4; ++i) { // level 4 - i at the extent of encoder stage 3 - i
            const int st = 3 - i, lw = dims[st][0], lh = dims[st][1];
            const long M = Mi[st];
            bf_block_weights const& bw = D.block[i];
            void* X = ex.take((size_t)M * bw.cin * 2);
            if (i == 0) {
                if (!dry) { // x4 = concat(squeeze output, ipt_blk5(patches))
                    ex.mark("dec_glue", 0, (double)M * cc * 4);
                    VX(vx_bf_resize_f16(carried, cc, X, bw.cin, B, ph, pw, cc, lh, lw, s)); // same extent: a strided copy
                }
            } else {
                if (!dry) { // upscale_to(p, x_st) + lateral conv (birefnet.cpp:194-198): the GEMM adds its output onto the upscaled map in place
                    ex.mark("resize", 0, (double)M * cc * 2.5);
                    VX(vx_bf_resize_f16(carried, cc, X, bw.cin, B, ph, pw, cc, lh, lw, s));
                }
                ex.gemm(D.lateral[i - 1], F[st], M, ccat[st], X, bw.cin, VX_EPI_F16_ADD, X, "dec_conv1x1");
            }
            { // ipt block on image_to_patches (birefnet.cpp:152-167, 176-181)
                const size_t mark1 = ex.cur;
                const int grid = 32 >> i, cp = 3 * grid * grid, c1 = D.ipt[i].conv1.n_real;
                void* patches = ex.take((size_t)M * cp * 2);
                void* t = ex.take((size_t)M * c1 * 2);
                if (!dry) {
                    ex.mark("dec_patches", 0, (double)B * w * h * 3 + (double)M * cp * 2);
                    VX(vx_bf_patches(rgb, patches, B, h, w, lh, lw, s));
                }
                ex.conv(D.ipt[i].conv1, patches, B, lh, lw, cp, 3, 1, t, c1, VX_EPI_F16, "ipt_conv3x3");
                ex.conv(D.ipt[i].conv_out, t, B, lh, lw, c1, 3, 1, bf_exec::off16(X, (size_t)cc), bw.cin, VX_EPI_F16, "ipt_conv3x3");
                ex.cur = mark1;
            }
            void* y = ex.take((size_t)M * bw.cout * 2);
            decoder_block(ex, bw, X, B, lh, lw, y, bw.cout);
            if (i < 3) { // gdt attention (birefnet.cpp:183-192)
                const size_t mark1 = ex.cur;
                const int cg = D.gdt[i].n_real;
                void* g = ex.take((size_t)M * cg * 2);
                void* a = ex.take((size_t)M * 8 * 2);
                ex.conv(D.gdt[i], y, B, lh, lw, bw.cout, 3, 1, g, cg, VX_EPI_F16_RELU, "dec_conv3x3");
                ex.gemm(D.gdt_attn[i], g, M, cg, a, 8, VX_EPI_F16, nullptr, "dec_conv1x1");
                if (!dry) {
                    ex.mark("dec_gdt_mul", 0, (double)M * bw.cout * 4);
                    VX(vx_bf_mul_sigmoid_f16(y, bw.cout, a, 8, M, bw.cout, s));
                }
                ex.cur = mark1;
            }
            ex.capture("p" + std::to_string(4 - i), y, B, lh, lw, bw.cout);
            carried = y;
            cc = bw.cout; pw = lw; ph = lh;
        }
        { // _p1 upscaled to the image, ipt_blk1 on the image itself, conv_out1 + sigmoid (birefnet.cpp:238-247)
            const long M = (long)B * w * h;
            const int c1 = D.ipt[4].conv1.n_real, ci = D.ipt[4].cout, ct = cc + ci;
            void* X = ex.take((size_t)M * ct * 2);
            void* in8 = ex.take((size_t)M * 8 * 2);
            void* t = ex.take((size_t)M * c1 * 2);
            void* a = ex.take((size_t)M * 8 * 2);
            if (!dry) {
                ex.mark("resize", 0, (double)M * cc * 2);
                VX(vx_bf_resize_f16(carried, cc, X, ct, B, ph, pw, cc, h, w, s));
                ex.mark("preprocess", 0, (double)M * 19);
                VX(vx_tv_preprocess(rgb, in8, (int64_t)M, s));
            }
            ex.conv(D.ipt[4].conv1, in8, B, h, w, 8, 3, 1, t, c1, VX_EPI_F16, "ipt_conv3x3");
            ex.conv(D.ipt[4].conv_out, t, B, h, w, c1, 3, 1, bf_exec::off16(X, (size_t)cc), ct, VX_EPI_F16, "ipt_conv3x3");
            ex.gemm(D.conv_out1, X, M, ct, a, 8, VX_EPI_F16, nullptr, "dec_conv1x1");
            if (!dry) {
                ex.mark("mask_out", 0, (double)M * 6);
                VX(vx_bf_sigmoid_out_f32(a, 8, static_cast<float*>(mask_dev), M, s));
            }
        }
        if (m.timing && !dry) { ex.tm.finish(s, timing, true); m.last_timing = timing; }
        return ex.peak;
    };

    const size_t need = pass(true) + 4096;
    if (m.dws.bytes < need) {
        VX(vx_stream_sync(m.backend->stream));
        VX(vx_stream_sync(s));
        vx_free(m.dws.ptr);
        m.dws = {};
        VX(vx_malloc(&m.dws.ptr, need));
        m.dws.bytes = need;
        VX(vx_memset(m.dws.ptr, 0, need, s)); // GEMM rows are read up to K padded to 64: stale bytes must be finite
    }
    pass(false);
    if (!stream) VX(vx_stream_sync(s));
}

void birefnet_compute_batch_host(birefnet_model& m, uint8_t const* rgb, int B, int w, int h, float* mask) {
    if (B < 1 || !rgb || !mask) throw except("birefnet: empty batch or null pointer");
    device_turn turn(*m.backend);
    void* s = m.backend->stream;
    void *in = nullptr, *out = nullptr;
    auto release = [&]() { vx_free(in); vx_free(out); };
    try {
        VX(vx_malloc(&in, (size_t)B * w * h * 3));
        VX(vx_malloc(&out, (size_t)B * w * h * 4));
        VX(vx_memcpy_h2d(in, rgb, (size_t)B * w * h * 3, s));
        birefnet_compute_batch_device(m, in, B, w, h, out, s);
        VX(vx_memcpy_d2h(mask, out, (size_t)B * w * h * 4, s));
        VX(vx_stream_sync(s));
    } catch (...) {
        release();
        throw;
    }
    release();
}

image_data birefnet_compute(birefnet_model& m, image_view image) { // vision.cpp:108-132
    if (is_float(image.format) || n_channels(image.format) < 3)
        throw except("birefnet: unsupported input image format [%d], expected an 8-bit colour image", int(image.format));
    const i32x2 res = birefnet_image_extent(image.extent, m.bparams);
    m.bparams.image_extent = res;
    // birefnet_process_input (birefnet.cpp:262-266) happens in the caller's format -- for rgba / bgra / argb stb resizes alpha-weighted (premultiplied),
    // so colours next to transparent pixels differ from a resize of the opaque rgb -- and the alpha channel is dropped afterwards
    // (image_u8_to_f32 to rgb_f32 in the reference)
    const i32x2 caller_extent = image.extent;
    image_data resized;
    if (image.extent != res) {
        resized = image_scale(image, res);
        image = view_of(resized);
    }
    image_data rgb = image_to_rgb_u8(image);
    image_view rgb_view = view_of(rgb);
    image_data mask = image_alloc(res, image_format::alpha_f32);
    birefnet_compute_batch_host(m, static_cast<const uint8_t*>(rgb_view.data), 1, res[0], res[1], reinterpret_cast<float*>(mask.data.get()));
    if (res != caller_extent) { // birefnet_process_output (birefnet.cpp:272-281)
        image_data scaled = image_scale(view_of(mask), caller_extent);
        return image_f32_to_u8(view_of(scaled), image_format::alpha_u8);
    }
    return image_f32_to_u8(view_of(mask), image_format::alpha_u8);
}

} // namespace visp
