#include "swin.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/visp_hip_kernels.h"
#include "packer.h"
#include "visp_util.h"

namespace visp {

swin_params swin_detect_params(model_file const& f) { // swin.cpp:293-302
    const int embed_dim = f.get_int("swin.embed_dim");
    swin_params p;
    if (embed_dim == 96) return p; // swin_t_params
    if (embed_dim == 192) {        // swin_l_params
        p.embed_dim = 192;
        p.window_size = 12;
        const swin_layer_t l[4] = {{2, 6, 192}, {2, 12, 384}, {18, 24, 768}, {2, 48, 1536}};
        std::copy(l, l + 4, p.layers);
        return p;
    }
    // files written by this repo's tests carry their layer table (vision.cpp_amd/synth.py write_swin_gguf); the reference knows
    // the two configurations above only
    if (!f.find_key("swin.depths") || !f.find_key("swin.n_heads") || !f.find_key("swin.window_size"))
        throw except("Unsupported Swin Transformer embed dim: %d", embed_dim);
    int depths[4], heads[4];
    f.get_array("swin.depths", depths, 4);
    f.get_array("swin.n_heads", heads, 4);
    p.embed_dim = embed_dim;
    p.window_size = f.get_int("swin.window_size");
    for (int i = 0; i < 4; ++i) p.layers[i] = {depths[i], heads[i], embed_dim << i};
    return p;
}

swin_model::~swin_model() {
    vx_free(ws.ptr);
    for (auto& c : capture_bufs) vx_free(c.second.dev);
    vx_free(weight_arena.ptr);
}

swin_model* swin_load_model(char const* filepath, backend_device const& dev, char const* prefix) {
    model_file file = model_load(filepath, /*header_only=*/false);
    if (file.arch() != "birefnet")
        throw except("Architecture expected to be 'birefnet', but was '%.*s' (%s)", (int)file.arch().size(), file.arch().data(), filepath); // birefnet.cpp:313-315
    auto model = std::make_unique<swin_model>();
    swin_load_into(*model, file, dev, prefix);
    return model.release();
}

void swin_load_into(swin_model& mdl, model_file const& file, backend_device const& dev, char const* prefix) {
    swin_model* const model = &mdl;
    model->backend = &dev;
    model->params = swin_detect_params(file);
    model->mask_shifted_only = getenv("VISP_SWIN_MASK_SHIFTED_ONLY") != nullptr && atoi(getenv("VISP_SWIN_MASK_SHIFTED_ONLY")) != 0;
    swin_params const& P = model->params;
    arena_builder ab;
    packer pk{file, ab, true, file.tensor_layout() != layout_cwhn, file.conv2d_weights()};
    swin_weights& Wt = model->weights;
    const std::string e = std::string(prefix) + ".";
    const int ws = P.window_size, N = ws * ws;
    if (ws > 16 || N > 256) throw except("swin: window size %d is not supported by this backend (at most 16)", ws);

    int k, cin;
    Wt.patch_embed = pk.conv(e + "patch_embed.proj", &k, &cin, /*dup_in=*/true); // the converter always stores it NHWC (convert.py:415-416)
    if (k != 4) throw except("swin: patch embedding kernel is %dx%d, expected 4x4", k, k);
    if (Wt.patch_embed.n_real != P.embed_dim) throw except("swin: patch embedding width %d, expected %d", Wt.patch_embed.n_real, P.embed_dim);
    Wt.pe_norm = file.find(e + "patch_embed.norm.weight") != nullptr; // nn.cpp:173-178
    if (Wt.pe_norm) {
        Wt.pe_norm_w = pk.vec(e + "patch_embed.norm.weight");
        Wt.pe_norm_b = pk.vec(e + "patch_embed.norm.bias");
    }
    for (int l = 0; l < 4; ++l) {
        swin_layer_t const& L = P.layers[l];
        const int C = L.n_features, heads = L.n_heads;
        if (C != (P.embed_dim << l) || heads <= 0 || C != heads * 32)
            throw except("swin: layer %d has %d features in %d heads; this backend implements head_dim 32", l, C, heads);
        for (int i = 0; i < L.depth; ++i) {
            const std::string p = e + "layers." + std::to_string(l) + ".blocks." + std::to_string(i);
            swin_block_weights b;
            b.norm1_w = pk.vec(p + ".norm1.weight");
            b.norm1_b = pk.vec(p + ".norm1.bias");
            b.norm2_w = pk.vec(p + ".norm2.weight");
            b.norm2_b = pk.vec(p + ".norm2.bias");
            { // qkv: torch rows are [3][heads][32] (split_qkv dim 2, nn.cpp:191-194); the attention kernel reads per head q | k | v
                gguf_tensor const& w = pk.get(p + ".attn.qkv.weight");
                gguf_tensor const* bias = file.find(p + ".attn.qkv.bias");
                if (w.ne[0] != C || w.ne[1] != 3 * C) throw except("tensor %s.attn.qkv.weight: expected [%d, %d]", p.c_str(), 3 * C, C);
                auto src_row = [C](int n) { const int h = n / 96, r = n % 96, part = r / 32, d = r % 32; return part * C + h * 32 + d; };
                b.qkv = pk.matrix(3 * C, C, [&](int n, int kk) { return tensor_at(w, (size_t)src_row(n) * C + kk); }, nullptr);
                if (bias) {
                    if (bias->n_elements() != 3 * C) throw except("tensor %s.attn.qkv.bias: %lld elements, expected %d", p.c_str(), (long long)bias->n_elements(), 3 * C);
                    b.qkv.b = ab.alloc((size_t)b.qkv.N * 4);
                    float* d = reinterpret_cast<float*>(ab.data.data() + b.qkv.b);
                    for (int n = 0; n < 3 * C; ++n) d[n] = tensor_at(*bias, src_row(n));
                }
            }
            {
                gguf_tensor const& t = pk.get(p + ".attn.relative_position_bias_table");
                const int64_t rows = (int64_t)(2 * ws - 1) * (2 * ws - 1);
                if (t.ne[0] != heads || t.ne[1] != rows) throw except("tensor %s.attn.relative_position_bias_table: expected [%lld, %d]", p.c_str(), (long long)rows, heads);
                std::vector<float> tb((size_t)rows * heads);
                for (size_t j = 0; j < tb.size(); ++j) tb[j] = tensor_at(t, j);
                const size_t bytes = 4 * vx_window_attention_bias_bytes(N, heads);
                b.bias.n = (int)(bytes / 2);
                b.bias.off = ab.alloc(bytes);
                VX(vx_swin_attention_pack_bias(tb.data(), ws, heads, ab.data.data() + b.bias.off));
            }
            b.proj = pk.linear(p + ".attn.proj");
            b.fc1 = pk.linear(p + ".mlp.fc1");
            b.fc2 = pk.linear(p + ".mlp.fc2");
            Wt.blocks[l].push_back(b);
        }
        if (l < 3) {
            const std::string p = e + "layers." + std::to_string(l) + ".downsample";
            Wt.merge_norm_w[l] = pk.vec(p + ".norm.weight");
            Wt.merge_norm_b[l] = pk.vec(p + ".norm.bias");
            Wt.merge_reduction[l] = pk.linear(p + ".reduction");
            if (Wt.merge_reduction[l].k_real != 4 * C || Wt.merge_reduction[l].n_real != 2 * C)
                throw except("swin: %s.reduction is [%d, %d], expected [%d, %d]", p.c_str(), Wt.merge_reduction[l].n_real, Wt.merge_reduction[l].k_real, 2 * C, 4 * C);
        }
        Wt.out_norm_w[l] = pk.vec(e + "norm" + std::to_string(l) + ".weight");
        Wt.out_norm_b[l] = pk.vec(e + "norm" + std::to_string(l) + ".bias");
    }
    device_turn turn(dev);
    model->weight_arena.bytes = round_up<size_t>(ab.data.size(), 256) + 4096;
    VX(vx_malloc(&model->weight_arena.ptr, model->weight_arena.bytes));
    VX(vx_memcpy_h2d(model->weight_arena.ptr, ab.data.data(), ab.data.size(), dev.stream));
    VX(vx_stream_sync(dev.stream));
}

void swin_output_dims(swin_model const& m, int w, int h, int dims[4][3]) {
    int cw = w / 4, ch = h / 4;
    for (int l = 0; l < 4; ++l) {
        dims[l][0] = cw; dims[l][1] = ch; dims[l][2] = m.params.layers[l].n_features;
        cw = (cw + 1) / 2; ch = (ch + 1) / 2; // swin.cpp:233
    }
}

void timing_marks::mark(const char* name, double flops, double bytes, void* stream) {
    void* ev = nullptr;
    VX(vx_event_create(&ev));
    VX(vx_event_record(ev, stream));
    marks.push_back({name, ev});
    acc.push_back({name, 0, 1, flops, bytes});
}

void timing_marks::finish(void* stream, std::vector<timing_entry>& out, bool append) {
    void* ev = nullptr;
    VX(vx_event_create(&ev));
    VX(vx_event_record(ev, stream));
    marks.push_back({"end", ev});
    std::map<std::string, timing_entry> by;
    std::vector<std::string> order;
    if (append)
        for (timing_entry const& t : out) { order.push_back(t.name); by[t.name] = t; }
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
    out.clear();
    for (auto& n : order) out.push_back(by[n]);
    for (auto& mk : marks) vx_event_destroy(mk.second);
    marks.clear();
    acc.clear();
}

namespace {

struct swin_exec {
    swin_model& m;
    void* stream;
    const uint8_t* wa;
    timing_marks tm;

    const float* fptr(packed_vec const& v) const { return reinterpret_cast<const float*>(wa + v.off); }
    void mark(const char* name, double flops, double bytes) { if (m.timing) tm.mark(name, flops, bytes, stream); }
    void finish_timing() { if (m.timing) tm.finish(stream, m.last_timing); }
    // win_ws > 0: rows are window tokens of a win_w x win_h map rolled by win_shift; out / res1 are addressed at their pixels
    void gemm(packed_gemm const& g, const void* A, long M, int lda, void* out, int epi, const void* res1, const char* group, int win_ws = 0,
              int win_w = 0, int win_h = 0, int win_shift = 0) {
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.lda = lda;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = (int)M; a.N = g.N; a.K = g.K; a.n_valid = g.n_real;
        a.epi = epi; a.out = out; a.ldo = g.n_real; a.res1 = res1;
        a.win_ws = win_ws; a.win_res = win_w; a.win_res_h = win_h; a.win_shift = win_shift;
        mark(group, 2.0 * M * g.n_real * g.k_real, (double)M * (g.k_real + g.n_real) * 2);
        VX(vx_gemm_f16(&a, stream));
    }
    void capture(const std::string& name, const void* src, int B, int h, int w, int C) {
        if (!m.captures) return;
        capture_entry& c = m.capture_bufs[name];
        const size_t bytes = (size_t)B * h * w * C * 2;
        vx_free(c.dev);
        c.dev = nullptr;
        VX(vx_malloc(&c.dev, bytes));
        c.shape[0] = B; c.shape[1] = h; c.shape[2] = w; c.shape[3] = C;
        c.f16 = true;
        VX(vx_memcpy_d2d(c.dev, src, bytes, stream));
    }
};

} // namespace

void swin_encode_batch_device(swin_model& m, void const* rgb_dev, int B, int w, int h, void* const outs[4], void* stream) {
    if (B < 1 || !rgb_dev || !outs) throw except("swin: empty batch or null pointer");
    swin_out so[4];
    for (int i = 0; i < 4; ++i) {
        if (!outs[i]) throw except("swin: output %d is null", i);
        so[i].ptr = outs[i];
    }
    swin_encode_pixels(m, nullptr, B, w, h, so, stream, rgb_dev);
}

// in8 == nullptr: pre-process rgb_dev (birefnet_process_input's normalisation, birefnet.cpp:259-270) into the workspace first
void swin_encode_pixels(swin_model& m, void const* in8_arg, int B, int w, int h, swin_out const outs[4], void* stream, void const* rgb_dev) {
    if (B < 1 || (!in8_arg && !rgb_dev)) throw except("swin: empty batch or null pointer");
    // patch 4, then three patch mergings that require even maps (swin.cpp:143)
    if (w < 32 || h < 32 || w % 32 || h % 32) throw except("swin: image extent %dx%d must be a positive multiple of 32", w, h);
    device_turn turn(*m.backend);
    void* s = stream ? stream : m.backend->stream;
    swin_params const& P = m.params;
    swin_weights const& Wt = m.weights;
    const int ws = P.window_size, N = ws * ws;
    int W0 = w / 4, H0 = h / 4;
    // scratch: 4 rotating activation buffers sized for stage 0 (every later stage halves tokens x 4C): padded window rows x 3C
    // (qkv) or tokens x 4C (mlp hidden) f16, plus the 8-channel input pixels
    auto padded_rows = [&](int cw, int ch) { return (size_t)B * ((cw + ws - 1) / ws) * ((ch + ws - 1) / ws) * N; };
    size_t big = 0;
    {
        int cw = W0, ch = H0;
        for (int l = 0; l < 4; ++l) {
            const size_t C = (size_t)P.layers[l].n_features;
            big = std::max(big, std::max(padded_rows(cw, ch) * 3 * C, (size_t)B * cw * ch * 4 * C) * 2);
            cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        }
        big = round_up<size_t>(big + 4096, 4096);
    }
    const size_t in_bytes = round_up<size_t>((size_t)B * w * h * 8 * 2 + 4096, 4096);
    const size_t need = 4 * big + in_bytes;
    if (m.ws.bytes < need) {
        VX(vx_stream_sync(m.backend->stream));
        VX(vx_stream_sync(s));
        vx_free(m.ws.ptr);
        m.ws = {};
        VX(vx_malloc(&m.ws.ptr, need));
        m.ws.bytes = need;
        VX(vx_memset(m.ws.ptr, 0, need, s)); // GEMM rows are read up to K padded to 64: stale bytes must be finite (pads of W are 0)
    }
    uint8_t* base = static_cast<uint8_t*>(m.ws.ptr);
    void *x = base, *t1 = base + big, *t2 = base + 2 * big, *t3 = base + 3 * big;
    const void* in8 = in8_arg ? in8_arg : base + 4 * big;

    swin_exec ex{m, s, static_cast<const uint8_t*>(m.weight_arena.ptr), {}};
    if (!in8_arg) {
        ex.mark("preprocess", 0, (double)B * w * h * 19);
        VX(vx_tv_preprocess(static_cast<const uint8_t*>(rgb_dev), base + 4 * big, (int64_t)B * w * h, s)); // same mean / std as sam (birefnet.cpp:259-270)
    }
    { // patch_embed (nn.cpp:166-180): 4x4 stride-4 conv as implicit GEMM on the 8-channel pixels, then LayerNorm
        packed_gemm const& g = Wt.patch_embed;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = in8;
        a.conv_kh = a.conv_kw = 4; a.conv_stride = 4; a.conv_pad = 0;
        a.conv_H = h; a.conv_W = w; a.conv_Cin = 8; a.conv_OH = H0; a.conv_OW = W0;
        a.W = ex.wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(ex.wa + g.b);
        a.M = B * H0 * W0; a.N = g.N; a.K = g.K; a.n_valid = g.n_real;
        a.epi = VX_EPI_F16; a.out = Wt.pe_norm ? t1 : x; a.ldo = g.n_real;
        ex.mark("patch_embed", 2.0 * a.M * g.n_real * 48, (double)B * w * h * 16 + (double)a.M * g.n_real * 2);
        VX(vx_gemm_f16(&a, s));
        if (Wt.pe_norm) {
            ex.mark("layernorm", 0, (double)a.M * g.n_real * 4);
            VX(vx_swin_layernorm_f16(t1, ex.fptr(Wt.pe_norm_w), ex.fptr(Wt.pe_norm_b), x, a.M, g.n_real, 1e-5f, 0, 0, 0, 0, 0, s));
        }
    }
    int cw = W0, ch = H0;
    ex.capture("patch_embed", x, B, ch, cw, P.embed_dim);
    for (int l = 0; l < 4; ++l) { // swin::layer (swin.cpp:215-235)
        swin_layer_t const& L = P.layers[l];
        const int C = L.n_features;
        const int nwx = (cw + ws - 1) / ws, nwy = (ch + ws - 1) / ws;
        const long T = (long)B * cw * ch, rows = (long)B * nwx * nwy * N;
        for (size_t bi = 0; bi < Wt.blocks[l].size(); ++bi) { // swin::block (swin.cpp:117-163)
            swin_block_weights const& b = Wt.blocks[l][bi];
            const int shift = bi % 2 == 0 ? 0 : ws / 2;
            const bool masked = shift > 0 || !m.mask_shifted_only; // swin.cpp:226-237: the reference masks every block
            ex.mark("layernorm", 0, (double)(T + rows) * C * 2);
            VX(vx_swin_layernorm_f16(x, ex.fptr(b.norm1_w), ex.fptr(b.norm1_b), t1, rows, C, 1e-5f, ch, cw, ws, shift, 0, s));
            ex.gemm(b.qkv, t1, rows, C, t2, VX_EPI_F16, nullptr, "gemm_qkv");
            ex.mark("window_attention", 4.0 * rows * N * C, (double)rows * C * 8);
            VX(vx_window_attention_masked_f16(t2, ex.wa + b.bias.off, t1, (int)(rows / N), N, L.n_heads, masked ? nwx : 0, masked ? nwy : 0, s));
            // proj + window_reverse + roll(+shift) + crop + shortcut: window rows are scattered to their pixels by the GEMM epilogue
            // (vx_swin_window_reverse_add_f16 is the unfused form, kept for the kernel-level parity test)
            ex.gemm(b.proj, t1, rows, C, t3, VX_EPI_F16_ADD, x, "gemm_proj", ws, cw, ch, shift);
            ex.mark("layernorm", 0, (double)T * C * 4);
            VX(vx_swin_layernorm_f16(t3, ex.fptr(b.norm2_w), ex.fptr(b.norm2_b), t1, T, C, 1e-5f, 0, 0, 0, 0, 0, s));
            ex.gemm(b.fc1, t1, T, C, t2, VX_EPI_F16_GELU, nullptr, "gemm_fc1");
            ex.gemm(b.fc2, t2, T, b.fc1.n_real, x, VX_EPI_F16_ADD, t3, "gemm_fc2");
            ex.capture("block_" + std::to_string(l) + "_" + std::to_string(bi), x, B, ch, cw, C);
        }
        ex.mark("layernorm", 0, (double)T * C * 6);
        VX(vx_swin_layernorm_strided_f16(x, ex.fptr(Wt.out_norm_w[l]), ex.fptr(Wt.out_norm_b[l]), outs[l].ptr, T, C, 1e-5f, outs[l].ld, outs[l].f32 ? 1 : 0, s)); // swin.cpp:255-258
        if (l < 3) { // patch_merging (swin.cpp:140-161)
            ex.mark("merge_layernorm", 0, (double)T * C * 4);
            VX(vx_swin_merge_layernorm_f16(x, ex.fptr(Wt.merge_norm_w[l]), ex.fptr(Wt.merge_norm_b[l]), t1, B, ch, cw, C, 1e-5f, s));
            ex.gemm(Wt.merge_reduction[l], t1, T / 4, 4 * C, t2, VX_EPI_F16, nullptr, "gemm_reduction");
            std::swap(x, t2);
            cw /= 2; ch /= 2;
        }
    }
    ex.finish_timing();
    if (!stream) VX(vx_stream_sync(s));
}

void swin_encode_batch_host(swin_model& m, uint8_t const* rgb, int B, int w, int h, float* const outs[4]) {
    if (B < 1 || !rgb || !outs) throw except("swin: empty batch or null pointer");
    device_turn turn(*m.backend);
    void* s = m.backend->stream;
    int dims[4][3];
    swin_output_dims(m, w, h, dims);
    void* in = nullptr;
    void* dev_out[4] = {nullptr, nullptr, nullptr, nullptr};
    auto release = [&]() { vx_free(in); for (void* p : dev_out) vx_free(p); };
    try {
        VX(vx_malloc(&in, (size_t)B * w * h * 3));
        VX(vx_memcpy_h2d(in, rgb, (size_t)B * w * h * 3, s));
        for (int i = 0; i < 4; ++i) VX(vx_malloc(&dev_out[i], (size_t)B * dims[i][0] * dims[i][1] * dims[i][2] * 4));
        swin_encode_batch_device(m, in, B, w, h, dev_out, s);
        for (int i = 0; i < 4; ++i) VX(vx_memcpy_d2h(outs[i], dev_out[i], (size_t)B * dims[i][0] * dims[i][1] * dims[i][2] * 4, s));
        VX(vx_stream_sync(s));
    } catch (...) {
        release();
        throw;
    }
    release();
}

} // namespace visp
