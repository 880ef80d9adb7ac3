#include "tinyvit.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

#include "../../include/visp_hip_kernels.h"
#include "packer.h"
#include "visp_util.h"

namespace visp {

sam_model* sam_load_model(char const* filepath, backend_device const& dev, int flags) {
    const bool with_data = !(flags & load_no_upload);
    model_file file = model_load(filepath, /*header_only=*/!with_data);
    if (file.arch() != "mobile-sam")
        throw except("Model %s has architecture '%.*s', expected 'mobile-sam'", filepath, (int)file.arch().size(), file.arch().data());
    auto model = std::make_unique<sam_model>();
    model->backend = &dev;
    tiny_vit_params const& P = model->params;
    arena_builder ab;
    packer pk{file, ab, with_data, file.tensor_layout() != layout_cwhn, file.conv2d_weights()};
    tinyvit_weights& Wt = model->weights;
    const std::string e = "enc.";
    int k, cin;
    Wt.pe0 = pk.conv(e + "patch_embed.seq.0.c", &k, &cin, /*dup_in=*/true);
    if (k != 3) throw except("mobile-sam: patch embedding kernel is %dx%d, expected 3x3", k, k);
    Wt.pe2 = pk.conv(e + "patch_embed.seq.2.c", &k, &cin);
    if (Wt.pe2.n_real != P.layers[0].embed_dim) throw except("mobile-sam: patch embedding width %d, expected %d", Wt.pe2.n_real, P.layers[0].embed_dim);
    auto merge = [&](std::string const& p) {
        tv_merge_weights m;
        m.conv1 = pk.conv(p + ".conv1.c");
        m.conv2 = pk.depthwise(p + ".conv2.c");
        m.conv3 = pk.conv(p + ".conv3.c");
        const int co = m.conv1.n_real;
        m.stride = (co == 320 || co == 448 || co == 576) ? 1 : 2; // mobile-sam.cpp:98-100
        return m;
    };
    for (int i = 0; i < P.layers[0].depth; ++i) {
        std::string p = e + "layers.0.blocks." + std::to_string(i);
        tv_mbconv_weights mb;
        mb.conv1 = pk.conv(p + ".conv1.c");
        mb.conv2 = pk.depthwise(p + ".conv2.c");
        mb.conv3 = pk.conv(p + ".conv3.c");
        if (vx_mbconv_dw_pw_supported(mb.conv2.C, mb.conv3.n_real, P.layers[0].resolution) && mb.conv3.N == mb.conv3.n_real && mb.conv3.K == mb.conv2.C) {
            mb.conv3_frag = ab.alloc((size_t)mb.conv3.N * mb.conv3.K * 2); // the fused kernel reads W3 in MFMA fragment order
            if (with_data) {
                std::vector<uint16_t> rows((size_t)mb.conv3.N * mb.conv3.K); // copy: alloc() may have moved the arena
                memcpy(rows.data(), ab.data.data() + mb.conv3.w, rows.size() * 2);
                VX(vx_mbconv_pack_w3(rows.data(), ab.data.data() + mb.conv3_frag));
            }
        }
        Wt.mbconv.push_back(mb);
    }
    Wt.merge[0] = merge(e + "layers.0.downsample");
    for (int l = 1; l < 4; ++l) {
        tiny_vit_layer const& L = P.layers[l];
        for (int i = 0; i < L.depth; ++i) {
            std::string p = e + "layers." + std::to_string(l) + ".blocks." + std::to_string(i);
            tv_block_weights b;
            b.attn_ln_w = pk.vec(p + ".attn.norm.weight");
            b.attn_ln_b = pk.vec(p + ".attn.norm.bias");
            b.qkv = pk.linear(p + ".attn.qkv");
            b.proj = pk.linear(p + ".attn.proj");
            b.bias = pk.attention_bias(p + ".attn.attention_biases_indexed", L.window_size * L.window_size, L.num_heads);
            if (b.qkv.n_real != 3 * L.embed_dim || b.qkv.k_real != L.embed_dim || L.embed_dim != 32 * L.num_heads)
                throw except("mobile-sam: %s.attn.qkv is %d x %d (this backend implements head_dim 32)", p.c_str(), b.qkv.n_real, b.qkv.k_real);
            b.local_conv = pk.depthwise(p + ".local_conv.c");
            b.mlp_ln_w = pk.vec(p + ".mlp.norm.weight");
            b.mlp_ln_b = pk.vec(p + ".mlp.norm.bias");
            b.fc1 = pk.linear(p + ".mlp.fc1");
            b.fc2 = pk.linear(p + ".mlp.fc2");
            Wt.blocks[l].push_back(b);
        }
        if (L.downsample) Wt.merge[l] = merge(e + "layers." + std::to_string(l) + ".downsample");
    }
    Wt.neck0 = pk.conv(e + "neck.0");
    Wt.neck1_w = pk.vec(e + "neck.1.weight");
    Wt.neck1_b = pk.vec(e + "neck.1.bias");
    Wt.neck2 = pk.conv(e + "neck.2");
    Wt.neck3_w = pk.vec(e + "neck.3.weight");
    Wt.neck3_b = pk.vec(e + "neck.3.bias");

    if (file.find("dec.iou_token.weight")) { // prompt encoder + mask decoder (mobile-sam.cpp:207-483)
        samdec_weights& D = model->dec;
        D.present = true;
        const int dim = D.dim;
        D.res = P.layers[3].resolution;
        D.gaussian = pk.host_vec("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", dim);
        for (int i = 0; i < 4; ++i) D.point_embed[i] = pk.host_vec("prompt_encoder.point_embeddings." + std::to_string(i) + ".weight", dim);
        D.not_a_point = pk.host_vec("prompt_encoder.not_a_point_embed.weight", dim);
        D.output_tokens = pk.host_vec("dec.iou_token.weight", dim);
        std::vector<float> mt = pk.host_vec("dec.mask_tokens.weight", 4 * dim);
        D.output_tokens.insert(D.output_tokens.end(), mt.begin(), mt.end());
        D.no_mask = pk.f16_vec("prompt_encoder.no_mask_embed.weight", dim);
        D.dense_pe = pk.f16_vec("dec.dense_positional_embedding", (int64_t)D.res * D.res * dim);
        auto attn = [&](std::string const& p) {
            sam_attn_weights a;
            a.q = pk.linear(p + ".q_proj");
            a.k = pk.linear(p + ".k_proj");
            a.v = pk.linear(p + ".v_proj");
            a.o = pk.linear(p + ".out_proj");
            if (a.q.k_real != dim || a.o.n_real != dim || a.q.n_real % D.heads || a.q.n_real / D.heads > 32)
                throw except("mobile-sam: %s has an unsupported shape (%d -> %d)", p.c_str(), a.q.k_real, a.q.n_real);
            return a;
        };
        for (int i = 0; i < 2; ++i) {
            std::string p = "dec.transformer.layers." + std::to_string(i);
            sam_twoway_weights L;
            L.self_attn = attn(p + ".self_attn");
            L.t2i = attn(p + ".cross_attn_t2i");
            L.i2t = attn(p + ".cross_attn_i2t");
            L.lin1 = pk.linear(p + ".mlp.lin1");
            L.lin2 = pk.linear(p + ".mlp.lin2");
            for (int n = 0; n < 4; ++n) {
                L.norm_w[n] = pk.vec(p + ".norm" + std::to_string(n + 1) + ".weight");
                L.norm_b[n] = pk.vec(p + ".norm" + std::to_string(n + 1) + ".bias");
            }
            D.layers.push_back(L);
        }
        D.final_attn = attn("dec.transformer.final_attn_t2i");
        D.final_norm_w = pk.vec("dec.transformer.norm_final_attn.weight");
        D.final_norm_b = pk.vec("dec.transformer.norm_final_attn.bias");
        D.up0 = pk.conv_transpose("dec.output_upscaling.0", 2, &D.up_c1);
        D.up_norm_w = pk.vec("dec.output_upscaling.1.weight");
        D.up_norm_b = pk.vec("dec.output_upscaling.1.bias");
        D.up3 = pk.conv_transpose("dec.output_upscaling.3", 2, &D.up_c2);
        if (D.up_c1 % 8 || D.up_c2 % 8 || D.up_c2 > 64) throw except("mobile-sam: upscaling widths %d / %d are not supported", D.up_c1, D.up_c2);
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 3; ++l) D.hyper[i][l] = pk.linear("dec.output_hypernetworks_mlps." + std::to_string(i) + ".layers." + std::to_string(l));
        for (int l = 0; l < 3; ++l) D.iou_head[l] = pk.linear("dec.iou_prediction_head.layers." + std::to_string(l));
        if (D.hyper[0][2].n_real != D.up_c2 || D.iou_head[2].n_real != 4) throw except("mobile-sam: hypernetwork / iou head output widths do not match");
        D.tables_off = ab.alloc((size_t)11 * dim * 4);
        if (with_data) {
            float* tb = reinterpret_cast<float*>(ab.data.data() + D.tables_off);
            memcpy(tb, D.gaussian.data(), (size_t)dim * 4);
            for (int i = 0; i < 4; ++i) memcpy(tb + (size_t)(1 + i) * dim, D.point_embed[i].data(), (size_t)dim * 4);
            memcpy(tb + (size_t)5 * dim, D.not_a_point.data(), (size_t)dim * 4);
            memcpy(tb + (size_t)6 * dim, D.output_tokens.data(), (size_t)5 * dim * 4);
        }
    }

    device_turn turn(dev);
    model->weight_arena.bytes = round_up<size_t>(ab.data.size(), 256) + 4096;
    VX(vx_malloc(&model->weight_arena.ptr, model->weight_arena.bytes));
    if (with_data) {
        VX(vx_memcpy_h2d(model->weight_arena.ptr, ab.data.data(), ab.data.size(), dev.stream));
        VX(vx_stream_sync(dev.stream));
        sam_weights_ready(*model);
    }
    return model.release();
}

// The arena is complete (uploaded here, or received from the rank that read the file): rebuild everything the host computes
// with from it -- the prompt encoder's tables and the f32 copy of the iou head (vision.cpp:80-82 picks a mask by its output).
void sam_weights_ready(sam_model& m) {
    samdec_weights& D = m.dec;
    if (D.present) {
        device_turn turn(*m.backend);
        void* s = m.backend->stream;
        const uint8_t* wa = static_cast<const uint8_t*>(m.weight_arena.ptr);
        const int dim = D.dim;
        std::vector<float> tb((size_t)11 * dim);
        VX(vx_memcpy_d2h(tb.data(), wa + D.tables_off, tb.size() * 4, s));
        D.gaussian.assign(tb.begin(), tb.begin() + dim);
        for (int i = 0; i < 4; ++i) D.point_embed[i].assign(tb.begin() + (size_t)(1 + i) * dim, tb.begin() + (size_t)(2 + i) * dim);
        D.not_a_point.assign(tb.begin() + (size_t)5 * dim, tb.begin() + (size_t)6 * dim);
        D.output_tokens.assign(tb.begin() + (size_t)6 * dim, tb.end());
        for (int l = 0; l < 3; ++l) {
            packed_gemm const& g = D.iou_head[l];
            std::vector<uint16_t> w16((size_t)g.N * g.K);
            VX(vx_memcpy_d2h(w16.data(), wa + g.w, w16.size() * 2, s));
            D.iou_w[l].resize((size_t)g.n_real * g.k_real);
            for (int n = 0; n < g.n_real; ++n)
                for (int k = 0; k < g.k_real; ++k) D.iou_w[l][(size_t)n * g.k_real + k] = f16_to_f32(w16[(size_t)n * g.K + k]);
            D.iou_b[l].assign((size_t)g.n_real, 0.0f);
            if (g.b != SIZE_MAX) VX(vx_memcpy_d2h(D.iou_b[l].data(), wa + g.b, (size_t)g.n_real * 4, s));
        }
    }
    m.weights_uploaded = true;
}

// e4m3 images of the stage MLPs' weights: read back from the packed f16 arena (what is on every rank), quantised on the host per output channel
void sam_set_fp8_mlp(sam_model& m, bool enable) {
    if (!enable || m.fp8_arena.ptr) { m.fp8_mlp = enable; return; }
    if (!m.weights_uploaded) throw except("sam: weights have not been uploaded");
    device_turn turn(*m.backend);
    std::vector<uint8_t> host;
    auto put = [&](size_t bytes) { const size_t off = (host.size() + 255) / 256 * 256; host.resize(off + bytes, 0); return off; };
    const uint8_t* wa = static_cast<const uint8_t*>(m.weight_arena.ptr);
    auto quantise = [&](packed_gemm const& g) {
        std::vector<uint16_t> h16((size_t)g.N * g.K);
        VX(vx_memcpy_d2h(h16.data(), wa + g.w, h16.size() * 2, m.backend->stream));
        std::vector<float> rows((size_t)g.n_real * g.k_real);
        for (int n = 0; n < g.n_real; ++n)
            for (int k = 0; k < g.k_real; ++k) rows[(size_t)n * g.k_real + k] = f16_to_f32(h16[(size_t)n * g.K + k]);
        fp8_linear f;
        f.N = (g.n_real + 127) / 128 * 128;
        f.Kp = (g.k_real + 127) / 128 * 128;
        f.w = put((size_t)f.N * f.Kp);
        f.s = put((size_t)f.N * 4);
        VX(vx_quantize_rows_e4m3_host(rows.data(), g.n_real, g.k_real, f.Kp, host.data() + f.w, reinterpret_cast<float*>(host.data() + f.s)));
        for (int n = g.n_real; n < f.N; ++n) reinterpret_cast<float*>(host.data() + f.s)[n] = 1.0f; // pad rows: zero bytes, any finite scale
        return f;
    };
    for (int l = 1; l < 4; ++l)
        for (tv_block_weights& b : m.weights.blocks[l]) {
            b.fc1_e4m3 = quantise(b.fc1);
            b.fc2_e4m3 = quantise(b.fc2);
        }
    VX(vx_malloc(&m.fp8_arena.ptr, host.size()));
    m.fp8_arena.bytes = host.size();
    VX(vx_memcpy_h2d(m.fp8_arena.ptr, host.data(), host.size(), m.backend->stream));
    VX(vx_stream_sync(m.backend->stream));
    m.fp8_mlp = true;
}

sam_model::~sam_model() {
    vx_free(fp8_arena.ptr);
    vx_free(fp8_ws.ptr);
    vx_free(ws.ptr);
    vx_free(embed.ptr);
    vx_free(dec_ws.ptr);
    for (auto& c : capture_bufs) vx_free(c.second.dev);
    vx_free(weight_arena.ptr);
}

namespace {

struct tv_exec {
    sam_model& m;
    void* stream;
    const uint8_t* wa;
    std::vector<std::pair<std::string, void*>> marks;
    std::vector<timing_entry> acc;
    int gemm_variant = 0; // vx_gemm_args.stages: tile-shape selector of the plain GEMM (0 = its default)

    const float* fptr(packed_vec const& v) const { return reinterpret_cast<const float*>(wa + v.off); }
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

    // C[M, N] = A[M, lda] * W^T (+bias) with the epilogues of the GEMM family
    void gemm(packed_gemm const& g, const void* A, long M, int lda, void* out, int epi, const void* res1, const char* group,
              bool post_gelu = false, int win_ws = 0, int win_res = 0) {
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.lda = lda;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = (int)M; a.N = g.N; a.K = g.K; a.n_valid = g.n_real;
        a.epi = epi; a.out = out; a.ldo = g.n_real; a.res1 = res1;
        a.stages = gemm_variant;
        a.post_gelu = post_gelu; a.win_ws = win_ws; a.win_res = win_res;
        mark(group, 2.0 * M * g.n_real * g.k_real, (double)M * (g.k_real + g.n_real) * 2);
        VX(vx_gemm_f16(&a, stream));
    }
    // k x k conv as implicit GEMM on NHWC f16
    void conv(packed_gemm const& g, const void* x, int B, int H, int W, int Cpix, int k, int stride, int pad, void* out, int epi, const char* group) {
        const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = x;
        a.conv_kh = a.conv_kw = k; a.conv_stride = stride; a.conv_pad = pad;
        a.conv_H = H; a.conv_W = W; a.conv_Cin = Cpix; a.conv_OH = OH; a.conv_OW = OW;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = B * OH * OW; a.N = g.N; a.K = g.K; a.n_valid = g.n_real;
        a.epi = epi; a.out = out; a.ldo = g.n_real;
        mark(group, 2.0 * a.M * g.n_real * g.k_real, (double)B * H * W * Cpix * 2 + (double)a.M * g.n_real * 2);
        VX(vx_gemm_f16(&a, stream));
    }
    // out f16 [M, n_out] = act(A W^T + b) [+ res] with A f16 [M, k_in] quantised to e4m3 rows first (scratch: m.fp8_ws)
    void fp8_linear_launch(fp8_linear const& f, packed_gemm const& g, const void* A, long M, int k_in, void* out, int n_out, int act, const void* res, const char* group) {
        const size_t need = (size_t)M * f.Kp + (size_t)M * 4 + 512;
        if (m.fp8_ws.bytes < need) {
            VX(vx_stream_sync(stream));
            vx_free(m.fp8_ws.ptr);
            m.fp8_ws = {};
            VX(vx_malloc(&m.fp8_ws.ptr, need));
            m.fp8_ws.bytes = need;
        }
        uint8_t* q = static_cast<uint8_t*>(m.fp8_ws.ptr);
        float* qs = reinterpret_cast<float*>(q + ((size_t)M * f.Kp + 255) / 256 * 256);
        const uint8_t* fa = static_cast<const uint8_t*>(m.fp8_arena.ptr);
        mark("quantise_e4m3", 0, (double)M * k_in * 3);
        VX(vx_quantize_rows_e4m3(A, k_in, q, qs, (int)M, k_in, f.Kp, stream));
        vx_gemm_fp8_args a;
        memset(&a, 0, sizeof a);
        a.A = q; a.a_scale = qs; a.W = fa + f.w; a.w_scale = reinterpret_cast<const float*>(fa + f.s);
        a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = (int)M; a.N = f.N; a.Kp = f.Kp; a.n_valid = n_out; a.out = out; a.ldo = n_out; a.act = act; a.res = res;
        mark(group, 2.0 * M * g.n_real * g.k_real, (double)M * (g.k_real + 2.0 * g.n_real));
        VX(vx_gemm_fp8(&a, stream));
    }
    void capture(const char* name, const void* src, int B, int res, int C) {
        if (!m.captures) return;
        capture_entry& c = m.capture_bufs[name];
        const size_t bytes = (size_t)B * res * res * C * 2;
        vx_free(c.dev);
        c.dev = nullptr;
        VX(vx_malloc(&c.dev, bytes));
        c.shape[0] = B; c.shape[1] = res; c.shape[2] = res; c.shape[3] = C;
        c.f16 = true;
        VX(vx_memcpy_d2d(c.dev, src, bytes, stream));
    }
    void dw(packed_dw const& d, const void* x, void* y, int B, int H, int W, int stride, bool gelu, const char* group) {
        mark(group, 2.0 * B * (H / stride) * (W / stride) * 9 * d.C, (double)B * H * W * d.C * 2 * (1.0 + 1.0 / (stride * stride)));
        VX(vx_dwconv3x3_f16(x, wa + d.w, reinterpret_cast<const float*>(wa + d.b), y, B, H, W, d.C, stride, gelu, stream));
    }
};

} // namespace

void sam_encode_batch_device(sam_model& m, void const* rgb_dev, int B, void* out_dev, void* stream) {
    if (!m.weights_uploaded) throw except("sam: weights have not been uploaded (load_no_upload without weights_ready)");
    if (B < 1 || !rgb_dev || !out_dev) throw except("sam: empty batch or null pointer");
    device_turn turn(*m.backend);
    void* s = stream ? stream : m.backend->stream;
    tiny_vit_params const& P = m.params;
    tinyvit_weights const& Wt = m.weights;
    const int S = P.img_size;
    // scratch: three rotating activation buffers (largest map: 256 x 256 x 256 f16 of an MBConv) + input / patch-embed buffers
    const size_t big = (size_t)B * 256 * 256 * 256 * 2 + 4096;
    const size_t in_bytes = (size_t)B * S * S * 8 * 2 + 4096, pe_bytes = (size_t)B * (S / 2) * (S / 2) * 32 * 2 + 4096;
    const size_t need = 4 * big + in_bytes + pe_bytes;
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
    void* buf[4] = {base, base + big, base + 2 * big, base + 3 * big};
    void* in8 = base + 4 * big;
    void* pe = base + 4 * big + in_bytes;

    tv_exec ex{m, s, static_cast<const uint8_t*>(m.weight_arena.ptr), {}, {}};
    if (const char* e = getenv("VISP_TV_GEMM_VARIANT")) ex.gemm_variant = atoi(e);
    const bool fuse_mbconv = !getenv("VISP_TV_NO_FUSED_MBCONV");
    ex.mark("preprocess", 0, (double)B * S * S * 19);
    VX(vx_tv_preprocess(static_cast<const uint8_t*>(rgb_dev), in8, (int64_t)B * S * S, s));
    // patch_embed (mobile-sam.cpp:70-75)
    ex.conv(Wt.pe0, in8, B, S, S, 8, 3, 2, 1, pe, VX_EPI_F16_GELU, "patch_embed");
    void *x = buf[0], *t1 = buf[1], *t2 = buf[2], *t3 = buf[3];
    ex.conv(Wt.pe2, pe, B, S / 2, S / 2, Wt.pe0.n_real, 3, 2, 1, x, VX_EPI_F16, "patch_embed");
    int res = P.layers[0].resolution, C = P.layers[0].embed_dim;
    ex.capture("patch_embed", x, B, res, C);
    // conv_layer (mobile-sam.cpp:162-170): mb_conv x depth
    for (tv_mbconv_weights const& mb : Wt.mbconv) {
        const long M = (long)B * res * res;
        ex.gemm(mb.conv1, x, M, C, t1, VX_EPI_F16_GELU, nullptr, "mbconv_1x1");
        if (fuse_mbconv && mb.conv3_frag != SIZE_MAX && mb.conv3.b != SIZE_MAX && vx_mbconv_dw_pw_supported(mb.conv2.C, mb.conv3.n_real, res)) {
            // depthwise + GELU + conv3 + residual + GELU in one launch: the depthwise output stays in LDS (kernels_mbconv.hip)
            ex.mark("mbconv_dw_pw", 2.0 * M * (9.0 * mb.conv2.C + (double)mb.conv2.C * C), (double)M * (mb.conv2.C + 2.0 * C) * 2);
            VX(vx_mbconv_dw_pw_f16(t1, ex.wa + mb.conv2.w, reinterpret_cast<const float*>(ex.wa + mb.conv2.b), ex.wa + mb.conv3_frag,
                                   reinterpret_cast<const float*>(ex.wa + mb.conv3.b), x, t3, B, res, res, mb.conv2.C, C, s));
        } else {
            ex.dw(mb.conv2, t1, t2, B, res, res, 1, true, "depthwise_mbconv"); // the step's largest kernel: [B, 256, 256, 256] in and out
            ex.gemm(mb.conv3, t2, M, mb.conv1.n_real, t3, VX_EPI_F16_ADD, x, "mbconv_1x1", /*post_gelu=*/true); // gelu(x + conv3) in the epilogue
        }
        std::swap(x, t3);
    }
    auto patch_merging = [&](tv_merge_weights const& mg) { // mobile-sam.cpp:94-110
        const long M = (long)B * res * res;
        ex.gemm(mg.conv1, x, M, C, t1, VX_EPI_F16_GELU, nullptr, "merge_1x1");
        const int co = mg.conv1.n_real;
        ex.dw(mg.conv2, t1, t2, B, res, res, mg.stride, true, "depthwise");
        res = (res + 2 - 3) / mg.stride + 1;
        ex.gemm(mg.conv3, t2, (long)B * res * res, co, t1, VX_EPI_F16, nullptr, "merge_1x1");
        std::swap(x, t1);
        C = mg.conv3.n_real;
    };
    patch_merging(Wt.merge[0]);
    ex.capture("layer_0", x, B, res, C);
    for (int l = 1; l < 4; ++l) { // basic_layer (mobile-sam.cpp:172-186)
        tiny_vit_layer const& L = P.layers[l];
        if (res != L.resolution || C != L.embed_dim) throw except("sam: layer %d gets %dx%dx%d, expects %dx%d", l, res, res, C, L.resolution, L.embed_dim);
        const int ws = L.window_size, nw = (res + ws - 1) / ws, N = ws * ws;
        const long T = (long)B * res * res, rows = (long)B * nw * nw * N;
        for (tv_block_weights const& b : Wt.blocks[l]) { // tiny_vit_block (mobile-sam.cpp:133-160)
            ex.mark("layernorm", 0, (double)(T + rows) * C * 2);
            VX(vx_layernorm_f16(x, ex.fptr(b.attn_ln_w), ex.fptr(b.attn_ln_b), t1, rows, C, 1e-5f, res, ws, 0, s));
            ex.gemm(b.qkv, t1, rows, C, t2, VX_EPI_F16, nullptr, "gemm_qkv");
            ex.mark("window_attention", 4.0 * rows * N * C, (double)rows * C * 8);
            VX(vx_window_attention_f16(t2, ex.wa + b.bias.off, t1, (int)(rows / N), N, L.num_heads, s));
            // proj + window_reverse + residual: window rows are scattered to their pixels by the epilogue
            ex.gemm(b.proj, t1, rows, C, t2, VX_EPI_F16_ADD, x, "gemm_proj", false, ws, res);
            ex.dw(b.local_conv, t2, t3, B, res, res, 1, false, "depthwise");
            ex.mark("layernorm", 0, (double)T * C * 4);
            VX(vx_layernorm_f16(t3, ex.fptr(b.mlp_ln_w), ex.fptr(b.mlp_ln_b), t1, T, C, 1e-5f, 0, 0, 0, s));
            if (m.fp8_mlp) { // opt-in: both products on the e4m3 matrix instruction, activations quantised per token in between
                const int hid = b.fc1.n_real;
                ex.fp8_linear_launch(b.fc1_e4m3, b.fc1, t1, T, C, t2, hid, 1, nullptr, "gemm_fc1_e4m3");
                ex.fp8_linear_launch(b.fc2_e4m3, b.fc2, t2, T, hid, x, C, 0, t3, "gemm_fc2_e4m3");
            } else {
            ex.gemm(b.fc1, t1, T, C, t2, VX_EPI_F16_GELU, nullptr, "gemm_fc1");
            ex.gemm(b.fc2, t2, T, b.fc1.n_real, x, VX_EPI_F16_ADD, t3, "gemm_fc2");
            }
        }
        if (L.downsample) patch_merging(Wt.merge[l]);
        ex.capture(("layer_" + std::to_string(l)).c_str(), x, B, res, C);
    }
    // neck (mobile-sam.cpp:197-205): conv 1x1, LayerNorm over channels, conv 3x3, LayerNorm -> f32 [B, res, res, 256]
    const long T = (long)B * res * res;
    ex.gemm(Wt.neck0, x, T, C, t1, VX_EPI_F16, nullptr, "neck");
    const int NC = Wt.neck0.n_real;
    ex.mark("layernorm", 0, (double)T * NC * 4);
    VX(vx_layernorm_f16(t1, ex.fptr(Wt.neck1_w), ex.fptr(Wt.neck1_b), t2, T, NC, 1e-5f, 0, 0, 0, s));
    ex.conv(Wt.neck2, t2, B, res, res, NC, 3, 1, 1, t1, VX_EPI_F16, "neck");
    ex.mark("layernorm", 0, (double)T * NC * 6);
    VX(vx_layernorm_f16(t1, ex.fptr(Wt.neck3_w), ex.fptr(Wt.neck3_b), out_dev, T, Wt.neck2.n_real, 1e-5f, 0, 0, 1, s));
    ex.finish_timing();
    if (!stream) VX(vx_stream_sync(s));
}

void sam_encode_batch_host(sam_model& m, uint8_t const* rgb, int B, float* out) {
    if (B < 1 || !rgb || !out) throw except("sam: empty batch or null pointer");
    device_turn turn(*m.backend);
    const int S = m.params.img_size, R = m.params.layers[3].resolution;
    const size_t in_bytes = (size_t)B * S * S * 3, out_bytes = (size_t)B * R * R * 256 * 4;
    void *din = nullptr, *dout = nullptr;
    VX(vx_malloc(&din, in_bytes));
    VX(vx_malloc(&dout, out_bytes));
    void* s = m.backend->stream;
    try {
        VX(vx_memcpy_h2d(din, rgb, in_bytes, s));
        sam_encode_batch_device(m, din, B, dout, s);
        VX(vx_memcpy_d2h(out, dout, out_bytes, s));
        VX(vx_stream_sync(s));
    } catch (...) {
        vx_free(din);
        vx_free(dout);
        throw;
    }
    vx_free(din);
    vx_free(dout);
}

void sam_encode(sam_model& m, image_view image) {
    if (is_float(image.format) || n_channels(image.format) < 3)
        throw except("sam: unsupported input image format [%d], expected an 8-bit colour image", int(image.format));
    if (image.extent[0] < 1 || image.extent[1] < 1) throw except("sam: empty image");
    const int S = m.params.img_size;
    image_data rgb = image_to_rgb_u8(image);
    image_view v = view_of(rgb);
    image_data resized;
    // sam_process_input (mobile-sam.cpp:533-547): resize_longest_side, then u8 -> f32 with source coordinates clamped
    const float scale = float(S) / float(std::max(image.extent[0], image.extent[1]));
    if (scale != 1) {
        resized = image_scale(v, i32x2{{int(image.extent[0] * scale + 0.5f), int(image.extent[1] * scale + 0.5f)}});
        v = view_of(resized);
    }
    std::vector<uint8_t> square((size_t)S * S * 3);
    const uint8_t* src = static_cast<const uint8_t*>(v.data);
    const int vw = std::min(v.extent[0], S), vh = std::min(v.extent[1], S);
    for (int y = 0; y < S; ++y) { // clamped source coordinates = edge replication (image.cpp convert<>, :216-226)
        uint8_t* dst = &square[(size_t)y * S * 3];
        if (y < vh) {
            memcpy(dst, src + (size_t)y * v.stride, (size_t)vw * 3);
            for (int x = vw; x < S; ++x) memcpy(dst + (size_t)x * 3, dst + (size_t)(vw - 1) * 3, 3);
        } else {
            memcpy(dst, dst - (size_t)S * 3, (size_t)S * 3);
        }
    }
    device_turn turn(*m.backend);
    const int R = m.params.layers[3].resolution;
    const size_t out_bytes = (size_t)R * R * 256 * 4;
    if (!m.embed.ptr) { // embedding + the staging copy of the input stay with the model
        VX(vx_malloc(&m.embed.ptr, out_bytes + square.size()));
        m.embed.bytes = out_bytes + square.size();
    }
    void* din = static_cast<uint8_t*>(m.embed.ptr) + out_bytes;
    void* s = m.backend->stream;
    VX(vx_memcpy_h2d(din, square.data(), square.size(), s));
    sam_encode_batch_device(m, din, 1, m.embed.ptr, s);
    VX(vx_stream_sync(s));
    m.image_extent = image.extent;
}

// ---- sam_compute: prompt encoder + mask decoder + mask post-processing ------------------------------------------------

namespace {

float sam_transform_coord(int p, float scale, int image_size) { // mobile-sam.cpp:213-217
    const float center_normalized = (float(p) * scale + 0.5f) / float(image_size);
    return 2.f * center_normalized - 1.f;
}

} // namespace

image_data sam_compute(sam_model& m, int const* prompt, int n_prompt) {
    samdec_weights const& D = m.dec;
    if (n_prompt != 2 && n_prompt != 4) throw except("sam: bad number of arguments (%d), must be 2 or 4", n_prompt);
    if (!D.present) throw except("sam: this model file holds no prompt encoder / mask decoder (dec.* tensors)");
    if (D.gaussian.empty()) throw except("sam: the decoder's host tables were not read (model loaded without data)");
    if (!m.embed.ptr || m.image_extent[0] <= 0) throw except("Missing image embeds, call sam_encode() first");
    device_turn turn(*m.backend);
    void* s = m.backend->stream;
    const int dim = D.dim, H = D.heads, res = D.res, Nk = res * res, Nt = 7, F = dim / 2;
    const int image_size = m.params.img_size, mask_size = 4 * res;
    const bool is_box = n_prompt == 4;

    // ---- prompt encoder on the host (two points of 256 sin/cos values; mobile-sam.cpp:219-286)
    const float scale = float(image_size) / float(std::max(m.image_extent[0], m.image_extent[1]));
    float coords[4] = {sam_transform_coord(prompt[0], scale, image_size), sam_transform_coord(prompt[1], scale, image_size), 0.f, 0.f};
    if (is_box) { coords[2] = sam_transform_coord(prompt[2], scale, image_size); coords[3] = sam_transform_coord(prompt[3], scale, image_size); }
    std::vector<float> tokens(D.output_tokens); // [iou | 4 mask tokens]
    tokens.resize((size_t)Nt * dim);
    for (int i = 0; i < 2; ++i) {
        float* t = tokens.data() + (size_t)(5 + i) * dim;
        for (int f = 0; f < F; ++f) {
            float v = coords[2 * i] * D.gaussian[f] + coords[2 * i + 1] * D.gaussian[F + f];
            v *= 2.f * 3.14159265358979323846f;
            t[f] = std::sin(v);
            t[F + f] = std::cos(v);
        }
        if (is_box) for (int c = 0; c < dim; ++c) t[c] += D.point_embed[2 + i][c];
        else if (i == 0) for (int c = 0; c < dim; ++c) t[c] += D.point_embed[1][c];
        else for (int c = 0; c < dim; ++c) t[c] = D.not_a_point[c];
    }
    std::vector<uint16_t> tokens_h((size_t)8 * dim, 0);
    for (size_t i = 0; i < tokens.size(); ++i) tokens_h[i] = f32_to_f16(tokens[i]);

    // ---- scratch (f16 elements)
    const size_t big = (size_t)Nk * dim, tok = (size_t)8 * dim;
    size_t cursor = 0;
    auto take = [&](size_t elems) { size_t o = cursor; cursor += round_up<size_t>(elems * 2 + 256, 256); return o; };
    const size_t o_T = take(tok), o_Q = take(tok), o_t1 = take(tok), o_t2 = take(tok), o_h = take((size_t)8 * 2048 + 64);
    const size_t o_K = take(big), o_k1 = take(big), o_k2 = take(big), o_pq = take(big), o_pk = take(big), o_pv = take(big), o_ao = take(big);
    const size_t o_u1 = take((size_t)4 * Nk * D.up_c1), o_u1n = take((size_t)4 * Nk * D.up_c1), o_u2p = take((size_t)16 * Nk * D.up_c2), o_u2 = take((size_t)16 * Nk * D.up_c2 + 64);
    const size_t o_hy = take((size_t)32 * 64), o_ha = take(tok), o_hb = take(tok), o_mask = take((size_t)16 * Nk * 8), o_iou = take(64);
    const size_t o_scaled = take((size_t)image_size * image_size * 2 /* f32 */), o_out = take(((size_t)m.image_extent[0] * m.image_extent[1] + 1) / 2);
    if (m.dec_ws.bytes < cursor) {
        VX(vx_stream_sync(s));
        vx_free(m.dec_ws.ptr);
        m.dec_ws = {};
        VX(vx_malloc(&m.dec_ws.ptr, cursor));
        m.dec_ws.bytes = cursor;
        VX(vx_memset(m.dec_ws.ptr, 0, cursor, s));
    }
    uint8_t* base = static_cast<uint8_t*>(m.dec_ws.ptr);
    auto P = [&](size_t off) { return static_cast<void*>(base + off); };
    void *T = P(o_T), *Q = P(o_Q), *t1 = P(o_t1), *t2 = P(o_t2), *hbuf = P(o_h), *K = P(o_K), *k1 = P(o_k1), *k2 = P(o_k2);
    void *pq = P(o_pq), *pk = P(o_pk), *pv = P(o_pv), *ao = P(o_ao), *u1 = P(o_u1), *u1n = P(o_u1n), *u2p = P(o_u2p), *u2 = P(o_u2);
    void *hy = P(o_hy), *ha = P(o_ha), *hb = P(o_hb), *masks_d = P(o_mask), *iou_d = P(o_iou), *scaled = P(o_scaled), *mask_u8 = P(o_out);

    tv_exec ex{m, s, static_cast<const uint8_t*>(m.weight_arena.ptr), {}, {}};
    const bool timing = m.timing;
    m.timing = false; // the per-group table belongs to the encoder
    const uint8_t* wa = ex.wa;
    VX(vx_memcpy_h2d(T, tokens_h.data(), tokens_h.size() * 2, s));
    VX(vx_memcpy_d2d(Q, T, tok * 2, s));
    // src = image_embeddings + dense prompt (no_mask_embed broadcast over the pixels), mobile-sam.cpp:437-440
    VX(vx_add_rows_f16(m.embed.ptr, 1, wa + D.no_mask.off, dim, K, (int64_t)Nk * dim, s));
    const void* KPE = wa + D.dense_pe.off;

    auto linear = [&](packed_gemm const& g, const void* A, int M, void* out, int epi = VX_EPI_F16, const void* res1 = nullptr) {
        ex.gemm(g, A, M, g.k_real, out, epi, res1, "decoder");
    };
    auto norm = [&](packed_vec const& w, packed_vec const& b, const void* x, void* y, long rows, int C) {
        VX(vx_layernorm_f16(x, ex.fptr(w), ex.fptr(b), y, rows, C, 1e-5f, 0, 0, 0, s));
    };
    // decoder_attention (mobile-sam.cpp:306-320); out = [res1 +] out_proj(attention(q_proj(q), k_proj(k), v_proj(v)))
    auto attention = [&](sam_attn_weights const& w, const void* q, int Nq, const void* k, const void* v, int Nkv, void* out, const void* res1) {
        linear(w.q, q, Nq, pq);
        linear(w.k, k, Nkv, pk);
        linear(w.v, v, Nkv, pv);
        VX(vx_small_attention_f16(pq, pk, pv, ao, Nq, Nkv, H, w.q.n_real / H, s));
        linear(w.o, ao, Nq, out, res1 ? VX_EPI_F16_ADD : VX_EPI_F16, res1);
    };
    auto add = [&](const void* a, const void* b, void* y, long n) { VX(vx_add_rows_f16(a, 0, b, n, y, n, s)); };

    for (size_t li = 0; li < D.layers.size(); ++li) { // two_way_attention_block (mobile-sam.cpp:322-364)
        sam_twoway_weights const& L = D.layers[li];
        if (li == 0) { // skip_first_layer_pe: queries = self_attn(queries)
            attention(L.self_attn, Q, Nt, Q, Q, Nt, t2, nullptr);
        } else {
            add(Q, T, t1, (long)Nt * dim);
            attention(L.self_attn, t1, Nt, t1, Q, Nt, t2, Q);
        }
        norm(L.norm_w[0], L.norm_b[0], t2, Q, Nt, dim);
        add(Q, T, t1, (long)Nt * dim); // tokens attending to the image embedding
        add(K, KPE, k1, (long)Nk * dim);
        attention(L.t2i, t1, Nt, k1, K, Nk, t2, Q);
        norm(L.norm_w[1], L.norm_b[1], t2, Q, Nt, dim);
        linear(L.lin1, Q, Nt, hbuf, VX_EPI_F16_RELU); // mlp_block
        linear(L.lin2, hbuf, Nt, t2, VX_EPI_F16_ADD, Q);
        norm(L.norm_w[2], L.norm_b[2], t2, Q, Nt, dim);
        add(Q, T, t1, (long)Nt * dim); // image embedding attending to the tokens (k1 = keys + key_pe from above)
        attention(L.i2t, k1, Nk, t1, Q, Nt, k2, K);
        norm(L.norm_w[3], L.norm_b[3], k2, K, Nk, dim);
    }
    add(Q, T, t1, (long)Nt * dim); // final attention from the points to the image (mobile-sam.cpp:386-392)
    add(K, KPE, k1, (long)Nk * dim);
    attention(D.final_attn, t1, Nt, k1, K, Nk, t2, Q);
    norm(D.final_norm_w, D.final_norm_b, t2, Q, Nt, dim);

    // upscale_outputs (mobile-sam.cpp:396-404): convT k2 s2 -> LayerNorm -> GELU -> convT k2 s2 -> GELU
    auto conv_transpose = [&](packed_gemm const& g, const void* x, int hw, int cin, int cout, void* out) {
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = x; a.lda = cin;
        a.W = wa + g.w; a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = hw * hw; a.N = g.N; a.K = g.K; a.n_valid = g.n_real;
        a.epi = VX_EPI_PIXSHUF; a.out = out; a.ldo = cout;
        a.ps_s = 2; a.ps_Cout = cout; a.ps_H = hw; a.ps_W = hw;
        VX(vx_gemm_f16(&a, s));
    };
    conv_transpose(D.up0, K, res, dim, D.up_c1, u1);
    norm(D.up_norm_w, D.up_norm_b, u1, u1n, (long)4 * Nk, D.up_c1);
    VX(vx_add_gelu_f16(u1n, nullptr, u1, (int64_t)4 * Nk * D.up_c1, s));
    conv_transpose(D.up3, u1, 2 * res, D.up_c1, D.up_c2, u2p);
    VX(vx_add_gelu_f16(u2p, nullptr, u2, (int64_t)16 * Nk * D.up_c2, s));

    // hypernetwork MLPs on the four mask tokens -> hyper_in rows (a [32][64] f16 operand, zero padded)
    auto raw_gemm = [&](const void* A, int M, int lda, const void* W, const float* bias, int N, int Kp, int n_valid, void* out, int ldo, int epi) {
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.M = M; a.N = N; a.K = Kp; a.n_valid = n_valid; a.epi = epi; a.out = out; a.ldo = ldo;
        VX(vx_gemm_f16(&a, s));
    };
    auto bias_of = [&](packed_gemm const& g) { return g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b); };
    for (int i = 0; i < 4; ++i) {
        const uint8_t* tok_i = static_cast<const uint8_t*>(Q) + (size_t)(1 + i) * dim * 2;
        linear(D.hyper[i][0], tok_i, 1, ha, VX_EPI_F16_RELU);
        linear(D.hyper[i][1], ha, 1, hb, VX_EPI_F16_RELU);
        packed_gemm const& g = D.hyper[i][2];
        raw_gemm(hb, 1, g.k_real, wa + g.w, bias_of(g), g.N, g.K, g.n_real, static_cast<uint8_t*>(hy) + (size_t)i * 64 * 2, 64, VX_EPI_F16);
    }
    // masks[i][pixel] = <upscaled[pixel], hyper_in[i]> (mobile-sam.cpp:472-473): M = 16 Nk pixels, K = up_c2, 4 of 8 stored columns real
    raw_gemm(u2, 16 * Nk, D.up_c2, hy, nullptr, 32, 64, 8, masks_d, 8, VX_EPI_F16);
    // iou prediction head on the iou token (mobile-sam.cpp:476-481): three small linears on ONE token whose output decides
    // which mask is returned (vision.cpp:80-82), so it runs in f32 on the host from the token's values: an f16 pipeline
    // (ulp 5e-4 near 1) can flip the choice between near-tied predictions and return an entirely different mask
    const size_t n_px = (size_t)mask_size * mask_size;
    (void)iou_d;
    {
        std::vector<uint16_t> t16((size_t)dim);
        VX(vx_memcpy_d2h(t16.data(), Q, t16.size() * 2, s));
        VX(vx_stream_sync(s));
        std::vector<float> cur((size_t)dim), nxt;
        for (int i = 0; i < dim; ++i) cur[(size_t)i] = f16_to_f32(t16[(size_t)i]);
        for (int l = 0; l < 3; ++l) {
            packed_gemm const& g = D.iou_head[l];
            nxt.assign((size_t)g.n_real, 0.0f);
            for (int n = 0; n < g.n_real; ++n) {
                float acc = 0.0f;
                const float* wr = D.iou_w[l].data() + (size_t)n * g.k_real;
                for (int k = 0; k < g.k_real; ++k) acc += wr[k] * cur[(size_t)k];
                acc += D.iou_b[l][(size_t)n];
                nxt[(size_t)n] = l < 2 ? std::max(acc, 0.0f) : acc;
            }
            cur.swap(nxt);
        }
        for (int i = 0; i < 4; ++i) m.last_iou[i] = cur[(size_t)i];
    }
    // the decoder's only host decision: best of the FIRST THREE masks by predicted iou (vision.cpp:80-82)
    const int idx = int(std::max_element(m.last_iou, m.last_iou + 3) - m.last_iou);
    // sam_process_mask (mobile-sam.cpp:556-583): mask -> image_size^2, crop to the scaled extent, resize to the image, threshold
    const i32x2 target = m.image_extent;
    const float up = float(image_size) / float(std::max(target[0], target[1]));
    const int sw = int(target[0] * up + 0.5f), sh = int(target[1] * up + 0.5f);
    VX(vx_sam_interpolate(static_cast<const uint16_t*>(masks_d) + idx, 1, mask_size, mask_size, mask_size, 8, scaled, image_size, image_size, 0, s));
    VX(vx_sam_interpolate(scaled, 0, sw, sh, image_size, 1, mask_u8, target[0], target[1], 1, s));
    image_data out = image_alloc(target, image_format::alpha_u8);
    VX(vx_memcpy_d2h(out.data.get(), mask_u8, (size_t)target[0] * target[1], s));
    if (m.captures) { // test hook: the four logit planes (f16-rounded)
        std::vector<uint16_t> mh(n_px * 8);
        VX(vx_memcpy_d2h(mh.data(), masks_d, mh.size() * 2, s));
        VX(vx_stream_sync(s));
        m.last_masks.resize(4 * n_px);
        for (size_t px = 0; px < n_px; ++px)
            for (int i = 0; i < 4; ++i) m.last_masks[(size_t)i * n_px + px] = f16_to_f32(mh[px * 8 + i]);
    } else {
        m.last_masks.clear();
    }
    VX(vx_stream_sync(s));
    m.timing = timing;
    return out;
}

} // namespace visp
