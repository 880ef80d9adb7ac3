#include "depthany.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

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
} // namespace

//
// backend (reference src/visp/ml.cpp:59-95)

device_turn::device_turn(backend_device const& dev) : lock(dev.turn) { VX(vx_set_device(dev.index)); }

backend_device* backend_init(int device_index) {
    int n = vx_device_count();
    if (n <= 0) throw except("Failed to initialize backend, no suitable device available");
    if (device_index < 0 || device_index >= n) throw except("Failed to initialize backend, device index %d out of range (%d devices)", device_index, n);
    VX(vx_set_device(device_index));
    char name[256] = {0}, arch[128] = {0};
    size_t total = 0;
    int n_cu = 0;
    VX(vx_device_info(device_index, name, sizeof name, arch, sizeof arch, &total, nullptr, &n_cu));
    if (strncmp(arch, "gfx950", 6) != 0)
        throw except("Failed to initialize backend: device %d is %s, this backend is built for gfx950 (MI355X) only", device_index, arch);
    auto* d = new backend_device;
    d->index = device_index;
    d->name = std::string("HIP") + std::to_string(device_index);
    d->description = std::string(name) + " (" + arch + ")";
    d->total_mem = total;
    d->n_cu = n_cu;
    VX(vx_stream_create(&d->stream));
    return d;
}
backend_device::~backend_device() {
    if (stream) vx_stream_destroy(stream);
}

//
// params (reference dino.cpp:119-126, depth-anything.cpp:112-128)

dino_params dino_detect_params(model_file const& file) {
    dino_params p{};
    p.patch_size = file.get_int("dino.patch_size");
    p.embed_dim = file.get_int("dino.embed_dim");
    p.n_heads = file.get_int("dino.n_heads");
    p.n_layers = file.get_int("dino.n_layers");
    return p;
}
depthany_params depthany_detect_params(model_file const& file) {
    depthany_params p;
    p.dino = dino_detect_params(file);
    p.image_size = file.get_int("depthanything.image_size");
    file.get_array("depthanything.feature_layers", p.feature_layers.data(), 4);
    return p;
}
i32x2 depthany_image_extent(i32x2 extent, depthany_params const& p) {
    int min_side = std::min(extent[0], extent[1]);
    int tgt_side = std::max(p.image_size, next_multiple(min_side, p.image_multiple));
    i32x2 target = {{extent[0] * tgt_side / min_side, extent[1] * tgt_side / min_side}};
    return i32x2{{next_multiple(target[0], p.image_multiple), next_multiple(target[1], p.image_multiple)}};
}

//
// weight packing: the counterpart of model_transfer (reference src/visp/ml.cpp:449-516).
// The reference converts f16->f32 and WHCN->CWHN for its CPU backend; this backend keeps f16
// (the file's type) for matrix operands, converts small vectors to f32, and lays every matrix
// out as [N][K] row-major with k = (ky, kx, cin) -- i.e. CWHN kernels flattened.

namespace {

struct arena_builder {
    std::vector<uint8_t> data;
    size_t alloc(size_t bytes) {
        size_t off = round_up<size_t>(data.size(), 256);
        data.resize(off + bytes, 0);
        return off;
    }
};

std::vector<float> to_f32(gguf_tensor const& t) {
    std::vector<float> out((size_t)t.n_elements());
    if (!t.data) throw except("tensor %s has no data (header-only load)", t.name.c_str());
    if (t.type == GGML_F32) memcpy(out.data(), t.data, out.size() * 4);
    else if (t.type == GGML_F16) {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(t.data);
        for (size_t i = 0; i < out.size(); ++i) out[i] = f16_to_f32(s[i]);
    } else throw except("tensor %s: unsupported type %d", t.name.c_str(), t.type);
    return out;
}

struct packer {
    model_file const& file;
    arena_builder& ab;
    bool with_data;
    bool file_whcn;
    std::vector<int32_t> conv2d;

    packer(model_file const& f, arena_builder& a, bool data)
        : file(f), ab(a), with_data(data), file_whcn(f.tensor_layout() == layout_whcn), conv2d(f.conv2d_weights()) {}

    bool is_listed_conv2d(std::string_view name) const {
        auto it = file.index.find(name);
        return it != file.index.end() && std::binary_search(conv2d.begin(), conv2d.end(), it->second);
    }

    packed_vec vec(std::string const& name) {
        gguf_tensor const& t = file.tensor(name);
        packed_vec v;
        v.n = (int)t.n_elements();
        v.off = ab.alloc((size_t)v.n * 4);
        if (with_data) {
            std::vector<float> f = to_f32(t);
            memcpy(ab.data.data() + v.off, f.data(), f.size() * 4);
        }
        return v;
    }

    // rows[n_real][k_real] f32 -> f16 [N][K] zero padded; bias f32 [N] zero padded
    packed_gemm matrix(const float* rows, int n_real, int k_real, const float* bias, int n_bias, int n_align) {
        packed_gemm g;
        g.n_real = n_real;
        g.k_real = k_real;
        g.N = round_up(n_real, n_align);
        g.K = round_up(k_real, 64);
        g.w = ab.alloc((size_t)g.N * g.K * 2);
        if (with_data && rows) {
            uint16_t* w = reinterpret_cast<uint16_t*>(ab.data.data() + g.w);
            for (int n = 0; n < n_real; ++n)
                for (int k = 0; k < k_real; ++k) w[(size_t)n * g.K + k] = f32_to_f16(rows[(size_t)n * k_real + k]);
        }
        if (n_bias > 0) {
            g.b = ab.alloc((size_t)g.N * 4);
            if (with_data && bias) memcpy(ab.data.data() + g.b, bias, (size_t)n_bias * 4);
        }
        return g;
    }

    // linear(): weight ggml ne [K, N] == torch [N][K] (reference nn.cpp:6-12)
    packed_gemm linear(std::string const& prefix, int n_align = 32) {
        gguf_tensor const& w = file.tensor(prefix + ".weight");
        gguf_tensor const* b = file.find(prefix + ".bias");
        int K = (int)w.ne[0], N = (int)w.ne[1];
        std::vector<float> wf, bf;
        if (with_data) {
            wf = to_f32(w);
            if (b) bf = to_f32(*b);
        }
        return matrix(with_data ? wf.data() : nullptr, N, K, b && with_data ? bf.data() : nullptr, b ? N : 0, n_align);
    }

    // unpadded f16 rows [N][K] of a linear weight (as stored in the file when it is f16) and its f32 bias
    void linear_rows(std::string const& prefix, std::vector<uint16_t>& rows, std::vector<float>& bias, int& N, int& K) {
        gguf_tensor const& w = file.tensor(prefix + ".weight");
        K = (int)w.ne[0]; N = (int)w.ne[1];
        if (!with_data) return;
        std::vector<float> wf = to_f32(w), bf = to_f32(file.tensor(prefix + ".bias"));
        size_t r0 = rows.size();
        rows.resize(r0 + wf.size());
        for (size_t i = 0; i < wf.size(); ++i) rows[r0 + i] = f32_to_f16(wf[i]);
        bias.insert(bias.end(), bf.begin(), bf.end());
    }
    void append_vec(std::vector<float>& dst, std::string const& name) {
        if (!with_data) return;
        std::vector<float> v = to_f32(file.tensor(name));
        dst.insert(dst.end(), v.begin(), v.end());
    }
    size_t put_floats(std::vector<float> const& v, size_t n) {
        size_t off = ab.alloc(n * 4);
        if (with_data) {
            if (v.size() != n) throw except("internal: packed vector has %zu floats, expected %zu", v.size(), n);
            memcpy(ab.data.data() + off, v.data(), n * 4);
        }
        return off;
    }

    // three linears concatenated along N (fused QKV projection)
    packed_gemm linear3(std::string const& a, std::string const& b, std::string const& c) {
        gguf_tensor const& wa = file.tensor(a + ".weight");
        int K = (int)wa.ne[0], N = (int)wa.ne[1];
        std::vector<float> rows, bias;
        if (with_data) {
            for (std::string const* p : {&a, &b, &c}) {
                gguf_tensor const& w = file.tensor(*p + ".weight");
                if (w.ne[0] != K || w.ne[1] != N) throw except("qkv weights of %s differ in shape", p->c_str());
                std::vector<float> wf = to_f32(w), bf = to_f32(file.tensor(*p + ".bias"));
                rows.insert(rows.end(), wf.begin(), wf.end());
                bias.insert(bias.end(), bf.begin(), bf.end());
            }
        }
        return matrix(with_data ? rows.data() : nullptr, 3 * N, K, with_data ? bias.data() : nullptr, 3 * N, 32);
    }

    // conv kernel as [Cout][kh][kw][Cin] f32. A tensor listed in <arch>.conv2d_weights of a WHCN
    // file is stored [kw,kh,Cin,Cout] (torch OIHW) and permuted here, exactly the tensors
    // model_transfer permutes (ml.cpp:462-502); anything else is already CWHN [Cin,kw,kh,Cout].
    std::vector<float> conv_ohwi(std::string const& name, int& cout, int& kh, int& kw, int& cin) {
        gguf_tensor const& t = file.tensor(name);
        bool permute = file_whcn && is_listed_conv2d(name);
        if (permute) { kw = (int)t.ne[0]; kh = (int)t.ne[1]; cin = (int)t.ne[2]; cout = (int)t.ne[3]; }
        else { cin = (int)t.ne[0]; kw = (int)t.ne[1]; kh = (int)t.ne[2]; cout = (int)t.ne[3]; }
        if (!with_data) return {};
        std::vector<float> src = to_f32(t);
        if (!permute) return src;
        std::vector<float> dst(src.size());
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < cin; ++c)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x)
                        dst[(((size_t)o * kh + y) * kw + x) * cin + c] = src[(((size_t)o * cin + c) * kh + y) * kw + x];
        return dst;
    }

    packed_gemm conv(std::string const& prefix, int n_align = 32, int* out_k = nullptr, int* out_cin = nullptr) {
        int cout, kh, kw, cin;
        std::vector<float> w = conv_ohwi(prefix + ".weight", cout, kh, kw, cin);
        gguf_tensor const* b = file.find(prefix + ".bias");
        std::vector<float> bf;
        if (b && with_data) bf = to_f32(*b);
        if (out_k) *out_k = kh;
        if (out_cin) *out_cin = cin;
        packed_gemm g = matrix(with_data ? w.data() : nullptr, cout, kh * kw * cin, b && with_data ? bf.data() : nullptr, b ? cout : 0, n_align);
        if (kh == 3 && kw == 3 && cin % 16 == 0 && (cout == 32 || cout == 64) && g.N == cout) {
            // second packing for the LDS-ring conv kernel (kernels_dconv.hip), used on the large DPT maps. Cin = 48
            // (neck conv 0) is padded to 64 with zero weights: the kernel then reads 16 channels of the next pixel
            // (or the descriptor's zero fill at the end of the map) against zeros.
            g.d_cin = round_up(cin, 32);
            g.dw = ab.alloc((size_t)g.d_cin * 9 * cout * 2);
            if (with_data) {
                uint16_t* dst = reinterpret_cast<uint16_t*>(ab.data.data() + g.dw);
                for (int n = 0; n < cout; ++n)
                    for (int tap = 0; tap < 9; ++tap)
                        for (int c = 0; c < cin; ++c) {
                            const size_t row = ((size_t)(c / 32) * 9 + tap) * cout + n;
                            dst[row * 32 + (size_t)(((c % 32) / 8) ^ ((n >> 2) & 3)) * 8 + c % 8] = f32_to_f16(w[((size_t)n * 9 + tap) * cin + c]);
                        }
            }
        }
        return g;
    }

    // conv_transpose_2d with kernel == stride (reference nn.cpp:117-129, weight ne [kw,kh,Cout,Cin] ==
    // torch [Cin][Cout][kh][kw], never permuted: convert.py:466-467). Row n = (dy*s+dx)*Cout + co.
    packed_gemm conv_transpose(std::string const& prefix, int stride, int k_pad_to) {
        gguf_tensor const& t = file.tensor(prefix + ".weight");
        int kw = (int)t.ne[0], kh = (int)t.ne[1], cout = (int)t.ne[2], cin = (int)t.ne[3];
        if (kw != stride || kh != stride) throw except("%s: conv_transpose kernel %dx%d != stride %d is not supported", prefix.c_str(), kw, kh, stride);
        gguf_tensor const* b = file.find(prefix + ".bias");
        int n_real = stride * stride * cout;
        std::vector<float> rows, bias;
        if (with_data) {
            std::vector<float> src = to_f32(t);
            rows.assign((size_t)n_real * k_pad_to, 0.0f);
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cout; ++co)
                    for (int dy = 0; dy < kh; ++dy)
                        for (int dx = 0; dx < kw; ++dx)
                            rows[((size_t)(dy * stride + dx) * cout + co) * k_pad_to + ci] = src[(((size_t)ci * cout + co) * kh + dy) * kw + dx];
            if (b) {
                std::vector<float> bf = to_f32(*b);
                bias.resize(n_real);
                for (int n = 0; n < n_real; ++n) bias[n] = bf[n % cout];
            }
        }
        return matrix(with_data ? rows.data() : nullptr, n_real, k_pad_to, b && with_data ? bias.data() : nullptr, b ? n_real : 0, 32);
    }
};

} // namespace

depthany_model* depthany_load_model(char const* filepath, backend_device const& dev, int flags) {
    const bool with_data = !(flags & load_no_upload);
    model_file file = model_load(filepath, /*header_only=*/!with_data);
    if (file.arch() != "depthanything")
        throw except("Model %s has architecture '%.*s', expected 'depthanything'", filepath, (int)file.arch().size(), file.arch().data());

    auto model = std::make_unique<depthany_model>();
    model->backend = &dev;
    model->params = depthany_detect_params(file);
    depthany_params const& P = model->params;
    const int D = P.dino.embed_dim;
    if (D % P.dino.n_heads != 0 || D / P.dino.n_heads != 64)
        throw except("Unsupported DINO head dim %d (this backend implements head_dim 64)", P.dino.n_heads ? D / P.dino.n_heads : 0);
    if (D % 128 != 0) throw except("Unsupported embed dim %d (must be a multiple of 128)", D);

    arena_builder ab;
    packer pk(file, ab, with_data);
    depthany_weights& Wt = model->weights;

    const std::string e = "backbone.embeddings";
    {
        // patch embed: stored NHWC [Cout][ps][ps][3] (convert.py:463-465), flattened k = (ky,kx,c)
        int k, cin;
        Wt.patch = pk.conv(e + ".patch_embeddings.projection", 128, &k, &cin);
        if (k != P.dino.patch_size || cin != 3) throw except("patch embedding kernel %dx%dx%d does not match patch size %d", k, k, cin, P.dino.patch_size);
        Wt.cls = pk.vec(e + ".cls_token");
        Wt.pos = pk.vec(e + ".position_embeddings");
        Wt.pos_tokens = (int)file.tensor(e + ".position_embeddings").ne[1];
    }
    Wt.layers.resize(P.dino.n_layers);
    for (int i = 0; i < P.dino.n_layers; ++i) {
        std::string p = "backbone.encoder.layer." + std::to_string(i);
        dino_layer_weights& L = Wt.layers[i];
        L.ln1_w = pk.vec(p + ".norm1.weight");
        L.ln1_b = pk.vec(p + ".norm1.bias");
        L.qkv = pk.linear3(p + ".attention.attention.query", p + ".attention.attention.key", p + ".attention.attention.value");
        L.out = pk.linear(p + ".attention.output.dense");
        L.lambda1 = pk.vec(p + ".layer_scale1.lambda1");
        L.ln2_w = pk.vec(p + ".norm2.weight");
        L.ln2_b = pk.vec(p + ".norm2.bias");
        L.fc1 = pk.linear(p + ".mlp.fc1");
        L.fc2 = pk.linear(p + ".mlp.fc2");
        L.lambda2 = pk.vec(p + ".layer_scale2.lambda1");
    }
    Wt.final_ln_w = pk.vec("backbone.layernorm.weight");
    Wt.final_ln_b = pk.vec("backbone.layernorm.bias");
    // second packing of the encoder for the block kernel (kernels_block16.hip): slab streams in the kernel's order of use
    Wt.use_block = P.dino.n_layers > 0 && vx_dino_block_supported(D, Wt.layers[0].fc1.n_real, D / P.dino.n_heads) != 0;
    if (Wt.use_block) {
        for (int i = 0; i < P.dino.n_layers; ++i) {
            std::string p = "backbone.encoder.layer." + std::to_string(i);
            dino_layer_weights& L = Wt.layers[i];
            std::vector<uint16_t> wo, w1, w2, wqkv;
            std::vector<float> bo, b1, b2, bqkv, vm, vq;
            int N, K;
            pk.linear_rows(p + ".attention.output.dense", wo, bo, N, K);
            pk.linear_rows(p + ".mlp.fc1", w1, b1, N, K);
            pk.linear_rows(p + ".mlp.fc2", w2, b2, N, K);
            for (const char* n : {".attention.attention.query", ".attention.attention.key", ".attention.attention.value"}) pk.linear_rows(p + n, wqkv, bqkv, N, K);
            L.blk_mlp = ab.alloc(vx_dino_block_mlp_bytes());
            L.blk_qkv = ab.alloc(vx_dino_block_qkv_bytes());
            if (with_data) {
                std::vector<float> l1, l2;
                pk.append_vec(l1, p + ".layer_scale1.lambda1");
                pk.append_vec(l2, p + ".layer_scale2.lambda1");
                // the kernel keeps the residual stream in its accumulators: LayerScale is folded into the two residual products (exact in
                // real arithmetic; the scaled weights are re-rounded to f16): x1 = x + Wo' att + bo', x2 = x1 + W2' h + b2'
                const int Hd = (int)b1.size();
                for (int n = 0; n < D; ++n) {
                    for (int k = 0; k < D; ++k) wo[(size_t)n * D + k] = f32_to_f16(f16_to_f32(wo[(size_t)n * D + k]) * l1[n]);
                    for (int k = 0; k < Hd; ++k) w2[(size_t)n * Hd + k] = f32_to_f16(f16_to_f32(w2[(size_t)n * Hd + k]) * l2[n]);
                    bo[n] *= l1[n];
                    b2[n] *= l2[n];
                }
                VX(vx_dino_block16_pack_mlp(wo.data(), w1.data(), w2.data(), ab.data.data() + L.blk_mlp));
                VX(vx_dino_block16_pack_qkv(wqkv.data(), ab.data.data() + L.blk_qkv));
                vm = bo;
                vm.insert(vm.end(), l1.begin(), l1.end());
                pk.append_vec(vm, p + ".norm2.weight");
                pk.append_vec(vm, p + ".norm2.bias");
                vm.insert(vm.end(), b1.begin(), b1.end());
                vm.insert(vm.end(), b2.begin(), b2.end());
                vm.insert(vm.end(), l2.begin(), l2.end());
                pk.append_vec(vq, p + ".norm1.weight");
                pk.append_vec(vq, p + ".norm1.bias");
                vq.insert(vq.end(), bqkv.begin(), bqkv.end());
            }
            L.vec_mlp = pk.put_floats(vm, (size_t)7 * D + Wt.layers[0].fc1.n_real - D); // bo l1 g2 b2 | b1 | bfc2 l2
            L.vec_qkv = pk.put_floats(vq, (size_t)5 * D);
        }
        std::vector<float> vt;
        pk.append_vec(vt, "backbone.layernorm.weight");
        pk.append_vec(vt, "backbone.layernorm.bias");
        Wt.vec_tap = pk.put_floats(vt, (size_t)2 * D);
    }

    const std::string r = "neck.reassemble_stage.layers.";
    for (int i = 0; i < 4; ++i) {
        Wt.re_proj[i] = pk.conv(r + std::to_string(i) + ".projection", 64);
        Wt.neck_c[i] = Wt.re_proj[i].n_real;
    }
    Wt.re_up0 = pk.conv_transpose(r + "0.resize", 4, Wt.re_proj[0].N);
    Wt.re_up1 = pk.conv_transpose(r + "1.resize", 2, Wt.re_proj[1].N);
    Wt.re_down3 = pk.conv(r + "3.resize");
    if (Wt.neck_c[0] % 8 || Wt.neck_c[1] % 8 || Wt.neck_c[2] % 64 || Wt.neck_c[3] % 64)
        throw except("Unsupported neck channel counts %d/%d/%d/%d", Wt.neck_c[0], Wt.neck_c[1], Wt.neck_c[2], Wt.neck_c[3]);
    for (int i = 0; i < 4; ++i) Wt.neck_conv[i] = pk.conv("neck.convs." + std::to_string(i));
    Wt.fusion_c = Wt.neck_conv[0].n_real;
    for (int i = 0; i < 4; ++i) {
        std::string p = "neck.fusion_stage.layers." + std::to_string(i);
        fusion_weights& F = Wt.fusion[i];
        F.proj = pk.conv(p + ".projection");
        // residual_layer1 of fusion layer 0 exists in the file but is never executed
        // (reference depth-anything.cpp:27-30, 74); it is packed anyway to keep the arena layout uniform
        F.rl1_c1 = pk.conv(p + ".residual_layer1.convolution1");
        F.rl1_c2 = pk.conv(p + ".residual_layer1.convolution2");
        F.rl2_c1 = pk.conv(p + ".residual_layer2.convolution1");
        F.rl2_c2 = pk.conv(p + ".residual_layer2.convolution2");
    }
    if (Wt.fusion_c % 32) throw except("Unsupported fusion width %d", Wt.fusion_c);
    Wt.head1 = pk.conv("head.conv1");
    Wt.head2 = pk.conv("head.conv2");
    Wt.head_c = Wt.head1.n_real;
    if (Wt.head2.n_real == 32 && Wt.head2.k_real == 288 && Wt.head2.N == 32) { // 3x3, 32 -> 32: the head kernel's operand
        Wt.head2_frag = ab.alloc(vx_headconv_frag_bytes());
        if (with_data) VX(vx_headconv_pack(ab.data.data() + Wt.head2.w, Wt.head2.K, ab.data.data() + Wt.head2_frag));
    }
    if (Wt.head_c != 8 && Wt.head_c != 16 && Wt.head_c != 32 && Wt.head_c != 64) throw except("Unsupported head width %d", Wt.head_c);
    Wt.head3_w = pk.vec("head.conv3.weight");
    // the scalar bias of the final 1x1 conv lives in the arena too, so that an arena received by
    // RCCL broadcast is self-contained (depthany_weights_ready reads it back)
    Wt.head3_b_off = ab.alloc(16);
    if (with_data) {
        Wt.head3_b = to_f32(file.tensor("head.conv3.bias"))[0];
        memcpy(ab.data.data() + Wt.head3_b_off, &Wt.head3_b, 4);
    }

    device_turn turn(dev);
    VX(vx_dconv_prepare());
    for (void*& s : model->aux_stream) VX(vx_stream_create(&s));
    VX(vx_event_create(&model->fork_event));
    for (void*& e : model->join_event) VX(vx_event_create(&e));
    model->weight_arena.bytes = round_up<size_t>(ab.data.size(), 256);
    VX(vx_malloc(&model->weight_arena.ptr, model->weight_arena.bytes));
    if (with_data) {
        VX(vx_memcpy_h2d(model->weight_arena.ptr, ab.data.data(), ab.data.size(), dev.stream));
        VX(vx_stream_sync(dev.stream));
        model->weights_uploaded = true;
    }
    return model.release();
}

void depthany_weights_ready(depthany_model& m) {
    device_turn turn(*m.backend);
    VX(vx_memcpy_d2h(&m.weights.head3_b, static_cast<const uint8_t*>(m.weight_arena.ptr) + m.weights.head3_b_off, 4, m.backend->stream));
    m.weights_uploaded = true;
}

depthany_model::~depthany_model() {
    delete shard_pipeline; // (its executor 0 is this model: only the slots, streams and clones go)
    shard_pipeline = nullptr;
    if (ws.graph_exec) vx_graph_destroy(ws.graph_exec);
    for (void* s : aux_stream) vx_stream_destroy(s);
    vx_event_destroy(fork_event);
    for (void* e : join_event) vx_event_destroy(e);
    for (auto& c : capture_bufs) vx_free(c.second.dev);
    vx_free(ws.arena.ptr);
    if (owns_weights) vx_free(weight_arena.ptr);
}

depthany_model* depthany_clone_executor(depthany_model const& src) {
    if (!src.weights_uploaded) throw except("depthany: cannot clone an executor before the weights are on the device");
    device_turn turn(*src.backend);
    auto m = std::make_unique<depthany_model>();
    m->backend = src.backend;
    m->params = src.params;
    m->weights = src.weights;
    m->weight_arena = src.weight_arena;
    m->owns_weights = false;
    m->weights_uploaded = true;
    m->use_graph = src.use_graph;
    m->schedule = src.schedule;
    for (void*& s : m->aux_stream) VX(vx_stream_create(&s));
    VX(vx_event_create(&m->fork_event));
    for (void*& e : m->join_event) VX(vx_event_create(&e));
    return m.release();
}

//
// workspace

namespace {

struct ws_layout {
    size_t total = 0;
    std::vector<std::pair<std::string, std::pair<size_t, size_t>>> items;
    void add(std::string name, size_t bytes) {
        size_t off = round_up<size_t>(total, 256);
        items.push_back({std::move(name), {off, bytes}});
        total = off + bytes;
    }
};

// bicubic (a = -0.75, half-pixel centres, clamped taps) resize of the patch position embeddings,
// the host-side counterpart of dino::interpolate_pos_encoding (reference dino.cpp:10-30)
void interpolate_pos(const float* pos /*[1+n, D]*/, int n_side, int D, int th, int tw, float* out /*[1+th*tw, D]*/) {
    memcpy(out, pos, (size_t)D * 4);
    const float* src = pos + D;
    auto coeffs = [](float t, float c[4]) {
        const float a = -0.75f;
        float x;
        x = t + 1.0f; c[0] = ((a * x - 5.0f * a) * x + 8.0f * a) * x - 4.0f * a;
        x = t;        c[1] = ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
        x = 1.0f - t; c[2] = ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
        x = 2.0f - t; c[3] = ((a * x - 5.0f * a) * x + 8.0f * a) * x - 4.0f * a;
    };
    float sfy = (float)th / (float)n_side, sfx = (float)tw / (float)n_side;
    for (int oy = 0; oy < th; ++oy) {
        float sy = ((float)oy + 0.5f) / sfy - 0.5f;
        int iy = (int)std::floor(sy);
        float cy[4];
        coeffs(sy - (float)iy, cy);
        for (int ox = 0; ox < tw; ++ox) {
            float sx = ((float)ox + 0.5f) / sfx - 0.5f;
            int ix = (int)std::floor(sx);
            float cx[4];
            coeffs(sx - (float)ix, cx);
            float* o = out + ((size_t)1 + (size_t)oy * tw + ox) * D;
            for (int d = 0; d < D; ++d) o[d] = 0.0f;
            for (int j = 0; j < 4; ++j) {
                int yy = std::clamp(iy - 1 + j, 0, n_side - 1);
                for (int i = 0; i < 4; ++i) {
                    int xx = std::clamp(ix - 1 + i, 0, n_side - 1);
                    float wgt = cy[j] * cx[i];
                    const float* s = src + ((size_t)yy * n_side + xx) * D;
                    for (int d = 0; d < D; ++d) o[d] += wgt * s[d];
                }
            }
        }
    }
}

} // namespace

void depthany_reserve(depthany_model& m, int B, int W, int H) {
    depthany_params const& P = m.params;
    depthany_weights const& Wt = m.weights;
    const int ps = P.dino.patch_size;
    if (B <= 0 || W <= 0 || H <= 0) throw except("depthany: invalid batch/extent %d x %dx%d", B, W, H);
    // the workspace caches the (possibly resized) position embeddings, so the weights must be in place first
    if (!m.weights_uploaded) throw except("depthany: weights were not uploaded (VISP_LOAD_NO_UPLOAD): fill the arena and call visp_depthany_weights_ready first");
    if (W % ps || H % ps) throw except("depthany: extent %dx%d is not a multiple of the patch size %d", W, H, ps);
    if (m.ws.B == B && m.ws.W == W && m.ws.H == H && m.ws.arena.ptr) return;

    const int pw = W / ps, ph = H / ps, Pn = pw * ph, T = Pn + 1, D = P.dino.embed_dim;
    const long M = (long)B * T;
    const int F = Wt.fusion_c, HC = Wt.head_c;
    const int h3 = (ph + 2 - 3) / 2 + 1, w3 = (pw + 2 - 3) / 2 + 1;

    ws_layout L;
    L.add("rgb", (size_t)B * H * W * 3);
    L.add("patches", (size_t)B * Pn * Wt.patch.K * 2);
    L.add("pos", (size_t)T * D * 4);
    L.add("x", (size_t)M * D * 4);
    L.add("ln", (size_t)M * D * 2);
    L.add("q", (size_t)M * D * 2);
    L.add("k", (size_t)M * D * 2);
    L.add("vt", (size_t)M * D * 2);
    L.add("att", (size_t)M * D * 2);
    L.add("hidden", (size_t)M * Wt.layers[0].fc1.N * 2);
    for (int j = 0; j < 4; ++j) L.add("feat" + std::to_string(j), (size_t)M * D * 2);
    for (int j = 0; j < 4; ++j) L.add("r" + std::to_string(j), (size_t)B * Pn * Wt.re_proj[j].N * 2);
    L.add("l0", (size_t)B * 16 * Pn * Wt.neck_c[0] * 2);
    L.add("l1", (size_t)B * 4 * Pn * Wt.neck_c[1] * 2);
    L.add("l3", (size_t)B * h3 * w3 * Wt.neck_c[3] * 2);
    L.add("c0", (size_t)B * 16 * Pn * F * 2);
    L.add("c1", (size_t)B * 4 * Pn * F * 2);
    L.add("c2", (size_t)B * Pn * F * 2);
    L.add("c3", (size_t)B * h3 * w3 * F * 2);
    for (const char* n : {"t1", "t2", "t3"}) L.add(n, (size_t)B * 16 * Pn * F * 2);
    L.add("up", (size_t)B * 64 * Pn * F * 2);
    L.add("fused", (size_t)B * 64 * Pn * F * 2);
    L.add("h1", (size_t)B * 64 * Pn * HC * 2);
    L.add("hup", (size_t)B * H * W * HC * 2);
    L.add("h2", (size_t)B * H * W * HC * 2);
    L.add("depth", (size_t)B * H * W * 4);
    L.add("out", (size_t)B * H * W * 4);
    L.add("minmax", (size_t)B * 2 * 4);

    device_turn turn(*m.backend);
    if (m.ws.graph_exec) { vx_graph_destroy(m.ws.graph_exec); m.ws.graph_exec = nullptr; }
    if (m.ws.arena.bytes < L.total) {
        VX(vx_free(m.ws.arena.ptr));
        m.ws.arena = {};
        VX(vx_malloc(&m.ws.arena.ptr, L.total));
        m.ws.arena.bytes = L.total;
    }
    m.ws.buf.clear();
    m.ws.bytes.clear();
    for (auto& it : L.items) {
        m.ws.buf[it.first] = static_cast<uint8_t*>(m.ws.arena.ptr) + it.second.first;
        m.ws.bytes[it.first] = it.second.second;
    }
    m.ws.B = B; m.ws.W = W; m.ws.H = H;
    void* s = m.backend->stream;

    // position embeddings for this grid (dino.cpp:10-30): as stored, or bicubic-resized on the host
    const uint8_t* wa = static_cast<const uint8_t*>(m.weight_arena.ptr);
    if (T == Wt.pos_tokens && W == H) {
        VX(vx_memcpy_d2d(m.ws.buf["pos"], wa + Wt.pos.off, (size_t)T * D * 4, s));
    } else {
        int n = Wt.pos_tokens - 1;
        int n_side = (int)(std::sqrt((float)n) + 0.01f);
        std::vector<float> stored((size_t)Wt.pos_tokens * D), resized((size_t)T * D);
        VX(vx_memcpy_d2h(stored.data(), wa + Wt.pos.off, stored.size() * 4, s));
        interpolate_pos(stored.data(), n_side, D, ph, pw, resized.data());
        VX(vx_memcpy_h2d(m.ws.buf["pos"], resized.data(), resized.size() * 4, s));
    }
    VX(vx_stream_sync(s));
}

//
// executor

namespace {

struct exec_ctx {
    depthany_model& m;
    void* stream;
    const uint8_t* wa;
    std::vector<std::pair<std::string, void*>> marks; // timing
    std::vector<timing_entry> acc;

    const void* wptr(size_t off) const { return wa + off; }
    const float* fptr(packed_vec const& v) const { return reinterpret_cast<const float*>(wa + v.off); }
    int sub_b0 = 0; // first image of the sub-batch being scheduled: every activation buffer is image-major, so a sub-batch is an offset
    void* buf(const char* name) {
        uint8_t* base = static_cast<uint8_t*>(m.ws.buf.at(name));
        if (sub_b0 == 0 || strcmp(name, "pos") == 0) return base;
        return base + (size_t)sub_b0 * (m.ws.bytes.at(name) / (size_t)m.ws.B);
    }

    void mark(const char* name, int launches, double flops, double bytes) {
        if (!m.timing) return;
        void* ev = nullptr;
        VX(vx_event_create(&ev));
        VX(vx_event_record(ev, stream));
        static const bool detail = getenv("VISP_TIMING_DETAIL") != nullptr; // one row per launch (tools/dpt_launches.py)
        if (detail) {
            char nm[64];
            snprintf(nm, sizeof nm, "%s#%zu", name, marks.size());
            marks.push_back({nm, ev});
            acc.push_back({nm, 0, launches, flops, bytes});
            return;
        }
        marks.push_back({name, ev});
        acc.push_back({name, 0, launches, flops, bytes});
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
            if (marks[i].first.rfind("__end", 0) == 0) continue; // the last launch of a sub-batch ends at its stream's end mark, recorded next
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

    void capture(const char* name, const void* dev, std::array<int64_t, 4> shape, bool f16) {
        if (!m.captures) return;
        size_t n = (size_t)(shape[0] * shape[1] * shape[2] * shape[3]);
        size_t bytes = n * (f16 ? 2 : 4);
        capture_entry& c = m.capture_bufs[name];
        if (c.dev) vx_free(c.dev);
        VX(vx_malloc(&c.dev, bytes));
        VX(vx_memcpy_d2d(c.dev, dev, bytes, stream));
        for (int i = 0; i < 4; ++i) c.shape[i] = shape[i];
        c.f16 = f16;
    }

    vx_gemm_args base(packed_gemm const& g, long M) {
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.W = wptr(g.w);
        a.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b);
        a.M = (int)M;
        a.N = g.N;
        a.K = g.K;
        a.n_valid = g.n_real;
        return a;
    }
    void gemm(vx_gemm_args const& a) {
        // 3x3 convs with few channels at high resolution: halo-in-LDS kernel (tile 8x32, so only where the
        // map is wide enough for the edge tiles not to dominate); everything else: (implicit) GEMM
        if (a.conv_kh == 3 && a.conv_W >= 96 && vx_conv3x3_supported(&a)) VX(vx_conv3x3_f16(&a, stream));
        else VX(vx_gemm_f16(&a, stream));
    }

    // NHWC 3x3 (or kxk) convolution as implicit GEMM
    // bil_hs > 0: x is the low-resolution map [B, bil_hs, bil_ws, Cin] and the conv runs on its bilinear (align_corners) resize to
    // H x W, interpolated by the conv's halo loader (callers check bil_ok first)
    void conv(packed_gemm const& g, const void* x, int B, int H, int W, int Cin, int k, int stride, int pad, void* y, int ldo,
              int epi, bool a_relu, bool relu, const void* res1, const void* res2, const char* group, int bil_hs = 0, int bil_ws = 0) {
        int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
        long M = (long)B * OH * OW;
        vx_gemm_args a = base(g, M);
        a.A = x;
        a.conv_kh = a.conv_kw = k;
        a.conv_stride = stride;
        a.conv_pad = pad;
        a.conv_H = H; a.conv_W = W; a.conv_Cin = Cin; a.conv_OH = OH; a.conv_OW = OW;
        a.a_relu = a_relu;
        a.epi = epi;
        a.relu = relu;
        a.out = y;
        a.ldo = ldo;
        a.res1 = res1; a.res2 = res2;
        mark(group, 1, 2.0 * M * g.n_real * g.k_real, (double)B * H * W * Cin * 2 + (double)M * g.n_real * 2 + (double)g.N * g.K * 2);
        if (dconv_ok(g, k, stride, pad, W, Cin, epi) && !relu) {
            vx_dconv_args d = dconv_base(g, x, B, H, W, Cin);
            d.bil_hs = bil_hs; d.bil_ws = bil_ws;
            d.epi = VX_DC_F16;
            d.act = epi == VX_EPI_F16_RELU ? 2 : 0;
            d.a_relu = a_relu;
            d.s1 = d.s2 = 1.0f;
            d.res1 = res1; d.res1_pix = ldo; d.res1_plane = 32;
            d.res2 = res1 ? res2 : nullptr; d.res2_pix = ldo; d.res2_plane = 32;
            if (!res1 && res2) { d.res1 = res2; d.res2 = nullptr; }
            d.out = y; d.out_pix = ldo; d.out_plane = 32;
            VX(vx_dconv3x3_f16(&d, stream));
            return;
        }
        if (bil_hs > 0) throw except("depthany: internal: bilinear input without the LDS-ring conv");
        gemm(a);
    }
    // may conv `g` (3x3 / 1 / 1 on an H x W map) take its input as the bilinear resize of an hs x ws map?
    static bool bil_ok(packed_gemm const& g, int H, int W, int Cin, int epi, int hs, int ws) {
        static const bool off = getenv("VISP_NO_BIL_FUSE") != nullptr;
        return !off && dconv_ok(g, 3, 1, 1, W, Cin, epi) && vx_dconv_bilinear_supported(g.N, H, W, hs, ws);
    }
    // 3x3 / stride 1 / pad 1 convs go to the persistent LDS-ring kernel written for the ESRGAN row (1.4-1.9x the halo kernel on
    // maps >= 64 wide, tools/conv_compare.py); it reads and writes the NHWC maps in place through its pixel / plane strides.
    // Round 3: also on the 37^2 and 19^2 maps (its 16x32 tiles are half empty there, and still: neck conv 192->64 @37^2 50 -> 27 us,
    // 384->64 @19^2 80 -> 40 us, the six small residual-unit convs 23 -> 17.5 us each at batch 32: the implicit-GEMM kernel runs one
    // workgroup per CU with a single LDS stage on these grids and every k-step is an exposed L2 round trip; profiles/r03_dpt_launches.txt)
    static bool dconv_ok(packed_gemm const& g, int k, int stride, int pad, int W, int Cin, int epi) {
        static const bool off = getenv("VISP_NO_DCONV") != nullptr;
        static const int min_w = getenv("VISP_DCONV_MINW") ? atoi(getenv("VISP_DCONV_MINW")) : 16;
        return !off && g.dw != SIZE_MAX && k == 3 && stride == 1 && pad == 1 && W >= min_w && round_up(Cin, 32) == g.d_cin &&
               (epi == VX_EPI_F16 || epi == VX_EPI_F16_RELU || epi == VX_EPI_F16_ADD || epi == VX_EPI_HEAD_OUT);
    }
    vx_dconv_args dconv_base(packed_gemm const& g, const void* x, int B, int H, int W, int Cin) {
        vx_dconv_args d;
        memset(&d, 0, sizeof d);
        d.x = x; d.x_pix = Cin; d.x_plane = 32; d.cin = g.d_cin;
        d.B = B; d.H = H; d.W = W;
        d.w = wptr(g.dw); d.bias = g.b == SIZE_MAX ? nullptr : reinterpret_cast<const float*>(wa + g.b); d.cout = g.N;
        return d;
    }
};

} // namespace

static void run_forward(depthany_model& m, const void* rgb, void* out_dev, void* raw_out_dev, void* stream) {
    depthany_params const& P = m.params;
    depthany_weights const& Wt = m.weights;
    const int B = m.ws.B, W = m.ws.W, H = m.ws.H;
    const int ps = P.dino.patch_size, pw = W / ps, ph = H / ps, Pn = pw * ph, T = Pn + 1, D = P.dino.embed_dim, NH = P.dino.n_heads;
    const long M = (long)B * T, MP = (long)B * Pn;
    const int F = Wt.fusion_c, HC = Wt.head_c;
    exec_ctx c{m, stream, static_cast<const uint8_t*>(m.weight_arena.ptr), {}, {}};

    float* x = static_cast<float*>(c.buf("x"));
    (void)0;
    const float* pos = static_cast<const float*>(c.buf("pos"));

    // ---- depthany_process_input (depth-anything.cpp:130-140) fused with the patch im2col
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float inv_std[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
    c.mark("preprocess", 2, 0, (double)B * H * W * 3 + (double)MP * Wt.patch.K * 2);
    VX(vx_preprocess_patches(static_cast<const uint8_t*>(rgb), c.buf("patches"), B, H, W, ps, Wt.patch.K, mean, inv_std, stream));
    VX(vx_write_cls_rows(x, c.fptr(Wt.cls), pos, B, T, D, stream));

    // ---- dino::prepare_tokens (dino.cpp:32-46): patch GEMM, epilogue adds bias + pos-embed, writes token rows
    {
        vx_gemm_args a = c.base(Wt.patch, MP);
        a.A = c.buf("patches");
        a.lda = Wt.patch.K;
        a.epi = VX_EPI_TOKENS;
        a.out = x;
        a.ldo = D;
        a.pos = pos;
        a.tokens_P = Pn;
        c.mark("patch_embed", 1, 2.0 * MP * D * Wt.patch.k_real, (double)MP * Wt.patch.K * 2 + (double)M * D * 4);
        c.gemm(a);
    }
    c.capture("tokens", x, {B, T, D, 1}, false);

    // ---- dino::layer x n_layers (dino.cpp:76-90)
    // (GEMM schedule: the residual adds are the out-proj / fc2 GEMMs' read-modify-write epilogues. Deferring them into the following
    // LayerNorm was measured in round 1 -- GEMMs -0.36 ms, LayerNorms +0.42 ms per step -- and removed in round 4.)
    // attention() scales the scores by 1/sqrt(head_dim) (nn.cpp:232-233); the attention kernel works in the exp2 domain, so log2(e)
    // rides along in the same factor (VX_ATTN_Q_SCALE for head_dim 64)
    const float q_scale = 1.4426950408889634f / std::sqrt((float)D / (float)NH);
    // The encoder of images [b0, b0 + nb) on stream `strm`. Every activation buffer is image-major, so a sub-batch is a row
    // offset into the same workspace; VISP_SPLIT=n runs n sub-batches on parallel streams (captured as parallel branches of
    // the hipGraph) so that kernels with different bottlenecks -- HBM-bound LayerNorms, VALU-bound attention, MFMA/LDS-bound
    // GEMMs -- of different sub-batches overlap.
    // Timing-only ablations (results invalid by construction): what the step would cost without a stage. They exist only in diagnostic
    // builds of the library (make ABLATE=1 -> -DVISP_TIMING_ABLATIONS); the shipped library has no switch that skips work.
#ifdef VISP_TIMING_ABLATIONS
    static const int ablate = getenv("VISP_ABLATE") ? atoi(getenv("VISP_ABLATE")) : 0; // 1 = no DPT, 2 = no bilinear launches, 4 = no attention
#else
    constexpr int ablate = 0;
#endif
    auto run_sub = [&](int b0, int nb, void* strm) {
    const int B = nb;
    const long M = (long)nb * T, MP = (long)nb * Pn;
    void* const stream = strm;
    c.stream = strm;
    c.sub_b0 = b0;
    float* const x = static_cast<float*>(c.buf("x"));
    auto sub = [&](const char* name, size_t) -> void* { return c.buf(name); };
    void* const ln = sub("ln", (size_t)D * 2);
    void* const qb = sub("q", (size_t)D * 2);
    void* const kb = sub("k", (size_t)D * 2);
    void* const vb = sub("vt", (size_t)D * 2);
    void* const attb = sub("att", (size_t)D * 2);
    void* const hidb = sub("hidden", (size_t)Wt.layers[0].fc1.N * 2);
    auto layernorm = [&](const float* w, const float* b, void* out) {
        c.mark("layernorm", 1, 0, (double)M * D * 6);
        VX(vx_layernorm_f32_f16(x, w, b, out, (int)M, D, 1e-6f, stream));
    };
    auto residual_gemm = [&](packed_gemm const& g, const void* A, int lda, packed_vec const& lambda, const char* group, double flops, double bytes) {
        vx_gemm_args a = c.base(g, M);
        a.A = A; a.lda = lda;
        a.epi = VX_EPI_RESID_F32;
        a.out = x; a.ldo = D;
        a.lambda = c.fptr(lambda);
        c.mark(group, 1, flops, bytes);
        c.gemm(a);
    };
    int tap = 0;
    // Token-stationary schedule (kernels_block16.hip): per layer one attention launch and ONE block launch that does the
    // output projection, both residual updates, LN2 + MLP, the tap's final LayerNorm and the next layer's LN1 + QKV for
    // 128 token rows per workgroup with everything but the weights in registers. Alone it is no faster than the launches it
    // replaces (381 vs 363 us per layer at batch 32, profiles/r02_block_kernel_anatomy.txt), but it has none of their HBM
    // round trips and, with the sub-batches below on parallel streams, its idle second round and memory phases are filled by
    // the other sub-batches' attention: 6.85 vs 7.21 ms per step (profiles/r02_split_streams.txt). Default where the model has
    // the kernel's shape; visp_depthany_set_schedule(model, 0) or VISP_NO_BLOCK=1 selects the GEMM schedule.
    static const bool block_off = getenv("VISP_NO_BLOCK") != nullptr;
    const bool use_block = Wt.use_block && m.schedule != 0 && !(block_off && m.schedule < 0);
    if (use_block) {
        const int hid = Wt.layers[0].fc1.n_real;
        auto block = [&](int li_mlp, int li_qkv, void* feat, const char* group) {
            vx_dino_block_args a;
            memset(&a, 0, sizeof a);
            a.x = x; a.M = (int)M; a.T = T; a.H = NH; a.q_scale = q_scale; a.eps = 1e-6f;
            double flops = 0, bytes = 0;
            if (li_mlp >= 0) {
                a.att = attb;
                a.w_mlp = c.wptr(Wt.layers[li_mlp].blk_mlp);
                a.vec_mlp = reinterpret_cast<const float*>(c.wptr(Wt.layers[li_mlp].vec_mlp));
                flops += 2.0 * M * D * (D + 2.0 * hid);
                bytes += (double)M * D * (2 + 4 * 4); // att in; x read, written, re-read, written
            } else {
                bytes += (double)M * D * 4;
            }
            if (li_qkv >= 0) {
                a.q = qb; a.k = kb; a.v = vb;
                a.w_qkv = c.wptr(Wt.layers[li_qkv].blk_qkv);
                a.vec_qkv = reinterpret_cast<const float*>(c.wptr(Wt.layers[li_qkv].vec_qkv));
                flops += 2.0 * M * D * 3.0 * D;
                bytes += (double)M * D * 2 * 3;
            }
            if (feat) {
                a.feat = feat;
                a.vec_tap = reinterpret_cast<const float*>(c.wptr(Wt.vec_tap));
                bytes += (double)M * D * 2;
            }
            c.mark(group, 1, flops, bytes);
            VX(vx_dino_block16_f16(&a, stream));
        };
        block(-1, 0, nullptr, "block_qkv0");
        int tap = 0;
        for (int i = 0; i < P.dino.n_layers; ++i) {
            c.mark("attention", 1, 4.0 * B * NH * (double)T * T * 64, (double)M * D * 2 * 4);
            if (!(ablate & 4)) VX(vx_attention_f16(qb, kb, vb, attb, B, NH, T, stream));
            // get_intermediate_layers (dino.cpp:100-107): every tap that names this layer (the first one is written by the kernel)
            void* first = nullptr;
            int tap0 = tap;
            for (int f = 0; f < 4; ++f)
                if (P.feature_layers[f] == i && tap < 4) {
                    std::string fb = "feat" + std::to_string(tap++);
                    if (!first) first = sub(fb.c_str(), (size_t)D * 2);
                }
            block(i, i + 1 < P.dino.n_layers ? i + 1 : -1, first, "block");
            for (int t2 = tap0 + 1; t2 < tap; ++t2) {
                std::string fb = "feat" + std::to_string(t2);
                VX(vx_memcpy_d2d(sub(fb.c_str(), (size_t)D * 2), first, (size_t)M * D * 2, stream));
            }
            if (m.captures) {
                std::string nm = "layer_" + std::to_string(i);
                c.capture(nm.c_str(), x, {B, T, D, 1}, false);
                for (int t2 = tap0; t2 < tap; ++t2) {
                    std::string fb = "feat" + std::to_string(t2), dn = "dino_layer_" + std::to_string(i);
                    c.capture(dn.c_str(), sub(fb.c_str(), (size_t)D * 2), {B, T, D, 1}, true);
                }
            }
        }
        if (tap != 4) throw except("depthany: expected 4 feature layers, found %d", tap);
    }
    for (int i = 0; i < (use_block ? 0 : P.dino.n_layers); ++i) {
        dino_layer_weights const& L = Wt.layers[i];
        layernorm(c.fptr(L.ln1_w), c.fptr(L.ln1_b), ln);
        {
            vx_gemm_args a = c.base(L.qkv, M);
            a.A = ln; a.lda = D;
            a.epi = VX_EPI_QKV;
            a.q = qb; a.k = kb; a.vt = vb;
            a.qkv_T = T; a.qkv_Tp = 0; a.qkv_H = NH;
            a.q_scale = q_scale;
            c.mark("gemm_qkv", 1, 2.0 * M * 3 * D * D, (double)M * D * 2 * 4 + 3.0 * D * D * 2);
            c.gemm(a);
        }
        c.mark("attention", 1, 4.0 * B * NH * (double)T * T * 64, (double)M * D * 2 * 4);
        VX(vx_attention_f16(qb, kb, vb, attb, B, NH, T, stream));
        residual_gemm(L.out, attb, D, L.lambda1, "gemm_out", 2.0 * M * D * D, (double)M * D * (2 + 8) + (double)D * D * 2);
        layernorm(c.fptr(L.ln2_w), c.fptr(L.ln2_b), ln);
        {
            vx_gemm_args a = c.base(L.fc1, M);
            a.A = ln; a.lda = D;
            a.epi = VX_EPI_F16_GELU;
            a.out = hidb; a.ldo = L.fc1.N;
            c.mark("gemm_fc1", 1, 2.0 * M * L.fc1.n_real * D, (double)M * (D + L.fc1.N) * 2 + (double)L.fc1.N * D * 2);
            c.gemm(a);
        }
        residual_gemm(L.fc2, hidb, L.fc1.N, L.lambda2, "gemm_fc2", 2.0 * M * D * L.fc2.k_real,
                      (double)M * (L.fc1.N * 2 + D * 8) + (double)L.fc2.K * D * 2);
        if (m.captures) { std::string nm = "layer_" + std::to_string(i); c.capture(nm.c_str(), x, {B, T, D, 1}, false); }
        // get_intermediate_layers (dino.cpp:100-107): the shared final LayerNorm on the tapped layers, once per tap that names this layer
        for (int f = 0; f < 4; ++f)
            if (P.feature_layers[f] == i && tap < 4) {
                std::string fb = "feat" + std::to_string(tap++);
                layernorm(c.fptr(Wt.final_ln_w), c.fptr(Wt.final_ln_b), sub(fb.c_str(), (size_t)D * 2));
                if (m.captures) { std::string nm = "dino_layer_" + std::to_string(i); c.capture(nm.c_str(), sub(fb.c_str(), (size_t)D * 2), {B, T, D, 1}, true); }
            }
    }
    if (!use_block && tap != 4) throw except("depthany: expected 4 feature layers, found %d", tap);

    if (ablate & 1) return;
    // ---- dpt::neck reassemble (depth-anything.cpp:44-64)
    const int lh[4] = {4 * ph, 2 * ph, ph, (ph + 2 - 3) / 2 + 1};
    const int lw[4] = {4 * pw, 2 * pw, pw, (pw + 2 - 3) / 2 + 1};
    const void* lay[4];
    void* cb[4] = {c.buf("c0"), c.buf("c1"), c.buf("c2"), c.buf("c3")};
    // Branch j = projection -> resize -> neck conv of tap j. (Running the four independent branches on parallel streams was measured
    // in round 1: 1-3 % slower than one stream; removed in round 4, the sub-batch split below uses the streams.)
    for (int j = 0; j < 4; ++j) {
        std::string fb = "feat" + std::to_string(j), rb = "r" + std::to_string(j);
        {
            vx_gemm_args a = c.base(Wt.re_proj[j], MP);
            a.A = c.buf(fb.c_str()); a.lda = D;
            a.a_group = Pn; a.a_group_stride = T; a.a_row_off = 1; // slice off the cls token (depth-anything.cpp:50)
            a.epi = VX_EPI_F16;
            a.out = c.buf(rb.c_str()); a.ldo = Wt.re_proj[j].N;
            a.n_valid = Wt.re_proj[j].N; // pad columns are exact zeros and feed the next GEMM's padded K
            c.mark("neck_proj", 1, 2.0 * MP * Wt.neck_c[j] * D, (double)MP * (D + Wt.re_proj[j].N) * 2);
            c.gemm(a);
        }
        if (j < 2) { // conv_transpose k == stride (4, then 2) as GEMM + pixel shuffle
            packed_gemm const& up = j == 0 ? Wt.re_up0 : Wt.re_up1;
            vx_gemm_args a = c.base(up, MP);
            a.A = c.buf(rb.c_str()); a.lda = Wt.re_proj[j].N;
            a.epi = VX_EPI_PIXSHUF;
            a.out = c.buf(j == 0 ? "l0" : "l1"); a.ldo = Wt.neck_c[j];
            a.ps_s = j == 0 ? 4 : 2; a.ps_Cout = Wt.neck_c[j]; a.ps_H = ph; a.ps_W = pw;
            c.mark("neck_convT", 1, 2.0 * MP * up.n_real * Wt.neck_c[j], (double)MP * up.n_real * 2);
            c.gemm(a);
            lay[j] = a.out;
        } else if (j == 2) {
            lay[2] = c.buf("r2");
        } else {
            c.conv(Wt.re_down3, c.buf("r3"), B, ph, pw, Wt.neck_c[3], 3, 2, 1, c.buf("l3"), Wt.neck_c[3], VX_EPI_F16, false, false, nullptr, nullptr, "neck_conv_s2");
            lay[3] = c.buf("l3");
        }
        if (m.captures) { std::string nm = "reassemble_" + std::to_string(j); c.capture(nm.c_str(), lay[j], {B, lh[j], lw[j], Wt.neck_c[j]}, true); }
        // neck.convs[j] (depth-anything.cpp:66-69): 3x3, no bias, -> F channels
        c.conv(Wt.neck_conv[j], lay[j], B, lh[j], lw[j], Wt.neck_c[j], 3, 1, 1, cb[j], F, VX_EPI_F16, false, false, nullptr, nullptr, "neck_convs");
        if (m.captures) { std::string nm = "neck_conv_" + std::to_string(j); c.capture(nm.c_str(), cb[j], {B, lh[j], lw[j], F}, true); }
    }

    // ---- fusion stage (depth-anything.cpp:25-42, 71-77)
    void *t1 = c.buf("t1"), *t2 = c.buf("t2"), *t3 = c.buf("t3"), *up = c.buf("up"), *fused = c.buf("fused");
    const void* prev = nullptr; // output of the previous fusion layer (lives in `fused`)
    const void* head_in = nullptr; // set when head.conv1 resizes the last stage's projection itself
    int head_in_h = 0, head_in_w = 0;
    for (int i = 0; i < 4; ++i) {
        fusion_weights const& FW = Wt.fusion[i];
        const int j = 3 - i; // feature consumed at this stage
        const int h = lh[j], w = lw[j];
        const void* xin;
        if (i == 0) {
            xin = cb[3];
        } else {
            // x = x0 + residual_layer1(x1) with x0 = prev, x1 = c_j:  t2 = conv2(relu(conv1(relu(x1)))) + x1 + x0
            c.conv(FW.rl1_c1, cb[j], B, h, w, F, 3, 1, 1, t1, F, VX_EPI_F16_RELU, true, false, nullptr, nullptr, "fusion_rcu");
            c.conv(FW.rl1_c2, t1, B, h, w, F, 3, 1, 1, t2, F, VX_EPI_F16_ADD, false, false, cb[j], prev, "fusion_rcu");
            xin = t2;
        }
        // residual_layer2: t3 = conv2(relu(conv1(relu(x)))) + x
        c.conv(FW.rl2_c1, xin, B, h, w, F, 3, 1, 1, t1, F, VX_EPI_F16_RELU, true, false, nullptr, nullptr, "fusion_rcu");
        c.conv(FW.rl2_c2, t1, B, h, w, F, 3, 1, 1, t3, F, VX_EPI_F16_ADD, false, false, xin, nullptr, "fusion_rcu");
        // bilinear (align_corners) to the next feature's size, or x2 for the last stage
        const int oh = i < 3 ? lh[j - 1] : 2 * h, ow = i < 3 ? lw[j - 1] : 2 * w;
        // The reference resizes and then applies the 1x1 projection (depth-anything.cpp:36-40). Both are
        // linear and the bilinear weights sum to 1, so projection (with its bias) and resize commute:
        // project at the low resolution (1/4 of the FLOPs, no full-resolution intermediate), then resize.
        {
            vx_gemm_args a = c.base(FW.proj, (long)B * h * w); // 1x1 projection (nn.cpp:76-81)
            a.A = t3; a.lda = F;
            a.epi = VX_EPI_F16;
            a.out = t1; a.ldo = F;
            c.mark("fusion_proj", 1, 2.0 * B * h * w * F * F, (double)B * h * w * F * 4);
            c.gemm(a);
        }
        // The last stage's resize (148^2 -> 296^2 at 518^2: 0.45 GB written and read back per 32 images) feeds head.conv1 only: that
        // conv interpolates it in its halo loader instead (kernels_dconv.hip BIL). Captures keep the unfused form (fusion_3 is one).
        if (i == 3 && !m.captures && exec_ctx::bil_ok(Wt.head1, oh, ow, F, VX_EPI_F16, h, w)) {
            head_in = t1; head_in_h = h; head_in_w = w;
            break;
        }
        c.mark("bilinear", 1, 0, (double)B * (h * w + oh * ow) * F * 2);
        if (!(ablate & 2)) VX(vx_bilinear_ac_f16(t1, fused, B, h, w, F, oh, ow, stream));
        (void)up;
        prev = fused;
        if (m.captures) { std::string nm = "fusion_" + std::to_string(i); c.capture(nm.c_str(), fused, {B, oh, ow, F}, true); }
    }

    // ---- dpt::head (depth-anything.cpp:81-96)
    const int fh = 8 * ph, fw = 8 * pw;
    if (head_in) c.conv(Wt.head1, head_in, B, fh, fw, F, 3, 1, 1, c.buf("h1"), HC, VX_EPI_F16, false, false, nullptr, nullptr, "head_conv1", head_in_h, head_in_w);
    else c.conv(Wt.head1, fused, B, fh, fw, F, 3, 1, 1, c.buf("h1"), HC, VX_EPI_F16, false, false, nullptr, nullptr, "head_conv1");
    c.capture("head_conv1", c.buf("h1"), {B, fh, fw, HC}, true);
    // head: interpolate to the image extent, then conv2 (depth-anything.cpp:84-87): conv2's loader resizes h1 itself where it can
    // (0.55 GB written and read back per 32 images otherwise)
    // ... or, where the shape is the north star's (32 -> 32 channels, scale <= 0.6), the kernel made for this tail: resize + conv2 + ReLU +
    // conv3 + ReLU with the 3x3 kernel in registers (kernels_headconv.hip)
    static const bool no_headconv = getenv("VISP_NO_HEADCONV") != nullptr;
    const bool head2_hc = !m.captures && !no_headconv && Wt.head2_frag != SIZE_MAX && Wt.head2.b != SIZE_MAX && vx_headconv_supported(HC, 32, H, W, fh, fw) &&
                         (size_t)B * fh * fw * HC * 2 < ((size_t)1 << 31); // (the kernel addresses its source through one 32-bit buffer descriptor)
    const bool head2_bil = !head2_hc && !m.captures && Wt.head2.N == 32 && exec_ctx::bil_ok(Wt.head2, H, W, HC, VX_EPI_HEAD_OUT, fh, fw);
    if (!head2_bil && !head2_hc) {
        c.mark("bilinear", 1, 0, (double)B * ((double)fh * fw + (double)H * W) * HC * 2);
        if (!(ablate & 2)) VX(vx_bilinear_ac_f16(c.buf("h1"), c.buf("hup"), B, fh, fw, HC, H, W, stream));
    }
    float* depth = raw_out_dev ? static_cast<float*>(raw_out_dev) + (size_t)b0 * H * W : static_cast<float*>(c.buf("depth"));
    if (head2_hc) {
        c.mark("head_conv2+3", 1, 2.0 * B * H * W * 32 * (Wt.head2.k_real + 1), (double)B * ((double)fh * fw * HC * 2 + (double)H * W * 4));
        VX(vx_headconv_bil_f16(c.buf("h1"), c.wptr(Wt.head2_frag), reinterpret_cast<const float*>(c.wa + Wt.head2.b), c.fptr(Wt.head3_w), Wt.head3_b, P.max_depth, depth, B,
                               H, W, fh, fw, stream));
    } else if (Wt.head2.N == 32) {
        // conv2 (3x3 -> 32) + ReLU + conv3 (1x1 -> 1) + ReLU [* max_depth] in one kernel: the 32-channel
        // full-resolution tensor never reaches HBM
        vx_gemm_args a = c.base(Wt.head2, (long)B * H * W);
        a.A = c.buf("hup");
        a.conv_kh = a.conv_kw = 3; a.conv_stride = 1; a.conv_pad = 1;
        a.conv_H = H; a.conv_W = W; a.conv_Cin = HC; a.conv_OH = H; a.conv_OW = W;
        a.epi = VX_EPI_HEAD_OUT;
        a.out = depth;
        a.lambda = c.fptr(Wt.head3_w);
        a.head_bias = Wt.head3_b;
        a.head_scale = P.max_depth;
        c.mark("head_conv2+3", 1, 2.0 * B * H * W * 32 * (Wt.head2.k_real + 1), (double)B * H * W * (HC * 2 + 4));
        if (exec_ctx::dconv_ok(Wt.head2, 3, 1, 1, W, HC, VX_EPI_HEAD_OUT)) {
            vx_dconv_args d = c.dconv_base(Wt.head2, head2_bil ? c.buf("h1") : c.buf("hup"), B, H, W, HC);
            if (head2_bil) { d.bil_hs = fh; d.bil_ws = fw; }
            d.epi = VX_DC_HEAD_F32;
            d.head_w = c.fptr(Wt.head3_w); d.head_bias = Wt.head3_b; d.head_scale = P.max_depth;
            d.out = depth;
            VX(vx_dconv3x3_f16(&d, stream));
        } else {
            c.gemm(a);
        }
    } else {
        c.conv(Wt.head2, c.buf("hup"), B, H, W, HC, 3, 1, 1, c.buf("h2"), Wt.head2.N, VX_EPI_F16_RELU, false, false, nullptr, nullptr, "head_conv2");
        c.mark("head_out", 1, 2.0 * B * H * W * Wt.head2.N, (double)B * H * W * (Wt.head2.N * 2 + 4));
        VX(vx_head_out_f32(c.buf("h2"), c.fptr(Wt.head3_w), Wt.head3_b, P.max_depth, depth, (int64_t)B * H * W, Wt.head2.N, stream));
    }
    c.capture("depth", depth, {B, H, W, 1}, false);

    // ---- depthany_process_output (depth-anything.cpp:142-149): per-image min-max to [0,1]
    c.mark("normalize", 3, 0, (double)B * H * W * 12);
    VX(vx_minmax_normalize(depth, static_cast<float*>(out_dev) + (size_t)b0 * H * W, static_cast<float*>(c.buf("minmax")), B, (int64_t)H * W, stream));
    }; // run_sub
    {
        // measured at batch 32 (profiles/r02_split_streams.txt): 7.70 / 7.24 / 7.21 / 7.33 ms per step for 1 / 2 / 3 / 4 sub-batches (GEMM schedule)
        static const int split_env = getenv("VISP_SPLIT") ? atoi(getenv("VISP_SPLIT")) : 0;
        const int want = m.split > 0 ? m.split : (split_env > 0 ? split_env : (B >= 24 ? 3 : (B >= 8 ? 2 : 1)));
        const int n_split = ((!m.timing || m.timing_split) && !m.captures && want > 1 && want <= 4 && B >= 2 * want) ? want : 1;
        if (n_split == 1) {
            run_sub(0, B, stream);
        } else {
            VX(vx_event_record(m.fork_event, stream));
            for (int j = 0; j < n_split; ++j) {
                const int b0 = (int)((long)B * j / n_split), b1 = (int)((long)B * (j + 1) / n_split);
                void* strm = j == 0 ? stream : m.aux_stream[j - 1];
                if (j > 0) VX(vx_stream_wait_event(strm, m.fork_event));
                run_sub(b0, b1 - b0, strm);
                if (m.timing) c.mark("__end", 0, 0, 0); // (c.stream is this sub-batch's stream)
                if (j > 0) {
                    VX(vx_event_record(m.join_event[j - 1], strm));
                    VX(vx_stream_wait_event(stream, m.join_event[j - 1]));
                }
            }
        }
        c.stream = stream;
        c.sub_b0 = 0;
    }
    c.finish_timing();
}

void depthany_compute_batch_device(depthany_model& m, void const* rgb_dev, int batch, int w, int h, void* out_dev,
                                   void* raw_out_dev, void* stream) {
    if (!m.weights_uploaded) throw except("depthany: weights were not uploaded (load_no_upload) and no arena broadcast was marked complete");
    device_turn turn(*m.backend);
    depthany_reserve(m, batch, w, h);
    bool own_stream = stream == nullptr;
    void* s = own_stream ? m.backend->stream : stream;
    if (m.use_graph && !m.captures && !m.timing) {
        // the captured launch sequence bakes pointers in: stage through workspace-owned buffers
        size_t in_bytes = (size_t)batch * h * w * 3, out_bytes = (size_t)batch * h * w * 4;
        VX(vx_memcpy_d2d(m.ws.buf["rgb"], rgb_dev, in_bytes, s));
        if (!m.ws.graph_exec) {
            VX(vx_graph_begin_capture(s));
            try {
                run_forward(m, m.ws.buf["rgb"], m.ws.buf["out"], m.ws.buf["depth"], s);
            } catch (...) {
                void* g = nullptr;
                vx_graph_end_capture(s, &g);
                if (g) vx_graph_destroy(g);
                throw;
            }
            VX(vx_graph_end_capture(s, &m.ws.graph_exec));
        }
        VX(vx_graph_launch(m.ws.graph_exec, s));
        VX(vx_memcpy_d2d(out_dev, m.ws.buf["out"], out_bytes, s));
        if (raw_out_dev) VX(vx_memcpy_d2d(raw_out_dev, m.ws.buf["depth"], out_bytes, s));
    } else {
        run_forward(m, rgb_dev, out_dev, raw_out_dev, s);
    }
    if (own_stream) VX(vx_stream_sync(s));
}

void depthany_compute_batch_host(depthany_model& m, uint8_t const* rgb, int batch, int w, int h, float* out, float* raw_out) {
    device_turn turn(*m.backend);
    depthany_reserve(m, batch, w, h);
    void* s = m.backend->stream;
    size_t in_bytes = (size_t)batch * h * w * 3, out_bytes = (size_t)batch * h * w * 4;
    VX(vx_memcpy_h2d(m.ws.buf["rgb"], rgb, in_bytes, s));
    bool g = m.use_graph;
    m.use_graph = false; // direct launches already use workspace buffers
    try {
        depthany_compute_batch_device(m, m.ws.buf["rgb"], batch, w, h, m.ws.buf["out"], m.ws.buf["depth"], s);
    } catch (...) {
        m.use_graph = g;
        throw;
    }
    m.use_graph = g;
    VX(vx_memcpy_d2h(out, m.ws.buf["out"], out_bytes, s));
    if (raw_out) VX(vx_memcpy_d2h(raw_out, m.ws.buf["depth"], out_bytes, s));
}

//
// overlapped host pipeline

depthany_pipeline* depthany_pipeline_create(depthany_model& m, int batch, int w, int h, int n_slots) {
    if (n_slots < 2 || n_slots > 8) throw except("depthany pipeline: %d slots (2..8)", n_slots);
    device_turn turn(*m.backend);
    depthany_reserve(m, batch, w, h);
    auto p = std::make_unique<depthany_pipeline>();
    p->model = &m; p->batch = batch; p->w = w; p->h = h; p->n_slots = n_slots;
    p->in_bytes = (size_t)batch * h * w * 3;
    p->out_bytes = (size_t)batch * h * w * 4;
    VX(vx_stream_create(&p->h2d_stream));
    VX(vx_stream_create(&p->d2h_stream));
    const char* ne = getenv("VISP_PIPELINE_EXECUTORS");
    const int n_exec = ne ? std::max(1, std::min(2, atoi(ne))) : 1; // two forwards in flight measured SLOWER (7.0-7.2 vs 6.85 ms per batch of 32)
    p->exec.push_back(&m);
    p->compute_stream.push_back(m.backend->stream);
    for (int e = 1; e < n_exec; ++e) {
        p->exec.push_back(depthany_clone_executor(m));
        void* cs = nullptr;
        VX(vx_stream_create(&cs));
        p->compute_stream.push_back(cs);
        depthany_reserve(*p->exec.back(), batch, w, h);
    }
    p->slots.resize((size_t)n_slots);
    for (auto& s : p->slots) {
        VX(vx_malloc_host(&s.pin_in, p->in_bytes));
        VX(vx_malloc_host(&s.pin_out, p->out_bytes));
        VX(vx_malloc(&s.dev_in, p->in_bytes));
        VX(vx_malloc(&s.dev_out, p->out_bytes));
        VX(vx_event_create(&s.uploaded));
        VX(vx_event_create(&s.computed));
        VX(vx_event_create(&s.downloaded));
    }
    return p.release();
}

depthany_pipeline::~depthany_pipeline() {
    if (model) vx_set_device(model->backend->index);
    for (auto& s : slots) {
        if (s.busy) vx_event_sync(s.downloaded);
        vx_free_host(s.pin_in); vx_free_host(s.pin_out);
        vx_free(s.dev_in); vx_free(s.dev_out);
        vx_event_destroy(s.uploaded); vx_event_destroy(s.computed); vx_event_destroy(s.downloaded);
    }
    if (h2d_stream) vx_stream_destroy(h2d_stream);
    if (d2h_stream) vx_stream_destroy(d2h_stream);
    for (size_t e = 1; e < exec.size(); ++e) {
        vx_stream_sync(compute_stream[e]);
        delete exec[e];
        vx_stream_destroy(compute_stream[e]);
    }
}

uint8_t* depthany_pipeline_input(depthany_pipeline& p) {
    auto& s = p.slots[(size_t)p.next];
    if (s.busy) throw except("depthany pipeline: slot %d still holds an unread result (wait for its ticket first)", p.next);
    return static_cast<uint8_t*>(s.pin_in);
}

int depthany_pipeline_submit(depthany_pipeline& p, uint8_t const* rgb) {
    depthany_model& m = *p.model;
    device_turn turn(*m.backend);
    const int ticket = p.next;
    auto& s = p.slots[(size_t)ticket];
    if (s.busy) throw except("depthany pipeline: slot %d still holds an unread result (wait for its ticket first)", ticket);
    const size_t e = (size_t)(p.n_submitted++ % (long)p.exec.size()); // consecutive batches alternate executors
    depthany_model& em = *p.exec[e];
    em.use_graph = m.use_graph;
    if (em.schedule != m.schedule) { // visp_depthany_set_schedule on the model after the pipeline was made: the executors follow it,
        em.schedule = m.schedule;    // and a launch sequence captured for the other schedule is dropped
        if (em.ws.graph_exec) {
            vx_graph_destroy(em.ws.graph_exec);
            em.ws.graph_exec = nullptr;
        }
    }
    if (em.ws.B != p.batch || em.ws.W != p.w || em.ws.H != p.h) depthany_reserve(em, p.batch, p.w, p.h);
    if (rgb && rgb != s.pin_in) memcpy(s.pin_in, rgb, p.in_bytes);
    void* cs = p.compute_stream[e];
    VX(vx_memcpy_h2d_async(s.dev_in, s.pin_in, p.in_bytes, p.h2d_stream));
    VX(vx_event_record(s.uploaded, p.h2d_stream));
    VX(vx_stream_wait_event(cs, s.uploaded));
    depthany_compute_batch_device(em, s.dev_in, p.batch, p.w, p.h, s.dev_out, nullptr, cs);
    VX(vx_event_record(s.computed, cs));
    VX(vx_stream_wait_event(p.d2h_stream, s.computed));
    VX(vx_memcpy_d2h_async(s.pin_out, s.dev_out, p.out_bytes, p.d2h_stream));
    VX(vx_event_record(s.downloaded, p.d2h_stream));
    s.busy = true;
    p.next = (p.next + 1) % p.n_slots;
    return ticket;
}

float const* depthany_pipeline_wait(depthany_pipeline& p, int ticket) {
    if (ticket < 0 || ticket >= p.n_slots) throw except("depthany pipeline: bad ticket %d", ticket);
    auto& s = p.slots[(size_t)ticket];
    if (!s.busy) throw except("depthany pipeline: ticket %d has nothing in flight", ticket);
    device_turn turn(*p.model->backend);
    VX(vx_event_sync(s.downloaded));
    s.busy = false;
    return static_cast<float const*>(s.pin_out);
}

// One shard of visp_depthany_compute_sharded on this model: pageable host images in, pageable results out, through the overlapped
// pipeline in chunks of the tuned step (32 images; pinned staging, H2D / forward / D2H of consecutive chunks on three streams,
// hipGraph replay) instead of one blocking pageable round trip. A last partial chunk runs as a full one (the pinned slot's other
// images are whatever it held: images are independent) -- no second workspace shape, no graph re-capture. Bit-identical to
// depthany_compute_batch_host.
void depthany_compute_shard_host(depthany_model& m, uint8_t const* rgb, int count, int w, int h, float* out) {
    constexpr int chunk = 32;
    if (count < chunk / 2) { // small shards: nothing to overlap, and a 32-image step for a few images wastes the device
        depthany_compute_batch_host(m, rgb, count, w, h, out, nullptr);
        return;
    }
    if (!m.shard_pipeline || m.shard_pipeline->w != w || m.shard_pipeline->h != h) {
        delete m.shard_pipeline;
        m.shard_pipeline = nullptr;
        m.shard_pipeline = depthany_pipeline_create(m, chunk, w, h, 3);
    }
    depthany_pipeline& p = *m.shard_pipeline;
    const size_t px = (size_t)w * h;
    const int n_chunks = (count + chunk - 1) / chunk;
    const bool graph = m.use_graph;
    m.use_graph = true;
    std::vector<std::pair<int, int>> flight; // (ticket, chunk index)
    auto retire = [&]() {
        const auto [ticket, k] = flight.front();
        flight.erase(flight.begin());
        float const* res = depthany_pipeline_wait(p, ticket);
        const int n = std::min(chunk, count - k * chunk);
        memcpy(out + (size_t)k * chunk * px, res, (size_t)n * px * 4);
    };
    try {
        for (int k = 0; k < n_chunks; ++k) {
            if ((int)flight.size() == p.n_slots) retire();
            const int n = std::min(chunk, count - k * chunk);
            uint8_t* pin = depthany_pipeline_input(p);
            memcpy(pin, rgb + (size_t)k * chunk * px * 3, (size_t)n * px * 3);
            flight.push_back({depthany_pipeline_submit(p, nullptr), k});
        }
        while (!flight.empty()) retire();
    } catch (...) {
        m.use_graph = graph;
        for (auto& f : flight) { // leave no slot busy behind an error
            try { depthany_pipeline_wait(p, f.first); } catch (...) {}
        }
        throw;
    }
    m.use_graph = graph;
}

// reference src/visp/vision.cpp:147-167
image_data depthany_compute(depthany_model& m, image_view image) {
    if (is_float(image.format) || n_channels(image.format) < 3)
        throw except("depthany: unsupported input image format [%d], expected an 8-bit colour image", int(image.format));
    i32x2 res = depthany_image_extent(image.extent, m.params);
    m.params.image_extent = res;
    // depthany_process_input: image_scale to the model extent happens in the caller's format -- for rgba / bgra / argb stb resizes alpha-weighted (premultiplied),
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
    image_data out = image_alloc(res, image_format::alpha_f32);
    depthany_compute_batch_host(m, static_cast<const uint8_t*>(rgb_view.data), 1, res[0], res[1],
                                reinterpret_cast<float*>(out.data.get()), nullptr);
    if (res != caller_extent) return image_scale(view_of(out), caller_extent); // depthany_process_output
    return out;
}

} // namespace visp
