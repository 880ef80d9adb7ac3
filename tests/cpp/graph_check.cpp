// The C++ face of the graph layer (include/visp/ml.h + nn.h) on a small network of this test's own making -- a token mixer followed by a
// little convolutional decoder -- whose every stage is also evaluated by plain float loops in this file:
//   tokens [128, 70, 2] -> x + lambda * fc2(gelu(fc1(layer_norm(x)))) -> reshape to a [128, 10, 7, 2] map -> conv 3x3 128->64 + relu
//   -> bilinear (align corners) to 20 x 14 -> conv 1x1 64->32 -> output
// Weights come from model_init + model_add_tensor, the input through transfer_to_backend, the result through transfer_from_backend.
// Built by __graft_entry__.build(), run by tests/test_gpu_graph.py on the GPU box. Exit code 0 = within the f16 tolerance.
#include <cmath>
#include <cstdio>
#include <vector>

#include "visp/nn.h"

using namespace visp;

namespace {

constexpr int C = 128, HID = 256, W = 10, H = 7, T = W * H, B = 2, C2 = 64, C3 = 32, W2 = 20, H2 = 14;

struct rng { // xorshift: deterministic inputs without <random>'s implementation-defined distributions
    uint64_t s;
    float next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return float(int64_t(s >> 40) - (1 << 23)) / float(1 << 23); }
};
std::vector<float> fill(rng& r, size_t n, float scale, float bias = 0.f) {
    std::vector<float> v(n);
    for (float& x : v) x = bias + scale * r.next();
    return v;
}
float f16_round(float v) { // what an f16 activation or matrix weight holds: to binary16 (round to nearest even) and back
    const uint16_t hb = detail::f32_to_f16(v);
    const uint32_t sign = uint32_t(hb & 0x8000u) << 16, e = (hb >> 10) & 31u, man = hb & 1023u;
    float mag = e == 0 ? std::ldexp(float(man), -24) : e == 31 ? (man ? NAN : INFINITY) : std::ldexp(float(man | 1024u), int(e) - 25);
    return sign ? -mag : mag;
}
void round_all(std::vector<float>& v) { for (float& x : v) x = f16_round(x); }
float gelu_tanh(float x) { return 0.5f * x * (1.f + std::tanh(0.79788456f * (x + 0.044715f * x * x * x))); }

} // namespace

int main() {
    try {
        rng r{0x9e3779b97f4a7c15ull};
        std::vector<float> x = fill(r, size_t(B) * T * C, 1.f), ln_w = fill(r, C, 0.1f, 1.f), ln_b = fill(r, C, 0.1f), w1 = fill(r, size_t(HID) * C, 0.08f),
                           b1 = fill(r, HID, 0.05f), w2 = fill(r, size_t(C) * HID, 0.06f), b2 = fill(r, C, 0.05f), lam = fill(r, C, 0.1f, 0.4f),
                           k3 = fill(r, size_t(C2) * 9 * C, 0.03f), kb = fill(r, C2, 0.05f), k1 = fill(r, size_t(C3) * C2, 0.1f);
        for (auto* v : {&x, &w1, &w2, &k3, &k1}) round_all(*v);

        backend_device dev = backend_init();
        model_weights weights = model_init();
        model_add_tensor(weights, "mix.norm.weight", GGML_TYPE_F32, {C, 1, 1, 1}, ln_w);
        model_add_tensor(weights, "mix.norm.bias", GGML_TYPE_F32, {C, 1, 1, 1}, ln_b);
        model_add_tensor(weights, "mix.fc1.weight", GGML_TYPE_F16, {C, HID, 1, 1}, w1);
        model_add_tensor(weights, "mix.fc1.bias", GGML_TYPE_F32, {HID, 1, 1, 1}, b1);
        model_add_tensor(weights, "mix.fc2.weight", GGML_TYPE_F16, {HID, C, 1, 1}, w2);
        model_add_tensor(weights, "mix.fc2.bias", GGML_TYPE_F32, {C, 1, 1, 1}, b2);
        model_add_tensor(weights, "mix.gate", GGML_TYPE_F32, {C, 1, 1, 1}, lam);
        model_add_tensor(weights, "dec.conv.weight", GGML_TYPE_F16, {C, 3, 3, C2}, k3);   // [Cin, kw, kh, Cout]
        model_add_tensor(weights, "dec.conv.bias", GGML_TYPE_F32, {C2, 1, 1, 1}, kb);
        model_add_tensor(weights, "dec.out.weight", GGML_TYPE_F16, {C2, 1, 1, C3}, k1);

        compute_graph graph = compute_graph_init(weights);
        model_ref m(graph);
        tensor in = compute_graph_input(m, GGML_TYPE_F16, {C, T, B, 1}, "tokens");
        model_ref mix = m["mix"];
        tensor hid = gelu(m, linear(mix["fc1"], layer_norm(mix["norm"], in, 1e-6f)));
        tensor mixed = add(m, in, mul(m, linear(mix["fc2"], hid), mix.weights("gate")));
        compute_graph_output(m, mixed, "mixed");
        tensor map = reshape_4d(m, mixed, C, W, H, B);
        tensor feat = relu(m, conv_2d(m["dec"]["conv"], map, 1, 1));
        tensor up = interpolate(m, feat, {W2, H2}, GGML_SCALE_MODE_BILINEAR | GGML_SCALE_FLAG_ALIGN_CORNERS);
        tensor out = compute_graph_output(m, conv_2d(m["dec"]["out"], up), "decoded");
        if (out->ne[0] != C3 || out->ne[1] != W2 || out->ne[2] != H2 || out->ne[3] != B) throw exception("unexpected output shape");
        compute_graph_allocate(graph, dev);
        transfer_to_backend(in, std::span<float const>(x));
        compute(graph, dev);
        tensor_data got_mixed = transfer_from_backend(mixed), got = transfer_from_backend(out);

        // ---- the same network in float loops (activations rounded to f16 where the device stores them)
        std::vector<float> ref_mixed(x.size()), n(C), h(HID);
        for (int row = 0; row < B * T; ++row) {
            float const* xr = &x[size_t(row) * C];
            float mean = 0, var = 0;
            for (int c = 0; c < C; ++c) mean += xr[c];
            mean /= C;
            for (int c = 0; c < C; ++c) var += (xr[c] - mean) * (xr[c] - mean);
            const float rstd = 1.f / std::sqrt(var / C + 1e-6f);
            for (int c = 0; c < C; ++c) n[size_t(c)] = f16_round((xr[c] - mean) * rstd * ln_w[size_t(c)] + ln_b[size_t(c)]);
            for (int j = 0; j < HID; ++j) {
                float a = b1[size_t(j)];
                for (int c = 0; c < C; ++c) a += w1[size_t(j) * C + c] * n[size_t(c)];
                h[size_t(j)] = f16_round(gelu_tanh(a));
            }
            for (int c = 0; c < C; ++c) {
                float a = b2[size_t(c)];
                for (int j = 0; j < HID; ++j) a += w2[size_t(c) * HID + j] * h[size_t(j)];
                ref_mixed[size_t(row) * C + c] = f16_round(xr[c] + lam[size_t(c)] * a);
            }
        }
        std::vector<float> feat_ref(size_t(B) * H * W * C2), up_ref(size_t(B) * H2 * W2 * C2), ref(size_t(B) * H2 * W2 * C3);
        auto at = [&](int b, int y, int xx, int c) { return ref_mixed[((size_t(b) * H + y) * W + xx) * C + c]; }; // token row = y * W + x
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx)
                    for (int o = 0; o < C2; ++o) {
                        float a = kb[size_t(o)];
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int sy = y + ky - 1, sx = xx + kx - 1;
                                if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
                                for (int c = 0; c < C; ++c) a += k3[((size_t(o) * 3 + ky) * 3 + kx) * C + c] * at(b, sy, sx, c);
                            }
                        feat_ref[((size_t(b) * H + y) * W + xx) * C2 + o] = f16_round(std::max(a, 0.f));
                    }
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H2; ++y)
                for (int xx = 0; xx < W2; ++xx) {
                    const float fy = float(y) * float(H - 1) / float(H2 - 1), fx = float(xx) * float(W - 1) / float(W2 - 1);
                    const int y0 = std::min(int(fy), H - 1), x0 = std::min(int(fx), W - 1), y1 = std::min(y0 + 1, H - 1), x1 = std::min(x0 + 1, W - 1);
                    const float ty = fy - float(y0), tx = fx - float(x0);
                    for (int c = 0; c < C2; ++c) {
                        auto f = [&](int yy, int xc) { return feat_ref[((size_t(b) * H + yy) * W + xc) * C2 + c]; };
                        up_ref[((size_t(b) * H2 + y) * W2 + xx) * C2 + c] =
                            f16_round((1 - ty) * ((1 - tx) * f(y0, x0) + tx * f(y0, x1)) + ty * ((1 - tx) * f(y1, x0) + tx * f(y1, x1)));
                    }
                }
        for (size_t p = 0; p < size_t(B) * H2 * W2; ++p)
            for (int o = 0; o < C3; ++o) {
                float a = 0;
                for (int c = 0; c < C2; ++c) a += k1[size_t(o) * C2 + c] * up_ref[p * C2 + c];
                ref[p * C3 + o] = a;
            }

        auto worst = [](std::span<float const> a, std::vector<float> const& b) {
            double err = 0, top = 0;
            for (size_t i = 0; i < b.size(); ++i) { err = std::max(err, std::fabs(double(a[i]) - double(b[i]))); top = std::max(top, std::fabs(double(b[i]))); }
            return err / top;
        };
        if (got_mixed.as_f32().size() != ref_mixed.size() || got.as_f32().size() != ref.size()) throw exception("result sizes differ from the host evaluation");
        const double e1 = worst(got_mixed.as_f32(), ref_mixed), e2 = worst(got.as_f32(), ref);
        std::string text = compute_graph_describe(graph);
        std::printf("graph_check: mixer max rel err %.2e, decoder %.2e; %s", e1, e2, text.substr(text.rfind("launches=")).c_str());
        if (!(e1 < 4e-3 && e2 < 1e-2)) return 1;
        std::printf("graph_check ok\n");
        return 0;
    } catch (std::exception const& e) {
        std::fprintf(stderr, "graph_check: %s\n", e.what());
        return 1;
    }
}
