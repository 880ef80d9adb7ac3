// GGUF tensors -> packed operands in a host-side arena image (the weight arena that is uploaded once): shared by the
// model loaders that use the GEMM family on f16 activations (tinyvit.cpp, swin.cpp). Internal header: everything lives in an
// anonymous namespace of the including translation unit.
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/visp_hip_kernels.h"
#include "tinyvit.h" // packed_gemm / packed_vec (depthany.h), packed_dw
#include "visp_util.h"

namespace visp {

#define VX(call)                                        \
    do {                                                \
        if (!(call)) throw except("%s", vx_last_error()); \
    } while (0)

namespace {

template <typename T>
T round_up(T x, T m) { return (x + m - 1) / m * m; }

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

// GGUF -> packed operands. Layout rules of the mobile-sam file (scripts/convert.py:204-247): BatchNorm fused into
// "<conv>.c.weight/.c.bias"; fused kernels are torch OIHW and listed in conv2d_weights when the file is whcn (otherwise
// already OHWI); local_conv kernels are always stored H W 1 C and never listed.
struct packer {
    model_file const& file;
    arena_builder& ab;
    bool with_data;
    bool file_whcn;
    std::vector<int32_t> conv2d;

    bool listed(std::string const& name) const {
        auto it = file.index.find(name);
        return it != file.index.end() && std::binary_search(conv2d.begin(), conv2d.end(), it->second);
    }
    gguf_tensor const& get(std::string const& name) const {
        gguf_tensor const& t = file.tensor(name);
        if (t.type != GGML_F32 && t.type != GGML_F16) throw except("tensor %s: unsupported type %d", name.c_str(), t.type);
        if (with_data && !t.data) throw except("tensor %s has no data", name.c_str());
        return t;
    }

    packed_vec vec(std::string const& name) {
        gguf_tensor const& t = get(name);
        packed_vec v;
        v.n = (int)t.n_elements();
        v.off = ab.alloc((size_t)v.n * 4);
        if (with_data) {
            float* d = reinterpret_cast<float*>(ab.data.data() + v.off);
            for (int i = 0; i < v.n; ++i) d[i] = tensor_at(t, i);
        }
        return v;
    }

    packed_vec f16_vec(std::string const& name, int64_t expect = -1) { // arena f16 copy (n = element count)
        gguf_tensor const& t = get(name);
        if (expect >= 0 && t.n_elements() != expect) throw except("tensor %s: %lld elements, expected %lld", name.c_str(), (long long)t.n_elements(), (long long)expect);
        packed_vec v;
        v.n = (int)t.n_elements();
        v.off = ab.alloc((size_t)v.n * 2 + 64);
        if (with_data) {
            uint16_t* d = reinterpret_cast<uint16_t*>(ab.data.data() + v.off);
            for (int i = 0; i < v.n; ++i) d[i] = f32_to_f16(tensor_at(t, i));
        }
        return v;
    }
    std::vector<float> host_vec(std::string const& name, int64_t expect) {
        gguf_tensor const& t = get(name);
        if (t.n_elements() != expect) throw except("tensor %s: %lld elements, expected %lld", name.c_str(), (long long)t.n_elements(), (long long)expect);
        std::vector<float> v(with_data ? (size_t)expect : 0);
        for (size_t i = 0; i < v.size(); ++i) v[i] = tensor_at(t, i);
        return v;
    }
    // conv_transpose_2d with kernel == stride (nn.cpp:117-129; weight ne [kw, kh, Cout, Cin] = torch [Cin][Cout][kh][kw], never
    // permuted by the converter) as a GEMM whose row n = (dy * s + dx) * Cout + co feeds the pixel-shuffle epilogue
    packed_gemm conv_transpose(std::string const& prefix, int stride, int* cout_out) {
        gguf_tensor const& w = get(prefix + ".weight");
        const int kw = (int)w.ne[0], kh = (int)w.ne[1], cout = (int)w.ne[2], cin = (int)w.ne[3];
        if (kw != stride || kh != stride) throw except("%s: conv_transpose kernel %dx%d with stride %d is not supported", prefix.c_str(), kw, kh, stride);
        *cout_out = cout;
        auto at = [&](int n, int k) {
            const int tap = n / cout, co = n % cout, dy = tap / stride, dx = tap % stride;
            return tensor_at(w, (((size_t)k * cout + co) * kh + dy) * kw + dx);
        };
        return matrix(stride * stride * cout, cin, at, file.find(prefix + ".bias"), cout);
    }

    // attention_biases_indexed [heads][N][N] -> the accumulator-order f16 image of kernels_winattn.hip
    packed_vec attention_bias(std::string const& name, int N, int heads) {
        gguf_tensor const& t = get(name);
        if (t.n_elements() != (int64_t)heads * N * N)
            throw except("mobile-sam: %s has %lld elements, expected %d", name.c_str(), (long long)t.n_elements(), heads * N * N);
        packed_vec v;
        v.n = (int)(vx_window_attention_bias_bytes(N, heads) / 2);
        v.off = ab.alloc((size_t)v.n * 2);
        if (with_data) {
            std::vector<float> f((size_t)heads * N * N);
            for (size_t i = 0; i < f.size(); ++i) f[i] = tensor_at(t, i);
            VX(vx_window_attention_pack_bias(f.data(), N, heads, ab.data.data() + v.off));
        }
        return v;
    }

    // rows [n][k] -> f16 [N pad 32][K pad 64] + f32 bias [N]
    packed_gemm matrix(int n, int k, std::function<float(int, int)> at, gguf_tensor const* bias, int bias_period = 0) {
        packed_gemm g;
        g.n_real = n; g.k_real = k;
        // N decides the GEMM's block tile (128 wide if N % 128 == 0, else 64, else 32): wider tiles re-read A fewer times, so pad
        // to 64, and to 128 when that costs at most 10% more columns (480 -> 512, 960 -> 1024; measured +2% on the encoder)
        g.N = round_up(n, n > 64 ? 64 : 32);
        if (round_up(n, 128) * 10 <= n * 11) g.N = round_up(n, 128);
        g.K = round_up(k, 64);
        g.w = ab.alloc((size_t)g.N * g.K * 2);
        if (with_data) {
            uint16_t* w = reinterpret_cast<uint16_t*>(ab.data.data() + g.w);
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < k; ++c) w[(size_t)r * g.K + c] = f32_to_f16(at(r, c));
        }
        if (bias) {
            const int period = bias_period ? bias_period : n; // conv_transpose: the Cout biases repeat for every (dy, dx)
            if ((int)bias->n_elements() != period) throw except("tensor %s: %d elements, expected %d", bias->name.c_str(), (int)bias->n_elements(), period);
            g.b = ab.alloc((size_t)g.N * 4);
            if (with_data) {
                float* b = reinterpret_cast<float*>(ab.data.data() + g.b);
                for (int r = 0; r < n; ++r) b[r] = tensor_at(*bias, r % period);
            }
        }
        return g;
    }

    packed_gemm linear(std::string const& prefix) { // weight ne [K, N] == torch [N][K]
        gguf_tensor const& w = get(prefix + ".weight");
        const int K = (int)w.ne[0], N = (int)w.ne[1];
        return matrix(N, K, [&](int n, int k) { return tensor_at(w, (size_t)n * K + k); }, file.find(prefix + ".bias"));
    }

    // dense conv as GEMM rows k = (ky, kx, c); dup_in: the 3 input channels are read as value + residue (channels 3..5
    // repeat 0..2, the rest of the 8-channel pixel is zero)
    packed_gemm conv(std::string const& prefix, int* ksize = nullptr, int* cin_out = nullptr, bool dup_in = false) {
        std::string name = prefix + ".weight";
        gguf_tensor const& w = get(name);
        const bool oihw = file_whcn && listed(name);
        int kw, kh, cin, cout = (int)w.ne[3];
        if (oihw) { kw = (int)w.ne[0]; kh = (int)w.ne[1]; cin = (int)w.ne[2]; }
        else { cin = (int)w.ne[0]; kw = (int)w.ne[1]; kh = (int)w.ne[2]; }
        if (kw != kh) throw except("tensor %s: non-square kernel", name.c_str());
        if (ksize) *ksize = kw;
        if (cin_out) *cin_out = cin;
        const int cpix = dup_in ? 8 : cin;
        if (dup_in && cin != 3) throw except("tensor %s: expected 3 input channels", name.c_str());
        auto at = [&](int n, int k) -> float {
            const int tap = k / cpix, c = k % cpix;
            int cs = c;
            if (dup_in) {
                if (c >= 6) return 0.0f;
                cs = c % 3;
            }
            const int ky = tap / kw, kx = tap % kw;
            const size_t src = oihw ? (((size_t)n * cin + cs) * kh + ky) * kw + kx : (((size_t)n * kh + ky) * kw + kx) * cin + cs;
            return tensor_at(w, src);
        };
        return matrix(cout, kh * kw * cpix, at, file.find(prefix + ".bias"));
    }

    // depthwise 3x3 -> f16 [9][C] (tap-major) + f32 bias; OIHW-listed: ne = [kw, kh, 1, C]; otherwise [C, 1, kw, kh]
    packed_dw depthwise(std::string const& prefix) {
        std::string name = prefix + ".weight";
        gguf_tensor const& w = get(name);
        const bool oihw = file_whcn && listed(name);
        packed_dw d;
        int kw, kh;
        if (oihw) { kw = (int)w.ne[0]; kh = (int)w.ne[1]; d.C = (int)w.ne[3]; if (w.ne[2] != 1) throw except("tensor %s is not depthwise", name.c_str()); }
        else { d.C = (int)w.ne[0]; kw = (int)w.ne[2]; kh = (int)w.ne[3]; if (w.ne[1] != 1) throw except("tensor %s is not depthwise", name.c_str()); }
        if (kw != 3 || kh != 3) throw except("tensor %s: expected a 3x3 depthwise kernel", name.c_str());
        d.w = ab.alloc((size_t)9 * d.C * 2);
        d.b = ab.alloc((size_t)d.C * 4);
        if (with_data) {
            uint16_t* dst = reinterpret_cast<uint16_t*>(ab.data.data() + d.w);
            for (int c = 0; c < d.C; ++c)
                for (int tap = 0; tap < 9; ++tap) {
                    const size_t src = oihw ? (size_t)c * 9 + tap : (size_t)tap * d.C + c;
                    dst[(size_t)tap * d.C + c] = f32_to_f16(tensor_at(w, src));
                }
            gguf_tensor const& b = get(prefix + ".bias");
            float* bd = reinterpret_cast<float*>(ab.data.data() + d.b);
            for (int c = 0; c < d.C; ++c) bd[c] = tensor_at(b, c);
        }
        return d;
    }
};

} // namespace

} // namespace visp
