// Kernels of the graph executor (csrc/graph.cpp) that are not one of the fused families: the glue ops the reference's arch code
// emits between its matrix products -- ggml_add / ggml_mul with broadcasting, ggml_gelu / relu / scale as stand-alone nodes,
// slice / concat / repeat / permute + cont as one strided copy, and the im2col of patch_embed (nn.cpp:166-180) on the f32 input
// tensor the reference uploads. All of them are HBM-bound byte movers: 16-byte lanes where the shapes allow it, grid-stride,
// 32-bit index math per element only in the generic path.
#include "vx_common.h"

namespace {

__device__ __forceinline__ float gelu_tanh_g(float x) { // ggml_gelu, the form of kernels_gemm.hip's epilogue
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    const float c3 = c1 * 0.044715f;
    float w = fmaf(x * x, c3, c1);
    float e = __builtin_amdgcn_exp2f(x * w);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

inline int blocks_for(int64_t n, int per_block = 256) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b)); // at most 64 blocks per CU, grid-stride beyond
}

struct copy_shape {
    long ne[4];
    long ss[4]; // source strides (elements); 0 = broadcast
    long ds[4]; // destination strides
};

// generic: one element per thread-iteration
__global__ __launch_bounds__(256) void copy_strided_kernel(const f16* __restrict__ src, f16* __restrict__ dst, copy_shape s, long n, float scale, int scaled) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long r = i;
        const long i0 = r % s.ne[0]; r /= s.ne[0];
        const long i1 = r % s.ne[1]; r /= s.ne[1];
        const long i2 = r % s.ne[2];
        const long i3 = r / s.ne[2];
        f16 v = src[i0 * s.ss[0] + i1 * s.ss[1] + i2 * s.ss[2] + i3 * s.ss[3]];
        if (scaled) v = (f16)((float)v * scale);
        dst[i0 * s.ds[0] + i1 * s.ds[1] + i2 * s.ds[2] + i3 * s.ds[3]] = v;
    }
}
// rows of 8-element groups: ne[0] counts groups, strides of dimension 0 are 1 (in groups), the others are in elements
__global__ __launch_bounds__(256) void copy_strided8_kernel(const f16* __restrict__ src, f16* __restrict__ dst, copy_shape s, long n, float scale, int scaled) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long r = i;
        const long i0 = r % s.ne[0]; r /= s.ne[0];
        const long i1 = r % s.ne[1]; r /= s.ne[1];
        const long i2 = r % s.ne[2];
        const long i3 = r / s.ne[2];
        f16x8 v = *reinterpret_cast<const f16x8*>(src + i0 * 8 + i1 * s.ss[1] + i2 * s.ss[2] + i3 * s.ss[3]);
        if (scaled) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] * scale);
        }
        *reinterpret_cast<f16x8*>(dst + i0 * 8 + i1 * s.ds[1] + i2 * s.ds[2] + i3 * s.ds[3]) = v;
    }
}

template <typename T>
__device__ __forceinline__ float ld(const T* p, long i) { return (float)p[i]; }

// y = a (op) b[i mod period]; f16 or f32 on every side
template <int OP, typename TA, typename TB, typename TY>
__global__ __launch_bounds__(256) void binary_kernel(const TA* __restrict__ a, const TB* __restrict__ b, long period, TY* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float av = ld(a, i), bv = ld(b, period == n ? i : i % period);
        y[i] = (TY)(OP == 0 ? av + bv : av * bv);
    }
}
// the common case: f16 (op) f16 -> f16, n and period multiples of 8
template <int OP, typename TB>
__global__ __launch_bounds__(256) void binary8_kernel(const f16* __restrict__ a, const TB* __restrict__ b, long period, f16* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f16x8 av = *reinterpret_cast<const f16x8*>(a + i * 8);
        const long bo = (i * 8) % period;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float bv = (float)b[bo + j];
            o[j] = (f16)(OP == 0 ? (float)av[j] + bv : (float)av[j] * bv);
        }
        *reinterpret_cast<f16x8*>(y + i * 8) = o;
    }
}

// 0 gelu, 1 relu, 2 scale
template <int OP>
__global__ __launch_bounds__(256) void unary_kernel(const f16* __restrict__ x, f16* __restrict__ y, long n, float s) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        f16x8 v = *reinterpret_cast<const f16x8*>(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j];
            v[j] = (f16)(OP == 0 ? gelu_tanh_g(f) : (OP == 1 ? fmaxf(f, 0.0f) : (OP == 3 ? fmaxf(f, f * s) : f * s)));
        }
        *reinterpret_cast<f16x8*>(y + i * 8) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { // tail
        const long i = (n8 << 3) + threadIdx.x;
        const float f = (float)x[i];
        y[i] = (f16)(OP == 0 ? gelu_tanh_g(f) : (OP == 1 ? fmaxf(f, 0.0f) : (OP == 3 ? fmaxf(f, f * s) : f * s)));
    }
}

// ggml_interpolate(NEAREST) on an NHWC f16 map (ml.cpp:782-788 -> ggml upscale: source index = floor(i / (out / in))); 8 channels per thread
__global__ __launch_bounds__(256) void nearest_kernel(const f16* __restrict__ x, f16* __restrict__ y, int H, int W, int C8, int OH, int OW, float sy, float sx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long p = i / C8;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH);
        const long b = p / OH;
        const int iy = min((int)floorf((float)oy / sy), H - 1), ix = min((int)floorf((float)ox / sx), W - 1);
        *reinterpret_cast<f16x8*>(y + i * 8) = *reinterpret_cast<const f16x8*>(x + (((b * H + iy) * W + ix) * C8 + c) * 8);
    }
}

// An f32 image [pixels][C], C <= 16, as ONE 32-channel f16 plane for the LDS-ring conv: channels 0..C-1 the values rounded to f16, channels C..2C-1 what
// the rounding dropped (x - f16(x), itself exact in f16 for an image in [0, 1]), the rest zero. The conv's operand repeats its C input columns, so the
// products see the f32 image (what csrc/esrgan.cpp's tile loader does for the u8 image).
__global__ __launch_bounds__(256) void image_planes_kernel(const float* __restrict__ x, f16* __restrict__ y, int C, long n_pix) {
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (long)gridDim.x * 256) {
        f16 v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = (f16)0;
        for (int c = 0; c < C; ++c) {
            const float f = x[p * C + c];
            const f16 hi = (f16)f;
            v[c] = hi;
            v[C + c] = (f16)(f - (float)hi);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[q * 8 + j];
            *reinterpret_cast<f16x8*>(y + p * 32 + q * 8) = o;
        }
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS* __restrict__ x, TD* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (TD)(float)x[i];
}

// patch_embed's im2col on the f32 image tensor [B, H, W, C] (ggml ne [C, W, H, B]): row = (b, py, px), k = (ky, kx, c), zero
// padded to Kp. One thread writes 8 consecutive k.
__global__ __launch_bounds__(256) void im2col_patches_kernel(const float* __restrict__ x, f16* __restrict__ patches, int H, int W, int C, int ps, int Kp, long n8) {
    const int chunks = Kp >> 3, pw = W / ps, ph = H / ps, kreal = ps * ps * C;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n8; t += (long)gridDim.x * 256) {
        const long row = t / chunks;
        const int ch = (int)(t - row * chunks);
        const int px = (int)(row % pw), py = (int)((row / pw) % ph);
        const long b = row / ((long)pw * ph);
        const float* img = x + (b * H * W + ((long)py * ps * W + px * ps)) * C;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ch * 8 + j;
            float v = 0.0f;
            if (k < kreal) {
                const int ky = k / (ps * C), r = k - ky * (ps * C);
                v = img[(long)ky * W * C + r];
            }
            o[j] = (f16)v;
        }
        *reinterpret_cast<f16x8*>(patches + row * Kp + ch * 8) = o;
    }
}

// 1x1 convolution to ONE output channel (depth / mask heads: depth-anything.cpp:91-94): out f32 [M] = scale * act(sum_c x[m, c] w[c] + bias).
// Eight lanes share a pixel when C >= 64; HBM-bound either way.
__global__ __launch_bounds__(256) void conv1x1_to1_kernel(const f16* __restrict__ x, const float* __restrict__ w, float bias, int relu, float scale,
                                                           float* __restrict__ out, long M, int C) {
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const f16* row = x + m * C;
        float acc = bias;
        for (int c = 0; c < C; c += 8) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(row + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf((float)v[j], w[c + j], acc);
        }
        if (relu) acc = fmaxf(acc, 0.0f);
        out[m] = acc * scale;
    }
}

} // namespace

extern "C" int vx_conv1x1_to1_f32(const void* x, const float* w, float bias, int relu, float scale, float* out, int64_t M, int C, void* stream) {
    VX_REQUIRE(x && w && out && M > 0 && C > 0 && C % 8 == 0, "vx_conv1x1_to1_f32: bad operands (C = %d: a multiple of 8)", C);
    hipLaunchKernelGGL(conv1x1_to1_kernel, dim3(blocks_for(M)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(x), w, bias, relu, scale, out, (long)M, C);
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_copy_strided_f16(const void* src, void* dst, const int64_t ne[4], const int64_t src_stride[4], const int64_t dst_stride[4], float scale,
                                   void* stream) {
    VX_REQUIRE(src && dst, "vx_copy_strided_f16: null operand");
    copy_shape s;
    long n = 1;
    for (int i = 0; i < 4; ++i) {
        VX_REQUIRE(ne[i] > 0 && src_stride[i] >= 0 && dst_stride[i] >= 0, "vx_copy_strided_f16: bad extent or stride in dimension %d", i);
        s.ne[i] = ne[i]; s.ss[i] = src_stride[i]; s.ds[i] = dst_stride[i];
        n *= ne[i];
    }
    const int scaled = scale != 1.0f;
    bool vec = ne[0] % 8 == 0 && src_stride[0] == 1 && dst_stride[0] == 1 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int i = 1; i < 4; ++i) vec = vec && src_stride[i] % 8 == 0 && dst_stride[i] % 8 == 0;
    if (vec) {
        s.ne[0] = ne[0] / 8;
        n /= 8;
        hipLaunchKernelGGL(copy_strided8_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(src), reinterpret_cast<f16*>(dst), s, n, scale, scaled);
    } else {
        hipLaunchKernelGGL(copy_strided_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(src), reinterpret_cast<f16*>(dst), s, n, scale, scaled);
    }
    VX_LAUNCH_CHECK();
    return 1;
}

template <int OP>
static int launch_binary(const void* a, int a_f32, const void* b, int b_f32, int64_t period, void* y, int y_f32, int64_t n, hipStream_t st) {
    const bool al16 = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (!a_f32 && !y_f32 && n % 8 == 0 && period % 8 == 0 && al16) {
        if (b_f32) hipLaunchKernelGGL((binary8_kernel<OP, float>), dim3(blocks_for(n / 8)), dim3(256), 0, st, (const f16*)a, (const float*)b, (long)period, (f16*)y, (long)(n / 8));
        else hipLaunchKernelGGL((binary8_kernel<OP, f16>), dim3(blocks_for(n / 8)), dim3(256), 0, st, (const f16*)a, (const f16*)b, (long)period, (f16*)y, (long)(n / 8));
        return 1;
    }
    const dim3 g(blocks_for(n)), t(256);
#define VX_BIN(TA, TB, TY) hipLaunchKernelGGL((binary_kernel<OP, TA, TB, TY>), g, t, 0, st, (const TA*)a, (const TB*)b, (long)period, (TY*)y, (long)n)
    switch ((a_f32 ? 4 : 0) | (b_f32 ? 2 : 0) | (y_f32 ? 1 : 0)) {
        case 0: VX_BIN(f16, f16, f16); break;
        case 1: VX_BIN(f16, f16, float); break;
        case 2: VX_BIN(f16, float, f16); break;
        case 3: VX_BIN(f16, float, float); break;
        case 4: VX_BIN(float, f16, f16); break;
        case 5: VX_BIN(float, f16, float); break;
        case 6: VX_BIN(float, float, f16); break;
        default: VX_BIN(float, float, float); break;
    }
#undef VX_BIN
    return 1;
}

extern "C" int vx_binary_rows(int op, const void* a, int a_f32, const void* b, int b_f32, int64_t b_period, void* y, int y_f32, int64_t n, void* stream) {
    VX_REQUIRE(a && b && y && n > 0 && b_period > 0 && n % b_period == 0, "vx_binary_rows: bad operands (n = %ld, period = %ld)", (long)n, (long)b_period);
    VX_REQUIRE(op == 0 || op == 1, "vx_binary_rows: op %d (0 add, 1 mul)", op);
    if (op == 0) launch_binary<0>(a, a_f32, b, b_f32, b_period, y, y_f32, n, as_stream(stream));
    else launch_binary<1>(a, a_f32, b, b_f32, b_period, y, y_f32, n, as_stream(stream));
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_unary_f16(int op, const void* x, void* y, int64_t n, float s, void* stream) {
    VX_REQUIRE(x && y && n > 0, "vx_unary_f16: bad operands");
    VX_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "vx_unary_f16: operands must be 16-byte aligned");
    const dim3 g(blocks_for(n / 8 + 1)), t(256);
    switch (op) {
        case 0: hipLaunchKernelGGL(unary_kernel<0>, g, t, 0, as_stream(stream), (const f16*)x, (f16*)y, (long)n, s); break;
        case 1: hipLaunchKernelGGL(unary_kernel<1>, g, t, 0, as_stream(stream), (const f16*)x, (f16*)y, (long)n, s); break;
        case 2: hipLaunchKernelGGL(unary_kernel<2>, g, t, 0, as_stream(stream), (const f16*)x, (f16*)y, (long)n, s); break;
        case 3: hipLaunchKernelGGL(unary_kernel<3>, g, t, 0, as_stream(stream), (const f16*)x, (f16*)y, (long)n, s); break;
        default: VX_REQUIRE(false, "vx_unary_f16: op %d (0 gelu, 1 relu, 2 scale, 3 leaky relu)", op);
    }
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_nearest_f16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, void* stream) {
    VX_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0, "vx_nearest_f16: bad operands (C a multiple of 8)");
    const long n = (long)B * OH * OW * (C / 8);
    hipLaunchKernelGGL(nearest_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), (const f16*)x, (f16*)y, H, W, C / 8, OH, OW, (float)OH / (float)H, (float)OW / (float)W, n);
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_image_planes_f32(const float* x, void* y, int64_t n_pix, int C, void* stream) {
    VX_REQUIRE(x && y && n_pix > 0 && C >= 1 && C <= 16, "vx_image_planes_f32: 1..16 channels");
    hipLaunchKernelGGL(image_planes_kernel, dim3(blocks_for(n_pix)), dim3(256), 0, as_stream(stream), x, (f16*)y, C, (long)n_pix);
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_convert(const void* x, int x_f32, void* y, int y_f32, int64_t n, void* stream) {
    VX_REQUIRE(x && y && n > 0 && x_f32 != y_f32, "vx_convert: bad operands");
    if (x_f32) hipLaunchKernelGGL((convert_kernel<float, f16>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), (const float*)x, (f16*)y, (long)n);
    else hipLaunchKernelGGL((convert_kernel<f16, float>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), (const f16*)x, (float*)y, (long)n);
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_im2col_patches_f32(const float* x, void* patches, int B, int H, int W, int C, int ps, int Kp, void* stream) {
    VX_REQUIRE(x && patches && B > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && C > 0, "vx_im2col_patches_f32: the image must be a whole number of patches");
    VX_REQUIRE(Kp % 8 == 0 && Kp >= ps * ps * C, "vx_im2col_patches_f32: Kp = %d must be a multiple of 8 and at least %d", Kp, ps * ps * C);
    const long n8 = (long)B * (H / ps) * (W / ps) * (Kp / 8);
    hipLaunchKernelGGL(im2col_patches_kernel, dim3(blocks_for(n8)), dim3(256), 0, as_stream(stream), x, reinterpret_cast<f16*>(patches), H, W, C, ps, Kp, n8);
    VX_LAUNCH_CHECK();
    return 1;
}
