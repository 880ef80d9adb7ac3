// HBM-bound kernels of the Depth-Anything path on gfx950: LayerNorm, pre/post-processing,
// bilinear resize, head output. Each is a coalesced streaming kernel (16-byte lane accesses
// where the layout allows) with wave64 shuffle reductions.
#include "vx_common.h"

namespace {

// ---- LayerNorm: one wave per row (reference src/visp/nn.cpp:14-19 = ggml_norm * w + b)
// x f32 [M, C] -> y f16 [M, C]; biased variance, eps inside the sqrt; two-pass in registers.
template <int VPL> // float2 vectors per lane: C == 64 * 2 * VPL
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, f16* __restrict__ y, int M,
                                                             int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float2* xr = reinterpret_cast<const float2*>(x + (long)row * C);
    float2 v[VPL];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[lane + 64 * i];
        sum += v[i].x + v[i].y;
    }
    const float inv_c = 1.0f / (float)C; // uniform: one scalar division per wave
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i].x -= mean; v[i].y -= mean;
        sq += v[i].x * v[i].x + v[i].y * v[i].y;
    }
    const float rstd = __builtin_amdgcn_rsqf(fmaf(wave_sum(sq), inv_c, eps)); // v_rsq_f32: no IEEE division / sqrt sequence per lane
    const float2* wr = reinterpret_cast<const float2*>(w);
    const float2* br = reinterpret_cast<const float2*>(b);
    f16x2* yr = reinterpret_cast<f16x2*>(y + (long)row * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float2 ww = wr[lane + 64 * i], bb = br[lane + 64 * i];
        f16x2 o = {(f16)(v[i].x * rstd * ww.x + bb.x), (f16)(v[i].y * rstd * ww.y + bb.y)};
        yr[lane + 64 * i] = o;
    }
}

// generic fallback: any C, one wave per row, three strided passes
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, f16* __restrict__ y,
                                                                 int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    float sum = 0.0f;
    for (int c = lane; c < C; c += 64) sum += xr[c];
    const float inv_c = 1.0f / (float)C; // uniform: one scalar division per wave
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.0f;
    for (int c = lane; c < C; c += 64) { float d = xr[c] - mean; sq += d * d; }
    const float rstd = __builtin_amdgcn_rsqf(fmaf(wave_sum(sq), inv_c, eps)); // v_rsq_f32: no IEEE division / sqrt sequence per lane
    for (int c = lane; c < C; c += 64) y[(long)row * C + c] = (f16)((xr[c] - mean) * rstd * w[c] + b[c]);
}

// ---- pre-processing + patch im2col (reference depth-anything.cpp:130-140, image.cpp:215-226,
// image-impl.h:23-27, nn.cpp:166-180). One thread writes 8 consecutive k of one patch row. The value of an element depends on its byte
// and its channel only, so the reference's arithmetic -- (u / 255 - mean) * (1 / std), an IEEE division per element -- is evaluated
// once per (channel, byte) into a 768-entry f16 table in LDS and the kernel is byte loads + table reads (round 3: it was bound by
// vector issue -- the division sequence and two integer divisions by the runtime patch size per element -- at 1.4 TB/s; PS = 14 is a
// compile-time constant for the DINOv2 models, 0 = any patch size).
template <int PS>
__global__ __launch_bounds__(256) void preprocess_patches_kernel(const uint8_t* __restrict__ rgb, f16* __restrict__ patches,
                                                                  int H, int W, int ps_rt, int Kp, float m0, float m1,
                                                                  float m2, float s0, float s1, float s2) {
    __shared__ f16 lut[3 * 256];
    for (int i = threadIdx.x; i < 768; i += 256) {
        const int c = i >> 8;
        const float u = (float)(i & 255);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), inv = c == 0 ? s0 : (c == 1 ? s1 : s2);
        lut[i] = (f16)((u / 255.0f - mean) * inv);
    }
    __syncthreads();
    // grid: (ceil(P*chunks / 256), B); thread = 8 consecutive k of one patch row, 32-bit index math
    const int ps = PS ? PS : ps_rt;
    const int chunks = Kp >> 3;
    const int pw = W / ps, ph = H / ps;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= pw * ph * chunks) return;
    const int prow = t / chunks, ch = t - prow * chunks;
    const int py = prow / pw, px = prow - py * pw;
    const int b = blockIdx.y;
    const int kreal = ps * ps * 3;
    const uint8_t* img = rgb + (long)b * H * W * 3 + ((long)py * ps * W + px * ps) * 3;
    f16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ch * 8 + j;
        f16 v = (f16)0.0f;
        if (k < kreal) {
            const int ky = k / (ps * 3), r = k - ky * (ps * 3); // r = kx * 3 + c: contiguous bytes of one image row
            const int c = r % 3;
            v = lut[c * 256 + img[ky * W * 3 + r]];
        }
        out[j] = v;
    }
    *reinterpret_cast<f16x8*>(patches + ((long)b * pw * ph + prow) * Kp + ch * 8) = out;
}

__global__ __launch_bounds__(256) void preprocess_f32_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ out, long n,
                                                              float m0, float m1, float m2, float s0, float s1, float s2) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % 3);
        float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        float inv = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[i] = ((float)rgb[i] / 255.0f + (-mean)) * inv;
    }
}

__global__ void write_cls_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int T, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(long)b * T * C + c] = cls[c] + pos[c];
}

// ---- bilinear, align_corners (ggml_interpolate BILINEAR|ALIGN_CORNERS; reference ml.cpp:782-788).
// NHWC f16, one thread = 8 channels of one output pixel; source coords as ggml computes them:
// sf = (out-1)/(in-1), src = i / sf.
__global__ __launch_bounds__(256) void bilinear_ac_kernel(const f16* __restrict__ x, f16* __restrict__ y, int H, int W,
                                                           int c8n, int OH, int OW, float sfy, float sfx) {
    // grid: (ceil(OW*c8n / 256), OH, B). One thread = 8 channels of one output pixel; the row terms are
    // block-uniform, everything else is 32-bit arithmetic (c8n = C/8 channel chunks per pixel).
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= OW * c8n) return;
    const int ox = t / c8n, c8 = t - ox * c8n;
    const int oy = blockIdx.y, b = blockIdx.z;
    const int C = c8n * 8;
    const float sy = (float)oy / sfy, sx = (float)ox / sfx;
    int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    int y1 = y0 + 1, x1 = x0 + 1;
    y0 = max(0, min(y0, H - 1)); y1 = max(0, min(y1, H - 1));
    x0 = max(0, min(x0, W - 1)); x1 = max(0, min(x1, W - 1));
    const float dy = fminf(fmaxf(sy - (float)y0, 0.0f), 1.0f), dx = fminf(fmaxf(sx - (float)x0, 0.0f), 1.0f);
    const f16* base = x + (long)b * H * W * C + c8 * 8;
    const f16x8 a = *reinterpret_cast<const f16x8*>(base + (y0 * W + x0) * C);
    const f16x8 bb = *reinterpret_cast<const f16x8*>(base + (y0 * W + x1) * C);
    const f16x8 c = *reinterpret_cast<const f16x8*>(base + (y1 * W + x0) * C);
    const f16x8 d = *reinterpret_cast<const f16x8*>(base + (y1 * W + x1) * C);
    const float w00 = (1 - dx) * (1 - dy), w01 = dx * (1 - dy), w10 = (1 - dx) * dy, w11 = dx * dy;
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (f16)((float)a[j] * w00 + (float)bb[j] * w01 + (float)c[j] * w10 + (float)d[j] * w11);
    *reinterpret_cast<f16x8*>(y + (((long)b * OH + oy) * OW + ox) * C + c8 * 8) = o;
}

// ---- head output: 1x1 conv C -> 1, ReLU, * max_depth (reference depth-anything.cpp:89-94)
template <int C>
__global__ __launch_bounds__(256) void head_out_kernel(const f16* __restrict__ x, const float* __restrict__ w, float bias,
                                                        float max_depth, float* __restrict__ depth, long n) {
    float wr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wr[c] = w[c];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const f16x8* px = reinterpret_cast<const f16x8*>(x + i * C);
        float acc = 0.0f;
#pragma unroll
        for (int c8 = 0; c8 < C / 8; ++c8) {
            f16x8 v = px[c8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[j] * wr[c8 * 8 + j];
        }
        depth[i] = fmaxf(acc + bias, 0.0f) * max_depth;
    }
}

// ---- post-processing (reference image.cpp:537-576): per-image min/max, then v*scale + offset
__device__ __forceinline__ unsigned f2ord(float f) { // order-preserving float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ void minmax_init_kernel(unsigned* mm, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ depth, unsigned* __restrict__ mm, long n) {
    const int b = blockIdx.y;
    const float* d = depth + (long)b * n;
    float mn = INFINITY, mx = -INFINITY;
    const long n4 = ((reinterpret_cast<uintptr_t>(d) & 15) == 0) ? n / 4 : 0; // 16-byte path when the image is aligned
    const float4* d4 = reinterpret_cast<const float4*>(d);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = d4[i];
        mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
    }
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = d[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    // one atomic pair per BLOCK: all 2 B results share a few cache lines, and same-line atomics execute one after the other
    // at the memory side (~5 ns each: with one pair per wave of a 2048-block grid this kernel took 94 us for 34 MB)
    __shared__ float s_mn[4], s_mx[4];
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        atomicMin(&mm[2 * b], f2ord(mn));
        atomicMax(&mm[2 * b + 1], f2ord(mx));
    }
}
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ depth, float* __restrict__ out,
                                                         const unsigned* __restrict__ mm, long n) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const float mn = ord2f(mm[2 * b]), mx = ord2f(mm[2 * b + 1]);
    float delta = mx - mn;
    delta = delta < 1e-5f ? 1.0f : delta;
    // the reference's scalar loop rounds the product and the sum separately (image.cpp:565-575),
    // so the minimum maps to exactly 0: keep the compiler from contracting them into an FMA
    const float scale = 1.0f / delta;          // (max - min) / delta with max=1, min=0
    const float offset = -mn * scale;
    const float* d = depth + (long)b * n;
    float* o = out + (long)b * n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float prod = d[i] * scale;
        o[i] = prod + offset;
    }
}
__global__ __launch_bounds__(256) void f32_to_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = fminf(fmaxf(src[i], 0.0f), 1.0f);
        dst[i] = (uint8_t)(v * 255.0f);
    }
}

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

} // namespace

extern "C" {

int vx_layernorm_f32_f16(const float* x, const float* w, const float* b, void* y, int M, int C, float eps, void* stream) {
    VX_REQUIRE(M > 0 && C > 0, "vx_layernorm: empty problem");
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = as_stream(stream);
    f16* yy = reinterpret_cast<f16*>(y);
    if (C == 384) hipLaunchKernelGGL(layernorm_vec_kernel<3>, grid, block, 0, s, x, w, b, yy, M, C, eps);
    else if (C == 128) hipLaunchKernelGGL(layernorm_vec_kernel<1>, grid, block, 0, s, x, w, b, yy, M, C, eps);
    else if (C == 768) hipLaunchKernelGGL(layernorm_vec_kernel<6>, grid, block, 0, s, x, w, b, yy, M, C, eps);
    else if (C == 1024) hipLaunchKernelGGL(layernorm_vec_kernel<8>, grid, block, 0, s, x, w, b, yy, M, C, eps);
    else hipLaunchKernelGGL(layernorm_generic_kernel, grid, block, 0, s, x, w, b, yy, M, C, eps);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_preprocess_patches(const uint8_t* rgb, void* patches, int B, int H, int W, int ps, int Kp, const float mean[3],
                          const float inv_std[3], void* stream) {
    VX_REQUIRE(H % ps == 0 && W % ps == 0, "vx_preprocess_patches: %dx%d not a multiple of patch size %d", W, H, ps);
    VX_REQUIRE(Kp % 8 == 0 && Kp >= ps * ps * 3, "vx_preprocess_patches: bad Kp=%d", Kp);
    const int per_image = (H / ps) * (W / ps) * (Kp / 8);
    if (ps == 14)
        hipLaunchKernelGGL(preprocess_patches_kernel<14>, dim3((per_image + 255) / 256, B), dim3(256), 0, as_stream(stream), rgb,
                           reinterpret_cast<f16*>(patches), H, W, ps, Kp, mean[0], mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2]);
    else
        hipLaunchKernelGGL(preprocess_patches_kernel<0>, dim3((per_image + 255) / 256, B), dim3(256), 0, as_stream(stream), rgb,
                           reinterpret_cast<f16*>(patches), H, W, ps, Kp, mean[0], mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2]);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_preprocess_f32(const uint8_t* rgb, float* out, int B, int H, int W, const float mean[3], const float inv_std[3],
                      void* stream) {
    long n = (long)B * H * W * 3;
    hipLaunchKernelGGL(preprocess_f32_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), rgb, out, n, mean[0],
                       mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2]);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_write_cls_rows(float* x, const float* cls, const float* pos, int B, int T, int C, void* stream) {
    hipLaunchKernelGGL(write_cls_kernel, dim3(B), dim3(128), 0, as_stream(stream), x, cls, pos, T, C);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bilinear_ac_f16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, void* stream) {
    VX_REQUIRE(C % 8 == 0, "vx_bilinear_ac_f16: C=%d must be a multiple of 8", C);
    float sfy = (OH > 1 && H > 1) ? (float)(OH - 1) / (float)(H - 1) : (float)OH / (float)H;
    float sfx = (OW > 1 && W > 1) ? (float)(OW - 1) / (float)(W - 1) : (float)OW / (float)W;
    VX_REQUIRE((long)H * W * C < 0x7fffffffL, "vx_bilinear_ac_f16: image too large for 32-bit offsets");
    const int c8n = C / 8;
    dim3 grid((OW * c8n + 255) / 256, OH, B);
    hipLaunchKernelGGL(bilinear_ac_kernel, grid, dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(x),
                       reinterpret_cast<f16*>(y), H, W, c8n, OH, OW, sfy, sfx);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_head_out_f32(const void* x, const float* w, float bias, float max_depth, float* depth, int64_t n_pixels, int C,
                    void* stream) {
    hipStream_t s = as_stream(stream);
    const f16* xx = reinterpret_cast<const f16*>(x);
    dim3 grid(grid_for(n_pixels)), block(256);
    if (C == 32) hipLaunchKernelGGL(head_out_kernel<32>, grid, block, 0, s, xx, w, bias, max_depth, depth, (long)n_pixels);
    else if (C == 64) hipLaunchKernelGGL(head_out_kernel<64>, grid, block, 0, s, xx, w, bias, max_depth, depth, (long)n_pixels);
    else if (C == 8) hipLaunchKernelGGL(head_out_kernel<8>, grid, block, 0, s, xx, w, bias, max_depth, depth, (long)n_pixels);
    else if (C == 16) hipLaunchKernelGGL(head_out_kernel<16>, grid, block, 0, s, xx, w, bias, max_depth, depth, (long)n_pixels);
    else { vx_set_error("vx_head_out_f32: unsupported C=%d", C); return 0; }
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_minmax_normalize(const float* depth, float* out, float* minmax, int B, int64_t n, void* stream) {
    hipStream_t s = as_stream(stream);
    unsigned* mm = reinterpret_cast<unsigned*>(minmax);
    hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, mm, B);
    int gx = grid_for(n, 256, 64);
    const int gmm = B >= 16 ? 16 : (B >= 4 ? 32 : 64); // blocks per image of the reduction: ~512 blocks, 2 atomics each
    hipLaunchKernelGGL(minmax_kernel, dim3(gmm < gx ? gmm : gx, B), dim3(256), 0, s, depth, mm, (long)n);
    hipLaunchKernelGGL(normalize_kernel, dim3(gx, B), dim3(256), 0, s, depth, out, mm, (long)n);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_f32_to_u8(const float* src, uint8_t* dst, int64_t n, void* stream) {
    hipLaunchKernelGGL(f32_to_u8_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), src, dst, (long)n);
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
