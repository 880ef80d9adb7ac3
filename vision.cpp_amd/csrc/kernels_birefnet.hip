// Glue kernels of BiRefNet's two-scale encode and decoder (reference src/visp/arch/birefnet.cpp) on gfx950. The convolutions are
// the GEMM family's (kernels_gemm.hip); what is here is everything between them, each fused with the copy the reference graph
// would make (concat, permute, interpolate, image_to_patches):
//
//  * bf_preprocess_half      birefnet_process_input's normalisation + downscale_by(x, 2) (bilinear, align_corners; birefnet.cpp:
//                            35-41, 259-270) -> the half-size image as value + residue pixels for the second SWIN pass
//  * bf_patches              image_to_patches of the normalised image (birefnet.cpp:158-167), straight from the u8 image
//  * bf_resize               bilinear align-corners resize of f16 NHWC maps with row strides on both sides: up / downscales
//                            write into channel slices of the concatenation buffers (encode_concat, upscale_to)
//  * bf_deform_cols          the sampling half of the deformable convolution (deformable_conv_2d, birefnet.cpp:83-92 =
//                            torchvision deform_conv2d): per pixel and tap, offset (dy, dx) and modulator logit come from one
//                            GEMM; the bilinear sample x 2 sigmoid(modulator) is written as the im2col row of the GEMM that
//                            applies the kernel weights (batch-norm scale folded in, ReLU in its epilogue)
//  * bf_mean, bf_broadcast   global_avg_pool (birefnet.cpp:94-108): pixel mean per channel; its 1x1 map "interpolated" back
//  * bf_mul_sigmoid, bf_sigmoid_out   gdt attention (p * sigmoid(a)) and the final mask
#include "vx_common.h"

namespace {

inline unsigned blocks_for(long items, int per_block = 256) { return (unsigned)((items + per_block - 1) / per_block); }

__device__ __forceinline__ float norm_px(unsigned char v, int c) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, inv_std[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};
    return ((float)v * (1.0f / 255.0f) - mean[c]) * inv_std[c]; // image_u8_to_f32(image, -mean, 1 / std): (u8/255 + offset) * scale
}

// out [B, H/2, W/2][8] f16: channels 0..2 the value, 3..5 its f16 rounding residue, 6..7 zero (vx_tv_preprocess layout)
__global__ __launch_bounds__(256) void bf_preprocess_half_kernel(const unsigned char* __restrict__ rgb, f16* __restrict__ out, int B, int H, int W) {
    const int OH = H / 2, OW = W / 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * OH * OW) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long b = i / ((long)OW * OH);
    const float fy = OH > 1 ? (float)oy * ((float)(H - 1) / (float)(OH - 1)) : 0.0f, fx = OW > 1 ? (float)ox * ((float)(W - 1) / (float)(OW - 1)) : 0.0f;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > H - 1) y0 = H - 1;
    if (x0 > W - 1) x0 = W - 1;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = fy - (float)y0, tx = fx - (float)x0;
    const unsigned char* p = rgb + b * (long)H * W * 3;
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = norm_px(p[((long)y0 * W + x0) * 3 + c], c), bq = norm_px(p[((long)y0 * W + x1) * 3 + c], c);
        const float cq = norm_px(p[((long)y1 * W + x0) * 3 + c], c), d = norm_px(p[((long)y1 * W + x1) * 3 + c], c);
        const float top = a + (bq - a) * tx, bot = cq + (d - cq) * tx, v = top + (bot - top) * ty;
        const f16 hv = (f16)v;
        o[c] = hv;
        o[3 + c] = (f16)(v - (float)hv);
    }
    *reinterpret_cast<f16x8*>(out + i * 8) = o;
}

// patches [B, h, w, gw*gh*3]: channel gx + gw (gy + gh c) of pixel (py, px) = normalised image (gy h + py, gx w + px, c).
// A thread writes 8 consecutive channels (16 bytes); gw*gh*3 is a multiple of 8 for the grids in use (4 .. 32), else scalar tail.
__global__ __launch_bounds__(256) void bf_patches_kernel(const unsigned char* __restrict__ rgb, f16* __restrict__ out, int B, int IH, int IW, int h, int w) {
    const int gw = IW / w, gh = IH / h, CP = gw * gh * 3, c8 = (CP + 7) / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * h * w * c8) return;
    const int ch0 = (int)(i % c8) * 8;
    const long pix = i / c8;
    const int px = (int)(pix % w), py = (int)((pix / w) % h);
    const long b = pix / ((long)w * h);
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = ch0 + j;
        if (ch < CP) {
            const int gx = ch % gw, gy = (ch / gw) % gh, c = ch / (gw * gh);
            o[j] = (f16)norm_px(rgb[((b * IH + (long)gy * h + py) * IW + (long)gx * w + px) * 3 + c], c);
        }
    }
    if (ch0 + 8 <= CP) *reinterpret_cast<f16x8*>(out + pix * CP + ch0) = o;
    else
        for (int j = 0; ch0 + j < CP; ++j) out[pix * CP + ch0 + j] = o[j];
}

// bilinear, align_corners, f16 NHWC; src row stride lds, dst row stride ldd (elements); C % 8 == 0
__global__ __launch_bounds__(256) void bf_resize_kernel(const f16* __restrict__ src, int lds, f16* __restrict__ dst, int ldd, int B, int h, int w, int C, int oh,
                                                        int ow, float sy, float sx) {
    const int c8 = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * oh * ow * c8) return;
    const int ch = (int)(i % c8) * 8;
    const long pix = i / c8;
    const int ox = (int)(pix % ow), oy = (int)((pix / ow) % oh);
    const long b = pix / ((long)ow * oh);
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ty = fy - (float)y0, tx = fx - (float)x0;
    const f16* base = src + b * (long)h * w * lds + ch;
    const f16x8 a = *reinterpret_cast<const f16x8*>(base + ((long)y0 * w + x0) * lds), bq = *reinterpret_cast<const f16x8*>(base + ((long)y0 * w + x1) * lds);
    const f16x8 cq = *reinterpret_cast<const f16x8*>(base + ((long)y1 * w + x0) * lds), d = *reinterpret_cast<const f16x8*>(base + ((long)y1 * w + x1) * lds);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float top = (float)a[j] + ((float)bq[j] - (float)a[j]) * tx, bot = (float)cq[j] + ((float)d[j] - (float)cq[j]) * tx;
        o[j] = (f16)(top + (bot - top) * ty);
    }
    *reinterpret_cast<f16x8*>(dst + pix * ldd + ch) = o;
}

// cols [B*h*w, k*k*C]: tap t = ky*k + kx of pixel p holds 2 sigmoid(mod[p, t]) * bilinear(x, y - pad + ky + dy, x - pad + kx + dx) with
// (dy, dx) = om[p, 2t], om[p, 2t+1] and mod[p, t] = om[p, 2 k*k + t] (one GEMM computes offsets and modulator logits side by side);
// torchvision's zero extension: nothing outside (-1, H) x (-1, W), corners outside the map count as zero. stride 1.
__global__ __launch_bounds__(256) void bf_deform_cols_kernel(const f16* __restrict__ x, const f16* __restrict__ om, int ldom, f16* __restrict__ cols, int B, int h,
                                                             int w, int C, int k, int pad) {
    const int c8 = C / 8, taps = k * k;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * h * w * taps * c8) return;
    const int ch = (int)(i % c8) * 8;
    const long pt = i / c8;
    const int t = (int)(pt % taps);
    const long pix = pt / taps;
    const int ox = (int)(pix % w), oy = (int)((pix / w) % h);
    const long b = pix / ((long)w * h);
    const f16* o = om + pix * ldom;
    const float py = (float)(oy - pad + t / k) + (float)o[2 * t], px = (float)(ox - pad + t % k) + (float)o[2 * t + 1];
    f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    if (py > -1.0f && py < (float)h && px > -1.0f && px < (float)w) {
        const float scale = 2.0f / (1.0f + __expf(-(float)o[2 * taps + t]));
        const int y0 = (int)floorf(py), x0 = (int)floorf(px);
        const float ly = py - (float)y0, lx = px - (float)x0;
        const f16* base = x + b * (long)h * w * C + ch;
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 v1 = (y0 >= 0 && x0 >= 0) ? *reinterpret_cast<const f16x8*>(base + ((long)y0 * w + x0) * C) : z;
        const f16x8 v2 = (y0 >= 0 && x0 + 1 <= w - 1) ? *reinterpret_cast<const f16x8*>(base + ((long)y0 * w + x0 + 1) * C) : z;
        const f16x8 v3 = (y0 + 1 <= h - 1 && x0 >= 0) ? *reinterpret_cast<const f16x8*>(base + ((long)(y0 + 1) * w + x0) * C) : z;
        const f16x8 v4 = (y0 + 1 <= h - 1 && x0 + 1 <= w - 1) ? *reinterpret_cast<const f16x8*>(base + ((long)(y0 + 1) * w + x0 + 1) * C) : z;
        const float w1 = (1.0f - ly) * (1.0f - lx), w2 = (1.0f - ly) * lx, w3 = ly * (1.0f - lx), w4 = ly * lx;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)((w1 * (float)v1[j] + w2 * (float)v2[j] + w3 * (float)v3[j] + w4 * (float)v4[j]) * scale);
    }
    *reinterpret_cast<f16x8*>(cols + pt * C + ch) = r;
}

// mean over the pixels of each image: x [B, n, C] (row stride ld) -> y [B, C]. One block per (image, 64-channel group): four pixel
// slices of 64 channels add their pixels in a FIXED order (p = slice, slice + 4, ...; then slice 0 + 1 + 2 + 3), f32, and the block writes
// the sum. No atomics: the first form added per-chunk partial sums with atomicAdd, whose order -- and with it the last bits of the mean,
// 4e-4 on the mask after the decoder -- changed from launch to launch (tests/test_gpu_birefnet.py::test_birefnet_lite_1024_batch_8 met it
// as a batch whose reversed order gave other bits). The map is the 1/32-scale ASPP input (32 x 32 pixels at 1024^2): 256 steps of four
// independent loads per thread.
__global__ __launch_bounds__(256) void bf_mean_partial_kernel(const f16* __restrict__ x, int ld, float* __restrict__ acc, long n, int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f; // four chains per thread, combined in a fixed order
    if (c < C) {
        const f16* xb = x + (long)b * n * ld + c;
        long p = slice;
        for (; p + 12 < n; p += 16) {
            s0 += (float)xb[p * ld];
            s1 += (float)xb[(p + 4) * ld];
            s2 += (float)xb[(p + 8) * ld];
            s3 += (float)xb[(p + 12) * ld];
        }
        for (; p < n; p += 4) s0 += (float)xb[p * ld];
    }
    part[slice][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < C) acc[(long)b * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void bf_mean_finish_kernel(const float* __restrict__ acc, f16* __restrict__ y, long n, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) y[i] = (f16)(acc[i] / (float)n);
}

// dst[b, p, 0:C] (row stride ldd) = g[b, 0:C] for every pixel p of image b (bilinear "interpolation" of a 1 x 1 map)
__global__ __launch_bounds__(256) void bf_broadcast_kernel(const f16* __restrict__ g, int ldg, f16* __restrict__ dst, int ldd, long n, int C, long total8) {
    const int c8 = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int ch = (int)(i % c8) * 8;
    const long row = i / c8, b = row / n;
    *reinterpret_cast<f16x8*>(dst + row * ldd + ch) = *reinterpret_cast<const f16x8*>(g + b * ldg + ch);
}

// y[p, :] *= sigmoid(a[p * lda]) (gdt attention, birefnet.cpp:189-192)
__global__ __launch_bounds__(256) void bf_mul_sigmoid_kernel(f16* __restrict__ y, int ldy, const f16* __restrict__ a, int lda, int C, long total8) {
    const int c8 = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int ch = (int)(i % c8) * 8;
    const long row = i / c8;
    const float s = 1.0f / (1.0f + __expf(-(float)a[row * lda]));
    f16x8 v = *reinterpret_cast<f16x8*>(y + row * ldy + ch);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] * s);
    *reinterpret_cast<f16x8*>(y + row * ldy + ch) = v;
}

__global__ __launch_bounds__(256) void bf_sigmoid_out_kernel(const f16* __restrict__ a, int lda, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0f / (1.0f + expf(-(float)a[i * lda]));
}

} // namespace

extern "C" {

int vx_bf_preprocess_half(const uint8_t* rgb, void* out8, int B, int H, int W, void* stream) {
    VX_REQUIRE(rgb && out8 && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "vx_bf_preprocess_half: bad operands (extent %dx%d must be even)", W, H);
    hipLaunchKernelGGL(bf_preprocess_half_kernel, dim3(blocks_for((long)B * (H / 2) * (W / 2))), dim3(256), 0, as_stream(stream), rgb, reinterpret_cast<f16*>(out8), B, H, W);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_patches(const uint8_t* rgb, void* out, int B, int IH, int IW, int h, int w, void* stream) {
    VX_REQUIRE(rgb && out && B > 0 && h > 0 && w > 0 && IH % h == 0 && IW % w == 0, "vx_bf_patches: grid %dx%d must divide the image %dx%d", w, h, IW, IH);
    const long n = (long)B * h * w * (((IW / w) * (IH / h) * 3 + 7) / 8);
    hipLaunchKernelGGL(bf_patches_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), rgb, reinterpret_cast<f16*>(out), B, IH, IW, h, w);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_resize_f16(const void* src, int lds, void* dst, int ldd, int B, int h, int w, int C, int oh, int ow, void* stream) {
    VX_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && C > 0 && C % 8 == 0 && lds >= C && ldd >= C && lds % 8 == 0 && ldd % 8 == 0,
               "vx_bf_resize_f16: bad operands (C = %d, strides %d / %d)", C, lds, ldd);
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.0f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.0f;
    hipLaunchKernelGGL(bf_resize_kernel, dim3(blocks_for((long)B * oh * ow * (C / 8))), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(src), lds,
                       reinterpret_cast<f16*>(dst), ldd, B, h, w, C, oh, ow, sy, sx);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_deform_cols_f16(const void* x, const void* offmod, int ldom, void* cols, int B, int h, int w, int C, int k, void* stream) {
    VX_REQUIRE(x && offmod && cols && B > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && k >= 1 && k <= 7 && (k & 1) && ldom >= 3 * k * k,
               "vx_bf_deform_cols_f16: bad operands (C = %d, kernel %d, offset row stride %d)", C, k, ldom);
    const long n = (long)B * h * w * k * k * (C / 8);
    hipLaunchKernelGGL(bf_deform_cols_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(x), reinterpret_cast<const f16*>(offmod),
                       ldom, reinterpret_cast<f16*>(cols), B, h, w, C, k, k / 2);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_mean_f16(const void* x, int ld, void* y, float* acc_scratch, int B, int64_t n, int C, void* stream) {
    VX_REQUIRE(x && y && acc_scratch && B > 0 && n > 0 && C > 0 && ld >= C, "vx_bf_mean_f16: bad operands");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(bf_mean_partial_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, reinterpret_cast<const f16*>(x), ld, acc_scratch, (long)n, C);
    hipLaunchKernelGGL(bf_mean_finish_kernel, dim3(blocks_for((long)B * C)), dim3(256), 0, s, acc_scratch, reinterpret_cast<f16*>(y), (long)n, B * C);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_broadcast_f16(const void* g, int ldg, void* dst, int ldd, int B, int64_t n, int C, void* stream) {
    VX_REQUIRE(g && dst && B > 0 && n > 0 && C > 0 && C % 8 == 0 && ldd % 8 == 0 && ldg % 8 == 0, "vx_bf_broadcast_f16: bad operands");
    const long total8 = (long)B * n * (C / 8);
    hipLaunchKernelGGL(bf_broadcast_kernel, dim3(blocks_for(total8)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(g), ldg, reinterpret_cast<f16*>(dst), ldd,
                       (long)n, C, total8);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_mul_sigmoid_f16(void* y, int ldy, const void* a, int lda, int64_t rows, int C, void* stream) {
    VX_REQUIRE(y && a && rows > 0 && C > 0 && C % 8 == 0 && ldy % 8 == 0, "vx_bf_mul_sigmoid_f16: bad operands");
    const long total8 = (long)rows * (C / 8);
    hipLaunchKernelGGL(bf_mul_sigmoid_kernel, dim3(blocks_for(total8)), dim3(256), 0, as_stream(stream), reinterpret_cast<f16*>(y), ldy, reinterpret_cast<const f16*>(a), lda, C,
                       total8);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_bf_sigmoid_out_f32(const void* a, int lda, float* out, int64_t n, void* stream) {
    VX_REQUIRE(a && out && n > 0 && lda > 0, "vx_bf_sigmoid_out_f32: bad operands");
    hipLaunchKernelGGL(bf_sigmoid_out_kernel, dim3(blocks_for((long)n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(a), lda, out, (long)n);
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
