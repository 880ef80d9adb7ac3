// Row kernels of the SWIN encoder (BiRefNet backbone; reference src/visp/arch/swin.cpp) on gfx950. All of them are HBM-bound
// gathers / scatters fused with the LayerNorm or residual add next to them, so that pad, roll, window_partition,
// window_reverse, slice and concat of the reference graph (swin.cpp:48-76, 117-161) never exist as copies:
//
//  * swin_ln_rows_kernel     norm1 + pad + roll(-shift) + window_partition: output row = window token, read from its source
//                            pixel (zeros where the padded map has no pixel); with ws = 0 a plain LayerNorm of rows (norm2,
//                            patch_embed.norm, the per-stage output norms with f32 output).
//  * swin_merge_ln_kernel    patch_merging's 2x2 gather + LayerNorm(4C) (swin.cpp:140-161), input of the reduction GEMM.
//  * swin_window_reverse_add window_reverse + roll(+shift) + crop + shortcut add.
//
// 16, 32 or 64 lanes per row (narrow rows share a wave), a lane holds up to NCH chunks of 8 channels (16-byte loads), statistics by
// sub-wave reduction, two-pass (mean, then centred variance) in registers as ggml_norm does (nn.cpp:14-19).
#include "vx_common.h"

namespace {

inline unsigned blocks_for(long items, int per_block = 256) { return (unsigned)((items + per_block - 1) / per_block); }

// sum over the LPR lanes that share a row (LPR = 16, 32 or 64 consecutive lanes)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LayerNorm of one row held by LPR lanes as v[NCH][8] (chunk c of row-lane l covers channels 8 (l + LPR c) .. +7; chunks beyond C
// hold zeros). Narrow rows share a wave: 64 / LPR rows per wave, so that C = 96 keeps 12 of 16 lanes busy instead of 12 of 64.
template <int LPR, int NCH>
__device__ __forceinline__ void ln_row_store(float (&v)[NCH][8], int C, const float* __restrict__ w, const float* __restrict__ b, float eps, void* yrow,
                                             int out_f32, int lane, bool store) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[c][j];
    const float mean = row_sum<LPR>(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool live = (lane + LPR * c) * 8 < C;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[c][j] = live ? v[c][j] - mean : 0.0f;
            q += v[c][j] * v[c][j];
        }
    }
    const float rstd = 1.0f / sqrtf(row_sum<LPR>(q) / (float)C + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = (lane + LPR * c) * 8;
        if (ch >= C || !store) continue;
        const float4 w0 = *reinterpret_cast<const float4*>(w + ch), w1 = *reinterpret_cast<const float4*>(w + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + ch), b1 = *reinterpret_cast<const float4*>(b + ch + 4);
        const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = v[c][j] * rstd * ww[j] + bb[j];
        if (out_f32) {
            float* o = static_cast<float*>(yrow) + ch;
            *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
        } else {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)r[j];
            *reinterpret_cast<f16x8*>(static_cast<f16*>(yrow) + ch) = o;
        }
    }
}

template <int LPR, int NCH>
__global__ __launch_bounds__(256) void swin_ln_rows_kernel(const f16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, void* y,
                                                           long rows_out, int C, float eps, int H, int W, int ws, int shift, int out_f32, int ldy) {
    constexpr int RPB = 256 / LPR; // rows per block
    const int lane = threadIdx.x % LPR;
    long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live_row = row < rows_out; // dead rows of the last wave still take part in the shuffles
    if (!live_row) row = rows_out - 1;
    long src = row;
    if (ws > 0) { // window token -> source pixel of the padded, rolled map (swin.cpp:128-139)
        const int N = ws * ws, nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws, wp = nwx * ws, hp = nwy * ws;
        const long win = row / N;
        const int t = (int)(row - win * N), iy = t / ws, ix = t - iy * ws;
        const long img = win / ((long)nwx * nwy);
        const int wi = (int)(win - img * nwx * nwy), wy = wi / nwx, wx = wi - wy * nwx;
        int sy = wy * ws + iy + shift, sx = wx * ws + ix + shift;
        if (sy >= hp) sy -= hp;
        if (sx >= wp) sx -= wp;
        src = (sy < H && sx < W) ? (img * H + sy) * W + sx : -1;
    }
    void* yrow = out_f32 ? static_cast<void*>(static_cast<float*>(y) + row * ldy) : static_cast<void*>(static_cast<f16*>(y) + row * ldy);
    const bool pad_row = src < 0;
    if (pad_row) { // padding: zeros, not a normalised zero row (the reference pads AFTER norm1)
        if (live_row) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = (lane + LPR * c) * 8;
                if (ch < C) *reinterpret_cast<uint4*>(static_cast<f16*>(yrow) + ch) = make_uint4(0, 0, 0, 0);
            }
        }
        src = 0; // keep the lanes in the shuffles below; nothing is stored
    }
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = (lane + LPR * c) * 8;
        f16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ch < C) t = *reinterpret_cast<const f16x8*>(x + src * C + ch);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = (float)t[j];
    }
    ln_row_store<LPR, NCH>(v, C, w, b, eps, yrow, out_f32, lane, live_row && !pad_row);
}

// output row (img, oy, ox) = LayerNorm over [x(2oy,2ox) | x(2oy+1,2ox) | x(2oy,2ox+1) | x(2oy+1,2ox+1)], 4C channels
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void swin_merge_ln_kernel(const f16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            f16* __restrict__ y, long rows_out, int C, float eps, int H, int W) {
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x % LPR;
    long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live_row = row < rows_out;
    if (!live_row) row = rows_out - 1;
    const int ow = W / 2, oh = H / 2;
    const long img = row / ((long)ow * oh);
    const int t = (int)(row - img * ow * oh), oy = t / ow, ox = t - oy * ow;
    const int C4 = 4 * C;
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = (lane + LPR * c) * 8;
        f16x8 tv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ch < C4) {
            const int part = ch / C, cc = ch - part * C; // C % 8 == 0: a chunk never straddles two parts
            const int sy = 2 * oy + (part & 1), sx = 2 * ox + (part >> 1);
            tv = *reinterpret_cast<const f16x8*>(x + ((img * H + sy) * W + sx) * C + cc);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = (float)tv[j];
    }
    ln_row_store<LPR, NCH>(v, C4, w, b, eps, y + row * C4, 0, lane, live_row);
}

// y[img, sy, sx, :] = x[img, sy, sx, :] + a[window row of the padded position that the roll mapped to (sy, sx), :]
__global__ __launch_bounds__(256) void swin_window_reverse_add_kernel(const f16* __restrict__ a, const f16* __restrict__ x, f16* __restrict__ y, long n8,
                                                                      int H, int W, int C, int ws, int shift) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c8 = C / 8;
    const long pix = i / c8;
    const int ch = (int)(i - pix * c8) * 8;
    const long img = pix / ((long)H * W);
    const int t = (int)(pix - img * H * W), sy = t / W, sx = t - sy * W;
    const int nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws, wp = nwx * ws, hp = nwy * ws, N = ws * ws;
    int py = sy - shift, px = sx - shift;
    if (py < 0) py += hp;
    if (px < 0) px += wp;
    const long arow = ((img * nwy + py / ws) * nwx + px / ws) * N + (py % ws) * ws + px % ws;
    const f16x8 av = *reinterpret_cast<const f16x8*>(a + arow * C + ch), xv = *reinterpret_cast<const f16x8*>(x + pix * C + ch);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)av[j] + (float)xv[j]);
    *reinterpret_cast<f16x8*>(y + pix * C + ch) = o;
}

} // namespace

extern "C" {

static int swin_layernorm_impl(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int H, int W, int ws, int shift,
                               int out_f32, int ldy, void* stream);

int vx_swin_layernorm_f16(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int H, int W, int ws, int shift,
                          int out_f32, void* stream) {
    return swin_layernorm_impl(x, w, b, y, rows_out, C, eps, H, W, ws, shift, out_f32, C, stream);
}

// plain rows with an output row stride (ldy elements, 0 = C): a stage output written straight into a wider concatenation buffer
int vx_swin_layernorm_strided_f16(const void* x, const float* w, const float* b, void* y, int64_t rows, int C, float eps, int ldy, int out_f32, void* stream) {
    VX_REQUIRE(ldy == 0 || (ldy >= C && ldy % 8 == 0), "vx_swin_layernorm_strided_f16: row stride %d for %d channels", ldy, C);
    return swin_layernorm_impl(x, w, b, y, rows, C, eps, 0, 0, 0, 0, out_f32, ldy ? ldy : C, stream);
}

static int swin_layernorm_impl(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int H, int W, int ws, int shift,
                               int out_f32, int ldy, void* stream) {
    VX_REQUIRE(x && w && b && y && rows_out > 0 && C > 0 && C % 8 == 0 && C <= 2048, "vx_swin_layernorm_f16: bad operands (C = %d: a multiple of 8, at most 2048)", C);
    VX_REQUIRE(ws == 0 || (H > 0 && W > 0 && shift >= 0 && shift < ws && !out_f32), "vx_swin_layernorm_f16: bad window arguments");
    if (ws > 0) {
        const int64_t per_image = (int64_t)((H + ws - 1) / ws) * ((W + ws - 1) / ws) * ws * ws;
        VX_REQUIRE(rows_out % per_image == 0, "vx_swin_layernorm_f16: rows do not form whole images of %dx%d maps in windows of %d", W, H, ws);
    }
    const f16* xp = reinterpret_cast<const f16*>(x);
    const long rows = (long)rows_out;
    hipStream_t s = as_stream(stream);
#define LN_LAUNCH(LPR, NCH) hipLaunchKernelGGL((swin_ln_rows_kernel<LPR, NCH>), dim3(blocks_for(rows, 256 / LPR)), dim3(256), 0, s, xp, w, b, y, rows, C, eps, H, W, ws, shift, out_f32, ldy)
    if (C <= 128) LN_LAUNCH(16, 1);
    else if (C <= 256) LN_LAUNCH(32, 1);
    else if (C <= 512) LN_LAUNCH(64, 1);
    else if (C <= 1024) LN_LAUNCH(64, 2);
    else LN_LAUNCH(64, 4);
#undef LN_LAUNCH
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_swin_merge_layernorm_f16(const void* x, const float* w, const float* b, void* y, int B, int H, int W, int C, float eps, void* stream) {
    VX_REQUIRE(x && w && b && y && B > 0 && C > 0 && C % 8 == 0 && 4 * C <= 4096, "vx_swin_merge_layernorm_f16: bad operands (C = %d: a multiple of 8, at most 1024)", C);
    VX_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "vx_swin_merge_layernorm_f16: patch merging expects even spatial dimensions, got %dx%d", W, H);
    const long rows = (long)B * (H / 2) * (W / 2);
    const f16* xp = reinterpret_cast<const f16*>(x);
    f16* yp = reinterpret_cast<f16*>(y);
    hipStream_t s = as_stream(stream);
#define MG_LAUNCH(LPR, NCH) hipLaunchKernelGGL((swin_merge_ln_kernel<LPR, NCH>), dim3(blocks_for(rows, 256 / LPR)), dim3(256), 0, s, xp, w, b, yp, rows, C, eps, H, W)
    if (4 * C <= 128) MG_LAUNCH(16, 1);
    else if (4 * C <= 256) MG_LAUNCH(32, 1);
    else if (4 * C <= 512) MG_LAUNCH(64, 1);
    else if (4 * C <= 1024) MG_LAUNCH(64, 2);
    else if (4 * C <= 2048) MG_LAUNCH(64, 4);
    else MG_LAUNCH(64, 8);
#undef MG_LAUNCH
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_swin_window_reverse_add_f16(const void* a, const void* x, void* y, int B, int H, int W, int C, int ws, int shift, void* stream) {
    VX_REQUIRE(a && x && y && B > 0 && H > 0 && W > 0 && ws > 0 && shift >= 0 && shift < ws && C > 0 && C % 8 == 0, "vx_swin_window_reverse_add_f16: bad operands");
    const long n8 = (long)B * H * W * (C / 8);
    hipLaunchKernelGGL(swin_window_reverse_add_kernel, dim3(blocks_for(n8)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(a),
                       reinterpret_cast<const f16*>(x), reinterpret_cast<f16*>(y), n8, H, W, C, ws, shift);
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
