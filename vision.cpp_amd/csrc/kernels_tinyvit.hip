// Non-GEMM kernels of the TinyViT image encoder (MobileSAM; reference src/visp/arch/mobile-sam.cpp:20-215): depthwise
// 3x3 convolution, LayerNorm with the window partition folded into its row order, window attention with relative
// position bias, window reverse + residual, and two elementwise helpers. The 1x1 convolutions, linears and the strided
// 3x3 convolutions run on the GEMM family (kernels_gemm.hip). First version of this row: written for parity and
// coalesced memory access; the attention core still runs on the VALU (one lane per query).
#include "vx_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float gelu_tanh_f(float x) { // ggml_gelu (tanh form), the reference's activation everywhere
    // 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + exp2(x * w)): exp2 -> inf gives -0, exp2 -> 0 gives x (no inf/inf);
    // v_rcp instead of an IEEE division (a dozen VALU ops per element: the depthwise kernels were VALU bound on it)
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f; // -2 sqrt(2/pi) log2(e)
    const float c3 = c1 * 0.044715f;
    const float w = fmaf(x * x, c3, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * w));
}

// ---- u8 rgb -> f16 [pixels][8]: (v/255 - mean) / std as value (channels 0..2) + f16 rounding residue (3..5), 6..7 zero
// (sam_process_input, mobile-sam.cpp:533-547; the first conv's weights are duplicated on channels 3..5)
__global__ void tv_preprocess_kernel(const uint8_t* __restrict__ rgb, f16* __restrict__ out, long n_pix) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = ((float)rgb[i * 3 + c] / 255.0f + (0.0f - mean[c])) * (1.0f / stdv[c]); // image_u8_to_f32(img, -mean, 1/std)
        const f16 hi = (f16)v;
        o[c] = hi;
        o[3 + c] = (f16)(v - (float)hi);
    }
    reinterpret_cast<f16x8*>(out)[i] = o;
}

// ---- depthwise 3x3, pad 1, stride 1 or 2, NHWC f16, weights [9][C] f16 (tap-major), bias f32 [C]; optional GELU
// (conv_2d_depthwise + add_bias_2d, nn.cpp:102-115). One thread = P consecutive output pixels of one row x 8 channels: an
// input row segment of (P-1)*S+3 pixels is loaded once and feeds all P outputs (3.75 loads per output at P = 8 instead
// of 9), the nine weight vectors stay packed f16 in registers (v_fma_mix reads them directly). Lanes run over the
// channel groups first, so every load / store instruction covers whole pixels (C * 2 contiguous bytes).
template <int S, int P>
__global__ __launch_bounds__(256) void tv_dwconv3x3_kernel(const f16* __restrict__ x, const f16* __restrict__ w, const float* __restrict__ bias,
                                                           f16* __restrict__ y, int B, int H, int W, int C, int OH, int OW, int gelu) {
    constexpr int NC = (P - 1) * S + 3;
    const int c8n = C >> 3, strips = (OW + P - 1) / P;
    const long n = (long)B * OH * strips * c8n;
    // XCD-aware block order: blocks b and b+8 share an XCD and its L2. Every XCD gets a contiguous run of the row-major
    // sequence, so the three output rows that read one input row fetch it through ONE L2 instead of three.
    long blk;
    {
        const long nwg = gridDim.x, b = blockIdx.x;
        const long q = nwg >> 3, rem = nwg & 7, xcd = b & 7;
        blk = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const long i = blk * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % c8n);
    long q = i / c8n;
    const int ox0 = (int)(q % strips) * P;
    q /= strips;
    const int oy = (int)(q % OH), b = (int)(q / OH);
    f16x8 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f16x8*>(w + (long)t * C + c8 * 8);
    float acc[P][8];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = 0.0f;
    const int ix0 = ox0 * S - 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * S - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
        const f16* row = x + ((long)b * H + iy) * W * C + c8 * 8;
        f16x8 col[NC];
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const int ix = ix0 + t;
            if ((unsigned)ix < (unsigned)W) col[t] = *reinterpret_cast<const f16x8*>(row + (long)ix * C);
            else col[t] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[p][j] = fmaf((float)col[p * S + kx][j], (float)wv[ky * 3 + kx][j], acc[p][j]);
    }
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias[c8 * 8 + j];
    f16* out = y + (((long)b * OH + oy) * OW + ox0) * C + c8 * 8;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (ox0 + p >= OW) break;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[p][j] + bv[j];
            if (gelu) v = gelu_tanh_f(v);
            o[j] = (f16)v;
        }
        *reinterpret_cast<f16x8*>(out + (long)p * C) = o;
    }
}

// ---- LayerNorm over channels of f16 rows (layer_norm, nn.cpp:14-19). A row is handled by a group of G lanes (G = 16, 32
// or 64, the smallest with 8 G >= C), each lane holding 8 consecutive channels (one 16-byte load), so a wave normalises
// 64 / G rows per pass; two-pass statistics in f32 by xor shuffles inside the group.
// ws > 0: the output rows are in window order (window_partition, mobile-sam.cpp:25-46): row = ((b*nw + wy)*nw + wx)*ws*ws
// + iy*ws + ix reads pixel (wy*ws+iy, wx*ws+ix); pixels beyond res are the zero padding, whose norm is the bias vector.
// out_f32: write f32 instead of f16 (the encoder's final LayerNorm2d).
template <int G>
__global__ __launch_bounds__(256) void tv_layernorm_kernel(const f16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           void* __restrict__ y, long rows_out, int C, float eps, int res, int ws, int nw,
                                                           int out_f32) {
    const int gl = threadIdx.x & (G - 1);
    const long row = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const bool live = row < rows_out; // dead groups still take part in the shuffles
    long src = row;
    bool padded = !live;
    if (ws > 0 && live) {
        const int N = ws * ws;
        const int in = (int)(row % N);
        long wq = row / N;
        const int wx = (int)(wq % nw);
        wq /= nw;
        const int wy = (int)(wq % nw), bb = (int)(wq / nw);
        const int py = wy * ws + in / ws, px = wx * ws + in % ws;
        padded = py >= res || px >= res;
        src = ((long)bb * res + py) * res + px;
    }
    const int c0 = gl * 8;
    const bool has = c0 < C;
    float v[8];
    if (has && !padded) {
        const f16x8 xv = *reinterpret_cast<const f16x8*>(x + src * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)xv[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv_c = 1.0f / (float)C; // uniform
    const float mean = sum * inv_c;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = has ? v[j] - mean : 0.0f;
        sq += v[j] * v[j];
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = __builtin_amdgcn_rsqf(fmaf(sq, inv_c, eps)); // v_rsq_f32 instead of an IEEE division + sqrt per lane
    if (!live || !has) return;
    const float4 w0 = *reinterpret_cast<const float4*>(w + c0), w1 = *reinterpret_cast<const float4*>(w + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + c0), b1 = *reinterpret_cast<const float4*>(b + c0 + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j] * rstd * wv[j] + bv[j];
    if (out_f32) {
        float* dst = reinterpret_cast<float*>(y) + row * C + c0;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (f16)o[j];
        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(y) + row * C + c0) = ov;
    }
}

// ---- window_reverse + residual (mobile-sam.cpp:48-64, 146-149): y[b, py, px, :] = x[b, py, px, :] + a[window row of (py, px), :]
__global__ void tv_window_reverse_add_kernel(const f16* __restrict__ a, const f16* __restrict__ x, f16* __restrict__ y, int B, int res, int ws,
                                             int nw, int C) {
    const int c8n = C >> 3;
    const long n = (long)B * res * res * c8n;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % c8n);
    long q = i / c8n;
    const int px = (int)(q % res);
    q /= res;
    const int py = (int)(q % res), b = (int)(q / res);
    const long row = (((long)b * nw + py / ws) * nw + px / ws) * ws * ws + (py % ws) * ws + px % ws;
    const f16x8 av = *reinterpret_cast<const f16x8*>(a + row * C + c8 * 8);
    const f16x8 xv = *reinterpret_cast<const f16x8*>(x + i * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)av[j] + (float)xv[j]);
    *reinterpret_cast<f16x8*>(y + i * 8) = o;
}

// ---- y = gelu(a + b) on f16 (tail of mb_conv, mobile-sam.cpp:88-90); b may be NULL (plain GELU)
__global__ void tv_add_gelu_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f16x8 av = reinterpret_cast<const f16x8*>(a)[i];
    f16x8 o;
    if (b) {
        const f16x8 bv = reinterpret_cast<const f16x8*>(b)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)gelu_tanh_f((float)av[j] + (float)bv[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)gelu_tanh_f((float)av[j]);
    }
    reinterpret_cast<f16x8*>(y)[i] = o;
}

unsigned blocks_for(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// ---- y = a + b[i mod period] on f16 (a may be f32): the decoder's "queries + query_pe", "keys + key_pe" and
// "image_embeddings + no_mask_embed" adds (mobile-sam.cpp:331-358, 437-440)
template <typename TA>
__global__ void tv_add_rows_kernel(const TA* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y, long n8, long period8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f16x8 bv = reinterpret_cast<const f16x8*>(b)[i % period8];
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[i * 8 + j] + (float)bv[j]);
    reinterpret_cast<f16x8*>(y)[i] = o;
}

// ---- attention with few queries or few keys (SAM mask decoder, mobile-sam.cpp:306-320 -> nn.cpp:210-244): q [Nq][H*hd],
// k, v [Nk][H*hd], out [Nq][H*hd], hd <= 32, Nk <= 4096. One block per (query, head): scores in LDS, two block reductions,
// then the 256 threads split the keys for the weighted sum of V. Launch-latency sized work (7 tokens x 4096 pixels).
constexpr int SA_MAX_KEYS = 4096;
__global__ __launch_bounds__(256) void tv_small_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                                 f16* __restrict__ out, int Nk, int H, int hd, float scale) {
    __shared__ float sc[SA_MAX_KEYS];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int qi = blockIdx.x / H, h = blockIdx.x - qi * H;
    const int C = H * hd;
    float qv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) qv[d] = d < hd ? (float)q[(long)qi * C + h * hd + d] * scale : 0.0f;
    float m = -INFINITY;
    for (int j = tid; j < Nk; j += 256) {
        const f16* kr = k + (long)j * C + h * hd;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < 32; ++d)
            if (d < hd) s = fmaf(qv[d], (float)kr[d], s);
        sc[j] = s;
        m = fmaxf(m, s);
    }
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float sum = 0.0f;
    for (int j = tid; j < Nk; j += 256) {
        const float p = __expf(sc[j] - m);
        sc[j] = p;
        sum += p;
    }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float inv = 1.0f / red[0];
    __syncthreads();
    const int d = tid % hd, g = tid / hd, G = 256 / hd;
    float acc = 0.0f;
    for (int j = g; j < Nk; j += G) acc = fmaf(sc[j], (float)v[(long)j * C + h * hd + d], acc);
    red[tid] = acc;
    __syncthreads();
    if (tid < hd) {
        float t = 0.0f;
        for (int gg = 0; gg < G; ++gg) t += red[gg * hd + tid];
        out[(long)qi * C + h * hd + tid] = (f16)(t * inv);
    }
}

// few keys (image -> token attention of the mask decoder: 4096 queries, 7 keys): one thread per (query, head), everything in registers
constexpr int SA_FEW_KEYS = 16;
__global__ __launch_bounds__(256) void tv_few_keys_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                                    f16* __restrict__ out, int Nq, int Nk, int H, int hd, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nq * H) return;
    const int qi = i / H, h = i - qi * H;
    const int C = H * hd;
    float qv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) qv[d] = d < hd ? (float)q[(long)qi * C + h * hd + d] * scale : 0.0f;
    float s[SA_FEW_KEYS];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SA_FEW_KEYS; ++j) {
        s[j] = -INFINITY;
        if (j < Nk) {
            const f16* kr = k + (long)j * C + h * hd;
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < 32; ++d)
                if (d < hd) a = fmaf(qv[d], (float)kr[d], a);
            s[j] = a;
        }
        m = fmaxf(m, s[j]);
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < SA_FEW_KEYS; ++j) {
        s[j] = __expf(s[j] - m); // exp(-inf) = 0 for the unused slots
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    for (int d = 0; d < hd; ++d) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < SA_FEW_KEYS; ++j)
            if (j < Nk) a = fmaf(s[j], (float)v[(long)j * C + h * hd + d], a);
        out[(long)qi * C + h * hd + d] = (f16)(a * inv);
    }
}

// ---- sam::interpolate_bilinear (mobile-sam.cpp:485-516): half-pixel centres, source clamped at 0 and extent - 1, for the two
// passes of sam_process_mask (:556-583). src element i lives at src[i * step] (a column of the [pixels][8] mask GEMM output);
// out_u8: threshold at 0 -> 0 / 255, else f32. Contraction is off so the coordinates round as in the reference's C++.
template <typename TS>
__global__ void tv_sam_interp_kernel(const TS* __restrict__ src, int sw, int sh, int sstride, int step, void* __restrict__ dst, int dw, int dh,
                                     int out_u8) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dw * dh) return;
    const int x = i % dw, y = i / dw;
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    const float sxf = fmaxf(((float)x + 0.5f) * scale_x - 0.5f, 0.0f), syf = fmaxf(((float)y + 0.5f) * scale_y - 0.5f, 0.0f);
    const int x0 = (int)sxf, y0 = (int)syf;
    const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    const float v00 = (float)src[((long)y0 * sstride + x0) * step], v01 = (float)src[((long)y0 * sstride + x1) * step];
    const float v10 = (float)src[((long)y1 * sstride + x0) * step], v11 = (float)src[((long)y1 * sstride + x1) * step];
    const float wx = sxf - (float)x0, wy = syf - (float)y0;
    const float v0 = (1 - wx) * v00 + wx * v01, v1 = (1 - wx) * v10 + wx * v11;
    const float v = (1 - wy) * v0 + wy * v1;
    if (out_u8) reinterpret_cast<uint8_t*>(dst)[i] = v > 0.0f ? 255 : 0;
    else reinterpret_cast<float*>(dst)[i] = v;
}

} // namespace

extern "C" {

int vx_tv_preprocess(const uint8_t* rgb, void* out, int64_t n_pixels, void* stream) {
    VX_REQUIRE(rgb && out && n_pixels > 0, "vx_tv_preprocess: bad operands");
    hipLaunchKernelGGL(tv_preprocess_kernel, dim3(blocks_for(n_pixels)), dim3(256), 0, as_stream(stream), rgb, reinterpret_cast<f16*>(out), (long)n_pixels);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_dwconv3x3_f16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int stride, int gelu, void* stream) {
    VX_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, "vx_dwconv3x3_f16: bad operands");
    VX_REQUIRE(C % 8 == 0 && (stride == 1 || stride == 2), "vx_dwconv3x3_f16: C %% 8 == 0 and stride 1 or 2 only (C = %d, stride = %d)", C, stride);
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    const f16* xp = reinterpret_cast<const f16*>(x);
    const f16* wp = reinterpret_cast<const f16*>(w);
    f16* yp = reinterpret_cast<f16*>(y);
    if (stride == 1) {
        constexpr int P = 8;
        const long n = (long)B * OH * ((OW + P - 1) / P) * (C / 8);
        hipLaunchKernelGGL((tv_dwconv3x3_kernel<1, P>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), xp, wp, bias, yp, B, H, W, C, OH, OW, gelu);
    } else {
        constexpr int P = 4;
        const long n = (long)B * OH * ((OW + P - 1) / P) * (C / 8);
        hipLaunchKernelGGL((tv_dwconv3x3_kernel<2, P>), dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), xp, wp, bias, yp, B, H, W, C, OH, OW, gelu);
    }
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_layernorm_f16(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int res, int ws, int out_f32,
                     void* stream) {
    VX_REQUIRE(x && w && b && y && rows_out > 0 && C > 0 && C <= 512 && C % 8 == 0, "vx_layernorm_f16: bad operands (C = %d: a multiple of 8, at most 512)", C);
    const int nw = ws > 0 ? (res + ws - 1) / ws : 0;
    VX_REQUIRE(ws == 0 || rows_out % ((int64_t)nw * nw * ws * ws) == 0, "vx_layernorm_f16: rows do not form whole images of %d x %d windows", nw, nw);
    const f16* xp = reinterpret_cast<const f16*>(x);
    const long rows = (long)rows_out;
    hipStream_t s = as_stream(stream);
    if (C <= 128) hipLaunchKernelGGL(tv_layernorm_kernel<16>, dim3(blocks_for(rows, 16)), dim3(256), 0, s, xp, w, b, y, rows, C, eps, res, ws, nw, out_f32);
    else if (C <= 256) hipLaunchKernelGGL(tv_layernorm_kernel<32>, dim3(blocks_for(rows, 8)), dim3(256), 0, s, xp, w, b, y, rows, C, eps, res, ws, nw, out_f32);
    else hipLaunchKernelGGL(tv_layernorm_kernel<64>, dim3(blocks_for(rows, 4)), dim3(256), 0, s, xp, w, b, y, rows, C, eps, res, ws, nw, out_f32);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_window_reverse_add_f16(const void* a, const void* x, void* y, int B, int res, int ws, int C, void* stream) {
    VX_REQUIRE(a && x && y && B > 0 && res > 0 && ws > 0 && C % 8 == 0, "vx_window_reverse_add_f16: bad operands");
    const long n = (long)B * res * res * (C / 8);
    hipLaunchKernelGGL(tv_window_reverse_add_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(a),
                       reinterpret_cast<const f16*>(x), reinterpret_cast<f16*>(y), B, res, ws, (res + ws - 1) / ws, C);
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_add_gelu_f16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    VX_REQUIRE(a && y && n > 0 && n % 8 == 0, "vx_add_gelu_f16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(tv_add_gelu_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(a),
                       reinterpret_cast<const f16*>(b), reinterpret_cast<f16*>(y), (long)(n / 8));
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_add_rows_f16(const void* a, int a_is_f32, const void* b, int64_t b_period, void* y, int64_t n, void* stream) {
    VX_REQUIRE(a && b && y && n > 0 && n % 8 == 0 && b_period > 0 && b_period % 8 == 0, "vx_add_rows_f16: n and the period must be positive multiples of 8");
    const long n8 = (long)(n / 8);
    if (a_is_f32)
        hipLaunchKernelGGL(tv_add_rows_kernel<float>, dim3(blocks_for(n8)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float*>(a),
                           reinterpret_cast<const f16*>(b), reinterpret_cast<f16*>(y), n8, (long)(b_period / 8));
    else
        hipLaunchKernelGGL(tv_add_rows_kernel<f16>, dim3(blocks_for(n8)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(a),
                           reinterpret_cast<const f16*>(b), reinterpret_cast<f16*>(y), n8, (long)(b_period / 8));
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_small_attention_f16(const void* q, const void* k, const void* v, void* out, int Nq, int Nk, int heads, int hd, void* stream) {
    VX_REQUIRE(q && k && v && out && Nq > 0 && Nk > 0 && heads > 0, "vx_small_attention_f16: bad operands");
    VX_REQUIRE(Nk <= SA_MAX_KEYS && hd > 0 && hd <= 32 && 256 % hd == 0, "vx_small_attention_f16: at most %d keys, head_dim a power of two <= 32 (Nk = %d, hd = %d)",
               SA_MAX_KEYS, Nk, hd);
    if (Nk <= SA_FEW_KEYS && Nq >= 256)
        hipLaunchKernelGGL(tv_few_keys_attention_kernel, dim3(blocks_for((long)Nq * heads)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                           reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v), reinterpret_cast<f16*>(out), Nq, Nk, heads, hd,
                           1.0f / sqrtf((float)hd));
    else
        hipLaunchKernelGGL(tv_small_attention_kernel, dim3((unsigned)Nq * heads), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                           reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v), reinterpret_cast<f16*>(out), Nk, heads, hd,
                           1.0f / sqrtf((float)hd));
    VX_LAUNCH_CHECK();
    return 1;
}

int vx_sam_interpolate(const void* src, int src_f16, int sw, int sh, int sstride, int step, void* dst, int dw, int dh, int out_u8, void* stream) {
    VX_REQUIRE(src && dst && sw > 0 && sh > 0 && dw > 0 && dh > 0 && step > 0 && sstride >= sw, "vx_sam_interpolate: bad operands");
    const long n = (long)dw * dh;
    if (src_f16)
        hipLaunchKernelGGL(tv_sam_interp_kernel<f16>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(src), sw, sh,
                           sstride, step, dst, dw, dh, out_u8);
    else
        hipLaunchKernelGGL(tv_sam_interp_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float*>(src), sw, sh,
                           sstride, step, dst, dw, dh, out_u8);
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
