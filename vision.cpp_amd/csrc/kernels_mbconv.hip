// MBConv second half in one launch (TinyViT mb_conv, reference src/visp/arch/mobile-sam.cpp:77-92):
//   y = gelu(x + conv3(gelu(depthwise3x3(h) + b2)) + b3)        h = gelu(conv1(x)) [B, H, W, 256] from the GEMM family
// The expanded map h is the largest tensor of the encoder (537 MB per 16 images). Unfused, the depthwise output is written
// and read back by the 1x1 conv (two more passes over it); here it only exists as a 64-pixel x 256-channel f16 tile in LDS.
//
// One block = 256 threads = one row segment of 64 output pixels:
//   phase 1 (VALU, as tv_dwconv3x3_kernel): thread = 8 channels x 8 consecutive pixels, 3 x 10 16-byte loads, f32 accumulate,
//            bias + GELU, f16 -> LDS tile A [64 px][256] (16-byte chunks XOR-swizzled with the pixel so MFMA fragment reads
//            and these writes are conflict free);
//   phase 2 (MFMA): wave w owns pixels 32 (w & 1) .. +31 and output channels 32 (w >> 1) .. +31: D^T = W3 * A^T on
//            v_mfma_f32_32x32x16_f16 (16 k-steps), so a lane owns one pixel and 4 consecutive channels per accumulator group.
//            The W3 fragments (L2 resident, 16 KB per wave) are loaded straight into the registers the depthwise phase has
//            released, issued before the barrier. The host packs W3 in fragment order [nt][k-step][lane][8]
//            (vx_mbconv_pack_w3) so one load instruction covers 1 KB contiguous: from the row-major [64][256] layout the same
//            loads touch 32 cache lines each and cost twice the whole depthwise phase in L1 cycles (measured: no gain over
//            depthwise + GEMM, as with W3 staged in LDS at 2 blocks per CU);
//   epilogue: + b3 + x (8-byte loads), GELU, 8-byte f16x4 stores.
// Blocks are ordered XCD-aware (blocks b and b+8 share an L2): the three output rows that read one input row stay in one L2.
#include "vx_common.h"

namespace {

constexpr int MB_C = 256, MB_CO = 64, MB_PX = 64, MB_P = 8;
constexpr int MB_ROW = MB_C * 2; // bytes per LDS row (one pixel of A, one output channel of W3)

__device__ __forceinline__ float mb_gelu(float x) { // ggml_gelu (tanh form) = x * sigmoid(2u) = x / (1 + exp2(x * w)): 5 VALU + v_exp + v_rcp
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f; // -2 sqrt(2/pi) log2(e)
    const float c3 = c1 * 0.044715f;
    const float w = fmaf(x * x, c3, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * w)); // exp2 -> inf gives -0, exp2 -> 0 gives x
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void mbconv_dw_pw_kernel(const f16* __restrict__ h, const f16* __restrict__ w2, const float* __restrict__ b2,
                                                           const f16* __restrict__ w3, const float* __restrict__ b3, const f16* __restrict__ x,
                                                           f16* __restrict__ y, int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[MB_PX * MB_ROW];
    unsigned char* const sa = smem; // A: [64 px][256] f16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long blk;
    {
        const long nwg = gridDim.x, b = blockIdx.x;
        const long q = nwg >> 3, rem = nwg & 7, xcd = b & 7;
        blk = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int segs = W / MB_PX;
    const int seg = (int)(blk % segs);
    long q = blk / segs;
    const int oy = (int)(q % H), b = (int)(q / H);
    const int x_seg = seg * MB_PX;

    // ---- phase 1: depthwise 3x3 + bias + GELU for 8 channels x 8 pixels
    const int c8 = tid & 31, strip = tid >> 5;
    const int ox0 = x_seg + strip * MB_P;
    {
        f16x8 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f16x8*>(w2 + (long)t * MB_C + c8 * 8);
        float acc[MB_P][8];
#pragma unroll
        for (int p = 0; p < MB_P; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            const f16* row = h + ((long)b * H + iy) * W * MB_C + c8 * 8;
            f16x8 col[MB_P + 2];
#pragma unroll
            for (int t = 0; t < MB_P + 2; ++t) {
                const int ix = ox0 - 1 + t;
                if ((unsigned)ix < (unsigned)W) col[t] = *reinterpret_cast<const f16x8*>(row + (long)ix * MB_C);
                else col[t] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int p = 0; p < MB_P; ++p)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[p][j] = fmaf((float)col[p + kx][j], (float)wv[ky * 3 + kx][j], acc[p][j]);
        }
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = b2[c8 * 8 + j];
#pragma unroll
        for (int p = 0; p < MB_P; ++p) {
            const int px = strip * MB_P + p;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)mb_gelu(acc[p][j] + bv[j]);
            *reinterpret_cast<f16x8*>(sa + px * MB_ROW + ((c8 ^ (px & 15)) * 16)) = o;
        }
    }
    // ---- phase 2: D^T[n][px] = sum_k W3[n][k] * A[px][k]; W3 fragments issued before the barrier
    const int r = lane & 31, hh = lane >> 5;
    const int mt = wave & 1, nt = wave >> 1;
    const int px = mt * 32 + r;
    f16x8 wf[MB_C / 16];
#pragma unroll
    for (int s = 0; s < MB_C / 16; ++s) wf[s] = *reinterpret_cast<const f16x8*>(w3 + ((long)(nt * (MB_C / 16) + s) * 64 + lane) * 8); // fragment order
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < MB_C / 16; ++s) {
        const int chunk = 2 * s + hh;
        const f16x8 af = *reinterpret_cast<const f16x8*>(sa + px * MB_ROW + ((chunk ^ (px & 15)) * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], af, acc, 0, 0, 0);
    }
    // ---- epilogue: lane owns pixel px, channels nt*32 + 8g + 4hh + j
    const long pix = ((long)b * H + oy) * W + x_seg + px;
    const f16* xr = x + pix * MB_CO + nt * 32;
    f16* yr = y + pix * MB_CO + nt * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * hh;
        const f16x4 xv = *reinterpret_cast<const f16x4*>(xr + c);
        const float4 bb = *reinterpret_cast<const float4*>(b3 + nt * 32 + c);
        const float bq[4] = {bb.x, bb.y, bb.z, bb.w};
        f16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (f16)mb_gelu(acc[g * 4 + j] + bq[j] + (float)xv[j]);
        *reinterpret_cast<f16x4*>(yr + c) = o;
    }
}

} // namespace

extern "C" {

// W3 row-major f16 [64][256] -> fragment order: out[((nt * 16 + s) * 64 + lane) * 8 + j] = W3[nt * 32 + (lane & 31)][(2 s + (lane >> 5)) * 8 + j]
int vx_mbconv_pack_w3(const void* w3_rows, void* packed) {
    VX_REQUIRE(w3_rows && packed, "vx_mbconv_pack_w3: null pointer");
    const uint16_t* src = static_cast<const uint16_t*>(w3_rows);
    uint16_t* dst = static_cast<uint16_t*>(packed);
    for (int nt = 0; nt < MB_CO / 32; ++nt)
        for (int s = 0; s < MB_C / 16; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    dst[((size_t)(nt * (MB_C / 16) + s) * 64 + lane) * 8 + j] = src[(size_t)(nt * 32 + (lane & 31)) * MB_C + (2 * s + (lane >> 5)) * 8 + j];
    return 1;
}

int vx_mbconv_dw_pw_supported(int C, int Cout, int W) { return C == MB_C && Cout == MB_CO && W > 0 && W % MB_PX == 0; }

int vx_mbconv_dw_pw_f16(const void* h, const void* w2, const float* b2, const void* w3, const float* b3, const void* x, void* y, int B, int H, int W,
                        int C, int Cout, void* stream) {
    VX_REQUIRE(h && w2 && b2 && w3 && b3 && x && y && B > 0 && H > 0, "vx_mbconv_dw_pw_f16: bad operands");
    VX_REQUIRE(vx_mbconv_dw_pw_supported(C, Cout, W), "vx_mbconv_dw_pw_f16: built for C = %d, Cout = %d, W %% %d == 0 (C = %d, Cout = %d, W = %d)", MB_C,
               MB_CO, MB_PX, C, Cout, W);
    const long blocks = (long)B * H * (W / MB_PX);
    hipLaunchKernelGGL(mbconv_dw_pw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(h),
                       reinterpret_cast<const f16*>(w2), b2, reinterpret_cast<const f16*>(w3), b3, reinterpret_cast<const f16*>(x),
                       reinterpret_cast<f16*>(y), B, H, W);
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
