// 3x3 / stride 1 / pad 1 NHWC convolution for small channel counts (Cin, Cout in {32, 64}) on
// gfx950 -- the DPT fusion residual units and the head convs at 148^2 .. 518^2
// (reference src/visp/nn.cpp:72-100 conv_2d; src/visp/arch/depth-anything.cpp:15-23, 81-94).
//
// The implicit-GEMM form (kernels_gemm.hip) re-reads every input pixel 9 times through L2 and
// is bound by that traffic at these sizes (Cout = 32 gives 25 FLOP per staged byte). Here a block
// owns an 8 x 32 output tile and stages its 10 x 34 input halo in LDS ONCE (global_load_lds_dwordx4,
// lane-linear image, 16-byte chunks XOR-swizzled through the source address so the MFMA fragment
// reads of 32 neighbouring pixels are bank-conflict free). The nine taps are then nine shifted reads
// of that halo: the pixel fragment of tap (ky,kx) for output row y is halo row y+ky, pixels kx..kx+31.
// Weight fragments (at most 73 KB, L2 / L1 resident, identical for every block) are loaded straight
// from global memory into registers, so the tap loop has no barrier at all.
// MFMA orientation is "swapped" (D = W-fragment x pixel-fragment): a lane owns one pixel and four
// consecutive output channels per register group; the tile is staged through LDS (reusing the halo
// space) and written with coalesced 16-byte stores, with the epilogues of the GEMM family
// (bias, ReLU, two residual addends, fused 1x1 head output).
#include "vx_common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile (rows x cols); one wave = 2 rows
constexpr int HH = TH + 2, HW = TW + 2;        // halo
constexpr int HALO_PIX = HH * HW;              // 340

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) unsigned char g_conv_zero_page[64];

template <int CIN>
__device__ __forceinline__ int halo_swz(int pix, int chunk) {
    if constexpr (CIN == 64) return chunk ^ ((pix >> 1) & 7);
    else return chunk ^ ((pix >> 2) & 3);
}

template <int CIN, int COUT, int EPI>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const vx_gemm_args p) {
    constexpr int CH = CIN / 8;                                // 16-byte chunks per pixel
    constexpr int PIX_BYTES = CIN * 2;
    constexpr int HALO_CHUNKS = HALO_PIX * CH;
    constexpr int HALO_INSTR = (HALO_CHUNKS + 63) / 64;        // 1 KiB wave instructions
    constexpr int HALO_BYTES = HALO_INSTR * 1024;
    constexpr int NI = COUT / 32;
    constexpr int KS = CIN / 16;                               // k-steps per tap
    constexpr int NCH16 = COUT / 8;                            // 16-byte chunks per staged output row
    constexpr int PITCH = COUT * 2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const s_bias = reinterpret_cast<float*>(smem + (HALO_BYTES > TH * TW * PITCH ? HALO_BYTES : TH * TW * PITCH));
    float* const s_w3 = s_bias + COUT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = p.conv_H, W = p.conv_W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tile = blockIdx.x;
    const int b = tile / (tiles_x * tiles_y), trem = tile - b * (tiles_x * tiles_y);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    if (tid < COUT) {
        s_bias[tid] = p.bias ? p.bias[tid] : 0.0f;
        if constexpr (EPI == VX_EPI_HEAD_OUT) s_w3[tid] = p.lambda[tid];
    }

    // ---- halo: chunk index L = pix*CH + phys, lane-linear; logical chunk = phys ^ swz(pix)
    const f16* __restrict__ X = reinterpret_cast<const f16*>(p.A) + (long)b * H * W * CIN;
    for (int i = wave; i < HALO_INSTR; i += 4) {
        const int L = i * 64 + lane;
        const int pix = L / CH, phys = L - pix * CH;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const f16* src = reinterpret_cast<const f16*>(g_conv_zero_page);
        if (pix < HALO_PIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            src = X + ((long)iy * W + ix) * CIN + halo_swz<CIN>(pix, phys) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + i * 1024), 16, 0, 0);
    }

    // ---- weight fragments straight from global: W [COUT][Kp], k = (ky*3+kx)*CIN + c
    const f16* __restrict__ Wg = reinterpret_cast<const f16*>(p.W);
    const int Kp = p.K;
    auto load_w = [&](int tap, f16x8 (&wf)[NI][KS]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[ni][ks] = *reinterpret_cast<const f16x8*>(Wg + (long)(ni * 32 + r) * Kp + tap * CIN + ks * 16 + h * 8);
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    f16x8 wf[2][NI][KS];
    load_w(0, wf[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // halo (and first weights) landed
    __syncthreads();

    // wave owns output rows 2*wave, 2*wave+1 of the tile; lane r = pixel column
    const bool relu_in = p.a_relu != 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        if (tap + 1 < 9) load_w(tap + 1, wf[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 af[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int pix = (2 * wave + mi + ky) * HW + kx + r;
                af[mi] = *reinterpret_cast<const f16x8*>(smem + pix * PIX_BYTES + halo_swz<CIN>(pix, ks * 2 + h) * 16);
                if (relu_in) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) af[mi][j] = af[mi][j] > (f16)0 ? af[mi][j] : (f16)0;
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap & 1][ni][ks], af[mi], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads(); // every wave is done with the halo: reuse it as the output staging buffer

    // ---- epilogue phase 1: lane = pixel (row 2*wave+mi, col r), registers = 4 consecutive channels
    unsigned char* const st = smem;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int ml = (2 * wave + mi) * TW + r; // staged row = pixel index in the tile
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = ni * 32 + 8 * g + 4 * h;
                const float4 bias = *reinterpret_cast<const float4*>(s_bias + nl);
                float v[4] = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y,
                              acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (EPI == VX_EPI_F16_RELU || EPI == VX_EPI_HEAD_OUT) v[j] = fmaxf(v[j], 0.0f);
                    if constexpr (EPI == VX_EPI_F16_ADD) { if (p.relu) v[j] = fmaxf(v[j], 0.0f); }
                }
                f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                const int c8 = nl >> 2;
                const int phys16 = (c8 >> 1) ^ (ml & (NCH16 - 1));
                *reinterpret_cast<f16x4*>(st + ml * PITCH + phys16 * 16 + (c8 & 1) * 8) = o;
            }
    }
    __syncthreads();
    // ---- phase 2: coalesced 16-byte chunks; NCH16 consecutive lanes cover one pixel
    constexpr int CHUNKS = TH * TW * NCH16;
#pragma unroll
    for (int it = 0; it < CHUNKS / 256; ++it) {
        const int id = tid + it * 256;
        const int ml = id / NCH16, j = id % NCH16;
        const int oy = y0 + ml / TW, ox = x0 + ml % TW;
        const bool valid = oy < H && ox < W;
        f16x8 v = *reinterpret_cast<const f16x8*>(st + ml * PITCH + (j ^ (ml & (NCH16 - 1))) * 16);
        const long pixel = ((long)b * H + oy) * W + ox;
        if constexpr (EPI == VX_EPI_HEAD_OUT) {
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) part += (float)v[q] * s_w3[j * 8 + q];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if (valid && j == 0) reinterpret_cast<float*>(p.out)[pixel] = fmaxf(part + p.head_bias, 0.0f) * p.head_scale;
        } else {
            if (!valid) continue;
            const long o = pixel * p.ldo + j * 8;
            if constexpr (EPI == VX_EPI_F16_ADD) {
                if (p.res1) {
                    f16x8 a = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.res1) + o);
                    if (p.res2) {
                        f16x8 c = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.res2) + o);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (f16)((float)v[q] + (float)a[q] + (float)c[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (f16)((float)v[q] + (float)a[q]);
                    }
                }
            }
            *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + o) = v;
        }
    }
}

template <int CIN, int COUT, int EPI>
int launch_conv(const vx_gemm_args& a, hipStream_t s) {
    constexpr int CH = CIN / 8;
    constexpr int HALO_BYTES = ((HALO_PIX * CH + 63) / 64) * 1024;
    constexpr int OUT_BYTES = TH * TW * COUT * 2;
    constexpr int smem = (HALO_BYTES > OUT_BYTES ? HALO_BYTES : OUT_BYTES) + 2 * COUT * 4;
    const int B = a.M / (a.conv_OH * a.conv_OW);
    const int tiles = B * ((a.conv_H + TH - 1) / TH) * ((a.conv_W + TW - 1) / TW);
    hipLaunchKernelGGL((conv3x3_halo_kernel<CIN, COUT, EPI>), dim3(tiles), dim3(256), smem, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

template <int CIN, int COUT>
int dispatch_conv_epi(const vx_gemm_args& a, hipStream_t s) {
    switch (a.epi) {
        case VX_EPI_F16: return launch_conv<CIN, COUT, VX_EPI_F16>(a, s);
        case VX_EPI_F16_RELU: return launch_conv<CIN, COUT, VX_EPI_F16_RELU>(a, s);
        case VX_EPI_F16_ADD: return launch_conv<CIN, COUT, VX_EPI_F16_ADD>(a, s);
        case VX_EPI_HEAD_OUT:
            if constexpr (COUT == 32) return launch_conv<CIN, COUT, VX_EPI_HEAD_OUT>(a, s);
            break;
        default: break;
    }
    vx_set_error("vx_conv3x3_f16: unsupported epilogue %d", a.epi);
    return 0;
}

} // namespace

extern "C" int vx_conv3x3_supported(const vx_gemm_args* a) {
    return a->conv_kh == 3 && a->conv_kw == 3 && a->conv_stride == 1 && a->conv_pad == 1 && (a->conv_Cin == 32 || a->conv_Cin == 64) &&
           (a->N == 32 || a->N == 64) && (a->n_valid == 0 || a->n_valid == a->N) && a->conv_OH == a->conv_H && a->conv_OW == a->conv_W &&
           (a->epi == VX_EPI_F16 || a->epi == VX_EPI_F16_RELU || a->epi == VX_EPI_F16_ADD || (a->epi == VX_EPI_HEAD_OUT && a->N == 32));
}

extern "C" int vx_conv3x3_f16(const vx_gemm_args* args, void* stream) {
    const vx_gemm_args& a = *args;
    VX_REQUIRE(vx_conv3x3_supported(args), "vx_conv3x3_f16: unsupported shape (3x3 s1 p1, Cin/Cout in {32,64} only)");
    VX_REQUIRE(a.K >= 9 * a.conv_Cin && a.A && a.W && a.out, "vx_conv3x3_f16: bad operands");
    VX_REQUIRE(a.M % (a.conv_OH * a.conv_OW) == 0, "vx_conv3x3_f16: M is not a whole number of images");
    hipStream_t s = as_stream(stream);
    if (a.conv_Cin == 64 && a.N == 64) return dispatch_conv_epi<64, 64>(a, s);
    if (a.conv_Cin == 64 && a.N == 32) return dispatch_conv_epi<64, 32>(a, s);
    if (a.conv_Cin == 32 && a.N == 32) return dispatch_conv_epi<32, 32>(a, s);
    return dispatch_conv_epi<32, 64>(a, s);
}
