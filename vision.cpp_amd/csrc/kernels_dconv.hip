// Dense-block 3x3 convolution for the ESRGAN / Real-ESRGAN RRDB stack on gfx950
// (reference src/visp/arch/esrgan.cpp:13-79: conv_block = conv_2d 3x3 s1 p1 + leaky_relu 0.2, dense concat,
//  x5*0.2 + x, nearest x2 upsample before the up-convs).
//
// What is different from the DPT halo kernel (kernels_conv.hip):
//  * the input is a CHANNEL PREFIX of a wider pixel row (x_ld elements per pixel): a residual dense block keeps
//    [x | x1 | x2 | x3 | x4] in one [pixels][192] buffer, every conv reads the first 64+32k channels and writes its
//    32 outputs into the next channel slice, so the reference's four concat copies never happen;
//  * Cin is walked in chunks of 32 channels: a chunk's 18x34 halo (39 KB) and its weight slab 9 x COUT x 32
//    (18/36 KB, pre-swizzled at load time so the copy is linear) are streamed by LDS-DMA into a 2-stage ring while
//    the MFMAs of the previous chunk run; ONE barrier per chunk. Weights come from LDS, not from L1: with 8 waves
//    sharing a slab the L1/TA path carries (halo + slab) once per block instead of one fragment load per wave
//    per k-step, which is what bounds the DPT kernel at Cout = 32;
//  * 512 threads = 8 waves, 16 x 32 output pixels per block, a wave owns two rows (two 32-pixel M-tiles) and all
//    COUT channels; MFMA orientation is swapped (D = W-fragment x pixel-fragment) so a lane owns a pixel and four
//    consecutive channels per register group;
//  * the 36 fragment addresses (9 taps x 2 rows x 2 k-steps, XOR-swizzled) are chunk-invariant and live in VGPRs;
//  * nearest x2 upsampling is folded into the halo source address (esrgan.cpp:13-19), LeakyReLU and the scaled
//    residuals (v*s1 + res1)*s2 + res2 into the epilogue (esrgan.cpp:38-40, 49-50, 64-65);
//  * the RGB head (Cout = 3 padded to 32) writes f32 straight from the accumulators.
#include "vx_common.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) unsigned char g_dconv_zero_page[64];

constexpr int CK = 32;      // channels per chunk
constexpr int PIXB = CK * 2; // bytes per pixel per chunk
constexpr int NW = 8;       // waves per block

template <int COUT, int MT, int TW, int EPI>
__global__ __launch_bounds__(512) void dconv3x3_kernel(const vx_dconv_args p) {
    constexpr int TH = NW * MT * 32 / TW;
    constexpr int HH = TH + 2, HW = TW + 2, HALO_PIX = HH * HW;
    constexpr int HALO_INSTR = (HALO_PIX * 4 + 63) / 64, HALO_BYTES = HALO_INSTR * 1024;
    constexpr int NI = COUT / 32;
    constexpr int W_BYTES = 9 * COUT * PIXB, W_INSTR = W_BYTES / 1024;
    constexpr int HJ = (HALO_INSTR + NW - 1) / NW, WJ = (W_INSTR + NW - 1) / NW;
    constexpr int W_BASE = 2 * HALO_BYTES;
    constexpr int BIAS_BASE = W_BASE + 2 * W_BYTES;
    constexpr int NCH16 = COUT / 8, PITCH = COUT * 2;
    static_assert(TH * TW == NW * MT * 32, "tile does not split into whole M-tiles");
    static_assert(TH * TW * PITCH <= W_BASE + 2 * W_BYTES, "output staging does not fit");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const s_bias = reinterpret_cast<float*>(smem + BIAS_BASE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;

    // XCD-aware order: blocks id and id+8 share an L2; give each XCD a contiguous run of tiles so neighbouring
    // tiles (which share halo rows) and one image's slabs meet in one L2.
    int tile;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int per = total >> 3, rem = total & 7, xcd = id & 7;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (id >> 3);
    }
    const int b = tile / (tiles_x * tiles_y), trem = tile - b * (tiles_x * tiles_y);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    if (tid < COUT) s_bias[tid] = p.bias ? p.bias[tid] : 0.0f;

    // ---- per-lane halo source addresses (chunk 0); later chunks add c*32 channels
    const int up = p.up2 ? 1 : 0;
    const int Hs = H >> up, Ws = W >> up;
    const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x) + (long)b * Hs * Ws * p.x_ld;
    const f16* hsrc[HJ];
    int hstep[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int i = wave + j * NW;
        const int L = i * 64 + lane;
        const int pix = L >> 2, phys = L & 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool valid = pix < HALO_PIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        hsrc[j] = valid ? X + ((long)(iy >> up) * Ws + (ix >> up)) * p.x_ld + ((phys ^ ((pix >> 2) & 3)) << 3)
                        : reinterpret_cast<const f16*>(g_dconv_zero_page);
        hstep[j] = valid ? CK : 0;
    }
    const f16* __restrict__ Wg = reinterpret_cast<const f16*>(p.w) + lane * 8;

    auto issue = [&](int c, int stage) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int i = wave + j * NW;
            if (i < HALO_INSTR)
                __builtin_amdgcn_global_load_lds((gptr_t)(hsrc[j] + c * hstep[j]), (lptr_t)(smem + stage * HALO_BYTES + i * 1024), 16, 0, 0);
        }
        const f16* wc = Wg + (long)c * (W_BYTES / 2);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int i = wave + j * NW;
            if (i < W_INSTR)
                __builtin_amdgcn_global_load_lds((gptr_t)(wc + i * 512), (lptr_t)(smem + W_BASE + stage * W_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // ---- chunk-invariant fragment addresses
    int a_addr[9][MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int f = (wave * MT + mi) * 32 + r;
        const int trow = f / TW, tcol = f - trow * TW;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int pix = (trow + tap / 3) * HW + tcol + tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a_addr[tap][mi][ks] = pix * PIXB + (((ks * 2 + h) ^ ((pix >> 2) & 3)) << 4);
        }
    }
    int w_addr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_addr[ks] = W_BASE + r * PIXB + (((ks * 2 + h) ^ ((r >> 2) & 3)) << 4);

    f32x16 acc[MT][NI];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    auto load_tap = [&](int stage, int tap, f16x8 (&af)[MT][2], f16x8 (&wf)[NI][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni][ks] = *reinterpret_cast<const f16x8*>(smem + w_addr[ks] + stage * W_BYTES + (tap * COUT + ni * 32) * PIXB);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                af[mi][ks] = *reinterpret_cast<const f16x8*>(smem + a_addr[tap][mi][ks] + stage * HALO_BYTES);
        }
    };

    auto compute = [&](int stage) {
        f16x8 af[2][MT][2], wf[2][NI][2];
        load_tap(stage, 0, af[0], wf[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_tap(stage, tap + 1, af[(tap + 1) & 1], wf[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap & 1][ni][ks], af[tap & 1][mi][ks], acc[mi][ni], 0, 0, 0);
        }
    };

    const int nch = p.cin / CK;
    issue(0, 0);
    for (int c = 0; c < nch; c += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // chunk c is in stage 0 for everyone; everyone is done reading stage 1
        if (c + 1 < nch) issue(c + 1, 1);
        compute(0);
        if (c + 1 < nch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c + 2 < nch) issue(c + 2, 0);
            compute(1);
        }
    }

    // ---- epilogue
    if constexpr (EPI == VX_DC_RGB_F32) {
        // channels 0..2 of pixel r sit in acc[mi][0][0..2] of the lanes with h == 0
        if (h == 0) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int f = (wave * MT + mi) * 32 + r;
                const int trow = f / TW, tcol = f - trow * TW;
                const int oy = y0 + trow, ox = x0 + tcol;
                if (oy < H && ox < W) {
                    float* o = reinterpret_cast<float*>(p.out) + (((long)b * H + oy) * W + ox) * 3;
                    o[0] = acc[mi][0][0] + s_bias[0];
                    o[1] = acc[mi][0][1] + s_bias[1];
                    o[2] = acc[mi][0][2] + s_bias[2];
                }
            }
        }
    } else {
        __syncthreads(); // every wave is done with the ring: reuse it as the output staging buffer
        unsigned char* const st = smem;
        const bool lrelu = p.act != 0;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int ml = (wave * MT + mi) * 32 + r; // staged row = pixel index in the tile
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = ni * 32 + 8 * g + 4 * h;
                    const float4 bias = *reinterpret_cast<const float4*>(s_bias + nl);
                    float v[4] = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y,
                                  acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
                    if (lrelu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.2f * v[j]);
                    }
                    f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    const int c8 = nl >> 2;
                    const int phys16 = (c8 >> 1) ^ (ml & (NCH16 - 1));
                    *reinterpret_cast<f16x4*>(st + ml * PITCH + phys16 * 16 + (c8 & 1) * 8) = o;
                }
        }
        __syncthreads();
        constexpr int CHUNKS = TH * TW * NCH16;
        const f16* __restrict__ R1 = reinterpret_cast<const f16*>(p.res1);
        const f16* __restrict__ R2 = reinterpret_cast<const f16*>(p.res2);
        const float s1 = p.s1, s2 = p.s2;
#pragma unroll
        for (int it = 0; it < CHUNKS / 512; ++it) {
            const int id = tid + it * 512;
            const int ml = id / NCH16, j = id % NCH16;
            const int trow = ml / TW, tcol = ml - trow * TW;
            const int oy = y0 + trow, ox = x0 + tcol;
            if (oy >= H || ox >= W) continue;
            f16x8 v = *reinterpret_cast<const f16x8*>(st + ml * PITCH + (j ^ (ml & (NCH16 - 1))) * 16);
            const long pixel = ((long)b * H + oy) * W + ox;
            if (R1) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(R1 + pixel * p.res1_ld + j * 8);
                if (R2) {
                    const f16x8 c = *reinterpret_cast<const f16x8*>(R2 + pixel * p.res2_ld + j * 8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (f16)(((float)v[q] * s1 + (float)a[q]) * s2 + (float)c[q]);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (f16)((float)v[q] * s1 + (float)a[q]);
                }
            }
            *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + pixel * p.ldo + j * 8) = v;
        }
    }
}

template <int COUT, int MT, int TW, int EPI>
int launch_dconv(const vx_dconv_args& a, hipStream_t s) {
    constexpr int TH = NW * MT * 32 / TW;
    constexpr int HALO_BYTES = (((TH + 2) * (TW + 2) * 4 + 63) / 64) * 1024;
    constexpr int smem = 2 * HALO_BYTES + 2 * 9 * COUT * PIXB + COUT * 4;
    static bool attr_set = false;
    if (!attr_set) {
        VX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv3x3_kernel<COUT, MT, TW, EPI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int tiles = a.B * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    hipLaunchKernelGGL((dconv3x3_kernel<COUT, MT, TW, EPI>), dim3(tiles), dim3(512), smem, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

// ---- ESRGAN pre/post-processing --------------------------------------------------------------------------------

// image_u8_to_f32 with a tile offset (reference src/visp/image.cpp:215-255, image-impl.h:17-34: reads are clamped to
// the image) for every tile of every image -> f16 [B*n_tiles][th][tw][32]: channels 0..2 = f16(v/255),
// channels 3..5 = f16(v/255 - f16(v/255)) (the rounding residue; the first conv's weights are duplicated on
// channels 3..5 so the f32 input value is reconstructed inside the f32 accumulator), the rest zero.
__global__ void esr_tiles_in_kernel(const uint8_t* __restrict__ img, int B, int w, int h, int ch, int ir, int ig, int ib,
                                    vx_tile_layout t, f16* __restrict__ out) {
    const long n = (long)B * t.n_x * t.n_y * t.tile_h * t.tile_w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % t.tile_w);
    long q = i / t.tile_w;
    const int y = (int)(q % t.tile_h);
    q /= t.tile_h;
    const int tile = (int)(q % (t.n_x * t.n_y)), b = (int)(q / (t.n_x * t.n_y));
    const int cx = tile % t.n_x, cy = tile / t.n_x;
    const int sx = min(cx * (t.tile_w - t.overlap_x) + x, w - 1), sy = min(cy * (t.tile_h - t.overlap_y) + y, h - 1);
    const uint8_t* s = img + (((long)b * h + sy) * w + sx) * ch;
    const int idx[3] = {ir, ig, ib};
    f16x8 lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)s[idx[c]] / 255.0f;
        const f16 hi = (f16)v;
        lo[c] = hi;
        lo[3 + c] = (f16)(v - (float)hi);
    }
    f16x8* o = reinterpret_cast<f16x8*>(out + i * 32);
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    o[0] = lo;
    o[1] = z;
    o[2] = z;
    o[3] = z;
}

// tile_merge over all tiles + image_f32_to_u8 (reference src/visp/image.cpp:653-693, 257-288): one thread per
// output pixel walks the tiles that cover it in the reference's order (t = cy*n_x + cx ascending) and repeats its
// arithmetic (dst += (weight/norm) * tile, or dst = tile where the weight is zero), so the f32 image is the
// reference's bit for bit given the same tiles.
__device__ __forceinline__ int tl_start(int c, int tile, int overlap, int pad) { return c * (tile - overlap) + (c == 0 ? 0 : pad); }
__device__ __forceinline__ int tl_end(int c, int n, int tile, int overlap, int pad, int image) {
    const int e = c * (tile - overlap) + tile - (c == n - 1 ? 0 : pad);
    return e < image ? e : image;
}

__global__ void esr_tiles_out_kernel(const float* __restrict__ tiles, int B, vx_tile_layout t, float* __restrict__ out_f32,
                                     uint8_t* __restrict__ out_rgba) {
    const long n = (long)B * t.image_h * t.image_w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % t.image_w);
    const int y = (int)((i / t.image_w) % t.image_h);
    const int b = (int)(i / ((long)t.image_w * t.image_h));
    float d[3] = {0.0f, 0.0f, 0.0f};
    {
#pragma clang fp contract(off)
        for (int cy = 0; cy < t.n_y; ++cy) {
            const int by = tl_start(cy, t.tile_h, t.overlap_y, 0), ey = tl_end(cy, t.n_y, t.tile_h, t.overlap_y, 0, t.image_h);
            if (y < by || y >= ey) continue;
            const int pby = tl_start(cy, t.tile_h, t.overlap_y, t.overlap_y), pey = tl_end(cy, t.n_y, t.tile_h, t.overlap_y, t.overlap_y, t.image_h);
            for (int cx = 0; cx < t.n_x; ++cx) {
                const int bx = tl_start(cx, t.tile_w, t.overlap_x, 0), ex = tl_end(cx, t.n_x, t.tile_w, t.overlap_x, 0, t.image_w);
                if (x < bx || x >= ex) continue;
                const int pbx = tl_start(cx, t.tile_w, t.overlap_x, t.overlap_x), pex = tl_end(cx, t.n_x, t.tile_w, t.overlap_x, t.overlap_x, t.image_w);
                float weight = 1.0f;
                int covx = 0, covy = 0;
                if (x < pbx) { weight *= (float)(t.overlap_x - (pbx - x) + 1); covx = t.overlap_x; }
                else if (x >= pex) { weight *= (float)(t.overlap_x - (x - pex)); covx = t.overlap_x; }
                if (y < pby) { weight *= (float)(t.overlap_y - (pby - y) + 1); covy = t.overlap_y; }
                else if (y >= pey) { weight *= (float)(t.overlap_y - (y - pey)); covy = t.overlap_y; }
                const float* tv = tiles + ((((long)b * t.n_y + cy) * t.n_x + cx) * t.tile_h + (y - by)) * (long)t.tile_w * 3 + (long)(x - bx) * 3;
                if (weight > 0.0f) {
                    const float blend = weight / (float)((covx + 1) * (covy + 1));
                    for (int c = 0; c < 3; ++c) d[c] = d[c] + blend * tv[c];
                } else {
                    for (int c = 0; c < 3; ++c) d[c] = tv[c];
                }
            }
        }
    }
    if (out_f32) {
        out_f32[i * 3 + 0] = d[0];
        out_f32[i * 3 + 1] = d[1];
        out_f32[i * 3 + 2] = d[2];
    }
    if (out_rgba) {
        uchar4 o;
        o.x = (uint8_t)(fminf(fmaxf(d[0], 0.0f), 1.0f) * 255.0f);
        o.y = (uint8_t)(fminf(fmaxf(d[1], 0.0f), 1.0f) * 255.0f);
        o.z = (uint8_t)(fminf(fmaxf(d[2], 0.0f), 1.0f) * 255.0f);
        o.w = 255;
        reinterpret_cast<uchar4*>(out_rgba)[i] = o;
    }
}

} // namespace

extern "C" int vx_dconv3x3_f16(const vx_dconv_args* args, void* stream) {
    const vx_dconv_args& a = *args;
    VX_REQUIRE(a.x && a.w && a.out, "vx_dconv3x3_f16: null operand");
    VX_REQUIRE(a.cin >= 32 && a.cin % 32 == 0 && a.x_ld >= a.cin && a.x_ld % 8 == 0, "vx_dconv3x3_f16: Cin %d / pixel stride %d must be multiples of 32 / 8", a.cin, a.x_ld);
    VX_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "vx_dconv3x3_f16: empty extent");
    VX_REQUIRE(!a.up2 || (a.H % 2 == 0 && a.W % 2 == 0), "vx_dconv3x3_f16: upsampled extent must be even");
    VX_REQUIRE((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(a.out) & 15) == 0, "vx_dconv3x3_f16: operands must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    if (a.epi == VX_DC_RGB_F32) {
        VX_REQUIRE(a.cout == 32, "vx_dconv3x3_f16: the rgb head takes weights padded to 32 outputs");
        return launch_dconv<32, 2, 32, VX_DC_RGB_F32>(a, s);
    }
    VX_REQUIRE(a.epi == VX_DC_F16, "vx_dconv3x3_f16: unknown epilogue %d", a.epi);
    VX_REQUIRE(a.ldo % 8 == 0 && (!a.res1 || a.res1_ld % 8 == 0) && (!a.res2 || a.res2_ld % 8 == 0) && (a.res1 || !a.res2),
               "vx_dconv3x3_f16: output/residual pixel strides must be multiples of 8 (res2 needs res1)");
    if (a.cout == 32) return launch_dconv<32, 2, 32, VX_DC_F16>(a, s);
    if (a.cout == 64) return launch_dconv<64, 2, 32, VX_DC_F16>(a, s);
    vx_set_error("vx_dconv3x3_f16: Cout %d not in {32, 64}", a.cout);
    return 0;
}

extern "C" int vx_esrgan_tiles_in(const uint8_t* img, int B, int w, int h, int format, const vx_tile_layout* t, void* out, void* stream) {
    int ch, ir, ig, ib;
    switch (format) { // visp::image_format (include/visp/image.h:17-29)
        case 0: ch = 4; ir = 0; ig = 1; ib = 2; break; // rgba_u8
        case 1: ch = 4; ir = 2; ig = 1; ib = 0; break; // bgra_u8
        case 2: ch = 4; ir = 1; ig = 2; ib = 3; break; // argb_u8
        case 3: ch = 3; ir = 0; ig = 1; ib = 2; break; // rgb_u8
        default: vx_set_error("vx_esrgan_tiles_in: unsupported image format %d", format); return 0;
    }
    const long n = (long)B * t->n_x * t->n_y * t->tile_h * t->tile_w;
    hipLaunchKernelGGL(esr_tiles_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), img, B, w, h, ch, ir, ig, ib, *t,
                       reinterpret_cast<f16*>(out));
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_esrgan_tiles_out(const float* tiles, int B, const vx_tile_layout* t, float* out_f32, uint8_t* out_rgba, void* stream) {
    const long n = (long)B * t->image_h * t->image_w;
    hipLaunchKernelGGL(esr_tiles_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), tiles, B, *t, out_f32, out_rgba);
    VX_LAUNCH_CHECK();
    return 1;
}
