// DPT head tail (depth-anything.cpp:84-94) as ONE kernel made for its shape: bilinear (align_corners) resize of the 32-channel
// map to the output extent, conv 3x3 32 -> 32 + ReLU, conv 1x1 32 -> 1 + ReLU, x max_depth -> f32 depth.
//
// Why not the LDS-ring conv (kernels_dconv.hip) that ran this before: that kernel is a persistent weight-streaming design for the
// ESRGAN dense blocks (Cin up to 160). At Cin = 32 its in-kernel stamps (profiles/r03_dconv_stamps_dpt.txt) show 1.35k cycles of
// MFMA loop in an 8.7k-cycle step -- the ring cursors, the per-piece DMA issue and the block-wide epilogue are fixed costs per tile
// that a 288-deep reduction cannot amortise. Here nothing streams:
//   * the whole 3x3x32x32 kernel lives in REGISTERS as MFMA A fragments (18 x 4 registers per lane), loaded once per block;
//   * a block = 4 waves, persistent, one 16 x 32 output tile at a time; its LDS holds two 12 x 21 source patches (the next tile's
//     arrives by LDS-DMA under this tile's work) and the interpolated 18 x 34 halo (76 KB), so TWO blocks share a CU and one block's
//     interpolation phase (VALU, LDS) runs under the other's MFMA phase -- no ring cursors, two barriers per tile. (A first,
//     non-persistent form -- one tile per block, everything loaded at block start -- was latency-bound at the old kernel's speed:
//     three dependent global round trips per tile.)
//   * pixels are the N dimension (D^T[cout, pixel] = W X^T): a wave owns 4 output rows x 32 pixels; every B fragment (one
//     ds_read_b128 per lane: 8 channels of one pixel) read from the halo feeds up to three MFMAs (the three output rows whose
//     window contains that halo row), 36 reads for 72 MFMAs = half the LDS bytes per MFMA of a one-read-per-MFMA loop, which is
//     exactly the LDS port's rate;
//   * with pixels in the lanes, conv3's reduction over channels is 16 in-lane FMAs and one v_permlane32_swap; a lane then owns one
//     output pixel and the 32 lanes of a row store 128 contiguous bytes.
// The interpolation is the arithmetic of kernels_dconv.hip's resizing loader and of vx_bilinear_ac_f16: src = i / sf with
// sf = (out - 1) / (in - 1), weights rounded to f16, v + f (w - v) evaluated as f w + (v - f v) on packed halves, x then y.
#include <algorithm>

#include "vx_common.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int TH = 16, TW = 32, HH = TH + 2, HW = TW + 2; // output tile, halo
constexpr int PR = 12, PC = 21;                           // source patch (rows, columns): covers the halo up to scale 0.58 (floor(17 s) + 3 rows, floor(33 s) + 3 columns)
constexpr int PIXB = 64;                                  // 32 channels f16
constexpr int HALO_BYTES = HH * HW * PIXB;                // 39168
constexpr int SRC_BYTES = PR * PC * PIXB + PIXB;          // + one zero pixel (what halo positions outside the map read)
constexpr int ZERO_OFF = PR * PC * PIXB;
constexpr int TAB_BYTES = (HH + HW) * 8;
constexpr int SMEM_BYTES = HALO_BYTES + 2 * SRC_BYTES + TAB_BYTES + 256; // 76.3 KB: two blocks per CU

// ReLU in ONE instruction the compiler can see: fmaxf on an MFMA result costs a canonicalising v_max first, and an inline-asm v_max_f32
// is invisible to the hazard recogniser -- placed right behind the accumulator's last MFMA it read the registers before the matrix pipe
// had written them (found by this kernel's first parity run: every wave's first row off by a few per cent). Signed-integer max with 0
// orders non-NaN floats the same way and maps -0.0 and every negative value to +0.
__device__ __forceinline__ float relu1(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ float other_half(float v) { // the same lane of the other 32-lane half
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

typedef __attribute__((address_space(3))) void* lptr_t;

struct tile_pos { int b, y0, x0; };

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void headconv_kernel(
    const f16* __restrict__ x, const f16* __restrict__ wfrag, const float* __restrict__ bias, const float* __restrict__ w3, float b3, float out_scale,
    float* __restrict__ out, int B, int H, int W, int Hs, int Ws, int tiles_x, int tiles_y, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;
    uint2* const tab_y = reinterpret_cast<uint2*>(smem + HALO_BYTES + 2 * SRC_BYTES);
    uint2* const tab_x = tab_y + HH;
    auto src_buf = [&](int k) { return smem + HALO_BYTES + k * SRC_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;

    // ---- persistent blocks, XCD-aware: blocks id, id + 8, ... share an XCD (and its L2); every XCD walks its own contiguous eighth of the
    // tile sequence (x fastest, then y, then image), its blocks side by side, so that neighbouring tiles -- whose source patches
    // overlap by half -- are fetched through one L2 at about the same time
    const int n_tiles = tiles_x * tiles_y * B;
    const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3, per_xcd_blocks = (gridDim.x + 7 - xcd) >> 3;
    const int per_xcd_tiles = (n_tiles + 7) >> 3;
    const int t_begin = xcd * per_xcd_tiles, t_end = min(t_begin + per_xcd_tiles, n_tiles);
    auto locate = [&](int t) {
        tile_pos p;
        const int per_image = tiles_x * tiles_y;
        p.b = t / per_image;
        const int k = t - p.b * per_image, ty = k / tiles_x;
        p.y0 = ty * TH;
        p.x0 = (k - ty * tiles_x) * TW;
        return p;
    };

    // ---- once per block: the 3x3 kernel as A fragments (lane (m = lane & 31 -> cout, h) holds W[m][tap][16 ks + 8 h .. + 7]), conv3's
    // weights and conv2's bias in accumulator order, the static part of this thread's interpolation items
    f16x8 wf[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) wf[i] = *reinterpret_cast<const f16x8*>(wfrag + ((size_t)i * 64 + lane) * 8);
    // conv2's bias and conv3's weights in accumulator order per lane half: LDS, read back per tile (32 registers the interpolation
    // phase needs)
    float* const w3_lds = reinterpret_cast<float*>(smem + HALO_BYTES + 2 * SRC_BYTES + TAB_BYTES);
    float* const bias_lds = w3_lds + 32;
    if (tid < 32) {
        const int c = (tid & 3) + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4);
        w3_lds[tid] = w3[c];
        bias_lds[tid] = bias[c];
    }
    constexpr int I_ITEMS = HH * HW * 4, I_IT = (I_ITEMS + 255) / 256; // (halo pixel, 8-channel chunk) items, 10 per thread
    unsigned ipack[I_IT]; // halo row | halo column << 8 | chunk << 16 (threads past the last item repeat an earlier one: identical
                          // bytes to the same place, no branch)
#pragma unroll
    for (int it = 0; it < I_IT; ++it) {
        int i = tid + it * 256;
        if (i >= I_ITEMS) i -= 256;
        const int pix = i >> 2, chunk = i & 3, hy = pix / HW, hx = pix - hy * HW;
        ipack[it] = (unsigned)hy | (unsigned)hx << 8 | (unsigned)chunk << 16;
    }
    unsigned boff[3][2]; // byte offset of the lane's B fragment in halo row 4 wave: pixel n + dx, channels 16 ks + 8 h
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = n + dx, chunk = ks * 2 + h;
            boff[dx][ks] = (unsigned)((4 * wave * HW + px) * PIXB + ((chunk ^ ((px >> 1) & 3)) * 16));
        }
    if (tid < 8) { // the zero pixel of both source buffers (what halo positions outside the map read)
        *reinterpret_cast<uint4*>(src_buf(tid >> 2) + ZERO_OFF + (tid & 3) * 16) = uint4{0, 0, 0, 0};
    }

    const float sfy = (H > 1 && Hs > 1) ? (float)(H - 1) / (float)(Hs - 1) : (float)H / (float)Hs;
    const float sfx = (W > 1 && Ws > 1) ? (float)(W - 1) / (float)(Ws - 1) : (float)W / (float)Ws;
    auto src_index = [](int o, float sf, int n_src, float& f) {
        const float sc = (float)o / sf;
        int i0 = (int)floorf(sc);
        i0 = max(0, min(i0, n_src - 1));
        f = fminf(fmaxf(sc - (float)i0, 0.0f), 1.0f);
        return i0;
    };
    auto patch_origin = [&](tile_pos const& p, int& py0, int& px0) {
        float dummy;
        py0 = src_index(max(p.y0 - 1, 0), sfy, Hs, dummy);
        px0 = src_index(max(p.x0 - 1, 0), sfx, Ws, dummy);
    };
    // source patch of a tile -> LDS by DMA (rows / columns past the map repeat its last one: never weighted, has-next is 0 there).
    // One descriptor over the whole tensor; a lane's 16 bytes land at its item's slot.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (int)((size_t)B * Hs * Ws * PIXB), 0x00020000);
    auto issue_patch = [&](tile_pos const& p, int k) {
        int py0, px0;
        patch_origin(p, py0, px0);
        constexpr int N_ITEMS = PR * PC * 4, N_IT = (N_ITEMS + 255) / 256;
        unsigned char* const dst = src_buf(k);
#pragma unroll
        for (int it = 0; it < N_IT; ++it) {
            const int i = tid + it * 256;
            const int pix = i >> 2, chunk = i & 3, pr = pix / PC, pc = pix - pr * PC;
            const int sy = min(py0 + pr, Hs - 1), sx = min(px0 + pc, Ws - 1);
            const unsigned off = (unsigned)(((p.b * Hs + sy) * Ws + sx) * PIXB + chunk * 16);
            if (i < N_ITEMS) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(dst + (it * 256 + wave * 64) * 16), 16, off, 0, 0, 0);
        }
    };

    // diagnostics (tools/headconv_stamps.py; NULL in the product): cycles per phase, summed over the block's tiles
    unsigned long long st[5] = {0, 0, 0, 0, 0}, c0 = 0;
    auto stamp = [&](int k) {
        if (stamps) {
            const unsigned long long c = __builtin_amdgcn_s_memtime();
            st[k] += c - c0;
            c0 = c;
        }
    };
    const unsigned long long t_start = stamps ? __builtin_amdgcn_s_memtime() : 0;
    int t = t_begin + in_xcd;
    if (t < t_end) issue_patch(locate(t), 0);
    int cur = 0, n_done = 0;
    if (stamps) c0 = __builtin_amdgcn_s_memtime();
    for (; t < t_end; t += per_xcd_blocks, cur ^= 1, ++n_done) {
        const tile_pos p = locate(t);
        int py0, px0;
        patch_origin(p, py0, px0);
        // ---- this tile's source coordinates of the halo rows / columns: .x = patch-relative index | has-next << 15 | outside << 31,
        // .y = weight f as packed f16 (f, f). (Every wave is past the previous tile's MFMA phase only after the barrier below; the
        // tables are read after it, and the previous tile's readers of them finished before its second barrier.)
        if (tid < HH + HW) {
            const bool isx = tid >= HH;
            const int o = (isx ? p.x0 + (tid - HH) : p.y0 + tid) - 1, n_out = isx ? W : H, n_src = isx ? Ws : Hs;
            uint2 e = {0x80000000u, 0u};
            if (o >= 0 && o < n_out) {
                float f;
                const int i0 = src_index(o, isx ? sfx : sfy, n_src, f);
                const unsigned short fh = __builtin_bit_cast(unsigned short, (f16)f);
                e.x = (unsigned)(i0 - (isx ? px0 : py0)) | (i0 + 1 < n_src ? 0x8000u : 0u);
                e.y = (unsigned)fh | (unsigned)fh << 16;
            }
            (isx ? tab_x[tid - HH] : tab_y[tid]) = e;
        }
        stamp(0); // tables
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this tile's patch (issued one tile ago) and the previous tile's stores
        __syncthreads();                                  // ... from every wave; and every wave is done reading the halo
        stamp(1); // wait + barrier

        // ---- interpolate the halo: chunk c of pixel column hx is stored at c ^ ((hx >> 1) & 3) so that the 8 lanes a ds_read_b128
        // serves per cycle (8 consecutive pixels, 64 bytes apart) hit 8 different 16-byte bank groups. The phase is LDS-latency bound
        // when an item waits for its own reads (first form: 7.7k of a tile's 13k cycles), so it is a software pipeline: all table
        // entries first, then batches of two items whose 8 source reads are issued one batch ahead of the arithmetic (the
        // accumulators are not live here: the registers are there).
        {
            const unsigned char* const src = src_buf(cur);
            auto lerp2 = [](h2 v, h2 w, h2 f) { return f * w + (v - f * v); };
            uint2 ty[I_IT], tx[I_IT];
#pragma unroll
            for (int k = 0; k < I_IT; ++k) {
                ty[k] = tab_y[ipack[k] & 0xff];
                tx[k] = tab_x[(ipack[k] >> 8) & 0xff];
            }
            f16x8 rd[2][2][4];
            auto load_batch = [&](int bt, f16x8 (&r)[2][4]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 2 * bt + u;
                    const unsigned chunk16 = ((ipack[k] >> 16) & 3) * 16;
                    const bool outside = (int)(ty[k].x | tx[k].x) < 0;
                    const unsigned sy = ty[k].x & 0x7fffu, sx = tx[k].x & 0x7fffu;
                    const unsigned o00 = (outside ? (unsigned)ZERO_OFF : (sy * PC + sx) * PIXB) + chunk16;
                    const unsigned o01 = o00 + (outside ? 0u : ((tx[k].x >> 15) & 1u) * PIXB);
                    const unsigned dy = outside ? 0u : ((ty[k].x >> 15) & 1u) * (PC * PIXB);
                    r[u][0] = *reinterpret_cast<const f16x8*>(src + o00);
                    r[u][1] = *reinterpret_cast<const f16x8*>(src + o01);
                    r[u][2] = *reinterpret_cast<const f16x8*>(src + o00 + dy);
                    r[u][3] = *reinterpret_cast<const f16x8*>(src + o01 + dy);
                }
            };
            load_batch(0, rd[0]);
#pragma unroll
            for (int bt = 0; bt < I_IT / 2; ++bt) {
                if (bt + 1 < I_IT / 2) load_batch(bt + 1, rd[(bt + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0); // the next batch's reads are in flight before this batch's arithmetic starts
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 2 * bt + u;
                    f16x8 const(&r)[4] = rd[bt & 1][u];
                    const h2 fx = __builtin_bit_cast(h2, tx[k].y), fy = __builtin_bit_cast(h2, ty[k].y);
                    f16x8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const h2 aa = {r[0][2 * q], r[0][2 * q + 1]}, bb = {r[1][2 * q], r[1][2 * q + 1]}, cc = {r[2][2 * q], r[2][2 * q + 1]}, dd = {r[3][2 * q], r[3][2 * q + 1]};
                        const h2 v = lerp2(lerp2(aa, bb, fx), lerp2(cc, dd, fx), fy);
                        o[2 * q] = v[0];
                        o[2 * q + 1] = v[1];
                    }
                    const unsigned pk = ipack[k], hy = pk & 0xff, hx = (pk >> 8) & 0xff, chunk = pk >> 16;
                    *reinterpret_cast<f16x8*>(halo + (hy * HW + hx) * PIXB + ((chunk ^ ((hx >> 1) & 3)) * 16)) = o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(2); // DMA issue + interpolation
        __syncthreads();
        stamp(1);

        // ---- conv2 on the matrix pipe: wave owns output rows 4 wave .. + 3; acc[j] = D^T[cout, pixel] of row j, starting at the bias
        // the next tile's source patch: issued here, where the wave's instruction issue is mostly idle (72 MFMAs hold it for 576 of the
        // phase's cycles); it lands under this phase. (Its buffer's last readers were the previous tile's interpolation.)
        if (t + per_xcd_blocks < t_end) issue_patch(locate(t + per_xcd_blocks), cur ^ 1);
        f32x16 acc[4], bias_tile; // the first MFMA of every accumulator takes the bias tile as its C operand: no 64 moves per tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bias_lds + 16 * h + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) bias_tile[4 * g + q] = v[q];
        }
        // B fragments one halo row ahead of their MFMAs (6 reads per row: 3 columns x 2 channel halves): a read consumed right
        // behind its issue exposes the LDS latency -- under the other block's interpolation traffic -- once per MFMA group
        f16x8 bf[2][6];
        auto load_row = [&](int hr, f16x8 (&r)[6]) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) r[dx * 2 + ks] = *reinterpret_cast<const f16x8*>(halo + boff[dx][ks] + hr * (HW * PIXB));
        };
        load_row(0, bf[0]);
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            if (hr + 1 < 6) load_row(hr + 1, bf[(hr + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int j = hr - dy; // output row whose window has halo row hr at tap row dy
                        const bool first = dy == 0 && dx == 0 && ks == 0; // (halo row hr = j: the accumulator's first product)
                        if (j >= 0 && j < 4)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 2 + ks], bf[hr & 1][dx * 2 + ks], first ? bias_tile : acc[j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- ReLU, conv3 (1x1 to one channel: 16 in-lane terms + the other lane half), ReLU, scale
        const int ox = p.x0 + n;
        float w3v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(w3_lds + 16 * h + 4 * g);
            w3v[4 * g] = v[0]; w3v[4 * g + 1] = v[1]; w3v[4 * g + 2] = v[2]; w3v[4 * g + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s = fmaf(relu1(acc[j][e]), w3v[e], s);
            s += other_half(s);
            const int oy = p.y0 + 4 * wave + j;
            if (h == 0 && ox < W && oy < H) out[((size_t)p.b * H + oy) * W + ox] = out_scale * fmaxf(s + b3, 0.0f);
        }
        stamp(3); // MFMA + epilogue
    }
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = (unsigned long long)n_done; o[5] = __builtin_amdgcn_s_memtime() - t_start;
    }
}

} // namespace

// conv kernel rows [32][Kp] f16 with k = (ky, kx, c) (the GEMM family's operand) -> the A fragments above: [tap * 2 + ks][lane][8]
extern "C" int vx_headconv_pack(const void* w_rows, int Kp, void* out_frag) {
    VX_REQUIRE(w_rows && out_frag && Kp >= 288, "vx_headconv_pack: rows of at least 288 elements");
    const uint16_t* w = static_cast<const uint16_t*>(w_rows);
    uint16_t* o = static_cast<uint16_t*>(out_frag);
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 8; ++q) o[(((size_t)tap * 2 + ks) * 64 + lane) * 8 + q] = w[(size_t)(lane & 31) * Kp + tap * 32 + ks * 16 + 8 * (lane >> 5) + q];
    return 1;
}
extern "C" size_t vx_headconv_frag_bytes(void) { return (size_t)18 * 64 * 8 * 2; }

extern "C" int vx_headconv_supported(int cin, int cout, int H, int W, int hs, int ws) {
    if (cin != 32 || cout != 32 || H < 2 || W < 2 || hs < 2 || ws < 2) return 0;
    const double sy = (double)(hs - 1) / (H - 1), sx = (double)(ws - 1) / (W - 1);
    // halo rows y0-1 .. y0+16 span 17 steps, columns 33: floor(first) .. floor(last) + 1 must fit the patch
    return (int)(17 * sy) + 3 <= PR && (int)(33 * sx) + 3 <= PC;
}

// diagnostics: u64 [blocks][4 waves][8] = cycles in {tables, wait + barriers, DMA issue + interpolation, MFMA + epilogue}, tiles, lifetime
static void* g_headconv_stamps = nullptr;
extern "C" void vx_headconv_set_stamps(void* stamps) { g_headconv_stamps = stamps; }

extern "C" int vx_headconv_bil_f16(const void* x, const void* wfrag, const float* bias, const float* w3, float b3, float scale, float* out, int B, int H, int W,
                                   int hs, int ws, void* stream) {
    VX_REQUIRE(x && wfrag && bias && w3 && out && B > 0, "vx_headconv_bil_f16: null operand");
    VX_REQUIRE(vx_headconv_supported(32, 32, H, W, hs, ws), "vx_headconv_bil_f16: %dx%d from %dx%d is outside the kernel's source patch (scale up to 0.58)", W, H, ws, hs);
    VX_REQUIRE((size_t)B * hs * ws * PIXB < (size_t)1 << 31, "vx_headconv_bil_f16: the source tensor must be below 2 GB (32-bit buffer offsets)");
    VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(headconv_kernel), SMEM_BYTES));
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long n_tiles = (long)tiles_x * tiles_y * B;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    static const int bpc = getenv("VISP_HEADCONV_BPC") ? atoi(getenv("VISP_HEADCONV_BPC")) : 2; // experiments
    const int blocks = (int)std::min<long>(n_tiles, (long)std::max(1, bpc) * n_cu); // two persistent blocks per CU
    hipLaunchKernelGGL(headconv_kernel, dim3(blocks), dim3(256), SMEM_BYTES, as_stream(stream), reinterpret_cast<const f16*>(x), reinterpret_cast<const f16*>(wfrag), bias,
                       w3, b3, scale, out, B, H, W, hs, ws, tiles_x, tiles_y, static_cast<unsigned long long*>(g_headconv_stamps));
    VX_LAUNCH_CHECK();
    return 1;
}
