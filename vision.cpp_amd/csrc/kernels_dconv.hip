// 3x3 / stride 1 / pad 1 convolution with 32 or 64 output channels on gfx950: the conv engine of the ESRGAN /
// Real-ESRGAN RRDB stack (reference src/visp/arch/esrgan.cpp:13-79: conv_block = conv_2d 3x3 + leaky_relu 0.2, dense
// concat, x5*0.2 + x, nearest x2 upsample before the up-convs) and of the large DPT maps of Depth-Anything
// (src/visp/arch/depth-anything.cpp:15-30, 81-94: residual units on relu(x), head convs).
//
//  * Layouts. Activations are f16 in groups of 32 channels; a map is addressed by (pixel stride, group stride):
//    PLANAR [C/32 planes][pixels][32] for ESRGAN (pixel stride 32, group stride = plane) or NHWC (pixel stride C, group
//    stride 32) for the DPT maps. In planar form a residual dense block keeps [x | x1 | x2 | x3 | x4] as six planes of
//    one buffer, every conv reads the first 2+k planes and writes its 32 outputs as the next plane, so the
//    reference's four concat copies never happen -- and a chunk's halo row is one contiguous 2 KB run in HBM.
//  * Persistent blocks (one 8-wave block per CU, its ring fills the LDS), XCD-aware contiguous tile runs. A block's
//    (tile, 32-channel chunk) steps form ONE stream: the next tile's first chunk lands under the current tile's last
//    chunk and epilogue. A step's 18x34-pixel halo (39 KB) and 9 x COUT x 32 weight slab (18/36 KB, pre-swizzled at
//    load time so the copy is linear) arrive by `buffer_load ... lds` (descriptor in SGPRs, 32-bit lane offsets,
//    out-of-map pixels zero-filled by the range check = conv padding). Halo ring: 3 stages for COUT = 32 (issued two
//    steps ahead), 2 for COUT = 64; slabs: 2 stages. ONE raw s_barrier per step; counted vmcnt waits leave younger
//    halos and the tile's output stores in flight.
//  * Tiles are 16x32 pixels; a remainder strip of 1..16 columns is tiled 32x16 (same pixel count, same halo size).
//  * MFMA 32x32x16 f16 with swapped operands (D = W-fragment x pixel-fragment: a lane owns a pixel and four
//    consecutive channels per register group). COUT = 32: per (tap column, k-step) 3 weight fragments and 4 pixel
//    WINDOWS feed 6 MFMAs (M-tile mi and tap row ky read window mi + ky); COUT = 64: per-tap loop with 2x2 register
//    blocking and the DMA pieces issued from inside the loop. Which form where was decided by same-device A/B runs.
//  * Epilogues: bias, LeakyReLU 0.2 / ReLU, v*s1 + res1, v*s2 + res2 through a staged f16 tile and 16-byte stores;
//    "+ x" of a dense block's conv5 without re-reading x (identity fold, see x_residual); nearest x2 upsampling is a
//    shift in the halo source address; the ESRGAN RGB head and the fused DPT depth head (conv2 + ReLU + 1x1 conv3 +
//    ReLU) write f32 straight from the accumulators; AR = ReLU on the input fragments.
#include "vx_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// out-of-map lanes of an edge tile store here (one KiB per wave) instead of being masked off, so every wave issues
// the same number of store instructions per tile and the ring can wait with a COUNTED vmcnt that leaves the
// stores in flight
constexpr int TRASH_BLOCKS = 1024;
__device__ __attribute__((aligned(16))) unsigned char g_dconv_trash[TRASH_BLOCKS * 8 * 1024];

constexpr int DCONV_TAB_MAX = 1280; // BIL: H + W entries (8 bytes each) of the row / column interpolation table
constexpr int CK = 32;      // channels per chunk
constexpr int PIXB = CK * 2; // bytes per pixel per chunk
constexpr int NW = 8;       // waves per block

// tile enumeration of one H x W map: 16x32 tiles over the columns that fill whole 32-wide tiles, and -- when the
// remainder is 1..16 columns -- a strip of 32x16 tiles for it (same 512 pixels and the same 612-pixel halo,
// transposed), so a 144-wide ESRGAN tile costs 41 blocks instead of 45.
struct tile_grid {
    int ncols, nrows, n_main, n_strip, strip_x0;
    __host__ __device__ tile_grid(int H, int W) {
        const int q = W / 32, rem = W % 32;
        const bool strip = rem > 0 && rem <= 16;
        ncols = q + ((rem > 16) ? 1 : 0);
        nrows = (H + 15) / 16;
        n_main = ncols * nrows;
        n_strip = strip ? (H + 31) / 32 : 0;
        strip_x0 = q * 32;
    }
    __host__ __device__ int total() const { return n_main + n_strip; }
};

// STAMP: diagnostics build (args.stamps != NULL) that sums shader-clock cycles per phase; the product instantiation
// carries none of it.
// AR: ReLU applied to the input while its fragments are read (the DPT residual units convolve relu(x),
// depth-anything.cpp:15-23).
// RES: number of weight slabs kept RESIDENT in LDS for the whole launch (all cin/32 chunks of the conv fit next to the
// halo ring), 0 = slabs stream through a 2-stage ring. HSV: halo ring stages of this variant.
// max(x, 0) as ONE instruction: fmaxf() on an MFMA result costs a canonicalising v_max_f32 x, x, x first (IEEE mode), and the depth
// head's epilogue is 32 of them per tile in a kernel bound by VALU issue. Signed-integer max with 0 (same order as floats for non-NaN
// values, -0.0 and negatives -> +0), NOT an inline-asm v_max_f32: the hazard recogniser does not look into inline asm, and one placed
// right behind an accumulator's last MFMA reads it before the matrix pipe has written it (kernels_headconv.hip met exactly that).
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// BIL: the input is a LOW-resolution map and the conv runs on its bilinear (align_corners) resize (the DPT head's two
// `interpolate` calls, depth-anything.cpp:36-38 / 84-85 -> ml.cpp:782-788 ggml_interpolate): the resized map never exists. Per step
// the loader DMAs the SOURCE patch under the halo (<= 12 x 21 pixels of the chunk's 32 channels, 16 KB) two steps ahead into a
// 2-stage source ring, and the waves interpolate the NEXT step's 18 x 34 halo out of it into the other halo stage from inside the
// current step's MFMA loop (4 ds_read_b128 + 16 v_pk_fma_f16 + 1 ds_write_b128 per pixel and 8 channels, five of those per lane
// and step). Source row / column and weight of every output row / column come from a table built in LDS at kernel start.
// ONE: Cin = 32, every tile is a single step: its accumulators need no initialisation, the first MFMA of each takes the bias tile as C.
template <int COUT, int EPI, bool AR, bool STAMP, int RES = 0, int HSV = (COUT == 32 ? 3 : 2), bool BIL = false, bool ONE = false>
__global__ __launch_bounds__(512) void dconv3x3_kernel(const vx_dconv_args p) {
    static_assert(!ONE || COUT == 32, "single-step tiles: COUT = 32 only");
    constexpr int MT = 2;                       // M-tiles (32 pixels) per wave
    constexpr int HALO_PIX = 18 * 34;           // both tile shapes
    // a halo stage holds [pixel][64 B] (the chunk's 32 channels) with the four 16-byte groups XOR-swizzled by
    // (pixel >> 2) & 3: 16 consecutive pixels cover all 64 banks, and a pixel is ONE 64-byte global segment for the
    // LDS-DMA (two 32-byte k-step planes would need one address register set less but double the number of
    // cache-line requests per byte: measured 15-25 % slower). 612 pixels = 38.25 KiB, padded to 40 wave
    // instructions so that every wave issues exactly HJ = 5 of them (the counted vmcnt below relies on it).
    constexpr int HJ = 5, HALO_INSTR = HJ * NW, HALO_BYTES = HALO_INSTR * 1024;
    constexpr int NI = COUT / 32;
    constexpr int W_BYTES = 9 * COUT * PIXB, W_INSTR = W_BYTES / 1024;
    constexpr int WJ = (W_INSTR + NW - 1) / NW;
    // The kernel is bound by the LDS-DMA stream (57/77 KB per 36/72 MFMAs per wave), i.e. by bytes in flight per
    // CU: with COUT = 32 the LDS has room for a THIRD halo stage, so halos are issued two steps ahead.
    constexpr int HS = HSV;                     // halo stages
    constexpr int WSLOTS = RES ? RES : 2;       // slab slots: the streaming ring has 2
    static_assert(!RES || COUT == 32, "resident slabs are built for COUT = 32 (the COUT = 64 epilogue stages through slab space)");
    static_assert(!BIL || (COUT == 32 && HSV == 2 && !RES && !AR), "the interpolating loader is built for COUT = 32 with two halo stages");
    constexpr int SRC_PIX = 256, SRC_BYTES = SRC_PIX * PIXB, SJ = SRC_BYTES / 1024 / NW; // source patch: 16 KB = 2 DMA instructions per wave
    constexpr int SRC_BASE = HS * HALO_BYTES;   // BIL: [halo 0 | halo 1 | src 0 | src 1 | slab 0 | slab 1 | bias | row/col table]
    constexpr int W_BASE = HS * HALO_BYTES + (BIL ? 2 * SRC_BYTES : 0); // LDS: [halo 0 .. HS-1 | slab slots | bias]
    constexpr int BIAS_BASE = W_BASE + WSLOTS * W_BYTES;
    constexpr int TAB_BASE = BIAS_BASE + ((COUT * 4 + 15) & ~15); // BIL: int [H + W], (source index << 16) | f16 weight
    static_assert(!BIL || TAB_BASE + (DCONV_TAB_MAX + 4) * 8 <= 160 * 1024, "interpolation tables do not fit");
    constexpr int NCH16 = COUT / 8, PITCH = COUT * 2;
    static_assert(BIAS_BASE + COUT * 4 <= 160 * 1024, "LDS ring too large");
    static_assert(256 * PITCH <= HALO_BYTES && 256 * PITCH <= W_BYTES + (COUT == 32 ? HALO_BYTES : 0), "output staging does not fit a stage");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const s_bias = reinterpret_cast<float*>(smem + BIAS_BASE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const tile_grid tg(H, W);
    const int per_image = tg.total();
    const int total = p.B * per_image;

    // persistent blocks; XCD-aware order: blocks id and id+8 share an L2. Every XCD owns a contiguous run of the
    // tile sequence (neighbouring tiles share halo rows, one image's tiles share their slabs) and its blocks walk
    // that run with a stride of the XCD's block count.
    int t_cur, t_end, t_step;
    {
        const int id = blockIdx.x, xcd = id & 7;
        const int per = total >> 3, rem = total & 7;
        const int lo = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per;
        t_end = lo + per + (xcd < rem ? 1 : 0);
        t_step = ((int)gridDim.x - xcd + 7) >> 3;
        t_cur = lo + (id >> 3);
    }
    if (t_cur >= t_end) return;

    if (tid < COUT) s_bias[tid] = p.bias ? p.bias[tid] : 0.0f;

    const int up = p.up2 ? 1 : 0;
    const int Hs = BIL ? p.bil_hs : H >> up, Ws = BIL ? p.bil_ws : W >> up;
    const int nch = p.cin / CK;
    const int nch_total_bytes = nch * 9 * COUT * PIXB;
    const long x_plane_bytes = p.x_plane * 2;
    const int x_pix = p.x_pix ? (int)p.x_pix : CK;           // elements between pixels (32 = planar, C = NHWC)
    const long out_pix = p.out_pix ? p.out_pix : CK, res1_pix = p.res1_pix ? p.res1_pix : CK, res2_pix = p.res2_pix ? p.res2_pix : CK;

    // BIL: row / column tables of 8-byte entries. Entry i + 1 describes output row (column) i; entries 0 and H + 1 (W + 1) stand for
    // the rows of the conv's zero padding just outside the map, so a halo pixel needs no range test:
    //   .x: bit 31 outside the map | bit 30 the source has a next row (column) | bits 29:16 first source index
    //   .y: packed f16 weights (1 - f) | f << 16 of the two source rows (columns); both 0 outside the map, so that the four
    //       products of a pixel's row and column pairs are its bilinear weights and vanish in the padding
    // Source coordinate = out / sf with sf = (out_extent - 1) / (in_extent - 1), as ggml computes it (and bilinear_ac_kernel).
    uint2* const tab_y = reinterpret_cast<uint2*>(smem + TAB_BASE);
    uint2* const tab_x = tab_y + H + 2;
    if constexpr (BIL) {
        const float sfy = (H > 1 && Hs > 1) ? (float)(H - 1) / (float)(Hs - 1) : (float)H / (float)Hs;
        const float sfx = (W > 1 && Ws > 1) ? (float)(W - 1) / (float)(Ws - 1) : (float)W / (float)Ws;
        for (int i = tid; i < H + W + 4; i += 512) {
            const bool isx = i >= H + 2;
            const int o = (isx ? i - (H + 2) : i) - 1, n_out = isx ? W : H, n_src = isx ? Ws : Hs;
            uint2 e = {0x80000000u, 0u};
            if (o >= 0 && o < n_out) {
                const float sc = (float)o / (isx ? sfx : sfy);
                int i0 = (int)floorf(sc);
                i0 = max(0, min(i0, n_src - 1));
                const float f = fminf(fmaxf(sc - (float)i0, 0.0f), 1.0f);
                e.x = (i0 + 1 < n_src ? 0x40000000u : 0u) | (unsigned)i0 << 16;
                e.y = (unsigned)__builtin_bit_cast(unsigned short, (f16)(1.0f - f)) | (unsigned)__builtin_bit_cast(unsigned short, (f16)f) << 16;
            }
            tab_y[i] = e;
        }
        __syncthreads(); // (no LDS-DMA in flight yet)
    }

    // ---- tile geometry (wave-uniform) and per-lane halo sources
    struct geom { int b, y0, x0, tws; }; // tws = log2(tile width): 5 (16x32) or 4 (32x16)
    auto locate = [&](int t) {
        geom g;
        g.b = t / per_image;
        const int k = t - g.b * per_image;
        if (k < tg.n_main) {
            const int ty = k / tg.ncols;
            g.y0 = ty * 16; g.x0 = (k - ty * tg.ncols) * 32; g.tws = 5;
        } else {
            g.y0 = (k - tg.n_main) * 32; g.x0 = tg.strip_x0; g.tws = 4;
        }
        return g;
    };
    // Both streams go through buffer_load ... lds: the descriptor (base, size) sits in SGPRs, a lane contributes a
    // 32-bit byte offset and the chunk a scalar offset, so no 64-bit per-lane pointers stay live across the tile
    // loop; halo pixels outside the map use an offset beyond the descriptor's size and are zero-filled by the
    // hardware's range check (conv zero padding, nn.cpp:83-97 pad = 1).
    constexpr unsigned OOB = 0x80000000u;
    unsigned hoff[HJ];  // byte offset of the lane's 16 bytes (chunk 0) inside the image the halo cursor is in
    unsigned hpack[HJ]; // shape-dependent, tile-invariant part: halo row | halo col << 8 | swizzled group << 16 | valid << 24
    auto setup_shape_src = [&](int tws) {
        const int hw = (1 << tws) + 2;
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int L = (wave + j * NW) * 64 + lane;
            const int pix = L >> 2, phys = L & 3;
            const int hy = tws == 5 ? pix / 34 : pix / 18;
            const int hx = pix - hy * hw;
            // (a slot past the halo's 612 pixels: not valid, and row 255 for the interpolating loader's table lookup)
            hpack[j] = (unsigned)(pix < HALO_PIX ? hy : 255) | (unsigned)hx << 8 | (unsigned)(phys ^ ((pix >> 2) & 3)) << 16 | (pix < HALO_PIX ? 1u << 24 : 0u);
        }
    };
    __amdgpu_buffer_rsrc_t x_rsrc;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, nch_total_bytes, 0x00020000);
    auto setup_src = [&](const geom& g) {
        const long img_bytes = (long)Hs * Ws * x_pix * 2;
        // planar: the descriptor spans image b of plane 0 .. image b of the last plane, chunk c is reached by a scalar
        // offset. NHWC (x_pix > 32): chunks sit inside the pixel, the descriptor ends with the image -- channels past
        // the map's last pixel (a Cin padded up to 32, e.g. 48 -> 64 with zero weights) are zero-filled, not read.
        const long span = x_pix > CK ? img_bytes : (nch - 1) * x_plane_bytes + img_bytes;
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + g.b * img_bytes, 0, (int)span, 0x00020000);
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            unsigned pk = hpack[j];
            asm volatile("" : "+v"(pk)); // keep the unpacked fields out of registers across the tile loop
            const int iy = g.y0 - 1 + (int)(pk & 0xff), ix = g.x0 - 1 + (int)((pk >> 8) & 0xff);
            const bool valid = (pk >> 24) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int e = ((iy >> up) * Ws + (ix >> up)) * x_pix + (int)(((pk >> 16) & 3) << 3);
            hoff[j] = valid ? (unsigned)(e * 2) : OOB;
        }
    };
    // ---- BIL: the source patch under a tile's halo. Patch pixel q = (lane-linear 16-byte slot) >> 2 is (q / PW, q % PW) of a
    // 12 x 21 (16x32 tiles) or 21 x 12 (32x16 tiles) window whose origin is the source pixel of the halo's first in-map row / column.
    unsigned spack[SJ]; // tile-shape part: patch row | patch col << 8 | 16-byte group << 16 | valid << 24
    unsigned soff[SJ];  // byte offset of the lane's 16 bytes (chunk 0) inside the source image of the halo cursor's tile
    auto patch_w = [](int tws) { return tws == 5 ? 21 : 12; };
    auto setup_shape_patch = [&](int tws) {
        const int pw = patch_w(tws);
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int L = (wave + j * NW) * 64 + lane, q = L >> 2;
            const int prow = q / pw;
            spack[j] = (unsigned)prow | (unsigned)(q - prow * pw) << 8 | (unsigned)(L & 3) << 16 | (q < 252 ? 1u << 24 : 0u);
        }
    };
    // origin of the patch of the tile at (y0, x0): wave-uniform table reads
    auto patch_origin = [&](int y0, int x0, int& py0, int& px0) {
        py0 = (__builtin_amdgcn_readfirstlane(tab_y[max(y0 - 1, 0) + 1].x) >> 16) & 0x3fff;
        px0 = (__builtin_amdgcn_readfirstlane(tab_x[max(x0 - 1, 0) + 1].x) >> 16) & 0x3fff;
    };
    auto setup_patch = [&](const geom& g) {
        const long img_bytes = (long)Hs * Ws * x_pix * 2;
        const long span = x_pix > CK ? img_bytes : (nch - 1) * x_plane_bytes + img_bytes;
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + g.b * img_bytes, 0, (int)span, 0x00020000);
        int py0, px0;
        patch_origin(g.y0, g.x0, py0, px0);
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            unsigned pk = spack[j];
            asm volatile("" : "+v"(pk));
            const int sy = py0 + (int)(pk & 0xff), sx = px0 + (int)((pk >> 8) & 0xff);
            const bool valid = (pk >> 24) && sy < Hs && sx < Ws;
            soff[j] = valid ? (unsigned)(((sy * Ws + sx) * x_pix + (int)(((pk >> 16) & 3) << 3)) * 2) : OOB;
        }
    };
    auto issue_patch = [&](int c, int sstage) {
#pragma unroll
        for (int j = 0; j < SJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(smem + SRC_BASE + sstage * SRC_BYTES + (wave + j * NW) * 1024), 16, soff[j],
                                                     c * (int)x_plane_bytes, 0, 0);
    };
    // ---- BIL: interpolation of one tile's halo. Unit j of a lane is halo slot L = (wave + 8 j) * 64 + lane: pixel L >> 2, physical
    // 16-byte group L & 3 (which holds channel group (L & 3) ^ ((pixel >> 2) & 3), as the DMA form stores it); its halo row / column
    // and channel group are the tile-shape constants of hpack[] (setup_shape_src). Per tile and unit, branch-free:
    //   ioff: byte offsets inside the patch of the top-left source pixel's group (bits 15:0) and of its right neighbour (31:16)
    //   idy : byte distance to the row below (0 on the source's last row)
    //   ifr : f16 column weight fx | f16 row weight fy << 16
    // A pixel of the conv's zero padding (or a slot past the halo) reads the patch's last four pixels, 252 .. 255, which no patch
    // uses and the DMA's range check zero-fills: the interpolation of zeros is the padding.
    unsigned ioff[HJ], ifr[HJ];
    unsigned short idy[HJ];
    int i_shape = -1; // tile shape hpack[] was set up for
    auto setup_interp = [&](const geom& g) {
        if (g.tws != i_shape) {
            i_shape = g.tws;
            setup_shape_src(i_shape);
        }
        int py0, px0;
        patch_origin(g.y0, g.x0, py0, px0);
        const int pw = patch_w(g.tws), org = py0 * pw + px0;
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            unsigned pk = hpack[j];
            asm volatile("" : "+v"(pk));
            // (an edge tile's halo runs past the map: every row / column beyond it is the "outside" entry H + 1 / W + 1; so is the
            // row of a slot past the halo's 612 pixels, whose hpack row is 255)
            const uint2 ty = tab_y[min(g.y0 + (int)(pk & 0xff), H + 1)], tx = tab_x[min(g.x0 + (int)((pk >> 8) & 0xff), W + 1)];
            const bool outside = (int)(ty.x | tx.x) < 0;
            const int sy = (ty.x >> 16) & 0x3fff, sx = (tx.x >> 16) & 0x3fff;
            const unsigned lg16 = ((pk >> 16) & 3) * 16;
            const unsigned o00 = (outside ? 252u * PIXB : (unsigned)((sy * pw + sx - org) * PIXB)) + lg16;
            ioff[j] = o00 | (o00 + (outside ? 0u : ((tx.x >> 30) & 1) * PIXB)) << 16;
            idy[j] = (unsigned short)(outside ? 0u : ((ty.x >> 30) & 1) * (unsigned)(pw * PIXB));
            ifr[j] = (tx.y >> 16) | (ty.y & 0xffff0000u);
        }
    };
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    // v + f * (w - v) as v - f v + f w: two fused multiply-adds whose weights add up to one EXACTLY whatever the rounding of f (a
    // weighted sum of four rounded f16 products does not: its weights sum to 1 +- 5e-4, a smooth bias on the resized map)
    auto lerp2 = [](h2 v, h2 w, h2 f) { return f * w + (v - f * v); };
    auto interp_unit = [&](int j, int sstage, int hstage) {
        const unsigned char* const sp = smem + SRC_BASE + sstage * SRC_BYTES;
        const unsigned o0 = ioff[j] & 0xffffu, o1 = ioff[j] >> 16, dy = idy[j];
        const f16x8 a = *reinterpret_cast<const f16x8*>(sp + o0), b = *reinterpret_cast<const f16x8*>(sp + o1);
        const f16x8 c = *reinterpret_cast<const f16x8*>(sp + o0 + dy), d = *reinterpret_cast<const f16x8*>(sp + o1 + dy);
        const h2 fr = __builtin_bit_cast(h2, ifr[j]);
        const h2 fx = __builtin_shufflevector(fr, fr, 0, 0), fy = __builtin_shufflevector(fr, fr, 1, 1);
        f16x8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const h2 aa = {a[2 * q], a[2 * q + 1]}, bb = {b[2 * q], b[2 * q + 1]}, cc = {c[2 * q], c[2 * q + 1]}, dd = {d[2 * q], d[2 * q + 1]};
            const h2 r = lerp2(lerp2(aa, bb, fx), lerp2(cc, dd, fx), fy);
            o[2 * q] = r[0];
            o[2 * q + 1] = r[1];
        }
        *reinterpret_cast<f16x8*>(smem + hstage * HALO_BYTES + ((wave + j * NW) * 64 + lane) * 16) = o;
    };

    // One LDS-DMA instruction costs its wave 60-180 issue cycles (measured: 8 of them in a row ~1050 cycles per step
    // with the MFMA pipe idle), so the ring is fed piece by piece from inside the MFMA loop: piece k of a step is
    // issued after the k-th fragment group's MFMAs. Slab pieces come first, then halo pieces (the counted vmcnt of
    // the next step leaves exactly the younger halo pieces and the tile's stores in flight).
    auto halo_piece = [&](int j, int c, int hstage) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(smem + hstage * HALO_BYTES + (wave + j * NW) * 1024), 16, hoff[j], c * (int)x_plane_bytes, 0, 0);
    };
    // every block streams the SAME slab at about the same time; each block starts at its own 1 KiB piece and wraps
    // around so that the CUs of an XCD do not walk the L2 channels in step
    const int slab_rot = (int)((blockIdx.x * 5u + (blockIdx.x >> 3) * 3u) % (unsigned)W_INSTR);
    auto slab_piece = [&](int j, int c, int wslot) {
        const int i = wave + j * NW;
        if (i < W_INSTR) {
            int ir = i;
            if constexpr (COUT == 64) { // (COUT = 32: constant offsets, the rotation bought nothing measurable there)
                ir += slab_rot;
                ir = ir >= W_INSTR ? ir - W_INSTR : ir;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(smem + W_BASE + wslot * W_BYTES + ir * 1024), 16, lane * 16, c * W_BYTES + ir * 1024, 0, 0);
        }
    };
    auto issue_halo = [&](int c, int hstage) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) halo_piece(j, c, hstage);
    };
    auto issue_slab = [&](int c, int wslot) {
#pragma unroll
        for (int j = 0; j < WJ; ++j) slab_piece(j, c, wslot);
    };

    // ---- pixel fragments. A wave owns MT = 2 M-tiles of 32 pixels; M-tile mi and tap row ky read WINDOW mi + ky of
    // the halo, so the wave needs only MT + 2 = 4 distinct windows per tap column kx instead of 6 fragments (the LDS
    // read port, not the MFMA pipe, bounds this loop: 54 -> 42 / 72 -> 60 ds_read_b128 per wave and step).
    //   16x32 tile: window u = halo row 2*wave + u, pixel r -> halo column r + kx
    //   32x16 tile: window u = halo rows 4*wave + u and 4*wave + u + 2 (lanes r >= 16), halo column (r & 15) + kx
    // and M-tile mi holds output rows 2*wave + mi, resp. 4*wave + mi (+2).
    constexpr int NWIN = MT + 2;
    int a_addr[3][NWIN][2]; // chunk-invariant, depend on the tile shape only
    auto setup_addr = [&](int tws) {
        const int hw = (1 << tws) + 2;
        const int row0 = tws == 5 ? 2 * wave : 4 * wave + 2 * (r >> 4), col0 = tws == 5 ? r : (r & 15);
#pragma unroll
        for (int u = 0; u < NWIN; ++u)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pix = (row0 + u) * hw + col0 + kx;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a_addr[kx][u][ks] = pix * PIXB + (((ks * 2 + h) ^ ((pix >> 2) & 3)) << 4);
            }
    };
    // output position of lane r in M-tile mi (tile coordinates)
    auto out_row = [&](int tws, int mi) { return tws == 5 ? 2 * wave + mi : 4 * wave + mi + 2 * (r >> 4); };
    auto out_col = [&](int tws) { return tws == 5 ? r : (r & 15); };
    int w_addr[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_addr[st][ks] = W_BASE + st * W_BYTES + r * PIXB + (((ks * 2 + h) ^ ((r >> 2) & 3)) << 4);

    f32x16 acc[MT][NI];
    // COUT = 32: the bias lives in an accumulator-shaped register tile (element 4g + q of a lane = channel 8g + 4h + q) that a tile's
    // accumulators start from, so no epilogue adds it; the fused depth head keeps conv3's weights the same way. These kernels are
    // bound by VALU issue at Cin = 32 (36 MFMAs per tile): every per-tile instruction counts.
    f32x16 bias_tile, headw_tile;
    if constexpr (COUT == 32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bias_tile[4 * g + q] = p.bias ? p.bias[8 * g + 4 * h + q] : 0.0f;
                if constexpr (EPI == VX_DC_HEAD_F32) headw_tile[4 * g + q] = p.head_w[8 * g + 4 * h + q];
            }
        }
    }

    // fragment groups: one (tap column kx, k-step ks) = 3 weight fragments per N-tile + 4 pixel windows for 6 NI
    // MFMAs, prefetched one group ahead (register double buffer)
    constexpr int NGRP = 6;
    auto load_group = [&](auto hs_c, auto ws_c, int wslot_off, int grp, f16x8 (&af)[NWIN], f16x8 (&wf)[3][NI]) {
        constexpr int HSt = decltype(hs_c)::value, WSt = decltype(ws_c)::value;
        const int kx = grp >> 1, ks = grp & 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ky][ni] = *reinterpret_cast<const f16x8*>(smem + w_addr[RES ? 0 : WSt][ks] + wslot_off + ((ky * 3 + kx) * COUT + ni * 32) * PIXB);
#pragma unroll
        for (int u = 0; u < NWIN; ++u) af[u] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][u][ks] + HSt * HALO_BYTES);
    };
    auto relu_frag = [](f16x8& a) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = a[j] > (f16)0 ? a[j] : (f16)0;
    };
    // x_residual (COUT = 64): "+ x" of a dense block's last conv (esrgan.cpp:38-40) without reading x again: the
    // centre-tap pixel fragments of chunks 0 and 1 ARE x[0:64], so two extra k-steps per chunk multiply them with
    // (1/s1) * identity (exact in f16) into the matching channel tile; the epilogue's * s1 makes it "+ x".
    const bool xres = NI == 2 && p.x_residual != 0;
    const f16 inv_s1 = (f16)(1.0f / p.s1);
    auto compute_win = [&](auto hs_c, auto ws_c, int chunk, auto&& feed) {
        f16x8 af[2][NWIN], wf[2][3][NI];
        const int wslot_off = RES ? chunk * W_BYTES : 0; // resident slabs: slot = chunk
        load_group(hs_c, ws_c, wslot_off, 0, af[0], wf[0]);
#pragma unroll
        for (int grp = 0; grp < NGRP; ++grp) {
            if (grp + 1 < NGRP) load_group(hs_c, ws_c, wslot_off, grp + 1, af[(grp + 1) & 1], wf[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AR) {
#pragma unroll
                for (int u = 0; u < NWIN; ++u) relu_frag(af[grp & 1][u]);
            }
#pragma unroll
            for (int u = 0; u < NWIN; ++u)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int ky = u - mi;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if (ONE && grp == 0 && ky == 0) // the tile's first MFMA of this accumulator (compile-time after unrolling)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][ky][ni], af[grp & 1][u], bias_tile, 0, 0, 0);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][ky][ni], af[grp & 1][u], acc[mi][ni], 0, 0, 0);
                    }
                }
            feed(2 * grp);
            feed(2 * grp + 1);
            if constexpr (NI == 2) {
                if ((grp >> 1) == 1 && xres && chunk < 2) { // centre tap column; centre tap of M-tile mi = window mi + 1
                    const int ks = grp & 1;
                    f16x8 id;
#pragma unroll
                    for (int j = 0; j < 8; ++j) id[j] = (r == ks * 16 + h * 8 + j) ? inv_s1 : (f16)0;
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        if (chunk == 0) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id, af[grp & 1][mi + 1], acc[mi][0], 0, 0, 0);
                        else acc[mi][NI - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id, af[grp & 1][mi + 1], acc[mi][NI - 1], 0, 0, 0);
                    }
                }
            }
        }
    };
    // COUT = 64 keeps the per-tap loop (one k-step of fragments ahead): its 2 x 2 register blocking already reads
    // one fragment per MFMA, and the wider groups of the window loop cost it 6 % (measured on the same device)
    // fragments are prefetched one GROUP of k-steps ahead (register double buffer): a whole tap (2 k-steps) for
    // COUT = 32, one k-step for COUT = 64 where the accumulators leave fewer registers
    constexpr int G = 1;
    auto load_group_tap = [&](auto hs_c, auto ws_c, int grp, f16x8 (&af)[G][MT], f16x8 (&wf)[G][NI]) {
        constexpr int HSt = decltype(hs_c)::value, WSt = decltype(ws_c)::value;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int step = grp * G + q, tap = step >> 1, ks = step & 1;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[q][ni] = *reinterpret_cast<const f16x8*>(smem + w_addr[WSt][ks] + (tap * COUT + ni * 32) * PIXB);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                af[q][mi] = *reinterpret_cast<const f16x8*>(smem + a_addr[tap % 3][mi + tap / 3][ks] + HSt * HALO_BYTES);
        }
    };
    auto compute_tap = [&](auto hs_c, auto ws_c, int chunk, auto&& feed) {
        f16x8 af[2][G][MT], wf[2][G][NI];
        load_group_tap(hs_c, ws_c, 0, af[0], wf[0]);
#pragma unroll
        for (int grp = 0; grp < 18 / G; ++grp) {
            if (grp + 1 < 18 / G) load_group_tap(hs_c, ws_c, grp + 1, af[(grp + 1) & 1], wf[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AR) {
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) relu_frag(af[grp & 1][q][mi]);
            }
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][q][ni], af[grp & 1][q][mi], acc[mi][ni], 0, 0, 0);
            feed(grp);
            if constexpr (NI == 2) {
                if ((grp * G) >> 1 == 4 && xres && chunk < 2) { // centre tap
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        const int ks = (grp * G + q) & 1;
                        f16x8 id;
#pragma unroll
                        for (int j = 0; j < 8; ++j) id[j] = (r == ks * 16 + h * 8 + j) ? inv_s1 : (f16)0;
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi) {
                            if (chunk == 0) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id, af[grp & 1][q][mi], acc[mi][0], 0, 0, 0);
                            else acc[mi][NI - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id, af[grp & 1][q][mi], acc[mi][NI - 1], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    auto compute = [&](auto hs_c, auto ws_c, int chunk, auto&& feed) {
        if constexpr (COUT == 32) compute_win(hs_c, ws_c, chunk, feed);
        else compute_tap(hs_c, ws_c, chunk, feed);
    };
    // 16-byte group swizzle of the staged output tile. COUT = 32 (64-byte rows): rows ml, ml+4, ml+8, ml+12 of a
    // ds_write_b64 lane group must not share banks (PMC: 14 % LDS conflict cycles with ml & 3 alone); COUT = 64
    // keeps ml & 7 (the wider form measured 1 % slower there)
    auto stage_swz = [](int ml) { return COUT == 32 ? ((ml ^ (ml >> 2)) & 3) : (ml & 7); };
    auto zero_acc = [&]() {
        if constexpr (ONE) return;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if constexpr (COUT == 32) {
                    acc[mi][ni] = bias_tile;
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
                }
            }
    };

    // ---- the block's stream of (tile, chunk) steps. Two cursors run over it: the compute cursor (cur, c) and the
    // halo cursor (h_t, h_c), HS-1 steps ahead. Per step, after the barrier: slab of the next step, then the halo
    // HS-1 steps ahead -- in that order, so that the counted vmcnt of the next step (vector memory returns in
    // order) can leave the younger halo and the tile's output stores in flight.
    geom cur = locate(t_cur);
    int shape = cur.tws;
    setup_addr(shape);
    int src_shape = shape;
    if constexpr (BIL) {
        setup_shape_patch(src_shape);
        setup_patch(cur);
    } else {
        setup_shape_src(src_shape);
        setup_src(cur);
    }
    int h_t = t_cur, h_c = 0;
    auto move_cursor = [&]() { // after the halo (BIL: the source patch) at the cursor has been issued
        if (++h_c == nch) {
            h_c = 0;
            h_t += t_step;
            if (h_t < t_end) {
                const geom nx = locate(h_t);
                if constexpr (BIL) {
                    if (nx.tws != src_shape) {
                        src_shape = nx.tws;
                        setup_shape_patch(src_shape);
                    }
                    setup_patch(nx);
                } else {
                    if (nx.tws != src_shape) {
                        src_shape = nx.tws;
                        setup_shape_src(src_shape);
                    }
                    setup_src(nx);
                }
            }
        }
    };
    auto advance_halo = [&](int hstage) -> bool { // prologue: issues the halo at the cursor, moves the cursor; false = stream ended
        if (h_t >= t_end) return false;
        if constexpr (BIL) issue_patch(h_c, hstage);
        else issue_halo(h_c, hstage);
        move_cursor();
        return true;
    };
    if constexpr (RES) {
        for (int k = 0; k < nch; ++k) issue_slab(k, k); // every slab, once per block; older than any halo, so the first wait covers them
    } else {
        issue_slab(0, 0);
    }
    bool prev_halo = false; // did the previous step issue a halo (younger than the slab this step waits for)?
    if constexpr (BIL) {
        // patches of steps 0 and 1 -> source stages 0 and 1; step 0's halo is interpolated here, every later one inside the step
        // before it. (The step loop's first wait + barrier make it visible.)
        advance_halo(0);
        advance_halo(1);
        setup_interp(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < HJ; ++j) interp_unit(j, 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < HS - 1; ++k) prev_halo = advance_halo(k);
    }
    if (HS == 2) prev_halo = false; // with two stages the only halo in flight is the one this step needs
    zero_acc();
    int c = 0; // chunk of the current tile
    bool stores_in_flight = false;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    unsigned n_steps = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[k] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr (STAMP) t_prev = __builtin_amdgcn_s_memtime();
    f16* const trash = reinterpret_cast<f16*>(g_dconv_trash + ((blockIdx.x % TRASH_BLOCKS) * 8 + wave) * 1024 + lane * 16);

    // One step, computed out of halo stage HSt and slab stage WSt. The HS x 2 instantiations run back to back in
    // the loop below (stage parity is a property of the code position, so every LDS address is base register +
    // immediate); a tile boundary may fall after any of them.
    auto step = [&](auto hs_c, auto ws_c) -> bool {
        constexpr int HSt = decltype(hs_c)::value, WSt = decltype(ws_c)::value;
        // this step's halo and slab must have landed; leave what is younger than them in flight
        {
            const bool st = EPI == VX_DC_F16 && stores_in_flight;
            if (HS == 3 && prev_halo) {
                if (st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HJ + NCH16) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HJ) : "memory");
            } else {
                if (st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH16) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        stores_in_flight = false;
        stamp(0); // waited for the step's DMA
        if constexpr (BIL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's share of the step's interpolated halo is written
        // raw s_barrier: __syncthreads() carries a fence that drains vmcnt, i.e. the halo prefetched for later steps
        __builtin_amdgcn_s_barrier(); // the step's data is in LDS for everyone; everyone has left the previous step's stages
        asm volatile("" ::: "memory");
        stamp(1); // barrier
        const bool last_chunk = c + 1 == nch;
        const bool do_slab = !RES && (!last_chunk || t_cur + t_step < t_end);
        const int slab_c = last_chunk ? 0 : c + 1;
        const bool do_halo = h_t < t_end;
        const int halo_c = h_c;
        constexpr int HNEXT = (HSt + HS - 1) % HS;
        if constexpr (BIL) {
            // slab of the next step; source patch two steps ahead into source stage HSt (this step's patch, consumed by the
            // interpolation during the previous step); the NEXT step's halo is interpolated from source stage HSt ^ 1 into halo stage
            // HSt ^ 1 (free since the barrier) in the shadow of this step's MFMAs: unit j after fragment group j
            if (do_slab) issue_slab(slab_c, WSt ^ 1);
            if (do_halo) {
                issue_patch(halo_c, HSt);
                move_cursor();
            }
            const bool do_interp = !last_chunk || t_cur + t_step < t_end;
            if (do_interp && last_chunk) setup_interp(locate(t_cur + t_step));
            stamp(2);
            compute(hs_c, ws_c, c, [&](int k) {
                if (do_interp && (k & 1) && (k >> 1) < HJ) interp_unit(k >> 1, HSt ^ 1, HSt ^ 1);
            });
        } else if constexpr (COUT == 32) {
            // 36 MFMAs per step: the 8 DMA pieces go out in one burst before the loop (measured 2-3 % faster than
            // feeding them from inside it)
            if (do_slab) issue_slab(slab_c, WSt ^ 1);
            if (do_halo) {
                issue_halo(halo_c, HNEXT);
                move_cursor(); // the next tile's halo sources are set up while the pieces are on their way
            }
            stamp(2); // DMA burst + halo cursor
            compute(hs_c, ws_c, c, [](int) {});
        } else {
            // 72 MFMAs per step: piece k of the ring is issued after the k-th k-step's MFMAs (3 % faster)
            compute(hs_c, ws_c, c, [&](int k) {
                if (k < WJ) {
                    if (do_slab) slab_piece(k, slab_c, WSt ^ 1);
                } else if (k - WJ < HJ) {
                    if (do_halo) halo_piece(k - WJ, halo_c, HNEXT);
                }
            });
        }
        stamp(3); // MFMA loop (COUT = 64: with the DMA pieces fed from inside)
        if (COUT == 64 && do_halo) move_cursor();
        prev_halo = do_halo;
        stamp(2);
        ++n_steps;
        if (++c < nch) return false;

        constexpr int done = HSt; // the halo stage of the last chunk: free once every wave has left compute()

        // ---- epilogue
        if constexpr (EPI == VX_DC_HEAD_F32) {
            // head.conv2 + ReLU + head.conv3 (1x1 -> 1) + ReLU [* max_depth] (depth-anything.cpp:87-94): the 32-channel
            // map never leaves the registers. A lane holds 16 of pixel r's 32 channels, its partner lane (h ^ 1) the rest.
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                float part = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) part += relu1(acc[mi][0][4 * g + q]) * headw_tile[4 * g + q];
                part += __shfl_xor(part, 32, 64);
                const int oy = cur.y0 + out_row(cur.tws, mi), ox = cur.x0 + out_col(cur.tws);
                if (h == 0 && oy < H && ox < W)
                    reinterpret_cast<float*>(p.out)[((long)cur.b * H + oy) * W + ox] = fmaxf(part + p.head_bias, 0.0f) * p.head_scale;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // data-dependent store count: drain before the next counted wait
            prev_halo = false;
        } else if constexpr (EPI == VX_DC_RGB_F32) {
            // channels 0..2 of pixel r sit in acc[mi][0][0..2] of the lanes with h == 0
            if (h == 0) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int oy = cur.y0 + out_row(cur.tws, mi), ox = cur.x0 + out_col(cur.tws);
                    if (oy < H && ox < W) {
                        float* o = reinterpret_cast<float*>(p.out) + (((long)cur.b * H + oy) * W + ox) * 3;
                        o[0] = acc[mi][0][0]; // (bias: accumulator start)
                        o[1] = acc[mi][0][1];
                        o[2] = acc[mi][0][2];
                    }
                }
            }
            // the number of store instructions is data dependent here: drain before the next counted wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prev_halo = false;
        } else {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier(); // every wave is done reading stage `done`: its space stages the f16 tile
            asm volatile("" ::: "memory");
            // rows 0..255 (waves 0-3) in the halo space of the stage, rows 256..511 (waves 4-7) in the slab space
            // of the step (COUT = 32: both halves fit the halo space); each wave stages and drains its own 64 rows
            unsigned char* st;
            if constexpr (COUT == 32) st = smem + done * HALO_BYTES + wave * 64 * PITCH;
            else st = (wave < 4 ? smem + done * HALO_BYTES : smem + W_BASE + WSt * W_BYTES) + (wave & 3) * 64 * PITCH;
            // The epilogue costs as many cycles as the tile's MFMA loops at Cin = 64 (in-kernel stamps, profiles/r03_dconv_stamps_dpt.txt:
            // 7.6k of 20.8k per tile) and it is VALU issue: the activation is chosen by a wave-uniform branch (identity costs nothing,
            // ReLU one instruction), residuals with unit scale are packed-f16 adds, and a row's pixel is tile base + a scalar.
            auto stage_tile = [&](auto act_c) {
                constexpr int ACT = decltype(act_c)::value;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int ml = mi * 32 + r; // row inside the wave's staging block
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nl = ni * 32 + 8 * g + 4 * h;
                            float4 bias = {0.f, 0.f, 0.f, 0.f}; // (COUT = 32: the accumulators started from the bias)
                            if constexpr (COUT != 32) bias = *reinterpret_cast<const float4*>(s_bias + nl);
                            float v[4] = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y,
                                          acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if constexpr (ACT == 2) v[j] = relu1(v[j]);
                                else if constexpr (ACT == 1) v[j] = fmaxf(v[j], 0.2f * v[j]);
                            }
                            if (xres) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] *= p.s1;
                            }
                            f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                            const int c8 = nl >> 2;
                            const int phys16 = (c8 >> 1) ^ stage_swz(ml);
                            *reinterpret_cast<f16x4*>(st + ml * PITCH + phys16 * 16 + (c8 & 1) * 8) = o;
                        }
                }
            };
            if (p.act == 2) stage_tile(std::integral_constant<int, 2>{});
            else if (p.act == 1) stage_tile(std::integral_constant<int, 1>{});
            else stage_tile(std::integral_constant<int, 0>{});
            // drain in batches of NB row groups: residual loads of a batch are issued together (out-of-map pixels
            // read the tile's first pixel, a valid address, and store to the trash page)
            const char* const R1 = reinterpret_cast<const char*>(p.res1);
            const char* const R2 = reinterpret_cast<const char*>(p.res2);
            constexpr int ROWS_PER_IT = 64 / NCH16, IT_PER_MT = 32 / ROWS_PER_IT;
            constexpr int NB = 4; // iterations whose loads are in flight together
            const int j = lane % NCH16, fr0 = lane / NCH16;
            const int jp = j >> 2, je = (j & 3) * 8; // plane of the lane's 8 channels, element offset inside the pixel
            // staged row of iteration t: ml = t * ROWS_PER_IT + fr0, i.e. M-tile t / IT_PER_MT, pixel fr0 + ROWS_PER_IT * (t % IT_PER_MT) of it.
            // 16x32 tiles: output (y0 + 2 wave + M-tile, x0 + pixel); 32x16 tiles: (y0 + 4 wave + M-tile + 2 (pixel >> 4), x0 + (pixel & 15)),
            // where pixel >> 4 and the multiple-of-8 part of pixel & 15 only depend on t: pixel index = lane base + a per-t SCALAR.
            const bool wide = cur.tws == 5;
            const int oy_b = cur.y0 + (wide ? 2 * wave : 4 * wave), ox_b = cur.x0 + (wide ? fr0 : (fr0 & 15));
            const int tile0 = (cur.b * H + cur.y0) * W + cur.x0;                 // always inside the map
            const int base = (cur.b * H + oy_b) * W + ox_b;                     // (pixel indices fit 32 bits: checked by the host)
            const long off1 = ((long)jp * p.res1_plane + je) * 2, off2 = ((long)jp * p.res2_plane + je) * 2, offo = ((long)jp * p.out_plane + je) * 2;
            const bool unit = p.s1 == 1.0f && p.s2 == 1.0f;
            const int rp1 = (int)res1_pix * 2, rp2 = (int)res2_pix * 2, rpo = (int)out_pix * 2; // byte strides between pixels
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-local hand-over of the staged rows: no block barrier
            const float s1 = p.s1, s2 = p.s2;
#pragma unroll
            for (int ib = 0; ib < NCH16; ib += NB) {
                int pixel[NB];
                bool ok[NB];
                f16x8 ra[NB], rc[NB], v[NB];
#pragma unroll
                for (int it = 0; it < NB; ++it) {
                    const int t = ib + it, fm = t / IT_PER_MT, dfr = ROWS_PER_IT * (t % IT_PER_MT); // compile-time after unrolling
                    const int dy = wide ? fm : fm + 2 * (dfr >> 4), dx = wide ? dfr : (dfr & 15);   // wave-uniform
                    ok[it] = oy_b + dy < H && ox_b + dx < W;
                    pixel[it] = ok[it] ? base + dy * W + dx : tile0;
                }
                if (R1) {
#pragma unroll
                    for (int it = 0; it < NB; ++it) ra[it] = *reinterpret_cast<const f16x8*>(R1 + off1 + (long)pixel[it] * rp1);
                }
                if (R2) {
#pragma unroll
                    for (int it = 0; it < NB; ++it) rc[it] = *reinterpret_cast<const f16x8*>(R2 + off2 + (long)pixel[it] * rp2);
                }
#pragma unroll
                for (int it = 0; it < NB; ++it) {
                    const int ml = (ib + it) * ROWS_PER_IT + fr0;
                    v[it] = *reinterpret_cast<const f16x8*>(st + ml * PITCH + (j ^ stage_swz(ml)) * 16);
                }
#pragma unroll
                for (int it = 0; it < NB; ++it) {
                    if (unit) { // the staged value is f16 already: a correctly rounded f16 add gives the same bits as the f32 form below
                        if (R1) v[it] = v[it] + ra[it];
                        if (R2) v[it] = v[it] + rc[it];
                    } else {
                        if (R1) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[it][q] = (f16)((float)v[it][q] * s1 + (float)ra[it][q]);
                        }
                        if (R2) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[it][q] = (f16)((float)v[it][q] * s2 + (float)rc[it][q]);
                        }
                    }
                    f16* dst = ok[it] ? reinterpret_cast<f16*>(reinterpret_cast<char*>(p.out) + offo + (long)pixel[it] * rpo) : trash;
                    *reinterpret_cast<f16x8*>(dst) = v[it];
                }
            }
            stores_in_flight = true;
        }
        stamp(4); // epilogue (its barrier included)

        t_cur += t_step;
        if (t_cur >= t_end) return true;
        cur = locate(t_cur);
        if (cur.tws != shape) {
            shape = cur.tws;
            setup_addr(shape);
        }
        zero_acc();
        c = 0;
        stamp(5); // next tile's geometry
        return false;
    };
    using std::integral_constant;
    for (;;) {
        if constexpr (HS == 3) {
            if (step(integral_constant<int, 0>{}, integral_constant<int, 0>{})) break;
            if (step(integral_constant<int, 1>{}, integral_constant<int, 1>{})) break;
            if (step(integral_constant<int, 2>{}, integral_constant<int, 0>{})) break;
            if constexpr (RES) continue; // no slab parity to track: the pattern repeats after HS steps
            if (step(integral_constant<int, 0>{}, integral_constant<int, 1>{})) break;
            if (step(integral_constant<int, 1>{}, integral_constant<int, 0>{})) break;
            if (step(integral_constant<int, 2>{}, integral_constant<int, 1>{})) break;
        } else {
            if (step(integral_constant<int, 0>{}, integral_constant<int, 0>{})) break;
            if (step(integral_constant<int, 1>{}, integral_constant<int, 1>{})) break;
        }
    }
    if constexpr (STAMP) {
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stamps) + (size_t)blockIdx.x * 8;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = ph[k];
            o[6] = n_steps;
            o[7] = __builtin_amdgcn_s_memtime();
        }
    }
}

int dconv_grid_blocks() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    return n_cu; // one 8-wave block per CU (its LDS ring takes the whole 160 KB)
}

template <int COUT, int RES, int HSV, bool BIL = false>
constexpr int dconv_smem_bytes() {
    return HSV * (5 * NW * 1024) + (BIL ? 2 * 256 * PIXB : 0) + (RES ? RES : 2) * 9 * COUT * PIXB + (BIL ? ((COUT * 4 + 15) & ~15) + (DCONV_TAB_MAX + 4) * 8 : COUT * 4);
}

template <int COUT, int EPI, bool AR, bool STAMP, int RES = 0, int HSV = (COUT == 32 ? 3 : 2), bool BIL = false, bool ONE = false>
int prepare_variant() { // > 64 KB of dynamic LDS needs the attribute; set once, outside any stream capture
    static_assert(dconv_smem_bytes<COUT, RES, HSV, BIL>() <= 160 * 1024, "variant does not fit the LDS");
    VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(&dconv3x3_kernel<COUT, EPI, AR, STAMP, RES, HSV, BIL, ONE>),
                                   dconv_smem_bytes<COUT, RES, HSV, BIL>())); // per (kernel, device), not per process
    return 1;
}

template <int COUT, int EPI, bool AR, bool STAMP, int RES = 0, int HSV = (COUT == 32 ? 3 : 2), bool BIL = false, bool ONE = false>
int launch_variant(const vx_dconv_args& a, hipStream_t s) {
    constexpr int smem = dconv_smem_bytes<COUT, RES, HSV, BIL>();
    if (!prepare_variant<COUT, EPI, AR, STAMP, RES, HSV, BIL, ONE>()) return 0;
    const long tiles = (long)a.B * tile_grid(a.H, a.W).total();
    // Persistent blocks, one per CU at most. The number of ROUNDS (tiles per block) is what the launch takes; given the rounds, the
    // fewest blocks that still do it in that many leave the other CUs to concurrent launches (550 tiles: 184 blocks x 3 instead of
    // 256 blocks doing 3 or 2 -- the step is CU-time bound). Per XCD, because the kernel deals the tiles out by XCD.
    int blocks = (int)(tiles < dconv_grid_blocks() ? tiles : dconv_grid_blocks());
    if (tiles > dconv_grid_blocks() && dconv_grid_blocks() % 8 == 0) {
        const long per_xcd = (tiles + 7) / 8, cu_per_xcd = dconv_grid_blocks() / 8;
        const long rounds = (per_xcd + cu_per_xcd - 1) / cu_per_xcd;
        const long blocks_per_xcd = (per_xcd + rounds - 1) / rounds;
        blocks = (int)(8 * blocks_per_xcd);
    }
    hipLaunchKernelGGL((dconv3x3_kernel<COUT, EPI, AR, STAMP, RES, HSV, BIL, ONE>), dim3(blocks), dim3(512), smem, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

template <int COUT, int EPI>
int launch_dconv(const vx_dconv_args& a, hipStream_t s) {
    if constexpr (COUT == 32 && EPI == VX_DC_HEAD_F32) { // the depth head at Cin = 32 (one step per tile): see ONE
        if (a.cin == 32) return a.bil_hs > 0 ? launch_variant<32, EPI, false, false, 0, 2, true, true>(a, s) : launch_variant<32, EPI, false, false, 0, 3, false, true>(a, s);
    }
    if constexpr (COUT == 32 && (EPI == VX_DC_F16 || EPI == VX_DC_HEAD_F32)) {
        if (a.bil_hs > 0) return launch_variant<32, EPI, false, false, 0, 2, true>(a, s); // interpolating loader
    }
    if constexpr (EPI == VX_DC_F16) {
        if (a.a_relu) return launch_variant<COUT, EPI, true, false>(a, s);
        if (a.stamps) return launch_variant<COUT, EPI, false, true>(a, s);
        if constexpr (COUT == 32) {
            // All slabs of a short conv can stay in LDS. Measured per conv of a dense block (same device, ms per 69
            // launches, streamed -> resident): cin 64 with 3 halo stages 6.62 -> 6.72, cin 96 with 2 stages 7.79 ->
            // 7.75, cin 128 with 2 stages 9.47 -> 9.04. Default: resident at cin = 128 only. VISP_DCONV_RES: 0 = never,
            // 1 = cin 64, 2 = cin <= 128, unset = cin 128.
            static const int res_mode = getenv("VISP_DCONV_RES") ? atoi(getenv("VISP_DCONV_RES")) : -1;
            if (res_mode >= 1 && a.cin == 64) return launch_variant<32, EPI, false, false, 2, 3>(a, s);   // 120 + 36 KB
            if ((res_mode >= 2 && a.cin <= 128) || (res_mode < 0 && a.cin == 128)) return launch_variant<32, EPI, false, false, 4, 2>(a, s); // 80 + 72 KB
        }
    }
    return launch_variant<COUT, EPI, false, false>(a, s);
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

// the same tiles as f32 rgb [tiles][tile_h][tile_w][3] (image_u8_to_f32 with a tile offset, vision.cpp:236-241): the input tensor of the generator's graph
__global__ void esr_tiles_in_f32_kernel(const uint8_t* __restrict__ img, int B, int w, int h, int ch, int ir, int ig, int ib, vx_tile_layout t, float* __restrict__ out) {
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
    out[i * 3 + 0] = (float)s[ir] / 255.0f;
    out[i * 3 + 1] = (float)s[ig] / 255.0f;
    out[i * 3 + 2] = (float)s[ib] / 255.0f;
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

extern "C" int vx_dconv_prepare(void) {
    return prepare_variant<32, VX_DC_F16, false, false>() && prepare_variant<32, VX_DC_F16, true, false>() &&
           prepare_variant<32, VX_DC_F16, false, false, 2, 3>() && prepare_variant<32, VX_DC_F16, false, false, 4, 2>() &&
           prepare_variant<64, VX_DC_F16, false, false>() && prepare_variant<64, VX_DC_F16, true, false>() &&
           prepare_variant<32, VX_DC_RGB_F32, false, false>() && prepare_variant<32, VX_DC_HEAD_F32, false, false>() &&
           prepare_variant<32, VX_DC_F16, false, false, 0, 2, true>() && prepare_variant<32, VX_DC_HEAD_F32, false, false, 0, 2, true>() &&
           prepare_variant<32, VX_DC_HEAD_F32, false, false, 0, 2, true, true>() && prepare_variant<32, VX_DC_HEAD_F32, false, false, 0, 3, false, true>();
}

// the interpolating loader's limits: an 18 x 34 (or 34 x 18) halo must map into a 12 x 21 (21 x 12) source patch
extern "C" int vx_dconv_bilinear_supported(int cout, int H, int W, int hs, int ws) {
    if (cout != 32 || H < 2 || W < 2 || hs < 2 || ws < 2 || hs > H || ws > W || H + W > DCONV_TAB_MAX) return 0;
    const double ry = (double)(hs - 1) / (double)(H - 1), rx = (double)(ws - 1) / (double)(W - 1);
    auto rows = [](double r, int n) { return (int)std::floor((n - 1) * r + 1.0) + 2; }; // source rows under n consecutive output rows, worst phase
    return rows(ry, 18) <= 12 && rows(rx, 34) <= 21 && rows(ry, 34) <= 21 && rows(rx, 18) <= 12;
}

extern "C" int vx_dconv3x3_f16(const vx_dconv_args* args, void* stream) {
    const vx_dconv_args& a = *args;
    VX_REQUIRE(a.x && a.w && a.out, "vx_dconv3x3_f16: null operand");
    VX_REQUIRE(a.cin >= 32 && a.cin % 32 == 0, "vx_dconv3x3_f16: Cin %d must be a multiple of 32", a.cin);
    VX_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "vx_dconv3x3_f16: empty extent");
    VX_REQUIRE((int64_t)a.B * a.H * a.W < (int64_t)0x7fffffff && a.out_pix < (1 << 20) && a.res1_pix < (1 << 20) && a.res2_pix < (1 << 20),
               "vx_dconv3x3_f16: pixel indices must fit 32 bits; run fewer images per call");
    VX_REQUIRE(!a.up2 || (a.H % 2 == 0 && a.W % 2 == 0), "vx_dconv3x3_f16: upsampled extent must be even");
    VX_REQUIRE(a.bil_hs <= 0 || (!a.up2 && !a.a_relu && !a.x_residual && !a.stamps && vx_dconv_bilinear_supported(a.cout, a.H, a.W, a.bil_hs, a.bil_ws)),
               "vx_dconv3x3_f16: bilinear input %dx%d -> %dx%d (cout %d) is outside the interpolating loader's limits", a.bil_ws, a.bil_hs, a.W, a.H, a.cout);
    VX_REQUIRE(a.res2_hs <= 0, "vx_dconv3x3_f16: bilinear res2 is not built");
    VX_REQUIRE((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(a.out) & 15) == 0, "vx_dconv3x3_f16: operands must be 16-byte aligned");
    {
        const int64_t src_pixels = a.bil_hs > 0 ? (int64_t)a.bil_hs * a.bil_ws : (int64_t)(a.H >> (a.up2 ? 1 : 0)) * (a.W >> (a.up2 ? 1 : 0));
        const int64_t xp = a.x_pix ? a.x_pix : 32;
        VX_REQUIRE(xp >= 32 && xp % 8 == 0 && a.out_pix % 8 == 0 && a.res1_pix % 8 == 0 && a.res2_pix % 8 == 0, "vx_dconv3x3_f16: pixel strides must be multiples of 8 (input >= 32)");
        VX_REQUIRE(a.cin == 32 || (a.x_plane % 8 == 0 && (xp > 32 ? a.x_plane >= 32 : a.x_plane >= (int64_t)a.B * src_pixels * 32)), "vx_dconv3x3_f16: bad input plane stride");
        // buffer descriptors address with 32-bit offsets; 2^31 is the zero-fill sentinel
        VX_REQUIRE((int64_t)(a.cin / 32 - 1) * a.x_plane * 2 + src_pixels * xp * 2 < (int64_t)0x7fffffff,
                   "vx_dconv3x3_f16: input planes span more than 2 GiB; run fewer images per call");
    }
    hipStream_t s = as_stream(stream);
    if (a.epi == VX_DC_RGB_F32) {
        VX_REQUIRE(a.cout == 32 && !a.a_relu, "vx_dconv3x3_f16: the rgb head takes weights padded to 32 outputs");
        return launch_dconv<32, VX_DC_RGB_F32>(a, s);
    }
    if (a.epi == VX_DC_HEAD_F32) {
        VX_REQUIRE(a.cout == 32 && a.head_w && !a.a_relu, "vx_dconv3x3_f16: the fused depth head is a 32-channel epilogue with head_w set");
        return launch_dconv<32, VX_DC_HEAD_F32>(a, s);
    }
    VX_REQUIRE(a.epi == VX_DC_F16, "vx_dconv3x3_f16: unknown epilogue %d", a.epi);
    VX_REQUIRE(a.out_plane % 8 == 0 && a.res1_plane % 8 == 0 && a.res2_plane % 8 == 0, "vx_dconv3x3_f16: plane strides must be multiples of 8");
    VX_REQUIRE(!a.x_residual || (a.cout == 64 && a.cin >= 64 && !a.res1 && !a.up2 && a.s1 != 0.0f),
               "vx_dconv3x3_f16: x_residual needs cout = 64 <= cin, no res1, no upsampling");
    if (a.cout == 32) return launch_dconv<32, VX_DC_F16>(a, s);
    if (a.cout == 64) return launch_dconv<64, VX_DC_F16>(a, s);
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

extern "C" int vx_esrgan_tiles_in_f32(const uint8_t* img, int B, int w, int h, int format, const vx_tile_layout* t, float* out, void* stream) {
    int ch, ir, ig, ib;
    switch (format) { // visp::image_format (include/visp/image.h:17-29)
        case 0: ch = 4; ir = 0; ig = 1; ib = 2; break; // rgba_u8
        case 1: ch = 4; ir = 2; ig = 1; ib = 0; break; // bgra_u8
        case 2: ch = 4; ir = 1; ig = 2; ib = 3; break; // argb_u8
        case 3: ch = 3; ir = 0; ig = 1; ib = 2; break; // rgb_u8
        default: vx_set_error("vx_esrgan_tiles_in_f32: unsupported image format %d", format); return 0;
    }
    const long n = (long)B * t->n_x * t->n_y * t->tile_h * t->tile_w;
    hipLaunchKernelGGL(esr_tiles_in_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), img, B, w, h, ch, ir, ig, ib, *t, out);
    VX_LAUNCH_CHECK();
    return 1;
}

extern "C" int vx_esrgan_tiles_out(const float* tiles, int B, const vx_tile_layout* t, float* out_f32, uint8_t* out_rgba, void* stream) {
    const long n = (long)B * t->image_h * t->image_w;
    hipLaunchKernelGGL(esr_tiles_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), tiles, B, *t, out_f32, out_rgba);
    VX_LAUNCH_CHECK();
    return 1;
}
