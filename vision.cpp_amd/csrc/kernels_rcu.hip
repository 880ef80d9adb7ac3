// One launch per DPT residual unit on the SMALL maps of the fusion stages (19^2 .. 74^2 at 518 x 518): reference
// src/visp/arch/depth-anything.cpp:15-33 -- residual_conv = x + conv2(relu(conv1(relu(x)))), feature_fusion = [x0 +] residual_conv(x1),
// then residual_conv again and the 1x1 out_conv.
//
//   out = P( conv2( relu( conv1( relu(x) ) + b1 ) ) + b2 + x [+ res2] ),   P = identity or the 1x1 projection (+ bp)
//
// Why a kernel of its own: on these maps the persistent LDS-ring conv (kernels_dconv.hip) is a 15-17 us launch whatever the map size
// (ring start-up, two chunk steps, block-wide epilogue) and a block holds its CU's whole LDS meanwhile; a residual unit was two of
// those plus an HBM round trip of the intermediate, a fusion stage four or five launches (profiles/r04_dpt_group_ablation.txt: the
// 19^2 / 37^2 / 74^2 stages cost 0.28 ms of the 5.0 ms step for 73 GFLOP).
//
//   * A block (8 waves) owns a TH x TW output tile (the host picks ~10..15 x 19 so that the tiles divide the map evenly). The input region
//     (TH+4) x (TW+4) is copied into LDS once, raw (conv1 reads relu of it, the skip reads it as it is); conv1 runs on the (TH+2) x (TW+2)
//     intermediate region (recomputing a one-pixel rim instead of exchanging it), writes relu(. + b1) as f16 into LDS -- zero where the
//     intermediate pixel lies outside the map: that is conv2's padding -- and conv2 runs on the tile. The intermediate never leaves the CU.
//   * Pixels are indexed FLAT inside a region (M-tile = 32 consecutive pixels of the row-major region, whatever its width): a 19-wide map
//     costs 19-wide rows, not 32. A lane's fragment address is (its pixel's region index + a per-tap scalar) * pitch; the pitch is 144 bytes
//     (128 + 16), so 16 consecutive pixels hit 16 different bank groups without a swizzle.
//   * MFMA 32x32x16 f16, swapped operands as in the other conv kernels (A = weight fragment, rows = output channels; B = pixel fragment): a
//     lane owns one pixel and four consecutive channels per register group. Weights stream through a 4-slot LDS ring, one 8 KB slab per tap
//     (one LDS-DMA instruction per wave, three taps ahead, one barrier per tap). The first form of this kernel read the fragments straight from
//     global memory one tap ahead: 34-44 us per block, every tap waiting on L2 -- slower than the two launches it replaced.
//   * The projection, where the graph has one behind the unit, is a third stage on the f16 tile staged in the (then free) intermediate space.
#include "vx_common.h"

#include <type_traits>

namespace {

constexpr int RC = 64;            // channels (in = mid = out)
constexpr int RPITCH = 144;       // bytes per pixel in LDS
constexpr int RNW = 8;            // waves per block

struct rcu_geom {
    int TH, TW, ncx, ncy;         // tile extent, tiles per row / column of tiles
    int n_in, n_mid, n_out;       // pixels of the three regions
    int lds;                      // bytes
};

__host__ inline rcu_geom rcu_geometry(int H, int W) {
    rcu_geom g;
    g.ncx = (W + 19) / 20;
    g.TW = (W + g.ncx - 1) / g.ncx;
    int th = 320 / g.TW;
    if (th > H) th = H;
    g.ncy = (H + th - 1) / th;
    g.TH = (H + g.ncy - 1) / g.ncy;
    g.n_in = (g.TH + 4) * (g.TW + 4);
    g.n_mid = (g.TH + 2) * (g.TW + 2);
    g.n_out = g.TH * g.TW;
    g.lds = (g.n_in + ((g.n_mid + 31) & ~31)) * RPITCH + 4 * (RC * RC * 2) + 3 * RC * 4;
    return g;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int RSLAB = RC * RC * 2; // one tap of a conv (or the projection): [64 output channels][64 input channels] f16, 8 KB
constexpr int RRING = 4;           // slabs in the LDS ring: three taps in flight ahead of the one being multiplied

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}

template <bool PROJ>
__global__ __launch_bounds__(64 * RNW) void rcu_fused_kernel(const vx_rcu_args p, const int TH, const int TW, const int ncx, const int ncy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int IW = TW + 4, MW = TW + 2;
    const int n_in = (TH + 4) * IW, n_mid = (TH + 2) * MW, n_out = TH * TW;
    unsigned char* const s_in = smem;
    unsigned char* const s_mid = smem + n_in * RPITCH;
    unsigned char* const s_ring = s_mid + ((n_mid + 31) & ~31) * RPITCH;
    float* const s_bias = reinterpret_cast<float*>(s_ring + RRING * RSLAB); // b1 | b2 | bp

    const int tile = blockIdx.x;
    const int b = tile / (ncx * ncy), trem = tile - b * (ncx * ncy);
    const int ty = trem / ncx, tx = trem - ty * ncx;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weight slabs: taps 0..8 of conv1, 0..8 of conv2, then the projection; each is a linear 8 KB LDS image (the host stored the 16-byte
    // groups of output channel n at position g ^ ((n >> 1) & 7)), copied by one LDS-DMA instruction per wave into slot (slab & 3) three slabs ahead
    constexpr int N_SLABS = PROJ ? 19 : 18;
    auto issue_slab = [&](int sl) {
        const unsigned char* src = sl < 9 ? reinterpret_cast<const unsigned char*>(p.w1) + sl * RSLAB
                                          : (sl < 18 ? reinterpret_cast<const unsigned char*>(p.w2) + (sl - 9) * RSLAB : reinterpret_cast<const unsigned char*>(p.wp));
        __builtin_amdgcn_global_load_lds((gptr_t)(src + wave * 1024 + lane * 16), (lptr_t)(s_ring + (sl & (RRING - 1)) * RSLAB + wave * 1024), 16, 0, 0);
    };

    if (tid < RC) {
        s_bias[tid] = p.b1 ? p.b1[tid] : 0.0f;
        s_bias[RC + tid] = p.b2 ? p.b2[tid] : 0.0f;
        s_bias[2 * RC + tid] = PROJ && p.bp ? p.bp[tid] : 0.0f;
    }

    issue_slab(0);
    issue_slab(1);
    issue_slab(2);

    // ---- input region, raw: pixel (y0 - 2 + ry, x0 - 2 + rx), zero outside the map (conv1's padding)
    {
        const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x) + (long)b * H * W * RC;
        const int chunks = n_in * 8;
        constexpr int PER = 8; // <= 8 * 512 = 4096 chunks = 512 region pixels (the host keeps n_in below that)
        f16x8 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + i * 64 * RNW;
            const int pix = id >> 3, c = id & 7;
            const int ry = pix / IW, rx = pix - ry * IW;
            const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
            v[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (id < chunks && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[i] = *reinterpret_cast<const f16x8*>(X + ((long)gy * W + gx) * RC + c * 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the region's loads (and the first three slabs) before the tap loop starts counting copies
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + i * 64 * RNW;
            if (id < chunks) *reinterpret_cast<f16x8*>(s_in + (id >> 3) * RPITCH + (id & 7) * 16) = v[i];
        }
    }
    // weight fragment of (32-channel half ni, k-step ks) in a slab: row ni * 32 + r, 16-byte group (2 ks + h) ^ ((r >> 1) & 7): the rows of 16
    // consecutive lanes alternate between the two 128-byte halves of the bank row and take 8 different groups inside each
    const int w_off = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);

    // one 3x3 conv stage over a flat region: this wave's M-tiles are wave and wave + 8. src_w = row length of the source region, dst_w of the
    // destination region (the source index of destination pixel (y, x) at tap (ky, kx) is (y + ky) * src_w + x + kx). Every wave runs the
    // tap loop (slab copies and barriers); a wave without an M-tile multiplies nothing.
    f32x16 acc[2][2];
    auto conv_stage = [&](auto first_slab, const unsigned char* src, int src_w, int dst_w, int n_dst, bool relu_in) {
        constexpr int S0 = decltype(first_slab)::value;
        const int n_mt = (n_dst + 31) >> 5;
        const bool any = wave < n_mt, two = wave + RNW < n_mt; // (wave-uniform)
        int base[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            int pd = (wave + mi * RNW) * 32 + r;
            if (pd >= n_dst) pd = 0; // a lane without a pixel computes pixel 0 again; nothing of it is kept
            const int y = pd / dst_w, x = pd - y * dst_w;
            base[mi] = (y * src_w + x) * RPITCH + h * 16;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sl = S0 + tap;
            // slab sl has landed once at most min(2, slabs issued after it) copies of this wave are in flight; the barrier extends that to every wave's
            // part of it and says that everybody is done with slab sl - 1, whose slot the copy issued next overwrites
            if (sl + 2 < N_SLABS) wait_vm<2>();
            else if (sl + 1 < N_SLABS) wait_vm<1>();
            else wait_vm<0>();
            __syncthreads();
            if (sl + 3 < N_SLABS) issue_slab(sl + 3);
            if (!any) continue;
            const unsigned char* ws = s_ring + (sl & (RRING - 1)) * RSLAB;
            f16x8 wf[2][4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wf[ni][ks] = *reinterpret_cast<const f16x8*>(ws + ni * 4096 + (w_off ^ (ks << 5)));
            const int toff = ((tap / 3) * src_w + tap % 3) * RPITCH; // (scalar)
            auto one = [&](int mi) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f16x8 af = *reinterpret_cast<const f16x8*>(src + base[mi] + toff + ks * 32);
                    if (relu_in) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) af[j] = af[j] > (f16)0 ? af[j] : (f16)0;
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni][ks], af, acc[mi][ni], 0, 0, 0);
                }
            };
            one(0);
            if (two) one(1);
        }
    };

    // ---- conv1 over the intermediate region: relu(conv1(relu(x)) + b1) -> LDS, zero outside the map
    // (its first tap's barrier publishes the input region and the biases)
    conv_stage(std::integral_constant<int, 0>{}, s_in, IW, MW, n_mid, true);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pd = (wave + mi * RNW) * 32 + r;
        if (pd >= n_mid) continue;
        const int y = pd / MW, x = pd - y * MW;
        const bool inside = (unsigned)(y0 - 1 + y) < (unsigned)H && (unsigned)(x0 - 1 + x) < (unsigned)W;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = ni * 32 + 8 * g + 4 * h;
                const float4 bias = *reinterpret_cast<const float4*>(s_bias + nl);
                f16x4 o;
                o[0] = (f16)(inside ? fmaxf(acc[mi][ni][4 * g + 0] + bias.x, 0.0f) : 0.0f);
                o[1] = (f16)(inside ? fmaxf(acc[mi][ni][4 * g + 1] + bias.y, 0.0f) : 0.0f);
                o[2] = (f16)(inside ? fmaxf(acc[mi][ni][4 * g + 2] + bias.z, 0.0f) : 0.0f);
                o[3] = (f16)(inside ? fmaxf(acc[mi][ni][4 * g + 3] + bias.w, 0.0f) : 0.0f);
                *reinterpret_cast<f16x4*>(s_mid + pd * RPITCH + nl * 2) = o;
            }
    }

    // ---- conv2 over the tile (its first tap's barrier publishes the intermediate), + b2 + x (+ res2)
    conv_stage(std::integral_constant<int, 9>{}, s_mid, MW, TW, n_out, false);
    if constexpr (PROJ) __syncthreads(); // every wave is done reading the intermediate: its space stages the tile for the projection
    const f16* __restrict__ R2 = reinterpret_cast<const f16*>(p.res2);
    f16* __restrict__ O = reinterpret_cast<f16*>(p.out);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pd = (wave + mi * RNW) * 32 + r;
        if (pd >= n_out) continue;
        const int y = pd / TW, x = pd - y * TW;
        const int gy = y0 + y, gx = x0 + x;
        const bool inside = gy < H && gx < W;
        const long gpix = ((long)b * H + gy) * W + gx;
        const unsigned char* xin = s_in + ((y + 2) * IW + x + 2) * RPITCH;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = ni * 32 + 8 * g + 4 * h;
                const float4 bias = *reinterpret_cast<const float4*>(s_bias + RC + nl);
                const f16x4 xs = *reinterpret_cast<const f16x4*>(xin + nl * 2);
                float v[4] = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y, acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
                // the unfused form rounds conv2 + b2 to f16 before the adds (an f16 tile + f16 residuals): keep its arithmetic
                f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                o = o + xs;
                if (R2 && inside) o = o + *reinterpret_cast<const f16x4*>(R2 + gpix * RC + nl);
                if constexpr (PROJ) *reinterpret_cast<f16x4*>(s_mid + pd * RPITCH + nl * 2) = o;
                else if (inside) *reinterpret_cast<f16x4*>(O + gpix * RC + nl) = o;
            }
    }
    if constexpr (!PROJ) return;

    // ---- the projection: 1x1, 64 -> 64 on the staged tile (slab 18)
    {
        wait_vm<0>();
        __syncthreads(); // the slab and the staged tile
        const int n_mt = (n_out + 31) >> 5;
        if (wave >= n_mt) return;
        const bool two = wave + RNW < n_mt;
        f16x8 wf[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[ni][ks] = *reinterpret_cast<const f16x8*>(s_ring + (18 & (RRING - 1)) * RSLAB + ni * 4096 + (w_off ^ (ks << 5)));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1 && !two) break;
            int pd = (wave + mi * RNW) * 32 + r;
            const bool have = pd < n_out;
            if (!have) pd = 0;
            f32x16 a2[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) a2[ni][e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(s_mid + pd * RPITCH + ks * 32 + h * 16);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) a2[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni][ks], af, a2[ni], 0, 0, 0);
            }
            const int y = pd / TW, x = pd - y * TW;
            const int gy = y0 + y, gx = x0 + x;
            if (!have || gy >= H || gx >= W) continue;
            const long gpix = ((long)b * H + gy) * W + gx;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = ni * 32 + 8 * g + 4 * h;
                    const float4 bias = *reinterpret_cast<const float4*>(s_bias + 2 * RC + nl);
                    f16x4 o = {(f16)(a2[ni][4 * g + 0] + bias.x), (f16)(a2[ni][4 * g + 1] + bias.y), (f16)(a2[ni][4 * g + 2] + bias.z), (f16)(a2[ni][4 * g + 3] + bias.w)};
                    *reinterpret_cast<f16x4*>(O + gpix * RC + nl) = o;
                }
        }
    }
}

} // namespace

extern "C" int vx_rcu_supported(int H, int W) {
    if (H < 1 || W < 1 || H > 96 || W > 96) return 0; // larger maps: the LDS-ring conv is the faster form
    const rcu_geom g = rcu_geometry(H, W);
    return g.n_in <= 512 && g.n_mid <= 512 && g.n_out <= 512 && g.lds <= 160 * 1024;
}

extern "C" int vx_rcu_fused_f16(const vx_rcu_args* args, void* stream) {
    const vx_rcu_args& a = *args;
    VX_REQUIRE(a.B > 0 && vx_rcu_supported(a.H, a.W), "vx_rcu_fused_f16: map %d x %d is not built (1..96 per side)", a.W, a.H);
    VX_REQUIRE(a.x && a.w1 && a.w2 && a.out, "vx_rcu_fused_f16: missing operands");
    const rcu_geom g = rcu_geometry(a.H, a.W);
    const long blocks = (long)a.B * g.ncx * g.ncy;
    VX_REQUIRE(blocks < (1l << 30), "vx_rcu_fused_f16: too many tiles");
    if (a.wp) {
        VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(&rcu_fused_kernel<true>), g.lds));
        hipLaunchKernelGGL((rcu_fused_kernel<true>), dim3((unsigned)blocks), dim3(64 * RNW), g.lds, as_stream(stream), a, g.TH, g.TW, g.ncx, g.ncy);
    } else {
        VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(&rcu_fused_kernel<false>), g.lds));
        hipLaunchKernelGGL((rcu_fused_kernel<false>), dim3((unsigned)blocks), dim3(64 * RNW), g.lds, as_stream(stream), a, g.TH, g.TW, g.ncx, g.ncy);
    }
    VX_LAUNCH_CHECK();
    return 1;
}
