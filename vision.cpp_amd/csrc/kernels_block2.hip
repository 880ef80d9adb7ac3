// Token-stationary DINOv2 block kernel, two waves per SIMD (schedule 2). Same contract, operands and slab streams as
// kernels_block.hip (vx_dino_block_args, vx_dino_block_pack_*): one launch does, for 128 token rows per workgroup,
//     x += lambda1 (att Wo^T + bo);  x += lambda2 (gelu(LN2(x) W1^T + b1) W2^T + b2);  [feat = LN_final(x)];
//     [q, k, v = LN1'(x) Wqkv'^T + b']                                                  (reference dino.cpp:48-107)
//
// Why a second form. One wave per SIMD (kernels_block.hip) is ISSUE bound: a hidden tile costs 48 MFMAs but ~385 issued
// instructions (a ds_read and a wait per MFMA, ~190 VALU for GELU, the ring feed), and a single in-order wave issues one
// instruction per ~4-16 cycles -- 3500 cycles per 48 MFMAs measured (profiles/r02_block_kernel_anatomy.txt) against 1536
// of matrix-pipe time. Two waves on a SIMD issue side by side, so the work of one 32-token group is split between a PAIR
// of waves (w, w + 4: same SIMD) with 256 registers each, everything in architectural VGPRs:
//   wave A (producer): the LayerNorm-ed token fragments (96 VGPRs), fc1 tile accumulation + GELU, the weight ring feed;
//   wave B (consumer): the 192 fc2 accumulators of the group; nothing else but MFMAs and their fragment reads.
// A hands each activated 32 x 32 hidden tile to B as 2 KiB of f16 B-operand fragments through LDS; the step barrier the
// weight ring needs anyway orders the hand-off. In the phases without the 192 accumulators live (output projection, QKV)
// both waves hold the token fragments and take every other feature tile with its epilogue; the residual stream passes
// between phases through global memory (x is written anyway), which also gives each wave the whole row for its LayerNorm.
#include "vx_common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int D = 384, HID = 1536, NT = D / 32, KS = D / 16, SLAB = KS * 1024, PIECES = KS / 4;
constexpr int N_OUT = NT, N_MLP = 2 * (HID / 32), N_QKV = 3 * NT, NU = HID / 32;
constexpr int V_BO = 0, V_LAM1 = 384, V_G2 = 768, V_B2 = 1152, V_B1 = 1536, V_BFC2 = 3072, V_LAM2 = 3456;
constexpr int V_GN = 3840, V_BN = 4224, V_BQKV = 4608, V_GF = 5760, V_BF = 6144, V_TOTAL = 6528;
constexpr int PF = 4;
constexpr int SMEM_RING = 4 * SLAB;           // two pairs of slabs: the pair in use, the pair being written
constexpr int SMEM_EXCH = 2 * 4 * 2048;       // hidden-tile hand-off: 2 buffers x 4 wave pairs x 2 KiB
constexpr int SMEM_BYTES = SMEM_RING + SMEM_EXCH + V_TOTAL * 4;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#define CI(x) (decltype(x)::value)

__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ u32x4 widen_pair(u32x2 a, u32x2 b) {
    auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
    return o;
}
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    f16x4 v = {(f16)a, (f16)b, (f16)c, (f16)d};
    return __builtin_bit_cast(u32x2, v);
}

template <bool MLP, bool QKV, bool TAP>
__global__ __launch_bounds__(512, 2) void dino_block2_kernel(const vx_dino_block_args args) {
    float* const a_x = args.x; const void* const a_att = args.att; const void* const a_wmlp = args.w_mlp; const void* const a_wqkv = args.w_qkv;
    const float* const a_vmlp = args.vec_mlp; const float* const a_vqkv = args.vec_qkv; const float* const a_vtap = args.vec_tap;
    void* const a_feat = args.feat; void* const a_q = args.q; void* const a_k = args.k; void* const a_v = args.v; float* const a_cap = args.cap_x1;
    const int a_M = args.M, a_T = args.T, a_H = args.H; const float a_qs = args.q_scale, a_eps = args.eps;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const exch = smem + SMEM_RING;
    float* const vec = reinterpret_cast<float*>(smem + SMEM_RING + SMEM_EXCH);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3;          // token group of this wave pair
    const bool role_b = wave >= 4;      // waves w and w + 4 share a SIMD
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + pair * 32 + r;

    constexpr int n_mlp_slabs = MLP ? N_OUT + N_MLP : 0;
    constexpr int n_slabs = n_mlp_slabs + (QKV ? N_QKV : 0);

    const unsigned row_bytes_f32 = D * 4, row_bytes_f16 = D * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a_x, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
    const unsigned xoff = (unsigned)m * row_bytes_f32 + 16 * h;
    auto ld_x = [&](int t, int g) __attribute__((always_inline)) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff + (32 * t + 8 * g) * 4, 0, 0));
    };
    auto st_x = [&](int t, int g, f32x4 v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x, xoff + (32 * t + 8 * g) * 4, 0, 0);
    };

    // ---- weight ring: 4 stages = the slab pair in use + the pair of the next step. Right after a step's barrier every wave
    // copies 3 + 3 pieces (1 KiB each) of the next pair global -> LDS with LDS-DMA: no staging registers (the 256-register
    // budget of a wave has none to spare) and, with two waves per SIMD, the other wave covers the DMA issue cost. The
    // step-closing __syncthreads() waits for the copies (hipcc drains vmcnt before a barrier while an LDS-DMA is in flight).
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned char* const src_mlp = static_cast<const unsigned char*>(a_wmlp) + (size_t)wave * 3 * 1024 + lane * 16;
    const unsigned char* const src_qkv = static_cast<const unsigned char*>(a_wqkv) + (size_t)wave * 3 * 1024 + lane * 16;
    auto slab_src = [&](int k) __attribute__((always_inline)) -> const unsigned char* {
        const int kk = k < n_slabs ? k : n_slabs - 1; // past the end: a harmless re-read instead of a branch
        if constexpr (MLP && QKV) return kk < n_mlp_slabs ? src_mlp + (size_t)kk * SLAB : src_qkv + (size_t)(kk - n_mlp_slabs) * SLAB;
        else if constexpr (MLP) return src_mlp + (size_t)kk * SLAB;
        else return src_qkv + (size_t)kk * SLAB;
    };
    unsigned char* const wr0 = ring + wave * 3 * 1024;      // this wave's first piece inside stage 0 (wave-uniform: DMA base)
    const unsigned char* const rd0 = ring + lane * 16;
    const bool no_dma = args.stamps != nullptr; // diagnostics (tools/bench_block.py): run without the weight stream (results are garbage)
    auto dma_pair = [&](int k, int stage) __attribute__((always_inline)) { // slabs k, k+1 -> stages stage, stage+1
        if (no_dma && k > 0) return;
        const unsigned char *s0 = slab_src(k), *s1 = slab_src(k + 1);
#pragma unroll
        for (int z = 0; z < 3; ++z) {
            __builtin_amdgcn_global_load_lds((gptr_t)(s0 + z * 1024), (lptr_t)(wr0 + stage * SLAB + z * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(s1 + z * 1024), (lptr_t)(wr0 + (stage + 1) * SLAB + z * 1024), 16, 0, 0);
        }
    };
    dma_pair(0, 0);
    {
        if constexpr (MLP)
            for (int i = tid; i < 3840 / 4; i += 512) reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(a_vmlp)[i];
        if constexpr (QKV)
            for (int i = tid; i < 1920 / 4; i += 512) reinterpret_cast<float4*>(vec + V_GN)[i] = reinterpret_cast<const float4*>(a_vqkv)[i];
        if constexpr (TAP)
            for (int i = tid; i < 768 / 4; i += 512) reinterpret_cast<float4*>(vec + V_GF)[i] = reinterpret_cast<const float4*>(a_vtap)[i];
    }

    int k = 0, st = 0;
    const unsigned char *curx = rd0, *cury = rd0 + SLAB;
    f16x8 wf[PF]; // fragment window of this wave's stream

    // A step = one workgroup barrier (the pair (k, k+1) has landed, the previous pair's stages are free) + the DMA of the next pair
    auto step_open = [&]() __attribute__((always_inline)) {
        __syncthreads();
        curx = rd0 + st * SLAB;
        cury = curx + SLAB;
        dma_pair(k + 2, st ^ 2);
        st ^= 2;
        k += 2;
    };
    // one slab as a stream of 24 fragments: mm(f, fragment) + side work per slot
    auto run_slab = [&](const unsigned char* base, auto&& mm, auto&& side) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF; ++i) wf[i] = *reinterpret_cast<const f16x8*>(base + i * 1024);
        static_for<KS>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = CI(fc);
            mm(fc, wf[f % PF]);
            if constexpr (f + PF < KS) wf[f % PF] = *reinterpret_cast<const f16x8*>(base + (f + PF) * 1024);
            side(fc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_side = [](auto) {};

    __syncthreads(); // slabs 0 and 1 and the vectors are in LDS

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto vec_tile = [&](const float* v) __attribute__((always_inline)) -> f32x16 {
        f32x16 c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 q = *reinterpret_cast<const float4*>(v + 8 * g + 4 * h);
            c[4 * g + 0] = q.x; c[4 * g + 1] = q.y; c[4 * g + 2] = q.z; c[4 * g + 3] = q.w;
        }
        return c;
    };

    f16x8 xb[KS];    // token fragments (B operand): A always; B in the symmetric phases
    f32x16 acc[NT];  // a token row (192 f32 per lane): the residual stream while it is normalised; B's fc2 accumulators
    auto chain = [&](f32x16& c) __attribute__((always_inline)) {
        return [&](auto fc, const f16x8& w) __attribute__((always_inline)) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xb[CI(fc)], c, 0, 0, 0); };
    };
    auto load_row = [&]() __attribute__((always_inline)) { // the residual stream row of this lane from memory
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = ld_x(t, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = v[i];
            }
    };
    float mean = 0.f, rstd = 0.f;
    auto ln_stats = [&]() __attribute__((always_inline)) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; e += 2) { s0 += acc[t][e]; s1 += acc[t][e + 1]; }
        float s = s0 + s1;
        s += other_half(s);
        mean = s * (1.0f / D);
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float d0 = acc[t][e] - mean, d1 = acc[t][e + 1] - mean;
                q0 = fmaf(d0, d0, q0);
                q1 = fmaf(d1, d1, q1);
            }
        float q = q0 + q1;
        q += other_half(q);
        rstd = __builtin_amdgcn_rsqf(fmaf(q, 1.0f / D, a_eps));
    };
    // tile by tile, fenced: a row tile's 16 values die as its 8 fragment registers are made (the row and the fragments together
    // would not fit the 256-register budget)
    auto ln_to_frags = [&](const float* gamma, const float* beta) __attribute__((always_inline)) {
        static_for<NT>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = CI(tc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + 32 * t + 8 * g + 4 * h);
                const float4 bt = *reinterpret_cast<const float4*>(beta + 32 * t + 8 * g + 4 * h);
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xb[2 * t + (g >> 1)][(g & 1) * 4 + i] = (f16)fmaf((acc[t][4 * g + i] - mean) * rstd, gg[i], bb[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // global memory hand-off between the waves of this workgroup: stores done and visible, then the barrier
    auto publish = [&]() __attribute__((always_inline)) {
        __threadfence_block();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if constexpr (MLP) {
        // ---- x += lambda1 * (att Wo^T + bo): both waves hold the attention rows; A takes the even tiles, B the odd ones
        {
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a_att), 0, (int)((long)a_M * row_bytes_f16), 0x00020000);
            const unsigned aoff = (unsigned)m * row_bytes_f16 + 16 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) xb[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff + 32 * s, 0, 0));
        }
        f32x16 cp = zero;
        f32x4 xin[4], xpv[4];
        const int t_own = role_b ? 1 : 0;
        auto resid_group = [&](int t, int g) __attribute__((always_inline)) {
            const float4 lm = *reinterpret_cast<const float4*>(vec + V_LAM1 + 32 * t + 8 * g + 4 * h);
            f32x4 o = {fmaf(cp[4 * g + 0], lm.x, xpv[g][0]), fmaf(cp[4 * g + 1], lm.y, xpv[g][1]),
                       fmaf(cp[4 * g + 2], lm.z, xpv[g][2]), fmaf(cp[4 * g + 3], lm.w, xpv[g][3])};
            st_x(t, g, o);
        };
#pragma unroll 1
        for (int j = 0; j < NT / 2; ++j) {
            const int t = 2 * j + t_own;
            f32x16 c = vec_tile(vec + V_BO + 32 * t);
            step_open();
            auto side = [&](auto fc) __attribute__((always_inline)) {
                constexpr int f = CI(fc);
                if constexpr (f < 4) xin[f] = ld_x(t, f);
                if constexpr (f >= 8 && f % 4 == 0) { if (j > 0) resid_group(t - 2, f / 4 - 2); }
            };
            if (role_b) run_slab(cury, chain(c), side);
            else run_slab(curx, chain(c), side);
            cp = c;
#pragma unroll
            for (int g = 0; g < 4; ++g) xpv[g] = xin[g];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) resid_group(NT - 2 + t_own, g);
        publish();
        __syncthreads(); // every tile of x (after the attention half) is in memory
        if (a_cap && !role_b) {
            load_row();
            const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a_cap, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, xoff + (32 * t + 8 * g) * 4, 0, 0);
                }
        }

        // ---- mlp. A: LN2 of the row -> token fragments; per step fc1 of tile u+1 and GELU of tile u. B: fc2 of tile u-1.
        // Slab pairs: [W1(0), W1(1)], [W1(u+1), W2(u-1)] for u = 1..46, [W2(46), W2(47)].
        const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f, c3 = c1 * 0.044715f;
        unsigned char* const ex = exch + pair * 2048 + lane * 16; // this lane's 16 bytes of fragment 0; + 1024 fragment 1; + 8192 other buffer
        if (!role_b) {
            load_row();
            ln_stats();
            ln_to_frags(vec + V_G2, vec + V_B2);
            f32x16 hc, hn;
            float gt[16], g1[16], g2[16];
            f16x8 hbn[2];
            auto gelu_op = [&](auto ec, auto opc) __attribute__((always_inline)) {
                constexpr int e = CI(ec), op = CI(opc);
                if constexpr (op == 0) g1[e] = gt[e] * gt[e];
                if constexpr (op == 1) g1[e] = fmaf(g1[e], c3, c1);
                if constexpr (op == 2) g1[e] = gt[e] * g1[e];
                if constexpr (op == 3) g2[e] = __builtin_amdgcn_exp2f(g1[e]);
                if constexpr (op == 4) g2[e] = 1.0f + g2[e];
                if constexpr (op == 5) g2[e] = __builtin_amdgcn_rcpf(g2[e]);
                if constexpr (op == 6) hbn[e >> 3][e & 7] = (f16)(gt[e] * g2[e]);
            };
            // 16 elements x 7 dependent operations over the slab's 24 slots: element e starts in slot e, one operation per slot
            auto gelu_side = [&](auto fc) __attribute__((always_inline)) {
                constexpr int f = CI(fc);
                if constexpr (f == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) gt[e] = hc[e];
                }
                static_for<16>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int op = f - 1 - CI(ec);
                    if constexpr (op >= 0 && op < 7) gelu_op(ec, std::integral_constant<int, (op >= 0 && op < 7 ? op : 0)>{});
                });
            };
            auto gelu_store = [&](int buf) __attribute__((always_inline)) { // hand the activated tile to wave B
                *reinterpret_cast<f16x8*>(ex + buf * 8192) = hbn[0];
                *reinterpret_cast<f16x8*>(ex + buf * 8192 + 1024) = hbn[1];
            };
            hc = vec_tile(vec + V_B1);
            hn = vec_tile(vec + V_B1 + 32);
            step_open();                                                  // [W1(0), W1(1)]
            run_slab(curx, chain(hc), no_side);
            run_slab(cury, chain(hn), no_side);
            static_for<KS>(gelu_side);                                    // GELU(0)
            gelu_store(0);
            hc = hn;
#pragma unroll 1
            for (int u = 1; u < NU - 1; ++u) {
                hn = vec_tile(vec + V_B1 + 32 * (u + 1));
                step_open();                                              // [W1(u+1), W2(u-1)]
                run_slab(curx, chain(hn), gelu_side);                     // GELU(u) in fc1(u+1)'s slots
                gelu_store(u & 1);
                hc = hn;
            }
            step_open();                                                  // [W2(46), W2(47)]: B's; A activates tile 47
            static_for<KS>(gelu_side);
            gelu_store((NU - 1) & 1);
            __syncthreads();                                              // tile 47 handed over
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = vec_tile(vec + V_BFC2 + 32 * t);
            f16x8 hb[2];
            auto fc2 = [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
                constexpr int f = CI(fc);
                acc[f % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, hb[f / NT], acc[f % NT], 0, 0, 0);
            };
            auto take = [&](int buf) __attribute__((always_inline)) {
                hb[0] = *reinterpret_cast<const f16x8*>(ex + buf * 8192);
                hb[1] = *reinterpret_cast<const f16x8*>(ex + buf * 8192 + 1024);
            };
            step_open();                                                  // [W1(0), W1(1)]: A's
#pragma unroll 1
            for (int u = 1; u < NU - 1; ++u) {
                step_open();                                              // [W1(u+1), W2(u-1)]
                take((u - 1) & 1);
                run_slab(cury, fc2, no_side);
            }
            step_open();                                                  // [W2(46), W2(47)]
            take((NU - 2) & 1);
            run_slab(curx, fc2, no_side);
            __syncthreads();                                              // tile 47 handed over
            take((NU - 1) & 1);
            run_slab(cury, fc2, no_side);
            // ---- x += lambda2 * (fc2 + b2)
            {
                f32x4 xi[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) xi[t][g] = ld_x(t, g);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 lm = *reinterpret_cast<const float4*>(vec + V_LAM2 + 32 * t + 8 * g + 4 * h);
                        f32x4 o = {fmaf(acc[t][4 * g + 0], lm.x, xi[t][g][0]), fmaf(acc[t][4 * g + 1], lm.y, xi[t][g][1]),
                                   fmaf(acc[t][4 * g + 2], lm.z, xi[t][g][2]), fmaf(acc[t][4 * g + 3], lm.w, xi[t][g][3])};
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = o[i];
                        st_x(t, g, o);
                    }
            }
            publish();
        }
        if constexpr (TAP || QKV) {
            __syncthreads(); // the block's output rows are in memory
            if (!role_b) load_row();
        }
    } else {
        load_row(); // first layer: the residual stream as it is
    }

    if constexpr (TAP || QKV) ln_stats(); // both waves hold the whole row: each normalises for its own tiles

    if constexpr (TAP) {
        if (role_b == (MLP ? true : false)) { // one wave of the pair writes the tap (B holds the row in registers already)
            const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(a_feat, 0, (int)((long)a_M * row_bytes_f16), 0x00020000);
            const unsigned foff = (unsigned)m * row_bytes_f16 + 16 * h;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    u32x2 pk[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = 2 * pr + gg;
                        const float4 gm = *reinterpret_cast<const float4*>(vec + V_GF + 32 * t + 8 * g + 4 * h);
                        const float4 bt = *reinterpret_cast<const float4*>(vec + V_BF + 32 * t + 8 * g + 4 * h);
                        pk[gg] = pack4(fmaf((acc[t][4 * g + 0] - mean) * rstd, gm.x, bt.x), fmaf((acc[t][4 * g + 1] - mean) * rstd, gm.y, bt.y),
                                       fmaf((acc[t][4 * g + 2] - mean) * rstd, gm.z, bt.z), fmaf((acc[t][4 * g + 3] - mean) * rstd, gm.w, bt.w));
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(widen_pair(pk[0], pk[1]), rs_f, foff + (32 * t + 16 * pr) * 2, 0, 0);
                }
        }
    }

    if constexpr (QKV) {
        // ---- next layer's q, k, v: both waves normalise the row; A takes the even tiles of every pair of slabs, B the odd ones
        ln_to_frags(vec + V_GN, vec + V_BN);
        const int b = m / a_T, tok = m - b * a_T;
        const int qkv_bytes = (int)((long)a_M * row_bytes_f16);
        const unsigned tok_off = m < a_M ? ((unsigned)b * a_H * a_T + tok) * 128 + 16 * h : 0x80000000u;
        const unsigned head_stride = (unsigned)a_T * 128;
        const int t_own = role_b ? 1 : 0;
        auto qkv_part = [&](auto wc, void* base) __attribute__((always_inline)) {
            constexpr int W = CI(wc);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, qkv_bytes, 0x00020000);
            const float sc = W == 0 ? a_qs : 1.0f;
            f32x16 cp = zero;
            int hp = 0;
            u32x2 pk[4];
            auto epi = [&](auto fc) __attribute__((always_inline)) { // epilogue of the previous own tile in slots 4..15
                constexpr int f = CI(fc);
                if constexpr (f >= 4 && f < 8) {
                    constexpr int g = f - 4;
                    pk[g] = pack4(cp[4 * g + 0] * sc, cp[4 * g + 1] * sc, cp[4 * g + 2] * sc, cp[4 * g + 3] * sc);
                }
                if constexpr (f == 10 || f == 14) {
                    constexpr int pr = (f - 10) / 4;
                    const unsigned off = tok_off + (unsigned)(hp >> 1) * head_stride + (hp & 1) * 64 + 32 * pr;
                    __builtin_amdgcn_raw_buffer_store_b128(widen_pair(pk[2 * pr], pk[2 * pr + 1]), rs, off, 0, 0);
                }
            };
#pragma unroll 1
            for (int j = 0; j < NT / 2; ++j) {
                const int hv = 2 * j + t_own;
                f32x16 c = vec_tile(vec + V_BQKV + 32 * (W * NT + hv));
                step_open();
                auto side = [&](auto fc) __attribute__((always_inline)) { if (j > 0) epi(fc); };
                if (role_b) run_slab(cury, chain(c), side);
                else run_slab(curx, chain(c), side);
                cp = c; hp = hv;
            }
            static_for<KS>(epi);
        };
        qkv_part(std::integral_constant<int, 0>{}, a_q);
        qkv_part(std::integral_constant<int, 1>{}, a_k);
        qkv_part(std::integral_constant<int, 2>{}, a_v);
    }
}

template <bool MLP, bool QKV, bool TAP>
int launch_block2(const vx_dino_block_args& a, hipStream_t s) {
    auto kern = dino_block2_kernel<MLP, QKV, TAP>;
    VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SMEM_BYTES));
    hipLaunchKernelGGL(kern, dim3((a.M + 127) / 128), dim3(512), SMEM_BYTES, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

} // namespace

extern "C" int vx_dino_block2_f16(const vx_dino_block_args* args, void* stream) {
    const vx_dino_block_args& a = *args;
    VX_REQUIRE(a.M > 0 && a.x, "vx_dino_block2_f16: empty problem");
    const bool mlp = a.att != nullptr, qkv = a.q != nullptr, tap = a.feat != nullptr;
    VX_REQUIRE(mlp || qkv, "vx_dino_block2_f16: nothing to do (neither att nor q given)");
    if (mlp) VX_REQUIRE(a.w_mlp && a.vec_mlp, "vx_dino_block2_f16: the MLP half needs w_mlp and vec_mlp");
    if (qkv) VX_REQUIRE(a.w_qkv && a.vec_qkv && a.k && a.v && a.T > 0 && a.H > 0 && a.M % a.T == 0, "vx_dino_block2_f16: the QKV half needs w_qkv, vec_qkv, k, v, T, H and M %% T == 0");
    if (tap) VX_REQUIRE(a.vec_tap, "vx_dino_block2_f16: the tap needs vec_tap");
    hipStream_t s = as_stream(stream);
    if (mlp && qkv && tap) return launch_block2<true, true, true>(a, s);
    if (mlp && qkv) return launch_block2<true, true, false>(a, s);
    if (mlp && tap) return launch_block2<true, false, true>(a, s);
    if (mlp) return launch_block2<true, false, false>(a, s);
    VX_REQUIRE(!tap, "vx_dino_block2_f16: a tap without the MLP half is not built");
    return launch_block2<false, true, false>(a, s);
}
