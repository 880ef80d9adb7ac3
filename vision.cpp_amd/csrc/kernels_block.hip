// Token-stationary DINOv2 block kernel for gfx950 (embed dim 384, MLP 1536, head dim 64).
//
// One launch replaces, for 128 token rows per workgroup, the reference's op chain between two attentions
// (src/visp/arch/dino.cpp:48-90 + :59-66 of the next layer):
//     x  += lambda1 * (att @ Wo^T + bo)                     self_attention output dense + layer_scale + residual
//     x  += lambda2 * (gelu(LN2(x) @ W1^T + b1) @ W2^T + b2)  mlp + layer_scale + residual
//     [feat = LN_final(x)]                                   get_intermediate_layers tap (dino.cpp:100-107)
//     [q,k,v = LN1'(x) @ Wqkv'^T + b']                       next layer's norm1 + query/key/value
// (the first layer's LN1 + QKV runs as the QKV-only instance of the same kernel).
//
// Structure: 4 waves per workgroup, ONE per SIMD (up to 512 VGPR + AGPR each). A wave owns 32 token rows for the whole
// launch and keeps everything that belongs to them in registers:
//   * every product is computed transposed, D^T[feature, token] = W[feature, k] * X^T[k, token], on
//     v_mfma_f32_32x32x16_f16 with the WEIGHT fragment as the A operand and the token fragment as the B operand, so a
//     lane owns one token column and 16 feature rows per 32-feature tile;
//   * such an accumulator tile IS the B operand of the next product (rows = the next reduction index): LayerNorm and
//     GELU run on the accumulators in place and their f16 results are used as fragments without any LDS or lane
//     traffic. The k order inside a 16-step is then permuted (element j of lane half h = feature 16s + 8(j>>2) + 4h +
//     (j&3)); the host packs the weights of every product that consumes such fragments in the same order;
//   * the 192 fc2 accumulators (384 features x 32 tokens per wave) stay live over the whole hidden loop, the hidden
//     activations only ever exist as one 32 x 32 tile per wave.
// Only weights move: the host packs them as a stream of 24 KiB slabs (the 24 A fragments of one tile, 1 KiB each, in
// the exact order of use); the four waves copy a PAIR of slabs with plain 16-byte loads one step ahead (registers), write it
// into the free half of a 4-stage LDS ring during the step, and every wave reads all fragments with conflict-free
// ds_read_b128. One workgroup barrier per pair of slabs (= per 48 MFMAs per wave). All waits are the compiler's counted
// waits: no LDS-DMA here, because one wave per SIMD cannot hide the issue cost of LDS-DMA pieces behind another wave's MFMAs
// (kernels_block16.hip, the default since round 2, has two waves per SIMD and does use it).
#include "vx_common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int D = 384;                  // embed dim
constexpr int HID = 1536;               // mlp hidden
constexpr int NT = D / 32;              // 12 feature tiles of the embed dim
constexpr int KS = D / 16;              // 24 k-steps over the embed dim
constexpr int SLAB = KS * 1024;         // bytes: 24 fragments x (64 lanes x 16 B)
constexpr int PIECES = KS / 4;          // 1 KiB pieces per wave per slab
constexpr int N_OUT = NT;               // slabs of the output projection
constexpr int N_MLP = 2 * (HID / 32);   // fc1 + fc2 slabs
constexpr int N_QKV = 3 * NT;

// per-feature f32 vectors, copied to LDS once per workgroup
constexpr int V_BO = 0, V_LAM1 = 384, V_G2 = 768, V_B2 = 1152, V_B1 = 1536, V_BFC2 = 3072, V_LAM2 = 3456; // MLP part: 3840 floats
constexpr int V_GN = 3840, V_BN = 4224, V_BQKV = 4608;                                                     // QKV part: 1920 floats
constexpr int V_GF = 5760, V_BF = 6144;                                                                    // tap part: 768 floats
constexpr int V_TOTAL = 6528;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gelu_tanh(float x) { // ggml_gelu = x * sigmoid(2u), as kernels_gemm.hip
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    const float c3 = c1 * 0.044715f;
    const float w = fmaf(x * x, c3, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * w));
}

// value held by the same lane of the other 32-lane half (lanes l and l+32 own the two halves of one token row)
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// Two packed f16x4 register pairs (feature groups g and g+1 of one tile: this lane's 4 + 4 features) -> 16 contiguous
// bytes per lane: the low half ends up with features 8g..8g+7 of its token, the high half with 8g+8..8g+15.
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

// ---- the schedule ---------------------------------------------------------------------------------------------------
// With one wave per SIMD nothing hides a stall, so the instruction stream is laid out by hand as SLOTS: one MFMA, the
// ds_read that refills the fragment window PF fragments ahead (crossing into the next slab, which the 4-stage ring
// made visible one barrier earlier), at most one piece of the slab feed (ds_write of slab k+2 / global load of slab
// k+4) and a few VALU instructions of "side work" (the GELU of the previous hidden tile, the epilogue of the previous
// output tile). __builtin_amdgcn_sched_barrier(0) closes every slot: left alone, hipcc issues ds_read -> wait ->
// MFMA with a single fragment register set and puts all VALU work after the MFMAs.
constexpr int PF = 4;
constexpr int SMEM_RING = 4 * SLAB;
constexpr int SMEM_BYTES = SMEM_RING + V_TOTAL * 4;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>). The slot schedule below is written with
// `if constexpr` on these indices; a plain unrolled loop left thousands of unfolded branches and put the register arrays in scratch.
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#define CI(x) (decltype(x)::value)

template <bool MLP, bool QKV, bool TAP, int DBG = 0> // DBG: diagnostic builds only (tools/bench_block.py): 1 no global weight loads, 2 no ring writes, 4 no MFMA, 8 no fragment reads
__global__ __launch_bounds__(256, 1) void dino_block_kernel(const vx_dino_block_args args) {
    // every field in a local: a lambda that captured the argument struct by reference made hipcc spill it to scratch and
    // address global memory through flat pointers
    float* const a_x = args.x; const void* const a_att = args.att; const void* const a_wmlp = args.w_mlp; const void* const a_wqkv = args.w_qkv;
    const float* const a_vmlp = args.vec_mlp; const float* const a_vqkv = args.vec_qkv; const float* const a_vtap = args.vec_tap;
    void* const a_feat = args.feat; void* const a_q = args.q; void* const a_k = args.k; void* const a_v = args.v; float* const a_cap = args.cap_x1;
    const int a_M = args.M, a_T = args.T, a_H = args.H; const float a_qs = args.q_scale, a_eps = args.eps;
    unsigned long long* const a_stamps = reinterpret_cast<unsigned long long*>(args.stamps);
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (a_stamps && threadIdx.x == 0) a_stamps[(size_t)blockIdx.x * 16 + slot] = slot == 15 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (a_stamps && threadIdx.x == 0) a_stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    float* const vec = reinterpret_cast<float*>(smem + SMEM_RING);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + wave * 32 + r;

    constexpr int n_mlp_slabs = MLP ? N_OUT + N_MLP : 0;
    constexpr int n_slabs = n_mlp_slabs + (QKV ? N_QKV : 0);

    // Rows are addressed through buffer descriptors: loads of rows >= M return zeros and their stores are dropped by the
    // range check, so the tail workgroup needs neither branches nor padded allocations.
    const unsigned row_bytes_f32 = D * 4, row_bytes_f16 = D * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a_x, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
    const unsigned xoff = (unsigned)m * row_bytes_f32 + 16 * h; // + (32 t + 8 g) * 4
    auto ld_x = [&](int t, int g) __attribute__((always_inline)) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff + (32 * t + 8 * g) * 4, 0, 0));
    };
    auto st_x = [&](int t, int g, f32x4 v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x, xoff + (32 * t + 8 * g) * 4, 0, 0);
    };

    // ---- weight slab stream: global -> registers (one step ahead) -> LDS ring (4 stages: the pair in use, the pair being
    // written) -> fragments. A STEP consumes two slabs behind one barrier as two interleaved MFMA streams X (even slab) and
    // Y (odd slab): two dependent MFMAs with VALU work between them run at ~64 cycles per MFMA (measured), so a tile's
    // 24-MFMA accumulation chain alternates with an independent stream (another tile's chain, or fc2's 12 accumulators).
    const u32x4* const src_mlp = reinterpret_cast<const u32x4*>(a_wmlp) + wave * PIECES * 64 + lane;
    const u32x4* const src_qkv = reinterpret_cast<const u32x4*>(a_wqkv) + wave * PIECES * 64 + lane;
    auto slab_src = [&](int k) __attribute__((always_inline)) -> const u32x4* {
        const int kk = k < n_slabs ? k : n_slabs - 1; // past the end: a harmless re-read instead of a branch
        if constexpr (MLP && QKV) return kk < n_mlp_slabs ? src_mlp + (long)kk * (SLAB / 16) : src_qkv + (long)(kk - n_mlp_slabs) * (SLAB / 16);
        else if constexpr (MLP) return src_mlp + (long)kk * (SLAB / 16);
        else return src_qkv + (long)kk * (SLAB / 16);
    };
    u32x4 G0[PIECES], G1[PIECES]; // staged pieces of the next even / odd slab
    unsigned char* const wr0 = ring + wave * PIECES * 1024 + lane * 16; // this lane's first piece inside stage 0
    const unsigned char* const rd0 = ring + lane * 16;                  // this lane's 16 bytes of fragment 0 inside stage 0

    {
        const u32x4 *s0 = slab_src(0), *s1 = slab_src(1);
#pragma unroll
        for (int z = 0; z < PIECES; ++z) { G0[z] = s0[z * 64]; G1[z] = s1[z * 64]; }
        if constexpr (MLP)
            for (int i = tid; i < 3840 / 4; i += 256) reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(a_vmlp)[i];
        if constexpr (QKV)
            for (int i = tid; i < 1920 / 4; i += 256) reinterpret_cast<float4*>(vec + V_GN)[i] = reinterpret_cast<const float4*>(a_vqkv)[i];
        if constexpr (TAP)
            for (int i = tid; i < 768 / 4; i += 256) reinterpret_cast<float4*>(vec + V_GF)[i] = reinterpret_cast<const float4*>(a_vtap)[i];
#pragma unroll
        for (int z = 0; z < PIECES; ++z) {
            *reinterpret_cast<u32x4*>(wr0 + z * 1024) = G0[z];
            *reinterpret_cast<u32x4*>(wr0 + SLAB + z * 1024) = G1[z];
        }
        const u32x4 *s2 = slab_src(2), *s3 = slab_src(3);
#pragma unroll
        for (int z = 0; z < PIECES; ++z) { G0[z] = s2[z * 64]; G1[z] = s3[z * 64]; }
    }

    int k = 0, st = 0;            // first slab of the step in flight and its ring stage (0 or 2)
    const unsigned char *curx = rd0, *cury = rd0 + SLAB;
    unsigned char* wr = wr0 + 2 * SLAB;
    const u32x4 *gsrc0 = slab_src(4), *gsrc1 = slab_src(5);
    f16x8 wfx[PF], wfy[PF];       // fragment windows of the two streams: fragment f lives in w?[f % PF]

    // step (k, k+1) starts: every wave is done with the previous pair (its stages become the write target), this pair is visible
    auto step_open = [&]() __attribute__((always_inline)) {
        if constexpr (!(DBG & 64)) __syncthreads();
        curx = rd0 + st * SLAB;
        cury = curx + SLAB;
        wr = wr0 + (st ^ 2) * SLAB;
        gsrc0 = slab_src(k + 4);
        gsrc1 = slab_src(k + 5);
        st ^= 2;
        k += 2;
        if constexpr (!(DBG & 8)) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                wfx[i] = *reinterpret_cast<const f16x8*>(curx + i * 1024);
                wfy[i] = *reinterpret_cast<const f16x8*>(cury + i * 1024);
            }
        }
    };
    // 48 slots: slot i runs fragment i/2 of stream X (i even) or Y (i odd), refills that window, moves one twelfth of the
    // next pair from registers to LDS or requests one twelfth of the pair after it, and runs the slot's side work
    auto step = [&](auto&& xm, auto&& ym, auto&& side) __attribute__((always_inline)) {
        step_open();
        static_for<2 * KS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = CI(ic), f = i >> 1;
            if constexpr ((i & 1) == 0) {
                xm(std::integral_constant<int, f>{}, wfx[f % PF]);
                if constexpr (f + PF < KS && !(DBG & 8)) wfx[f % PF] = *reinterpret_cast<const f16x8*>(curx + (f + PF) * 1024);
            } else {
                ym(std::integral_constant<int, f>{}, wfy[f % PF]);
                if constexpr (f + PF < KS && !(DBG & 8)) wfy[f % PF] = *reinterpret_cast<const f16x8*>(cury + (f + PF) * 1024);
            }
            constexpr int z = i >> 2; // piece 0..11 of the pair: 0..5 even slab, 6..11 odd slab
            if constexpr (i % 4 == 1 && !(DBG & 2)) {
                if constexpr (z < PIECES) *reinterpret_cast<u32x4*>(wr + z * 1024) = G0[z];
                else *reinterpret_cast<u32x4*>(wr + SLAB + (z - PIECES) * 1024) = G1[z - PIECES];
            }
            if constexpr (i % 4 == 3 && !(DBG & 1)) {
                if constexpr (z < PIECES) G0[z] = gsrc0[z * 64];
                else G1[z - PIECES] = gsrc1[(z - PIECES) * 64];
            }
            side(ic);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_side = [](auto) {};

    __syncthreads(); // slabs 0 and 1 and the vectors are in LDS
    stamp(1);

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // accumulator tile initialised with a per-feature vector (bias folded into the MFMA chain): 4 LDS reads, no VALU
    auto vec_tile = [&](const float* v) __attribute__((always_inline)) -> f32x16 {
        f32x16 c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 q = *reinterpret_cast<const float4*>(v + 8 * g + 4 * h);
            c[4 * g + 0] = q.x; c[4 * g + 1] = q.y; c[4 * g + 2] = q.z; c[4 * g + 3] = q.w;
        }
        return c;
    };
    // a finished tile is copied out of the accumulator registers in one burst (hipcc keeps MFMA results in AGPRs; the asm
    // pins the copy, so the VALU code of the following slots works on VGPRs)
    auto to_vgprs = [&](const f32x16& src, float(&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { dst[e] = src[e]; asm volatile("" : "+v"(dst[e])); }
    };

    f16x8 xb[KS];    // token fragments (B operand) of the product in flight
    f32x16 acc[NT];  // this lane's half of 32 token rows: row-resident residual stream / fc2 accumulators
    auto chain = [&](f32x16& c) __attribute__((always_inline)) { // a tile's accumulation chain over the embed dim as a stream
        return [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
            if constexpr (!(DBG & 4)) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xb[CI(fc)], c, 0, 0, 0);
        };
    };

    // LayerNorm over the row held by lanes (r, 0) and (r, 1); two passes in registers (nn.cpp:14-19)
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
    // normalised row as the next product's B fragments: register e = 4g + i of tile t is element (g&1)*4 + i of k-step 2t + (g>>1)
    auto ln_to_frags = [&](const float* gamma, const float* beta) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + 32 * t + 8 * g + 4 * h);
                const float4 bt = *reinterpret_cast<const float4*>(beta + 32 * t + 8 * g + 4 * h);
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xb[2 * t + (g >> 1)][(g & 1) * 4 + i] = (f16)fmaf((acc[t][4 * g + i] - mean) * rstd, gg[i], bb[i]);
            }
    };

    if constexpr (MLP) {
        // ---- x += lambda1 * (att Wo^T + bo)   (dino.cpp:59-74 output dense, :80-83)
        {
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a_att), 0, (int)((long)a_M * row_bytes_f16), 0x00020000);
            const unsigned aoff = (unsigned)m * row_bytes_f16 + 16 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) xb[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff + 32 * s, 0, 0));
        }
        stamp(2);
        // the residual stream rows go straight into acc (the youngest loads in the queue: nothing waits for them before the
        // first epilogue, one step of MFMAs later)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = ld_x(t, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = v[i];
            }
        // step j: tiles 2j (stream X) and 2j+1 (stream Y); side work = the residual update of tiles 2j-2 and 2j-1
        f32x16 cpa = zero, cpb = zero; // finished pair awaiting its epilogue
        float ct[16];
        auto resid_group = [&](int t, int g) __attribute__((always_inline)) {
            const float4 lm = *reinterpret_cast<const float4*>(vec + V_LAM1 + 32 * t + 8 * g + 4 * h);
            f32x4 o = {fmaf(ct[4 * g + 0], lm.x, acc[t][4 * g + 0]), fmaf(ct[4 * g + 1], lm.y, acc[t][4 * g + 1]),
                       fmaf(ct[4 * g + 2], lm.z, acc[t][4 * g + 2]), fmaf(ct[4 * g + 3], lm.w, acc[t][4 * g + 3])};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = o[i];
            st_x(t, g, o); // re-read by the fc2 epilogue below
        };
        auto resid_side = [&](auto t0c, auto ic) __attribute__((always_inline)) { // tile t0 in slots 6..21, tile t0+1 in slots 26..41
            constexpr int t0 = CI(t0c), i = CI(ic);
            if constexpr (i == 6) to_vgprs(cpa, ct);
            if constexpr (i >= 8 && i < 24 && i % 4 == 0) resid_group(t0, i / 4 - 2);
            if constexpr (i == 26) to_vgprs(cpb, ct);
            if constexpr (i >= 28 && i < 44 && i % 4 == 0) resid_group(t0 + 1, i / 4 - 7);
        };
        static_for<NT / 2>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = CI(jc);
            f32x16 ca = vec_tile(vec + V_BO + 32 * (2 * j)), cb = vec_tile(vec + V_BO + 32 * (2 * j + 1));
            step(chain(ca), chain(cb), [&](auto ic) __attribute__((always_inline)) {
                if constexpr (j > 0) resid_side(std::integral_constant<int, (j > 0 ? 2 * j - 2 : 0)>{}, ic);
            });
            cpa = ca; cpb = cb;
        });
        static_for<2 * KS>([&](auto ic) __attribute__((always_inline)) { resid_side(std::integral_constant<int, NT - 2>{}, ic); });
        if (a_cap) { // parity captures only (tests): the residual stream after the attention half
            const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a_cap, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, xoff + (32 * t + 8 * g) * 4, 0, 0);
                }
        }

        // ---- mlp (dino.cpp:52-57). Hidden tile u (32 units) = gelu(W1[u] LN2(x)^T + b1[u]) goes from the fc1 accumulators
        // straight into fc2's B operand. Software pipeline over u, one step each:
        //     stream X: fc1(u+1) (slab W1(u+1)) | stream Y: fc2(u-1) (slab W2(u-1)) | side work: GELU(u)
        // Slab order: [W1(0), W1(1)], [W1(u+1), W2(u-1)] for u = 1..46, [W2(46), W2(47)].
        stamp(3);
        ln_stats();
        ln_to_frags(vec + V_G2, vec + V_B2);
        stamp(4);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = vec_tile(vec + V_BFC2 + 32 * t); // fc2 bias folded into the accumulators

        const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f, c3 = c1 * 0.044715f;
        f32x16 hc, hn;        // fc1 tile being activated / being accumulated
        f16x8 hbp[2], hbn[2]; // activated tile feeding fc2 / being produced
        float gt[16], g1[16], g2[16];
        // GELU (ggml_gelu: x * sigmoid(2u), see kernels_gemm.hip) of the copied tile, 7 dependent operations per element.
        // Dependent VALU instructions issued back to back cost ~8 cycles each beside this wave's MFMAs (measured), so an
        // element's operations sit in 7 CONSECUTIVE SLOTS and a slot mixes operations of up to three different elements:
        // element e starts in slot 1 + (5 e) / 2.
        auto gelu_op = [&](auto ec, auto opc) __attribute__((always_inline)) {
            constexpr int e = CI(ec), op = CI(opc);
            if constexpr (DBG & 32) { if constexpr (op == 6) hbn[e >> 3][e & 7] = (f16)gt[e]; return; }
            if constexpr (op == 0) g1[e] = gt[e] * gt[e];
            if constexpr (op == 1) g1[e] = fmaf(g1[e], c3, c1);
            if constexpr (op == 2) g1[e] = gt[e] * g1[e];
            if constexpr (op == 3) g2[e] = __builtin_amdgcn_exp2f(g1[e]);
            if constexpr (op == 4) g2[e] = 1.0f + g2[e];
            if constexpr (op == 5) g2[e] = __builtin_amdgcn_rcpf(g2[e]);
            if constexpr (op == 6) hbn[e >> 3][e & 7] = (f16)(gt[e] * g2[e]);
        };
        auto gelu_side = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = CI(ic);
            if constexpr (i == 0) to_vgprs(hc, gt);
            static_for<16>([&](auto ec) __attribute__((always_inline)) {
                constexpr int op = i - (1 + (5 * CI(ec)) / 2);
                if constexpr (op >= 0 && op < 7) gelu_op(ec, std::integral_constant<int, (op >= 0 && op < 7 ? op : 0)>{});
            });
        };
        auto fc2_stream = [&](const f16x8(&hb)[2]) __attribute__((always_inline)) { // fragment f = (k-step f / 12 of the hidden tile, feature tile f % 12)
            return [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
                constexpr int f = CI(fc);
                if constexpr (!(DBG & 4)) acc[f % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, hb[f / NT], acc[f % NT], 0, 0, 0);
            };
        };
        hc = vec_tile(vec + V_B1);
        hn = vec_tile(vec + V_B1 + 32);
        step(chain(hc), chain(hn), no_side);                                    // [W1(0), W1(1)]
        static_for<2 * KS>(gelu_side);                                          // GELU(0), no cover
        hbp[0] = hbn[0]; hbp[1] = hbn[1];
        hc = hn;
#pragma unroll 1
        for (int u = 1; u < HID / 32 - 1; ++u) {
            hn = vec_tile(vec + V_B1 + 32 * (u + 1));
            step(chain(hn), fc2_stream(hbp), gelu_side);                        // [W1(u+1), W2(u-1)] | GELU(u)
            hbp[0] = hbn[0]; hbp[1] = hbn[1];
            hc = hn;
        }
        // hc = fc1(47), hbp = gelu(46)
        {
            f16x8 hb46[2] = {hbp[0], hbp[1]};
            static_for<2 * KS>(gelu_side);                                      // GELU(47), no cover
            step(fc2_stream(hb46), fc2_stream(hbn), no_side);                   // [W2(46), W2(47)]
        }

        stamp(5);
        // ---- x += lambda2 * (fc2 + b2)   (dino.cpp:85-87). All 48 re-reads of x are issued before the first use: one memory
        // round trip for the whole epilogue (loads interleaved with the stores waited for every store's acknowledgement)
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
    } else {
        // QKV-only instance (first layer): the residual stream comes from memory
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = ld_x(t, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = v[i];
            }
    }

    stamp(6);
    if constexpr (TAP || QKV) ln_stats(); // both LayerNorms below normalise the same row: shared statistics
    stamp(7);

    if constexpr (TAP) {
        // ---- get_intermediate_layers: feat = LN_final(x), f16 rows (dino.cpp:100-107)
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

    if constexpr (QKV) {
        // ---- next layer: q, k, v = LN1(x) Wqkv^T + b, head-major [B, H, T, 64], q pre-scaled (dino.cpp:59-66, nn.cpp:210-216)
        stamp(8);
        ln_to_frags(vec + V_GN, vec + V_BN);
        stamp(9);
        const int b = m / a_T, tok = m - b * a_T;
        const int qkv_bytes = (int)((long)a_M * row_bytes_f16); // each of q, k, v: [B, H, T, 64] f16 = M * 384 * 2 bytes
        // rows past M get an offset beyond any buffer (and far from wrapping): the range check drops their stores
        const unsigned tok_off = m < a_M ? ((unsigned)b * a_H * a_T + tok) * 128 + 16 * h : 0x80000000u;
        const unsigned head_stride = (unsigned)a_T * 128;
        // One part per output tensor (W = 0 q, 1 k, 2 v: its descriptor and scale are compile-time choices), 12 tiles = 6 steps
        // each. The epilogue of a pair of tiles (scale, round to f16, pair the lane halves into 16-byte stores) runs in the
        // slots of the next step; the last pair of a part is flushed without cover.
        auto qkv_part = [&](auto wc, void* base) __attribute__((always_inline)) {
            constexpr int W = decltype(wc)::value;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, qkv_bytes, 0x00020000);
            const float sc = W == 0 ? a_qs : 1.0f;
            f32x16 cpa = zero, cpb = zero;
            int hp = 0; // first tile (inside the part) of the finished pair
            u32x2 pk[4];
            float ct[16];
            auto epi_tile = [&](auto i0c, auto ic, const f32x16& c, int hv) __attribute__((always_inline)) { // slots i0 .. i0+12
                constexpr int i0 = CI(i0c), i = CI(ic);
                if constexpr (i == i0) to_vgprs(c, ct);
                if constexpr (i >= i0 + 2 && i < i0 + 6) {
                    constexpr int g = i - i0 - 2;
                    pk[g] = pack4(ct[4 * g + 0] * sc, ct[4 * g + 1] * sc, ct[4 * g + 2] * sc, ct[4 * g + 3] * sc);
                }
                if constexpr (i == i0 + 8 || i == i0 + 12) {
                    constexpr int pr = (i - i0 - 8) / 4;
                    const unsigned off = tok_off + (unsigned)(hv >> 1) * head_stride + (hv & 1) * 64 + 32 * pr;
                    __builtin_amdgcn_raw_buffer_store_b128(widen_pair(pk[2 * pr], pk[2 * pr + 1]), rs, off, 0, 0);
                }
            };
            auto epi_side = [&](auto ic) __attribute__((always_inline)) {
                epi_tile(std::integral_constant<int, 4>{}, ic, cpa, hp);
                epi_tile(std::integral_constant<int, 26>{}, ic, cpb, hp + 1);
            };
            {
                f32x16 ca = vec_tile(vec + V_BQKV + 32 * (W * NT)), cb = vec_tile(vec + V_BQKV + 32 * (W * NT + 1));
                step(chain(ca), chain(cb), no_side);
                cpa = ca; cpb = cb; hp = 0;
            }
#pragma unroll 1
            for (int i2 = 2; i2 < NT; i2 += 2) {
                f32x16 ca = vec_tile(vec + V_BQKV + 32 * (W * NT + i2)), cb = vec_tile(vec + V_BQKV + 32 * (W * NT + i2 + 1));
                step(chain(ca), chain(cb), epi_side);
                cpa = ca; cpb = cb; hp = i2;
            }
            static_for<2 * KS>(epi_side);
        };
        qkv_part(std::integral_constant<int, 0>{}, a_q);
        qkv_part(std::integral_constant<int, 1>{}, a_k);
        qkv_part(std::integral_constant<int, 2>{}, a_v);
    }
    stamp(10);
    stamp(15);
}

template <bool MLP, bool QKV, bool TAP, int DBG = 0>
int launch_block(const vx_dino_block_args& a, hipStream_t s) {
    auto kern = dino_block_kernel<MLP, QKV, TAP, DBG>;
    VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SMEM_BYTES));
    hipLaunchKernelGGL(kern, dim3((a.M + 127) / 128), dim3(256), SMEM_BYTES, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

// A-operand fragment of 32 rows x 16 k of a row-major f16 matrix: lane l holds row l&31; element j is
// k = 16s + 8(l>>5) + j (natural) or 16s + 8(j>>2) + 4(l>>5) + (j&3) (the order of an accumulator tile reused as B operand)
void pack_frag(const uint16_t* w, long ld, int row0, int s, bool permuted, uint16_t* dst) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int hh = l >> 5;
            const int k = permuted ? 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) : 16 * s + 8 * hh + j;
            dst[l * 8 + j] = w[(long)(row0 + (l & 31)) * ld + k];
        }
}

} // namespace

extern "C" {

int vx_dino_block_supported(int embed_dim, int hidden, int head_dim) { return embed_dim == D && hidden == HID && head_dim == 64; }
size_t vx_dino_block_mlp_bytes(void) { return (size_t)(N_OUT + N_MLP) * SLAB; }
size_t vx_dino_block_qkv_bytes(void) { return (size_t)N_QKV * SLAB; }

// host code: row-major f16 weights -> the slab streams the kernel consumes (see the header)
int vx_dino_block_pack_mlp(const void* wo, const void* w1, const void* w2, void* out) {
    VX_REQUIRE(wo && w1 && w2 && out, "vx_dino_block_pack_mlp: null pointer");
    const uint16_t *o = static_cast<const uint16_t*>(wo), *a = static_cast<const uint16_t*>(w1), *b = static_cast<const uint16_t*>(w2);
    uint16_t* dst = static_cast<uint16_t*>(out);
    auto slab = [&](int k) { return dst + (size_t)k * (SLAB / 2); };
    for (int t = 0; t < NT; ++t)
        for (int s = 0; s < KS; ++s) pack_frag(o, D, 32 * t, s, false, slab(t) + s * 512);
    auto fc1 = [&](int u, uint16_t* sl) {
        for (int s = 0; s < KS; ++s) pack_frag(a, D, 32 * u, s, true, sl + s * 512);
    };
    auto fc2 = [&](int u, uint16_t* sl) {
        for (int s2 = 0; s2 < 2; ++s2)
            for (int t = 0; t < NT; ++t) pack_frag(b, HID, 32 * t, 2 * u + s2, true, sl + (s2 * NT + t) * 512);
    };
    // consumption order of the kernel's software pipeline: W1(0), W1(1), [W1(u+1), W2(u-1)] for u = 1..46, W2(46), W2(47)
    constexpr int NU = HID / 32;
    int k = N_OUT;
    fc1(0, slab(k++));
    fc1(1, slab(k++));
    for (int u = 1; u < NU - 1; ++u) {
        fc1(u + 1, slab(k++));
        fc2(u - 1, slab(k++));
    }
    fc2(NU - 2, slab(k++));
    fc2(NU - 1, slab(k++));
    return k == N_OUT + N_MLP ? 1 : 0;
}
int vx_dino_block_pack_qkv(const void* wqkv, void* out) {
    VX_REQUIRE(wqkv && out, "vx_dino_block_pack_qkv: null pointer");
    uint16_t* dst = static_cast<uint16_t*>(out);
    for (int v = 0; v < N_QKV; ++v)
        for (int s = 0; s < KS; ++s) pack_frag(static_cast<const uint16_t*>(wqkv), D, 32 * v, s, true, dst + (size_t)v * (SLAB / 2) + s * 512);
    return 1;
}

int vx_dino_block_f16(const vx_dino_block_args* args, void* stream) {
    const vx_dino_block_args& a = *args;
    VX_REQUIRE(a.M > 0 && a.x, "vx_dino_block_f16: empty problem");
    const bool mlp = a.att != nullptr, qkv = a.q != nullptr, tap = a.feat != nullptr;
    VX_REQUIRE(mlp || qkv, "vx_dino_block_f16: nothing to do (neither att nor q given)");
    if (mlp) VX_REQUIRE(a.w_mlp && a.vec_mlp, "vx_dino_block_f16: the MLP half needs w_mlp and vec_mlp");
    if (qkv) VX_REQUIRE(a.w_qkv && a.vec_qkv && a.k && a.v && a.T > 0 && a.H > 0 && a.M % a.T == 0, "vx_dino_block_f16: the QKV half needs w_qkv, vec_qkv, k, v, T, H and M %% T == 0");
    if (tap) VX_REQUIRE(a.vec_tap, "vx_dino_block_f16: the tap needs vec_tap");
    hipStream_t s = as_stream(stream);
    if (mlp && qkv && tap) return launch_block<true, true, true>(a, s);
    if (mlp && qkv) {
#ifdef VISP_BLOCK_DIAG
        static const int dbg = getenv("VISP_BLOCK_DBG") ? atoi(getenv("VISP_BLOCK_DBG")) : 0;
        switch (dbg) {
            case 1: return launch_block<true, true, false, 1>(a, s);
            case 3: return launch_block<true, true, false, 3>(a, s);
            case 4: return launch_block<true, true, false, 4>(a, s);
            case 7: return launch_block<true, true, false, 7>(a, s);
            case 11: return launch_block<true, true, false, 11>(a, s);
            case 15: return launch_block<true, true, false, 15>(a, s);
            case 43: return launch_block<true, true, false, 43>(a, s);
            case 171: return launch_block<true, true, false, 171>(a, s);
            case 427: return launch_block<true, true, false, 427>(a, s);
            case 555: return launch_block<true, true, false, 555>(a, s);
            case 1067: return launch_block<true, true, false, 1067>(a, s);
            case 32: return launch_block<true, true, false, 32>(a, s);
            default: break;
        }
#endif
        return launch_block<true, true, false>(a, s);
    }
    if (mlp && tap) return launch_block<true, false, true>(a, s);
    if (mlp) return launch_block<true, false, false>(a, s);
    VX_REQUIRE(!tap, "vx_dino_block_f16: a tap without the MLP half is not built");
    return launch_block<false, true, false>(a, s);
}

} // extern "C"
