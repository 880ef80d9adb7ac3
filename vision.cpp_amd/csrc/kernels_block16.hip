// Token-stationary DINOv2 block kernel (embed dim 384, MLP 1536, head dim 64; argument block vx_dino_block_args in
// include/visp_hip_kernels.h), built on v_mfma_f32_16x16x32_f16: the rest of a dino::layer after attention and the next layer's
// LN1 + QKV in one launch (reference dino.cpp:48-90, 100-107). LayerScale is expected FOLDED into the residual products (rows of
// Wo / W2 and bo / b2 scaled by lambda1 / lambda2 before packing), so that the residual stream can live in the MFMA accumulators from
// the attention output to the next layer's q, k, v: x is read once and written once per launch. (The first form of this kernel --
// 32 tokens per wave, one wave per SIMD on 512 registers, 364 us against 247 -- is recorded in profiles/r02_block_kernel_anatomy.txt
// and was removed in round 4.)
//
//   * 8 waves per workgroup, TWO per SIMD (256 registers each), a wave owns 16 token rows: D^T[16 features, 16 tokens] =
//     W[16 features, 32 k] * X^T[32 k, 16 tokens]; lane l = (token n = l & 15, group g = l >> 4) holds features 4g .. 4g+3 of
//     every 16-feature tile. Two accumulator tiles are the B operand of the next product's 32-wide k-block with a permuted k
//     order (element j of group g = feature 16 (j >> 2) + 4g + (j & 3) of the block), which the host packer bakes into the
//     weights; LayerNorm statistics are a per-lane sum + two cross-group shuffles.
//   * The state of a wave is 96 accumulators (fc2 / residual rows) + 48 registers of LayerNorm fragments instead of 192 + 96,
//     so two waves share a SIMD and one wave's waits (LDS latency, barrier skew, global latency, dependent VALU) are covered
//     by the other's MFMAs -- the 32-token form's limit (one wave per SIMD: every wait is an idle SIMD).
//     The price: a weight fragment (1 KiB) feeds 16 instead of 32 tokens, i.e. one ds_read_b128 per 16-cycle MFMA -- the LDS
//     array's full read rate (256 B/clk/CU) at full MFMA rate.
//   * Weights stream as 24 KiB slabs (24 fragments) in consumption order -- 144 per launch, global -> LDS by DMA
//     (global_load_lds_dwordx4 issued from inline asm so that hipcc's vmcnt bookkeeping does not see it; the kernel places its own
//     counted waits) into a 4-stage ring = two slab PAIRS. A step consumes one pair as two interleaved MFMA streams (four
//     accumulation chains); one workgroup barrier per step (72 per launch). Outside the MLP loop the barrier in front of the next
//     pair is taken one MFMA group before the end of the step, so the last group runs under the barrier skew and the window fill.
//     Slab contents differ from the 32-token form (16-row fragments), so the kernel has its own packers (vx_dino_block16_pack_*).
//   * What was measured and dropped (slab-granular ring, four-wave workgroups, priorities, store placement): profiles/r02_block16_variants.txt.
#include "vx_common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int D = 384, HID = 1536;
constexpr int NT = D / 16;            // 24 feature tiles of the embed dim
constexpr int KB = D / 32;            // 12 k-blocks over the embed dim
constexpr int FR = 24;                // fragments per slab
constexpr int SLAB = FR * 1024;
constexpr int N_OUT = NT / 2;         // 12 slabs of the output projection (2 tiles each)
constexpr int N_MLP = 2 * (HID / 32); // fc1 slabs (2 hidden tiles each) + fc2 slabs (one 32-wide hidden block each)
constexpr int N_QKV = 3 * NT / 2;     // 36

constexpr int V_BO = 0, V_G2 = 768, V_B2 = 1152, V_B1 = 1536, V_BFC2 = 3072; // vec_mlp: bo' | (lambda1) | g2 | b2 | b1 | b2' | (lambda2); bo and b2 arrive scaled by lambda1 / lambda2, the lambda slots are not read
constexpr int V_GN = 3840, V_BN = 4224, V_BQKV = 4608;
constexpr int V_GF = 5760, V_BF = 6144;
constexpr int V_TOTAL = 6528;

#ifndef VISP_BLOCK16_PF
#define VISP_BLOCK16_PF 4
#endif
#ifndef VISP_BLOCK16_DEFER
#define VISP_BLOCK16_DEFER 1
#endif
#ifndef VISP_BLOCK16_DEFER_MLP
#define VISP_BLOCK16_DEFER_MLP 0
#endif
#ifndef VISP_BLOCK16_LN_FENCE
#define VISP_BLOCK16_LN_FENCE 1
#endif
#ifndef VISP_BLOCK16_SPREAD
#define VISP_BLOCK16_SPREAD 1
#endif
constexpr int LN_FENCE = VISP_BLOCK16_LN_FENCE; // scheduling fence after every LN_FENCE k-blocks of a LayerNorm's fragment pass (0 = none)
// SPREAD: the 6 LDS-DMA copies a wave owes per slab pair are issued one per MFMA group in groups 0 .. 5 of the step that follows the
// boundary, not as a burst behind the barrier. An LDS-DMA instruction costs its wave 60-180 issue cycles (MI355X_MICROARCH.md, 'LDS-DMA
// piece issue cost'): six in a row right after the barrier are ~600 cycles in which BOTH waves of a SIMD issue no MFMA (they pass the
// barrier together); one per group rides under the group's own MFMAs. 0: the burst (A/B builds).
constexpr int SPREAD = VISP_BLOCK16_SPREAD;
constexpr int PF = VISP_BLOCK16_PF;   // fragment window per stream (A/B builds: tools/block16_diag.sh)
constexpr int DG = VISP_BLOCK16_DEFER; // groups of a step that run after the next pair's boundary (0: every step opens with its own boundary)
static_assert(PF % 2 == 0 && FR % PF == 0 && PF - 2 * DG >= 2, "the window must hold the deferred groups and the next step's first group");
constexpr int SMEM_RING = 4 * SLAB;
constexpr int SMEM_BYTES = SMEM_RING + V_TOTAL * 4;
constexpr int WP = 2 * FR / 8;        // 1 KiB pieces of a slab pair per wave: 6

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#define CI(x) (decltype(x)::value)

// sum over the four lane groups that hold one token row (lanes n, n+16, n+32, n+48)
__device__ __forceinline__ float row_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// 16 bytes per lane global -> LDS (global_load_lds_dwordx4; LDS destination = M0 + 16 * lane), issued from inline asm so that hipcc
// does not count it: with the builtin it waits vmcnt(0) in front of every later LDS read (it cannot tell the ring stages apart).
// The kernel's own vmcnt(0) in front of the step barrier is the wait these copies need. M0 is written in the same statement.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// DBG: diagnostic builds only (-DVISP_BLOCK16_DBG=n, tools/bench_block.py): 1 no global weight loads, 2 no ring writes, 8 no fragment
// reads, 32 no GELU, 64 no step barrier, 128 no q/k/v stores, 1024 time the waits at every boundary (valid results) -- results are garbage, the launch time shows what each part of the stream costs
// 2048: a second workgroup barrier in the middle of every step (before group VISP_BLOCK16_MID); 4096 (with 2048): waves 4-7 run half a step behind
// waves 0-3 (one extra barrier at their start, one at the leaders' end) -- the timing of a staggered schedule without its ring bookkeeping (data races)
#ifndef VISP_BLOCK16_DBG
#define VISP_BLOCK16_DBG 0
#endif
constexpr int DBG = VISP_BLOCK16_DBG;
#ifndef VISP_BLOCK16_MID
#define VISP_BLOCK16_MID 5
#endif

template <bool MLP, bool QKV, bool TAP>
__global__ __launch_bounds__(512) void dino_block16_kernel(const vx_dino_block_args args) {
    // every field in a local (a lambda that captures the argument struct by reference puts it in scratch)
    float* const a_x = args.x; const void* const a_att = args.att; const void* const a_wmlp = args.w_mlp; const void* const a_wqkv = args.w_qkv;
    const float* const a_vmlp = args.vec_mlp; const float* const a_vqkv = args.vec_qkv; const float* const a_vtap = args.vec_tap;
    void* const a_feat = args.feat; void* const a_q = args.q; void* const a_k = args.k; void* const a_v = args.v; float* const a_cap = args.cap_x1;
    const int a_M = args.M, a_T = args.T, a_H = args.H;
    const float a_qs = args.q_scale, a_eps = args.eps;
    unsigned long long* const a_stamps = static_cast<unsigned long long*>(args.stamps); // diagnostics (tools/bench_block.py --stamps): s_memtime per phase, NULL in the product
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (a_stamps && threadIdx.x == 0) a_stamps[(size_t)blockIdx.x * 16 + slot] = slot >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    stamp(14);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    float* const vec = reinterpret_cast<float*>(smem + SMEM_RING);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int m = blockIdx.x * 128 + wave * 16 + n; // this lane's token row

    constexpr int row_bytes_f32 = D * 4, row_bytes_f16 = D * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a_x, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
    // rows past M: an offset beyond any buffer (and far from wrapping) -- loads return 0, stores are dropped
    const unsigned xoff = m < a_M ? (unsigned)m * row_bytes_f32 + 16 * g : 0x80000000u;
    // (the per-tile constant goes into the instruction's scalar offset: added to the lane offset it becomes 24 more live registers,
    // which hipcc spills and reloads in between the stores -- each reload a vmcnt wait behind the stores just issued)
    auto ld_x = [&](int T) __attribute__((always_inline)) -> f32x4 { // features 16T + 4g .. +3 of this token
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, 64 * T, 0));
    };
    auto st_x = [&](int T, f32x4 v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_x, xoff, 64 * T, 0);
    };

    // ---- weight stream: slab pair (k, k+1) = 48 contiguous 1 KiB pieces; wave w copies pieces 6w .. 6w+5 by LDS-DMA into the
    // ring pair that the step before last finished reading: issued right after the step barrier, waited for (vmcnt(0)) in front of
    // the next one -- a whole step of MFMAs in between, no registers, no VGPR -> LDS store path
    auto pair_src = [&](int k) __attribute__((always_inline)) -> const unsigned char* {
        constexpr int first_qkv = MLP ? N_OUT + N_MLP : 0;
        const unsigned char* base = (MLP && k < first_qkv) ? static_cast<const unsigned char*>(a_wmlp) + (size_t)k * SLAB
                                                          : static_cast<const unsigned char*>(a_wqkv) + (size_t)(k - first_qkv) * SLAB;
        return base + (wave * WP) * 1024 + lane * 16;
    };
    constexpr int n_slabs = (MLP ? N_OUT + N_MLP : 0) + (QKV ? N_QKV : 0);
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)ring;
    auto feed_pair = [&](int k, int stage) __attribute__((always_inline)) { // slabs k, k+1 -> stages stage, stage+1
        if constexpr (DBG & 1) return;
        const unsigned char* src = pair_src(k);
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + stage * SLAB + wave * WP * 1024);
#pragma unroll
        for (int z = 0; z < WP; ++z) lds_dma16(src + z * 1024, dst + z * 1024);
    };
    // SPREAD: the copies of the pair after the one a boundary opened, owed during the step that follows that boundary
    const unsigned char* owed_src = nullptr;
    unsigned owed_dst = 0;
    bool owed = false;
    auto feed_piece = [&](int z) __attribute__((always_inline)) {
        if constexpr (DBG & 1) return;
        if (owed) lds_dma16(owed_src + z * 1024, owed_dst + z * 1024);
    };
    const unsigned char* const rd0 = ring + lane * 16;
    {
        feed_pair(0, 0);
        if constexpr (MLP)
            for (int i = tid; i < 3840 / 4; i += 512) reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(a_vmlp)[i];
        if constexpr (QKV)
            for (int i = tid; i < 1920 / 4; i += 512) reinterpret_cast<float4*>(vec + V_GN)[i] = reinterpret_cast<const float4*>(a_vqkv)[i];
        if constexpr (TAP)
            for (int i = tid; i < 768 / 4; i += 512) reinterpret_cast<float4*>(vec + V_GF)[i] = reinterpret_cast<const float4*>(a_vtap)[i];
    }

    stamp(1);
    if constexpr ((DBG & 4096) != 0) { if (wave >= 4) __builtin_amdgcn_s_barrier(); }
    int k = 0, st = 0; // first slab of the step in flight and its ring stage (0 or 2)
    const unsigned char *curx = rd0, *cury = rd0 + SLAB;
    f16x8 wfx[PF] = {}, wfy[PF] = {};

    unsigned long long w_vm = 0, w_bar = 0, w_n = 0; // DBG & 1024 only
    // A BOUNDARY in front of a slab pair: this wave's copies of the pair have landed (counted vmcnt wait), everybody's have and every
    // wave is done reading the other pair (workgroup barrier), the window is (re)filled and the copies of the pair after it are requested
    // into the stages just freed. YOUNGER = vector-memory operations this wave issued AFTER the copies it waits for (output stores):
    // they retire in order behind the copies, so the counted wait leaves them in flight instead of exposing a store's acknowledgement
    // latency at every boundary (CDNA4 counts stores in vmcnt). 0 is always safe.
    auto boundary = [&](auto yc, auto nfill) __attribute__((always_inline)) {
        constexpr int YOUNGER = CI(yc);
        if constexpr (DBG & 1024) { // diagnostic: how long this wave waits for its copies / at the barrier (slots 11 .. 13 of the stamps)
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            w_vm += t1 - t0; w_bar += t2 - t1; w_n += 1;
        } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
        if constexpr (!(DBG & 64)) __syncthreads();
        }
        curx = rd0 + st * SLAB;
        cury = curx + SLAB;
        // the window fill goes first: the scalar work and the issue of the 6 copies run under its LDS latency
        if constexpr (!(DBG & 8)) {
#pragma unroll
            for (int i = 0; i < CI(nfill); ++i) {
                wfx[i] = *reinterpret_cast<const f16x8*>(curx + i * 1024);
                wfy[i] = *reinterpret_cast<const f16x8*>(cury + i * 1024);
            }
        }
        if constexpr (SPREAD) {
            owed = k + 2 < n_slabs;
            if (owed) {
                owed_src = pair_src(k + 2);
                owed_dst = __builtin_amdgcn_readfirstlane(ring_lds + (st ^ 2) * SLAB + wave * WP * 1024);
            }
        } else {
            if (k + 2 < n_slabs) feed_pair(k + 2, st ^ 2);
        }
        st ^= 2;
        k += 2;
    };
    // A step = one slab pair = 12 groups of 4 MFMAs: fragments 2g, 2g+1 of stream X and of stream Y, then their window refills, then 4
    // slots of side work. The group's first MFMA takes the fragment that was requested LAST (Y, 2g+1): LDS reads return in order, so
    // hipcc emits one lgkmcnt wait per group instead of one per MFMA (a wait is an issue slot of the wave like any other instruction).
    // COLD: the step begins with its own boundary (after a LayerNorm phase). CONT: another step follows directly -- its boundary is
    // taken DG groups before the end of this one: those groups' fragments are in registers already, so their MFMAs run under the
    // barrier skew and the LDS latency of the next pair's first fragments instead of an idle MFMA pipe; the window slots they free
    // are refilled from the new pair.
    auto stepx = [&](auto&& xm, auto&& ym, auto&& side, auto coldc, auto contc, auto yo, auto yc) __attribute__((always_inline)) {
        constexpr bool COLD = CI(coldc) != 0, CONT = CI(contc) != 0 && DG > 0;
        if constexpr (COLD || DG == 0) boundary(yo, std::integral_constant<int, PF>{});
        static_for<FR / 2>([&](auto gc) __attribute__((always_inline)) {
            constexpr int gi = CI(gc), f0 = 2 * gi, f1 = f0 + 1;
            if constexpr (CONT && gi == FR / 2 - DG) boundary(yc, std::integral_constant<int, PF - 2 * DG>{});
            if constexpr ((DBG & 2048) != 0 && gi == VISP_BLOCK16_MID) __builtin_amdgcn_s_barrier();
            ym(std::integral_constant<int, f1>{}, wfy[f1 % PF]);
            xm(std::integral_constant<int, f1>{}, wfx[f1 % PF]);
            ym(std::integral_constant<int, f0>{}, wfy[f0 % PF]);
            xm(std::integral_constant<int, f0>{}, wfx[f0 % PF]);
            // refill: PF fragments ahead in this pair; from the deferred groups on, the next pair's (curx / cury moved at the boundary)
            constexpr int r0 = f0 + PF < FR ? f0 + PF : (CONT && gi >= FR / 2 - DG ? f0 + PF - FR : -1);
            if constexpr (r0 >= 0 && !(DBG & 8)) {
                wfx[f0 % PF] = *reinterpret_cast<const f16x8*>(curx + r0 * 1024);
                wfy[f0 % PF] = *reinterpret_cast<const f16x8*>(cury + r0 * 1024);
                wfx[f1 % PF] = *reinterpret_cast<const f16x8*>(curx + (r0 + 1) * 1024);
                wfy[f1 % PF] = *reinterpret_cast<const f16x8*>(cury + (r0 + 1) * 1024);
            }
            if constexpr (SPREAD && gi < WP) feed_piece(gi);
            side(std::integral_constant<int, 2 * f0>{});
            side(std::integral_constant<int, 2 * f0 + 1>{});
            side(std::integral_constant<int, 2 * f0 + 2>{});
            side(std::integral_constant<int, 2 * f0 + 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    constexpr std::integral_constant<int, 1> yes{};
    auto no_side = [](auto) {};
    constexpr std::integral_constant<int, 0> none{};

    auto vec4 = [&](const float* v, int T) __attribute__((always_inline)) -> f32x4 { // per-feature vector entries of tile T for this lane
        const float4 q = *reinterpret_cast<const float4*>(v + 16 * T + 4 * g);
        f32x4 c = {q.x, q.y, q.z, q.w};
        return c;
    };
    auto mfma = [](const f16x8& w, const f16x8& b, f32x4 c) __attribute__((always_inline)) -> f32x4 {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, b, c, 0, 0, 0);
    };

    f32x4 acc[NT];   // residual-stream rows, then the fc2 accumulators
    f16x8 xb[KB];    // B fragments over the embed dim (attention output, then LayerNorm rows)
    // a slab of two 16-feature tiles over the embed dim, fragments ordered [k-block][tile]: accumulation chains of both tiles
    auto pair_chain = [&](f32x4 (&c)[2]) __attribute__((always_inline)) {
        return [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
            constexpr int f = CI(fc);
            c[f & 1] = mfma(w, xb[f >> 1], c[f & 1]);
        };
    };

    float mean = 0.f, rstd = 0.f;
    auto ln_stats = [&]() __attribute__((always_inline)) { // two passes in registers (nn.cpp:14-19)
        float s = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T) s += (acc[T][0] + acc[T][1]) + (acc[T][2] + acc[T][3]);
        mean = row_sum4(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = acc[T][j] - mean; q = fmaf(d, d, q); }
        rstd = __builtin_amdgcn_rsqf(fmaf(row_sum4(q), 1.0f / D, a_eps));
    };
    // normalised row as B fragments: element j of k-block kb = feature 32 kb + 16 (j >> 2) + 4g + (j & 3) = register j & 3 of tile 2 kb + (j >> 2)
    auto ln_to_frags = [&](const float* gamma, const float* beta) __attribute__((always_inline)) {
        // The vectors' LDS addresses are known long before the statistics are: left alone, hipcc issues all 48 float4 reads ahead of the
        // variance pass, runs out of registers and SPILLS THE VALUES IT JUST READ (11 + 12 16-byte scratch stores per launch and lane,
        // 73 MB of scratch writes per launch at batch 32, profiles/r02_pmc). The pointers are therefore made to depend on rstd
        // (an empty asm the optimiser cannot see through), again every LN_FENCE k-blocks: the reads stay next to their use.
        int lo = 4 * g; // (an offset, not a pointer: a pointer that went through the asm would come back generic -> flat loads)
        if constexpr (LN_FENCE) asm volatile("" : "+v"(lo) : "v"(rstd));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float4 gq = *reinterpret_cast<const float4*>(gamma + lo + 16 * (2 * kb + t2)), bq = *reinterpret_cast<const float4*>(beta + lo + 16 * (2 * kb + t2));
                const float gm[4] = {gq.x, gq.y, gq.z, gq.w}, bt[4] = {bq.x, bq.y, bq.z, bq.w};
                // as x * (rstd g) + (b - mean rstd g): written as (x - mean) * rstd * g + b, hipcc keeps the 96 differences of the
                // variance pass alive for this loop (common subexpression) and spills 56 registers around it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = rstd * gm[j];
                    xb[kb][4 * t2 + j] = (f16)fmaf(acc[2 * kb + t2][j], a, fmaf(-mean, a, bt[j]));
                }
            }
            if constexpr (LN_FENCE) {
                if (kb % LN_FENCE == LN_FENCE - 1 && kb + 1 < KB) asm volatile("" : "+v"(lo) : "v"(xb[kb])); // (the whole fragment: with one element the other seven are sunk past the fence)
            }
        }
    };

    if constexpr (MLP) {
        // ---- x += lambda1 * (att Wo^T + bo)   (dino.cpp:59-74 output dense, :80-83)
        {
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a_att), 0, (int)((long)a_M * row_bytes_f16), 0x00020000);
            const unsigned aoff = m < a_M ? (unsigned)m * row_bytes_f16 + 16 * g : 0x80000000u;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) xb[kb] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff, 64 * kb, 0));
        }
        // LayerScale is folded into the weights by the caller (Wo' = lambda1 Wo, bo' = lambda1 bo; W2', b2' likewise), so the residual
        // stream itself is the accumulator: acc = x + bo', the out-proj chains add Wo' att onto it and acc IS x1 afterwards -- no
        // epilogue arithmetic, and x1 never goes to memory (the fc2 chains continue on it).
        // Only the tiles of the first two steps are loaded up front; step j requests the tiles of step j+2 and adds bo' to those of step
        // j+1 just before the boundary in front of it (where the wait for the copies issued before these loads is due anyway): the cold
        // HBM read of the residual stream (192 KiB per workgroup, all workgroups at once) runs under the out-proj MFMAs.
#pragma unroll
        for (int T = 0; T < 8; ++T) acc[T] = ld_x(T);
        __syncthreads(); // the vectors staged by all waves in the prologue are visible (the row loads above are in flight meanwhile)
#pragma unroll
        for (int T = 0; T < 4; ++T) acc[T] = acc[T] + vec4(vec + V_BO, T);
        stamp(2);
        auto acc_chain = [&](auto t0c) __attribute__((always_inline)) { // slab of tiles t0, t0+1, fragments [k-block][tile]
            return [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
                constexpr int f = CI(fc), T = CI(t0c) + (f & 1);
                acc[T] = mfma(w, xb[f >> 1], acc[T]);
            };
        };
        static_for<N_OUT / 2>([&](auto jc) __attribute__((always_inline)) { // step j: slabs 2j, 2j+1 = tiles 4j .. 4j+3
            constexpr int j = CI(jc);
            auto x_side = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = CI(ic);
                if constexpr (i >= 2 && i < 10 && i % 2 == 0 && j + 2 < N_OUT / 2) { constexpr int T = 4 * (j + 2) + (i - 2) / 2; acc[T] = ld_x(T); }
                if constexpr (i >= 40 && i < 44 && j + 1 < N_OUT / 2) { constexpr int T = 4 * (j + 1) + (i - 40); acc[T] = acc[T] + vec4(vec + V_BO, T); }
            };
            stepx(acc_chain(std::integral_constant<int, 4 * j>{}), acc_chain(std::integral_constant<int, 4 * j + 2>{}), x_side,
                  std::integral_constant<int, j == 0>{}, std::integral_constant<int, (j + 1 < N_OUT / 2)>{}, none, none);
        });
        stamp(3);
        if (a_cap) { // parity captures only (tests): the residual stream after the attention half
            const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a_cap, 0, (int)((long)a_M * row_bytes_f32), 0x00020000);
#pragma unroll
            for (int T = 0; T < NT; ++T) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[T]), rs_c, xoff, 64 * T, 0);
        }

        // ---- mlp (dino.cpp:52-57): hidden block u (32 units = two 16-unit tiles) = gelu(W1[u] LN2(x)^T + b1[u]) goes from the fc1
        // accumulators straight into fc2's B operand. Software pipeline over u, one step each:
        //     stream X: fc1(u+1) | stream Y: fc2(u-1) | side work: GELU(u)
        // Slab order: [W1(0), W1(1)], [W1(u+1), W2(u-1)] for u = 1..46, [W2(46), W2(47)].
        ln_stats();
        ln_to_frags(vec + V_G2, vec + V_B2);
        stamp(4);
        // acc keeps x1: the fc2 chains below accumulate W2' h (lambda2 folded in) onto the residual stream directly

        const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f, c3 = c1 * 0.044715f;
        f32x4 hc[2], hn[2];   // fc1 tiles being activated / being accumulated
        f16x8 hbp, hbn;       // activated block feeding fc2 / being produced
        auto gelu_block = [&](const f32x4 (&h)[2]) __attribute__((always_inline)) -> f16x8 { // ggml_gelu = x * sigmoid(2u), as kernels_gemm.hip
            f16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = h[e >> 2][e & 3];
                const float w = fmaf(x * x, c3, c1);
                r[e] = (f16)(x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * w)));
            }
            return r;
        };
        // GELU spread over the step, on PACKED f16 pairs (v_pk_mul / v_pk_fma / v_pk_add, v_exp_f16, v_rcp_f16): the result is an f16
        // MFMA operand anyway, and the wave's issue slots are what the MLP loop runs out of. Pair q = elements 2q, 2q+1 in slots
        // 12q .. 12q+10. (GELU_F32: the f32 form, 6 slots per element, kept for A/B builds.)
#ifndef VISP_BLOCK16_GELU_F32
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 gp[4], ga[4], ge[4];
        const h2 hc1 = {(f16)c1, (f16)c1}, hc3 = {(f16)c3, (f16)c3}, one = {(f16)1.0f, (f16)1.0f}, lowest = {(f16)-65504.0f, (f16)-65504.0f};
        auto gelu_side = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = CI(ic), q = i / 12, op = i % 12;
            if constexpr (q < 4 && !(DBG & 32)) {
                if constexpr (op == 0) { const h2 v = {(f16)hc[q >> 1][2 * (q & 1)], (f16)hc[q >> 1][2 * (q & 1) + 1]}; gp[q] = v; }
                // Range: |x| > 255 makes x^2 = inf, which still ends right (x > 0: exp2(-inf) = 0 -> y = x; x < 0: rcp(inf) = 0 -> y = -0).
                // x < -65504 would convert to -inf and give -inf * 0 = NaN where gelu is 0: clamp the low side (one packed op).
                // x > 65504 is +inf in any f16 hidden map, here as in the GEMM schedule's f16 store.
                if constexpr (op == 1) gp[q] = __builtin_elementwise_max(gp[q], lowest);
                if constexpr (op == 2) ga[q] = gp[q] * gp[q];
                if constexpr (op == 3) ga[q] = ga[q] * hc3 + hc1;
                if constexpr (op == 4) ga[q] = ga[q] * gp[q];
                if constexpr (op == 5) ge[q][0] = __builtin_exp2f16(ga[q][0]);
                if constexpr (op == 6) ge[q][1] = __builtin_exp2f16(ga[q][1]);
                if constexpr (op == 7) ge[q] = ge[q] + one;
                if constexpr (op == 8) ge[q][0] = __builtin_amdgcn_rcph(ge[q][0]);
                if constexpr (op == 9) ge[q][1] = __builtin_amdgcn_rcph(ge[q][1]);
                if constexpr (op == 10) { const h2 y = gp[q] * ge[q]; hbn[2 * q] = y[0]; hbn[2 * q + 1] = y[1]; }
            }
        };
#else
        float gt[8], g1[8];
        auto gelu_side = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = CI(ic), e = i / 6, op = i % 6;
            if constexpr (e < 8 && !(DBG & 32)) {
                if constexpr (op == 0) { gt[e] = hc[e >> 2][e & 3]; g1[e] = gt[e] * gt[e]; }
                if constexpr (op == 1) g1[e] = fmaf(g1[e], c3, c1) * gt[e];
                if constexpr (op == 2) g1[e] = __builtin_amdgcn_exp2f(g1[e]);
                if constexpr (op == 3) g1[e] = 1.0f + g1[e];
                if constexpr (op == 4) g1[e] = __builtin_amdgcn_rcpf(g1[e]);
                if constexpr (op == 5) hbn[e] = (f16)(gt[e] * g1[e]);
            }
        };
#endif
        auto fc2_stream = [&](const f16x8& hb) __attribute__((always_inline)) { // fragment T = feature tile T of this hidden block
            return [&](auto fc, const f16x8& w) __attribute__((always_inline)) {
                constexpr int T = CI(fc);
                acc[T] = mfma(w, hb, acc[T]);
            };
        };
        hc[0] = vec4(vec + V_B1, 0); hc[1] = vec4(vec + V_B1, 1);
        hn[0] = vec4(vec + V_B1, 2); hn[1] = vec4(vec + V_B1, 3);
        // (in the MLP loop the early boundary costs more than it hides -- measured 161k against 154k cycles -- so every MLP step opens
        // with its own boundary: MLPC = 0)
        constexpr std::integral_constant<int, VISP_BLOCK16_DEFER_MLP != 0> mlpc{};
        constexpr std::integral_constant<int, VISP_BLOCK16_DEFER_MLP == 0> mlpo{};
        stepx(pair_chain(hc), pair_chain(hn), no_side, yes, mlpc, none, none); // [W1(0), W1(1)]
        hbp = gelu_block(hc);                                                // GELU(0), no cover
        hc[0] = hn[0]; hc[1] = hn[1];
#pragma unroll 1
        for (int u = 1; u < HID / 32 - 1; ++u) {
            hn[0] = vec4(vec + V_B1, 2 * (u + 1)); hn[1] = vec4(vec + V_B1, 2 * (u + 1) + 1);
            stepx(pair_chain(hn), fc2_stream(hbp), gelu_side, mlpo, mlpc, none, none); // [W1(u+1), W2(u-1)] | GELU(u)
            hbp = hbn;
            hc[0] = hn[0]; hc[1] = hn[1];
        }
        {   // hc = fc1(47), hbp = gelu(46)
            const f16x8 hb46 = hbp;
            hbn = gelu_block(hc);                                            // GELU(47), no cover
            stepx(fc2_stream(hb46), fc2_stream(hbn), no_side, mlpo, none, none, none); // [W2(46), W2(47)]
        }

        stamp(5);
        // ---- x2 = x1 + lambda2 (fc2 + b2) (dino.cpp:85-87) = acc + b2': the only write of the residual stream
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            acc[T] = acc[T] + vec4(vec + V_BFC2, T);
            st_x(T, acc[T]);
        }
    } else {
        // QKV-only instance (first layer): the residual stream comes from memory
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T] = ld_x(T);
        __syncthreads(); // the vectors staged by all waves in the prologue are visible
    }

    stamp(6);
    if constexpr (TAP || QKV) ln_stats(); // both LayerNorms below normalise the same row: shared statistics
    stamp(7);

    if constexpr (TAP) {
        // ---- get_intermediate_layers: feat = LN_final(x), f16 rows (dino.cpp:100-107)
        const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(a_feat, 0, (int)((long)a_M * row_bytes_f16), 0x00020000);
        const unsigned foff = m < a_M ? (unsigned)m * row_bytes_f16 + 8 * g : 0x80000000u;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const f32x4 gm = vec4(vec + V_GF, T), bt = vec4(vec + V_BF, T);
            f16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float a = rstd * gm[j]; o[j] = (f16)fmaf(acc[T][j], a, fmaf(-mean, a, bt[j])); }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs_f, foff, 32 * T, 0);
        }
    }

    if constexpr (QKV) {
        // ---- next layer: q, k, v = LN1(x) Wqkv^T + b, head-major [B, H, T, 64], q pre-scaled (dino.cpp:59-66, nn.cpp:210-216)
        stamp(8);
        ln_to_frags(vec + V_GN, vec + V_BN);
        stamp(9);
        const int b = m / a_T, tok = m - b * a_T;
        const int qkv_bytes = (int)((long)a_M * row_bytes_f16); // each of q, k, v: [B, H, T, 64] f16 = M * 384 * 2 bytes
        const unsigned tok_off = m < a_M ? ((unsigned)b * a_H * a_T + tok) * 128 + 16 * g : 0x80000000u;
        const unsigned head_stride = (unsigned)a_T * 128;
        // Everything a head needs around its 48 MFMAs is side work of the NEIGHBOURING steps, so that nothing but register renaming lies
        // between two steps: its bias tiles are read during the step before (nb), its results are converted and stored (two 16-byte
        // stores: 8 waves x 2 KiB = a few hundred cycles of the CU's store path) during the step after (raw -> pend).
        f32x4 nb[4];                 // bias tiles of the next head
        f32x4 raw[4];                // results of the head before, f32
        float raw_sc = 1.0f;
        u32x4 pend0 = {}, pend1 = {};
        unsigned pend_off = 0x80000000u;
        __amdgpu_buffer_rsrc_t pend_rs = __builtin_amdgcn_make_buffer_rsrc(a_q, 0, qkv_bytes, 0x00020000);
        auto bias4 = [&](int R, int o) __attribute__((always_inline)) -> f32x4 { // tile group R (4 tiles = one head), floats o .. o+3 of this lane's 8
            const float4 q = *reinterpret_cast<const float4*>(vec + V_BQKV + 16 * R + 8 * g + o);
            f32x4 c = {q.x, q.y, q.z, q.w};
            return c;
        };
        auto cvt4 = [&](const f32x4& c, float sc, u32x4& dst, auto halfc) __attribute__((always_inline)) { // 4 results -> half of a 16-byte store
            typedef _Float16 h2t __attribute__((ext_vector_type(2)));
            const h2t a = {(f16)(c[0] * sc), (f16)(c[1] * sc)}, bb = {(f16)(c[2] * sc), (f16)(c[3] * sc)};
            dst[2 * CI(halfc)] = __builtin_bit_cast(unsigned, a);
            dst[2 * CI(halfc) + 1] = __builtin_bit_cast(unsigned, bb);
        };
        // side work of a head's step: R = its first tile (global over q | k | v), PREV: a head before it left raw results, NEXT: one follows
        auto head_side = [&](int R, auto prevc, auto nextc) __attribute__((always_inline)) {
            return [&, R](auto ic) __attribute__((always_inline)) {
                constexpr int i = CI(ic);
                (void)R;
                if constexpr (CI(prevc) != 0) {
                    if constexpr (i == 2) cvt4(raw[0], raw_sc, pend0, none);
                    if constexpr (i == 4) cvt4(raw[1], raw_sc, pend0, yes);
                    if constexpr (i == 6) cvt4(raw[2], raw_sc, pend1, none);
                    if constexpr (i == 8) cvt4(raw[3], raw_sc, pend1, yes);
                    if constexpr (!(DBG & 128)) {
                        // (both behind the last spread copy of group 5: the boundary's counted wait then leaves exactly these two in flight)
                        if constexpr (i == (SPREAD ? 24 : 16)) __builtin_amdgcn_raw_buffer_store_b128(pend0, pend_rs, pend_off, 0, 0);
                        if constexpr (i == (SPREAD ? 38 : 32)) __builtin_amdgcn_raw_buffer_store_b128(pend1, pend_rs, pend_off, 64, 0);
                    } else { // keep the results alive so that the MFMAs stay
                        if constexpr (i == 16) asm volatile("" ::"v"(pend0));
                        if constexpr (i == 32) asm volatile("" ::"v"(pend1));
                    }
                }
                if constexpr (CI(nextc) != 0) {
                    if constexpr (i == 34) nb[0] = bias4(R + 4, 0);
                    if constexpr (i == 35) nb[1] = bias4(R + 4, 4);
                    if constexpr (i == 36) nb[2] = bias4(R + 4, 32);
                    if constexpr (i == 37) nb[3] = bias4(R + 4, 36);
                }
            };
        };
        auto qkv_part = [&](auto wc, void* base) __attribute__((always_inline)) {
            constexpr int W = decltype(wc)::value;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, qkv_bytes, 0x00020000);
            const float sc = W == 0 ? a_qs : 1.0f;
            auto head = [&](int i4, auto coldc, auto contc, auto yo, auto yc, auto pendc) __attribute__((always_inline)) { // tiles i4 .. i4+3 of this part = one head
                // vx_dino_block16_pack_qkv orders the rows of a tile pair so that this lane's 4 + 4 results are the 8 CONSECUTIVE features
                // 32 p + 8g .. + 7 of the head: one 16-byte store per pair, 64 contiguous bytes per token row
                const int R = W * NT + i4;
                f32x4 cx[2] = {nb[0], nb[1]}, cy[2] = {nb[2], nb[3]};
                // the stores of the head before go to ITS part's buffer: pend_off / pend_rs still describe it during this step
                stepx(pair_chain(cx), pair_chain(cy), head_side(R, pendc, contc), coldc, contc, yo, yc);
                raw[0] = cx[0]; raw[1] = cx[1]; raw[2] = cy[0]; raw[3] = cy[1];
                raw_sc = sc;
                pend_off = tok_off + (unsigned)(i4 >> 2) * head_stride;
                pend_rs = rs;
            };
            // YOUNGER. The first head of q opens COLD: behind the copies of its pair lie the 24 residual-stream stores of the fc2 epilogue
            // (+ 24 of the tap) (QKV-only instance: the x loads, already consumed). A boundary taken inside a head's step (for the
            // next head) follows the stores issued in that step: the 2 pending ones of the head before (none in q's first head).
            // (yo of the later heads only matters for DG == 0, where every step opens with its own boundary.)
            constexpr std::integral_constant<int, 2> two{};
            if constexpr (W == 0) {
                nb[0] = bias4(0, 0); nb[1] = bias4(0, 4); nb[2] = bias4(0, 32); nb[3] = bias4(0, 36);
                head(0, yes, yes, std::integral_constant<int, MLP ? (TAP ? 2 * NT : NT) : 0>{}, none, none);
                head(4, none, yes, none, two, yes);
#pragma unroll 1
                for (int i4 = 8; i4 < NT; i4 += 4) head(i4, none, yes, two, two, yes);
            } else if constexpr (W == 1) {
#pragma unroll 1
                for (int i4 = 0; i4 < NT; i4 += 4) head(i4, none, yes, two, two, yes);
            } else {
#pragma unroll 1
                for (int i4 = 0; i4 < NT - 4; i4 += 4) head(i4, none, yes, two, two, yes);
                head(NT - 4, none, none, two, two, yes); // the last step of the launch
            }
        };
        qkv_part(std::integral_constant<int, 0>{}, a_q);
        qkv_part(std::integral_constant<int, 1>{}, a_k);
        qkv_part(std::integral_constant<int, 2>{}, a_v);
        cvt4(raw[0], raw_sc, pend0, none); cvt4(raw[1], raw_sc, pend0, yes);
        cvt4(raw[2], raw_sc, pend1, none); cvt4(raw[3], raw_sc, pend1, yes);
        if constexpr (!(DBG & 128)) {
            __builtin_amdgcn_raw_buffer_store_b128(pend0, pend_rs, pend_off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(pend1, pend_rs, pend_off, 64, 0);
        }
    }
    if constexpr ((DBG & 4096) != 0) { if (wave < 4) __builtin_amdgcn_s_barrier(); }
    stamp(10);
    stamp(15);
    if constexpr (DBG & 1024) {
        if (a_stamps && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) { // the two waves of SIMD 0
            unsigned long long* o = a_stamps + (size_t)blockIdx.x * 16 + (wave == 0 ? 11 : 12);
            o[0] = (w_vm << 32) | (w_bar & 0xffffffffull);
            if (wave == 0) a_stamps[(size_t)blockIdx.x * 16 + 13] = w_n;
        }
    }
}

// ---- host: weight packing ------------------------------------------------------------------------------------------------
// fragment of 16 output rows x one 32-wide k-block: lane l = (row i = l & 15, group g = l >> 4), element j: k = 32 kb + 8g + j in
// natural order, or 32 kb + 16 (j >> 2) + 4g + (j & 3) when the B operand of the product is a pair of accumulator tiles
void pack_frag16(const uint16_t* w, int ld, int row0, int kb, bool permuted, uint16_t* dst) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int i = l & 15, g = l >> 4;
            const int k = permuted ? 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3) : 32 * kb + 8 * g + j;
            dst[l * 8 + j] = w[(size_t)(row0 + i) * ld + k];
        }
}
// slab of two 16-row tiles over the embed dim, fragments ordered [k-block][tile]
void pack_pair_slab(const uint16_t* w, int tile0, bool permuted, uint16_t* slab) {
    for (int f = 0; f < FR; ++f) pack_frag16(w, D, 16 * (tile0 + (f & 1)), f >> 1, permuted, slab + f * 512);
}

template <bool MLP, bool QKV, bool TAP>
int launch16(const vx_dino_block_args& a, void* stream) {
    auto kern = dino_block16_kernel<MLP, QKV, TAP>;
    VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SMEM_BYTES));
    hipLaunchKernelGGL(kern, dim3((a.M + 127) / 128), dim3(512), SMEM_BYTES, as_stream(stream), a);
    VX_LAUNCH_CHECK();
    return 1;
}

} // namespace

extern "C" {

int vx_dino_block_supported(int embed_dim, int hidden, int head_dim) { return embed_dim == D && hidden == HID && head_dim == 64; }
size_t vx_dino_block_mlp_bytes(void) { return (size_t)(N_OUT + N_MLP) * SLAB; }
size_t vx_dino_block_qkv_bytes(void) { return (size_t)N_QKV * SLAB; }

int vx_dino_block16_pack_mlp(const void* wo, const void* w1, const void* w2, void* out) {
    VX_REQUIRE(wo && w1 && w2 && out, "vx_dino_block16_pack_mlp: null pointer");
    const uint16_t *o = static_cast<const uint16_t*>(wo), *a = static_cast<const uint16_t*>(w1), *b = static_cast<const uint16_t*>(w2);
    uint16_t* dst = static_cast<uint16_t*>(out);
    auto slab = [&](int k) { return dst + (size_t)k * (SLAB / 2); };
    for (int s = 0; s < N_OUT; ++s) pack_pair_slab(o, 2 * s, false, slab(s)); // B operand = attention rows in natural order
    auto fc1 = [&](int u, uint16_t* sl) { pack_pair_slab(a, 2 * u, true, sl); };
    auto fc2 = [&](int u, uint16_t* sl) { // the 24 feature tiles of hidden block u
        for (int T = 0; T < NT; ++T) pack_frag16(b, HID, 16 * T, u, true, sl + T * 512);
    };
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

int vx_dino_block16_pack_qkv(const void* wqkv, void* out) {
    VX_REQUIRE(wqkv && out, "vx_dino_block16_pack_qkv: null pointer");
    uint16_t* dst = static_cast<uint16_t*>(out);
    const uint16_t* w = static_cast<const uint16_t*>(wqkv);
    // slab v = output features 32v .. 32v+31 as two tiles; row i of tile e is feature 32v + 8 (i >> 2) + 4e + (i & 3), so that the
    // lane holding rows 4g .. 4g+3 of both tiles owns 8 consecutive features (one 16-byte store in the kernel)
    for (int v = 0; v < N_QKV; ++v)
        for (int f = 0; f < FR; ++f) {
            const int e = f & 1, kb = f >> 1;
            uint16_t* frag = dst + (size_t)v * (SLAB / 2) + f * 512;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int i = l & 15, g = l >> 4;
                    const int row = 32 * v + 8 * (i >> 2) + 4 * e + (i & 3);
                    const int k = 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3);
                    frag[l * 8 + j] = w[(size_t)row * D + k];
                }
        }
    return 1;
}

int vx_dino_block16_f16(const vx_dino_block_args* args, void* stream) {
    const vx_dino_block_args& a = *args;
    VX_REQUIRE(a.M > 0 && a.x, "vx_dino_block16_f16: empty problem");
    const bool mlp = a.att != nullptr, qkv = a.q != nullptr, tap = a.feat != nullptr;
    VX_REQUIRE(mlp || qkv, "vx_dino_block16_f16: nothing to do (neither att nor q given)");
    if (mlp) VX_REQUIRE(a.w_mlp && a.vec_mlp, "vx_dino_block16_f16: the MLP half needs w_mlp and vec_mlp");
    if (qkv) VX_REQUIRE(a.w_qkv && a.vec_qkv && a.k && a.v && a.T > 0 && a.H > 0 && a.M % a.T == 0, "vx_dino_block16_f16: the QKV half needs w_qkv, vec_qkv, k, v, T, H and M %% T == 0");
    if (tap) VX_REQUIRE(a.vec_tap, "vx_dino_block16_f16: the tap needs vec_tap");
    VX_REQUIRE((long)a.M * D * 4 < 0x7fffffffL, "vx_dino_block16_f16: M = %d rows exceed the 2 GiB buffer range", a.M);
    if (mlp && qkv && tap) return launch16<true, true, true>(a, stream);
    if (mlp && qkv) return launch16<true, true, false>(a, stream);
    if (mlp && tap) return launch16<true, false, true>(a, stream);
    if (mlp) return launch16<true, false, false>(a, stream);
    VX_REQUIRE(!tap, "vx_dino_block16_f16: a tap without the MLP half is not an instance of this kernel");
    return launch16<false, true, false>(a, stream);
}

} // extern "C"
