// Fused multi-head self-attention for head_dim 64 on gfx950.
// Replaces the reference's permute/cont/mul_mat/soft_max_ext/mul_mat/permute/cont chain
// (src/visp/nn.cpp:210-244 non-flash branch; flash branch :217-227) emitted by
// dino::self_attention (src/visp/arch/dino.cpp:59-74).
//
// One block = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32 queries.
// "Swapped" formulation: S^T = K * Q^T on v_mfma_f32_32x32x16_f16, so that a lane owns ONE
// query column (lane & 31) and the softmax row statistics are per-lane scalars:
//   S^T[key, q]:  A = K tile rows (keys, d contiguous), B = Q rows (d contiguous, in registers)
//   O^T[d, q]  :  A = V^T fragments read from the row-major V tile with ds_read_b64_tr_b16
//                 (hardware transpose), B = P^T taken straight from the S^T accumulators
//                 (element j of k-step s is key 16s + 8(j>>2) + 4h + (j&3)).
// K and V tiles (64 keys x 64 d, 8 KB each) go global -> LDS directly (global_load_lds_dwordx4)
// into a 2-deep ring, the next tile's loads are issued before the current tile's math. The
// lane-linear LDS image is swizzled through the per-lane source address: K chunks by (row>>1)&7
// (ds_read_b128 of 32 rows conflict free), V chunks by ((row>>1)&1)<<2 (the 4-row transposed
// reads of a half-wave hit 4 different 64-byte bank groups).
// Softmax is online, f32, in the exp2 domain: the producer of q folds log2(e) / sqrt(64) into it (VX_ATTN_Q_SCALE), so a
// score needs no multiply. Round 3: the kernel is VALU-issue bound at head_dim 64 (per 64-key tile and wave 16 MFMAs = 512
// pipe cycles against ~135 VALU instructions ~ 800 issue cycles), so the steady-state tile does no maximum and no subtract:
// the QK^T chains start from a tile that holds -m_run (the running maximum so far) instead of zero, the accumulators ARE
// s - m_run, and P = exp2 of them. That is an online softmax whose reference point lags (guide T13 taken to its end): exact in
// real arithmetic for any reference point, and P in f16 keeps its 2^-11 relative precision as long as it does not overflow. The
// tile's partial row sum, needed anyway, doubles as the overflow check: if any lane's sum exceeds 2^10 (so a P could exceed 2^10)
// the wave drops the tile's fast result and redoes it on the full path (maximum, rescale of O and l, new reference point). The
// first tile (no reference point yet) and the last one (key mask) always take the full path.
#include "vx_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

#ifndef VISP_ATTN_WPE
#define VISP_ATTN_WPE 4 // waves per SIMD the register budget is held to (experiments: 3 = 168 registers)
#endif
#ifndef VISP_ATTN_RS
#define VISP_ATTN_RS 1 // row sums: 1 = packed-f16 add tree (default), 2 = two f16 levels then f32, 0 = ones-MFMAs (profiles/r03_attention_stamps.txt)
#endif
#ifndef VISP_ATTN_SEQ
#define VISP_ATTN_SEQ 1 // A/B builds: 0 lets hipcc overlap the two 32-key blocks of a tile (profiles/r03_attention_ab_fastpath.txt)
#endif
constexpr int HD = 64;        // head dim
constexpr int KV_TILE = 64;   // keys per tile
constexpr int Q_PER_WAVE = 32;
constexpr int TILE_BYTES = KV_TILE * HD * 2;
constexpr float FAST_LIMIT_DEFAULT = 1024.0f; // a lane's partial row sum above this sends the tile to the full path

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ int k_swz(int row, int chunk16) { return chunk16 ^ ((row >> 1) & 7); }
__device__ __forceinline__ int v_swz(int row, int chunk16) { return chunk16 ^ (((row >> 1) & 1) << 2); }

// value held by the same lane of the other 32-lane half (lanes l and l+32 own the same query column)
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // after the swap r[0] = {low: own low, high: low half's values}, r[1] = {low: high half's values, high: own high}
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// four waves per SIMD: the register allocation is held at 128 (the steady-state tile runs spill-free in it, the rare full path keeps
// a few invariants in scratch). NW waves per block = 32 NW queries share one K/V tile stream: the 16 LDS-DMA instructions of a tile
// are split over the block's waves, and every one of them costs its wave 60-180 issue cycles in a kernel that is issue bound.
// STAMP (diagnostics only, tools/attn_stamps.py): per wave, cycles spent in {wait + barrier + DMA issue, scores + exponentials + row sum, PV product}
// and the wave's lifetime, by s_memtime around the phases of every tile; never instantiated on the product path.
template <int NW, bool STAMP = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(VISP_ATTN_WPE, VISP_ATTN_WPE))) void attention_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                         const f16* __restrict__ V, f16* __restrict__ O, int H, int T, float FAST_LIMIT,
                                                         unsigned long long* __restrict__ stamps = nullptr) {
    unsigned long long st_wait = 0, st_soft = 0, st_pv = 0, st_t0 = 0;
    if constexpr (STAMP) st_t0 = __builtin_amdgcn_s_memtime();
    // LDS ring: 2 stages x (K tile, V tile)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware block order: blocks id and id+8 share an XCD (and its L2). Give every XCD a contiguous run of
    // the (b*H+head major, q-block minor) sequence, so the q-blocks that sweep the same K/V (350 KB per head)
    // hit it in ONE L2 instead of fetching it through all eight (measured: 5.5x the algorithmic fabric reads).
    constexpr int Q_PER_BLOCK = Q_PER_WAVE * NW;
    const int n_qblk = (T + Q_PER_BLOCK - 1) / Q_PER_BLOCK;
    int logical;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int per = total >> 3, rem = total & 7, xcd = id & 7;
        logical = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (id >> 3);
    }
    const int bh = logical / n_qblk;            // b * H + head
    const int q0 = (logical - bh * n_qblk) * Q_PER_BLOCK + wave * Q_PER_WAVE;

    const f16* Qb = Q + (long)bh * T * HD;
    const f16* Kb = K + (long)bh * T * HD;
    const f16* Vb = V + (long)bh * T * HD;

    // Q fragments (B operand): lane holds Q[q0 + r][16*s + 8h .. +7], s = 0..3
    f16x8 qf[4];
    {
        int qrow = q0 + r;
        if (qrow >= T) qrow = T - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(Qb + (long)qrow * HD + 16 * s + 8 * h);
    }

    // staging: the block's waves fill the 64 rows of both tiles with 8 + 8 LDS-DMA instructions of 8 rows, 16 / NW each. The tiles come through
    // buffer descriptors (base and size of this head's K / V in SGPRs, a 32-bit byte offset per lane): no 64-bit lane pointers stay
    // live across the tile loop, and rows at or beyond T are zero-filled by the range check (their scores are masked on the last
    // tile, their P is 0).
    const int l_row = lane >> 3, l_pos = lane & 7;
    const int n_tiles = (T + KV_TILE - 1) / KV_TILE;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(Kb), 0, T * HD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(Vb), 0, T * HD * 2, 0x00020000);

    auto issue_loads = [&](int t, int buf) {
        unsigned char* sk = smem + buf * (2 * TILE_BYTES);
        unsigned char* sv = sk + TILE_BYTES;
        const int tile_bytes0 = t * (KV_TILE * HD * 2);
        constexpr int PER = 8 / NW; // 8-row pieces of each tile per wave
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int piece = wave * PER + i, row = piece * 8 + l_row;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(sk + piece * 1024), 16,
                                                     tile_bytes0 + row * (HD * 2) + k_swz(row, l_pos) * 16, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(sv + piece * 1024), 16,
                                                     tile_bytes0 + row * (HD * 2) + v_swz(row, l_pos) * 16, 0, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = 0.0f;      // reference point of the exponentials (log2 domain), shared by both lane halves of a query; the first
                             // tile sets it to the tile's row maximum, later full-path tiles raise it
    float l_run = 0.0f;      // running row sum (the same in both lane halves of a query)

    // transposed-read lane roles: 16-lane group g = lane>>4 covers d0 = 16*(g&1) .. +15 of key-half g>>1 (== h);
    // lane 4q+p of the group supplies the address of key row q, columns d0 + 4p .. +3
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_d0 = 16 * ((lane >> 4) & 1);
    // LDS byte offsets: K fragment of d-step st, key block kb = (k_off0 ^ st*32) + kb*4096 (the swizzle (r>>1)&7 is per lane and
    // the d-step only flips bits 5-6 of it); V^T tr-read of d block db, key step ks, half a/b = (v_off0 ^ db*64) + ks*2048 (+1024).
    const int k_off0 = r * 128 + k_swz(r, h) * 16;
    int v_off0;
    {
        const int key = 4 * h + tr_q, dd = tr_d0 + 4 * tr_p;
        v_off0 = key * 128 + v_swz(key, dd >> 3) * 16 + (dd & 7) * 2;
    }
    auto k_off = [&](int st) { return k_off0 ^ (st << 5); };
    auto v_off = [&](int db) { return v_off0 ^ (db << 6); };

    // -m_run as an accumulator tile: the C operand of the first QK^T step (a lane = one query column, so its 16 registers hold
    // the same value)
    f32x16 negm;
#pragma unroll
    for (int e = 0; e < 16; ++e) negm[e] = 0.0f;

    // S^T - m_run of one 32-key block: 4 d-steps on top of negm (keys beyond T masked on the last tile)
    auto qk = [&](const unsigned char* sk, int kb, int k0, bool masked) {
        f32x16 s;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8 kf = *reinterpret_cast<const f16x8*>(sk + k_off(st) + kb * 4096);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], st == 0 ? negm : s, 0, 0, 0);
        }
        if (masked) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (key >= T) s[e] = -INFINITY;
            }
        }
        return s;
    };
    // (Row sum, round 3 late: a packed-f16 add tree over the 16 P registers -- 15 v_pk_add_f16, two converts, one lane-half swap -- replaced
    // the four ones-MFMAs described below: timing-only ablations showed the kernel is the SUM of its parts (no row-sum MFMAs: -10 %, no PV:
    // -25 %, no exponentials: -8 %, no barrier / DMA: -11 %), not bound by one pipe; 125 -> 119 us at batch 32, MAE vs the oracle unchanged
    // (1.6e-4 / 2.2e-4 before). The partial sums are f16: 4 roundings of 2^-11 on sums of <= 16 values, zero-mean; an overflowed P still
    // makes the sum inf and sends the tile to the full path.)
    // P = exp2(S^T - m_run) of both key blocks as packed f16 fragments (element j of k-step ks <-> key 16ks + 8(j>>2) + 4h + (j&3)),
    // one block at a time so that only 16 score registers are live. Returns the tile's row sum of the f16 P -- the values the PV
    // product uses -- taken on the matrix pipe: ones[32 x 64] P^T has the 64-key sum of query column q in every row, 4 MFMAs (32
    // issue cycles) instead of 32 v_add_f32 (128) in a kernel that is bound by instruction issue, not by the MFMA pipe. Both lane
    // halves of a query get the full sum.
    const f16x8 ones = {(f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1};
    auto softmax_p = [&](const unsigned char* sk, int k0, bool masked, f16x8 (&pf)[4]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = qk(sk, kb, k0, masked);
#pragma unroll
            for (int e = 0; e < 16; ++e) pf[2 * kb + (e >> 3)][e & 7] = (f16)__builtin_amdgcn_exp2f(s[e]);
            if (VISP_ATTN_SEQ) __builtin_amdgcn_sched_barrier(0); // keep the blocks in sequence: the register budget is 128 (4 waves per SIMD)
        }
        if constexpr (VISP_ATTN_RS) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = h2{pf[i >> 2][2 * (i & 3)], pf[i >> 2][2 * (i & 3) + 1]};
#pragma unroll
            for (int w = 8; w >= (VISP_ATTN_RS == 2 ? 4 : 1); w >>= 1)
#pragma unroll
                for (int i = 0; i < w; ++i) t[i] = t[i] + t[i + w];
            float half_sum;
            if (VISP_ATTN_RS == 2) // two f16 levels (sums of 4 values), the rest in f32
                half_sum = (((float)t[0][0] + (float)t[0][1]) + ((float)t[1][0] + (float)t[1][1])) + (((float)t[2][0] + (float)t[2][1]) + ((float)t[3][0] + (float)t[3][1]));
            else
                half_sum = (float)t[0][0] + (float)t[0][1];
            return half_sum + other_half(half_sum);
        }
        f32x16 rs = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) rs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[ks], rs, 0, 0, 0);
        return rs[0];
    };
    // O^T += V^T P^T : 4 key steps x 2 d blocks; V^T fragments by transposed LDS reads
    auto pv = [&](const unsigned char* sv, const f16x8 (&pf)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                // two 4-key blocks per fragment, each one tr read
                hv4 va = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) hv4*)(sv + v_off(db) + ks * 2048));
                hv4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) hv4*)(sv + v_off(db) + ks * 2048 + 1024));
                f16x8 vf = {(f16)va[0], (f16)va[1], (f16)va[2], (f16)va[3], (f16)vb[0], (f16)vb[1], (f16)vb[2], (f16)vb[3]};
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], o[db], 0, 0, 0);
            }
        }
    };

    // one 64-key tile; BUF is a compile-time constant so every LDS offset folds into an immediate
    auto tile_body = [&](int t, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        if constexpr (STAMP) c0 = __builtin_amdgcn_s_memtime();
        // (one barrier per TWO tiles with a 4-stage ring was measured in round 3: no change -- the wait is landing + issue, not skew)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // tile t landed; every wave finished tile t-1, so the other stage is free
        if (t + 1 < n_tiles) issue_loads(t + 1, BUF ^ 1);
        if constexpr (STAMP) { c1 = __builtin_amdgcn_s_memtime(); st_wait += c1 - c0; }
        const unsigned char* sk = smem + BUF * (2 * TILE_BYTES);
        const unsigned char* sv = sk + TILE_BYTES;
        if (q0 >= T) return; // (wave-uniform) a wave without queries only feeds the ring and keeps the barriers
        const int k0 = t * KV_TILE;
        const bool last = k0 + KV_TILE > T;
        f16x8 pf[4];
        float psum = 0.0f;

        // ---- fast path: exponentials relative to the reference point the wave already has. Every P of a row is <= the row's sum
        // over the tile: below the limit nothing can overflow f16 (an overflowed P = inf makes the sum inf or NaN: caught).
        bool done = false;
        if (t > 0 && !last) {
            psum = softmax_p(sk, k0, false, pf);
            done = !__any(!(psum <= FAST_LIMIT));
        }
        // ---- full path (first tile, last tile, or a maximum ran away from the reference point): the relative scores' maximum
        // first -- delta = max(0, max s') (first tile: max s' itself, so the reference follows rows whose scores all lie far below
        // zero) -- then m_run += delta, O and l scaled by 2^-delta, and the exponentials against the new reference point. The
        // scores are computed twice (8 more MFMAs) rather than kept: this path is rare and the registers are not there.
        if (!done) {
            float mloc = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f32x16 s = qk(sk, kb, k0, last);
#pragma unroll
                for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s[e]);
                __builtin_amdgcn_sched_barrier(0);
            }
            mloc = fmaxf(mloc, other_half(mloc)); // v_permlane32_swap: lanes l and l+32 own the same query
            const float delta = t == 0 ? mloc : fmaxf(mloc, 0.0f);
            if (t == 0 || __any(delta > 0.0f)) { // wave-uniform
                if (t > 0) {                     // (first tile: O and l are still zero)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l_run *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
                }
                m_run += delta;
#pragma unroll
                for (int e = 0; e < 16; ++e) negm[e] = -m_run;
            }
            psum = softmax_p(sk, k0, last, pf);
        }
        l_run += psum;
        if constexpr (STAMP) { c2 = __builtin_amdgcn_s_memtime(); st_soft += c2 - c1; }
        pv(sv, pf); // the one place O is accumulated
        if constexpr (STAMP) {
            asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][0])); // the last PV results are due before the clock is read
            st_pv += __builtin_amdgcn_s_memtime() - c2;
        }
    };

    issue_loads(0, 0);
    for (int t = 0; t < n_tiles; t += 2) {
        tile_body(t, std::integral_constant<int, 0>{});
        if (t + 1 < n_tiles) tile_body(t + 1, std::integral_constant<int, 1>{});
    }

    if constexpr (STAMP) {
        if (lane == 0) {
            unsigned long long* s = stamps + ((size_t)blockIdx.x * NW + wave) * 4;
            s[0] = st_wait; s[1] = st_soft; s[2] = st_pv; s[3] = __builtin_amdgcn_s_memtime() - st_t0;
        }
    }
    // ---- finalize: O[q, head*64 + d] = o / l
    float inv = 1.0f / l_run;
    const int q = q0 + r;
    if (q < T) {
        const int b = bh / H, head = bh - b * H;
        f16* orow = O + ((long)b * T + q) * (H * HD) + head * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (f16)(o[db][g * 4 + j] * inv);
                *reinterpret_cast<f16x4*>(orow + db * 32 + 8 * g + 4 * h) = v; // rows d = db*32 + 8g + 4h + j
            }
    }
}

} // namespace

// test hook (guide rule 26, threshold sweep): 0 sends every tile down the full path, a huge value never leaves the fast path
static float g_fast_limit = FAST_LIMIT_DEFAULT;
extern "C" void vx_attention_set_fast_limit(float limit) { g_fast_limit = limit < 0.0f ? FAST_LIMIT_DEFAULT : limit; }

// diagnostics: u64 [blocks][waves][4] = {wait, softmax, pv, lifetime} cycles of the next launches; NULL = the product kernel
static void* g_attn_stamps = nullptr;
extern "C" void vx_attention_set_stamps(void* stamps) { g_attn_stamps = stamps; }

extern "C" int vx_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int T, void* stream) {
    VX_REQUIRE(B > 0 && H > 0 && T > 0, "vx_attention_f16: empty problem");
    static const int nw_env = getenv("VISP_ATTN_WAVES") ? atoi(getenv("VISP_ATTN_WAVES")) : 0;
    const int nw = nw_env == 4 || nw_env == 8 ? nw_env : (T > 512 ? 8 : 4); // short sequences: smaller blocks fill the chip better
    const int qpb = Q_PER_WAVE * nw;
    dim3 grid(((T + qpb - 1) / qpb) * B * H);
    if (g_attn_stamps)
        hipLaunchKernelGGL((nw == 8 ? attention_kernel<8, true> : attention_kernel<4, true>), grid, dim3(64 * nw), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                           reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v), reinterpret_cast<f16*>(out), H, T, g_fast_limit,
                           static_cast<unsigned long long*>(g_attn_stamps));
    else
        hipLaunchKernelGGL((nw == 8 ? attention_kernel<8, false> : attention_kernel<4, false>), grid, dim3(64 * nw), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                           reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v), reinterpret_cast<f16*>(out), H, T, g_fast_limit,
                           static_cast<unsigned long long*>(nullptr));
    VX_LAUNCH_CHECK();
    return 1;
}
