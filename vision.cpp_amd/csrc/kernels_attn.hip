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
// Softmax is online, f32, in the exp2 domain (log2(e) folded into one FMA per score).
#include "vx_common.h"

#include <type_traits>

namespace {

constexpr int HD = 64;        // head dim
constexpr int KV_TILE = 64;   // keys per tile
constexpr int Q_PER_WAVE = 32;
constexpr int Q_PER_BLOCK = 128;
constexpr int TILE_BYTES = KV_TILE * HD * 2;
constexpr float LOG2E = 1.44269504088896340736f;

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

__global__ __launch_bounds__(256) void attention_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                         const f16* __restrict__ V, f16* __restrict__ O, int H, int T) {
    // LDS ring: 2 stages x (K tile, V tile)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware block order: blocks id and id+8 share an XCD (and its L2). Give every XCD a contiguous run of
    // the (b*H+head major, q-block minor) sequence, so the q-blocks that sweep the same K/V (350 KB per head)
    // hit it in ONE L2 instead of fetching it through all eight (measured: 5.5x the algorithmic fabric reads).
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

    // staging: each wave fills rows 16*wave .. +15 of both tiles with 2 + 2 instructions of 8 rows
    const int l_row = lane >> 3, l_pos = lane & 7;
    const int n_tiles = (T + KV_TILE - 1) / KV_TILE;

    auto issue_loads = [&](int t, int buf) {
        unsigned char* sk = smem + buf * (2 * TILE_BYTES);
        unsigned char* sv = sk + TILE_BYTES;
        const int k0 = t * KV_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + l_row;
            int key = k0 + row;
            if (key >= T) key = T - 1; // clamped rows: scores masked to -inf, V rows multiplied by P = 0
            __builtin_amdgcn_global_load_lds((gptr_t)(Kb + (long)key * HD + k_swz(row, l_pos) * 8),
                                             (lptr_t)(sk + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(Vb + (long)key * HD + v_swz(row, l_pos) * 8),
                                             (lptr_t)(sv + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = -INFINITY; // running max (exp2 domain), shared by both lane halves of a query
    float l_run = 0.0f;      // this lane half's partial row sum

    // transposed-read lane roles: 16-lane group g = lane>>4 covers d0 = 16*(g&1) .. +15 of key-half g>>1 (== h);
    // lane 4q+p of the group supplies the address of key row q, columns d0 + 4p .. +3
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_d0 = 16 * ((lane >> 4) & 1);
    // loop-invariant LDS byte offsets (everything else is a compile-time immediate):
    //  K fragment of d-step st, key block kb: k_off[st] + kb*4096   (swizzle term (r>>1)&7 is per lane)
    //  V^T tr-read of d block db, key step ks, half a/b: v_off[db] + ks*2048 (+1024 for b)
    int k_off[4], v_off[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) k_off[st] = r * 128 + k_swz(r, st * 2 + h) * 16;
    {
        const int key = 4 * h + tr_q, dd = tr_d0 + 4 * tr_p;
#pragma unroll
        for (int db = 0; db < 2; ++db) v_off[db] = key * 128 + v_swz(key, (db * 32 + dd) >> 3) * 16 + (dd & 7) * 2;
    }

    // one 64-key tile; BUF is a compile-time constant so every LDS offset folds into an immediate
    auto tile_body = [&](int t, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // tile t landed; every wave finished tile t-1, so the other stage is free
        if (t + 1 < n_tiles) issue_loads(t + 1, BUF ^ 1);
        const unsigned char* sk = smem + BUF * (2 * TILE_BYTES);
        const unsigned char* sv = sk + TILE_BYTES;

        // ---- S^T = K Q^T : 2 key blocks x 4 d-steps (first step accumulates onto a constant zero)
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8 kf = *reinterpret_cast<const f16x8*>(sk + k_off[st] + kb * 4096);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], st == 0 ? zero : s[kb], 0, 0, 0);
            }
        }
        // ---- mask keys beyond T (last tile only), running max in the exp2 domain
        const int k0 = t * KV_TILE;
        if (k0 + KV_TILE > T) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key >= T) s[kb][e] = -INFINITY;
                }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s[kb][e]);
        mloc = fmaxf(mloc, other_half(mloc)); // v_permlane32_swap: no LDS round trip on the softmax critical path
        const float m_new = fmaxf(m_run, mloc * LOG2E);
        // rescale the running output only when some query of this wave saw a larger maximum
        // (wave-uniform branch; after the first tiles the maxima rarely move)
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
            m_run = m_new;
        }
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][e], LOG2E, -m_run));
                s[kb][e] = pv;
                psum += pv;
            }
        l_run += psum;

        // ---- O^T += V^T P^T : 4 key steps x 2 d blocks; V^T fragments by transposed LDS reads
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ks >> 1][8 * (ks & 1) + j];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                // element j <-> key 16ks + 8(j>>2) + 4h + (j&3): two 4-key blocks, each one tr read
                hv4 va = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) hv4*)(sv + v_off[db] + ks * 2048));
                hv4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) hv4*)(sv + v_off[db] + ks * 2048 + 1024));
                f16x8 vf = {(f16)va[0], (f16)va[1], (f16)va[2], (f16)va[3], (f16)vb[0], (f16)vb[1], (f16)vb[2], (f16)vb[3]};
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
    };

    issue_loads(0, 0);
    for (int t = 0; t < n_tiles; t += 2) {
        tile_body(t, std::integral_constant<int, 0>{});
        if (t + 1 < n_tiles) tile_body(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- finalize: O[q, head*64 + d] = o / l
    float l_tot = l_run + other_half(l_run);
    float inv = 1.0f / l_tot;
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

extern "C" int vx_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int T, void* stream) {
    VX_REQUIRE(B > 0 && H > 0 && T > 0, "vx_attention_f16: empty problem");
    dim3 grid(((T + Q_PER_BLOCK - 1) / Q_PER_BLOCK) * B * H);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                       reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v), reinterpret_cast<f16*>(out), H, T);
    VX_LAUNCH_CHECK();
    return 1;
}
