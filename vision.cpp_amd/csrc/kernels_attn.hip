// Fused multi-head self-attention for head_dim 64 on gfx950.
// Replaces the reference's permute/cont/mul_mat/soft_max_ext/mul_mat/permute/cont chain
// (src/visp/nn.cpp:210-244 non-flash branch; flash branch :217-227) emitted by
// dino::self_attention (src/visp/arch/dino.cpp:59-74).
//
// One block = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32 queries.
// "Swapped" formulation: S^T = K * Q^T on v_mfma_f32_32x32x16_f16, so that a lane owns ONE
// query column (lane & 31) and the softmax row statistics are per-lane scalars:
//   S^T[key, q]:  A = K tile rows (keys, d contiguous), B = Q rows (d contiguous, in registers)
//   O^T[d, q]  :  A = V^T tile rows (d, keys contiguous), B = P^T taken straight from the S^T
//                 accumulators (register e of k-step s is key 16s + 8(j>>2) + 4h + (j&3)).
// K and V^T tiles (64 keys) are staged through LDS, register double-buffered; 16-byte (K) and
// 8-byte (V^T) chunks are XOR-swizzled so the MFMA operand reads are bank-conflict free.
// Softmax is online, f32, in the exp2 domain (log2(e) folded into one FMA per score).
#include "vx_common.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr int KV_TILE = 64;   // keys per tile
constexpr int Q_PER_WAVE = 32;
constexpr int Q_PER_BLOCK = 128;
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ int k_swz(int row, int chunk16) { return chunk16 ^ ((row >> 1) & 7); }
__device__ __forceinline__ int v_swz(int row, int chunk8) { return chunk8 ^ ((row >> 1) & 15); }

__global__ __launch_bounds__(256) void attention_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                         const f16* __restrict__ Vt, f16* __restrict__ O,
                                                         int H, int T, int Tp) {
    // LDS: 2 stages x (K tile 64x64 f16 = 8 KB, V^T tile 64x64 f16 = 8 KB)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * KV_TILE * HD * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y;                  // b * H + head
    const int q0 = blockIdx.x * Q_PER_BLOCK + wave * Q_PER_WAVE;

    const f16* Qb = Q + (long)bh * T * HD;
    const f16* Kb = K + (long)bh * T * HD;
    const f16* Vb = Vt + (long)bh * HD * Tp;

    // Q fragments (B operand): lane holds Q[q0 + r][16*s + 8h .. +7], s = 0..3
    f16x8 qf[4];
    {
        int qrow = q0 + r;
        if (qrow >= T) qrow = T - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f16x8*>(Qb + (long)qrow * HD + 16 * s + 8 * h);
    }

    // staging coordinates: 512 16-byte chunks per tile, 2 per thread for K and for V^T
    int s_row[2], s_ch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int idx = tid + i * 256;
        s_row[i] = idx >> 3;
        s_ch[i] = idx & 7;
    }
    f16x8 kreg[2], vreg[2];
    const int n_tiles = (T + KV_TILE - 1) / KV_TILE;

    auto load_tile = [&](int t) {
        const int k0 = t * KV_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = k0 + s_row[i];
            if (key >= T) key = T - 1;          // clamped rows are masked in the scores
            kreg[i] = *reinterpret_cast<const f16x8*>(Kb + (long)key * HD + s_ch[i] * 8);
            // V^T row d = s_row, keys k0 + 8*ch .. +7 (Tp >= n_tiles*64, pad columns are finite)
            vreg[i] = *reinterpret_cast<const f16x8*>(Vb + (long)s_row[i] * Tp + k0 + s_ch[i] * 8);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sk = smem + buf * (2 * KV_TILE * HD * 2);
        unsigned char* sv = sk + KV_TILE * HD * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f16x8*>(sk + s_row[i] * 128 + k_swz(s_row[i], s_ch[i]) * 16) = kreg[i];
            f16x4 lo = {vreg[i][0], vreg[i][1], vreg[i][2], vreg[i][3]};
            f16x4 hi = {vreg[i][4], vreg[i][5], vreg[i][6], vreg[i][7]};
            *reinterpret_cast<f16x4*>(sv + s_row[i] * 128 + v_swz(s_row[i], s_ch[i] * 2) * 8) = lo;
            *reinterpret_cast<f16x4*>(sv + s_row[i] * 128 + v_swz(s_row[i], s_ch[i] * 2 + 1) * 8) = hi;
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = -INFINITY; // running max (exp2 domain), shared by both lane halves of a query
    float l_run = 0.0f;      // this lane half's partial row sum

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) load_tile(t + 1);
        const unsigned char* sk = smem + buf * (2 * KV_TILE * HD * 2);
        const unsigned char* sv = sk + KV_TILE * HD * 2;

        // ---- S^T = K Q^T : 2 key blocks x 4 d-steps
        f32x16 s[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[0][e] = 0.0f; s[1][e] = 0.0f; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                int row = kb * 32 + r;
                f16x8 kf = *reinterpret_cast<const f16x8*>(sk + row * 128 + k_swz(row, st * 2 + h) * 16);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], s[kb], 0, 0, 0);
            }
        }
        // ---- mask keys beyond T (last tile only), scale into exp2 domain, running max
        const int k0 = t * KV_TILE;
        float mloc = -INFINITY;
        if (k0 + KV_TILE > T) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key >= T) s[kb][e] = -INFINITY;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s[kb][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc * LOG2E);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][e], LOG2E, -m_new));
                s[kb][e] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }

        // ---- O^T += V^T P^T : 4 key steps x 2 d blocks
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ks >> 1][8 * (ks & 1) + j];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                int row = db * 32 + r;
                // element j <-> key 16ks + 8(j>>2) + 4h + (j&3): two 8-byte reads
                int c0 = (16 * ks + 4 * h) >> 2;       // 8-byte chunk index of keys 16ks+4h..+3
                int c1 = (16 * ks + 8 + 4 * h) >> 2;
                f16x4 v0 = *reinterpret_cast<const f16x4*>(sv + row * 128 + v_swz(row, c0) * 8);
                f16x4 v1 = *reinterpret_cast<const f16x4*>(sv + row * 128 + v_swz(row, c1) * 8);
                f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
        if (t + 1 < n_tiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- finalize: O[q, head*64 + d] = o / l
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
                // rows (d) = db*32 + 8g + 4h + j
                *reinterpret_cast<f16x4*>(orow + db * 32 + 8 * g + 4 * h) = v;
            }
    }
}

} // namespace

extern "C" int vx_attention_f16(const void* q, const void* k, const void* vt, void* out, int B, int H, int T, int Tp,
                                void* stream) {
    VX_REQUIRE(B > 0 && H > 0 && T > 0, "vx_attention_f16: empty problem");
    const int n_tiles = (T + KV_TILE - 1) / KV_TILE;
    VX_REQUIRE(Tp >= n_tiles * KV_TILE && Tp % 8 == 0, "vx_attention_f16: Tp=%d must be >= %d and a multiple of 8", Tp,
               n_tiles * KV_TILE);
    dim3 grid((T + Q_PER_BLOCK - 1) / Q_PER_BLOCK, B * H);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, as_stream(stream), reinterpret_cast<const f16*>(q),
                       reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(vt), reinterpret_cast<f16*>(out), H, T, Tp);
    VX_LAUNCH_CHECK();
    return 1;
}
