// Window attention with relative position bias, head_dim 32, on the matrix cores of gfx950
// (TinyViT attention_rel_bias, reference src/visp/arch/mobile-sam.cpp:112-131: q k^T * scale + bias -> softmax -> v).
//
// qkv f16 [n_windows * N][heads * 96] (per head q | k | v, as the qkv linear writes it); out f16 [n_windows * N][heads * 32].
// One block = one (window, head); wave w owns queries 32w .. 32w+31, so a block has QB = ceil(N / 32) waves (2 for the
// 7x7 windows, 7 for 14x14). K and V of the head (N x 32 each) go global -> LDS once (global_load_lds_dwordx4).
// "Swapped" formulation as in kernels_attn.hip: S^T = K Q^T on v_mfma_f32_32x32x16_f16, a lane owns one query column, the
// softmax is online over 32-key blocks, P^T feeds O^T = V^T P^T straight from the accumulators, V^T fragments come from
// ds_read_b64_tr_b16.
// The bias is pre-packed by the host in exactly the accumulator order (vx_window_attention_pack_bias): f16
// [head][query block][key block][lane][16] -- two coalesced 16-byte loads per lane and key block -- with -inf on the
// padded keys, so masking costs nothing. GGUF stores attention_biases_indexed as f16, so the packing is lossless.
#include "vx_common.h"

#include <cmath>
#include <cstring>

namespace {

constexpr int WA_HD = 32;
constexpr float WA_LOG2E = 1.44269504088896340736f;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ float wa_other_half(float v) { // value of the lane that owns the same query in the other 32-lane half
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

template <int QB>
__global__ __launch_bounds__(QB * 64) void window_attention_kernel(const f16* __restrict__ qkv, const f16* __restrict__ bias, f16* __restrict__ out,
                                                                   int N, int heads, float scale) {
    constexpr int NP = QB * 32;                      // padded tokens
    constexpr int ROW = WA_HD * 2;                   // bytes per K / V row in LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP * ROW];
    unsigned char* const sk = smem;
    unsigned char* const sv = smem + NP * ROW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int win = blockIdx.x / heads, head = blockIdx.x - win * heads;
    const int ld = heads * 3 * WA_HD;
    const f16* base = qkv + (long)win * N * ld + head * 3 * WA_HD;

    // ---- K, V -> LDS: one instruction = 16 rows of 64 bytes, lane l lands at row l>>2, 16-byte chunk l&3.
    // K chunks are swizzled by (row>>2)&3 through the source address (ds_read_b128 of 16 rows then hits 16 bank groups).
    {
        const int l_row = lane >> 2, l_pos = lane & 3;
        for (int i = wave; i < NP / 16; i += QB) {
            int key = i * 16 + l_row;
            if (key >= N) key = N - 1; // clamped rows: P is exactly 0 there (bias -inf)
            const f16* src = base + (long)key * ld;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + WA_HD + (l_pos ^ ((l_row >> 2) & 3)) * 8), (lptr_t)(sk + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 2 * WA_HD + l_pos * 8), (lptr_t)(sv + i * 1024), 16, 0, 0);
        }
    }
    // Q fragments (B operand): lane holds Q[q][16 s + 8 h .. +7]
    const int q = wave * 32 + r;
    f16x8 qf[2];
    {
        const f16* qrow = base + (long)(q < N ? q : N - 1) * ld;
        qf[0] = *reinterpret_cast<const f16x8*>(qrow + 8 * h);
        qf[1] = *reinterpret_cast<const f16x8*>(qrow + 16 + 8 * h);
    }
    const f16* bp = bias + ((long)(head * QB + wave) * QB * 64 + lane) * 16;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // loop-invariant LDS offsets: K fragment of d-step s at k_off[s] + kb * 32 rows; V^T transposed reads as in kernels_attn.hip
    // (16-lane group g covers d = 16 (g & 1) .. +15 of key half g >> 1 == h; lane 4 t + p supplies key row t, columns 4 p .. +3)
    int k_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) k_off[s] = r * ROW + ((2 * s + h) ^ ((r >> 2) & 3)) * 16;
    const int v_off = (4 * h + ((lane & 15) >> 2)) * ROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kb = 0; kb < QB; ++kb) {
        const f16x8 b0 = *reinterpret_cast<const f16x8*>(bp + kb * 64 * 16);
        const f16x8 b1 = *reinterpret_cast<const f16x8*>(bp + kb * 64 * 16 + 8);
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(sk + k_off[0] + kb * 32 * ROW), qf[0], zero, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(sk + k_off[1] + kb * 32 * ROW), qf[1], s, 0, 0, 0);
        float mloc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = fmaf(s[e], scale, (float)(e < 8 ? b0[e & 7] : b1[e & 7])); // -inf on keys >= N
            mloc = fmaxf(mloc, s[e]);
        }
        mloc = fmaxf(mloc, wa_other_half(mloc));
        const float m_new = fmaxf(m_run, mloc * WA_LOG2E); // every key block holds at least one real key: finite
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] *= alpha;
            m_run = m_new;
        }
        float psum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[e], WA_LOG2E, -m_run));
            s[e] = p;
            psum += p;
        }
        l_run += psum;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { // element j of the k-step <-> key 32 kb + 16 ks + 8 (j >> 2) + 4 h + (j & 3)
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[8 * ks + j];
            const unsigned char* vrow = sv + v_off + (kb * 32 + ks * 16) * ROW;
            hv4 va = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(vrow));
            hv4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(vrow + 8 * ROW));
            f16x8 vf = {(f16)va[0], (f16)va[1], (f16)va[2], (f16)va[3], (f16)vb[0], (f16)vb[1], (f16)vb[2], (f16)vb[3]};
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
        }
    }

    const float inv = 1.0f / (l_run + wa_other_half(l_run));
    if (q < N) {
        f16* orow = out + ((long)win * N + q) * (heads * WA_HD) + head * WA_HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) { // accumulator rows d = 8 g + 4 h + j
            f16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (f16)(o[g * 4 + j] * inv);
            *reinterpret_cast<f16x4*>(orow + 8 * g + 4 * h) = v;
        }
    }
}

int query_blocks(int N) { // instantiated block shapes
    for (int qb : {1, 2, 4, 7, 8})
        if (N <= qb * 32) return qb;
    return 0;
}

uint16_t to_f16_bits(float f) {
    __fp16 hv = (__fp16)f;
    uint16_t u;
    memcpy(&u, &hv, 2);
    return u;
}

} // namespace

extern "C" {

size_t vx_window_attention_bias_bytes(int N, int heads) {
    const int qb = query_blocks(N);
    return (size_t)heads * qb * qb * 64 * 16 * 2;
}

int vx_window_attention_pack_bias(const float* bias, int N, int heads, void* packed_host) {
    const int QB = query_blocks(N);
    VX_REQUIRE(bias && packed_host && heads > 0 && QB > 0, "vx_window_attention_pack_bias: bad operands (N = %d, at most 256)", N);
    uint16_t* out = static_cast<uint16_t*>(packed_host);
    const uint16_t ninf = 0xFC00;
    for (int head = 0; head < heads; ++head)
        for (int qb = 0; qb < QB; ++qb)
            for (int kb = 0; kb < QB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 16; ++e) {
                        const int q = qb * 32 + (lane & 31), key = kb * 32 + (e >> 2) * 8 + 4 * (lane >> 5) + (e & 3);
                        uint16_t v = key >= N ? ninf : (q < N ? to_f16_bits(bias[((size_t)head * N + q) * N + key]) : (uint16_t)0);
                        out[((((size_t)head * QB + qb) * QB + kb) * 64 + lane) * 16 + e] = v;
                    }
    return 1;
}

int vx_window_attention_f16(const void* qkv, const void* bias_packed, void* out, int n_windows, int N, int heads, void* stream) {
    VX_REQUIRE(qkv && bias_packed && out && n_windows > 0 && heads > 0, "vx_window_attention_f16: bad operands");
    const int QB = query_blocks(N);
    VX_REQUIRE(N > 0 && QB > 0, "vx_window_attention_f16: %d tokens per window (at most 256)", N);
    const f16* q = reinterpret_cast<const f16*>(qkv);
    const f16* b = reinterpret_cast<const f16*>(bias_packed);
    f16* o = reinterpret_cast<f16*>(out);
    const dim3 grid((unsigned)n_windows * heads);
    const float scale = 1.0f / sqrtf((float)WA_HD);
    hipStream_t s = as_stream(stream);
    switch (QB) {
        case 1: hipLaunchKernelGGL(window_attention_kernel<1>, grid, dim3(64), 0, s, q, b, o, N, heads, scale); break;
        case 2: hipLaunchKernelGGL(window_attention_kernel<2>, grid, dim3(128), 0, s, q, b, o, N, heads, scale); break;
        case 4: hipLaunchKernelGGL(window_attention_kernel<4>, grid, dim3(256), 0, s, q, b, o, N, heads, scale); break;
        case 7: hipLaunchKernelGGL(window_attention_kernel<7>, grid, dim3(448), 0, s, q, b, o, N, heads, scale); break;
        default: hipLaunchKernelGGL(window_attention_kernel<8>, grid, dim3(512), 0, s, q, b, o, N, heads, scale); break;
    }
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
