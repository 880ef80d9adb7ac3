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
                                                                   int N, int heads, float scale, int nwx, int nwy, long cls_stride) {
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
    // SWIN shifted windows (swin.cpp:165-213): the windows of the last row / column of an image add a -inf mask between tokens
    // that the cyclic shift brought together; the host packs bias + mask for the four window classes (interior, last column,
    // last row, corner) as four consecutive images (vx_swin_attention_pack_bias). nwx = 0: one image, no classes.
    if (nwx > 0) {
        const int wi = win % (nwx * nwy), wy = wi / nwx, wx = wi - wy * nwx;
        bias += ((wy == nwy - 1 ? 2 : 0) + (wx == nwx - 1 ? 1 : 0)) * cls_stride;
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
    // finite floor instead of -inf: with the SWIN shift masks a whole 32-key block can be masked for a query, and
    // exp2(-inf - (-inf)) would be NaN; with the floor such a block contributes exact zeros
    float m_run = -1e30f, l_run = 0.0f;
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
        const float m_new = fmaxf(m_run, mloc * WA_LOG2E);
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
    return vx_window_attention_masked_f16(qkv, bias_packed, out, n_windows, N, heads, 0, 0, stream);
}

// SWIN: relative_position_bias_table [(2 ws - 1)^2][heads] f32 -> four packed images (bias, bias + last-column mask, bias +
// last-row mask, bias + corner mask) of vx_window_attention_bias_bytes(ws*ws, heads) bytes each. The bias is rounded to f16
// first, as the reference casts it (swin.cpp:88-92); index = compute_relative_position_index (swin.cpp:26-38).
int vx_swin_attention_pack_bias(const float* table, int ws, int heads, void* packed_host) {
    VX_REQUIRE(table && packed_host && ws > 0 && ws <= 16 && heads > 0, "vx_swin_attention_pack_bias: bad operands (window %d)", ws);
    const int N = ws * ws, shift = ws / 2;
    const size_t img = vx_window_attention_bias_bytes(N, heads);
    float* b = new float[(size_t)heads * N * N];
    for (int cls = 0; cls < 4; ++cls) {
        for (int i = 0; i < N; ++i)       // query (y1, x1) = slow index of the reference's table, key (y0, x0) = fast index
            for (int j = 0; j < N; ++j) {
                const int y1 = i / ws, x1 = i % ws, y0 = j / ws, x0 = j % ws;
                const int idx = (y1 - y0 + ws - 1) * (2 * ws - 1) + (x1 - x0 + ws - 1);
                // inside the last window row / column, positions >= ws - shift come from the other side of the image
                const bool cut_y = (cls & 2) && ((y0 < ws - shift) != (y1 < ws - shift));
                const bool cut_x = (cls & 1) && ((x0 < ws - shift) != (x1 < ws - shift));
                for (int h = 0; h < heads; ++h)
                    b[((size_t)h * N + i) * N + j] = (cut_y || cut_x) ? -INFINITY : (float)(__fp16)table[(size_t)idx * heads + h];
            }
        if (!vx_window_attention_pack_bias(b, N, heads, static_cast<unsigned char*>(packed_host) + cls * img)) { delete[] b; return 0; }
    }
    delete[] b;
    return 1;
}

int vx_window_attention_masked_f16(const void* qkv, const void* bias_packed, void* out, int n_windows, int N, int heads, int nwx, int nwy,
                                   void* stream) {
    VX_REQUIRE(qkv && bias_packed && out && n_windows > 0 && heads > 0, "vx_window_attention_f16: bad operands");
    VX_REQUIRE((nwx == 0 && nwy == 0) || (nwx > 0 && nwy > 0 && n_windows % (nwx * nwy) == 0), "vx_window_attention_masked_f16: %d windows are not whole images of %dx%d", n_windows, nwx, nwy);
    const long cls_stride = (long)(vx_window_attention_bias_bytes(N, heads) / 2);
    const int QB = query_blocks(N);
    VX_REQUIRE(N > 0 && QB > 0, "vx_window_attention_f16: %d tokens per window (at most 256)", N);
    const f16* q = reinterpret_cast<const f16*>(qkv);
    const f16* b = reinterpret_cast<const f16*>(bias_packed);
    f16* o = reinterpret_cast<f16*>(out);
    const dim3 grid((unsigned)n_windows * heads);
    const float scale = 1.0f / sqrtf((float)WA_HD);
    hipStream_t s = as_stream(stream);
    switch (QB) {
        case 1: hipLaunchKernelGGL(window_attention_kernel<1>, grid, dim3(64), 0, s, q, b, o, N, heads, scale, nwx, nwy, cls_stride); break;
        case 2: hipLaunchKernelGGL(window_attention_kernel<2>, grid, dim3(128), 0, s, q, b, o, N, heads, scale, nwx, nwy, cls_stride); break;
        case 4: hipLaunchKernelGGL(window_attention_kernel<4>, grid, dim3(256), 0, s, q, b, o, N, heads, scale, nwx, nwy, cls_stride); break;
        case 7: hipLaunchKernelGGL(window_attention_kernel<7>, grid, dim3(448), 0, s, q, b, o, N, heads, scale, nwx, nwy, cls_stride); break;
        default: hipLaunchKernelGGL(window_attention_kernel<8>, grid, dim3(512), 0, s, q, b, o, N, heads, scale, nwx, nwy, cls_stride); break;
    }
    VX_LAUNCH_CHECK();
    return 1;
}

} // extern "C"
