// f16 MFMA GEMM family for gfx950: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, plus the
// implicit-GEMM form of the NHWC 3x3 convolutions. One kernel template covers SURVEY.md rows
// K1, K4, K6-K11, K14 (reference call sites: src/visp/nn.cpp:6-12, 72-100, 117-129;
// src/visp/arch/dino.cpp:48-90; src/visp/arch/depth-anything.cpp:15-96).
//
// Tiling: 256 threads = 4 wave64; block tile BM x BN x 64, wave tile WM x WN built from
// v_mfma_f32_32x32x16_f16 (A and B fragments both K-contiguous: lane l holds row l&31,
// k = 8*(l>>5)..+7). Both operands are staged through LDS in 128-byte rows whose 16-byte
// chunks are XOR-swizzled with (row>>1)&7 so that ds_read_b128 of 32 different rows at one
// k-column is bank-conflict free. Global->LDS staging is register double-buffered: loads of
// k-tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after
// them (one barrier per k-tile).
#include "vx_common.h"

namespace {

constexpr int BK = 64;          // k elements per LDS tile row (128 bytes)
constexpr int THREADS = 256;

struct ConvRow {                // per staged A row of an implicit-GEMM conv
    int base;                   // element offset of (b, 0, 0, 0) in the NHWC image, -1 if row >= M
    int iy0, ix0;               // oy*stride - pad, ox*stride - pad
};

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ float gelu_tanh(float x) {
    // ggml_gelu: 0.5*x*(1+tanh(sqrt(2/pi)*x*(1+0.044715*x*x)))
    const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
    float u = k0 * x * (1.0f + k1 * x * x);
    // tanh(u) = 1 - 2/(exp(2u)+1); exp via exp2
    float e = __builtin_amdgcn_exp2f(u * 2.88539008177792681472f); // 2*log2(e)
    float t = 1.0f - 2.0f / (e + 1.0f);
    return 0.5f * x * (1.0f + t);
}

template <int BM, int BN, int WM, int WN, int EPI, bool CONV>
__global__ __launch_bounds__(THREADS) void gemm_kernel(const vx_gemm_args p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_CHUNKS = BM * 8 / THREADS; // 16-byte chunks per thread per k-tile
    constexpr int B_CHUNKS = BN * 8 / THREADS;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1, "tile too small");
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int r = lane & 31, h = lane >> 5;

    const int n_tiles_n = p.N / BN;
    const int tile_m = blockIdx.x / n_tiles_n, tile_n = blockIdx.x % n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BK;

    const f16* __restrict__ Ag = reinterpret_cast<const f16*>(p.A);
    const f16* __restrict__ Wg = reinterpret_cast<const f16*>(p.W);

    // ---- per-thread staging coordinates -------------------------------------------------
    int a_row[A_CHUNKS], a_ch[A_CHUNKS];
    long a_off[A_CHUNKS];           // plain mode: element offset of the row start
    ConvRow a_conv[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        int idx = tid + i * THREADS;
        a_row[i] = idx >> 3;
        a_ch[i] = idx & 7;
        int m = m0 + a_row[i];
        if constexpr (CONV) {
            if (m < p.M) {
                int ohw = p.conv_OH * p.conv_OW;
                int b = m / ohw, rem = m - b * ohw;
                int oy = rem / p.conv_OW, ox = rem - oy * p.conv_OW;
                a_conv[i].base = b * p.conv_H * p.conv_W * p.conv_Cin;
                a_conv[i].iy0 = oy * p.conv_stride - p.conv_pad;
                a_conv[i].ix0 = ox * p.conv_stride - p.conv_pad;
            } else {
                a_conv[i].base = -1;
                a_conv[i].iy0 = a_conv[i].ix0 = 0;
            }
            a_off[i] = 0;
        } else {
            if (m >= p.M) m = p.M - 1; // clamp: rows beyond M are computed but never stored
            long arow = m;
            if (p.a_group > 0) arow = (long)(m / p.a_group) * p.a_group_stride + p.a_row_off + (m % p.a_group);
            a_off[i] = arow * p.lda;
        }
    }
    int b_row[B_CHUNKS], b_ch[B_CHUNKS];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
        int idx = tid + i * THREADS;
        b_row[i] = idx >> 3;
        b_ch[i] = idx & 7;
    }

    f16x8 a_reg[A_CHUNKS], b_reg[B_CHUNKS];
    const int cin8 = CONV ? (p.conv_Cin >> 3) : 1;
    const int ntaps = CONV ? p.conv_kh * p.conv_kw : 1;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            if constexpr (CONV) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                int kc = (k0 >> 3) + a_ch[i];
                int tap = kc / cin8, c8 = kc - tap * cin8;
                if (a_conv[i].base >= 0 && tap < ntaps) {
                    int ky = tap / p.conv_kw, kx = tap - ky * p.conv_kw;
                    int iy = a_conv[i].iy0 + ky, ix = a_conv[i].ix0 + kx;
                    if ((unsigned)iy < (unsigned)p.conv_H && (unsigned)ix < (unsigned)p.conv_W) {
                        v = *reinterpret_cast<const f16x8*>(Ag + a_conv[i].base + ((long)iy * p.conv_W + ix) * p.conv_Cin + c8 * 8);
                        if (p.a_relu) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = v[j] > (f16)0 ? v[j] : (f16)0;
                        }
                    }
                }
                a_reg[i] = v;
            } else {
                a_reg[i] = *reinterpret_cast<const f16x8*>(Ag + a_off[i] + k0 + a_ch[i] * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            b_reg[i] = *reinterpret_cast<const f16x8*>(Wg + (long)(n0 + b_row[i]) * p.K + k0 + b_ch[i] * 8);
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE_BYTES;
        unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            *reinterpret_cast<f16x8*>(sa + a_row[i] * 128 + swz_chunk(a_row[i], a_ch[i]) * 16) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            *reinterpret_cast<f16x8*>(sb + b_row[i] * 128 + swz_chunk(b_row[i], b_ch[i]) * 16) = b_reg[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* sa = smem + buf * STAGE_BYTES;
        const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int row = wr * WM + mi * 32 + r;
                af[mi] = *reinterpret_cast<const f16x8*>(sa + row * 128 + swz_chunk(row, ks * 2 + h) * 16);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                int row = wc * WN + ni * 32 + r;
                bf[ni] = *reinterpret_cast<const f16x8*>(sb + row * 128 + swz_chunk(row, ks * 2 + h) * 16);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ------
    const int n_valid = p.n_valid > 0 ? p.n_valid : p.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wc * WN + ni * 32 + r;
            const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wr * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[mi][ni][e] + bias;
                if constexpr (EPI == VX_EPI_F16 || EPI == VX_EPI_F16_GELU || EPI == VX_EPI_F16_RELU) {
                    if constexpr (EPI == VX_EPI_F16_GELU) v = gelu_tanh(v);
                    if constexpr (EPI == VX_EPI_F16_RELU) v = fmaxf(v, 0.0f);
                    if (n < n_valid) reinterpret_cast<f16*>(p.out)[(long)m * p.ldo + n] = (f16)v;
                } else if constexpr (EPI == VX_EPI_F16_ADD) {
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (n < n_valid) {
                        long o = (long)m * p.ldo + n;
                        if (p.res1) v += (float)reinterpret_cast<const f16*>(p.res1)[o];
                        if (p.res2) v += (float)reinterpret_cast<const f16*>(p.res2)[o];
                        reinterpret_cast<f16*>(p.out)[o] = (f16)v;
                    }
                } else if constexpr (EPI == VX_EPI_RESID_F32) {
                    float* x = reinterpret_cast<float*>(p.out) + (long)m * p.ldo + n;
                    *x = *x + v * p.lambda[n];
                } else if constexpr (EPI == VX_EPI_TOKENS) {
                    int b = m / p.tokens_P, t = m - b * p.tokens_P;
                    long row = (long)b * (p.tokens_P + 1) + 1 + t;
                    reinterpret_cast<float*>(p.out)[row * p.ldo + n] = v + p.pos[(long)(1 + t) * p.N + n];
                } else if constexpr (EPI == VX_EPI_QKV) {
                    const int C = p.qkv_H * 64;
                    int which = n / C, cc = n - which * C;
                    int hh = cc >> 6, d = cc & 63;
                    int b = m / p.qkv_T, t = m - b * p.qkv_T;
                    long bh = (long)b * p.qkv_H + hh;
                    if (which == 0) reinterpret_cast<f16*>(p.q)[(bh * p.qkv_T + t) * 64 + d] = (f16)(v * p.q_scale);
                    else if (which == 1) reinterpret_cast<f16*>(p.k)[(bh * p.qkv_T + t) * 64 + d] = (f16)v;
                    else reinterpret_cast<f16*>(p.vt)[(bh * 64 + d) * p.qkv_Tp + t] = (f16)v;
                } else if constexpr (EPI == VX_EPI_PIXSHUF) {
                    if (n < n_valid) {
                        int s = p.ps_s;
                        int tap = n / p.ps_Cout, co = n - tap * p.ps_Cout;
                        int dy = tap / s, dx = tap - dy * s;
                        int hw = p.ps_H * p.ps_W;
                        int b = m / hw, rem = m - b * hw;
                        int y = rem / p.ps_W, x = rem - y * p.ps_W;
                        long o = (((long)b * p.ps_H * s + (y * s + dy)) * (p.ps_W * s) + (x * s + dx)) * p.ldo + co;
                        reinterpret_cast<f16*>(p.out)[o] = (f16)v;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI, bool CONV>
int launch(const vx_gemm_args& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_kernel<BM, BN, WM, WN, EPI, CONV>;
    static bool attr_set = false;
    if (!attr_set && smem > 48 * 1024) {
        VX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(THREADS), smem, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

template <int EPI, bool CONV>
int dispatch_tile(const vx_gemm_args& a, hipStream_t s) {
    if (a.N % 128 == 0) return launch<128, 128, 64, 64, EPI, CONV>(a, s);
    if (a.N % 64 == 0) return launch<128, 64, 64, 32, EPI, CONV>(a, s);
    return launch<128, 32, 32, 32, EPI, CONV>(a, s);
}

template <bool CONV>
int dispatch_epi(const vx_gemm_args& a, hipStream_t s) {
    switch (a.epi) {
        case VX_EPI_F16: return dispatch_tile<VX_EPI_F16, CONV>(a, s);
        case VX_EPI_F16_GELU: return dispatch_tile<VX_EPI_F16_GELU, CONV>(a, s);
        case VX_EPI_F16_RELU: return dispatch_tile<VX_EPI_F16_RELU, CONV>(a, s);
        case VX_EPI_F16_ADD: return dispatch_tile<VX_EPI_F16_ADD, CONV>(a, s);
        default: break;
    }
    if constexpr (!CONV) {
        switch (a.epi) {
            case VX_EPI_RESID_F32: return dispatch_tile<VX_EPI_RESID_F32, false>(a, s);
            case VX_EPI_TOKENS: return dispatch_tile<VX_EPI_TOKENS, false>(a, s);
            case VX_EPI_QKV: return dispatch_tile<VX_EPI_QKV, false>(a, s);
            case VX_EPI_PIXSHUF: return dispatch_tile<VX_EPI_PIXSHUF, false>(a, s);
            default: break;
        }
    }
    vx_set_error("vx_gemm_f16: unsupported epilogue %d (conv=%d)", a.epi, (int)CONV);
    return 0;
}

} // namespace

extern "C" int vx_gemm_f16(const vx_gemm_args* args, void* stream) {
    const vx_gemm_args& a = *args;
    VX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "vx_gemm_f16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    VX_REQUIRE(a.K % BK == 0, "vx_gemm_f16: K=%d must be a multiple of %d (pad the weights)", a.K, BK);
    VX_REQUIRE(a.N % 32 == 0, "vx_gemm_f16: N=%d must be a multiple of 32", a.N);
    VX_REQUIRE(a.A && a.W, "vx_gemm_f16: null operand");
    if (a.conv_kh > 0) {
        VX_REQUIRE(a.conv_Cin % 8 == 0, "vx_gemm_f16: conv Cin=%d must be a multiple of 8", a.conv_Cin);
        VX_REQUIRE(a.K >= a.conv_kh * a.conv_kw * a.conv_Cin, "vx_gemm_f16: conv K too small");
        VX_REQUIRE((long)a.M * 1 <= 0x7fffffffL && (long)a.conv_H * a.conv_W * a.conv_Cin * ((a.M + a.conv_OH * a.conv_OW - 1) / (a.conv_OH * a.conv_OW)) < 0x7fffffffL,
                   "vx_gemm_f16: conv image too large for 32-bit offsets");
        return dispatch_epi<true>(a, as_stream(stream));
    }
    VX_REQUIRE(a.lda % 8 == 0, "vx_gemm_f16: lda=%ld must be a multiple of 8", (long)a.lda);
    return dispatch_epi<false>(a, as_stream(stream));
}
