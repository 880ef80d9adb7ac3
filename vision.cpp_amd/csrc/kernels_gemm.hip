// f16 MFMA GEMM family for gfx950: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, plus the
// implicit-GEMM form of the NHWC 3x3 convolutions. One kernel template covers SURVEY.md rows
// K1, K4, K6-K11, K14 (reference call sites: src/visp/nn.cpp:6-12, 72-100, 117-129;
// src/visp/arch/dino.cpp:48-90; src/visp/arch/depth-anything.cpp:15-96).
//
// Structure (256 threads = 4 wave64 per block, block tile BM x BN x 64):
//  * both operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
//    instruction = 8 tile rows of 128 B); no staging registers, so 3-4 blocks fit a CU and
//    one block's load latency / epilogue hides under the others' MFMA phases (K is only
//    384..1536 on this path, so prologue and epilogue are a large share of a tile's life);
//  * the LDS image is lane-linear, the (row>>1)&7 XOR swizzle of the 16-byte chunks is applied
//    to the per-lane SOURCE address and again on the ds_read_b128 fragment reads, which makes
//    the reads of 32 different rows at one k-column bank-conflict free;
//  * v_mfma_f32_32x32x16_f16, A and B fragments both K-contiguous (lane l: row l&31,
//    k = 8*(l>>5)..+7); wave tile WM x WN;
//  * every epilogue runs the MFMA with swapped operands (D = W-frag x A-frag, so a lane owns one
//    output row and 4 consecutive columns per register group): f16 tiles are staged through LDS
//    and written with coalesced 16-byte stores, the f32 read-modify-write epilogues (residual
//    stream, token rows) use batched 16-byte loads/stores straight from the registers.
#include "vx_common.h"

#include <algorithm>

extern "C" int vx_gemm_f16_raw(const vx_gemm_args* args, void* stream); // the single launch behind vx_gemm_f16

namespace {

constexpr int BK = 64;          // k elements per LDS tile row (128 bytes)
constexpr int WAVE = 64;

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[64]; // source of padded conv taps

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ float gelu_tanh(float x) {
    // ggml_gelu: 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*x*(1+0.044715*x*x). With tanh(u) = 2*sigmoid(2u) - 1
    // this is x * sigmoid(2u) = x / (1 + exp(-2u)): 7 VALU ops, two of them transcendental.
    const float c1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f; // -2*sqrt(2/pi)*log2(e)
    const float c3 = c1 * 0.044715f;
    float w = fmaf(x * x, c3, c1);
    float e = __builtin_amdgcn_exp2f(x * w);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr bool epi_is_f16_tile(int epi) {
    return epi == VX_EPI_F16 || epi == VX_EPI_F16_GELU || epi == VX_EPI_F16_RELU || epi == VX_EPI_F16_ADD ||
           epi == VX_EPI_QKV || epi == VX_EPI_PIXSHUF || epi == VX_EPI_HEAD_OUT;
}

template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool CONV>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void gemm_kernel(const vx_gemm_args p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int NWAVES = (BM / WM) * WAVES_N;
    constexpr int THREADS = WAVE * NWAVES;
    constexpr int A_INSTR = BM / (8 * NWAVES); // global_load_lds instructions per wave per k-tile (8 rows each)
    constexpr int B_INSTR = BN / (8 * NWAVES);
    static_assert(A_INSTR >= 1 && B_INSTR >= 1 && (128 * (BN / 8)) % THREADS == 0 && THREADS >= BM, "tile / wave layout");

    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int LOADS = A_INSTR + B_INSTR; // global_load_lds per wave per k-tile (vmcnt units)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // optional phase stamps (diagnostic runs only: args.debug_stamps is null in the product path)
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(p.debug_stamps);
    auto stamp = [&](int slot) {
        if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime(); // 100 MHz reference
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give every XCD a contiguous
    // run of logical tiles so that the n-tiles of one A row panel are fetched through one L2.
    const int n_tiles_n = p.N / BN;
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = b & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    // split-K: the grid holds k_splits copies of the tile set, copy ks reduces k-tiles [kbase, kbase + nk)
    int ks = 0;
    if (p.k_splits > 1) {
        const int n_tiles = gridDim.x / p.k_splits;
        ks = tile / n_tiles;
        tile -= ks * n_tiles;
    }
    const int tile_m = tile / n_tiles_n, tile_n = tile - tile_m * n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    int kbase = 0, nk = p.K / BK;
    if (p.k_splits > 1) {
        const int per = (nk + p.k_splits - 1) / p.k_splits;
        kbase = ks * per;
        nk = min(per, nk - kbase); // >= 1: the host picks k_splits so that no range is empty
    }

    // per-column epilogue vectors (bias, LayerScale lambda) are fetched ONCE, now, into a small LDS side
    // buffer behind the operand ring: their L2 latency hides under the k-loop instead of stalling the
    // epilogue (16 dependent float4 loads per lane measured 10-30k cycles there)
    float* const s_bias = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);
    float* const s_lambda = s_bias + BN;
    int* const s_pix = reinterpret_cast<int*>(s_lambda + BN); // VX_EPI_F16_ADD with win_ws: pixel row of every tile row, -1 = dropped
    if constexpr (EPI == VX_EPI_F16_ADD) {
        // window-order row m -> pixel row (window_reverse), once per tile row instead of six integer divisions per 16-byte chunk
        if (p.win_ws > 0 && tid < BM) {
            const int m = m0 + tid;
            int pix = -1;
            if (m < p.M) {
                const int ws = p.win_ws, rw = p.win_res, rh = p.win_res_h > 0 ? p.win_res_h : p.win_res, N = ws * ws;
                const int nwx = (rw + ws - 1) / ws, nwy = (rh + ws - 1) / ws;
                const int in = m % N;
                int wq = m / N;
                const int wx = wq % nwx;
                wq /= nwx;
                const int wy = wq % nwy, bb = wq / nwy;
                int py = wy * ws + in / ws + p.win_shift, px = wx * ws + in % ws + p.win_shift; // roll(+shift) of the padded map
                if (py >= nwy * ws) py -= nwy * ws;
                if (px >= nwx * ws) px -= nwx * ws;
                if (py < rh && px < rw) pix = (bb * rh + py) * rw + px;
            }
            s_pix[tid] = pix;
        }
    }
    if (tid < BN) {
        s_bias[tid] = p.bias ? p.bias[n0 + tid] : 0.0f;
        if constexpr (EPI == VX_EPI_RESID_F32 || EPI == VX_EPI_HEAD_OUT) s_lambda[tid] = p.lambda[n0 + tid];
    }

    const f16* __restrict__ Ag = reinterpret_cast<const f16*>(p.A);
    const f16* __restrict__ Wg = reinterpret_cast<const f16*>(p.W);

    // ---- per-lane source coordinates: instruction i of this wave fills tile rows (wave*A_INSTR+i)*8 .. +7,
    // lane l lands at row + l/8, physical chunk l%8, so it must fetch logical chunk (l%8) ^ swz(row)
    const int l_row = lane >> 3, l_pos = lane & 7;
    long a_off[A_INSTR];   // plain: element offset of the row start; conv: image base
    int a_iy0[A_INSTR], a_ix0[A_INSTR], a_chunk[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (wave * A_INSTR + i) * 8 + l_row;
        a_chunk[i] = swz_chunk(row, l_pos);
        int m = m0 + row;
        if constexpr (CONV) {
            if (m < p.M) {
                int ohw = p.conv_OH * p.conv_OW;
                int b = m / ohw, rm = m - b * ohw;
                int oy = rm / p.conv_OW, ox = rm - oy * p.conv_OW;
                a_off[i] = (long)b * p.conv_H * p.conv_W * p.conv_Cin;
                a_iy0[i] = oy * p.conv_stride - p.conv_pad;
                a_ix0[i] = ox * p.conv_stride - p.conv_pad;
            } else {
                a_off[i] = -1;
                a_iy0[i] = a_ix0[i] = 0;
            }
        } else {
            if (m >= p.M) m = p.M - 1; // rows beyond M are computed but never stored
            long arow = m;
            if (p.a_group > 0) arow = (long)(m / p.a_group) * p.a_group_stride + p.a_row_off + (m % p.a_group);
            a_off[i] = arow * p.lda;
            a_iy0[i] = a_ix0[i] = 0;
        }
    }
    long b_off[B_INSTR];
    int b_chunk[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave * B_INSTR + i) * 8 + l_row;
        b_chunk[i] = swz_chunk(row, l_pos);
        b_off[i] = (long)(n0 + row) * p.K;
    }
    const int cin8 = CONV ? (p.conv_Cin >> 3) : 1;
    const int ntaps = CONV ? p.conv_kh * p.conv_kw : 1;

    auto issue_loads = [&](int kt, int stage) {
        unsigned char* const sa = smem + stage * STAGE_BYTES;
        unsigned char* const sb = sa + BM * BK * 2;
        const int k0 = (kbase + kt) * BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const f16* src;
            if constexpr (CONV) {
                src = reinterpret_cast<const f16*>(g_zero_page);
                int kc = (k0 >> 3) + a_chunk[i];
                int tap = kc / cin8, c8 = kc - tap * cin8;
                if (a_off[i] >= 0 && tap < ntaps) {
                    int ky = tap / p.conv_kw, kx = tap - ky * p.conv_kw;
                    int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                    if ((unsigned)iy < (unsigned)p.conv_H && (unsigned)ix < (unsigned)p.conv_W)
                        src = Ag + a_off[i] + ((long)iy * p.conv_W + ix) * p.conv_Cin + c8 * 8;
                }
            } else {
                src = Ag + a_off[i] + k0 + a_chunk[i] * 8;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + (wave * A_INSTR + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(Wg + b_off[i] + k0 + b_chunk[i] * 8),
                                             (lptr_t)(sb + (wave * B_INSTR + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

    // fragment read offsets: the swizzle term (row>>1)&7 only depends on the lane (tile row offsets are
    // multiples of 32), so 4 per-lane offsets cover every (mi / ni, k-step) through immediates
    int f_off[BK / 16];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) f_off[ks] = r * 128 + swz_chunk(r, ks * 2 + h) * 16;

    auto compute = [&](int stage) {
        const unsigned char* const sa = smem + stage * STAGE_BYTES + wr * WM * 128;
        const unsigned char* const sb = smem + stage * STAGE_BYTES + BM * BK * 2 + wc * WN * 128;
        // fragments are double-buffered in registers: the ds_reads of k-step ks+1 are in flight while the
        // MFMAs of k-step ks issue (the compiler keeps one register set and serialises them otherwise)
        f16x8 af[2][MI], bf[2][NI];
        auto load_frags = [&](int ks, int set) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[set][mi] = *reinterpret_cast<const f16x8*>(sa + f_off[ks] + mi * 32 * 128);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[set][ni] = *reinterpret_cast<const f16x8*>(sb + f_off[ks] + ni * 32 * 128);
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < BK / 16) load_frags(ks + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0); // keep the prefetch ahead of this step's MFMAs
            if constexpr (CONV) {
                if (p.a_relu) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < 8; ++j) af[set][mi][j] = af[set][mi][j] > (f16)0 ? af[set][mi][j] : (f16)0;
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
        }
    };
    // raw barrier: __syncthreads() would drain the LDS-DMA queue (vmcnt(0)) and serialise the ring
    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    stamp(1);
    if constexpr (STAGES == 1) {
        issue_loads(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier(); // k-tile kt has landed in LDS for every wave
            if (kt == 0) stamp(2);
            compute(0);
            barrier(); // every wave is done reading this k-tile
            if (kt + 1 < nk) issue_loads(kt + 1, 0);
        }
    } else {
        // ring of STAGES k-tiles, STAGES-1 of them in flight; one barrier per k-tile
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) issue_loads(s, s);
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(STAGES - 2, nk - 1 - kt); // tiles that may stay in flight behind tile kt
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier(); // tile kt visible to all waves; all waves are done with tile kt-1, its stage is free
            int next_stage = stage == 0 ? STAGES - 1 : stage - 1; // == (kt + STAGES - 1) % STAGES
            if (kt + STAGES - 1 < nk) issue_loads(kt + STAGES - 1, next_stage);
            compute(stage);
            stage = stage + 1 == STAGES ? 0 : stage + 1;
        }
        barrier(); // the epilogue reuses the ring as its staging buffer
    }

    const int n_valid = p.n_valid > 0 ? p.n_valid : p.N;
    stamp(3);

    if (p.k_splits > 1) { // raw partial sums: a lane owns row m and 4 consecutive columns per register group
        float* const part = p.k_partial + (long)ks * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wr * WM + mi * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = {acc[mi][ni][4 * g + 0], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                    *reinterpret_cast<float4*>(part + (long)m * p.N + n0 + wc * WN + ni * 32 + 8 * g + 4 * h) = v;
                }
        }
        return;
    }

    if constexpr (!epi_is_f16_tile(EPI)) {
        // ---- f32 outputs (residual stream x): a lane owns row m and 4 consecutive columns per register
        // group, so the read-modify-write runs on 16-byte accesses (16 + 16 per lane instead of 64 + 64)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wr * WM + mi * 32 + r;
            if (m >= p.M) continue;
            long row = m;
            int t = 0;
            if constexpr (EPI == VX_EPI_TOKENS) {
                int b = m / p.tokens_P;
                t = m - b * p.tokens_P;
                row = (long)b * (p.tokens_P + 1) + 1 + t;
            }
            float* xrow = reinterpret_cast<float*>(p.out) + row * p.ldo + n0;
            const float* add_row = nullptr; // the tensor read-modify-written with the result
            if constexpr (EPI == VX_EPI_RESID_F32) add_row = xrow;
            else add_row = p.pos + (long)(1 + t) * p.N + n0;
            // issue all NI*4 row loads first (one latency, not NI*4 dependent ones), then combine and store
            float4 xin[NI][4];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xin[ni][g] = *reinterpret_cast<const float4*>(add_row + wc * WN + ni * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wc * WN + ni * 32 + 8 * g + 4 * h;
                    const float4 bias = *reinterpret_cast<const float4*>(s_bias + nl);
                    float4 v = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y,
                                acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
                    float4 x = xin[ni][g];
                    if constexpr (EPI == VX_EPI_RESID_F32) {
                        const float4 lam = *reinterpret_cast<const float4*>(s_lambda + nl);
                        x.x += v.x * lam.x; x.y += v.y * lam.y; x.z += v.z * lam.z; x.w += v.w * lam.w;
                    } else { // VX_EPI_TOKENS: patch token + position embedding
                        x.x += v.x; x.y += v.y; x.z += v.z; x.w += v.w;
                    }
                    *reinterpret_cast<float4*>(xrow + nl) = x;
                }
            }
        }
    } else {
        // ---- swapped orientation: lane&31 = row m, registers = 4 consecutive columns per group.
        // Phase 1: bias / activation, pack to f16, stage the BM x BN tile in LDS (16-byte chunks
        // XOR-swizzled with the row so both the 8-byte writes and the 16-byte reads are conflict free).
        constexpr int NCH16 = BN / 8;              // 16-byte chunks per staged row
        constexpr int PITCH = BN * 2;
        unsigned char* const st = smem;            // the operand stage is dead after the last barrier
        // QKV: a 128-column tile lies entirely inside q, k or v (C % 128 == 0), so the q scale is block uniform
        float out_scale = 1.0f;
        if constexpr (EPI == VX_EPI_QKV) out_scale = n0 < p.qkv_H * 64 ? p.q_scale : 1.0f;
        // The staging buffer holds 128 rows; taller block tiles are written in BM/128 passes, pass ps
        // taking MI/PASSES of every wave's row blocks (staged row sl <-> tile row wr*WM + ps*WPP + sl%WPP).
        constexpr int MPP = (MI % 2 == 0 && BM / WM <= 2) ? 2 : 1; // row blocks per wave per pass (staging <= 128 rows)
        constexpr int PASSES = MI / MPP;
        constexpr int WPP = 32 * MPP;               // rows per wave per pass
        constexpr int STAGE_ROWS = (BM / WM) * WPP;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
        if (ps > 0) __syncthreads();                // previous pass fully read out
#pragma unroll
        for (int mp = 0; mp < MPP; ++mp) {
            const int mi = ps * MPP + mp;
            const int ml = wr * WPP + mp * 32 + r;  // staged row
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wc * WN + ni * 32 + 8 * g + 4 * h; // first of 4 consecutive columns
                    const float4 bias = *reinterpret_cast<const float4*>(s_bias + nl);
                    float v[4] = {acc[mi][ni][4 * g + 0] + bias.x, acc[mi][ni][4 * g + 1] + bias.y,
                                  acc[mi][ni][4 * g + 2] + bias.z, acc[mi][ni][4 * g + 3] + bias.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (EPI == VX_EPI_F16_GELU) v[j] = gelu_tanh(v[j]);
                        if constexpr (EPI == VX_EPI_F16_RELU || EPI == VX_EPI_HEAD_OUT) v[j] = fmaxf(v[j], 0.0f);
                        if constexpr (EPI == VX_EPI_F16_ADD) { if (p.relu) v[j] = fmaxf(v[j], 0.0f); }
                        if constexpr (EPI == VX_EPI_QKV) v[j] *= out_scale;
                    }
                    f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    const int c8 = nl >> 2;
                    const int phys16 = (c8 >> 1) ^ (ml & (NCH16 - 1));
                    *reinterpret_cast<f16x4*>(st + ml * PITCH + phys16 * 16 + (c8 & 1) * 8) = o;
                }
            }
        }
        __syncthreads();
        if (ps == 0) stamp(4);
        // Phase 2: coalesced 16-byte stores (consecutive lanes = consecutive chunks of one row)
        constexpr int CHUNKS = STAGE_ROWS * NCH16;
        static_assert(CHUNKS % THREADS == 0, "staging read-out");
#pragma unroll
        for (int it = 0; it < CHUNKS / THREADS; ++it) {
            const int id = tid + it * THREADS;
            const int ml = id / NCH16, j = id % NCH16;
            const int m = m0 + (ml / WPP) * WM + ps * WPP + (ml % WPP), n = n0 + j * 8;
            if (m >= p.M || n >= n_valid) continue;
            f16x8 v = *reinterpret_cast<const f16x8*>(st + ml * PITCH + (j ^ (ml & (NCH16 - 1))) * 16);
            if constexpr (EPI == VX_EPI_HEAD_OUT) {
                // 1x1 conv to one channel: the NCH16 (= 4) consecutive lanes of a row each dot 8 channels
                float part = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q) part += (float)v[q] * s_lambda[j * 8 + q];
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                if (j == 0) reinterpret_cast<float*>(p.out)[m] = fmaxf(part + p.head_bias, 0.0f) * p.head_scale;
            } else if constexpr (EPI == VX_EPI_F16 || EPI == VX_EPI_F16_GELU || EPI == VX_EPI_F16_RELU) {
                *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + (long)m * p.ldo + n) = v;
            } else if constexpr (EPI == VX_EPI_F16_ADD) {
                long o = (long)m * p.ldo + n;
                if (p.win_ws > 0) { // window_reverse: rows of the zero padding are dropped
                    const int pix = s_pix[m - m0];
                    if (pix < 0) continue;
                    o = (long)pix * p.ldo + n;
                }
                if (p.res1) {
                    f16x8 a = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.res1) + o);
                    if (p.res2) {
                        f16x8 b = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.res2) + o);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (f16)((float)v[q] + (float)a[q] + (float)b[q]);
                    } else if (p.post_gelu) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (f16)gelu_tanh((float)v[q] + (float)a[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (f16)((float)v[q] + (float)a[q]);
                    }
                }
                *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + o) = v;
            } else if constexpr (EPI == VX_EPI_QKV) {
                // q, k, v all head-major [B, H, T, 64]; a 16-byte chunk is 8 consecutive d of one head
                // block-uniform: which of q/k/v (tile inside one of them) and the first image of the tile;
                // a 128-row tile spans at most two images when T >= 128 (else fall back to a division)
                const int C = p.qkv_H * 64;
                const int which = n0 / C, cc = n - which * C;
                const int hh = cc >> 6, d = cc & 63;
                const int b0 = m0 / p.qkv_T;
                int b = b0 + (m >= (b0 + 1) * p.qkv_T ? 1 : 0);
                if (p.qkv_T < BM) b = m / p.qkv_T; // (more than two images per tile)
                const int t = m - b * p.qkv_T;
                f16* dst = reinterpret_cast<f16*>(which == 0 ? p.q : (which == 1 ? p.k : p.vt));
                *reinterpret_cast<f16x8*>(dst + (((long)b * p.qkv_H + hh) * p.qkv_T + t) * 64 + d) = v;
            } else if constexpr (EPI == VX_EPI_PIXSHUF) {
                const int s = p.ps_s;
                const int tap = n / p.ps_Cout, co = n - tap * p.ps_Cout;
                const int dy = tap / s, dx = tap - dy * s;
                const int hw = p.ps_H * p.ps_W;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.ps_W, x = rem - y * p.ps_W;
                const long o = (((long)b * p.ps_H * s + (y * s + dy)) * (p.ps_W * s) + (x * s + dx)) * p.ldo + co;
                *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.out) + o) = v;
            }
        }
        } // passes
    }
    stamp(5);
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}

template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool CONV>
int launch(const vx_gemm_args& a, hipStream_t s) {
    constexpr int ring = STAGES * (BM + BN) * BK * 2;
    constexpr int out_stage = epi_is_f16_tile(EPI) ? 128 * BN * 2 : 0;
    constexpr int smem = (ring > out_stage ? ring : out_stage) + 2 * BN * 4 + (EPI == VX_EPI_F16_ADD ? BM * 4 : 0); // + bias / lambda (+ pixel row) side buffers
    auto kern = gemm_kernel<BM, BN, WM, WN, STAGES, EPI, CONV>;
    if constexpr (smem > 48 * 1024) VX_CHECK(vx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem)); // per (kernel, device)
    int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n * (a.k_splits > 1 ? a.k_splits : 1)), dim3(64 * (BM / WM) * (BN / WN)), smem, s, a);
    VX_LAUNCH_CHECK();
    return 1;
}

template <int EPI, bool CONV>
int dispatch_tile(const vx_gemm_args& a, hipStream_t s) {
    if (a.N % 128 == 0) {
        // stages: 0 = default = 1. Measured (tools/bench_kernels.py, M=43840): the single-stage form at
        // 3 blocks/CU beats a 2-deep ring at 2 blocks/CU (587 vs 490 TF for K=384, equal for K=1536)
        // and a 3-deep ring at 1 block/CU (320 TF): occupancy hides the load latency better here.
        if constexpr (!CONV) {
            if (a.stages == 2) return launch<128, 128, 64, 64, 2, EPI, CONV>(a, s);
            if (a.stages == 4) return launch<256, 128, 128, 64, 1, EPI, CONV>(a, s); // 256x128 block tile, 4 waves
            if (a.stages == 8) return launch<256, 128, 64, 64, 1, EPI, CONV>(a, s);  // 256x128 block tile, 8 waves
            if (a.stages == 16) return launch<128, 64, 64, 32, 1, EPI, CONV>(a, s);  // 128x64 block tile
            if (a.stages == 32) return launch<128, 128, 64, 32, 1, EPI, CONV>(a, s); // 128x128, 8 waves of 64x32
            if (a.stages == 64) return launch<192, 128, 96, 64, 1, EPI, CONV>(a, s); // 192x128 block tile
            if (a.stages == 0) {
                // Wave quantisation: a block lives ~as long as its tile is tall, and the grid runs in
                // ceil(tiles / resident slots) rounds (256 CUs x 4 blocks of 128x128, x 3 blocks of 192x128).
                // Pick the tile height with the cheaper rounds x height product; e.g. M = 43840, N = 384 is
                // 1029 tiles of 128 rows (2 rounds, the second one 5 blocks) but 687 tiles of 192 rows (1 round).
                const long n_t = a.N / 128;
                const long t128 = ((a.M + 127) / 128) * n_t, t192 = ((a.M + 191) / 192) * n_t;
                const long c128 = ((t128 + 1023) / 1024) * 128, c192 = ((t192 + 767) / 768) * 192;
                if (c192 < c128) return launch<192, 128, 96, 64, 1, EPI, CONV>(a, s);
            }
            return launch<128, 128, 64, 64, 1, EPI, CONV>(a, s);
        } else {
            // (few tiles and a deep reduction -- the DPT neck's 3x3 stride-2 conv, 273 tiles, K = 3456 -- was tried on a 3-stage ring:
            // 138 us against 103 single-stage; removed in round 4)
            return launch<128, 128, 64, 64, 1, EPI, CONV>(a, s);
        }
    }
    if (a.N % 64 == 0) return launch<128, 64, 64, 32, 1, EPI, CONV>(a, s);
    return launch<128, 32, 32, 32, 1, EPI, CONV>(a, s);
}

template <bool CONV>
int dispatch_epi(const vx_gemm_args& a, hipStream_t s) {
    switch (a.epi) {
        case VX_EPI_F16: return dispatch_tile<VX_EPI_F16, CONV>(a, s);
        case VX_EPI_F16_GELU: return dispatch_tile<VX_EPI_F16_GELU, CONV>(a, s);
        case VX_EPI_F16_RELU: return dispatch_tile<VX_EPI_F16_RELU, CONV>(a, s);
        case VX_EPI_F16_ADD: return dispatch_tile<VX_EPI_F16_ADD, CONV>(a, s);
        case VX_EPI_HEAD_OUT: return launch<128, 32, 32, 32, 1, VX_EPI_HEAD_OUT, CONV>(a, s); // N == 32 only
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

// second launch of a split-K GEMM: out[m, n] = act(sum_s partial[s][m][n] + bias[n]) (+ res1), 8 columns per thread, fixed order
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int S, long M, int N, const float* __restrict__ bias, int epi,
                                                            f16* __restrict__ out, long ldo, int n_valid, const f16* __restrict__ res1) {
    const int c8 = n_valid / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * c8) return;
    const long m = i / c8;
    const int n = (int)(i - m * c8) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bias ? bias[n + j] : 0.0f;
    for (int s2 = 0; s2 < S; ++s2) {
        const float* p = part + ((long)s2 * M + m) * N + n;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    f16x8 o;
    if (epi == VX_EPI_F16_ADD && res1) {
        const f16x8 rr = *reinterpret_cast<const f16x8*>(res1 + m * ldo + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rr[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(epi == VX_EPI_F16_RELU ? fmaxf(v[j], 0.0f) : v[j]);
    *reinterpret_cast<f16x8*>(out + m * ldo + n) = o;
}

} // namespace

// how many k ranges are worth it for an M x N x K product: only when the tile grid leaves most CUs idle and the reduction is deep
extern "C" int vx_gemm_pick_k_splits(int M, int N, int K) {
    const int bn = N % 128 == 0 ? 128 : (N % 64 == 0 ? 64 : 32);
    const long tiles = (long)((M + 127) / 128) * (N / bn);
    const int nk = K / BK;
    if (tiles >= 128 || nk < 16) return 1;
    int s = (int)std::min<long>(std::min(8, nk / 8), (256 + tiles - 1) / tiles);
    while (s > 1 && ((nk + s - 1) / s) * (s - 1) >= nk) --s; // no empty range
    return s > 1 ? s : 1;
}

extern "C" int vx_gemm_f16(const vx_gemm_args* args, void* stream) {
    if (args->k_splits > 1) { // split-K: partial sums + deterministic finish
        vx_gemm_args a = *args;
        VX_REQUIRE(a.k_partial, "vx_gemm_f16: split-K needs the k_partial scratch (k_splits * M * N floats)");
        VX_REQUIRE(a.epi == VX_EPI_F16 || a.epi == VX_EPI_F16_RELU || (a.epi == VX_EPI_F16_ADD && !a.res2 && !a.post_gelu && a.win_ws == 0),
                   "vx_gemm_f16: split-K supports the F16, F16_RELU and plain F16_ADD epilogues");
        VX_REQUIRE(a.K % BK == 0 && a.k_splits <= a.K / BK && ((a.K / BK + a.k_splits - 1) / a.k_splits) * (a.k_splits - 1) < a.K / BK,
                   "vx_gemm_f16: k_splits = %d leaves an empty k range for K = %d", a.k_splits, a.K);
        const int nv = a.n_valid > 0 ? a.n_valid : a.N;
        VX_REQUIRE(nv % 8 == 0 && a.ldo % 8 == 0, "vx_gemm_f16: split-K output columns and row stride must be multiples of 8");
        const int epi = a.epi;
        const void* res1 = a.res1;
        a.epi = VX_EPI_F16; // the GEMM launch only stores raw partial sums
        a.res1 = nullptr;
        a.k_splits = 0;
        vx_gemm_args g = a;
        g.k_splits = args->k_splits;
        if (!vx_gemm_f16_raw(&g, stream)) return 0;
        const long items = (long)a.M * (nv / 8);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), a.k_partial, args->k_splits, (long)a.M, a.N, a.bias, epi,
                           reinterpret_cast<f16*>(a.out), (long)a.ldo, nv, reinterpret_cast<const f16*>(res1));
        VX_LAUNCH_CHECK();
        return 1;
    }
    return vx_gemm_f16_raw(args, stream);
}

extern "C" int vx_gemm_f16_raw(const vx_gemm_args* args, void* stream) {
    const vx_gemm_args& a = *args;
    VX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "vx_gemm_f16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    VX_REQUIRE(a.K % BK == 0, "vx_gemm_f16: K=%d must be a multiple of %d (pad the weights)", a.K, BK);
    VX_REQUIRE(a.N % 32 == 0, "vx_gemm_f16: N=%d must be a multiple of 32", a.N);
    VX_REQUIRE(a.A && a.W, "vx_gemm_f16: null operand");
    const int nv = a.n_valid > 0 ? a.n_valid : a.N;
    if (epi_is_f16_tile(a.epi)) {
        VX_REQUIRE(nv % 8 == 0, "vx_gemm_f16: n_valid=%d must be a multiple of 8 for f16 outputs", nv);
        VX_REQUIRE(a.epi == VX_EPI_QKV || a.epi == VX_EPI_HEAD_OUT || a.ldo % 8 == 0, "vx_gemm_f16: ldo=%ld must be a multiple of 8", (long)a.ldo);
    }
    if (a.epi == VX_EPI_PIXSHUF) VX_REQUIRE(a.ps_Cout % 8 == 0, "vx_gemm_f16: pixel-shuffle Cout=%d must be a multiple of 8", a.ps_Cout);
    if (a.epi == VX_EPI_QKV) VX_REQUIRE(a.N == 3 * a.qkv_H * 64, "vx_gemm_f16: QKV epilogue needs N == 3*H*64");
    if (a.post_gelu) VX_REQUIRE(a.epi == VX_EPI_F16_ADD && a.res1 && !a.res2, "vx_gemm_f16: post_gelu needs the F16_ADD epilogue with exactly one residual");
    if (a.win_ws > 0) {
        VX_REQUIRE(a.epi == VX_EPI_F16_ADD && a.conv_kh == 0 && a.win_res > 0, "vx_gemm_f16: window-order output needs the plain F16_ADD form");
        const long nw = (a.win_res + a.win_ws - 1) / a.win_ws, nwh = ((a.win_res_h > 0 ? a.win_res_h : a.win_res) + a.win_ws - 1) / a.win_ws;
        VX_REQUIRE(a.M % (nw * nwh * a.win_ws * a.win_ws) == 0, "vx_gemm_f16: M=%d is not a whole number of %ld x %ld window maps", a.M, nw, nwh);
        VX_REQUIRE(a.win_shift >= 0 && a.win_shift < a.win_ws && a.win_res_h >= 0, "vx_gemm_f16: bad window shift / height");
    }
    if (a.epi == VX_EPI_HEAD_OUT) VX_REQUIRE(a.N == 32 && a.lambda, "vx_gemm_f16: head epilogue needs N == 32 and conv3 weights");
    if (a.conv_kh > 0) {
        VX_REQUIRE(a.conv_Cin % 8 == 0, "vx_gemm_f16: conv Cin=%d must be a multiple of 8", a.conv_Cin);
        VX_REQUIRE(a.K >= a.conv_kh * a.conv_kw * a.conv_Cin, "vx_gemm_f16: conv K too small");
        return dispatch_epi<true>(a, as_stream(stream));
    }
    VX_REQUIRE(a.lda % 8 == 0, "vx_gemm_f16: lda=%ld must be a multiple of 8", (long)a.lda);
    return dispatch_epi<false>(a, as_stream(stream));
}
