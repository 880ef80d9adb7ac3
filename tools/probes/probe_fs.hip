// Probe (not part of the library): the steady-state step loop of the token-stationary block kernel in two shapes, to price a
// feature-split redesign before writing it. 8 waves, a 4 x 24 KiB LDS ring fed by LDS-DMA, one workgroup barrier per step, 12 groups per step:
//   MODE 0  today's shape: per group 4 x v_mfma_f32_16x16x32_f16, each with its own 1 KiB weight fragment (48 fragment reads per wave and step)
//   MODE 1  feature-split: per group 2 x v_mfma_f32_32x32x16_f16, each with its own fragment (24 reads per wave and step; a wave reads only
//           the half of the pair that belongs to its feature half)
//   MODE 2  feature-split on 16x16x32: per group 4 MFMAs from 2 fragments (each fragment feeds two token tiles)
// GELU = 1 adds the packed-f16 GELU of the MLP loop as side work (8 elements per lane and step, as in the real kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/probe_fs.hip -o /tmp/probe_fs && /tmp/probe_fs
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int SLAB = 24 * 1024, STEPS = 72;

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE, int GELU>
__global__ __launch_bounds__(512) void probe(const unsigned char* w, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2; // feature half of a feature-split wave
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring;
    auto feed = [&](int pair, int stage) {
        const unsigned char* src = w + (size_t)(pair % 48) * 2 * SLAB + (wave * 6) * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + stage * SLAB + wave * 6 * 1024);
#pragma unroll
        for (int z = 0; z < 6; ++z) lds_dma16(src + z * 1024, dst + z * 1024);
    };
    f32x4 acc4[24] = {};
    f32x16 acc16[6] = {};
    f16x8 b[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (f16)(0.01f * (float)((lane + i + j) % 7));
    f32x4 hn[2] = {};
    f16x8 hb = b[0];
    feed(0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int st = 0;
    const h2 c1 = {(f16)-2.3f, (f16)-2.3f}, c3 = {(f16)-0.1f, (f16)-0.1f}, one = {(f16)1.f, (f16)1.f};
#pragma unroll 1
    for (int s = 0; s < STEPS; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned char* cx = ring + st * SLAB + lane * 16;
        const unsigned char* cy = cx + SLAB;
        if (s + 1 < STEPS) feed(s + 1, st ^ 2);
        st ^= 2;
        h2 gp[4], ga[4], ge[4];
        if constexpr (MODE == 0) {
            f16x8 wx[4], wy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { wx[i] = *reinterpret_cast<const f16x8*>(cx + i * 1024); wy[i] = *reinterpret_cast<const f16x8*>(cy + i * 1024); }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int f0 = 2 * g, f1 = f0 + 1;
                acc4[f1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy[f1 % 4], hb, acc4[f1], 0, 0, 0);
                hn[f1 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[f1 % 4], b[f1 >> 1], hn[f1 & 1], 0, 0, 0);
                acc4[f0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy[f0 % 4], hb, acc4[f0], 0, 0, 0);
                hn[f0 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[f0 % 4], b[f0 >> 1], hn[f0 & 1], 0, 0, 0);
                if (f0 + 4 < 24) {
                    wx[f0 % 4] = *reinterpret_cast<const f16x8*>(cx + (f0 + 4) * 1024); wy[f0 % 4] = *reinterpret_cast<const f16x8*>(cy + (f0 + 4) * 1024);
                    wx[f1 % 4] = *reinterpret_cast<const f16x8*>(cx + (f1 + 4) * 1024); wy[f1 % 4] = *reinterpret_cast<const f16x8*>(cy + (f1 + 4) * 1024);
                }
                if constexpr (GELU) {
                    if (g < 4) { // one pair per group in the first four groups (11 packed / transcendental ops each)
                        const h2 v = {(f16)hn[g >> 1][2 * (g & 1)], (f16)hn[g >> 1][2 * (g & 1) + 1]};
                        gp[g] = v; ga[g] = gp[g] * gp[g]; ga[g] = ga[g] * c3 + c1; ga[g] = ga[g] * gp[g];
                        ge[g][0] = __builtin_exp2f16(ga[g][0]); ge[g][1] = __builtin_exp2f16(ga[g][1]); ge[g] = ge[g] + one;
                        ge[g][0] = __builtin_amdgcn_rcph(ge[g][0]); ge[g][1] = __builtin_amdgcn_rcph(ge[g][1]);
                        const h2 y = gp[g] * ge[g]; hb[2 * g] = y[0]; hb[2 * g + 1] = y[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (MODE == 1) {
            // this wave's fragments: 12 of slab X and 12 of slab Y (its half), one 32x32x16 MFMA each
            const unsigned char* hx = cx + half * 12 * 1024;
            const unsigned char* hy = cy + half * 12 * 1024;
            f16x8 wx[4], wy[4];
            f32x16 p = {};
#pragma unroll
            for (int i = 0; i < 4; ++i) { wx[i] = *reinterpret_cast<const f16x8*>(hx + i * 1024); wy[i] = *reinterpret_cast<const f16x8*>(hy + i * 1024); }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                acc16[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wy[g % 4], hb, acc16[g >> 1], 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx[g % 4], b[g], p, 0, 0, 0);
                if (g + 4 < 12) { wx[g % 4] = *reinterpret_cast<const f16x8*>(hx + (g + 4) * 1024); wy[g % 4] = *reinterpret_cast<const f16x8*>(hy + (g + 4) * 1024); }
                if constexpr (GELU) {
                    if (g < 4) {
                        const h2 v = {(f16)hn[0][g & 3], (f16)hn[1][g & 3]};
                        gp[g] = v; ga[g] = gp[g] * gp[g]; ga[g] = ga[g] * c3 + c1; ga[g] = ga[g] * gp[g];
                        ge[g][0] = __builtin_exp2f16(ga[g][0]); ge[g][1] = __builtin_exp2f16(ga[g][1]); ge[g] = ge[g] + one;
                        ge[g][0] = __builtin_amdgcn_rcph(ge[g][0]); ge[g][1] = __builtin_amdgcn_rcph(ge[g][1]);
                        const h2 y = gp[g] * ge[g]; hb[2 * g] = y[0]; hb[2 * g + 1] = y[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            hn[0][0] = p[0]; hn[0][1] = p[5]; hn[0][2] = p[10]; hn[0][3] = p[15]; hn[1][0] = p[1]; hn[1][1] = p[6]; hn[1][2] = p[11]; hn[1][3] = p[12];
        } else {
            const unsigned char* hx = cx + half * 12 * 1024;
            const unsigned char* hy = cy + half * 12 * 1024;
            f16x8 wx[4], wy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { wx[i] = *reinterpret_cast<const f16x8*>(hx + i * 1024); wy[i] = *reinterpret_cast<const f16x8*>(hy + i * 1024); }
#pragma unroll
            for (int g = 0; g < 12; ++g) { // fragment g of each stream feeds two token tiles
                acc4[2 * g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy[g % 4], hb, acc4[2 * g], 0, 0, 0);
                hn[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[g % 4], b[g], hn[0], 0, 0, 0);
                acc4[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy[g % 4], b[(g + 5) % 12], acc4[2 * g + 1], 0, 0, 0);
                hn[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[g % 4], b[(g + 7) % 12], hn[1], 0, 0, 0);
                if (g + 4 < 12) { wx[g % 4] = *reinterpret_cast<const f16x8*>(hx + (g + 4) * 1024); wy[g % 4] = *reinterpret_cast<const f16x8*>(hy + (g + 4) * 1024); }
                if constexpr (GELU) {
                    if (g < 4) {
                        const h2 v = {(f16)hn[g >> 1][2 * (g & 1)], (f16)hn[g >> 1][2 * (g & 1) + 1]};
                        gp[g] = v; ga[g] = gp[g] * gp[g]; ga[g] = ga[g] * c3 + c1; ga[g] = ga[g] * gp[g];
                        ge[g][0] = __builtin_exp2f16(ga[g][0]); ge[g][1] = __builtin_exp2f16(ga[g][1]); ge[g] = ge[g] + one;
                        ge[g][0] = __builtin_amdgcn_rcph(ge[g][0]); ge[g][1] = __builtin_amdgcn_rcph(ge[g][1]);
                        const h2 y = gp[g] * ge[g]; hb[2 * g] = y[0]; hb[2 * g + 1] = y[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = hn[0][0] + hn[1][1];
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc4[i][0] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += acc16[i][0] + acc16[i][9];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int GELU>
int run(int wgs, const unsigned char* w, float* out, unsigned long long* cyc, const char* name) {
    auto k = probe<MODE, GELU>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SLAB + 26112));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 4 * SLAB + 26112, 0, w, out, cyc);
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 4 * SLAB + 26112, 0, w, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(wgs);
    CK(hipMemcpy(c.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
    unsigned long long med = c[wgs / 2];
    printf("%-34s %3d wgs: %7.1f us per launch, %6.0f cycles per step (workgroup %d)\n", name, wgs, ms * 1000 / it, (double)med / STEPS, wgs / 2);
    return 0;
}

int main() {
    unsigned char* w; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, (size_t)96 * SLAB));
    CK(hipMalloc(&out, 512 * 512 * 4));
    CK(hipMalloc(&cyc, 512 * 8));
    std::vector<unsigned short> hw((size_t)96 * SLAB / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x2000 + (i * 2654435761u >> 20) % 0x1800); // small positive f16 values
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int wgs : {247, 118}) {
            if (run<0, 1>(wgs, w, out, cyc, "16x16x32, 48 frags, gelu")) return 1;
            if (run<1, 1>(wgs, w, out, cyc, "feature-split 32x32x16, gelu")) return 1;
            if (run<2, 1>(wgs, w, out, cyc, "feature-split 16x16x32 x2, gelu")) return 1;
            if (run<0, 0>(wgs, w, out, cyc, "16x16x32, 48 frags")) return 1;
            if (run<1, 0>(wgs, w, out, cyc, "feature-split 32x32x16")) return 1;
            if (run<2, 0>(wgs, w, out, cyc, "feature-split 16x16x32 x2")) return 1;
        }
    return 0;
}
