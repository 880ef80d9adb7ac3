// Probe (not part of the library): issue cost of the instructions of the attention kernel's softmax on gfx950 -- v_exp_f32 (transcendental), v_cvt_pk_f16_f32,
// v_pk_fma_f16, v_pk_add_f16 -- and of a packed-f16 polynomial 2^x as an alternative to v_exp_f32 + convert; one wave per SIMD slot, long independent chains,
// shader clock around the loop (s_memtime runs at 100 MHz: reported as cycles per instruction via the measured launch time instead).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_exp.hip -o /tmp/probe_exp && /tmp/probe_exp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

// 2^x for a pair of f16 x <= 0 by packed arithmetic only: n = round(x) through the 1536 magic constant, f = x - n in [-0.5, 0.5], cubic for 2^f, times 2^n
// built in the exponent field (0 below the smallest normal).
__device__ __forceinline__ h2 exp2_pk(h2 x) {
    const h2 magic = {(_Float16)1536.0f, (_Float16)1536.0f};
    const h2 t = x + magic;                  // 1536 + round(x) (|x| < 512)
    const h2 n = t - magic;
    const h2 f = x - n;
    const h2 c3 = {(_Float16)0.05550411f, (_Float16)0.05550411f}, c2 = {(_Float16)0.24022651f, (_Float16)0.24022651f}, c1 = {(_Float16)0.69314718f, (_Float16)0.69314718f},
             one = {(_Float16)1.0f, (_Float16)1.0f};
    h2 p = c3 * f + c2;
    p = p * f + c1;
    p = p * f + one;
    // exponent field of 2^n: (n + 15) << 10, clamped at 0; t's bits are 0x6600 + round(x)
    s2 e = __builtin_bit_cast(s2, t) - (s2){(short)(0x6600 - 15), (short)(0x6600 - 15)};
    e = __builtin_elementwise_max(e, (s2){0, 0});
    e = e << (s2){10, 10};
    return p * __builtin_bit_cast(h2, e);
}

template <int MODE>
__global__ void probe(float* out, int iters, float seed) {
    float a[8];
    h2 b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (float)(i + 1) - 0.001f * (float)threadIdx.x; b[i] = h2{(_Float16)a[i], (_Float16)(a[i] * 0.5f)}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]) - 1.5f;                 // v_exp_f32 + v_add
            if (MODE == 1) a[i] = a[i] * 0.999f - 1.5f;                                // v_fma only (baseline for mode 0's add)
            if (MODE == 2) b[i] = exp2_pk(b[i]) - h2{(_Float16)1.5f, (_Float16)1.5f};  // packed polynomial + v_pk_add
            if (MODE == 3) b[i] = b[i] * h2{(_Float16)0.999f, (_Float16)0.999f} - h2{(_Float16)1.5f, (_Float16)1.5f}; // v_pk_fma only
            if (MODE == 4) { h2 c = h2{(_Float16)a[i], (_Float16)a[(i + 1) & 7]}; a[i] = (float)c[0] * 0.999f - (float)c[1] * 0.001f; } // cvt_pk + cvt back
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)b[i][0] + (float)b[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    const int iters = 4096, blocks = 1024, threads = 256; // 4 waves per block: one per SIMD, 4 blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"v_exp_f32 + v_add_f32", "v_fma_f32", "packed-f16 2^x polynomial (2 values) + v_pk_add", "v_pk_fma_f16", "v_cvt_pk_f16_f32 + 2 cvt back + fma"};
    float ms[5];
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 5; ++m) {
            hipEventRecord(e0);
            switch (m) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, d, iters, -0.37f); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, -0.37f); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, d, iters, -0.37f); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(threads), 0, 0, d, iters, -0.37f); break;
                default: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, d, iters, -0.37f); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[m], e0, e1);
        }
    // per SIMD: blocks * 4 waves / 1024 SIMDs = 4 waves in sequence-ish (4 blocks per CU resident, 1 wave per SIMD each): wave-instructions per SIMD
    const double per_simd = (double)blocks * (threads / 64) / 1024.0 * iters * 8;
    for (int m = 0; m < 5; ++m) std::printf("%-52s %8.3f ms  -> %6.2f ns per loop body per SIMD-resident wave set\n", names[m], ms[m], ms[m] * 1e6 / per_simd);
    // accuracy of the polynomial on the host (same arithmetic in float, rounded to f16 at each step by the device: checked separately in the kernel test)
    return 0;
}
