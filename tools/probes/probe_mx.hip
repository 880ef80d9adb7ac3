// Probe (not part of the library): operand and scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, checked with exact small
// integers against a host product. Hypothesis: lane l supplies A[row l & 15][k = 32 (l >> 4) + 0 .. 31] and B[k likewise][col l & 15] as 32
// consecutive bytes (8 VGPRs, little endian); the e8m0 scale byte picked by opsel from the lane's scale VGPR multiplies that lane's 32-element block;
// D is the usual 16x16 map (col l & 15, rows 4 (l >> 4) + i).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mx.hip -o /tmp/probe_mx && /tmp/probe_mx
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void mx(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out) {
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = c[i];
}

static unsigned char e4m3(int v) { // small non-negative integers 0 .. 15, exact (bias 7, 3 mantissa bits)
    if (v == 0) return 0;
    int e = 0;
    while ((1 << (e + 1)) <= v) ++e;
    const int man = ((v << 3) >> e) & 7;
    return (unsigned char)(((e + 7) << 3) | man);
}

int main() {
    float A[16][128], B[128][16];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 24) % 8); };
    for (auto& r : A) for (float& v : r) v = (float)rnd();
    for (auto& r : B) for (float& v : r) v = (float)rnd();
    std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            ha[l * 32 + j] = e4m3((int)A[l & 15][32 * (l >> 4) + j]);
            hb[l * 32 + j] = e4m3((int)B[32 * (l >> 4) + j][l & 15]);
        }
    std::vector<int> sa(64, 127), sb(64, 127); // e8m0 127 = 2^0 in byte 0
    for (int l = 16; l < 32; ++l) sa[l] = 128;  // A's k-block 1 (k 32..63) x 2 for every row
    sb[5] = 129;                                 // B's k-block 0 of column 5 x 4
    void *da, *db, *dsa, *dsb, *dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 1024);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mx, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (float*)dout);
    float out[256];
    if (hipMemcpy(out, dout, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (l >> 4) + i, col = l & 15;
            double want = 0;
            for (int k = 0; k < 128; ++k) {
                double t = (double)A[row][k] * B[k][col];
                if (k >= 32 && k < 64) t *= 2;  // scale_a of lanes 16..31
                if (k < 32 && col == 5) t *= 4;  // scale_b of lane 5
                want += t;
            }
            worst = std::fmax(worst, std::fabs(want - out[l * 4 + i]));
        }
    printf("max |device - host| under the hypothesis: %g  (D[0][0] = %g)\n", worst, out[0]);
    return 0;
}
