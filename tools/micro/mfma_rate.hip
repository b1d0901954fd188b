// micro-benchmark: sustained issue rate of the MFMA forms the scans use on gfx950 (cycles per instruction per SIMD and the
// chip-wide TFLOP/s a pure MFMA loop reaches: the practical ceiling a kernel's `roofline.frac` should be read against)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int KIND, int NACC>
__global__ void k(float *out, int iters, float a, float b) {
    float s = 0;
    if constexpr (KIND <= 1) {
        f4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (KIND == 2) {          // v_mfma_f32_16x16x32_f16
        f4 acc[NACC];
        h8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a; hb[i] = (_Float16)b; }
        for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {                                   // v_mfma_f32_32x32x16_f16
        f16v acc[NACC];
        h8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a; hb[i] = (_Float16)b; }
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NACC>
void run(const char *name, int waves_per_simd, double flops_per_instr) {
    float *out;
    hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    const int threads = 64 * 4 * waves_per_simd;   // one block per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, 100, 1.0f, 2.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * NACC * waves_per_simd;
    printf("%s nacc=%d waves/simd=%d: %.3f ms, %.1f ns per instr per SIMD (%.1f cycles @2.4GHz), %.1f TFLOP/s on 256 CUs\n", name, NACC, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, instr_per_simd * 1024 * flops_per_instr / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0, 16>("v_mfma_f32_4x4x1_16b_f32", 1, 512);
    run<0, 16>("v_mfma_f32_4x4x1_16b_f32", 2, 512);
    run<1, 16>("v_mfma_f32_16x16x4_f32", 1, 2048);
    run<1, 16>("v_mfma_f32_16x16x4_f32", 2, 2048);
    run<1, 4>("v_mfma_f32_16x16x4_f32", 2, 2048);
    run<2, 16>("v_mfma_f32_16x16x32_f16", 1, 16384);
    run<2, 16>("v_mfma_f32_16x16x32_f16", 2, 16384);
    run<3, 8>("v_mfma_f32_32x32x16_f16", 1, 32768);
    run<3, 8>("v_mfma_f32_32x32x16_f16", 2, 32768);
    return 0;
}
