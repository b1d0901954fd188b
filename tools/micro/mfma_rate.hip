// micro-benchmark: issue rate of the f32 MFMA forms on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND, int NACC>
__global__ void k(float *out, int iters, float a, float b) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NACC>
void run(const char *name, int waves_per_simd) {
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
    printf("%s nacc=%d waves/simd=%d: %.3f ms, %.1f ns per instr per SIMD (%.1f cycles @2.4GHz)\n", name, NACC, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main() {
    run<0, 16>("4x4x1_16b", 1);
    run<0, 16>("4x4x1_16b", 2);
    run<0, 4>("4x4x1_16b", 1);
    run<1, 16>("16x16x4", 1);
    run<1, 4>("16x16x4", 2);
    return 0;
}
