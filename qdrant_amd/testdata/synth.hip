// synth.hip — libqmx_testdata.so: the synthetic row generators of bench.py, tools/ and tests/.  NOT part of the product library
// (libqdrant_amd.so exports scoring only); built next to it by `make` and loaded by qdrant_amd/_ffi.py for the harnesses.
//
// Counter-based generator (integer Irwin-Hall of four 16-bit uniforms, no libm), reproducible on any host: element (row, col) depends only on
// (seed, row, col); the CPU oracle's twins (qo_synth_fill_f32 / qo_synth_fill_latent_f32) are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#define QMX_TD_API extern "C" __attribute__((visibility("default")))

namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ inline float irwin_hall(uint64_t h) {
    const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)(h >> 48);
    return (float)(s - 131070) * (1.0f / 37837.0f);
}

__global__ void synth_fill_kernel(uint64_t seed_mixed, uint64_t row0, uint64_t n, uint32_t dim, float *out) {
    const uint64_t total = n * dim;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = irwin_hall(splitmix64(seed_mixed ^ (row0 * dim + i)));
}

// Rows of low intrinsic dimension (what embedding models produce, and what an ANN index is for): x[r][c] = sum_k z[r][k] * W[k][c]
// (+ noise * e[r][c]) with z, W, e from the generator above (z: seed, W: seed ^ 0x57, e: seed ^ 0xE5) and the sum as ONE fmaf chain
// in k order, so the CPU oracle reproduces every element bit for bit.  Block = 8 rows; their latent coordinates sit in LDS.
__global__ __launch_bounds__(256) void synth_latent_kernel(uint64_t seed_z, uint64_t seed_e, const float *W, uint64_t row0, uint64_t n, uint32_t dim,
                                                           uint32_t K, float noise, float *out) {
    extern __shared__ float z[];                                   // [8][K]
    for (uint64_t rb = (uint64_t)blockIdx.x * 8; rb < n; rb += (uint64_t)gridDim.x * 8) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 8 * K; i += 256) {
            const uint64_t r = rb + i / K;
            z[i] = irwin_hall(splitmix64(seed_z ^ ((row0 + r) * K + i % K)));
        }
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < dim; c += 256) {
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.0f;
            for (uint32_t k = 0; k < K; ++k) {
                const float w = W[(uint64_t)k * dim + c];
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_fmaf(z[r * K + k], w, acc[r]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint64_t row = rb + r;
                if (row >= n) break;
                float v = acc[r];
                if (noise != 0.0f) v = __builtin_fmaf(noise, irwin_hall(splitmix64(seed_e ^ ((row0 + row) * dim + c))), v);
                out[row * dim + c] = v;
            }
        }
    }
}

// status codes of include/qdrant_amd.h (qmx_status): OK = 0, OTHER = 6, BAD_ARG = 8, NO_DEVICE = 9
int32_t device_ok(int32_t device_id, const void *out_dev) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); return 9; }
    if (device_id < 0 || device_id >= count) return 9;
    if (hipSetDevice(device_id) != hipSuccess) return 9;
    hipPointerAttribute_t attr;
    if (!out_dev || hipPointerGetAttributes(&attr, out_dev) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
        (void)hipGetLastError();
        return 8;
    }
    return 0;
}

}  // namespace

QMX_TD_API int32_t qmx_synth_fill_f32(int32_t device_id, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out_dev) {
    if (dim == 0) return 8;
    const int32_t rc = device_ok(device_id, out_dev);
    if (rc) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(4096), dim3(256), 0, nullptr, splitmix64(seed), row0, n, dim, out_dev);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 6;
}

QMX_TD_API int32_t qmx_synth_fill_latent_f32(int32_t device_id, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, uint32_t latent_dim, float noise,
                                             float *out_dev) {
    if (dim == 0 || latent_dim < 1 || latent_dim > 1024) return 8;
    const int32_t rc = device_ok(device_id, out_dev);
    if (rc) return rc;
    if (n == 0) return 0;
    float *W = nullptr;
    if (hipMalloc((void **)&W, (size_t)latent_dim * dim * sizeof(float)) != hipSuccess) return 1;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(4096), dim3(256), 0, nullptr, splitmix64(seed ^ 0x57ull), (uint64_t)0, (uint64_t)latent_dim, dim, W);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 7) / 8, 1u << 16);
    hipLaunchKernelGGL(synth_latent_kernel, dim3(grid), dim3(256), (size_t)8 * latent_dim * sizeof(float), nullptr, splitmix64(seed), splitmix64(seed ^ 0xE5ull),
                       W, row0, n, dim, latent_dim, noise, out_dev);
    const bool ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipFree(W);
    return ok ? 0 : 6;
}
