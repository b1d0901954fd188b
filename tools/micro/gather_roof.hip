// micro-benchmark: what a RANDOM gather of stored rows can draw from HBM on gfx950, as a function of the footprint the rows are spread over and of the
// row length - the ceiling the HNSW walk's hop (<= 32 rows of a few hundred bytes, ids from the graph) should be read against, not the 8 TB/s of a stream.
//
//   gather   : every 8-lane group of a wave reads one row (ROW_BYTES, 16 bytes per lane per step, the rows' steps issued back to back: the shape of
//              group_score_multi in scan_common.hpp), row index = hash(counter) % n_rows; 4 rows per group in flight (R = 4) - independent requests only,
//              no dependent chain: this is the memory system's rate for such requests, not a latency measurement
//   +atomics : beside every 16 rows, 32 atomicOr test-and-sets on a bitmap region of `bitmap_bytes` per wave slot (the walk's visited set: one bit per point,
//              1.25 MB per search at 10 M points), random words
//   chain    : the same gathers issued as a DEPENDENT chain per wave (the next row ids come from the rows just read), `waves` chains per CU: the walk's shape
//
// usage: gather_roof [max_footprint_GiB]       (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// independent gathers (+ optional atomics)
template <int STEPS, int R>
__global__ __launch_bounds__(64) void gather_kernel(const unsigned char *rows, uint64_t n_rows, uint64_t row_stride, uint32_t *bitmap, uint64_t bitmap_words,
                                                    int atomics, uint32_t iters, uint32_t *sink) {
    const int lane = threadIdx.x, sub = lane & 7, g = lane >> 3;
    uint32_t acc = 0;
    uint32_t *my_bits = bitmap + (uint64_t)blockIdx.x * bitmap_words;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[STEPS][R];
        const unsigned char *rp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t id = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)it << 8) ^ (uint64_t)(g * R + r)) % n_rows;
            rp[r] = rows + id * row_stride + sub * 16;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) v[s][r] = *reinterpret_cast<const uint4 *>(rp[r] + (uint64_t)s * 128);
        if (atomics) {       // 32 test-and-sets per 16 rows = per 32 rows here: 64 lanes, every lane one
            const uint64_t w = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)it << 8) ^ 0x80u ^ (uint64_t)lane) % bitmap_words;
            acc += atomicOr(&my_bits[w], 1u << (lane & 31));
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) acc += v[s][r].x ^ v[s][r].y ^ v[s][r].z ^ v[s][r].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// dependent chain: the ids of hop h + 1 come from the bytes read at hop h
template <int STEPS, int R>
__global__ __launch_bounds__(64) void chain_kernel(const unsigned char *rows, uint64_t n_rows, uint64_t row_stride, uint32_t iters, uint32_t *sink) {
    const int lane = threadIdx.x, sub = lane & 7, g = lane >> 3;
    uint32_t acc = blockIdx.x * 2654435761u;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[STEPS][R];
        const unsigned char *rp[R];
        const uint32_t seed = __shfl(acc, 0, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t id = mix(((uint64_t)seed << 20) ^ ((uint64_t)it << 8) ^ (uint64_t)(g * R + r)) % n_rows;
            rp[r] = rows + id * row_stride + sub * 16;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) v[s][r] = *reinterpret_cast<const uint4 *>(rp[r] + (uint64_t)s * 128);
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) acc += v[s][r].x ^ v[s][r].y ^ v[s][r].z ^ v[s][r].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// 4-byte gathers from PER-SLOT tables: the shape of the PQ walk's exact hop score (pq.hip HopPQ): a search owns a LUT of `chunks` rows of 256 f32 entries
// (96 x 1 KiB = 96 KiB at m = 96) and a surviving candidate costs one entry per chunk - `chunks` dword loads, each from another 1 KiB row: another 64-byte
// sector per 4 useful bytes.  Every wave slot owns one table; per link of the chain the wave scores `surv` candidates (chunk c of candidate j: lane
// (j * chunks + c) % 64, entry = hash); DEP: the next link's entries depend on the values just loaded (the walk: the next hop's candidates come from this
// one's result).  What it measures: the rate at which the memory system returns such requests once the live tables (slots x table bytes) exceed L2.
template <bool DEP>
__global__ __launch_bounds__(64) void lut_gather_kernel(const unsigned char *tables, uint64_t table_bytes, uint32_t chunks, uint32_t surv, uint32_t iters, uint32_t *sink) {
    const int lane = threadIdx.x;
    const unsigned char *my = tables + (uint64_t)blockIdx.x * table_bytes;
    uint32_t acc = blockIdx.x * 2654435761u + 12345u;
    const uint32_t per_link = chunks * surv, rounds = (per_link + 63) / 64;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t seed = DEP ? (uint32_t)__shfl((int)acc, 0, 64) : blockIdx.x;
        uint32_t got = 0;
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t item = r * 64 + (uint32_t)lane;
            if (item < per_link) {
                const uint32_t c = item % chunks;
                const uint32_t code = (uint32_t)mix(((uint64_t)seed << 24) ^ ((uint64_t)it << 12) ^ item) & 255u;
                got += *reinterpret_cast<const uint32_t *>(my + ((uint64_t)c * 256 + code) * 4);
            }
        }
        acc += got;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float time_ms(F &&launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const double max_gib = argc > 1 ? atof(argv[1]) : 16.0;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t region = (size_t)(max_gib * (1ull << 30));
    unsigned char *rows = nullptr;
    uint32_t *sink = nullptr, *bitmap = nullptr;
    CK(hipMalloc((void **)&rows, region));
    CK(hipMemset(rows, 1, region));
    CK(hipMalloc((void **)&sink, 64));
    const uint64_t bm_total = 6ull << 30;
    CK(hipMalloc((void **)&bitmap, bm_total));
    CK(hipMemset(bitmap, 0, bm_total));
    printf("# %s, %d CUs; rows gathered at random from a region of the given footprint; GB/s = bytes of rows read / time\n", p.name, cus);
    const uint32_t iters = 200;
    // ---- independent gathers: footprint sweep, 768-byte rows (6 steps), R = 4, 12 and 16 waves per CU ----
    for (int per_cu : {12, 16, 24}) {
        for (double gib : {0.25, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0}) {
            if (gib > max_gib) break;
            const uint64_t stride = 768, n_rows = (uint64_t)(gib * (1ull << 30)) / stride;
            const uint32_t grid = (uint32_t)cus * per_cu;
            const float ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<6, 4>), dim3(grid), dim3(64), 0, 0, rows, n_rows, stride, bitmap, (uint64_t)1, 0, iters, sink); }, 3);
            const double bytes = (double)grid * iters * 32 * 768;
            printf("gather   row 768 B  waves/CU %2d  footprint %6.2f GiB  %8.1f GB/s\n", per_cu, gib, bytes / (ms * 1e-3) / 1e9);
        }
    }
    // ---- row length at a fixed 8 GiB footprint ----
    {
        const double gib = max_gib < 8.0 ? max_gib : 8.0;
        const uint32_t grid = (uint32_t)cus * 16;
        {
            const uint64_t stride = 128, n_rows = (uint64_t)(gib * (1ull << 30)) / stride;
            const float ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<1, 4>), dim3(grid), dim3(64), 0, 0, rows, n_rows, stride, bitmap, (uint64_t)1, 0, iters, sink); }, 3);
            printf("gather   row 128 B  waves/CU 16  footprint %6.2f GiB  %8.1f GB/s\n", gib, (double)grid * iters * 32 * 128 / (ms * 1e-3) / 1e9);
        }
        {
            const uint64_t stride = 3072, n_rows = (uint64_t)(gib * (1ull << 30)) / stride;
            const float ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<24, 1>), dim3(grid), dim3(64), 0, 0, rows, n_rows, stride, bitmap, (uint64_t)1, 0, iters, sink); }, 3);
            printf("gather   row 3072 B waves/CU 16  footprint %6.2f GiB  %8.1f GB/s\n", gib, (double)grid * iters * 8 * 3072 / (ms * 1e-3) / 1e9);
        }
    }
    // ---- with the visited-set atomics: bitmap of 1.25 MB / 256 KB / 16 KB per wave slot ----
    for (uint64_t bm_bytes : {1250000ull, 262144ull, 16384ull}) {
        const double gib = max_gib < 8.0 ? max_gib : 8.0;
        const uint64_t stride = 768, n_rows = (uint64_t)(gib * (1ull << 30)) / stride;
        for (int per_cu : {12, 16}) {
            const uint32_t grid = (uint32_t)cus * per_cu;
            if ((uint64_t)grid * bm_bytes > bm_total) continue;
            const float ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<6, 4>), dim3(grid), dim3(64), 0, 0, rows, n_rows, stride, bitmap, bm_bytes / 4, 1, iters, sink); }, 3);
            printf("gather+atomics row 768 B  waves/CU %2d  footprint %6.2f GiB  bitmap %7.0f KB per slot (%5.2f GB in all)  %8.1f GB/s of rows\n", per_cu, gib,
                   bm_bytes / 1e3, (double)grid * bm_bytes / 1e9, (double)grid * iters * 32 * 768 / (ms * 1e-3) / 1e9);
        }
    }
    // ---- dependent chains (one gather of 32 rows per link of the chain) ----
    for (int per_cu : {12, 16, 24, 32}) {
        const double gib = max_gib < 8.0 ? max_gib : 8.0;
        const uint64_t stride = 768, n_rows = (uint64_t)(gib * (1ull << 30)) / stride;
        const uint32_t grid = (uint32_t)cus * per_cu;
        const float ms = time_ms([&] { hipLaunchKernelGGL((chain_kernel<6, 4>), dim3(grid), dim3(64), 0, 0, rows, n_rows, stride, iters, sink); }, 3);
        printf("chain    row 768 B  waves/CU %2d  footprint %6.2f GiB  %8.1f GB/s   %.2f us per link of the chain (32 rows)\n", per_cu, gib,
               (double)grid * iters * 32 * 768 / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
    }
    // ---- 4-byte gathers from per-slot 96 KiB tables (the PQ walk's LUT gathers): requests per second, independent and as dependent chains ----
    {
        const uint64_t table = 96 * 1024;
        const uint32_t chunks = 96;
        for (int dep = 0; dep <= 1; ++dep)
            for (int per_cu : {8, 12, 16})
                for (uint32_t surv : {2u, 4u, 8u}) {
                    const uint32_t grid = (uint32_t)cus * per_cu;
                    if ((uint64_t)grid * table > region) continue;
                    const uint32_t it4 = 400;
                    const float ms = dep ? time_ms([&] { hipLaunchKernelGGL((lut_gather_kernel<true>), dim3(grid), dim3(64), 0, 0, rows, table, chunks, surv, it4, sink); }, 3)
                                         : time_ms([&] { hipLaunchKernelGGL((lut_gather_kernel<false>), dim3(grid), dim3(64), 0, 0, rows, table, chunks, surv, it4, sink); }, 3);
                    const double reqs = (double)grid * it4 * chunks * surv;
                    printf("lut4     %s  tables %5u x 96 KiB = %6.1f MB live  waves/CU %2d  %u candidates per link  %7.2f G requests/s  = %7.1f GB/s of 64-byte sectors, %6.1f GB/s useful"
                           "   %.2f us per link\n", dep ? "chain      " : "independent", grid, (double)grid * table / 1e6, per_cu, surv, reqs / (ms * 1e-3) / 1e9,
                           reqs * 64 / (ms * 1e-3) / 1e9, reqs * 4 / (ms * 1e-3) / 1e9, ms * 1e3 / it4);
                }
    }
    return 0;
}
