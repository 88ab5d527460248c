// Microbenchmark behind k_spmv_dia_march*: how fast can a march over z read S slot arrays (stride n) + x and write y, as a function of
// the piece a wave reads per instruction (8 B/lane = 512 B, 16 B/lane = 1 KiB) and of the patch shape?  No LDS, no arithmetic to speak
// of: the memory system's answer for this access pattern.    hipcc -O3 --offload-arch=gfx950 tools/micro/streams.hip -o /tmp/streams && /tmp/streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// 64 x 8 patch, thread = (lane, wave): rows 2 wave, 2 wave + 1 - the shape of k_spmv_dia_march2
template <int S>
__global__ __launch_bounds__(256) void k_patch8(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int64_t n, int nx, int ny, int nz, int zchunk) {
    const int tiles_x = nx / 64, tiles_y = ny / 8, per = tiles_x * tiles_y;
    const int chunk = blockIdx.x / per, tile = blockIdx.x - chunk * per, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t P = (int64_t)nx * ny, b0 = tx * 64 + lane + (int64_t)nx * (ty * 8 + 2 * wv), b1 = b0 + nx;
    for (int z = chunk * zchunk; z < min(nz, (chunk + 1) * zchunk); ++z) {
        const int64_t r0 = b0 + P * z, r1 = b1 + P * z;
        double a0 = x[r0], a1 = x[r1];
#pragma unroll
        for (int s = 0; s < S; ++s) { a0 += u[s * n + r0]; a1 += u[s * n + r1]; }
        y[r0] = a0; y[r1] = a1;
    }
}
// 128 x 4 patch, thread = (lane, wave): row wave, columns 2 lane, 2 lane + 1: 16 B per lane, 1 KiB per wave instruction
template <int S>
__global__ __launch_bounds__(256) void k_patch16(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int64_t n, int nx, int ny, int nz, int zchunk) {
    const int tiles_x = nx / 128, tiles_y = ny / 4, per = tiles_x * tiles_y;
    const int chunk = blockIdx.x / per, tile = blockIdx.x - chunk * per, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t P = (int64_t)nx * ny, b0 = tx * 128 + 2 * lane + (int64_t)nx * (ty * 4 + wv);
    for (int z = chunk * zchunk; z < min(nz, (chunk + 1) * zchunk); ++z) {
        const int64_t r0 = b0 + P * z;
        d2 a = *reinterpret_cast<const d2 *>(x + r0);
#pragma unroll
        for (int s = 0; s < S; ++s) a += *reinterpret_cast<const d2 *>(u + s * n + r0);
        *reinterpret_cast<d2 *>(y + r0) = a;
    }
}
// 128 x 8 patch, thread: rows 2 wave, 2 wave + 1, columns 2 lane, 2 lane + 1 (four rows of work per thread)
template <int S>
__global__ __launch_bounds__(256) void k_patch16x2(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int64_t n, int nx, int ny, int nz, int zchunk) {
    const int tiles_x = nx / 128, tiles_y = ny / 8, per = tiles_x * tiles_y;
    const int chunk = blockIdx.x / per, tile = blockIdx.x - chunk * per, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t P = (int64_t)nx * ny, b0 = tx * 128 + 2 * lane + (int64_t)nx * (ty * 8 + 2 * wv), b1 = b0 + nx;
    for (int z = chunk * zchunk; z < min(nz, (chunk + 1) * zchunk); ++z) {
        const int64_t r0 = b0 + P * z, r1 = b1 + P * z;
        d2 a0 = *reinterpret_cast<const d2 *>(x + r0), a1 = *reinterpret_cast<const d2 *>(x + r1);
#pragma unroll
        for (int s = 0; s < S; ++s) { a0 += *reinterpret_cast<const d2 *>(u + s * n + r0); a1 += *reinterpret_cast<const d2 *>(u + s * n + r1); }
        *reinterpret_cast<d2 *>(y + r0) = a0; *reinterpret_cast<d2 *>(y + r1) = a1;
    }
}
// linear order, 16 B per lane (the shape of the vector kernels): the ceiling for S + 1 read streams and one write stream
template <int S>
__global__ __launch_bounds__(256) void k_linear16(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int64_t stride, int64_t n) {
    for (int64_t i = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 2 * (int64_t)gridDim.x * 256) {
        d2 a = *reinterpret_cast<const d2 *>(x + i);
#pragma unroll
        for (int s = 0; s < S; ++s) a += *reinterpret_cast<const d2 *>(u + s * stride + i);
        *reinterpret_cast<d2 *>(y + i) = a;
    }
}

// ---- the SAME bytes with the S slot values of 64 consecutive rows side by side ([group of 64 rows][S][64]: one stream of S x 512 B
// pieces instead of S streams of 512 B pieces) - would the march be faster on such a layout?
template <int S>
__global__ __launch_bounds__(256) void k_patch8_blocked(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int nx, int ny, int nz, int zchunk) {
    const int tiles_x = nx / 64, tiles_y = ny / 8, per = tiles_x * tiles_y;
    const int chunk = blockIdx.x / per, tile = blockIdx.x - chunk * per, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t P = (int64_t)nx * ny, b0 = tx * 64 + (int64_t)nx * (ty * 8 + 2 * wv), b1 = b0 + nx;      // first row of the wave's two 64-row segments
    for (int z = chunk * zchunk; z < min(nz, (chunk + 1) * zchunk); ++z) {
        const int64_t r0 = b0 + P * z, r1 = b1 + P * z;
        double a0 = x[r0 + lane], a1 = x[r1 + lane];
        const double *u0 = u + (r0 / 64) * (S * 64) + lane, *u1 = u + (r1 / 64) * (S * 64) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) { a0 += u0[s * 64]; a1 += u1[s * 64]; }
        y[r0 + lane] = a0; y[r1 + lane] = a1;
    }
}
// ... and with 16 B per lane: [group of 128 rows][S][128], 128 x 8 patch (two rows per thread)
template <int S>
__global__ __launch_bounds__(256) void k_patch16x2_blocked(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int nx, int ny, int nz, int zchunk) {
    const int tiles_x = nx / 128, tiles_y = ny / 8, per = tiles_x * tiles_y;
    const int chunk = blockIdx.x / per, tile = blockIdx.x - chunk * per, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t P = (int64_t)nx * ny, b0 = tx * 128 + (int64_t)nx * (ty * 8 + 2 * wv), b1 = b0 + nx;
    for (int z = chunk * zchunk; z < min(nz, (chunk + 1) * zchunk); ++z) {
        const int64_t r0 = b0 + P * z, r1 = b1 + P * z;
        d2 a0 = *reinterpret_cast<const d2 *>(x + r0 + 2 * lane), a1 = *reinterpret_cast<const d2 *>(x + r1 + 2 * lane);
        const double *u0 = u + (r0 / 128) * (S * 128) + 2 * lane, *u1 = u + (r1 / 128) * (S * 128) + 2 * lane;
#pragma unroll
        for (int s = 0; s < S; ++s) { a0 += *reinterpret_cast<const d2 *>(u0 + s * 128); a1 += *reinterpret_cast<const d2 *>(u1 + s * 128); }
        *reinterpret_cast<d2 *>(y + r0 + 2 * lane) = a0; *reinterpret_cast<d2 *>(y + r1 + 2 * lane) = a1;
    }
}
template <int S>
__global__ __launch_bounds__(256) void k_linear16_blocked(const double *__restrict__ u, const double *__restrict__ x, double *__restrict__ y, int64_t n) {
    for (int64_t i = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 2 * (int64_t)gridDim.x * 256) {
        d2 a = *reinterpret_cast<const d2 *>(x + i);
        const double *ub = u + (i / 128) * (S * 128) + (i & 127);
#pragma unroll
        for (int s = 0; s < S; ++s) a += *reinterpret_cast<const d2 *>(ub + s * 128);
        *reinterpret_cast<d2 *>(y + i) = a;
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const int64_t n = (int64_t)N * N * N;
    constexpr int S = 7;
    double *u, *x, *y;
    const int64_t pad = argc > 2 ? atoll(argv[2]) : 0;                  // doubles between two slot arrays beyond n
    const int64_t stride = n + pad;
    CK(hipMalloc(&u, sizeof(double) * stride * S)); CK(hipMalloc(&x, sizeof(double) * n)); CK(hipMalloc(&y, sizeof(double) * n));
    CK(hipMemset(u, 0, sizeof(double) * stride * S)); CK(hipMemset(x, 0, sizeof(double) * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 8.0 * n * (S + 2);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %6.0f GB/s  %.3f of 8 TB/s\n", name, 1e3 * ms / reps, bytes / (1e6 * ms / reps), bytes / (1e6 * ms / reps) / 8000.0);
    };
    printf("N = %d, %d slot arrays + x read, y written (%.0f B per row), slot stride n + %lld doubles\n", N, S, 8.0 * (S + 2), (long long)pad);
    for (int zc : {8, 16, 32, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "64 x 8 patch, 8 B/lane, march %d", zc);
        run(nm, [&] { k_patch8<S><<<(N / 64) * (N / 8) * ((N + zc - 1) / zc), 256>>>(u, x, y, stride, N, N, N, zc); });
        snprintf(nm, sizeof nm, "128 x 4 patch, 16 B/lane, march %d", zc);
        run(nm, [&] { k_patch16<S><<<(N / 128) * (N / 4) * ((N + zc - 1) / zc), 256>>>(u, x, y, stride, N, N, N, zc); });
        snprintf(nm, sizeof nm, "128 x 8 patch, 16 B/lane x 2 rows, march %d", zc);
        run(nm, [&] { k_patch16x2<S><<<(N / 128) * (N / 8) * ((N + zc - 1) / zc), 256>>>(u, x, y, stride, N, N, N, zc); });
    }
    for (int zc : {8, 16, 32, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "BLOCKED [64 rows][7][64]: 64 x 8 patch, march %d", zc);
        run(nm, [&] { k_patch8_blocked<S><<<(N / 64) * (N / 8) * ((N + zc - 1) / zc), 256>>>(u, x, y, N, N, N, zc); });
        snprintf(nm, sizeof nm, "BLOCKED [128][7][128]: 128 x 8 patch 16 B, march %d", zc);
        run(nm, [&] { k_patch16x2_blocked<S><<<(N / 128) * (N / 8) * ((N + zc - 1) / zc), 256>>>(u, x, y, N, N, N, zc); });
    }
    run("BLOCKED linear, 16 B/lane, 2048 workgroups", [&] { k_linear16_blocked<S><<<2048, 256>>>(u, x, y, n); });
    run("BLOCKED linear, 16 B/lane, 8192 workgroups", [&] { k_linear16_blocked<S><<<8192, 256>>>(u, x, y, n); });
    run("linear, 16 B/lane, 2048 workgroups", [&] { k_linear16<S><<<2048, 256>>>(u, x, y, stride, n); });
    run("linear, 16 B/lane, 8192 workgroups", [&] { k_linear16<S><<<8192, 256>>>(u, x, y, stride, n); });
    run("linear, 16 B/lane, 3 slot arrays (but priced as 7)", [&] { k_linear16<3><<<4096, 256>>>(u, x, y, stride, n); });
    run("linear, 16 B/lane, 1 slot array  (but priced as 7)", [&] { k_linear16<1><<<4096, 256>>>(u, x, y, stride, n); });
    hipDeviceSynchronize();
    const hipError_t last = hipGetLastError();
    if (last != hipSuccess) { printf("error: %s\n", hipGetErrorString(last)); return 1; }
    return 0;
}
