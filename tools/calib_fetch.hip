// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, for the access patterns
// of the integrator kernels (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Three read kernels over a 2 GiB buffer (8x the 256 MiB Infinity Cache, so
// every byte comes from HBM exactly once) and one write kernel:
//   calib_read8      every lane reads 8 B, unit stride across the wave           (y0 / params / tvals loads)
//   calib_read16     every lane reads 16 B, unit stride (global_load_dwordx4)    (the guide's x2 case)
//   calib_records    every lane reads its own 160-byte record with ten 16-byte loads, records contiguous by lane
//                    (how sa_k_backward reads a trajectory record of the n = 2 problem)
//   calib_write8     every lane writes 8 B, unit stride
// Build: hipcc --offload-arch=gfx950 -O2 tools/calib_fetch.hip -o gpurun_out/calib_fetch ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- gpurun_out/calib_fetch   (and again with --pmc WRITE_SIZE).
// Each kernel is launched once over the whole buffer: bytes per dispatch = 2^31 (reads) / 2^31 (write).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void calib_read8(const double *p, size_t n, double *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678) sink[0] = acc;
}

__global__ void calib_read16(const double2 *p, size_t n, double *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) sink[0] = acc;
}

__global__ void calib_records(const double2 *p, size_t nrec, double *sink)       // 160-byte records = 10 x double2
{
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        const double2 *q = p + r * 10;
#pragma unroll
        for (int k = 0; k < 10; k++) { double2 v = q[k]; acc += v.x + v.y; }
    }
    if (acc == 12345.678) sink[0] = acc;
}

__global__ void calib_write8(double *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main()
{
    const size_t bytes = (size_t)1 << 31;
    double *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    const int grid = 256 * 16, block = 256;
    calib_write8<<<grid, block>>>(buf, bytes / 8);
    CHECK(hipDeviceSynchronize());
    calib_read8<<<grid, block>>>(buf, bytes / 8, sink);
    CHECK(hipDeviceSynchronize());
    calib_read16<<<grid, block>>>((const double2 *)buf, bytes / 16, sink);
    CHECK(hipDeviceSynchronize());
    calib_records<<<grid, block>>>((const double2 *)buf, bytes / 160, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_fetch: bytes per dispatch %zu (records: %zu)\n", bytes, bytes / 160 * 160);
    return 0;
}
