// Calibration: what one MI355X delivers for plain streaming kernels (read / write / copy / read+write mixes), to set against the
// ~3 TB/s of real traffic the partition kernels reach.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream.bin && tools/ubench_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ a, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ a, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
// every workgroup copies whole 96 KB chunks at scattered places (the apply's region fill / write-back pattern)
__global__ void __launch_bounds__(1024) k_chunks(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n_chunks, size_t chunk16) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t where = (c * 2654435761ull) % n_chunks * chunk16;
        for (size_t i = threadIdx.x; i < chunk16; i += blockDim.x) b[where + i] = a[where + i];
    }
}

int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    uint4 *a, *b; uint32_t* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    auto time = [&](const char* name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-34s %7.2f ms  %6.2f TB/s\n", name, best, moved / best / 1e9);
    };
    for (int per_cu : {4, 8, 16}) {
        const int grid = cus * per_cu;
        char nm[64];
        snprintf(nm, sizeof nm, "read   8 GB, %2d WG/CU", per_cu); time(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, sink); });
        snprintf(nm, sizeof nm, "write  8 GB, %2d WG/CU", per_cu); time(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
        snprintf(nm, sizeof nm, "copy   8+8 GB, %2d WG/CU", per_cu); time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    const size_t chunk = 96 * 1024, n_chunks = bytes / chunk;
    time("96 KB chunks copy, 1 WG(1024)/CU", 2.0 * n_chunks * chunk, [&] { hipLaunchKernelGGL(k_chunks, dim3(cus), dim3(1024), 0, 0, a, b, n_chunks, chunk / 16); });
    time("96 KB chunks copy, 2 WG(1024)/CU", 2.0 * n_chunks * chunk, [&] { hipLaunchKernelGGL(k_chunks, dim3(cus * 2), dim3(1024), 0, 0, a, b, n_chunks, chunk / 16); });
    return 0;
}
