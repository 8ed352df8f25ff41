// csrc/mlrt_multi_kernels.hip - the two small kernels either side of the one gather of a multi-device frame
// (mlrt_multi.cpp): finished int32 rows -> bytes on every device, gathered bytes -> the int32 frame on the root device.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {
// values are 0..255 (to_rgb, raytrace.rs:59-68): one byte per channel on the wire, 240 KB per peer for 1/8 of final_scene 800x800
__global__ void rows_to_bytes(const int32_t* rgb, uint8_t* bytes, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bytes[i] = (uint8_t)rgb[i];
}
// gathered: [n_devices][stride] bytes, device k's rows in its row-plan order (row j of the frame belongs to device j % n_devices,
// position j / n_devices): the frame, bottom row first, as Vec<Vec<RGB>> flattened
__global__ void assemble_frame(const uint8_t* gathered, size_t stride, int n_devices, int width, int height, int32_t* frame) {
    const int row_elems = width * 3;
    const int n = height * row_elems;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i / row_elems, e = i - j * row_elems;
        const int k = j % n_devices, pos = j / n_devices;
        frame[i] = (int32_t)gathered[(size_t)k * stride + (size_t)pos * row_elems + e];
    }
}
}  // namespace

extern "C" int mlrt_multi_rows_to_bytes(const int32_t* d_rgb, uint8_t* d_bytes, int n, hipStream_t stream) {
    if (n <= 0) return 0;
    const int grid = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
    hipLaunchKernelGGL(rows_to_bytes, dim3((unsigned)grid), dim3(256), 0, stream, d_rgb, d_bytes, n);
    return (int)hipGetLastError();
}
extern "C" int mlrt_multi_assemble(const uint8_t* d_gathered, size_t stride, int n_devices, int width, int height, int32_t* d_frame, hipStream_t stream) {
    const int n = width * height * 3;
    const int grid = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
    hipLaunchKernelGGL(assemble_frame, dim3((unsigned)grid), dim3(256), 0, stream, d_gathered, stride, n_devices, width, height, d_frame);
    return (int)hipGetLastError();
}
