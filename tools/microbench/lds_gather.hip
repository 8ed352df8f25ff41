// LDS gather microbenchmark: the traversal step's entry fetch (two ds_read_b128 of a 32-byte entry at a data-dependent index) with the
// entries as an array of 32-byte structs (the kernel's layout: each read touches half of the banks) and as two arrays of 16-byte
// halves (each read spreads over all banks).  Dependent chain: the next index comes out of the loaded data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <bool SOA>
__global__ __launch_bounds__(1024) void k(const uint4* src, int n_entries, int iters, unsigned* out) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 2 * n_entries; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) % (unsigned)n_entries, acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint4 a = SOA ? lds[idx] : lds[2 * idx];
        const uint4 b = SOA ? lds[n_entries + idx] : lds[2 * idx + 1];
        acc += a.y ^ b.z;
        idx = (a.x + b.w + acc) % (unsigned)n_entries;  // data-dependent next entry
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}
int main() {
    const int n = 1942, iters = 20000, blocks = 256, threads = 1024;  // `random`'s f64 blob: 1942 slots
    std::vector<uint4> h(2 * n);
    srand(1);
    for (auto& v : h) v = make_uint4((unsigned)rand(), (unsigned)rand(), (unsigned)rand(), (unsigned)rand());
    uint4* d; unsigned* o;
    hipMalloc(&d, h.size() * sizeof(uint4)); hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipMalloc(&o, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++)
        for (int soa = 0; soa < 2; soa++) {
            hipEventRecord(e0);
            if (soa) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(threads), 2 * n * sizeof(uint4), 0, d, n, iters, o);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(threads), 2 * n * sizeof(uint4), 0, d, n, iters, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms for %d dependent gathers per lane, 4 waves/SIMD -> %.1f ns per gather step per wave (%.0f cycles at 2.4 GHz)\n", soa ? "two arrays of 16-byte halves" : "array of 32-byte entries ", ms, iters, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
        }
    return 0;
}
