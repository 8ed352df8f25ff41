// tools/microbench/valu_issue.hip - what ONE SIMD of gfx950 sustains per vector instruction class, measured.
//
// The render kernels are bound by vector-instruction issue, so bench.py's roofline prices the kernel's VALU
// work against the chip's VALU issue capacity.  This microbenchmark supplies the prices: for each instruction class, a
// saturating stream of independent instructions from W waves per SIMD gives cycles per wave-instruction per SIMD.
// Round 3 (the round-2 numbers had v_fma_f32 at half the rate of v_add_f32 and two clocks that disagreed by 2x):
//  * W waves per SIMD BY CONSTRUCTION: one workgroup of 256 x W threads per CU (W <= 4; two workgroups of 1024 for W = 8), each
//    asking for so much LDS that no other fits on its CU - 256-thread workgroups had been packed unevenly over the CUs;
//  * the constant operands are SGPRs / inline constants: with all three sources of v_fma_f32 in VGPRs the instruction ran at half
//    rate (register-bank conflicts of this benchmark, not of the instruction);
//  * time = launch wall time (HIP events over 4 back-to-back launches after 4 warm-up launches) x the clock read with
//    hipDeviceGetAttribute at run time; the in-wave s_memtime count is printed as raw ticks only (it does not tick at the shader clock).
//
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_issue.hip -o gpurun_out/valu_issue && gpurun_out/valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kUnroll = 16;     // independent instructions per loop body (different destination registers)
constexpr int kIters = 65536;  // (launches of 4-90 ms: long enough for the clocks to settle and for the events to time)

// One kernel per instruction class: OP is the asm template with %0 = in/out register of chain k, %1, %2 = constants.
#define DEFINE_KERNEL_F32(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(1024) void NAME(float* out, unsigned long long* cycles, float b, float c) { \
        float a[kUnroll];                                                                                    \
        for (int k = 0; k < kUnroll; k++) a[k] = (float)(threadIdx.x + k) * 1e-3f + 1.0f;                    \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < kIters; it++) {                                                                \
            _Pragma("unroll") for (int k = 0; k < kUnroll; k++) asm volatile(ASM : "+v"(a[k]) : "s"(b), "s"(c) : "vcc"); \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        float s = 0; for (int k = 0; k < kUnroll; k++) s += a[k];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;        \
    }
#define DEFINE_KERNEL_F64(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(1024) void NAME(double* out, unsigned long long* cycles, double b, double c) { \
        double a[kUnroll];                                                                                    \
        for (int k = 0; k < kUnroll; k++) a[k] = (double)(threadIdx.x + k) * 1e-3 + 1.0;                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                           \
        for (int it = 0; it < kIters; it++) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < kUnroll; k++) asm volatile(ASM : "+v"(a[k]) : "s"(b), "s"(c) : "vcc"); \
        }                                                                                                     \
        const unsigned long long t1 = __builtin_readcyclecounter();                                           \
        double s = 0; for (int k = 0; k < kUnroll; k++) s += a[k];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;         \
    }

DEFINE_KERNEL_F32(k_fma_f32, "v_fma_f32 %0, %0, %1, 1.0")
DEFINE_KERNEL_F32(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEFINE_KERNEL_F32(k_add_f32, "v_add_f32 %0, %0, %1")
DEFINE_KERNEL_F32(k_max_f32, "v_max_f32 %0, %0, %1")
DEFINE_KERNEL_F32(k_rcp_f32, "v_rcp_f32 %0, %0")
DEFINE_KERNEL_F32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
DEFINE_KERNEL_F32(k_sin_f32, "v_sin_f32 %0, %0")
DEFINE_KERNEL_F32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL_F32(k_xor_b32, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL_F32(k_cndmask_b32, "v_cndmask_b32 %0, %0, 1.0, vcc")
DEFINE_KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %1, 1.0")
DEFINE_KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")
DEFINE_KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEFINE_KERNEL_F64(k_max_f64, "v_max_f64 %0, %0, %1")
DEFINE_KERNEL_F64(k_min_f64, "v_min_f64 %0, %0, %1")
DEFINE_KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")
DEFINE_KERNEL_F64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
DEFINE_KERNEL_F64(k_rsq_f64, "v_rsq_f64 %0, %0")
// v_mad_u64_u32 (Philox's 32x32->64 multiply): 64-bit accumulator, 32-bit factors
__global__ __launch_bounds__(1024) void k_mad_u64_u32(double* out, unsigned long long* cycles, double b, double c) {
    unsigned long long a[kUnroll];
    const unsigned int f0 = (unsigned int)(b * 1e6) | 1u, f1 = (unsigned int)(c * 1e12) | 3u;
    for (int k = 0; k < kUnroll; k++) a[k] = threadIdx.x + k;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k]) : "s"(f0), "v"(f1) : "vcc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long s = 0; for (int k = 0; k < kUnroll; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// The same v_fma_f32 stream with part of the wave switched off (EXEC set by the branch around the loop): which lane groups cost issue cycles?
// MASK: lanes for which (lane & AND) == EQ run the loop.
#define DEFINE_KERNEL_MASKED(NAME, AND, EQ)                                                                  \
    __global__ __launch_bounds__(1024) void NAME(float* out, unsigned long long* cycles, float b, float c) { \
        float a[kUnroll];                                                                                    \
        for (int k = 0; k < kUnroll; k++) a[k] = (float)(threadIdx.x + k) * 1e-3f + 1.0f;                    \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        if (((threadIdx.x & 63) & (AND)) == (EQ)) {                                                          \
            for (int it = 0; it < kIters; it++) {                                                            \
                _Pragma("unroll") for (int k = 0; k < kUnroll; k++) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[k]) : "s"(b), "s"(c) : "vcc"); \
            }                                                                                                \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        float s = 0; for (int k = 0; k < kUnroll; k++) s += a[k];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;        \
    }
DEFINE_KERNEL_MASKED(k_fma_low32, 32, 0)    // lanes 0-31
DEFINE_KERNEL_MASKED(k_fma_high32, 32, 32)  // lanes 32-63
DEFINE_KERNEL_MASKED(k_fma_low16, 48, 0)    // lanes 0-15
DEFINE_KERNEL_MASKED(k_fma_even, 1, 0)      // every other lane
DEFINE_KERNEL_MASKED(k_fma_q0q2, 16, 0)     // lanes 0-15 and 32-47
DEFINE_KERNEL_MASKED(k_fma_lane0, 63, 0)    // one lane
DEFINE_KERNEL_F64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
DEFINE_KERNEL_F64(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, 1.0")
DEFINE_KERNEL_F64(k_mov_b64, "v_mov_b64 %0, %1")

// The chip idles at a low clock and needs milliseconds of sustained load to reach its full clock, so every measurement is
// kReps back-to-back launches of which only the last kTimed are timed.
constexpr int kReps = 8, kTimed = 4;
static double g_clock_ghz = 2.4;
template <typename T, typename K>
void run(const char* name, K kernel, int waves_per_simd, T* d_out, unsigned long long* d_cycles, int n_cus) {
    // W <= 4: one workgroup of 256 x W threads per CU, 100 KB of LDS each (a second one does not fit the CU's 160 KB);
    // W = 8: two workgroups of 1024 threads per CU, 64 KB each (a third does not fit)
    const int per_cu = waves_per_simd <= 4 ? 1 : 2;
    const int threads = 256 * waves_per_simd / per_cu;
    const size_t lds = per_cu == 1 ? 100 * 1024 : 64 * 1024;
    const int grid = n_cus * per_cu;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned long long> cyc((size_t)grid * (threads / 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < kReps; r++) {
        if (r == kReps - kTimed) CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, d_out, d_cycles, (T)1.0000001, (T)1e-9);
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)kTimed;
    CHECK(hipMemcpy(cyc.data(), d_cycles, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (unsigned long long c : cyc) mean += (double)c;
    mean /= (double)cyc.size();
    const double insts = (double)kUnroll * kIters;
    const double wall_cycles = (double)ms * 1e-3 * g_clock_ghz * 1e9;
    std::printf("%-16s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (launch wall time x %.2f GHz); %.1f Ginstr/s chip-wide; launch %.3f ms; s_memtime ticks per instruction of a wave %.3f\n",
                name, waves_per_simd, wall_cycles / (insts * waves_per_simd), g_clock_ghz, insts * waves_per_simd * 4.0 * n_cus / (ms * 1e-3) / 1e9, ms, mean / insts);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0) == hipSuccess && khz > 0) g_clock_ghz = khz / 1e6;
    std::printf("# %s, %d CUs, clock %.0f MHz; %d x %d independent instructions per wave; waves per SIMD by construction (see the header)\n", prop.name, n_cus, g_clock_ghz * 1e3, kIters, kUnroll);
    double* d_out;
    unsigned long long* d_cycles;
    CHECK(hipMalloc(&d_out, (size_t)n_cus * 8 * 256 * sizeof(double) + 4096));
    CHECK(hipMalloc(&d_cycles, (size_t)n_cus * 8 * 4 * sizeof(unsigned long long)));
    // `valu_issue <w>`: one occupancy only (for a rocprofv3 --pmc GRBM_GUI_ACTIVE run: cycles per launch instead of an assumed clock)
    const int only = argc > 1 ? std::atoi(argv[1]) : 0;
    for (int w : {1, 2, 4, 8}) {
        if (only && w != only) continue;
#define RUN32(k) run<float>(#k, k, w, reinterpret_cast<float*>(d_out), d_cycles, n_cus)
#define RUN64(k) run<double>(#k, k, w, d_out, d_cycles, n_cus)
        RUN32(k_fma_f32); RUN32(k_mul_f32); RUN32(k_add_f32); RUN32(k_max_f32); RUN32(k_rcp_f32); RUN32(k_sqrt_f32); RUN32(k_sin_f32); RUN32(k_mul_lo_u32); RUN32(k_xor_b32);
        RUN64(k_fma_f64); RUN64(k_add_f64); RUN64(k_mul_f64); RUN64(k_max_f64); RUN64(k_min_f64); RUN64(k_rcp_f64); RUN64(k_sqrt_f64); RUN64(k_rsq_f64);
        RUN64(k_mad_u64_u32); RUN64(k_cmp_f64); RUN64(k_div_fixup_f64); RUN64(k_mov_b64);
        RUN32(k_fma_low32); RUN32(k_fma_high32); RUN32(k_fma_low16); RUN32(k_fma_even); RUN32(k_fma_q0q2); RUN32(k_fma_lane0);
    }
    return 0;
}
