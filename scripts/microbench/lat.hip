// Latency / issue-interval probes for one wavefront on gfx950 (no other wavefront on its SIMD), the
// numbers the single-wavefront kernels are designed around.  Cycles = s_memtime ticks.
// hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double tile_t __attribute__((ext_vector_type(4)));
constexpr int kReps = 512;

__device__ __forceinline__ double bcast(double x, int j)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void probes(uint64_t *out, double *sink, double seed)
{
    __shared__ volatile double lds[256];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3;
    uint64_t t0, t1;
    int slot = 0;
#define BEGIN() t0 = __builtin_amdgcn_s_memtime()
#define END()                                                       \
    do {                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        t1 = __builtin_amdgcn_s_memtime();                          \
        if (lane == 0) out[slot] = t1 - t0;                         \
        ++slot;                                                     \
    } while (0)

    // 0: dependent v_fma_f64
    BEGIN();
    for (int i = 0; i < kReps; ++i) x = fma(x, 1.0000001, 1e-9);
    END();
    // 1: eight independent chains of v_fma_f64
    double y[8];
    for (int k = 0; k < 8; ++k) y[k] = x + k;
    BEGIN();
    for (int i = 0; i < kReps; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = fma(y[k], 1.0000001, 1e-9);
    END();
    for (int k = 0; k < 8; ++k) x += y[k];
    // 2: dependent MFMA f64 16x16x4 (same accumulator)
    tile_t acc = {x, x, x, x};
    BEGIN();
    for (int i = 0; i < kReps; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
    END();
    x += acc[0] + acc[1] + acc[2] + acc[3];
    // 3: eight independent MFMA accumulators
    tile_t a8[8];
    for (int k = 0; k < 8; ++k) a8[k] = tile_t{x, x, x, x};
    BEGIN();
    for (int i = 0; i < kReps; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, a8[k], 0, 0, 0);
    END();
    for (int k = 0; k < 8; ++k) x += a8[k][0] + a8[k][3];
    // 4: MFMA result read by a vector instruction, then fed back as an operand
    acc = tile_t{x, x, x, x};
    BEGIN();
    for (int i = 0; i < kReps; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-3, acc, 0, 0, 0);
        x = acc[0] * 1e-3;
    }
    END();
    // 5: LDS round trip: write own slot, read the neighbour's
    BEGIN();
    for (int i = 0; i < kReps; ++i) {
        lds[lane] = x;
        x = lds[(lane + 1) & 63] + 1e-9;
    }
    END();
    // 6: v_readlane broadcast (two per double) and a multiply-add on it
    BEGIN();
    for (int i = 0; i < kReps; ++i) x = fma(bcast(x, i & 63), 1e-9, x);
    END();
    // 7: DPP quad exchange (two v_mov_dpp per double) and an add
    BEGIN();
    for (int i = 0; i < kReps; ++i) x = x + dpp<0xB1>(x) * 1e-9;
    END();
    // 8: ds_bpermute exchange (__shfl_xor) and an add
    BEGIN();
    for (int i = 0; i < kReps; ++i) x = x + __shfl_xor(x, 16, 64) * 1e-9;
    END();
    // 9: v_rsq_f64 and two Newton steps, dependent
    x = fabs(x) + 1.0;
    BEGIN();
    for (int i = 0; i < kReps; ++i) {
        double r = __builtin_amdgcn_rsq(x);
        double e = fma(-(x * r), r, 1.0);
        r = fma(0.5 * r, e, r);
        e = fma(-(x * r), r, 1.0);
        r = fma(0.5 * r, e, r);
        x = r + 1.0;
    }
    END();
    // 10: v_accvgpr_write / read round trip
    BEGIN();
    for (int i = 0; i < kReps; ++i) {
        int a, lo = __double2loint(x);
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(lo));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a));
        x = __hiloint2double(__double2hiint(x), lo) + 1e-9;
    }
    END();
    sink[lane] = x;
}

int main()
{
    uint64_t *d_out;
    double *d_sink;
    hipMalloc(&d_out, 16 * sizeof(uint64_t));
    hipMalloc(&d_sink, 64 * sizeof(double));
    const char *names[] = {"v_fma_f64, dependent", "v_fma_f64, 8 independent chains (per instruction)",
                           "v_mfma_f64_16x16x4, dependent (same accumulator)", "v_mfma_f64_16x16x4, 8 independent (per instruction)",
                           "v_mfma_f64_16x16x4 -> v_mul_f64 on its result -> operand of the next", "LDS write + read of another lane's slot + add",
                           "v_readlane pair + v_fma_f64, dependent", "DPP quad exchange of a double + multiply + add, dependent",
                           "ds_bpermute exchange of a double + multiply + add, dependent", "v_rsq_f64 + two Newton steps + add, dependent",
                           "v_accvgpr_write + v_accvgpr_read + add, dependent"};
    const int per[] = {1, 8, 1, 8, 1, 1, 1, 1, 1, 1, 1};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probes, dim3(1), dim3(64), 0, 0, d_out, d_sink, 1.0 + rep);
        hipDeviceSynchronize();
    }
    uint64_t h[16];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    for (int k = 0; k < 11; ++k) printf("%-75s %7.1f ticks\n", names[k], (double)h[k] / (kReps * per[k]));
    return 0;
}
