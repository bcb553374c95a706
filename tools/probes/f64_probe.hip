// Issue-rate probe for float64 VALU on gfx950: cycles per wave-instruction for fma/add/mul, with 1..8 independent
// chains per lane and 1 or 2 waves per SIMD.  hipcc -O3 --offload-arch=gfx950 f64_probe.hip -o f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double s) {
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double b = s, c = s * 0.5;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
                if (OP == 1) a[i] = a[i] + b;
                if (OP == 2) a[i] = a[i] * b;
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"((float)b), "v"((float)c));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int ILP>
void run(const char* name, int wgs_per_cu, double* out, long long* cyc) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP, ILP><<<grid, 256>>>(out, cyc, 10, 1.0000001);
    hipEventRecord(e0);
    k<OP, ILP><<<grid, 256>>>(out, cyc, iters, 1.0000001);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double inst = (double)iters * 16 * ILP;            // wave-instructions per wave
    // waves per SIMD = wgs_per_cu (each WG = 4 waves = one per SIMD)
    printf("%-8s ILP %d  waves/SIMD %d : %.2f ms  cyclecounter/inst/wave %.2f   SIMD-ns per inst %.3f\n", name, ILP,
           wgs_per_cu, ms, (double)h / inst, ms * 1e6 / (inst * wgs_per_cu));
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 8);
    hipMalloc(&cyc, 4096 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>("fma64", w, out, cyc);
        run<0, 2>("fma64", w, out, cyc);
        run<0, 4>("fma64", w, out, cyc);
        run<0, 8>("fma64", w, out, cyc);
        run<1, 1>("add64", w, out, cyc);
        run<1, 4>("add64", w, out, cyc);
        run<2, 1>("mul64", w, out, cyc);
        run<2, 4>("mul64", w, out, cyc);
        run<3, 1>("fma32", w, out, cyc);
        run<3, 4>("fma32", w, out, cyc);
    }
    return 0;
}
