// Device-to-host rate of the frame readback on this box: one hipMemcpyAsync stream, two concurrent streams, and a
// copy kernel storing straight into mapped pinned memory.  (hipcc --offload-arch=gfx950 -O3 d2h_probe.hip -o d2h_probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) __builtin_nontemporal_store(src[i], dst + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)64 * 3840 * 2160 * 3;            // one ring slot of the 4K export
    void *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 7, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    hipStream_t s[2];
    CK(hipStreamCreate(&s[0]));
    CK(hipStreamCreate(&s[1]));
    for (int variant = 0; variant < 4; ++variant) {
        const int reps = 6;
        double best = 0.0;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            if (variant == 0) {
                CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s[0]));
            } else if (variant == 1) {
                CK(hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s[0]));
                CK(hipMemcpyAsync((char*)h + bytes / 2, (char*)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s[1]));
            } else if (variant == 2) {
                void* hd = nullptr;
                CK(hipHostGetDevicePointer(&hd, h, 0));
                hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s[0], (const v4u*)d, (v4u*)hd, bytes / 16);
            } else {
                for (int k = 0; k < 8; ++k)
                    CK(hipMemcpyAsync((char*)h + k * (bytes / 8), (char*)d + k * (bytes / 8), bytes / 8,
                                      hipMemcpyDeviceToHost, s[k & 1]));
            }
            CK(hipDeviceSynchronize());
            const double gbs = bytes / (now() - t0) / 1e9;
            if (r > 0 && gbs > best) best = gbs;
        }
        const char* names[] = {"one hipMemcpyAsync, 1.6 GB", "two streams, 0.8 GB each", "copy kernel into mapped pinned memory",
                               "eight pieces over two streams"};
        printf("%-42s %.2f GB/s\n", names[variant], best);
    }
    return 0;
}
