// Which host allocations can be BOTH a DMA target and a vmsplice source?  For each variant: hipHostRegister result,
// device-to-host copy rate, vmsplice into a pipe drained by a forked reader.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/hostreg_probe.hip -o tools/probes/hostreg_probe
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static double splice_rate(void* p, size_t n, int reps, int* err) {
    int fd[2];
    if (pipe(fd)) return -1;
    fcntl(fd[1], F_SETPIPE_SZ, 1 << 20);
    pid_t pid = fork();                     // the reader never touches HIP: read() and exit
    if (pid == 0) {
        close(fd[1]);
        static char buf[4 << 20];
        while (read(fd[0], buf, sizeof buf) > 0) {}
        _exit(0);
    }
    close(fd[0]);
    *err = 0;
    double t0 = now();
    for (int r = 0; r < reps && !*err; ++r) {
        size_t off = 0;
        while (off < n) {
            iovec iov{(char*)p + off, n - off};
            ssize_t k = vmsplice(fd[1], &iov, 1, 0);
            if (k < 0) { *err = errno; break; }
            off += (size_t)k;
        }
    }
    close(fd[1]);
    int st; waitpid(pid, &st, 0);
    return *err ? 0.0 : reps * (double)n / (now() - t0) / 1e9;
}

int main() {
    const size_t n = (size_t)512 << 20;
    void* dsrc; if (hipMalloc(&dsrc, n) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipStream_t st; hipStreamCreate(&st);
    struct V { const char* name; int huge, prefault, flags; bool anon; };
    V vs[] = {{"hipHostMalloc", 0, 0, 0, false},
              {"mmap + register(Default)", 0, 0, hipHostRegisterDefault, true},
              {"mmap + prefault + register(Default)", 0, 1, hipHostRegisterDefault, true},
              {"mmap + THP + prefault + register(Default)", 1, 1, hipHostRegisterDefault, true},
              {"mmap + prefault + register(Portable|Mapped)", 0, 1, hipHostRegisterPortable | hipHostRegisterMapped, true},
              {"posix_memalign + prefault + register(Default)", 2, 1, hipHostRegisterDefault, true}};
    for (V& v : vs) {
        void* p = nullptr;
        hipError_t e = hipSuccess;
        if (!v.anon) e = hipHostMalloc(&p, n, hipHostMallocDefault);
        else {
            if (v.huge == 2) { if (posix_memalign(&p, 4096, n)) p = nullptr; }
            else { p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (p == MAP_FAILED) p = nullptr; }
            if (!p) { printf("%-48s alloc failed\n", v.name); continue; }
            if (v.huge == 1) madvise(p, n, MADV_HUGEPAGE);
            if (v.prefault) memset(p, 1, n);
            e = hipHostRegister(p, n, v.flags);
        }
        if (e != hipSuccess) { printf("%-48s register: %s\n", v.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemcpyAsync(p, dsrc, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        double t0 = now();
        for (int k = 0; k < 4; ++k) hipMemcpyAsync(p, dsrc, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        double d2h = 4 * (double)n / (now() - t0) / 1e9;
        int err; double sp = splice_rate(p, n, 6, &err);
        printf("%-48s d2h %.1f GB/s   vmsplice %.1f GB/s (errno %d)\n", v.name, d2h, sp, err);
        fflush(stdout);
    }
    return 0;
}
