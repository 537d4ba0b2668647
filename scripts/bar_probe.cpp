// Probe: is device memory writable/readable from the host through the BAR, and how long does a round trip
// through it take compared with pinned host memory?
#include <hip/hip_runtime.h>
#include <chrono>
#include <csignal>
#include <csetjmp>
#include <cstdio>
#include <cstdint>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void echo(volatile uint64_t* in, volatile uint64_t* out, int rounds) {
    uint64_t last = 0;
    for (int r = 0; r < rounds; ++r) {
        uint64_t v;
        long spins = 0;
        do { v = __hip_atomic_load((uint64_t*)in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (v == last && ++spins < 3000000L);
        if (v == last) return;
        last = v;
        __hip_atomic_store((uint64_t*)out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double run(const char* name, volatile uint64_t* in_h, uint64_t* in_d, volatile uint64_t* out_h, uint64_t* out_d, int rounds) {
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *in_h = 0; *out_h = 0;
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, s, (volatile uint64_t*)in_d, (volatile uint64_t*)out_d, rounds);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
        *in_h = (uint64_t)r;
        __builtin_ia32_sfence();
        long spins = 0;
        while (*out_h != (uint64_t)r && ++spins < 300000000L) {}
        if (*out_h != (uint64_t)r) { printf("%s: timeout at round %d\n", name, r); fflush(stdout); break; }
    }
    auto t1 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds;
    printf("%-52s %.2f us per round trip\n", name, us); fflush(stdout);
    return us;
}
int main() {
    const int rounds = 5000; setvbuf(stdout, nullptr, _IONBF, 0);
    uint64_t *pin_in, *pin_out, *pin_in_d, *pin_out_d;
    (void)hipHostMalloc((void**)&pin_in, 64, hipHostMallocMapped); (void)hipHostMalloc((void**)&pin_out, 64, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&pin_in_d, pin_in, 0); (void)hipHostGetDevicePointer((void**)&pin_out_d, pin_out, 0);
    run("request in pinned host memory, reply in pinned host", pin_in, pin_in_d, pin_out, pin_out_d, rounds);
    uint64_t* dev = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&dev, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { e = hipMalloc((void**)&dev, 4096); printf("hipMalloc: %s\n", hipGetErrorString(e)); }
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
        volatile uint64_t* hv = (volatile uint64_t*)dev;
        *hv = 42;                               // host store into device memory
        uint64_t back = *hv;                    // host load from device memory
        printf("host access to device memory works: wrote 42, read %llu\n", (unsigned long long)back);
        run("request in DEVICE memory (host writes over the BAR), reply in pinned host", hv, dev, pin_out, pin_out_d, rounds);
        run("request and reply in DEVICE memory", hv, dev, hv + 8, dev + 8, rounds);
    } else {
        printf("host access to device memory FAULTS (no CPU-visible VRAM)\n");
    }
    return 0;
}
