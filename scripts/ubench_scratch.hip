// Probe: is a tiny private (scratch) segment private?  Every lane parks one dword in a 4-, 12- or 164-byte
// stack object, spins, reads it back.  Written after a WAVE build with a 12-byte private segment returned wrong
// results while a 164-byte one did not.   hipcc --offload-arch=gfx950 -O3 ubench_scratch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int WORDS, int LDS_KB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1))) void probe(unsigned* bad, int spin) {
    __shared__ unsigned pad[LDS_KB * 256];
    volatile unsigned slot[WORDS];
    const unsigned mine = blockIdx.x * 64u + threadIdx.x + 0x1000000u;
    pad[threadIdx.x] = mine;
    for (int i = 0; i < WORDS; ++i) slot[i] = mine + (unsigned)i * 0x10000000u;
    unsigned acc = 0;
    for (int it = 0; it < spin; ++it) {
        acc += pad[(threadIdx.x + it) & 63];
        for (int i = 0; i < WORDS; ++i)
            if (slot[i] != mine + (unsigned)i * 0x10000000u) atomicAdd(bad, 1u);
    }
    if (acc == 0x12345u) bad[1] = acc;
}
template <int WORDS, int LDS_KB> void run(unsigned* d, int blocks) {
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<WORDS, LDS_KB>), dim3(blocks), dim3(64), 0, 0, d, 2000);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("scratch object %3d bytes, LDS %2d KB per workgroup, %5d workgroups of one wave: %u wrong reads\n", WORDS * 4, LDS_KB, blocks, h[0]);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 8);
    for (int blocks : {256, 1024, 4096}) {
        run<1, 1>(d, blocks);
        run<1, 40>(d, blocks);
        run<3, 40>(d, blocks);
        run<41, 40>(d, blocks);
    }
    return 0;
}
