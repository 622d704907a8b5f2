// cumask_probe.hip -- does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a launch, and how do mask bits map to
// (XCC, SE, CU)?  Every workgroup records where it ran (HW_ID, XCC_ID); the host prints, per mask, the set of distinct
// (xcc, se, cu) it saw.  Diagnostic tool, not part of the library:  hipcc --offload-arch=gfx950 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
__global__ void census(unsigned *out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long) spin) __builtin_amdgcn_s_sleep(8);   // hold the CU so that blocks spread
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char **argv)
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("CUs %d\n", pr.multiProcessorCount);
    const int nb = 2048;
    unsigned *d; hipMalloc((void **) &d, 8 * nb);
    std::vector<unsigned> h(2 * nb);
    struct M { const char *name; unsigned w[8]; } masks[] = {
        {"all", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
        {"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 32-63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"bits 0-127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"bits = 0 mod 8", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
    };
    for (auto &m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%-16s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, 8 * nb, s);
        hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, s, d, 20000);      // 200 us per block
        e = hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 8 * nb, hipMemcpyDeviceToHost);
        std::set<unsigned> places; int per_xcc[16] = {0};
        std::set<unsigned> cu_in_xcc[16];
        for (int b = 0; b < nb; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;      // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
            places.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
            cu_in_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        printf("%-16s %s: %zu distinct CUs; per XCC:", m.name, hipGetErrorString(e), places.size());
        for (int x = 0; x < 8; ++x) printf(" %zu", cu_in_xcc[x].size());
        printf("\n");
        if (places.size() <= 40) { printf("   (xcc.se.cu):"); for (unsigned p : places) printf(" %u.%u.%u", p >> 16, (p >> 8) & 7, p & 15); printf("\n"); }
        hipStreamDestroy(s);
    }
    return 0;
}
