// Random 8-byte gathers from a table larger than every cache, with each cache-policy flavour of
// global_load_dwordx2 (sc0 / sc1 / nt bits).  Reports time; run under rocprofv3 --pmc
// TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum to see the fabric request sizes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gatherbench.hip -o tools/gatherbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int MODE> __device__ __forceinline__ double gload(const double *p)
{
    double v;
    if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// each lane issues PER independent gathers (index hash of a counter), MODE picks the cache policy
template <int MODE, int PER>
__global__ __launch_bounds__(256) void gather_kernel(const double *table, size_t mask, double *out)
{
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) s += gload<MODE>(table + (mix(t * PER + k) & mask));
    if (s == 12345.678) out[0] = s;
}

template <int MODE> static void run(const double *table, size_t mask, double *out, size_t gathers)
{
    constexpr int PER = 8;
    unsigned blocks = (unsigned)(gathers / PER / 256);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<MODE, PER>), dim3(blocks), dim3(256), 0, 0, table, mask, out);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < 5; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_kernel<MODE, PER>), dim3(blocks), dim3(256), 0, 0, table, mask, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("mode %d (sc0=%d sc1=%d nt=%d): %.3f ms  %.1f G gathers/s\n", MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, ts[2],
           gathers / ts[2] / 1e6);
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)1 << 27;               // 1 GiB of doubles: beyond L2 and the 256 MiB cache
    double *table, *out;
    CK(hipMalloc(&table, n * 8)); CK(hipMalloc(&out, 8));
    CK(hipMemset(table, 0, n * 8));
    const size_t gathers = (size_t)1 << 28;
    if (argc > 1 && !strcmp(argv[1], "sizes")) {    // the gather rate against the size of the table
        for (int lg = 17; lg <= 27; lg++) {
            printf("table %7.1f MiB: ", (double)((size_t)8 << lg) / (1 << 20));
            run<0>(table, ((size_t)1 << lg) - 1, out, gathers);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "l2modes")) {  // every cache policy on tables that fit the L2 / the L1
        for (int lg : {11, 14, 18, 21}) {
            printf("table %.0f KiB:\n", (double)((size_t)8 << lg) / 1024);
            const size_t mask = ((size_t)1 << lg) - 1;
            run<0>(table, mask, out, gathers); run<1>(table, mask, out, gathers); run<2>(table, mask, out, gathers);
            run<3>(table, mask, out, gathers); run<4>(table, mask, out, gathers); run<6>(table, mask, out, gathers);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "memtypes")) { // the same gathers from memory of other types
        struct { const char *name; unsigned flag; } kinds[] = {{"uncached", hipDeviceMallocUncached}, {"fine-grained", hipDeviceMallocFinegrained}};
        for (auto k : kinds) {
            double *t2 = nullptr;
            if (hipExtMallocWithFlags((void **)&t2, n * 8, k.flag) != hipSuccess) { printf("%s: allocation refused\n", k.name); continue; }
            CK(hipMemset(t2, 0, n * 8));
            printf("%s memory, 1 GiB table:\n", k.name);
            run<0>(t2, n - 1, out, gathers); run<2>(t2, n - 1, out, gathers); run<4>(t2, n - 1, out, gathers);
            printf("%s memory, 2 MiB table:\n", k.name);
            run<0>(t2, ((size_t)1 << 18) - 1, out, gathers);
            CK(hipFree(t2));
        }
        printf("default memory, 1 GiB table:\n");
        run<0>(table, n - 1, out, gathers);
        return 0;
    }
    run<0>(table, n - 1, out, gathers); run<1>(table, n - 1, out, gathers); run<2>(table, n - 1, out, gathers);
    run<3>(table, n - 1, out, gathers); run<4>(table, n - 1, out, gathers); run<5>(table, n - 1, out, gathers);
    run<6>(table, n - 1, out, gathers); run<7>(table, n - 1, out, gathers);
    return 0;
}
