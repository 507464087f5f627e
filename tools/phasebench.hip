// Does it pay to write y in chip-wide phases?  The SpMV kernels stream ~4-5.5 GB and write 128 MB of
// y; spread over the whole run in 512-byte pieces, those writes cost ~0.16 ms beside the saturated
// read stream (7x their bytes).  Here a persistent grid reads its slices in rounds, parks each
// round's results in LDS, meets at a timing-only rendezvous (no data is handed over, so no fences; a
// bounded spin, so a missing workgroup cannot hang the grid) and then every workgroup writes its
// results at once.
// Build: hipcc -O3 --offload-arch=gfx950 tools/phasebench.hip -o tools/phasebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nb)
{
    unsigned q = nb >> 3, r = nb & 7u, g = bid & 7u, idx = bid >> 3;
    return (g < r ? g * (q + 1) : r * (q + 1) + (g - r) * q) + idx;
}

template <int U, bool NT>
__device__ __forceinline__ double slice_sum(const double *val, const int *col, int s, int W, int lane)
{
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = NT ? __builtin_nontemporal_load(c + (size_t)(j + u) * 64) : c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = NT ? __builtin_nontemporal_load(v + (size_t)(j + u) * 64) : v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    return acc;
}

// baseline: one wave per slice, result stored at once (STORE) or dropped
template <int U, bool NT, bool STORE>
__global__ __launch_bounds__(256) void direct(const double *val, const int *col, int nslices, int W, double *y, double *out)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    double acc = slice_sum<U, NT>(val, col, s, W, lane);
    if (STORE) y[(size_t)s * 64 + lane] = acc;
    else if (acc == 12345.678) out[0] = acc;
}

// one wave per slice; the result leaves as an atomic without return value (performed in the L2, whose line then
// stays dirty until it is evicted, instead of a store that the L2 writes through at once): MODE 0 exchange, 1 add
template <int U, bool NT, int MODE>
__global__ __launch_bounds__(256) void direct_atomic(const double *val, const int *col, int nslices, int W, double *y)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    double acc = slice_sum<U, NT>(val, col, s, W, lane);
    double *p = y + (size_t)s * 64 + lane;
    if (MODE == 0) (void)__hip_atomic_exchange(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else (void)__hip_atomic_fetch_add(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave per slice; only every K-th slice stores its result (MODE 0), or every slice stores into a ring of
// `wrap` doubles (MODE 1), or every K-th WORKGROUP (4 consecutive slices) stores (MODE 2)
template <int U, bool NT, int MODE>
__global__ __launch_bounds__(256) void direct_some(const double *val, const int *col, int nslices, int W, double *y, double *out,
                                                   int K, size_t wrap)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    double acc = slice_sum<U, NT>(val, col, s, W, lane);
    bool st = MODE == 0 ? s % K == 0 : MODE == 2 ? (int)b % K == 0 : true;
    size_t at = (size_t)s * 64 + lane;
    if (MODE == 1) at &= wrap - 1;
    if (st) y[at] = acc;
    else if (acc == 12345.678) out[0] = acc;
}

// roles: the first `writers` workgroups each fill a contiguous share of y with 16-byte stores as fast as they
// can; the others read one slice per wave and store nothing.  Dense writes beside the read stream.
template <int U, bool NT>
__global__ __launch_bounds__(256) void roles(const double *val, const int *col, int nslices, int W, double *y, double *out,
                                             int writers, int delay_blocks)
{
    if ((int)blockIdx.x < writers) {
        const size_t total = (size_t)nslices * 32;                  // double2 elements of y
        const size_t a = total * blockIdx.x / writers, b = total * (blockIdx.x + 1) / writers;
        double2 *p = (double2 *)y;
        for (size_t i = a + threadIdx.x; i < b; i += 256) p[i] = double2{1.0, 2.0};
        return;
    }
    unsigned nb = gridDim.x - writers;
    unsigned b = xcd_block(blockIdx.x - writers, nb);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    double acc = slice_sum<U, NT>(val, col, s, W, lane);
    if (acc == 12345.678) out[0] = acc;
    (void)delay_blocks;
}

// bursts: the grid is reader workgroups (4 slices each, nothing stored) with a group of `wpb` writer workgroups
// after every `period` readers; each writer fills `bytes_per_writer` contiguous bytes of y, the group a
// contiguous wpb * bytes_per_writer.  Workgroups start in grid order, so a group's writers run together: a
// dense burst.  All of y is written once.
template <int U, bool NT>
__global__ __launch_bounds__(256) void bursts(const double *val, const int *col, int nslices, int W, double *y, double *out,
                                              int period, int wpb, size_t bytes_per_writer, size_t y_bytes)
{
    const unsigned cycle = (unsigned)(period + wpb);
    const unsigned c = blockIdx.x / cycle, r = blockIdx.x % cycle;
    if (r >= (unsigned)period) {                                     // writer r - period of burst c
        const size_t a = ((size_t)c * wpb + (r - period)) * bytes_per_writer;
        if (a >= y_bytes) return;
        const size_t b = a + bytes_per_writer < y_bytes ? a + bytes_per_writer : y_bytes;
        double2 *p = (double2 *)((char *)y + a);
        const size_t n = (b - a) / 16;
        for (size_t i = threadIdx.x; i < n; i += 256) p[i] = double2{1.0, 2.0};
        return;
    }
    const unsigned b = c * period + r;                               // reader index in storage order
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    double acc = slice_sum<U, NT>(val, col, s, W, lane);
    if (acc == 12345.678) out[0] = acc;
}

// persistent: workgroup g of G owns the slices [first_g, first_{g+1}) of its XCD's eighth, walks them in `rounds`
// rounds of (waves x per_wave) slices; SYNC 0 = write at the end of each round, no rendezvous; 1 = rendezvous
// of all workgroups before the writes
template <int U, bool NT, int SYNC, int WAVES, bool STORE = true>
__global__ __launch_bounds__(WAVES * 64) void phased(const double *val, const int *col, int nslices, int W, int per_wave,
                                                     int rounds, double *y, unsigned *counter, unsigned spin_limit)
{
    extern __shared__ double park[];                     // [WAVES][per_wave][64]
    const unsigned g = xcd_block(blockIdx.x, gridDim.x);
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int per_round = WAVES * per_wave;
    // slices of this workgroup: a contiguous range, walked in order
    const long long total = nslices;
    const int first = (int)(total * g / gridDim.x), last = (int)(total * (g + 1) / gridDim.x);
    double *mine = park + (size_t)wave * per_wave * 64;
    for (int r = 0; r < rounds; r++) {
        const int base = first + r * per_round + wave * per_wave;
        for (int k = 0; k < per_wave; k++) {
            const int s = __builtin_amdgcn_readfirstlane(base + k);
            if (s < last) mine[k * 64 + lane] = slice_sum<U, NT>(val, col, s, W, lane);
        }
        if (SYNC) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned target = (unsigned)(r + 1) * gridDim.x;
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < spin_limit)
                    __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
        }
        for (int k = 0; k < per_wave; k++) {
            const int s = base + k;
            if (STORE) { if (s < last) y[(size_t)s * 64 + lane] = mine[k * 64 + lane]; }
            else if (s < last && mine[k * 64 + lane] == 12345.678) y[0] = 1.0;
        }
    }
}

// writer wave: persistent workgroups of P producer waves and one writer wave.  A producer never issues a
// global store: it parks the 64 results of a slice in one of its two LDS slots and raises the slot's
// sequence number; the writer wave polls the slots, copies a parked slice to y and reports the slot free.
// What this separates: on this ISA loads and stores share one in-order counter per wave, so a wave that
// has stored cannot use the data of any later load before the store is acknowledged, and a wave that ends
// with a store keeps its slot until then.  Here only the writer waits for write acknowledgements.
__device__ __forceinline__ int lds_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_poke(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int U, bool NT, int P, bool STORE>
__global__ __launch_bounds__((P + 1) * 64) void writer_wave(const double *val, const int *col, int nslices, int W, double *y,
                                                            double *out)
{
    __shared__ double parked[P][2][64];
    __shared__ int seq[P][2];                      // 1-based number of the producer's slice parked in the slot
    __shared__ int taken[P];                       // slices of the producer the writer has copied out
    if (threadIdx.x < P) { seq[threadIdx.x][0] = 0; seq[threadIdx.x][1] = 0; taken[threadIdx.x] = 0; }
    __syncthreads();
    const unsigned g = xcd_block(blockIdx.x, gridDim.x);
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const long long total = nslices;
    const int first = (int)(total * g / gridDim.x), last = (int)(total * (g + 1) / gridDim.x);
    if (wave < P) {
        int q = 0;
        for (int s = first + wave; s < last; s += P, q++) {
            const double acc = slice_sum<U, NT>(val, col, __builtin_amdgcn_readfirstlane(s), W, lane);
            while (lds_peek(&taken[wave]) < q - 1) __builtin_amdgcn_s_sleep(1);      // slot q & 1 held slice q - 2
            asm volatile("" ::: "memory");
            parked[wave][q & 1][lane] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_poke(&seq[wave][q & 1], q + 1);
        }
    } else {
        int copied = 0;                            // lane w: slices taken from producer w
        int remaining = last - first;
        while (remaining > 0) {
            bool any = false;
            for (int w = 0; w < P; w++) {
                const int c = __builtin_amdgcn_readlane(copied, w);
                if (first + w + c * P >= last) continue;
                if (lds_peek(&seq[w][c & 1]) != c + 1) continue;
                asm volatile("" ::: "memory");
                const double v = parked[w][c & 1][lane];
                const int s = first + w + c * P;
                if (STORE) y[(size_t)s * 64 + lane] = v;
                else if (v == 12345.678) out[0] = v;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) lds_poke(&taken[w], c + 1);
                if (lane == w) copied++;
                any = true;
                remaining--;
            }
            if (!any) __builtin_amdgcn_s_sleep(1);
        }
    }
}

template <typename F> static double time_ms(F launch, int reps = 20)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main()
{
    const int W = 27, nslices = 250047;
    const size_t n = (size_t)nslices * W * 64;
    double *val, *y, *out; int *col; unsigned *counter;
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&counter, 256));
    if (getenv("PHASEBENCH_CONTIG")) {               // every array physically contiguous (or uncached: PHASEBENCH_CONTIG=2)
        const unsigned fl = atoi(getenv("PHASEBENCH_CONTIG")) == 2 ? hipDeviceMallocUncached : hipDeviceMallocContiguous;
        CK(hipExtMallocWithFlags((void **)&val, n * 8, fl)); CK(hipExtMallocWithFlags((void **)&col, n * 4, fl));
        CK(hipExtMallocWithFlags((void **)&y, (size_t)nslices * 64 * 8, fl));
    } else {
        CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&col, n * 4));
        CK(hipMalloc(&y, (size_t)nslices * 64 * 8));
    }
    CK(hipMemset(val, 0, n * 8)); CK(hipMemset(col, 0, n * 4));
    const double gb = n * 12 / 1e9;
    unsigned g = (nslices + 3) / 4;
    const bool writer_only = getenv("PHASEBENCH_WRITER_ONLY") != nullptr || getenv("PHASEBENCH_YTYPES") != nullptr || getenv("PHASEBENCH_ATOMIC") != nullptr;
#define RUN(NAME, ...) { double t = time_ms([&] { __VA_ARGS__; }); printf("%-58s %.4f ms  %.1f GB/s\n", NAME, t, gb / t * 1e3); fflush(stdout); }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    if (!writer_only) {
    RUN("direct U9 plain nostore", hipLaunchKernelGGL((direct<9, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    RUN("direct U9 nt    nostore", hipLaunchKernelGGL((direct<9, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    RUN("direct U9 plain store", hipLaunchKernelGGL((direct<9, false, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    RUN("direct U9 nt    store", hipLaunchKernelGGL((direct<9, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    for (int K : {1, 2, 4, 16, 64, 256, 4096}) {
        char name[96]; snprintf(name, sizeof name, "direct U9 nt store every %d-th slice", K);
        RUN(name, hipLaunchKernelGGL((direct_some<9, true, 0>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out, K, 0));
    }
    for (int K : {4, 64}) {
        char name[96]; snprintf(name, sizeof name, "direct U9 nt store every %d-th workgroup (4 slices)", K);
        RUN(name, hipLaunchKernelGGL((direct_some<9, true, 2>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out, K, 0));
    }
    for (size_t wrap : {(size_t)1 << 10, (size_t)1 << 15, (size_t)1 << 18, (size_t)1 << 22}) {
        char name[96]; snprintf(name, sizeof name, "direct U9 nt store into a ring of %zu KB", wrap * 8 / 1024);
        RUN(name, hipLaunchKernelGGL((direct_some<9, true, 1>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out, 1, wrap));
    }
    for (int writers : {8, 32, 128, 512, 2048}) {
        char name[96]; snprintf(name, sizeof name, "roles: %d writer workgroups fill y, the rest read", writers);
        RUN(name, hipLaunchKernelGGL((roles<9, true>), dim3(g + writers), dim3(256), 0, 0, val, col, nslices, W, y, out, writers, 0));
    }
    {
        const size_t y_bytes = (size_t)nslices * 512;
        struct { int wpb; size_t bpw; } shapes[] = {{1, 2048}, {1, 65536}, {4, 65536}, {16, 65536}, {32, 131072}, {64, 131072},
                                                   {256, 131072}, {32, 32768}, {128, 8192}, {256, 2048}, {1024, 2048}, {64, 16384}};
        for (auto sh : shapes) {
            const size_t burst = (size_t)sh.wpb * sh.bpw;
            const int period = (int)(burst / 2048);                  // readers that produce one burst's worth of y
            const unsigned nbursts = (unsigned)((y_bytes + burst - 1) / burst);
            const unsigned grid = nbursts * (unsigned)(period + sh.wpb);
            char name[96]; snprintf(name, sizeof name, "bursts of %zu KB: %d writers x %zu KB", burst / 1024, sh.wpb, sh.bpw / 1024);
            RUN(name, hipLaunchKernelGGL((bursts<9, true>), dim3(grid), dim3(256), 0, 0, val, col, nslices, W, y, out, period, sh.wpb,
                                         sh.bpw, y_bytes));
        }
    }
    }
    auto run_phased = [&](auto kernel, const char *what, int waves, int wgs_per_cu, int rounds, int sync) {
        const int G = cus * wgs_per_cu;
        const int per_wg = (nslices + G - 1) / G + 1;
        const int per_wave = (per_wg + waves * rounds - 1) / (waves * rounds);
        const size_t lds = (size_t)waves * per_wave * 64 * 8;
        if (lds > 160 * 1024 / wgs_per_cu) { printf("%-58s skipped: %zu B of LDS per workgroup\n", what, lds); return; }
        CK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        char name[128];
        snprintf(name, sizeof name, "%s %d wg/cu x %d waves, %d rounds, %d slices/wave/round, lds %zu", what, wgs_per_cu, waves, rounds,
                 per_wave, lds);
        double t = time_ms([&] {
            CK(hipMemsetAsync(counter, 0, 4, 0));
            hipLaunchKernelGGL(kernel, dim3(G), dim3(waves * 64), lds, 0, val, col, nslices, W, per_wave, rounds, y, counter, 200000u);
        });
        printf("%-58s %.4f ms  %.1f GB/s\n", name, t, gb / t * 1e3); fflush(stdout);
        (void)sync;
    };
    if (getenv("PHASEBENCH_ATOMIC")) {
        RUN("direct U9 nt    nostore", hipLaunchKernelGGL((direct<9, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
        RUN("direct U9 nt    store", hipLaunchKernelGGL((direct<9, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
        RUN("direct U9 nt    atomic exchange instead of the store", hipLaunchKernelGGL((direct_atomic<9, true, 0>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
        RUN("direct U9 nt    atomic add instead of the store", hipLaunchKernelGGL((direct_atomic<9, true, 1>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
        return 0;
    }
    if (getenv("PHASEBENCH_YTYPES")) {
        // the same store into y allocated in other ways
        struct { const char *name; unsigned flags; } kinds[] = {{"hipMalloc", 0xffffffffu}, {"fine-grained", hipDeviceMallocFinegrained},
                                                                {"uncached", hipDeviceMallocUncached}, {"contiguous", hipDeviceMallocContiguous}};
        for (auto k : kinds) {
            double *yy = nullptr;
            hipError_t e = k.flags == 0xffffffffu ? hipMalloc(&yy, (size_t)nslices * 512) : hipExtMallocWithFlags((void **)&yy, (size_t)nslices * 512, k.flags);
            if (e != hipSuccess) { printf("y %s: allocation failed (%s)\n", k.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            char name[96]; snprintf(name, sizeof name, "direct U9 nt store, y %s", k.name);
            RUN(name, hipLaunchKernelGGL((direct<9, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, yy, out));
            CK(hipFree(yy));
        }
        double *yh = nullptr;
        if (hipHostMalloc((void **)&yh, (size_t)nslices * 512, hipHostMallocMapped) == hipSuccess) {
            RUN("direct U9 nt store, y in pinned host memory", hipLaunchKernelGGL((direct<9, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, yh, out));
            CK(hipHostFree(yh));
        }
        return 0;
    }
    if (getenv("PHASEBENCH_WRITER_ONLY")) {
        RUN("direct U9 nt    nostore", hipLaunchKernelGGL((direct<9, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
        RUN("direct U9 nt    store", hipLaunchKernelGGL((direct<9, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    }
#define WRITER(P, PERCU) \
    RUN("writer wave: " #P " producers + 1 writer, " #PERCU " wg/cu, nostore", hipLaunchKernelGGL((writer_wave<9, true, P, false>), dim3(cus * PERCU), dim3((P + 1) * 64), 0, 0, val, col, nslices, W, y, out)); \
    RUN("writer wave: " #P " producers + 1 writer, " #PERCU " wg/cu, store", hipLaunchKernelGGL((writer_wave<9, true, P, true>), dim3(cus * PERCU), dim3((P + 1) * 64), 0, 0, val, col, nslices, W, y, out));
    WRITER(15, 2)
    WRITER(7, 4)
    WRITER(3, 8)
    WRITER(15, 1)
    RUN("direct U27 nt    nostore", hipLaunchKernelGGL((direct<27, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    RUN("direct U27 nt    store", hipLaunchKernelGGL((direct<27, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    return 0;
}
