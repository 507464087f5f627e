// Read-bandwidth ceilings on MI355X for the access shapes the SpMV kernels use.
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nb, bool remap)
{
    if (!remap) return bid;
    unsigned q = nb >> 3, r = nb & 7u, g = bid & 7u, idx = bid >> 3;
    return (g < r ? g * (q + 1) : r * (q + 1) + (g - r) * q) + idx;
}

// grid-stride read of 16 B per lane
__global__ __launch_bounds__(256) void read16(const double2 *p, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void read8(const double *p, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
    if (s == 12345.678) out[0] = s;
}
// one wave per "slice" of W columns x 64 lanes: 8-B values + 4-B indices (two streams), U loads in flight
template <int U, bool REMAP, bool NT>
__global__ __launch_bounds__(256) void slice_stream(const double *val, const int *col, int nslices, int W, double *out)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, REMAP);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
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
    if (acc == 12345.678) out[0] = acc;
}
// slice stream + one 8-B result per lane per slice (y), different store flavours
// ST: 0 plain, 1 nontemporal, 2 through LDS: one wave of the block writes the block's 256 results as 16 B/lane
template <int U, int ST>
__global__ __launch_bounds__(256) void slice_stream_store(const double *val, const int *col, int nslices, int W, double *y)
{
    __shared__ double stage[256];
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    int lane = threadIdx.x & 63;
    double acc = 0;
    if (s < nslices) {
        const double *v = val + (size_t)s * W * 64 + lane;
        const int *c = col + (size_t)s * W * 64 + lane;
        int j = 0;
        for (; j + U <= W; j += U) {
            int cc[U]; double vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
        }
        for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    }
    if (ST == 0) { if (s < nslices) y[(size_t)s * 64 + lane] = acc; }
    else if (ST == 1) { if (s < nslices) __builtin_nontemporal_store(acc, y + (size_t)s * 64 + lane); }
    else {
        stage[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < 128) {
            size_t at = (size_t)b * 256 + threadIdx.x * 2;
            if (at + 1 < (size_t)nslices * 64) *(double2 *)(y + at) = *(double2 *)(stage + threadIdx.x * 2);
        }
    }
}

__global__ __launch_bounds__(256) void write16(double2 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = double2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void copy16(const double2 *a, double2 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// like slice_stream_store<U,0> but the stores wrap inside a window of `wrap` doubles (power of two)
template <int U>
__global__ __launch_bounds__(256) void slice_stream_store_wrap(const double *val, const int *col, int nslices, int W, double *y, size_t wrap)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    y[((size_t)s * 64 + lane) & (wrap - 1)] = acc;
}
// persistent workgroups of 1024 threads: block b owns slices [b*S, (b+1)*S); results are staged in LDS
// and dumped every D slices-per-wave rounds, so all blocks write at about the same times
template <int U>
__global__ __launch_bounds__(1024) void slice_stream_phased(const double *val, const int *col, int nslices, int W, int S, int D, double *y)
{
    extern __shared__ double stage[];                 // D * 16 waves * 64 doubles
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int s_begin = blockIdx.x * S, s_end = min(s_begin + S, nslices);
    for (int r0 = s_begin; r0 < s_end; r0 += 16 * D) {
        for (int d = 0; d < D; d++) {
            int s = r0 + d * 16 + wave;
            double acc = 0;
            if (s < s_end) {
                const double *v = val + (size_t)s * W * 64 + lane;
                const int *c = col + (size_t)s * W * 64 + lane;
                int j = 0;
                for (; j + U <= W; j += U) {
                    int cc[U]; double vv[U];
#pragma unroll
                    for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
                    for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
                    for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
                }
                for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
            }
            stage[(d * 16 + wave) * 64 + lane] = acc;
        }
        __syncthreads();
        int nres = min(16 * D, s_end - r0) * 64;
        for (int i = threadIdx.x * 2; i < nres; i += 2048)
            *(double2 *)(y + (size_t)r0 * 64 + i) = *(double2 *)(stage + i);
        __syncthreads();
    }
}

// store issued BEFORE the stream (garbage value): no store outstanding at wave end
template <int U>
__global__ __launch_bounds__(256) void slice_stream_store_first(const double *val, const int *col, int nslices, int W, double *y, double *out)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    y[(size_t)s * 64 + lane] = (double)lane;
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    if (acc == 12345.678) out[0] = acc;
}
// K slices per wave, the result of slice k is stored after the first batch of slice k+1 has been issued
template <int U, int K>
__global__ __launch_bounds__(256) void slice_stream_store_delayed(const double *val, const int *col, int nslices, int W, double *y)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int w = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    int lane = threadIdx.x & 63;
    double prev = 0; int prev_s = -1;
    for (int k = 0; k < K; k++) {
        int s = w * K + k;
        if (s >= nslices) break;
        const double *v = val + (size_t)s * W * 64 + lane;
        const int *c = col + (size_t)s * W * 64 + lane;
        double acc = 0;
        int j = 0;
        for (; j + U <= W; j += U) {
            int cc[U]; double vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
            if (j == 0 && prev_s >= 0) y[(size_t)prev_s * 64 + lane] = prev;
#pragma unroll
            for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
        }
        for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
        prev = acc; prev_s = s;
    }
    if (prev_s >= 0) y[(size_t)prev_s * 64 + lane] = prev;
}

// ST: 0 agent-scope (sc1) store, 1 system-scope store, 2 plain store on every other slice,
//     3 store into the slice's own (just read) value area, 4 float store
template <int U, int ST>
__global__ __launch_bounds__(256) void slice_stream_store2(double *val, const int *col, int nslices, int W, double *y)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    double *dst = y + (size_t)s * 64 + lane;
    if (ST == 0) __hip_atomic_store(dst, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (ST == 1) __hip_atomic_store(dst, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (ST == 2) { if (s & 1) *dst = acc; }
    else if (ST == 3) val[(size_t)s * W * 64 + lane] = acc * 0.0;
    else ((float *)y)[(size_t)s * 64 + lane] = (float)acc;
}

// full SpMV shape: NT/plain stream loads, gather of x through the (real) column indices, plain store
template <int U, bool NT, bool GATHER, bool STORE>
__global__ __launch_bounds__(256) void slice_spmv(const double *val, const int *col, int nslices, int W, const double *x, double *y, double *out)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = NT ? __builtin_nontemporal_load(c + (size_t)(j + u) * 64) : c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = NT ? __builtin_nontemporal_load(v + (size_t)(j + u) * 64) : v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) xx[u] = GATHER ? x[cc[u]] : (double)cc[u];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * xx[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * (GATHER ? x[c[(size_t)j * 64]] : 1.0);
    if (STORE) y[(size_t)s * 64 + lane] = acc;
    else if (acc == 12345.678) out[0] = acc;
}
// column indices of a 27-point stencil on an n^3 grid, interior pattern clamped into range
__global__ void fill_cols(int *col, int nslices, int W, int n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t total = (size_t)nslices * W * 64;
    if (i >= total) return;
    int lane = i & 63; size_t sj = i >> 6; int j = sj % W; size_t s = sj / W;
    long r = (long)s * 64 + lane;
    int dz = j / 9 - 1, dy = (j / 3) % 3 - 1, dx = j % 3 - 1;
    long cidx = r + (long)dz * n * n + (long)dy * n + dx;
    long nr = (long)nslices * 64;
    col[i] = (int)(cidx < 0 ? 0 : (cidx >= nr ? nr - 1 : cidx));
}

// stores confined to a 256-KB window (never reach HBM during the kernel); LANES lanes store BYTES each
template <int U, int LANES, int BYTES>
__global__ __launch_bounds__(256) void slice_stream_store_small(const double *val, const int *col, int nslices, int W, double *y)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, true);
    int s = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    if (s >= nslices) return;
    int lane = threadIdx.x & 63;
    const double *v = val + (size_t)s * W * 64 + lane;
    const int *c = col + (size_t)s * W * 64 + lane;
    double acc = 0;
    int j = 0;
    for (; j + U <= W; j += U) {
        int cc[U]; double vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
    }
    for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    size_t at = ((size_t)s * 64 + lane) & 32767;
    if (lane < LANES) {
        if (BYTES == 8) y[at] = acc; else ((float *)y)[at] = (float)acc;
    }
}

// persistent: each wave walks K consecutive slices
template <int U, bool REMAP>
__global__ __launch_bounds__(256) void slice_stream_multi(const double *val, const int *col, int nslices, int W, int K, double *out)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, REMAP);
    int w = __builtin_amdgcn_readfirstlane((int)(b * 4 + threadIdx.x / 64));
    int lane = threadIdx.x & 63;
    double acc = 0;
    for (int s = w * K; s < min((w + 1) * K, nslices); s++) {
        const double *v = val + (size_t)s * W * 64 + lane;
        const int *c = col + (size_t)s * W * 64 + lane;
        int j = 0;
        for (; j + U <= W; j += U) {
            int cc[U]; double vv[U];
#pragma unroll
            for (int u = 0; u < U; u++) cc[u] = c[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) vv[u] = v[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < U; u++) acc += vv[u] * cc[u];
        }
        for (; j < W; j++) acc += v[(size_t)j * 64] * c[(size_t)j * 64];
    }
    if (acc == 12345.678) out[0] = acc;
}

template <typename F> static double time_ms(F f, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const int W = 27, nslices = 250047;
    const size_t n = (size_t)nslices * W * 64;          // entries
    double *val; int *col; double *out;
    CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&col, n * 4)); CK(hipMalloc(&out, 8));
    CK(hipMemset(val, 0, n * 8)); CK(hipMemset(col, 0, n * 4));
    double gb = n * 12 / 1e9;
    printf("entries %zu, %.3f GB (8B val + 4B col)\n", n, gb);
    for (int blocks : {2048, 4096, 8192, 16384}) {
        double t = time_ms([&] { hipLaunchKernelGGL(read16, dim3(blocks), dim3(256), 0, 0, (const double2 *)val, n / 2, out); });
        printf("read16 grid-stride %5d blocks: %.4f ms  %.1f GB/s\n", blocks, t, n * 8 / t / 1e6);
    }
    { double t = time_ms([&] { hipLaunchKernelGGL(read16, dim3((unsigned)(n / 2 / 256)), dim3(256), 0, 0, (const double2 *)val, n / 2, out); });
      printf("read16 one elem/thread: %.4f ms  %.1f GB/s\n", t, n * 8 / t / 1e6); }
    for (int blocks : {2048, 8192}) {
        double t = time_ms([&] { hipLaunchKernelGGL(read8, dim3(blocks), dim3(256), 0, 0, (const double *)val, n, out); });
        printf("read8  grid-stride %5d blocks: %.4f ms  %.1f GB/s\n", blocks, t, n * 8 / t / 1e6);
    }
    unsigned g = (nslices + 3) / 4;
#define RUN(NAME, ...) { double t = time_ms([&] { __VA_ARGS__; }); printf("%-44s %.4f ms  %.1f GB/s\n", NAME, t, gb / t * 1e3); }
    RUN("slice_stream U8 remap", hipLaunchKernelGGL((slice_stream<8, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U8 noremap", hipLaunchKernelGGL((slice_stream<8, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U8 remap nt", hipLaunchKernelGGL((slice_stream<8, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U9 remap", hipLaunchKernelGGL((slice_stream<9, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U9 noremap", hipLaunchKernelGGL((slice_stream<9, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U27 remap", hipLaunchKernelGGL((slice_stream<27, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U27 noremap", hipLaunchKernelGGL((slice_stream<27, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    RUN("slice_stream U4 remap", hipLaunchKernelGGL((slice_stream<4, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    double *y; CK(hipMalloc(&y, (size_t)nslices * 64 * 8 + 4096));
    RUN("slice_stream_store U8 plain", hipLaunchKernelGGL((slice_stream_store<8, 0>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("slice_stream_store U8 nt", hipLaunchKernelGGL((slice_stream_store<8, 1>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("slice_stream_store U8 lds16", hipLaunchKernelGGL((slice_stream_store<8, 2>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    for (size_t wrap : {(size_t)1 << 15, (size_t)1 << 20, (size_t)1 << 24}) {
        char name[64]; snprintf(name, sizeof name, "slice_stream_store_wrap %zu KB", wrap * 8 / 1024);
        RUN(name, hipLaunchKernelGGL((slice_stream_store_wrap<8>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, wrap));
    }
    {
        size_t nw = (size_t)1 << 27;   // 2 GB of double2
        double2 *wbuf; CK(hipMalloc(&wbuf, nw * 16));
        double t = time_ms([&] { hipLaunchKernelGGL(write16, dim3(2048), dim3(256), 0, 0, wbuf, nw); });
        printf("write16 2 GB: %.4f ms %.1f GB/s\n", t, nw * 16 / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const double2 *)val, wbuf, nw); });
        printf("copy16 2+2 GB: %.4f ms %.1f GB/s\n", t, 2 * nw * 16 / t / 1e6);
        CK(hipFree(wbuf));
    }
    for (int D : {2, 4, 8}) {
        for (int nb : {256, 512}) {
            int Sb = (nslices + nb - 1) / nb;
            size_t lds = (size_t)D * 16 * 64 * 8;
            CK(hipFuncSetAttribute((const void *)slice_stream_phased<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            char name[64]; snprintf(name, sizeof name, "slice_stream_phased D=%d blocks=%d", D, nb);
            RUN(name, hipLaunchKernelGGL((slice_stream_phased<8>), dim3(nb), dim3(1024), lds, 0, val, col, nslices, W, Sb, D, y));
        }
    }
    RUN("slice_stream_store_first U8", hipLaunchKernelGGL((slice_stream_store_first<8>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y, out));
    { unsigned gk = ((nslices + 3) / 4 + 3) / 4; RUN("slice_stream_store_delayed U8 K=4", hipLaunchKernelGGL((slice_stream_store_delayed<8, 4>), dim3(gk), dim3(256), 0, 0, val, col, nslices, W, y)); }
    { unsigned gk = ((nslices + 15) / 16 + 3) / 4; RUN("slice_stream_store_delayed U8 K=16", hipLaunchKernelGGL((slice_stream_store_delayed<8, 16>), dim3(gk), dim3(256), 0, 0, val, col, nslices, W, y)); }
    RUN("store2 sc1 (agent)", hipLaunchKernelGGL((slice_stream_store2<8, 0>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store2 sc0 sc1 (system)", hipLaunchKernelGGL((slice_stream_store2<8, 1>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store2 every other slice", hipLaunchKernelGGL((slice_stream_store2<8, 2>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store2 in place (own slice)", hipLaunchKernelGGL((slice_stream_store2<8, 3>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store2 float", hipLaunchKernelGGL((slice_stream_store2<8, 4>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    {
        double *xv; CK(hipMalloc(&xv, (size_t)nslices * 64 * 8)); CK(hipMemset(xv, 0, (size_t)nslices * 64 * 8));
        hipLaunchKernelGGL(fill_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, col, nslices, W, 252);
        CK(hipDeviceSynchronize());
        RUN("spmv plain  nogather nostore", hipLaunchKernelGGL((slice_spmv<8, false, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv nt     nogather nostore", hipLaunchKernelGGL((slice_spmv<8, true, false, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv plain  gather   nostore", hipLaunchKernelGGL((slice_spmv<8, false, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv nt     gather   nostore", hipLaunchKernelGGL((slice_spmv<8, true, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv plain  nogather store", hipLaunchKernelGGL((slice_spmv<8, false, false, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv nt     nogather store", hipLaunchKernelGGL((slice_spmv<8, true, false, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv plain  gather   store", hipLaunchKernelGGL((slice_spmv<8, false, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
        RUN("spmv nt     gather   store", hipLaunchKernelGGL((slice_spmv<8, true, true, true>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, xv, y, out));
    }
    RUN("store_small 64 lanes x 8B (wrap)", hipLaunchKernelGGL((slice_stream_store_small<8, 64, 8>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store_small 64 lanes x 4B (wrap)", hipLaunchKernelGGL((slice_stream_store_small<8, 64, 4>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store_small  8 lanes x 8B (wrap)", hipLaunchKernelGGL((slice_stream_store_small<8, 8, 8>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("store_small  1 lane  x 8B (wrap)", hipLaunchKernelGGL((slice_stream_store_small<8, 1, 8>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, y));
    RUN("slice_stream U8 remap (again)", hipLaunchKernelGGL((slice_stream<8, true, false>), dim3(g), dim3(256), 0, 0, val, col, nslices, W, out));
    for (int K : {4}) {
        unsigned gk = ((nslices + K - 1) / K + 3) / 4;
        char name[64];
        snprintf(name, sizeof name, "slice_stream_multi U9 remap K=%d", K);
        RUN(name, hipLaunchKernelGGL((slice_stream_multi<9, true>), dim3(gk), dim3(256), 0, 0, val, col, nslices, W, K, out));
        snprintf(name, sizeof name, "slice_stream_multi U9 noremap K=%d", K);
        RUN(name, hipLaunchKernelGGL((slice_stream_multi<9, false>), dim3(gk), dim3(256), 0, 0, val, col, nslices, W, K, out));
    }
    return 0;
}
