// Diagnostic: what copy / read / write rates does this MI355X reach, and with which access pattern?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_explore tools/copy_explore.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct alignas(16) U4 { uint32_t x, y, z, w; };
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ U4 ld(const U4 *p)
{
    if (NT) { const uint32_t *q = (const uint32_t *)p; U4 v; v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1); v.z = __builtin_nontemporal_load(q + 2); v.w = __builtin_nontemporal_load(q + 3); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(U4 *p, const U4 &v)
{
    if (NT) { uint32_t *q = (uint32_t *)p; __builtin_nontemporal_store(v.x, q); __builtin_nontemporal_store(v.y, q + 1); __builtin_nontemporal_store(v.z, q + 2); __builtin_nontemporal_store(v.w, q + 3); }
    else *p = v;
}

// grid-stride, U chunks in flight per lane
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(1024) void copy_gs(const U4 *src, U4 *dst, uint64_t chunks)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < chunks; i += U * stride) {
        U4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NTL>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) st<NTS>(dst + i + u * stride, v[u]);
    }
    for (; i < chunks; i += stride) st<NTS>(dst + i, ld<NTL>(src + i));
}

// the packet kernel's pattern: workgroup = (frame, slice); it walks the rows of a frame touching `piece` bytes of each
// `rowbytes`-byte row (lanes_per_row lanes x 16 B), U row batches in flight
template <bool NT, int U, bool XCD = false>
__global__ __launch_bounds__(1024) void copy_sliced(const uint8_t *src, uint8_t *dst, int rows, int rowbytes, int piece, int nslices)
{
    int64_t f = blockIdx.x / nslices;
    int sl = blockIdx.x % nslices;
    if (XCD) {   // workgroups are dealt round-robin over the 8 XCDs: keep the slices of a frame on one XCD
        const int64_t x = blockIdx.x & 7, i = blockIdx.x >> 3;
        f = (i / nslices) * 8 + x;
        sl = (int)(i % nslices);
    }
    const int lpr = piece / 16;
    const int rpb = blockDim.x / lpr;          // rows per pass of the workgroup
    const int r0 = threadIdx.x / lpr, gl = threadIdx.x % lpr;
    const uint8_t *s = src + f * (int64_t)rows * rowbytes + (int64_t)sl * piece + gl * 16;
    uint8_t *d = dst + f * (int64_t)rows * rowbytes + (int64_t)sl * piece + gl * 16;
    int r = r0;
    for (; r + (U - 1) * rpb < rows; r += U * rpb) {
        U4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>((const U4 *)(s + (int64_t)(r + u * rpb) * rowbytes));
#pragma unroll
        for (int u = 0; u < U; u++) st<NT>((U4 *)(d + (int64_t)(r + u * rpb) * rowbytes), v[u]);
    }
    for (; r < rows; r += rpb) st<NT>((U4 *)(d + (int64_t)r * rowbytes), ld<NT>((const U4 *)(s + (int64_t)r * rowbytes)));
}

template <bool NT, int U>
__global__ __launch_bounds__(1024) void read_gs(const U4 *src, uint32_t *sink, uint64_t chunks)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < chunks; i += U * stride) {
        U4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(1024) void fill_gs(U4 *dst, uint64_t chunks)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const U4 v = {1, 2, 3, 4};
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < chunks; i += stride) st<NT>(dst + i, v);
}

template <typename F> static double timeit(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) f();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    return ms / reps;
}

int main()
{
    const int rows = 2040, rowbytes = 1024;
    const int64_t frames = 4096;
    const uint64_t bytes = (uint64_t)frames * rows * rowbytes, chunks = bytes / 16;
    uint8_t *a, *b; uint32_t *sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    auto rep = [&](const char *name, double ms, double moved) { printf("%-58s %7.3f ms  %8.1f GB/s\n", name, ms, moved / ms / 1e6); fflush(stdout); };
    const double rw = 2.0 * bytes, ro = (double)bytes;
    for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy grid-stride NT  U=1 grid=%d x1024", wgs);
        rep(nm, timeit([&] { copy_gs<true, true, 1><<<wgs, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    }
    rep("copy grid-stride plain U=1 grid=4096", timeit([&] { copy_gs<false, false, 1><<<4096, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT-load plain-store U=1", timeit([&] { copy_gs<true, false, 1><<<4096, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride plain-load NT-store U=1", timeit([&] { copy_gs<false, true, 1><<<4096, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT  U=2 grid=4096", timeit([&] { copy_gs<true, true, 2><<<4096, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT  U=4 grid=4096", timeit([&] { copy_gs<true, true, 4><<<4096, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT  U=4 grid=512 x1024", timeit([&] { copy_gs<true, true, 4><<<512, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT  U=8 grid=512 x1024", timeit([&] { copy_gs<true, true, 8><<<512, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride NT  U=4 grid=2048 x256", timeit([&] { copy_gs<true, true, 4><<<2048, 256>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy grid-stride plain U=4 grid=512", timeit([&] { copy_gs<false, false, 4><<<512, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("hipMemcpyDtoD", timeit([&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }), rw);
    rep("read-only  NT U=1 grid=4096", timeit([&] { read_gs<true, 1><<<4096, 1024>>>((const U4 *)a, sink, chunks); }), ro);
    rep("read-only  NT U=4 grid=512", timeit([&] { read_gs<true, 4><<<512, 1024>>>((const U4 *)a, sink, chunks); }), ro);
    rep("read-only  plain U=4 grid=512", timeit([&] { read_gs<false, 4><<<512, 1024>>>((const U4 *)a, sink, chunks); }), ro);
    rep("write-only NT grid=4096", timeit([&] { fill_gs<true><<<4096, 1024>>>((U4 *)b, chunks); }), ro);
    rep("write-only plain grid=4096", timeit([&] { fill_gs<false><<<4096, 1024>>>((U4 *)b, chunks); }), ro);
    for (int piece : {1024, 512, 256, 128}) {
        char nm[128];
        const int ns = rowbytes / piece;
        snprintf(nm, sizeof nm, "copy sliced NT piece=%4d U=1 (wg = frame x slice)", piece);
        rep(nm, timeit([&] { copy_sliced<true, 1><<<(int)(frames * ns), 1024>>>(a, b, rows, rowbytes, piece, ns); }), rw);
        snprintf(nm, sizeof nm, "copy sliced NT piece=%4d U=2", piece);
        rep(nm, timeit([&] { copy_sliced<true, 2><<<(int)(frames * ns), 1024>>>(a, b, rows, rowbytes, piece, ns); }), rw);
        snprintf(nm, sizeof nm, "copy sliced NT piece=%4d U=4", piece);
        rep(nm, timeit([&] { copy_sliced<true, 4><<<(int)(frames * ns), 1024>>>(a, b, rows, rowbytes, piece, ns); }), rw);
    }
    for (int piece : {512, 256}) {
        char nm[128];
        const int ns = rowbytes / piece;
        snprintf(nm, sizeof nm, "copy sliced NT piece=%4d U=1 XCD-mapped", piece);
        rep(nm, timeit([&] { copy_sliced<true, 1, true><<<(int)(frames * ns), 1024>>>(a, b, rows, rowbytes, piece, ns); }), rw);
        snprintf(nm, sizeof nm, "copy sliced NT piece=%4d U=2 XCD-mapped", piece);
        rep(nm, timeit([&] { copy_sliced<true, 2, true><<<(int)(frames * ns), 1024>>>(a, b, rows, rowbytes, piece, ns); }), rw);
    }
    rep("copy grid-stride NT  U=1 grid=16384 (again, clocks warm)", timeit([&] { copy_gs<true, true, 1><<<16384, 1024>>>((const U4 *)a, (U4 *)b, chunks); }), rw);
    rep("copy sliced NT piece= 256 U=1 (again)", timeit([&] { copy_sliced<true, 1><<<(int)(frames * 4), 1024>>>(a, b, rows, rowbytes, 256, 4); }), rw);
    rep("read-only  NT U=1 grid=4096 (again)", timeit([&] { read_gs<true, 1><<<4096, 1024>>>((const U4 *)a, sink, chunks); }), ro);
    rep("write-only plain grid=4096 (again)", timeit([&] { fill_gs<false><<<4096, 1024>>>((U4 *)b, chunks); }), ro);
    return 0;
}
