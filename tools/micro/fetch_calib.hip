// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern of the eikonal solve: 8-byte records
// gathered out of 512-B tiles.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide streaming reads (reports half the
// bytes: 128-B requests tallied at 64 B).  Every kernel below touches a known number of 128-B lines and 64-B halves of a
// buffer much larger than the caches, once, so "KB reported per line touched" tells whether a miss is one 128-B request
// (=> double the counter) or 64-B sectors (=> take it as is):
//   rd_stream16   16 B per lane, contiguous                      bytes known: the whole buffer
//   rd_stream8     8 B per lane, contiguous (one 512-B tile per wave-load)
//   rd_line1       one 8-B record per 128-B line
//   rd_line2same   two 8-B records per line, same 64-B half
//   rd_line2diff   two 8-B records per line, different halves
//   rd_tile1       one 8-B record per 512-B tile, pseudo-random position in the tile
//   wr_stream16 / wr_line1 / wr_line2diff: the same for stores
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE -d out/f -- tools/micro/fetch_calib ; rocprofv3 --pmc WRITE_SIZE -d out/w -- tools/micro/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
__device__ __forceinline__ size_t gid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }

__global__ void rd_stream16(const uint4* p, size_t n, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < n) { uint4 v = p[i]; a = v.x + v.y + v.z + v.w; } if (a == 0x12345679u) *sink = a; }
__global__ void rd_stream8(const uint2* p, size_t n, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < n) { uint2 v = p[i]; a = v.x + v.y; } if (a == 0x12345679u) *sink = a; }
__global__ void rd_line1(const uint2* p, size_t nlines, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < nlines) { uint2 v = p[i * 16 + (i % 7)]; a = v.x + v.y; } if (a == 0x12345679u) *sink = a; }
__global__ void rd_line2same(const uint2* p, size_t nlines, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < nlines) { uint2 v = p[i * 16 + (i % 3)], w = p[i * 16 + 4 + (i % 3)]; a = v.x + v.y + w.x + w.y; } if (a == 0x12345679u) *sink = a; }
__global__ void rd_line2diff(const uint2* p, size_t nlines, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < nlines) { uint2 v = p[i * 16 + (i % 7)], w = p[i * 16 + 8 + (i % 5)]; a = v.x + v.y + w.x + w.y; } if (a == 0x12345679u) *sink = a; }
__global__ void rd_tile1(const uint2* p, size_t ntiles, u64* sink) { size_t i = gid(); unsigned a = 0; if (i < ntiles) { size_t t = (i * 2654435761ull) % ntiles; uint2 v = p[t * 64 + ((t * 40503ull) & 63)]; a = v.x + v.y; } if (a == 0x12345679u) *sink = a; }
__global__ void wr_stream16(uint4* p, size_t n) { size_t i = gid(); if (i < n) p[i] = make_uint4((unsigned)i, 1, 2, 3); }
__global__ void wr_line1(uint2* p, size_t nlines) { size_t i = gid(); if (i < nlines) p[i * 16 + (i % 7)] = make_uint2((unsigned)i, 7); }
__global__ void wr_line2diff(uint2* p, size_t nlines) { size_t i = gid(); if (i < nlines) { p[i * 16 + (i % 7)] = make_uint2((unsigned)i, 7); p[i * 16 + 8 + (i % 5)] = make_uint2((unsigned)i, 9); } }

int main()
{
    const size_t bytes = (size_t)2 << 30;                 // 2 GiB: eight times the Infinity Cache
    void* buf; u64* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 1, bytes));
    const size_t nlines = bytes / 128, ntiles = bytes / 512;
    auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL(rd_stream16, blocks(bytes / 16), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(rd_stream8, blocks(bytes / 8), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(rd_line1, blocks(nlines), dim3(256), 0, 0, (const uint2*)buf, nlines, sink);
    hipLaunchKernelGGL(rd_line2same, blocks(nlines), dim3(256), 0, 0, (const uint2*)buf, nlines, sink);
    hipLaunchKernelGGL(rd_line2diff, blocks(nlines), dim3(256), 0, 0, (const uint2*)buf, nlines, sink);
    hipLaunchKernelGGL(rd_tile1, blocks(ntiles), dim3(256), 0, 0, (const uint2*)buf, ntiles, sink);
    hipLaunchKernelGGL(wr_stream16, blocks(bytes / 16), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    hipLaunchKernelGGL(wr_line1, blocks(nlines), dim3(256), 0, 0, (uint2*)buf, nlines);
    hipLaunchKernelGGL(wr_line2diff, blocks(nlines), dim3(256), 0, 0, (uint2*)buf, nlines);
    CHECK(hipDeviceSynchronize());
    printf("buffer %zu B = %zu lines of 128 B = %zu tiles of 512 B; per kernel: stream = all bytes, line* = %zu lines, tile1 = %zu tiles\n", bytes, nlines, ntiles, nlines, ntiles);
    return 0;
}
