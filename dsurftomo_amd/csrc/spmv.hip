// The matrix-vector products of the inversion step on the device (reference aprod.f90:7-60, called by
// LSMR at lsmrModule.f90:390, :486, :497): y += A x and x += A^T y for the COO matrix that CalSurfG
// produced (plus whatever rows the host appended).
//
// The reference accumulates entry by entry in storage order, in fp32.  To give the same bits, every
// output element is owned by one lane that adds its entries in storage order: a stable sort by row
// (for A x) and by column (for A^T y) is made once per matrix on the host, the permuted value / index
// arrays live in HBM, and a product is one pass over them.  HBM-bound: 8 bytes per entry and product.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"
#include "spmv_state.h"

namespace dsa {

// out[seg] += sum over the entries of the segment, in storage order (one multiply and one add per entry, no contraction).
// One LANE per segment: the reference's accumulation is a serial chain per output element, so a wavefront runs 64 chains
// side by side.  The storage is transposed per slice of 64 segments (spmv_state.h), which makes the value / index loads of a
// step one coalesced 256-B read each; the input vector (0.5-1 MB) is gathered from L2.  HBM-bound: 8 bytes per entry.
// The loads of UNROLL steps are issued before the first addition needs them.  Measured on the headline matrix (190 M entries, 128 k
// rows x 133 k columns, profiles/r02_spmv_trace.txt): 0.87 ms per product = 1.75 TB/s of the 8 B per entry; what binds it is not HBM
// but the gathers of the input vector: 64 lanes = 64 unrelated rows, so every 4-byte operand pulls its own 128-B line out of L2
// (UNROLL = 32 is slower, 1.11 ms: more lines in flight, same L2 line rate).  Next step: input vector staged in LDS per column block.
template <bool ABS>
__global__ __launch_bounds__(256) void k_spmv_sliced(int nslices, const long long* __restrict__ off, const int* __restrict__ seg, const int* __restrict__ len,
                                                     const float* __restrict__ val, const int* __restrict__ idx, const float* __restrict__ in, float* __restrict__ out)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= nslices) return;
    const int mylen = len[(size_t)j * 64 + lane];
    const int s = seg[(size_t)j * 64 + lane];
    if (mylen == 0) return;                       // (padding lanes of the last slice and empty segments; sorted: a suffix of the wave)
    const float* __restrict__ v = val + off[j] + lane;
    const int* __restrict__ ix = idx + off[j] + lane;
    float acc = out[s];
    constexpr int UNROLL = 8;
    int k = 0;
    for (; k + UNROLL <= mylen; k += UNROLL) {
        float pv[UNROLL]; int pi[UNROLL]; float pin[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { pv[u] = v[(size_t)(k + u) * 64]; pi[u] = ix[(size_t)(k + u) * 64]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) pin[u] = ABS ? 1.0f : in[pi[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = acc + (ABS ? fabsf(pv[u]) : pv[u] * pin[u]);
    }
    for (; k + 4 <= mylen; k += 4) {
        float pv[4]; int pi[4]; float pin[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pv[u] = v[(size_t)(k + u) * 64]; pi[u] = ix[(size_t)(k + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) pin[u] = ABS ? 1.0f : in[pi[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = acc + (ABS ? fabsf(pv[u]) : pv[u] * pin[u]);
    }
    for (; k < mylen; ++k) { const float a = v[(size_t)k * 64]; acc = acc + (ABS ? fabsf(a) : a * in[ix[(size_t)k * 64]]); }
    out[s] = acc;
}

}  // namespace dsa

using dsa::Engine;
using dsa::SpmvState;

namespace dsa {

__global__ void k_iota(long long n, int* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)i;
}
// counts[key - 1] += 1 (1-based keys); the totals do not depend on the order of the atomics
__global__ void k_histogram(long long n, const int* __restrict__ key, unsigned long long* __restrict__ counts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[key[i] - 1], 1ull);
}
// exclusive scan of nseg counts in place into ptr[0..nseg] (one workgroup; nseg is a row or column count)
__global__ __launch_bounds__(1024) void k_scan64(int nseg, long long* __restrict__ ptr)
{
    __shared__ long long s_sum[1024];
    const int tid = threadIdx.x;
    const int per = (nseg + 1023) / 1024;
    const int lo = min(tid * per, nseg), hi = min(lo + per, nseg);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += ptr[i];
    s_sum[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = tid >= d ? s_sum[tid - d] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    long long run = s_sum[tid] - s;
    for (int i = lo; i < hi; ++i) { const long long c = ptr[i]; ptr[i] = run; run += c; }
    if (tid == 1023) ptr[nseg] = s_sum[1023];
}
// sort keys of the segments: ~length (so that an ascending sort puts the longest first), value = segment; slots beyond the
// last segment are empty padding (length 0, segment 0) that sorts to the end
__global__ void k_segment_keys(int nseg, int nslots, const unsigned long long* __restrict__ counts, int* __restrict__ key, int* __restrict__ seg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const int len = i < nseg ? (int)counts[i] : 0;
    key[i] = ~len & 0x7fffffff;
    seg[i] = i < nseg ? i : 0;
}
// per (slice, lane): segment and length; per slice: 64 x the longest length (the first lane's), scanned into offsets afterwards
__global__ void k_slice_table(int nslices, const int* __restrict__ key_sorted, const int* __restrict__ seg_sorted, int* __restrict__ len, int* __restrict__ seg,
                              long long* __restrict__ off)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslices * 64) return;
    const int l = ~key_sorted[i] & 0x7fffffff;
    len[i] = l;
    seg[i] = seg_sorted[i];
    if ((i & 63) == 0) off[i >> 6] = 64ll * (long long)l;
    if (i == 0) off[nslices] = 0;
}
// transposed copy: entry k of the segment of (slice j, lane l) = entry ptr[seg] + k of the key-sorted matrix
__global__ __launch_bounds__(256) void k_fill_slices(int nslices, const long long* __restrict__ off, const int* __restrict__ seg, const int* __restrict__ len,
                                                     const long long* __restrict__ ptr, const int* __restrict__ perm, const float* __restrict__ rw,
                                                     const int* __restrict__ other, float* __restrict__ val, int* __restrict__ idx)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= nslices) return;
    const int mylen = len[(size_t)j * 64 + lane];
    const long long first = ptr[seg[(size_t)j * 64 + lane]];
    const long long base = off[j] + lane;
    for (int k = 0; k < mylen; ++k) {
        const int e = perm[first + k];
        val[base + (long long)k * 64] = rw[e];
        idx[base + (long long)k * 64] = other[e] - 1;
    }
}

}  // namespace dsa

namespace {

#define SP_TRY(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) { (e)->fail(DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); return DSA_ERR_DEVICE; } \
    } while (0)

// One ordering of the matrix (by row or by column) built on the device.  A STABLE radix sort of the entry numbers by key
// keeps the storage order inside every segment (what the accumulation order needs); the segments are then sorted by
// length (longest first, so the lanes of a slice finish together and the long slices start first) and laid out in
// slices of 64.
int build_order(Engine* e, long long nar, int nkeys, const int* d_key, const int* d_other, const float* d_rw, dsa::SpmvState::Sliced& S,
                dsa::DevBuf<int>& keys_out, dsa::DevBuf<int>& perm_in, dsa::DevBuf<int>& perm_out, dsa::DevBuf<unsigned char>& tmp)
{
    const int nslices = (nkeys + 63) / 64;
    dsa::DevBuf<long long> ptr;
    dsa::DevBuf<int> lens, lens_sorted, segs, segs_sorted;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    auto done = [&](int rc) { rel(ptr); rel(lens); rel(lens_sorted); rel(segs); rel(segs_sorted); return rc; };
    const size_t nn = std::max<size_t>((size_t)nar, 1), ns = (size_t)nslices * 64;
    if (e->ensure(ptr, (size_t)nkeys + 1) || e->ensure(keys_out, nn) || e->ensure(perm_in, nn) || e->ensure(perm_out, nn) ||
        e->ensure(lens, ns) || e->ensure(lens_sorted, ns) || e->ensure(segs, ns) || e->ensure(segs_sorted, ns) ||
        e->ensure(S.off, (size_t)nslices + 1) || e->ensure(S.seg, ns) || e->ensure(S.len, ns)) return done(e->status);
    S.nslices = nslices;
    SP_TRY(e, hipMemsetAsync(ptr.p, 0, ((size_t)nkeys + 1) * 8, e->stream));
    const unsigned blocks = (unsigned)((nn + 255) / 256);
    if (nar > 0) {
        hipLaunchKernelGGL(dsa::k_iota, dim3(blocks), dim3(256), 0, e->stream, nar, perm_in.p);
        hipLaunchKernelGGL(dsa::k_histogram, dim3(blocks), dim3(256), 0, e->stream, nar, d_key, reinterpret_cast<unsigned long long*>(ptr.p));
        int bits = 1;
        while ((1ll << bits) <= nkeys) ++bits;
        size_t tmp_bytes = 0;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, keys_out.p, perm_in.p, perm_out.p, (int)nar, 0, bits, e->stream));
        if (e->ensure(tmp, tmp_bytes)) return done(e->status);
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, d_key, keys_out.p, perm_in.p, perm_out.p, (int)nar, 0, bits, e->stream));
    }
    // segment lengths (descending order = ascending order of ~len), padded with empty segments up to a whole slice
    hipLaunchKernelGGL(dsa::k_segment_keys, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, e->stream, nkeys, (int)ns, reinterpret_cast<const unsigned long long*>(ptr.p), lens.p, segs.p);
    {
        size_t tmp_bytes = 0;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, lens.p, lens_sorted.p, segs.p, segs_sorted.p, (int)ns, 0, 32, e->stream));
        if (e->ensure(tmp, tmp_bytes)) return done(e->status);
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, lens.p, lens_sorted.p, segs.p, segs_sorted.p, (int)ns, 0, 32, e->stream));
    }
    hipLaunchKernelGGL(dsa::k_scan64, dim3(1), dim3(1024), 0, e->stream, nkeys, ptr.p);            // counts -> first entry of every segment
    hipLaunchKernelGGL(dsa::k_slice_table, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, e->stream, nslices, lens_sorted.p, segs_sorted.p, S.len.p, S.seg.p, S.off.p);
    hipLaunchKernelGGL(dsa::k_scan64, dim3(1), dim3(1024), 0, e->stream, nslices, S.off.p);        // slice sizes -> slice offsets
    long long padded = 0;
    SP_TRY(e, hipMemcpyAsync(&padded, S.off.p + nslices, 8, hipMemcpyDeviceToHost, e->stream));
    SP_TRY(e, hipStreamSynchronize(e->stream));
    S.padded = padded;
    if (e->ensure(S.val, std::max<size_t>((size_t)padded, 1)) || e->ensure(S.idx, std::max<size_t>((size_t)padded, 1))) return done(e->status);
    if (nar > 0)
        hipLaunchKernelGGL(dsa::k_fill_slices, dim3((unsigned)((nslices + 3) / 4)), dim3(256), 0, e->stream, nslices, S.off.p, S.seg.p, S.len.p, ptr.p, perm_out.p, d_rw, d_other, S.val.p, S.idx.p);
    SP_TRY(e, hipGetLastError());
    SP_TRY(e, hipStreamSynchronize(e->stream));
    return done(0);
}

// both orderings of a COO matrix that is already on the device (1-based row / col)
int load_from_device(Engine* e, int m, int n, long long nar, const float* d_rw, const int* d_row, const int* d_col)
{
    if (!e->spmv) e->spmv = new SpmvState();
    SpmvState& S = *e->spmv;
    S.m = m; S.n = n; S.nar = nar;
    dsa::DevBuf<int> keys_out, perm_in, perm_out;
    dsa::DevBuf<unsigned char> tmp;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    int rc = 0;
    if (e->ensure(S.x, (size_t)n) || e->ensure(S.y, (size_t)m)) rc = e->status;
    if (rc == 0) rc = build_order(e, nar, m, d_row, d_col, d_rw, S.by_row, keys_out, perm_in, perm_out, tmp);
    if (rc == 0) rc = build_order(e, nar, n, d_col, d_row, d_rw, S.by_col, keys_out, perm_in, perm_out, tmp);
    rel(keys_out); rel(perm_in); rel(perm_out); rel(tmp);
    return rc;
}

}  // namespace

extern "C" {

int dsa_spmv_load(dsa_engine* h, int m, int n, long long nar, const float* rw, const int* row, const int* col)
{
    if (!h) return DSA_ERR_ARGUMENT;
    Engine* e = reinterpret_cast<Engine*>(h);
    if (m < 1 || n < 1 || nar < 0 || nar > 0x7fffffffll || (nar > 0 && (!rw || !row || !col))) { e->fail(DSA_ERR_ARGUMENT, "spmv_load: bad arguments"); return DSA_ERR_ARGUMENT; }
    for (long long k = 0; k < nar; ++k)
        if (row[k] < 1 || row[k] > m || col[k] < 1 || col[k] > n) { e->fail(DSA_ERR_ARGUMENT, "spmv_load: entry %lld has index (%d, %d) outside %d x %d", k, row[k], col[k], m, n); return DSA_ERR_ARGUMENT; }
    SP_TRY(e, hipSetDevice(e->device));
    dsa::DevBuf<float> d_rw;
    dsa::DevBuf<int> d_row, d_col;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    int rc = 0;
    const size_t nn = std::max<size_t>((size_t)nar, 1);
    if (e->ensure(d_rw, nn) || e->ensure(d_row, nn) || e->ensure(d_col, nn)) rc = e->status;
    if (rc == 0 && nar > 0) {
        if (hipMemcpyAsync(d_rw.p, rw, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d_row.p, row, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d_col.p, col, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "spmv_load: upload failed"); rc = DSA_ERR_DEVICE; }
    }
    if (rc == 0) rc = load_from_device(e, m, n, nar, d_rw.p, d_row.p, d_col.p);
    rel(d_rw); rel(d_row); rel(d_col);
    return rc;
}

// mode 1: y += A x (x: n in, y: m in/out); mode 2: x += A^T y (y: m in, x: n in/out); host vectors
int dsa_spmv(dsa_engine* h, int mode, float* x, float* y)
{
    if (!h) return DSA_ERR_ARGUMENT;
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e->spmv) { e->fail(DSA_ERR_STATE, "spmv: call dsa_spmv_load first"); return DSA_ERR_STATE; }
    if ((mode != 1 && mode != 2) || !x || !y) { e->fail(DSA_ERR_ARGUMENT, "spmv: bad arguments"); return DSA_ERR_ARGUMENT; }
    SpmvState& S = *e->spmv;
    SP_TRY(e, hipSetDevice(e->device));
    SP_TRY(e, hipMemcpyAsync(S.x.p, x, (size_t)S.n * 4, hipMemcpyHostToDevice, e->stream));
    SP_TRY(e, hipMemcpyAsync(S.y.p, y, (size_t)S.m * 4, hipMemcpyHostToDevice, e->stream));
    if (mode == 1) {
        dsa::spmv_device(e, 1, S.x.p, S.y.p);
        SP_TRY(e, hipMemcpyAsync(y, S.y.p, (size_t)S.m * 4, hipMemcpyDeviceToHost, e->stream));
    } else {
        dsa::spmv_device(e, 2, S.x.p, S.y.p);
        SP_TRY(e, hipMemcpyAsync(x, S.x.p, (size_t)S.n * 4, hipMemcpyDeviceToHost, e->stream));
    }
    SP_TRY(e, hipGetLastError());
    SP_TRY(e, hipStreamSynchronize(e->stream));
    return 0;
}

}  // extern "C"

namespace dsa {
void spmv_device(Engine* e, int mode, float* d_x, float* d_y)
{
    SpmvState& S = *e->spmv;
    const SpmvState::Sliced& L = mode == 1 ? S.by_row : S.by_col;
    if (L.nslices <= 0) return;
    hipLaunchKernelGGL(k_spmv_sliced<false>, dim3((unsigned)((L.nslices + 3) / 4)), dim3(256), 0, e->stream, L.nslices, L.off.p, L.seg.p, L.len.p, L.val.p, L.idx.p,
                       mode == 1 ? d_x : d_y, mode == 1 ? d_y : d_x);
}

// out[c] += sum over the entries of column c of |value|, in storage order (the DWS of main.f90:378-385)
// d_len: entries to take per (slice, lane) -- a prefix of every column
void spmv_abs_column_sums(Engine* e, const int* d_len, float* d_out)
{
    const SpmvState::Sliced& L = e->spmv->by_col;
    if (L.nslices <= 0) return;
    hipLaunchKernelGGL(k_spmv_sliced<true>, dim3((unsigned)((L.nslices + 3) / 4)), dim3(256), 0, e->stream, L.nslices, L.off.p, L.seg.p, d_len, L.val.p, L.idx.p,
                       (const float*)nullptr, d_out);
}

int spmv_load_from_device(Engine* e, int m, int n, long long nar, const float* d_rw, const int* d_row, const int* d_col)
{
    return load_from_device(e, m, n, nar, d_rw, d_row, d_col);
}

void release_spmv(SpmvState* s)
{
    if (!s) return;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    for (SpmvState::Sliced* L : { &s->by_row, &s->by_col }) { rel(L->off); rel(L->seg); rel(L->len); rel(L->val); rel(L->idx); }
    rel(s->x); rel(s->y);
    rel(s->u); rel(s->v); rel(s->h); rel(s->hbar); rel(s->xs); rel(s->localV); rel(s->scal);
    if (s->hu) (void)hipHostFree(s->hu);
    if (s->hv) (void)hipHostFree(s->hv);
    delete s;
}
}  // namespace dsa
