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
        for (int u = 0; u < UNROLL; ++u) { pv[u] = v[(size_t)(k + u) * 64]; pi[u] = ABS ? 0 : ix[(size_t)(k + u) * 64]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) pin[u] = ABS ? 1.0f : in[pi[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = acc + (ABS ? fabsf(pv[u]) : pv[u] * pin[u]);
    }
    for (; k + 4 <= mylen; k += 4) {
        float pv[4]; int pi[4]; float pin[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pv[u] = v[(size_t)(k + u) * 64]; pi[u] = ABS ? 0 : ix[(size_t)(k + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) pin[u] = ABS ? 1.0f : in[pi[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = acc + (ABS ? fabsf(pv[u]) : pv[u] * pin[u]);
    }
    for (; k < mylen; ++k) { const float a = v[(size_t)k * 64]; acc = acc + (ABS ? fabsf(a) : a * in[ix[(size_t)k * 64]]); }
    out[s] = acc;
}

// The same chain with the input vector's block staged in LDS (dynamic shared memory, `nin` floats): 512 threads = 8 slices per
// workgroup share one copy, the gathers become LDS reads.  idx holds block-local indices.
constexpr int kSpmvBlock = 32768;            // floats of input vector per block: 128 KB of the CU's 160 KB LDS
template <int NT>
__global__ __launch_bounds__(NT) void k_spmv_block(int nslices, const long long* __restrict__ off, const int* __restrict__ seg, const int* __restrict__ len,
                                                   const float* __restrict__ val, const unsigned short* __restrict__ idx, const float* __restrict__ in, int nin,
                                                   float* __restrict__ out)
{
    constexpr bool ABS = false;
    extern __shared__ float xs[];
    if (!ABS) {
        const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in);       // (block starts are multiples of 32768 floats: aligned)
        float4* xs4 = reinterpret_cast<float4*>(xs);
        const int n4 = nin >> 2;
        for (int i = threadIdx.x; i < n4; i += NT) xs4[i] = in4[i];
        for (int i = (n4 << 2) + threadIdx.x; i < nin; i += NT) xs[i] = in[i];
        __syncthreads();
    }
    // slices are sorted by length: deal them round-robin so that every workgroup (= every CU: about one workgroup per CU) streams the same share of the bytes
    const int j = blockIdx.x + (int)gridDim.x * (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= nslices) return;
    const int mylen = len[(size_t)j * 64 + lane];
    if (mylen == 0) return;
    const int s = seg[(size_t)j * 64 + lane];
    const float* __restrict__ v = val + off[j] + lane;
    const unsigned short* __restrict__ ix = idx + off[j] + lane;
    float acc = out[s];
    // A block launch lasts as long as its longest slice (a ray that runs along a column block has several times the average
    // count), and a slice is a chain of load batches: two batches of UNROLL steps are kept in flight (the loads of batch
    // t + 1 are issued before batch t is added up).
    constexpr int UNROLL = 16;
    int k = 0;
    float pv[UNROLL]; int pi[UNROLL];
    const bool first = UNROLL <= mylen;
    if (first) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { pv[u] = v[(size_t)u * 64]; pi[u] = (int)ix[(size_t)u * 64]; }
    }
    for (; k + UNROLL <= mylen; k += UNROLL) {
        float nv[UNROLL]; int ni[UNROLL];
        const bool more = k + 2 * UNROLL <= mylen;
        if (more) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { nv[u] = v[(size_t)(k + UNROLL + u) * 64]; ni[u] = (int)ix[(size_t)(k + UNROLL + u) * 64]; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = acc + pv[u] * xs[pi[u]];
        if (more) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { pv[u] = nv[u]; pi[u] = ni[u]; }
        }
    }
    for (; k < mylen; ++k) { const float a = v[(size_t)k * 64]; acc = acc + (ABS ? fabsf(a) : a * xs[ix[(size_t)k * 64]]); }
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
// mono[seg] = 1 unless the segment's entries are stored in ascending (non-decreasing) input order; keys: sorted 1-based segment per position
__global__ void k_not_monotone(long long n, const int* __restrict__ keys, const int* __restrict__ perm, const int* __restrict__ other, int* __restrict__ flag)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 1 || p >= n) return;
    if (keys[p] == keys[p - 1] && other[perm[p]] < other[perm[p - 1]]) flag[keys[p] - 1] = 1;
}
// block of every position: input index / block size, or nblocks for the segments that are not in ascending order; plus the
// per (block, segment) counts (all entries, and the data entries = original entry number below nar_data)
__global__ void k_block_ids(long long n, const int* __restrict__ keys, const int* __restrict__ perm, const int* __restrict__ other, const int* __restrict__ flag,
                            int block, int nblocks, int nseg, long long nar_data, int* __restrict__ blk, unsigned long long* __restrict__ cnt,
                            unsigned long long* __restrict__ cnt_data)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int sg = keys[p] - 1, e = perm[p];
    const int b = flag[sg] ? nblocks : (other[e] - 1) / block;
    blk[p] = b;
    atomicAdd(&cnt[(size_t)b * nseg + sg], 1ull);
    if (cnt_data && e < nar_data) atomicAdd(&cnt_data[(size_t)b * nseg + sg], 1ull);
}
// transposed copy of one block: entry k of the segment of (slice j, lane l) = position pos2[start + k] of the key-sorted matrix
__global__ __launch_bounds__(256) void k_fill_block(int nslices, const long long* __restrict__ off, const int* __restrict__ seg, const int* __restrict__ len,
                                                    const long long* __restrict__ start, const int* __restrict__ pos2, const int* __restrict__ perm,
                                                    const float* __restrict__ rw, const int* __restrict__ other, int index_base, float* __restrict__ val,
                                                    int* __restrict__ idx, unsigned short* __restrict__ idx16)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= nslices) return;
    const int mylen = len[(size_t)j * 64 + lane];
    const long long first = start[seg[(size_t)j * 64 + lane]];
    const long long base = off[j] + lane;
    for (int k = 0; k < mylen; ++k) {
        const int e = perm[pos2[first + k]];
        val[base + (long long)k * 64] = rw[e];
        if (idx16) idx16[base + (long long)k * 64] = (unsigned short)(other[e] - 1 - index_base);
        else idx[base + (long long)k * 64] = other[e] - 1 - index_base;
    }
}
__global__ void k_slot_lengths(int nslots, const int* __restrict__ seg, const int* __restrict__ len, const unsigned long long* __restrict__ count, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) out[i] = len[i] > 0 ? (int)count[seg[i]] : 0;
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
}  // namespace dsa

namespace {

#define SP_TRY(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) { (e)->fail(DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); return DSA_ERR_DEVICE; } \
    } while (0)

// One orientation of the matrix (by row or by column) built on the device.  A STABLE radix sort of the entry numbers by
// segment keeps the storage order inside every segment (what the accumulation order needs); a second stable sort by block of the
// input vector groups every segment's entries per block, still in storage order; per block the segments are sorted by
// length (longest first, so the lanes of a slice finish together) and laid out in slices of 64 (spmv_state.h).
int build_order(Engine* e, long long nar, int nkeys, int ninput, const int* d_key, const int* d_other, const float* d_rw, long long nar_data,
                dsa::SpmvState::Ordering& O)
{
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    for (auto& S : O.blocks) { rel(S.off); rel(S.seg); rel(S.len); rel(S.val); rel(S.idx); rel(S.idx16); }
    for (auto& d : O.data_len) rel(d);
    O.block = dsa::kSpmvBlock; O.ninput = ninput;
    O.nblocks = (ninput + O.block - 1) / O.block;
    const int nb1 = O.nblocks + 1;
    O.blocks.assign(nb1, dsa::SpmvState::Sliced());
    O.data_len.assign(nar_data >= 0 ? nb1 : 0, dsa::DevBuf<int>());
    const int nslices = (nkeys + 63) / 64;
    const size_t nn = std::max<size_t>((size_t)nar, 1), ns = (size_t)nslices * 64, ncnt = (size_t)nb1 * nkeys;
    dsa::DevBuf<int> keys_out, perm_in, perm_out, flag, blk, blk_sorted, pos2, lens, lens_sorted, segs, segs_sorted;
    dsa::DevBuf<long long> cnt, cnt_data, start;
    dsa::DevBuf<unsigned char> tmp;
    auto done = [&](int rc) { rel(keys_out); rel(perm_in); rel(perm_out); rel(flag); rel(blk); rel(blk_sorted); rel(pos2); rel(lens); rel(lens_sorted); rel(segs);
                              rel(segs_sorted); rel(cnt); rel(cnt_data); rel(start); rel(tmp); return rc; };
    if (e->ensure(keys_out, nn) || e->ensure(perm_in, nn) || e->ensure(perm_out, nn) || e->ensure(flag, (size_t)nkeys) || e->ensure(blk, nn) ||
        e->ensure(blk_sorted, nn) || e->ensure(pos2, nn) || e->ensure(lens, ns) || e->ensure(lens_sorted, ns) || e->ensure(segs, ns) || e->ensure(segs_sorted, ns) ||
        e->ensure(cnt, ncnt + 1) || e->ensure(start, ncnt + 1) || (nar_data >= 0 && e->ensure(cnt_data, ncnt + 1))) return done(e->status);
    SP_TRY(e, hipMemsetAsync(flag.p, 0, (size_t)nkeys * 4, e->stream));
    SP_TRY(e, hipMemsetAsync(cnt.p, 0, (ncnt + 1) * 8, e->stream));
    if (nar_data >= 0) SP_TRY(e, hipMemsetAsync(cnt_data.p, 0, (ncnt + 1) * 8, e->stream));
    const unsigned gb = (unsigned)((nn + 255) / 256);
    auto sort_pairs = [&](const int* kin, int* kout, const int* vin, int* vout, size_t n, int bits) -> int {
        size_t tmp_bytes = 0;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kin, kout, vin, vout, (int)n, 0, bits, e->stream));
        if (e->ensure(tmp, tmp_bytes)) return e->status;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, kin, kout, vin, vout, (int)n, 0, bits, e->stream));
        return 0;
    };
    if (nar > 0) {
        hipLaunchKernelGGL(dsa::k_iota, dim3(gb), dim3(256), 0, e->stream, nar, perm_in.p);
        int bits = 1;
        while ((1ll << bits) <= nkeys) ++bits;
        if (sort_pairs(d_key, keys_out.p, perm_in.p, perm_out.p, (size_t)nar, bits)) return done(e->status);
        hipLaunchKernelGGL(dsa::k_not_monotone, dim3(gb), dim3(256), 0, e->stream, nar, keys_out.p, perm_out.p, d_other, flag.p);
        hipLaunchKernelGGL(dsa::k_block_ids, dim3(gb), dim3(256), 0, e->stream, nar, keys_out.p, perm_out.p, d_other, flag.p, O.block, O.nblocks, nkeys, nar_data,
                           blk.p, reinterpret_cast<unsigned long long*>(cnt.p), nar_data >= 0 ? reinterpret_cast<unsigned long long*>(cnt_data.p) : nullptr);
        int bbits = 1;
        while ((1 << bbits) <= O.nblocks) ++bbits;
        if (sort_pairs(blk.p, blk_sorted.p, perm_in.p, pos2.p, (size_t)nar, bbits)) return done(e->status);     // perm_in is still 0, 1, 2, ...: positions
    }
    SP_TRY(e, hipMemcpyAsync(start.p, cnt.p, (ncnt + 1) * 8, hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(dsa::k_scan64, dim3(1), dim3(1024), 0, e->stream, (int)ncnt, start.p);       // first position (in pos2) of every (block, segment)
    for (int b = 0; b < nb1; ++b) {
        dsa::SpmvState::Sliced& S = O.blocks[b];
        if (e->ensure(S.off, (size_t)nslices + 1) || e->ensure(S.seg, ns) || e->ensure(S.len, ns)) return done(e->status);
        S.nslices = nslices;
        hipLaunchKernelGGL(dsa::k_segment_keys, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, e->stream, nkeys, (int)ns,
                           reinterpret_cast<const unsigned long long*>(cnt.p + (size_t)b * nkeys), lens.p, segs.p);
        if (sort_pairs(lens.p, lens_sorted.p, segs.p, segs_sorted.p, ns, 32)) return done(e->status);
        hipLaunchKernelGGL(dsa::k_slice_table, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, e->stream, nslices, lens_sorted.p, segs_sorted.p, S.len.p, S.seg.p, S.off.p);
        hipLaunchKernelGGL(dsa::k_scan64, dim3(1), dim3(1024), 0, e->stream, nslices, S.off.p);        // slice sizes -> slice offsets
        long long padded = 0;
        SP_TRY(e, hipMemcpyAsync(&padded, S.off.p + nslices, 8, hipMemcpyDeviceToHost, e->stream));
        SP_TRY(e, hipStreamSynchronize(e->stream));
        S.padded = padded;
        // only the slices that hold entries are launched later (the segments are sorted by length)
        const bool local = b < O.nblocks;       // block-local 16-bit indices; the unblocked rest keeps 32-bit global ones
        if (e->ensure(S.val, std::max<size_t>((size_t)padded, 1)) || (local ? e->ensure(S.idx16, std::max<size_t>((size_t)padded, 1)) : e->ensure(S.idx, std::max<size_t>((size_t)padded, 1)))) return done(e->status);
        if (padded > 0)
            hipLaunchKernelGGL(dsa::k_fill_block, dim3((unsigned)((nslices + 3) / 4)), dim3(256), 0, e->stream, nslices, S.off.p, S.seg.p, S.len.p,
                               start.p + (size_t)b * nkeys, pos2.p, perm_out.p, d_rw, d_other, local ? b * O.block : 0, S.val.p, local ? nullptr : S.idx.p,
                               local ? S.idx16.p : nullptr);
        if (nar_data >= 0) {
            if (e->ensure(O.data_len[b], ns)) return done(e->status);
            hipLaunchKernelGGL(dsa::k_slot_lengths, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, e->stream, (int)ns, S.seg.p, S.len.p,
                               reinterpret_cast<const unsigned long long*>(cnt_data.p + (size_t)b * nkeys), O.data_len[b].p);
        }
    }
    SP_TRY(e, hipGetLastError());
    SP_TRY(e, hipStreamSynchronize(e->stream));
    return done(0);
}

// both orderings of a COO matrix that is already on the device (1-based row / col); nar_data >= 0: the first nar_data entries are
// data entries (the DWS column sums of main.f90:378-385 run over them only)
int load_from_device(Engine* e, int m, int n, long long nar, const float* d_rw, const int* d_row, const int* d_col, long long nar_data = -1)
{
    if (!e->spmv) e->spmv = new SpmvState();
    SpmvState& S = *e->spmv;
    S.m = m; S.n = n; S.nar = nar;
    int rc = 0;
    if (e->ensure(S.x, (size_t)n) || e->ensure(S.y, (size_t)m)) rc = e->status;
    if (rc == 0) rc = build_order(e, nar, m, n, d_row, d_col, d_rw, -1, S.by_row);
    if (rc == 0) rc = build_order(e, nar, n, m, d_col, d_row, d_rw, nar_data, S.by_col);
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
static void run_ordering(Engine* e, const SpmvState::Ordering& O, bool abs_sums, const float* d_in, float* d_out)
{
    // the attribute is per device (an engine per GPU in one process: DSA_DEVICES): set it for the device this engine runs on, once per engine
    if (!e->spmv_attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv_block<512>), hipFuncAttributeMaxDynamicSharedMemorySize, kSpmvBlock * 4) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv_block<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, kSpmvBlock * 4) != hipSuccess) {
            e->fail(DSA_ERR_DEVICE, "spmv: the device refuses %d bytes of dynamic LDS per workgroup", kSpmvBlock * 4);
            return;
        }
        e->spmv_attr_set = true;
    }
    for (int b = 0; b <= O.nblocks; ++b) {
        const SpmvState::Sliced& L = O.blocks[b];
        if (L.padded <= 0) continue;
        const int* len = abs_sums ? O.data_len[b].p : L.len.p;
        if (abs_sums)
            hipLaunchKernelGGL(k_spmv_sliced<true>, dim3((unsigned)((L.nslices + 3) / 4)), dim3(256), 0, e->stream, L.nslices, L.off.p, L.seg.p, len, L.val.p, (const int*)nullptr,
                               (const float*)nullptr, d_out);
        else if (b < O.nblocks) {
            // one copy of the input block per workgroup: sixteen slices share it when that still gives every CU a workgroup
            const int nin = std::min(O.block, O.ninput - b * O.block);
            if (L.nslices >= 16 * 200)
                hipLaunchKernelGGL(k_spmv_block<1024>, dim3((unsigned)((L.nslices + 15) / 16)), dim3(1024), (size_t)kSpmvBlock * 4, e->stream, L.nslices, L.off.p, L.seg.p, len,
                                   L.val.p, L.idx16.p, d_in + (size_t)b * O.block, nin, d_out);
            else
                hipLaunchKernelGGL(k_spmv_block<512>, dim3((unsigned)((L.nslices + 7) / 8)), dim3(512), (size_t)kSpmvBlock * 4, e->stream, L.nslices, L.off.p, L.seg.p, len,
                                   L.val.p, L.idx16.p, d_in + (size_t)b * O.block, nin, d_out);
        }
        else
            hipLaunchKernelGGL(k_spmv_sliced<false>, dim3((unsigned)((L.nslices + 3) / 4)), dim3(256), 0, e->stream, L.nslices, L.off.p, L.seg.p, len, L.val.p, L.idx.p,
                               d_in, d_out);
    }
}

void spmv_device(Engine* e, int mode, float* d_x, float* d_y)
{
    SpmvState& S = *e->spmv;
    if (mode == 1) run_ordering(e, S.by_row, false, d_x, d_y);
    else run_ordering(e, S.by_col, false, d_y, d_x);
}

// out[c] += sum over the DATA entries of column c of |value|, in storage order (the DWS of main.f90:378-385); needs a matrix
// loaded with nar_data
void spmv_abs_column_sums(Engine* e, float* d_out)
{
    run_ordering(e, e->spmv->by_col, true, nullptr, d_out);
}

int spmv_load_from_device(Engine* e, int m, int n, long long nar, const float* d_rw, const int* d_row, const int* d_col, long long nar_data)
{
    return load_from_device(e, m, n, nar, d_rw, d_row, d_col, nar_data);
}

void release_spmv(SpmvState* s)
{
    if (!s) return;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    for (SpmvState::Ordering* O : { &s->by_row, &s->by_col }) {
        for (auto& L : O->blocks) { rel(L.off); rel(L.seg); rel(L.len); rel(L.val); rel(L.idx); rel(L.idx16); }
        for (auto& d : O->data_len) rel(d);
    }
    rel(s->x); rel(s->y);
    rel(s->u); rel(s->v); rel(s->h); rel(s->hbar); rel(s->xs); rel(s->localV); rel(s->scal);
    if (s->hu) (void)hipHostFree(s->hu);
    if (s->hv) (void)hipHostFree(s->hv);
    delete s;
}
}  // namespace dsa
