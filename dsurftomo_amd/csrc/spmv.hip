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

// out[r] += sum over the entries of segment r, in storage order (one multiply and one add per entry, no
// contraction).  One wavefront per segment: the lanes fetch 64 consecutive entries at once (coalesced values
// and indices, gathered inputs) and form the products in parallel; the additions stay a serial chain -- the
// reference's order -- fed from the lanes with readlane, two instructions per entry.
__global__ __launch_bounds__(256) void k_spmv_segments(int nseg, const long long* __restrict__ ptr, const float* __restrict__ val,
                                                       const int* __restrict__ idx, const float* __restrict__ in, float* __restrict__ out)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nseg) return;
    const long long a = ptr[r], b = ptr[r + 1];
    if (a == b) return;
    float acc = out[r];
    for (long long k0 = a; k0 < b; k0 += 64) {
        const long long k = k0 + lane;
        const float prod = k < b ? val[k] * in[idx[k]] : 0.0f;
        const int cnt = (int)((b - k0) < 64 ? (b - k0) : 64);
        if (cnt == 64) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prod), i));
        } else {
            for (int i = 0; i < cnt; ++i) acc = acc + __shfl(prod, i);
        }
    }
    if (lane == 0) out[r] = acc;
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
// permuted copies: val_out[i] = rw[perm[i]], idx_out[i] = other[perm[i]] - 1
__global__ void k_gather(long long n, const int* __restrict__ perm, const float* __restrict__ rw, const int* __restrict__ other,
                         float* __restrict__ val_out, int* __restrict__ idx_out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = perm[i];
    val_out[i] = rw[k];
    idx_out[i] = other[k] - 1;
}

}  // namespace dsa

namespace {

#define SP_TRY(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) { (e)->fail(DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); return DSA_ERR_DEVICE; } \
    } while (0)

// One ordering of the matrix (by row or by column) built on the device: a STABLE radix sort of the entry
// numbers by key keeps the storage order inside every segment, which is what the accumulation order needs.
int build_order(Engine* e, long long nar, int nkeys, const int* d_key, const int* d_other, const float* d_rw,
                dsa::DevBuf<long long>& dptr, dsa::DevBuf<float>& dval, dsa::DevBuf<int>& didx,
                dsa::DevBuf<int>& keys_out, dsa::DevBuf<int>& perm_in, dsa::DevBuf<int>& perm_out, dsa::DevBuf<unsigned char>& tmp)
{
    if (e->ensure(dptr, (size_t)nkeys + 1) || e->ensure(dval, std::max<size_t>((size_t)nar, 1)) || e->ensure(didx, std::max<size_t>((size_t)nar, 1)) ||
        e->ensure(keys_out, std::max<size_t>((size_t)nar, 1)) || e->ensure(perm_in, std::max<size_t>((size_t)nar, 1)) || e->ensure(perm_out, std::max<size_t>((size_t)nar, 1))) return e->status;
    SP_TRY(e, hipMemsetAsync(dptr.p, 0, ((size_t)nkeys + 1) * 8, e->stream));
    if (nar > 0) {
        const unsigned blocks = (unsigned)((nar + 255) / 256);
        hipLaunchKernelGGL(dsa::k_iota, dim3(blocks), dim3(256), 0, e->stream, nar, perm_in.p);
        hipLaunchKernelGGL(dsa::k_histogram, dim3(blocks), dim3(256), 0, e->stream, nar, d_key, reinterpret_cast<unsigned long long*>(dptr.p));
        int bits = 1;
        while ((1ll << bits) <= nkeys) ++bits;
        size_t tmp_bytes = 0;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, keys_out.p, perm_in.p, perm_out.p, (int)nar, 0, bits, e->stream));
        if (e->ensure(tmp, tmp_bytes)) return e->status;
        SP_TRY(e, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, d_key, keys_out.p, perm_in.p, perm_out.p, (int)nar, 0, bits, e->stream));
        hipLaunchKernelGGL(dsa::k_gather, dim3(blocks), dim3(256), 0, e->stream, nar, perm_out.p, d_rw, d_other, dval.p, didx.p);
    }
    hipLaunchKernelGGL(dsa::k_scan64, dim3(1), dim3(1024), 0, e->stream, nkeys, dptr.p);
    SP_TRY(e, hipGetLastError());
    return 0;
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
    if (!e->spmv) e->spmv = new SpmvState();
    SpmvState& S = *e->spmv;
    S.m = m; S.n = n; S.nar = nar;
    dsa::DevBuf<float> d_rw;
    dsa::DevBuf<int> d_row, d_col, keys_out, perm_in, perm_out;
    dsa::DevBuf<unsigned char> tmp;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    int rc = 0;
    const size_t nn = std::max<size_t>((size_t)nar, 1);
    if (e->ensure(d_rw, nn) || e->ensure(d_row, nn) || e->ensure(d_col, nn) || e->ensure(S.x, (size_t)n) || e->ensure(S.y, (size_t)m)) rc = e->status;
    if (rc == 0 && nar > 0) {
        if (hipMemcpyAsync(d_rw.p, rw, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d_row.p, row, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d_col.p, col, (size_t)nar * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "spmv_load: upload failed"); rc = DSA_ERR_DEVICE; }
    }
    if (rc == 0) rc = build_order(e, nar, m, d_row.p, d_col.p, d_rw.p, S.rowptr, S.val_r, S.col_r, keys_out, perm_in, perm_out, tmp);
    if (rc == 0) rc = build_order(e, nar, n, d_col.p, d_row.p, d_rw.p, S.colptr, S.val_c, S.row_c, keys_out, perm_in, perm_out, tmp);
    if (rc == 0 && hipStreamSynchronize(e->stream) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "spmv_load: device ordering failed"); rc = DSA_ERR_DEVICE; }
    rel(d_rw); rel(d_row); rel(d_col); rel(keys_out); rel(perm_in); rel(perm_out); rel(tmp);
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
        hipLaunchKernelGGL(dsa::k_spmv_segments, dim3((S.m + 3) / 4), dim3(256), 0, e->stream, S.m, S.rowptr.p, S.val_r.p, S.col_r.p, S.x.p, S.y.p);
        SP_TRY(e, hipMemcpyAsync(y, S.y.p, (size_t)S.m * 4, hipMemcpyDeviceToHost, e->stream));
    } else {
        hipLaunchKernelGGL(dsa::k_spmv_segments, dim3((S.n + 3) / 4), dim3(256), 0, e->stream, S.n, S.colptr.p, S.val_c.p, S.row_c.p, S.y.p, S.x.p);
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
    if (mode == 1) hipLaunchKernelGGL(k_spmv_segments, dim3((S.m + 3) / 4), dim3(256), 0, e->stream, S.m, S.rowptr.p, S.val_r.p, S.col_r.p, d_x, d_y);
    else hipLaunchKernelGGL(k_spmv_segments, dim3((S.n + 3) / 4), dim3(256), 0, e->stream, S.n, S.colptr.p, S.val_c.p, S.row_c.p, d_y, d_x);
}

void release_spmv(SpmvState* s)
{
    if (!s) return;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    rel(s->rowptr); rel(s->colptr); rel(s->val_r); rel(s->val_c); rel(s->x); rel(s->y); rel(s->col_r); rel(s->row_c);
    rel(s->u); rel(s->v); rel(s->h); rel(s->hbar); rel(s->xs); rel(s->localV); rel(s->scal);
    if (s->hu) (void)hipHostFree(s->hu);
    if (s->hv) (void)hipHostFree(s->hv);
    delete s;
}
}  // namespace dsa
