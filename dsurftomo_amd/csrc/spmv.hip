// The matrix-vector products of the inversion step on the device (reference aprod.f90:7-60, called by
// LSMR at lsmrModule.f90:390, :486, :497): y += A x and x += A^T y for the COO matrix that CalSurfG
// produced (plus whatever rows the host appended).
//
// The reference accumulates entry by entry in storage order, in fp32.  To give the same bits, every
// output element is owned by one lane that adds its entries in storage order: a stable sort by row
// (for A x) and by column (for A^T y) is made once per matrix on the host, the permuted value / index
// arrays live in HBM, and a product is one pass over them.  HBM-bound: 8 bytes per entry and product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"

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

struct SpmvState {
    int m = 0, n = 0;
    long long nar = 0;
    DevBuf<long long> rowptr, colptr;
    DevBuf<float> val_r, val_c, x, y;
    DevBuf<int> col_r, row_c;
};

}  // namespace dsa

using dsa::Engine;
using dsa::SpmvState;

namespace {

#define SP_TRY(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) { (e)->fail(DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); return DSA_ERR_DEVICE; } \
    } while (0)

// stable counting sort of entry indices by key (1-based keys in [1, nkeys])
void stable_order(long long nar, const int* key, int nkeys, std::vector<long long>& ptr, std::vector<long long>& order)
{
    ptr.assign((size_t)nkeys + 1, 0);
    for (long long k = 0; k < nar; ++k) ptr[(size_t)key[k]] += 1;          // key is 1-based: counts land at [key]
    for (int r = 0; r < nkeys; ++r) ptr[(size_t)r + 1] += ptr[(size_t)r];   // ptr[r] = first entry of 0-based segment r
    order.resize((size_t)nar);
    std::vector<long long> next(ptr.begin(), ptr.end() - 1);
    for (long long k = 0; k < nar; ++k) order[(size_t)next[(size_t)key[k] - 1]++] = k;
}

}  // namespace

extern "C" {

int dsa_spmv_load(dsa_engine* h, int m, int n, long long nar, const float* rw, const int* row, const int* col)
{
    if (!h) return DSA_ERR_ARGUMENT;
    Engine* e = reinterpret_cast<Engine*>(h);
    if (m < 1 || n < 1 || nar < 0 || (nar > 0 && (!rw || !row || !col))) { e->fail(DSA_ERR_ARGUMENT, "spmv_load: bad arguments"); return DSA_ERR_ARGUMENT; }
    for (long long k = 0; k < nar; ++k)
        if (row[k] < 1 || row[k] > m || col[k] < 1 || col[k] > n) { e->fail(DSA_ERR_ARGUMENT, "spmv_load: entry %lld has index (%d, %d) outside %d x %d", k, row[k], col[k], m, n); return DSA_ERR_ARGUMENT; }
    SP_TRY(e, hipSetDevice(e->device));
    if (!e->spmv) e->spmv = new SpmvState();
    SpmvState& S = *e->spmv;
    S.m = m; S.n = n; S.nar = nar;
    std::vector<long long> ptr, order;
    std::vector<float> v((size_t)nar);
    std::vector<int> ix((size_t)nar);
    auto upload = [&](const int* key, int nkeys, const int* other, dsa::DevBuf<long long>& dptr, dsa::DevBuf<float>& dval, dsa::DevBuf<int>& didx) -> int {
        stable_order(nar, key, nkeys, ptr, order);
        for (long long k = 0; k < nar; ++k) { v[(size_t)k] = rw[order[(size_t)k]]; ix[(size_t)k] = other[order[(size_t)k]] - 1; }
        if (e->ensure(dptr, (size_t)nkeys + 1) || e->ensure(dval, std::max<size_t>((size_t)nar, 1)) || e->ensure(didx, std::max<size_t>((size_t)nar, 1))) return e->status;
        SP_TRY(e, hipMemcpy(dptr.p, ptr.data(), ((size_t)nkeys + 1) * 8, hipMemcpyHostToDevice));
        if (nar) {
            SP_TRY(e, hipMemcpy(dval.p, v.data(), (size_t)nar * 4, hipMemcpyHostToDevice));
            SP_TRY(e, hipMemcpy(didx.p, ix.data(), (size_t)nar * 4, hipMemcpyHostToDevice));
        }
        return 0;
    };
    int rc;
    if ((rc = upload(row, m, col, S.rowptr, S.val_r, S.col_r)) != 0) return rc;
    if ((rc = upload(col, n, row, S.colptr, S.val_c, S.row_c)) != 0) return rc;
    if (e->ensure(S.x, (size_t)n) || e->ensure(S.y, (size_t)m)) return e->status;
    return 0;
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
void release_spmv(SpmvState* s)
{
    if (!s) return;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    rel(s->rowptr); rel(s->colptr); rel(s->val_r); rel(s->val_c); rel(s->x); rel(s->y); rel(s->col_r); rel(s->row_c);
    delete s;
}
}  // namespace dsa
