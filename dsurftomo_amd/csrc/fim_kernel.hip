// Block fixed-point eikonal solve for gfx950 (CDNA4): the replacement for the reference's serial
// narrow-band march `travel`/`fouds2` + binary tree (CalSurfG.f90:288-487, :587-759, :768-921).
//
// One workgroup owns one problem (one source's field) from start to convergence, so every
// hand-off between waves stays on one CU (workgroup-scope ordering only, no cross-XCD traffic).
// The field is cut into 8x8-node blocks; a wave (64 lanes) owns one block at a time, one node per
// lane, with the block's 12x12 neighbourhood staged in LDS.  Work is driven by
//   * a 64-bit dirty mask per block in HBM (bit = node whose neighbourhood changed),
//   * the active-block list and its membership bitset in LDS,
//   * a causal window: per round only nodes that can be accepted before theta = (earliest pending
//     time) + window are evaluated.  Without it the iteration needs >100 evaluations per node at
//     1025^2 (and can oscillate); with it ~5, independent of the window width.
// Node convergence inside a block is decided with wave ballots; the per-node update is
// dsa::solve_node, whose fixed point is the Fast-Marching field.
//
// HBM-bound?  No: ~5 evaluations x ~600 fp32 instructions per node against 8 bytes of
// algorithmic traffic per node; the kernel is latency/VALU bound (roofline).
#include "kernels.h"

namespace dsa {

namespace {

constexpr int kTile = 12;          // 8 + 2*2 halo
constexpr int kMaxInner = 16;      // inner sweeps per block visit

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }

// shift a 64-bit block mask (bit = lx*8 + lz) by (dx, dz) inside the block, dropping what leaves it
__device__ __forceinline__ unsigned long long shift_x(unsigned long long m, int dx)
{
    return dx > 0 ? (m << (8 * dx)) : (m >> (8 * -dx));
}
__device__ __forceinline__ unsigned long long shift_z(unsigned long long m, int dz)
{
    // columns are bytes; shift every byte, masking bits that cross a byte boundary
    if (dz > 0) {
        const unsigned long long keep = 0x0101010101010101ull * (unsigned long long)(0xffu >> dz);
        return (m & keep) << dz;
    }
    const int s = -dz;
    const unsigned long long keep = 0x0101010101010101ull * (unsigned long long)((0xffu << s) & 0xffu);
    return (m & keep) >> s;
}

struct Lists {
    int* list[2];
    unsigned* member;       // bitset: block is in the current or next list
    int cap;
};

__device__ __forceinline__ void push_block(const Lists& L, int nxt, int b, int* next_cnt, int* overflow)
{
    const unsigned bit = 1u << (b & 31);
    const unsigned old = atomicOr(&L.member[b >> 5], bit);
    if (old & bit) return;
    const int pos = atomicAdd(next_cnt, 1);
    if (pos < L.cap) L.list[nxt][pos] = b;
    else { atomicAnd(&L.member[b >> 5], ~bit); *overflow = 1; }
}

}  // namespace

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_fim(const FimProblem* __restrict__ problems, int list_cap,
                                                      int member_words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FimProblem p = problems[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblocks = p.nbx * p.nbz;

    // LDS carve: lists | member bitset | per-wave tiles | scalars
    Lists L;
    L.list[0] = reinterpret_cast<int*>(smem);
    L.list[1] = L.list[0] + list_cap;
    L.member = reinterpret_cast<unsigned*>(L.list[1] + list_cap);
    L.cap = list_cap;
    float* tiles = reinterpret_cast<float*>(L.member + member_words);
    float* tile = tiles + wave * (kTile * kTile);
    int* sc = reinterpret_cast<int*>(tiles + NWAVES * kTile * kTile);
    // sc[0] next count, sc[1] overflow, sc[2] tmin bits, sc[3] current count
    for (int i = tid; i < member_words; i += NWAVES * 64) L.member[i] = 0u;
    if (tid == 0) { sc[0] = 0; sc[1] = 1; sc[2] = 0; sc[3] = 0; }   // overflow=1 forces the initial scan
    __syncthreads();

    const int lz = lane & 7, lx = lane >> 3;
    int cur = 0, rounds = 0, visits = 0, overflows = 0;

    for (;;) {
        int cnt = sc[3];
        if (cnt == 0) {
            // (re)build the list from the dirty masks; also the start of the solve
            if (sc[1] == 0) break;
            __syncthreads();
            if (tid == 0) { sc[1] = 0; sc[0] = 0; }
            __syncthreads();
            for (int b = tid; b < nblocks; b += NWAVES * 64)
                if (__hip_atomic_load(&p.mask[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)
                    push_block(L, cur, b, &sc[0], &sc[1]);
            __syncthreads();
            cnt = sc[0] < list_cap ? sc[0] : list_cap;
            __syncthreads();
            if (tid == 0) { sc[3] = cnt; sc[0] = 0; if (sc[1]) ++overflows; }
            __syncthreads();
            if (cnt == 0) break;
        }
        // earliest pending time over the active blocks
        if (tid == 0) sc[2] = 0x7f800000;
        __syncthreads();
        for (int i = tid; i < cnt; i += NWAVES * 64) {
            const float k = __hip_atomic_load(&p.key[L.list[cur][i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMin(reinterpret_cast<unsigned*>(&sc[2]), f2u(k));
        }
        __syncthreads();
        const float tmin = u2f((unsigned)sc[2]);
        const float theta = tmin + p.window;          // +inf when every key is +inf: evaluate everything
        const bool open = !(theta < kInf);
        const int nxt = cur ^ 1;

        for (int i = wave; i < cnt; i += NWAVES) {
            const int b = L.list[cur][i];
            const float kb = __hip_atomic_load(&p.key[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!open && !(kb < theta)) {
                // deferred: stays a member, just carried over
                if (lane == 0) {
                    const int pos = atomicAdd(&sc[0], 1);
                    if (pos < list_cap) L.list[nxt][pos] = b;
                    else { atomicAnd(&L.member[b >> 5], ~(1u << (b & 31))); sc[1] = 1; }
                }
                continue;
            }
            ++visits;
            const int bx = b / p.nbz, bz = b - bx * p.nbz;
            const int iz0 = bz * 8, ix0 = bx * 8;            // 0-based origin of the block
            unsigned long long m = 0ull;
            if (lane == 0) {
                atomicAnd(&L.member[b >> 5], ~(1u << (b & 31)));
                __hip_atomic_store(&p.key[b], kInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m = atomicExch(&p.mask[b], 0ull);
            }
            m = __shfl(m, 0);
            if (m == 0ull) continue;

            // stage the 12x12 neighbourhood
            for (int t = lane; t < kTile * kTile; t += 64) {
                const int tx = t / kTile, tz = t - tx * kTile;
                const int gz = iz0 - 2 + tz, gx = ix0 - 2 + tx;
                float v = kInf;
                if (gz >= 0 && gz < p.nnz && gx >= 0 && gx < p.nnx) v = p.T[(size_t)gx * p.nnz + gz];
                tile[tx * kTile + tz] = v;
            }
            const int gz = iz0 + lz, gx = ix0 + lx;           // 0-based node of this lane
            const bool valid = gz < p.nnz && gx < p.nnx;
            const size_t gid = (size_t)gx * p.nnz + gz;
            const float slown = valid ? p.slow[gid] : 1.0f;
            NodeGeom geom = { p.ri, valid ? p.risti[gx] : 1.0f, p.dnx, p.dnz };
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int c0 = (lx + 2) * kTile + (lz + 2);
            float own = tile[c0];
            const bool can = valid && !t_pinned(own);
            bool dirty = can && ((m >> lane) & 1ull);
            bool ever = false;
            Hood h;
            h.in[0] = gx - 1 >= 0;      h.in_outer[0] = gx - 2 >= 0;
            h.in[1] = gx + 1 < p.nnx;   h.in_outer[1] = gx + 2 < p.nnx;
            h.in[2] = gz - 1 >= 0;      h.in_outer[2] = gz - 2 >= 0;
            h.in[3] = gz + 1 < p.nnz;   h.in_outer[3] = gz + 2 < p.nnz;
            unsigned long long call = 0ull;
            float lb = kInf;
            for (int it = 0; it < kMaxInner; ++it) {
                h.near_[0] = tile[c0 - kTile];     h.outer[0] = tile[c0 - 2 * kTile];
                h.near_[1] = tile[c0 + kTile];     h.outer[1] = tile[c0 + 2 * kTile];
                h.near_[2] = tile[c0 - 1];         h.outer[2] = tile[c0 - 2];
                h.near_[3] = tile[c0 + 1];         h.outer[3] = tile[c0 + 2];
                lb = fminf(fminf(t_value(h.near_[0]), t_value(h.near_[1])), fminf(t_value(h.near_[2]), t_value(h.near_[3])));
                const bool go = dirty && (open || lb < theta);
                if (__ballot(go) == 0ull) break;
                bool changed = false;
                if (go) {
                    const float c = solve_node(h, slown, geom);
                    changed = f2u(c) != f2u(own);
                    dirty = false;
                    if (changed) { own = c; ever = true; }
                }
                __builtin_amdgcn_wave_barrier();              // all lanes have read the old tile
                if (changed) tile[c0] = own;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long C = __ballot(changed);
                call |= C;
                const unsigned long long D = shift_x(C, 1) | shift_x(C, -1) | shift_x(C, 2) | shift_x(C, -2) |
                                             shift_z(C, 1) | shift_z(C, -1) | shift_z(C, 2) | shift_z(C, -2);
                if (can && ((D >> lane) & 1ull)) dirty = true;
            }
            if (ever) p.T[gid] = own;

            // what is left dirty here waits for a later round
            const unsigned long long rem = __ballot(dirty);
            float kmin = dirty ? lb : kInf;
            float vmin = ever ? own : kInf;
            for (int o = 32; o > 0; o >>= 1) {
                kmin = fminf(kmin, __shfl_xor(kmin, o));
                vmin = fminf(vmin, __shfl_xor(vmin, o));
            }
            if (lane == 0) {
                if (rem) {
                    atomicOr(&p.mask[b], rem);
                    atomicMin(reinterpret_cast<unsigned*>(&p.key[b]), f2u(kmin));
                    push_block(L, nxt, b, &sc[0], &sc[1]);
                }
                if (call) {
                    // nodes within two steps of a block face depend on changed nodes of this block
                    const unsigned long long colmask0 = 0x00000000000000ffull;   // lx = 0
                    const unsigned long long rowmask0 = 0x0101010101010101ull;   // lz = 0
                    // x- neighbour block: its columns lx=7 (from our lx=0,1) and lx=6 (from our lx=0)
                    const unsigned long long c_x0 = call & colmask0, c_x1 = (call >> 8) & colmask0;
                    const unsigned long long c_x7 = (call >> 56) & colmask0, c_x6 = (call >> 48) & colmask0;
                    const unsigned long long to_xm = ((c_x0 | c_x1) << 56) | (c_x0 << 48);
                    const unsigned long long to_xp = (c_x7 | c_x6) | (c_x7 << 8);
                    const unsigned long long c_z0 = call & rowmask0, c_z1 = (call >> 1) & rowmask0;
                    const unsigned long long c_z7 = (call >> 7) & rowmask0, c_z6 = (call >> 6) & rowmask0;
                    const unsigned long long to_zm = ((c_z0 | c_z1) << 7) | (c_z0 << 6);
                    const unsigned long long to_zp = (c_z7 | c_z6) | (c_z7 << 1);
                    const unsigned vk = f2u(vmin);
                    if (to_xm && bx > 0) { const int nb = b - p.nbz; atomicOr(&p.mask[nb], to_xm); atomicMin(reinterpret_cast<unsigned*>(&p.key[nb]), vk); push_block(L, nxt, nb, &sc[0], &sc[1]); }
                    if (to_xp && bx + 1 < p.nbx) { const int nb = b + p.nbz; atomicOr(&p.mask[nb], to_xp); atomicMin(reinterpret_cast<unsigned*>(&p.key[nb]), vk); push_block(L, nxt, nb, &sc[0], &sc[1]); }
                    if (to_zm && bz > 0) { const int nb = b - 1; atomicOr(&p.mask[nb], to_zm); atomicMin(reinterpret_cast<unsigned*>(&p.key[nb]), vk); push_block(L, nxt, nb, &sc[0], &sc[1]); }
                    if (to_zp && bz + 1 < p.nbz) { const int nb = b + 1; atomicOr(&p.mask[nb], to_zp); atomicMin(reinterpret_cast<unsigned*>(&p.key[nb]), vk); push_block(L, nxt, nb, &sc[0], &sc[1]); }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { const int n = sc[0]; sc[3] = n < list_cap ? n : list_cap; sc[0] = 0; if (sc[1]) ++overflows; }
        cur = nxt;
        ++rounds;
        __syncthreads();
        if (rounds > 400000) { if (tid == 0) p.info[3] = 1; break; }
    }
    // totals (visits summed over waves)
    if (lane == 0) atomicAdd(&p.info[1], visits);
    if (tid == 0) { p.info[0] = rounds; p.info[2] = overflows; }
}

size_t fim_lds_bytes(const FimLaunch& l, int nwaves)
{
    const size_t member_words = (size_t)(l.max_blocks + 31) / 32;
    return (size_t)2 * l.list_cap * sizeof(int) + member_words * sizeof(unsigned) +
           (size_t)nwaves * kTile * kTile * sizeof(float) + 16 * sizeof(int);
}

void launch_fim(const FimProblem* d_problems, int nproblems, const FimLaunch& l, hipStream_t stream)
{
    if (nproblems <= 0) return;
    constexpr int NW = 8;
    const int member_words = (l.max_blocks + 31) / 32;
    const size_t lds = fim_lds_bytes(l, NW);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fim<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fim<NW>, dim3(nproblems), dim3(NW * 64), lds, stream, d_problems, l.list_cap, member_words);
}

}  // namespace dsa
