// Fixed-point eikonal solve for gfx950 (CDNA4): the replacement for the reference's serial
// narrow-band march `travel`/`fouds2` + binary tree (CalSurfG.f90:288-487, :587-759, :768-921).
//
// One workgroup owns one problem (one source's field) from start to convergence, so all
// communication stays on one CU: workgroup-scope ordering, LDS counters, no cross-XCD traffic.
// The unit of work is a NODE, kept in a compacted active list in LDS:
//
//   round:  pass A  every listed node computes its lower bound (earliest acceptance time of a
//                   neighbour).  Nodes inside the causal window [tmin, tmin + window) are moved to
//                   a dense `ready` list, the rest carry over to the next round.
//           pass B  the ready list is evaluated with full lanes (dsa::solve_node).  A node whose
//                   (T, tau) changed stores it and activates its 8 stencil dependents with an
//                   atomic test-and-set on the `queued` bit (the sign bit of tau).
//
// Why this shape (measured, DESIGN.md "scheduling"): the local solver is ~600 fp32 instructions;
// what matters is how many times it runs per node and how full the lanes are.  Evaluating nodes
// in lockstep ahead of the front costs 50-260 evaluations per node; evaluating only what the
// causal window allows costs ~5, and compaction keeps the lanes full (a front layer inside an
// 8x8 tile is 8-11 nodes, i.e. 15 % of a wave).
//
// Correctness of the hand-offs inside a round:
//   * a node's queued bit is cleared in pass A, before the barrier, so every change that lands
//     while the node is being evaluated re-queues it (no lost update);
//   * a changed node stores (T, tau) first and activates dependents afterwards; a dependent that
//     read a torn or stale state this round has its bit clear and is therefore re-queued;
//   * the fixed point is schedule independent, so races only cost re-evaluations.
#include "kernels.h"

namespace dsa {

namespace {

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }

enum { SC_CUR = 0, SC_NEXT, SC_READY, SC_TMIN, SC_OVERFLOW, SC_THETA, SC_READY_ODD, SC_FREEZE, SC_COUNT = 8 };
constexpr int kStallRounds = 12;

struct Lists {
    int* cur;
    int* next;
    int* ready;
    int cap, rcap;
    int* sc;
};

__device__ __forceinline__ void push_next(const Lists& L, int id)
{
    const int pos = atomicAdd(&L.sc[SC_NEXT], 1);
    if (pos < L.cap) L.next[pos] = id;
    else L.sc[SC_OVERFLOW] = 1;        // the node keeps its queued bit; a rescan picks it up
}

}  // namespace

template <int NT>
__global__ __launch_bounds__(NT) void k_fim(const FimProblem* __restrict__ problems, int cap, int rcap)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const FimProblem p = problems[blockIdx.x];
    const int tid = threadIdx.x;
    Lists L;
    L.cur = smem; L.next = smem + cap; L.ready = smem + 2 * cap; L.sc = smem + 2 * cap + rcap;
    L.cap = cap; L.rcap = rcap;
    int* sc = L.sc;
    unsigned* tau_bits = reinterpret_cast<unsigned*>(p.tau);
    const int nnz = p.nnz, nnx = p.nnx;
    const int rhalf = rcap / 2;

    const int nseed = *p.seed_count;
    if (tid == 0) {
        sc[SC_CUR] = nseed < cap ? nseed : cap; sc[SC_NEXT] = 0; sc[SC_READY] = 0; sc[SC_READY_ODD] = 0;
        sc[SC_TMIN] = 0x7f800000; sc[SC_OVERFLOW] = nseed > cap ? 1 : 0; sc[SC_THETA] = 0x7f800000;
        sc[SC_FREEZE] = (int)0xff800000u;      // -inf: nothing frozen
    }
    for (int i = tid; i < nseed && i < cap; i += NT) L.cur[i] = p.seed[i];
    __syncthreads();

    int rounds = 0, rescans = 0, stall = 0, freezes = 0;   // stall bookkeeping is used by thread 0 only
    float best_tmin = -kInf;
    unsigned long long evals = 0;
    for (;;) {
        int cnt = sc[SC_CUR];
        if (cnt == 0) {
            if (sc[SC_OVERFLOW] == 0) break;
            // some queued nodes did not fit into a list: collect them again from the field
            __syncthreads();
            if (tid == 0) { sc[SC_OVERFLOW] = 0; sc[SC_NEXT] = 0; }
            __syncthreads();
            const int n = nnx * nnz;
            for (int id = tid; id < n; id += NT)
                if ((__hip_atomic_load(&tau_bits[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & kQueuedBit) &&
                    !t_pinned(p.T[id])) {
                    const int pos = atomicAdd(&sc[SC_NEXT], 1);
                    if (pos < cap) L.cur[pos] = id; else sc[SC_OVERFLOW] = 1;
                }
            __syncthreads();
            cnt = sc[SC_NEXT] < cap ? sc[SC_NEXT] : cap;
            __syncthreads();
            if (tid == 0) { sc[SC_CUR] = cnt; sc[SC_NEXT] = 0; sc[SC_READY] = 0; sc[SC_READY_ODD] = 0; sc[SC_THETA] = 0x7f800000; sc[SC_TMIN] = 0x7f800000; }
            ++rescans;
            __syncthreads();
            if (cnt == 0) break;
        }
        const float theta = u2f((unsigned)sc[SC_THETA]);
        const bool open = !(theta < kInf);
        const float freeze = u2f((unsigned)sc[SC_FREEZE]);
        const bool frozen_any = freeze > -kInf;

        // ---- pass A: lower bounds, routing ---------------------------------------------------
        for (int base = 0; base < cnt; base += NT) {
            const int i = base + tid;
            if (i < cnt) do {
                const int id = L.cur[i];
                const int ix = id / nnz, iz = id - ix * nnz;        // 0-based
                // accepted below the freeze horizon: final (see "stall" at the end of the round)
                if (frozen_any && tau_value(p.tau[id]) < freeze) { atomicAnd(&tau_bits[id], ~kQueuedBit); continue; }
                float lb = kInf;
                if (ix > 0) lb = fminf(lb, tau_value(p.tau[id - nnz]));
                if (ix + 1 < nnx) lb = fminf(lb, tau_value(p.tau[id + nnz]));
                if (iz > 0) lb = fminf(lb, tau_value(p.tau[id - 1]));
                if (iz + 1 < nnz) lb = fminf(lb, tau_value(p.tau[id + 1]));
                bool ready = open || lb < theta;
                if (ready) {
                    // even nodes use the first half of the ready buffer, odd nodes the second half
                    const bool odd = ((ix + iz) & 1) != 0;
                    const int mine = atomicAdd(&sc[odd ? SC_READY_ODD : SC_READY], 1);
                    if (mine < rhalf) {
                        L.ready[odd ? rhalf + mine : mine] = id;
                        atomicAnd(&tau_bits[id], ~kQueuedBit);      // before the barrier: see header
                    } else ready = false;                           // counter is clamped when read
                }
                if (!ready) {
                    push_next(L, id);
                    atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(lb));
                }
            } while (0);
        }
        __syncthreads();

        // ---- pass B: evaluate the ready nodes, even nodes first, then odd ones.  Adjacent nodes are
        // never evaluated in the same sub-pass, so the second half sees the first half's results
        // (red-black Gauss-Seidel: fewer rounds and fewer evaluations than one simultaneous pass).
        const int nready_even = sc[SC_READY] < rhalf ? sc[SC_READY] : rhalf;
        const int nready_odd = sc[SC_READY_ODD] < rhalf ? sc[SC_READY_ODD] : rhalf;
        for (int half = 0; half < 2; ++half) {
            const int nready = half ? nready_odd : nready_even;
            for (int j = tid; j < nready; j += NT) {
                const int id = L.ready[half ? rhalf + j : j];
                const int ix = id / nnz, iz = id - ix * nnz;
                Hood h;
                const int off[4] = { -nnz, nnz, -1, 1 };
                h.in[0] = ix > 0;          h.in_outer[0] = ix > 1;
                h.in[1] = ix + 1 < nnx;    h.in_outer[1] = ix + 2 < nnx;
                h.in[2] = iz > 0;          h.in_outer[2] = iz > 1;
                h.in[3] = iz + 1 < nnz;    h.in_outer[3] = iz + 2 < nnz;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h.near_[q] = h.in[q] ? p.T[id + off[q]] : kInf;
                    h.near_tau[q] = h.in[q] ? p.tau[id + off[q]] : kInf;
                    h.outer[q] = h.in_outer[q] ? p.T[id + 2 * off[q]] : kInf;
                    h.outer_tau[q] = h.in_outer[q] ? p.tau[id + 2 * off[q]] : kInf;
                }
                const float t_old = p.T[id];
                const float k_old = tau_value(p.tau[id]);
                if (t_pinned(t_old)) continue;
                const NodeGeom geom = { p.ri, p.risti[ix], p.dnx, p.dnz };
                float k;
                const float c = solve_node(h, p.slow[id], geom, &k);
                ++evals;
                if (f2u(c) != f2u(t_old) || f2u(k) != f2u(k_old)) {
                    p.T[id] = c;
                    p.tau[id] = k;                                     // queued bit clear
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (h.in[q] && !t_pinned(h.near_[q]) && !(f2u(h.near_tau[q]) & kQueuedBit)) {
                            const unsigned old = atomicOr(&tau_bits[id + off[q]], kQueuedBit);
                            if (!(old & kQueuedBit)) push_next(L, id + off[q]);
                        }
                        // The node two steps away uses this one only through the node in between;
                        // while that one is unreached the dependency is moot (and it will activate
                        // the outer node itself when it changes).  Queuing it anyway floods the
                        // list with nodes that can never become ready.
                        if (h.in_outer[q] && tau_value(h.near_tau[q]) < kInf && !t_pinned(h.outer[q]) &&
                            !(f2u(h.outer_tau[q]) & kQueuedBit)) {
                            const unsigned old = atomicOr(&tau_bits[id + 2 * off[q]], kQueuedBit);
                            if (!(old & kQueuedBit)) push_next(L, id + 2 * off[q]);
                        }
                    }
                    atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(k));
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int n = sc[SC_NEXT];
            sc[SC_CUR] = n < cap ? n : cap;
            // a round that dropped nodes and evaluated nothing is clogged: rebuild from the field
            if (sc[SC_OVERFLOW] && nready_even + nready_odd == 0) sc[SC_CUR] = 0;
            sc[SC_NEXT] = 0; sc[SC_READY] = 0; sc[SC_READY_ODD] = 0;
            const float tmin = u2f((unsigned)sc[SC_TMIN]);
            sc[SC_THETA] = (int)f2u(tmin + p.window);
            sc[SC_TMIN] = 0x7f800000;
            // Stall: by causality every node accepted before the earliest pending bound is final, so
            // that bound must keep rising.  If it does not for kStallRounds rounds, what is left in the
            // window is a cluster of mutually tied nodes flipping by an ulp (Fast Marching never sees
            // this: a popped node is frozen).  Freeze everything accepted below the window's edge.
            if (tmin > best_tmin) { best_tmin = tmin; stall = 0; }
            else if (tmin < kInf && ++stall >= kStallRounds) { sc[SC_FREEZE] = (int)f2u(best_tmin + p.window); stall = 0; ++freezes; }
        }
        int* t = L.cur; L.cur = L.next; L.next = t;
        ++rounds;
        __syncthreads();
        if (rounds > p.max_rounds) { if (tid == 0) p.info[2] = -1; break; }
    }
    // counters: evaluations summed over threads
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
    if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 4), evals);
    if (tid == 0) { p.info[0] = rounds; p.info[1] = rescans; p.info[3] = freezes; }
}

size_t fim_lds_bytes(const FimLaunch& l)
{
    return ((size_t)2 * l.list_cap + l.ready_cap + SC_COUNT) * sizeof(int);
}

void launch_fim(const FimProblem* d_problems, int nproblems, const FimLaunch& l, hipStream_t stream)
{
    if (nproblems <= 0) return;
    constexpr int NT = 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fim<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fim<NT>, dim3(nproblems), dim3(NT), fim_lds_bytes(l), stream, d_problems, l.list_cap, l.ready_cap);
}

}  // namespace dsa
