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
#if defined(DSA_LEDGER)      // probe build of tools/isa_ledger.py, run on the GPU: named markers in the assembly and one wave-trip counter per marker
#define DSA_LEDGER_PARAM , unsigned* dsa_lc
#define DSA_LEDGER_PASS , dsa_lc
#define DSA_LEDGER_COUNT(k, name)                                                                                              \
    do {                                                                                                                       \
        asm volatile("; LEDGER " #k " " name);                                                                                 \
        const unsigned long long ex_ = __builtin_amdgcn_read_exec();                                                           \
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)ex_) - 1) dsa_lc[k] += 1u;                                          \
    } while (0)
#elif defined(DSA_LEDGER_MARKS)   // the markers alone (assembly comments: no instructions): what tools/isa_ledger.py counts the static mix on
#define DSA_LEDGER_PARAM
#define DSA_LEDGER_PASS
#define DSA_LEDGER_COUNT(k, name) asm volatile("; LEDGER " #k " " name)
#endif
#include "kernels.h"
#include "receiver_core.h"
#include "wave_ops.h"

#include <type_traits>

namespace dsa {

namespace {

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }

enum { SC_CUR = 0, SC_NEXT, SC_READY, SC_TMIN, SC_OVERFLOW, SC_THETA, SC_READY_ODD, SC_FREEZE, SC_HASH, SC_COUNT = 12 };
constexpr int kCycleRounds = 8;     // rounds of exactly repeating changes before the window is frozen

typedef __attribute__((address_space(1))) int GI32;
struct Lists {
    GI32* cur;      // lists live in per-problem global scratch (sequential access, L2 resident);
    GI32* next;     // only the counters are in LDS, so several problems fit on one CU
    GI32* ready;
    int cap, rcap;
    int* sc;
};

// Wave-aggregated slot allocation: one LDS atomic per wave instead of one per lane (same-address LDS
// atomics serialise lane by lane).  Must be reached by all lanes that are active at the call site;
// returns the lane's slot, or -1 for lanes that do not want one.
__device__ __forceinline__ int wave_alloc(int* counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// A list entry is (node, key): key is an upper bound of the node's lower bound at the time it was
// queued (the acceptance time of the neighbour that queued it), so `key < theta` routes the node
// without touching its neighbourhood; +inf means "unknown, compute it".
__device__ __forceinline__ void push_next(const Lists& L, int id, float key, bool want)
{
    const int pos = wave_alloc(&L.sc[SC_NEXT], want);
    if (want) {
        if (pos < L.cap) { L.next[2 * pos] = id; L.next[2 * pos + 1] = (int)f2u(key); }
        else L.sc[SC_OVERFLOW] = 1;    // the node keeps its queued bit; a rescan picks it up
    }
}

}  // namespace

#ifndef DSA_FIM_WAVES
#define DSA_FIM_WAVES 4
#endif
// DSA_ODD_CLEAR (kernels.h): 1 = an odd node that is evaluated in a round ignores what the even half of that round activated
// (its evaluation has seen those changes): -16 % evaluations at the same fixed point (tests/tools/sched_lab.cpp mode 6).  The
// active set of a tile is then three masks and a stamp, {E, O, R, round}: activations of the even half go to E, those of the odd
// half to O, R holds the nodes the odd half of round `round - 1` evaluated; pass A takes (E & ~R) | O and rewrites the record
// with plain stores (a tile has one owner in pass A and nobody activates then).  [Clearing the odd nodes' bits with atomics
// after the even half, in three placements, lost: the mask lines are cold by then and every clear went to memory.]
#ifndef DSA_FIM_GROUP_SHIFT
#define DSA_FIM_GROUP_SHIFT -1     // log2 of the bitmap words per group handed to a wave (pass A of k_fim_sorted); -1: one tile column
#endif
// One workgroup of NT threads per problem.  The solver wants ~124 VGPRs, i.e. 4 waves per SIMD: tell the
// compiler so, otherwise it targets 8 waves/SIMD (64 VGPRs) and spills the solver.
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(DSA_FIM_WAVES, DSA_FIM_WAVES))) void k_fim(const FimProblem* __restrict__ problems, int cap, int rcap)
{
    __shared__ int smem[SC_COUNT];
    const FimProblem p = problems[blockIdx.x];
    const int tid = threadIdx.x;
    Lists L;
    L.cur = (GI32*)p.lists; L.next = L.cur + 2 * cap; L.ready = L.cur + 4 * cap; L.sc = smem;
    L.cap = cap; L.rcap = rcap;
    int* sc = L.sc;
    // The pointers come out of a struct in memory, so the compiler would use FLAT instructions (which
    // also count against the LDS wait counter at every barrier); they are global memory.
    typedef __attribute__((address_space(1))) Rec GRec;
    typedef __attribute__((address_space(1))) unsigned GU32;
    typedef __attribute__((address_space(1))) const float GCF32;
    GRec* const F = (GRec*)p.F;
    GCF32* const slow = (GCF32*)p.slow;
    GCF32* const risti = (GCF32*)p.risti;
    const int nnz = p.nnz, nnx = p.nnx, nbz = p.nbz;
    auto tau_word = [&](int id) { return (unsigned*)(GU32*)&F[id].tau; };
    auto ld = [&](int id) { Rec r; r.T = F[id].T; r.tau = F[id].tau; return r; };
    int max_cnt = 0;
    const int rhalf = rcap / 2;

    const int nseed = *p.seed_count;
    if (tid == 0) {
        sc[SC_CUR] = nseed <= cap && nseed <= p.seed_cap ? nseed : 0; sc[SC_NEXT] = 0; sc[SC_READY] = 0; sc[SC_READY_ODD] = 0;
        sc[SC_TMIN] = 0x7f800000; sc[SC_OVERFLOW] = (nseed > cap || nseed > p.seed_cap) ? 1 : 0; sc[SC_THETA] = 0x7f800000;
        sc[SC_FREEZE] = (int)0xff800000u;      // -inf: nothing frozen
        sc[SC_HASH] = 0;
    }
    if (nseed <= cap && nseed <= p.seed_cap)
        for (int i = tid; i < nseed; i += NT) { L.cur[2 * i] = p.seed[i]; L.cur[2 * i + 1] = 0x7f800000; }
    __syncthreads();

    int rounds = 0, rescans = 0, stall = 0, freezes = 0;   // cycle bookkeeping is used by thread 0 only
    unsigned hist[4] = { 1u, 2u, 3u, 4u };
    unsigned long long tA = 0, tB0 = 0, tB1 = 0, tE = 0, t0 = wall_clock64(), sum_cnt = 0, sum_ready = 0;   // phase clocks (thread 0)
    float best_tmin = -kInf;
    unsigned long long evals = 0, nchanged = 0;
    for (;;) {
        int cnt = sc[SC_CUR];
        if (cnt == 0) {
            if (sc[SC_OVERFLOW] == 0) break;
            // some queued nodes did not fit into a list: collect them again from the field
            __syncthreads();
            if (tid == 0) { sc[SC_OVERFLOW] = 0; sc[SC_NEXT] = 0; }
            __syncthreads();
            const int n = p.nbx * nbz * kTileRecs;
            for (int id = tid; id < n; id += NT)
                if ((__hip_atomic_load(tau_word(id), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & kQueuedBit) &&
                    !t_pinned(F[id].T)) {
                    const int pos = atomicAdd(&sc[SC_NEXT], 1);
                    if (pos < cap) { L.cur[2 * pos] = id; L.cur[2 * pos + 1] = 0x7f800000; } else sc[SC_OVERFLOW] = 1;
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

        // ---- pass A: lower bounds, routing.  Four entries per thread per trip so that their
        // neighbour loads are in flight together (the pass is pure latency otherwise).
        constexpr int UA = 4;
        for (int base = 0; base < cnt; base += NT * UA) {
            int ids[UA];
            float lbs[UA], own[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int i = base + u * NT + tid;
                ids[u] = i < cnt ? L.cur[2 * i] : -1;
                lbs[u] = i < cnt ? u2f((unsigned)L.cur[2 * i + 1]) : kInf;     // the entry's key
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                own[u] = kInf;
                if (ids[u] < 0) continue;
                const int id = ids[u];
                if (frozen_any) own[u] = F[id].tau;
                if (open || lbs[u] < theta) continue;               // routed by its key: no neighbour loads
#ifdef DSA_KEY_ROUTING
                // wait for the neighbour that queued this node to become final (its acceptance time inside the
                // window) instead of evaluating against its provisional value; seeds (+inf) use the bound
                if (lbs[u] < kInf && !frozen_any) continue;
#endif
                int iz, ix;                                         // 0-based
                rec_coords(nbz, id, &iz, &ix);
                const float a = ix > 0 ? F[rec_index(nbz, iz, ix - 1)].tau : kInf;
                const float b = ix + 1 < nnx ? F[rec_index(nbz, iz, ix + 1)].tau : kInf;
                const float c = iz > 0 ? F[rec_index(nbz, iz - 1, ix)].tau : kInf;
                const float d = iz + 1 < nnz ? F[rec_index(nbz, iz + 1, ix)].tau : kInf;
                lbs[u] = fminf(fminf(tau_value(a), tau_value(b)), fminf(tau_value(c), tau_value(d)));
            }
            float tmin_lane = kInf;
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int id = ids[u];
                const bool have = id >= 0;
                // accepted below the freeze horizon: final (see "cycle" at the end of the round)
                const bool frozen = have && frozen_any && tau_value(own[u]) < freeze;
                const float lb = lbs[u];
                const bool cand = have && !frozen;
                int iz = 0, ix = 0;
                if (have) rec_coords(nbz, id, &iz, &ix);
                const bool odd = ((ix + iz) & 1) != 0;
                // even nodes use the first half of the ready buffer, odd nodes the second half
                const bool want_e = cand && (open || lb < theta) && !odd;
                const bool want_o = cand && (open || lb < theta) && odd;
                const int pe = wave_alloc(&sc[SC_READY], want_e);
                const int po = wave_alloc(&sc[SC_READY_ODD], want_o);
                const bool got = (want_e && pe < rhalf) || (want_o && po < rhalf);   // counters are clamped when read
                if (got) L.ready[want_o ? rhalf + po : pe] = id;
                if (got || frozen) atomicAnd(tau_word(id), ~kQueuedBit);             // before the barrier: see header
                const bool defer = cand && !got;
                push_next(L, id, lb, defer);
                if (defer) tmin_lane = fminf(tmin_lane, lb);
            }
            tmin_lane = wave_min(tmin_lane);
            if ((tid & 63) == 0 && tmin_lane < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(tmin_lane));
        }
        __syncthreads();
        { const unsigned long long t1 = wall_clock64(); tA += t1 - t0; t0 = t1; sum_cnt += cnt; if (cnt > max_cnt) max_cnt = cnt; }

        // ---- pass B: evaluate the ready nodes, even nodes first, then odd ones.  Adjacent nodes are
        // never evaluated in the same sub-pass, so the second half sees the first half's results
        // (red-black Gauss-Seidel: fewer rounds and fewer evaluations than one simultaneous pass).
        const int nready_even = sc[SC_READY] < rhalf ? sc[SC_READY] : rhalf;
        const int nready_odd = sc[SC_READY_ODD] < rhalf ? sc[SC_READY_ODD] : rhalf;
        for (int half = 0; half < 2; ++half) {
            const int nready = half ? nready_odd : nready_even;
            for (int j0 = 0; j0 < nready; j0 += NT) {
                const int j = j0 + tid;
                const bool act = j < nready;                         // whole waves stay in the loop body
                const int id = act ? L.ready[half ? rhalf + j : j] : 0;
                int iz, ix;
                rec_coords(nbz, id, &iz, &ix);
                Hood h;
                h.in[0] = act && ix > 0;          h.in_outer[0] = act && ix > 1;
                h.in[1] = act && ix + 1 < nnx;    h.in_outer[1] = act && ix + 2 < nnx;
                h.in[2] = act && iz > 0;          h.in_outer[2] = act && iz > 1;
                h.in[3] = act && iz + 1 < nnz;    h.in_outer[3] = act && iz + 2 < nnz;
                // record indices of the 8 stencil nodes (x-, x+, z-, z+; near then outer)
                int nid[8];
                nid[0] = rec_index(nbz, iz, ix - 1); nid[4] = rec_index(nbz, iz, ix - 2);
                nid[1] = rec_index(nbz, iz, ix + 1); nid[5] = rec_index(nbz, iz, ix + 2);
                nid[2] = rec_index(nbz, iz - 1, ix); nid[6] = rec_index(nbz, iz - 2, ix);
                nid[3] = rec_index(nbz, iz + 1, ix); nid[7] = rec_index(nbz, iz + 2, ix);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const Rec a = h.in[q] ? ld(nid[q]) : Rec{ kInf, kInf };
                    const Rec b = h.in_outer[q] ? ld(nid[4 + q]) : Rec{ kInf, kInf };
                    h.near_[q] = a.T; h.near_tau[q] = a.tau;
                    h.outer[q] = b.T; h.outer_tau[q] = b.tau;
                }
                const Rec own = act ? ld(id) : Rec{ -1.0f, 0.0f };    // inactive lanes read as pinned
                const float t_old = own.T;
                const float k_old = tau_value(own.tau);
                bool changed = false;
                float c = 0.0f, k = kInf;
                if (!t_pinned(t_old)) {
                    const NodeGeom geom = { p.ri, risti[ix], p.dnx, p.dnz };
                    c = solve_node(h, slow[id], geom, &k);
                    ++evals;
                    changed = f2u(c) != f2u(t_old) || f2u(k) != f2u(k_old);
                }
                if (changed) { F[id].T = c; F[id].tau = k; ++nchanged; }   // adjacent stores (8 bytes); queued bit clear
                // Dependents.  A change of this node X can only matter to
                //   * a near node Y if X can enter Y's walk: min(tau_X old, new) <= tau_Y.  (If X is and
                //     was accepted later than Y, Y's walk stopped at or before X with a value <= tau_Y,
                //     and still does.)
                //   * an outer node Z only through a second-order leg over the in-between node Y: Y must
                //     be reached, T_Y > min(T_X old, new), and X must be able to be alive for Z:
                //     min(tau_X) < tau_Z.  While Y is unreached the dependency is moot (Y activates Z
                //     itself when it changes), and queuing Z anyway floods the list with nodes that can
                //     never become ready.
                // This more than halves the evaluations (5.2 -> 2.2 per node) with bit-identical fields
                // (tests/test_hostcheck.py runs both variants).  All test-and-sets are issued before any
                // result is consumed, so their L2 round trips overlap; list slots are allocated per wave.
                const float t_lo = fminf(t_value(t_old), c), k_lo = fminf(k_old, k);
                bool want[8];
                unsigned olds[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ky = tau_value(h.near_tau[q]);
                    want[q] = changed && h.in[q] && !t_pinned(h.near_[q]) && !(f2u(h.near_tau[q]) & kQueuedBit) && k_lo <= ky;
                    want[4 + q] = changed && h.in_outer[q] && ky < kInf && !t_pinned(h.outer[q]) &&
                                  !(f2u(h.outer_tau[q]) & kQueuedBit) && t_value(h.near_[q]) > t_lo &&
                                  k_lo < tau_value(h.outer_tau[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    olds[q] = want[q] ? atomicOr(tau_word(nid[q]), kQueuedBit) : kQueuedBit;
                    olds[4 + q] = want[4 + q] ? atomicOr(tau_word(nid[4 + q]), kQueuedBit) : kQueuedBit;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    push_next(L, nid[q], k, !(olds[q] & kQueuedBit));                                    // lb(Y) <= tau(X)
                    push_next(L, nid[4 + q], tau_value(h.near_tau[q]), !(olds[4 + q] & kQueuedBit));    // lb(Z) <= tau(Y)
                }
                const unsigned hv = wave_sum(changed ? (((unsigned)id * 2654435761u) ^ (f2u(c) * 40503u) ^ (f2u(k) * 2246822519u)) : 0u);
                const float kmin = wave_min(changed ? k : kInf);
                if ((tid & 63) == 0) {
                    if (hv) atomicAdd(reinterpret_cast<unsigned*>(&sc[SC_HASH]), hv);
                    if (kmin < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(kmin));
                }
            }
            __syncthreads();
            { const unsigned long long t1 = wall_clock64(); (half ? tB1 : tB0) += t1 - t0; t0 = t1; }
        }
        sum_ready += nready_even + nready_odd;
        if (tid == 0) {
            const int n = sc[SC_NEXT];
            sc[SC_CUR] = n < cap ? n : cap;
            // a round that dropped nodes and evaluated nothing is clogged: rebuild from the field
            if (sc[SC_OVERFLOW] && nready_even + nready_odd == 0) sc[SC_CUR] = 0;
            sc[SC_NEXT] = 0; sc[SC_READY] = 0; sc[SC_READY_ODD] = 0;
            const float tmin = u2f((unsigned)sc[SC_TMIN]);
            sc[SC_THETA] = (int)f2u(tmin + p.window);
            sc[SC_TMIN] = 0x7f800000;
            // Cycle: a cluster of mutually tied nodes can flip by an ulp forever (Fast Marching never
            // sees this: a popped node is frozen).  It shows as the complete set of changes repeating
            // with a short period, which the per-round hash of (node, T, tau) changes detects; settling
            // along a front never repeats exactly.  When that holds for kCycleRounds rounds, nothing
            // else inside the window is still moving, and by causality everything accepted before the
            // window's edge can be frozen.
            const unsigned hsh = (unsigned)sc[SC_HASH];
            sc[SC_HASH] = 0;
            if (tmin > best_tmin) best_tmin = tmin;
            const bool repeat = hsh != 0u && (hsh == hist[1] || hsh == hist[2] || hsh == hist[3] || hsh == hist[0]);
            hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = hsh;
            if (repeat) { if (++stall >= kCycleRounds) { sc[SC_FREEZE] = (int)f2u(best_tmin + p.window); stall = 0; ++freezes; } }
            else stall = 0;
        }
        GI32* t = L.cur; L.cur = L.next; L.next = t;
        ++rounds;
        __syncthreads();
        { const unsigned long long t1 = wall_clock64(); tE += t1 - t0; t0 = t1; }
        if (rounds > p.max_rounds) { if (tid == 0) p.info[2] = -1; break; }
    }
    // counters: evaluations summed over threads
    for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); nchanged += __shfl_xor(nchanged, o); }
    if ((tid & 63) == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 4), evals); atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 6), nchanged); }
    if (tid == 0) {
        p.info[0] = rounds; p.info[1] = rescans; p.info[3] = freezes;
        if (p.clocks) { p.clocks[0] = tA; p.clocks[1] = tB0; p.clocks[2] = tB1; p.clocks[3] = tE; p.clocks[4] = sum_cnt; p.clocks[5] = sum_ready; p.clocks[6] = (unsigned long long)max_cnt; }
    }
}


// ---------------------------------------------------------------------------------------------
// Variant with a spatially ordered active set.
//
// Measured on the list kernel above (profiles/r01_pmc_k_fim_tiled_pruned.txt): at full occupancy the
// SIMDs are 26 % busy and waves sit in s_waitcnt 70 % of the time; adding waves does not help.  What
// saturates is the per-CU vector memory path: lists grow in activation order, so the 64 lanes of a
// load touch 64 different 128-B lines (43.6 M L1 line accesses and 15 M L2 requests per solve for
// 2.35 M evaluations).  Here the active set is a 64-bit node mask per 8x8 tile (global scratch) plus
// a tile bitmap in LDS; every round the node list is rebuilt from them in record order (a block
// scan), so neighbouring lanes work on neighbouring nodes of the front and share lines.  Activation
// is an atomicOr on the tile's mask (dedupe = the old value), deferred nodes simply keep their bit,
// and the lists cannot overflow.  The round structure, the local solver, the pruning rules, the
// window and the cycle freeze are those of k_fim; results are bit-identical (same fixed point).
typedef __attribute__((address_space(1))) unsigned long long GU64;

template <int NT, bool COMPACT, bool TIE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(DSA_FIM_WAVES, DSA_FIM_WAVES))) void k_fim_sorted(const FimProblem* __restrict__ problems, int cap, int rcap, const FimEnds* __restrict__ ends)
{
    extern __shared__ unsigned dyn_lds[];
    __shared__ int smem[SC_COUNT];
    constexpr int kWaveBuf = 256;                            // nodes a wave expands at a time (128: -4 %, 512: -6 %)
    __shared__ int wbuf[(NT / 64) * kWaveBuf];
    constexpr int kTileBuf = NT > 512 ? 128 : 256;           // tiles a wave gathers at a time (16 waves: the per-tile scratch has to fit the LDS)
    __shared__ int wtile[(NT / 64) * kTileBuf];
    __shared__ unsigned wclr[(NT / 64) * kTileBuf * (DSA_ODD_CLEAR ? 4 : 2)];      // per gathered tile: the node bits that leave the active set (and the odd ones evaluated this round)
    const FimProblem p = problems[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* sc = smem;
    typedef __attribute__((address_space(1))) Rec GRec;
    typedef __attribute__((address_space(1))) const float GCF32;
    // All per-problem arrays are addressed as (uniform base, 32-bit byte offset): the loads and atomics then take the
    // base from scalar registers and one VGPR of offset instead of a 64-bit address pair computed on the VALU
    // (the largest field, 513^2 tiles of 512 B, is 135 MB).
    typedef __attribute__((address_space(1))) char GChar;
    // COMPACT (coarse problems): one float per node plus the exception table (eikonal_core.h); else (T, tau) records
    GChar* Fb = COMPACT ? (GChar*)p.Tc : (GChar*)p.F;       // (a claimed field slot replaces the three slot pointers below)
    GChar* excb = (GChar*)p.exc;
    const int xlog = p.exc_log2cap;
    GChar* const slowb = (GChar*)p.slow;
    GCF32* const risti = (GCF32*)p.risti;
    const int nnz = p.nnz, nnx = p.nnx, nbz = p.nbz;
    const int ntile = p.nbx * nbz, nwords = (ntile + 31) >> 5;
    GChar* maskb = (GChar*)p.lists;                          // ntile node masks
    auto rec = [&](int id) -> GRec* { return (GRec*)(Fb + ((unsigned)id << 3)); };
    auto slow_at = [&](int id) -> float { return *(GCF32*)(slowb + ((unsigned)id << 2)); };
    constexpr bool kOddR = DSA_ODD_CLEAR != 0;
    constexpr bool kKeyMasks = DSA_KEY_MASKS != 0;          // (implies the tile records of DSA_ODD_CLEAR)
    static_assert(!kKeyMasks || kOddR, "DSA_KEY_MASKS needs DSA_ODD_CLEAR");
    constexpr int kMaskShift = kKeyMasks ? 6 : (kOddR ? 5 : 3), kClrWords = kOddR ? 4 : 2;
    auto mask_at = [&](int tile) -> GU64* { return (GU64*)(maskb + ((size_t)(unsigned)tile << kMaskShift)); };
    constexpr int rhalf = NT * 4;                            // ready nodes of one colour a round can take (8 per thread: no gain)
    __shared__ int ready[2 * rhalf];                         // (the rest stay in their masks for the next round)
    unsigned* const tb = dyn_lds;                            // tile bitmap, nwords
    auto ld = [&](int id) { Rec r; r.T = rec(id)->T; r.tau = rec(id)->tau; return r; };
    typedef __attribute__((address_space(1))) float GF32;
    auto tc = [&](int id) -> GF32* { return (GF32*)(Fb + ((unsigned)id << 2)); };
    auto exc_at = [&](unsigned h) -> GU64* { return (GU64*)(excb + ((size_t)h << 3)); };
    // tau (and the pinned flag) of an exceptional node of the compact field: rare (~0.02 % of the nodes, most of them
    // around the source), so the probe loop is only entered by the waves that meet one
    auto exc_lookup = [&](int id, bool* pinned) -> float {
        const unsigned mask = (1u << xlog) - 1u;
        unsigned h = exc_hash(id, xlog);
        for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
            const unsigned long long e = *exc_at(h);
            const int k = exc_key(e);
            if (k == -1) break;
            if ((k & 0x3fffffff) == id) { *pinned = (k & kExcPinned) != 0; return exc_tau(e); }
        }
        *pinned = false;
        return kInf;
    };
    // new or changed entry of a node this lane owns (never a pinned one); false: table full
    auto exc_upsert = [&](int id, float tau) -> bool {
        const unsigned mask = (1u << xlog) - 1u;
        unsigned h = exc_hash(id, xlog);
        const unsigned long long mine = exc_pack(id, tau);
        for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
            unsigned long long e = *exc_at(h);
            if (exc_key(e) == -1) {
                e = atomicCAS((unsigned long long*)exc_at(h), kExcEmpty, mine);
                if (e == kExcEmpty) return true;
            }
            if ((exc_key(e) & 0x3fffffff) == id) { *exc_at(h) = mine; return true; }
        }
        return false;
    };
    // tile -> (bx, bz) without an integer division: floor(t / nbz) = hi32(t * ceil(2^32 / nbz)) while t * nbz < 2^32
    const bool by_mul = nbz > 1 && (unsigned long long)ntile * (unsigned long long)nbz < (1ull << 32);
    const unsigned nbz_inv = by_mul ? 0xffffffffu / (unsigned)nbz + 1u : 0u;
    auto coords = [&](int id, int* iz0, int* ix0) {
        const unsigned tile = (unsigned)id >> 6;
        const unsigned bx = by_mul ? __umulhi(tile, nbz_inv) : tile / (unsigned)nbz;
        const unsigned bz = tile - bx * (unsigned)nbz;
        *ix0 = (int)(bx << kTileShift) + rec_ix_in_tile(id);
        *iz0 = (int)(bz << kTileShift) + rec_iz_in_tile(id);
    };

    bool dead = false;                                       // an error before the first round: skip the rounds, still hand the slot on
    int my_slot = -1;                                        // (the claimed field slot, when the launch recycles slots)
    if (COMPACT && ends) {
        // The field slot (FimEnds): wait for its previous user, then every node unreached, the exception table empty, and the nodes the
        // serial prologue pinned (window records) into both.  All by this workgroup, in this order.
        const FimEnds* const E = ends + blockIdx.x;
        if (E->slot_busy) {
            // Fewer field slots than units in the launch: thread 0 claims the first free one at or after blockIdx.x % pool by compare-and-swap
            // (as the bundle kernel does) and frees it at the very end.  No assumption about the order workgroups are dispatched in: a
            // workgroup only ever waits for RUNNING workgroups to finish (ADVICE r03 / VERDICT r03 item 5; before: "slot r % pool once
            // workgroup r - pool is done").  The wait is still bounded so that a corrupted flag array fails loudly (-3) instead of hanging.
            if (tid == 0) {
                const int P = E->nslots;
                int q = (int)(blockIdx.x % (unsigned)P), tries = 0;
                long long sweeps = 0;
                for (;;) {
                    // (look before the compare-and-swap: thousands of waiting workgroups polling with atomics slow the running ones down)
                    if (__hip_atomic_load(&E->slot_busy[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicCAS(&E->slot_busy[q], 0, 1) == 0) break;
                    q = q + 1 == P ? 0 : q + 1;
                    if (++tries >= P || tries >= 64) { tries = 0; if (++sweeps >= (1ll << 24)) { q = -1; break; } __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
                }
                __threadfence();                        // (what the slot's previous user wrote is behind us)
                sc[0] = q;
            }
            __syncthreads();
            my_slot = __builtin_amdgcn_readfirstlane(sc[0]);       // (uniform: the slot's arrays are addressed from scalar registers, as the assigned ones were)
            __syncthreads();
            if (my_slot < 0) { if (tid == 0) { p.info[2] = -3; p.info[0] = 0; } return; }
            Fb = (GChar*)(E->Tc_pool + (size_t)my_slot * ntile * kTileRecs);
            excb = (GChar*)(E->exc_pool + ((size_t)my_slot << xlog));
            maskb = (GChar*)(E->lists_pool + (size_t)my_slot * E->lists_stride);
        }
        typedef float __attribute__((ext_vector_type(4))) V4;
        typedef __attribute__((address_space(1))) V4 GV4;
        const V4 inf4 = { kInf, kInf, kInf, kInf };
        for (int i = tid; i < ntile * (kTileRecs / 4); i += NT) ((GV4*)Fb)[i] = inf4;
        for (int i = tid; i < (1 << xlog); i += NT) *exc_at((unsigned)i) = kExcEmpty;
        __threadfence_block();
        __syncthreads();
        typedef __attribute__((address_space(1))) const Rec GCRec;
        GCRec* const W = (GCRec*)E->W;
        const int cwz0 = E->cwz0, cwx0 = E->cwx0, cwnz = E->cwnz, nw = E->cwnx * cwnz;
        for (int q = tid; q < nw; q += NT) {
            const float wt = W[q].T, wk = W[q].tau;
            if (!t_pinned(wt)) continue;
            const int lx = q / cwnz, lz = q - lx * cwnz;
            const int id = rec_index(nbz, cwz0 + lz, cwx0 + lx);
            const unsigned long long mine = exc_pack(id | kExcPinned, wk);
            const unsigned mask = (1u << xlog) - 1u;
            unsigned h = exc_hash(id, xlog);
            bool placed = false;
            for (unsigned n = 0; n <= mask && !placed; ++n, h = (h + 1u) & mask)
                placed = atomicCAS((unsigned long long*)exc_at(h), kExcEmpty, mine) == kExcEmpty;      // (every node is inserted once: no key to match)
            if (!placed) { p.info[2] = -2; }
            *tc(id) = wt;                                   // -T: the sign bit marks the exceptional node
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i = tid; i < (ntile << (kMaskShift - 3)); i += NT) *(GU64*)(maskb + ((size_t)i << 3)) = 0ull;
    for (int i = tid; i < nwords; i += NT) tb[i] = 0u;
    if (tid == 0) {
        sc[SC_READY] = 0; sc[SC_READY_ODD] = 0; sc[SC_TMIN] = 0x7f800000; sc[SC_THETA] = 0x7f800000;
        sc[SC_FREEZE] = (int)0xff800000u; sc[SC_HASH] = 0; sc[SC_OVERFLOW] = 0; sc[SC_CUR] = 0;
    }
    __syncthreads();
    const int nseed = *p.seed_count;
    if (COMPACT && nseed > p.seed_cap) { if (tid == 0) p.info[2] = -1; dead = true; }   // (cannot happen: kSeedC covers the march window)
    else if (nseed <= p.seed_cap) {
        for (int i = tid; i < nseed; i += NT) {
            const int id = p.seed[i];
            if (!COMPACT) atomicAnd((unsigned*)&rec(id)->tau, ~kQueuedBit);
            atomicOr((unsigned long long*)mask_at(id >> 6), 1ull << (id & 63));
            atomicOr(&tb[(id >> 6) >> 5], 1u << ((id >> 6) & 31));
        }
    } else if (!COMPACT) {
        // more seeds than the prologue's list holds: they are flagged on the field (queued bit of tau)
        for (int id = tid; id < ntile * kTileRecs; id += NT) {
            const unsigned w = __hip_atomic_load((unsigned*)&rec(id)->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!(w & kQueuedBit)) continue;
            atomicAnd((unsigned*)&rec(id)->tau, ~kQueuedBit);
            if (t_pinned(rec(id)->T)) continue;
            atomicOr((unsigned long long*)mask_at(id >> 6), 1ull << (id & 63));
            atomicOr(&tb[(id >> 6) >> 5], 1u << ((id >> 6) & 31));
        }
    }
    __syncthreads();

    int rounds = 0, stall = 0, freezes = 0;
    unsigned hist[4] = { 1u, 2u, 3u, 4u };
    // Phase clocks and list statistics (thread 0) are probe instrumentation: seven 64-bit accumulators are fourteen VGPRs the
    // solver's live set does not have, so they exist only in builds with DSA_PHASE_CLOCKS (tools/perf_probe.py prints them).
#ifdef DSA_PHASE_CLOCKS
    int max_cnt = 0;
    unsigned long long tA = 0, tB0 = 0, tB1 = 0, tE = 0, t0 = wall_clock64(), sum_cnt = 0, sum_ready = 0;
#define DSA_PHASE(acc, extra) do { const unsigned long long t1 = wall_clock64(); acc += t1 - t0; t0 = t1; extra; } while (0)
#else
#define DSA_PHASE(acc, extra) do { } while (0)
#endif
    float best_tmin = -kInf;
    unsigned evals = 0, nchanged = 0;                        // per lane (a lane evaluates < 2^32 nodes)
#ifdef DSA_LEDGER
    unsigned dsa_lc[24] = {};                                // wave trips per marker (held by the lowest active lane of each trip)
#endif
    unsigned tie_n = 0;                                      // TIE: evaluations that ended on an exact tie with influence, and the largest influence
    float tie_max = 0.0f;
    unsigned tie_any = 0, tie_sum = 0;                       // ... with any influence at all, and their sum (kTieSumUnit): evaluations of the iteration, transient ties among them
#ifdef DSA_PASSA_CLOCKS
    unsigned long long sub[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tsub = wall_clock64();
#define DSA_TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1_ = wall_clock64(); sub[k] += t1_ - tsub; tsub = t1_; } while (0)
#else
#define DSA_TICK(k) do { } while (0)
#endif
    // Barrier probe (DSA_BARRIER_CLOCKS builds only): how long each wave sits at each of the round's four barriers, i.e. how uneven
    // the waves' shares of a phase are; summed over waves into clocks[0..3], thread 0's total into clocks[4].
#ifdef DSA_BARRIER_CLOCKS
    unsigned long long bwait[4] = { 0, 0, 0, 0 };
    const unsigned long long bstart = wall_clock64();
#define DSA_SYNC(k) do { const unsigned long long b0_ = wall_clock64(); __syncthreads(); bwait[k] += wall_clock64() - b0_; } while (0)
#else
#define DSA_SYNC(k) __syncthreads()
#endif
    // Issue priorities of a wave's phases (s_setprio; four waves share a SIMD's issue port): the phases that put loads in flight go
    // first, the solver -- several hundred VALU instructions that need nothing from memory -- yields to them, so another front's loads
    // are on their way while this one computes.  -3.9 % kernel time at full occupancy (profiles/r02_ab_priorities.txt; the inverse
    // assignment changes nothing, the variants around this one lie within 1 %).
#ifndef DSA_NO_PRIO
#define DSA_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define DSA_PRIO(n) do { } while (0)
#endif
#ifndef DSA_PRIO_A
#define DSA_PRIO_A 3     // pass A: record loads, neighbour loads
#define DSA_PRIO_R 2     // pass A: routing
#define DSA_PRIO_BL 3    // pass B: the neighbourhood's loads
#define DSA_PRIO_S 0     // pass B: solve_node
#define DSA_PRIO_W 1     // pass B: store, activation
#endif
    for (; !dead;) {
        DSA_LEDGER_COUNT(0, "round_head");
        const float theta = u2f((unsigned)sc[SC_THETA]);
        const bool open = !(theta < kInf);
        const float freeze = u2f((unsigned)sc[SC_FREEZE]);
        const bool frozen_any = freeze > -kInf;

#ifdef DSA_PASSA_CLOCKS
        tsub = wall_clock64();
#endif
        // ---- pass A: every wave sweeps its share of the tile bitmap (no workgroup barrier inside).  The active
        // tiles of the share are first collected in LDS, in tile order; then their node masks are fetched
        // with up to four loads in flight per lane, the set bits are expanded into a wave-local buffer in record
        // order, and lanes = nodes (four per lane, so sixteen loads in flight) compute the lower bounds and
        // route.  The pass costs two dependent memory round trips however the front is spread over the bitmap.
        int seen = 0;
        float tmin_lane = kInf;
        int* const nbuf = wbuf + wave * kWaveBuf;
        int* const tbuf = wtile + wave * kTileBuf;
        unsigned* const clr = wclr + wave * kTileBuf * kClrWords;
        constexpr int NW = NT / 64;
        constexpr int kQ = kTileBuf / 64, kI = kWaveBuf / 64;
#if DSA_KEY_MASKS
        // Tile record {E, O, R, DE, DO, round}: E / O = nodes activated by the even / odd half of the last round from a node whose new
        // acceptance time lay below that round's theta -- their lower bound is below this round's theta as well (theta does not go back),
        // so they are ready as they are: no coordinates, no neighbour loads; DE / DO = activated from a later node, or carried over: lower
        // bound from the four near neighbours, as before.  R filters what the even half activated at nodes the odd half of the same
        // round evaluated (both kinds).  Same ready sets as with one kind of mask, i.e. the same schedule; pass A only does less.
        auto sweep_tiles = [&](int ntiles) {
            DSA_TICK(0);
            DSA_PRIO(DSA_PRIO_A);
            DSA_LEDGER_COUNT(2, "tile_records");
            int tl[kQ];
            unsigned long long mN[kQ], mL[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                tl[q] = q * 64 + lane < ntiles ? tbuf[q * 64 + lane] : -1;
                mN[q] = 0ull; mL[q] = 0ull;
                if (tl[q] >= 0) {
                    GU64* const r8 = mask_at(tl[q]);
                    const unsigned long long E = r8[0], O = r8[1], R = r8[2], DE = r8[3], DO = r8[4];
                    const unsigned stamp = (unsigned)r8[5];
                    const unsigned long long Rf = stamp == (unsigned)rounds ? R : 0ull;
                    mN[q] = (E & ~Rf) | O;
                    mL[q] = ((DE & ~Rf) | DO) & ~mN[q];
                    if (frozen_any) { mL[q] |= mN[q]; mN[q] = 0ull; }      // a freeze horizon is up: every node shows its own acceptance time
                }
                if (q * 64 < ntiles)
                    for (int w4 = 0; w4 < kClrWords; ++w4) clr[kClrWords * (q * 64 + lane) + w4] = 0u;
                if (q * 64 < ntiles && tl[q] >= 0 && (mN[q] | mL[q]) == 0ull) atomicAnd(&tb[tl[q] >> 5], ~(1u << (tl[q] & 31)));      // the tile has drained
            }
            DSA_TICK(1);
            auto process = [&](const unsigned long long* m, auto tag) {
                constexpr bool LB = decltype(tag)::value;       // true: lower bounds from the neighbourhood; false: ready as they are
                if (LB) DSA_LEDGER_COUNT(4, "scan_L"); else DSA_LEDGER_COUNT(3, "scan_N");
                int off[kQ], total = 0;
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    off[q] = total;
                    if (q * 64 >= ntiles) continue;                           // wave-uniform: no tiles in this group
                    const int n = __popcll(m[q]);
                    const int incl = wave_scan_incl(n);
                    off[q] = total + incl - n;
                    total += wave_last(incl);
                }
                seen += total;
                for (int base = 0; base < total; base += kWaveBuf) {
#pragma unroll
                    for (int q = 0; q < kQ; ++q) {
                        unsigned long long mm = m[q];
                        int idx = off[q];
                        while (mm) {
                            DSA_LEDGER_COUNT(5, "expand_bit");
                            const int nb = __ffsll((long long)mm) - 1;
                            mm &= mm - 1ull;
                            if (idx >= base && idx < base + kWaveBuf) nbuf[idx - base] = ((q * 64 + lane) << 6) + nb;     // (tile slot, node)
                            ++idx;
                        }
                    }
                    const int nn = min(total - base, kWaveBuf);
                    DSA_TICK(2);
                    DSA_PRIO(DSA_PRIO_A);
                    if (LB) DSA_LEDGER_COUNT(7, "nodes_L_window"); else DSA_LEDGER_COUNT(6, "nodes_N_window");
                    int id[kI], par[kI], slot[kI];
                    float lb[kI], own[kI];
#pragma unroll
                    for (int i = 0; i < kI; ++i) {
                        id[i] = -1; par[i] = 0; slot[i] = 0; lb[i] = kInf; own[i] = kInf;
                        if (i * 64 >= nn) continue;                           // wave-uniform: this group of 64 is empty
                        const bool have = i * 64 + lane < nn;
                        const int e = have ? nbuf[i * 64 + lane] : 0;
                        slot[i] = e >> 6;
                        id[i] = have ? (tbuf[slot[i]] << 6) + (e & 63) : -1;
                        par[i] = ((e >> 3) ^ e) & 1;                           // tiles start at even coordinates: the parity of (ix + iz) is that of the node inside its tile
                        if (!LB) { lb[i] = -kInf; continue; }
                        DSA_LEDGER_COUNT(8, "nodes_L_group64");
                        if (have) {
                            int iz, ix;
                            coords(id[i], &iz, &ix);
                            int nid[8];
                            rec_stencil(nbz, id[i], nid);
                            float a, b2, c2, d2;
                            if (COMPACT) {
                                a = ix > 0 ? *tc(nid[0]) : kInf; b2 = ix + 1 < nnx ? *tc(nid[1]) : kInf;
                                c2 = iz > 0 ? *tc(nid[2]) : kInf; d2 = iz + 1 < nnz ? *tc(nid[3]) : kInf;
                                if (frozen_any) own[i] = *tc(id[i]);
                            } else {
                                a = ix > 0 ? rec(nid[0])->tau : kInf; b2 = ix + 1 < nnx ? rec(nid[1])->tau : kInf;
                                c2 = iz > 0 ? rec(nid[2])->tau : kInf; d2 = iz + 1 < nnz ? rec(nid[3])->tau : kInf;
                                if (frozen_any) own[i] = rec(id[i])->tau;
                            }
                            if (COMPACT && (__builtin_signbit(a) || __builtin_signbit(b2) || __builtin_signbit(c2) || __builtin_signbit(d2) || __builtin_signbit(own[i]))) {
                                bool pin;
                                if (__builtin_signbit(a)) a = exc_lookup(nid[0], &pin);
                                if (__builtin_signbit(b2)) b2 = exc_lookup(nid[1], &pin);
                                if (__builtin_signbit(c2)) c2 = exc_lookup(nid[2], &pin);
                                if (__builtin_signbit(d2)) d2 = exc_lookup(nid[3], &pin);
                                if (__builtin_signbit(own[i])) own[i] = exc_lookup(id[i], &pin);
                            }
                            lb[i] = fminf(fminf(a, b2), fminf(c2, d2));
                        }
                    }
                    DSA_TICK(3);
                    DSA_PRIO(DSA_PRIO_R);
                    DSA_LEDGER_COUNT(9, "routing_window");
                    // routing: one slot allocation per colour for the whole window
                    unsigned long long be[kI], bo[kI];
                    bool frozen[kI];
                    int ne = 0, no = 0;
#pragma unroll
                    for (int i = 0; i < kI; ++i) {
                        be[i] = 0ull; bo[i] = 0ull; frozen[i] = false;
                        if (i * 64 >= nn) continue;
                        const bool have = id[i] >= 0;
                        frozen[i] = LB && have && frozen_any && own[i] < freeze;
                        const bool want = have && !frozen[i] && (!LB || open || lb[i] < theta);
                        be[i] = __ballot(want && par[i] == 0);
                        bo[i] = __ballot(want && par[i] != 0);
                        ne += __popcll(be[i]); no += __popcll(bo[i]);
                    }
                    int base_e = 0, base_o = 0;
                    if (lane == 0) {
                        if (ne) base_e = atomicAdd(&sc[SC_READY], ne);
                        if (no) base_o = atomicAdd(&sc[SC_READY_ODD], no);
                    }
                    base_e = __builtin_amdgcn_readfirstlane(base_e);
                    base_o = __builtin_amdgcn_readfirstlane(base_o);
                    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                    for (int i = 0; i < kI; ++i) {
                        if (i * 64 >= nn) continue;
                        const bool have = id[i] >= 0;
                        const bool want_e = (be[i] >> lane) & 1ull, want_o = (bo[i] >> lane) & 1ull;
                        const int pe = base_e + __popcll(be[i] & below), po = base_o + __popcll(bo[i] & below);
                        base_e += __popcll(be[i]); base_o += __popcll(bo[i]);
                        const bool got = (want_e && pe < rhalf) || (want_o && po < rhalf);
                        if (got) ready[want_o ? rhalf + po : pe] = id[i];
                        if ((got && !want_o) || frozen[i]) atomicOr(&clr[kClrWords * slot[i] + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));
                        if (got && want_o) atomicOr(&clr[kClrWords * slot[i] + 2 + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));     // evaluated by this round's odd half
                        // what stays behind bounds the next theta from below: its lower bound -- or, for a ready node the lists had no room for, theta itself
                        if (have && !frozen[i] && !got) tmin_lane = fminf(tmin_lane, LB ? lb[i] : theta);
                    }
                    DSA_TICK(4);
                }
            };
            process(mN, std::false_type{});
            process(mL, std::true_type{});
            DSA_LEDGER_COUNT(16, "record_rewrite");
            // the tile's record for the next round: what stays queued needs its lower bound again (DO), nothing new yet, and who is evaluated by
            // this round's odd half; plain stores -- a tile has one owner in pass A and nobody activates then
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (tl[q] >= 0) {
                    const unsigned* const cw = clr + kClrWords * (q * 64 + lane);
                    const unsigned long long c = (unsigned long long)cw[0] | ((unsigned long long)cw[1] << 32);
                    const unsigned long long ro = (unsigned long long)cw[2] | ((unsigned long long)cw[3] << 32);
                    GU64* const r8 = mask_at(tl[q]);
                    r8[0] = 0ull; r8[1] = 0ull; r8[2] = ro; r8[3] = 0ull; r8[4] = (mN[q] | mL[q]) & ~c & ~ro; r8[5] = (unsigned long long)(unsigned)(rounds + 1);
                }
        };
#else
        auto sweep_tiles = [&](int ntiles) {
            DSA_TICK(0);
            DSA_PRIO(DSA_PRIO_A);
            DSA_LEDGER_COUNT(2, "tile_records");
            int tl[kQ];
            unsigned long long m[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                tl[q] = q * 64 + lane < ntiles ? tbuf[q * 64 + lane] : -1;
                if (kOddR) {
                    m[q] = 0ull;
                    if (tl[q] >= 0) {
                        GU64* const rec4 = mask_at(tl[q]);
                        const unsigned long long E = rec4[0], O = rec4[1], R = rec4[2];
                        const unsigned stamp = (unsigned)rec4[3];
                        m[q] = (E & ~(stamp == (unsigned)rounds ? R : 0ull)) | O;
                    }
                } else m[q] = tl[q] >= 0 ? *mask_at(tl[q]) : 0ull;
                if (q * 64 < ntiles)
                    for (int w4 = 0; w4 < kClrWords; ++w4) clr[kClrWords * (q * 64 + lane) + w4] = 0u;
            }
            DSA_TICK(1);
            DSA_LEDGER_COUNT(3, "scan");
            int off[kQ], total = 0;
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                off[q] = total;
                if (q * 64 >= ntiles) continue;                           // wave-uniform: no tiles in this group
                if (tl[q] >= 0 && m[q] == 0ull) atomicAnd(&tb[tl[q] >> 5], ~(1u << (tl[q] & 31)));      // the tile has drained
                const int n = __popcll(m[q]);
                const int incl = wave_scan_incl(n);
                off[q] = total + incl - n;
                total += wave_last(incl);
            }
            seen += total;
            for (int base = 0; base < total; base += kWaveBuf) {
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    unsigned long long mm = m[q];
                    int idx = off[q];
                    while (mm) {
                        DSA_LEDGER_COUNT(5, "expand_bit");
                        const int nb = __ffsll((long long)mm) - 1;
                        mm &= mm - 1ull;
                        if (idx >= base && idx < base + kWaveBuf) nbuf[idx - base] = ((q * 64 + lane) << 6) + nb;     // (tile slot, node)
                        ++idx;
                    }
                }
                const int nn = min(total - base, kWaveBuf);
                DSA_TICK(2);
                DSA_PRIO(DSA_PRIO_A);
                DSA_LEDGER_COUNT(7, "nodes_window");
                int id[kI], par[kI], slot[kI];
                float lb[kI], own[kI];
#pragma unroll
                for (int i = 0; i < kI; ++i) {
                    id[i] = -1; par[i] = 0; slot[i] = 0; lb[i] = kInf; own[i] = kInf;
                    if (i * 64 >= nn) continue;                           // wave-uniform: this group of 64 is empty
                    DSA_LEDGER_COUNT(8, "nodes_group64");
                    const bool have = i * 64 + lane < nn;
                    const int e = have ? nbuf[i * 64 + lane] : 0;
                    slot[i] = e >> 6;
                    id[i] = have ? (tbuf[slot[i]] << 6) + (e & 63) : -1;
                    int iz, ix;
                    coords(have ? id[i] : 0, &iz, &ix);
                    par[i] = (ix + iz) & 1;
                    lb[i] = kInf; own[i] = kInf;
                    if (have) {
                        int nid[8];
                        rec_stencil(nbz, id[i], nid);
                        float a, b2, c2, d2;
                        if (COMPACT) {
                            a = ix > 0 ? *tc(nid[0]) : kInf; b2 = ix + 1 < nnx ? *tc(nid[1]) : kInf;
                            c2 = iz > 0 ? *tc(nid[2]) : kInf; d2 = iz + 1 < nnz ? *tc(nid[3]) : kInf;
                            if (frozen_any) own[i] = *tc(id[i]);
                        } else {
                            a = ix > 0 ? rec(nid[0])->tau : kInf; b2 = ix + 1 < nnx ? rec(nid[1])->tau : kInf;
                            c2 = iz > 0 ? rec(nid[2])->tau : kInf; d2 = iz + 1 < nnz ? rec(nid[3])->tau : kInf;
                            if (frozen_any) own[i] = rec(id[i])->tau;
                        }
                        if (COMPACT && (__builtin_signbit(a) || __builtin_signbit(b2) || __builtin_signbit(c2) || __builtin_signbit(d2) || __builtin_signbit(own[i]))) {
                            bool pin;
                            if (__builtin_signbit(a)) a = exc_lookup(nid[0], &pin);
                            if (__builtin_signbit(b2)) b2 = exc_lookup(nid[1], &pin);
                            if (__builtin_signbit(c2)) c2 = exc_lookup(nid[2], &pin);
                            if (__builtin_signbit(d2)) d2 = exc_lookup(nid[3], &pin);
                            if (__builtin_signbit(own[i])) own[i] = exc_lookup(id[i], &pin);
                        }
                        lb[i] = fminf(fminf(a, b2), fminf(c2, d2));
                    }
                }
                DSA_TICK(3);
                DSA_PRIO(DSA_PRIO_R);
                DSA_LEDGER_COUNT(9, "routing_window");
                // routing: one slot allocation per colour for the whole window
                unsigned long long be[kI], bo[kI];
                bool frozen[kI];
                int ne = 0, no = 0;
#pragma unroll
                for (int i = 0; i < kI; ++i) {
                    be[i] = 0ull; bo[i] = 0ull; frozen[i] = false;
                    if (i * 64 >= nn) continue;
                    const bool have = id[i] >= 0;
                    frozen[i] = have && frozen_any && own[i] < freeze;
                    const bool want = have && !frozen[i] && (open || lb[i] < theta);
                    be[i] = __ballot(want && par[i] == 0);
                    bo[i] = __ballot(want && par[i] != 0);
                    ne += __popcll(be[i]); no += __popcll(bo[i]);
                }
                int base_e = 0, base_o = 0;
                if (lane == 0) {
                    if (ne) base_e = atomicAdd(&sc[SC_READY], ne);
                    if (no) base_o = atomicAdd(&sc[SC_READY_ODD], no);
                }
                base_e = __builtin_amdgcn_readfirstlane(base_e);
                base_o = __builtin_amdgcn_readfirstlane(base_o);
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int i = 0; i < kI; ++i) {
                    if (i * 64 >= nn) continue;
                    const bool have = id[i] >= 0;
                    const bool want_e = (be[i] >> lane) & 1ull, want_o = (bo[i] >> lane) & 1ull;
                    const int pe = base_e + __popcll(be[i] & below), po = base_o + __popcll(bo[i] & below);
                    base_e += __popcll(be[i]); base_o += __popcll(bo[i]);
                    const bool got = (want_e && pe < rhalf) || (want_o && po < rhalf);
                    if (got) ready[want_o ? rhalf + po : pe] = id[i];
                    if ((got && !(kOddR && want_o)) || frozen[i]) atomicOr(&clr[kClrWords * slot[i] + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));
                    if (kOddR && got && want_o) atomicOr(&clr[kClrWords * slot[i] + 2 + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));     // evaluated by this round's odd half
                    if (have && !frozen[i] && !got) tmin_lane = fminf(tmin_lane, lb[i]);
                }
                DSA_TICK(4);
            }
            // leaving the active set, one atomic per tile; it happens before the barrier, so a change that lands
            // while a node is being evaluated sets its bit again
            DSA_LEDGER_COUNT(16, "record_rewrite");
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (tl[q] >= 0) {
                    const unsigned* const cw = clr + kClrWords * (q * 64 + lane);
                    const unsigned long long c = (unsigned long long)cw[0] | ((unsigned long long)cw[1] << 32);
                    if (kOddR) {
                        // the tile's record for the next round: what stays queued, nothing from an odd half yet, and who is evaluated by this round's odd half
                        const unsigned long long ro = (unsigned long long)cw[2] | ((unsigned long long)cw[3] << 32);
                        GU64* const rec4 = mask_at(tl[q]);
                        rec4[0] = m[q] & ~c & ~ro; rec4[1] = 0ull; rec4[2] = ro; rec4[3] = (unsigned long long)(unsigned)(rounds + 1);
                    } else if (c) atomicAnd((unsigned long long*)mask_at(tl[q]), ~c);
                }
        };
#endif
        int ntw = 0;                                                               // tiles collected, wave-uniform
        // 64 bitmap words of this wave per trip: groups of 2^gs consecutive words dealt round-robin to the waves,
        // ascending (16-word groups on large grids: -2.5 % against single words at 1025^2; single words on the small
        // refined grids, whose whole bitmap is a dozen words)
        // Round 2: a group is about one tile column (nbz / 32 words: 4 at 1025^2, 16 at 4097^2), so that a front, which crosses every
        // column, spreads evenly over the waves (the wait at the pass's barrier was a seventh of the round with 16-word groups at
        // 1025^2: -2 % time at full occupancy, -7 % at half, profiles/r02_ab_group_shift.txt).
        const int colw = nbz >> 5;
        const int gs_col = colw >= 16 ? 4 : colw >= 8 ? 3 : colw >= 4 ? 2 : colw >= 2 ? 1 : 0;
        const int gs = nwords >= 64 * NW ? (DSA_FIM_GROUP_SHIFT >= 0 ? DSA_FIM_GROUP_SHIFT : gs_col) : 0;
        // (one call site of sweep_tiles: its body is three thousand instructions, and two inlined copies were what the loop used to hold)
        {
            int wb = 0, tbase = 0, ttotal = 0, toff = 0, w = 0;
            unsigned bits = 0u;
            bool words_left = true;
            while (words_left || ntw) {
                while (words_left && ntw < kTileBuf) {
                    if (tbase >= ttotal) {                                         // next 64 words of this wave
                        if (!(((wb * (64 >> gs) * NW + wave) << gs) < nwords)) { words_left = false; break; }
                        DSA_LEDGER_COUNT(1, "bitmap_words");
                        w = (((wb * (64 >> gs) + (lane >> gs)) * NW + wave) << gs) + (lane & ((1 << gs) - 1));
                        bits = w < nwords ? tb[w] : 0u;
                        const int nt_lane = __popc(bits);
                        const int tincl = wave_scan_incl(nt_lane);
                        ttotal = wave_last(tincl);
                        toff = tincl - nt_lane;
                        tbase = 0;
                        ++wb;
                        if (ttotal == 0) continue;
                    }
                    const int take = min(kTileBuf - ntw, ttotal - tbase);
                    unsigned bb = bits;
                    int idx = toff;
                    while (bb) {
                        const int b2 = __ffs((int)bb) - 1;
                        bb &= bb - 1u;
                        if (idx >= tbase && idx < tbase + take) tbuf[ntw + idx - tbase] = (w << 5) + b2;
                        ++idx;
                    }
                    ntw += take; tbase += take;
                }
                if (ntw) { sweep_tiles(ntw); ntw = 0; }
            }
        }
        DSA_TICK(0);
        tmin_lane = wave_min(tmin_lane);
        if (lane == 0) {
            if (tmin_lane < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(tmin_lane));
            if (seen) atomicAdd(&sc[SC_CUR], seen);
        }
        DSA_SYNC(0);
        const int cnt = sc[SC_CUR];
        if (cnt == 0) break;
        DSA_PHASE(tA, sum_cnt += cnt; if (cnt > max_cnt) max_cnt = cnt);

        // ---- pass B: evaluate, even nodes first
        const int nready_even = sc[SC_READY] < rhalf ? sc[SC_READY] : rhalf;
        const int nready_odd = sc[SC_READY_ODD] < rhalf ? sc[SC_READY_ODD] : rhalf;
        unsigned hv_lane = 0u;
        float kmin_lane = kInf;
        for (int half = 0; half < 2; ++half) {
            const int nready = half ? nready_odd : nready_even;
            for (int j0 = 0; j0 < nready; j0 += NT) {
#ifdef DSA_PASSA_CLOCKS
                tsub = wall_clock64();
#endif
                const int j = j0 + tid;
                const bool act = j < nready;
                if (j0 + (tid & ~63) >= nready) continue;                 // wave-uniform: none of this wave's 64 slots holds a node
                const int id = act ? ready[half ? rhalf + j : j] : 0;
                int iz, ix;
                DSA_PRIO(DSA_PRIO_BL);
                DSA_LEDGER_COUNT(17, "passB_loads");
                coords(id, &iz, &ix);
                Hood h;
                h.in[0] = act && ix > 0;          h.in_outer[0] = act && ix > 1;
                h.in[1] = act && ix + 1 < nnx;    h.in_outer[1] = act && ix + 2 < nnx;
                h.in[2] = act && iz > 0;          h.in_outer[2] = act && iz > 1;
                h.in[3] = act && iz + 1 < nnz;    h.in_outer[3] = act && iz + 2 < nnz;
                int nid[8];
                rec_stencil(nbz, id, nid);
                Rec own;
                if (COMPACT) {
                    bool flagged = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a = h.in[q] ? *tc(nid[q]) : kInf;
                        const float b = h.in_outer[q] ? *tc(nid[4 + q]) : kInf;
                        h.near_[q] = a; h.near_tau[q] = a; h.outer[q] = b; h.outer_tau[q] = b;
                        flagged = flagged || __builtin_signbit(a) || __builtin_signbit(b);
                    }
                    const float vo = act ? *tc(id) : 0.0f;
                    own = Rec{ vo, vo };
                    if (!act) own.T = -1.0f;                              // inactive lanes read as pinned
                    flagged = flagged || (act && __builtin_signbit(vo));
                    if (flagged) {                                       // exceptional nodes in the neighbourhood: tau (and pinned) from the table
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bool pin;
                            if (__builtin_signbit(h.near_[q])) { const float v = h.near_[q]; h.near_tau[q] = exc_lookup(nid[q], &pin); h.near_[q] = pin ? v : -v; }
                            if (__builtin_signbit(h.outer[q])) { const float v = h.outer[q]; h.outer_tau[q] = exc_lookup(nid[4 + q], &pin); h.outer[q] = pin ? v : -v; }
                        }
                        if (act && __builtin_signbit(vo)) { bool pin; own.tau = exc_lookup(id, &pin); own.T = pin ? vo : -vo; }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const Rec a = h.in[q] ? ld(nid[q]) : Rec{ kInf, kInf };
                        const Rec b = h.in_outer[q] ? ld(nid[4 + q]) : Rec{ kInf, kInf };
                        h.near_[q] = a.T; h.near_tau[q] = a.tau;
                        h.outer[q] = b.T; h.outer_tau[q] = b.tau;
                    }
                    own = act ? ld(id) : Rec{ -1.0f, 0.0f };
                }
                DSA_TICK(5);
                DSA_PRIO(DSA_PRIO_S);
                const float t_old = own.T;
                const float k_old = own.tau;
                bool changed = false;
                float c = 0.0f, k = kInf;
                if (!t_pinned(t_old)) {
                    const NodeGeom geom = { p.ri, risti[ix], p.dnx, p.dnz };
                    if (TIE) {
                        float tie;
                        c = solve_node_t<true>(h, slow_at(id), geom, &k, &tie DSA_LEDGER_PASS);
                        if (tie > p.tie_threshold) { ++tie_n; tie_max = fmaxf(tie_max, tie); }
                        if (tie > 0.0f) { ++tie_any; tie_sum += (unsigned)(fminf(tie, 1.0f) * (1.0f / kTieSumUnit)); }
                    } else c = solve_node_t<false>(h, slow_at(id), geom, &k, nullptr DSA_LEDGER_PASS);
#ifdef DSA_PROBE_EXTRA_READ
                    if (COMPACT) {   // bandwidth probe: one more cold line per evaluated node group (the slowness half a grid away); result unused
                        const float extra = slow_at(id < 524288 ? id + 524288 : id - 524288);
                        if (extra == 12345.678f) p.info[5] = 1;
                    }
#endif
#ifdef DSA_PROBE_SOLVE2
                    {   // instruction-count probe: the solver once more on (opaquely) the same inputs; SQ_INSTS_VALU difference = its share
                        Hood h2 = h;
                        asm volatile("" : "+v"(h2.near_[0]), "+v"(h2.near_tau[1]), "+v"(h2.outer[2]));
                        float k2;
                        const float c2 = solve_node(h2, slow_at(id), geom, &k2);
                        if (f2u(c2) != f2u(c) || f2u(k2) != f2u(k)) p.info[5] = 1;
                    }
#endif
                    ++evals;
                    changed = f2u(c) != f2u(t_old) || f2u(k) != f2u(k_old);
                }
                DSA_TICK(6);
                DSA_PRIO(DSA_PRIO_W);
                DSA_LEDGER_COUNT(18, "store_activate");
                if (changed) {
                    if (COMPACT) {
                        // causal node (the rule): one float.  Else the table entry first, then the flagged value.
                        if (f2u(c) == f2u(k)) *tc(id) = c;
                        else { if (!exc_upsert(id, k)) { p.info[2] = -2; sc[SC_OVERFLOW] = 1; } *tc(id) = -c; }
                    } else { rec(id)->T = c; rec(id)->tau = k; }
                    ++nchanged;
                }
                // dependents: same pruning as k_fim; the tile mask's old value tells whether the node was
                // already active and whether its tile has to enter the bitmap
                const float t_lo = fminf(t_value(t_old), c), k_lo = fminf(k_old, k);
                bool want[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ky = h.near_tau[q];
                    want[q] = changed && h.in[q] && !t_pinned(h.near_[q]) && k_lo <= ky;
                    want[4 + q] = changed && h.in_outer[q] && ky < kInf && !t_pinned(h.outer[q]) &&
                                  t_value(h.near_[q]) > t_lo && k_lo < h.outer_tau[q];
                }
                // The eight dependents lie in the node's own tile and in at most one other tile per direction (when the near one
                // has left the tile, the outer one is in the same neighbour tile): five atomics at most.  Their mask bits are
                // constant shifts of the node's own bit b = 1 << r (record r = 8 (ix & 7) + (iz & 7)): a step in x moves a
                // whole byte, a step in z one bit inside the byte; what a shift pushes out of the byte / word is exactly the
                // part that belongs to the neighbouring tile, where it reappears shifted the other way.
                const int own_tile = id >> 6;
                const unsigned long long b = 1ull << (id & 63);
                auto sel = [](bool w, unsigned long long v) -> unsigned long long { return w ? v : 0ull; };
                unsigned long long own_bits, fxm, fxp, fzm, fzp;
                own_bits = sel(want[0], b >> 8) | sel(want[4], b >> 16) | sel(want[1], b << 8) | sel(want[5], b << 16) |
                           sel(want[2], (b >> 1) & 0x7f7f7f7f7f7f7f7full) | sel(want[6], (b >> 2) & 0x3f3f3f3f3f3f3f3full) |
                           sel(want[3], (b << 1) & 0xfefefefefefefefeull) | sel(want[7], (b << 2) & 0xfcfcfcfcfcfcfcfcull);
                fxm = sel(want[0], b << 56) | sel(want[4], b << 48);
                fxp = sel(want[1], b >> 56) | sel(want[5], b >> 48);
                fzm = sel(want[2], (b << 7) & 0x8080808080808080ull) | sel(want[6], (b << 6) & 0xc0c0c0c0c0c0c0c0ull);
                fzp = sel(want[3], (b >> 7) & 0x0101010101010101ull) | sel(want[7], (b >> 6) & 0x0303030303030303ull);
                // which mask: activated from a node accepted inside this round's window (ready without a lower bound) or from a later one
                const int mword = kKeyMasks ? ((k < theta) ? half : 3 + half) : (kOddR ? half : 0);
                auto activate = [&](int tile, unsigned long long bits) {
                    if (bits) {
                        atomicOr((unsigned long long*)(mask_at(tile) + mword), bits);
                        atomicOr(&tb[tile >> 5], 1u << (tile & 31));
                    }
                };
                activate(own_tile - nbz, fxm);
                activate(own_tile + nbz, fxp);
                activate(own_tile - 1, fzm);
                activate(own_tile + 1, fzp);
                activate(own_tile, own_bits);
                // change hash and earliest change: accumulated per lane, reduced once per round (below)
                if (changed) { hv_lane += ((unsigned)id * 2654435761u) ^ (f2u(c) * 40503u) ^ (f2u(k) * 2246822519u); kmin_lane = fminf(kmin_lane, k); }
                DSA_TICK(7);
            }
            if (half == 1) {
                const unsigned hv = wave_sum(hv_lane);
                const float kmin = wave_min(kmin_lane);
                if (lane == 0) {
                    if (hv) atomicAdd(reinterpret_cast<unsigned*>(&sc[SC_HASH]), hv);
                    if (kmin < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[SC_TMIN]), f2u(kmin));
                }
            }
            if (half) DSA_SYNC(2); else DSA_SYNC(1);
            DSA_PHASE((half ? tB1 : tB0), );
        }
#ifdef DSA_PHASE_CLOCKS
        sum_ready += nready_even + nready_odd;
#endif
        DSA_LEDGER_COUNT(19, "round_end");
        if (tid == 0) {
            sc[SC_READY] = 0; sc[SC_READY_ODD] = 0; sc[SC_CUR] = 0;
            const float tmin = u2f((unsigned)sc[SC_TMIN]);
            sc[SC_THETA] = (int)f2u(tmin + p.window);
            sc[SC_TMIN] = 0x7f800000;
            const unsigned hsh = (unsigned)sc[SC_HASH];
            sc[SC_HASH] = 0;
            if (tmin > best_tmin) best_tmin = tmin;
            const bool repeat = hsh != 0u && (hsh == hist[1] || hsh == hist[2] || hsh == hist[3] || hsh == hist[0]);
            hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = hsh;
            if (repeat) { if (++stall >= kCycleRounds) { sc[SC_FREEZE] = (int)f2u(best_tmin + p.window); stall = 0; ++freezes; } }
            else stall = 0;
        }
        ++rounds;
        DSA_SYNC(3);
        DSA_PHASE(tE, );
        if (COMPACT && sc[SC_OVERFLOW]) break;          // the exception table is full (info[2] = -2): the host grows it and solves the chunk again
        if (rounds > p.max_rounds) { if (tid == 0 && p.info[2] != -2) p.info[2] = -1; break; }
    }
    DSA_LEDGER_COUNT(20, "kernel_tail");
    {
        unsigned long long e64 = evals, c64 = nchanged;
        for (int o = 32; o > 0; o >>= 1) { e64 += __shfl_xor(e64, o); c64 += __shfl_xor(c64, o); }
        if (lane == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 4), e64); atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 6), c64); }
    }
#ifdef DSA_LEDGER
    if (p.clocks) {
#pragma unroll
        for (int q = 0; q < 24; ++q) { const unsigned v = wave_sum(dsa_lc[q]); if (lane == 0 && v) atomicAdd(p.clocks + 8 + q, (unsigned long long)v); }
    }
#endif
    if (TIE && p.tie) {
        const unsigned tn = wave_sum(tie_n);
        const float tm = -wave_min(-tie_max);
        if (lane == 0 && tn) { atomicAdd((unsigned*)p.tie, tn); atomicMax((unsigned*)p.tie + 1, f2u(tm)); }
        const unsigned ta = wave_sum(tie_any), ts = wave_sum(tie_sum);
        if (lane == 0 && ta) { atomicAdd((unsigned*)p.tie + 2, ta); atomicAdd((unsigned*)p.tie + 3, ts); }
    }
#ifdef DSA_BARRIER_CLOCKS
    if (p.clocks && lane == 0) {
        for (int q = 0; q < 4; ++q) atomicAdd(p.clocks + q, bwait[q]);
        if (tid == 0) p.clocks[4] = wall_clock64() - bstart;
    }
#endif
    if (COMPACT && ends) {
        // the unit's receiver times from its own field (reference srtimes), then the slot goes to the next unit
        const FimEnds* const E = ends + blockIdx.x;
        if (E->rays) {
            __threadfence_block();
            __syncthreads();
            const GridDesc g = E->g;
            for (int r = tid; r < E->nrays; r += NT) {
                const RayDesc rd = E->rays[r];
                if (!(rd.flags & kRayTime)) continue;
                float t;
                if (!receiver_time(g, E->scx, E->scz, rd, (const float*)Fb, E->veln, E->dpl, &t)) atomicExch(E->err, E->ray0 + r + 1);
                E->out[rd.data] = t;
            }
        }
        if (E->slot_busy) {
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&E->slot_busy[my_slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        p.info[0] = rounds; p.info[1] = 0; p.info[3] = freezes;
#ifdef DSA_PHASE_CLOCKS
        if (p.clocks) { p.clocks[0] = tA; p.clocks[1] = tB0; p.clocks[2] = tB1; p.clocks[3] = tE; p.clocks[4] = sum_cnt; p.clocks[5] = sum_ready; p.clocks[6] = (unsigned long long)max_cnt; }
#endif
#ifdef DSA_PASSA_CLOCKS
        if (p.clocks) { p.clocks[7] = sub[2]; p.clocks[0] = sub[0] + sub[1]; p.clocks[1] = sub[2] + sub[3]; p.clocks[2] = sub[4]; p.clocks[3] = sub[5]; p.clocks[4] = sub[6]; p.clocks[5] = sub[7]; }
#endif
    }
}

size_t fim_lds_bytes(const FimLaunch& l) { return (l.sorted ? (size_t)l.tile_words * 4 : 0) + (size_t)l.lds_pad; }


void launch_fim(const FimProblem* d_problems, int nproblems, const FimLaunch& l, hipStream_t stream, const FimEnds* d_ends)
{
    if (nproblems <= 0) return;
    if (l.sorted) {
        const size_t lds = fim_lds_bytes(l);
#define DSA_LAUNCH_SORTED_T(NT, C, T) hipLaunchKernelGGL((k_fim_sorted<NT, C, T>), dim3(nproblems), dim3(NT), lds, stream, d_problems, l.list_cap, l.ready_cap, d_ends)
#define DSA_LAUNCH_SORTED(NT, C) do { if (l.tie) DSA_LAUNCH_SORTED_T(NT, C, true); else DSA_LAUNCH_SORTED_T(NT, C, false); } while (0)
        if (l.compact) {
            if (l.threads == 128) DSA_LAUNCH_SORTED(128, true); else if (l.threads == 256) DSA_LAUNCH_SORTED(256, true);
            else if (l.threads == 512) DSA_LAUNCH_SORTED(512, true); else DSA_LAUNCH_SORTED(1024, true);
        } else {
            if (l.threads == 128) DSA_LAUNCH_SORTED(128, false); else if (l.threads == 256) DSA_LAUNCH_SORTED(256, false);
            else if (l.threads == 512) DSA_LAUNCH_SORTED(512, false); else DSA_LAUNCH_SORTED(1024, false);
        }
#undef DSA_LAUNCH_SORTED
#undef DSA_LAUNCH_SORTED_T
        return;
    }
    const size_t pad = (size_t)l.lds_pad;
    if (l.threads == 128) hipLaunchKernelGGL(k_fim<128>, dim3(nproblems), dim3(128), pad, stream, d_problems, l.list_cap, l.ready_cap);
    else if (l.threads == 256) hipLaunchKernelGGL(k_fim<256>, dim3(nproblems), dim3(256), pad, stream, d_problems, l.list_cap, l.ready_cap);
    else if (l.threads == 512) hipLaunchKernelGGL(k_fim<512>, dim3(nproblems), dim3(512), pad, stream, d_problems, l.list_cap, l.ready_cap);
    else hipLaunchKernelGGL(k_fim<1024>, dim3(nproblems), dim3(1024), pad, stream, d_problems, l.list_cap, l.ready_cap);
}

}  // namespace dsa
