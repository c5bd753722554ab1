// Coarse fixed-point solve of a BUNDLE for gfx950 (CDNA4): the G periods of one source under one shared round schedule
// (kernels.h: FimBundle).  Same fixed point per member as k_fim_sorted (fim_kernel.hip: the replacement for the reference's narrow-band
// march `travel` / `fouds2` + binary tree, CalSurfG.f90:288-487, :587-759, :768-921); what changes is what a round costs:
//
//   * pass A (the tile sweep, the expansion of the node masks, lower bounds, routing) runs ONCE per bundle on the pilot's times
//     (member 0) instead of once per unit -- it was 41 % of the solo kernel's vector instructions (the ledger);
//   * pass B evaluates a ready node for all members: a wave takes 64 / (G/4) nodes, G/4 lanes per node, four members per lane.  The
//     node's coordinates, stencil addresses and grid flags are computed once per lane and serve four evaluations; the nine field values of
//     four members come in nine 16-byte loads (member-minor field: a node's G values are one contiguous segment, so a fetched line carries
//     nothing that is not used); pruning and the activation masks are formed once per node from the members' OR;
//   * the four barriers and the dependent memory round trips of a round are shared by G units.
//
// Schedule: routed by the pilot alone -- a listed node is ready when the pilot's lower bound lies inside the pilot's window (or the pilot
// is pinned there: the members' march windows differ by a few nodes around the source).  A member whose front runs differently from the
// pilot's is evaluated early or late at some nodes; early evaluations are repeated when the member's own upstream changes (its
// activations go to the shared masks), so the fixed point is reached for every member -- tests/tools/bundle_lab.cpp replays exactly this
// schedule on the CPU with the product's solve_node and compares every member with its solo run, bit for bit.
// Cycles (exact 2-cycles among ulp-tied nodes, fim_kernel.hip): detected on the bundle's combined change hash, frozen by the pilot's
// acceptance times; a cycle far behind the front pulls the window back to itself first (pass B, "stale" changes).  A bundle that runs out
// of rounds -- members whose fronts have nothing in common -- reports -1 to all its members and the engine solves them one by one.
// Field slots: a bundle claims a free one of the pool when it starts and releases it at the end (FimBundle::slot_busy).
#include "kernels.h"
#include "receiver_core.h"
#include "wave_ops.h"

namespace dsa {

namespace {

__device__ __forceinline__ unsigned bf2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bu2f(unsigned u) { return __uint_as_float(u); }

enum { BC_READY = 0, BC_READY_ODD, BC_TMIN, BC_THETA, BC_FREEZE, BC_HASH, BC_OVERFLOW, BC_CUR, BC_STALE, BC_STALEMIN, BC_SLOWN, BC_FARALL, BC_COUNT };
constexpr int kFarStrikes = 8;              // rounds in which more than a twelfth of the member evaluations went through the slow queue before the bundle fetches all outer neighbours
constexpr int kBundleCycleRounds = 8;
constexpr float kStaleWindows = 16.0f;      // see pass B: a change this many windows behind the pilot's front counts as stale
constexpr int kStaleRounds = 6;             // ... and pulls the window back when it has stayed at one node for so many rounds

typedef __attribute__((address_space(1))) unsigned long long BGU64;
typedef __attribute__((address_space(1))) char BGChar;
typedef __attribute__((address_space(1))) float BGF32;
typedef __attribute__((address_space(1))) const float BGCF32;
typedef float __attribute__((ext_vector_type(4))) BV4;
typedef __attribute__((address_space(1))) BV4 BGV4;
typedef float __attribute__((ext_vector_type(2))) BV2;
// the MPL members a lane carries (round 4: four, or two -- half the live values per lane, twice the lanes per node)
template <int MPL> struct BMem;
template <> struct BMem<4> { typedef BV4 V; typedef __attribute__((address_space(1))) BV4 GV; };
template <> struct BMem<2> { typedef BV2 V; typedef __attribute__((address_space(1))) BV2 GV; };

// Round 5: the outer neighbours a node trip fetches.  The regular walk (eikonal_core.h: solve_regular) uses the outer value of ONE
// neighbour per direction -- the upwind one, the smaller of the two near values -- so the two downwind outer vectors of a trip were 128
// fetched bytes each that no member read.  Pass A has the pilot's four near times of every listed node in registers anyway: it says
// which outer neighbours are worth fetching (bit q of a 4-bit code in the top bits of the ready-list entry; record indices stay below
// 2^28: Engine::choose_bundle_size / fits), pass B fetches those, and a member whose own upwind side is not among them (fronts that
// collide right there) goes through the slow queue, which reads everything.  A direction whose two near times are both unreached, or
// closer than a quarter of the causal window (the members' fronts may order them differently), keeps both sides.
constexpr unsigned kCandAlways = 0x80000000u;      // (round 6) a listed tie candidate (node << 4 | member) that goes to the census' second look whatever the field's values say (record indices stay below 2^27: Engine::choose_bundle_size / fits)
constexpr int kFarShift = 28;
__device__ __forceinline__ unsigned far_sides(float lo, float hi, float margin)
{
    if (!(fabsf(lo - hi) > margin)) return 3u;              // both unreached (inf - inf = NaN), or too close to call
    return hi < lo ? 2u : 1u;
}

// OR over the lanes of a node (CH consecutive lanes): quad permutes
template <int CH>
__device__ __forceinline__ unsigned node_or(unsigned v)
{
    if (CH >= 2) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    if (CH >= 4) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    if (CH >= 8) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);     // row_half_mirror: the other quad of the eight
    return v;
}

}  // namespace

// Waves per SIMD the register allocation aims at (256-thread workgroups; three waves per SIMD = three workgroups per CU, 168 VGPRs).
// Round 4: with the irregular members out of the trip (the slow queue of pass B) and the walk of the regular ones written out
// (solve_regular) TWO members per lane fit 168 VGPRs without a spill, four members per lane with two spilled registers (179 left alone).
// Measured at three per CU, 768 bundles at 1025^2 (profiles/r04_bundle_occupancy.log): bundles of 16 -- four per lane 240.9 ms, two per
// lane 256.4; bundles of 8 -- four per lane 194.9, two per lane 187.2.  So: 16 members four per lane, 8 and 4 members two per lane when
// the launch fills more than two workgroups per CU (four per lane, two per CU, below that).  512-thread workgroups (grids beyond 1500
// nodes per side) stay at two waves per SIMD.  -DDSA_BUNDLE_WAVES=n overrides for experiments.
#ifdef DSA_BUNDLE_WAVES
#define DSA_BUNDLE_OCC(G, NT, MPL) __attribute__((amdgpu_waves_per_eu(DSA_BUNDLE_WAVES, DSA_BUNDLE_WAVES)))
#else
#define DSA_BUNDLE_OCC(G, NT, MPL) __attribute__((amdgpu_waves_per_eu((((NT) == 256 && ((MPL) == 2 || (G) == 16)) || (NT) == 768) ? 3 : 1, (((NT) == 256 && ((MPL) == 2 || (G) == 16)) || (NT) == 768) ? 3 : 2)))
#endif
// NT = 256 threads per workgroup up to 1500 nodes per side (128: -1.7 %, 512: -13 % at 1025^2), 512 beyond: a 4097^2 front has ~2700 ready
// nodes per round, four times what 256 threads and their 2 x 1024 ready slots take
// TIE: the engine's tie detector (options exact_ties / tie_detect).  Round 4 sent every member evaluation whose walk stopped at an exact tie
// through the slow pass (+10 % kernel time on a medium without a single tie that matters, and transient ties of the iteration were counted
// like final ones).  Round 5: the detector is a CENSUS OF THE CONVERGED FIELD behind the round loop -- a node's walk stops at a tie exactly
// when one of its four near neighbours carries its value bit for bit, so one streaming pass over the bundle's field compares every node with
// its x+ and z+ neighbours (the members of a node side by side in one 16-byte vector), and only the tied (node, member) pairs are evaluated once
// more, by solve_node_t<true>, for the tie's influence on the node; influences above the unit's threshold go into the unit's tie record
// (count, largest influence).  That sweep reads the whole field once more (72 MB per bundle: +4.5 % of the step), so it is the FALLBACK:
// the round loop marks CANDIDATES -- solve_regular says for six instructions whether a member's walk stopped at a tie; a final tie between
// two nodes is seen by the last evaluation of the one evaluated later, so the candidates of the iteration contain every tie of the converged
// field -- into a list per bundle (FimBundle::cand: a wave collects them in LDS and appends once per half-round), and behind the loop only
// the listed (node, member) pairs are looked at: still tied in the converged field?  then the influence.  A list that overflows (a medium
// that ties everywhere) leaves the job to the sweep.
template <int G, int NT, int MPL = 4, bool TIE = false>
__global__ __launch_bounds__(NT) DSA_BUNDLE_OCC(G, NT, MPL) void k_fim_bundle(const FimBundle* __restrict__ bundles, const FimProblem* __restrict__ problems,
                                                    const FimEnds* __restrict__ ends)
{
    constexpr int NW = NT / 64;
    constexpr int CH = G / MPL;                // lanes per node in pass B, MPL (four, or two) members each
    typedef typename BMem<MPL>::V BV;
    typedef typename BMem<MPL>::GV BGV;
    constexpr int NPW = 64 / CH;               // nodes per wave trip
    constexpr unsigned GB = G * 4u * DSA_BSTRIDE;            // bytes per node (DSA_BSTRIDE = 2, probe builds: every node's segment alone in its 128-byte line -- more fetched bytes, the same instructions)
    extern __shared__ unsigned dyn_lds[];
    __shared__ int sc[BC_COUNT];
    constexpr int kWaveBuf = 256, kTileBuf = NT > 256 ? 128 : 256, kClrWords = 4;      // (512 threads: the per-tile scratch of eight waves has to fit the static LDS)
    __shared__ int wbuf[NW * kWaveBuf];
    __shared__ int wtile[NW * kTileBuf];
    __shared__ unsigned wclr[NW * kTileBuf * kClrWords];
    constexpr int rhalf = NT < 256 ? 1024 : NT > 512 ? 2048 : NT * 4;      // ready nodes of one colour a round can take (the rest keep their bits)
    __shared__ int ready[2 * rhalf];
    __shared__ int s_member[kBundleMax], s_map[kBundleMax];
    constexpr int kSlowQ = NT > 512 ? 512 : 1024;         // entries of a wave's slow queue (pass B)
    __shared__ int slowq[NW * kSlowQ];

    const FimBundle* const bd = bundles + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nmem = bd->nmem;
    if (tid < kBundleMax) { s_member[tid] = bd->member[tid < nmem ? tid : 0]; s_map[tid] = bd->map[tid < nmem ? tid : 0]; }
    const FimProblem p = problems[bd->member[0]];      // grid, window, tables: the same for all members
    // the bundle's field slot: its own, or the first free one of the pool (claimed by thread 0; released at the very end)
    __shared__ int s_slot;
    if (tid == 0) {
        int slot = bd->slot;
        if (bd->slot_busy) {
            const int P = bd->nslots;
            int s0 = (int)(blockIdx.x % (unsigned)P), q = s0;
            for (;;) {
                if (atomicCAS(&bd->slot_busy[q], 0, 1) == 0) break;
                q = q + 1 == P ? 0 : q + 1;
                if (q == s0) __builtin_amdgcn_s_sleep(64);
            }
            __threadfence();                            // (what the slot's previous user wrote is behind us)
            slot = q;
        }
        s_slot = slot;
    }
    __syncthreads();
    const int my_slot = __builtin_amdgcn_readfirstlane(s_slot);
    float* const Bslot = bd->B + (size_t)my_slot * bd->b_stride;
    BGChar* const Bb = (BGChar*)Bslot;
    BGChar* const excb = (BGChar*)(bd->exc + (size_t)my_slot * bd->exc_stride);
    const int xlog = bd->exc_log2cap;
    BGChar* const slowb = (BGChar*)bd->slowI;
    const unsigned npb = (unsigned)bd->np * 4u;         // bytes of slowness per node
    BGCF32* const risti = (BGCF32*)p.risti;
    const float far_margin0 = 0.25f * p.window;
    const int nnz = p.nnz, nnx = p.nnx, nbz = p.nbz;
    const int ntile = p.nbx * nbz, nwords = (ntile + 31) >> 5;
    BGChar* const maskb = (BGChar*)(bd->lists + (size_t)my_slot * bd->lists_stride);
    constexpr int kMaskShift = 5;
    auto mask_at = [&](int tile) -> BGU64* { return (BGU64*)(maskb + ((size_t)(unsigned)tile << kMaskShift)); };
    unsigned* const tb = dyn_lds;
    auto exc_at = [&](unsigned h) -> BGU64* { return (BGU64*)(excb + ((size_t)h << 3)); };
    // exception entries are keyed by id * G + member
    auto exc_lookup = [&](int key, bool* pinned) -> float {
        const unsigned mask = (1u << xlog) - 1u;
        unsigned h = exc_hash(key, xlog);
        for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
            const unsigned long long e = *exc_at(h);
            const int k = exc_key(e);
            if (k == -1) break;
            if ((k & 0x3fffffff) == key) { *pinned = (k & kExcPinned) != 0; return exc_tau(e); }
        }
        *pinned = false;
        return kInf;
    };
    auto exc_upsert = [&](int key, float tau) -> bool {
        const unsigned mask = (1u << xlog) - 1u;
        unsigned h = exc_hash(key, xlog);
        const unsigned long long mine = exc_pack(key, tau);
        for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
            unsigned long long e = *exc_at(h);
            if (exc_key(e) == -1) {
                e = atomicCAS((unsigned long long*)exc_at(h), kExcEmpty, mine);
                if (e == kExcEmpty) return true;
            }
            if ((exc_key(e) & 0x3fffffff) == key) { *exc_at(h) = mine; return true; }
        }
        return false;
    };
    const bool by_mul = nbz > 1 && (unsigned long long)ntile * (unsigned long long)nbz < (1ull << 32);
    const unsigned nbz_inv = by_mul ? 0xffffffffu / (unsigned)nbz + 1u : 0u;
    auto coords = [&](int id, int* iz0, int* ix0) {
        const unsigned tile = (unsigned)id >> 6;
        const unsigned bx = by_mul ? __umulhi(tile, nbz_inv) : tile / (unsigned)nbz;
        const unsigned bz = tile - bx * (unsigned)nbz;
        *ix0 = (int)(bx << kTileShift) + rec_ix_in_tile(id);
        *iz0 = (int)(bz << kTileShift) + rec_iz_in_tile(id);
    };
    // The pilot's value of a node (member 0), from its dense shadow copy P (one float per node, tiled like the field): pass A routes ~700
    // listed nodes per round by five pilot values each, and reading them from the member-minor field moved a 128-byte line per value --
    // two thirds of the kernel's fetched bytes (profiles/r03_bundle_sizes.log).  Written next to the field by the lane that owns member 0.
    BGChar* const Pb = (BGChar*)(Bslot + bd->p_offset);
    auto pv = [&](int id) -> float { return *(BGF32*)(Pb + ((unsigned)id << 2)); };

    // ---- the bundle's field slot (claimed above): every node of every member unreached, the table empty; the nodes each member's serial
    // prologue pinned (window records, k_coarse_march) into both
    __syncthreads();
    bool dead = false;
    {
        const BV4 inf4 = { kInf, kInf, kInf, kInf };
        for (int i = tid; i < ntile * (kTileRecs * G / 4) * DSA_BSTRIDE; i += NT) ((BGV4*)Bb)[i] = inf4;
        for (int i = tid; i < ntile * (kTileRecs / 4); i += NT) ((BGV4*)Pb)[i] = inf4;
        for (int i = tid; i < (1 << xlog); i += NT) *exc_at((unsigned)i) = kExcEmpty;
        for (int i = tid; i < (ntile << (kMaskShift - 3)); i += NT) *(BGU64*)(maskb + ((size_t)i << 3)) = 0ull;
        for (int i = tid; i < nwords; i += NT) tb[i] = 0u;
        if (tid == 0) {
            sc[BC_READY] = 0; sc[BC_READY_ODD] = 0; sc[BC_TMIN] = 0x7f800000; sc[BC_THETA] = 0x7f800000;
            sc[BC_FREEZE] = (int)0xff800000u; sc[BC_HASH] = 0; sc[BC_OVERFLOW] = 0; sc[BC_CUR] = 0; sc[BC_STALE] = (int)0xff800000u; sc[BC_STALEMIN] = 0x7f800000;
            sc[BC_SLOWN] = 0; sc[BC_FARALL] = bd->far_all ? 1 : 0;
            if (TIE && bd->cand) *(bd->cand + (size_t)my_slot * bd->cand_stride) = 0;
        }
        if (TIE && bd->cand) {          // the census' "looked at" set: a hash table of kTieSeenSlots (node, member) keys behind the candidate list
            BGV4* const seen4 = (BGV4*)(bd->cand + (size_t)my_slot * bd->cand_stride + ((bd->cand_cap + 4) & ~3));
            const BV4 zero4 = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int i = tid; i < kTieSeenSlots / 4; i += NT) seen4[i] = zero4;
        }
        __threadfence_block();
        __syncthreads();
        typedef __attribute__((address_space(1))) const Rec GCRec;
        for (int m = 0; m < nmem; ++m) {
            const FimEnds* const E = ends + s_member[m];
            const FimProblem* const pm = problems + s_member[m];
            // pinned nodes: the records of the coarse march window -- or (the refined boxes in bundles) the unit's own tiled records, which the
            // start-up march pinned: same tiling as the bundle's field, so a record's index is the node's
            const bool from_records = E->Fpin != nullptr;
            GCRec* const W = from_records ? (GCRec*)E->Fpin : (GCRec*)E->W;
            const int cwz0 = E->cwz0, cwx0 = E->cwx0, cwnz = from_records ? 1 : E->cwnz, nw = from_records ? ntile * kTileRecs : E->cwnx * cwnz;
            for (int q = tid; q < nw; q += NT) {
                const float wt = W[q].T, wk = W[q].tau;
                if (!t_pinned(wt)) continue;
                const int lx = q / cwnz, lz = q - lx * cwnz;
                const int id = from_records ? q : rec_index(nbz, cwz0 + lz, cwx0 + lx);
                const int key = id * G + m;
                const unsigned long long mine = exc_pack(key | kExcPinned, wk);
                const unsigned mask = (1u << xlog) - 1u;
                unsigned h = exc_hash(key, xlog);
                bool placed = false;
                for (unsigned n = 0; n <= mask && !placed; ++n, h = (h + 1u) & mask)
                    placed = atomicCAS((unsigned long long*)exc_at(h), kExcEmpty, mine) == kExcEmpty;
                if (!placed) { pm->info[2] = -2; sc[BC_OVERFLOW] = 1; }
                *(BGF32*)(Bb + (unsigned)id * GB + (unsigned)m * 4u) = wt;      // -T: the sign bit marks the exceptional node
                if (m == 0) *(BGF32*)(Pb + ((unsigned)id << 2)) = wt;
            }
            // the member's seeds: the rim of its pinned set
            const int nseed = *pm->seed_count;
            if (nseed > pm->seed_cap) { if (tid == 0) pm->info[2] = -1; dead = true; }
            else
                for (int i = tid; i < nseed; i += NT) {
                    const int id = pm->seed[i];
                    atomicOr((unsigned long long*)mask_at(id >> 6), 1ull << (id & 63));
                    atomicOr(&tb[(id >> 6) >> 5], 1u << ((id >> 6) & 31));
                }
        }
        __threadfence_block();
        __syncthreads();
        if (sc[BC_OVERFLOW]) dead = true;
    }

    // dependents of node id: the mask bits are constant shifts of the node's own bit (fim_kernel.hip); wm: bit q near, bit 4 + q outer neighbour q
    auto activate_node = [&](int id, unsigned wm, int half) {
        const int own_tile = id >> 6;
        const unsigned long long b = 1ull << (id & 63);
        auto sel = [](unsigned w, unsigned long long v) -> unsigned long long { return w ? v : 0ull; };
        const unsigned long long own_bits =
            sel(wm & 1u, b >> 8) | sel(wm & 16u, b >> 16) | sel(wm & 2u, b << 8) | sel(wm & 32u, b << 16) |
            sel(wm & 4u, (b >> 1) & 0x7f7f7f7f7f7f7f7full) | sel(wm & 64u, (b >> 2) & 0x3f3f3f3f3f3f3f3full) |
            sel(wm & 8u, (b << 1) & 0xfefefefefefefefeull) | sel(wm & 128u, (b << 2) & 0xfcfcfcfcfcfcfcfcull);
        const unsigned long long fxm = sel(wm & 1u, b << 56) | sel(wm & 16u, b << 48);
        const unsigned long long fxp = sel(wm & 2u, b >> 56) | sel(wm & 32u, b >> 48);
        const unsigned long long fzm = sel(wm & 4u, (b << 7) & 0x8080808080808080ull) | sel(wm & 64u, (b << 6) & 0xc0c0c0c0c0c0c0c0ull);
        const unsigned long long fzp = sel(wm & 8u, (b >> 7) & 0x0101010101010101ull) | sel(wm & 128u, (b >> 6) & 0x0303030303030303ull);
        auto activate = [&](int tile, unsigned long long bits) {
            if (bits) {
                atomicOr((unsigned long long*)(mask_at(tile) + half), bits);
                atomicOr(&tb[tile >> 5], 1u << (tile & 31));
            }
        };
        activate(own_tile - nbz, fxm);
        activate(own_tile + nbz, fxp);
        activate(own_tile - 1, fzm);
        activate(own_tile + 1, fzp);
        activate(own_tile, own_bits);
    };
    // the neighbourhood of one member (index mo of the bundle) of node id from memory, exception table included; returns the node's own state
    auto load_hood = [&](int id, int mo, int ix, int iz, Hood& h, float* t_old, float* k_old) {
        int nid[8];
        rec_stencil(nbz, id, nid);
        const unsigned mb = (unsigned)mo * 4u;
        h.in[0] = ix > 0;          h.in_outer[0] = ix > 1;
        h.in[1] = ix + 1 < nnx;    h.in_outer[1] = ix + 2 < nnx;
        h.in[2] = iz > 0;          h.in_outer[2] = iz > 1;
        h.in[3] = iz + 1 < nnz;    h.in_outer[3] = iz + 2 < nnz;
        // the nine values of the member's neighbourhood and, for the exceptional ones (sign bit), their acceptance times from the table: ONE
        // copy of the look-up, the nine positions rotated through it (nine inlined copies kept ~20 more registers alive, and this loop -- not
        // the regular members' bodies -- decided the kernel's register count)
        float val[9], tau[9];
        int kid[9];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            val[q] = h.in[q] ? *(BGF32*)(Bb + (unsigned)nid[q] * GB + mb) : kInf;
            val[4 + q] = h.in_outer[q] ? *(BGF32*)(Bb + (unsigned)nid[4 + q] * GB + mb) : kInf;
            kid[q] = nid[q]; kid[4 + q] = nid[4 + q];
        }
        val[8] = *(BGF32*)(Bb + (unsigned)id * GB + mb);
        kid[8] = id;
#pragma unroll
        for (int q = 0; q < 9; ++q) tau[q] = val[q];
#pragma nounroll
        for (int it = 0; it < 9; ++it) {
            float v = val[0], t = tau[0];
            if (__builtin_signbit(v)) { bool pin; t = exc_lookup(kid[0] * G + mo, &pin); v = pin ? v : -v; }
            const int k0 = kid[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) { val[j] = val[j + 1]; tau[j] = tau[j + 1]; kid[j] = kid[j + 1]; }
            val[8] = v; tau[8] = t; kid[8] = k0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { h.near_[q] = val[q]; h.near_tau[q] = tau[q]; h.outer[q] = val[4 + q]; h.outer_tau[q] = tau[4 + q]; }
        *t_old = val[8]; *k_old = tau[8];
    };
    // one member of node id, from memory: the slow pass of pass B
    auto slow_member = [&](int id, int mo, int half, float stale, unsigned& evals, unsigned& nchanged, unsigned& hv_lane, float& kmin_lane, float& smin_lane) -> bool {
        if (mo >= nmem) return false;
        int iz, ix;
        coords(id, &iz, &ix);
        const unsigned mb = (unsigned)mo * 4u;
        Hood h;
        float t_old, k_old;
        load_hood(id, mo, ix, iz, h, &t_old, &k_old);
        const int key = id * G + mo;
        if (t_pinned(t_old)) return false;
        float k = kInf;
        const float slown = *(BGCF32*)(slowb + (unsigned)id * npb + (unsigned)s_map[mo] * 4u);
        const NodeGeom geom = { p.ri, risti[ix], p.dnx, p.dnz };
        // (TIE) a candidate of the census: the detector's walk says so -- it stopped at an exact tie, an exceptional outer node was accepted at the clock of
        // the neighbour taken in last, or the node took a neighbour in that its raised key may not have waited for (eikonal_core.h: solve_node_t<true>; same
        // value and acceptance time as the plain walk).  The list may hold more than the converged field's ties: the census' second look decides.
        // (TIE) a candidate of the census -- the list may hold more than the converged field's ties, the second look decides: the value equals a near
        // neighbour's acceptance time (every walk that stops at an exact tie does).  The other kinds of tie involve an exceptional node and are found
        // behind the loop from the exception table (the census); marking them here -- the detector's own walk, or the plain walk noting raised keys and a
        // check of the outer nodes' acceptance times -- cost 7.6 / 14 ms of the headline's 335 (profiles/r06_ab_census.log).
        float amb = -1.0f;
#ifdef DSA_BUNDLE_MARK_RAISED_KEYS      // (the slow pass notes raised keys itself: +7 ms on the headline's 335; without it they are found where the node ends exceptional)
        const float c = TIE ? solve_node_t<false, true>(h, slown, geom, &k, &amb) : solve_node_t<false>(h, slown, geom, &k, nullptr);
#else
        const float c = solve_node_t<false>(h, slown, geom, &k, nullptr);
#endif
        bool tied = TIE && amb >= 0.0f;
        if (TIE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tied = tied || (h.in[q] && c < kInf && c == h.near_tau[q]);
        }
        ++evals;
        if (bf2u(c) == bf2u(t_old) && bf2u(k) == bf2u(k_old)) return tied;
        float newv = c;
        if (bf2u(c) != bf2u(k)) { if (!exc_upsert(key, k)) { p.info[2] = -2; sc[BC_OVERFLOW] = 1; } newv = -c; }
        *(BGF32*)(Bb + (unsigned)id * GB + mb) = newv;
        if (mo == 0) { *(BGF32*)(Pb + ((unsigned)id << 2)) = newv; kmin_lane = fminf(kmin_lane, k); }
        ++nchanged;
        hv_lane += ((unsigned)key * 2654435761u) ^ (bf2u(c) * 40503u) ^ (bf2u(k) * 2246822519u);
        {   // (a change far behind the front: see pass B)
            const float pt = mo == 0 ? newv : pv(id);
            if (!__builtin_signbit(pt) && pt < stale) smin_lane = fminf(smin_lane, pt);
        }
        unsigned wm = 0u;
        const float t_lo = fminf(t_value(t_old), c), k_lo = fminf(k_old, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ky = h.near_tau[q];
            if (h.in[q] && !t_pinned(h.near_[q]) && k_lo <= ky) wm |= 1u << q;
            if (h.in_outer[q] && ky < kInf && !t_pinned(h.outer[q]) && t_value(h.near_[q]) > t_lo && k_lo < h.outer_tau[q]) wm |= 16u << q;
        }
        if (wm) activate_node(id, wm, half);
        return tied;
    };
    // the tie census' second look (TIE): member mo of node id in the converged field, evaluated once more with the detector's walk
    unsigned* const seen_g = (TIE && bd->cand) ? (unsigned*)(bd->cand + (size_t)my_slot * bd->cand_stride + ((bd->cand_cap + 4) & ~3)) : nullptr;
    auto tie_member = [&](int id, int mo) {
        if (mo >= nmem) return;
        int iz, ix;
        coords(id, &iz, &ix);
        if (ix >= nnx || iz >= nnz) return;
        if (seen_g) {
            // every (node, member) pair once -- the list names a node as often as its evaluations ended on a tie, the sweep once per tied side --: its key
            // goes into a small open-addressing table (a bitmap over the field cost 2.1 MB of clearing per bundle, 2.7 ms of the headline step).  A key
            // that finds no room within eight probes (a medium with tens of thousands of ties per bundle) is looked at again: the largest influence
            // does not mind, the count and the sum then count that pair more than once.
            const unsigned key = (unsigned)id * 16u + (unsigned)mo + 1u;
            unsigned hs = (key * 2654435761u) >> (32 - kTieSeenLog2);
            for (int n = 0; n < 8; ++n, hs = (hs + 1u) & (unsigned)(kTieSeenSlots - 1)) {
                const unsigned old = atomicCAS(&seen_g[hs], 0u, key);
                if (old == key) return;
                if (old == 0u) break;
            }
        }
        Hood h;
        float t_old, k_old;
        load_hood(id, mo, ix, iz, h, &t_old, &k_old);
        if (t_pinned(t_old)) return;
        float k = kInf, ti = -1.0f;
        const float slown = *(BGCF32*)(slowb + (unsigned)id * npb + (unsigned)s_map[mo] * 4u);
        const NodeGeom geom = { p.ri, risti[ix], p.dnx, p.dnz };
        (void)solve_node_t<true>(h, slown, geom, &k, &ti);
        const FimProblem* const pm = problems + s_member[mo];
        if (ti >= 0.0f && pm->tie) {
            if (ti > 0.0f) { atomicAdd((unsigned*)pm->tie + 2, 1u); atomicAdd((unsigned*)pm->tie + 3, (unsigned)(fminf(ti, 1.0f) * (1.0f / kTieSumUnit))); }
            else atomicAdd((unsigned*)pm->tie + 5, 1u);         // (a tie whose tied neighbour changes nothing at this node)
            if (ti > pm->tie_threshold) { atomicAdd((unsigned*)pm->tie, 1u); atomicMax((unsigned*)pm->tie + 1, bf2u(ti)); }
        }
    };
    int* const wq = slowq + wave * kSlowQ;              // the wave's queue of (node << 4 | member) left to the slow pass
    int qn = 0;
    int cn = 0;                                         // (TIE) tie candidates of the half-round, kept from the queue's top end downwards
    int* const cand_g = (TIE && bd->cand && bd->cand_list) ? bd->cand + (size_t)my_slot * bd->cand_stride : nullptr;
    // a candidate per lane with `on` (wave-uniform control flow): into the LDS list while it has room beside the slow queue, else the count alone
    // grows -- the flush then reports more candidates than the list holds and the census sweeps the field
    bool clost = false;                                 // (a candidate found no room beside the slow queue: the bundle's list counts as overflowed)
    auto cand_push = [&](bool on, int word) {
        const unsigned long long bal = __ballot(on);
        if (!bal) return;
        const int pos = cn + __popcll(bal & ((1ull << lane) - 1ull));
        cn += __popcll(bal);
        if (qn + cn >= kSlowQ - 1) { clost = true; return; }
        if (on) wq[kSlowQ - 1 - pos] = word;
    };
    auto cand_flush = [&]() {          // (behind the half-round's slow pass: qn is 0 again)
        if (!cn || !cand_g) { cn = 0; clost = false; return; }
        // (the count saturates just above the capacity: a medium that ties everywhere overflows the list every half-round, and a counter that kept
        // adding would wrap -- ADVICE r05; the census reads "more than cand_cap" as overflow and sweeps the field)
        int base = -1;
        if (lane == 0) {
            if (clost) (void)atomicMax(cand_g, bd->cand_cap + 1);
            else if (__hip_atomic_load(cand_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= bd->cand_cap) base = atomicAdd(cand_g, cn);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (!clost && base >= 0) for (int i = lane; i < cn; i += 64) if (base + i < bd->cand_cap) cand_g[1 + base + i] = wq[kSlowQ - 1 - i];
        cn = 0; clost = false;
    };

    int rounds = 0, stall = 0, freezes = 0, sm_same = 0, far_strikes = 0;
    float sm_prev = kInf, sm_prev2 = kInf;
    unsigned hist[4] = { 1u, 2u, 3u, 4u };
    float best_tmin = -kInf;
    unsigned evals = 0, nchanged = 0;                        // member evaluations of this lane
    // pass B: this lane's node slot, member chunk, member maps
    const int sub = lane % CH, nslot = lane / CH;
    const unsigned sub_b = (unsigned)sub * (4u * MPL);
    unsigned mp[MPL], vmask = 0u;                             // (bit m: member sub * MPL + m exists)
#pragma unroll
    for (int m = 0; m < MPL; ++m) { mp[m] = (unsigned)s_map[sub * MPL + m] * 4u; if (sub * MPL + m < nmem) vmask |= 1u << m; }
    // (round 5) the lane's members on consecutive maps, the first one a multiple of MPL (the periods of a source in order: the usual
    // case): their slowness values are one aligned vector of the member-minor copy -- one load instead of MPL
    bool sl_vec = (bd->np % MPL) == 0;
    {
        const int base = sub * MPL < nmem ? s_map[sub * MPL] : 0;
        sl_vec = sl_vec && (base % MPL) == 0 && base + MPL <= bd->np;
#pragma unroll
        for (int m = 1; m < MPL; ++m) if (sub * MPL + m < nmem && s_map[sub * MPL + m] != base + m) sl_vec = false;
        if (sl_vec) mp[0] = (unsigned)base * 4u;
    }
    sl_vec = __all(sl_vec);                                   // (one path per wave)

#ifdef DSA_BUNDLE_CLOCKS      // probe: where a round's wall clock goes (thread 0: pass A incl. its barrier, even half, odd half, bookkeeping), into clocks[0..3] of the pilot
    unsigned long long bt[4] = { 0, 0, 0, 0 }, bt0 = wall_clock64();
#define DSA_BCLK(k) do { const unsigned long long t1_ = wall_clock64(); bt[k] += t1_ - bt0; bt0 = t1_; } while (0)
#else
#define DSA_BCLK(k) do { } while (0)
#endif
    for (; !dead;) {
        const float theta = bu2f((unsigned)sc[BC_THETA]);
        const bool open = !(theta < kInf);
        const float freeze = bu2f((unsigned)sc[BC_FREEZE]);
        const bool frozen_any = freeze > -kInf;
        const float stale = bu2f((unsigned)sc[BC_STALE]);      // the farthest the window's lower edge has been, less kStaleWindows windows (pass B)
        // (members whose fronts run differently from the pilot's -- unrelated maps per period -- find their upwind outer value missing and queue for
        // the slow pass; a bundle in which that keeps happening goes back to fetching all four outer neighbours: bookkeeping below)
        const float far_margin = sc[BC_FARALL] ? kInf : far_margin0;

        // ---- pass A (as in k_fim_sorted, on the pilot's times): every wave sweeps its share of the tile bitmap, gathers the active tiles,
        // expands their node masks in record order, computes the lower bounds and routes
        int seen = 0;
        float tmin_lane = kInf;
        int* const nbuf = wbuf + wave * kWaveBuf;
        int* const tbuf = wtile + wave * kTileBuf;
        unsigned* const clr = wclr + wave * kTileBuf * kClrWords;
        constexpr int kQ = kTileBuf / 64, kI = kWaveBuf / 64;
        auto sweep_tiles = [&](int ntiles) {
            __builtin_amdgcn_s_setprio(3);
            int tl[kQ];
            unsigned long long m[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                tl[q] = q * 64 + lane < ntiles ? tbuf[q * 64 + lane] : -1;
                m[q] = 0ull;
                if (tl[q] >= 0) {
                    BGU64* const rec4 = mask_at(tl[q]);
                    const unsigned long long E = rec4[0], O = rec4[1], R = rec4[2];
                    const unsigned stamp = (unsigned)rec4[3];
                    m[q] = (E & ~(stamp == (unsigned)rounds ? R : 0ull)) | O;
                }
                if (q * 64 < ntiles)
                    for (int w4 = 0; w4 < kClrWords; ++w4) clr[kClrWords * (q * 64 + lane) + w4] = 0u;
            }
            int off[kQ], total = 0;
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                off[q] = total;
                if (q * 64 >= ntiles) continue;
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
                        const int nb = __ffsll((long long)mm) - 1;
                        mm &= mm - 1ull;
                        if (idx >= base && idx < base + kWaveBuf) nbuf[idx - base] = ((q * 64 + lane) << 6) + nb;     // (tile slot, node)
                        ++idx;
                    }
                }
                const int nn = min(total - base, kWaveBuf);
                int id[kI], par[kI], slot[kI];
                float lb[kI], own[kI];
                bool ppin[kI];
                unsigned far_code[kI];                                   // (round 5) which outer neighbours pass B fetches: see kFarShift
#pragma unroll
                for (int i = 0; i < kI; ++i) {
                    id[i] = -1; par[i] = 0; slot[i] = 0; lb[i] = kInf; own[i] = kInf; ppin[i] = false; far_code[i] = 15u;
                    if (i * 64 >= nn) continue;
                    const bool have = i * 64 + lane < nn;
                    const int e = have ? nbuf[i * 64 + lane] : 0;
                    slot[i] = e >> 6;
                    id[i] = have ? (tbuf[slot[i]] << 6) + (e & 63) : -1;
                    int iz, ix;
                    coords(have ? id[i] : 0, &iz, &ix);
                    par[i] = (ix + iz) & 1;
                    if (have) {
                        int nid[8];
                        rec_stencil(nbz, id[i], nid);
                        float a = ix > 0 ? pv(nid[0]) : kInf, b2 = ix + 1 < nnx ? pv(nid[1]) : kInf;
                        float c2 = iz > 0 ? pv(nid[2]) : kInf, d2 = iz + 1 < nnz ? pv(nid[3]) : kInf;
                        own[i] = pv(id[i]);
                        if (__builtin_signbit(a) || __builtin_signbit(b2) || __builtin_signbit(c2) || __builtin_signbit(d2) || __builtin_signbit(own[i])) {
                            bool pin;
                            if (__builtin_signbit(a)) a = exc_lookup(nid[0] * G, &pin);
                            if (__builtin_signbit(b2)) b2 = exc_lookup(nid[1] * G, &pin);
                            if (__builtin_signbit(c2)) c2 = exc_lookup(nid[2] * G, &pin);
                            if (__builtin_signbit(d2)) d2 = exc_lookup(nid[3] * G, &pin);
                            if (__builtin_signbit(own[i])) { own[i] = exc_lookup(id[i] * G, &pin); ppin[i] = pin; }
                        }
                        lb[i] = fminf(fminf(a, b2), fminf(c2, d2));
                        far_code[i] = far_sides(a, b2, far_margin) | (far_sides(c2, d2, far_margin) << 2);
                    }
                }
                __builtin_amdgcn_s_setprio(2);
                unsigned long long be[kI], bo[kI];
                bool frozen[kI];
                int ne = 0, no = 0;
#pragma unroll
                for (int i = 0; i < kI; ++i) {
                    be[i] = 0ull; bo[i] = 0ull; frozen[i] = false;
                    if (i * 64 >= nn) continue;
                    const bool have = id[i] >= 0;
                    frozen[i] = have && frozen_any && !ppin[i] && own[i] < freeze;
                    const bool want = have && !frozen[i] && (open || lb[i] < theta || ppin[i]);     // (the pilot pinned here: the other members' turn cannot be told from its times)
                    be[i] = __ballot(want && par[i] == 0);
                    bo[i] = __ballot(want && par[i] != 0);
                    ne += __popcll(be[i]); no += __popcll(bo[i]);
                }
                int base_e = 0, base_o = 0;
                if (lane == 0) {
                    if (ne) base_e = atomicAdd(&sc[BC_READY], ne);
                    if (no) base_o = atomicAdd(&sc[BC_READY_ODD], no);
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
                    if (got) ready[want_o ? rhalf + po : pe] = (int)((unsigned)id[i] | (far_code[i] << kFarShift));
                    if ((got && !want_o) || frozen[i]) atomicOr(&clr[kClrWords * slot[i] + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));
                    if (got && want_o) atomicOr(&clr[kClrWords * slot[i] + 2 + ((id[i] >> 5) & 1)], 1u << (id[i] & 31));     // evaluated by this round's odd half
                    if (have && !frozen[i] && !got) tmin_lane = fminf(tmin_lane, lb[i]);
                }
            }
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (tl[q] >= 0) {
                    const unsigned* const cw = clr + kClrWords * (q * 64 + lane);
                    const unsigned long long c = (unsigned long long)cw[0] | ((unsigned long long)cw[1] << 32);
                    const unsigned long long ro = (unsigned long long)cw[2] | ((unsigned long long)cw[3] << 32);
                    BGU64* const rec4 = mask_at(tl[q]);
                    rec4[0] = m[q] & ~c & ~ro; rec4[1] = 0ull; rec4[2] = ro; rec4[3] = (unsigned long long)(unsigned)(rounds + 1);
                }
        };
        int ntw = 0;
        const int colw = nbz >> 5;
        const int gs_col = colw >= 16 ? 4 : colw >= 8 ? 3 : colw >= 4 ? 2 : colw >= 2 ? 1 : 0;
        const int gs = nwords >= 64 * NW ? gs_col : 0;
        {
            int wb = 0, tbase = 0, ttotal = 0, toff = 0, w = 0;
            unsigned bits = 0u;
            bool words_left = true;
            while (words_left || ntw) {
                while (words_left && ntw < kTileBuf) {
                    if (tbase >= ttotal) {
                        if (!(((wb * (64 >> gs) * NW + wave) << gs) < nwords)) { words_left = false; break; }
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
        tmin_lane = wave_min(tmin_lane);
        if (lane == 0) {
            if (tmin_lane < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[BC_TMIN]), bf2u(tmin_lane));
            if (seen) atomicAdd(&sc[BC_CUR], seen);
        }
        __syncthreads();
        DSA_BCLK(0);
        const int cnt = sc[BC_CUR];
        if (cnt == 0) break;

        // ---- pass B: the ready nodes for all members, even nodes first
        const int nready_even = sc[BC_READY] < rhalf ? sc[BC_READY] : rhalf;
        const int nready_odd = sc[BC_READY_ODD] < rhalf ? sc[BC_READY_ODD] : rhalf;
        unsigned hv_lane = 0u;
        float kmin_lane = kInf, smin_lane = kInf;
        for (int half = 0; half < 2; ++half) {
            const int nready = half ? nready_odd : nready_even;
            for (int j0 = wave * NPW; j0 < nready; j0 += NW * NPW) {
                const int j = j0 + nslot;
                const bool act = j < nready;
                const unsigned ent = act ? (unsigned)ready[half ? rhalf + j : j] : 0u;
                const int id = (int)(ent & ((1u << kFarShift) - 1u));
                const unsigned farc = ent >> kFarShift;                      // outer neighbours to fetch (pass A)
                int iz, ix;
                __builtin_amdgcn_s_setprio(3);
                coords(id, &iz, &ix);
                bool in[4], in_outer[4];
                in[0] = act && ix > 0;          in_outer[0] = act && ix > 1;
                in[1] = act && ix + 1 < nnx;    in_outer[1] = act && ix + 2 < nnx;
                in[2] = act && iz > 0;          in_outer[2] = act && iz > 1;
                in[3] = act && iz + 1 < nnz;    in_outer[3] = act && iz + 2 < nnz;
                int nid[8];
                rec_stencil(nbz, id, nid);
                BV infv, nanv, vown, sl;
#pragma unroll
                for (int m = 0; m < MPL; ++m) { infv[m] = kInf; nanv[m] = __builtin_nanf(""); vown[m] = -1.0f; sl[m] = 1.0f; }          // (inactive lanes read as pinned)
                BV vn[4], vo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    vn[q] = in[q] ? *(BGV*)(Bb + (unsigned)nid[q] * GB + sub_b) : infv;
                    // (an outer neighbour that exists but is not fetched reads as NaN: solve_regular refuses a member whose upwind side it is, and
                    // the pruning below counts it as "later than this node", i.e. activates: the conservative side)
                    vo[q] = !in_outer[q] ? infv : ((farc >> q) & 1u) ? *(BGV*)(Bb + (unsigned)nid[4 + q] * GB + sub_b) : nanv;
                }
                if (act) {
                    vown = *(BGV*)(Bb + (unsigned)id * GB + sub_b);
                    const unsigned sb = (unsigned)id * npb;
                    if (sl_vec) sl = *(BGV*)(slowb + sb + mp[0]);
                    else {
#pragma unroll
                        for (int m = 0; m < MPL; ++m) sl[m] = *(BGCF32*)(slowb + sb + mp[m]);
                    }
                }
                const NodeGeom geom = { p.ri, risti[ix], p.dnx, p.dnz };
                __builtin_amdgcn_s_setprio(0);
                BV outv = vown;
                unsigned wm = 0u;                                    // dependents some member wants activated: bit q near, bit 4 + q outer
                bool any_changed = false;
                const int key0 = id * G + sub * MPL;
                // Round 4: the member bodies below are the REGULAR case only -- no exceptional node (a pinned one, or one whose acceptance time
                // differs from its value: ~0.2 % of the evaluations) in the member's neighbourhood, and a result that is causal (tau = T).  A member
                // that is not regular is left for the slow pass behind the store (bit m of `slow`), which evaluates it from memory with the
                // exception table at hand.  What this buys is registers: the table look-ups and the table insert, inlined into every member
                // body, kept ~30 more VGPRs alive (profiles/r04_bundle_vgprs.txt), and 168 is the line for a third workgroup per CU.
                unsigned slow = 0u, tiem = 0u;                               // (tiem, TIE: members whose walk stopped at an exact tie)
                const bool interior = in[0] && in[1] && in[2] && in[3];      // (a node on the grid's edge: the general walk's business)
#pragma unroll
                for (int m = 0; m < MPL; ++m) {
                    float tn[4], t2[4];
                    bool flagged = !interior;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        tn[q] = vn[q][m]; t2[q] = vo[q][m];
                        flagged = flagged | __builtin_signbit(tn[q]) | __builtin_signbit(t2[q]);
                    }
                    const float raw = vown[m];
                    const bool valid = act && ((vmask >> m) & 1u);
                    flagged = flagged | __builtin_signbit(raw);
                    float c = 0.0f, k = kInf;
                    bool changed = false;
                    if (valid && !flagged) {
                        // (round 4) the walk written out for the regular neighbourhood: straight-line code, about half the instructions of
                        // solve_node_t's loop; where it does not apply (a third neighbour taken in, the opposite neighbour second: ~1 % of
                        // the evaluations) it says so and the member goes to the slow pass like an exceptional one
                        bool ok, tie = false;
                        c = solve_regular(tn, t2, sl[m], geom, &k, &ok, TIE ? &tie : nullptr);
                        if (!ok || bf2u(c) != bf2u(k)) flagged = true;               // (a non-causal result: the table's business)
                        else { ++evals; changed = bf2u(c) != bf2u(raw); if (TIE && tie) tiem |= 1u << m; }
                    }
                    if (valid && flagged) slow |= 1u << m;
                    if (changed) {
                        outv[m] = c;
                        ++nchanged;
                        any_changed = true;
                        hv_lane += ((unsigned)(key0 + m) * 2654435761u) ^ (bf2u(c) * 40503u) ^ (bf2u(k) * 2246822519u);
                        if (sub == 0 && m == 0) kmin_lane = fminf(kmin_lane, k);                 // the pilot's changes hold the window back
                        const float t_lo = fminf(raw, c), k_lo = t_lo;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float ky = tn[q];
                            if (k_lo <= ky) wm |= 1u << q;                               // (interior: the four near neighbours exist)
                            if (in_outer[q] && ky < kInf && ky > t_lo && !(k_lo >= t2[q])) wm |= 16u << q;      // (t2 NaN: not fetched, see above)
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(1);
                if (any_changed) {
                    *(BGV*)(Bb + (unsigned)id * GB + sub_b) = outv;
                    if (sub == 0) *(BGF32*)(Pb + ((unsigned)id << 2)) = outv[0];       // the pilot's shadow copy (unchanged pilots rewrite their value)
                }
                // the members left out above go to the wave's queue (node, member of the bundle); a full queue (it holds the pinned neighbourhood of
                // the first rounds several times over) puts the node back on its tile's mask instead: it is listed again next round
                if (__any(slow != 0u)) {
                    bool again = false;
#pragma unroll
                    for (int m = 0; m < MPL; ++m) {
                        const bool push = ((slow >> m) & 1u) != 0u;
                        const unsigned long long bal = __ballot(push);
                        const int pos = qn + __popcll(bal & ((1ull << lane) - 1ull));
                        if (push && pos < kSlowQ - 1 - cn) wq[pos] = (id << 4) | (sub * MPL + m);      // (the queue's top end holds the half-round's tie candidates)
                        again = again || (push && pos >= kSlowQ - 1 - cn);
                        qn += __popcll(bal);
                    }
                    if (again) { atomicOr((unsigned long long*)(mask_at(id >> 6) + half), 1ull << (id & 63)); atomicOr(&tb[(id >> 6) >> 5], 1u << ((id >> 6) & 31)); }
                }
                if (TIE && cand_g && __any(tiem != 0u)) {
#pragma unroll
                    for (int m = 0; m < MPL; ++m) cand_push(((tiem >> m) & 1u) != 0u, (id << 4) | (sub * MPL + m));
                }
                // dependents: the members' OR, one lane per node issues the activations (fim_kernel.hip: the mask bits are constant shifts
                // of the node's own bit)
                wm = node_or<CH>(wm);
                // A change far BEHIND the front (the window's lower edge has been kStaleWindows windows beyond this node) is a member that lags
                // that much, a late refinement on its way downstream -- or a cycle (exact 2-cycles among ulp-tied nodes, fim_kernel.hip), which
                // in a member other than the pilot would flip on, and spread its flips downstream, until the rest of the bundle has converged
                // (+2000 rounds for such a bundle at a 1.5-cell window).  The earliest such change of the round is recorded; when it stays at
                // the SAME node round after round (a cycle does, a wave moves on) the bookkeeping below pulls the window back to it, as every
                // change does in a unit-by-unit solve: nothing ahead moves, the repetition shows in the change hash, the freeze horizon covers
                // the node, and the window returns to the front.
                {
                    const unsigned chg = node_or<CH>(any_changed ? 1u : 0u);
                    const float pt = outv[0];                                   // (lane sub == 0: the pilot's value at this node, new or unchanged)
                    if (sub == 0 && chg && !__builtin_signbit(pt) && pt < stale) smin_lane = fminf(smin_lane, pt);
                }
                if (sub == 0 && wm) activate_node(id, wm, half);
            }
            // the wave's slow queue: one thread per (node, member), the member body of round 3 whole -- exception table look-ups, a result
            // whose acceptance time differs from its value into the table -- on values read from memory, with its own activations
            if (lane == 0 && qn) atomicAdd(&sc[BC_SLOWN], qn);
            qn = qn < kSlowQ - 1 - cn ? qn : max(kSlowQ - 1 - cn, 0);
            for (int base = 0; base < qn; base += 64) {
                const int e = base + lane < qn ? wq[base + lane] : -1;
                bool tied = false;
                if (e >= 0) tied = slow_member(e >> 4, e & 15, half, stale, evals, nchanged, hv_lane, kmin_lane, smin_lane);
                if (TIE && cand_g) cand_push(tied, (int)((unsigned)e | kCandAlways));      // (the detector's walk found it: not a matter of equal values, the second look decides)
            }
            qn = 0;
            if (TIE) cand_flush();
            if (half == 1) {
                const unsigned hv = wave_sum(hv_lane);
                const float kmin = wave_min(kmin_lane), smin = wave_min(smin_lane);
                if (lane == 0) {
                    if (hv) atomicAdd(reinterpret_cast<unsigned*>(&sc[BC_HASH]), hv);
                    if (kmin < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[BC_TMIN]), bf2u(kmin));
                    if (smin < kInf) atomicMin(reinterpret_cast<unsigned*>(&sc[BC_STALEMIN]), bf2u(smin));
                }
            }
            __syncthreads();
            DSA_BCLK(1 + half);
        }
        if (tid == 0) {
            {   // the slow queue's share of this round's member evaluations (regular media: 1-2.5 %)
                const int slown = sc[BC_SLOWN];
                sc[BC_SLOWN] = 0;
                if (!sc[BC_FARALL]) {
                    if (slown * 12 > (nready_even + nready_odd) * nmem && nready_even + nready_odd >= 64) { if (++far_strikes >= kFarStrikes) sc[BC_FARALL] = 1; }
                    else if (far_strikes > 0) --far_strikes;
                }
            }
            sc[BC_READY] = 0; sc[BC_READY_ODD] = 0; sc[BC_CUR] = 0;
            float tmin = bu2f((unsigned)sc[BC_TMIN]);
            {   // a stale change that stays at one node (see pass B): the window goes back there until the freeze has dealt with it
                const float sm = bu2f((unsigned)sc[BC_STALEMIN]);
                sc[BC_STALEMIN] = 0x7f800000;
                if (sm < kInf && (bf2u(sm) == bf2u(sm_prev) || bf2u(sm) == bf2u(sm_prev2))) ++sm_same; else sm_same = 0;
                sm_prev2 = sm_prev; sm_prev = sm;
                if (sm_same >= kStaleRounds) tmin = fminf(tmin, sm);
            }
            sc[BC_THETA] = (int)bf2u(tmin + p.window);
            sc[BC_TMIN] = 0x7f800000;
            const unsigned hsh = (unsigned)sc[BC_HASH];
            sc[BC_HASH] = 0;
            if (tmin > best_tmin && tmin < kInf) best_tmin = tmin;
            sc[BC_STALE] = (int)bf2u(best_tmin - kStaleWindows * p.window);
            const bool repeat = hsh != 0u && (hsh == hist[1] || hsh == hist[2] || hsh == hist[3] || hsh == hist[0]);
            hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = hsh;
            // (with the window pulled back to a cycle, see above: everything up to THAT window's edge has stopped moving, no more -- members
            // that lag may still be busy between there and the front)
            if (repeat) { if (++stall >= kBundleCycleRounds) { const float fz = (sm_same >= kStaleRounds ? sm_prev : best_tmin) + p.window;
                                                                 if (fz > bu2f((unsigned)sc[BC_FREEZE])) sc[BC_FREEZE] = (int)bf2u(fz); stall = 0; ++freezes; } }
            else stall = 0;
        }
        ++rounds;
        __syncthreads();
        DSA_BCLK(3);
        if (sc[BC_OVERFLOW]) break;                      // the exception table is full (info[2] = -2 at the pilot): the host grows it and solves the chunk again
        if (rounds > p.max_rounds) { dead = true; break; }
    }
    const bool failed = rounds > p.max_rounds;
#ifdef DSA_BUNDLE_CLOCKS
    if (tid == 0 && p.clocks) { p.clocks[0] = bt[0]; p.clocks[1] = bt[1]; p.clocks[2] = bt[2]; p.clocks[3] = bt[3]; }
#endif

    // ---- the members' ends: statistics, receiver times from the bundle's field (reference srtimes), the field itself where the engine
    // keeps one per unit (rays and rows, field downloads), then the slot goes to the next bundle
    {
        unsigned long long e64 = evals, c64 = nchanged;
        for (int o = 32; o > 0; o >>= 1) { e64 += __shfl_xor(e64, o); c64 += __shfl_xor(c64, o); }
        if (lane == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 4), e64); atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 6), c64); }
    }
    __threadfence_block();
    __syncthreads();
    if (TIE && !failed && !sc[BC_OVERFLOW]) {
        // ---- tie census of the converged field (see the head of the kernel): a wave takes every NW-th tile, 64 / (G/4) nodes per trip, a lane
        // four members of its node -- the node against its x+ and z+ neighbours, so that every pair of neighbours is looked at once
        constexpr int LPN = G / 4, NPI = 64 / LPN;
        const int csub = lane % LPN, cnode = lane / LPN;
        const BV4 inf4 = { kInf, kInf, kInf, kInf };
        int cq = 0;
        bool lost = false;                     // (a trip with more ties than the queue holds: every member of the bundle counts as tied)
        const int ncand = cand_g ? __hip_atomic_load(cand_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;      // (counted by atomics at the L2: not through this CU's L1)
        const bool by_list = cand_g && ncand >= 0 && ncand <= bd->cand_cap;
        auto census_flush = [&]() {          // (the one call site of the second look)
            lost = lost || cq > kSlowQ;
            cq = cq < kSlowQ ? cq : kSlowQ;
            for (int base = 0; base < cq; base += 64) {
                const int e = base + lane < cq ? wq[base + lane] : -1;
                if (e >= 0) tie_member(e >> 4, e & 15);
            }
            cq = 0;
        };
        if (by_list) {
            // the listed candidates: still tied in the converged field (the member's value equal, bit for bit, to a near neighbour's)?  then the node
            // and its partners -- a partner's own last evaluation may have come before the tie existed: it is not on the list itself -- go to the
            // second look; a wave takes 64 candidates at a time
            for (int i0 = wave * 64; i0 < ncand; i0 += NW * 64) {
                const int i = i0 + lane;
                const unsigned eraw = i < ncand ? (unsigned)cand_g[1 + i] : ~0u;
                const int e = eraw == ~0u ? -1 : (int)(eraw & ~kCandAlways);
                unsigned still = 0u;               // bit q: near neighbour q carries the member's value
                int nid[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                if (e >= 0) {
                    const int id = e >> 4, mo = e & 15;
                    if (eraw & kCandAlways) still |= 16u;
                    int iz, ix;
                    coords(id, &iz, &ix);
                    rec_stencil(nbz, id, nid);
                    const unsigned mb = (unsigned)mo * 4u;
                    const float raw = *(BGF32*)(Bb + (unsigned)id * GB + mb);
                    const float a = fabsf(raw);
                    if (a < kInf) {
                        // (round 6) the walk's tie is against the neighbour's ACCEPTANCE time: where that is not the value -- an exceptional node
                        // (sign bit) among the node, its near and its outer neighbours -- the values cannot tell, and the second look decides by itself
                        const bool in0 = ix > 0, in1 = ix + 1 < nnx, in2 = iz > 0, in3 = iz + 1 < nnz;
                        const float v0 = in0 ? *(BGF32*)(Bb + (unsigned)nid[0] * GB + mb) : kInf, v1 = in1 ? *(BGF32*)(Bb + (unsigned)nid[1] * GB + mb) : kInf;
                        const float v2 = in2 ? *(BGF32*)(Bb + (unsigned)nid[2] * GB + mb) : kInf, v3 = in3 ? *(BGF32*)(Bb + (unsigned)nid[3] * GB + mb) : kInf;
                        if (in0 && a == fabsf(v0)) still |= 1u;
                        if (in1 && a == fabsf(v1)) still |= 2u;
                        if (in2 && a == fabsf(v2)) still |= 4u;
                        if (in3 && a == fabsf(v3)) still |= 8u;
                        bool exc = __builtin_signbit(raw) || __builtin_signbit(v0) || __builtin_signbit(v1) || __builtin_signbit(v2) || __builtin_signbit(v3);
                        if (ix > 1) exc = exc || __builtin_signbit(*(BGF32*)(Bb + (unsigned)nid[4] * GB + mb));
                        if (ix + 2 < nnx) exc = exc || __builtin_signbit(*(BGF32*)(Bb + (unsigned)nid[5] * GB + mb));
                        if (iz > 1) exc = exc || __builtin_signbit(*(BGF32*)(Bb + (unsigned)nid[6] * GB + mb));
                        if (iz + 2 < nnz) exc = exc || __builtin_signbit(*(BGF32*)(Bb + (unsigned)nid[7] * GB + mb));
                        if (exc) still |= 16u;             // (the node itself goes to the second look; no partner is named)
                    }
                }
                if (__any(still != 0u)) {
                    if (cq + 5 * 64 > kSlowQ) census_flush();
#pragma unroll
                    for (int q = -1; q < 4; ++q) {
                        const bool on = q < 0 ? still != 0u : ((still >> (q < 0 ? 0 : q)) & 1u) != 0u;
                        const unsigned long long bal = __ballot(on);
                        const int pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                        if (on && pos < kSlowQ) wq[pos] = ((q < 0 ? (e >> 4) : nid[q < 0 ? 0 : q]) << 4) | (e & 15);
                        cq += __popcll(bal);
                    }
                }
            }
            // (round 6) the ties that values do not show -- against the acceptance time of an EXCEPTIONAL node (accepted later than its value), among them
            // the outer node accepted at the clock of the neighbour taken in last -- are found from the other end: the bundle's exception table names
            // every exceptional (node, member) there ever was (a few thousand); those that still are go to the second look with the eight nodes whose
            // walks look at them.  [Marking them in the round loop's slow pass instead cost 4 ms of the headline's 335: profiles/r06_ab_census.log.]
            {
                const int xn = 1 << xlog;
#ifndef DSA_XU
#define DSA_XU 4
#endif
                constexpr int XU = DSA_XU;        // (entries per lane and trip, fetched together: the table is 98 % empty and the walk is the latency of its loads)
                for (int i00 = wave * 64 * XU; i00 < xn; i00 += NW * 64 * XU) {
                  unsigned long long env[XU];
#pragma unroll
                  for (int xu = 0; xu < XU; ++xu) env[xu] = i00 + xu * 64 + lane < xn ? *exc_at((unsigned)(i00 + xu * 64 + lane)) : kExcEmpty;
#pragma unroll
                  for (int xu = 0; xu < XU; ++xu) {
                    const int i0 = i00 + xu * 64;
                    const unsigned long long en = env[xu];
                    const int kx = exc_key(en);
                    bool on = i0 + lane < xn && kx != -1 && !(kx & kExcPinned);
                    if (!__any(on)) continue;
                    const int key = kx & 0x3fffffff, id = on ? key / G : 0, mo = on ? key - id * G : 0;
                    int iz = 0, ix = 0, nid[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                    if (on) {
                        coords(id, &iz, &ix);
                        on = ix < nnx && iz < nnz && mo < nmem && __builtin_signbit(*(BGF32*)(Bb + (unsigned)id * GB + (unsigned)mo * 4u));      // (still exceptional)
                        rec_stencil(nbz, id, nid);
                    }
                    if (!__any(on)) continue;
                    const unsigned inq = (ix > 0 ? 1u : 0u) | (ix + 1 < nnx ? 2u : 0u) | (iz > 0 ? 4u : 0u) | (iz + 1 < nnz ? 8u : 0u) |
                                         (ix > 1 ? 16u : 0u) | (ix + 2 < nnx ? 32u : 0u) | (iz > 1 ? 64u : 0u) | (iz + 2 < nnz ? 128u : 0u);
#pragma nounroll
                    for (int q = -1; q < 8; ++q) {
                        if (cq + 64 > kSlowQ) census_flush();
                        const bool push = on && (q < 0 || ((inq >> q) & 1u) != 0u);
                        int node = id;
#pragma unroll
                        for (int j = 0; j < 8; ++j) node = q == j ? nid[j] : node;
                        const unsigned long long bal = __ballot(push);
                        const int pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                        if (push && pos < kSlowQ) wq[pos] = (node << 4) | mo;
                        cq += __popcll(bal);
                    }
                  }
                }
            }
        }
        // (kCU trips' loads -- the node, its x+ and its z+ neighbour, 16 bytes per lane each -- are issued before the first comparison: a lone
        // workgroup's census is bound by the latency of its loads, 23 ms per launch with two trips in flight, profiles/r05_ab_bundle_kernel.log)
        constexpr int kCU = 4, TPT = kTileRecs / NPI;
        const int ntrips = by_list ? 0 : ntile * TPT;
        for (int tb = wave * kCU; tb < ntrips; tb += NW * kCU) {
            int idv[kCU], nxv[kCU], nzv[kCU];
            BV4 own[kCU], vx[kCU], vz[kCU];
#pragma unroll
            for (int k = 0; k < kCU; ++k) {
                const int trip = tb + k < ntrips ? tb + k : ntrips - 1;
                const int tile = trip / TPT, r0 = (trip - tile * TPT) * NPI;
                const int id = (tile << 6) + r0 + cnode;
                int iz, ix;
                coords(id, &iz, &ix);
                const bool here = tb + k < ntrips && ix < nnx && iz < nnz;
                int nid[8];
                rec_stencil(nbz, id, nid);
                idv[k] = id; nxv[k] = nid[1]; nzv[k] = nid[3];
                own[k] = here ? *(BGV4*)(Bb + (unsigned)id * GB + (unsigned)csub * 16u) : inf4;
                vx[k] = (here && ix + 1 < nnx) ? *(BGV4*)(Bb + (unsigned)nid[1] * GB + (unsigned)csub * 16u) : inf4;
                vz[k] = (here && iz + 1 < nnz) ? *(BGV4*)(Bb + (unsigned)nid[3] * GB + (unsigned)csub * 16u) : inf4;
            }
#pragma unroll
            for (int k = 0; k < kCU; ++k) {
                unsigned tm = 0u;              // bit m: member m ties with the x+ neighbour, bit 4 + m: with the z+ neighbour
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float a = fabsf(own[k][m]);
                    if (a < kInf) { if (a == fabsf(vx[k][m])) tm |= 1u << m; if (a == fabsf(vz[k][m])) tm |= 16u << m; }
                }
                if (__any(tm != 0u)) {
                    if (cq + 3 * 4 * 64 > kSlowQ) census_flush();
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        // (the node itself, then the partner of each tie)
                        const bool p0 = ((tm >> m) & 0x11u) != 0u, p1 = ((tm >> m) & 1u) != 0u, p2 = ((tm >> (4 + m)) & 1u) != 0u;
                        const int mo = csub * 4 + m;
                        unsigned long long bal = __ballot(p0);
                        int pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                        if (p0 && pos < kSlowQ) wq[pos] = (idv[k] << 4) | mo;
                        cq += __popcll(bal);
                        bal = __ballot(p1);
                        pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                        if (p1 && pos < kSlowQ) wq[pos] = (nxv[k] << 4) | mo;
                        cq += __popcll(bal);
                        bal = __ballot(p2);
                        pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                        if (p2 && pos < kSlowQ) wq[pos] = (nzv[k] << 4) | mo;
                        cq += __popcll(bal);
                    }
                }
                // (round 6) an exceptional node -- accepted later than its value: the sign bit -- ties by its acceptance time, which the values
                // above cannot show: the node and the eight nodes whose walks look at it go to the second look (0.02 % of a field's nodes)
                unsigned xm = 0u;
#pragma unroll
                for (int m = 0; m < 4; ++m) if (__builtin_signbit(own[k][m]) && fabsf(own[k][m]) < kInf) xm |= 1u << m;
                if (__any(xm != 0u)) {
                    int iz, ix, nid[8];
                    coords(idv[k], &iz, &ix);
                    rec_stencil(nbz, idv[k], nid);
                    // (bit q: stencil node q lies inside the grid; a rare path: plain loops, the words picked by shifts, nothing unrolled)
                    const unsigned inq = (ix > 0 ? 1u : 0u) | (ix + 1 < nnx ? 2u : 0u) | (iz > 0 ? 4u : 0u) | (iz + 1 < nnz ? 8u : 0u) |
                                         (ix > 1 ? 16u : 0u) | (ix + 2 < nnx ? 32u : 0u) | (iz > 1 ? 64u : 0u) | (iz + 2 < nnz ? 128u : 0u);
#pragma nounroll
                    for (int m = 0; m < 4; ++m) {
                        if (!__any(((xm >> m) & 1u) != 0u)) continue;
#pragma nounroll
                        for (int q = -1; q < 8; ++q) {
                            if (cq + 64 > kSlowQ) census_flush();
                            const bool on = ((xm >> m) & 1u) != 0u && (q < 0 || ((inq >> q) & 1u) != 0u);
                            int node = idv[k];
#pragma unroll
                            for (int j = 0; j < 8; ++j) node = q == j ? nid[j] : node;
                            const unsigned long long bal = __ballot(on);
                            const int pos = cq + __popcll(bal & ((1ull << lane) - 1ull));
                            if (on && pos < kSlowQ) wq[pos] = (node << 4) | (csub * 4 + m);
                            cq += __popcll(bal);
                        }
                    }
                }
            }
        }
        census_flush();
        if (lost && lane < nmem) {
            const FimProblem* const pm = problems + s_member[lane];
            if (pm->tie) { atomicAdd((unsigned*)pm->tie, 1u); atomicMax((unsigned*)pm->tie + 1, bf2u(kInf)); }
        }
        __syncthreads();
    }
    for (int m = 0; m < nmem; ++m) {
        const FimEnds* const E = ends + s_member[m];
        const FimProblem* const pm = problems + s_member[m];
        if (tid == 0) {
            pm->info[0] = rounds; pm->info[1] = 0; pm->info[3] = m == 0 ? freezes : 0;      // (the statistic, counted once per bundle)
            if (TIE && pm->tie) { pm->tie[4] = freezes; pm->tie[7] = cand_g ? __hip_atomic_load(cand_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1; }      // ([7]: candidates the bundle listed -- above cand_cap: the census swept the field)                                      // (every member knows its bundle froze a cycle: a frozen 2-cycle sits an ulp or two from a tie state)
            if (failed && pm->info[2] != -2) pm->info[2] = -1;
            if (sc[BC_OVERFLOW]) pm->info[2] = -2;
        }
        if (E->rays) {
            const GridDesc g = E->g;
            for (int r = tid; r < E->nrays; r += NT) {
                const RayDesc rd = E->rays[r];
                if (!(rd.flags & kRayTime)) continue;
                float t;
                if (!receiver_time(g, E->scx, E->scz, rd, (const float*)Bslot + m, E->veln, E->dpl, &t, G * DSA_BSTRIDE)) atomicExch(E->err, E->ray0 + r + 1);
                E->out[rd.data] = t;
            }
        }
    }
    {   // the members' fields into their unit slots (the engine keeps one per unit when rays follow): one pass over the bundle's field,
        // a node's G values read as 16-byte pieces, every member's array written in node order
        __shared__ float* s_dst[kBundleMax];
        __syncthreads();
        if (tid < kBundleMax) s_dst[tid] = tid < nmem ? problems[s_member[tid]].Tc : nullptr;
        __syncthreads();
        bool any_dst = false;
        for (int m = 0; m < nmem; ++m) any_dst = any_dst || s_dst[m] != nullptr;
        if (any_dst)
            for (int i = tid; i < ntile * kTileRecs; i += NT) {
#pragma unroll
                for (int c4 = 0; c4 < G / 4; ++c4) {
                    const BV4 v = *(BGV4*)(Bb + (unsigned)i * GB + (unsigned)c4 * 16u);
                    BGF32* d;
                    if ((d = (BGF32*)s_dst[4 * c4 + 0])) d[i] = fabsf(v.x);
                    if ((d = (BGF32*)s_dst[4 * c4 + 1])) d[i] = fabsf(v.y);
                    if ((d = (BGF32*)s_dst[4 * c4 + 2])) d[i] = fabsf(v.z);
                    if ((d = (BGF32*)s_dst[4 * c4 + 3])) d[i] = fabsf(v.w);
                }
            }
    }
    if (bd->slot_busy) {
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&bd->slot_busy[my_slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

size_t bundle_lds_bytes(int tile_words) { return (size_t)tile_words * 4; }

void launch_fim_bundles(const FimBundle* d_bundles, int nbundles, int G, int threads, const FimProblem* d_problems, const FimEnds* d_ends, int tile_words, hipStream_t stream, int members_per_lane, bool tie)
{
    if (nbundles <= 0) return;
    const size_t lds = bundle_lds_bytes(tile_words);
#define DSA_LAUNCH_BUNDLE_T(GG, TT, MM, TIE_) hipLaunchKernelGGL((k_fim_bundle<GG, TT, MM, TIE_>), dim3(nbundles), dim3(TT), lds, stream, d_bundles, d_problems, d_ends)
#define DSA_LAUNCH_BUNDLE_G(TT, MM, TIE_) { if (G == 16) DSA_LAUNCH_BUNDLE_T(16, TT, MM, TIE_); else if (G == 8) DSA_LAUNCH_BUNDLE_T(8, TT, MM, TIE_); else DSA_LAUNCH_BUNDLE_T(4, TT, MM, TIE_); }
#define DSA_LAUNCH_BUNDLE(TT, MM) { if (tie) DSA_LAUNCH_BUNDLE_G(TT, MM, true) else DSA_LAUNCH_BUNDLE_G(TT, MM, false) }
    if (members_per_lane == 2 && threads == 256) DSA_LAUNCH_BUNDLE(256, 2)
    else if (threads == 512) DSA_LAUNCH_BUNDLE(512, 4)
    else if (threads == 768) DSA_LAUNCH_BUNDLE(768, 4)
    else DSA_LAUNCH_BUNDLE(256, 4)
#undef DSA_LAUNCH_BUNDLE_T
#undef DSA_LAUNCH_BUNDLE_G
#undef DSA_LAUNCH_BUNDLE
}

// slowI[id * np + m] = slow_all[m * field_stride + id]: the maps' slowness, member-minor
__global__ void k_interleave_maps(const float* __restrict__ slow_all, size_t field_stride, int np, float* __restrict__ slowI)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= field_stride * (size_t)np) return;
    const size_t id = i / (size_t)np;
    const int m = (int)(i - id * (size_t)np);
    slowI[i] = slow_all[(size_t)m * field_stride + id];
}

// ---- the refined boxes in bundles (round 5) -----------------------------------------------------------------------------------------------
// slowI[(b * nrec + id) * G + m] = slowness of member m of bundle b at record id, from the member's own tiled slowness (FimProblem::slow)
template <int G>
__global__ void k_bundle_refined_slowness(const FimBundle* __restrict__ bundles, const FimProblem* __restrict__ problems, int nrec, float* __restrict__ slowI)
{
    const FimBundle* const bd = bundles + blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec * G) return;
    const int id = i / G, m = i - id * G;
    const int mm = m < bd->nmem ? m : 0;
    slowI[((size_t)blockIdx.y * nrec + id) * G + m] = problems[bd->member[mm]].slow[id];
}
// the converged members of the bundles back into their (T, tau) records (FimProblem::F), as the unit-by-unit refined solve leaves them: T with the
// sign bit of a pinned node, tau the acceptance time (the value itself unless the node is in the bundle's exception table)
template <int G>
__global__ void k_bundle_export_records(const FimBundle* __restrict__ bundles, const FimProblem* __restrict__ problems, int nrec)
{
    const FimBundle* const bd = bundles + blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec * G) return;
    const int id = i / G, m = i - id * G;
    if (m >= bd->nmem) return;
    // (round 6) a unit whose serial start-up march ended the refined stage by itself (a source in the last cell before an open edge) has nothing for the
    // fixed point to do, and its records hold that march's trial values, which the hand-off reads: they stay.  [Overwritten by the bundle's empty field
    // they left the hand-off without a band, the coarse solve without a seed, and the call returned zeros for the unit -- silently: tools/edge_probe.py.]
    const FimProblem* const pm = problems + bd->member[m];
    if (pm->ended && *pm->ended) return;
    const float v = bd->B[(size_t)bd->slot * bd->b_stride + (size_t)id * G * DSA_BSTRIDE + m];
    Rec r{ v, v };
    if (__builtin_signbit(v)) {
        bool pinned = false;
        const float tau = exc_find(bd->exc + (size_t)bd->slot * bd->exc_stride, bd->exc_log2cap, id * G + m, &pinned);
        r.T = pinned ? v : -v; r.tau = tau;
    }
    pm->F[id] = r;
}

void launch_bundle_refined_slowness(const FimBundle* d_bundles, int nbundles, int G, const FimProblem* d_problems, int nrec, float* d_slowI, hipStream_t stream)
{
    if (nbundles <= 0) return;
    const dim3 grid((unsigned)((nrec * G + 255) / 256), (unsigned)nbundles), block(256);
    if (G == 16) hipLaunchKernelGGL(k_bundle_refined_slowness<16>, grid, block, 0, stream, d_bundles, d_problems, nrec, d_slowI);
    else if (G == 8) hipLaunchKernelGGL(k_bundle_refined_slowness<8>, grid, block, 0, stream, d_bundles, d_problems, nrec, d_slowI);
    else hipLaunchKernelGGL(k_bundle_refined_slowness<4>, grid, block, 0, stream, d_bundles, d_problems, nrec, d_slowI);
}

void launch_bundle_export_records(const FimBundle* d_bundles, int nbundles, int G, const FimProblem* d_problems, int nrec, hipStream_t stream)
{
    if (nbundles <= 0) return;
    const dim3 grid((unsigned)((nrec * G + 255) / 256), (unsigned)nbundles), block(256);
    if (G == 16) hipLaunchKernelGGL(k_bundle_export_records<16>, grid, block, 0, stream, d_bundles, d_problems, nrec);
    else if (G == 8) hipLaunchKernelGGL(k_bundle_export_records<8>, grid, block, 0, stream, d_bundles, d_problems, nrec);
    else hipLaunchKernelGGL(k_bundle_export_records<4>, grid, block, 0, stream, d_bundles, d_problems, nrec);
}

void launch_interleave_maps(const float* d_slow_all, size_t field_stride, int np, float* d_slowI, hipStream_t stream)
{
    const size_t n = field_stride * (size_t)np;
    if (n == 0) return;
    hipLaunchKernelGGL(k_interleave_maps, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_slow_all, field_stride, np, d_slowI);
}

}  // namespace dsa
