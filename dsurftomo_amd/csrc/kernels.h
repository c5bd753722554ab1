// Internal interface between the engine (host) and the HIP kernels of the CalSurfG path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "source_stage.h"

namespace dsa {

// data = 0-based index of the datum (travel time / Frechet row) a ray belongs to;
// flags: kRayTime = write the receiver time, kRayPath = trace the ray and emit its row
struct RayDesc { int src; float rx, rz; float sin_rx; int data; int flags; };   // sin_rx = libm sinf(rx), made on the host
constexpr int kRayTime = 1, kRayPath = 2;

constexpr int kClockSlots = 32;     // probe counters per unit (FimProblem::clocks)
// The tie record of a unit (FimProblem::tie): kTieWords int32, the refined stage's half first.  A tie's influence enters the sum as
// min(influence, 1 s) / kTieSumUnit, saturating at 2^32 - 1 (4.3 s of influences: nothing that matters is that large).
constexpr int kTieWords = 16;
constexpr float kTieSumUnit = 0x1p-30f;
constexpr int kTieSeenLog2 = 14, kTieSeenSlots = 1 << kTieSeenLog2;   // the census' set of (node, member) pairs it has looked at: an open-addressing table per bundle slot, behind the candidate list

// One fixed-point problem: a travel-time field on an (nnz, nnx) grid stored as tiled (T, tau)
// records (eikonal_core.h).  The records carry the boundary condition: pinned nodes (sign bit of T)
// are never recomputed, every other node starts at +inf.  `seed` lists the record indices to
// evaluate first (their queued bit, the sign bit of tau, is already set).
struct FimProblem {
    Rec* F;                // refined problems: tiled (T, tau) records
    float* Tc;             // coarse problems: the compact field (one float per node, eikonal_core.h) ...
    unsigned long long* exc;   // ... and its exception table of 2^exc_log2cap entries
    int exc_log2cap;
    const float* slow;     // tiled like the field
    const float* risti;
    const int* seed;
    const int* seed_count;
    int seed_cap;          // entries of `seed`; a larger count means: collect the queued nodes from the field
    int* lists;            // scratch for the active lists: 2 lists of list_cap (node, key) pairs + ready_cap ints
    int nnx, nnz, nbx, nbz;
    float ri, dnx, dnz;
    float window;          // causal window (seconds of travel time) evaluated per round
    int max_rounds;
    unsigned long long* clocks;   // optional, kClockSlots u64: [0..7] phase clocks of thread 0 (wall_clock64 ticks) + list sizes; [8..31] trip counters of DSA_LEDGER builds
    int32_t* info;         // 8 ints: [0] rounds, [1] rescans, [2] -1 = no convergence, [3] stall freezes, [4..5] evaluations (u64)
    int32_t* tie;          // tie detector (kernels launched with FimLaunch::tie), kTieWords / 2 words per stage: [0] ties whose influence exceeds tie_threshold, [1] largest influence (float bits),
                           // [2] ties with any influence at all, [3] the sum of the influences in units of kTieSumUnit (round 6: sub-threshold ties add up along a front),
                           // [4] cycles the unit's bundle froze (bundle kernel), [5] ties without an influence on their node (bundle kernel's census),
                           // [6] refined half: rank ties at the hand-off that change a STATUS the coarse grid receives, or more of them than the probe holds (k_handoff_probe); coarse half: the band march left its tree no heap (k_coarse_march), [7] tie candidates the unit's bundle listed (bundle kernel; a statistic)
    float tie_threshold;
    const int32_t* ended;  // refined problems: the unit's flag "the serial start-up march ended the refined stage by itself" (SourceScratch::flags[0]) -- its records then
                           // hold the march's own trial values, which nothing may overwrite (round 6: k_bundle_export_records did); coarse problems: null
};

// What the COARSE solve does around its fixed point: the workgroup owns its unit's field slot from the first store to the last read.
//   start: (recycled slots) wait until the previous user of the slot is done; every node unreached (+inf), exception table empty, the
//          serial prologue's pinned nodes (records of the coarse march window, k_coarse_march) into the field and the table;
//   end:   the unit's receiver times (reference srtimes) when `rays` is set -- the field is then not needed by anyone else -- and the
//          slot is handed on.
// Round 3: this replaces the global fill of all fields before the launch (67 GB per step at the headline size) and lets a launch
// of 16 000 units run on a pool of a few thousand field slots.
struct FimEnds {
    const Rec* W;            // window records, (cwnz, cwnx) column-major
    int cwz0, cwx0, cwnz, cwnx;
    const Rec* Fpin;         // (round 5, the refined boxes solved in bundles) non-null: the unit's tiled (T, tau) records -- the bundle takes its pinned nodes
                             // from them instead of a window, and k_bundle_export_records writes the converged member back into them
    int* slot_busy;          // recycled field slots: the pool's busy flags (the workgroup claims a free slot by compare-and-swap and
    int nslots;              // clears the flag when done); null: the slot named in the FimProblem is this unit's alone
    float* Tc_pool;          // the pool's arrays: slot q at Tc_pool + q * (tile records of the grid), exc_pool + (q << exc_log2cap),
    unsigned long long* exc_pool;      // lists_pool + q * lists_stride
    int* lists_pool;
    unsigned lists_stride;
    const RayDesc* rays;     // the unit's receivers; null: the receiver kernel (k_srtimes) runs after the launch
    int nrays, ray0;         // ray0: index of the first of them in the plan (error reporting)
    const float* veln;       // row-major velocity grid of the unit's period
    float scx, scz, dpl;
    float* out;
    int32_t* err;
    GridDesc g;
};

#ifndef DSA_ODD_CLEAR
#define DSA_ODD_CLEAR 1            // fim_kernel.hip: 1 = tile records {E, O, R, round} (default), 0 = one mask per tile (see there)
#endif
// DSA_KEY_MASKS (round 3): the tile record also says HOW a node was activated -- by a neighbour whose acceptance time already lay inside
// the causal window (the node's lower bound is then inside it too: it goes to the ready list without a look at its neighbourhood) or by a
// later one (lower bound from four loads, as before).  Record {E, O, R, DE, DO, round} in 64 bytes; the schedule itself is unchanged.
// MEASURED AND SWITCHED OFF (profiles/r03_ab_key_masks.txt): -7 % solves/s at 16 384 units (smooth), -9 % (checkerboard).  55 % of the listed
// nodes skip their loads, but pass A then expands two masks per tile and routes in two passes, the record doubles (64 B), and the kernel
// spills more (10 VGPRs / 51 SGPRs against 4 / 32).  Kept as a build variant for the record.
#ifndef DSA_KEY_MASKS
#define DSA_KEY_MASKS 0
#endif
constexpr int kFimMaskInts = DSA_KEY_MASKS ? 16 : (DSA_ODD_CLEAR ? 8 : 2);     // ints of active-set record per tile in the per-problem scratch

struct FimLaunch {
    int list_cap;          // entries per active list
    int ready_cap;         // entries of the dense ready list
    int threads;           // workgroup size: 256, 512 or 1024
    int sorted;            // 1: k_fim_sorted (tile masks + LDS tile bitmap), 0: k_fim (lists in activation order; refined problems only)
    int compact;           // 1: coarse problems on the compact field (k_fim_sorted only)
    int tile_words;        // words of the LDS tile bitmap (sorted variant)
    int lds_pad;           // extra dynamic LDS per workgroup (bytes): limits the workgroups resident per CU
    int tie;               // 1: the variant with the tie detector (k_fim_sorted only; eikonal_core.h solve_node_t<true>)
};

// A BUNDLE of coarse problems (round 3, bundle_kernel.hip): up to kBundleMax units of ONE source -- its periods -- solved by one workgroup
// under one shared round schedule.  The members keep their own arithmetic (slowness, travel times, exception entries); what they share is
// the active set (tile records, ready lists, causal window: routed by member 0, the pilot) and, per evaluated node, the addresses and the
// stencil geometry.  The field is member-minor: B[id * G + m] (one 64-byte segment per node at G = 16), so every byte of a fetched line is
// used; the slowness comes from the maps' member-minor copy slowI[id * np + map].  The fixed point is schedule independent
// (tests/tools/bundle_lab.cpp measures what sharing costs: +4...+9 % evaluations for maps that are multiples of one pattern, +13...+20 % for
// mixtures of patterns, +56 % for unrelated random maps), so each member's field is the one its own solve produces.
constexpr int kBundleMax = 16;
#ifndef DSA_BSTRIDE
#define DSA_BSTRIDE 1          // probe builds (-DDSA_BSTRIDE=2): the nodes of a bundle field twice as far apart, i.e. one 64-byte segment per 128-byte line (tools/bytes_probe.sh)
#endif
struct FimBundle {
    // the pool of bundle field slots (round 3, late: a bundle CLAIMS a free slot when it starts -- `slot_busy`, one flag per slot -- instead
    // of being given one by number: with the bundles launched longest first, "slot = number mod slots" made the second generation wait
    // for the longest bundles of the first, +37 % at the headline size with 768 slots for 1000 bundles; claimed slots need no more than the
    // bundles resident at a time)
    float* B;                     // slot 0: G floats per node record, tiled like the compact field; then member 0's values once more, one float
    size_t b_stride;              //         per node (what pass A routes by), at B + G * nrec; floats per slot
    unsigned long long* exc;      // slot 0: exception table of the bundle, keyed by id * G + m
    size_t exc_stride;            //         entries per slot = 2^exc_log2cap
    int exc_log2cap;
    int* lists;                   // slot 0: tile records of the shared active set (kFimMaskInts ints per tile)
    size_t lists_stride;
    size_t p_offset;              // floats from a slot's B to its P
    int* slot_busy;               // null: the bundle owns slot `slot` (as many slots as bundles)
    int nslots, slot;
    const float* slowI;           // member-minor slowness of all maps: slowI[id * np + map]
    int np;
    int* cand;                    // slot 0: tie candidates of the bundle (TIE kernels): [0] their number, then (node << 4 | member) words; null: none kept (the census sweeps the field)
    size_t cand_stride;           //         ints per slot
    int cand_cap;                 //         entries per slot; behind them kTieSeenSlots words: the (node, member) pairs the census has looked at (each once)
    int cand_list;                // 1: the round loop lists candidates; 0: the census sweeps the converged field (option tie_list = 0, A/B)
    int far_all;                  // 1: pass A asks for all four outer neighbours of every node (option bundle_far_all; A/B of round 5's upwind-only loads)
    int nmem;
    int member[kBundleMax];       // indices into the launch's FimProblem / FimEnds arrays (grid, seeds, window records, receivers, info)
    int map[kBundleMax];
};
size_t bundle_lds_bytes(int tile_words);
// G = 16, 8 or 4 members per bundle slot (members beyond nmem idle)
void launch_fim_bundles(const FimBundle* d_bundles, int nbundles, int G, int threads /* 256, 512 or 768 */, const FimProblem* d_problems, const FimEnds* d_ends, int tile_words, hipStream_t stream,
                        int members_per_lane = 4 /* 4, or 2 (256 threads): half the live values per lane, twice the lanes per node */,
                        bool tie = false /* the tie detector's variant (exact_ties = 1) */);
// slowI[id * np + m] = slow_all[m * field_stride + id]
void launch_interleave_maps(const float* d_slow_all, size_t field_stride, int np, float* d_slowI, hipStream_t stream);

size_t fim_lds_bytes(const FimLaunch& l);
void launch_fim(const FimProblem* d_problems, int nproblems, const FimLaunch& l, hipStream_t stream, const FimEnds* d_ends = nullptr);

// period-level tables ---------------------------------------------------------------------------
// velv: fp32 vertex values (ny, nx); basis: (gd+1) x 4; outputs veln (nnz, nnx row-major, for the
// receiver / ray kernels) and slow (tiled, for the solve)
void launch_gridder(const GridDesc& g, const float* d_velv, const float* d_basis, float* d_veln, float* d_slow,
                    hipStream_t stream);

// per-source stages ------------------------------------------------------------------------------
struct BatchPtrs {
    const SourceDesc* src;       // [nsrc]
    // refined, per source: tiled slowness / records (stride kRefRecs), row-major results (stride kRefMax^2)
    float* slow_r; Rec* F_r; float* Tfin_r; int8_t* S_r;
    float* risti_r;              // stride kRefMax (uploaded by the host)
    float* vcorner;              // stride 4
    int* seed_r; int* nseed_r;   // stride kSeedR / 1
    int16_t* rst;                // stride kRWin*kRWin
    int16_t* cst; int8_t* cinit; // stride kCWinMax*kCWinMax
    int32_t* heap;               // stride kHeapCap
    int32_t* flags;              // stride 4: [0] ended early, [1] error, [2] e* iz, [3] e* ix
    // coarse, per source
    float* T_c;                  // compact coarse field, stride nbx*nbz*64 (tiled, one float per node)
    unsigned long long* exc_c;   // exception tables of the coarse fields, stride 2^exc_log2cap
    int exc_log2cap;
    Rec* W_c;                    // (T, tau) records of the coarse march window, stride kCWinMax*kCWinMax
    int* seed_c; int* nseed_c;   // stride kSeedC / 1
    int* lists; size_t lists_stride;   // active-list scratch of the refined solve, per unit
    int* lists_c; size_t lists_c_stride;   // tile records of the coarse solve, per field slot
    // field slots of the coarse solve: T_c / exc_c / lists_c hold `pool` slots; pool >= units of the launch: unit s owns slot s for the whole
    // chunk; fewer: a workgroup of the launch claims a free slot (busy flags pool_gen) and frees it when its unit is done
    int pool;
    int* pool_gen;
};
constexpr int kSeedR = kRWin * kRWin;              // the start-up march cannot pin more than its window
constexpr int kSeedC = kCWinMax * kCWinMax + 4 * kCWinMax;    // every node of the window at most once, plus its outer rim

void launch_refine(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_velv_all, size_t velv_stride,
                   const float* d_rbasis, hipStream_t stream);
void launch_refined_startup(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream);
void launch_handoff(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream, int32_t* d_tie = nullptr /* the units' tie records: the hand-off probes its rank ties */, float tie_threshold = 0.0f,
                    int32_t* d_replay = nullptr /* [0] count (zeroed by the caller), [1 ..] units whose refined box is marched literally: k_handoff_replay */, int replay_cap = 0,
                    void* d_replay_scratch = nullptr /* replay_cap x handoff_replay_bytes() */, int32_t* d_xinfo = nullptr /* four words per unit: the replayed marches' accepts and guards */);
constexpr int kReplayHeap = 4096;                    // tree slots of a replayed refined march beyond its LDS part (narrow bands of a 129^2 box: a few hundred)
constexpr size_t handoff_replay_bytes() { return (size_t)kReplayHeap * 8; }      // per listed unit: tree entries
void launch_refined_replay(const GridDesc& g, const BatchPtrs& b, const int32_t* d_list, int cap, void* d_heap, int gcap, int32_t* d_xinfo, hipStream_t stream);     // exact_kernel.hip
void launch_coarse_march(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                         size_t field_stride, const float* d_risti_c, hipStream_t stream, int32_t* d_tie = nullptr /* the units' tie records: a band march that leaves its tree no heap says so */);
void launch_make_problems(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                          size_t field_stride, const float* d_risti_c, float window_r, float window_c,
                          FimProblem* d_prob_r, FimProblem* d_prob_c, int32_t* d_info, unsigned long long* d_clocks,
                          const int* d_launch_rank, int32_t* d_tie, float tie_threshold, FimEnds* d_ends_c, const RayDesc* d_rays /* null: no receiver times inside the solve */,
                          const float* d_veln_all, size_t veln_stride, float dpl, float* d_out, int32_t* d_err,
                          const int* d_member_flag /* null or per unit: 1 = solved inside a bundle */, float window_b /* causal window of the bundles */,
                          int max_rounds_b /* > 0: round limit of the bundles (tests of the fallback) */, hipStream_t stream,
                          float window_t = 0.0f /* > 0: causal window of the members flagged 2 (the wide bundles behind a launch's first generation) */,
                          FimEnds* d_ends_r = nullptr /* non-null: the refined problems go to d_prob_r by launch rank like the coarse ones, with these ends (Fpin) beside them: the refined boxes of bundled units are solved in bundles too */);
// the refined boxes in bundles: member-minor slowness of a bundle's members from their own tiled slowness (problems[member].slow), and the converged members
// back into their (T, tau) records (problems[member].F) for the hand-off
void launch_bundle_refined_slowness(const FimBundle* d_bundles, int nbundles, int G, const FimProblem* d_problems, int nrec, float* d_slowI, hipStream_t stream);
void launch_bundle_export_records(const FimBundle* d_bundles, int nbundles, int G, const FimProblem* d_problems, int nrec, hipStream_t stream);

// exact mode (exact_kernel.hip): the reference's Fast Marching replayed for the chunk-local units d_units[0..n), FOUR units per wavefront
// (a group of sixteen lanes each), unit j marching in pool slot j (pool_stride records of 8 bytes per slot; gcap tree slots of 8 bytes per
// slot beyond the lcap -- odd -- kept in LDS; d_starts / d_nstart: exact_start_bytes() + 4 bytes per slot for the coarse stage's starting tree);
// launches: reset, refined march, snapshot + hand-off, coarse march, then the compact copy and / or the batch's receiver times.
// xinfo[4 u ..]: accepts of the refined / coarse stage, error code (1 tree capacity)
size_t exact_lds_bytes(int lcap);
size_t exact_heap_blocked_entries(int lcap, int gcap);
size_t exact_start_bytes();
// (round 5) POOLED TILES for times-only calls on large grids: a marching unit keeps only the 8x8-node tiles its narrow band has touched and not yet
// left behind (exact_kernel.hip: xg_tile_*), tcap of them, instead of a word per node of the whole grid -- 2.6 MB instead of 67 MB at 4097^2.  The
// batch's arrays, unit slot j at offset j * (entries per unit): tt = exact_tile_table_entries(ntile) two-byte entries, tp = tcap * 64 words, ring =
// tcap words, freestk = tcap two-byte entries, pins = ceil(ntile / 32) words
struct XTiles { void* tt; void* tp; void* ring; void* freestk; void* pins; int tcap; };
size_t exact_tile_table_entries(int ntile);
size_t exact_tile_unit_bytes(int ntile, int tcap);
// the batch's receiver times from the marched fields themselves (RayDesc::src indexes the resident chunk like d_units does)
struct XReceivers { const RayDesc* rays; const float* veln_all; size_t veln_stride; float dpl; float* out; int32_t* err; };
void launch_exact(const GridDesc& g, const BatchPtrs& b, const int* d_units, int n, const float* d_slow_all, size_t field_stride,
                  const float* d_risti_c, void* d_pool, size_t pool_stride, void* d_heap_pool, int gcap, int lcap, void* d_starts, int* d_nstart,
                  int32_t* d_xinfo, unsigned long long* d_clocks /* probe builds (DSA_X_CLOCKS): cycle counts per phase of the accept step */,
                  const XReceivers* receivers /* null: no receiver times here */, bool compact_copy /* the units' compact fields into BatchPtrs::T_c */, hipStream_t stream,
                  const XTiles* tiles = nullptr /* pooled tiles on the propagation grid (times-only calls: needs `receivers`, no compact copy) */,
                  int gstride = 0 /* entries per unit in d_heap_pool; 0: gcap */, int lb = 0 /* > 0: lcap = 2^lb - 1, global tree part in blocks (exact_heap_blocked_entries) */);

// receivers: one thread per ray; reference srtimes (CalSurfG.f90:1636-1759)
// RayDesc::src is a global unit index; unit_base is the first unit held by the batch arrays
void launch_srtimes(const GridDesc& g, const BatchPtrs& b, int unit_base, const RayDesc* d_rays, int nrays,
                    const float* d_veln_all, size_t field_stride, float dpl, float* d_out, int32_t* d_err,
                    hipStream_t stream);

// rays and Frechet rows (ray_kernels.hip) ------------------------------------------------------------
// trace_ids: indices into d_rays of the rays to trace (flag kRayPath), `n` of them; ray t of the
// launch owns slab t (slab_stride floats, zeroed by the caller) and rayinfo[2t..2t+1] = (flags, steps); lanes_per_ray: 1, or 4 (small launches)
void launch_rays(const GridDesc& g, const BatchPtrs& b, int unit_base, const RayDesc* d_rays, const int* d_trace_ids, int n,
                 const float* d_veln_all, size_t field_stride, float dpl, float* d_slabs, size_t slab_stride,
                 int32_t* d_rayinfo, int32_t* d_err, float* d_paths, int path_cap, int* d_path_n, hipStream_t stream, int lanes_per_ray = 1);

// S[(k * kmax + slot) * ncol + c] = (sen_vp * coe_a + sen_rho * coe_rho) + sen_vs, the depth-kernel
// factor of a Frechet row entry (reference CalSurfG.f90:1395-1423); vels: (nz, ny*nx) fp32
void launch_sen_combine(int ncol, int kmax, int nz, const float* d_vels, const double* d_sen_vs, const double* d_sen_vp,
                        const double* d_sen_rho, int shallow, double* d_S, hipStream_t stream);

struct RowArgs {
    const RayDesc* rays; const int* trace_ids; int n;      // the rays of this launch
    const SourceDesc* src; int unit_base;
    const float* slabs; size_t slab_stride;
    int* vlist; size_t vlist_stride; int* nv;              // kept vertices per ray (scan order jj outer, kk inner)
    const double* S; int kmax, nz;
    int* counts;                                           // entries per ray
    const long long* offsets;                              // exclusive scan of counts (64-bit: a launch may hold every ray of a call)
    float* rw; int* iw; int* col;                          // COO out: value, 1-based row, 1-based column
};
void launch_row_list(const GridDesc& g, const RowArgs& a, hipStream_t stream);
void launch_row_emit(const GridDesc& g, const RowArgs& a, bool write, hipStream_t stream);
// offsets[0..n] = exclusive prefix sums of counts[0..n-1]
void launch_scan(const int* d_counts, int n, long long* d_offsets, hipStream_t stream);

// dispersion (disp_kernels.hip) ---------------------------------------------------------------------
struct LayerGeom;
// curves: (npert, kmax, ncol) fp64; ws: 4 * rmax * nlanes floats of layer workspace; iwave 1 Love, 2 Rayleigh
// d_fail_list (or null): the curves without a root, slot by slot as they report (curve << 8 | period index, up to fail_cap of them)
void launch_dispersion(int iwave, const LayerGeom* d_geom, const float* d_vels, int ncol, int npert, int igr, int kmax,
                       const double* d_t, float* d_ws, size_t nlanes, double* d_curves, int rmax, int layers_in_lds, int gshift,
                       unsigned long long* d_diag /* [0] curves without a root, [1] min of (curve << 16 | period index) */,
                       unsigned long long* d_fail_list, int fail_cap, hipStream_t stream);
// One curve once more on the host, with what the reference's unit-66 block prints about a curve that ended without a root
// (surfdisp96.f:327-337): vs = the column's nz grid values, pert = the perturbation index of k_dispersion (0 = the model itself).
// table: 4 x 200 floats (d, a, b, rho of the flattened layers), c: 60 doubles (roots of the periods before k).  Returns the period index
// k the curve failed at (0: it did not fail).
int disp_replay_failure(const LayerGeom& G, const float* vs, int pert, int iwave, int igr, int kmax, const double* t,
                        int* mmax, float* table, double* cc_cm_c1, double* c);
void launch_depth_kernels(const float* d_vels, int ncol, int nz, int kmax, const double* d_curves, int with_kernels, double* d_pv,
                          double* d_sen_vs, double* d_sen_vp, double* d_sen_rho, int kmax_total, int slot0, hipStream_t stream);
void launch_to_float(const double* d_in, float* d_out, size_t n, hipStream_t stream);

}  // namespace dsa
