// Engine state behind the opaque dsa_engine handle.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "host_geometry.h"
#include "kernels.h"

namespace dsa {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
};

struct SpmvState;

// Defaults of the tie handling (round 5; DESIGN.md "Modes").  exact_ties = 1: the fixed point, a census of its exact ties, and the units
// that hold a tie whose influence on its node exceeds tie_threshold solved again by the reference's march itself.  Where nothing is flagged
// the census costs a few per cent of the solve; a flagged unit costs a march -- one unit's accepts are sequential, 2.3 us each: 35 ms at
// 121^2, 2.4 s at 1025^2, whatever the number of units marching beside it.  tie_threshold 2e-5 s: below it the ties of a smooth medium
// (one ulp of a 100 s travel time is 7.6e-6 s) would flag half of the headline's units for differences that stay inside 1e-4 s
// (profiles/r05_ties_headline.log); it is a heuristic, not a bound -- the guarantee is exact_ties = 2.  exact_ties = 0 opts out
// (DSA_EXACT_TIES=0 for an unchanged Fortran host); the census still runs (tie_detect) and the call reports what it would have flagged.
constexpr int kDefaultExactTies = 1;
constexpr float kDefaultTieThreshold = 2.0e-5f;
// (round 6) the census keeps the SUM and the COUNT of a unit's tie influences beside the largest one; the defaults of the rule that uses them
// are set from the scans under profiles/r06_tie_*: 0 = that part of the rule is off
constexpr float kDefaultTieSumThreshold = 0.0f;
constexpr int kDefaultTieCountThreshold = 0;
constexpr int kDefaultTieFrozenBundles = 0;
constexpr int kDefaultTieMapStrict = 1;
constexpr int kHandoffReplayCap = 256;       // units of a launch whose refined box the hand-off may march literally (the rest of such units are flagged: the whole unit marched)
constexpr float kTieUlpsAt1025 = 26.0f;      // what a receiver time of the fixed point differs by from the reference's downstream of one-ulp ties, at most, in ulps, on grids up to 1025 nodes per side (measured: 26 = 9.92e-5 s at 32-64 s, the worst of the first 2.2 M fuzzed units; two units of the next 0.7 M reached 36 and 27: not a bound; Engine::tie_verdicts)
// Rays of a launch up to which four lanes trace a ray together (ray_kernels.hip: launch_rays; profiles/r05_ab_rays.log)
constexpr int kRayGroupMax = 81920;

struct Engine {
    int device = 0;
    std::string arch;
    hipStream_t stream = nullptr;
    hipEvent_t events[8] = {};
    std::string error;
    int status = 0;

    // model
    GridDesc g{};
    int nmaps = 0;
    size_t nfield = 0;         // nodes of the coarse grid
    size_t nrec_c = 0;         // records of its tiled storage (padded to whole tiles)
    bool have_maps = false;
    float dpl = 0.0f;          // minimum cell width (km)
    float hmin_slow = 0.0f;    // smallest slowness of the maps (window scale)
    DevBuf<float> velv, veln, slow, risti_c, cbasis, rbasis;

    // plan
    bool planned = false;
    bool keep_fields = false;
    std::vector<SourceDesc> h_src;
    std::vector<float> h_risti_r;
    std::vector<RayDesc> h_rays;
    size_t mem_budget = 0;
    size_t per_unit_bytes = 0;
    int chunk = 0;
    int max_chunk = 0;
    float window_cells = 1.25f;        // causal window of the coarse solve in cell travel times (measured optimum 1.0-1.5)
    int list_cap = 0, ready_cap = 0;   // 0 = derive from the grid
    int last_chunk_first = -1, last_chunk_n = 0;
    bool fields_resident = false;      // the last chunk's coarse fields are still in their slots (not so after a launch that recycled them)

    DevBuf<SourceDesc> src;
    DevBuf<RayDesc> rays;
    DevBuf<float> out;
    DevBuf<int32_t> err;
    DevBuf<float> slow_r, Tfin_r, risti_r, vcorner;
    DevBuf<Rec> F_r, W_c;              // refined records; records of the coarse march windows
    DevBuf<float> T_c;                 // compact coarse fields (eikonal_core.h)
    DevBuf<unsigned long long> exc_c;  // their exception tables
    int exc_log2cap = 0;
    int exc_log2cap_opt = 0;           // option exc_log2cap: initial table size (0 = from the grid); the table grows by itself when it overflows
    int grown_nnx = 0, grown_nnz = 0;
    int exc_log2cap_grown = 0;         // ... and the size it grew to is kept for the later plans of this grid (an inversion solves the same geometry every iteration)
    DevBuf<int> seed_r, nseed_r, seed_c, nseed_c, lists, launch_rank;
    // field slots of the coarse solve (kernels.h: FimEnds): T_c, exc_c and lists_c hold `pool_slots` slots; a launch with more units than
    // slots recycles them (only when nobody needs the fields afterwards: no rows, no exact mode, no keep_fields)
    DevBuf<int> lists_c, pool_gen;
    DevBuf<int32_t> replay_list;       // (round 6) units whose refined box is marched literally behind the hand-off's probe: [0] count, [1 ..] units (kernels.h launch_handoff)
    DevBuf<unsigned char> replay_scratch;
    int handoff_replay = 1;            // option handoff_replay: 1 = as above; 0 = a hand-off tie that changes what the coarse grid receives flags the unit (the whole unit marched)
    DevBuf<FimEnds> ends_c;
    size_t lists_c_stride = 0;
    int pool_slots = 0;
    int field_pool_opt = 0;            // option field_pool: 0 = automatic (four times the workgroups the chip holds), -1 = one slot per unit, > 0 = that many slots
    size_t per_slot_bytes = 0;
    size_t plan_budget = 0;            // bytes plan() worked with (solve() grows the slot pool inside it when a call needs every field afterwards)
    std::vector<int> h_launch_rank;
    // bundles (bundle_kernel.hip, kernels.h: FimBundle): the units of one source -- its periods -- solved by one workgroup under one shared
    // round schedule.  Option `bundle`: 0 = off, 1 = automatic (default: 16, 8 or 4 members by the sources' unit counts and the memory),
    // 4 / 8 / 16 = that many members per bundle.  Default mode only (the tie detector and the literal march work per unit).
    int bundle_opt = 1;
    int ray_lanes_opt = 0;             // option ray_lanes: lanes per ray of the back-trace, 0 = automatic (4 for launches of up to kRayGroupMax rays, else 1), 1, 4
    float bundle_window_opt = 0.0f;    // option bundle_window_cells: causal window of the bundles, 0 = automatic (bundle_window(): 0.6 cells -- a round's fixed costs are shared by the members, so fewer evaluations per round pay; 1.25 for small wide launches)
    int bundle_G_now = 0;              // members per bundle of the current solve
                                       // (measured at 1025^2, 16 members: 0.4 / 0.5 / 0.6 / 0.8 / 1.25 cells -> 24.4 / 24.6 / 24.4 / 23.9 / 22.7 k solves/s)
    int bundle_threads_opt = 0;        // option bundle_threads: workgroup size of the bundle kernel (0 = automatic: 256; 768 beyond 1500 nodes per side and for small launches)
    int bundle_max_rounds = 0;         // option bundle_max_rounds (tests): > 0 = round limit of the bundles; a bundle that hits it sends its chunk to the unit-by-unit solve
    int bundle_mpl = 0;                // option bundle_members_per_lane: 0 = automatic (bundle_mpl_of), 4, or 2
    bool bundle_wide = false;          // this call's bundles run 768 threads wide on a small grid (choose_bundle_size: a CU per bundle)
    int bundle_mpl_now = 4, bundle_mpl_b = 2;      // ... what the current launch uses (whole bundles; the halved last ones)
    int bundle_threads_b = 256;                    // workgroup size of the second group of a launch (plan_bundles)
    int bundle_tail_opt = 1;                       // option bundle_tail: what a launch of 768 .. 1500 bundles does with the ones beyond the first generation: 1 (default) = whole and 768 threads wide when they are at most 256 (a CU each), 0 = cut in halves (256 threads; also beyond 256)
    int bundle_far_all = 0;                        // option bundle_far_all (A/B): 1 = every node trip fetches all four outer neighbours (round 4's loads)
    int bundles_a = 0, bundles_b = 0, bundle_Gb = 0;      // the launch's bundles: whole ones, and (plan_bundles) the last ones cut in halves of bundle_Gb members on a second stream
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_b0 = nullptr, ev_b1 = nullptr;
    hipError_t make_stream2();
    // Round 5: the REFINED boxes of bundled units are solved in bundles too (option bundle_refined, default on): the same member lists as the
    // coarse bundles, fields of the 129^2 box (1.3 MB per bundle), the members' own refined slowness member-minor, the converged members written back
    // into their (T, tau) records for the hand-off (kernels.h: FimEnds::Fpin, launch_bundle_*).  The unit-by-unit refined solves took 29 ms of the
    // headline step: 5 times the coarse bundles' cost per evaluation.
    int bundle_refined_opt = 1;
    bool refined_bundles_failed = false;      // a refined bundle ran out of rounds / table space: unit by unit until the maps change
    bool refined_bundles_now = false;
    std::vector<FimBundle> h_bundles_r;
    DevBuf<FimBundle> bundles_r_d;
    DevBuf<FimEnds> ends_r;
    DevBuf<float> Br_pool, slowIr;
    DevBuf<unsigned long long> exc_br;
    DevBuf<int> lists_br, cand_br;
    size_t slowIr_off_b = 0;                   // floats from slowIr to the second group's block
    int bundle_pool_opt = 0;           // option bundle_pool: bundle field slots (0 = up to 1024; fewer than the bundles of a launch: recycled like the unit slots)
    DevBuf<float> slowI, B_pool;       // member-minor slowness of all maps; bundle field slots
    bool slowI_ready = false;
    bool bundles_failed = false;       // a bundle of the current maps ran out of rounds: the automatic mode stays unit by unit until the maps change
    DevBuf<unsigned long long> exc_b;  // exception tables of the bundle slots
    DevBuf<int> lists_b, bpool_gen, member_flag, cand_b;      // (cand_b: tie candidates per bundle slot, kernels.h FimBundle::cand)
    DevBuf<FimBundle> bundles_d;
    std::vector<FimBundle> h_bundles;
    std::vector<int> h_member_flag;
    int bundle_slots = 0;
    size_t solve_stage_bytes() const;
    size_t bundle_room(size_t free_b) const;
    bool grow_unit_pool();
    void release_march_pool();
    bool march_pool_kept = false, released_bundles_for_march = false;      // (exact_ties = 1: the marching pool stays allocated between calls when the device has room, run_exact)
    int bundle_threads() const;
    float bundle_window() const;
    float bundle_window_tail() const;
    size_t bundles_resident(int G, int mpl, int threads = 0) const;
    int bundle_mpl_of(int G, long nb) const;
    int choose_bundle_size(int step, long* solo_units = nullptr);
    int plan_bundles(int first, int n, int G, int* nsolo, int* nbundles);
    size_t lists_stride = 0;
    int fim_threads = 0;               // workgroup size of the solve kernel; 0 = by grid size (launch_shape)
    int fim_lds_pad = 0;               // dynamic LDS bytes per workgroup of the solve kernel (occupancy limiter)
    int fim_sorted = 1;                // 1: k_fim_sorted (tile masks, record-order sweep), 0: k_fim (lists); same fixed point
    // exact mode (exact_kernel.hip): 0 off; 1 = units whose fixed-point solve met an exact time tie (or froze a cycle) are solved
    // again by the literal Fast Marching; 2 = every unit by the literal Fast Marching only
    int exact_ties = kDefaultExactTies;
    float tie_threshold = kDefaultTieThreshold;     // a tie counts when taking the tied neighbour in moves the node's value by more than this (s); 0 = any tie
    int tie_list_opt = 1;              // option tie_list: 1 = the bundles mark tie candidates as they iterate and the census looks at those only; 0 = the census sweeps the converged field (A/B; also the fallback of a list that overflows)
    int tie_detect = 1;                // option tie_detect: exact_ties = 0 runs the detector too and reports the units it would have flagged (no second solve)
    int exact_lds_slots = 0;           // tree slots kept in LDS per marching unit (8 bytes each, made odd); 0 = by the number of units marching (.. 4799)
    int exact_heap_blocked = 1;        // option exact_heap_blocked: the march's tree beyond its LDS part in blocks of three levels (exact_kernel.hip: xg_gi): 0 never, 1 batches that fill the chip, 2 whenever the LDS part is whole levels
    int exact_pool = 0;                // units marching at a time (0 = by free memory, at most exact_pool_max)
    size_t exact_pool_max = 16384;     // option exact_pool_max: four units per wavefront, sixteen wavefronts per CU (measured at 1025^2: 10 240 units 1 500, 12 288 1 600, 16 384 1 700 solves/s)
    DevBuf<unsigned> X_pool;                     // per marching unit: one packed word per node of the whole grid (exact_kernel.hip)
    DevBuf<unsigned long long> X_heap;           // ... and the tree slots beyond the LDS part
    // pooled tiles of a times-only march on a large grid (kernels.h XTiles; run_exact)
    DevBuf<unsigned short> X_tt, X_free;
    DevBuf<unsigned> X_tp, X_ring, X_pins;
    int exact_tiles_opt = 0, exact_tile_cap = 0;
    bool marched_in_tiles = false;
    DevBuf<int> x_units, x_nstart;
    DevBuf<unsigned long long> x_starts;         // the coarse stage's starting tree per marching unit (kernels.h: exact_start_bytes)
    DevBuf<int32_t> xinfo, tieinfo;
    std::vector<unsigned char> h_unit_flags;     // per planned unit after a solve: bit 0 tie met, bit 1 solved by the exact mode
    std::vector<int> h_unit_rounds;               // rounds of the unit's coarse solve (of its bundle's, for a bundled unit)
    std::vector<float> h_unit_tie;               // largest tie influence of the unit (s)
    std::vector<float> h_unit_tie_sum;           // (round 6) sum of the influences of the unit's ties (s) ...
    std::vector<int> h_unit_tie_count;           // ... and how many had one; cycles the unit (its bundle) froze
    std::vector<int> h_unit_froze;
    float tie_sum_threshold = kDefaultTieSumThreshold;      // option tie_sum_threshold: a unit whose ties' influences add up to more than this (s) is flagged; 0 = off
    int tie_count_threshold = kDefaultTieCountThreshold;    // option tie_count_threshold: ... or that holds more ties with an influence than this; 0 = off
    int tie_frozen_bundles = kDefaultTieFrozenBundles;      // option tie_frozen_bundles: 1 = every member of a bundle that froze a cycle is flagged
    int bundle_order_opt = 0;          // option bundle_order: 0 = the first generation's bundles longest first | 1 = bundles of similar length on one CU (plan_bundles)
    int tie_scale_guard = 1;           // option tie_scale_guard: units whose time scale lies outside the measured envelope of the fixed point's tie errors are marched when they hold a tie (Engine::tie_verdicts)
    float tie_tolerance = 1.0e-4f;     // option tie_tolerance: the bar the envelope is held against (s)
    std::vector<float> h_unit_reach_km;                    // per planned unit: great-circle distance to its farthest receiver (plan)
    std::vector<float> map_mean_slow;                      // per map: mean slowness of its vertices (set_maps)
    void mean_slowness_of_maps(const float* hv, size_t nv, int nm);
    int tie_map_strict = kDefaultTieMapStrict;              // option tie_map_strict: on a map where some unit holds a tie above tie_threshold, every unit holding a tie with any influence is flagged
    int tie_verdict(int unit, const int32_t* tie_words, const int32_t* info16, bool member);
    std::vector<char> tie_verdicts(int first, int n, const int32_t* tie_words, const int32_t* info16, bool bundled);
    int run_exact(int first, int n, const std::vector<int>& local_units, bool receivers, bool compact, bool may_pool_tiles = false);
    DevBuf<int8_t> S_r, cinit;
    DevBuf<int16_t> rst, cst;
    DevBuf<int32_t> heap, flags, info;
    DevBuf<FimProblem> prob_r, prob_c;
    DevBuf<unsigned long long> clocks;
    double phase_ticks[kClockSlots] = {};

    // Frechet rows: depth-kernel factor S (ray_kernels.hip) and per-launch ray scratch
    bool have_sens = false;
    int sens_nz = 0, sens_kmax = 0;
    DevBuf<double> Srow, sen_vs, sen_vp, sen_rho;
    DevBuf<float> vels_d;
    std::vector<int> h_trace;          // ids of the rays to trace (flag kRayPath), ascending
    size_t ndata = 0;                  // data (travel times / rows) addressed by the planned rays
    size_t ray_budget = 0;             // bytes for ray slabs per launch (0 = default)
    DevBuf<int> trace_ids, vlist, nvv, counts, coo_col, coo_iw;
    DevBuf<long long> offsets;
    DevBuf<float> slabs, coo_rw;
    DevBuf<int32_t> rayinfo;
    int ray_path_cap = 0;              // > 0: keep up to that many points of every traced ray (dsa_ray_paths)
    DevBuf<float> paths;               // [traced ray][point][colatitude, longitude]
    DevBuf<int> path_n;                // points of every traced ray

    // dispersion stage (disp_kernels.hip): Vs model -> pv maps + depth kernels, all resident
    bool disp_ready = false;
    int disp_nx = 0, disp_ny = 0, disp_nz = 0, disp_kmax_total = 0, disp_nmaps = 0;
    std::vector<float> h_depz;
    LayerGeom h_geom{};
    DevBuf<LayerGeom> geom;
    DevBuf<double> pvstore, curves, tper;
    DevBuf<float> disp_ws;
    DevBuf<unsigned long long> disp_diag;
    // non-fatal diagnostics of the boundary since dispersion_begin / the last plan (dsa_dispersion_diagnostics, dsa_ray_diagnostics)
    long long disp_fail_count = 0;     // dispersion curves that ended with "no zero found" (surfdisp96.f:308-339)
    int disp_fail_first[5] = {};       // first of them in call order: iwave, igr, column (1-based), perturbation (0 = the model itself), period index k
    double disp_fail_period = 0.0;
    // option disp_failure_log: keep up to that many of the curves without a root, in the reference's call order (column, then
    // perturbation, wave type by wave type), so that a host can print the reference's unit-66 block per failing surfdisp96 call
    // (dsa_dispersion_failure replays the curve on the host for the numbers and the layer table)
    int disp_failure_log = 0;
    struct DispFailRec { int iwave, igr, nper, column, pert, k; double t[60]; };
    std::vector<DispFailRec> disp_failures;
    std::vector<float> h_vels;         // the model of dsa_dispersion_begin (ncol * nz): the replay's input
    DevBuf<unsigned long long> disp_fail_list;
    int dispersion_failure(int index, int* info, double* vals, float* table, double* c) const;
    long long rays_clamped = 0;        // traced rays that were clamped at the model boundary (reference rbint, CalSurfG.f90:2082-2101)
    int first_clamped_unit = -1;       // planned unit of the first of them
    int disp_group_shift = -1;         // lanes per Rayleigh curve = 2^shift; -1 = by the number of curves, 0 = one lane per curve
    int disp_layers_lds = -1;          // layer tables of k_dispersion: 1 LDS, 0 global scratch, -1 LDS when they fit

    // optional growing host destination of the COO rows (used when several engines share one call)
    std::vector<float>* grow_rw = nullptr;
    std::vector<int>* grow_iw = nullptr;
    std::vector<int>* grow_col = nullptr;

    // COO rows kept on the device across the chunks of a solve (rows_on_device): what dsa_iteration_system_device and LSMR
    // work on without the matrix ever visiting the host (reference: rw / iw / col of main.f90:349-359, 487-489)
    bool rows_on_device = false;
    DevBuf<float> G_rw;
    DevBuf<int> G_row, G_col;          // 1-based datum (row) and model parameter (column)
    long long G_nar = 0;
    template <class T> int ensure_keep(DevBuf<T>& b, size_t n, size_t used);

    bool spmv_attr_set = false;        // LDS attribute of the blocked SpmV kernels set on this engine's device
    SpmvState* spmv = nullptr;         // device copy of a COO matrix for dsa_spmv (spmv.hip)
    int lsmr_device_vectors = 0;       // dsa_lsmr: 1 = vectors and ordered reductions on the device, 0 = on the host (lsmr.hip)

    double stats[DSA_STAT_COUNT + 2] = {};

    ~Engine();
    void fail(int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
    template <class T> int ensure(DevBuf<T>& b, size_t n);
    int init(int device_index);
    int set_maps(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int dicing, int nm, const double* pv);
    int plan(int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec, const float* rcx, const float* rcz,
             const int* mode, const int* sen_slot, const int* data_first);
    int set_sensitivity(int nz, int kmax, const float* vels, const float* depz, const double* svs, const double* svp, const double* srho, bool on_device);
    int finish_maps(int nm);
    int dispersion_begin(int nx, int ny, int nz, const float* vels, const float* depz, float minthk, int kmax_total, int nmaps_total);
    int dispersion_run(int iwave, int igr, int nper, const double* t, int with_kernels, int sen_slot, int map_first);
    int dispersion_copy_map(int from, int to, int n);
    int dispersion_fetch(int map_first, int nper, double* pv, int with_kernels, int sen_slot, double* svs, double* svp, double* srho);
    int maps_from_dispersion(float goxd, float gozd, float dvxd, float dvzd, int dicing);
    int kernels_from_dispersion();
    int solve(float* dsurf, float* rw, int* iw, int* col, long long cap, long long* nar);
    int trace_chunk(int first_unit, int n, float* rw, int* iw, int* col, long long cap, long long* nar);
    BatchPtrs batch() const;
    FimLaunch launch_shape(int nnx, int nnz) const;
    FimLaunch shape_c{}, shape_r{};    // launch shapes fixed by plan(): lists_stride is sized from them, solve() must use the same
    void launch_srtimes_chunk(int r0, int nr, int first_unit);
    int get_field(int unit, float* ttn);
    int fetch_tiled(const Rec* dev, int nnx, int nnz, int which, float* out);
    int fetch_compact(int slot, int which, float* out);
    int get_refined(int unit, int* rnx, int* rnz, float* ttnr, int8_t* st);
    int get_velocity(int map, float* out_v);
};

}  // namespace dsa
