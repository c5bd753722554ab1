/* dsurftomo_amd -- C ABI of the MI355X forward-modelling engine for DSurfTomo's CalSurfG path.
 *
 * Plain C, pointers and sizes only.  Two levels:
 *
 *   (1) Drop-in level: dsa_calsurfg / dsa_synthetic take exactly the argument lists of the
 *       reference's Fortran subroutines CalSurfG (reference src/CalSurfG.f90:939-943, argument
 *       declarations :987-1002) and synthetic (:2412-2415, :2460-2472): every argument by
 *       pointer, arrays column-major, 1-based indices in iw/col.  The Fortran shim
 *       dsurftomo_amd/fortran/calsurfg_shim.f90 exports the link symbols `calsurfg_` and
 *       `synthetic_` that the reference's main program imports (main.f90:355-359, :338-342) and
 *       forwards to these two functions.  See INTEGRATION.md.
 *
 *   (2) Engine level (own design): an explicit context, phase-velocity maps given per period
 *       slot (this is what the benchmark and the parity tests at synthetic sizes use: "identical
 *       grids" by construction), and batched (period, source) units.
 *
 * All functions return 0 on success or a negative dsa_status; dsa_error_string() gives the text.
 * Where the reference prints a message and STOPs (source or receiver outside the model,
 * CalSurfG.f90:1214-1220, :1686-1692, :1898-1904) the engine returns DSA_ERR_OUTSIDE and the shim
 * prints the reference's message and stops the program, so callers see the same behaviour.
 * There is no CPU fallback: without a usable GPU every entry point fails with DSA_ERR_DEVICE.
 */
#ifndef DSURFTOMO_AMD_H
#define DSURFTOMO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsa_engine dsa_engine;

enum dsa_status {
    DSA_OK = 0,
    DSA_ERR_DEVICE = -1,     /* no GPU / HIP runtime error */
    DSA_ERR_ARGUMENT = -2,
    DSA_ERR_OUTSIDE = -3,    /* a source or receiver lies outside the model */
    DSA_ERR_INTERNAL = -4,   /* a device-side guard fired (window / tree overflow, no convergence) */
    DSA_ERR_STATE = -5,      /* call order (e.g. solve before plan) */
    DSA_ERR_CAPACITY = -6    /* an output array is too small (the reference: stop 'increase sparsity fraction') */
};

/* ---- context ------------------------------------------------------------------------------- */
int dsa_create(dsa_engine** out, int device_index);
void dsa_destroy(dsa_engine* e);
const char* dsa_error_string(const dsa_engine* e);   /* e may be NULL: last creation error */
/* memory the engine may use for per-source fields (bytes, 0 = default: 60 % of free HBM, at most 150 GB) */
int dsa_set_memory_budget(dsa_engine* e, size_t bytes);

/* options: "window_cells" (causal window of the fixed-point solve, in cell travel times, default 1.25),
 * "max_chunk" (cap on sources resident per chunk, 0 = memory budget only), "list_cap" / "ready_cap" (active-list
 * sizes of the list variant of the solve kernel, 0 = derived from the grid), "fim_threads" (workgroup size of the
 * solve kernel: 128, 256, 512 or 1024; 0 default = 128 up to 700 nodes per side, 256 up to 1500, 512 up to 3000, 1024 beyond), "fim_sorted" (1, default = the solve kernel that keeps its active
 * set in tile masks and walks it in record order; 0 = the variant with lists in activation order; same fixed
 * point), "fim_lds_pad" (extra dynamic LDS bytes per workgroup of the solve kernel, limits the workgroups resident
 * per CU), "ray_budget" (bytes of per-ray vertex slabs per launch of the ray tracer, 0 = a third of free HBM up
 * to 40 GB: the tracer runs one lane per ray, so a launch should hold every ray of the call), "ray_path_cap" (points kept per traced ray for dsa_ray_paths, 0 = none), "disp_layers_lds" (layer
 * tables of the dispersion kernel: 1 LDS, 0 global scratch, -1 default = LDS when they fit), "disp_group_shift" (lanes per dispersion
 * curve = 2^shift, 0 = one lane per curve, -1 default = 8 lanes up to 4096 curves, 4 up to 32768), "lsmr_device_vectors"
 * (dsa_lsmr: 0 default = ordered reductions on the host, 1 = all vectors on the device; same results), "field_pool" (field slots of
 * the coarse solve: 0 default = four times the workgroups the GPU holds at once -- a dsa_solve over more units than that recycles the
 * slots, each workgroup claiming a free slot (compare-and-swap, no assumption about dispatch order), resetting it, solving, and writing its unit's receiver times before it frees the slot; -1 = one slot per
 * unit; > 0 = that many.  Fields stay readable (dsa_get_field) only when the units of the call fit the slots; rows / exact mode / keep_fields
 * calls never recycle), "bundle" (the units of one source -- its periods -- solved side by side by one workgroup under one shared round
 * schedule: 1 default = automatic -- 16 / 8 / 4 members per bundle, whichever the measured rates promise most for the call's sources and
 * periods, on grids of at least 120 nodes per side (below 400: only launches of at least 384 bundles of 8 or 16) when the bundles fill the GPU and their field slots fit the memory, else none; 0 = off; 4 / 8 / 16 = that size whatever the count.  Same travel times as unit by unit -- the fixed point does not
 * depend on the schedule; where a field has exact ties (two self-consistent states) the two can settle differently: measured
 * identical on the headline and checkerboard media, 4 of 262 144 receiver times apart by up to 4.2e-5 s on unrelated random maps; exact_ties = 0 and 1 (the tie detector runs inside the bundles), not 2; "bundle_window_cells" = causal window of the bundles, 0 default = 0.6 (1.25 for launches of at most 256 bundles of 8 or 4, which run 768 threads wide); "bundle_threads" = workgroup size of the bundle kernel, 0 default = 256 (three workgroups per CU) -- 768 (one per CU, twelve waves sharing a round) beyond 1500 nodes per side and for launches of at most 256 bundles --, or 256 / 512 / 768; "bundle_members_per_lane" = 0 default (bundles of 16 four members per lane, bundles of 8 / 4 two per lane when the launch holds more than 512 of them), 4 or 2; "bundle_max_rounds" = round limit of the bundles, 0 default = the solver's own (a bundle that hits it sends its chunk to the unit-by-unit solve; used by the tests of that fallback); "bundle_pool" = bundle field slots, 0 default = as many as bundles can be resident at a time and an eighth more (864 at three 256-thread workgroups per CU / 288 for the wide ones; within the memory budget), claimed by the bundles as they start), "exact_ties" / "tie_threshold" / "exact_lds_slots" / "exact_pool" (see dsa_unit_ties).
 * Grid size limit: the coarse solve keeps one bit per 8x8-node tile in LDS (36 KB): up to about 4340 nodes per side (nx <= 545 at dicing 8);
 * dsa_plan returns DSA_ERR_ARGUMENT beyond. */
int dsa_set_option(dsa_engine* e, const char* name, double value);

/* ---- engine level --------------------------------------------------------------------------- */
/* Grid of reference CalSurfG.f90:1032-1065 (dicing 8) / :2487-2520 (dicing 5) and `nmaps`
 * velocity maps pv[m][nx*ny] (fp64, latitude index fastest: pv[(jj-1)*nx + ii - 1], the layout
 * of the reference's pvRc etc.).  Runs the dicing kernel once per map (the reference repeats it
 * per source, :1186). */
int dsa_set_maps(dsa_engine* e, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd,
                 int dicing, int nmaps, const double* pv);

/* Describe the (map, source) units and their receivers (colatitude / longitude in radians, the
 * convention of scxf/sczf/rcxf/rczf).  Receivers of unit u are rcx[first .. first+nrec[u]) with
 * first = sum of nrec over earlier units.  Host-side preparation and upload only. */
int dsa_plan(dsa_engine* e, int nunits, const int* map_index, const float* scx, const float* scz,
             const int* nrec, const float* rcx, const float* rcz);

/* Same with per-unit extras (any may be NULL): mode[u] bit 0 = produce receiver times, bit 1 = trace
 * rays / emit Frechet rows (default both); sen_slot[u] = period slot of the depth kernels used
 * by unit u's rows; data_first[u] = 0-based index of the datum of unit u's first receiver
 * (default: running receiver count).  Group-velocity data take two units with the same
 * data_first: times on the group-velocity map, rays on the phase-velocity map
 * (reference CalSurfG.f90:1166-1183, :1360-1376). */
int dsa_plan_units(dsa_engine* e, int nunits, const int* map_index, const float* scx, const float* scz,
                   const int* nrec, const float* rcx, const float* rcz, const int* mode,
                   const int* sen_slot, const int* data_first);

/* Depth kernels for the Frechet rows, in the reference's layout: vels(nx,ny,nz) fp32,
 * depz(nz), sen_*(nx*ny, kmax, nz) fp64 (CalSurfG.f90:1005-1016).  Uploaded and folded with the
 * Brocher chain-rule factors (:1385-1423) once. */
int dsa_set_depth_kernels(dsa_engine* e, int nz, int kmax, const float* vels, const float* depz,
                          const double* sen_vs, const double* sen_vp, const double* sen_rho);

/* Dispersion stage on the device (reference depthkernel CalSurfG.f90:1-169 / caldespersion
 * :2866-2927 over surfdisp96.f): spherical earth, fundamental mode.
 *   begin: the Vs model vels(nx,ny,nz), depths depz(nz), sublayering minthk; room for
 *          `nmaps_total` phase/group-velocity maps and `kmax_total` depth-kernel slots.
 *   run:   one wave type (iwave 1 Love / 2 Rayleigh, igr 0 phase / 1 group) at periods t[nper]:
 *          maps [map_first, map_first+nper) and, if with_kernels, slots [sen_slot, sen_slot+nper).
 *   copy_maps: duplicate maps inside the store (the reference overwrites the head of pvRc / pvLc
 *          with the phase velocities at the group periods, CalSurfG.f90:1110, :1128).
 *   fetch: host copies in the reference's layouts pv(nx*ny, nper), sen_*(nx*ny, nper, nz).
 *   maps_from_dispersion / kernels_from_dispersion: hand the resident results to the solve
 *          (same effect as dsa_set_maps / dsa_set_depth_kernels, no host round trip). */
int dsa_dispersion_begin(dsa_engine* e, int nx, int ny, int nz, const float* vels, const float* depz,
                         float minthk, int kmax_total, int nmaps_total);
int dsa_dispersion_run(dsa_engine* e, int iwave, int igr, int nper, const double* t, int with_kernels,
                       int sen_slot, int map_first);
int dsa_dispersion_copy_maps(dsa_engine* e, int from, int to, int n);
int dsa_dispersion_fetch(dsa_engine* e, int map_first, int nper, double* pv, int with_kernels,
                         int sen_slot, double* sen_vs, double* sen_vp, double* sen_rho);
int dsa_maps_from_dispersion(dsa_engine* e, float goxd, float gozd, float dvxd, float dvzd, int dicing);
int dsa_kernels_from_dispersion(dsa_engine* e);

/* Solve every planned unit: eikonal field per unit, then receiver times into dsurf (host,
 * one float per datum, unit-major order == the reference's (knumi, srcnum, istep) order). */
int dsa_solve(dsa_engine* e, float* dsurf);

/* dsa_solve with the receiver times left on the device: d_dsurf is a DEVICE pointer (memory of the engine's GPU, one float per
 * datum).  For the multi-GPU path: the rank's slice goes into the RCCL all-gather straight from HBM (bench.py, sharding.py). */
int dsa_solve_device(dsa_engine* e, void* d_dsurf);

/* dsa_solve plus rays and Frechet rows (reference rpaths + row loop, CalSurfG.f90:1377-1432):
 * COO triplets in the reference's order -- rw[k] value, iw[k] 1-based row (datum), col[k]
 * 1-based column (k-1)*nvx*nvz + (jj-1)*nvx + kk -- *nar entries, at most `capacity`. */
int dsa_solve_rows(dsa_engine* e, float* dsurf, float* rw, int* iw, int* col, long long capacity,
                   long long* nar);

/* Ray paths (the reference's disabled raypath.out dump, CalSurfG.f90:2276-2283, read by its
 * scripts/plotpath.py): with dsa_set_option(e, "ray_path_cap", C) the next dsa_solve_rows keeps up to C points per traced
 * ray.  For the R traced rays (DSA_STAT_RAYS) in data order: datum[R] 1-based row, npts[R] points of the ray (may exceed
 * C), latlon[R * C * 2] (latitude, longitude) in degrees: receiver first, source last. */
int dsa_ray_paths(dsa_engine* e, int* datum, int* npts, float* latlon);

/* ---- next to the path: the matrix-vector products of the inversion step (reference aprod.f90:7-60) ----
 * load: COO matrix (rw[k], 1-based row[k] <= m, col[k] <= n), kept on the device in row-major and
 * column-major order; spmv mode 1: y += A x, mode 2: x += A^T y on host vectors x[n], y[m].
 * Every output element adds its entries in storage order in fp32, like the reference's loop. */
int dsa_spmv_load(dsa_engine* e, int m, int n, long long nar, const float* rw, const int* row, const int* col);
int dsa_spmv(dsa_engine* e, int mode, float* x, float* y);

/* LSMR of the inversion step (reference lsmrModule.f90:36-750, single precision as shipped, called at main.f90:487) on
 * the matrix of the last dsa_spmv_load, all vectors resident on the device.  b[m] right-hand side (host, not
 * modified), x[n] solution (host, out); the other arguments and results are the reference's (damp, atol, btol,
 * conlim, itnlim, localSize -> istop, itn, normA, condA, normr, normAr, normx).  Sums run in the reference's order:
 * results are bit-identical to the reference's LSMR. */
int dsa_lsmr(dsa_engine* e, const float* b, float damp, float atol, float btol, float conlim, int itnlim,
             int localSize, float* x, int* istop, int* itn, float* normA, float* condA, float* normr,
             float* normAr, float* normx);

/* One outer iteration's host glue (reference main.f90:361-466 and :520-535; plain host code, no device):
 * iteration_system: residual cbst = obst - dsyn, percentile weights (getpercentile.f90), rows scaled by their weights,
 *   DWS norm[maxvp] with dws = {max, mean}, regularisation rows appended.  In/out rw, col (capacity entries) and iw
 *   (2*capacity + 1: iw[0] = final nar, then rows, then columns -- the layout aprod / LSMR take); cbst has dall + maxvp
 *   values, *m_out = dall + maxvp rows.
 * model_update: dv clipped to +-0.5, vsf(nx, ny, nz) += dv on the interior, clipped to [minvel, maxvel]. */
int dsa_iteration_system(int nx, int ny, int nz, int dall, long long nar_in, long long capacity, float* rw, int* iw,
                         int* col, const float* obst, const float* dsyn, float threshold0, float weight0, float* cbst,
                         float* datweight, float* norm, int* m_out, long long* nar_out, float* dws);
int dsa_model_update(int nx, int ny, int nz, float* dv, float* vsf, float minvel, float maxvel);
/* The same system built where the rows are: on the COO rows that dsa_solve_rows (option rows_on_device) or dsa_calsurfg
 * (called with null rw / iw / col) left on the device.  Weights, regularisation rows, DWS and both orderings of the matrix
 * are made on the device; the matrix (12 bytes per entry) never crosses PCIe (reference: main.f90:349-359 -> :361-466 ->
 * :487-489 all on host arrays).  cbst has dall + maxvp elements; afterwards dsa_lsmr(e, cbst, ...) solves on that matrix.
 * Bit-identical to dsa_iteration_system + dsa_spmv_load. */
int dsa_iteration_system_device(dsa_engine* e, int nx, int ny, int nz, int dall, const float* obst, const float* dsyn,
                                float threshold0, float weight0, float* cbst, float* datweight, float* norm, int* m_out,
                                long long* nar_out, float* dws);

/* copy one unit's coarse travel-time field (nnz, nnx column-major) back; valid after dsa_solve
 * for units of the last chunk only unless keep_fields was requested */
int dsa_get_dims(const dsa_engine* e, int* nnx, int* nnz);
int dsa_keep_fields(dsa_engine* e, int on);
int dsa_get_field(dsa_engine* e, int unit, float* ttn);
int dsa_get_velocity(dsa_engine* e, int map, float* veln);
/* refined snapshot of a unit: ttnr (rnz, rnx) and status (-1 far, 0 alive, 1 close) */
int dsa_get_refined(dsa_engine* e, int unit, int* rnx, int* rnz, float* ttnr, int8_t* status);

/* raw state of a resident unit for diagnostics: which = 0 coarse T (sign bit = pinned), 1 coarse
 * tau (sign bit = queued), 2 refined T, 3 refined tau (rnx*rnz floats, leading dimension rnz) */
int dsa_debug_field(dsa_engine* e, int unit, int which, float* out);

/* Non-fatal diagnostics of the boundary (SURVEY.md 8b "Error convention").
 * dispersion: curves of the dispersion runs since dsa_dispersion_begin that ended with the reference's "improper initial value in
 *   disper - no zero found" (surfdisp96.f:308-339; the rest of such a curve is zero, :342-348): their number, the first one in call
 *   order as first[5] = { iwave (1 Love, 2 Rayleigh), igr, column (1-based, (jj-1)*nx+ii), perturbation (0 = the model itself, else
 *   1 + 6 depth + 2 parameter + sign of the depth-kernel differences), period index k }, and that period.
 * rays: traced rays of the last dsa_solve_rows that were clamped at the model boundary (reference rbint, CalSurfG.f90:2082-2101,
 *   reported by the note of :1447-1454), and the planned unit of the first of them (-1: none). */
int dsa_dispersion_diagnostics(const dsa_engine* e, long long* count, int* first, double* period);
/* The failing curves one by one (reference surfdisp96.f:308-339 writes its block, with the call's layer table, once per failing
 * surfdisp96 call).  Option "disp_failure_log" = N > 0 keeps the first N of them in the reference's single-thread call order (wave type
 * by wave type; column by column, the model itself, then its depth-kernel perturbations: CalSurfG.f90:44-150); dsa_dispersion_failure
 * replays curve `index` (0-based) on the host and returns what the block prints: info[8] = { ifunc (1 L, 2 R), igr, column
 * (jj-1)*nx+ii, perturbation, k, ie (periods of the call; is = 1), mmax, failures logged }, vals[4] = { t(k), cc, cm, c1 },
 * table[800] = d, a, b, rho of the flattened layers (200 each, mmax used), c[60] = the roots of the periods before k (the reference
 * also prints c(k), an element it never assigned: 0 here).  DSA_ERR_ARGUMENT beyond the logged ones. */
int dsa_dispersion_failure(const dsa_engine* e, int index, int* info, double* vals, float* table, double* c);
int dsa_ray_diagnostics(const dsa_engine* e, long long* clamped, int* first_unit);

/* Exact time ties: the fixed-point solve lands on the reference's Fast-Marching travel times except downstream of
 * bit-equal times of two neighbouring narrow-band nodes, where the reference's own answer depends on the layout of its binary
 * tree (CalSurfG.f90:417-485, :768-921).  Option "exact_ties": 0 (default) fixed point only; 1 = the solve kernel detects such
 * ties (option "tie_threshold", seconds: the influence on the node's value a tie must have to count; default 2e-5, 0 = any tie) and the
 * units that met one are solved again by the reference's march itself, replayed on the device four units per wavefront -- their
 * fields are then bit-identical to the reference's; 2 = every unit by the literal march (2 100 solves/s at 1025^2 with 16 000 units in
 * flight).  Options "exact_lds_slots" (tree slots in LDS per marching unit, the rest of the tree in global memory: 64 ..
 * 4991, made odd; default 0 = what lets every wavefront of a batch be resident: 148 KB of a CU's LDS divided among them), "exact_pool"
 * (units marching at a time, 0 = by free memory: 80 % of it at 4 bytes per node and unit -- 50 solves/s at 4097^2 --, at most "exact_pool_max", default 16384).  The march's tree
 * holds at most 65 534 nodes per unit (16-bit slots): a narrow band longer than that returns DSA_ERR_INTERNAL (grids beyond ~8000
 * nodes per side; the grid size limit below is lower).
 * dsa_unit_ties: per planned unit of the last solve, flags (bit 0: met a tie, bit 1: solved by the literal march) and the largest
 * tie influence in seconds (either array may be NULL). */
int dsa_unit_ties(const dsa_engine* e, int nunits, int* flags, float* influence);
/* rounds the coarse fixed-point solve of each planned unit took in the last dsa_solve (a bundled unit: its bundle's) */
int dsa_unit_rounds(const dsa_engine* e, int nunits, int* rounds);

/* probe builds only (-DDSA_LEDGER, tools/isa_ledger.py): trip counters of the coarse solve's phases, summed over the units of the last solve */
int dsa_debug_counters(const dsa_engine* e, double* out24);

/* counters of the last dsa_solve: see DSA_STAT_* */
enum { DSA_STAT_MS_TOTAL = 0, DSA_STAT_MS_FIM_COARSE, DSA_STAT_MS_FIM_REFINED, DSA_STAT_MS_STAGES,
       DSA_STAT_LAUNCHES_FIM_COARSE, DSA_STAT_UNITS, DSA_STAT_ROUNDS_MAX, DSA_STAT_EVALS_TOTAL,
       DSA_STAT_CHUNK, DSA_STAT_RESCANS, DSA_STAT_FREEZES, DSA_STAT_RAYS, DSA_STAT_RAY_STEPS,
       DSA_STAT_RAYS_CLAMPED, DSA_STAT_MS_RAYS, DSA_STAT_MS_ROWS, DSA_STAT_NAR, DSA_STAT_MS_DISPERSION,
       DSA_STAT_CURVES, DSA_STAT_CHANGES_TOTAL, DSA_STAT_TIE_UNITS, DSA_STAT_EXACT_UNITS, DSA_STAT_EXACT_POPS,
       DSA_STAT_MS_EXACT, DSA_STAT_FIELD_SLOTS, DSA_STAT_FOOTPRINT_MB, DSA_STAT_BUNDLE_SIZE, DSA_STAT_BUNDLES,
       DSA_STAT_BUNDLED_UNITS, DSA_STAT_BUNDLE_SLOTS, DSA_STAT_BUNDLE_THREADS,
       DSA_STAT_TIE_UNITS_LEFT,       /* units the tie detector flagged (DSA_STAT_TIE_UNITS, filled in every mode) that stayed with the fixed point: exact_ties = 0 */
       DSA_STAT_TIE_INFLUENCE_MAX,    /* largest tie influence met (seconds) */
       DSA_STAT_COUNT };
int dsa_get_stats(const dsa_engine* e, double* out /* DSA_STAT_COUNT + 8: counters, then 8 phase-clock sums (probe builds) */);

/* ---- drop-in level -------------------------------------------------------------------------- */
int dsa_calsurfg(const int* nx, const int* ny, const int* nz, const int* nparpi, const float* vels,
                 int* iw, float* rw, int* col, float* dsurf,
                 const float* goxdf, const float* gozdf, const float* dvxdf, const float* dvzdf,
                 const int* kmaxRc, const int* kmaxRg, const int* kmaxLc, const int* kmaxLg,
                 const double* tRc, const double* tRg, const double* tLc, const double* tLg,
                 const int* wavetype, const int* igrt, const int* periods, const float* depz,
                 const float* minthk, const float* scxf, const float* sczf, const float* rcxf,
                 const float* rczf, const int* nrc1, const int* nsrcsurf1, const int* kmax,
                 const int* nsrcsurf, const int* nrcf, int* nar);

int dsa_synthetic(const int* nx, const int* ny, const int* nz, const int* nparpi, const float* vels,
                  float* obst,
                  const float* goxdf, const float* gozdf, const float* dvxdf, const float* dvzdf,
                  const int* kmaxRc, const int* kmaxRg, const int* kmaxLc, const int* kmaxLg,
                  const double* tRc, const double* tRg, const double* tLc, const double* tLg,
                  const int* wavetype, const int* igrt, const int* periods, const float* depz,
                  const float* minthk, const float* scxf, const float* sczf, const float* rcxf,
                  const float* rczf, const int* nrc1, const int* nsrcsurf1, const int* kmax,
                  const int* nsrcsurf, const int* nrcf, const float* noiselevel);

/* Capacity (entries) of the rw / iw(2:) / col arrays handed to dsa_calsurfg from now on.  The reference's interface
 * (CalSurfG.f90:939-943) does not carry it -- main.f90:287 sizes the arrays as spfra*dall*nx*ny*nz and only checks
 * afterwards (main.f90:467); with it dsa_calsurfg returns DSA_ERR_CAPACITY instead of writing past the arrays.
 * 0 = not stated (then DSA_MAXNAR from the environment, else unlimited). */
int dsa_dropin_set_capacity(long long maxnar);

/* the reference's aprod (aprod.f90:7-60: mode 1 y += A x, mode 2 x += A^T y; iw = [nar, rows, cols]) on the
 * device; the matrix is uploaded when first seen (dsurftomo_amd/fortran/aprod_shim.f90 exports `aprod_`) */
int dsa_aprod(const int* mode, const int* m, const int* n, float* x, float* y, const int* leniw,
              const int* lenrw, const int* iw, const float* rw);
/* Contract of the device copy behind dsa_aprod: the matrix is uploaded when its address, size or a sample of its
 * entries changes, and whenever a new LSMR solve starts (two mode-2 products in a row: lsmrModule.f90:390 opens a
 * solve with mode 2 and :497 ends every iteration with it) -- which covers the reference's main program, that rebuilds
 * rw / iw in place before each solve (main.f90:361-466).  Any other in-place edit must be announced with this call. */
int dsa_aprod_invalidate(void);

/* the reference's LSMR with its own argument list (lsmrModule.f90:36-39: every argument by reference; iw = [nar,
 * rows, cols], nout ignored) on the device; dsurftomo_amd/fortran/lsmr_shim.f90 exports module lsmrModule with it */
int dsa_lsmr_dropin(const int* m, const int* n, const int* leniw, const int* lenrw, const int* iw, const float* rw,
                    const float* b, const float* damp, const float* atol, const float* btol, const float* conlim,
                    const int* itnlim, const int* localSize, const int* nout, float* x, int* istop, int* itn,
                    float* normA, float* condA, float* normr, float* normAr, float* normx);

/* pv(nx*ny, kmaxXX) of the last drop-in call: which = 0 Rc, 1 Rg, 2 Lc, 3 Lg (what the reference's synthetic
 * writes to velmap2d*.dat, CalSurfG.f90:2559-2617) */
int dsa_dropin_velocity_maps(const int* which, double* pv);

/* Non-fatal diagnostics of the last dsa_calsurfg call, for the caller to print where the reference prints them:
 *   rbint_notes: how many times the reference would have written its six-line boundary note to unit 6 -- once after every
 *     (period, source) iteration from the first one with a clamped ray on (rbint is set once and tested inside the source loop,
 *     CalSurfG.f90:1088, :1447-1454); 0 = no ray touched the boundary;
 *   disp_count / disp_first[5] / disp_period: see dsa_dispersion_diagnostics (the reference writes its block to unit 66).
 * The C level prints nothing itself; dsurftomo_amd/fortran/calsurfg_shim.f90 writes the reference's texts. */
int dsa_dropin_diagnostics(int* rbint_notes, long long* disp_count, int* disp_first, double* disp_period);
/* Tie census of the last dsa_calsurfg / dsa_synthetic call (see dsa_unit_ties): units holding an exact time tie whose influence exceeds the
 * threshold; how many of them stayed with the fixed point (exact_ties = 0 -- their times may differ from the reference's Fast Marching by more
 * than 1e-4 s; the shim writes one line about it to unit 6) and how many were solved again by the reference's march (exact_ties = 1, the default);
 * the largest influence met, in seconds.  Any pointer may be NULL. */
int dsa_dropin_tie_diagnostics(long long* flagged_units, long long* left_to_fixed_point, long long* marched_units, float* largest_influence);
/* dsa_dispersion_failure of the last dsa_calsurfg call (environment DSA_DISP_FAILURE_LOG = N switches the log on): the shim then writes
 * the reference's unit-66 block once per failing surfdisp96 call, layer table included */
int dsa_dropin_dispersion_failure(int index, int* info, double* vals, float* table, double* c);

/* text of the last error of the process-wide engine used by the drop-in level */
const char* dsa_dropin_error(void);
/* that engine (created on first use; null if no GPU), for engine-level calls that continue a drop-in call on the device */
dsa_engine* dsa_dropin_engine(void);

#ifdef __cplusplus
}
#endif
#endif
