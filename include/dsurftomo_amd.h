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

/* Options (dsa_set_option; every value is a double).  Defaults in [brackets]; DSA_ERR_ARGUMENT for an unknown name or a value out of range.
 *
 *   tie handling (see dsa_unit_ties below)
 *     exact_ties               [1]     0 fixed point only | 1 fixed point + tie census + the reference's march for the flagged units | 2 the march for every unit
 *     tie_threshold            [2e-5]  seconds: the influence on its node's value a tie must have to flag its unit; 0 = any tie
 *     tie_detect               [1]     exact_ties = 0 runs the census too and reports what it would have flagged (DSA_STAT_TIE_UNITS, dsa_unit_ties); 0 = off
 *     tie_map_strict           [1]     on a map where some unit holds a tie above tie_threshold, every unit holding a tie with any influence is flagged; 0 = the per-unit rule alone
 *     tie_scale_guard          [1]     a unit that holds a tie and whose travel times lie outside the envelope in which the fixed point's tie errors were measured to stay
 *                                      within tie_tolerance (26 ulps of the time on grids up to 1025 nodes per side, growing with the grid beyond) is flagged; 0 = off
 *     tie_tolerance            [1e-4]  seconds: the bar that envelope is held to (a lower value marches more units)
 *     handoff_replay           [1]     a unit in which a node ranks equal with the one that ended the refined stage, and the choice changes what the coarse grid receives, has its
 *                                      refined box (<= 129^2 nodes) marched literally and is handed off from that (up to 256 units a launch); 0 = such a unit is flagged (marched whole)
 *     tie_sum_threshold        [0]     seconds: a unit whose ties' influences add up to more than this is flagged; 0 = off (measured: separates nothing, see dsa_unit_tie_sums)
 *     tie_count_threshold      [0]     a unit holding more ties with an influence than this is flagged; 0 = off
 *     tie_frozen_bundles       [0]     1 = every member of a bundle that froze a cycle is flagged (a unit-by-unit solve that froze one always is)
 *     exact_lds_slots          [0]     tree slots kept in LDS per marching unit, 63 .. 4975 (made odd); 0 = what lets every wavefront of a batch be resident
 *     exact_heap_blocked       [1]     the march's tree beyond its LDS part stored in blocks of three levels: 0 never | 1 batches that fill the chip (>= 12 wavefronts per CU) | 2 whenever the LDS part is whole levels
 *     exact_pool               [0]     units marching at a time, up to 65535; 0 = by free memory, at most exact_pool_max
 *     exact_pool_max           [16384] 4 .. 32768
 *     exact_tiles              [0]     times-only calls: 0 = the march keeps pooled 8x8-node tiles per unit instead of whole fields when whole fields would bound the
 *                                      units marching side by side (4097^2: 3 MB per unit instead of 67) | 1 always | -1 never
 *     exact_tile_cap           [0]     tiles per marching unit, 64 .. 65000; 0 = 8 (tiles per grid side, both sides added)
 *
 *   fixed-point solve, unit by unit (csrc/fim_kernel.hip)
 *     window_cells             [1.25]  causal window in cell travel times
 *     fim_threads              [0]     workgroup size 128 | 256 | 512 | 1024; 0 = by grid size (128 up to 700 nodes per side, 256 up to 1500, 512 up to 3000, 1024 beyond)
 *     fim_sorted               [1]     refined boxes: 1 tile masks walked in record order | 0 lists in activation order (same fixed point)
 *     fim_lds_pad              [0]     extra dynamic LDS bytes per workgroup (limits the workgroups resident per CU; experiments)
 *     list_cap, ready_cap      [0]     active-list sizes of the list variant; 0 = from the grid
 *     field_pool               [0]     coarse field slots: 0 = four times the workgroups the GPU holds (a call with more units recycles them: a workgroup
 *                                      claims a free slot by compare-and-swap, resets it, solves, writes its unit's receiver times, frees it) | -1 one per
 *                                      unit | n > 0.  Fields stay readable (dsa_get_field) only when the call's units fit the slots; calls that need the
 *                                      fields afterwards (rows, exact_ties = 2, keep_fields) never recycle
 *     exc_log2cap              [0]     log2 of the exception table's entries, 6 .. 24; 0 = from the grid (the table grows by itself when it overflows)
 *     max_chunk                [0]     cap on units resident per launch; 0 = memory budget only
 *
 *   bundles: the periods of one source solved by one workgroup under one shared round schedule (csrc/bundle_kernel.hip).  Same travel times as unit by
 *   unit -- the fixed point does not depend on the schedule; a field with exact ties has two self-consistent states there and the two solves can settle
 *   differently (measured: identical on the headline and checkerboard media, 4 of 262 144 times apart by up to 4.2e-5 s on unrelated random maps)
 *     bundle                   [1]     1 automatic: 16 / 8 / 4 members, whichever the launch-time model promises most for the call's sources and periods,
 *                                      on grids of at least 120 nodes per side (below 400: only launches of at least 384 bundles of 8 or 16) | 0 off | 4, 8, 16
 *     bundle_window_cells      [0]     causal window of the bundles; 0 = 0.6 (768 threads wide: 1.0 for bundles of 16, 1.75 for bundles of 8 or 4)
 *     bundle_threads           [0]     256 | 512 | 768; 0 = 256 (three workgroups per CU), 768 beyond 1500 nodes per side and for launches of at most 256 bundles
 *     bundle_members_per_lane  [0]     4 | 2; 0 = four for bundles of 16, two for bundles of 8 / 4 in launches of more than 512
 *     bundle_pool              [0]     bundle field slots; 0 = the bundles resident at a time and an eighth more, within the memory budget
 *     bundle_tail              [1]     a launch of 768 .. 1500 bundles, the ones beyond the first generation (768 = three workgroups per CU): 1 whole and 768 threads
 *                                      wide on a second stream, a CU each, when there are at most 256 of them | 0 cut in halves (256 threads; also beyond 256)
 *     bundle_refined           [1]     the 129^2 refined boxes of bundled units in bundles too, in launches of at least 128 bundles | 0 unit by unit | 2 always
 *     bundle_max_rounds        [0]     round limit of a bundle; 0 = the solver's own.  A bundle that hits it sends its chunk to the unit-by-unit solve (tests)
 *     bundle_far_all           [0]     1 = every node trip fetches all four outer neighbours (round 4's loads; A/B switch)
 *
 *   rays, rows, dispersion, inversion step
 *     ray_budget               [0]     bytes of per-ray vertex slabs per launch of the tracer; 0 = a third of free HBM, up to 40 GB
 *     ray_lanes                [0]     lanes that trace a ray together: 0 = by the size of the launch (4 up to 81 920 rays, else 1) | 1 | 4 (same rows)
 *     ray_path_cap             [0]     points kept per traced ray for dsa_ray_paths
 *     rows_on_device           [0]     1 = dsa_solve_rows leaves the COO rows on the device (dsa_iteration_system_device, dsa_lsmr)
 *     disp_layers_lds          [-1]    layer tables of the dispersion kernel: 1 LDS | 0 global scratch | -1 LDS when they fit
 *     disp_group_shift         [-1]    lanes per dispersion curve = 2^shift, 0 .. 3; -1 = 8 lanes up to 4096 curves, 4 up to 32768, else 1
 *     disp_failure_log         [0]     keep the first N curves without a root in the reference's call order (dsa_dispersion_failure)
 *     lsmr_device_vectors      [0]     dsa_lsmr: 0 ordered reductions on the host | 1 all vectors on the device (same results)
 *
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

/* Exact time ties (DESIGN.md "Ties").  The fixed-point solve lands on the reference's Fast-Marching travel times except downstream of bit-equal
 * times of two neighbouring narrow-band nodes, where the reference's own answer depends on the layout of its binary tree (CalSurfG.f90:417-485,
 * :768-921).  Option "exact_ties":
 *   1 (default)  fixed point, then a census of the converged fields.  A node holds a tie when a near neighbour's ACCEPTANCE time equals its value bit for
 *                bit (for all but ~0.02 % of the nodes that is the neighbour's value), or -- round 6 -- when an exceptional outer node was accepted at
 *                the very clock of the neighbour taken in last, or when a node of the refined box ranks equal with the node that ended the refined
 *                stage (the hand-off's probe); the tie's influence is what deciding it the other way changes at the node.  Flagged and solved again by
 *                the reference's march itself (four units per wavefront; field, refined snapshot and receiver times then the reference's bit for bit):
 *                  - a unit holding a tie whose influence exceeds "tie_threshold";
 *                  - "tie_map_strict" (on): on a map where some unit holds such a tie -- a tie-prone medium: sharp contrasts, second-order stencils
 *                    switching along ridges, where a one-ulp difference grows downstream -- every unit holding a tie with ANY influence;
 *                  - (a tie at the hand-off that changes what the coarse grid receives -- a difference of first order, not an ulp -- is not flagged but
 *                    resolved: the refined box is marched literally and handed off again, "handoff_replay"; only a full list of such units, 256 a launch,
 *                    flags the rest; where the refined box's slowness does not vary along x -- a 1-D model: the two choices are mirror images -- it counts);
 *                  - a unit whose band march could not leave its tree a heap, or whose bundle froze a cycle ("tie_frozen_bundles");
 *                  - "tie_scale_guard" (on): a unit holding a tie whose times lie outside the MEASURED ENVELOPE.  Downstream of one-ulp ties the fixed
 *                    point's field differs from the reference's by a number of ulps of the travel time that grows with the grid -- at a receiver at
 *                    most 26 ulps on grids up to 1025 nodes per side in 2.2 M fuzzed units (9.92e-5 s), 36 and 27 ulps in two units of the next 0.7 M (1.37e-4 s, 1.03e-4 s: at 1025^2 with
 *                    times of 32-64 s about one unit in 400 000 ends beyond the bar -- the tolerance is statistical there; lower "tie_tolerance" for a margin), 35 at 2049^2, 110 at 4097^2 -- so against the absolute bar
 *                    ("tie_tolerance", 1e-4 s) it is the size of the times that decides: a unit whose farthest receiver (great-circle distance x the
 *                    map's mean slowness) lies at 64 s or beyond on grids up to 1025^2 -- 32 s at 2049^2, 16 s at 4097^2 -- is marched.
 *                What stays with the fixed point: units without a tie (measured: bit-identical to the reference but for 5 of 83 000 such units, off by an
 *                ulp at a receiver: DESIGN.md "Ties", known residuals), and -- on maps where no tie reaches the threshold -- units holding ties of an
 *                ulp or two: within 1e-4 s of the reference BY MEASUREMENT (about 450 000 smooth-medium units x 32 receivers: worst 5.7e-6 s at 121^2-193^2,
 *                9.5e-5 s at 1025^2 with times of 32-64 s: the envelope the scale guard above holds the call to), not by
 *                construction (DSA_STAT_TIE_UNITS_TIED counts them, the
 *                shim says so once per call).  No rule on a unit's own ties -- largest, summed, counted influence -- separates the rare unit that ends
 *                beyond 1e-4 s from the thousands that do not (profiles/r06_tie_rule_scan_*.log).
 *                Cost: a few per cent where nothing is flagged; a flagged unit costs one march (sequential accepts, ~2.3 us each: 35 ms at 121^2,
 *                2.4 s at 1025^2, a minute at 4097^2 -- the same for one unit or thousands side by side): a tie-prone medium runs at the march's rate;
 *   2            every unit by the march (the guarantee; ~2 000 solves/s at 1025^2 with 16 000 units in flight, ~50 at 4097^2);
 *   0            the fixed point alone; the census still runs ("tie_detect") and DSA_STAT_TIE_UNITS / DSA_STAT_TIE_UNITS_LEFT / dsa_unit_ties say
 *                which units a default call would have marched.
 * The march's tree holds at most 65 534 nodes per unit (16-bit slots): a narrow band longer than that returns DSA_ERR_INTERNAL (grids beyond ~8000
 * nodes per side; the grid size limit above is lower).
 * dsa_unit_ties: per planned unit of the last solve, flags (bit 0: holds a tie above the threshold, bit 1: solved by the march) and the largest
 * tie influence in seconds (either array may be NULL). */
int dsa_unit_ties(const dsa_engine* e, int nunits, int* flags, float* influence);
/* (round 6) what else the census keeps per planned unit of the last solve: how many of the unit's ties have an influence at all, the sum of those
 * influences in seconds (sub-threshold ties add up along a front: options tie_sum_threshold / tie_count_threshold flag by them), and the cycles
 * the unit -- or, for a bundled unit, its bundle -- froze (option tie_frozen_bundles).  Any array may be NULL. */
int dsa_unit_tie_sums(const dsa_engine* e, int nunits, int* count, float* sum, int* frozen);
/* rounds the coarse fixed-point solve of each planned unit took in the last dsa_solve (a bundled unit: its bundle's) */
int dsa_unit_rounds(const dsa_engine* e, int nunits, int* rounds);

/* Device self-check, on no product path (tests/test_gpu_boundary.py): the hand-expanded divisions of the dispersion and ray kernels
 * (dispersion_core.h: recip_of / div_by; ray_core.h: recipf_of / divf_by) against the compiler's IEEE division on `millions` x 1e6 random operand
 * pairs per precision, five numerators per denominator, zeros / infinities / NaN among the numerators.  exponents8: unbiased binary exponent
 * ranges {numerator lo, hi, denominator lo, hi} for fp64, then for fp32.  out4: fp64 pairs, fp64 quotients that differ bitwise, fp32 pairs, fp32
 * quotients that differ.  Needs no engine; uses the current device. */
int dsa_selfcheck_divisions(unsigned long long seed, int millions, const int* exponents8, unsigned long long* out4);

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
       DSA_STAT_EXACT_POOL,           /* units the last march held side by side */
       DSA_STAT_EXACT_TILES,          /* > 0: it marched in pooled tiles, that many 8x8-node tiles per unit */
       DSA_STAT_TIE_UNITS_STRICT,     /* (round 6) of DSA_STAT_TIE_UNITS: flagged only because their map is tie-prone (option tie_map_strict) */
       DSA_STAT_TIE_PRONE_MAPS,       /* maps on which some unit held a tie above tie_threshold, summed over the call's launches */
       DSA_STAT_TIE_UNITS_TIED,       /* units that stayed with the fixed point although the census found a tie with an influence in them: their times are the
                                         reference's to ~1e-4 s statistically, not by construction (DESIGN.md "Ties") */
       DSA_STAT_TIE_UNITS_BY_SCALE,   /* of DSA_STAT_TIE_UNITS: flagged because they hold a tie and their travel times lie outside the envelope in which the
                                         fixed point's tie errors were measured to stay within the tolerance (option tie_scale_guard) */
       DSA_STAT_HANDOFFS_REPLAYED,    /* units whose refined box was marched literally behind the hand-off's probe (a node ranking equal with the one that ended the refined
                                         stage changed what the coarse grid receives: the reference's own tree decides; option handoff_replay) */
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
/* (round 6) the same call's units that stayed with the fixed point although the census found a tie with a (sub-threshold) influence in them -- within 1e-4 s
 * of the reference by measurement, not by construction: the shim writes one line about them unless DSA_TIE_NOTE=0 --, the maps found tie-prone (a unit on
 * them holds a tie above the threshold; summed over the call's launches) and the units marched because of their map (option tie_map_strict). */
int dsa_dropin_tie_census(long long* tied_units_left, long long* tie_prone_maps, long long* flagged_by_map);
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
