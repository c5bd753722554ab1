// Internal interface between the engine (host) and the HIP kernels of the CalSurfG path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "source_stage.h"

namespace dsa {

// One fixed-point problem: a travel-time field on an (nnz, nnx) grid stored as tiled (T, tau)
// records (eikonal_core.h).  The records carry the boundary condition: pinned nodes (sign bit of T)
// are never recomputed, every other node starts at +inf.  `seed` lists the record indices to
// evaluate first (their queued bit, the sign bit of tau, is already set).
struct FimProblem {
    Rec* F;
    const float* slow;     // tiled like F
    const float* risti;
    const int* seed;
    const int* seed_count;
    int* lists;            // scratch for the active lists: 2 * list_cap + ready_cap ints
    int nnx, nnz, nbx, nbz;
    float ri, dnx, dnz;
    float window;          // causal window (seconds of travel time) evaluated per round
    int max_rounds;
    unsigned long long* clocks;   // optional, 8 u64: phase clocks of thread 0 (wall_clock64 ticks) + list sizes
    int32_t* info;         // 8 ints: [0] rounds, [1] rescans, [2] -1 = no convergence, [3] stall freezes, [4..5] evaluations (u64)
};

struct FimLaunch {
    int list_cap;          // entries per active list
    int ready_cap;         // entries of the dense ready list
    int threads;           // workgroup size: 256, 512 or 1024
};

size_t fim_lds_bytes(const FimLaunch& l);
void launch_fim(const FimProblem* d_problems, int nproblems, const FimLaunch& l, hipStream_t stream);

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
    Rec* F_c;                    // stride nbx*nbz*64 (tiled records)
    int* seed_c; int* nseed_c;   // stride kSeedC / 1
    int* lists; size_t lists_stride;   // active-list scratch, shared by the refined and the coarse solve
};
constexpr int kSeedR = kRWin * kRWin;              // the start-up march cannot pin more than its window
constexpr int kSeedC = kCWinMax * kCWinMax;

void launch_fill(float* d, size_t n, float v, hipStream_t stream);
void launch_refine(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_velv_all, size_t velv_stride,
                   const float* d_rbasis, hipStream_t stream);
void launch_refined_startup(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream);
void launch_handoff(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream);
void launch_coarse_march(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                         size_t field_stride, const float* d_risti_c, hipStream_t stream);
void launch_make_problems(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                          size_t field_stride, const float* d_risti_c, float window_r, float window_c,
                          FimProblem* d_prob_r, FimProblem* d_prob_c, int32_t* d_info, unsigned long long* d_clocks,
                          hipStream_t stream);

// receivers: one thread per ray; reference srtimes (CalSurfG.f90:1636-1759)
struct RayDesc { int src; float rx, rz; float sin_rx; };   // sin_rx = libm sinf(rx), made on the host
// RayDesc::src is a global unit index; unit_base is the first unit held by the batch arrays
void launch_srtimes(const GridDesc& g, const BatchPtrs& b, int unit_base, const RayDesc* d_rays, int nrays,
                    const float* d_veln_all, size_t field_stride, float dpl, float* d_out, int32_t* d_err,
                    hipStream_t stream);

}  // namespace dsa
