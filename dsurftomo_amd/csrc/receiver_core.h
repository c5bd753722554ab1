// Receiver travel time of one ray from a unit's coarse field: reference srtimes (CalSurfG.f90:1636-1759) + bilinear (:2328-2349).
// Shared by the receiver kernel k_srtimes (stage_kernels.hip) and the tail of the coarse solve (fim_kernel.hip), which computes the
// unit's receiver times itself when its field slot is about to be handed to another unit.
#pragma once

#include "kernels.h"

namespace dsa {

// the travel time of record `id` as the field's storage has it: the compact field of the fixed-point solve (one float per node, or G of
// them for a bundle) ...
struct CompactFieldT {
    const float* Tc; int tstride;
    __device__ __forceinline__ float operator()(int id) const { return t_value(Tc[(size_t)id * tstride]); }
};
// returns false when the receiver lies outside the grid (the caller reports it); *t: the time, 0 where the reference leaves 0
template <class Field>
__device__ __forceinline__ bool receiver_time_f(const GridDesc& g, float scx, float scz, const RayDesc& rd, Field field,
                                                const float* __restrict__ veln, float dpl, float* t)
{
    const float gox = g.gox, goz = g.goz, dnx = g.dnx, dnz = g.dnz, earth = g.earth;
    const float rcx1 = rd.rx, rcz1 = rd.rz;
    int irx = (int)((rcx1 - gox) / dnx) + 1;
    int irz = (int)((rcz1 - goz) / dnz) + 1;
    *t = 0.0f;
    if (irx < 1 || irx > g.nnx || irz < 1 || irz > g.nnz) return false;
    if (irx == g.nnx) irx -= 1;
    if (irz == g.nnz) irz -= 1;
    const int isx = (int)((scx - gox) / dnx) + 1;
    const int isz = (int)((scz - goz) / dnz) + 1;
    float sred = sq((scx - rcx1) * earth);
    sred = sred + sq((scz - rcz1) * earth * rd.sin_rx);
    sred = sqrtf(sred);
    bool nearsrc = sred < dpl;
    if (isx == irx && isz == irz) nearsrc = true;
    float trr;
    const size_t ld = g.nnz;
    if (nearsrc) {
        // The reference does not clamp the source cell here (CalSurfG.f90:1703-1704), so a source on
        // the last node row/column makes it read one node past the grid.  Clamp the read instead.
        float vss[2][2];
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) {
                const int cx = min(isx - 1 + k - 1, g.nnx - 1), cz = min(isz - 1 + l - 1, g.nnz - 1);
                vss[k - 1][l - 1] = veln[(size_t)cx * ld + cz];
            }
        float drx = (scx - gox) - (float)(isx - 1) * dnx;
        float drz = (scz - goz) - (float)(isz - 1) * dnz;
        const float vels = bilinear4(vss, dnx, dnz, drx, drz);
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) vss[k - 1][l - 1] = veln[(size_t)(irx - 1 + k - 1) * ld + (irz - 1 + l - 1)];
        drx = (rcx1 - gox) - (float)(irx - 1) * dnx;
        drz = (rcz1 - goz) - (float)(irz - 1) * dnz;
        const float velr = bilinear4(vss, dnx, dnz, drx, drz);
        trr = 2.0f * sred / (vels + velr);
    } else {
        const float drx = (rcx1 - gox) - (float)(irx - 1) * dnx;
        const float drz = (rcz1 - goz) - (float)(irz - 1) * dnz;
        trr = 0.0f;
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) {
                const float produ = (1.0f - fabsf(((float)(l - 1) * dnz - drz) / dnz)) *
                                    (1.0f - fabsf(((float)(k - 1) * dnx - drx) / dnx));
                trr = trr + field(rec_index(g.nbz, irz - 1 + l - 1, irx - 1 + k - 1)) * produ;
            }
    }
    // A source inside the last cell next to a high model edge ends the reference's refined stage at once and
    // leaves its whole field at the initial 0 (the literal open-edge test, CalSurfG.f90:396-407); here such a
    // field is +inf (never reached).  Report the reference's 0 rather than a non-finite time.
    if (!(trr < kInf)) trr = 0.0f;
    *t = trr;
    return true;
}
__device__ __forceinline__ bool receiver_time(const GridDesc& g, float scx, float scz, const RayDesc& rd, const float* __restrict__ Tc,
                                              const float* __restrict__ veln, float dpl, float* t, int tstride = 1 /* floats between the field's nodes (bundles: G) */)
{
    return receiver_time_f(g, scx, scz, rd, CompactFieldT{ Tc, tstride }, veln, dpl, t);
}

}  // namespace dsa
