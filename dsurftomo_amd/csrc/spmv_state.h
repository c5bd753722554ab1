// Device copy of a COO matrix in row order and in column order (spmv.hip builds it, lsmr.hip uses it).
#pragma once

#include <vector>

#include "engine.h"

namespace dsa {

struct SpmvState {
    int m = 0, n = 0;
    long long nar = 0;
    // One ordering of the matrix (by row for A x, by column for A^T y) in slices of 64 segments, longest segments first:
    // entry k of the segment held by lane l of slice j sits at off[j] + 64 k + l, so the 64 lanes of a wavefront read 64
    // consecutive values / indices per step (coalesced) while every lane adds the entries of ITS segment in storage order.
    struct Sliced {
        DevBuf<long long> off;       // nslices + 1 slice offsets (entries)
        DevBuf<int> seg, len;        // per (slice, lane): segment (row / column) and its entry count; nslices * 64
        DevBuf<float> val;
        DevBuf<int> idx;             // 0-based index into the input vector (the unblocked part), or
        DevBuf<unsigned short> idx16; // block-local index (blocks of 32768 input elements)
        int nslices = 0;
        long long padded = 0;        // entries of storage (>= nar)
    };
    // One orientation of the matrix: the input vector is cut into blocks of `block` elements that fit the LDS; block b holds,
    // slice-transposed, the entries whose input index falls into it, for the segments whose entries are stored in ascending
    // input order (every data row and every column are; the reference's regularisation rows are not), and a product runs the
    // blocks one after the other -- so every output element still adds its entries in storage order, but the gathers of the
    // input vector are LDS reads instead of one 128-B L2 line per 4-byte operand.  blocks[nblocks] holds the other segments with
    // global indices (gathers from L2).  data_len: per block, how many entries of each (slice, lane) are data entries (DWS).
    struct Ordering {
        int block = 0, nblocks = 0, ninput = 0;
        std::vector<Sliced> blocks;
        std::vector<DevBuf<int>> data_len;
    };
    Ordering by_row, by_col;
    DevBuf<float> x, y;
    // LSMR work vectors (lsmr.hip)
    DevBuf<float> u, v, h, hbar, xs, localV, scal;
    // pinned host copies of u and v for the host-vector placement (they cross PCIe twice per LSMR iteration)
    float* hu = nullptr; float* hv = nullptr;
    size_t hu_cap = 0, hv_cap = 0;
};

// y += A x (mode 1; x: n, y: m) or x += A^T y (mode 2) on device vectors, on the engine's stream; every output
// element adds its entries in storage order (reference aprod.f90:7-60)
void spmv_device(Engine* e, int mode, float* d_x, float* d_y);

}  // namespace dsa
