// Device copy of a COO matrix in row order and in column order (spmv.hip builds it, lsmr.hip uses it).
#pragma once

#include "engine.h"

namespace dsa {

struct SpmvState {
    int m = 0, n = 0;
    long long nar = 0;
    DevBuf<long long> rowptr, colptr;
    DevBuf<float> val_r, val_c, x, y;
    DevBuf<int> col_r, row_c;
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
