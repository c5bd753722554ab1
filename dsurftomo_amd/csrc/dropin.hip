// Drop-in level of the C ABI: the reference's CalSurfG / synthetic argument lists
// (reference src/CalSurfG.f90:939-943 and :2412-2415) on top of one process-wide engine.
#include <string>

#include "../../include/dsurftomo_amd.h"

namespace {
std::string g_dropin_error = "";
}

extern "C" {

const char* dsa_dropin_error(void) { return g_dropin_error.c_str(); }

int dsa_calsurfg(const int*, const int*, const int*, const int*, const float*, int*, float*, int*, float*,
                 const float*, const float*, const float*, const float*, const int*, const int*, const int*,
                 const int*, const double*, const double*, const double*, const double*, const int*, const int*,
                 const int*, const float*, const float*, const float*, const float*, const float*, const float*,
                 const int*, const int*, const int*, const int*, const int*, int*)
{
    g_dropin_error = "dsa_calsurfg: ray/Frechet and dispersion stages are not built yet";
    return DSA_ERR_STATE;
}

int dsa_synthetic(const int*, const int*, const int*, const int*, const float*, float*, const float*, const float*,
                  const float*, const float*, const int*, const int*, const int*, const int*, const double*,
                  const double*, const double*, const double*, const int*, const int*, const int*, const float*,
                  const float*, const float*, const float*, const float*, const float*, const int*, const int*,
                  const int*, const int*, const int*, const float*)
{
    g_dropin_error = "dsa_synthetic: dispersion stage is not built yet";
    return DSA_ERR_STATE;
}

}  // extern "C"
