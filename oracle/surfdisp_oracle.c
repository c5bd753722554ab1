/* TEST INFRASTRUCTURE ONLY -- dispersion side of the oracle (filled in below). */
#include "dsurf_oracle.h"
