// Device self-check of the hand-expanded divisions (dispersion_core.h: recip_of / div_by, ray_core.h: recipf_of / divf_by) against the
// compiler's IEEE division, operand pair by operand pair, bit for bit.  Not on any product path: tests/test_gpu_boundary.py runs it, so that
// "the same instructions on the same operands" is a measured statement and the operand ranges the headers name are the ones checked.
#include "../../include/dsurftomo_amd.h"
#include "kernels.h"
#include "dispersion_core.h"
#include "ray_core.h"

namespace dsa {

__device__ __forceinline__ unsigned long long sc_next(unsigned long long& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long z = s;
    z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33;
    return z;
}
// a double with a uniform mantissa, a random sign and an exponent drawn from [elo, ehi] (unbiased)
__device__ __forceinline__ double sc_f64(unsigned long long& s, int elo, int ehi)
{
    const unsigned long long r = sc_next(s), q = sc_next(s);
    const int e = elo + (int)(q % (unsigned long long)(ehi - elo + 1));
    const unsigned long long bits = ((r & 1ull) << 63) | ((unsigned long long)(e + 1023) << 52) | ((r >> 12) & 0xfffffffffffffull);
    return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ float sc_f32(unsigned long long& s, int elo, int ehi)
{
    const unsigned long long r = sc_next(s), q = sc_next(s);
    const int e = elo + (int)(q % (unsigned long long)(ehi - elo + 1));
    const unsigned bits = ((unsigned)(r & 1ull) << 31) | ((unsigned)(e + 127) << 23) | ((unsigned)(r >> 41) & 0x7fffffu);
    return __uint_as_float(bits);
}

// out[0]: fp64 pairs tried, out[1]: fp64 quotients that differ; out[2], out[3]: the same for fp32.  Every denominator serves `share`
// numerators, as it does in the product (the layer product's norm, a ray's cell sizes); numerators also take the special values 0, -0,
// +inf and NaN now and then (v_div_fixup's business).
__global__ void k_selfcheck_divisions(unsigned long long seed, int per_thread, int share, int nlo64, int nhi64, int dlo64, int dhi64,
                                      int nlo32, int nhi32, int dlo32, int dhi32, unsigned long long* __restrict__ out)
{
    unsigned long long s = seed ^ ((unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull);
    unsigned long long n64 = 0, bad64 = 0, n32 = 0, bad32 = 0;
    for (int i = 0; i < per_thread; ++i) {
        const double d = sc_f64(s, dlo64, dhi64);
        const Recip R = recip_of(d);
        const float df = sc_f32(s, dlo32, dhi32);
        const RecipF Rf = recipf_of(df);
        for (int k = 0; k < share; ++k) {
            double x = sc_f64(s, nlo64, nhi64);
            float xf = sc_f32(s, nlo32, nhi32);
            const unsigned sp = (unsigned)(sc_next(s) & 255ull);
            if (sp == 0) { x = 0.0; xf = 0.0f; } else if (sp == 1) { x = -0.0; xf = -0.0f; }
            else if (sp == 2) { x = __longlong_as_double(0x7ff0000000000000ll); xf = __uint_as_float(0x7f800000u); }
            else if (sp == 3) { x = __longlong_as_double(0x7ff8000000000000ll); xf = __uint_as_float(0x7fc00000u); }
            const double a = x / d, b = div_by(x, R);
            const float af = xf / df, bf = divf_by(xf, Rf);
            ++n64; ++n32;
            const bool nan64 = a != a && b != b, nan32 = af != af && bf != bf;      // (NaN payloads are not part of the contract)
            if (!nan64 && __double_as_longlong(a) != __double_as_longlong(b)) ++bad64;
            if (!nan32 && __float_as_uint(af) != __float_as_uint(bf)) ++bad32;
        }
    }
    atomicAdd(out + 0, n64); atomicAdd(out + 1, bad64); atomicAdd(out + 2, n32); atomicAdd(out + 3, bad32);
}

}  // namespace dsa

extern "C" int dsa_selfcheck_divisions(unsigned long long seed, int millions, const int* exponents8, unsigned long long* out4)
{
    using namespace dsa;
    if (!exponents8 || !out4 || millions < 1 || millions > 4096) return DSA_ERR_ARGUMENT;
    unsigned long long* d_out = nullptr;
    if (hipMalloc(&d_out, 4 * sizeof(unsigned long long)) != hipSuccess) return DSA_ERR_DEVICE;
    if (hipMemset(d_out, 0, 4 * sizeof(unsigned long long)) != hipSuccess) { (void)hipFree(d_out); return DSA_ERR_DEVICE; }
    const int share = 5, per_thread = 16, threads = 256;
    const long long pairs = (long long)millions * 1000000ll;
    const int blocks = (int)((pairs + (long long)threads * per_thread * share - 1) / ((long long)threads * per_thread * share));
    hipLaunchKernelGGL(k_selfcheck_divisions, dim3(blocks), dim3(threads), 0, 0, seed, per_thread, share, exponents8[0], exponents8[1], exponents8[2],
                       exponents8[3], exponents8[4], exponents8[5], exponents8[6], exponents8[7], d_out);
    const hipError_t rc = hipMemcpy(out4, d_out, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    return rc == hipSuccess ? 0 : DSA_ERR_INTERNAL;
}
