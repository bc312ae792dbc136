// Device-side math helpers shared by the kernels: wave reductions, exp, counter-based RNG.
#pragma once
#include <hip/hip_runtime.h>

// Fixed-order butterfly sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fp64 exp used in the hot loops.  Range reduction x = n ln2 + r, |r| <= ln2/2, degree-13 Taylor
// polynomial in Horner form (|rel err| < 3e-16 before the final scaling), result scaled by
// 2^n with v_ldexp_f64.  Underflows to 0 and overflows to +inf like exp(); a NaN argument is
// NOT propagated (the clamp swallows it) -- the host ABI rejects non-finite theta instead.
__device__ __forceinline__ double vb_exp(double x) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  x = fmin(fmax(x, -800.0), 800.0);  // saturate: exp(-800) -> 0, exp(800) -> +inf after ldexp
  double nf = __builtin_rint(x * LOG2E);
  double r = fma(nf, -LN2_HI, x);
  r = fma(nf, -LN2_LO, r);
  double p = 1.6059043836821614599e-10;            // 1/13!
  p = fma(p, r, 2.0876756987868098979e-09);        // 1/12!
  p = fma(p, r, 2.5052108385441718775e-08);        // 1/11!
  p = fma(p, r, 2.7557319223985890653e-07);        // 1/10!
  p = fma(p, r, 2.7557319223985890653e-06);        // 1/9!
  p = fma(p, r, 2.4801587301587301566e-05);        // 1/8!
  p = fma(p, r, 1.9841269841269841253e-04);        // 1/7!
  p = fma(p, r, 1.3888888888888888942e-03);        // 1/6!
  p = fma(p, r, 8.3333333333333332177e-03);        // 1/5!
  p = fma(p, r, 4.1666666666666664354e-02);        // 1/4!
  p = fma(p, r, 1.6666666666666665741e-01);        // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)nf);
}

// Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,c2,c3), key = (k0,k1).
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    unsigned n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Four standard normals for (sample b, component j, restart r, dim-block q4) under `seed`.
// Box-Muller on 24-bit uniforms in fp32 (the draws only need to be N(0,1) to MC accuracy; they
// are then *defined* as the fp64 values returned here, which vbmc_rng_dump reproduces bit for
// bit -- hence noinline: one body, identical code in every caller).
__device__ __noinline__ void vb_normal4(unsigned long long seed, unsigned b, unsigned j, unsigned r,
                                        unsigned q4, double z[4]) {
  unsigned c[4] = {b, j, r, q4};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-08f;      // (0,1)
    float u2 = (float)(c[2 * h + 1] >> 8) * 5.9604644775390625e-08f;            // [0,1)
    float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z[2 * h] = (double)(rad * cs);
    z[2 * h + 1] = (double)(rad * sn);
  }
}
