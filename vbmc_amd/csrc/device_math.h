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

static __constant__ double c_exp2_tab[64] = {  // 2^(j/64), correctly rounded (generated with mpmath)
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};

// Table-driven exp for the MFMA entropy kernel: x = (64 m + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^m * T[j] * (1 + expm1(r)), T[j] = 2^(j/64) from a 64-entry LDS table, expm1 by a
// degree-5 Taylor polynomial (remainder r^6/720 < 4e-17).  12 fp64 + 4 int ops + one ds_read_b64
// per value instead of 22 fp64 ops; |rel err| <= ~1 ulp.  Underflows to 0 / overflows to +inf through
// v_ldexp_f64; the argument must stay within +-2e7 (see below).
__device__ __forceinline__ double vb_exp_tab(double x, const double* __restrict__ tab) {
  const double INV = 92.332482616893656759;            // 64/ln2
  const double C_HI = 6.93147180369123816490e-01 / 64;  // ln2/64 split (exact scaling of fdlibm's pair)
  const double C_LO = 1.90821492927058770002e-10 / 64;
  const double MAGIC = 6755399441055744.0;              // 1.5 * 2^52: round-to-nearest-integer by addition
  // n = rint(x * 64/ln2) through the magic-number trick: the integer lands in the low mantissa word,
  // so no v_rndne / v_cvt is needed.  Valid for |x| < 2^31 ln2/64 = 2.3e7; beyond that the low word
  // wraps -- hence the clamp below (arguments above +2e7 cannot occur: they would need exp() = inf anyway).
  // one-instruction lower clamp (plain v_max_f64; fmax() would add a canonicalising v_max first): keeps the
  // magic-number trick valid for arbitrarily negative arguments (tiny sigma_k early in a VBMC run)
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(-1.0e6));
  double t = fma(x, INV, MAGIC);
  int ni = __double2loint(t);
  double nf = t - MAGIC;
  double r = fma(nf, -C_HI, x);
  r = fma(nf, -C_LO, r);
  double T = tab[ni & 63];
  double p = fma(r, 8.3333333333333332177e-03, 4.1666666666666664354e-02);
  p = fma(p, r, 1.6666666666666665741e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = p * r;                       // expm1(r)
  return ldexp(fma(T, p, T), ni >> 6);
}

// Four independent exps in straight-line code (the scheduler interleaves the four Horner chains).
typedef double vb_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vb_d4 vb_exp4(vb_d4 x) {
  vb_d4 y;
  y[0] = vb_exp(x[0]); y[1] = vb_exp(x[1]); y[2] = vb_exp(x[2]); y[3] = vb_exp(x[3]);
  return y;
}
__device__ __forceinline__ vb_d4 vb_exp_tab4(vb_d4 x, const double* __restrict__ tab) {
  vb_d4 y;
  y[0] = vb_exp_tab(x[0], tab); y[1] = vb_exp_tab(x[1], tab); y[2] = vb_exp_tab(x[2], tab); y[3] = vb_exp_tab(x[3], tab);
  return y;
}

// 1/q for q > 0 finite: v_rcp_f64 seed + two Newton steps (<= 1 ulp), instead of the IEEE division sequence
__device__ __forceinline__ double vb_rcp(double q) {
  double r = __builtin_amdgcn_rcp(q);
  double e = fma(-q, r, 1.0);
  r = fma(r, e, r);
  e = fma(-q, r, 1.0);
  return fma(r, e, r);
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
    // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in turns)
    float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
    float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);                   // sin/cos(2 pi u2)
    z[2 * h] = (double)(rad * cs);
    z[2 * h + 1] = (double)(rad * sn);
  }
}
