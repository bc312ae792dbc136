// Device-side math helpers shared by the kernels: wave reductions, exp, counter-based RNG.
#pragma once
#include <hip/hip_runtime.h>

// Fixed-order butterfly sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// v + (v of lane ^ 16) / v + (v of lane ^ 32) on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows of one
// operand with the even rows of the other, so with both operands = v the two results ARE the pair (v[lane], v[lane ^ 16]) in some order --
// and a sum does not care which.  Two instructions per double and no trip through the LDS pipeline (__shfl_xor: two ds_bpermute_b32 and
// their wait).  Bit-identical to v + __shfl_xor(v, 16 | 32, 64).
__device__ __forceinline__ double xor_sum16(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double xor_sum32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Wave sum without the LDS pipeline: lanes ^1, ^2 by quad permutes, the two halves of 8 and of 16 by the row (half-)mirror --
// after the quad steps every lane of a quad holds the quad's sum, so mirroring pairs the same groups an xor would --, the rows
// by the permlane swaps above.  Twelve VALU instructions per double where wave_sum's six __shfl_xor are twelve ds_bpermute_b32
// and their waits.  Fixed order (1, 2, 4, 8, 16, 32): deterministic, every lane ends with the total; NOT the order of wave_sum.
template <int CTRL>
__device__ __forceinline__ double dpp_pair_sum(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_valu(double v) {
  v = dpp_pair_sum<0xB1>(v);     // quad_perm [1, 0, 3, 2]
  v = dpp_pair_sum<0x4E>(v);     // quad_perm [2, 3, 0, 1]
  v = dpp_pair_sum<0x141>(v);    // row_half_mirror
  v = dpp_pair_sum<0x140>(v);    // row_mirror
  v = xor_sum16(v);
  return xor_sum32(v);
}

// Sums over the wave of NV per-lane values at once, as a TREE: lanes ^32 and ^16 by the permlane swaps with TWO values per exchange (the
// swap of registers X, Y leaves [X low half | Y low half] and [X high half | Y high half]: their sum is X folded into lanes 0-31 and Y
// into lanes 32-63 -- three instructions for the pair), then every register holds four values, one per row of sixteen lanes, reduced
// by the row steps of wave_sum_valu.  NV = 24: 126 VALU instructions where 24 wave_sum_valu are 720.  Fixed order: deterministic.
// On return register m = 0 .. (NV + 3) / 4 - 1 of `out` holds, in EVERY lane of row q = lane >> 4, the total of value 4 m + {0, 2, 1, 3}[q]
// (values beyond NV: zero).
__device__ __forceinline__ double swap_sum32(double x, double y) {
  const auto a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_sum16(double x, double y) {
  const auto a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
template <int NV>
__device__ __forceinline__ void wave_sum_tree(const double (&v)[NV], double (&out)[(NV + 3) / 4]) {
  constexpr int N4 = (NV + 3) / 4;
  double h[2 * N4];
#pragma unroll
  for (int i = 0; i < 2 * N4; ++i) {
    const double x = 2 * i < NV ? v[2 * i < NV ? 2 * i : 0] : 0.0, y = 2 * i + 1 < NV ? v[2 * i + 1 < NV ? 2 * i + 1 : 0] : 0.0;
    h[i] = swap_sum32(x, y);                  // lanes 0-31: value 2 i, lanes 32-63: value 2 i + 1
  }
#pragma unroll
  for (int m = 0; m < N4; ++m) {
    double t = swap_sum16(h[2 * m], h[2 * m + 1]);   // rows: values 4 m, 4 m + 2, 4 m + 1, 4 m + 3
    t = dpp_pair_sum<0xB1>(t);
    t = dpp_pair_sum<0x4E>(t);
    t = dpp_pair_sum<0x141>(t);
    out[m] = dpp_pair_sum<0x140>(t);
  }
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

static __constant__ double c_exp2_tab[256] = {  // 2^(j/256), correctly rounded (generated with mpmath)
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402,
};

// Table-driven exp: x = (256 m + j) ln2/256 + r, |r| <= ln2/512, exp(x) = 2^m * T[j] * (1 + expm1(r)),
// T[j] = 2^(j/256) from a 256-entry (2 KB) LDS table, expm1 by a degree-4 Taylor polynomial (remainder
// r^5/120 < 4e-17).  MODE 0 ("exact"): two-constant Cody-Waite reduction, |rel err| <= ~1 ulp over the whole
// range -- 10 fp64 + 3 int ops + one ds_read_b64 per value.  MODE 1 ("sum"): one-constant reduction, for
// terms of a sum that contains an O(1) term (the mixture density shifted by the sample's own component):
// the reduction error is |n| * 9e-20 absolute in r, i.e. a relative error of exp(x) of ~1e-16 * (1 + |x|/3),
// which weighted by exp(x) itself is < 1e-16 of the O(1) term for every x <= 0 -- one fp64 op fewer.
// Underflows to 0 / overflows to +inf through v_ldexp_f64; the argument must stay within +-5e6 (see below).
#define VB_EXP_TAB_N 256
template <int MODE = 0>
__device__ __forceinline__ double vb_exp_tab(double x, const double* __restrict__ tab) {
  const double INV = 369.3299304675746322841407;        // 256/ln2
  const double C_HI = 6.93147180369123816490e-01 / 256;  // ln2/256 split (exact scaling of fdlibm's pair)
  const double C_LO = 1.90821492927058770002e-10 / 256;
  const double C_1 = 0.693147180559945309417232 / 256;
  const double MAGIC = 6755399441055744.0;              // 1.5 * 2^52: round-to-nearest-integer by addition
  // n = rint(x * 256/ln2) through the magic-number trick: the integer lands in the low mantissa word,
  // so no v_rndne / v_cvt is needed.  Valid for |x| < 2^31 ln2/256 = 5.8e6; beyond that the low word
  // wraps -- hence the clamp below (arguments above +5e6 cannot occur: they would need exp() = inf anyway).
  // one-instruction lower clamp (plain v_max_f64; fmax() would add a canonicalising v_max first): keeps the
  // magic-number trick valid for arbitrarily negative arguments (tiny sigma_k early in a VBMC run)
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(-1.0e6));
  double t = fma(x, INV, MAGIC);
  int ni = __double2loint(t);
  double nf = t - MAGIC;
  double r;
  if (MODE == 0) {
    r = fma(nf, -C_HI, x);
    r = fma(nf, -C_LO, r);
  } else {
    r = fma(nf, -C_1, x);
  }
  double T = tab[ni & (VB_EXP_TAB_N - 1)];
  double v = r * r;
  double u = fma(r, 1.6666666666666665741e-01, 0.5);
  u = fma(v, 4.1666666666666664354e-02, u);
  double p = fma(v, u, r);         // expm1(r) = r + r^2 (1/2 + r/6 + r^2/24)
  return ldexp(fma(T, p, T), ni >> 8);
}

// Four independent exps in straight-line code (the scheduler interleaves the four Horner chains).
typedef double vb_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vb_d4 vb_exp4(vb_d4 x) {
  vb_d4 y;
  y[0] = vb_exp(x[0]); y[1] = vb_exp(x[1]); y[2] = vb_exp(x[2]); y[3] = vb_exp(x[3]);
  return y;
}
template <int MODE = 0>
__device__ __forceinline__ vb_d4 vb_exp_tab4(vb_d4 x, const double* __restrict__ tab) {
  vb_d4 y;
  y[0] = vb_exp_tab<MODE>(x[0], tab); y[1] = vb_exp_tab<MODE>(x[1], tab);
  y[2] = vb_exp_tab<MODE>(x[2], tab); y[3] = vb_exp_tab<MODE>(x[3], tab);
  return y;
}

// The entropy kernel's exp (entropy_mfma.h): the argument arrives PRE-SCALED, y = x * 1024/ln2 (the factor is folded into the
// S-step's MFMA operands), so the reduction is r' = y - rint(y) in [-1/2, 1/2] -- no multiply, and the saturating
// v_cvt_i32_f64 makes the clamp of the magic-number variant unnecessary (y < -2^31: index 0, exponent -2^21 -> 0; y > 2^31:
// +inf).  1024-entry table (8 KB of LDS), exp(x) = 2^(n >> 10) T[n & 1023] (1 + c r' (1 + c r'/2 + (c r')^2/6)), c = ln2/1024:
// remainder (c/2)^4/24 < 6e-16 before, 1.4e-16 after the economisation below.  11 VALU ops + one ds_read_b64 (vb_exp_tab<1>: 13; tools/exp_variants.hip: +11 % on the box).
// Accuracy as vb_exp_tab<1> ("sum" mode): the argument's own rounding, |x| 1e-16, dominates.
#define VB_EXP_TAB1K_N 1024
#define VB_EXP_TAB1K_SCALE 1477.3197218702985291365628   // 1024 / ln 2
// QUAD (round 5: what the matrix-core entropy kernel runs; the lane kernel and every other user keep the cubic): the polynomial behind the table is
// the economised QUADRATIC 1 + c r' (1 + c r'/2) -- one fused multiply-add fewer per value (10 VALU operations), relative error
// (c/2)^3/24 = 1.6e-12 of every term, an odd function of r' to leading order, so that over the terms of a mixture density it averages out:
// measured on the headline shape, H and dH move by 1.3e-16 and 2.2e-16 (profiles/r04_experiments.md section 2); the 50-digit vectors of
// tests/golden stay at 1e-11, and the per-value bound is tests/test_gpu_elbo.py::test_device_exp_sum_mode_accuracy.
// The accuracy this trades: a mixture with ONE component, or one that dominates, has no terms to average over -- H and dH
// then carry the per-value 1.6e-12.  Stated in INTEGRATION.md; tests/test_gpu_elbo.py::test_single_and_dominated_mixtures_against_the_50_digit_vectors
// bounds it against mpmath vectors (1e-11 holds there too).
#define VB_EXP_TAB1K_QUAD true
template <bool QUAD = false>
__device__ __forceinline__ double vb_exp_tab1k(double y, const double* __restrict__ tab) {
  const double c = 0.693147180559945309417232 / 1024;
  const double nr = __builtin_rint(y);
  // the conversion must SATURATE for |y| >= 2^31 (that is what makes a clamp unnecessary): the hardware instruction does, a
  // C++ cast of an out-of-range value is undefined -- hence the instruction itself
  int ni;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(ni) : "v"(nr));
  const double r = y - nr;
  const double T = tab[ni & (VB_EXP_TAB1K_N - 1)];
  // cubic: the dropped quartic term (c r')^4/24 is economised into the quadratic one (r'^4 ~ r'^2/4 - 1/128 on [-1/2, 1/2]:
  // Chebyshev), which leaves a remainder of c^4/24/64 = 1.4e-16 instead of 5.4e-16 at no cost.  Quadratic: the dropped cubic term
  // (c r')^3/6 is economised into the linear one (r'^3 ~ 3 r'/16): remainder c^3/6/32 = 1.6e-12 instead of 6.5e-12
  double u;
  if (QUAD) u = fma(r, c * c / 2, c + c * c * c / 32);
  else { u = fma(r, c * c * c / 6, c * c / 2 + c * c * c * c / 96); u = fma(r, u, c); }
  const double Tr = T * r;
  return ldexp(fma(Tr, u, T), ni >> 10);
}
// The same exp with the rounding by the magic-number addition: three full-rate adds in place of v_rndne_f64 + v_cvt_i32_f64 (both
// quarter-rate) + v_sub; the integer sits in the low mantissa bits of t -- table index = low 10 bits, binary exponent = bits 10..41 (one
// v_alignbit).  Valid for |y| < 2^41: the argument is clamped from below at -2^40 (exp -> 0 there anyway); large positive arguments
// do not occur (they would mean exp = inf).  For the VALU-only kernels (entropy_lane.h), where the quarter-rate pair is a third of the
// exp's cycles; the matrix-core kernel keeps vb_exp_tab1k (measured there in round 5: no gain).
template <bool QUAD = false>
__device__ __forceinline__ double vb_exp_tab1k_m(double y, const double* __restrict__ tab) {
  const double c = 0.693147180559945309417232 / 1024;
  const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
  asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(y), "v"(-1099511627776.0));
  const double t = y + MAGIC;
  const int lo = __double2loint(t), hi = __double2hiint(t);
  const double r = y - (t - MAGIC);
  const double T = tab[lo & (VB_EXP_TAB1K_N - 1)];
  double u;
  if (QUAD) u = fma(r, c * c / 2, c + c * c * c / 32);
  else { u = fma(r, c * c * c / 6, c * c / 2 + c * c * c * c / 96); u = fma(r, u, c); }
  const double Tr = T * r;
  return ldexp(fma(Tr, u, T), (int)__builtin_amdgcn_alignbit((unsigned)hi, (unsigned)lo, 10));
}
template <bool QUAD = false>
__device__ __forceinline__ vb_d4 vb_exp_tab1k4(vb_d4 y, const double* __restrict__ tab) {
  vb_d4 e;
  e[0] = vb_exp_tab1k<QUAD>(y[0], tab); e[1] = vb_exp_tab1k<QUAD>(y[1], tab);
  e[2] = vb_exp_tab1k<QUAD>(y[2], tab); e[3] = vb_exp_tab1k<QUAD>(y[3], tab);
  return e;
}

// 1/q for q > 0 finite: v_rcp_f64 seed + two Newton steps (<= 1 ulp), instead of the IEEE division sequence
// log(x) = vb_log_pos(x, &e) + ln2 * e for a positive normal x: the classic reduction x = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1,
// s = f / (2 + f), log(m) = f - (f^2/2 - s (f^2/2 + R(s^2))) with the degree-14 odd minimax R (< 1 ulp).  The coefficients are READ from
// constant memory through a pointer the compiler cannot see through: the entropy kernel takes one logarithm per (wave, segment), and as
// literals the library log's twelve polynomial constants were hoisted out of the segment loop and sat in VGPRs across every tile.
static __constant__ double c_vb_log_coef[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
                                        1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
__device__ __forceinline__ double vb_log_pos(double x, int* e) {
  const double* lc = c_vb_log_coef;
  asm volatile("" : "+s"(lc));
  int k = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  k -= lo ? 1 : 0;
  *e += k;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, lc[5], lc[3]), lc[1]);
  const double t2 = z * fma(w, fma(w, fma(w, lc[6], lc[4]), lc[2]), lc[0]);
  const double hfsq = 0.5 * f * f;
  return f - (hfsq - s * (hfsq + (t2 + t1)));
}

__device__ __forceinline__ double vb_rcp(double q) {
  double r = __builtin_amdgcn_rcp(q);
  double e = fma(-q, r, 1.0);
  r = fma(r, e, r);
  e = fma(-q, r, 1.0);
  return fma(r, e, r);
}

// Philox4x32-7 (Salmon, Moraes, Dror & Shaw, SC'11: seven rounds are the fewest that pass BigCrush for the 4x32 variant;
// ten is the conservative default), counter = (c0,c1,c2,c3), key = (k0,k1).  The 32x32 -> 64-bit products are written
// as one 64-bit multiply each (v_mad_u64_u32: one quarter-rate instruction instead of v_mul_hi + v_mul_lo).
#define VB_PHILOX_ROUNDS 7
__device__ __forceinline__ void philox4x32(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < VB_PHILOX_ROUNDS; ++i) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    unsigned n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Four standard normals for (sample b, component j, restart r, dim-block q4) under `seed`.
// Box-Muller on 24-bit uniforms in fp32 (the draws only need to be N(0,1) to MC accuracy; they
// are then *defined* as the fp64 values returned here, which vbmc_rng_dump reproduces bit for
// bit.  ONE body (vb_normal4i): integer arithmetic, exactly rounded fp32 operations none of which the compiler may
// contract or reassociate (an add followed by a multiply, single multiplies), and the hardware transcendentals -- so
// every inlined copy computes the same bits; the out-of-line vb_normal4v is that body behind a call (the dump kernel and the
// kernels that are short of registers use it).  Inlined into the MFMA entropy kernel (round 4) the key schedule and the
// first round's product with the (uniform) restart index run on the scalar unit and the call's argument / result moves go away.
__device__ __forceinline__ vb_d4 vb_normal4i(unsigned long long seed, unsigned b, unsigned j, unsigned r, unsigned q4) {
  vb_d4 z;
  unsigned c[4] = {b, j, r, q4};
  philox4x32(c, (unsigned)seed, (unsigned)(seed >> 32));
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
  return z;
}
__device__ __noinline__ vb_d4 vb_normal4v(unsigned long long seed, unsigned b, unsigned j, unsigned r, unsigned q4) {
  return vb_normal4i(seed, b, j, r, q4);
}
__device__ __forceinline__ void vb_normal4(unsigned long long seed, unsigned b, unsigned j, unsigned r, unsigned q4, double z[4]) {
  const vb_d4 v = vb_normal4v(seed, b, j, r, q4);
  z[0] = v[0]; z[1] = v[1]; z[2] = v[2]; z[3] = v[3];
}
