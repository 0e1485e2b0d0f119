// fp64 elementary functions for the step loop, written for the argument ranges the simulations produce: a few ulp
// accurate, branch-free on the common path, and several times cheaper than the general-purpose library versions
// (which carry huge-argument reduction, double-double exponentials and loop-based fmod).  Arguments outside the
// fast range fall back to the library routine, so results stay defined everywhere.
//
// Polynomial kernels: the classic minimax coefficients of the Sun/fdlibm sin/cos kernels and the Taylor series of
// exp on |r| <= ln2/2 (truncation below 2^-57).
#pragma once

#include <hip/hip_runtime.h>

#define PSH_MATH __device__ __forceinline__

// d in [0.5, 4]: reciprocal refined to full precision (the hardware estimate plus two Newton steps)
PSH_MATH double psh_rcp_refined(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// n / d for moderate d (no scaling needed): one correction step after the refined reciprocal
PSH_MATH double psh_div_moderate(double n, double d) {
    const double r = psh_rcp_refined(d);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// min(max(x, lo), hi) for an x that is not a NaN.  As instructions of their own: fmin/fmax first canonicalise each operand
// (a v_max_f64 x, x per operand) because the kernels run in IEEE mode.
PSH_MATH double psh_clip(double x, double lo, double hi) {
    double y;
    asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(x), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(y) : "v"(y), "v"(hi));
    return y;
}

// psh_clip(x, -hi, hi): the negation is an operand modifier, not an instruction
PSH_MATH double psh_clip_sym(double x, double hi) {
    double y;
    asm("v_max_f64 %0, %1, -%2" : "=v"(y) : "v"(x), "v"(hi));
    asm("v_min_f64 %0, %1, %2" : "=v"(y) : "v"(y), "v"(hi));
    return y;
}

// Plain v_min_f64 / v_max_f64 (fmin / fmax would first canonicalise each operand, see psh_clip).  One of the two operands
// comes back unchanged; a NaN only if both are.
PSH_MATH double psh_min_raw(double a, double b) {
    double y;
    asm("v_min_f64 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
PSH_MATH double psh_max_raw(double a, double b) {
    double y;
    asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}

// One Horner step p z + c for a literal c.  Written out because the compiler selects the two-address form for it
// (v_fmac_f64 with c as the accumulator), which wants c in a vector register pair: two v_mov_b32 per step on the
// vector ALU.  Here c sits in a scalar register pair (two s_mov_b32 on the scalar unit); the arithmetic is the same fma.
PSH_MATH double psh_horner(double p, double z, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
}

// n / d given r ~ 1/d (relative error far below 2^-26): product plus one residual correction
PSH_MATH double psh_div_by(double n, double d, double r) {
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// The two quotients for the simulations' dynamics: the special cases of IEEE division put back (v_div_fixup_f64: an infinite
// numerator or divisor, a zero divisor, NaNs), which the residual correction turns into NaN -- inf - inf, 0 x inf.
PSH_MATH double psh_quot_by(double n, double d, double r) { return __builtin_amdgcn_div_fixup(psh_div_by(n, d, r), d, n); }
PSH_MATH double psh_quot(double n, double d) { return __builtin_amdgcn_div_fixup(psh_div_moderate(n, d), d, n); }

// The nearest integer to x c for |x c| < 2^31, as a double and as an int: the magic-number addition 1.5 * 2^52 leaves it
// in the low word of the sum in two's complement (an fma and a subtraction, one rounding, instead of a product, v_rndne
// and a quarter-rate conversion).
PSH_MATH double psh_round_scaled(double x, double c, int& n) {
    const double t = fma(x, c, 6755399441055744.0);
    n = __double2loint(t);
    return t - 6755399441055744.0;
}

// exp(x) - 1 split as (k, q): exp(x) = 2^k * (1 + q), |q| <= 0.42.  Shared by exp, tanh and sigmoid.
PSH_MATH double psh_exp_poly(double x, int& k) {
    const double kd = psh_round_scaled(x, 1.4426950408889634074, k);
    double r = fma(-kd, 6.93147180369123816490e-01, x);   // ln2 high part (33 bits): the product is exact
    r = fma(-kd, 1.90821492927058770002e-10, r);          // ln2 low part
    // q = r + r^2/2! + ... + r^13/13!
    double p = 1.6059043836821613e-10;                     // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                   // 1/12!
    p = psh_horner(p, r, 2.505210838544172e-08);                  // 1/11!
    p = psh_horner(p, r, 2.755731922398589e-07);                  // 1/10!
    p = psh_horner(p, r, 2.7557319223985893e-06);                 // 1/9!
    p = psh_horner(p, r, 2.48015873015873e-05);                   // 1/8!
    p = psh_horner(p, r, 1.984126984126984e-04);                  // 1/7!
    p = psh_horner(p, r, 1.388888888888889e-03);                  // 1/6!
    p = psh_horner(p, r, 8.333333333333333e-03);                  // 1/5!
    p = psh_horner(p, r, 4.1666666666666664e-02);                 // 1/4!
    p = psh_horner(p, r, 1.6666666666666666e-01);                 // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    return p * r;
}

// exp for |x| <= 746 (psh_exp_clamp brings any other argument there)
PSH_MATH double psh_exp_core(double x) {
    int k;
    const double q = psh_exp_poly(x, k);
    return ldexp(1.0 + q, k);
}
#define PSH_EXP_FAST 700.0   // arguments up to here need no clamp
PSH_MATH double psh_exp_clamp(double x) {
    x = (x < -746.0) ? -746.0 : x;   // NaN stays NaN (comparisons are false); below this exp underflows to 0
    return (x > 710.0) ? 710.0 : x;  // above this exp overflows to +inf
}
PSH_MATH double psh_exp(double x) {
    if (!(fabs(x) <= PSH_EXP_FAST)) {    // rare: one comparison on the common path instead of two and four selects
        asm volatile("");                // (keeps it a branch: as selects the three comparisons would all run)
        x = psh_exp_clamp(x);
    }
    return psh_exp_core(x);
}

// log for a positive, normal, finite x; anything else takes the library routine.  The library's log carries a
// double-double reduction for its last half ulp (about 100 vector instructions); this is the classic scheme in plain
// doubles (about 35): x = 2^k m with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log m = f - f^2/2 +
// s (f^2/2 + R(s^2)) with the 7-term minimax polynomial R of the Sun/fdlibm kernel (truncation below 2^-58), log 2 in
// two parts.  Within 1 ulp.
PSH_MATH double psh_log_normal(double x) {   // x positive, normal and finite
    double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 7.07106781186547524401e-01;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double dk = (double)k;
    const double f = m - 1.0;
    const double s = psh_div_moderate(f, 2.0 + f);   // the denominator lies in [1.70, 2.42)
    const double z = s * s, w = z * z;
    double t1 = fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);   // Lg6, Lg4
    t1 = psh_horner(t1, w, 3.999999999940941908e-01);                          // Lg2
    double t2 = fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);   // Lg7, Lg5
    t2 = psh_horner(t2, w, 2.857142874366239149e-01);                          // Lg3
    t2 = psh_horner(t2, w, 6.666666666666735130e-01);                          // Lg1
    const double R = fma(t2, z, t1 * w);
    const double hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
PSH_MATH double psh_log(double x) {
    if (!(x >= 2.2250738585072014e-308 && x <= 1.79e308)) return log(x);
    return psh_log_normal(x);
}

PSH_MATH double psh_tanh(double x) {
    const double a = fabs(x);
    double y = -2.0 * a;
    y = (y < -80.0) ? -80.0 : y;     // tanh is 1 to the last bit long before; keeps k small
    int k;
    const double q = psh_exp_poly(y, k);
    // E = exp(-2a) - 1 without cancellation: for k == 0 it is q itself
    const double E = (k == 0) ? q : (ldexp(1.0 + q, k) - 1.0);
    const double t = psh_div_moderate(-E, 2.0 + E);   // (1 - e)/(1 + e), denominator in (1, 2]
    return copysign(t, x);
}

// tanh from a table and the addition theorem, in two variants that differ in table spacing.  Both split
// |x| = a_j + d with a_j the nearest table node, take T_j = tanh(a_j) from the table (`tab` may live in LDS or in
// global memory; the lookup runs on the LDS pipe, not on the vector ALU) and tanh d from its odd series.
//
// Fine table, tanh(j/256), j = 0..5120, |d| <= 2^-9 (17 fp64 + 3 32-bit vector instructions, about 1.2 ulp, 2 ulp next to 0):
//   tanh(a_j + d) = T + (1 - T^2) td / (1 + T td),   1/(1 + u) = (1 - u)(1 + u^2 + u^4) + O(u^6),  u = T td <= 2^-9.
// Adding 1.5 * 2^44 rounds |x| to a multiple of 1/256 and leaves that multiple in the low word of the sum.
// The work is split into an index step and a finish step so that a caller can issue all its table loads first.
#define PSH_TANH8_N 5121
//
// |x| is clamped in the integer domain (two 32-bit instructions instead of an fp64 compare and two selects): with
// the high word limited to that of 20.0, a lies in [20, 20 + 2^-16), which is node 5120 with T = 1 and 1 - T^2 = 0,
// so the result is exactly 1.  The clamp also turns a NaN into ~20: the caller gets |x|'s high word back (`hi_abs`)
// and restores the NaN itself; x is a NaN iff hi_abs > PSH_HI_INF (arithmetic only produces quiet NaNs, whose high
// word is >= 0x7ff80000).
#define PSH_HI_INF 0x7ff00000u
PSH_MATH unsigned psh_tanh8_index(double x, double* d, unsigned* hi_abs) {
    const unsigned hi = (unsigned)__double2hiint(x) & 0x7fffffffu;
    *hi_abs = hi;
    const unsigned hc = hi < 0x40340000u ? hi : 0x40340000u;           // high word of 20.0
    const double a = __hiloint2double((int)hc, __double2loint(x));
    const double jm = a + 26388279066624.0;    // 1.5 * 2^44
    *d = a - (jm - 26388279066624.0);
    return (unsigned)__double2loint(jm);       // <= 5120 by the clamp
}
// The same split for an argument KNOWN to lie within [-20, 20] (a hidden layer behind a tanh layer whose weights bound
// its pre-activations): no clamp of |x| and no NaN bookkeeping, five 32-bit instructions less per element.  A NaN
// needs no care here: it makes d a NaN, which the finish step carries into the result, and the index is limited to
// the table, so whatever its low word holds, a table entry is read.
PSH_MATH unsigned psh_tanh8_index_bounded(double x, double* d) {
    const double a = fabs(x);
    const double jm = a + 26388279066624.0;    // 1.5 * 2^44
    *d = a - (jm - 26388279066624.0);
    const unsigned j = (unsigned)__double2loint(jm);
    return j < (unsigned)(PSH_TANH8_N - 1) ? j : (unsigned)(PSH_TANH8_N - 1);
}
// The same for an argument known to be FINITE and below 2^40 in magnitude whose table index needs no limit either: either
// |x| < 20 is known, or the caller limits the index it gets (beyond 20 the split is still exact, node 5120 has T = 1 and
// the finish step then returns exactly 1: td - T u = 0).
PSH_MATH unsigned psh_tanh8_index_free(double x, double* d) {
    const double a = fabs(x);
    const double jm = a + 26388279066624.0;    // 1.5 * 2^44
    *d = a - (jm - 26388279066624.0);
    return (unsigned)__double2loint(jm);
}
// Signed form for a table that continues to negative nodes, tanh(j/256) for j = -5120..5120 (`tab` points at node 0):
// x + 1.5 * 2^44 rounds x itself to a multiple of 1/256, the signed multiple is the low word of the sum in two's
// complement, and d, T, td keep their signs through the finish step (u = T td and w are even in x, g and T are odd), so
// neither |x| nor the final sign transfer is needed.  For x > 0 the bits of the unsigned form; for x < 0 its mirror image,
// except that a tie between two nodes may go to the other one (both are within 2^-9, the results agree to rounding).
// For finite |x| < 2^40; the caller limits the index to +-5120 unless |x| < 20 is known.
#define PSH_TANH_ODD_N 10241
PSH_MATH int psh_tanh8_index_signed(double x, double* d) {
    const double jm = x + 26388279066624.0;    // 1.5 * 2^44
    *d = x - (jm - 26388279066624.0);
    return __double2loint(jm);
}
PSH_MATH double psh_tanh8_finish_signed(double d, double T) {
    const double z = d * d;
    const double p = fma(z, 1.3333333333333333e-01, -3.3333333333333331e-01) * z;
    const double td = fma(p, d, d);
    const double u = T * td;
    const double r0 = 1.0 - u;
    const double e = u * u;
    const double w = fma(r0, fma(e, e, e), r0);
    const double g = fma(-T, u, td);
    return fma(w, g, T);
}
PSH_MATH double psh_tanh8_finish(double x, double d, double T) {
    const double z = d * d;
    const double p = fma(z, 1.3333333333333333e-01, -3.3333333333333331e-01) * z;   // 2/15, -1/3; d^7 is below 2^-58
    const double td = fma(p, d, d);
    const double u = T * td;
    const double r0 = 1.0 - u;
    const double e = u * u;
    const double w = fma(r0, fma(e, e, e), r0);
    // (1 - T^2) td w = w (td - T u): one instruction less than forming 1 - T^2; the cancellation in td - T u loses
    // bits of a term that is itself below 2^-9 of the result
    const double g = fma(-T, u, td);
    return copysign(fma(w, g, T), x);
}
PSH_MATH double psh_tanh8(double x, const double* tab) {
    double d;
    unsigned hi;
    const unsigned j = psh_tanh8_index(x, &d, &hi);
    const double r = psh_tanh8_finish(x, d, tab[j]);
    return hi > PSH_HI_INF ? x : r;
}

// Coarse table, tanh(j/64), j = 0..1280, |d| <= 2^-7 (23 vector instructions, 2 ulp): for kernels whose LDS has no
// room for the fine table.  tanh(a_j + d) = (T + td)/(1 + T td) with a series seed of the reciprocal and one
// residual correction of the quotient.
#define PSH_TANH_TAB_N 1281
PSH_MATH double psh_tanh_tab(double x, const double* tab, unsigned* hi_abs) {
    const unsigned hi = (unsigned)__double2hiint(x) & 0x7fffffffu;   // clamp and NaN handling as in psh_tanh8_index
    *hi_abs = hi;
    const unsigned hc = hi < 0x40340000u ? hi : 0x40340000u;
    const double a = __hiloint2double((int)hc, __double2loint(x));
    const double jm = a + 105553116266496.0;    // 1.5 * 2^46: rounds to a multiple of 1/64, the multiple in the low word
    const double d = a - (jm - 105553116266496.0);
    const double T = tab[(unsigned)__double2loint(jm)];
    const double z = d * d;
    double p = fma(z, -5.396825396825397e-02, 1.3333333333333333e-01);   // -17/315, 2/15
    p = fma(p, z, -3.3333333333333331e-01);
    p = p * z;
    const double td = fma(p, d, d);
    const double u = T * td;
    const double num = T + td;
    const double r0 = 1.0 - u;
    const double r1 = fma(r0, u * u, r0);       // (1 - u)(1 + u^2) = (1 - u^4)/(1 + u)
    const double q = num * r1;
    const double res = fma(-u, q, num - q);     // num - (1 + u) q; num - q is exact
    return copysign(fma(res, r1, q), x);
}

PSH_MATH double psh_sigmoid(double x) {
    // 1/(1 + exp(-x)); for x < 0 use exp(x)/(1 + exp(x)) so that the exponential never overflows
    const double e = psh_exp(-fabs(x));
    const double s = psh_div_moderate(1.0, 1.0 + e);  // sigmoid(|x|)
    return (x >= 0.0) ? s : psh_div_moderate(e, 1.0 + e);
}

// sin and cos together.  Cody-Waite reduction by pi/2 in three parts, exact for |x| <= 1e5; larger or non-finite
// arguments take the library path.
PSH_MATH void psh_sincos_moderate(double x, double* s, double* c) {   // |x| <= 1e5
    int n;
    const double nd = psh_round_scaled(x, 6.36619772367581382433e-01, n);  // 2/pi
    double r = fma(-nd, 1.57079632673412561417e+00, x);      // pi/2, first 33 bits
    r = fma(-nd, 6.07710050630396597660e-11, r);             // next 33 bits
    r = fma(-nd, 2.02226624879595063154e-21, r);             // tail
    const double z = r * r;
    // sin kernel on [-pi/4, pi/4]
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = psh_horner(ps, z, 2.75573137070700676789e-06);
    ps = psh_horner(ps, z, -1.98412698298579493134e-04);
    ps = psh_horner(ps, z, 8.33333333332248946124e-03);
    ps = psh_horner(ps, z, -1.66666666666666324348e-01);
    const double sn = fma(ps * z, r, r);
    // cos kernel on [-pi/4, pi/4]
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = psh_horner(pc, z, -2.75573143513906633035e-07);
    pc = psh_horner(pc, z, 2.48015872894767294178e-05);
    pc = psh_horner(pc, z, -1.38888888888741095749e-03);
    pc = psh_horner(pc, z, 4.16666666666666019037e-02);
    const double cs = fma(pc * z, z, fma(-0.5, z, 1.0));
    const double ss = (n & 1) ? cs : sn;
    const double cc = (n & 1) ? sn : cs;
    // quadrants 2, 3 negate the sine and quadrants 1, 2 the cosine: bit 1 of n (of n + 1) moved onto the sign bit
    *s = __hiloint2double(__double2hiint(ss) ^ (int)(((unsigned)n << 30) & 0x80000000u), __double2loint(ss));
    *c = __hiloint2double(__double2hiint(cc) ^ (int)(((unsigned)(n + 1) << 30) & 0x80000000u), __double2loint(cc));
}
PSH_MATH void psh_sincos(double x, double* s, double* c) {
    if (!(fabs(x) <= 1.0e5)) {
        sincos(x, s, c);
        return;
    }
    psh_sincos_moderate(x, s, c);
}

// C fmod (sign of x, exact) for quotients below 2^52; library path otherwise
PSH_MATH double psh_fmod(double x, double m) {
    const double am = fabs(m);
    if (fabs(x) < am) return x;   // already reduced: the common case in the step loop
    const double qd = x / am;
    if (!(fabs(qd) < 4503599627370496.0) || am == 0.0) return fmod(x, m);
    const double q = trunc(qd);
    double r = fma(-q, am, x);
    // x / am may have rounded up to the next integer: the remainder then has the wrong sign
    if (x > 0.0 && r < 0.0) r += am;
    if (x < 0.0 && r > 0.0) r -= am;
    return (r == 0.0) ? copysign(0.0, x) : r;
}
