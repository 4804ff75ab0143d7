// Device-side special functions (float64) for the VB E-step kernels.
//
// The reference takes these from SciPy: scipy.special.psi at
// variational_bayes.py:177 / inferencer.py:17-18 and scipy.special.gammaln at
// variational_bayes.py:195,197.  They are re-derived here from the published
// definitions (upward recurrence + Bernoulli asymptotic series); arguments on
// this path are always > 0 (alpha > 0, eta >= beta > 0), so there is no
// reflection branch.  Pinned against scipy samples in tests/test_gpu_special.py.
#pragma once
#include <hip/hip_runtime.h>

namespace pylda {

// 1/x for normal positive x: v_rcp_f64 seed + two Newton steps (the refinement the
// compiler's IEEE division uses, without its scaling fix-ups).
__device__ __forceinline__ double rcp_newton(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// Asymptotic psi(x), valid to < 1e-16 absolute for x >= 10.
__device__ __forceinline__ double digamma_asymptotic(double x)
{
    const double inv = rcp_newton(x);
    const double w = inv * inv;
    // sum_{n>=1} B_2n / (2n x^2n)
    double s = 1.0 / 12.0;                       // highest kept term: x^-14
    s = fma(-s, w, 691.0 / 32760.0);
    s = fma(-s, w, 1.0 / 132.0);
    s = fma(-s, w, 1.0 / 240.0);
    s = fma(-s, w, 1.0 / 252.0);
    s = fma(-s, w, 1.0 / 120.0);
    s = fma(-s, w, 1.0 / 12.0);
    return log(x) - 0.5 * inv - s * w;
}

// psi(x), x > 0.  For x < 10 the ten recurrence terms
//   psi(x) = psi(x + 10) - sum_{i=0..9} 1/(x+i)
// are folded into ONE division (fp64 division is the expensive operation on
// the CDNA4 VALU): pairs (x+i)(x+9-i) share the numerator 2x+9, so
//   sum_i 1/(x+i) = (2x+9) * sum_{j=0..4} 1/q_j,   q_j = (x+j)(x+9-j)
// and the five reciprocals are combined over a common denominator.  All
// quantities are positive, so there is no cancellation.
__device__ __forceinline__ double digamma(double x)
{
    if (x >= 10.0) return digamma_asymptotic(x);
    const double q0 = x * (x + 9.0);
    const double q1 = (x + 1.0) * (x + 8.0);
    const double q2 = (x + 2.0) * (x + 7.0);
    const double q3 = (x + 3.0) * (x + 6.0);
    const double q4 = (x + 4.0) * (x + 5.0);
    // 1/q1+1/q2 = (q1+q2)/(q1 q2),  1/q3+1/q4 = (q3+q4)/(q3 q4)
    const double n12 = q1 + q2, d12 = q1 * q2;
    const double n34 = q3 + q4, d34 = q3 * q4;
    const double n1234 = fma(n12, d34, n34 * d12), d1234 = d12 * d34;
    // + 1/q0
    const double num = fma(n1234, q0, d1234), den = d1234 * q0;
    const double shift = (2.0 * x + 9.0) * (num * rcp_newton(den));
    return digamma_asymptotic(x + 10.0) - shift;
}

// exp(x) for |x| < 700, Estrin-evaluated degree-13 Taylor polynomial on the
// reduced argument |r| <= ln2/2 (truncation 4e-18): 1-2 ulp, dependency depth 9.
__device__ __forceinline__ double exp_shallow(double x)
{
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double p0 = fma(r, 1.0, 1.0);
    const double p1 = fma(r, 1.0 / 6.0, 0.5);
    const double p2 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    const double p3 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
    const double p4 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
    const double p5 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double p6 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double q0 = fma(p1, r2, p0), q1 = fma(p3, r2, p2), q2 = fma(p5, r2, p4);
    const double o0 = fma(q1, r4, q0), o1 = fma(p6, r4, q2);
    return ldexp(fma(o1, r8, o0), (int)kf);
}

// fma(x, m, a) with the multiplier in a scalar and the addend in a vector register, as ONE
// VOP3 instruction: for a constant addend the compiler otherwise copies it into the
// accumulator of a v_fmac first (an extra issue slot per term).
__device__ __forceinline__ double fma_scalar_mul(double x, double m, double a)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(m), "v"(a));
    return d;
}

// exp_shallow with its five two-constant terms issued through fma_scalar_mul
struct ExpCoef {
    double a2, a3, a4, a5, a6;      // addends 1/4!, 1/6!, 1/8!, 1/10!, 1/12!
    __device__ __forceinline__ void load()
    {
        a2 = 1.0 / 24.0, a3 = 1.0 / 720.0, a4 = 1.0 / 40320.0, a5 = 1.0 / 3628800.0, a6 = 1.0 / 479001600.0;
        asm volatile("" : "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
    }
};

__device__ __forceinline__ double exp_shallow_with(double x, const ExpCoef& k)
{
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double p0 = r + 1.0;
    const double p1 = fma(r, 1.0 / 6.0, 0.5);
    const double p2 = fma_scalar_mul(r, 1.0 / 120.0, k.a2);
    const double p3 = fma_scalar_mul(r, 1.0 / 5040.0, k.a3);
    const double p4 = fma_scalar_mul(r, 1.0 / 362880.0, k.a4);
    const double p5 = fma_scalar_mul(r, 1.0 / 39916800.0, k.a5);
    const double p6 = fma_scalar_mul(r, 1.0 / 6227020800.0, k.a6);
    const double q0 = fma(p1, r2, p0), q1 = fma(p3, r2, p2), q2 = fma(p5, r2, p4);
    const double o0 = fma(q1, r4, q0), o1 = fma(p6, r4, q2);
    return ldexp(fma(o1, r8, o0), (int)kf);
}

// exp(psi(x) - c) for x > 0 without the log/exp round trip:
//   psi(y) = log y - 1/(2y) - S(y)   =>   exp(psi(y) - z) = y * exp(-1/(2y) - S(y) - z)
// with y = x (x >= 10) or y = x + 10 and the recurrence shift
// sum_{i<10} 1/(x+i) = (2x+9) * sum_{j<5} 1/((x+j)(x+9-j)) added to z.
// Branch-free (lanes with gamma ~ alpha next to gamma ~ 100 cost nothing extra).  It runs in
// the gamma phase of the register kernels, where two wavefronts are active: Estrin instead
// of Horner keeps the dependency chains short, two reciprocals keep the issue count low.
// Coefficients of exp_digamma_minus that a kernel wants resident in VGPRs for the whole inner
// loop: 64-bit literals cannot be encoded in VOP3 and the scalar registers are taken by the
// exp() coefficients, so without this the compiler re-creates them with v_mov in every
// gamma phase (one issue slot each on a phase that is issue-bound).
struct ExpDigammaCoef {
    double nine, ten;
    double d3, d2, d1, d0;          // D(P)/P = P^4 + 60 P^3 + 1308 P^2 + 12176 P + 40320
    double n4, n3, n2, n1;          // D'(P)  = 5 P^4 + 240 P^3 + 3924 P^2 + 24352 P + 40320
    double b1, b2, b3, b4, b5, b6;  // B_2n / 2n, alternating signs folded in
    ExpCoef e;
    __device__ __forceinline__ double exp_of(double x) const { return exp_shallow_with(x, e); }
    __device__ __forceinline__ void load()
    {
        e.load();
        nine = 9.0, ten = 10.0;
        d3 = 60.0, d2 = 1308.0, d1 = 12176.0, d0 = 40320.0;
        n4 = 5.0, n3 = 240.0, n2 = 3924.0, n1 = 24352.0;
        b1 = 1.0 / 12.0, b2 = -1.0 / 120.0, b3 = 1.0 / 252.0, b4 = -1.0 / 240.0, b5 = 1.0 / 132.0, b6 = -691.0 / 32760.0;
        asm volatile("" : "+v"(nine), "+v"(ten), "+v"(d3), "+v"(d2), "+v"(d1), "+v"(d0), "+v"(n4), "+v"(n3));
        asm volatile("" : "+v"(n2), "+v"(n1), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6));
    }
};

template <typename Coef>
__device__ __forceinline__ double exp_digamma_minus_with(double x, double c, const Coef& k)
{
    const bool small = x < k.ten;
    const double y = small ? x + k.ten : x;
    const double inv = rcp_newton(y);
    // The ten recurrence terms pair up as 1/(x+j) + 1/(x+9-j) = (2x+9)/(P + c_j) with P = x(x+9)
    // and c = {0, 8, 14, 18, 20}; their sum is D'(P)/D(P) for D(P) = prod_j (P + c_j), two
    // polynomials with positive integer coefficients (no cancellation for P > 0) and ONE
    // reciprocal: v_rcp_f64 issues at quarter rate, and the gamma phase is issue-bound.
    // xs keeps D finite in lanes that discard the shift.
    const double xs = fmin(x, k.ten);
    const double P = xs * (xs + k.nine);
    double den = P + k.d3;
    den = fma(den, P, k.d2);
    den = fma(den, P, k.d1);
    den = fma(den, P, k.d0) * P;
    double num = fma(P, k.n4, k.n3);
    num = fma(num, P, k.n2);
    num = fma(num, P, k.n1);
    num = fma(num, P, k.d0);
    const double recip_sum = num * rcp_newton(den);
    const double shift = small ? fma(2.0, x, k.nine) * recip_sum : 0.0;
    // S(y) = w (a1 + a2 w + ... + a7 w^6), w = 1/y^2, alternating Bernoulli coefficients
    const double w = inv * inv, w2 = w * w, w4 = w2 * w2;
    const double p01 = fma(w, k.b2, k.b1);
    const double p23 = fma(w, k.b4, k.b3);
    const double p45 = fma(w, k.b6, k.b5);
    const double e0 = fma(p23, w2, p01), e1 = fma(k.b1, w2, p45);
    const double series = fma(e1, w4, e0) * w;
    const double tail = fma(-0.5, inv, -series) - (shift + c);     // psi(x) - log(y) - c
    return y * k.exp_of(tail);
}

// The same coefficients fetched into SCALAR registers from constant memory each time they are
// used (kernels that have no vector registers to spare: a VOP3 FMA takes one scalar operand, and
// literals the compiler hoists out of the inner loop would otherwise be spilled to scratch).
// The opaque pointer keeps the loads where they are written.
__constant__ double kExpDigammaTable[24] = {
    9.0, 10.0, 60.0, 1308.0, 12176.0, 40320.0, 5.0, 240.0, 3924.0, 24352.0,
    1.0 / 12.0, -1.0 / 120.0, 1.0 / 252.0, -1.0 / 240.0, 1.0 / 132.0, -691.0 / 32760.0,
    1.0 / 24.0, 1.0 / 720.0, 1.0 / 40320.0, 1.0 / 3628800.0, 1.0 / 479001600.0, 0.0, 0.0, 0.0};

struct ExpDigammaScalarCoef {
    double nine, ten, d3, d2, d1, d0, n4, n3, n2, n1, b1, b2, b3, b4, b5, b6;
    double a2, a3, a4, a5, a6;
    __device__ __forceinline__ void load()
    {
        typedef const double __attribute__((address_space(4)))* const_table_ptr;
        const_table_ptr t = (const_table_ptr)kExpDigammaTable;
        asm volatile("" : "+s"(t));
        nine = t[0], ten = t[1], d3 = t[2], d2 = t[3], d1 = t[4], d0 = t[5], n4 = t[6], n3 = t[7], n2 = t[8], n1 = t[9];
        b1 = t[10], b2 = t[11], b3 = t[12], b4 = t[13], b5 = t[14], b6 = t[15];
        a2 = t[16], a3 = t[17], a4 = t[18], a5 = t[19], a6 = t[20];
    }
    // exp_shallow with the two-constant terms taking their addend from the table
    __device__ __forceinline__ double exp_of(double x) const
    {
        const double kf = __builtin_rint(x * 1.4426950408889634074);
        double r = fma(-kf, 6.93147180369123816490e-01, x);
        r = fma(-kf, 1.90821492927058770002e-10, r);
        const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
        const double p0 = r + 1.0;
        const double p1 = fma(r, 1.0 / 6.0, 0.5);
        const double p2 = fma(r, 1.0 / 120.0, a2);
        const double p3 = fma(r, 1.0 / 5040.0, a3);
        const double p4 = fma(r, 1.0 / 362880.0, a4);
        const double p5 = fma(r, 1.0 / 39916800.0, a5);
        const double p6 = fma(r, 1.0 / 6227020800.0, a6);
        const double q0 = fma(p1, r2, p0), q1 = fma(p3, r2, p2), q2 = fma(p5, r2, p4);
        const double o0 = fma(q1, r4, q0), o1 = fma(p6, r4, q2);
        return ldexp(fma(o1, r8, o0), (int)kf);
    }
};

struct ExpDigammaLiterals {
    __device__ __forceinline__ double exp_of(double x) const { return exp_shallow(x); }
    static constexpr double nine = 9.0, ten = 10.0;
    static constexpr double d3 = 60.0, d2 = 1308.0, d1 = 12176.0, d0 = 40320.0;
    static constexpr double n4 = 5.0, n3 = 240.0, n2 = 3924.0, n1 = 24352.0;
    static constexpr double b1 = 1.0 / 12.0, b2 = -1.0 / 120.0, b3 = 1.0 / 252.0, b4 = -1.0 / 240.0, b5 = 1.0 / 132.0,
                            b6 = -691.0 / 32760.0;
};

__device__ __forceinline__ double exp_digamma_minus(double x, double c)
{
    return exp_digamma_minus_with(x, c, ExpDigammaLiterals());
}

// ln Gamma(x), x > 0: Stirling series for x >= 12, otherwise shifted up by
// the recurrence lnG(x) = lnG(x+m) - ln(x (x+1) ... (x+m-1)).
__device__ __forceinline__ double lgamma_stirling(double x)
{
    const double inv = rcp_newton(x);
    const double w = inv * inv;
    // sum_{n>=1} B_2n / (2n (2n-1) x^(2n-1))
    double s = 43867.0 / 244188.0;               // x^-17
    s = fma(-s, w, 3617.0 / 122400.0);           // x^-15
    s = fma(-s, w, 1.0 / 156.0);                 // x^-13
    s = fma(-s, w, 691.0 / 360360.0);            // x^-11
    s = fma(-s, w, 1.0 / 1188.0);                // x^-9
    s = fma(-s, w, 1.0 / 1680.0);                // x^-7
    s = fma(-s, w, 1.0 / 1260.0);                // x^-5
    s = fma(-s, w, 1.0 / 360.0);                 // x^-3
    s = fma(-s, w, 1.0 / 12.0);                  // x^-1
    const double half_log_2pi = 0.91893853320467274178;
    return (x - 0.5) * log(x) - x + half_log_2pi + s * inv;
}

__device__ __forceinline__ double lgamma_pos(double x)
{
    if (x >= 12.0) return lgamma_stirling(x);
    // product of x..x+11 can reach 23!/11! ~ 6.5e14 for x<12: no overflow.
    // For tiny x the product ~ x * 11! keeps full relative precision.
    double prod = x;
#pragma unroll
    for (int i = 1; i < 12; ++i) prod *= (x + (double)i);
    return lgamma_stirling(x + 12.0) - log(prod);
}

}  // namespace pylda
