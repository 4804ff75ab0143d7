// Device-side special functions (float64) for the VB E-step kernels.
//
// The reference takes these from SciPy: scipy.special.psi at
// variational_bayes.py:177 / inferencer.py:17-18 and scipy.special.gammaln at
// variational_bayes.py:195,197.  They are re-derived here from the published
// definitions (upward recurrence + Bernoulli asymptotic series); arguments on
// this path are always > 0 (alpha > 0, eta >= beta > 0), so there is no
// reflection branch.  Pinned against scipy samples (tests/golden/special_fn.npz) by tests/test_gpu_estep.py::test_device_special_functions.
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

// psi'(x) (scipy.special.polygamma(1, x)), x > 0: recurrence psi'(x) = psi'(x + 1) + 1 / x^2 up to x >= 12, then
//   psi'(y) = 1/y + 1/(2 y^2) + sum_n B_2n / y^(2n+1)        (truncation < 2e-18 relative at y = 12)
// Only the alpha update (mstep_kernels.h, K evaluations per Newton iteration) uses it: plain divisions.
__device__ __forceinline__ double trigamma(double x)
{
    double shift = 0.0;
    while (x < 12.0) {
        shift += 1.0 / (x * x);
        x += 1.0;
    }
    const double inv = 1.0 / x, w = inv * inv;
    double s = 7.0 / 6.0;                         // B_14
    s = fma(s, w, -691.0 / 2730.0);               // B_12
    s = fma(s, w, 5.0 / 66.0);                    // B_10
    s = fma(s, w, -1.0 / 30.0);                   // B_8
    s = fma(s, w, 1.0 / 42.0);                    // B_6
    s = fma(s, w, -1.0 / 30.0);                   // B_4
    s = fma(s, w, 1.0 / 6.0);                     // B_2
    return shift + (inv + 0.5 * w + s * w * inv);
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
    // (the clamp: x below ~1e-50 makes the exponent -1/x too large for exp's argument reduction, or -inf; the
    //  result is a clean 0 either way)
    const double tail = fmax(fma(-0.5, inv, -series) - (shift + c), -1100.0);     // psi(x) - log(y) - c
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

// ---- exp(psi(x) - c), ordered level by level --------------------------------------------------
// The gamma phase of the register kernels is ONE dependent chain per thread, run by a lone
// wavefront per SIMD while the document's other wavefronts wait at a barrier: tools/valu_bench.hip
// puts a dependent fp64 instruction 12.8 ticks behind its producer, an independent one 4.9.  hipcc
// schedules these kernels for register pressure and ran the ~140 instructions of
// exp_digamma_minus_with at ~10 ticks each (s_memtime stamps: 1400-1470 ticks per gamma phase).
// This form fixes the order by hand instead: the instructions of one dependency LEVEL are written
// together and a scheduling barrier keeps the levels apart, so two or three independent chains are
// always in flight (recurrence shift | asymptotic series, then the even | odd halves of the exp
// polynomial).  It is also shorter:
//   * always y = x + 10 (the recurrence holds for every x > 0; the compare/select pairs go), with the
//     shift polynomials evaluated at min(x, 1e25) so that they stay finite (beyond that the shift is
//     < 1e-24 and drops out of the sum);
//   * every FMA takes at most ONE scalar constant (VOP3 reads one SGPR pair): Horner chains on the even
//     and odd coefficients instead of Estrin pairs, no v_mov of constants into accumulators;
//   * the two coefficient tables sit in scalar registers one after the other (the second is fetched
//     while the first half computes), 34 + 28 SGPRs, never more than 42 at a time.
// 63 VALU instructions, 30 levels.  Pinned to scipy like the other forms
// (tests/test_gpu_estep.py::test_device_fused_exp_digamma).
__constant__ double kExpDigammaLevelsA[16] = {
    10.0, 1e25, 9.0, 60.0, 1308.0, 12176.0, 40320.0, 240.0, 3924.0, 24352.0,
    1.0 / 12.0, -1.0 / 120.0, 1.0 / 252.0, -1.0 / 240.0, 1.0 / 132.0, -691.0 / 32760.0};
__constant__ double kExpDigammaLevelsB[14] = {
    1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
    1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
    1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0};

// d = a * b + s  /  d = a * s + v  /  d = min(a, s), exactly these VOP3 encodings (s: scalar register pair)
__device__ __forceinline__ double fma_vvs(double a, double b, double s)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(s));
    return d;
}
__device__ __forceinline__ double min_vs(double a, double s)
{
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(s));
    return d;
}
__device__ __forceinline__ double mov_s(double s)
{
    double d;
    asm("v_mov_b64_e32 %0, %1" : "=v"(d) : "s"(s));
    return d;
}

#define PYLDA_LEVEL() __builtin_amdgcn_sched_barrier(0)

// the first table (the recurrence shift and the asymptotic series): fetched by the caller ahead of the
// barrier in front of the gamma phase, so that the scalar-cache round trip is not on the chain
struct ExpDigammaLevelsA {
    double ten, big, nine, d3, d2, d1, d0, n3, n2, n1, b1, b2, b3, b4, b5, b6;
    __device__ __forceinline__ void load()
    {
        typedef const double __attribute__((address_space(4)))* const_table_ptr;
        const_table_ptr t = (const_table_ptr)kExpDigammaLevelsA;
        asm volatile("" : "+s"(t));
        ten = t[0], big = t[1], nine = t[2], d3 = t[3], d2 = t[4], d1 = t[5], d0 = t[6], n3 = t[7], n2 = t[8], n1 = t[9];
        b1 = t[10], b2 = t[11], b3 = t[12], b4 = t[13], b5 = t[14], b6 = t[15];
    }
};

// the second table (exp): fetched inside exp_digamma_minus_levels while the first half computes (PRELOADED
// false: its scalar registers are then the first table's), or by the caller together with the first one
struct ExpDigammaLevelsB {
    double log2e, nln2hi, nln2lo, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13;
    __device__ __forceinline__ void load()
    {
        typedef const double __attribute__((address_space(4)))* const_table_ptr;
        const_table_ptr t = (const_table_ptr)kExpDigammaLevelsB;
        asm volatile("" : "+s"(t));
        log2e = t[0], nln2hi = t[1], nln2lo = t[2];
        c3 = t[3], c4 = t[4], c5 = t[5], c6 = t[6], c7 = t[7], c8 = t[8], c9 = t[9], c10 = t[10], c11 = t[11], c12 = t[12], c13 = t[13];
    }
};

template <bool PRELOADED = false>
__device__ __forceinline__ double exp_digamma_minus_levels(double x, double c, const ExpDigammaLevelsA& k,
                                                           const ExpDigammaLevelsB* kb_in = nullptr)
{
    const double ten = k.ten, big = k.big, nine = k.nine, d3 = k.d3, d2 = k.d2, d1 = k.d1, d0 = k.d0;
    const double n3 = k.n3, n2 = k.n2, n1 = k.n1, b1 = k.b1, b2 = k.b2, b3 = k.b3, b4 = k.b4, b5 = k.b5, b6 = k.b6;
    PYLDA_LEVEL();
    const double y = x + ten;                                   // 1
    const double xs = min_vs(x, big);
    const double vb5 = mov_s(b5);
    PYLDA_LEVEL();
    double inv = __builtin_amdgcn_rcp(y);                       // 2
    const double xs9 = xs + nine;
    const double vb4 = mov_s(b4);
    PYLDA_LEVEL();
    const double P = xs * xs9;                                  // 3
    double e = fma(-y, inv, 1.0);
    const double tx = xs + xs9;                                 //    2x + 9
    PYLDA_LEVEL();
    double den = P + d3;                                        // 4
    double num = P + n3;                                        //    D'(P) = 5 P^4 + 240 P^3 + ... : 5 P + 240 = 4 P + (P + 240)
    inv = fma(inv, e, inv);
    PYLDA_LEVEL();
    den = fma_vvs(den, P, d2);                                  // 5
    num = fma(P, 4.0, num);
    e = fma(-y, inv, 1.0);
    PYLDA_LEVEL();
    den = fma_vvs(den, P, d1);                                  // 6
    num = fma_vvs(num, P, n2);
    inv = fma(inv, e, inv);
    PYLDA_LEVEL();
    den = fma_vvs(den, P, d0);                                  // 7
    num = fma_vvs(num, P, n1);
    const double w = inv * inv;
    PYLDA_LEVEL();
    den = den * P;                                              // 8
    num = fma_vvs(num, P, d0);
    const double w2 = w * w;
    ExpDigammaLevelsB kb;
    if constexpr (PRELOADED) kb = *kb_in;
    else kb.load();
    const double log2e = kb.log2e, nln2hi = kb.nln2hi, nln2lo = kb.nln2lo;
    const double c3 = kb.c3, c4 = kb.c4, c5 = kb.c5, c6 = kb.c6, c7 = kb.c7, c8 = kb.c8, c9 = kb.c9, c10 = kb.c10, c11 = kb.c11,
                 c12 = kb.c12, c13 = kb.c13;
    PYLDA_LEVEL();
    double rd = __builtin_amdgcn_rcp(den);                      // 9
    double se = fma_scalar_mul(w2, b1, vb5);                    //    (B14 / 14 = B2 / 2 = 1/12)
    double so = fma_scalar_mul(w2, b6, vb4);
    PYLDA_LEVEL();
    double f = fma(-den, rd, 1.0);                              // 10
    se = fma_vvs(se, w2, b3);
    so = fma_vvs(so, w2, b2);
    PYLDA_LEVEL();
    rd = fma(rd, f, rd);                                        // 11
    se = fma_vvs(se, w2, b1);
    const double nt = num * tx;
    PYLDA_LEVEL();
    f = fma(-den, rd, 1.0);                                     // 12
    double ser = fma(so, w, se);
    PYLDA_LEVEL();
    rd = fma(rd, f, rd);                                        // 13
    ser = ser * w;
    const double vc10 = mov_s(c10);
    PYLDA_LEVEL();
    double tail = fma(inv, -0.5, -ser);                         // 14   psi(y) - log y
    const double vc11 = mov_s(c11);
    PYLDA_LEVEL();
    tail = tail - c;                                            // 15
    PYLDA_LEVEL();
    tail = fma(-nt, rd, tail);                                  // 16   psi(x) - log y - c
    PYLDA_LEVEL();
    // x below ~1e-50 (an alpha_k that small: pylda_set_alpha accepts any positive value) makes the exponent -1/x so
    // large that the reduced argument below is garbage (or -inf - -inf): exp of anything under -745 is 0 anyway
    tail = fmax(tail, -1100.0);                                 // 16'
    PYLDA_LEVEL();
    const double kx = tail * log2e;                             // 17
    PYLDA_LEVEL();
    const double kf = __builtin_rint(kx);                       // 18
    PYLDA_LEVEL();
    double r = fma_scalar_mul(kf, nln2hi, tail);                // 19
    const int ki = (int)kf;
    PYLDA_LEVEL();
    r = fma_scalar_mul(kf, nln2lo, r);                          // 20
    PYLDA_LEVEL();
    const double r2 = r * r;                                    // 21
    PYLDA_LEVEL();
    double pe = fma_scalar_mul(r2, c12, vc10);                  // 22
    double po = fma_scalar_mul(r2, c13, vc11);
    PYLDA_LEVEL();
    pe = fma_vvs(pe, r2, c8);                                   // 23
    po = fma_vvs(po, r2, c9);
    PYLDA_LEVEL();
    pe = fma_vvs(pe, r2, c6);                                   // 24
    po = fma_vvs(po, r2, c7);
    PYLDA_LEVEL();
    pe = fma_vvs(pe, r2, c4);                                   // 25
    po = fma_vvs(po, r2, c5);
    PYLDA_LEVEL();
    pe = fma(pe, r2, 0.5);                                      // 26
    po = fma_vvs(po, r2, c3);
    PYLDA_LEVEL();
    pe = fma(pe, r2, 1.0);                                      // 27
    po = fma(po, r2, 1.0);
    PYLDA_LEVEL();
    double res = fma(po, r, pe);                                // 28
    PYLDA_LEVEL();
    res = ldexp(res, ki);                                       // 29
    PYLDA_LEVEL();
    return y * res;                                             // 30
}

struct ExpDigammaLiterals {
    __device__ __forceinline__ double exp_of(double x) const { return exp_shallow(x); }
    static constexpr double nine = 9.0, ten = 10.0;
    static constexpr double d3 = 60.0, d2 = 1308.0, d1 = 12176.0, d0 = 40320.0;
    static constexpr double n4 = 5.0, n3 = 240.0, n2 = 3924.0, n1 = 24352.0;
    static constexpr double b1 = 1.0 / 12.0, b2 = -1.0 / 120.0, b3 = 1.0 / 252.0, b4 = -1.0 / 240.0, b5 = 1.0 / 132.0,
                            b6 = -691.0 / 32760.0;
};

__device__ __forceinline__ double exp_digamma_minus_levels(double x, double c)
{
    ExpDigammaLevelsA k;
    k.load();
    return exp_digamma_minus_levels(x, c, k);
}

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
