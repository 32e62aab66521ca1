/* sph3d_atan2f.h — the ONE atan2f used by every bin-id computation in this repo.
 *
 * Why it exists: the reference's spherical-kernel binning
 * (tf_ops/buildkernel/tf_buildkernel_gpu.cu:55-56) calls the CUDA libdevice
 * atan2f.  glibc, CUDA libdevice and ROCm ocml each return slightly different
 * last bits, and a last-bit difference can move a neighbour across a bin
 * boundary.  Integer outputs must be bit-exact between the HIP kernels and the
 * CPU oracle, so both sides compile THIS function (plain IEEE-754 double
 * arithmetic: +, -, *, / only; no libm call; build with -ffp-contract=off).
 *
 * Algorithm: the published fdlibm argument reduction + degree-11 odd minimax
 * polynomial for atan in double precision (error < 1 ulp of double), wrapped
 * in the usual atan2 quadrant logic, then rounded once to float.  The result is
 * the correctly rounded atan2f except when the exact value lies within ~1e-16
 * relative of a float rounding boundary.
 *
 * Header is valid C99, C++ and HIP (host + device).
 */
#ifndef SPH3D_ATAN2F_H
#define SPH3D_ATAN2F_H

#if defined(__HIPCC__) || defined(__HIP__)
#define SPH3D_HD __host__ __device__
#else
#define SPH3D_HD
#endif

#ifdef __cplusplus
#define SPH3D_INLINE static inline
#else
#define SPH3D_INLINE static inline
#endif

/* atan(x) for finite x >= 0, double precision. */
SPH3D_HD SPH3D_INLINE double sph3d_atan_pos(double x)
{
    /* high/low parts of atan(0.5), atan(1), atan(1.5), atan(inf) */
    const double hi0 = 4.63647609000806093515e-01, lo0 = 2.26987774529616870924e-17;
    const double hi1 = 7.85398163397448278999e-01, lo1 = 3.06161699786838301793e-17;
    const double hi2 = 9.82793723247329054082e-01, lo2 = 1.39033110312309984516e-17;
    const double hi3 = 1.57079632679489655800e+00, lo3 = 6.12323399573676603587e-17;
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01;
    const double a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01;
    const double a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02;
    const double a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02;
    const double a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02;
    const double a10 = 1.62858201153657823623e-02;
    double hi = 0.0, lo = 0.0;
    int reduced = 1;
    if (x > 1.0e18) {
        return hi3 + lo3; /* atan(huge) = pi/2 */
    }
    if (x < 0.4375) {
        reduced = 0;
    } else if (x < 0.6875) {
        hi = hi0; lo = lo0; x = (2.0 * x - 1.0) / (2.0 + x);
    } else if (x < 1.1875) {
        hi = hi1; lo = lo1; x = (x - 1.0) / (x + 1.0);
    } else if (x < 2.4375) {
        hi = hi2; lo = lo2; x = (x - 1.5) / (1.0 + 1.5 * x);
    } else {
        hi = hi3; lo = lo3; x = -1.0 / x;
    }
    {
        double z = x * x;
        double w = z * z;
        double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
        double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
        if (!reduced) return x - x * (s1 + s2);
        return hi - ((x * (s1 + s2) - lo) - x);
    }
}

/* atan2f(y, x): correctly rounded (see header note).  NaN in -> NaN out. */
SPH3D_HD SPH3D_INLINE float sph3d_atan2f(float yf, float xf)
{
    const double pi = 3.14159265358979311600e+00, pi_lo = 1.22464679914735317720e-16;
    const double pio2 = 1.57079632679489655800e+00;
    const double pio4 = 7.85398163397448278999e-01;
    double y = (double)yf, x = (double)xf;
    double ay, ax, z;
    int xneg, yneg, xinf, yinf;
    if (xf != xf || yf != yf) return xf + yf; /* NaN */
    /* sign bits, honouring -0.0 (1/-0 = -inf) */
    xneg = (xf < 0.0f) || (xf == 0.0f && (1.0f / xf) < 0.0f);
    yneg = (yf < 0.0f) || (yf == 0.0f && (1.0f / yf) < 0.0f);
    ay = yneg ? -y : y;
    ax = xneg ? -x : x;
    if (ay == 0.0) {
        z = xneg ? pi : 0.0;            /* atan2(+-0, -x) = +-pi ; atan2(+-0, +x) = +-0 */
        return yneg ? (float)(-z) : (float)z;
    }
    if (ax == 0.0) {
        return yneg ? (float)(-pio2) : (float)pio2;
    }
    xinf = ax > 1.0e300;
    yinf = ay > 1.0e300;
    if (xinf || yinf) {
        if (xinf && yinf) z = xneg ? 3.0 * pio4 : pio4;
        else if (yinf) z = pio2;
        else z = xneg ? pi : 0.0;
        return yneg ? (float)(-z) : (float)z;
    }
    /* both finite, non-zero.  float inputs: |y/x| in [2^-277, 2^277], no overflow in double */
    z = sph3d_atan_pos(ay / ax);
    if (xneg) z = pi - (z - pi_lo);
    return yneg ? (float)(-z) : (float)z;
}

#endif /* SPH3D_ATAN2F_H */
