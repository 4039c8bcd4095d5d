/* sincos_f32.h -- sine and cosine of a float angle as ONE fixed sequence of IEEE operations.
 *
 * The rotation-scale projection turns (scale, theta) into a covariance whose integer radius and
 * tile box are part of the operator's result.  Library sincosf implementations differ in their
 * last bits from one platform to the next (the device's and libm's do), and a last-bit change
 * can move a ceil(); so every build of this path -- the gfx950 kernels and the CPU checker --
 * includes this text and evaluates the same operations in the same order: no fused
 * multiply-add (both compile with -ffp-contract=off), no library call besides rint, which is
 * exact.
 *
 * Method: k = rint(theta * 2/pi) and the two-constant Cody-Waite reduction
 * r = theta - k*pio2_hi - k*pio2_lo in double precision (|r| <= pi/4 + rounding), then the
 * single-precision minimax polynomials of degree 7 (sine) and 8 (cosine) on r and the quadrant
 * fix-up.  Error: <= 1.5e-7 absolute against the exact functions for |theta| <= 1e5 (checked
 * by the CPU tests).  Angles beyond 1e9 rad, infinities and NaN give NaN for both results (the
 * projection then takes its early-out and writes zeros).
 */
#ifndef GI2D_SINCOS_F32_H
#define GI2D_SINCOS_F32_H

#ifndef GI2D_SINCOS_FN
#define GI2D_SINCOS_FN static inline
#endif

GI2D_SINCOS_FN void gi2d_sincos_f32(float theta, float *sin_out, float *cos_out)
{
    if (!(theta >= -1.0e9f && theta <= 1.0e9f)) { /* huge, infinite or NaN */
        *sin_out = *cos_out = __builtin_nanf("");
        return;
    }
    const double x = (double)theta;
    const double kd = rint(x * 0.63661977236758138);        /* 2/pi */
    double rd = x - kd * 1.5707963267341256;                 /* pi/2, upper 33 bits */
    rd = rd - kd * 6.0771005065061922e-11;                   /* pi/2 - the constant above */
    const float r = (float)rd;
    const int quadrant = (int)kd & 3; /* |kd| <= 6.4e8: fits */
    const float z = r * r;
    /* sin(r) = r + r*z*(S1 + z*(S2 + z*S3)) */
    float ps = -1.9515295891e-4f * z;
    ps = ps + 8.3321608736e-3f;
    ps = ps * z;
    ps = ps + -1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r;
    const float sr = ps + r;
    /* cos(r) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    float pc = 2.443315711809948e-5f * z;
    pc = pc + -1.388731625493765e-3f;
    pc = pc * z;
    pc = pc + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    const float half_z = 0.5f * z;
    pc = pc - half_z;
    const float cr = pc + 1.0f;
    float s = sr, c = cr;
    if (quadrant == 1) {
        s = cr;
        c = -sr;
    } else if (quadrant == 2) {
        s = -sr;
        c = -cr;
    } else if (quadrant == 3) {
        s = -cr;
        c = sr;
    }
    *sin_out = s;
    *cos_out = c;
}

#endif /* GI2D_SINCOS_F32_H */
