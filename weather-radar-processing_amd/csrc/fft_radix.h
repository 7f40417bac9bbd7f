// fft_radix.h -- in-register radix-2/4/8/16 butterflies for gfx950 (wave64).
//
// Each function transforms R complex values held in one lane's VGPRs, natural
// order in, natural order out:  X[k] = sum_r x[r] * exp(SIGN * 2*pi*i * r*k / R).
// SIGN = -1 is the reference's FFTW_FORWARD / CUFFT_FORWARD (rpv2.cu:426),
// SIGN = +1 the conjugate transform that conj -> FFT -> conj amounts to
// (rpv2.cu:439,464,471).  All twiddles inside a butterfly are compile-time
// constants; the compiler folds them into v_fma/v_mul literals.
#pragma once
#include <hip/hip_runtime.h>

namespace wrp {

typedef float2 cf;

// (A packed form -- every complex add / scale / multiply as ONE v_pk_*_f32, rotations on op_sel / neg modifiers -- was
// built and measured in round 3: 28 % fewer vector instructions in the fused kernel, bit-identical output, the SAME speed
// (profiles/r03/ab_packed_math.log): a packed f32 instruction occupies the SIMD as long as the two it replaces.  It is in
// the history of this file.)
__device__ __forceinline__ cf cscale(cf a, float w) { return make_float2(a.x * w, a.y * w); }
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * w
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    return make_float2(fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y));
}
// a * (SIGN * i)
template <int SIGN>
__device__ __forceinline__ cf mul_si(cf a)
{
    return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// a * exp(SIGN * i*pi/4) = a * (1 + SIGN*i)/sqrt2
template <int SIGN>
__device__ __forceinline__ cf mul_w8_1(cf a)
{
    constexpr float h = 0.70710678118654752440f;
    return SIGN > 0 ? make_float2((a.x - a.y) * h, (a.x + a.y) * h)
                    : make_float2((a.x + a.y) * h, (a.y - a.x) * h);
}
// a * exp(SIGN * 3*i*pi/4) = a * (-1 + SIGN*i)/sqrt2
template <int SIGN>
__device__ __forceinline__ cf mul_w8_3(cf a)
{
    constexpr float h = 0.70710678118654752440f;
    return SIGN > 0 ? make_float2(-(a.x + a.y) * h, (a.x - a.y) * h)
                    : make_float2((a.y - a.x) * h, -(a.x + a.y) * h);
}

template <int SIGN>
__device__ __forceinline__ void fft2(cf &a, cf &b)
{
    cf t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// 4-point: X1 = (x0 - x2) + SIGN*i*(x1 - x3)
template <int SIGN>
__device__ __forceinline__ void fft4(cf &x0, cf &x1, cf &x2, cf &x3)
{
    cf t0 = cadd(x0, x2), t1 = csub(x0, x2);
    cf t2 = cadd(x1, x3), t3 = mul_si<SIGN>(csub(x1, x3));
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = cadd(t1, t3);
    x3 = csub(t1, t3);
}

// 8-point, decimation in time over (even, odd).  The two odd terms that carry exp(SIGN*i*pi/4) and exp(SIGN*3*i*pi/4)
// are never formed: with s = a.x -+ a.y their products only ever appear as e +- h*s, which is one fused multiply-add
// per output component (6 instructions per pair of outputs instead of 8; 52 per transform instead of 56).
template <int SIGN>
__device__ __forceinline__ void fft8(cf (&v)[8])
{
    constexpr float h = 0.70710678118654752440f;
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4<SIGN>(e0, e1, e2, e3);
    fft4<SIGN>(o0, o1, o2, o3);
    o2 = mul_si<SIGN>(o2);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    {   // o1 * (1 + SIGN*i) * h:  re = (o1.x - SIGN*o1.y) h, im = (SIGN*o1.x + o1.y) h
        const float re = SIGN > 0 ? o1.x - o1.y : o1.x + o1.y;
        const float im = SIGN > 0 ? o1.x + o1.y : o1.y - o1.x;
        v[1] = make_float2(fmaf(re, h, e1.x), fmaf(im, h, e1.y));
        v[5] = make_float2(fmaf(-re, h, e1.x), fmaf(-im, h, e1.y));
    }
    {   // o3 * (-1 + SIGN*i) * h:  re = -(o3.x + SIGN*o3.y) h, im = (SIGN*o3.x - o3.y) h
        const float re = SIGN > 0 ? o3.x + o3.y : o3.x - o3.y;   // negated below
        const float im = SIGN > 0 ? o3.x - o3.y : -(o3.x + o3.y);
        v[3] = make_float2(fmaf(-re, h, e3.x), fmaf(im, h, e3.y));
        v[7] = make_float2(fmaf(re, h, e3.x), fmaf(-im, h, e3.y));
    }
}

// 16-point as 4 x 4: F_r = fft4(x[r], x[r+4], x[r+8], x[r+12]);
// X[k' + 4k''] = fft4 over r of (W16^{r k'} F_r[k'])
// the first level of fft16 on inputs that still want their real weights: x_i = w_i a_i.  w0 a0 +- w2 a2 is one product and
// two fused multiply-adds per component instead of two products, an add and a subtract: 20 instructions instead of 24
template <int SIGN>
__device__ __forceinline__ void fft4_scaled(cf &x0, cf &x1, cf &x2, cf &x3, float w0, float w1, float w2, float w3)
{
    const cf s2 = cscale(x2, w2), s3 = cscale(x3, w3);
    const cf t0 = make_float2(fmaf(x0.x, w0, s2.x), fmaf(x0.y, w0, s2.y)), t1 = make_float2(fmaf(x0.x, w0, -s2.x), fmaf(x0.y, w0, -s2.y));
    const cf t2 = make_float2(fmaf(x1.x, w1, s3.x), fmaf(x1.y, w1, s3.y));
    const cf t3 = mul_si<SIGN>(make_float2(fmaf(x1.x, w1, -s3.x), fmaf(x1.y, w1, -s3.y)));
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = cadd(t1, t3);
    x3 = csub(t1, t3);
}
template <int SIGN> __device__ __forceinline__ void fft16_tail(cf (&v)[16]);
template <int SIGN>
__device__ __forceinline__ void fft16(cf (&v)[16])
{
#pragma unroll
    for (int r = 0; r < 4; r++) fft4<SIGN>(v[r], v[r + 4], v[r + 8], v[r + 12]);
    fft16_tail<SIGN>(v);
}
// fft16 of (w[r] s) * v[r] (the window of the range stage: row weight x column weight): the weights ride on the first
// level, each product w[r] s formed where it is used
template <int SIGN>
__device__ __forceinline__ void fft16_scaled(cf (&v)[16], const float (&w)[16], float s)
{
#pragma unroll
    for (int r = 0; r < 4; r++) fft4_scaled<SIGN>(v[r], v[r + 4], v[r + 8], v[r + 12], w[r] * s, w[r + 4] * s, w[r + 8] * s, w[r + 12] * s);
    fft16_tail<SIGN>(v);
}
template <int SIGN>
__device__ __forceinline__ void fft16_tail(cf (&v)[16])
{
    constexpr float c1 = 0.92387953251128675613f; // cos(pi/8)
    constexpr float s1 = 0.38268343236508977173f; // sin(pi/8)
    constexpr float sg = (float)SIGN;
    // now v[r + 4k'] = F_r[k'].  Apply W16^{r k'}:
    // k' = 1: r=1 -> W16^1, r=2 -> W16^2 = W8^1, r=3 -> W16^3
    v[5] = cmul(v[5], make_float2(c1, sg * s1));
    v[6] = mul_w8_1<SIGN>(v[6]);
    v[7] = cmul(v[7], make_float2(s1, sg * c1));
    // k' = 2: r=1 -> W8^1, r=2 -> W16^4 = SIGN*i, r=3 -> W16^6 = W8^3
    v[9] = mul_w8_1<SIGN>(v[9]);
    v[10] = mul_si<SIGN>(v[10]);
    v[11] = mul_w8_3<SIGN>(v[11]);
    // k' = 3: r=1 -> W16^3, r=2 -> W16^6 = W8^3, r=3 -> W16^9 = -W16^1
    v[13] = cmul(v[13], make_float2(s1, sg * c1));
    v[14] = mul_w8_3<SIGN>(v[14]);
    v[15] = cmul(v[15], make_float2(-c1, -sg * s1));
#pragma unroll
    for (int k = 0; k < 4; k++) fft4<SIGN>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    // now v[4k' + k''] = X[k' + 4k''] -> transpose to natural order
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = a + 1; b < 4; b++) {
            cf t = v[4 * a + b];
            v[4 * a + b] = v[4 * b + a];
            v[4 * b + a] = t;
        }
}


} // namespace wrp
