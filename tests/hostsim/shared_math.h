/* shared_math.h -- TEST INFRASTRUCTURE ONLY: expf, logf, sinf, cosf, atan2f as fixed sequences of fp32 operations.
 *
 * Why: the device demodulator and the oracle agree bit for bit wherever they run the same arithmetic; they do not where one calls
 * glibc's libm and the other the GPU's hardware transcendentals or device libm.  To SHOW that what is left between them is rounding
 * inside those functions and the order of a few sums -- and nothing else -- both sides can be built against THIS file instead:
 * the oracle through orc_variant.shared_math, the device through the test-only build -DHFDL_DM_STRICT (tests/hostsim/serial_demod.h
 * compiled for gfx950).  Both builds compile it with floating-point contraction off, so every function below is the same chain of
 * IEEE-754 single-precision adds, multiplies and divides on an x86 core and on a gfx950 lane.
 *
 * The approximations are the classic Cephes single-precision kernels (Moshier; published coefficients): ~1 ulp, which is what the
 * loops they sit in (AGC gain, carrier NCO, PSK slicer) see from any libm.  Never included by the product build. */
#ifndef HFDL_SHARED_MATH_H
#define HFDL_SHARED_MATH_H
#include <stdint.h>

#ifndef SM_FN
#define SM_FN static inline
#endif

SM_FN uint32_t sm_bits(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
SM_FN float sm_float(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

/* x = m * 2^e, m in [0.5, 1) for normal positive x (the callers' arguments: an energy estimate above 1e-6) */
SM_FN float sm_frexp(float x, int *e)
{
	const uint32_t u = sm_bits(x);
	*e = (int)((u >> 23) & 0xFFu) - 126;
	return sm_float((u & 0x807FFFFFu) | 0x3F000000u);
}

/* x * 2^n for results that stay normal */
SM_FN float sm_ldexp(float x, int n)
{
	return x * sm_float((uint32_t)(n + 127) << 23);
}

SM_FN float sm_floor(float x)
{
	const float t = (float)(int)x;          /* |x| < 2^31 at every call site */
	return t > x ? t - 1.0f : t;
}

SM_FN float sm_logf(float x)
{
	int e;
	float m = sm_frexp(x, &e);
	if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else m = m - 1.0f;
	float z = m * m;
	float y = ((((((((7.0376836292E-2f * m - 1.1514610310E-1f) * m + 1.1676998740E-1f) * m - 1.2420140846E-1f) * m + 1.4249322787E-1f) * m
		- 1.6668057665E-1f) * m + 2.0000714765E-1f) * m - 2.4999993993E-1f) * m + 3.3333331174E-1f) * m * z;
	const float fe = (float)e;
	y += -2.12194440e-4f * fe;
	y += -0.5f * z;
	z = m + y;
	z += 0.693359375f * fe;
	return z;
}

SM_FN float sm_expf(float x)
{
	if (x > 88.0f) x = 88.0f;
	if (x < -87.0f) x = -87.0f;
	float z = sm_floor(1.44269504088896341f * x + 0.5f);
	x -= z * 0.693359375f;
	x -= z * -2.12194440e-4f;
	const int n = (int)z;
	z = x * x;
	z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
	return sm_ldexp(z, n);
}

/* sin and cos of |x| < 8192 (the carrier phase is kept within a step of [-pi, pi]) */
SM_FN void sm_sincosf(float xx, float *s, float *c)
{
	float x = xx < 0 ? -xx : xx;
	int ssign = xx < 0 ? -1 : 1, csign = 1;
	int j = (int)(1.27323954473516f * x);         /* 4 / pi */
	float y = (float)j;
	if (j & 1) { j += 1; y += 1.0f; }
	j &= 7;
	if (j > 3) { ssign = -ssign; csign = -csign; j -= 4; }
	if (j > 1) csign = -csign;
	x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
	const float z = x * x;
	const float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
	const float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z - 0.5f * z + 1.0f;
	const float sv = (j == 1 || j == 2) ? pc : ps, cv = (j == 1 || j == 2) ? ps : pc;
	*s = ssign < 0 ? -sv : sv;
	*c = csign < 0 ? -cv : cv;
}
SM_FN float sm_sinf(float x) { float s, c; sm_sincosf(x, &s, &c); return s; }
SM_FN float sm_cosf(float x) { float s, c; sm_sincosf(x, &s, &c); return c; }

SM_FN float sm_atanf(float xx)
{
	float x = xx < 0 ? -xx : xx, y;
	if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
	else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
	else y = 0.0f;
	const float z = x * x;
	y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
	return xx < 0 ? -y : y;
}

SM_FN float sm_atan2f(float y, float x)
{
	const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
	int code = 0;
	if (x < 0.0f) code = 2;
	if (y < 0.0f) code |= 1;
	if (x == 0.0f) {
		if (code & 1) return -PIO2F;
		if (y == 0.0f) return 0.0f;
		return PIO2F;
	}
	if (y == 0.0f) return (code & 2) ? PIF : 0.0f;
	const float w = code == 2 ? PIF : (code == 3 ? -PIF : 0.0f);
	return w + sm_atanf(y / x);
}

#endif
