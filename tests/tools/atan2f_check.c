#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#define S2_DEV static inline
static inline int32_t __float_as_int(float f){int32_t i; memcpy(&i,&f,4); return i;}
S2_DEV float s2_atanf(float x)
{
	const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
	const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
	const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
						  9.0908870101e-02f,  -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
						  4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
	int32_t hx = __float_as_int(x);
	int32_t ix = hx & 0x7fffffff;
	int id;
	if (ix >= 0x4c000000) { if (ix > 0x7f800000) return x + x; return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3]; }
	if (ix < 0x3ee00000) { if (ix < 0x31000000) return x; id = -1; }
	else {
		x = fabsf(x);
		if (ix < 0x3f980000) { if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); } else { id = 1; x = (x - 1.0f) / (x + 1.0f); } }
		else { if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); } else { id = 3; x = -1.0f / x; } }
	}
	float z = x * x; float w = z * z;
	float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
	float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
	if (id < 0) return x - x * (s1 + s2);
	z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
	return (hx < 0) ? -z : z;
}
S2_DEV float s2_atan2f(float y, float x)
{
	const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
	if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
	if (hx == 0x3f800000) return s2_atanf(y);
	int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
	if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; case 3: return -pi - tiny; } }
	if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000) { if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; case 3: return -3.0f * pi_o_4 - tiny; } } else { switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; case 3: return -pi - tiny; } } }
	if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	int32_t k = (iy - ix) >> 23; float z;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo; else if (hx < 0 && k < -60) z = 0.0f; else z = s2_atanf(fabsf(y / x));
	switch (m) { case 0: return z; case 1: return -z; case 2: return pi - (z - pi_lo); default: return (z - pi_lo) - pi; }
}
int main(){
  uint64_t s=88172645463325252ull; long bad=0, n=0;
  for(long i=0;i<40000000;i++){
    s^=s<<13; s^=s>>7; s^=s<<17; uint32_t a=(uint32_t)s, b=(uint32_t)(s>>32);
    float y,x;
    if(i&1){ // unit-circle-like
      float ang = (float)(a)*(6.2831853f/4294967296.0f)-3.14159265f; y=sinf(ang); x=cosf(ang); if(i&2){y*=1.0000001f;}
    } else { memcpy(&y,&a,4); memcpy(&x,&b,4); }
    float r1=atan2f(y,x), r2=s2_atan2f(y,x);
    uint32_t u1,u2; memcpy(&u1,&r1,4); memcpy(&u2,&r2,4);
    if(u1!=u2 && !(r1!=r1 && r2!=r2)){ if(bad<10) printf("mismatch y=%a x=%a libm=%a mine=%a\n",y,x,r1,r2); bad++; }
    n++;
  }
  printf("n=%ld bad=%ld\n",n,bad);
}
