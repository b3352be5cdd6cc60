/* Exhaustive check, on the host, of the three-instruction float division by the constant cp = 1004 that showalterIndex
 * uses on the device (mifc_pointwise.hip):   q = a * y;  r = fma(-cp, q, a);  q' = fma(r, y, q)   with y = RN(1 / cp).
 * Every finite float a whose quotient is a normal number or zero is compared with the IEEE division a / cp.
 *   gcc -O2 -mfma -fopenmp -o /tmp/verify_div_by_cp tools/verify_div_by_cp.c -lm && /tmp/verify_div_by_cp
 * Prints the number of mismatches (expected: 0) and the first few of them. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float from_bits(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(void)
{
  const float cp = 1004.0f;
  const float y = (float)(1.0 / 1004.0); /* the double quotient rounded to float: RN(1/cp) (1/cp is not near a float midpoint) */
  unsigned long long bad = 0, tested = 0;
#pragma omp parallel for reduction(+ : bad, tested) schedule(static)
  for (int64_t u = 0; u <= 0xffffffffLL; ++u) {
    const float a = from_bits((uint32_t)u);
    if (!isfinite(a))
      continue;
    const float want = a / cp;
    if (want != 0.0f && fabsf(want) < 1.17549435e-38f)
      continue; /* subnormal quotients: the kernel's range test keeps them away (a temperature times cp) */
    const float q = a * y;
    const float r = __builtin_fmaf(-cp, q, a);
    const float got = __builtin_fmaf(r, y, q);
    ++tested;
    if (memcmp(&got, &want, 4) != 0) {
      if (!(got == 0.0f && want == 0.0f)) { /* +0 / -0 */
        ++bad;
        if (bad < 5)
          printf("mismatch a=%a want=%a got=%a\n", a, want, got);
      }
    }
  }
  printf("y = %a; %llu floats tested, %llu mismatches\n", y, tested, bad);
  return bad != 0;
}
