/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Bit-compatible restatement of glibc's rand()/srand() (random_r TYPE_3,
 * x**31 + x**3 + 1 additive feedback generator).  The reference consumes the
 * libc stream in LCP::rand_min (src/LCP.cpp:199-209) and in lcp_lemke's
 * restart bases (src/LCP.cpp:618-621,639-642,686-688) and never calls srand
 * (SURVEY F7), so every process starts from seed 1.  The many-worlds build
 * gives each world its own copy of that stream.
 *
 * The state is a 31-word ring: word i of the glibc sequence r[] lives in slot
 * i % 31, which is exactly the slot r[i-31] occupied; `idx` is the slot the
 * next output overwrites.  This is the layout the HIP kernels use too
 * (moby_amd/csrc/mh_rand.h) so states can be copied between the two.
 */
#ifndef ORACLE_GLIBC_RAND_H
#define ORACLE_GLIBC_RAND_H
#include <stdint.h>

#define ORACLE_RAND_WORDS 32 /* 31 ring words + idx */

typedef struct { uint32_t r[31]; uint32_t idx; } oracle_rand_t;

static inline void oracle_srand(oracle_rand_t* s, uint32_t seed)
{
  int32_t r[34];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) {
    /* 16807 * r[i-1] % 2147483647 without overflow (Schrage), as glibc does */
    int64_t hi = r[i-1] / 127773, lo = r[i-1] % 127773;
    int64_t word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    r[i] = (int32_t)word;
  }
  /* slots 0..30 hold r[0..30]; glibc then sets r[31..33]=r[0..2] and discards
   * 310 outputs: run the recurrence for words 31..343. */
  for (int i = 0; i < 31; i++) s->r[i] = (uint32_t)r[i];
  s->idx = 0; /* word 31 goes to slot 31 % 31 = 0 */
  /* words 31,32,33 are copies r[i-31] (not sums) */
  /* slot (i%31) already contains r[i-31]; copying is a no-op, advance idx */
  s->idx = 3; /* next word is 34 -> slot 34 % 31 = 3 */
  for (int i = 34; i < 344; i++) {
    uint32_t a = s->r[s->idx];                  /* r[i-31] */
    uint32_t b = s->r[(s->idx + 28) % 31];      /* r[i-3]  */
    s->r[s->idx] = a + b;
    s->idx = (s->idx + 1) % 31;
  }
}

static inline int oracle_rand(oracle_rand_t* s)
{
  uint32_t a = s->r[s->idx];
  uint32_t b = s->r[(s->idx + 28) % 31];
  uint32_t v = a + b;
  s->r[s->idx] = v;
  s->idx = (s->idx + 1) % 31;
  return (int)(v >> 1);
}

#endif
