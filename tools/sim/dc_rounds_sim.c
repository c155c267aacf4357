// chain rounds on noise-only input (-DFMT16=1: sc16 samples, -DFMT16=0: uc8): passes per round for several refinement rules; gcc -O2 -ffp-contract=off -DFMT16=1 dc_rounds_sim.c -lm; ./a.out <sigma in uc8 LSB> <offset in uc8 LSB> <variant 0..2>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static double urand (void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
static double nrand (void) { return sqrt (-2 * log (urand () + 1e-300)) * cos (6.283185307179586 * urand ()); }
static uint32_t bits (float f) { uint32_t u; memcpy (&u, &f, 4); return u; }
#define NL 64
#define PER 16
int main (int argc, char **argv)
{
  const double rate = 2.4e6;
  const float B = (float) exp (-2 * M_PI / rate), A = 1.0f - B;
  const double sigma = argc > 1 ? atof (argv[1]) : 1.5, dc = argc > 2 ? atof (argv[2]) : -2.17;
  const int variant = argc > 3 ? atoi (argv[3]) : 0;
  const int rounds = 3000;
  const long warm = 3000000;
  float z = (float) (dc / 127.5);
  // settle sequentially
  for (long n = 0; n < warm; n++) { const float f = FMT16 ? (float) (int) lrint ((dc + sigma * nrand ()) * 257.0) * (1.0f / 32768.0f) : (float) ((fmin (255, fmax (0, lrint (127.5 + dc + sigma * nrand ()))) - 127.5) / 127.5); const float t = f * A; z = t + z * B; }
  long passes = 0, nrounds = 0, partial = 0;
  float t[NL * PER];
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < NL * PER; i++) { const float f = FMT16 ? (float) (int) lrint ((dc + sigma * nrand ()) * 257.0) * (1.0f / 32768.0f) : (float) ((fmin (255, fmax (0, lrint (127.5 + dc + sigma * nrand ()))) - 127.5) / 127.5); t[i] = f * A; }
    int base = 0;
    while (base < NL * PER) {
      const int nl = (NL * PER - base) / PER;
      const float z0 = z;
      float start[NL], end[NL], pstart[NL], pend[NL];
      // guess
      { const float y0 = z0 * B; float acc = 0; for (int l = 0; l < nl; l++) { start[l] = z0 + acc; float sum = 0; for (int q = 0; q < PER; q++) sum += (t[base + l * PER + q] + y0) - z0; acc += sum; } }
      int good = 1, it;
      int have_prev = 0;
      for (it = 0; it < 16; it++) {
        passes++;
        for (int l = 0; l < nl; l++) { float zz = start[l]; for (int q = 0; q < PER; q++) zz = t[base + l * PER + q] + zz * B; end[l] = zz; }
        good = nl;
        for (int l = 0; l < nl; l++) { const float below = l ? end[l - 1] : z0; if (bits (below) != bits (start[l])) { good = l; break; } }
        if (good == nl || it == 15) break;
        float ns[NL];
        if (variant == 0) {          // proven front + prefix of increments
          float acc = 0; for (int l = good; l < nl; l++) { ns[l] = end[good - 1] + acc; acc += end[l] - start[l]; }
        } else if (variant == 1) {   // the lane below's end, as it stands
          for (int l = good; l < nl; l++) ns[l] = end[l - 1];
        } else {                      // secant: slope of every lane from the last two passes (1 where unknown), composed upwards
          float cur = end[good - 1];
          for (int l = good; l < nl; l++) {
            ns[l] = cur;
            float slope = 1.0f;
            if (have_prev && bits (pstart[l]) != bits (start[l])) { slope = (end[l] - pend[l]) / (start[l] - pstart[l]); if (!(slope >= 0.0f)) slope = 0.0f; if (slope > 1.0f) slope = 1.0f; }
            cur = end[l] + slope * (cur - start[l]);
          }
        }
        memcpy (pstart, start, sizeof start); memcpy (pend, end, sizeof end); have_prev = 1;
        for (int l = good; l < nl; l++) start[l] = ns[l];
      }
      if (good < nl) partial++;
      z = end[good - 1];
      base += PER * good;
      nrounds++;
    }
  }
  printf ("sigma %.2f dc %.2f variant %d: %.3f passes per round, %ld rounds (%ld partial), z %.9g\n", sigma, dc, variant, (double) passes / nrounds, nrounds, partial, z);
  return 0;
}
