// chain rounds: proven front extended by translation inside margins; passes per round, every value checked against the sequential chain.  gcc -O2 -ffp-contract=off -DFMT16=1 dc_margin_sim.c -lm; ./a.out <sigma, uc8 LSB> <offset, uc8 LSB> <first pass with margins, 99 = never> [b = with bursts]
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static double urand (void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
static double nrand (void) { return sqrt (-2 * log (urand () + 1e-300)) * cos (6.283185307179586 * urand ()); }
static uint32_t bits (float f) { uint32_t u; memcpy (&u, &f, 4); return u; }
static float fromb (uint32_t u) { float f; memcpy (&f, &u, 4); return f; }
#define NL 64
#define PER 16
static float B;
static float leakf (float z) { return z - z * B; }
int main (int argc, char **argv)
{
  const double rate = 2.4e6;
  B = (float) exp (-2 * M_PI / rate); const float A = 1.0f - B;
  const double sigma = argc > 1 ? atof (argv[1]) : 1.27, dc = argc > 2 ? atof (argv[2]) : 1.4;
  const int margins_from = argc > 3 ? atoi (argv[3]) : 1;      // pass index from which margins are formed (99: never)
  const int rounds = 3000;
  const long warm = 3000000;
  float z = (float) (dc / 127.5);
#define SAMPLE() (FMT16 ? (float) (int) lrint ((dc + sigma * nrand () + burst) * 257.0) * (1.0f / 32768.0f) : (float) ((fmin (255, fmax (0, lrint (127.5 + dc + sigma * nrand () + burst))) - 127.5) / 127.5))
  double burst = 0;
  for (long n = 0; n < warm; n++) { const float f = SAMPLE (); const float t = f * A; z = t + z * B; }
  long passes = 0, nrounds = 0, partial = 0, shifted = 0, wrong = 0;
  float t[NL * PER], zseq[NL * PER];
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < NL * PER; i++) { burst = (argc > 4 && (r % 7 == 0) && (i / 100) % 4 == 0) ? 40.0 * ((i & 1) ? 1 : -1) : 0; const float f = SAMPLE (); t[i] = f * A; }
    { float zz = z; for (int i = 0; i < NL * PER; i++) { zz = t[i] + zz * B; zseq[i] = zz; } }     // the truth
    int base = 0;
    while (base < NL * PER) {
      const int nl = (NL * PER - base) / PER;
      const float z0 = z;
      // the stair of z0: [elo, ehi) = the z with leakf (z) == leakf (z0), cut to z0's binade
      float elo, ehi;
      {
        const float L0 = leakf (z0);
        uint32_t u = bits (z0);
        const int neg = z0 < 0;
        // walk in ulps (a search by doubling then bisection on the monotone leakf)
        long lo = 0, hi = 1; while (leakf (fromb (u + (neg ? -hi : hi))) == L0 && hi < (1l << 22)) hi *= 2;          // towards larger z for z > 0 (bits grow), for z < 0 bits shrink
        while (hi - lo > 1) { long m = (lo + hi) / 2; if (leakf (fromb (u + (neg ? -m : m))) == L0) lo = m; else hi = m; }
        ehi = fromb (u + (neg ? -hi : hi));                      // first z above z0 with another leak
        lo = 0; hi = 1; while (leakf (fromb (u + (neg ? hi : -hi))) == L0 && hi < (1l << 22)) hi *= 2;
        while (hi - lo > 1) { long m = (lo + hi) / 2; if (leakf (fromb (u + (neg ? m : -m))) == L0) lo = m; else hi = m; }
        elo = fromb (u + (neg ? lo : -lo));                      // lowest z with z0's leak
        int ex; frexpf (z0, &ex); const float p_lo = ldexpf (0.5f, ex), p_hi = ldexpf (1.0f, ex);      // |z0| in [p_lo, p_hi)
        if (!neg) { if (elo < p_lo) elo = p_lo; if (ehi > p_hi) ehi = p_hi; } else { if (elo < -p_hi) elo = nextafterf (-p_hi, 0); if (ehi > -p_lo) ehi = nextafterf (-p_lo, 1); }
      }
      const float ulp = fabsf (nextafterf (z0, 2 * z0) - z0);
      float start[NL], end[NL], zt[NL][PER];
      { const float y0 = z0 * B; float acc = 0; for (int l = 0; l < nl; l++) { start[l] = z0 + acc; float sum = 0; for (int q = 0; q < PER; q++) sum += (t[base + l * PER + q] + y0) - z0; acc += sum; } }
      int good = 1, it;
      for (it = 0; it < 16; it++) {
        passes++;
        float marg[NL]; int tie[NL];
        for (int l = 0; l < nl; l++) {
          float zz = start[l]; float m = 1e30f; int ti = 0;
          // the start itself must lie in the stair too
          m = fminf (m, fminf (zz - elo + ulp, ehi - zz));
          for (int q = 0; q < PER; q++) {
            const float tt = t[base + l * PER + q];
            const float y = zz * B; zz = tt + y; zt[l][q] = zz;
            // margin: how far every z of the lane (start included) is from leaving [elo, ehi): shifts by less keep every leak and the binade
            if (q < PER - 1) m = fminf (m, fminf (zz - elo + ulp, ehi - zz));
            const float fr = tt / ulp - floorf (tt / ulp); if (fr == 0.5f) ti = 1;
          }
          end[l] = zz; marg[l] = m; tie[l] = ti;
        }
        good = nl;
        for (int l = 0; l < nl; l++) { const float below = l ? end[l - 1] : z0; if (bits (below) != bits (start[l])) { good = l; break; } }
        if (good == nl) break;
        float ns[NL]; { float acc = 0; for (int l = good; l < nl; l++) { ns[l] = end[good - 1] + acc; acc += end[l] - start[l]; } }
        // translation: lanes from `good` on whose error stays inside their margin are known without stepping them again
        int front = good;
        if (it >= margins_from) {
          while (front < nl) {
            const float e = ns[front] - start[front];
            // shift e moves every z of the lane by e: allowed while start + e and all z + e stay inside [elo, ehi): e < ehi - z (all z), e >= elo - z
            const int ok = !tie[front] && (e >= 0 ? e < marg[front] : -e < marg[front]);
            if (!ok) break;
            for (int q = 0; q < PER; q++) zt[front][q] += e;
            end[front] += e; start[front] = ns[front];
            front++; shifted++;
          }
        }
        if (front == nl || it == 15) { good = front; break; }
        for (int l = front; l < nl; l++) start[l] = ns[l];
        good = front;
      }
      if (good < nl) partial++;
      for (int l = 0; l < good; l++) for (int q = 0; q < PER; q++) if (bits (zt[l][q]) != bits (zseq[base + l * PER + q])) wrong++;
      z = end[good - 1];
      base += PER * good;
      nrounds++;
    }
  }
  printf ("fmt16 %d sigma %.2f dc %.2f margins from pass %d: %.3f passes per round, %ld rounds (%ld partial), %ld lanes shifted, %ld WRONG values\n", FMT16, sigma, dc, margins_from, (double) passes / nrounds, nrounds, partial, shifted, wrong);
  return 0;
}
