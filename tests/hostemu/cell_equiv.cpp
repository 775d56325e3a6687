// The branch-free cell update of K2b / K3 (smg_logic.hpp cell_update) against the case tree the reference spells out
// (alignment.c:884-983), over every ordering of E, F, H, 0 and gi: all the update does is compare and subtract.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include "../../smalt_amd/csrc/smg_logic.hpp"
using namespace smg;
static int cell_old(int &Hj, int &E, int &F, int H, int gi, int ge, bool &cand) {
  int dir;
  cand = false;
  if (F > 0) {
    if (E > 0) {
      if (H > E) {
        if (H > F) {
          Hj = H; F -= ge; E -= ge; dir = DIR_DIA;
          if (H > gi) { cand = true; int t = H - gi; if (F < t) F = t; if (E < t) E = t; }
        } else { Hj = F; F -= ge; E -= ge; dir = DIR_ROW; }
      } else {
        if (E >= F) { Hj = E; dir = DIR_COL; } else { Hj = F; dir = DIR_ROW; }
        E -= ge; F -= ge;
      }
    } else {
      if (H > F) {
        Hj = H; F -= ge; dir = DIR_DIA;
        if (H > gi) { cand = true; E = H - gi; if (F < E) F = E; }
      } else { Hj = F; F -= ge; dir = DIR_ROW; }
    }
  } else if (E > 0) {
    if (H > E) {
      Hj = H; E -= ge; dir = DIR_DIA;
      if (H > gi) { cand = true; F = H - gi; if (E < F) E = F; }
    } else { Hj = E; E -= ge; dir = DIR_COL; }
  } else {
    if (H > 0) {
      Hj = H; dir = DIR_DIA;
      if (H > gi) { cand = true; F = E = H - gi; }
    } else { Hj = 0; dir = 0; }
  }
  return dir;
}
int main() {
  long n = 0;
  for (int gi = 0; gi <= 6; gi++) for (int ge = 0; ge <= 4; ge++)
  for (int E = -5; E <= 12; E++) for (int F = -5; F <= 12; F++) for (int H = -6; H <= 14; H++) for (int h0 = 0; h0 < 2; h0++) {
    int Hj1 = h0 * 7, Hj2 = h0 * 7, E1 = E, E2 = E, F1 = F, F2 = F; bool c1, c2;
    int d1 = cell_old(Hj1, E1, F1, H, gi, ge, c1), d2 = cell_update(Hj2, E2, F2, H, gi, ge, c2);
    if (d1 != d2 || Hj1 != Hj2 || E1 != E2 || F1 != F2 || c1 != c2) { printf("MISMATCH gi %d ge %d E %d F %d H %d: old d%d H%d E%d F%d c%d new d%d H%d E%d F%d c%d\n", gi, ge, E, F, H, d1, Hj1, E1, F1, c1, d2, Hj2, E2, F2, c2); return 1; }
    n++;
  }
  printf("ok %ld cases\n", n);
}
