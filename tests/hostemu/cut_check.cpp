// tests/hostemu/cut_check.cpp -- TEST DRIVER: smgpost::cut_window (smalt_amd/csrc/smg_post.hpp) over vectors from stdin.
// Input lines: <hex string> <lo> <hi>; output lines: <code> [<hex out> <ref_first> <ref_last> <read_first> <read_last>].
#include <stdio.h>
#include <string>
#include "../../smalt_amd/csrc/smg_post.hpp"

int main() {
  char hex[8192];
  long lo, hi;
  smgpost::Piece pc;
  while (scanf("%8191s %ld %ld", hex, &lo, &hi) == 3) {
    std::vector<uint8_t> s;
    for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) { unsigned v; sscanf(hex + i, "%2x", &v); s.push_back((uint8_t)v); }
    s.push_back(0);
    const int rv = smgpost::cut_window(s.data(), lo, hi, pc);
    if (rv != smgpost::CUT_OK) { printf("%d\n", rv); continue; }
    printf("0 ");
    for (uint8_t b : pc.str) printf("%02x", b);
    printf(" %lld %lld %lld %lld\n", (long long)pc.ref_first, (long long)pc.ref_last, (long long)pc.read_first, (long long)pc.read_last);
  }
  return 0;
}
