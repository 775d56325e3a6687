// tests/hostemu/dump_record.hpp -- TEST CODE: the calls of a `refdump -P` / `refdump -s` record (oracle/DUMPFORMAT.md) as the replay
// drivers need them: per pair (or split read) the mapSingleRead calls with their arguments, the alignments each call added and its
// counters.  Lines of other kinds (stage state, the set after a call) are skipped.
#ifndef DUMP_RECORD_HPP
#define DUMP_RECORD_HPP
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/smaltgpu.h"

struct RecCall {
  int mate, niv, fine, minscor;
  int prevmax[2];
  std::vector<smaltgpu_interval> ivs;
  std::vector<smaltgpu_result> res;
  std::vector<std::vector<uint8_t>> strs;
  int rx[7];                 // nnew swmax sw2nd nseg nseg_tot nhit nhit_tot
  int max1 = 0;              // best first-pass score (RC lines)
  bool used = false;
};
struct RecPair { std::vector<RecCall> calls; };

static std::vector<RecPair> load_dump(const char *path) {
  std::vector<RecPair> pairs;
  std::ifstream f(path);
  std::string ln;
  RecCall *call = nullptr;
  while (std::getline(f, ln)) {
    std::istringstream is(ln);
    std::string tag;
    is >> tag;
    if (tag == "PAIR") { pairs.emplace_back(); call = nullptr; }
    else if (tag == "MS") {
      pairs.back().calls.emplace_back();
      call = &pairs.back().calls.back();
      std::string kv;
      is >> kv;                                      // call number
      while (is >> kv) {
        const size_t e = kv.find('=');
        const std::string key = kv.substr(0, e), val = kv.substr(e + 1);
        if (key == "mate") call->mate = atoi(val.c_str());
        else if (key == "niv") call->niv = atoi(val.c_str());
        else if (key == "fine") call->fine = atoi(val.c_str());
        else if (key == "minscor") call->minscor = atoi(val.c_str());
        else if (key == "prevmax") sscanf(val.c_str(), "%d,%d", &call->prevmax[0], &call->prevmax[1]);
      }
    } else if (tag == "IV") { smaltgpu_interval v; long long sx; unsigned lo, hi; is >> sx >> lo >> hi; v.sidx = (int32_t)sx; v.lo = lo; v.hi = hi; call->ivs.push_back(v); }
    else if (tag == "RC") { std::vector<long long> v; long long x; while (is >> x) v.push_back(x); if (!v.empty() && v.back() > call->max1) call->max1 = (int)v.back(); }
    else if (tag == "RS") {
      unsigned idx; char strand; smaltgpu_result r; long long ss, se, sx; std::string hex;
      memset(&r, 0, sizeof(r));
      is >> idx >> strand >> r.swatscor >> r.q_start >> r.q_end >> ss >> se >> sx >> hex;
      r.s_start = (uint64_t)ss; r.s_end = (uint64_t)se; r.sidx = (int32_t)sx; r.reverse = (strand == 'R') ? 1u : 0u;
      std::vector<uint8_t> s;
      for (size_t i = 0; i + 1 < hex.size(); i += 2) s.push_back((uint8_t)strtoul(hex.substr(i, 2).c_str(), nullptr, 16));
      call->res.push_back(r); call->strs.push_back(s);
    } else if (tag == "RX") { for (int i = 0; i < 7; i++) is >> call->rx[i]; }
  }
  return pairs;
}

#endif
