// smg_indexbuild.h -- device-side construction of the index image (smg_indexbuild.hip)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace smg {

struct BuiltIndex {                 // device arrays (hipMalloc), owned by the caller
  int typ, nbits_key, nbits_lo;
  uint32_t nkeys, npos, nwords, maxpos;
  uint32_t *idx, *pos, *wordidx, *posidx, *packed;
  float build_ms;                   // device time of the whole construction (HIP events)
};

// selectHashTyp (smalt.c:268-332): 0 or -1 (unsupported)
int index_geometry(int k, int s, uint64_t totlen, int *typ, int *nbits_key, int *nbits_lo);

// d_ascii: the concatenated reference sequences in HBM (tot bytes, letters as in FASTA); h_sop: nseq + 1 offsets (host)
int build_index_device(const uint8_t *d_ascii, uint64_t tot, const uint64_t *h_sop, int nseq, int k, int s, BuiltIndex *out, char *err, size_t errlen);

}  // namespace smg
