// smg_indexfile.hpp -- host-side reader of SMALT's index files into plain vectors:
//   <prefix>.sma  packed reference set   (written by seqSetWriteBinFil, sequence.c:2448-2519)
//   <prefix>.smi  k-mer hash index       (written by hashTableWrite,   hashidx.c:1214-1255)
// both inside the 12-word binary container of filio.c:48-68.  Same-endian files only.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace smg {

struct HostIndex {
  int32_t k = 0, s = 0, typ = 0, nbits_key = 0, nbits_lo = 0;
  uint32_t nkeys = 0, npos = 0, nwords = 0, maxpos = 0;
  std::vector<uint32_t> idx, pos, wordidx, posidx, packed;
  std::vector<uint64_t> sop;
  std::vector<std::string> names;
  int64_t nseq = 0;
  uint64_t totlen = 0;
};

namespace detail {
enum { FILIO_NHEAD = 12, FILIO_SIG = 0x73212173, FILIO_ENDIAN = 0x6E378A19, FILTYP_SEQSET = 1, FILTYP_HASHTAB = 2 };
inline bool read_container(FILE *fp, uint32_t typ_want, uint32_t *version, uint32_t *h, uint32_t nh, std::string &err) {
  uint32_t f[FILIO_NHEAD];
  if (fread(f, 4, FILIO_NHEAD, fp) != FILIO_NHEAD) { err = "short file header"; return false; }
  if (f[0] != (uint32_t)FILIO_SIG) { err = "not a SMALT binary file"; return false; }
  if (f[1] != (uint32_t)FILIO_ENDIAN) { err = "file written with different endianness"; return false; }
  if ((f[3] & 0xff) != typ_want || f[5] > nh) { err = "unexpected file type / header size"; return false; }
  *version = f[4];
  if (fread(h, 4, f[5], fp) != f[5]) { err = "short type-specific header"; return false; }
  return true;
}
}  // namespace detail

inline bool read_index_files(const std::string &prefix, HostIndex &ix, std::string &err) {
  uint32_t h[8] = {0}, ver = 0;
  FILE *fp = fopen((prefix + ".sma").c_str(), "rb");
  if (!fp) { err = "cannot open " + prefix + ".sma"; return false; }
  if (!detail::read_container(fp, detail::FILTYP_SEQSET, &ver, h, 8, err)) { fclose(fp); return false; }
  if (ver != 4) { fclose(fp); err = "unsupported .sma version"; return false; }     // sequence.c:79
  ix.nseq = (int64_t)(((uint64_t)h[1] << 32) + h[0]);
  const uint64_t namsiz = ((uint64_t)h[3] << 32) + h[2];
  ix.totlen = ((uint64_t)h[5] << 32) + h[4];
  std::vector<char> names(namsiz + 1, 0);
  if (fread(names.data(), 1, namsiz, fp) != namsiz) { fclose(fp); err = "short .sma (names)"; return false; }
  ix.names.clear();
  for (uint64_t o = 0; o < namsiz && (int64_t)ix.names.size() < ix.nseq;) { ix.names.emplace_back(names.data() + o); o += ix.names.back().size() + 1; }
  std::vector<uint32_t> seqlen((size_t)ix.nseq);
  if (fread(seqlen.data(), 4, (size_t)ix.nseq, fp) != (size_t)ix.nseq) { fclose(fp); err = "short .sma (lengths)"; return false; }
  ix.sop.assign((size_t)ix.nseq + 1, 0);
  for (int64_t i = 0; i < ix.nseq; i++) ix.sop[(size_t)i + 1] = ix.sop[(size_t)i] + seqlen[(size_t)i];
  const size_t nw = (size_t)(ix.totlen / 10 + 1);
  ix.packed.resize(nw);
  if (fread(ix.packed.data(), 4, nw, fp) != nw) { fclose(fp); err = "short .sma (sequence)"; return false; }
  fclose(fp);

  fp = fopen((prefix + ".smi").c_str(), "rb");
  if (!fp) { err = "cannot open " + prefix + ".smi"; return false; }
  if (!detail::read_container(fp, detail::FILTYP_HASHTAB, &ver, h, 8, err)) { fclose(fp); return false; }
  if (ver != 3) { fclose(fp); err = "unsupported .smi version"; return false; }     // hashidx.c:45
  ix.k = (int32_t)h[0]; ix.s = (int32_t)h[1]; ix.npos = h[2]; ix.maxpos = h[3]; ix.typ = (int32_t)h[4];
  ix.nbits_key = (int32_t)h[5]; ix.nbits_lo = (int32_t)h[6]; ix.nwords = h[7];
  if (ix.typ == 0) { ix.nbits_key = 2 * ix.k; ix.nbits_lo = 0; }
  if (ix.k < 1 || ix.k > 21 || ix.s < 1 || ix.nbits_key > 32 || ix.nbits_key < 1) { fclose(fp); err = "bad .smi header"; return false; }
  ix.nkeys = 1u << ix.nbits_key;
  ix.idx.resize((size_t)ix.nkeys + 1);
  ix.pos.resize((size_t)ix.npos + 1);
  if (fread(ix.idx.data(), 4, (size_t)ix.nkeys + 1, fp) != (size_t)ix.nkeys + 1 ||
      fread(ix.pos.data(), 4, ix.npos, fp) != ix.npos) { fclose(fp); err = "short .smi"; return false; }
  if (ix.typ != 0) {
    // hashTableRead (hashidx.c:1334) reads only 2*nwords+1 of the 2*(nwords+1) collision words, so
    // the reference runs with posidx[nwords] == 0.  A drop-in must see the same table.
    const size_t nr = 2 * (size_t)ix.nwords + 1;
    std::vector<uint32_t> w(2 * ((size_t)ix.nwords + 1), 0);
    if (fread(w.data(), 4, nr, fp) != nr) { fclose(fp); err = "short .smi (collision table)"; return false; }
    ix.wordidx.assign(w.begin(), w.begin() + ix.nwords + 1);
    ix.posidx.assign(w.begin() + ix.nwords + 1, w.end());
  }
  fclose(fp);
  return true;
}

}  // namespace smg
