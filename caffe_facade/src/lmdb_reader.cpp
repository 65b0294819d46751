// lmdb_reader.cpp -- see lmdb_reader.hpp.  On-disk layout of LMDB 0.9 (mdb.c):
//   page header (16 B): pgno u64 | pad u16 | flags u16 | lower u16 | upper u16   (overflow: lower/upper = u32 page count)
//     flags: BRANCH 0x01, LEAF 0x02, OVERFLOW 0x04, META 0x08, LEAF2 0x20, SUBP 0x40
//   node pointers: u16 offsets from the page start, (lower - 16) / 2 of them
//   node (8 B header): lo u16 | hi u16 | flags u16 | ksize u16 | key bytes | data bytes
//     leaf:   data size = lo | hi << 16; flags BIGDATA 0x01 -> data is a u64 overflow page number,
//             SUBDATA 0x02 / DUPDATA 0x04 (sub-databases, not supported)
//     branch: child page = lo | hi << 16 | flags << 32
//   meta (at page + 16): magic u32 0xBEEFC0DE | version u32 | address u64 | mapsize u64 |
//     dbs[2] x {pad u32 | flags u16 | depth u16 | branch_pages u64 | leaf_pages u64 | overflow_pages u64 |
//               entries u64 | root u64} | last_pg u64 | txnid u64;   dbs[0].pad is the page size
#include "caffe/lmdb_reader.hpp"

#include <sys/stat.h>

#include <cstring>
#include <fstream>

namespace caffe {

namespace {
template <typename T> T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return v; }
constexpr int kPageHdr = 16;
constexpr uint32_t kMagic = 0xBEEFC0DE;
}

bool LmdbReader::Open(const std::string& path, std::string* err) {
  std::string file = path;
  struct stat st;
  if (stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) file = path + (path.back() == '/' ? "" : "/") + "data.mdb";
  std::ifstream f(file, std::ios::binary | std::ios::ate);
  if (!f) { *err = "mdb_env_open failed: cannot open " + file; return false; }
  const std::streamsize n = f.tellg();
  f.seekg(0);
  data_.resize((size_t)n);
  if (!f.read((char*)data_.data(), n)) { *err = "short read of " + file; return false; }
  if (n < 2 * 512) { *err = file + " is too small to be an LMDB file"; return false; }
  // page size from meta page 0, then pick the meta page with the larger txnid
  auto meta_ok = [&](const uint8_t* pg) { return (rd<uint16_t>(pg + 10) & 0x08) && rd<uint32_t>(pg + kPageHdr) == kMagic; };
  if (!meta_ok(data_.data())) { *err = file + ": bad LMDB magic"; return false; }
  if (rd<uint32_t>(data_.data() + kPageHdr + 4) != 1) { *err = file + ": unsupported LMDB data version"; return false; }
  psize_ = rd<uint32_t>(data_.data() + kPageHdr + 24);
  if (psize_ < 512 || (psize_ & (psize_ - 1)) || (size_t)n < 2ull * psize_) { *err = file + ": bad page size"; return false; }
  const uint8_t* best = nullptr; uint64_t best_txn = 0;
  for (int i = 0; i < 2; ++i) {
    const uint8_t* pg = Page(i);
    if (!meta_ok(pg)) continue;
    const uint8_t* m = pg + kPageHdr;
    const uint64_t txn = rd<uint64_t>(m + 24 + 2 * 48 + 8);
    if (!best || txn >= best_txn) { best = m; best_txn = txn; }
  }
  const uint8_t* main_db = best + 24 + 48;
  if (rd<uint16_t>(main_db + 4) & 0x04) { *err = file + ": DUPSORT databases are not supported"; return false; }
  entries_ = (size_t)rd<uint64_t>(main_db + 32);
  root_ = rd<uint64_t>(main_db + 40);
  last_pg_ = rd<uint64_t>(best + 24 + 2 * 48);
  if (root_ != ~0ull && root_ >= data_.size() / psize_) { *err = file + ": root page beyond the end of the file"; return false; }
  return true;
}

bool LmdbReader::WalkImpl(uint64_t pgno, int depth, const Fn& f, std::string* err) const {
  const uint64_t n_pages = data_.size() / psize_;           // page numbers are compared, never multiplied: no 64-bit wrap
  if (depth > 64 || pgno >= n_pages) { *err = "corrupt LMDB tree (page out of range)"; return false; }
  const uint8_t* pg = Page(pgno);
  const uint16_t flags = rd<uint16_t>(pg + 10);
  const uint32_t lower = rd<uint16_t>(pg + 12), upper = rd<uint16_t>(pg + 14);
  if (lower < kPageHdr || lower > upper || upper > psize_) { *err = "corrupt LMDB page header (lower / upper)"; return false; }
  const int nkeys = (int)(lower - kPageHdr) / 2;
  if (flags & 0x20) { *err = "LEAF2 pages (DUPFIXED) are not supported"; return false; }
  for (int i = 0; i < nkeys; ++i) {
    const uint16_t off = rd<uint16_t>(pg + kPageHdr + 2 * i);
    if (off < kPageHdr || off + 8u > psize_) { *err = "corrupt LMDB node offset"; return false; }
    const uint8_t* node = pg + off;
    const uint32_t lo = rd<uint16_t>(node), hi = rd<uint16_t>(node + 2);
    const uint16_t nflags = rd<uint16_t>(node + 4), ksize = rd<uint16_t>(node + 6);
    if (off + 8u + ksize + ((flags & 0x02) && (nflags & 0x01) ? 8u : 0u) > psize_) { *err = "corrupt LMDB node (key beyond the page)"; return false; }
    if (flags & 0x01) {                 // branch
      const uint64_t child = (uint64_t)lo | ((uint64_t)hi << 16) | ((uint64_t)nflags << 32);
      if (!WalkImpl(child, depth + 1, f, err)) return false;
    } else if (flags & 0x02) {          // leaf
      if (nflags & 0x06) { *err = "sub-databases / duplicates are not supported"; return false; }
      const uint32_t dsize = lo | (hi << 16);
      std::string key((const char*)node + 8, ksize), val;
      if (nflags & 0x01) {              // F_BIGDATA: value lives on overflow pages
        const uint64_t ov = rd<uint64_t>(node + 8 + ksize);
        if (ov >= n_pages || (uint64_t)kPageHdr + dsize > (n_pages - ov) * psize_) { *err = "overflow value beyond the end of the file"; return false; }
        val.assign((const char*)Page(ov) + kPageHdr, dsize);
      } else {
        if (off + 8u + ksize + dsize > psize_) { *err = "corrupt LMDB leaf node"; return false; }
        val.assign((const char*)node + 8 + ksize, dsize);
      }
      f(key, val);
    } else { *err = "unexpected page type in the tree"; return false; }
  }
  return true;
}

}  // namespace caffe
