// lmdb_reader.hpp -- read-only walker of an LMDB environment file (data.mdb), enough for what the
// reference does with it: open, cursor MDB_FIRST / MDB_NEXT over the main database
// (src/caffe/layers/video_sampled_shots_data_layer.cpp:121-135,285-293,836-842).  There is no liblmdb in the
// target image, so the on-disk layout of LMDB 0.9 (meta pages 0/1, B+tree of branch / leaf pages, overflow
// pages for large values; 16-byte page header, 8-byte node header) is decoded directly.  Limits: the
// unnamed main DB only, no DUPSORT sub-databases, native little-endian 64-bit files, no concurrent writer.
// NOTE: validated against files produced by tests/lmdb_writer.py (written from the same format
// description); no liblmdb-produced file was available in this image to cross-check.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace caffe {

class LmdbReader {
 public:
  // path: the environment directory (containing data.mdb) or the data file itself
  bool Open(const std::string& path, std::string* err);
  size_t entries() const { return entries_; }
  // in-order traversal of every (key, value); value bytes are copied (they may span overflow pages)
  typedef std::function<void(const std::string&, const std::string&)> Fn;
  bool ForEach(const Fn& f, std::string* err) const { return root_ == ~0ull ? true : WalkImpl(root_, 0, f, err); }
 private:
  bool WalkImpl(uint64_t pgno, int depth, const Fn& f, std::string* err) const;
  const uint8_t* Page(uint64_t pgno) const { return data_.data() + pgno * psize_; }
  std::vector<uint8_t> data_;
  uint32_t psize_ = 0;
  uint64_t root_ = ~0ull, last_pg_ = 0;
  size_t entries_ = 0;
};

}  // namespace caffe
