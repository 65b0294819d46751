// facade.cpp -- Caffe singleton, logging, Blob, the layer classes and the dataset behind `source:`.
#include <sys/time.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <sys/stat.h>
#include "caffe/lmdb_reader.hpp"

#include <fstream>
#include <random>

#include "caffe/layer.hpp"

namespace caffe {

// ------------------------------------------------------------------------------- logging -------
static FILE* g_log_file = nullptr;
void SetLogFile(const std::string& path) {
  if (g_log_file) fclose(g_log_file);
  g_log_file = path.empty() ? nullptr : fopen(path.c_str(), "a");
}
LogMessage::LogMessage(const char* file, int line, char sev) : sev_(sev) {
  struct timeval tv; gettimeofday(&tv, nullptr);
  struct tm tmv; localtime_r(&tv.tv_sec, &tmv);
  const char* base = strrchr(file, '/');
  char head[128];
  snprintf(head, sizeof(head), "%c%02d%02d %02d:%02d:%02d.%06ld %5d %s:%d] ", sev, tmv.tm_mon + 1, tmv.tm_mday,
           tmv.tm_hour, tmv.tm_min, tmv.tm_sec, (long)tv.tv_usec, (int)getpid(), base ? base + 1 : file, line);
  ss_ << head;
}
LogMessage::~LogMessage() {
  ss_ << "\n";
  const std::string s = ss_.str();
  fputs(s.c_str(), stderr);
  if (g_log_file) { fputs(s.c_str(), g_log_file); fflush(g_log_file); }
  if (sev_ == 'F') { fflush(stderr); abort(); }
}

// ------------------------------------------------------------------------------- Caffe ---------
// gflags of the reference's layer files, settable from the tools' command lines
int FLAGS_max_tries_for_negs = 100;      // video_sampled_shots_data_layer.cpp:20
int FLAGS_num_classes = 15;              // retrieval_stats_layer.cpp:16

static int env_int(const char* a, const char* b, int def) {
  const char* v = getenv(a);
  if (!v || !*v) v = getenv(b);
  return v && *v ? atoi(v) : def;
}
Caffe::Caffe() {
  world_ = env_int("VV_WORLD_SIZE", "WORLD_SIZE", 1);
  rank_ = env_int("VV_RANK", "RANK", 0);
  local_rank_ = env_int("VV_LOCAL_RANK", "LOCAL_RANK", rank_);
  if (world_ < 1 || rank_ < 0 || rank_ >= world_) { world_ = 1; rank_ = 0; local_rank_ = 0; }
  const char* j = getenv("VV_JOB_ID");
  if (!j || !*j) j = getenv("TORCHELASTIC_RUN_ID");
  if (!j || !*j) j = getenv("MASTER_PORT");
  job_id_ = j && *j ? j : "0";
  for (char& ch : job_id_) if (!isalnum((unsigned char)ch)) ch = '_';
}
Caffe& Caffe::Get() { static Caffe c; return c; }
void Caffe::set_mode(Brew mode) {
  CHECK(mode == GPU) << "This build has no CPU execution path (the reference's CPU path exists only "
                        "as the test oracle); use solver_mode: GPU";
  Get().mode_ = mode;
}
void Caffe::SetDevice(const int device_id) {
  Caffe& c = Get();
  if (c.ctx_ && c.device_ == device_id) return;
  CHECK(!c.ctx_) << "SetDevice after the context was created";
  c.device_ = device_id;
}
void Caffe::set_precision(const std::string& p) {
  CHECK(!Get().ctx_) << "set_precision after the context was created";
  CHECK(p == "f16" || p == "bf16") << "unknown precision " << p;
  Get().prec_ = p == "f16" ? VV_PREC_F16 : VV_PREC_BF16;
}
vv_ctx* Caffe::set_current_ctx(vv_ctx* c) { vv_ctx* old = Get().current_; Get().current_ = c; return old; }
vv_ctx* Caffe::ctx() {
  Caffe& c = Get();
  if (c.current_) return c.current_;
  if (!c.ctx_) {
    const char* env = getenv("VV_PREC");
    if (env) c.prec_ = !strcmp(env, "bf16") ? VV_PREC_BF16 : VV_PREC_F16;
    VV_CHECK(vv_create(c.device_, c.prec_, &c.ctx_));
  }
  return c.ctx_;
}
bool Caffe::has_ctx() { return Get().ctx_ != nullptr; }
void Caffe::register_ctx(vv_ctx* c) { Get().live_.push_back(c); }
void Caffe::unregister_ctx(vv_ctx* c) {
  vector<vv_ctx*>& l = Get().live_;
  l.erase(std::remove(l.begin(), l.end(), c), l.end());
}
bool Caffe::is_live(vv_ctx* c) {
  if (!c) return false;
  const Caffe& g = Get();
  return c == g.ctx_ || std::find(g.live_.begin(), g.live_.end(), c) != g.live_.end();
}
void Caffe::Reset() {
  Caffe& c = Get();
  if (c.ctx_) vv_destroy(c.ctx_);
  c.ctx_ = nullptr;
}

// ------------------------------------------------------------------------------- SyncedMemory --
// syncedmem.cpp:10-109 with HIP memory of the process context in place of cudaMalloc / cudaMemcpy
SyncedMemory::~SyncedMemory() {
  if (!dev_) return;
  if (Caffe::is_live(owner_)) vv_dev_free(owner_, dev_);
  else if (Caffe::has_ctx()) vv_dev_free(Caffe::ctx(), dev_);
}
// Transfers run on the stream of the context the device copy was allocated under (a TEST net has its own): reading a
// blob after Net::Forward returned -- outside any net's pass -- must still be ordered after that net's kernels.
vv_ctx* SyncedMemory::ctx() {
  if (dev_ && Caffe::is_live(owner_)) return owner_;
  return owner_ = Caffe::ctx();
}
void SyncedMemory::to_cpu() {
  switch (head_) {
    case UNINITIALIZED: host_.assign(size_, 0); head_ = HEAD_AT_CPU; break;
    case HEAD_AT_GPU:
      if (host_.size() != size_) host_.assign(size_, 0);
      VV_CHECK(vv_dev_download(ctx(), host_.data(), dev_, size_));
      head_ = SYNCED; break;
    case HEAD_AT_CPU: case SYNCED: break;
  }
}
void SyncedMemory::to_gpu() {
  switch (head_) {
    case UNINITIALIZED: VV_CHECK(vv_dev_alloc(ctx(), size_, &dev_)); head_ = HEAD_AT_GPU; break;   // zero-filled
    case HEAD_AT_CPU:
      if (!dev_) VV_CHECK(vv_dev_alloc(ctx(), size_, &dev_));
      VV_CHECK(vv_dev_upload(ctx(), dev_, host_.data(), size_));
      head_ = SYNCED; break;
    case HEAD_AT_GPU: case SYNCED: break;
  }
}
const void* SyncedMemory::cpu_data() { to_cpu(); return host_.data(); }
void* SyncedMemory::mutable_cpu_data() { to_cpu(); head_ = HEAD_AT_CPU; return host_.data(); }
const void* SyncedMemory::gpu_data() { to_gpu(); return dev_; }
void* SyncedMemory::mutable_gpu_data() { to_gpu(); head_ = HEAD_AT_GPU; return dev_; }

// ------------------------------------------------------------------------------- Blob ----------
template <typename Dtype>
void Blob<Dtype>::Reshape(const int num, const int channels, const int height, const int width) {
  CHECK_GE(num, 0); CHECK_GE(channels, 0); CHECK_GE(height, 0); CHECK_GE(width, 0);
  num_ = num; channels_ = channels; height_ = height; width_ = width;
  count_ = num * channels * height * width;
  if (!data_ || capacity_ < count_) {              // never shrinks (blob.cpp:20-24); memory is allocated on first touch
    capacity_ = count_;
    data_.reset(new SyncedMemory((size_t)count_ * sizeof(Dtype)));
    diff_.reset(new SyncedMemory((size_t)count_ * sizeof(Dtype)));
  }
}
template <typename Dtype>
void Blob<Dtype>::Update() {
  Dtype* d = mutable_cpu_data(); const Dtype* g = cpu_diff();
  for (int i = 0; i < count_; ++i) d[i] -= g[i];
}
template <typename Dtype>
Dtype Blob<Dtype>::asum_data() const { Dtype s = 0; for (int i = 0; i < count_; ++i) s += std::fabs(cpu_data()[i]); return s; }
template <typename Dtype>
Dtype Blob<Dtype>::asum_diff() const { Dtype s = 0; for (int i = 0; i < count_; ++i) s += std::fabs(cpu_diff()[i]); return s; }
template <typename Dtype>
void Blob<Dtype>::CopyFrom(const Blob<Dtype>& source, bool copy_diff, bool reshape) {
  if (num_ != source.num() || channels_ != source.channels() || height_ != source.height() || width_ != source.width()) {
    if (reshape) ReshapeLike(source);
    else LOG(FATAL) << "Trying to copy blobs of different sizes.";
  }
  if (copy_diff) memcpy(mutable_cpu_diff(), source.cpu_diff(), sizeof(Dtype) * count_);
  else memcpy(mutable_cpu_data(), source.cpu_data(), sizeof(Dtype) * count_);
}
template <typename Dtype>
void Blob<Dtype>::FromProto(const pl::Message& proto) {
  Reshape((int)proto.get_int("num"), (int)proto.get_int("channels"), (int)proto.get_int("height"), (int)proto.get_int("width"));
  const vector<float>& d = proto.floats("data");
  CHECK(d.empty() || (int)d.size() == count_) << "BlobProto data size " << d.size() << " vs count " << count_;
  for (size_t i = 0; i < d.size(); ++i) mutable_cpu_data()[i] = (Dtype)d[i];
  const vector<float>& g = proto.floats("diff");
  if (!g.empty()) { CHECK_EQ((int)g.size(), count_); for (size_t i = 0; i < g.size(); ++i) mutable_cpu_diff()[i] = (Dtype)g[i]; }
}
template <typename Dtype>
void Blob<Dtype>::ToProto(pl::Message* proto, bool write_diff) const {
  proto->set_int("num", num_); proto->set_int("channels", channels_);
  proto->set_int("height", height_); proto->set_int("width", width_);
  proto->clear("data"); proto->clear("diff");
  vector<float>* d = proto->mutable_floats("data");
  d->assign(cpu_data(), cpu_data() + count_);
  if (write_diff) proto->mutable_floats("diff")->assign(cpu_diff(), cpu_diff() + count_);
}
template class Blob<float>;

// ------------------------------------------------------------------------------- Layer ---------
template <typename Dtype>
Layer<Dtype>::Layer(const LayerParameter& param) : layer_param_(param) {
  const int n = layer_param_.size("blobs");
  blobs_.resize(n);
  for (int i = 0; i < n; ++i) { blobs_[i].reset(new Blob<Dtype>()); blobs_[i]->FromProto(layer_param_.get_msg("blobs", i)); }
}
template <typename Dtype>
void Layer<Dtype>::ToProto(LayerParameter* param, bool write_diff) {
  *param = layer_param_;
  param->clear("blobs");
  for (size_t i = 0; i < blobs_.size(); ++i) blobs_[i]->ToProto(param->add_msg("blobs"), write_diff);
}
template <typename Dtype>
void Layer<Dtype>::CheckBlobCounts(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const int nb = (int)bottom.size(), nt = (int)top.size();
  if (ExactNumBottomBlobs() >= 0) CHECK_EQ(ExactNumBottomBlobs(), nb) << type_name() << " Layer takes " << ExactNumBottomBlobs() << " bottom blob(s) as input.";
  if (MinBottomBlobs() >= 0) CHECK_LE(MinBottomBlobs(), nb) << type_name() << " Layer takes at least " << MinBottomBlobs() << " bottom blob(s) as input.";
  if (MaxBottomBlobs() >= 0) CHECK_GE(MaxBottomBlobs(), nb) << type_name() << " Layer takes at most " << MaxBottomBlobs() << " bottom blob(s) as input.";
  if (ExactNumTopBlobs() >= 0) CHECK_EQ(ExactNumTopBlobs(), nt) << type_name() << " Layer produces " << ExactNumTopBlobs() << " top blob(s) as output.";
  if (MinTopBlobs() >= 0) CHECK_LE(MinTopBlobs(), nt) << type_name() << " Layer produces at least " << MinTopBlobs() << " top blob(s) as output.";
  if (MaxTopBlobs() >= 0) CHECK_GE(MaxTopBlobs(), nt) << type_name() << " Layer produces at most " << MaxTopBlobs() << " top blob(s) as output.";
  if (EqualNumBottomTopBlobs()) CHECK_EQ(nb, nt) << type_name() << " Layer produces one top blob as output for each bottom blob input.";
}
template class Layer<float>;

// ------------------------------------------------------------------------------- dataset -------
static uint64_t mix64(uint64_t seed, uint64_t x) {     // == videovector_amd/synth.py:mix64
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + x;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// `source:` forms understood here (the reference opens an LMDB of VideoShots records,
// video_sampled_shots_data_layer.cpp:121-135):
//   <directory>  : LMDB environment (backend: LMDB) of VideoShots / TestVideoShotWindows records
//   synthetic://videos=2048;seed=1701;features=4096[;lo=16;span=49]
//   <file>.vvds : "VVDS1\0\0\0", int32 n_videos, int32 F, per video {int32 video_id, int32 n_shots,
//                 int32 shot_ids[n_shots]}, then float32 features[total_shots][F]
// ---- protobuf wire decoding of the DB records (generic pl::Message parsing would box every float) ----
namespace {
struct Wire {
  const uint8_t* p; const uint8_t* e; bool ok = true;
  Wire(const void* d, size_t n) : p((const uint8_t*)d), e((const uint8_t*)d + n) {}
  bool more() const { return ok && p < e; }
  uint64_t varint() {
    uint64_t v = 0; int sh = 0;
    while (p < e && sh < 64) { const uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; }
    ok = false; return 0;
  }
  Wire sub() { const uint64_t n = varint(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return Wire(p, 0); } Wire w(p, (size_t)n); p += n; return w; }
  void skip(int wt) {
    if (wt == 0) varint(); else if (wt == 1) { if (e - p < 8) ok = false; else p += 8; }
    else if (wt == 2) sub(); else if (wt == 5) { if (e - p < 4) ok = false; else p += 4; } else ok = false;
  }
};
// caffe.Datum (src/caffe/proto/caffe.proto:23-37): only float_data (field 6, repeated float, packed or not) matters
void ParseDatumFloats(Wire w, vector<float>* out, bool* ok) {
  while (w.more()) {
    const uint64_t tag = w.varint(); const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 6 && wt == 5) { if (w.e - w.p < 4) { w.ok = false; break; } float v; memcpy(&v, w.p, 4); w.p += 4; out->push_back(v); }
    else if (fn == 6 && wt == 2) { Wire s = w.sub(); const size_t n = (size_t)(s.e - s.p) / 4, o = out->size(); out->resize(o + n); memcpy(out->data() + o, s.p, 4 * n); }
    else w.skip(wt);
  }
  if (!w.ok) *ok = false;
}
void ParseInt32s(Wire& w, int wt, vector<int32_t>* out) {        // repeated int32, packed or not
  if (wt == 0) out->push_back((int32_t)w.varint());
  else if (wt == 2) { Wire s = w.sub(); while (s.more()) out->push_back((int32_t)s.varint()); if (!s.ok) w.ok = false; }
  else w.skip(wt);
}
}  // namespace

// VideoShots records (video_shot_sentences.proto:14-19), read in key order like the reference's cursor
// (video_sampled_shots_data_layer.cpp:121-135, 285-293): one video per record
shared_ptr<VideoDataset> VideoDataset::OpenLmdbVideoShots(const string& source) {
  shared_ptr<VideoDataset> ds(new VideoDataset());
  LmdbReader db; string err;
  CHECK(db.Open(source, &err)) << err;
  LOG(INFO) << "Opening lmdb " << source;
  bool ok = true;
  vector<float> feat;
  CHECK(db.ForEach([&](const string& key, const string& val) {
    Wire w(val.data(), val.size());
    int32_t vid = 0; vector<int32_t> ids; int n = 0;
    while (w.more()) {
      const uint64_t tag = w.varint(); const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
      if (fn == 1 && wt == 0) vid = (int32_t)w.varint();
      else if (fn == 2) ParseInt32s(w, wt, &ids);
      else if (fn == 3 && wt == 2) {
        feat.clear(); ParseDatumFloats(w.sub(), &feat, &ok);
        if (ds->F == 0) ds->F = (int)feat.size();
        CHECK_EQ((int)feat.size(), ds->F) << "record " << key << ": shot word with " << feat.size() << " features";
        ds->features.insert(ds->features.end(), feat.begin(), feat.end());
        ++n;
      } else w.skip(wt);
    }
    CHECK(w.ok && ok) << "record " << key << " is not a valid VideoShots message";
    CHECK_GE(n, 1) << "No shot word found: " << vid;
    CHECK_EQ((int)ids.size(), n) << "record " << key << ": shot_ids / shot_words size mismatch";
    ds->video_id.push_back(vid); ds->n_shots.push_back(n); ds->row_base.push_back(ds->n_rows);
    ds->shot_ids.insert(ds->shot_ids.end(), ids.begin(), ids.end());
    ds->n_rows += n;
  }, &err)) << err;
  CHECK_GE(ds->video_id.size(), 1u) << "empty database " << source;
  LOG(INFO) << "Read " << ds->video_id.size() << " videos, " << ds->n_rows << " frames, " << ds->F << " features";
  return ds;
}

// TestVideoShotWindows records (video_shot_sentences.proto:22-30; video_shot_window_test_data_layer.cpp:153-235):
// each record's words become consecutive table rows in the layer's channel order -- context words, then positive
// words, then negative words -- whatever order the fields were serialised in
shared_ptr<VideoDataset> VideoDataset::OpenLmdbTestWindows(const string& source) {
  shared_ptr<VideoDataset> ds(new VideoDataset());
  LmdbReader db; string err;
  CHECK(db.Open(source, &err)) << err;
  LOG(INFO) << "Opening lmdb " << source;
  bool ok = true;
  vector<float> feat, group[3];                 // 0 context (field 5), 1 positives (field 4), 2 negatives (field 6)
  bool first = true;
  CHECK(db.ForEach([&](const string& key, const string& val) {
    Wire w(val.data(), val.size());
    bool has_vid = false; int32_t vid = 0; int cnt[3] = {0, 0, 0}; int npos_ids = 0;
    for (auto& g : group) g.clear();
    while (w.more()) {
      const uint64_t tag = w.varint(); const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
      const int g = fn == 5 ? 0 : fn == 4 ? 1 : fn == 6 ? 2 : -1;
      if (fn == 1 && wt == 0) { vid = (int32_t)w.varint(); has_vid = true; }
      else if (g >= 0 && wt == 2) {
        feat.clear(); ParseDatumFloats(w.sub(), &feat, &ok);
        if (ds->F == 0) ds->F = (int)feat.size();
        CHECK_EQ((int)feat.size(), ds->F) << "record " << key << ": shot word with " << feat.size() << " features";
        group[g].insert(group[g].end(), feat.begin(), feat.end());
        ++cnt[g];
      } else if (fn == 2 && wt == 0) { ++npos_ids; w.varint(); }
      else if (fn == 2 && wt == 2) { Wire pk = w.sub(); while (pk.more()) { pk.varint(); ++npos_ids; } }   // packed
      else w.skip(wt);
    }
    CHECK(w.ok && ok) << "record " << key << " is not a valid TestVideoShotWindows message";
    CHECK(has_vid) << "No video id found for shot window";                          // …test_data_layer.cpp:188
    if (first) { ds->win_k = cnt[0]; ds->win_pos = cnt[1]; ds->win_neg = cnt[2]; first = false; }   // …:98-99, 112
    CHECK_EQ(cnt[0], ds->win_k);                                                    // …:193
    // the reference compares positives / negatives with the first record only when they are included (:190-201);
    // the rows are stored either way, so every record must agree
    CHECK_EQ(cnt[1], ds->win_pos) << "record " << key << ": positive_shot_words";
    CHECK_EQ(cnt[2], ds->win_neg) << "record " << key << ": negative_shot_words";
    ds->win_pos_ids.push_back(npos_ids);
    for (int g = 0; g < 3; ++g) {
      ds->features.insert(ds->features.end(), group[g].begin(), group[g].end());
      for (int j = 0; j < cnt[g]; ++j) ds->win_rows.push_back((int32_t)ds->n_rows++);
    }
    ds->win_video_id.push_back(vid);
  }, &err)) << err;
  CHECK_GE(ds->win_video_id.size(), 1u) << "empty database " << source;
  LOG(INFO) << "Read " << ds->win_video_id.size() << " test windows of " << ds->win_k << " context frames, " << ds->win_pos
            << " positives, " << ds->win_neg << " negatives, " << ds->F << " features";
  return ds;
}

shared_ptr<VideoDataset> VideoDataset::Open(const string& source, Kind kind, const string& backend) {
  shared_ptr<VideoDataset> ds(new VideoDataset());
  const string syn = "synthetic://";
  if (source.compare(0, syn.size(), syn) == 0) {
    long videos = 2048, seed = 1701, feat = 4096, lo = 16, span = 49;
    string rest = source.substr(syn.size());
    size_t p = 0;
    while (p < rest.size()) {
      size_t e = rest.find(';', p); if (e == string::npos) e = rest.size();
      const string kv = rest.substr(p, e - p);
      const size_t eq = kv.find('=');
      CHECK(eq != string::npos) << "bad synthetic source option '" << kv << "'";
      const string k = kv.substr(0, eq); const long v = atol(kv.c_str() + eq + 1);
      if (k == "videos") videos = v; else if (k == "seed") seed = v; else if (k == "features") feat = v;
      else if (k == "lo") lo = v; else if (k == "span") span = v;
      else LOG(FATAL) << "unknown synthetic source option " << k;
      p = e + 1;
    }
    ds->synthetic = true; ds->seed = (uint64_t)seed; ds->F = (int)feat;
    for (long v = 0; v < videos; ++v) {
      const int n = (int)(lo + mix64(ds->seed, (uint64_t)v) % (uint64_t)span);
      ds->video_id.push_back((int32_t)v); ds->n_shots.push_back(n); ds->row_base.push_back(ds->n_rows);
      for (int j = 0; j < n; ++j) ds->shot_ids.push_back(j);
      ds->n_rows += n;
    }
    LOG(INFO) << "Opening synthetic dataset: " << videos << " videos, " << ds->n_rows << " frames, " << feat << " features";
    return ds;
  }
  const string synw = "synthetic-windows://";
  if (source.compare(0, synw.size(), synw) == 0) {
    // the synthetic videos above plus `windows` test records of `context` consecutive frames each:
    // window w -> video mix64(wseed, w) % videos, first frame mix64(wseed, 2^32 + w) % (n - context + 1)
    long windows = 673, context = 4, wseed = 7;
    string base = "synthetic://", rest = source.substr(synw.size());
    size_t p = 0;
    while (p < rest.size()) {
      size_t e = rest.find(';', p); if (e == string::npos) e = rest.size();
      const string kv = rest.substr(p, e - p);
      const size_t eq = kv.find('=');
      CHECK(eq != string::npos) << "bad source option '" << kv << "'";
      const string k = kv.substr(0, eq); const long v = atol(kv.c_str() + eq + 1);
      if (k == "windows") windows = v; else if (k == "context") context = v; else if (k == "wseed") wseed = v;
      else base += kv + ";";
      p = e + 1;
    }
    if (base.back() == ';') base.pop_back();
    ds = Open(base, kShots, backend);
    ds->win_k = (int)context;
    for (long w = 0; w < windows; ++w) {
      const int v = (int)(mix64((uint64_t)wseed, (uint64_t)w) % ds->video_id.size());
      CHECK_GE(ds->n_shots[v], (int)context);
      const int64_t st = (int64_t)(mix64((uint64_t)wseed, (1ull << 32) + (uint64_t)w) % (uint64_t)(ds->n_shots[v] - context + 1));
      ds->win_video_id.push_back(ds->video_id[v]);
      for (long j = 0; j < context; ++j) ds->win_rows.push_back((int32_t)(ds->row_base[v] + st + j));
    }
    return ds;
  }
  struct stat st;
  if (stat(source.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) {
    CHECK(backend == "LMDB") << "source " << source << " is a database directory with backend " << backend
                             << ": only LMDB databases are readable (LevelDB is not built)";
    return kind == kTestWindows ? OpenLmdbTestWindows(source) : OpenLmdbVideoShots(source);
  }
  std::ifstream f(source, std::ios::binary);
  CHECK(f.good()) << "Failed to open dataset " << source;
  char magic[8]; f.read(magic, 8);
  CHECK(!memcmp(magic, "VVDS1\0\0\0", 8)) << source << " is not a .vvds dataset";
  int32_t nv = 0, F = 0; f.read((char*)&nv, 4); f.read((char*)&F, 4);
  CHECK_GE(nv, 1); CHECK_GE(F, 1);
  ds->F = F;
  for (int v = 0; v < nv; ++v) {
    int32_t vid, n; f.read((char*)&vid, 4); f.read((char*)&n, 4);
    CHECK_GE(n, 1) << "No shot word found: " << vid;
    ds->video_id.push_back(vid); ds->n_shots.push_back(n); ds->row_base.push_back(ds->n_rows);
    const size_t o = ds->shot_ids.size(); ds->shot_ids.resize(o + n);
    f.read((char*)&ds->shot_ids[o], 4 * n);
    ds->n_rows += n;
  }
  ds->features.resize((size_t)ds->n_rows * F);
  f.read((char*)ds->features.data(), sizeof(float) * ds->features.size());
  CHECK(f.good()) << "truncated dataset " << source;
  LOG(INFO) << "Opening dataset " << source << ": " << nv << " videos, " << ds->n_rows << " frames, " << F << " features";
  return ds;
}
// negative_dataset: its feature rows follow this dataset's in the one table the context holds
void VideoDataset::AppendNegatives(const VideoDataset& neg) {
  CHECK(!synthetic && !neg.synthetic) << "negative_dataset needs stored (not synthetic://) sources";
  CHECK_EQ(F, neg.F) << "negative_dataset has a different feature size";
  neg_video_id = neg.video_id; neg_n_shots = neg.n_shots; neg_shot_ids = neg.shot_ids;
  for (int64_t rb : neg.row_base) neg_row_base.push_back(n_rows + rb);
  features.insert(features.end(), neg.features.begin(), neg.features.end());
  n_rows += neg.n_rows;
}
void VideoDataset::UploadTable(vv_ctx* ctx) const {
  if (synthetic) VV_CHECK(vv_table_synth(ctx, seed, n_rows, F));
  else VV_CHECK(vv_table_set(ctx, features.data(), n_rows, F));
}

// ------------------------------------------------------------------------------- data layer ----
template <typename Dtype>
VideoSampledShotsDataLayer<Dtype>::~VideoSampledShotsDataLayer() {
  if (ring_ && !sampler_) vv_batch_ring_detach(ring_);
  if (sampler_) vv_sampler_destroy(sampler_);        // stops the prefetch threads
}
// BasePrefetchingDataLayer (base_data_layer.cpp:52-95) starts a thread per batch and joins it in Forward.  Here the
// sampler's own prefetch threads (vv_sampler_prefetch_start) run kPrefetchDepth batches ahead for the whole life of the
// layer; "create" is a no-op after the first call and "join" is the wait inside vv_sampler_next.
template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::CreatePrefetchThread() {
  if (prefetching_) return;
  const int threads = getenv("VV_SAMPLER_THREADS") ? atoi(getenv("VV_SAMPLER_THREADS")) : 4;
  const int world = Caffe::world();
  // data-parallel: ONE sampler per node (rank 0) draws the global batch and publishes it in shared memory; every rank
  // takes its batch_size items of it
  const string ring_name = "vv_caffe_" + Caffe::job_id() + "_" + this->layer_param_.get_str("name");
  if (sampler_) {
    const bool shared = world > 1 && !per_rank_;
    CHECK_EQ(vv_sampler_prefetch_start(sampler_, kPrefetchDepth, threads < 1 ? 1 : threads, shared ? ring_name.c_str() : nullptr, shared ? world : 1), 0);
    CHECK_EQ(vv_sampler_ring(sampler_, &ring_), 0);
  } else {
    CHECK_EQ(vv_batch_ring_attach(ring_name.c_str(), 300.0, &ring_), 0) << "rank " << Caffe::rank() << ": no batch ring " << ring_name
                                                                       << " from rank 0";
  }
  prefetching_ = true;
}
template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::JoinPrefetchThread() {}

template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>* top) {
  const pl::Message& p = this->layer_param_.get_msg("video_sampled_shots_data_param");
  const string ctype = p.get_enum("context_type");
  // rand_skip (…data_layer.cpp:156-180): skip = caffe_rng_rand() % rand_skip records.  caffe_rng_rand is the first draw of
  // the process-wide mt19937 after Caffe::set_random_seed (the data layer is set up first), and boost::mt19937 is the
  // standard generator; without `random_seed` in the solver the reference seeds from /dev/urandom, here the default seed is used.
  if (p.get_int("rand_skip") > 0) {
    std::mt19937 gen(Caffe::random_seed());
    const unsigned skip = (unsigned)gen() % (unsigned)p.get_int("rand_skip");
    LOG(INFO) << "Skipping first " << skip << " data points.";
    rand_skip_ = (int)skip;
  }
  dataset_ = VideoDataset::Open(p.get_str("source"), VideoDataset::kShots, p.get_enum("backend"));
  if (!p.get_str("negative_dataset").empty()) {                    // …data_layer.cpp:105-151: a second DB of the same backend
    shared_ptr<VideoDataset> neg = VideoDataset::Open(p.get_str("negative_dataset"), VideoDataset::kShots, p.get_enum("backend"));
    dataset_->AppendNegatives(*neg);
  }
  vv_sampler_param sp;
  vv_sampler_param_default(&sp);
  sp.batch_size = batch_size_ = (int)p.get_int("batch_size");
  sp.context_size = context_size_ = (int)p.get_int("context_size");
  sp.num_negative_samples = num_negative_samples_ = (int)p.get_int("num_negative_samples");
  sp.max_buffer_size = (int)p.get_int("max_buffer_size");
  sp.negative_swap_percentage = (int)p.get_int("negative_swap_percentage");
  sp.max_same_video_negs = (int)p.get_int("max_same_video_negs");
  sp.max_tries_for_negs = FLAGS_max_tries_for_negs;                                       // …data_layer.cpp:20, 245
  sp.initial_cursor = rand_skip_;
  feature_size_ = dataset_->F;
  CHECK_GE(feature_size_, 1); CHECK_GE(context_size_, 2); CHECK_GE(batch_size_, 1);      // …data_layer.cpp:206-209
  if (ctype == "WINDOW") { CHECK(context_size_ % 2 == 1) << "Context size should be even in this setting!"; }   // …:434 (sic)
  sp.context_type = ctype == "WINDOW" ? VV_CONTEXT_WINDOW : ctype == "PAST" ? VV_CONTEXT_PAST :
                    ctype == "PAST_CONTINUOUS" ? VV_CONTEXT_PAST_CONTINUOUS :
                    ctype == "PAIRWISE" ? VV_CONTEXT_PAIRWISE : VV_CONTEXT_PAST_CONTINUOUS_FIXED;
  if (ctype == "PAIRWISE") sp.context_size = context_size_ = 2;                           // …data_layer.cpp:200-201
  sp.output_shot_distance = p.get_bool("output_shot_distance");                           // …:71 (read by PAIRWISE only, :407)
  sp.max_shot_distance = (float)p.get_num("max_shot_distance");
  CHECK_LE(sp.max_same_video_negs, sp.num_negative_samples)
      << "max_same_video_negs exceeds num_negative_samples: the reference writes past the item's channels (…data_layer.cpp:484-502)";
  // Data-parallel jobs.  Default: every rank runs this layer as the reference wrote it, for its own batch_size
  // items, with its own draw stream (srand(1 + rank); rank 0 keeps the reference's never-seeded stream) and its own starting
  // record (rand_skip + rank * records / world): N independent reference samplers, no serial section shared by the ranks.
  // VV_SAMPLER_MODE=node: ONE logical sampler (rank 0) draws the global batch of world * batch_size items exactly as a
  // single process with that batch size would and every rank takes its slice (bit-identical to the
  // one-process run, bound by the sampler's serial walk: ~0.17 ms per 1024 items).
  if (Caffe::world() > 1) {
    const char* sm = getenv("VV_SAMPLER_MODE");
    per_rank_ = !(sm && string(sm) == "node");
    CHECK(!sm || string(sm) == "node" || string(sm) == "rank") << "VV_SAMPLER_MODE: node or rank";
    if (per_rank_) {
      const int nrec = (int)dataset_->video_id.size();
      sp.rand_seed = 1 + Caffe::rank();
      sp.initial_cursor = (int)(((int64_t)rand_skip_ + (int64_t)Caffe::rank() * nrec / Caffe::world()) % nrec);
      LOG(INFO) << "Data-parallel rank " << Caffe::rank() << ": own sampler, srand(" << sp.rand_seed << "), first record " << sp.initial_cursor;
    } else {
      CHECK_LE(sp.max_same_video_negs, 0) << "VV_SAMPLER_MODE=node needs max_same_video_negs: 0";
      sp.batch_size = batch_size_ * Caffe::world();    // the prototxt's batch_size is per GPU; the sampler draws the global batch
      LOG(INFO) << "Data-parallel rank " << Caffe::rank() << ": one logical sampler of the global batch on rank 0";
    }
  }
  if (Caffe::rank() == 0 || per_rank_) {
    const int rc = vv_sampler_create_neg(&sp, (int)dataset_->video_id.size(), dataset_->video_id.data(), dataset_->n_shots.data(),
                                         dataset_->row_base.data(), dataset_->shot_ids.data(), (int)dataset_->neg_video_id.size(),
                                         dataset_->neg_video_id.data(), dataset_->neg_n_shots.data(), dataset_->neg_row_base.data(),
                                         dataset_->neg_shot_ids.data(), &sampler_);
    CHECK_EQ(rc, 0) << "Could not add requested number of negatives";                     // …:344
  }
  (*top)[0]->Reshape(batch_size_, context_size_ + num_negative_samples_, feature_size_, 1);  // …:214-218
  LOG(INFO) << "output data size: " << (*top)[0]->num() << "," << (*top)[0]->channels() << "," << (*top)[0]->height()
            << "," << (*top)[0]->width();
  if (top->size() > 1) (*top)[1]->Reshape(batch_size_, 1, 1, 1);
  CreatePrefetchThread();                                          // base_data_layer.cpp:63-66
}
template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::NextBatch(vector<int32_t>* idx, vector<int32_t>* last_src, vector<int32_t>* label) {
  JoinPrefetchThread();                                            // base_data_layer.cpp:81-95: join, hand over, respawn
  const size_t n = (size_t)batch_size_ * (context_size_ + num_negative_samples_);
  idx->resize(n); last_src->resize(n); label->resize(batch_size_);
  if (Caffe::world() == 1 || per_rank_) {
    CHECK_EQ(vv_sampler_next(sampler_, idx->data(), last_src->data(), label->data()), 0);
  } else {                                                         // this rank's items of the global batch
    CHECK_EQ(vv_batch_ring_next(ring_, Caffe::rank(), Caffe::rank() * batch_size_, batch_size_, idx->data(), label->data(), 600.0), 0)
        << "no batch from rank 0's sampler";
    *last_src = *idx;
  }
  CreatePrefetchThread();
}
template class VideoSampledShotsDataLayer<float>;

template <typename Dtype>
void VideoShotWindowTestDataLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>* top) {
  const pl::Message& p = this->layer_param_.get_msg("video_shot_window_test_data_param");
  dataset_ = VideoDataset::Open(p.get_str("source"), VideoDataset::kTestWindows, p.get_enum("backend"));
  CHECK_GE(dataset_->win_k, 1) << "source " << p.get_str("source") << " holds no test windows";   // …test_data_layer.cpp:113
  // …:98-106: the first record sets the sizes, include_positives / include_negatives: false drop the group
  positive_size_ = p.get_bool("include_positives") ? dataset_->win_pos : 0;
  negative_size_ = p.get_bool("include_negatives") ? dataset_->win_neg : 0;
  LOG(INFO) << "Pos-size: " << positive_size_ << "Neg-size: " << negative_size_;
  if (positive_size_ > 0)
    for (size_t w = 0; w < dataset_->win_pos_ids.size(); ++w)
      CHECK_EQ(dataset_->win_pos_ids[w], positive_size_) << "positive_shot_id count of test window " << w;   // …:191
  batch_size_ = (int)p.get_int("batch_size");
  CHECK_GE(batch_size_, 1);
  (*top)[0]->Reshape(batch_size_, channels(), dataset_->F, 1);                                    // …:117-121
  LOG(INFO) << "output data size: " << (*top)[0]->num() << "," << (*top)[0]->channels() << "," << (*top)[0]->height()
            << "," << (*top)[0]->width();
  if (top->size() > 1) (*top)[1]->Reshape(batch_size_, 1, 1, 1);
}
template <typename Dtype>
void VideoShotWindowTestDataLayer<Dtype>::NextBatch(vector<int32_t>* rows, vector<int32_t>* video_ids) {
  const int k = dataset_->win_k, stride = k + dataset_->win_pos + dataset_->win_neg, ch = channels();
  const size_t nw = dataset_->win_video_id.size();
  rows->resize((size_t)batch_size_ * ch); video_ids->resize(batch_size_);
  for (int i = 0; i < batch_size_; ++i) {                          // cursor wraps (…:250-262)
    const int32_t* src = &dataset_->win_rows[cursor_ * stride];
    int32_t* dst = &(*rows)[(size_t)i * ch];
    for (int j = 0; j < k; ++j) *dst++ = src[j];                                   // …:207-214 context
    for (int j = 0; j < positive_size_; ++j) *dst++ = src[k + j];                  // …:217-222 positives
    for (int j = 0; j < negative_size_; ++j) *dst++ = src[k + dataset_->win_pos + j];   // …:225-232 negatives
    (*video_ids)[i] = dataset_->win_video_id[cursor_];
    cursor_ = (cursor_ + 1) % nw;
  }
}
template class VideoShotWindowTestDataLayer<float>;

template <typename Dtype>
void RetrievalStatsLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>*) {
  const pl::Message& p = this->layer_param_.get_msg("retrieval_stats_param");
  CHECK(!p.get_bool("video_level_retrieval")) << "video_level_retrieval is not built";
  CHECK(p.get_str("stats_output_file").empty()) << "stats_output_file is not built";
  std::ifstream f(p.get_str("id_to_class_file"));
  string line;
  while (std::getline(f, line)) {                                  // retrieval_stats_layer.cpp:31-43
    if (line.empty()) continue;
    const size_t c = line.find(',');
    CHECK(c != string::npos && line.find(',', c + 1) == string::npos) << "Line: " << line;
    map_ids_.push_back(atoi(line.substr(0, c).c_str()));
    map_cls_.push_back(atoi(line.substr(c + 1).c_str()));
  }
  CHECK_GE(map_ids_.size(), 1u) << "need atleast one entry in id-to-class map!";      // :49
  // --num_classes (retrieval_stats_layer.cpp:16) sizes the reference's per-class accumulators (:218-223); a class id past
  // it indexes them out of bounds there.  Checked here instead.
  for (size_t i = 0; i < map_cls_.size(); ++i)
    CHECK_LT(map_cls_[i], FLAGS_num_classes) << "class " << map_cls_[i] << " of video " << map_ids_[i] << " needs --num_classes > " << map_cls_[i];
}
template <typename Dtype>
void RetrievalStatsLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  CHECK_EQ(bottom[0]->num(), bottom[1]->num()) << "The data and label should have the same number.";   // :76-77
  CHECK_EQ(bottom[1]->channels(), 1); CHECK_EQ(bottom[1]->height(), 1); CHECK_EQ(bottom[1]->width(), 1);
  for (int i = 0; i < 3; ++i) (*top)[i]->Reshape(1, 1, 1, 1);
}
template class RetrievalStatsLayer<float>;

// ------------------------------------------------------------------------------- shape layers --
template <typename Dtype>
void SliceLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const pl::Message& p = this->layer_param_.get_msg("slice_param");
  const int dim = (int)p.get_int("slice_dim");
  CHECK(dim == 0 || dim == 1) << "Slice dim should be 0 or 1";                               // slice_layer.cpp:17-19
  CHECK_EQ(p.size("slice_point"), 0) << "explicit slice_point is not used by the videovec graph";
  const int total = dim == 0 ? bottom[0]->num() : bottom[0]->channels();
  const int nt = (int)top->size();
  CHECK_EQ(total % nt, 0) << "Number of top blobs (" << nt << ") should evenly divide input";  // :46-50
  for (int i = 0; i < nt; ++i) {
    if (dim == 0) (*top)[i]->Reshape(total / nt, bottom[0]->channels(), bottom[0]->height(), bottom[0]->width());
    else (*top)[i]->Reshape(bottom[0]->num(), total / nt, bottom[0]->height(), bottom[0]->width());
  }
}
template <typename Dtype>
void ConcatLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int dim = (int)this->layer_param_.get_msg("concat_param").get_int("concat_dim");
  CHECK(dim == 0 || dim == 1) << "concat_dim should be 0 or 1";
  int num = bottom[0]->num(), ch = bottom[0]->channels();
  for (size_t i = 1; i < bottom.size(); ++i) {
    if (dim == 0) { num += bottom[i]->num(); CHECK_EQ(ch, bottom[i]->channels()); }
    else { ch += bottom[i]->channels(); CHECK_EQ(num, bottom[i]->num()); }
    CHECK_EQ(bottom[0]->height(), bottom[i]->height()); CHECK_EQ(bottom[0]->width(), bottom[i]->width());
  }
  (*top)[0]->Reshape(num, ch, bottom[0]->height(), bottom[0]->width());
}
template <typename Dtype>
void InnerProductLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>*) {
  const pl::Message& p = this->layer_param_.get_msg("inner_product_param");
  N_ = (int)p.get_int("num_output");
  bias_term_ = p.get_bool("bias_term");
  K_ = bottom[0]->count() / bottom[0]->num();
  CHECK_GE(N_, 1) << "num_output must be set";
  if (this->blobs_.size() > 0) { LOG(INFO) << "Skipping parameter initialization"; return; }   // :23-25
  this->blobs_.resize(bias_term_ ? 2 : 1);
  this->blobs_[0].reset(new Blob<Dtype>(1, 1, N_, K_));                                       // :29
  std::mt19937 rng(Caffe::random_seed());
  auto fill = [&](const pl::Message& fp, Blob<Dtype>* b) {
    const string t = fp.get_str("type");
    Dtype* d = b->mutable_cpu_data();
    if (t == "constant") for (int i = 0; i < b->count(); ++i) d[i] = (Dtype)fp.get_num("value");
    else if (t == "gaussian") { std::normal_distribution<float> nd((float)fp.get_num("mean"), (float)fp.get_num("std")); for (int i = 0; i < b->count(); ++i) d[i] = nd(rng); }
    else if (t == "uniform") { std::uniform_real_distribution<float> ud((float)fp.get_num("min"), (float)fp.get_num("max")); for (int i = 0; i < b->count(); ++i) d[i] = ud(rng); }
    else if (t == "xavier") { const float s = std::sqrt(3.0f / (b->count() / b->num())); std::uniform_real_distribution<float> ud(-s, s); for (int i = 0; i < b->count(); ++i) d[i] = ud(rng); }
    else LOG(FATAL) << "Unknown filler name: " << t;
  };
  fill(p.get_msg("weight_filler"), this->blobs_[0].get());
  if (bias_term_) { this->blobs_[1].reset(new Blob<Dtype>(1, 1, 1, N_)); fill(p.get_msg("bias_filler"), this->blobs_[1].get()); }   // :36
}
template <typename Dtype>
void EltwiseLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>*) {
  const pl::Message& p = this->layer_param_.get_msg("eltwise_param");
  const int nc = p.size("coeff");
  CHECK(nc == 0 || nc == (int)bottom.size()) << "Eltwise Layer takes one coefficient per bottom blob.";   // eltwise_layer.cpp:14-16
  CHECK(!(p.get_enum("operation") == "PROD" && nc)) << "Eltwise layer only takes coefficients for summation.";
  coeffs_.assign(bottom.size(), Dtype(1));
  for (int i = 0; i < nc; ++i) coeffs_[i] = (Dtype)p.get_num("coeff", i);
}
template <typename Dtype>
void EltwiseLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  for (size_t i = 1; i < bottom.size(); ++i) {
    CHECK_EQ(bottom[0]->num(), bottom[i]->num()); CHECK_EQ(bottom[0]->channels(), bottom[i]->channels());
    CHECK_EQ(bottom[0]->height(), bottom[i]->height()); CHECK_EQ(bottom[0]->width(), bottom[i]->width());
  }
  (*top)[0]->ReshapeLike(*bottom[0]);
}
template <typename Dtype>
void MaxMarginLossLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>*) {
  // LossLayer::LayerSetUp (loss_layer.cpp:13-20): default loss weight 1 on the first top
  if (this->layer_param_.size("loss_weight") == 0) this->layer_param_.add_num("loss_weight", 1.0);
  const pl::Message& mp = this->layer_param_.get_msg("max_margin_loss_param");
  const string file = mp.get_str("id_to_weight_file");
  if (!file.empty()) {                                               // max_margin_loss_layer.cpp:21-37
    std::ifstream f(file);
    string line;
    while (std::getline(f, line)) {
      const size_t c = line.find(',');
      CHECK(c != string::npos && line.find(',', c + 1) == string::npos) << "Line: " << line;
      size_t sval = 0;
      const int video_id = std::stoi(line.substr(0, c), &sval);
      CHECK_EQ(sval, c);
      const float weight = std::stof(line.substr(c + 1), &sval);
      CHECK_EQ(sval, line.size() - c - 1);
      CHECK_GE(weight, 0) << "All weights should be greater than 0";
      video_id_to_weight_.insert(std::make_pair(video_id, weight));
    }
  }
  use_direct_weight_ = mp.get_bool("use_direct_weight");
}
template <typename Dtype>
float MaxMarginLossLayer<Dtype>::WeightOf(float v) const {
  if (use_direct_weight_) { CHECK_GE(v, 0.f); return v; }            // :85
  auto it = video_id_to_weight_.find(static_cast<int>(v));
  return it == video_id_to_weight_.end() ? 0.f : it->second;
}
template <typename Dtype>
void MaxMarginLossLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  CHECK_EQ(bottom[0]->num(), bottom[1]->num()) << "The data and label should have the same number.";   // loss_layer.cpp:24-26
  CHECK_EQ(bottom[0]->count(), bottom[1]->count()) << "target_score and negative_scores must have the same count "
      << "(SUM num_output must equal num_negative_samples; the reference never checks this, quirk Q3)";
  (*top)[0]->Reshape(1, 1, 1, 1);
  if (top->size() >= 2) (*top)[1]->Reshape(1, 1, 1, 1);
}
template class SliceLayer<float>;
template class ConcatLayer<float>;
template class InnerProductLayer<float>;
template class EltwiseLayer<float>;
template class MaxMarginLossLayer<float>;

// ------------------------------------------------------------------------------- factory -------
template <typename Dtype>
Layer<Dtype>* GetLayer(const LayerParameter& param) {
  const string name = param.get_str("name");
  const string type = param.get_enum("type");
  if (type == "VIDEO_SAMPLED_SHOTS_DATA") return new VideoSampledShotsDataLayer<Dtype>(param);
  if (type == "SLICE") return new SliceLayer<Dtype>(param);
  if (type == "CONCAT") return new ConcatLayer<Dtype>(param);
  if (type == "FLATTEN") return new FlattenLayer<Dtype>(param);
  if (type == "SPLIT") return new SplitLayer<Dtype>(param);
  if (type == "INNER_PRODUCT") return new InnerProductLayer<Dtype>(param);
  if (type == "RELU") return new ReLULayer<Dtype>(param);
  if (type == "DROPOUT") return new DropoutLayer<Dtype>(param);
  if (type == "ELTWISE") return new EltwiseLayer<Dtype>(param);
  if (type == "NORMALIZATION") return new NormalizationLayer<Dtype>(param);
  if (type == "SUM") return new SumLayer<Dtype>(param);
  if (type == "MAX_MARGIN_LOSS") return new MaxMarginLossLayer<Dtype>(param);
  if (type == "VIDEO_SHOT_WINDOW_TEST_DATA") return new VideoShotWindowTestDataLayer<Dtype>(param);
  if (type == "RETRIEVAL_STATS") return new RetrievalStatsLayer<Dtype>(param);
  if (type == "NONE") LOG(FATAL) << "Layer " << name << " has unspecified type.";            // layer_factory.cpp:303
  LOG(FATAL) << "Layer " << name << " has type " << type << ", which is outside the videovec training path built here.";
  return nullptr;
}
template Layer<float>* GetLayer(const LayerParameter& param);

}  // namespace caffe
