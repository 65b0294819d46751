// proto_lite.cpp -- schema tables + text / wire codecs (see proto_lite.hpp).
#include "caffe/proto_lite.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace caffe {
namespace pl {

// ------------------------------------------------------------------------------------------------
// Schema.  Transcribed from the reference's src/caffe/proto/caffe.proto (line numbers in comments)
// and video_shot_sentences.proto.  "Opaque" = a message type this build stores but never interprets.
// ------------------------------------------------------------------------------------------------
#define OPT(n, name, t) {n, name, t, false, false, "", ""}
#define OPTD(n, name, t, d) {n, name, t, false, false, "", d}
#define REP(n, name, t) {n, name, t, true, false, "", ""}
#define PACKEDF(n, name) {n, name, T_FLOAT, true, true, "", ""}
#define OMSG(n, name, tn) {n, name, T_MSG, false, false, tn, ""}
#define RMSG(n, name, tn) {n, name, T_MSG, true, false, tn, ""}
#define OENUM(n, name, tn, d) {n, name, T_ENUM, false, false, tn, d}

static const std::vector<EnumDef>& Enums() {
  static const std::vector<EnumDef> e = {
      {"Phase", {{"TRAIN", 0}, {"TEST", 1}}},                                       // :182-185
      {"SolverMode", {{"CPU", 0}, {"GPU", 1}}},                                     // :141-144
      {"SolverType", {{"SGD", 0}, {"NESTEROV", 1}, {"ADAGRAD", 2}}},                // :155-159
      {"DimCheckMode", {{"STRICT", 0}, {"PERMISSIVE", 1}}},
      {"DB", {{"LEVELDB", 0}, {"LMDB", 1}}},
      {"CONTEXT", {{"PAIRWISE", 0}, {"WINDOW", 1}, {"PAST", 2}, {"PAST_CONTINUOUS", 3},
                   {"PAST_CONTINUOUS_FIXED", 4}}},                                  // :597-603
      {"EltwiseOp", {{"PROD", 0}, {"SUM", 1}, {"MAX", 2}}},                         // :721-725
      {"Norm", {{"L1", 1}, {"L2", 2}}},                                             // :859-862
      {"Engine", {{"DEFAULT", 0}, {"CAFFE", 1}, {"CUDNN", 2}}},
      {"LayerType",                                                                 // :236-302
       {{"NONE", 0}, {"ABSVAL", 35}, {"ACCURACY", 1}, {"ARGMAX", 30}, {"BNLL", 2},
        {"CLASSIFICATION_STATS", 39}, {"CONCAT", 3}, {"CONTRASTIVE_LOSS", 37}, {"CONVOLUTION", 4},
        {"DATA", 5}, {"DROPOUT", 6}, {"DUMMY_DATA", 32}, {"EUCLIDEAN_LOSS", 7}, {"ELTWISE", 25},
        {"FLATTEN", 8}, {"FLATTEN_BATCH", 55}, {"FIXED_VIDEO_SHOT_TEST_DATA", 51},
        {"FLEXIBLE_DATA", 38}, {"HDF5_DATA", 9}, {"HDF5_OUTPUT", 10}, {"HINGE_LOSS", 28},
        {"ID_TO_WEIGHT_MAPPING", 42}, {"IM2COL", 11}, {"IMAGE_DATA", 12}, {"INFOGAIN_LOSS", 13},
        {"INNER_PRODUCT", 14}, {"LRN", 15}, {"LSTM", 52}, {"LSTM_CONDITIONAL", 57},
        {"LSTM_ENC_DEC", 53}, {"LSTM_LINEAR", 59}, {"LSTM_SINGLE_STEP", 60}, {"MAX_MARGIN_LOSS", 43},
        {"MEMORY_DATA", 29}, {"MULTINOMIAL_LOGISTIC_LOSS", 16}, {"MVN", 34}, {"NORMALIZATION", 41},
        {"POOLING", 17}, {"POWER", 26}, {"RELU", 18}, {"RETRIEVAL_RANK_STATS", 47},
        {"RETRIEVAL_RANK_STATS_FIXED_REF", 50}, {"RETRIEVAL_STATS", 45}, {"SIGMOID", 19},
        {"SIGMOID_CROSS_ENTROPY_LOSS", 27}, {"SILENCE", 36}, {"SOCIAL_POOLING", 61}, {"SOFTMAX", 20},
        {"SOFTMAX_LOSS", 21}, {"SPLIT", 22}, {"SLICE", 33}, {"SUM", 44}, {"TANH", 23},
        {"TRACKING_WINDOWS_DATA", 54}, {"TRACKING_WINDOWS_SOCIAL_DATA", 62},
        {"VIDEO_SAMPLED_SHOTS_DATA", 49}, {"VIDEO_SHOT_WINDOW_TEST_DATA", 48},
        {"VIDEO_SHOT_WINDOW_DATA", 40}, {"VIDEO_SHOTS_DATA", 46}, {"WINDOW_DATA", 24},
        {"WRITE_TO_FILE", 56}, {"THRESHOLD", 31}}},
  };
  return e;
}

static const std::vector<MsgDef>& Msgs() {
  static const std::vector<MsgDef> m = {
      {"Opaque", {}},
      {"BlobProto",                                                                 // :5-15
       {OPTD(1, "num", T_INT32, "0"), OPTD(2, "channels", T_INT32, "0"), OPTD(3, "height", T_INT32, "0"),
        OPTD(4, "width", T_INT32, "0"), PACKEDF(5, "data"), PACKEDF(6, "diff"),
        OPTD(7, "truncated_num", T_INT32, "0"), OPTD(8, "truncated_height", T_INT32, "0")}},
      {"Datum",                                                                     // :23-37
       {OPT(1, "channels", T_INT32), OPT(2, "height", T_INT32), OPT(3, "width", T_INT32),
        OPT(4, "data", T_BYTES), OPT(5, "label", T_INT32), {6, "float_data", T_FLOAT, true, false, "", ""},
        REP(7, "mean", T_FLOAT), REP(8, "min", T_FLOAT), REP(9, "max", T_FLOAT)}},
      {"FillerParameter",                                                           // :39-49
       {OPTD(1, "type", T_STRING, "constant"), OPTD(2, "value", T_FLOAT, "0"), OPTD(3, "min", T_FLOAT, "0"),
        OPTD(4, "max", T_FLOAT, "1"), OPTD(5, "mean", T_FLOAT, "0"), OPTD(6, "std", T_FLOAT, "1"),
        OPTD(7, "sparse", T_INT32, "-1")}},
      {"NetState", {OENUM(1, "phase", "Phase", "TEST"), OPTD(2, "level", T_INT32, "0"), REP(3, "stage", T_STRING)}},
      {"NetStateRule",
       {OENUM(1, "phase", "Phase", ""), OPT(2, "min_level", T_INT32), OPT(3, "max_level", T_INT32),
        REP(4, "stage", T_STRING), REP(5, "not_stage", T_STRING)}},
      {"NetParameter",                                                              // :51-66
       {OPT(1, "name", T_STRING), RMSG(2, "layers", "LayerParameter"), REP(3, "input", T_STRING),
        REP(4, "input_dim", T_INT32), OPTD(5, "force_backward", T_BOOL, "false"), OMSG(6, "state", "NetState")}},
      {"SolverParameter",                                                           // :75-173
       {OPT(24, "net", T_STRING), OMSG(25, "net_param", "NetParameter"), OPT(1, "train_net", T_STRING),
        REP(2, "test_net", T_STRING), OMSG(21, "train_net_param", "NetParameter"),
        RMSG(22, "test_net_param", "NetParameter"), OMSG(26, "train_state", "NetState"),
        RMSG(27, "test_state", "NetState"), REP(3, "test_iter", T_INT32),
        OPTD(4, "test_interval", T_INT32, "0"), OPTD(19, "test_compute_loss", T_BOOL, "false"),
        OPTD(32, "test_initialization", T_BOOL, "true"), OPT(5, "base_lr", T_FLOAT), OPT(6, "display", T_INT32),
        OPT(7, "max_iter", T_INT32), OPT(8, "lr_policy", T_STRING), OPT(9, "gamma", T_FLOAT),
        OPT(10, "power", T_FLOAT), OPT(11, "momentum", T_FLOAT), OPT(12, "weight_decay", T_FLOAT),
        OPTD(29, "regularization_type", T_STRING, "L2"), OPT(13, "stepsize", T_INT32),
        OPTD(14, "snapshot", T_INT32, "0"), OPT(15, "snapshot_prefix", T_STRING),
        OPTD(16, "snapshot_diff", T_BOOL, "false"), OENUM(17, "solver_mode", "SolverMode", "GPU"),
        OPTD(18, "device_id", T_INT32, "0"), OPTD(20, "random_seed", T_INT64, "-1"),
        OENUM(30, "solver_type", "SolverType", "SGD"), OPTD(31, "delta", T_FLOAT, "1e-8"),
        OPTD(23, "debug_info", T_BOOL, "false"), OPTD(28, "snapshot_after_train", T_BOOL, "true"),
        OPT(33, "snapshot_vis", T_INT32), OPT(34, "snapshot_vis_blobs", T_STRING),
        OPT(35, "snapshot_vis_truncate_len", T_INT32), OPT(36, "snapshot_vis_dir", T_STRING)}},
      {"SolverState",                                                               // :176-180
       {OPT(1, "iter", T_INT32), OPT(2, "learned_net", T_STRING), RMSG(3, "history", "BlobProto")}},
      {"LayerParameter",                                                            // :215-389
       {REP(2, "bottom", T_STRING), REP(3, "top", T_STRING), OPT(4, "name", T_STRING),
        RMSG(32, "include", "NetStateRule"), RMSG(33, "exclude", "NetStateRule"),
        OENUM(5, "type", "LayerType", "NONE"), RMSG(6, "blobs", "BlobProto"), REP(1001, "param", T_STRING),
        {1002, "blob_share_mode", T_ENUM, true, false, "DimCheckMode", ""},
        REP(7, "blobs_lr", T_FLOAT), REP(8, "weight_decay", T_FLOAT), REP(35, "loss_weight", T_FLOAT),
        OMSG(27, "accuracy_param", "Opaque"), OMSG(23, "argmax_param", "Opaque"),
        OMSG(42, "classification_stats_param", "Opaque"), OMSG(9, "concat_param", "ConcatParameter"),
        OMSG(40, "contrastive_loss_param", "Opaque"), OMSG(10, "convolution_param", "Opaque"),
        OMSG(11, "data_param", "Opaque"), OMSG(12, "dropout_param", "DropoutParameter"),
        OMSG(26, "dummy_data_param", "Opaque"), OMSG(24, "eltwise_param", "EltwiseParameter"),
        OMSG(57, "euclidean_loss_param", "Opaque"), OMSG(56, "flatten_batch_param", "Opaque"),
        OMSG(53, "fixed_video_shot_test_data_param", "Opaque"), OMSG(41, "flexible_data_param", "Opaque"),
        OMSG(13, "hdf5_data_param", "Opaque"), OMSG(14, "hdf5_output_param", "Opaque"),
        OMSG(29, "hinge_loss_param", "Opaque"), OMSG(44, "id_to_weight_mapping_param", "Opaque"),
        OMSG(15, "image_data_param", "Opaque"), OMSG(16, "infogain_loss_param", "Opaque"),
        OMSG(17, "inner_product_param", "InnerProductParameter"), OMSG(54, "lstm_param", "Opaque"),
        OMSG(18, "lrn_param", "Opaque"), OMSG(45, "max_margin_loss_param", "MaxMarginLossParameter"),
        OMSG(22, "memory_data_param", "Opaque"), OMSG(34, "mvn_param", "Opaque"),
        OMSG(19, "pooling_param", "Opaque"), OMSG(21, "power_param", "Opaque"),
        OMSG(30, "relu_param", "ReLUParameter"), OMSG(49, "retrieval_rank_stats_param", "Opaque"),
        OMSG(52, "retrieval_rank_stats_fixed_ref_param", "Opaque"),
        OMSG(47, "retrieval_stats_param", "RetrievalStatsParameter"), OMSG(38, "sigmoid_param", "Opaque"),
        OMSG(39, "softmax_param", "Opaque"), OMSG(31, "slice_param", "SliceParameter"),
        OMSG(59, "social_pooling_param", "Opaque"), OMSG(46, "sum_param", "SumParameter"),
        OMSG(37, "tanh_param", "Opaque"), OMSG(55, "tracking_windows_data_param", "Opaque"),
        OMSG(50, "video_shot_window_test_data_param", "VideoShotWindowTestDataParameter"),
        OMSG(25, "threshold_param", "Opaque"),
        OMSG(51, "video_sampled_shots_data_param", "VideoSampledShotsDataParameter"),
        OMSG(48, "video_shots_data_param", "Opaque"), OMSG(43, "video_shot_window_data_param", "Opaque"),
        OMSG(58, "write_to_file_param", "Opaque"), OMSG(20, "window_data_param", "Opaque"),
        OMSG(36, "transform_param", "Opaque"), OMSG(1, "layer", "Opaque")}},
      {"VideoSampledShotsDataParameter",                                            // :562-620
       {OPT(1, "source", T_STRING), OPT(4, "batch_size", T_UINT32), OPTD(7, "rand_skip", T_UINT32, "0"),
        OENUM(8, "backend", "DB", "LEVELDB"), OPTD(9, "num_negative_samples", T_UINT32, "0"),
        OPTD(10, "max_buffer_size", T_UINT32, "0"), OPTD(11, "negative_swap_percentage", T_UINT32, "0"),
        OPTD(12, "negative_dataset", T_STRING, ""), OENUM(14, "context_type", "CONTEXT", "PAIRWISE"),
        OPTD(15, "context_size", T_UINT32, "1"), OPTD(16, "output_shot_distance", T_BOOL, "false"),
        OPTD(17, "max_shot_distance", T_FLOAT, "5.0"), OPTD(18, "max_same_video_negs", T_UINT32, "0")}},
      {"VideoShotWindowTestDataParameter",                                          // :538-559
       {OPT(1, "source", T_STRING), OPT(4, "batch_size", T_UINT32), OENUM(8, "backend", "DB", "LEVELDB"),
        OPTD(13, "display_all_ids", T_BOOL, "false"), OPTD(14, "include_positives", T_BOOL, "true"),
        OPTD(15, "include_negatives", T_BOOL, "true")}},
      {"RetrievalStatsParameter",                                                   // :955-966
       {OPT(1, "id_to_class_file", T_STRING), OPTD(2, "stats_output_file", T_STRING, ""),
        OPTD(3, "exclude_same_video_shots", T_BOOL, "true"), OPTD(4, "video_level_retrieval", T_BOOL, "false"),
        OPTD(5, "max_num_videos", T_INT32, "0")}},
      {"InnerProductParameter",                                                     // :831-837
       {OPT(1, "num_output", T_UINT32), OPTD(2, "bias_term", T_BOOL, "true"),
        OMSG(3, "weight_filler", "FillerParameter"), OMSG(4, "bias_filler", "FillerParameter"),
        OPTD(5, "regularization", T_DOUBLE, "0")}},
      {"DropoutParameter", {OPTD(1, "dropout_ratio", T_FLOAT, "0.5")}},              // :697-699
      {"EltwiseParameter",                                                          // :720-732
       {OENUM(1, "operation", "EltwiseOp", "SUM"), REP(2, "coeff", T_FLOAT),
        OPTD(3, "stable_prod_grad", T_BOOL, "true")}},
      {"SumParameter", {OPTD(1, "num_output", T_FLOAT, "1")}},                       // :742-744
      {"SliceParameter", {OPTD(1, "slice_dim", T_UINT32, "1"), REP(2, "slice_point", T_UINT32)}},
      {"ConcatParameter", {OPTD(1, "concat_dim", T_UINT32, "1")}},
      {"ReLUParameter", {OPTD(1, "negative_slope", T_FLOAT, "0"), OENUM(2, "engine", "Engine", "DEFAULT")}},
      {"MaxMarginLossParameter",                                                    // :858-868
       {OENUM(1, "norm", "Norm", "L1"), OPTD(2, "id_to_weight_file", T_STRING, ""),
        OPTD(3, "use_direct_weight", T_BOOL, "false"), OPTD(4, "margin", T_FLOAT, "1.0")}},
      // video_shot_sentences.proto:15-20, 22-30
      {"VideoShots",
       {OPT(1, "video_id", T_INT32), REP(2, "shot_ids", T_INT32), RMSG(3, "shot_words", "Datum"),
        OPT(4, "video_name", T_STRING)}},
      {"TestVideoShotWindows",
       {OPT(1, "video_id", T_INT32), REP(2, "positive_shot_id", T_INT32), OPT(3, "video_name", T_STRING),
        RMSG(4, "positive_shot_words", "Datum"), RMSG(5, "context_shot_words", "Datum"),
        RMSG(6, "negative_shot_words", "Datum"), REP(7, "negative_shot_id", T_INT32)}},
  };
  return m;
}

const MsgDef* FindMsg(const std::string& name) {
  for (const auto& m : Msgs()) if (name == m.name) return &m;
  return nullptr;
}
const EnumDef* FindEnum(const std::string& name) {
  for (const auto& e : Enums()) if (name == e.name) return &e;
  return nullptr;
}

static void die(const std::string& msg) {
  fprintf(stderr, "F proto_lite] %s\n", msg.c_str());
  abort();
}

// ------------------------------------------------------------------------------------------------
Message::Message(const std::string& type) : def_(FindMsg(type)) {
  if (!def_) die("unknown message type " + type);
}
Message& Message::operator=(const Message& o) {
  if (this == &o) return *this;
  def_ = o.def_; f_ = o.f_; packed_f_ = o.packed_f_; unknown_ = o.unknown_;
  for (auto& kv : f_)
    for (auto& v : kv.second)
      if (v.m) v.m.reset(new Message(*v.m));
  return *this;
}
const FieldDef* Message::field(const char* f) const {
  for (const auto& fd : def_->fields) if (!strcmp(fd.name, f)) return &fd;
  die(std::string("no field '") + f + "' in " + def_->name);
  return nullptr;
}
const FieldDef* Message::field_by_num(int num) const {
  for (const auto& fd : def_->fields) if (fd.num == num) return &fd;
  return nullptr;
}
static bool is_packed_float(const FieldDef* fd) { return fd->repeated && fd->type == T_FLOAT; }

bool Message::has(const char* f) const { return size(f) > 0; }
int Message::size(const char* f) const {
  const FieldDef* fd = field(f);
  if (is_packed_float(fd)) { auto it = packed_f_.find(fd->num); return it == packed_f_.end() ? 0 : (int)it->second.size(); }
  auto it = f_.find(fd->num);
  return it == f_.end() ? 0 : (int)it->second.size();
}
static int enum_value(const FieldDef* fd, const std::string& name, bool* ok) {
  const EnumDef* e = FindEnum(fd->tname);
  for (const auto& kv : e->values) if (name == kv.first) { *ok = true; return kv.second; }
  *ok = false;
  return 0;
}
static double default_num(const FieldDef* fd) {
  if (!fd->def[0]) return 0;
  if (fd->type == T_BOOL) return !strcmp(fd->def, "true");
  if (fd->type == T_ENUM) { bool ok; return enum_value(fd, fd->def, &ok); }
  return atof(fd->def);
}
int64_t Message::get_int(const char* f, int idx) const {
  const FieldDef* fd = field(f);
  auto it = f_.find(fd->num);
  if (it == f_.end() || idx >= (int)it->second.size()) return (int64_t)default_num(fd);
  const Value& v = it->second[idx];
  return (fd->type == T_FLOAT || fd->type == T_DOUBLE) ? (int64_t)v.d : v.i;
}
double Message::get_num(const char* f, int idx) const {
  const FieldDef* fd = field(f);
  if (is_packed_float(fd)) {
    auto it = packed_f_.find(fd->num);
    if (it == packed_f_.end() || idx >= (int)it->second.size()) return default_num(fd);
    return it->second[idx];
  }
  auto it = f_.find(fd->num);
  if (it == f_.end() || idx >= (int)it->second.size()) return default_num(fd);
  const Value& v = it->second[idx];
  return (fd->type == T_FLOAT || fd->type == T_DOUBLE) ? v.d : (double)v.i;
}
const std::string& Message::get_str(const char* f, int idx) const {
  static thread_local std::string tmp;
  const FieldDef* fd = field(f);
  auto it = f_.find(fd->num);
  if (it == f_.end() || idx >= (int)it->second.size()) { tmp = fd->def; return tmp; }
  return it->second[idx].s;
}
std::string Message::get_enum(const char* f, int idx) const {
  const FieldDef* fd = field(f);
  auto it = f_.find(fd->num);
  if (it == f_.end() || idx >= (int)it->second.size()) return fd->def;
  const EnumDef* e = FindEnum(fd->tname);
  for (const auto& kv : e->values) if (kv.second == it->second[idx].i) return kv.first;
  return "";
}
const Message& Message::get_msg(const char* f, int idx) const {
  const FieldDef* fd = field(f);
  auto it = f_.find(fd->num);
  if (it == f_.end() || idx >= (int)it->second.size()) {
    static thread_local std::map<std::string, std::shared_ptr<Message>> empties;
    auto& e = empties[fd->tname];
    if (!e) e.reset(new Message(std::string(fd->tname)));
    return *e;
  }
  return *it->second[idx].m;
}
const std::vector<float>& Message::floats(const char* f) const {
  static const std::vector<float> empty;
  auto it = packed_f_.find(field(f)->num);
  return it == packed_f_.end() ? empty : it->second;
}
void Message::clear(const char* f) { const FieldDef* fd = field(f); f_.erase(fd->num); packed_f_.erase(fd->num); }
void Message::set_int(const char* f, int64_t v) { auto& vv = vals(field(f)); vv.resize(1); vv[0].i = v; vv[0].d = (double)v; }
void Message::set_num(const char* f, double v) { auto& vv = vals(field(f)); vv.resize(1); vv[0].d = v; vv[0].i = (int64_t)v; }
void Message::set_str(const char* f, const std::string& v) { auto& vv = vals(field(f)); vv.resize(1); vv[0].s = v; }
void Message::set_enum(const char* f, const std::string& name) {
  const FieldDef* fd = field(f);
  bool ok; const int v = enum_value(fd, name, &ok);
  if (!ok) die("bad enum value " + name + " for " + f);
  auto& vv = vals(fd); vv.resize(1); vv[0].i = v;
}
void Message::add_int(const char* f, int64_t v) { Value x; x.i = v; x.d = (double)v; vals(field(f)).push_back(x); }
void Message::add_num(const char* f, double v) {
  const FieldDef* fd = field(f);
  if (is_packed_float(fd)) { packed_f_[fd->num].push_back((float)v); return; }
  Value x; x.d = v; x.i = (int64_t)v; vals(fd).push_back(x);
}
void Message::add_str(const char* f, const std::string& v) { Value x; x.s = v; vals(field(f)).push_back(x); }
Message* Message::mutable_msg(const char* f) {
  const FieldDef* fd = field(f);
  auto& vv = vals(fd);
  if (vv.empty()) { vv.resize(1); vv[0].m.reset(new Message(std::string(fd->tname))); }
  return vv[0].m.get();
}
Message* Message::add_msg(const char* f) {
  const FieldDef* fd = field(f);
  Value x; x.m.reset(new Message(std::string(fd->tname)));
  vals(fd).push_back(x);
  return vals(fd).back().m.get();
}
std::vector<float>* Message::mutable_floats(const char* f) { return &packed_f_[field(f)->num]; }

// ------------------------------------------------------------------------------------------------
// text format
// ------------------------------------------------------------------------------------------------
class TextParser {
 public:
  TextParser(const std::string& s) : s_(s), p_(0), line_(1) {}
  bool ParseMessage(Message* m, bool top, std::string* err) {
    for (;;) {
      skip();
      if (p_ >= s_.size()) { if (top) return true; return fail(err, "unexpected end of input"); }
      if (!top && (s_[p_] == '}' || s_[p_] == '>')) { ++p_; return true; }
      std::string name;
      if (!ident(&name)) return fail(err, "expected a field name");
      const FieldDef* fd = nullptr;
      for (const auto& f : m->def_->fields) if (name == f.name) fd = &f;
      skip();
      bool colon = false;
      if (p_ < s_.size() && s_[p_] == ':') { colon = true; ++p_; skip(); }
      if (!fd || (fd->type == T_MSG && !strcmp(fd->tname, "Opaque"))) {
        // unknown field (or a message this build does not model): skip its value
        if (!fd && strcmp(m->def_->name, "Opaque"))
          fprintf(stderr, "W proto_lite] line %d: ignoring unknown field '%s' in %s\n", line_, name.c_str(), m->def_->name);
        if (!skip_value(err)) return false;
        if (fd) m->mutable_msg(fd->name);
        continue;
      }
      if (fd->type == T_MSG) {
        if (p_ >= s_.size() || (s_[p_] != '{' && s_[p_] != '<')) return fail(err, "expected '{' after " + name);
        ++p_;
        Message* sub = fd->repeated ? m->add_msg(fd->name) : m->mutable_msg(fd->name);
        if (!ParseMessage(sub, false, err)) return false;
        continue;
      }
      if (!colon) return fail(err, "expected ':' after " + name);
      if (p_ < s_.size() && s_[p_] == '[') {        // name: [a, b, c]
        ++p_;
        for (;;) {
          skip();
          if (p_ < s_.size() && s_[p_] == ']') { ++p_; break; }
          if (!scalar(m, fd, err)) return false;
          skip();
          if (p_ < s_.size() && s_[p_] == ',') ++p_;
        }
      } else if (!scalar(m, fd, err)) return false;
      skip();
      if (p_ < s_.size() && (s_[p_] == ',' || s_[p_] == ';')) ++p_;
    }
  }

 private:
  bool fail(std::string* err, const std::string& what) {
    std::ostringstream o; o << "text format, line " << line_ << ": " << what;
    *err = o.str();
    return false;
  }
  void skip() {
    while (p_ < s_.size()) {
      const char c = s_[p_];
      if (c == '\n') { ++line_; ++p_; }
      else if (isspace((unsigned char)c)) ++p_;
      else if (c == '#') { while (p_ < s_.size() && s_[p_] != '\n') ++p_; }
      else break;
    }
  }
  bool ident(std::string* out) {
    size_t b = p_;
    while (p_ < s_.size() && (isalnum((unsigned char)s_[p_]) || s_[p_] == '_' || s_[p_] == '.')) ++p_;
    *out = s_.substr(b, p_ - b);
    return p_ > b;
  }
  bool token(std::string* out) {       // number / identifier token
    size_t b = p_;
    while (p_ < s_.size() && (isalnum((unsigned char)s_[p_]) || strchr("_.+-", s_[p_]))) ++p_;
    *out = s_.substr(b, p_ - b);
    return p_ > b;
  }
  bool quoted(std::string* out, std::string* err) {
    out->clear();
    for (;;) {                         // adjacent string literals concatenate
      skip();
      if (p_ >= s_.size() || (s_[p_] != '"' && s_[p_] != '\'')) return true;
      const char q = s_[p_++];
      while (p_ < s_.size() && s_[p_] != q) {
        char c = s_[p_++];
        if (c == '\\' && p_ < s_.size()) {
          c = s_[p_++];
          switch (c) {
            case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break;
            case '0': c = '\0'; break; default: break;
          }
        }
        out->push_back(c);
      }
      if (p_ >= s_.size()) return fail(err, "unterminated string");
      ++p_;
    }
  }
  bool scalar(Message* m, const FieldDef* fd, std::string* err) {
    skip();
    if (fd->type == T_STRING || fd->type == T_BYTES) {
      if (p_ >= s_.size() || (s_[p_] != '"' && s_[p_] != '\'')) return fail(err, std::string("expected a string for ") + fd->name);
      std::string v;
      if (!quoted(&v, err)) return false;
      if (fd->repeated) m->add_str(fd->name, v); else m->set_str(fd->name, v);
      return true;
    }
    std::string t;
    if (!token(&t)) return fail(err, std::string("expected a value for ") + fd->name);
    if (fd->type == T_ENUM) {
      bool ok; int v = enum_value(fd, t, &ok);
      if (!ok) {
        char* e; long n = strtol(t.c_str(), &e, 10);
        if (*e) return fail(err, "unknown enum value " + t + " for " + fd->name);
        v = (int)n;
      }
      if (fd->repeated) m->add_int(fd->name, v); else m->set_int(fd->name, v);
      return true;
    }
    if (fd->type == T_BOOL) {
      int v;
      if (t == "true" || t == "True" || t == "t" || t == "1") v = 1;
      else if (t == "false" || t == "False" || t == "f" || t == "0") v = 0;
      else return fail(err, "bad bool " + t);
      if (fd->repeated) m->add_int(fd->name, v); else m->set_int(fd->name, v);
      return true;
    }
    if (fd->type == T_FLOAT || fd->type == T_DOUBLE) {
      std::string u = t;
      if (!u.empty() && (u.back() == 'f' || u.back() == 'F') && u != "inf" && u != "-inf") u.pop_back();
      char* e; double v = strtod(u.c_str(), &e);
      if (*e) return fail(err, "bad number " + t + " for " + fd->name);
      if (fd->type == T_FLOAT) v = (double)(float)v;
      if (fd->repeated) m->add_num(fd->name, v); else m->set_num(fd->name, v);
      return true;
    }
    char* e; long long v = strtoll(t.c_str(), &e, 0);
    if (*e) return fail(err, "bad integer " + t + " for " + fd->name);
    if (fd->repeated) m->add_int(fd->name, v); else m->set_int(fd->name, v);
    return true;
  }
  bool skip_value(std::string* err) {
    skip();
    if (p_ < s_.size() && (s_[p_] == '{' || s_[p_] == '<')) {
      int depth = 0;
      while (p_ < s_.size()) {
        const char c = s_[p_];
        if (c == '"' || c == '\'') { std::string d; if (!quoted(&d, err)) return false; continue; }
        if (c == '#') { skip(); continue; }
        if (c == '\n') ++line_;
        if (c == '{' || c == '<') ++depth;
        if (c == '}' || c == '>') { --depth; if (depth == 0) { ++p_; return true; } }
        ++p_;
      }
      return fail(err, "unbalanced braces");
    }
    if (p_ < s_.size() && (s_[p_] == '"' || s_[p_] == '\'')) { std::string d; return quoted(&d, err); }
    std::string t;
    return token(&t) ? true : fail(err, "expected a value");
  }
  const std::string& s_;
  size_t p_;
  int line_;
};

bool Message::ParseText(const std::string& text, std::string* err) {
  f_.clear(); packed_f_.clear(); unknown_.clear();
  TextParser p(text);
  return p.ParseMessage(this, true, err);
}

static std::string fmt_float(double v, bool is_float) {
  // shortest decimal that round-trips (what protobuf's TextFormat prints)
  char buf[64];
  if (is_float) {
    const float f = (float)v;
    for (int prec = 6; prec <= 9; ++prec) { snprintf(buf, sizeof(buf), "%.*g", prec, (double)f); if (strtof(buf, nullptr) == f) break; }
  } else {
    for (int prec = 15; prec <= 17; ++prec) { snprintf(buf, sizeof(buf), "%.*g", prec, v); if (strtod(buf, nullptr) == v) break; }
  }
  return buf;
}
static std::string escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); }
    else if (c == '\n') o += "\\n";
    else o.push_back(c);
  }
  return o;
}

std::string Message::PrintText(int indent) const {
  std::ostringstream o;
  const std::string pad(indent, ' ');
  for (const auto& fd : def_->fields) {
    if (is_packed_float(&fd)) {
      auto it = packed_f_.find(fd.num);
      if (it != packed_f_.end()) for (float v : it->second) o << pad << fd.name << ": " << fmt_float(v, true) << "\n";
      continue;
    }
    auto it = f_.find(fd.num);
    if (it == f_.end()) continue;
    for (const Value& v : it->second) {
      switch (fd.type) {
        case T_MSG:
          o << pad << fd.name << " {\n" << v.m->PrintText(indent + 2) << pad << "}\n";
          break;
        case T_STRING: case T_BYTES: o << pad << fd.name << ": \"" << escape(v.s) << "\"\n"; break;
        case T_ENUM: {
          const EnumDef* e = FindEnum(fd.tname);
          const char* nm = nullptr;
          for (const auto& kv : e->values) if (kv.second == v.i) nm = kv.first;
          if (nm) o << pad << fd.name << ": " << nm << "\n"; else o << pad << fd.name << ": " << v.i << "\n";
          break;
        }
        case T_BOOL: o << pad << fd.name << ": " << (v.i ? "true" : "false") << "\n"; break;
        case T_FLOAT: o << pad << fd.name << ": " << fmt_float(v.d, true) << "\n"; break;
        case T_DOUBLE: o << pad << fd.name << ": " << fmt_float(v.d, false) << "\n"; break;
        default: o << pad << fd.name << ": " << v.i << "\n";
      }
    }
  }
  return o.str();
}

// ------------------------------------------------------------------------------------------------
// wire format (varint / 64-bit / length-delimited / 32-bit)
// ------------------------------------------------------------------------------------------------
static void put_varint(std::string* o, uint64_t v) {
  while (v >= 0x80) { o->push_back((char)(v | 0x80)); v >>= 7; }
  o->push_back((char)v);
}
static bool get_varint(const uint8_t*& p, const uint8_t* e, uint64_t* v) {
  *v = 0;
  for (int shift = 0; p < e && shift < 64; shift += 7) {
    const uint8_t b = *p++;
    *v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
static void put_tag(std::string* o, int num, int wt) { put_varint(o, ((uint64_t)num << 3) | wt); }

void Message::SerializeBinary(std::string* out) const {
  for (const auto& fd : def_->fields) {
    if (is_packed_float(&fd)) {
      auto it = packed_f_.find(fd.num);
      if (it == packed_f_.end() || it->second.empty()) continue;
      if (fd.packed) {
        put_tag(out, fd.num, 2);
        put_varint(out, it->second.size() * 4);
        out->append((const char*)it->second.data(), it->second.size() * 4);
      } else {
        for (float v : it->second) { put_tag(out, fd.num, 5); out->append((const char*)&v, 4); }
      }
      continue;
    }
    auto it = f_.find(fd.num);
    if (it == f_.end()) continue;
    for (const Value& v : it->second) {
      switch (fd.type) {
        case T_MSG: {
          std::string sub;
          v.m->SerializeBinary(&sub);
          put_tag(out, fd.num, 2); put_varint(out, sub.size()); out->append(sub);
          break;
        }
        case T_STRING: case T_BYTES:
          put_tag(out, fd.num, 2); put_varint(out, v.s.size()); out->append(v.s);
          break;
        case T_FLOAT: { float f = (float)v.d; put_tag(out, fd.num, 5); out->append((const char*)&f, 4); break; }
        case T_DOUBLE: { double d = v.d; put_tag(out, fd.num, 1); out->append((const char*)&d, 8); break; }
        default:      // int32 / uint32 / int64 / bool / enum: varint (negative int32 sign-extended)
          put_tag(out, fd.num, 0); put_varint(out, (uint64_t)v.i);
      }
    }
  }
  out->append(unknown_);
}

bool Message::ParseBinary(const void* data, size_t n, std::string* err) {
  f_.clear(); packed_f_.clear(); unknown_.clear();
  const uint8_t* p = (const uint8_t*)data;
  const uint8_t* e = p + n;
  while (p < e) {
    const uint8_t* field_start = p;
    uint64_t tag;
    if (!get_varint(p, e, &tag)) { *err = "wire: bad tag"; return false; }
    const int num = (int)(tag >> 3), wt = (int)(tag & 7);
    const FieldDef* fd = field_by_num(num);
    uint64_t v = 0; const uint8_t* payload = nullptr; uint64_t len = 0;
    switch (wt) {
      case 0: if (!get_varint(p, e, &v)) { *err = "wire: bad varint"; return false; } break;
      case 1: if (e - p < 8) { *err = "wire: truncated"; return false; } payload = p; len = 8; p += 8; break;
      case 5: if (e - p < 4) { *err = "wire: truncated"; return false; } payload = p; len = 4; p += 4; break;
      case 2:
        if (!get_varint(p, e, &len) || (uint64_t)(e - p) < len) { *err = "wire: bad length"; return false; }
        payload = p; p += len;
        break;
      default: *err = "wire: unsupported wire type"; return false;
    }
    if (!fd) { unknown_.append((const char*)field_start, p - field_start); continue; }
    if (fd->type == T_MSG) {
      if (wt != 2) { *err = "wire: message field with non-length wire type"; return false; }
      Message* sub = fd->repeated ? add_msg(fd->name) : mutable_msg(fd->name);
      if (!strcmp(fd->tname, "Opaque")) sub->unknown_.assign((const char*)payload, len);
      else if (!sub->ParseBinary(payload, len, err)) return false;
      continue;
    }
    if (fd->type == T_STRING || fd->type == T_BYTES) {
      if (wt != 2) { *err = "wire: string field with non-length wire type"; return false; }
      const std::string s((const char*)payload, len);
      if (fd->repeated) add_str(fd->name, s); else set_str(fd->name, s);
      continue;
    }
    if (fd->type == T_FLOAT) {
      if (wt == 2) {       // packed
        if (len % 4) { *err = "wire: packed float length"; return false; }
        if (is_packed_float(fd)) {
          auto& dst = packed_f_[fd->num];
          const size_t old = dst.size();
          dst.resize(old + len / 4);
          memcpy(dst.data() + old, payload, len);
        } else { *err = "wire: packed data for a scalar float"; return false; }
      } else if (wt == 5) {
        float f; memcpy(&f, payload, 4);
        if (fd->repeated) add_num(fd->name, f); else set_num(fd->name, f);
      } else { *err = "wire: float with wrong wire type"; return false; }
      continue;
    }
    if (fd->type == T_DOUBLE) {
      if (wt != 1) { *err = "wire: double with wrong wire type"; return false; }
      double d; memcpy(&d, payload, 8);
      if (fd->repeated) add_num(fd->name, d); else set_num(fd->name, d);
      continue;
    }
    // integer kinds
    auto store = [&](uint64_t raw) {
      int64_t x = (int64_t)raw;
      if (fd->type == T_INT32 || fd->type == T_ENUM) x = (int32_t)raw;
      if (fd->type == T_UINT32) x = (uint32_t)raw;
      if (fd->repeated) add_int(fd->name, x); else set_int(fd->name, x);
    };
    if (wt == 0) store(v);
    else if (wt == 2) {    // packed repeated varints
      const uint8_t* q = payload; const uint8_t* qe = payload + len;
      while (q < qe) { uint64_t x; if (!get_varint(q, qe, &x)) { *err = "wire: bad packed varint"; return false; } store(x); }
    } else { *err = "wire: integer with wrong wire type"; return false; }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
static bool slurp(const std::string& fn, std::string* out) {
  std::ifstream f(fn, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss; ss << f.rdbuf();
  *out = ss.str();
  return true;
}
bool ReadProtoFromTextFile(const std::string& fn, Message* proto) {
  std::string s, err;
  if (!slurp(fn, &s)) { fprintf(stderr, "E proto_lite] File not found: %s\n", fn.c_str()); return false; }
  if (!proto->ParseText(s, &err)) { fprintf(stderr, "E proto_lite] %s: %s\n", fn.c_str(), err.c_str()); return false; }
  return true;
}
void ReadProtoFromTextFileOrDie(const std::string& fn, Message* proto) {
  if (!ReadProtoFromTextFile(fn, proto)) die("Check failed: ReadProtoFromTextFile(" + fn + ")");
}
void WriteProtoToTextFile(const Message& proto, const std::string& fn) {
  std::ofstream f(fn);
  if (!f) die("cannot write " + fn);
  f << proto.PrintText();
}
bool ReadProtoFromBinaryFile(const std::string& fn, Message* proto) {
  std::string s, err;
  if (!slurp(fn, &s)) { fprintf(stderr, "E proto_lite] File not found: %s\n", fn.c_str()); return false; }
  if (!proto->ParseBinary(s.data(), s.size(), &err)) { fprintf(stderr, "E proto_lite] %s: %s\n", fn.c_str(), err.c_str()); return false; }
  return true;
}
void ReadProtoFromBinaryFileOrDie(const std::string& fn, Message* proto) {
  if (!ReadProtoFromBinaryFile(fn, proto)) die("Check failed: ReadProtoFromBinaryFile(" + fn + ")");
}
void WriteProtoToBinaryFile(const Message& proto, const std::string& fn) {
  std::string s;
  proto.SerializeBinary(&s);
  std::ofstream f(fn, std::ios::binary | std::ios::trunc);
  if (!f) die("cannot write " + fn);
  f.write(s.data(), s.size());
}

}  // namespace pl
}  // namespace caffe
