// proto_lite.hpp -- a small schema-driven protobuf implementation (text format + binary wire
// format) for exactly the messages the videovec path reads and writes.  There is no libprotobuf /
// protoc in the target image, and the reference's formats must stay drop-in:
//   prototxt       : ReadProtoFromTextFile  (reference src/caffe/util/io.cpp:31-39)
//   caffemodel /   : Read/WriteProtoToBinaryFile (io.cpp:49-67) of NetParameter / SolverState
//   solverstate      (src/caffe/proto/caffe.proto:51-66, 176-180)
// Field numbers, types, defaults and packedness are transcribed from the reference's .proto files
// (src/caffe/proto/caffe.proto, video_shot_sentences.proto); messages this build does not interpret
// are kept opaque (text: skipped with the braces balanced; wire: bytes preserved) so that files
// round-trip.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace caffe {
namespace pl {

enum FType { T_INT32, T_UINT32, T_INT64, T_FLOAT, T_DOUBLE, T_BOOL, T_STRING, T_BYTES, T_ENUM, T_MSG };

struct FieldDef {
  int num;
  const char* name;
  FType type;
  bool repeated;
  bool packed;
  const char* tname;   // enum or message type name
  const char* def;     // default as text ("" = type default)
};
struct EnumDef { const char* name; std::vector<std::pair<const char*, int>> values; };
struct MsgDef { const char* name; std::vector<FieldDef> fields; };

const MsgDef* FindMsg(const std::string& name);
const EnumDef* FindEnum(const std::string& name);

class Message;
struct Value {
  int64_t i = 0;
  double d = 0;
  std::string s;
  std::shared_ptr<Message> m;
};

class Message {
 public:
  explicit Message(const std::string& type);
  explicit Message(const MsgDef* def) : def_(def) {}
  Message(const Message& o) { *this = o; }
  Message& operator=(const Message& o);      // deep copy (sub-messages are cloned)
  const char* type_name() const { return def_->name; }

  // --- read access (scalar getters return the schema default when the field is absent)
  bool has(const char* f) const;
  int size(const char* f) const;
  int64_t get_int(const char* f, int idx = 0) const;
  double get_num(const char* f, int idx = 0) const;      // float / double / int as double
  bool get_bool(const char* f) const { return get_int(f) != 0; }
  const std::string& get_str(const char* f, int idx = 0) const;
  std::string get_enum(const char* f, int idx = 0) const;  // enum value NAME
  const Message& get_msg(const char* f, int idx = 0) const;  // empty default message if absent
  const std::vector<float>& floats(const char* f) const;     // packed float storage (BlobProto.data)

  // --- write access
  void clear(const char* f);
  void set_int(const char* f, int64_t v);
  void set_num(const char* f, double v);
  void set_str(const char* f, const std::string& v);
  void set_enum(const char* f, const std::string& name);
  void add_int(const char* f, int64_t v);
  void add_num(const char* f, double v);
  void add_str(const char* f, const std::string& v);
  Message* mutable_msg(const char* f);
  Message* add_msg(const char* f);
  std::vector<float>* mutable_floats(const char* f);

  // --- formats
  bool ParseText(const std::string& text, std::string* err);
  std::string PrintText(int indent = 0) const;
  bool ParseBinary(const void* data, size_t n, std::string* err);
  void SerializeBinary(std::string* out) const;

 private:
  friend class TextParser;
  const FieldDef* field(const char* f) const;
  const FieldDef* field_by_num(int num) const;
  std::vector<Value>& vals(const FieldDef* fd) { return f_[fd->num]; }
  const MsgDef* def_;
  std::map<int, std::vector<Value>> f_;
  std::map<int, std::vector<float>> packed_f_;   // repeated float fields (kept contiguous)
  std::string unknown_;                          // unknown wire fields, preserved verbatim
};

// file helpers with the reference's names (src/caffe/util/io.cpp)
bool ReadProtoFromTextFile(const std::string& filename, Message* proto);
void ReadProtoFromTextFileOrDie(const std::string& filename, Message* proto);
void WriteProtoToTextFile(const Message& proto, const std::string& filename);
bool ReadProtoFromBinaryFile(const std::string& filename, Message* proto);
void ReadProtoFromBinaryFileOrDie(const std::string& filename, Message* proto);
void WriteProtoToBinaryFile(const Message& proto, const std::string& filename);

}  // namespace pl
}  // namespace caffe
