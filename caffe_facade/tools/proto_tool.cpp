// proto_tool -- exercises proto_lite from the command line (used by the CPU tests):
//   proto_tool text2bin <MessageType> <in.prototxt> <out.bin>
//   proto_tool bin2text <MessageType> <in.bin> <out.prototxt>
//   proto_tool filter   <in net.prototxt> <TRAIN|TEST> <out.prototxt>   (Net::FilterNet)
//   proto_tool lmdbdump <db dir> <out.txt>          one line per record: "<key> <value bytes> <fnv1a64 of value>"
//   proto_tool dbload   <db dir> <shots|windows> <out.txt>   dataset summary + feature checksum after record decode
#include <cstdio>
#include <cstring>
#include <fstream>

#include "caffe/lmdb_reader.hpp"
#include "caffe/net.hpp"

using namespace caffe;

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: see source\n"); return 2; }
  const std::string cmd = argv[1];
  if (cmd == "text2bin") {
    pl::Message m((std::string(argv[2])));
    pl::ReadProtoFromTextFileOrDie(argv[3], &m);
    pl::WriteProtoToBinaryFile(m, argv[4]);
  } else if (cmd == "bin2text") {
    pl::Message m((std::string(argv[2])));
    pl::ReadProtoFromBinaryFileOrDie(argv[3], &m);
    pl::WriteProtoToTextFile(m, argv[4]);
  } else if (cmd == "filter") {
    NetParameter in("NetParameter"), out("NetParameter");
    pl::ReadProtoFromTextFileOrDie(argv[2], &in);
    in.mutable_msg("state")->set_enum("phase", argv[3]);
    Net<float>::FilterNet(in, &out);
    pl::WriteProtoToTextFile(out, argv[4]);
  } else if (cmd == "lmdbdump") {
    LmdbReader db; std::string err;
    if (!db.Open(argv[2], &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE* o = fopen(argv[3], "w");
    fprintf(o, "entries %zu\n", db.entries());
    const bool ok = db.ForEach([&](const std::string& k, const std::string& v) {
      uint64_t h = 1469598103934665603ull;
      for (unsigned char c : v) { h ^= c; h *= 1099511628211ull; }
      fprintf(o, "%s %zu %016llx\n", k.c_str(), v.size(), (unsigned long long)h);
    }, &err);
    fclose(o);
    if (!ok) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  } else if (cmd == "dbload") {
    if (argc < 5) return 2;
    const bool win = !strcmp(argv[3], "windows");
    shared_ptr<VideoDataset> ds = VideoDataset::Open(argv[2], win ? VideoDataset::kTestWindows : VideoDataset::kShots, "LMDB");
    FILE* o = fopen(argv[4], "w");
    double sum = 0; for (float f : ds->features) sum += f;
    fprintf(o, "rows %lld F %d videos %zu windows %zu k %d pos %d neg %d sum %.6f\n", (long long)ds->n_rows, ds->F, ds->video_id.size(),
            ds->win_video_id.size(), ds->win_k, ds->win_pos, ds->win_neg, sum);
    for (size_t v = 0; v < ds->video_id.size(); ++v) {
      fprintf(o, "video %d n %d base %lld ids", ds->video_id[v], ds->n_shots[v], (long long)ds->row_base[v]);
      for (int j = 0; j < ds->n_shots[v]; ++j) fprintf(o, " %d", ds->shot_ids[ds->row_base[v] + j]);
      fprintf(o, "\n");
    }
    const size_t stride = (size_t)ds->win_k + ds->win_pos + ds->win_neg;
    for (size_t w = 0; w < ds->win_video_id.size(); ++w) {
      fprintf(o, "window %d row0 %d", ds->win_video_id[w], ds->win_rows[w * stride]);
      if (stride > (size_t)ds->win_k) {           // first feature of every row, in channel order
        fprintf(o, " first");
        for (size_t j = 0; j < stride; ++j) fprintf(o, " %g", ds->features[(size_t)ds->win_rows[w * stride + j] * ds->F]);
      }
      fprintf(o, "\n");
    }
    fclose(o);
  } else return 2;
  return 0;
}
