// proto_tool -- exercises proto_lite from the command line (used by the CPU tests):
//   proto_tool text2bin <MessageType> <in.prototxt> <out.bin>
//   proto_tool bin2text <MessageType> <in.bin> <out.prototxt>
//   proto_tool filter   <in net.prototxt> <TRAIN|TEST> <out.prototxt>   (Net::FilterNet)
#include <cstdio>
#include <cstring>
#include <fstream>

#include "caffe/net.hpp"

using namespace caffe;

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: see source\n"); return 2; }
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
  } else return 2;
  return 0;
}
