// stdlib_pin -- prints sequences produced by the real C / C++ standard library of this machine
// (glibc rand(), libstdc++ std::random_shuffle, std::sort), the third-party algorithms the
// reference's sampler delegates to (video_sampled_shots_data_layer.cpp:27,29,306,437,482;
// include/caffe/util/rng.hpp:43-54).  Test infrastructure only.
//
// usage: stdlib_pin rand N            -> N values of rand() from the never-seeded stream
//        stdlib_pin script a,b,c,...  -> for each positive n: random_unique-style partial shuffle
//                                        of iota(n) taking min(n,5), sort of the head, then
//                                        std::random_shuffle of the tail; prints the permutation.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  if (!strcmp(argv[1], "rand")) {
    long n = atol(argv[2]);
    for (long i = 0; i < n; ++i) printf("%d\n", rand());
    return 0;
  }
  if (!strcmp(argv[1], "script")) {
    char* tok = strtok(argv[2], ",");
    while (tok) {
      int n = atoi(tok);
      std::vector<int> v(n);
      std::iota(v.begin(), v.end(), 0);
      int take = std::min(n, 5), left = n;
      std::vector<int>::iterator first = v.begin();
      for (int t = 0; t < take; ++t) {           // same draw pattern as random_unique
        std::vector<int>::iterator r = first + rand() % left;
        std::swap(*first, *r);
        ++first; --left;
      }
      std::sort(v.begin(), v.begin() + take);
      std::random_shuffle(v.begin() + take, v.end());
      for (int i = 0; i < n; ++i) printf("%d%c", v[i], i + 1 == n ? '\n' : ' ');
      tok = strtok(NULL, ",");
    }
    return 0;
  }
  return 2;
}
