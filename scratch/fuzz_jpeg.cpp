// AddressSanitizer fuzz of the host half of the JPEG decoder (CPU build only; sanitizers do not run on the GPU pool):
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 scratch/fuzz_jpeg.cpp -o /tmp/fuzz_jpeg && /tmp/fuzz_jpeg a.jpg b.jpg ...
// 20,000 mutated streams (byte flips, truncations, insertions, deletions) from the given seeds, each in an exact-size heap block.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "../ovmono3d_amd/csrc/jpeg_host.hpp"
int main(int argc, char** argv) {
  std::vector<std::vector<unsigned char>> seeds;
  for (int i = 1; i < argc; ++i) { FILE* f = fopen(argv[i], "rb"); std::vector<unsigned char> b; int c; while ((c = fgetc(f)) != EOF) b.push_back((unsigned char)c); fclose(f); seeds.push_back(b); }
  const long N = getenv("FUZZ_N") ? atol(getenv("FUZZ_N")) : 20000;      // FUZZ_N / FUZZ_SEED: longer runs with other streams
  std::mt19937 g(getenv("FUZZ_SEED") ? (unsigned)atol(getenv("FUZZ_SEED")) : 1u);
  long ok = 0, bad = 0;
  for (long it = 0; it < N; ++it) {
    std::vector<unsigned char> d = seeds[it % seeds.size()];
    if (d.size() > 20000) d.resize(20000);
    int kind = g() % 4;
    if (kind == 0) { int k = 1 + g() % 6; for (int j = 0; j < k; ++j) d[2 + g() % (d.size() - 2)] = (unsigned char)g(); }
    else if (kind == 1) d.resize(2 + g() % (d.size() - 2));
    else if (kind == 2) { size_t i = 2 + g() % (d.size() - 2); int k = 1 + g() % 40; std::vector<unsigned char> ins(k); for (auto& x : ins) x = (unsigned char)g(); d.insert(d.begin() + i, ins.begin(), ins.end()); }
    else { size_t i = 2 + g() % (d.size() - 10); d.erase(d.begin() + i, d.begin() + i + 1 + g() % 8); }
    // exact-size heap copy so that ASan sees any over-read
    unsigned char* buf = (unsigned char*)malloc(d.size()); memcpy(buf, d.data(), d.size());
    OvmJpegInfo info;
    int rc = ovm_jpeg::host_info(buf, d.size(), &info);
    if (rc == 0 && info.coef_blocks <= (1 << 18)) {
      int16_t* coef = (int16_t*)malloc((size_t)info.coef_blocks * 128);
      rc = ovm_jpeg::host_entropy_decode(buf, d.size(), coef, (int64_t)info.coef_blocks * 64, &info);
      free(coef);
    }
    if (rc == 0) ++ok; else ++bad;
    free(buf);
  }
  printf("ok %ld refused %ld\n", ok, bad);
  return 0;
}
