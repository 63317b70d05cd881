// Host-side robustness check, built with g++ -fsanitize=address,undefined (no HIP): feeds the cascade XML reader
// mutated copies of a valid file, the .vec reader truncated / corrupted files, and the grouping / scale-plan helpers
// hostile arguments. Every call must return a status (or a result) without tripping a sanitizer.
//   usage: fuzz_host <valid cascade.xml> <valid .vec> <iterations> <tmpdir>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "cascadeclassifier_amd.h"

static std::string slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::string xml = slurp(argv[1]), vec = slurp(argv[2]);
  const int iters = std::atoi(argv[3]);
  const std::string tmp = argv[4];
  if (xml.empty() || vec.empty()) return 2;
  std::mt19937 rng(12345);
  int loaded = 0, refused = 0;
  const char* tokens[] = {"<_>", "</_>", "-1", "99999999", "1e39", "nan", "<stages>", "</stages>", "0 0 0 0", "-5 3 900 1 2.",
                          "<internalNodes>", "</leafValues>", "<maxCatCount>300</maxCatCount>", "\0", "<", ">", "&"};
  for (int it = 0; it < iters; it++) {
    std::string m = xml;
    const int edits = 1 + (int)(rng() % 4);
    for (int e = 0; e < edits; e++) {
      const size_t pos = rng() % m.size();
      switch (rng() % 6) {
        case 0: m[pos] = (char)(rng() & 0xFF); break;
        case 1: m.erase(pos, 1 + rng() % 40); break;
        case 2: m.insert(pos, tokens[rng() % (sizeof(tokens) / sizeof(tokens[0]))]); break;
        case 3: m.resize(pos); break;
        case 4: {  // swap two digits for a huge / negative number
          size_t p = m.find_first_of("0123456789", pos);
          if (p != std::string::npos) m.replace(p, 1, (rng() & 1) ? "-7" : "123456789");
          break;
        }
        default: {  // duplicate a chunk
          const size_t len = 1 + rng() % 200;
          m.insert(pos, m.substr(pos, len));
        }
      }
      if (m.empty()) m = "<";
    }
    cc_cascade* c = nullptr;
    const cc_status st = cc_cascade_load_xml_mem(m.data(), m.size(), &c);
    if (st == CC_OK) {
      loaded++;
      cc_cascade_info info;
      cc_cascade_info_get(c, &info);
      const std::string out = tmp + "/fuzz_saved.xml";
      cc_cascade_save_xml(c, out.c_str());  // may refuse (no float pre-image), must not crash
      cc_cascade_save_xml_legacy(c, out.c_str());
      cc_cascade_destroy(c);
    } else {
      refused++;
      if (c != nullptr) return 1;
    }
  }
  // .vec reader
  for (int it = 0; it < iters / 4; it++) {
    std::string m = vec.substr(0, 12 + rng() % (vec.size() - 12));
    if (rng() & 1) m[rng() % 12] = (char)(rng() & 0xFF);
    const std::string path = tmp + "/fuzz.vec";
    std::ofstream(path, std::ios::binary).write(m.data(), (std::streamsize)m.size());
    int32_t count = 0, vsize = 0;
    if (cc_vec_read(path.c_str(), &count, &vsize, nullptr, 0) == CC_OK && count > 0 && vsize > 0 && (long long)count * vsize < (1 << 26)) {
      std::vector<uint8_t> px((size_t)count * vsize);
      cc_vec_read(path.c_str(), &count, &vsize, px.data(), count);
    }
  }
  // grouping and scale plan with odd arguments
  for (int it = 0; it < 200; it++) {
    const int n = (int)(rng() % 300);
    std::vector<cc_rect> r((size_t)n), out((size_t)n + 1);
    for (auto& q : r) q = cc_rect{(int)(rng() % 4000) - 100, (int)(rng() % 4000) - 100, (int)(rng() % 500), (int)(rng() % 500)};
    int got = 0;
    cc_group_rectangles(r.data(), n, (int)(rng() % 5) - 1, (rng() % 100) / 100.0, out.data(), n + 1, &got);
    cc_detect_params p = {1.0 + (rng() % 300) / 100.0, 3, (int)(rng() % 50), (int)(rng() % 50), (int)(rng() % 3000), (int)(rng() % 3000)};
    std::vector<cc_scale_info> sc(256);
    int ns = 0;
    cc_scale_plan(3 + (int)(rng() % 60), 3 + (int)(rng() % 60), 1 + (int)(rng() % 2500), 1 + (int)(rng() % 1500), &p, sc.data(), 256, &ns);
  }
  std::printf("mutated cascades: %d loaded, %d refused\n", loaded, refused);
  return 0;
}
