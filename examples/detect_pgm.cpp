// The reference's detection tool (tools/detection/Cpp/main.cpp) on the MI355X library, minus the GUI: reads a binary
// PGM (P5) instead of cv::imread, runs the cascade with the tool's parameters (scaleFactor 4, minNeighbors 50 unless
// overridden) and prints one "x y w h" line per detection.
//   usage: detect_pgm <cascade.xml> <image.pgm> [scaleFactor=4] [minNeighbors=50]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "ccamd/traincascade_features.hpp"

static bool read_pgm(const char* path, cv::Mat& gray) {
  std::ifstream f(path, std::ios::binary);
  std::string magic;
  int w = 0, h = 0, maxv = 0;
  f >> magic;
  auto skip = [&]() {
    while (f >> std::ws && f.peek() == '#') f.ignore(1 << 20, '\n');
  };
  skip();
  f >> w;
  skip();
  f >> h;
  skip();
  f >> maxv;
  f.get();
  if (!f || magic != "P5" || w < 1 || h < 1 || maxv != 255) return false;
  gray = cv::Mat(h, w, CV_8UC1);
  f.read(reinterpret_cast<char*>(gray.data), (std::streamsize)w * h);
  return (bool)f;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <cascade.xml> <image.pgm> [scaleFactor=4] [minNeighbors=50]\n", argv[0]);
    return 2;
  }
  ccamd::CascadeClassifier cascade(argv[1]);  // main.cpp:42
  if (cascade.empty()) {
    std::fprintf(stderr, "cannot load cascade: %s\n", cascade.lastError().c_str());
    return 1;
  }
  cv::Mat gray;
  if (!read_pgm(argv[2], gray)) {
    std::fprintf(stderr, "cannot read %s (binary 8-bit PGM expected)\n", argv[2]);
    return 1;
  }
  const double scaleFactor = argc > 3 ? std::atof(argv[3]) : 4.0;
  const int minNeighbors = argc > 4 ? std::atoi(argv[4]) : 50;
  std::vector<cv::Rect> objects;
  try {
    cascade.detectMultiScale(gray, objects, scaleFactor, minNeighbors);  // main.cpp:45
  } catch (const cv::Exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  for (const cv::Rect& r : objects) std::printf("%d %d %d %d\n", r.x, r.y, r.width, r.height);
  return 0;
}
