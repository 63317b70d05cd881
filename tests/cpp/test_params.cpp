// CPU-only test of the parameter classes of the C++ host adaptor (ccamd/traincascade_features.hpp): the call sequence
// of the reference's trainer driver -- CvFeatureParams::create / makePtr per feature type, printDefaults, scanAttr over
// the command line (traincascade.cpp:59-149), printAttrs (cascadeclassifier.cpp:200), write inside "featureParams {"
// (cascadeclassifier.cpp:359-364) and read back from the params file (cascadeclassifier.cpp:388-401) -- and the known
// answers the reference's own tests hold for it (traincascade/test/test_features.cpp:60-125). No device is touched.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>

#include "ccamd/traincascade_features.hpp"

static int g_failed = 0, g_checked = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    g_checked++;                                                           \
    if (!(cond)) {                                                         \
      g_failed++;                                                          \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
    }                                                                      \
  } while (0)

struct CoutCapture {  // printDefaults / printAttrs write to stdout like the reference
  std::ostringstream os;
  std::streambuf* old;
  CoutCapture() : old(std::cout.rdbuf(os.rdbuf())) {}
  ~CoutCapture() { std::cout.rdbuf(old); }
};

int main() {
  // defaults and names (features.cpp:36-39, haarfeatures.cpp:12-20, lbpfeatures.cpp:9-13)
  {
    CvHaarFeatureParams h;
    CvLBPFeatureParams l;
    CHECK(h.mode == CvHaarFeatureParams::BASIC && h.maxCatCount == 0 && h.featSize == 1 && h.name == "haarFeatureParams");
    CHECK(l.maxCatCount == 256 && l.featSize == 1 && l.name == "lbpFeatureParams");
    CHECK(CvHaarFeatureParams(CvHaarFeatureParams::ALL).mode == CvHaarFeatureParams::ALL);
    CHECK(dynamic_cast<CvHaarFeatureParams*>(CvFeatureParams::create(CvFeatureParams::HAAR).get()) != nullptr);
    CHECK(dynamic_cast<CvLBPFeatureParams*>(CvFeatureParams::create(CvFeatureParams::LBP).get()) != nullptr);
    CHECK(!CvFeatureParams::create(42));  // unknown type: empty Ptr (features.cpp:62-68)
    CvParams* base = &h;                  // the driver handles them through the polymorphic base
    CHECK(base->name == HFP_NAME);
  }
  // scanAttr: the reference parses "-mode" but returns false either way (haarfeatures.cpp:70-85, test_features.cpp:96-106)
  {
    CvHaarFeatureParams p;
    CHECK(!p.scanAttr("-mode", "GARBAGE") && p.mode == -1);
    CHECK(!p.scanAttr("-mode", "ALL") && p.mode == CvHaarFeatureParams::ALL);
    CHECK(!p.scanAttr("-mode", "CORE") && p.mode == CvHaarFeatureParams::CORE);
    CHECK(!p.scanAttr("-mode", "BASIC") && p.mode == CvHaarFeatureParams::BASIC);
    CHECK(!p.scanAttr("-w", "24") && p.mode == CvHaarFeatureParams::BASIC);  // not a feature attribute: untouched
    CvLBPFeatureParams l;
    CHECK(!l.scanAttr("-mode", "ALL"));
  }
  // the driver's loop over the command line (traincascade.cpp:131-149) with the three feature-parameter objects
  {
    cv::Ptr<CvFeatureParams> fp[] = {cv::Ptr<CvFeatureParams>(new CvHaarFeatureParams), cv::Ptr<CvFeatureParams>(new CvLBPFeatureParams)};
    const char* argv[] = {"prog", "-mode", "ALL"};
    const int argc = 3;
    for (int i = 1; i < argc; i++) {
      bool set = false;
      for (int fi = 0; fi < 2; fi++) {
        set = fp[fi]->scanAttr(argv[i], argv[i + 1]);
        if (!set) {
          i++;
          break;
        }
      }
    }
    CHECK(static_cast<CvHaarFeatureParams*>(fp[0].get())->mode == CvHaarFeatureParams::ALL);
  }
  // printDefaults / printAttrs (features.cpp:28-30, haarfeatures.cpp:54-68)
  {
    CvHaarFeatureParams p(CvHaarFeatureParams::CORE);
    std::string defaults, attrs;
    {
      CoutCapture c;
      p.printDefaults();
      defaults = c.os.str();
    }
    {
      CoutCapture c;
      p.printAttrs();
      attrs = c.os.str();
    }
    CHECK(defaults == "--haarFeatureParams--\n  [-mode <BASIC(default) | CORE | ALL\n");
    CHECK(attrs == "mode: CORE\n");
    CvLBPFeatureParams l;
    {
      CoutCapture c;
      l.printDefaults();
      l.printAttrs();
      defaults = c.os.str();
    }
    CHECK(defaults == "--lbpFeatureParams--\n");
  }
  // write inside "featureParams {" and read back, as save / load of params.xml do
  for (int mode = CvHaarFeatureParams::BASIC; mode <= CvHaarFeatureParams::ALL; mode++) {
    CvHaarFeatureParams p(mode);
    cv::FileStorage fs("unused.xml", cv::FileStorage::WRITE | cv::FileStorage::MEMORY);
    fs << "params"
       << "{";
    fs << CC_FEATURE_PARAMS << "{";
    p.write(fs);
    fs << "}";
    fs << "}";
    const std::string text = fs.releaseAndGetString();
    CHECK(text.find("<featureParams>") != std::string::npos && text.find("<maxCatCount>0</maxCatCount>") != std::string::npos);
    cv::FileStorage in(text, cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(in.isOpened() || !in.root().empty());
    cv::FileNode node = in.getFirstTopLevelNode();
    CHECK(!node.empty() && node.isMap() && node.name() == "params");
    cv::Ptr<CvFeatureParams> q = CvFeatureParams::create(CvFeatureParams::HAAR);
    cv::FileNode rnode = node[CC_FEATURE_PARAMS];
    CHECK(q->read(rnode));
    CHECK(static_cast<CvHaarFeatureParams*>(q.get())->mode == mode && q->maxCatCount == 0 && q->featSize == 1);
    // init() clones (haarfeatures.cpp:22-26)
    CvHaarFeatureParams r;
    r.init(*q);
    CHECK(r.mode == mode);
  }
  {
    CvLBPFeatureParams p;
    cv::FileStorage fs("", cv::FileStorage::WRITE | cv::FileStorage::MEMORY);
    fs << CC_FEATURE_PARAMS << "{";
    p.write(fs);
    fs << "}";
    cv::FileStorage in(fs.releaseAndGetString(), cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CvLBPFeatureParams q;
    q.maxCatCount = 7;
    CHECK(q.read(in[CC_FEATURE_PARAMS]) && q.maxCatCount == 256 && q.featSize == 1);
  }
  // malformed input: read() returns false (the reference's error convention), never throws
  {
    CvHaarFeatureParams q;
    CHECK(!q.read(cv::FileNode()));  // missing node (features.cpp:55-56)
    const char* bad_mode = "<?xml version=\"1.0\"?>\n<opencv_storage><featureParams><maxCatCount>0</maxCatCount><featSize>1</featSize>"
                           "<mode>SOMETHING</mode></featureParams></opencv_storage>";
    cv::FileStorage a(bad_mode, cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(!q.read(a[CC_FEATURE_PARAMS]) && q.mode == -1);
    const char* numeric_mode = "<opencv_storage><featureParams><maxCatCount>0</maxCatCount><featSize>1</featSize><mode>2</mode>"
                               "</featureParams></opencv_storage>";
    cv::FileStorage b(numeric_mode, cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(!q.read(b[CC_FEATURE_PARAMS]));  // mode must be a string (haarfeatures.cpp:43-45)
    const char* bad_size = "<opencv_storage><featureParams><maxCatCount>0</maxCatCount><featSize>0</featSize><mode>BASIC</mode>"
                           "</featureParams></opencv_storage>";
    cv::FileStorage c(bad_size, cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(!q.read(c[CC_FEATURE_PARAMS]));  // featSize >= 1 (features.cpp:59)
    const char* no_count = "<opencv_storage><featureParams><featSize>1</featSize><mode>BASIC</mode></featureParams></opencv_storage>";
    cv::FileStorage d(no_count, cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(q.read(d[CC_FEATURE_PARAMS]) && q.maxCatCount == 0);  // a missing number reads as 0, like cv::FileNode
    cv::FileStorage e("<opencv_storage><a><b></a></opencv_storage>", cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CHECK(e.root().empty());  // not well formed
    // a stock-style header with attributes and a comment parses
    cv::FileStorage f("<?xml version=\"1.0\"?>\n<!-- c -->\n<opencv_storage>\n<cascade type_id=\"opencv-cascade-classifier\"><featureParams>"
                      "<maxCatCount>256</maxCatCount><featSize>1</featSize></featureParams></cascade>\n</opencv_storage>\n",
                      cv::FileStorage::READ | cv::FileStorage::MEMORY);
    CvLBPFeatureParams l;
    CHECK(l.read(f.getFirstTopLevelNode()[CC_FEATURE_PARAMS]) && l.maxCatCount == 256);
  }
  std::printf("%d checks, %d failed\n", g_checked, g_failed);
  return g_failed ? 1 : 0;
}
