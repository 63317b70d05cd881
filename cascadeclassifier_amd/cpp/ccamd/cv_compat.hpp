// Minimal stand-in for the few OpenCV core types the reference's evaluator / detector interfaces mention
// (cv::Mat as a 2-D view, Size, Rect, Ptr, Exception / CV_Assert, a write-only XML FileStorage).
// It exists ONLY so that the C++ adaptor keeps the reference's signatures where OpenCV is not installed
// (this image has none). With a real OpenCV, define CCAMD_USE_OPENCV and the real headers are used instead.
#pragma once

#ifdef CCAMD_USE_OPENCV
#include <opencv2/core.hpp>
#else

#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_32SC1 CV_32S
#define CV_32FC1 CV_32F

namespace cv {

class Exception : public std::runtime_error {
 public:
  Exception(int code_, const std::string& msg) : std::runtime_error(msg), code(code_) {}
  int code;
};

template <class T>
using Ptr = std::shared_ptr<T>;
typedef std::string String;

struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};

struct Rect {
  int x = 0, y = 0, width = 0, height = 0;
  Rect() {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
  bool operator==(const Rect& o) const { return x == o.x && y == o.y && width == o.width && height == o.height; }
};

struct Scalar {
  double v[4];
  Scalar(double a = 0) : v{a, 0, 0, 0} {}
};

// Dense 2-D matrix / view: owns its buffer (shared) or wraps external memory.
class Mat {
 public:
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  size_t step = 0;  // bytes per row

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) {
    create(r, c, type);
    setTo(s);
  }
  Mat(Size sz, int type) { create(sz.height, sz.width, type); }
  Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), data((uchar*)ext), type_(type) {
    step = step_ ? step_ : (size_t)c * elemSize();
  }
  void create(int r, int c, int type) {
    rows = r;
    cols = c;
    type_ = type;
    step = (size_t)c * elemSize();
    buf_ = std::shared_ptr<uchar>(new uchar[std::max<size_t>(step * r, 1)](), std::default_delete<uchar[]>());
    data = buf_.get();
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  size_t elemSize() const { return type_ == CV_8U ? 1 : 4; }
  size_t step1() const { return step / elemSize(); }
  Size size() const { return Size(cols, rows); }
  template <class T>
  T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
  template <class T>
  const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
  template <class T>
  T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T>
  const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  Mat& setTo(const Scalar& s) {
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) {
        if (type_ == CV_8U) at<uchar>(r, c) = (uchar)s.v[0];
        else if (type_ == CV_32S) at<int>(r, c) = (int)s.v[0];
        else at<float>(r, c) = (float)s.v[0];
      }
    return *this;
  }
  Mat operator()(const Rect& r) const {  // view into the same buffer
    Mat m;
    m.rows = r.height;
    m.cols = r.width;
    m.type_ = type_;
    m.step = step;
    m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize();
    m.buf_ = buf_;
    return m;
  }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }

 private:
  int type_ = CV_8U;
  std::shared_ptr<uchar> buf_;
};

// Read side: one element of a parsed <opencv_storage> document. Covers what the reference's parameter readers use
// (features.cpp:53-60, haarfeatures.cpp:38-52, cascadeclassifier.cpp:388-401): empty(), operator[], isString(),
// conversion to int / float / double / std::string, operator>>, and iteration over children.
class FileNode {
 public:
  struct Elem {
    std::string name, text;
    std::vector<std::shared_ptr<Elem>> kids;
  };
  FileNode() {}
  explicit FileNode(std::shared_ptr<Elem> e) : e_(std::move(e)) {}
  bool empty() const { return !e_; }
  bool isNone() const { return !e_; }
  bool isMap() const { return e_ && !e_->kids.empty() && e_->kids[0]->name != "_"; }
  bool isSeq() const { return e_ && !e_->kids.empty() && e_->kids[0]->name == "_"; }
  bool isInt() const;
  bool isReal() const;
  bool isString() const { return e_ && e_->kids.empty() && !text().empty() && !isInt() && !isReal(); }
  size_t size() const { return e_ ? e_->kids.size() : 0; }
  std::string name() const { return e_ ? e_->name : std::string(); }
  FileNode operator[](const std::string& key) const;
  FileNode operator[](const char* key) const { return (*this)[std::string(key)]; }
  FileNode operator[](int i) const { return e_ && i >= 0 && (size_t)i < e_->kids.size() ? FileNode(e_->kids[(size_t)i]) : FileNode(); }
  operator int() const;
  operator float() const { return (float)(double)*this; }
  operator double() const;
  operator std::string() const { return text(); }
  std::string text() const;  // trimmed character data

 private:
  std::shared_ptr<Elem> e_;
};
inline void operator>>(const FileNode& n, std::string& v) { v = (std::string)n; }
inline void operator>>(const FileNode& n, int& v) { v = (int)n; }
inline void operator>>(const FileNode& n, float& v) { v = (float)n; }
inline void operator>>(const FileNode& n, double& v) { v = (double)n; }

// XML FileStorage: the `<<` streaming protocol the reference's writers use (names, scalars, "{" "}" "[" "]" "[:" "{:"),
// producing the <opencv_storage> layout OpenCV reads back; opened with READ it parses such a document (file, or the text
// itself with READ | MEMORY) into FileNodes.
class FileStorage {
 public:
  enum { READ = 0, WRITE = 1, MEMORY = 4 };
  FileStorage() {}
  FileStorage(const std::string& filename, int flags) { open(filename, flags); }
  ~FileStorage() { release(); }
  bool open(const std::string& filename, int flags);
  bool isOpened() const { return opened_; }
  void release();
  std::string releaseAndGetString();
  FileNode root() const { return FileNode(root_); }                // the <opencv_storage> element
  FileNode getFirstTopLevelNode() const { return root()[0]; }
  FileNode operator[](const std::string& key) const { return root()[key]; }
  FileNode operator[](const char* key) const { return root()[std::string(key)]; }
  FileStorage& put(const std::string& s);
  FileStorage& putNumber(const std::string& text);

 private:
  struct Level {
    bool is_map;
    bool flow;
    std::string tag;
  };
  void element_open(const std::string& tag);
  void indent();
  std::shared_ptr<FileNode::Elem> root_;  // READ mode
  std::string filename_;
  std::ostringstream out_;
  std::vector<Level> stack_;
  std::string pending_key_;
  bool have_key_ = false, opened_ = false, memory_ = false, line_open_ = false;
};
FileStorage& operator<<(FileStorage& fs, const std::string& s);
FileStorage& operator<<(FileStorage& fs, const char* s);
FileStorage& operator<<(FileStorage& fs, int v);
FileStorage& operator<<(FileStorage& fs, bool v);
FileStorage& operator<<(FileStorage& fs, float v);
FileStorage& operator<<(FileStorage& fs, double v);

}  // namespace cv

#define CV_StsAssert -215
#define CV_Assert(expr)                                                                             \
  do {                                                                                              \
    if (!(expr)) throw cv::Exception(CV_StsAssert, std::string("Assertion failed: ") + #expr);     \
  } while (0)

#endif  // CCAMD_USE_OPENCV
