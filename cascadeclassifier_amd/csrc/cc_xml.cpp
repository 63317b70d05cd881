// Minimal XML DOM reader for the subset of XML that cv::FileStorage emits (elements, attributes, character data,
// comments, the <?xml?> prolog and the five predefined entities). Host code; no device work here.
#include <cstring>

#include "cc_internal.h"

namespace ccamd {
namespace {

struct Parser {
  const char* p;
  const char* end;
  std::string err;
  int depth = 0;

  bool fail(const char* msg) {
    if (err.empty()) err = msg;
    return false;
  }
  void skip_ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  bool starts(const char* s) const {
    size_t n = std::strlen(s);
    return (size_t)(end - p) >= n && std::memcmp(p, s, n) == 0;
  }
  bool skip_until(const char* s) {
    size_t n = std::strlen(s);
    while ((size_t)(end - p) >= n) {
      if (std::memcmp(p, s, n) == 0) {
        p += n;
        return true;
      }
      p++;
    }
    return fail("unterminated comment / processing instruction");
  }
  // skips comments, PIs, DOCTYPE; stops at anything else
  bool skip_misc() {
    for (;;) {
      skip_ws();
      if (starts("<!--")) {
        if (!skip_until("-->")) return false;
      } else if (starts("<?")) {
        if (!skip_until("?>")) return false;
      } else if (starts("<!")) {
        if (!skip_until(">")) return false;
      } else
        return true;
    }
  }
  static bool name_char(char c) {
    return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-' || c == '.' ||
           c == ':';
  }
  bool parse_name(std::string& out) {
    const char* s = p;
    while (p < end && name_char(*p)) p++;
    if (p == s) return fail("expected a name");
    out.assign(s, p);
    return true;
  }
  static void append_decoded(std::string& out, const char* s, const char* e) {
    while (s < e) {
      if (*s == '&') {
        const char* semi = (const char*)std::memchr(s, ';', (size_t)(e - s));
        if (semi) {
          std::string ent(s + 1, semi);
          char r = 0;
          if (ent == "lt") r = '<';
          else if (ent == "gt") r = '>';
          else if (ent == "amp") r = '&';
          else if (ent == "quot") r = '"';
          else if (ent == "apos") r = '\'';
          if (r) {
            out.push_back(r);
            s = semi + 1;
            continue;
          }
        }
      }
      out.push_back(*s++);
    }
  }
  bool parse_element(XmlNode& node) {
    if (++depth > 64) return fail("nesting too deep");
    if (p >= end || *p != '<') return fail("expected '<'");
    p++;
    if (!parse_name(node.name)) return false;
    for (;;) {  // attributes
      skip_ws();
      if (p >= end) return fail("unterminated start tag");
      if (*p == '/') {
        if (p + 1 < end && p[1] == '>') {
          p += 2;
          depth--;
          return true;
        }
        return fail("malformed empty-element tag");
      }
      if (*p == '>') {
        p++;
        break;
      }
      std::string an;
      if (!parse_name(an)) return false;
      skip_ws();
      if (p >= end || *p != '=') return fail("expected '=' in attribute");
      p++;
      skip_ws();
      if (p >= end || (*p != '"' && *p != '\'')) return fail("expected quoted attribute value");
      char q = *p++;
      const char* s = p;
      while (p < end && *p != q) p++;
      if (p >= end) return fail("unterminated attribute value");
      std::string av;
      append_decoded(av, s, p);
      p++;
      node.attrs.emplace_back(an, av);
    }
    for (;;) {  // content
      const char* s = p;
      while (p < end && *p != '<') p++;
      append_decoded(node.text, s, p);
      if (p >= end) return fail("unterminated element");
      if (starts("<!--")) {
        if (!skip_until("-->")) return false;
        continue;
      }
      if (starts("<![CDATA[")) {
        p += 9;
        const char* c0 = p;
        if (!skip_until("]]>")) return false;
        node.text.append(c0, p - 3);
        continue;
      }
      if (starts("<?")) {
        if (!skip_until("?>")) return false;
        continue;
      }
      if (starts("</")) {
        p += 2;
        std::string cn;
        if (!parse_name(cn)) return false;
        if (cn != node.name) return fail("mismatched end tag");
        skip_ws();
        if (p >= end || *p != '>') return fail("malformed end tag");
        p++;
        depth--;
        return true;
      }
      node.children.emplace_back();
      if (!parse_element(node.children.back())) return false;
    }
  }
};

}  // namespace

bool xml_parse(const char* text, size_t len, XmlNode& root, std::string& err) {
  Parser ps{text, text + len, {}, 0};
  if (!ps.skip_misc()) {
    err = ps.err;
    return false;
  }
  if (ps.p >= ps.end) {
    err = "empty document";
    return false;
  }
  if (!ps.parse_element(root)) {
    err = ps.err;
    return false;
  }
  return true;
}

}  // namespace ccamd
