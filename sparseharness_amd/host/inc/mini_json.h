// mini_json.h -- tiny recursive-descent JSON reader, enough for the kernel
// config schema (objects, arrays, strings with escapes, numbers, literals).
// Scalars are kept as their source text so that "512" and 512 read the same,
// which is how the reference's property-tree reader behaves
// (src/kernel_config.cpp:10-36).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace minijson {

struct Value {
  enum Kind { Null, Scalar, Array, Object } kind = Null;
  std::string text;                                   // Scalar
  std::vector<Value> items;                           // Array
  std::vector<std::pair<std::string, Value>> members; // Object (file order)

  const Value *find(const std::string &key) const {
    for (auto &m : members)
      if (m.first == key)
        return &m.second;
    return nullptr;
  }
  const Value &at(const std::string &key) const {
    const Value *v = find(key);
    if (!v)
      throw std::runtime_error("missing JSON key: " + key);
    return *v;
  }
};

class Parser {
public:
  explicit Parser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
  Value parse() {
    Value v = value();
    ws();
    if (p != end)
      fail("trailing characters");
    return v;
  }

private:
  const char *p, *end;
  [[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("JSON: ") + what); }
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  void append_utf8(std::string &out, unsigned cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  }
  std::string string() {
    if (*p != '"') fail("expected string");
    ++p;
    std::string out;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) fail("bad escape");
        switch (*p) {
        case 'n': out += '\n'; break;
        case 't': out += '\t'; break;
        case 'r': out += '\r'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'u': {
          if (end - p < 5) fail("bad \\u escape");
          append_utf8(out, (unsigned)std::stoul(std::string(p + 1, p + 5), nullptr, 16));
          p += 4;
          break;
        }
        default: out += *p; // \" \\ \/
        }
        ++p;
      } else {
        out += *p++;
      }
    }
    if (p >= end) fail("unterminated string");
    ++p;
    return out;
  }
  Value value() {
    ws();
    if (p >= end) fail("unexpected end");
    Value v;
    if (*p == '{') {
      v.kind = Value::Object;
      ++p; ws();
      if (p < end && *p == '}') { ++p; return v; }
      for (;;) {
        ws();
        std::string k = string();
        ws();
        if (p >= end || *p != ':') fail("expected ':'");
        ++p;
        v.members.emplace_back(std::move(k), value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; break; }
        fail("expected ',' or '}'");
      }
    } else if (*p == '[') {
      v.kind = Value::Array;
      ++p; ws();
      if (p < end && *p == ']') { ++p; return v; }
      for (;;) {
        v.items.push_back(value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; break; }
        fail("expected ',' or ']'");
      }
    } else if (*p == '"') {
      v.kind = Value::Scalar;
      v.text = string();
    } else {
      const char *s = p;
      while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;
      if (p == s) fail("unexpected character");
      v.text.assign(s, p);
      v.kind = (v.text == "null") ? Value::Null : Value::Scalar;
    }
    return v;
  }
};

} // namespace minijson
