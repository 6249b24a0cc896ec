// Minimal JSON reader for the scene / config files (the reference uses
// serde_json, /root/reference/src/render/integrator.rs:63-82).  Supports the
// full JSON value grammar; numbers are parsed with strtod so that every f64 in
// a scene file round-trips exactly as serde_json reads it.
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {

class Json {
 public:
  enum Kind { Null, Bool, Number, String, Array, Object };

  Kind kind = Null;
  bool boolean = false;
  double number = 0.0;
  std::string string;
  std::vector<Json> array;
  // Insertion-ordered: externally tagged serde enums have exactly one key and
  // its name is the variant.
  std::vector<std::pair<std::string, Json>> object;

  bool is_null() const { return kind == Null; }
  bool has(const std::string &key) const { return find(key) != nullptr; }

  const Json *find(const std::string &key) const {
    if (kind != Object) return nullptr;
    for (const auto &kv : object)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const Json &at(const std::string &key) const {
    const Json *j = find(key);
    if (!j) throw std::runtime_error("missing field `" + key + "`");
    return *j;
  }
  double as_number() const {
    if (kind != Number) throw std::runtime_error("expected a number");
    return number;
  }
  const std::string &as_string() const {
    if (kind != String) throw std::runtime_error("expected a string");
    return string;
  }

  static Json parse(const std::string &text) {
    Parser p{text, 0};
    Json v = p.value();
    p.skip_ws();
    if (p.pos != text.size()) p.fail("trailing characters");
    return v;
  }

 private:
  struct Parser {
    const std::string &s;
    size_t pos;

    [[noreturn]] void fail(const std::string &what) const {
      throw std::runtime_error("JSON: " + what + " at byte " + std::to_string(pos));
    }
    void skip_ws() {
      while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) pos++;
    }
    char peek() {
      skip_ws();
      if (pos >= s.size()) fail("unexpected end");
      return s[pos];
    }
    void expect(char c) {
      if (peek() != c) fail(std::string("expected '") + c + "'");
      pos++;
    }
    bool literal(const char *lit) {
      size_t n = 0;
      while (lit[n]) n++;
      if (s.compare(pos, n, lit) == 0) {
        pos += n;
        return true;
      }
      return false;
    }
    std::string parse_string() {
      expect('"');
      std::string out;
      while (true) {
        if (pos >= s.size()) fail("unterminated string");
        char c = s[pos++];
        if (c == '"') break;
        if (c == '\\') {
          if (pos >= s.size()) fail("bad escape");
          char e = s[pos++];
          switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
              if (pos + 4 > s.size()) fail("bad \\u escape");
              unsigned cp = (unsigned)std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
              pos += 4;
              if (cp < 0x80) {
                out += (char)cp;
              } else if (cp < 0x800) {
                out += (char)(0xC0 | (cp >> 6));
                out += (char)(0x80 | (cp & 0x3F));
              } else {
                out += (char)(0xE0 | (cp >> 12));
                out += (char)(0x80 | ((cp >> 6) & 0x3F));
                out += (char)(0x80 | (cp & 0x3F));
              }
              break;
            }
            default: fail("bad escape");
          }
        } else {
          out += c;
        }
      }
      return out;
    }
    Json value() {
      char c = peek();
      Json v;
      if (c == '{') {
        pos++;
        v.kind = Object;
        if (peek() == '}') {
          pos++;
          return v;
        }
        while (true) {
          skip_ws();
          std::string key = parse_string();
          expect(':');
          v.object.emplace_back(std::move(key), value());
          char d = peek();
          pos++;
          if (d == '}') break;
          if (d != ',') fail("expected ',' or '}'");
        }
      } else if (c == '[') {
        pos++;
        v.kind = Array;
        if (peek() == ']') {
          pos++;
          return v;
        }
        while (true) {
          v.array.push_back(value());
          char d = peek();
          pos++;
          if (d == ']') break;
          if (d != ',') fail("expected ',' or ']'");
        }
      } else if (c == '"') {
        v.kind = String;
        v.string = parse_string();
      } else if (literal("true")) {
        v.kind = Bool;
        v.boolean = true;
      } else if (literal("false")) {
        v.kind = Bool;
      } else if (literal("null")) {
        v.kind = Null;
      } else {
        const char *begin = s.c_str() + pos;
        char *end = nullptr;
        double d = std::strtod(begin, &end);
        if (end == begin) fail("unexpected character");
        pos += (size_t)(end - begin);
        v.kind = Number;
        v.number = d;
      }
      return v;
    }
  };
};

}  // namespace rt
