// Minimal JSON reader for the scene front end (objects keep key order).
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rth {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;

    bool is_null() const { return kind == Null; }
    bool is_number() const { return kind == Number; }
    bool is_string() const { return kind == String; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    const JsonValue *find(const char *key) const {
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &text) : s_(text), p_(0) {}

    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t p_;

    [[noreturn]] void fail(const char *what) const {
        size_t line = 1, col = 1;
        for (size_t i = 0; i < p_ && i < s_.size(); ++i) {
            if (s_[i] == '\n') {
                ++line;
                col = 1;
            } else {
                ++col;
            }
        }
        throw std::runtime_error(std::string("JSON: ") + what + " at line " + std::to_string(line) +
                                 " column " + std::to_string(col));
    }
    void ws() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_;
    }
    bool eat(char c) {
        ws();
        if (p_ < s_.size() && s_[p_] == c) {
            ++p_;
            return true;
        }
        return false;
    }
    JsonValue value() {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        char c = s_[p_];
        JsonValue v;
        if (c == '{') {
            ++p_;
            v.kind = JsonValue::Object;
            if (eat('}')) return v;
            do {
                ws();
                if (p_ >= s_.size() || s_[p_] != '"') fail("expected key string");
                std::string k = string();
                if (!eat(':')) fail("expected ':'");
                v.obj.emplace_back(std::move(k), value());
            } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            ++p_;
            v.kind = JsonValue::Array;
            if (eat(']')) return v;
            do {
                v.arr.push_back(value());
            } while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') {
            v.kind = JsonValue::String;
            v.str = string();
        } else if (s_.compare(p_, 4, "true") == 0) {
            v.kind = JsonValue::Bool;
            v.b = true;
            p_ += 4;
        } else if (s_.compare(p_, 5, "false") == 0) {
            v.kind = JsonValue::Bool;
            p_ += 5;
        } else if (s_.compare(p_, 4, "null") == 0) {
            p_ += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char *start = s_.c_str() + p_;
            char *end = nullptr;
            v.kind = JsonValue::Number;
            v.num = std::strtod(start, &end);  // correctly rounded, as serde_json's f64 path
            if (end == start) fail("bad number");
            p_ += static_cast<size_t>(end - start);
        } else {
            fail("unexpected character");
        }
        return v;
    }
    std::string string() {
        std::string out;
        ++p_;  // opening quote
        while (true) {
            if (p_ >= s_.size()) fail("unterminated string");
            char c = s_[p_++];
            if (c == '"') break;
            if (c == '\\') {
                if (p_ >= s_.size()) fail("bad escape");
                char e = s_[p_++];
                switch (e) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (p_ + 4 > s_.size()) fail("bad \\u escape");
                        unsigned cp = static_cast<unsigned>(std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16));
                        p_ += 4;
                        if (cp < 0x80) {
                            out.push_back(static_cast<char>(cp));
                        } else if (cp < 0x800) {
                            out.push_back(static_cast<char>(0xC0 | (cp >> 6)));
                            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        } else {
                            out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
                            out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
                            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        }
                        break;
                    }
                    default: out.push_back(e); break;  // \" \\ \/
                }
            } else {
                out.push_back(c);
            }
        }
        return out;
    }
};

}  // namespace rth
