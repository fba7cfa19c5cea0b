#include "s72_scene.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace {

// A small JSON document model: enough for the scene array's objects.
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    const Value* find(const std::string& key) const {
        for (auto const& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class Parser {
  public:
    explicit Parser(const std::string& text) : s(text) {}

    Value document() {
        Value v = value();
        skip();
        if (pos != s.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string& s;
    size_t pos = 0;

    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error("Scene loading error: JSON " + std::string(what) + " at byte " + std::to_string(pos));
    }
    void skip() {
        while (pos < s.size() && std::isspace((unsigned char)s[pos])) pos++;
    }
    bool eat(char c) {
        skip();
        if (pos < s.size() && s[pos] == c) { pos++; return true; }
        return false;
    }
    void literal(const char* word) {
        for (const char* p = word; *p; p++, pos++)
            if (pos >= s.size() || s[pos] != *p) fail("bad literal");
    }
    static void utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xc0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3f)); }
        else { out += (char)(0xe0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3f)); out += (char)(0x80 | (cp & 0x3f)); }
    }
    std::string string() {
        if (!eat('"')) fail("expected string");
        std::string out;
        while (true) {
            if (pos >= s.size()) fail("unterminated string");
            char c = s[pos++];
            if (c == '"') return out;
            if (c != '\\') { out += c; continue; }
            if (pos >= s.size()) fail("unterminated escape");
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
                    if (pos + 4 > s.size()) fail("short \\u escape");
                    unsigned cp = (unsigned)std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
                    pos += 4;
                    utf8(out, cp);
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    Value value() {
        skip();
        if (pos >= s.size()) fail("unexpected end");
        Value v;
        char c = s[pos];
        if (c == '{') {
            pos++;
            v.kind = Value::Object;
            if (eat('}')) return v;
            do {
                skip();
                std::string key = string();
                if (!eat(':')) fail("expected ':'");
                v.obj.emplace_back(std::move(key), value());
            } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            pos++;
            v.kind = Value::Array;
            if (eat(']')) return v;
            do v.arr.push_back(value()); while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = string();
        } else if (c == 't') { literal("true"); v.kind = Value::Bool; v.b = true; }
        else if (c == 'f') { literal("false"); v.kind = Value::Bool; }
        else if (c == 'n') { literal("null"); }
        else {
            char* end = nullptr;
            v.num = std::strtod(s.c_str() + pos, &end);
            if (end == s.c_str() + pos) fail("unexpected character");
            pos = (size_t)(end - s.c_str());
            v.kind = Value::Number;
        }
        return v;
    }
};

}  // namespace

std::vector<SceneEnvironment> scene_environments(const std::filesystem::path& scene_file) {
    std::ifstream in(scene_file, std::ios::binary);
    if (!in) throw std::runtime_error("Scene loading error: cannot open " + scene_file.string());
    std::stringstream buf;
    buf << in.rdbuf();
    const std::string text = buf.str();
    Value doc = Parser(text).document();
    // scene_loader.cpp:56-61
    if (doc.kind != Value::Array) throw std::runtime_error("Scene loading error: scene JSON is not an array");
    if (doc.arr.empty() || doc.arr[0].kind != Value::String || doc.arr[0].str != "s72-v1")
        throw std::runtime_error("Scene loading error: first element of scene JSON is not correct magic string");

    const std::filesystem::path directory = std::filesystem::absolute(scene_file).parent_path();
    std::vector<SceneEnvironment> out;
    for (size_t id = 1; id < doc.arr.size(); id++) {
        const Value& o = doc.arr[id];
        if (o.kind != Value::Object) continue;
        const Value* type = o.find("type");
        if (!type || type->kind != Value::String || type->str != "ENVIRONMENT") continue;
        const uint32_t number = (uint32_t)out.size();          // scene_loader.cpp:113-116
        SceneEnvironment env;
        const Value* name = o.find("name");
        env.name = (name && name->kind == Value::String) ? name->str : std::string();
        env.local = false;
        if (const Value* rad = o.find("radiance")) {
            // scene_loader.cpp:526-530 with loadTextureObj, :361-379
            if (rad->kind != Value::Object)
                throw std::runtime_error("Scene loading error: environment does not contain a radiance map");
            const Value* src = rad->find("src");
            if (!src || src->kind != Value::String)
                throw std::runtime_error("Scene loading error: texture does not have a source");
            const Value* ttype = rad->find("type");
            const std::string texture_type = (ttype && ttype->kind == Value::String) ? ttype->str : "2D";
            if (texture_type != "cube") throw std::runtime_error("Scene loading error: texture has an unexpected type");
            env.file = directory / src->str;
        } else if (o.find("local")) {
            // scene_loader.cpp:532-534
            env.local = true;
            env.file = std::filesystem::absolute(scene_file);
            env.file.replace_extension(".localIBL_" + std::to_string(number) + ".png");
        } else {
            out.push_back(env);    // keeps the numbering; nothing to prefilter
            continue;
        }
        out.push_back(env);
    }
    return out;
}
