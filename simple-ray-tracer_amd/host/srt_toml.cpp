/* srt_toml.cpp -- see srt_toml.h. */
#include "srt_toml.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace srt::toml {

static const Value kNone {};

const Value& Value::operator[](const std::string& key) const
{
    if (kind != TableKind)
        return kNone;
    const auto it = tab->find(key);
    return it == tab->end() ? kNone : it->second;
}

const Value& Value::at(size_t index) const
{
    if (kind != ArrayKind || index >= arr->size())
        return kNone;
    return (*arr)[index];
}

namespace {

Value makeTable()
{
    Value v;
    v.kind = Value::TableKind;
    v.tab = std::make_shared<Table>();
    return v;
}

Value makeArray()
{
    Value v;
    v.kind = Value::ArrayKind;
    v.arr = std::make_shared<Array>();
    return v;
}

class Parser {
public:
    explicit Parser(const std::string& text)
        : m_text(text)
    {
    }

    Value run()
    {
        Value root = makeTable();
        Value* current = &root;
        for (;;) {
            skipBlank(true);
            if (eof())
                break;
            if (peek() == '[') {
                const bool isArray = m_pos + 1 < m_text.size() && m_text[m_pos + 1] == '[';
                m_pos += isArray ? 2 : 1;
                skipBlank(false);
                const std::vector<std::string> path = keyPath();
                skipBlank(false);
                expect(']');
                if (isArray)
                    expect(']');
                current = isArray ? openArrayTable(root, path) : openTable(root, path);
                endOfLine();
            } else {
                const std::vector<std::string> path = keyPath();
                skipBlank(false);
                expect('=');
                skipBlank(false);
                Value v = value();
                assign(*current, path, std::move(v));
                endOfLine();
            }
        }
        return root;
    }

private:
    const std::string& m_text;
    size_t m_pos = 0;
    int m_line = 1;

    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error("line " + std::to_string(m_line) + ": " + what); }
    [[nodiscard]] bool eof() const { return m_pos >= m_text.size(); }
    [[nodiscard]] char peek() const { return eof() ? '\0' : m_text[m_pos]; }
    char next()
    {
        if (eof())
            fail("unexpected end of file");
        const char c = m_text[m_pos++];
        if (c == '\n')
            ++m_line;
        return c;
    }
    void expect(char c)
    {
        if (peek() != c)
            fail(std::string("expected '") + c + "'");
        next();
    }
    /* spaces and tabs; with newlines also line ends and comments */
    void skipBlank(bool newlines)
    {
        for (;;) {
            const char c = peek();
            if (c == ' ' || c == '\t' || (newlines && (c == '\n' || c == '\r'))) {
                next();
            } else if (c == '#' && newlines) {
                while (!eof() && peek() != '\n')
                    next();
            } else {
                return;
            }
        }
    }
    void endOfLine()
    {
        skipBlank(false);
        if (peek() == '#')
            while (!eof() && peek() != '\n')
                next();
        if (peek() == '\r')
            next();
        if (!eof() && peek() != '\n')
            fail("unexpected text after a value");
    }

    static bool bareKeyChar(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-'; }

    std::vector<std::string> keyPath()
    {
        std::vector<std::string> path;
        for (;;) {
            skipBlank(false);
            if (peek() == '"') {
                path.push_back(basicString());
            } else if (peek() == '\'') {
                path.push_back(literalString());
            } else {
                std::string k;
                while (bareKeyChar(peek()))
                    k.push_back(next());
                if (k.empty())
                    fail("expected a key");
                path.push_back(k);
            }
            skipBlank(false);
            if (peek() != '.')
                return path;
            next();
        }
    }

    /* Walks (creating) the tables of a dotted path; the last element of an array of tables stands for the array. */
    Value* descend(Value* at, const std::string& key, bool forHeader)
    {
        auto [it, created] = at->tab->try_emplace(key);
        if (created)
            it->second = makeTable();
        Value* v = &it->second;
        if (v->kind == Value::ArrayKind && !v->sealed && !v->arr->empty() && v->arr->back().kind == Value::TableKind)
            v = &v->arr->back();
        if (v->kind != Value::TableKind || (v->sealed && forHeader))
            fail("key '" + key + "' is already a value");
        return v;
    }

    Value* openTable(Value& root, const std::vector<std::string>& path)
    {
        Value* at = &root;
        for (const std::string& k : path)
            at = descend(at, k, true);
        return at;
    }

    Value* openArrayTable(Value& root, const std::vector<std::string>& path)
    {
        Value* at = &root;
        for (size_t i = 0; i + 1 < path.size(); ++i)
            at = descend(at, path[i], true);
        auto [it, created] = at->tab->try_emplace(path.back());
        if (created)
            it->second = makeArray();
        if (it->second.kind != Value::ArrayKind || it->second.sealed)
            fail("'" + path.back() + "' is not an array of tables");
        it->second.arr->push_back(makeTable());
        return &it->second.arr->back();
    }

    void assign(Value& table, const std::vector<std::string>& path, Value v)
    {
        Value* at = &table;
        for (size_t i = 0; i + 1 < path.size(); ++i)
            at = descend(at, path[i], false);
        if (!at->tab->try_emplace(path.back(), std::move(v)).second)
            fail("key '" + path.back() + "' is defined twice");
    }

    void appendUtf8(std::string& out, uint32_t cp)
    {
        if (cp < 0x80) {
            out.push_back((char)cp);
        } else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }

    void escape(std::string& out, bool multiline)
    {
        const char c = next();
        switch (c) {
        case 'b': out.push_back('\b'); break;
        case 't': out.push_back('\t'); break;
        case 'n': out.push_back('\n'); break;
        case 'f': out.push_back('\f'); break;
        case 'r': out.push_back('\r'); break;
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case 'u':
        case 'U': {
            uint32_t cp = 0;
            for (int k = 0; k < (c == 'u' ? 4 : 8); ++k) {
                const char h = next();
                if (!std::isxdigit((unsigned char)h))
                    fail("bad unicode escape");
                cp = cp * 16 + (uint32_t)(std::isdigit((unsigned char)h) ? h - '0' : (std::tolower(h) - 'a' + 10));
            }
            appendUtf8(out, cp);
            break;
        }
        case ' ':
        case '\t':
        case '\r':
        case '\n':
            if (!multiline)
                fail("bad escape");
            /* line-ending backslash: trim up to the next non-blank character */
            while (peek() == ' ' || peek() == '\t' || peek() == '\r' || peek() == '\n')
                next();
            break;
        default:
            fail("bad escape");
        }
    }

    bool startsWith(const char* lit) const { return m_text.compare(m_pos, std::char_traits<char>::length(lit), lit) == 0; }

    std::string basicString()
    {
        std::string out;
        if (startsWith("\"\"\"")) {
            m_pos += 3;
            if (peek() == '\r')
                next();
            if (peek() == '\n')
                next();
            for (;;) {
                if (startsWith("\"\"\"")) {
                    m_pos += 3;
                    while (peek() == '"') /* up to two quotes may precede the delimiter */
                        out.push_back(next());
                    return out;
                }
                const char c = next();
                if (c == '\\')
                    escape(out, true);
                else
                    out.push_back(c);
            }
        }
        expect('"');
        for (;;) {
            const char c = next();
            if (c == '"')
                return out;
            if (c == '\n')
                fail("line break in a string");
            if (c == '\\')
                escape(out, false);
            else
                out.push_back(c);
        }
    }

    std::string literalString()
    {
        std::string out;
        if (startsWith("'''")) {
            m_pos += 3;
            if (peek() == '\r')
                next();
            if (peek() == '\n')
                next();
            for (;;) {
                if (startsWith("'''")) {
                    m_pos += 3;
                    while (peek() == '\'')
                        out.push_back(next());
                    return out;
                }
                out.push_back(next());
            }
        }
        expect('\'');
        for (;;) {
            const char c = next();
            if (c == '\'')
                return out;
            if (c == '\n')
                fail("line break in a string");
            out.push_back(c);
        }
    }

    Value number()
    {
        std::string tok;
        while (!eof() && (std::isalnum((unsigned char)peek()) || peek() == '+' || peek() == '-' || peek() == '.' || peek() == '_' || peek() == ':'))
            tok.push_back(next());
        if (tok.empty())
            fail("expected a value");
        Value v;
        const std::string body = (tok[0] == '+' || tok[0] == '-') ? tok.substr(1) : tok;
        const double sign = tok[0] == '-' ? -1.0 : 1.0;
        if (body == "inf" || body == "nan") {
            v.kind = Value::Float;
            v.f = body == "inf" ? sign * INFINITY : NAN;
            return v;
        }
        if (tok.find(':') != std::string::npos || (body.size() >= 5 && std::isdigit((unsigned char)body[0]) && body[4] == '-' && body.find('e') == std::string::npos && body.find('E') == std::string::npos && body.find("0x") != 0))
            fail("dates and times are not supported");
        std::string clean;
        for (size_t k = 0; k < tok.size(); ++k) {
            if (tok[k] == '_') {
                if (k == 0 || k + 1 == tok.size() || !std::isalnum((unsigned char)tok[k - 1]) || !std::isalnum((unsigned char)tok[k + 1]))
                    fail("misplaced underscore in a number");
                continue;
            }
            clean.push_back(tok[k]);
        }
        int base = 10;
        if (clean.size() > 2 && clean[0] == '0' && (clean[1] == 'x' || clean[1] == 'o' || clean[1] == 'b'))
            base = clean[1] == 'x' ? 16 : clean[1] == 'o' ? 8 : 2;
        char* end = nullptr;
        if (base != 10) {
            v.kind = Value::Int;
            v.i = (int64_t)std::strtoull(clean.c_str() + 2, &end, base);
        } else if (clean.find_first_of(".eE") != std::string::npos) {
            v.kind = Value::Float;
            v.f = std::strtod(clean.c_str(), &end);
        } else {
            v.kind = Value::Int;
            v.i = std::strtoll(clean.c_str(), &end, 10);
        }
        if (!end || *end != '\0' || !std::isdigit((unsigned char)body[0]))
            fail("malformed number '" + tok + "'");
        return v;
    }

    Value value()
    {
        const char c = peek();
        Value v;
        if (c == '"') {
            v.kind = Value::String;
            v.s = basicString();
        } else if (c == '\'') {
            v.kind = Value::String;
            v.s = literalString();
        } else if (c == '[') {
            next();
            v = makeArray();
            v.sealed = true;
            for (;;) {
                skipBlank(true);
                if (peek() == ']') {
                    next();
                    break;
                }
                v.arr->push_back(value());
                skipBlank(true);
                if (peek() == ',') {
                    next();
                } else if (peek() != ']') {
                    fail("expected ',' or ']' in an array");
                }
            }
        } else if (c == '{') {
            next();
            v = makeTable();
            skipBlank(false);
            if (peek() == '}') {
                next();
            } else {
                for (;;) {
                    const std::vector<std::string> path = keyPath();
                    skipBlank(false);
                    expect('=');
                    skipBlank(false);
                    Value inner = value();
                    assign(v, path, std::move(inner));
                    skipBlank(false);
                    if (peek() == ',') {
                        next();
                        skipBlank(false);
                    } else {
                        expect('}');
                        break;
                    }
                }
            }
            v.sealed = true;
        } else if (startsWith("true") && !bareKeyChar(m_pos + 4 < m_text.size() ? m_text[m_pos + 4] : ' ')) {
            m_pos += 4;
            v.kind = Value::Bool;
            v.b = true;
        } else if (startsWith("false") && !bareKeyChar(m_pos + 5 < m_text.size() ? m_text[m_pos + 5] : ' ')) {
            m_pos += 5;
            v.kind = Value::Bool;
            v.b = false;
        } else {
            v = number();
        }
        return v;
    }
};

} // namespace

Value parse(const std::string& text)
{
    return Parser(text).run();
}

Value parseFile(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open '" + path + "'");
    std::stringstream ss;
    ss << in.rdbuf();
    return parse(ss.str());
}

} // namespace srt::toml
