/*
 * srt_toml.h -- a small TOML reader for the configuration files of the command-line front end.
 *
 * The reference reads its configuration with toml++ (src/config.cpp:6-7, vendored third-party code); this is an
 * independent reader of the part of TOML v1.0 such files use: comments, bare / quoted / dotted keys, [tables],
 * [[arrays of tables]], basic and literal strings (single and multi-line), integers (decimal, 0x, 0o, 0b, with
 * underscores), floats (incl. inf / nan), booleans, arrays (multi-line, trailing comma) and inline tables.
 * Dates and times are not supported and are reported as errors.
 */
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace srt::toml {

struct Value;
using Table = std::map<std::string, Value>;
using Array = std::vector<Value>;

struct Value {
    enum Kind { None, Bool, Int, Float, String, ArrayKind, TableKind } kind = None;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Table> tab;
    bool sealed = false; /* inline tables and value arrays cannot be extended by later headers */

    [[nodiscard]] bool isNumber() const { return kind == Int || kind == Float; }
    /* member of a table, or a None value (also when this is not a table) */
    [[nodiscard]] const Value& operator[](const std::string& key) const;
    [[nodiscard]] const Value& at(size_t index) const;
    [[nodiscard]] size_t size() const { return kind == ArrayKind ? arr->size() : 0; }
};

/* Parses a document; throws std::runtime_error("line N: ...") on malformed input. */
Value parse(const std::string& text);
Value parseFile(const std::string& path);

} // namespace srt::toml
